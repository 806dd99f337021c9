R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt3d -- python3 $R/bench.py --workload minecraft_3D_maze-narrow --steps 1500 --warmup 300 --no-cpu-baseline --rollout-launches 5 > $R/gpurun_out/kt3d.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, statistics
f = glob.glob("gpurun_out/kt3d/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "m3_kernel" in r["Kernel_Name"] and ("m3_kernel<0" in r["Kernel_Name"] or "M3Mode)0" in r["Kernel_Name"])]
d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
n = len(d)
print("launches", n, "mean %.2f median %.2f p10 %.2f p90 %.2f p99 %.2f max %.2f us" % (statistics.mean(d)/1e3, d[n//2]/1e3, d[n//10]/1e3, d[9*n//10]/1e3, d[99*n//100]/1e3, d[-1]/1e3))
per = [int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in range(n-1)]
per.sort(); print("start-to-start median %.2f us" % (per[len(per)//2]/1e3))
PY
rm -rf gpurun_out/kt3d
tail -1 gpurun_out/kt3d.log | cut -c1-250
