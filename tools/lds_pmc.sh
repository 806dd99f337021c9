R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lds_pmc -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --rollout-steps 0 > $R/gpurun_out/lds_pmc.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/lds_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print({k: v / max(n[k], 1) for k, v in acc.items()})
print("conflict rate", acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1))
PY
rm -rf gpurun_out/lds_pmc
