#!/bin/bash
# A/B of two builds of the library on the observation-bound sizes (development): us per step launch per workload / batch, and the
# LDS bank-conflict rate of the step kernel.  bash tools/ab_obs.sh libA.so libB.so   (files inside control_pcgrl_amd/csrc)
R=$(pwd)
COMMON="--no-cpu-baseline --rollout-steps 0 --rllib-adapter 0 --closed-loop-steps 0 --sub-batches="
for L in "$@"; do
  export PCGRL_LIB=$R/control_pcgrl_amd/csrc/$L
  for WE in binary-narrow:4096 binary-narrow:16384 binary-narrow:65536 zelda-turtle:4096 zelda-turtle:16384 sokoban-wide:2048 sokoban-wide:32768; do
    W=${WE%%:*}; E=${WE##*:}; S=4000; [ $E -ge 16384 ] && S=1500
    timeout 300 python bench.py --workload $W --envs $E --steps $S --warmup 200 $COMMON 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$L $W $E', round(l['ms_per_step']*1e3,3),'us', round(l['roofline']['frac'],3))"
  done
  for WE in binary-narrow:4096 zelda-turtle:4096 sokoban-wide:2048; do
    W=${WE%%:*}; E=${WE##*:}
    (export TMPDIR=/tmp; cd /tmp; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lds_pmc -- python3 $R/bench.py --workload $W --steps 300 --warmup 50 $COMMON > /dev/null 2>&1)
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/lds_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print("$L $W LDS bank-conflict rate", round(acc["SQ_LDS_BANK_CONFLICT"] / max(acc["SQ_LDS_IDX_ACTIVE"], 1), 3))
PY
    rm -rf gpurun_out/lds_pmc
  done
done
