#!/bin/bash
# instruction / wait counters of the solver launch of tools/solver_phase.py --plain (two separate --pmc passes)
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/sk_pmc1 -- python3 $R/tools/solver_phase.py --plain 64 > $R/gpurun_out/sk_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/sk_pmc2 -- python3 $R/tools/solver_phase.py --plain 64 > $R/gpurun_out/sk_pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/sk_pmc1", "gpurun_out/sk_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); dur = {}
        for r in csv.DictReader(open(f)):
            if "stats_for_grids" in r["Kernel_Name"]:
                acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        # the last dispatch is the 64-level launch
        ids = sorted({k[0] for k in acc}, key=int)
        if ids:
            last = ids[-1]
            print(d, {k[1]: v for k, v in acc.items() if k[0] == last})
PY
rm -rf gpurun_out/sk_pmc1 gpurun_out/sk_pmc2
