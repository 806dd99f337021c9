#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: for every BASELINE workload a kernel trace (+stats) and three separate PMC
# passes (never combined with a trace domain).  Run on the GPU box from the repo root:  bash tools/profile_all.sh r02
# Raw output goes to gpurun_out/prof_* and is condensed by tools/summarize_profiles.py into gpurun_out/summary/.
TAG=${1:-r03}
WORKLOADS=${2:-"binary-narrow zelda-turtle sokoban-wide minecraft_3D_maze-narrow"}
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
for W in $WORKLOADS; do
  S=3000; P=1000
  case "$W" in
    minecraft_3D_maze-narrow) S=1500; P=500;;
    binary_big*|zelda_big*|*stats-for-grids) S=1000; P=300;;
    binary_bigger*) S=600; P=200;;
    minecraft_3D_maze-narrow-15) S=6000; P=200;;  # (the step time moves along the 10 126-step episode: the trace covers most of one)
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${W}_kt -- python3 $R/bench.py --workload $W --steps $S --warmup 100 --no-cpu-baseline --rllib-adapter 0 --rollout-launches 20 > $R/gpurun_out/prof_${W}_kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${W}_fetch -- python3 $R/bench.py --workload $W --steps $P --warmup 100 --no-cpu-baseline --rllib-adapter 0 --rollout-steps 0 > $R/gpurun_out/prof_${W}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${W}_write -- python3 $R/bench.py --workload $W --steps $P --warmup 100 --no-cpu-baseline --rllib-adapter 0 --rollout-steps 0 > $R/gpurun_out/prof_${W}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_${W}_sq -- python3 $R/bench.py --workload $W --steps $P --warmup 100 --no-cpu-baseline --rllib-adapter 0 --rollout-steps 0 > $R/gpurun_out/prof_${W}_sq.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/prof_${W}_sq2 -- python3 $R/bench.py --workload $W --steps $P --warmup 100 --no-cpu-baseline --rllib-adapter 0 --rollout-steps 0 > $R/gpurun_out/prof_${W}_sq2.log 2>&1
done
cd $R
python3 tools/summarize_profiles.py $TAG gpurun_out/summary > gpurun_out/summary.log 2>&1
for W in $WORKLOADS; do rm -rf gpurun_out/prof_${W}_kt gpurun_out/prof_${W}_fetch gpurun_out/prof_${W}_write gpurun_out/prof_${W}_sq gpurun_out/prof_${W}_sq2; done
tail -5 gpurun_out/summary.log
