#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: for every workload a kernel trace (+stats) and separate PMC passes (never
# combined with a trace domain).  Run on the GPU box from the repo root:
#   bash tools/profile_all.sh r05 "binary-narrow zelda-turtle@16384 binary-narrow+rollout"
# An entry is  workload[@envs][+rollout]:  @envs = a batch off the BASELINE size (saturation sweeps), +rollout = the same passes
# over the open-loop rollout kernel (pcgrl_rollout, 64 steps per launch) instead of the step kernel.
# Raw output goes to gpurun_out/prof_* and is condensed by tools/summarize_profiles.py into gpurun_out/summary/.
TAG=${1:-r05}
ENTRIES=${2:-"binary-narrow zelda-turtle sokoban-wide minecraft_3D_maze-narrow"}
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
# PROFILE_EXTRA: more bench.py arguments for every entry of the call (e.g. "--solver-budget 16" for sokoban-wide-solver)
COMMON="--no-cpu-baseline --rllib-adapter 0 --closed-loop-steps 0 --sub-batches= $PROFILE_EXTRA"
for ENT in $ENTRIES; do
  RO=0; case "$ENT" in *+rollout) RO=1;; esac
  WE=${ENT%+rollout}
  W=${WE%@*}; E=""; case "$WE" in *@*) E=${WE#*@};; esac
  S=3000; P=1000
  case "$W" in
    minecraft_3D_maze-narrow) S=1500; P=500;;
    binary_big*|zelda_big*|*stats-for-grids) S=1000; P=300;;
    binary_bigger*|zelda_bigger*) S=600; P=200;;
    minecraft_3D_maze-narrow-15) S=6000; P=200;;  # (the step time moves along the 10 126-step episode: the trace covers most of one)
  esac
  if [ -n "$E" ] && [ "$E" -ge 16384 ]; then S=$((S / 3)); P=$((P / 3)); fi
  ARGS="--workload $W $COMMON"
  [ -n "$E" ] && ARGS="$ARGS --envs $E"
  if [ $RO = 1 ]; then  # few step launches, many rollout launches: the rollout kernel dominates the passes
    KT="$ARGS --steps 20 --warmup 5 --rollout-launches 100"; PM="$ARGS --steps 20 --warmup 5 --rollout-launches 40"
  else
    # (no rollouts in a step-kernel entry: on the larger 2-D maps pcgrl_rollout issues step launches, which would mix into the
    # dominant kernel's statistics with their own, larger output footprint)
    # (a long warm-up: the first hundreds of launches of a fresh process run while the clocks are still ramping -- three kernel
    # traces of the same command read 6.01 / 6.29 / 6.02 us mean with 100 warm-up launches)
    KT="$ARGS --steps $S --warmup 2000 --rollout-steps 0"; PM="$ARGS --steps $P --warmup 100 --rollout-steps 0"
  fi
  D=$R/gpurun_out/prof_${ENT}
  rocprofv3 --kernel-trace --stats --output-format csv -d ${D}_kt -- python3 $R/bench.py $KT > ${D}_kt.log 2>&1
  # (the profiler's per-dispatch overhead has two modes from run to run -- six traces of the same binary: means 6.00 / 6.05 / 6.05 /
  # 6.09 / 6.38 / 6.54 us --, so the trace is taken three times; the summary keeps all three means and reports the least disturbed run)
  if [ $RO = 0 ]; then
    for REP in 2 3; do
      rocprofv3 --kernel-trace --stats --output-format csv -d ${D}_ktrep$REP -- python3 $R/bench.py $KT > ${D}_ktrep$REP.log 2>&1
    done
  fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${D}_fetch -- python3 $R/bench.py $PM > ${D}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${D}_write -- python3 $R/bench.py $PM > ${D}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d ${D}_sq -- python3 $R/bench.py $PM > ${D}_sq.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS --output-format csv -d ${D}_sq2 -- python3 $R/bench.py $PM > ${D}_sq2.log 2>&1
done
cd $R
python3 tools/summarize_profiles.py $TAG gpurun_out/summary > gpurun_out/summary.log 2>&1
for ENT in $ENTRIES; do rm -rf gpurun_out/prof_${ENT}_ktrep2 gpurun_out/prof_${ENT}_ktrep3 gpurun_out/prof_${ENT}_kt gpurun_out/prof_${ENT}_fetch gpurun_out/prof_${ENT}_write gpurun_out/prof_${ENT}_sq gpurun_out/prof_${ENT}_sq2; done
tail -5 gpurun_out/summary.log
