#!/bin/bash
# every bench line behind profiles/<tag>_bench_lines.json (collected by tools/finish_round.py).  On the GPU box, from the
# repo root:  bash tools/bench_all.sh r4 [base|stock|sweep|all]
T=${1:-r5}
WHAT=${2:-all}
O=gpurun_out
mkdir -p $O
if [ "$WHAT" = "base" ] || [ "$WHAT" = "all" ]; then
  python bench.py > $O/bench_${T}_binary-narrow.log 2>&1
  for W in zelda-turtle sokoban-wide minecraft_3D_maze-narrow zelda-turtle-bfs binary-narrow-static binary-narrow-patch3x3; do
    python bench.py --workload $W > $O/bench_${T}_$W.log 2>&1
  done
  # solver-active sokoban: synchronous pcgrl_step (+ the rollout / sub-batch forms of the same workload), then asynchronous
  # stepping (pcgrl_step_ready) over a sweep of solver budgets; the budget-16 line carries the CPU baseline
  python bench.py --workload sokoban-wide-solver --steps 40 --warmup 8 --solver-forms 48 > $O/bench_${T}_sokoban-wide-solver.log 2>&1
  for B in 4 8 32 64 256; do
    python bench.py --workload sokoban-wide-solver --solver-budget $B --steps 3000 --warmup 300 --no-cpu-baseline > $O/bench_${T}_sokoban-wide-solver-async$B.log 2>&1
  done
  python bench.py --workload sokoban-wide-solver --solver-budget 16 --steps 3000 --warmup 300 > $O/bench_${T}_sokoban-wide-solver-async16.log 2>&1
  python bench.py --envs 65536 --no-cpu-baseline > $O/bench_${T}_binary-narrow-65536.log 2>&1
  python bench.py --graph-steps 0 --no-cpu-baseline > $O/bench_${T}_binary-narrow-eager.log 2>&1
  python bench.py --steps 20 --warmup 5 > $O/bench_${T}_driver_20_5.log 2>&1
  # the driver's command line through the N > 1 exchange (a world-size-1 RCCL group), and the older short-run protocols
  python bench.py --steps 20 --warmup 5 --force-collective --no-cpu-baseline --rllib-adapter 0 > $O/bench_${T}_driver_20_5_nccl1.log 2>&1
  python bench.py --steps 20 --warmup 5 --force-collective --exchange serial --no-cpu-baseline --rllib-adapter 0 --sub-batches "" > $O/bench_${T}_driver_20_5_nccl1_serial.log 2>&1
  for i in 2 3; do  # (single 160 us regions move by a few us from run to run: two more of each)
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rllib-adapter 0 --sub-batches "" --closed-loop-steps 0 > $O/bench_${T}_driver_20_5_run$i.log 2>&1
    python bench.py --steps 20 --warmup 5 --force-collective --no-cpu-baseline --rllib-adapter 0 --sub-batches "" --closed-loop-steps 0 > $O/bench_${T}_driver_20_5_nccl1_run$i.log 2>&1
  done
  python bench.py --steps 20 --warmup 5 --short-protocol gcd --no-cpu-baseline --rllib-adapter 0 --sub-batches "" > $O/bench_${T}_driver_20_5_gcd.log 2>&1
  python bench.py --steps 20 --warmup 5 --short-protocol one --no-cpu-baseline --rllib-adapter 0 --sub-batches "" > $O/bench_${T}_driver_20_5_one.log 2>&1
  python bench.py --steps 20 --warmup 5 --graph-steps 0 --no-cpu-baseline --rllib-adapter 0 --sub-batches "" > $O/bench_${T}_driver_20_5_eager.log 2>&1
  python tools/write_ceiling.py > $O/write_ceiling.json 2>/dev/null
  python tools/solver_bench.py > $O/solver_bench.log 2>&1
fi
if [ "$WHAT" = "stock" ] || [ "$WHAT" = "all" ]; then
  # the reference's stock task configs off 16x16 and the evolution driver's call pattern
  for W in binary_big-narrow binary_bigger-narrow zelda_big-turtle zelda_bigger-turtle minecraft_3D_maze-narrow-15 binary-narrow-evo binary-stats-for-grids zelda-stats-for-grids; do
    python bench.py --workload $W --cpu-seconds 8 > $O/bench_${T}_$W.log 2>&1
  done
fi
if [ "$WHAT" = "sweep" ] || [ "$WHAT" = "all" ]; then
  # saturation sweeps: envs x4, x16 of the BASELINE batch
  for WE in binary-narrow:16384 binary-narrow:262144 zelda-turtle:16384 zelda-turtle:65536 sokoban-wide:8192 sokoban-wide:32768 minecraft_3D_maze-narrow:4096 minecraft_3D_maze-narrow:16384; do
    W=${WE%%:*}; E=${WE##*:}
    python bench.py --workload $W --envs $E --steps 4000 --warmup 400 --no-cpu-baseline > $O/bench_${T}_$W-$E.log 2>&1
  done
fi
tail -qn1 $O/bench_${T}_*.log | cut -c1-160
