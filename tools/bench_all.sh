#!/bin/bash
# every bench line behind profiles/<tag>_bench_lines.json (collected by tools/finish_round.py).  On the GPU box, from the
# repo root:  bash tools/bench_all.sh r2
T=${1:-r3}
O=gpurun_out
python bench.py > $O/bench_${T}_binary-narrow.log 2>&1
for W in zelda-turtle sokoban-wide minecraft_3D_maze-narrow zelda-turtle-bfs binary-narrow-static binary-narrow-patch3x3; do
  python bench.py --workload $W > $O/bench_${T}_$W.log 2>&1
done
python bench.py --workload sokoban-wide-solver --steps 40 --warmup 8 > $O/bench_${T}_sokoban-wide-solver.log 2>&1
python bench.py --envs 65536 --no-cpu-baseline > $O/bench_${T}_binary-narrow-65536.log 2>&1
python bench.py --graph-steps 0 --no-cpu-baseline > $O/bench_${T}_binary-narrow-eager.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_${T}_driver_20_5.log 2>&1
python tools/write_ceiling.py > $O/write_ceiling.json 2>/dev/null
python tools/solver_bench.py > $O/solver_bench.log 2>&1
tail -qn1 $O/bench_${T}_*.log | cut -c1-160
