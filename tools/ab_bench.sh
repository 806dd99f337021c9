#!/bin/bash
# quick A/B pass over the BASELINE workloads (no CPU baseline, no rollout figure): us per step launch and roofline
# fraction per workload, plus binary-narrow at 65 536 envs.  On the GPU box, from the repo root:  bash tools/ab_bench.sh
for w in binary-narrow zelda-turtle sokoban-wide minecraft_3D_maze-narrow; do
  timeout 200 python bench.py --workload $w --no-cpu-baseline --rollout-steps 0 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$w', round(l['ms_per_step']*1e3,2),'us', round(l['roofline']['frac'],3))"
done
timeout 200 python bench.py --envs 65536 --no-cpu-baseline --rollout-steps 0 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('65536', round(l['ms_per_step']*1e3,2),'us', round(l['roofline']['frac'],3))"
timeout 120 python tools/solver_bench.py 2>&1 | tail -2
timeout 200 python bench.py --workload sokoban-wide-solver --steps 40 --warmup 8 --no-cpu-baseline --rollout-steps 0 2>&1 | tail -1 | cut -c1-250
