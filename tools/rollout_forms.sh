#!/bin/bash
# open-loop rollout, the forms of pcgrl_set_rollout_form side by side (us per step, HIP events).  On the GPU box: bash tools/rollout_forms.sh
for WN in binary-narrow:4096 binary-narrow:8192 binary-narrow:16384 zelda-turtle:2048 zelda-turtle:4096 zelda-turtle:16384 sokoban-wide:2048 sokoban-wide:4096 sokoban-wide:16384; do
  W=${WN%%:*}; N=${WN##*:}
  for F in 1 -1 2; do python tools/rollout_bench.py $W $N $F ${1:-8,64}; done
done
