#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_probe.sh libA.so libB.so ... (alternating, 3 rounds; prints the sweep time of each run)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    S4B_LIB_PATH=$PWD/stan4bart_amd/csrc/$lib python tools/step_probe.py ${AB_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', 'sweep_us %.1f per_tree %.2f step_kernel %.2f' % (d['sweep_wall_us'], d['per_tree_wall_us'], d['stats_us']))"
  done
done
