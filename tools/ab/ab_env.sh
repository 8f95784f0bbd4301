#!/bin/bash
# usage: ab_env.sh VAR  -> alternate VAR=1 / VAR=0 bench runs, print mean/median ms
V=$1
for i in 1 2 3; do
  for val in 1 0; do
    env $V=$val python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$val', d['ms_per_step'], d['ms_per_step_median'])"
  done
done
