#!/bin/bash
# usage: ab_env_cfg.sh VAR CONFIG
V=$1; C=$2
for i in 1 2 3; do
  for val in 1 0; do
    env $V=$val python bench.py --no-cpu-baseline --config $C 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$val', d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
