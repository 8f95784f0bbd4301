#!/bin/bash
# alternate the built library against tools/ab/lib_old.so
for i in 1 2 3; do
  for lib in new old; do
    if [ $lib = old ]; then export DMC_HIP_LIB=$PWD/tools/ab/lib_old.so; else unset DMC_HIP_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
