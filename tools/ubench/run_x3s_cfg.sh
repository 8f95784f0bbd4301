#!/bin/bash
B=./tools/ubench/bin/conv_x3s_bench
CFG=104 $B 3 28 28 64 64 3 3 | grep -E "^(fwd|dgrad) *: max" | cut -c1-110
for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do
  for c in 101 104 101 104; do echo "== $sh CFG=$c"; CFG=$c $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"; done
done
