#!/bin/bash
# A/B: staggered transfer issue (default) vs every wave behind the barrier (conv_ablate 512), forward / data gradient
B=./tools/ubench/bin/conv_x3s_bench
$B 2 13 9 64 64 3 3 | grep -E "^(fwd|dgrad) *:" | cut -c1-120
$B 37 14 14 64 128 3 3 | grep -E "^(fwd|dgrad) *:" | cut -c1-120
for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do
  echo "== $sh staggered"; $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"
  echo "== $sh all waves"; ABL=512 $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"
  echo "== $sh staggered"; $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"
done
