#!/bin/bash
# A/B: persistent (default) vs non-persistent (CFG 111) forward / data gradient on the four layer shapes
B=./tools/ubench/bin/conv_x3s_bench
$B 2 13 9 64 64 3 3 | grep -E "^(fwd|dgrad) *:"
$B 5 14 14 64 128 3 3 | grep -E "^(fwd|dgrad) *:"
$B 7 7 7 64 64 3 3 | grep -E "^(fwd|dgrad) *:"
for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do
  echo "== $sh persistent"; $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *:"
  echo "== $sh non-persistent"; CFG=111 $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"
done
