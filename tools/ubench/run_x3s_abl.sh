#!/bin/bash
B=./tools/ubench/bin/conv_x3s_bench
for abl in ${ABLS:-0 256 496}; do
  echo "=== ABL $abl"
  ABL=$abl $B 120 56 56 64 64 30 3 | grep -E "^(fwd|dgrad) *: x3s"
  ABL=$abl $B 120 28 28 128 128 30 3 | grep -E "^(fwd|dgrad) *: x3s"
done
for cfg in 101 102 103; do
  echo "=== CFG $cfg"
  for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do CFG=$cfg $B 120 $sh 30 3 | grep -E "^(fwd|dgrad) *: x3s"; done
done
