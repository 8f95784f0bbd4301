#!/bin/bash
# correctness on small / ragged shapes, then the ResNet-18 shapes at 120 frames
B=./tools/ubench/bin/conv_x3s_bench
W=${1:-3}
for sh in "2 13 9 64 64" "3 7 7 128 64" "5 14 14 64 128" "37 14 14 64 64" "4 28 28 64 64" "3 56 56 64 64"; do $B $sh 3 $W | grep -E "^(fwd|dgrad|wgrad) *: max"; done
for sh in "3 14 14 64 128" "5 7 7 128 64"; do $B $sh 3 8 | grep "dgrad2: max"; done
for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do $B 120 $sh 30 $W | grep -E "^(fwd|dgrad|wgrad) *: x3s"; done
for sh in "28 28 64 128" "14 14 128 256" "7 7 256 512"; do $B 120 $sh 20 8 | grep "dgrad2: x3s"; done
