#!/bin/bash
# random vs all-zero operands: same kernels, same instruction counts; the difference is the clock the chip sustains
B=./tools/ubench/bin/conv_x3s_bench
for sh in "56 56 64 64" "28 28 128 128"; do
  for f in 1 0 1 0; do echo "== $sh FILL=$f"; FILL=$f $B 120 $sh 40 7 | grep -E "^(fwd|dgrad|wgrad) *: x3s"; done
done
