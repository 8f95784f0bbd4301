#!/bin/bash
# correctness on small / ragged shapes, then the ResNet-18 shapes at 120 frames
B=./tools/ubench/bin/conv_x3s_bench
W=${1:-3}
$B 2 13 9 64 64 3 $W
$B 3 7 7 128 64 3 $W
$B 5 14 14 64 128 3 $W
$B 4 28 28 64 64 3 $W
$B 3 56 56 64 64 3 $W
$B 120 56 56 64 64 20 $W
$B 120 28 28 128 128 20 $W
$B 120 14 14 256 256 20 $W
$B 120 7 7 512 512 20 $W
