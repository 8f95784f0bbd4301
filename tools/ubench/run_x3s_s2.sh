#!/bin/bash
B=./tools/ubench/bin/conv_x3s_bench
$B 2 5 3 64 64 3 8 | grep dgrad2
$B 3 14 14 64 128 3 8 | grep dgrad2
$B 5 7 7 128 64 3 8 | grep dgrad2
$B 120 28 28 64 128 20 8 | grep dgrad2
$B 120 14 14 128 256 20 8 | grep dgrad2
$B 120 7 7 256 512 20 8 | grep dgrad2
