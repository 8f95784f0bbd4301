#!/bin/bash
# builds the stand-alone timing of gen_wgrad.hip in its measurement variants (run from anywhere): bin/gen_wgrad_time[_noload|_nomfma]
cd "$(dirname "$0")" && mkdir -p bin
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -I ../../include -I ../../dmc-net_amd/csrc"
/opt/rocm/bin/hipcc $F gen_wgrad_time.hip -o bin/gen_wgrad_time 2>&1 | grep -E "error" -A5 | head -20
/opt/rocm/bin/hipcc $F -DWR_NO_LOAD gen_wgrad_time.hip -o bin/gen_wgrad_time_noload 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_NO_MFMA gen_wgrad_time.hip -o bin/gen_wgrad_time_nomfma 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_NO_MFMA -DWR_NO_LOAD gen_wgrad_time.hip -o bin/gen_wgrad_time_neither 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_PROF gen_wgrad_time.hip -o bin/gen_wgrad_time_prof 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_PROF -DWR_NO_LOAD gen_wgrad_time.hip -o bin/gen_wgrad_time_prof_noload 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_PRIO=2 gen_wgrad_time.hip -o bin/gen_wgrad_time_prio 2>&1 | grep -E "error" -A5 | head
/opt/rocm/bin/hipcc $F -DWR_PRIO=2 -DWR_NO_LOAD gen_wgrad_time.hip -o bin/gen_wgrad_time_prio_noload 2>&1 | grep -E "error" -A5 | head
ls bin/gen_wgrad_time*
