#!/bin/bash
# build the fused-forward profile harness (run from the repo root); prints the kernel's register / instruction counts
cd "$(dirname "$0")" && mkdir -p bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDMC_MEASURE -Wno-unused-value -I ../../include -I ../../dmc-net_amd/csrc gen_fused_prof.hip -o bin/gen_fused_prof -save-temps=obj 2>&1 | grep -E "error" -A5 | head -30
S=bin/gen_fused_prof-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.vgpr_count|vgpr_spill|private_segment_fixed" $S | head -5
echo "mfma $(grep -c v_mfma $S)  v_mov $(grep -c v_mov_b32 $S)  barriers $(grep -c s_barrier $S)  ds_read $(grep -c ds_read $S)  code bytes $(ls -l bin/gen_fused_prof-hip-amdgcn-amd-amdhsa-gfx950.out )"
rm -f bin/gen_fused_prof-h* bin/gen_fused_prof.hip-*
# the same harness without the per-step clock reads (s_memtime round trips cost ~10 % of a step): the timing to quote
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDMC_MEASURE -DFZ_NOPROF -Wno-unused-value -I ../../include -I ../../dmc-net_amd/csrc gen_fused_prof.hip -o bin/gen_fused_time 2>&1 | grep -E "error" -A5 | head
