#!/bin/bash
# What the small per-layer launches cost an I3D micro-step (graph replay): bench.py --config i3d on the measurement library with the
# launches switched off one family at a time (conv_ablate bits 256 / 512 / 1024: results wrong, timing only).
#   tools/i3d_launch_ablation.sh      (needs dmc-net_amd/libdmcnet_hip_measure.so: python dmc-net_amd/build.py --measure)
cd $GRAFT_REPO_ROOT
export DMC_HIP_LIB=$GRAFT_REPO_ROOT/dmc-net_amd/libdmcnet_hip_measure.so
for v in 0 256 512 1024 1792; do
  echo "conv_ablate=$v: $(timeout 300 python bench.py --config i3d --no-cpu-baseline --steps 30 --warmup 8 --option conv_ablate=$v 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("ms_per_step_median"))')"
done
