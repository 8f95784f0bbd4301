cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DW2_TIMING -I include -I dmc-net_amd/csrc -c dmc-net_amd/csrc/stem3d_bf16.hip -o /tmp/s3t.o 2>/dev/null
objs=$(ls dmc-net_amd/csrc/*.o | grep -v measure_ | grep -v stem3d_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_w2t.so $objs /tmp/s3t.o 2>/dev/null
DMC_HIP_LIB=/tmp/lib_w2t.so timeout 200 python tools/stem3d_w2_phases.py
