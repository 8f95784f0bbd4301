mkdir -p gpurun_out/s5e; O=gpurun_out/s5e
timeout 600 python -m pytest tests/test_conv3d_gpu.py -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 400 python bench.py --config i3d --steps 10 --warmup 3 > $O/b_i3d_own.json 2> $O/b_i3d_own.err; cut -c1-300 $O/b_i3d_own.json; tail -3 $O/b_i3d_own.err
for cfg in 0 7 8 9 10; do echo "== conv_cfg=$cfg"; DMC_MB_MIOPEN=0 timeout 200 python tools/conv_microbench.py rn.layer2 rn.layer3 rn.layer4 conv_cfg=$cfg 2>&1 | grep -v amdgpu.ids; done > $O/mb_cfg.txt 2>&1; cat $O/mb_cfg.txt
