mkdir -p gpurun_out/s5c; O=gpurun_out/s5c
timeout 600 python -m pytest tests/test_conv3d_gpu.py -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python tools/conv3d_microbench.py > $O/mb3d.txt 2>&1; cat $O/mb3d.txt
