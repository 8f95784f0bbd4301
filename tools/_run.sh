mkdir -p gpurun_out/s5l; timeout 900 python tools/_diag.py > gpurun_out/s5l/diag.txt 2>&1; grep -v amdgpu.ids gpurun_out/s5l/diag.txt | tail -60
