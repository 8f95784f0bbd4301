#!/usr/bin/env python3
"""HIP-event timing of MaxPool3dTFPadding's forward on the I3D trunk's pool shapes: the key form (conv_cfg 0, stride-1 pools) against
the scan form (conv_cfg 9).   python tools/pool3d_microbench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import i3d, ops
lib=dmcnet_amd._lib.load()
for shape,k,s in (((3,192,32,28,28),(3,3,3),(1,1,1)),((3,480,16,14,14),(3,3,3),(1,1,1)),((3,64,32,112,112),(1,3,3),(1,2,2)),((3,480,32,28,28),(3,3,3),(2,2,2))):
    x=torch.relu(torch.randn(shape,device='cuda')).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    pool=i3d.MaxPool3dTFPadding(k,s)
    for cfg in (0,9):
        lib.dmc_set_option(b"conv_cfg",cfg)
        with torch.no_grad():
            for _ in range(3): pool(x)
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): pool(x)
            b.record(); torch.cuda.synchronize()
        print(shape,k,s,'cfg',cfg,'%.1f us'%(a.elapsed_time(b)/20*1e3))
    lib.dmc_set_option(b"conv_cfg",0)
