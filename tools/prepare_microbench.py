#!/usr/bin/env python3
"""bench.py's prepare_inputs sub-line alone (120 frames 256x340x7 uint8 -> 224x224 fp32 planes), for PMC passes:
    python tools/prepare_microbench.py [flow_ds_factor=16] [iters=20]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print(json.dumps(bench.bench_prepare(torch.device("cuda", 0), 120, ds, iters=iters)))
