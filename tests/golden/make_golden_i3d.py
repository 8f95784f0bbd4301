#!/usr/bin/env python3
"""G8: forward of the reference's own I3D (code/dmcnet_I3D/network/i3d.py, imported as-is) with the
per-frame DenseNetTiny generator and the Discriminator node, eval mode, seeded weights/inputs.
Run in the build container: ``python tests/golden/make_golden_i3d.py``."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dmc_oracle as O            # noqa: E402
from tests.golden.make_golden import checksum  # noqa: E402

sys.path.insert(0, "/root/reference/code/dmcnet_I3D/network")     # i3d.py does `import initializer`
spec = importlib.util.spec_from_file_location(
    "ref_i3d", "/root/reference/code/dmcnet_I3D/network/i3d.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

torch.set_num_threads(8)
net = ref.I3D(51, modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny",
              arch_d="Discriminator")
O.seeded_state_fill(net, seed=81).eval()
data = torch.from_numpy(np.random.RandomState(82).standard_normal((1, 7, 16, 224, 224)).astype(np.float32))
with torch.no_grad():
    logits, flow = net(data[:, :5], node="flow+logit")
    validity = net(flow.transpose(1, 2).reshape(-1, 2, 224, 224)[:4], node="D")
out = {"logits": logits.numpy(), "flow_checksum": checksum(flow),
       "flow_slice": flow[0, :, 3, 100:104, 50:66].numpy(), "validity": validity.numpy(),
       "keys": np.array(list(net.state_dict().keys()))}
np.savez_compressed(os.path.join(HERE, "g8_i3d_eval.npz"), **out)
print("G8 done", logits.shape, float(logits.abs().max()), len(out["keys"]))
