"""A few-thousand-parameter stand-in with the I3D's forward contract (``node`` / ``detach``) and the
parameter-name prefixes the reference's trainer routes by (``gen_flow_model``, ``discriminator``,
``conv3d_0c_1x1``, ``classifier``, anything else = pretrained trunk).  Used on BOTH sides of the
trainer-policy golden G10: the reference's ``model.fit`` loop (tests/golden/make_golden_i3d_trainer.py)
and this package's ``I3DTrainer`` (tests/test_host_cpu.py), so that the comparison pins the policy
(which optimizer steps when, with which learning rate and gradient scale) bit for bit."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class TinyI3D(nn.Module):
    def __init__(self, num_classes=5):
        super().__init__()
        self.gen_flow_model = nn.Conv2d(5, 2, 3, padding=1)
        self.discriminator = nn.Sequential(nn.Conv2d(2, 4, 3, stride=8, padding=1), nn.LeakyReLU(0.2),
                                           nn.Flatten(), nn.Linear(4 * 28 * 28, 2))
        self.conv3d_1a_7x7 = nn.Conv3d(2, 4, (1, 7, 7), stride=(1, 8, 8), padding=(0, 3, 3))
        self.conv3d_0c_1x1 = nn.Conv3d(4, 8, 1)
        self.classifier = nn.Linear(8, num_classes)

    def forward(self, inp, node="logit", detach=False):
        if node == "D":
            return self.discriminator(inp)
        b, c, t, h, w = inp.shape
        flow = self.gen_flow_model(inp.transpose(1, 2).reshape(-1, c, h, w)).reshape(b, t, 2, h, w).transpose(1, 2)
        x = flow.detach() if detach else flow
        x = F.relu(self.conv3d_1a_7x7(x))
        out = self.classifier(self.conv3d_0c_1x1(x).mean(dim=(2, 3, 4)))
        if node == "flow+logit":
            return out, flow
        return out


def build(seed):
    torch.manual_seed(seed)
    return TinyI3D()


def batches(seed, epochs, per_epoch, b=1, t=2, num_classes=5):
    """[[(data [b,7,t,224,224], target [b])]] -- 224 x 224 because the reference hard-codes it for the
    discriminator input (code/dmcnet_I3D/train/model.py:153-154)."""
    rs = np.random.RandomState(seed)
    return [[(torch.from_numpy(rs.standard_normal((b, 7, t, 224, 224)).astype(np.float32)),
              torch.from_numpy(rs.randint(0, num_classes, size=(b,)).astype(np.int64)))
             for _ in range(per_epoch)] for _ in range(epochs)]


def snapshot(net):
    return np.concatenate([p.detach().reshape(-1).numpy() for p in net.parameters()]).copy()
