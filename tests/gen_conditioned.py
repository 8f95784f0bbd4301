"""Branch-conditioned comparison of the generator's gradients with an fp64 evaluation (shared by the GPU tests and
tools/gen_flip_lottery.py).

A parameter gradient of the dense estimator is a sum over millions of pixels in which every LeakyReLU contributes slope 1 or
0.1; a pre-activation within rounding of zero takes a different branch in fp32 than in fp64, and ONE flipped branch in a layer
with two output channels moves a bias gradient (a cancelling sum) by 1e-4 of its size: comparing a fp32 run with the plain fp64
run measures how many such coins fell differently, not the arithmetic (profiles/r4_gen_wino.txt: 1x .. 200x over seeds for every
kernel selection, the exact ones included).  Conditioned form: evaluate the fp64 backward with the slopes FORCED to the signs the
fp32 run itself took (read from its saved features) -- what is left is rounding -- and COUNT the sign disagreements separately.
Reference semantics: code/dmcnet/model.py:111-119 (conv + LeakyReLU(0.1)), :172-194 (dense stack)."""
import torch

WIDTHS = (8, 8, 6, 4, 2)


def split_features(feat):
    """[N,28,H,W] in physical order y0 | y1 | y2 | y3 | y4 -> list of five tensors."""
    out, c = [], 0
    for w in WIDTHS:
        out.append(feat[:, c:c + w])
        c += w
    return out


def oracle_features(o, x):
    """The features y0 .. y4 of a DenseEstimator `o` (any dtype) for input x, no autograd."""
    feats = []
    with torch.no_grad():
        xin = x
        for i in range(5):
            f = getattr(o, "conv_%d" % i)(xin)
            feats.append(f)
            xin = torch.cat((f, xin), 1)
    return feats


def forced_fp64(o64, x64, r64, masks, add_mv=True):
    """Forward + backward of the fp64 estimator with LeakyReLU slopes forced: masks[i] (bool, True = slope 1) replaces the sign
    of layer i's pre-activation.  Returns (output, [parameter gradients in named_parameters() order])."""
    for p in o64.parameters():
        p.grad = None
    xin = x64
    one, tenth = torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.1, dtype=torch.float64)
    for i in range(5):
        pre = getattr(o64, "conv_%d" % i)[0](xin)
        xin = torch.cat((pre * torch.where(masks[i], one, tenth), xin), 1)
    y = o64.predict_flow(xin)
    if add_mv:
        y = y + x64[:, :2]
    (y * r64).sum().backward()
    return y.detach(), [p.grad.clone() for p in o64.parameters()]


def conditioned_report(o, o64, x, r, hip_out, hip_grads, hip_saved, add_mv=True):
    """o: fp32 CPU estimator (its .grad fields are overwritten), o64: its double copy, x [N,5,H,W], r [N,2,H,W] (CPU fp32),
    hip_out / hip_grads / hip_saved: the device run's output, parameter gradients (named_parameters() order) and saved feature
    planes [N,28,H,W].  Returns a dict: per-parameter (e_hip, e_ref) against the fp64 backward forced to the run's OWN signs,
    the output errors, and the numbers of sign disagreements with the plain fp64 forward."""
    x64, r64 = x.double(), r.double()
    for p in o.parameters():
        p.grad = None
    yo = o(x) + (x[:, :2] if add_mv else 0)
    (yo * r).sum().backward()
    ref_grads = [p.grad.clone() for p in o.parameters()]
    f64 = oracle_features(o64, x64)
    m64 = [f > 0 for f in f64]
    m_ref = [f > 0 for f in oracle_features(o, x)]
    m_hip = [f.cpu() > 0 for f in split_features(hip_saved)]
    flips_ref = sum(int((a != b).sum()) for a, b in zip(m_ref, m64))
    flips_hip = sum(int((a != b).sum()) for a, b in zip(m_hip, m64))
    y_h, g_h = forced_fp64(o64, x64, r64, m_hip, add_mv)
    y_r, g_r = forced_fp64(o64, x64, r64, m_ref, add_mv)
    rep = {"flips_hip": flips_hip, "flips_ref": flips_ref, "params": {},
           "e_out_hip": float((hip_out.double().cpu() - y_h).abs().max() / y_h.abs().max()),
           "e_out_ref": float((yo.detach().double() - y_r).abs().max() / y_r.abs().max())}
    for (k, _), gh, gr, g64h, g64r in zip(o.named_parameters(), hip_grads, ref_grads, g_h, g_r):
        rep["params"][k] = (float((gh.double().cpu() - g64h).abs().max() / g64h.abs().max()),
                            float((gr.double() - g64r).abs().max() / g64r.abs().max()))
    return rep
