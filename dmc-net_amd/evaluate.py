"""Evaluation path: per-video scores with TSN consensus, score dump and late fusion.

Mirrors the reference's ``test.py`` (code/dmcnet/test.py:139-198: forward over
``test_segments x test_crops`` frames, mean over them, ``np.savez(scores, labels, names)`` with
the videos re-ordered by sorted name) and ``combine.py`` (code/dmcnet/combine.py:35-57: weighted
sum of the I-frame / MV / residual / DMC score files, arg-max accuracy).  The forward pass is the
training path's HIP generator + classifier in eval mode; nothing here needs a gradient.
"""
import numpy as np
import torch


@torch.no_grad()
def forward_video(model, input_mv, input_residual, test_segments, test_crops=1):
    """[B, segments*crops, C, H, W] inputs -> [B, num_class] numpy scores (mean over the frames)."""
    out = model(input_mv, input_residual)
    scores = out[0]
    scores = scores.view((-1, test_segments * test_crops) + tuple(scores.shape[1:])).mean(dim=1)
    return scores.cpu().numpy().copy()


@torch.no_grad()
def evaluate(model, loader, test_segments, test_crops=1, device="cuda:0"):
    """Returns (output, accuracy %) with ``output`` = [(video_scores [1,C], label), ...] in loader
    order, exactly what the reference accumulates before saving."""
    model.eval()
    output = []
    for _flow, input_mv, input_residual, label in loader:
        s = forward_video(model, input_mv.to(device, non_blocking=True),
                          input_residual.to(device, non_blocking=True), test_segments, test_crops)
        for b in range(s.shape[0]):
            output.append((s[b:b + 1], int(label[b])))
    pred = np.array([int(np.argmax(x[0])) for x in output])
    labels = np.array([x[1] for x in output])
    return output, float((pred == labels).mean() * 100.0) if len(output) else 0.0


def save_scores(path, output, name_list):
    """Reference layout: entries re-ordered by sorted video name (code/dmcnet/test.py:183-198)."""
    order = {e: i for i, e in enumerate(sorted(name_list))}
    n = len(output)
    scores, labels, names = [None] * n, [None] * n, [None] * n
    for i in range(n):
        idx = order[name_list[i]]
        scores[idx], labels[idx], names[idx] = output[i], output[i][1], name_list[i]
    obj = np.empty(n, dtype=object)
    for i, (s, l) in enumerate(scores):
        obj[i] = (s, l)
    np.savez(path, scores=obj, labels=np.array(labels), names=np.array(names))


def load_scores(path):
    """Reads a score file written by the reference (pickled object array of (scores, label)) or
    by :func:`save_scores` -> (scores [n, C] float, labels [n] int, names [n])."""
    with np.load(path, allow_pickle=True) as d:
        scores = np.array([np.asarray(s[0]).reshape(-1, np.asarray(s[0]).shape[-1])[0]
                           for s in d["scores"]])
        labels = np.array([int(s[1]) for s in d["scores"]], dtype=np.int64)
        names = np.array([str(x) for x in d["names"]])
    return scores, labels, names


def combine(iframe, mv, res, flow=None, wi=2.0, wm=1.0, wr=1.0, wf=1.0):
    """Late fusion of (scores, labels) pairs -> (accuracy in [0,1], combined scores)."""
    (si, li), (sm, lm), (sr, lr) = iframe, mv, res
    if not (np.array_equal(li, lm) and np.array_equal(li, lr)):
        raise ValueError("score files disagree on the labels")
    combined = si * wi + sm * wm + sr * wr
    if flow is not None:
        combined = combined + wf * flow[0]
    return float(np.sum(np.argmax(combined, axis=1) == li)) / len(li), combined
