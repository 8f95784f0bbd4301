#!/usr/bin/env python3
"""Score-fusion fixtures from the reference's SHIPPED outputs (exp_my/**/**_score_model_best.npz,
the only machine-checkable artefacts the reference holds).

Run in the build container: ``python tests/golden/make_golden_scores.py``.

* The reference's own ``code/dmcnet/combine.py`` is executed on them (its ``np.load`` needs
  ``allow_pickle=True`` on modern numpy, and ``np.alltrue`` is gone in numpy 2 -- both supplied by
  a shim, the arithmetic is untouched) and the accuracies it prints are stored as the expected
  values for every split of the three shipped experiments.
* For HMDB-51 split 1 the four score files are stored as compact float32 ``[n, 51]`` arrays
  (data, ~1.2 MB) so that the fusion can be recomputed from inputs on any machine.
"""
import contextlib
import io
import os
import re
import runpy
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
EXP = os.path.join(REF, "exp_my")


def ref_combine(iframe, mv, res, flow):
    """Accuracy printed by the reference's combine.py for these four files."""
    load = np.load
    np.load = lambda p, *a, **k: load(p, *a, **dict(k, allow_pickle=True))
    had = hasattr(np, "alltrue")
    if not had:
        np.alltrue = np.all
    argv = sys.argv
    sys.argv = ["combine.py", "--iframe", iframe, "--mv", mv, "--res", res, "--flow", flow]
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(REF, "code", "dmcnet", "combine.py"), run_name="__main__")
    finally:
        sys.argv, np.load = argv, load
        if not had:
            del np.alltrue
    m = re.search(r"Accuracy: ([0-9.]+) \((\d+)\)", buf.getvalue())
    return float(m.group(1)), int(m.group(2))


def files(dataset, exp, split):
    cov = os.path.join(EXP, "%s_coviar" % dataset)
    sub = (lambda r: os.path.join(cov, r if dataset == "hmdb51" else "ucf101_" + r, split,
                                  r + "_score_model_best.npz"))
    return sub("iframe"), sub("mv"), sub("residual"), \
        os.path.join(EXP, "%s_%s" % (dataset, exp), split, "mv_score_model_best.npz")


def compact(path):
    d = np.load(path, allow_pickle=True)
    scores = np.array([s[0][0] for s in d["scores"]], dtype=np.float32)
    labels = np.array([int(s[1]) for s in d["scores"]], dtype=np.int64)
    return scores, labels, np.array([str(n) for n in d["names"]])


def main():
    out = {}
    table = []
    for dataset, exp in (("hmdb51", "gen_flow"), ("hmdb51", "gan"), ("ucf101", "gen_flow")):
        for split in ("split1", "split2", "split3"):
            f = files(dataset, exp, split)
            acc, n = ref_combine(*f)
            table.append((dataset, exp, split, acc, n))
            print(dataset, exp, split, acc, n)
    out["expected_names"] = np.array(["%s/%s/%s" % t[:3] for t in table])
    out["expected_acc"] = np.array([t[3] for t in table], dtype=np.float64)
    out["expected_n"] = np.array([t[4] for t in table], dtype=np.int64)
    i, m, r, d = files("hmdb51", "gen_flow", "split1")
    g = files("hmdb51", "gan", "split1")[3]
    for tag, p in (("iframe", i), ("mv", m), ("residual", r), ("dmc", d), ("dmc_gan", g)):
        s, l, names = compact(p)
        out["hmdb51_split1_" + tag] = s
        out["hmdb51_split1_labels_" + tag] = l
    out["hmdb51_split1_names"] = names
    np.savez_compressed(os.path.join(HERE, "g7_score_fusion.npz"), **out)


if __name__ == "__main__":
    main()
