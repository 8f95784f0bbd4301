"""MIOpen algorithm search for the classifier / discriminator convolutions.

The reference's drivers set ``cudnn.benchmark = True`` (code/dmcnet/train.py:118,
code/dmcnet_GAN/train.py:119): on ROCm that makes PyTorch ask MIOpen to *find* the fastest solver
per convolution shape instead of taking its heuristic pick.  On this path that is worth 6 % of the
dmcnet step (17.9 -> 16.7 ms) -- but the search costs 40 s for the dmcnet shapes and 12 minutes for
the GAN discriminator's on a fresh machine.  ``miopen_db/`` therefore ships the search results for
MI355X (gfx950, 256 CUs) as an MIOpen *user find-db*; with it the first step starts at once.
MIOpen ignores the files if its version differs (their names carry it) and simply searches again.
"""
import os
import shutil
import tempfile

import torch

_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def enable_find(use_shipped_db=True):
    """``torch.backends.cudnn.benchmark = True`` plus, unless the user already points MIOpen
    somewhere, ``MIOPEN_USER_DB_PATH`` = a writable copy of the shipped find-db.  Call before the
    first convolution of the process (MIOpen reads the variable when its handle is created)."""
    torch.backends.cudnn.benchmark = True
    # MIOpen's reference ("naive") direct solvers take part in every search they apply to; for the I3D stem's data
    # gradient (2 <- 64 channels, 7x7x7, stride 2) one evaluation of ConvDirectNaiveConvBwd lasts 69 s, and the search
    # is repeated by every new process even with the result in the find-db (~10 minutes before the first step).
    # They never win a search here, so they are switched off unless the user has set the variables.
    for var in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
                "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
        os.environ.setdefault(var, "0")
    if not use_shipped_db or "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(_DB_DIR):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    path = _DB_DIR
    # MIOpen appends to its user db, so it must be writable; with several ranks on one node each
    # process gets its own copy (no concurrent writers on the shipped files)
    if not os.access(path, os.W_OK) or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        path = tempfile.mkdtemp(prefix="dmc_miopen_db_")
        for name in os.listdir(_DB_DIR):
            shutil.copy(os.path.join(_DB_DIR, name), path)
    os.environ["MIOPEN_USER_DB_PATH"] = path
    return path
