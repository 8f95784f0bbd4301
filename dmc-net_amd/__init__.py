"""dmcnet_amd: MI355X-native implementation of DMC-Net's data-parallel training hot path.

Public surface mirrors what the reference's drivers import (``from model import Model``,
``from dataset import CoviarDataSet``): see ``model``, ``dataset``, ``train``, ``ddp``.
The arithmetic of the path lives in ``libdmcnet_hip.so`` (``csrc/``, C ABI in
``include/dmcnet_hip.h``), bound through ``_lib`` / ``ops``.
"""
from . import _lib, ops  # noqa: F401
from .model import Model  # noqa: F401

__version__ = "0.1.0"
