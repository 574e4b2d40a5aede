"""Helpers to import the read-only reference tree (this container only).

The reference needs `glog`, `onnx` and `k2.swoosh_*`, none of which are
installed here.  We register tiny in-memory stand-ins for those *third-party*
modules (never for reference code) so that `model.encoder.zipformer` etc. can
be imported and run to produce golden vectors.  The swoosh stand-ins follow
the formulas the reference itself states (model/layer/scaling.py:1340-1343,
1418-1423, 1496-1509).  Nothing from /root/reference is copied.
"""
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


def install_stubs():
    import torch

    sys.dont_write_bytecode = True
    if "glog" not in sys.modules:
        g = types.ModuleType("glog")
        for n in ("info", "warn", "warning", "error", "debug", "fatal", "check",
                  "setLevel"):
            setattr(g, n, lambda *a, **k: None)
        g.__spec__ = importlib.machinery.ModuleSpec("glog", None)
        sys.modules["glog"] = g
    if "onnx" not in sys.modules:
        o = types.ModuleType("onnx")
        o.__spec__ = importlib.machinery.ModuleSpec("onnx", None)
        sys.modules["onnx"] = o
    if "k2" not in sys.modules:
        k = types.ModuleType("k2")
        k.__spec__ = importlib.machinery.ModuleSpec("k2", None)

        def _sl(x):
            return torch.logaddexp(torch.zeros((), dtype=x.dtype), x - 4.0) - 0.08 * x - 0.035

        def _sr(x):
            return torch.logaddexp(torch.zeros((), dtype=x.dtype), x - 1.0) - 0.08 * x - 0.313261687

        k.swoosh_l = _sl
        k.swoosh_l_forward = _sl
        k.swoosh_r = _sr
        k.swoosh_r_forward = _sr
        k.swoosh_l_forward_and_deriv = lambda x: (_sl(x), torch.sigmoid(x - 4.0) - 0.08)
        k.swoosh_r_forward_and_deriv = lambda x: (_sr(x), torch.sigmoid(x - 1.0) - 0.08)
        sys.modules["k2"] = k
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)
