"""CPU checks of the drop-in boundary: the C-ABI library exports every symbol the header
declares, the ctypes binding covers them all, operators refuse CPU tensors (no fallback),
and the host mirror modules are checkpoint-compatible with the reference's."""
import ctypes
import re

import numpy as np
import pytest
import torch

from conftest import load_golden
from deep3d_aerial_amd import _lib


def _header_symbols():
    text = open(_lib.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    syms = _header_symbols()
    assert len(syms) >= 19
    raw = ctypes.CDLL(_lib.SO_PATH)
    for s in syms:
        assert hasattr(raw, s), "library does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    lib = _lib.load()
    assert lib.d3d_version() == _lib.ABI_VERSION
    assert lib.d3d_last_error() is not None
    # the production build carries no experiment knob: one wrong -D in the Makefile must not ship silently
    assert lib.d3d_build_flags() == b"", lib.d3d_build_flags()
    counts = (ctypes.c_ulonglong * 4)()
    assert lib.d3d_debug_dispatch_counts(counts, 1) == 0 and lib.d3d_debug_dispatch_counts(None, 0) == -1


def test_invalid_arguments_are_reported_not_thrown():
    lib = _lib.load()
    # null pointers / bad sizes must come back as D3D_ERR_INVALID_ARG before any launch
    assert lib.d3d_compose_projections(None, 3, None, None) == -1
    assert b"null" in lib.d3d_last_error()
    assert lib.d3d_softargmin_conf4(None, None, 0, 8, 4, 4, None, None, None) == -1
    assert lib.d3d_depth_range_samples(None, 0, 4, 0.0, 0, 0, None, None) == -1


def test_operators_refuse_cpu_tensors():
    from deep3d_aerial_amd import ops

    x = torch.zeros(4, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.homo_warp(x, torch.zeros(12), torch.zeros(2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.variance_volume([x, x], torch.zeros(1, 12), torch.zeros(2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.softargmin_conf4(torch.zeros(4, 8, 8), torch.zeros(4))


@pytest.mark.parametrize("tag", ["model_casmvsnet_v3", "model_adamvs_v3", "model_msrednet_v3"])
def test_state_dict_is_checkpoint_compatible(tag):
    """Same keys, same order, same shapes as the reference module (recorded in the golden file)."""
    from deep3d_aerial_amd.adamvs import Infer_AdaMVSNet
    from deep3d_aerial_amd.cas_mvsnet import Infer_CascadeMVSNet
    from deep3d_aerial_amd.msrednet import Infer_CascadeREDNet

    g = load_golden(tag)
    ctor = {"casmvsnet": Infer_CascadeMVSNet, "adamvs": Infer_AdaMVSNet, "msrednet": Infer_CascadeREDNet}[tag.split("_")[1]]
    net = ctor(num_depth=int(g["num_depth"]))
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["state_shapes"]]
    # DataParallel-style 'module.' prefixed checkpoints (predict.py:100-106) load after wrapping
    wrapped = torch.nn.DataParallel(net)
    wrapped.load_state_dict({"module." + k: v for k, v in sd.items()})
