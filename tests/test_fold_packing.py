"""Host logic of the folded implicit GEMM (ops._packed_fold): a numpy emulation of the kernel's contract
   out[co, g*s + b + f] = sum_t sum_ci wpack[t, ci, m=(f,co)] * in[ci, g*c + tap_t]   (zero padding)
must reproduce the oracle's convolutions for every fold / parity split the dispatcher can choose.  CPU only."""
import numpy as np
import pytest
import torch

from deep3d_aerial_amd import ops


def emulate(x, launches, od, transposed):
    Ci, D, H, W = x.shape
    out = None
    for (wpack, taps, T, M, mpad, tail) in launches:
        cz, cy, cx, sz, sy, sx, bz, by, bx, fz, fy, fx = tail
        wp = wpack.numpy()
        Co = M // (fz * fy * fx)
        if out is None:
            out = np.full((Co,) + od, np.nan, np.float32)
        tp = np.frombuffer(taps, np.int8).reshape(T, 3)
        assert (np.diff(tp[:, 0]) >= 0).all(), "taps must be sorted by z"
        assert wp.shape == (T, Ci, mpad) and (wp[:, :, M:] == 0).all()
        G = (D, H, W) if transposed else tuple(-(-od[i] // (fz, fy, fx)[i]) for i in range(3))
        xp = np.zeros((Ci, D + 16, H + 16, W + 16), np.float64)
        xp[:, 8:8 + D, 8:8 + H, 8:8 + W] = x
        for gz in range(G[0]):
            for gy in range(G[1]):
                for gx in range(G[2]):
                    col = np.zeros(M)
                    for t in range(T):
                        v = xp[:, 8 + gz * cz + tp[t, 0], 8 + gy * cy + tp[t, 1], 8 + gx * cx + tp[t, 2]]
                        col += v @ wp[t, :, :M]
                    m = 0
                    for a in range(fz):
                        for b in range(fy):
                            for c in range(fx):
                                oz, oy, ox = gz * sz + bz + a, gy * sy + by + b, gx * sx + bx + c
                                if oz < od[0] and oy < od[1] and ox < od[2]:
                                    assert np.isnan(out[0, oz, oy, ox]), "output written twice"
                                    out[:, oz, oy, ox] = col[m:m + Co]
                                m += Co
    assert not np.isnan(out).any(), "output not fully covered"
    return out


@pytest.mark.parametrize("Co", [1, 3, 8, 16])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv3d_fold_packing(oracle, Co, stride):
    rng = np.random.default_rng(Co * 10 + stride)
    x = rng.standard_normal((3, 4, 7, 9)).astype(np.float32)
    w = (0.3 * rng.standard_normal((Co, 3, 3, 3, 3))).astype(np.float32)
    want = oracle.conv3d_k3(x, w, stride=stride)
    got = emulate(x, ops._packed_fold(torch.from_numpy(w), False, stride), want.shape[1:], False)
    assert np.abs(got - want).max() <= 1e-5


@pytest.mark.parametrize("Co", [1, 8, 16])
def test_conv2d_fold_packing(oracle, Co):
    rng = np.random.default_rng(Co)
    x = rng.standard_normal((5, 9, 11)).astype(np.float32)
    w = (0.3 * rng.standard_normal((Co, 5, 3, 3))).astype(np.float32)
    for stride in (1, 2):
        want = oracle.conv2d_k3(x, w, None, stride=stride)
        # images go through the kernel as [C, rows, 1, W] (rows are the streamed planes)
        got = emulate(x[:, :, None, :], ops._packed_fold(torch.from_numpy(w), False, stride),
                      (want.shape[1], 1, want.shape[2]), False)
        assert np.abs(got[:, :, 0] - want).max() <= 1e-5


@pytest.mark.parametrize("Co", [1, 8, 16, 32])
def test_convtranspose_fold_packing(oracle, Co):
    rng = np.random.default_rng(40 + Co)
    x = rng.standard_normal((4, 3, 5, 6)).astype(np.float32)
    wt = (0.3 * rng.standard_normal((4, Co, 3, 3, 3))).astype(np.float32)
    want = oracle.convtranspose3d_k3s2(x, wt)
    launches = ops._packed_fold(torch.from_numpy(wt), True, 2)
    assert len(launches) == {1: 1, 8: 1, 16: 2, 32: 4}[Co]
    got = emulate(x, launches, want.shape[1:], True)
    assert np.abs(got - want).max() <= 1e-5
    wt2 = (0.3 * rng.standard_normal((4, Co, 3, 3))).astype(np.float32)
    want = oracle.convtranspose2d_k3s2(x[:, 0], wt2)
    got = emulate(x[:, 0][:, :, None, :], ops._packed_fold(torch.from_numpy(wt2), True, 2),
                  (want.shape[1], 1, want.shape[2]), True)
    assert np.abs(got[:, :, 0] - want).max() <= 1e-5


@pytest.mark.parametrize("Ci,Co,K,stride", [(3, 8, 3, 1), (8, 16, 5, 2), (16, 32, 5, 2), (32, 32, 1, 1), (8, 1, 1, 1)])
def test_odd_kernel_packing_matches_torch_conv2d(Ci, Co, K, stride):
    """1x1 and 5x5 stride-2 layers of the feature pyramids (module.py:657-679) through the same packer; the
    checker is torch's CPU convolution (the oracle only restates the 3x3 family)."""
    import torch.nn.functional as F

    rng = np.random.default_rng(Ci + Co + K)
    x = rng.standard_normal((Ci, 9, 14)).astype(np.float32)
    w = (0.3 * rng.standard_normal((Co, Ci, K, K))).astype(np.float32)
    want = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), stride=stride, padding=K // 2)[0].numpy()
    got = emulate(x[:, :, None, :], ops._packed_fold(torch.from_numpy(w), False, stride),
                  (want.shape[1], 1, want.shape[2]), False)
    assert np.abs(got[:, :, 0] - want).max() <= 3e-5
