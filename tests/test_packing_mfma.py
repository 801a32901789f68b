"""Host logic of the bf16 / fp32 matrix-core kernels of round 2: the weight packers of ops.py (operand fragments in the lane
order of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32) against a numpy emulation of each kernel's K-index contract
    fragment[kb][tile][lane][j]  <->  B[K = kblock * kb + krows * (lane >> 4) + j][column = 16 * tile + (lane & 15)]
and, through it, against the oracle's convolutions (weights are chosen exactly representable in bf16).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from deep3d_aerial_amd import ops


def bf16_exact(rng, shape):
    """Random weights that survive the bf16 rounding of the packers unchanged."""
    return (rng.integers(-32, 33, shape) / 64.0).astype(np.float32)


def dense_from_fragments(frag, kblock, dt=None):
    """[nkb][ntn][64][krows] fragments (16-bit bits or fp32) -> dense B [K, ntn * 16].  16-bit fragments are in the library's h16
    format (ops.h16_dtype()) unless `dt` says otherwise (the pieces of a three-way split are bfloat16 in every build)."""
    f = frag.view(dt or ops.h16_dtype()).float().numpy() if frag.dtype == torch.int16 else frag.numpy()
    if f.ndim == 3:                                    # fp32 fragments: one K row per lane
        f = f[..., None]
    nkb, ntn, _, kr = f.shape
    assert kblock == 4 * kr
    B = np.zeros((nkb * kblock, ntn * 16), np.float32)
    for lane in range(64):
        for j in range(kr):
            B[np.arange(nkb)[:, None] * kblock + kr * (lane >> 4) + j, np.arange(ntn)[None, :] * 16 + (lane & 15)] = f[:, :, lane, j]
    return B


def conv_from_B(x, B, taps, Co, out_shape, stride=1):
    """Direct evaluation of out[co, o] = sum_{t, ci} B[t * Ci + ci, co] * x[ci, o * stride + tap_t] with zero padding."""
    Ci = x.shape[0]
    nd = x.ndim - 1
    pad = 2
    xp = np.pad(x.astype(np.float64), [(0, 0)] + [(pad, pad)] * nd)
    out = np.zeros((Co,) + out_shape)
    for t, off in enumerate(taps):
        sl = tuple(slice(pad + o, pad + o + stride * n, stride) for o, n in zip(off, out_shape))
        out += np.tensordot(B[t * Ci:(t + 1) * Ci, :Co].T.astype(np.float64), xp[(slice(None),) + sl], axes=(1, 0))
    return out


@pytest.mark.parametrize("Ci,Co", [(8, 8), (16, 8), (32, 8), (16, 16), (32, 32), (64, 64), (8, 16)])
def test_pack_conv3d_fragments(Ci, Co):
    """ops._pack_c8_bf16 (d3d_conv3d_k3_zs_h16 / _cl_bf16, stride 1 and 2): per k_z slice K = (k_y, k_x, c_in)."""
    rng = np.random.default_rng(Ci + Co)
    w = bf16_exact(rng, (Co, Ci, 3, 3, 3))
    x = rng.standard_normal((Ci, 3, 4, 5)).astype(np.float32)
    frag = ops._pack_c8_bf16(torch.from_numpy(w))
    assert frag.dtype == torch.int16 and frag.shape[0] == 3
    out = np.zeros((Co, 3, 4, 5))
    for kz in range(3):
        B = dense_from_fragments(frag[kz], 32)
        assert (B[9 * Ci:] == 0).all() and (B[:, Co:] == 0).all()       # padded K rows / columns are zero
        taps = [(kz - 1, ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
        out += conv_from_B(x, B, taps, Co, (3, 4, 5))
    assert np.abs(out - oracle.conv3d_k3(x, w, None)).max() <= 1e-4


def test_pack_probability_layer_kz_folded():
    """ops._pack_c8_kzfold_bf16 (d3d_conv3d_k3_c1_cl_h16): the k_z slices are columns 0..2 of one tile."""
    rng = np.random.default_rng(3)
    w = bf16_exact(rng, (1, 8, 3, 3, 3))
    x = rng.standard_normal((8, 4, 3, 6)).astype(np.float32)
    B = dense_from_fragments(ops._pack_c8_kzfold_bf16(torch.from_numpy(w))[:, None], 32)
    assert (B[:, 3:] == 0).all()
    out = np.zeros((1, 4, 3, 6))
    for kz in range(3):
        taps = [(kz - 1, ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
        out += conv_from_B(x, B[:, kz:kz + 1], taps, 1, (4, 3, 6))
    assert np.abs(out - oracle.conv3d_k3(x, w, None)).max() <= 1e-4


def _transposed_from_classes(x, classes, Co, nd):
    """Assemble a stride-2 transposed convolution from its per-parity dense convolutions: classes[p] = (B, taps)."""
    sp = x.shape[1:]
    out = np.zeros((Co,) + tuple(2 * n for n in sp))
    for p, (B, taps) in classes.items():
        y = conv_from_B(x, B, taps, Co, sp)
        out[(slice(None),) + tuple(slice(pi, None, 2) for pi in p)] = y
    return out


@pytest.mark.parametrize("Ci,Co", [(16, 8), (32, 16), (64, 32), (16, 16)])
def test_pack_convtranspose3d_parity_classes(Ci, Co):
    """ops._pack_t2_bf16 (d3d_convtranspose3d_k3s2_zs_h16 / _cl_bf16): eight parity classes, taps (dz,dy,dx) dz-major."""
    rng = np.random.default_rng(Ci * 3 + Co)
    w = bf16_exact(rng, (Ci, Co, 3, 3, 3))
    x = rng.standard_normal((Ci, 2, 3, 4)).astype(np.float32)
    frag = ops._pack_t2_bf16(torch.from_numpy(w))
    ntn = max(Co, 16) // 16
    classes, base = {}, 0
    for p in range(8):
        pz, py, px = p >> 2, (p >> 1) & 1, p & 1
        taps = [(dz, dy, dx) for dz in range(1 + pz) for dy in range(1 + py) for dx in range(1 + px)]
        nkb = (len(taps) * Ci + 31) // 32
        B = dense_from_fragments(frag[base:base + nkb * ntn * 64].reshape(nkb, ntn, 64, 8), 32)
        base += nkb * ntn * 64
        classes[(pz, py, px)] = (B, taps)
    assert base == frag.shape[0]
    assert np.abs(_transposed_from_classes(x, classes, Co, 3) - oracle.convtranspose3d_k3s2(x, w, None)).max() <= 1e-4


def test_pack_convtranspose3d_x_folded():
    """ops._pack_t2_fold_bf16 (conv11, channel-last): rows 0..7 = even output column, rows 8..15 = odd one, per (pz, py)."""
    rng = np.random.default_rng(11)
    Ci, Co = 16, 8
    w = bf16_exact(rng, (Ci, Co, 3, 3, 3))
    x = rng.standard_normal((Ci, 2, 3, 4)).astype(np.float32)
    frag = ops._pack_t2_fold_bf16(torch.from_numpy(w))
    classes, base = {}, 0
    for c in range(4):
        pz, py = c >> 1, c & 1
        taps = [(dz, dy, dx) for dz in range(1 + pz) for dy in range(1 + py) for dx in range(2)]
        nkb = (len(taps) * Ci + 31) // 32
        B = dense_from_fragments(frag[base:base + nkb * 64].reshape(nkb, 1, 64, 8), 32)
        base += nkb * 64
        classes[(pz, py, 0)] = (B[:, 0:8], taps)
        classes[(pz, py, 1)] = (B[:, 8:16], taps)
    assert base == frag.shape[0]
    assert np.abs(_transposed_from_classes(x, classes, Co, 3) - oracle.convtranspose3d_k3s2(x, w, None)).max() <= 1e-4


@pytest.mark.parametrize("Ci,Co,packer,kblock", [(16, 16, "_pack_z2_bf16", 32), (32, 32, "_pack_z2_bf16", 32), (8, 1, "_pack_z2_bf16", 32),
                                                  (16, 16, "_pack_z2_f32", 4), (32, 8, "_pack_z2_f32", 4)])
def test_pack_conv2d_fragments(Ci, Co, packer, kblock):
    """ops._pack_z2_bf16 / _pack_z2_f32 (d3d_conv2d_k3_zs_* and the stride-2 form): K = (k_y, k_x, c_in)."""
    rng = np.random.default_rng(Ci + Co + kblock)
    w = bf16_exact(rng, (Co, Ci, 3, 3))
    x = rng.standard_normal((Ci, 6, 9)).astype(np.float32)
    B = dense_from_fragments(getattr(ops, packer)(torch.from_numpy(w)), kblock)
    taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    assert np.abs(conv_from_B(x, B, taps, Co, (6, 9)) - oracle.conv2d_k3(x, w, None)).max() <= 1e-4
    assert np.abs(conv_from_B(x, B, taps, Co, (3, 5), stride=2) - oracle.conv2d_k3(x, w, None, stride=2)).max() <= 1e-4


@pytest.mark.parametrize("Ci,Co", [(16, 16), (32, 24), (8, 1)])
def test_pack_conv2d_three_way_bf16_split(Ci, Co):
    """ops._pack_z2_bf16x3 (d3d_conv2d_k3_zs_bf16x3): hi + mid + lo is the fp32 weight exactly, each part is a bf16 number, and
    the six products the kernel takes (against the same split of the activations) reproduce the fp32 convolution."""
    rng = np.random.default_rng(Ci * 3 + Co)
    w = (rng.standard_normal((Co, Ci, 3, 3)) * np.exp(rng.uniform(-6, 6, (Co, Ci, 3, 3)))).astype(np.float32)
    x = (rng.standard_normal((Ci, 6, 9)) * np.exp(rng.uniform(-6, 6, (Ci, 6, 9)))).astype(np.float32)
    ws = [t.numpy() for t in ops._split3_bf16(torch.from_numpy(w))]
    xs = [t.numpy() for t in ops._split3_bf16(torch.from_numpy(x))]
    assert np.array_equal(ws[0] + ws[1] + ws[2], w) and np.array_equal(xs[0] + xs[1] + xs[2], x)
    for part in ws + xs:
        assert np.array_equal(torch.from_numpy(part).to(torch.bfloat16).to(torch.float32).numpy(), part)
    frag = ops._pack_z2_bf16x3(torch.from_numpy(w))
    assert frag.shape[0] == 3
    taps = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
    B = [dense_from_fragments(frag[s], 32, torch.bfloat16) for s in range(3)]
    got = np.zeros((Co, 6, 9), np.float64)
    for sa, sb in [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]:
        got += conv_from_B(xs[sa], B[sb], taps, Co, (6, 9)).astype(np.float64)
    Bsum = B[0].astype(np.float64) + B[1].astype(np.float64) + B[2].astype(np.float64)
    want = conv_from_B(x, Bsum, taps, Co, (6, 9))                           # float64 arithmetic
    assert np.abs(want - oracle.conv2d_k3(x, w, None)).max() <= 1e-5 * np.abs(want).max()
    scale = conv_from_B(np.abs(x), np.abs(Bsum), taps, Co, (6, 9))
    assert (np.abs(got - want) <= 2.0 ** -22 * scale).all()


def test_pack_convtranspose2d_three_way_split_is_three_bf16_packings():
    """ops._pack_t2d_bf16x3: [hi | mid | lo] x _pack_t2d_bf16, and the parts add up to the fp32 weight."""
    rng = np.random.default_rng(5)
    w = torch.from_numpy((rng.standard_normal((16, 8, 3, 3)) * np.exp(rng.uniform(-4, 4, (16, 8, 3, 3)))).astype(np.float32))
    frag = ops._pack_t2d_bf16x3(w)
    parts = ops._split3_bf16(w)
    assert frag.shape[0] == 3 and torch.equal(parts[0] + parts[1] + parts[2], w)
    for s_, part in enumerate(parts):
        assert torch.equal(frag[s_], ops._pack_t2d_bf16(part, torch.bfloat16))


@pytest.mark.parametrize("Ci,Co,packer,kblock", [(16, 8, "_pack_t2d_bf16", 32), (8, 1, "_pack_t2d_bf16", 32), (32, 16, "_pack_t2d_f32", 4),
                                                  (8, 1, "_pack_t2d_f32", 4)])
def test_pack_convtranspose2d_parity_classes(Ci, Co, packer, kblock):
    """ops._pack_t2d_bf16 / _pack_t2d_f32 (d3d_convtranspose2d_k3s2_zs_*): four parity classes, taps (dy,dx) dy-major."""
    rng = np.random.default_rng(Ci * 5 + Co + kblock)
    w = bf16_exact(rng, (Ci, Co, 3, 3))
    x = rng.standard_normal((Ci, 3, 5)).astype(np.float32)
    frag = getattr(ops, packer)(torch.from_numpy(w))
    classes, base = {}, 0
    for c in range(4):
        py, px = c >> 1, c & 1
        taps = [(dy, dx) for dy in range(1 + py) for dx in range(1 + px)]
        K = len(taps) * Ci
        nkb = (K + 31) // 32 if kblock == 32 else K // 4
        rows = nkb * 64 if kblock == 32 else nkb          # bf16: [fragment lane][8 values]; fp32: [K block][64 lanes]
        blk = frag[base:base + rows]
        B = dense_from_fragments(blk.reshape(nkb, 1, 64, 8) if kblock == 32 else blk.reshape(nkb, 1, 64), kblock)
        base += rows
        classes[(py, px)] = (B, taps)
    assert base == frag.shape[0]
    assert np.abs(_transposed_from_classes(x, classes, Co, 2) - oracle.convtranspose2d_k3s2(x, w, None)).max() <= 1e-4


def test_upsampled_conv_weight_is_a_k4_transposed_convolution():
    """ops.upsampled_conv_weight: conv2d(nearest_x2(f), w3, padding 1) == conv_transpose2d(f, ., stride 2, padding 1), zero
    padding included (the input side of module.fpn_output)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    f = torch.randn(1, 5, 6, 7, generator=g, dtype=torch.float64)
    w3 = torch.randn(4, 5, 3, 3, generator=g)
    want = F.conv2d(F.interpolate(f, scale_factor=2, mode="nearest"), w3.double(), padding=1)
    got = F.conv_transpose2d(f, ops.upsampled_conv_weight(w3).double(), stride=2, padding=1)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("Ci,Co", [(32, 8), (16, 1), (32, 16), (8, 12)])
def test_pack_convtranspose2d_k4_parity_classes(Ci, Co):
    """ops._pack_t2d_k4_bf16 (16 columns per parity class) and _pack_t2d_k4fold_bf16 (C_out <= 8: both column parities in one
    tile over three patch columns) against torch's ConvTranspose2d(k 4, stride 2, padding 1): output 2i + p reads input
    i - 1 + p + d through kernel index 3 - p - 2d."""
    import torch.nn.functional as F
    rng = np.random.default_rng(Ci + Co)
    w = torch.from_numpy(bf16_exact(rng, (Ci, Co, 4, 4)))
    x = rng.standard_normal((Ci, 3, 5)).astype(np.float32)
    want = F.conv_transpose2d(torch.from_numpy(x).double()[None], w.double(), stride=2, padding=1)[0].numpy()
    out = np.zeros((Co, 6, 10))
    if Co <= 8:
        frag = ops._pack_t2d_k4fold_bf16(w)
        nkb = (6 * Ci + 31) // 32
        assert frag.shape[0] == 2 * nkb * 64
        for py in range(2):
            B = dense_from_fragments(frag[py * nkb * 64:(py + 1) * nkb * 64].reshape(nkb, 1, 64, 8), 32)
            taps = [(dy + py - 1, dxx - 1) for dy in range(2) for dxx in range(3)]      # patch column dxx = input column m - 1 + dxx
            for px in range(2):
                out[:, py::2, px::2] = conv_from_B(x, B[:, px * 8:px * 8 + 8], taps, Co, (3, 5))
    else:
        frag = ops._pack_t2d_k4_bf16(w)
        nkb = (4 * Ci + 31) // 32
        assert frag.shape[0] == 4 * nkb * 64
        for c in range(4):
            py, px = c >> 1, c & 1
            B = dense_from_fragments(frag[c * nkb * 64:(c + 1) * nkb * 64].reshape(nkb, 1, 64, 8), 32)
            taps = [(dy + py - 1, dx + px - 1) for dy in range(2) for dx in range(2)]
            out[:, py::2, px::2] = conv_from_B(x, B, taps, Co, (3, 5))
    assert np.abs(out - want).max() <= 1e-4


def test_pack_conv2d_5x5_fragments():
    """ops._pack_z2_bf16 on a 5 x 5 weight (d3d_conv2d_k5s2_zs_bf16x3): K = (k_y, k_x, c_in) over 25 taps."""
    import torch.nn.functional as F
    rng = np.random.default_rng(55)
    w = bf16_exact(rng, (16, 8, 5, 5))
    x = rng.standard_normal((8, 9, 12)).astype(np.float32)
    B = dense_from_fragments(ops._pack_z2_bf16(torch.from_numpy(w)), 32)
    taps = [(ky - 2, kx - 2) for ky in range(5) for kx in range(5)]
    want = F.conv2d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double(), stride=2, padding=2)[0].numpy()
    assert np.abs(conv_from_B(x, B, taps, 16, (5, 6), stride=2) - want).max() <= 1e-4


def test_noted_depth_range_is_used_until_the_tensor_changes():
    """ops.note_depth_range / depth_range_host: the drivers take (dmin, dmax) from the host pair a caller left for THIS tensor
    (predict_views, bench.py: no device -> host copy per view), and read the tensor again once it was written to, or for any
    other tensor."""
    import torch
    from deep3d_aerial_amd import ops

    dv = torch.tensor([[400.0, 800.0]])
    assert ops.depth_range_host(dv) == (400.0, 800.0)            # nothing noted: read from the tensor
    ops.note_depth_range(dv, 1.0, 2.0)
    assert ops.depth_range_host(dv) == (1.0, 2.0)                # the noted pair (deliberately not the tensor's: proves which one is used)
    other = dv.clone()
    assert ops.depth_range_host(other) == (400.0, 800.0)         # another tensor: its own values
    dv[0, 0] = 500.0                                             # written to: the note is stale
    assert ops.depth_range_host(dv) == (500.0, 800.0)
    n = len(ops._depth_ranges)
    ops.note_depth_range(other, 3.0, 4.0)
    del other
    assert len(ops._depth_ranges) <= n                           # a dead tensor's note goes with it
