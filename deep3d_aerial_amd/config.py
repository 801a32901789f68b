"""Kernel-selection switches of the host side: read from the environment ONCE, at import, into one table.

Every switch below selects between kernels that compute the same thing (A/B measurements, cross-checks in the tests, a
fall-back to MIOpen for the feature pyramids); none is needed in normal use.  Round 4: eight entries (thirty in round 3) --
the fourteen "kernel X off" booleans are ONE list (D3D_KERNELS_OFF, names in KERNELS), and the eleven switches nothing
used any more are gone with the choices they made frozen at their defaults.  The operators look them up here -- a dict
access -- instead of calling os.environ on every convolution (round 2 read ~50 of them per call).  Code that wants another
value at run time (the tests, the tools) sets `config.switches[name]` -- for a scope: `with config.override(name=value):` --
or calls `config.reload()` after changing os.environ.

The two pieces of per-call STATE that used to be process-global lists (the forced convolution precision and the "slice
loop" flag of the AdaMVS / RED-Net regularisers) live in `state`, a threading.local: one thread's context manager can no
longer flip the kernel selection -- and so the numerics -- of another thread's forward.
"""
import contextlib
import os
import threading

# name -> (default, what it selects)
SWITCHES = {
    "D3D_FORCE_PATH": ("", "sweep kernels: 'direct' | 'tiled' | 'window' forces one kernel family (tests, profiling); '' = dispatcher"),
    "D3D_CONV": ("mfma", "3-D / 2-D convolutions: 'mfma' (matrix cores) | 'mfma_slice' | 'direct' (vector-unit cross-check)"),
    "D3D_CONV_PRECISION": ("fp32", "default operand precision of the regularisers: 'fp32' | 'h16' (16-bit operands in the library's format; ops.set_conv_precision overrides)"),
    "D3D_CONV_C8X3": ("1", "fp32 mode of conv0 / conv2 / conv11 on the split-operand (3 x bf16) matrix-core kernels: '1' | '0' off | 'all' also the probability layer (slower there)"),
    "D3D_CONV2D_FP32": ("x3", "fp32 mode of the 2-D tile kernels: 'x3' (three-way bf16 split) | 'f32' (fp32 MFMA)"),
    "D3D_FEATURE_PRECISION": ("fp32", "feature pyramids: 'fp32' | 'follow' (the regularisers' precision)"),
    "D3D_FEATURE_CONV": ("mfma", "feature pyramids: 'mfma' (own kernels) | 'miopen'"),
    "D3D_KERNELS_OFF": ("", "comma-separated specialised kernels to leave out of the dispatch, so that the next-best kernel serves the "
                            "call (cross-checks in the tests, A/B measurements): " + "see KERNELS below"),
}

# The specialised kernels D3D_KERNELS_OFF can take out of the dispatch (each one's fallback computes the same thing):
KERNELS = {
    "co8": "8-output-channel 3-D streaming kernels",
    "c8": "bf16 z-streaming conv0 kernel",
    "co1": "single-output-channel probability kernel",
    "t2": "transposed stride-2 3-D streaming kernel",
    "cl": "channel-last bf16 activations between the layers of a CostRegNet (planar bf16 path instead)",
    "kzfold": "k_z-folded probability layer",
    "t2fold": "column-parity fold of conv11",
    "t2prob": "conv11 + probability layer of a CostRegNet in one kernel (bf16 mode, channel-last; the two layers as two launches instead)",
    "upskip": "fused 1x1 + upsample + skip (FPN lateral)",
    "conv2d_stream": "row-streamed 2-D vector kernel",
    "context_fused": "fused pooled-context heads of the AdaMVS pyramid",
    "conv2d_zs": "2-D tile kernels (stride 1 / 2 / transposed)",
    "conv2d_k5": "5x5 stride-2 tile kernel of the feature trunks",
    "conv2d_wide": "bf16 tile kernel for 64 | 128 input channels (RED-Net's coarse conv-GRU levels; the generic fp32 matrix-core kernel instead)",
    "gru_fused": "one-launch conv-GRU cell of bf16 mode (three tile-kernel launches instead)",
    "pair_streams": "the per-pair visibility passes of AdaMVS's first stage on three HIP streams (independent pairs; one stream instead)",
    "fpn_streams": "the feature pyramids of a view set on three HIP streams (the images are independent; one stream instead)",
    "hand_over": "record_stream on tensors that cross the streams of a multi-stream forward (measurement only: off is unsafe with two forwards in flight)",
    "convt_wide": "RED-Net's 64 -> 32 transposed layer as the wide stride-1 tile kernel over the zero-stuffed input (round-1 stream kernel instead)",
    "corr_cl8": "AdaMVS's weighted-correlation volume written by the sweep as CL8 16-bit cells and staged by the fused conv-GRU cell with 16-byte loads (planar fp32 volume instead)",
    "gru_gates_split": "ConvGRUCell2: the reset half alone + the update gate inside the state update (the two-output gates kernel instead)",
    "conv2d_k1": "1 x 1 convolutions of the feature pyramids on the fp32 streaming kernel (the round-1 matrix-core stream kernel instead)",
    "tail_same": "upconv1 + skip + same-resolution head + regression update in one kernel (AdaMVS's last stage, RED-Net; two launches instead)",
    "red_encoder": "RED-Net's encoder (conv1 .. conv3) for every depth slice of a stage in three batched launches before the loop (inside every slice instead)",
    "gru2_cell": "ConvGRUCell2 as one library call that issues its four launches (the four entry points from Python instead)",
    "red_graph": "RED-Net's slice loop of a stage as ONE captured HIP graph (the launch loop instead)",
    "slice_graph": "AdaMVS's slice loop of a stage as ONE captured HIP graph of three chains on three streams -- cell 1 of slice d + 2, cell 2 of slice d + 1, head / regression of slice d in flight together (the serial launch loop instead)",
    "red_streams": "the four conv-GRU levels of a RED-Net depth slice on four HIP streams (they depend on the encoder only; one stream instead)",
    "gn_fused": "GroupNorm statistics of ConvGRUCell2's convolutions accumulated in their epilogues (bf16 mode; d3d_groupnorm_stats over the stored tensor instead)",
    "conv0_pair": "conv0 of a feature trunk (3 -> 8 -> 8 at full resolution) in one kernel, the 8-channel map never written (the two layers as two launches instead)",
    "head_fused": "slice regulariser head + online regression update in one kernel (bf16 mode; the 8 -> 1 layer and the update as two launches instead)",
    "tail_fused": "upconv1 + head + online regression update of a depth slice in one kernel (bf16 mode, up-sampling stages; the transposed tile kernel and the fused head instead)",
    "fpn_split": "FPN output levels without the wide tensor",
}


def _read():
    return {name: os.environ.get(name, default) for name, (default, _) in SWITCHES.items()}


switches = _read()


def get(name):
    return switches[name]


_off_cache = ("", frozenset())


def off(kernel):
    """True when `kernel` (a key of KERNELS) is listed in D3D_KERNELS_OFF."""
    global _off_cache
    text = switches["D3D_KERNELS_OFF"]
    if text != _off_cache[0]:
        names = frozenset(n.strip() for n in text.split(",") if n.strip())
        unknown = names - set(KERNELS)
        if unknown:
            raise KeyError("D3D_KERNELS_OFF: unknown kernel(s) %s (known: %s)" % (", ".join(sorted(unknown)), ", ".join(sorted(KERNELS))))
        _off_cache = (text, names)
    if kernel not in KERNELS:
        raise KeyError("unknown kernel %r" % kernel)
    return kernel in _off_cache[1]


def reload():
    """Re-read every switch from os.environ (for code that changed the environment after import)."""
    switches.update(_read())


@contextlib.contextmanager
def override(**values):
    saved = {k: switches[k] for k in values}
    for k, v in values.items():
        if k not in SWITCHES:
            raise KeyError("unknown switch %s" % k)
        switches[k] = str(v)
    try:
        yield
    finally:
        switches.update(saved)


class _State(threading.local):
    conv_precision = None    # None: follow the D3D_CONV_PRECISION switch | "fp32" | "h16"
    tile_kernels = False     # inside a slice regulariser's plane loop (ops.slice_tile_kernels)


state = _State()
