"""Kernel-selection switches of the host side: read from the environment ONCE, at import, into one table.

Every switch below selects between kernels that compute the same thing (A/B measurements, cross-checks in the tests, a
fall-back to MIOpen for the feature pyramids); none is needed in normal use.  The operators look them up here -- a dict
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
    "D3D_CONV_PRECISION": ("fp32", "default operand precision of the regularisers: 'fp32' | 'bf16' (ops.set_conv_precision overrides)"),
    "D3D_CONV_CO8": ("1", "0: the 8-output-channel streaming kernels off"),
    "D3D_CONV_C8": ("1", "0: the bf16 z-streaming conv0 kernel off"),
    "D3D_CONV_C8X3": ("1", "fp32 mode of conv0 / conv2 / conv11 on the split-operand (3 x bf16) matrix-core kernels: '1' | '0' off | 'all' also the probability layer (slower there)"),
    "D3D_CONV_CO1": ("1", "0: the single-output-channel probability kernel off"),
    "D3D_CONV_T2": ("1", "0: the transposed stride-2 streaming kernel off"),
    "D3D_CONV_CL": ("1", "0: channel-last bf16 activations off (planar bf16 path)"),
    "D3D_CONV_KZFOLD": ("1", "0: kz-folded probability layer off"),
    "D3D_CONV_T2_FOLD": ("1", "0: conv11 column-parity fold off"),
    "D3D_CONV1X1_UPSKIP": ("1", "0: fused 1x1 + upsample + skip off"),
    "D3D_CONV2D_STREAM": ("1", "0: row-streamed 2-D vector kernel off"),
    "D3D_CONTEXT_FUSED": ("1", "0: fused pooled-context heads of the AdaMVS pyramid off"),
    "D3D_CONV2D_ZS": ("1", "0: 2-D tile kernels off"),
    "D3D_CONV2D_FP32": ("x3", "fp32 mode of the 2-D tile kernels: 'x3' (three-way bf16 split) | 'f32' (fp32 MFMA)"),
    "D3D_CONV2D_ZS_MINPIX": (str(256 * 256), "smallest image (pixels) the bf16 2-D tile kernel takes"),
    "D3D_CONV2D_ZS_ALL": ("0", "1: 2-D tile kernels for every layer in fp32 mode too"),
    "D3D_CONV2D_ZS_SLICE": ("1", "0: slice-loop layers off the tile kernels"),
    "D3D_CONV2D_ZS_F32": ("1", "0: fp32-mode tile kernels off"),
    "D3D_CONVT2D_ZS_ALL": ("1", "0: transposed 2-D tile kernels only inside slice loops"),
    "D3D_CONVT2D_STUFF": ("1", "0: 48-channel transposed layers not as zero-stuffed convolutions"),
    "D3D_CONV2D_K5": ("1", "0: 5x5 stride-2 tile kernel off"),
    "D3D_CONV_NOFOLD": ("", "non-empty: never fold taps into K"),
    "D3D_CONV_FOLD_KB": ("48", "LDS budget (KB) of the tap fold"),
    "D3D_GRU_GATES": ("stream", "GRU gate kernel: 'stream' | 'separate'"),
    "D3D_GRU_FUSED": ("1", "0: the one-launch conv-GRU cell of bf16 mode off (three tile-kernel launches instead)"),
    "D3D_FEATURE_PRECISION": ("fp32", "feature pyramids: 'fp32' | 'follow' (the regularisers' precision)"),
    "D3D_FEATURE_CONV": ("mfma", "feature pyramids: 'mfma' (own kernels) | 'miopen'"),
    "D3D_FPN_SPLIT": ("1", "0: FPN output levels through the wide tensor"),
}


def _read():
    return {name: os.environ.get(name, default) for name, (default, _) in SWITCHES.items()}


switches = _read()


def get(name):
    return switches[name]


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
    conv_precision = None    # None: follow the D3D_CONV_PRECISION switch | "fp32" | "bf16"
    tile_kernels = False     # inside a slice regulariser's plane loop (ops.slice_tile_kernels)


state = _State()
