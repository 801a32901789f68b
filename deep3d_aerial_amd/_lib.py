"""ctypes binding of the C-ABI library (include/deep3d_planesweep.h).

The library is built in-tree by deep3d_aerial_amd/csrc/Makefile (see __graft_entry__.build).
There is NO fallback: if the shared object is missing or lacks a symbol, importing the
operators raises, and every operator refuses tensors that are not on the GPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# D3D_LIBRARY: load another build of the library (tools/run_ab.sh links its -D variants to a scratch path and points this
# at them, so the in-tree production library is never overwritten by an experiment).  Read once, at import.
SO_PATH = os.environ.get("D3D_LIBRARY") or os.path.join(CSRC, "libdeep3d_planesweep.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "deep3d_planesweep.h")

ABI_VERSION = 10

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> argtypes; restype is int except where noted.  Mirrors include/deep3d_planesweep.h.
SIGNATURES = {
    "d3d_version": [],
    "d3d_last_error": [],
    "d3d_compose_projections": [_vp, _i, _vp, _vp],
    "d3d_debug_force_path": [_i],
    "d3d_debug_dispatch_counts": [_vp, _i],
    "d3d_build_flags": [],
    "d3d_h16_format": [],
    "d3d_compose_projections_f64": [_vp, _i, _vp, _vp],
    "d3d_homo_warp_f64coord": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_sweep_workspace_bytes": [_i, _i, _i, _i, _i, _i],  # returns size_t
    "d3d_sweep_workspace_bytes_for": [_i, _i, _i, _i, _i, _i, _i],  # returns size_t
    "d3d_homo_warp": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_variance_volume": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_variance_volume_planes": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_variance_volume_f16": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_variance_volume_cl_h16": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_variance_volume_cl8_h16": [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_pair_corr_mean": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_weighted_corr": [ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_softargmin_conf4": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "d3d_online_regress_update": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "d3d_online_regress_finalize": [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "d3d_depth_range_samples": [_vp, _i, _i, _f, _i, _i, _vp, _vp],
    "d3d_resize_bilinear": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_co8": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_c8_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_zs_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_c1_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose3d_k3s2_zs_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose3d_k3s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3_zs_h16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3_zs_f32": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3_zs_bf16x3": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_avgpool2d_4_8": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "d3d_conv3x3_bias_border": [_vp, _vp, _i, _i, _i, _vp],
    "d3d_conv2d_k1_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv1x1_context": [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3s2_zs_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3s2_zs_h16_batched": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _i64, _vp, _vp],
    "d3d_conv2d_k3_wide_h16": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "d3d_slice_head_regress_h16": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "d3d_gru_cell_fused_h16": [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_gru_cell_fused_cl8_h16": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_weighted_corr_cl8_h16": [ctypes.POINTER(_vp), _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp],
    "d3d_convtranspose2d_k3s2_zs_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose2d_k3s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose2d_k4s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k5s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3s2_zs_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose2d_k3s2_zs_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3_cl_h16": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "d3d_conv3d_k3_c1_cl_h16": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv3d_k3s2_cl_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose3d_k3s2_cl_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "d3d_conv2d_k3_zs_h16_gn": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "d3d_conv2d_k3_wide_h16_gn": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "d3d_convtranspose2d_k3s2_wide_h16": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3_pair3_bf16x3": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "d3d_slice_tail_regress_h16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "d3d_slice_tail_regress_same_h16": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "d3d_conv3d_k3s2_zs_bf16x3": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose3d_prob_cl_h16": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp],
    "d3d_volume_planar_to_cl_h16": [_vp, _i, _sz, _vp, _vp],
    "d3d_volume_cl_h16_to_planar": [_vp, _i, _sz, _vp, _vp],
    "d3d_convtranspose3d_k3s2_co8": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose3d_k3s2": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv1x1_upskip": [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3_stream": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv2d_k3": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_convtranspose2d_k3s2": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "d3d_conv_gemm_f32": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                          _i, _i, _i, _i, _i, ctypes.c_char_p, _vp, _vp],
    "d3d_conv_fold_f32": [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                          ctypes.POINTER(ctypes.c_int), _i, ctypes.c_char_p, _vp, _vp],
    "d3d_conv_fold_h16": [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                           ctypes.POINTER(ctypes.c_int), _i, ctypes.c_char_p, _vp, _vp],
    "d3d_gru_gates": [_vp, _vp, _i, _i64, _vp, _vp, _vp],
    "d3d_gru_update": [_vp, _vp, _vp, _i64, _vp, _vp],
    "d3d_groupnorm_stats": [_vp, _i64, _i, _vp, _vp],
    "d3d_gru_gates_gn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, ctypes.c_float, _i, _vp, _vp, _vp],
    "d3d_gru_update_gn": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, ctypes.c_float, _i, _vp, _vp],
    "d3d_gru_reset_gn": [_vp, _vp, _vp, _vp, _vp, _i, _i64, ctypes.c_float, _i, _vp, _vp],
    "d3d_gru_update_gates_gn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, ctypes.c_float, _i, _vp, _vp],
    "d3d_gru2_cell_gn_h16": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_softargmin_conf4_var": [_vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _vp, _vp, _vp],
    "d3d_uncertainty_samples": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "d3d_pair_softmax_max": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "d3d_consistency_check": [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_double), _i, _i, _i, _i, ctypes.c_double,
                              _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_fusion_ref_init": [_vp, _vp, ctypes.POINTER(ctypes.c_double), _i, _i, _vp, _vp, _vp, _vp, _vp],
    "d3d_fusion_accumulate": [_vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_double), _i, _i, _i, _i,
                              ctypes.c_double, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_fusion_finalize": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "d3d_fusion_points_scratch_bytes": [_i, _i],  # returns size_t
    "d3d_fusion_mark_points": [_vp, _vp, _i, _i, _i, ctypes.POINTER(ctypes.c_double), _vp, _vp, _vp, _vp],
    "d3d_fusion_gather_points": [_vp, _vp, ctypes.POINTER(_vp), _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "d3d_flip_rows": [ctypes.POINTER(_vp), _i, _i, _i, _vp, _vp],
    "d3d_center_image_u8": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
}


class LibraryMissing(RuntimeError):
    pass


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of the C-ABI library, in-tree (cross-compiles on CPU)."""
    cmd = ["make", "-C", CSRC, "-j4", "libdeep3d_planesweep.so"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libdeep3d_planesweep.so failed")
    return SO_PATH


_lib = None


def load():
    """Load the library and bind every symbol the header declares. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LibraryMissing(
            "%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C deep3d_aerial_amd/csrc`. There is no CPU fallback." % SO_PATH)
    lib = ctypes.CDLL(SO_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LibraryMissing("symbol %s missing from %s" % (name, SO_PATH)) from e
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name in ("d3d_last_error", "d3d_build_flags", "d3d_h16_format") else
                      ctypes.c_size_t if name in ("d3d_sweep_workspace_bytes", "d3d_sweep_workspace_bytes_for", "d3d_fusion_points_scratch_bytes") else ctypes.c_int)
    if lib.d3d_version() != ABI_VERSION:
        raise LibraryMissing("ABI version mismatch: library %d, binding %d" % (lib.d3d_version(), ABI_VERSION))
    _lib = lib
    return lib


ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP = -1, -2, -3


def h16_format():
    """"f16" | "bf16": the 16-bit operand format the loaded library was built with (d3d_h16_format, ABI 9)."""
    return load().d3d_h16_format().decode("ascii")


def code_objects(path=None):
    """The gfx950 code objects (ELF images, as bytes) bundled in the built library's .hip_fatbin section."""
    import struct

    data = open(path or SO_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl]
            q += 24 + tl
            if b"amdgcn" not in triple or size == 0:
                continue
            elf = data[pos + off:pos + off + size]
            if elf[:4] == b"\x7fELF":
                yield elf
        pos = data.find(magic, pos + len(magic))


def kernel_code_sha256(symbol_prefix, path=None):
    """SHA-256 of the gfx950 machine code of ONE kernel in the built library: the bytes of the (unique) function symbol whose
    mangled name starts with `symbol_prefix`, read from the code objects bundled in the .so's .hip_fatbin section.  A counter
    profile of a kernel stays valid exactly as long as this hash does -- an edit elsewhere in the same source file does not
    invalidate it, a change of compiler flags does (bench.profiled_traffic, tools/profile_round.sh).  None if not found."""
    import hashlib
    import struct

    found = []
    for elf in code_objects(path):
        shoff, = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
        for sh in secs:
            if sh[1] != 2:   # SHT_SYMTAB
                continue
            strtab = secs[sh[6]]
            for k in range(sh[5] // sh[9]):
                st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, sh[4] + k * sh[9])
                if (st_info & 0xF) != 2 or st_size == 0 or st_shndx >= len(secs):   # STT_FUNC
                    continue
                end = elf.index(b"\0", strtab[4] + st_name)
                if elf[strtab[4] + st_name:end].decode("ascii", "replace").startswith(symbol_prefix):
                    sec = secs[st_shndx]
                    o = sec[4] + (st_value - sec[3])
                    found.append(hashlib.sha256(elf[o:o + st_size]).hexdigest())
    return found[0] if len(found) == 1 else None


def check(rc, what):
    if rc != 0:
        msg = load().d3d_last_error().decode("utf-8", "replace")
        kind = {-1: "invalid argument", -2: "unsupported", -3: "HIP error"}.get(rc, "error %d" % rc)
        raise RuntimeError("%s failed (%s): %s" % (what, kind, msg))
