"""ctypes binding of libviai_hip.so (the C ABI declared in include/viai_hip.h).

The HIP library is the product; there is NO CPU fallback.  Every wrapper raises
if the library is missing or a launch returns a non-zero hipError_t.
"""
from __future__ import annotations

import ctypes as C

import torch  # noqa: F401  -- FIRST: torch bundles its own libamdhip64; our .so must bind to that already-loaded runtime
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libviai_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class ViaiLibraryError(RuntimeError):
    pass


# ABI version THIS file's SIGNATURES / struct mirrors were written against: bumped together with them.  load() compares it with the
# library, and with the committed header where that is present (a source checkout), so a stale _lib.py cannot call a rebuilt .so.
ABI_VERSION = 17


def _header_abi_version():
    """VIAI_ABI_VERSION of include/viai_hip.h, or None when the package was installed without the header"""
    import re
    try:
        with open(os.path.join(os.path.dirname(_HERE), "include", "viai_hip.h")) as f:
            m = re.search(r"#define\s+VIAI_ABI_VERSION\s+(\d+)", f.read())
    except OSError:
        return None
    if m is None:
        raise ViaiLibraryError("include/viai_hip.h does not define VIAI_ABI_VERSION")
    return int(m.group(1))


class Conv2dDesc(C.Structure):
    """mirror of `viai_conv2d` (include/viai_hip.h)."""
    _fields_ = [(n, C.c_int) for n in (
        "N", "IH", "IW", "C1", "C2", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "transposed", "dh", "dw", "ph2", "pw2")]


class PackJob(C.Structure):
    """mirror of `viai_pack_job`."""
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("n_out", C.c_int), ("k_in", C.c_int), ("taps", C.c_int), ("frag", C.c_int),
                ("s_no", C.c_long), ("s_ki", C.c_long), ("blk0", C.c_int), ("nblk", C.c_int)]


class WnLayer(C.Structure):
    """mirror of `viai_wn_layer`."""
    _fields_ = [(n, C.c_void_p) for n in ("w_conv", "b_conv", "w_c", "b_c", "w_out", "b_out", "w_skip", "b_skip", "ring")] + \
               [("dilation", C.c_int), ("ring_len", C.c_int), ("g_add", C.c_void_p), ("w_stage", C.c_void_p), ("b_stage", C.c_void_p)]


class WnSynth(C.Structure):
    """mirror of `viai_wn_synth`."""
    _fields_ = [(n, C.c_int) for n in ("B", "C", "G", "S", "cin", "n_layers", "out_ch", "T", "n_test")] + [("log_scale_min", C.c_float)] + \
               [("layers", C.POINTER(WnLayer))] + \
               [(n, C.c_void_p) for n in ("w_first", "b_first", "w_l1", "b_l1", "w_l2", "b_l2", "cond", "test_inputs", "u1", "u2",
                                          "out", "z", "skips", "yhat_dbg", "step", "z2")] + [("fused", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_D = C.c_double
_CP = C.POINTER(Conv2dDesc)
_IP = C.POINTER(C.c_int)

# name -> (restype, argtypes); must list every symbol declared in include/viai_hip.h
SIGNATURES = {
    "viai_abi_version": (_I, []),
    "viai_conv2d_out_hw": (_I, [_CP, _IP, _IP]),
    "viai_conv2d_packed_floats": (C.c_size_t, [_CP]),
    "viai_conv2d_pack_fwd": (_I, [_CP, _P, _P, _P]),
    "viai_conv2d_pack_dgrad": (_I, [_CP, _P, _P, _P]),
    "viai_conv2d_stat_geom": (_I, [_CP, _IP, _IP]),
    "viai_conv2d_fwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _P]),
    "viai_conv2d_fwd_f16_ok": (_I, [_CP]),
    "viai_conv2d_fwd_amax": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "viai_absmax": (_I, [_P, _L, _P, _P]),
    "viai_bn_act_fwd_amax": (_I, [_P, _P, _P, _P, _L, _I, _I, _F, _P, _P]),
    "viai_conv2d_dgrad": (_I, [_CP, _P, _P, _P, _P, _P]),
    "viai_conv2d_wgrad_ws_bytes": (C.c_size_t, [_CP]),
    "viai_conv2d_wgrad": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _P]),
    "viai_conv2d_last_kernel": (_I, [C.c_char_p, _I]),
    "viai_step_scalars": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P]),
    "viai_pack_weight": (_I, [_P, _P, _I, _I, _I, _L, _L, _P]),
    "viai_bn_finalize": (_I, [_P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "viai_bn_finalize_tiles": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "viai_bn_finalize_lin": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "viai_conv2d_stat_tiles": (_I, [_CP, _IP, _IP]),
    "viai_bn_eval_coeffs": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P]),
    "viai_bn_act_fwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "viai_bn_bwd_blocks": (_I, [_L, _I]),
    "viai_bn_act_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P]),
    "viai_bn_act_bwd_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P, _P]),
    "viai_conv2d_dgrad_f16_ok": (_I, [_CP]),
    "viai_conv2d_wgrad_f16_ok": (_I, [_CP]),
    "viai_conv2d_wgrad_f16": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "viai_conv2d_pack_dgrad_f16": (_I, [_CP, _P, _P, _P]),
    "viai_conv2d_dgrad_f16": (_I, [_CP, _P, _P, _P, _P, _P, _P]),
    "viai_act_bwd_from_output": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "viai_add_act_bwd_from_output": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "viai_bilinear_ac_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "viai_bilinear_ac_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "viai_bn_act_bilinear_fwd_amax": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "viai_avgpool_h_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "viai_avgpool_h_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "viai_nchw_to_nhwc4": (_I, [_P, _P, _L, _I, _L, _P]),
    "viai_nchw_to_nhwc4_amax": (_I, [_P, _P, _L, _I, _L, _P, _P]),
    "viai_maxpool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "viai_maxpool_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "viai_avgpool_hw_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "viai_avgpool_hw_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "viai_avgpool2d_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "viai_avgpool2d_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "viai_add_relu_fwd": (_I, [_P, _P, _P, _L, _P]),
    "viai_bn_add_act_fwd_amax": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _P]),
    "viai_bn_act_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "viai_bn_act_pool_bwd_amax": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P, _P]),
    "viai_bn_act_pool_bwd_amax2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P, _P]),
    "viai_relu_bwd": (_I, [_P, _P, _P, _L, _P]),
    "viai_reduce_blocks": (_I, [_L]),
    "viai_bce_fwd": (_I, [_P, _F, _L, _P, _P, _P]),
    "viai_bce_bwd": (_I, [_P, _F, _L, _P, _P, _P]),
    "viai_mse_fwd": (_I, [_P, _F, _L, _P, _P, _P]),
    "viai_mse_bwd": (_I, [_P, _F, _L, _P, _P, _P]),
    "viai_l1_fwd": (_I, [_P, _P, _L, _P, _P, _P]),
    "viai_l1_bwd": (_I, [_P, _P, _L, _P, _P, _P]),
    "viai_l2c_fwd": (_I, [_P, _P, _I, _I, _F, _I, _P, _P, _P]),
    "viai_l2c_bwd": (_I, [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P]),
    "viai_weight_norm_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "viai_weight_norm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "viai_glu_fwd": (_I, [_P, _P, _P, _L, _I, _P]),
    "viai_glu_bwd": (_I, [_P, _P, _P, _P, _L, _I, _P]),
    "viai_add_scale": (_I, [_P, _P, _P, _F, _L, _P]),
    "viai_relu_fwd": (_I, [_P, _P, _L, _P]),
    "viai_outer_fwd": (_I, [_P, _P, _P, _P, _L, _I, _P]),
    "viai_outer_bwd_blocks": (_I, [_L]),
    "viai_outer_bwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "viai_upsample_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "viai_upsample_bwd_blocks": (_I, []),
    "viai_upsample_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "viai_mol_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _F, _P]),
    "viai_scale_by_scalar": (_I, [_P, _P, _L, _P]),
    "viai_mol_sample": (_I, [_P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "viai_wavenet_synth_step": (_I, [C.POINTER(WnSynth), _P]),
    "viai_wavenet_synth_run": (_I, [C.POINTER(WnSynth), _I, _I, _P]),
    "viai_wn_pipe_ok": (_I, [C.POINTER(WnSynth)]),
    "viai_wn_pipe_profile": (_I, [_P, _I]),
    "viai_wn_pipe_image_floats": (C.c_long, [_I]),
    "viai_wn_pipe_token_granules": (C.c_long, [_I, C.POINTER(C.c_int)]),
    "viai_wn_pipe_run": (_I, [C.POINTER(WnSynth), _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "viai_mask_mul": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "viai_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _D, _D, _D, _F, _P]),
    "viai_colsum_blocks": (_I, [_L, _I]),
    "viai_colsum": (_I, [_P, _L, _I, _P, _P, _I, _P]),
    "viai_axpy": (_I, [_F, _P, _P, _L, _P]),
    "viai_range_count": (_I, [_P, _L, _F, _P, _P]),
    "viai_stft_mel": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "viai_stft_mel_banded": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "viai_conv2d_pack_job": (_I, [_CP, _I, _P, _P, C.POINTER(PackJob)]),
    "viai_pack_jobs_run": (_I, [_P, _I, _I, _P]),
    "viai_frames_prep": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    "viai_slice_clips": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _P]),
    "viai_ema_update": (_I, [_P, _P, _L, _D, _P]),
    "viai_mel_denorm_amp": (_I, [_P, _P, _L, _F, _P]),
    "viai_l2_ranks": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "viai_conv2d_cin1_bn_ok": (_I, [_CP]),
    "viai_pair_cout1_ok": (_I, [_CP]),
    "viai_pair_cout1_bn_bwd_blocks": (_I, [_CP]),
    "viai_pair_cout1_fwd": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    "viai_pair_cout1_fwd_dots": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "viai_pair_cout1_wgrad": (_I, [_CP, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    "viai_pair_cout1_bn_bwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "viai_conv2d_cin1_bn_fwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "viai_conv2d_cin1_bn_bwd": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "viai_conv2d_cin1_bn_wgrad": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "viai_conv2d_cin1_bn_dgrad": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "viai_conv2d_p16_ok": (_I, [_CP]),
    "viai_bn_act_fwd_p16": (_I, [_P, _P, _P, _P, _P, _L, _P, _L, _I, _I, _F, _P, _P]),
    "viai_bn_act_maxpool_fwd_twin": (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "viai_bn_add_act_fwd_twin": (_I, [_P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _I, _I, _F, _P, _P, _P]),
    "viai_bn_act_bilinear_fwd_p16": (_I, [_P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "viai_conv2d_cin1_bn_fwd_p16": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _P, _P]),
    "viai_pair_cout1_bn_bwd_p16": (_I, [_CP, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "viai_bn_act_bwd_p16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P, _P]),
    "viai_bn_join_bwd_p16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "viai_bn_act_bwd_p16_twin": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P, _P]),
    "viai_p16_decode": (_I, [_P, _P, _L, _I, _P, _P]),
    "viai_conv2d_fwd_p16": (_I, [_CP, _P, _P, _P, _P, _P, _I, _P, _P]),
    "viai_conv2d_dgrad_f16_p16": (_I, [_CP, _P, _P, _P, _P, _P, _P]),
    "viai_conv2d_wgrad_f16_p16": (_I, [_CP, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "viai_plan_log_begin": (_I, []),
    "viai_plan_log_end": (_I, []),
    "viai_plan_build": (_I, [_P, _P, C.POINTER(C.c_void_p)]),
    "viai_plan_replay": (_I, [_P, _P]),
    "viai_plan_info": (_I, [_P, C.POINTER(C.c_int), _I]),
    "viai_plan_destroy": (None, [_P]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libviai_hip.so (in-tree)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise ViaiLibraryError("hipcc build of libviai_hip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


def load() -> C.CDLL:
    """dlopen the library and type every entry point.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ViaiLibraryError(
            "libviai_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ViaiLibraryError("libviai_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    hdr = _header_abi_version()
    if lib.viai_abi_version() != ABI_VERSION or (hdr is not None and hdr != ABI_VERSION):
        raise ViaiLibraryError("ABI mismatch: libviai_hip.so is version %d, viai_amd/_lib.py binds version %d, include/viai_hip.h declares %s: "
                               "rebuild (viai_amd._lib.build()) / update the bindings" % (lib.viai_abi_version(), ABI_VERSION, hdr))
    _lib = lib
    return lib


def check(err: int, what: str) -> None:
    if err != 0:
        raise ViaiLibraryError("%s failed with hipError_t %d" % (what, err))
