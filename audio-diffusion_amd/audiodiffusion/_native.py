"""ctypes binding of the C-ABI in include/adm.h (libadm_hip.so, built for gfx950 by csrc/build.sh).

This is plumbing only: torch supplies device memory and streams, every arithmetic op on the hot path is a
hand-written HIP kernel inside the library. There is NO CPU fallback: if the library is missing the import
of any product class fails loudly. (`load(path)` with an explicit path exists so the test-suite can point the
same binding at the CPU-emulation build of the very same kernel sources, tests/emu/libadm_emu.so.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libadm_hip.so")

c_float_p = C.POINTER(C.c_float)


class SchedCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("sqrt_beta", "sqrt_alpha", "clip", "k_x0", "k_x", "k_eps", "k_noise", "timestep")]


class ConvArgs(C.Structure):
    _fields_ = [
        ("x1", C.c_void_p), ("C1", C.c_int),
        ("x2", C.c_void_p), ("C2", C.c_int),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("up", C.c_int), ("stride", C.c_int), ("ks", C.c_int), ("pad_lo", C.c_int),
        ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("act", C.c_int),
        ("wpacked", C.c_void_p), ("bias", C.c_void_p), ("Cout", C.c_int),
        ("chan_add", C.c_void_p), ("chan_add_stride", C.c_int),
        ("residual", C.c_void_p),
        ("out", C.c_void_p),
        ("x1_bstride", C.c_long), ("x2_bstride", C.c_long), ("w_bstride", C.c_long),
        ("wino_packed", C.c_void_p),
        ("bf16_packed", C.c_void_p),
        ("stats_out", C.c_void_p), ("stats_tiles", C.c_int),
        ("wino6_rule", C.c_int),
        ("single_sample", C.c_int),
    ]


class OpProfile(C.Structure):
    _fields_ = [("kind", C.c_int), ("variant", C.c_int), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


class UNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("layers_per_block", C.c_int), ("n_blocks", C.c_int),
        ("block_out_channels", C.c_int * 8), ("down_attn", C.c_int * 8), ("up_attn", C.c_int * 8),
        ("attention_head_dim", C.c_int), ("norm_num_groups", C.c_int), ("norm_eps", C.c_float),
        ("flip_sin_to_cos", C.c_int), ("freq_shift", C.c_float), ("sample_h", C.c_int), ("sample_w", C.c_int),
        ("cross_attention_dim", C.c_int),
    ]


_SIGS = {
    "adm_version": (C.c_int, []),
    "adm_last_error": (C.c_char_p, []),
    "adm_is_device_build": (C.c_int, []),
    "adm_last_conv_variant": (C.c_int, []),
    "adm_release_stream": (C.c_int, [C.c_void_p]),
    "adm_has_experiments": (C.c_int, []),
    "adm_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "adm_sched_step": (C.c_int, [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "adm_add_noise": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                C.c_int, C.c_int, C.c_long, C.c_void_p]),
    "adm_dequant_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "adm_slerp_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_conv2d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "adm_conv_stats_tiles": (C.c_int, [C.POINTER(ConvArgs)]),
    "adm_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adm_pack_conv_weight_T": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adm_winograd_packed_floats": (C.c_long, [C.c_int, C.c_int, C.c_int]),
    "adm_pack_winograd_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "adm_pack_winograd_weight_T": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "adm_pack_bf16_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adm_pack_bf16_weight_ks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adm_conv_out_dims": (None, [C.c_int] * 6 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "adm_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "adm_layernorm_nct": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_long, C.c_float, C.c_void_p]),
    "adm_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p]),
    "adm_cross_attention": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "adm_attention_blocked": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "adm_attention_mfma_eligible": (C.c_int, [C.c_int] * 3),
    "adm_layernorm_nct_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_long, C.c_float, C.c_void_p]),
    "adm_geglu_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_long, C.c_void_p]),
    "adm_cross_attention_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 6 + [C.c_void_p]),
    "adm_attention_backward_blocked": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "adm_sepconv_block": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "adm_dense_act": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "adm_unet_create": (C.c_int, [C.POINTER(UNetConfig), C.POINTER(C.c_void_p)]),
    "adm_unet_destroy": (None, [C.c_void_p]),
    "adm_unet_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "adm_unet_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "adm_unet_set_encoding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "adm_unet_missing_params": (C.c_int, [C.c_void_p]),
    "adm_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, c_float_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "adm_unet_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "adm_unet_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.POINTER(OpProfile), C.c_int,
                                   C.POINTER(C.c_int), C.c_void_p]),
    "adm_sample_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(SchedCoef), C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "adm_encode_loop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(SchedCoef), C.c_int, C.c_int, C.c_void_p]),
}
# entry points added by later translation units (k_mel.hip); bound when present in the header AND the library
_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
_OPTIONAL_SIGS = {
    "adm_unet_bind_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "adm_unet_enable_training": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long]),
    "adm_unet_refresh_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adm_unet_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_void_p]),
    "adm_unet_set_grad_bucket_hook": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_long), C.c_void_p, C.c_void_p]),
    "adm_groupnorm_stats_ex": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "adm_groupnorm_backward": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "adm_conv_wgrad_workspace": (_l, [C.POINTER(ConvArgs)]),
    "adm_conv2d_wgrad": (_i, [C.POINTER(ConvArgs), _vp, _vp, _i, _vp, _vp]),
    "adm_blocked_image_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "adm_blocked_sums_scratch": (_l, [_i, _i, _i, _i]),
    "adm_blocked_apply": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "adm_conv2d_bf16_blocked_eligible": (_i, [_i, _i, _i, _i]),
    "adm_conv2d_bf16_blocked": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "adm_conv2d_wgrad_bf16_blocked_eligible": (_i, [_i, _i, _i, _i]),
    "adm_conv_wgrad_blocked_workspace": (_l, [_i, _i, _i, _i, _i]),
    "adm_conv2d_wgrad_bf16_blocked": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "adm_sumpool2x2": (_i, [_vp, _vp, _i, _i, _l, _i, _vp]),
    "adm_accumulate": (_i, [_vp, _l, _vp, _l, _l, _i, _i, _vp]),
    "adm_chan_sums": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "adm_attention_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "adm_linear_backward": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "adm_conv_small_cin_wgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "adm_conv_small_cout_backward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "adm_mse_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_grad_norm_clip": (C.c_int, [C.c_void_p, C.c_long, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_grad_norm_clip_scaled": (C.c_int, [C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "adm_unet_set_loss_scale": (C.c_int, [C.c_void_p, C.c_float]),
    "adm_adamw_ema_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long] + [C.c_float] * 5 +
                           [C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "adm_flat_op": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p]),
    "adm_vae_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "adm_vae_destroy": (None, [C.c_void_p]),
    "adm_vae_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "adm_vae_latent_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "adm_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "adm_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "adm_mel_create": (C.c_int, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.POINTER(C.c_void_p)]),
    "adm_mel_destroy": (None, [C.c_void_p]),
    "adm_mel_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "adm_mel_forward_power": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "adm_mel_set_nnls_solver": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "adm_mel_last_nnls": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "adm_mel_inverse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_float), C.c_void_p]),
}

_lib = None
_lib_path = None


BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int)   # include/adm.h: adm_bucket_fn


class NativeError(RuntimeError):
    pass


def load(path=None):
    """Load the native library (idempotent). Raises if it is missing — there is no fallback."""
    global _lib, _lib_path
    path = os.path.abspath(path or DEFAULT_LIB)
    if _lib is not None and _lib_path == path:
        return _lib
    if not os.path.exists(path):
        raise NativeError(
            f"native library {path} not found: build it with `bash audio-diffusion_amd/csrc/build.sh` "
            "(hipcc --offload-arch=gfx950). The hot path has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in list(_SIGS.items()) + list(_OPTIONAL_SIGS.items()):
        if name in _OPTIONAL_SIGS and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib, _lib_path = lib, path
    return lib


def lib():
    return _lib if _lib is not None else load()


def is_device_build():
    return bool(lib().adm_is_device_build())


def default_device():
    """Where product objects allocate: the CURRENT HIP device of this process (one process per GPU: train_unet.py and
    bench.py call torch.cuda.set_device(LOCAL_RANK) before building anything), or the host for the emulation build."""
    if is_device_build():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def check(rc):
    if rc != 0:
        raise NativeError(lib().adm_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or, for the emulation build, host) address of a contiguous fp32/u8/int32 tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "native ops need contiguous tensors"
    if is_device_build():
        assert t.is_cuda, "native ops need tensors on the GPU (cuda == HIP device on ROCm)"
    else:
        assert not t.is_cuda
    return C.c_void_p(t.data_ptr())


def stream_for(t):
    """The HIP stream torch is currently using for t's device (NULL on the emulation build)."""
    if t is not None and t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None
