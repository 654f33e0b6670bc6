"""ctypes binding of libganinpaint.so (include/ganinpaint.h). Torch tensors are containers only:
device memory + streams; every pointer handed over is `tensor.data_ptr()`."""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GI_LIB_PATH: another build of the same library (tools/: A/B runs against an ablation build); there is no other fallback
LIB_PATH = os.environ.get("GI_LIB_PATH") or os.path.join(_HERE, "libganinpaint.so")

GI_F32, GI_F16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes). This table is also what tests/test_abi.py checks against the header.
PROTOTYPES = {
    "gi_last_error": (C.c_char_p, []),
    "gi_version": (_i, []),
    "gi_set_option": (_i, [C.c_char_p, _i]),
    "gi_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "gi_debug_last_kernel": (C.c_char_p, []),
    "gi_debug_fold_count": (_i, []),
    "gi_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "gi_ctx_destroy": (_i, [_vp]),
    "gi_ctx_sync": (_i, [_vp]),
    "gi_unet_create": (_i, [_vp, _i, _i, _f, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gi_unet_create_ex": (_i, [_vp, _i, _i, _i, _f, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gi_unet_create_norm": (_i, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gi_unet_create_padded": (_i, [_vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gi_patchgan_create": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "gi_net_destroy": (_i, [_vp]),
    "gi_net_tensor_count": (_i, [_vp]),
    "gi_net_tensor_desc": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i),
                                C.POINTER(_i64), C.POINTER(_i64)]),
    "gi_net_param_floats": (_i64, [_vp]),
    "gi_net_buffer_floats": (_i64, [_vp]),
    "gi_net_workspace_bytes": (_i64, [_vp]),
    "gi_net_bind": (_i, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "gi_net_sync_weights": (_i, [_vp]),
    "gi_net_set_train": (_i, [_vp, _i]),
    "gi_net_set_inference": (_i, [_vp, _i]),
    "gi_net_set_loss_scale": (_i, [_vp, _f]),
    "gi_net_set_bn_groups": (_i, [_vp, _i]),
    "gi_net_set_dropout_seed": (_i, [_vp, _u64]),
    "gi_net_dropout_mask": (_i, [_vp, _i, _i, _vp, _i64]),
    "gi_net_saved_activation": (_i, [_vp, _i, _i, _i, _vp, _i64]),
    "gi_net_debug_nonzero_tickets": (_i, [_vp, _vp]),
    "gi_net_set_dropout_mask": (_i, [_vp, _i, _i, _vp]),
    "gi_net_forward": (_i, [_vp, _i, _vp, _vp, _i]),
    "gi_net_backward": (_i, [_vp, _i, _vp, _vp, _i]),
    "gi_net_backward_phase": (_i, [_vp, _i, _vp, _vp, _i, _i]),
    "gi_net_phase_split": (_i64, [_vp]),
    "gi_net_phase_split2": (_i64, [_vp]),
    "gi_comm_unique_id": (_i, [C.c_char_p]),
    "gi_comm_create": (_i, [C.c_char_p, _i, _i, _i, C.POINTER(_vp)]),
    "gi_comm_destroy": (_i, [_vp]),
    "gi_allreduce_sum_f32": (_i, [_vp, _vp, _i64, _vp]),
    "gi_net_allreduce_grads_async": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "gi_allreduce_wait": (_i, [_vp, _vp, _vp]),
    "gi_patchgan_gradient_penalty": (_i, [_vp, _vp, _i, _f, _vp]),
    "gi_interpolate": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "gi_mask_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i]),
    "gi_mask_composite": (_i, [_vp, _vp, _vp, _vp, _vp, _i64]),
    "gi_mul": (_i, [_vp, _vp, _vp, _vp, _i64]),
    "gi_add": (_i, [_vp, _vp, _vp, _vp, _i64, _f]),
    "gi_loss_l1": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _f, _vp]),
    "gi_loss_rmse": (_i, [_vp, _vp, _vp, _i64, _f, _vp, _vp, _f, _vp]),
    "gi_loss_mse": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _f, _vp]),
    "gi_loss_local": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _f, _vp]),
    "gi_loss_adv": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _f]),
    "gi_loss_adv_pair": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _f, _f]),
    "gi_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f]),
    "gi_rmsprop_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f]),
    "gi_clamp": (_i, [_vp, _vp, _i64, _f, _f]),
    "gi_grad_absmean": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "gi_ssim_scratch_floats": (_i64, [_i, _i, _i, _i, _i]),
    "gi_ssim": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gi_eval_recon_scratch_floats": (_i64, [_i64]),
    "gi_eval_recon": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _vp, _vp, _vp, _vp]),
    "gi_seg_metrics": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp, _i, _vp, _vp, _vp]),
    "gi_loss_tv": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _f, _vp]),
    "gi_loss_cross_entropy": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp, _i, _vp, _vp, _f, _vp]),
    "gi_vgg19_create": (_i, [_vp, _i, _i, _i, _vp]),
    "gi_vgg19_destroy": (None, [_vp]),
    "gi_vgg19_param_floats": (_i64, [_vp]),
    "gi_vgg19_workspace_bytes": (_i64, [_vp]),
    "gi_vgg19_num_tensors": (_i, [_vp]),
    "gi_vgg19_tensor_desc": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "gi_vgg19_bind": (_i, [_vp, _vp, _vp, _i64]),
    "gi_vgg19_sync_weights": (_i, [_vp]),
    "gi_vgg19_perceptual_style": (_i, [_vp, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "gi_vgg19_features": (_i, [_vp, _vp, _i, _i, _vp]),
    "gi_resize_output_size": (_i, [_i, _i, _i, _vp, _vp]),
    "gi_resize_table_bytes": (_i64, [_i, _i, _i, _i]),
    "gi_resize_build_tables": (_i, [_vp, _i, _i, _i, _i, _vp]),
    "gi_resize_to_tensor": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gi_check_finite": (_i, [_vp, _vp, _i64, _vp]),
    "gi_check_finite_scan": (_i, [_vp, _vp, _i64, _vp]),
    "gi_check_finite_finish": (_i, [_vp, _vp]),
    "gi_adam_step_guarded": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp]),
    "gi_adam_step_guarded2": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _i, _f, _vp]),
    "gi_rmsprop_step_guarded": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp]),
    "gi_check_finite_scan_word": (_i, [_vp, _vp, _i64, _vp, _i]),
    "gi_adam_step_scan": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _i, _f, _vp, _i, _i]),
    "gi_rmsprop_step_scan": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp, _i, _i]),
    "gi_wgrad_s2_scratch_bytes": (_i64, [_i, _i, _i, _i, _i, _i]),
    "gi_wgrad_s2_ws": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i64]),
    "gi_conv_s2_forward": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64]),
    "gi_convT_s2_forward": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64]),
    "gi_wgrad_s2": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "gi_conv_s2_forward_ex": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "gi_convT_s2_forward_ex": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "gi_stat_acc_words": (_i64, [_i]),
    "gi_stat_acc_read": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "gi_pack_weights": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "gi_convert": (_i, [_vp, _i, _vp, _vp, _i64]),
    "gi_convert_back": (_i, [_vp, _i, _vp, _vp, _i64]),
    "gi_time_convT_s2": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_f)]),
    "gi_time_conv_s2": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_f)]),
}

_lib = None
_lock = threading.Lock()


class IgemmEx(C.Structure):
    """gi_igemm_ex (include/ganinpaint.h): the fused epilogues of the single-layer _ex entry points."""
    _fields_ = [("relu_cend", _i),
                ("mask", _vp), ("ldmask", _i), ("mask_slope", _f),
                ("add", _vp), ("ldadd", _i),
                ("mask_bits", _vp),
                ("stat_acc", _vp), ("stat_reps", _i), ("stat_pg", _i),
                ("partials", _vp),
                ("bwd_x", _vp), ("bwd_ldx", _i),
                ("bwd_scale", _vp), ("bwd_shift", _vp), ("bwd_mean", _vp), ("bwd_inv", _vp), ("bwd_stride", _i),
                ("bwd_slope", _f), ("bwd_acc", _vp), ("bwd_reps", _i), ("bwd_pg", _i64),
                ("bwd_c0", _i), ("bwd_c", _i),
                ("mask_applied", _i), ("bwd_applied", _i), ("stat_used", _i), ("ntiles_out", _i)]


class BackendError(RuntimeError):
    pass


def lib():
    """The loaded library. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise BackendError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(gan-inpainting_amd/csrc/build.sh). There is no CPU fallback.")
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in PROTOTYPES.items():
                    if os.environ.get("GI_LIB_PATH") and not hasattr(handle, name):
                        continue    # an OLDER build named for an A/B run (tools/r4_ab.sh) may lack newer entries: calling one raises
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(status):
    if status != 0:
        raise BackendError(f"libganinpaint error {status}: {lib().gi_last_error().decode(errors='replace')}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_ctx_cache = {}


def get_ctx(device=None):
    """One gi_ctx per (device, current torch stream)."""
    if not torch.cuda.is_available():
        raise BackendError("no MI355X visible (torch.cuda.is_available() is False); the HIP backend has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    key = (idx, stream)
    if key not in _ctx_cache:
        h = _vp()
        check(lib().gi_ctx_create(idx, _vp(stream), C.byref(h)))
        _ctx_cache[key] = h
    return _ctx_cache[key]


def dtype_code(dtype):
    if dtype in (GI_F16, "fp16", "f16", torch.float16):
        return GI_F16
    if dtype in (GI_F32, "fp32", "f32", torch.float32):
        return GI_F32
    raise ValueError(f"unsupported compute dtype {dtype!r} (fp16 or fp32)")


def torch_dtype(code):
    return torch.float16 if code == GI_F16 else torch.float32


def set_option(name, value):
    """Process-wide kernel-path option (include/ganinpaint.h lists them); value < 0 restores the default."""
    check(lib().gi_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int()
    check(lib().gi_get_option(name.encode(), C.byref(v)))
    return v.value


def last_kernel():
    """Name of the GEMM / weight-gradient kernel launched most recently (gi_debug_last_kernel)."""
    return (lib().gi_debug_last_kernel() or b"").decode()
