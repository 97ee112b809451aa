"""ctypes binding of libvxm_hip.so (include/vxm_hip.h) — the only way the host reaches the GPU.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  `import torch` happens before `ctypes.CDLL` so that the library's libamdhip64
SONAME resolves to the runtime torch already loaded (one HIP runtime per process: streams,
device pointers and the caching allocator are shared — SURVEY.md Appendix C).
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvxm_hip.so")

_c = ctypes
_P, _I, _L, _F, _S = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t


class Bf16PackJob(_c.Structure):
    """`VxmBf16PackJob` of include/vxm_hip.h"""
    _fields_ = [("w", _c.c_void_p), ("wpacked", _c.c_void_p), ("Cw_in", _c.c_int), ("Cw_out", _c.c_int), ("ci_lo", _c.c_int),
                ("ci_n", _c.c_int), ("transpose_flip", _c.c_int)]


class S3PackJob(_c.Structure):
    """`VxmS3PackJob` of include/vxm_hip.h"""
    _fields_ = [("w", _c.c_void_p), ("wpacked", _c.c_void_p), ("Cw_in", _c.c_int), ("Cw_out", _c.c_int), ("ci_lo", _c.c_int),
                ("ci_n", _c.c_int), ("transpose_flip", _c.c_int), ("seg0", _c.c_int), ("pieces", _c.c_int)]


# name -> argtypes (return type is int unless listed in _RESTYPES); mirrors include/vxm_hip.h
SIGNATURES = {
    "vxm_version": [],
    "vxm_last_error_string": [],
    "vxm_workspace_bytes": [_I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_bwd_data": [_P, _I, _L, _P, _I, _I, _I, _P, _P, _L, _P, _L, _F, _I, _I, _I, _I, _P],
    "vxm_warp3d_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vxm_warp3d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vxm_warp3d_up_ok": [_I, _I, _I, _I, _I, _I],
    "vxm_warp3d_up_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "vxm_warp3d_up_bwd": [_P, _P, _P, _P, _P, _S, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "vxm_vecint_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_vecint_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_vecint_bwd_ws": [_P, _P, _P, _P, _P, _S, _I, _I, _I, _I, _I, _P],
    "vxm_resize3d_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_resize3d_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_conv3d_k3_packed_elems": [_I, _I],
    "vxm_conv3d_k3_pack_weights": [_P, _P, _I, _I, _I, _P],
    "vxm_conv3d_k3_pack_weights_range": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fwd": [_P, _I, _L, _I, _P, _I, _L, _P, _P, _P, _L, _I, _F, _P, _L, _F, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_up_ok": [_P, _I, _L, _P, _I, _L, _P, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_up_packed_elems": [_I, _I, _I],
    "vxm_conv3d_k3_up_pack_weights": [_P, _P, _I, _I, _I, _P],
    "vxm_conv3d_k3_up_fwd": [_P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _I, _F, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_up_bwd_low_ok": [_P, _L, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_up_bwd_low_packed_elems": [_I, _I],
    "vxm_conv3d_k3_up_bwd_low": [_P, _L, _I, _P, _I, _I, _P, _P, _L, _P, _L, _F, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fewout_ok": [_P, _L, _P, _L, _I, _I, _I],
    "vxm_conv3d_k3_fewout_fwd": [_P, _I, _L, _P, _P, _P, _L, _I, _F, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fewout_fwd_layout": [_P, _I, _L, _P, _P, _P, _L, _I, _F, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fwd_layout_ok": [_P, _I, _L, _P, _I, _L, _P, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_fwd_layout": [_P, _I, _L, _I, _P, _I, _L, _P, _P, _P, _L, _I, _F, _P, _L, _F, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fwd_variant": [_P, _I, _L, _P, _I, _L, _P, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_bwd_weight_variant": [_P, _I, _L, _I, _P, _I, _L, _P, _L, _I, _I, _I, _I],
    "vxm_conv3d_k3_bwd_weight_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_bwd_weight": [_P, _I, _L, _I, _P, _I, _L, _P, _L, _I, _P, _P, _P, _S, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_bwd_weight_up_segment": [_P, _I, _L, _P, _I, _L, _P, _L, _I, _P, _P, _S, _I, _I, _I, _I, _P],
    "vxm_lrelu_bwd": [_P, _L, _P, _L, _P, _L, _F, _I, _I, _L, _P],
    "vxm_maxpool2_fwd": [_P, _L, _P, _I, _I, _I, _I, _I, _P],
    "vxm_maxpool2_fwd_code": [_P, _L, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_maxpool3d_k_fwd": [_P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vxm_maxpool3d_k_bwd": [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vxm_upsample3d_k_cat": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vxm_upsample3d_k_bwd": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vxm_maxpool2_bwd": [_P, _L, _P, _P, _L, _P, _F, _I, _I, _I, _I, _I, _P],
    "vxm_upsample2_bwd": [_P, _L, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "vxm_upsample2_cat": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "vxm_ncc_fused": [_I, _I],
    "vxm_ncc_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_ncc_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_gradloss_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_gradloss_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_mse_fwd": [_P, _P, _P, _P, _L, _P],
    "vxm_mse_bwd": [_P, _P, _P, _P, _P, _L, _P],
    "vxm_dice_fwd": [_P, _P, _P, _P, _I, _I, _L, _P],
    "vxm_dice_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _L, _P],
    "vxm_warp2d_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_warp2d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_vecint2d_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "vxm_vecint2d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vxm_resize2d_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_resize2d_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vxm_maxpool2d_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "vxm_maxpool2d_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vxm_upsample2d_cat": [_P, _I, _P, _I, _P, _I, _I, _I, _P],
    "vxm_upsample2d_bwd": [_P, _I, _P, _I, _I, _I, _I, _P],
    "vxm_ncc2d_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vxm_ncc2d_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vxm_ncc1d_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vxm_ncc1d_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vxm_gradloss2d_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "vxm_gradloss2d_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "vxm_bf16_to_blocked": [_P, _I, _L, _P, _I, _L, _P, _I, _I, _L, _P],
    "vxm_bf16_from_blocked": [_P, _I, _P, _I, _I, _L, _P],
    "vxm_bf16_conv_packed_bytes": [_I, _I],
    "vxm_bf16_conv_pack_weights": [_P, _I, _I, _I, _I, _I, _P, _P],
    "vxm_bf16_conv_pack_weights_batch": [_P, _I, _P],
    "vxm_bf16_conv_fwd": [_P, _I, _I, _P, _I, _P, _P, _P, _I, _I, _F, _P, _F, _I, _I, _I, _I, _P],
    "vxm_bf16_conv_bwd_data_up": [_P, _I, _P, _P, _I, _P, _F, _I, _I, _I, _I, _P],
    "vxm_bf16_conv_bwd_weight_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "vxm_bf16_conv_bwd_weight": [_P, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _S, _I, _I, _I, _I, _P],
    "vxm_bf16_maxpool2_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vxm_bf16_maxpool2_bwd": [_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "vxm_bf16_upsample2_bwd": [_P, _P, _P, _F, _I, _I, _I, _I, _I, _P],
    "vxm_bf16_lrelu_bwd": [_P, _P, _P, _F, _L, _P],
    "vxm_conv3d_k3_s3_ok": [_I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_variant": [_I],
    "vxm_conv3d_k3_s3_tile_rows": [_I, _I, _I],
    "vxm_conv3d_k3_s3_tile_rows_at": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_producer_consumer": [_I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_layout_ok": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_packed_bytes": [_I, _I, _I, _I],
    "vxm_conv3d_k3_s3_pack_weights_batch": [_P, _I, _P],
    "vxm_conv3d_k3_s3_fwd": [_P, _I, _L, _I, _P, _I, _L, _P, _P, _P, _L, _I, _F, _P, _L, _F, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3_bwd_weight_ok": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_bwd_weight_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3_bwd_weight_kernel": [_I, _I, _I],
    "vxm_conv3d_k3_s3_bwd_weight": [_P, _I, _L, _P, _L, _I, _P, _I, _I, _P, _P, _S, _I, _I, _I, _I, _I, _P],
    "vxm_ncc_win_elems": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vxm_ncc_win_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vxm_ncc_win_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_ok": [_I, _I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_packed_bytes": [_I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_pack_weights": [_P, _P, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_fwd": [_P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _I, _F, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_fwd_signs": [_P, _I, _L, _P, _I, _L, _P, _P, _P, _L, _I, _F, _I, _I, _I, _I, _I, _P, _L, _P],
    "vxm_conv3d_k3_s3u_fwd_kernel": [_L, _L, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_data_ok": [_I, _I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_skip_packed_bytes": [_I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_skip_pack_weights": [_P, _P, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_bwd_data": [_P, _L, _I, _P, _P, _L, _I, _P, _L, _F, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_bwd_low_ok": [_I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_low_packed_bytes": [_I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_low_pack_weights": [_P, _P, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_bwd_low": [_P, _L, _I, _P, _P, _L, _I, _P, _L, _F, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_s3u_bwd_weight_ok": [_I, _I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "vxm_conv3d_k3_s3u_bwd_weight": [_P, _I, _L, _P, _L, _I, _P, _I, _P, _S, _I, _I, _I, _I, _I, _P],
    "vxm_s3_range_probe": [_P, _I, _L, _I, _I, _I, _I, _I, _P, _P],
    "vxm_conv3d_k3_fewch_bwd_weight_ok": [_P, _I, _L, _P, _I, _L, _P, _L, _I, _I, _I],
    "vxm_conv3d_k3_fewch_bwd_weight": [_P, _I, _L, _P, _I, _L, _P, _L, _I, _P, _P, _P, _S, _I, _I, _I, _I, _I, _P],
    "vxm_conv3d_k3_fewch_bwd_weight_pool": [_P, _I, _L, _P, _I, _L, _P, _L, _P, _P, _F, _P, _P, _P, _S, _I, _I, _I, _I, _I, _P],
    "vxm_loss_combine_fwd": [_P, _P, _I, _P, _P, _P],
    "vxm_loss_combine_bwd": [_P, _P, _I, _P, _P],
    "vxm_fill_zero": [_P, _S, _P],
    "vxm_add2": [_P, _P, _P, _L, _P],
    "vxm_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P],
    "vxm_adam_step_dev": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _F, _P],
}
_RESTYPES = {
    "vxm_workspace_bytes": _S,
    "vxm_last_error_string": _c.c_char_p,
    "vxm_conv3d_k3_packed_elems": _S,
    "vxm_conv3d_k3_up_packed_elems": _S,
    "vxm_conv3d_k3_up_bwd_low_packed_elems": _S,
    "vxm_conv3d_k3_bwd_weight_workspace_bytes": _S,
    "vxm_bf16_conv_packed_bytes": _S,
    "vxm_conv3d_k3_s3_packed_bytes": _S,
    "vxm_conv3d_k3_s3_bwd_weight_workspace_bytes": _S,
    "vxm_bf16_conv_bwd_weight_workspace_bytes": _S,
    "vxm_ncc_win_elems": _L,
    "vxm_conv3d_k3_s3u_packed_bytes": _S,
    "vxm_conv3d_k3_s3u_bwd_low_packed_bytes": _S,
    "vxm_conv3d_k3_s3u_bwd_skip_packed_bytes": _S,
    "vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes": _S,
}

_lib = None


class VxmHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VxmHipError(
                "libvxm_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or voxelmorph_amd/csrc/build.sh. There is no CPU / ATen fallback for this path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, _I)
        _lib = handle
    return _lib


def call(name, *args):
    """Invoke an int-status entry point; raise VxmHipError(vxm_last_error_string()) on failure."""
    h = lib()
    status = getattr(h, name)(*args)
    if status != 0:
        raise VxmHipError("%s failed (status %d): %s" % (name, status, h.vxm_last_error_string().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise VxmHipError("voxelmorph_amd runs on an MI355X (HIP) device only; got a %s tensor. "
                              "Move the model/inputs with .to('cuda'); there is no CPU fallback." % t.device)
        if t.dtype != torch.float32:
            raise VxmHipError("voxelmorph_amd kernels compute in fp32; got %s" % t.dtype)
