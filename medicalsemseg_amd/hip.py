"""ctypes binding of ``libmsseg_hip.so`` (C ABI: ``include/msseg.h``) + thin tensor-level wrappers.

There is NO CPU fallback: if the shared library is missing, or a tensor is not on a GPU, these
functions raise.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSSEG_LIB") or os.path.join(_HERE, "libmsseg_hip.so")   # MSSEG_LIB: A/B another build

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}
_LAB = {torch.float32: 0, torch.bfloat16: 1, torch.uint8: 2, torch.int64: 3}

_lib = None

_vp, _ll, _i, _f, _sz = C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_size_t

# name -> argtypes (every symbol include/msseg.h declares; tests check the .so exports all of them)
SIGNATURES = {
    "msseg_abi_version": ([], _i),
    "msseg_last_error": ([], C.c_char_p),
    "msseg_num_cus": ([], _i),
    "msseg_packed_weight_bytes": ([_i, _i, _i, _i, _i], _sz),
    "msseg_pack_weights": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _ll, _i, _i, _vp], _i),
    "msseg_pack_weights_batch": ([_vp, _i, _ll, _ll, _i, _vp], _i),
    "msseg_cout_block": ([_i], _i),
    "msseg_conv3d_k3_cout_block": ([_i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_variant": ([_i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_kernel": ([_i, _i, _i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k3_fwd_accumulate": ([_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp], _i),
    "msseg_dwconv3d_k3_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_dwconv3d_k3_wgrad": ([_vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_interp_trilinear_fwd": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_interp_trilinear_bwd_workspace_bytes": ([_i, _i, _i, _i, _i, _i, _i, _i], _sz),
    "msseg_interp_trilinear_bwd": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_kv_attention_fwd": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "msseg_kv_attention_bwd_workspace_bytes": ([_i, _i, _i, _i, _i], _sz),
    "msseg_kv_attention_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _sz, _i, _vp], _i),
    "msseg_scale_channels": ([_vp, _vp, _vp, _i, _ll, _i, _i, _vp], _i),
    "msseg_reduce_scratch_bytes": ([], _sz),
    "msseg_conv3d_k3_dgrad_inbwd": ([_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp, _ll, _vp, _f, _f, _vp,
                                     _vp, _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k1_dgrad_inbwd": ([_vp, _ll, _vp, _vp, _ll, _i, _ll, _i, _i, _vp, _ll, _vp, _ll, _vp, _f, _f, _vp,
                                     _vp, _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_deconv_k2s2_bwd_data_inbwd": ([_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp, _ll, _vp, _f, _f,
                                          _vp, _vp, _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_deconv_k2s2_bwd_fused": ([_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp, _ll, _vp, _f, _f,
                                     _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k3s2_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_zero_stuff2": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_window_attention_fwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_window_attention_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_window_attention_bwd_ws": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp,
                                       _sz, _vp], _i),
    "msseg_window_attention_fwd2": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_window_attention_bwd2": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp,
                                     _sz, _vp], _i),
    "msseg_window_attention_bwd_workspace_bytes": ([_i, _i, _i, _i, _i, _i, _i, _i, _i], _sz),
    "msseg_layernorm_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _vp, _vp, _ll, _i, _f, _i, _vp], _i),
    "msseg_layernorm_bwd": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _ll, _i, _i, _vp], _i),
    "msseg_layernorm_bwd_add": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _ll, _ll, _i, _i, _vp], _i),
    "msseg_layernorm_param_grad": ([_vp, _ll, _vp, _vp, _vp, _ll, _vp, _vp, _i, _ll, _i, _vp, _sz, _i, _vp], _i),
    "msseg_gelu_fwd": ([_vp, _vp, _ll, _i, _vp], _i),
    "msseg_gelu_bwd": ([_vp, _vp, _vp, _ll, _i, _vp], _i),
    "msseg_linear_wgrad_ok": ([_ll, _i, _i, _i], _i),
    "msseg_linear_wgrad": ([_vp, _ll, _vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_linear_gelu_ok": ([_ll, _i, _i, _i], _i),
    "msseg_linear_gelu_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _vp, _ll, _ll, _i, _i, _i, _vp], _i),
    "msseg_linear_gelu_bwd": ([_vp, _ll, _vp, _vp, _ll, _vp, _ll, _ll, _i, _i, _i, _vp], _i),
    "msseg_linear_add_ok": ([_ll, _i, _i, _i], _i),
    "msseg_linear_add_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _vp, _ll, _ll, _i, _i, _i, _vp], _i),
    "msseg_conv3d_stem_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_stem_k1_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_stem_norm_fwd": ([_vp, _ll, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_conv3d_k1_head_dgrad_inbwd": ([_vp, _ll, _vp, _vp, _ll, _i, _ll, _i, _i, _vp, _ll, _vp, _vp, _vp, _f, _f, _vp, _vp,
                                          _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k1_head_norm_fwd": ([_vp, _ll, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _ll, _i, _ll, _i, _i, _i, _vp], _i),
    "msseg_conv3d_k1_head_bwd_fused": ([_vp, _ll, _vp, _vp, _ll, _i, _ll, _i, _i, _vp, _ll, _vp, _vp, _vp, _f, _f, _vp, _vp,
                                        _vp, _i, _vp, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k1_head_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _ll, _i, _i, _i, _vp], _i),
    "msseg_conv3d_k1_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _ll, _i, _i, _i, _vp], _i),
    "msseg_conv3d_gather_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_deconv_k2s2_fwd": ([_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_deconv_k2s2_bwd_data": ([_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_wgrad_workspace_bytes": ([_i, _i, _i], _sz),
    "msseg_conv3d_k3_wgrad_kernel": ([_i, _i, _i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_wgrad": ([_vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_k1_wgrad": ([_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_conv3d_gather_wgrad": ([_vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_deconv_k2s2_wgrad": ([_vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_channel_stats": ([_vp, _ll, _vp, _i, _ll, _i, _vp, _sz, _i, _vp], _i),
    "msseg_instnorm_act_fwd": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _i, _ll, _i, _f, _f, _i, _vp], _i),
    "msseg_instnorm_act_poolbwd_reduce": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i,
                                           _i, _i, _i, _i, _i, _f, _f, _vp, _sz, _i, _vp], _i),
    "msseg_instnorm_act_pool_fwd": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _f, _f, _i, _vp], _i),
    "msseg_instnorm_act_bwd_reduce": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _i, _i, _ll, _i, _f, _f, _vp, _sz, _i, _vp], _i),
    "msseg_instnorm_act_bwd_apply": ([_vp, _ll, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _ll, _vp, _ll, _i, _ll, _i, _f, _f, _i, _vp], _i),
    "msseg_maxpool2_fwd": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_maxpool2_bwd": ([_vp, _ll, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_ncdhw_to_ndhwc": ([_vp, _i, _vp, _ll, _i, _i, _i, _ll, _vp], _i),
    "msseg_ndhwc_to_ncdhw": ([_vp, _ll, _i, _vp, _i, _i, _i, _ll, _vp], _i),
    "msseg_channel_sum": ([_vp, _ll, _vp, _ll, _i, _i, _vp, _sz, _i, _vp], _i),
    "msseg_axpy_rows": ([_vp, _vp, _vp, _vp, _i, _ll, _i, _vp], _i),
    "msseg_add": ([_vp, _ll, _vp, _ll, _vp, _ll, _ll, _i, _i, _vp], _i),
    "msseg_dice_ce_partials": ([_vp, _ll, _i, _vp, _i, _vp, _vp, _i, _ll, _i, _vp], _i),
    "msseg_dice_ce_fwd": ([_vp, _ll, _i, _vp, _i, _vp, _vp, _vp, _i, _ll, _i, _f, _f, _vp, _sz, _vp], _i),
    "msseg_dice_ce_finalize": ([_vp, _vp, _i, _ll, _i, _f, _f, _vp], _i),
    "msseg_dice_ce_bwd": ([_vp, _ll, _i, _vp, _i, _vp, _vp, _vp, _ll, _i, _ll, _i, _f, _f, _vp], _i),
    "msseg_adamw_step": ([_vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _vp, _vp, _vp], _i),
    "msseg_sumsq": ([_vp, _ll, _vp, _vp, _i, _vp], _i),
    "msseg_sw_blend": ([_vp, _ll, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_sw_normalize": ([_vp, _vp, _i, _ll, _vp], _i),
    "msseg_sw_gather": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp], _i),
    "msseg_box_copy": ([_vp, _ll, _i, _i, _i, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_merge_gather_fwd": ([_vp, _ll, _i, _i, _i, _vp, _ll, _i, _i, C.c_uint, _i, _vp], _i),
    "msseg_merge_gather_bwd": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, C.c_uint, _i, _vp], _i),
    "msseg_argmax_u8": ([_vp, _i, _ll, _vp, _vp], _i),
    "msseg_resample_nearest_u8": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _vp], _i),
    "msseg_majority_vote_u8": ([_vp, _i, _ll, _i, _vp, _vp], _i),
    "msseg_aug_crop_batch": ([_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp], _i),
    "msseg_sw_gather_batch": ([_vp, _ll, _vp, _ll, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp], _i),
    "msseg_sw_blend_batch": ([_vp, _ll, _i, _vp, _vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_deconv_k2s2_bwd_partials_ok": ([_i, _i, _i], _i),
    "msseg_deconv_k2s2_bwd_partials": ([_vp, _ll, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_conv3d_k3_small_ok": ([_i, _i, _i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_small_workspace_bytes": ([_i, _i, _i, _i, _i, _i], _sz),
    "msseg_conv3d_k3_small_stage_groups": ([_i, _i, _i, _i, _i, _i], _i),
    "msseg_conv3d_k3_small_partials": ([_vp, _ll, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_conv3d_k3_small_fwd_finish": ([_vp, _i, _vp, _vp, _vp, _f, _f, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i,
                                          _i, _vp], _i),
    "msseg_conv3d_k3_small_fwd_finish_res": ([_vp, _i, _vp, _vp, _vp, _f, _f, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _i, _i, _i, _i,
                                              _i, _vp], _i),
    "msseg_conv3d_k3_small_bwd_finish": ([_vp, _i, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _f, _f, _vp, _vp, _i, _i, _i, _i, _i,
                                          _i, _vp], _i),
    "msseg_avgpool3d_k3": ([_vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp], _i),
    "msseg_hd_edges": ([_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "msseg_hd_directed_workspace_bytes": ([_i, _i, _i], _sz),
    "msseg_hd_directed_hist": ([_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _i, _vp], _i),
    "msseg_ktimer_enable": ([_i], _i),
    "msseg_ktimer_reset": ([], _i),
    "msseg_ktimer_count": ([], _i),
    "msseg_ktimer_get": ([_i, C.c_char_p, _i, C.POINTER(C.c_float)], _i),
}


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and set argtypes.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C medicalsemseg_amd/csrc`).  There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.msseg_abi_version() != 1:
        raise RuntimeError("libmsseg_hip.so ABI version mismatch")
    _lib = lib
    return lib


def lib():
    real = _lib if _lib is not None else load_library()
    if TIMER.enabled:
        return _TimedLib(real)
    return real


class MssegError(RuntimeError):
    pass


def _ck(rc: int, what: str):
    if rc != 0:
        raise MssegError(f"{what} failed ({rc}): {lib().msseg_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("medicalsemseg_amd kernels run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback")


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def ld(t: torch.Tensor) -> int:
    """voxel stride of a channels-last [..., C] tensor (possibly a channel slice of a wider buffer)."""
    C = t.shape[-1]
    if C > 1 and t.stride(-1) != 1:
        raise ValueError("channels-last tensor must have unit channel stride")
    # voxel stride = stride of the innermost spatial dim that has extent > 1 (size-1 dims carry arbitrary strides)
    l = None
    exp = None
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] > 1:
            if l is None:
                l = t.stride(d)
                exp = l * t.shape[d]
            else:
                if t.stride(d) != exp:
                    raise ValueError(f"tensor is not a dense channels-last volume: shape {tuple(t.shape)} "
                                     f"stride {t.stride()}")
                exp *= t.shape[d]
    if l is None:
        l = C
    if l < C:
        raise ValueError("voxel stride smaller than the channel count")
    return l


def _nbytes(*ts) -> float:
    """algorithmic bytes of a streaming pass: every operand read or written once"""
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------
def cout_block(M: int) -> int:
    return lib().msseg_cout_block(M)


def conv_k3_cout_block(N, D, H, W, cout) -> int:
    return lib().msseg_conv3d_k3_cout_block(N, D, H, W, cout)


class PackJob(C.Structure):
    """mirror of msseg_pack_job (include/msseg.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p),
                ("s_m1", C.c_longlong), ("s_m0", C.c_longlong), ("s_t", C.c_longlong), ("s_k1", C.c_longlong),
                ("s_k0", C.c_longlong), ("total", C.c_longlong),
                ("M", C.c_int), ("M0", C.c_int), ("T", C.c_int), ("K", C.c_int), ("K0", C.c_int), ("flip", C.c_int),
                ("cout_block", C.c_int), ("nkb", C.c_int)]


# the most recent pack_weights call as a (src, dst, PackJob) triple: layers.PackedCache registers it for the
# once-per-step batched refresh
LAST_PACK_JOBS = []     # (src, dst, PackJob) of every image packed since a PackedCache last cleared the list


def pack_weights(src: torch.Tensor, dtype: torch.dtype, M, M0, T, K, K0, s_m1, s_m0, s_t, s_k1, s_k0, flip=False,
                 out: Optional[torch.Tensor] = None, cb: Optional[int] = None) -> torch.Tensor:
    _need_gpu(src)
    assert src.dtype == torch.float32      # any layout the explicit strides describe (a channel slice of a conv weight)
    cb = cb or cout_block(M)
    nbytes = lib().msseg_packed_weight_bytes(M, T, K, cb, _DT[dtype])
    esz = 4 if dtype == torch.float32 else 2
    if out is None:
        out = torch.empty(nbytes // esz, dtype=dtype, device=src.device)
    assert out.numel() * esz == nbytes
    _ck(lib().msseg_pack_weights(_p(src), _p(out), _DT[dtype], M, M0, T, K, K0, s_m1, s_m0, s_t, s_k1, s_k0,
                                 int(flip), cb, _stream()), "pack_weights")
    LAST_PACK_JOBS.append((src, out, PackJob(_p(src), _p(out), s_m1, s_m0, s_t, s_k1, s_k0, nbytes // esz, M, M0, T, K, K0,
                                             int(flip), cb, -(-K // (64 // esz)))))
    return out


def pack_weights_batch(table: torch.Tensor, njobs: int, max_total: int, sum_total: int, dtype: torch.dtype):
    """table: uint8 device tensor holding njobs consecutive PackJob structs; max_total / sum_total: largest / summed job.total"""
    _ck(lib().msseg_pack_weights_batch(_p(table), njobs, max_total, sum_total, _DT[dtype], _stream()), "pack_weights_batch")


def pack_conv_k3(w: torch.Tensor, dtype, dgrad=False, out=None, vol=None, cb=None):
    """w: [Cout, Cin, 3,3,3] fp32 (dense, or an input-channel slice w_full[:, a:b] of a dense weight: forward image only).
    Forward image W[co][tap][ci]; dgrad image W'[ci][26-tap][co].
    vol = (N, D, H, W) of the stride-1 problem the image will be used for (selects the cout block), or cb = the block."""
    co, ci = w.shape[0], w.shape[1]
    if cb is None and vol is not None and out is None and dtype == torch.bfloat16 and (co if dgrad else ci) == 48 \
            and lib().msseg_conv3d_k3_kernel(*vol, 48, ci if dgrad else co, BF16) == 4:
        return pack_conv_k3_c48(w, dgrad)
    if not dgrad:
        cb = cb or (conv_k3_cout_block(*vol, co) if vol is not None else None)
        assert w.stride(1) == 27 and w.stride(4) == 1 and w.stride(0) % 27 == 0
        return pack_weights(w, dtype, co, co, 27, ci, ci, 0, w.stride(0), 1, 0, 27, False, out, cb)
    assert w.is_contiguous()
    cb = cb or (conv_k3_cout_block(*vol, ci) if vol is not None else None)
    return pack_weights(w, dtype, ci, ci, 27, co, co, 0, 27, 1, 0, ci * 27, True, out, cb)


def pack_conv_k3_c48(w: torch.Tensor, dgrad=False):
    """bf16 image of the 48-input-channel ping-pong kernel (csrc/conv3d_k3_c48.hip; msseg_conv3d_k3_kernel() == 4) for the
    layer y = conv(x[48 ch], w) -- w: [Cout, 48, 3,3,3] (dense, or an input-channel slice of a dense weight) -- or, dgrad,
    for dx = conv(dy[48 ch], w') of a layer w: [48, Cin, 3,3,3].  Four msseg_pack_weights images back to back, 16-wide cout
    blocks: channels 0..31 of the 27 taps; channels 32..47 of the tap pairs (kw 0, kw 1) per (kd, kh); of (kd 0, kd 1)
    at kw 2 per kh; of (kd 2, kw 2) per kh (upper half of the k-step zero)."""
    co, ci = w.shape[0], w.shape[1]
    n_el = w.untyped_storage().nbytes() // 4 - w.storage_offset()
    flat = w.as_strided((n_el,), (1,))                            # the storage from w's first element on
    if not dgrad:
        assert ci == 48 and w.stride(1) == 27 and w.stride(4) == 1 and w.stride(0) % 27 == 0
        M, s_m, s_c = co, w.stride(0), 27                         # strides of the output / input channel index
    else:
        assert co == 48 and w.is_contiguous()
        M, s_m, s_c = ci, 27, ci * 27
    assert M % 16 == 0
    ncb = M // 16
    buf = torch.empty(ncb * 42 * 512, dtype=torch.bfloat16, device=w.device)
    o = [0, ncb * 27 * 512, ncb * 36 * 512, ncb * 39 * 512, ncb * 42 * 512]
    rest = 32 * s_c
    bf = torch.bfloat16
    pack_weights(flat, bf, M, M, 27, 32, 32, 0, s_m, 1, 0, s_c, dgrad, buf[o[0]:o[1]], 16)
    if not dgrad:
        # T index = kd * 3 + kh (x 3 taps); k = [16 channels of tap A | 16 channels of tap B]
        pack_weights(flat[rest:], bf, M, M, 9, 32, 16, 0, s_m, 3, 1, s_c, False, buf[o[1]:o[2]], 16)       # B = kw + 1
        pack_weights(flat[rest + 2:], bf, M, M, 3, 32, 16, 0, s_m, 3, 9, s_c, False, buf[o[2]:o[3]], 16)   # kw 2: B = kd + 1
        pack_weights(flat[rest + 20:], bf, M, M, 3, 16, 16, 0, s_m, 3, 0, s_c, False, buf[o[3]:o[4]], 16)  # (kd 2, kw 2)
    else:
        # the image's tap (kd, kh, kw) is the layer's tap (2-kd, 2-kh, 2-kw): the packer's flip reverses the T index, the
        # base offset is the flipped kw / kd of tap A and tap B lies one tap / one plane BEFORE it
        pack_weights(flat[rest + 2:], bf, M, M, 9, 32, 16, 0, s_m, 3, -1, s_c, True, buf[o[1]:o[2]], 16)
        pack_weights(flat[rest + 18:], bf, M, M, 3, 32, 16, 0, s_m, 3, -9, s_c, True, buf[o[2]:o[3]], 16)
        pack_weights(flat[rest:], bf, M, M, 3, 16, 16, 0, s_m, 3, 0, s_c, True, buf[o[3]:o[4]], 16)
    return buf


def pack_conv_k1(w: torch.Tensor, dtype, dgrad=False, out=None):
    co, ci = w.shape[0], w.shape[1]
    if not dgrad:
        return pack_weights(w, dtype, co, co, 1, ci, ci, 0, ci, 0, 0, 1, False, out)
    return pack_weights(w, dtype, ci, ci, 1, co, co, 0, 1, 0, 0, ci, False, out)


def pack_conv_gather(w: torch.Tensor, dtype, out=None):
    """w: [Cout, Cin, k,k,k]; logical K index = tap*Cin + ci."""
    co, ci = w.shape[0], w.shape[1]
    kt = w.shape[2] * w.shape[3] * w.shape[4]
    return pack_weights(w, dtype, co, co, 1, ci * kt, ci, 0, ci * kt, 0, 1, kt, False, out)


def pack_deconv(w: torch.Tensor, dtype, bwd=False, out=None):
    """w: ConvTranspose3d weight [Cin, Cout, 2,2,2].  fwd: M = abc*Cout+co, K = ci.  bwd-data: M = ci, K = abc*Cout+co."""
    ci, co = w.shape[0], w.shape[1]
    if not bwd:
        return pack_weights(w, dtype, 8 * co, co, 1, ci, ci, 1, 8, 0, 0, co * 8, False, out)
    return pack_weights(w, dtype, ci, ci, 1, 8 * co, co, 0, co * 8, 0, 1, 8, False, out)


# --------------------------------------------------------------------------------------------
# optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg)
# --------------------------------------------------------------------------------------------
class KernelTimer:
    """Per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  While `enabled`, EVERY call into
    the library that launches kernels is bracketed by an event pair and filed under its entry point's name (`lib()` hands
    out a timing proxy), so the dominant group of a step is found by measurement; `launch()` lets a wrapper file a call
    under its own key with the call's algorithmic flops / bytes.  An entry point's time includes the small follow-up
    kernels it launches itself (statistics finalize, slab reduction); the main kernels' own durations come from the
    library's `msseg_ktimer_*` events, recorded directly around those launches (`ktimer_summary`)."""

    def __init__(self):
        self.records = {}
        self.enabled = False
        self._pending = None

    def launch(self, key, flops, nbytes, fn):
        if not self.enabled:
            return fn()
        self._pending = (key, flops, nbytes)
        try:
            return fn()
        finally:
            self._pending = None

    def call(self, name, cfn, args):
        ann, self._pending = self._pending, None
        key, flops, nbytes = ann if ann is not None else (name[6:], 0.0, 0.0)
        st = torch.cuda.current_stream()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        r = cfn(*args)
        e1.record(st)
        self.records.setdefault(key, []).append((e0, e1, flops, nbytes))
        return r

    def summary(self):
        out = {}
        for k, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in recs]
            out[k] = {"launches": len(recs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms),
                      "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs)}
        return out


class _TimedLib:
    """what `lib()` returns while TIMER.enabled: every launching entry point goes through TIMER.call"""
    _QUERY = ("_bytes", "_block", "_variant", "_kernel", "_ok", "_groups", "abi_version", "last_error", "num_cus")

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name.endswith(self._QUERY) or name.startswith("msseg_ktimer"):
            return fn
        return lambda *a: TIMER.call(name, fn, a)


def ktimer_enable(on: bool):
    """library-side event pairs directly around the main conv kernels (msseg_ktimer_*, include/msseg.h)"""
    load_library().msseg_ktimer_enable(int(on))


def ktimer_summary(reset=True):
    """{kernel name: {"launches", "total_ms", "avg_ms"}} of the launches recorded since the last reset (synchronises)"""
    L = load_library()
    torch.cuda.synchronize()
    out = {}
    buf = C.create_string_buffer(96)
    ms = C.c_float()
    for i in range(L.msseg_ktimer_count()):
        if L.msseg_ktimer_get(i, buf, 96, C.byref(ms)) == 0:
            r = out.setdefault(buf.value.decode(), {"launches": 0, "total_ms": 0.0})
            r["launches"] += 1
            r["total_ms"] += float(ms.value)
    for r in out.values():
        r["avg_ms"] = r["total_ms"] / r["launches"]
    if reset:
        L.msseg_ktimer_reset()
    return out


TIMER = KernelTimer()


# --------------------------------------------------------------------------------------------
# igemm forward-shaped ops.  x, y: [N, D, H, W, C] channels-last (views allowed)
# --------------------------------------------------------------------------------------------
_scratch_cache = {}


def scratch(device) -> torch.Tensor:
    """zero-initialised scratch for the deterministic two-stage reductions (one per device, stable address)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    buf = _scratch_cache.get(key)
    if buf is None:
        buf = torch.zeros(lib().msseg_reduce_scratch_bytes() + 256, dtype=torch.uint8, device=device)
        off = (-buf.data_ptr()) % 256
        buf = buf[off:off + lib().msseg_reduce_scratch_bytes()]
        _scratch_cache[key] = buf
    return buf


def conv3d_k3(x, wp, bias, y, cin, cout, stats=None):
    """stats: optional [N, cout, 2] fp32 output = (sum, sum of squares) of y (fused InstanceNorm statistics)."""
    _need_gpu(x, wp, y)
    N, D, H, W = x.shape[:4]
    nv = N * D * H * W
    esz = x.element_size()
    sc = scratch(x.device) if stats is not None else None

    def go():
        _ck(lib().msseg_conv3d_k3_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y), N, D, H, W, cin, cout, _p(stats),
                                      _p(sc), sc.numel() if sc is not None else 0, dt(x), _stream()), "conv3d_k3_fwd")
    key = "conv3d_k3_fwd"
    if TIMER.enabled:
        key += "/v%d" % lib().msseg_conv3d_k3_kernel(N, D, H, W, cin, cout, dt(x))
    TIMER.launch(key, 2.0 * nv * 27 * cin * cout, nv * (cin + cout) * esz + 27 * cin * cout * esz, go)
    return y


def conv3d_k3_accumulate(x, wp, y, cin, cout, stats):
    """y += conv k3 (x, packed image) on the ping-pong kernel; stats [N, cout, 2] = statistics of the sums"""
    _need_gpu(x, wp, y, stats)
    N, D, H, W = x.shape[:4]
    nv = N * D * H * W
    sc = scratch(x.device)
    TIMER.launch("conv3d_k3_fwd/v%d" % (4 if cin == 48 else 3), 2.0 * nv * 27 * cin * cout, nv * (cin + 2 * cout) * x.element_size() + 27 * cin * cout * 2,
                 lambda: _ck(lib().msseg_conv3d_k3_fwd_accumulate(_p(x), ld(x), _p(wp), _p(y), ld(y), N, D, H, W, cin, cout,
                                                                 _p(stats), _p(sc), sc.numel(), dt(x), _stream()),
                             "conv3d_k3_fwd_accumulate"))
    return y


def conv3d_k3_dgrad_inbwd(dy, wp, da, cin, cout, yraw, act, fwd_stats, slope, eps, dgamma=None, dbeta=None,
                          accumulate=False):
    """da = conv(dy, dgrad image) and, in the same launch, red[N][cout][2] = InstanceNorm-backward reductions of the
    layer (yraw, act, fwd_stats) that receives da.  Returns red."""
    _need_gpu(dy, wp, da, yraw, act, fwd_stats)
    N, D, H, W = dy.shape[:4]
    nv = N * D * H * W
    esz = dy.element_size()
    red = torch.empty(N, cout, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)

    def go():
        _ck(lib().msseg_conv3d_k3_dgrad_inbwd(_p(dy), ld(dy), _p(wp), _p(da), ld(da), N, D, H, W, cin, cout, _p(yraw),
                                              ld(yraw), _p(act), ld(act), _p(fwd_stats), slope, eps, _p(red),
                                              _p(dgamma), _p(dbeta), int(accumulate), _p(sc), sc.numel(), dt(dy),
                                              _stream()), "conv3d_k3_dgrad_inbwd")
    key = "conv3d_k3_fwd"
    if TIMER.enabled:
        key += "/v%d" % lib().msseg_conv3d_k3_kernel(N, D, H, W, cin, cout, dt(dy))
    TIMER.launch(key, 2.0 * nv * 27 * cin * cout, nv * (cin + 3 * cout) * esz + 27 * cin * cout * esz, go)
    return red


# --------------------------------------------------------------------------------------------
# conv k3 on small grids: split-K partials + a finish kernel that carries the rest of the unit (conv3d_k3_small.hip)
# --------------------------------------------------------------------------------------------
def conv3d_k3_small_ok(x, cin, cout) -> bool:
    """can the split-K small-grid path run conv k3 cin -> cout on a volume shaped like x ([N, D, H, W, C])?"""
    if os.environ.get("MSSEG_NO_K3_SMALL") or x.dim() != 5 or x.dtype != torch.bfloat16:
        return False
    N, D, H, W = x.shape[:4]
    return bool(lib().msseg_conv3d_k3_small_ok(N, D, H, W, cin, cout, BF16))


_k3s_ws = {}


def _k3s_workspace(nbytes, device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    buf = _k3s_ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            _ws_retired.append(buf)       # captured graphs keep the old address
        buf = torch.empty(max(nbytes, 32 << 20) // 4, dtype=torch.float32, device=device)
        _k3s_ws[key] = buf
    return buf


def conv3d_k3_small_partials(x, wp, cin, cout):
    """(fp32 partial sums [stage groups][N * D * H * W][cout] of conv k3 (x, packed image with cout block 32), number of
    stage groups); the buffer is the module's grow-only scratch: consume it (a *_finish call) before the next partials
    call on the stream"""
    _need_gpu(x, wp)
    N, D, H, W = x.shape[:4]
    nv = N * D * H * W
    ng = lib().msseg_conv3d_k3_small_stage_groups(N, D, H, W, cin, cout)
    part = _k3s_workspace(lib().msseg_conv3d_k3_small_workspace_bytes(N, D, H, W, cin, cout), x.device)
    TIMER.launch("conv3d_k3_small", 2.0 * nv * 27 * cin * cout, nv * (cin * x.element_size() + ng * cout * 4) + 27 * cin * cout * 2,
                 lambda: _ck(lib().msseg_conv3d_k3_small_partials(_p(x), ld(x), _p(wp), _p(part), part.numel() * 4, N, D, H, W, cin,
                                                                  cout, _stream()), "conv3d_k3_small_partials"))
    return part, ng


def deconv_k2s2_small_unit_ok(dx_shape, cin, cout, dtype) -> bool:
    """can the input gradient of ConvTranspose3d k2 s2 (cin -> cout) for a coarse volume dx_shape = (N, D, H, W, cin) go out as a
    partial block for conv3d_k3_small_bwd_finish (which then runs the receiving unit's whole InstanceNorm backward)?"""
    if os.environ.get("MSSEG_NO_K3_SMALL") or dtype != torch.bfloat16 or len(dx_shape) != 5:
        return False
    N, D, H, W = dx_shape[:4]
    return N <= 8 and D * H * W <= 2048 and bool(lib().msseg_deconv_k2s2_bwd_partials_ok(cin, cout, BF16))


def deconv_k2s2_bwd_partials(dy, wp, cin, cout):
    """fp32 partial block [cin / 4][N * D * H * W][4] (ONE stage group) of the input gradient of ConvTranspose3d k2 s2; dy: fine
    [N, 2D, 2H, 2W, cout]; the buffer is the small-grid scratch: consume it (conv3d_k3_small_bwd_finish(part, 1, ...)) before
    the next partials call on the stream"""
    _need_gpu(dy, wp)
    N, D, H, W = dy.shape[0], dy.shape[1] // 2, dy.shape[2] // 2, dy.shape[3] // 2
    nv = N * D * H * W
    part = _k3s_workspace(nv * cin * 4, dy.device)
    TIMER.launch("deconv_k2s2_bwd_partials", 2.0 * nv * 8 * cin * cout, nv * (8 * cout * dy.element_size() + cin * 4) + 8 * cin * cout * 2,
                 lambda: _ck(lib().msseg_deconv_k2s2_bwd_partials(_p(dy), ld(dy), _p(wp), _p(part), part.numel() * 4, N, D, H, W, cin,
                                                                  cout, dt(dy), _stream()), "deconv_k2s2_bwd_partials"))
    return part


def conv3d_k3_small_fwd_finish(part, nstages, bias, gamma, beta, eps, slope, yraw, act, pooled, stats, residual=None):
    """residual (optional, same shape as act): act = lrelu(instance_norm(y) * gamma + beta + residual)"""
    _need_gpu(part, yraw, act, stats)
    N, D, H, W, cout = yraw.shape
    _ck(lib().msseg_conv3d_k3_small_fwd_finish_res(_p(part), nstages, _p(bias), _p(gamma), _p(beta), eps, slope, _p(yraw), ld(yraw),
                                                   _p(act), ld(act), _p(residual), ld(residual) if residual is not None else 0,
                                                   _p(pooled), ld(pooled) if pooled is not None else 0,
                                                   _p(stats), N, D, H, W, cout, _stream()), "conv3d_k3_small_fwd_finish")


def conv3d_k3_small_bwd_finish(part, nstages, dx, unit=None, dgamma=None, dbeta=None, accumulate=False):
    """unit = (yraw, stats, gamma, beta, eps, slope) of the conv + InstanceNorm + LeakyReLU unit whose activation was the
    conv's input: dx then receives that unit's dy (and dgamma / dbeta its affine gradients); None: dx = the plain sum"""
    _need_gpu(part, dx)
    N, D, H, W, cin = dx.shape
    uy = ustats = ug = ub = None
    eps, slope = 1e-5, 1.0
    if unit is not None:
        uy, ustats, ug, ub, eps, slope = unit
    _ck(lib().msseg_conv3d_k3_small_bwd_finish(_p(part), nstages, _p(dx), ld(dx), _p(uy), ld(uy) if uy is not None else 0,
                                               _p(ustats), _p(ug), _p(ub), eps, slope, _p(dgamma), _p(dbeta),
                                               int(accumulate), N, D, H, W, cin, _stream()), "conv3d_k3_small_bwd_finish")
    return dx


def conv3d_k1_head(x, w, bias, y, cin, cout):
    """1x1x1 conv with 1..4 output channels straight from the fp32 weight [cout, cin] (streaming kernel)"""
    _need_gpu(x, w, y)
    nv = x.numel() // x.shape[-1]
    _ck(lib().msseg_conv3d_k1_head_fwd(_p(x), ld(x), _p(w), _p(bias), _p(y), ld(y), nv, cin, cout, dt(x), _stream()),
        "conv3d_k1_head_fwd")
    return y


def conv3d_k1_head_dgrad_inbwd(dy, w, da, cin, cout, yraw, fwd_stats, gamma, beta, slope, eps, dgamma=None, dbeta=None,
                               accumulate=False):
    """da = dy . w for a head with `cout` <= 4 classes (w fp32 [cout, cin]) and the InstanceNorm-backward sums of the layer
    (yraw, fwd_stats, gamma, beta) that receives da.  Returns red[N][cin][2]."""
    _need_gpu(dy, w, da, yraw, fwd_stats)
    N = dy.shape[0]
    S = dy.numel() // dy.shape[-1] // N
    red = torch.empty(N, cin, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)
    _ck(lib().msseg_conv3d_k1_head_dgrad_inbwd(_p(dy), ld(dy), _p(w), _p(da), ld(da), N, S, cin, cout, _p(yraw), ld(yraw),
                                               _p(fwd_stats), _p(gamma), _p(beta), slope, eps, _p(red), _p(dgamma),
                                               _p(dbeta), int(accumulate), _p(sc), sc.numel(), dt(dy), _stream()),
        "conv3d_k1_head_dgrad_inbwd")
    return red


def conv3d_k1_head_norm(yraw, stats, gamma, beta, slope, eps, w, bias, y, cin, cout):
    """y = conv1x1(lrelu(instance_norm(yraw))) for a head with <= 4 classes, normalisation applied on load"""
    _need_gpu(yraw, stats, w, y)
    N = yraw.shape[0]
    S = yraw.numel() // yraw.shape[-1] // N
    _ck(lib().msseg_conv3d_k1_head_norm_fwd(_p(yraw), ld(yraw), _p(stats), _p(gamma), _p(beta), slope, eps, _p(w), _p(bias),
                                            _p(y), ld(y), N, S, cin, cout, dt(yraw), _stream()), "conv3d_k1_head_norm_fwd")
    return y


def conv3d_k1_head_bwd_fused(dy, w, da, cin, cout, yraw, fwd_stats, gamma, beta, slope, eps, dw, dw_accumulate,
                             dgamma=None, dbeta=None, accumulate=False):
    """conv3d_k1_head_dgrad_inbwd + the head's weight gradient dw[cout][cin] from the recomputed activation"""
    _need_gpu(dy, w, da, yraw, fwd_stats, dw)
    N = dy.shape[0]
    S = dy.numel() // dy.shape[-1] // N
    red = torch.empty(N, cin, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)
    _ck(lib().msseg_conv3d_k1_head_bwd_fused(_p(dy), ld(dy), _p(w), _p(da), ld(da), N, S, cin, cout, _p(yraw), ld(yraw),
                                             _p(fwd_stats), _p(gamma), _p(beta), slope, eps, _p(red), _p(dgamma),
                                             _p(dbeta), int(accumulate), _p(dw), int(dw_accumulate), _p(sc), sc.numel(),
                                             dt(dy), _stream()), "conv3d_k1_head_bwd_fused")
    return red


def conv3d_k1(x, wp, bias, y, cin, cout):
    _need_gpu(x, wp, y)
    nv = x.numel() // x.shape[-1]
    _ck(lib().msseg_conv3d_k1_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y), nv, cin, cout, dt(x), _stream()),
        "conv3d_k1_fwd")
    return y


def conv3d_k1_dgrad_inbwd(dy, wp, da, cin, cout, yraw, act, fwd_stats, slope, eps, dgamma=None, dbeta=None,
                          accumulate=False):
    """da = dy @ W (1x1x1 input gradient, cin = channels of dy, cout = channels of da) + the InstanceNorm-backward sums of
    the layer (yraw, act, fwd_stats) that receives da, in one launch.  Returns red[N][cout][2]."""
    _need_gpu(dy, wp, da, yraw, act, fwd_stats)
    N = da.shape[0]
    S = da.numel() // (N * da.shape[-1])
    red = torch.empty(N, cout, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)
    _ck(lib().msseg_conv3d_k1_dgrad_inbwd(_p(dy), ld(dy), _p(wp), _p(da), ld(da), N, S, cin, cout, _p(yraw), ld(yraw),
                                          _p(act), ld(act), _p(fwd_stats), slope, eps, _p(red), _p(dgamma), _p(dbeta),
                                          int(accumulate), _p(sc), sc.numel(), dt(dy), _stream()), "conv3d_k1_dgrad_inbwd")
    return red


def deconv_k2s2_bwd_data_inbwd(dy, wp, dx, cin, cout, yraw, act, fwd_stats, slope, eps, dgamma=None, dbeta=None,
                               accumulate=False):
    """dx of ConvTranspose3d k2 s2 + the InstanceNorm-backward sums of the layer that receives dx.  Returns red[N][cin][2]."""
    _need_gpu(dy, wp, dx, yraw, act, fwd_stats)
    N, D, H, W = dx.shape[:4]
    red = torch.empty(N, cin, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)
    _ck(lib().msseg_deconv_k2s2_bwd_data_inbwd(_p(dy), ld(dy), _p(wp), _p(dx), ld(dx), N, D, H, W, cin, cout, _p(yraw),
                                               ld(yraw), _p(act), ld(act), _p(fwd_stats), slope, eps, _p(red),
                                               _p(dgamma), _p(dbeta), int(accumulate), _p(sc), sc.numel(), dt(dy),
                                               _stream()), "deconv_k2s2_bwd_data_inbwd")
    return red


def deconv_k2s2_bwd_fused(dy, wp, dx, cin, cout, next_norm=None, dbias=None, dbias_accumulate=False, dgamma=None, dbeta=None,
                          accumulate=False):
    """dx of ConvTranspose3d k2 s2; next_norm = (yraw, act, fwd_stats, slope, eps) of the layer that receives dx adds its
    InstanceNorm-backward sums (returned as red[N][cin][2]); dbias (fp32 [cout]) receives the bias gradient."""
    _need_gpu(dy, wp, dx)
    N, D, H, W = dx.shape[:4]
    red = None
    yraw = act = stats = None
    slope = eps = 0.0
    if next_norm is not None:
        yraw, act, stats, slope, eps = next_norm
        red = torch.empty(N, cin, 2, dtype=torch.float32, device=dy.device)
    sc = scratch(dy.device)
    _ck(lib().msseg_deconv_k2s2_bwd_fused(_p(dy), ld(dy), _p(wp), _p(dx), ld(dx), N, D, H, W, cin, cout, _p(yraw),
                                          ld(yraw) if yraw is not None else 0, _p(act), ld(act) if act is not None else 0,
                                          _p(stats), slope, eps, _p(red), _p(dgamma), _p(dbeta), int(accumulate), _p(dbias),
                                          int(dbias_accumulate), _p(sc), sc.numel(), dt(dy), _stream()),
        "deconv_k2s2_bwd_fused")
    return red


def conv3d_gather(x, wp, bias, y, cin, cout, k, s, p):
    _need_gpu(x, wp, y)
    N, D, H, W = x.shape[:4]
    _ck(lib().msseg_conv3d_gather_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y), N, D, H, W, cin, cout, k, s, p,
                                      dt(x), _stream()), "conv3d_gather_fwd")
    return y


def conv3d_stem_norm(x, wp, bias, stats, gamma, beta, eps, slope, y, cout):
    """y = lrelu(instance_norm(conv k3 (x) + bias) * gamma + beta) of a ONE-channel volume with the statistics of a
    statistics-only conv3d_stem(x, ..., y=None, stats=...) call (inference: no raw output)"""
    _need_gpu(x, wp, y, stats)
    N, D, H, W = x.shape[:4]
    _ck(lib().msseg_conv3d_stem_norm_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(stats), _p(gamma), _p(beta), eps, slope, _p(y), ld(y),
                                         N, D, H, W, cout, dt(x), _stream()), "conv3d_stem_norm_fwd")
    return y


def conv3d_stem(x, wp, bias, y, cout, stats=None, k=3):
    """conv3d k3 p1 (k = 1: the 1x1x1 conv, image with K = 1) of a ONE-channel volume (bf16, cout % 32 == 0 or % 48 == 0) with
    optional fused InstanceNorm statistics; y None (with stats): statistics only"""
    _need_gpu(x, wp)
    N, D, H, W = x.shape[:4]
    sc = scratch(x.device) if stats is not None else None
    fn = lib().msseg_conv3d_stem_fwd if k == 3 else lib().msseg_conv3d_stem_k1_fwd
    _ck(fn(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y) if y is not None else 0, N, D, H, W, cout, _p(stats), _p(sc),
                                    sc.numel() if sc is not None else 0, dt(x), _stream()), "conv3d_stem_fwd")
    return y


def deconv_k2s2(x, wp, bias, y, cin, cout):
    _need_gpu(x, wp, y)
    N, D, H, W = x.shape[:4]
    _ck(lib().msseg_deconv_k2s2_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y), N, D, H, W, cin, cout, dt(x),
                                    _stream()), "deconv_k2s2_fwd")
    return y


def deconv_k2s2_bwd_data(dy, wp, dx, cin, cout):
    _need_gpu(dy, wp, dx)
    N, D, H, W = dx.shape[:4]
    _ck(lib().msseg_deconv_k2s2_bwd_data(_p(dy), ld(dy), _p(wp), _p(dx), ld(dx), N, D, H, W, cin, cout, dt(dy),
                                         _stream()), "deconv_k2s2_bwd_data")
    return dx


# --------------------------------------------------------------------------------------------
# weight gradients
# --------------------------------------------------------------------------------------------
_ws_cache = {}
_ws_retired = []   # superseded workspaces: captured hipGraphs may still launch kernels against them, so they are never freed


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per device.  A larger request allocates a new buffer; the old one is retired, not
    released, because graphs captured earlier keep its address baked into their kernel arguments."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _wg_ws(M, T, K, device):
    return workspace(lib().msseg_wgrad_workspace_bytes(M, T, K), device)


def conv3d_k3_wgrad(x, dy, dw, cin, cout, accumulate=False):
    _need_gpu(x, dy, dw)
    N, D, H, W = x.shape[:4]
    ws = _wg_ws(cout, 27, cin, x.device)
    nv = N * D * H * W

    def go():
        _ck(lib().msseg_conv3d_k3_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), N, D, H, W, cin, cout, int(accumulate),
                                        _p(ws), ws.numel(), dt(x), _stream()), "conv3d_k3_wgrad")
    key = "conv3d_k3_wgrad"
    if TIMER.enabled:
        key += "/v%d" % lib().msseg_conv3d_k3_wgrad_kernel(N, D, H, W, cin, cout, dt(x))
    TIMER.launch(key, 2.0 * nv * 27 * cin * cout, nv * (cin + cout) * x.element_size() + 27 * cin * cout * 4, go)


def conv3d_k1_wgrad(x, dy, dw, cin, cout, accumulate=False):
    _need_gpu(x, dy, dw)
    nv = x.numel() // x.shape[-1]
    ws = _wg_ws(cout, 1, cin, x.device)
    _ck(lib().msseg_conv3d_k1_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), nv, cin, cout, int(accumulate), _p(ws),
                                    ws.numel(), dt(x), _stream()), "conv3d_k1_wgrad")


def linear_wgrad_ok(x, cin, cout) -> bool:
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and
                lib().msseg_linear_wgrad_ok(x.numel() // x.shape[-1], cin, cout, BF16))


def linear_wgrad(x, dy, dw, dbias, cin, cout, accumulate_w=False, accumulate_b=False):
    """dw[cout][cin] (+)= dy^T x and dbias[cout] (+)= dy.sum(tokens) (dbias None: weight only) in one pass over the tokens"""
    _need_gpu(x, dy, dw)
    nv = x.numel() // x.shape[-1]
    ws = _wg_ws(cout, 1, cin, x.device)
    esz = x.element_size()
    TIMER.launch("linear_wgrad", 2.0 * nv * cin * cout, nv * (cin + cout) * esz + cin * cout * 4,
                 lambda: _ck(lib().msseg_linear_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), _p(dbias), nv, cin, cout,
                                                      int(accumulate_w), int(accumulate_b), _p(ws), ws.numel(), dt(x),
                                                      _stream()), "linear_wgrad"))


def conv3d_gather_wgrad(x, dy, dw, cin, cout, k, s, p, accumulate=False):
    _need_gpu(x, dy, dw)
    N, D, H, W = x.shape[:4]
    ws = _wg_ws(cout, 1, cin * k ** 3, x.device)
    _ck(lib().msseg_conv3d_gather_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), N, D, H, W, cin, cout, k, s, p,
                                        int(accumulate), _p(ws), ws.numel(), dt(x), _stream()), "conv3d_gather_wgrad")


def deconv_k2s2_wgrad(x, dy, dw, cin, cout, accumulate=False):
    _need_gpu(x, dy, dw)
    N, D, H, W = x.shape[:4]
    ws = _wg_ws(cin, 1, 8 * cout, x.device)
    _ck(lib().msseg_deconv_k2s2_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), N, D, H, W, cin, cout, int(accumulate),
                                      _p(ws), ws.numel(), dt(x), _stream()), "deconv_k2s2_wgrad")


# --------------------------------------------------------------------------------------------
# norm / act / pool / layout
# --------------------------------------------------------------------------------------------
def _nsc(x):
    N, C = x.shape[0], x.shape[-1]
    return N, x.numel() // (N * C), C


def dwconv3d_k3(x, w_taps, bias, y, flip=False):
    """depthwise conv k3 p1 on channels-last x; w_taps [27, C] (tap-major, dtype of x); flip: the input gradient"""
    if w_taps.dtype != x.dtype:
        raise ValueError("dwconv3d_k3: the weight table must have the activation dtype")
    _need_gpu(x, w_taps, y)
    N, D, H, W, C = x.shape
    _ck(lib().msseg_dwconv3d_k3_fwd(_p(x), ld(x), _p(w_taps), _p(bias), _p(y), ld(y), N, D, H, W, C, int(flip), dt(x),
                                    _stream()), "dwconv3d_k3_fwd")
    return y


def avgpool3d_k3(x, y):
    """AvgPool3d(3, 1, 1) with count_include_pad: fp32 sum of the 27 taps x 1/27, one rounding"""
    _need_gpu(x, y)
    N, D, H, W, Cc = x.shape
    _ck(lib().msseg_avgpool3d_k3(_p(x), ld(x), _p(y), ld(y), N, D, H, W, Cc, dt(x), _stream()), "avgpool3d_k3")
    return y


def dwconv3d_k3_wgrad(x, dy, dw, dbias, acc_w=False, acc_b=False):
    """dw fp32 [C, 1, 3, 3, 3], dbias fp32 [C] (either may be None)"""
    _need_gpu(x, dy)
    N, D, H, W, C = x.shape
    sc = scratch(x.device)
    _ck(lib().msseg_dwconv3d_k3_wgrad(_p(x), ld(x), _p(dy), ld(dy), _p(dw), _p(dbias), int(acc_w), int(acc_b), N, D, H, W, C,
                                      _p(sc), sc.numel(), dt(x), _stream()), "dwconv3d_k3_wgrad")


def interp_trilinear(x, y):
    """y = F.interpolate(x, size=y.shape[1:4], mode='trilinear', align_corners=False), both channels-last"""
    _need_gpu(x, y)
    N, ID, IH, IW, C = x.shape
    _ck(lib().msseg_interp_trilinear_fwd(_p(x), ld(x), _p(y), ld(y), N, ID, IH, IW, y.shape[1], y.shape[2], y.shape[3], C,
                                         dt(x), _stream()), "interp_trilinear_fwd")
    return y


def interp_trilinear_bwd(dy, dx):
    _need_gpu(dy, dx)
    N, ID, IH, IW, C = dx.shape
    OD, OH, OW = dy.shape[1:4]
    nb = lib().msseg_interp_trilinear_bwd_workspace_bytes(N, ID, IH, IW, OD, OH, OW, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dx.device)
    _ck(lib().msseg_interp_trilinear_bwd(_p(dy), ld(dy), _p(dx), ld(dx), N, ID, IH, IW, OD, OH, OW, C, _p(ws), nb, dt(dx),
                                         _stream()), "interp_trilinear_bwd")
    return dx


def kv_attention_fwd(q, kv, heads, scale):
    """q [B, N, C], kv [B, M, 2C] -> (o [B, N, C], lse [B, heads, N])"""
    _need_gpu(q, kv)
    B, N, C = q.shape
    M = kv.shape[1]
    o = torch.empty_like(q)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=q.device)
    _ck(lib().msseg_kv_attention_fwd(_p(q), _p(kv), _p(o), _p(lse), B, N, M, heads, C // heads, scale, dt(q), _stream()),
        "kv_attention_fwd")
    return o, lse


def kv_attention_bwd(q, kv, o, lse, dout, heads, scale):
    _need_gpu(q, kv, o, lse, dout)
    B, N, C = q.shape
    M = kv.shape[1]
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    nb = lib().msseg_kv_attention_bwd_workspace_bytes(B, N, M, heads, C // heads)
    ws = torch.empty(nb, dtype=torch.uint8, device=q.device)
    _ck(lib().msseg_kv_attention_bwd(_p(q), _p(kv), _p(o), _p(lse), _p(dout), _p(dq), _p(dkv), B, N, M, heads, C // heads,
                                     scale, _p(ws), nb, dt(q), _stream()), "kv_attention_bwd")
    return dq, dkv


def scale_channels(x, scale, y):
    """y[n, ..., c] = x[n, ..., c] * scale[n, c] (fp32 scale)"""
    _need_gpu(x, scale, y)
    N, C = x.shape[0], x.shape[-1]
    _ck(lib().msseg_scale_channels(_p(x), _p(scale), _p(y), N, x.numel() // (N * C), C, dt(x), _stream()), "scale_channels")
    return y


def channel_stats(x, stats=None):
    _need_gpu(x)
    N, S, Cc = _nsc(x)
    if stats is None:
        stats = torch.empty(N, Cc, 2, dtype=torch.float32, device=x.device)
    sc = scratch(x.device)
    _ck(lib().msseg_channel_stats(_p(x), ld(x), _p(stats), N, S, Cc, _p(sc), sc.numel(), dt(x), _stream()),
        "channel_stats")
    return stats


def instnorm_act_fwd(x, stats, gamma, beta, y, slope, eps=1e-5, residual=None):
    _need_gpu(x, stats, y)
    N, S, Cc = _nsc(x)
    TIMER.launch("instnorm_act_fwd", 0.0, _nbytes(x, y, residual), lambda: _ck(lib().msseg_instnorm_act_fwd(_p(x), ld(x), _p(stats), _p(gamma), _p(beta), _p(residual),
                                     ld(residual) if residual is not None else 0, _p(y), ld(y), N, S, Cc, eps, slope,
                                     dt(x), _stream()), "instnorm_act_fwd"))
    return y


def instnorm_pool_ok(x, y, pooled) -> bool:
    """the fused InstanceNorm + LeakyReLU + MaxPool3d(2) kernel wants even spatial sizes and 16-byte channel chunks"""
    epc = 16 // x.element_size()
    return (x.dim() == 5 and all(int(d) % 2 == 0 for d in x.shape[1:4]) and x.shape[-1] % epc == 0
            and all(ld(t) % epc == 0 and t.data_ptr() % 16 == 0 for t in (x, y, pooled)))


def instnorm_act_pool_fwd(x, stats, gamma, beta, y, pooled, slope, eps=1e-5):
    """y = lrelu(instance_norm(x)) and pooled = max_pool3d(y, 2) in one pass over x (channels-last [N, D, H, W, C])"""
    _need_gpu(x, stats, y, pooled)
    N, D, H, W, Cc = x.shape
    TIMER.launch("instnorm_act_pool_fwd", 0.0, _nbytes(x, y, pooled), lambda: _ck(lib().msseg_instnorm_act_pool_fwd(_p(x), ld(x), _p(stats), _p(gamma), _p(beta), _p(y), ld(y), _p(pooled), ld(pooled),
                                          N, D, H, W, Cc, eps, slope, dt(x), _stream()), "instnorm_act_pool_fwd"))
    return y, pooled


def instnorm_act_poolbwd_reduce(x, stats, gamma, beta, skip, g, da, slope, eps=1e-5, dgamma=None, dbeta=None,
                                accumulate=False):
    """da = skip + max_pool3d-backward(lrelu(instance_norm(x)), g) (dense) and the InstanceNorm-backward sums of da;
    returns red[N][C][2].  Follow with instnorm_act_bwd_apply(x, ..., dy=da, red)."""
    _need_gpu(x, stats, skip, g, da)
    N, D, H, W, Cc = x.shape
    red = torch.empty(N, Cc, 2, dtype=torch.float32, device=x.device)
    sc = scratch(x.device)
    TIMER.launch("instnorm_act_poolbwd_reduce", 0.0, _nbytes(x, skip, g, da), lambda: _ck(lib().msseg_instnorm_act_poolbwd_reduce(_p(x), ld(x), _p(stats), _p(gamma), _p(beta), _p(skip), ld(skip), _p(g),
                                                ld(g), _p(da), ld(da), _p(red), _p(dgamma), _p(dbeta), int(accumulate),
                                                N, D, H, W, Cc, eps, slope, _p(sc), sc.numel(), dt(x), _stream()),
        "instnorm_act_poolbwd_reduce"))
    return red


def instnorm_act_bwd(x, stats, gamma, y, dy, dx, slope, eps=1e-5, dres=None, dgamma=None, dbeta=None,
                     accumulate=False, beta=None):
    """dx (and dres) from dy; the affine gradients dgamma/dbeta (fp32 [C]) are written (or accumulated) by the
    reduce kernel's finalising block.  Returns red[N][C][2] = (sum dz, sum dz*xhat)."""
    _need_gpu(x, stats, dy, dx)
    red = instnorm_act_bwd_reduce(x, stats, gamma, y, dy, slope, eps, dgamma, dbeta, accumulate, beta)
    instnorm_act_bwd_apply(x, stats, gamma, y, dy, red, dx, slope, eps, dres, beta)
    return red


def instnorm_act_bwd_reduce(x, stats, gamma, y, dy, slope, eps=1e-5, dgamma=None, dbeta=None, accumulate=False, beta=None):
    """first half of instnorm_act_bwd: red[N][C][2] = (sum dz, sum dz*xhat) (+ dgamma / dbeta)"""
    _need_gpu(x, stats, dy)
    N, S, Cc = _nsc(x)
    red = torch.empty(N, Cc, 2, dtype=torch.float32, device=x.device)
    sc = scratch(x.device)
    TIMER.launch("instnorm_act_bwd_reduce", 0.0, _nbytes(x, y, dy), lambda: _ck(lib().msseg_instnorm_act_bwd_reduce(_p(x), ld(x), _p(stats), _p(gamma), _p(beta), _p(y),
                                            ld(y) if y is not None else 0, _p(dy), ld(dy), _p(red),
                                            _p(dgamma), _p(dbeta), int(accumulate), N, S, Cc, eps, slope, _p(sc),
                                            sc.numel(), dt(x), _stream()), "instnorm_act_bwd_reduce"))
    return red


def instnorm_act_bwd_apply(x, stats, gamma, y, dy, red, dx, slope, eps=1e-5, dres=None, beta=None):
    """y None: the sign of the pre-activation is recomputed from x (layers without residual)"""
    _need_gpu(x, stats, dy, dx, red)
    N, S, Cc = _nsc(x)
    TIMER.launch("instnorm_act_bwd_apply", 0.0, _nbytes(x, y, dy, dx, dres), lambda: _ck(lib().msseg_instnorm_act_bwd_apply(_p(x), ld(x), _p(stats), _p(gamma), _p(beta), _p(y),
                                           ld(y) if y is not None else 0, _p(dy), ld(dy), _p(red),
                                           _p(dx), ld(dx), _p(dres), ld(dres) if dres is not None else 0, N, S, Cc, eps,
                                           slope, dt(x), _stream()), "instnorm_act_bwd_apply"))
    return red


def maxpool2_fwd(x, y):
    _need_gpu(x, y)
    N, D, H, W, Cc = x.shape
    _ck(lib().msseg_maxpool2_fwd(_p(x), ld(x), _p(y), ld(y), N, D, H, W, Cc, dt(x), _stream()), "maxpool2_fwd")
    return y


def maxpool2_bwd(x, dy, dx, accumulate=False):
    _need_gpu(x, dy, dx)
    N, D, H, W, Cc = x.shape
    _ck(lib().msseg_maxpool2_bwd(_p(x), ld(x), _p(dy), ld(dy), _p(dx), ld(dx), N, D, H, W, Cc, int(accumulate), dt(x),
                                 _stream()), "maxpool2_bwd")
    return dx


def to_channels_last(src: torch.Tensor, dst: torch.Tensor):
    """src NCDHW contiguous -> dst [N, D, H, W, C] (dtype cast included)."""
    _need_gpu(src, dst)
    src = src.contiguous()
    N, Cc = src.shape[0], src.shape[1]
    S = src.numel() // (N * Cc)
    _ck(lib().msseg_ncdhw_to_ndhwc(_p(src), dt(src), _p(dst), ld(dst), dt(dst), N, Cc, S, _stream()), "ncdhw_to_ndhwc")
    return dst


def to_channels_first(src: torch.Tensor, dst: torch.Tensor):
    """src [N, D, H, W, C] (view allowed) -> dst NCDHW contiguous."""
    _need_gpu(src, dst)
    N, Cc = dst.shape[0], dst.shape[1]
    S = dst.numel() // (N * Cc)
    assert dst.is_contiguous()
    _ck(lib().msseg_ndhwc_to_ncdhw(_p(src), ld(src), dt(src), _p(dst), dt(dst), N, Cc, S, _stream()), "ndhwc_to_ncdhw")
    return dst


def channel_sum(x, out, accumulate=False):
    _need_gpu(x, out)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    sc = scratch(x.device)
    _ck(lib().msseg_channel_sum(_p(x), ld(x), _p(out), rows, Cc, int(accumulate), _p(sc), sc.numel(), dt(x),
                                _stream()), "channel_sum")
    return out


def axpy_rows(a, b, scale, y):
    """y[n] = a[n] + scale[n] * b[n] (a None: scale * b; scale None: plain add) on dense tensors [N, ...]"""
    _need_gpu(b, y)
    assert b.is_contiguous() and y.is_contiguous() and (a is None or a.is_contiguous())
    N = b.shape[0]
    _ck(lib().msseg_axpy_rows(_p(a), _p(b), _p(scale), _p(y), N, b.numel() // N, dt(b), _stream()), "axpy_rows")
    return y


def add(a, b, y):
    _need_gpu(a, b, y)
    Cc = a.shape[-1]
    rows = a.numel() // Cc
    _ck(lib().msseg_add(_p(a), ld(a), _p(b), ld(b), _p(y), ld(y), rows, Cc, dt(a), _stream()), "add")
    return y


# --------------------------------------------------------------------------------------------
# loss / metric
# --------------------------------------------------------------------------------------------
def dice_ce_partials(logits, labels, n_cls, channels_last_ld=0, want_hard=False):
    """logits: NCDHW (channels_last_ld == 0) or channels-last with voxel stride ld.  Returns (partial, hard)."""
    _need_gpu(logits, labels)
    N = logits.shape[0]
    S = labels.numel() // N
    partial = torch.zeros(N, n_cls, 4, dtype=torch.float32, device=logits.device)
    hard = torch.zeros(N, n_cls, 3, dtype=torch.float32, device=logits.device) if want_hard else None
    _ck(lib().msseg_dice_ce_partials(_p(logits), channels_last_ld, dt(logits), _p(labels), _LAB[labels.dtype],
                                     _p(partial), _p(hard), N, S, n_cls, _stream()), "dice_ce_partials")
    return partial, hard


def dice_ce_fwd(logits, labels, n_cls, smooth_nr, smooth_dr, channels_last_ld=0, want_hard=True):
    """deterministic fused forward: returns (partial [N,C,4], hard [N,C,3] or None, loss3 = (total, dice, ce))"""
    _need_gpu(logits, labels)
    N = logits.shape[0]
    S = labels.numel() // N
    partial = torch.empty(N, n_cls, 4, dtype=torch.float32, device=logits.device)
    hard = torch.empty(N, n_cls, 3, dtype=torch.float32, device=logits.device) if want_hard else None
    loss = torch.empty(3, dtype=torch.float32, device=logits.device)
    sc = scratch(logits.device)
    _ck(lib().msseg_dice_ce_fwd(_p(logits), channels_last_ld, dt(logits), _p(labels), _LAB[labels.dtype], _p(partial),
                                _p(hard), _p(loss), N, S, n_cls, smooth_nr, smooth_dr, _p(sc), sc.numel(), _stream()),
        "dice_ce_fwd")
    return partial, hard, loss


def dice_ce_finalize(partial, S, smooth_nr, smooth_dr):
    N, Cc = partial.shape[0], partial.shape[1]
    loss = torch.empty(3, dtype=torch.float32, device=partial.device)
    _ck(lib().msseg_dice_ce_finalize(_p(partial), _p(loss), N, S, Cc, smooth_nr, smooth_dr, _stream()),
        "dice_ce_finalize")
    return loss


def dice_ce_bwd(logits, labels, partial, gscale, dlogits, n_cls, smooth_nr, smooth_dr, ld_in=0, ld_out=0):
    _need_gpu(logits, labels, partial, dlogits)
    N = logits.shape[0]
    S = labels.numel() // N
    _ck(lib().msseg_dice_ce_bwd(_p(logits), ld_in, dt(logits), _p(labels), _LAB[labels.dtype], _p(partial),
                                _p(gscale), _p(dlogits), ld_out, N, S, n_cls, smooth_nr, smooth_dr, _stream()),
        "dice_ce_bwd")
    return dlogits


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
def adamw_step(param, grad, exp_avg, exp_avg_sq, decay_mask, lr, beta1, beta2, eps, weight_decay, step,
               grad_scale=None, dev_hyper=None):
    _need_gpu(param, grad, exp_avg, exp_avg_sq)
    _ck(lib().msseg_adamw_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(decay_mask), param.numel(), lr,
                               beta1, beta2, eps, weight_decay, step, _p(grad_scale), _p(dev_hyper), _stream()),
        "adamw_step")


def sumsq(x, out, partials):
    """out[0] = sum of x^2 in a fixed order; `partials`: fp32 buffer of the caller for the per-block sums (overwritten)."""
    _need_gpu(x, out, partials)
    if partials.dtype != torch.float32 or not partials.is_contiguous():
        raise ValueError("sumsq: partials must be a contiguous fp32 buffer")
    _ck(lib().msseg_sumsq(_p(x), x.numel(), _p(out), _p(partials), partials.numel(), _stream()), "sumsq")
    return out


# --------------------------------------------------------------------------------------------
# sliding window
# --------------------------------------------------------------------------------------------
def sw_blend(win, imp, out, cnt, start, channels_last_ld=0):
    """win: [C, *roi] (NCDHW, ld 0) ; out: [C, *vol] fp32 ; cnt: [*vol] fp32."""
    _need_gpu(win, imp, out, cnt)
    Cc = out.shape[0]
    VD, VH, VW = out.shape[1:]
    RD, RH, RW = imp.shape
    _ck(lib().msseg_sw_blend(_p(win), channels_last_ld, dt(win), _p(imp), _p(out), _p(cnt), Cc, VD, VH, VW, RD, RH, RW,
                             int(start[0]), int(start[1]), int(start[2]), _stream()), "sw_blend")


def sw_normalize(out, cnt):
    _need_gpu(out, cnt)
    _ck(lib().msseg_sw_normalize(_p(out), _p(cnt), out.shape[0], cnt.numel(), _stream()), "sw_normalize")
    return out


def sw_gather(vol, win, start, cval=0.0):
    """vol: [C, *vol] fp32 ; win: [C, *roi] (dtype of win)."""
    _need_gpu(vol, win)
    Cc = vol.shape[0]
    VD, VH, VW = vol.shape[1:]
    RD, RH, RW = win.shape[1:]
    _ck(lib().msseg_sw_gather(_p(vol), _p(win), dt(win), Cc, VD, VH, VW, RD, RH, RW, int(start[0]), int(start[1]),
                              int(start[2]), float(cval), _stream()), "sw_gather")
    return win


def sw_gather_batch(vol, win, table, nwin, cval=0.0, channels_last_ld=0):
    """vol: [B, C, *vol] fp32 ; win: [nwin, C, *roi] (ld 0) or channels-last [nwin, *roi, ld] ; table: int32 [>=nwin, 4]
    device tensor of (b, z0, y0, x0), b < 0 = unused slot."""
    _need_gpu(vol, win, table)
    assert vol.dtype == torch.float32 and vol.is_contiguous() and win.is_contiguous()
    assert table.dtype == torch.int32 and table.is_contiguous() and table.shape[0] >= nwin
    Cc = vol.shape[1]
    VD, VH, VW = vol.shape[2:]
    RD, RH, RW = win.shape[2:] if channels_last_ld == 0 else win.shape[1:4]
    _ck(lib().msseg_sw_gather_batch(_p(vol), vol.stride(0), _p(win), channels_last_ld, dt(win), _p(table), nwin, Cc, VD, VH,
                                    VW, RD, RH, RW, float(cval), _stream()), "sw_gather_batch")
    return win


def sw_blend_batch(win, imp, out, cnt, table, nwin, channels_last_ld=0):
    """out: [B, C, *vol] fp32 ; cnt: [B, *vol] fp32 ; win as in sw_gather_batch (C = out.shape[1] channels used)."""
    _need_gpu(win, imp, out, cnt, table)
    assert out.is_contiguous() and cnt.is_contiguous() and win.is_contiguous() and imp.is_contiguous()
    assert table.dtype == torch.int32 and table.is_contiguous() and table.shape[0] >= nwin
    Cc = out.shape[1]
    VD, VH, VW = out.shape[2:]
    RD, RH, RW = imp.shape
    _ck(lib().msseg_sw_blend_batch(_p(win), channels_last_ld, dt(win), _p(imp), _p(out), out.stride(0), _p(cnt),
                                   cnt.stride(0), _p(table), nwin, Cc, VD, VH, VW, RD, RH, RW, _stream()), "sw_blend_batch")


# --------------------------------------------------------------------------------------------
# post-inference
# --------------------------------------------------------------------------------------------
def argmax_u8(logits: torch.Tensor) -> torch.Tensor:
    """logits [C, D, H, W] fp32 (one sample, contiguous) -> uint8 [D, H, W], first maximum over C"""
    _need_gpu(logits)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 4
    out = torch.empty(logits.shape[1:], dtype=torch.uint8, device=logits.device)
    _ck(lib().msseg_argmax_u8(_p(logits), logits.shape[0], out.numel(), _p(out), _stream()), "argmax_u8")
    return out


def resample_nearest_u8(src: torch.Tensor, target_size) -> torch.Tensor:
    """uint8 [D, H, W] -> uint8 target_size, scipy.ndimage.zoom(order=0) semantics"""
    _need_gpu(src)
    assert src.dtype == torch.uint8 and src.is_contiguous() and src.dim() == 3
    td, th, tw = (int(v) for v in target_size)
    out = torch.empty(td, th, tw, dtype=torch.uint8, device=src.device)
    _ck(lib().msseg_resample_nearest_u8(_p(src), src.shape[0], src.shape[1], src.shape[2], _p(out), td, th, tw, _stream()),
        "resample_nearest_u8")
    return out


def majority_vote_u8(labels: torch.Tensor, n_classes: int) -> torch.Tensor:
    """labels [F, D, H, W] uint8 (one label map per fold) -> uint8 [D, H, W]"""
    _need_gpu(labels)
    assert labels.dtype == torch.uint8 and labels.is_contiguous() and labels.dim() >= 2
    out = torch.empty(labels.shape[1:], dtype=torch.uint8, device=labels.device)
    _ck(lib().msseg_majority_vote_u8(_p(labels), labels.shape[0], out.numel(), n_classes, _p(out), _stream()),
        "majority_vote_u8")
    return out


def hd_edges(pred: torch.Tensor, gt: torch.Tensor, n_classes: int):
    """pred / gt: uint8 label maps [D, H, W] -> (edge map of pred, edge map of gt, stats int32 [n_classes, 8])"""
    _need_gpu(pred, gt)
    assert pred.dtype == torch.uint8 and gt.dtype == torch.uint8 and pred.shape == gt.shape and pred.dim() == 3
    pred, gt = pred.contiguous(), gt.contiguous()
    D, H, W = pred.shape
    ep, eg = torch.empty_like(pred), torch.empty_like(gt)
    stats = torch.empty(n_classes, 8, dtype=torch.int32, device=pred.device)
    _ck(lib().msseg_hd_edges(_p(pred), _p(gt), D, H, W, n_classes, _p(ep), _p(eg), _p(stats), _stream()), "hd_edges")
    return ep, eg, stats


def hd_directed_hist(edges_src, edges_tgt, cls: int, box, hist: torch.Tensor):
    """histogram (int32, on the device) of the squared distances from the class-`cls` surface voxels of edges_tgt to the
    nearest class-`cls` surface voxel of edges_src inside box = (z0, y0, x0, z1, y1, x1); the last bin counts 'no source'"""
    _need_gpu(edges_src, edges_tgt, hist)
    D, H, W = edges_src.shape
    bz, by, bx = box[3] - box[0], box[4] - box[1], box[5] - box[2]
    ws = workspace(lib().msseg_hd_directed_workspace_bytes(bz, by, bx), edges_src.device)
    b6 = (C.c_int * 6)(*[int(v) for v in box])
    _ck(lib().msseg_hd_directed_hist(_p(edges_src), _p(edges_tgt), int(cls), D, H, W, b6, _p(ws), ws.numel(), _p(hist),
                                     hist.numel(), _stream()), "hd_directed_hist")
    return hist


def aug_crop_batch(img, lab, table, out_img, out_lab, roi):
    """img fp32 [C, D, H, W]; lab uint8 [D, H, W] or None; table: uint8 device tensor of npatch msseg_aug_row structs"""
    _need_gpu(img, table, out_img)
    assert img.dtype == torch.float32 and img.is_contiguous() and out_img.is_contiguous()
    npatch = out_img.shape[0]
    assert table.numel() >= npatch * 32
    _ck(lib().msseg_aug_crop_batch(_p(img), _p(lab), img.shape[0], img.shape[1], img.shape[2], img.shape[3], _p(table),
                                   npatch, _p(out_img), dt(out_img), _p(out_lab), roi, _stream()), "aug_crop_batch")
    return out_img, out_lab


# --------------------------------------------------------------------------------------------
# Swin pieces
# --------------------------------------------------------------------------------------------
def conv3d_k3s2(x, wp, bias, y, cin, cout):
    _need_gpu(x, wp, y)
    N, D, H, W = x.shape[:4]
    _ck(lib().msseg_conv3d_k3s2_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(y), ld(y), N, D, H, W, cin, cout, dt(x),
                                    _stream()), "conv3d_k3s2_fwd")
    return y


def box_copy(src, dst):
    """dst[n, d, h, w, :] = src[n, d, h, w, :] inside the common box, zero elsewhere in dst (channels-last [N, D, H, W, C]): zero
    padding at the high end of the spatial axes, or the crop back -- each is the other's adjoint (csrc/layout.hip)"""
    _need_gpu(src, dst)
    N, SD, SH, SW, Cc = src.shape
    _, DD, DH, DW, Cd = dst.shape
    if Cc != Cd or dst.shape[0] != N or src.dtype != dst.dtype:
        raise ValueError("box_copy: batch, channel count and dtype must agree")
    _ck(lib().msseg_box_copy(_p(src), ld(src), SD, SH, SW, _p(dst), ld(dst), DD, DH, DW, N, Cc, dt(src), _stream()), "box_copy")
    return dst


def merge_subs(offsets) -> int:
    """eight (a, b, c) sub-grid offsets along (D, H, W) -> the packed `subs` word of merge_gather"""
    assert len(offsets) == 8
    word = 0
    for s, (a, b, c) in enumerate(offsets):
        word |= (a | (b << 1) | (c << 2)) << (3 * s)
    return word


def merge_gather(x, out, subs):
    """out[n, i, j, k, s*C + c] = x[n, 2i + a_s, 2j + b_s, 2k + c_s, c] (zero beyond the grid): PatchMerging's eight strided
    slices + concat in one pass"""
    _need_gpu(x, out)
    N, D, H, W, Cc = x.shape
    _ck(lib().msseg_merge_gather_fwd(_p(x), ld(x), D, H, W, _p(out), ld(out), N, Cc, subs, dt(x), _stream()), "merge_gather_fwd")
    return out


def merge_gather_bwd(dy, dx, subs):
    """adjoint of merge_gather: dx [N, D, H, W, C] from dy [N, ceil(D/2), ceil(H/2), ceil(W/2), 8C]"""
    _need_gpu(dy, dx)
    N, D, H, W, Cc = dx.shape
    _ck(lib().msseg_merge_gather_bwd(_p(dy), ld(dy), _p(dx), ld(dx), N, D, H, W, Cc, subs, dt(dx), _stream()), "merge_gather_bwd")
    return dx


def zero_stuff2(dy, out):
    """dy [N,OD,OH,OW,C] -> out [N,ID,IH,IW,C] with out[:, ::2, ::2, ::2] = dy and zeros elsewhere."""
    _need_gpu(dy, out)
    N, OD, OH, OW, Cc = dy.shape
    ID, IH, IW = out.shape[1:4]
    _ck(lib().msseg_zero_stuff2(_p(dy), ld(dy), _p(out), ld(out), N, OD, OH, OW, ID, IH, IW, Cc, dt(dy), _stream()),
        "zero_stuff2")
    return out


def window_attention_fwd(qkv, qkv_bias, table, out, heads, ws, shift, bias_ws=None):
    """qkv [B,S,H,W,3C] contiguous -> out [B,S,H,W,C]; returns lse for the backward.  bias_ws: window edge the bias table
    was built for (MONAI SwinUNETR: 7 even where the window is clamped to a smaller grid)."""
    _need_gpu(qkv, table, out)
    assert qkv.is_contiguous() and out.is_contiguous()
    B, S, H, W, C3 = qkv.shape
    Cc = C3 // 3
    nW = -(-S // ws) * -(-H // ws) * -(-W // ws)
    lse = torch.empty(B * nW, heads, ws ** 3, dtype=torch.float32, device=qkv.device)
    _ck(lib().msseg_window_attention_fwd2(_p(qkv), _p(qkv_bias), _p(table), _p(out), _p(lse), B, S, H, W, Cc, heads, ws,
                                          shift, bias_ws or ws, dt(qkv), _stream()), "window_attention_fwd")
    return lse


def window_attention_bwd(qkv, qkv_bias, table, out, lse, dout, dqkv, dtable, heads, ws, shift, bias_ws=None):
    _need_gpu(qkv, table, out, lse, dout, dqkv)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    B, S, H, W, C3 = qkv.shape
    wsb = 0
    bws = bias_ws or ws
    if dtable is not None:
        wsb = int(lib().msseg_window_attention_bwd_workspace_bytes(B, S, H, W, C3 // 3, heads, ws, shift, dt(qkv)))
    work = torch.empty(wsb, dtype=torch.uint8, device=qkv.device) if wsb else None
    _ck(lib().msseg_window_attention_bwd2(_p(qkv), _p(qkv_bias), _p(table), _p(out), _p(lse), _p(dout), _p(dqkv),
                                          _p(dtable), B, S, H, W, C3 // 3, heads, ws, shift, bws, dt(qkv), _p(work), wsb,
                                          _stream()), "window_attention_bwd")
    return dqkv


def layernorm_fwd(x, gamma, beta, y, eps=1e-5, save=True):
    _need_gpu(x, y)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    _ck(lib().msseg_layernorm_fwd(_p(x), ld(x), _p(gamma), _p(beta), _p(y), ld(y), _p(mean), _p(rstd), rows, Cc, eps,
                                  dt(x), _stream()), "layernorm_fwd")
    return mean, rstd


def layernorm_bwd_add_ok(x) -> bool:
    """does layernorm_bwd(..., add=) take rows shaped like x's (the vector kernel: 16-byte chunks, at most 4096 channels)?"""
    epc = 16 // x.element_size()
    return x.shape[-1] % epc == 0 and x.shape[-1] <= 512 * epc and x.data_ptr() % 16 == 0


def layernorm_bwd(x, gamma, mean, rstd, dy, dx, dgamma=None, dbeta=None, accumulate=False, add=None):
    """add (optional, shaped like x): dx = layernorm backward + add in the same pass (layernorm_bwd_add_ok(x))"""
    _need_gpu(x, dy, dx)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if add is not None:
        _ck(lib().msseg_layernorm_bwd_add(_p(x), ld(x), _p(gamma), _p(mean), _p(rstd), _p(dy), ld(dy), _p(add), ld(add), _p(dx),
                                          ld(dx), rows, Cc, dt(x), _stream()), "layernorm_bwd_add")
    else:
        _ck(lib().msseg_layernorm_bwd(_p(x), ld(x), _p(gamma), _p(mean), _p(rstd), _p(dy), ld(dy), _p(dx), ld(dx),
                                      rows, Cc, dt(x), _stream()), "layernorm_bwd")
    if dgamma is not None:
        sc = scratch(x.device)
        _ck(lib().msseg_layernorm_param_grad(_p(x), ld(x), _p(mean), _p(rstd), _p(dy), ld(dy), _p(dgamma), _p(dbeta),
                                             int(accumulate),
                                             rows, Cc, _p(sc), sc.numel(), dt(x), _stream()), "layernorm_param_grad")
    return dx


def gelu_fwd(x, y):
    _need_gpu(x, y)
    assert x.is_contiguous() and y.is_contiguous()
    _ck(lib().msseg_gelu_fwd(_p(x), _p(y), x.numel(), dt(x), _stream()), "gelu_fwd")
    return y


def linear_gelu_ok(x, cin, cout) -> bool:
    """fc1 + GELU (and fc2's input gradient + GELU backward) of a cin -> cout MLP on x's tokens run as one launch each"""
    return bool(x.is_cuda and x.dtype == torch.bfloat16 and
                lib().msseg_linear_gelu_ok(x.numel() // x.shape[-1], cin, cout, BF16))


def linear_gelu_fwd(x, wp, bias, pre, act, cin, cout):
    """pre = x W^T + b, act = gelu(pre) in one launch (bit-identical to conv3d_k1 + gelu_fwd)"""
    _need_gpu(x, wp, pre, act)
    nv = x.numel() // x.shape[-1]
    esz = x.element_size()
    TIMER.launch("linear_gelu_fwd", 2.0 * nv * cin * cout, nv * (cin + 2 * cout) * esz + cin * cout * esz,
                 lambda: _ck(lib().msseg_linear_gelu_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(pre), ld(pre), _p(act), ld(act), nv,
                                                         cin, cout, dt(x), _stream()), "linear_gelu_fwd"))
    return pre, act


def linear_gelu_bwd(dy, wp, pre, dpre, cin, cout):
    """dpre = (dy W) * gelu'(pre): cin = channels of dy, cout = channels of pre / dpre (bit-identical to conv3d_k1 + gelu_bwd)"""
    _need_gpu(dy, wp, pre, dpre)
    nv = dy.numel() // dy.shape[-1]
    esz = dy.element_size()
    TIMER.launch("linear_gelu_bwd", 2.0 * nv * cin * cout, nv * (cin + 2 * cout) * esz + cin * cout * esz,
                 lambda: _ck(lib().msseg_linear_gelu_bwd(_p(dy), ld(dy), _p(wp), _p(pre), ld(pre), _p(dpre), ld(dpre), nv, cin,
                                                         cout, dt(dy), _stream()), "linear_gelu_bwd"))
    return dpre


def linear_add_ok(x, res, cin, cout) -> bool:
    """can y = res + Linear(x) run as one launch (bf16, shapes of the register-resident-weight kernel, aligned operands)?"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and res.dtype == x.dtype and x.data_ptr() % 16 == 0 and res.data_ptr() % 16 == 0):
        return False
    return bool(lib().msseg_linear_add_ok(x.numel() // x.shape[-1], cin, cout, BF16))


def linear_add(x, wp, bias, res, y, cin, cout):
    """y = res + (x W^T + b) (the bf16-rounded Linear output added to res: bit-identical to conv3d_k1 followed by an add pass)"""
    _need_gpu(x, wp, res, y)
    nv = x.numel() // x.shape[-1]
    esz = x.element_size()
    TIMER.launch("linear_add_fwd", 2.0 * nv * cin * cout, nv * (cin + 2 * cout) * esz + cin * cout * esz,
                 lambda: _ck(lib().msseg_linear_add_fwd(_p(x), ld(x), _p(wp), _p(bias), _p(res), ld(res), _p(y), ld(y), nv, cin, cout,
                                                        dt(x), _stream()), "linear_add_fwd"))
    return y


def gelu_bwd(x, dy, dx):
    _need_gpu(x, dy, dx)
    assert x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
    _ck(lib().msseg_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), dt(x), _stream()), "gelu_bwd")
    return dx
