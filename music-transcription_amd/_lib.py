"""ctypes binding of libmt_hip.so (the C ABI declared in include/mt_hip.h).

The product path has no CPU or torch fallback: if the shared library is missing,
importing this module raises, and every wrapper raises MtError on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmt_hip.so")
MAX_LSTM_LAYERS = 8
N_PITCH = 88
DT_BF16, DT_F16 = 0, 1          # MT_DT_* (include/mt_hip.h): 16-bit operand type of the GEMM / conv kernels
GX_F16 = 0x10                   # MT_GX_F16: gate pre-activations travel GEMM -> recurrence as f16


class MtError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libmt_hip.so (hipcc cross-compiles without a GPU)."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise MtError("building libmt_hip.so failed")
    return LIB_PATH


if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc)")

# torch ships its own libamdhip64; load it first so that libmt_hip.so binds to the SAME HIP runtime instance
# (two runtimes in one process do not share devices, streams or allocations).
import torch as _torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

vp, sz, i32, f32p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p


class MelDesc(C.Structure):
    """mt_mel_desc (include/mt_hip.h)."""
    _fields_ = [("sr", i32), ("hop", i32), ("n_mels", i32), ("ell_rows", i32)]


class CnnRnnWeights(C.Structure):
    """mt_cnnrnn_weights (include/mt_hip.h)."""
    _fields_ = [("n_mels", i32), ("hidden", i32), ("layers", i32), ("operand_dtype", i32),
                ("conv1_w", vp), ("conv1_b", vp), ("conv2_w", vp), ("conv2_b", vp),
                ("w_ih", vp * MAX_LSTM_LAYERS), ("b_gates", vp * MAX_LSTM_LAYERS), ("w_hh", vp * MAX_LSTM_LAYERS),
                ("fc_w", vp), ("fc_b", vp), ("w_ihx", vp * MAX_LSTM_LAYERS)]


class CnnRnnLargeWeights(C.Structure):
    """mt_cnnrnn_large_weights (include/mt_hip.h)."""
    _fields_ = ([(n, i32) for n in ("n_mels", "hidden", "layers", "hidden_local", "use_attention", "use_heads", "heads", "head_dim_pad")]
                + [("attn_scale", C.c_float), ("operand_dtype", i32)]
                + [(n, vp) for n in ("conv1_w", "conv1_b", "rb1c1_w", "rb1c1_b", "rb1c2_w", "rb1c2_b", "rb2c1_w", "rb2c1_b",
                                     "rb2c2_w", "rb2c2_b", "fa_w", "fa_b")]
                + [("main_w_ih", vp * MAX_LSTM_LAYERS), ("main_b", vp * MAX_LSTM_LAYERS), ("main_w_hh", vp * MAX_LSTM_LAYERS)]
                + [(n, vp) for n in ("local_w_ih", "local_b", "local_w_hh", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln_g", "ln_b",
                                     "shared_w", "shared_b", "heads_w", "heads_b", "fc_w", "fc_b")]
                + [("main_w_ihx", vp * MAX_LSTM_LAYERS)])


ll = C.c_longlong
_SIGS = {
    "mt_version": (i32, []),
    "mt_last_error": (C.c_char_p, []),
    "mt_device_count": (i32, []),
    "mt_init": (i32, [i32]),
    "mt_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "mt_allreduce": (i32, [vp, sz, i32, vp, vp]),
    "mt_mel_filterbank_host": (i32, [vp, i32, i32]),
    "mt_mel_num_frames": (i32, [i32, i32]),
    "mt_mel_plan_bytes": (sz, [i32]),
    "mt_mel_plan_init": (i32, [vp, sz, i32, i32, i32, C.POINTER(MelDesc), vp]),
    "mt_mel_db_f32": (i32, [vp, C.POINTER(MelDesc), vp, i32, i32, vp, vp, i32, vp]),
    "mt_conv1_bn_relu_pool": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mt_conv2_bn_relu_pool": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mt_conv1_bn_relu_pool_dt": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mt_conv2_bn_relu_pool_dt": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_cnnrnn_conv_fused": (i32, []),
    "mt_conv12_bn_relu_pool_dt": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_f32acc_dt": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_lstm_gx_dt": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_logits_dt": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_conv_cl_dt": (i32, [vp, vp, vp, vp, vp] + [i32] * 12 + [vp]),
    "mt_gemm_batched_f32_dt": (i32, [vp, i32, ll, ll, vp, i32, ll, ll, vp, vp, i32, ll, ll, i32, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_batched_h16out_dt": (i32, [vp, i32, ll, ll, vp, i32, ll, ll, vp, vp, i32, ll, ll, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mt_lstm_relayout_dt": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mt_attn_softmax_clamped_dt": (i32, [vp, i32, vp, i32, i32, ll, C.c_float, C.c_float, i32, vp]),
    "mt_layernorm_residual_dt": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, ll, i32, C.c_float, i32, vp]),
    "mt_gemm_lstm_gx_from_hx": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "mt_gemm_lstm_gx_from_hx_ex": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_logits_from_hx": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "mt_gemm_bf16_f32acc": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "mt_gemm_lstm_gx": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "mt_gemm_sched_bytes": (sz, []),
    "mt_gemm_lstm_gx_sched": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "mt_gemm_lstm_gx_from_hx_sched": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "mt_gemm_lstm_dh": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_gemm_logits": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "mt_lstm_gx_bytes": (sz, [i32, i32, i32]),
    "mt_lstm_hx_bytes": (sz, [i32, i32, i32]),
    "mt_lstm_sync_bytes": (sz, [i32, i32]),
    "mt_lstm_bidir_fwd": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "mt_lstm_bidir_fwd_xproj": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "mt_lstm_bidir_fwd_ex": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "mt_persistent_cus_in_flight": (i32, [vp]),
    "mt_lstm_relayout_bf16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "mt_lstm_unpack_f32": (i32, [vp, vp, i32, i32, i32, vp]),
    "mt_conv_cl_bf16": (i32, [vp, vp, vp, vp, vp] + [i32] * 11 + [vp]),
    "mt_gemm_batched_f32": (i32, [vp, i32, ll, ll, vp, i32, ll, ll, vp, vp, i32, ll, ll, i32, i32, i32, i32, i32, vp]),
    "mt_gemm_batched_bf16out": (i32, [vp, i32, ll, ll, vp, i32, ll, ll, vp, vp, i32, ll, ll, i32, i32, i32, i32, i32, i32, vp]),
    "mt_lstm_relayout_ex": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mt_attn_softmax_clamped": (i32, [vp, i32, vp, i32, i32, ll, C.c_float, C.c_float, vp]),
    "mt_attn_transpose_v": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "mt_attn_fused_clamped": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.c_float, C.c_float, vp, i32, i32, vp]),
    "mt_layernorm_residual": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, ll, i32, C.c_float, vp]),
    "mt_cnnrnn_large_workspace_bytes": (sz, [C.POINTER(CnnRnnLargeWeights), i32, i32]),
    "mt_cnnrnn_large_status_offset": (sz, [C.POINTER(CnnRnnLargeWeights), i32, i32, i32]),
    "mt_cnnrnn_large_forward": (i32, [C.POINTER(CnnRnnLargeWeights), vp, vp, i32, i32, vp, vp, sz, vp]),
    "mt_cnnrnn_large_forward_ex": (i32, [C.POINTER(CnnRnnLargeWeights), vp, vp, i32, i32, vp, vp, sz, vp, vp, vp, vp]),
    "mt_cnnrnn_large_num_stages": (i32, [i32]),
    "mt_cnnrnn_large_forward_ev": (i32, [C.POINTER(CnnRnnLargeWeights), vp, vp, i32, i32, vp, vp, sz, C.POINTER(vp), i32, vp]),
    "mt_adam_workspace_bytes": (sz, []),
    "mt_adam_clip_step": (i32, [vp, vp, vp, vp, ll] + [C.c_float] * 6 + [i32, vp, vp, sz, vp]),
    "mt_adam_clip_step_ex": (i32, [vp, vp, vp, vp, ll] + [C.c_float] * 6 + [i32, C.c_float, vp, i32, vp, vp, sz, vp]),
    "mt_bce_workspace_bytes": (sz, []),
    "mt_bce_masked_fwd_bwd": (i32, [vp, vp, vp, C.c_longlong, C.c_float, i32, vp, vp, vp, sz, i32, i32, i32, vp]),
    "mt_onset_offset_targets": (i32, [vp, vp, vp, C.c_longlong, i32, vp]),
    "mt_predict_threshold": (i32, [vp, vp, C.c_longlong, C.c_float, vp]),
    "mt_f1_counts": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "mt_f1_sweep_counts": (i32, [vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "mt_roll_to_notes": (i32, [vp, i32, C.c_float, i32, i32, i32, vp, vp, vp, i32, vp]),
    "mt_conv1_stats": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "mt_bn_finalize": (i32, [vp, C.c_double, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, i32, vp, vp, vp, vp, i32, vp]),
    "mt_bn_stats_cl": (i32, [vp, ll, i32, vp, vp]),
    "mt_bn_relu_pool_apply": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mt_bn_pool_bwd": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mt_conv_cl_tie": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mt_bn_pool_bwd_tie": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mt_conv2_wgrad_workgroups": (i32, []),
    "mt_conv2_wgrad": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mt_pack_wih_cf": (i32, [vp, vp, ll, i32, i32, i32, i32, i32, i32, vp]),
    "mt_pack_jobs": (i32, [vp, i32, i32, vp]),
    "mt_transpose_bf16": (i32, [vp, ll, ll, i32, vp, ll, i32, vp]),
    "mt_gather4_f32": (i32, [vp, vp, i32, i32, i32, i32, ll, ll, ll, ll, C.c_float, vp]),
    "mt_sum_slices_f32": (i32, [vp, ll, i32, i32, vp, i32, i32, i32, vp]),
    "mt_rowsum_bf16": (i32, [vp, ll, ll, vp, i32, vp]),
    "mt_conv1_bwd": (i32, [vp] * 8 + [i32] + [vp] * 5 + [i32, i32, i32, vp]),
    "mt_lstm_cx_bytes": (sz, [i32, i32, i32]),
    "mt_lstm_bidir_fwd_train": (i32, [vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "mt_lstm_relayout_train": (i32, [vp, vp, i32, i32, i32, i32, i32, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_lstm_dh_relayout": (i32, [vp, i32, vp, i32, i32, i32, i32, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_lstm_dgx_bytes": (sz, [i32, i32, i32]),
    "mt_lstm_bwd_part_bytes": (sz, [i32, i32, i32]),
    "mt_lstm_bidir_bwd": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, sz, i32, i32, i32, vp]),
    "mt_lstm_bidir_bwd_ex": (i32, [vp, vp, vp, vp, vp, vp, sz, vp, sz, i32, i32, i32, i32, vp]),
    "mt_lstm_bwd_poison": (i32, [vp, sz, i32, i32, i32, vp]),
    "mt_lstm_dg_unpack": (i32, [vp, vp, i32, vp, ll, i32, i32, i32, vp]),
    "mt_lstm_hprev_t": (i32, [vp, vp, ll, i32, i32, i32, i32, vp]),
    "mt_dlogits_pack": (i32, [vp, vp, vp, ll, i32, i32, i32, vp]),
    "mt_dropout2d_mask": (i32, [vp, i32, i32, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_bn_act_fwd": (i32, [vp] * 12 + [i32] * 8 + [vp]),
    "mt_bn_act_bwd": (i32, [vp, i32, vp, i32] + [vp] * 13 + [i32, vp, vp, i32, vp] + [vp] * 4 + [i32] * 6 + [vp]),
    "mt_conv_wgrad_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "mt_conv_wgrad": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
    "mt_conv_cl_ex": (i32, [vp, i32, vp, i32, vp, vp, vp] + [i32] * 13 + [vp]),
    "mt_transpose_bf16_batched": (i32, [vp, ll, ll, i32, i32, vp, ll, ll, i32, i32, vp]),
    "mt_attn_softmax_train": (i32, [vp, i32, vp, i32, i32, ll, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_attn_clamped_bwd": (i32, [vp, i32, vp, i32, vp, i32, i32, ll, C.c_float, C.c_float, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_layernorm_residual_train": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, vp, ll, i32, C.c_float, vp]),
    "mt_layernorm_residual_bwd_slices": (i32, []),
    "mt_layernorm_residual_bwd": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, vp, i32, vp, ll, i32, vp]),
    "mt_heads_relu_dropout_bwd": (i32, [vp, i32, vp, i32, vp, i32, ll, i32, C.c_float, vp]),
    "mt_dlogits_pack_heads": (i32, [vp, vp, i32, vp, ll, i32, i32, i32, i32, vp]),
    "mt_f32_to_bf16_rows": (i32, [vp, i32, vp, i32, ll, i32, C.c_float, vp]),
    "mt_dropout_bf16_rows": (i32, [vp, i32, ll, i32, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_dropout_f32": (i32, [vp, ll, C.c_float, C.c_uint, C.c_uint, vp]),
    "mt_axpby_rows_f32": (i32, [vp, i32, vp, i32, vp, i32, ll, i32, C.c_float, C.c_float, vp]),
    "mt_resample_poly": (i32, [vp, ll, i32, i32, vp, i32, i32, i32, ll, vp, ll, vp]),
    "mt_resample_polyphase": (i32, [vp, ll, i32, i32, vp, i32, i32, i32, ll, vp, ll, vp]),
    "mt_cnnrnn_workspace_bytes": (sz, [C.POINTER(CnnRnnWeights), i32, i32]),
    "mt_cnnrnn_status_offset": (sz, [C.POINTER(CnnRnnWeights), i32, i32, i32]),
    "mt_cnnrnn_forward": (i32, [C.POINTER(CnnRnnWeights), vp, vp, i32, i32, vp, vp, sz, vp]),
    "mt_cnnrnn_num_stages": (i32, [i32]),
    "mt_cnnrnn_forward_ex": (i32, [C.POINTER(CnnRnnWeights), vp, vp, i32, i32, vp, vp, sz, C.POINTER(vp), i32, vp]),
}
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library disagree
    _fn.restype, _fn.argtypes = _res, _args


def last_error() -> str:
    return (lib.mt_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise MtError(f"{what or 'libmt_hip'} failed (code {rc}): {last_error()}")


def ptr(t) -> int:
    """Device (or host) address of a contiguous torch tensor, or NULL for None."""
    if t is None:
        return None
    assert t.is_contiguous(), "libmt_hip takes contiguous buffers"
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
