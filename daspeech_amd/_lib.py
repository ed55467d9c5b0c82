"""ctypes binding of the C ABI in include/daspeech_dag.h (libdaspeech_hip.so).

The product path has NO CPU fallback: if the shared object is missing or a call fails this raises.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "lib", "libdaspeech_hip.so")

ABI_VERSION = 2
DTYPE_CODES = {"torch.float32": 0, "torch.float16": 1, "torch.bfloat16": 2}

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_p = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes).  Must list every symbol declared in include/daspeech_dag.h (tests check this).
SIGNATURES = {
    "dsp_abi_version": (_c_int, []),
    "dsp_last_error": (ctypes.c_char_p, []),
    "dsp_logsoftmax_gather": (_c_int, [_c_p, _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64,
                                       _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_logsoftmax_gather_bwd": (_c_int, [_c_p, _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64,
                                           _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_logsoftmax_gather_stats": (_c_int, [_c_p, _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64,
                                             _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_logsoftmax_gather_bwd_lazy": (_c_int, [_c_p, _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64,
                                                _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dsp_dag_loss_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                  _c_p, _c_sz, _c_p]),
    "dsp_dag_loss_bwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                  _c_p, _c_sz, _c_p]),
    "dsp_dag_loss_fwd_ld": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int,
                                     _c_p, _c_sz, _c_p]),
    "dsp_dag_loss_bwd_ld": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int,
                                     _c_p, _c_sz, _c_p]),
    "dsp_dag_best_alignment_ld": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                           _c_p, _c_sz, _c_p]),
    "dsp_dag_loss_fwd_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_loss_bwd_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_best_alignment_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_pitch_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "dsp_dag_best_alignment": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_alignment_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dsp_dag_best_alignment_ws": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_sz, _c_p]),
    "dsp_dag_max_alpha": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_backtrace": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_max_alpha_blocks_supported": (_c_int, [_c_int, _c_int]),
    "dsp_dag_max_alpha_blocks": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_dag_backtrace_blocks": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    # include/daspeech_decode.h
    "dsp_argmax_logp": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_lookahead_next": (_c_int, [_c_p, _c_p, ctypes.c_float, _c_int, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_follow_path": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_viterbi_collect": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_gather_rows": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_extract_links": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p]),
    "dsp_extract_links_train": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p]),
    "dsp_extract_links_bwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       ctypes.c_float, _c_p]),
    "dsp_extract_links_workspace": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_extract_links_ws": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p, _c_sz, _c_p]),
    "dsp_extract_links_bwd_ws": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          ctypes.c_float, _c_p, _c_sz, _c_p]),
    "dsp_extract_links_debug_ran": (ctypes.c_uint, []),
    "dsp_extract_links_debug_range": (ctypes.c_uint, []),
    "dsp_posterior": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_posterior_features": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_posterior_features_bwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_durations": (_c_int, [_c_p, _c_p, ctypes.c_float, _c_p, _c_i64, _c_p]),
    "dsp_bucketize_embed_add": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_p, _c_i64, _c_int, _c_p]),
    "dsp_length_regulator_lens": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p]),
    "dsp_length_regulator_expand": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    # include/daspeech_hifigan.h
    "dsp_conv1d_split_packed_elems": (ctypes.c_long, [_c_int, _c_int, _c_int]),
    "dsp_conv1d_split_pack": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_conv1d_split": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_conv1d_split_residual": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, ctypes.c_float, _c_p, ctypes.c_long, _c_int, _c_int,
                                           _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_conv1d_split_ragged": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, ctypes.c_float, _c_p, ctypes.c_long, _c_int, _c_int,
                                         _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_int, _c_p]),
    "dsp_relpos_attention": (_c_int, [_c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_attention_split": (_c_int, [_c_p, ctypes.c_long, _c_p, ctypes.c_long, _c_p, ctypes.c_long, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                                     _c_p, _c_int, _c_p]),
    "dsp_conv1d_split_ksplit_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "dsp_conv1d_split_ksplit": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, ctypes.c_float, _c_p, ctypes.c_long, _c_int, _c_int,
                                         _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_sz, _c_p, _c_int, _c_p]),
    "dsp_ffn_split_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int, _c_int]),
    "dsp_ffn_split": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, ctypes.c_float, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, ctypes.c_float,
                               _c_p, ctypes.c_long, _c_p, _c_sz, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, ctypes.c_float, _c_p, _c_p]),
    "dsp_linear_ln_split": (_c_int, [_c_p, ctypes.c_long, _c_p, _c_p, ctypes.c_float, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, ctypes.c_float, _c_p, ctypes.c_long,
                                     _c_int, _c_int, _c_int, _c_int, _c_p, _c_int, _c_p]),
    "dsp_layer_norm": (_c_int, [_c_p, _c_p, _c_p, ctypes.c_float, _c_p, ctypes.c_long, _c_int, _c_p]),
    "dsp_dwconv_bn_silu": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_float, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_conv": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.c_float, ctypes.c_float, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_conv_chain": (_c_int, [_c_p, _c_int, _c_int, _c_p]),
    "dsp_hifigan_conv_chain_lens": (_c_int, [_c_p, _c_int, _c_int, _c_p, _c_int, _c_p]),
    "dsp_hifigan_post_lens": (_c_int, [_c_p, _c_p, ctypes.c_float, _c_p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p, _c_int, _c_p]),
    "dsp_hifigan_resunit": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                                     ctypes.c_float, _c_int, _c_p]),
    "dsp_hifigan_resunit_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "dsp_hifigan_packed_weight_elems": (ctypes.c_long, [_c_int, _c_int, _c_int]),
    "dsp_hifigan_pack_weights": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_pack_input": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_post": (_c_int, [_c_p, _c_p, ctypes.c_float, _c_p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p]),
    "dsp_hifigan_pack_weights_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_resunit_f32_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "dsp_hifigan_conv_chain_f32": (_c_int, [_c_p, _c_int, _c_int, _c_p, _c_int, _c_p]),
    "dsp_hifigan_pad_input_f32": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "dsp_hifigan_post_f32": (_c_int, [_c_p, _c_p, ctypes.c_float, _c_p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_p, _c_int, _c_p]),
    "dsp_dag_alignment_trace_optional": (_c_int, [_c_int, _c_int]),
    "dsp_dag_debug_k5": (_c_int, [ctypes.POINTER(ctypes.c_uint)]),
    "dsp_dag_set_option": (_c_int, [ctypes.c_char_p, _c_int]),
    "dsp_dag_last_launch_status": (_c_int, [_c_p, ctypes.POINTER(ctypes.c_uint)]),
    "dsp_dag_last_fallback_count": (ctypes.c_uint, []),
    "dsp_dag_debug_words": (ctypes.POINTER(ctypes.c_uint), []),
}

_lib = None


class DaspeechHipError(RuntimeError):
    pass


def load():
    """Load libdaspeech_hip.so (once).  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one this process uses (loading /opt/rocm's libamdhip64 before
    # torch leaves two runtimes in the process and launches then fail with "no ROCm-capable device")
    import torch  # noqa: F401
    if not os.path.exists(SO_PATH):
        raise DaspeechHipError(
            f"{SO_PATH} is missing: build it with `python -m daspeech_amd.build` (hipcc, gfx950). "
            "daspeech_amd has no CPU fallback for its HIP ops.")
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.dsp_abi_version() != ABI_VERSION:
        raise DaspeechHipError(f"ABI version mismatch: library {lib.dsp_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dsp_last_error().decode("utf-8", "replace")
        raise DaspeechHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream_handle():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_option(name: str, value: int):
    check(load().dsp_dag_set_option(name.encode(), int(value)), "dsp_dag_set_option")


def last_launch_status() -> int:
    """Status word of the last fast-path DP launch on the current stream (synchronises it)."""
    w = ctypes.c_uint(0)
    check(load().dsp_dag_last_launch_status(current_stream_handle(), ctypes.byref(w)), "dsp_dag_last_launch_status")
    return int(w.value)


def last_fallback_count() -> int:
    """Exact-fallback cell count of the launch inspected by the latest last_launch_status() call."""
    return int(load().dsp_dag_last_fallback_count())


def last_dense_gave_up() -> bool:
    """True when the launch inspected by last_launch_status() was a dense-window matrix-core DP that gave up on its batch (exact-redo
    budget spent) and left the result to its stand-by log-space kernels (dag_dp_dense_mfma.hip, `aborted`)."""
    return bool(load().dsp_dag_debug_words()[2])


def debug_fallback_cells():
    """(sample, row, column, S) of the first fallback cells of the launch inspected by last_launch_status()."""
    import struct
    w = load().dsp_dag_debug_words()
    out = []
    for i in range(min(14, int(w[1]))):
        b, t, j, sb = w[7 + 4 * i], w[8 + 4 * i], w[9 + 4 * i], w[10 + 4 * i]
        out.append((int(b), int(t), int(j), struct.unpack("f", struct.pack("I", sb))[0]))
    return out
