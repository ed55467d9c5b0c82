/* daspeech_dag.h — C ABI of the MI355X-native DASpeech hot path (libdaspeech_hip.so).
 *
 * Drop-in boundary.  These entry points replace, one for one, the four functions the reference binds
 * through pybind11 in DASpeech/custom_ops/dag_loss.cpp:19-29 (module `dag_loss_fn`):
 *     dag_loss / dag_loss_backward / dag_best_alignment / logsoftmax_gather
 * plus the decode / TTS-glue steps the reference runs as Python loops (cited per function).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless the name says host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued on it,
 *     nothing synchronises the host (the reference used legacy stream 0 + private streams, dag_loss.cu:334,355);
 *   - inputs are borrowed and must be contiguous in the stated layout; outputs are caller-allocated
 *     (the reference's callee allocated with at::zeros, dag_loss.cu:339-340 — here the kernels fill every
 *     element themselves, so outputs need no pre-initialisation);
 *   - return value: 0 = success, <0 = DSP_E* argument error (nothing launched), >0 = hipError_t of a failed
 *     launch.  dsp_last_error() gives the message (thread-local).
 *   - dtype codes for logits: 0 = fp32, 1 = fp16, 2 = bf16.
 */
#ifndef DASPEECH_DAG_H
#define DASPEECH_DAG_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_ABI_VERSION 2      /* 2 (r06): the *_ld entry points (row pitches); every ABI-1 symbol is unchanged */

#define DSP_OK 0
#define DSP_EINVAL (-1)   /* bad size / null pointer / unsupported dtype */
#define DSP_ENOSPC (-2)   /* workspace too small */

#define DSP_F32 0
#define DSP_F16 1
#define DSP_BF16 2

typedef void* dsp_stream_t;

int dsp_abi_version(void);
const char* dsp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * K1  logsoftmax + gather            replaces `logsoftmax_gather` (dag_loss.cpp:28; logsoftmax_gather.cu:313-377)
 *   logits  [B,L,V] contiguous, dtype per `dtype`; if write_softmax != 0 it is OVERWRITTEN with softmax(logits)
 *           in its own dtype (the reference's in-place contract, logsoftmax_gather.cu:296-307).
 *   idx     int64, addressed idx[b*idx_sb + j*idx_sj + s*idx_ss] (element strides) — the caller's
 *           targets.unsqueeze(1).expand(-1,L,-1) (nat_dag_loss.py:127) is passed as (T, 0, 1) without a copy.
 *   match   fp32, written at match[b*out_sb + j*out_sj + s*out_ss]; (S*L, 1, L) produces the [B,S,L]
 *           ("match_all") layout directly, (L*S, S, 1) the reference's [B,L,S].
 *   Indices outside [0,V) are an error the reference does not check either; here they are clamped. */
int dsp_logsoftmax_gather(void* logits, int dtype,
                          const int64_t* idx, int64_t idx_sb, int64_t idx_sj, int64_t idx_ss,
                          float* match, int64_t out_sb, int64_t out_sj, int64_t out_ss,
                          int B, int L, int V, int S, int write_softmax, dsp_stream_t stream);

/* K1 backward                         replaces the Python in DASpeech/custom_ops/dag_loss.py:293-295
 *   softmax_inout [B,L,V] holds softmax (left by K1) and is overwritten with d loss / d logits:
 *       gx = softmax * (-(sum_s g[b,j,s]));  gx[b,j,idx[b,j,s]] += g[b,j,s]   (duplicates accumulate)
 *   g fp32 addressed with element strides like `match` above. */
int dsp_logsoftmax_gather_bwd(void* softmax_inout, int dtype,
                              const int64_t* idx, int64_t idx_sb, int64_t idx_sj, int64_t idx_ss,
                              const float* g, int64_t g_sb, int64_t g_sj, int64_t g_ss,
                              int B, int L, int V, int S, dsp_stream_t stream);

/* K1 "lazy" pair: same match and the same gradient, but the logits are NOT overwritten by the forward pass.
 *   The reference stores the softmax in place purely as backward state and forbids every other use of the buffer
 *   ("DO NOT use word_ins_out after this function", dag_loss.py:249-251).  Here the forward writes two floats per row
 *   (row_stats[(b*L+j)*2] = max, [..+1] = 1/sum exp(x - max)) and the backward recomputes softmax = exp(x - max) * inv from
 *   the logits it overwrites with the gradient: the forward's B*L*V store disappears (4.3 of 8.9 GB at C2). */
int dsp_logsoftmax_gather_stats(const void* logits, int dtype,
                                const int64_t* idx, int64_t idx_sb, int64_t idx_sj, int64_t idx_ss,
                                float* match, int64_t out_sb, int64_t out_sj, int64_t out_ss,
                                float* row_stats, int B, int L, int V, int S, dsp_stream_t stream);
int dsp_logsoftmax_gather_bwd_lazy(void* logits_inout, int dtype,
                                   const int64_t* idx, int64_t idx_sb, int64_t idx_sj, int64_t idx_ss,
                                   const float* g, int64_t g_sb, int64_t g_sj, int64_t g_ss,
                                   const float* row_stats, int B, int L, int V, int S, dsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K2/K3  forward / backward DP       replaces `dag_loss` (dag_loss.cpp:25; dag_loss.cu:313-375)
 *   match [B,T,L] fp32, links [B,L,TR] fp32 (links[b,i,d] = log P(i -> i+d+1)), out_len/tgt_len int64 [B].
 *   alpha [B,T,L] fp32 out; beta [B,T,L] fp32 out or NULL (= the reference's require_gradient=false).
 *   loss  [B] fp32 out or NULL: beta[b,0,0] when beta != NULL else alpha[b,T_b-1,L_b-1] (dag_loss.py:107-110).
 *   Cells the recurrence never reaches are -inf.  Unreachable ends give loss = -inf (no device assert).
 *   workspace: dsp_dag_workspace_bytes(B,T,L,TR) bytes of device scratch owned by the CALLER (the reference allocates its scratch
 *   per call with ATen, dag_loss.cu:154,339-340).  It is zeroed on `stream` by the call itself and nothing about it outlives the call
 *   (the status words dsp_dag_last_launch_status reads are moved to a 256-byte library buffer on `stream` before the call returns: the
 *   workspace may be released, stream-ordered, at once) — so the launch (memset + kernels + that copy) can be captured in a hipGraph.
 *   workspace == NULL (or too small) selects a library-owned grow-only buffer per (device, stream) instead: not capturable.
 *   For dense windows (TR > 64) the size includes the stand-by log-space path's scratch (a B*L*TR*4-byte re-laid-out copy of links). */
size_t dsp_dag_workspace_bytes(int B, int T, int L, int TR);
int dsp_dag_loss_fwd(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                     float* alpha, float* beta, float* loss, int B, int T, int L, int TR,
                     void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* ... with ROW PITCHES (ABI 2).  ld_match / ld_ab = elements between consecutive target rows of match and of alpha / beta (batch stride =
 *   T * ld; dense tensors: ld = L, which is what dsp_dag_loss_fwd passes).  A graph length is floor(src_upsample * frames) — three graphs in
 *   four are not a multiple of 4 and their dense rows are not 16-byte aligned; a pitch rounded up to 4 keeps them aligned and the TR <= 32
 *   strip kernels (16-byte row loads) serve such a graph WITHOUT a padded copy of match / alpha / beta: columns L .. ld-1 are never read as
 *   data, alpha / beta come back -inf there.  dsp_logsoftmax_gather writes `match` with any pitch (its out_ss stride), so the gather can
 *   produce the pitched layout directly.  Only the TR <= 32 families take pitched rows (ld multiples of 4, >= L rounded up to 4, 16-byte
 *   aligned bases); every other window needs ld = L and the call fails with DSP_EINVAL otherwise.  Workspace: size it for L rounded up to 4.
 *   Reference contract this replaces: dag_loss.py:103-104 (`.contiguous()` on match_all / links before the CUDA call). */
int dsp_dag_loss_fwd_ld(const float* match, int ld_match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                        float* alpha, float* beta, int ld_ab, float* loss, int B, int T, int L, int TR,
                        void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* K4/K5  gradients                    replaces `dag_loss_backward` (dag_loss.cpp:26; dag_loss.cu:518-571)
 *   grad_out [B]; grad_match [B,T,L]; grad_links [B,L,TR]; formulas SURVEY.md §9.1 K4/K5. Either output may be NULL. */
int dsp_dag_loss_bwd(const float* grad_out, const float* alpha, const float* beta, const float* match,
                     const float* links, const int64_t* out_len, const int64_t* tgt_len,
                     float* grad_match, float* grad_links, int B, int T, int L, int TR,
                     void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* ... with row pitches (see dsp_dag_loss_fwd_ld): alpha / beta as the pitched forward left them (ld_ab), match with ld_match, grad_match
 *   written with ld_grad_match (columns past L inside the pitch may be written with unspecified values).  TR <= 32 only when ld_ab != L. */
int dsp_dag_loss_bwd_ld(const float* grad_out, const float* alpha, const float* beta, int ld_ab, const float* match, int ld_match,
                        const float* links, const int64_t* out_len, const int64_t* tgt_len,
                        float* grad_match, int ld_grad_match, float* grad_links, int B, int T, int L, int TR,
                        void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* K6/K7  Viterbi alignment            replaces `dag_best_alignment` (dag_loss.cpp:27; dag_best_alignment.cu:209-253)
 *   alpha_max [B,T,L] fp32 out, trace [B,T,L] int32 out (scratch the caller owns), path [B,L] int64 out
 *   (the reference returns int32 and casts in Python, dag_loss.py:228).  path[b,j] = t or -1.
 *   Tie rule: smallest predecessor index among equal maxima (torch.max rule, dag_loss.py:320).
 *   trace may be NULL where dsp_dag_alignment_trace_optional(L, TR) returns 1 (r06: every window — TR <= 32 on 16-byte rows, 33 .. 128 on the
 *   values-only strips of dag_dp_maxstripw.hip, dense windows; 0 remains for TR <= 32 with L off the 16-byte grid or L > 8192, and under kernel pins): the DP then keeps values only and the
 *   back-trace recomputes the arg-max of the T cells it visits (same tie rule) — no B*T*L int32 trace tensor is produced
 *   (for dense windows, TR > 64, the scratch of dsp_dag_alignment_workspace_bytes holds a 2-byte block index per cell that narrows
 *   that recomputation to 64 candidates).
 */
int dsp_dag_best_alignment(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                           float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                           dsp_stream_t stream);
/* The same with caller-owned scratch (dsp_dag_alignment_workspace_bytes bytes; semantics as for dsp_dag_loss_fwd's workspace). */
size_t dsp_dag_alignment_workspace_bytes(int B, int T, int L, int TR);
int dsp_dag_best_alignment_ws(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                              float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                              void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* ... with row pitches (see dsp_dag_loss_fwd_ld): served by the values-only strip DP + lazy back-trace (TR <= 32, L <= 8192, i.e. where
 *   dsp_dag_alignment_trace_optional(L rounded up to 4, TR) is 1); `trace` is ignored there.  path stays dense [B,L]. */
int dsp_dag_best_alignment_ld(const float* match, int ld_match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                              float* alpha_max, int ld_alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                              void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* 1 if the op (0: dsp_dag_loss_fwd_ld / _bwd_ld, 1: dsp_dag_best_alignment_ld) serves pitched rows for this graph under the calling thread's
 * kernel pin — else pass dense tensors (ld = L). */
int dsp_dag_pitch_supported(int op, int L, int TR);

/* ------------------------------------------------------------------------------------------------
 * The three DP operators in DOUBLE precision (r06; csrc/dag_dp_f64.hip) — the reference dispatches its kernels for double as well
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, dag_loss.cu:160,294,415,499, dag_best_alignment.cu:143,219).  Dense [B,T,L] / [B,L,TR] double tensors,
 * every intermediate a double, log space, one workgroup per (sample, direction): a correctness path (L <= 10240), not a fast one.  Same
 * semantics as the fp32 entry points; dsp_dag_best_alignment_f64 always needs its int32 trace [B,T,L]. */
int dsp_dag_loss_fwd_f64(const double* match, const double* links, const int64_t* out_len, const int64_t* tgt_len,
                         double* alpha, double* beta, double* loss, int B, int T, int L, int TR, dsp_stream_t stream);
int dsp_dag_loss_bwd_f64(const double* grad_out, const double* alpha, const double* beta, const double* match, const double* links,
                         const int64_t* out_len, const int64_t* tgt_len, double* grad_match, double* grad_links,
                         int B, int T, int L, int TR, dsp_stream_t stream);
int dsp_dag_best_alignment_f64(const double* match, const double* links, const int64_t* out_len, const int64_t* tgt_len,
                               double* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream);

/* The two halves of the alignment, separately — what the Viterbi graph decode needs
 * (s2s_conformer_dag_fastspeech2.py:244-304: max-product steps over the links, THEN the length is chosen, THEN the back-trace):
 *   dsp_dag_max_alpha   alpha_max[b,t,j] = match[b,t,j] + max_d(alpha_max[b,t-1,j-d] + links[b,j-d,d-1]) and its arg-max
 *                       trace[b,t,j] = j-d (smallest index among equal maxima, -1 if none) for EVERY cell j >= t of rows
 *                       t < tgt_len[b] — no pruning of cells that cannot reach (tgt_len-1, out_len-1), unlike the fused op;
 *   dsp_dag_backtrace   path[b,j] = t for the vertices on the chain trace[...] from (tgt_len[b]-1, out_len[b]-1), else -1;
 *                       tgt_len may differ from the one the DP ran with (any row it filled).
 */
int dsp_dag_max_alpha(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                      float* alpha_max, int32_t* trace, int B, int T, int L, int TR, dsp_stream_t stream);
int dsp_dag_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L,
                      dsp_stream_t stream);

/* The same two halves on the dense-window kernels (TR > 32 — the model's default window, --max-transition-length 99999): the max-DP as blocked
 * max-plus products (alpha_max bit-identical to dsp_dag_max_alpha) leaving a 2-byte BLOCK trace [B,T,L] instead of the 4-byte arg-max trace, and a
 * back-trace that recomputes the arg-max of the cells it visits from alpha_max, the block trace and the links (same tie rule: smallest index).
 * dsp_dag_max_alpha_blocks_supported(L, TR) tells whether a shape is served (1) or needs dsp_dag_max_alpha / dsp_dag_backtrace (0). */
int dsp_dag_max_alpha_blocks_supported(int L, int TR);
int dsp_dag_max_alpha_blocks(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                             float* alpha_max, uint16_t* block_trace, int B, int T, int L, int TR, dsp_stream_t stream);
int dsp_dag_backtrace_blocks(const float* alpha_max, const uint16_t* block_trace, const float* links, const int64_t* out_len,
                             const int64_t* tgt_len, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Diagnostics (no reference counterpart).
 *   dsp_dag_set_option("dp_path", n) pins the DP kernel family FOR THE CALLING THREAD: 0 = auto, 1 = generic row-sequential /
 *   log-space dense, 2 = banded 2-column log-space strips, 4 = strip2 (2 vertices per lane), 5 = strip4g (exp-space, one exponent per
 *   lane group; the auto choice for TR <= 32), 7 = values-only max-DP strips + lazy back-trace (dag_best_alignment), 8 = strip2g / strip1g (r06: exp space,
 *   2 vertices x 64 transitions / 1 x 128 per lane; the auto choice of the forward for windows 33 .. 64 / 65 .. 128, where 2 / 9 keep the log-space
 *   strips / the dense-window kernels),
 *   9 = dense-window blocked products on the f32 matrix cores (the auto choice for TR > 64); used by tests to
 *   cross-check the families.  "k5_path": 0 = auto, 1 = tiled log-space grad_links kernel, 2 = exp-space (TR <= 32) / block products
 *   (TR > 64).  Windows 33 .. 128 with 16-byte aligned rows (r06): the TR <= 32 kernel with one plane of workgroups per block of 32 transitions, beta
 *   read 32 k columns to the right (family 6) — 3 pins it, auto takes it where its estimated cost is under the tiled kernel's (long target axes);
 *   the block products are the auto choice above 128 only.  "k5_fuse" 0|1|2|3 (r06): dsp_dag_loss_bwd on a banded graph (TR <= 32) asked for BOTH gradients — 0 = auto (2), 1 / 2 = ONE
 *   launch writes grad_match and grad_links (alpha, beta, match read once; match rows prefetched into registers with 4-row passes / by
 *   LDS-DMA with 3-row passes), 3 = the two launches of r01-r05 (K4, then K5); the three are bit-identical.  "dm_mt" 1|2: rows per chunk of the dense kernel in MFMA row tiles (default 2 = 32 rows), "dm_depth" 1|2: its register
 *   stages in flight with dm_mt 1 (9 with dm_mt 2: the one-workgroup-per-CU build), "force_generic", "dm_*" affect speed only.
 *   "dm_budget": the dense kernel hands a batch exp space cannot hold (finite transitions under e^-86, or more exact-redo work than
 *   one visited predecessor per (row, 64-column block)) to log-space stand-by kernels queued behind it — 0 = auto, n > 0 = that many
 *   visited predecessors, -1 = no stand-by (diagnostics only: such batches are then slow and their weakest terms unguarded).
 *   "mx_cpl" 0|1|2|4 (r04): vertices per lane of the banded max-DP (0 = auto: 4 when that still gives >= 200 workgroups, else 2; 1 exists
 *   for the co-residency measurement of profiles/r04_dp_coresidency.txt), "bt_ring" 1|0 (r04): the LDS-ring back-trace (TR == 32) or the
 *   r01-r03 window kernel — every combination returns bit-identical paths.
 *   "dm_mt" 3|4|14 (r05): 48- / 64-row chunks at two workgroups per CU, 64 rows at one — measurement builds, bit-identical, slower.
 *   "dx_mt" 0|1|2 (r05): rows per chunk of the dense max-plus alignment kernel in 16-row tiles (0 = auto: 2 for launches of >= 6 rounds of
 *   workgroups); bit-identical paths.  "xl_tile" n (r05): n > 0 forces the TILED extract_links kernels (include/daspeech_decode.h) with a tile
 *   of n slots on any window — by default they serve the windows whose one-image score tile does not fit LDS (TR above ~1100).
 *   "xl_mfma" -1|0|1 (r05): the matrix-core extract_links kernels (dsp_extract_links_ws / _bwd_ws) by size | never | wherever H = 8, CK = 64
 *   — it steers what dsp_extract_links_workspace reports.  "xl_contract" -1|0|1 (r05): the contractions of dsp_extract_links_bwd_ws as
 *   exact-fp32 MFMAs (0), as bf16-triple products (1: both operands split into three bf16 pieces, six MFMAs per product, fp32 range and
 *   ~2^-23 relative) or by size (-1: the latter above ~1 500 vertices).
 *   dsp_dag_last_launch_status copies the device-side status word of the last fast-path launch on `stream` to *host_word (0 = clean, bit0 = a bounded hand-off spin timed out); it synchronises
 *   the stream and is meant for tests. */
int dsp_dag_alignment_trace_optional(int L, int TR);
/* diagnostics of the grad_links kernels since the last call: out4 = {lanes redone exactly, unsafe factor, weak link (exp-space kernel),
 * family of the last grad_links launch: 1 tiled log space, 2 exp space, 3 dense block products, 4 / 5 exp space fused with grad_match
 * (k5_fuse 1 / 2), 6 exp space, a plane of workgroups per 32 transitions (windows 33 .. 128), 0 none} */
int dsp_dag_debug_k5(unsigned int* out4);
int dsp_dag_set_option(const char* name, int value);
int dsp_dag_last_launch_status(dsp_stream_t stream, unsigned int* host_word);
/* cells that took the exp-space kernel's exact log-space fallback in the launch last queried by the call above */
unsigned int dsp_dag_last_fallback_count(void);
/* raw status words (63) captured by the last dsp_dag_last_launch_status call: [0] error, [1] fallback count,
 * [7+4i..] = {sample, row, column, bits of S} of the first 14 fallback cells */
const unsigned int* dsp_dag_debug_words(void);

#ifdef __cplusplus
}
#endif
#endif /* DASPEECH_DAG_H */
