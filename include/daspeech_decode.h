/* daspeech_decode.h — C ABI of the inference-side steps of the DASpeech hot path (libdaspeech_hip.so).
 *
 * These entry points replace Python / torch code of the reference that runs on the host or as chains of small torch ops:
 *   graph decode (lookahead / greedy)   DASpeech/models/s2s_conformer_dag_fastspeech2.py:201-243   (F2, F3 in SURVEY.md §2.3)
 *   posterior of the "expect" strategy   DASpeech/criterions/s2s_dag_fastspeech2_loss.py:259-261     (F1)
 *   variance-adaptor glue               fairseq/fairseq/models/text_to_speech/fastspeech2.py:169-210 (F6)
 *   length regulator                    fairseq/fairseq/models/text_to_speech/fastspeech2.py:98-114  (F7)
 * Conventions as in daspeech_dag.h (device pointers, caller-allocated outputs, hipStream_t as void*, int return codes). */
#ifndef DASPEECH_DECODE_H
#define DASPEECH_DECODE_H

#include "daspeech_dag.h"

#ifdef __cplusplus
extern "C" {
#endif

/* F2a  per-vertex argmax token and its log-probability            (s2s_conformer_dag_fastspeech2.py:207-208)
 *   logits [B,L,V] (dtype code as in daspeech_dag.h, read only); tok [B,L] int32 = argmax_v (first maximum);
 *   score [B,L] fp32 = max_v log_softmax(logits) = -log sum_v exp(x_v - max). */
int dsp_argmax_logp(const void* logits, int dtype, int32_t* tok, float* score, int B, int L, int V, dsp_stream_t stream);

/* F2b  best successor of every vertex on the COMPACT links layout  (:209-217; replaces restore_valid_links + dense argmax)
 *   links [B,L,TR] fp32; score [B,L] fp32 (ignored when greedy != 0); next[b,i] = argmax_j (links[b,i,j-i-1] + score[b,j]*beta)
 *   with the dense row's tie rule: first maximum = smallest j; a row without any finite entry gives 0. */
int dsp_lookahead_next(const float* links, const float* score, float beta, int greedy, int32_t* next,
                       int B, int L, int TR, dsp_stream_t stream);

/* F3a  follow the path 0 -> ... -> L_b-1, collapse repeats, drop pads        (:219-233)
 *   out_tokens [B,cap] int64 (pad-filled; [b,0] = token of vertex 0), keep_idx [B,cap] int32 = vertex of each kept non-bos
 *   token (-1 padded), n_feat [B] int32 = number of kept non-bos tokens. cap >= L is always enough. */
int dsp_follow_path(const int32_t* next, const int32_t* tok, const int64_t* out_len, int pad,
                    int64_t* out_tokens, int32_t* keep_idx, int32_t* n_feat, int B, int L, int cap, dsp_stream_t stream);

/* F3a' the token pass of the viterbi / jointviterbi strategies                 (s2s_conformer_dag_fastspeech2.py:283-299)
 *   path [B,L] int64 (DP row of every vertex on the back-traced chain, -1 elsewhere: dsp_dag_backtrace / dsp_dag_backtrace_blocks),
 *   pred_length [B] int64 (chosen length), unreachable [B] uint8 (no length reaches the final vertex: the token of vertex 0 is emitted,
 *   as the reference does), tok [B,L] int32.  Visited = DP rows 1 .. pred_length, graph order; a token is kept if it is the last visited
 *   one, or not <pad> and different from the next visited token.  out_tokens [B,cap] int64 (pad-filled), keep_idx [B,cap] int32 (vertex of
 *   each kept token, -1 padded: the index list of dsp_gather_rows), n_keep [B] int32. */
int dsp_viterbi_collect(const int64_t* path, const int64_t* pred_length, const unsigned char* unreachable, const int32_t* tok, int pad,
                        int64_t* out_tokens, int32_t* keep_idx, int32_t* n_keep, int B, int L, int cap, dsp_stream_t stream);

/* F3b  gather the decoder states of the kept vertices, zero padded            (:232,234,241; _collate_frames)
 *   features [B,L,D] (dtype code), keep_idx [B,cap]; out [B,Fmax,D] same dtype: out[b,k] = features[b,keep_idx[b,k]] for
 *   k < n_feat[b], 0 after.  Pure copy: bit-exact. */
int dsp_gather_rows(const void* features, int dtype, const int32_t* keep_idx, const int32_t* n_feat, void* out,
                    int B, int L, int D, int cap, int Fmax, dsp_stream_t stream);

/* F0   transition log-probabilities of the graph, compact layout             (DAGDecoder.extract_links, s2t_conformer_dag.py:171-212)
 *   q, k [B,L,H,CK] fp32, CK = 32, 64 or 128 (query_linear / key_linear of [features ; link positional embedding], H = 8 heads),
 *   log_gates [B,L,H] fp32 (log_softmax of gate_linear), out_len [B] int64, dist_bias [TR] fp32 or NULL (benchmark calibration),
 *   scale = 1/sqrt(CK).  links[b,i,d] = logsumexp_h( log_softmax_d(q_i.k_{i+d+1} * scale, over valid successors) + log_gates[b,i,h] ),
 *   -inf where i+d+1 >= out_len[b] or >= L; rows without a successor are all -inf.  Only the band is computed — the reference's
 *   [B,L,L,H] content tensor and its gather (:183-196) never exist.  Inference path (no gradient).
 *   Any TR up to L-1: windows whose [4 vertices][TR][8 heads] score image exceeds LDS (TR above ~1100) are walked in tiles of 512 successors
 *   (online soft-max state in a first pass, per-tile emission in a second); the same holds for the two training entry points below. */
int dsp_extract_links(const float* q, const float* k, const float* log_gates, const int64_t* out_len,
                      const float* dist_bias, float* links, int B, int L, int H, int CK, int TR, float scale,
                      dsp_stream_t stream);
/* The TRAINING side of F0 (the step in front of dag_loss: s2t_conformer_dag.py:171-212 under autograd), still on the compact band only:
 *   dsp_extract_links_train   the same forward, additionally writing `stats` [B,L,H,2] = (window maximum, log of the window's sum of
 *                             exp(score - maximum)) per source vertex and head — all the state the backward needs besides q, k, links;
 *   dsp_extract_links_bwd     grad_links [B,L,TR] (entries of -inf links are ignored) -> grad_q, grad_k [B,L,H,CK], grad_log_gates [B,L,H].
 *                             Scores are recomputed per 4-vertex tile, d(score) lives in LDS only: no [B,L,L,H] tensor in either direction. */
int dsp_extract_links_train(const float* q, const float* k, const float* log_gates, const int64_t* out_len,
                            const float* dist_bias, float* links, float* stats, int B, int L, int H, int CK, int TR, float scale,
                            dsp_stream_t stream);
int dsp_extract_links_bwd(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                          const float* links, const float* grad_links, const float* stats,
                          float* grad_q, float* grad_k, float* grad_log_gates, int B, int L, int H, int CK, int TR, float scale,
                          dsp_stream_t stream);

/* F0 on the matrix cores (r05; csrc/extract_links_mfma.hip): the same three operators for the released link predictor (H = 8, CK = 64), scores as
 *   32 x 32 blocks on the fp16 matrix cores with fp32 accuracy (operands split hi + lo 2^-11, three MFMAs per product), the backward's
 *   contractions on the fp32 matrix-core path; they need a scratch buffer the caller owns (the split k / q rows in MFMA fragment order):
 *   dsp_extract_links_workspace  bytes for one call (phase 0 = inference forward, 1 = training forward, 2 = backward); 0 bytes = this shape is
 *                                served by the entry points above (other head widths, short graphs / narrow windows — or option "xl_mfma" 0;
 *                                "xl_mfma" 1 forces the matrix-core kernels wherever H = 8, CK = 64);
 *   dsp_extract_links_ws         dsp_extract_links (stats = NULL) / dsp_extract_links_train (stats [B,L,H,2]) — same outputs, same `stats`;
 *   dsp_extract_links_bwd_ws     dsp_extract_links_bwd.
 *   Operand range: the split keeps k and q * scale * log2(e) as fp16 pairs — finite for |k|, |q| * 0.18 < 65 504 (the link predictor's
 *   projections are O(1..10)); beyond that the scores become inf / NaN where the fp32-FMA entry points above still work.
 *   B = 32, L = 4096, TR = L-1 (BASELINE's graph with the README's --max-transition-length 99999): see DESIGN.md §8 for the measured times. */
int dsp_extract_links_workspace(int B, int L, int H, int CK, int TR, int phase, size_t* bytes);
int dsp_extract_links_ws(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                         float* links, float* stats, int B, int L, int H, int CK, int TR, float scale,
                         void* workspace, size_t workspace_bytes, dsp_stream_t stream);
int dsp_extract_links_bwd_ws(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                             const float* links, const float* grad_links, const float* stats,
                             float* grad_q, float* grad_k, float* grad_log_gates, int B, int L, int H, int CK, int TR, float scale,
                             void* workspace, size_t workspace_bytes, dsp_stream_t stream);

/* diagnostics (r06): the extract_links kernel families launched by this process since the last call (then cleared) — bit 0 one-image forward,
 *   1 tiled forward, 2 matrix-core forward, 3 one-image backward, 4 tiled backward, 5 matrix-core backward with exact-fp32 contractions,
 *   6 matrix-core backward with bf16-triple contractions.  The "xl_tile" / "xl_mfma" / "xl_contract" options are PROCESS-wide (PyTorch runs an
 *   autograd backward on its own worker thread); tests assert through this word that the family they pinned is the one that ran. */
unsigned int dsp_extract_links_debug_ran(void);
/* RANGE of the matrix-core kernels (dsp_extract_links_ws / _bwd_ws): operands are split into fp16 hi / lo pieces, so |k| and |q| * scale * log2(e)
 *   must stay under 65 000 (link-predictor inputs are projections of layer-normed features: |x| ~ 1-10).  A larger operand is clamped — the call
 *   returns finite, wrong values for it — and raises a device flag; this call returns the flag (1 = some operand was clamped since the last call)
 *   and clears it.  It synchronises the device: for tests and debugging.  The fp32-FMA kernels (dsp_extract_links, "xl_mfma" 0) have no limit. */
unsigned int dsp_extract_links_debug_range(void);

/* F1   posterior of the forward-backward pass                                  (s2s_dag_fastspeech2_loss.py:259-261)
 *   score[b,t,:] = exp(alpha+beta - logsumexp_j(alpha+beta)), NaN -> 0 (rows without any finite entry). fp32 [B,T,L]. */
int dsp_posterior(const float* alpha, const float* beta, float* score, int B, int T, int L, dsp_stream_t stream);
/* F1 fused with its consumer (:259-262): out[b,t,:] = sum_j score[b,t,j] * features[b,j,:] without the [B,T,L] score tensor.
 *   features [B,L,D], out [B,T,D] fp32 (D even), lse [B,T] (row log-sum-exp of alpha+beta; -inf for rows without a finite entry,
 *   whose output is 0) or NULL.  dsp_posterior_features_bwd: grad_features[b,j,:] = sum_t score[b,t,j] * grad_out[b,t,:], the
 *   score rebuilt from alpha, beta and lse (alpha / beta carry no gradient: the reference detaches them, dag_loss.py:180-186). */
int dsp_posterior_features(const float* alpha, const float* beta, const float* features, float* out, float* lse,
                           int B, int T, int L, int D, dsp_stream_t stream);
int dsp_posterior_features_bwd(const float* alpha, const float* beta, const float* lse, const float* grad_out, float* grad_features,
                               int B, int T, int L, int D, dsp_stream_t stream);

/* F6a  predicted durations                                                     (fastspeech2.py:202-205)
 *   dur = clamp(round((exp(log_dur) - 1) * factor), 0) as int64, 0 where pad_mask != 0 (uint8/bool). */
int dsp_durations(const float* log_dur, const uint8_t* pad_mask, float factor, int64_t* dur, int64_t n, dsp_stream_t stream);

/* F6b  x += Embedding[bucketize(v, bins)]                                      (fastspeech2.py:169-177,207-210)
 *   x [n,C] fp32 in/out, v [n] fp32, bins [nb] fp32 ascending (torch.bucketize right=False), emb [nb+1,C] fp32. */
int dsp_bucketize_embed_add(float* x, const float* v, const float* bins, int nb, const float* emb, int64_t n, int C,
                            dsp_stream_t stream);

/* F7   length regulator                                                        (fastspeech2.py:98-114)
 *   step 1: dsp_length_regulator_lens  : out_lens[b] = sum_t dur[b,t]; cum [B,N] int64 scratch = inclusive prefix sums.
 *   step 2: dsp_length_regulator_expand: out [B,maxlen,C] (dtype code) = rows of x [B,N,C] repeated dur times, zero padded.
 *   The caller reads max(out_lens) between the two (the output shape depends on it; the reference syncs B*N times). */
int dsp_length_regulator_lens(const int64_t* dur, int64_t* cum, int64_t* out_lens, int B, int N, dsp_stream_t stream);
int dsp_length_regulator_expand(const void* x, int dtype, const int64_t* cum, void* out, int B, int N, int C, int maxlen,
                                dsp_stream_t stream);

/* Conformer convolution module, eval mode (fairseq conformer_layer.py ConvolutionModule: depthwise_conv -> batch_norm -> SiLU),
 * on the channels-last tensor:   y[b,t,c] = SiLU( BN_eval( sum_k w[c,k] * x[b,t+k-(K-1)/2,c] ) ),  zero padding outside [0,T).
 *   x, y [B,T,C] fp32 (16-byte aligned, C % 4 == 0, y != x); w [C,K] fp32 (the Conv1d(C,C,K,groups=C) weight [C,1,K]);
 *   bn_w / bn_b may be NULL (affine off); K in {3, 7, 15, 31}. */
int dsp_dwconv_bn_silu(const float* x, const float* w, const float* bn_w, const float* bn_b, const float* bn_mean,
                       const float* bn_var, float eps, float* y, int B, int T, int C, int K, dsp_stream_t stream);

/* fp32-accurate Conv1d ("same" padding, stride 1) on the fp16 matrix cores by operand splitting (x = xh + xl/2048, w = wh + wl/2048;
 * the products xh.wh, xh.wl, xl.wh are exact in the fp32 accumulator, the dropped xl.wl term is 2^-22 relative) — for the
 * FastSpeech2 FFT feed-forward convolutions (fairseq fastspeech2.py:42-63), which MIOpen runs at 60-70 TFLOP/s in fp32.
 * Range: operands are split into fp16 hi / lo parts, so |x| and |w| must stay below the fp16 maximum (65504; larger values become
 * inf) — true of layer-normalised activations and trained weights, not of arbitrary data; values under 6e-5 keep fewer than 22 bits.
 *   dsp_conv1d_split_pack   fp32 weight, tap-major [ntaps][M][CI] -> w_hi, w_lo (dsp_conv1d_split_packed_elems halves each)
 *   dsp_conv1d_split        x [B,T,nslices*CI] fp32, row stride ldx (a channel slice of a wider tensor is fine); out [B,T,M] fp32, row
 *                           stride ldo;  out = act([out +] bias + conv(x)), act = relu: 0 none, 1 ReLU, 2 SiLU, 3 GELU (erf).
 *                           ntaps = 1 is a Linear layer.  CI in {128, 256, 512} is the SLICE width: a wider input is nslices slices
 *                           walked inside one launch (w_hi / w_lo hold the slices' packed weights one after the other);
 *                           accumulate = 1 adds to what `out` holds.  ntaps odd, M % 4 == 0. */
long dsp_conv1d_split_packed_elems(int ntaps, int M, int CI);
int dsp_conv1d_split_pack(const float* w_tap_major, void* w_hi, void* w_lo, int ntaps, int M, int CI, dsp_stream_t stream);
int dsp_conv1d_split(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, float* out, long ldo,
                     int B, int T, int CI, int nslices, int M, int ntaps, int relu, int accumulate, dsp_stream_t stream);
/* same with the layer's residual connection in the epilogue:  out = res + alpha * act(bias + conv(x))   (res [B,T,M], row stride ldr;
 * out may be res) — `x + 0.5 * ffn(x)`, `x + attn(x)` without separate scale / add launches */
int dsp_conv1d_split_residual(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res, long ldr,
                              float alpha, float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps, int relu,
                              dsp_stream_t stream);


/* the same for a ragged batch: lens [B] (device, int32) are the samples' valid lengths; a time tile that starts at or after
 * lens[b] + slack is padding no valid output depends on (slack = the frames of padding later convolutions still reach into: 0 for
 * position-wise layers, (K-1)/2 per convolution that follows) — it is not computed and its output rows are written as ZEROS (finite, so
 * that masked attention keys and later position-wise layers stay finite).  Rows below that bound get exactly the bits of
 * dsp_conv1d_split_residual.  res may be NULL with alpha = 1 (plain layer); lens may be NULL (dense batch). */
int dsp_conv1d_split_ragged(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res, long ldr,
                            float alpha, float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps, int relu,
                            const int* lens, int slack, dsp_stream_t stream);

/* act(W . LayerNorm(x) + b) [as residual / alpha of dsp_conv1d_split_residual] for a Linear layer over exactly 256 input channels: the
 * LayerNorm (torch semantics, weight and bias [256]) is applied while the row tile is staged — a row's 256 channels sit in half a wave —
 * so pre-norm blocks (the Conformer's self-attention and convolution modules: LayerNorm -> linear_q|k|v, LayerNorm -> pointwise_conv1,
 * conformer_layer.py:254-281) need no LayerNorm launch and no normalised copy of x in HBM.  w_hi / w_lo as dsp_conv1d_split_pack packs a
 * one-tap layer; lens / slack as dsp_conv1d_split_ragged (NULL: dense). */
int dsp_linear_ln_split(const float* x, long ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* w_hi, const void* w_lo,
                        const float* bias, const float* res, long ldr, float alpha, float* out, long ldo, int B, int T, int M, int act,
                        const int* lens, int slack, dsp_stream_t stream);

/* the same layer for SHORT sequences (few time tiles: the FastSpeech2 encoder's K = 9 convolutions over ~60 phoneme positions leave
 * three quarters of the CUs idle and run a 288-step reduction per workgroup): the K dimension is split over nslices * tap_groups
 * workgroups per output tile (one 512-channel input slice and ceil(ntaps / tap_groups) taps each), raw partial sums go to `workspace`
 * (dsp_conv1d_split_ksplit_workspace_bytes) and a second launch adds them in a fixed order with bias, activation and residual:
 *   out = res + alpha * act(bias + sum over parts).   Not bit-identical to dsp_conv1d_split (different association of the K sum).
 * lens / slack as dsp_conv1d_split_ragged (NULL: dense); skipped tiles contribute zero partial sums, so their rows come back as
 * res + alpha * act(bias) — finite padding. */
size_t dsp_conv1d_split_ksplit_workspace_bytes(int B, int T, int M, int nslices, int tap_groups);
int dsp_conv1d_split_ksplit(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res, long ldr, float alpha,
                            float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps, int act, int tap_groups,
                            void* workspace, size_t workspace_bytes, const int* lens, int slack, dsp_stream_t stream);

/* The Conformer's feed-forward module in one matrix-core launch (+ a fixed-order reduction of the hidden-channel groups), fp32 accuracy:
 *   out = res + alpha * (W2 . act(W1 . LN(x) + b1) + b2)          (fairseq conformer_layer.py:140-146 called as x + 0.5 * ffn(x), :254-281)
 * x [B,T,C] (row stride ldx), ln_w / ln_b [C] or both NULL (no LayerNorm), W1 [H,C] and W2 [C,H] as packed by dsp_conv1d_split_pack
 * (one tap; W2 in 512-channel input slices as dsp_conv1d_split takes them), b1 [H], b2 [C] or NULL, res [B,T,C] (row stride ldr) or
 * NULL, out [B,T,C] (row stride ldo; may be res).  act as dsp_conv1d_split's relu argument (0 none, 1 ReLU, 2 SiLU, 3 GELU).
 * C = 256, H a multiple of 512.  workspace: dsp_ffn_split_workspace_bytes(B, T, C, H) bytes of device memory (partial sums).
 * post_ln_w / post_ln_b (both or neither) + out_ln [B,T,C] contiguous: the reduction also writes LayerNorm(out) there (the block that
 * follows in a pre-norm layer starts with one); out itself may then be NULL when only the normalised rows are needed. */
size_t dsp_ffn_split_workspace_bytes(int B, int T, int C, int H);
int dsp_ffn_split(const float* x, long ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* w1_hi, const void* w1_lo, const float* b1,
                  const void* w2_hi, const void* w2_lo, const float* b2, const float* res, long ldr, float alpha, float* out, long ldo,
                  void* workspace, size_t workspace_bytes, int B, int T, int C, int H, int act, const float* post_ln_w, const float* post_ln_b,
                  float post_ln_eps, float* out_ln, dsp_stream_t stream);

/* LayerNorm over the last dimension (torch.nn.LayerNorm semantics: biased variance, eps inside the square root), one wave per row:
 * x, y [rows, C] fp32 contiguous (y may be x), w / b [C] or NULL, C % 4 == 0, C <= 2048, all pointers 16-byte aligned. */
int dsp_layer_norm(const float* x, const float* w, const float* b, float eps, float* y, long rows, int C, dsp_stream_t stream);

/* Conformer relative-position self-attention (fairseq conformer_layer.py / espnet RelPositionMultiHeadedAttention), fused, fp32:
 *   out[b,i,h,:] = sum_j softmax_j( ((q_i + u_h).k_j + (q_i + v_h).p_{(T-1)-i+j}) / sqrt(dk) ; keys with pad_mask[b,j] != 0 -> -inf ) v_j
 * q, k, v [B,T,H,dk] fp32 with `ld` floats between consecutive positions (H*dk for separate linear_q / linear_k / linear_v outputs,
 * 3*H*dk for the slices of a fused projection; samples T*ld apart), out [B,T,H,dk] contiguous, p [2T-1,H,dk] (linear_pos of
 * the relative positional encoding, rows for relative positions T-1 .. -(T-1)), bias_u / bias_v [H,dk], pad_mask [B,T] bytes or NULL.
 * dk = 64, any T.  Runs on the fp16 matrix cores at fp32 accuracy like dsp_attention_split (below); the [query][relative position]
 * product is formed per 32-query x 64-position block and shifted into [query][key] through LDS. */
int dsp_relpos_attention(const float* q, const float* k, const float* v, long ld, const float* p, const float* bias_u, const float* bias_v,
                         const unsigned char* pad_mask, float* out, int B, int T, int H, int DK, dsp_stream_t stream);

/* Multi-head attention with a key padding mask (fairseq modules/multihead_attention.py in eval mode: softmax(q k^T * scale + mask) v), at fp32
 * accuracy on the fp16 matrix cores (every operand split into an fp16 hi / lo pair, three MFMAs per product):
 *   out[b,i,h,:] = sum_j softmax_j( scale * q[b,i,h,:] . k[b,j,h,:] ; keys with key_pad_mask[b,j] != 0 -> -inf ) v[b,j,h,:]
 * q [B,N,H,dk], k / v [B,M,H,dk] fp32 as row-strided views (ldq / ldk / ldv floats between consecutive positions, >= H*dk, %4 == 0;
 * samples N*ldq / M*ldk / M*ldv apart: the slices of a fused q|k|v projection are served without a copy), key_pad_mask [B,M] bytes or
 * NULL, out [B,N,H*dk] contiguous.  dk = 64 or 128.  A sample whose keys are all masked gets NaN rows, as torch's soft-max does.
 * q_lens [B] (device int32) or NULL: queries at or after q_lens[b] + q_slack are padding no valid output depends on; their 32-query
 * groups are not computed and their output rows are written as zeros (see dsp_conv1d_split_ragged). */
int dsp_attention_split(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const unsigned char* key_pad_mask,
                        float* out, int B, int N, int M, int H, int DK, float scale, const int* q_lens, int q_slack, dsp_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* DASPEECH_DECODE_H */
