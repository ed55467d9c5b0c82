// decode_tts.hip — inference-side steps of the DASpeech hot path for gfx950: graph decode on the compact links layout,
// posterior of the expect strategy, variance-adaptor glue, length regulator.  All HBM-/latency-bound integer & copy work:
// coalesced loads, LDS staging of per-sample state, no host round trips inside (the reference runs these as Python loops with
// .tolist()/utils.item syncs, s2s_conformer_dag_fastspeech2.py:209-243 and fastspeech2.py:106-112).
#include "common.h"
#include "../../include/daspeech_decode.h"

namespace dsp {

// ---------------------------------------------------------------- F2a: argmax token + log-prob of it, one row per wave-group
template <typename T>
__global__ __launch_bounds__(256) void argmax_logp_kernel(const T* __restrict__ x, int32_t* __restrict__ tok,
                                                          float* __restrict__ score, long nrows, int V)
{
    __shared__ float r_m[4], r_s[4];
    __shared__ int r_a[4];
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const T* p = x + (size_t)row * V;
        float m = NEG_INF, s = 0.f; int a = 0x7fffffff;
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            const float f = to_f(p[v]);
            if (f > m) { s = (m == NEG_INF) ? 1.f : s * __expf(m - f) + 1.f; m = f; a = v; }      // strict >: first maximum
            else if (f != NEG_INF) s += __expf(f - m);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64); const int a2 = __shfl_xor(a, o, 64);
            const float nm = fmaxf(m, m2);
            const float ns = (nm == NEG_INF) ? 0.f : s * __expf(m - nm) + s2 * __expf(m2 - nm);
            a = (m2 > m) ? a2 : ((m2 == m) ? min(a, a2) : a);
            m = nm; s = ns;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { r_m[threadIdx.x >> 6] = m; r_s[threadIdx.x >> 6] = s; r_a[threadIdx.x >> 6] = a; }
        __syncthreads();
        if (threadIdx.x == 0) {
            m = r_m[0]; s = r_s[0]; a = r_a[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
                const float m2 = r_m[w], s2 = r_s[w]; const int a2 = r_a[w];
                const float nm = fmaxf(m, m2);
                const float ns = (nm == NEG_INF) ? 0.f : s * __expf(m - nm) + s2 * __expf(m2 - nm);
                a = (m2 > m) ? a2 : ((m2 == m) ? min(a, a2) : a);
                m = nm; s = ns;
            }
            tok[row] = (a == 0x7fffffff) ? 0 : a;
            score[row] = -__logf(s);
        }
    }
}

// ---------------------------------------------------------------- F2b: lookahead / greedy successor on compact links
__global__ __launch_bounds__(256) void lookahead_next_kernel(const float* __restrict__ links, const float* __restrict__ score,
                                                             float beta, int greedy, int32_t* __restrict__ next,
                                                             int B, int L, int TR)
{
    const long n = (long)B * L;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int b = (int)(e / L), i = (int)(e % L);
        const float* lk = links + (size_t)e * TR;
        const float* sc = score + (size_t)b * L;
        float best = NEG_INF; int arg = 0;
        const int dmax = min(TR, L - 1 - i);
        for (int d = 0; d < dmax; ++d) {
            float v = lk[d];
            if (!greedy) v = __fadd_rn(v, __fmul_rn(sc[i + d + 1], beta));      // two roundings, like the torch expression
            if (v > best) { best = v; arg = i + d + 1; }
        }
        next[e] = arg;
    }
}

// wide windows (TR > 64, README's --max-transition-length 99999): one WAVE per source vertex, lanes stride the successors so that the
// row is read in 256-byte runs (a thread per vertex walks its own 1.6 KB row alone: 179 us at B=64, L=400); same result: the
// first maximum (smallest successor index among equal values)
__global__ __launch_bounds__(256) void lookahead_next_wave_kernel(const float* __restrict__ links, const float* __restrict__ score,
                                                                  float beta, int greedy, int32_t* __restrict__ next,
                                                                  int B, int L, int TR)
{
    const int lane = threadIdx.x & 63;
    const long n = (long)B * L;
    for (long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += (long)gridDim.x * 4) {
        const int b = (int)(e / L), i = (int)(e % L);
        const float* lk = links + (size_t)e * TR;
        const float* sc = score + (size_t)b * L;
        float best = NEG_INF; int arg = 0x7fffffff;
        const int dmax = min(TR, L - 1 - i);
        for (int d = lane; d < dmax; d += 64) {
            float v = lk[d];
            if (!greedy) v = __fadd_rn(v, __fmul_rn(sc[i + d + 1], beta));
            if (v > best) { best = v; arg = i + d + 1; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64); const int a2 = __shfl_xor(arg, o, 64);
            if (v2 > best || (v2 == best && a2 < arg)) { best = v2; arg = a2; }
        }
        if (lane == 0) next[e] = (best == NEG_INF) ? 0 : arg;       // nothing beat -inf: index 0, as the sequential scan leaves it
    }
}

// ---------------------------------------------------------------- F3a: path follow (one workgroup per sample, walk in LDS)
__global__ __launch_bounds__(256) void follow_path_kernel(const int32_t* __restrict__ next, const int32_t* __restrict__ tok,
                                                          const int64_t* __restrict__ out_len, int pad,
                                                          int64_t* __restrict__ out_tokens, int32_t* __restrict__ keep_idx,
                                                          int32_t* __restrict__ n_feat, int L, int cap)
{
    extern __shared__ int32_t sm[];          // nxt[L], tk[L], otok[cap], okeep[cap]
    int32_t* nxt = sm; int32_t* tk = sm + L; int32_t* otok = tk + L; int32_t* okeep = otok + cap;
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < L; j += blockDim.x) { nxt[j] = next[(size_t)b * L + j]; tk[j] = tok[(size_t)b * L + j]; }
    for (int k = threadIdx.x; k < cap; k += blockDim.x) { otok[k] = pad; okeep[k] = -1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int Lb = (int)out_len[b];
        int n = 0;
        if (Lb >= 1 && Lb <= L) {
            int last = tk[0], j = 0;
            otok[0] = last;
            for (int guard = 0; j != Lb - 1 && guard < L; ++guard) {           // valid edges strictly increase j
                j = nxt[j];
                if (j < 0 || j >= L) break;
                const int now = tk[j];
                if (now != pad && now != last) {
                    if (n + 1 < cap) otok[n + 1] = now;
                    if (n < cap) okeep[n] = j;
                    ++n;
                }
                last = now;
            }
        }
        n_feat[b] = n;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < cap; k += blockDim.x) {
        out_tokens[(size_t)b * cap + k] = otok[k];
        keep_idx[(size_t)b * cap + k] = okeep[k];
    }
}

// ---------------------------------------------------------------- F3b: row gather with zero padding (16-byte copies)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ feat, const int32_t* __restrict__ keep_idx,
                                                          const int32_t* __restrict__ n_feat, char* __restrict__ out,
                                                          int L, long row_bytes, int cap, int Fmax)
{
    const int b = blockIdx.y, k = blockIdx.x;
    const int n = n_feat[b];
    char* o = out + ((size_t)b * Fmax + k) * row_bytes;
    const int src = (k < n && k < cap) ? keep_idx[(size_t)b * cap + k] : -1;
    const char* s = (src >= 0 && src < L) ? feat + ((size_t)b * L + src) * row_bytes : nullptr;
    if (((row_bytes | (uintptr_t)feat | (uintptr_t)out) & 15) == 0) {
        for (long i = threadIdx.x * 16L; i < row_bytes; i += blockDim.x * 16L)
            *reinterpret_cast<uint4*>(o + i) = s ? *reinterpret_cast<const uint4*>(s + i) : make_uint4(0, 0, 0, 0);
    } else {
        for (long i = threadIdx.x; i < row_bytes; i += blockDim.x) o[i] = s ? s[i] : 0;
    }
}

// ---------------------------------------------------------------- F1: posterior = softmax_j(alpha + beta), NaN -> 0
__global__ __launch_bounds__(256) void posterior_kernel(const float* __restrict__ alpha, const float* __restrict__ beta,
                                                        float* __restrict__ score, long nrows, int L)
{
    __shared__ float red[8];
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const float* a = alpha + (size_t)row * L; const float* b = beta + (size_t)row * L;
        float* o = score + (size_t)row * L;
        float m = NEG_INF;
        for (int j = threadIdx.x; j < L; j += blockDim.x) m = fmaxf(m, a[j] + b[j]);
        m = wave_max(m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        if (!(m > NEG_INF) || isinf(m)) {                     // all -inf (or +inf/NaN garbage): the reference's NaN -> 0
            for (int j = threadIdx.x; j < L; j += blockDim.x) o[j] = 0.f;
            continue;
        }
        float s = 0.f;
        for (int j = threadIdx.x; j < L; j += blockDim.x) s += __expf(a[j] + b[j] - m);
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        s = red[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        const float lse = m + __logf(s);
        for (int j = threadIdx.x; j < L; j += blockDim.x) o[j] = __expf(a[j] + b[j] - lse);
    }
}

// ---------------------------------------------------------------- F1 fused: expected hidden states without the [B,T,L] score tensor
//   out[b,t,:] = sum_j softmax_j(alpha + beta)[b,t,j] * features[b,j,:]            (s2s_dag_fastspeech2_loss.py:259-262)
// One workgroup per (sample, PF_TT target rows): the rows' posteriors are built in LDS (log-sum-exp per row by one wave each), then
// every thread owns feature columns and walks the L vertices once for all PF_TT rows (the feature row is read once per workgroup,
// the posterior is an LDS broadcast).  `lse` [B,T] is kept for the backward.  Rows without a finite entry give 0 (the NaN -> 0).
constexpr int PF_TT = 8;
__global__ __launch_bounds__(256) void posterior_features_kernel(const float* __restrict__ alpha, const float* __restrict__ beta,
                                                                 const float* __restrict__ feats, float* __restrict__ out,
                                                                 float* __restrict__ lse_out, int T, int L, int D)
{
    extern __shared__ float pf_smem[];                     // [PF_TT][L]
    const int b = blockIdx.y, t0 = blockIdx.x * PF_TT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int tt = wave; tt < PF_TT; tt += 4) {
        const int t = t0 + tt;
        float* pr = pf_smem + (size_t)tt * L;
        if (t >= T) { for (int j = lane; j < L; j += 64) pr[j] = 0.f; continue; }
        const float* a = alpha + ((size_t)b * T + t) * L; const float* bb = beta + ((size_t)b * T + t) * L;
        float m = NEG_INF;
        for (int j = lane; j < L; j += 64) m = fmaxf(m, a[j] + bb[j]);
        m = wave_max(m);
        float lse = NEG_INF;
        if (m > NEG_INF && !isinf(m)) {
            float sum = 0.f;
            for (int j = lane; j < L; j += 64) sum += __expf(a[j] + bb[j] - m);
            sum = wave_sum(sum);
            lse = m + __logf(sum);
        }
        for (int j = lane; j < L; j += 64) pr[j] = (lse == NEG_INF) ? 0.f : __expf(a[j] + bb[j] - lse);
        if (lane == 0 && lse_out) lse_out[(size_t)b * T + t] = lse;
    }
    __syncthreads();
    const float* F = feats + (size_t)b * L * D;
    for (int d = 2 * tid; d < D; d += 512) {
        float2 acc[PF_TT];
#pragma unroll
        for (int tt = 0; tt < PF_TT; ++tt) acc[tt] = make_float2(0.f, 0.f);
        for (int j = 0; j < L; ++j) {
            const float2 f = *reinterpret_cast<const float2*>(F + (size_t)j * D + d);
#pragma unroll
            for (int tt = 0; tt < PF_TT; ++tt) { const float pv = pf_smem[(size_t)tt * L + j]; acc[tt].x = fmaf(pv, f.x, acc[tt].x); acc[tt].y = fmaf(pv, f.y, acc[tt].y); }
        }
#pragma unroll
        for (int tt = 0; tt < PF_TT; ++tt)
            if (t0 + tt < T) *reinterpret_cast<float2*>(out + ((size_t)b * T + t0 + tt) * D + d) = acc[tt];
    }
}

// backward wrt the features:  dF[b,j,:] = sum_t posterior[b,t,j] * dOut[b,t,:]  — a workgroup owns PF_TT vertices, rebuilds their
// posterior columns from alpha, beta and the saved row log-sum-exps, and walks the T rows of dOut once
__global__ __launch_bounds__(256) void posterior_features_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ beta,
                                                                     const float* __restrict__ lse, const float* __restrict__ dout,
                                                                     float* __restrict__ dfeats, int T, int L, int D)
{
    extern __shared__ float pf_smem[];                     // [T][PF_TT]
    const int b = blockIdx.y, j0 = blockIdx.x * PF_TT;
    const int tid = threadIdx.x;
    for (int e = tid; e < T * PF_TT; e += 256) {
        const int t = e / PF_TT, jj = e - t * PF_TT, j = j0 + jj;
        float pv = 0.f;
        if (j < L) {
            const float ls = lse[(size_t)b * T + t];
            const size_t o = ((size_t)b * T + t) * L + j;
            if (ls != NEG_INF) pv = __expf(alpha[o] + beta[o] - ls);
        }
        pf_smem[e] = pv;
    }
    __syncthreads();
    const float* G = dout + (size_t)b * T * D;
    for (int d = 2 * tid; d < D; d += 512) {
        float2 acc[PF_TT];
#pragma unroll
        for (int jj = 0; jj < PF_TT; ++jj) acc[jj] = make_float2(0.f, 0.f);
        for (int t = 0; t < T; ++t) {
            const float2 g = *reinterpret_cast<const float2*>(G + (size_t)t * D + d);
#pragma unroll
            for (int jj = 0; jj < PF_TT; ++jj) { const float pv = pf_smem[t * PF_TT + jj]; acc[jj].x = fmaf(pv, g.x, acc[jj].x); acc[jj].y = fmaf(pv, g.y, acc[jj].y); }
        }
#pragma unroll
        for (int jj = 0; jj < PF_TT; ++jj)
            if (j0 + jj < L) *reinterpret_cast<float2*>(dfeats + ((size_t)b * L + j0 + jj) * D + d) = acc[jj];
    }
}

// ---------------------------------------------------------------- F6
__global__ void durations_kernel(const float* __restrict__ log_dur, const uint8_t* __restrict__ pad, float factor,
                                 int64_t* __restrict__ dur, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = rintf(__fmul_rn(expf(log_dur[i]) - 1.0f, factor));      // torch.round = half-to-even
        long d = (long)v; if (d < 0) d = 0;
        dur[i] = pad[i] ? 0 : d;
    }
}

__global__ __launch_bounds__(256) void bucketize_embed_add_kernel(float* __restrict__ x, const float* __restrict__ v,
                                                                  const float* __restrict__ bins, int nb,
                                                                  const float* __restrict__ emb, long n, int C)
{
    for (long r = blockIdx.x; r < n; r += gridDim.x) {
        const float val = v[r];
        int lo = 0, hi = nb;                          // first index with bins[idx] >= val  (right=False)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (bins[mid] >= val) hi = mid; else lo = mid + 1; }
        const float* e = emb + (size_t)lo * C;
        float* xr = x + (size_t)r * C;
        for (int c = threadIdx.x; c < C; c += blockDim.x) xr[c] += e[c];
    }
}

// ---------------------------------------------------------------- F7
__global__ __launch_bounds__(256) void lr_lens_kernel(const int64_t* __restrict__ dur, int64_t* __restrict__ cum,
                                                      int64_t* __restrict__ out_lens, int N)
{
    __shared__ long wsum[4];
    __shared__ long carry;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < N; base += blockDim.x) {
        const int t = base + threadIdx.x;
        long v = (t < N) ? (long)dur[(size_t)b * N + t] : 0;
        if (v < 0) v = 0;
        long incl = v;                                   // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long u = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += u; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        long off = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
        if (t < N) cum[(size_t)b * N + t] = off + incl;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_lens[b] = carry;
}

__global__ __launch_bounds__(256) void lr_expand_kernel(const char* __restrict__ x, const int64_t* __restrict__ cum,
                                                        char* __restrict__ out, int N, long row_bytes, int maxlen)
{
    extern __shared__ long scum[];            // [N]
    const int b = blockIdx.y;
    for (int t = threadIdx.x; t < N; t += blockDim.x) scum[t] = cum[(size_t)b * N + t];
    __syncthreads();
    const long total = N ? scum[N - 1] : 0;
    const bool vec = ((row_bytes | (uintptr_t)x | (uintptr_t)out) & 15) == 0;
    // each wave copies one output frame at a time
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int o = blockIdx.x * nw + wave; o < maxlen; o += gridDim.x * nw) {
        char* dst = out + ((size_t)b * maxlen + o) * row_bytes;
        const char* src = nullptr;
        if (o < total) {
            int lo = 0, hi = N - 1;                 // first t with cum[t] > o
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (scum[mid] > o) hi = mid; else lo = mid + 1; }
            src = x + ((size_t)b * N + lo) * row_bytes;
        }
        if (vec) {
            for (long i = lane * 16L; i < row_bytes; i += 64 * 16L)
                *reinterpret_cast<uint4*>(dst + i) = src ? *reinterpret_cast<const uint4*>(src + i) : make_uint4(0, 0, 0, 0);
        } else {
            for (long i = lane; i < row_bytes; i += 64) dst[i] = src ? src[i] : 0;
        }
    }
}

static int elem_size(int dtype) { return dtype == DSP_F32 ? 4 : ((dtype == DSP_F16 || dtype == DSP_BF16) ? 2 : 0); }

}  // namespace dsp

using namespace dsp;

extern "C" int dsp_argmax_logp(const void* logits, int dtype, int32_t* tok, float* score, int B, int L, int V, dsp_stream_t stream)
{
    if (B < 0 || L < 0 || V <= 0) { set_error("argmax_logp: bad sizes"); return DSP_EINVAL; }
    if (B == 0 || L == 0) return DSP_OK;
    if (!logits || !tok || !score) { set_error("argmax_logp: null pointer"); return DSP_EINVAL; }
    const long nrows = (long)B * L;
    const int grid = (int)(nrows < 4096 ? nrows : 4096);
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case DSP_F32: hipLaunchKernelGGL(argmax_logp_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)logits, tok, score, nrows, V); break;
        case DSP_F16: hipLaunchKernelGGL(argmax_logp_kernel<__half>, dim3(grid), dim3(256), 0, st, (const __half*)logits, tok, score, nrows, V); break;
        case DSP_BF16: hipLaunchKernelGGL(argmax_logp_kernel<__hip_bfloat16>, dim3(grid), dim3(256), 0, st, (const __hip_bfloat16*)logits, tok, score, nrows, V); break;
        default: set_error("argmax_logp: unsupported dtype %d", dtype); return DSP_EINVAL;
    }
    return check_launch("argmax_logp");
}

extern "C" int dsp_lookahead_next(const float* links, const float* score, float beta, int greedy, int32_t* next,
                                  int B, int L, int TR, dsp_stream_t stream)
{
    if (B < 0 || L < 1 || TR < 1) { set_error("lookahead_next: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!links || !next || (!greedy && !score)) { set_error("lookahead_next: null pointer"); return DSP_EINVAL; }
    const long n = (long)B * L;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (TR > 64) {
        const long g4 = (n + 3) / 4;
        hipLaunchKernelGGL(lookahead_next_wave_kernel, dim3((unsigned)(g4 < 65535 * 4 ? g4 : 65535 * 4)), dim3(256), 0, as_stream(stream),
                           links, greedy ? links : score, beta, greedy, next, B, L, TR);
        return check_launch("lookahead_next");
    }
    hipLaunchKernelGGL(lookahead_next_kernel, dim3(grid), dim3(256), 0, as_stream(stream), links, greedy ? links : score, beta, greedy, next, B, L, TR);
    return check_launch("lookahead_next");
}

// F3a' — the token pass of the Viterbi strategies (s2s_conformer_dag_fastspeech2.py:283-299): the vertices the back-trace visited at DP rows
// 1 .. pred_length, in graph order; a token is kept if it is the LAST visited one, or (not <pad> and different from the token of the next
// visited vertex).  One workgroup per sample: thread = a contiguous chunk of vertices; the "next visited token" of a chunk's last visited
// vertex is the first visited token of the chunks behind it (one serial pass over 256 chunk heads), the compaction an exclusive scan of
// the per-chunk counts.  unreach[b] != 0: the final vertex was out of reach for every length — the reference's arg-maxes all return
// index 0 and it emits the token of vertex 0 (reproduced).
__global__ __launch_bounds__(256) void viterbi_collect_kernel(const int64_t* __restrict__ path, const int64_t* __restrict__ pred_len,
                                                              const unsigned char* __restrict__ unreach, const int32_t* __restrict__ tok, int pad,
                                                              int64_t* __restrict__ out_tokens, int32_t* __restrict__ keep_idx, int32_t* __restrict__ n_keep,
                                                              int L, int cap)
{
    extern __shared__ __attribute__((aligned(16))) int32_t vc_smem[];
    int32_t* tk = vc_smem;                         // [L] token of a visited vertex, SENT otherwise
    __shared__ int32_t head[256], cnt[256], carry[256];
    constexpr int32_t SENT = -12345;
    const int b = blockIdx.x, tid = threadIdx.x;
    const long pl = pred_len[b];
    const bool un = unreach[b] != 0;
    for (int j = tid; j < L; j += 256) {
        const long pv = path[(size_t)b * L + j];
        const bool on = un ? (j == 0) : (pv >= 1 && pv <= pl);
        tk[j] = on ? tok[(size_t)b * L + j] : SENT;
    }
    __syncthreads();
    const int per = (L + 255) / 256, j0 = tid * per, j1 = min(L, j0 + per);
    int32_t first = SENT;
    for (int j = j0; j < j1; ++j) if (tk[j] != SENT) { first = tk[j]; break; }
    head[tid] = first;
    __syncthreads();
    if (tid == 0) {                                // carry[c] = first visited token of the chunks behind c
        int32_t nx = SENT;
        for (int c = 255; c >= 0; --c) { carry[c] = nx; if (head[c] != SENT) nx = head[c]; }
    }
    __syncthreads();
    // backward over the chunk (the decision needs the NEXT visited token): count first, emit in a second identical walk
    int32_t nx = carry[tid];
    int kept = 0;
    for (int j = j1 - 1; j >= j0; --j) {
        const int32_t t = tk[j];
        if (t == SENT) continue;
        if (nx == SENT || (t != pad && t != nx)) ++kept;
        nx = t;
    }
    cnt[tid] = kept;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int c = 0; c < 256; ++c) { const int k = cnt[c]; cnt[c] = run; run += k; } n_keep[b] = run < cap ? run : cap; head[0] = run; }
    __syncthreads();
    const int total = head[0];
    // forward emission needs the decisions in forward order: walk backward again, writing at base + kept - 1 downwards
    int pos = cnt[tid] + kept - 1;
    nx = carry[tid];
    for (int j = j1 - 1; j >= j0; --j) {
        const int32_t t = tk[j];
        if (t == SENT) continue;
        if (nx == SENT || (t != pad && t != nx)) {
            if (pos < cap) { out_tokens[(size_t)b * cap + pos] = t; keep_idx[(size_t)b * cap + pos] = j; }
            --pos;
        }
        nx = t;
    }
    for (int e = min(total, cap) + tid; e < cap; e += 256) { out_tokens[(size_t)b * cap + e] = pad; keep_idx[(size_t)b * cap + e] = -1; }
}

extern "C" int dsp_viterbi_collect(const int64_t* path, const int64_t* pred_length, const unsigned char* unreachable, const int32_t* tok, int pad,
                                   int64_t* out_tokens, int32_t* keep_idx, int32_t* n_keep, int B, int L, int cap, dsp_stream_t stream)
{
    if (B < 0 || L < 1 || cap < 1) { set_error("viterbi_collect: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!path || !pred_length || !unreachable || !tok || !out_tokens || !keep_idx || !n_keep) { set_error("viterbi_collect: null pointer"); return DSP_EINVAL; }
    const size_t lds = (size_t)L * sizeof(int32_t);
    if (lds > 150 * 1024) { set_error("viterbi_collect: L=%d too large for the LDS token image", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)viterbi_collect_kernel, (int)lds);
    hipLaunchKernelGGL(viterbi_collect_kernel, dim3(B), dim3(256), lds, as_stream(stream), path, pred_length, unreachable, tok, pad, out_tokens, keep_idx, n_keep, L, cap);
    return check_launch("viterbi_collect");
}

extern "C" int dsp_follow_path(const int32_t* next, const int32_t* tok, const int64_t* out_len, int pad,
                               int64_t* out_tokens, int32_t* keep_idx, int32_t* n_feat, int B, int L, int cap, dsp_stream_t stream)
{
    if (B < 0 || L < 1 || cap < 1) { set_error("follow_path: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!next || !tok || !out_len || !out_tokens || !keep_idx || !n_feat) { set_error("follow_path: null pointer"); return DSP_EINVAL; }
    const size_t lds = (size_t)(2 * L + 2 * cap) * sizeof(int32_t);
    if (lds > 160 * 1024) { set_error("follow_path: L=%d / cap=%d too large for the LDS walk", L, cap); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)follow_path_kernel, (int)lds);
    hipLaunchKernelGGL(follow_path_kernel, dim3(B), dim3(256), lds, as_stream(stream), next, tok, out_len, pad, out_tokens, keep_idx, n_feat, L, cap);
    return check_launch("follow_path");
}

extern "C" int dsp_gather_rows(const void* features, int dtype, const int32_t* keep_idx, const int32_t* n_feat, void* out,
                               int B, int L, int D, int cap, int Fmax, dsp_stream_t stream)
{
    const int es = elem_size(dtype);
    if (!es || B < 0 || L < 1 || D < 1 || cap < 1 || Fmax < 0) { set_error("gather_rows: bad arguments"); return DSP_EINVAL; }
    if (B == 0 || Fmax == 0) return DSP_OK;
    if (!features || !keep_idx || !n_feat || !out) { set_error("gather_rows: null pointer"); return DSP_EINVAL; }
    hipLaunchKernelGGL(gather_rows_kernel, dim3(Fmax, B), dim3(64), 0, as_stream(stream), (const char*)features, keep_idx, n_feat,
                       (char*)out, L, (long)D * es, cap, Fmax);
    return check_launch("gather_rows");
}

extern "C" int dsp_posterior_features(const float* alpha, const float* beta, const float* features, float* out, float* lse,
                                      int B, int T, int L, int D, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || L < 1 || D < 2 || (D & 1)) { set_error("posterior_features: bad sizes (D must be even)"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!alpha || !beta || !features || !out) { set_error("posterior_features: null pointer"); return DSP_EINVAL; }
    const size_t lds = (size_t)PF_TT * L * sizeof(float);
    if (lds > 150 * 1024) { set_error("posterior_features: L=%d too large for the posterior rows in LDS", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)posterior_features_kernel, (int)lds);
    hipLaunchKernelGGL(posterior_features_kernel, dim3((T + PF_TT - 1) / PF_TT, B), dim3(256), lds, as_stream(stream), alpha, beta, features, out, lse, T, L, D);
    return check_launch("posterior_features");
}

extern "C" int dsp_posterior_features_bwd(const float* alpha, const float* beta, const float* lse, const float* grad_out, float* grad_features,
                                          int B, int T, int L, int D, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || L < 1 || D < 2 || (D & 1)) { set_error("posterior_features_bwd: bad sizes (D must be even)"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!alpha || !beta || !lse || !grad_out || !grad_features) { set_error("posterior_features_bwd: null pointer"); return DSP_EINVAL; }
    const size_t lds = (size_t)PF_TT * T * sizeof(float);
    if (lds > 150 * 1024) { set_error("posterior_features_bwd: T=%d too large for the posterior columns in LDS", T); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)posterior_features_bwd_kernel, (int)lds);
    hipLaunchKernelGGL(posterior_features_bwd_kernel, dim3((L + PF_TT - 1) / PF_TT, B), dim3(256), lds, as_stream(stream), alpha, beta, lse, grad_out, grad_features, T, L, D);
    return check_launch("posterior_features_bwd");
}

extern "C" int dsp_posterior(const float* alpha, const float* beta, float* score, int B, int T, int L, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || L < 1) { set_error("posterior: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!alpha || !beta || !score) { set_error("posterior: null pointer"); return DSP_EINVAL; }
    const long nrows = (long)B * T;
    const int grid = (int)(nrows < 4096 ? nrows : 4096);
    hipLaunchKernelGGL(posterior_kernel, dim3(grid), dim3(256), 0, as_stream(stream), alpha, beta, score, nrows, L);
    return check_launch("posterior");
}

extern "C" int dsp_durations(const float* log_dur, const uint8_t* pad_mask, float factor, int64_t* dur, int64_t n, dsp_stream_t stream)
{
    if (n < 0) { set_error("durations: bad size"); return DSP_EINVAL; }
    if (n == 0) return DSP_OK;
    if (!log_dur || !pad_mask || !dur) { set_error("durations: null pointer"); return DSP_EINVAL; }
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(durations_kernel, dim3(grid), dim3(256), 0, as_stream(stream), log_dur, pad_mask, factor, dur, (long)n);
    return check_launch("durations");
}

extern "C" int dsp_bucketize_embed_add(float* x, const float* v, const float* bins, int nb, const float* emb, int64_t n, int C,
                                       dsp_stream_t stream)
{
    if (n < 0 || nb < 0 || C < 1) { set_error("bucketize_embed_add: bad sizes"); return DSP_EINVAL; }
    if (n == 0) return DSP_OK;
    if (!x || !v || (nb && !bins) || !emb) { set_error("bucketize_embed_add: null pointer"); return DSP_EINVAL; }
    const int grid = (int)(n < 4096 ? n : 4096);
    hipLaunchKernelGGL(bucketize_embed_add_kernel, dim3(grid), dim3(C >= 256 ? 256 : 64), 0, as_stream(stream), x, v, bins, nb, emb, (long)n, C);
    return check_launch("bucketize_embed_add");
}

extern "C" int dsp_length_regulator_lens(const int64_t* dur, int64_t* cum, int64_t* out_lens, int B, int N, dsp_stream_t stream)
{
    if (B < 0 || N < 0) { set_error("length_regulator_lens: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if ((N && (!dur || !cum)) || !out_lens) { set_error("length_regulator_lens: null pointer"); return DSP_EINVAL; }
    hipLaunchKernelGGL(lr_lens_kernel, dim3(B), dim3(256), 0, as_stream(stream), dur, cum, out_lens, N);
    return check_launch("length_regulator_lens");
}

extern "C" int dsp_length_regulator_expand(const void* x, int dtype, const int64_t* cum, void* out, int B, int N, int C, int maxlen,
                                           dsp_stream_t stream)
{
    const int es = elem_size(dtype);
    if (!es || B < 0 || N < 0 || C < 1 || maxlen < 0) { set_error("length_regulator_expand: bad arguments"); return DSP_EINVAL; }
    if (B == 0 || maxlen == 0) return DSP_OK;
    if (!out || (N && (!x || !cum))) { set_error("length_regulator_expand: null pointer"); return DSP_EINVAL; }
    const size_t lds = (size_t)(N > 0 ? N : 1) * sizeof(long);
    if (lds > 160 * 1024) { set_error("length_regulator_expand: N=%d too large", N); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)lr_expand_kernel, (int)lds);
    int gx = (maxlen + 3) / 4; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(lr_expand_kernel, dim3(gx, B), dim3(256), lds, as_stream(stream), (const char*)x, cum, (char*)out, N,
                       (long)C * es, maxlen);
    return check_launch("length_regulator_expand");
}
