// extract_links.hip — the DA-Transformer's transition producer, fused, compact layout.
//
// Replaces the inference path of DAGDecoder.extract_links (DASpeech/models/s2t_conformer_dag.py:171-212):
//     content[b,i,j,h] = q[b,i,h,:] . k[b,j,h,:] / sqrt(ck)                      (:183-186: an [B,L,L,H] einsum)
//     banded:   keep j = i+1 .. i+TR (:191-196), mask j >= out_len[b] (:197), log_softmax over the kept j per head (:199),
//               rows without any successor -> -inf (:200-201)
//     links[b,i,d] = logsumexp_h(content_ls[b,i,d,h] + log_gates[b,i,h])         (:208-210)
// The reference materialises the L x L x H content tensor and gathers the band out of it; here only the band is ever
// computed: one workgroup per (sample, 4 source vertices), thread = (head h, successor j): 8 heads x 32 successors per step, soft-max
// over the slots inside the 32 lanes of a head, the log-sum-exp over heads through LDS; the per-head scores of the four vertices are
// parked in LDS (any TR, incl. README's --max-transition-length 99999: TR = L-1).
#include "common.h"
#include <atomic>

namespace dsp {

constexpr int XL_H = 8;                       // attention heads of the link predictor (fixed by the architecture: 8)
constexpr int XL_IT = 4;                      // source vertices per workgroup

// ck = head width (64 in the released model, any multiple of 4 up to 128)
// A workgroup owns XL_IT consecutive source vertices and walks the successors j ONCE for all of them: thread (h, lane) holds the
// k row of successor j in registers and takes the XL_IT dot products against the queries in LDS (broadcast reads), so a k row is
// read from L2 once per workgroup instead of once per source vertex (with TR = L-1 the one-vertex-at-a-time form moved
// L^2/2 x 2 KB per sample through L2: 0.92 ms at B=32, L=400).  Same FMA order per dot product as before.
template <int CK4>                            // head width / 4
__global__ __launch_bounds__(256) void extract_links_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, float* __restrict__ links,
    float* __restrict__ stats,                // [B,L,H,2] (window max, log of the window's sum) per head: the backward's state, or NULL
    int B, int L, int TR, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    constexpr int CK = CK4 * 4;
    const int TRp = ((TR + 31) / 32) * 32;
    float* qs = xl_smem;                       // [IT][H][CK]   the source vertices' queries
    float* sc = qs + XL_IT * XL_H * CK;        // [IT][TRp][H]  scores
    float* red = sc + (size_t)XL_IT * TRp * XL_H;   // [IT][H][2]  per-head max / log-sum
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = (int)out_len[b];
    const size_t rowstride = (size_t)XL_H * CK;
    const int i0 = blockIdx.x * XL_IT;
    const int nit = min(XL_IT, L - i0);
    for (int e = tid; e < nit * XL_H * CK; e += 256) qs[e] = q[((size_t)b * L + i0) * rowstride + e];
    for (int e = tid; e < XL_IT * TRp * XL_H; e += 256) sc[e] = NEG_INF;
    __syncthreads();
    // ---- scores: successors j = i0+1 .. i0+nit-1+TR in chunks of 32 (lane <-> j), XL_IT dot products per k row
    float mx[XL_IT];
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) mx[ii] = NEG_INF;
    const int jend = min(min(L, Lb), i0 + nit + TR);           // successors beyond the graph never score (:197)
    for (int jc = i0 + 1; jc < jend; jc += 32) {
        const int j = jc + d0;
        const bool live = j < jend;
        float4 kv[CK4];
        const float4* kr = reinterpret_cast<const float4*>(k + ((size_t)b * L + (live ? j : jc)) * rowstride + (size_t)h * CK);
#pragma unroll
        for (int c = 0; c < CK4; ++c) kv[c] = kr[c];            // unconditional (clamped row): the requests go out together
#pragma unroll
        for (int ii = 0; ii < XL_IT; ++ii) {
            const int d = j - (i0 + ii) - 1;
            const float4* qr = reinterpret_cast<const float4*>(qs + (ii * XL_H + h) * CK);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int c = 0; c < CK4; ++c) {
                const float4 qv = qr[c];
                a0 = fmaf(qv.x, kv[c].x, a0); a1 = fmaf(qv.y, kv[c].y, a1); a2 = fmaf(qv.z, kv[c].z, a2); a3 = fmaf(qv.w, kv[c].w, a3);
            }
            if (live && ii < nit && d >= 0 && d < TR) {
                float sv = ((a0 + a1) + (a2 + a3)) * scale;
                if (dist_bias) sv += dist_bias[d];
                sc[((size_t)ii * TRp + d) * XL_H + h] = sv;
                mx[ii] = fmaxf(mx[ii], sv);
            }
        }
    }
    // soft-max over the slots of head h: its 32 lanes are one half of a wave
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii)
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mx[ii] = fmaxf(mx[ii], __shfl_xor(mx[ii], o, 32));
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) {
        float sum = 0.f;
        if (mx[ii] != NEG_INF)
            for (int dc = 0; dc < TR; dc += 32) { const int d = dc + d0; if (d < TR) sum += __expf(sc[((size_t)ii * TRp + d) * XL_H + h] - mx[ii]); }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 32);
        if (d0 == 0) {
            const float ls = (mx[ii] == NEG_INF) ? 0.f : __logf(sum);
            red[(ii * XL_H + h) * 2] = mx[ii]; red[(ii * XL_H + h) * 2 + 1] = ls;
            if (stats && ii < nit) { float* st = stats + (((size_t)b * L + i0 + ii) * XL_H + h) * 2; st[0] = mx[ii]; st[1] = ls; }
        }
    }
    __syncthreads();
    // ---- links[b,i,d] = logsumexp_h(score - max_h - logsum_h + log_gate_h); every thread takes slots tid, tid+256, ...
    for (int ii = 0; ii < nit; ++ii) {
        const int i = i0 + ii;
        float gate[XL_H], mh[XL_H], lh[XL_H];
#pragma unroll
        for (int hh = 0; hh < XL_H; ++hh) {
            gate[hh] = log_gates[((size_t)b * L + i) * XL_H + hh]; mh[hh] = red[(ii * XL_H + hh) * 2]; lh[hh] = red[(ii * XL_H + hh) * 2 + 1];
        }
        for (int d = tid; d < TR; d += 256) {
            float v[XL_H], m2 = NEG_INF;
#pragma unroll
            for (int hh = 0; hh < XL_H; ++hh) {
                const float s = sc[((size_t)ii * TRp + d) * XL_H + hh];
                v[hh] = (s == NEG_INF) ? NEG_INF : ((s - mh[hh]) - lh[hh]) + gate[hh];
                m2 = fmaxf(m2, v[hh]);
            }
            float r = NEG_INF;
            if (m2 != NEG_INF) {
                float acc = 0.f;
#pragma unroll
                for (int hh = 0; hh < XL_H; ++hh) acc += __expf(v[hh] - m2);
                r = m2 + __logf(acc);
            }
            links[((size_t)b * L + i) * TR + d] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the fused link producer on the COMPACT band (training side): no [B,L,L,H] tensor exists
// in either direction.  With  ls[i,d,h] = log_softmax_d(s[i,d,h]),  links[i,d] = logsumexp_h(ls[i,d,h] + g[i,h])  and the incoming
// gradient G[i,d]:
//     A[i,d,h]  = G[i,d] * exp(ls[i,d,h] + g[i,h] - links[i,d])            (the head's share of the link)
//     dg[i,h]   = SA[i,h] = sum_d A[i,d,h]
//     ds[i,d,h] = A[i,d,h] - exp(ls[i,d,h]) * SA[i,h]                      (log-soft-max backward)
//     dq[i,h,:] = scale * sum_d ds[i,d,h] * k[i+d+1,h,:]        dk[j,h,:] = scale * sum_{i+d+1=j} ds[i,d,h] * q[i,h,:]
// Two launches of one kernel.  OWNER rows (4 per workgroup) are source vertices i for dq / dg (partners = successors j) and
// successors j for dk (TRANSPOSED: partners = sources i); the scores of the owner rows against their partners are recomputed
// as in the forward (thread = (head, partner slot), partner row in registers), ds is built in the LDS score image from the
// forward's per-row soft-max state (`stats`) and — for dk — the dg / SA the first launch wrote, and then contracted with the
// partner rows by threads (owner pair, head, 4 channels).  Nothing but q, k, the compact links / G and [B,L,H] state moves.
template <int CK4, bool TRANSPOSED>
__global__ __launch_bounds__(256) void extract_links_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, const float* __restrict__ links,
    const float* __restrict__ G, const float* __restrict__ stats, float* __restrict__ dgate /* = SA: written by the first launch */,
    float* __restrict__ dout /* dq or dk */, int B, int L, int TR, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    constexpr int CK = CK4 * 4;
    const int TRp = ((TR + 31) / 32) * 32;
    float* own = xl_smem;                          // [IT][H][CK]   the owner rows (q for dq, k for dk)
    float* sc = own + XL_IT * XL_H * CK;           // [IT][TRp][H]  scores, then ds
    float* red = sc + (size_t)XL_IT * TRp * XL_H;  // [IT][H]       SA of the owner rows (first launch)
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = min((int)out_len[b], L);
    const size_t rowstride = (size_t)XL_H * CK;
    const int o0 = blockIdx.x * XL_IT;
    const int nit = min(XL_IT, L - o0);
    const float* OWN = TRANSPOSED ? k : q;
    const float* PAR = TRANSPOSED ? q : k;
    for (int e = tid; e < nit * XL_H * CK; e += 256) own[e] = OWN[((size_t)b * L + o0) * rowstride + e];
    for (int e = tid; e < XL_IT * TRp * XL_H; e += 256) sc[e] = TRANSPOSED ? 0.f : NEG_INF;
    __syncthreads();
    // partner range: successors o0+1 .. o0+nit-1+TR (inside the graph), or sources o0-TR .. o0+nit-2
    const int pbeg = TRANSPOSED ? max(0, o0 - TR) : (o0 + 1);
    const int pend = TRANSPOSED ? min(o0 + nit - 1, Lb) : min(Lb, o0 + nit + TR);
    for (int pc = pbeg; pc < pend; pc += 32) {
        const int pp = pc + d0;
        const bool live = pp < pend;
        float4 pv[CK4];
        const float4* pr = reinterpret_cast<const float4*>(PAR + ((size_t)b * L + (live ? pp : pc)) * rowstride + (size_t)h * CK);
#pragma unroll
        for (int c = 0; c < CK4; ++c) pv[c] = pr[c];
        float st_mx = 0.f, st_ls = 0.f, st_g = 0.f, st_sa = 0.f;
        if (TRANSPOSED && live) {                   // the soft-max row is the PARTNER (source vertex pp)
            const size_t so = ((size_t)b * L + pp) * XL_H + h;
            st_mx = stats[2 * so]; st_ls = stats[2 * so + 1]; st_g = log_gates[so]; st_sa = dgate[so];
        }
#pragma unroll
        for (int oo = 0; oo < XL_IT; ++oo) {
            const int o = o0 + oo;
            const int d = TRANSPOSED ? (o - pp - 1) : (pp - o - 1);
            const float4* orow = reinterpret_cast<const float4*>(own + (oo * XL_H + h) * CK);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int c = 0; c < CK4; ++c) {
                const float4 ov = orow[c];
                a0 = fmaf(ov.x, pv[c].x, a0); a1 = fmaf(ov.y, pv[c].y, a1); a2 = fmaf(ov.z, pv[c].z, a2); a3 = fmaf(ov.w, pv[c].w, a3);
            }
            if (live && oo < nit && d >= 0 && d < TR && (!TRANSPOSED || o < Lb)) {
                float sv = ((a0 + a1) + (a2 + a3)) * scale;
                if (dist_bias) sv += dist_bias[d];
                if (!TRANSPOSED) sc[((size_t)oo * TRp + d) * XL_H + h] = sv;
                else {
                    const size_t lo = ((size_t)b * L + pp) * TR + d;
                    const float lk = links[lo], ls = (sv - st_mx) - st_ls;
                    const float A = (lk == NEG_INF || st_mx == NEG_INF) ? 0.f : G[lo] * __expf(ls + st_g - lk);
                    sc[((size_t)oo * TRp + d) * XL_H + h] = (st_mx == NEG_INF) ? 0.f : (A - __expf(ls) * st_sa);
                }
            }
        }
    }
    __syncthreads();
    if (!TRANSPOSED) {
        // ---- SA of the owner rows, then ds in place
        for (int oo = 0; oo < nit; ++oo) {
            const int i = o0 + oo;
            const size_t so = ((size_t)b * L + i) * XL_H + h;
            const float mxv = stats[2 * so], lsv = stats[2 * so + 1], gv = log_gates[so];
            float sa = 0.f;
            if (mxv != NEG_INF)
                for (int dc = 0; dc < TR; dc += 32) {
                    const int d = dc + d0;
                    if (d < TR) {
                        const float sv = sc[((size_t)oo * TRp + d) * XL_H + h];
                        const size_t lo = ((size_t)b * L + i) * TR + d;
                        const float lk = links[lo];
                        if (sv != NEG_INF && lk != NEG_INF) sa += G[lo] * __expf(((sv - mxv) - lsv) + gv - lk);
                    }
                }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) sa += __shfl_xor(sa, o, 32);
            if (d0 == 0) { red[oo * XL_H + h] = sa; dgate[so] = sa; }
            for (int dc = 0; dc < TRp; dc += 32) {
                const int d = dc + d0;
                const float sv = sc[((size_t)oo * TRp + d) * XL_H + h];
                float dsv = 0.f;
                if (d < TR && sv != NEG_INF && mxv != NEG_INF) {
                    const size_t lo = ((size_t)b * L + i) * TR + d;
                    const float lk = links[lo], ls = (sv - mxv) - lsv;
                    const float A = (lk == NEG_INF) ? 0.f : G[lo] * __expf(ls + gv - lk);
                    dsv = A - __expf(ls) * sa;
                }
                sc[((size_t)oo * TRp + d) * XL_H + h] = dsv;
            }
        }
        for (int oo = nit; oo < XL_IT; ++oo)
            for (int dc = 0; dc < TRp; dc += 32) sc[((size_t)oo * TRp + dc + d0) * XL_H + h] = 0.f;
        __syncthreads();
    }
    // ---- contraction with the partner rows: thread = (owner pair, head, 4 channels)
    {
        const int c4 = tid & (CK4 - 1), hh = (tid / CK4) % XL_H, op = tid / (CK4 * XL_H);       // CK4 * 8 * 2 = 256 for CK = 64
        constexpr int NOP = 256 / (CK4 * XL_H);                                                 // owner groups per pass (2 for CK 64, 4 for CK 32, 1 for CK 128)
        constexpr int OPG = XL_IT / NOP;                                                        // owner rows per thread
        float4 acc[OPG];
#pragma unroll
        for (int x = 0; x < OPG; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int pp = pbeg; pp < pend; ++pp) {
            const float4 pv = *reinterpret_cast<const float4*>(PAR + ((size_t)b * L + pp) * rowstride + (size_t)hh * CK + c4 * 4);
#pragma unroll
            for (int x = 0; x < OPG; ++x) {
                const int oo = op * OPG + x;
                const int d = TRANSPOSED ? (o0 + oo - pp - 1) : (pp - o0 - oo - 1);
                const float dsv = (d >= 0 && d < TR) ? sc[((size_t)oo * TRp + d) * XL_H + hh] : 0.f;
                acc[x].x = fmaf(dsv, pv.x, acc[x].x); acc[x].y = fmaf(dsv, pv.y, acc[x].y);
                acc[x].z = fmaf(dsv, pv.z, acc[x].z); acc[x].w = fmaf(dsv, pv.w, acc[x].w);
            }
        }
#pragma unroll
        for (int x = 0; x < OPG; ++x) {
            const int oo = op * OPG + x;
            if (oo < nit)
                *reinterpret_cast<float4*>(dout + ((size_t)b * L + o0 + oo) * rowstride + (size_t)hh * CK + c4 * 4) =
                    make_float4(acc[x].x * scale, acc[x].y * scale, acc[x].z * scale, acc[x].w * scale);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// r05 — windows whose score image does not fit LDS (TR above ~1100 at 8 heads x 64: the README's --max-transition-length 99999 on graphs
// longer than ~1100 vertices, up to BASELINE's L = 4096).  Same arithmetic, but the partner range is walked in TILES of TW slots:
//   forward:   pass 1 streams every successor once and keeps an online (max, sum) per (vertex, head) — the soft-max state; pass 2
//              recomputes the scores tile by tile into a [IT][TW][H] image and emits the tile's links (the dot products run twice,
//              nothing of size TR is ever held on chip);
//   backward:  (dq, dgate) pass 1 accumulates SA = sum_d A the same streaming way, pass 2 builds ds per tile and contracts it with the
//              tile's partner rows into accumulators that live across the tiles; (dk) needs no pre-pass (its soft-max rows are the
//              partners, whose state the first launch left in `stats` / `dgate`).
// With TW >= the partner range these kernels compute exactly what the one-image kernels above compute (tests force small tiles on small
// graphs and compare the two, and both with torch autograd).
template <int CK4>
__device__ __forceinline__ float xl_dot(const float* __restrict__ qrow, const float4 (&kv)[CK4])
{
    const float4* qr = reinterpret_cast<const float4*>(qrow);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < CK4; ++c) {
        const float4 qv = qr[c];
        a0 = fmaf(qv.x, kv[c].x, a0); a1 = fmaf(qv.y, kv[c].y, a1); a2 = fmaf(qv.z, kv[c].z, a2); a3 = fmaf(qv.w, kv[c].w, a3);
    }
    return (a0 + a1) + (a2 + a3);
}

template <int CK4>
__global__ __launch_bounds__(256) void extract_links_tiled_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, float* __restrict__ links,
    float* __restrict__ stats, int B, int L, int TR, float scale, int TW)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    constexpr int CK = CK4 * 4;
    float* qs = xl_smem;                       // [IT][H][CK]
    float* sc = qs + XL_IT * XL_H * CK;        // [IT][TW][H]   scores of the current tile, slot = successor - tile start
    float* red = sc + (size_t)XL_IT * TW * XL_H;    // [IT][H][2]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = (int)out_len[b];
    const size_t rowstride = (size_t)XL_H * CK;
    const int i0 = blockIdx.x * XL_IT;
    const int nit = min(XL_IT, L - i0);
    for (int e = tid; e < nit * XL_H * CK; e += 256) qs[e] = q[((size_t)b * L + i0) * rowstride + e];
    __syncthreads();
    const int jend = min(min(L, Lb), i0 + nit + TR);
    // score of (owner ii, successor j) — the SAME expression in both passes, so pass 2 reproduces pass 1's values bit for bit
    auto score = [&](int ii, int j, const float4 (&kv)[CK4], float& sv) -> bool {
        const int d = j - (i0 + ii) - 1;
        const float dot = xl_dot<CK4>(qs + (ii * XL_H + h) * CK, kv);
        if (!(ii < nit && d >= 0 && d < TR)) return false;
        sv = dot * scale;
        if (dist_bias) sv += dist_bias[d];
        return true;
    };
    // ---- pass 1: online soft-max state per (owner, head) over all successors
    float m[XL_IT], sm[XL_IT];
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) { m[ii] = NEG_INF; sm[ii] = 0.f; }
    for (int jc = i0 + 1; jc < jend; jc += 32) {
        const int j = jc + d0;
        const bool live = j < jend;
        float4 kv[CK4];
        const float4* kr = reinterpret_cast<const float4*>(k + ((size_t)b * L + (live ? j : jc)) * rowstride + (size_t)h * CK);
#pragma unroll
        for (int c = 0; c < CK4; ++c) kv[c] = kr[c];
#pragma unroll
        for (int ii = 0; ii < XL_IT; ++ii) {
            float sv;
            if (score(ii, j, kv, sv) && live) {
                if (sv > m[ii]) { sm[ii] = sm[ii] * __expf(m[ii] - sv) + 1.f; m[ii] = sv; }       // (m = -inf: exp(-inf) = 0)
                else sm[ii] += __expf(sv - m[ii]);
            }
        }
    }
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) {
        float M = m[ii];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 32));
        float S = (m[ii] == NEG_INF) ? 0.f : sm[ii] * __expf(m[ii] - M);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) S += __shfl_xor(S, o, 32);
        if (d0 == 0) {
            const float ls = (M == NEG_INF) ? 0.f : __logf(S);
            red[(ii * XL_H + h) * 2] = M; red[(ii * XL_H + h) * 2 + 1] = ls;
            if (stats && ii < nit) { float* st = stats + (((size_t)b * L + i0 + ii) * XL_H + h) * 2; st[0] = M; st[1] = ls; }
        }
    }
    __syncthreads();
    // ---- pass 2: tiles of successors
    for (int jt = i0 + 1; jt < jend; jt += TW) {
        const int jte = min(jt + TW, jend);
        for (int e = tid; e < XL_IT * TW * XL_H; e += 256) sc[e] = NEG_INF;
        __syncthreads();
        for (int jc = jt; jc < jte; jc += 32) {
            const int j = jc + d0;
            const bool live = j < jte;
            float4 kv[CK4];
            const float4* kr = reinterpret_cast<const float4*>(k + ((size_t)b * L + (live ? j : jc)) * rowstride + (size_t)h * CK);
#pragma unroll
            for (int c = 0; c < CK4; ++c) kv[c] = kr[c];
#pragma unroll
            for (int ii = 0; ii < XL_IT; ++ii) {
                float sv;
                if (score(ii, j, kv, sv) && live) sc[((size_t)ii * TW + (j - jt)) * XL_H + h] = sv;
            }
        }
        __syncthreads();
        for (int ii = 0; ii < nit; ++ii) {
            const int i = i0 + ii;
            float gate[XL_H], mh[XL_H], lh[XL_H];
#pragma unroll
            for (int hh = 0; hh < XL_H; ++hh) {
                gate[hh] = log_gates[((size_t)b * L + i) * XL_H + hh]; mh[hh] = red[(ii * XL_H + hh) * 2]; lh[hh] = red[(ii * XL_H + hh) * 2 + 1];
            }
            for (int js = tid; js < jte - jt; js += 256) {
                const int d = jt + js - i - 1;
                if (d < 0 || d >= TR) continue;
                float v[XL_H], m2 = NEG_INF;
#pragma unroll
                for (int hh = 0; hh < XL_H; ++hh) {
                    const float sx = sc[((size_t)ii * TW + js) * XL_H + hh];
                    v[hh] = (sx == NEG_INF) ? NEG_INF : ((sx - mh[hh]) - lh[hh]) + gate[hh];
                    m2 = fmaxf(m2, v[hh]);
                }
                float r = NEG_INF;
                if (m2 != NEG_INF) {
                    float acc = 0.f;
#pragma unroll
                    for (int hh = 0; hh < XL_H; ++hh) acc += __expf(v[hh] - m2);
                    r = m2 + __logf(acc);
                }
                links[((size_t)b * L + i) * TR + d] = r;
            }
        }
        __syncthreads();
    }
    // ---- slots without a successor inside the graph
    for (int ii = 0; ii < nit; ++ii) {
        const int i = i0 + ii;
        for (int d = max(0, jend - i - 1) + tid; d < TR; d += 256) links[((size_t)b * L + i) * TR + d] = NEG_INF;
    }
}

template <int CK4, bool TRANSPOSED>
__global__ __launch_bounds__(256) void extract_links_bwd_tiled_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, const float* __restrict__ links,
    const float* __restrict__ G, const float* __restrict__ stats, float* __restrict__ dgate,
    float* __restrict__ dout, int B, int L, int TR, float scale, int TW)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    constexpr int CK = CK4 * 4;
    float* own = xl_smem;                          // [IT][H][CK]
    float* sc = own + XL_IT * XL_H * CK;           // [IT][TW][H]   ds of the current tile, slot = partner - tile start
    float* red = sc + (size_t)XL_IT * TW * XL_H;   // [IT][H]       SA of the owner rows (first launch)
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = min((int)out_len[b], L);
    const size_t rowstride = (size_t)XL_H * CK;
    const int o0 = blockIdx.x * XL_IT;
    const int nit = min(XL_IT, L - o0);
    const float* OWN = TRANSPOSED ? k : q;
    const float* PAR = TRANSPOSED ? q : k;
    for (int e = tid; e < nit * XL_H * CK; e += 256) own[e] = OWN[((size_t)b * L + o0) * rowstride + e];
    __syncthreads();
    const int pbeg = TRANSPOSED ? max(0, o0 - TR) : (o0 + 1);
    const int pend = TRANSPOSED ? min(o0 + nit - 1, Lb) : min(Lb, o0 + nit + TR);
    // soft-max state of the owner rows (dq launch): per (owner, head), this thread's head
    float o_mx[XL_IT], o_ls[XL_IT], o_g[XL_IT];
    if (!TRANSPOSED) {
#pragma unroll
        for (int oo = 0; oo < XL_IT; ++oo) {
            const size_t so = ((size_t)b * L + min(o0 + oo, L - 1)) * XL_H + h;
            o_mx[oo] = stats[2 * so]; o_ls[oo] = stats[2 * so + 1]; o_g[oo] = log_gates[so];
        }
        // ---- pass 1: SA[owner][head] = sum_d A
        float sa[XL_IT];
#pragma unroll
        for (int oo = 0; oo < XL_IT; ++oo) sa[oo] = 0.f;
        for (int pc = pbeg; pc < pend; pc += 32) {
            const int pp = pc + d0;
            const bool live = pp < pend;
            float4 pv[CK4];
            const float4* pr = reinterpret_cast<const float4*>(PAR + ((size_t)b * L + (live ? pp : pc)) * rowstride + (size_t)h * CK);
#pragma unroll
            for (int c = 0; c < CK4; ++c) pv[c] = pr[c];
#pragma unroll
            for (int oo = 0; oo < XL_IT; ++oo) {
                const int o = o0 + oo, d = pp - o - 1;
                const float dot = xl_dot<CK4>(own + (oo * XL_H + h) * CK, pv);
                if (live && oo < nit && d >= 0 && d < TR && o_mx[oo] != NEG_INF) {
                    float sv = dot * scale;
                    if (dist_bias) sv += dist_bias[d];
                    const size_t lo = ((size_t)b * L + o) * TR + d;
                    const float lk = links[lo];
                    if (lk != NEG_INF) sa[oo] += G[lo] * __expf(((sv - o_mx[oo]) - o_ls[oo]) + o_g[oo] - lk);
                }
            }
        }
#pragma unroll
        for (int oo = 0; oo < XL_IT; ++oo) {
            float v = sa[oo];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 32);
            if (d0 == 0) { red[oo * XL_H + h] = v; if (oo < nit) dgate[((size_t)b * L + o0 + oo) * XL_H + h] = v; }
        }
        __syncthreads();
    }
    const int c4 = tid & (CK4 - 1), hh = (tid / CK4) % XL_H, op = tid / (CK4 * XL_H);
    constexpr int NOP = 256 / (CK4 * XL_H);
    constexpr int OPG = XL_IT / NOP;
    float4 acc[OPG];
#pragma unroll
    for (int x = 0; x < OPG; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pt = pbeg; pt < pend; pt += TW) {
        const int pte = min(pt + TW, pend);
        for (int e = tid; e < XL_IT * TW * XL_H; e += 256) sc[e] = 0.f;
        __syncthreads();
        for (int pc = pt; pc < pte; pc += 32) {
            const int pp = pc + d0;
            const bool live = pp < pte;
            float4 pv[CK4];
            const float4* pr = reinterpret_cast<const float4*>(PAR + ((size_t)b * L + (live ? pp : pc)) * rowstride + (size_t)h * CK);
#pragma unroll
            for (int c = 0; c < CK4; ++c) pv[c] = pr[c];
            float st_mx = 0.f, st_ls = 0.f, st_g = 0.f, st_sa = 0.f;
            if (TRANSPOSED && live) {
                const size_t so = ((size_t)b * L + pp) * XL_H + h;
                st_mx = stats[2 * so]; st_ls = stats[2 * so + 1]; st_g = log_gates[so]; st_sa = dgate[so];
            }
#pragma unroll
            for (int oo = 0; oo < XL_IT; ++oo) {
                const int o = o0 + oo;
                const int d = TRANSPOSED ? (o - pp - 1) : (pp - o - 1);
                const float dot = xl_dot<CK4>(own + (oo * XL_H + h) * CK, pv);
                if (live && oo < nit && d >= 0 && d < TR && (!TRANSPOSED || o < Lb)) {
                    float sv = dot * scale;
                    if (dist_bias) sv += dist_bias[d];
                    const float mxv = TRANSPOSED ? st_mx : o_mx[oo], lsv = TRANSPOSED ? st_ls : o_ls[oo], gv = TRANSPOSED ? st_g : o_g[oo];
                    const float sav = TRANSPOSED ? st_sa : red[oo * XL_H + h];
                    const size_t lo = TRANSPOSED ? (((size_t)b * L + pp) * TR + d) : (((size_t)b * L + o) * TR + d);
                    float dsv = 0.f;
                    if (mxv != NEG_INF) {
                        const float lk = links[lo], ls = (sv - mxv) - lsv;
                        const float A = (lk == NEG_INF) ? 0.f : G[lo] * __expf(ls + gv - lk);
                        dsv = A - __expf(ls) * sav;
                    }
                    sc[((size_t)oo * TW + (pp - pt)) * XL_H + h] = dsv;
                }
            }
        }
        __syncthreads();
        for (int pp = pt; pp < pte; ++pp) {
            const float4 pv = *reinterpret_cast<const float4*>(PAR + ((size_t)b * L + pp) * rowstride + (size_t)hh * CK + c4 * 4);
#pragma unroll
            for (int x = 0; x < OPG; ++x) {
                const int oo = op * OPG + x;
                const int d = TRANSPOSED ? (o0 + oo - pp - 1) : (pp - o0 - oo - 1);
                const float dsv = (d >= 0 && d < TR) ? sc[((size_t)oo * TW + (pp - pt)) * XL_H + hh] : 0.f;
                acc[x].x = fmaf(dsv, pv.x, acc[x].x); acc[x].y = fmaf(dsv, pv.y, acc[x].y);
                acc[x].z = fmaf(dsv, pv.z, acc[x].z); acc[x].w = fmaf(dsv, pv.w, acc[x].w);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < OPG; ++x) {
        const int oo = op * OPG + x;
        if (oo < nit)
            *reinterpret_cast<float4*>(dout + ((size_t)b * L + o0 + oo) * rowstride + (size_t)hh * CK + c4 * 4) =
                make_float4(acc[x].x * scale, acc[x].y * scale, acc[x].z * scale, acc[x].w * scale);
    }
}

// PROCESS-wide (r06): the backward of an autograd Function runs on PyTorch's device worker thread, which never saw a thread_local pin set by the
// test's main thread (the "forced" tiled / matrix-core backward never ran).  g_xl_ran collects the kernel families launched since the
// last dsp_extract_links_debug_ran() so a test can assert its pin took: bit 0 one-image forward, 1 tiled forward, 2 matrix-core forward,
// 3 one-image backward, 4 tiled backward, 5 matrix-core backward (exact-fp32 contraction), 6 matrix-core backward (bf16-triple contraction).
static std::atomic<int> g_xl_tile{0};           // dsp_dag_set_option("xl_tile", n): n > 0 forces the tiled kernels with TW = n (tests); 0 = only where the image does not fit
std::atomic<unsigned int> g_xl_ran{0};
void set_xl_tile(int v) { g_xl_tile = v > 0 ? ((v + 31) / 32) * 32 : 0; }
}  // namespace dsp

static int xl_check(const char* fn, const void* q, const void* k, const void* g, const void* ol, const void* links, int B, int L, int H, int CK, int TR, size_t* lds)
{
    using namespace dsp;
    if (B < 0 || L < 1 || TR < 1 || CK < 4) { set_error("%s: bad sizes B=%d L=%d TR=%d CK=%d", fn, B, L, TR, CK); return DSP_EINVAL; }
    if (H != XL_H) { set_error("%s: needs %d heads (got H=%d)", fn, XL_H, H); return DSP_EINVAL; }
    if (B > 0 && (!q || !k || !g || !ol || !links)) { set_error("%s: null pointer", fn); return DSP_EINVAL; }
    if ((((uintptr_t)q) | ((uintptr_t)k)) & 15) { set_error("%s: q / k must be 16-byte aligned", fn); return DSP_EINVAL; }
    *lds = ((size_t)XL_IT * XL_H * CK + (size_t)XL_IT * ((TR + 31) / 32) * 32 * XL_H + 2 * XL_IT * XL_H) * sizeof(float);
    if (!(CK == 32 || CK == 64 || CK == 128)) { set_error("%s: head width %d (32, 64 or 128)", fn, CK); return DSP_EINVAL; }
    return DSP_OK;
}

// tile width of the tiled kernels for this call, 0 = the one-image kernels serve it
static int xl_tile_width(size_t lds_one_image)
{
    if (const int forced = dsp::g_xl_tile.load()) return forced;
    return lds_one_image > 150 * 1024 ? 512 : 0;              // 512 slots: a 64 KB image, two workgroups per CU
}
static size_t xl_tiled_lds(int CK, int TW) { return ((size_t)dsp::XL_IT * dsp::XL_H * CK + (size_t)dsp::XL_IT * TW * dsp::XL_H + 2 * dsp::XL_IT * dsp::XL_H) * sizeof(float); }

extern "C" int dsp_extract_links_train(const float* q, const float* k, const float* log_gates, const int64_t* out_len,
                                       const float* dist_bias, float* links, float* stats, int B, int L, int H, int CK, int TR, float scale,
                                       dsp_stream_t stream)
{
    using namespace dsp;
    size_t lds;
    int rc = xl_check("extract_links_train", q, k, log_gates, out_len, links, B, L, H, CK, TR, &lds);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!stats) { set_error("extract_links_train: null stats"); return DSP_EINVAL; }
    if (const int TW = xl_tile_width(lds)) {
        const size_t l2 = xl_tiled_lds(CK, TW);
        auto kt = CK == 64 ? extract_links_tiled_kernel<16> : (CK == 32 ? extract_links_tiled_kernel<8> : extract_links_tiled_kernel<32>);
        if (l2 > 48 * 1024) set_max_dynamic_lds((const void*)kt, (int)l2);
        hipLaunchKernelGGL(kt, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), l2, as_stream(stream), q, k, log_gates, out_len, dist_bias, links, stats, B, L, TR, scale, TW);
        g_xl_ran |= 2u;
        return check_launch("extract_links_train(tiled)");
    }
    auto kern = CK == 64 ? extract_links_kernel<16> : (CK == 32 ? extract_links_kernel<8> : extract_links_kernel<32>);
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)kern, (int)lds);
    hipLaunchKernelGGL(kern, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), lds, as_stream(stream),
                       q, k, log_gates, out_len, dist_bias, links, stats, B, L, TR, scale);
    g_xl_ran |= 1u;
    return check_launch("extract_links_train");
}

template <int CK4>
static int xl_bwd_launch(const float* q, const float* k, const float* g, const int64_t* ol, const float* bias, const float* links, const float* G,
                         const float* stats, float* dq, float* dk, float* dg, int B, int L, int TR, float scale, size_t lds, hipStream_t st)
{
    using namespace dsp;
    auto ka = extract_links_bwd_kernel<CK4, false>;
    auto kb = extract_links_bwd_kernel<CK4, true>;
    if (lds > 48 * 1024) {
        set_max_dynamic_lds((const void*)ka, (int)lds);
        set_max_dynamic_lds((const void*)kb, (int)lds);
    }
    const dim3 grid((L + XL_IT - 1) / XL_IT, B);
    hipLaunchKernelGGL(ka, grid, dim3(256), lds, st, q, k, g, ol, bias, links, G, stats, dg, dq, B, L, TR, scale);
    int rc = check_launch("extract_links_bwd(dq, dgate)");
    if (rc) return rc;
    hipLaunchKernelGGL(kb, grid, dim3(256), lds, st, q, k, g, ol, bias, links, G, stats, dg, dk, B, L, TR, scale);
    return check_launch("extract_links_bwd(dk)");
}

extern "C" int dsp_extract_links_bwd(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                                     const float* links, const float* grad_links, const float* stats,
                                     float* grad_q, float* grad_k, float* grad_log_gates, int B, int L, int H, int CK, int TR, float scale,
                                     dsp_stream_t stream)
{
    using namespace dsp;
    size_t lds;
    int rc = xl_check("extract_links_bwd", q, k, log_gates, out_len, links, B, L, H, CK, TR, &lds);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!grad_links || !stats || !grad_q || !grad_k || !grad_log_gates) { set_error("extract_links_bwd: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)grad_q) | ((uintptr_t)grad_k)) & 15) { set_error("extract_links_bwd: grad_q / grad_k must be 16-byte aligned"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (const int TW = xl_tile_width(lds)) {
        g_xl_ran |= 16u;
        const size_t l2 = xl_tiled_lds(CK, TW);
        const dim3 grid((L + XL_IT - 1) / XL_IT, B);
        auto go = [&](auto ka, auto kb) -> int {
            if (l2 > 48 * 1024) {
                set_max_dynamic_lds((const void*)ka, (int)l2);
                set_max_dynamic_lds((const void*)kb, (int)l2);
            }
            hipLaunchKernelGGL(ka, grid, dim3(256), l2, st, q, k, log_gates, out_len, dist_bias, links, grad_links, stats, grad_log_gates, grad_q, B, L, TR, scale, TW);
            int r = check_launch("extract_links_bwd(tiled: dq, dgate)");
            if (r) return r;
            hipLaunchKernelGGL(kb, grid, dim3(256), l2, st, q, k, log_gates, out_len, dist_bias, links, grad_links, stats, grad_log_gates, grad_k, B, L, TR, scale, TW);
            return check_launch("extract_links_bwd(tiled: dk)");
        };
        if (CK == 64) return go(extract_links_bwd_tiled_kernel<16, false>, extract_links_bwd_tiled_kernel<16, true>);
        if (CK == 32) return go(extract_links_bwd_tiled_kernel<8, false>, extract_links_bwd_tiled_kernel<8, true>);
        return go(extract_links_bwd_tiled_kernel<32, false>, extract_links_bwd_tiled_kernel<32, true>);
    }
    g_xl_ran |= 8u;
    if (CK == 64) return xl_bwd_launch<16>(q, k, log_gates, out_len, dist_bias, links, grad_links, stats, grad_q, grad_k, grad_log_gates, B, L, TR, scale, lds, st);
    if (CK == 32) return xl_bwd_launch<8>(q, k, log_gates, out_len, dist_bias, links, grad_links, stats, grad_q, grad_k, grad_log_gates, B, L, TR, scale, lds, st);
    return xl_bwd_launch<32>(q, k, log_gates, out_len, dist_bias, links, grad_links, stats, grad_q, grad_k, grad_log_gates, B, L, TR, scale, lds, st);
}

extern "C" int dsp_extract_links(const float* q, const float* k, const float* log_gates, const int64_t* out_len,
                                 const float* dist_bias, float* links, int B, int L, int H, int CK, int TR, float scale,
                                 dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 1 || TR < 1 || CK < 4) { set_error("extract_links: bad sizes B=%d L=%d TR=%d CK=%d", B, L, TR, CK); return DSP_EINVAL; }
    if (H != XL_H) { set_error("extract_links: needs %d heads (got H=%d)", XL_H, H); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !log_gates || !out_len || !links) { set_error("extract_links: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)q) | ((uintptr_t)k)) & 15) { set_error("extract_links: q / k must be 16-byte aligned"); return DSP_EINVAL; }
    const size_t lds = ((size_t)XL_IT * XL_H * CK + (size_t)XL_IT * ((TR + 31) / 32) * 32 * XL_H + 2 * XL_IT * XL_H) * sizeof(float);
    if (!(CK == 32 || CK == 64 || CK == 128)) { set_error("extract_links: head width %d (32, 64 or 128)", CK); return DSP_EINVAL; }
    if (const int TW = xl_tile_width(lds)) {
        const size_t l2 = xl_tiled_lds(CK, TW);
        auto kt = CK == 64 ? extract_links_tiled_kernel<16> : (CK == 32 ? extract_links_tiled_kernel<8> : extract_links_tiled_kernel<32>);
        if (l2 > 48 * 1024) set_max_dynamic_lds((const void*)kt, (int)l2);
        hipLaunchKernelGGL(kt, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), l2, as_stream(stream), q, k, log_gates, out_len, dist_bias, links, (float*)nullptr, B, L, TR, scale, TW);
        g_xl_ran |= 2u;
        return check_launch("extract_links(tiled)");
    }
    auto kern = CK == 64 ? extract_links_kernel<16> : (CK == 32 ? extract_links_kernel<8> : extract_links_kernel<32>);
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)kern, (int)lds);
    hipLaunchKernelGGL(kern, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), lds, as_stream(stream),
                       q, k, log_gates, out_len, dist_bias, links, (float*)nullptr, B, L, TR, scale);
    g_xl_ran |= 1u;
    return check_launch("extract_links");
}

extern "C" unsigned int dsp_extract_links_debug_ran(void) { return dsp::g_xl_ran.exchange(0u); }
