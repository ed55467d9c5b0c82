// dag_grad.hip — K4 (grad wrt match_all) and K5 (grad wrt links) for gfx950, generic log-space form.
// Replaces calculate_grad_match_all_kernel (dag_loss.cu:378-401) and calculate_grad_links_kernel (:432-485).
#include "common.h"
#include <atomic>

namespace dsp {

// K4: pure elementwise, HBM-bound: 3 reads + 1 write per cell, float4 wide.
__global__ __launch_bounds__(256) void dag_grad_match_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ match, float* __restrict__ g_match, int B, size_t TL)
{
    const int b = blockIdx.y;
    const float b00 = beta[(size_t)b * TL];
    const float go = g_out[b];
    const bool dead = isinf(b00);
    const float* A = alpha + (size_t)b * TL; const float* Bt = beta + (size_t)b * TL;
    const float* M = match + (size_t)b * TL; float* G = g_match + (size_t)b * TL;
    const size_t n4 = TL / 4;
    const bool al = ((((uintptr_t)A) | ((uintptr_t)Bt) | ((uintptr_t)M) | ((uintptr_t)G)) & 15) == 0;
    if (al) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 a = reinterpret_cast<const float4*>(A)[i], be = reinterpret_cast<const float4*>(Bt)[i];
            float4 m = reinterpret_cast<const float4*>(M)[i], r;
            r.x = (dead || isinf(m.x)) ? 0.f : __expf(a.x + be.x - m.x - b00) * go;      // dag_loss.cu:394-398
            r.y = (dead || isinf(m.y)) ? 0.f : __expf(a.y + be.y - m.y - b00) * go;
            r.z = (dead || isinf(m.z)) ? 0.f : __expf(a.z + be.z - m.z - b00) * go;
            r.w = (dead || isinf(m.w)) ? 0.f : __expf(a.w + be.w - m.w - b00) * go;
            reinterpret_cast<float4*>(G)[i] = r;
        }
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TL; i += (size_t)gridDim.x * blockDim.x)
            G[i] = (dead || isinf(M[i])) ? 0.f : __expf(A[i] + Bt[i] - M[i] - b00) * go;
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TL; i += (size_t)gridDim.x * blockDim.x)
            G[i] = (dead || isinf(M[i])) ? 0.f : __expf(A[i] + Bt[i] - M[i] - b00) * go;
    }
}

// K5 generic: thread (dx = d, iy = i) sums over t.  alpha[t,i] is a broadcast within the 32 d-lanes,
// beta[t+1, i+d+1] is contiguous across them.
__global__ __launch_bounds__(256) void dag_grad_links_generic_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR)
{
    const int b = blockIdx.z;
    const int d = blockIdx.x * 32 + (threadIdx.x & 31);
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= L || d >= TR) return;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float b00 = beta[(size_t)b * TL];
    float* out = g_links + ((size_t)b * L + i) * TR + d;
    const int nx = i + d + 1;
    if (i >= Lb || nx >= Lb || isinf(b00) || Tb > T || Lb > L) { *out = 0.f; return; }     // dag_loss.cu:461-466 (+ zeros init :541)
    const float* A = alpha + (size_t)b * TL + i;
    const float* Bt = beta + (size_t)b * TL + L + nx;
    const float extra = links[((size_t)b * L + i) * TR + d] - b00;                          // :469
    float acc = 0.f;
    for (int t = 0; t + 1 < Tb; ++t)                                                        // :471-475
        acc += __expf(A[(size_t)t * L] + Bt[(size_t)t * L] + extra);
    *out = acc * g_out[b];
}

// K5 tiled (TR-agnostic, used for every TR): one workgroup = 64 source vertices x 32 transition slots of one sample.
// The reference walks alpha[t][i] / beta[t+1][i+d+1] down the t axis with stride-L loads per thread
// (dag_loss.cu:471-475); here row tiles of alpha (64 wide) and beta (96 wide) are staged through LDS with coalesced
// loads (pre-scaled to the log2 domain), so HBM/L2 see each alpha row once and each beta row 1.5x per 64-vertex block,
// and the inner loop is LDS reads + v_exp_f32 only.
constexpr int K5_TC = 32;
__global__ __launch_bounds__(256) void dag_grad_links_tiled_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR)
{
    __shared__ float At[K5_TC][64];
    __shared__ float Bt[K5_TC][96];
    constexpr float LOG2E = 1.4426950408889634f;
    const int b = blockIdx.z;
    const int i0 = blockIdx.x * 64, dc0 = blockIdx.y * 32;
    const int tid = threadIdx.x, i = tid & 63, dg = tid >> 6;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float* A = alpha + (size_t)b * TL;
    const float* Bp = beta + (size_t)b * TL;
    const float b00 = Bp[0];
    const bool dead = isinf(b00) || Tb > T || Lb > L || Tb < 1 || Lb < 1;
    const int vi = i0 + i;
    float extra[8], acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int d = dc0 + dg * 8 + u;
        const bool ok = !dead && d < TR && vi < Lb && vi + d + 1 < Lb;                 // dag_loss.cu:461-466
        extra[u] = ok ? (links[((size_t)b * L + vi) * TR + d] - b00) * LOG2E : NEG_INF;   // :469
        acc[u] = 0.f;
    }
    const int nt = dead ? 0 : (Tb - 1);                    // t = 0 .. T_b-2   (:471-475)
    const int bcol0 = i0 + dc0 + 1;                        // first beta column of the tile
    for (int t0 = 0; t0 < nt; t0 += K5_TC) {
        const int rows = min(K5_TC, nt - t0);
        for (int e = tid; e < K5_TC * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            float v = NEG_INF;
            if (r < rows && i0 + c < L) v = A[(size_t)(t0 + r) * L + i0 + c] * LOG2E;
            At[r][c] = v;
        }
        for (int e = tid; e < K5_TC * 96; e += 256) {
            const int r = e / 96, c = e - r * 96;
            float v = NEG_INF;
            if (r < rows && bcol0 + c < L) v = Bp[(size_t)(t0 + r + 1) * L + bcol0 + c] * LOG2E;
            Bt[r][c] = v;
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const float a = At[r][i];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] += __builtin_amdgcn_exp2f(a + Bt[r][i + dg * 8 + u] + extra[u]);
        }
        __syncthreads();
    }
    if (vi < L) {
        const float go = g_out[b];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int d = dc0 + dg * 8 + u;
            if (d < TR) g_links[((size_t)b * L + vi) * TR + d] = (extra[u] == NEG_INF) ? 0.f : acc[u] * go;
        }
    }
}

// K5 for banded graphs (TR <= 32) in EXP SPACE.
// grad_links[i][d] = go * sum_t exp(alpha[t][i] + beta[t+1][i+d+1] + link[i][d] - beta[0][0])       (dag_loss.cu:461-475)
// The log-space kernels spend one v_exp_f32 (a quarter-rate instruction) per (t, i, d) term: 2.1e9 of them at C2.  Here the term is
// factored as   W[t+1][j] * Ga[t][i] * Elink[i][d]   with
//     W     = 2^(beta2[t+1][j] - R)        R = the lane's reference: the largest group exponent of its 36-value beta window
//     Ga    = 2^(alpha2[t][i] - b00_2 + R) one v_exp per vertex and row
//     Elink = 2^(link2[i][d])              applied once, after the sum over t
// so the inner loop is one FMA per term.  beta rows are converted once per (row, 4-vertex group) into (value, group exponent)
// pairs in LDS, as in dag_dp_strip4g.hip; there is no dependency between rows, so GX_TC rows are converted and consumed per
// pass and each of the workgroup's four waves takes its own quarter of the rows (their sums meet in LDS at the end).
// Every factor is bounded through  term <= 1  =>  W * Ga <= 1 / Elink, except for transitions weaker than 2^-100 (a finite
// link more than 69 nats under 0) or a W * Ga beyond 2^120: such lanes redo their vertices with the exact per-term form.
// A W under 2^-126 flushes: the dropped term is < 2^-126 * Ga <= 2^-6 ... in the scaled sum, i.e. an ABSOLUTE error below
// 2^-100 * ... of the final gradient — the reference's relative accuracy on gradients that are themselves < 1e-30 is not kept.
constexpr int GX_TC = 4;
constexpr int GX_P = 296;                     // pitch of a converted beta row: 256 own + 36 halo columns, 16-byte multiple
constexpr int GX_G = 76;                      // group exponents per row (73 used)
constexpr int GX_NEG = -(1 << 30);
constexpr int GX_WAVE_WORDS = 2 * GX_TC * GX_P + GX_TC * GX_G + 2 * GX_TC * 256;   // LDS words per wave
__device__ unsigned int g_gx_diag[4];      // [0] lanes that took the exact redo, [1] of them: unsafe factor, [2] weak transition

__global__ __launch_bounds__(256, 2) void dag_grad_links_exp_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) char gx_smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    const int b = blockIdx.y, i0 = blockIdx.x * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Bq = reinterpret_cast<float*>(gx_smem) + (size_t)wave * GX_WAVE_WORDS;   // [TC][GX_P] values
    int* Xq = reinterpret_cast<int*>(Bq + GX_TC * GX_P);                                            // [TC][GX_G] group exponents
    float* Aq = reinterpret_cast<float*>(Xq + GX_TC * GX_G);                                        // [TC][256]  alpha rows (own vertices)
    float* Braw = Aq + GX_TC * 256;                                                                 // [TC][GX_P] beta rows as they arrive (LDS-DMA)
    float* Araw = Braw + GX_TC * GX_P;                                                              // [TC][256]  alpha rows as they arrive
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float* A = alpha + (size_t)b * TL;
    const float* Bp = beta + (size_t)b * TL;
    const float b00 = Bp[0];
    const bool dead = isinf(b00) || Tb > T || Lb > L || Tb < 1 || Lb < 1;
    const float b00_2 = b00 * LOG2E;
    const int v0 = i0 + 4 * lane;                               // this lane's four source vertices v0 .. v0+3
    float acc[4][32];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 32; ++d) acc[c][d] = 0.f;
    // rows t = 0 .. Tb-2, a contiguous quarter per wave
    const int nt = dead ? 0 : (Tb - 1);
    const int per = (nt + 3) >> 2;
    const int tlo = min(nt, wave * per), thi = min(nt, tlo + per);
    bool bad = false;                                            // a factor left its safe range: redo this lane exactly
    // The rows of pass n+1 stream into LDS (LDS-DMA, no registers) while pass n is consumed: a memory round trip per pass
    // would otherwise be exposed (the conversion needs the data, and 190+ VGPRs of accumulators leave no room to prefetch).
    auto request = [&](int tb0) {
        const int nr = min(GX_TC, thi - tb0);
        for (int r = 0; r < nr; ++r) {
            const float* brow = Bp + (size_t)(tb0 + r + 1) * L;
            const int c0 = i0 + 4 * lane, c1 = i0 + 256 + 4 * lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow + (c0 < L ? c0 : 0)),
                                             (__attribute__((address_space(3))) void*)(Braw + r * GX_P), 16, 0, 0);
            if (lane < 9)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow + (c1 < L ? c1 : 0)),
                                                 (__attribute__((address_space(3))) void*)(Braw + r * GX_P + 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)(tb0 + r) * L + (c0 < L ? c0 : 0)),
                                             (__attribute__((address_space(3))) void*)(Araw + r * 256), 16, 0, 0);
        }
    };
    if (tlo < thi) request(tlo);
    for (int tb = tlo; tb < thi; tb += GX_TC) {
        const int rows = min(GX_TC, thi - tb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this pass's rows have landed
        // ---- convert beta rows tb+1 .. tb+rows: columns i0 .. i0+291 as 73 groups of 4 (window element q <-> column i0 + q)
        // (r04: every raw read of the pass leaves as ONE batch — twelve ds_read_b128, unconditional, masked afterwards — before the first
        //  conversion; the branchy per-row form this replaces cost one exposed LDS round trip per (row, group) and pass: ten per pass)
        float4 ra[GX_TC], rb0[GX_TC], rb1[GX_TC];
        const int g1 = lane + 64, g1c = g1 < 73 ? g1 : 72;
#pragma unroll
        for (int r = 0; r < GX_TC; ++r) {
            ra[r] = *reinterpret_cast<const float4*>(Araw + r * 256 + 4 * lane);
            rb0[r] = *reinterpret_cast<const float4*>(Braw + r * GX_P + 4 * lane);
            rb1[r] = *reinterpret_cast<const float4*>(Braw + r * GX_P + 4 * g1c);
        }
        const float4 ninf4 = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        auto convert = [&](float4 v, bool live, int r, int g, bool store) {
            if (!live) v = ninf4;
            const float x0 = v.x * LOG2E, x1 = v.y * LOG2E, x2 = v.z * LOG2E, x3 = v.w * LOG2E;
            const float gm = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
            const bool gd = gm == NEG_INF;
            const float cf = gd ? 0.f : ceilf(gm);
            const float4 o = make_float4(__builtin_amdgcn_exp2f(x0 - cf), __builtin_amdgcn_exp2f(x1 - cf), __builtin_amdgcn_exp2f(x2 - cf),
                                         __builtin_amdgcn_exp2f(x3 - cf));
            if (store) {
                *reinterpret_cast<float4*>(Bq + r * GX_P + 4 * g) = o;
                Xq[r * GX_G + g] = gd ? GX_NEG : (int)cf;
            }
        };
#pragma unroll
        for (int r = 0; r < GX_TC; ++r) {
            float4 a4 = ra[r];
            if (!(r < rows && v0 < L)) a4 = ninf4;
            *reinterpret_cast<float4*>(Aq + r * 256 + 4 * lane) = a4;
            convert(rb0[r], r < rows && i0 + 4 * lane < L, r, lane, true);
        }
        if (g1 < 73) {
#pragma unroll
            for (int r = 0; r < GX_TC; ++r) convert(rb1[r], r < rows && i0 + 4 * g1 < L, r, g1, true);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS operations execute in order
        if (tb + GX_TC < thi) request(tb + GX_TC);               // the raw rows are free again: next pass streams in meanwhile
        // ---- consume: vertex v0+c, transition d -> window element q = c + 1 + d of the lane's 36-value window (groups lane .. lane+8)
        for (int r = 0; r < rows; ++r) {
            int xw[9]; float w[36];
#pragma unroll
            for (int g = 0; g < 9; ++g) xw[g] = Xq[r * GX_G + lane + g];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(Bq + r * GX_P + 4 * lane + 4 * k);
                w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
            }
            // Three scale domains per lane-row, so that no factor can leave fp32 whatever the slope of the rows (next to the
            // DP's diagonal neighbouring vertices sit 25-35 binades apart):
            //   groups 1..7 (inside the band of all four vertices): reference R = their largest exponent, W = value * 2^(X - R) <= 1,
            //                 Ga[c] = 2^(alpha2 - b00_2 + R) <= 2 / Elink  because some in-band W is >= 1/2;
            //   group 0 (elements 1..3: successors of vertices 0..2 only) and group 8 (elements 32..35): used unscaled
            //                 (values <= 1 relative to their own exponent) with their own G0[c] / G8[c].
            int R = max(max(max(xw[1], xw[2]), max(xw[3], xw[4])), max(max(xw[5], xw[6]), xw[7]));
            const bool liveM = R != GX_NEG, live0 = xw[0] != GX_NEG, live8 = xw[8] != GX_NEG;
            if (!liveM) R = 0;
#pragma unroll
            for (int g = 1; g < 8; ++g) {
                const float f = ldexpf(1.0f, xw[g] - R);                    // <= 1; 0 for dead groups (ldexp saturates)
                w[4 * g] *= f; w[4 * g + 1] *= f; w[4 * g + 2] *= f; w[4 * g + 3] *= f;
            }
            const float4 av = *reinterpret_cast<const float4*>(Aq + r * 256 + 4 * lane);
            const float ua[4] = {av.x, av.y, av.z, av.w};
            const float RM = (float)R - b00_2, R0 = (float)(live0 ? xw[0] : 0) - b00_2, R8 = (float)(live8 ? xw[8] : 0) - b00_2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float u2 = ua[c] * LOG2E;
                const float eM = u2 + RM, e0 = u2 + R0, e8 = u2 + R8;
                bad |= (liveM & (eM > 126.f)) | (live0 & (c < 3) & (e0 > 126.f)) | (live8 & (e8 > 126.f));   // only with transitions < 2^-100
                const float GM = (liveM & (eM <= 126.f)) ? __builtin_amdgcn_exp2f(eM) : 0.f;
                const float G0 = (live0 & (c < 3) & (e0 <= 126.f)) ? __builtin_amdgcn_exp2f(e0) : 0.f;
                const float G8 = (live8 & (e8 <= 126.f)) ? __builtin_amdgcn_exp2f(e8) : 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) {
                    const int q = c + 1 + d;
                    acc[c][d] = fmaf(w[q], q < 4 ? G0 : (q >= 32 ? G8 : GM), acc[c][d]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next pass overwrites the rows
    }
    // ---- transition weights; lanes with an unsafe factor or a transition under 2^-100 redo their sums term by term ----
    const float go = dead ? 0.f : g_out[b];
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        const int vi = v0 + c;
        float e2[32]; bool okd[32]; bool weak = false;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            okd[d] = !dead && d < TR && vi < Lb && vi + d + 1 < Lb;                                  // dag_loss.cu:461-466
            e2[d] = okd[d] ? links[((size_t)b * L + vi) * TR + d] * LOG2E : NEG_INF;
            weak |= okd[d] & (e2[d] != NEG_INF) & (e2[d] < -100.f);
        }
        float s[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) s[d] = (c == 0) ? acc[0][d] : (c == 1) ? acc[1][d] : (c == 2) ? acc[2][d] : acc[3][d];
        if (__builtin_expect(bad | weak, 0)) {
            if (c == 0) { atomicAdd(&g_gx_diag[0], 1u); if (bad) atomicAdd(&g_gx_diag[1], 1u); if (weak) atomicAdd(&g_gx_diag[2], 1u); }
            // exact form of dag_loss.cu:471-475 over this wave's rows (the scaled sum is discarded)
#pragma unroll
            for (int d = 0; d < 32; ++d) s[d] = 0.f;
            for (int t = tlo; t < thi; ++t) {
                const float a = A[(size_t)t * L + vi] * LOG2E - b00_2;
#pragma unroll
                for (int d = 0; d < 32; ++d)
                    if (okd[d]) s[d] += __builtin_amdgcn_exp2f(a + Bp[(size_t)(t + 1) * L + vi + d + 1] * LOG2E + e2[d]);
            }
#pragma unroll
            for (int d = 0; d < 32; ++d) s[d] = okd[d] ? s[d] * go : 0.f;
        } else {
#pragma unroll
            for (int d = 0; d < 32; ++d) s[d] = okd[d] ? s[d] * __builtin_amdgcn_exp2f(e2[d]) * go : 0.f;
        }
#pragma unroll
        for (int d = 0; d < 32; ++d) { if (c == 0) acc[0][d] = s[d]; else if (c == 1) acc[1][d] = s[d]; else if (c == 2) acc[2][d] = s[d]; else acc[3][d] = s[d]; }
    }
    // ---- the four waves' partial sums meet in LDS (one vertex column at a time), wave 0 stores ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(gx_smem);               // [3 waves][64 lanes][33]
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (wave > 0) {
#pragma unroll
            for (int d = 0; d < 32; ++d)
                red[((wave - 1) * 64 + lane) * 33 + d] = (c == 0) ? acc[0][d] : (c == 1) ? acc[1][d] : (c == 2) ? acc[2][d] : acc[3][d];
        }
        __syncthreads();
        if (wave == 0) {
            const int vi = v0 + c;
            float s[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                s[d] = (c == 0) ? acc[0][d] : (c == 1) ? acc[1][d] : (c == 2) ? acc[2][d] : acc[3][d];
                s[d] += red[(0 * 64 + lane) * 33 + d] + red[(1 * 64 + lane) * 33 + d] + red[(2 * 64 + lane) * 33 + d];
            }
            if (vi < L) {
                float* out = g_links + ((size_t)b * L + vi) * TR;
                if (TR == 32) {
#pragma unroll
                    for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(out + d) = make_float4(s[d], s[d + 1], s[d + 2], s[d + 3]);
                } else {
#pragma unroll
                    for (int d = 0; d < 32; ++d) if (d < TR) out[d] = s[d];
                }
            }
        }
        __syncthreads();
    }
}

// PROCESS-wide: dsp_dag_loss_bwd is called from PyTorch's autograd worker thread, not from the thread that pins the kernel (a
// thread_local pin was silently ignored by every backward driven through torch.autograd, r02 ADVICE).  g_k5_last records which
// family the last backward launched (1 tiled log space, 2 exp space, 3 dense block products) so a test can assert its pin took.
static std::atomic<int> g_k5_path{0};                      // 0 auto, 1 tiled log-space kernel, 2 exp-space kernel (TR <= 32) / dense block products (TR > 64)
static std::atomic<unsigned int> g_k5_last{0};
int k5_diag(unsigned int* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gx_diag), 16);
    unsigned int z[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_gx_diag), z, 16);
    out[3] = g_k5_last.exchange(0u);
    return (int)e;
}
bool grad_dense_supported(int L, int TR);
int launch_dag_grad_links_dense(const float*, const float*, const float*, const float*, const int64_t*, const int64_t*, float*, int, int, int, int, hipStream_t);

void set_k5_path(int v) { g_k5_path = v; }

int launch_dag_bwd_generic(const float* g_out, const float* alpha, const float* beta, const float* match, const float* links,
                           const int64_t* out_len, const int64_t* tgt_len, float* g_match, float* g_links,
                           int B, int T, int L, int TR, hipStream_t st)
{
    if (g_match) {
        const size_t TL = (size_t)T * L;
        int gx = (int)((TL / 4 + 255) / 256); if (gx < 1) gx = 1; if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(dag_grad_match_kernel, dim3(gx, B), dim3(256), 0, st, g_out, alpha, beta, match, g_match, B, TL);
        int rc = check_launch("dag_loss_bwd(grad_match)");
        if (rc) return rc;
    }
    const bool expk = TR <= 32 && (L & 3) == 0 && ((((uintptr_t)alpha) | ((uintptr_t)beta) | ((uintptr_t)g_links)) & 15) == 0;
    if (g_links && expk && g_k5_path != 1) {
        const size_t lds = (size_t)4 * GX_WAVE_WORDS * 4;
        (void)hipFuncSetAttribute((const void*)dag_grad_links_exp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(dag_grad_links_exp_kernel, dim3((L + 255) / 256, B), dim3(256), lds, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR);
        int rc = check_launch("dag_loss_bwd(grad_links, exp space)");
        if (rc) return rc;
        g_k5_last = 2u;
    } else if (g_links && g_k5_path != 1 && grad_dense_supported(L, TR)) {
        // dense window: block products over the target axis on the f32 matrix cores (dag_grad_dense.hip).  Half of the compact
        // [L][TR] layout addresses vertices past the graph (i + d + 1 >= L): zeros, as the reference's at::zeros leaves them.
        hipError_t e = hipMemsetAsync(g_links, 0, (size_t)B * L * TR * sizeof(float), st);
        if (e != hipSuccess) { set_error("hipMemsetAsync(grad_links): %s", hipGetErrorString(e)); return (int)e; }
        int rc = launch_dag_grad_links_dense(g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, st);
        if (rc) return rc;
        g_k5_last = 3u;
    } else if (g_links) {
        hipLaunchKernelGGL(dag_grad_links_tiled_kernel, dim3((L + 63) / 64, (TR + 31) / 32, B), dim3(256), 0, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR);
        int rc = check_launch("dag_loss_bwd(grad_links)");
        if (rc) return rc;
        g_k5_last = 1u;
    }
    return DSP_OK;
}

}  // namespace dsp
