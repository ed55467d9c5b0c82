// dag_grad.hip — K4 (grad wrt match_all) and K5 (grad wrt links) for gfx950, generic log-space form.
// Replaces calculate_grad_match_all_kernel (dag_loss.cu:378-401) and calculate_grad_links_kernel (:432-485).
#include "common.h"
#include <atomic>
#include <cmath>
#include <type_traits>
#include <stdlib.h>

namespace dsp {

// K4: pure elementwise, HBM-bound: 3 reads + 1 write per cell, float4 wide.  Rows are addressed through their pitches (r06: lda for alpha / beta,
// ldm for match, ldg for grad_match — all equal to L for dense tensors); with 16-byte aligned rows a lane's float4 may straddle L: it then
// reads / writes inside the pitch padding, which belongs to the caller's buffers.
__global__ __launch_bounds__(256) void dag_grad_match_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ match, float* __restrict__ g_match, int B, int T, int L, int lda, int ldm, int ldg)
{
    const int b = blockIdx.y;
    const float b00 = beta[(size_t)b * T * lda];
    const float go = g_out[b];
    const bool dead = isinf(b00);
    const float* A = alpha + (size_t)b * T * lda; const float* Bt = beta + (size_t)b * T * lda;
    const float* M = match + (size_t)b * T * ldm; float* G = g_match + (size_t)b * T * ldg;
    const bool al = ((((uintptr_t)A) | ((uintptr_t)Bt) | ((uintptr_t)M) | ((uintptr_t)G)) & 15) == 0 && ((lda | ldm | ldg) & 3) == 0;
    if (al) {
        const int n4 = (L + 3) >> 2;                                  // float4 groups per row (the last may reach into the padding)
        const size_t tot = (size_t)T * n4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
            const size_t t = i / n4; const int c = (int)(i - t * n4) * 4;
            const float4 a = *reinterpret_cast<const float4*>(A + t * lda + c), be = *reinterpret_cast<const float4*>(Bt + t * lda + c);
            const float4 m = *reinterpret_cast<const float4*>(M + t * ldm + c);
            float4 r;
            r.x = (dead || isinf(m.x)) ? 0.f : __expf(a.x + be.x - m.x - b00) * go;      // dag_loss.cu:394-398
            r.y = (dead || isinf(m.y)) ? 0.f : __expf(a.y + be.y - m.y - b00) * go;
            r.z = (dead || isinf(m.z)) ? 0.f : __expf(a.z + be.z - m.z - b00) * go;
            r.w = (dead || isinf(m.w)) ? 0.f : __expf(a.w + be.w - m.w - b00) * go;
            *reinterpret_cast<float4*>(G + t * ldg + c) = r;
        }
    } else {
        const size_t tot = (size_t)T * L;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
            const size_t t = i / L; const int c = (int)(i - t * L);
            const float m = M[t * ldm + c];
            G[t * ldg + c] = (dead || isinf(m)) ? 0.f : __expf(A[t * lda + c] + Bt[t * lda + c] - m - b00) * go;
        }
    }
}

// K5 generic: thread (dx = d, iy = i) sums over t.  alpha[t,i] is a broadcast within the 32 d-lanes,
// beta[t+1, i+d+1] is contiguous across them.
__global__ __launch_bounds__(256) void dag_grad_links_generic_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR, int lda)
{
    const int b = blockIdx.z;
    const int d = blockIdx.x * 32 + (threadIdx.x & 31);
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= L || d >= TR) return;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * lda;
    const float b00 = beta[(size_t)b * TL];
    float* out = g_links + ((size_t)b * L + i) * TR + d;
    const int nx = i + d + 1;
    if (i >= Lb || nx >= Lb || isinf(b00) || Tb > T || Lb > L) { *out = 0.f; return; }     // dag_loss.cu:461-466 (+ zeros init :541)
    const float* A = alpha + (size_t)b * TL + i;
    const float* Bt = beta + (size_t)b * TL + lda + nx;
    const float extra = links[((size_t)b * L + i) * TR + d] - b00;                          // :469
    float acc = 0.f;
    for (int t = 0; t + 1 < Tb; ++t)                                                        // :471-475
        acc += __expf(A[(size_t)t * lda] + Bt[(size_t)t * lda] + extra);
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
    float* __restrict__ g_links, int B, int T, int L, int TR, int lda)
{
    __shared__ float At[K5_TC][64];
    __shared__ float Bt[K5_TC][96];
    constexpr float LOG2E = 1.4426950408889634f;
    const int b = blockIdx.z;
    const int i0 = blockIdx.x * 64, dc0 = blockIdx.y * 32;
    const int tid = threadIdx.x, i = tid & 63, dg = tid >> 6;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * lda;
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
            if (r < rows && i0 + c < L) v = A[(size_t)(t0 + r) * lda + i0 + c] * LOG2E;
            At[r][c] = v;
        }
        for (int e = tid; e < K5_TC * 96; e += 256) {
            const int r = e / 96, c = e - r * 96;
            float v = NEG_INF;
            if (r < rows && bcol0 + c < L) v = Bp[(size_t)(t0 + r + 1) * lda + bcol0 + c] * LOG2E;
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
//
// FUSE (r06): the same launch also writes grad_match (K4, dag_loss.cu:378-401).  A pass already holds alpha[t] and beta[t+1] of the lane's
// own four vertices in registers before it converts them; with beta[t] carried over from the previous pass (four registers) and the
// match rows of the pass (MREG: prefetched into 4 x GX_TC registers across the consume phase; else LDS-DMA into a third raw buffer)
// grad_match[t] = exp(alpha + beta - match - beta00) * go leaves the wave as one 16-byte store per lane and row: alpha and beta are read
// ONCE for both gradients (K4 + K5 as two launches fetched them twice: 1.68 GB of HBM traffic for 1.107 GB of operands).  Rows
// t >= T_b - 1 (no transition starts there: the last real row and the padding) go through a plain streaming tail, a row per wave.
constexpr int GX_P = 296;                     // pitch of a converted beta row: 256 own + 36 halo columns, 16-byte multiple
constexpr int GX_G = 76;                      // group exponents per row (73 used)
constexpr int GX_NEG = -(1 << 30);
constexpr int gx_wave_words(int TC, bool mraw) { return 2 * TC * GX_P + TC * GX_G + 2 * TC * 256 + (mraw ? TC * 256 : 0); }   // LDS words per wave
__device__ unsigned int g_gx_diag[4];      // [0] lanes that took the exact redo, [1] of them: unsafe factor, [2] weak transition

// K4's cell, bit for bit (dag_grad_match_kernel above): the fused kernel's grad_match equals the two-launch form exactly
__device__ __forceinline__ float4 gx_match_cell(float4 a, float4 be, float4 m, float b00, float go, bool dead)
{
    float4 r;
    r.x = (dead || isinf(m.x)) ? 0.f : __expf(a.x + be.x - m.x - b00) * go;
    r.y = (dead || isinf(m.y)) ? 0.f : __expf(a.y + be.y - m.y - b00) * go;
    r.z = (dead || isinf(m.z)) ? 0.f : __expf(a.z + be.z - m.z - b00) * go;
    r.w = (dead || isinf(m.w)) ? 0.f : __expf(a.w + be.w - m.w - b00) * go;
    return r;
}

template <int GX_TC, bool FUSE, bool MREG, int AUX = 0>
__global__ __launch_bounds__(256, 2) void dag_grad_links_exp_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, const float* __restrict__ match, float* __restrict__ g_match, int B, int T, int L, int TR,
    int LDA, int LDM, int LDG, int remap)        // row pitches (elements) of alpha / beta, match, grad_match: multiples of 4, >= L rounded up to 4
{
    // r06: windows 33 .. 128 run the SAME kernel with gridDim.y = ceil(TR / 32): the workgroups of plane y own transitions S .. S+31, S = 32 y, and
    // read beta S columns to the right (a 128-byte shift keeps every 16-byte granule aligned); TR stays the row stride of links / grad_links.
    // grad_match is written by plane 0 only (fz).
    const int S = 32 * (int)blockIdx.y;
    const bool fz = FUSE && S == 0;
    extern __shared__ __attribute__((aligned(16))) char gx_smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int GX_WAVE_WORDS = gx_wave_words(GX_TC, FUSE && !MREG);
    // Tile order.  Workgroups are dealt round-robin to the 8 XCDs; remap = 1 gives each XCD a CONTIGUOUS run of (sample, column tile) pairs, so
    // that neighbouring tiles re-read each other's 36 halo columns out of one L2 — measured SLOWER (285 vs 269 us at C2: the XCD's 64 resident
    // workgroups then stream from a few adjacent address ranges); off by default, kept as a measurement switch.
    const int ntile = (L + 255) / 256, nwg = ntile * B;
    int wg = blockIdx.x;
    if (remap && (nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int b = wg / ntile, i0 = (wg - b * ntile) * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* Bq = reinterpret_cast<float*>(gx_smem) + (size_t)wave * GX_WAVE_WORDS;   // [TC][GX_P] values
    int* Xq = reinterpret_cast<int*>(Bq + GX_TC * GX_P);                                            // [TC][GX_G] group exponents
    float* Aq = reinterpret_cast<float*>(Xq + GX_TC * GX_G);                                        // [TC][256]  alpha rows (own vertices)
    float* Braw = Aq + GX_TC * 256;                                                                 // [TC][GX_P] beta rows as they arrive (LDS-DMA)
    float* Araw = Braw + GX_TC * GX_P;                                                              // [TC][256]  alpha rows as they arrive
    float* Mraw = Araw + GX_TC * 256;                                                               // [TC][256]  match rows (FUSE && !MREG only)
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* A = alpha + (size_t)b * T * LDA;
    const float* Bp = beta + (size_t)b * T * LDA;
    const float* Bs = Bp + S;                                    // window rows: column q of this launch is graph column q + S
    const float b00 = Bp[0];
    const bool dead = isinf(b00) || Tb > T || Lb > L || Tb < 1 || Lb < 1;
    const float b00_2 = b00 * LOG2E;
    const int v0 = i0 + 4 * lane;                               // this lane's four source vertices v0 .. v0+3
    // Accumulators in the pairing v_pk_fma_f32 wants (r06).  Vertex c, transition d meets window element q = c + 1 + d; a packed FMA takes an
    // EVEN-aligned register pair (w[2i], w[2i+1]) — which is exactly how the window arrives from LDS (ds_read_b128).  For odd c, q is even when
    // d is even: pairs (d, d+1), d = 0, 2, .. 30.  For even c, q is even when d is ODD: pairs (d, d+1), d = 1, 3, .. 29, and the two ends
    // d = 0 and d = 31 as single FMAs.  62 packed + 4 single FMAs per lane-row and no register shuffling: left to the compiler's own pairing the
    // asm-loaded window cost 90 v_mov per lane-row to re-align (256 -> 166 VALU instructions per lane-row in this loop).
    typedef float gx_v2f __attribute__((ext_vector_type(2)));
    gx_v2f accP[4][16];                                          // even c: [0..14] = d pairs (1,2) .. (29,30), [15] = (d = 0, d = 31)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accP[c][i].x = 0.f; accP[c][i].y = 0.f; }
    // rows t = 0 .. Tb-2, a contiguous quarter per wave
    const int nt = dead ? 0 : (Tb - 1);
    const int per = (nt + 3) >> 2;
    const int tlo = min(nt, wave * per), thi = min(nt, tlo + per);
    bool bad = false;                                            // a factor left its safe range: redo this lane exactly
    // The rows of pass n+1 stream into LDS (LDS-DMA, no registers) while pass n is consumed: a memory round trip per pass
    // would otherwise be exposed (the conversion needs the data, and 190+ VGPRs of accumulators leave no room to prefetch).
    const float* Mp = FUSE ? match + (size_t)b * T * LDM : nullptr;
    float* Gp = FUSE ? g_match + (size_t)b * T * LDG : nullptr;
    const float gom = FUSE ? g_out[b] : 0.f;
    const bool dead4 = isinf(b00);                               // K4's own test (dag_loss.cu:394)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rm[GX_TC];                                            // MREG: the match rows of the NEXT pass, in flight across the consume phase
    float4 carry = zero4;                                        // beta[tb][own vertices]: the last beta row of the previous pass
#pragma unroll
    for (int r = 0; r < GX_TC; ++r) rm[r] = zero4;
    if (FUSE && fz && tlo < thi && v0 < L) carry = *reinterpret_cast<const float4*>(Bp + (size_t)tlo * LDA + v0);
    auto request = [&](int tb0) {
        const int nr = min(GX_TC, thi - tb0);
        if (FUSE && MREG && fz) {
#pragma unroll
            for (int r = 0; r < GX_TC; ++r)
                if (r < nr && v0 < L) rm[r] = *reinterpret_cast<const float4*>(Mp + (size_t)(tb0 + r) * LDM + v0);
        }
        for (int r = 0; r < nr; ++r) {
            const float* brow = Bs + (size_t)(tb0 + r + 1) * LDA;
            const int c0 = i0 + 4 * lane, c1 = i0 + 256 + 4 * lane;
            if (FUSE && !MREG && fz)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Mp + (size_t)(tb0 + r) * LDM + (c0 < L ? c0 : 0)),
                                                 (__attribute__((address_space(3))) void*)(Mraw + r * 256), 16, 0, AUX);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow + (c0 + S < L ? c0 : 0)),
                                             (__attribute__((address_space(3))) void*)(Braw + r * GX_P), 16, 0, AUX);
            if (lane < 9)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow + (c1 + S < L ? c1 : 0)),
                                                 (__attribute__((address_space(3))) void*)(Braw + r * GX_P + 256), 16, 0, AUX);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (size_t)(tb0 + r) * LDA + (c0 < L ? c0 : 0)),
                                             (__attribute__((address_space(3))) void*)(Araw + r * 256), 16, 0, AUX);
        }
    };
    if (tlo < thi) request(tlo);
    for (int tb = tlo; tb < thi; tb += GX_TC) {
        const int rows = min(GX_TC, thi - tb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this pass's rows have landed
        // ---- convert beta rows tb+1 .. tb+rows: columns i0 .. i0+291 as 73 groups of 4 (window element q <-> column i0 + q)
        // (r04: every raw read of the pass leaves as ONE batch — twelve ds_read_b128, unconditional, masked afterwards — before the first
        //  conversion; the branchy per-row form this replaces cost one exposed LDS round trip per (row, group) and pass: ten per pass)
        // (r06: the 9 halo groups of ALL the pass's rows are converted by ONE call — lane h < 9 TC takes group 64 + h % 9 of row h / 9 — instead
        //  of one exec-masked call per row; rows past `rows` in the last pass are converted like the others, nobody consumes them)
        float4 ra[GX_TC], rb0[GX_TC];
        const int hr = lane < 9 * GX_TC ? lane / 9 : 0, hg = 64 + (lane < 9 * GX_TC ? lane % 9 : 0);
#pragma unroll
        for (int r = 0; r < GX_TC; ++r) {
            ra[r] = *reinterpret_cast<const float4*>(Araw + r * 256 + 4 * lane);
            rb0[r] = *reinterpret_cast<const float4*>(Braw + r * GX_P + 4 * lane);
            if (FUSE && !MREG) rm[r] = *reinterpret_cast<const float4*>(Mraw + r * 256 + 4 * lane);
        }
        const float4 rbh = *reinterpret_cast<const float4*>(Braw + hr * GX_P + 4 * hg);
        if (FUSE && fz) {
            // grad_match rows tb .. tb+rows-1: alpha[t] = ra[r], beta[t] = the previous row's rb0 (row tb: carried from the last pass)
#pragma unroll
            for (int r = 0; r < GX_TC; ++r) {
                const float4 bt = r == 0 ? carry : rb0[r > 0 ? r - 1 : 0];
                if (r < rows && v0 < L) {
                    const float4 gv = gx_match_cell(ra[r], bt, rm[r], b00, gom, dead4);
                    if (AUX) {
                        typedef float gx_s4 __attribute__((ext_vector_type(4)));
                        gx_s4 sv; sv.x = gv.x; sv.y = gv.y; sv.z = gv.z; sv.w = gv.w;
                        __builtin_nontemporal_store(sv, reinterpret_cast<gx_s4*>(Gp + (size_t)(tb + r) * LDG + v0));
                    } else {
                        *reinterpret_cast<float4*>(Gp + (size_t)(tb + r) * LDG + v0) = gv;
                    }
                }
            }
            carry = rb0[GX_TC - 1];
        }
        // a group outside the graph (columns >= L: the DMA fetched a clamped address) converts to "dead": its exponent is the sentinel and its
        // values 2^(x - inf) = 0 — one add and one or instead of four selects
        auto convert = [&](float4 v, float pen, bool live, int r, int g, bool store) {
            const float x0 = v.x * LOG2E, x1 = v.y * LOG2E, x2 = v.z * LOG2E, x3 = v.w * LOG2E;
            const float gm = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
            const bool gd = gm == NEG_INF;
            const float cf = gd ? 0.f : ceilf(gm);
            const float ce = cf + pen;
            const float4 o = make_float4(__builtin_amdgcn_exp2f(x0 - ce), __builtin_amdgcn_exp2f(x1 - ce), __builtin_amdgcn_exp2f(x2 - ce),
                                         __builtin_amdgcn_exp2f(x3 - ce));
            if (store) {
                *reinterpret_cast<float4*>(Bq + r * GX_P + 4 * g) = o;
                Xq[r * GX_G + g] = (gd | !live) ? GX_NEG : (int)cf;
            }
        };
        const bool live_own = i0 + 4 * lane + S < L, live_halo = i0 + 4 * hg + S < L;
        const float pen_own = live_own ? 0.f : __builtin_huge_valf(), pen_halo = live_halo ? 0.f : __builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < GX_TC; ++r) {
            *reinterpret_cast<float4*>(Aq + r * 256 + 4 * lane) = ra[r];       // (lanes past L hold a clamped column's values: finite, and never stored)
            convert(rb0[r], pen_own, live_own, r, lane, true);
        }
        convert(rbh, pen_halo, live_halo, hr, hg, lane < 9 * GX_TC);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS operations execute in order
        if (tb + GX_TC < thi) request(tb + GX_TC);               // the raw rows are free again: next pass streams in meanwhile
        // ---- consume: vertex v0+c, transition d -> window element q = c + 1 + d of the lane's 36-value window (groups lane .. lane+8)
        for (int r = 0; r < rows; ++r) {
            int xw[9];
            // r06: the row's 15 LDS reads (alpha, nine group exponents, the 36-value window) leave as ONE inline-asm issue group.  Written as
            // C++ loads the compiler cannot prove them disjoint from the LDS-DMA writes of the NEXT pass (same extern array) and put an
            // `s_waitcnt vmcnt(0)` in front of the first of them: every pass then waited for the next pass's rows BEFORE consuming its own —
            // the prefetch overlapped nothing (r03-r05: 257 us per launch).  LDS returns in order; the waits below are counted.
            typedef int gx_v2i __attribute__((ext_vector_type(2)));
            typedef float gx_v4f __attribute__((ext_vector_type(4)));
            gx_v4f avv, pv[9]; gx_v2i x01, x23, x45, x67; int x8;
            {
                const unsigned aaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Aq + r * 256 + 4 * lane);
                const unsigned xaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Xq + r * GX_G + lane);
                const unsigned vaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(Bq + r * GX_P + 4 * lane);
                asm volatile("ds_read_b128 %0, %15\n\t"
                             "ds_read2_b32 %1, %16 offset1:1\n\t"
                             "ds_read2_b32 %2, %16 offset0:2 offset1:3\n\t"
                             "ds_read2_b32 %3, %16 offset0:4 offset1:5\n\t"
                             "ds_read2_b32 %4, %16 offset0:6 offset1:7\n\t"
                             "ds_read_b32 %5, %16 offset:32\n\t"
                             "ds_read_b128 %6, %17\n\t"
                             "ds_read_b128 %7, %17 offset:16\n\t"
                             "ds_read_b128 %8, %17 offset:32\n\t"
                             "ds_read_b128 %9, %17 offset:48\n\t"
                             "ds_read_b128 %10, %17 offset:64\n\t"
                             "ds_read_b128 %11, %17 offset:80\n\t"
                             "ds_read_b128 %12, %17 offset:96\n\t"
                             "ds_read_b128 %13, %17 offset:112\n\t"
                             "ds_read_b128 %14, %17 offset:128"
                             : "=&v"(avv), "=&v"(x01), "=&v"(x23), "=&v"(x45), "=&v"(x67), "=&v"(x8),
                               "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8])
                             : "v"(aaddr), "v"(xaddr), "v"(vaddr)
                             : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(avv), "+v"(x01), "+v"(x23), "+v"(x45), "+v"(x67), "+v"(x8));      // alpha + exponents are in
            xw[0] = x01.x; xw[1] = x01.y; xw[2] = x23.x; xw[3] = x23.y; xw[4] = x45.x; xw[5] = x45.y; xw[6] = x67.x; xw[7] = x67.y; xw[8] = x8;
            // Three scale domains per lane-row, so that no factor can leave fp32 whatever the slope of the rows (next to the
            // DP's diagonal neighbouring vertices sit 25-35 binades apart):
            //   groups 1..7 (inside the band of all four vertices): reference R = their largest exponent, W = value * 2^(X - R) <= 1,
            //                 Ga[c] = 2^(alpha2 - b00_2 + R) <= 2 / Elink  because some in-band W is >= 1/2;
            //   group 0 (elements 1..3: successors of vertices 0..2 only) and group 8 (elements 32..35): used unscaled
            //                 (values <= 1 relative to their own exponent) with their own G0[c] / G8[c].
            int R = max(max(max(xw[1], xw[2]), max(xw[3], xw[4])), max(max(xw[5], xw[6]), xw[7]));
            const bool liveM = R != GX_NEG, live0 = xw[0] != GX_NEG, live8 = xw[8] != GX_NEG;
            if (!liveM) R = 0;
            float fg[9];
#pragma unroll
            for (int g = 1; g < 8; ++g) fg[g] = ldexpf(1.0f, xw[g] - R);    // <= 1; 0 for dead groups (ldexp saturates)
            const float ua[4] = {avv.x, avv.y, avv.z, avv.w};
            // a dead domain gets the reference -inf: its factor is 2^-inf = 0 without a select per vertex
            const float RM = liveM ? (float)R - b00_2 : NEG_INF, R0 = live0 ? (float)xw[0] - b00_2 : NEG_INF, R8 = live8 ? (float)xw[8] - b00_2 : NEG_INF;
            float GMs[4], G0s[4], G8s[4];
            float emax = NEG_INF;
#pragma unroll
            for (int c = 0; c < 4; ++c) {                                   // (the twelve v_exp run under the window reads still in flight)
                const float u2 = ua[c] * LOG2E;
                const float eM = u2 + RM, e0 = (c < 3) ? u2 + R0 : NEG_INF, e8 = u2 + R8;
                emax = fmaxf(emax, fmaxf(fmaxf(eM, e0), e8));
                GMs[c] = __builtin_amdgcn_exp2f(eM);                        // (a factor beyond 2^126 — only with transitions < 2^-100 — sets `bad`:
                G0s[c] = __builtin_amdgcn_exp2f(e0);                        //  the lane's scaled sums are then discarded and redone term by term)
                G8s[c] = __builtin_amdgcn_exp2f(e8);
            }
            bad |= emax > 126.f;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]));
            gx_v2f W2[18];                                                  // W2[i] = (w[2i], w[2i+1]); groups 1..7 scaled to the lane's reference
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                gx_v2f lo, hi; lo.x = pv[k].x; lo.y = pv[k].y; hi.x = pv[k].z; hi.y = pv[k].w;
                if (k >= 1 && k <= 7) { gx_v2f f2; f2.x = fg[k]; f2.y = fg[k]; lo = lo * f2; hi = hi * f2; }
                W2[2 * k] = lo; W2[2 * k + 1] = hi;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float GM = GMs[c], G0 = G0s[c], G8 = G8s[c];
                if (c & 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {                           // d = 2i, 2i+1 -> q = c + 1 + 2i (even), pair index (c + 1) / 2 + i
                        const int qi = (c + 1) / 2 + i, q = 2 * qi;
                        const float g = q < 4 ? G0 : (q >= 32 ? G8 : GM);
                        gx_v2f g2; g2.x = g; g2.y = g;
                        accP[c][i] = __builtin_elementwise_fma(W2[qi], g2, accP[c][i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 15; ++i) {                           // d = 2i+1, 2i+2 -> q = c + 2 + 2i (even)
                        const int qi = (c + 2) / 2 + i, q = 2 * qi;
                        const float g = q < 4 ? G0 : (q >= 32 ? G8 : GM);
                        gx_v2f g2; g2.x = g; g2.y = g;
                        accP[c][i] = __builtin_elementwise_fma(W2[qi], g2, accP[c][i]);
                    }
                    // d = 0 -> q = c + 1 (odd: the .y half of pair c / 2), domain of q < 4 -> G0;  d = 31 -> q = c + 32 (even: .x of pair c / 2 + 16) -> G8
                    accP[c][15].x = fmaf(W2[c / 2].y, G0, accP[c][15].x);
                    accP[c][15].y = fmaf(W2[c / 2 + 16].x, G8, accP[c][15].y);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next pass overwrites the rows
    }
    float acc[4][32];                                            // (renaming only: [vertex][transition] view of the paired accumulators for the epilogue)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            if (c & 1) acc[c][d] = (d & 1) ? accP[c][d >> 1].y : accP[c][d >> 1].x;
            else acc[c][d] = d == 0 ? accP[c][15].x : (d == 31 ? accP[c][15].y : ((d & 1) ? accP[c][(d - 1) >> 1].x : accP[c][(d - 1) >> 1].y));
        }
    }
    if (FUSE && fz && v0 < L) {
        // rows T_b-1 .. T-1 (and every row of a dead sample): no transition term, K4 alone — streamed, four rows in flight per wave
        for (int t = nt + wave; t < T; t += 16) {
            float4 a4[4], b4[4], m4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + 4 * u;
                if (tt < T) {
                    a4[u] = *reinterpret_cast<const float4*>(A + (size_t)tt * LDA + v0);
                    b4[u] = *reinterpret_cast<const float4*>(Bp + (size_t)tt * LDA + v0);
                    m4[u] = *reinterpret_cast<const float4*>(Mp + (size_t)tt * LDM + v0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + 4 * u;
                if (tt < T) *reinterpret_cast<float4*>(Gp + (size_t)tt * LDG + v0) = gx_match_cell(a4[u], b4[u], m4[u], b00, gom, dead4);
            }
        }
    }
    // ---- transition weights; lanes with an unsafe factor or a transition under 2^-100 redo their sums term by term ----
    const float go = dead ? 0.f : g_out[b];
    // (r06: both loops over the lane's four vertices are unrolled with a STATIC vertex index — as `#pragma unroll 1` loops every access to
    //  acc[c][*] was a 4-way select chain, ~1 000 v_cndmask per vertex; the compiler barrier between the vertices keeps their live ranges apart)
    auto scale_vertex = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        const int vi = v0 + c;
        float e2[32]; bool okd[32]; bool weak = false;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            okd[d] = !dead && S + d < TR && vi < Lb && vi + S + d + 1 < Lb;                          // dag_loss.cu:461-466
            e2[d] = okd[d] ? links[((size_t)b * L + vi) * TR + S + d] * LOG2E : NEG_INF;
            weak |= okd[d] & (e2[d] != NEG_INF) & (e2[d] < -100.f);
        }
        if (__builtin_expect(bad | weak, 0)) {
            if (c == 0) { atomicAdd(&g_gx_diag[0], 1u); if (bad) atomicAdd(&g_gx_diag[1], 1u); if (weak) atomicAdd(&g_gx_diag[2], 1u); }
            // exact form of dag_loss.cu:471-475 over this wave's rows (the scaled sum is discarded)
            float s[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) s[d] = 0.f;
            for (int t = tlo; t < thi; ++t) {
                const float a = A[(size_t)t * LDA + vi] * LOG2E - b00_2;
#pragma unroll
                for (int d = 0; d < 32; ++d)
                    if (okd[d]) s[d] += __builtin_amdgcn_exp2f(a + Bs[(size_t)(t + 1) * LDA + vi + d + 1] * LOG2E + e2[d]);
            }
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[c][d] = okd[d] ? s[d] * go : 0.f;
        } else {
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[c][d] = okd[d] ? acc[c][d] * __builtin_amdgcn_exp2f(e2[d]) * go : 0.f;
        }
        asm volatile("" ::: "memory");
    };
    scale_vertex(std::integral_constant<int, 0>{}); scale_vertex(std::integral_constant<int, 1>{});
    scale_vertex(std::integral_constant<int, 2>{}); scale_vertex(std::integral_constant<int, 3>{});
    // ---- the four waves' partial sums meet in LDS (one vertex column at a time), wave 0 stores ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(gx_smem);               // [3 waves][64 lanes][33]
    auto reduce_vertex = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (wave > 0) {
#pragma unroll
            for (int d = 0; d < 32; ++d) red[((wave - 1) * 64 + lane) * 33 + d] = acc[c][d];
        }
        __syncthreads();
        if (wave == 0) {
            const int vi = v0 + c;
            float s[32];
#pragma unroll
            for (int d = 0; d < 32; ++d)
                s[d] = acc[c][d] + red[(0 * 64 + lane) * 33 + d] + red[(1 * 64 + lane) * 33 + d] + red[(2 * 64 + lane) * 33 + d];
            if (vi < L) {
                float* out = g_links + ((size_t)b * L + vi) * TR + S;
                if (TR - S >= 32 && (TR & 3) == 0) {
#pragma unroll
                    for (int d = 0; d < 32; d += 4) *reinterpret_cast<float4*>(out + d) = make_float4(s[d], s[d + 1], s[d + 2], s[d + 3]);
                } else {
#pragma unroll
                    for (int d = 0; d < 32; ++d) if (S + d < TR) out[d] = s[d];
                }
            }
        }
        __syncthreads();
    };
    reduce_vertex(std::integral_constant<int, 0>{}); reduce_vertex(std::integral_constant<int, 1>{});
    reduce_vertex(std::integral_constant<int, 2>{}); reduce_vertex(std::integral_constant<int, 3>{});
}

// PROCESS-wide: dsp_dag_loss_bwd is called from PyTorch's autograd worker thread, not from the thread that pins the kernel (a
// thread_local pin was silently ignored by every backward driven through torch.autograd, r02 ADVICE).  g_k5_last records which
// family the last backward launched (1 tiled log space, 2 exp space, 3 dense block products) so a test can assert its pin took.
static std::atomic<int> g_k5_path{0};                      // 0 auto, 1 tiled log-space kernel, 2 exp-space kernel (TR <= 32) / dense block products (TR > 64)
static std::atomic<int> g_k5_fuse{0};                      // TR <= 32 with both gradients wanted: 0 auto (= 2), 1 one fused launch (match rows in registers),
                                                           // 2 one fused launch (match rows by LDS-DMA, 3-row passes), 3 two launches (K4, then K5)
                                                           // C2 (r06): 297 / 277 / 445 us, gradients bit-identical
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
void set_k5_fuse(int v) { g_k5_fuse = v; }

template <int TC, bool FUSE, bool MREG, int AUX = 0>
static int launch_gx(const float* g_out, const float* alpha, const float* beta, const float* links, const int64_t* out_len,
                     const int64_t* tgt_len, float* g_links, const float* match, float* g_match, int B, int T, int L, int TR,
                     int lda, int ldm, int ldg, hipStream_t st)
{
    const size_t lds = (size_t)4 * gx_wave_words(TC, FUSE && !MREG) * 4;
    auto kern = dag_grad_links_exp_kernel<TC, FUSE, MREG, AUX>;
    static const char* const e_rm = getenv("DSP_GX_REMAP");
    set_max_dynamic_lds((const void*)kern, (int)lds);
    hipLaunchKernelGGL(kern, dim3(((L + 255) / 256) * B, (TR + 31) / 32), dim3(256), lds, st,
                       g_out, alpha, beta, links, out_len, tgt_len, g_links, match, g_match, B, T, L, TR, lda, ldm, ldg, e_rm ? 1 : 0);
    return check_launch(FUSE ? "dag_loss_bwd(grad_match + grad_links, exp space, one launch)" : "dag_loss_bwd(grad_links, exp space)");
}

int launch_dag_bwd_generic(const float* g_out, const float* alpha, const float* beta, const float* match, const float* links,
                           const int64_t* out_len, const int64_t* tgt_len, float* g_match, float* g_links,
                           int B, int T, int L, int TR, int lda, int ldm, int ldg, hipStream_t st)
{
    // lda / ldm / ldg: row pitches (elements) of alpha & beta / match / grad_match.  The exp-space kernels need 16-byte aligned ROWS (pitches that
    // are multiples of 4 and cover L rounded up to 4 — the columns past L hold -inf in alpha / beta, as the strip kernels leave them); every
    // other family takes dense tensors only.
    const int L4 = (L + 3) & ~3;
    const bool rows16 = (lda & 3) == 0 && lda >= L4 && ((((uintptr_t)alpha) | ((uintptr_t)beta) | ((uintptr_t)g_links)) & 15) == 0;
    const bool expk = TR <= 32 && rows16;
    const bool pitched_m = (ldm & 3) == 0 && ldm >= L4 && (ldg & 3) == 0 && ldg >= L4;
    const int fuse = g_k5_fuse.load();
    // windows 33 .. 128 (r06): the TR <= 32 kernel with one plane of workgroups per block of 32 transitions (gridDim.y) — plane k reads beta 32 k
    // columns to the right and owns slots 32 k .. 32 k + 31 of links / grad_links, plane 0 writes grad_match too.  ONE launch (k5_last 6).
    // C2 at TR = 64 / 96 / 128: 0.92 / 1.23 / 1.52 ms (grad_match pass + tiled log-space kernel) -> 0.40 / 0.56 / 0.73 ms.  Its cost is a latency
    // chain per wave (~37 us + 0.2-0.3 us per target row, per round of 512 resident workgroups), the tiled kernel's is proportional to the terms
    // (~6e12 / s): auto takes the cheaper estimate; k5_path 3 pins the planes, 1 / 2 the tiled kernel / the dense block products.
    if (g_links && TR > 32 && TR <= 128 && rows16 && (g_k5_path == 0 || g_k5_path == 3)) {
        const double planes = (TR + 31) / 32, nwg = (double)((L + 255) / 256) * B * planes;
        const double est_planes = std::ceil(nwg / 512.0) * (37.0 + (nwg <= 256 ? 0.20 : 0.30) * T);      // fitted on ten shapes, profiles/r06_bwd_windows_33_128.txt
        const double est_tiled = 40.0 + (double)B * T * L * TR / 6.0e6;
        if (g_k5_path == 3 || est_planes < est_tiled) {
            const bool fa = g_match && pitched_m && fuse != 3 && ((((uintptr_t)match) | ((uintptr_t)g_match)) & 15) == 0;
            int rc;
            if (fa) {
                rc = launch_gx<3, true, false, 2>(g_out, alpha, beta, links, out_len, tgt_len, g_links, match, g_match, B, T, L, TR, lda, ldm, ldg, st);
            } else {
                if (g_match) {
                    const size_t TL = (size_t)T * L;
                    int gx = (int)((TL / 4 + 255) / 256); if (gx < 1) gx = 1; if (gx > 1024) gx = 1024;
                    hipLaunchKernelGGL(dag_grad_match_kernel, dim3(gx, B), dim3(256), 0, st, g_out, alpha, beta, match, g_match, B, T, L, lda, ldm, ldg);
                    rc = check_launch("dag_loss_bwd(grad_match)");
                    if (rc) return rc;
                }
                rc = launch_gx<4, false, false>(g_out, alpha, beta, links, out_len, tgt_len, g_links, nullptr, nullptr, B, T, L, TR, lda, lda, lda, st);
            }
            if (rc) return rc;
            g_k5_last = 6u;
            return DSP_OK;
        }
    }
    // both gradients of a banded graph: ONE launch reads alpha / beta / match once (k5_last 4 / 5)
    if (g_match && g_links && expk && pitched_m && g_k5_path != 1 && fuse != 3 && ((((uintptr_t)match) | ((uintptr_t)g_match)) & 15) == 0) {
        // measured at C2 (r06, us per launch): default cache policy + XCD-contiguous tiles 285, nt 276, round-robin tiles 269, round-robin + nt 257
        // (nt = aux 2 on the LDS-DMA row loads and a non-temporal grad_match store: every byte is touched once).  DSP_GX_NT=0 / DSP_GX_REMAP=1
        // bring the other variants back for measurements.
        static const char* const e_nt = getenv("DSP_GX_NT");
        const bool nt = !(e_nt && e_nt[0] == '0');
        int rc = (fuse != 1 && nt) ? launch_gx<3, true, false, 2>(g_out, alpha, beta, links, out_len, tgt_len, g_links, match, g_match, B, T, L, TR, lda, ldm, ldg, st)
               : fuse != 1 ? launch_gx<3, true, false>(g_out, alpha, beta, links, out_len, tgt_len, g_links, match, g_match, B, T, L, TR, lda, ldm, ldg, st)
                           : launch_gx<4, true, true>(g_out, alpha, beta, links, out_len, tgt_len, g_links, match, g_match, B, T, L, TR, lda, ldm, ldg, st);
        if (rc) return rc;
        g_k5_last = fuse != 1 ? 5u : 4u;
        return DSP_OK;
    }
    if (g_match) {
        const size_t TL = (size_t)T * L;
        int gx = (int)((TL / 4 + 255) / 256); if (gx < 1) gx = 1; if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(dag_grad_match_kernel, dim3(gx, B), dim3(256), 0, st, g_out, alpha, beta, match, g_match, B, T, L, lda, ldm, ldg);
        int rc = check_launch("dag_loss_bwd(grad_match)");
        if (rc) return rc;
    }
    if (g_links && expk && g_k5_path != 1) {
        int rc = launch_gx<4, false, false>(g_out, alpha, beta, links, out_len, tgt_len, g_links, nullptr, nullptr, B, T, L, TR, lda, lda, lda, st);
        if (rc) return rc;
        g_k5_last = 2u;
    } else if (g_links && lda != L) {
        // pitched alpha / beta outside the exp-space kernel's reach (TR > 32 never gets here: the wrappers hand those families dense tensors)
        hipLaunchKernelGGL(dag_grad_links_tiled_kernel, dim3((L + 63) / 64, (TR + 31) / 32, B), dim3(256), 0, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, lda);
        int rc = check_launch("dag_loss_bwd(grad_links)");
        if (rc) return rc;
        g_k5_last = 1u;
    } else if (g_links && g_k5_path != 1 && (TR > 128 || g_k5_path == 2) && grad_dense_supported(L, TR)) {
        // (auto keeps the tiled kernel up to TR = 128: 1.23 vs 1.86 ms at C2 / TR = 96, 1.52 vs 1.58 at 128, 0.23 vs 0.31 at T = 64 — r06)
        // dense window: block products over the target axis on the f32 matrix cores (dag_grad_dense.hip).  Half of the compact
        // [L][TR] layout addresses vertices past the graph (i + d + 1 >= L): zeros, as the reference's at::zeros leaves them.
        hipError_t e = hipMemsetAsync(g_links, 0, (size_t)B * L * TR * sizeof(float), st);
        if (e != hipSuccess) { set_error("hipMemsetAsync(grad_links): %s", hipGetErrorString(e)); return (int)e; }
        int rc = launch_dag_grad_links_dense(g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, st);
        if (rc) return rc;
        g_k5_last = 3u;
    } else if (g_links) {
        hipLaunchKernelGGL(dag_grad_links_tiled_kernel, dim3((L + 63) / 64, (TR + 31) / 32, B), dim3(256), 0, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, lda);
        int rc = check_launch("dag_loss_bwd(grad_links)");
        if (rc) return rc;
        g_k5_last = 1u;
    }
    return DSP_OK;
}

}  // namespace dsp
