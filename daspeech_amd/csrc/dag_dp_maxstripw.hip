// dag_dp_maxstripw.hip — dag_best_alignment for windows 33 .. 128 without a trace tensor (r06): values-only max-DP on column strips + a
// back-trace that recomputes the arg-max of the T cells it visits.
//
// dag_dp_maxstrip.hip does this for TR <= 32.  Windows 33 .. 64 ran the log-space strip kernel in its max / trace mode plus a pointer chase
// over a [B,T,L] int32 trace (C2 / TR = 64: 1.9 ms), windows above 64 the blocked max-plus DP built for dense windows (B = 32, T = 64,
// L = 4096: 1.66 ms at TR = 65 against 0.29 at TR = 64).  The (max, +) DP needs none of the exp-space machinery of the log-sum strips
// (dag_dp_strip2g / strip1g.hip): the previous row is ONE array of natural-log values in LDS, a cell is 2 x W adds and maxima,
//     alpha_max[t][j] = match[t][j] + max_d (alpha_max[t-1][j-d] + links[j-d][d-1])                    (dag_best_alignment.cu:84-118)
// in exactly the oracle's fp32 operations (an add per term, order-free maxima, one add of the emission): bit-identical values.
//   * CPL = 2: two vertices x 64 transitions per lane, strips of 512 vertices (windows 33 .. 64); CPL = 1: one vertex x 128, strips of 256
//     (65 .. 128).  The lane's window starts at the 16-byte boundary under its first predecessor and streams through two 6-read register
//     buffers (chunk c+1 requested before chunk c is consumed); its transitions sit in registers against that window (-inf where an element
//     is no predecessor), loaded through an LDS tile in halves of 64 slots.
//   * loader / fetch / publish helper waves, tagged granules, tickets: as the other strip kernels.
//   * back-trace (dag_backtrace_wide_kernel): one wave per sample; at (t, pos) lane d evaluates the predecessors at distances d+1 and d+65,
//     smallest predecessor index among equal maxima (the reference's tie rule as torch states it), -1 / stop where every
//     candidate is -inf (dag_best_alignment.cu:170-206 chases a stored trace instead).
#include "common.h"
#include <stdlib.h>

#define MW_CHUNK6_0 \
    "ds_read_b128 %0, %6\n\t" \
    "ds_read_b128 %1, %6 offset:16\n\t" \
    "ds_read_b128 %2, %6 offset:32\n\t" \
    "ds_read_b128 %3, %6 offset:48\n\t" \
    "ds_read_b128 %4, %6 offset:64\n\t" \
    "ds_read_b128 %5, %6 offset:80"
#define MW_CHUNK6_1 \
    "ds_read_b128 %0, %6 offset:96\n\t" \
    "ds_read_b128 %1, %6 offset:112\n\t" \
    "ds_read_b128 %2, %6 offset:128\n\t" \
    "ds_read_b128 %3, %6 offset:144\n\t" \
    "ds_read_b128 %4, %6 offset:160\n\t" \
    "ds_read_b128 %5, %6 offset:176"
#define MW_CHUNK6_2 \
    "ds_read_b128 %0, %6 offset:192\n\t" \
    "ds_read_b128 %1, %6 offset:208\n\t" \
    "ds_read_b128 %2, %6 offset:224\n\t" \
    "ds_read_b128 %3, %6 offset:240\n\t" \
    "ds_read_b128 %4, %6 offset:256\n\t" \
    "ds_read_b128 %5, %6 offset:272"
#define MW_CHUNK6_3 \
    "ds_read_b128 %0, %6 offset:288\n\t" \
    "ds_read_b128 %1, %6 offset:304\n\t" \
    "ds_read_b128 %2, %6 offset:320\n\t" \
    "ds_read_b128 %3, %6 offset:336\n\t" \
    "ds_read_b128 %4, %6 offset:352\n\t" \
    "ds_read_b128 %5, %6 offset:368"
#define MW_CHUNK6_4 \
    "ds_read_b128 %0, %6 offset:384\n\t" \
    "ds_read_b128 %1, %6 offset:400\n\t" \
    "ds_read_b128 %2, %6 offset:416\n\t" \
    "ds_read_b128 %3, %6 offset:432\n\t" \
    "ds_read_b128 %4, %6 offset:448\n\t" \
    "ds_read_b128 %5, %6 offset:464"
#define MW_CHUNK6_5 \
    "ds_read_b128 %0, %6 offset:480\n\t" \
    "ds_read_b128 %1, %6 offset:496\n\t" \
    "ds_read_b128 %2, %6 offset:512\n\t" \
    "ds_read_b128 %3, %6 offset:528\n\t" \
    "ds_read_b128 %4, %6 offset:544\n\t" \
    "ds_read_b128 %5, %6 offset:560"
#define MW_CHUNK5_2 \
    "ds_read_b128 %0, %5 offset:192\n\t" \
    "ds_read_b128 %1, %5 offset:208\n\t" \
    "ds_read_b128 %2, %5 offset:224\n\t" \
    "ds_read_b128 %3, %5 offset:240\n\t" \
    "ds_read_b128 %4, %5 offset:256"
#define MW_CHUNK3_5 \
    "ds_read_b128 %0, %3 offset:480\n\t" \
    "ds_read_b128 %1, %3 offset:496\n\t" \
    "ds_read_b128 %2, %3 offset:512"

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float mw_v4f __attribute__((ext_vector_type(4)));

struct MWParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS;
    int ldm, ldo;                             // row pitches (elements) of match / of the max-alpha table (>= L)
};

constexpr int MW_NT = 256;
constexpr int MW_RING = 8;
constexpr int MW_CH = 4;
constexpr u32 MW_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ u64 mw_gran_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mw_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int CPL>
__device__ __forceinline__ void maxstripw_body(const MWParams& p, char* smem_raw, int b, int s, int so)
{
    constexpr int TRP = 128 / CPL, W = CPL * MW_NT, RL = W + TRP, NCW = MW_NT / 64, NW = TRP + 4, NG = NW / 4;      // window: NW values = NG groups of 4
    constexpr int GPL = TRP / 64;                              // halo granules per helper lane
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  alpha_max rows (natural log)
    float* Mring = Abuf + 2 * RL;                              // [RING][W] match rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, L = p.L, TR = p.TR;
    const int j0 = s * W;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * p.ldm;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = p.alpha + (size_t)b * T * p.ldo;
    const int LDO = p.ldo;
    const int nrows = Tb;
    const bool has_producer = so > 0;
    const bool has_consumer = s < p.NS - 1 && j0 + W < Lb;
    const u64* hin = p.halo + ((size_t)b * p.NS + (has_producer ? s - 1 : 0)) * (size_t)T * TRP;
    u64* hout = p.halo + ((size_t)b * p.NS + s) * (size_t)T * TRP;
    // LDS geometry: li = col - j0 + TRP (halo [0, TRP))

    // ---- prologue: transitions -> registers through an LDS tile, in halves of 64 slots: tile[r][dd] = links[j0 - TRP + r][64 h + dd] (pitch 65)
    const int l = tid;
    const int par = (CPL * l) & 3;               // the window starts `par` elements before the lane's first predecessor slot
    const int j = j0 + CPL * l;
    float E[CPL][NW];                            // E[c][q]: transition into vertex c from window element q (natural log; -inf where q is no predecessor)
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int q = 0; q < NW; ++q) E[c][q] = NEG_INF;
    for (int h = 0; h < TRP / 64; ++h) {
        float* tile = reinterpret_cast<float*>(smem_raw);
        {
            constexpr int NTHR = MW_NT + 192, RPP = NTHR / 64;
            const int rlo = j0 - TRP;
            const int dd = tid & 63, r0 = tid >> 6, slot = 64 * h + dd;
            for (int rb = r0; rb < W + TRP; rb += 8 * RPP) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = rlo + rb + u * RPP;
                    const bool ok = slot < TR && i >= 0 && i < L;
                    const float raw = K[(size_t)(ok ? i : 0) * TR + (ok ? slot : 0)];
                    v[u] = ok ? raw : NEG_INF;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int r = rb + u * RPP; if (r < W + TRP) tile[r * 65 + dd] = v[u]; }
            }
        }
        __syncthreads();
        if (wave < NCW) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    const int d = TRP + par + c - q;                                           // distance of window element q from vertex c (runtime: par)
                    if (d >= 64 * h + 1 && d <= 64 * h + 64) E[c][q] = tile[(CPL * l + c - d + TRP) * 65 + (d - 1 - 64 * h)];
                }
        }
        __syncthreads();
    }

    if (wave < NCW) {
        // =========================================================== compute waves
        __builtin_amdgcn_s_setprio(2);
        const bool col_ok = j < L;
        mw_barrier();                            // prologue barrier: match row 0 is in the ring
        for (int it = 0; it < nrows; ++it) {
            const int t = it;
            const int cur = it & 1, prv = cur ^ 1;
            float a[CPL];
            float m[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { a[c] = NEG_INF; m[c] = Mring[(size_t)(it % MW_RING) * W + CPL * l + c]; }
            if (it == 0) {
                if (j == 0) a[0] = m[0];                                                       // alpha_max[0][0] = match[0][0]
            } else {
                float mx[CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) mx[c] = NEG_INF;
                mw_v4f pa[6], pb[6];
                const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Abuf + prv * RL + CPL * l - par);
#define MW_ISSUE6(buf, c) asm volatile(MW_CHUNK6_##c : "=&v"(buf[0]), "=&v"(buf[1]), "=&v"(buf[2]), "=&v"(buf[3]), "=&v"(buf[4]), "=&v"(buf[5]) : "v"(vaddr) : "memory");
#define MW_GROUP(buf, k, g, n) \
                { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(buf[k])); \
                  _Pragma("unroll") for (int c = 0; c < CPL; ++c) { \
                      const float x0 = buf[k].x + E[c][4 * (g)], x1 = buf[k].y + E[c][4 * (g) + 1], x2 = buf[k].z + E[c][4 * (g) + 2], x3 = buf[k].w + E[c][4 * (g) + 3]; \
                      mx[c] = fmaxf(fmaxf(mx[c], fmaxf(x0, x1)), fmaxf(x2, x3)); } }
                MW_ISSUE6(pa, 0)
                MW_ISSUE6(pb, 1)
                MW_GROUP(pa, 0, 0, 11) MW_GROUP(pa, 1, 1, 10) MW_GROUP(pa, 2, 2, 9) MW_GROUP(pa, 3, 3, 8) MW_GROUP(pa, 4, 4, 7) MW_GROUP(pa, 5, 5, 6)
                if constexpr (CPL == 2) {
                    asm volatile(MW_CHUNK5_2 : "=&v"(pa[0]), "=&v"(pa[1]), "=&v"(pa[2]), "=&v"(pa[3]), "=&v"(pa[4]) : "v"(vaddr) : "memory");      // groups 12 .. 16
                    MW_GROUP(pb, 0, 6, 10) MW_GROUP(pb, 1, 7, 9) MW_GROUP(pb, 2, 8, 8) MW_GROUP(pb, 3, 9, 7) MW_GROUP(pb, 4, 10, 6) MW_GROUP(pb, 5, 11, 5)
                    MW_GROUP(pa, 0, 12, 4) MW_GROUP(pa, 1, 13, 3) MW_GROUP(pa, 2, 14, 2) MW_GROUP(pa, 3, 15, 1) MW_GROUP(pa, 4, 16, 0)
                } else {
                    MW_ISSUE6(pa, 2)
                    MW_GROUP(pb, 0, 6, 11) MW_GROUP(pb, 1, 7, 10) MW_GROUP(pb, 2, 8, 9) MW_GROUP(pb, 3, 9, 8) MW_GROUP(pb, 4, 10, 7) MW_GROUP(pb, 5, 11, 6)
                    MW_ISSUE6(pb, 3)
                    MW_GROUP(pa, 0, 12, 11) MW_GROUP(pa, 1, 13, 10) MW_GROUP(pa, 2, 14, 9) MW_GROUP(pa, 3, 15, 8) MW_GROUP(pa, 4, 16, 7) MW_GROUP(pa, 5, 17, 6)
                    MW_ISSUE6(pa, 4)
                    MW_GROUP(pb, 0, 18, 11) MW_GROUP(pb, 1, 19, 10) MW_GROUP(pb, 2, 20, 9) MW_GROUP(pb, 3, 21, 8) MW_GROUP(pb, 4, 22, 7) MW_GROUP(pb, 5, 23, 6)
                    asm volatile(MW_CHUNK3_5 : "=&v"(pb[0]), "=&v"(pb[1]), "=&v"(pb[2]) : "v"(vaddr) : "memory");                                   // groups 30 .. 32
                    MW_GROUP(pa, 0, 24, 8) MW_GROUP(pa, 1, 25, 7) MW_GROUP(pa, 2, 26, 6) MW_GROUP(pa, 3, 27, 5) MW_GROUP(pa, 4, 28, 4) MW_GROUP(pa, 5, 29, 3)
                    MW_GROUP(pb, 0, 30, 2) MW_GROUP(pb, 1, 31, 1) MW_GROUP(pb, 2, 32, 0)
                }
#undef MW_GROUP
#undef MW_ISSUE6
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const bool act = (j + c >= t) && (j + c < Lb);                              // dag_best_alignment.cu:84
                    a[c] = act ? mx[c] + m[c] : NEG_INF;
                }
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                Abuf[cur * RL + TRP + CPL * l + c] = a[c];
                if (j + c < L) O[(size_t)t * LDO + j + c] = a[c];
            }
            mw_barrier();
        }
        if (col_ok) for (int t = Tb; t < T; ++t) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (j + c < L) O[(size_t)t * LDO + j + c] = NEG_INF;
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA, 4 bytes per lane)
        auto issue_row = [&](int itr) {
            const float* rowp = M + (size_t)itr * p.ldm;
            float* slot = Mring + (size_t)(itr % MW_RING) * W;
#pragma unroll
            for (int i = 0; i < W / 64; ++i) {
                const int col = j0 + i * 64 + lane;
                const float* g = rowp + (col < L ? col : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 64), 4, 0, 0);
            }
        };
        for (int r = 0; r < MW_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mw_barrier();
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + MW_RING - 1;
            if (nx < nrows) {
                issue_row(nx);
                if (W == 512) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");        // rows it+2 .. it+7 may stay in flight: 6 x (W / 64) DMAs
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            mw_barrier();
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: the left strip's TRP boundary values -> LDS (GPL per lane)
        u64 g[MW_CH][GPL];
#pragma unroll
        for (int k = 0; k < MW_CH; ++k)
#pragma unroll
            for (int e = 0; e < GPL; ++e) g[k][e] = 0;
        auto load_row = [&](int itr, u64 (&dst)[GPL]) {
#pragma unroll
            for (int e = 0; e < GPL; ++e) dst[e] = itr < nrows ? mw_gran_load(hin + (size_t)itr * TRP + GPL * lane + e) : 0;
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < MW_CH; ++k) load_row(k, g[k]);
        }
        mw_barrier();
        for (int itb = 0; itb < nrows; itb += MW_CH) {
#pragma unroll
            for (int k = 0; k < MW_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int cur = it & 1;
                float hv[GPL];
#pragma unroll
                for (int e = 0; e < GPL; ++e) hv[e] = NEG_INF;
                if (has_producer) {
                    const u32 want = p.tag_base + 1u + (u32)it;
                    u32 spins = 0;
                    while (true) {
                        bool ok = true;
#pragma unroll
                        for (int e = 0; e < GPL; ++e) ok &= (u32)(g[k][e] >> 32) == want;
                        if (__all(ok)) break;
#pragma unroll
                        for (int e = 0; e < GPL; ++e) if ((u32)(g[k][e] >> 32) != want) g[k][e] = mw_gran_load(hin + (size_t)it * TRP + GPL * lane + e);
                        if (++spins > MW_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int e = 0; e < GPL; ++e) hv[e] = __uint_as_float((u32)g[k][e]);
                }
#pragma unroll
                for (int e = 0; e < GPL; ++e) Abuf[cur * RL + GPL * lane + e] = hv[e];
                if (has_producer) load_row(it + MW_CH, g[k]);
                mw_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: the strip's last TRP columns -> granules
        const bool pl = has_consumer;
        mw_barrier();
        auto publish = [&](int itp) {            // row itp - 1 is complete
            const int tp = itp - 1;
#pragma unroll
            for (int e = 0; e < GPL; ++e)
                mw_gran_store(hout + (size_t)tp * TRP + GPL * lane + e, p.tag_base + 1u + (u32)tp, Abuf[((itp - 1) & 1) * RL + W + GPL * lane + e]);
        };
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) publish(it);
            mw_barrier();
        }
        if (pl && nrows > 0) publish(nrows);
    }
}

template <int CPL>
__global__ __launch_bounds__(MW_NT + 192) void dag_maxstripw_kernel(MWParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = CPL * MW_NT;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);
    const int tid = threadIdx.x;
    if (tid == 0) *s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = *s_ticket;                              // producers hold smaller tickets than their consumers
    const int so = (int)(ticket / p.B);
    const int b = (int)(ticket % p.B);
    const int s = so;
    const int j0 = s * W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {
        float* O = p.alpha + (size_t)b * T * p.ldo;
        for (int jj = j0 + tid; jj < j0 + W && jj < L; jj += MW_NT + 192)
            for (int t = 0; t < T; ++t) O[(size_t)t * p.ldo + jj] = NEG_INF;
        return;
    }
    maxstripw_body<CPL>(p, smem_raw + 16, b, s, so);
}

// ---- back-trace over a window of up to 128 predecessors: one wave per sample -------------------------------------------------------------
__global__ __launch_bounds__(64) void dag_backtrace_wide_kernel(
    const float* __restrict__ amax, const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    int64_t* __restrict__ path, int B, int T, int L, int TR, int LDA)
{
    extern __shared__ __attribute__((aligned(16))) int32_t bw_lp[];       // [L] path image
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int jj = lane; jj < L; jj += 64) bw_lp[jj] = -1;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    const float* A = amax + (size_t)b * T * LDA;
    const float* K = links + (size_t)b * L * TR;
    __syncthreads();
    if (valid) {
        int pos = Lb - 1;
        for (int t = Tb - 1; t >= 0 && pos >= 0; --t) {
            if (lane == 0) bw_lp[pos] = t;
            if (t == 0) break;
            // candidates: predecessor pos - 1 - d at distance d + 1, d = lane and lane + 64
            float x0 = NEG_INF, x1 = NEG_INF;
            const int i0 = pos - 1 - lane, i1 = pos - 65 - lane;
            if (lane < TR && i0 >= 0) x0 = A[(size_t)(t - 1) * LDA + i0] + K[(size_t)i0 * TR + lane];
            if (lane + 64 < TR && i1 >= 0) x1 = A[(size_t)(t - 1) * LDA + i1] + K[(size_t)i1 * TR + lane + 64];
            float mx = fmaxf(x0, x1);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (mx == NEG_INF) { pos = -1; break; }                 // no live predecessor: trace = -1 (dag_best_alignment.cu:120-127)
            // smallest predecessor index among the maxima = the LARGEST distance: the far half first
            const unsigned long long h1 = __ballot(x1 == mx), h0 = __ballot(x0 == mx);
            const int dsel = h1 ? (64 + 63 - __builtin_clzll(h1)) : (63 - __builtin_clzll(h0));
            pos = pos - 1 - dsel;
        }
    }
    __syncthreads();
    for (int jj = lane; jj < L; jj += 64) path[(size_t)b * L + jj] = bw_lp[jj];
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool maxstripw_supported(int L, int TR) { return TR > 32 && TR <= 128 && (size_t)L * 4 <= 150 * 1024; }
size_t maxstripw_ws_bytes(int B, int T, int L, int TR)
{
    const int cpl = TR <= 64 ? 2 : 1, W = cpl * MW_NT, TRP = 128 / cpl;
    return 256 + (size_t)B * ((L + W - 1) / W) * T * TRP * sizeof(u64);
}

template <int CPL>
static int launch_mw(MWParams& p, int B, int T, int L, hipStream_t st)
{
    constexpr int W = CPL * MW_NT, TRP = 128 / CPL, RL = W + TRP;
    p.NS = (L + W - 1) / W;
    const size_t halo_bytes = (size_t)B * p.NS * T * TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const size_t lds_main = (size_t)(2 * RL + MW_RING * W) * 4 + 16;
    const size_t lds_tile = (size_t)(W + TRP) * 65 * 4 + 16;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    auto k = dag_maxstripw_kernel<CPL>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(B * p.NS)), dim3(MW_NT + 192), lds, st, p);
    return check_launch("dag_best_alignment(maxstripw)");
}

// alpha_max by column strips (values only), then the wide back-trace: no trace tensor
int launch_dag_maxstripw(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                         float* alpha_max, int64_t* path, int B, int T, int L, int TR, int ldm, int ldo, hipStream_t st)
{
    MWParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha_max;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.ldm = ldm; p.ldo = ldo;
    int rc = TR <= 64 ? launch_mw<2>(p, B, T, L, st) : launch_mw<1>(p, B, T, L, st);
    if (rc) return rc;
    const size_t lds = (size_t)L * 4;
    set_max_dynamic_lds((const void*)dag_backtrace_wide_kernel, (int)lds);
    hipLaunchKernelGGL(dag_backtrace_wide_kernel, dim3(B), dim3(64), lds, st, alpha_max, links, out_len, tgt_len, path, B, T, L, TR, ldo);
    return check_launch("dag_best_alignment(wide back-trace)");
}

}  // namespace dsp
