// dag_dp_strip2.hip — banded (TR <= 32) DAG DP for gfx950, occupancy-oriented variant of dag_dp_strip4.hip:
// K2 alpha || K3 beta in exp space, K6 max-DP + trace.  Read dag_dp_banded.hip (strips, granule hand-off, tickets) and
// dag_dp_strip4.hip (exp-space recurrence, exactness guard) first; this file changes the WORK DISTRIBUTION:
//
//   * 2 COLUMNS PER LANE.  With 4 columns per lane the C2 problem (B=32, L=4096, both directions) is exactly 1024 compute
//     waves = ONE per SIMD, and rocprofv3 showed the kernel bound by exposed dependency stalls / LDS round trips / the row
//     barrier (VALU busy 1200 of 2550 cycles per row).  Two columns per lane halve the registers (68 weight registers
//     instead of 144), double the compute waves and let two workgroups share a CU: 2+ compute waves per SIMD.
//   * ONE HELPER WAVE PER WORKGROUP, LOADS ONLY.  vmcnt retires loads in order among loads but not against stores, so a
//     counted wait is only sound in a wave that issues nothing but loads.  The loader wave streams match rows AND the
//     neighbour strip's halo granules into LDS rings with global_load_lds (sc1 for the granules), PD rows ahead, retired by
//     ONE counted s_waitcnt per row.  Compute waves only store (alpha / trace / boundary granules) and never wait on vmcnt;
//     the lanes next to the halo convert the landed granules (tag check; a stale tag falls back to a direct poll).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));

struct StripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, [1] = error word, [2] = exact-fallback count
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int dbg;
};

constexpr int S2_RING = 8;
constexpr int S2_PD = 6;                      // prefetch distance (rows) of the loader wave
constexpr int S2_NEGSENT = -(1 << 30);     // "dead" exponent; far below any finite fp32 score
constexpr u32 S2_SPIN_LIMIT = 1u << 22;
constexpr float S2_LOG2E = 1.4426950408889634f;
constexpr float S2_LN2 = 0.6931471805599453f;

__device__ __forceinline__ u64 s2_gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void s2_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void s2_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void s2_wait_vmcnt() {
    if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// window element of (column c, distance d): alpha predecessor -> q = 32 + c - d ; beta successor -> q = c + d   (q in [0, 34))
template <bool BETA> __device__ __forceinline__ constexpr int q2(int c, int d) { return BETA ? (c + d) : (32 + c - d); }

template <int NT, int MODE, bool BETA>
__device__ __forceinline__ void strip2_body(const StripParams& p, char* smem_raw, int b, int s, int dirslot, int so)
{
    constexpr int W = 2 * NT, RL = W + 32, NCW = NT / 64, DPR = (W + 255) / 256;     // 1 KiB LDS-DMA pieces per match row
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL] a2 (MODE 0) / alpha_max (MODE 1)
    float* Pbuf = Abuf + 2 * RL;                               // [2][RL] mantissa 2^(a2 - ceil a2)
    int* Cbuf = reinterpret_cast<int*>(Pbuf + 2 * RL);         // [2][RL] exponent ceil(a2)
    float* Mring = reinterpret_cast<float*>(Cbuf + 2 * RL);    // [RING][W] match rows
    u64* Hring = reinterpret_cast<u64*>(Mring + S2_RING * W);  // [RING][32] halo granules as landed by LDS-DMA

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, L = p.L, TR = p.TR;
    const int j0 = s * W;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * L;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = (BETA ? p.beta : p.alpha) + (size_t)b * T * L;
    const int nrows = Tb;

    const bool has_producer = so > 0 && (BETA ? (j0 + W < Lb) : true);
    const bool has_consumer = BETA ? (s > 0) : (s < p.NS - 1 && j0 + W < Lb);
    const int prod_strip = BETA ? s + 1 : s - 1;
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * 32;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * 32;
    const int halo_li0 = BETA ? W : 0;
    const int own_li0 = BETA ? 0 : 32;

    // ---- prologue: the strip's transition rows -> LDS tile -> registers, in TWO passes (half the lanes each) so the tile
    //      (NT+34 rows x 33 floats = 38 KB) stays below the main-loop LDS footprint and two workgroups fit a CU ----
    constexpr int TROWS = NT + 34;
    auto stage_pass = [&](int pass) {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = NT + 64, RPP = NTHR / 32;
        const int rlo = (BETA ? j0 : (j0 - 32)) + pass * NT;
        const int dd = tid & 31, r0 = tid >> 5;
        for (int rb = r0; rb < TROWS; rb += 8 * RPP) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = rlo + rb + u * RPP;
                const bool ok = dd < TR && i >= 0 && i < L;
                const float raw = K[(size_t)(ok ? i : 0) * TR + (ok ? dd : 0)];
                v[u] = ok ? raw : NEG_INF;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = rb + u * RPP; if (r < TROWS) tile[r * 33 + dd] = v[u]; }
        }
    };

    if (wave < NCW) {
        // =========================================================== compute waves
        const int l = tid;
        const int j = j0 + 2 * l;
        const bool col_ok = j < L;
        auto cell_active = [&](int col, int t) -> bool {
            if (col < t || col >= Lb) return false;
            if (MODE == 1) return true;                              // max-DP: trace must be bit-exact incl. unreachable cells
            if (!BETA) return (long)col <= (long)t * TR;
            return (long)(Lb - 1 - col) <= (long)(Tb - 1 - t) * TR;
        };
        const float* tile = reinterpret_cast<const float*>(smem_raw);
        float lmax[2];
        v2f E2[2][17];                 // MODE 0: pair layout of 2^(link2 - lmax);  MODE 1: raw links in the same layout
        const int my_pass = (l >= NT / 2) ? 1 : 0;
        const int lt = l - my_pass * (NT / 2);          // lane index inside its pass's tile
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            stage_pass(pass);
            __syncthreads();
            if (pass == my_pass) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float raw[32];
                    float mx = NEG_INF;
#pragma unroll
                    for (int d = 1; d <= 32; ++d) {
                        float v;
                        if (!BETA) v = tile[(2 * lt + c - d + 32) * 33 + (d - 1)];
                        else { v = tile[(2 * lt + c) * 33 + (d - 1)]; if (j + c + d >= Lb) v = NEG_INF; }
                        raw[d - 1] = (MODE == 0) ? v * S2_LOG2E : v;
                        mx = fmaxf(mx, raw[d - 1]);
                    }
                    if (MODE == 0) {
                        if (mx == NEG_INF) mx = 0.f;
                        lmax[c] = mx;
#pragma unroll
                        for (int d = 0; d < 32; ++d) raw[d] = __builtin_amdgcn_exp2f(raw[d] - mx);
                    } else {
                        lmax[c] = 0.f;
                    }
                    const float fill = (MODE == 0) ? 0.f : NEG_INF;
#pragma unroll
                    for (int i = 0; i < 17; ++i) {
                        const int qa = 2 * i, qb = 2 * i + 1;
                        const int da = BETA ? (qa - c) : (32 + c - qa), db = BETA ? (qb - c) : (32 + c - qb);
                        E2[c][i].x = (da >= 1 && da <= 32) ? raw[(da >= 1 && da <= 32) ? da - 1 : 0] : fill;
                        E2[c][i].y = (db >= 1 && db <= 32) ? raw[(db >= 1 && db <= 32) ? db - 1 : 0] : fill;
                    }
                }
            }
            __syncthreads();
        }
        auto Eval = [&](int c, int d) -> float { const int q = q2<BETA>(c, d); return (q & 1) ? E2[c][q >> 1].y : E2[c][q >> 1].x; };
        // lanes that convert the landed halo granules (2 columns each) and lanes that publish this strip's boundary
        const bool halo_lane = tid < 16;
        const int pub_c = BETA ? (2 * l) : (2 * l - (W - 32));
        const bool pub_lane = has_consumer && pub_c >= 0 && pub_c < 32;
        s2_barrier();                            // prologue barrier: rows 0 .. PD-1 of match / halo are in the rings

        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            const int slot = it % S2_RING;
            const float2 mt = *reinterpret_cast<const float2*>(Mring + (size_t)slot * W + 2 * l);
            const float m2[2] = {mt.x, mt.y};

            // ---- halo of THIS row -> cur (consumed by the next iteration) ----
            if (halo_lane) {
                float hv[2] = {NEG_INF, NEG_INF};
                if (has_producer) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    const ulonglong2 gg = *reinterpret_cast<const ulonglong2*>(Hring + (size_t)slot * 32 + 2 * tid);
                    u64 x0 = gg.x, x1 = gg.y;
                    if ((u32)(x0 >> 32) != want || (u32)(x1 >> 32) != want) {      // DMA ran ahead of the producer: poll directly
                        u32 spins = 0;
                        for (;;) {
                            x0 = s2_gran_load(hin + (size_t)t * 32 + 2 * tid);
                            x1 = s2_gran_load(hin + (size_t)t * 32 + 2 * tid + 1);
                            if ((u32)(x0 >> 32) == want && (u32)(x1 >> 32) == want) break;
                            if (++spins > S2_SPIN_LIMIT) { atomicOr(&p.counters[1], 1u); break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    hv[0] = __uint_as_float((u32)x0); hv[1] = __uint_as_float((u32)x1);
                }
                *reinterpret_cast<float2*>(Abuf + cur * RL + halo_li0 + 2 * tid) = make_float2(hv[0], hv[1]);
                if (MODE == 0) {
                    float pn[2]; int cn[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bool dead = hv[c] == NEG_INF;
                        const float cf = dead ? 0.f : ceilf(hv[c]);
                        pn[c] = __builtin_amdgcn_exp2f(hv[c] - cf);
                        cn[c] = dead ? S2_NEGSENT : (int)cf;
                    }
                    *reinterpret_cast<float2*>(Pbuf + cur * RL + halo_li0 + 2 * tid) = make_float2(pn[0], pn[1]);
                    *reinterpret_cast<int2*>(Cbuf + cur * RL + halo_li0 + 2 * tid) = make_int2(cn[0], cn[1]);
                }
            }

            float a2[2] = {NEG_INF, NEG_INF};
            int arg[2] = {-1, -1};
            if (it == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool seed = BETA ? (j + c == Lb - 1) : (j + c == 0);
                    if (seed) a2[c] = (MODE == 0) ? m2[c] * S2_LOG2E : m2[c];
                }
            } else if (MODE == 0) {
                int cw[34];
                float pw[34];
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    const int2 ci = *reinterpret_cast<const int2*>(Cbuf + prv * RL + 2 * l + 2 * i);
                    const float2 pv = *reinterpret_cast<const float2*>(Pbuf + prv * RL + 2 * l + 2 * i);
                    cw[2 * i] = ci.x; cw[2 * i + 1] = ci.y; pw[2 * i] = pv.x; pw[2 * i + 1] = pv.y;
                }
                // per-column maximum exponent over its predecessors: alpha c0: q 0..31, c1: q 1..32; beta c0: q 1..32, c1: q 2..33
                int cm[2];
                if (!BETA) {
                    int common = cw[1];
#pragma unroll
                    for (int q = 2; q <= 31; ++q) common = max(common, cw[q]);
                    cm[0] = max(common, cw[0]); cm[1] = max(common, cw[32]);
                } else {
                    int common = cw[2];
#pragma unroll
                    for (int q = 3; q <= 32; ++q) common = max(common, cw[q]);
                    cm[0] = max(common, cw[1]); cm[1] = max(common, cw[33]);
                }
                const int hi = max(cm[0], cm[1]);
                int refi = 0x7fffffff;
                if (cm[0] != S2_NEGSENT) refi = cm[0];
                if (cm[1] != S2_NEGSENT) refi = min(refi, cm[1]);
                const bool any_live = hi != S2_NEGSENT;
                if (!any_live) refi = 0;
                const bool wide = (hi - refi) > 120;
                v2f S2[2];
                S2[0].x = S2[0].y = S2[1].x = S2[1].y = 0.f;
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    v2f w2;
                    w2.x = ldexpf(pw[2 * i], cw[2 * i] - refi);
                    w2.y = ldexpf(pw[2 * i + 1], cw[2 * i + 1] - refi);
                    S2[0] = __builtin_elementwise_fma(w2, E2[0][i], S2[0]);
                    S2[1] = __builtin_elementwise_fma(w2, E2[1][i], S2[1]);
                }
                float S[2] = {S2[0].x + S2[0].y, S2[1].x + S2[1].y};
                const float ref = (float)refi;
                bool need_fb = false;
                bool flag[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float cand = __builtin_amdgcn_logf(S[c]) + ref + lmax[c] + m2[c] * S2_LOG2E;
                    const bool okc = cell_active(j + c, t) & (cm[c] != S2_NEGSENT);
                    flag[c] = okc & (wide | !(S[c] >= 0x1p-97f));
                    a2[c] = (okc & !flag[c]) ? cand : NEG_INF;
                    need_fb |= flag[c];
                }
                if (__builtin_expect(need_fb, 0)) {
                    // (a) medium path: the flagged column against its own exact maximum, registers + LDS only
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (flag[c]) {
                            float cmx = NEG_INF;
#pragma unroll
                            for (int d = 1; d <= 32; ++d) cmx = fmaxf(cmx, Abuf[prv * RL + 2 * l + q2<BETA>(c, d)]);
                            float sc = 0.f;
#pragma unroll
                            for (int d = 1; d <= 32; ++d)
                                sc = fmaf(__builtin_amdgcn_exp2f(Abuf[prv * RL + 2 * l + q2<BETA>(c, d)] - cmx), Eval(c, d), sc);
                            S[c] = sc;
                            if (sc >= 0x1p-97f) a2[c] = __builtin_amdgcn_logf(sc) + cmx + lmax[c] + m2[c] * S2_LOG2E;
                        } else {
                            S[c] = 1.f;
                        }
                    }
                    // (b) exact log-space path for what is still below the threshold (raw links from HBM, 8 at a time)
#pragma unroll 1
                    for (int c = 0; c < 2; ++c) {
                        const float Sc = (c == 0) ? S[0] : S[1];
                        if (Sc >= 0x1p-97f) continue;
                        float amax = NEG_INF;
                        for (int d = 1; d <= 32; ++d) amax = fmaxf(amax, Abuf[prv * RL + 2 * l + (BETA ? (c + d) : (32 + c - d))]);
                        float r = NEG_INF;
                        if (amax != NEG_INF) {
                            atomicAdd(&p.counters[2], 1u);
                            float mx = NEG_INF, sum = 0.f;
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                float lk[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const int row = BETA ? (j + c) : (j + c - d);
                                    const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                    const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                    lk[u] = ok ? raw * S2_LOG2E : NEG_INF;
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const float v = Abuf[prv * RL + 2 * l + (BETA ? (c + d) : (32 + c - d))] + lk[u];
                                    const float nm = fmaxf(mx, v);
                                    if (nm != NEG_INF) sum = sum * __builtin_amdgcn_exp2f(mx - nm) + __builtin_amdgcn_exp2f(v - nm);
                                    mx = nm;
                                }
                            }
                            if (mx != NEG_INF) r = __builtin_amdgcn_logf(sum) + mx + ((c == 0) ? m2[0] : m2[1]) * S2_LOG2E;
                        }
                        if (c == 0) a2[0] = r; else a2[1] = r;
                    }
                }
            } else {
                // MODE 1: max-DP, natural domain.  The reference scans predecessors in ascending index with a strict '>' (the
                // smallest index wins a tie, -1 if every candidate is -inf).  A serial scan is a 32-deep compare -> select
                // chain per vertex; the same rule as a TOURNAMENT (the right contender wins only if strictly greater) is five
                // levels deep, so the 93 compare/select nodes of a vertex issue back to back.
                float wv[34];
#pragma unroll
                for (int i = 0; i < 17; ++i) {
                    const float2 v = *reinterpret_cast<const float2*>(Abuf + prv * RL + 2 * l + 2 * i);
                    wv[2 * i] = v.x; wv[2 * i + 1] = v.y;
                }
                float mxv[2]; int av[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float xv[32]; int xi[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) {                 // candidate k: window element q = c + k, transition d = 32 - k
                        const int q = c + k;
                        xv[k] = wv[q] + ((q & 1) ? E2[c][q >> 1].y : E2[c][q >> 1].x);
                        xi[k] = q;
                    }
#pragma unroll
                    for (int n = 16; n >= 1; n >>= 1) {
#pragma unroll
                        for (int pidx = 0; pidx < n; ++pidx) {
                            const bool gt = xv[2 * pidx + 1] > xv[2 * pidx];
                            xv[pidx] = gt ? xv[2 * pidx + 1] : xv[2 * pidx];
                            xi[pidx] = gt ? xi[2 * pidx + 1] : xi[2 * pidx];
                        }
                    }
                    mxv[c] = xv[0];
                    av[c] = (xv[0] == NEG_INF) ? -1 : (j - 32 + xi[0]);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) if (cell_active(j + c, t)) { a2[c] = mxv[c] + m2[c]; arg[c] = av[c]; }
            }

            // ---- write the row: LDS state for the next row, HBM output, boundary granules ----
            if (MODE == 0) {
                float pn[2]; int cn[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool dead = a2[c] == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(a2[c]);
                    pn[c] = __builtin_amdgcn_exp2f(a2[c] - cf);
                    cn[c] = dead ? S2_NEGSENT : (int)cf;
                }
                *reinterpret_cast<float2*>(Pbuf + cur * RL + own_li0 + 2 * l) = make_float2(pn[0], pn[1]);
                *reinterpret_cast<int2*>(Cbuf + cur * RL + own_li0 + 2 * l) = make_int2(cn[0], cn[1]);
            }
            *reinterpret_cast<float2*>(Abuf + cur * RL + own_li0 + 2 * l) = make_float2(a2[0], a2[1]);
            if (col_ok) {
                const float2 o = (MODE == 0) ? make_float2(a2[0] * S2_LN2, a2[1] * S2_LN2) : make_float2(a2[0], a2[1]);
                *reinterpret_cast<float2*>(O + (size_t)t * L + j) = o;
                if (MODE == 1) *reinterpret_cast<int2*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int2(arg[0], arg[1]);
            }
            if (pub_lane) {
                const u32 tag = p.tag_base + 1u + (u32)t;
                s2_gran_store(hout + (size_t)t * 32 + pub_c, tag, a2[0]);
                s2_gran_store(hout + (size_t)t * 32 + pub_c + 1, tag, a2[1]);
            }
            s2_barrier();
        }
        if (col_ok) for (int t = Tb; t < T; ++t) {
            *reinterpret_cast<float2*>(O + (size_t)t * L + j) = make_float2(NEG_INF, NEG_INF);
            if (MODE == 1) *reinterpret_cast<int2*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int2(-1, -1);
        }
    } else {
        // =========================================================== loader wave: LOADS ONLY (counted vmcnt is sound)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * L;
            float* mslot = Mring + (size_t)(itr % S2_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);
                if (i * 256 + lane * 4 < W)                      // the last piece of a 384-column strip is half a wave
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(mslot + i * 256), 16, 0, 0);
            }
            if (has_producer) {
                // 32 granules = 256 B: lanes 0..15 carry them, the others re-read lane 0's piece into a scratch slot tail
                const u64* g = hin + (size_t)t * 32 + 2 * (lane & 15);
                u64* hs = Hring + (size_t)(itr % S2_RING) * 32;
                if (lane < 16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)hs, 16, 0, 16 /* sc1 */);
            }
        };
        for (int pass = 0; pass < 2; ++pass) { stage_pass(pass); __syncthreads(); __syncthreads(); }
        for (int r = 0; r < S2_PD && r < nrows; ++r) issue_row(r);
        s2_wait_vmcnt<0>();
        s2_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + S2_PD;
            if (nx < nrows) {
                issue_row(nx);
                // rows it+2 .. it+PD may stay in flight: (PD-1) * loads-per-row younger than row it+1's
                if (has_producer) s2_wait_vmcnt<(S2_PD - 1) * (DPR + 1)>();
                else s2_wait_vmcnt<(S2_PD - 1) * DPR>();
            } else {
                s2_wait_vmcnt<0>();
            }
            s2_barrier();
        }
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(NT + 64, 3) void dag_strip2_kernel(StripParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = 2 * NT;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);          // 16-byte header; everything else starts at +16
    const int tid = threadIdx.x;
    if (tid == 0) *s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = *s_ticket;
    const int per = p.ndir * p.B;
    const int so = (int)(ticket / per);
    const int rem = (int)(ticket % per);
    const bool is_beta = (MODE == 0) && (p.alpha == nullptr || (p.ndir == 2 && rem >= p.B));
    const int b = rem % p.B;
    const int dirslot = (p.ndir == 2 && rem >= p.B) ? 1 : 0;
    const int s = is_beta ? (p.NS - 1 - so) : so;
    const int j0 = s * W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    u64* census = nullptr;
    if (p.dbg && tid == 0) {                          // residency census (DSP_DEBUG=census): hw id + start clock per workgroup
        census = p.halo + (size_t)p.ndir * p.B * p.NS * T * 32 + (size_t)ticket * 4;
        u32 hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        census[0] = ((u64)xcc << 32) | hwid;
        census[1] = __builtin_readcyclecounter();
        census[2] = wall_clock64();
    }
    if (!valid || j0 >= Lb) {
        if (tid < NT) {
            const int j = j0 + 2 * tid;
            if (j < L) {
                float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
                for (int t = 0; t < T; ++t) {
                    *reinterpret_cast<float2*>(O + (size_t)t * L + j) = make_float2(NEG_INF, NEG_INF);
                    if (MODE == 1) *reinterpret_cast<int2*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int2(-1, -1);
                }
            }
        }
        return;
    }
    __syncthreads();                               // everyone has read the ticket before the tile overlays the header area
    if (MODE == 0 && is_beta) strip2_body<NT, MODE, true>(p, smem_raw + 16, b, s, dirslot, so);
    else strip2_body<NT, MODE, false>(p, smem_raw + 16, b, s, dirslot, so);
    if (census) census[3] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip2_supported(const void* match, const void* alpha, const void* beta, const void* trace, int L, int TR)
{
    if (TR > 32 || (L & 3)) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta | (uintptr_t)trace;
    return (a & 15) == 0;
}

template <int NT, int MODE>
static int launch_strip2(const StripParams& p, int nwg, hipStream_t st)
{
    constexpr int W = 2 * NT, RL = W + 32;
    const size_t lds_main = (size_t)(6 * RL + S2_RING * W) * 4 + (size_t)S2_RING * 32 * 8;
    const size_t lds_tile = (size_t)(NT + 34) * 33 * 4;
    size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    { static const char* const e = getenv("DSP_S2_LDS"); if (e) lds = (size_t)atoi(e); }
    auto k = dag_strip2_kernel<NT, MODE>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    static const char* const e_dbg = getenv("DSP_DEBUG");
    if (e_dbg) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k, NT + 64, lds);
        fprintf(stderr, "[dsp] strip2<%d,%d>: lds=%zu bytes, occupancy API = %d blocks/CU, grid=%d\n", NT, MODE, lds, nb, nwg);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(NT + 64), lds, st, p);
    return check_launch(MODE == 0 ? "dag_loss_fwd(strip2)" : "dag_best_alignment(strip2)");
}

int launch_dag_strip2(int mode, const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                      float* alpha, float* beta, int32_t* trace, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (mode == 0 && alpha && beta) ? 2 : 1;
    // 192 compute lanes + 64 loader lanes = 4 waves per workgroup = ONE WAVE PER SIMD per workgroup: with 5-wave
    // workgroups the hardware admitted a single workgroup per CU although the occupancy API reported 2 (census in
    // tools/census.py); 4-wave workgroups stack 3 deep at 168 VGPRs.
    constexpr int NT = 192, W = 2 * NT;
    const int NS = (L + W - 1) / W;
    StripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta; p.trace = trace;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.dbg = 0;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * 32 * sizeof(u64);
    const int nwg = ndir * B * NS;
    static const char* const dbg = getenv("DSP_DEBUG");
    const bool census = dbg && !strcmp(dbg, "census");
    p.dbg = census ? 1 : 0;
    int rc = banded_acquire_ws(st, halo_bytes + (census ? (size_t)nwg * 32 : 0), T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    rc = mode == 0 ? launch_strip2<NT, 0>(p, nwg, st) : launch_strip2<NT, 1>(p, nwg, st);
    if (census && rc == 0) {
        std::vector<u64> h((size_t)nwg * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), p.halo + (size_t)ndir * B * NS * T * 32, h.size() * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen("/tmp/census.txt", "w");
        if (f) { for (int i = 0; i < nwg; ++i) fprintf(f, "%d %llx %llu %llu %llu\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]); fclose(f); }
    }
    return rc;
}

}  // namespace dsp
