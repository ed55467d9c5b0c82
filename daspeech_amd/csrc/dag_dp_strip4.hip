// dag_dp_strip4.hip — the banded (TR <= 32) DAG DP fast path for gfx950: K2 alpha || K3 beta in EXP SPACE, K6 max-DP.
//
// Replaces calculate_alpha_kernel / calculate_beta_kernel (DASpeech/custom_ops/dag_loss.cu:40-140,178-274) and
// calculate_maxalpha_kernel (dag_best_alignment.cu:39-130).  Same strip / tagged-granule / ticket structure as
// dag_dp_banded.hip (read its header first); what is new here:
//
//   * 4 COLUMNS PER LANE, EXP-SPACE RECURRENCE.  A lane owns 4 adjacent vertices and keeps E = 2^(link - lmax) of their
//     4 x 32 incoming (alpha) / outgoing (beta) edges in registers.  Per row it reads its 36-value window of the
//     previous row a2 (= alpha * log2 e, kept in LDS), rescales it ONCE against a per-lane reference,
//         w[q] = 2^(a2[q] - ref),   ref = min over the lane's 4 columns of (max a2 over that column's predecessors),
//     and every cell is then 32 FMAs: 36 + 4 transcendentals per lane-row instead of 4 x 32 x 2.
//     The reference point is the SMALLEST of the four per-column maxima, so every column's dominant predecessor maps to
//     >= 1.0 — next to the diagonal adjacent columns of a row differ by 30+ binades and a shared maximum would flush
//     the left columns.
//   * EXACTNESS GUARD.  E <= 1 by construction.  A cell whose scaled sum S is below 2^-97 (or whose window spans more
//     than 2^120) may have lost terms that matter to fp32 and is recomputed exactly in log space from the a2 row and
//     the raw links; otherwise everything flushed is < 2^-24 of S.  Rare: needs ~65 nats between a column's largest
//     predecessor term and the rest (emission cliffs), never on smooth data.
//   * WAVE SPECIALISATION, NO vmcnt STALLS ON THE CRITICAL PATH.  Compute waves touch global memory only with stores.
//       - loader wave : streams match rows into an 8-slot LDS ring with global_load_lds (LDS-DMA, no VGPRs), 7 rows
//                       ahead, retired with a COUNTED s_waitcnt vmcnt(N) — never 0 in steady state;
//       - fetch wave  : polls the neighbour strip's halo granules in chunks of 8 rows (one memory round trip per 8 rows);
//       - publish wave: stores this strip's boundary columns as {tag,value} sc1 granules; it never waits.
//     One raw s_barrier per DP row for everybody.
#include "common.h"

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));

struct StripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int dbg;
};

constexpr int S4_TRP = 32;
constexpr int S4_RING = 8;
constexpr int S4_CH = 8;                      // halo chunk (rows per memory round trip of the fetch wave)
constexpr int NEGSENT = -(1 << 30);        // "dead" exponent; far below any finite fp32 score
constexpr u32 S4_SPIN_LIMIT = 1u << 22;
constexpr float S4_LOG2E = 1.4426950408889634f;
constexpr float S4_LN2 = 0.6931471805599453f;

__device__ __forceinline__ u64 s4_gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void s4_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void s4_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// window element index of (column c, distance d):  alpha: predecessor j+c-d -> q = 32 + c - d ; beta: successor -> q = c + d
template <bool BETA> __device__ __forceinline__ constexpr int qidx(int c, int d) { return BETA ? (c + d) : (32 + c - d); }

template <int NT, int MODE, bool BETA>
__device__ __forceinline__ void strip4_body(const StripParams& p, char* smem_raw, int b, int s, int dirslot, int so)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8, NCW = NT / 64, DPR = W / 256;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  a2 (MODE 0, log2 domain) / alpha_max (MODE 1)
    float* Pbuf = Abuf + 2 * RL;                               // [2][RL]  mantissa 2^(a2 - ceil a2) in (0.5, 1]   (MODE 0)
    int* Cbuf = reinterpret_cast<int*>(Pbuf + 2 * RL);         // [2][RL]  integer exponent ceil(a2)               (MODE 0)
    float* Mring = reinterpret_cast<float*>(Cbuf + 2 * RL);    // [RING][W] match rows
    (void)GL;

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
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * S4_TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * S4_TRP;
    // LDS geometry: alpha li = col - j0 + 32 (halo [0,32)); beta li = col - j0 (halo [W, W+32))
    const int halo_li0 = BETA ? W : 0;
    const int own_li0 = BETA ? 0 : 32;
    const int pub_li0 = BETA ? 0 : W;          // boundary columns handed to the consumer: alpha last 32, beta first 32

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once), then -> registers ----
    // tile[r][d] = links[rlo + r][d] (pitch 33), -inf outside the graph / beyond TR.  The tile overlays the main-loop
    // buffers, which are not live yet.
    {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = NT + 192, RPP = NTHR / 32;       // rows per pass
        const int rlo = BETA ? j0 : (j0 - 32);
        const int dd = tid & 31, r0 = tid >> 5;
        for (int rb = r0; rb < W + 32; rb += 8 * RPP) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                       // 8 independent (clamped, unconditional) loads in flight
                const int i = rlo + rb + u * RPP;
                const bool ok = dd < TR && i >= 0 && i < L;
                const float raw = K[(size_t)(ok ? i : 0) * TR + (ok ? dd : 0)];
                v[u] = ok ? raw : NEG_INF;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = rb + u * RPP; if (r < W + 32) tile[r * 33 + dd] = v[u]; }
        }
    }
    __syncthreads();

    if (wave < NCW) {
        // =========================================================== compute waves
        const int l = tid;                       // lane's group
        const int j = j0 + 4 * l;
        const bool col_ok = j < L;
        // structural reachability (cells outside are -inf in the reference too: their LSE runs over -inf terms only):
        // alpha: t <= col <= min(L_b-1, t*TR);  beta: col >= t, col < L_b, L_b-1-col <= (T_b-1-t)*TR
        auto cell_active = [&](int col, int t) -> bool {
            if (col < t || col >= Lb) return false;
            if (!BETA) return (long)col <= (long)t * TR;
            return (long)(Lb - 1 - col) <= (long)(Tb - 1 - t) * TR;
        };
        float E[4][32];
        float lmax[4];
        const float* tile = reinterpret_cast<const float*>(smem_raw);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float raw[32];
            float mx = NEG_INF;
#pragma unroll
            for (int d = 1; d <= 32; ++d) {
                float v;
                if (!BETA) v = tile[(4 * l + c - d + 32) * 33 + (d - 1)];
                else { v = tile[(4 * l + c) * 33 + (d - 1)]; if (j + c + d >= Lb) v = NEG_INF; }
                raw[d - 1] = (MODE == 0) ? v * S4_LOG2E : v;
                mx = fmaxf(mx, raw[d - 1]);
            }
            if (MODE == 0) {
                if (mx == NEG_INF) mx = 0.f;
                lmax[c] = mx;
#pragma unroll
                for (int d = 0; d < 32; ++d) E[c][d] = __builtin_amdgcn_exp2f(raw[d] - mx);
            } else {
                lmax[c] = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) E[c][d] = raw[d];
            }
        }
        // MODE 0: pair layout for v_pk_fma_f32 — E2[c][i] = (weight of window element 2i, weight of 2i+1) for column c,
        // zero where the element is not a predecessor of that column; the window is consumed as 18 (w[2i], w[2i+1]) pairs.
        v2f E2[4][18];
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int q0 = 2 * i, q1 = 2 * i + 1;
                    const int d0 = BETA ? (q0 - c) : (32 + c - q0), d1 = BETA ? (q1 - c) : (32 + c - q1);
                    E2[c][i].x = (d0 >= 1 && d0 <= 32) ? E[c][(d0 >= 1 && d0 <= 32) ? d0 - 1 : 0] : 0.f;
                    E2[c][i].y = (d1 >= 1 && d1 <= 32) ? E[c][(d1 >= 1 && d1 <= 32) ? d1 - 1 : 0] : 0.f;
                }
            }
        }
        auto Eval = [&](int c, int d) -> float {        // E(c, d) recovered from the pair layout (static indices only)
            const int q = BETA ? (c + d) : (32 + c - d);
            return (q & 1) ? E2[c][q >> 1].y : E2[c][q >> 1].x;
        };
        __syncthreads();                         // tile consumed: the loader may start filling the ring over it
        s4_barrier();                            // prologue barrier: match row 0 is in the ring

        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            const float4 mt = *reinterpret_cast<const float4*>(Mring + (size_t)(it % S4_RING) * W + 4 * l);
            float m2[4] = {mt.x, mt.y, mt.z, mt.w};
            float a2[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
            int arg[4] = {-1, -1, -1, -1};
            if (it == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool seed = BETA ? (j + c == Lb - 1) : (j + c == 0);
                    if (seed) a2[c] = (MODE == 0) ? m2[c] * S4_LOG2E : m2[c];
                }
            } else if (MODE == 0) {
                int cw[36];
                float pw[36];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int4 ci = *reinterpret_cast<const int4*>(Cbuf + prv * RL + 4 * l + 4 * k);
                    const float4 pv = *reinterpret_cast<const float4*>(Pbuf + prv * RL + 4 * l + 4 * k);
                    cw[4 * k] = ci.x; cw[4 * k + 1] = ci.y; cw[4 * k + 2] = ci.z; cw[4 * k + 3] = ci.w;
                    pw[4 * k] = pv.x; pw[4 * k + 1] = pv.y; pw[4 * k + 2] = pv.z; pw[4 * k + 3] = pv.w;
                }
                // per-column maximum exponent over that column's predecessors: alpha q in [c, c+31], beta q in [c+1, c+32]
                int common = cw[4];
#pragma unroll
                for (int q = 5; q <= 31; ++q) common = max(common, cw[q]);
                int cm[4];
                if (!BETA) {
                    cm[0] = max(max(common, cw[0]), max(max(cw[1], cw[2]), cw[3]));
                    cm[1] = max(max(common, cw[32]), max(max(cw[1], cw[2]), cw[3]));
                    cm[2] = max(max(common, cw[32]), max(max(cw[33], cw[2]), cw[3]));
                    cm[3] = max(max(common, cw[32]), max(max(cw[33], cw[34]), cw[3]));
                } else {
                    const int c32 = max(common, cw[32]);
                    cm[0] = max(c32, max(max(cw[1], cw[2]), cw[3]));
                    cm[1] = max(max(c32, cw[33]), max(cw[2], cw[3]));
                    cm[2] = max(max(c32, cw[33]), max(cw[34], cw[3]));
                    cm[3] = max(max(c32, cw[33]), max(cw[34], cw[35]));
                }
                const int hi = max(max(cm[0], cm[1]), max(cm[2], cm[3]));
                int refi = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (cm[c] != NEGSENT) refi = min(refi, cm[c]);
                const bool any_live = hi != NEGSENT;
                if (!any_live) refi = 0;
                const bool wide = (hi - refi) > 120;               // the four column maxima are > 2^120 apart
                // one full-rate v_ldexp per window element (no transcendental): w = P * 2^(C - ref)
                v2f S2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { S2[c].x = 0.f; S2[c].y = 0.f; }
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    v2f w2;
                    w2.x = ldexpf(pw[2 * i], cw[2 * i] - refi);
                    w2.y = ldexpf(pw[2 * i + 1], cw[2 * i + 1] - refi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) S2[c] = __builtin_elementwise_fma(w2, E2[c][i], S2[c]);
                }
                float S[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) S[c] = S2[c].x + S2[c].y;
                const float ref = (float)refi;
                // all four cells unconditionally (independent FMA chains stay interleaved); masks applied by selects
                bool need_fb = false;
                const bool R_live = any_live;
                bool flag[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float cand = __builtin_amdgcn_logf(S[c]) + ref + lmax[c] + m2[c] * S4_LOG2E;
                    const bool okc = cell_active(j + c, t) & (cm[c] != NEGSENT);
                    flag[c] = okc & (wide | !(S[c] >= 0x1p-97f));
                    a2[c] = (okc & !flag[c]) ? cand : NEG_INF;
                    need_fb |= flag[c];
                }
                if (__builtin_expect(need_fb, 0)) {
                    // (a) MEDIUM path, registers only: redo the flagged column against ITS OWN maximum (covers windows whose
                    //     four column maxima are > 2^120 apart — the diagonal at large t).  Falls through to the exact
                    //     path only if the column's own sum is still below the exactness threshold.
                    float aw[36];
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const float4 v = *reinterpret_cast<const float4*>(Abuf + prv * RL + 4 * l + 4 * k);
                        aw[4 * k] = v.x; aw[4 * k + 1] = v.y; aw[4 * k + 2] = v.z; aw[4 * k + 3] = v.w;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (flag[c]) {
                            float cmx = NEG_INF;
#pragma unroll
                            for (int d = 1; d <= 32; ++d) cmx = fmaxf(cmx, aw[qidx<BETA>(c, d)]);
                            float sc = 0.f;
#pragma unroll
                            for (int d = 1; d <= 32; ++d)
                                sc = fmaf(__builtin_amdgcn_exp2f(aw[qidx<BETA>(c, d)] - cmx), Eval(c, d), sc);
                            S[c] = sc;
                            if (sc >= 0x1p-97f) a2[c] = __builtin_amdgcn_logf(sc) + cmx + lmax[c] + m2[c] * S4_LOG2E;
                        } else {
                            S[c] = 1.f;                      // settled by the fast path (or inactive)
                        }
                    }
                    // (b) EXACT path for what is left
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        { const int cmc = (c == 0) ? cm[0] : (c == 1) ? cm[1] : (c == 2) ? cm[2] : cm[3];
                          const float Sc = (c == 0) ? S[0] : (c == 1) ? S[1] : (c == 2) ? S[2] : S[3];
                          if (!(cell_active(j + c, t) && R_live && cmc != NEGSENT && !(Sc >= 0x1p-97f))) continue; }
                        // (1) cheap structural test: is any predecessor alive (a2 row in LDS)?  if not the cell is -inf
                        float amax = NEG_INF;
                        for (int d = 1; d <= 32; ++d) amax = fmaxf(amax, Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))]);
                        float r = NEG_INF;
                        if (amax != NEG_INF) {
                            // (2) exact log-space value; raw links re-read from HBM 8 at a time (independent loads)
                            { const u32 slot = atomicAdd(&p.counters[2], 1u); if (slot < 14) { p.counters[8 + 4 * slot] = (u32)b; p.counters[9 + 4 * slot] = (u32)t; p.counters[10 + 4 * slot] = (u32)(j + c); p.counters[11 + 4 * slot] = (u32)refi; } }
                            float mx = NEG_INF, sum = 0.f;
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                float lk[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const int row = BETA ? (j + c) : (j + c - d);
                                    const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                    const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                    lk[u] = ok ? raw * S4_LOG2E : NEG_INF;
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const float v = Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))] + lk[u];
                                    const float nm = fmaxf(mx, v);
                                    if (nm != NEG_INF) sum = sum * __builtin_amdgcn_exp2f(mx - nm) + __builtin_amdgcn_exp2f(v - nm);
                                    mx = nm;
                                }
                            }
                            if (mx != NEG_INF) {
                                const float mm = (c == 0) ? m2[0] : (c == 1) ? m2[1] : (c == 2) ? m2[2] : m2[3];
                                r = __builtin_amdgcn_logf(sum) + mx + mm * S4_LOG2E;
                            }
                        }
                        if (c == 0) a2[0] = r; else if (c == 1) a2[1] = r; else if (c == 2) a2[2] = r; else a2[3] = r;
                    }
                }
            } else {
                // MODE 1: max-DP in the natural domain; ascending predecessor index, strict '>' (smallest index wins ties)
                float mxv[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
                int av[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(Abuf + prv * RL + 4 * l + 4 * k);
                    const float wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = 4 * k + i;               // ascending q == ascending predecessor index
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int d = 32 + c - q;
                            if (d >= 1 && d <= 32) {
                                const float x = wv[i] + E[c][d - 1];
                                if (x > mxv[c]) { mxv[c] = x; av[c] = j + c - d; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool act = (j + c >= t) && (j + c < Lb);     // (no reach mask here: bit-exact trace incl. -1s)
                    if (act) { a2[c] = mxv[c] + m2[c]; arg[c] = av[c]; }
                }
            }
            // ---- write the row: LDS state for the next row, HBM output ----
            if (MODE == 0) {
                float pn[4]; int cn[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool dead = a2[c] == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(a2[c]);
                    pn[c] = __builtin_amdgcn_exp2f(a2[c] - cf);
                    cn[c] = dead ? NEGSENT : (int)cf;
                }
                *reinterpret_cast<float4*>(Pbuf + cur * RL + own_li0 + 4 * l) = make_float4(pn[0], pn[1], pn[2], pn[3]);
                *reinterpret_cast<int4*>(Cbuf + cur * RL + own_li0 + 4 * l) = make_int4(cn[0], cn[1], cn[2], cn[3]);
            }
            *reinterpret_cast<float4*>(Abuf + cur * RL + own_li0 + 4 * l) = make_float4(a2[0], a2[1], a2[2], a2[3]);
            if (col_ok) {
                float4 o;
                if (MODE == 0) o = make_float4(a2[0] * S4_LN2, a2[1] * S4_LN2, a2[2] * S4_LN2, a2[3] * S4_LN2);
                else o = make_float4(a2[0], a2[1], a2[2], a2[3]);
                *reinterpret_cast<float4*>(O + (size_t)t * L + j) = o;
                if (MODE == 1) *reinterpret_cast<int4*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int4(arg[0], arg[1], arg[2], arg[3]);
            }
            s4_barrier();
        }
        // rows the recurrence never reaches
        if (col_ok) for (int t = Tb; t < T; ++t) {
            *reinterpret_cast<float4*>(O + (size_t)t * L + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            if (MODE == 1) *reinterpret_cast<int4*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int4(-1, -1, -1, -1);
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * L;
            float* slot = Mring + (size_t)(itr % S4_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 256), 16, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < S4_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s4_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + S4_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                // rows it+2 .. it+7 may stay in flight: 6*DPR DMAs younger than row it+1's
                if (DPR == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (DPR == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s4_barrier();
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: neighbour strip's halo -> LDS
        const bool hl = lane < S4_TRP;
        u64 g[S4_CH];
#pragma unroll
        for (int k = 0; k < S4_CH; ++k) g[k] = 0;
        auto load_chunk = [&](int it0) {
#pragma unroll
            for (int k = 0; k < S4_CH; ++k) {
                const int itr = it0 + k;
                if (itr < nrows && hl) { const int t = BETA ? (Tb - 1 - itr) : itr; g[k] = s4_gran_load(hin + (size_t)t * S4_TRP + lane); }
            }
        };
        if (has_producer) load_chunk(0);
        __syncthreads();                         // link tile consumed
        s4_barrier();                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += S4_CH) {
#pragma unroll
            for (int k = 0; k < S4_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int t = BETA ? (Tb - 1 - it) : it;
                const int cur = it & 1;
                float hv = NEG_INF;
                if (has_producer && hl) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    u64 x = g[k];
                    u32 spins = 0;
                    while (!__all((u32)(x >> 32) == want)) {
                        if ((u32)(x >> 32) != want) x = s4_gran_load(hin + (size_t)t * S4_TRP + lane);
                        if (++spins > S4_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv = __uint_as_float((u32)x);
                }
                if (hl) {
                    Abuf[cur * RL + halo_li0 + lane] = hv;
                    if (MODE == 0) {
                        const bool dead = hv == NEG_INF;
                        const float cf = dead ? 0.f : ceilf(hv);
                        Pbuf[cur * RL + halo_li0 + lane] = __builtin_amdgcn_exp2f(hv - cf);
                        Cbuf[cur * RL + halo_li0 + lane] = dead ? NEGSENT : (int)cf;
                    }
                }
                if (k == S4_CH - 1 && has_producer) load_chunk(itb + S4_CH);   // next chunk: one round trip per 8 rows
                s4_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: boundary columns -> granules
        const bool pl = has_consumer && lane < S4_TRP;
        __syncthreads();                         // link tile consumed
        s4_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete (barrier it-1 passed); compute now writes the other buffer
                const int tp = BETA ? (Tb - it) : (it - 1);
                const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
                s4_gran_store(hout + (size_t)tp * S4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
            }
            s4_barrier();
        }
        if (pl && nrows > 0) {
            const int it = nrows;
            const int tp = BETA ? (Tb - it) : (it - 1);
            const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
            s4_gran_store(hout + (size_t)tp * S4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
        }
        (void)pub_li0;
    }
}

template <int NT, int MODE>
__global__ __launch_bounds__(NT + 192) void dag_strip4_kernel(StripParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);          // 16-byte header; everything else starts at +16
    (void)RL; (void)GL;
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
    if (!valid || j0 >= Lb) {                    // nothing reachable in this strip: -inf everywhere, no hand-off
        if (tid < NT) {
            const int j = j0 + 4 * tid;
            if (j < L) {
                float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
                for (int t = 0; t < T; ++t) {
                    *reinterpret_cast<float4*>(O + (size_t)t * L + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
                    if (MODE == 1) *reinterpret_cast<int4*>(p.trace + (size_t)b * T * L + (size_t)t * L + j) = make_int4(-1, -1, -1, -1);
                }
            }
        }
        return;
    }
    if (MODE == 0 && is_beta) strip4_body<NT, MODE, true>(p, smem_raw + 16, b, s, dirslot, so);
    else strip4_body<NT, MODE, false>(p, smem_raw + 16, b, s, dirslot, so);
}

// ------------------------------------------------------------------------------------------------ host side
struct BandedWS;
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip4_supported(const void* match, const void* alpha, const void* beta, const void* trace, int L, int TR)
{
    if (TR > 32 || (L & 3)) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta | (uintptr_t)trace;
    return (a & 15) == 0;
}

template <int NT, int MODE>
static int launch_one(const StripParams& p, int nwg, hipStream_t st)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8;
    const size_t lds_main = (size_t)(6 * RL + S4_RING * W) * 4 + 16; (void)GL;
    const size_t lds_tile = (size_t)(W + 32) * 33 * 4;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    auto k = dag_strip4_kernel<NT, MODE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(NT + 192), lds, st, p);
    return check_launch(MODE == 0 ? "dag_loss_fwd(strip4)" : "dag_best_alignment(strip4)");
}

int launch_dag_strip4(int mode, const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                      float* alpha, float* beta, int32_t* trace, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (mode == 0 && alpha && beta) ? 2 : 1;
    // strip width: 1024 columns when that still yields >= ~200 workgroups, else 512
    const int ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
    const bool wide = (long)ndir * B * ns1024 >= 200;
    const int NS = wide ? ns1024 : ns512;
    StripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta; p.trace = trace;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.dbg = 0;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * S4_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const int nwg = ndir * B * NS;
    if (mode == 0) return wide ? launch_one<256, 0>(p, nwg, st) : launch_one<128, 0>(p, nwg, st);
    return wide ? launch_one<256, 1>(p, nwg, st) : launch_one<128, 1>(p, nwg, st);
}

}  // namespace dsp
