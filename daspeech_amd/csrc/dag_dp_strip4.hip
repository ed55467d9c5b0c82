// dag_dp_strip4.hip — the banded (TR <= 32) DAG DP fast path for gfx950: K2 alpha || K3 beta in EXP SPACE, K6 max-DP.
//
// Replaces calculate_alpha_kernel / calculate_beta_kernel (DASpeech/custom_ops/dag_loss.cu:40-140,178-274) and
// calculate_maxalpha_kernel (dag_best_alignment.cu:39-130).  Same strip / tagged-granule / ticket structure as
// dag_dp_banded.hip (read its header first); what is new here:
//
//   * 4 COLUMNS PER LANE, EXP-SPACE RECURRENCE.  A lane owns 4 adjacent vertices and keeps E = 2^(link) of their
//     4 x 32 incoming (alpha) / outgoing (beta) edges in registers.  The previous row is kept in LDS as
//         P[k] = 2^(a2[k] - c[g])   with one integer scale c[g] per GROUP OF 4 COLUMNS (= per lane, no reduction),
//     so a cell costs 32 FMAs + (36 v_ldexp + 9 integer ops)/4 instead of 32 x (add, max, sub, exp, add):
//     2 transcendentals per cell instead of 32.  (a2 = alpha * log2 e.)
//   * EXACTNESS GUARD.  P, E <= 1 by construction.  If a cell's scaled sum S falls below 2^-90 some term that matters
//     to fp32 may have been flushed, so the cell is recomputed in log space from the a2 row that is also kept in LDS;
//     otherwise every flushed term is < 2^-36 of S.  Rare (needs > 62 nats between a cell and its 36-column window).
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

struct StripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
};

constexpr int S4_TRP = 32;
constexpr int S4_RING = 8;
constexpr int S4_CH = 8;                      // halo chunk (rows per memory round trip of the fetch wave)
constexpr int NEGSENT = -100000;
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
    float* Pbuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  scaled linear values (MODE 0)
    float* Abuf = Pbuf + 2 * RL;                               // [2][RL]  a2 (MODE 0, log2 domain) / alpha_max (MODE 1)
    int* Cbuf = reinterpret_cast<int*>(Abuf + 2 * RL);         // [2][GL]  group scales
    float* Mring = reinterpret_cast<float*>(Cbuf + 2 * GL);    // [RING][W] match rows

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
    const int halo_li0 = BETA ? W : 0, halo_g0 = BETA ? NT : 0;
    const int own_li0 = BETA ? 0 : 32, own_g0 = BETA ? 0 : 8;
    const int pub_li0 = BETA ? 0 : W;          // boundary columns handed to the consumer: alpha last 32, beta first 32

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once), then -> registers ----
    // tile[r][d] = links[rlo + r][d] (pitch 33), -inf outside the graph / beyond TR.  The tile overlays the main-loop
    // buffers, which are not live yet.
    {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = NT + 192, RPP = NTHR / 32;       // rows per pass
        const int rlo = BETA ? j0 : (j0 - 32);
        const int dd = tid & 31, r0 = tid >> 5;
        for (int r = r0; r < W + 32; r += RPP) {
            const int i = rlo + r;
            float v = NEG_INF;
            if (dd < TR && i >= 0 && i < L) v = K[(size_t)i * TR + dd];
            tile[r * 33 + dd] = v;
        }
    }
    __syncthreads();

    if (wave < NCW) {
        // =========================================================== compute waves
        const int l = tid;                       // lane's group
        const int j = j0 + 4 * l;
        const bool col_ok = j < L;
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
                int cg[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) cg[k] = Cbuf[prv * GL + l + k];
                int ref = cg[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) ref = max(ref, cg[k]);
                // window walked group by group (4 live values instead of 36): element q of the window feeds the cells
                // (c, d) with qidx(c, d) == q
                float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float4 v = *reinterpret_cast<const float4*>(Pbuf + prv * RL + 4 * l + 4 * k);
                    const int e = max(cg[k] - ref, -250);
                    const float wv[4] = {ldexpf(v.x, e), ldexpf(v.y, e), ldexpf(v.z, e), ldexpf(v.w, e)};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = 4 * k + i;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int d = BETA ? (q - c) : (32 + c - q);
                            if (d >= 1 && d <= 32) S[c] = fmaf(wv[i], E[c][d - 1], S[c]);
                        }
                    }
                }
                const float reff = (float)ref;
                bool need_fb = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool act = (j + c >= t) && (j + c < Lb);
                    if (act && ref != NEGSENT) {
                        if (S[c] < 0x1p-90f) need_fb = true;
                        else a2[c] = __builtin_amdgcn_logf(S[c]) + reff + lmax[c] + m2[c] * S4_LOG2E;
                    }
                }
                if (__builtin_expect(need_fb, 0)) {
                    // exact log-space recomputation of the flagged cells (rare): a2 row from LDS, raw links from HBM
                    // (E may have flushed links that are far below the column's largest one)
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        const bool act = (j + c >= t) && (j + c < Lb);
                        if (!(act && ref != NEGSENT && S[c] < 0x1p-90f)) continue;
                        float mx = NEG_INF;
                        for (int d = 1; d <= 32; ++d) {
                            const float av = Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))];
                            float lk = NEG_INF;
                            if (!BETA) { const int i = j + c - d; if (d <= TR && i >= 0) lk = K[(size_t)i * TR + (d - 1)] * S4_LOG2E; }
                            else { if (d <= TR && j + c + d < Lb) lk = K[(size_t)(j + c) * TR + (d - 1)] * S4_LOG2E; }
                            mx = fmaxf(mx, av + lk);
                        }
                        float r = NEG_INF;
                        if (mx != NEG_INF) {
                            float sum = 0.f;
                            for (int d = 1; d <= 32; ++d) {
                                const float av = Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))];
                                float lk = NEG_INF;
                                if (!BETA) { const int i = j + c - d; if (d <= TR && i >= 0) lk = K[(size_t)i * TR + (d - 1)] * S4_LOG2E; }
                                else { if (d <= TR && j + c + d < Lb) lk = K[(size_t)(j + c) * TR + (d - 1)] * S4_LOG2E; }
                                sum += __builtin_amdgcn_exp2f(av + lk - mx);
                            }
                            const float mm = (c == 0) ? m2[0] : (c == 1) ? m2[1] : (c == 2) ? m2[2] : m2[3];
                            r = __builtin_amdgcn_logf(sum) + mx + mm * S4_LOG2E;
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
                    const bool act = (j + c >= t) && (j + c < Lb);
                    if (act) { a2[c] = mxv[c] + m2[c]; arg[c] = av[c]; }
                }
            }
            // ---- write the row: LDS state for the next row, HBM output ----
            if (MODE == 0) {
                const float mx = fmaxf(fmaxf(a2[0], a2[1]), fmaxf(a2[2], a2[3]));
                int cn = NEGSENT; float cf = 0.f;
                if (mx != NEG_INF) { cf = ceilf(mx); cn = (int)cf; }
                float4 pv;
                pv.x = __builtin_amdgcn_exp2f(a2[0] - cf); pv.y = __builtin_amdgcn_exp2f(a2[1] - cf);
                pv.z = __builtin_amdgcn_exp2f(a2[2] - cf); pv.w = __builtin_amdgcn_exp2f(a2[3] - cf);
                *reinterpret_cast<float4*>(Pbuf + cur * RL + own_li0 + 4 * l) = pv;
                Cbuf[cur * GL + own_g0 + l] = cn;
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
                        float mx = fmaxf(hv, __shfl_xor(hv, 1, 64));
                        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                        int cn = NEGSENT; float cf = 0.f;
                        if (mx != NEG_INF) { cf = ceilf(mx); cn = (int)cf; }
                        Pbuf[cur * RL + halo_li0 + lane] = __builtin_amdgcn_exp2f(hv - cf);
                        if ((lane & 3) == 0) Cbuf[cur * GL + halo_g0 + (lane >> 2)] = cn;
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
    const size_t lds_main = (size_t)(4 * RL + 2 * GL + S4_RING * W) * 4 + 16;
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
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * S4_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const int nwg = ndir * B * NS;
    if (mode == 0) return wide ? launch_one<256, 0>(p, nwg, st) : launch_one<128, 0>(p, nwg, st);
    return wide ? launch_one<256, 1>(p, nwg, st) : launch_one<128, 1>(p, nwg, st);
}

}  // namespace dsp
