// dag_dp_strip4h.hip — banded (TR <= 32) DAG DP, K2 alpha || K3 beta: strip4g with TWO COMPUTE WAVES PER SIMD.
//
// Same row representation as dag_dp_strip4g.hip (one exponent per lane group of 4 vertices, values stored as plain fp32; read
// that header first) and the same strip / tagged-granule / ticket / helper-wave structure.  What changes: the C2 problem has
// exactly one 4-vertices-per-lane compute wave per SIMD, so in strip4g every LDS round trip, dependent-VALU bubble and barrier
// of the row is exposed (VALU busy ~50 %).  Here each lane group is served by TWO waves that split the 32 transitions of a
// vertex:
//     half 0 ("finisher"): transitions d = 1..16  -> window elements of groups 4..8 (alpha) / 0..4 (beta), 40 v_pk_fma
//     half 1             : transitions d = 17..32 -> the other five groups, 40 v_pk_fma
// Half 1 leaves its four partial sums and its reference exponent in LDS; after a workgroup barrier half 0 merges them with its
// own, runs the row tail (log-domain value, exactness guard, next row's V / X) and writes the row.  Waves w and w + NT/64 land
// on the same SIMD (waves are dealt to SIMDs cyclically), so the two halves of a lane group share a SIMD and fill each
// other's stalls.  Two workgroup barriers per DP row.
//
// Exactness guard as in strip4g, except that a flagged cell goes straight to the exact log-space path (the register-only
// "medium" path would need all 32 weights of the vertex in one wave).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));

struct HStripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int dbg;
};

constexpr int H4_TRP = 32;
constexpr int H4_RING = 8;
constexpr int H4_CH = 4;                      // halo prefetch distance of the fetch wave (rows)
constexpr int HNEGSENT = -(1 << 30);       // "dead" exponent; far below any finite fp32 score
constexpr u32 H4_SPIN_LIMIT = 1u << 22;
constexpr float H4_LOG2E = 1.4426950408889634f;
constexpr float H4_LN2 = 0.6931471805599453f;

__device__ __forceinline__ u64 h4_gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void h4_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void h4_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

typedef int h_v2i __attribute__((ext_vector_type(2)));
typedef float h_v4f __attribute__((ext_vector_type(4)));

// everything a compute half needs from the body's set-up
struct HCtx {
    float* Abuf; float* Vbuf; int* Xbuf; float* Mring; float* Pbuf; int* Rbuf;
    const float* K; float* O; u32* counters;
    int l, j, j0, Lb, Tb, T, L, TR, nrows, b, dbg;
    bool col_ok;
};

// One half of a lane group's transitions.  H = 0: d in 1..16 (the finisher), H = 1: d in 17..32.
template <int NT, bool BETA, int H>
__device__ __forceinline__ void strip4h_compute(const HCtx& x, const float* tile)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8;
    constexpr int GOFS = BETA ? (H == 0 ? 0 : 16) : (H == 0 ? 16 : 0);      // first window element of this half's 20
    constexpr int DLO = H == 0 ? 1 : 17, DHI = DLO + 15;
    constexpr int own_li0 = BETA ? 0 : 32;
    const int l = x.l, j = x.j, Lb = x.Lb, Tb = x.Tb, TR = x.TR, L = x.L;
    float* Abuf = x.Abuf; float* Vbuf = x.Vbuf; int* Xbuf = x.Xbuf;

    // structural reachability (cells outside are -inf in the reference too: their LSE runs over -inf terms only):
    // alpha: t <= col <= min(L_b-1, t*TR);  beta: col >= t, T_b-1-t <= L_b-1-col <= (T_b-1-t)*TR
    auto cell_active = [&](int col, int t) -> bool {       // (T * TR fits an int: T, L < 2^20 and TR <= 32)
        if (!BETA) return col >= t && col < Lb && col <= t * TR;
        const int rem = Tb - 1 - t, gap = Lb - 1 - col;
        return col >= t && gap >= rem && gap <= rem * TR;
    };

    // ---- weights: E = 2^(link - lmax) for this half's 16 transitions; lmax / flush test over all 32 (both halves agree) ----
    float lmax[4], sthr[4];
    v2f E2[4][10];                              // pair i = window elements GOFS + 2i, GOFS + 2i + 1
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float raw[32];
        float mx = NEG_INF;
#pragma unroll
        for (int d = 1; d <= 32; ++d) {
            float v;
            if (!BETA) v = tile[(4 * l + c - d + 32) * 33 + (d - 1)];
            else { v = tile[(4 * l + c) * 33 + (d - 1)]; if (j + c + d >= Lb) v = NEG_INF; }
            raw[d - 1] = v * H4_LOG2E;
            mx = fmaxf(mx, raw[d - 1]);
        }
        if (mx == NEG_INF) mx = 0.f;
        lmax[c] = mx;
        bool flushed = false;                   // a finite link more than ~120 binades under the vertex's strongest
#pragma unroll
        for (int d = 0; d < 32; ++d) flushed |= (raw[d] != NEG_INF) & (raw[d] - mx < -120.f);
        sthr[c] = flushed ? 0x1p10f : 0x1p-97f; // see strip4g: a dropped weight can cost up to 2^-20 in the scaled domain
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int q0 = GOFS + 2 * i, q1 = q0 + 1;
            const int d0 = BETA ? (q0 - c) : (32 + c - q0), d1 = BETA ? (q1 - c) : (32 + c - q1);
            const bool in0 = d0 >= DLO && d0 <= DHI, in1 = d1 >= DLO && d1 <= DHI;
            E2[c][i].x = in0 ? __builtin_amdgcn_exp2f(raw[in0 ? d0 - 1 : 0] - mx) : 0.f;
            E2[c][i].y = in1 ? __builtin_amdgcn_exp2f(raw[in1 ? d1 - 1 : 0] - mx) : 0.f;
        }
    }
    __syncthreads();                             // tile consumed: the loader may start filling the ring over it
    h4_barrier();                                // prologue barrier: match row 0 is in the ring

    for (int it = 0; it < x.nrows; ++it) {
        const int t = BETA ? (Tb - 1 - it) : it;
        const int cur = it & 1, prv = cur ^ 1;
        float a2[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
        float S[4] = {0.f, 0.f, 0.f, 0.f};
        float base[4] = {0.f, 0.f, 0.f, 0.f};
        float m2[4] = {0.f, 0.f, 0.f, 0.f};
        bool okc[4] = {false, false, false, false};
        int refh = HNEGSENT;
        if (it == 0) {
            if (H == 0) {
                const float4 mt = *reinterpret_cast<const float4*>(x.Mring + (size_t)(it % H4_RING) * W + 4 * l);
                const float mm[4] = {mt.x, mt.y, mt.z, mt.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool seed = BETA ? (j + c == Lb - 1) : (j + c == 0);
                    if (seed) a2[c] = mm[c] * H4_LOG2E;
                }
            }
        } else {
            // ---- row head: this half's LDS reads (five group exponents, 20 window values; the finisher also its match
            // values) leave as one issue group; consumers wait with counted lgkmcnt (LDS returns in order and nothing else
            // is issued to LDS / scalar memory in between).
            h_v4f mt; h_v2i x01, x23; int x4; h_v4f pv[5];
            {
                const u32 maddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(x.Mring + (size_t)(it % H4_RING) * W + 4 * l);
                const u32 xaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Xbuf + prv * GL + l + GOFS / 4);
                const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Vbuf + prv * RL + 4 * l + GOFS);
                asm volatile(
                    "ds_read_b128 %0, %9\n\t"
                    "ds_read2_b32 %1, %10 offset1:1\n\t"
                    "ds_read2_b32 %2, %10 offset0:2 offset1:3\n\t"
                    "ds_read_b32 %3, %10 offset:16\n\t"
                    "ds_read_b128 %4, %11\n\t"
                    "ds_read_b128 %5, %11 offset:16\n\t"
                    "ds_read_b128 %6, %11 offset:32\n\t"
                    "ds_read_b128 %7, %11 offset:48\n\t"
                    "ds_read_b128 %8, %11 offset:64"
                    : "=&v"(mt), "=&v"(x01), "=&v"(x23), "=&v"(x4),
                      "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4])
                    : "v"(maddr), "v"(xaddr), "v"(vaddr)
                    : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(mt));
            if (H == 0) {
                m2[0] = mt.x; m2[1] = mt.y; m2[2] = mt.z; m2[3] = mt.w;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    okc[c] = cell_active(j + c, t);
                    base[c] = lmax[c] + m2[c] * H4_LOG2E;                  // log2(strongest link * emission)
                }
            }
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(x01), "+v"(x23), "+v"(x4));
            const int xw[5] = {x01.x, x01.y, x23.x, x23.y, x4};
            refh = max(max(max(xw[0], xw[1]), max(xw[2], xw[3])), xw[4]);
            const int refi = (refh != HNEGSENT) ? refh : 0;
            float fg[5];
#pragma unroll
            for (int g = 0; g < 5; ++g) fg[g] = ldexpf(1.0f, xw[g] - refi);   // <= 1; 0 for dead groups (ldexp saturates)
            v2f S2[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { S2[c].x = 0.f; S2[c].y = 0.f; }
#define H4_GROUP(k, n) \
            { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(pv[k])); \
              v2f wa, wb; wa.x = pv[k].x * fg[k]; wa.y = pv[k].y * fg[k]; wb.x = pv[k].z * fg[k]; wb.y = pv[k].w * fg[k]; \
              _Pragma("unroll") for (int c = 0; c < 4; ++c) { \
                  S2[c] = __builtin_elementwise_fma(wa, E2[c][2 * k], S2[c]); \
                  S2[c] = __builtin_elementwise_fma(wb, E2[c][2 * k + 1], S2[c]); } }
            H4_GROUP(0, 4) H4_GROUP(1, 3) H4_GROUP(2, 2) H4_GROUP(3, 1) H4_GROUP(4, 0)
#undef H4_GROUP
#pragma unroll
            for (int c = 0; c < 4; ++c) S[c] = S2[c].x + S2[c].y;
            if (H == 1) {
                *reinterpret_cast<float4*>(x.Pbuf + 4 * l) = make_float4(S[0], S[1], S[2], S[3]);
                x.Rbuf[l] = refh;
            }
        }
        h4_barrier();                            // (1) partial sums of half 1 are in LDS
        if (H == 0) {
            if (it > 0) {
                const float4 sb = *reinterpret_cast<const float4*>(x.Pbuf + 4 * l);
                const int rb = x.Rbuf[l];
                int R = max(refh, rb);
                const bool any_live = R != HNEGSENT;
                if (!any_live) R = 0;
                const float fa = ldexpf(1.0f, refh - R), fb = ldexpf(1.0f, rb - R);
                const float SB[4] = {sb.x, sb.y, sb.z, sb.w};
                const float ref = (float)R;
                bool flag[4];
                bool need_fb = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float Sc = fmaf(SB[c], fb, S[c] * fa);
                    const bool okl = okc[c] & any_live;
                    flag[c] = okl & !(Sc >= sthr[c] && Sc <= 0x1p110f);          // too small, NaN (escaped input) or inf
                    need_fb |= flag[c];
                    a2[c] = (okl & !flag[c]) ? (__builtin_amdgcn_logf(Sc) + (ref + base[c])) : NEG_INF;
                }
                if (__builtin_expect(need_fb, 0)) {
                    // EXACT path: the cell in log space from the a2 row and the raw links (re-read from HBM / L2)
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        if (!((c == 0) ? flag[0] : (c == 1) ? flag[1] : (c == 2) ? flag[2] : flag[3])) continue;
                        float amax = NEG_INF;
                        for (int d = 1; d <= 32; ++d) amax = fmaxf(amax, Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))]);
                        float r = NEG_INF;
                        if (amax != NEG_INF) {
                            { const u32 slot = atomicAdd(&x.counters[2], 1u); if (slot < 14) { x.counters[8 + 4 * slot] = (u32)x.b; x.counters[9 + 4 * slot] = (u32)t; x.counters[10 + 4 * slot] = (u32)(j + c); x.counters[11 + 4 * slot] = (u32)R; } }
                            float mx = NEG_INF, sum = 0.f;
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                float lk[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const int row = BETA ? (j + c) : (j + c - d);
                                    const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                    const float rawl = x.K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                    lk[u] = ok ? rawl * H4_LOG2E : NEG_INF;
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
                                r = __builtin_amdgcn_logf(sum) + mx + mm * H4_LOG2E;
                            }
                        }
                        if (c == 0) a2[0] = r; else if (c == 1) a2[1] = r; else if (c == 2) a2[2] = r; else a2[3] = r;
                    }
                }
            }
            // ---- write the row: LDS state for the next row, HBM output ----
            // group exponent X = ceil(largest of the four) - 100, so V = 2^(a2 - X) spans (2^-120, 2^100]: a live vertex more
            // than 220 binades below its group's maximum is "escaped" (NaN; its exact value is in the a2 row)
            const float amax = fmaxf(fmaxf(a2[0], a2[1]), fmaxf(a2[2], a2[3]));
            const bool dead = amax == NEG_INF;
            const float cf = dead ? 0.f : ceilf(amax) - 100.f;
            float vn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float e = a2[c] - cf;
                const float v = __builtin_amdgcn_exp2f(e);
                vn[c] = v;                 // flushes to 0 more than 226 binades under the group maximum — as the scaling would
            }
            *reinterpret_cast<float4*>(Vbuf + cur * RL + own_li0 + 4 * l) = make_float4(vn[0], vn[1], vn[2], vn[3]);
            Xbuf[cur * GL + (own_li0 >> 2) + l] = dead ? HNEGSENT : (int)cf;
            *reinterpret_cast<float4*>(Abuf + cur * RL + own_li0 + 4 * l) = make_float4(a2[0], a2[1], a2[2], a2[3]);
            if (x.col_ok)
                *reinterpret_cast<float4*>(x.O + (size_t)t * L + j) = make_float4(a2[0] * H4_LN2, a2[1] * H4_LN2, a2[2] * H4_LN2, a2[3] * H4_LN2);
        }
        h4_barrier();                            // (2) the row is complete
    }
    // rows the recurrence never reaches
    if (H == 0 && x.col_ok) for (int t = Tb; t < x.T; ++t)
        *reinterpret_cast<float4*>(x.O + (size_t)t * L + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
}

template <int NT, bool BETA>
__device__ __forceinline__ void strip4h_body(const HStripParams& p, char* smem_raw, int b, int s, int dirslot, int so)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8, NCW = NT / 64, DPR = W / 256;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  a2 = alpha * log2(e)  (exact row, log2 domain)
    float* Vbuf = Abuf + 2 * RL;                               // [2][RL]  V = 2^(a2 - X[group])  (NaN = escaped, 0 = dead)
    int* Xbuf = reinterpret_cast<int*>(Vbuf + 2 * RL);         // [2][GL]  group exponents; group gi covers li 4gi..4gi+3
    float* Mring = reinterpret_cast<float*>(Xbuf + 2 * GL);    // [RING][W] match rows
    float* Pbuf = Mring + H4_RING * W;                         // [NT][4]  half 1's partial sums
    int* Rbuf = reinterpret_cast<int*>(Pbuf + 4 * NT);         // [NT]     half 1's reference exponents

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
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * H4_TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * H4_TRP;
    // LDS geometry: alpha li = col - j0 + 32 (halo [0,32)); beta li = col - j0 (halo [W, W+32))
    const int halo_li0 = BETA ? W : 0;

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once), then -> registers ----
    // tile[r][d] = links[rlo + r][d] (pitch 33), -inf outside the graph / beyond TR.  The tile overlays the main-loop
    // buffers, which are not live yet.
    {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = 2 * NT + 192, RPP = NTHR / 32;    // rows per pass
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

    if (wave < 2 * NCW) {
        // =========================================================== compute waves: wave w and w + NCW serve the same lanes
        HCtx x;
        x.Abuf = Abuf; x.Vbuf = Vbuf; x.Xbuf = Xbuf; x.Mring = Mring; x.Pbuf = Pbuf; x.Rbuf = Rbuf;
        x.K = K; x.O = O; x.counters = p.counters;
        x.l = tid & (NT - 1); x.j0 = j0; x.j = j0 + 4 * x.l; x.Lb = Lb; x.Tb = Tb; x.T = T; x.L = L; x.TR = TR; x.nrows = nrows;
        x.b = b; x.dbg = p.dbg; x.col_ok = x.j < L;
        const float* tile = reinterpret_cast<const float*>(smem_raw);
        if (wave < NCW) strip4h_compute<NT, BETA, 0>(x, tile);
        else strip4h_compute<NT, BETA, 1>(x, tile);
    } else if (wave == 2 * NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * L;
            float* slot = Mring + (size_t)(itr % H4_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 256), 16, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < H4_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        h4_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + H4_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                // rows it+2 .. it+7 may stay in flight: 6*DPR DMAs younger than row it+1's
                if (DPR == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (DPR == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            h4_barrier();
            h4_barrier();
        }
    } else if (wave == 2 * NCW + 1) {
        // =========================================================== fetch wave: neighbour strip's halo -> LDS
        const bool hl = lane < H4_TRP;
        u64 g[H4_CH];
#pragma unroll
        for (int k = 0; k < H4_CH; ++k) g[k] = 0;
        // rolling prefetch: row it+CH is requested when row it has been consumed (see strip4g)
        auto load_row = [&](int itr) -> u64 {
            if (itr < nrows && hl) { const int t = BETA ? (Tb - 1 - itr) : itr; return h4_gran_load(hin + (size_t)t * H4_TRP + lane); }
            return 0;
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < H4_CH; ++k) g[k] = load_row(k);
        }
        __syncthreads();                         // link tile consumed
        h4_barrier();                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += H4_CH) {
#pragma unroll
            for (int k = 0; k < H4_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int t = BETA ? (Tb - 1 - it) : it;
                const int cur = it & 1;
                float hv = NEG_INF;
                if (has_producer && hl) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    u64 xg = g[k];
                    u32 spins = 0;
                    while (!__all((u32)(xg >> 32) == want)) {
                        if ((u32)(xg >> 32) != want) xg = h4_gran_load(hin + (size_t)t * H4_TRP + lane);
                        if (++spins > H4_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv = __uint_as_float((u32)xg);
                }
                {
                    // the halo's eight lane groups: exponent = ceil(max of 4) by two quad-permute steps
                    float gm = fmaxf(hv, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hv), 0xB1, 0xF, 0xF, false)));
                    gm = fmaxf(gm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gm), 0x4E, 0xF, 0xF, false)));
                    const bool dead = gm == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(gm) - 100.f;
                    const float e = hv - cf;
                    const float v = __builtin_amdgcn_exp2f(e);
                    if (hl) {
                        Abuf[cur * RL + halo_li0 + lane] = hv;
                        Vbuf[cur * RL + halo_li0 + lane] = v;
                        if ((lane & 3) == 0) Xbuf[cur * GL + (halo_li0 >> 2) + (lane >> 2)] = dead ? HNEGSENT : (int)cf;
                    }
                }
                if (has_producer) g[k] = load_row(it + H4_CH);
                h4_barrier();
                h4_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: boundary columns -> granules
        const bool pl = has_consumer && lane < H4_TRP;
        __syncthreads();                         // link tile consumed
        h4_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete; compute now writes the other buffer
                const int tp = BETA ? (Tb - it) : (it - 1);
                const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
                h4_gran_store(hout + (size_t)tp * H4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
            }
            h4_barrier();
            h4_barrier();
        }
        if (pl && nrows > 0) {
            const int it = nrows;
            const int tp = BETA ? (Tb - it) : (it - 1);
            const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
            h4_gran_store(hout + (size_t)tp * H4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(2 * NT + 192) void dag_strip4h_kernel(HStripParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = 4 * NT;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);          // 16-byte header; everything else starts at +16
    const int tid = threadIdx.x;
    if (tid == 0) *s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = *s_ticket;
    const int per = p.ndir * p.B;
    const int so = (int)(ticket / per);
    const int rem = (int)(ticket % per);
    const bool is_beta = p.alpha == nullptr || (p.ndir == 2 && rem >= p.B);
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
                for (int t = 0; t < T; ++t)
                    *reinterpret_cast<float4*>(O + (size_t)t * L + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            }
        }
        return;
    }
    if (is_beta) strip4h_body<NT, true>(p, smem_raw + 16, b, s, dirslot, so);
    else strip4h_body<NT, false>(p, smem_raw + 16, b, s, dirslot, so);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip4h_supported(const void* match, const void* alpha, const void* beta, int L, int TR)
{
    if (TR > 32 || (L & 3)) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta;
    return (a & 15) == 0;
}

template <int NT>
static int launch_one_h(const HStripParams& p, int nwg, hipStream_t st)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8;
    const size_t lds_main = (size_t)(4 * RL + 2 * GL + H4_RING * W + 5 * NT) * 4 + 16;
    const size_t lds_tile = (size_t)(W + 32) * 33 * 4;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    auto k = dag_strip4h_kernel<NT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(2 * NT + 192), lds, st, p);
    return check_launch("dag_loss_fwd(strip4h)");
}

int launch_dag_strip4h(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                       float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    // strip width: 1024 columns when that still yields >= ~200 workgroups, else 512
    const int ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
    const bool wide = (long)ndir * B * ns1024 >= 200;
    const int NS = wide ? ns1024 : ns512;
    HStripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta; p.trace = nullptr;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.dbg = 0;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * H4_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const int nwg = ndir * B * NS;
    return wide ? launch_one_h<256>(p, nwg, st) : launch_one_h<128>(p, nwg, st);
}

}  // namespace dsp
