// dag_dp_strip4g.hip — banded (TR <= 32) DAG DP, K2 alpha || K3 beta in EXP SPACE with ONE EXPONENT PER LANE GROUP.
//
// Same launch structure as dag_dp_strip4.hip (column strips, tagged-granule hand-off, tickets, loader / fetch / publish
// helper waves — read that header first).  What changes is the representation of the previous DP row in LDS:
//
//   strip4 : per VERTEX (mantissa P, exponent C)  -> per lane-row 18 ds_read_b128, a 35-op max tree, 36 sub + 36 ldexp
//   strip4g: per LANE GROUP of 4 vertices one integer exponent X, the 4 values stored as plain f32  V = 2^(a2 - X)
//            -> per lane-row 9 ds_read_b128 + 9 dwords of X, a 4-op max tree, 9 group shifts and 36 v_ldexp_f32;
//               the row head no longer waits for 36 exponents before the first FMA can issue.
//
// Exactness: X = ceil(max of the group's four a2) - 120, so V = 2^(a2 - X) lies in (2^119, 2^120] for the largest; a vertex more
// than 246 binades under its group's maximum flushes to 0 in storage — exactly what the window scaling would do to it, the
// window reference being at least its group's exponent (an earlier version stored such vertices as NaN "escapes": they
// poisoned every consumer's sums and bought nothing).  The window is shifted (v_ldexp per value) against the largest of its
// nine group exponents: scaled values are <= 2^120 and nothing above 2^-246 of the window maximum is flushed, so a sum
// S >= 2^-97 has lost at most 36 * 2^-126: exact to fp32.  S < 2^-97 or inf sends the cell to the register-only "medium" path
// (own maximum, log-domain row) and then the exact log-space path as in strip4.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));

struct GStripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int ldm, ldo;                             // row pitch (elements) of match and of alpha / beta: >= L, multiples of 4 (r06: graphs whose length is not)
    int dbg;
};

constexpr int G4_TRP = 32;
constexpr int G4_RING = 8;
constexpr int G4_CH = 4;                      // halo prefetch distance of the fetch wave (rows)
constexpr int GNEGSENT = -(1 << 30);       // "dead" exponent; far below any finite fp32 score
constexpr u32 G4_SPIN_LIMIT = 1u << 22;
constexpr float G4_LOG2E = 1.4426950408889634f;
constexpr float G4_BIAS = 120.f;              // stored / scaled values reach 2^120, a row sum of 36 stays under 2^126
constexpr float G4_LN2 = 0.6931471805599453f;

__device__ __forceinline__ u64 g4_gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void g4_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// DSP_DEBUG=prof: per-wave cycle accounting (s_memtime) of own work / barrier wait / LDS-read wait, for two workgroups
struct G4Prof { u64 last, work, wait, rd, fma, tmid; };
template <bool PROF>
__device__ __forceinline__ void g4_barrier(G4Prof& pf) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PROF) {
        const u64 t0 = __builtin_amdgcn_s_memtime();
        pf.work += t0 - pf.last;
        __builtin_amdgcn_s_barrier();
        const u64 t1 = __builtin_amdgcn_s_memtime();
        pf.wait += t1 - t0; pf.last = t1;
    } else {
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("" ::: "memory");
}

// window element index of (column c, distance d):  alpha: predecessor j+c-d -> q = 32 + c - d ; beta: successor -> q = c + d
template <bool BETA> __device__ __forceinline__ constexpr int gqidx(int c, int d) { return BETA ? (c + d) : (32 + c - d); }

// does weight pair i (window elements 2i, 2i+1) hold a transition of column c at all?  (6 of the 72 pairs do not: skipped statically)
template <bool BETA> __device__ __forceinline__ constexpr bool gpair_live(int c, int i) {
    const int lo = BETA ? c + 1 : c, hi = BETA ? c + 32 : c + 31;
    return 2 * i + 1 >= lo && 2 * i <= hi;
}

template <int NT, int MODE, bool BETA, bool PROF>
__device__ __forceinline__ void strip4g_body(const GStripParams& p, char* smem_raw, int b, int s, int dirslot, int so, int profslot)
{
    G4Prof pf; pf.last = PROF ? __builtin_amdgcn_s_memtime() : 0; pf.work = pf.wait = pf.rd = pf.fma = pf.tmid = 0;
    if (PROF && profslot >= 0 && threadIdx.x == 0) p.counters[50 + profslot * 3] = (u32)__builtin_amdgcn_s_memrealtime();
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8, NCW = NT / 64, DPR = W / 256;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  a2 = alpha * log2(e)  (exact row, log2 domain)
    float* Vbuf = Abuf + 2 * RL;                               // [2][RL]  V = 2^(a2 - X[group])  (NaN = escaped, 0 = dead)
    int* Xbuf = reinterpret_cast<int*>(Vbuf + 2 * RL);         // [2][GL]  group exponents; group gi covers li 4gi..4gi+3
    float* Mring = reinterpret_cast<float*>(Xbuf + 2 * GL);    // [RING][W] match rows
    static_assert(MODE == 0, "strip4g implements the log-sum DP only");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, L = p.L, TR = p.TR;
    const int j0 = s * W;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * p.ldm;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = (BETA ? p.beta : p.alpha) + (size_t)b * T * p.ldo;
    const int LDM = p.ldm, LDO = p.ldo;
    const int nrows = Tb;

    const bool has_producer = so > 0 && (BETA ? (j0 + W < Lb) : true);
    const bool has_consumer = BETA ? (s > 0) : (s < p.NS - 1 && j0 + W < Lb);
    const int prod_strip = BETA ? s + 1 : s - 1;
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * G4_TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * G4_TRP;
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
        __builtin_amdgcn_s_setprio(2);           // the compute wave of a SIMD goes before the helper wave that shares it
        const int l = tid;                       // lane's group
        const int j = j0 + 4 * l;
        const bool col_ok = j < L;
        // structural reachability (cells outside are -inf in the reference too: their LSE runs over -inf terms only):
        // alpha: t <= col <= min(L_b-1, t*TR);  beta: col >= t, T_b-1-t <= L_b-1-col <= (T_b-1-t)*TR
        auto cell_active = [&](int col, int t) -> bool {       // (T * TR fits an int: T, L < 2^20 and TR <= 32)
            if (!BETA) return col >= t && col < Lb && col <= t * TR;
            const int rem = Tb - 1 - t, gap = Lb - 1 - col;       // rows left / vertices left: 1..TR vertices per row
            return col >= t && gap >= rem && gap <= rem * TR;
        };
        float E[4][32];
        float lmax[4];
        float sthr[4];
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
                raw[d - 1] = (MODE == 0) ? v * G4_LOG2E : v;
                mx = fmaxf(mx, raw[d - 1]);
            }
            if (MODE == 0) {
                if (mx == NEG_INF) mx = 0.f;
                lmax[c] = mx;
                bool flushed = false;                      // a finite link more than ~120 binades under the column's strongest
#pragma unroll
                for (int d = 0; d < 32; ++d) {
                    E[c][d] = __builtin_amdgcn_exp2f(raw[d] - mx);
                    flushed |= (raw[d] != NEG_INF) & (raw[d] - mx < -120.f);
                }
                // Such a weight is 0 (or inexact) in fp32, and a scaled window value can be as large as 2^120, so the term it
                // drops can reach 2^0: a column that has one only trusts sums that dwarf that; everything else is redone by
                // the medium path, whose values are <= 1 (dropped terms < 2^-120 against a sum >= 2^-97).
                sthr[c] = flushed ? 0x1p30f : 0x1p-97f;
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
        g4_barrier<PROF>(pf);                            // prologue barrier: match row 0 is in the ring

        typedef int v2i __attribute__((ext_vector_type(2)));
        typedef float v4f __attribute__((ext_vector_type(4)));
        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            if (PROF && profslot >= 0 && tid == 0 && it == 64) p.counters[51 + profslot * 3] = (u32)__builtin_amdgcn_s_memrealtime();
            float a2[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
            float vn[4] = {0.f, 0.f, 0.f, 0.f};
            int xn = GNEGSENT;
            if (it == 0) {
                const float4 mt = *reinterpret_cast<const float4*>(Mring + (size_t)(it % G4_RING) * W + 4 * l);
                const float m2[4] = {mt.x, mt.y, mt.z, mt.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool seed = BETA ? (j + c == Lb - 1) : (j + c == 0);
                    if (seed) a2[c] = m2[c] * G4_LOG2E;
                }
            } else {
                // ---- row head: all 15 LDS reads (match, nine group exponents, the 36-value window: li 4l .. 4l+35, groups
                // l .. l+8) leave as ONE issue group; consumers wait with counted lgkmcnt (LDS returns in order, and this
                // stretch issues no other LDS / scalar-memory operation).  Left to the scheduler the window reads are sunk
                // between the FMA groups one or two at a time to save registers, which with one compute wave per SIMD
                // exposes six LDS round trips per row.
                v4f mt; v2i x01, x23, x45, x67; int x8; v4f pv[9];
                {
                    const u32 maddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Mring + (size_t)(it % G4_RING) * W + 4 * l);
                    const u32 xaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Xbuf + prv * GL + l);
                    const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Vbuf + prv * RL + 4 * l);
#define G4_ROW_HEAD_READS \
                        "ds_read_b128 %0, %15\n\t" \
                        "ds_read2_b32 %1, %16 offset1:1\n\t" \
                        "ds_read2_b32 %2, %16 offset0:2 offset1:3\n\t" \
                        "ds_read2_b32 %3, %16 offset0:4 offset1:5\n\t" \
                        "ds_read2_b32 %4, %16 offset0:6 offset1:7\n\t" \
                        "ds_read_b32 %5, %16 offset:32\n\t" \
                        "ds_read_b128 %6, %17\n\t" \
                        "ds_read_b128 %7, %17 offset:16\n\t" \
                        "ds_read_b128 %8, %17 offset:32\n\t" \
                        "ds_read_b128 %9, %17 offset:48\n\t" \
                        "ds_read_b128 %10, %17 offset:64\n\t" \
                        "ds_read_b128 %11, %17 offset:80\n\t" \
                        "ds_read_b128 %12, %17 offset:96\n\t" \
                        "ds_read_b128 %13, %17 offset:112\n\t" \
                        "ds_read_b128 %14, %17 offset:128"
#define G4_ROW_HEAD_OPERANDS \
                        : "=&v"(mt), "=&v"(x01), "=&v"(x23), "=&v"(x45), "=&v"(x67), "=&v"(x8), \
                          "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), \
                          "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8]) \
                        : "v"(maddr), "v"(xaddr), "v"(vaddr) \
                        : "memory"
                    if (PROF) asm volatile(G4_ROW_HEAD_READS "\n\ts_waitcnt lgkmcnt(0)" G4_ROW_HEAD_OPERANDS);   // stamps are scalar-memory ops
                    else asm volatile(G4_ROW_HEAD_READS G4_ROW_HEAD_OPERANDS);
                }
                if (PROF) {
                    const u64 tm = __builtin_amdgcn_s_memtime();
                    pf.rd += tm - pf.last; pf.tmid = tm;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (1) match row landed: everything that depends on it alone is computed under the remaining reads
                asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(mt));
                const float m2[4] = {mt.x, mt.y, mt.z, mt.w};
                float base[4]; bool okc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    okc[c] = cell_active(j + c, t);
                    base[c] = lmax[c] + m2[c] * G4_LOG2E;                    // log2(strongest link * emission)
                }
                // (2) group exponents landed.  Reference = largest of the nine: groups are stored with a +120 bias (see the
                // row write), so scaled values reach 2^120 at most (sums < 2^126) and a column whose predecessors all sit up
                // to ~190 binades under the window maximum still sums to >= 2^-97.  Next to the DP's diagonal neighbouring
                // vertices are 25-35 binades apart, so this headroom is used on every row.
                asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(x01), "+v"(x23), "+v"(x45), "+v"(x67), "+v"(x8));
                const int xw[9] = {x01.x, x01.y, x23.x, x23.y, x45.x, x45.y, x67.x, x67.y, x8};
                int refi = max(max(max(xw[0], xw[1]), max(xw[2], xw[3])), max(max(xw[4], xw[5]), max(max(xw[6], xw[7]), xw[8])));
                const bool any_live = refi != GNEGSENT;
                if (!any_live) refi = 0;
                // group shifts X - ref <= 0, applied to every value with v_ldexp_f32: a group FACTOR 2^(X - ref) would itself
                // flush below 2^-126 and cut the window at 126 binades although the stored values (bias +120) reach 246
                int kg[9];
#pragma unroll
                for (int g = 0; g < 9; ++g) kg[g] = xw[g] - refi;                 // <= 0; hugely negative for dead groups
                // (3) the window, group by group as it lands
                v2f S2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { S2[c].x = 0.f; S2[c].y = 0.f; }
#define G4_GROUP(k, n) \
                { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(pv[k])); \
                  v2f wa, wb; wa.x = ldexpf(pv[k].x, kg[k]); wa.y = ldexpf(pv[k].y, kg[k]); wb.x = ldexpf(pv[k].z, kg[k]); wb.y = ldexpf(pv[k].w, kg[k]); \
                  _Pragma("unroll") for (int c = 0; c < 4; ++c) { \
                      if (gpair_live<BETA>(c, 2 * k)) S2[c] = __builtin_elementwise_fma(wa, E2[c][2 * k], S2[c]); \
                      if (gpair_live<BETA>(c, 2 * k + 1)) S2[c] = __builtin_elementwise_fma(wb, E2[c][2 * k + 1], S2[c]); } }
                G4_GROUP(0, 8) G4_GROUP(1, 7) G4_GROUP(2, 6) G4_GROUP(3, 5) G4_GROUP(4, 4)
                G4_GROUP(5, 3) G4_GROUP(6, 2) G4_GROUP(7, 1) G4_GROUP(8, 0)
#undef G4_GROUP
                float S[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) S[c] = S2[c].x + S2[c].y;
                if (PROF) {
                    // the stamp must follow the sums: tie it to them through an empty asm the scheduler cannot cross
                    asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]));
                    __builtin_amdgcn_sched_barrier(0);
                    const u64 tf = __builtin_amdgcn_s_memtime();
                    pf.fma += tf - pf.tmid;
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float ref = (float)refi;
                // ---- row tail
                bool need_fb = false;
                const bool R_live = any_live;
                bool flag[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool okl = okc[c] & any_live;
                    flag[c] = okl & !(S[c] >= sthr[c] && S[c] <= 0x1p126f);       // too small (or inf / NaN)
                    need_fb |= flag[c];
                    a2[c] = (okl & !flag[c]) ? (__builtin_amdgcn_logf(S[c]) + (ref + base[c])) : NEG_INF;
                }
                if (__builtin_expect(need_fb && p.dbg != 4, 0)) {      // (DSP_DEBUG=nofallback: timing experiment, WRONG results)
                    // diagnostics (DSP_DEBUG=medium only: a returning global atomic costs the wave a memory round trip per entry)
                    if (p.dbg == 1) {
                        const u32 slot = atomicAdd(&p.counters[3], 1u);
                        if (slot < 14) { const int fc = flag[0] ? 0 : flag[1] ? 1 : flag[2] ? 2 : 3;
                          p.counters[8 + 4 * slot] = (u32)b | (BETA ? 0x100u : 0u); p.counters[9 + 4 * slot] = (u32)t; p.counters[10 + 4 * slot] = (u32)(j + fc);
                          p.counters[11 + 4 * slot] = __float_as_uint(fc == 0 ? S[0] : fc == 1 ? S[1] : fc == 2 ? S[2] : S[3]); }
                    }
                    // (0) the DP's diagonal cell (vertex = row, counted from the direction's start) has ONE live transition; with
                    //     peaked scores it sits hundreds of binades under its neighbours and lands here on every row:
                    //     a2 = a2_prev(predecessor) + log2(weight) + base, no sum.  (A flushed weight leaves the cell flagged.)
                    bool still = false;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int dl = BETA ? (Lb - Tb + 1 + t - (j + c)) : (j + c - t + 1);
                        if (flag[c] && dl == 1 && sthr[c] == 0x1p-97f) {
                            const float ap = Abuf[prv * RL + 4 * l + gqidx<BETA>(c, 1)];
                            const float e1 = Eval(c, 1);
                            a2[c] = (e1 > 0.f && ap != NEG_INF) ? (ap + __builtin_amdgcn_logf(e1) + base[c]) : NEG_INF;
                            flag[c] = false;
                        }
                        still |= flag[c];
                    }
                  if (still) {
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
                            // transitions that can be alive at all: next to the DP's diagonal (where this path fires on
                            // peaked scores: the diagonal falls hundreds of binades under its right-hand neighbours) a vertex
                            // has a handful of live predecessors, so the 32 terms are walked in chunks of 8 that the wave skips
                            // live transitions d_lo .. d_hi: bounded by the diagonal on one side and the reach frontier on the other
                            const int dlim = min(32, max(0, BETA ? (Lb - Tb + 1 + t - (j + c)) : (j + c - t + 1)));
                            const int dlo = max(1, BETA ? (Lb - 1 - (Tb - 2 - t) * TR - (j + c)) : (j + c - (t - 1) * TR));
                            float cmx = NEG_INF;
#pragma unroll
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                if (__any(dlim >= d0 && dlo <= d0 + 7)) {
#pragma unroll
                                    for (int d = d0; d < d0 + 8; ++d) cmx = fmaxf(cmx, d <= TR ? aw[gqidx<BETA>(c, d)] : NEG_INF);   // (r05: only inside the window —
                                    // with TR < 32 the padded slots hold LIVE cells with zero weight; the frontier cell's one real predecessor sits
                                    // hundreds of binades under them and flushed against their maximum: every frontier cell took the exact path,
                                    // 4.5 ms instead of 0.46 ms at C2 / TR = 16)
                                }
                            }
                            float sc = 0.f;
#pragma unroll
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                if (__any(dlim >= d0 && dlo <= d0 + 7)) {
#pragma unroll
                                    for (int d = d0; d < d0 + 8; ++d)
                                        sc = fmaf(__builtin_amdgcn_exp2f(d <= TR ? aw[gqidx<BETA>(c, d)] - cmx : NEG_INF), Eval(c, d), sc);   // (slots past the window: 2^(-inf) = 0, not inf x 0)
                                }
                            }
                            S[c] = sc;
                            if (sc >= 0x1p-97f) a2[c] = __builtin_amdgcn_logf(sc) + cmx + base[c];
                        } else {
                            S[c] = 1.f;                      // settled by the fast path (or inactive)
                        }
                    }
                    // (b) EXACT path for what is left
#pragma unroll 1
                    for (int c = 0; c < 4; ++c) {
                        { const float Sc = (c == 0) ? S[0] : (c == 1) ? S[1] : (c == 2) ? S[2] : S[3];
                          if (!(cell_active(j + c, t) && R_live && !(Sc >= 0x1p-97f))) continue; }
                        // (1) cheap structural test: is any predecessor alive (a2 row in LDS)?  if not the cell is -inf
                        float amax = NEG_INF;
                        for (int d = 1; d <= 32; ++d) amax = fmaxf(amax, Abuf[prv * RL + 4 * l + (BETA ? (c + d) : (32 + c - d))]);
                        float r = NEG_INF;
                        if (amax != NEG_INF) {
                            // (2) exact log-space value; raw links re-read from HBM 8 at a time (independent loads)
                            { const u32 slot = atomicAdd(&p.counters[2], 1u); if (!p.dbg && slot < 14) { p.counters[8 + 4 * slot] = (u32)b; p.counters[9 + 4 * slot] = (u32)t; p.counters[10 + 4 * slot] = (u32)(j + c); p.counters[11 + 4 * slot] = (u32)refi; } }
                            float mx = NEG_INF, sum = 0.f;
                            for (int d0 = 1; d0 <= 32; d0 += 8) {
                                float lk[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const int row = BETA ? (j + c) : (j + c - d);
                                    const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                    const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                    lk[u] = ok ? raw * G4_LOG2E : NEG_INF;
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
                                r = __builtin_amdgcn_logf(sum) + mx + mm * G4_LOG2E;
                            }
                        }
                        if (c == 0) a2[0] = r; else if (c == 1) a2[1] = r; else if (c == 2) a2[2] = r; else a2[3] = r;
                    }
                  }
                }

            }
            // ---- write the row: LDS state for the next row, HBM output ----
            {
                // group exponent X = ceil(largest of the four) - 120, so V = 2^(a2 - X) spans [2^-126, 2^120]
                const float amax = fmaxf(fmaxf(a2[0], a2[1]), fmaxf(a2[2], a2[3]));
                const bool dead = amax == NEG_INF;
                const float cf = dead ? 0.f : ceilf(amax) - G4_BIAS;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float e = a2[c] - cf;
                    const float v = __builtin_amdgcn_exp2f(e);
                    vn[c] = v;                 // flushes to 0 more than 246 binades under the group maximum — as the scaling would
                }
                xn = dead ? GNEGSENT : (int)cf;
            }
            *reinterpret_cast<float4*>(Vbuf + cur * RL + own_li0 + 4 * l) = make_float4(vn[0], vn[1], vn[2], vn[3]);
            Xbuf[cur * GL + (own_li0 >> 2) + l] = xn;
            *reinterpret_cast<float4*>(Abuf + cur * RL + own_li0 + 4 * l) = make_float4(a2[0], a2[1], a2[2], a2[3]);
            if (col_ok)
                *reinterpret_cast<float4*>(O + (size_t)t * LDO + j) = make_float4(a2[0] * G4_LN2, a2[1] * G4_LN2, a2[2] * G4_LN2, a2[3] * G4_LN2);
            g4_barrier<PROF>(pf);
        }
        // rows the recurrence never reaches
        if (col_ok) for (int t = Tb; t < T; ++t) {
            *reinterpret_cast<float4*>(O + (size_t)t * LDO + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * LDM;
            float* slot = Mring + (size_t)(itr % G4_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 256), 16, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < G4_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g4_barrier<PROF>(pf);                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + G4_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                // rows it+2 .. it+7 may stay in flight: 6*DPR DMAs younger than row it+1's
                if (DPR == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (DPR == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            g4_barrier<PROF>(pf);
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: neighbour strip's halo -> LDS
        const bool hl = lane < G4_TRP;
        u64 g[G4_CH];
#pragma unroll
        for (int k = 0; k < G4_CH; ++k) g[k] = 0;
        // rolling prefetch: row it+CH is requested when row it has been consumed, so every request has CH row times to land
        // (the strip-0 speed is only reached if no row head waits on a memory round trip), and a consumer settles about
        // CH + 3 rows behind its producer.
        auto load_row = [&](int itr) -> u64 {
            if (itr < nrows && hl) { const int t = BETA ? (Tb - 1 - itr) : itr; return g4_gran_load(hin + (size_t)t * G4_TRP + lane); }
            return 0;
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < G4_CH; ++k) g[k] = load_row(k);
        }
        __syncthreads();                         // link tile consumed
        g4_barrier<PROF>(pf);                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += G4_CH) {
#pragma unroll
            for (int k = 0; k < G4_CH; ++k) {
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
                        if ((u32)(x >> 32) != want) x = g4_gran_load(hin + (size_t)t * G4_TRP + lane);
                        if (++spins > G4_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv = __uint_as_float((u32)x);
                }
                {
                    // the halo's eight lane groups: exponent = ceil(max of 4) by two quad-permute steps
                    float gm = fmaxf(hv, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hv), 0xB1, 0xF, 0xF, false)));
                    gm = fmaxf(gm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gm), 0x4E, 0xF, 0xF, false)));
                    const bool dead = gm == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(gm) - G4_BIAS;
                    const float e = hv - cf;
                    const float v = __builtin_amdgcn_exp2f(e);
                    if (hl) {
                        Abuf[cur * RL + halo_li0 + lane] = hv;
                        Vbuf[cur * RL + halo_li0 + lane] = v;
                        if ((lane & 3) == 0) Xbuf[cur * GL + (halo_li0 >> 2) + (lane >> 2)] = dead ? GNEGSENT : (int)cf;
                    }
                }
                if (has_producer) g[k] = load_row(it + G4_CH);
                g4_barrier<PROF>(pf);
            }
        }
    } else {
        // =========================================================== publish wave: boundary columns -> granules
        const bool pl = has_consumer && lane < G4_TRP;
        __syncthreads();                         // link tile consumed
        g4_barrier<PROF>(pf);                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete (barrier it-1 passed); compute now writes the other buffer
                const int tp = BETA ? (Tb - it) : (it - 1);
                const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
                g4_gran_store(hout + (size_t)tp * G4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
            }
            g4_barrier<PROF>(pf);
        }
        if (pl && nrows > 0) {
            const int it = nrows;
            const int tp = BETA ? (Tb - it) : (it - 1);
            const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : 32) + (BETA ? 0 : (W - 32)) + lane];
            g4_gran_store(hout + (size_t)tp * G4_TRP + lane, p.tag_base + 1u + (u32)tp, v);
        }
        (void)pub_li0;
    }
    if (PROF && profslot >= 0 && threadIdx.x == 0) p.counters[52 + profslot * 3] = (u32)__builtin_amdgcn_s_memrealtime();
    if (PROF && profslot >= 0 && lane == 0) {
        u32* o = p.counters + 8 + profslot * 21 + wave * 3;
        o[0] = (u32)pf.work; o[1] = (u32)pf.wait; o[2] = (u32)pf.rd;
        if (wave == 0) p.counters[56 + profslot] = (u32)pf.fma;
    }
}

template <int NT, int MODE, bool PROF>
__global__ __launch_bounds__(NT + 192) void dag_strip4g_kernel(GStripParams p)
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
                float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * p.ldo;
                for (int t = 0; t < T; ++t) {
                    *reinterpret_cast<float4*>(O + (size_t)t * p.ldo + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
                }
            }
        }
        return;
    }
    const int profslot = !PROF ? -1 : (ticket == 0 ? 0 : (ticket == 2u * (u32)per ? 1 : -1));
    if (MODE == 0 && is_beta) strip4g_body<NT, MODE, true, PROF>(p, smem_raw + 16, b, s, dirslot, so, profslot);
    else strip4g_body<NT, MODE, false, PROF>(p, smem_raw + 16, b, s, dirslot, so, profslot);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip4g_supported(const void* match, const void* alpha, const void* beta, int L, int TR, int ldm, int ldo)
{
    // rows start on 16-byte boundaries: the PITCHES are multiples of 4, the graph length need not be (a lane whose four columns straddle L
    // loads / stores inside the pitch padding; columns >= L are outside every sample's graph and come out -inf)
    if (TR > 32 || (ldm & 3) || (ldo & 3) || ldm < ((L + 3) & ~3) || ldo < ((L + 3) & ~3)) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta;
    return (a & 15) == 0;
}

template <int NT, bool PROF>
static int launch_one_g(const GStripParams& p, int nwg, hipStream_t st)
{
    constexpr int W = 4 * NT, RL = W + 32, GL = NT + 8;
    const size_t lds_main = (size_t)(4 * RL + 2 * GL + G4_RING * W) * 4 + 16;
    const size_t lds_tile = (size_t)(W + 32) * 33 * 4;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    auto k = dag_strip4g_kernel<NT, 0, PROF>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(NT + 192), lds, st, p);
    return check_launch("dag_loss_fwd(strip4g)");
}

int launch_dag_strip4g(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                       float* alpha, float* beta, int B, int T, int L, int TR, int ldm, int ldo, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    // strip width: 1024 columns when that still yields >= ~200 workgroups, else 512
    const int ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
    const bool wide = (long)ndir * B * ns1024 >= 200;
    const int NS = wide ? ns1024 : ns512;
    GStripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta; p.trace = nullptr;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.ldm = ldm; p.ldo = ldo;
    { static const char* const e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "medium")) ? 1 : (e && !strcmp(e, "prof")) ? 2 : (e && !strcmp(e, "nofallback")) ? 4 : 0; }
    const size_t halo_bytes = (size_t)ndir * B * NS * T * G4_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const int nwg = ndir * B * NS;
    if (p.dbg == 2 && wide) return launch_one_g<256, true>(p, nwg, st);
    return wide ? launch_one_g<256, false>(p, nwg, st) : launch_one_g<128, false>(p, nwg, st);
}

}  // namespace dsp
