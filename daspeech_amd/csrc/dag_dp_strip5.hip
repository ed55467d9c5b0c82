// dag_dp_strip5.hip — banded (TR <= 32) DAG DP, K2 alpha || K3 beta in EXP SPACE with ONE EXPONENT PER 64-COLUMN SUPERBLOCK.
//
// Replaces dag_dp_strip4g.hip on the C2-class shapes.  Same launch structure (column strips of W vertices, one workgroup per
// (sample, direction, strip); tagged-granule hand-off between strips; loader / fetch / publish helper waves; tickets) — what
// changes is how the previous DP row is kept in LDS, and with it the instruction count of a row:
//
//   strip4g : V = 2^(a2 - X[group of 4])         -> a lane's 36-value window carries NINE exponents: 9 exponent reads, a max tree,
//             9 shifts and 36 v_ldexp per lane-row before the 128 FMAs — 235 VALU instructions per lane-row, one compute wave per SIMD.
//   strip5  : the strip is cut into blocks of 32 columns; SUPERBLOCK k = blocks k, k+1 (64 columns, overlapping by one block) has
//             ONE exponent X_k and its own 64-value LDS slot holding 2^(a2 - X_k).  Every 36-value window lies inside exactly one
//             superblock (window base / 32), so a lane reads 9 x ds_read_b128 from ONE slot and feeds the FMAs directly: no per-value
//             shift, no exponent tree.  The price is on the write side: a column belongs to two superblocks, so each new value is
//             exponentiated and stored twice.
//   X_k for row t is an UPPER BOUND taken from the block maxima H of row t-1 (a cell is at most its strongest predecessor + the
//   incoming-link mass): X_k[t] = ceil(max(H[t-1][k-1..k+1])) - 96 (alpha; beta mirrored).  No cross-lane reduction sits on the
//   row's critical path: H of row t is reduced (DPP, 8 or 16 lanes) after the row's stores, the bound is evaluated in the next
//   row head under the window reads.  For the halo block the fetch wave supplies the ACTUAL maximum of the next halo row.
//   Values span [2^-126, 2^~120] inside a superblock: 220+ binades under the bound before anything is flushed.
//
// With ~75 (CPL = 2) / ~135 (CPL = 4) VALU instructions per lane-row the 2-columns-per-lane geometry becomes affordable: 8 compute
// waves per 1024-column strip = TWO compute waves per SIMD, which cover each other's LDS round trips and log/exp tails.
//
// Exactness (as strip4g): a sum S >= 2^-97 has lost at most 36 * 2^-126: exact to fp32.  S < 2^-97 (or inf / NaN) on a structurally
// reachable cell is redone: diagonal shortcut -> "medium" path (own maximum, from the exact log2-domain row kept beside the scaled
// one) -> exact log-space path from the raw links.  Columns with a transition weight that fp32 flushed only trust sums >= 2^30.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

struct S5Params {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word, [2] exact cells, [3] medium lane-rows
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int dbg;                                  // 4 = DSP_DEBUG=nofallback (timing experiment, wrong results), 1 = count medium entries
};

constexpr int S5_TRP = 32;
constexpr int S5_RING = 8;
constexpr int S5_CH = 4;                      // halo prefetch distance of the fetch wave (rows)
constexpr int S5_SBS = 96;                    // dwords per superblock slot (64 used): slot stride 96 keeps a wave's window reads conflict-free
constexpr u32 S5_SPIN_LIMIT = 1u << 22;
constexpr float S5_LOG2E = 1.4426950408889634f;
constexpr float S5_LN2 = 0.6931471805599453f;
constexpr float S5_BIAS = 96.f;               // stored values reach 2^96 at the bound; 24 binades of slack for incoming-link mass / positive emissions

__device__ __forceinline__ u64 s5_gran_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void s5_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void s5_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// maximum over aligned groups of 8 (STEPS = 3) or 16 (STEPS = 4) lanes, result in every lane of the group: v_max_f32 with a DPP source
// (one instruction per step; the builtin form costs a v_mov of the "old" value, the DPP move and two canonicalising maxima per step).
// A DPP read needs two wait states behind the VALU write of its source.
template <int STEPS> __device__ __forceinline__ float s5_group_max(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    if (STEPS == 4) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    return v;
}
__device__ __forceinline__ float s5_expo(float bound) {          // superblock exponent from the bound on its values
    return bound == NEG_INF ? 0.f : ceilf(bound) - S5_BIAS;
}

// DSP_DEBUG=prof: per-wave cycle accounting (s_memtime) of one workgroup: compute waves split a row into window-read wait / FMA
// stretch / tail+stores / barrier wait, helper waves into own work / barrier wait.  Timing experiment only (adds ~10 % overhead).
struct S5Prof { u64 last, a, b, c, d; };
template <bool PROF> __device__ __forceinline__ void s5_stamp(S5Prof& pf, u64& acc) {
    if (PROF) {
        __builtin_amdgcn_sched_barrier(0);
        const u64 t = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += t - pf.last; pf.last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int W, int CPL, bool BETA, bool PROF>
__device__ __forceinline__ void strip5_body(const S5Params& p, char* smem_raw, int b, int s, int dirslot, int so, bool profwg)
{
    S5Prof pf; pf.last = PROF ? __builtin_amdgcn_s_memtime() : 0; pf.a = pf.b = pf.c = pf.d = 0;
    constexpr int NT = W / CPL, NCW = NT / 64, RL = W + 32, NB = W / 32 + 1, NSB = W / 32, DPR = W / 256;
    constexpr int HR = (NB + 2 + 3) & ~3, XS = (NSB + 3) & ~3;
    constexpr int BLK_LANES = 32 / CPL;       // lanes per 32-column block
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]        exact row, a2 = value * log2(e)
    float* SBarr = Abuf + 2 * RL;                              // [2][NSB][96]   2^(a2 - X_k) for the 64 columns of superblock k
    float* Hrow = SBarr + 2 * NSB * S5_SBS;                    // [2][HR]        entry e = block e-1's row maximum; the free end slot = next halo row's
    float* Xsb = Hrow + 2 * HR;                                // [2][XS]        superblock exponents of the stored row
    float* Mring = Xsb + 2 * XS;                               // [RING][W]      match rows

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
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * S5_TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * S5_TRP;
    // LDS geometry: alpha li = col - j0 + 32 (halo = block 0); beta li = col - j0 (halo = block NB-1)
    const int halo_li0 = BETA ? W : 0;
    const int own_li0 = BETA ? 0 : 32;
    // the DP's seed (alpha: (0, 0); beta: (T_b-1, L_b-1)) fixes every exponent of row 0
    const int seed_col = BETA ? Lb - 1 : 0;
    const bool seed_here = seed_col >= j0 && seed_col < j0 + W;

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once), then -> registers.  tile[r][d] = links[rlo + r][d]
    //      (pitch 33), -inf outside the graph / beyond TR.  The tile overlays the main-loop buffers, which are not live yet.
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
        const int l = tid;
        const int li0 = own_li0 + CPL * l;       // LDS index of the lane's first column
        const int j = j0 + CPL * l;              // ... and the column itself
        const bool col_ok = j < L;
        const int wb = BETA ? (li0 & ~3) : ((li0 - 32) & ~3);       // window base: 36 values li wb .. wb+35, 16-byte aligned
        const int delta = li0 - wb;              // alpha: 32 (+2 on odd lanes when CPL = 2); beta: 0 (+2)
        const bool odd = (CPL == 2) && (l & 1);
        const int ksr = wb >> 5, pr = wb & 31;   // superblock the window lies in, offset inside its slot
        const int kb = li0 >> 5, pos = li0 & 31; // the lane's own block / offset inside it
        // structural reachability (cells outside are -inf in the reference too: their LSE runs over -inf terms only):
        // alpha: t <= col <= min(L_b-1, t*TR);  beta: col >= t, T_b-1-t <= L_b-1-col <= (T_b-1-t)*TR
        auto cell_active = [&](int col, int t) -> bool {
            if (!BETA) return col >= t && col < Lb && col <= t * TR;
            const int rem = Tb - 1 - t, gap = Lb - 1 - col;
            return col >= t && gap >= rem && gap <= rem * TR;
        };
        float lmax[CPL], sthr[CPL];
        v2f E2[CPL][18];                         // (weight of window element 2i, of 2i+1) for column c; 0 where not a predecessor
        {
            const float* tile = reinterpret_cast<const float*>(smem_raw);
            auto tileval = [&](int c, int d) -> float {            // log-weight of (column c, distance d), d may be out of range
                if (d < 1 || d > 32) return NEG_INF;
                if (!BETA) return tile[(li0 + c - d) * 33 + (d - 1)];
                return (j + c + d >= Lb) ? NEG_INF : tile[(li0 + c) * 33 + (d - 1)];
            };
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                float mx = NEG_INF;
#pragma unroll 8
                for (int d = 1; d <= 32; ++d) mx = fmaxf(mx, tileval(c, d));
                mx = (mx == NEG_INF) ? 0.f : mx * S5_LOG2E;
                lmax[c] = mx;
                bool flushed = false;            // a finite link more than ~120 binades under the column's strongest
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int q0 = 2 * i, q1 = 2 * i + 1;
                    const int d0 = BETA ? (q0 - delta - c) : (delta + c - q0), d1 = BETA ? (q1 - delta - c) : (delta + c - q1);
                    const float r0 = tileval(c, d0) * S5_LOG2E, r1 = tileval(c, d1) * S5_LOG2E;
                    E2[c][i].x = __builtin_amdgcn_exp2f(r0 - mx);
                    E2[c][i].y = __builtin_amdgcn_exp2f(r1 - mx);
                    flushed |= ((r0 != NEG_INF) & (r0 - mx < -120.f)) | ((r1 != NEG_INF) & (r1 - mx < -120.f));
                }
                // Such a weight is 0 (or inexact) in fp32 and a scaled value can be as large as 2^120, so the term it drops can reach
                // 2^0: a column that has one only trusts sums that dwarf that; the rest is redone by the medium / exact paths.
                sthr[c] = flushed ? 0x1p30f : 0x1p-97f;
            }
        }
        // weight of (column c, distance d) recovered from the pair layout: static register indices, lane parity by select
        auto Eval = [&](int c, int d) -> float {
            const int qe = BETA ? (c + d) : (32 + c - d);             // window position on an even lane (delta = 32 / 0)
            if (CPL == 2) {
                const v2f e = odd ? E2[c][(qe >> 1) + 1] : E2[c][qe >> 1];
                return (qe & 1) ? e.y : e.x;
            }
            return (qe & 1) ? E2[c][qe >> 1].y : E2[c][qe >> 1].x;
        };
        __syncthreads();                         // tile consumed: the loader may start filling the ring over it
        if (tid < 2 * HR) Hrow[tid] = NEG_INF;   // slots no one writes (the far end) must read as "dead"
        s5_barrier();                            // prologue barrier: match row 0 is in the ring
        const float seedX = seed_here ? (ceilf(Mring[seed_col - j0] * S5_LOG2E) - S5_BIAS) : 0.f;

        const u32 a_v = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(SBarr + ksr * S5_SBS + pr);
        const u32 a_x = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Xsb + ksr);
        const u32 a_h = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Hrow + kb - 1 + (BETA ? 1 : 0));
        const u32 a_m = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Mring + CPL * l);

        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            s5_stamp<PROF>(pf, pf.d);             // barrier wait (+ loop overhead)
            float a2[CPL];
            float XA = seedX, XB = seedX;        // exponents of the two superblocks this lane's block belongs to (as first / second block)
#pragma unroll
            for (int c = 0; c < CPL; ++c) a2[c] = NEG_INF;
            if (it == 0) {
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (j + c == seed_col) a2[c] = Mring[CPL * l + c] * S5_LOG2E;
            } else {
                // ---- row head: all LDS reads of the row (match, read exponent, 4 block maxima, the 36-value window) leave as ONE issue
                // group; consumers wait with counted lgkmcnt (LDS returns in order; this stretch issues no other LDS / scalar-memory op)
                v4f mt; float xr; v2f h01, h23; v4f pv[9];
                {
                    const u32 maddr = a_m + (u32)((it % S5_RING) * W * 4);
                    const u32 xaddr = a_x + (u32)(prv * XS * 4);
                    const u32 haddr = a_h + (u32)(prv * HR * 4);
                    const u32 vaddr = a_v + (u32)(prv * NSB * S5_SBS * 4);
#define S5_WINDOW_READS \
                        "ds_read_b32 %1, %14\n\t" \
                        "ds_read2_b32 %2, %15 offset1:1\n\t" \
                        "ds_read2_b32 %3, %15 offset0:2 offset1:3\n\t" \
                        "ds_read_b128 %4, %16\n\t" \
                        "ds_read_b128 %5, %16 offset:16\n\t" \
                        "ds_read_b128 %6, %16 offset:32\n\t" \
                        "ds_read_b128 %7, %16 offset:48\n\t" \
                        "ds_read_b128 %8, %16 offset:64\n\t" \
                        "ds_read_b128 %9, %16 offset:80\n\t" \
                        "ds_read_b128 %10, %16 offset:96\n\t" \
                        "ds_read_b128 %11, %16 offset:112\n\t" \
                        "ds_read_b128 %12, %16 offset:128"
#define S5_HEAD_OPERANDS \
                        : "=&v"(mt), "=&v"(xr), "=&v"(h01), "=&v"(h23), \
                          "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), \
                          "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8]) \
                        : "v"(maddr), "v"(xaddr), "v"(haddr), "v"(vaddr) \
                        : "memory"
                    if (CPL == 4) asm volatile("ds_read_b128 %0, %13\n\t" S5_WINDOW_READS S5_HEAD_OPERANDS);
                    else {
                        v2f m2v;
                        asm volatile("ds_read_b64 %0, %13\n\t" S5_WINDOW_READS
                                     : "=&v"(m2v), "=&v"(xr), "=&v"(h01), "=&v"(h23),
                                       "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]),
                                       "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8])
                                     : "v"(maddr), "v"(xaddr), "v"(haddr), "v"(vaddr) : "memory");
                        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(m2v));
                        mt.x = m2v.x; mt.y = m2v.y; mt.z = 0.f; mt.w = 0.f;
                    }
                }
                if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s5_stamp<PROF>(pf, pf.a); }      // window-read wait
                // (1) match row landed: what depends on it alone is computed under the remaining reads
                if (CPL == 4) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(mt));
                const float m2[4] = {mt.x, mt.y, mt.z, mt.w};
                float base[CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) base[c] = lmax[c] + m2[c] * S5_LOG2E;            // log2(strongest link * emission)
                // (2) exponent of the superblock read + the block maxima of the previous row: the two write exponents of this row
                asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(xr), "+v"(h01), "+v"(h23));
                {
                    const float mid = fmaxf(h01.y, h23.x);
                    XA = s5_expo(fmaxf(mid, h23.y));
                    XB = s5_expo(fmaxf(mid, h01.x));
                }
                // (3) the window, 4 values at a time as it lands: two accumulator sets so consecutive groups do not chain
                v2f acc[2][CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) { acc[0][c].x = acc[0][c].y = 0.f; acc[1][c].x = acc[1][c].y = 0.f; }
#define S5_GROUP(k, n) \
                { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(pv[k])); \
                  v2f wa, wc; wa.x = pv[k].x; wa.y = pv[k].y; wc.x = pv[k].z; wc.y = pv[k].w; \
                  _Pragma("unroll") for (int c = 0; c < CPL; ++c) { \
                      acc[(k) & 1][c] = __builtin_elementwise_fma(wa, E2[c][2 * (k)], acc[(k) & 1][c]); \
                      acc[(k) & 1][c] = __builtin_elementwise_fma(wc, E2[c][2 * (k) + 1], acc[(k) & 1][c]); } }
                S5_GROUP(0, 8) S5_GROUP(1, 7) S5_GROUP(2, 6) S5_GROUP(3, 5) S5_GROUP(4, 4)
                S5_GROUP(5, 3) S5_GROUP(6, 2) S5_GROUP(7, 1) S5_GROUP(8, 0)
#undef S5_GROUP
                float S[CPL];
                bool flag[CPL];
                bool need_fb = false;
#pragma unroll
                for (int c = 0; c < CPL; ++c) S[c] = (acc[0][c].x + acc[0][c].y) + (acc[1][c].x + acc[1][c].y);
                if (PROF) { asm volatile("" : "+v"(S[0]), "+v"(S[CPL - 1])); s5_stamp<PROF>(pf, pf.b); }               // FMA stretch
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    flag[c] = !(S[c] >= sthr[c] && S[c] <= 0x1p126f);              // too small (0 included) or inf / NaN
                    need_fb |= flag[c];
                    // S = 0 (structurally unreachable, or every predecessor dead): log2 -> -inf, which is the cell's value
                    a2[c] = __builtin_amdgcn_logf(S[c]) + (xr + base[c]);
                    // beta: the reference's K3 only visits columns >= t (dag_loss.cu: j in t..L_b-1); a column left of the diagonal
                    // can still reach the end by the backward recursion alone, so it has to be masked (alpha needs no mask: its
                    // unreachable cells sum to exactly 0)
                    if (BETA && j + c < t) a2[c] = NEG_INF;
                }
                if (__builtin_expect(__any(need_fb), 0)) {
                    // a flagged cell outside the reachable region is just -inf (its sum is exactly 0); inside it, the sum was not trustworthy
                    bool still = false;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        if (flag[c]) {
                            if (!cell_active(j + c, t)) { a2[c] = NEG_INF; flag[c] = false; }
                            else a2[c] = NEG_INF;               // settled below
                        }
                        still |= flag[c];
                    }
                    if (still && p.dbg != 4 && p.dbg != 2) {                   // (DSP_DEBUG=nofallback: timing experiment, WRONG results)
                        if (p.dbg == 1) atomicAdd(&p.counters[3], 1u);
                        // (0) the DP's diagonal cell (vertex = row, counted from the direction's start) has ONE live transition; with peaked
                        //     scores it sits hundreds of binades under its neighbours and lands here on every row: no sum needed
                        bool still2 = false;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            const int dl = BETA ? (Lb - Tb + 1 + t - (j + c)) : (j + c - t + 1);
                            if (flag[c] && dl == 1 && sthr[c] == 0x1p-97f) {
                                const float ap = Abuf[prv * RL + li0 + c + (BETA ? 1 : -1)];
                                const float e1 = Eval(c, 1);
                                a2[c] = (e1 > 0.f && ap != NEG_INF) ? (ap + __builtin_amdgcn_logf(e1) + base[c]) : NEG_INF;
                                flag[c] = false;
                            }
                            still2 |= flag[c];
                        }
                        if (still2) {
                            // (a) MEDIUM path: redo the flagged column against ITS OWN maximum from the exact log2-domain row (read from LDS
                            //     eight values at a time: the path is rare and must not cost the main loop registers)
                            const float* arow = Abuf + prv * RL + li0;
#pragma unroll
                            for (int c = 0; c < CPL; ++c) {
                                if (flag[c]) {
                                    // live transitions d_lo .. d_hi: bounded by the diagonal on one side and the reach frontier on the other;
                                    // the 32 terms are walked in chunks of 8 that the wave skips when no flagged lane has a live one there
                                    const int dlim = min(32, max(0, BETA ? (Lb - Tb + 1 + t - (j + c)) : (j + c - t + 1)));
                                    const int dlo = max(1, BETA ? (Lb - 1 - (Tb - 2 - t) * TR - (j + c)) : (j + c - (t - 1) * TR));
                                    float cmx = NEG_INF;
#pragma unroll
                                    for (int d0 = 1; d0 <= 32; d0 += 8) {
                                        if (__any(dlim >= d0 && dlo <= d0 + 7)) {
#pragma unroll
                                            for (int d = d0; d < d0 + 8; ++d) cmx = fmaxf(cmx, arow[c + (BETA ? d : -d)]);
                                        }
                                    }
                                    float sc = 0.f;
#pragma unroll
                                    for (int d0 = 1; d0 <= 32; d0 += 8) {
                                        if (__any(dlim >= d0 && dlo <= d0 + 7)) {
#pragma unroll
                                            for (int d = d0; d < d0 + 8; ++d)
                                                sc = fmaf(__builtin_amdgcn_exp2f(arow[c + (BETA ? d : -d)] - cmx), Eval(c, d), sc);
                                        }
                                    }
                                    // (values <= 1 here: a weight that fp32 flushed drops a term < 2^-120 against a sum >= 2^-97)
                                    if (sc >= 0x1p-97f) { a2[c] = __builtin_amdgcn_logf(sc) + cmx + base[c]; flag[c] = false; }
                                }
                            }
                            // (b) EXACT path for what is left (flushed weights, or a column whose own sum is still under the threshold)
#pragma unroll 1
                            for (int c = 0; c < CPL; ++c) {
                                bool fc = false; float mm = 0.f;
#pragma unroll
                                for (int cc = 0; cc < CPL; ++cc) if (cc == c) { fc = flag[cc]; mm = m2[cc]; }
                                if (!fc) continue;
                                float amax = NEG_INF;
                                for (int d = 1; d <= 32; ++d) amax = fmaxf(amax, Abuf[prv * RL + li0 + c + (BETA ? d : -d)]);
                                float r = NEG_INF;
                                if (amax != NEG_INF) {
                                    { const u32 slot = atomicAdd(&p.counters[2], 1u); if (!p.dbg && slot < 14) { p.counters[8 + 4 * slot] = (u32)b | (BETA ? 0x100u : 0u); p.counters[9 + 4 * slot] = (u32)t; p.counters[10 + 4 * slot] = (u32)(j + c); p.counters[11 + 4 * slot] = __float_as_uint(xr); } }
                                    float mx = NEG_INF, sum = 0.f;
                                    for (int d0 = 1; d0 <= 32; d0 += 8) {
                                        float lk[8];
#pragma unroll
                                        for (int u = 0; u < 8; ++u) {          // raw links re-read from HBM 8 at a time (independent loads)
                                            const int d = d0 + u;
                                            const int row = BETA ? (j + c) : (j + c - d);
                                            const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                            const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                            lk[u] = ok ? raw * S5_LOG2E : NEG_INF;
                                        }
#pragma unroll
                                        for (int u = 0; u < 8; ++u) {
                                            const int d = d0 + u;
                                            const float v = Abuf[prv * RL + li0 + c + (BETA ? d : -d)] + lk[u];
                                            const float nm = fmaxf(mx, v);
                                            if (nm != NEG_INF) sum = sum * __builtin_amdgcn_exp2f(mx - nm) + __builtin_amdgcn_exp2f(v - nm);
                                            mx = nm;
                                        }
                                    }
                                    if (mx != NEG_INF) r = __builtin_amdgcn_logf(sum) + mx + mm * S5_LOG2E;
                                }
#pragma unroll
                                for (int cc = 0; cc < CPL; ++cc) if (cc == c) a2[cc] = r;
                            }
                        }
                    }
                }
            }
            // ---- write the row: both superblock copies, the exact row, the HBM output; then the block maximum for the next row's bounds
            float gm = a2[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) gm = fmaxf(gm, a2[c]);
            if (CPL == 4) {
                float4 va, vb;
                va.x = __builtin_amdgcn_exp2f(a2[0] - XA); va.y = __builtin_amdgcn_exp2f(a2[1] - XA);
                va.z = __builtin_amdgcn_exp2f(a2[2] - XA); va.w = __builtin_amdgcn_exp2f(a2[3] - XA);
                vb.x = __builtin_amdgcn_exp2f(a2[0] - XB); vb.y = __builtin_amdgcn_exp2f(a2[1] - XB);
                vb.z = __builtin_amdgcn_exp2f(a2[2] - XB); vb.w = __builtin_amdgcn_exp2f(a2[3] - XB);
                if (kb < NSB) *reinterpret_cast<float4*>(SBarr + (cur * NSB + kb) * S5_SBS + pos) = va;
                if (kb >= 1) *reinterpret_cast<float4*>(SBarr + (cur * NSB + kb - 1) * S5_SBS + 32 + pos) = vb;
                *reinterpret_cast<float4*>(Abuf + cur * RL + li0) = make_float4(a2[0], a2[1], a2[2], a2[3]);
                if (col_ok) *reinterpret_cast<float4*>(O + (size_t)t * L + j) = make_float4(a2[0] * S5_LN2, a2[1] * S5_LN2, a2[2] * S5_LN2, a2[3] * S5_LN2);
            } else {
                float2 va, vb;
                va.x = __builtin_amdgcn_exp2f(a2[0] - XA); va.y = __builtin_amdgcn_exp2f(a2[1] - XA);
                vb.x = __builtin_amdgcn_exp2f(a2[0] - XB); vb.y = __builtin_amdgcn_exp2f(a2[1] - XB);
                if (kb < NSB) *reinterpret_cast<float2*>(SBarr + (cur * NSB + kb) * S5_SBS + pos) = va;
                if (kb >= 1) *reinterpret_cast<float2*>(SBarr + (cur * NSB + kb - 1) * S5_SBS + 32 + pos) = vb;
                *reinterpret_cast<float2*>(Abuf + cur * RL + li0) = make_float2(a2[0], a2[1]);
                if (col_ok) *reinterpret_cast<float2*>(O + (size_t)t * L + j) = make_float2(a2[0] * S5_LN2, a2[1] * S5_LN2);
            }
            gm = s5_group_max<(CPL == 4) ? 3 : 4>(gm);
            if (pos == 0) {
                Hrow[cur * HR + kb + 1] = gm;
                if (kb < NSB) Xsb[cur * XS + kb] = XA;
            }
            if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s5_stamp<PROF>(pf, pf.c); }              // tail + stores
            s5_barrier();
        }
        // rows the recurrence never reaches
        if (col_ok) for (int t = Tb; t < T; ++t) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) O[(size_t)t * L + j + c] = NEG_INF;
        }
        (void)BLK_LANES;
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * L;
            float* slot = Mring + (size_t)(itr % S5_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 256), 16, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < S5_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s5_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + S5_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                // rows it+2 .. it+7 may stay in flight: 6*DPR DMAs younger than row it+1's
                if (DPR == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (DPR == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s5_stamp<PROF>(pf, pf.a); s5_barrier(); s5_stamp<PROF>(pf, pf.d);
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: neighbour strip's halo -> LDS, one row AHEAD for its maximum
        const bool hl = lane < S5_TRP;
        u64 g[S5_CH];
#pragma unroll
        for (int k = 0; k < S5_CH; ++k) g[k] = 0;
        auto load_row = [&](int itr) -> u64 {
            if (itr < nrows && hl) { const int t = BETA ? (Tb - 1 - itr) : itr; return s5_gran_load(hin + (size_t)t * S5_TRP + lane); }
            return 0;
        };
        auto wait_row = [&](u64 x, int itr) -> float {               // spin until row itr's granules carry this launch's tag
            float hv = NEG_INF;
            if (has_producer && hl && itr < nrows) {
                const int t = BETA ? (Tb - 1 - itr) : itr;
                const u32 want = p.tag_base + 1u + (u32)t;
                u32 spins = 0;
                while (!__all((u32)(x >> 32) == want)) {
                    if ((u32)(x >> 32) != want) x = s5_gran_load(hin + (size_t)t * S5_TRP + lane);
                    if (++spins > S5_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                hv = __uint_as_float((u32)x);
            }
            return hv;
        };
        auto max32 = [&](float v) -> float {                         // maximum over lanes 0..31, wave-uniform
            v = s5_group_max<4>(v);
            return fmaxf(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < S5_CH; ++k) g[k] = load_row(k);
        }
        __syncthreads();                         // link tile consumed
        s5_barrier();                            // prologue barrier (Hrow initialised, match row 0 landed)
        const float seedX = seed_here ? (ceilf(Mring[seed_col - j0] * S5_LOG2E) - S5_BIAS) : 0.f;
        float hv_next = wait_row(g[0], 0);
        if (has_producer) g[0] = load_row(S5_CH);
        float hh_next = max32(hl ? hv_next : NEG_INF);
        const int e_halo = BETA ? NB : 1, e_next = BETA ? NB + 1 : 0;          // Hrow entries of the halo block / of the NEXT halo row
        const int k_halo = BETA ? NSB - 1 : 0;                                  // the superblock the halo block belongs to
        for (int itb = 0; itb < nrows; itb += S5_CH) {
#pragma unroll
            for (int k = 0; k < S5_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int cur = it & 1, prv = cur ^ 1;
                const float hv = hv_next, hh = hh_next;
                // the halo superblock's exponent for this row: the bound every compute lane of that superblock evaluates too
                float Xh = seedX;
                if (it > 0) {
                    const float* hp = Hrow + prv * HR + (BETA ? NB - 1 : 0);
                    Xh = s5_expo(fmaxf(fmaxf(hp[0], hp[1]), hp[2]));
                }
                if (hl) {
                    Abuf[cur * RL + halo_li0 + lane] = hv;
                    SBarr[(cur * NSB + k_halo) * S5_SBS + (BETA ? 32 : 0) + lane] = __builtin_amdgcn_exp2f(hv - Xh);
                }
                // next row's halo: needed NOW for its maximum (slot (k+1) % CH holds it; requested CH-1 rows ago)
                hv_next = wait_row(g[(k + 1) % S5_CH], it + 1);
                if (has_producer) g[(k + 1) % S5_CH] = load_row(it + 1 + S5_CH);
                hh_next = max32(hl ? hv_next : NEG_INF);
                if (lane == 0) {
                    Hrow[cur * HR + e_halo] = hh;
                    Hrow[cur * HR + e_next] = hh_next;
                    if (!BETA) Xsb[cur * XS + 0] = Xh;
                }
                s5_stamp<PROF>(pf, pf.a); s5_barrier(); s5_stamp<PROF>(pf, pf.d);
            }
        }
    } else {
        // =========================================================== publish wave: boundary columns -> granules
        const bool pl = has_consumer && lane < S5_TRP;
        __syncthreads();                         // link tile consumed
        s5_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete (barrier it-1 passed); compute now writes the other buffer
                const int tp = BETA ? (Tb - it) : (it - 1);
                const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : W) + lane];
                s5_gran_store(hout + (size_t)tp * S5_TRP + lane, p.tag_base + 1u + (u32)tp, v);
            }
            s5_stamp<PROF>(pf, pf.a); s5_barrier(); s5_stamp<PROF>(pf, pf.d);
        }
        if (pl && nrows > 0) {
            const int it = nrows;
            const int tp = BETA ? (Tb - it) : (it - 1);
            const float v = Abuf[((it - 1) & 1) * RL + (BETA ? 0 : W) + lane];
            s5_gran_store(hout + (size_t)tp * S5_TRP + lane, p.tag_base + 1u + (u32)tp, v);
        }
    }
    if (PROF && profwg && lane == 0 && wave < 11) {
        u32* o = p.counters + 8 + wave * 4;
        o[0] = (u32)pf.a; o[1] = (u32)pf.b; o[2] = (u32)pf.c; o[3] = (u32)pf.d;
    }
}

// 512-column strips are meant to run TWO workgroups per CU (alpha and beta strips, or neighbours, with independent barriers: one's FMA
// stretch covers the other's tail and LDS round trip): 14 waves per CU = 4 per SIMD -> at most 128 registers
template <int W, int CPL, bool PROF>
__global__ __launch_bounds__(W / CPL + 192, (W == 512 && CPL == 2) ? 4 : 1) void dag_strip5_kernel(S5Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NT = W / CPL;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);          // 16-byte header; everything else starts at +16
    const int tid = threadIdx.x;
    if (tid == 0) *s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = *s_ticket;
    const int per = p.ndir * p.B;
    const int so = (int)(ticket / per);
    const int rem = (int)(ticket % per);
    const bool is_beta = (p.alpha == nullptr || (p.ndir == 2 && rem >= p.B));
    const int b = rem % p.B;
    const int dirslot = (p.ndir == 2 && rem >= p.B) ? 1 : 0;
    const int s = is_beta ? (p.NS - 1 - so) : so;
    const int j0 = s * W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {                    // nothing reachable in this strip: -inf everywhere, no hand-off
        if (tid < NT) {
            const int j = j0 + CPL * tid;
            if (j < L) {
                float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) O[(size_t)t * L + j + c] = NEG_INF;
            }
        }
        return;
    }
    const bool profwg = PROF && ticket == 0;
    if (is_beta) strip5_body<W, CPL, true, PROF>(p, smem_raw + 16, b, s, dirslot, so, profwg);
    else strip5_body<W, CPL, false, PROF>(p, smem_raw + 16, b, s, dirslot, so, profwg);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip5_supported(const void* match, const void* alpha, const void* beta, int L, int TR)
{
    if (TR > 32 || (L & 3)) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta;
    return (a & 15) == 0;
}

template <int W, int CPL>
static int launch_one_s5(const S5Params& p, int nwg, hipStream_t st)
{
    constexpr int RL = W + 32, NB = W / 32 + 1, NSB = W / 32;
    constexpr int HR = (NB + 2 + 3) & ~3, XS = (NSB + 3) & ~3;
    const size_t lds_main = (size_t)(2 * RL + 2 * NSB * S5_SBS + 2 * HR + 2 * XS + S5_RING * W) * 4 + 16;
    const size_t lds_tile = (size_t)(W + 32) * 33 * 4 + 16;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    if (p.dbg == 2) {
        auto kp = dag_strip5_kernel<W, CPL, true>;
        (void)hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kp, dim3((unsigned)nwg), dim3(W / CPL + 192), lds, st, p);
        return check_launch("dag_loss_fwd(strip5, prof)");
    }
    auto k = dag_strip5_kernel<W, CPL, false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(W / CPL + 192), lds, st, p);
    return check_launch("dag_loss_fwd(strip5)");
}

static int g_s5_cpl = 0;      // 0 = auto; 2 / 4 pinned by dsp_dag_set_option("s5_cpl", n) (sweeps)
static int g_s5_w = 0;        // 0 = auto; 512 / 1024 pinned by dsp_dag_set_option("s5_w", n)
void set_s5_cpl(int v) { g_s5_cpl = v; }
void set_s5_w(int v) { g_s5_w = v; }

int launch_dag_strip5(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                      float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    // strip width: 1024 columns when that still yields >= ~200 workgroups, else 512
    const int ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
    const bool wide = g_s5_w ? (g_s5_w == 1024) : ((long)ndir * B * ns1024 >= 200);
    const int NS = wide ? ns1024 : ns512;
    S5Params p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir;
    { const char* e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "medium")) ? 1 : (e && !strcmp(e, "prof")) ? 2 : (e && !strcmp(e, "nofallback")) ? 4 : 0; }
    const size_t halo_bytes = (size_t)ndir * B * NS * T * S5_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const int nwg = ndir * B * NS;
    const int cpl = g_s5_cpl ? g_s5_cpl : 2;
    if (cpl == 4) return wide ? launch_one_s5<1024, 4>(p, nwg, st) : launch_one_s5<512, 4>(p, nwg, st);
    return wide ? launch_one_s5<1024, 2>(p, nwg, st) : launch_one_s5<512, 2>(p, nwg, st);
}

}  // namespace dsp
