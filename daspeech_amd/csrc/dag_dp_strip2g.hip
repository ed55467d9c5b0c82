// dag_dp_strip2g.hip — banded DAG DP for windows 33 .. 64 in EXP SPACE: K2 alpha || K3 beta, two vertices per lane (r06).
//
// The TR <= 32 kernel (dag_dp_strip4g.hip: read its header first) keeps a lane's 4 x 32 transition weights in registers.  A 64-wide window
// has the same 128 weights for TWO vertices, so this file is the same machine with the lane cut in half:
//   * column strips of 512 vertices, one workgroup (4 compute waves + loader / fetch / publish helper waves) per (sample, direction, strip)
//     for all T rows, tagged-granule hand-off of the 64 boundary columns, tickets — as strip4g;
//   * the previous row in LDS as plain values V = 2^(a2 - X) with ONE integer exponent X per group of 4 vertices (= a PAIR of lanes: the
//     pair's maximum meets through one quad-permute), the exact log2-domain row beside it for the hand-off and the fallback paths;
//   * a lane reads the 68-value window that starts at the 16-byte boundary under its first predecessor (17 ds_read_b128 + 17 exponents in one
//     issue group); its weights are stored against THAT window, so the odd lane of a pair — whose vertices sit two elements further in —
//     simply holds its weights two slots later (zeros where an element is no predecessor);
//   * exactness guard and fallbacks as strip4g: scaled values <= 2^120, a sum S >= 2^-97 is exact to fp32 (68 terms of at most 2^-126 lost);
//     smaller sums take the single-transition shortcut on the DP's diagonal, then the exact log-space form (previous row from LDS,
//     transitions re-read from HBM).  No register-only "medium" pass: at two vertices per lane its 2 x 64 window does not fit beside the
//     weights, and the exact path serves the same cells.
// Until r06 these windows ran the log-space strips of dag_dp_banded.hip (2 x 64 v_exp per lane-row: 2.5 ms at C2 / TR = 64).
// Replaces calculate_alpha_kernel / calculate_beta_kernel (dag_loss.cu:40-140,178-274) for 32 < translen <= 64.
#include "common.h"
#include <stdlib.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float h2_v2f __attribute__((ext_vector_type(2)));
typedef float h2_v4f __attribute__((ext_vector_type(4)));
typedef int h2_v2i __attribute__((ext_vector_type(2)));

struct H2Params {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word, counters[2] = exact-path cells
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int ldm, ldo;                             // row pitches (elements) of match and of alpha / beta (r06: >= L; the pad columns L .. round4(L)-1 of alpha / beta get -inf)
};

constexpr int H2_NT = 256;                    // compute lanes
constexpr int H2_W = 2 * H2_NT;               // 512 columns per strip
constexpr int H2_TRP = 64;                    // window / halo width
constexpr int H2_RL = H2_W + H2_TRP;          // LDS row: strip + halo
constexpr int H2_GL = H2_RL / 4;              // groups per LDS row
constexpr int H2_RING = 8;
constexpr int H2_CH = 4;                      // halo prefetch distance of the fetch wave (rows)
constexpr int H2_NEG = -(1 << 30);            // "dead" exponent
constexpr u32 H2_SPIN_LIMIT = 1u << 22;
constexpr float H2_LOG2E = 1.4426950408889634f;
constexpr float H2_LN2 = 0.6931471805599453f;
constexpr float H2_BIAS = 120.f;

__device__ __forceinline__ u64 h2_gran_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void h2_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void h2_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// maximum with the other lane of the pair (lanes 2m, 2m+1): quad_perm [1,0,3,2]
__device__ __forceinline__ float h2_pair_max(float v) {
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
}

template <bool BETA>
__device__ __forceinline__ void strip2g_body(const H2Params& p, char* smem_raw, int b, int s, int dirslot, int so)
{
    constexpr int W = H2_W, RL = H2_RL, GL = H2_GL, NCW = H2_NT / 64, DPR = W / 256, TRP = H2_TRP;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  a2 = alpha * log2(e)  (exact row, log2 domain)
    float* Vbuf = Abuf + 2 * RL;                               // [2][RL]  V = 2^(a2 - X[group])
    int* Xbuf = reinterpret_cast<int*>(Vbuf + 2 * RL);         // [2][GL]  group exponents; group gi covers li 4gi..4gi+3
    float* Mring = reinterpret_cast<float*>(Xbuf + 2 * GL);    // [RING][W] match rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, L = p.L, TR = p.TR;
    const int j0 = s * W;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * p.ldm;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = (BETA ? p.beta : p.alpha) + (size_t)b * T * p.ldo;
    const int LDO = p.ldo, LPAD = min(p.ldo, (L + 3) & ~3);
    const int nrows = Tb;

    const bool has_producer = so > 0 && (BETA ? (j0 + W < Lb) : true);
    const bool has_consumer = BETA ? (s > 0) : (s < p.NS - 1 && j0 + W < Lb);
    const int prod_strip = BETA ? s + 1 : s - 1;
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * TRP;
    // LDS geometry: alpha li = col - j0 + 64 (halo [0,64)); beta li = col - j0 (halo [W, W+64))
    const int halo_li0 = BETA ? W : 0;
    const int own_li0 = BETA ? 0 : TRP;

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once): tile[r][d] = links[rlo + r][d] (pitch 65), -inf outside
    // the graph / beyond TR.  The tile overlays the main-loop buffers, which are not live yet.
    {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = H2_NT + 192, RPP = NTHR / 64;     // rows per pass (7)
        const int rlo = BETA ? j0 : (j0 - TRP);
        const int dd = tid & 63, r0 = tid >> 6;
        for (int rb = r0; rb < W + TRP; rb += 8 * RPP) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                       // 8 independent (clamped, unconditional) loads in flight
                const int i = rlo + rb + u * RPP;
                const bool ok = dd < TR && i >= 0 && i < L;
                const float raw = K[(size_t)(ok ? i : 0) * TR + (ok ? dd : 0)];
                v[u] = ok ? raw : NEG_INF;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int r = rb + u * RPP; if (r < W + TRP) tile[r * 65 + dd] = v[u]; }
        }
    }
    __syncthreads();

    if (wave < NCW) {
        // =========================================================== compute waves
        __builtin_amdgcn_s_setprio(2);
        const int l = tid;                       // lane's vertex pair
        const int par = l & 1;                   // position inside the 4-vertex group: the window starts 2 * par elements before the lane's own
        const int j = j0 + 2 * l;
        const bool col_ok = j < L;
        auto cell_active = [&](int col, int t) -> bool {
            if (!BETA) return col >= t && col < Lb && (long)col <= (long)t * TR;
            const int rem = Tb - 1 - t, gap = Lb - 1 - col;
            return col >= t && gap >= rem && (long)gap <= (long)rem * TR;
        };
        // window element (index into the lane's 68-value window) of (vertex c, distance d): alpha 64 + 2 par + c - d, beta 2 par + c + d
        auto qidx = [&](int c, int d) -> int { return BETA ? (2 * par + c + d) : (TRP + 2 * par + c - d); };
        // LDS row index of the same element
        auto liidx = [&](int c, int d) -> int { return BETA ? (2 * l + c + d) : (TRP + 2 * l + c - d); };
        const float* tile = reinterpret_cast<const float*>(smem_raw);
        // raw (log2-domain) transition of (vertex c, distance d) out of the tile; -inf outside the window / the graph
        auto raw_link = [&](int c, int d) -> float {
            float v;
            if (!BETA) v = tile[(2 * l + c - d + TRP) * 65 + (d - 1)];
            else { v = tile[(2 * l + c) * 65 + (d - 1)]; if (j + c + d >= Lb) v = NEG_INF; }
            return v * H2_LOG2E;
        };
        float lmax[2], sthr[2];
        h2_v2f E2[2][34];                        // E2[c][i] = (weight of window element 2i, of 2i+1) for vertex c; 0 where the element is no predecessor
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float mx = NEG_INF;
#pragma unroll 8
            for (int d = 1; d <= TRP; ++d) mx = fmaxf(mx, raw_link(c, d));
            if (mx == NEG_INF) mx = 0.f;
            lmax[c] = mx;
            bool flushed = false;                // a finite link more than ~120 binades under the column's strongest (see strip4g)
#pragma unroll
            for (int i = 0; i < 34; ++i) {
                float w2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * i + h;
                    const int d = BETA ? (q - 2 * par - c) : (TRP + 2 * par + c - q);          // runtime (lane parity): the tile is read at a computed address
                    float r = NEG_INF;
                    if (d >= 1 && d <= TRP) r = raw_link(c, d);
                    w2[h] = __builtin_amdgcn_exp2f(r - mx);
                    flushed |= (r != NEG_INF) & (r - mx < -120.f);
                }
                E2[c][i].x = w2[0]; E2[c][i].y = w2[1];
            }
            sthr[c] = flushed ? 0x1p30f : 0x1p-97f;
        }
        // weight of (vertex c, distance d) recovered from the pair layout: the slot depends on the lane's parity, both candidates are static
        auto Eval = [&](int c, int d) -> float {
            const int q0 = BETA ? (c + d) : (TRP + c - d), q1 = q0 + 2;
            const float e0 = (q0 & 1) ? E2[c][q0 >> 1].y : E2[c][q0 >> 1].x;
            const float e1 = (q1 & 1) ? E2[c][q1 >> 1].y : E2[c][q1 >> 1].x;
            return par ? e1 : e0;
        };
        // first (alpha) / last (beta) window group that holds a predecessor of either vertex: window element of distance TR is 64 + 2 par - TR
        // (alpha, vertex 0) resp. 2 par + 1 + TR (beta, vertex 1)
        const int gcut = BETA ? ((2 * par + 1 + TR) >> 2) : ((TRP + 2 * par - TR) >> 2);
        __syncthreads();                         // tile consumed: the loader may start filling the ring over it
        h2_barrier();                            // prologue barrier: match row 0 is in the ring

        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            float a2[2] = {NEG_INF, NEG_INF};
            if (it == 0) {
                const float2 mt = *reinterpret_cast<const float2*>(Mring + (size_t)(it % H2_RING) * W + 2 * l);
                const float m2[2] = {mt.x, mt.y};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool seed = BETA ? (j + c == Lb - 1) : (j + c == 0);
                    if (seed) a2[c] = m2[c] * H2_LOG2E;
                }
            } else {
                // ---- row head: match (8 bytes), 17 group exponents, the 68-value window — one issue group, counted waits
                h2_v2f mt; h2_v2i xa[8]; int x16; h2_v4f pv[17];
                {
                    const u32 maddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Mring + (size_t)(it % H2_RING) * W + 2 * l);
                    const u32 xaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Xbuf + prv * GL + (l >> 1));
                    const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Vbuf + prv * RL + 4 * (l >> 1));
                    asm volatile("ds_read_b64 %0, %27\n\t"
                                 "ds_read2_b32 %1, %28 offset1:1\n\t"
                                 "ds_read2_b32 %2, %28 offset0:2 offset1:3\n\t"
                                 "ds_read2_b32 %3, %28 offset0:4 offset1:5\n\t"
                                 "ds_read2_b32 %4, %28 offset0:6 offset1:7\n\t"
                                 "ds_read2_b32 %5, %28 offset0:8 offset1:9\n\t"
                                 "ds_read2_b32 %6, %28 offset0:10 offset1:11\n\t"
                                 "ds_read2_b32 %7, %28 offset0:12 offset1:13\n\t"
                                 "ds_read2_b32 %8, %28 offset0:14 offset1:15\n\t"
                                 "ds_read_b32 %9, %28 offset:64\n\t"
                                 "ds_read_b128 %10, %29\n\t"
                                 "ds_read_b128 %11, %29 offset:16\n\t"
                                 "ds_read_b128 %12, %29 offset:32\n\t"
                                 "ds_read_b128 %13, %29 offset:48\n\t"
                                 "ds_read_b128 %14, %29 offset:64\n\t"
                                 "ds_read_b128 %15, %29 offset:80\n\t"
                                 "ds_read_b128 %16, %29 offset:96\n\t"
                                 "ds_read_b128 %17, %29 offset:112\n\t"
                                 "ds_read_b128 %18, %29 offset:128\n\t"
                                 "ds_read_b128 %19, %29 offset:144\n\t"
                                 "ds_read_b128 %20, %29 offset:160\n\t"
                                 "ds_read_b128 %21, %29 offset:176\n\t"
                                 "ds_read_b128 %22, %29 offset:192\n\t"
                                 "ds_read_b128 %23, %29 offset:208\n\t"
                                 "ds_read_b128 %24, %29 offset:224\n\t"
                                 "ds_read_b128 %25, %29 offset:240\n\t"
                                 "ds_read_b128 %26, %29 offset:256"
                                 : "=&v"(mt), "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xa[2]), "=&v"(xa[3]), "=&v"(xa[4]), "=&v"(xa[5]), "=&v"(xa[6]), "=&v"(xa[7]), "=&v"(x16),
                                   "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8]),
                                   "=&v"(pv[9]), "=&v"(pv[10]), "=&v"(pv[11]), "=&v"(pv[12]), "=&v"(pv[13]), "=&v"(pv[14]), "=&v"(pv[15]), "=&v"(pv[16])
                                 : "v"(maddr), "v"(xaddr), "v"(vaddr)
                                 : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(mt));                     // (LDS returns in order; the counter saturates at 15)
                const float m2[2] = {mt.x, mt.y};
                float base[2]; bool okc[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) { okc[c] = cell_active(j + c, t); base[c] = lmax[c] + m2[c] * H2_LOG2E; }
                asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]), "+v"(x16));
                // 26 LDS operations are in flight after the match; lgkmcnt only counts to 15: wait for "at most 15 outstanding" twice — the
                // 17 window reads minus two — then treat the exponents as landed only once the count says so: the exponents are operations
                // 2 .. 10 of 27, so they have returned when at most 17 are outstanding.  15 is the tightest wait the counter can express.
                int xw[17];
#pragma unroll
                for (int g = 0; g < 8; ++g) { xw[2 * g] = xa[g].x; xw[2 * g + 1] = xa[g].y; }
                xw[16] = x16;
                // TR < 61: the 68-value window reaches past the lane's real predecessors.  Those slots hold LIVE cells with zero weight; next to
                // the reachability frontier they sit hundreds of binades above the one real predecessor, which would flush against them and send
                // every frontier cell to the exact path (strip4g's r05 finding, there solved in its register-only redo).  Groups without a
                // predecessor of either vertex are therefore dropped from the window here: dead exponent, shifted to 0, out of the reference.
                if (TR < 61) {
#pragma unroll
                    for (int g = 0; g < 17; ++g) if (BETA ? (g > gcut) : (g < gcut)) xw[g] = H2_NEG;
                }
                int refi = xw[0];
#pragma unroll
                for (int g = 1; g < 17; ++g) refi = max(refi, xw[g]);
                const bool any_live = refi != H2_NEG;
                if (!any_live) refi = 0;
                int kg[17];
#pragma unroll
                for (int g = 0; g < 17; ++g) kg[g] = xw[g] - refi;                 // <= 0; hugely negative for dead groups
                h2_v2f S2[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c) { S2[c][0].x = 0.f; S2[c][0].y = 0.f; S2[c][1].x = 0.f; S2[c][1].y = 0.f; }
#define H2_GROUP(k, n) \
                { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(pv[k])); \
                  h2_v2f wa, wb; wa.x = ldexpf(pv[k].x, kg[k]); wa.y = ldexpf(pv[k].y, kg[k]); wb.x = ldexpf(pv[k].z, kg[k]); wb.y = ldexpf(pv[k].w, kg[k]); \
                  _Pragma("unroll") for (int c = 0; c < 2; ++c) { \
                      S2[c][0] = __builtin_elementwise_fma(wa, E2[c][2 * k], S2[c][0]); \
                      S2[c][1] = __builtin_elementwise_fma(wb, E2[c][2 * k + 1], S2[c][1]); } }
                H2_GROUP(0, 15) H2_GROUP(1, 15) H2_GROUP(2, 14) H2_GROUP(3, 13) H2_GROUP(4, 12) H2_GROUP(5, 11) H2_GROUP(6, 10) H2_GROUP(7, 9)
                H2_GROUP(8, 8) H2_GROUP(9, 7) H2_GROUP(10, 6) H2_GROUP(11, 5) H2_GROUP(12, 4) H2_GROUP(13, 3) H2_GROUP(14, 2) H2_GROUP(15, 1) H2_GROUP(16, 0)
#undef H2_GROUP
                float S[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) { const h2_v2f t2 = S2[c][0] + S2[c][1]; S[c] = t2.x + t2.y; }
                const float ref = (float)refi;
                // ---- row tail
                bool need_fb = false;
                bool flag[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool okl = okc[c] & any_live;
                    flag[c] = okl & !(S[c] >= sthr[c] && S[c] <= 0x1p126f);       // too small (or inf / NaN)
                    need_fb |= flag[c];
                    a2[c] = (okl & !flag[c]) ? (__builtin_amdgcn_logf(S[c]) + (ref + base[c])) : NEG_INF;
                }
                if (__builtin_expect(need_fb, 0)) {
#pragma unroll 1
                    for (int c = 0; c < 2; ++c) {
                        const bool fc = c == 0 ? flag[0] : flag[1];
                        if (!fc) continue;
                        const float bc = c == 0 ? base[0] : base[1];
                        const float st = c == 0 ? sthr[0] : sthr[1];
                        float r = NEG_INF;
                        // (0) the DP's diagonal cell has ONE live transition: a2 = a2_prev(predecessor) + log2(weight) + base, no sum
                        const int dl = BETA ? (Lb - Tb + 1 + t - (j + c)) : (j + c - t + 1);
                        bool done = false;
                        if (dl == 1 && st == 0x1p-97f) {
                            const float ap = Abuf[prv * RL + liidx(c, 1)];
                            const float e1 = c == 0 ? Eval(0, 1) : Eval(1, 1);
                            if (e1 > 0.f) { r = (ap != NEG_INF) ? (ap + __builtin_amdgcn_logf(e1) + bc) : NEG_INF; done = true; }
                        }
                        if (!done) {
                            // (b) exact log-space value: previous row from LDS, raw transitions re-read from HBM
                            float amax = NEG_INF;
                            for (int d = 1; d <= TRP; ++d) amax = fmaxf(amax, Abuf[prv * RL + liidx(c, d)]);
                            if (amax != NEG_INF) {
                                atomicAdd(&p.counters[2], 1u);
                                float mx = NEG_INF, sum = 0.f;
                                for (int d0 = 1; d0 <= TRP; d0 += 8) {
                                    float lk[8];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const int d = d0 + u;
                                        const int row = BETA ? (j + c) : (j + c - d);
                                        const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + c + d < Lb);
                                        const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                        lk[u] = ok ? raw * H2_LOG2E : NEG_INF;
                                    }
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const int d = d0 + u;
                                        const float v = Abuf[prv * RL + liidx(c, d)] + lk[u];
                                        const float nm = fmaxf(mx, v);
                                        if (nm != NEG_INF) sum = sum * __builtin_amdgcn_exp2f(mx - nm) + __builtin_amdgcn_exp2f(v - nm);
                                        mx = nm;
                                    }
                                }
                                if (mx != NEG_INF) r = __builtin_amdgcn_logf(sum) + mx + (c == 0 ? m2[0] : m2[1]) * H2_LOG2E;
                            }
                        }
                        if (c == 0) a2[0] = r; else a2[1] = r;
                    }
                }
            }
            // ---- write the row: LDS state for the next row, HBM output.  Group exponent X = ceil(largest of the PAIR's four) - 120.
            float vn[2]; int xn;
            {
                const float amax = h2_pair_max(fmaxf(a2[0], a2[1]));
                const bool dead = amax == NEG_INF;
                const float cf = dead ? 0.f : ceilf(amax) - H2_BIAS;
                vn[0] = __builtin_amdgcn_exp2f(a2[0] - cf); vn[1] = __builtin_amdgcn_exp2f(a2[1] - cf);
                xn = dead ? H2_NEG : (int)cf;
            }
            *reinterpret_cast<float2*>(Vbuf + cur * RL + own_li0 + 2 * l) = make_float2(vn[0], vn[1]);
            if (!par) Xbuf[cur * GL + (own_li0 >> 2) + (l >> 1)] = xn;
            *reinterpret_cast<float2*>(Abuf + cur * RL + own_li0 + 2 * l) = make_float2(a2[0], a2[1]);
            if (col_ok) {
                if (j + 1 < L) { O[(size_t)t * LDO + j] = a2[0] * H2_LN2; O[(size_t)t * LDO + j + 1] = a2[1] * H2_LN2; }
                else O[(size_t)t * LDO + j] = a2[0] * H2_LN2;
                if (j + 2 >= L) for (int c = L; c < LPAD; ++c) O[(size_t)t * LDO + c] = NEG_INF;       // the owner of the last column fills the pitch padding
            }
            h2_barrier();
        }
        // rows the recurrence never reaches
        if (col_ok) for (int t = Tb; t < T; ++t) {
            O[(size_t)t * LDO + j] = NEG_INF;
            if (j + 1 < L) O[(size_t)t * LDO + j + 1] = NEG_INF;
            if (j + 2 >= L) for (int c = L; c < LPAD; ++c) O[(size_t)t * LDO + c] = NEG_INF;
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA, 4 bytes per lane: rows
        // of a dense tensor are not 16-byte aligned in general)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * p.ldm;
            float* slot = Mring + (size_t)(itr % H2_RING) * W;
#pragma unroll
            for (int i = 0; i < W / 64; ++i) {
                const int col = j0 + i * 64 + lane;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 64), 4, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < H2_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        h2_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + H2_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                asm volatile("s_waitcnt vmcnt(48)" ::: "memory");      // rows it+2 .. it+7 may stay in flight: 6 x 8 DMAs younger than row it+1's
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            h2_barrier();
        }
        (void)DPR;
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: neighbour strip's 64 boundary values -> LDS
        u64 g[H2_CH];
#pragma unroll
        for (int k = 0; k < H2_CH; ++k) g[k] = 0;
        auto load_row = [&](int itr) -> u64 {
            if (itr < nrows) { const int t = BETA ? (Tb - 1 - itr) : itr; return h2_gran_load(hin + (size_t)t * TRP + lane); }
            return 0;
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < H2_CH; ++k) g[k] = load_row(k);
        }
        __syncthreads();                         // link tile consumed
        h2_barrier();                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += H2_CH) {
#pragma unroll
            for (int k = 0; k < H2_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int t = BETA ? (Tb - 1 - it) : it;
                const int cur = it & 1;
                float hv = NEG_INF;
                if (has_producer) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    u64 x = g[k];
                    u32 spins = 0;
                    while (!__all((u32)(x >> 32) == want)) {
                        if ((u32)(x >> 32) != want) x = h2_gran_load(hin + (size_t)t * TRP + lane);
                        if (++spins > H2_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv = __uint_as_float((u32)x);
                }
                {
                    // the halo's sixteen groups: exponent = ceil(max of 4) by two quad-permute steps
                    float gm = fmaxf(hv, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, hv), 0xB1, 0xF, 0xF, false)));
                    gm = fmaxf(gm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gm), 0x4E, 0xF, 0xF, false)));
                    const bool dead = gm == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(gm) - H2_BIAS;
                    const float v = __builtin_amdgcn_exp2f(hv - cf);
                    Abuf[cur * RL + halo_li0 + lane] = hv;
                    Vbuf[cur * RL + halo_li0 + lane] = v;
                    if ((lane & 3) == 0) Xbuf[cur * GL + (halo_li0 >> 2) + (lane >> 2)] = dead ? H2_NEG : (int)cf;
                }
                if (has_producer) g[k] = load_row(it + H2_CH);
                h2_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: 64 boundary columns -> granules
        const bool pl = has_consumer;
        const int pub_li0 = BETA ? 0 : (TRP + W - TRP);       // alpha: the strip's last 64 columns (li W .. W+63); beta: its first 64 (li 0 .. 63)
        __syncthreads();                         // link tile consumed
        h2_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete (barrier it-1 passed); compute now writes the other buffer
                const int tp = BETA ? (Tb - it) : (it - 1);
                const float v = Abuf[((it - 1) & 1) * RL + pub_li0 + lane];
                h2_gran_store(hout + (size_t)tp * TRP + lane, p.tag_base + 1u + (u32)tp, v);
            }
            h2_barrier();
        }
        if (pl && nrows > 0) {
            const int it = nrows;
            const int tp = BETA ? (Tb - it) : (it - 1);
            const float v = Abuf[((it - 1) & 1) * RL + pub_li0 + lane];
            h2_gran_store(hout + (size_t)tp * TRP + lane, p.tag_base + 1u + (u32)tp, v);
        }
    }
}

__global__ __launch_bounds__(H2_NT + 192) void dag_strip2g_kernel(H2Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
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
    const int j0 = s * H2_W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {                    // nothing reachable in this strip: -inf everywhere, no hand-off
        float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * p.ldo;
        const int lpad = min(p.ldo, (L + 3) & ~3);
        for (int jj = j0 + tid; jj < j0 + H2_W && jj < lpad; jj += H2_NT + 192)
            for (int t = 0; t < T; ++t) O[(size_t)t * p.ldo + jj] = NEG_INF;
        return;
    }
    __syncthreads();                             // everyone has read the ticket before the tile overlays it
    if (is_beta) strip2g_body<true>(p, smem_raw + 16, b, s, dirslot, so);
    else strip2g_body<false>(p, smem_raw + 16, b, s, dirslot, so);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip2g_supported(int L, int TR) { return TR > 32 && TR <= H2_TRP && L >= 1; }

int launch_dag_strip2g(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                       float* alpha, float* beta, int B, int T, int L, int TR, int ldm, int ldo, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    const int NS = (L + H2_W - 1) / H2_W;
    H2Params p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.ldm = ldm; p.ldo = ldo;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * H2_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const size_t lds_main = (size_t)(4 * H2_RL + 2 * H2_GL + H2_RING * H2_W) * 4 + 16;
    const size_t lds_tile = (size_t)(H2_W + H2_TRP) * 65 * 4 + 16;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    set_max_dynamic_lds((const void*)dag_strip2g_kernel, (int)lds);
    hipLaunchKernelGGL(dag_strip2g_kernel, dim3((unsigned)(ndir * B * NS)), dim3(H2_NT + 192), lds, st, p);
    return check_launch("dag_loss_fwd(strip2g)");
}

}  // namespace dsp
