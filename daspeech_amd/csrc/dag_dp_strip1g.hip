// dag_dp_strip1g.hip — banded DAG DP for windows 65 .. 128 in EXP SPACE: K2 alpha || K3 beta, ONE vertex per lane (r06).
//
// Third member of the strip family (dag_dp_strip4g.hip: 4 vertices x 32 transitions per lane, dag_dp_strip2g.hip: 2 x 64): the same 128
// transition weights in a lane's registers, here for one vertex with a 128-wide window.  Until r06 these windows fell onto the dense-window
// matrix-core DP, which is built for windows of thousands of vertices and is at its worst on narrow bands (B = 32, T = 64, L = 4096: 5.1 ms at
// TR = 65 against 0.19 ms for TR = 64 on the strips — a 27 x step at a dispatch boundary).
//   * column strips of 256 vertices, one workgroup (4 compute waves + loader / fetch / publish helpers) per (sample, direction, strip), tagged
//     granules for the 128 boundary columns (two per helper lane), tickets;
//   * previous row in LDS as values 2^(a2 - X) with one integer exponent per group of 4 vertices = a QUAD of lanes (two quad-permutes);
//   * a lane's window is the 132 values from the 16-byte boundary under its first predecessor: 33 ds_read_b128, too many to hold at once next to
//     the weights, so they stream through two 6-read register buffers — chunk c+1 is requested before chunk c is consumed; the 33 group
//     exponents are read first (the row's reference exponent needs all of them);
//   * the transition tile of the prologue is loaded in two halves of 64 slots (a 384 x 129 tile would not fit the LDS);
//   * guard and fallbacks as strip2g: sums under 2^-97 (2^30 for a column with a flushed weight) take the diagonal's single-transition
//     shortcut or the exact log-space form; windows under 125 drop the groups without a predecessor from the row's reference.
// Replaces calculate_alpha_kernel / calculate_beta_kernel (dag_loss.cu:40-140,178-274) for 64 < translen <= 128.
#include "common.h"
#include <stdlib.h>

// the row's LDS issue groups (generated: operand numbers and byte offsets)
#define HEAD_ASM "ds_read_b32 %0, %24\n\t" \
                                 "ds_read2_b32 %1, %25 offset0:0 offset1:1\n\t" \
                                 "ds_read2_b32 %2, %25 offset0:2 offset1:3\n\t" \
                                 "ds_read2_b32 %3, %25 offset0:4 offset1:5\n\t" \
                                 "ds_read2_b32 %4, %25 offset0:6 offset1:7\n\t" \
                                 "ds_read2_b32 %5, %25 offset0:8 offset1:9\n\t" \
                                 "ds_read2_b32 %6, %25 offset0:10 offset1:11\n\t" \
                                 "ds_read2_b32 %7, %25 offset0:12 offset1:13\n\t" \
                                 "ds_read2_b32 %8, %25 offset0:14 offset1:15\n\t" \
                                 "ds_read2_b32 %9, %25 offset0:16 offset1:17\n\t" \
                                 "ds_read2_b32 %10, %25 offset0:18 offset1:19\n\t" \
                                 "ds_read2_b32 %11, %25 offset0:20 offset1:21\n\t" \
                                 "ds_read2_b32 %12, %25 offset0:22 offset1:23\n\t" \
                                 "ds_read2_b32 %13, %25 offset0:24 offset1:25\n\t" \
                                 "ds_read2_b32 %14, %25 offset0:26 offset1:27\n\t" \
                                 "ds_read2_b32 %15, %25 offset0:28 offset1:29\n\t" \
                                 "ds_read2_b32 %16, %25 offset0:30 offset1:31\n\t" \
                                 "ds_read_b32 %17, %25 offset:128\n\t" \
                                 "ds_read_b128 %18, %26\n\t" \
                                 "ds_read_b128 %19, %26 offset:16\n\t" \
                                 "ds_read_b128 %20, %26 offset:32\n\t" \
                                 "ds_read_b128 %21, %26 offset:48\n\t" \
                                 "ds_read_b128 %22, %26 offset:64\n\t" \
                                 "ds_read_b128 %23, %26 offset:80"
#define CHUNK6_1 "ds_read_b128 %0, %6 offset:96\n\t" \
                                 "ds_read_b128 %1, %6 offset:112\n\t" \
                                 "ds_read_b128 %2, %6 offset:128\n\t" \
                                 "ds_read_b128 %3, %6 offset:144\n\t" \
                                 "ds_read_b128 %4, %6 offset:160\n\t" \
                                 "ds_read_b128 %5, %6 offset:176"
#define CHUNK6_2 "ds_read_b128 %0, %6 offset:192\n\t" \
                                 "ds_read_b128 %1, %6 offset:208\n\t" \
                                 "ds_read_b128 %2, %6 offset:224\n\t" \
                                 "ds_read_b128 %3, %6 offset:240\n\t" \
                                 "ds_read_b128 %4, %6 offset:256\n\t" \
                                 "ds_read_b128 %5, %6 offset:272"
#define CHUNK6_3 "ds_read_b128 %0, %6 offset:288\n\t" \
                                 "ds_read_b128 %1, %6 offset:304\n\t" \
                                 "ds_read_b128 %2, %6 offset:320\n\t" \
                                 "ds_read_b128 %3, %6 offset:336\n\t" \
                                 "ds_read_b128 %4, %6 offset:352\n\t" \
                                 "ds_read_b128 %5, %6 offset:368"
#define CHUNK6_4 "ds_read_b128 %0, %6 offset:384\n\t" \
                                 "ds_read_b128 %1, %6 offset:400\n\t" \
                                 "ds_read_b128 %2, %6 offset:416\n\t" \
                                 "ds_read_b128 %3, %6 offset:432\n\t" \
                                 "ds_read_b128 %4, %6 offset:448\n\t" \
                                 "ds_read_b128 %5, %6 offset:464"
#define CHUNK3_5 "ds_read_b128 %0, %3 offset:480\n\t" \
                                 "ds_read_b128 %1, %3 offset:496\n\t" \
                                 "ds_read_b128 %2, %3 offset:512"

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float h1_v2f __attribute__((ext_vector_type(2)));
typedef float h1_v4f __attribute__((ext_vector_type(4)));
typedef int h1_v2i __attribute__((ext_vector_type(2)));

struct H1Params {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word, counters[2] = exact-path cells
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int ldm, ldo;                             // row pitches (elements) of match and of alpha / beta (>= L; pad columns L .. round4(L)-1 of alpha / beta get -inf)
};

constexpr int H1_NT = 256;                    // compute lanes = columns per strip
constexpr int H1_W = H1_NT;
constexpr int H1_TRP = 128;                   // window / halo width
constexpr int H1_RL = H1_W + H1_TRP;          // 384
constexpr int H1_GL = H1_RL / 4;              // 96
constexpr int H1_RING = 8;
constexpr int H1_CH = 4;
constexpr int H1_NEG = -(1 << 30);
constexpr u32 H1_SPIN_LIMIT = 1u << 22;
constexpr float H1_LOG2E = 1.4426950408889634f;
constexpr float H1_LN2 = 0.6931471805599453f;
constexpr float H1_BIAS = 120.f;

__device__ __forceinline__ u64 h1_gran_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void h1_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void h1_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ float h1_pair_max(float v) {          // lanes 2m, 2m+1
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));
}
__device__ __forceinline__ float h1_quad_max(float v) {          // lanes 4m .. 4m+3
    v = h1_pair_max(v);
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));
}

template <bool BETA>
__device__ __forceinline__ void strip1g_body(const H1Params& p, char* smem_raw, int b, int s, int dirslot, int so)
{
    constexpr int W = H1_W, RL = H1_RL, GL = H1_GL, NCW = H1_NT / 64, TRP = H1_TRP;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  exact row, log2 domain
    float* Vbuf = Abuf + 2 * RL;                               // [2][RL]  V = 2^(a2 - X[group])
    int* Xbuf = reinterpret_cast<int*>(Vbuf + 2 * RL);         // [2][GL]  group exponents
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
    // LDS geometry: alpha li = col - j0 + 128 (halo [0,128)); beta li = col - j0 (halo [W, W+128))
    const int halo_li0 = BETA ? W : 0;
    const int own_li0 = BETA ? 0 : TRP;

    // ---- prologue: transitions -> registers through an LDS tile, in two halves of 64 slots.  tile[r][dd] = links[rlo + r][64 h + dd]
    // (pitch 65), -inf outside the graph / beyond TR.  The tile overlays the main-loop buffers, which are not live yet.
    const int l = tid;                           // compute lanes: the lane's vertex
    const int par = l & 3;                       // position inside the 4-vertex group: the window starts `par` elements before the lane's own
    const int j = j0 + l;
    h1_v2f E2[66];                               // E2[i] = (weight of window element 2i, of 2i+1); 0 where the element is no predecessor
#pragma unroll
    for (int i = 0; i < 66; ++i) { E2[i].x = NEG_INF; E2[i].y = NEG_INF; }
    for (int h = 0; h < 2; ++h) {
        float* tile = reinterpret_cast<float*>(smem_raw);
        {
            constexpr int NTHR = H1_NT + 192, RPP = NTHR / 64;     // 7 rows per pass
            const int rlo = BETA ? j0 : (j0 - TRP);
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
            for (int q = 0; q < 132; ++q) {
                const int d = BETA ? (q - par) : (TRP + par - q);              // runtime (lane's position in its group): computed tile address
                if (d >= 64 * h + 1 && d <= 64 * h + 64) {
                    float v = BETA ? tile[l * 65 + (d - 1 - 64 * h)] : tile[(l - d + TRP) * 65 + (d - 1 - 64 * h)];
                    if (BETA && j + d >= Lb) v = NEG_INF;
                    if (q & 1) E2[q >> 1].y = v * H1_LOG2E; else E2[q >> 1].x = v * H1_LOG2E;
                }
            }
        }
        __syncthreads();                         // tile consumed: the next half (or the main-loop buffers) may overwrite it
    }

    if (wave < NCW) {
        // =========================================================== compute waves
        __builtin_amdgcn_s_setprio(2);
        const bool col_ok = j < L;
        auto cell_active = [&](int col, int t) -> bool {
            if (!BETA) return col >= t && col < Lb && (long)col <= (long)t * TR;
            const int rem = Tb - 1 - t, gap = Lb - 1 - col;
            return col >= t && gap >= rem && (long)gap <= (long)rem * TR;
        };
        auto liidx = [&](int d) -> int { return BETA ? (l + d) : (TRP + l - d); };          // LDS row index of the element at distance d
        float lmax, sthr;
        {
            float mx = NEG_INF;
#pragma unroll
            for (int i = 0; i < 66; ++i) mx = fmaxf(mx, fmaxf(E2[i].x, E2[i].y));
            if (mx == NEG_INF) mx = 0.f;
            lmax = mx;
            bool flushed = false;
#pragma unroll
            for (int i = 0; i < 66; ++i) {
                flushed |= ((E2[i].x != NEG_INF) & (E2[i].x - mx < -120.f)) | ((E2[i].y != NEG_INF) & (E2[i].y - mx < -120.f));
                E2[i].x = __builtin_amdgcn_exp2f(E2[i].x - mx); E2[i].y = __builtin_amdgcn_exp2f(E2[i].y - mx);
            }
            sthr = flushed ? 0x1p30f : 0x1p-97f;
        }
        // weight of distance 1 (the diagonal shortcut): window element 127 + par (alpha) / 1 + par (beta), four static candidates
        auto Eval1 = [&]() -> float {
            const int q0 = BETA ? 1 : 127;
            const float c0 = (q0 & 1) ? E2[q0 >> 1].y : E2[q0 >> 1].x, c1 = ((q0 + 1) & 1) ? E2[(q0 + 1) >> 1].y : E2[(q0 + 1) >> 1].x;
            const float c2 = ((q0 + 2) & 1) ? E2[(q0 + 2) >> 1].y : E2[(q0 + 2) >> 1].x, c3 = ((q0 + 3) & 1) ? E2[(q0 + 3) >> 1].y : E2[(q0 + 3) >> 1].x;
            return par == 0 ? c0 : (par == 1 ? c1 : (par == 2 ? c2 : c3));
        };
        // first (alpha) / last (beta) window group that holds a predecessor: window element of distance TR is 128 + par - TR resp. par + TR
        const int gcut = BETA ? ((par + TR) >> 2) : ((TRP + par - TR) >> 2);
        h1_barrier();                            // prologue barrier: match row 0 is in the ring

        for (int it = 0; it < nrows; ++it) {
            const int t = BETA ? (Tb - 1 - it) : it;
            const int cur = it & 1, prv = cur ^ 1;
            float a2 = NEG_INF;
            if (it == 0) {
                const float m0 = Mring[(size_t)(it % H1_RING) * W + l];
                const bool seed = BETA ? (j == Lb - 1) : (j == 0);
                if (seed) a2 = m0 * H1_LOG2E;
            } else {
                // ---- row head: match, 33 group exponents, the first 6 window groups — one issue group; the other 27 groups stream through
                // two register buffers (chunk c+1 requested before chunk c is consumed).  LDS returns in order; lgkmcnt counts to 15.
                float mt; h1_v2i xa[16]; int x32; h1_v4f pa[6], pb[6];
                const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Vbuf + prv * RL + 4 * (l >> 2));
                {
                    const u32 maddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Mring + (size_t)(it % H1_RING) * W + l);
                    const u32 xaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Xbuf + prv * GL + (l >> 2));
                    asm volatile(HEAD_ASM
                                 : "=&v"(mt), "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xa[2]), "=&v"(xa[3]), "=&v"(xa[4]), "=&v"(xa[5]), "=&v"(xa[6]), "=&v"(xa[7]),
                                   "=&v"(xa[8]), "=&v"(xa[9]), "=&v"(xa[10]), "=&v"(xa[11]), "=&v"(xa[12]), "=&v"(xa[13]), "=&v"(xa[14]), "=&v"(xa[15]), "=&v"(x32),
                                   "=&v"(pa[0]), "=&v"(pa[1]), "=&v"(pa[2]), "=&v"(pa[3]), "=&v"(pa[4]), "=&v"(pa[5])
                                 : "v"(maddr), "v"(xaddr), "v"(vaddr)
                                 : "memory");
                }
                // match + the 17 exponent reads are the first 18 of 24 operations: done when at most 6 are outstanding
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(mt), "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]),
                             "+v"(xa[8]), "+v"(xa[9]), "+v"(xa[10]), "+v"(xa[11]), "+v"(xa[12]), "+v"(xa[13]), "+v"(xa[14]), "+v"(xa[15]), "+v"(x32));
                const bool okc = cell_active(j, t);
                const float base = lmax + mt * H1_LOG2E;
                int xw[33];
#pragma unroll
                for (int g = 0; g < 16; ++g) { xw[2 * g] = xa[g].x; xw[2 * g + 1] = xa[g].y; }
                xw[32] = x32;
                if (TR < 125) {                  // groups without a predecessor: out of the window (see strip2g)
#pragma unroll
                    for (int g = 0; g < 33; ++g) if (BETA ? (g > gcut) : (g < gcut)) xw[g] = H1_NEG;
                }
                int refi = xw[0];
#pragma unroll
                for (int g = 1; g < 33; ++g) refi = max(refi, xw[g]);
                const bool any_live = refi != H1_NEG;
                if (!any_live) refi = 0;
                h1_v2f S2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { S2[c].x = 0.f; S2[c].y = 0.f; }
#define H1_ISSUE6(buf, c) asm volatile(CHUNK6_##c : "=&v"(buf[0]), "=&v"(buf[1]), "=&v"(buf[2]), "=&v"(buf[3]), "=&v"(buf[4]), "=&v"(buf[5]) : "v"(vaddr) : "memory");
#define H1_GROUP(buf, k, g, n) \
                { asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(buf[k])); \
                  const int kg = xw[g] - refi; \
                  h1_v2f wa, wb; wa.x = ldexpf(buf[k].x, kg); wa.y = ldexpf(buf[k].y, kg); wb.x = ldexpf(buf[k].z, kg); wb.y = ldexpf(buf[k].w, kg); \
                  S2[(2 * (g)) & 3] = __builtin_elementwise_fma(wa, E2[2 * (g)], S2[(2 * (g)) & 3]); \
                  S2[(2 * (g) + 1) & 3] = __builtin_elementwise_fma(wb, E2[2 * (g) + 1], S2[(2 * (g) + 1) & 3]); }
                // chunk 0 (groups 0..5) is in pa; request chunk 1 into pb, consume pa: element k of pa has landed when at most (5 - k) + 6 are outstanding
                H1_ISSUE6(pb, 1)
                H1_GROUP(pa, 0, 0, 11) H1_GROUP(pa, 1, 1, 10) H1_GROUP(pa, 2, 2, 9) H1_GROUP(pa, 3, 3, 8) H1_GROUP(pa, 4, 4, 7) H1_GROUP(pa, 5, 5, 6)
                H1_ISSUE6(pa, 2)
                H1_GROUP(pb, 0, 6, 11) H1_GROUP(pb, 1, 7, 10) H1_GROUP(pb, 2, 8, 9) H1_GROUP(pb, 3, 9, 8) H1_GROUP(pb, 4, 10, 7) H1_GROUP(pb, 5, 11, 6)
                H1_ISSUE6(pb, 3)
                H1_GROUP(pa, 0, 12, 11) H1_GROUP(pa, 1, 13, 10) H1_GROUP(pa, 2, 14, 9) H1_GROUP(pa, 3, 15, 8) H1_GROUP(pa, 4, 16, 7) H1_GROUP(pa, 5, 17, 6)
                H1_ISSUE6(pa, 4)
                H1_GROUP(pb, 0, 18, 11) H1_GROUP(pb, 1, 19, 10) H1_GROUP(pb, 2, 20, 9) H1_GROUP(pb, 3, 21, 8) H1_GROUP(pb, 4, 22, 7) H1_GROUP(pb, 5, 23, 6)
                asm volatile(CHUNK3_5 : "=&v"(pb[0]), "=&v"(pb[1]), "=&v"(pb[2]) : "v"(vaddr) : "memory");        // groups 30 .. 32
                H1_GROUP(pa, 0, 24, 8) H1_GROUP(pa, 1, 25, 7) H1_GROUP(pa, 2, 26, 6) H1_GROUP(pa, 3, 27, 5) H1_GROUP(pa, 4, 28, 4) H1_GROUP(pa, 5, 29, 3)
                H1_GROUP(pb, 0, 30, 2) H1_GROUP(pb, 1, 31, 1) H1_GROUP(pb, 2, 32, 0)
#undef H1_GROUP
#undef H1_ISSUE6
                const h1_v2f t2 = (S2[0] + S2[1]) + (S2[2] + S2[3]);
                const float S = t2.x + t2.y;
                const float ref = (float)refi;
                // ---- row tail
                const bool okl = okc & any_live;
                const bool flag = okl & !(S >= sthr && S <= 0x1p126f);
                a2 = (okl & !flag) ? (__builtin_amdgcn_logf(S) + (ref + base)) : NEG_INF;
                if (__builtin_expect(flag, 0)) {
                    float r = NEG_INF;
                    // (0) the DP's diagonal cell has ONE live transition: a2 = a2_prev(predecessor) + log2(weight) + base, no sum
                    const int dl = BETA ? (Lb - Tb + 1 + t - j) : (j - t + 1);
                    bool done = false;
                    if (dl == 1 && sthr == 0x1p-97f) {
                        const float ap = Abuf[prv * RL + liidx(1)];
                        const float e1 = Eval1();
                        if (e1 > 0.f) { r = (ap != NEG_INF) ? (ap + __builtin_amdgcn_logf(e1) + base) : NEG_INF; done = true; }
                    }
                    if (!done) {
                        // (b) exact log-space value: previous row from LDS, raw transitions re-read from HBM
                        float amax = NEG_INF;
                        for (int d = 1; d <= TRP; ++d) amax = fmaxf(amax, Abuf[prv * RL + liidx(d)]);
                        if (amax != NEG_INF) {
                            atomicAdd(&p.counters[2], 1u);
                            float mx = NEG_INF, sum = 0.f;
                            for (int d0 = 1; d0 <= TRP; d0 += 8) {
                                float lk[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const int row = BETA ? j : (j - d);
                                    const bool ok = d <= TR && row >= 0 && row < L && (!BETA || j + d < Lb);
                                    const float raw = K[(size_t)(ok ? row : 0) * TR + (ok ? d - 1 : 0)];
                                    lk[u] = ok ? raw * H1_LOG2E : NEG_INF;
                                }
#pragma unroll
                                for (int u = 0; u < 8; ++u) {
                                    const int d = d0 + u;
                                    const float v = Abuf[prv * RL + liidx(d)] + lk[u];
                                    const float nm = fmaxf(mx, v);
                                    if (nm != NEG_INF) sum = sum * __builtin_amdgcn_exp2f(mx - nm) + __builtin_amdgcn_exp2f(v - nm);
                                    mx = nm;
                                }
                            }
                            if (mx != NEG_INF) r = __builtin_amdgcn_logf(sum) + mx + mt * H1_LOG2E;
                        }
                    }
                    a2 = r;
                }
            }
            // ---- write the row: group exponent X = ceil(largest of the QUAD's four) - 120
            float vn; int xn;
            {
                const float amax = h1_quad_max(a2);
                const bool dead = amax == NEG_INF;
                const float cf = dead ? 0.f : ceilf(amax) - H1_BIAS;
                vn = __builtin_amdgcn_exp2f(a2 - cf);
                xn = dead ? H1_NEG : (int)cf;
            }
            Vbuf[cur * RL + own_li0 + l] = vn;
            if (par == 0) Xbuf[cur * GL + (own_li0 >> 2) + (l >> 2)] = xn;
            Abuf[cur * RL + own_li0 + l] = a2;
            if (col_ok) {
                O[(size_t)t * LDO + j] = a2 * H1_LN2;
                if (j + 1 == L) for (int c = L; c < LPAD; ++c) O[(size_t)t * LDO + c] = NEG_INF;         // the owner of the last column fills the pitch padding
            }
            h1_barrier();
        }
        if (col_ok) for (int t = Tb; t < T; ++t) {                                      // rows the recurrence never reaches
            O[(size_t)t * LDO + j] = NEG_INF;
            if (j + 1 == L) for (int c = L; c < LPAD; ++c) O[(size_t)t * LDO + c] = NEG_INF;
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA, 4 bytes per lane)
        auto issue_row = [&](int itr) {
            const int t = BETA ? (Tb - 1 - itr) : itr;
            const float* rowp = M + (size_t)t * p.ldm;
            float* slot = Mring + (size_t)(itr % H1_RING) * W;
#pragma unroll
            for (int i = 0; i < W / 64; ++i) {
                const int col = j0 + i * 64 + lane;
                const float* g = rowp + (col < L ? col : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 64), 4, 0, 0);
            }
        };
        for (int r = 0; r < H1_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        h1_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + H1_RING - 1;
            if (nx < nrows) {
                issue_row(nx);
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // rows it+2 .. it+7 may stay in flight: 6 x 4 DMAs younger than row it+1's
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            h1_barrier();
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: the neighbour strip's 128 boundary values -> LDS, two per lane
        u64 g0[H1_CH], g1[H1_CH];
#pragma unroll
        for (int k = 0; k < H1_CH; ++k) { g0[k] = 0; g1[k] = 0; }
        auto load_row = [&](int itr, u64& a, u64& c) {
            a = 0; c = 0;
            if (itr < nrows) { const int t = BETA ? (Tb - 1 - itr) : itr; a = h1_gran_load(hin + (size_t)t * TRP + 2 * lane); c = h1_gran_load(hin + (size_t)t * TRP + 2 * lane + 1); }
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < H1_CH; ++k) load_row(k, g0[k], g1[k]);
        }
        h1_barrier();                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += H1_CH) {
#pragma unroll
            for (int k = 0; k < H1_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int t = BETA ? (Tb - 1 - it) : it;
                const int cur = it & 1;
                float hv0 = NEG_INF, hv1 = NEG_INF;
                if (has_producer) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    u64 x = g0[k], y = g1[k];
                    u32 spins = 0;
                    while (!__all((u32)(x >> 32) == want && (u32)(y >> 32) == want)) {
                        if ((u32)(x >> 32) != want) x = h1_gran_load(hin + (size_t)t * TRP + 2 * lane);
                        if ((u32)(y >> 32) != want) y = h1_gran_load(hin + (size_t)t * TRP + 2 * lane + 1);
                        if (++spins > H1_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv0 = __uint_as_float((u32)x); hv1 = __uint_as_float((u32)y);
                }
                {
                    const float gm = h1_pair_max(fmaxf(hv0, hv1));        // the halo's 32 groups of 4 columns = pairs of lanes
                    const bool dead = gm == NEG_INF;
                    const float cf = dead ? 0.f : ceilf(gm) - H1_BIAS;
                    *reinterpret_cast<float2*>(Abuf + cur * RL + halo_li0 + 2 * lane) = make_float2(hv0, hv1);
                    *reinterpret_cast<float2*>(Vbuf + cur * RL + halo_li0 + 2 * lane) = make_float2(__builtin_amdgcn_exp2f(hv0 - cf), __builtin_amdgcn_exp2f(hv1 - cf));
                    if ((lane & 1) == 0) Xbuf[cur * GL + (halo_li0 >> 2) + (lane >> 1)] = dead ? H1_NEG : (int)cf;
                }
                if (has_producer) load_row(it + H1_CH, g0[k], g1[k]);
                h1_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: 128 boundary columns -> granules, two per lane
        const bool pl = has_consumer;
        const int pub_li0 = BETA ? 0 : W;        // alpha: the strip's last 128 columns (li W .. W+127); beta: its first 128 (li 0 .. 127)
        h1_barrier();                            // prologue barrier
        auto publish = [&](int itp) {
            const int tp = BETA ? (Tb - itp) : (itp - 1);
            const float2 v = *reinterpret_cast<const float2*>(Abuf + ((itp - 1) & 1) * RL + pub_li0 + 2 * lane);
            h1_gran_store(hout + (size_t)tp * TRP + 2 * lane, p.tag_base + 1u + (u32)tp, v.x);
            h1_gran_store(hout + (size_t)tp * TRP + 2 * lane + 1, p.tag_base + 1u + (u32)tp, v.y);
        };
        for (int it = 0; it < nrows; ++it) {
            if (it > 0 && pl) publish(it);       // row it-1 is complete (barrier it-1 passed); compute now writes the other buffer
            h1_barrier();
        }
        if (pl && nrows > 0) publish(nrows);
    }
}

__global__ __launch_bounds__(H1_NT + 192) void dag_strip1g_kernel(H1Params p)
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
    const int j0 = s * H1_W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {                    // nothing reachable in this strip: -inf everywhere, no hand-off
        float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * p.ldo;
        const int lpad = min(p.ldo, (L + 3) & ~3);
        for (int jj = j0 + tid; jj < j0 + H1_W && jj < lpad; jj += H1_NT + 192)
            for (int t = 0; t < T; ++t) O[(size_t)t * p.ldo + jj] = NEG_INF;
        return;
    }
    if (is_beta) strip1g_body<true>(p, smem_raw + 16, b, s, dirslot, so);
    else strip1g_body<false>(p, smem_raw + 16, b, s, dirslot, so);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool strip1g_supported(int L, int TR) { return TR > 64 && TR <= H1_TRP && L >= 1; }
size_t strip1g_ws_bytes(int B, int T, int L, int ndir) { return 256 + (size_t)ndir * B * ((L + H1_W - 1) / H1_W) * T * H1_TRP * sizeof(u64); }

int launch_dag_strip1g(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                       float* alpha, float* beta, int B, int T, int L, int TR, int ldm, int ldo, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    const int NS = (L + H1_W - 1) / H1_W;
    H1Params p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.ldm = ldm; p.ldo = ldo;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * H1_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    const size_t lds_main = (size_t)(4 * H1_RL + 2 * H1_GL + H1_RING * H1_W) * 4 + 16;
    const size_t lds_tile = (size_t)(H1_W + H1_TRP) * 65 * 4 + 16;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    set_max_dynamic_lds((const void*)dag_strip1g_kernel, (int)lds);
    hipLaunchKernelGGL(dag_strip1g_kernel, dim3((unsigned)(ndir * B * NS)), dim3(H1_NT + 192), lds, st, p);
    return check_launch("dag_loss_fwd(strip1g)");
}

}  // namespace dsp
