// extract_links_mfma.hip — the transition producer of extract_links.hip (DAGDecoder.extract_links, DASpeech/models/s2t_conformer_dag.py:171-212)
// on the matrix cores, for the released link predictor (8 heads x 64 channels), forward and backward, compact band layout.
//
// extract_links.hip forms every score q_i . k_j with fp32 FMAs, four source vertices per workgroup: 64 FMAs per (i, j, head), k rows re-read
// from L2 once per four vertices — VALU- and L2-bound from L ~ 1 000 on.  Here the scores are 32 x 32 blocks of  S^T = K . Q^T  on the fp16
// matrix cores with fp32 accuracy ("3 x fp16": x = xh + xl 2^-11, three MFMAs per product, see conv1d_split.hip / attention_split.hip), and the
// backward's contractions  dq = ds . K,  dk = ds^T . Q  run on the fp32 matrix-core path (v_mfma_f32_32x32x2_f32: exact fp32 products, no range
// assumption on the incoming gradient) or, on graphs above ~1 500 vertices, as bf16-TRIPLE products: ds and the partner rows are cut into three
// bf16 pieces each (8 + 8 + 8 mantissa bits by truncation: exact, and bf16 has fp32's exponent range — the gradient needs no scaling), the six
// piece products down to 2^-16 of the leading one run on the bf16 matrix cores into fp32 accumulators (24 MFMAs of 32 cycles per block
// instead of 32 of 64; ~2^-23 relative).  B = 32, L = 4096, TR = L-1 (BASELINE's graph with the README's --max-transition-length 99999), ms:
// inference 31.8 -> 3.0, forward + backward under autograd 43.2 -> 13.2; L = 1024: 1.82 -> 0.27, 2.8 -> 1.2; L = 400: 0.26 -> 0.08, 0.55 -> 0.34
// (tools/xl_mfma_time.py, profiles/r05_links_matrix_core.txt).
//
// Decomposition: a workgroup is 8 waves = the 8 heads of a tile of OWNER rows (32 or 64 source vertices i; for dk: successors j); a lane owns
// ONE owner row (the B operand: its split q row, pre-multiplied by scale log2(e), lives in registers), the PARTNER rows (successors j; for dk:
// sources i) stream by in tiles of 32 as MFMA A-fragments that a pre-pass wrote in fragment order (split once per call, O(L), instead of once
// per tile, O(L^2 / 32)).  The accumulator layout then hands a lane 16 of the 32 partners of ITS row: the window soft-max state, SA and the row
// sums are lane-local, the two lane halves are merged once at the end.  Everything between the MFMAs is base 2 (v_exp_f32 / v_log_f32 are):
// s2 = score log2(e) comes straight out of the accumulators, masked slots are -inf and drop out as exp2(-inf) = 0 without a select.  Five
// kernels share that loop:
//   STATS   per (vertex, head): window maximum and log-sum (online) — the soft-max state, = `stats` of dsp_extract_links_train
//   EMIT    links[i, d] = logsumexp_h(log_softmax + log_gate): the 8 waves park their head's term in an LDS image [head][owner][partner]
//           (64-owner tiles: double-buffered, one barrier per tile; graphs up to ~1 500 vertices: 32-owner tiles, four slots, one barrier per
//           two tiles); all 512 threads then reduce over the heads and store rows of the compact band
//   SA      dgate[i, h] = sum_d A[i, d, h],  A = G exp(ls + g - links)          (the tile of links / G is staged through LDS, coalesced)
//   DQ      ds = A - exp(ls) SA, contracted with the partner rows: dq^T[c, i] += K^T[c, j] ds^T[j, i]  (B operand = ds straight from the registers)
//   DK      the same with owners = successors j and partners = sources i (per-partner soft-max state from an LDS table)
// L2: a workgroup reads 64 KB of fragments per tile and nothing of it twice, so reuse has to come from its neighbours — with B % 8 == 0 an
// XCD keeps whole samples, its resident workgroups are neighbouring owner tiles of one sample, and they all walk the partner tiles from the
// same end (lock-step for a dense window): L2 hit rate 3 % -> 85-90 %, STATS 1.98 -> 1.41 ms at the C2 shape.
// Same arithmetic as extract_links.hip (same masks, same -inf conventions, same `stats` layout), so the two families are interchangeable
// per call (tests compare them element by element, and both with torch autograd / the fp64 oracle).
#include "common.h"
#include <atomic>
#include "../../include/daspeech_decode.h"
#include <type_traits>

namespace dsp {

constexpr int XM_H = 8, XM_CK = 64;
typedef _Float16 xm_h8 __attribute__((ext_vector_type(8)));
typedef float xm_f16 __attribute__((ext_vector_type(16)));
typedef __bf16 xm_b8 __attribute__((ext_vector_type(8)));
constexpr size_t XM_FRAG = 1024;                   // one MFMA operand fragment: 64 lanes x 16 bytes
constexpr size_t XM_TTILE = 12 * XM_FRAG;          // 32 rows of one head, TRANSPOSED, as three bf16 pieces: [2 channel blocks][2 steps][3 pieces]
constexpr size_t XM_TILE = 8 * XM_FRAG;            // 32 rows of one head: [4 channel steps][hi, lo]
constexpr int XM_PITCH = 36;                       // floats per owner row of an LDS tile (float4 slots of the 8 lanes of a store group on distinct banks)

enum { XM_STATS = 0, XM_EMIT = 1, XM_SA = 2, XM_DQ = 3, XM_DK = 4 };

struct XmParams {
    const float* q; const float* k; const float* gates; const int64_t* out_len; const float* bias;
    float* links; float* stats;                                          // forward outputs
    const float* clinks; const float* G; const float* cstats; float* dgate; float* dout;     // backward
    const char* pa;                                                      // the partner rows, split, in A-fragment order
    const char* pt;                                                      // the partner rows TRANSPOSED as bf16 triples (contraction operand), or NULL
    int B, L, TR, NT; float scale;
};

// RANGE (r05 ADVICE): the hi piece is an fp16 — an operand beyond +-65504 (k itself, or q * scale * log2(e)) has no split.  Such a value is
// CLAMPED to +-65 000 (the result stays finite, and is wrong for that element) and the launch raises g_xl_range, which
// dsp_extract_links_debug_range() returns and clears; the fp32-FMA kernels (xl_mfma 0) have no such limit.  Link-predictor inputs are
// projections of layer-normed features (|x| ~ 1-10): four orders of magnitude of head-room.
__device__ unsigned int g_xl_range;
__device__ __forceinline__ void xm_split8(const float* x, xm_h8& hi, xm_h8& lo) {
    bool over = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool big = fabsf(x[e]) > 65000.f;                                    // (false for NaN: a NaN operand stays a NaN)
        const float xc = big ? copysignf(65000.f, x[e]) : x[e];
        over |= big;
        hi[e] = (_Float16)xc; lo[e] = (_Float16)((xc - (float)hi[e]) * 2048.f);
    }
    if (__builtin_expect(over, 0)) atomicOr(&g_xl_range, 1u);
}

// pre-pass: x [B,L,8,64] fp32 -> A fragments [B][8 heads][NT tiles of 32 rows][4 steps][hi, lo][64 lanes][8 halves]; rows past L are zeros
// (blockIdx.z = 1: the second tensor of the backward's pair, x2 -> xa + second)
__global__ __launch_bounds__(256) void xl_mfma_split_kernel(const float* __restrict__ x, char* __restrict__ xa, int L, int NT,
                                                            const float* __restrict__ x2, size_t second)
{
    const int t = blockIdx.x, b = blockIdx.y;
    if (blockIdx.z) { x = x2; xa += second; }
    const size_t rs = XM_H * XM_CK;
    for (int it = threadIdx.x; it < 2048; it += 256) {
        const int g = it & 1, c = (it >> 1) & 3, h = (it >> 3) & 7, row = it >> 6;
        const int r = 32 * t + row;
        float v[8];
        if (r < L) {
            const float* src = x + ((size_t)b * L + r) * rs + h * XM_CK + 16 * c + 8 * g;
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 4);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        xm_h8 hi, lo; xm_split8(v, hi, lo);
        char* dst = xa + (((size_t)b * XM_H + h) * NT + t) * XM_TILE + (size_t)(c * 2) * XM_FRAG + (g * 32 + row) * 16;
        *reinterpret_cast<xm_h8*>(dst) = hi;
        *reinterpret_cast<xm_h8*>(dst + XM_FRAG) = lo;
    }
}

// pre-pass for the backward's contractions: x [B,L,8,64] fp32 -> x^T as three bf16 pieces (x = x1 + x2 + x3 exactly: 8 + 8 + 8 mantissa bits by
// truncation) in A-fragment order of out^T[channel][owner] += x^T[channel][partner] ds^T[partner][owner]: lane (col = channel, g), element e
// <-> partner (2 c2 + e / 4) * 8 + 4 g + e % 4 of the tile — the order in which the score accumulators hand a lane its partners.
__global__ __launch_bounds__(256) void xl_mfma_split_t_kernel(const float* __restrict__ x, char* __restrict__ xt, int L, int NT,
                                                              const float* __restrict__ x2, size_t second)
{
    const int t = blockIdx.x, b = blockIdx.y;
    if (blockIdx.z) { x = x2; xt += second; }
    const size_t rs = XM_H * XM_CK;
    for (int it = threadIdx.x; it < 2048; it += 256) {
        const int lane = it & 63, c2 = (it >> 6) & 1, db = (it >> 7) & 1, h = it >> 8;
        const int col = lane & 31, g = lane >> 5;
        unsigned w1[4], w2[4], w3[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned pc[3][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * w + q;
                const int r = 32 * t + (2 * c2 + (e >> 2)) * 8 + 4 * g + (e & 3);
                const float v = r < L ? x[((size_t)b * L + r) * rs + h * XM_CK + 32 * db + col] : 0.f;
                const unsigned b1 = __float_as_uint(v) & 0xFFFF0000u;
                const float r1 = v - __uint_as_float(b1);
                const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
                const float r2 = r1 - __uint_as_float(b2);
                pc[0][q] = b1; pc[1][q] = b2; pc[2][q] = __float_as_uint(r2) & 0xFFFF0000u;
            }
            w1[w] = pc[0][1] | (pc[0][0] >> 16); w2[w] = pc[1][1] | (pc[1][0] >> 16); w3[w] = pc[2][1] | (pc[2][0] >> 16);
        }
        char* dst = xt + (((size_t)b * XM_H + h) * NT + t) * XM_TTILE + (size_t)((db * 2 + c2) * 3) * XM_FRAG + lane * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        *reinterpret_cast<uint4*>(dst + XM_FRAG) = make_uint4(w2[0], w2[1], w2[2], w2[3]);
        *reinterpret_cast<uint4*>(dst + 2 * XM_FRAG) = make_uint4(w3[0], w3[1], w3[2], w3[3]);
    }
}

template <int MODE, int QG, int CT = 0>            // CT = 1: the contraction as bf16-triple products (6 MFMAs per 16 partners) instead of exact-fp32 MFMAs
__global__ __launch_bounds__(512) void xl_mfma_kernel(XmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char xm_smem[];
    constexpr bool TRANSPOSED = MODE == XM_DK;
    constexpr bool BWD = MODE >= XM_SA;
    constexpr bool CONTRACT = MODE == XM_DQ || MODE == XM_DK;
    constexpr int OT = 32 * QG;                                  // owner rows of the workgroup
    constexpr int NST = OT * 32 / 512;                           // staged elements per thread and tile (links and G each)
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6, col = lane & 31, g = lane >> 5;
    const int L = p.L, TR = p.TR;
    // workgroup -> (sample, owner tile).  Consecutive workgroup ids go round the 8 XCDs (one L2 each); with B % 8 == 0 every XCD keeps whole
    // samples to itself (sample = XCD + 8 n), its resident workgroups are neighbouring owner tiles of ONE sample, and they walk the partner tiles
    // in the same order from the same end (below) — one read of a partner tile from HBM / MALL then serves all of them out of that L2.
    const int NQ = (L + OT - 1) / OT;
    int b, x;
    if ((p.B & 7) == 0) { const int c = blockIdx.x & 7, sl = blockIdx.x >> 3; b = c + 8 * (sl / NQ); x = sl % NQ; }
    else { b = blockIdx.x / NQ; x = blockIdx.x % NQ; }
    // heavy tiles first: sources with a small index have the most successors, successors with a large index the most sources
    const int ot = TRANSPOSED ? (NQ - 1 - x) : x;
    const int o0 = ot * OT;
    const int Lb = min((int)p.out_len[b], L);
    const size_t rs = XM_H * XM_CK;
    const float* OWN = TRANSPOSED ? p.k : p.q;
    const float* PAR = TRANSPOSED ? p.q : p.k;

    // everything below lives in the log2 domain (v_exp_f32 / v_log_f32 are base 2): s2 = score log2(e)
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const float sc2 = p.scale * LOG2E;
    // ---- the lane's owner rows as B fragments (contraction index = channel, column = owner), hi / lo
    xm_h8 oh[QG][4], ol[QG][4];
#pragma unroll
    for (int grp = 0; grp < QG; ++grp) {
        const int o = min(o0 + 32 * grp + col, L - 1);
        const float* src = OWN + ((size_t)b * L + o) * rs + h * XM_CK + 8 * g;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + 16 * c);
            *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + 16 * c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= sc2;             // scores come out of the MFMAs as  q.k scale log2(e)
            xm_split8(v, oh[grp][c], ol[grp][c]);
        }
    }
    // ---- partner tiles: successors o0+1 .. o0+OT-1+TR inside the graph, or sources o0-TR .. o0+OT-2 (with a successor inside the graph)
    const int pbeg = TRANSPOSED ? max(0, o0 - TR) : (o0 + 1);
    const int pend = TRANSPOSED ? min(o0 + OT - 1, Lb) : min(Lb, o0 + OT + TR);
    const int t0 = pbeg >> 5;
    const int t1 = pend > pbeg ? ((pend + 31) >> 5) : t0;       // live tiles [t0, t1)
    const int te = MODE == XM_EMIT ? max(t1, (o0 + OT + TR + 31) >> 5) : t1;      // EMIT also covers the slots beyond the graph (-inf)

    // ---- per-owner state (the soft-max rows are the owners, except for DK): running (max, sum) for STATS; otherwise
    //      ca = (gate - max - logsum) log2(e)  (log-soft-max + gate = s2 + ca, base 2), cp = -(max + logsum) log2(e), sa = SA
    float mrow[QG], lrow[QG], ca[QG], cp[QG], sarow[QG];
#pragma unroll
    for (int grp = 0; grp < QG; ++grp) {
        mrow[grp] = NEG_INF; lrow[grp] = 0.f; ca[grp] = 0.f; cp[grp] = 0.f; sarow[grp] = 0.f;
        if constexpr (MODE == XM_EMIT || MODE == XM_SA || MODE == XM_DQ) {
            const int o = min(o0 + 32 * grp + col, L - 1);
            const size_t so = ((size_t)b * L + o) * XM_H + h;
            const float* st = (MODE == XM_EMIT) ? p.stats : p.cstats;
            const float mx = st[2 * so], ls = st[2 * so + 1], gt = p.gates[so];
            // rows without a successor (max = -inf) have nothing but masked slots (s2 = -inf): any finite constant keeps them at exp2(-inf) = 0
            ca[grp] = (mx == NEG_INF) ? 0.f : ((gt - mx) - ls) * LOG2E; cp[grp] = (mx == NEG_INF) ? 0.f : -(mx + ls) * LOG2E;
            if constexpr (MODE == XM_DQ) sarow[grp] = p.dgate[so];
        }
    }
    xm_f16 acc[CONTRACT ? QG : 1][2];
    if constexpr (CONTRACT) {
#pragma unroll
        for (int grp = 0; grp < QG; ++grp)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[grp][db][r] = 0.f;
    }

    // ---- LDS
    float* img = reinterpret_cast<float*>(xm_smem);                                     // EMIT: [2 or 4 slots][8 heads][OT][PITCH]
    float* stg = reinterpret_cast<float*>(xm_smem);                                     // BWD:  [4 slots][links, G][OT][PITCH]  (owner-major in both directions); two tiles are staged per barrier
    constexpr int STG_ONE = OT * XM_PITCH;
    float4* tab = reinterpret_cast<float4*>(xm_smem + 4 * 2 * STG_ONE * sizeof(float));    // DK: [4 slots][32 partners][8 heads] (ca, cp, SA, -)

    // buffer descriptors over this sample's blocks (32-bit offsets, hardware bounds check: rows past the sample read 0)
    const __amdgpu_buffer_rsrc_t r_par = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PAR + (size_t)b * L * rs), 0, (int)((size_t)L * rs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_lk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BWD ? p.clinks : p.q) + (BWD ? (size_t)b * L * TR : 0)), 0,
                                                                          BWD ? (int)((size_t)L * TR * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_gg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BWD ? p.G : p.q) + (BWD ? (size_t)b * L * TR : 0)), 0,
                                                                          BWD ? (int)((size_t)L * TR * 4) : 0, 0x00020000);
    const int xa_voff = (int)(((size_t)(4 * g) * rs + h * XM_CK + col) * 4);

    // staging registers (one tile ahead)
    float r_lkv2[2][BWD ? NST : 1], r_gv2[2][BWD ? NST : 1];
    float4 r_tab2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    auto stage_load = [&](int t, auto which_tag) {
        constexpr int W = decltype(which_tag)::value;
        float (&r_lkv)[BWD ? NST : 1] = r_lkv2[W]; float (&r_gv)[BWD ? NST : 1] = r_gv2[W]; float4& r_tab = r_tab2[W];
        if constexpr (BWD) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = it * 512 + tid;
                int i, j;
                if constexpr (TRANSPOSED) { i = 32 * t + idx / OT; j = o0 + idx % OT; }
                else { i = o0 + (idx >> 5); j = 32 * t + (idx & 31); }
                const int d = j - i - 1;
                const bool ok = i < L && d >= 0 && d < TR;
                const int off = ok ? (i * TR + d) * 4 : 0;
                const float lk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_lk, off, 0, 0));
                const float gv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_gg, off, 0, 0));
                // links log2(e); a -inf link (or a slot outside the band) becomes +3e38: exp2(s2 + ca - 3e38) = 0, its share A is 0 without a test
                r_lkv[it] = (ok && lk != NEG_INF) ? lk * LOG2E : 3.0e38f; r_gv[it] = ok ? gv : 0.f;
            }
            if constexpr (TRANSPOSED) {
                if (tid < 256) {
                    const int i = 32 * t + (tid >> 3), hh = tid & 7;
                    if (i < L) {
                        const size_t so = ((size_t)b * L + i) * XM_H + hh;
                        const float mx = p.cstats[2 * so], ls = p.cstats[2 * so + 1];
                        r_tab = (mx == NEG_INF) ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                : make_float4(((p.gates[so] - mx) - ls) * LOG2E, -(mx + ls) * LOG2E, p.dgate[so], 0.f);
                    } else r_tab = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };
    auto stage_store = [&](int buf, auto which_tag) {
        constexpr int W = decltype(which_tag)::value;
        const float (&r_lkv)[BWD ? NST : 1] = r_lkv2[W]; const float (&r_gv)[BWD ? NST : 1] = r_gv2[W]; const float4& r_tab = r_tab2[W];
        if constexpr (BWD) {
            float* s_lk = stg + (size_t)buf * 2 * STG_ONE; float* s_g = s_lk + STG_ONE;
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = it * 512 + tid;
                const int at = TRANSPOSED ? (idx % OT) * XM_PITCH + idx / OT : (idx >> 5) * XM_PITCH + (idx & 31);
                s_lk[at] = r_lkv[it]; s_g[at] = r_gv[it];
            }
            if constexpr (TRANSPOSED) { if (tid < 256) tab[buf * 256 + tid] = r_tab; }
        }
    };

    // ---- A fragments of a partner tile (this wave's head)
    const char* pa_h = p.pa + (((size_t)b * XM_H + h) * p.NT) * XM_TILE + lane * 16;
    const char* pt_h = CT == 1 ? p.pt + (((size_t)b * XM_H + h) * p.NT) * XM_TTILE + lane * 16 : nullptr;
    xm_h8 fa[8], fb[8];
    auto frag_load = [&](int t, xm_h8 (&f)[8]) {
        const char* src = pa_h + (size_t)t * XM_TILE;
#pragma unroll
        for (int u = 0; u < 8; ++u) f[u] = *reinterpret_cast<const xm_h8*>(src + u * XM_FRAG);
    };
    // step n -> tile: the successors are walked from the END of the window backwards (all owner tiles of a dense window start at the same
    // tile: lock-step), the sources of DK forwards from tile 0; EMIT's slots beyond the graph follow the live tiles
    const int nlive = t1 - t0, nstep = te - t0;
    auto tile_of = [&](int n) { return TRANSPOSED ? (t0 + n) : (n < nlive ? (t1 - 1 - n) : (t1 + (n - nlive))); };

    // one 32 x 32 block of one owner group: scores, then the mode's work.  FULL: every slot of the block is inside the band and the graph
    auto block = [&](auto full_tag, int t, int grp, int buf, const xm_h8 (&a)[8], const float (&xa)[(CONTRACT && CT == 0) ? 32 : 1]) {
        constexpr bool FULL = decltype(full_tag)::value;
        xm_f16 shh, slo;
#pragma unroll
        for (int r = 0; r < 16; ++r) { shh[r] = 0.f; slo[r] = 0.f; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            shh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * c], oh[grp][c], shh, 0, 0, 0);
            slo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * c + 1], oh[grp][c], slo, 0, 0, 0);
            slo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * c], ol[grp][c], slo, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int o = o0 + 32 * grp + col;
        float s[16];                                               // s2 = score log2(e); masked slots -inf
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fmaf(slo[r], 1.f / 2048.f, shh[r]);
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = 32 * t + 8 * (r >> 2) + 4 * g + (r & 3);
                const int d = TRANSPOSED ? (o - pp - 1) : (pp - o - 1);
                s[r] = fmaf(p.bias[min(max(d, 0), TR - 1)], LOG2E, s[r]);
            }
        }
        if constexpr (!FULL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = 32 * t + 8 * (r >> 2) + 4 * g + (r & 3);
                const int d = TRANSPOSED ? (o - pp - 1) : (pp - o - 1);
                const bool ok = d >= 0 && d < TR && (TRANSPOSED ? (o < Lb) : (pp < Lb && o < L));
                s[r] = ok ? s[r] : NEG_INF;
            }
        }
        if constexpr (MODE == XM_STATS) {
            float mx = NEG_INF;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            const float m_new = fmaxf(mrow[grp], mx);
            const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(s[r] - m_use);
            lrow[grp] = lrow[grp] * __builtin_amdgcn_exp2f(mrow[grp] - m_use) + ps;
            mrow[grp] = m_new;
        } else if constexpr (MODE == XM_EMIT) {
            float* dst = img + (((size_t)buf * XM_H + h) * OT + 32 * grp + col) * XM_PITCH + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j)            // -inf + ca = -inf: masked slots stay masked
                *reinterpret_cast<float4*>(dst + 8 * j) = make_float4(s[4 * j] + ca[grp], s[4 * j + 1] + ca[grp], s[4 * j + 2] + ca[grp], s[4 * j + 3] + ca[grp]);
        } else {
            // ---- A = G exp(ls + gate - links),  ds = A - exp(ls) SA   (masked slots: exp2(-inf) = 0 twice)
            const float* s_lk = stg + (size_t)buf * 2 * STG_ONE + (32 * grp + col) * XM_PITCH + 4 * g;
            const float* s_g = s_lk + STG_ONE;
            float ds[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 lk4 = *reinterpret_cast<const float4*>(s_lk + 8 * j);
                const float4 gv4 = *reinterpret_cast<const float4*>(s_g + 8 * j);
                const float lkv[4] = {lk4.x, lk4.y, lk4.z, lk4.w}, gvv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * j + e;
                    float cae, cpe, sae;
                    if constexpr (TRANSPOSED) { const float4 st = tab[buf * 256 + (8 * j + 4 * g + e) * XM_H + h]; cae = st.x; cpe = st.y; sae = st.z; }
                    else { cae = ca[grp]; cpe = cp[grp]; sae = sarow[grp]; }
                    const float A = gvv[e] * __builtin_amdgcn_exp2f((s[r] + cae) - lkv[e]);
                    if constexpr (MODE == XM_SA) ds[r] = A;
                    else ds[r] = fmaf(-__builtin_amdgcn_exp2f(s[r] + cpe), sae, A);
                }
            }
            if constexpr (MODE == XM_SA) {
                float sa = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sa += ds[r];
                sarow[grp] += sa;
            } else {
                if constexpr (CT == 1) {
                    // ds = d1 + d2 + d3 (bf16 pieces by truncation, exact), partner rows x = x1 + x2 + x3 from the pre-pass: the six products down to
                    // 2^-16 of the leading one (d1 x1, d1 x2, d2 x1, d1 x3, d2 x2, d3 x1) on the bf16 matrix cores, fp32 accumulators — fp32 range,
                    // ~2^-23 relative; 24 MFMAs of 32 cycles per block instead of 32 of 64
                    xm_b8 dp[3][2];
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        unsigned w1[4], w2[4], w3[4];
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            unsigned pc[3][2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const float v = ds[8 * c2 + 2 * w + q];
                                const unsigned b1 = __float_as_uint(v) & 0xFFFF0000u;
                                const float r1 = v - __uint_as_float(b1);
                                const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
                                const float r2 = r1 - __uint_as_float(b2);
                                pc[0][q] = __float_as_uint(v); pc[1][q] = __float_as_uint(r1); pc[2][q] = __float_as_uint(r2);
                            }
                            w1[w] = __builtin_amdgcn_perm(pc[0][1], pc[0][0], 0x07060302u);
                            w2[w] = __builtin_amdgcn_perm(pc[1][1], pc[1][0], 0x07060302u);
                            w3[w] = __builtin_amdgcn_perm(pc[2][1], pc[2][0], 0x07060302u);
                        }
                        dp[0][c2] = __builtin_bit_cast(xm_b8, make_uint4(w1[0], w1[1], w1[2], w1[3]));
                        dp[1][c2] = __builtin_bit_cast(xm_b8, make_uint4(w2[0], w2[1], w2[2], w2[3]));
                        dp[2][c2] = __builtin_bit_cast(xm_b8, make_uint4(w3[0], w3[1], w3[2], w3[3]));
                    }
                    const char* tsrc = pt_h + (size_t)t * XM_TTILE;
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        xm_b8 kt[2][3];
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) kt[c2][pc] = *reinterpret_cast<const xm_b8*>(tsrc + (size_t)((db * 2 + c2) * 3 + pc) * XM_FRAG);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) {
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][0], dp[0][c2], acc[grp][db], 0, 0, 0);
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][1], dp[0][c2], acc[grp][db], 0, 0, 0);
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][0], dp[1][c2], acc[grp][db], 0, 0, 0);
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][2], dp[0][c2], acc[grp][db], 0, 0, 0);
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][1], dp[1][c2], acc[grp][db], 0, 0, 0);
                            acc[grp][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[c2][0], dp[2][c2], acc[grp][db], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                // the 32 MFMAs leave back to back: a VALU instruction between two MFMAs on one accumulator costs tens of cycles each (the
                // other wave of the SIMD fills the matrix-core time with ITS VALU phase)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[grp][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * r], ds[r], acc[grp][0], 0, 0, 0);
                    acc[grp][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * r + 1], ds[r], acc[grp][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // EMIT: links[i, d] = logsumexp over the heads of one tile's image; lanes along the partners: a wave stores two rows of 128 contiguous bytes
    constexpr bool EMIT_PAIR = MODE == XM_EMIT && QG == 1;       // 32-owner tiles: 4 image slots, one barrier per pair of tiles; 64-owner tiles: 2 slots, one per tile
    auto combine = [&](int n) {
        if constexpr (MODE == XM_EMIT) {
            const int t = tile_of(n), buf = EMIT_PAIR ? (n & 3) : (n & 1);
            const bool live_tile = n < nlive;
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int idx = it * 512 + tid, qq = idx >> 5, kk = idx & 31;
                const int i = o0 + qq, d = 32 * t + kk - i - 1;
                if (i < L && d >= 0 && d < TR) {
                    float r = NEG_INF;
                    if (live_tile) {
                        float v[XM_H], m2 = NEG_INF;
#pragma unroll
                        for (int hh = 0; hh < XM_H; ++hh) { v[hh] = img[(((size_t)buf * XM_H + hh) * OT + qq) * XM_PITCH + kk]; m2 = fmaxf(m2, v[hh]); }
                        if (m2 != NEG_INF) {
                            float e = 0.f;
#pragma unroll
                            for (int hh = 0; hh < XM_H; ++hh) e += __builtin_amdgcn_exp2f(v[hh] - m2);
                            r = (m2 + __builtin_amdgcn_logf(e)) * LN2;
                        }
                    }
                    p.links[((size_t)b * L + i) * TR + d] = r;
                }
            }
        }
    };

    auto step = [&](int n, xm_h8 (&cur)[8], xm_h8 (&nxt)[8]) {
        const int t = tile_of(n);
        const int buf = (BWD || EMIT_PAIR) ? (n & 3) : (n & 1);
        const bool live_tile = n < nlive;
        if constexpr (BWD) {
            if ((n & 1) == 0) {                         // one barrier per PAIR of tiles: both tiles' links / G (and DK's partner table) go to LDS together
                if (n < nlive) stage_store(n & 3, std::integral_constant<int, 0>{});
                if (n + 1 < nlive) stage_store((n + 1) & 3, std::integral_constant<int, 1>{});
                __syncthreads();
                if (n + 2 < nlive) stage_load(tile_of(n + 2), std::integral_constant<int, 0>{});
                if (n + 3 < nlive) stage_load(tile_of(n + 3), std::integral_constant<int, 1>{});
            }
        }
        if (live_tile) {
            if (n + 1 < nlive) frag_load(tile_of(n + 1), nxt);
            float xa[(CONTRACT && CT == 0) ? 32 : 1];
            if constexpr (CONTRACT && CT == 0) {
                // the partner rows in fp32 for the contraction: lane (col = channel, g) <-> row 8 j' + 4 g + e of the tile
                const int soff = (int)((size_t)(32 * t) * rs * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = soff + (8 * (r >> 2) + (r & 3)) * (int)(rs * 4);
                    xa[2 * r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_par, xa_voff, ro, 0));
                    xa[2 * r + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_par, xa_voff, ro + 128, 0));
                }
            }
#pragma unroll
            for (int grp = 0; grp < QG; ++grp) {
                const int omin = o0 + 32 * grp, omax = omin + 31, pmin = 32 * t, pmax = pmin + 31;
                const int dmin = TRANSPOSED ? (omin - pmax - 1) : (pmin - omax - 1);
                const int dmax = TRANSPOSED ? (omax - pmin - 1) : (pmax - omin - 1);
                if (dmax < 0 || dmin >= TR) continue;                          // wave-uniform: the block is outside the band
                const bool full = dmin >= 0 && dmax < TR && (TRANSPOSED ? (omax < Lb) : (pmax < Lb && omax < L));
                if (full) block(std::true_type{}, t, grp, buf, cur, xa);
                else block(std::false_type{}, t, grp, buf, cur, xa);
            }
        }
        if constexpr (MODE == XM_EMIT) {
            // the heads of a tile (of a PAIR of tiles with 32-owner tiles) meet, then all 512 threads reduce them
            if constexpr (EMIT_PAIR) {
                if ((n & 1) || n + 1 >= nstep) {
                    __syncthreads();
                    if (n & 1) combine(n - 1);
                    combine(n);
                }
            } else {
                __syncthreads();
                combine(n);
            }
        }
    };
    if (nlive > 0) { frag_load(tile_of(0), fa); stage_load(tile_of(0), std::integral_constant<int, 0>{}); }
    if (nlive > 1) stage_load(tile_of(1), std::integral_constant<int, 1>{});
    for (int n = 0; n < nstep; n += 2) {
        step(n, fa, fb);
        if (n + 1 < nstep) step(n + 1, fb, fa);
    }

    // ---- epilogues
    if constexpr (MODE == XM_STATS) {
#pragma unroll
        for (int grp = 0; grp < QG; ++grp) {
            const float m1 = __shfl_xor(mrow[grp], 32, 64), l1 = __shfl_xor(lrow[grp], 32, 64);
            const float m = fmaxf(mrow[grp], m1), mu = (m == NEG_INF) ? 0.f : m;
            const float l = lrow[grp] * __builtin_amdgcn_exp2f(mrow[grp] - mu) + l1 * __builtin_amdgcn_exp2f(m1 - mu);
            const int o = o0 + 32 * grp + col;
            if (g == 0 && o < L) {
                float* st = p.stats + (((size_t)b * L + o) * XM_H + h) * 2;
                st[0] = m * LN2; st[1] = (m == NEG_INF) ? 0.f : __logf(l);     // back to natural logarithms: the `stats` of extract_links.hip
            }
        }
    } else if constexpr (MODE == XM_SA) {
#pragma unroll
        for (int grp = 0; grp < QG; ++grp) {
            const float sa = sarow[grp] + __shfl_xor(sarow[grp], 32, 64);
            const int o = o0 + 32 * grp + col;
            if (g == 0 && o < L) p.dgate[((size_t)b * L + o) * XM_H + h] = sa;
        }
    } else if constexpr (CONTRACT) {
        // out^T[channel][owner]: the lane's owner row, channels 32 db + 8 j + 4 g + 0..3
#pragma unroll
        for (int grp = 0; grp < QG; ++grp) {
            const int o = o0 + 32 * grp + col;
            if (o < L) {
                float* dst = p.dout + ((size_t)b * L + o) * rs + h * XM_CK + 4 * g;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *reinterpret_cast<float4*>(dst + 32 * db + 8 * j) =
                            make_float4(acc[grp][db][4 * j] * p.scale, acc[grp][db][4 * j + 1] * p.scale, acc[grp][db][4 * j + 2] * p.scale, acc[grp][db][4 * j + 3] * p.scale);
            }
        }
    }
}

template <int MODE, int QG, int CT = 0>
static int xm_launch(const XmParams& p, hipStream_t st, const char* what)
{
    constexpr int OT = 32 * QG;
    size_t lds = 16;
    if (MODE == XM_EMIT) lds = (size_t)(QG == 1 ? 4 : 2) * XM_H * OT * XM_PITCH * sizeof(float);
    else if (MODE >= XM_SA) lds = (size_t)4 * 2 * OT * XM_PITCH * sizeof(float) + (MODE == XM_DK ? 4 * 256 * sizeof(float4) : 0);
    auto k = xl_mfma_kernel<MODE, QG, CT>;
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(((p.L + OT - 1) / OT) * p.B)), dim3(512), lds, st, p);
    return check_launch(what);
}

extern std::atomic<unsigned int> g_xl_ran;       // extract_links.hip
static std::atomic<int> g_xl_mfma{-1};        // dsp_dag_set_option("xl_mfma", v): 1 = matrix-core kernels wherever they apply, 0 = never, -1 = by size
void set_xl_mfma(int v) { g_xl_mfma = v; }

// (32-bit byte offsets inside a sample's links / gradient block: L * TR * 4 < 2^31)
static bool xm_supported(int L, int H, int CK, int TR) { return H == XM_H && CK == XM_CK && TR >= 1 && L >= 2 && (long)L * TR < (1L << 29); }
static bool xm_preferred(int B, int L, int H, int CK, int TR)
{
    const int pin = g_xl_mfma.load();
    if (!xm_supported(L, H, CK, TR) || pin == 0) return false;
    if (pin > 0) return true;
    // by size (tools/xl_mfma_time.py): three launches and a split pre-pass need a band of ~half a million slots to pay — below that the one launch of
    // extract_links.hip wins on launch latency (ms, FMA -> matrix cores — B = 1, L = 800: 0.071 -> 0.071; B = 2, L = 400: 0.041 -> 0.046;
    // B = 1, L = 1200: 0.22 -> 0.10; B = 8, L = 400: 0.078 -> 0.049)
    return L >= 128 && TR >= 64 && (long)B * L * TR >= 500000;
}

static size_t xm_split_bytes(int B, int L) { return (size_t)B * XM_H * ((L + 31) / 32) * XM_TILE; }
static size_t xm_split_t_bytes(int B, int L) { return (size_t)B * XM_H * ((L + 31) / 32) * XM_TTILE; }
static std::atomic<int> g_xl_contract{-1};    // dsp_dag_set_option("xl_contract", v): 0 = exact-fp32 MFMA contraction, 1 = bf16-triple products, -1 = default
void set_xl_contract(int v) { g_xl_contract = v; }
// by size: the transposed pre-pass costs 23 us at B = 32, L = 400 for 15 us saved in DQ + DK; at L = 4096 0.23 ms for 0.9 ms (DQ 4.53 -> 4.03, DK 4.88 -> 4.50)
static bool xm_bf16_contraction(int L) { const int c = g_xl_contract.load(); return c > 0 || (c < 0 && L > 1536); }
static size_t xm_bwd_bytes(int B, int L) { return 2 * xm_split_bytes(B, L) + 2 * xm_split_t_bytes(B, L); }

}  // namespace dsp

extern "C" unsigned int dsp_extract_links_debug_range(void)
{
    unsigned int v = 0, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(dsp::g_xl_range), sizeof(v)) != hipSuccess) return 0xffffffffu;       // (synchronises the device: tests only)
    if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(dsp::g_xl_range), &z, sizeof(z));
    return v;
}

extern "C" int dsp_extract_links_workspace(int B, int L, int H, int CK, int TR, int training, size_t* bytes)
{
    using namespace dsp;
    if (!bytes) { set_error("extract_links_workspace: null pointer"); return DSP_EINVAL; }
    *bytes = 0;
    if (B <= 0 || !xm_preferred(B, L, H, CK, TR)) return DSP_OK;            // 0 bytes: the fp32-FMA kernels of extract_links.hip serve this call
    // forward: the split k rows + a stats scratch for the inference entry point; backward: the split k and q rows
    *bytes = training == 2 ? xm_bwd_bytes(B, L) : xm_split_bytes(B, L) + (size_t)B * L * XM_H * 2 * sizeof(float);
    return DSP_OK;
}

extern "C" int dsp_extract_links_ws(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                                    float* links, float* stats, int B, int L, int H, int CK, int TR, float scale,
                                    void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 1 || TR < 1) { set_error("extract_links_ws: bad sizes B=%d L=%d TR=%d", B, L, TR); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!xm_supported(L, H, CK, TR)) { set_error("extract_links_ws: H=%d CK=%d L=%d TR=%d is not served by the matrix-core kernels (8 heads x 64 channels only: call dsp_extract_links)", H, CK, L, TR); return DSP_EINVAL; }
    if (!q || !k || !log_gates || !out_len || !links || !workspace) { set_error("extract_links_ws: null pointer"); return DSP_EINVAL; }
    const size_t need = xm_split_bytes(B, L) + (size_t)B * L * XM_H * 2 * sizeof(float);
    if (workspace_bytes < need || ((uintptr_t)workspace & 15) || (((uintptr_t)q | (uintptr_t)k) & 15)) {
        set_error("extract_links_ws: workspace of %zu bytes (16-byte aligned) needed, %zu given; q / k 16-byte aligned", need, workspace_bytes); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    XmParams p{};
    p.q = q; p.k = k; p.gates = log_gates; p.out_len = out_len; p.bias = dist_bias; p.links = links;
    p.pa = static_cast<const char*>(workspace);
    p.stats = stats ? stats : reinterpret_cast<float*>(static_cast<char*>(workspace) + xm_split_bytes(B, L));
    p.B = B; p.L = L; p.TR = TR; p.NT = (L + 31) / 32; p.scale = scale;
    hipLaunchKernelGGL(xl_mfma_split_kernel, dim3((unsigned)p.NT, (unsigned)B), dim3(256), 0, st, k, static_cast<char*>(workspace), L, p.NT, (const float*)nullptr, (size_t)0);
    if (int rc = check_launch("extract_links(split)")) return rc;
    g_xl_ran |= 4u;
    static const char* const e_sq = getenv("DSP_XM_STATS_QG");
    const int sq = e_sq ? atoi(e_sq) : (L <= 1536 ? 1 : 2);          // as EMIT below (us at B = 32, 64- / 32-owner tiles — L = 256: 20 / 14, L = 1024: 121 / 93)
    if (int rc = sq == 1 ? xm_launch<XM_STATS, 1>(p, st, "extract_links(matrix-core soft-max state)")
                         : xm_launch<XM_STATS, 2>(p, st, "extract_links(matrix-core soft-max state)")) return rc;
    static const char* const e_qg = getenv("DSP_XM_EMIT_QG");
    const int qg = e_qg ? atoi(e_qg) : (L <= 1536 ? 1 : 2);
    // 32-owner tiles with one barrier per two tiles on graphs up to ~1 500 vertices (us at B = 32, 64- vs 32-owner tiles — L = 256: 28 / 18,
    // L = 400: 42 / 39, L = 1024: 174 / 153), 64-owner tiles above (half the fragment traffic — L = 2048: 552 / 585, L = 4096: 1 950 / 2 130)
    if (qg == 1) return xm_launch<XM_EMIT, 1>(p, st, "extract_links(matrix-core emission)");
    return xm_launch<XM_EMIT, 2>(p, st, "extract_links(matrix-core emission)");
}

extern "C" int dsp_extract_links_bwd_ws(const float* q, const float* k, const float* log_gates, const int64_t* out_len, const float* dist_bias,
                                        const float* links, const float* grad_links, const float* stats,
                                        float* grad_q, float* grad_k, float* grad_log_gates, int B, int L, int H, int CK, int TR, float scale,
                                        void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 1 || TR < 1) { set_error("extract_links_bwd_ws: bad sizes B=%d L=%d TR=%d", B, L, TR); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!xm_supported(L, H, CK, TR)) { set_error("extract_links_bwd_ws: H=%d CK=%d L=%d TR=%d is not served by the matrix-core kernels", H, CK, L, TR); return DSP_EINVAL; }
    if (!q || !k || !log_gates || !out_len || !links || !grad_links || !stats || !grad_q || !grad_k || !grad_log_gates || !workspace) {
        set_error("extract_links_bwd_ws: null pointer"); return DSP_EINVAL; }
    const size_t one = xm_split_bytes(B, L);
    const size_t onet = xm_split_t_bytes(B, L);
    if (workspace_bytes < xm_bwd_bytes(B, L) || ((uintptr_t)workspace & 15) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)grad_q | (uintptr_t)grad_k) & 15)) {
        set_error("extract_links_bwd_ws: workspace of %zu bytes (16-byte aligned) needed, %zu given; q / k / grads 16-byte aligned", xm_bwd_bytes(B, L), workspace_bytes); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    XmParams p{};
    p.q = q; p.k = k; p.gates = log_gates; p.out_len = out_len; p.bias = dist_bias;
    p.clinks = links; p.G = grad_links; p.cstats = stats; p.dgate = grad_log_gates;
    p.B = B; p.L = L; p.TR = TR; p.NT = (L + 31) / 32; p.scale = scale;
    hipLaunchKernelGGL(xl_mfma_split_kernel, dim3((unsigned)p.NT, (unsigned)B, 2), dim3(256), 0, st, k, ws, L, p.NT, q, one);      // k -> ws, q -> ws + one
    if (int rc = check_launch("extract_links_bwd(split)")) return rc;
    p.pa = ws;
    static const char* const e_aq = getenv("DSP_XM_SA_QG");
    const int aq = e_aq ? atoi(e_aq) : (L <= 1536 ? 1 : 2);          // (L = 256: 27 / 18 us, L = 1024: 183 / 173)
    if (int rc = aq == 1 ? xm_launch<XM_SA, 1>(p, st, "extract_links_bwd(matrix-core SA)") : xm_launch<XM_SA, 2>(p, st, "extract_links_bwd(matrix-core SA)")) return rc;
    p.dout = grad_q;
    if (xm_bf16_contraction(L)) {
        g_xl_ran |= 64u;
        char* wt = ws + 2 * one;
        hipLaunchKernelGGL(xl_mfma_split_t_kernel, dim3((unsigned)p.NT, (unsigned)B, 2), dim3(256), 0, st, k, wt, L, p.NT, q, onet);           // k^T -> wt, q^T -> wt + onet
        if (int rc = check_launch("extract_links_bwd(transposed split)")) return rc;
        p.pt = wt;
        if (int rc = xm_launch<XM_DQ, 1, 1>(p, st, "extract_links_bwd(matrix-core dq)")) return rc;
        p.pa = ws + one; p.pt = wt + onet; p.dout = grad_k;
        return xm_launch<XM_DK, 1, 1>(p, st, "extract_links_bwd(matrix-core dk)");
    }
    g_xl_ran |= 32u;
    if (int rc = xm_launch<XM_DQ, 1>(p, st, "extract_links_bwd(matrix-core dq)")) return rc;
    p.pa = ws + one; p.dout = grad_k;
    return xm_launch<XM_DK, 1>(p, st, "extract_links_bwd(matrix-core dk)");
}
