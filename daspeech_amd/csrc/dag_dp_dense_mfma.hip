// dag_dp_dense_mfma.hip — DENSE-window (TR > 64, README's --max-transition-length 99999 => TR = L-1) DAG DP, K2 alpha || K3 beta,
// in EXP SPACE as a blocked triangular matrix product on the f32 matrix cores.
//
// Replaces (for the log-sum DP) the row-sequential log-space kernel dag_dense_kernel<0> of dag_dp_generic.hip, which — like the
// reference's calculate_alpha_kernel / calculate_beta_kernel (dag_loss.cu:94-127, :232-262) — evaluates one exp per (row, vertex,
// predecessor) term and re-reads the whole transition matrix for every DP row (C1: 7.4 ms, C2 at TR = 4095: 260 ms).
//
// Formulation.  Columns are cut into blocks of 64 vertices.  With A[t][i] = 2^(a2[t][i] - s[t][I]) (a2 = alpha * log2 e, s = one integer
// exponent per (row, block)) and E[i][j] = 2^(link2[i][j]) in [0, 1]:
//     P[t][j] = sum_{I < J} 2^(s[t-1][I]) * ( A[t-1][I-block] . E[I-block][J-block] )[j]            <- OFF-DIAGONAL: plain GEMMs over 16 rows t
//             + sum_{i in J, i < j} 2^(a2[t-1][i]) E[i][j]                                          <- DIAGONAL block: sequential in t
//     a2[t][j] = log2 P[t][j] + match2[t][j]
// For a column block J and a chunk of 16 MT rows (MT = 2) all off-diagonal products need only rows of blocks I < J, which are complete when
// block J-1 has finished the same chunk — so block J runs one chunk behind block J-1 (a wavefront over (chunk, block)), and inside
// a tile the [16 x 64] . [64 x 64] products run on v_mfma_f32_16x16x4_f32: exact f32 arithmetic (an fmaf chain), 1/16 of the VALU
// work per term gone to the matrix pipe, one exp per matrix ELEMENT per chunk instead of one per term, and the transition matrix is
// read T/16 times instead of T times.  The diagonal block's recurrence is done by one wave, one column per lane, the previous row
// broadcast from LDS, its 64 x 64 weights resident in registers for the whole kernel.
// beta is the same recurrence in mirrored coordinates (u = L-1-j, rows from T_b-1 down) with the weight addressed as links[j][i-j-1].
//
// One workgroup (4 waves) per (sample, direction, column block) walks the chunks; it waits for its left neighbour's progress word
// (tagged with the launch epoch) before a chunk's products.  Tickets are handed out block-major, so a workgroup only ever waits for
// one that holds a smaller ticket: no co-residency assumption.  Everything another workgroup reads (alpha rows, block exponents,
// progress words) is written with agent-scope (sc1) stores and read with agent-scope loads.
//
// Exactness.  Within a source block all cells are predecessors of every cell of a later block, so scaling a block's row by its own
// maximum loses only terms 126 binades under the largest one of that block; the per-block partial products are combined against a
// running maximum of the block exponents (as an online soft-max would).  In the diagonal block the 64 previous values carry one
// exponent per 8 columns and a column only uses the groups that hold predecessors of it.  A result below 2^-90 of its reference on a
// cell that has a live predecessor is recomputed exactly in log space (all flagged columns of the row at once, one lane each, over the
// live predecessors).  That guard assumes no finite transition flushes to zero in exp space: a range check ahead of the kernel
// (dag_links_weak_kernel) and a budget on the redo work hand batches for which it does not hold to stand-by log-space kernels (`aborted`).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <algorithm>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct DMParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u32* counters;            // [0] ticket, [1] error word, [2] exact-fallback cells, [3] abort flag, [4] predecessors visited by the exact redo
    u32* progress;            // [ndir * B][NJ]        tag_base + chunks completed
    float2* S;                // [ndir * B][T][NJ]     (.x block exponent or DM_SENT, .y first live column of the block (0..63) or 64)
    u32 tag_base;
    int B, T, L, TR, NJ, ndir;
    int dbg;                  // 2 = DSP_DEBUG=prof: cycle accounting of one workgroup (counters[40..47])
    u32 exact_budget;         // predecessors the exact redo may visit before the launch gives up: see launch_dag_dense_mfma
};

constexpr int DM_BW = 64;                   // column block
constexpr float DM_SENT = -1.0e30f;         // "dead" exponent (finite: differences of two of them stay finite)
constexpr float DM_LOG2E = 1.4426950408889634f;
constexpr float DM_LN2 = 0.6931471805599453f;
constexpr u32 DM_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ float dm_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dm_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float dm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// (the builtin is typed int -> int: a float argument would be CONVERTED, i.e. truncated to an integer value)
__device__ __forceinline__ float dm_readlane(float v, int l) { return __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }

// maximum over aligned groups of 8 lanes, result in every lane of the group (v_max_f32 with a DPP source; 2 wait states behind the write)
__device__ __forceinline__ float dm_max8(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    return v;
}
// inclusive prefix maximum over the wave's 8-lane groups of a value that is uniform inside each group (result again uniform per group):
// shift by one group inside each 16-lane row, then the classic row_bcast:15 / row_bcast:31 steps of a wave64 scan
__device__ __forceinline__ float dm_prefmax_groups(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return v;
}
constexpr int DM_EP = 68;                   // pitch of a weight-tile row in LDS (64 + 4: the staging b128 stores of 8 lanes hit 8 different bank groups)
constexpr int DM_NG = 8;                    // diagonal block: exponent groups of 8 columns (a vertex 8 columns right of the DP's diagonal
                                            // already carries ~2^45 times the paths: 16-column groups pushed the diagonal under the guard)

// One chunk = MT MFMA M-tiles of 16 rows (default 2: see launch_dag_dense_mfma).  The products of a tile run as a software pipeline over the source
// blocks: the register stage of block V + D is requested while block V is converted (exp2) into one of two LDS buffers under the MFMAs
// of block V - 1, so a source block costs max(MFMA, conversion) instead of a memory round trip (2.2 us per block before: C1's last block
// spent 1.08 of its 2.1 ms there, and C2 at TR = 4095 was bound by the same per-block latency on every CU).
constexpr int DM_TM = 16;                   // rows of one MFMA M-tile; a chunk is MT of them
constexpr int DM_AP = 68;                   // pitch of an A row (64 + 4, as DM_EP: fragment reads of 16 rows spread over all banks)
constexpr int DM_ET = 64 * DM_EP;           // floats of one E buffer

template <int D, int MT, bool BETA>
__device__ __forceinline__ void dense_mfma_body(const DMParams& p, char* smem_raw, int b, int U, int sd)
{
    constexpr int TM = DM_TM * MT, AT = TM * DM_AP;            // rows per chunk, floats of one A buffer
    float* At = reinterpret_cast<float*>(smem_raw);            // [2][TM][4][16]   A fragment order: [m][k % 4][k / 4], row pitch DM_AP
    float* Et = At + 2 * AT;                                // [2][64][4][16]   B fragment order: [n][k % 4][k / 4], row pitch DM_EP
    float* Sb = Et + 2 * DM_ET;                                // [2][TM]          (unused since the rescale factors are published: SCb)
    float* Xd = Sb + 2 * TM;                                   // [8] (of 2 TM)    diagonal block: exponent of each 8-column group of the previous row
    float* SCb = Xd + 2 * TM;                                  // [2][TM]          log2 of the factor a row's sums take before this block (0: none)
    float* SCf = SCb + 2 * TM;                                 // [2] (of 4)       1 if any row of the block rescales, else 0
    // (r05) the tile's off-diagonal sums and the chunk's emissions are only alive between the products and the end of the diagonal phase,
    // when nobody touches the A buffers: they live IN them (2 TM x 68 floats >= 2 x TM x 64), which is what lets a 64-row chunk keep two
    // workgroups on a CU (72 KB instead of 105)
    float* Poff = At;                                          // [TM][64]         off-diagonal sums of the tile            (aliases At)
    float* Roff = SCf + 4;                                     // [TM]             their reference exponents
    float* FLo = Roff + TM;                                    // [TM]             first live column (global u) among the source blocks
    float* Vd = FLo + TM;                                      // [64]             diagonal block: previous row, 2^(a2 - X[group of 8])
    int* RDY = reinterpret_cast<int*>(Vd + 64);                // [4]              [0] broadcast slot of the readiness poll, [1] "the launch gave up"
    float* A2d = Vd + 68;                                      // [64]             diagonal block: previous row, exact log2 values
    float* Md = At + TM * 64;                                  // [TM][64]         diagonal block: the chunk's emissions   (aliases At)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = tid, wg = wave;
    const int T = p.T, L = p.L, TR = p.TR, NJ = p.NJ;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * L;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = (BETA ? p.beta : p.alpha) + (size_t)b * T * L;
    float2* S = p.S + (size_t)sd * T * NJ;
    u32* prog = p.progress + (size_t)sd * NJ;
    // mirrored coordinates: column u <-> vertex col(u), DP step tt <-> target row(tt); the recurrence runs over predecessors v < u
    auto col = [&](int u) -> int { return BETA ? (L - 1 - u) : u; };
    auto row = [&](int tt) -> int { return BETA ? (Tb - 1 - tt) : tt; };
    const int u0 = BETA ? (L - Lb) : 0;                    // the seed's column; nothing left of u0 + tt is reachable
    // log2 weight of the transition v -> u (v < u): alpha links[v][u-v-1]; beta links[j][i-j-1] with j = col(u), i = col(v)
    // (the load is unconditional at a clamped address and masked afterwards: a guarded load per element compiles to one exec-masked
    //  block and one memory round trip per ELEMENT — the staging loop then runs at ~6 us per source block instead of one latency)
    auto wlog2 = [&](int v, int u) -> float {
        const int d = u - v - 1;
        const bool ok = !(d < 0 || d >= TR || u >= L || v < 0);
        const int src = BETA ? (L - 1 - u) : v;
        const float raw = K[ok ? ((size_t)src * TR + d) : (size_t)0];
        return ok ? raw * DM_LOG2E : NEG_INF;
    };
    const int ub = U * DM_BW;                               // first column of the block
    const int nchunks = (Tb + TM - 1) / TM;

    // ---- rows the recurrence never reaches, and the seed row (tt = 0) of this block
    for (int t = Tb; t < T; ++t)
        for (int ul = tid; ul < DM_BW; ul += 256) { const int u = ub + ul; if (u < L) O[(size_t)t * L + col(u)] = NEG_INF; }

    // ---- diagonal-block state of wave 0 (lane = column ul of the block)
    const int ul = lane, u = ub + lane;
    v2f Ec[32];                                              // 2^(weight of v = ub + i -> u), 0 for i >= ul; pairs for v_pk_fma_f32
    float ref = DM_SENT;                                     // largest group exponent of the previous row among groups 0 .. ul / 8
    int fl_prev = 1 << 30;                                   // first live column (global u) of the previous row inside this block
    // end of a row of the diagonal block: output, next row's broadcast state (values scaled per 8-column group, exact log2 values,
    // group exponents, their prefix maximum), block exponent / first live column for the blocks to the right
    auto row_end = [&](float a2, int tt) {
        if (u < L) dm_st(O + (size_t)row(tt) * L + col(u), a2 * DM_LN2);
        const float gm = dm_max8(a2);
        const float xs = (gm == NEG_INF) ? DM_SENT : ceilf(gm);            // the lane's own group exponent (gm is uniform inside a group)
        Vd[ul] = dm_exp2(a2 - xs);                                         // (-inf - x = -inf -> 0)
        A2d[ul] = a2;
        Xd[ul >> 3] = xs;                                                  // (8 lanes, one address, one value)
        ref = dm_prefmax_groups(xs);
        const u64 lv = __ballot(a2 != NEG_INF);
        fl_prev = lv ? (ub + (int)__builtin_ctzll(lv)) : (1 << 30);
        const float sblk = dm_readlane(ref, 63);
        if (lane == 0) {
            dm_st(&S[(size_t)tt * NJ + U].x, sblk);
            dm_st(&S[(size_t)tt * NJ + U].y, lv ? (float)__builtin_ctzll(lv) : 64.f);
        }
    };
    if (wave == 0) {
        // (-inf for i >= ul: the distance is negative.  All 64 requests first, unguarded, no LDS store between them: guarded loads, or
        //  loads separated by stores through pointers the compiler cannot tell from global memory, are 64 serialized memory round
        //  trips at the head of every block's critical path, ~0.1 ms)
#pragma unroll
        for (int i = 0; i < 32; ++i) { Ec[i].x = wlog2(ub + 2 * i, u); Ec[i].y = wlog2(ub + 2 * i + 1, u); }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            Ec[i].x = dm_exp2(Ec[i].x); Ec[i].y = dm_exp2(Ec[i].y);
        }
        // seed row
        if (lane == 0) RDY[1] = 0;
        const bool seed = (u == u0) && u < L;
        const float m0 = seed ? M[(size_t)row(0) * L + col(u)] * DM_LOG2E : NEG_INF;
        row_end(m0, 0);
    }
    __syncthreads();

    // loop-invariant 32-bit thread offsets of the fast-path loads (see prefetchE): element (it, e) of a source block, relative to the
    // block's uniform base — in BYTES, added to a uniform pointer: that is the shape the compiler turns into  global_load v, voffset, s[base]
    // (no 64-bit VGPR address pair per load; an element index it scales in 64 bits first)
    // The empty asm keeps the 32-bit offset a value DEFINED IN THE BLOCK of the load: hoisted out of the loop as a zero-extended 64-bit
    // pair (LICM does that), instruction selection no longer sees  base + zext(offset)  and emits a 64-bit VALU add per load again.
    auto at_f = [](const float* base, unsigned byte_off) -> const float* {
        asm volatile("" : "+v"(byte_off));
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
    };
    unsigned offE[16];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (columns past the graph — the ragged last block — take the last real column's weights: in-bounds addresses, and a product's
            //  column depends on that column of E alone, so the sums nobody reads are the only ones affected)
            if (!BETA) {
                const int n = min(tid & 63, L - 1 - ub), g = (tid >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4, cc = 4 * (kk0 + e) + kq;
                offE[4 * it + e] = 4u * (unsigned)(cc * (TR - 1) + ub + n - 1);               // links[vb + cc][ub + n - vb - cc - 1], BYTES
            } else {
                const int e0 = tid + 256 * it, hi = min(e0 >> 4, L - 1 - ub), cc = 4 * (e0 & 15) + e;
                offE[4 * it + e] = 4u * (unsigned)((L - 1 - ub - hi) * TR + hi + 63 - cc);    // links[L-1-u][u - vb - cc - 1], base K + ub - 64 - vb, BYTES
            }
        }
    const int lr = lane & 15, lq = lane >> 4;
    const bool prof = (p.dbg & 2) && sd == 0 && U == p.NJ - 1;
    u64 pf_ready = 0, pf_gemm = 0, pf_diag = 0, pf_last = prof ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](u64& acc) { if (prof) { const u64 t = __builtin_amdgcn_s_memtime(); acc += t - pf_last; pf_last = t; } };
    // Giving up.  Data the exp-space products cannot represent (transitions weaker than 2^-126 everywhere: every sum is exactly zero
    // and every cell takes the exact log-space redo) would make this kernel many times slower than the row-sequential log-space one
    // (r02: 12.5 ms against 0.7 ms on a training batch with such links).  The redo row events are counted; past the launch's budget a
    // flag goes up (counters[3]), every workgroup sees it at its next poll / chunk and returns, and the log-space kernels queued
    // behind this launch — which return at once while the flag is down — compute the whole result instead.
    bool aborted = false;
    for (int c = 0; c < nchunks && !aborted; ++c) {
        const int tt0 = c * TM;
        // ================================================================ off-diagonal products: source blocks V < U
        // The tile's sums stay in the MFMA accumulators across ALL source blocks (reading them back per block stalls the wave for the
        // matrix pipe's latency every step).  Row m is scaled by ONE reference exponent: that of its first live source block, moved
        // (and the sums rescaled) only when a block's exponent exceeds it by more than 60 binades — A <= 2^60, E <= 1, <= 4096 terms:
        // the sums stay under 2^72.  Blocks far below the reference flush to zero exactly as they would against a running maximum.
        // Two parties follow the same rule on the same sequence of block exponents and so agree without talking: the thread that
        // converts row tid / 16 (Rm) and thread `row` < TM (Rt, for the diagonal wave, with the first live column FLt) — which also
        // publishes the rescale factor of its row with the block (SCb), so the lanes that own the accumulators only test it
        // (evaluating the rule for their 4 MT rows cost ~10 instructions per row and block).
        v4f pa[MT], pb[MT];
        float Rm[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { pa[mt] = (v4f){0.f, 0.f, 0.f, 0.f}; pb[mt] = (v4f){0.f, 0.f, 0.f, 0.f}; Rm[mt] = DM_SENT; }
        float Rt = DM_SENT, FLt = 1.0e9f;
        auto ref_rule = [](float& ref, float sx) -> float {        // returns log2 of the factor the row's sums take (0: none); selects only
            const bool livex = sx != DM_SENT, first = ref == DM_SENT;
            const bool jump = livex && !first && sx > ref + 60.f;
            const float sc = jump ? ref - sx : 0.f;
            ref = (livex && (first || jump)) ? sx : ref;
            return sc;
        };
        // the chunk's emissions [TM x 64], 4 per thread, requested now and parked in LDS after the products (unconditional loads at
        // clamped addresses: guarded ones compile to one exec-masked block and one memory round trip EACH — 16 us per chunk)
        float em[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * mt + (tid >> 4), q4 = tid & 15, tt = tt0 + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ue = ub + 4 * q4 + e;
                const bool ok = tt >= 1 && tt < Tb && ue < L;
                const float raw = M[ok ? ((size_t)row(tt) * L + col(ue)) : (size_t)0];
                em[mt][e] = ok ? raw * DM_LOG2E : NEG_INF;
            }
        }
        // Rows of the chunk without a source row (step 0 of the first chunk: the seed row; steps >= T_b of the last one) take their
        // operands from the nearest real step instead — valid, published rows — and compute sums nobody reads (the diagonal wave walks
        // m_lo .. m_hi only; a row of a product depends on that row of A alone).  So EVERY chunk runs without row predicates: before,
        // the first and a ragged last chunk took the predicated path (the training shapes, T <= 100: half of their chunks).
        constexpr bool chunk_full = true;
        auto src_step = [&](int m) -> int { return min(max(tt0 + m - 1, 0), Tb - 1); };
        const unsigned offS0 = 8u * (unsigned)(src_step(tid % TM) * NJ);              // (bytes, as offE)
        unsigned offS1[MT], offA[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            offS1[mt] = 8u * (unsigned)(src_step(16 * mt + (tid >> 4)) * NJ);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int srow = row(src_step(16 * mt + (tid >> 4))), q4 = tid & 15;
                offA[mt][e] = 4u * (BETA ? (unsigned)(srow * L + L - 1 - 4 * q4 - e - (ub - DM_BW)) : (unsigned)(srow * L + 4 * q4 + e));
            }
        }
        if (U > 0) {
            // Source block V is usable for this chunk once progress[V] >= tag + c + 1 (then every block left of it is too).  Blocks are
            // consumed left to right and only the LAST ones (the left neighbours, still working on this chunk) are ever waited for, so
            // the products over the earlier blocks overlap the neighbours' work.  `ready_hi` = largest V known complete, refreshed by
            // one vector poll of the next 64 progress words.
            const u32 want = p.tag_base + (u32)c + 1u;
            int ready_hi = -1;
            auto ensure_ready = [&](int V) {
                if (V <= ready_hi) return;
                stamp(pf_gemm);
                if (wave == 0) {
                    u32 spins = 0;
                    for (;;) {
                        const int vq = ready_hi + 1 + lane;
                        const u32 pv = (vq < U) ? __hip_atomic_load(prog + vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (want - 1u);
                        const u64 okm = __ballot((int)(pv - want) >= 0);
                        const int npref = (~okm) ? (int)__builtin_ctzll(~okm) : 64;        // complete blocks in a row from ready_hi + 1
                        if (ready_hi + npref >= V) { if (lane == 0) RDY[0] = ready_hi + npref; break; }
                        if (__hip_atomic_load(&p.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {       // nobody will publish any more
                            if (lane == 0) { RDY[0] = U; RDY[1] = 1; }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > DM_SPIN_LIMIT) { if (lane == 0) { atomicOr(&p.counters[1], 4u); RDY[0] = U; } break; }
                    }
                }
                __syncthreads();
                ready_hi = RDY[0];
                aborted |= RDY[1] != 0;
                __syncthreads();
                stamp(pf_ready);
            };
            // first source block inside the transition window
            int Vmin = 0;
            { const int lim = ub - TR - DM_BW; if (lim >= 0) Vmin = lim / DM_BW + 1; }
            // ... that can hold a live vertex: nothing left of column u0 + tt is reachable at step tt, so in every source row of this chunk
            // (steps >= tt0 - 1) the blocks left of the one holding column u0 + tt0 - 1 are dead.  They are not requested at all (the
            // liveness vote would only drop them after their tile was loaded: 12 % of C2's products at TR = 4095)
            if (tt0 >= 1) Vmin = max(Vmin, (u0 + tt0 - 1) / DM_BW);
            // register stage of a block: RAW loaded words only — exponent (row tl % 16: LDS copy + liveness vote; row tl / 16: this thread's
            // A row) and first-live per source row, 4 alpha values and 16 link values per thread.  Nothing touches a stage between its
            // request and its conversion D steps later (a select right behind the load would put the memory round trip back on every
            // step: that, not the MFMAs, was the 2.2 us per source block of the first version); the validity predicates are integer
            // functions of (V, thread) and are recomputed at conversion time.  All indices are 32-bit (dense_mfma_supported bounds them).
            float st_s[D], st_sa[D][MT], st_f[D], st_a[D][MT][4], st_e[D][16];
            // E element (it, e) of source block V: transition v -> u, address and validity
            auto e_vu = [&](int V, int it, int e, int& v, int& uu) {
                const int vb = V * DM_BW;
                if (!BETA) {
                    // lane <-> column n (coalesced along the row of links), the thread's 4 values of a step share (n, k % 4) and have
                    // consecutive k / 4: one 16-byte LDS store in fragment order
                    const int n = tl & 63, g = (tl >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4;
                    v = vb + 4 * (kk0 + e) + kq; uu = ub + n;
                } else {
                    // W[v][u] = links[col(u)][u-v-1]: for a fixed u contiguous in v
                    const int e0 = tl + 256 * it;
                    v = vb + 4 * (e0 & 15) + e; uu = ub + (e0 >> 4);
                }
            };
            auto e_ok = [&](int v, int uu) -> bool { return (uu - v - 1) < TR && uu < L; };          // (v < ub <= uu: the distance is >= 0)
            // Fast path (every block pair of a dense window; the ragged last block by the clamped columns of offE): all transitions exist, so
            // the loads are  uniform base (scalar, moves with V) + loop-invariant 32-bit thread offset  and the conversion has no
            // predicates — the predicated version below spends ~450 VALU/SALU instructions per source block, 3x the MFMA time.
            auto e_full = [&](int V) -> bool { return min(ub + 63, L - 1) - 1 - V * DM_BW < TR; };            // largest distance of the pair
            auto prefetchE = [&](auto SC, int V) {
                constexpr int s = decltype(SC)::value;
                if (e_full(V)) {
                    const float* Kv = K + (BETA ? (size_t)(ub - DM_BW - V * DM_BW) : (size_t)(V * DM_BW) * (size_t)(TR - 1));
#pragma unroll
                    for (int i = 0; i < 16; ++i) st_e[s][i] = *at_f(Kv, offE[i]);
                    return;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int v, uu; e_vu(V, it, e, v, uu);
                        const int src = BETA ? (L - 1 - uu) : v;
                        const unsigned idx = e_ok(v, uu) ? (unsigned)(src * TR + (uu - v - 1)) : 0u;
                        st_e[s][4 * it + e] = K[idx];
                    }
            };
            auto row_ok = [&](int m) -> bool { const int tt = tt0 + m; return tt >= 1 && tt < Tb; };      // this row's source row is tt - 1
            auto prefetchA = [&](auto SC, int V) {
                constexpr int s = decltype(SC)::value;
                const int vb = V * DM_BW;
                if (chunk_full) {
                    const float* Sv = reinterpret_cast<const float*>(S + V);
                    st_s[s] = dm_ld(at_f(Sv, offS0)); st_f[s] = dm_ld(at_f(Sv, offS0 + 4u));
                    const float* Ov = O + (BETA ? (ub - DM_BW - vb) : vb);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        st_sa[s][mt] = dm_ld(at_f(Sv, offS1[mt]));
#pragma unroll
                        for (int e = 0; e < 4; ++e) st_a[s][mt][e] = dm_ld(at_f(Ov, offA[mt][e]));
                    }
                    return;
                }
                {
                    const int m = tl % TM;
                    const unsigned si = row_ok(m) ? (unsigned)((tt0 + m - 1) * NJ + V) : 0u;
                    st_s[s] = dm_ld(&S[si].x); st_f[s] = dm_ld(&S[si].y);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = 16 * mt + (tl >> 4), q4 = tl & 15;          // A: row m, source columns 4 q4 .. +3
                    {
                        const unsigned si = row_ok(m) ? (unsigned)((tt0 + m - 1) * NJ + V) : 0u;
                        st_sa[s][mt] = dm_ld(&S[si].x);
                    }
                    const int srow = row(tt0 + m - 1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = vb + 4 * q4 + e;
                        const unsigned idx = (row_ok(m) && v < L) ? (unsigned)(srow * L + col(v)) : 0u;
                        st_a[s][mt][e] = dm_ld(O + idx);
                    }
                }
            };
            auto stage_live = [&](auto SC) -> bool {                     // lanes 0..15 of every wave hold the 16 rows: same vote in all waves
                constexpr int s = decltype(SC)::value;
                return __any(row_ok(tl % TM) && st_s[s] != DM_SENT);
            };
            // convert a register stage into LDS buffer nb: exponents, A = 2^(a2 - s), E = 2^(weight)
            auto commit = [&](auto SC, int V, int nb) {
                constexpr int s = decltype(SC)::value;
                const int vb = V * DM_BW;
                if (tl < TM) {
                    const bool ok = row_ok(tl);
                    const float sx = ok ? st_s[s] : DM_SENT;
                    const float sc = ref_rule(Rt, sx);
                    SCb[nb * TM + tl] = sc;
                    const bool anysc = __any(sc != 0.f);                 // (threads < TM <= 64: all in wave 0)
                    if (tl == 0) SCf[nb] = anysc ? 1.f : 0.f;
                    FLt = fminf(FLt, (ok && st_f[s] < 64.f) ? (float)vb + st_f[s] : 1.0e9f);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = 16 * mt + (tl >> 4), q4 = tl & 15;
                    float* Ab = At + nb * AT;
                    if (chunk_full) {           // a dead row has exponent DM_SENT and 64 values -inf: 2^(-inf - ref) = 0 without a select
                        (void)ref_rule(Rm[mt], st_sa[s][mt]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) Ab[m * DM_AP + e * 16 + q4] = dm_exp2(fmaf(st_a[s][mt][e], DM_LOG2E, -Rm[mt]));
                    } else {
                        const float sx = row_ok(m) ? st_sa[s][mt] : DM_SENT;
                        (void)ref_rule(Rm[mt], sx);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool ok = sx != DM_SENT && (vb + 4 * q4 + e) < L;
                            Ab[m * DM_AP + e * 16 + q4] = ok ? dm_exp2(st_a[s][mt][e] * DM_LOG2E - Rm[mt]) : 0.f;       // [m][k % 4][k / 4]
                        }
                    }
                }
                float* Eb = Et + nb * DM_ET;
                const bool efull = e_full(V);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    float w[4];
                    if (efull) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = dm_exp2(st_e[s][4 * it + e] * DM_LOG2E);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int v, uu; e_vu(V, it, e, v, uu);
                            w[e] = e_ok(v, uu) ? dm_exp2(st_e[s][4 * it + e] * DM_LOG2E) : 0.f;
                        }
                    }
                    if (!BETA) {
                        const int n = tl & 63, g = (tl >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4;
                        *reinterpret_cast<v4f*>(Eb + n * DM_EP + kq * 16 + kk0) = (v4f){w[0], w[1], w[2], w[3]};
                    } else {
                        const int e0 = tl + 256 * it;
                        const int hi = e0 >> 4, q4 = e0 & 15;
#pragma unroll
                        for (int e = 0; e < 4; ++e) Eb[hi * DM_EP + e * 16 + q4] = w[e];      // k = 4 q4 + e, n = hi
                    }
                }
            };
            int cur = 0;                       // LDS buffer of the block whose products are pending
            bool have = false;
            auto mfma_issue = [&](int nb) {    // 16 MT x (16x16x4): this wave's slice = columns 16*wave .. +15 of the block; two chains (40-cycle dependent latency)
                const float* Ab = At + nb * AT; const float* Eb = Et + nb * DM_ET;
                float bf[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v4f t4 = *reinterpret_cast<const v4f*>(Eb + (16 * wg + lr) * DM_EP + lq * 16 + 4 * q);
                    bf[4 * q] = t4.x; bf[4 * q + 1] = t4.y; bf[4 * q + 2] = t4.z; bf[4 * q + 3] = t4.w;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // a row whose reference exponent moved with this block has its sums rescaled first (rare: one flag per block)
                    if (SCf[nb] != 0.f) {
                        const v4f sc4 = *reinterpret_cast<const v4f*>(SCb + nb * TM + 16 * mt + 4 * lq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = dm_exp2(sc4[r]); pa[mt][r] *= f; pb[mt][r] *= f; }
                    }
                    float af[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f a4 = *reinterpret_cast<const v4f*>(Ab + (16 * mt + lr) * DM_AP + lq * 16 + 4 * q);
                        af[4 * q] = a4.x; af[4 * q + 1] = a4.y; af[4 * q + 2] = a4.z; af[4 * q + 3] = a4.w;
                    }
#pragma unroll
                    for (int kk = 0; kk < 16; kk += 2) {
                        pa[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bf[kk], pa[mt], 0, 0, 0);
                        pb[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk + 1], bf[kk + 1], pb[mt], 0, 0, 0);
                    }
                }
            };
            // Readiness is never waited for AHEAD of need: a block beyond `ready_hi` when its stage is requested gets its (static) weights
            // requested anyway, and its alpha rows / exponents re-requested when its turn comes (st_ok).  Only the left neighbour(s) are
            // ever in that state, and waiting for them D blocks early would put D - 1 conversions + products behind the wait, on the
            // (chunk, block) wavefront's critical path.
            bool st_ok[D];
            // The common step, fused: (pending block in LDS) x (block V converted) x (block V + D requested), every load and the whole
            // conversion on the predicate-free paths.  The 8 fragment reads go out together (inline asm: the compiler otherwise reuses
            // 8 registers and serialises read -> 4 MFMAs -> read ...: four exposed LDS round trips per block), the alpha conversion
            // covers their latency, and the transition-matrix conversion and the next requests are interleaved with the 16 MFMAs.
            auto fused_step = [&](auto SC, int V, int Wc) {
                constexpr int s = decltype(SC)::value;
                const int nb = cur, wb = cur ^ 1;
                const u32 a_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(At + nb * AT + lr * DM_AP + lq * 16);
                const u32 e_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Et + nb * DM_ET + (16 * wg + lr) * DM_EP + lq * 16);
                v4f fa[MT][4], fb0, fb1, fb2, fb3;
                asm volatile("ds_read_b128 %0, %4\n\t" "ds_read_b128 %1, %4 offset:16\n\t" "ds_read_b128 %2, %4 offset:32\n\t" "ds_read_b128 %3, %4 offset:48"
                             : "=&v"(fb0), "=&v"(fb1), "=&v"(fb2), "=&v"(fb3) : "v"(e_addr) : "memory");
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32 am = a_addr + (u32)(mt * 16 * DM_AP * 4);
                    asm volatile("ds_read_b128 %0, %4\n\t" "ds_read_b128 %1, %4 offset:16\n\t" "ds_read_b128 %2, %4 offset:32\n\t" "ds_read_b128 %3, %4 offset:48"
                                 : "=&v"(fa[mt][0]), "=&v"(fa[mt][1]), "=&v"(fa[mt][2]), "=&v"(fa[mt][3]) : "v"(am) : "memory");
                }
                const float scf = SCf[nb];
                // ---- alpha rows of block V -> A (buffer wb), exponents
                const int vb = V * DM_BW;
                if (tl < TM) {
                    const float sx = st_s[s];
                    const float sc = ref_rule(Rt, sx);
                    SCb[wb * TM + tl] = sc;
                    const bool anysc = __any(sc != 0.f);
                    if (tl == 0) SCf[wb] = anysc ? 1.f : 0.f;
                    FLt = fminf(FLt, (st_f[s] < 64.f) ? (float)vb + st_f[s] : 1.0e9f);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = 16 * mt + (tl >> 4), q4 = tl & 15;
                    float* Ab = At + wb * AT;
                    (void)ref_rule(Rm[mt], st_sa[s][mt]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ab[m * DM_AP + e * 16 + q4] = dm_exp2(fmaf(st_a[s][mt][e], DM_LOG2E, -Rm[mt]));
                }
                // ---- the pending block's rows: a moved reference exponent rescales the sums first (rare)
                if (scf != 0.f) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const v4f c4 = *reinterpret_cast<const v4f*>(SCb + nb * TM + 16 * mt + 4 * lq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = dm_exp2(c4[r]); pa[mt][r] *= f; pb[mt][r] *= f; }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb0), "+v"(fb1), "+v"(fb2), "+v"(fb3), "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]),
                             "+v"(fa[MT - 1][0]), "+v"(fa[MT - 1][1]), "+v"(fa[MT - 1][2]), "+v"(fa[MT - 1][3]) :: "memory");
                float* Eb = Et + wb * DM_ET;
                auto convE = [&](int it) {
                    v4f w4;
                    w4.x = dm_exp2(st_e[s][4 * it] * DM_LOG2E); w4.y = dm_exp2(st_e[s][4 * it + 1] * DM_LOG2E);
                    w4.z = dm_exp2(st_e[s][4 * it + 2] * DM_LOG2E); w4.w = dm_exp2(st_e[s][4 * it + 3] * DM_LOG2E);
                    if (!BETA) {
                        const int n = tl & 63, g = (tl >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4;
                        *reinterpret_cast<v4f*>(Eb + n * DM_EP + kq * 16 + kk0) = w4;
                    } else {
                        const int e0 = tl + 256 * it, hi = e0 >> 4, q4 = e0 & 15;
                        Eb[hi * DM_EP + q4] = w4.x; Eb[hi * DM_EP + 16 + q4] = w4.y; Eb[hi * DM_EP + 32 + q4] = w4.z; Eb[hi * DM_EP + 48 + q4] = w4.w;
                    }
                };
                const float* Kv = K + (BETA ? (size_t)(ub - DM_BW - Wc * DM_BW) : (size_t)(Wc * DM_BW) * (size_t)(TR - 1));
                const float* Sv = reinterpret_cast<const float*>(S + Wc);
                const float* Ov = O + (BETA ? (ub - DM_BW - Wc * DM_BW) : Wc * DM_BW);
                // k-steps q = 0..3 (fragments fa[.][q], fb_q), two MFMAs per chain and step pair, every row tile of the chunk
                auto mf = [&](int q, const v4f& fbq, int j) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        pa[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][q][j], fbq[j], pa[mt], 0, 0, 0);
                        pb[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][q][j + 1], fbq[j + 1], pb[mt], 0, 0, 0);
                    }
                };
                mf(0, fb0, 0); convE(0);
                mf(0, fb0, 2); convE(1);
                mf(1, fb1, 0); convE(2);
                mf(1, fb1, 2); convE(3);
                mf(2, fb2, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) st_e[s][i] = *at_f(Kv, offE[i]);
                mf(2, fb2, 2);
#pragma unroll
                for (int i = 8; i < 16; ++i) st_e[s][i] = *at_f(Kv, offE[i]);
                mf(3, fb3, 0);
                st_s[s] = dm_ld(at_f(Sv, offS0)); st_f[s] = dm_ld(at_f(Sv, offS0 + 4u));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    st_sa[s][mt] = dm_ld(at_f(Sv, offS1[mt]));
#pragma unroll
                    for (int e = 0; e < 4; ++e) st_a[s][mt][e] = dm_ld(at_f(Ov, offA[mt][e]));
                }
                mf(3, fb3, 2);
                st_ok[s] = (V + D) <= ready_hi;
                __syncthreads();
                cur ^= 1;
            };
            auto step = [&](auto SC, int V) {
                constexpr int s = decltype(SC)::value;
                if (V < U && !st_ok[s]) {
                    ensure_ready(V); prefetchA(SC, V);
                    // (consumed here, so that the wait for this re-request sits inside the branch: at the join the compiler must otherwise
                    //  assume "no younger loads behind the stage" on every path and emits vmcnt(0) for the common one too)
                    asm volatile("" :: "v"(st_a[s][MT - 1][3]), "v"(st_a[s][0][0]), "v"(st_s[s]) : "memory");
                }
                const bool live = V < U && stage_live(SC);
                const int W = V + D, Wc = min(W, U - 1);
                // (no per-step time stamps: one s_memtime + s_waitcnt costs 0.2 - 0.6 us under load, more than the step's own phases —
                //  DSP_DEBUG=prof accounts per chunk: readiness waits / products / diagonal block)
                if (have && live && chunk_full && e_full(V)) { fused_step(SC, V, Wc); return; }      // (e_full(V) => e_full(Wc))
                if (have) mfma_issue(cur);
                if (live) commit(SC, V, cur ^ 1);
                prefetchE(SC, Wc);
                prefetchA(SC, Wc);
                st_ok[s] = W <= ready_hi;
                if (have || live) __syncthreads();
                if (live) cur ^= 1;
                have = live;
            };
            auto fill = [&](auto SC) {
                constexpr int s = decltype(SC)::value;
                const int W = Vmin + s, Wc = min(W, U - 1);
                prefetchE(SC, Wc);
                if (s == 0) ensure_ready(Vmin);                             // the chunk's one poll ahead of need: blocks done so far
                prefetchA(SC, Wc);
                st_ok[s] = W <= ready_hi;
            };
            if (Vmin < U) {
            fill(std::integral_constant<int, 0>{});
            if constexpr (D > 1) fill(std::integral_constant<int, 1>{});
            if constexpr (D > 2) fill(std::integral_constant<int, 2>{});
            if constexpr (D > 3) fill(std::integral_constant<int, 3>{});
            }
            for (int Vb = Vmin; Vb < U; Vb += D) {
                step(std::integral_constant<int, 0>{}, Vb);
                if constexpr (D > 1) step(std::integral_constant<int, 1>{}, Vb + 1);
                if constexpr (D > 2) step(std::integral_constant<int, 2>{}, Vb + 2);
                if constexpr (D > 3) step(std::integral_constant<int, 3>{}, Vb + 3);
            }
            if (have) mfma_issue(cur);
        }
        stamp(pf_gemm);
        // ---- hand the tile's off-diagonal sums and the chunk's emissions to the diagonal wave (through the A buffers: every wave's last
        // fragment reads of them must be done first)
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Poff[(16 * mt + 4 * lq + r) * 64 + 16 * wg + lr] = pa[mt][r] + pb[mt][r];
            *reinterpret_cast<v4f*>(Md + (16 * mt + (tid >> 4)) * 64 + 4 * (tid & 15)) = (v4f){em[mt][0], em[mt][1], em[mt][2], em[mt][3]};
        }
        if (tid < TM) { Roff[tid] = Rt; FLo[tid] = FLt; }
        __syncthreads();

        // ================================================================ diagonal block: rows of the chunk in sequence (wave 0)
        if (wave == 0 && !aborted) {
            const int m_lo = (tt0 == 0) ? 1 : 0, m_hi = min(TM, Tb - tt0);
            bool gave_up = false;
            // the exact redo's accounts: kept in registers and posted (one atomic each, plus a look at the give-up flag) once 256
            // predecessors have been visited or the chunk ends — the everyday redo next to the diagonal must not pay a memory round
            // trip per row, it sits on the critical path of every block to its right
            u32 loc_vis = 0, loc_cells = 0;
            auto post = [&]() {
                u32 ev = 0;
                if (lane == 0) { ev = atomicAdd(&p.counters[4], loc_vis); atomicAdd(&p.counters[2], loc_cells); }
                ev = (u32)__builtin_amdgcn_readfirstlane((int)ev);
                if (ev + loc_vis >= p.exact_budget || __hip_atomic_load(&p.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) gave_up = true;
                loc_vis = 0; loc_cells = 0;
            };
            // this row's LDS operands are requested one row ahead
            float n_m2 = Md[m_lo * 64 + ul], n_ro = Roff[m_lo], n_flo = FLo[m_lo], n_po = Poff[m_lo * 64 + ul];
            const u32 vd_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)Vd;
            const u32 xd_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)Xd;
#pragma unroll 1
            for (int m = m_lo; m < m_hi; ++m) {
                const int tt = tt0 + m;
                const float m2 = n_m2, ro = n_ro, flo = n_flo, po = n_po;
                // next row's LDS operands first (older than the batch below: done when it is)
                {
                    const int mn = min(m + 1, TM - 1);
                    n_m2 = Md[mn * 64 + ul]; n_ro = Roff[mn]; n_flo = FLo[mn]; n_po = Poff[mn * 64 + ul];
                }
                // previous row of the block, broadcast: one partial sum per 8-column group (each in its group's scale).  All 18 reads go
                // out together into their own registers (the compiler keeps ~3 in flight in 12 registers: six exposed LDS latencies a row)
                v4f t[16], x0, x1;
                asm volatile("ds_read_b128 %0, %18\n\t"            "ds_read_b128 %1, %18 offset:16\n\t"
                             "ds_read_b128 %2, %18 offset:32\n\t"  "ds_read_b128 %3, %18 offset:48\n\t"
                             "ds_read_b128 %4, %18 offset:64\n\t"  "ds_read_b128 %5, %18 offset:80\n\t"
                             "ds_read_b128 %6, %18 offset:96\n\t"  "ds_read_b128 %7, %18 offset:112\n\t"
                             "ds_read_b128 %8, %18 offset:128\n\t" "ds_read_b128 %9, %18 offset:144\n\t"
                             "ds_read_b128 %10, %18 offset:160\n\t" "ds_read_b128 %11, %18 offset:176\n\t"
                             "ds_read_b128 %12, %18 offset:192\n\t" "ds_read_b128 %13, %18 offset:208\n\t"
                             "ds_read_b128 %14, %18 offset:224\n\t" "ds_read_b128 %15, %18 offset:240\n\t"
                             "ds_read_b128 %16, %19\n\t"           "ds_read_b128 %17, %19 offset:16"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]),
                               "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15]),
                               "=&v"(x0), "=&v"(x1)
                             : "v"(vd_addr), "v"(xd_addr) : "memory");
                v2f acc2[DM_NG];
                auto grp = [&](int g) {
                    const v4f a = t[2 * g], b2 = t[2 * g + 1];
                    v2f r = (v2f){a.x, a.y} * Ec[4 * g];
                    r = __builtin_elementwise_fma((v2f){a.z, a.w}, Ec[4 * g + 1], r);
                    r = __builtin_elementwise_fma((v2f){b2.x, b2.y}, Ec[4 * g + 2], r);
                    acc2[g] = __builtin_elementwise_fma((v2f){b2.z, b2.w}, Ec[4 * g + 3], r);
                };
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]) :: "memory");
                grp(0); grp(1); grp(2);
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]) :: "memory");
                grp(3); grp(4); grp(5);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]), "+v"(x0), "+v"(x1) :: "memory");
                grp(6); grp(7);
                // a column only uses the groups that hold predecessors of it (`ref` = largest exponent among groups 0 .. ul / 8; the sums of
                // the groups to its right are exactly 0, and ldexp(0, anything) = 0 where 0 * 2^(big) would be 0 * inf)
                const float rt = fmaxf(ro, ref);
                const float xg[DM_NG] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                float P = ldexpf(po, (int)(ro - rt));
#pragma unroll
                for (int g = 0; g < DM_NG; ++g) P += ldexpf(acc2[g].x + acc2[g].y, (int)(xg[g] - rt));
                float a2 = __builtin_amdgcn_logf(P) + rt + m2;                                // P = 0 -> -inf
                // ---- exactness guard.  A sum under the threshold is only trusted as "dead" when the cell has no live predecessor.
                const int flp = min((int)fminf(flo, 1.0e9f), fl_prev);                      // first live column of the previous row (global)
                const bool has_pred = flp < u;
                const bool in_graph = (u >= u0 + tt) && (BETA ? true : (u < Lb)) && u < L;
                bool flag = in_graph && has_pred && (m2 != NEG_INF) && !(P >= 0x1p-90f && P <= 0x1p126f);
                if (!in_graph || !has_pred) a2 = NEG_INF;
                if (BETA && (L - 1 - u) < row(tt)) { a2 = NEG_INF; flag = false; }          // K3 only visits columns j >= t (dag_loss.cu loop bounds)
                if (__any(flag)) {
                    // ---- second opinion on the row, still in exp space: every lane against ITS OWN reference.  The shared exponents above
                    // (one per 8 columns, and a column's reference the largest of the groups up to its own) are set by a group's strongest
                    // vertex; where the previous row climbs steeply — left of the band a trained model's emissions draw, 5 - 60 binades per
                    // column — the few predecessors that make up a column's sum sit 100+ binades under the vertices to their right in the
                    // same group and arrive as zeros.  Here lane u uses r_u = max(reference of the source blocks, ceil(max of the block's
                    // exact previous row over v < u)): no term exceeds 1, the strongest predecessor is within a factor 2 of it, and the
                    // weights are the register-resident ones.  64 v_exp + 64 FMA per lane, no memory access: about 1.5 fast rows, on the
                    // rows that need it.  (r02 sent every such cell through the log-space redo below, a chain of gathers per predecessor:
                    // C1 on such scores 89 ms un-budgeted, 7 - 9 ms after the hand-over to the stand-by kernels, against 1.4 ms.)
                    {
                        float pm = A2d[ul];
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(pm, o); pm = (lane >= o) ? fmaxf(pm, y) : pm; }
                        pm = __shfl_up(pm, 1);
                        if (lane == 0) pm = NEG_INF;
                        const float rl = fmaxf(ro, (pm == NEG_INF) ? DM_SENT : ceilf(pm));
                        float P2 = (ro == DM_SENT) ? 0.f : ldexpf(po, (int)fmaxf(ro - rl, -400.f));
                        const v4f* a4 = reinterpret_cast<const v4f*>(A2d);
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const v4f a = a4[g];                                                // (v >= u: weight 0; the clamp keeps 2^(.) finite there)
                            P2 = fmaf(dm_exp2(fminf(a.x - rl, 0.f)), Ec[2 * g].x, P2);
                            P2 = fmaf(dm_exp2(fminf(a.y - rl, 0.f)), Ec[2 * g].y, P2);
                            P2 = fmaf(dm_exp2(fminf(a.z - rl, 0.f)), Ec[2 * g + 1].x, P2);
                            P2 = fmaf(dm_exp2(fminf(a.w - rl, 0.f)), Ec[2 * g + 1].y, P2);
                        }
                        if (flag && P2 >= 0x1p-90f && P2 <= 0x1p126f) { a2 = __builtin_amdgcn_logf(P2) + rl + m2; flag = false; }
                    }
                    const u64 fm = __ballot(flag);
                    if (fm) {
                        // Exact log-space redo of ALL flagged columns of the row at once: the previous row is walked once, 64 columns per
                        // coalesced load (this block's own part from LDS: the exact row), and only its LIVE columns left of the last flagged
                        // one are visited — each flagged lane adds that predecessor's term (its own transition weight: one gather per
                        // lane and live predecessor).  The everyday customers are the cells next to the DP's diagonal (tens of binades per
                        // column under their right-hand neighbours, beyond any shared exponent: a handful of predecessors); the expensive
                        // ones are rows whose sums underflow wholesale — they are what the launch's budget counts (one unit per visited
                        // predecessor).  One column at a time, each with its own wave-wide scan over the whole row, this was 13.5 ms of
                        // a training step's 0.3 ms forward (r02: 263 k such cells per batch).
                        float mx = NEG_INF, sum = 0.f;
                        u32 visited = 0;
                        const int v_first = __builtin_amdgcn_readfirstlane(flp);
                        const int u_hi = ub + 63 - (int)__builtin_clzll(fm);                               // last flagged column
                        const float* prow = O + (size_t)row(tt - 1) * L;
                        for (int v0 = v_first - (v_first & 63); v0 < u_hi; v0 += 64) {        // (chunks aligned with the column blocks)
                            const int v = v0 + lane;
                            float pv;
                            if (v0 >= ub) pv = A2d[lane];                                                  // this block: exact log2 values of the previous row
                            else pv = (v >= 0 && v < L) ? dm_ld(prow + col(v)) * DM_LOG2E : NEG_INF;
                            if (v < v_first || v >= u_hi) pv = NEG_INF;                                    // (no flagged column has it as a predecessor)
                            u64 lm = __ballot(pv != NEG_INF);
                            visited += (u32)__builtin_popcountll(lm);
                            while (lm) {
                                const int kq = (int)__builtin_ctzll(lm); lm &= lm - 1;
                                const float av = dm_readlane(pv, kq);
                                const int vv = v0 + kq;
                                const float x = (flag && vv < u) ? av + wlog2(vv, u) : NEG_INF;                 // (-inf outside the window)
                                const float nm = fmaxf(mx, x);
                                if (nm != NEG_INF) sum = sum * dm_exp2(mx - nm) + dm_exp2(x - nm);
                                mx = nm;
                            }
                        }
                        if (flag) a2 = (mx == NEG_INF) ? NEG_INF : (__builtin_amdgcn_logf(sum) + mx + m2);
                        // diagnostics: the first flagged cells of the launch (sample | dir, step, column, the distrusted sum) — slots by row event
                        if (lane == 0 && loc_cells == 0) {
                            const u32 slot = __hip_atomic_load(&p.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (slot < 14) {
                                const int fu_l = (int)__builtin_ctzll(fm);
                                p.counters[8 + 4 * slot] = (u32)sd | (BETA ? 0x100u : 0u); p.counters[9 + 4 * slot] = (u32)tt; p.counters[10 + 4 * slot] = (u32)(ub + fu_l);
                                p.counters[11 + 4 * slot] = __float_as_uint(dm_readlane(P, fu_l));
                            }
                        }
                        loc_vis += visited; loc_cells += (u32)__builtin_popcountll(fm);
                        if (loc_vis >= 256u) post();
                    }
                }
                row_end(a2, tt);
                if (gave_up) break;
            }
            if (loc_cells && !gave_up) post();
            // ---- publish the chunk: everything above was stored write-through; drain, then the progress word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (gave_up) {
                if (lane == 0) { __hip_atomic_store(&p.counters[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); RDY[1] = 1; }
            } else {
                if (lane == 0) __hip_atomic_store(prog + U, p.tag_base + (u32)c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (one look per chunk, so that workgroups which never wait — the first blocks — stop as well)
                if (__hip_atomic_load(&p.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && lane == 0) RDY[1] = 1;
            }
        }
        __syncthreads();
        aborted |= RDY[1] != 0;
        stamp(pf_diag);
    }
    if (prof && tid == 0) { p.counters[40] = (u32)(pf_ready >> 4); p.counters[41] = (u32)(pf_gemm >> 4); p.counters[42] = (u32)(pf_diag >> 4); p.counters[43] = (u32)nchunks; }
}

template <int D, int MT>
__global__ __launch_bounds__(256) void dag_dense_mfma_kernel(DMParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ u32 s_ticket, s_gone;
    const int tid = threadIdx.x;
    if (tid == 0) { s_ticket = atomicAdd(&p.counters[0], 1u); s_gone = __hip_atomic_load(&p.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (s_gone) return;                              // the launch gave up (see `aborted` in the body): the log-space kernels behind it do the work
    // (readfirstlane: the ticket comes out of LDS and the divisions run on the VALU, so without it the block index, the sample and every
    //  base pointer derived from them live in VGPRs — each "scalar base + thread offset" load then needs a 64-bit VALU add)
    const u32 ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int per = p.ndir * p.B;
    const int U = __builtin_amdgcn_readfirstlane((int)(ticket / per));               // block-major: a workgroup only waits for smaller tickets
    const int rem = __builtin_amdgcn_readfirstlane((int)(ticket % per));
    const bool is_beta = (p.alpha == nullptr) || (p.ndir == 2 && rem >= p.B);
    const int b = __builtin_amdgcn_readfirstlane(rem % p.B);
    const int sd = ((p.ndir == 2 && rem >= p.B) ? 1 : 0) * p.B + b;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid) {                                    // invalid sample: -inf everywhere, no trap; its other blocks do the same, nobody waits
        float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
        for (int t = 0; t < T; ++t)
            for (int ul = tid; ul < DM_BW; ul += 256) { const int u = U * DM_BW + ul; if (u < L) O[(size_t)t * L + (is_beta ? (L - 1 - u) : u)] = NEG_INF; }
        return;
    }
    if (is_beta) dense_mfma_body<D, MT, true>(p, smem_raw, b, U, sd);
    else dense_mfma_body<D, MT, false>(p, smem_raw, b, U, sd);
}

// The same kernel limited to 256 VGPRs, so that TWO workgroups share a CU (LDS: 70 KB each): with one wave per SIMD every instruction
// and every wait of the in-order stream is exposed; a second workgroup fills them.  110 VGPRs spill to scratch for the
// 32-row chunk and it is still the faster build at every shape (C2 at TR = 4095: 27.5 -> 21.6 ms), results bit-identical.
template <int D, int MT>
__global__ __launch_bounds__(256, 2) void dag_dense_mfma_kernel_occ2(DMParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ u32 s_ticket, s_gone;
    const int tid = threadIdx.x;
    if (tid == 0) { s_ticket = atomicAdd(&p.counters[0], 1u); s_gone = __hip_atomic_load(&p.counters[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (s_gone) return;                              // the launch gave up (see `aborted` in the body): the log-space kernels behind it do the work
    // (readfirstlane: the ticket comes out of LDS and the divisions run on the VALU, so without it the block index, the sample and every
    //  base pointer derived from them live in VGPRs — each "scalar base + thread offset" load then needs a 64-bit VALU add)
    const u32 ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int per = p.ndir * p.B;
    const int U = __builtin_amdgcn_readfirstlane((int)(ticket / per));               // block-major: a workgroup only waits for smaller tickets
    const int rem = __builtin_amdgcn_readfirstlane((int)(ticket % per));
    const bool is_beta = (p.alpha == nullptr) || (p.ndir == 2 && rem >= p.B);
    const int b = __builtin_amdgcn_readfirstlane(rem % p.B);
    const int sd = ((p.ndir == 2 && rem >= p.B) ? 1 : 0) * p.B + b;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid) {                                    // invalid sample: -inf everywhere, no trap; its other blocks do the same, nobody waits
        float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
        for (int t = 0; t < T; ++t)
            for (int ul = tid; ul < DM_BW; ul += 256) { const int u = U * DM_BW + ul; if (u < L) O[(size_t)t * L + (is_beta ? (L - 1 - u) : u)] = NEG_INF; }
        return;
    }
    if (is_beta) dense_mfma_body<D, MT, true>(p, smem_raw, b, U, sd);
    else dense_mfma_body<D, MT, false>(p, smem_raw, b, U, sd);
}

// Can exp space hold this batch's transitions at all?  A finite weight under 2^-126 converts to an exact zero; its term is lost, and with
// A values of up to 2^60 (the rows' reference exponents only move in jumps of 60 binades) the lost term can be 2^24 times LARGER than a
// sum the 2^-90 guard still trusts (r02: 2 cells of a 6 x 40 x 330 batch with -100 ... -400 nat transitions off by 3.4 nats).  Batches
// without such weights cannot lose a term that way — log_softmax outputs of trained models stay far above e^-86 — so one streaming pass
// over the transition matrix certifies the fast path; a batch that fails it raises the give-up flag before the DP kernel starts and
// goes to the stand-by log-space kernels as a whole.
__global__ __launch_bounds__(256) void dag_links_weak_kernel(const float* __restrict__ links, size_t n, u32* flag)
{
    constexpr float WEAK = -86.0f;                    // nats; exp2(w log2 e) flushes to zero under -87.3
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsz = (size_t)gridDim.x * 256;
    bool weak = false;
    const size_t head = min(n, (size_t)((16 - ((uintptr_t)links & 15)) & 15) / 4);              // elements before the first 16-byte boundary
    const size_t n4 = (n - head) / 4;
    const v4f* L4 = reinterpret_cast<const v4f*>(links + head);
    for (size_t i = gtid; i < n4; i += gsz) {
        const v4f x = L4[i];
        weak |= (x.x < WEAK && x.x != NEG_INF) | (x.y < WEAK && x.y != NEG_INF) | (x.z < WEAK && x.z != NEG_INF) | (x.w < WEAK && x.w != NEG_INF);
    }
    if (gtid < head) { const float x = links[gtid]; weak |= x < WEAK && x != NEG_INF; }
    if (gtid < n - head - 4 * n4) { const float x = links[head + 4 * n4 + gtid]; weak |= x < WEAK && x != NEG_INF; }
    if (__any(weak) && (threadIdx.x & 63) == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);
size_t dense_rows_gated_bytes(int B, int L, int ndir);
bool dense_rows_gated_supported(int L);
int launch_dag_dense_rows_gated(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int,
                                unsigned int*, unsigned long long*, unsigned int, const unsigned int*, hipStream_t);

// (L beyond what the stand-by log-space kernels take has no exact fallback for transitions exp space flushes: such shapes stay with the
// generic log-space kernel)
// (r05: windows 33 .. 64 too — until then they fell through to the row-sequential generic kernel: 112 ms at C2 / TR = 64; the partially
//  masked tiles of such a window take the predicated conversion path, at most two source blocks per column block)
bool dense_mfma_supported(int L, int TR) { return TR > 32 && L >= 128 && (long)L * TR < (1L << 31) && dense_rows_gated_supported(L); }

template <int D, int MT>
static int launch_dm(const DMParams& p, int nwg, hipStream_t st)
{
    constexpr int TM = DM_TM * MT;
    const size_t lds = (size_t)(2 * TM * DM_AP + 2 * DM_ET + 6 * TM + 4 + 2 * TM + 68 + 64) * 4 + 64;
    auto k = dag_dense_mfma_kernel<D, MT>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(256), lds, st, p);
    return check_launch("dag_loss_fwd(dense mfma)");
}

static thread_local int g_dm_depth = 0, g_dm_mt = 0, g_dm_budget = 0;      // (diagnostic switches are per calling thread, like dp_path)
void set_dm_depth(int v) { g_dm_depth = v; }
void set_dm_mt(int v) { g_dm_mt = v; }
void set_dm_budget(int v) { g_dm_budget = v; }          // 0 = auto, -1 = no stand-by (never give up), n > 0 = that many visited predecessors

int launch_dag_dense_mfma(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                          float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    const int NJ = (L + DM_BW - 1) / DM_BW;
    DMParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NJ = NJ; p.ndir = ndir;
    { static const char* const e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "prof")) ? 2 : 0; }      // (read once per process: no getenv on the launch path)
    const size_t prog_bytes = ((size_t)ndir * B * NJ * sizeof(u32) + 255) / 256 * 256;
    const size_t s_bytes = ((size_t)ndir * B * T * NJ * sizeof(float2) + 255) / 256 * 256;
    // the stand-by log-space kernels (see `aborted` in the kernel).  A predecessor visited by the exact redo costs about one memory
    // latency, roughly what a DP row of a block costs when nothing is flagged: the launch may visit one per (row, block) pair on average
    // before it hands the batch over (ordinary batches: a handful per row next to the diagonal)
    const bool standby = g_dm_budget >= 0 && dense_rows_gated_supported(L);
    const size_t gran_bytes = standby ? dense_rows_gated_bytes(B, L, ndir) : 0;
    p.exact_budget = !standby ? 0xFFFFFFFFu : g_dm_budget > 0 ? (u32)g_dm_budget : (u32)(4096 + (size_t)ndir * B * T * NJ);
    u64* area = nullptr;
    int rc = banded_acquire_ws(st, prog_bytes + s_bytes + gran_bytes, T, &p.counters, &area, &p.tag_base);
    if (rc) return rc;
    p.progress = reinterpret_cast<u32*>(area);
    p.S = reinterpret_cast<float2*>(reinterpret_cast<char*>(area) + prog_bytes);
    u64* gran = reinterpret_cast<u64*>(reinterpret_cast<char*>(area) + prog_bytes + s_bytes);
    if (standby) {
        const size_t n = (size_t)B * L * TR;
        const unsigned g = (unsigned)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
        hipLaunchKernelGGL(dag_links_weak_kernel, dim3(g), dim3(256), 0, st, links, n, p.counters + 3);
        if ((rc = check_launch("dag_loss_fwd(dense mfma, transition range)"))) return rc;
    }
    auto then_standby = [&](int rc0) -> int {
        if (rc0 || !standby) return rc0;
        return launch_dag_dense_rows_gated(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, p.counters, gran, p.tag_base, p.counters + 3, st);
    };
    const int nwg = ndir * B * NJ;
    const int depth = g_dm_depth ? g_dm_depth : 2;       // source blocks in flight per workgroup
    // 16-row MFMA tiles per chunk: 2 (32-row chunks, one register stage) halves the source blocks per DP row and was faster or equal at
    // every shape of the r02 sweep on the pipelined kernel (C1 1.53 / 1.53 ms, B=16 T=150 L=1024 0.79 / 0.71, C2 at TR = 4095 38.6 / 31.2
    // for 16- / 32-row chunks; the pre-pipeline kernel had it the other way round: its blocks cost a memory round trip each)
    const int mt = g_dm_mt ? g_dm_mt : 2;
    if (mt == 14) return then_standby(launch_dm<1, 4>(p, nwg, st));    // (64-row chunks, one workgroup per CU: comparison build)
    if (mt == 3 || mt == 4) {                            // 48- / 64-row chunks, two workgroups per CU: 2/3 / half the passes over the transition matrix
        const int TM = DM_TM * mt;
        const size_t lds = (size_t)(2 * TM * DM_AP + 2 * DM_ET + 6 * TM + 4 + 2 * TM + 68 + 64) * 4 + 64;
        auto k = mt == 3 ? dag_dense_mfma_kernel_occ2<1, 3> : dag_dense_mfma_kernel_occ2<1, 4>;
        set_max_dynamic_lds((const void*)k, (int)lds);
        hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(256), lds, st, p);
        return then_standby(check_launch("dag_loss_fwd(dense mfma)"));
    }
    if (mt >= 2 && g_dm_depth != 9) {                    // default: 32-row chunks, one stage, two workgroups per CU
        constexpr int TM = DM_TM * 2;
        const size_t lds = (size_t)(2 * TM * DM_AP + 2 * DM_ET + 6 * TM + 4 + 2 * TM + 68 + 64) * 4 + 64;
        auto k = dag_dense_mfma_kernel_occ2<1, 2>;
        set_max_dynamic_lds((const void*)k, (int)lds);
        hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(256), lds, st, p);
        return then_standby(check_launch("dag_loss_fwd(dense mfma)"));
    }
    if (mt >= 2) return then_standby(launch_dm<1, 2>(p, nwg, st));     // (dm_depth 9: the one-workgroup-per-CU build of the same kernel, for comparison)
    if (depth <= 1) return then_standby(launch_dm<1, 1>(p, nwg, st));
    return then_standby(launch_dm<2, 1>(p, nwg, st));                  // (three stages: 27 spills)
}

}  // namespace dsp
