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
// For a column block J and a chunk of 16 rows all off-diagonal products need only rows of blocks I < J, which are complete when
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
// exponent per 16 columns and a column only uses the groups that hold predecessors of it.  A result below 2^-90 of its reference on a
// cell that has a live predecessor is recomputed exactly in log space (wave-cooperative scan of the predecessors).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v4f __attribute__((ext_vector_type(4)));

struct DMParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u32* counters;            // [0] ticket, [1] error word, [2] exact-fallback cells
    u32* progress;            // [ndir * B][NJ]        tag_base + chunks completed
    float2* S;                // [ndir * B][T][NJ]     (.x block exponent or DM_SENT, .y first live column of the block (0..63) or 64)
    u32 tag_base;
    int B, T, L, TR, NJ, ndir;
    int dbg;                  // 2 = DSP_DEBUG=prof: cycle accounting of one workgroup (counters[40..47])
};

constexpr int DM_BW = 64;                   // column block
constexpr float DM_SENT = -1.0e30f;         // "dead" exponent (finite: differences of two of them stay finite)
constexpr float DM_LOG2E = 1.4426950408889634f;
constexpr float DM_LN2 = 0.6931471805599453f;
constexpr u32 DM_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ float dm_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dm_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float dm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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
constexpr int DM_EP = 68;                   // pitch of a weight-tile row in LDS (64 + 4: the staging b128 stores of 8 lanes hit 8 different bank groups)
constexpr int DM_NG = 8;                    // diagonal block: exponent groups of 8 columns (a vertex 8 columns right of the DP's diagonal
                                            // already carries ~2^45 times the paths: 16-column groups pushed the diagonal under the guard)

// TM rows per chunk (16 * MT).  BETA: mirrored coordinates, see the header.
// NG wave-groups of 4 waves split the source blocks of a tile between them (block V goes to group V % NG): NG blocks are staged and
// multiplied per loop iteration, so the memory round trip of a stage is paid once per NG blocks (one group: 2.2 us per source block at
// C1, the last block's 31 x 16 products = 1.1 ms on the critical path).
template <int NG> constexpr int dm_group_floats(int TM) { return TM * 64 + 64 * DM_EP + TM + TM + TM * 64 + TM + TM; }

template <int MT, int NG, bool BETA>
__device__ __forceinline__ void dense_mfma_body(const DMParams& p, char* smem_raw, int b, int U, int sd)
{
    constexpr int TM = 16 * MT;
    constexpr int GF = dm_group_floats<NG>(TM);
    const int grp = threadIdx.x >> 8;                          // wave-group
    float* gbase = reinterpret_cast<float*>(smem_raw) + grp * GF;
    float* At = gbase;                                         // [TM][4][16]   A fragment order: [m][k % 4][k / 4]
    float* Et = At + TM * 64;                                  // [64][4][16]   B fragment order: [n][k % 4][k / 4], row pitch DM_EP
    float* Sb = Et + 64 * DM_EP;                               // [TM]          exponent of the source block per row
    float* FLb = Sb + TM;                                      // [TM]          first live column of the source block per row (global u, or 1e9)
    float* Poff = FLb + TM;                                    // [TM][64]      off-diagonal sums of the tile (this group's share)
    float* Roff = Poff + TM * 64;                              // [TM]          their reference exponents
    float* FLo = Roff + TM;                                    // [TM]          first live column (global u) among the group's source blocks
    float* Vd = reinterpret_cast<float*>(smem_raw) + NG * GF;  // [64]          diagonal block: previous row, 2^(a2 - X[group of 8])
    int* RDY = reinterpret_cast<int*>(Vd + 64);                // [4]           broadcast slot of the readiness poll
    float* A2d = Vd + 68;                                      // [64]          diagonal block: previous row, exact log2 values
    float* Wd = A2d + 64;                                      // [64][64]      diagonal block: log2 weights [source i][column], -inf for i >= column
    float* Md = Wd + 64 * 64;                                  // [TM][64]      diagonal block: the chunk's emissions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = tid & 255, wg = wave & 3;                   // thread / wave inside the wave-group
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
        for (int ul = tid; ul < DM_BW; ul += 256 * NG) { const int u = ub + ul; if (u < L) O[(size_t)t * L + col(u)] = NEG_INF; }

    // ---- diagonal-block state of wave 0 (lane = column ul of the block)
    const int ul = lane, u = ub + lane;
    float Ecol[64];                                          // 2^(weight of v = ub + i -> u), 0 for i >= ul
    float a2prev = NEG_INF;
    float Xg[DM_NG];                                         // exponent of each 8-column group of the previous row (wave-uniform)
#pragma unroll
    for (int g = 0; g < DM_NG; ++g) Xg[g] = DM_SENT;
    int fl_prev = 1 << 30;                                   // first live column (global u) of the previous row inside this block
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float wl = (i < ul) ? wlog2(ub + i, u) : NEG_INF;
            Ecol[i] = dm_exp2(wl);
            Wd[i * 64 + ul] = wl;                // kept for the in-block exact redo of near-diagonal cells
        }
        // seed row
        const bool seed = (u == u0) && u < L;
        const float m0 = seed ? M[(size_t)row(0) * L + col(u)] * DM_LOG2E : NEG_INF;
        a2prev = m0;
        if (u < L) dm_st(O + (size_t)row(0) * L + col(u), a2prev * DM_LN2);
        const float gm = dm_max8(a2prev);
#pragma unroll
        for (int g = 0; g < DM_NG; ++g) { const float x = __builtin_amdgcn_readlane(gm, 8 * g); Xg[g] = (x == NEG_INF) ? DM_SENT : ceilf(x); }
        const float xs0 = (gm == NEG_INF) ? DM_SENT : ceilf(gm);          // the lane's own group exponent (gm is uniform inside a group)
        Vd[ul] = (a2prev == NEG_INF) ? 0.f : dm_exp2(a2prev - xs0);
        A2d[ul] = a2prev;
        const u64 lv = __ballot(a2prev != NEG_INF);
        fl_prev = lv ? (ub + (int)__builtin_ctzll(lv)) : (1 << 30);
        if (lane == 0) {
            float sblk = Xg[0];
#pragma unroll
            for (int g = 1; g < DM_NG; ++g) sblk = fmaxf(sblk, Xg[g]);
            float2 sv; sv.x = sblk; sv.y = lv ? (float)__builtin_ctzll(lv) : 64.f;
            dm_st(&S[(size_t)0 * NJ + U].x, sv.x); dm_st(&S[(size_t)0 * NJ + U].y, sv.y);
        }
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    const bool prof = p.dbg == 2 && sd == 0 && U == p.NJ - 1;
    u64 pf_ready = 0, pf_gemm = 0, pf_diag = 0, pf_last = prof ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](u64& acc) { if (prof) { const u64 t = __builtin_amdgcn_s_memtime(); acc += t - pf_last; pf_last = t; } };
    for (int c = 0; c < nchunks; ++c) {
        const int tt0 = c * TM;
        // ================================================================ off-diagonal products: source blocks V < U
        v4f acc[MT];                          // running sums of this wave's 16-column slice, rows 4*lq + r of each 16-row subtile
        float R[MT][4];
        int FL[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) { R[mt][r] = DM_SENT; FL[mt][r] = 1 << 30; } }
        if (U > 0) {
            // Source block V is usable for this chunk once progress[V] >= tag + c + 1 (then every block left of it is too).  Blocks are
            // consumed left to right and only the LAST one (the left neighbour, still working on this chunk) is ever waited for long, so
            // the products over V <= U-2 overlap the neighbour's work.  `ready_hi` = largest V known complete, refreshed by one vector
            // poll of the next 64 progress words.
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
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > DM_SPIN_LIMIT) { if (lane == 0) { atomicOr(&p.counters[1], 4u); RDY[0] = U; } break; }
                    }
                }
                __syncthreads();
                ready_hi = RDY[0];
                __syncthreads();
                stamp(pf_ready);
            };
            // first source block inside the transition window
            int Vmin = 0;
            { const int lim = ub - TR - DM_BW; if (lim >= 0) Vmin = lim / DM_BW + 1; }
            // register stage of block V: exponent / first-live per source row (threads < TM), 4 raw a2 values and 16 raw link values per thread
            float st_s = DM_SENT, st_f = 64.f, st_a[MT][4], st_e[16];
            auto prefetch = [&](int V) {
                const int vb = V * DM_BW;
                if (tl < TM) {
                    const int tt = tt0 + tl;                             // this row's source row is tt - 1
                    const bool ok = tt >= 1 && tt < Tb;
                    const size_t si = ok ? ((size_t)(tt - 1) * NJ + V) : (size_t)0;
                    const float sx = dm_ld(&S[si].x), sy = dm_ld(&S[si].y);
                    st_s = ok ? sx : DM_SENT; st_f = ok ? sy : 64.f;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {                        // A: rows m = 16 mt + tid / 16, source columns 4 (tid % 16) .. +3
                    const int m = 16 * mt + (tl >> 4), q4 = tl & 15;
                    const int tt = tt0 + m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = vb + 4 * q4 + e;
                        const bool ok = tt >= 1 && tt < Tb && v < L;
                        const float raw = dm_ld(O + (ok ? ((size_t)row(tt - 1) * L + col(v)) : (size_t)0));
                        st_a[mt][e] = ok ? raw : NEG_INF;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {                         // E: 64 x 64 weights, 16 per thread
                    if (!BETA) {
                        // lane <-> column n (coalesced along the row of links), the thread's 4 values of a step share (n, k % 4) and have
                        // consecutive k / 4: one 16-byte LDS store in fragment order
                        const int n = tl & 63, g = (tl >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) st_e[4 * it + e] = wlog2(vb + 4 * (kk0 + e) + kq, ub + n);
                    } else {
                        // W[v][u] = links[col(u)][u-v-1]: for a fixed u contiguous in v
                        const int e0 = tl + 256 * it;
                        const int hi = e0 >> 4, q4 = e0 & 15;
#pragma unroll
                        for (int e = 0; e < 4; ++e) st_e[4 * it + e] = wlog2(vb + 4 * q4 + e, ub + hi);
                    }
                }
            };
            // iteration Vp handles blocks Vp .. Vp + NG - 1, one per wave-group (a group past the end idles through the barriers)
            if (Vmin < U) { ensure_ready(min(Vmin + NG - 1, U - 1)); if (Vmin + grp < U) prefetch(Vmin + grp); }
            for (int Vp = Vmin; Vp < U; Vp += NG) {
                const int V = Vp + grp;
                const bool mine = V < U;
                const int vb = V * DM_BW;
                // ---- commit the register stage to LDS: exponents first (A needs them), then A = 2^(a2 - s) and E = 2^(weight)
                if (tl < TM) { Sb[tl] = mine ? st_s : DM_SENT; FLb[tl] = (mine && st_f < 64.f) ? (float)vb + st_f : 1.0e9f; }
                __syncthreads();
                bool any_live = false;
#pragma unroll
                for (int m = 0; m < TM; ++m) any_live |= (Sb[m] != DM_SENT);
                if (any_live) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int m = 16 * mt + (tl >> 4), q4 = tl & 15;
                        const float sx = Sb[m];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            At[(m * 4 + e) * 16 + q4] = (sx != DM_SENT) ? dm_exp2(st_a[mt][e] * DM_LOG2E - sx) : 0.f;       // [m][k % 4][k / 4]
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (!BETA) {
                            const int n = tl & 63, g = (tl >> 6) * 4 + it, kq = g & 3, kk0 = (g >> 2) * 4;
                            v4f w4;
                            w4.x = dm_exp2(st_e[4 * it]); w4.y = dm_exp2(st_e[4 * it + 1]); w4.z = dm_exp2(st_e[4 * it + 2]); w4.w = dm_exp2(st_e[4 * it + 3]);
                            *reinterpret_cast<v4f*>(Et + n * DM_EP + kq * 16 + kk0) = w4;
                        } else {
                            const int e0 = tl + 256 * it;
                            const int hi = e0 >> 4, q4 = e0 & 15;
#pragma unroll
                            for (int e = 0; e < 4; ++e) Et[hi * DM_EP + e * 16 + q4] = dm_exp2(st_e[4 * it + e]);      // k = 4 q4 + e, n = hi
                        }
                    }
                }
                __syncthreads();
                // ---- the next blocks' loads go out now and land under these blocks' MFMAs
                if (Vp + NG < U) { ensure_ready(min(Vp + 2 * NG - 1, U - 1)); if (V + NG < U) prefetch(V + NG); }
                if (any_live) {
                    // ---- 16 x (16x16x4) MFMA per 16-row subtile; this wave's slice = columns 16*wave .. +15 of the block
                    float bf[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f t4 = *reinterpret_cast<const v4f*>(Et + (16 * wg + lr) * DM_EP + lq * 16 + 4 * q);
                        bf[4 * q] = t4.x; bf[4 * q + 1] = t4.y; bf[4 * q + 2] = t4.z; bf[4 * q + 3] = t4.w;
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        float af[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const v4f t4 = *reinterpret_cast<const v4f*>(At + ((16 * mt + lr) * 4 + lq) * 16 + 4 * q);
                            af[4 * q] = t4.x; af[4 * q + 1] = t4.y; af[4 * q + 2] = t4.z; af[4 * q + 3] = t4.w;
                        }
                        v4f pa = (v4f){0.f, 0.f, 0.f, 0.f}, pb = (v4f){0.f, 0.f, 0.f, 0.f};          // two chains: 40-cycle dependent latency
#pragma unroll
                        for (int kk = 0; kk < 16; kk += 2) {
                            pa = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], bf[kk], pa, 0, 0, 0);
                            pb = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk + 1], bf[kk + 1], pb, 0, 0, 0);
                        }
                        // fold into the running sums against the running maximum of the block exponents (rows 4*lq + r)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = 16 * mt + 4 * lq + r;
                            const float sx = Sb[m];
                            const float rn = fmaxf(R[mt][r], sx);
                            acc[mt][r] = acc[mt][r] * dm_exp2(R[mt][r] - rn) + (pa[r] + pb[r]) * dm_exp2(sx - rn);
                            R[mt][r] = rn;
                            FL[mt][r] = min(FL[mt][r], (int)fminf(FLb[m], 1.0e9f));
                        }
                    }
                }
                __syncthreads();
            }
        }
        stamp(pf_gemm);
        // ---- hand the tile's off-diagonal sums to the diagonal wave
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mt + 4 * lq + r;
                Poff[m * 64 + 16 * wg + lr] = acc[mt][r];
                if (wg == 0 && lr == 0) { Roff[m] = R[mt][r]; FLo[m] = (float)FL[mt][r]; }
            }
        __syncthreads();

        // ================================================================ diagonal block: rows of the chunk in sequence (wave 0)
        if (wave == 0) {
            // the chunk's emissions for this column: all loads in flight at once (one per row made the row time a memory round trip)
            {
                float mrow[TM];
#pragma unroll
                for (int m = 0; m < TM; ++m) { const int tt = tt0 + m; mrow[m] = (tt >= 1 && tt < Tb && u < L) ? M[(size_t)row(tt) * L + col(u)] : NEG_INF; }
#pragma unroll
                for (int m = 0; m < TM; ++m) Md[m * 64 + ul] = mrow[m] * DM_LOG2E;
            }
#pragma unroll 1
            for (int m = 0; m < TM; ++m) {
                const int tt = tt0 + m;
                if (tt == 0) continue;
                if (tt >= Tb) break;
                const float m2 = Md[m * 64 + ul];
                // previous row of the block, broadcast: one partial sum per 8-column group (each in its group's scale)
                float part[DM_NG];
#pragma unroll
                for (int g = 0; g < DM_NG; ++g) part[g] = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const v4f t4 = *reinterpret_cast<const v4f*>(Vd + 4 * q);
                    part[q >> 1] = fmaf(t4.x, Ecol[4 * q], part[q >> 1]);
                    part[q >> 1] = fmaf(t4.y, Ecol[4 * q + 1], part[q >> 1]);
                    part[q >> 1] = fmaf(t4.z, Ecol[4 * q + 2], part[q >> 1]);
                    part[q >> 1] = fmaf(t4.w, Ecol[4 * q + 3], part[q >> 1]);
                }
                // a column only uses the groups that hold predecessors of it: reference = largest exponent among groups 0 .. ul/8
                const int gl = ul >> 3;
                float ref = Xg[0];
#pragma unroll
                for (int g = 1; g < DM_NG; ++g) if (gl >= g) ref = fmaxf(ref, Xg[g]);
                float ro = Roff[m], flo = FLo[m];                  // (wave 0 belongs to group 0: its pointers are group 0's)
#pragma unroll
                for (int g2 = 1; g2 < NG; ++g2) { ro = fmaxf(ro, Roff[g2 * GF + m]); flo = fminf(flo, FLo[g2 * GF + m]); }
                const float rt = fmaxf(ro, ref);
                float Pd = 0.f;
#pragma unroll
                for (int g = 0; g < DM_NG; ++g) Pd += (g <= gl) ? part[g] * dm_exp2(Xg[g] - rt) : 0.f;   // (a group right of the column has a
                                                                                       // zero sum but may have a LARGER exponent: 0 * inf)
                float P = Pd;
#pragma unroll
                for (int g2 = 0; g2 < NG; ++g2) P += Poff[g2 * GF + m * 64 + ul] * dm_exp2(Roff[g2 * GF + m] - rt);
                float a2 = __builtin_amdgcn_logf(P) + rt + m2;                                // P = 0 -> -inf
                // ---- exactness guard.  A sum under the threshold is only trusted as "dead" when the cell has no live predecessor.
                const int flp = min((int)fminf(flo, 1.0e9f), fl_prev);                      // first live column of the previous row (global)
                const bool has_pred = flp < u;
                const bool in_graph = (u >= u0 + tt) && (BETA ? true : (u < Lb)) && u < L;
                bool flag = in_graph && has_pred && (m2 != NEG_INF) && !(P >= 0x1p-90f && P <= 0x1p126f);
                if (!in_graph || !has_pred) a2 = NEG_INF;
                if (BETA && (L - 1 - u) < row(tt)) { a2 = NEG_INF; flag = false; }          // K3 only visits columns j >= t (dag_loss.cu loop bounds)
                // (a) every live predecessor inside this block (the DP's diagonal runs through it: the cells next to the diagonal are
                //     tens of binades per column under their right-hand neighbours, beyond any shared exponent): a handful of terms,
                //     summed in log space from the exact row and the log weights in LDS, all flagged lanes at once
                if (__any(flag) && flp >= ub) {
                    if (flag) {
                        float mx = NEG_INF, sum = 0.f;
                        for (int i = flp - ub; i < ul; ++i) {
                            const float x = A2d[i] + Wd[i * 64 + ul];
                            const float nm = fmaxf(mx, x);
                            if (nm != NEG_INF) sum = sum * dm_exp2(mx - nm) + dm_exp2(x - nm);
                            mx = nm;
                        }
                        a2 = (mx == NEG_INF) ? NEG_INF : (__builtin_amdgcn_logf(sum) + mx + m2);
                        flag = false;
                    }
                }
                u64 fm = __ballot(flag);
                while (fm) {                     // (b) exact log-space redo, one flagged column at a time, the wave scans its predecessors
                    const int fu_l = (int)__builtin_ctzll(fm); fm &= fm - 1;
                    const int fu = ub + fu_l;
                    const int v_lo = max(__builtin_amdgcn_readfirstlane(flp), fu - TR);
                    float mx = NEG_INF, sum = 0.f;
                    for (int v = v_lo + lane; v < fu; v += 64) {
                        const float x = dm_ld(O + (size_t)row(tt - 1) * L + col(v)) * DM_LOG2E + wlog2(v, fu);
                        const float nm = fmaxf(mx, x);
                        if (nm != NEG_INF) sum = sum * dm_exp2(mx - nm) + dm_exp2(x - nm);
                        mx = nm;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const float m2o = __shfl_xor(mx, o, 64), s2o = __shfl_xor(sum, o, 64);
                        const float nm = fmaxf(mx, m2o);
                        sum = (nm == NEG_INF) ? 0.f : sum * dm_exp2(mx - nm) + s2o * dm_exp2(m2o - nm);
                        mx = nm;
                    }
                    const float exact = (mx == NEG_INF) ? NEG_INF : (__builtin_amdgcn_logf(sum) + mx);
                    if (lane == fu_l) a2 = exact + m2;
                    if (lane == fu_l) {      // diagnostics: first flagged cells of the launch (sample | dir, step, column, the distrusted sum)
                        const u32 slot = atomicAdd(&p.counters[2], 1u);
                        if (slot < 14) { p.counters[8 + 4 * slot] = (u32)sd | (BETA ? 0x100u : 0u); p.counters[9 + 4 * slot] = (u32)tt; p.counters[10 + 4 * slot] = (u32)fu; p.counters[11 + 4 * slot] = __float_as_uint(P); }
                    }
                }
                // ---- the row: output, next row's broadcast state, block exponent / first live column for the blocks to the right
                if (u < L) dm_st(O + (size_t)row(tt) * L + col(u), a2 * DM_LN2);
                a2prev = a2;
                const float gm = dm_max8(a2);
#pragma unroll
                for (int g = 0; g < DM_NG; ++g) { const float x = __builtin_amdgcn_readlane(gm, 8 * g); Xg[g] = (x == NEG_INF) ? DM_SENT : ceilf(x); }
                const float xs = (gm == NEG_INF) ? DM_SENT : ceilf(gm);            // own group's exponent (gm is uniform inside a group)
                Vd[ul] = (a2 == NEG_INF) ? 0.f : dm_exp2(a2 - xs);               // (all reads of Vd for this row are done: same wave, program order)
                A2d[ul] = a2;
                const u64 lv = __ballot(a2 != NEG_INF);
                fl_prev = lv ? (ub + (int)__builtin_ctzll(lv)) : (1 << 30);
                if (lane == 0) {
                    float sblk = Xg[0];
#pragma unroll
                    for (int g = 1; g < DM_NG; ++g) sblk = fmaxf(sblk, Xg[g]);
                    dm_st(&S[(size_t)tt * NJ + U].x, sblk);
                    dm_st(&S[(size_t)tt * NJ + U].y, lv ? (float)__builtin_ctzll(lv) : 64.f);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            // ---- publish the chunk: everything above was stored write-through; drain, then the progress word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog + U, p.tag_base + (u32)c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        stamp(pf_diag);
    }
    if (prof && tid == 0) { p.counters[40] = (u32)(pf_ready >> 4); p.counters[41] = (u32)(pf_gemm >> 4); p.counters[42] = (u32)(pf_diag >> 4); p.counters[43] = (u32)nchunks; }
    (void)a2prev;
}

template <int MT, int NG>
__global__ __launch_bounds__(256 * NG) void dag_dense_mfma_kernel(DMParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ u32 s_ticket;
    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = s_ticket;
    const int per = p.ndir * p.B;
    const int U = (int)(ticket / per);               // block-major: a workgroup only waits for smaller tickets
    const int rem = (int)(ticket % per);
    const bool is_beta = (p.alpha == nullptr) || (p.ndir == 2 && rem >= p.B);
    const int b = rem % p.B;
    const int sd = ((p.ndir == 2 && rem >= p.B) ? 1 : 0) * p.B + b;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid) {                                    // invalid sample: -inf everywhere, no trap; its other blocks do the same, nobody waits
        float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
        for (int t = 0; t < T; ++t)
            for (int ul = tid; ul < DM_BW; ul += 256 * NG) { const int u = U * DM_BW + ul; if (u < L) O[(size_t)t * L + (is_beta ? (L - 1 - u) : u)] = NEG_INF; }
        return;
    }
    if (is_beta) dense_mfma_body<MT, NG, true>(p, smem_raw, b, U, sd);
    else dense_mfma_body<MT, NG, false>(p, smem_raw, b, U, sd);
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool dense_mfma_supported(int L, int TR) { return TR > 64 && L >= 128; }

template <int MT, int NG>
static int launch_dm(const DMParams& p, int nwg, hipStream_t st)
{
    constexpr int TM = 16 * MT;
    const size_t lds = (size_t)(NG * dm_group_floats<NG>(TM) + 64 + 4 + 64 + 64 * 64 + TM * 64) * 4 + 64;
    auto k = dag_dense_mfma_kernel<MT, NG>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(256 * NG), lds, st, p);
    return check_launch("dag_loss_fwd(dense mfma)");
}

static int g_dm_mt = 0, g_dm_ng = 0;
void set_dm_mt(int v) { g_dm_mt = v; }
void set_dm_ng(int v) { g_dm_ng = v; }

int launch_dag_dense_mfma(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                          float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const int ndir = (alpha && beta) ? 2 : 1;
    const int NJ = (L + DM_BW - 1) / DM_BW;
    DMParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NJ = NJ; p.ndir = ndir;
    { const char* e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "prof")) ? 2 : 0; }
    const size_t prog_bytes = ((size_t)ndir * B * NJ * sizeof(u32) + 255) / 256 * 256;
    const size_t s_bytes = (size_t)ndir * B * T * NJ * sizeof(float2);
    u64* area = nullptr;
    int rc = banded_acquire_ws(st, prog_bytes + s_bytes, T, &p.counters, &area, &p.tag_base);
    if (rc) return rc;
    p.progress = reinterpret_cast<u32*>(area);
    p.S = reinterpret_cast<float2*>(reinterpret_cast<char*>(area) + prog_bytes);
    const int nwg = ndir * B * NJ;
    // rows per chunk: 16 keeps the (chunk, block) wavefront short — it won the r02 sweep at every shape tried (C1: 2.2 / 3.0 / 3.4 ms,
    // C2 at TR = 4095: 35 / 65 / 52 ms for 16 / 32 / 64 rows, which halve / quarter the passes over the transition matrix)
    int mt = g_dm_mt ? g_dm_mt : 1;
    // two wave-groups only with 16-row chunks: with 32 / 64 rows the second group's accumulators no longer fit the register file
    const int ng = (mt == 1) ? (g_dm_ng ? g_dm_ng : 1) : 1;      // (two groups halve the last block's product time at C1 but not the launch: r02 sweep)
    if (ng == 1) {
        if (mt >= 4) return launch_dm<4, 1>(p, nwg, st);
        if (mt == 2) return launch_dm<2, 1>(p, nwg, st);
        return launch_dm<1, 1>(p, nwg, st);
    }
    if (mt >= 4) return launch_dm<4, 2>(p, nwg, st);
    if (mt == 2) return launch_dm<2, 2>(p, nwg, st);
    return launch_dm<1, 2>(p, nwg, st);
}

}  // namespace dsp
