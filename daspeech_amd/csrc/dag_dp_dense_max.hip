// dag_dp_dense_max.hip — DENSE-window (TR > 64) max-DP of dag_best_alignment (K6) as a blocked max-plus product, plus a trace-free
// back-trace (K7).
//
// Replaces, for the fused alignment op, dag_dense_kernel<1> of dag_dp_generic.hip (row-sequential: every DP row re-reads the whole
// transition matrix — C1: 4.7 ms, C2 at TR = 4095: 119 ms) and, like dag_dp_maxstrip.hip does for banded graphs, the B*T*L int32 trace
// tensor of the reference (dag_best_alignment.cu:39-130 keeps the arg-max of every cell; the back-trace reads T of them).
//
// Same decomposition as dag_dp_dense_mfma.hip: 64-column blocks, chunks of DX_TM rows, block U runs one chunk behind block U-1 (progress
// words tagged with the launch epoch, tickets block-major).  The off-diagonal part of a tile is a [16 x 64] (+, max) [64 x 64] product —
// there is no matrix-core form of it, so it runs on the VALU: a thread owns one column and four rows, the source rows are LDS
// broadcasts, the weights LDS reads at unit stride.  No exponents, no guard: add and max are exact, so alpha_max is bit-identical to
// the sequential scan, whatever the blocking.  The diagonal block is walked row by row by one wave (column per lane, its weights in
// registers, max3).
// The back-trace recomputes the arg-max of the T cells it visits from alpha_max and the links (smallest predecessor index among equal
// maxima: the torch rule).  r03: the max-DP also leaves, per cell, the index of the 64-column BLOCK that holds its arg-max
// (2 bytes instead of the reference's 4-byte trace entry), so a hop evaluates 64 candidates instead of up to L
// (dag_dense_backtrace_blk_kernel; the full scan it replaces took 6 us per hop at C2 / TR = 4095: 0.5 MB of cache lines for one cell).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v4f __attribute__((ext_vector_type(4)));

struct DXParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha;             // alpha_max [B,T,L]
    u32* counters;            // [0] ticket, [1] error word
    u32* progress;            // [B][NJ]        tag_base + chunks completed
    float* S;                 // [B][T][NJ]     block maximum of the row (-inf = nothing alive)
    u32 tag_base;
    int B, T, L, TR, NJ;
    unsigned short* btrace;   // [B][T][L]  the 64-column block that holds a cell's arg-max predecessor (0xFFFF: none) — see dag_dense_backtrace_blk_kernel
};

// chunks of DX_MT x 16 rows.  32-row chunks (DX_MT = 2: a weight read serves 8 rows of a thread, the per-block overhead is paid half
// as often) measured WORSE here except at the largest shape — C1 2.12 -> 2.32 ms, B=32 T=100 L=400 0.41 -> 0.45, C2 at TR=4095
// 22.8 -> 22.1 — unlike the matrix-core kernel: the (+, max) product is VALU work proportional to the rows either way.
// r05: the 16-row kernel of r04 moves the whole transition matrix T/16 times — 68.7 GB per launch at C2 / TR = 4095, 5.5 TB/s over its 12.5 ms: it had
// become HBM-bound.  DX_MT is a template parameter now; 32-row chunks halve that traffic (see launch_dag_dense_max for who gets which).
constexpr int DX_BW = 64, DX_WP = 68;      // weight tile [n = column][k = source], row pitch 68 floats: a thread reads its column's next four k as ONE ds_read_b128
                                                                          // (16-lane groups land on 16 different 16-byte slots: 272-byte stride); [k][n] with pitch 65 took four ds_read_b32 per four k
constexpr u32 DX_SPIN_LIMIT = 1u << 24;

// wave-wide maximum, wave-uniform result: 4 DPP steps inside the 16-lane rows, then the four rows through readlane
__device__ __forceinline__ float dx_wave_max(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    // (the readlane builtin is typed int -> int: a float argument would be CONVERTED — -inf became -2^31, and no dead block was ever skipped)
    auto rl = [](float x, int l) -> float { return __uint_as_float((u32)__builtin_amdgcn_readlane((int)__float_as_uint(x), l)); };
    return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}

__device__ __forceinline__ float dx_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dx_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int DX_MT>
__device__ __forceinline__ void dag_dense_max_body(const DXParams& p)
{
    constexpr int DX_TM = 16 * DX_MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ u32 s_ticket;
    float* At = smem;                                  // [2][TM][64]   source rows (previous DP row of block V), row-major
    float* Wt = At + 2 * DX_TM * 64;                   // [2][64][68]   weights [n = column][k = source]
    float* Sb = Wt + 2 * 64 * DX_WP;                   // [2][TM]       block maximum per source row
    float* Poff = Sb + 2 * DX_TM;                      // [TM][64]      off-diagonal maxima of the tile
    float* Vd = Poff + DX_TM * 64;                     // [64]          diagonal block: previous row
    float* Md = Vd + 64;                               // [TM][64]      the chunk's emissions
    int* RDY = reinterpret_cast<int*>(Md + DX_TM * 64);
    int* Boff = RDY + 4;                               // [TM][64]      block of the off-diagonal maximum (see `bv`)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = s_ticket;
    const int U = (int)(ticket / p.B), b = (int)(ticket % p.B);         // block-major: a workgroup only waits for smaller tickets
    const int T = p.T, L = p.L, TR = p.TR, NJ = p.NJ;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * L;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = p.alpha + (size_t)b * T * L;
    float* S = p.S + (size_t)b * T * NJ;
    unsigned short* BT = p.btrace + (size_t)b * T * L;
    u32* prog = p.progress + (size_t)b * NJ;
    const int ub = U * DX_BW;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    for (int t = valid ? Tb : 0; t < T; ++t)
        for (int ul = tid; ul < DX_BW; ul += 256) { const int u = ub + ul; if (u < L) O[(size_t)t * L + u] = NEG_INF; }
    if (!valid) return;
    // weight of the transition v -> u (v < u): links[v][u-v-1]; unconditional load at a clamped address, masked afterwards
    // (32-bit indices: dense_max_supported bounds L * TR)
    auto w_ok = [&](int v, int u) -> bool { const int d = u - v - 1; return !(d < 0 || d >= TR || u >= L || v < 0); };
    auto w_idx = [&](int v, int u) -> unsigned { return w_ok(v, u) ? (unsigned)(v * TR + (u - v - 1)) : 0u; };
    const int nchunks = (Tb + DX_TM - 1) / DX_TM;

    // ---- diagonal-block state of wave 0 (lane = column) and the seed row
    const int ul = lane, u = ub + lane;
    float Wcol[64];
    if (wave == 0) {
        // all 64 requests first, unguarded (a guarded load is an exec-masked block with its own s_waitcnt vmcnt(0): 64 serialized round
        // trips at the head of every block's critical path — dag_dp_dense_mfma.hip found the same)
#pragma unroll
        for (int i = 0; i < 64; ++i) Wcol[i] = K[w_idx(ub + i, u)];
#pragma unroll
        for (int i = 0; i < 64; ++i) Wcol[i] = w_ok(ub + i, u) ? Wcol[i] : NEG_INF;
        const float a0 = (u == 0) ? M[0] : NEG_INF;                        // alpha_max[0,0] = match[0,0]   (dag_best_alignment.cu:72-74)
        if (u < L) dx_st(O + u, a0);
        Vd[ul] = a0;
        const float bm = dx_wave_max(a0);
        if (lane == 0) dx_st(&S[U], bm);
    }
    __syncthreads();

    // product phase: a thread owns column n of the block and rows 4 mg .. 4 mg + 3 of the chunk
    const int n = tid & 63, mg = tid >> 6;
    // loop-invariant offsets of the predicate-free loads (full tiles): weight element (it, e) = source row k = 16 mg + 4 it + e, column n
    unsigned offW[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) offW[i] = (unsigned)((16 * mg + i) * (TR - 1) + ub + min(n, L - 1 - ub) - 1);          // links[vb + k][ub + n - vb - k - 1]
    // (columns past the graph — the ragged last block — take the last real column's weights: in-bounds, and a column of the product
    //  depends on that column of the weights alone; rows without a source step likewise take the nearest real step's: dag_dp_dense_mfma.hip)
    for (int c = 0; c < nchunks; ++c) {
        const int tt0 = c * DX_TM;
        float acc[DX_MT][4];
#pragma unroll
        for (int mt = 0; mt < DX_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = NEG_INF;
        // Which source block holds the maximum.  The back-trace needs, for the T cells it visits, the arg-max predecessor; recomputing
        // it from alpha_max means scanning up to L predecessors whose transition weights sit in L different cache lines (C2 at
        // TR = 4095: 0.5 MB of lines and 6 us per hop).  A block index per cell (2 bytes) narrows that to 64.  Blocks arrive in
        // ascending order and only a STRICT improvement moves the index, so among equal maxima the smallest predecessor index wins,
        // as in the sequential scan (the back-trace takes the smallest index inside the block).
        int bv[DX_MT][4];
#pragma unroll
        for (int mt = 0; mt < DX_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[mt][r] = 0xFFFF;
        // the chunk's emissions, 4 per thread, requested now and parked in LDS after the products
        float em[DX_MT][4];
#pragma unroll
        for (int mt = 0; mt < DX_MT; ++mt) {
            const int m = 16 * mt + (tid >> 4), q4 = tid & 15, tt = tt0 + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ue = ub + 4 * q4 + e;
                const bool ok = tt >= 1 && tt < Tb && ue < L;
                const float raw = M[ok ? (unsigned)(tt * L + ue) : 0u];
                em[mt][e] = ok ? raw : NEG_INF;
            }
        }
        constexpr bool chunk_full = true;
        auto src_step = [&](int m) -> int { return min(max(tt0 + m - 1, 0), Tb - 1); };
        const unsigned offS = (unsigned)(src_step(tid & (DX_TM - 1)) * NJ);
        unsigned offA[DX_MT];                                  // (clamped per row tile: a row past T_b - 1 takes the last real step's operands)
#pragma unroll
        for (int mt = 0; mt < DX_MT; ++mt) offA[mt] = (unsigned)(src_step(16 * mt + (tid >> 4)) * L + 4 * (tid & 15));
        auto row_ok = [&](int m) -> bool { const int tt = tt0 + m; return tt >= 1 && tt < Tb; };
        if (U > 0) {
            const u32 want = p.tag_base + (u32)c + 1u;
            int ready_hi = -1;
            auto ensure_ready = [&](int V) {                      // as dag_dp_dense_mfma.hip: one vector poll of the next 64 progress words
                if (V <= ready_hi) return;
                if (wave == 0) {
                    u32 spins = 0;
                    for (;;) {
                        const int vq = ready_hi + 1 + lane;
                        const u32 pv = (vq < U) ? __hip_atomic_load(prog + vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (want - 1u);
                        const u64 okm = __ballot((int)(pv - want) >= 0);
                        const int npref = (~okm) ? (int)__builtin_ctzll(~okm) : 64;
                        if (ready_hi + npref >= V) { if (lane == 0) RDY[0] = ready_hi + npref; break; }
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > DX_SPIN_LIMIT) { if (lane == 0) { atomicOr(&p.counters[1], 4u); RDY[0] = U; } break; }
                    }
                }
                __syncthreads();
                ready_hi = RDY[0];
                __syncthreads();
            };
            int Vmin = 0;
            { const int lim = ub - TR - DX_BW; if (lim >= 0) Vmin = lim / DX_BW + 1; }
            // ... that can hold a live vertex: at step tt nothing left of column tt is reachable (dag_dp_dense_mfma.hip, same rule)
            if (tt0 >= 1) Vmin = max(Vmin, (tt0 - 1) / DX_BW);
            // Two register stages of RAW loaded words (block maximum of row tid % 16, 4 alpha_max values of row tid / 16, 16 weights),
            // unconditional requests, validity recomputed at conversion time, two LDS buffers and one barrier per source block, readiness
            // waited for only at need — the pipeline of dag_dp_dense_mfma.hip (its header explains each point).
            float st_s[2], st_a[2][DX_MT][4], st_w[2][16];
            bool st_ok[2];
            auto w_full = [&](int V) -> bool { return min(ub + 63, L - 1) - 1 - V * DX_BW < TR; };            // largest distance of the pair
            auto prefetchW = [&](int s, int V) {
                if (w_full(V)) {
                    const float* Kv = K + (size_t)(V * DX_BW) * (size_t)(TR - 1);
#pragma unroll
                    for (int i = 0; i < 16; ++i) st_w[s][i] = Kv[offW[i]];
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) st_w[s][i] = K[w_idx(V * DX_BW + 16 * mg + i, ub + n)];
                }
            };
            auto prefetchA = [&](int s, int V) {
                const int vb = V * DX_BW;
                if (chunk_full) {
                    st_s[s] = dx_ld(S + offS + V);
                    const float* Ov = O + vb;
#pragma unroll
                    for (int mt = 0; mt < DX_MT; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) st_a[s][mt][e] = dx_ld(Ov + offA[mt] + e);
                    return;
                }
                {
                    const int m = tid & (DX_TM - 1);
                    st_s[s] = dx_ld(&S[row_ok(m) ? (unsigned)((tt0 + m - 1) * NJ + V) : 0u]);
                }
#pragma unroll
                for (int mt = 0; mt < DX_MT; ++mt) {
                    const int m = 16 * mt + (tid >> 4), q4 = tid & 15;       // A: row m, source columns 4 q4 .. +3
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = vb + 4 * q4 + e;
                        st_a[s][mt][e] = dx_ld(O + ((row_ok(m) && v < L) ? (unsigned)((tt0 + m - 1) * L + v) : 0u));
                    }
                }
            };
            auto stage_live = [&](int s) -> bool { return __any(row_ok(tid & (DX_TM - 1)) && st_s[s] != NEG_INF); };   // lanes 0 .. DX_TM-1 of every wave: all rows
            auto commit = [&](int s, int V, int nb) {
                const int vb = V * DX_BW;
                if (tid < DX_TM) Sb[nb * DX_TM + tid] = row_ok(tid) ? st_s[s] : NEG_INF;
#pragma unroll
                for (int mt = 0; mt < DX_MT; ++mt) {
                    const int m = 16 * mt + (tid >> 4), q4 = tid & 15;
                    v4f a4;
                    if (chunk_full) {                                    // (source columns of a block V < U are always inside the graph)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4[e] = st_a[s][mt][e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4[e] = (row_ok(m) && (vb + 4 * q4 + e) < L) ? st_a[s][mt][e] : NEG_INF;
                    }
                    *reinterpret_cast<v4f*>(At + nb * DX_TM * 64 + m * 64 + 4 * q4) = a4;
                }
                float* Wb = Wt + nb * 64 * DX_WP;
                if (w_full(V)) {                                         // full tile: no predicates at all
#pragma unroll
                    for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f*>(Wb + n * DX_WP + 16 * mg + 4 * i) = (v4f){st_w[s][4 * i], st_w[s][4 * i + 1], st_w[s][4 * i + 2], st_w[s][4 * i + 3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) Wb[n * DX_WP + 16 * mg + i] = w_ok(vb + 16 * mg + i, ub + n) ? st_w[s][i] : NEG_INF;
                }
            };
            auto product = [&](int nb, int Vb) {       // (+, max) product: acc[mt][r] = max_k ( A[16 mt + 4 mg + r][k] + W[k][n] )
                const float* Ab = At + nb * DX_TM * 64; const float* Wb = Wt + nb * 64 * DX_WP;
                float old[DX_MT][4];
#pragma unroll
                for (int mt = 0; mt < DX_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[mt][r] = acc[mt][r];
#pragma unroll 2
                for (int kk = 0; kk < 16; ++kk) {
                    const v4f w4 = *reinterpret_cast<const v4f*>(Wb + n * DX_WP + 4 * kk);
                    const float w0 = w4.x, w1 = w4.y, w2 = w4.z, w3 = w4.w;
#pragma unroll
                    for (int mt = 0; mt < DX_MT; ++mt) {
                        v4f a4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) a4[r] = *reinterpret_cast<const v4f*>(Ab + (16 * mt + 4 * mg + r) * 64 + 4 * kk);      // broadcast
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[mt][r] = fmaxf(fmaxf(acc[mt][r], a4[r].x + w0), a4[r].y + w1);
                            acc[mt][r] = fmaxf(fmaxf(acc[mt][r], a4[r].z + w2), a4[r].w + w3);
                        }
                    }
                }
#pragma unroll
                for (int mt = 0; mt < DX_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[mt][r] = (acc[mt][r] > old[mt][r]) ? Vb : bv[mt][r];
            };
            int cur = 0, curV = -1; bool have = false;
            auto step = [&](int s, int V) {
                if (V < U && !st_ok[s]) {
                    ensure_ready(V); prefetchA(s, V);
                    asm volatile("" :: "v"(st_a[s][DX_MT - 1][3]), "v"(st_a[s][0][0]), "v"(st_s[s]) : "memory");      // the wait for the re-request stays in this branch
                }
                const bool live = V < U && stage_live(s);
                if (live) commit(s, V, cur ^ 1);
                const int W = V + 2, Wc = min(W, U - 1);
                prefetchW(s, Wc); prefetchA(s, Wc);
                st_ok[s] = W <= ready_hi;
                if (have) product(cur, curV);
                if (have || live) __syncthreads();
                if (live) { cur ^= 1; curV = V; }
                have = live;
            };
            if (Vmin < U) {
                const int W0 = Vmin, W1 = min(Vmin + 1, U - 1);
                prefetchW(0, W0); ensure_ready(Vmin); prefetchA(0, W0); st_ok[0] = W0 <= ready_hi;
                prefetchW(1, W1); prefetchA(1, W1); st_ok[1] = (Vmin + 1) <= ready_hi;
            }
            for (int Vb = Vmin; Vb < U; Vb += 2) { step(0, Vb); step(1, Vb + 1); }
            if (have) product(cur, curV);
        }
#pragma unroll
        for (int mt = 0; mt < DX_MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { Poff[(16 * mt + 4 * mg + r) * 64 + n] = acc[mt][r]; Boff[(16 * mt + 4 * mg + r) * 64 + n] = bv[mt][r]; }
            *reinterpret_cast<v4f*>(Md + (16 * mt + (tid >> 4)) * 64 + 4 * (tid & 15)) = (v4f){em[mt][0], em[mt][1], em[mt][2], em[mt][3]};
        }
        __syncthreads();

        // ================================================================ diagonal block, row by row (wave 0)
        if (wave == 0) {
            const int m_lo = (tt0 == 0) ? 1 : 0, m_hi = min(DX_TM, Tb - tt0);
            float n_po = Poff[m_lo * 64 + ul], n_m = Md[m_lo * 64 + ul]; int n_bo = Boff[m_lo * 64 + ul];
#pragma unroll 1
            for (int m = m_lo; m < m_hi; ++m) {
                const int tt = tt0 + m;
                float best = n_po; const float mm = n_m; const float po = n_po; int blk = n_bo;
                { const int mn = min(m + 1, DX_TM - 1); n_po = Poff[mn * 64 + ul]; n_m = Md[mn * 64 + ul]; n_bo = Boff[mn * 64 + ul]; }       // next row's operands
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const v4f t4 = *reinterpret_cast<const v4f*>(Vd + 4 * q);
                    best = fmaxf(fmaxf(best, t4.x + Wcol[4 * q]), t4.y + Wcol[4 * q + 1]);
                    best = fmaxf(fmaxf(best, t4.z + Wcol[4 * q + 2]), t4.w + Wcol[4 * q + 3]);
                }
                // cells outside t <= j < L_b have no live predecessor / only -inf links: -inf by the arithmetic alone
                const float a = best + mm;                                     // mx + match   (dag_best_alignment.cu:120)
                if (u < L) dx_st(O + (size_t)tt * L + u, a);
                if (best > po) blk = U;                                       // (strictly better than every earlier block: the arg-max is in this one)
                if (u < L) BT[(size_t)tt * L + u] = (unsigned short)blk;
                Vd[ul] = a;                                                   // (this row's reads are done: same wave, program order)
                const float bm = dx_wave_max(a);
                if (lane == 0) dx_st(&S[(size_t)tt * NJ + U], bm);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog + U, p.tag_base + (u32)c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

// Two builds of the same body: 256 VGPRs (31 spilled) so that two workgroups share a CU — the faster one when there are more workgroups
// than CUs (C2 at TR = 4095: 22.8 -> 18.2 ms) — and the unconstrained one for launches that put at most one workgroup on a CU anyway
// (C1: 2.18 vs 2.25 ms).
template <int DX_MT> __global__ __launch_bounds__(256) void dag_dense_max_kernel(DXParams p) { dag_dense_max_body<DX_MT>(p); }
template <int DX_MT> __global__ __launch_bounds__(256, 2) void dag_dense_max_kernel_occ2(DXParams p) { dag_dense_max_body<DX_MT>(p); }

// K7: path[b][pos] = t along the chain of arg-max predecessors from (T_b-1, L_b-1), with the block trace of the max-DP (r03): the arg-max predecessor of cell (t, pos) lies in block V = btrace[t][pos], so a hop
// evaluates 64 candidates — one wave, one gather of 64 transition weights — instead of every predecessor.  The rows the hops will
// need (alpha_max[t-1][*] and btrace[t][*]) do not depend on the path: the other waves stream them into an LDS ring NR rows ahead, so
// that the only memory round trip on the hop chain is the gather of the weights.  Tie rule as before: smallest predecessor index.
// ring = 0: no LDS ring (L too large for it), the rows are read from memory.
__global__ __launch_bounds__(256) void dag_dense_backtrace_blk_kernel(const float* __restrict__ alpha, const unsigned short* __restrict__ btrace,
                                                                      const float* __restrict__ links, const int64_t* __restrict__ out_len,
                                                                      const int64_t* __restrict__ tgt_len, int64_t* __restrict__ path,
                                                                      int B, int T, int L, int TR, int ring)
{
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    int* lp = reinterpret_cast<int*>(bsm);                                    // [L] path image
    float* Ar = reinterpret_cast<float*>(lp + L);                             // [ring][L]  alpha_max rows   (row r in slot r % ring)
    unsigned short* Br = reinterpret_cast<unsigned short*>(Ar + (size_t)ring * L);      // [ring][L]  block-trace rows
    __shared__ int s_pos[2];                                                  // hop t reads slot t & 1, its result goes to slot (t - 1) & 1: no wave
                                                                              // can see the next position before the hop's barrier (every wave takes the same exit)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < L; j += 256) lp[j] = -1;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (tid == 0) { s_pos[0] = s_pos[1] = valid ? Lb - 1 : -1; }
    const float* A = alpha + (size_t)b * T * L;
    const unsigned short* BT = btrace + (size_t)b * T * L;
    const float* K = links + (size_t)b * L * TR;
    // hop t (from row t to row t - 1) reads btrace row t and alpha row t - 1: "row pair t"
    auto load_pair = [&](int t, int first, int nthr) {
        if (t < 1) return;
        float* ar = Ar + (size_t)((t - 1) % ring) * L; unsigned short* br = Br + (size_t)(t % ring) * L;
        for (int j = tid - first; j < L; j += nthr) { ar[j] = A[(size_t)(t - 1) * L + j]; br[j] = BT[(size_t)t * L + j]; }
    };
    // the loader waves (1..3) fetch a pair into REGISTERS during one hop and park it in the ring during the next, so that their memory
    // round trip never sits between the hop wave and the hop's barrier
    constexpr int NS = 48;                                                        // ceil(L / 192) for every L that gets a ring
    float ra[NS]; unsigned short rb[NS]; int rq = 0;
    const int nsa = (L + 191) / 192;
    auto issue = [&](int q) {
        rq = q;
        if (q < 1) return;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (s2 >= nsa) break;                                                 // (uniform: the unrolled tail is skipped, not predicated)
            const int j = min((tid - 64) + 192 * s2, L - 1);
            ra[s2] = A[(size_t)(q - 1) * L + j]; rb[s2] = BT[(size_t)q * L + j];
        }
    };
    auto stash = [&]() {
        if (rq < 1) return;
        float* ar = Ar + (size_t)((rq - 1) % ring) * L; unsigned short* br = Br + (size_t)(rq % ring) * L;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            if (s2 >= nsa) break;
            const int j = (tid - 64) + 192 * s2;
            if (j < L) { ar[j] = ra[s2]; br[j] = rb[s2]; }
        }
    };
    if (valid && ring > 0) {
        for (int k = 0; k < ring - 1; ++k) load_pair(Tb - 1 - k, 0, 256);
        if (wave != 0) issue(Tb - ring);
    }
    __syncthreads();
    if (valid) {
        for (int t = Tb - 1; t >= 0; --t) {
            const int pos = s_pos[t & 1];
            if (pos < 0) break;
            if (tid == 0) lp[pos] = t;
            if (t == 0) break;
            if (ring > 0 && wave != 0) { stash(); issue(t - ring); }         // (waves 1..3) park pair t - ring + 1 (its slots were pair t + 1's: that hop is done), request the next
            if (wave == 0) {
                const int V = ring > 0 ? (int)Br[(size_t)(t % ring) * L + pos] : (int)BT[(size_t)t * L + pos];
                const int lo = max(t - 1, pos - TR);
                const int i = V * 64 + lane;
                const bool ok = V != 0xFFFF && i < pos && i >= lo;
                const float av = ring > 0 ? Ar[(size_t)((t - 1) % ring) * L + (ok ? i : 0)] : A[(size_t)(t - 1) * L + (ok ? i : 0)];
                const float kv = K[ok ? ((size_t)i * TR + (pos - i - 1)) : (size_t)0];
                float best = ok ? av + kv : NEG_INF; int arg = ok ? i : (1 << 30);
                if (best == NEG_INF) arg = 1 << 30;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float b2 = __shfl_xor(best, o, 64); const int a2 = __shfl_xor(arg, o, 64);
                    if (b2 > best || (b2 == best && a2 < arg)) { best = b2; arg = a2; }
                }
                if (lane == 0) s_pos[(t - 1) & 1] = (best == NEG_INF) ? -1 : arg;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int j = tid; j < L; j += 256) path[(size_t)b * L + j] = lp[j];
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

static thread_local int g_dx_mt = 0;                   // diagnostic switch (dsp_dag_set_option "dx_mt"): 0 = auto, 1 / 2 = 16- / 32-row chunks
void set_dx_mt(int v) { g_dx_mt = (v == 1 || v == 2) ? v : 0; }

bool dense_max_supported(int L, int TR) { return TR > 32 && L >= 128 && (size_t)L * 4 <= 150 * 1024 && (long)L * TR < (1L << 31); }

static int launch_dense_backtrace(const float* alpha_max, const unsigned short* btrace, const float* links, const int64_t* out_len,
                                  const int64_t* tgt_len, int64_t* path, int B, int T, int L, int TR, hipStream_t st);

// block_trace != NULL: the caller keeps the [B,T,L] block trace (the two halves of the alignment run as separate calls: Viterbi graph decode
// chooses the length between them) and path may be NULL (no back-trace here); NULL: the trace lives in the launch workspace.
int launch_dag_dense_max(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                         float* alpha_max, int64_t* path, int B, int T, int L, int TR, hipStream_t st, unsigned short* block_trace)
{
    const int NJ = (L + DX_BW - 1) / DX_BW;
    DXParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha_max;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NJ = NJ;
    const size_t prog_bytes = ((size_t)B * NJ * sizeof(u32) + 255) / 256 * 256;
    const size_t s_bytes = ((size_t)B * T * NJ * sizeof(float) + 255) / 256 * 256;
    const size_t bt_bytes = block_trace ? 0 : (size_t)B * T * L * sizeof(unsigned short);
    u64* area = nullptr;
    int rc = banded_acquire_ws(st, prog_bytes + s_bytes + bt_bytes, T, &p.counters, &area, &p.tag_base);
    if (rc) return rc;
    p.progress = reinterpret_cast<u32*>(area);
    p.S = reinterpret_cast<float*>(reinterpret_cast<char*>(area) + prog_bytes);
    p.btrace = block_trace ? block_trace : reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(area) + prog_bytes + s_bytes);
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    // chunk height (r05 sweep, ms at 16 / 32 rows — C2 at TR = 4095: 13.5 / 12.4; B=8 L=4096 T=512: 5.95 / 6.17; C1: 1.61 / 1.81; B=16 L=1024 T=150:
    // 0.70 / 0.79; B=32 L=400 T=100: 0.33 / 0.38; paths bit-identical): 32 rows only pay when the launch holds many rounds of workgroups — then
    // half the passes over the transition matrix (68.7 -> 34.4 GB at C2) outweigh the longer (chunk, block) wavefront; dx_mt pins it
    const bool big = B * NJ > ncu;
    const int mt = g_dx_mt ? g_dx_mt : (B * NJ >= 6 * ncu) ? 2 : 1;
    const int TM = 16 * mt;
    const size_t lds = (size_t)(2 * TM * 64 + 2 * 64 * DX_WP + 2 * TM + TM * 64 + 64 + TM * 64 + 4 + TM * 64) * 4 + 64;
    auto k = mt == 2 ? (big ? dag_dense_max_kernel_occ2<2> : dag_dense_max_kernel<2>) : (big ? dag_dense_max_kernel_occ2<1> : dag_dense_max_kernel<1>);
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(B * NJ)), dim3(256), lds, st, p);
    rc = check_launch("dag_best_alignment(dense max-plus)");
    if (rc || !path) return rc;
    return launch_dense_backtrace(alpha_max, p.btrace, links, out_len, tgt_len, path, B, T, L, TR, st);
}

static int launch_dense_backtrace(const float* alpha_max, const unsigned short* btrace, const float* links, const int64_t* out_len,
                                  const int64_t* tgt_len, int64_t* path, int B, int T, int L, int TR, hipStream_t st)
{
    // LDS ring of (alpha row, block-trace row) pairs for the back-trace: as deep as fits (5 at L = 4096), none beyond L ~ 9000
    int ring = 5;
    while (ring > 1 && (size_t)L * 4 + (size_t)ring * L * 6 > 150 * 1024) --ring;
    if (ring < 3 || L > 48 * 192) ring = 0;
    const size_t lds2 = (size_t)L * 4 + (size_t)ring * L * 6;
    set_max_dynamic_lds((const void*)dag_dense_backtrace_blk_kernel, (int)lds2);
    hipLaunchKernelGGL(dag_dense_backtrace_blk_kernel, dim3((unsigned)B), dim3(256), lds2, st, alpha_max, btrace, links, out_len, tgt_len, path, B, T, L, TR, ring);
    return check_launch("dag_best_alignment(dense back-trace)");
}

// the second half alone: tgt_len may differ from the one the max-DP ran with (any row it filled)
int launch_dag_dense_backtrace(const float* alpha_max, const unsigned short* block_trace, const float* links, const int64_t* out_len,
                               const int64_t* tgt_len, int64_t* path, int B, int T, int L, int TR, hipStream_t st)
{
    return launch_dense_backtrace(alpha_max, block_trace, links, out_len, tgt_len, path, B, T, L, TR, st);
}

}  // namespace dsp
