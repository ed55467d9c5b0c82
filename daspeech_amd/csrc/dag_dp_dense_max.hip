// dag_dp_dense_max.hip — DENSE-window (TR > 64) max-DP of dag_best_alignment (K6) as a blocked max-plus product, plus a trace-free
// back-trace (K7).
//
// Replaces, for the fused alignment op, dag_dense_kernel<1> of dag_dp_generic.hip (row-sequential: every DP row re-reads the whole
// transition matrix — C1: 4.7 ms, C2 at TR = 4095: 119 ms) and, like dag_dp_maxstrip.hip does for banded graphs, the B*T*L int32 trace
// tensor of the reference (dag_best_alignment.cu:39-130 keeps the arg-max of every cell; the back-trace reads T of them).
//
// Same decomposition as dag_dp_dense_mfma.hip: 64-column blocks, chunks of 16 rows, block U runs one chunk behind block U-1 (progress
// words tagged with the launch epoch, tickets block-major).  The off-diagonal part of a tile is a [16 x 64] (+, max) [64 x 64] product —
// there is no matrix-core form of it, so it runs on the VALU: a thread owns one column and four rows, the source rows are LDS
// broadcasts, the weights LDS reads at unit stride.  No exponents, no guard: add and max are exact, so alpha_max is bit-identical to
// the sequential scan, whatever the blocking.  The diagonal block is walked row by row by one wave (column per lane, its weights in
// registers, max3).
// The back-trace recomputes the arg-max of the T cells it visits from alpha_max and the links (smallest predecessor index among equal
// maxima: the torch rule): one workgroup per sample scans the up-to-L predecessors of a cell cooperatively.
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
};

constexpr int DX_BW = 64, DX_TM = 16, DX_WP = 65;      // weight tile pitch 65: a thread column walks k at a bank stride of 1
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
    return fmaxf(fmaxf(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                 fmaxf(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ float dx_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dx_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void dag_dense_max_kernel(DXParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ u32 s_ticket;
    float* At = smem;                                  // [16][64]   source rows (previous DP row of block V), row-major
    float* Wt = At + DX_TM * 64;                       // [64][65]   weights [k = source][n = column]
    float* Sb = Wt + 64 * DX_WP;                       // [16]       block maximum per source row
    float* Poff = Sb + DX_TM;                          // [16][64]   off-diagonal maxima of the tile
    float* Vd = Poff + DX_TM * 64;                     // [64]       diagonal block: previous row
    float* Md = Vd + 64;                               // [16][64]   the chunk's emissions
    int* RDY = reinterpret_cast<int*>(Md + DX_TM * 64);
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
    u32* prog = p.progress + (size_t)b * NJ;
    const int ub = U * DX_BW;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    for (int t = valid ? Tb : 0; t < T; ++t)
        for (int ul = tid; ul < DX_BW; ul += 256) { const int u = ub + ul; if (u < L) O[(size_t)t * L + u] = NEG_INF; }
    if (!valid) return;
    // weight of the transition v -> u (v < u): links[v][u-v-1]; unconditional load at a clamped address, masked afterwards
    auto wraw = [&](int v, int u) -> float {
        const int d = u - v - 1;
        const bool ok = !(d < 0 || d >= TR || u >= L || v < 0);
        const float raw = K[ok ? ((size_t)v * TR + d) : (size_t)0];
        return ok ? raw : NEG_INF;
    };
    const int nchunks = (Tb + DX_TM - 1) / DX_TM;

    // ---- diagonal-block state of wave 0 (lane = column) and the seed row
    const int ul = lane, u = ub + lane;
    float Wcol[64];
    float aprev = NEG_INF;
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 64; ++i) Wcol[i] = (i < ul) ? wraw(ub + i, u) : NEG_INF;
        aprev = (u == 0) ? M[0] : NEG_INF;                                 // alpha_max[0,0] = match[0,0]   (dag_best_alignment.cu:72-74)
        if (u < L) dx_st(O + u, aprev);
        Vd[ul] = aprev;
        const float bm = dx_wave_max(aprev);
        if (lane == 0) dx_st(&S[U], bm);
    }
    __syncthreads();

    const int n = tid & 63, mg = tid >> 6;             // product phase: column n of the block, rows 4 mg .. 4 mg + 3 of the chunk
    for (int c = 0; c < nchunks; ++c) {
        const int tt0 = c * DX_TM;
        float acc[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
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
            float st_s = NEG_INF, st_a[4], st_w[16];
            auto prefetch = [&](int V) {
                const int vb = V * DX_BW;
                if (tid < DX_TM) {
                    const int tt = tt0 + tid;
                    const bool ok = tt >= 1 && tt < Tb;
                    const float sx = dx_ld(&S[ok ? ((size_t)(tt - 1) * NJ + V) : (size_t)0]);
                    st_s = ok ? sx : NEG_INF;
                }
                {
                    const int m = tid >> 4, q4 = tid & 15;                   // A: row m, source columns 4 q4 .. +3
                    const int tt = tt0 + m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = vb + 4 * q4 + e;
                        const bool ok = tt >= 1 && tt < Tb && v < L;
                        const float raw = dx_ld(O + (ok ? ((size_t)(tt - 1) * L + v) : (size_t)0));
                        st_a[e] = ok ? raw : NEG_INF;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {                             // W: source row k = (tid >> 6) * 16 + 4 it + e, column n = tid & 63
#pragma unroll
                    for (int e = 0; e < 4; ++e) st_w[4 * it + e] = wraw(vb + (tid >> 6) * 16 + 4 * it + e, ub + (tid & 63));
                }
            };
            if (Vmin < U) { ensure_ready(Vmin); prefetch(Vmin); }
            for (int V = Vmin; V < U; ++V) {
                if (tid < DX_TM) Sb[tid] = st_s;
                __syncthreads();
                bool any_live = false;
#pragma unroll
                for (int m = 0; m < DX_TM; ++m) any_live |= (Sb[m] != NEG_INF);
                if (any_live) {
                    const int m = tid >> 4, q4 = tid & 15;
                    *reinterpret_cast<v4f*>(At + m * 64 + 4 * q4) = (v4f){st_a[0], st_a[1], st_a[2], st_a[3]};
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int e = 0; e < 4; ++e) Wt[((tid >> 6) * 16 + 4 * it + e) * DX_WP + (tid & 63)] = st_w[4 * it + e];
                }
                __syncthreads();
                if (V + 1 < U) { ensure_ready(V + 1); prefetch(V + 1); }
                if (any_live) {
                    // (+, max) product: acc[r] = max_k ( A[4 mg + r][k] + W[k][n] )
#pragma unroll 4
                    for (int kk = 0; kk < 16; ++kk) {
                        v4f a4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) a4[r] = *reinterpret_cast<const v4f*>(At + (4 * mg + r) * 64 + 4 * kk);      // broadcast
                        const float w0 = Wt[(4 * kk) * DX_WP + n], w1 = Wt[(4 * kk + 1) * DX_WP + n];
                        const float w2 = Wt[(4 * kk + 2) * DX_WP + n], w3 = Wt[(4 * kk + 3) * DX_WP + n];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[r] = fmaxf(fmaxf(acc[r], a4[r].x + w0), a4[r].y + w1);
                            acc[r] = fmaxf(fmaxf(acc[r], a4[r].z + w2), a4[r].w + w3);
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Poff[(4 * mg + r) * 64 + n] = acc[r];
        __syncthreads();

        // ================================================================ diagonal block, row by row (wave 0)
        if (wave == 0) {
            {
                float mrow[DX_TM];
#pragma unroll
                for (int m = 0; m < DX_TM; ++m) { const int tt = tt0 + m; mrow[m] = (tt >= 1 && tt < Tb && u < L) ? M[(size_t)tt * L + u] : NEG_INF; }
#pragma unroll
                for (int m = 0; m < DX_TM; ++m) Md[m * 64 + ul] = mrow[m];
            }
#pragma unroll 1
            for (int m = 0; m < DX_TM; ++m) {
                const int tt = tt0 + m;
                if (tt == 0) continue;
                if (tt >= Tb) break;
                float best = Poff[m * 64 + ul];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const v4f t4 = *reinterpret_cast<const v4f*>(Vd + 4 * q);
                    best = fmaxf(fmaxf(best, t4.x + Wcol[4 * q]), t4.y + Wcol[4 * q + 1]);
                    best = fmaxf(fmaxf(best, t4.z + Wcol[4 * q + 2]), t4.w + Wcol[4 * q + 3]);
                }
                // cells outside t <= j < L_b have no live predecessor / only -inf links: -inf by the arithmetic alone
                const float a = best + Md[m * 64 + ul];                       // mx + match   (dag_best_alignment.cu:120)
                if (u < L) dx_st(O + (size_t)tt * L + u, a);
                aprev = a;
                Vd[ul] = a;                                                   // (this row's reads are done: same wave, program order)
                const float bm = dx_wave_max(a);
                if (lane == 0) dx_st(&S[(size_t)tt * NJ + U], bm);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(prog + U, p.tag_base + (u32)c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    (void)aprev;
}

// K7 without a trace tensor: path[b][pos] = t along the chain of arg-max predecessors from (T_b-1, L_b-1), the arg-max recomputed from
// alpha_max and the links for the one cell per row the chain visits.  Tie rule: smallest predecessor index among equal maxima.
__global__ __launch_bounds__(256) void dag_dense_backtrace_kernel(const float* __restrict__ alpha, const float* __restrict__ links,
                                                                  const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
                                                                  int64_t* __restrict__ path, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) int lp[];          // [L] path image
    __shared__ float rv[4]; __shared__ int ri[4]; __shared__ int s_pos;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < L; j += 256) lp[j] = -1;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (tid == 0) s_pos = valid ? Lb - 1 : -1;
    __syncthreads();
    if (valid) {
        const float* A = alpha + (size_t)b * T * L;
        const float* K = links + (size_t)b * L * TR;
        // the final score must be finite, otherwise there is no alignment (the reference asserts, dag_best_alignment.cu:117-119)
        for (int t = Tb - 1; t >= 0; --t) {
            const int pos = s_pos;
            if (pos < 0) break;
            if (tid == 0) lp[pos] = t;
            if (t == 0) break;
            const int lo = max(t - 1, pos - TR);
            float best = NEG_INF; int arg = 1 << 30;
            for (int i = lo + tid; i < pos; i += 256) {                   // ascending i per thread, strict >: the thread's smallest index
                const float v = A[(size_t)(t - 1) * L + i] + K[(size_t)i * TR + (pos - i - 1)];
                if (v > best) { best = v; arg = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float b2 = __shfl_xor(best, o, 64); const int a2 = __shfl_xor(arg, o, 64);
                if (b2 > best || (b2 == best && a2 < arg)) { best = b2; arg = a2; }
            }
            if (lane == 0) { rv[wave] = best; ri[wave] = arg; }
            __syncthreads();
            if (tid == 0) {
                float bb = rv[0]; int aa = ri[0];
                for (int w = 1; w < 4; ++w) if (rv[w] > bb || (rv[w] == bb && ri[w] < aa)) { bb = rv[w]; aa = ri[w]; }
                s_pos = (bb == NEG_INF) ? -1 : aa;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int j = tid; j < L; j += 256) path[(size_t)b * L + j] = lp[j];
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool dense_max_supported(int L, int TR) { return TR > 64 && L >= 128 && (size_t)L * 4 <= 150 * 1024; }

int launch_dag_dense_max(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                         float* alpha_max, int64_t* path, int B, int T, int L, int TR, hipStream_t st)
{
    const int NJ = (L + DX_BW - 1) / DX_BW;
    DXParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha_max;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NJ = NJ;
    const size_t prog_bytes = ((size_t)B * NJ * sizeof(u32) + 255) / 256 * 256;
    const size_t s_bytes = (size_t)B * T * NJ * sizeof(float);
    u64* area = nullptr;
    int rc = banded_acquire_ws(st, prog_bytes + s_bytes, T, &p.counters, &area, &p.tag_base);
    if (rc) return rc;
    p.progress = reinterpret_cast<u32*>(area);
    p.S = reinterpret_cast<float*>(reinterpret_cast<char*>(area) + prog_bytes);
    const size_t lds = (size_t)(DX_TM * 64 + 64 * DX_WP + DX_TM + DX_TM * 64 + 64 + DX_TM * 64 + 4) * 4 + 64;
    (void)hipFuncSetAttribute((const void*)dag_dense_max_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dag_dense_max_kernel, dim3((unsigned)(B * NJ)), dim3(256), lds, st, p);
    rc = check_launch("dag_best_alignment(dense max-plus)");
    if (rc) return rc;
    const size_t lds2 = (size_t)L * 4;
    (void)hipFuncSetAttribute((const void*)dag_dense_backtrace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(dag_dense_backtrace_kernel, dim3((unsigned)B), dim3(256), lds2, st, alpha_max, links, out_len, tgt_len, path, B, T, L, TR);
    return check_launch("dag_best_alignment(dense back-trace)");
}

}  // namespace dsp
