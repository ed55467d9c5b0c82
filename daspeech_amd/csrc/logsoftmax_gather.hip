// logsoftmax_gather.hip — K1 and its backward for gfx950.
//
// What it replaces: DASpeech/custom_ops/logsoftmax_gather.cu:256-377 (forward, in-place softmax side effect)
// and the Python backward in DASpeech/custom_ops/dag_loss.py:293-295.
//
// Design (HBM-bound op; SURVEY.md §8d: algorithmic bytes = B*L*V*s_in [+ the same again when the softmax is
// stored] + B*S*L*4):
//   * one 256-thread workgroup walks a tile of RT consecutive vertices (rows of V logits);
//   * pass A reads the row ONCE from HBM with 16-byte loads keeping an online (max, sum) per lane, wave shuffles
//     + one LDS hop combine them; the gather and the optional softmax store re-read the row while it is still
//     L2-resident (a row is <= a few tens of KB), so HBM sees one read and at most one write per logit;
//   * gathered values are staged in LDS as [S][RT] so that the [B,S,L] ("match_all") layout is written in
//     RT-float contiguous runs instead of 4-byte scatters — the reference wrote [B,L,S] and paid a transpose copy
//     (nat_dag_loss.py:128 + dag_loss.py:103).
#include "common.h"
#include <stdlib.h>

namespace dsp {

// streaming accesses of the row kernels.  NT = non-temporal.  The in-place BACKWARD reads a row and overwrites it: with both the loads
// and the stores non-temporal it runs 1.70 -> 1.59 ms at C2 fp32 (either alone: no change; r01f sweep, same box, same run).
typedef unsigned int lsg_u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 lsg_ld16(const void* p)
{
    if constexpr (NT) {
        const lsg_u4 v = __builtin_nontemporal_load(reinterpret_cast<const lsg_u4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void lsg_st16(void* p, const uint4& o)
{
    if constexpr (NT) {
        lsg_u4 v; v.x = o.x; v.y = o.y; v.z = o.z; v.w = o.w;
        __builtin_nontemporal_store(v, reinterpret_cast<lsg_u4*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = o;
    }
}


template <typename T> struct Vec;                 // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<__half> { static constexpr int N = 8; };
template <> struct Vec<__hip_bfloat16> { static constexpr int N = 8; };

template <typename T>
__device__ __forceinline__ void load16(const T* p, float (&f)[Vec<T>::N]) {
    uint4 raw = *reinterpret_cast<const uint4*>(p);
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < Vec<T>::N; ++i) f[i] = to_f(e[i]);
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const float (&f)[Vec<T>::N]) {
    uint4 raw;
    T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
    for (int i = 0; i < Vec<T>::N; ++i) e[i] = from_f<T>(f[i]);
    *reinterpret_cast<uint4*>(p) = raw;
}

// Rows that do not start on a 16-byte boundary (a vocabulary that is not a multiple of 4 fp32 / 8 half elements — fairseq pads its
// dictionaries to multiples of 8, an unpadded one is legal): up to N-1 head elements and up to N-1 tail elements go one by one, the body in
// between is 16-byte aligned and moves as vectors (r05: the all-scalar form ran at 1.1-1.8 TB/s against 3.5 for aligned rows).
template <typename T> struct Peel {
    int head, nb, tail0, nscalar;
    __device__ __forceinline__ Peel(const T* row, int V) {
        constexpr int N = Vec<T>::N;
        const int h = (int)(((16 - (int)((uintptr_t)row & 15)) & 15) / (int)sizeof(T));
        head = h < V ? h : V; nb = (V - head) / N; tail0 = head + nb * N; nscalar = head + (V - tail0);
    }
    __device__ __forceinline__ int scalar_index(int e) const { return e < head ? e : tail0 + (e - head); }
};

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    float nm = fmaxf(m, m2);
    if (nm == NEG_INF) { s = 0.f; m = nm; return; }
    s = s * __expf(m - nm) + s2 * __expf(m2 - nm);
    m = nm;
}

// block-wide (max, sum-exp) of one row; result broadcast to every thread. red = 2*8 floats of LDS.
template <typename T, bool VEC>
__device__ __forceinline__ void row_max_sum(const T* row, int V, float* red, float& m_out, float& s_out) {
    constexpr int N = Vec<T>::N;
    float m = NEG_INF, s = 0.f;
    if (VEC) {
        for (int v = threadIdx.x * N; v < V; v += blockDim.x * N) {
            float f[N];
            load16(row + v, f);
            float lm = f[0];
#pragma unroll
            for (int i = 1; i < N; ++i) lm = fmaxf(lm, f[i]);
            float nm = fmaxf(m, lm);
            if (nm != NEG_INF) {
                float acc = s * __expf(m - nm);
#pragma unroll
                for (int i = 0; i < N; ++i) acc += __expf(f[i] - nm);
                s = acc;
            }
            m = nm;
        }
    } else {
        const Peel<T> pl(row, V);
        for (int e = threadIdx.x; e < pl.nscalar; e += blockDim.x) {
            float x = to_f(row[pl.scalar_index(e)]);
            float nm = fmaxf(m, x);
            if (nm != NEG_INF) s = s * __expf(m - nm) + __expf(x - nm);
            m = nm;
        }
        for (int i = threadIdx.x; i < pl.nb; i += blockDim.x) {
            float f[N];
            load16(row + pl.head + i * N, f);
            float lm = f[0];
#pragma unroll
            for (int q = 1; q < N; ++q) lm = fmaxf(lm, f[q]);
            float nm = fmaxf(m, lm);
            if (nm != NEG_INF) {
                float acc = s * __expf(m - nm);
#pragma unroll
                for (int q = 0; q < N; ++q) acc += __expf(f[q] - nm);
                s = acc;
            }
            m = nm;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();                          // red[] free (previous row's readers are done)
    if ((threadIdx.x & 63) == 0) { red[wave] = m; red[8 + wave] = s; }
    __syncthreads();
    m = red[0]; s = red[8];
    for (int w = 1; w < nw; ++w) online_merge(m, s, red[w], red[8 + w]);
    m_out = m; s_out = s;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void lsg_fwd_kernel(
    T* __restrict__ x, const int64_t* __restrict__ idx, int64_t isb, int64_t isj, int64_t iss,
    float* __restrict__ out, int64_t osb, int64_t osj, int64_t oss,
    int B, int L, int V, int S, int RT, int write_softmax, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                // 16 floats
    float* stage = smem + 16;         // [S][RT]
    constexpr int N = Vec<T>::N;
    const int tiles_per_b = (L + RT - 1) / RT;
    const long ntiles = (long)B * tiles_per_b;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_b);
        const int j0 = (int)(tile % tiles_per_b) * RT;
        const int nr = min(RT, L - j0);
        for (int r = 0; r < nr; ++r) {
            T* row = x + ((size_t)b * L + (j0 + r)) * V;
            float m, s;
            row_max_sum<T, VEC>(row, V, red, m, s);
            const float ls = __logf(s);
            if (stats && threadIdx.x == 0) { float* st2 = stats + 2 * ((size_t)b * L + (j0 + r)); st2[0] = m; st2[1] = 1.f / s; }
            for (int k = threadIdx.x; k < S; k += blockDim.x) {
                int64_t t = idx[b * isb + (int64_t)(j0 + r) * isj + k * iss];
                t = t < 0 ? 0 : (t >= V ? V - 1 : t);
                stage[k * RT + r] = (to_f(row[t]) - m) - ls;        // logsoftmax_gather.cu:293
            }
            if (write_softmax) {
                __syncthreads();                                      // gathers read the ORIGINAL logits
                const float inv = 1.f / s;
                if (VEC) {
                    for (int v = threadIdx.x * N; v < V; v += blockDim.x * N) {
                        float f[N];
                        load16(row + v, f);
#pragma unroll
                        for (int i = 0; i < N; ++i) f[i] = __expf(f[i] - m) * inv;
                        store16(row + v, f);
                    }
                } else {
                    const Peel<T> pl(row, V);
                    for (int e = threadIdx.x; e < pl.nscalar; e += blockDim.x) { const int v = pl.scalar_index(e); row[v] = from_f<T>(__expf(to_f(row[v]) - m) * inv); }
                    for (int i = threadIdx.x; i < pl.nb; i += blockDim.x) {
                        float f[N];
                        load16(row + pl.head + i * N, f);
#pragma unroll
                        for (int q = 0; q < N; ++q) f[q] = __expf(f[q] - m) * inv;
                        store16(row + pl.head + i * N, f);
                    }
                }
            }
        }
        __syncthreads();
        const int tot = S * nr;
        if (osj == 1 || oss != 1) {          // runs along the vertex axis: [B,S,L] layout
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                int k = e / nr, r = e - k * nr;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        } else {                             // runs along S: the reference's [B,L,S] layout
            for (int e = threadIdx.x; e < tot; e += blockDim.x) {
                int r = e / S, k = e - r * S;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        }
        __syncthreads();
    }
}

// Register-resident forward: the whole row (<= NV x 256 x 16 bytes) is read ONCE into registers with 16-byte loads, the
// next row of the tile is prefetched while the current one is reduced, and the softmax is stored straight from the
// registers — no second pass over the logits (the generic kernel above re-reads the row from L2 for the store).
template <typename T, int NV>
__global__ __launch_bounds__(256) void lsg_fwd_reg_kernel(
    T* __restrict__ x, const int64_t* __restrict__ idx, int64_t isb, int64_t isj, int64_t iss,
    float* __restrict__ out, int64_t osb, int64_t osj, int64_t oss,
    int B, int L, int V, int S, int RT, int write_softmax, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                // 2 x 16 floats (alternating per row)
    float* stage = smem + 32;         // [S][RT]
    constexpr int N = Vec<T>::N;
    const int tid = threadIdx.x;
    const int tiles_per_b = (L + RT - 1) / RT;
    const long ntiles = (long)B * tiles_per_b;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_b);
        const int j0 = (int)(tile % tiles_per_b) * RT;
        const int nr = min(RT, L - j0);
        uint4 cur[NV], nxt[NV];
        auto load_row = [&](int r, uint4 (&dst)[NV]) {
            const T* row = x + ((size_t)b * L + (j0 + r)) * V;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                if (v < V) dst[k] = lsg_ld16<false>(row + v);
            }
        };
        // The S gathered logits of a row are requested TOGETHER WITH the row itself (one iteration before they are used): the
        // row's lines are then in flight or freshly in L2 and the 4-byte gathers merge with them.  Requested an iteration
        // later (when the row is reduced) ~40 % of them missed — a streaming workgroup's 64 KB per row turns the XCD's 4 MB
        // L2 over in about one row time — and each miss was a 64-byte HBM fetch for 4 bytes (+1.7 GB of FETCH_SIZE at C2).
        float gnx[8], graw[8];
        auto load_gather = [&](int r, float (&dst)[8]) {
            const T* row = x + ((size_t)b * L + (j0 + r)) * V;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = tid + u * 256;
                dst[u] = 0.f;
                if (k < S) {
                    int64_t t = idx[b * isb + (int64_t)(j0 + r) * isj + k * iss];
                    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
                    dst[u] = to_f(row[t]);
                }
            }
        };
        // rows (and their gathers) are requested TWO iterations ahead: a workgroup's requests come in 32 KB bursts, and with one
        // row ahead the bytes in flight per CU average below what the HBM latency-bandwidth product asks for
        uint4 nx2[NV]; float gn2[8];
        load_row(0, cur);
        load_gather(0, graw);
        if (nr > 1) { load_row(1, nxt); load_gather(1, gnx); }
        for (int r = 0; r < nr; ++r) {
            T* row = x + ((size_t)b * L + (j0 + r)) * V;
            if (r + 2 < nr) { load_row(r + 2, nx2); load_gather(r + 2, gn2); }
            float f[NV][N];
            float m = NEG_INF;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                const T* e = reinterpret_cast<const T*>(&cur[k]);
#pragma unroll
                for (int i = 0; i < N; ++i) { f[k][i] = (v < V) ? to_f(e[i]) : NEG_INF; m = fmaxf(m, f[k][i]); }
            }
            float s = 0.f;
            if (m != NEG_INF) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int i = 0; i < N; ++i) s += __expf(f[k][i] - m);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
                online_merge(m, s, m2, s2);
            }
            // ONE workgroup barrier per row: the cross-wave slots alternate (no "slots free" barrier), and every thread's
            // gathers of this row have LANDED before it (they read the ORIGINAL logits; the softmax stores follow the barrier)
            float* rs = red + ((r & 1) << 4);
            if ((tid & 63) == 0) { rs[tid >> 6] = m; rs[8 + (tid >> 6)] = s; }
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(graw[u]));
            __syncthreads();
            m = rs[0]; s = rs[8];
#pragma unroll
            for (int w = 1; w < 4; ++w) online_merge(m, s, rs[w], rs[8 + w]);
            const float ls = __logf(s);
            if (stats && tid == 0) { float* st2 = stats + 2 * ((size_t)b * L + (j0 + r)); st2[0] = m; st2[1] = 1.f / s; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = tid + u * 256;
                if (k < S) stage[k * RT + r] = (graw[u] - m) - ls;
            }
            if (write_softmax) {
                const float inv = 1.f / s;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int v = (k * 256 + tid) * N;
                    if (v < V) {
                        uint4 o;
                        T* e = reinterpret_cast<T*>(&o);
#pragma unroll
                        for (int i = 0; i < N; ++i) e[i] = from_f<T>(__expf(f[k][i] - m) * inv);
                        lsg_st16<false>(row + v, o);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) { cur[k] = nxt[k]; nxt[k] = nx2[k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { graw[u] = gnx[u]; gnx[u] = gn2[u]; }
        }
        __syncthreads();
        const int tot = S * nr;
        if (osj == 1 || oss != 1) {
            for (int e = tid; e < tot; e += 256) {
                int k = e / nr, r = e - k * nr;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        } else {
            for (int e = tid; e < tot; e += 256) {
                int r = e / S, k = e - r * S;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        }
        __syncthreads();
    }
}

// Register-resident forward, gathers from LDS.  lsg_fwd_reg_kernel takes the S gathered logits of a row from global memory and
// therefore needs the row's lines to stay in L2 (temporal loads).  Here the raw row is also parked in LDS (V * sizeof(T) bytes) and
// the gathers read it there, so the row stream can be non-temporal both ways like the backward's (C2 fp32: 1.75 -> 1.62 ms with the softmax store, 1.13 -> 1.02 ms
// without; bf16 1.19 -> 1.05 ms), the
// gather costs no L2 traffic at all, and the two-rows-ahead gather registers are gone.  Two workgroup barriers per row: the row
// image is single (a second 32 KB image would halve the occupancy).
// DEEP = true: two rows requested ahead (three row images in registers: NV <= 8); false: one row ahead — the NV = 16 instance (r05: rows of up to
// 16 384 fp32 / 32 768 half-precision logits stay on this kernel instead of the two-pass generic one: 3.5 -> 5 TB/s)
template <typename T, int NV, bool DEEP = true>
__global__ __launch_bounds__(256) void lsg_fwd_regl_kernel(
    T* __restrict__ x, const int64_t* __restrict__ idx, int64_t isb, int64_t isj, int64_t iss,
    float* __restrict__ out, int64_t osb, int64_t osj, int64_t oss,
    int B, int L, int V, int S, int RT, int write_softmax, float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                // 2 x 16 floats (alternating per row)
    float* stage = smem + 32;         // [S][RT]
    T* rowbuf = reinterpret_cast<T*>(stage + (size_t)S * RT);     // [V] raw logits of the current row
    constexpr int N = Vec<T>::N;
    const int tid = threadIdx.x;
    const int tiles_per_b = (L + RT - 1) / RT;
    const long ntiles = (long)B * tiles_per_b;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_b);
        const int j0 = (int)(tile % tiles_per_b) * RT;
        const int nr = min(RT, L - j0);
        uint4 cur[NV], nxt[NV], nx2[DEEP ? NV : 1];
        auto load_row = [&](int r, uint4 (&dst)[NV]) {
            const T* row = x + ((size_t)b * L + (j0 + r)) * V;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                if (v < V) dst[k] = lsg_ld16<true>(row + v);
            }
        };
        int tk[8], tkn[8];
        auto load_tok = [&](int r, int (&dst)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = tid + u * 256;
                dst[u] = 0;
                if (k < S) {
                    const int64_t t = idx[b * isb + (int64_t)(j0 + r) * isj + k * iss];
                    dst[u] = (int)(t < 0 ? 0 : (t >= V ? V - 1 : t));
                }
            }
        };
        load_row(0, cur); load_tok(0, tk);
        if (DEEP && nr > 1) { load_row(1, nxt); load_tok(1, tkn); }
        for (int r = 0; r < nr; ++r) {
            T* row = x + ((size_t)b * L + (j0 + r)) * V;
            if constexpr (DEEP) { if (r + 2 < nr) load_row(r + 2, nx2); }
            else { if (r + 1 < nr) { load_row(r + 1, nxt); load_tok(r + 1, tkn); } }
            float f[NV][N];
            float m = NEG_INF;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                const T* e = reinterpret_cast<const T*>(&cur[k]);
#pragma unroll
                for (int i = 0; i < N; ++i) { f[k][i] = (v < V) ? to_f(e[i]) : NEG_INF; m = fmaxf(m, f[k][i]); }
            }
            float s = 0.f;
            if (m != NEG_INF) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int i = 0; i < N; ++i) s += __expf(f[k][i] - m);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
                online_merge(m, s, m2, s2);
            }
            __syncthreads();                                       // every wave has finished gathering the previous row from rowbuf
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                if (v < V) *reinterpret_cast<uint4*>(rowbuf + v) = cur[k];
            }
            float* rs = red + ((r & 1) << 4);
            if ((tid & 63) == 0) { rs[tid >> 6] = m; rs[8 + (tid >> 6)] = s; }
            __syncthreads();
            m = rs[0]; s = rs[8];
#pragma unroll
            for (int w = 1; w < 4; ++w) online_merge(m, s, rs[w], rs[8 + w]);
            const float ls = __logf(s);
            if (stats && tid == 0) { float* st2 = stats + 2 * ((size_t)b * L + (j0 + r)); st2[0] = m; st2[1] = 1.f / s; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = tid + u * 256;
                if (k < S) stage[k * RT + r] = (to_f(rowbuf[tk[u]]) - m) - ls;
            }
            if (write_softmax) {
                const float inv = 1.f / s;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int v = (k * 256 + tid) * N;
                    if (v < V) {
                        uint4 o;
                        T* e = reinterpret_cast<T*>(&o);
#pragma unroll
                        for (int i = 0; i < N; ++i) e[i] = from_f<T>(__expf(f[k][i] - m) * inv);
                        lsg_st16<true>(row + v, o);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) { cur[k] = nxt[k]; if constexpr (DEEP) nxt[k] = nx2[k]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) tk[u] = tkn[u];
            if constexpr (DEEP) { if (r + 2 < nr) load_tok(r + 2, tkn); }
        }
        __syncthreads();
        const int tot = S * nr;
        if (osj == 1 || oss != 1) {
            for (int e = tid; e < tot; e += 256) {
                int k = e / nr, r = e - k * nr;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        } else {
            for (int e = tid; e < tot; e += 256) {
                int r = e / S, k = e - r * S;
                out[b * osb + (int64_t)(j0 + r) * osj + k * oss] = stage[k * RT + r];
            }
        }
        __syncthreads();
    }
}

// backward: row <- softmax * (-(sum_s g)) + scatter_add(g)     (dag_loss.py:293-295)
// The per-row scatter targets are accumulated in an LDS image of the row (ds_add_f32), so duplicates add up
// exactly like scatter_add_ and the row is still written once.
template <typename T, bool VEC, bool LAZY>
__global__ __launch_bounds__(256) void lsg_bwd_kernel(
    T* __restrict__ x, const int64_t* __restrict__ idx, int64_t isb, int64_t isj, int64_t iss,
    const float* __restrict__ g, int64_t gsb, int64_t gsj, int64_t gss,
    int B, int L, int V, int S, const float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;            // 16 floats
    float* delta = smem + 16;     // [V]
    constexpr int N = Vec<T>::N;
    for (int v = threadIdx.x; v < V; v += blockDim.x) delta[v] = 0.f;
    __syncthreads();
    const long nrows = (long)B * L;
    for (long rowi = blockIdx.x; rowi < nrows; rowi += gridDim.x) {
        const int b = (int)(rowi / L), j = (int)(rowi % L);
        T* row = x + (size_t)rowi * V;
        float gs = 0.f;
        for (int k = threadIdx.x; k < S; k += blockDim.x) {
            float gv = g[b * gsb + (int64_t)j * gsj + k * gss];
            int64_t t = idx[b * isb + (int64_t)j * isj + k * iss];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            gs += gv;
            atomicAdd(&delta[t], gv);
        }
        gs = wave_sum(gs);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
        const float neg = -tot;
        // LAZY: the buffer still holds the LOGITS; softmax = exp(x - m) * inv from the forward's row statistics
        const float rm = LAZY ? stats[2 * rowi] : 0.f, rinv = LAZY ? stats[2 * rowi + 1] : 1.f;
        if (VEC) {
            for (int v = threadIdx.x * N; v < V; v += blockDim.x * N) {
                float f[N];
                load16(row + v, f);
#pragma unroll
                for (int i = 0; i < N; ++i) f[i] = (LAZY ? __expf(f[i] - rm) * rinv : f[i]) * neg + delta[v + i];
                store16(row + v, f);
            }
        } else {
            const Peel<T> pl(row, V);
            for (int e = threadIdx.x; e < pl.nscalar; e += blockDim.x) {
                const int v = pl.scalar_index(e);
                const float xv = to_f(row[v]);
                row[v] = from_f<T>((LAZY ? __expf(xv - rm) * rinv : xv) * neg + delta[v]);
            }
            for (int i0 = threadIdx.x; i0 < pl.nb; i0 += blockDim.x) {
                const int v = pl.head + i0 * N;
                float f[N];
                load16(row + v, f);
#pragma unroll
                for (int i = 0; i < N; ++i) f[i] = (LAZY ? __expf(f[i] - rm) * rinv : f[i]) * neg + delta[v + i];
                store16(row + v, f);
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < S; k += blockDim.x) {       // re-zero only what was touched
            int64_t t = idx[b * isb + (int64_t)j * isj + k * iss];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            delta[t] = 0.f;
        }
        __syncthreads();
    }
}

// Backward, tiled: a workgroup owns RT CONSECUTIVE rows.  The row-per-workgroup kernel above reads g[b][s][j] with a stride of
// L floats between the S gradients of a row, and the 16 rows that share each 64-byte line of g are in the hands of 16
// different workgroups on 8 XCDs: 0.27 GB of gradients cost 1.2 GB of FETCH_SIZE at C2.  Here the [S][RT] gradient tile is
// staged through LDS with 4*RT-byte runs along j, the softmax row is register-resident with the next row prefetched (as in
// lsg_fwd_reg_kernel), and the per-row scatter image in LDS is unchanged.
template <typename T, int NV, bool LAZY, int NTM = 3, bool DEEP = true>       // NTM: bit 0 non-temporal row loads, bit 1 non-temporal row stores; DEEP: two rows ahead (false: one — the wide-row instances, r05)
__global__ __launch_bounds__(256) void lsg_bwd_reg_kernel(
    T* __restrict__ x, const int64_t* __restrict__ idx, int64_t isb, int64_t isj, int64_t iss,
    const float* __restrict__ g, int64_t gsb, int64_t gsj, int64_t gss,
    int B, int L, int V, int S, int RT, const float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;                // 16 floats
    float* delta = smem + 16;         // [V]   scatter image of the current row
    float* gt = delta + V;            // [S][RT]
    constexpr int N = Vec<T>::N;
    const int tid = threadIdx.x;
    for (int v = tid; v < V; v += 256) delta[v] = 0.f;
    const int tiles_per_b = (L + RT - 1) / RT;
    const long ntiles = (long)B * tiles_per_b;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_b);
        const int j0 = (int)(tile % tiles_per_b) * RT;
        const int nr = min(RT, L - j0);
        uint4 cur[NV], nxt[NV];
        auto load_row = [&](int r, uint4 (&dst)[NV]) {
            const T* row = x + ((size_t)b * L + (j0 + r)) * V;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                if (v < V) dst[k] = lsg_ld16<(NTM & 1) != 0>(row + v);
            }
        };
        uint4 nx2[DEEP ? NV : 1];
        load_row(0, cur);
        if (DEEP && nr > 1) load_row(1, nxt);
        __syncthreads();                                   // previous tile's gt / delta use is over
        const int tot = S * nr;
        if (gsj == 1) {                                    // [B][S][L] gradients: runs along j
            for (int e = tid; e < tot; e += 256) { const int k = e / nr, r = e - k * nr; gt[k * RT + r] = g[b * gsb + (int64_t)(j0 + r) + k * gss]; }
        } else {
            for (int e = tid; e < tot; e += 256) { const int r = e / S, k = e - r * S; gt[k * RT + r] = g[b * gsb + (int64_t)(j0 + r) * gsj + k * gss]; }
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            T* row = x + ((size_t)b * L + (j0 + r)) * V;
            if constexpr (DEEP) { if (r + 2 < nr) load_row(r + 2, nx2); }          // two rows ahead (see lsg_fwd_reg_kernel)
            else { if (r + 1 < nr) load_row(r + 1, nxt); }
            float gs = 0.f;
            for (int k = tid; k < S; k += 256) {
                const float gv = gt[k * RT + r];
                int64_t t = idx[b * isb + (int64_t)(j0 + r) * isj + k * iss];
                t = t < 0 ? 0 : (t >= V ? V - 1 : t);
                gs += gv;
                atomicAdd(&delta[t], gv);
            }
            gs = wave_sum(gs);
            if ((tid & 63) == 0) red[tid >> 6] = gs;
            __syncthreads();
            const float neg = -(red[0] + red[1] + red[2] + red[3]);
            // LAZY: `cur` holds the LOGITS; softmax = exp(x - m) * inv from the forward's row statistics
            const size_t srow = (size_t)b * L + (j0 + r);
            const float rm = LAZY ? stats[2 * srow] : 0.f, rinv = LAZY ? stats[2 * srow + 1] : 1.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int v = (k * 256 + tid) * N;
                if (v < V) {
                    const T* e = reinterpret_cast<const T*>(&cur[k]);
                    uint4 o;
                    T* eo = reinterpret_cast<T*>(&o);
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        eo[i] = from_f<T>((LAZY ? __expf(to_f(e[i]) - rm) * rinv : to_f(e[i])) * neg + delta[v + i]);     // dag_loss.py:293-295
                    lsg_st16<(NTM & 2) != 0>(row + v, o);
                }
            }
            __syncthreads();
            for (int k = tid; k < S; k += 256) {           // re-zero only what was touched
                int64_t t = idx[b * isb + (int64_t)(j0 + r) * isj + k * iss];
                t = t < 0 ? 0 : (t >= V ? V - 1 : t);
                delta[t] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) { cur[k] = nxt[k]; if constexpr (DEEP) nxt[k] = nx2[k]; }
            __syncthreads();
        }
    }
}

template <typename T>
static int launch_fwd(void* logits, const int64_t* idx, int64_t isb, int64_t isj, int64_t iss, float* match,
                      int64_t osb, int64_t osj, int64_t oss, int B, int L, int V, int S, int ws, hipStream_t st, float* stats = nullptr)
{
    constexpr int N = Vec<T>::N;
    const bool vec = (V % N == 0) && ((uintptr_t)logits % 16 == 0);
    int RT = 16;
    while (RT > 1 && (size_t)S * RT * 4 > 60 * 1024) RT >>= 1;
    if ((size_t)S * RT * 4 > 150 * 1024) { set_error("logsoftmax_gather: S=%d too large for LDS staging", S); return DSP_EINVAL; }
    if (RT > L) { RT = 1; while (RT * 2 <= L) RT *= 2; }
    const size_t lds = (16 + (size_t)S * RT) * sizeof(float);
    const long ntiles = (long)B * ((L + RT - 1) / RT);
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    // register-resident variant when the row fits NV x 256 sixteen-byte vectors
    const int nvec = (V + 256 * N - 1) / (256 * N);
    if (vec && sizeof(T) == 4 && nvec > 8 && nvec <= 16 && S <= 8 * 256) {      // (half precision: 128 elements per lane — the generic kernel is faster, 3.6 vs 1.9-2.3 TB/s)
        // wide rows (8 192 < V <= 16 384 fp32, 16 384 < V <= 32 768 half precision): the LDS-gather kernel with 16 vectors per lane and one row ahead
        int RTg = 16;
        while (RTg > 1 && (size_t)V * sizeof(T) + (32 + (size_t)S * RTg) * 4 > 78 * 1024) RTg >>= 1;
        const size_t ldsg = (size_t)V * sizeof(T) + (32 + (size_t)S * RTg) * 4;
        if (ldsg <= 78 * 1024 && L >= RTg && (S * RTg) % 4 == 0) {
            auto kg = nvec <= 10 ? lsg_fwd_regl_kernel<T, 10, false> : nvec <= 12 ? lsg_fwd_regl_kernel<T, 12, false> : nvec <= 14 ? lsg_fwd_regl_kernel<T, 14, false>
                                                                                                                       : lsg_fwd_regl_kernel<T, 16, false>;
            const long nt = (long)B * ((L + RTg - 1) / RTg);
            const int gridg = (int)(nt < 4096 ? nt : 4096);
            set_max_dynamic_lds((const void*)kg, (int)ldsg);
            hipLaunchKernelGGL(kg, dim3(gridg), dim3(256), ldsg, st, (T*)logits, idx, isb, isj, iss, match, osb, osj, oss, B, L, V, S, RTg, ws, stats);
            return check_launch("logsoftmax_gather(reg, LDS gather, wide rows)");
        }
    }
    if (vec && nvec <= 8 && S <= 8 * 256) {
        auto kr = nvec <= 2 ? lsg_fwd_reg_kernel<T, 2> : (nvec <= 4 ? lsg_fwd_reg_kernel<T, 4> : nvec <= 6 ? lsg_fwd_reg_kernel<T, 6> : lsg_fwd_reg_kernel<T, 8>);
        {
            // LDS-gather variant: row image V * sizeof(T) + stage [S][RTg] must leave two workgroups per CU
            static int gl = -1;
            if (gl < 0) { const char* e = getenv("DSP_K1_GL"); gl = e ? atoi(e) : 1; }          // DSP_K1_GL=0: the global-gather kernel
            int RTg = 16;
            while (RTg > 1 && (size_t)V * sizeof(T) + (32 + (size_t)S * RTg) * 4 > 78 * 1024) RTg >>= 1;
            const size_t ldsg = (size_t)V * sizeof(T) + (32 + (size_t)S * RTg) * 4;
            if (gl && ldsg <= 78 * 1024 && L >= RTg && (S * RTg) % 4 == 0) {
                auto kg = nvec <= 2 ? lsg_fwd_regl_kernel<T, 2> : (nvec <= 4 ? lsg_fwd_regl_kernel<T, 4> : nvec <= 6 ? lsg_fwd_regl_kernel<T, 6> : lsg_fwd_regl_kernel<T, 8>);
                const long nt = (long)B * ((L + RTg - 1) / RTg);
                int gridg = (int)(nt < 4096 ? nt : 4096);
                static const char* const e_grid = getenv("DSP_K1_GRID");        // (tuning switches: read once per process)
                if (e_grid) gridg = atoi(e_grid);
                if (ldsg > 48 * 1024) set_max_dynamic_lds((const void*)kg, (int)ldsg);
                hipLaunchKernelGGL(kg, dim3(gridg), dim3(256), ldsg, st, (T*)logits, idx, isb, isj, iss, match, osb, osj, oss,
                                   B, L, V, S, RTg, ws, stats);
                return check_launch("logsoftmax_gather(reg, LDS gather)");
            }
        }
        int RTr = RT, gridr = grid;
        // storing the softmax makes the launch a 1:1 read/write stream: two resident workgroups per CU (64 KB of stage per
        // workgroup at RT = 32) sustain 4.9 TB/s, four only 4.1 TB/s (sweep in tools/k1_bench.py, r01)
        if (ws && (size_t)S * 32 * 4 <= 96 * 1024 && L >= 32) { RTr = 32; const long nt = (long)B * ((L + 31) / 32); gridr = (int)(nt < 4096 ? nt : 4096); }
        static const char* const e_rt = getenv("DSP_K1_RT"); static const char* const e_grid2 = getenv("DSP_K1_GRID");
        if (e_rt) { RTr = atoi(e_rt); const long nt = (long)B * ((L + RTr - 1) / RTr); gridr = (int)(nt < 65535 * 4 ? nt : 65535 * 4); if (e_grid2) gridr = atoi(e_grid2); }
        const size_t ldsr = (32 + (size_t)S * RTr) * sizeof(float);
        if (ldsr > 48 * 1024) set_max_dynamic_lds((const void*)kr, (int)ldsr);
        hipLaunchKernelGGL(kr, dim3(gridr), dim3(256), ldsr, st, (T*)logits, idx, isb, isj, iss, match, osb, osj, oss,
                           B, L, V, S, RTr, ws, stats);
        return check_launch("logsoftmax_gather(reg)");
    }
    auto k = vec ? lsg_fwd_kernel<T, true> : lsg_fwd_kernel<T, false>;
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (T*)logits, idx, isb, isj, iss, match, osb, osj, oss,
                       B, L, V, S, RT, ws, stats);
    return check_launch("logsoftmax_gather");
}

template <typename T, bool LAZY>
static int launch_bwd(void* sm, const int64_t* idx, int64_t isb, int64_t isj, int64_t iss, const float* g,
                      int64_t gsb, int64_t gsj, int64_t gss, int B, int L, int V, int S, hipStream_t st, const float* stats)
{
    constexpr int N = Vec<T>::N;
    const bool vec = (V % N == 0) && ((uintptr_t)sm % 16 == 0);
    size_t lds = (16 + (size_t)V) * sizeof(float);
    if (lds > 160 * 1024) { set_error("logsoftmax_gather_bwd: V=%d exceeds the LDS row image (max ~40k)", V); return DSP_EINVAL; }
    { static const char* const e = getenv("DSP_K1B_LDS"); if (e) { const size_t want = (size_t)atoi(e); if (want > lds) lds = want; } }
    const long nrows = (long)B * L;
    const int nvec = (V + 256 * N - 1) / (256 * N);
    static const char* const e_old = getenv("DSP_K1B_OLD");
    if (vec && sizeof(T) == 4 && nvec > 8 && nvec <= 16 && !e_old) {  // wide rows (8 192 < V <= 16 384 fp32): 10-16 vectors per lane, one row ahead (r05)
        int RT = 16;
        while (RT > 1 && (16 + (size_t)V + (size_t)S * RT) * sizeof(float) > 76 * 1024) RT >>= 1;
        const size_t ldsr = (16 + (size_t)V + (size_t)S * RT) * sizeof(float);
        if (ldsr <= 76 * 1024 && L >= RT) {
            auto kr = nvec <= 10 ? lsg_bwd_reg_kernel<T, 10, LAZY, 3, false> : nvec <= 12 ? lsg_bwd_reg_kernel<T, 12, LAZY, 3, false>
                    : nvec <= 14 ? lsg_bwd_reg_kernel<T, 14, LAZY, 3, false> : lsg_bwd_reg_kernel<T, 16, LAZY, 3, false>;
            const long nt = (long)B * ((L + RT - 1) / RT);
            const int gridr = (int)(nt < 4096 ? nt : 4096);
            set_max_dynamic_lds((const void*)kr, (int)ldsr);
            hipLaunchKernelGGL(kr, dim3(gridr), dim3(256), ldsr, st, (T*)sm, idx, isb, isj, iss, g, gsb, gsj, gss, B, L, V, S, RT, stats);
            return check_launch("logsoftmax_gather_bwd(reg, wide rows)");
        }
    }
    {
        int RT = 16;
        { static const char* const e = getenv("DSP_K1B_RT"); if (e) RT = atoi(e); }
        const size_t ldsr = (16 + (size_t)V + (size_t)S * RT) * sizeof(float);
        if (vec && nvec <= 8 && ldsr <= 76 * 1024 && L >= RT && !e_old) {     // two workgroups per CU
            auto kr = nvec <= 2 ? lsg_bwd_reg_kernel<T, 2, LAZY> : (nvec <= 4 ? lsg_bwd_reg_kernel<T, 4, LAZY> : nvec <= 6 ? lsg_bwd_reg_kernel<T, 6, LAZY> : lsg_bwd_reg_kernel<T, 8, LAZY>);
            const long nt = (long)B * ((L + RT - 1) / RT);
            int gridr = (int)(nt < 4096 ? nt : 4096);
            { static const char* const e = getenv("DSP_K1B_GRID"); if (e) gridr = atoi(e); }
            if (ldsr > 48 * 1024) set_max_dynamic_lds((const void*)kr, (int)ldsr);
            hipLaunchKernelGGL(kr, dim3(gridr), dim3(256), ldsr, st, (T*)sm, idx, isb, isj, iss, g, gsb, gsj, gss, B, L, V, S, RT, stats);
            return check_launch("logsoftmax_gather_bwd(reg)");
        }
    }
    const int grid = (int)(nrows < 2048 ? nrows : 2048);
    auto k = vec ? lsg_bwd_kernel<T, true, LAZY> : lsg_bwd_kernel<T, false, LAZY>;
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (T*)sm, idx, isb, isj, iss, g, gsb, gsj, gss, B, L, V, S, stats);
    return check_launch("logsoftmax_gather_bwd");
}

}  // namespace dsp

extern "C" int dsp_logsoftmax_gather(void* logits, int dtype, const int64_t* idx, int64_t idx_sb, int64_t idx_sj,
                                     int64_t idx_ss, float* match, int64_t out_sb, int64_t out_sj, int64_t out_ss,
                                     int B, int L, int V, int S, int write_softmax, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 0 || V <= 0 || S < 0) { set_error("logsoftmax_gather: bad sizes B=%d L=%d V=%d S=%d", B, L, V, S); return DSP_EINVAL; }
    if (B == 0 || L == 0) return DSP_OK;
    if (!logits || (S > 0 && (!idx || !match))) { set_error("logsoftmax_gather: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case DSP_F32: return launch_fwd<float>(logits, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, write_softmax, st);
        case DSP_F16: return launch_fwd<__half>(logits, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, write_softmax, st);
        case DSP_BF16: return launch_fwd<__hip_bfloat16>(logits, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, write_softmax, st);
    }
    set_error("logsoftmax_gather: unsupported dtype code %d", dtype);
    return DSP_EINVAL;
}

extern "C" int dsp_logsoftmax_gather_bwd(void* softmax_inout, int dtype, const int64_t* idx, int64_t idx_sb,
                                         int64_t idx_sj, int64_t idx_ss, const float* g, int64_t g_sb, int64_t g_sj,
                                         int64_t g_ss, int B, int L, int V, int S, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 0 || V <= 0 || S < 0) { set_error("logsoftmax_gather_bwd: bad sizes"); return DSP_EINVAL; }
    if (B == 0 || L == 0) return DSP_OK;
    if (!softmax_inout || (S > 0 && (!idx || !g))) { set_error("logsoftmax_gather_bwd: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case DSP_F32: return launch_bwd<float, false>(softmax_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, nullptr);
        case DSP_F16: return launch_bwd<__half, false>(softmax_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, nullptr);
        case DSP_BF16: return launch_bwd<__hip_bfloat16, false>(softmax_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, nullptr);
    }
    set_error("logsoftmax_gather_bwd: unsupported dtype code %d", dtype);
    return DSP_EINVAL;
}

// ---- "lazy" pair: the logits are NOT overwritten in the forward pass.  The reference stores the softmax in place only as
// backward state ("word_ins_out is modified in place for storing backward tensors.  DO NOT use word_ins_out after this
// function", dag_loss.py:249-251); the same gradient follows from the logits and two floats per row (max, 1/sum-exp), which
// saves the forward's B*L*V store (4.3 GB of 8.9 GB at C2) — the backward reads the logits instead of the softmax.
extern "C" int dsp_logsoftmax_gather_stats(const void* logits, int dtype, const int64_t* idx, int64_t idx_sb, int64_t idx_sj,
                                           int64_t idx_ss, float* match, int64_t out_sb, int64_t out_sj, int64_t out_ss,
                                           float* row_stats, int B, int L, int V, int S, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 0 || V <= 0 || S < 0) { set_error("logsoftmax_gather_stats: bad sizes B=%d L=%d V=%d S=%d", B, L, V, S); return DSP_EINVAL; }
    if (B == 0 || L == 0) return DSP_OK;
    if (!logits || !row_stats || (S > 0 && (!idx || !match))) { set_error("logsoftmax_gather_stats: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    void* lg = const_cast<void*>(logits);
    switch (dtype) {
        case DSP_F32: return launch_fwd<float>(lg, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, 0, st, row_stats);
        case DSP_F16: return launch_fwd<__half>(lg, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, 0, st, row_stats);
        case DSP_BF16: return launch_fwd<__hip_bfloat16>(lg, idx, idx_sb, idx_sj, idx_ss, match, out_sb, out_sj, out_ss, B, L, V, S, 0, st, row_stats);
    }
    set_error("logsoftmax_gather_stats: unsupported dtype code %d", dtype);
    return DSP_EINVAL;
}

extern "C" int dsp_logsoftmax_gather_bwd_lazy(void* logits_inout, int dtype, const int64_t* idx, int64_t idx_sb,
                                              int64_t idx_sj, int64_t idx_ss, const float* g, int64_t g_sb, int64_t g_sj,
                                              int64_t g_ss, const float* row_stats, int B, int L, int V, int S, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 0 || V <= 0 || S < 0) { set_error("logsoftmax_gather_bwd_lazy: bad sizes"); return DSP_EINVAL; }
    if (B == 0 || L == 0) return DSP_OK;
    if (!logits_inout || !row_stats || (S > 0 && (!idx || !g))) { set_error("logsoftmax_gather_bwd_lazy: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case DSP_F32: return launch_bwd<float, true>(logits_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, row_stats);
        case DSP_F16: return launch_bwd<__half, true>(logits_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, row_stats);
        case DSP_BF16: return launch_bwd<__hip_bfloat16, true>(logits_inout, idx, idx_sb, idx_sj, idx_ss, g, g_sb, g_sj, g_ss, B, L, V, S, st, row_stats);
    }
    set_error("logsoftmax_gather_bwd_lazy: unsupported dtype code %d", dtype);
    return DSP_EINVAL;
}
