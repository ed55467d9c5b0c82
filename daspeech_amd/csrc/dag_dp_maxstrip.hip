// dag_dp_maxstrip.hip — dag_best_alignment (K6 + K7) for banded graphs (TR <= 32) WITHOUT a trace tensor.
//
// Replaces calculate_maxalpha_kernel + the host-driven back-trace (DASpeech/custom_ops/dag_best_alignment.cu:39-130,160-201).
// The reference (and dag_dp_strip2.hip / the generic kernel here) keep, for every one of the B*T*L cells, the arg-max
// predecessor: 4 VALU instructions per transition (add, compare, two selects) and a B*T*L int32 tensor written to HBM, of
// which the back-trace then reads T entries per sample.  Here:
//   * the DP keeps VALUES only: per vertex 32 adds and a 3-input max tree (1.5 instructions per transition, no trace store);
//     same strip / tagged-granule / ticket / helper-wave structure as dag_dp_strip4g.hip, 4 vertices per lane, log domain
//     (add / max only => alpha_max is bit-identical to the sequential scan);
//   * the back-trace recomputes the arg-max for the one cell per row it visits, from alpha_max and the links, with the
//     reference's tie rule (smallest predecessor index; -1 when every candidate is -inf).
// Used when the caller passes trace == NULL (the Python operator does); with a trace pointer the eager kernels run.
#include "common.h"
#include <string.h>
#include <stdlib.h>
// Cycle accounting / timing ablations of the max-DP (tools/prof_maxstrip.py) exist only in a build with -DDSP_MX_PROF (add it as a
// `// HIPCC_FLAGS:` line here to reproduce profiles/r03h): even wave-uniform run-time checks of the switches cost the row loop 4 %.
#ifdef DSP_MX_PROF
#define MX_DBG(p) ((p).dbg)
#else
#define MX_DBG(p) 0
#endif

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;

struct MStripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS;
    int ldm, ldo;                             // row pitch (elements) of match / alpha_max: >= L rounded up to 4, multiples of 4 (r06)
    int dbg;                                  // DSP_DEBUG=prof (2): cycle accounting of one compute wave (counters[8..12]); DSP_MX_ABLATE bits 4 / 8 / 16: no alpha
                                              // store / no max trees / no adds either (timing experiments, results wrong)
};

constexpr int MX_TRP = 32;
constexpr int MX_RING = 8;
constexpr int MX_CH = 4;                      // halo prefetch distance of the fetch wave (rows)
constexpr u32 MX_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ u64 mx_gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mx_gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mx_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

typedef float mx_v4f __attribute__((ext_vector_type(4)));
typedef float mx_v2f __attribute__((ext_vector_type(2)));

template <int NT, int CPL>
__device__ __forceinline__ void maxstrip_body(const MStripParams& p, char* smem_raw, int b, int s, int so)
{
    constexpr int W = CPL * NT, RL = W + 32, NCW = NT / 64, DPR = W / 256;
    float* Abuf = reinterpret_cast<float*>(smem_raw);          // [2][RL]  alpha_max rows (natural log domain)
    float* Mring = Abuf + 2 * RL;                              // [RING][W] match rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = p.T, L = p.L, TR = p.TR;
    const int j0 = s * W;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * p.ldm;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = p.alpha + (size_t)b * T * p.ldo;
    const int LDM = p.ldm, LDO = p.ldo;
    const int nrows = Tb;

    const bool has_producer = so > 0;
    const bool has_consumer = s < p.NS - 1 && j0 + W < Lb;
    const u64* hin = p.halo + ((size_t)b * p.NS + (has_producer ? s - 1 : 0)) * (size_t)T * MX_TRP;
    u64* hout = p.halo + ((size_t)b * p.NS + s) * (size_t)T * MX_TRP;
    // LDS geometry: li = col - j0 + 32 (halo [0,32), own [32, W+32))

    // ---- prologue: the strip's transition rows -> LDS tile (coalesced, once), then -> registers ----
    {
        float* tile = reinterpret_cast<float*>(smem_raw);
        constexpr int NTHR = NT + 192, RPP = NTHR / 32;       // rows per pass
        const int rlo = j0 - 32;
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
        // =========================================================== compute waves: CPL vertices per lane
        const int l = tid;
        const int j = j0 + CPL * l;
        const bool col_ok = j < L;
        // E[c][k]: link of predecessor j + c - 32 + k into vertex j + c (k ascending = index ascending).
        // CPL == 2 (r04): the window is read as nine ds_read_b128 from the 16-byte aligned base 4 * (l >> 1) (256 B/clk; the eight-byte aligned
        // ds_read2_b64 form it replaces runs at 128 B/clk, and the four compute waves read right after the barrier: the LDS pipe, not the
        // latency, paced the row head), so a lane's weights are stored against THAT window: EW[c][q], q = 0..35, -inf where window element q
        // is no predecessor of vertex c (odd lanes sit two elements into the window).  add / max only: bit-identical to the 32-term form.
        constexpr int NW = (CPL == 2) ? 36 : 32;
        float E[CPL][NW];
        const float* tile = reinterpret_cast<const float*>(smem_raw);
        const int woff = (CPL == 2) ? 2 * (l & 1) : 0;           // the lane's own window starts woff elements into the aligned one
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const int k = q - woff - (CPL == 2 ? c : 0);     // CPL == 2: element q of the aligned window is predecessor index k of vertex c
                if constexpr (CPL == 2) E[c][q] = (k >= 0 && k < 32) ? tile[(CPL * l + c + (k >= 0 && k < 32 ? k : 0)) * 33 + (31 - (k >= 0 && k < 32 ? k : 0))] : NEG_INF;
                else E[c][q] = tile[(CPL * l + c + q) * 33 + (31 - q)];   // row (j+c-32+k) - (j0-32), transition d-1 = 31-k
            }
        __syncthreads();                         // tile consumed: the loader may start filling the ring over it
        mx_barrier();                            // prologue barrier: match row 0 is in the ring

        const bool prof = (MX_DBG(p) & 2) && b == 0 && s == p.NS - 1 && wave == 0;
        u64 pf[4] = {0, 0, 0, 0}, pf_last = prof ? __builtin_amdgcn_s_memtime() : 0;
        auto stamp = [&](int i) { if (prof) { const u64 tn = __builtin_amdgcn_s_memtime(); pf[i] += tn - pf_last; pf_last = tn; } };
        for (int it = 0; it < nrows; ++it) {
            const int t = it;
            const int cur = it & 1, prv = cur ^ 1;
            float a[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) a[c] = NEG_INF;
            if (it == 0) {
                if (j == 0) a[0] = Mring[(size_t)(it % MX_RING) * W];        // the start vertex
            } else {
                // row head: the match values and the (32 + CPL)-value window leave as one issue group (see dag_dp_strip4g.hip)
                float w[(CPL == 2) ? 36 : 32 + CPL], m2[CPL];
                const u32 maddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Mring + (size_t)(it % MX_RING) * W + CPL * l);
                const u32 vaddr = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Abuf + prv * RL + CPL * l);
                if constexpr (CPL == 4) {
                    mx_v4f mt, pv[9];
                    asm volatile(
                        "ds_read_b128 %0, %10\n\t"
                        "ds_read_b128 %1, %11\n\t"
                        "ds_read_b128 %2, %11 offset:16\n\t"
                        "ds_read_b128 %3, %11 offset:32\n\t"
                        "ds_read_b128 %4, %11 offset:48\n\t"
                        "ds_read_b128 %5, %11 offset:64\n\t"
                        "ds_read_b128 %6, %11 offset:80\n\t"
                        "ds_read_b128 %7, %11 offset:96\n\t"
                        "ds_read_b128 %8, %11 offset:112\n\t"
                        "ds_read_b128 %9, %11 offset:128\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(mt), "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]),
                          "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8])
                        : "v"(maddr), "v"(vaddr)
                        : "memory");
#pragma unroll
                    for (int k = 0; k < 9; ++k) { w[4 * k] = pv[k].x; w[4 * k + 1] = pv[k].y; w[4 * k + 2] = pv[k].z; w[4 * k + 3] = pv[k].w; }
                    m2[0] = mt.x; m2[1] = mt.y; m2[2] = mt.z; m2[3] = mt.w;
                } else if constexpr (CPL == 1) {
                    const float* wp = Abuf + prv * RL + l;                      // 33 dwords, 4-byte aligned: ds_read2_b32 pairs
#pragma unroll
                    for (int k = 0; k < 33; ++k) w[k] = wp[k];
                    m2[0] = Mring[(size_t)(it % MX_RING) * W + l];
                } else if constexpr (CPL == 2) {
                    mx_v2f mt; mx_v4f pv[9];
                    const u32 vaddr16 = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(Abuf + prv * RL + 4 * (l >> 1));
                    asm volatile(
                        "ds_read_b64 %0, %10\n\t"
                        "ds_read_b128 %1, %11\n\t"
                        "ds_read_b128 %2, %11 offset:16\n\t"
                        "ds_read_b128 %3, %11 offset:32\n\t"
                        "ds_read_b128 %4, %11 offset:48\n\t"
                        "ds_read_b128 %5, %11 offset:64\n\t"
                        "ds_read_b128 %6, %11 offset:80\n\t"
                        "ds_read_b128 %7, %11 offset:96\n\t"
                        "ds_read_b128 %8, %11 offset:112\n\t"
                        "ds_read_b128 %9, %11 offset:128\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(mt), "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]),
                          "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8])
                        : "v"(maddr), "v"(vaddr16)
                        : "memory");
#pragma unroll
                    for (int k = 0; k < 9; ++k) { w[4 * k] = pv[k].x; w[4 * k + 1] = pv[k].y; w[4 * k + 2] = pv[k].z; w[4 * k + 3] = pv[k].w; }
                    m2[0] = mt.x; m2[1] = mt.y;
                } else {
                    mx_v2f mt, pl; mx_v4f pq[8];          // (CPL == 2 before r04) 17 eight-byte slots: eight ds_read2_b64 + one ds_read_b64
                    asm volatile(
                        "ds_read_b64 %0, %10\n\t"
                        "ds_read2_b64 %1, %11 offset1:1\n\t"
                        "ds_read2_b64 %2, %11 offset0:2 offset1:3\n\t"
                        "ds_read2_b64 %3, %11 offset0:4 offset1:5\n\t"
                        "ds_read2_b64 %4, %11 offset0:6 offset1:7\n\t"
                        "ds_read2_b64 %5, %11 offset0:8 offset1:9\n\t"
                        "ds_read2_b64 %6, %11 offset0:10 offset1:11\n\t"
                        "ds_read2_b64 %7, %11 offset0:12 offset1:13\n\t"
                        "ds_read2_b64 %8, %11 offset0:14 offset1:15\n\t"
                        "ds_read_b64 %9, %11 offset:128\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(mt), "=&v"(pq[0]), "=&v"(pq[1]), "=&v"(pq[2]), "=&v"(pq[3]), "=&v"(pq[4]),
                          "=&v"(pq[5]), "=&v"(pq[6]), "=&v"(pq[7]), "=&v"(pl)
                        : "v"(maddr), "v"(vaddr)
                        : "memory");
#pragma unroll
                    for (int k = 0; k < 8; ++k) { w[4 * k] = pq[k].x; w[4 * k + 1] = pq[k].y; w[4 * k + 2] = pq[k].z; w[4 * k + 3] = pq[k].w; }
                    w[32] = pl.x; w[33] = pl.y;
                    m2[0] = mt.x; m2[1] = mt.y;
                }
                stamp(0);                                   // barrier exit -> window and match values in registers
                if (!(MX_DBG(p) & 16))
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    // max over the 32 predecessors: window element c + k, a 3-input max tree (values only — the arg-max is
                    // recomputed by the back-trace for the 1 cell per row that needs it)
                    float mx;
                    if constexpr (CPL == 2) {
                        float x[36];
#pragma unroll
                        for (int q = 0; q < 36; ++q) x[q] = w[q] + E[c][q];
                        float m12[12];
#pragma unroll
                        for (int g = 0; g < 12; ++g) m12[g] = fmaxf(fmaxf(x[3 * g], x[3 * g + 1]), x[3 * g + 2]);
                        const float m4a = fmaxf(fmaxf(m12[0], m12[1]), m12[2]), m4b = fmaxf(fmaxf(m12[3], m12[4]), m12[5]);
                        const float m4c = fmaxf(fmaxf(m12[6], m12[7]), m12[8]), m4d = fmaxf(fmaxf(m12[9], m12[10]), m12[11]);
                        mx = (MX_DBG(p) & 8) ? x[5 + c] : fmaxf(fmaxf(m4a, m4b), fmaxf(m4c, m4d));
                    } else {
                        float x[32];
#pragma unroll
                        for (int k = 0; k < 32; ++k) x[k] = w[c + k] + E[c][k];
                        float m10[11];
#pragma unroll
                        for (int g = 0; g < 10; ++g) m10[g] = fmaxf(fmaxf(x[3 * g], x[3 * g + 1]), x[3 * g + 2]);
                        m10[10] = fmaxf(x[30], x[31]);
                        const float m4a = fmaxf(fmaxf(m10[0], m10[1]), m10[2]), m4b = fmaxf(fmaxf(m10[3], m10[4]), m10[5]);
                        const float m4c = fmaxf(fmaxf(m10[6], m10[7]), m10[8]), m4d = fmaxf(m10[9], m10[10]);
                        mx = (MX_DBG(p) & 8) ? x[5 + c] : fmaxf(fmaxf(m4a, m4b), fmaxf(m4c, m4d));
                    }
                    const bool act = (j + c >= t) && (j + c < Lb);
                    const float cand = mx + m2[c];
                    a[c] = act ? cand : NEG_INF;                 // (a select, not a branch: the two vertices' trees interleave)
                }
            }
            stamp(1);                                       // adds + max trees
            const bool st_ok = col_ok && !(MX_DBG(p) & 4);
            if constexpr (CPL == 4) {
                *reinterpret_cast<float4*>(Abuf + cur * RL + 32 + 4 * l) = make_float4(a[0], a[1], a[2], a[3]);
                if (st_ok) *reinterpret_cast<float4*>(O + (size_t)t * LDO + j) = make_float4(a[0], a[1], a[2], a[3]);
            } else if constexpr (CPL == 1) {
                Abuf[cur * RL + 32 + l] = a[0];
                if (st_ok) O[(size_t)t * LDO + j] = a[0];
            } else {
                *reinterpret_cast<float2*>(Abuf + cur * RL + 32 + 2 * l) = make_float2(a[0], a[1]);
                if (st_ok) *reinterpret_cast<float2*>(O + (size_t)t * LDO + j) = make_float2(a[0], a[1]);
            }
            if (prof) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp(2);                                       // stores issued, LDS write done
            mx_barrier();
            stamp(3);                                       // barrier
        }
        if (prof && lane == 0) { for (int i = 0; i < 4; ++i) p.counters[8 + i] = (u32)(pf[i] >> 4); p.counters[12] = (u32)nrows; }
        if (col_ok) for (int t = Tb; t < T; ++t) {
            if constexpr (CPL == 4) *reinterpret_cast<float4*>(O + (size_t)t * LDO + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            else if constexpr (CPL == 1) O[(size_t)t * LDO + j] = NEG_INF;
            else *reinterpret_cast<float2*>(O + (size_t)t * LDO + j) = make_float2(NEG_INF, NEG_INF);
        }
    } else if (wave == NCW) {
        // =========================================================== loader wave: match rows -> LDS ring (LDS-DMA)
        auto issue_row = [&](int itr) {
            const float* rowp = M + (size_t)itr * LDM;
            float* slot = Mring + (size_t)(itr % MX_RING) * W;
#pragma unroll
            for (int i = 0; i < DPR; ++i) {
                const int col = j0 + i * 256 + lane * 4;
                const float* g = rowp + (col < L ? col : 0);          // out-of-range lanes re-read a valid address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(slot + i * 256), 16, 0, 0);
            }
        };
        __syncthreads();                         // link tile consumed
        for (int r = 0; r < MX_RING - 1 && r < nrows; ++r) issue_row(r);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mx_barrier();                            // prologue barrier
        for (int it = 0; it < nrows; ++it) {
            const int nx = it + MX_RING - 1;     // slot (it-1) % RING was last read during iteration it-1: free now
            if (nx < nrows) {
                issue_row(nx);
                if (DPR == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // rows it+2 .. it+7 may stay in flight
                else if (DPR == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            mx_barrier();
        }
    } else if (wave == NCW + 1) {
        // =========================================================== fetch wave: left neighbour's last 32 vertices -> LDS halo
        const bool hl = lane < MX_TRP;
        u64 g[MX_CH];
#pragma unroll
        for (int k = 0; k < MX_CH; ++k) g[k] = 0;
        auto load_row = [&](int itr) -> u64 {    // rolling prefetch, MX_CH rows ahead (see dag_dp_strip4g.hip)
            if (itr < nrows && hl) return mx_gran_load(hin + (size_t)itr * MX_TRP + lane);
            return 0;
        };
        if (has_producer) {
#pragma unroll
            for (int k = 0; k < MX_CH; ++k) g[k] = load_row(k);
        }
        __syncthreads();                         // link tile consumed
        mx_barrier();                            // prologue barrier
        for (int itb = 0; itb < nrows; itb += MX_CH) {
#pragma unroll
            for (int k = 0; k < MX_CH; ++k) {
                const int it = itb + k;
                if (it >= nrows) break;
                const int cur = it & 1;
                float hv = NEG_INF;
                if (has_producer && hl) {
                    const u32 want = p.tag_base + 1u + (u32)it;
                    u64 xg = g[k];
                    u32 spins = 0;
                    while (!__all((u32)(xg >> 32) == want)) {
                        if ((u32)(xg >> 32) != want) xg = mx_gran_load(hin + (size_t)it * MX_TRP + lane);
                        if (++spins > MX_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    hv = __uint_as_float((u32)xg);
                }
                if (hl) Abuf[cur * RL + lane] = hv;
                if (has_producer) g[k] = load_row(it + MX_CH);
                mx_barrier();
            }
        }
    } else {
        // =========================================================== publish wave: last 32 vertices -> granules
        const bool pl = has_consumer && lane < MX_TRP;
        __syncthreads();                         // link tile consumed
        mx_barrier();                            // prologue barrier
        for (int it = 0; it <= nrows; ++it) {
            if (it > 0 && pl) {                  // row it-1 is complete; compute now writes the other buffer
                const float v = Abuf[((it - 1) & 1) * RL + W + lane];
                mx_gran_store(hout + (size_t)(it - 1) * MX_TRP + lane, p.tag_base + 1u + (u32)(it - 1), v);
            }
            if (it < nrows) mx_barrier();
        }
    }
}

template <int NT, int CPL>
__global__ __launch_bounds__(NT + 192) void dag_maxstrip_kernel(MStripParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int W = CPL * NT;
    u32* s_ticket = reinterpret_cast<u32*>(smem_raw);          // 16-byte header; everything else starts at +16
    const int tid = threadIdx.x;
    if (tid == 0) *s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = *s_ticket;                              // producers hold smaller tickets than their consumers
    const int so = (int)(ticket / p.B);
    const int b = (int)(ticket % p.B);
    const int s = so;
    const int j0 = s * W;
    const int T = p.T, L = p.L;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {                    // nothing reachable in this strip: -inf everywhere, no hand-off
        float* O = p.alpha + (size_t)b * T * p.ldo;
        for (int j = j0 + 4 * tid; j < j0 + W && j < L; j += 4 * (NT + 192))
            for (int t = 0; t < T; ++t)
                *reinterpret_cast<float4*>(O + (size_t)t * p.ldo + j) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        return;
    }
    maxstrip_body<NT, CPL>(p, smem_raw + 16, b, s, so);
}

// ---- lazy back-trace (replaces the trace tensor of dag_best_alignment.cu:39-130 + the pointer chase of :160-201) ----------
// One workgroup per sample.  At row t and vertex j the predecessor is  argmax_i alpha_max[t-1][i] + links[i][j-i-1]  over
// i in [j-TR, j-1], the smallest i on ties, -1 if every candidate is -inf: exactly what the eager kernels store in
// trace[t][j] — but only T cells per sample are ever asked for, not T*L.
// A pointer chase costs one dependent memory round trip per row.  Here a round trip buys BT_HOPS rows: the path moves left by
// 1..TR vertices per row, so the next h-th hop can only need alpha_max[t-h][pos-32h .. pos-h]; all BT_HOPS segments are
// fetched at once into LDS by the whole workgroup, and the transition rows below `pos` sit in an LDS window that is refilled
// in bulk every few dozen hops.  Wave 0 then resolves the hops from LDS (lane d evaluates predecessor pos-1-d).
constexpr int BT_HOPS = 10;
constexpr int BT_LW = 768;                    // transition rows cached in LDS
constexpr int BT_SEG = 768;                   // segment pitch: >= 31 * 2 * HOPS + 1 (frame one iteration old: up to 2*HOPS rows back) and a
                                              // multiple of the 192 fetch lanes, so that a lane's loads of a segment are base + q + 192 j
constexpr int BT_PER = BT_HOPS * (BT_SEG / 192);           // loads per lane of the three fetch waves (wave 0 resolves hops)
static_assert(BT_SEG % 192 == 0 && BT_SEG > 62 * BT_HOPS, "segment pitch");

// Maximum of lanes 0..31 of a wave, as a wave-uniform value.  One DPP-modified v_max per step (the compiler's fmaxf + update_dpp
// form is mov_dpp + two canonicalising v_max + v_max per step, and the hop is a chain of dependent instructions on ONE wave, so
// instruction count is latency): four steps inside each row of 16, row_bcast:15 carries row 0's result into row 1, one readlane.
// The inputs are never NaN.  s_nop 1: VALU write -> DPP read needs two wait states, which inline asm must provide itself.
__device__ __forceinline__ float bt_max_lanes32(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 0" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
}

__global__ __launch_bounds__(256) void dag_backtrace_lazy_kernel(
    const float* __restrict__ amax, const float* __restrict__ links, const int64_t* __restrict__ out_len,
    const int64_t* __restrict__ tgt_len, int64_t* __restrict__ path, int B, int T, int L, int TR, int LDA)
{
    extern __shared__ __attribute__((aligned(16))) char bt_smem[];
    float* lk = reinterpret_cast<float*>(bt_smem);                       // [BT_LW][TR]  rows lbase .. lbase + BT_LW - 1
    float* seg = lk + (size_t)BT_LW * TR;                                // [BT_HOPS][BT_SEG]
    int32_t* lp = reinterpret_cast<int32_t*>(seg + BT_HOPS * BT_SEG);    // [L]
    __shared__ int s_state[4];                                           // pos, t, done
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    for (int j = tid; j < L; j += 256) lp[j] = -1;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    const float* A = amax + (size_t)b * T * LDA;
    const float* K = links + (size_t)b * L * TR;
    int pos = Lb - 1, t = Tb - 1, lbase = 0x3fffffff;
    bool done = !valid;
    // segments are fetched one iteration AHEAD, relative to the frame (tF, pF) = (t, pos) at that time: slot k holds
    // alpha_max[tF - HOPS - k][pF - 32(HOPS+k) .. pF - (HOPS+k)], a superset of what hop k of the next iteration can touch
    // (the path moves 1..32 vertices left per row), so the memory round trip overlaps the hop resolution of this iteration.
    float pre[BT_PER];
    // (the per-element form of this loop — e / BT_SEG, e % BT_SEG, 64-bit address per load — spent ~1.4 us per iteration on
    //  address arithmetic before the last load was even issued: r01g)
    auto fetch = [&](int tF, int pF) {
        if (tid < 64) return;
        const int q0 = tid - 64;
#pragma unroll
        for (int k = 1; k <= BT_HOPS; ++k) {
            const int row = tF - BT_HOPS - k, col0 = pF - 32 * (BT_HOPS + k), qmax = 31 * (BT_HOPS + k);
            const float* base = A + (long)(row < 0 ? 0 : row) * LDA + col0;       // wave-uniform
#pragma unroll
            for (int j = 0; j < BT_SEG / 192; ++j) {
                const int q = q0 + 192 * j, col = col0 + q;
                float v = NEG_INF;
                if (row >= 0 && q <= qmax && col >= 0 && col < L) v = base[q];
                pre[(k - 1) * (BT_SEG / 192) + j] = v;
            }
        }
    };
    auto stash = [&]() {
        if (tid < 64) return;
        const int q0 = tid - 64;
#pragma unroll
        for (int k = 1; k <= BT_HOPS; ++k)
#pragma unroll
            for (int j = 0; j < BT_SEG / 192; ++j) seg[(k - 1) * BT_SEG + q0 + 192 * j] = pre[(k - 1) * (BT_SEG / 192) + j];
    };
    int tF = t + BT_HOPS, pF = pos + BT_HOPS;                            // pretend frame of the iteration before the first
    if (!done) { fetch(tF, pF); stash(); }
    __syncthreads();
    while (!done) {
        // (a) transition window: rows [pos - 32*HOPS, pos) must be cached
        if ((pos - 32 * BT_HOPS < lbase && lbase > 0) || pos > lbase + BT_LW) {
            lbase = max(pos - BT_LW, 0);
            const int nrow = min(BT_LW, L - lbase);                          // rows of the sample that exist
            const long lo = (long)lbase * TR;
            const int n = nrow * TR;
            if ((TR & 3) == 0 && ((uintptr_t)(K + lo) & 15) == 0) {
                // The window is 96 KB; the eight-loads-at-a-time loop this replaces took twelve dependent round trips (11 us per
                // refill, 45 % of the kernel: r01g s_memtime accounting).  LDS-DMA: 24 requests per lane back to back, no registers —
                // a register-staged version is re-serialised by the scheduler (load / s_waitcnt vmcnt(0) / ds_write per quad).
                constexpr int NQ = BT_LW * 32 / 4 / 256;                     // 24 chunks of 64 quads per wave at TR = 32
                const float4* src = reinterpret_cast<const float4*>(K + lo);
                const int n4 = n >> 2, wave = tid >> 6;
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int chunk = u * 4 + wave, e = chunk * 64 + lane;
                    if (chunk * 64 < n4)                                    // wave-uniform; lanes past the end re-read quad 0 into rows nobody reads
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (e < n4 ? e : 0)),
                                                         (__attribute__((address_space(3))) void*)(lk + chunk * 256), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                for (int e0 = tid; e0 < n; e0 += 8 * 256) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; v[u] = (e < n) ? K[lo + e] : NEG_INF; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; if (e < n) lk[e] = v[u]; }
                }
            }
            __syncthreads();
        }
        // (b) request the NEXT iteration's segments (frame = now); they land while wave 0 works
        const int t0 = t, p0 = pos;
        fetch(t0, p0);
        // (c) wave 0 resolves up to HOPS hops from LDS (segments of frame (tF, pF))
        if (tid < 64) {
            for (int h = 1; h <= BT_HOPS; ++h) {
                if (lane == 0) lp[pos] = t;
                if (t == 0 || pos < t) { done = true; break; }           // row 0 / under the diagonal: trace = -1
                const int d = lane, i = pos - 1 - d;
                float x = NEG_INF;
                // (lk offset = pos * TR [scalar multiply] + a per-lane constant: the per-lane v_mul_lo_u32 was a quarter-rate instruction on the hop chain)
                if (d < TR && i >= 0) x = seg[(h - 1) * BT_SEG + (i - (pF - 32 * (BT_HOPS + h)))] + lk[pos * TR + (d - (1 + d + lbase) * TR)];
                const float mx = bt_max_lanes32(x);                       // lanes >= TR (and so all of 32..63) hold -inf
                --t;
                if (mx == NEG_INF) { pos = -1; done = true; break; }
                // the LARGEST d (smallest predecessor index) among the lanes that attain the maximum: only candidates can equal a finite mx
                const unsigned long long hit = __builtin_amdgcn_fcmpf(x, mx, 1 /* FCMP_OEQ */);      // (the f32 form: __builtin_amdgcn_fcmp takes doubles — two v_cvt_f64_f32 and a v_cmp_eq_f64 on every hop)
                pos = pos - 1 - (63 - __builtin_clzll(hit));
            }
            if (lane == 0) { s_state[0] = pos; s_state[1] = t; s_state[2] = done ? 1 : 0; }
        }
        __syncthreads();
        pos = s_state[0]; t = s_state[1]; done = s_state[2] != 0;
        stash();                                 // seg <- the segments requested in (b); their frame:
        tF = t0; pF = p0;
        __syncthreads();
    }
    __syncthreads();
    for (int j = tid; j < L; j += 256) path[(size_t)b * L + j] = lp[j];
}

// ---- r04: the same lazy back-trace with NOTHING but the hops on its critical path (TR == 32) --------------------------------------------------
// dag_backtrace_lazy_kernel above stops the hop wave twice per iteration (segments: registers -> LDS after the hops; a second barrier) and
// every few dozen hops for a synchronous 96 KB refill of the transition window (9 refills of ~4 us at C2: a quarter of the kernel).  Here
//   * the transition rows live in a RING of 512 rows (64 KB, row r in slot r & 511): every iteration the three helper waves request the
//     rows the NEXT iteration can need, [pos - 64 H, pos - 32 H), as LDS-DMA (1 KB = 8 rows per request, no registers) — they land while
//     wave 0 hops.  With H = 7 the slots they overwrite belong to rows >= pos + 57: already behind the path;
//   * the alpha_max segments of the next iteration are LDS-DMA too (4-byte requests: 64 consecutive columns per request), into the OTHER of two
//     segment buffers: no staging registers, no stash phase, ONE barrier per iteration;
//   * lanes whose column / row falls outside the table request a clamped address: what lands there is never read (a hop only reads
//     predecessors i >= 0 of a row t - 1 >= 0, inside the segment's range by the 1..32-vertices-per-row bound).
constexpr int BR_H = 7;                        // hops per iteration
constexpr int BR_RW = 512;                     // ring rows (>= 64 H + 8 + 32 spare: see above)
constexpr int BR_SEG = 512;                    // segment pitch: >= 62 H + 1 + 3 = 438 (the base is aligned down to 4 columns), a multiple of 256 (one request = 64 lanes x 16 bytes)

__global__ __launch_bounds__(256) void dag_backtrace_ring_kernel(
    const float* __restrict__ amax, const float* __restrict__ links, const int64_t* __restrict__ out_len,
    const int64_t* __restrict__ tgt_len, int64_t* __restrict__ path, int B, int T, int L, int LDA)
{
    constexpr int TR = 32;
    extern __shared__ __attribute__((aligned(16))) char br_smem[];
    float* ring = reinterpret_cast<float*>(br_smem);                          // [BR_RW][32]
    float* seg = ring + (size_t)BR_RW * TR;                                   // [2][BR_H][BR_SEG]
    int32_t* lp = reinterpret_cast<int32_t*>(seg + 2 * BR_H * BR_SEG);        // [L]
    __shared__ int s_state[2][4];                                             // per iteration parity: pos, t, done
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < L; j += 256) lp[j] = -1;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    const float* A = amax + (size_t)b * T * LDA;
    const float* K = links + (size_t)b * L * TR;
    int pos = Lb - 1, t = Tb - 1;
    bool done = !valid;

    // alpha_max segments for the iteration AFTER the one that starts at frame (tF, pF): slot k - 1 = row tF - H - k, columns pF - 32 (H + k) ...
    // (r06: 16 bytes per lane — a request moves 256 columns, two per segment instead of five to eight 64-column ones: the helpers' round trip
    //  paced the iteration (1.7 us per 7 hops against 1.1 us of hops), and a third of it was the issue of ~18 requests per wave.  The segment's
    //  base column is aligned DOWN to 4; rows are 16-byte aligned by the launcher's pitch condition.)
    auto request_segments = [&](int tF, int pF, int buf, int w, int nw) {   // wave w of nw takes every nw-th request
        int r = 0;
#pragma unroll
        for (int k = 1; k <= BR_H; ++k) {
            const int row = tF - BR_H - k, col0 = (pF - 32 * (BR_H + k)) & ~3, nreq = (31 * (BR_H + k) + 4 + 255) >> 8;
            const float* rowp = A + (size_t)(row < 0 ? 0 : row) * LDA;          // wave-uniform
            float* dst = seg + ((size_t)buf * BR_H + (k - 1)) * BR_SEG;
#pragma unroll
            for (int j = 0; j < (31 * (2 * BR_H) + 4 + 255) / 256; ++j) {
                if (j < nreq) {
                    if (r % nw == w) {
                        int col = col0 + 256 * j + 4 * lane;                   // a multiple of 4: the lane's four columns lie inside the row's pitch or are clamped away together
                        col = col < 0 ? 0 : (col > LDA - 4 ? LDA - 4 : col);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + col),
                                                         (__attribute__((address_space(3))) void*)(dst + 256 * j), 16, 0, 0);
                    }
                    ++r;
                }
            }
        }
    };
    // transition rows [8 c0, 8 c1) into the ring, 8 rows (1 KB) per request
    auto request_rows = [&](int c0, int c1, int w, int nw) {
        for (int c = c0 + w; c < c1; c += nw) {
            const int row = 8 * c + (lane >> 3);                              // lane = 16-byte quad (lane & 7) of row 8 c + (lane >> 3)
            const float* src = K + (size_t)(row < L ? row : L - 1) * TR + 4 * (lane & 7);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ring + (size_t)((8 * c) & (BR_RW - 1)) * TR), 16, 0, 0);
        }
    };
    int lo_chunk = 0;                                                         // rows >= 8 * lo_chunk (up to the start vertex) are in the ring
    if (!done) {
        const int c1 = (pos >> 3) + 1;
        lo_chunk = max(0, (pos - 64 * BR_H) >> 3);
        if (pos - 64 * BR_H < 0) lo_chunk = 0;
        request_rows(lo_chunk, c1, wave, 4);
        request_segments(t + BR_H, pos + BR_H, 0, wave, 4);                   // pretend frame of "the iteration before the first"
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int tF = t + BR_H, pF = pos + BR_H;                                       // frame of the segments in buffer (it & 1)
    for (int it = 0; !done; ++it) {
        const int t0 = t, p0 = pos;
        if (wave != 0) {
            // ---- helpers: everything the NEXT iteration reads, requested now
            request_segments(t0, p0, (it + 1) & 1, wave - 1, 3);
            const int want = (p0 - 64 * BR_H) < 0 ? 0 : ((p0 - 64 * BR_H) >> 3);
            if (want < lo_chunk) request_rows(want, lo_chunk, wave - 1, 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // ---- wave 0: up to H hops out of LDS (segments of frame (tF, pF), buffer it & 1).  The hop is a chain of dependent instructions on
            // one wave, so every instruction and every taken branch is latency: unconditional clamped reads + a select instead of an exec-mask
            // region, the visited positions collected in ONE register (lane h = position of hop h) and written to the path image after the
            // hops, the end test on the scalar bits of the maximum.
            const float* sg = seg + (size_t)(it & 1) * BR_H * BR_SEG;
            const int tstart = t;
            int hist = -1, nh = 0;                                            // lane h - 1: the vertex visited at row tstart - (h - 1)
#pragma unroll
            for (int h = 1; h <= BR_H; ++h) {
                hist = (lane == h - 1) ? pos : hist;                             // (off the chain: pos is known when the hop starts)
                nh = h;
                if (t == 0 || pos < t) { done = true; break; }               // row 0 / under the diagonal: trace = -1
                const int i = pos - 1 - lane;
                const int ic = i < 0 ? 0 : i;
                const float av = sg[(h - 1) * BR_SEG + (ic - ((pF - 32 * (BR_H + h)) & ~3))];
                const float kv = ring[((ic & (BR_RW - 1)) << 5) + (lane & 31)];
                const float x = (lane < TR && i >= 0) ? av + kv : NEG_INF;
                const float mx = bt_max_lanes32(x);                           // lanes >= 32 hold -inf
                --t;
                if (__builtin_bit_cast(unsigned, mx) == 0xff800000u) { pos = -1; done = true; break; }
                const unsigned long long hit = __builtin_amdgcn_fcmpf(x, mx, 1 /* FCMP_OEQ */);
                pos = pos - 1 - (63 - __builtin_clzll(hit));                  // the LARGEST d = smallest predecessor index among the maxima
            }
            if (lane < nh) lp[hist] = tstart - lane;
            if (lane == 0) { s_state[it & 1][0] = pos; s_state[it & 1][1] = t; s_state[it & 1][2] = done ? 1 : 0; }
        }
        {
            const int want = (p0 - 64 * BR_H) < 0 ? 0 : ((p0 - 64 * BR_H) >> 3);
            if (want < lo_chunk) lo_chunk = want;
        }
        __syncthreads();
        pos = s_state[it & 1][0]; t = s_state[it & 1][1]; done = s_state[it & 1][2] != 0;
        tF = t0; pF = p0;
    }
    __syncthreads();
    for (int j = tid; j < L; j += 256) path[(size_t)b * L + j] = lp[j];
}

// ------------------------------------------------------------------------------------------------ host side
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base);

bool maxstrip_supported(const void* match, const void* alpha_max, int L, int TR, int ldm, int ldo)
{
    if (TR > 32 || (ldm & 3) || (ldo & 3) || ldm < ((L + 3) & ~3) || ldo < ((L + 3) & ~3)) return false;
    if (L > 8192) return false;                                // back-trace: path image (4L) + transition window (96 KB) + segments (25 KB) in LDS
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha_max;
    return (a & 15) == 0;
}

template <int NT, int CPL>
static int launch_one_mx(const MStripParams& p, int nwg, hipStream_t st)
{
    constexpr int W = CPL * NT, RL = W + 32;
    const size_t lds_main = (size_t)(2 * RL + MX_RING * W) * 4 + 16;
    const size_t lds_tile = (size_t)(W + 32) * 33 * 4;
    const size_t lds = (lds_main > lds_tile ? lds_main : lds_tile) + 32;
    auto k = dag_maxstrip_kernel<NT, CPL>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(NT + 192), lds, st, p);
    return check_launch("dag_best_alignment(maxstrip)");
}

static int g_bt_ring = 1;                     // dsp_dag_set_option("bt_ring", 0): the r01-r03 back-trace kernel (cross-check)
void set_bt_ring(int v) { g_bt_ring = v; }
static int g_mx_cpl = 0;                      // experiment switch (dsp_dag_set_option("mx_cpl", 1 | 2 | 4)): vertices per lane of the max-DP; 0 = auto
void set_mx_cpl(int v) { g_mx_cpl = v; }

// alpha_max by column strips (values only), then the lazy back-trace: no trace tensor
int launch_dag_maxstrip(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                        float* alpha_max, int64_t* path, int B, int T, int L, int TR, int ldm, int ldo, hipStream_t st)
{
    // one direction only: 4 vertices per lane in 1024-vertex strips when that still fills the chip (>= ~200 workgroups),
    // otherwise 2 vertices per lane in 512-vertex strips (twice the waves for the same vertices)
    const int ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512, ns256 = (L + 255) / 256;
    const int cpl = g_mx_cpl == 1 || g_mx_cpl == 2 || g_mx_cpl == 4 ? g_mx_cpl : ((long)B * ns1024 >= 200 ? 4 : 2);
    const bool wide = cpl == 4;
    const int NS = cpl == 4 ? ns1024 : (cpl == 2 ? ns512 : ns256);
    MStripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha_max;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ldm = ldm; p.ldo = ldo;
#ifdef DSP_MX_PROF                                  // instrumentation build only (tools/prof_maxstrip.py): nothing on the product's launch path
    { static const char* const e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "prof")) ? 2 : 0; static const char* const a = getenv("DSP_MX_ABLATE"); if (a) p.dbg |= atoi(a) & 28; }
#else
    p.dbg = 0;
#endif
    const size_t halo_bytes = (size_t)B * NS * T * MX_TRP * sizeof(u64);
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    rc = wide ? launch_one_mx<256, 4>(p, B * NS, st) : (cpl == 2 ? launch_one_mx<256, 2>(p, B * NS, st) : launch_one_mx<256, 1>(p, B * NS, st));
    if (rc) return rc;
    if (TR == 32 && g_bt_ring && (((uintptr_t)links) & 15) == 0) {
        const size_t lds3 = ((size_t)BR_RW * 32 + 2 * BR_H * BR_SEG + (size_t)L) * 4;
        set_max_dynamic_lds((const void*)dag_backtrace_ring_kernel, (int)lds3);
        hipLaunchKernelGGL(dag_backtrace_ring_kernel, dim3(B), dim3(256), lds3, st, alpha_max, links, out_len, tgt_len, path, B, T, L, ldo);
        return check_launch("dag_best_alignment(ring back-trace)");
    }
    const size_t lds2 = ((size_t)BT_LW * TR + BT_HOPS * BT_SEG + (size_t)L) * 4;
    set_max_dynamic_lds((const void*)dag_backtrace_lazy_kernel, (int)lds2);
    hipLaunchKernelGGL(dag_backtrace_lazy_kernel, dim3(B), dim3(256), lds2, st, alpha_max, links, out_len, tgt_len, path, B, T, L, TR, ldo);
    return check_launch("dag_best_alignment(lazy back-trace)");
}

}  // namespace dsp
