// dag_dp_colsweep.hip — banded (TR <= 32) DAG DP, K2 alpha || K3 beta as a COLUMN SWEEP with one DP ROW per lane pair.
//
// The row-sweep kernels (dag_dp_strip4g.hip) walk t = 0..T-1 and spread a row's L vertices over the chip: 512 dependent steps, each a
// workgroup-wide chain  barrier -> LDS window -> FMAs -> log/exp tail -> LDS -> barrier  with ONE wave per SIMD to hide it (0.29 of
// the HBM roofline at C2, 45 % of the cycles idle).  This kernel turns the recurrence
//
//     A[t, j] = M[t, j] * sum_{d=1..TR} A[t-1, j-d] * W(j-d -> j)          (dag_loss.cu:94-127 in exp space)
//
// by 90 degrees: for a fixed column j the T rows are independent given the previous columns, and W(j-d -> j) is the same for every
// row.  So LANES ARE ROWS and the sweep runs over COLUMNS:
//   * a wave owns 32 consecutive DP rows, two lanes per row (lanes 0-31: the 16 most recent columns of the row's 32-column window,
//     lanes 32-63: the 16 older ones), the window lives in 16 VGPRs per lane as a circular buffer (slot = column & 15, fully unrolled);
//   * per column step a lane does 8 v_pk_fma_f32 against 16 transition weights that are wave-uniform per half: 4 broadcast
//     ds_read_b128 from a per-wave LDS tile the wave stages itself (exp2 of the raw transition rows, one 32-column block ahead);
//   * the finished cell goes to the lane of the next row with ONE DPP wave shift (value + exponent), the two half sums meet through
//     v_permlane32_swap; the dependent chain of a step is ~12 VALU ops against ~45 issued, so one wave per SIMD never waits on itself,
//     and there is NO barrier and no LDS round trip in the recurrence;
//   * numerics: every row carries ONE integer exponent Y (its window is 2^-Y scaled f32); a cell crosses to the next row as
//     (value, exponent) and is rescaled by the receiver with one v_ldexp — against 36 per lane-row in the strip kernel; rows
//     renormalise themselves every 8 columns when their window drifts, empty rows adopt the scale of their first arrival;
//   * wave -> wave (row 32k+31 -> row 32k+32) hand-off by 8-byte {value, exponent|valid} sc1 granules (G16 R2), prefetched two
//     8-column groups ahead; workgroups take tickets so a producer always holds a smaller ticket than its consumer;
//   * match / alpha are read and written directly by the owning lane, 16 bytes per 4 columns (every 128-byte line is used whole).
//
// Exactness (same contract as strip4g): a step's sum S is exact to fp32 when S >= 2^-97 (everything the window flushed is < 2^-126);
// a cell that cannot be certified (0 < S < 2^-97, S >= 2^120 / NaN, S == 0 after a deep arrival, a finite transition weight outside
// [2^-64, 2^30]) raises the sample's ABORT word and the stand-by launch (strip4g, gated on that word) recomputes that sample.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct CSParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta;
    u64* gran;                 // [ndir*B][NWT][LP] row hand-off granules (zeroed per launch)
    u32* counters;             // [0] ticket, [1] error word, [2] aborted samples, [3] debug
    u32* abort_words;          // [ndir*B] non-zero = this (direction, sample) must be recomputed by the stand-by launch
    int B, T, L, TR, ndir, NRB, NWT, LP;
    int dbg;
    int trace_b, trace_beta, trace_u;        // DSP_CS_TRACE=b,beta,u: per-step printf of one row (debugging)
};

constexpr int CS_NW = 4;                       // waves per workgroup (independent except for the row hand-off)
constexpr float CS_LOG2E = 1.4426950408889634f;
constexpr float CS_LN2 = 0.6931471805599453f;
constexpr int CS_ETGT = 40;                    // frexp exponent of the window maximum after a renormalisation
constexpr int CS_EHI = 80, CS_ELO = 0;         // renormalise when it leaves [ELO, EHI]
constexpr int CS_GUARD = 100;                  // ... and at once when an arrival lands above 2^GUARD
constexpr u32 CS_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ u64 cs_gran_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cs_gran_store(u64* p, float x, int e) {
    const u32 hi = ((u32)(e + (1 << 30)) << 1) | 1u;
    __hip_atomic_store(p, ((u64)hi << 32) | (u64)__float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float cs_shr1(float v) {           // lane i <- lane i-1 (lane 0 <- 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ int cs_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true); }
// both halves get  lower-half value + upper-half value  /  max of the two
__device__ __forceinline__ float cs_half_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float cs_half_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
// upper half <- lower half's value (what the lower half gets is not used)
__device__ __forceinline__ float cs_lower_to_upper(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return __builtin_bit_cast(float, (int)r[0]);
}

__device__ __forceinline__ void cs_fill_row(float* row, int c0, int c1, int lane) {      // [c0, c1) <- -inf, c0 % 4 == 0
    for (int c = c0 + 4 * lane; c < c1; c += 256) *reinterpret_cast<float4*>(row + c) = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
}

// Per-wave LDS (floats): [0, 1024) transition weights of the current block [step][half][slot]; [1024, 3072) two match tiles
// [32 rows][8 chunks of 4 columns], chunk c of row r at 16-byte unit r*8 + (c ^ ((r >> 1) & 7)) (conflict-free for "every lane its own
// row" ds_read_b128 / ds_write_b128 AND for the row-major global side); [3072, 4096) the alpha / beta tile of the current block in the
// same layout; [4096, 4224) a dump slot per upper-half lane.
constexpr int CS_WT = 0, CS_MT = 1024, CS_OT = 3072, CS_DUMP = 4096, CS_LDS_WAVE = 4224;

template <bool BETA>
__device__ __forceinline__ void colsweep_body(const CSParams& p, float* lds, int b, int gw, int dirslot)
{
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    const int T = p.T, L = p.L, TR = p.TR;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    float* O = (BETA ? p.beta : p.alpha) + (size_t)b * T * L;
    const float* M = p.match + (size_t)b * T * L;
    const float* K = p.links + (size_t)b * L * TR;
    const int u0 = gw * 32;
    const bool valid = Tb >= 1 && Lb >= 1 && Tb <= T && Lb <= L;
    if (!valid || u0 >= Tb) {                                   // nothing to sweep: this wave's rows are -inf
        for (int rr = 0; rr < 32; ++rr) { const int t = u0 + rr; if (t < T) cs_fill_row(O + (size_t)t * L, 0, L, lane); }
        return;
    }
    const int NB = (Lb + 31) >> 5, Lceil = NB * 32, off = BETA ? (Lceil - Lb) : 0;
    const int u = u0 + r;
    const bool row_ok = u < Tb;
    const bool has_prod = gw > 0 && !(p.dbg & 2), has_cons = (gw + 1) * 32 < Tb && !(p.dbg & 2);     // (DSP_DEBUG=cs_nohandoff: timing experiment, wrong results)
    const u64* gin = p.gran + ((size_t)(dirslot * p.B + b) * p.NWT + (has_prod ? gw - 1 : 0)) * (size_t)p.LP;
    u64* gout = p.gran + ((size_t)(dirslot * p.B + b) * p.NWT + gw) * (size_t)p.LP;
    const int vmax_u = u + Lceil - Tb;                          // beta: real column >= real row  <=>  v <= vmax_u
    const bool pub_lane = has_cons && lane == 31;
    const bool cert_lane = h == 0 && row_ok;                    // (the upper half carries no row scale: its flags mean nothing)
    float* wt = lds + CS_WT;
    const int swz = (r >> 1) & 7;
    float* otile_w = h == 0 ? (lds + CS_OT + r * 32) : (lds + CS_DUMP + r * 4 - 0);     // where this lane's 4-column results go (+ chunk)

    // ---- tile <-> HBM geometry: lane handles 16-byte chunks q = i*64 + lane of a [32 rows][8 chunks] tile
    int trow_off[4];                                            // element offset of the chunk's row, -1 = row not in the graph
    int tunit[4];                                               // LDS unit of the chunk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 64 + lane, rr = q >> 3, k = q & 7;
        const int uu = u0 + rr;
        trow_off[i] = uu < Tb ? (BETA ? (Tb - 1 - uu) : uu) * L : -1;
        tunit[i] = rr * 8 + (k ^ ((rr >> 1) & 7));
    }
    const int kcol = (lane & 7) * 4;
    v4f mraw[4];
    auto load_mtile = [&](int cb) {                             // block cb's match tile -> registers
        const int jb = BETA ? (Lceil - 32 - 32 * cb) : 32 * cb;
        int jc = jb + kcol; jc = jc < 0 ? 0 : (jc > L - 4 ? L - 4 : jc);
#pragma unroll
        for (int i = 0; i < 4; ++i) mraw[i] = *reinterpret_cast<const v4f*>(M + (trow_off[i] < 0 ? 0 : trow_off[i]) + jc);
    };
    auto write_mtile = [&](int cb) {
        float* mt = lds + CS_MT + (cb & 1) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f*>(mt + tunit[i] * 4) = mraw[i];
    };
    auto store_otile = [&](int cb) {                            // block cb's results: LDS -> HBM, whole 128-byte lines
        const int jb = BETA ? (Lceil - 32 - 32 * cb) : 32 * cb;
        const int jc = jb + kcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4f o = *reinterpret_cast<const v4f*>(lds + CS_OT + tunit[i] * 4);
            if (trow_off[i] >= 0 && jc < L) *reinterpret_cast<v4f*>(O + trow_off[i] + jc) = o;
        }
    };

    // ---- per-lane state
    float Wn[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) Wn[s] = 0.f;
    int Y = 0;
    bool has = false;                                           // the row's window holds a non-zero value
    int lossage = 0;                                            // groups since an arrival was flushed (0 = none in the window)
    int lastnz = -(1 << 30);                                    // column of the row's last non-zero arrival
    u32 lobits = 0x03000000u;                                   // smallest certified sum: 2^-121 (2^-97 while a flushed arrival is in the window)
    bool abortf = false;
    bool abortw = false;                                        // wave-uniform: a transition weight outside the certified range
    bool lostg = false;
    if (gw == 0 && off == 0 && lane == 0) { Wn[15] = 1.f; has = true; }      // virtual column -1 carries the DP's start
    if (gw == 0 && off == 0 && lane == 32) has = true;

    // ---- transition weights of a 32-column block: exp2(raw * log2 e) into the wave's LDS tile [step][half][slot]
    float wraw[16];
    auto load_wraw = [&](int cb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = i * 64 + lane, sv = qd >> 3, hh = (qd >> 2) & 1, s4 = (qd & 3) * 4;
            const int v = 32 * cb + sv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int slot = s4 + k;
                const int dd = 1 + ((v - 1 - slot) & 15) + 16 * hh;
                const int c = v - dd;
                const bool seed = (c == off - 1) && (v == off);
                const bool ok = (v >= off) && (v - off < Lb) && (c >= off) && (dd <= TR) && (cb < NB);
                const int jrow = BETA ? (Lceil - 1 - v) : c;
                const float raw = K[(size_t)(ok ? jrow : 0) * TR + (ok ? dd - 1 : 0)];
                wraw[4 * i + k] = ok ? raw : (seed ? 0.f : NEG_INF);
            }
        }
    };
    auto write_wt = [&]() {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qd = i * 64 + lane;
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float l2 = wraw[4 * i + k] * CS_LOG2E;
                w[k] = __builtin_amdgcn_exp2f(l2);
                // finite and below -64, or above 30: outside what the exactness argument covers
                bad |= (__float_as_uint(l2) - 0xC2800001u) < (0xFF800000u - 0xC2800001u);
                bad |= l2 > 30.f;
            }
            *reinterpret_cast<float4*>(wt + qd * 4) = make_float4(w[0], w[1], w[2], w[3]);
        }
#ifdef CS_DEBUG
        if (p.dbg && __any(bad) && lane == 0) printf("colsweep abort: transition weight out of range, b=%d beta=%d gw=%d\n", b, (int)BETA, gw);
#endif
        abortw |= __any(bad);
    };

    // ---- row hand-off granules: lanes 0..7 hold one 8-column group each, two groups ahead
    auto load_group = [&](int G) -> u64 {
        const int v = 8 * G + (lane & 7);
        return (has_prod && v < Lceil) ? cs_gran_load(gin + v) : 0ull;
    };
    u64 gq[2];
    gq[0] = load_group(0); gq[1] = load_group(1);

    load_wraw(0);
    load_mtile(0);
    float oq[4] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
    v4f mcur = {0.f, 0.f, 0.f, 0.f};

    for (int hb = 0; hb < 2 * NB; ++hb) {
        // The step code is unrolled over 16 columns (the period of the window's circular slots) and looped: two passes per 32-column
        // block.  Unrolling the whole block put 90 KB of code into the 64 KB instruction cache two CUs share.
        const int cb = hb >> 1, svb = (hb & 1) * 16;
        const int vb = 32 * cb;
        if (svb == 0) {
            // ---- block prologue: what was loaded during the previous block goes to LDS, the next block's loads leave
            if (cb > 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); store_otile(cb - 1); }
            write_wt();
            write_mtile(cb);
            load_wraw(cb + 1);
            load_mtile(cb + 1 < NB ? cb + 1 : cb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const float* mt = lds + CS_MT + (cb & 1) * 1024 + r * 32;
#pragma unroll
        for (int g8 = 0; g8 < 2; ++g8) {
            // ===================================================== group prologue (every 8 columns)
            const int G = 2 * hb + g8;
            // (1) certification of the last group
            if (__builtin_expect(abortw || __any(abortf && cert_lane), 0)) {
                if (lane == 0 && atomicOr(&p.abort_words[dirslot * p.B + b], 1u) == 0u) atomicAdd(&p.counters[2], 1u);
                abortf = false; abortw = false;
            }
            lossage = lostg ? 5 : (lossage > 0 ? lossage - 1 : 0);
            lostg = false;
            lobits = lossage > 0 ? 0x0F000000u : 0x03000000u;
            // (2) window scale: keep the largest window value's frexp exponent inside [ELO, EHI]
            {
                float wm = fmaxf(fmaxf(fmaxf(Wn[0], Wn[1]), fmaxf(Wn[2], Wn[3])), fmaxf(fmaxf(Wn[4], Wn[5]), fmaxf(Wn[6], Wn[7])));
                wm = fmaxf(wm, fmaxf(fmaxf(fmaxf(Wn[8], Wn[9]), fmaxf(Wn[10], Wn[11])), fmaxf(fmaxf(Wn[12], Wn[13]), fmaxf(Wn[14], Wn[15]))));
                wm = cs_half_max(wm);
                const int e = __builtin_amdgcn_frexp_expf(wm);
                has = wm > 0.f;
                const bool need = has && ((u32)(e - CS_ELO) > (u32)(CS_EHI - CS_ELO));
                if (__any(need)) {
                    const int dlt = has ? (e - CS_ETGT) : 0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) Wn[q] = ldexpf(Wn[q], -dlt);
                    Y += dlt;
                }
            }
            // (3) this group's arrivals for the wave's first row (lanes 0..7 = columns 8G .. 8G+7)
            float gx; int ge;
            if (has_prod) {
                u64 x = gq[g8 & 1];
                auto all_here = [&](u64 y) { return __all(lane >= 8 || (8 * G + lane) >= Lceil || (u32)(y >> 32) != 0u); };
                if (!all_here(x)) {
                    // The producer is less than two groups + one memory round trip ahead, so the prefetch missed.  Fall back far enough
                    // that the following prefetches hit (the waves run at the same speed: the lag then stays): wait for the LAST column
                    // of group G+3, then take this group and the next afresh.
                    const int vw = min(8 * (G + 3) + 7, Lceil - 1);
                    u32 spins = 0;
                    while ((u32)(cs_gran_load(gin + vw) >> 32) == 0u) {
                        if (++spins > CS_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    x = load_group(G);
                    gq[(g8 + 1) & 1] = load_group(G + 1);
                    spins = 0;
                    while (!all_here(x)) {
                        x = load_group(G);
                        if (++spins > CS_SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 2u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                gx = __uint_as_float((u32)x);
                ge = (int)((u32)(x >> 32) >> 1) - (1 << 30);
            } else {
                gx = (8 * G + lane == off - 1) ? 1.f : 0.f;
                ge = 0;
            }
            gq[g8 & 1] = load_group(G + 2);

#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                // ================================================= one column step
                const int s = 8 * g8 + k8;               // slot of column v (written at the end of the step); static
                const int sv = svb + s;                  // step inside the 32-column block
                const int v = vb + sv;
                const int sn = (s + 15) & 15;            // slot of column v-1: the newest value of the lower half
                if ((s & 3) == 0) {                       // the row's next four emissions
                    const int c = BETA ? (7 - (sv >> 2)) : (sv >> 2);
                    mcur = *reinterpret_cast<const v4f*>(mt + ((c ^ swz) << 2));
                }
                // -- transition weights of column v for this half: 4 broadcast ds_read_b128
                const float* wp = wt + sv * 32 + h * 16;
                const v4f w0 = *reinterpret_cast<const v4f*>(wp), w1 = *reinterpret_cast<const v4f*>(wp + 4);
                const v4f w2 = *reinterpret_cast<const v4f*>(wp + 8), w3 = *reinterpret_cast<const v4f*>(wp + 12);
                const v2f wpair[8] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}, {w2.x, w2.y}, {w2.z, w2.w}, {w3.x, w3.y}, {w3.z, w3.w}};
                // -- emission of (row, column v): exp-space factor split into 2^mi * ef, ef in [1, 2)
                const int mc = BETA ? (3 - (s & 3)) : (s & 3);
                float m = mc == 0 ? mcur.x : mc == 1 ? mcur.y : mc == 2 ? mcur.z : mcur.w;
                if (BETA) m = (v > vmax_u) ? NEG_INF : m;                    // beta is only defined for column >= row
                const float m2 = m * CS_LOG2E;
                const float mif = floorf(fmaxf(m2, -1048576.f));
                const float ef = __builtin_amdgcn_exp2f(m2 - mif);           // m = -inf -> 0
                const int mi = (int)mif;
                // -- the sum: the pair holding the newest value last
                v2f accA = {0.f, 0.f}, accB = {0.f, 0.f};
                const int ip = sn >> 1;
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i == ip) continue;
                    const v2f x = {Wn[2 * i], Wn[2 * i + 1]};
                    if (cnt & 1) accB = __builtin_elementwise_fma(x, wpair[i], accB); else accA = __builtin_elementwise_fma(x, wpair[i], accA);
                    ++cnt;
                }
                accA += accB;
                { const v2f x = {Wn[2 * ip], Wn[2 * ip + 1]}; accA = __builtin_elementwise_fma(x, wpair[ip], accA); }
                const float S = cs_half_sum(accA.x + accA.y);
                // -- certification: lo <= S < 2^126, or S == 0 with no live predecessor inside the window of transitions
                {
                    const u32 sb = __float_as_uint(S);
                    const bool inr = (sb - lobits) < (0x7E800000u - lobits);
                    const bool bad = !inr && !(S == 0.f && lastnz < v - TR);
#ifdef CS_DEBUG
                    if (p.dbg && bad && cert_lane) {
                        const u32 slot = atomicAdd(&p.counters[3], 1u);
                        if (slot < 6) printf("colsweep abort: b=%d beta=%d u=%d v=%d S=%g lossage=%d Y=%d lastnz=%d\n", b, (int)BETA, u, v, S, lossage, Y, lastnz);
                    }
#endif
                    abortf |= bad;
                }
                // -- output
                const float Yf = (float)Y;
                const float cc = fmaf(Yf, CS_LN2, m);
                oq[s & 3] = fmaf(__builtin_amdgcn_logf(S), CS_LN2, cc);
                if ((s & 3) == 3) {
                    const int c = BETA ? (7 - (sv >> 2)) : (sv >> 2);
                    const v4f o4 = BETA ? (v4f){oq[3], oq[2], oq[1], oq[0]} : (v4f){oq[0], oq[1], oq[2], oq[3]};
                    *reinterpret_cast<v4f*>(otile_w + (h == 0 ? ((c ^ swz) << 2) : 0)) = o4;
                }
                // -- send (value, exponent) to the next row; the wave's last row publishes it
                const float xs = S * ef;
                const int Es = mi + Y;
                if (pub_lane) cs_gran_store(gout + v, xs, Es);
                float xr = cs_shr1(xs);
                int Er = cs_shr1(Es);
                {
                    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gx), k8));
                    const int se = __builtin_amdgcn_readlane(ge, k8);
                    xr = lane == 0 ? sx : xr;
                    Er = lane == 0 ? se : Er;
                }
                // -- receive: rescale into this row's window scale; an empty row adopts the arrival's scale
                const bool nz = xr != 0.f;
                const int exr = __builtin_amdgcn_frexp_expf(xr);
                int sh = has ? (Er - Y) : (CS_ETGT - exr);
                Y = has ? Y : (Er - sh);
                // an arrival far above the window (rows are rough: 2^20 from one column to the next happens) would overflow the
                // next sums: bring the window down first
                if (__builtin_expect(__any(h == 0 && nz && (exr + sh > CS_GUARD)), 0)) {
                    int dlt = (nz && (exr + sh > CS_GUARD)) ? (exr + sh - CS_ETGT) : 0;
                    dlt = __builtin_bit_cast(int, cs_lower_to_upper(__builtin_bit_cast(float, dlt)));        // the row's two halves move together
                    dlt = h == 0 ? ((nz && (exr + sh > CS_GUARD)) ? (exr + sh - CS_ETGT) : 0) : dlt;
#pragma unroll
                    for (int q = 0; q < 16; ++q) Wn[q] = ldexpf(Wn[q], -dlt);
                    Y += dlt; sh -= dlt;
                }
                const float xin = ldexpf(xr, sh);
                has |= nz;
                lastnz = (nz && u != 0) ? v : lastnz;
                lostg |= nz && (xin < 0x1p-126f);
#ifdef CS_DEBUG
                if (p.dbg && p.trace_b == b && p.trace_beta == (int)BETA && p.trace_u == u && h == 0 && v >= u - 3 && v <= u + 14)
                    printf("trace v=%d S=%g m=%g ef=%g mi=%d xs=%g Es=%d | xr=%g Er=%d exr=%d sh=%d xin=%g Y=%d has=%d\n", v, S, m, ef, mi, xs, Es, xr, Er, exr, sh, xin, Y, (int)has);
#endif
                // -- window update: the lower half takes the arrival, the upper half the value the lower half retires
                const float mig = cs_lower_to_upper(Wn[s]);
                Wn[s] = h == 0 ? xin : mig;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_otile(NB - 1);

    // ---- what the sweep does not reach
    for (int rr = 0; rr < 32; ++rr) {
        const int uu = u0 + rr;
        if (uu >= T) break;
        if (uu < Tb) { if (Lceil < L) cs_fill_row(O + (size_t)(BETA ? (Tb - 1 - uu) : uu) * L, Lceil, L, lane); }
        else cs_fill_row(O + (size_t)uu * L, 0, L, lane);
    }
    if ((abortw || __any(abortf && cert_lane)) && lane == 0 && atomicOr(&p.abort_words[dirslot * p.B + b], 1u) == 0u) atomicAdd(&p.counters[2], 1u);
}

__global__ __launch_bounds__(64 * CS_NW) void dag_colsweep_kernel(CSParams p)
{
    __shared__ __attribute__((aligned(16))) float s_lds[CS_NW][CS_LDS_WAVE];
    __shared__ u32 s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = s_ticket;
    const int per = p.ndir * p.B;
    const int rb = (int)(ticket / per), rem = (int)(ticket % per);
    const bool is_beta = p.alpha == nullptr || (p.ndir == 2 && rem >= p.B);
    const int b = rem % p.B;
    const int dirslot = (p.ndir == 2 && rem >= p.B) ? 1 : 0;
    const int wave = threadIdx.x >> 6;
    const int gw = rb * CS_NW + wave;
    if (is_beta) colsweep_body<true>(p, s_lds[wave], b, gw, dirslot);
    else colsweep_body<false>(p, s_lds[wave], b, gw, dirslot);
}

// ------------------------------------------------------------------------------------------------ host side
int colsweep_acquire_ws(hipStream_t st, size_t abort_bytes, size_t gran_bytes, u32** counters, u32** abort_words, u64** gran);

bool colsweep_supported(const void* match, const void* alpha, const void* beta, int B, int T, int L, int TR)
{
    if (TR > 32 || (L & 3) || L < 32) return false;
    const uintptr_t a = (uintptr_t)match | (uintptr_t)alpha | (uintptr_t)beta;
    if (a & 15) return false;
    (void)B; (void)T;
    return true;
}

// waves the sweep keeps busy: below ~600 the row-sweep strips fill the chip better
long colsweep_waves(const void* alpha, const void* beta, int B, int T) { return (long)((alpha && beta) ? 2 : 1) * B * ((T + 31) / 32); }

size_t colsweep_gran_bytes(int B, int T, int L, int ndir)
{
    const size_t NWT = (size_t)(T + 31) / 32, LP = (size_t)((L + 31) / 32) * 32;
    return (size_t)ndir * B * NWT * LP * sizeof(u64);
}

int launch_dag_colsweep(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                        float* alpha, float* beta, int B, int T, int L, int TR, u32** abort_words_out, u32** counters_out, hipStream_t st)
{
    CSParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len; p.alpha = alpha; p.beta = beta;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.ndir = (alpha && beta) ? 2 : 1;
    p.NWT = (T + 31) / 32; p.NRB = (p.NWT + CS_NW - 1) / CS_NW; p.LP = ((L + 31) / 32) * 32;
    { const char* e = getenv("DSP_DEBUG"); p.dbg = (e && !strcmp(e, "cs")) ? 1 : (e && !strcmp(e, "cs_nohandoff")) ? 2 : 0; }
    p.trace_b = p.trace_beta = p.trace_u = -1;
    { const char* e = getenv("DSP_CS_TRACE"); if (e) sscanf(e, "%d,%d,%d", &p.trace_b, &p.trace_beta, &p.trace_u); }
    int rc = colsweep_acquire_ws(st, (size_t)p.ndir * B * 4, colsweep_gran_bytes(B, T, L, p.ndir), &p.counters, &p.abort_words, &p.gran);
    if (rc) return rc;
    if (abort_words_out) *abort_words_out = p.abort_words;
    if (counters_out) *counters_out = p.counters;
    const int nwg = p.ndir * B * p.NRB;
    hipLaunchKernelGGL(dag_colsweep_kernel, dim3((unsigned)nwg), dim3(64 * CS_NW), 0, st, p);
    return check_launch("dag_loss_fwd(colsweep)");
}

}  // namespace dsp
