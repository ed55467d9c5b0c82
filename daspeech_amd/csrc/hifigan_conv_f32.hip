// hifigan_conv_f32.hip — the HiFi-GAN generator layer at the REFERENCE's precision (fp32 activations and weights,
// hifi-gan/models.py:100-119 run in fp32 by inference_e2e.py:47-56) on the fp16 matrix cores, by operand splitting.
//
// hifigan_conv.hip stores activations and weights in fp16 (fp32 accumulate): 1.5e-3 off the reference waveform.  This file is the
// same layer formulation (include/daspeech_hifigan.h: taps, shifts, STORE / ACCUM / UPSAMPLE epilogues, per-sample lengths) with
//     x = xh + xl / 2048,   w = wh + wl / 2048          (xh, xl, wh, wl fp16; the split is exact to 2^-22 relative)
//     acc_main += wh . xh          acc_corr += wh . xl + wl . xh          out = acc_main + acc_corr / 2048
// Every product is exact in the fp32 accumulator; the dropped wl . xl term is 2^-22 of the result.  Activations live in HBM as fp32
// (channels-last [B][T][C]); the leaky_relu and the split happen once per element while the input tile is staged into two LDS tiles
// (hi, lo; the XOR swizzle of hifigan_conv.hip), weights come pre-split in MFMA fragment order.  Three MFMAs per fragment pair and
// twice the activation bytes: the price of the reference's arithmetic (measured beside the fp16-storage path in bench.py).
// Range: |x|, |w| < 65504 (fp16 hi part); values under 6e-5 keep fewer than 22 bits — both far from what a vocoder holds.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/daspeech_hifigan.h"

namespace dsp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct HgsParams {
    const float* x; const _Float16* wh; const _Float16* wl; const float* bias; const float* res; float* out;
    int B, T, M, ntaps, Tout, Cout, out_mode, up_u, up_pad;
    float pre_slope, scale;
    int shifts[DSP_HG_MAX_TAPS];
    int min_shift, max_shift;
    const int* lens; int len_mul;
};

constexpr float HGS_LO = 2048.f, HGS_LO_INV = 1.f / 2048.f;

__device__ __forceinline__ int hgs_valid_len(const int* lens, int len_mul, int b, int T) {
    if (!lens) return T;
    const int v = lens[b] * len_mul;
    return v < T ? v : T;
}

template <int CI>
__device__ __forceinline__ int hgs_swz(int row, int chunk) {
    constexpr int CH = CI / 8;                          // 16-byte chunks per row
    // ds_read_b128 is serviced in four NON-contiguous groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md
    // §LDS): a B-fragment read puts 8 rows at k-chunk q and the other 8 rows of the same 16 at chunk q ^ 1 into one group.  The r01
    // swizzle (chunk ^ row) is conflict-free for 16 rows at ONE chunk; with the real groups it collides whenever the tile row of
    // lane 0 is odd (every odd tap shift): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.27 - 0.46 (profiles/r03f_pmc_hifigan.txt).
    // XOR-ing only EVEN values leaves bit 0 of the slot to tell the two halves of a group apart, and 8 rows x 8 even values are
    // distinct for any base row: conflict-free for every shift.  (256-byte bank row = 16 slots of 16 bytes; rows narrower than that
    // share a bank row: the row's position inside it supplies the remaining slot bits.)
    if constexpr ((CH & (CH - 1)) != 0) return chunk;   // CI = 96: 12 chunks, not a power of two -> no swizzle
    else if constexpr (CH >= 16) return chunk ^ ((row & 7) << 1);
    else if constexpr (CH == 8) return chunk ^ (((row >> 1) & 3) << 1);
    else if constexpr (CH == 4) return chunk ^ (((row >> 2) & 1) << 1);
    else return chunk;
}

// rows [t_first, t_first + R) of the fp32 channels-last input -> lrelu -> (hi, lo) fp16 tiles
template <int CI, int U>
__device__ __forceinline__ void hgs_stage_tile(char* thi, char* tlo, const float* __restrict__ X, int T, int t_first, int R, float slope, int tid)
{
    constexpr int CH = CI / 8;
    const int n = R * CH;
    for (int e0 = tid; e0 < n; e0 += 512 * U) {
        f4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 512;
            const int row = e / CH, ch = e - row * CH;
            const int tg = t_first + row;
            va[u] = (f4){0.f, 0.f, 0.f, 0.f}; vb[u] = (f4){0.f, 0.f, 0.f, 0.f};
            if (e < n && tg >= 0 && tg < T) {
                const float* src = X + (size_t)tg * CI + ch * 8;
                va[u] = *reinterpret_cast<const f4*>(src); vb[u] = *reinterpret_cast<const f4*>(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 512;
            if (e < n) {
                const int row = e / CH, ch = e - row * CH;
                h8 hi, lo;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = i < 4 ? va[u][i] : vb[u][i - 4];
                    v = v > 0.f ? v : v * slope;
                    const _Float16 h = (_Float16)v;
                    hi[i] = h;
                    lo[i] = (_Float16)((v - (float)h) * HGS_LO);
                }
                const size_t o = ((size_t)row * CH + hgs_swz<CI>(row, ch)) * 16;
                *reinterpret_cast<h8*>(thi + o) = hi;
                *reinterpret_cast<h8*>(tlo + o) = lo;
            }
        }
    }
}

template <int CI, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(512) void hifigan_conv_f32_kernel(HgsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = CI / 8;
    constexpr int MI = MT / WM / 16, NI = NT / WN / 16;
    static_assert(WM * WN == 8 && MI >= 1 && NI >= 1, "8 waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * NT;
    const int m0 = blockIdx.y * MT;
    const int R = NT + (p.max_shift - p.min_shift);
    const float* X = p.x + (size_t)b * p.T * CI;
    const int Tb = hgs_valid_len(p.lens, p.len_mul, b, p.T);
    if (p.lens) {                                         // nobody reads past a sample's valid length (see hifigan_conv_kernel)
        if (p.out_mode == DSP_HG_OUT_UPSAMPLE ? (t0 * p.up_u - p.up_pad >= Tb * p.up_u) : (t0 >= Tb)) return;
    }
    char* thi = smem;
    char* tlo = smem + (size_t)R * CI * 2;
    hgs_stage_tile<CI, 4>(thi, tlo, X, Tb, t0 + p.min_shift, R, p.pre_slope, tid);
    __syncthreads();

    f4 accm[MI][NI], accc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) { accm[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; accc[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; }

    const int co_base = m0 + wm * (MI * 16);
    const int tl_base = wn * (NI * 16);
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NC = CI / 32;
    const int nsteps = p.ntaps * NC;
    const int Mt = (p.M + 15) >> 4;
    auto load_a = [&](int step, h8 (&ah)[MI], h8 (&al)[MI]) {
        const size_t o = (size_t)step * Mt * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int tile = (co_base >> 4) + i;
            const bool in = tile < Mt;
            ah[i] = in ? *reinterpret_cast<const h8*>(p.wh + o + (size_t)tile * 512) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
            al[i] = in ? *reinterpret_cast<const h8*>(p.wl + o + (size_t)tile * 512) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    h8 a0h[MI], a0l[MI], a1h[MI], a1l[MI];
    if (nsteps > 0) load_a(0, a0h, a0l);
    auto do_step = [&](int step, const h8 (&ah)[MI], const h8 (&al)[MI]) {
        const int k = step / NC, c = step - k * NC;
        const int rshift = p.shifts[k] - p.min_shift;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = tl_base + j * 16 + lr + rshift;
            const size_t o = ((size_t)row * CH + hgs_swz<CI>(row, c * 4 + lk)) * 16;
            const h8 bh = *reinterpret_cast<const h8*>(thi + o);
            const h8 bl = *reinterpret_cast<const h8*>(tlo + o);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, accm[i][j], 0, 0, 0);
                accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, accc[i][j], 0, 0, 0);
                accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, accc[i][j], 0, 0, 0);
            }
        }
    };
    for (int step = 0; step < nsteps; step += 2) {          // two-deep weight ring (static register names)
        if (step + 1 < nsteps) load_a(step + 1, a1h, a1l);
        do_step(step, a0h, a0l);
        if (step + 1 < nsteps) {
            if (step + 2 < nsteps) load_a(step + 2, a0h, a0l);
            do_step(step + 1, a1h, a1l);
        }
    }

    // ---- epilogue through LDS: [NT][MT] fp32 tile, then row-contiguous 16-byte residual / accumulate loads and stores ----
    constexpr int OPITCH = MT + 4;
    __syncthreads();
    float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wm * (MI * 16) + i * 16 + lk * 4;
        const int mrow = m0 + ml;
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && mrow < p.M) {
            const int co = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? (mrow % p.Cout) : mrow;
            bv = *reinterpret_cast<const f4*>(p.bias + co);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int tl = tl_base + j * 16 + lr;
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = accm[i][j][e] + accc[i][j][e] * HGS_LO_INV + bv[e];
            *reinterpret_cast<f4*>(otile + (size_t)tl * OPITCH + ml) = v;
        }
    }
    __syncthreads();
    constexpr int CPR = MT / 4;
    constexpr int EU = 4;
    for (int e0 = tid; e0 < NT * CPR; e0 += 512 * EU) {
        f4 r4[EU], a4[EU];
        size_t off[EU];
        bool live[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            const int mrow = m0 + ch * 4;
            const int q = t0 + tl;
            int tout = q, co = mrow;
            if (p.out_mode == DSP_HG_OUT_UPSAMPLE) { const int r = mrow / p.Cout; co = mrow - r * p.Cout; tout = q * p.up_u + r - p.up_pad; }
            live[u] = e < NT * CPR && mrow < p.M && tout >= 0 && tout < p.Tout;
            off[u] = ((size_t)b * p.Tout + tout) * p.Cout + co;
            r4[u] = (f4){0.f, 0.f, 0.f, 0.f}; a4[u] = (f4){0.f, 0.f, 0.f, 0.f};
            if (live[u] && p.res) r4[u] = *reinterpret_cast<const f4*>(p.res + off[u]);
            if (live[u] && p.out_mode == DSP_HG_OUT_ACCUM) a4[u] = *reinterpret_cast<const f4*>(p.out + off[u]);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (!live[u]) continue;
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            const f4 v = *reinterpret_cast<const f4*>(otile + (size_t)tl * OPITCH + ch * 4);
            f4 w4;
#pragma unroll
            for (int x = 0; x < 4; ++x) w4[x] = p.scale * (v[x] + r4[u][x]) + a4[u][x];
            *reinterpret_cast<f4*>(p.out + off[u]) = w4;
        }
    }
}

template <int CI, int MT, int NT, int WM, int WN>
static int hgs_launch(const HgsParams& p, hipStream_t st)
{
    const int R = NT + (p.max_shift - p.min_shift);
    size_t lds = (size_t)R * CI * 4;
    const size_t lds_out = (size_t)NT * (MT + 4) * 4;
    if (lds_out > lds) lds = lds_out;
    if (lds > 160 * 1024) { set_error("hifigan_conv_f32: tiles of %zu bytes exceed LDS (CI=%d, halo %d)", lds, CI, p.max_shift - p.min_shift); return DSP_EINVAL; }
    auto k = hifigan_conv_f32_kernel<CI, MT, NT, WM, WN>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    const int ncol = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? p.T + 1 : p.T;
    dim3 grid((ncol + NT - 1) / NT, (p.M + MT - 1) / MT, p.B);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
    return check_launch("hifigan_conv_f32");
}


// ---------------------------------------------------------------------------------------------------------------------------
// One ResBlock1 unit (hifi-gan/models.py:38-42) at fp32 accuracy in ONE launch: out = scale * (x + b2 + c2(lrelu(b1 + c1(lrelu(x))))) [+ out].
// hifigan_resunit_kernel's structure (x tile -> c1 over NT+16 intermediate columns -> intermediate in LDS -> c2 -> output tile, one LDS
// region with three tenants in turn) with split operands: the x tile and the intermediate each live as (hi, lo) fp16 planes, the
// intermediate is split from the fp32 accumulators exactly where the layer chain would split the fp32 tensor it stores, so the result
// is bit-identical to the two hifigan_conv_f32 launches it replaces — without their fp32 round trip of the intermediate through HBM.
struct HgsUnitParams {
    const float* x; const _Float16* w1; const float* b1; const _Float16* w2; const float* b2; float* out;     // w1 / w2: [hi | lo] packed
    int B, T, ntaps, dil, accumulate;
    float slope, scale;
    const int* lens; int len_mul;
};

template <int C, int NT, int WM, int WN>
__global__ __launch_bounds__(512, ((C == 32 || C == 128) ? 4 : 2)) void hifigan_resunit_f32_kernel(HgsUnitParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = C / 8, NC = C / 32;
    constexpr int NTI = NT + 16;
    constexpr int MI = C / WM / 16, NI = NTI / WN / 16;
    constexpr int OPITCH = C + 4;
    static_assert(WM * WN == 8 && MI >= 1 && NI >= 1 && NTI % (WN * 16) == 0, "8 waves, intermediate tile divisible");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z, t0 = blockIdx.x * NT;
    const int h1 = p.dil * (p.ntaps - 1) / 2, h2 = (p.ntaps - 1) / 2;
    const int R1 = NTI + 2 * h1;
    constexpr int RM = NTI + 16;                            // intermediate rows + the slack c2's last 16 columns read (never stored)
    const float* X = p.x + (size_t)b * p.T * C;
    const int Tb = hgs_valid_len(p.lens, p.len_mul, b, p.T);
    if (t0 >= Tb) return;
    char* xhi = smem; char* xlo = smem + (size_t)R1 * C * 2;
    char* mhi = smem; char* mlo = smem + (size_t)RM * C * 2;
    hgs_stage_tile<C, 4>(xhi, xlo, X, Tb, t0 - 8 - h1, R1, p.slope, tid);
    __syncthreads();

    f4 accm[MI][NI], accc[MI][NI];
    const int co_base = wm * (MI * 16);
    const int nsteps = p.ntaps * NC;
    const size_t wn_elems = (size_t)nsteps * (C / 16) * 512;       // halves in the hi (and in the lo) part of a packed weight buffer
    // NJ: 16-column tiles this wave computes (c2 needs NT of the NTI columns: with one wave column the last tile is skipped — a quarter
    // of c2's MFMAs at C = 256, an eighth at C = 128)
    auto conv = [&](auto njc, const _Float16* W, const char* thi, const char* tlo, int row0, int rstep) {
        constexpr int NJ = decltype(njc)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) { accm[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; accc[i][j] = (f4){0.f, 0.f, 0.f, 0.f}; }
        auto load_a = [&](int step, h8 (&ah)[MI], h8 (&al)[MI]) {
            const _Float16* Ws = W + (size_t)step * (C / 16) * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const h8*>(Ws + (size_t)((co_base >> 4) + i) * 512);
                al[i] = *reinterpret_cast<const h8*>(Ws + wn_elems + (size_t)((co_base >> 4) + i) * 512);
            }
        };
        auto do_step = [&](int step, const h8 (&ah)[MI], const h8 (&al)[MI]) {
            const int k = step / NC, c = step - k * NC;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = (wn * NI + j) * 16 + lr + row0 + k * rstep;
                const size_t o = ((size_t)row * CH + hgs_swz<C>(row, c * 4 + lk)) * 16;
                const h8 bh = *reinterpret_cast<const h8*>(thi + o);
                const h8 bl = *reinterpret_cast<const h8*>(tlo + o);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    accm[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, accm[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, accc[i][j], 0, 0, 0);
                    accc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, accc[i][j], 0, 0, 0);
                }
            }
        };
        h8 a0h[MI], a0l[MI], a1h[MI], a1l[MI];
        load_a(0, a0h, a0l);
        for (int step = 0; step < nsteps; step += 2) {
            if (step + 1 < nsteps) load_a(step + 1, a1h, a1l);
            do_step(step, a0h, a0l);
            if (step + 1 < nsteps) {
                if (step + 2 < nsteps) load_a(step + 2, a0h, a0l);
                do_step(step + 1, a1h, a1l);
            }
        }
    };

    // ---- c1 over the NTI intermediate columns -> mid = split(lrelu(acc + b1)), zero outside [0, Tb) ----
    conv(std::integral_constant<int, NI>{}, p.w1, xhi, xlo, 0, p.dil);
    __syncthreads();                                  // every wave is done reading the x tile: its space becomes the intermediate
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int co = co_base + i * 16 + lk * 4;
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.b1) bv = *reinterpret_cast<const f4*>(p.b1 + co);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int m = (wn * NI + j) * 16 + lr;
            const int tm = t0 - 8 + m;
            _Float16 hv[4], lv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = accm[i][j][e] + accc[i][j][e] * HGS_LO_INV + bv[e];
                v = v > 0.f ? v : v * p.slope;
                if (!(tm >= 0 && tm < Tb)) v = 0.f;
                const _Float16 hh = (_Float16)v;
                hv[e] = hh; lv[e] = (_Float16)((v - (float)hh) * HGS_LO);
            }
            const size_t o = ((size_t)m * CH + hgs_swz<C>(m, co >> 3)) * 16 + (co & 4) * 2;
            *reinterpret_cast<uint2*>(mhi + o) = *reinterpret_cast<uint2*>(hv);
            *reinterpret_cast<uint2*>(mlo + o) = *reinterpret_cast<uint2*>(lv);
        }
    }
    __syncthreads();                                  // intermediate complete

    // ---- c2 over the NT output columns ----
    conv(std::integral_constant<int, (WN == 1 ? NT / 16 : NI)>{}, p.w2, mhi, mlo, 8 - h2, 1);
    __syncthreads();                                  // every wave is done reading the intermediate: its space becomes the output tile
    float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = co_base + i * 16 + lk * 4;
        f4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.b2) bv = *reinterpret_cast<const f4*>(p.b2 + ml);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (wn * NI + j < NT / 16) {
                const int tl = (wn * NI + j) * 16 + lr;
                f4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accm[i][j][e] + accc[i][j][e] * HGS_LO_INV + bv[e];
                *reinterpret_cast<f4*>(otile + (size_t)tl * OPITCH + ml) = v;
            }
        }
    }
    __syncthreads();
    constexpr int CPR = C / 4, EU = 4;
    for (int e0 = tid; e0 < NT * CPR; e0 += 512 * EU) {
        f4 r4[EU], a4[EU];
        bool live[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            live[u] = e < NT * CPR && t0 + tl < p.T;
            const size_t o = ((size_t)b * p.T + t0 + tl) * C + ch * 4;
            r4[u] = (f4){0.f, 0.f, 0.f, 0.f}; a4[u] = (f4){0.f, 0.f, 0.f, 0.f};
            if (live[u]) r4[u] = *reinterpret_cast<const f4*>(p.x + o);              // the unit's residual is its own input
            if (live[u] && p.accumulate) a4[u] = *reinterpret_cast<const f4*>(p.out + o);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (!live[u]) continue;
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            const size_t o = ((size_t)b * p.T + t0 + tl) * C + ch * 4;
            const f4 v = *reinterpret_cast<const f4*>(otile + (size_t)tl * OPITCH + ch * 4);
            f4 w4;
#pragma unroll
            for (int x = 0; x < 4; ++x) w4[x] = p.scale * (v[x] + r4[u][x]) + a4[u][x];
            *reinterpret_cast<f4*>(p.out + o) = w4;
        }
    }
}

static size_t hgs_unit_lds(int C, int NT, int h1)
{
    const size_t xin = (size_t)(NT + 16 + 2 * h1) * C * 4, mid = (size_t)(NT + 32) * C * 4, ot = (size_t)NT * (C + 4) * 4;
    const size_t m = xin > mid ? xin : mid;
    return ((m > ot ? m : ot) + 255) / 256 * 256;
}

template <int C, int NT, int WM, int WN>
static int hgs_unit_launch(const HgsUnitParams& p, hipStream_t st)
{
    const int h1 = p.dil * (p.ntaps - 1) / 2;
    const size_t lds = hgs_unit_lds(C, NT, h1);
    if (lds > 160 * 1024) { set_error("hifigan_resunit_f32: tiles need %zu bytes of LDS", lds); return DSP_EINVAL; }
    auto k = hifigan_resunit_f32_kernel<C, NT, WM, WN>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((p.T + NT - 1) / NT, 1, p.B), dim3(512), lds, st, p);
    return check_launch("hifigan_resunit_f32");
}

static int hgs_unit_nt(int C) { return C == 32 ? 496 : C == 64 ? 240 : C == 128 ? 112 : 48; }

// fp32 tap-major [ntaps][M][CI] -> (hi, lo * 2048) in the fragment order of hifigan_conv.hip's hg_pack_weights_kernel
__global__ void hgs_pack_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ oh, _Float16* __restrict__ ol, int ntaps, int M, int CI)
{
    const int Mt = (M + 15) >> 4, NC = CI / 32;
    const long n = (long)ntaps * NC * Mt * 512;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e & 7), ln = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int tile = (int)(r % Mt); r /= Mt;
        const int c = (int)(r % NC); const int k = (int)(r / NC);
        const int co = tile * 16 + (ln & 15), ci = c * 32 + (ln >> 4) * 8 + h;
        const float v = (co < M) ? w[((size_t)k * M + co) * CI + ci] : 0.f;
        const _Float16 hi = (_Float16)v;
        oh[e] = hi;
        ol[e] = (_Float16)((v - (float)hi) * HGS_LO);
    }
}

__global__ void hgs_pad_kernel(const float* __restrict__ x, float* __restrict__ out, long n_rows, int C, int Cpad)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * Cpad; e += (long)gridDim.x * blockDim.x) {
        const long r = e / Cpad; const int c = (int)(e - r * Cpad);
        out[e] = (c < C) ? x[r * C + c] : 0.f;
    }
}

// conv_post + tanh.  A workgroup takes 256 consecutive time steps of one sample: the (256 + K - 1) x C input rows are staged once,
// coalesced, with the leaky_relu applied, into an LDS tile of row pitch C + 1 words (row-per-lane reads then hit distinct banks); every
// thread then owns one output.  (The row-per-thread global reads this replaces ran at 150 GB/s: 2.3 ms of a 27 ms vocoder call.)
__global__ __launch_bounds__(256) void hgs_post_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                       float* __restrict__ wav, int T, int C, int K, float slope,
                                                       const int* __restrict__ lens, int len_mul)
{
    extern __shared__ float ps[];              // [K][C] weights, then [256 + K - 1][C + 1] rows
    float* ws = ps;
    float* tile = ps + K * C;
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const int Tb = hgs_valid_len(lens, len_mul, b, T);
    const int R = 256 + K - 1, P = C + 1, half = (K - 1) / 2;
    const float* X = x + (size_t)b * T * C;
    for (int i = tid; i < K * C; i += 256) ws[i] = w[i];
    const int C4 = C >> 2;
    for (int e = tid; e < R * C4; e += 256) {
        const int r = e / C4, c4 = e - r * C4;
        const int tg = t0 - half + r;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (tg >= 0 && tg < Tb) v = *reinterpret_cast<const f4*>(X + (size_t)tg * C + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float f = v[i]; tile[r * P + 4 * c4 + i] = f > 0.f ? f : f * slope; }
    }
    __syncthreads();
    const int t = t0 + tid;
    if (t >= T) return;
    if (t >= Tb) { wav[(size_t)b * T + t] = 0.f; return; }        // past the utterance: silence (the layers above skipped it)
    float acc = bias;
    for (int k = 0; k < K; ++k) {
        const float* row = tile + (tid + k) * P;
        const float* wk = ws + k * C;
        for (int c = 0; c < C; ++c) acc = fmaf(row[c], wk[c], acc);
    }
    wav[(size_t)b * T + t] = tanhf(acc);
}

}  // namespace dsp

using namespace dsp;

extern "C" int dsp_hifigan_resunit_f32_supported(int C, int ntaps, int dil);

static int hgs_conv_one(const dsp_hg_layer& l, int B, hipStream_t st, const int* lens, int len_mul)
{
    if (B < 0 || l.T < 1 || l.M < 1 || l.ntaps < 1 || l.ntaps > DSP_HG_MAX_TAPS) { set_error("hifigan_conv_f32: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!l.x || !l.w || !l.out) { set_error("hifigan_conv_f32: null pointer"); return DSP_EINVAL; }
    if ((l.Cout & 3) || (l.out_mode == DSP_HG_OUT_UPSAMPLE ? (l.M != l.up_u * l.Cout) : (l.M != l.Cout || l.Tout != l.T))) {
        set_error("hifigan_conv_f32: inconsistent M=%d Cout=%d mode=%d", l.M, l.Cout, l.out_mode); return DSP_EINVAL; }
    HgsParams p;
    p.x = (const float*)l.x; p.wh = (const _Float16*)l.w; p.wl = p.wh + dsp_hifigan_packed_weight_elems(l.ntaps, l.M, l.CI);      /* w = [hi | lo] */
    p.bias = l.bias; p.res = (const float*)l.res; p.out = (float*)l.out;
    p.B = B; p.T = l.T; p.M = l.M; p.ntaps = l.ntaps; p.Tout = l.Tout; p.Cout = l.Cout; p.out_mode = l.out_mode; p.up_u = l.up_u; p.up_pad = l.up_pad;
    p.pre_slope = l.pre_slope; p.scale = l.scale; p.lens = lens; p.len_mul = len_mul;
    p.min_shift = p.max_shift = l.shifts[0];
    for (int k = 0; k < l.ntaps; ++k) { p.shifts[k] = l.shifts[k]; p.min_shift = min(p.min_shift, l.shifts[k]); p.max_shift = max(p.max_shift, l.shifts[k]); }
    // tiles: the fp16-storage kernel's wave grids with the time tile cut to what two input tiles (hi, lo) and an fp32 output tile leave
    // of the 160 KB (the widest halo is the K = 11, dilation 5 unit: 50 rows)
    switch (l.CI) {
        case 512: return hgs_launch<512, 256, 64, 8, 1>(p, st);
        case 256: return hgs_launch<256, 256, 64, 8, 1>(p, st);
        case 128: return hgs_launch<128, 128, 256, 4, 2>(p, st);
        case 96:  return hgs_launch<96, 256, 128, 8, 1>(p, st);
        case 64:  return hgs_launch<64, 64, 512, 2, 4>(p, st);
        case 32:  return hgs_launch<32, 32, 512, 1, 8>(p, st);
    }
    set_error("hifigan_conv_f32: unsupported input channel count %d", l.CI);
    return DSP_EINVAL;
}

extern "C" int dsp_hifigan_conv_chain_f32(const dsp_hg_layer* layers, int n_layers, int B, const int* lens, int T0, dsp_stream_t stream)
{
    if (n_layers < 0 || (n_layers > 0 && !layers)) { set_error("hifigan_conv_chain_f32: bad layer table"); return DSP_EINVAL; }
    if (lens && T0 < 1) { set_error("hifigan_conv_chain_f32: lens given without the padded frame count T0"); return DSP_EINVAL; }
    for (int i = 0; i < n_layers; ++i) {
        const dsp_hg_layer& l = layers[i];
        int mul = 1;
        if (lens) {
            if (l.T % T0) { set_error("hifigan_conv_chain_f32: layer %d length %d is not a multiple of T0 = %d", i, l.T, T0); return DSP_EINVAL; }
            mul = l.T / T0;
        }
        if (l.w2) {                                                        // fused ResBlock unit: x -> c1 -> c2 -> + x
            const int dil = l.ntaps > 1 ? l.shifts[l.ntaps / 2 + 1] : 1;
            if (!dsp_hifigan_resunit_f32_supported(l.CI, l.ntaps, dil) || l.M != l.CI) { set_error("hifigan_conv_chain_f32: layer %d is not a supported fused unit", i); return DSP_EINVAL; }
            HgsUnitParams u;
            u.x = (const float*)l.x; u.w1 = (const _Float16*)l.w; u.b1 = l.bias; u.w2 = (const _Float16*)l.w2; u.b2 = l.bias2; u.out = (float*)l.out;
            u.B = B; u.T = l.T; u.ntaps = l.ntaps; u.dil = dil; u.accumulate = l.out_mode == DSP_HG_OUT_ACCUM; u.slope = l.pre_slope; u.scale = l.scale;
            u.lens = lens; u.len_mul = mul;
            if (!u.x || !u.out || u.x == u.out) { set_error("hifigan_conv_chain_f32: null or aliased pointer in unit %d", i); return DSP_EINVAL; }
            int rc;
            // (tile sweep of r03, ms per 32 x 329-frame call and stage — C=256: NT 32 / 48 / 64 = 5.30 / 4.69 / 5.04; C=128 as 8x1 waves: NT 64 / 80 / 96 /
            //  112 = 9.34 / 8.82 / 8.93 / 8.71, as 4x2 waves at 132 VGPRs and one workgroup per CU 9.8; C=64: <240,2,4> 4.40, <240,4,2> 4.63,
            //  <176,4,2> 4.87, <112,4,2> 5.38; C=32: <496,1,8> 2.88, <496,2,4> 3.02, <240,2,4> 3.04, <240,1,8> 3.43;
            //  r04: C=128 as 4x2 waves under the 128-VGPR bound, two workgroups per CU: 24.9 vs 21.2 ms per call — profiles/r04_vocoder_ablation.txt)
            switch (l.CI) {
                case 256: rc = hgs_unit_launch<256, 48, 8, 1>(u, as_stream(stream)); break;      // (32-column tiles for the wide-halo units, to fit two per CU: slower, 21.9 vs 21.2 ms per call)
                case 128: {
                    // two workgroups per CU need <= 80 KB each: the wide-halo units (k = 7 / 11 at dilation 3 / 5) take narrower tiles for it
                    const int h1u = dil * (l.ntaps - 1) / 2;
                    if (hgs_unit_lds(128, 112, h1u) <= 80 * 1024) rc = hgs_unit_launch<128, 112, 8, 1>(u, as_stream(stream));
                    else if (hgs_unit_lds(128, 96, h1u) <= 80 * 1024) rc = hgs_unit_launch<128, 96, 8, 1>(u, as_stream(stream));
                    else rc = hgs_unit_launch<128, 80, 8, 1>(u, as_stream(stream));
                    break;
                }
                case 64:  rc = hgs_unit_launch<64, 240, 2, 4>(u, as_stream(stream)); break;
                default:  rc = hgs_unit_launch<32, 496, 1, 8>(u, as_stream(stream)); break;
            }
            if (rc) return rc;
            continue;
        }
        int rc = hgs_conv_one(l, B, as_stream(stream), lens, mul);
        if (rc) return rc;
    }
    return DSP_OK;
}

extern "C" int dsp_hifigan_resunit_f32_supported(int C, int ntaps, int dil)
{
    if (!(C == 32 || C == 64 || C == 128 || C == 256) || ntaps < 1 || !(ntaps & 1) || ntaps > DSP_HG_MAX_TAPS || dil < 1) return 0;
    return hgs_unit_lds(C, hgs_unit_nt(C), dil * (ntaps - 1) / 2) <= 160 * 1024;
}

extern "C" int dsp_hifigan_pack_weights_f32(const float* w, void* w_hi_lo, int ntaps, int M, int CI, dsp_stream_t stream)
{
    const long n = dsp_hifigan_packed_weight_elems(ntaps, M, CI);
    if (n < 0 || !w || !w_hi_lo) { set_error("hifigan_pack_weights_f32: bad arguments"); return DSP_EINVAL; }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hgs_pack_weights_kernel, dim3(grid), dim3(256), 0, as_stream(stream), w, (_Float16*)w_hi_lo, (_Float16*)w_hi_lo + n, ntaps, M, CI);
    return check_launch("hifigan_pack_weights_f32");
}

extern "C" int dsp_hifigan_pad_input_f32(const float* x, float* out, int B, int T, int C, int Cpad, dsp_stream_t stream)
{
    if (B < 0 || T < 0 || C < 1 || Cpad < C) { set_error("hifigan_pad_input_f32: bad sizes"); return DSP_EINVAL; }
    const long n = (long)B * T;
    if (n == 0) return DSP_OK;
    int grid = (int)((n * Cpad + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hgs_pad_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, out, n, C, Cpad);
    return check_launch("hifigan_pad_input_f32");
}

extern "C" int dsp_hifigan_post_f32(const float* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                                    const int* lens, int len_mul, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || C < 4 || (C & 3) || K < 1) { set_error("hifigan_post_f32: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    const size_t lds = ((size_t)K * C + (size_t)(256 + K - 1) * (C + 1)) * 4;
    if (lds > 64 * 1024) { set_error("hifigan_post_f32: C=%d, K=%d too large for the LDS tile", C, K); return DSP_EINVAL; }
    hipLaunchKernelGGL(hgs_post_kernel, dim3((T + 255) / 256, B), dim3(256), lds, as_stream(stream), x, w, bias, wav, T, C, K, slope, lens, len_mul);
    return check_launch("hifigan_post_f32");
}
