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
    constexpr int CH = CI / 8;
    constexpr int RPB = (CI * 2 >= 256) ? 1 : 256 / (CI * 2);
    constexpr int MASK = (CH < 16 ? CH : 16) - 1;
    if constexpr ((CH & (CH - 1)) != 0) return chunk;
    else return chunk ^ ((row / RPB) & MASK);
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
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int ncol = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? p.T + 1 : p.T;
    dim3 grid((ncol + NT - 1) / NT, (p.M + MT - 1) / MT, p.B);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
    return check_launch("hifigan_conv_f32");
}

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

__global__ __launch_bounds__(256) void hgs_post_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                       float* __restrict__ wav, int T, int C, int K, float slope,
                                                       const int* __restrict__ lens, int len_mul)
{
    extern __shared__ float ws[];              // [K][C]
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const float* X = x + (size_t)b * T * C;
    const int Tb = hgs_valid_len(lens, len_mul, b, T);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        if (t >= Tb) { wav[(size_t)b * T + t] = 0.f; continue; }
        float acc = bias;
        for (int k = 0; k < K; ++k) {
            const int tt = t + k - (K - 1) / 2;
            if (tt < 0 || tt >= Tb) continue;
            const float* xr = X + (size_t)tt * C;
            for (int c = 0; c < C; c += 4) {
                const f4 v = *reinterpret_cast<const f4*>(xr + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) { float f = v[i]; f = f > 0.f ? f : f * slope; acc += f * ws[k * C + c + i]; }
            }
        }
        wav[(size_t)b * T + t] = tanhf(acc);
    }
}

}  // namespace dsp

using namespace dsp;

static int hgs_conv_one(const dsp_hg_layer& l, int B, hipStream_t st, const int* lens, int len_mul)
{
    if (B < 0 || l.T < 1 || l.M < 1 || l.ntaps < 1 || l.ntaps > DSP_HG_MAX_TAPS) { set_error("hifigan_conv_f32: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!l.x || !l.w || !l.w2 || !l.out) { set_error("hifigan_conv_f32: null pointer (w = hi part, w2 = lo part of the split weights)"); return DSP_EINVAL; }
    if ((l.Cout & 3) || (l.out_mode == DSP_HG_OUT_UPSAMPLE ? (l.M != l.up_u * l.Cout) : (l.M != l.Cout || l.Tout != l.T))) {
        set_error("hifigan_conv_f32: inconsistent M=%d Cout=%d mode=%d", l.M, l.Cout, l.out_mode); return DSP_EINVAL; }
    HgsParams p;
    p.x = (const float*)l.x; p.wh = (const _Float16*)l.w; p.wl = (const _Float16*)l.w2; p.bias = l.bias; p.res = (const float*)l.res; p.out = (float*)l.out;
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
        int rc = hgs_conv_one(l, B, as_stream(stream), lens, mul);
        if (rc) return rc;
    }
    return DSP_OK;
}

extern "C" int dsp_hifigan_pack_weights_f32(const float* w, void* w_hi, void* w_lo, int ntaps, int M, int CI, dsp_stream_t stream)
{
    const long n = dsp_hifigan_packed_weight_elems(ntaps, M, CI);
    if (n < 0 || !w || !w_hi || !w_lo) { set_error("hifigan_pack_weights_f32: bad arguments"); return DSP_EINVAL; }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hgs_pack_weights_kernel, dim3(grid), dim3(256), 0, as_stream(stream), w, (_Float16*)w_hi, (_Float16*)w_lo, ntaps, M, CI);
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
    int gx = (T + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(hgs_post_kernel, dim3(gx, B), dim3(256), (size_t)K * C * 4, as_stream(stream), x, w, bias, wav, T, C, K, slope, lens, len_mul);
    return check_launch("hifigan_post_f32");
}
