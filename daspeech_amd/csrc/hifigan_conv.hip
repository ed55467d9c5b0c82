// hifigan_conv.hip — the HiFi-GAN generator convolution stack on MFMA (gfx950).
//
// Replaces the torch/MIOpen conv calls of hifi-gan/models.py:35-43 (ResBlock1.forward) and :100-119 (Generator.forward).
// One kernel serves every layer (see include/daspeech_hifigan.h for the "taps" formulation):
//     out = scale * ( bias + [res] + sum_k  W_k[M x CI] . lrelu(X)[CI x t + shift_k] )
// MI355X mapping
//   * activations channels-last fp16, so an MFMA B fragment (8 consecutive input channels of one time step) is ONE 16-byte
//     LDS read and a D fragment stores 4 consecutive output channels;
//   * the workgroup stages the lrelu'd input tile [NT + halo][CI] in LDS once (XOR-swizzled 16-byte chunks: the 16 lanes of
//     a read group hit different rows at the same channel offset) and re-uses it for all taps and all output channels —
//     the leaky_relu is applied once per element, not once per tap;
//   * weights [tap][co][ci] stream from L2 straight into A fragments (each element is used by exactly one wave, for all of
//     that wave's time tiles);
//   * mfma_f32_16x16x32_f16, fp32 accumulate; 8 waves as WM (channel) x WN (time) sub-tiles.
#include "common.h"
#include <stdlib.h>
#include "../../include/daspeech_hifigan.h"

namespace dsp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct HgParams {
    const _Float16* x; const _Float16* w; const float* bias; const _Float16* res; _Float16* out;
    int B, T, M, ntaps, Tout, Cout, out_mode, up_u, up_pad;
    float pre_slope, scale;
    int shifts[DSP_HG_MAX_TAPS];
    int min_shift, max_shift;
    int dbg;                      // ablation bits (HG_ABLATE env, timing experiments only): 1 no staging loads, 2 no MFMA loop, 4 no epilogue
};

template <int CI>
__device__ __forceinline__ int hg_swz(int row, int chunk) {
    constexpr int CH = CI / 8;                          // 16-byte chunks per row
    constexpr int RPB = (CI * 2 >= 256) ? 1 : 256 / (CI * 2);      // rows per 256-byte LDS bank row
    constexpr int MASK = (CH < 16 ? CH : 16) - 1;
    if constexpr ((CH & (CH - 1)) != 0) return chunk;   // CI = 96: 12 chunks, not a power of two -> no swizzle
    else return chunk ^ ((row / RPB) & MASK);
}

template <int CI, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(512) void hifigan_conv_kernel(HgParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = CI / 8;
    constexpr int MI = MT / WM / 16, NI = NT / WN / 16;
    static_assert(WM * WN == 8 && MI >= 1 && NI >= 1, "8 waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * NT;                      // first output column (input time index q) of the tile
    const int m0 = blockIdx.y * MT;
    const int R = NT + (p.max_shift - p.min_shift);
    const _Float16* X = p.x + (size_t)b * p.T * CI;

    // ---- stage lrelu(x) tile: rows t0+min_shift .. t0+NT-1+max_shift, zero outside [0,T) ----
    const _Float16 slope = (_Float16)p.pre_slope;
    for (int e = tid; e < R * CH; e += 512) {
        const int row = e / CH, ch = e - row * CH;
        const int tg = t0 + p.min_shift + row;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tg >= 0 && tg < p.T && !(p.dbg & 1)) {
            v = *reinterpret_cast<const h8*>(X + (size_t)tg * CI + ch * 8);
            if (p.pre_slope != 1.0f) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] > (_Float16)0 ? v[i] : v[i] * slope;
            }
        }
        *reinterpret_cast<h8*>(smem + ((size_t)row * CH + hg_swz<CI>(row, ch)) * 16) = v;
    }
    __syncthreads();

    f4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int co_base = m0 + wm * (MI * 16);
    const int tl_base = wn * (NI * 16);
    const int lr = lane & 15, lk = lane >> 4;
    // flattened (tap, 32-channel chunk) loop; 3-deep register ring of weight fragments (static indices: unrolled by 3) so the
    // L2 latency of the weight stream is covered by two steps of MFMAs
    constexpr int NC = CI / 32;
    const int nsteps = (p.dbg & 2) ? 0 : p.ntaps * NC;
    auto load_a = [&](int step, h8 (&a)[MI]) {
        const int k = step / NC, c = step - k * NC;
        const _Float16* Wk = p.w + (size_t)k * p.M * CI;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int co = co_base + i * 16 + lr;
            a[i] = (co < p.M) ? *reinterpret_cast<const h8*>(Wk + (size_t)co * CI + c * 32 + lk * 8) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    h8 a0[MI], a1[MI], a2[MI];
    if (nsteps > 0) load_a(0, a0);
    if (nsteps > 1) load_a(1, a1);
    auto do_step = [&](int step, const h8 (&a)[MI]) {
        const int k = step / NC, c = step - k * NC;
        const int rshift = p.shifts[k] - p.min_shift;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = tl_base + j * 16 + lr + rshift;
            const h8 bf = *reinterpret_cast<const h8*>(smem + ((size_t)row * CH + hg_swz<CI>(row, c * 4 + lk)) * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bf, acc[i][j], 0, 0, 0);
        }
    };
    for (int step = 0; step < nsteps; step += 3) {
        if (step + 2 < nsteps) load_a(step + 2, a2);
        do_step(step, a0);
        if (step + 1 < nsteps) {
            if (step + 3 < nsteps) load_a(step + 3, a0);
            do_step(step + 1, a1);
        }
        if (step + 2 < nsteps) {
            if (step + 4 < nsteps) load_a(step + 4, a1);
            do_step(step + 2, a2);
        }
    }
    if (p.dbg & 4) return;

    // ---- epilogue through LDS: the D fragments (4 channels x 1 time step per lane) are transposed into a [NT][MT] fp16 tile
    //      so that global memory sees 16-byte, row-contiguous accesses for the residual / accumulate loads and the stores
    //      (fragment-shaped 8-byte stores at a row stride cost as much as the whole MFMA loop: 6.1 of 15.8 ms, ablation r01) ----
    constexpr int OPITCH = MT + 8;                    // halfs; +16 bytes per row de-conflicts the transposing ds_write_b64
    __syncthreads();                                  // all waves are done reading the input tile
    _Float16* otile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wm * (MI * 16) + i * 16 + lk * 4;          // row inside the M tile
        const int mrow = m0 + ml;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && mrow < p.M) {
            const int co = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? (mrow % p.Cout) : mrow;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = p.bias[co + e];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int tl = tl_base + j * 16 + lr;
            _Float16 hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(acc[i][j][e] + bv[e]);
            *reinterpret_cast<uint2*>(otile + (size_t)tl * OPITCH + ml) = *reinterpret_cast<uint2*>(hv);
        }
    }
    __syncthreads();
    constexpr int CPR = MT / 8;                       // 16-byte chunks per tile row
    for (int e = tid; e < NT * CPR; e += 512) {
        const int tl = e / CPR, ch = e - tl * CPR;
        const int mrow = m0 + ch * 8;
        if (mrow >= p.M) continue;
        const int q = t0 + tl;
        int tout = q, co = mrow;
        if (p.out_mode == DSP_HG_OUT_UPSAMPLE) { const int r = mrow / p.Cout; co = mrow - r * p.Cout; tout = q * p.up_u + r - p.up_pad; }
        if (tout < 0 || tout >= p.Tout) continue;
        const size_t o = ((size_t)b * p.Tout + tout) * p.Cout + co;
        const h8 v = *reinterpret_cast<const h8*>(otile + (size_t)tl * OPITCH + ch * 8);
        h8 r8 = {0, 0, 0, 0, 0, 0, 0, 0}, a8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.res) r8 = *reinterpret_cast<const h8*>(p.res + o);
        if (p.out_mode == DSP_HG_OUT_ACCUM) a8 = *reinterpret_cast<const h8*>(p.out + o);
        h8 w8;
#pragma unroll
        for (int x = 0; x < 8; ++x) w8[x] = (_Float16)(p.scale * ((float)v[x] + (float)r8[x]) + (float)a8[x]);
        *reinterpret_cast<h8*>(p.out + o) = w8;
    }
}

template <int CI, int MT, int NT, int WM, int WN>
static int hg_launch(const HgParams& p, hipStream_t st)
{
    const int R = NT + (p.max_shift - p.min_shift);
    size_t lds = (size_t)R * CI * 2;
    const size_t lds_out = (size_t)NT * (MT + 8) * 2;
    if (lds_out > lds) lds = lds_out;
    if (lds > 160 * 1024) { set_error("hifigan_conv: input tile %zu bytes exceeds LDS", lds); return DSP_EINVAL; }
    auto k = hifigan_conv_kernel<CI, MT, NT, WM, WN>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // columns: conv -> T outputs; upsample -> q = 0..T (the last column feeds the tail of the transposed conv)
    const int ncol = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? p.T + 1 : p.T;
    dim3 grid((ncol + NT - 1) / NT, (p.M + MT - 1) / MT, p.B);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
    return check_launch("hifigan_conv");
}

__global__ void hg_pack_kernel(const float* __restrict__ x, _Float16* __restrict__ out, long n_rows, int C, int Cpad)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * Cpad; e += (long)gridDim.x * blockDim.x) {
        const long r = e / Cpad; const int c = (int)(e - r * Cpad);
        out[e] = (c < C) ? (_Float16)x[r * C + c] : (_Float16)0;
    }
}

__global__ __launch_bounds__(256) void hg_post_kernel(const _Float16* __restrict__ x, const float* __restrict__ w, float bias,
                                                      float* __restrict__ wav, int T, int C, int K, float slope)
{
    extern __shared__ float ws[];              // [K][C]
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const _Float16* X = x + (size_t)b * T * C;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        float acc = bias;
        for (int k = 0; k < K; ++k) {
            const int tt = t + k - (K - 1) / 2;
            if (tt < 0 || tt >= T) continue;
            const _Float16* xr = X + (size_t)tt * C;
            for (int c = 0; c < C; c += 8) {
                const h8 v = *reinterpret_cast<const h8*>(xr + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) { float f = (float)v[i]; f = f > 0.f ? f : f * slope; acc += f * ws[k * C + c + i]; }
            }
        }
        wav[(size_t)b * T + t] = tanhf(acc);
    }
}

}  // namespace dsp

using namespace dsp;

static int hg_conv_one(const void* x, const void* w, const float* bias, const void* res, void* out,
                       int B, int T, int CI, int M, int ntaps, const int* host_shifts, float pre_slope, float scale,
                       int out_mode, int up_u, int up_pad, int Tout, int Cout, hipStream_t st)
{
    if (B < 0 || T < 1 || M < 1 || ntaps < 1 || ntaps > DSP_HG_MAX_TAPS || !host_shifts) { set_error("hifigan_conv: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w || !out) { set_error("hifigan_conv: null pointer"); return DSP_EINVAL; }
    if ((Cout & 3) || (out_mode == DSP_HG_OUT_UPSAMPLE ? (M != up_u * Cout) : (M != Cout || Tout != T))) {
        set_error("hifigan_conv: inconsistent M=%d Cout=%d mode=%d", M, Cout, out_mode); return DSP_EINVAL; }
    HgParams p;
    p.x = (const _Float16*)x; p.w = (const _Float16*)w; p.bias = bias; p.res = (const _Float16*)res; p.out = (_Float16*)out;
    p.B = B; p.T = T; p.M = M; p.ntaps = ntaps; p.Tout = Tout; p.Cout = Cout; p.out_mode = out_mode; p.up_u = up_u; p.up_pad = up_pad;
    p.pre_slope = pre_slope; p.scale = scale;
    { static int ablate = -1; if (ablate < 0) { const char* ab = getenv("HG_ABLATE"); ablate = ab ? atoi(ab) : 0; } p.dbg = ablate; }
    p.min_shift = p.max_shift = host_shifts[0];
    for (int k = 0; k < ntaps; ++k) { p.shifts[k] = host_shifts[k]; p.min_shift = min(p.min_shift, host_shifts[k]); p.max_shift = max(p.max_shift, host_shifts[k]); }
    switch (CI) {
        case 512: return hg_launch<512, 256, 128, 8, 1>(p, st);
        case 256: return hg_launch<256, 256, 128, 8, 1>(p, st);
        case 128: return hg_launch<128, 128, 256, 4, 2>(p, st);
        case 96:  return hg_launch<96, 256, 128, 8, 1>(p, st);
        case 64:  return hg_launch<64, 64, 512, 2, 4>(p, st);      // 512-column tiles: 12.1 vs 13.0 ms per B=32 pass (sweep r01e;
                                                                    // 64x64 wave tiles, and larger tiles for C = 32 / 128 / 256, were slower or equal)
        case 32:  return hg_launch<32, 32, 512, 1, 8>(p, st);
    }
    set_error("hifigan_conv: unsupported input channel count %d", CI);
    return DSP_EINVAL;
}

extern "C" int dsp_hifigan_conv(const void* x, const void* w, const float* bias, const void* res, void* out,
                                int B, int T, int CI, int M, int ntaps, const int* host_shifts, float pre_slope, float scale,
                                int out_mode, int up_u, int up_pad, int Tout, int Cout, dsp_stream_t stream)
{
    return hg_conv_one(x, w, bias, res, out, B, T, CI, M, ntaps, host_shifts, pre_slope, scale, out_mode, up_u, up_pad, Tout, Cout,
                       as_stream(stream));
}

// The generator is ~100 of these layers per call; driven one ctypes call at a time the host, not the GPU, sets the pace at
// vocoder batch sizes.  One call walks a whole layer table.
extern "C" int dsp_hifigan_conv_chain(const dsp_hg_layer* layers, int n_layers, int B, dsp_stream_t stream)
{
    if (n_layers < 0 || (n_layers > 0 && !layers)) { set_error("hifigan_conv_chain: bad layer table"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    for (int i = 0; i < n_layers; ++i) {
        const dsp_hg_layer& l = layers[i];
        int rc = hg_conv_one(l.x, l.w, l.bias, l.res, l.out, B, l.T, l.CI, l.M, l.ntaps, l.shifts, l.pre_slope, l.scale,
                             l.out_mode, l.up_u, l.up_pad, l.Tout, l.Cout, st);
        if (rc) return rc;
    }
    return DSP_OK;
}

extern "C" int dsp_hifigan_pack_input(const float* x, void* out, int B, int T, int C, int Cpad, dsp_stream_t stream)
{
    if (B < 0 || T < 0 || C < 1 || Cpad < C) { set_error("hifigan_pack_input: bad sizes"); return DSP_EINVAL; }
    const long n = (long)B * T;
    if (n == 0) return DSP_OK;
    int grid = (int)((n * Cpad + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hg_pack_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, (_Float16*)out, n, C, Cpad);
    return check_launch("hifigan_pack_input");
}

extern "C" int dsp_hifigan_post(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                                dsp_stream_t stream)
{
    if (B < 0 || T < 1 || C < 8 || (C & 7) || K < 1) { set_error("hifigan_post: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    int gx = (T + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(hg_post_kernel, dim3(gx, B), dim3(256), (size_t)K * C * 4, as_stream(stream), (const _Float16*)x, w, bias, wav, T, C, K, slope);
    return check_launch("hifigan_post");
}
