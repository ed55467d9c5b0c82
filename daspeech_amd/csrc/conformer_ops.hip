// conformer_ops.hip — inference-side fusion inside the Conformer convolution module (caller of the hot path).
//
// fairseq's ConvolutionModule (conformer_layer.py: pointwise_conv1 -> GLU -> depthwise_conv -> batch_norm -> SiLU -> pointwise_conv2)
// runs the depthwise Conv1d(C, C, K, groups = C) on a [B,C,T] transpose; MIOpen serves it with its naive direct kernel
// (naive_conv_ab_nonpacked_fwd_nchw_float: 65 us per layer at B=64) between two transposes, a batch-norm and a SiLU launch.
// Here, in eval mode, the whole   y = SiLU(BN_eval(depthwise(x)))   is one pass over the channels-last [B,T,C] tensor the layer
// already has: a thread owns 4 channels (one 16-byte lane) and DW_TT consecutive frames, the K + DW_TT - 1 input rows it needs
// slide through registers, weights [C][K] are read through L1.  HBM bound: B*T*C*4 bytes in, the same out.
#include "common.h"
#include "../../include/daspeech_decode.h"

namespace dsp {

constexpr int DW_TT = 8;

template <int K>
__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_w, const float* __restrict__ bn_b,
    const float* __restrict__ bn_mean, const float* __restrict__ bn_var, float eps, float* __restrict__ y, int B, int T, int C)
{
    const int c4n = C >> 2;
    const int nt = (T + DW_TT - 1) / DW_TT;
    const long total = (long)B * nt * c4n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n); const long r = e / c4n;
        const int tb = (int)(r % nt), b = (int)(r / nt);
        const int c = c4 * 4, t0 = tb * DW_TT;
        const float* X = x + (size_t)b * T * C + c;
        float4 acc[DW_TT];
#pragma unroll
        for (int u = 0; u < DW_TT; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        float wr[4][K];                                  // this lane's 4 x K taps, in registers (read per FMA they were ~1000 L1 loads per lane)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) wr[i][k] = w[(c + i) * K + k];
        constexpr int P = (K - 1) / 2;
        // input row t0 - P + s contributes to output t0 + u through tap k = s - u
#pragma unroll
        for (int s = 0; s < K + DW_TT - 1; ++s) {
            const int ti = t0 - P + s;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ti >= 0 && ti < T) v = *reinterpret_cast<const float4*>(X + (size_t)ti * C);
#pragma unroll
            for (int u = 0; u < DW_TT; ++u) {
                const int k = s - u;
                if (k >= 0 && k < K) {
                    acc[u].x = fmaf(v.x, wr[0][k], acc[u].x); acc[u].y = fmaf(v.y, wr[1][k], acc[u].y);
                    acc[u].z = fmaf(v.z, wr[2][k], acc[u].z); acc[u].w = fmaf(v.w, wr[3][k], acc[u].w);
                }
            }
        }
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = bn_w ? bn_w[c + i] : 1.f, be = bn_b ? bn_b[c + i] : 0.f;
            sc[i] = g * rsqrtf(bn_var[c + i] + eps);
            sh[i] = be - bn_mean[c + i] * sc[i];
        }
#pragma unroll
        for (int u = 0; u < DW_TT; ++u) {
            const int t = t0 + u;
            if (t < T) {
                float o[4] = {acc[u].x * sc[0] + sh[0], acc[u].y * sc[1] + sh[1], acc[u].z * sc[2] + sh[2], acc[u].w * sc[3] + sh[3]};
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = o[i] / (1.f + __expf(-o[i]));
                *reinterpret_cast<float4*>(y + ((size_t)b * T + t) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace dsp

extern "C" int dsp_dwconv_bn_silu(const float* x, const float* w, const float* bn_w, const float* bn_b, const float* bn_mean,
                                  const float* bn_var, float eps, float* y, int B, int T, int C, int K, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || T < 1 || C < 4 || (C & 3)) { set_error("dwconv_bn_silu: bad sizes B=%d T=%d C=%d", B, T, C); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w || !bn_mean || !bn_var || !y || x == y) { set_error("dwconv_bn_silu: null or aliased pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) { set_error("dwconv_bn_silu: x / y must be 16-byte aligned"); return DSP_EINVAL; }
    const long total = (long)B * ((T + DW_TT - 1) / DW_TT) * (C / 4);
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    hipStream_t st = as_stream(stream);
    switch (K) {
        case 31: hipLaunchKernelGGL(dwconv_bn_silu_kernel<31>, dim3(grid), dim3(256), 0, st, x, w, bn_w, bn_b, bn_mean, bn_var, eps, y, B, T, C); break;
        case 15: hipLaunchKernelGGL(dwconv_bn_silu_kernel<15>, dim3(grid), dim3(256), 0, st, x, w, bn_w, bn_b, bn_mean, bn_var, eps, y, B, T, C); break;
        case 7:  hipLaunchKernelGGL(dwconv_bn_silu_kernel<7>, dim3(grid), dim3(256), 0, st, x, w, bn_w, bn_b, bn_mean, bn_var, eps, y, B, T, C); break;
        case 3:  hipLaunchKernelGGL(dwconv_bn_silu_kernel<3>, dim3(grid), dim3(256), 0, st, x, w, bn_w, bn_b, bn_mean, bn_var, eps, y, B, T, C); break;
        default: set_error("dwconv_bn_silu: kernel size %d (3, 7, 15, 31)", K); return DSP_EINVAL;
    }
    return check_launch("dwconv_bn_silu");
}

// ---- LayerNorm over the last dimension, one wave per row -----------------------------------------------------------------------
// torch's kernel takes 19.5 us for 12800 rows x 256 (13 MB in, 13 MB out: 5 us of HBM time) and the inference pipelines call it ~100
// times per batch.  Here a wave holds its row in registers (C/64 values per lane, 16-byte loads), mean and the variance of the
// centred values are two wave reductions, four rows per workgroup.  Same definition as torch (biased variance, eps inside the sqrt).
namespace dsp {

template <int NV>      // float4 per lane: C <= 256 * NV
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         float eps, float* __restrict__ y, long rows, int C)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* X = x + row * C; float* Y = y + row * C;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        v[k] = (c < C) ? *reinterpret_cast<const float4*>(X + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < C) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < C) {
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w) g = *reinterpret_cast<const float4*>(w + c);
            if (b) be = *reinterpret_cast<const float4*>(b + c);
            *reinterpret_cast<float4*>(Y + c) = make_float4((v[k].x - mean) * rstd * g.x + be.x, (v[k].y - mean) * rstd * g.y + be.y,
                                                            (v[k].z - mean) * rstd * g.z + be.z, (v[k].w - mean) * rstd * g.w + be.w);
        }
    }
}

}  // namespace dsp

extern "C" int dsp_layer_norm(const float* x, const float* w, const float* b, float eps, float* y, long rows, int C, dsp_stream_t stream)
{
    using namespace dsp;
    if (rows < 0 || C < 4 || (C & 3) || C > 2048) { set_error("layer_norm: bad sizes rows=%ld C=%d (C %% 4 == 0, <= 2048)", rows, C); return DSP_EINVAL; }
    if (rows == 0) return DSP_OK;
    if (!x || !y) { set_error("layer_norm: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w) | ((uintptr_t)b)) & 15) { set_error("layer_norm: pointers must be 16-byte aligned"); return DSP_EINVAL; }
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t st = as_stream(stream);
    if (C <= 256) hipLaunchKernelGGL(layer_norm_kernel<1>, dim3(grid), dim3(256), 0, st, x, w, b, eps, y, rows, C);
    else if (C <= 512) hipLaunchKernelGGL(layer_norm_kernel<2>, dim3(grid), dim3(256), 0, st, x, w, b, eps, y, rows, C);
    else if (C <= 1024) hipLaunchKernelGGL(layer_norm_kernel<4>, dim3(grid), dim3(256), 0, st, x, w, b, eps, y, rows, C);
    else hipLaunchKernelGGL(layer_norm_kernel<8>, dim3(grid), dim3(256), 0, st, x, w, b, eps, y, rows, C);
    return check_launch("layer_norm");
}

