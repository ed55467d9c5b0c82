// conformer_ops.hip — inference-side fusion inside the Conformer convolution module (caller of the hot path).
// HIPCC_FLAGS: -fno-slp-vectorize
// (the SLP vectoriser pairs the score FMAs of relpos_attention_kernel into v_pk_fma_f32 through a v_mov per operand and AGPR spills)
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

// ---- Conformer relative-position self-attention, fused (eval, fp32) -------------------------------------------------------------
// espnet-style RelPositionMultiHeadedAttention as used by fairseq's conformer_layer.py:
//     ac[i,j] = (q_i + u) . k_j          bd[i,j] = (q_i + v) . p_{(T-1) - i + j}        (the rel_shift of the [T, 2T-1] product)
//     out_i   = sum_j softmax_j((ac + bd) / sqrt(dk), keys masked by the padding mask)[j] * v_j
// torch runs this as two batched GEMMs, a pad/view/slice shift, add, scale, masked_fill, softmax and a third GEMM over [B,h,T,T] /
// [B,h,T,2T-1] tensors (≈10 launches and ≈0.5 GB of traffic per layer at B=64, T=200).  Here one workgroup owns 32 queries of one
// (sample, head): k and the T+31 position rows it can touch sit in LDS half a head width at a time, a thread accumulates 4 x 4
// score micro-tiles in registers (19 16-byte LDS reads per 128 FMAs), the scores go through a [32][T] LDS image for the soft-max,
// and the value product reads v from the same LDS space.  fp32 FMAs throughout.  dk = 64, T <= 256.
namespace dsp {

constexpr int RA_QT = 32, RA_DK = 64, RA_HS = 36, RA_QS = 68;      // query tile, head width, half-row pitch (32 + 4), q row pitch
constexpr int RA_U = 7;                                             // global requests in flight per lane while staging

__global__ __launch_bounds__(512) void relpos_attention_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ p,
    const float* __restrict__ bias_u, const float* __restrict__ bias_v, const unsigned char* __restrict__ pad_mask,
    float* __restrict__ out, int B, int T, int H, float scale, long ld)
{
    extern __shared__ __attribute__((aligned(16))) float ra_smem[];
    const int Tp = (T + 3) & ~3;
    float* Qu = ra_smem;                               // [QT][QS]  q + u
    float* Qv = Qu + RA_QT * RA_QS;                    // [QT][QS]  q + v
    float* Sc = Qv + RA_QT * RA_QS;                    // [QT][Tp + 4]
    float* Kh = Sc + RA_QT * (Tp + 4);                 // [Tp][HS]        (later: V [Tp][QS])
    float* Ph = Kh + (size_t)Tp * RA_HS;               // [Tp + QT][HS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * RA_QT, h = blockIdx.y, b = blockIdx.z;
    const int P = 2 * T - 1, C = H * RA_DK, SP = Tp + 4;
    const size_t rowstride = (size_t)ld, prow = (size_t)C;             // q / k / v rows (slices of a fused projection: ld = 3C), p / out rows
    const float* Qb = q + (size_t)b * T * rowstride + (size_t)h * RA_DK;
    const float* Kb = k + (size_t)b * T * rowstride + (size_t)h * RA_DK;
    const float* Vb = v + (size_t)b * T * rowstride + (size_t)h * RA_DK;
    const float* Pb = p + (size_t)h * RA_DK;           // [P][C]
    // ---- queries + biases
    for (int e = tid; e < RA_QT * (RA_DK / 4); e += 512) {
        const int i = e >> 4, c = (e & 15) * 4;
        float4 qv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + i < T) qv4 = *reinterpret_cast<const float4*>(Qb + (size_t)(i0 + i) * rowstride + c);
        const float4 u4 = *reinterpret_cast<const float4*>(bias_u + h * RA_DK + c), v4 = *reinterpret_cast<const float4*>(bias_v + h * RA_DK + c);
        *reinterpret_cast<float4*>(Qu + i * RA_QS + c) = make_float4(qv4.x + u4.x, qv4.y + u4.y, qv4.z + u4.z, qv4.w + u4.w);
        *reinterpret_cast<float4*>(Qv + i * RA_QS + c) = make_float4(qv4.x + v4.x, qv4.y + v4.y, qv4.z + v4.z, qv4.w + v4.w);
    }
    // ---- scores: 4 x 4 micro-tiles (ig, jg); a thread owns up to two of them across both half-width passes
    const int njg = Tp >> 2, nmt = (RA_QT / 4) * njg;
    float acc[1][4][4];                                // one micro-tile per thread: 8 * Tp/4 <= 512 of them (T <= 256)
#pragma unroll
    for (int m = 0; m < 1; ++m)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) acc[m][a][bb] = 0.f;
    const int r0 = (T - 1) - (i0 + RA_QT - 1);         // position row held in Ph[0]
    for (int half = 0; half < 2; ++half) {
        const int c0 = half * 32;
        __syncthreads();                               // previous pass done with Kh / Ph (and Qu / Qv written)
        // rows are requested RA_U at a time, unconditionally from a clamped row and zeroed by a select afterwards (a load under a lane
        // predicate waits for the one before it: ~43 dependent round trips per workgroup made this kernel 369 us per layer)
        for (int e0 = tid; e0 < Tp * 8; e0 += 512 * RA_U) {
            float4 x[RA_U];
#pragma unroll
            for (int u = 0; u < RA_U; ++u) {
                const int e = e0 + u * 512, j = e >> 3, c = (e & 7) * 4;
                x[u] = *reinterpret_cast<const float4*>(Kb + (size_t)(j < T ? j : 0) * rowstride + c0 + c);
            }
#pragma unroll
            for (int u = 0; u < RA_U; ++u) {
                const int e = e0 + u * 512, j = e >> 3, c = (e & 7) * 4;
                if (e < Tp * 8) *reinterpret_cast<float4*>(Kh + j * RA_HS + c) = (j < T) ? x[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int e0 = tid; e0 < (Tp + RA_QT) * 8; e0 += 512 * RA_U) {
            float4 x[RA_U];
#pragma unroll
            for (int u = 0; u < RA_U; ++u) {
                const int e = e0 + u * 512, rr = e >> 3, c = (e & 7) * 4, r = r0 + rr;
                x[u] = *reinterpret_cast<const float4*>(Pb + (size_t)((r >= 0 && r < P) ? r : 0) * prow + c0 + c);
            }
#pragma unroll
            for (int u = 0; u < RA_U; ++u) {
                const int e = e0 + u * 512, rr = e >> 3, c = (e & 7) * 4, r = r0 + rr;
                if (e < (Tp + RA_QT) * 8) *reinterpret_cast<float4*>(Ph + rr * RA_HS + c) = (r >= 0 && r < P) ? x[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 1; ++m) {
            const int mt = tid;
            if (mt < nmt) {
                // STRIDED micro-tile: queries ig + 8a, keys jg + njg*b.  Consecutive lanes (consecutive jg) then read consecutive LDS
                // rows — with the 36-word pitch 16 lanes cover all 64 banks; the contiguous 4 x 4 tile had them 16-way conflicted
                // (rows 4*jg apart: 429 us per layer instead of ~80)
                const int ig = mt / njg, jg = mt - ig * njg;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    float4 qu[4], qw[4], kk[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        qu[a] = *reinterpret_cast<const float4*>(Qu + (ig + 8 * a) * RA_QS + c0 + c);
                        qw[a] = *reinterpret_cast<const float4*>(Qv + (ig + 8 * a) * RA_QS + c0 + c);
                        kk[a] = *reinterpret_cast<const float4*>(Kh + (jg + njg * a) * RA_HS + c);
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int bb = 0; bb < 4; ++bb) {
                            const float4 pv = *reinterpret_cast<const float4*>(Ph + (31 - (ig + 8 * a) + jg + njg * bb) * RA_HS + c);
                            float s = acc[m][a][bb];
                            s = fmaf(qu[a].x, kk[bb].x, s); s = fmaf(qu[a].y, kk[bb].y, s); s = fmaf(qu[a].z, kk[bb].z, s); s = fmaf(qu[a].w, kk[bb].w, s);
                            s = fmaf(qw[a].x, pv.x, s); s = fmaf(qw[a].y, pv.y, s); s = fmaf(qw[a].z, pv.z, s); s = fmaf(qw[a].w, pv.w, s);
                            acc[m][a][bb] = s;
                        }
                }
            }
        }
    }
    // ---- scores -> LDS (scaled, masked keys -> -inf)
#pragma unroll
    for (int m = 0; m < 1; ++m) {
        const int mt = tid;
        if (mt < nmt) {
            const int ig = mt / njg, jg = mt - ig * njg;
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const int j = jg + njg * bb;
                const bool dead = j >= T || (pad_mask && pad_mask[(size_t)b * T + j]);
#pragma unroll
                for (int a = 0; a < 4; ++a) Sc[(ig + 8 * a) * SP + j] = dead ? NEG_INF : acc[m][a][bb] * scale;
            }
        }
    }
    __syncthreads();
    // ---- values into the K/P space, soft-max of 8 rows per wave meanwhile
    float* Vs = Kh;                                     // [Tp][QS]
    for (int e0 = tid; e0 < Tp * 16; e0 += 512 * RA_U) {
        float4 x[RA_U];
#pragma unroll
        for (int u = 0; u < RA_U; ++u) {
            const int e = e0 + u * 512, j = e >> 4, c = (e & 15) * 4;
            x[u] = *reinterpret_cast<const float4*>(Vb + (size_t)(j < T ? j : 0) * rowstride + c);
        }
#pragma unroll
        for (int u = 0; u < RA_U; ++u) {
            const int e = e0 + u * 512, j = e >> 4, c = (e & 15) * 4;
            if (e < Tp * 16) *reinterpret_cast<float4*>(Vs + j * RA_QS + c) = (j < T) ? x[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int rr = 0; rr < RA_QT / 8; ++rr) {
        float* row = Sc + (wave * (RA_QT / 8) + rr) * SP;
        float mx = NEG_INF;
        for (int j = lane; j < Tp; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < Tp; j += 64) { const float e = (mx == NEG_INF) ? 0.f : __expf(row[j] - mx); row[j] = e; sum += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.f / sum;                    // every key masked: 0 / 0 = NaN, as torch's soft-max of an all -inf row
        for (int j = lane; j < Tp; j += 64) row[j] *= inv;
    }
    __syncthreads();
    // ---- out[i][c .. c+7] = sum_j P[i][j] * v[j][c .. c+7]
    {
        const int i = tid >> 4, c = (tid & 15) * 4;
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* row = Sc + i * SP;
        for (int j = 0; j < Tp; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(row + j);
            const float4 a0 = *reinterpret_cast<const float4*>(Vs + (j + 0) * RA_QS + c), a1 = *reinterpret_cast<const float4*>(Vs + (j + 1) * RA_QS + c);
            const float4 a2 = *reinterpret_cast<const float4*>(Vs + (j + 2) * RA_QS + c), a3 = *reinterpret_cast<const float4*>(Vs + (j + 3) * RA_QS + c);
            o0.x = fmaf(w4.x, a0.x, o0.x); o0.y = fmaf(w4.x, a0.y, o0.y); o0.z = fmaf(w4.x, a0.z, o0.z); o0.w = fmaf(w4.x, a0.w, o0.w);
            o0.x = fmaf(w4.y, a1.x, o0.x); o0.y = fmaf(w4.y, a1.y, o0.y); o0.z = fmaf(w4.y, a1.z, o0.z); o0.w = fmaf(w4.y, a1.w, o0.w);
            o0.x = fmaf(w4.z, a2.x, o0.x); o0.y = fmaf(w4.z, a2.y, o0.y); o0.z = fmaf(w4.z, a2.z, o0.z); o0.w = fmaf(w4.z, a2.w, o0.w);
            o0.x = fmaf(w4.w, a3.x, o0.x); o0.y = fmaf(w4.w, a3.y, o0.y); o0.z = fmaf(w4.w, a3.z, o0.z); o0.w = fmaf(w4.w, a3.w, o0.w);
        }
        if (i0 + i < T) {
            float* O = out + ((size_t)b * T + i0 + i) * prow + (size_t)h * RA_DK + c;
            *reinterpret_cast<float4*>(O) = o0;
        }
    }
}

}  // namespace dsp

extern "C" int dsp_relpos_attention(const float* q, const float* k, const float* v, long ld, const float* p, const float* bias_u, const float* bias_v,
                                    const unsigned char* pad_mask, float* out, int B, int T, int H, int DK, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || T < 1 || H < 1 || DK != RA_DK || T > 256) { set_error("relpos_attention: needs head width 64 and T <= 256 (got T=%d, dk=%d)", T, DK); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !v || !p || !bias_u || !bias_v || !out) { set_error("relpos_attention: null pointer"); return DSP_EINVAL; }
    if (ld < (long)H * DK || (ld & 3)) { set_error("relpos_attention: row stride %ld must be >= H * dk and a multiple of 4", ld); return DSP_EINVAL; }
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)p) | ((uintptr_t)out) | ((uintptr_t)bias_u) | ((uintptr_t)bias_v)) & 15) {
        set_error("relpos_attention: pointers must be 16-byte aligned"); return DSP_EINVAL; }
    const int Tp = (T + 3) & ~3;
    const size_t kp = (size_t)Tp * RA_HS + (size_t)(Tp + RA_QT) * RA_HS, vs = (size_t)Tp * RA_QS;
    const size_t lds = (2 * (size_t)RA_QT * RA_QS + (size_t)RA_QT * (Tp + 4) + (kp > vs ? kp : vs)) * sizeof(float);
    if (lds > 160 * 1024) { set_error("relpos_attention: T=%d needs %zu bytes of LDS", T, lds); return DSP_EINVAL; }
    (void)hipFuncSetAttribute((const void*)relpos_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(relpos_attention_kernel, dim3((T + RA_QT - 1) / RA_QT, H, B), dim3(512), lds, as_stream(stream),
                       q, k, v, p, bias_u, bias_v, pad_mask, out, B, T, H, 1.f / sqrtf((float)DK), ld);
    return check_launch("relpos_attention");
}
