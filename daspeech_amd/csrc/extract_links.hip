// extract_links.hip — the DA-Transformer's transition producer, fused, compact layout.
//
// Replaces the inference path of DAGDecoder.extract_links (DASpeech/models/s2t_conformer_dag.py:171-212):
//     content[b,i,j,h] = q[b,i,h,:] . k[b,j,h,:] / sqrt(ck)                      (:183-186: an [B,L,L,H] einsum)
//     banded:   keep j = i+1 .. i+TR (:191-196), mask j >= out_len[b] (:197), log_softmax over the kept j per head (:199),
//               rows without any successor -> -inf (:200-201)
//     links[b,i,d] = logsumexp_h(content_ls[b,i,d,h] + log_gates[b,i,h])         (:208-210)
// The reference materialises the L x L x H content tensor and gathers the band out of it; here only the band is ever
// computed: one workgroup per (sample, 4 source vertices), thread = (head h, successor slot d): 8 heads x 32 slots, the
// 64-term dot products straight from L2-resident k rows (a k row serves <= TR neighbouring source vertices), soft-max over d
// inside the 32 lanes of a head, the log-sum-exp over heads through LDS.  TR > 32 (README's --max-transition-length 99999: TR =
// L-1) runs the same code in chunks of 32 slots with the per-head scores parked in LDS.
#include "common.h"

namespace dsp {

constexpr int XL_H = 8;                       // attention heads of the link predictor (fixed by the architecture: 8)
constexpr int XL_IT = 4;                      // source vertices per workgroup

// ck = head width (64 in the released model, any multiple of 4 up to 128)
__global__ __launch_bounds__(256) void extract_links_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, float* __restrict__ links,
    int B, int L, int CK, int TR, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    float* qs = xl_smem;                       // [H][CK]   the source vertex's queries
    float* sc = qs + XL_H * CK;                // [TRp][H]  scores of the current source vertex (TRp = TR rounded up to 32)
    float* red = sc + ((TR + 31) / 32) * 32 * XL_H;   // [H][2]  per-head max / log-sum
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = (int)out_len[b];
    const size_t rowstride = (size_t)XL_H * CK;
    for (int ii = 0; ii < XL_IT; ++ii) {
        const int i = blockIdx.x * XL_IT + ii;
        if (i >= L) break;
        __syncthreads();
        for (int e = tid; e < XL_H * CK; e += 256) qs[e] = q[((size_t)b * L + i) * rowstride + e];
        __syncthreads();
        // ---- scores of head h for successor slots d0, d0+32, ...; running max for the soft-max
        float mx = NEG_INF;
        for (int dc = 0; dc < TR; dc += 32) {
            const int d = dc + d0, j = i + d + 1;
            float s = NEG_INF;
            if (d < TR && j < L && j < Lb) {                                             // (:197) successors beyond the graph
                const float4* kr = reinterpret_cast<const float4*>(k + ((size_t)b * L + j) * rowstride + (size_t)h * CK);
                const float4* qr = reinterpret_cast<const float4*>(qs + h * CK);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int c = 0; c < CK / 4; ++c) {
                    const float4 kv = kr[c], qv = qr[c];
                    a0 = fmaf(qv.x, kv.x, a0); a1 = fmaf(qv.y, kv.y, a1); a2 = fmaf(qv.z, kv.z, a2); a3 = fmaf(qv.w, kv.w, a3);
                }
                s = ((a0 + a1) + (a2 + a3)) * scale;
                if (dist_bias) s += dist_bias[d];
            }
            if (d < ((TR + 31) / 32) * 32) sc[d * XL_H + h] = s;
            mx = fmaxf(mx, s);
        }
        // soft-max over the slots of head h: its 32 lanes are one half of a wave
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
        __syncthreads();
        float sum = 0.f;
        if (mx != NEG_INF)
            for (int dc = 0; dc < TR; dc += 32) { const int d = dc + d0; if (d < TR) sum += __expf(sc[d * XL_H + h] - mx); }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 32);
        if (d0 == 0) { red[2 * h] = mx; red[2 * h + 1] = (mx == NEG_INF) ? 0.f : __logf(sum); }
        __syncthreads();
        // ---- links[b,i,d] = logsumexp_h(score - max_h - logsum_h + log_gate_h); every thread takes slots tid, tid+256, ...
        float gate[XL_H], mh[XL_H], lh[XL_H];
#pragma unroll
        for (int hh = 0; hh < XL_H; ++hh) {
            gate[hh] = log_gates[((size_t)b * L + i) * XL_H + hh]; mh[hh] = red[2 * hh]; lh[hh] = red[2 * hh + 1];
        }
        for (int d = tid; d < TR; d += 256) {
            float v[XL_H], m2 = NEG_INF;
#pragma unroll
            for (int hh = 0; hh < XL_H; ++hh) {
                const float s = sc[d * XL_H + hh];
                v[hh] = (s == NEG_INF) ? NEG_INF : ((s - mh[hh]) - lh[hh]) + gate[hh];
                m2 = fmaxf(m2, v[hh]);
            }
            float r = NEG_INF;
            if (m2 != NEG_INF) {
                float acc = 0.f;
#pragma unroll
                for (int hh = 0; hh < XL_H; ++hh) acc += __expf(v[hh] - m2);
                r = m2 + __logf(acc);
            }
            links[((size_t)b * L + i) * TR + d] = r;
        }
    }
}

}  // namespace dsp

extern "C" int dsp_extract_links(const float* q, const float* k, const float* log_gates, const int64_t* out_len,
                                 const float* dist_bias, float* links, int B, int L, int H, int CK, int TR, float scale,
                                 dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || L < 1 || TR < 1 || CK < 4) { set_error("extract_links: bad sizes B=%d L=%d TR=%d CK=%d", B, L, TR, CK); return DSP_EINVAL; }
    if (H != XL_H || (CK & 3) || CK > 128) { set_error("extract_links: needs %d heads and a head width that is a multiple of 4 up to 128 (got H=%d, CK=%d)", XL_H, H, CK); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !log_gates || !out_len || !links) { set_error("extract_links: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)q) | ((uintptr_t)k)) & 15) { set_error("extract_links: q / k must be 16-byte aligned"); return DSP_EINVAL; }
    const size_t lds = ((size_t)XL_H * CK + (size_t)((TR + 31) / 32) * 32 * XL_H + 2 * XL_H) * sizeof(float);
    if (lds > 150 * 1024) { set_error("extract_links: TR=%d too large for the score image", TR); return DSP_EINVAL; }
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)extract_links_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(extract_links_kernel, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), lds, as_stream(stream),
                       q, k, log_gates, out_len, dist_bias, links, B, L, CK, TR, scale);
    return check_launch("extract_links");
}
