// extract_links.hip — the DA-Transformer's transition producer, fused, compact layout.
//
// Replaces the inference path of DAGDecoder.extract_links (DASpeech/models/s2t_conformer_dag.py:171-212):
//     content[b,i,j,h] = q[b,i,h,:] . k[b,j,h,:] / sqrt(ck)                      (:183-186: an [B,L,L,H] einsum)
//     banded:   keep j = i+1 .. i+TR (:191-196), mask j >= out_len[b] (:197), log_softmax over the kept j per head (:199),
//               rows without any successor -> -inf (:200-201)
//     links[b,i,d] = logsumexp_h(content_ls[b,i,d,h] + log_gates[b,i,h])         (:208-210)
// The reference materialises the L x L x H content tensor and gathers the band out of it; here only the band is ever
// computed: one workgroup per (sample, 4 source vertices), thread = (head h, successor j): 8 heads x 32 successors per step, soft-max
// over the slots inside the 32 lanes of a head, the log-sum-exp over heads through LDS; the per-head scores of the four vertices are
// parked in LDS (any TR, incl. README's --max-transition-length 99999: TR = L-1).
#include "common.h"

namespace dsp {

constexpr int XL_H = 8;                       // attention heads of the link predictor (fixed by the architecture: 8)
constexpr int XL_IT = 4;                      // source vertices per workgroup

// ck = head width (64 in the released model, any multiple of 4 up to 128)
// A workgroup owns XL_IT consecutive source vertices and walks the successors j ONCE for all of them: thread (h, lane) holds the
// k row of successor j in registers and takes the XL_IT dot products against the queries in LDS (broadcast reads), so a k row is
// read from L2 once per workgroup instead of once per source vertex (with TR = L-1 the one-vertex-at-a-time form moved
// L^2/2 x 2 KB per sample through L2: 0.92 ms at B=32, L=400).  Same FMA order per dot product as before.
template <int CK4>                            // head width / 4
__global__ __launch_bounds__(256) void extract_links_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ log_gates,
    const int64_t* __restrict__ out_len, const float* __restrict__ dist_bias, float* __restrict__ links,
    int B, int L, int TR, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float xl_smem[];
    constexpr int CK = CK4 * 4;
    const int TRp = ((TR + 31) / 32) * 32;
    float* qs = xl_smem;                       // [IT][H][CK]   the source vertices' queries
    float* sc = qs + XL_IT * XL_H * CK;        // [IT][TRp][H]  scores
    float* red = sc + (size_t)XL_IT * TRp * XL_H;   // [IT][H][2]  per-head max / log-sum
    const int b = blockIdx.y;
    const int tid = threadIdx.x, d0 = tid & 31, h = tid >> 5;
    const int Lb = (int)out_len[b];
    const size_t rowstride = (size_t)XL_H * CK;
    const int i0 = blockIdx.x * XL_IT;
    const int nit = min(XL_IT, L - i0);
    for (int e = tid; e < nit * XL_H * CK; e += 256) qs[e] = q[((size_t)b * L + i0) * rowstride + e];
    for (int e = tid; e < XL_IT * TRp * XL_H; e += 256) sc[e] = NEG_INF;
    __syncthreads();
    // ---- scores: successors j = i0+1 .. i0+nit-1+TR in chunks of 32 (lane <-> j), XL_IT dot products per k row
    float mx[XL_IT];
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) mx[ii] = NEG_INF;
    const int jend = min(min(L, Lb), i0 + nit + TR);           // successors beyond the graph never score (:197)
    for (int jc = i0 + 1; jc < jend; jc += 32) {
        const int j = jc + d0;
        const bool live = j < jend;
        float4 kv[CK4];
        const float4* kr = reinterpret_cast<const float4*>(k + ((size_t)b * L + (live ? j : jc)) * rowstride + (size_t)h * CK);
#pragma unroll
        for (int c = 0; c < CK4; ++c) kv[c] = kr[c];            // unconditional (clamped row): the requests go out together
#pragma unroll
        for (int ii = 0; ii < XL_IT; ++ii) {
            const int d = j - (i0 + ii) - 1;
            const float4* qr = reinterpret_cast<const float4*>(qs + (ii * XL_H + h) * CK);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int c = 0; c < CK4; ++c) {
                const float4 qv = qr[c];
                a0 = fmaf(qv.x, kv[c].x, a0); a1 = fmaf(qv.y, kv[c].y, a1); a2 = fmaf(qv.z, kv[c].z, a2); a3 = fmaf(qv.w, kv[c].w, a3);
            }
            if (live && ii < nit && d >= 0 && d < TR) {
                float sv = ((a0 + a1) + (a2 + a3)) * scale;
                if (dist_bias) sv += dist_bias[d];
                sc[((size_t)ii * TRp + d) * XL_H + h] = sv;
                mx[ii] = fmaxf(mx[ii], sv);
            }
        }
    }
    // soft-max over the slots of head h: its 32 lanes are one half of a wave
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii)
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) mx[ii] = fmaxf(mx[ii], __shfl_xor(mx[ii], o, 32));
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < XL_IT; ++ii) {
        float sum = 0.f;
        if (mx[ii] != NEG_INF)
            for (int dc = 0; dc < TR; dc += 32) { const int d = dc + d0; if (d < TR) sum += __expf(sc[((size_t)ii * TRp + d) * XL_H + h] - mx[ii]); }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 32);
        if (d0 == 0) { red[(ii * XL_H + h) * 2] = mx[ii]; red[(ii * XL_H + h) * 2 + 1] = (mx[ii] == NEG_INF) ? 0.f : __logf(sum); }
    }
    __syncthreads();
    // ---- links[b,i,d] = logsumexp_h(score - max_h - logsum_h + log_gate_h); every thread takes slots tid, tid+256, ...
    for (int ii = 0; ii < nit; ++ii) {
        const int i = i0 + ii;
        float gate[XL_H], mh[XL_H], lh[XL_H];
#pragma unroll
        for (int hh = 0; hh < XL_H; ++hh) {
            gate[hh] = log_gates[((size_t)b * L + i) * XL_H + hh]; mh[hh] = red[(ii * XL_H + hh) * 2]; lh[hh] = red[(ii * XL_H + hh) * 2 + 1];
        }
        for (int d = tid; d < TR; d += 256) {
            float v[XL_H], m2 = NEG_INF;
#pragma unroll
            for (int hh = 0; hh < XL_H; ++hh) {
                const float s = sc[((size_t)ii * TRp + d) * XL_H + hh];
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
    if (H != XL_H) { set_error("extract_links: needs %d heads (got H=%d)", XL_H, H); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !log_gates || !out_len || !links) { set_error("extract_links: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)q) | ((uintptr_t)k)) & 15) { set_error("extract_links: q / k must be 16-byte aligned"); return DSP_EINVAL; }
    const size_t lds = ((size_t)XL_IT * XL_H * CK + (size_t)XL_IT * ((TR + 31) / 32) * 32 * XL_H + 2 * XL_IT * XL_H) * sizeof(float);
    if (lds > 150 * 1024) { set_error("extract_links: TR=%d too large for the score image", TR); return DSP_EINVAL; }
    if (!(CK == 32 || CK == 64 || CK == 128)) { set_error("extract_links: head width %d (32, 64 or 128)", CK); return DSP_EINVAL; }
    auto kern = CK == 64 ? extract_links_kernel<16> : (CK == 32 ? extract_links_kernel<8> : extract_links_kernel<32>);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((L + XL_IT - 1) / XL_IT, B), dim3(256), lds, as_stream(stream),
                       q, k, log_gates, out_len, dist_bias, links, B, L, TR, scale);
    return check_launch("extract_links");
}
