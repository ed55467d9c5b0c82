// dag_grad.hip — K4 (grad wrt match_all) and K5 (grad wrt links) for gfx950, generic log-space form.
// Replaces calculate_grad_match_all_kernel (dag_loss.cu:378-401) and calculate_grad_links_kernel (:432-485).
#include "common.h"

namespace dsp {

// K4: pure elementwise, HBM-bound: 3 reads + 1 write per cell, float4 wide.
__global__ __launch_bounds__(256) void dag_grad_match_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ match, float* __restrict__ g_match, int B, size_t TL)
{
    const int b = blockIdx.y;
    const float b00 = beta[(size_t)b * TL];
    const float go = g_out[b];
    const bool dead = isinf(b00);
    const float* A = alpha + (size_t)b * TL; const float* Bt = beta + (size_t)b * TL;
    const float* M = match + (size_t)b * TL; float* G = g_match + (size_t)b * TL;
    const size_t n4 = TL / 4;
    const bool al = ((((uintptr_t)A) | ((uintptr_t)Bt) | ((uintptr_t)M) | ((uintptr_t)G)) & 15) == 0;
    if (al) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 a = reinterpret_cast<const float4*>(A)[i], be = reinterpret_cast<const float4*>(Bt)[i];
            float4 m = reinterpret_cast<const float4*>(M)[i], r;
            r.x = (dead || isinf(m.x)) ? 0.f : __expf(a.x + be.x - m.x - b00) * go;      // dag_loss.cu:394-398
            r.y = (dead || isinf(m.y)) ? 0.f : __expf(a.y + be.y - m.y - b00) * go;
            r.z = (dead || isinf(m.z)) ? 0.f : __expf(a.z + be.z - m.z - b00) * go;
            r.w = (dead || isinf(m.w)) ? 0.f : __expf(a.w + be.w - m.w - b00) * go;
            reinterpret_cast<float4*>(G)[i] = r;
        }
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TL; i += (size_t)gridDim.x * blockDim.x)
            G[i] = (dead || isinf(M[i])) ? 0.f : __expf(A[i] + Bt[i] - M[i] - b00) * go;
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TL; i += (size_t)gridDim.x * blockDim.x)
            G[i] = (dead || isinf(M[i])) ? 0.f : __expf(A[i] + Bt[i] - M[i] - b00) * go;
    }
}

// K5 generic: thread (dx = d, iy = i) sums over t.  alpha[t,i] is a broadcast within the 32 d-lanes,
// beta[t+1, i+d+1] is contiguous across them.
__global__ __launch_bounds__(256) void dag_grad_links_generic_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR)
{
    const int b = blockIdx.z;
    const int d = blockIdx.x * 32 + (threadIdx.x & 31);
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= L || d >= TR) return;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float b00 = beta[(size_t)b * TL];
    float* out = g_links + ((size_t)b * L + i) * TR + d;
    const int nx = i + d + 1;
    if (i >= Lb || nx >= Lb || isinf(b00) || Tb > T || Lb > L) { *out = 0.f; return; }     // dag_loss.cu:461-466 (+ zeros init :541)
    const float* A = alpha + (size_t)b * TL + i;
    const float* Bt = beta + (size_t)b * TL + L + nx;
    const float extra = links[((size_t)b * L + i) * TR + d] - b00;                          // :469
    float acc = 0.f;
    for (int t = 0; t + 1 < Tb; ++t)                                                        // :471-475
        acc += __expf(A[(size_t)t * L] + Bt[(size_t)t * L] + extra);
    *out = acc * g_out[b];
}

int launch_dag_bwd_generic(const float* g_out, const float* alpha, const float* beta, const float* match, const float* links,
                           const int64_t* out_len, const int64_t* tgt_len, float* g_match, float* g_links,
                           int B, int T, int L, int TR, hipStream_t st)
{
    if (g_match) {
        const size_t TL = (size_t)T * L;
        int gx = (int)((TL / 4 + 255) / 256); if (gx < 1) gx = 1; if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(dag_grad_match_kernel, dim3(gx, B), dim3(256), 0, st, g_out, alpha, beta, match, g_match, B, TL);
        int rc = check_launch("dag_loss_bwd(grad_match)");
        if (rc) return rc;
    }
    if (g_links) {
        hipLaunchKernelGGL(dag_grad_links_generic_kernel, dim3((TR + 31) / 32, (L + 7) / 8, B), dim3(256), 0, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR);
        int rc = check_launch("dag_loss_bwd(grad_links)");
        if (rc) return rc;
    }
    return DSP_OK;
}

}  // namespace dsp
