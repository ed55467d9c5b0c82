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

// K5 tiled (TR-agnostic, used for every TR): one workgroup = 64 source vertices x 32 transition slots of one sample.
// The reference walks alpha[t][i] / beta[t+1][i+d+1] down the t axis with stride-L loads per thread
// (dag_loss.cu:471-475); here row tiles of alpha (64 wide) and beta (96 wide) are staged through LDS with coalesced
// loads (pre-scaled to the log2 domain), so HBM/L2 see each alpha row once and each beta row 1.5x per 64-vertex block,
// and the inner loop is LDS reads + v_exp_f32 only.
constexpr int K5_TC = 32;
__global__ __launch_bounds__(256) void dag_grad_links_tiled_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ g_links, int B, int T, int L, int TR)
{
    __shared__ float At[K5_TC][64];
    __shared__ float Bt[K5_TC][96];
    constexpr float LOG2E = 1.4426950408889634f;
    const int b = blockIdx.z;
    const int i0 = blockIdx.x * 64, dc0 = blockIdx.y * 32;
    const int tid = threadIdx.x, i = tid & 63, dg = tid >> 6;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float* A = alpha + (size_t)b * TL;
    const float* Bp = beta + (size_t)b * TL;
    const float b00 = Bp[0];
    const bool dead = isinf(b00) || Tb > T || Lb > L || Tb < 1 || Lb < 1;
    const int vi = i0 + i;
    float extra[8], acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int d = dc0 + dg * 8 + u;
        const bool ok = !dead && d < TR && vi < Lb && vi + d + 1 < Lb;                 // dag_loss.cu:461-466
        extra[u] = ok ? (links[((size_t)b * L + vi) * TR + d] - b00) * LOG2E : NEG_INF;   // :469
        acc[u] = 0.f;
    }
    const int nt = dead ? 0 : (Tb - 1);                    // t = 0 .. T_b-2   (:471-475)
    const int bcol0 = i0 + dc0 + 1;                        // first beta column of the tile
    for (int t0 = 0; t0 < nt; t0 += K5_TC) {
        const int rows = min(K5_TC, nt - t0);
        for (int e = tid; e < K5_TC * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            float v = NEG_INF;
            if (r < rows && i0 + c < L) v = A[(size_t)(t0 + r) * L + i0 + c] * LOG2E;
            At[r][c] = v;
        }
        for (int e = tid; e < K5_TC * 96; e += 256) {
            const int r = e / 96, c = e - r * 96;
            float v = NEG_INF;
            if (r < rows && bcol0 + c < L) v = Bp[(size_t)(t0 + r + 1) * L + bcol0 + c] * LOG2E;
            Bt[r][c] = v;
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const float a = At[r][i];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u] += __builtin_amdgcn_exp2f(a + Bt[r][i + dg * 8 + u] + extra[u]);
        }
        __syncthreads();
    }
    if (vi < L) {
        const float go = g_out[b];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int d = dc0 + dg * 8 + u;
            if (d < TR) g_links[((size_t)b * L + vi) * TR + d] = (extra[u] == NEG_INF) ? 0.f : acc[u] * go;
        }
    }
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
        hipLaunchKernelGGL(dag_grad_links_tiled_kernel, dim3((L + 63) / 64, (TR + 31) / 32, B), dim3(256), 0, st,
                           g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR);
        int rc = check_launch("dag_loss_bwd(grad_links)");
        if (rc) return rc;
    }
    return DSP_OK;
}

}  // namespace dsp
