// dag_dp_f64.hip — the DAG operators in DOUBLE precision (r06).
//
// The reference instantiates every kernel for float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF: dag_loss.cu:160,294,415,499,
// dag_best_alignment.cu:143,219).  The fast paths of this library compute in fp32 (exp-space strips, fp32 / fp16 matrix cores); float64
// tensors take THESE kernels instead: the same recurrences in log space, every intermediate a double, so a caller that checks the loss or
// its gradients in double precision gets double precision (r05 routed such inputs through a T-step torch loop that kept T tensors of
// [B,L,TR] doubles alive for autograd and ran out of memory on dense windows).  A correctness path: one 1024-thread workgroup per
// (sample, direction) walks the rows with the previous row in LDS — no attempt at the fp32 kernels' speed.
//
// Replaces, for scalar_t = double: calculate_alpha_kernel (dag_loss.cu:40-140), calculate_beta_kernel (:178-274),
// calculate_grad_match_all_kernel (:378-401), calculate_grad_links_kernel (:432-485), calculate_maxalpha_kernel
// (dag_best_alignment.cu:39-130) + calculate_backtrace_kernel (:170-206).  Semantics as the fp32 operators: cells outside
// {t < T_b, t <= j < L_b} are -inf, an empty predecessor set stays -inf, invalid / unreachable samples give -inf and zero gradients,
// Viterbi ties take the smallest predecessor index.
#include "common.h"

namespace dsp {

int launch_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L, hipStream_t st);

constexpr int D64_THREADS = 1024;
#define D64_NEG (-__builtin_huge_val())

// grid (B, ndir): blockIdx.y = 1 (or alpha == NULL) computes beta
__global__ __launch_bounds__(D64_THREADS) void dag64_logsum_kernel(
    const double* __restrict__ match, const double* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    double* __restrict__ alpha, double* __restrict__ beta, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) char d64_smem[];
    double* prev = reinterpret_cast<double*>(d64_smem);
    double* cur = prev + L;
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool do_beta = (alpha == nullptr) ? true : (blockIdx.y == 1);
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const double* M = match + (size_t)b * T * L;
    const double* K = links + (size_t)b * L * TR;
    double* O = (do_beta ? beta : alpha) + (size_t)b * T * L;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    for (int t = valid ? Tb : 0; t < T; ++t)
        for (int j = tid; j < L; j += D64_THREADS) O[(size_t)t * L + j] = D64_NEG;
    if (!valid) return;
    if (!do_beta) {
        for (int j = tid; j < L; j += D64_THREADS) { const double v = (j == 0) ? M[0] : D64_NEG; prev[j] = v; O[j] = v; }      // dag_loss.cu:75-77
        __syncthreads();
        for (int t = 1; t < Tb; ++t) {
            const double* Mt = M + (size_t)t * L;
            for (int j = tid; j < L; j += D64_THREADS) {
                double res = D64_NEG;
                if (j >= t && j < Lb) {                                                      // :84
                    const int maxd = min(j, TR);                                             // :96
                    double mx = D64_NEG;
                    for (int d = 1; d <= maxd; ++d) mx = fmax(mx, prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)]);
                    if (mx != D64_NEG) {                                                     // :113-115
                        double s = 0.0;
                        for (int d = 1; d <= maxd; ++d) s += exp(prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)] - mx);
                        res = log(s) + mx + Mt[j];                                           // :126
                    }
                }
                cur[j] = res; O[(size_t)t * L + j] = res;
            }
            __syncthreads();
            double* tmp = prev; prev = cur; cur = tmp;
        }
    } else {
        {
            const int t = Tb - 1;                                                            // :208-211
            for (int j = tid; j < L; j += D64_THREADS) { const double v = (j == Lb - 1) ? M[(size_t)t * L + j] : D64_NEG; prev[j] = v; O[(size_t)t * L + j] = v; }
        }
        __syncthreads();
        for (int t = Tb - 2; t >= 0; --t) {
            const double* Mt = M + (size_t)t * L;
            for (int j = tid; j < L; j += D64_THREADS) {
                double res = D64_NEG;
                if (j >= t && j < Lb) {                                                      // :229-230
                    const int maxd = min(Lb - 1 - j, TR);                                    // :232
                    const double* Kj = K + (size_t)j * TR;
                    double mx = D64_NEG;
                    for (int d = 1; d <= maxd; ++d) mx = fmax(mx, prev[j + d] + Kj[d - 1]);
                    if (mx != D64_NEG) {
                        double s = 0.0;
                        for (int d = 1; d <= maxd; ++d) s += exp(prev[j + d] + Kj[d - 1] - mx);
                        res = log(s) + mx + Mt[j];
                    }
                }
                cur[j] = res; O[(size_t)t * L + j] = res;
            }
            __syncthreads();
            double* tmp = prev; prev = cur; cur = tmp;
        }
    }
}

__global__ void dag64_pick_loss_kernel(const double* alpha, const double* beta, const int64_t* out_len, const int64_t* tgt_len, double* loss, int B, int T, int L)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (beta) { loss[b] = beta[(size_t)b * T * L]; return; }                                 // dag_loss.py:107-110
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    loss[b] = (Tb >= 1 && Tb <= T && Lb >= 1 && Lb <= L) ? alpha[(size_t)b * T * L + (size_t)(Tb - 1) * L + (Lb - 1)] : D64_NEG;
}

// K4 (dag_loss.cu:394-398)
__global__ __launch_bounds__(256) void dag64_grad_match_kernel(const double* __restrict__ g_out, const double* __restrict__ alpha, const double* __restrict__ beta,
                                                              const double* __restrict__ match, double* __restrict__ g_match, int B, size_t TL)
{
    const int b = blockIdx.y;
    const double b00 = beta[(size_t)b * TL], go = g_out[b];
    const bool dead = isinf(b00);
    const double* A = alpha + (size_t)b * TL; const double* Bt = beta + (size_t)b * TL; const double* M = match + (size_t)b * TL;
    double* G = g_match + (size_t)b * TL;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TL; i += (size_t)gridDim.x * blockDim.x)
        G[i] = (dead || isinf(M[i])) ? 0.0 : exp(A[i] + Bt[i] - M[i] - b00) * go;
}

// K5 (dag_loss.cu:461-475): one thread per (vertex i, slot d) sums over t; 32 d-lanes share alpha[t,i] and read beta[t+1] contiguously
__global__ __launch_bounds__(256) void dag64_grad_links_kernel(const double* __restrict__ g_out, const double* __restrict__ alpha, const double* __restrict__ beta,
                                                              const double* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
                                                              double* __restrict__ g_links, int B, int T, int L, int TR)
{
    const int b = blockIdx.z;
    const int d = blockIdx.x * 32 + (threadIdx.x & 31);
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= L || d >= TR) return;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const double b00 = beta[(size_t)b * TL];
    double* out = g_links + ((size_t)b * L + i) * TR + d;
    const int nx = i + d + 1;
    if (i >= Lb || nx >= Lb || isinf(b00) || Tb > T || Lb > L || Tb < 1) { *out = 0.0; return; }
    const double* A = alpha + (size_t)b * TL + i;
    const double* Bt = beta + (size_t)b * TL + L + nx;
    const double extra = links[((size_t)b * L + i) * TR + d] - b00;                          // :469
    double acc = 0.0;
    for (int t = 0; t + 1 < Tb; ++t) acc += exp(A[(size_t)t * L] + Bt[(size_t)t * L] + extra);
    *out = acc * g_out[b];
}

// K6 (dag_best_alignment.cu:39-130): max-DP + arg-max trace; predecessors scanned in ascending index, strict > : smallest index on ties
__global__ __launch_bounds__(D64_THREADS) void dag64_maxalpha_kernel(
    const double* __restrict__ match, const double* __restrict__ links, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    double* __restrict__ alpha, int32_t* __restrict__ trace, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) char d64_smem[];
    double* prev = reinterpret_cast<double*>(d64_smem);
    double* cur = prev + L;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const double* M = match + (size_t)b * T * L;
    const double* K = links + (size_t)b * L * TR;
    double* O = alpha + (size_t)b * T * L;
    int32_t* Tr = trace + (size_t)b * T * L;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    for (int t = valid ? Tb : 0; t < T; ++t)
        for (int j = tid; j < L; j += D64_THREADS) { O[(size_t)t * L + j] = D64_NEG; Tr[(size_t)t * L + j] = -1; }
    if (!valid) return;
    for (int j = tid; j < L; j += D64_THREADS) { const double v = (j == 0) ? M[0] : D64_NEG; prev[j] = v; O[j] = v; Tr[j] = -1; }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        const double* Mt = M + (size_t)t * L;
        for (int j = tid; j < L; j += D64_THREADS) {
            double res = D64_NEG; int arg = -1;
            if (j >= t && j < Lb) {
                const int maxd = min(j, TR);
                double mx = D64_NEG;
                for (int d = maxd; d >= 1; --d) {
                    const double v = prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)];
                    if (v > mx) { mx = v; arg = j - d; }
                }
                res = mx + Mt[j];
            }
            cur[j] = res; O[(size_t)t * L + j] = res; Tr[(size_t)t * L + j] = arg;
        }
        __syncthreads();
        double* tmp = prev; prev = cur; cur = tmp;
    }
}

static int d64_check(const char* fn, int B, int T, int L, int TR, size_t* lds)
{
    if (B < 0 || T < 1 || L < 1 || TR < 1) { set_error("%s: bad sizes B=%d T=%d L=%d TR=%d", fn, B, T, L, TR); return DSP_EINVAL; }
    *lds = (size_t)2 * L * sizeof(double);
    if (*lds > 160 * 1024) { set_error("%s: graph of %d vertices too large for the double-precision path (max 10240)", fn, L); return DSP_EINVAL; }
    return DSP_OK;
}

}  // namespace dsp

using namespace dsp;

extern "C" int dsp_dag_loss_fwd_f64(const double* match, const double* links, const int64_t* out_len, const int64_t* tgt_len,
                                    double* alpha, double* beta, double* loss, int B, int T, int L, int TR, dsp_stream_t stream)
{
    size_t lds;
    int rc = d64_check("dag_loss_fwd_f64", B, T, L, TR, &lds);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || (!alpha && !beta)) { set_error("dag_loss_fwd_f64: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag64_logsum_kernel, (int)lds);
    hipLaunchKernelGGL(dag64_logsum_kernel, dim3(B, (alpha && beta) ? 2 : 1), dim3(D64_THREADS), lds, st, match, links, out_len, tgt_len, alpha, beta, B, T, L, TR);
    if ((rc = check_launch("dag_loss_fwd_f64"))) return rc;
    if (loss) {
        hipLaunchKernelGGL(dag64_pick_loss_kernel, dim3((B + 63) / 64), dim3(64), 0, st, alpha, beta, out_len, tgt_len, loss, B, T, L);
        rc = check_launch("dag_loss_fwd_f64(pick)");
    }
    return rc;
}

extern "C" int dsp_dag_loss_bwd_f64(const double* grad_out, const double* alpha, const double* beta, const double* match, const double* links,
                                    const int64_t* out_len, const int64_t* tgt_len, double* grad_match, double* grad_links,
                                    int B, int T, int L, int TR, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || L < 1 || TR < 1) { set_error("dag_loss_bwd_f64: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!grad_out || !alpha || !beta || !match || !links || !out_len || !tgt_len) { set_error("dag_loss_bwd_f64: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (grad_match) {
        const size_t TL = (size_t)T * L;
        int gx = (int)((TL + 255) / 256); if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(dag64_grad_match_kernel, dim3(gx, B), dim3(256), 0, st, grad_out, alpha, beta, match, grad_match, B, TL);
        if (int rc = check_launch("dag_loss_bwd_f64(grad_match)")) return rc;
    }
    if (grad_links) {
        hipLaunchKernelGGL(dag64_grad_links_kernel, dim3((TR + 31) / 32, (L + 7) / 8, B), dim3(256), 0, st,
                           grad_out, alpha, beta, links, out_len, tgt_len, grad_links, B, T, L, TR);
        if (int rc = check_launch("dag_loss_bwd_f64(grad_links)")) return rc;
    }
    return DSP_OK;
}

extern "C" int dsp_dag_best_alignment_f64(const double* match, const double* links, const int64_t* out_len, const int64_t* tgt_len,
                                          double* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream)
{
    size_t lds;
    int rc = d64_check("dag_best_alignment_f64", B, T, L, TR, &lds);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || !alpha_max || !trace || !path) { set_error("dag_best_alignment_f64: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag64_maxalpha_kernel, (int)lds);
    hipLaunchKernelGGL(dag64_maxalpha_kernel, dim3(B), dim3(D64_THREADS), lds, st, match, links, out_len, tgt_len, alpha_max, trace, B, T, L, TR);
    if ((rc = check_launch("dag_best_alignment_f64"))) return rc;
    return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, st);
}
