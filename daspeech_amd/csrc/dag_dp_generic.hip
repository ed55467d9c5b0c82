// dag_dp_generic.hip — generic (any TR, any L <= 16k) row-sequential DP kernels: K2 alpha, K3 beta, K6 max-alpha
// + trace, K7 back-trace.  One 1024-thread workgroup per sample keeps the previous DP row in LDS and walks the
// T_b rows with one workgroup barrier per row: no inter-workgroup hand-off, so none of the reference's
// spin-wait protocol (dag_loss.cu:50-62,86-88,133-137) and none of its latent seg-2 race.
//
// This is the correctness baseline of the library (log-space, two-pass max/sum exactly like
// dag_loss.cu:94-127); the banded / dense fast paths in dag_dp_banded.hip are checked against it.
//
// Replaces: calculate_alpha_kernel (dag_loss.cu:40-140), calculate_beta_kernel (:178-274),
//           calculate_maxalpha_kernel (dag_best_alignment.cu:39-130), calculate_backtrace_kernel (:170-206).
#include "common.h"

namespace dsp {

constexpr int DP_THREADS = 1024;

// mode: 0 = alpha (K2), 1 = beta (K3).  grid = (B, ndir); blockIdx.y selects the direction when both run.
__global__ __launch_bounds__(DP_THREADS) void dag_logsum_generic_kernel(
    const float* __restrict__ match, const float* __restrict__ links,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ alpha, float* __restrict__ beta, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rowA = smem;
    float* rowB = smem + L;
    const int b = blockIdx.x;
    const bool do_beta = (alpha == nullptr) ? true : (blockIdx.y == 1);
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* M = match + (size_t)b * T * L;
    const float* K = links + (size_t)b * L * TR;
    float* O = (do_beta ? beta : alpha) + (size_t)b * T * L;
    const int tid = threadIdx.x;

    // rows outside [0, T_b) are never reached: -inf (the reference pre-fills with at::zeros().fill_(-inf), dag_loss.cu:162)
    for (int t = max(Tb, 0); t < T; ++t)
        for (int j = tid; j < L; j += DP_THREADS) O[(size_t)t * L + j] = NEG_INF;
    if (Tb <= 0 || Lb <= 0 || Tb > T || Lb > L) {      // invalid sample: everything -inf, no trap
        for (int t = 0; t < min(max(Tb, 0), T); ++t)
            for (int j = tid; j < L; j += DP_THREADS) O[(size_t)t * L + j] = NEG_INF;
        return;
    }

    float* prev = rowA;
    float* cur = rowB;
    if (!do_beta) {
        // t = 0: alpha[0,0] = match[0,0]   (dag_loss.cu:75-77)
        for (int j = tid; j < L; j += DP_THREADS) {
            float v = (j == 0) ? M[0] : NEG_INF;
            prev[j] = v; O[j] = v;
        }
        __syncthreads();
        for (int t = 1; t < Tb; ++t) {
            const float* Mt = M + (size_t)t * L;
            float* Ot = O + (size_t)t * L;
            for (int j = tid; j < L; j += DP_THREADS) {
                float res = NEG_INF;
                if (j >= t && j < Lb) {                                  // dag_loss.cu:84 (pos + t < output_len)
                    const int maxd = min(j, TR);                         // :96
                    float mx = NEG_INF;
                    for (int d = 1; d <= maxd; ++d)
                        mx = fmaxf(mx, prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)]);
                    if (mx != NEG_INF) {                                 // :113-115
                        float s = 0.f;
                        for (int d = 1; d <= maxd; ++d)
                            s += __expf(prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)] - mx);
                        res = __logf(s) + mx + Mt[j];                    // :126
                    }
                }
                cur[j] = res; Ot[j] = res;
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        }
    } else {
        // t = T_b-1: beta[T_b-1, L_b-1] = match[...]   (dag_loss.cu:208-211)
        {
            const int t = Tb - 1;
            for (int j = tid; j < L; j += DP_THREADS) {
                float v = (j == Lb - 1) ? M[(size_t)t * L + j] : NEG_INF;
                prev[j] = v; O[(size_t)t * L + j] = v;
            }
        }
        __syncthreads();
        for (int t = Tb - 2; t >= 0; --t) {
            const float* Mt = M + (size_t)t * L;
            float* Ot = O + (size_t)t * L;
            for (int j = tid; j < L; j += DP_THREADS) {
                float res = NEG_INF;
                if (j >= t && j < Lb) {                                  // dag_loss.cu:229-230
                    const int maxd = min(Lb - 1 - j, TR);                // :232
                    const float* Kj = K + (size_t)j * TR;
                    float mx = NEG_INF;
                    for (int d = 1; d <= maxd; ++d) mx = fmaxf(mx, prev[j + d] + Kj[d - 1]);
                    if (mx != NEG_INF) {
                        float s = 0.f;
                        for (int d = 1; d <= maxd; ++d) s += __expf(prev[j + d] + Kj[d - 1] - mx);
                        res = __logf(s) + mx + Mt[j];
                    }
                }
                cur[j] = res; Ot[j] = res;
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        }
    }
}

// loss[b] = beta[b,0,0] (with beta) or alpha[b,T_b-1,L_b-1]      (dag_loss.py:107-110)
__global__ void dag_pick_loss_kernel(const float* alpha, const float* beta, const int64_t* out_len,
                                     const int64_t* tgt_len, float* loss, int B, int T, int L)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (beta) { loss[b] = beta[(size_t)b * T * L]; return; }
    int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    loss[b] = (Tb >= 1 && Tb <= T && Lb >= 1 && Lb <= L) ? alpha[(size_t)b * T * L + (size_t)(Tb - 1) * L + (Lb - 1)] : NEG_INF;
}

// K6: max-DP + trace.  Tie rule = smallest predecessor index (scan predecessors ascending, strict >).
__global__ __launch_bounds__(DP_THREADS) void dag_maxalpha_generic_kernel(
    const float* __restrict__ match, const float* __restrict__ links,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ alpha, int32_t* __restrict__ trace, int B, int T, int L, int TR)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* prev = smem;
    float* cur = smem + L;
    const int b = blockIdx.x;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* M = match + (size_t)b * T * L;
    const float* K = links + (size_t)b * L * TR;
    float* O = alpha + (size_t)b * T * L;
    int32_t* Tr = trace + (size_t)b * T * L;
    const int tid = threadIdx.x;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    const int Tv = valid ? Tb : 0;
    for (int t = Tv; t < T; ++t)
        for (int j = tid; j < L; j += DP_THREADS) { O[(size_t)t * L + j] = NEG_INF; Tr[(size_t)t * L + j] = -1; }
    if (!valid) return;
    for (int j = tid; j < L; j += DP_THREADS) {
        float v = (j == 0) ? M[0] : NEG_INF;
        prev[j] = v; O[j] = v; Tr[j] = -1;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        const float* Mt = M + (size_t)t * L;
        for (int j = tid; j < L; j += DP_THREADS) {
            float res = NEG_INF; int arg = -1;
            if (j >= t && j < Lb) {
                const int maxd = min(j, TR);
                float mx = NEG_INF;
                for (int d = maxd; d >= 1; --d) {
                    float v = prev[j - d] + K[(size_t)(j - d) * TR + (d - 1)];
                    if (v > mx) { mx = v; arg = j - d; }
                }
                res = mx + Mt[j];
            }
            cur[j] = res; O[(size_t)t * L + j] = res; Tr[(size_t)t * L + j] = arg;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
    }
}

// K7: back-trace.  One workgroup per sample: lane 0 chases the T_b pointers into an LDS image of the path row,
// then the whole workgroup stores it as int64 in one coalesced sweep (the reference stored int32 and cast in
// Python, dag_loss.py:228).
__global__ __launch_bounds__(256) void dag_backtrace_kernel(
    const int32_t* __restrict__ trace, const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    int64_t* __restrict__ path, int B, int T, int L)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lp[];
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < L; j += blockDim.x) lp[j] = -1;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
        if (!(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L)) {
            const int32_t* Tr = trace + (size_t)b * T * L;
            int pos = Lb - 1;
            for (int t = Tb - 1; t >= 0 && pos >= 0; --t) {     // dag_best_alignment.cu:192-201
                lp[pos] = t;
                pos = Tr[(size_t)t * L + pos];
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += blockDim.x) path[(size_t)b * L + j] = lp[j];
}

int launch_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L, hipStream_t st)
{
    const size_t lds2 = (size_t)L * sizeof(int32_t);
    if (lds2 > 160 * 1024) { set_error("dag_best_alignment: graph size L=%d too large for the back-trace row image", L); return DSP_EINVAL; }
    if (lds2 > 48 * 1024) (void)hipFuncSetAttribute((const void*)dag_backtrace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(dag_backtrace_kernel, dim3(B), dim3(256), lds2, st, trace, out_len, tgt_len, path, B, T, L);
    return check_launch("dag_best_alignment(back-trace)");
}

int launch_dag_fwd_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                           float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const size_t lds = 2 * (size_t)L * sizeof(float);
    if (lds > 160 * 1024) { set_error("dag_loss: graph size L=%d exceeds the generic kernel's LDS rows (max 20480)", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dag_logsum_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int ndir = (alpha && beta) ? 2 : 1;
    hipLaunchKernelGGL(dag_logsum_generic_kernel, dim3(B, ndir), dim3(DP_THREADS), lds, st, match, links, out_len, tgt_len,
                       alpha, beta, B, T, L, TR);
    return check_launch("dag_loss_fwd(generic)");
}

int launch_pick_loss(const float* alpha, const float* beta, const int64_t* out_len, const int64_t* tgt_len, float* loss,
                     int B, int T, int L, hipStream_t st)
{
    hipLaunchKernelGGL(dag_pick_loss_kernel, dim3((B + 63) / 64), dim3(64), 0, st, alpha, beta, out_len, tgt_len, loss, B, T, L);
    return check_launch("dag_pick_loss");
}

int launch_best_alignment_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                  float* alpha, int32_t* trace, int64_t* path, int B, int T, int L, int TR, hipStream_t st)
{
    const size_t lds = 2 * (size_t)L * sizeof(float);
    if (lds > 160 * 1024) { set_error("dag_best_alignment: graph size L=%d too large (max 20480)", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)dag_maxalpha_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dag_maxalpha_generic_kernel, dim3(B), dim3(DP_THREADS), lds, st, match, links, out_len, tgt_len,
                       alpha, trace, B, T, L, TR);
    int rc = check_launch("dag_best_alignment(max-alpha)");
    if (rc) return rc;
    return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, st);
}

}  // namespace dsp
