// dag_grad_dense.hip — K5 (gradient w.r.t. the links) for DENSE windows (TR > 64) as block products over the target axis on the f32
// matrix cores.
//
//     grad_links[b][i][j-i-1] = g[b] * exp(link[i][j-i-1]) * sum_{t=0}^{T_b-2} exp( alpha[t][i] + beta[t+1][j] - beta[0][0] )        (dag_loss.cu:461-475)
//
// The reference (and the tiled log-space kernel of dag_grad.hip this replaces for dense windows) spends one exp per (t, i, j):
// B * T * L^2 / 2 of them — C2 at TR = 4095: 1.4e11, 35.7 ms at the v_exp_f32 rate.  The sum over t is a matrix product
// [i x t] . [t x j]: with, per row t and column block J, sb = ceil(max_j beta2[t+1][J]) (log2 domain)
//     A[t][i] = 2^(alpha2[t][i] + sb - Z2)        B[t][j] = 2^(beta2[t+1][j] - sb)  in (0, 1]
// the term is A * B exactly, one exp per matrix ELEMENT and 64-vertex block pair instead of one per term, and the products run on
// v_mfma_f32_16x16x4_f32 (exact f32).  For a block pair I < J every i precedes the vertex j* that carries the block maximum of beta, so
// alpha2[t][i] + sb - Z2 <= -link2[i][j*] + 1: A is bounded by the weakest link into j* (no overflow), and whatever flushes in A or B is
// a term at least 100 binades under 1.  That argument needs i < j* and a link inside the window, so block pairs on the diagonal (I = J)
// or cut by the window edge (some j - i - 1 >= TR) are summed term by term in log space instead (1/NJ of the work).
#include "common.h"

namespace dsp {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr float GD_LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float gd_max16(float v) {          // maximum over the 16 lanes of a DPP row, result in every lane
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
    return v;
}

// Two passes over the same grid.  PASS 0: the block products (48 VGPRs, 8 waves per SIMD).  A pair whose scaled alpha hits the 2^100
// clamp cannot be done in exp space; it leaves a marker (a NaN with a payload no arithmetic produces) in its first entry.  PASS 1:
// the term-by-term log-space form for the diagonal / window-edge pairs and for marked pairs; every other workgroup returns at once.
// (One kernel holding both forms needed 105 VGPRs — half the occupancy, C2 at TR = 4095: 5.6 -> 7.1 ms.)
constexpr unsigned GD_REDO_MARK = 0x7FC0DA65u;
// grid: (NJ * (NJ + 1) / 2 block pairs I <= J, B); 256 threads
template <int PASS>
__global__ __launch_bounds__(256, PASS == 0 ? 8 : 4) void dag_grad_links_dense_kernel(
    const float* __restrict__ g_out, const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ links,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len, float* __restrict__ g_links, int B, int T, int L, int TR, int NJ)
{
    __shared__ __attribute__((aligned(16))) float As[16 * 64];          // [t][i]
    __shared__ __attribute__((aligned(16))) float Bs[16 * 64];          // [t][j]
    const int b = blockIdx.y;
    // pair index -> (I, J), I <= J, enumerated row by row: pairs of row I start at I * NJ - I (I - 1) / 2
    int I = 0;
    { int rem = blockIdx.x; while (rem >= NJ - I) { rem -= NJ - I; ++I; } }
    const int J = I + (int)blockIdx.x - (I * NJ - I * (I - 1) / 2);
    const int ib = I * 64, jb = J * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const size_t TL = (size_t)T * L;
    const float* A = alpha + (size_t)b * TL;
    const float* Bt = beta + (size_t)b * TL;
    const float* K = links + (size_t)b * L * TR;
    float* G = g_links + (size_t)b * L * TR;
    const float b00 = Bt[0];
    const bool dead = isinf(b00) || Tb > T || Lb > L || Tb < 1 || Lb < 1;
    const int dmin = jb - (ib + 63) - 1, dmax = (jb + 63) - ib - 1;      // distance range of the pair
    if (dmin >= TR) return;                                             // entirely outside the window: no such entries in the compact layout
    const float z2 = b00 * GD_LOG2E, go = g_out[b];
    const int nt = dead ? 0 : (Tb - 1);                                  // t = 0 .. T_b - 2
    const bool structural = (I == J) || dmax >= TR;
    float* mark = G + (size_t)ib * TR + (jb - ib - 1);                  // entry (ib, jb) of an off-diagonal pair inside the window
    if (PASS == 0 && structural) return;
    if (PASS == 1 && !structural && __float_as_uint(*mark) != GD_REDO_MARK) return;

    // element (i, j) of the pair -> grad entry; zero where the reference leaves its zero-initialised output (i >= L_b or j >= L_b) or Z = -inf.
    // The link meets the sum in the LOG domain: e^link * sum is 0 * (large) for a transition weaker than e^-87 although the entry — the
    // posterior of that transition — can be anything up to 1 (r02 fuzzing: NaN = 0 * inf and lost entries on batches with such links).
    auto store = [&](int i, int j, float sum) {
        const int d = j - i - 1;
        if (i >= L || j >= L || d < 0 || d >= TR) return;
        const bool ok = !dead && i < Lb && j < Lb;
        float v = 0.f;
        if (ok) { const float lk = K[(size_t)i * TR + d]; v = (lk == NEG_INF) ? 0.f : go * __builtin_amdgcn_exp2f(lk * GD_LOG2E + __builtin_amdgcn_logf(sum)); }
        G[(size_t)i * TR + d] = v;
    };

    const int r = tid >> 4, c4 = tid & 15;                               // staging: row r of the chunk, columns 4 c4 .. +3
    if (PASS == 0) {
        const int lr = lane & 15, lq = lane >> 4;
        v4f acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = (v4f){0.f, 0.f, 0.f, 0.f};
        float sa[4], sbv[4];
        bool clamped = false;                    // an A value hit the 2^100 clamp: the bound "A <= 1 / (weakest link into j*)" says a
                                                 // transition weaker than 2^-100 enters this pair — exp space cannot hold its terms
        auto prefetch = [&](int t0) {
            const int t = t0 + r;
            const bool okr = t < nt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = ib + 4 * c4 + e, j = jb + 4 * c4 + e;
                const float av = A[(okr && i < L) ? ((size_t)t * L + i) : (size_t)0];
                const float bv = Bt[(okr && j < L) ? ((size_t)(t + 1) * L + j) : (size_t)0];
                sa[e] = (okr && i < L) ? av * GD_LOG2E : NEG_INF;
                sbv[e] = (okr && j < L) ? bv * GD_LOG2E : NEG_INF;
            }
        };
        prefetch(0);
        for (int t0 = 0; t0 < nt; t0 += 16) {
            // this chunk's rows: per-row reference sb = ceil(max of the 64 beta values) (the 16 lanes of a DPP row hold one row)
            float bm = fmaxf(fmaxf(sbv[0], sbv[1]), fmaxf(sbv[2], sbv[3]));
            bm = gd_max16(bm);
            const bool rdead = bm == NEG_INF;
            const float sb = rdead ? 0.f : ceilf(bm);
            const float sh = sb - z2;
            clamped |= !rdead && fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3])) + sh > 100.f;
            v4f a4, b4;
            a4.x = rdead ? 0.f : __builtin_amdgcn_exp2f(fminf(sa[0] + sh, 100.f)); a4.y = rdead ? 0.f : __builtin_amdgcn_exp2f(fminf(sa[1] + sh, 100.f));
            a4.z = rdead ? 0.f : __builtin_amdgcn_exp2f(fminf(sa[2] + sh, 100.f)); a4.w = rdead ? 0.f : __builtin_amdgcn_exp2f(fminf(sa[3] + sh, 100.f));
            b4.x = rdead ? 0.f : __builtin_amdgcn_exp2f(sbv[0] - sb); b4.y = rdead ? 0.f : __builtin_amdgcn_exp2f(sbv[1] - sb);
            b4.z = rdead ? 0.f : __builtin_amdgcn_exp2f(sbv[2] - sb); b4.w = rdead ? 0.f : __builtin_amdgcn_exp2f(sbv[3] - sb);
            __syncthreads();                                             // previous chunk's fragment reads are done
            *reinterpret_cast<v4f*>(As + r * 64 + 4 * c4) = a4;
            *reinterpret_cast<v4f*>(Bs + r * 64 + 4 * c4) = b4;
            __syncthreads();
            if (t0 + 16 < nt) prefetch(t0 + 16);                         // next chunk's loads land under the MFMAs
            // wave w: rows i = 16 w .. 16 w + 15 of the tile, all four 16-column slices; 4 k-steps of 4 target rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float af = As[(4 * q + lq) * 64 + 16 * wave + lr];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float bf = Bs[(4 * q + lq) * 64 + 16 * s + lr];
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[s], 0, 0, 0);
                }
            }
        }
        if (__syncthreads_or(clamped)) {
            if (tid == 0) *mark = __uint_as_float(GD_REDO_MARK);        // ... the whole pair again, term by term, in pass 1
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) store(ib + 16 * wave + 4 * lq + rr, jb + 16 * s + lr, acc[s][rr]);
        }
    } else {
        // diagonal / window-edge pair: term by term in log space, rows staged through LDS (log2 domain), 16 elements per thread:
        // thread -> column j = jb + (tid & 63), rows i = ib + (tid >> 6) * 16 + e
        // (every term exp(alpha + beta + link - Z) is a probability: the link rides in the exponent, nothing can overflow)
        float sum[16], lk2[16];
        const int jl = tid & 63, ig = tid >> 6;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sum[e] = 0.f;
            const int i = ib + ig * 16 + e, j = jb + jl, d = j - i - 1;
            const bool ok = i < L && j < L && d >= 0 && d < TR;
            const float lk = K[ok ? ((size_t)i * TR + d) : (size_t)0];
            lk2[e] = ok ? lk * GD_LOG2E : NEG_INF;
        }
        for (int t0 = 0; t0 < nt; t0 += 16) {
            __syncthreads();
            {
                const int t = t0 + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = ib + 4 * c4 + e, j = jb + 4 * c4 + e;
                    As[r * 64 + 4 * c4 + e] = (t < nt && i < L) ? A[(size_t)t * L + i] * GD_LOG2E - z2 : NEG_INF;
                    Bs[r * 64 + 4 * c4 + e] = (t < nt && j < L) ? Bt[(size_t)(t + 1) * L + j] * GD_LOG2E : NEG_INF;
                }
            }
            __syncthreads();
            const int rows = min(16, nt - t0);
#pragma unroll 2
            for (int rr = 0; rr < rows; ++rr) {
                const float bv = Bs[rr * 64 + jl];
#pragma unroll
                for (int e = 0; e < 16; ++e) sum[e] += __builtin_amdgcn_exp2f((As[rr * 64 + ig * 16 + e] + bv) + lk2[e]);         // exp2(-inf) = 0
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = ib + ig * 16 + e, j = jb + jl, d = j - i - 1;
            if (i >= L || j >= L || d < 0 || d >= TR) continue;
            G[(size_t)i * TR + d] = (!dead && i < Lb && j < Lb) ? go * sum[e] : 0.f;
        }
    }
}

bool grad_dense_supported(int L, int TR) { return TR > 64 && L >= 128; }

int launch_dag_grad_links_dense(const float* g_out, const float* alpha, const float* beta, const float* links, const int64_t* out_len,
                                const int64_t* tgt_len, float* g_links, int B, int T, int L, int TR, hipStream_t st)
{
    const int NJ = (L + 63) / 64;
    const long npairs = (long)NJ * (NJ + 1) / 2;
    hipLaunchKernelGGL(dag_grad_links_dense_kernel<0>, dim3((unsigned)npairs, (unsigned)B), dim3(256), 0, st,
                       g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, NJ);
    int rc = check_launch("dag_loss_bwd(grad_links, dense block products)");
    if (rc) return rc;
    hipLaunchKernelGGL(dag_grad_links_dense_kernel<1>, dim3((unsigned)npairs, (unsigned)B), dim3(256), 0, st,
                       g_out, alpha, beta, links, out_len, tgt_len, g_links, B, T, L, TR, NJ);
    return check_launch("dag_loss_bwd(grad_links, dense log-space pairs)");
}

}  // namespace dsp
