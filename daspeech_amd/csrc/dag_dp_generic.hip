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
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace dsp {

constexpr int DP_THREADS = 1024;

// mode: 0 = alpha (K2), 1 = beta (K3).  grid = (B, ndir); blockIdx.y selects the direction when both run.
__global__ __launch_bounds__(DP_THREADS) void dag_logsum_generic_kernel(
    const float* __restrict__ match, const float* __restrict__ links,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ alpha, float* __restrict__ beta, int B, int T, int L, int TR, long sR, long sD, long sB)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rowA = smem;
    float* rowB = smem + L;
    const int b = blockIdx.x;
    const bool do_beta = (alpha == nullptr) ? true : (blockIdx.y == 1);
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* M = match + (size_t)b * T * L;
    const float* K = links + (size_t)b * sB;        // links[b][i][d] at K[i*sR + d*sD] (original or transposed copy)
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
                        mx = fmaxf(mx, prev[j - d] + K[(size_t)(j - d) * sR + (size_t)(d - 1) * sD]);
                    if (mx != NEG_INF) {                                 // :113-115
                        float s = 0.f;
                        for (int d = 1; d <= maxd; ++d)
                            s += __expf(prev[j - d] + K[(size_t)(j - d) * sR + (size_t)(d - 1) * sD] - mx);
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
                    const float* Kj = K + (size_t)j * sR;
                    float mx = NEG_INF;
                    for (int d = 1; d <= maxd; ++d) mx = fmaxf(mx, prev[j + d] + Kj[(size_t)(d - 1) * sD]);
                    if (mx != NEG_INF) {
                        float s = 0.f;
                        for (int d = 1; d <= maxd; ++d) s += __expf(prev[j + d] + Kj[(size_t)(d - 1) * sD] - mx);
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
// (st_src / st_dst: the launch's 64 status words move from caller memory to the library's status buffer on the way — see status_end in
//  dag_dp_banded.hip; a separate 256-byte device-to-device copy costs 5 us, 1 % of the C2 forward)
__global__ void dag_pick_loss_kernel(const float* alpha, const float* beta, const int64_t* out_len,
                                     const int64_t* tgt_len, float* loss, int B, int T, int L, int ld,
                                     const unsigned int* st_src, unsigned int* st_dst)
{
    if (st_src && blockIdx.x == 0 && threadIdx.x < 64) st_dst[threadIdx.x] = st_src[threadIdx.x];
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (beta) { loss[b] = beta[(size_t)b * T * ld]; return; }                 // ld: row pitch of alpha / beta (= L for dense tensors)
    int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    loss[b] = (Tb >= 1 && Tb <= T && Lb >= 1 && Lb <= L) ? alpha[(size_t)b * T * ld + (size_t)(Tb - 1) * ld + (Lb - 1)] : NEG_INF;
}

// K6: max-DP + trace.  Tie rule = smallest predecessor index (scan predecessors ascending, strict >).
__global__ __launch_bounds__(DP_THREADS) void dag_maxalpha_generic_kernel(
    const float* __restrict__ match, const float* __restrict__ links,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ alpha, int32_t* __restrict__ trace, int B, int T, int L, int TR, long sR, long sD, long sB)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* prev = smem;
    float* cur = smem + L;
    const int b = blockIdx.x;
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* M = match + (size_t)b * T * L;
    const float* K = links + (size_t)b * sB;
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
                    float v = prev[j - d] + K[(size_t)(j - d) * sR + (size_t)(d - 1) * sD];
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

// links[b][i][d] -> ET[b][d][i]: with lanes mapped to vertices the row-sequential kernels read links[j-d][d-1] (alpha) /
// links[j][d-1] (beta) for consecutive j — a stride-TR gather in the original layout (one 64-byte sector per lane and term;
// 39 GB of L2 sector traffic per training step at L~400, TR=L-1), unit stride in the transposed copy.
__global__ __launch_bounds__(256) void dag_transpose_links_kernel(const float* __restrict__ links, float* __restrict__ et, int L, int TR)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 32, d0 = blockIdx.x * 32;
    const float* src = links + (size_t)b * L * TR;
    float* dst = et + (size_t)b * L * TR;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) { const int i = i0 + r, d = d0 + tx; tile[r][tx] = (i < L && d < TR) ? src[(size_t)i * TR + d] : 0.f; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int d = d0 + r, i = i0 + tx; if (i < L && d < TR) dst[(size_t)d * L + i] = tile[tx][r]; }
}

static std::mutex g_et_mutex;
static std::unordered_map<unsigned long long, std::pair<void*, size_t>> g_et;

void* caller_ws_take(size_t n);

static const float* transposed_links(const float* links, int B, int L, int TR, hipStream_t st, long* sR, long* sD, long* sB)
{
    *sR = TR; *sD = 1; *sB = (long)L * TR;
    if (TR <= 64) return links;                         // short windows: the original layout is fine
    if (void* c = caller_ws_take((size_t)B * L * TR * sizeof(float))) {          // caller workspace first (capi: dsp_dag_*_workspace_bytes)
        hipLaunchKernelGGL(dag_transpose_links_kernel, dim3((TR + 31) / 32, (L + 31) / 32, B), dim3(256), 0, st, links, (float*)c, L, TR);
        *sR = 1; *sD = L;
        return (const float*)c;
    }
    std::lock_guard<std::mutex> lock(g_et_mutex);
    int dev = 0; (void)hipGetDevice(&dev);
    const unsigned long long key = ((unsigned long long)dev << 48) ^ (unsigned long long)(uintptr_t)st;
    auto& e = g_et[key];
    const size_t need = (size_t)B * L * TR * sizeof(float);
    if (e.second < need) {
        if (e.first) (void)hipFree(e.first);
        e.first = nullptr; e.second = 0;
        if (hipMalloc(&e.first, need) != hipSuccess) { (void)hipGetLastError(); return links; }      // fall back to the gather
        e.second = need;
    }
    hipLaunchKernelGGL(dag_transpose_links_kernel, dim3((TR + 31) / 32, (L + 31) / 32, B), dim3(256), 0, st, links, (float*)e.first, L, TR);
    *sR = 1; *sD = L;
    return (const float*)e.first;
}


// ------------------------------------------------------------------------------------------------ dense window (TR > 64)
// README's --max-transition-length 99999 makes TR = L-1: every vertex sees ALL earlier vertices.  Thread-per-column then
// serialises up to L terms per lane (94 us per DP row at L=400).  Here a WAVE owns a column and its 64 lanes split the
// predecessor distance d: the previous row is read from LDS at consecutive addresses and the transition weights at unit stride
// — for alpha / max-alpha from the "incoming" copy IN[b][j][d-1] = links[b][j-d][d-1] built by dag_incoming_links_kernel, for
// beta straight from links[b][j][:].  One online (max, sum) per lane, a wave shuffle reduction per column, one coalesced row
// store per DP row.
// `gate` (both kernels below): when given, the launch is a stand-by — it returns at once unless the word is non-zero.  That is how the
// matrix-core DP (dag_dp_dense_mfma.hip) hands a batch it gave up on to these log-space kernels without a host round trip.
__global__ __launch_bounds__(256) void dag_incoming_links_kernel(const float* __restrict__ links, float* __restrict__ in, int L, int TR,
                                                                 const unsigned int* gate)
{
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const int b = blockIdx.z;
    const float* src = links + (size_t)b * L * TR;
    for (int j = blockIdx.y; j < L; j += gridDim.y) {
        float* dst = in + ((size_t)b * L + j) * TR;
        for (int d = blockIdx.x * blockDim.x + threadIdx.x + 1; d <= TR; d += gridDim.x * blockDim.x)
            dst[d - 1] = (j - d >= 0) ? src[(size_t)(j - d) * TR + (d - 1)] : NEG_INF;
    }
}

template <int MODE>      // 0: log-sum (alpha, or beta when blockIdx.y == 1 / alpha == nullptr); 1: max + trace (alpha direction)
__global__ __launch_bounds__(DP_THREADS) void dag_dense_kernel(
    const float* __restrict__ match, const float* __restrict__ links, const float* __restrict__ incoming,
    const int64_t* __restrict__ out_len, const int64_t* __restrict__ tgt_len,
    float* __restrict__ alpha, float* __restrict__ beta, int32_t* __restrict__ trace, int B, int T, int L, int TR,
    unsigned int* __restrict__ row_count, unsigned long long* __restrict__ gran, unsigned int tag_base, const unsigned int* gate)
{
    if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;        // stand-by launch, not needed
    // gridDim.z = NS workgroups share one (sample, direction): workgroup s takes every NS-th group of NW*4 columns of a row
    // (interleaved: the work per column grows with the column index), writes its cells straight to the output row in HBM,
    // and publishes them in a tagged hand-off row from which all NS workgroups re-read the complete row into LDS.
    // The launcher keeps B * ndir * NS within what is co-resident (the hand-off spins), and the spin is bounded.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* prev = smem;
    float* cur = smem + L;
    int* carg = reinterpret_cast<int*>(smem + 2 * L);          // MODE 1: argmax row
    const int b = blockIdx.x;
    const int NS = gridDim.z, slice = blockIdx.z;
    unsigned int* err = row_count + 1;                                              // the strip kernels' status word (dsp_dag_last_launch_status)
    const bool do_beta = (MODE == 0) && ((alpha == nullptr) ? true : (blockIdx.y == 1));
    const int Lb = (int)out_len[b], Tb = (int)tgt_len[b];
    const float* M = match + (size_t)b * T * L;
    float* O = (do_beta ? beta : alpha) + (size_t)b * T * L;
    int32_t* Tr = (MODE == 1) ? trace + (size_t)b * T * L : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = DP_THREADS / 64;
    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    for (int t = (valid ? Tb : 0) + slice; t < T; t += NS)
        for (int j = tid; j < L; j += DP_THREADS) { O[(size_t)t * L + j] = NEG_INF; if (MODE == 1) Tr[(size_t)t * L + j] = -1; }
    if (!valid) return;
    {   // seed row (every workgroup keeps it in LDS; slice 0 writes it out)
        const int t = do_beta ? Tb - 1 : 0;
        for (int j = tid; j < L; j += DP_THREADS) {
            const float v = (do_beta ? (j == Lb - 1) : (j == 0)) ? M[(size_t)t * L + j] : NEG_INF;
            prev[j] = v;
            if (slice == 0) { O[(size_t)t * L + j] = v; if (MODE == 1) Tr[(size_t)t * L + j] = -1; }
        }
    }
    __syncthreads();
    __shared__ unsigned int s_abort;                 // set by a thread whose hand-off poll timed out: the workgroup gives up
    if (tid == 0) s_abort = 0;
    __syncthreads();
    for (int it = 1; it < Tb; ++it) {
        const int t = do_beta ? (Tb - 1 - it) : it;
        const float* Mt = M + (size_t)t * L;
        for (int j = tid; j < L; j += DP_THREADS) { cur[j] = NEG_INF; if (MODE == 1) carg[j] = -1; }
        __syncthreads();
        // QUARTER-WAVE per column: 16 lanes split the predecessor distance d, 4 columns per wave in flight; the per-column
        // reduction is 4 DPP steps inside a 16-lane row (no LDS permutes).
        const int q = lane >> 4, l16 = lane & 15;
        for (int jb = t + ((slice * NW + wave) * 4); jb < Lb; jb += NS * NW * 4) {
            const int j = jb + q;
            const bool live = j < Lb;
            const int maxd = live ? (do_beta ? min(Lb - 1 - j, TR) : min(j, TR)) : 0;
            const float* Kj = (do_beta ? links : incoming) + ((size_t)b * L + (live ? j : 0)) * TR;
            if (MODE == 0) {
                float m = NEG_INF, sacc = 0.f;
                for (int d0 = 0; d0 < maxd; d0 += 128) {                     // 8 independent loads per lane in flight
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int d = d0 + 16 * u + l16 + 1;
                        const int dc = min(d, max(maxd, 1));                    // unconditional (clamped) loads: all 8 in flight
                        const float kk = Kj[dc - 1];
                        const float pp = do_beta ? prev[min(j + dc, L - 1)] : prev[max(j - dc, 0)];
                        v[u] = (d <= maxd) ? (pp + kk) : NEG_INF;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (v[u] > m) { sacc = sacc * __expf(m - v[u]) + 1.f; m = v[u]; }
                        else if (v[u] != NEG_INF) sacc += __expf(v[u] - m);
                    }
                }
                auto merge = [&](float m2, float s2) {
                    const float nm = fmaxf(m, m2);
                    sacc = (nm == NEG_INF) ? 0.f : sacc * __expf(m - nm) + s2 * __expf(m2 - nm);
                    m = nm;
                };
                merge(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)),
                      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0xB1, 0xF, 0xF, true)));
                merge(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)),
                      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x4E, 0xF, 0xF, true)));
                merge(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)),
                      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x141, 0xF, 0xF, true)));
                merge(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)),
                      __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x140, 0xF, 0xF, true)));
                if (l16 == 0 && live) cur[j] = (m == NEG_INF) ? NEG_INF : (__logf(sacc) + m + Mt[j]);
            } else {
                float best = NEG_INF; int bd = 0;                  // among equal maxima keep the LARGEST d (smallest predecessor)
                for (int d0 = 0; d0 < maxd; d0 += 128) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int d = d0 + 16 * u + l16 + 1;
                        const int dc = min(d, max(maxd, 1));
                        const float kk = Kj[dc - 1];
                        const float pp = prev[max(j - dc, 0)];
                        v[u] = (d <= maxd) ? (pp + kk) : NEG_INF;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int d = d0 + 16 * u + l16 + 1;
                        if (v[u] > best || (v[u] == best && v[u] != NEG_INF)) { best = v[u]; bd = d; }
                    }
                }
                auto mergeb = [&](float b2, int d2) { if (b2 > best || (b2 == best && d2 > bd)) { best = b2; bd = d2; } };
                mergeb(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0xB1, 0xF, 0xF, true)),
                       __builtin_amdgcn_update_dpp(0, bd, 0xB1, 0xF, 0xF, true));
                mergeb(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0x4E, 0xF, 0xF, true)),
                       __builtin_amdgcn_update_dpp(0, bd, 0x4E, 0xF, 0xF, true));
                mergeb(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0x141, 0xF, 0xF, true)),
                       __builtin_amdgcn_update_dpp(0, bd, 0x141, 0xF, 0xF, true));
                mergeb(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, best), 0x140, 0xF, 0xF, true)),
                       __builtin_amdgcn_update_dpp(0, bd, 0x140, 0xF, 0xF, true));
                if (l16 == 0 && live) { cur[j] = best + Mt[j]; carg[j] = (best == NEG_INF) ? -1 : (j - bd); }
            }
        }
        __syncthreads();
        if (NS == 1) {
            for (int j = tid; j < L; j += DP_THREADS) {               // coalesced row store
                O[(size_t)t * L + j] = cur[j];
                if (MODE == 1) Tr[(size_t)t * L + j] = carg[j];
            }
            __syncthreads();
            float* tmp = prev; prev = cur; cur = tmp;
        } else {
            // own cells -> HBM output, and -> the hand-off row as 8-byte {tag, value} granules (agent-scope relaxed stores: a
            // granule is valid exactly when its tag says so — no fence, no counter; cdna_hip_programming.md G16 R2).  Two
            // hand-off rows alternate: a workgroup can only write row it+1 after it has read all of row it.
            unsigned long long* G = gran + ((size_t)(b * gridDim.y + blockIdx.y) * 2 + (it & 1)) * L;
            const unsigned int tag = tag_base + (unsigned int)it;
            for (int j = tid; j < L; j += DP_THREADS) {
                const int grp = (j - t) >> 2;                                       // 4-column group index along the row
                const bool mine = (j < t) ? (slice == 0) : (j >= Lb ? (slice == 0) : ((grp / NW) % NS == slice));
                if (mine) {
                    const float v = cur[j];
                    O[(size_t)t * L + j] = v; if (MODE == 1) Tr[(size_t)t * L + j] = carg[j];
                    __hip_atomic_store(G + j, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // the complete row, all workgroups' cells
            for (int j = tid; j < L; j += DP_THREADS) {
                unsigned long long x = __hip_atomic_load(G + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned int spins = 0;
                while ((unsigned int)(x >> 32) != tag) {
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(G + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > (1u << 18)) { atomicOr(err, 2u); s_abort = 1u; break; }
                }
                prev[j] = __uint_as_float((unsigned int)x);
            }
            __syncthreads();
            if (s_abort) return;                     // every workgroup of the launch times out once at most (status word says so)
        }
    }
}

static std::mutex g_in_mutex;
static std::unordered_map<unsigned long long, std::pair<void*, size_t>> g_in;

static const float* incoming_links(const float* links, int B, int L, int TR, hipStream_t st, const unsigned int* gate = nullptr)
{
    int gx0 = (TR + 255) / 256; if (gx0 > 8) gx0 = 8;
    const int gy = gate ? (L < 128 ? L : 128) : L;              // a stand-by launch keeps its (normally idle) grid small
    if (void* c = caller_ws_take((size_t)B * L * TR * sizeof(float))) {
        hipLaunchKernelGGL(dag_incoming_links_kernel, dim3(gx0, gy, B), dim3(256), 0, st, links, (float*)c, L, TR, gate);
        return (const float*)c;
    }
    std::lock_guard<std::mutex> lock(g_in_mutex);
    int dev = 0; (void)hipGetDevice(&dev);
    const unsigned long long key = ((unsigned long long)dev << 48) ^ (unsigned long long)(uintptr_t)st;
    auto& e = g_in[key];
    const size_t need = (size_t)B * L * TR * sizeof(float);
    if (e.second < need) {
        if (e.first) (void)hipFree(e.first);
        e.first = nullptr; e.second = 0;
        if (hipMalloc(&e.first, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        e.second = need;
    }
    hipLaunchKernelGGL(dag_incoming_links_kernel, dim3(gx0, gy, B), dim3(256), 0, st, links, (float*)e.first, L, TR, gate);
    return (const float*)e.first;
}

int launch_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L, hipStream_t st)
{
    const size_t lds2 = (size_t)L * sizeof(int32_t);
    if (lds2 > 160 * 1024) { set_error("dag_best_alignment: graph size L=%d too large for the back-trace row image", L); return DSP_EINVAL; }
    if (lds2 > 48 * 1024) set_max_dynamic_lds((const void*)dag_backtrace_kernel, (int)lds2);
    hipLaunchKernelGGL(dag_backtrace_kernel, dim3(B), dim3(256), lds2, st, trace, out_len, tgt_len, path, B, T, L);
    return check_launch("dag_best_alignment(back-trace)");
}

int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, unsigned int** counters, unsigned long long** halo, unsigned int* tag_base);

// how many workgroups may share one (sample, direction): everything the launch puts on the device must be co-resident (the
// per-row hand-off spins), with a factor 2 of slack for whatever else is running
template <typename K>
static int dense_slices(K kernel, size_t lds, int groups, int L)
{
    static int forced = -1;                      // DSP_DENSE_NS: sweeps only — the co-residency bound below still applies
    if (forced < 0) { const char* e = getenv("DSP_DENSE_NS"); forced = e ? atoi(e) : 0; }
    int nb = 0, dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, DP_THREADS, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); return 1; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { (void)hipGetLastError(); return 1; }
    // one workgroup per CU is assumed whatever the occupancy query says (it has over-reported for this launch shape, §5 census),
    // and a quarter of the CUs is left to whatever else is running: a launch that is not co-resident would time out
    (void)nb;
    int ns = (int)(((long)cus * 3 / 4) / (groups > 0 ? groups : 1));
    const int by_cols = L / (4 * (DP_THREADS / 64));               // at least one 64-column round per workgroup and row
    if (ns > by_cols) ns = by_cols;
    if (ns > 32) ns = 32;
    if (forced > 0 && forced < ns) ns = forced;
    return ns < 1 ? 1 : ns;
}

int launch_dag_fwd_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                           float* alpha, float* beta, int B, int T, int L, int TR, hipStream_t st)
{
    const size_t lds = 2 * (size_t)L * sizeof(float);
    if (lds > 160 * 1024) { set_error("dag_loss: graph size L=%d exceeds the generic kernel's LDS rows (max 20480)", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag_logsum_generic_kernel, (int)lds);
    const int ndir = (alpha && beta) ? 2 : 1;
    if (TR > 64) {                                       // dense window: wave-per-column kernel
        const float* in = alpha ? incoming_links(links, B, L, TR, st) : links;
        if (in) {
            if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag_dense_kernel<0>, (int)lds);
            const int NS = dense_slices(dag_dense_kernel<0>, lds, B * ndir, L);
            unsigned int* cnt = nullptr; unsigned long long* gran = nullptr; unsigned int tag_base = 0;
            int rcw = banded_acquire_ws(st, (size_t)B * ndir * 2 * L * sizeof(unsigned long long), T, &cnt, &gran, &tag_base);
            if (rcw) return rcw;
            hipLaunchKernelGGL(dag_dense_kernel<0>, dim3(B, ndir, NS), dim3(DP_THREADS), lds, st, match, links, in, out_len, tgt_len,
                               alpha, beta, (int32_t*)nullptr, B, T, L, TR, cnt, gran, tag_base, (const unsigned int*)nullptr);
            return check_launch("dag_loss_fwd(dense)");
        }
    }
    long sR, sD, sB;
    const float* lk = transposed_links(links, B, L, TR, st, &sR, &sD, &sB);
    hipLaunchKernelGGL(dag_logsum_generic_kernel, dim3(B, ndir), dim3(DP_THREADS), lds, st, match, lk, out_len, tgt_len,
                       alpha, beta, B, T, L, TR, sR, sD, sB);
    return check_launch("dag_loss_fwd(generic)");
}

// The log-space row kernels as a stand-by behind the matrix-core DP (see `gate` above).  The caller owns the status words (`cnt`: the
// error word is shared with its own kernel) and the hand-off rows (`gran`, dense_rows_gated_bytes()).  false = no stand-by possible for
// this shape (the DP row does not fit the LDS, or no room for the re-laid-out transition copy): the caller then never gives up.
size_t dense_rows_gated_bytes(int B, int L, int ndir) { return (size_t)B * ndir * 2 * L * sizeof(unsigned long long); }
bool dense_rows_gated_supported(int L) { return 2 * (size_t)L * sizeof(float) <= 160 * 1024; }
int launch_dag_dense_rows_gated(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                float* alpha, float* beta, int B, int T, int L, int TR,
                                unsigned int* cnt, unsigned long long* gran, unsigned int tag_base, const unsigned int* gate, hipStream_t st)
{
    const size_t lds = 2 * (size_t)L * sizeof(float);
    const int ndir = (alpha && beta) ? 2 : 1;
    const float* in = alpha ? incoming_links(links, B, L, TR, st, gate) : links;
    if (!in) { set_error("dag_loss_fwd: no memory for the stand-by log-space path (B*L*TR*4 bytes)"); return DSP_ENOSPC; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag_dense_kernel<0>, (int)lds);
    const int NS = dense_slices(dag_dense_kernel<0>, lds, B * ndir, L);
    hipLaunchKernelGGL(dag_dense_kernel<0>, dim3(B, ndir, NS), dim3(DP_THREADS), lds, st, match, links, in, out_len, tgt_len,
                       alpha, beta, (int32_t*)nullptr, B, T, L, TR, cnt, gran, tag_base, gate);
    return check_launch("dag_loss_fwd(dense, stand-by)");
}

bool status_export(hipStream_t st, const unsigned int** src, unsigned int** dst);
int launch_pick_loss(const float* alpha, const float* beta, const int64_t* out_len, const int64_t* tgt_len, float* loss,
                     int B, int T, int L, int ld, hipStream_t st)
{
    const unsigned int* ssrc = nullptr; unsigned int* sdst = nullptr;
    (void)status_export(st, &ssrc, &sdst);
    hipLaunchKernelGGL(dag_pick_loss_kernel, dim3((B + 63) / 64), dim3(64), 0, st, alpha, beta, out_len, tgt_len, loss, B, T, L, ld, ssrc, sdst);
    return check_launch("dag_pick_loss");
}

// max-DP + trace only (every cell (t, j >= t) of rows < T_b, no end-reach mask): the forward half of the alignment and the DP of
// the Viterbi graph decode (dsp_dag_max_alpha)
int launch_max_alpha_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                             float* alpha, int32_t* trace, int B, int T, int L, int TR, hipStream_t st)
{
    const size_t lds = 2 * (size_t)L * sizeof(float);
    if (lds > 160 * 1024) { set_error("dag_best_alignment: graph size L=%d too large (max 20480)", L); return DSP_EINVAL; }
    if (lds > 48 * 1024) set_max_dynamic_lds((const void*)dag_maxalpha_generic_kernel, (int)lds);
    if (TR > 64) {
        const float* in = incoming_links(links, B, L, TR, st);
        const size_t lds3 = 3 * (size_t)L * sizeof(float);
        if (in && lds3 <= 160 * 1024) {
            if (lds3 > 48 * 1024) set_max_dynamic_lds((const void*)dag_dense_kernel<1>, (int)lds3);
            const int NS = dense_slices(dag_dense_kernel<1>, lds3, B, L);
            unsigned int* cnt = nullptr; unsigned long long* gran = nullptr; unsigned int tag_base = 0;
            int rcw = banded_acquire_ws(st, (size_t)B * 2 * L * sizeof(unsigned long long), T, &cnt, &gran, &tag_base);
            if (rcw) return rcw;
            hipLaunchKernelGGL(dag_dense_kernel<1>, dim3(B, 1, NS), dim3(DP_THREADS), lds3, st, match, links, in, out_len, tgt_len,
                               alpha, (float*)nullptr, trace, B, T, L, TR, cnt, gran, tag_base, (const unsigned int*)nullptr);
            return check_launch("dag_best_alignment(dense)");
        }
    }
    long sR, sD, sB;
    const float* lk = transposed_links(links, B, L, TR, st, &sR, &sD, &sB);
    hipLaunchKernelGGL(dag_maxalpha_generic_kernel, dim3(B), dim3(DP_THREADS), lds, st, match, lk, out_len, tgt_len,
                       alpha, trace, B, T, L, TR, sR, sD, sB);
    return check_launch("dag_best_alignment(max-alpha)");
}

int launch_best_alignment_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                  float* alpha, int32_t* trace, int64_t* path, int B, int T, int L, int TR, hipStream_t st)
{
    int rc = launch_max_alpha_generic(match, links, out_len, tgt_len, alpha, trace, B, T, L, TR, st);
    if (rc) return rc;
    return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, st);
}

}  // namespace dsp
