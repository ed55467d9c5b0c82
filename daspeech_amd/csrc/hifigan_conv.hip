// hifigan_conv.hip — the HiFi-GAN generator convolution stack on MFMA (gfx950).
//
// Replaces the torch/MIOpen conv calls of hifi-gan/models.py:35-43 (ResBlock1.forward) and :100-119 (Generator.forward).
// One kernel serves every layer (see include/daspeech_hifigan.h for the "taps" formulation):
//     out = scale * ( bias + [res] + sum_k  W_k[M x CI] . lrelu(X)[CI x t + shift_k] )
// MI355X mapping
//   * activations channels-last fp16, so an MFMA B fragment (8 consecutive input channels of one time step) is ONE 16-byte
//     LDS read and a D fragment stores 4 consecutive output channels;
//   * the workgroup stages the lrelu'd input tile [NT + halo][CI] in LDS once (XOR-swizzled 16-byte chunks: the 16 lanes of
//     a read group hit different rows at the same channel offset) and re-uses it for all taps and all output channels —
//     the leaky_relu is applied once per element, not once per tap;
//   * weights [tap][co][ci] stream from L2 straight into A fragments (each element is used by exactly one wave, for all of
//     that wave's time tiles);
//   * mfma_f32_16x16x32_f16, fp32 accumulate; 8 waves as WM (channel) x WN (time) sub-tiles.
#include "common.h"
#include <stdlib.h>
#include "../../include/daspeech_hifigan.h"

namespace dsp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct HgParams {
    const _Float16* x; const _Float16* w; const float* bias; const _Float16* res; _Float16* out;
    int B, T, M, ntaps, Tout, Cout, out_mode, up_u, up_pad;
    float pre_slope, scale;
    int shifts[DSP_HG_MAX_TAPS];
    int min_shift, max_shift;
    int dbg;                      // ablation bits (HG_ABLATE env, timing experiments only): 1 no staging loads, 2 no MFMA loop, 4 no epilogue
    const int* lens; int len_mul; // per-sample valid input length = lens[b] * len_mul (NULL: T): rows beyond it are read as ZERO, so a
                                  // padded batch computes, on each sample's valid region, what the sample computes alone
};

__device__ __forceinline__ int hg_valid_len(const int* lens, int len_mul, int b, int T) {
    if (!lens) return T;
    const int v = lens[b] * len_mul;
    return v < T ? v : T;
}

template <int CI>
__device__ __forceinline__ int hg_swz(int row, int chunk) {
    constexpr int CH = CI / 8;                          // 16-byte chunks per row
    // ds_read_b128 is serviced in four NON-contiguous groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md
    // §LDS): a B-fragment read puts 8 rows at k-chunk q and the other 8 rows of the same 16 at chunk q ^ 1 into one group.  The r01
    // swizzle (chunk ^ row) is conflict-free for 16 rows at ONE chunk; with the real groups it collides whenever the tile row of
    // lane 0 is odd (every odd tap shift): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.27 - 0.46 (profiles/r03f_pmc_hifigan.txt).
    // XOR-ing only EVEN values leaves bit 0 of the slot to tell the two halves of a group apart, and 8 rows x 8 even values are
    // distinct for any base row: conflict-free for every shift.  (256-byte bank row = 16 slots of 16 bytes; rows narrower than that
    // share a bank row: the row's position inside it supplies the remaining slot bits.)
    if constexpr ((CH & (CH - 1)) != 0) return chunk;   // CI = 96: 12 chunks, not a power of two -> no swizzle
    else if constexpr (CH >= 16) return chunk ^ ((row & 7) << 1);
    else if constexpr (CH == 8) return chunk ^ (((row >> 1) & 3) << 1);
    else if constexpr (CH == 4) return chunk ^ (((row >> 2) & 1) << 1);
    else return chunk;
}

// Stage rows [t_first, t_first + R) of channels-last x (zero outside [0,T)) into the swizzled LDS tile, leaky_relu applied once
// per element.  U requests per thread are in flight before the first is consumed: the one-load-one-wait loop this replaces paid
// ~10 dependent HBM round trips per tile (r01f ISA reading: global_load / s_waitcnt vmcnt(0) / ds_write per iteration).
template <int CI, int U>
__device__ __forceinline__ void hg_stage_tile(char* tile, const _Float16* __restrict__ X, int T, int t_first, int R, float pre_slope,
                                              bool skip_loads, int tid)
{
    constexpr int CH = CI / 8;
    const _Float16 slope = (_Float16)pre_slope;
    const int n = R * CH;
    for (int e0 = tid; e0 < n; e0 += 512 * U) {
        h8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 512;
            const int row = e / CH, ch = e - row * CH;
            const int tg = t_first + row;
            v[u] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            if (e < n && tg >= 0 && tg < T && !skip_loads) v[u] = *reinterpret_cast<const h8*>(X + (size_t)tg * CI + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * 512;
            if (e < n) {
                const int row = e / CH, ch = e - row * CH;
                if (pre_slope != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[u][i] = v[u][i] > (_Float16)0 ? v[u][i] : v[u][i] * slope;
                }
                *reinterpret_cast<h8*>(tile + ((size_t)row * CH + hg_swz<CI>(row, ch)) * 16) = v[u];
            }
        }
    }
}

template <int CI, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(512) void hifigan_conv_kernel(HgParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = CI / 8;
    constexpr int MI = MT / WM / 16, NI = NT / WN / 16;
    static_assert(WM * WN == 8 && MI >= 1 && NI >= 1, "8 waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * NT;                      // first output column (input time index q) of the tile
    const int m0 = blockIdx.y * MT;
    const int R = NT + (p.max_shift - p.min_shift);
    const _Float16* X = p.x + (size_t)b * p.T * CI;
    // A tile whose outputs all lie past the sample's valid length is never read by anyone (every consumer clamps at ITS valid input
    // length = this layer's valid output length): with per-sample lengths a padded batch costs the sum of the lengths, not B * max.
    if (p.lens) {
        const int Tb = hg_valid_len(p.lens, p.len_mul, b, p.T);
        if (p.out_mode == DSP_HG_OUT_UPSAMPLE ? (t0 * p.up_u - p.up_pad >= Tb * p.up_u) : (t0 >= Tb)) return;
    }

    // ---- stage lrelu(x) tile: rows t0+min_shift .. t0+NT-1+max_shift, zero outside [0,T) ----
    hg_stage_tile<CI, 6>(smem, X, hg_valid_len(p.lens, p.len_mul, b, p.T), t0 + p.min_shift, R, p.pre_slope, (p.dbg & 1) != 0, tid);
    __syncthreads();

    f4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int co_base = m0 + wm * (MI * 16);
    const int tl_base = wn * (NI * 16);
    const int lr = lane & 15, lk = lane >> 4;
    // flattened (tap, 32-channel chunk) loop; 3-deep register ring of weight fragments (static indices: unrolled by 3) so the
    // L2 latency of the weight stream is covered by two steps of MFMAs
    constexpr int NC = CI / 32;
    const int nsteps = (p.dbg & 2) ? 0 : p.ntaps * NC;
    // weights are in FRAGMENT ORDER (dsp_hifigan_pack_weights): [tap][32-channel chunk][16-row tile][lane][8 halves], so one A
    // fragment is one contiguous 1 KB block — 8 full cache lines per load instead of 16 half-used ones (the vector L1's line rate,
    // not MFMA issue, was pacing the loop: r01f PMC + tiling sweep)
    const int Mt = (p.M + 15) >> 4;
    auto load_a = [&](int step, h8 (&a)[MI]) {
        const _Float16* Ws = p.w + (size_t)step * Mt * 512 + lane * 8;          // step = tap * NC + chunk
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int tile = (co_base >> 4) + i;
            a[i] = (tile < Mt) ? *reinterpret_cast<const h8*>(Ws + (size_t)tile * 512) : (h8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    h8 a0[MI], a1[MI], a2[MI];
    if (nsteps > 0) load_a(0, a0);
    if (nsteps > 1) load_a(1, a1);
    auto do_step = [&](int step, const h8 (&a)[MI]) {
        const int k = step / NC, c = step - k * NC;
        const int rshift = p.shifts[k] - p.min_shift;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = tl_base + j * 16 + lr + rshift;
            const h8 bf = *reinterpret_cast<const h8*>(smem + ((size_t)row * CH + hg_swz<CI>(row, c * 4 + lk)) * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bf, acc[i][j], 0, 0, 0);
        }
    };
    for (int step = 0; step < nsteps; step += 3) {
        if (step + 2 < nsteps) load_a(step + 2, a2);
        do_step(step, a0);
        if (step + 1 < nsteps) {
            if (step + 3 < nsteps) load_a(step + 3, a0);
            do_step(step + 1, a1);
        }
        if (step + 2 < nsteps) {
            if (step + 4 < nsteps) load_a(step + 4, a1);
            do_step(step + 2, a2);
        }
    }
    if (p.dbg & 4) return;

    // ---- epilogue through LDS: the D fragments (4 channels x 1 time step per lane) are transposed into a [NT][MT] fp16 tile
    //      so that global memory sees 16-byte, row-contiguous accesses for the residual / accumulate loads and the stores
    //      (fragment-shaped 8-byte stores at a row stride cost as much as the whole MFMA loop: 6.1 of 15.8 ms, ablation r01) ----
    constexpr int OPITCH = MT + 8;                    // halfs; +16 bytes per row de-conflicts the transposing ds_write_b64
    __syncthreads();                                  // all waves are done reading the input tile
    _Float16* otile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wm * (MI * 16) + i * 16 + lk * 4;          // row inside the M tile
        const int mrow = m0 + ml;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && mrow < p.M) {
            const int co = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? (mrow % p.Cout) : mrow;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = p.bias[co + e];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int tl = tl_base + j * 16 + lr;
            _Float16 hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(acc[i][j][e] + bv[e]);
            *reinterpret_cast<uint2*>(otile + (size_t)tl * OPITCH + ml) = *reinterpret_cast<uint2*>(hv);
        }
    }
    __syncthreads();
    constexpr int CPR = MT / 8;                       // 16-byte chunks per tile row
    constexpr int EU = 4;                             // residual / accumulate loads of EU chunks in flight together
    static_assert((NT * CPR) % 512 == 0, "tile chunks divide evenly");
    for (int e0 = tid; e0 < NT * CPR; e0 += 512 * EU) {
        h8 r8[EU], a8[EU];
        size_t off[EU];
        bool live[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            const int mrow = m0 + ch * 8;
            const int q = t0 + tl;
            int tout = q, co = mrow;
            if (p.out_mode == DSP_HG_OUT_UPSAMPLE) { const int r = mrow / p.Cout; co = mrow - r * p.Cout; tout = q * p.up_u + r - p.up_pad; }
            live[u] = e < NT * CPR && mrow < p.M && tout >= 0 && tout < p.Tout;
            off[u] = ((size_t)b * p.Tout + tout) * p.Cout + co;
            r8[u] = (h8){0, 0, 0, 0, 0, 0, 0, 0}; a8[u] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            if (live[u] && p.res) r8[u] = *reinterpret_cast<const h8*>(p.res + off[u]);
            if (live[u] && p.out_mode == DSP_HG_OUT_ACCUM) a8[u] = *reinterpret_cast<const h8*>(p.out + off[u]);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (!live[u]) continue;
            const int e = e0 + u * 512;
            const int tl = e / CPR, ch = e - tl * CPR;
            const h8 v = *reinterpret_cast<const h8*>(otile + (size_t)tl * OPITCH + ch * 8);
            h8 w8;
#pragma unroll
            for (int x = 0; x < 8; ++x) w8[x] = (_Float16)(p.scale * ((float)v[x] + (float)r8[u][x]) + (float)a8[u][x]);
            *reinterpret_cast<h8*>(p.out + off[u]) = w8;
        }
    }
}

template <int CI, int MT, int NT, int WM, int WN>
static int hg_launch(const HgParams& p, hipStream_t st)
{
    const int R = NT + (p.max_shift - p.min_shift);
    size_t lds = (size_t)R * CI * 2;
    const size_t lds_out = (size_t)NT * (MT + 8) * 2;
    if (lds_out > lds) lds = lds_out;
    if (lds > 160 * 1024) { set_error("hifigan_conv: input tile %zu bytes exceeds LDS", lds); return DSP_EINVAL; }
    auto k = hifigan_conv_kernel<CI, MT, NT, WM, WN>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    // columns: conv -> T outputs; upsample -> q = 0..T (the last column feeds the tail of the transposed conv)
    const int ncol = (p.out_mode == DSP_HG_OUT_UPSAMPLE) ? p.T + 1 : p.T;
    dim3 grid((ncol + NT - 1) / NT, (p.M + MT - 1) / MT, p.B);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);
    return check_launch("hifigan_conv");
}

// ---------------------------------------------------------------------------------------------------------------------------
// One ResBlock1 unit (hifi-gan/models.py:38-42) in ONE launch:   out = scale * ( x + b2 + c2( lrelu( b1 + c1( lrelu(x) ) ) ) ) [+ out]
// c1 = Conv1d(C, C, K, dilation d), c2 = Conv1d(C, C, K, dilation 1).  The layer-at-a-time chain moves 5 activation tensors per
// unit through HBM (x in, h out, h in, x in again as the residual, out); here the intermediate h lives in LDS only: x tile
// (with both halos) in, out tile out, the residual re-read hits L2.  h is rounded to fp16 exactly where the chain rounds it, the
// MFMA step order is the chain's, so the result is bit-identical to the two-launch path (tests/test_gpu_hifigan.py).
//   tile: NT output columns; intermediate columns m = 0..NTI-1 (NTI = NT+16) stand for times t0-8+m, so c2's halo (<= 8) is inside;
//   x rows r = 0..NTI+2*h1-1 stand for times t0-8-h1+r.  c1 of column m reads rows m + k*d; c2 of column n reads mid rows
//   n + 8 - h2 + k.  h outside [0,T) is ZERO (c2's zero padding), not c1 of zero-padded x.
struct HgUnitParams {
    const _Float16* x; const _Float16* w1; const float* b1; const _Float16* w2; const float* b2; _Float16* out;
    int B, T, ntaps, dil, accumulate;
    float slope, scale;
    const int* lens; int len_mul;          // as HgParams
};

template <int C, int NT, int WM, int WN>
__global__ __launch_bounds__(512) void hifigan_resunit_kernel(HgUnitParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = C / 8, NC = C / 32;
    constexpr int NTI = NT + 16;
    constexpr int MI = C / WM / 16, NI = NTI / WN / 16;
    constexpr int OPITCH = C + 8;
    static_assert(WM * WN == 8 && MI >= 1 && NI >= 1 && NTI % (WN * 16) == 0, "8 waves, intermediate tile divisible");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z, t0 = blockIdx.x * NT;
    const int h1 = p.dil * (p.ntaps - 1) / 2, h2 = (p.ntaps - 1) / 2;
    const int R1 = NTI + 2 * h1;
    const size_t xin_bytes = (size_t)R1 * C * 2, ot_bytes = (size_t)NT * OPITCH * 2;
    // ONE LDS region, three tenants in turn: the x tile, then the intermediate (written from the accumulators only after every wave
    // has finished c1 — the whole intermediate tile sits in registers across that barrier), then the output tile.  Half the LDS of
    // separate regions: C = 128 fits twice per CU, C = 64 four times.
    (void)xin_bytes; (void)ot_bytes;
    char* xin = smem;
    char* mid = smem;
    const _Float16* X = p.x + (size_t)b * p.T * C;
    const _Float16 slope = (_Float16)p.slope;

    const int Tb = hg_valid_len(p.lens, p.len_mul, b, p.T);
    if (t0 >= Tb) return;                             // (see hifigan_conv_kernel: nobody reads past the valid length)
    hg_stage_tile<C, 6>(xin, X, Tb, t0 - 8 - h1, R1, p.slope, false, tid);
    __syncthreads();

    f4 acc[MI][NI];
    const int co_base = wm * (MI * 16);
    const int nsteps = p.ntaps * NC;
    // one convolution over an LDS tile: the flattened (tap, 32-channel chunk) loop of hifigan_conv_kernel, 3-deep weight ring
    auto conv = [&](const _Float16* W, const char* tile, int row0, int rstep) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        auto load_a = [&](int step, h8 (&a)[MI]) {                    // fragment-order weights, see hifigan_conv_kernel
            const _Float16* Ws = W + (size_t)step * (C / 16) * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const h8*>(Ws + (size_t)((co_base >> 4) + i) * 512);
        };
        auto do_step = [&](int step, const h8 (&a)[MI]) {
            const int k = step / NC, c = step - k * NC;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // straight-line: c2's 16 columns past NT (last wave, last j) read the 16 slack rows behind `mid`; never stored
                const int row = (wn * NI + j) * 16 + lr + row0 + k * rstep;
                const h8 bf = *reinterpret_cast<const h8*>(tile + ((size_t)row * CH + hg_swz<C>(row, c * 4 + lk)) * 16);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], bf, acc[i][j], 0, 0, 0);
            }
        };
        h8 a0[MI], a1[MI], a2[MI];
        load_a(0, a0);
        if (nsteps > 1) load_a(1, a1);
        for (int step = 0; step < nsteps; step += 3) {
            if (step + 2 < nsteps) load_a(step + 2, a2);
            do_step(step, a0);
            if (step + 1 < nsteps) {
                if (step + 3 < nsteps) load_a(step + 3, a0);
                do_step(step + 1, a1);
            }
            if (step + 2 < nsteps) {
                if (step + 4 < nsteps) load_a(step + 4, a1);
                do_step(step + 2, a2);
            }
        }
    };

    // ---- c1 over the NTI intermediate columns -> mid = lrelu(fp16(acc + b1)), zero outside [0,T) ----
    conv(p.w1, xin, 0, p.dil);
    __syncthreads();                                  // every wave is done reading the x tile: its space becomes the intermediate
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int co = co_base + i * 16 + lk * 4;
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.b1 ? p.b1[co + e] : 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int m = (wn * NI + j) * 16 + lr;
            const int tm = t0 - 8 + m;
            _Float16 hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 h = (_Float16)(acc[i][j][e] + bv[e]);
                h = h > (_Float16)0 ? h : h * slope;
                hv[e] = (tm >= 0 && tm < Tb) ? h : (_Float16)0;
            }
            *reinterpret_cast<uint2*>(mid + ((size_t)m * CH + hg_swz<C>(m, co >> 3)) * 16 + (co & 4) * 2) = *reinterpret_cast<uint2*>(hv);
        }
    }
    __syncthreads();                                  // intermediate complete

    // ---- c2 over the NT output columns ----
    conv(p.w2, mid, 8 - h2, 1);
    __syncthreads();                                  // every wave is done reading the intermediate: its space becomes the output tile
    _Float16* otile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = co_base + i * 16 + lk * 4;
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = p.b2 ? p.b2[ml + e] : 0.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (wn * NI + j < NT / 16) {
                const int tl = (wn * NI + j) * 16 + lr;
                _Float16 hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(acc[i][j][e] + bv[e]);
                *reinterpret_cast<uint2*>(otile + (size_t)tl * OPITCH + ml) = *reinterpret_cast<uint2*>(hv);
            }
        }
    }
    __syncthreads();
    constexpr int EU = 4;
    for (int e0 = tid; e0 < NT * CH; e0 += 512 * EU) {
        h8 r8[EU], a8[EU];
        bool live[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + u * 512;
            const int tl = e / CH, ch = e - tl * CH;
            live[u] = e < NT * CH && t0 + tl < p.T;
            const size_t o = ((size_t)b * p.T + t0 + tl) * C + ch * 8;
            r8[u] = (h8){0, 0, 0, 0, 0, 0, 0, 0}; a8[u] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            if (live[u]) r8[u] = *reinterpret_cast<const h8*>(p.x + o);              // the unit's residual is its own input
            if (live[u] && p.accumulate) a8[u] = *reinterpret_cast<const h8*>(p.out + o);
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (!live[u]) continue;
            const int e = e0 + u * 512;
            const int tl = e / CH, ch = e - tl * CH;
            const size_t o = ((size_t)b * p.T + t0 + tl) * C + ch * 8;
            const h8 v = *reinterpret_cast<const h8*>(otile + (size_t)tl * OPITCH + ch * 8);
            h8 w8;
#pragma unroll
            for (int x = 0; x < 8; ++x) w8[x] = (_Float16)(p.scale * ((float)v[x] + (float)r8[u][x]) + (float)a8[u][x]);
            *reinterpret_cast<h8*>(p.out + o) = w8;
        }
    }
}

// the three tenants of the unit's LDS region: x tile, intermediate (+ 16 slack rows, see conv()), output tile
static size_t hg_unit_lds(int C, int NT, int h1)
{
    const size_t xin = (size_t)(NT + 16 + 2 * h1) * C * 2, mid = (size_t)(NT + 32) * C * 2, ot = (size_t)NT * (C + 8) * 2;
    const size_t m = xin > mid ? xin : mid;
    return ((m > ot ? m : ot) + 255) / 256 * 256;
}

template <int C, int NT, int WM, int WN>
static int hg_unit_launch(const HgUnitParams& p, hipStream_t st)
{
    const int h1 = p.dil * (p.ntaps - 1) / 2;
    const size_t lds = hg_unit_lds(C, NT, h1);
    if (lds > 160 * 1024) { set_error("hifigan_resunit: tiles need %zu bytes of LDS", lds); return DSP_EINVAL; }
    auto k = hifigan_resunit_kernel<C, NT, WM, WN>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((p.T + NT - 1) / NT, 1, p.B), dim3(512), lds, st, p);
    return check_launch("hifigan_resunit");
}

// [ntaps][M][CI] (tap-major rows) -> fragment order [ntaps][CI/32][ceil(M/16)][64 lanes][8]: lane = lk*16 + lr holds row
// tile*16 + lr, channels chunk*32 + lk*8 .. +7 (the A operand of mfma_f32_16x16x32_f16); rows >= M are zero
__global__ void hg_pack_weights_kernel(const _Float16* __restrict__ w, _Float16* __restrict__ out, int ntaps, int M, int CI)
{
    const int Mt = (M + 15) >> 4, NC = CI / 32;
    const long n = (long)ntaps * NC * Mt * 512;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e & 7), ln = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int tile = (int)(r % Mt); r /= Mt;
        const int c = (int)(r % NC); const int k = (int)(r / NC);
        const int co = tile * 16 + (ln & 15), ci = c * 32 + (ln >> 4) * 8 + h;
        out[e] = (co < M) ? w[((size_t)k * M + co) * CI + ci] : (_Float16)0;
    }
}

__global__ void hg_pack_kernel(const float* __restrict__ x, _Float16* __restrict__ out, long n_rows, int C, int Cpad)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_rows * Cpad; e += (long)gridDim.x * blockDim.x) {
        const long r = e / Cpad; const int c = (int)(e - r * Cpad);
        out[e] = (c < C) ? (_Float16)x[r * C + c] : (_Float16)0;
    }
}

__global__ __launch_bounds__(256) void hg_post_kernel(const _Float16* __restrict__ x, const float* __restrict__ w, float bias,
                                                      float* __restrict__ wav, int T, int C, int K, float slope,
                                                      const int* __restrict__ lens, int len_mul)
{
    extern __shared__ float ws[];              // [K][C]
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const _Float16* X = x + (size_t)b * T * C;
    const int Tb = hg_valid_len(lens, len_mul, b, T);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        if (t >= Tb) { wav[(size_t)b * T + t] = 0.f; continue; }       // past the utterance: silence (the layers above skipped it)
        float acc = bias;
        for (int k = 0; k < K; ++k) {
            const int tt = t + k - (K - 1) / 2;
            if (tt < 0 || tt >= Tb) continue;
            const _Float16* xr = X + (size_t)tt * C;
            for (int c = 0; c < C; c += 8) {
                const h8 v = *reinterpret_cast<const h8*>(xr + c);
#pragma unroll
                for (int i = 0; i < 8; ++i) { float f = (float)v[i]; f = f > 0.f ? f : f * slope; acc += f * ws[k * C + c + i]; }
            }
        }
        wav[(size_t)b * T + t] = tanhf(acc);
    }
}

}  // namespace dsp

using namespace dsp;

static int hg_conv_one(const void* x, const void* w, const float* bias, const void* res, void* out,
                       int B, int T, int CI, int M, int ntaps, const int* host_shifts, float pre_slope, float scale,
                       int out_mode, int up_u, int up_pad, int Tout, int Cout, hipStream_t st, const int* lens = nullptr, int len_mul = 1)
{
    if (B < 0 || T < 1 || M < 1 || ntaps < 1 || ntaps > DSP_HG_MAX_TAPS || !host_shifts) { set_error("hifigan_conv: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w || !out) { set_error("hifigan_conv: null pointer"); return DSP_EINVAL; }
    if ((Cout & 3) || (out_mode == DSP_HG_OUT_UPSAMPLE ? (M != up_u * Cout) : (M != Cout || Tout != T))) {
        set_error("hifigan_conv: inconsistent M=%d Cout=%d mode=%d", M, Cout, out_mode); return DSP_EINVAL; }
    HgParams p;
    p.x = (const _Float16*)x; p.w = (const _Float16*)w; p.bias = bias; p.res = (const _Float16*)res; p.out = (_Float16*)out;
    p.B = B; p.T = T; p.M = M; p.ntaps = ntaps; p.Tout = Tout; p.Cout = Cout; p.out_mode = out_mode; p.up_u = up_u; p.up_pad = up_pad;
    p.pre_slope = pre_slope; p.scale = scale; p.lens = lens; p.len_mul = len_mul;
    { static int ablate = -1; if (ablate < 0) { const char* ab = getenv("HG_ABLATE"); ablate = ab ? atoi(ab) : 0; } p.dbg = ablate; }
    p.min_shift = p.max_shift = host_shifts[0];
    for (int k = 0; k < ntaps; ++k) { p.shifts[k] = host_shifts[k]; p.min_shift = min(p.min_shift, host_shifts[k]); p.max_shift = max(p.max_shift, host_shifts[k]); }
    // tile / wave-grid choices: sweeps r01e (tap-major weights) and r01f (fragment-order weights; 512-column tiles and MI=4 wave tiles
    // for C = 128 / 256 were 20-40 % slower: fewer, longer workgroups and a longer un-overlapped stage/epilogue per tile)
    switch (CI) {
        case 512: return hg_launch<512, 256, 128, 8, 1>(p, st);
        case 256: return hg_launch<256, 256, 128, 8, 1>(p, st);
        case 128: return hg_launch<128, 128, 256, 4, 2>(p, st);       // half-size tiles (2-3 workgroups per CU) are 7-10 % slower too
        case 96:  return hg_launch<96, 256, 128, 8, 1>(p, st);
        case 64:  return hg_launch<64, 64, 512, 2, 4>(p, st);      // 512-column tiles: 12.1 vs 13.0 ms per B=32 pass (sweep r01e;
                                                                    // 64x64 wave tiles, and larger tiles for C = 32 / 128 / 256, were slower or equal)
        case 32:  return hg_launch<32, 32, 512, 1, 8>(p, st);
    }
    set_error("hifigan_conv: unsupported input channel count %d", CI);
    return DSP_EINVAL;
}

extern "C" int dsp_hifigan_conv(const void* x, const void* w, const float* bias, const void* res, void* out,
                                int B, int T, int CI, int M, int ntaps, const int* host_shifts, float pre_slope, float scale,
                                int out_mode, int up_u, int up_pad, int Tout, int Cout, dsp_stream_t stream)
{
    return hg_conv_one(x, w, bias, res, out, B, T, CI, M, ntaps, host_shifts, pre_slope, scale, out_mode, up_u, up_pad, Tout, Cout,
                       as_stream(stream));
}

static int hg_unit_one(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
                       int B, int T, int C, int ntaps, int dil, float slope, float scale, int accumulate, hipStream_t st,
                       const int* lens = nullptr, int len_mul = 1)
{
    if (B < 0 || T < 1 || ntaps < 1 || ntaps > DSP_HG_MAX_TAPS || !(ntaps & 1) || dil < 1) { set_error("hifigan_resunit: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w1 || !w2 || !out || x == out) { set_error("hifigan_resunit: null or aliased pointer"); return DSP_EINVAL; }
    HgUnitParams p;
    p.x = (const _Float16*)x; p.w1 = (const _Float16*)w1; p.b1 = b1; p.w2 = (const _Float16*)w2; p.b2 = b2; p.out = (_Float16*)out;
    p.B = B; p.T = T; p.ntaps = ntaps; p.dil = dil; p.accumulate = accumulate; p.slope = slope; p.scale = scale;
    p.lens = lens; p.len_mul = len_mul;
    // C = 128: 78 KB per workgroup, two per CU (353 us per unit vs 2 x 200 us for its two-launch chain at B=32)
    switch (C) {
        // 112-column tiles (91 KB, one workgroup per CU) when they fill the chip; 48-column tiles (58 KB, two per CU) for small batches:
        // B=8 x 330 frames is 192 workgroups at 112 columns (70.6 -> 60.3 us per unit), B=32 prefers 112 (175 vs 208 us)
        case 256: return (long)((T + 111) / 112) * B >= 384 ? hg_unit_launch<256, 112, 8, 1>(p, st) : hg_unit_launch<256, 48, 8, 1>(p, st);
        case 128: return hg_unit_launch<128, 240, 4, 2>(p, st);
        // the wider tile (half the weight-fragment loads per MFMA, 6 % faster at B=32) only when it still gives two workgroups per CU
        case 64:  return (long)((T + 495) / 496) * B >= 512 ? hg_unit_launch<64, 496, 2, 4>(p, st) : hg_unit_launch<64, 240, 2, 4>(p, st);
        case 32:  return (long)((T + 1007) / 1008) * B >= 512 ? hg_unit_launch<32, 1008, 1, 8>(p, st) : hg_unit_launch<32, 496, 1, 8>(p, st);
    }
    set_error("hifigan_resunit: unsupported channel count %d (32, 64, 128)", C);
    return DSP_EINVAL;
}

extern "C" int dsp_hifigan_resunit(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
                                   int B, int T, int C, int ntaps, int dil, float slope, float scale, int accumulate,
                                   dsp_stream_t stream)
{
    return hg_unit_one(x, w1, b1, w2, b2, out, B, T, C, ntaps, dil, slope, scale, accumulate, as_stream(stream));
}

extern "C" int dsp_hifigan_resunit_supported(int C, int ntaps, int dil)
{
    if (!(C == 32 || C == 64 || C == 128 || C == 256) || ntaps < 1 || !(ntaps & 1) || ntaps > DSP_HG_MAX_TAPS || dil < 1) return 0;
    const int h1 = dil * (ntaps - 1) / 2, NT = (C == 32) ? 1008 : (C == 64) ? 496 : (C == 128) ? 240 : 112;        // the largest tile the launcher may pick
    return hg_unit_lds(C, NT, h1) <= 160 * 1024;
}

// The generator is ~100 of these layers per call; driven one ctypes call at a time the host, not the GPU, sets the pace at
// vocoder batch sizes.  One call walks a whole layer table.
static int hg_chain(const dsp_hg_layer* layers, int n_layers, int B, const int* lens, int T0, hipStream_t st)
{
    if (n_layers < 0 || (n_layers > 0 && !layers)) { set_error("hifigan_conv_chain: bad layer table"); return DSP_EINVAL; }
    if (lens && T0 < 1) { set_error("hifigan_conv_chain: lens given without the padded frame count T0"); return DSP_EINVAL; }
    for (int i = 0; i < n_layers; ++i) {
        const dsp_hg_layer& l = layers[i];
        int mul = 1;
        if (lens) {                                                        // this layer's input runs at l.T / T0 steps per mel frame
            if (l.T % T0) { set_error("hifigan_conv_chain: layer %d length %d is not a multiple of T0 = %d", i, l.T, T0); return DSP_EINVAL; }
            mul = l.T / T0;
        }
        if (l.w2) {                                                        // fused ResBlock unit: x -> c1 -> c2 -> + x
            const int dil = l.ntaps > 1 ? l.shifts[l.ntaps / 2 + 1] : 1;
            int rc = hg_unit_one(l.x, l.w, l.bias, l.w2, l.bias2, l.out, B, l.T, l.CI, l.ntaps, dil, l.pre_slope, l.scale,
                                 l.out_mode == DSP_HG_OUT_ACCUM, st, lens, mul);
            if (rc) return rc;
            continue;
        }
        int rc = hg_conv_one(l.x, l.w, l.bias, l.res, l.out, B, l.T, l.CI, l.M, l.ntaps, l.shifts, l.pre_slope, l.scale,
                             l.out_mode, l.up_u, l.up_pad, l.Tout, l.Cout, st, lens, mul);
        if (rc) return rc;
    }
    return DSP_OK;
}

extern "C" int dsp_hifigan_conv_chain(const dsp_hg_layer* layers, int n_layers, int B, dsp_stream_t stream)
{
    return hg_chain(layers, n_layers, B, nullptr, 0, as_stream(stream));
}

extern "C" int dsp_hifigan_conv_chain_lens(const dsp_hg_layer* layers, int n_layers, int B, const int* lens, int T0, dsp_stream_t stream)
{
    return hg_chain(layers, n_layers, B, lens, T0, as_stream(stream));
}

extern "C" long dsp_hifigan_packed_weight_elems(int ntaps, int M, int CI)
{
    if (ntaps < 1 || M < 1 || CI < 32 || (CI & 31)) return -1;
    return (long)ntaps * (CI / 32) * ((M + 15) / 16) * 512;
}

extern "C" int dsp_hifigan_pack_weights(const void* w, void* out, int ntaps, int M, int CI, dsp_stream_t stream)
{
    const long n = dsp_hifigan_packed_weight_elems(ntaps, M, CI);
    if (n < 0 || !w || !out || w == out) { set_error("hifigan_pack_weights: bad arguments"); return DSP_EINVAL; }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hg_pack_weights_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const _Float16*)w, (_Float16*)out, ntaps, M, CI);
    return check_launch("hifigan_pack_weights");
}

extern "C" int dsp_hifigan_pack_input(const float* x, void* out, int B, int T, int C, int Cpad, dsp_stream_t stream)
{
    if (B < 0 || T < 0 || C < 1 || Cpad < C) { set_error("hifigan_pack_input: bad sizes"); return DSP_EINVAL; }
    const long n = (long)B * T;
    if (n == 0) return DSP_OK;
    int grid = (int)((n * Cpad + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(hg_pack_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, (_Float16*)out, n, C, Cpad);
    return check_launch("hifigan_pack_input");
}

static int hg_post(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope, const int* lens, int len_mul,
                   hipStream_t st)
{
    if (B < 0 || T < 1 || C < 8 || (C & 7) || K < 1) { set_error("hifigan_post: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    int gx = (T + 255) / 256; if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(hg_post_kernel, dim3(gx, B), dim3(256), (size_t)K * C * 4, st, (const _Float16*)x, w, bias, wav, T, C, K, slope, lens, len_mul);
    return check_launch("hifigan_post");
}

extern "C" int dsp_hifigan_post(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                                dsp_stream_t stream)
{
    return hg_post(x, w, bias, wav, B, T, C, K, slope, nullptr, 1, as_stream(stream));
}

extern "C" int dsp_hifigan_post_lens(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                                     const int* lens, int len_mul, dsp_stream_t stream)
{
    return hg_post(x, w, bias, wav, B, T, C, K, slope, lens, len_mul, as_stream(stream));
}
