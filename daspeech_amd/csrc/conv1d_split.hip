// conv1d_split.hip — fp32-accurate Conv1d on the fp16 matrix cores ("3 x fp16" split), for the FastSpeech2 FFT feed-forward
// (fairseq/models/text_to_speech/fastspeech2.py:42-63 PositionwiseFeedForward: Conv1d(256,1024,9) - ReLU - Conv1d(1024,256,9)).
//
// MIOpen serves these fp32 convolutions at 60-70 TFLOP/s (1.0 / 1.2 ms each at B=32 x 483 frames: 9 of the 41 ms of the S2ST
// pipeline).  The mel tolerance (1e-4) rules out plain fp16 operands, but not the matrix cores: with
//     x = xh + xl * 2^-11,  w = wh + wl * 2^-11      (xh = fp16(x), xl = fp16((x - xh) * 2^11); same for w)
// the three products  xh.wh,  xh.wl,  xl.wh  are EXACT in the fp32 accumulator (11 x 11 significant bits) and the dropped xl.wl term
// is 2^-22 relative: fp32-GEMM accuracy at a third of the fp16 MFMA rate.  The lo parts carry their own 2^11 scale so that they live
// in the normal fp16 range; their products go to a second accumulator that is folded in with 2^-11 at the end.
//   x [B,T,CI] fp32 channels-last (row stride ldx), taps k with shift k - (K-1)/2 ("same" padding), weights pre-split and stored in
//   MFMA fragment order (dsp_conv1d_split_pack), out [B,T,M] fp32 = [out +] bias + conv, optional ReLU.
#include "common.h"
#include <stdlib.h>
#include "../../include/daspeech_decode.h"

namespace dsp {

typedef _Float16 cs_h8 __attribute__((ext_vector_type(8)));
typedef float cs_f4 __attribute__((ext_vector_type(4)));

struct CsParams {
    const float* x; const _Float16* wh; const _Float16* wl; const float* bias; float* out;
    int B, T, M, ntaps; long ldx, ldo;
    int relu, accumulate;          // relu: activation after the bias — 0 none, 1 ReLU, 2 SiLU, 3 GELU(erf)
    int nslices; long wslice;      // the input is nslices x CI channels wide; slice s uses weights + s * wslice (halves)
    const float* res; long ldr; float alpha;      // out = res + alpha * act(bias + conv)   (res may be NULL: out = alpha * act(...))
    const float* ln_w; const float* ln_b; float ln_eps;   // LayerNorm of the input rows while they are staged (256-channel one-tap layers: the row sits in half a wave)
    int kparts, tap_groups; float* part;          // split-K: kparts = nslices * tap_groups workgroups per output tile, raw partial sums to part [kparts][B][T][M]
    const int* lens; int slack;                   // ragged batch: rows >= lens[b] + slack of sample b are padding nobody reads — their
};                                                // tiles are not computed, the output rows are written as zeros

template <int CI>
__device__ __forceinline__ int cs_swz(int row, int chunk) {
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

template <int CI, int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv1d_split_kernel(CsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    constexpr int CH = CI / 8, NC = CI / 32;
    constexpr int MI = MT / WM / 16, NI = NT / WN / 16;
    constexpr int NTH = WM * WN * 64;                     // 8 waves, or 16 (r05: four waves per SIMD under one resident workgroup)
    static_assert((WM * WN == 8 || WM * WN == 16) && MI >= 1 && NI >= 1, "8 or 16 waves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 15, lk = lane >> 4;
    const int kp = p.kparts > 1 ? (int)(blockIdx.y % p.kparts) : 0;
    const int b = blockIdx.z, t0 = blockIdx.x * NT, m0 = (p.kparts > 1 ? (int)(blockIdx.y / p.kparts) : (int)blockIdx.y) * MT;
    if (p.lens && t0 >= p.lens[b] + p.slack) {          // a tile of padding rows (block-uniform): zeros, so that they stay finite
        if (p.accumulate) return;
        const int cw = min(MT, p.M - m0) >> 2;          // float4 columns of this tile
        const int rows = min(NT, p.T - t0);
        float* dst = p.kparts > 1 ? p.part + ((size_t)kp * p.B + b) * p.T * p.M : p.out + (size_t)b * p.T * p.ldo;     // split-K: the partial sums are zero
        const size_t ld = p.kparts > 1 ? (size_t)p.M : (size_t)p.ldo;
        for (int e = tid; e < rows * cw; e += NTH) {
            const int r = e / cw, c4 = e - r * cw;
            *reinterpret_cast<float4*>(dst + (size_t)(t0 + r) * ld + m0 + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int P = (p.ntaps - 1) / 2;
    const int R = NT + p.ntaps - 1;
    char* th = cs_smem;                                   // hi tile  [R][CI] halves, 16-byte chunks XOR-swizzled
    char* tl = cs_smem + (size_t)R * CI * 2;              // lo tile
    const float* X = p.x + (size_t)b * p.T * p.ldx;

    cs_f4 acc0[MI][NI], acc1[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) { acc0[i][j] = (cs_f4){0.f, 0.f, 0.f, 0.f}; acc1[i][j] = (cs_f4){0.f, 0.f, 0.f, 0.f}; }
    const int co_base = m0 + wm * (MI * 16);
    const int tl_base = wn * (NI * 16);
    const int Mt = (p.M + 15) >> 4;
    // split-K: this workgroup owns ONE input slice and one group of taps (short sequences: a 1024 -> 256, K = 9 layer on 61 positions
    // is 64 workgroups of 288 K-steps otherwise); the partial sums meet in cs_reduce_kernel
    int sl_lo = 0, sl_hi = p.nslices, k_lo = 0, k_hi = p.ntaps;
    if (p.kparts > 1) {
        const int tpg = (p.ntaps + p.tap_groups - 1) / p.tap_groups;
        sl_lo = kp / p.tap_groups; sl_hi = sl_lo + 1;
        k_lo = (kp % p.tap_groups) * tpg; k_hi = min(p.ntaps, k_lo + tpg);
    }
    const int sb = k_lo * NC, nsteps = (k_hi - k_lo) * NC;
    // input slices of CI channels one after the other through the same LDS tiles; the accumulators stay in registers
    for (int sl = sl_lo; sl < sl_hi; ++sl) {
    const float* Xs = X + (size_t)sl * CI;
    const _Float16* WH = p.wh + (size_t)sl * p.wslice;
    const _Float16* WL = p.wl + (size_t)sl * p.wslice;
    if (sl > sl_lo) __syncthreads();                      // every wave is done with the previous slice's tiles
    // ---- stage: rows t0-P .. t0+NT-1+P, zero outside [0,T); split into hi / lo*2^11.  All of a lane's row chunks (R*CH/512 <= 10) are
    //      requested together, UNCONDITIONALLY from a clamped row: predicated loads wait for one another (r01h s_memtime accounting: 10 k
    //      cycles of staging per slice, eight dependent round trips, against 17 k cycles of MFMA loop) ----
    {
        constexpr int CS_U = 4;                                // 8 sixteen-byte requests per lane in flight (more spills: 256 VGPRs)
        const int n = R * CH;
        for (int e0 = tid; e0 < n; e0 += NTH * CS_U) {
            float4 va[CS_U], vc[CS_U];
            bool ok[CS_U];
#pragma unroll
            for (int u = 0; u < CS_U; ++u) {
                const int e = e0 + u * NTH, ec = e < n ? e : 0;
                const int row = ec / CH, ch = ec - row * CH;
                const int tg = t0 - P + row;
                ok[u] = e < n && tg >= 0 && tg < p.T;
                const float* src = Xs + (size_t)(ok[u] ? tg : 0) * p.ldx + ch * 8;
                va[u] = *reinterpret_cast<const float4*>(src);
                vc[u] = *reinterpret_cast<const float4*>(src + 4);
            }
#pragma unroll
            for (int u = 0; u < CS_U; ++u) {
                const int e = e0 + u * NTH;
                if (e < n) {
                    const int row = e / CH, ch = e - row * CH;
                    float f[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vc[u].x, vc[u].y, vc[u].z, vc[u].w};
                    if constexpr (CI == 256 && NT == 64 && MT == 256) {
                        if (p.ln_w) {                        // one tap, one slice (host-checked): every lane is here; the row's 32 chunks are 32 consecutive lanes
                            float s1 = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) s1 += f[i];
#pragma unroll
                            for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
                            const float mean = s1 * (1.f / CI);
                            float q = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) { const float d = f[i] - mean; q += d * d; }
#pragma unroll
                            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                            const float rstd = rsqrtf(q * (1.f / CI) + p.ln_eps);
                            const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + ch * 8), w1 = *reinterpret_cast<const float4*>(p.ln_w + ch * 8 + 4);
                            const float4 b0 = *reinterpret_cast<const float4*>(p.ln_b + ch * 8), b1 = *reinterpret_cast<const float4*>(p.ln_b + ch * 8 + 4);
                            const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w}, b8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * w8[i] + b8[i];
                        }
                    }
                    cs_h8 vh, vl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x = ok[u] ? f[i] : 0.f;
                        vh[i] = (_Float16)x; vl[i] = (_Float16)((x - (float)vh[i]) * 2048.f);
                    }
                    const size_t o = ((size_t)row * CH + cs_swz<CI>(row, ch)) * 16;
                    *reinterpret_cast<cs_h8*>(th + o) = vh;
                    *reinterpret_cast<cs_h8*>(tl + o) = vl;
                }
            }
        }
    }
    __syncthreads();

    auto load_a = [&](int step, cs_h8 (&ah)[MI], cs_h8 (&al)[MI]) {
        const size_t off = (size_t)step * Mt * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int tile = (co_base >> 4) + i;
            const size_t o = off + (size_t)(tile < Mt ? tile : 0) * 512;
            ah[i] = *reinterpret_cast<const cs_h8*>(WH + o);
            al[i] = *reinterpret_cast<const cs_h8*>(WL + o);
        }
    };
    auto do_step = [&](int step, const cs_h8 (&ah)[MI], const cs_h8 (&al)[MI]) {
        const int k = step / NC, c = step - k * NC;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = tl_base + j * 16 + lr + k;
            const size_t o = ((size_t)row * CH + cs_swz<CI>(row, c * 4 + lk)) * 16;
            const cs_h8 bh = *reinterpret_cast<const cs_h8*>(th + o);
            const cs_h8 bl = *reinterpret_cast<const cs_h8*>(tl + o);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc1[i][j], 0, 0, 0);
            }
        }
    };
    // 3-deep register ring of weight fragments: a request is two steps (>= 48 MFMAs per wave) ahead of its use
    cs_h8 ah0[MI], al0[MI], ah1[MI], al1[MI], ah2[MI], al2[MI];
    if (nsteps > 0) load_a(sb, ah0, al0);
    if (nsteps > 1) load_a(sb + 1, ah1, al1);
    for (int step = 0; step < nsteps; step += 3) {
        if (step + 2 < nsteps) load_a(sb + step + 2, ah2, al2);
        do_step(sb + step, ah0, al0);
        if (step + 1 < nsteps) {
            if (step + 3 < nsteps) load_a(sb + step + 3, ah0, al0);
            do_step(sb + step + 1, ah1, al1);
        }
        if (step + 2 < nsteps) {
            if (step + 4 < nsteps) load_a(sb + step + 4, ah1, al1);
            do_step(sb + step + 2, ah2, al2);
        }
    }

    }

    if (p.kparts > 1) {                                   // raw partial sums; bias, activation, residual: cs_reduce_kernel
        float* P = p.part + ((size_t)kp * p.B + b) * p.T * p.M;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int co = co_base + i * 16 + lk * 4;
            if (co >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int t = t0 + tl_base + j * 16 + lr;
                if (t >= p.T) continue;
                *reinterpret_cast<float4*>(P + (size_t)t * p.M + co) = make_float4(acc0[i][j][0] + acc1[i][j][0] * (1.f / 2048.f), acc0[i][j][1] + acc1[i][j][1] * (1.f / 2048.f),
                                                                                 acc0[i][j][2] + acc1[i][j][2] * (1.f / 2048.f), acc0[i][j][3] + acc1[i][j][3] * (1.f / 2048.f));
            }
        }
        return;
    }
    // ---- epilogue: D fragment = 4 consecutive output channels of one frame per lane -> one 16-byte fp32 store ----
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int co = co_base + i * 16 + lk * 4;
        if (co >= p.M) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = p.bias[co + e];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int t = t0 + tl_base + j * 16 + lr;
            if (t >= p.T) continue;
            float* O = p.out + ((size_t)b * p.T + t) * p.ldo + co;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc0[i][j][e] + acc1[i][j][e] * (1.f / 2048.f)) + bv[e];
            if (p.accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(O);
                v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
            }
            if (p.relu == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.relu == 2) {                  // SiLU
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.f + __expf(-v[e]));
            } else if (p.relu == 3) {                  // GELU (erf form, torch's default)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
            }
            if (p.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.res + ((size_t)b * p.T + t) * p.ldr + co);
                v[0] = r4.x + p.alpha * v[0]; v[1] = r4.y + p.alpha * v[1]; v[2] = r4.z + p.alpha * v[2]; v[3] = r4.w + p.alpha * v[3];
            } else if (p.alpha != 1.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
            }
            *reinterpret_cast<float4*>(O) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// fp32 weight [ntaps][M][CI] (tap-major) -> hi / lo fp16 in fragment order [ntaps][CI/32][ceil(M/16)][64][8] (see hifigan_conv.hip)
__global__ void conv1d_split_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wh, _Float16* __restrict__ wl,
                                         int ntaps, int M, int CI)
{
    const int Mt = (M + 15) >> 4, NC = CI / 32;
    const long n = (long)ntaps * NC * Mt * 512;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e & 7), ln = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int tile = (int)(r % Mt); r /= Mt;
        const int c = (int)(r % NC); const int k = (int)(r / NC);
        const int co = tile * 16 + (ln & 15), ci = c * 32 + (ln >> 4) * 8 + h;
        const float v = (co < M) ? w[((size_t)k * M + co) * CI + ci] : 0.f;
        const _Float16 hi = (_Float16)v;
        wh[e] = hi;
        wl[e] = (_Float16)((v - (float)hi) * 2048.f);
    }
}

// out = res + alpha * act(bias + part[0] + part[1] + ...): the K-parts in a fixed order
__global__ __launch_bounds__(256) void cs_reduce_kernel(const float* __restrict__ part, int KP, long n, const float* __restrict__ bias, int M, int act,
                                                        const float* __restrict__ res, long ldr, float alpha, float* __restrict__ out, long ldo)
{
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long idx = e << 2, row = idx / M; const int c = (int)(idx - row * M);
        float4 s4 = *reinterpret_cast<const float4*>(part + idx);
        for (int k = 1; k < KP; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(part + (size_t)k * n + idx);
            s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w;
        }
        float v[4] = {s4.x, s4.y, s4.z, s4.w};
        if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + c); v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (act == 1) v[i] = fmaxf(v[i], 0.f);
            else if (act == 2) v[i] = v[i] / (1.f + __expf(-v[i]));
            else if (act == 3) v[i] = 0.5f * v[i] * (1.f + erff(v[i] * 0.70710678118654752f));
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4*>(res + row * ldr + c);
        *reinterpret_cast<float4*>(out + row * ldo + c) = make_float4(r.x + alpha * v[0], r.y + alpha * v[1], r.z + alpha * v[2], r.w + alpha * v[3]);
    }
}

template <int CI, int MT, int NT, int WM, int WN>
static int cs_launch(const CsParams& p, hipStream_t st)
{
    const size_t lds = (size_t)2 * (NT + p.ntaps - 1) * CI * 2;
    if (lds > 160 * 1024) { set_error("conv1d_split: tiles need %zu bytes of LDS", lds); return DSP_EINVAL; }
    auto k = conv1d_split_kernel<CI, MT, NT, WM, WN>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((p.T + NT - 1) / NT, ((p.M + MT - 1) / MT) * (p.kparts > 1 ? p.kparts : 1), p.B), dim3(WM * WN * 64), lds, st, p);
    return check_launch("conv1d_split");
}

}  // namespace dsp

using namespace dsp;

extern "C" long dsp_conv1d_split_packed_elems(int ntaps, int M, int CI)
{
    if (ntaps < 1 || M < 1 || CI < 32 || (CI & 31)) return -1;
    return (long)ntaps * (CI / 32) * ((M + 15) / 16) * 512;
}

extern "C" int dsp_conv1d_split_pack(const float* w_tap_major, void* w_hi, void* w_lo, int ntaps, int M, int CI, dsp_stream_t stream)
{
    const long n = dsp_conv1d_split_packed_elems(ntaps, M, CI);
    if (n < 0 || !w_tap_major || !w_hi || !w_lo) { set_error("conv1d_split_pack: bad arguments"); return DSP_EINVAL; }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv1d_split_pack_kernel, dim3(grid), dim3(256), 0, as_stream(stream), w_tap_major, (_Float16*)w_hi, (_Float16*)w_lo, ntaps, M, CI);
    return check_launch("conv1d_split_pack");
}

static int cs_run(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, float* out, long ldo,
                  int B, int T, int CI, int nslices, int M, int ntaps, int relu, int accumulate, const float* res, long ldr, float alpha,
                  dsp_stream_t stream, const int* lens = nullptr, int slack = 0, int tap_groups = 0, float* part = nullptr,
                  const float* ln_w = nullptr, const float* ln_b = nullptr, float ln_eps = 0.f)
{
    if (B < 0 || T < 1 || M < 4 || (M & 3) || ntaps < 1 || !(ntaps & 1) || ntaps > 31 || nslices < 1 || ldx < (long)CI * nslices || ldo < M ||
        (ldx & 3) || (ldo & 3)) {
        set_error("conv1d_split: bad sizes B=%d T=%d CI=%d M=%d taps=%d", B, T, CI, M, ntaps); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w_hi || !w_lo || !out) { set_error("conv1d_split: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)x) | ((uintptr_t)out)) & 15) { set_error("conv1d_split: x / out must be 16-byte aligned"); return DSP_EINVAL; }
    CsParams p;
    p.x = x; p.wh = (const _Float16*)w_hi; p.wl = (const _Float16*)w_lo; p.bias = bias; p.out = out;
    p.B = B; p.T = T; p.M = M; p.ntaps = ntaps; p.ldx = ldx; p.ldo = ldo; p.relu = relu; p.accumulate = accumulate;
    p.nslices = nslices; p.wslice = dsp_conv1d_split_packed_elems(ntaps, M, CI);
    p.res = res; p.ldr = ldr; p.alpha = alpha; p.lens = lens; p.slack = slack;
    p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps;
    if (ln_w && (!ln_b || CI != 256 || nslices != 1 || ntaps != 1 || tap_groups > 0 || (((uintptr_t)ln_w | (uintptr_t)ln_b) & 15))) {
        set_error("conv1d_split: the staged LayerNorm needs a one-tap layer over exactly 256 input channels"); return DSP_EINVAL; }
    p.kparts = tap_groups > 0 ? nslices * tap_groups : 1; p.tap_groups = tap_groups > 0 ? tap_groups : 1; p.part = part;
    const long kmul = p.kparts;                              // workgroups per output tile
    if (res && (((uintptr_t)res & 15) || ldr < M || (ldr & 3))) { set_error("conv1d_split: residual must be 16-byte aligned with row stride >= M"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (ln_w) return cs_launch<256, 256, 64, 8, 1>(p, st);      // the instance that carries the staged LayerNorm
    switch (CI) {
        case 256: {
            // 128-row tiles (8 time sub-tiles per wave, one workgroup per CU) when they fill the chip, 64-row tiles (two per CU) for the
            // narrow layers: 256 -> 256 projections on 12.6 k rows are 99 workgroups at 128 rows
            const long wgs128 = (long)((T + 127) / 128) * ((M + 255) / 256) * B * kmul;
            if (wgs128 >= 256) return cs_launch<256, 256, 128, 8, 1>(p, st);
            // under-filled launches (the Conformer's 256 -> 256 projections on 4.4 k positions: 69 workgroups of 256 output channels) take
            // 128-channel output tiles: twice the workgroups at 124 VGPRs, two per CU (the 256-channel tile needs 188: one per CU).
            // Acoustic stage 14.60 -> 14.25 ms; applied to every 64-row launch: 14.47 (the input tile is staged twice)
            const long wgs64 = (long)((T + 63) / 64) * ((M + 255) / 256) * B * kmul;
            if (wgs64 < 150) return cs_launch<256, 128, 64, 8, 1>(p, st);
            return cs_launch<256, 256, 64, 8, 1>(p, st);
        }
        case 512: {
            // (r03: a 512-wide Linear layer run as two 256-wide slices — 65 KB tiles, two workgroups per CU — is no faster: 14.59 vs 14.55 ms of
            //  GPU time per acoustic batch.)
            // launches that would leave CUs idle with 64-frame tiles (the Conformer's 2048 -> 256 on 4.4 k positions: 69 workgroups)
            // take 32-frame tiles, two workgroups per CU: 52.7 -> 36.7 us; where the 64-frame tiles fill the chip they are 10-20 % faster
            // (at 192 workgroups the 64-frame tiles still win: 34.3 vs 38.9 us for 1024 -> 256 on 10.6 k positions; 128-channel tiles: no gain)
            const long wgs64 = (long)((T + 63) / 64) * ((M + 255) / 256) * B * kmul;
            if (ntaps <= 9 && wgs64 < 128) return cs_launch<512, 256, 32, 8, 1>(p, st);
            // r05: one-tap, one-slice layers (the NAT decoder's 512 -> 512 / 1536 / 2048 projections) as 16-wave workgroups, 16 output channels per
            // wave: four waves per SIMD under the one resident workgroup (122 VGPRs), bit-identical results, 5-8 % faster (tools/cs_var_bench.py:
            // 39.8 -> 36.7, 98.7 -> 91.4, 116.6 -> 108.6 us; 512 output channels per workgroup: 2.2 x slower (spills), 8 x 2 waves: +18 %;
            // multi-slice and K = 9 layers: no difference; the 256-wide instances: no gain at 64 rows, +20 % at 128 rows)
            if (ntaps == 1 && nslices == 1 && kmul == 1) return cs_launch<512, 256, 64, 16, 1>(p, st);
            return cs_launch<512, 256, 64, 8, 1>(p, st);
        }
        case 128: return cs_launch<128, 128, 256, 4, 2>(p, st);
    }
    set_error("conv1d_split: unsupported slice width %d (128, 256, 512; wider inputs are nslices slices)", CI);
    return DSP_EINVAL;
}

extern "C" int dsp_conv1d_split(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, float* out, long ldo,
                                int B, int T, int CI, int nslices, int M, int ntaps, int relu, int accumulate, dsp_stream_t stream)
{
    return cs_run(x, ldx, w_hi, w_lo, bias, out, ldo, B, T, CI, nslices, M, ntaps, relu, accumulate, nullptr, 0, 1.f, stream);
}

extern "C" int dsp_conv1d_split_residual(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res,
                                         long ldr, float alpha, float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps,
                                         int relu, dsp_stream_t stream)
{
    return cs_run(x, ldx, w_hi, w_lo, bias, out, ldo, B, T, CI, nslices, M, ntaps, relu, 0, res, ldr, alpha, stream);
}

extern "C" int dsp_conv1d_split_ragged(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res,
                                       long ldr, float alpha, float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps,
                                       int relu, const int* lens, int slack, dsp_stream_t stream)
{
    if (lens && slack < 0) { set_error("conv1d_split_ragged: negative slack"); return DSP_EINVAL; }
    return cs_run(x, ldx, w_hi, w_lo, bias, out, ldo, B, T, CI, nslices, M, ntaps, relu, 0, res, ldr, alpha, stream, lens, slack);
}

extern "C" size_t dsp_conv1d_split_ksplit_workspace_bytes(int B, int T, int M, int nslices, int tap_groups)
{
    if (B < 1 || T < 1 || M < 4 || nslices < 1 || tap_groups < 1) return 0;
    return (size_t)nslices * tap_groups * B * T * M * sizeof(float);
}

extern "C" int dsp_conv1d_split_ksplit(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res, long ldr,
                                       float alpha, float* out, long ldo, int B, int T, int CI, int nslices, int M, int ntaps, int act,
                                       int tap_groups, void* workspace, size_t workspace_bytes, const int* lens, int slack, dsp_stream_t stream)
{
    if (tap_groups < 1 || tap_groups > ntaps || nslices * tap_groups < 2 || nslices * tap_groups > 64) {
        set_error("conv1d_split_ksplit: tap_groups=%d with %d slices and %d taps", tap_groups, nslices, ntaps); return DSP_EINVAL; }
    const size_t need = dsp_conv1d_split_ksplit_workspace_bytes(B, T, M, nslices, tap_groups);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) { set_error("conv1d_split_ksplit: workspace of %zu bytes, %zu needed (16-byte aligned)", workspace_bytes, need); return DSP_EINVAL; }
    if (act < 0 || act > 3 || (M & 3)) { set_error("conv1d_split_ksplit: bad activation / width"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (lens && slack < 0) { set_error("conv1d_split_ksplit: negative slack"); return DSP_EINVAL; }
    int rc = cs_run(x, ldx, w_hi, w_lo, nullptr, out, ldo, B, T, CI, nslices, M, ntaps, 0, 0, nullptr, 0, 1.f, stream, lens, slack, tap_groups, (float*)workspace);
    if (rc != DSP_OK) return rc;
    if (res && (((uintptr_t)res & 15) || ldr < M || (ldr & 3))) { set_error("conv1d_split_ksplit: residual must be 16-byte aligned with row stride >= M"); return DSP_EINVAL; }
    const long n = (long)B * T * M;
    int grid = (int)((n / 4 + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cs_reduce_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)workspace, nslices * tap_groups, n, bias, M, act, res, ldr, alpha, out, ldo);
    return check_launch("conv1d_split_ksplit(reduce)");
}

extern "C" int dsp_linear_ln_split(const float* x, long ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* w_hi, const void* w_lo,
                                   const float* bias, const float* res, long ldr, float alpha, float* out, long ldo, int B, int T, int M, int act,
                                   const int* lens, int slack, dsp_stream_t stream)
{
    if (!ln_w || !ln_b) { set_error("linear_ln_split: LayerNorm weight and bias are required"); return DSP_EINVAL; }
    if (lens && slack < 0) { set_error("linear_ln_split: negative slack"); return DSP_EINVAL; }
    return cs_run(x, ldx, w_hi, w_lo, bias, out, ldo, B, T, 256, 1, M, 1, act, 0, res, ldr, alpha, stream, lens, slack, 0, nullptr, ln_w, ln_b, ln_eps);
}
