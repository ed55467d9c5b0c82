// conv1d_stream.hip — fp32-accurate Conv1d / Linear on the fp16 matrix cores (the "3 x fp16" operand split of conv1d_split.hip), as a
// K-STREAMING kernel for the layers of the acoustic model (fairseq/models/text_to_speech/fastspeech2.py:42-95 FFT convolutions,
// modules/transformer_layer.py / conformer_layer.py Linear layers — everything decode_ops.linear / SplitConv1d serves at inference).
//
// conv1d_split.hip stages a whole [time tile] x [<= 512 input channels] slice, splits it, and only then starts its matrix-core loop: at 512
// channels that is 128 KB of LDS (one workgroup per CU) and a staging phase as long as the loop it precedes — 25-30 % of the matrix-core
// rate on the acoustic stage (profiles/r03g_acoustic_stage_kernels.txt).  Here the input streams through LDS in 64-CHANNEL CHUNKS, double
// buffered: chunk c+1 is requested from HBM / L2 into registers before the matrix-core loop of chunk c and is split (hi, lo x 2^11) into the
// other LDS buffer after it — one barrier per chunk, 70 KB of LDS for any channel count, two workgroups per CU, loads under the MFMAs.
//   * v_mfma_f32_32x32x16_f16: 32 flops per operand byte where the 16x16x32 form has 16 — half the LDS reads and weight loads per flop;
//   * a workgroup = 8 waves as 4 (output channels) x 2 (time): 128 output channels x 128 (or 64) time rows, accumulators 64 (32) VGPRs;
//   * weights pre-split in fragment order [tap][CI/16][ceil(M/32)][64 lanes][8] (dsp_conv1d_stream_pack), read straight from L2 into registers
//     one step ahead; activations from LDS with an XOR swizzle that is conflict-free for ds_read_b128's lane groups at any tap shift;
//   * workgroups are numbered so that all output-channel tiles of one time tile run back to back on ONE XCD (the input tile and the layer's
//     weights stay in that XCD's L2).
// Arithmetic as conv1d_split.hip: x = xh + xl 2^-11, w = wh + wl 2^-11, acc0 += wh.xh, acc1 += wh.xl + wl.xh, out = acc0 + acc1 2^-11.
#include "common.h"
#include <stdlib.h>
#include "../../include/daspeech_decode.h"

namespace dsp {

typedef _Float16 s3_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 s3_h4 __attribute__((ext_vector_type(4)));
typedef float s3_f16 __attribute__((ext_vector_type(16)));

struct S3Params {
    const float* x; const _Float16* wh; const _Float16* wl; const float* bias; float* out; const float* res;
    int B, T, M, CI, ntaps; long ldx, ldo, ldr;
    int act;                       // 0 none, 1 ReLU, 2 SiLU, 3 GELU(erf)
    float alpha;                   // out = res + alpha * act(bias + conv)   (res may be NULL)
    int tiles_t, tiles_m;          // time tiles per sample, output-channel tiles
};

constexpr int S3_KC = 64;          // input channels per chunk

template <int NJ>                  // 32-row time tiles per wave: NT = 64 * NJ
__global__ __launch_bounds__(512, 2) void conv1d_stream_kernel(S3Params p)
{
    extern __shared__ __attribute__((aligned(16))) char s3_smem[];
    constexpr int NT = 64 * NJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 3, wn = wave >> 2;
    const int ln = lane & 31, lg = lane >> 5;
    // workgroup w: XCD = w & 7 (round-robin dispatch); an XCD walks its (time tile, channel tile) pairs with the channel tile fastest
    const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
    const int mt = slot % p.tiles_m;
    const long tt = (long)(slot / p.tiles_m) * 8 + xcd;                   // global time-tile index over all samples
    if (tt >= (long)p.B * p.tiles_t) return;
    const int b = (int)(tt / p.tiles_t), t0 = (int)(tt - (long)b * p.tiles_t) * NT, m0 = mt * 128;
    const int P = (p.ntaps - 1) >> 1, R = NT + p.ntaps - 1;
    const int plane = R * 128;                                            // bytes of one (hi or lo) plane of one buffer: R rows x 64 halves
    const float* X = p.x + (size_t)b * p.T * p.ldx;
    const int nch = p.CI / S3_KC;

    // ---- chunk staging: rows t0-P .. t0+NT-1+P, 64 channels = 16 float4 per row; a lane takes quads e = tid + 512 u ----
    constexpr int U = 5;                                                  // R <= 160 rows (ntaps <= 33 at NT = 128)
    float4 pre[U];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 512, row = e >> 4, q = e & 15;
            const int tg = t0 - P + row;
            const bool ok = row < R && tg >= 0 && tg < p.T;
            const float* src = X + (size_t)(ok ? tg : 0) * p.ldx + c * S3_KC + q * 4;     // unconditional load from a clamped row
            float4 v = *reinterpret_cast<const float4*>(src);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            pre[u] = v;
        }
    };
    auto stash = [&](int buf) {
        char* hi = s3_smem + (size_t)buf * 2 * plane;
        char* lo = hi + plane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = tid + u * 512, row = e >> 4, q = e & 15;
            if (row < R) {
                const float f[4] = {pre[u].x, pre[u].y, pre[u].z, pre[u].w};
                s3_h4 vh, vl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { vh[i] = (_Float16)f[i]; vl[i] = (_Float16)((f[i] - (float)vh[i]) * 2048.f); }
                // 16-byte slot (q >> 1) of the row, XOR-swizzled by (row >> 1) & 7: a ds_read_b128 lane group ({0-3,12-15,20-27}, ...: 8 even and
                // 8 odd rows at one k-chunk) then covers all 16 slots of the 256-byte bank row, whatever the tap shift (MI355X_MICROARCH.md §LDS)
                const int o = row * 128 + ((((q >> 1) ^ ((row >> 1) & 7))) << 4) + ((q & 1) << 3);
                *reinterpret_cast<s3_h4*>(hi + o) = vh;
                *reinterpret_cast<s3_h4*>(lo + o) = vl;
            }
        }
    };

    s3_f16 accm[NJ], accc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) { accm[j][v] = 0.f; accc[j][v] = 0.f; }

    // ---- weights: fragment (tap, k16, tile) = 64 lanes x 8 halves, contiguous; the wave's tile is (m0 >> 5) + wm (clamped: zero-padded rows beyond M
    //      exist in the packed buffer up to a multiple of 32, tiles beyond that are never stored) ----
    const int Mt = (p.M + 31) >> 5, K16 = p.CI >> 4;
    const int mtile = min((m0 >> 5) + wm, Mt - 1);
    const size_t wbase = (size_t)mtile * 512 + lane * 8;
    auto load_a = [&](int tap, int kg, s3_h8& ah, s3_h8& al) {           // kg: global k16 index (chunk * 4 + k16)
        const size_t o = ((size_t)tap * K16 + kg) * Mt * 512 + wbase;
        ah = *reinterpret_cast<const s3_h8*>(p.wh + o);
        al = *reinterpret_cast<const s3_h8*>(p.wl + o);
    };

    fetch(0);
    stash(0);
    __syncthreads();
    // four-deep weight ring over the GLOBAL step sequence g = c * nsteps + s (a fragment is requested three steps before its use: loads return
    // in order, so a wait for a weight fragment also waits for every older request — the next chunk's input rows included; with three steps
    // of MFMAs between request and use the input rows have landed by then)
    const int nsteps = p.ntaps * 4;                                       // (tap, k16) pairs of a chunk, k16 fastest
    const int gtot = nch * nsteps;
    s3_h8 ah[4], al[4];
    auto load_g = [&](int g, s3_h8& h, s3_h8& l) {
        if (g < gtot) { const int cc = g / nsteps, ss = g - cc * nsteps; load_a(ss >> 2, cc * 4 + (ss & 3), h, l); }
    };
    load_g(0, ah[0], al[0]); load_g(1, ah[1], al[1]); load_g(2, ah[2], al[2]);
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) fetch(c + 1);                                    // in flight during this chunk's matrix-core loop
        const char* hi = s3_smem + (size_t)(c & 1) * 2 * plane;
        const char* lo = hi + plane;
        auto do_step = [&](int s, const s3_h8& wh8, const s3_h8& wl8) {
            const int tap = s >> 2, k16 = s & 3;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = (wn * NJ + j) * 32 + ln + tap;
                const int o = row * 128 + (((k16 * 2 + lg) ^ ((row >> 1) & 7)) << 4);
                const s3_h8 bh = *reinterpret_cast<const s3_h8*>(hi + o);
                const s3_h8 bl = *reinterpret_cast<const s3_h8*>(lo + o);
                accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh8, bh, accm[j], 0, 0, 0);
                accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh8, bl, accc[j], 0, 0, 0);
                accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl8, bh, accc[j], 0, 0, 0);
            }
        };
        const int g0 = c * nsteps;
        for (int s = 0; s < nsteps; s += 4) {                             // nsteps is a multiple of 4: ring slot = step & 3, static names
            load_g(g0 + s + 3, ah[3], al[3]); do_step(s, ah[0], al[0]);
            load_g(g0 + s + 4, ah[0], al[0]); do_step(s + 1, ah[1], al[1]);
            load_g(g0 + s + 5, ah[1], al[1]); do_step(s + 2, ah[2], al[2]);
            load_g(g0 + s + 6, ah[2], al[2]); do_step(s + 3, ah[3], al[3]);
        }
        if (c + 1 < nch) stash((c + 1) & 1);                              // the other buffer: its last readers passed the barrier below one chunk ago
        __syncthreads();
    }

    // ---- epilogue: lane (ln = time row, lg) holds out channels (v >> 2) * 8 + lg * 4 + (v & 3): four consecutive channels per 16-byte store ----
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = t0 + (wn * NJ + j) * 32 + ln;
        if (t >= p.T) continue;
        const size_t rowo = (size_t)b * p.T + t;
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int co = m0 + wm * 32 + v4 * 8 + lg * 4;
            if (co >= p.M) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = accm[j][v4 * 4 + e] + accc[j][v4 * 4 + e] * (1.f / 2048.f);
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.f + __expf(-v[e]));
            } else if (p.act == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
            }
            if (p.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.res + rowo * p.ldr + co);
                v[0] = r4.x + p.alpha * v[0]; v[1] = r4.y + p.alpha * v[1]; v[2] = r4.z + p.alpha * v[2]; v[3] = r4.w + p.alpha * v[3];
            } else if (p.alpha != 1.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
            }
            *reinterpret_cast<float4*>(p.out + rowo * p.ldo + co) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// fp32 weight [ntaps][M][CI] (tap-major) -> hi / lo fp16 in the 32x32x16 A-fragment order [ntaps][CI/16][ceil(M/32)][64][8]:
// lane = (m & 31) + 32 * kk holds W[tile * 32 + (lane & 31)][k16 * 16 + (lane >> 5) * 8 + 0..7]; rows beyond M are zero
__global__ void conv1d_stream_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wh, _Float16* __restrict__ wl, int ntaps, int M, int CI)
{
    const int Mt = (M + 31) >> 5, K16 = CI >> 4;
    const long n = (long)ntaps * K16 * Mt * 512;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int h = (int)(e & 7), ln = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int tile = (int)(r % Mt); r /= Mt;
        const int kg = (int)(r % K16); const int k = (int)(r / K16);
        const int co = tile * 32 + (ln & 31), ci = kg * 16 + (ln >> 5) * 8 + h;
        const float v = (co < M) ? w[((size_t)k * M + co) * CI + ci] : 0.f;
        const _Float16 hi = (_Float16)v;
        wh[e] = hi;
        wl[e] = (_Float16)((v - (float)hi) * 2048.f);
    }
}

template <int NJ>
static int s3_launch(S3Params& p, hipStream_t st)
{
    constexpr int NT = 64 * NJ;
    p.tiles_t = (p.T + NT - 1) / NT; p.tiles_m = (p.M + 127) / 128;
    const long pairs = ((long)p.B * p.tiles_t + 7) / 8 * 8 * p.tiles_m;   // every XCD gets whole time tiles
    const size_t lds = (size_t)2 * 2 * (NT + p.ntaps - 1) * 128;
    if (lds > 160 * 1024 || pairs > 0x7fffffffL) { set_error("conv1d_stream: %zu bytes of LDS / %ld workgroups", lds, pairs); return DSP_EINVAL; }
    auto k = conv1d_stream_kernel<NJ>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)pairs), dim3(512), lds, st, p);
    return check_launch("conv1d_stream");
}

}  // namespace dsp

using namespace dsp;

extern "C" long dsp_conv1d_stream_packed_elems(int ntaps, int M, int CI)
{
    if (ntaps < 1 || M < 1 || CI < 64 || (CI & 63)) return -1;
    return (long)ntaps * (CI / 16) * ((M + 31) / 32) * 512;
}

extern "C" int dsp_conv1d_stream_pack(const float* w_tap_major, void* w_hi, void* w_lo, int ntaps, int M, int CI, dsp_stream_t stream)
{
    const long n = dsp_conv1d_stream_packed_elems(ntaps, M, CI);
    if (n < 0 || !w_tap_major || !w_hi || !w_lo) { set_error("conv1d_stream_pack: bad arguments"); return DSP_EINVAL; }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv1d_stream_pack_kernel, dim3(grid), dim3(256), 0, as_stream(stream), w_tap_major, (_Float16*)w_hi, (_Float16*)w_lo, ntaps, M, CI);
    return check_launch("conv1d_stream_pack");
}

extern "C" int dsp_conv1d_stream(const float* x, long ldx, const void* w_hi, const void* w_lo, const float* bias, const float* res, long ldr,
                                 float alpha, float* out, long ldo, int B, int T, int CI, int M, int ntaps, int act, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || M < 4 || (M & 3) || ntaps < 1 || !(ntaps & 1) || ntaps > 31 || CI < 64 || (CI & 63) || ldx < CI || ldo < M || (ldx & 3) || (ldo & 3)) {
        set_error("conv1d_stream: bad sizes B=%d T=%d CI=%d M=%d taps=%d", B, T, CI, M, ntaps); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!x || !w_hi || !w_lo || !out) { set_error("conv1d_stream: null pointer"); return DSP_EINVAL; }
    if ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)bias)) & 15) { set_error("conv1d_stream: x / out / bias must be 16-byte aligned"); return DSP_EINVAL; }
    if (res && (((uintptr_t)res & 15) || ldr < M || (ldr & 3))) { set_error("conv1d_stream: residual must be 16-byte aligned with row stride >= M"); return DSP_EINVAL; }
    S3Params p;
    p.x = x; p.wh = (const _Float16*)w_hi; p.wl = (const _Float16*)w_lo; p.bias = bias; p.out = out; p.res = res;
    p.B = B; p.T = T; p.M = M; p.CI = CI; p.ntaps = ntaps; p.ldx = ldx; p.ldo = ldo; p.ldr = ldr; p.act = act; p.alpha = alpha;
    // 128-row time tiles when they still give every CU a workgroup, 64-row tiles otherwise
    const long wg128 = (long)B * ((T + 127) / 128) * ((M + 127) / 128);
    if (wg128 >= 256) return s3_launch<2>(p, as_stream(stream));
    return s3_launch<1>(p, as_stream(stream));
}
