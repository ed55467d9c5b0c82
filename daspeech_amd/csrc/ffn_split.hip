// ffn_split.hip — the Conformer's position-wise feed-forward module in ONE matrix-core launch at fp32 accuracy (the "3 x fp16" operand
// split of conv1d_split.hip):   out = res + alpha * (W2 . act(W1 . LN(x) + b1) + b2)
// (fairseq/modules/conformer_layer.py:140-146 FeedForwardModule: layer_norm - w_1 - swish - w_2, called as x + 0.5 * ffn(x) by
// ConformerEncoderLayer.forward :254-281; the reference runs it as LayerNorm + two fp32 GEMMs + activation + scale + add).
//
// As two conv1d_split launches the module costs 102 us per call at B=32, T=197 (C=256, H=2048): the first GEMM's workgroups stage a
// 128-row tile, run 8 K-steps and leave through a 2048-wide epilogue (matrix cores busy 30 % of the time), the second has only 128
// workgroups of a 2048-deep reduction — and the [B,T,2048] intermediate makes a round trip through HBM.  Here a workgroup owns 64 rows
// and a GROUP of hidden channels: per 256-channel chunk of its group it forms h = act(W1[chunk] . x + b1) from the x tile it staged
// once (LayerNorm applied while staging: a row's 256 channels sit in half a wave), splits h into LDS and accumulates
// y += W2[:, chunk] . h in registers.  With few rows (6.3 k here: 25 per CU) the hidden dimension is what fills the chip: G groups
// write partial sums that ffn_reduce_kernel adds in a FIXED order with the bias and the residual (deterministic; no atomics).
// Weight traffic per workgroup is 1/G of the two matrices; the intermediate never leaves the CU.
#include "common.h"
#include "../../include/daspeech_decode.h"

namespace dsp {

typedef _Float16 ff_h8 __attribute__((ext_vector_type(8)));
typedef float ff_f4 __attribute__((ext_vector_type(4)));

struct FfParams {
    const float* x; long ldx; const float* ln_w; const float* ln_b; float ln_eps; int has_ln;
    const _Float16* w1h; const _Float16* w1l; const float* b1;        // packed [C/32 steps][H/16 tiles][64][8]   (dsp_conv1d_split_pack, 1 tap)
    const _Float16* w2h; const _Float16* w2l; long w2slice;           // packed per 512-channel input slice [16 steps][C/16 tiles][64][8]
    float* part;                                                       // [G][B][T][C] partial sums of W2 . h
    int B, T, H, act, G;
};

// 16-byte chunk swizzle of a [rows][256 halves] tile (conv1d_split.hip cs_swz<256>: 32 chunks per row)
__device__ __forceinline__ int ff_swz(int row, int chunk) { return chunk ^ ((row & 7) << 1); }

template <int C, int NT>
__global__ __launch_bounds__(512) void ffn_split_kernel(FfParams p)
{
    static_assert(C == 256 && NT == 64, "one instance: 256 channels in and out, 64-row tiles");
    extern __shared__ __attribute__((aligned(16))) char ff_smem[];
    constexpr int CH = C / 8, NS = C / 32, MI = 2, NI = NT / 16, HC = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lk = lane >> 4;
    const int t0 = blockIdx.x * NT, g = blockIdx.y, b = blockIdx.z;
    char* xh = ff_smem; char* xl = xh + NT * C * 2;
    char* hh = xl + NT * C * 2; char* hl = hh + NT * HC * 2;
    const float* X = p.x + (size_t)b * p.T * p.ldx;

    // ---- stage the x tile: LayerNorm per row (two passes over the row's registers), split into hi / lo * 2^11
    {
        constexpr int U = 4;                                   // 4 x 16 rows per pass, every request issued before the first use
        for (int e0 = tid; e0 < NT * CH; e0 += 512 * U) {
            float f[U][8]; bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * 512, row = e / CH, ch = e - row * CH, tg = t0 + row;
                ok[u] = tg < p.T;
                const float* src = X + (size_t)(ok[u] ? tg : 0) * p.ldx + ch * 8;
                *reinterpret_cast<float4*>(f[u]) = *reinterpret_cast<const float4*>(src);
                *reinterpret_cast<float4*>(f[u] + 4) = *reinterpret_cast<const float4*>(src + 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * 512, row = e / CH, ch = e - row * CH;
                if (p.has_ln) {                                // the row's 32 chunks are 32 consecutive lanes
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s += f[u][i];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    const float mean = s * (1.f / C);
                    float q = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = f[u][i] - mean; q += d * d; }
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
                    const float rstd = rsqrtf(q * (1.f / C) + p.ln_eps);
                    float w8[8], b8[8];
                    *reinterpret_cast<float4*>(w8) = *reinterpret_cast<const float4*>(p.ln_w + ch * 8);
                    *reinterpret_cast<float4*>(w8 + 4) = *reinterpret_cast<const float4*>(p.ln_w + ch * 8 + 4);
                    *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(p.ln_b + ch * 8);
                    *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(p.ln_b + ch * 8 + 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[u][i] = (f[u][i] - mean) * rstd * w8[i] + b8[i];
                }
                ff_h8 vh, vl;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = ok[u] ? f[u][i] : 0.f;
                    vh[i] = (_Float16)v; vl[i] = (_Float16)((v - (float)vh[i]) * 2048.f);
                }
                const size_t o = ((size_t)row * CH + ff_swz(row, ch)) * 16;
                *reinterpret_cast<ff_h8*>(xh + o) = vh;
                *reinterpret_cast<ff_h8*>(xl + o) = vl;
            }
        }
    }
    __syncthreads();

    ff_f4 y0[MI][NI], y1[MI][NI];                              // W2 . h, main and correction accumulators (persist over the chunks)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) { y0[i][j] = (ff_f4){0.f, 0.f, 0.f, 0.f}; y1[i][j] = (ff_f4){0.f, 0.f, 0.f, 0.f}; }

    const int Mt1 = p.H >> 4;                                  // 16-channel tiles of W1's packed rows
    constexpr int Mt2 = C / 16;
    const int nch = (p.H / HC) / p.G;                          // chunks of this group
    static_assert(NS == 8 && HC / 32 == 8, "both GEMMs run 8 K-steps per chunk");
    // Iteration c runs ONE stream of K-steps: the 8 steps of A(c): h = W1[chunk c] . x  (x tile, accumulators h0 / h1) followed by the 8
    // steps of B(c-1): y += W2[:, chunk c-1] . h(c-1)  (h tile in LDS, accumulators y0 / y1) — back to back through one 3-deep register
    // ring of weight fragments, so the ring fills once per chunk; then h(c) = act(h + b1) is split into the LDS tile (after a barrier:
    // B(c-1) was its last reader) with the first two fragments of the next stream already requested.  (As separate phases every GEMM
    // start waited a full L2 round trip: 74 us per call instead of ~50.)
    ff_f4 h0[MI][NI], h1[MI][NI];
    ff_h8 rh[3][MI], rl[3][MI];                                // the ring
    const size_t lane8 = (size_t)lane * 8;
    // fragment offsets (halves): W1 step k of chunk ch | W2 step k of chunk ch (a 512-channel input slice holds two chunks)
    auto off1 = [&](int ch, int k) -> size_t { return ((size_t)k * Mt1 + ch * 16 + wave * 2) * 512 + lane8; };
    auto off2 = [&](int ch, int k) -> size_t { return (size_t)(ch >> 1) * p.w2slice + ((size_t)((ch & 1) * 8 + k) * Mt2 + wave * 2) * 512 + lane8; };
    auto load_w = [&](const _Float16* WH, const _Float16* WL, size_t off, ff_h8 (&ah)[MI], ff_h8 (&al)[MI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            ah[i] = *reinterpret_cast<const ff_h8*>(WH + off + (size_t)i * 512);
            al[i] = *reinterpret_cast<const ff_h8*>(WL + off + (size_t)i * 512);
        }
    };
    auto do_step = [&](int s, const char* th, const char* tl, const ff_h8 (&ah)[MI], const ff_h8 (&al)[MI], ff_f4 (&a0)[MI][NI], ff_f4 (&a1)[MI][NI]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = j * 16 + lr;
            const size_t o = ((size_t)row * CH + ff_swz(row, s * 4 + lk)) * 16;
            const ff_h8 bh = *reinterpret_cast<const ff_h8*>(th + o);
            const ff_h8 bl = *reinterpret_cast<const ff_h8*>(tl + o);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                a0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, a0[i][j], 0, 0, 0);
                a1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, a1[i][j], 0, 0, 0);
                a1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, a1[i][j], 0, 0, 0);
            }
        }
    };
    // h(c) = act(h + b1) -> split -> LDS tile
    auto write_h = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int hcl = wave * 32 + i * 16 + lk * 4;       // channel inside the chunk
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.b1) *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(p.b1 + chunk * HC + hcl);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int m = j * 16 + lr;
                _Float16 hv[4], lv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = h0[i][j][e] + h1[i][j][e] * (1.f / 2048.f) + bv[e];
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    else if (p.act == 2) v = v / (1.f + __expf(-v));
                    else if (p.act == 3) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                    const _Float16 q = (_Float16)v;
                    hv[e] = q; lv[e] = (_Float16)((v - (float)q) * 2048.f);
                }
                const size_t o = ((size_t)m * CH + ff_swz(m, hcl >> 3)) * 16 + (hcl & 4) * 2;
                *reinterpret_cast<uint2*>(hh + o) = *reinterpret_cast<uint2*>(hv);
                *reinterpret_cast<uint2*>(hl + o) = *reinterpret_cast<uint2*>(lv);
            }
        }
    };
    auto zero_h = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) { h0[i][j] = (ff_f4){0.f, 0.f, 0.f, 0.f}; h1[i][j] = (ff_f4){0.f, 0.f, 0.f, 0.f}; }
    };
    const int ch0 = g * nch;
    // ---- first stream: A(0) alone
    zero_h();
    load_w(p.w1h, p.w1l, off1(ch0, 0), rh[0], rl[0]); load_w(p.w1h, p.w1l, off1(ch0, 1), rh[1], rl[1]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k + 2 < 8) load_w(p.w1h, p.w1l, off1(ch0, k + 2), rh[(k + 2) % 3], rl[(k + 2) % 3]);
        do_step(k, xh, xl, rh[k % 3], rl[k % 3], h0, h1);
    }
    // ---- steady state: write h(c-1), then A(c) and B(c-1) as one 16-step stream
    for (int c = 1; c < nch; ++c) {
        const int ch = ch0 + c;
        load_w(p.w1h, p.w1l, off1(ch, 0), rh[0], rl[0]); load_w(p.w1h, p.w1l, off1(ch, 1), rh[1], rl[1]);     // in flight under write_h
        if (c > 1) __syncthreads();                            // every wave is done reading h(c-2)
        write_h(ch - 1);
        zero_h();
        __syncthreads();                                       // h(c-1) complete
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k + 2 < 8) load_w(p.w1h, p.w1l, off1(ch, k + 2), rh[(k + 2) % 3], rl[(k + 2) % 3]);
            else if (k + 2 < 16) load_w(p.w2h, p.w2l, off2(ch - 1, k + 2 - 8), rh[(k + 2) % 3], rl[(k + 2) % 3]);
            if (k < 8) do_step(k, xh, xl, rh[k % 3], rl[k % 3], h0, h1);
            else do_step(k - 8, hh, hl, rh[k % 3], rl[k % 3], y0, y1);
        }
    }
    // ---- last stream: write h(nch-1), B(nch-1) alone
    {
        const int ch = ch0 + nch - 1;
        load_w(p.w2h, p.w2l, off2(ch, 0), rh[0], rl[0]); load_w(p.w2h, p.w2l, off2(ch, 1), rh[1], rl[1]);
        if (nch > 1) __syncthreads();
        write_h(ch);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k + 2 < 8) load_w(p.w2h, p.w2l, off2(ch, k + 2), rh[(k + 2) % 3], rl[(k + 2) % 3]);
            do_step(k, hh, hl, rh[k % 3], rl[k % 3], y0, y1);
        }
    }
    // ---- partial sums out (bias, scale and residual belong to ffn_reduce_kernel)
    float* P = p.part + ((size_t)g * p.B + b) * p.T * C;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int co = wave * 32 + i * 16 + lk * 4;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int t = t0 + j * 16 + lr;
            if (t >= p.T) continue;
            float4 v;
            v.x = y0[i][j][0] + y1[i][j][0] * (1.f / 2048.f); v.y = y0[i][j][1] + y1[i][j][1] * (1.f / 2048.f);
            v.z = y0[i][j][2] + y1[i][j][2] * (1.f / 2048.f); v.w = y0[i][j][3] + y1[i][j][3] * (1.f / 2048.f);
            *reinterpret_cast<float4*>(P + (size_t)t * C + co) = v;
        }
    }
}

// out = res + alpha * ((part[0] + part[1] + ...) + b2): the groups in a fixed order
__global__ __launch_bounds__(256) void ffn_reduce_kernel(const float* __restrict__ part, int G, long n, const float* __restrict__ b2, int C,
                                                         const float* __restrict__ res, long ldr, float alpha, float* __restrict__ out, long ldo)
{
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long idx = e << 2, row = idx / C; const int c = (int)(idx - row * C);
        float4 s = *reinterpret_cast<const float4*>(part + idx);
        for (int k = 1; k < G; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(part + (size_t)k * n + idx);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        if (b2) { const float4 bb = *reinterpret_cast<const float4*>(b2 + c); s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w; }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4*>(res + row * ldr + c);
        *reinterpret_cast<float4*>(out + row * ldo + c) = make_float4(r.x + alpha * s.x, r.y + alpha * s.y, r.z + alpha * s.z, r.w + alpha * s.w);
    }
}

// the same with LayerNorm(out) as a second (or the only) result: one wave per row of 256 channels
__global__ __launch_bounds__(256) void ffn_reduce_ln_kernel(const float* __restrict__ part, int G, long rows, const float* __restrict__ b2,
                                                            const float* __restrict__ res, long ldr, float alpha, float* __restrict__ out, long ldo,
                                                            const float* __restrict__ lw, const float* __restrict__ lb, float eps, float* __restrict__ out_ln)
{
    constexpr int C = 256;
    const int lane = threadIdx.x & 63;
    const long n = rows * C;
    for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
        const long idx = row * C + lane * 4;
        float4 s = *reinterpret_cast<const float4*>(part + idx);
        for (int k = 1; k < G; ++k) {
            const float4 q = *reinterpret_cast<const float4*>(part + (size_t)k * n + idx);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        if (b2) { const float4 bb = *reinterpret_cast<const float4*>(b2 + lane * 4); s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w; }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = *reinterpret_cast<const float4*>(res + row * ldr + lane * 4);
        const float4 v = make_float4(r.x + alpha * s.x, r.y + alpha * s.y, r.z + alpha * s.z, r.w + alpha * s.w);
        if (out) *reinterpret_cast<float4*>(out + row * ldo + lane * 4) = v;
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / C) + eps);
        const float4 w4 = *reinterpret_cast<const float4*>(lw + lane * 4), b4 = *reinterpret_cast<const float4*>(lb + lane * 4);
        *reinterpret_cast<float4*>(out_ln + row * C + lane * 4) = make_float4(dx * rstd * w4.x + b4.x, dy * rstd * w4.y + b4.y, dz * rstd * w4.z + b4.z,
                                                                               dw * rstd * w4.w + b4.w);
    }
}

static int ff_groups(int B, int T, int H)
{
    const long tiles = (long)((T + 63) / 64) * B;
    int G = 1;
    while (tiles * G < 224 && G < 8 && (H / 256) % (2 * G) == 0) G *= 2;
    return G;
}

}  // namespace dsp

using namespace dsp;

extern "C" size_t dsp_ffn_split_workspace_bytes(int B, int T, int C, int H)
{
    if (B < 1 || T < 1 || C != 256 || H < 512 || (H & 511)) return 0;
    return (size_t)ff_groups(B, T, H) * B * T * C * sizeof(float);
}

extern "C" int dsp_ffn_split(const float* x, long ldx, const float* ln_w, const float* ln_b, float ln_eps, const void* w1_hi, const void* w1_lo,
                             const float* b1, const void* w2_hi, const void* w2_lo, const float* b2, const float* res, long ldr, float alpha,
                             float* out, long ldo, void* workspace, size_t workspace_bytes, int B, int T, int C, int H, int act,
                             const float* post_ln_w, const float* post_ln_b, float post_ln_eps, float* out_ln, dsp_stream_t stream)
{
    if (B < 0 || T < 1 || C != 256 || H < 512 || (H & 511) || act < 0 || act > 3 || ldx < C || ldo < C || (ldx & 3) || (ldo & 3)) {
        set_error("ffn_split: bad sizes B=%d T=%d C=%d H=%d (C = 256, H a multiple of 512)", B, T, C, H); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    const bool post = post_ln_w != nullptr;
    if (post != (post_ln_b != nullptr) || post != (out_ln != nullptr) || (post && ((((uintptr_t)post_ln_w) | ((uintptr_t)post_ln_b) | ((uintptr_t)out_ln)) & 15))) {
        set_error("ffn_split: the post-LayerNorm needs weight, bias and an output, 16-byte aligned"); return DSP_EINVAL; }
    if (!x || !w1_hi || !w1_lo || !w2_hi || !w2_lo || (!out && !post) || !workspace) { set_error("ffn_split: null pointer"); return DSP_EINVAL; }
    if ((ln_w == nullptr) != (ln_b == nullptr)) { set_error("ffn_split: LayerNorm needs weight and bias"); return DSP_EINVAL; }
    if (((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)workspace) | ((uintptr_t)res) | ((uintptr_t)ln_w) | ((uintptr_t)ln_b) | ((uintptr_t)b1) |
          ((uintptr_t)b2)) & 15) || (res && (ldr < C || (ldr & 3)))) {
        set_error("ffn_split: pointers must be 16-byte aligned, row strides >= C and %% 4 == 0"); return DSP_EINVAL; }
    const size_t need = dsp_ffn_split_workspace_bytes(B, T, C, H);
    if (workspace_bytes < need) { set_error("ffn_split: workspace of %zu bytes, %zu needed", workspace_bytes, need); return DSP_EINVAL; }
    FfParams p;
    p.x = x; p.ldx = ldx; p.ln_w = ln_w; p.ln_b = ln_b; p.ln_eps = ln_eps; p.has_ln = ln_w != nullptr;
    p.w1h = (const _Float16*)w1_hi; p.w1l = (const _Float16*)w1_lo; p.b1 = b1;
    p.w2h = (const _Float16*)w2_hi; p.w2l = (const _Float16*)w2_lo; p.w2slice = dsp_conv1d_split_packed_elems(1, C, 512);
    p.part = (float*)workspace; p.B = B; p.T = T; p.H = H; p.act = act; p.G = ff_groups(B, T, H);
    hipStream_t st = as_stream(stream);
    const size_t lds = (size_t)4 * 64 * 256 * 2;
    auto k = ffn_split_kernel<256, 64>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((T + 63) / 64, p.G, B), dim3(512), lds, st, p);
    int rc = check_launch("ffn_split");
    if (rc != DSP_OK) return rc;
    const long n = (long)B * T * C;
    if (post) {
        const long rows = (long)B * T;
        int grid = (int)((rows + 3) / 4); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(ffn_reduce_ln_kernel, dim3(grid), dim3(256), 0, st, (const float*)workspace, p.G, rows, b2, res, ldr, alpha, out, ldo, post_ln_w,
                           post_ln_b, post_ln_eps, out_ln);
        return check_launch("ffn_reduce_ln");
    }
    int grid = (int)((n / 4 + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(ffn_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)workspace, p.G, n, b2, C, res, ldr, alpha, out, ldo);
    return check_launch("ffn_reduce");
}
