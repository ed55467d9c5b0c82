// attention_split.hip — fp32-accurate multi-head attention on the fp16 matrix cores ("3 x fp16" split, see conv1d_split.hip), for the
// eval-mode NAT decoder and FastSpeech2 FFT blocks (fairseq/modules/multihead_attention.py:  softmax(q k^T / sqrt(dk) + key padding) v;
// reference call sites: DASpeech/models/nat_s2s/s2s_conformer_dag_fastspeech2.py decoder layers, fairseq fastspeech2.py:65-98 FFTLayer).
//
// torch serves these as fp32 flash attention at ~75 TFLOP/s (138 us per call at B=32, 8 heads x 400 x 400 x 64: 2.2 of the 14.7 ms of
// the acoustic stage).  Here every operand is split  x = xh + xl * 2^-11  (fp16 pairs) and both products run as three fp16 MFMAs with
// an fp32 accumulator (xh.yh | xh.yl + xl.yh, the lo terms in their own accumulator, folded with 2^-11): 2^-22 relative, fp32 accuracy.
//
// One wave owns 32 queries, a workgroup (4 waves) 128 queries of one (sample, head) and streams the keys in tiles of 32:
//   S^T = K . Q^T  (v_mfma_f32_32x32x16_f16, A = key rows, B = query columns): a lane then holds ONE query's scores for 16 of the
//         tile's 32 keys (the lane 32 away holds the other 16), so the online soft-max is lane-local plus one exchange, and
//   O^T = V^T . P^T  takes P^T straight from those registers as the B operand — the keys of a 16-wide contraction step are the ones
//         the accumulator layout hands a lane (key(c, g, e) = (2c + e/4) * 8 + 4g + e%4 for step c, lane half g, element e) and the
//         V^T fragments are written to LDS in that same order, so no transpose of P ever happens.
// K and V tiles are read from global fp32, split and laid out as MFMA A-fragments in LDS by all four waves (double-buffered, one
// barrier per tile); trailing key tiles that are padding for the whole sample are skipped (exact: their weights are 0).
//
// REL = true is the Conformer's relative-position attention (fairseq/modules/espnet_multihead_attention.py:172-254
// RelPositionMultiHeadedAttention):  score(i, j) = (q_i + u) . k_j + (q_i + v) . pos[T-1 - i + j].  The second term is formed per wave
// and key tile as G^T = pos[rbase .. rbase + 63] . (Q + v)^T (two 32-row blocks of the position projection, kept in an LDS ring that
// takes one new block per tile) and read back from a per-wave scratch at [query][key - query + 31]: the rel_shift of the torch
// formulation without its [B,h,T,2T-1] tensor.  (r03's fp32-FMA kernel for this: 113 us per layer at B=32, T=200; this one ~15.)
#include "common.h"
#include "../../include/daspeech_decode.h"

namespace dsp {

typedef _Float16 at_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_h4 __attribute__((ext_vector_type(4)));
typedef float at_f16 __attribute__((ext_vector_type(16)));

struct AtParams {
    const float* q; const float* k; const float* v; const unsigned char* kmask; float* out;
    long ldq, ldk, ldv, bq, bk, bv;      // row strides and sample strides (floats)
    int B, N, M, H, ntq; float scale;
    const float* pos; const float* bias_u; const float* bias_v;      // relative-position variant: pos [2T-1][H*dk], biases [H][dk]
    const int* qlens; int qslack;        // ragged batch: queries >= qlens[b] + qslack are padding nobody reads: zeros, not computed
};

constexpr int AT_QW = 32, AT_WAVES = 4, AT_QT = AT_QW * AT_WAVES, AT_KT = 32;
constexpr int AT_HALF = 32 * 16 + 64;              // bytes between the two lane halves of a fragment (the 64-byte pad keeps the 8-lane
constexpr int AT_FRAG = 2 * AT_HALF;               // store groups of ds_write_b128 on distinct banks); one fragment = 64 lanes x 16 bytes

template <int DK> struct AtLds {
    static constexpr int KF = (DK / 16) * 2 * AT_FRAG;             // K fragments: [dk/16 steps][hi, lo]
    static constexpr int VF = (DK / 32) * 2 * 2 * AT_FRAG;         // V^T fragments: [dk/32 row blocks][2 key steps][hi, lo]
    static constexpr int STAGE = KF + VF + AT_KT * 4;              // + additive key bias (0 / -inf)
    // relative-position variant: a ring of 32-row blocks of the position projection (K's fragment layout) and a [32][64] (+4) score
    // scratch per wave for the shift of the [query][relative position] product into [query][key]
    static constexpr int PRING = 6, PBLK = KF, SCR_PITCH = 68, SCR = 32 * SCR_PITCH * 4;
};

__device__ __forceinline__ void at_split8(const float* x, at_h8& hi, at_h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)x[e]; lo[e] = (_Float16)((x[e] - (float)hi[e]) * 2048.f); }
}

template <int DK, bool REL>
__global__ __launch_bounds__(256, (DK == 64 && !REL) ? 2 : 1) void attention_split_kernel(AtParams p)
{
    extern __shared__ __attribute__((aligned(16))) char at_smem[];
    __shared__ int s_last;
    constexpr int NC = DK / 16, NDB = DK / 32, KP = DK / 64, VJ = DK / 32;
    using L = AtLds<DK>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, g = lane >> 5;
    // consecutive block ids go round the 8 XCDs: give every XCD a contiguous range of work items, query tiles of one (sample, head)
    // fastest, so that the tiles sharing K / V share an L2
    const int nwork = gridDim.x;
    int w = blockIdx.x;
    if ((nwork & 7) == 0) w = (blockIdx.x & 7) * (nwork >> 3) + (blockIdx.x >> 3);
    const int qt = w % p.ntq, h = (w / p.ntq) % p.H, b = w / (p.ntq * p.H);
    const int q0 = qt * AT_QT + wave * AT_QW;
    const int qlim = p.qlens ? min(p.N, p.qlens[b] + p.qslack) : p.N;
    const bool live = q0 < qlim;                       // wave-uniform: waves past the last query only help staging
    if (!live && q0 < p.N) {                           // padding queries of a ragged batch: zero rows (finite for the layers that follow)
        for (int e = lane; e < AT_QW * (DK / 4); e += 64) {
            const int r = e / (DK / 4), c4 = e - r * (DK / 4);
            if (q0 + r < p.N) *reinterpret_cast<float4*>(p.out + ((size_t)b * p.N + q0 + r) * ((size_t)p.H * DK) + (size_t)h * DK + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (qt * AT_QT >= qlim) return;                    // block-uniform: nothing to compute for this tile
    const float* Kb = p.k + (size_t)b * p.bk + (size_t)h * DK;
    const float* Vb = p.v + (size_t)b * p.bv + (size_t)h * DK;
    const unsigned char* mk = p.kmask ? p.kmask + (size_t)b * p.M : nullptr;

    // ---- last key that is not padding -> number of key tiles
    if (tid == 0) s_last = mk ? -1 : p.M - 1;
    __syncthreads();
    if (mk) {
        int last = -1;
        for (int j = tid; j < p.M; j += 256) if (!mk[j]) last = j;
        if (last >= 0) atomicMax(&s_last, last);
        __syncthreads();
    }
    const int nt = s_last < 0 ? 1 : (s_last + AT_KT) / AT_KT;

    // ---- this lane's query as B fragments (k = d, column = query), hi / lo
    at_h8 qh[NC], ql[NC];                              // REL: q + u (content term)
    at_h8 rh[REL ? NC : 1], rl[REL ? NC : 1];          // REL: q + v (position term)
    {
        const int qi = min(q0 + col, p.N - 1);
        const float* Q = p.q + (size_t)b * p.bq + (size_t)qi * p.ldq + (size_t)h * DK + 8 * g;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float x[8];
            *reinterpret_cast<float4*>(x) = *reinterpret_cast<const float4*>(Q + c * 16);
            *reinterpret_cast<float4*>(x + 4) = *reinterpret_cast<const float4*>(Q + c * 16 + 4);
            if constexpr (REL) {
                float bu[8], bw[8], y[8];
                *reinterpret_cast<float4*>(bu) = *reinterpret_cast<const float4*>(p.bias_u + h * DK + 8 * g + c * 16);
                *reinterpret_cast<float4*>(bu + 4) = *reinterpret_cast<const float4*>(p.bias_u + h * DK + 8 * g + c * 16 + 4);
                *reinterpret_cast<float4*>(bw) = *reinterpret_cast<const float4*>(p.bias_v + h * DK + 8 * g + c * 16);
                *reinterpret_cast<float4*>(bw + 4) = *reinterpret_cast<const float4*>(p.bias_v + h * DK + 8 * g + c * 16 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) { y[e] = x[e] + bw[e]; x[e] += bu[e]; }
                at_split8(y, rh[c], rl[c]);
            }
            at_split8(x, qh[c], ql[c]);
        }
    }
    // ---- staging registers: K rows (8 floats per pass) and V columns (4 keys of one d per item)
    struct Stg { float kr[KP][8], vr[VJ][4], pr[REL ? 8 : 1]; unsigned char mb; bool pr_ok; };
    Stg sA, sB;                                        // two sets: the loads of a tile are issued TWO tiles ahead (one iteration is shorter
                                                       // than a trip to HBM: with one set in flight every tile waited for its loads)
    const int k_kl = tid & 3, k_d8 = (tid >> 2) & 7, k_kh = tid >> 5, k_key = k_kh * 4 + k_kl;
    // REL: block u of the position projection holds its rows R0 + 32 (u - 3) .. + 31 in K's fragment layout.  The row for (query i,
    // key j) is T-1 - i + j, so tile t of wave w reads blocks u = t - w + 3 and u + 1, the workgroup blocks t .. t + 4: a ring of 6
    // with one new block per tile.  The wave's [relative position][query] product goes through a per-wave scratch for the shift.
    char* pring = at_smem + 2 * L::STAGE;
    float* scr = reinterpret_cast<float*>(pring + L::PRING * L::PBLK + wave * L::SCR);
    const int R0 = p.N - 32 - qt * AT_QT, nrel = 2 * p.N - 1;
    const float* Pb = REL ? p.pos + (size_t)h * DK + k_d8 * 8 : nullptr;
    const long ldp = (long)p.H * DK;
    auto pos_load = [&](int u, float* dst) -> bool {
        const int r = R0 + 32 * (u - 3) + k_key;
        const bool ok = r >= 0 && r < nrel;
        const float* src = Pb + (size_t)(ok ? r : 0) * ldp;
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(src + 4);
        return ok;
    };
    auto pos_store = [&](int u, const float* x, bool ok) {
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = ok ? x[e] : 0.f;
        at_h8 hi, lo; at_split8(z, hi, lo);
        char* dst = pring + (u % L::PRING) * L::PBLK + ((k_d8 >> 1) * 2) * AT_FRAG + (k_d8 & 1) * AT_HALF + k_key * 16;
        *reinterpret_cast<at_h8*>(dst) = hi;
        *reinterpret_cast<at_h8*>(dst + AT_FRAG) = lo;
    };
    auto stage_load = [&](int t, Stg& r) {
        const int key0 = t * AT_KT;
        {
            const int kj = min(key0 + k_key, p.M - 1);
            const float* src = Kb + (size_t)kj * p.ldk + k_d8 * 8;
#pragma unroll
            for (int ps = 0; ps < KP; ++ps) {
                *reinterpret_cast<float4*>(r.kr[ps]) = *reinterpret_cast<const float4*>(src + ps * 64);
                *reinterpret_cast<float4*>(r.kr[ps] + 4) = *reinterpret_cast<const float4*>(src + ps * 64 + 4);
            }
        }
#pragma unroll
        for (int j = 0; j < VJ; ++j) {
            const int it = tid + 256 * j, d = it % DK, kq = it / DK;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.vr[j][i] = Vb[(size_t)min(key0 + 4 * kq + i, p.M - 1) * p.ldv + d];
        }
        r.mb = 0;
        if (tid < AT_KT) { const int kj = key0 + tid; r.mb = (kj >= p.M) ? 1 : (mk ? mk[kj] : 0); }
        if constexpr (REL) r.pr_ok = pos_load(t + 4, r.pr);
    };
    auto stage_store = [&](char* st, int t, const Stg& r) {
        if constexpr (REL) pos_store(t + 4, r.pr, r.pr_ok);
#pragma unroll
        for (int ps = 0; ps < KP; ++ps) {
            at_h8 hi, lo; at_split8(r.kr[ps], hi, lo);
            const int d8 = k_d8 + 8 * ps, c = d8 >> 1, gk = d8 & 1;
            char* dst = st + (c * 2) * AT_FRAG + gk * AT_HALF + k_key * 16;
            *reinterpret_cast<at_h8*>(dst) = hi;
            *reinterpret_cast<at_h8*>(dst + AT_FRAG) = lo;
        }
#pragma unroll
        for (int j = 0; j < VJ; ++j) {
            const int it = tid + 256 * j, d = it % DK, kq = it / DK;
            const int db = d >> 5, c2 = kq >> 2, e4 = (kq >> 1) & 1, gv = kq & 1;
            at_h4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) { hi[i] = (_Float16)r.vr[j][i]; lo[i] = (_Float16)((r.vr[j][i] - (float)hi[i]) * 2048.f); }
            char* dst = st + L::KF + ((db * 2 + c2) * 2) * AT_FRAG + gv * AT_HALF + (d & 31) * 16 + e4 * 8;
            *reinterpret_cast<at_h4*>(dst) = hi;
            *reinterpret_cast<at_h4*>(dst + AT_FRAG) = lo;
        }
        if (tid < AT_KT) reinterpret_cast<float*>(st + L::KF + L::VF)[tid] = r.mb ? NEG_INF : 0.f;
    };

    at_f16 ohh[NDB], olo[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ohh[db][r] = 0.f; olo[db][r] = 0.f; }
    float m_run = NEG_INF, l_run = 0.f;
    const int lane_off = g * AT_HALF + col * 16;

    if constexpr (REL) {                               // blocks 0 .. 3 (tile 0 adds block 4)
        float x[4][8]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ok[u] = pos_load(u, x[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) pos_store(u, x[u], ok[u]);
    }
    auto tile = [&](int t, const char* cur) {
        if (live) {
            // ---- S^T tile: 32 keys x 32 queries
            at_f16 shh, slo;
#pragma unroll
            for (int r = 0; r < 16; ++r) { shh[r] = 0.f; slo[r] = 0.f; }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const at_h8 kh = *reinterpret_cast<const at_h8*>(cur + (c * 2) * AT_FRAG + lane_off);
                const at_h8 kl = *reinterpret_cast<const at_h8*>(cur + (c * 2 + 1) * AT_FRAG + lane_off);
                shh = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], shh, 0, 0, 0);
                slo = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], slo, 0, 0, 0);
                slo = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], slo, 0, 0, 0);
            }
            const float* bias = reinterpret_cast<const float*>(cur + L::KF + L::VF);
            float s[16], mx = NEG_INF;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = shh[r] + slo[r] * (1.f / 2048.f);
            if constexpr (REL) {
                // G^T[rr][query] = pos[rbase + rr] . (q + v), rr = 0 .. 63 (two row blocks); the score of key jj wants rr = jj - query + 31
                const int u0 = t - wave + 3;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const char* pf = pring + ((u0 + blk) % L::PRING) * L::PBLK;
                    at_f16 ghh, glo;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ghh[r] = 0.f; glo[r] = 0.f; }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const at_h8 ph_ = *reinterpret_cast<const at_h8*>(pf + (c * 2) * AT_FRAG + lane_off);
                        const at_h8 pl_ = *reinterpret_cast<const at_h8*>(pf + (c * 2 + 1) * AT_FRAG + lane_off);
                        ghh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph_, rh[c], ghh, 0, 0, 0);
                        glo = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl_, rh[c], glo, 0, 0, 0);
                        glo = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph_, rl[c], glo, 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float4 o;
                        o.x = ghh[4 * j + 0] + glo[4 * j + 0] * (1.f / 2048.f); o.y = ghh[4 * j + 1] + glo[4 * j + 1] * (1.f / 2048.f);
                        o.z = ghh[4 * j + 2] + glo[4 * j + 2] * (1.f / 2048.f); o.w = ghh[4 * j + 3] + glo[4 * j + 3] * (1.f / 2048.f);
                        *reinterpret_cast<float4*>(scr + col * L::SCR_PITCH + 32 * blk + 8 * j + 4 * g) = o;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const float* sq = scr + col * (L::SCR_PITCH - 1) + 31 + 4 * g;      // [query][jj - query + 31]
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += sq[8 * (r >> 2) + (r & 3)];
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + 8 * j + 4 * g);
                s[4 * j + 0] = s[4 * j + 0] * p.scale + b4.x;
                s[4 * j + 1] = s[4 * j + 1] * p.scale + b4.y;
                s[4 * j + 2] = s[4 * j + 2] * p.scale + b4.z;
                s[4 * j + 3] = s[4 * j + 3] * p.scale + b4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == NEG_INF) ? 0.f : m_new;       // nothing but masked keys so far: every weight is exp(-inf) = 0
            const float alpha = __expf(m_run - m_use);
            float ps = 0.f;
            at_h8 ph[2], pl[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(s[r] - m_use);
                ps += e;
                const _Float16 hi = (_Float16)e;
                ph[r >> 3][r & 7] = hi;
                pl[r >> 3][r & 7] = (_Float16)((e - (float)hi) * 2048.f);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ohh[db][r] *= alpha; olo[db][r] *= alpha; }
            }
            // ---- O^T += V^T . P^T
            const char* vf = cur + L::KF;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const at_h8 vh = *reinterpret_cast<const at_h8*>(vf + ((db * 2 + c2) * 2) * AT_FRAG + lane_off);
                    const at_h8 vl = *reinterpret_cast<const at_h8*>(vf + ((db * 2 + c2) * 2 + 1) * AT_FRAG + lane_off);
                    ohh[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[c2], ohh[db], 0, 0, 0);
                    olo[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[c2], olo[db], 0, 0, 0);
                    olo[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[c2], olo[db], 0, 0, 0);
                }
        }
    };
    stage_load(0, sA);
    stage_store(at_smem, 0, sA);
    if (nt > 1) stage_load(1, sB);
    __syncthreads();
    for (int t = 0; t < nt; t += 2) {
        if (t + 2 < nt) stage_load(t + 2, sA);
        tile(t, at_smem);
        if (t + 1 >= nt) break;
        stage_store(at_smem + L::STAGE, t + 1, sB);
        __syncthreads();
        if (t + 3 < nt) stage_load(t + 3, sB);
        tile(t + 1, at_smem + L::STAGE);
        if (t + 2 < nt) stage_store(at_smem, t + 2, sA);
        __syncthreads();
    }
    // ---- out[b, q, h, d] = O^T[d][q] / l   (a lane: one query, d = 32 db + 8 j + 4 g + 0..3)
    if (live) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;                 // every key masked: 0 / 0 = NaN, as torch's soft-max of an all -inf row
        const int qi = q0 + col;
        if (qi < p.N) {
            float* O = p.out + ((size_t)b * p.N + qi) * ((size_t)p.H * DK) + (size_t)h * DK + 4 * g;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 o;
                    o.x = (ohh[db][4 * j + 0] + olo[db][4 * j + 0] * (1.f / 2048.f)) * inv;
                    o.y = (ohh[db][4 * j + 1] + olo[db][4 * j + 1] * (1.f / 2048.f)) * inv;
                    o.z = (ohh[db][4 * j + 2] + olo[db][4 * j + 2] * (1.f / 2048.f)) * inv;
                    o.w = (ohh[db][4 * j + 3] + olo[db][4 * j + 3] * (1.f / 2048.f)) * inv;
                    *reinterpret_cast<float4*>(O + db * 32 + 8 * j) = o;
                }
        }
    }
}

template <int DK, bool REL>
static int at_launch(const AtParams& p, hipStream_t st)
{
    using L = AtLds<DK>;
    const size_t lds = 2 * (size_t)L::STAGE + (REL ? (size_t)L::PRING * L::PBLK + (size_t)AT_WAVES * L::SCR : 0);
    auto k = attention_split_kernel<DK, REL>;
    set_max_dynamic_lds((const void*)k, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(p.ntq * p.H * p.B)), dim3(256), lds, st, p);
    return check_launch("attention_split");
}

}  // namespace dsp

extern "C" int dsp_attention_split(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const unsigned char* key_pad_mask,
                                   float* out, int B, int N, int M, int H, int DK, float scale, const int* q_lens, int q_slack, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || N < 1 || M < 1 || H < 1 || (DK != 64 && DK != 128)) {
        set_error("attention_split: bad sizes B=%d N=%d M=%d H=%d dk=%d (head width 64 or 128)", B, N, M, H, DK); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !v || !out) { set_error("attention_split: null pointer"); return DSP_EINVAL; }
    if (((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) || ((ldq | ldk | ldv) & 3) || ldq < (long)H * DK || ldk < (long)H * DK
        || ldv < (long)H * DK) {
        set_error("attention_split: q / k / v / out must be 16-byte aligned with row strides >= H * dk and %% 4 == 0"); return DSP_EINVAL; }
    AtParams p;
    p.q = q; p.k = k; p.v = v; p.kmask = key_pad_mask; p.out = out;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.bq = (long)N * ldq; p.bk = (long)M * ldk; p.bv = (long)M * ldv;
    p.B = B; p.N = N; p.M = M; p.H = H; p.ntq = (N + AT_QT - 1) / AT_QT; p.scale = scale;
    hipStream_t st = as_stream(stream);
    p.pos = nullptr; p.bias_u = nullptr; p.bias_v = nullptr; p.qlens = q_lens; p.qslack = q_slack < 0 ? 0 : q_slack;
    return DK == 64 ? at_launch<64, false>(p, st) : at_launch<128, false>(p, st);
}

extern "C" int dsp_relpos_attention(const float* q, const float* k, const float* v, long ld, const float* pos, const float* bias_u, const float* bias_v,
                                          const unsigned char* pad_mask, float* out, int B, int T, int H, int DK, dsp_stream_t stream)
{
    using namespace dsp;
    if (B < 0 || T < 1 || H < 1 || DK != 64) { set_error("relpos_attention: bad sizes B=%d T=%d H=%d dk=%d (head width 64)", B, T, H, DK); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!q || !k || !v || !pos || !bias_u || !bias_v || !out) { set_error("relpos_attention: null pointer"); return DSP_EINVAL; }
    if (((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out) | ((uintptr_t)pos) | ((uintptr_t)bias_u) | ((uintptr_t)bias_v)) & 15) || (ld & 3)
        || ld < (long)H * DK) {
        set_error("relpos_attention: pointers must be 16-byte aligned, row stride >= H * dk and %% 4 == 0"); return DSP_EINVAL; }
    AtParams p;
    p.q = q; p.k = k; p.v = v; p.kmask = pad_mask; p.out = out;
    p.ldq = p.ldk = p.ldv = ld; p.bq = p.bk = p.bv = (long)T * ld;
    p.B = B; p.N = T; p.M = T; p.H = H; p.ntq = (T + AT_QT - 1) / AT_QT; p.scale = 1.f / sqrtf((float)DK);
    p.pos = pos; p.bias_u = bias_u; p.bias_v = bias_v; p.qlens = nullptr; p.qslack = 0;
    return at_launch<64, true>(p, as_stream(stream));
}
