// Which MFMA shape feeds the "3 x fp16" split-precision conv loop better on gfx950 — the question behind DESIGN §8's "32x32x16 in the unit kernel"?
// Same data flow as hifigan_resunit_f32_kernel's inner loop at C = 128 (weights: fragment-order hi / lo from an L2-resident buffer, one step
// ahead; activations: hi / lo tiles in LDS, swizzled; three MFMAs per fragment pair; 8 waves per workgroup, 2 workgroups per CU, every CU busy):
//   shape 0: v_mfma_f32_16x16x32_f16, 8 x 1 waves (16 channels x 128 columns per wave): per 32-K step 2 weight loads, 16 ds_read_b128, 24 MFMAs
//   shape 1: v_mfma_f32_32x32x16_f16, 4 x 2 waves (32 channels x  64 columns per wave): per 16-K step 2 weight loads,  4 ds_read_b128,  6 MFMAs
// Prints TFLOP/s of MFMA issue and the clock the kernel ran at (wall clock vs s_memtime is not needed: both run the same flops).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shapes tools/ubench/mfma_split_shapes.hip && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int C = 128, NT = 128, CH = C / 8;          // tile: 128 columns x 128 channels, hi and lo planes (2 x 32 KB)

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row & 7) << 1); }

template <int SHAPE>
__global__ __launch_bounds__(512, 4) void k(const _Float16* __restrict__ w, float* __restrict__ out, int nsteps32, int reps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* thi = smem; char* tlo = smem + NT * C * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NT * CH * 2; e += 512) reinterpret_cast<h8*>(smem)[e] = (h8){(_Float16)0.01f, (_Float16)0.02f, (_Float16)-0.01f, (_Float16)0.03f, (_Float16)0.f, (_Float16)0.01f, (_Float16)0.02f, (_Float16)-0.02f};
    __syncthreads();
    float sum = 0.f;
    if (SHAPE == 0) {
        const int lr = lane & 15, lk = lane >> 4;
        f4 am[8], ac[8];
        for (int j = 0; j < 8; ++j) { am[j] = (f4){0, 0, 0, 0}; ac[j] = (f4){0, 0, 0, 0}; }
        for (int r = 0; r < reps; ++r) {
            h8 ah = *reinterpret_cast<const h8*>(w + (size_t)wave * 512 + lane * 8), al = *reinterpret_cast<const h8*>(w + 65536 + (size_t)wave * 512 + lane * 8);
            for (int s = 0; s < nsteps32; ++s) {
                const int sn = (s + 1) % nsteps32;
                const h8 nh = *reinterpret_cast<const h8*>(w + ((size_t)sn * 8 + wave) * 512 + lane * 8);
                const h8 nl = *reinterpret_cast<const h8*>(w + 65536 + ((size_t)sn * 8 + wave) * 512 + lane * 8);
                const int c = s & 3, sh = s >> 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = (j * 16 + lr + sh) & (NT - 1);
                    const size_t o = ((size_t)row * CH + swz(row, c * 4 + lk)) * 16;
                    const h8 bh = *reinterpret_cast<const h8*>(thi + o), bl = *reinterpret_cast<const h8*>(tlo + o);
                    am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, am[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, ac[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, ac[j], 0, 0, 0);
                }
                ah = nh; al = nl;
            }
        }
        for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) sum += am[j][e] + ac[j][e];
    } else {
        const int ln = lane & 31, lg = lane >> 5, wm = wave & 3, wn = wave >> 2;
        f16v am[2], ac[2];
        for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) { am[j][v] = 0.f; ac[j][v] = 0.f; }
        for (int r = 0; r < reps; ++r) {
            h8 ah = *reinterpret_cast<const h8*>(w + (size_t)wm * 512 + lane * 8), al = *reinterpret_cast<const h8*>(w + 65536 + (size_t)wm * 512 + lane * 8);
            for (int s = 0; s < 2 * nsteps32; ++s) {                      // 16-K steps
                const int sn = (s + 1) % (2 * nsteps32);
                const h8 nh = *reinterpret_cast<const h8*>(w + ((size_t)sn * 4 + wm) * 512 + lane * 8);
                const h8 nl = *reinterpret_cast<const h8*>(w + 65536 + ((size_t)sn * 4 + wm) * 512 + lane * 8);
                const int k16 = s & 7, sh = s >> 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = ((wn * 2 + j) * 32 + ln + sh) & (NT - 1);
                    const size_t o = ((size_t)row * CH + swz(row, k16 * 2 + lg)) * 16;
                    const h8 bh = *reinterpret_cast<const h8*>(thi + o), bl = *reinterpret_cast<const h8*>(tlo + o);
                    am[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, am[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac[j], 0, 0, 0);
                }
                ah = nh; al = nl;
            }
        }
        for (int j = 0; j < 2; ++j) for (int v = 0; v < 16; ++v) sum += am[j][v] + ac[j][v];
    }
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}

template <int SHAPE> void run(const char* name, const _Float16* w, float* out)
{
    const int nsteps32 = 28, reps = 40, grid = 512 * 4;                   // k = 7 taps x 4 channel chunks per "conv", 40 convs per workgroup
    auto kern = k<SHAPE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT * C * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * NT * C * 2, 0, w, out, nsteps32, reps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
    }
    const double flops = (double)grid * reps * nsteps32 * 8 /*waves*/ * 24 /*MFMAs of 16x16x32 per 32-K step*/ * 16384.0;
    printf("%-34s %8.3f ms  %7.1f TFLOP/s of fp16 MFMA issue (%.0f %% of 2.5 PF)\n", name, best, flops / best / 1e9, 100 * flops / best / 1e9 / 2500);
}

int main()
{
    _Float16* w; float* out;
    hipMalloc(&w, 2 * 65536 * sizeof(_Float16) * 8); hipMemset(w, 0x11, 2 * 65536 * sizeof(_Float16) * 8);
    hipMalloc(&out, (size_t)2048 * 512 * 4);
    run<0>("16x16x32, 8x1 waves, 16 reads/step", w, out);
    run<1>("32x32x16, 4x2 waves,  4 reads/step", w, out);
    run<0>("16x16x32 again", w, out);
    return 0;
}
