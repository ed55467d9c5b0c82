// v_pk_fma_f32 issue cost on gfx950 as the DP kernels use it: acc[c] += w[k] * E[c][k], all three operands VGPR pairs, NCH independent
// accumulator chains, 1 wave per SIMD (256 threads) and 2.  Prints cycles per instruction per wave (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NCH, int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    v2f acc[NCH]; v2f w[8]; v2f E[16];
    for (int i = 0; i < NCH; ++i) { acc[i].x = 0.f; acc[i].y = 0.f; }
    for (int i = 0; i < 8; ++i) { w[i].x = seed + threadIdx.x * 1e-6f + i; w[i].y = w[i].x * 0.5f; }
    for (int i = 0; i < 16; ++i) { E[i].x = 1e-3f * (i + 1) + seed; E[i].y = 2e-3f * (i + 1); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (KIND == 0) acc[c] = __builtin_elementwise_fma(w[kk], E[(kk * NCH + c) & 15], acc[c]);           // packed, 3 VGPR-pair operands
                else { acc[c].x = fmaf(w[kk].x, E[(kk * NCH + c) & 15].x, acc[c].x); acc[c].y = fmaf(w[kk].y, E[(kk * NCH + c) & 15].y, acc[c].y); }
            }
        }
        asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NCH; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int NCH, int KIND> void run(const char* name, float* d, int threads) {
    const int iters = 4000;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<NCH, KIND>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f); hipDeviceSynchronize(); }
    float c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NCH * (KIND == 0 ? 1 : 2);
    printf("%-28s chains=%d waves/SIMD=%d: %.2f memtime ticks per instr (x24 = shader cycles at 2.4 GHz / 100 MHz)  -> %.2f cycles\n", name, NCH, threads / 256, c / n, c / n * 24.0);
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int th : {256, 512}) {
        run<4, 0>("v_pk_fma_f32", d, th); run<8, 0>("v_pk_fma_f32", d, th); run<16, 0>("v_pk_fma_f32", d, th);
        run<4, 1>("v_fma_f32 (2 per pair)", d, th); run<8, 1>("v_fma_f32 (2 per pair)", d, th);
    }
    return 0;
}
