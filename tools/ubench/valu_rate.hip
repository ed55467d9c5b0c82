// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP>
__global__ void k(float* out, int iters, int e) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    int ia[8]; for (int i = 0; i < 8; ++i) ia[i] = threadIdx.x + i;
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = fmaf(a[i], 1.0001f, 0.5f);
                if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], (v2f){1.0001f, 0.9999f}, (v2f){0.5f, 0.25f});
                if (OP == 2) a[i] = ldexpf(a[i], e);
                if (OP == 3) a[i] = __builtin_amdgcn_exp2f(a[i] * 1e-3f);
                if (OP == 4) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);
                if (OP == 5) ia[i] = max(ia[i] - e, ia[(i + 1) & 7]);
                if (OP == 6) a[i] = fmaxf(a[i], a[(i + 1) & 7] * 0.5f);
                if (OP == 7) ia[i] = ia[i] - e;
                if (OP == 8) a[i] = a[i] > 0.5f ? a[i] * 0.999f : a[i];
            }
        }
    }
    long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + ia[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int OP> void run(const char* name, float* d, int threads) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 0);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 0);
    hipDeviceSynchronize();
    float c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
    printf("%-14s threads/WG=%4d (waves/SIMD=%d): %.2f cycles per instr per wave (clock64 units)\n", name, threads, threads / 256, c / (iters * (double)REP));
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", d, th); run<1>("v_pk_fma_f32", d, th); run<2>("v_ldexp_f32", d, th); run<3>("v_exp_f32(+mul)", d, th);
        run<4>("v_log_f32(+add)", d, th); run<5>("sub+max_i32", d, th); run<6>("mul+max_f32", d, th); run<7>("v_sub_u32", d, th); run<8>("cmp+cndmask+mul", d, th);
    }
    return 0;
}
