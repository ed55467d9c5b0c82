// f64 VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction at 1, 2 and 4 waves per SIMD.
// Decides whether a double-precision exp-space DP row (128 DFMA per SIMD-row) beats the f32 window-local one.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, int iters, int e, double w0) {
    double a[8], w[8];
    float f[8]; v2f p2[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; w[i] = w0 + i * 1e-9; f[i] = (float)a[i]; p2[i].x = f[i]; p2[i].y = f[i] + 0.5f; }
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_fma(a[i], w[i], w[(i + 1) & 7]);
                if (OP == 1) a[i] = a[i] * w[i];
                if (OP == 2) a[i] = __builtin_ldexp(a[i], e);
                if (OP == 3) a[i] = (double)f[i] + 0.0 * a[i], f[i] += 1.f;   // cvt_f64_f32 + fma
                if (OP == 4) a[i] = __builtin_amdgcn_frexp_mant(a[i]) + w[i];
                if (OP == 5) f[i] = (float)a[i], a[i] = a[i] + w[i];           // cvt_f32_f64 + add
                if (OP == 6) a[i] = a[i] + w[i];
                if (OP == 7) f[i] = fmaf(f[i], 1.0001f, 0.5f);
                if (OP == 8) { p2[i] = __builtin_elementwise_fma(p2[i], (v2f){1.0001f, 0.9999f}, (v2f){0.5f, 0.25f}); }
                if (OP == 9) f[i] = ldexpf(f[i], e);
            }
        }
    }
    long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i] + p2[i].x + p2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int OP> void run(const char* name, float* d, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, iters, 0, 1.0000001);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    float c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
    // wall: ns per SIMD per wave-instruction (all waves of the SIMD together)
    double ns_per_simd_instr = ms * 1e6 / ((double)iters * REP * (threads / 256));
    printf("%-24s waves/SIMD=%d: wave0 %.2f ticks/instr; wall %.3f ns per SIMD-instr (= %.2f cycles at 2.4 GHz)\n", name, threads / 256,
           c / (iters * (double)REP), ns_per_simd_instr, ns_per_simd_instr * 2.4);
}

// LDS window read: every lane reads NR consecutive 16-byte chunks at a 16*STRIDE-byte lane stride (the f64 row window)
template <int NR, int STRIDE>
__global__ void lds_k(float* out, int iters) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 acc = {0, 0};
    const d2* base = (const d2*)lds + (threadIdx.x & 63) * STRIDE + (threadIdx.x >> 6) * 64;
    long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        d2 v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = base[r + (it & 1)];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc += v[r];
    }
    long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(acc.x + acc.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int NR, int STRIDE> void run_lds(float* d, int threads) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((lds_k<NR, STRIDE>), dim3(256), dim3(threads), 65536, 0, d, iters);
        hipDeviceSynchronize();
    }
    float c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
    printf("lds b128 x%d stride %2dB threads=%4d: %.1f cycles per iteration per wave (%.2f per read; incl %d f64 pk adds)\n", NR, STRIDE * 16, threads,
           c / (double)iters, c / (double)iters / NR, NR);
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f64", d, th); run<1>("v_mul_f64", d, th); run<2>("v_ldexp_f64", d, th); run<3>("cvt_f64_f32+fma64+add32", d, th);
        run<4>("frexp_mant_f64+add64", d, th); run<5>("cvt_f32_f64+add64", d, th); run<6>("v_add_f64", d, th); run<7>("v_fma_f32", d, th); run<8>("v_pk_fma_f32", d, th); run<9>("v_ldexp_f32", d, th);
    }
    for (int th : {256, 512}) { run_lds<17, 1>(d, th); run_lds<17, 2>(d, th); run_lds<9, 1>(d, th); }
    return 0;
}
