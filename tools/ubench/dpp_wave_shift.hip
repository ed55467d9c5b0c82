// dpp_wave_shift.hip — does gfx950 execute the GFX9 whole-wave DPP shifts (wave_shr:1 / wave_shl:1), and what does a chain of them cost next to
// the LDS round trip it could replace in the strip kernels' row head?   hipcc --offload-arch=gfx950 -O3 dpp_wave_shift.hip -o dpp_wave_shift
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void sem(float* out_r, float* out_l, const float* in) {
    const int l = threadIdx.x;
    const float v = in[l];
    const float old = -1.f;                  // lanes without a source keep `old` (bound_ctrl off)
    out_r[l] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));   // wave_shr:1: lane l <- lane l-1
    out_l[l] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xf, 0xf, false));   // wave_shl:1: lane l <- lane l+1
}
// N dependent steps of a window exchange per iteration: (a) 8 chained wave_shr on 5 registers, (b) write 5 dwords to LDS, barrier-free wave-local
// s_waitcnt, read back 9 x b128 + 9 dwords (the strip kernels' row head)
template <int MODE>
__global__ void cost(float* out, const float* in, int iters, long long* ticks) {
    __shared__ __attribute__((aligned(16))) float buf[2][64 * 4 + 64];
    __shared__ int xb[2][64 + 16];
    const int l = threadIdx.x;
    float v[4] = {in[l], in[l + 64], in[l + 128], in[l + 192]}; int x = l;
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            float w[4] = {v[0], v[1], v[2], v[3]}; int xs = x;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(w[e]), __float_as_int(w[e]), 0x138, 0xf, 0xf, false));
                xs = __builtin_amdgcn_update_dpp(xs, xs, 0x138, 0xf, 0xf, false);
                acc += w[0] * w[1] + w[2] * w[3] + (float)xs;
            }
        } else {
            const int cur = it & 1;
            *reinterpret_cast<float4*>(&buf[cur][4 * l + 32]) = make_float4(v[0], v[1], v[2], v[3]);
            xb[cur][l + 8] = x;
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float4 p = *reinterpret_cast<const float4*>(&buf[cur][4 * l + 4 * k]);
                acc += p.x * p.y + p.z * p.w + (float)xb[cur][l + k];
            }
        }
        v[0] += acc * 1e-30f; x += 1;
    }
    const long long t1 = clock64();
    out[l] = acc;
    if (l == 0) ticks[MODE] = t1 - t0;
}
int main() {
    float *in, *o1, *o2; long long* tk;
    hipMalloc(&in, 1024 * 4); hipMalloc(&o1, 1024 * 4); hipMalloc(&o2, 1024 * 4); hipMalloc(&tk, 16);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, o1, o2, in);
    std::vector<float> r(64), l(64);
    hipMemcpy(r.data(), o1, 256, hipMemcpyDeviceToHost); hipMemcpy(l.data(), o2, 256, hipMemcpyDeviceToHost);
    printf("wave_shr:1 lanes 0,1,15,16,17,31,32,33,63: %g %g %g %g %g %g %g %g %g\n", r[0], r[1], r[15], r[16], r[17], r[31], r[32], r[33], r[63]);
    printf("wave_shl:1 lanes 0,1,15,16,30,31,32,62,63: %g %g %g %g %g %g %g %g %g\n", l[0], l[1], l[15], l[16], l[30], l[31], l[32], l[62], l[63]);
    const int iters = 20000;
    hipLaunchKernelGGL(cost<0>, dim3(1), dim3(64), 0, 0, o1, in, iters, tk);
    hipLaunchKernelGGL(cost<1>, dim3(1), dim3(64), 0, 0, o2, in, iters, tk);
    long long t[2]; hipMemcpy(t, tk, 16, hipMemcpyDeviceToHost);
    printf("per iteration (clock64 ticks, one wave alone): 8-step wave_shr chain on 5 registers %.1f | LDS write + 9 x b128 + 9 dword read-back %.1f\n",
           (double)t[0] / iters, (double)t[1] / iters);
    return 0;
}
