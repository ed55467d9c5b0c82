// common.h — shared device/host helpers for the gfx950 (CDNA4) kernels of the DASpeech hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "../../include/daspeech_dag.h"

#define DSP_WAVE 64
#define NEG_INF (-__builtin_huge_valf())

namespace dsp {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
void set_max_dynamic_lds(const void* fn, int bytes);      // hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per (kernel, device) instead of per launch

static inline hipStream_t as_stream(dsp_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- wave / block reductions (wave = 64 lanes on CDNA) ----
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// element <-> float conversion for the three logits dtypes
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ __hip_bfloat16 from_f<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

}  // namespace dsp
