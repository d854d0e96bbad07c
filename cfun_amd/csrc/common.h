// Shared device/host helpers for libcfun_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cfun_hip.h"

#define CFUN_WAVE 64

// dynamic LDS (all LDS scratch lives in the dynamic region, 16-byte aligned: guide G17)
#ifdef CFUN_HIP_EMULATION   /* tests/emu: host build of the same sources */
#define CFUN_DYN_LDS(T, name) T* name = reinterpret_cast<T*>(hipemu::g.dyn_lds)
#else
#define CFUN_DYN_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

#define CFUN_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

static inline hipStream_t cfun_st(cfun_stream_t s) { return (hipStream_t)s; }

static inline size_t cfun_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline bool cfun_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// where a weight gradient goes: the packed layout [taps][Ci][CoP] (oidhw = 0) or torch's OIDHW [Co][Ci][taps]
// (oidhw = 1; the chunk reduction and the layout change are then one kernel, conv3d_direct.hip)
struct CfunWgradDst { float* ptr; int oidhw; };

__device__ __forceinline__ float cfun_apply_act(float v, int act, float slope) {
  if (act == CFUN_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == CFUN_ACT_LRELU) return v > 0.f ? v : v * slope;
  return v;
}

// 1-ulp hardware square root / reciprocal (v_sqrt_f32 / v_rcp_f32): one quarter-rate instruction instead of the ~10 of
// the correctly rounded sqrtf() / division expansions; sqrt(0) = 0, rcp(0) = inf (so 0 * rcp(0) is NaN like 0 / 0)
__device__ __forceinline__ float cfun_fast_sqrt(float v) {
#ifdef CFUN_HIP_EMULATION
  return sqrtf(v);
#else
  return __builtin_amdgcn_sqrtf(v);
#endif
}
__device__ __forceinline__ float cfun_fast_rcp(float v) {
#ifdef CFUN_HIP_EMULATION
  return 1.0f / v;
#else
  return __builtin_amdgcn_rcpf(v);
#endif
}

__device__ __forceinline__ float cfun_fast_rsq(float v) {      // v_rsq_f32 (1 ulp); rsq(0) = inf
#ifdef CFUN_HIP_EMULATION
  return 1.0f / sqrtf(v);
#else
  return __builtin_amdgcn_rsqf(v);
#endif
}

// wave-level sum (all 64 lanes receive the total)
__device__ __forceinline__ float cfun_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double cfun_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
