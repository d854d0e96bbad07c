// Hardware probes used by tools/probe_mfma.py (not part of the public ABI): dump the lane layout of
// v_mfma_f32_4x4x1_16b_f32 so kernels that use it can be written against measured facts.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CFUN_HIP_EMULATION
__global__ void k_probe_mfma_4x4x1(const float* a, const float* b, float* d) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}

extern "C" int cfun_debug_mfma_4x4x1(const float* a, const float* b, float* d, cfun_stream_t stream) {
  hipLaunchKernelGGL(k_probe_mfma_4x4x1, dim3(1), dim3(64), 0, cfun_st(stream), a, b, d);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}
#else
extern "C" int cfun_debug_mfma_4x4x1(const float*, const float*, float*, cfun_stream_t) { return CFUN_EINVAL; }
#endif
