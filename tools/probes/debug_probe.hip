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

// Where and when does each workgroup of a launch run?  Every block records HW_ID, XCC_ID and its start / end time
// (s_memrealtime, 100 MHz) around `spin` dependent FMAs, holding `lds_bytes` of LDS (tools/probe_dispatch.py).
__global__ void __launch_bounds__(256) k_probe_dispatch(int spin, unsigned long long* out) {
  CFUN_DYN_LDS(float, smem);
  const unsigned long long t0 = __builtin_readcyclecounter() * 0 + wall_clock64();
  float v = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0000001f + 1e-7f;
  if (v == 12345.678f) smem[threadIdx.x] = v;   // keeps the loop and the LDS allocation alive
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = t0;
    out[blockIdx.x * 4 + 3] = t1;
  }
}

extern "C" int cfun_debug_dispatch(int nblocks, int lds_bytes, int spin, unsigned long long* out, cfun_stream_t stream) {
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe_dispatch),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k_probe_dispatch, dim3(nblocks), dim3(256), lds_bytes, cfun_st(stream), spin, out);
  CFUN_LAUNCH_CHECK();
  return CFUN_OK;
}
#else
extern "C" int cfun_debug_dispatch(int, int, int, unsigned long long*, cfun_stream_t) { return CFUN_EINVAL; }
extern "C" int cfun_debug_mfma_4x4x1(const float*, const float*, float*, cfun_stream_t) { return CFUN_EINVAL; }
#endif
