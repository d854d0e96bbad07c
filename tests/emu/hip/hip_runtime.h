// TEST INFRASTRUCTURE -- a tiny functional emulator of the HIP subset used by cfun_amd/csrc/*.hip.
//
// It lets the CPU-only test tier (-m "not gpu") execute the *same kernel sources* that hipcc compiles for
// gfx950, built here as plain host C++ (clang++ -x c++): every thread of a workgroup is a ucontext fiber,
// __syncthreads() / wave shuffles / v_mfma_f32_16x16x4_f32 are rendezvous points with the documented lane
// semantics, workgroups run one after another.  "Device" pointers are host pointers.  It is slow and is
// only ever loaded when a test sets CFUN_LIB_PATH explicitly; the product loads libcfun_hip.so and fails
// loudly without it.  Nothing here is shipped or measured.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CFUN_HIP_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

typedef int hipError_t;
#define hipSuccess 0
typedef void* hipStream_t;
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated hip error"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define hipDeviceAttributeMaxSharedMemoryPerBlock 0
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 160 * 1024; return hipSuccess; }   // MI355X: 160 KB LDS per CU

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct alignas(16) float4 {
  float x, y, z, w;
};
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct alignas(8) float2 {
  float x, y;
};
inline float2 make_float2(float a, float b) { return float2{a, b}; }

namespace hipemu {

struct State {
  dim3 tidx, bidx, bdim, gdim;
  void* dyn_lds = nullptr;
};
extern State g;
void launch(dim3 grid, dim3 block, size_t lds, const std::function<void()>& body);
void block_sync();
// one rendezvous of the calling thread's wave; returns a pointer to the wave's 64 x 16-byte exchange slots of
// the current collective (double buffered, see hip_emu.cpp)
struct alignas(16) Slot {
  unsigned char b[32];
};
Slot* wave_exchange(const void* mine, size_t bytes);
int lane_id();

template <class T>
T shfl_from(T v, int src) {
  static_assert(sizeof(T) <= 16, "");
  Slot* s = wave_exchange(&v, sizeof(T));
  T r;
  memcpy(&r, s[src & 63].b, sizeof(T));
  return r;
}

}  // namespace hipemu

#define threadIdx (hipemu::g.tidx)
#define blockIdx (hipemu::g.bidx)
#define blockDim (hipemu::g.bdim)
#define gridDim (hipemu::g.gdim)

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  hipemu::launch((grid), (block), (size_t)(lds), [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { hipemu::block_sync(); }
template <class T>
T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_from(v, hipemu::lane_id() ^ mask); }
template <class T>
T __shfl(T v, int src, int = 64) { return hipemu::shfl_from(v, src); }

template <class T>
T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }

// scheduling / uniformity hints: no-ops on the host
inline void __builtin_amdgcn_sched_barrier(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }

// IEEE single operations without contraction (the emulator is compiled with -ffp-contract=off)
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fdiv_rn(float a, float b) { return a / b; }

// v_mfma_f32_16x16x4_f32: D[i][j] = C[i][j] + sum_k A[i][k]*B[k][j], k ascending, one fma per product.
// lane l holds A[l&15][l>>4], B[l>>4][l&15]; C/D: col j = l&15, row i = (l>>4)*4 + reg.
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  float ab[2] = {a, b};
  hipemu::Slot* s = hipemu::wave_exchange(ab, sizeof(ab));
  const int l = hipemu::lane_id(), j = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, s[i + 16 * k].b, 4);
      memcpy(&bv, s[j + 16 * k].b + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32

// v_mfma_f32_16x16x32_bf16: lane l holds 8 bf16 of A row l&15 and of B column l&15 for the K-block l>>4 (as raw bits in
// a 16-byte vector); D[i][j] = C[i][j] + sum over the 4 K-blocks x 8 elements of A[i][k] * B[k][j].  The hardware's
// internal summation order is not specified; fp32 fma in ascending (block, element) order here, tests use tolerances.
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
inline float hipemu_bf16_bits(const unsigned char* p, int e) {
  unsigned short h;
  memcpy(&h, p + 2 * e, 2);
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x4 c) {
  unsigned char ab[32];
  memcpy(ab, &a, 16);
  memcpy(ab + 16, &b, 16);
  hipemu::Slot* s = hipemu::wave_exchange(ab, sizeof(ab));
  const int l = hipemu::lane_id(), j = l & 15;
  hipemu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int kb = 0; kb < 4; ++kb)
      for (int e = 0; e < 8; ++e)
        acc = fmaf(hipemu_bf16_bits(s[i + 16 * kb].b, e), hipemu_bf16_bits(s[j + 16 * kb].b + 16, e), acc);
    d[r] = acc;
  }
  return d;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent blocks b = lane>>2; D_b[i][j] = C + A_b[i]*B_b[j] with A_b[i] from lane
// 4b+i, B_b[j] from lane 4b+j; result lane 4b+j, register i  (layout measured on MI355X: tools/probe_mfma.py)
inline hipemu_f32x4 hipemu_mfma_4x4x1f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  float ab[2] = {a, b};
  hipemu::Slot* s = hipemu::wave_exchange(ab, sizeof(ab));
  const int l = hipemu::lane_id(), blk = l >> 2;
  float bv;
  memcpy(&bv, s[l].b + 4, 4);
  hipemu_f32x4 d = c;
  for (int i = 0; i < 4; ++i) {
    float av;
    memcpy(&av, s[4 * blk + i].b, 4);
    d[i] = fmaf(av, bv, c[i]);
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_4x4x1f32 hipemu_mfma_4x4x1f32
