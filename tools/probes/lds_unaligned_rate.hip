// Probe: cost of ds_read_b128 by address alignment on gfx950 (cycles per wave-instruction, one wave per SIMD).
//   hipcc --offload-arch=gfx950 -O2 lds_unaligned_rate.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U16 { u32x4 v; };
__global__ void __launch_bounds__(256) k(unsigned* out, long long* cyc, int shift_bytes, int lane_pitch) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const unsigned char* p = lds + (threadIdx.x & 63) * lane_pitch + shift_bytes;
  u32x4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const u32x4 v = reinterpret_cast<const U16*>(p + j * 1024 + (it & 1) * 16)->v;
      acc += v;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned* out; long long* cyc; long long h[1];
  hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 8);
  const int pitches[2] = {16, 880};
  for (int pi = 0; pi < 2; ++pi)
    for (int shift : {0, 2, 4, 6, 8, 12}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, cyc, shift, pitches[pi] == 880 ? 36 : 16);
      hipDeviceSynchronize();
      hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
      printf("lane pitch %3d B, shift %2d B: %.1f clock64 ticks per ds_read_b128 (4 waves sharing the LDS)\n",
             pitches[pi] == 880 ? 36 : 16, shift, (double)h[0] / (256.0 * 16.0));
    }
  return 0;
}
