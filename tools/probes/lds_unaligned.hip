// Probe: does ds_read_b128 / ds_read_b64 at a 2-byte-aligned (not dword-aligned) LDS address return the right bytes on
// gfx950 under the ROCm runtime's alignment mode?  (hipcc --offload-arch=gfx950 lds_unaligned.hip -o probe && ./probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int shift) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const unsigned short* p = lds + threadIdx.x * 24 + shift;       // 48-byte lane pitch, `shift` elements of 2 bytes
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
  unsigned short r[8];
  memcpy(r, &v, 16);
  for (int e = 0; e < 8; ++e) out[threadIdx.x * 8 + e] = r[e];
}
int main() {
  unsigned short h[4096], *din, *dout, o[512];
  for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(dout, 0xff, sizeof(o));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, shift);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 8; ++j) bad += o[t * 8 + j] != (unsigned short)(t * 24 + shift + j);
    printf("shift %d elements (%d bytes): err=%d mismatches=%d  lane1: %u %u %u ... %u\n", shift, 2 * shift, (int)e, bad,
           o[8], o[9], o[10], o[15]);
  }
  return 0;
}
