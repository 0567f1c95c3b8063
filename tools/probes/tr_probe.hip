// Probe the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (authoring aid, not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = lane * 8;
  else if (mode == 1) addr = (lane & 15) * 128 + (lane >> 4) * 8;
  else if (mode == 2) addr = (lane & 3) * 128 + (lane >> 2) * 8;
  else addr = (lane & 15) * 8 + (lane >> 4) * 1024;
  addr += (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)lds);
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = v.x & 0xffff;
  out[lane * 4 + 1] = v.x >> 16;
  out[lane * 4 + 2] = v.y & 0xffff;
  out[lane * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipError_t e = hipMalloc(&d, 64 * 4 * 2);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("malloc %s dev %s arch %s\n", hipGetErrorString(e), pr.name, pr.gcnArchName);
  uint16_t h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("launch %s sync %s\n", hipGetErrorString(e), hipGetErrorString(e2));
    printf("mode %d (element index = byte/2):\n", mode);
    int show[] = {0,1,2,3,4,5,15,16,17,18,31,32,33,47,48,49,63};
    for (int l : show) printf(" %d:%d,%d,%d,%d", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    printf("\n");
  }
  return 0;
}
