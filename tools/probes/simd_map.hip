// Which SIMD does each wavefront of a workgroup land on?  Prints wave -> (cu, simd) for one
// 512-thread and one 256-thread workgroup.  hipcc --offload-arch=gfx950 simd_map.hip -o simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hwid;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4096);
  for (int threads : {512, 256}) {
    hipMemset(d, 0, 4096);
    hipLaunchKernelGGL(probe, dim3(4), dim3(threads), 0, 0, d);
    unsigned h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
      printf("threads=%d block %d:", threads, b);
      for (int w = 0; w < threads / 64; ++w) {
        const unsigned v = h[b * 16 + w];
        printf("  w%d:cu%u.simd%u.slot%u", w, (v >> 8) & 15, (v >> 4) & 3, v & 15);
      }
      printf("\n");
    }
  }
  return 0;
}
