// Do matrix (MFMA) and vector (VALU) instructions of two wavefronts that share a SIMD run at
// the same time?  512-thread workgroups (waves w and w+4 share a SIMD), one per CU.  Waves 0-3
// issue independent 32x32x16 bf16 MFMAs, waves 4-7 a VALU stream (fma or exp2); each arm is
// timed alone and together.  hipcc -O2 --offload-arch=gfx950 coexec.hip -o coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VKIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int do_mfma, int do_valu) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!do_mfma) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    if (s == 12345.f) out[threadIdx.x] = s;
  } else {
    if (!do_valu) return;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (VKIND == 0) x[i] = fmaf(x[i], 0.999f, 0.001f);
          else x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.f) out[threadIdx.x] = s;
  }
}

template <int VKIND>
float run(float* d, int iters, int m, int v) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<VKIND>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<VKIND>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  const int iters = 2000;  // per wave: 64000 MFMAs (x32 clk) or 512000 VALU ops
  printf("VALU = v_fma_f32:  mfma alone %.0f us, valu alone %.0f us, both %.0f us\n",
         run<0>(d, iters, 1, 0), run<0>(d, iters, 0, 1), run<0>(d, iters, 1, 1));
  printf("VALU = v_exp_f32+mul: mfma alone %.0f us, valu alone %.0f us, both %.0f us\n",
         run<1>(d, iters, 1, 0), run<1>(d, iters, 0, 1), run<1>(d, iters, 1, 1));
  return 0;
}
