// In-wave co-execution: one wave per SIMD runs, per iteration, 4 independent 32x32x16 MFMAs and
// NV independent VALU ops placed behind them.  If the VALU ops run in the MFMAs' shadow the
// time stays at the MFMA-only time until the VALU work exceeds it.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NV / 4; ++u) {
        const int j = (i * (NV / 4) + u) & 7;
        if (KIND == 0) x[j] = fmaf(x[j], 0.999f, 0.001f);
        else x[j] = __builtin_amdgcn_exp2f(x[j]);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NV, int KIND, int WAVES>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, KIND, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, KIND, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d; hipMalloc(&d, 4096);
  const int it = 16000;  // 64000 MFMAs per wave
  printf("1 wave/SIMD, fma : NV=0 %.0f  NV=8 %.0f  NV=16 %.0f  NV=24 %.0f  NV=32 %.0f  NV=48 %.0f us\n",
         run<0, 0, 4>(d, it), run<8, 0, 4>(d, it), run<16, 0, 4>(d, it), run<24, 0, 4>(d, it),
         run<32, 0, 4>(d, it), run<48, 0, 4>(d, it));
  printf("1 wave/SIMD, exp2: NV=0 %.0f  NV=4 %.0f  NV=8 %.0f  NV=12 %.0f us\n",
         run<0, 1, 4>(d, it), run<4, 1, 4>(d, it), run<8, 1, 4>(d, it), run<12, 1, 4>(d, it));
  printf("2 waves/SIMD (each half the iterations), fma: NV=0 %.0f  NV=16 %.0f  NV=32 %.0f us\n",
         run<0, 0, 8>(d, it / 2), run<16, 0, 8>(d, it / 2), run<32, 0, 8>(d, it / 2));
  return 0;
}
