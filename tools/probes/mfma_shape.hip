// FLOP/s of bare MFMA loops by instruction shape (random operands held in registers, one or
// two waves per SIMD, 4 independent accumulators): the chip lowers its clock under load, and
// the clock it holds can depend on the MFMA shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void loop(const int* seed, float* out, int iters) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  i32x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = seed[(t * 8 + e) & 0xffff];
    b[e] = seed[(t * 8 + e + 4096) & 0xffff];
  }
  float s = 0.f;
  if (KIND == 0 || KIND == 2) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0)
          acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)),
                                                          __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3)), acc[i], 0, 0, 0);
      }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  } else {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 1)
          acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)),
                                                          __builtin_bit_cast(bf16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3)), acc[i], 0, 0, 0);
      }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  }
  if (s == 12345.f) out[t] = s;
}

template <int KIND>
void run(const char* name, double flop_per_mfma, const int* seed, float* out, int waves) {
  const int iters = 40000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<KIND>, dim3(256), dim3(waves * 64), 0, 0, seed, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = flop_per_mfma * 4.0 * iters * waves * 256;
  printf("%-28s %d waves/CU: %8.2f ms  %7.1f TF/s\n", name, waves, ms, flops / ms / 1e9);
}

int main() {
  int* seed; float* out;
  hipMalloc(&seed, 65536 * 4); hipMalloc(&out, 1 << 20);
  int* h = (int*)malloc(65536 * 4);
  srand(1);
  for (int i = 0; i < 65536; ++i) {
    // random bytes that are finite in both e4m3 and bf16 views
    unsigned v = 0;
    for (int b = 0; b < 4; ++b) v |= (unsigned)((rand() & 0x77) | ((rand() & 1) << 7)) << (8 * b);
    h[i] = (int)v;
  }
  hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
  for (int waves : {4, 8}) {
    run<0>("fp8 32x32x64", 2.0 * 32 * 32 * 64, seed, out, waves);
    run<1>("fp8 16x16x128", 2.0 * 16 * 16 * 128, seed, out, waves);
    run<2>("bf16 32x32x16", 2.0 * 32 * 32 * 16, seed, out, waves);
    run<3>("bf16 16x16x32", 2.0 * 16 * 16 * 32, seed, out, waves);
  }
  return 0;
}
