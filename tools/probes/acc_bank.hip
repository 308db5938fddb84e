// Does an MFMA whose accumulator lives in the vector registers (VGPR form) keep independent
// vector instructions of the same wave / of the SIMD's other wave from issuing, and does the
// AccVGPR form behave differently?  One MFMA + K independent v_fma_f32 (or v_exp_f32) per loop
// iteration, timed with s_memtime; 256 threads = one wave per SIMD, 512 = two.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/acc_bank.hip -o /tmp/acc_bank && /tmp/acc_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool AG, int K, bool EXP>
__global__ void probe(float* out, long long* cyc, int iters) {
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = 0.5f + 0.01f * k + 0.001f * threadIdx.x;
  const float c1 = 0.999f, c2 = 0.0001f;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[e];
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// two INDEPENDENT accumulators: MFMA(acc0), K fillers, MFMA(acc1), K fillers per iteration
template <int K, bool EXP>
__global__ void probe2(float* out, long long* cyc, int iters) {
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = 0.5f + 0.01f * k + 0.001f * threadIdx.x;
  const float c1 = 0.999f, c2 = 0.0001f;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
    }
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K, bool EXP>
void run2(int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(long long));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe2<K, EXP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("two independent accumulators threads %3d  K=%d %s: %.1f cycles per MFMA (slowest wave)\n", threads, K,
         EXP ? "v_exp" : "v_fma", (double)mx / iters / 2);
  hipFree(out); hipFree(cyc);
}

template <bool AG, int K, bool EXP>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(long long));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<AG, K, EXP>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("%-28s threads %3d  K=%d %s: %.1f cycles per iteration (slowest wave)\n", name, threads, K,
         EXP ? "v_exp" : "v_fma", (double)mx / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<false, 0, false>("acc in VGPRs", threads);
    run<true, 0, false>("acc in AccVGPRs", threads);
    run<false, 4, false>("acc in VGPRs", threads);
    run<true, 4, false>("acc in AccVGPRs", threads);
    run<false, 6, false>("acc in VGPRs", threads);
    run<true, 6, false>("acc in AccVGPRs", threads);
    run<false, 8, false>("acc in VGPRs", threads);
    run<true, 8, false>("acc in AccVGPRs", threads);
    run<false, 2, true>("acc in VGPRs", threads);
    run<true, 2, true>("acc in AccVGPRs", threads);
    run<false, 4, true>("acc in VGPRs", threads);
    run<true, 4, true>("acc in AccVGPRs", threads);
  }
  for (int threads : {256, 512}) {
    run2<0, false>(threads);
    run2<2, false>(threads);
    run2<4, false>(threads);
    run2<5, false>(threads);
    run2<6, false>(threads);
    run2<8, false>(threads);
    run2<1, true>(threads);
    run2<2, true>(threads);
    run2<3, true>(threads);
  }
  return 0;
}
