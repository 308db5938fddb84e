// What shader clock does the chip hold when every SIMD runs MFMAs back to back, and how does
// it move when vector ALU work and LDS reads are added?  A full-chip grid (2 waves per SIMD)
// runs two independent MFMA accumulators per wave with K v_fma_f32 and L ds_read_b128 per
// MFMA; s_memtime (shader clocks) is compared with s_memrealtime (constant 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_under_load.hip -o /tmp/clk && /tmp/clk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int L>
__global__ __launch_bounds__(512) void load_kernel(float* out, long long* cyc, long long* real, int iters) {
  __shared__ f32x4 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)i;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * ((threadIdx.x & 63) + e)); b[e] = (_Float16)(0.002f * e); }
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = 0.5f + 0.01f * k + 0.001f * threadIdx.x;
  const float c1 = 0.999f, c2 = 0.0001f;
  f32x4 ld[4];
  for (int l = 0; l < 4; ++l) ld[l] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned addr = (unsigned)(threadIdx.x * 16);
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
#pragma unroll
    for (int l = 0; l < L; ++l) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[l]) : "v"(addr), "n"(l * 8192));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(c1), "v"(c2));
#pragma unroll
    for (int l = 0; l < L; ++l) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[2 + l]) : "v"(addr), "n"(l * 8192 + 16384));
    if (L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
  for (int k = 0; k < 8; ++k) s += x[k];
  for (int l = 0; l < 4; ++l) s += ld[l][0] + ld[l][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

template <int K, int L>
void run(int blocks, int iters) {
  float* out; long long *cyc, *real;
  hipMalloc(&out, (size_t)blocks * 512 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(long long)); hipMalloc(&real, blocks * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((load_kernel<K, L>), dim3(blocks), dim3(512), 0, 0, out, cyc, real, iters / 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL((load_kernel<K, L>), dim3(blocks), dim3(512), 0, 0, out, cyc, real, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c, r; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r, real, 8, hipMemcpyDeviceToHost);
  const double mhz = (double)c / ((double)r / 100.0);
  const double tf = 2.0 * iters * 32768.0 * 8 * blocks / (ms * 1e-3) / 1e12;
  printf("blocks %4d  fma/MFMA %d  ds_read_b128/MFMA %d : %7.2f ms  %6.1f shader clk per MFMA per SIMD  %7.1f MHz (s_memtime/s_memrealtime)  %6.1f TF/s\n",
         blocks, K, L, ms, (double)c / iters / 2 / 2, mhz, tf);
  hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
  const int iters = 40000;
  for (int blocks : {1, 256}) {
    run<0, 0>(blocks, iters);
    run<4, 0>(blocks, iters);
    run<0, 2>(blocks, iters);
    run<4, 2>(blocks, iters);
    run<6, 2>(blocks, iters);
  }
  return 0;
}
