// A caller of libmi355dit.so that knows nothing about Python or torch: plain HIP runtime calls
// for memory, the C ABI of include/mi355_dit.h for the work, a CPU loop as the check.
//   1. mi355_fp8_scaled_mm on small-integer data (every product and sum exact in fp32)
//   2. mi355_per_token_quant_fp8 -> mi355_fp8_scaled_mm, the W8A8 linear, against a CPU loop
//   3. mi355_cfg_combine + mi355_euler_step, the scheduler side of a denoise step
//   4. error reporting: a bad argument returns MI355_EINVAL and sets mi355_last_error()
// Exit code 0 = all checks passed.  Built by `make -C sglang-diffusion_amd/csrc smoke`.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mi355_dit.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define ABI_OK(x)                                                              \
  do {                                                                         \
    int rc_ = (x);                                                             \
    if (rc_ != MI355_OK) {                                                     \
      std::printf("%s -> %d: %s\n", #x, rc_, mi355_last_error());              \
      return 3;                                                                \
    }                                                                          \
  } while (0)

static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf(float f) {  // round to nearest even
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// e4m3fn decode (no infinities; 0x7f / 0xff are NaN and never produced here)
static float fp8_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  const float v = e == 0 ? std::ldexp((float)m / 8.f, -6) : std::ldexp(1.f + (float)m / 8.f, e - 7);
  return s ? -v : v;
}
static uint8_t small_int_fp8(int v) {  // -2 .. 2, exactly representable
  static const uint8_t tab[5] = {0xC0, 0xB8, 0x00, 0x38, 0x40};
  return tab[v + 2];
}
template <class T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  std::printf("libmi355dit ABI %d\n", mi355_abi_version());
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };

  // ---- 1. exact integer GEMM ------------------------------------------------------------
  const int M = 300, N = 264, K = 256;
  std::vector<uint8_t> a(M * K), w(N * K);
  std::vector<int> ai(M * K), wi(N * K);
  for (int i = 0; i < M * K; ++i) { ai[i] = (int)(rnd() % 5) - 2; a[i] = small_int_fp8(ai[i]); }
  for (int i = 0; i < N * K; ++i) { wi[i] = (int)(rnd() % 5) - 2; w[i] = small_int_fp8(wi[i]); }
  std::vector<float> ones_m(M, 1.f), ones_n(N, 1.f);
  uint8_t *da = to_device(a), *dw = to_device(w);
  float *dsa = to_device(ones_m), *dsb = to_device(ones_n);
  uint16_t* dout = nullptr;
  HIP_OK(hipMalloc((void**)&dout, (size_t)M * N * 2));
  if (!da || !dw || !dsa || !dsb) return 2;
  ABI_OK(mi355_fp8_scaled_mm(da, dw, dsa, dsb, nullptr, dout, M, N, K, K, K, N, nullptr));
  std::vector<uint16_t> out(M * N);
  HIP_OK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      int s = 0;
      for (int k = 0; k < K; ++k) s += ai[m * K + k] * wi[n * K + k];
      if (out[m * N + n] != f2bf((float)s)) ++bad;
    }
  std::printf("1. integer fp8 GEMM %dx%dx%d: %d mismatches\n", M, N, K, bad);
  if (bad) return 1;

  // ---- 2. per-token quant + GEMM = the W8A8 linear --------------------------------------
  std::vector<uint16_t> x(M * K);
  for (auto& v : x) v = f2bf(((float)(rnd() % 2001) - 1000.f) / 400.f);
  std::vector<float> sbv(N);
  for (auto& v : sbv) v = 0.001f + (float)(rnd() % 100) / 5000.f;
  uint16_t* dx = to_device(x);
  float* dsb2 = to_device(sbv);
  uint8_t* dq = nullptr;
  float* dxs = nullptr;
  HIP_OK(hipMalloc((void**)&dq, (size_t)M * K));
  HIP_OK(hipMalloc((void**)&dxs, (size_t)M * 4));
  if (!dx || !dsb2) return 2;
  ABI_OK(mi355_per_token_quant_fp8(dx, M, K, K, dq, dxs, nullptr));
  ABI_OK(mi355_fp8_scaled_mm(dq, dw, dxs, dsb2, nullptr, dout, M, N, K, K, K, N, nullptr));
  std::vector<uint8_t> q(M * K);
  std::vector<float> xs(M);
  HIP_OK(hipMemcpy(q.data(), dq, q.size(), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(xs.data(), dxs, xs.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  int bad_scale = 0;
  for (int m = 0; m < M; ++m) {
    float amax = 0.f;
    for (int k = 0; k < K; ++k) amax = std::fmax(amax, std::fabs(bf2f(x[m * K + k])));
    if (xs[m] != amax / 448.f) ++bad_scale;
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)fp8_decode(q[m * K + k]) * (double)wi[n * K + k];
      const double ref = s * xs[m] * sbv[n], got = bf2f(out[m * N + n]);
      num += (got - ref) * (got - ref);
      den += ref * ref;
    }
  }
  const double rel = std::sqrt(num / den);
  std::printf("2. W8A8 linear: %d wrong token scales, relative L2 vs CPU loop %.2e\n", bad_scale, rel);
  if (bad_scale || rel > 3e-3) return 1;  // bf16 output rounding alone is ~1.7e-3

  // ---- 3. guidance combine + Euler step ------------------------------------------------
  const int64_t n_el = 4096 * 64;
  std::vector<uint16_t> c(n_el), u(n_el), lat(n_el);
  for (int64_t i = 0; i < n_el; ++i) {
    c[i] = f2bf(((float)(rnd() % 2001) - 1000.f) / 500.f);
    u[i] = f2bf(((float)(rnd() % 2001) - 1000.f) / 500.f);
    lat[i] = f2bf(((float)(rnd() % 2001) - 1000.f) / 300.f);
  }
  uint16_t *dc = to_device(c), *du = to_device(u), *dlat = to_device(lat), *dnoise = nullptr;
  HIP_OK(hipMalloc((void**)&dnoise, n_el * 2));
  if (!dc || !du || !dlat) return 2;
  const float scale = 3.5f, sigma = 0.75f, sigma_next = 0.7f;
  ABI_OK(mi355_cfg_combine(dnoise, dc, du, scale, n_el, nullptr));
  ABI_OK(mi355_euler_step(dlat, dlat, dnoise, sigma, sigma_next, n_el, nullptr));
  std::vector<uint16_t> got(n_el);
  HIP_OK(hipMemcpy(got.data(), dlat, n_el * 2, hipMemcpyDeviceToHost));
  bad = 0;
  const float dt = bf2f(f2bf(sigma_next - sigma));
  for (int64_t i = 0; i < n_el; ++i) {
    const float cu = bf2f(f2bf(bf2f(c[i]) - bf2f(u[i])));
    const float noise = bf2f(f2bf(bf2f(u[i]) + bf2f(f2bf(scale * cu))));
    const uint16_t ref = f2bf(bf2f(lat[i]) + bf2f(f2bf(dt * noise)));
    if (got[i] != ref) ++bad;
  }
  std::printf("3. cfg_combine + euler_step: %d mismatches of %ld\n", bad, (long)n_el);
  if (bad) return 1;

  // ---- 4. errors are codes + a message, never a crash --------------------------------
  const int rc = mi355_fp8_scaled_mm(da, dw, dsa, dsb, nullptr, dout, M, N, 100, 100, 100, N, nullptr);
  std::printf("4. K = 100 -> %d (%s)\n", rc, mi355_last_error());
  if (rc != MI355_EINVAL || std::strlen(mi355_last_error()) == 0) return 1;
  HIP_OK(hipDeviceSynchronize());
  std::printf("C ABI smoke: OK\n");
  return 0;
}
