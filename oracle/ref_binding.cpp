// Registers a handful of the reference's OWN CPU kernels (compiled in place from
// /root/reference/sgl-kernel/csrc/cpu by oracle/Makefile) as torch ops in the
// namespace `sgl_ref_cpu`, so the oracle's restatements can be checked against
// outputs of the reference implementation itself.  Test infrastructure only:
// never loaded by the product path, never shipped.
#include <ATen/ATen.h>
#include <torch/library.h>

#include <optional>
#include <tuple>
#include <vector>

// Declarations as in sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:27-45,113-172,301-307.
at::Tensor gelu_tanh_and_mul_cpu(const at::Tensor& input);
at::Tensor rmsnorm_cpu(at::Tensor& input, at::Tensor& weight, double eps);
void fused_add_rmsnorm_cpu(at::Tensor& input, at::Tensor& residual, at::Tensor& weight, double eps);
at::Tensor flash_attn_varlen_func(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v,
                                  const at::Tensor& cu_seqlens_q, const at::Tensor& cu_seqlens_k,
                                  int64_t max_seqlen_q, int64_t max_seqlen_k, bool causal);
at::Tensor convert_weight_packed(at::Tensor& weight);
at::Tensor fp8_scaled_mm_cpu(at::Tensor& mat1, at::Tensor& mat2, at::Tensor& scales2,
                             std::vector<int64_t> block_size, const std::optional<at::Tensor>& bias,
                             at::ScalarType out_dtype, bool is_vnni);
std::tuple<at::Tensor, at::Tensor> rotary_embedding_cpu(at::Tensor& positions, at::Tensor& query,
                                                        at::Tensor& key, int64_t head_size,
                                                        at::Tensor& cos_sin_cache, bool is_neox);

TORCH_LIBRARY(sgl_ref_cpu, m) {
  m.def("gelu_tanh_and_mul_cpu(Tensor input) -> Tensor");
  m.impl("gelu_tanh_and_mul_cpu", c10::kCPU, &gelu_tanh_and_mul_cpu);
  m.def("rmsnorm_cpu(Tensor input, Tensor weight, float eps) -> Tensor");
  m.impl("rmsnorm_cpu", c10::kCPU, &rmsnorm_cpu);
  m.def("fused_add_rmsnorm_cpu(Tensor(a!) input, Tensor(a!) residual, Tensor weight, float eps) -> ()");
  m.impl("fused_add_rmsnorm_cpu", c10::kCPU, &fused_add_rmsnorm_cpu);
  m.def(
      "flash_attn_varlen_func(Tensor q, Tensor k, Tensor v, Tensor cu_seqlens_q, Tensor cu_seqlens_k, "
      "int max_seqlen_q, int max_seqlen_k, bool causal) -> Tensor");
  m.impl("flash_attn_varlen_func", c10::kCPU, &flash_attn_varlen_func);
  m.def("convert_weight_packed(Tensor weight) -> Tensor");
  m.impl("convert_weight_packed", c10::kCPU, &convert_weight_packed);
  m.def(
      "fp8_scaled_mm_cpu(Tensor mat1, Tensor mat2, Tensor scales2, int[] block_size, Tensor? bias, "
      "ScalarType out_dtype, bool is_vnni) -> Tensor");
  m.impl("fp8_scaled_mm_cpu", c10::kCPU, &fp8_scaled_mm_cpu);
  m.def(
      "rotary_embedding_cpu(Tensor positions, Tensor query, Tensor key, int head_size, Tensor "
      "cos_sin_cache, bool is_neox) -> (Tensor, Tensor)");
  m.impl("rotary_embedding_cpu", c10::kCPU, &rotary_embedding_cpu);
}
