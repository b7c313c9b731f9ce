// adj-mode front end (MultipleEmbedding, Modules.py:125-201): placeholder until the grouped gather-GEMM lands.
#include "kernels.hpp"

namespace matcha {

size_t adj_workspace_bytes(const matcha_shape& s, int64_t T) { (void)s; (void)T; return 0; }

int adj_forward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o,
                const int64_t* x, int64_t T, float* node_out, float* recon_out, void* ws, size_t ws_bytes, hipStream_t st) {
  set_error("adj mode is not built into this library yet");
  return MATCHA_EINVAL;
}

int adj_backward(const matcha_shape& s, const matcha_tensors& p, const matcha_frozen& f, const matcha_step_opts& o,
                 const int64_t* x, int64_t T, const float* dnode, const float* drecon, matcha_tensors& g, int32_t* touched,
                 void* ws, size_t ws_bytes, hipStream_t st) {
  set_error("adj mode is not built into this library yet");
  return MATCHA_EINVAL;
}

}  // namespace matcha
