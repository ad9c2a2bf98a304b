// bf16 instances of the patch-tile weight-gradient kernel (see conv_wgrad.inc).
#include "kernels.hpp"
#include "tile_common.hpp"
#include "conv_wgrad.inc"

namespace mmvae {
int launch_wgrad2_bf16(const Wgrad2Args& a, dim3 grid, int ta16, int tb16, int maxg, hipStream_t s) {
  return launch_wgrad2_t<bf16_t>(a, DT_BF16, grid, ta16, tb16, maxg, s);
}
}  // namespace mmvae
