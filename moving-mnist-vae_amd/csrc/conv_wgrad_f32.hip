// f32 (validation mode) instances of the patch-tile weight-gradient kernel (see conv_wgrad.inc).
#include "kernels.hpp"
#include "tile_common.hpp"
#include "conv_wgrad.inc"

namespace mmvae {
int launch_wgrad2_f32(const Wgrad2Args& a, dim3 grid, int ta16, int tb16, int maxg, hipStream_t s) {
  return launch_wgrad2_t<float>(a, DT_F32, grid, ta16, tb16, maxg, s);
}
}  // namespace mmvae
