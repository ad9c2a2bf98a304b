// position-major implicit GEMM (conv_pos.inc): the ConvTranspose2d geometries (k4 s2 p1, k2 s2 p0) and their data gradients
#include "conv_pos.inc"
namespace mmvae {
#define POS_CASE(K_, S_, P_, UP_, HI_, HO_, CIN_, IG_) \
  if (K == K_ && S == S_ && P == P_ && up == (UP_ ? 1 : 0) && HI == HI_ && HO == HO_ && CIN == CIN_) return a ? pos_launch<K_, S_, P_, UP_, HI_, HO_, CIN_, IG_>(*a, s) : 1;
int pos_conv_tu_b(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s) {
  POS_CASE(4, 2, 1, true, 2, 4, 128, 2)       // decoder.uplayer1.conv2 / .upsample
  POS_CASE(4, 2, 1, true, 4, 8, 64, 1)        // decoder.uplayer2.conv2
  POS_CASE(4, 2, 1, true, 4, 8, 128, 1)       // decoder.uplayer2.upsample
  POS_CASE(4, 2, 1, false, 4, 2, 128, 2)      // data gradients: uplayer1
  POS_CASE(4, 2, 1, false, 8, 4, 64, 1)       // uplayer2
  POS_CASE(2, 2, 0, true, 1, 2, 128, 4)       // decoder.conv1 (z = 128)
  POS_CASE(2, 2, 0, true, 1, 2, 512, 4)       // (z = 512)
  POS_CASE(2, 2, 0, false, 2, 1, 128, 4)      // its data gradient
  return 0;
}
}  // namespace mmvae
