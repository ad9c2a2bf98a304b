// position-major implicit GEMM (conv_pos.inc): the 3x3 geometries
#include "conv_pos.inc"
namespace mmvae {
#define POS_CASE(K_, S_, P_, UP_, HI_, HO_, CIN_, IG_) \
  if (K == K_ && S == S_ && P == P_ && up == (UP_ ? 1 : 0) && HI == HI_ && HO == HO_ && CIN == CIN_) return a ? pos_launch<K_, S_, P_, UP_, HI_, HO_, CIN_, IG_>(*a, s) : 1;
int pos_conv_tu_a(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s) {
  POS_CASE(3, 1, 1, false, 4, 4, 128, 1)      // encoder.layer3.conv2 (and the deeper variant's 128-channel identity blocks)
  POS_CASE(3, 1, 1, false, 2, 2, 256, 2)      // encoder.layer4.conv2
  POS_CASE(3, 1, 1, false, 2, 2, 128, 2)      // deeper variant: decoder.uplayer1's identity block
  POS_CASE(3, 1, 1, true, 4, 4, 128, 1)       // their data gradients
  POS_CASE(3, 1, 1, true, 2, 2, 256, 2)
  POS_CASE(3, 1, 1, true, 2, 2, 128, 2)
  POS_CASE(3, 2, 1, false, 4, 2, 128, 2)      // encoder.layer4.conv1
  // (encoder.layer3.conv1, 8x8x64 -> 4x4: a 16-image tile is 131 KB of LDS, one block per CU -- deep2_conv_kernel is faster: 28 vs 31 us)
  POS_CASE(3, 2, 1, true, 2, 4, 256, 2)       // their data gradients
  POS_CASE(3, 2, 1, true, 4, 8, 128, 1)
  return 0;
}
}  // namespace mmvae
