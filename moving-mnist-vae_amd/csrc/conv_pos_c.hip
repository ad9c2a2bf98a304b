// position-major implicit GEMM (conv_pos.inc): the 1x1 geometries + the dispatcher
#include "conv_pos.inc"
namespace mmvae {
#define POS_CASE(K_, S_, P_, UP_, HI_, HO_, CIN_, IG_) \
  if (K == K_ && S == S_ && P == P_ && up == (UP_ ? 1 : 0) && HI == HI_ && HO == HO_ && CIN == CIN_) return a ? pos_launch<K_, S_, P_, UP_, HI_, HO_, CIN_, IG_>(*a, s) : 1;
static int pos_conv_tu_c(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s) {
  if (K == 1 && S == 1) up = 0;                // a 1x1 stride-1 conv is its own transpose form
  POS_CASE(1, 1, 0, false, 2, 2, 128, 2)      // decoder.uplayer1.conv1 (+ data gradient)
  POS_CASE(1, 1, 0, false, 4, 4, 128, 1)      // decoder.uplayer2.conv1
  POS_CASE(1, 1, 0, false, 4, 4, 64, 1)       // its data gradient
  POS_CASE(1, 2, 0, false, 4, 2, 128, 2)      // encoder.layer4.downsample
  POS_CASE(1, 2, 0, true, 2, 4, 256, 2)       // their data gradients (accumulating launches: only phase (0, 0) has a tap)
  POS_CASE(1, 2, 0, true, 4, 8, 128, 1)
  return 0;
}
int pos_conv_tu_a(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s);
int pos_conv_tu_b(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s);

static int pos_dispatch(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs* a, hipStream_t s) {
  int rc = pos_conv_tu_a(K, S, P, up, HI, HO, CIN, a, s);
  if (rc == 0) rc = pos_conv_tu_b(K, S, P, up, HI, HO, CIN, a, s);
  if (rc == 0) rc = pos_conv_tu_c(K, S, P, up, HI, HO, CIN, a, s);
  return rc;
}
bool pos_conv_takes(int K, int S, int P, int up, int HI, int HO, int CIN) { return pos_dispatch(K, S, P, up, HI, HO, CIN, nullptr, nullptr) > 0; }
int launch_pos_conv(int K, int S, int P, int up, int HI, int HO, int CIN, const PosArgs& a, hipStream_t s) {
  return pos_dispatch(K, S, P, up, HI, HO, CIN, &a, s);
}
}  // namespace mmvae
