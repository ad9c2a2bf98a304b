// f32 instances of the barrier-free deep-layer conv kernel (see conv_deep2.inc).
#define DEEP_TU 0
#include "conv_deep2.inc"
