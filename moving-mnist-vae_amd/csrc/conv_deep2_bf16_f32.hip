// bf16_f32 instances of the barrier-free deep-layer conv kernel (see conv_deep2.inc).
#define DEEP_TU 2
#include "conv_deep2.inc"
