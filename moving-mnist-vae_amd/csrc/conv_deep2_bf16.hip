// bf16 instances of the barrier-free deep-layer conv kernel (see conv_deep2.inc).
#define DEEP_TU 1
#include "conv_deep2.inc"
