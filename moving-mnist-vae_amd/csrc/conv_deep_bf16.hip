// bf16 instances of the deep-layer implicit-GEMM conv kernel (see conv_deep.inc).
#define DEEP_TU 1
#include "conv_deep.inc"
