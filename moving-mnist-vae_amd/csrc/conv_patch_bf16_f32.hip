// bf16_f32 instances of the pipelined patch-tile conv kernel (see conv_patch.inc).
#define PATCH_TU 2
#include "conv_patch.inc"
