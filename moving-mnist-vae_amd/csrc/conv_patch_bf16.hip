// bf16 instances of the pipelined patch-tile conv kernel (see conv_patch.inc).
#define PATCH_TU 1
#include "conv_patch.inc"
