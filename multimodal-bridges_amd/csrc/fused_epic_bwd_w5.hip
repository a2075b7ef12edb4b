// The reverse kernels of fused_epic_bwd.hip compiled for FIVE waves per jet (N <= 160); entry points
// mmb_fused_{epic,uni}_train_backward_w5, reached through the plain entry points when N > 128.
#define MMB_FE_WAVES 5
#include "fused_epic_bwd.hip"
