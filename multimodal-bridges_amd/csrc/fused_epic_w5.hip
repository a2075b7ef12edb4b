// The forward / generation kernels of fused_epic.hip compiled for FIVE waves per jet: up to 160 particles (AOJ's
// 150-particle clouds), one workgroup per CU.  Entry points mmb_fused_{epic,uni}_{forward,generate}_w5, reached
// through the plain entry points when N > 128.
#define MMB_FE_WAVES 5
#include "fused_epic.hip"
