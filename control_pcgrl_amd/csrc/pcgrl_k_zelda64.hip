// pcgrl_k_zelda64.hip -- translation unit: the ZELDA kernels with 64-bit row masks (see pcgrl_dispatch.h)
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"


PCGRL_DEFINE_LAUNCH64(launch_zelda64, PCGRL_PROB_ZELDA)
