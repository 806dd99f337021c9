// pcgrl_k_zelda32.hip -- translation unit: the ZELDA kernels with 32-bit row masks (see pcgrl_dispatch.h)
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"


PCGRL_DEFINE_LAUNCH32(launch_zelda32, PCGRL_PROB_ZELDA)
