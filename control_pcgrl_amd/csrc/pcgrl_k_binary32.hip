// pcgrl_k_binary32.hip -- translation unit: the BINARY kernels with 32-bit row masks (see pcgrl_dispatch.h)
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"


PCGRL_DEFINE_LAUNCH32(launch_binary32, PCGRL_PROB_BINARY)
