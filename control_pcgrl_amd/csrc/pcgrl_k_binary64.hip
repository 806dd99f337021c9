// pcgrl_k_binary64.hip -- translation unit: the BINARY kernels with 64-bit row masks (see pcgrl_dispatch.h)
#define PCGRL_KERNEL_TU
#include "pcgrl_dispatch.h"


PCGRL_DEFINE_LAUNCH64(launch_binary64, PCGRL_PROB_BINARY)
