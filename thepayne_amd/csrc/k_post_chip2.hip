#define PAYNE_TU_CHIP2
// k_post_chip2.hip -- one compilation unit of libpayne_hip.so (kernels only; the C ABI is payne_hip.hip): payne_post_chip2_kernel.
#include <hip/hip_runtime.h>

#include "../../include/payne_hip.h"
#include "post_seq.hpp"

using namespace payne;
#include "post_kernels.hpp"
