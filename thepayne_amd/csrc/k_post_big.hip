#define PAYNE_TU_BIG
// k_post_big.hip -- one compilation unit of libpayne_hip.so (kernels only; the C ABI is payne_hip.hip).
#include <hip/hip_runtime.h>

#include "../../include/payne_hip.h"
#include "post_seq.hpp"

using namespace payne;
#include "post_kernels.hpp"
