#define PAYNE_TU_POST_FULL_A
// k_post_full_a.hip -- one compilation unit of libpayne_hip.so (kernels only; the C ABI is payne_hip.hip).
#include <hip/hip_runtime.h>

#include "../../include/payne_hip.h"
#include "post_seq.hpp"

using namespace payne;
#include "post_kernels.hpp"
PAYNE_POST_FULL_A_LIST(PAYNE_POST_DEFINE)
