#define PAYNE_TU_POST_LEAN
// k_post_lean.hip -- one compilation unit of libpayne_hip.so (kernels only; the C ABI is payne_hip.hip).
#include <hip/hip_runtime.h>

#include "../../include/payne_hip.h"
#include "post_seq.hpp"

using namespace payne;
#include "post_kernels.hpp"
PAYNE_POST_LEAN_LIST(PAYNE_POST_DEFINE)
