/* cooperative_groups/reduce.h is included by the reference but nothing of it is used */
#include "../cuda_on_hip.h"
