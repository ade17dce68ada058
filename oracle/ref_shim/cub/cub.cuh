/* part of the host-side CUDA stand-in used only to build oracle/_ref (see cuda_host_shim.h) */
#include "../cuda_host_shim.h"
