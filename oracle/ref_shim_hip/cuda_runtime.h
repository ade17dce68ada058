/* CUDA header name -> ROCm (device build of oracle/_ref only; see cuda_on_hip.h) */
#include "cuda_on_hip.h"
