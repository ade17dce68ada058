/*
 * cuda_on_hip.h -- the CUDA spellings the reference's .cu files use, mapped onto the ROCm headers, so that hipcc can
 * compile the reference's OWN rasterizer and simple-knn sources for gfx950 (oracle/build_ref.py build_device()).
 *
 * TEST INFRASTRUCTURE ONLY: oracle/_ref/libref_raster_gfx950.so is a checker and a reported baseline ("the reference's
 * own kernels on this MI355X", bench.py reference_on_device) -- the product never includes, links or loads it, and nothing
 * of this style (CUDA names over HIP) exists in luciddreamer_amd/.
 */
#ifndef LUCID_REF_CUDA_ON_HIP_H
#define LUCID_REF_CUDA_ON_HIP_H
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdint>
#include <cstdio>
#include <cfloat>
#include <iostream>
#include <stdexcept>

#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetErrorString hipGetErrorString
/* device trap of auxiliary.h:158-159 */
#ifndef __trap
#define __trap() __builtin_trap()
#endif
#endif
