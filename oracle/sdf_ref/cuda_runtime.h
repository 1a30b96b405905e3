// TEST INFRASTRUCTURE: stand-in for <cuda_runtime.h> -- thread / block indices, __global__ / __device__ and dim3 come from the
// HIP-on-CPU shim (tests/hipcpu/hip/hip_runtime.h); the error API used by the reference's host wrapper is stubbed.
#pragma once
#include <hip/hip_runtime.h>
typedef int cudaError_t;
#define cudaSuccess 0
static inline cudaError_t cudaGetLastError() { return 0; }
static inline const char* cudaGetErrorString(cudaError_t) { return ""; }
