// Internal helpers shared by the kernel translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/vrdone_hip.h"

namespace vrd {

void set_error(const char* fmt, ...);

// RAII profiling scope: when profiling is on, records a HIP event pair on `stream`
// around the launch(es) issued inside the scope.
struct ProfScope {
    int id;
    hipStream_t stream;
    void* slot;
    ProfScope(int kernel_id, hipStream_t s, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
};

#define VRD_CHECK_ARG(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            vrd::set_error(__VA_ARGS__);              \
            return -1;                                \
        }                                             \
    } while (0)

#define VRD_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            vrd::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                           hipGetErrorString(e_));                                \
            return -2;                                                            \
        }                                                                         \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

}  // namespace vrd
