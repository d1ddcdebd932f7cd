// Host stand-in for <hip/hip_runtime.h> -- SANITIZER HARNESS ONLY (tests/tools/hostsim).
//
// Lets the lane-per-instance (v1) kernels of ilqr_kernels.hip and the C-ABI orchestration of ilqr_capi.cpp be compiled as plain host
// C++ with -fsanitize=address,undefined, so that index overruns, use-after-free across problem/context teardown and undefined
// behaviour in the device code show up on the CPU under a sanitizer instead of as a GPU fault.  A kernel launch becomes a serial loop
// over (block, thread); that is valid only for kernels whose lanes never communicate (no LDS, no shuffles, no barriers) -- the v1 set.
// This is NOT a CPU path of the product: it is never built by build(), never loaded by capi.load(), and the cooperative kernels
// cannot run on it at all (stubs.cpp aborts).
#pragma once
#include <math.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using std::isfinite;
using std::isnan;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
// LDS of the kernels built here is lane-private scratch (k_kp_derivs: slot [entry][threadIdx.x] for the rolled FK loop); the lanes of a
// launch run one after the other in this harness, so a static array is the same thing
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

inline const char* hipGetErrorString(hipError_t) { return "hostsim error"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)0x1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }  // ASan red zones around every "device" buffer
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }

namespace hostsim {
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid;
    blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++)
                for (unsigned tz = 0; tz < block.z; tz++)
                    for (unsigned ty = 0; ty < block.y; ty++)
                        for (unsigned tx = 0; tx < block.x; tx++) {
                            blockIdx = dim3(bx, by, bz);
                            threadIdx = dim3(tx, ty, tz);
                            kernel(args...);
                        }
}
}  // namespace hostsim
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hostsim::launch(kernel, grid, block, __VA_ARGS__)
