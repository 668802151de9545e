/* Stand-in for <hip/hip_runtime.h> used ONLY by tests/native/kernel_on_cpu.cpp (TEST INFRASTRUCTURE):
 * it lets ra_amd/csrc/rgb_kernels.hip compile as plain x86 C++ so that the per-lane transition code
 * (process_message<N, KIND>, the pack/unpack kernels) can be executed lane by lane on the CPU and compared
 * with the checker without a GPU.  Whole kernels run too: emu::launch makes every lane of a block a fiber,
 * so __syncthreads, __shfl and __shared__ arrays behave as on one workgroup (blocks run one after another). */
#ifndef RGB_FAKE_HIP_RUNTIME_H
#define RGB_FAKE_HIP_RUNTIME_H
#include <stdint.h>
#include <string.h>
#include <chrono>

#define __host__
#define __device__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 v; v.x = x; v.y = y; return v; }
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
/* Block emulation (tests/native/kernel_on_cpu.cpp): every lane of a block is a fiber; __syncthreads() hands
 * control back to the scheduler until all live lanes of the block have arrived, __shfl exchanges through a
 * per-block buffer between two such barriers.  Outside emu::launch both are inert. */
namespace emu {
void barrier();
void shfl(void *value, size_t size, int src_lane);
int lane();
unsigned long long ballot(int pred);
}
static inline void __syncthreads() { emu::barrier(); }
template <typename T> static inline T __shfl(T v, int src, int = 64) { emu::shfl(&v, sizeof v, src); return v; }
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred); }
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { emu::shfl(&v, sizeof v, emu::lane() ^ mask); return v; }
static inline unsigned long long wall_clock64() {
  return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
}
template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

typedef void *hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
#include <functional>
namespace emu { void launch(dim3 grid, dim3 block, const std::function<void()> &body); }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  do { (void)(stream); emu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); }); } while (0)
/* "device memory" is host memory, copies are memcpy, streams are synchronous */
#include <stdlib.h>
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; ++r) memcpy((char *)d + r * dpitch, (const char *)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
#define hipHostMallocDefault 0
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
typedef void *hipEvent_t;
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
/* v_dot4_u32_u8 */
static inline unsigned emu_udot4(unsigned a, unsigned b, unsigned c) {
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
  return c;
}
#define __builtin_amdgcn_udot4(a, b, c, clamp) emu_udot4((a), (b), (c))
#endif
