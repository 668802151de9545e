// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_* counters on the request shapes of the train kernel
// (MI355X_MICROARCH.md, "HBM": only 16-B/lane coalesced streaming reads are calibrated, x2; "calibrate on a known byte
// count in your own access pattern").  Every kernel moves a KNOWN number of bytes in ONE shape over a working set far
// beyond the 32 MiB of L2 (rows picked by a multiplicative permutation, so no row is touched twice in a launch):
//   rd128  8 lanes x 16 B of one 128-byte row, LDS-DMA, sc1      (the hot / peers / run-table rows of a train)
//   rd64   4 lanes x 16 B of a 64-byte record, LDS-DMA, nt        (message records)
//   rd32   2 lanes x 16 B, the first half of a 64-byte record, nt (half records)
//   rd1    one byte per lane, 64 contiguous bytes per wavefront    (the sequence-byte poll)
//   wr16   one 16-byte piece of a row per lane (plain store)       (state write-back, one piece)
//   wr32   two neighbouring lanes write one aligned 32-byte unit   (pair_store)
//   wr64nt four lanes write a whole 64-byte record, non-temporal   (decisions)
//   wr32nt two lanes write the first half of a 64-byte record, nt  (compact decisions)
//   wr1    one byte per lane, scattered over a packed byte array   (the sequence-byte publish)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/traffic_calib.hip -o gpurun_out/traffic_calib
// Run:   traffic_calib <kernel name> [rows]   (prints the bytes the launch moved; one launch per process run, so a
//        rocprofv3 --pmc pass of the process has exactly one dispatch of the named kernel besides the fill)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at line %d\n", (int)e, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u32 perm(u32 i, u32 n_pow2) { return (i * 2654435761u + 12345u) & (n_pow2 - 1u); }   // odd multiplier: a bijection mod 2^k

template <int POLICY>
__device__ __forceinline__ void glds16(const void *g, void *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, POLICY);
}
__device__ __forceinline__ void store16_nt(void *p, ulonglong2 v) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u d; d.x = (unsigned)v.x; d.y = (unsigned)(v.x >> 32); d.z = (unsigned)v.y; d.w = (unsigned)(v.y >> 32);
  __builtin_nontemporal_store(d, reinterpret_cast<v4u *>(p));
}

// LANES lanes x 16 B per unit of STRIDE bytes; a wavefront instruction covers 64 / LANES units
template <int LANES, int STRIDE, int POLICY>
__global__ __launch_bounds__(64) void rd_kernel(const char *base, u32 n_units, u32 n_pow2, u64 *sink) {
  __shared__ ulonglong2 io[64 * 8];
  const u32 lane = threadIdx.x;
  constexpr u32 UPI = 64 / LANES;                  // units per instruction
  const u32 first = blockIdx.x * UPI * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 u = first + k * UPI + lane / LANES;
    const u32 r = perm(u < n_units ? u : 0, n_pow2);
    glds16<POLICY>(base + (size_t)r * STRIDE + (lane % LANES) * 16, io + k * 64);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
  __syncthreads();
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += io[k * 64 + lane].x;
  if (acc == 0x1234567ull) sink[0] = acc;
}
// the train's state access: a 128-byte row is gathered (LDS-DMA, sc1), then WB_LANES x 16 bytes of it are written back
// (plain stores, the line is in the L2): what do partial write-backs do to the read side's latency?
template <int WB_LANES>
__global__ __launch_bounds__(64) void rw_kernel(char *base, u32 n_units, u32 n_pow2, u64 *sink) {
  __shared__ ulonglong2 io[64 * 8];
  const u32 lane = threadIdx.x;
  const u32 first = blockIdx.x * 64;
  u32 rows[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 u = first + k * 8 + lane / 8;
    rows[k] = perm(u < n_units ? u : 0, n_pow2);
    glds16<16>(base + (size_t)rows[k] * 128 + (lane % 8) * 16, io + k * 64);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("" ::: "memory");
  __syncthreads();
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const ulonglong2 v = io[k * 64 + lane];
    acc += v.x;
    if ((int)(lane % 8) < WB_LANES)
      *reinterpret_cast<ulonglong2 *>(base + (size_t)rows[k] * 128 + (lane % 8) * 16) = make_ulonglong2(v.x + 1, v.y);
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}
__global__ __launch_bounds__(64) void rd1_kernel(const unsigned char *base, u32 n_units, u32 n_pow2, u64 *sink) {
  // unit = 64 contiguous bytes read by one wavefront instruction (agent-scope byte loads, like the poll)
  const u32 lane = threadIdx.x;
  u32 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 u = blockIdx.x * 8 + k;
    const u32 r = perm(u < n_units ? u : 0, n_pow2);
    acc += __hip_atomic_load(base + (size_t)r * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 0x12345u) sink[0] = acc;
}
// LANES lanes x 16 B written per unit of STRIDE bytes
template <int LANES, int STRIDE, bool NT>
__global__ __launch_bounds__(64) void wr_kernel(char *base, u32 n_units, u32 n_pow2) {
  const u32 lane = threadIdx.x;
  constexpr u32 UPI = 64 / LANES;
  const u32 first = blockIdx.x * UPI * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 u = first + k * UPI + lane / LANES;
    if (u >= n_units) continue;
    const u32 r = perm(u, n_pow2);
    ulonglong2 *p = reinterpret_cast<ulonglong2 *>(base + (size_t)r * STRIDE + (lane % LANES) * 16);
    const ulonglong2 v = make_ulonglong2(u, lane);
    if (NT) store16_nt(p, v); else *p = v;
  }
}
__global__ __launch_bounds__(64) void wr1_kernel(unsigned char *base, u32 n_units, u32 n_pow2) {
  // 64 one-byte stores per instruction, each lane its own byte of a packed array (84 per line and tick in the train)
  const u32 lane = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 u = (blockIdx.x * 8 + k) * 64 + lane;
    if (u >= n_units) continue;
    base[perm(u, n_pow2)] = (unsigned char)u;
  }
}
__global__ void fill_kernel(u64 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull;
}

int main(int argc, char **argv) {
  const char *what = argc > 1 ? argv[1] : "rd128";
  const u32 n_pow2 = 1u << 22;                      // 4 M rows of 128 B = 512 MiB: sixteen times the L2s, twice the MALL
  const u32 n_units = argc > 2 ? (u32)atoi(argv[2]) : (1u << 21);
  char *d; u64 *sink;
  CK(hipMalloc(&d, (size_t)n_pow2 * 128)); CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (u64 *)d, (size_t)n_pow2 * 16);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double bytes = 0;
  CK(hipEventRecord(e0));
#define RD(NAME, L, S, P, B) if (!strcmp(what, NAME)) { hipLaunchKernelGGL((rd_kernel<L, S, P>), dim3((n_units + (64 / L) * 8 - 1) / ((64 / L) * 8)), dim3(64), 0, 0, d, n_units, n_pow2, sink); bytes = (double)n_units * B; }
#define WR(NAME, L, S, NT, B) if (!strcmp(what, NAME)) { hipLaunchKernelGGL((wr_kernel<L, S, NT>), dim3((n_units + (64 / L) * 8 - 1) / ((64 / L) * 8)), dim3(64), 0, 0, d, n_units, n_pow2); bytes = (double)n_units * B; }
  RD("rd128", 8, 128, 16, 128) RD("rd128nt", 8, 128, 2, 128) RD("rd64", 4, 128, 2, 64) RD("rd32", 2, 128, 2, 32) RD("rd64sc1", 4, 128, 16, 64)
  WR("wr16", 1, 128, false, 16) WR("wr32", 2, 128, false, 32) WR("wr64nt", 4, 128, true, 64) WR("wr32nt", 2, 128, true, 32)
  WR("wr128", 8, 128, false, 128)
#define RW(NAME, W) if (!strcmp(what, NAME)) { hipLaunchKernelGGL((rw_kernel<W>), dim3((n_units + 63) / 64), dim3(64), 0, 0, d, n_units, n_pow2, sink); bytes = (double)n_units * (128 + 16 * W); }
  RW("rw0", 0) RW("rw16", 1) RW("rw32", 2) RW("rw64", 4) RW("rw128", 8)
  if (!strcmp(what, "rd1")) { hipLaunchKernelGGL(rd1_kernel, dim3((n_units + 7) / 8), dim3(64), 0, 0, (unsigned char *)d, n_units, n_pow2 * 2u, sink); bytes = (double)n_units * 64; }
  if (!strcmp(what, "wr1")) { hipLaunchKernelGGL(wr1_kernel, dim3((n_units + 511) / 512), dim3(64), 0, 0, (unsigned char *)d, n_units, 1u << 19); bytes = (double)n_units; }
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  if (bytes == 0) { printf("unknown kernel %s\n", what); return 2; }
  printf("{\"kernel\": \"%s\", \"units\": %u, \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", what, n_units, bytes, ms, bytes / ms / 1e6);
  return 0;
}
