// Micro-probe: how fast can gfx950 gather random 128-byte lines (the per-server "hot" line access
// pattern of rgb_tick_kernel) per-lane vs cooperatively?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1);} } while (0)

// (a) each lane reads its own line with 8 x 16B loads, sums, writes 16 B back
__global__ void per_lane(const u32 *idx, ulonglong2 *lines, u32 n, u64 *out, int wr) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s = idx[i];
  ulonglong2 *p = lines + (size_t)s * 8;
  ulonglong2 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = p[k];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += v[k].x ^ v[k].y;
  if (wr) p[0] = make_ulonglong2(acc, v[0].y + 1);
  out[i] = acc;
}
template <int P>
__global__ void per_lane_p(const u32 *idx, ulonglong2 *lines, u32 n, u64 *out, int wr) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s = idx[i];
  ulonglong2 *p = lines + (size_t)s * 8;     // lines stay 128 B apart; only the first P pieces are read
  ulonglong2 v[P];
#pragma unroll
  for (int k = 0; k < P; ++k) v[k] = p[k];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < P; ++k) acc += v[k].x ^ v[k].y;
  if (wr) p[0] = make_ulonglong2(acc, v[0].y + 1);
  out[i] = acc;
}
// (b) cooperative: 8 lanes per line (16 B each), LDS transpose, each lane then reads its line from LDS
__global__ void coop(const u32 *idx, ulonglong2 *lines, u32 n, u64 *out, int wr) {
  __shared__ ulonglong2 buf[64 * 9];
  __shared__ u32 sidx[64];
  u32 lane = threadIdx.x, base = blockIdx.x * 64;
  u32 i = base + lane;
  sidx[lane] = i < n ? idx[i] : 0;
  __syncthreads();
  ulonglong2 t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    u32 j = k * 8 + (lane >> 3), part = lane & 7;
    t[k] = lines[(size_t)sidx[j] * 8 + part];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    u32 j = k * 8 + (lane >> 3), part = lane & 7;
    buf[j * 9 + part] = t[k];
  }
  __syncthreads();
  if (i >= n) return;
  u64 acc = 0;
  ulonglong2 v0 = buf[lane * 9];
#pragma unroll
  for (int k = 0; k < 8; ++k) { ulonglong2 v = buf[lane * 9 + k]; acc += v.x ^ v.y; }
  if (wr) lines[(size_t)sidx[lane] * 8] = make_ulonglong2(acc, v0.y + 1);
  out[i] = acc;
}
int main() {
  const u32 S = 327680, n = 211000;
  std::vector<u32> h(n);
  u64 x = 88172645463325252ull;
  std::vector<u32> perm(S);
  for (u32 i = 0; i < S; ++i) perm[i] = i;
  for (u32 i = S - 1; i > 0; --i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; u32 j = x % (i + 1); std::swap(perm[i], perm[j]); }
  u32 *d_idx; ulonglong2 *d_lines; u64 *d_out;
  CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&d_lines, (size_t)S * 128)); CK(hipMalloc(&d_out, n * 8));
  CK(hipMemset(d_lines, 1, (size_t)S * 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int sorted = 0; sorted < 2; ++sorted)
  for (int wr = 0; wr < 2; ++wr)
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 30; ++rep) {
      // new random subset each repetition (like a new tick)
      for (u32 i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = perm[(x % S)]; }
      if (sorted) std::sort(h.begin(), h.end());
      CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(per_lane, dim3((n + 63) / 64), dim3(64), 0, 0, d_idx, d_lines, n, d_out, wr);
      else hipLaunchKernelGGL(coop, dim3((n + 63) / 64), dim3(64), 0, 0, d_idx, d_lines, n, d_out, wr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 2 && ms < best) best = ms;
    }
    printf("%s %s %s : %.2f us for %u lines (%.2f TB/s of 128B lines)\n", sorted ? "sorted " : "random ",
           mode ? "coop    " : "per-lane", wr ? "rd+wr16" : "rd     ", best * 1e3, n, n * 128.0 / (best * 1e-3) / 1e12);
  }
  for (int wr = 0; wr < 2; ++wr)
  for (int P = 1; P <= 8; P *= 2) {
    float best = 1e9;
    for (int rep = 0; rep < 30; ++rep) {
      for (u32 i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = perm[(x % S)]; }
      CK(hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      dim3 g((n + 63) / 64), b(64);
      if (P == 1) hipLaunchKernelGGL(per_lane_p<1>, g, b, 0, 0, d_idx, d_lines, n, d_out, wr);
      if (P == 2) hipLaunchKernelGGL(per_lane_p<2>, g, b, 0, 0, d_idx, d_lines, n, d_out, wr);
      if (P == 4) hipLaunchKernelGGL(per_lane_p<4>, g, b, 0, 0, d_idx, d_lines, n, d_out, wr);
      if (P == 8) hipLaunchKernelGGL(per_lane_p<8>, g, b, 0, 0, d_idx, d_lines, n, d_out, wr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 2 && ms < best) best = ms;
    }
    printf("random per-lane %3d B of each 128-B line %s : %.2f us\n", P * 16, wr ? "rd+wr16" : "rd     ", best * 1e3);
  }
  // launch floor
  { float best = 1e9; for (int rep = 0; rep < 20; ++rep) { CK(hipEventRecord(e0));
      hipLaunchKernelGGL(per_lane_p<1>, dim3(1), dim3(64), 0, 0, d_idx, d_lines, 1, d_out, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("launch floor (1 block): %.2f us\n", best * 1e3); }
  return 0;
}
