// Probe (tools/, not part of the product): which XCCs does a stream created with hipExtStreamCreateWithCUMask run on?
// Prints, for several mask patterns, the number of blocks that ran on every XCC (s_getreg HW_REG_XCC_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where(unsigned *cnt, unsigned *cu) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    atomicAdd(cnt + (x & 15u), 1u);
    atomicOr(cu + (x & 15u), 1u << ((hw >> 8) & 15u));      // CU id bits of HW_ID (gfx9: [11:8])
    for (volatile int i = 0; i < 2000; ++i) { }
  }
}
static void run(const char *name, const std::vector<unsigned> &mask) {
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: create failed %d\n", name, (int)e); return; }
  unsigned *d; hipMalloc(&d, 32 * sizeof(unsigned)); hipMemset(d, 0, 32 * sizeof(unsigned));
  hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, st, d, d + 16);
  hipStreamSynchronize(st);
  unsigned h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("%-28s blocks per XCC:", name);
  for (int i = 0; i < 8; ++i) printf(" %5u", h[i]);
  printf("\n");
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  const int W = 8;                                   // 256 bits
  { std::vector<unsigned> m(W, 0xFFFFFFFFu); run("all", m); }
  { std::vector<unsigned> m(W, 0); m[0] = 0xFFFFFFFFu; run("bits 0..31", m); }
  { std::vector<unsigned> m(W, 0); m[1] = 0xFFFFFFFFu; run("bits 32..63", m); }
  { std::vector<unsigned> m(W, 0x01010101u); run("bits = 0 mod 8", m); }
  { std::vector<unsigned> m(W, 0x02020202u); run("bits = 1 mod 8", m); }
  { std::vector<unsigned> m(W, 0x80808080u); run("bits = 7 mod 8", m); }
  return 0;
}
