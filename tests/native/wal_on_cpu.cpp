/*
 * wal_on_cpu.cpp -- TEST INFRASTRUCTURE: ra_amd/csrc/rgb_wal.hip (checksum and framing kernels, staging,
 * validation) and rgb_wal_host.cpp compiled as x86 C++ over tests/native/fake_hip, kernels executed by the
 * fiber-per-lane block emulation of kernel_on_cpu.cpp.  The C entry points of include/ra_gpu_wal.h are exported
 * unchanged; "device" buffers are host buffers.
 */
#define RGB_HOST_EMULATION 1
#include <stdlib.h>
#include <hip/hip_runtime.h>
#ifndef RGB_EMU_FULL_API   /* stand-alone WAL build: a dummy context; with rgb_api.hip the real one is used */
struct rgb_ctx { int unused; };
extern "C" void *rgb_ctx_stream(rgb_ctx *) { return nullptr; }
extern "C" int rgb_ctx_device(rgb_ctx *) { return 0; }
#endif
#include "../../ra_amd/csrc/rgb_wal.hip"
#include "../../ra_amd/csrc/rgb_wal_host.cpp"
#ifndef RGB_EMU_FULL_API
extern "C" rgb_ctx *emu_wal_ctx() { static rgb_ctx c; return &c; }
#endif
