/*
 * api_on_cpu.cpp -- TEST INFRASTRUCTURE: ra_amd/csrc/rgb_api.hip (the C ABI: contexts, the staging ring,
 * rgb_submit's sub-tick rounds and family ordering, rgb_collect, the device-resident entry points) compiled as
 * x86 C++ over tests/native/fake_hip.  Linked with kernel_on_cpu.cpp and wal_on_cpu.cpp (-DRGB_EMU_FULL_API) it
 * yields a library with exactly the exports of libra_gpu_batch.so whose kernels run on the block emulation:
 * tests/test_c_abi_on_cpu.py points ra_amd.engine at it (in the test process only) and re-runs the C-ABI tests.
 */
#define RGB_HOST_EMULATION 1
#include <hip/hip_runtime.h>
#include "../../ra_amd/csrc/rgb_api.hip"
