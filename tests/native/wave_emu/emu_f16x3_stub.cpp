// emu_f16x3_stub.cpp - the split-f16 trunk (csrc/raz_net_f16x3.hip: f16 matrix cores, LDS-DMA) is not emulated: its entry points
// exist so that csrc/raz_net.hip links, and refuse.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include "../../../include/raz.h"
int raz_fail(int code, const char* msg);
void raz_net_build_f16x3(const float*, float*, int, int, int) {}
size_t raz_net_f16x3_scratch_bytes(int F, size_t n) { return (size_t)2 * n * F * 256; }
const float* raz_net_f16x3_flag(const float* W, int, int, int) { return W; }
int raz_net_forward_f16x3(const float*, int, int, int, const uint64_t*, const uint64_t*, const uint8_t*, float*, float*, size_t, void*, size_t,
                          hipStream_t, const uint32_t*, const uint32_t*) {
    return raz_fail(RAZ_EINVAL, "wave emulator: the split-f16 kernels are not emulated");
}
