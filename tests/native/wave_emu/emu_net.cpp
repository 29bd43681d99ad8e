// emu_net.cpp — raz_net_* of the wave-emulator build: the net forward is NOT what this build is for (the MFMA kernels cannot run
// on a CPU); leaves are evaluated by the oracle's C net (oracle/orc_net.c), which the real exact-f32 kernels equal bit for
// bit (tests/test_engine_gpu.py), so that the emulated tree kernels see exactly the inputs the real ones see.
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include "../../../include/raz.h"
extern "C" {
#include "../../../oracle/orc.h"
}

extern "C" size_t raz_net_weight_bytes(int F, int R, int V) {
    if (F <= 0 || R < 0 || V <= 0) return 0;
    const size_t params = (size_t)2 * F * 9 + F + (size_t)2 * R * ((size_t)F * F * 9 + F) + (2 * F + 2) + (128 * 64 + 64) + (F + 1) +
                          ((size_t)64 * V + V) + (V + 1);
    return 32 + 4 * params + 4096;
}
extern "C" size_t raz_net_scratch_bytes(int, int, size_t) { return 0; }
extern "C" int raz_net_load(raz_net* net, const void* blob, size_t blob_bytes, void* d_weights, size_t d_bytes, raz_stream_t) {
    if (!net || !blob || !d_weights || d_bytes < blob_bytes) return RAZ_EINVAL;
    const int32_t* h = (const int32_t*)blob;
    net->filters = h[2];
    net->res_layers = h[3];
    net->value_fc = h[4];
    memcpy(d_weights, blob, blob_bytes);
    net->d_weights = d_weights;
    net->weight_bytes = blob_bytes;
    return RAZ_OK;
}
extern "C" int raz_net_forward(const raz_net* net, const uint64_t* own, const uint64_t* enemy, const uint8_t* active, float* policy,
                               float* value, size_t n, void*, size_t, raz_stream_t) {
    for (size_t i = 0; i < n; ++i) {
        if (active && !active[i]) continue;
        if (orc_net_forward(net->d_weights, net->weight_bytes, own[i], enemy[i], policy + 64 * i, value + i) != 0) return RAZ_EINVAL;
    }
    return RAZ_OK;
}
int raz_net_forward_compact(const raz_net* net, const uint64_t* own, const uint64_t* enemy, const uint8_t* active, float* policy,
                            float* value, size_t n, void* scratch, size_t scratch_bytes, hipStream_t stream, const uint32_t*, const uint32_t*) {
    return raz_net_forward(net, own, enemy, active, policy, value, n, scratch, scratch_bytes, (raz_stream_t)stream);
}
extern "C" int raz_net_range_check(const raz_net*, int* overflowed, raz_stream_t) {
    if (overflowed) *overflowed = 0;
    return RAZ_OK;
}
// the fused tree + net kernel evaluates leaves on the (emulated) matrix cores: it is part of the whole-product build
// (libraz_emu_full.so, tests/test_engine_fused_emu.py), not of this one
struct raz_engine_dev;
int raz_fail(int code, const char* msg);
int raz_launch_tree_net(const raz_engine_dev&, bool, uint32_t, const float*, int, int, hipStream_t) {
    return raz_fail(RAZ_EINVAL, "wave emulator (tree-only build): the fused tree + net kernel needs libraz_emu_full.so");
}
