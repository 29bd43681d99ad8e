// raz_internal.h — error plumbing shared by the translation units of libraz.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/raz.h"

// Record `msg` as the calling thread's last error and return `code`.
int raz_fail(int code, const char* msg);
// Same, formatting a HIP error.
int raz_fail_hip(hipError_t e, const char* where);
// hipGetLastError() after a launch -> RAZ_OK / RAZ_EDEVICE.
int raz_check_launch(const char* where);

#define RAZ_HIP_TRY(expr, where)                              \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return raz_fail_hip(_e, where); \
    } while (0)

// raz_engine_fused.hip: `n_steps` simulation steps of the whole batch with k_tree_net (tree + narrow net in one kernel)
struct raz_engine_dev;
int raz_launch_tree_net(const raz_engine_dev& d, bool solver, uint32_t n_steps, const float* W, int R, int V, hipStream_t s);

// raz_net_f16x3.hip: the sticky range flag of the split-f16 path and its exact-f32 heads
unsigned* raz_net_f16x3_flag(const float* W, int F, int R, int V);
int raz_net_heads_split(const float* W, int F, int R, int V, const unsigned char* trunk, const uint8_t* active, float* policy, float* value,
                        size_t n, hipStream_t s, const uint32_t* list, const uint32_t* n_ptr);
