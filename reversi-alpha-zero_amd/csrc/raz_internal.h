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

// raz_net_f16x3.hip: the sticky range flag of the split-f16 path (flag[0]; flag[1] counts the rows repaired since the net was
// loaded) and its exact-f32 heads
unsigned* raz_net_f16x3_flag(const float* W, int F, int R, int V);
// raz_net.hip: rows i < n (< *n_ptr) of a split-f16 forward whose range flag is up - an activation left the f16 range - are
// evaluated again by the exact-f32 one-wave-per-position kernel (bit-identical to raznet-forward-v1; activations in LDS) into the
// same output rows (list[i], or i).  Row i's flag is word i * RAZ_NET_ROWFLAG_WORDS of `rowflag` (a 64-byte area per row keeps the
// scratch size linear in n, which the engine's slices rely on); word 1 counts the rows repaired by THIS forward, and a forward
// that repairs more than RAZ_NET_REPAIR_ROWS raises `sticky` (answers stay valid; the caller should move to the f32 kernels).
#define RAZ_NET_ROWFLAG_WORDS 16
#define RAZ_NET_REPAIR_ROWS 32
int raz_net_repair_rows(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy, float* policy, float* value,
                        size_t n, unsigned* rowflag, unsigned* sticky, const uint32_t* list, const uint32_t* n_ptr, hipStream_t s);
int raz_net_heads_split(const float* W, int F, int R, int V, const unsigned char* trunk, const uint8_t* active, float* policy, float* value,
                        size_t n, hipStream_t s, const uint32_t* list, const uint32_t* n_ptr);
