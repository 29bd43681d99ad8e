// raz_capi.hip — scalar host entry points and error state of libraz (C ABI in include/raz.h).
//
// The scalar functions are the host instantiation of the same __host__ __device__ primitives the
// kernels use (raz_bitboard.h); they replace lib/bitboard.py / lib/alt/bitboard_cython.pyx for
// single positions (ReversiEnv facade, record emission).  They are not a fallback for the batched
// device paths: every *_batch / engine entry point is device-only and fails with RAZ_EDEVICE when
// no GPU is present.
#include <stdio.h>
#include <string.h>
#include "raz_bitboard.h"
#include "raz_internal.h"

namespace {
thread_local char g_err[512] = "";
}

int raz_fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg ? msg : "(null)");
    return code;
}

int raz_fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof g_err, "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
    return RAZ_EDEVICE;
}

int raz_check_launch(const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return raz_fail_hip(e, where);
    return RAZ_OK;
}

extern "C" int raz_abi_version(void) { return RAZ_ABI_VERSION; }
extern "C" const char* raz_last_error(void) { return g_err; }

extern "C" uint64_t raz_find_correct_moves(uint64_t own, uint64_t enemy) {
    return bb_legal_moves(own, enemy);
}

extern "C" uint64_t raz_calc_flip(int pos, uint64_t own, uint64_t enemy) {
    if (pos < 0 || pos > 63) {
        raz_fail(RAZ_EINVAL, "raz_calc_flip: pos out of range 0..63");
        return 0;
    }
    return bb_calc_flip(pos, own, enemy);
}

extern "C" int raz_bit_count(uint64_t x) { return bb_popcount(x); }
extern "C" uint64_t raz_flip_vertical(uint64_t x) { return bb_flip_vertical(x); }
extern "C" uint64_t raz_flip_diag_a1h8(uint64_t x) { return bb_flip_diag_a1h8(x); }
extern "C" uint64_t raz_rotate90(uint64_t x) { return bb_rotate90(x); }
extern "C" uint64_t raz_rotate180(uint64_t x) { return bb_rotate180(x); }

extern "C" int raz_bit_to_array(uint64_t x, int size, uint8_t* out) {
    if (!out || size < 0 || size > 64) return raz_fail(RAZ_EINVAL, "raz_bit_to_array: bad size/out");
    for (int i = 0; i < size; ++i) out[i] = (uint8_t)((x >> i) & 1);
    return RAZ_OK;
}

extern "C" int raz_env_step(uint64_t* black, uint64_t* white, uint8_t* player, uint8_t* status,
                            uint64_t* legal, int action) {
    if (!black || !white || !player || !status || !legal)
        return raz_fail(RAZ_EINVAL, "raz_env_step: NULL pointer");
    if (!((action >= 0 && action <= 63) || action == RAZ_ACTION_RESIGN))
        return raz_fail(RAZ_EINVAL, "raz_env_step: action must be 0..63 or 255");
    if (*player != RAZ_PLAYER_BLACK && *player != RAZ_PLAYER_WHITE)
        return raz_fail(RAZ_EINVAL, "raz_env_step: player must be 1 or 2");
    raz_step_result r = bb_env_step(*black, *white, *player, action);
    *black = r.black;
    *white = r.white;
    *player = r.player;
    *status = r.status;
    *legal = r.legal;
    return RAZ_OK;
}
