// raz_net_layout.h — float offsets inside the device weight buffer of a loaded net (raz_net_load).
// Region 1 ("wave" layout, used by k_net_wave): conv layer l at conv_off(l): [F/16][9][Cin][16] + bias[F];
//   then the heads exactly as in the blob.
// Region 2 ("mfma" layout, used by k_net_mfma) at mfma_off(): per conv layer, per 16-channel N tile,
//   per k-step s, 64 floats: the B operand of v_mfma_f32_16x16x4_f32 for lane l, i.e. the weight of
//   out-channel nt*16 + (l&15) at reduction index k = 4s + (l>>4), where k enumerates (16-channel chunk,
//   tap, channel in chunk) — the order raznet-forward-v1 fixes; k beyond 9*Cin (layer 0) is 0.
#pragma once
#include <stddef.h>

#define RAZ_HD_LAYOUT __host__ __device__ inline

RAZ_HD_LAYOUT size_t conv_floats(int F, int cin) { return (size_t)F * 9 * cin + F; }
RAZ_HD_LAYOUT size_t conv_off(int F, int l) {
    return l == 0 ? 0 : conv_floats(F, 2) + (size_t)(l - 1) * conv_floats(F, F);
}
RAZ_HD_LAYOUT size_t heads_off(int F, int R) { return conv_off(F, 2 * R + 1); }
RAZ_HD_LAYOUT size_t wave_floats(int F, int R, int V) {
    return heads_off(F, R) + (2 * (size_t)F + 2) + (128 * 64 + 64) + ((size_t)F + 1) + (64 * (size_t)V + V) + ((size_t)V + 1);
}
RAZ_HD_LAYOUT int mfma_ksteps(int F, int l) { return l == 0 ? 5 : 9 * F / 4; }
RAZ_HD_LAYOUT size_t mfma_layer_floats(int F, int l) { return (size_t)(F / 16) * mfma_ksteps(F, l) * 64; }
RAZ_HD_LAYOUT size_t mfma_off(int F, int R, int V) { return (wave_floats(F, R, V) + 63) / 64 * 64; }
RAZ_HD_LAYOUT size_t mfma_layer_off(int F, int R, int V, int l) {
    return mfma_off(F, R, V) + (l == 0 ? 0 : mfma_layer_floats(F, 0) + (size_t)(l - 1) * mfma_layer_floats(F, 1));
}
// Region 3 ("wide" layout, used by k_conv3x3_wide, F % 64 == 0 and F >= 128): per conv layer l >= 1,
//   per 16-channel input chunk c, per 64-channel output tile nt: the A operands of
//   v_mfma_f32_32x32x2_f32 for the chunk's 72 k-steps: [s 72][mt 2][64 lanes], lane l holding the
//   weight of out-channel nt*64 + mt*32 + (l&31) at k = 2s + (l>>5) within the chunk (k = tap*16 + ic).
RAZ_HD_LAYOUT bool wide_supported(int F) { return F >= 128 && F % 64 == 0; }
RAZ_HD_LAYOUT size_t wide_off(int F, int R, int V) { return (mfma_layer_off(F, R, V, 2 * R + 1) + 63) / 64 * 64; }
RAZ_HD_LAYOUT size_t wide_layer_floats(int F) { return (size_t)F * F * 9; }
RAZ_HD_LAYOUT size_t wide_tile_off(int F, int R, int V, int l, int c, int nt) {  // l >= 1
    return wide_off(F, R, V) + (size_t)(l - 1) * wide_layer_floats(F) + ((size_t)c * (F / 64) + nt) * (72 * 128);
}
// Region 4 ("f16x3" layout, used by k_conv3x3_f16x3, F % 128 == 0): every weight w of a conv layer l >= 1 as the pair of
//   halfs (hi, lo) = (f16(w * S_l), f16(w * S_l - hi)) with S_l a power of two chosen per layer so that max |w| * S_l is in
//   [2^14, 2^15); 4 bytes per weight like region 3.  Per layer: [oc tile of 128][16-channel chunk][tap group of 3][tap in
//   group][k group of 8 channels: 2][hi, lo][oc in tile: 128][8 halfs] - i.e. one (oc tile, chunk, tap group) stage of the
//   kernel is 24,576 contiguous bytes in exactly its LDS image order (lane-linear 16-byte units for global_load_lds).  Then
//   2R floats: 1 / S_l per layer.
RAZ_HD_LAYOUT bool f16x3_supported(int F) { return F >= 128 && F % 128 == 0; }
RAZ_HD_LAYOUT size_t f16x3_off(int F, int R, int V) { return (wide_off(F, R, V) + (size_t)2 * R * wide_layer_floats(F) + 63) / 64 * 64; }
RAZ_HD_LAYOUT size_t f16x3_layer_off(int F, int R, int V, int l) { return f16x3_off(F, R, V) + (size_t)(l - 1) * wide_layer_floats(F); }  // l >= 1
RAZ_HD_LAYOUT size_t f16x3_scale_off(int F, int R, int V) { return f16x3_off(F, R, V) + (size_t)2 * R * wide_layer_floats(F); }
RAZ_HD_LAYOUT size_t total_floats(int F, int R, int V) {
    if (f16x3_supported(F)) return f16x3_scale_off(F, R, V) + (size_t)2 * R + 64;
    return wide_supported(F) ? wide_off(F, R, V) + (size_t)2 * R * wide_layer_floats(F) : mfma_layer_off(F, R, V, 2 * R + 1);
}
