/* orc_net.c — CPU oracle: forward pass of the policy/value residual CNN (agent/model.py:28-72,
 * evaluated by agent/api.py:30-45 `predict_on_batch`) on the folded "raznet v1" weight blob.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 * The arithmetic of the reference's net lives in un-vendored third-party code (Keras 2.1.2 /
 * TensorFlow 1.4.1, requirements.txt:25,59) and no reference test exercises it, so parity for
 * p/v is "unpinned by the reference" (SURVEY §8(c)).  This restatement fixes ONE evaluation order
 * — every output is a single k-ordered fmaf chain, exactly what v_mfma_f32_* computes on gfx950
 * (tools/probe_numerics.hip) — and is itself checked against the fp32 PyTorch restatement of the
 * Keras graph to 1e-5 (tests/test_net_oracle.py).
 *
 * "raznet-forward-v1":
 *   conv3x3 (same padding):  acc = b[oc]; for c in chunks of 16 input channels (one chunk if Cin <= 16):
 *                              for tap = ky*3+kx in 0..8: for ic in chunk c:
 *                                acc = fmaf(x[ic][y+ky-1][x+kx-1] (0 off-board), w[oc][ic][ky][kx], acc)
 *                            (chunk-major so a GEMM kernel can stage 16 channels of activations in LDS
 *                             and reuse them for all nine taps; identical to tap-major for Cin <= 16)
 *   stem / first conv of a block: out = max(acc, 0);  second conv of a block: out = max(acc + skip, 0)
 *   conv1x1 heads:           acc = b[oc]; for ic: acc = fmaf(x[ic][sq], w[oc][ic], acc); out = max(acc,0)
 *   dense:                   acc = b[o];  for j:  acc = fmaf(h[j], W[j][o], acc)
 *   policy: softmax with m = max, e_i = det_expf(l_i - m), s = xor-butterfly sum (1,2,4,8,16,32), e_i / s
 *   value:  det_tanhf(dense2(relu(dense1(flatten(relu(conv1x1))))))
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define RAZNET_MAGIC 0x4E5A4152

typedef struct {
    int F, R, V;
    const float* w;
} net_view;

static int net_open(const void* blob, size_t bytes, net_view* nv) {
    if (bytes < 32) return -1;
    int32_t h[8];
    memcpy(h, blob, 32);
    if (h[0] != RAZNET_MAGIC || h[1] != 1 || h[5] != 3) return -1;
    nv->F = h[2]; nv->R = h[3]; nv->V = h[4];
    size_t F = (size_t)nv->F, R = (size_t)nv->R, V = (size_t)nv->V;
    size_t n = (F * 18 + F) + R * 2 * (F * F * 9 + F) + (2 * F + 2) + (128 * 64 + 64) + (F + 1) + (64 * V + V) + (V + 1);
    if (bytes != 32 + 4 * n) return -1;
    nv->w = (const float*)((const char*)blob + 32);
    return 0;
}

/* in: [cin][64]; w: [cout][cin][9]; out: [cout][64]; skip may be NULL.  Vectorisable over the 64
 * squares; each square's chain order is tap-major then ic, as specified. */
__attribute__((target_clones("arch=haswell", "default")))
static void conv3x3(const float* in, int cin, const float* w, const float* b, int cout,
                    const float* skip, float* out) {
    float* pad = (float*)calloc((size_t)cin * 100, sizeof(float)); /* 10x10 zero-padded planes */
    for (int ic = 0; ic < cin; ++ic)
        for (int y = 0; y < 8; ++y)
            for (int x = 0; x < 8; ++x) pad[ic * 100 + (y + 1) * 10 + (x + 1)] = in[ic * 64 + y * 8 + x];
    for (int oc = 0; oc < cout; ++oc) {
        float acc[64];
        for (int s = 0; s < 64; ++s) acc[s] = b[oc];
        for (int c0 = 0; c0 < cin; c0 += 16)
        for (int tap = 0; tap < 9; ++tap) {
            int ky = tap / 3, kx = tap % 3;
            for (int ic = c0; ic < cin && ic < c0 + 16; ++ic) {
                float wv = w[((size_t)oc * cin + ic) * 9 + tap];
                const float* p = pad + ic * 100 + ky * 10 + kx;
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8; ++x)
                        acc[y * 8 + x] = fmaf(p[y * 10 + x], wv, acc[y * 8 + x]);
            }
        }
        for (int s = 0; s < 64; ++s) {
            float v = acc[s];
            if (skip) v = v + skip[oc * 64 + s];
            out[oc * 64 + s] = v > 0.0f ? v : 0.0f;
        }
    }
    free(pad);
}

static void conv1x1_relu(const float* in, int cin, const float* w, const float* b, int cout, float* out) {
    for (int oc = 0; oc < cout; ++oc)
        for (int s = 0; s < 64; ++s) {
            float acc = b[oc];
            for (int ic = 0; ic < cin; ++ic) acc = fmaf(in[ic * 64 + s], w[oc * cin + ic], acc);
            out[oc * 64 + s] = acc > 0.0f ? acc : 0.0f;
        }
}

static void dense(const float* h, int nin, const float* W, const float* b, int nout, float* out) {
    for (int o = 0; o < nout; ++o) {
        float acc = b[o];
        for (int j = 0; j < nin; ++j) acc = fmaf(h[j], W[j * nout + o], acc);
        out[o] = acc;
    }
}

/* planes: [2][64] floats (own, enemy) as the player builds them (agent/player.py:307-309). */
int orc_net_forward_planes(const void* blob, size_t bytes, const float* planes, float* policy, float* value) {
    net_view nv;
    if (net_open(blob, bytes, &nv)) return -1;
    const int F = nv.F, R = nv.R, V = nv.V;
    const float* w = nv.w;
    float* a = (float*)malloc((size_t)F * 64 * sizeof(float));
    float* t = (float*)malloc((size_t)F * 64 * sizeof(float));
    float* u = (float*)malloc((size_t)F * 64 * sizeof(float));
    conv3x3(planes, 2, w, w + F * 18, F, NULL, a);
    w += F * 18 + F;
    for (int r = 0; r < R; ++r) {
        conv3x3(a, F, w, w + (size_t)F * F * 9, F, NULL, t);
        w += (size_t)F * F * 9 + F;
        conv3x3(t, F, w, w + (size_t)F * F * 9, F, a, u);
        w += (size_t)F * F * 9 + F;
        float* sw = a; a = u; u = sw;
    }
    float ph[128], logits[64];
    conv1x1_relu(a, F, w, w + 2 * F, 2, ph);
    w += 2 * F + 2;
    dense(ph, 128, w, w + 128 * 64, 64, logits);
    w += 128 * 64 + 64;
    float m = logits[0];
    for (int i = 1; i < 64; ++i) m = logits[i] > m ? logits[i] : m;
    float e[64], s[64], s2[64];
    for (int i = 0; i < 64; ++i) s[i] = e[i] = orc_det_expf(logits[i] - m);
    for (int d = 1; d < 64; d <<= 1) { /* xor butterfly: every slot ends up with the same total */
        for (int i = 0; i < 64; ++i) s2[i] = s[i] + s[i ^ d];
        memcpy(s, s2, sizeof s);
    }
    for (int i = 0; i < 64; ++i) policy[i] = e[i] / s[i];
    float vh[64];
    conv1x1_relu(a, F, w, w + F, 1, vh);
    w += F + 1;
    float* h1 = (float*)malloc((size_t)V * sizeof(float));
    dense(vh, 64, w, w + 64 * V, V, h1);
    for (int i = 0; i < V; ++i) h1[i] = h1[i] > 0.0f ? h1[i] : 0.0f;
    w += 64 * V + V;
    float acc = w[V];
    for (int j = 0; j < V; ++j) acc = fmaf(h1[j], w[j], acc);
    *value = orc_det_tanhf(acc);
    free(h1); free(a); free(t); free(u);
    return 0;
}

int orc_net_forward(const void* blob, size_t bytes, u64 own, u64 enemy, float* policy, float* value) {
    float planes[128];
    for (int i = 0; i < 64; ++i) {
        planes[i] = (float)((own >> i) & 1);
        planes[64 + i] = (float)((enemy >> i) & 1);
    }
    return orc_net_forward_planes(blob, bytes, planes, policy, value);
}
