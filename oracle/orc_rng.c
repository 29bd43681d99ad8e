/* orc_rng.c — CPU oracle: the counter-based random stream "raz-rng-v1" and the deterministic
 * elementary functions "raz-math-v1".  TEST INFRASTRUCTURE (see orc.h).
 *
 * Why this exists: the reference draws from numpy's global MT19937 (np.random.choice
 * agent/player.py:112, numpy.random.random x2 :300-301, np.random.dirichlet lib/bitboard.py:164,
 * random.random worker/self_play.py:144, np.random.random :182).  A per-process global stream
 * cannot be reproduced by thousands of concurrent on-device games, so — as SURVEY.md §7 hard
 * part 2 prescribes — the harness injects a counter-based stream at exactly those call sites of
 * the UNMODIFIED reference (oracle/ref_harness.py + tests/golden/make_golden_mcts.py patch the
 * module-level names), and the engine implements the same stream on device.
 *
 *   raz-rng-v1: Philox4x32-10 (Salmon et al., SC'11), key = (seed, global game id),
 *   counter = (idx, sub, event, purpose).  One block = 4 x u32 = two 53-bit uniforms in [0,1).
 *     purpose 0 EXPAND   : event = # of leaf expansions so far in this game; d0 -> flip (<0.5),
 *                          d1 -> rot = int(d1*4)                     (agent/player.py:300-301)
 *     purpose 1 CHOICE   : event = # of move choices so far; d0 -> np.random.choice uniform (:112)
 *     purpose 2 DIRICHLET: event = # of root-noise draws so far; sub = index of the legal move in
 *                          ascending bit order; idx = rejection-loop attempt (lib/bitboard.py:164)
 *     purpose 3 GAME     : event 0: d0 -> enable_resign draw (worker/self_play.py:144),
 *                          d1 -> drop-draw draw (:182)
 *
 *   raz-math-v1: log/exp in f64 and exp/tanh in f32 built ONLY from IEEE-754 + - * / (correctly
 *   rounded on both x86-64 and gfx950, verified by tools/probe_numerics.hip) in a fixed order, so
 *   the device reproduces them bit for bit.  Accuracy ~1e-15 rel. (f64), ~2e-7 (f32): they are
 *   sampling/activation helpers, not libm replacements.
 */
#include <math.h>
#include <string.h>
#include "orc.h"

/* ---- Philox4x32-10 ---- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t a, uint32_t b) { /* numpy's 53-bit recipe: (a>>5, b>>6) */
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* Two uniforms in [0,1) for (seed, game, purpose, event, sub, idx). */
void orc_rng_pair(uint32_t seed, uint32_t game, uint32_t purpose, uint32_t event, uint32_t sub,
                  uint32_t idx, double out[2]) {
    uint32_t ctr[4] = {idx, sub, event, purpose}, key[2] = {seed, game}, r[4];
    orc_philox4x32_10(ctr, key, r);
    out[0] = u53(r[0], r[1]);
    out[1] = u53(r[2], r[3]);
}

/* ---- raz-math-v1, f64 ---- */
static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

#define LN2_HI 0x1.62e42fee00000p-1
#define LN2_LO 0x1.a39ef35793c76p-33
#define INV_LN2 0x1.71547652b82fep+0

/* log(x) for finite x > 0 (x <= 0 returns -inf-like sentinel -1e308; callers never pass it). */
double orc_det_log(double x) {
    if (!(x > 0.0)) return -1.0e308;
    int e = 0;
    if (x < 0x1p-1022) { x *= 0x1p54; e = -54; } /* subnormal */
    uint64_t b = to_bits(x);
    e += (int)(b >> 52) - 1023;
    double m = from_bits((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }                     /* (0.707,1.414] */
    double s = (m - 1.0) / (m + 1.0), z = s * s;
    /* atanh series: log m = 2 s (1 + z/3 + z^2/5 + ... + z^12/25), Estrin association */
    double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    double a0 = 1.0 + 0x1.5555555555555p-2 * z;
    double a1 = 0x1.999999999999ap-3 + 0x1.2492492492492p-3 * z;
    double a2 = 0x1.c71c71c71c71cp-4 + 0x1.745d1745d1746p-4 * z;
    double a3 = 0x1.3b13b13b13b14p-4 + 0x1.1111111111111p-4 * z;
    double a4 = 0x1.e1e1e1e1e1e1ep-5 + 0x1.af286bca1af28p-5 * z;
    double a5 = 0x1.8618618618618p-5 + 0x1.642c8590b2164p-5 * z;
    double a6 = 0x1.47ae147ae147bp-5;
    double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2, b2 = a4 + a5 * z2;
    double d0 = b0 + b1 * z4, d1 = b2 + a6 * z4;
    double p = d0 + d1 * z8;
    double de = (double)e;
    return de * LN2_HI + (de * LN2_LO + (2.0 * s) * p);
}

static double floor_det(double v) { /* floor for |v| < 2^31 */
    double t = (double)(long long)v;
    return (t > v) ? t - 1.0 : t;
}

/* exp(x), x finite; underflows to 0 below -745, saturates to 1.79e308 above 709.78. */
double orc_det_exp(double x) {
    if (x < -745.0) return 0.0;
    if (x > 709.78) return 0x1.fffffffffffffp+1023;
    double k = floor_det(x * INV_LN2 + 0.5);
    double r = (x - k * LN2_HI) - k * LN2_LO;
    /* sum_{k=0..14} r^k/k!, Estrin association */
    double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    double a0 = 1.0 + r;
    double a1 = 0.5 + 0x1.5555555555555p-3 * r;
    double a2 = 0x1.5555555555555p-5 + 0x1.1111111111111p-7 * r;
    double a3 = 0x1.6c16c16c16c17p-10 + 0x1.a01a01a01a01ap-13 * r;
    double a4 = 0x1.a01a01a01a01ap-16 + 0x1.71de3a556c734p-19 * r;
    double a5 = 0x1.27e4fb7789f5cp-22 + 0x1.ae64567f544e4p-26 * r;
    double a6 = 0x1.1eed8eff8d898p-29 + 0x1.6124613a86d09p-33 * r;
    double a7 = 0x1.93974a8c07c9dp-37;
    double b0 = a0 + a1 * r2, b1 = a2 + a3 * r2, b2 = a4 + a5 * r2, b3 = a6 + a7 * r2;
    double d0 = b0 + b1 * r4, d1 = b2 + b3 * r4;
    double p = d0 + d1 * r8;
    int ki = (int)k;
    if (ki >= -1021 && ki <= 1023) return p * from_bits((uint64_t)(ki + 1023) << 52);
    if (ki < -1021) return (p * from_bits((uint64_t)(ki + 1000 + 1023) << 52)) * 0x1p-1000;
    return (p * 0x1p+1023) * from_bits((uint64_t)(ki - 1023 + 1023) << 52);
}

/* x^y for x >= 0: exp(y * log x); 0^y = 0 (y > 0). */
double orc_det_pow(double x, double y) {
    if (!(x > 0.0)) return 0.0;
    return orc_det_exp(y * orc_det_log(x));
}

/* ---- raz-math-v1, f32 ---- */
static float f_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

float orc_det_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * 0x1.715476p+0f + 0.5f;
    float k = (float)(int)t;
    if (k > t) k = k - 1.0f; /* floor */
    float r = (x - k * 0x1.63p-1f) - k * -0x1.bd0106p-13f;
    float p = 0x1.a01a02p-13f;   /* 1/5040 */
    p = p * r + 0x1.6c16c2p-10f; /* 1/720  */
    p = p * r + 0x1.111112p-7f;  /* 1/120  */
    p = p * r + 0x1.555556p-5f;  /* 1/24   */
    p = p * r + 0x1.555556p-3f;  /* 1/6    */
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    int ki = (int)k; /* in [-126, 127] given the clamps above */
    return p * f_from_bits((uint32_t)(ki + 127) << 23);
}

float orc_det_tanhf(float x) {
    float ax = x < 0.0f ? -x : x;
    if (ax > 10.0f) ax = 10.0f;
    float t = orc_det_expf(2.0f * ax);
    float r = (t - 1.0f) / (t + 1.0f);
    return x < 0.0f ? -r : r;
}

/* ---- root noise: Dirichlet([alpha]*k) as normalised Gamma(alpha,1) draws -------------------
 * Gamma(alpha) for alpha < 1 by the rejection scheme numpy's legacy generator uses for shape < 1
 * (numpy/random/src/legacy/legacy-distributions.c legacy_standard_gamma, an Ahrens-Dieter GS
 * variant); alpha == 1 is an exponential; alpha > 1 the Marsaglia-Tsang scheme the same function uses for shape > 1
 * (no shipped config needs it - config.py:138 dirichlet_alpha = 0.5 - but lib/bitboard.py:162-171 takes any alpha). */
double orc_det_cos2(double u);

double orc_gamma_sample(double alpha, uint32_t seed, uint32_t game, uint32_t event, uint32_t sub) {
    double d[2];
    if (alpha == 1.0) {
        orc_rng_pair(seed, game, 2, event, sub, 0, d);
        return -orc_det_log(1.0 - d[0]);
    }
    if (!(alpha > 0.0)) return -1.0;
    if (alpha > 1.0) {
        /* numpy's legacy_standard_gamma for shape > 1 (Marsaglia-Tsang), what np.random.dirichlet (lib/bitboard.py:164) runs
         * per component: b = shape - 1/3, c = 1/sqrt(9b); X ~ N(0,1), V = (1 + cX)^3 > 0, U ~ U(0,1]; accept b V when
         * U < 1 - 0.0331 X^4 or log U < X^2/2 + b (1 - V + log V).  raz-rng-v1: attempt t draws the normal from block
         * (.., sub, 2t) (Box-Muller: |X| = sqrt(2 E cos^2 theta)), its sign and U from block (.., sub, 2t + 1). */
        const double b = alpha - 0x1.5555555555555p-2, c = 1.0 / sqrt(9.0 * b);
        for (uint32_t t = 0;; ++t) {
            double e[2];
            orc_rng_pair(seed, game, 2, event, sub, 2u * t, d);
            orc_rng_pair(seed, game, 2, event, sub, 2u * t + 1u, e);
            const double E = -orc_det_log(1.0 - d[0]);
            const double z2 = 2.0 * (E * orc_det_cos2(d[1]));
            const double az = sqrt(z2);
            const double Z = e[0] < 0.5 ? -az : az;
            double V = 1.0 + c * Z;
            if (V <= 0.0) continue;
            V = (V * V) * V;
            const double U = 1.0 - e[1];
            if (U < 1.0 - 0.0331 * (z2 * z2)) return b * V;
            if (orc_det_log(U) < 0.5 * z2 + b * ((1.0 - V) + orc_det_log(V))) return b * V;
        }
    }
    for (uint32_t t = 0;; ++t) {
        orc_rng_pair(seed, game, 2, event, sub, t, d);
        double U = d[0], V = -orc_det_log(1.0 - d[1]);
        if (U <= 1.0 - alpha) {
            double X = orc_det_pow(U, 1.0 / alpha);
            if (X <= V) return X;
        } else {
            double Y = -orc_det_log((1.0 - U) / alpha);
            double X = orc_det_pow(1.0 - alpha + alpha * Y, 1.0 / alpha);
            if (X <= V + Y) return X;
        }
    }
}

/* cos^2(2 pi u), u in [0,1): exact octant reduction + 9-term Taylor sine on [0, pi/4] (Estrin). */
double orc_det_cos2(double u) {
    double w = 4.0 * u;
    int q = (int)w;
    double f = w - (double)q;
    int swap = f > 0.5;
    double gq = swap ? 1.0 - f : f;
    double x = gq * 0x1.921fb54442d18p+0;
    double z = x * x, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    double a0 = 1.0 + -0x1.5555555555555p-3 * z;
    double a1 = 0x1.1111111111111p-7 + -0x1.a01a01a01a01ap-13 * z;
    double a2 = 0x1.71de3a556c734p-19 + -0x1.ae64567f544e4p-26 * z;
    double a3 = 0x1.6124613a86d09p-33 + -0x1.ae7f3e733b81fp-41 * z;
    double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2;
    double p = (b0 + b1 * z4) + 0x1.952c77030ad4ap-49 * z8;
    double sn = x * p, s2 = sn * sn;
    int use_s2 = ((q & 1) != 0) != swap;
    return use_s2 ? s2 : 1.0 - s2;
}

/* The k Gamma(alpha,1) variates of one Dirichlet draw (raz-rng-v1 DIRICHLET).
 * alpha == 0.5 (config.py:138, every shipped config): Box-Muller pairs — block m gives
 * g[2m] = E cos^2(2 pi d1), g[2m+1] = E sin^2(2 pi d1) with E = -log(1 - d0) (Z^2/2 ~ Gamma(1/2)).
 * Other alpha: orc_gamma_sample (exponential / numpy-legacy rejection below 1 / Marsaglia-Tsang above 1). */
void orc_dirichlet_gammas(double alpha, int k, uint32_t seed, uint32_t game, uint32_t event, double* g) {
    if (alpha == 0.5) {
        for (int m = 0; 2 * m < k; ++m) {
            double d[2];
            orc_rng_pair(seed, game, 2, event, (uint32_t)m, 0, d);
            double E = -orc_det_log(1.0 - d[0]);
            double c2 = orc_det_cos2(d[1]);
            g[2 * m] = E * c2;
            if (2 * m + 1 < k) g[2 * m + 1] = E * (1.0 - c2);
        }
        return;
    }
    for (int j = 0; j < k; ++j) g[j] = orc_gamma_sample(alpha, seed, game, event, (uint32_t)j);
}

/* dirichlet_noise_of_mask (lib/bitboard.py:162-171): noise[i] for the set bits of `mask` in
 * ascending order, 0 elsewhere.  Sum of the gammas is accumulated in ascending order. */
void orc_dirichlet_noise_of_mask(u64 mask, double alpha, uint32_t seed, uint32_t game,
                                 uint32_t event, double out[64]) {
    double g[64], acc = 0.0;
    int k = orc_bit_count(mask);
    orc_dirichlet_gammas(alpha, k, seed, game, event, g);
    for (int j = 0; j < k; ++j) acc += g[j];
    k = 0;
    for (int i = 0; i < 64; ++i) out[i] = (mask >> i & 1) ? g[k++] / acc : 0.0;
}
