// raz_detmath.h — device side of the two small specifications the search shares with its checker:
//
//   raz-rng-v1   counter-based random stream: Philox4x32-10, key = (seed, global game id),
//                counter = (idx, sub, event, purpose); a block yields two 53-bit uniforms.
//                It replaces numpy's process-global MT19937 at the reference's four draw sites
//                (agent/player.py:112, :300-301, lib/bitboard.py:164, worker/self_play.py:144,182)
//                so that thousands of concurrent games are reproducible and shard-independent.
//   raz-math-v1  log/exp/pow (f64) and exp/tanh (f32) composed only of IEEE-754 + - * / in a fixed
//                order.  gfx950 rounds those exactly like x86-64 (tools/probe_numerics.hip), so
//                these functions are bit-reproducible anywhere; ocml's log/exp are not.
//
// Compiled with -ffp-contract=off: no a*b+c here may be fused.
#pragma once
#include <stdint.h>
#include "raz_bitboard.h"

#define RAZ_RNG_EXPAND 0u
#define RAZ_RNG_CHOICE 1u
#define RAZ_RNG_DIRICHLET 2u
#define RAZ_RNG_GAME 3u

struct raz_u32x4 {
    uint32_t x, y, z, w;
};

RAZ_HD raz_u32x4 raz_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                   uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    raz_u32x4 o = {c0, c1, c2, c3};
    return o;
}

RAZ_HD double raz_u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// Two uniforms in [0,1) for (seed, game, purpose, event, sub, idx).
RAZ_HD void raz_rng_pair(uint32_t seed, uint32_t game, uint32_t purpose, uint32_t event, uint32_t sub,
                         uint32_t idx, double& d0, double& d1) {
    raz_u32x4 r = raz_philox4x32_10(idx, sub, event, purpose, seed, game);
    d0 = raz_u53(r.x, r.y);
    d1 = raz_u53(r.z, r.w);
}

RAZ_HD double raz_bits_to_f64(uint64_t b) {
    union {
        uint64_t u;
        double d;
    } c;
    c.u = b;
    return c.d;
}
RAZ_HD uint64_t raz_f64_to_bits(double d) {
    union {
        uint64_t u;
        double d;
    } c;
    c.d = d;
    return c.u;
}
RAZ_HD float raz_bits_to_f32(uint32_t b) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = b;
    return c.f;
}

// Natural log of a finite x > 0: split x = m * 2^e with m in (sqrt(1/2), sqrt(2)], then the
// atanh series  log m = 2s(1 + z/3 + z^2/5 + ... + z^12/25),  s = (m-1)/(m+1), z = s^2.
RAZ_HD double raz_det_log(double x) {
    if (!(x > 0.0)) return -1.0e308;
    int e = 0;
    if (x < 0x1p-1022) {
        x *= 0x1p54;
        e = -54;
    }
    const uint64_t b = raz_f64_to_bits(x);
    e += (int)(b >> 52) - 1023;
    double m = raz_bits_to_f64((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 0x1.6a09e667f3bcdp+0) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    // p(z) = sum_{k=0..12} z^k / (2k+1), Estrin scheme (fixed association: the oracle mirrors it)
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = 1.0 + 0x1.5555555555555p-2 * z;
    const double a1 = 0x1.999999999999ap-3 + 0x1.2492492492492p-3 * z;
    const double a2 = 0x1.c71c71c71c71cp-4 + 0x1.745d1745d1746p-4 * z;
    const double a3 = 0x1.3b13b13b13b14p-4 + 0x1.1111111111111p-4 * z;
    const double a4 = 0x1.e1e1e1e1e1e1ep-5 + 0x1.af286bca1af28p-5 * z;
    const double a5 = 0x1.8618618618618p-5 + 0x1.642c8590b2164p-5 * z;
    const double a6 = 0x1.47ae147ae147bp-5;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2, b2 = a4 + a5 * z2;
    const double d0 = b0 + b1 * z4, d1 = b2 + a6 * z4;
    const double p = d0 + d1 * z8;
    const double de = (double)e;
    return de * 0x1.62e42fee00000p-1 + (de * 0x1.a39ef35793c76p-33 + (2.0 * s) * p);
}

// exp(x): x = k ln2 + r, |r| <= ln2/2, degree-14 Taylor polynomial, exact 2^k scaling.
RAZ_HD double raz_det_exp(double x) {
    if (x < -745.0) return 0.0;
    if (x > 709.78) return 0x1.fffffffffffffp+1023;
    const double v = x * 0x1.71547652b82fep+0 + 0.5;
    double k = (double)(long long)v;
    if (k > v) k = k - 1.0;
    const double r = (x - k * 0x1.62e42fee00000p-1) - k * 0x1.a39ef35793c76p-33;
    // sum_{k=0..14} r^k / k!, Estrin scheme (fixed association: the oracle mirrors it)
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = 1.0 + r;
    const double a1 = 0.5 + 0x1.5555555555555p-3 * r;
    const double a2 = 0x1.5555555555555p-5 + 0x1.1111111111111p-7 * r;
    const double a3 = 0x1.6c16c16c16c17p-10 + 0x1.a01a01a01a01ap-13 * r;
    const double a4 = 0x1.a01a01a01a01ap-16 + 0x1.71de3a556c734p-19 * r;
    const double a5 = 0x1.27e4fb7789f5cp-22 + 0x1.ae64567f544e4p-26 * r;
    const double a6 = 0x1.1eed8eff8d898p-29 + 0x1.6124613a86d09p-33 * r;
    const double a7 = 0x1.93974a8c07c9dp-37;
    const double b0 = a0 + a1 * r2, b1 = a2 + a3 * r2, b2 = a4 + a5 * r2, b3 = a6 + a7 * r2;
    const double d0 = b0 + b1 * r4, d1 = b2 + b3 * r4;
    const double p = d0 + d1 * r8;
    const int ki = (int)k;
    if (ki >= -1021 && ki <= 1023) return p * raz_bits_to_f64((uint64_t)(ki + 1023) << 52);
    if (ki < -1021) return (p * raz_bits_to_f64((uint64_t)(ki + 2023) << 52)) * 0x1p-1000;
    return (p * 0x1p+1023) * raz_bits_to_f64((uint64_t)(ki) << 52);
}

RAZ_HD double raz_det_pow(double x, double y) {
    if (!(x > 0.0)) return 0.0;
    return raz_det_exp(y * raz_det_log(x));
}

RAZ_HD float raz_det_expf(float x) {
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float t = x * 0x1.715476p+0f + 0.5f;
    float k = (float)(int)t;
    if (k > t) k = k - 1.0f;
    const float r = (x - k * 0x1.63p-1f) - k * -0x1.bd0106p-13f;
    float p = 0x1.a01a02p-13f;
    p = p * r + 0x1.6c16c2p-10f;
    p = p * r + 0x1.111112p-7f;
    p = p * r + 0x1.555556p-5f;
    p = p * r + 0x1.555556p-3f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    return p * raz_bits_to_f32((uint32_t)((int)k + 127) << 23);
}

RAZ_HD float raz_det_tanhf(float x) {
    float ax = x < 0.0f ? -x : x;
    if (ax > 10.0f) ax = 10.0f;
    const float t = raz_det_expf(2.0f * ax);
    const float r = (t - 1.0f) / (t + 1.0f);
    return x < 0.0f ? -r : r;
}

// cos^2(2 pi u) for u in [0,1): exact octant reduction (u is a dyadic rational), then the Taylor
// series of sin on [0, pi/4] (9 terms, Estrin association), cos^2 = 1 - sin^2.
RAZ_HD double raz_det_cos2(double u) {
    const double w = 4.0 * u;            // exact
    const int q = (int)w;                // quadrant 0..3
    const double f = w - (double)q;      // exact, in [0,1)
    const bool swap = f > 0.5;
    const double gq = swap ? 1.0 - f : f;  // exact, in [0,0.5]
    const double x = gq * 0x1.921fb54442d18p+0;  // angle in [0, pi/4]
    const double z = x * x, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = 1.0 + -0x1.5555555555555p-3 * z;
    const double a1 = 0x1.1111111111111p-7 + -0x1.a01a01a01a01ap-13 * z;
    const double a2 = 0x1.71de3a556c734p-19 + -0x1.ae64567f544e4p-26 * z;
    const double a3 = 0x1.6124613a86d09p-33 + -0x1.ae7f3e733b81fp-41 * z;
    const double b0 = a0 + a1 * z2, b1 = a2 + a3 * z2;
    const double p = (b0 + b1 * z4) + 0x1.952c77030ad4ap-49 * z8;
    const double sn = x * p, s2 = sn * sn;
    const bool use_s2 = ((q & 1) != 0) != swap;
    return use_s2 ? s2 : 1.0 - s2;
}

// Two independent Gamma(1/2, 1) variates from one Philox block (d0, d1): with E = -log(1 - d0) ~ Exp(1)
// and theta = 2 pi d1, (sqrt(2E) cos theta, sqrt(2E) sin theta) are independent standard normals
// (Box-Muller) and Z^2/2 ~ Gamma(1/2): g0 = E cos^2 theta, g1 = E sin^2 theta.  No rejection loop.
// This is how dirichlet_alpha = 0.5 (config.py:138, every shipped config) is sampled: pair m
// (sub = m, idx = 0) yields samples 2m and 2m+1.
RAZ_HD void raz_gamma_half_pair(uint32_t seed, uint32_t game, uint32_t event, uint32_t m, double& g0, double& g1) {
    double d0, d1;
    raz_rng_pair(seed, game, RAZ_RNG_DIRICHLET, event, m, 0, d0, d1);
    const double E = -raz_det_log(1.0 - d0);
    const double c2 = raz_det_cos2(d1);
    g0 = E * c2;
    g1 = E * (1.0 - c2);
}

// One attempt `t` of the Gamma(alpha, 1) sampler for alpha != 0.5 (0.5 uses the pair construction above): alpha == 1 is
// an exponential (always accepted); alpha < 1 is the rejection scheme of numpy's legacy_standard_gamma for shape < 1;
// alpha > 1 its Marsaglia-Tsang scheme for shape > 1 (np.random.dirichlet -> legacy_standard_gamma, the sampler behind
// lib/bitboard.py:162-171).  Attempt t of sample `sub` of draw `event` uses the Philox block (seed, game, DIRICHLET, event,
// sub, t) - for alpha > 1 the two blocks (.., 2t) and (.., 2t + 1): a standard normal from the first (Box-Muller: |Z| =
// sqrt(2 E cos^2 theta)), its sign and the acceptance uniform from the second - so attempts can be evaluated in any order
// or in parallel; the sample is the accepted attempt with the smallest t.
RAZ_HD bool raz_gamma_attempt(double alpha, uint32_t seed, uint32_t game, uint32_t event, uint32_t sub,
                              uint32_t t, double& X) {
    double d0, d1;
    raz_rng_pair(seed, game, RAZ_RNG_DIRICHLET, event, sub, alpha > 1.0 ? 2u * t : t, d0, d1);
    if (alpha == 1.0) {
        X = -raz_det_log(1.0 - d0);
        return true;
    }
    if (alpha > 1.0) {
        double e0, e1;
        raz_rng_pair(seed, game, RAZ_RNG_DIRICHLET, event, sub, 2u * t + 1u, e0, e1);
        const double E = -raz_det_log(1.0 - d0);
        const double z2 = 2.0 * (E * raz_det_cos2(d1));   // Z^2, Z ~ N(0, 1)
        const double az = sqrt(z2);
        const double Z = e0 < 0.5 ? -az : az;
        const double b = alpha - 0x1.5555555555555p-2;    // alpha - 1/3
        const double c = 1.0 / sqrt(9.0 * b);
        double V = 1.0 + c * Z;
        X = 0.0;
        if (V <= 0.0) return false;
        V = (V * V) * V;
        X = b * V;
        const double U = 1.0 - e1;                        // uniform on (0, 1]
        if (U < 1.0 - 0.0331 * (z2 * z2)) return true;
        return raz_det_log(U) < 0.5 * z2 + b * ((1.0 - V) + raz_det_log(V));
    }
    const double inv = 1.0 / alpha;
    const double U = d0, V = -raz_det_log(1.0 - d1);
    if (U <= 1.0 - alpha) {
        X = raz_det_pow(U, inv);
        return X <= V;
    }
    const double Y = -raz_det_log((1.0 - U) / alpha);
    X = raz_det_pow(1.0 - alpha + alpha * Y, inv);
    return X <= V + Y;
}

RAZ_HD double raz_gamma_sample(double alpha, uint32_t seed, uint32_t game, uint32_t event, uint32_t sub) {
    double X;
    for (uint32_t t = 0;; ++t)
        if (raz_gamma_attempt(alpha, seed, game, event, sub, t, X)) return X;
}
