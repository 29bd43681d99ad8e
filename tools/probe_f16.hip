// probe_f16.hip — one-off hardware probe (not product): numerics of v_mfma_f32_32x32x16_f16 / _bf16 on gfx950.
//   (1) operand layout A[i = l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][n = l&31], j = 0..7 (checked with integers);
//   (2) how the 16 products and C are summed: compared against (H1) exact sum, one round-to-nearest-even;
//       (H2) a k-ordered fmaf chain; (H3) exact sum of the 16 products rounded, then + C rounded;
//       (H4) two exact half sums (k 0..7, 8..15) ...  Raw cases are dumped for offline analysis;
//   (3) f16 subnormal inputs: kept or flushed.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_f16.hip -o tools/probe_f16 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32][16] halfs row-major, B: [16][32], C/D: [32][32] floats
template <int BF>
__global__ void k_mfma(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    s8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (short)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    if (BF) {
        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), acc, 0, 0, 0);
    } else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float rnd() { return (float)rand() / RAND_MAX; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* dump = argc > 1 ? argv[1] : nullptr;
    FILE* fd = dump ? fopen(dump, "wb") : nullptr;
    uint16_t *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, 32 * 16 * 2)); CK(hipMalloc(&dB, 16 * 32 * 2)); CK(hipMalloc(&dC, 4096)); CK(hipMalloc(&dD, 4096));
    std::vector<uint16_t> A(512), B(512);
    std::vector<float> C(1024), D(1024);
    for (int bf = 0; bf < 2; ++bf) {
        auto enc = bf ? f2bf : f2h;
        auto dec = bf ? bf2f : h2f;
        const char* nm = bf ? "bf16" : "f16";
        // (1) layout: small integers, asymmetric
        for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = enc((float)((i * 3 + k * 5) % 7 - 3));
        for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = enc((float)((k * 2 + n * 7) % 5 - 2));
        for (int i = 0; i < 1024; ++i) C[i] = (float)(i % 11);
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
        if (bf) k_mfma<1><<<1, 64>>>(dA, dB, dC, dD); else k_mfma<0><<<1, 64>>>(dA, dB, dC, dD);
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
            float s = C[i * 32 + n];
            for (int k = 0; k < 16; ++k) s += dec(A[i * 16 + k]) * dec(B[k * 32 + n]);
            bad += s != D[i * 32 + n];
        }
        printf("%s layout check (integers): %d / 1024 wrong\n", nm, bad);
        // (2) summation hypotheses over several magnitude regimes
        for (int regime = 0; regime < 5; ++regime) {
            long m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, tot = 0;
            double maxrel = 0;
            for (int rep = 0; rep < 48; ++rep) {
                for (auto& v : A) { float x = (rnd() - 0.5f) * 2; if (regime == 1) x = ldexpf(x, rand() % 12 - 6); if (regime == 2) x = ldexpf(x, rand() % 24 - 12); if (regime == 3) x = fabsf(x); v = enc(x); }
                for (auto& v : B) { float x = (rnd() - 0.5f) * 2; if (regime == 1) x = ldexpf(x, rand() % 12 - 6); if (regime == 2) x = ldexpf(x, rand() % 8 - 4); if (regime == 3) x = fabsf(x); v = enc(x); }
                for (auto& v : C) { v = (rnd() - 0.5f) * (regime == 4 ? 2000.f : 8.f); if (regime == 3) v = fabsf(v); }
                CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
                if (bf) k_mfma<1><<<1, 64>>>(dA, dB, dC, dD); else k_mfma<0><<<1, 64>>>(dA, dB, dC, dD);
                CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
                if (fd) { int hdr[2] = {bf, regime}; fwrite(hdr, 4, 2, fd); fwrite(A.data(), 2, 512, fd); fwrite(B.data(), 2, 512, fd); fwrite(C.data(), 4, 1024, fd); fwrite(D.data(), 4, 1024, fd); }
                for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
                    __float128 ex = C[i * 32 + n], pe = 0, lo = 0, hi = 0;
                    float ch = C[i * 32 + n];
                    for (int k = 0; k < 16; ++k) {
                        const float a = dec(A[i * 16 + k]), b = dec(B[k * 32 + n]);
                        const __float128 p = (__float128)a * (__float128)b;
                        ex += p; pe += p; (k < 8 ? lo : hi) += p;
                        ch = fmaf(a, b, ch);
                    }
                    const float h1 = (float)ex;
                    const float h3 = (float)((__float128)(float)pe + (__float128)C[i * 32 + n]);
                    const float h4 = (float)((__float128)(float)((__float128)C[i * 32 + n] + lo) + hi);
                    const float h5 = (float)((__float128)(float)lo + (__float128)(float)hi + (__float128)C[i * 32 + n]);
                    const float d = D[i * 32 + n];
                    m1 += memcmp(&h1, &d, 4) != 0; m2 += memcmp(&ch, &d, 4) != 0; m3 += memcmp(&h3, &d, 4) != 0;
                    m4 += memcmp(&h4, &d, 4) != 0; m5 += memcmp(&h5, &d, 4) != 0; ++tot;
                    const double rel = fabs((double)d - (double)ex) / (fabs((double)ex) + 1e-30);
                    if (rel > maxrel) maxrel = rel;
                }
            }
            printf("%s regime %d: of %ld outputs differ from  H1 exact-sum-one-RNE %ld | H2 fmaf chain %ld | H3 round(sum16)+C %ld | H4 (C+lo8 rounded)+hi8 %ld | H5 round(lo)+round(hi)+C %ld ; max rel err vs exact %.3g\n",
                   nm, regime, tot, m1, m2, m3, m4, m5, maxrel);
        }
        // (3) subnormal inputs (f16 only meaningful): a = 2^-20 (subnormal in f16), b = 2^10 -> 2^-10 if kept
        for (auto& v : A) v = 0; for (auto& v : B) v = 0; for (auto& v : C) v = 0;
        A[0] = enc(ldexpf(1.f, bf ? -130 : -20)); B[0] = enc(1024.f);
        A[16] = enc(1.0f); B[1] = enc(ldexpf(1.f, bf ? -130 : -20));   // output (1,1): 1 * subnormal
        B[32 + 1] = 0;
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice));
        if (bf) k_mfma<1><<<1, 64>>>(dA, dB, dC, dD); else k_mfma<0><<<1, 64>>>(dA, dB, dC, dD);
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        printf("%s subnormal input a=%g * 1024 -> D[0][0] = %g (exact %g);  1 * subnormal b -> D[1][1] = %g\n", nm, dec(A[0]), D[0], dec(A[0]) * 1024.0, D[33]);
    }
    if (fd) fclose(fd);
    return 0;
}
