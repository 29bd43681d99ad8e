// probe_numerics.hip — one-off hardware probe (not product): are the IEEE ops the deterministic
// search arithmetic relies on bit-identical between gfx950 and the host?  Checks f64 sqrt/div/add,
// f32 div/fma, and that v_mfma_f32_16x16x4_f32 accumulation equals a k-ordered fmaf chain.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__global__ void k_f64(const double* a, const double* b, double* sq, double* dv, double* ad, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        sq[i] = sqrt(a[i]);
        dv[i] = a[i] / b[i];
        ad[i] = (a[i] - 3.0) + (3.0 + b[i]);
    }
}
__global__ void k_f32(const float* a, const float* b, const float* c, float* dv, float* fm, float* sq, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dv[i] = a[i] / b[i];
        fm[i] = fmaf(a[i], b[i], c[i]);
        sq[i] = sqrtf(fabsf(a[i]));
    }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// C[16x16] = A[16xK] * B[Kx16] + bias, K multiple of 4, one wave.
__global__ void k_mfma(const float* A, const float* B, const float* bias, float* C, int K) {
    int l = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = bias[((l >> 4) * 4 + r) * 16 + (l & 15)];
    for (int k0 = 0; k0 < K; k0 += 4) {
        float a = A[(l & 15) * K + k0 + (l >> 4)];
        float b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma32(const float* A, const float* B, float* C, int K) {  // 32x32, K mult of 2
    int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) {
        float a = A[(l & 31) * K + k0 + (l >> 5)];
        float b = B[(k0 + (l >> 5)) * 32 + (l & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        C[row * 32 + (l & 31)] = acc[r];
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int n = 1 << 22;
    std::vector<double> a(n), b(n), sq(n), dv(n), ad(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        if (i < (1 << 21)) { a[i] = (double)i; b[i] = (double)(i % 977 + 1) + 1e-5; }
        else { a[i] = ldexp((double)rand() / RAND_MAX, rand() % 40 - 20); b[i] = ldexp((double)rand() / RAND_MAX + 1e-9, rand() % 40 - 20); }
    }
    double *da, *db, *dsq, *ddv, *dad;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&dsq, n * 8)); CK(hipMalloc(&ddv, n * 8)); CK(hipMalloc(&dad, n * 8));
    CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
    k_f64<<<n / 256, 256>>>(da, db, dsq, ddv, dad, n);
    CK(hipMemcpy(sq.data(), dsq, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(dv.data(), ddv, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ad.data(), dad, n * 8, hipMemcpyDeviceToHost));
    long bad_sq = 0, bad_dv = 0, bad_ad = 0;
    for (int i = 0; i < n; ++i) {
        double s = sqrt(a[i]), d = a[i] / b[i]; volatile double t1 = a[i] - 3.0; volatile double t2 = 3.0 + b[i]; double e = t1 + t2;
        bad_sq += memcmp(&s, &sq[i], 8) != 0; bad_dv += memcmp(&d, &dv[i], 8) != 0; bad_ad += memcmp(&e, &ad[i], 8) != 0;
    }
    printf("f64: sqrt mismatches %ld, div mismatches %ld, add-chain mismatches %ld of %d\n", bad_sq, bad_dv, bad_ad, n);

    std::vector<float> fa(n), fb(n), fc(n), fdv(n), ffm(n), fsq(n);
    for (int i = 0; i < n; ++i) { fa[i] = ((float)rand() / RAND_MAX - 0.5f) * 8; fb[i] = ((float)rand() / RAND_MAX) * 4 + 1e-6f; fc[i] = ((float)rand() / RAND_MAX - 0.5f); }
    float *xa, *xb, *xc, *xdv, *xfm, *xsq;
    CK(hipMalloc(&xa, n * 4)); CK(hipMalloc(&xb, n * 4)); CK(hipMalloc(&xc, n * 4)); CK(hipMalloc(&xdv, n * 4)); CK(hipMalloc(&xfm, n * 4)); CK(hipMalloc(&xsq, n * 4));
    CK(hipMemcpy(xa, fa.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, fb.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(xc, fc.data(), n * 4, hipMemcpyHostToDevice));
    k_f32<<<n / 256, 256>>>(xa, xb, xc, xdv, xfm, xsq, n);
    CK(hipMemcpy(fdv.data(), xdv, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ffm.data(), xfm, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(fsq.data(), xsq, n * 4, hipMemcpyDeviceToHost));
    long bd = 0, bf = 0, bs = 0;
    for (int i = 0; i < n; ++i) {
        float d = fa[i] / fb[i], f = fmaf(fa[i], fb[i], fc[i]), s = sqrtf(fabsf(fa[i]));
        bd += memcmp(&d, &fdv[i], 4) != 0; bf += memcmp(&f, &ffm[i], 4) != 0; bs += memcmp(&s, &fsq[i], 4) != 0;
    }
    printf("f32: div mismatches %ld, fma mismatches %ld, sqrt mismatches %ld of %d\n", bd, bf, bs, n);

    for (int K : {4, 64, 144, 2304}) {
        std::vector<float> A(16 * K), B(K * 16), bias(256), C(256);
        for (auto& v : A) v = ((float)rand() / RAND_MAX - 0.5f) * 2; for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 2; for (auto& v : bias) v = ((float)rand() / RAND_MAX - 0.5f);
        float *dA, *dB, *dbias, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dbias, 1024)); CK(hipMalloc(&dC, 1024));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), 1024, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dbias, dC, K);
        CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float acc = bias[i * 16 + j];
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
            bad += memcmp(&acc, &C[i * 16 + j], 4) != 0;
        }
        printf("mfma 16x16x4 f32, K=%d: %ld / 256 outputs differ from the k-ordered fmaf chain\n", K, bad);
        hipFree(dA); hipFree(dB); hipFree(dbias); hipFree(dC);
    }
    for (int K : {2, 64, 2304}) {
        std::vector<float> A(32 * K), B(K * 32), C(1024);
        for (auto& v : A) v = ((float)rand() / RAND_MAX - 0.5f) * 2; for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 2;
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        k_mfma32<<<1, 64>>>(dA, dB, dC, K);
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
            bad += memcmp(&acc, &C[i * 32 + j], 4) != 0;
        }
        printf("mfma 32x32x2 f32, K=%d: %ld / 1024 outputs differ from the k-ordered fmaf chain\n", K, bad);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
