// probe_conv.hip — stand-alone measurement harness (not product) for the headline conv kernel k_conv3x3_f16x3
// (reversi-alpha-zero_amd/csrc/raz_net_f16x3.hip, included here with its timeline instrumentation compiled in).  No torch, no
// Python: starts in a second on the GPU box.  One JSON line per experiment:
//   time     ms per layer launch over `reps` launches, on random operands and on ZERO operands (the DVFS give-back test of
//            MI355X_MICROARCH.md: same instruction stream, no switching activity - what the power limit costs), for several
//            position counts (tile-round quantisation: 2 x ceil(n / 8) tiles over 256 CUs)
//   stamps   one launch with per-wave timestamps: where a workgroup's time goes (start-up until stage 0 has landed, K loop,
//            barrier wait inside the loop, epilogue, store drain), the gap a CU shows between two consecutive workgroups, the
//            shader clock (s_memtime ticks per s_memrealtime tick at 100 MHz)
// Build (cross-compiles without a GPU):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe_conv.hip -o tools/probe_conv
// -DPROBE_WINO builds tools/probe_wino instead: the same `time` / `pmc` experiments on k_conv3x3_wino (the Winograd experiment,
// tools/experiments/raz_net_wino.hip: 4 positions x 128 output channels per workgroup, transformed activations in and out).
#ifdef PROBE_WINO
#include "experiments/raz_net_wino.hip"
unsigned* raz_net_f16x3_flag(const float*, int, int, int) { return nullptr; }
int raz_net_heads_split(const float*, int, int, int, const unsigned char*, const uint8_t*, float*, float*, size_t, hipStream_t, const uint32_t*, const uint32_t*) { return 0; }
#define IN_BYTES_PER_F 512
#define W_TAPS 12
#define POS_PER_WG NPOS
#else
#define RAZ_F16X3_STAMPS 1
#include "../reversi-alpha-zero_amd/csrc/raz_net_f16x3.hip"
#define IN_BYTES_PER_F 256
#define W_TAPS 9
#define POS_PER_WG NWAVE
// (the forward's range repair lives in raz_net.hip, which the probe does not link)
int raz_net_repair_rows(const float*, int, int, int, const uint64_t*, const uint64_t*, float*, float*, size_t, unsigned*, unsigned*, const uint32_t*, const uint32_t*, hipStream_t) { return 0; }
#endif

#include <stdio.h>
#include <time.h>
#include <algorithm>
#include <map>

int raz_fail(int code, const char* msg) { fprintf(stderr, "raz_fail %d: %s\n", code, msg); return code; }
int raz_fail_hip(hipError_t e, const char* where) { fprintf(stderr, "hip error %s at %s\n", hipGetErrorString(e), where); return RAZ_EDEVICE; }
int raz_check_launch(const char* where) { hipError_t e = hipGetLastError(); return e == hipSuccess ? RAZ_OK : raz_fail_hip(e, where); }

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static float urand() { return (rnd() >> 8) * (1.0f / 16777216.0f); }

// split images in the kernel's layouts: a 16-byte unit holds 8 halfs; units come in (hi, lo) plane pairs `stride` units apart
// (activations: planes of 64 squares, weights: planes of 128 output channels)
template <class Gen>
static void fill_split(std::vector<_Float16>& v, size_t units, size_t stride, Gen gen) {
    v.assign(units * 8, (_Float16)0.f);
    for (size_t u = 0; u < units; ++u) {
        if ((u / stride) & 1) continue;   // a lo plane: written with its hi plane
        for (int j = 0; j < 8; ++j) {
            const float x = gen();
            const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
            v[u * 8 + j] = hi;
            v[(u + stride) * 8 + j] = lo;
        }
    }
}

int main(int argc, char** argv) {
    const int F = 256;
    int nmax = 8192, reps = 20;
    const char* mode = argc > 1 ? argv[1] : "all";
    if (argc > 2) reps = atoi(argv[2]);
    const size_t pos_bytes = (size_t)F * IN_BYTES_PER_F;
    const size_t wl_bytes = (size_t)F * F * W_TAPS * 4;
    unsigned char *dW, *dWz, *dA, *dB, *dAz;
    float *dbias, *dscale;
    unsigned* dflag;
    unsigned long long* dstamps;
    CK(hipMalloc(&dW, wl_bytes));
    CK(hipMalloc(&dWz, wl_bytes));
    CK(hipMalloc(&dA, nmax * pos_bytes));
    CK(hipMalloc(&dAz, nmax * pos_bytes));
    CK(hipMalloc(&dB, nmax * pos_bytes));
    CK(hipMalloc(&dbias, F * 4));
    CK(hipMalloc(&dscale, 4));
    CK(hipMalloc(&dflag, (size_t)nmax * 64 + 64));   // a 64-byte range-flag area per row (raz_internal.h)
    const size_t nblocks = (size_t)nmax / POS_PER_WG * 2;
    CK(hipMalloc(&dstamps, nblocks * 8 * 8 * 8));
    {
        std::vector<_Float16> h;
        // weights: uniform in +-2^15 after scaling; activations: post-ReLU, half of them zero (the value statistics set the switching activity)
        fill_split(h, wl_bytes / 16, W_TAPS == 12 ? 64 : 128, [] { return (urand() * 2.f - 1.f) * 32000.f; });
        CK(hipMemcpy(dW, h.data(), wl_bytes, hipMemcpyHostToDevice));
        CK(hipMemset(dWz, 0, wl_bytes));
#ifdef PROBE_WINO
        // transformed activations: sums / differences of two post-ReLU values (a quarter of them zero); planes of 32 columns
        fill_split(h, nmax * pos_bytes / 16, 32, [] { const float a = urand() < 0.5f ? 0.f : urand() * 2.0f, b = urand() < 0.5f ? 0.f : urand() * 2.0f; return urand() < 0.5f ? a + b : a - b; });
#else
        fill_split(h, nmax * pos_bytes / 16, 64, [] { return urand() < 0.5f ? 0.f : urand() * 2.0f; });
#endif
        CK(hipMemcpy(dA, h.data(), nmax * pos_bytes, hipMemcpyHostToDevice));
        CK(hipMemset(dAz, 0, nmax * pos_bytes));
        std::vector<float> b(F, 0.01f);
        CK(hipMemcpy(dbias, b.data(), F * 4, hipMemcpyHostToDevice));
        const float sc = 1.0f / 32768.f / 2304.f;   // keeps the outputs O(1): no range flag
        CK(hipMemcpy(dscale, &sc, 4, hipMemcpyHostToDevice));
        CK(hipMemset(dflag, 0, (size_t)nmax * 64 + 64));
    }
#ifdef PROBE_WINO
    unsigned char* dP;
    CK(hipMalloc(&dP, (size_t)nmax * F * 256));
    CK(hipFuncSetAttribute((const void*)k_conv3x3_wino, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#else
    CK(hipFuncSetAttribute((const void*)k_conv3x3_f16x3, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#endif
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&](const unsigned char* W, const unsigned char* in, int n, unsigned long long* st) {
        const unsigned groups = (unsigned)((n + POS_PER_WG - 1) / POS_PER_WG);
        const unsigned tiles = ((groups + 7) / 8) * 8 * (unsigned)(F / 128);
#ifdef PROBE_WINO
        // a block's second convolution: transformed in, transformed + plain out (the heaviest form; no skip read)
        hipLaunchKernelGGL(k_conv3x3_wino, dim3(tiles), dim3(NWAVE * 64), LDS_BYTES, s, W, dbias, dscale, in, dB, dP, (const unsigned char*)nullptr,
                           (const uint8_t*)nullptr, n, F, dflag, (const uint32_t*)nullptr);
        (void)st;
#else
        hipLaunchKernelGGL(k_conv3x3_f16x3, dim3(tiles), dim3(NWAVE * 64), LDS_BYTES, s, W, dbias, dscale, in, dB, (const unsigned char*)nullptr,
                           (const uint8_t*)nullptr, n, F, dflag, (const uint32_t*)nullptr, st);
#endif
    };
    auto timeit = [&](const unsigned char* W, const unsigned char* in, int n) {
        for (int i = 0; i < 3; ++i) launch(W, in, n, nullptr);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch(W, in, n, nullptr);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    const bool all = !strcmp(mode, "all");
    if (all || !strcmp(mode, "time")) {
        const int ns[] = {8192, 7680, 7501, 7168, 6144, 4096, 2048, 1024};
        for (int n : ns) {
            const float r = timeit(dW, dA, n), z = timeit(dWz, dAz, n), r2 = timeit(dW, dA, n);
            const double flop = 2.0 * n * 64.0 * F * F * 9;
            printf("{\"experiment\": \"time\", \"positions\": %d, \"tiles\": %u, \"tile_rounds_over_256_cus\": %.3f, \"ms_random\": %.4f, \"ms_random_again\": %.4f, "
                   "\"ms_zero_operands\": %.4f, \"algorithmic_tflops_random\": %.1f, \"executed_f16_tflops_random\": %.1f, \"zero_over_random_speedup\": %.3f}\n",
                   n, (unsigned)((n + POS_PER_WG - 1) / POS_PER_WG * 2), (n + POS_PER_WG - 1) / POS_PER_WG * 2 / 256.0, r, r2, z, flop / (r * 1e-3) / 1e12,
                   3 * flop * W_TAPS / 9.0 * (W_TAPS == 12 ? 0.5 : 1.0) / (r * 1e-3) / 1e12, r / z);
            fflush(stdout);
        }
    }
#ifdef PROBE_WINO
    if (all || !strcmp(mode, "check")) {
        // one launch against a double-precision restatement of the layer as a function of the SAME operand images (the split halves
        // summed): data movement, the ring, the tile swap and both output forms, for a few positions x all 256 channels
        const int n = 8190;   // a ragged last group
        std::vector<_Float16> hW(wl_bytes / 2), hV((size_t)nmax * pos_bytes / 2);
        CK(hipMemcpy(hW.data(), dW, wl_bytes, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hV.data(), dA, (size_t)nmax * pos_bytes, hipMemcpyDeviceToHost));
        CK(hipMemset(dB, 0xff, (size_t)nmax * pos_bytes));
        CK(hipMemset(dP, 0xff, (size_t)nmax * F * 256));
        launch(dW, dA, n, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<_Float16> oV((size_t)nmax * pos_bytes / 2), oP((size_t)nmax * F * 128);
        CK(hipMemcpy(oV.data(), dB, oV.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(oP.data(), dP, oP.size() * 2, hipMemcpyDeviceToHost));
        float hsc;
        CK(hipMemcpy(&hsc, dscale, 4, hipMemcpyDeviceToHost));
        auto U = [&](int a, int oc, int ic, int dy) {
            const size_t o = (((((size_t)(oc >> 7) * 16 + (ic >> 4)) * 3 + dy) * 4 + ((oc >> 5) & 3)) * 4 + a) * 2 * 512 + (size_t)(((ic >> 3) & 1) * 32 + (oc & 31)) * 8 + (ic & 7);
            return (double)(float)hW[o] + (double)(float)hW[o + 512];
        };
        auto Vin = [&](int pos, int a, int ic, int col) {
            const size_t o = (size_t)pos * F * 256 + (size_t)(ic >> 4) * 4096 + (size_t)((a * 4 + ((ic >> 3) & 1) * 2) * 32 + col) * 8 + (ic & 7);
            return (double)(float)hV[o] + (double)(float)hV[o + 256];
        };
        double worst_p = 0, worst_v = 0, biggest = 0;
        size_t untouched = 0;
        const int probe_pos[] = {0, 1, 2, 3, 4093, 8189};
        for (int pos : probe_pos) {
            std::vector<double> e(F * 32), o(F * 32);
            for (int oc = 0; oc < F; ++oc)
                for (int col = 0; col < 32; ++col) {
                    double M[4] = {0, 0, 0, 0};
                    for (int a = 0; a < 4; ++a)
                        for (int dy = 0; dy < 3; ++dy) {
                            const int yy = (col >> 2) + dy - 1;
                            if (yy < 0 || yy > 7) continue;
                            for (int ic = 0; ic < F; ++ic) M[a] += U(a, oc, ic, dy) * Vin(pos, a, ic, yy * 4 + (col & 3));
                        }
                    const double ev = (M[0] + M[1] + M[2]) * hsc + 0.01, ov = (M[1] - M[2] - M[3]) * hsc + 0.01;
                    e[oc * 32 + col] = ev > 0 ? ev : 0;
                    o[oc * 32 + col] = ov > 0 ? ov : 0;
                    biggest = std::max(biggest, std::max(e[oc * 32 + col], o[oc * 32 + col]));
                }
            for (int oc = 0; oc < F; ++oc)
                for (int col = 0; col < 32; ++col) {
                    const size_t po = (size_t)pos * F * 128 + (size_t)(oc >> 4) * 2048 + (size_t)(((oc >> 3) & 1) * 2 * 64 + 2 * col) * 8 + (oc & 7);
                    const double ge = (double)(float)oP[po] + (double)(float)oP[po + 512], go = (double)(float)oP[po + 8] + (double)(float)oP[po + 8 + 512];
                    worst_p = std::max(worst_p, std::max(fabs(ge - e[oc * 32 + col]), fabs(go - o[oc * 32 + col])));
                    const int t = col & 3;
                    const double L = t ? o[oc * 32 + col - 1] : 0.0, R = t < 3 ? e[oc * 32 + col + 1] : 0.0, E = e[oc * 32 + col], O = o[oc * 32 + col];
                    const double want[4] = {L - O, E + O, O - E, E - R};
                    for (int a = 0; a < 4; ++a) {
                        const size_t vo = (size_t)pos * F * 256 + (size_t)(oc >> 4) * 4096 + (size_t)((a * 4 + ((oc >> 3) & 1) * 2) * 32 + col) * 8 + (oc & 7);
                        const double gv = (double)(float)oV[vo] + (double)(float)oV[vo + 256];
                        if (gv != gv) ++untouched;
                        worst_v = std::max(worst_v, fabs(gv - want[a]));
                    }
                }
        }
        // rows beyond n must stay untouched (0xffff halfs = NaN)
        const float beyond = (float)oP[(size_t)8191 * F * 128 + 5];
        printf("{\"experiment\": \"check\", \"positions_checked\": 6, \"largest_output\": %.4f, \"max_abs_err_plain\": %.3e, \"max_abs_err_transformed\": %.3e, "
               "\"nan_outputs\": %zu, \"row_beyond_n_untouched\": %s, \"ok\": %s}\n", biggest, worst_p, worst_v, untouched, beyond != beyond ? "true" : "false",
               (worst_p < 1e-4 * std::max(1.0, biggest) && worst_v < 2e-4 * std::max(1.0, biggest) && !untouched && beyond != beyond) ? "true" : "false");
        fflush(stdout);
    }
#endif
    if (!strcmp(mode, "pmc")) {   // the launches a counter pass looks at: 8192 positions, random operands (argv[3] = positions)
        const int n = argc > 3 ? atoi(argv[3]) : 8192;
        const float r = timeit(dW, dA, n);
        printf("{\"experiment\": \"pmc\", \"positions\": %d, \"ms_random\": %.4f}\n", n, r);
    }
    if (!strcmp(mode, "power")) {   // sustained phases for the SMI sampler (tools/smi_sampler.py): argv[2] = seconds per phase, argv[3] = positions
        const double secs = argc > 2 ? atof(argv[2]) : 6.0;
        const int n = argc > 3 ? atoi(argv[3]) : 8192;
        auto wall = [] { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
        const char* names[] = {"idle", "random_operands", "zero_operands", "random_operands_again", "idle_after"};
        for (int ph = 0; ph < 5; ++ph) {
            const double t0 = wall();
            long launches = 0;
            double gpu_ms = 0;
            if (ph == 0 || ph == 4) {
                timespec req = {(time_t)(secs / 2), (long)((secs / 2 - (time_t)(secs / 2)) * 1e9)};
                nanosleep(&req, nullptr);
            } else {
                const unsigned char *W = ph == 2 ? dWz : dW, *A = ph == 2 ? dAz : dA;
                while (wall() - t0 < secs) {   // chunks of 100 launches (~0.12 s), timed with events: ms per launch over the phase
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < 100; ++i) launch(W, A, n, nullptr);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    gpu_ms += ms;
                    launches += 100;
                }
            }
            const double t1 = wall();
            printf("{\"experiment\": \"power\", \"phase\": \"%s\", \"positions\": %d, \"t_start_unix\": %.4f, \"t_end_unix\": %.4f, \"launches\": %ld, \"ms_per_launch\": %.4f}\n",
                   names[ph], n, t0, t1, launches, launches ? gpu_ms / launches : 0.0);
            fflush(stdout);
        }
    }
#ifndef PROBE_WINO
    if (all || !strcmp(mode, "stamps")) {
        for (int pass = 0; pass < 2; ++pass) {
            const int n = pass == 0 ? 8192 : 7501;
            for (int i = 0; i < 3; ++i) launch(dW, dA, n, nullptr);   // warm clocks and caches with un-instrumented-output launches
            CK(hipMemsetAsync(dstamps, 0, nblocks * 8 * 8 * 8, s));
            CK(hipEventRecord(e0, s));
            launch(dW, dA, n, dstamps);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> st(nblocks * 64);
            CK(hipMemcpy(st.data(), dstamps, st.size() * 8, hipMemcpyDeviceToHost));
            // per wave
            double sum_pro = 0, sum_loop = 0, sum_bar = 0, sum_epi = 0, sum_drain = 0, sum_total = 0;
            size_t waves = 0;
            unsigned long long tmin = ~0ull, tmax = 0, rmin = ~0ull, rmax = 0;
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;   // cu key -> (start, end) of workgroups
            for (size_t b = 0; b < nblocks; ++b) {
                unsigned long long bs = ~0ull, be = 0, key = 0;
                bool any = false;
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* t = &st[(b * 8 + w) * 8];
                    if (!t[4]) continue;
                    any = true;
                    sum_pro += (double)(t[1] - t[0]);
                    sum_loop += (double)(t[2] - t[1]);
                    sum_bar += (double)t[5];
                    sum_epi += (double)(t[3] - t[2]);
                    sum_drain += (double)(t[4] - t[3]);
                    sum_total += (double)(t[4] - t[0]);
                    ++waves;
                    tmin = std::min(tmin, t[0]);
                    tmax = std::max(tmax, t[4]);
                    rmin = std::min(rmin, t[7]);
                    rmax = std::max(rmax, t[7]);
                    bs = std::min(bs, t[0]);
                    be = std::max(be, t[4]);
                    // HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13; XCC_ID 3:0 in the high word
                    const unsigned hw = (unsigned)t[6], xcc = (unsigned)(t[6] >> 32) & 15;
                    key = ((unsigned long long)xcc << 16) | (hw & 0xff00);   // xcc, se, sh, cu
                }
                if (any) per_cu[key].push_back({bs, be});
            }
            double gap_sum = 0, busy_sum = 0;
            size_t gaps = 0, max_wg = 0, min_wg = ~(size_t)0;
            for (auto& kv : per_cu) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end());
                max_wg = std::max(max_wg, v.size());
                min_wg = std::min(min_wg, v.size());
                for (size_t i = 0; i < v.size(); ++i) {
                    busy_sum += (double)(v[i].second - v[i].first);
                    if (i) { gap_sum += (double)((long long)v[i].first - (long long)v[i - 1].second); ++gaps; }
                }
            }
            const double span = (double)(tmax - tmin);
            const double mfma_cycles_per_wave = 48.0 * 72 * 32;   // issue time of a wave's matrix instructions on its SIMD (2 waves share it)
            printf("{\"experiment\": \"stamps\", \"positions\": %d, \"launch_ms_event\": %.4f, \"waves_stamped\": %zu, \"cus_seen\": %zu, \"workgroups_per_cu_min_max\": [%zu, %zu], "
                   "\"kernel_span_shader_cycles\": %.0f, \"shader_clock_ghz_from_event_time\": %.3f, \"realtime_ticks_span_100mhz\": %llu, "
                   "\"per_wave_mean_cycles\": {\"start_until_stage0_landed\": %.0f, \"k_loop\": %.0f, \"of_which_inside_barriers\": %.0f, \"epilogue_until_last_store_issued\": %.0f, "
                   "\"store_drain\": %.0f, \"total\": %.0f}, \"mfma_issue_cycles_per_simd_per_workgroup\": %.0f, \"k_loop_over_mfma_issue\": %.3f, "
                   "\"mean_gap_between_workgroups_on_a_cu_cycles\": %.0f, \"cu_busy_fraction_of_span\": %.3f}\n",
                   n, ms, waves, per_cu.size(), min_wg, max_wg, span, span / (ms * 1e-3) / 1e9, rmax - rmin, sum_pro / waves, sum_loop / waves, sum_bar / waves,
                   sum_epi / waves, sum_drain / waves, sum_total / waves, 2 * mfma_cycles_per_wave, (sum_loop / waves) / (2 * mfma_cycles_per_wave),
                   gaps ? gap_sum / gaps : 0.0, busy_sum / (per_cu.size() * span));
            fflush(stdout);
            if (pass == 0) {
                FILE* f = fopen("gpurun_out/probe_conv_stamps_8192.bin", "wb");
                if (f) { fwrite(st.data(), 8, st.size(), f); fclose(f); }
            }
        }
    }
#endif
    return 0;
}
