// litmus_slot_handoff.hip — hardware litmus (not product): what a wave of ANOTHER kernel on ANOTHER stream sees of a 32-byte slot
// that one lane publishes as "payload words, then a tag" - the hand-off csrc/raz_engine_core.h (memo_claim_and_write, solver_solve's
// request header) and csrc/raz_solver_pool.h (the answer word, the workers' memo entries) rely on when the solver pool's round runs
// beside the tree launches (raz_engine.hip pool_every > 1).  VERDICT r5 #3 / ADVICE r5 (medium).
//
// Two kernels run CONCURRENTLY on two streams, NB single-wave workgroups each (NB <= 128: both grids are co-resident on any MI355X).
// Publisher block b owns slots [b][0..R).  Reader block b' = (b + shift) % NB reads them: block i of a grid lands on XCD i % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"), so shift = 0 pairs same-XCD blocks of the two grids and shift = 1 cross-XCD ones.
// The reader first reads every slot it will look at (plain loads: its CU's L1 and its XCD's L2 now hold the EMPTY slot - the state a
// memo probe that missed a moment ago leaves behind), raises its `warm` flag; the publisher waits for it and then publishes slot r:
//   form 0 "wave fence" (rounds 1-5 of this repository):  plain stores of the two key words; fence(release, "wavefront");
//                                                         relaxed agent-scope store of the tag
//   form 1 "sc1 words":      relaxed agent-scope (sc1, write-through) 8-byte stores of the keys; s_waitcnt vmcnt(0); the same tag store
//   form 2 "agent release":  plain key stores; fence(release, "agent"); s_waitcnt vmcnt(0); the same tag store
// and the reader polls the tag with relaxed agent-scope loads (bounded: never a hang), then reads the keys
//   read 0: plain loads                       read 1: relaxed agent-scope (sc1) loads        read 2: fence(acquire, "agent") + plain loads
// and counts slots whose keys are not the published ones although the tag was ("stale": in the product a memo MISS, in the request
// header a WRONG POSITION), plus slots whose tag never arrived within the poll budget ("late").
// A second experiment per form: `single look` - the reader does NOT poll; it looks once at every slot with the product's former plain
// loads (tag and keys requested together), a fixed delay after the publisher finished and drained: "unseen" = the tag itself was not
// visible to a kernel that was already running (the price of plain loads on the reading side: an answer seen one launch later).
//
// Build: hipcc --offload-arch=gfx950 -O2 tools/litmus_slot_handoff.hip -o tools/litmus_slot_handoff ; run on the GPU box: one JSON document.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct Slot { unsigned long long black, white; uint32_t tag, link; unsigned long long pad; };   // csrc/raz_engine.h raz_slot
static_assert(sizeof(Slot) == 32, "slot");

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned long long key_of(uint32_t b, uint32_t r, uint32_t epoch) {
    unsigned long long x = ((unsigned long long)(b * 8191u + r) << 20) ^ (0x9E3779B97F4A7C15ULL * (epoch + 1u));
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    return x | 1ULL;
}

// PLAIN loads as the product's `s->idx_tag` compiles to (global_load_dword / dwordx2, no sc bits: served by this CU's L1 and this XCD's
// L2).  Inline asm because a C++ `volatile` load is emitted with sc0 sc1 on gfx950 (it bypasses both caches - the first version of this
// litmus used volatile and therefore saw no staleness at all) and a non-volatile one in a loop is hoisted.
__device__ __forceinline__ unsigned long long plain64(const void* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t plain32(const void* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// the three words of a slot requested together (one wait), as memo_find_lane does
__device__ __forceinline__ void plain_slot(const void* s, unsigned long long& kb, unsigned long long& kw, uint32_t& tag) {
    asm volatile("global_load_dword %2, %3, off offset:16\n\tglobal_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(kb), "=&v"(kw), "=&v"(tag) : "v"(s) : "memory");
}

struct Ctl { uint32_t warm[128]; uint32_t done[128]; };   // per publisher block: the reader is warm / the publisher has finished

template <int FORM>
__global__ __launch_bounds__(64) void k_publish(Slot* slots, Ctl* ctl, uint32_t R, uint32_t epoch, uint32_t gap) {
    const uint32_t b = blockIdx.x;
    if (threadIdx.x != 0) return;
    for (uint32_t spin = 0; spin < (1u << 22); ++spin)   // wait for the reader's warm flag (bounded)
        if (__hip_atomic_load(&ctl->warm[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch + 1u) break;
    for (uint32_t r = 0; r < R; ++r) {
        Slot* s = slots + (size_t)b * R + r;
        const unsigned long long k = key_of(b, r, epoch);
        if (FORM == 1) {
            __hip_atomic_store(&s->black, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&s->white, ~k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            s->black = k;
            s->white = ~k;
            if (FORM == 0)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __hip_atomic_store(&s->tag, 0x80000000u | (epoch << 16) | (r & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t i = 0; i < gap; ++i) __builtin_amdgcn_s_sleep(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&ctl->done[b], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[rb * 4 + ...]: 0 stale, 1 late, 2 polls, 3 ticks
template <int READ>
__global__ __launch_bounds__(64) void k_read(Slot* slots, Ctl* ctl, uint32_t R, uint32_t epoch, uint32_t shift, uint32_t nb,
                                             unsigned long long* out) {
    const uint32_t rb = blockIdx.x, b = (rb + nb - shift) % nb;
    if (threadIdx.x != 0) return;
    unsigned long long sink = 0;
    for (uint32_t r = 0; r < R; ++r) {   // warm: this CU's L1 and this XCD's L2 hold the slots as they are BEFORE they are published
        const Slot* s = slots + (size_t)b * R + r;
        sink += plain64(&s->black) + plain64(&s->white) + plain32(&s->tag);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&ctl->warm[b], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long stale = 0, late = 0, polls = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t r = 0; r < R; ++r) {
        Slot* s = slots + (size_t)b * R + r;
        const uint32_t want = 0x80000000u | (epoch << 16) | (r & 0xffffu);
        bool seen = false;
        for (uint32_t spin = 0; spin < (1u << 20); ++spin) {
            ++polls;
            if (__hip_atomic_load(&s->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) { seen = true; break; }
        }
        if (!seen) { ++late; continue; }
        unsigned long long kb, kw;
        if (READ == 1) {
            kb = __hip_atomic_load(&s->black, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            kw = __hip_atomic_load(&s->white, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (READ == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            kb = plain64(&s->black);
            kw = plain64(&s->white);
        }
        const unsigned long long k = key_of(b, r, epoch);
        if (kb != k || kw != ~k) ++stale;
    }
    out[rb * 4 + 0] = stale;
    out[rb * 4 + 1] = late;
    out[rb * 4 + 2] = polls + (sink & 0ULL);
    out[rb * 4 + 3] = __builtin_readcyclecounter() - t0;
}

// single look: wait (bounded) for the publisher's `done`, idle `delay` more, then look at every slot ONCE with plain (READ 0) or
// agent-scope (READ 1) loads, tag and keys requested together.  out: 0 unseen tags, 1 tag seen but keys stale, 2 looked at
template <int READ>
__global__ __launch_bounds__(64) void k_look(Slot* slots, Ctl* ctl, uint32_t R, uint32_t epoch, uint32_t shift, uint32_t nb, uint32_t delay,
                                             unsigned long long* out) {
    const uint32_t rb = blockIdx.x, b = (rb + nb - shift) % nb;
    if (threadIdx.x != 0) return;
    unsigned long long sink = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const Slot* s = slots + (size_t)b * R + r;
        sink += plain64(&s->black) + plain64(&s->white) + plain32(&s->tag);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&ctl->warm[b], epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t spin = 0; spin < (1u << 24); ++spin)
        if (__hip_atomic_load(&ctl->done[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch + 1u) break;
    for (uint32_t i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(64);
    unsigned long long unseen = 0, stale = 0;
    for (uint32_t r = 0; r < R; ++r) {
        Slot* s = slots + (size_t)b * R + r;
        uint32_t tag;
        unsigned long long kb, kw;
        if (READ == 1) {
            tag = __hip_atomic_load(&s->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            kb = __hip_atomic_load(&s->black, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            kw = __hip_atomic_load(&s->white, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            plain_slot(s, kb, kw, tag);
        }
        const unsigned long long k = key_of(b, r, epoch);
        if (tag != (0x80000000u | (epoch << 16) | (r & 0xffffu))) ++unseen;
        else if (kb != k || kw != ~k) ++stale;
    }
    out[rb * 4 + 0] = unseen;
    out[rb * 4 + 1] = stale;
    out[rb * 4 + 2] = R + (sink & 0ULL);
    out[rb * 4 + 3] = 0;
}

// background load on a third stream: streams through a buffer so that the L2s and the fabric are busy (hand-offs fail under load
// that pass on an idle chip)
__global__ __launch_bounds__(256) void k_noise(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, int reps) {
    for (int rep = 0; rep < reps; ++rep)
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
            uint4 v = src[i];
            v.x += rep;
            dst[i] = v;
        }
}

template <int FORM, int READ>
static int run_poll(Slot* slots, Ctl* ctl, unsigned long long* out, hipStream_t sp, hipStream_t sr, uint32_t nb, uint32_t R, uint32_t epoch,
                    uint32_t shift, uint32_t gap, unsigned long long* res) {
    CK(hipMemsetAsync(slots, 0, (size_t)nb * R * sizeof(Slot), sp));
    CK(hipMemsetAsync(out, 0, nb * 32, sp));
    CK(hipStreamSynchronize(sp));
    hipLaunchKernelGGL((k_read<READ>), dim3(nb), dim3(64), 0, sr, slots, ctl, R, epoch, shift, nb, out);
    hipLaunchKernelGGL((k_publish<FORM>), dim3(nb), dim3(64), 0, sp, slots, ctl, R, epoch, gap);
    CK(hipStreamSynchronize(sp));
    CK(hipStreamSynchronize(sr));
    std::vector<unsigned long long> h(nb * 4);
    CK(hipMemcpy(h.data(), out, nb * 32, hipMemcpyDeviceToHost));
    res[0] = res[1] = res[2] = res[3] = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (int j = 0; j < 4; ++j) res[j] += h[i * 4 + j];
    return 0;
}

template <int FORM, int READ>
static int run_look(Slot* slots, Ctl* ctl, unsigned long long* out, hipStream_t sp, hipStream_t sr, uint32_t nb, uint32_t R, uint32_t epoch,
                    uint32_t shift, uint32_t delay, unsigned long long* res) {
    CK(hipMemsetAsync(slots, 0, (size_t)nb * R * sizeof(Slot), sp));
    CK(hipMemsetAsync(out, 0, nb * 32, sp));
    CK(hipStreamSynchronize(sp));
    hipLaunchKernelGGL((k_look<READ>), dim3(nb), dim3(64), 0, sr, slots, ctl, R, epoch, shift, nb, delay, out);
    hipLaunchKernelGGL((k_publish<FORM>), dim3(nb), dim3(64), 0, sp, slots, ctl, R, epoch, 0u);
    CK(hipStreamSynchronize(sp));
    CK(hipStreamSynchronize(sr));
    std::vector<unsigned long long> h(nb * 4);
    CK(hipMemcpy(h.data(), out, nb * 32, hipMemcpyDeviceToHost));
    res[0] = res[1] = res[2] = res[3] = 0;
    for (uint32_t i = 0; i < nb; ++i)
        for (int j = 0; j < 4; ++j) res[j] += h[i * 4 + j];
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t nb = 64, R = argc > 1 ? (uint32_t)atoi(argv[1]) : 512u;
    Slot* slots;
    Ctl* ctl;
    unsigned long long* out;
    uint4 *na, *nbuf;
    const size_t noise_n = (256u << 20) / 16;
    CK(hipMalloc(&slots, (size_t)nb * R * sizeof(Slot)));
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&out, nb * 32));
    CK(hipMalloc(&na, noise_n * 16));
    CK(hipMalloc(&nbuf, noise_n * 16));
    CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipMemset(na, 1, noise_n * 16));
    hipStream_t sp, sr, sn;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sr, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    const char* form_name[3] = {"plain keys, fence(release, wavefront), relaxed agent tag store (rounds 1-5)",
                                "relaxed agent-scope (sc1) 8-byte key stores, s_waitcnt vmcnt(0), relaxed agent tag store",
                                "plain keys, fence(release, agent), s_waitcnt vmcnt(0), relaxed agent tag store"};
    const char* read_name[3] = {"plain loads", "relaxed agent-scope (sc1) loads", "fence(acquire, agent) then plain loads"};
    uint32_t epoch = 0;
    printf("{\"what\": \"32-byte slot published by one lane of one kernel, read by one lane of another kernel running concurrently on another stream; "
           "%u single-wave blocks per kernel, %u slots per block, reader warmed on the empty slots\",\n \"polling_reader\": [\n", nb, R);
    bool first = true;
    for (int load = 0; load < 2; ++load)
        for (uint32_t shift = 0; shift < 2; ++shift)
            for (int form = 0; form < 3; ++form)
                for (int read = 0; read < 3; ++read) {
                    unsigned long long res[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
                    const int reps = 4;
                    for (int rep = 0; rep < reps; ++rep) {
                        if (load) hipLaunchKernelGGL(k_noise, dim3(1024), dim3(256), 0, sn, na, nbuf, noise_n, 6);
                        int rc = 0;
                        const uint32_t gap = rep & 1 ? 4u : 0u;
#define RUN(F, Rd) rc = run_poll<F, Rd>(slots, ctl, out, sp, sr, nb, R, epoch, shift, gap, res)
                        switch (form * 3 + read) {
                            case 0: RUN(0, 0); break; case 1: RUN(0, 1); break; case 2: RUN(0, 2); break;
                            case 3: RUN(1, 0); break; case 4: RUN(1, 1); break; case 5: RUN(1, 2); break;
                            case 6: RUN(2, 0); break; case 7: RUN(2, 1); break; default: RUN(2, 2); break;
                        }
#undef RUN
                        if (rc) return rc;
                        ++epoch;
                        if (load) CK(hipStreamSynchronize(sn));
                        for (int j = 0; j < 4; ++j) tot[j] += res[j];
                    }
                    const double n = (double)nb * R * reps;
                    printf("%s  {\"chip\": \"%s\", \"pairing\": \"%s\", \"publish\": \"%s\", \"read\": \"%s\", \"slots\": %.0f, \"stale_keys\": %llu, "
                           "\"stale_fraction\": %.6f, \"tag_never_seen\": %llu, \"polls_per_slot\": %.2f, \"reader_ticks_per_slot\": %.1f}",
                           first ? "" : ",\n", load ? "streaming load on a third stream" : "idle", shift ? "cross-XCD (reader block b+1)" : "same-XCD (reader block b)",
                           form_name[form], read_name[read], n, tot[0], tot[0] / n, tot[1], tot[2] / n, tot[3] / n);
                    first = false;
                }
    printf("\n ],\n \"single_look_after_the_publisher_finished\": [\n");
    first = true;
    for (uint32_t shift = 0; shift < 2; ++shift)
        for (int form = 0; form < 3; ++form)
            for (int read = 0; read < 2; ++read)
                for (uint32_t delay = 0; delay <= 64; delay += 64) {
                    unsigned long long res[4], tot[4] = {0, 0, 0, 0};
                    for (int rep = 0; rep < 4; ++rep) {
                        int rc = 0;
#define RUN(F, Rd) rc = run_look<F, Rd>(slots, ctl, out, sp, sr, nb, R, epoch, shift, delay, res)
                        switch (form * 2 + read) {
                            case 0: RUN(0, 0); break; case 1: RUN(0, 1); break; case 2: RUN(1, 0); break;
                            case 3: RUN(1, 1); break; case 4: RUN(2, 0); break; default: RUN(2, 1); break;
                        }
#undef RUN
                        if (rc) return rc;
                        ++epoch;
                        for (int j = 0; j < 4; ++j) tot[j] += res[j];
                    }
                    printf("%s  {\"pairing\": \"%s\", \"publish\": \"%s\", \"read\": \"%s\", \"idle_s_sleep64_before_the_look\": %u, \"slots\": %llu, "
                           "\"tag_unseen\": %llu, \"unseen_fraction\": %.6f, \"tag_seen_keys_stale\": %llu}",
                           first ? "" : ",\n", shift ? "cross-XCD" : "same-XCD", form_name[form], read_name[read], delay, tot[2], tot[0],
                           tot[2] ? (double)tot[0] / tot[2] : 0.0, tot[1]);
                    first = false;
                }
    printf("\n ]\n}\n");
    return 0;
}
