// raz_net.hip — forward pass of the policy/value residual CNN (reference: agent/model.py:28-72,
// evaluated through agent/api.py:30-45) for a batch of leaf positions given as bitboards.
//
// Numerics contract "raznet-forward-v1" (DESIGN.md §5): every output is ONE k-ordered fmaf chain
//   conv3x3: acc = b[oc]; for tap=ky*3+kx: for ic: acc = fmaf(x[ic][nbr(sq,tap)] or 0, w, acc)
//   block:   relu(conv1) ; relu(conv2 + skip)        heads: 1x1 conv chains over ic, dense chains over j
//   softmax: max, det_expf(l - max), xor-butterfly sum (1,2,4,8,16,32), divide;  value: det_tanhf
// which is what v_mfma_f32_* accumulates bit for bit, so this wave-per-position kernel, the MFMA
// implicit-GEMM kernel for wide nets and the CPU oracle all agree to the last bit.
//
// Kernel `k_net_wave`: one wavefront per position, lane = board square.  Activations [F][64] live
// in LDS (three buffers) when they fit, else in a caller-provided HBM scratch; 16 output channels
// are accumulated at a time in registers; the 3x3 taps read the neighbouring lanes' activations
// from LDS with an off-board predicate; weights are wave-uniform and pre-arranged as
// [layer][oc/16][tap][ic][16] so one scalar s_load_dwordx16 feeds 16 fmafs.  Layer 0 reads its
// input planes straight out of the two bitboards.  This is the right shape for narrow nets
// (mini.yml: F=16, R=1) where a GEMM tile would be mostly padding; wide nets (F=256) go to the
// MFMA kernel.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_internal.h"
#include "raz_net_layout.h"

bool raz_net_mfma_supported(int F, int V);
size_t raz_net_wide_scratch_bytes(int F, size_t n);
int raz_net_forward_wide(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                         const uint8_t* active, float* policy, float* value, size_t n, void* scratch,
                         size_t scratch_bytes, hipStream_t s);
void raz_net_build_f16x3(const float* src, float* dst, int F, int R, int V);
size_t raz_net_f16x3_scratch_bytes(int F, size_t n);
unsigned* raz_net_f16x3_flag(const float* W, int F, int R, int V);
int raz_net_forward_f16x3(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                          const uint8_t* active, float* policy, float* value, size_t n, void* scratch, size_t scratch_bytes,
                          hipStream_t s, const uint32_t* list, const uint32_t* n_ptr);
int raz_net_forward_mfma(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                         const uint8_t* active, float* policy, float* value, size_t n, hipStream_t s,
                         unsigned long long* prof);

namespace {

constexpr int32_t kMagic = 0x4E5A4152;

struct NetDims {
    int F, R, V;
};

// One position on one wave: bo / be = the position, policy_row / value_out = where its answer goes, slot = its activation
// buffers in `scratch` when they do not live in LDS.  TWO_BUF: two activation buffers instead of three - the second conv of a
// block writes relu(conv + skip) IN PLACE over its skip input (element (channel, square) reads the skip value it overwrites and
// nothing else of that buffer) - so that a 256-filter position fits the LDS of a CU (128 KB + heads); same chains, same bits.
template <bool LDS_ACT, bool TWO_BUF = false>
__device__ __forceinline__ void net_wave_position(const float* __restrict__ W, NetDims d, raz_bb bo, raz_bb be,
                                                  float* __restrict__ policy_row, float* __restrict__ value_out,
                                                  float* __restrict__ scratch, size_t slot, float* smem) {
    const int lane = threadIdx.x;
    const int F = d.F, R = d.R, V = d.V;
    float* buf0;
    float* buf1;
    float* buf2;
    float* head;  // ph[128] vh[64] h1[V]
    if (LDS_ACT) {
        buf0 = smem;
        buf1 = smem + F * 64;
        buf2 = TWO_BUF ? buf0 : smem + 2 * F * 64;
        head = smem + (TWO_BUF ? 2 : 3) * F * 64;
    } else {
        float* base = scratch + slot * 3 * F * 64;
        buf0 = base;
        buf1 = base + F * 64;
        buf2 = base + 2 * F * 64;
        head = smem;
    }
    // neighbour index and validity for the 9 taps
    const int y = lane >> 3, x = lane & 7;
    int nbr[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        ok[t] = (yy >= 0) && (yy < 8) && (xx >= 0) && (xx < 8);
        nbr[t] = ok[t] ? yy * 8 + xx : lane;
    }
    const int nlayers = 2 * R + 1;
    float* bufs[3] = {buf0, buf1, buf2};
    int ia = 0;  // which buffer holds the block input `a` (the stem writes bufs[0])
    for (int l = 0; l < nlayers; ++l) {
        const int cin = l == 0 ? 2 : F;
        const float* w = W + conv_off(F, l);
        const float* bias = w + (size_t)F * 9 * cin;
        // stem: planes -> a.  Block: conv1 a -> t, conv2 t (+ skip a) -> u, then a := u.
        const bool second = (l > 0) && ((l & 1) == 0);
        const float* in = l == 0 ? nullptr : (second ? bufs[(ia + 1) % 3] : bufs[ia]);
        const float* skip = second ? bufs[ia] : nullptr;
        float* out = l == 0 ? bufs[0] : (second ? bufs[TWO_BUF ? ia : (ia + 2) % 3] : bufs[(ia + 1) % 3]);
        for (int ocb = 0; ocb < F / 16; ++ocb) {
            float acc[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = bias[ocb * 16 + o];
            const float* wb = w + (size_t)ocb * 9 * cin * 16;
            // k order of raznet-forward-v1: 16-channel chunks, then tap, then channel within the chunk
            for (int c0 = 0; c0 < cin; c0 += 16) {
#pragma unroll 1
                for (int t = 0; t < 9; ++t) {
                    const float* wt = wb + (size_t)t * cin * 16;
                    if (l == 0) {
                        const float x0 = ok[t] ? (float)((bo >> nbr[t]) & 1) : 0.0f;
                        const float x1 = ok[t] ? (float)((be >> nbr[t]) & 1) : 0.0f;
#pragma unroll
                        for (int o = 0; o < 16; ++o) acc[o] = fmaf(x0, wt[o], acc[o]);
#pragma unroll
                        for (int o = 0; o < 16; ++o) acc[o] = fmaf(x1, wt[16 + o], acc[o]);
                    } else {
#pragma unroll 4
                        for (int ic = c0; ic < c0 + 16; ++ic) {
                            const float xv = ok[t] ? in[ic * 64 + nbr[t]] : 0.0f;
                            const float* w16 = wt + ic * 16;
#pragma unroll
                            for (int o = 0; o < 16; ++o) acc[o] = fmaf(xv, w16[o], acc[o]);
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                float v = acc[o];
                if (second) v = v + skip[(ocb * 16 + o) * 64 + lane];
                out[(ocb * 16 + o) * 64 + lane] = v > 0.0f ? v : 0.0f;
            }
        }
        if (!LDS_ACT) __threadfence_block();
        __syncthreads();  // single-wave block: orders this layer's writes before neighbour reads
        if (second && !TWO_BUF) ia = (ia + 2) % 3;
    }
    const float* a = bufs[ia];  // trunk output [F][64]
    const float* H = W + heads_off(F, R);
    const float* pol_w = H;
    const float* pol_b = pol_w + 2 * F;
    const float* pfc_w = pol_b + 2;
    const float* pfc_b = pfc_w + 128 * 64;
    const float* val_w = pfc_b + 64;
    const float* val_b = val_w + F;
    const float* v1_w = val_b + 1;
    const float* v1_b = v1_w + 64 * V;
    const float* v2_w = v1_b + V;
    const float* v2_b = v2_w + V;
    float* ph = head;
    float* vh = head + 128;
    float* h1 = head + 192;
    {
        float p0 = pol_b[0], p1 = pol_b[1], v0 = val_b[0];
        for (int ic = 0; ic < F; ++ic) {
            const float xv = a[ic * 64 + lane];
            p0 = fmaf(xv, pol_w[ic], p0);
            p1 = fmaf(xv, pol_w[F + ic], p1);
            v0 = fmaf(xv, val_w[ic], v0);
        }
        ph[lane] = p0 > 0.0f ? p0 : 0.0f;
        ph[64 + lane] = p1 > 0.0f ? p1 : 0.0f;
        vh[lane] = v0 > 0.0f ? v0 : 0.0f;
    }
    __syncthreads();
    // policy dense 128 -> 64, lane = output square
    float logit = pfc_b[lane];
#pragma unroll 8
    for (int j = 0; j < 128; ++j) logit = fmaf(ph[j], pfc_w[j * 64 + lane], logit);
    float m = logit;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    const float e = raz_det_expf(logit - m);
    float sum = e;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) sum = sum + __shfl_xor(sum, s);
    policy_row[lane] = e / sum;
    // value dense 64 -> V (relu), lane = hidden unit (loop if V > 64)
    for (int o0 = 0; o0 < V; o0 += 64) {
        const int o = o0 + lane;
        if (o < V) {
            float acc = v1_b[o];
#pragma unroll 8
            for (int j = 0; j < 64; ++j) acc = fmaf(vh[j], v1_w[j * V + o], acc);
            h1[o] = acc > 0.0f ? acc : 0.0f;
        }
    }
    __syncthreads();
    float acc = v2_b[0];
    for (int j = 0; j < V; ++j) acc = fmaf(h1[j], v2_w[j], acc);
    if (lane == 0) *value_out = raz_det_tanhf(acc);
}

template <bool LDS_ACT>
__global__ __launch_bounds__(64) void k_net_wave(const float* __restrict__ W, NetDims d,
                                                 const raz_bb* __restrict__ own,
                                                 const raz_bb* __restrict__ enemy,
                                                 const uint8_t* __restrict__ active,
                                                 float* __restrict__ policy, float* __restrict__ value,
                                                 float* __restrict__ scratch, int n) {
    alignas(16) static float smem[RAZ_EMU_LDS_FLOATS];
    const int pos = blockIdx.x;
    if (pos >= n) return;
    if (active && !active[pos]) return;
    net_wave_position<LDS_ACT>(W, d, own[pos], enemy[pos], policy + (size_t)pos * 64, value + pos, scratch, (size_t)pos, smem);
}

// The rows of a split-f16 forward whose activations left the f16 range, once more on the exact-f32 chains (raz_internal.h
// raz_net_repair_rows).  A FIXED grid (<= 256 blocks: each block's 130 KB of LDS holds a CU): block b looks at its contiguous
// share of the row flags 64 at a time and repairs the rows whose flag is up - none, normally.  The row's activations live in
// LDS (two buffers, in-place second conv): no scratch, no limit on the rows.
__global__ __launch_bounds__(64) void k_net_wave_repair(const float* __restrict__ W, NetDims d, const raz_bb* __restrict__ own,
                                                        const raz_bb* __restrict__ enemy, float* __restrict__ policy,
                                                        float* __restrict__ value, int n, unsigned* __restrict__ rowflag,
                                                        unsigned* __restrict__ sticky, const uint32_t* __restrict__ list,
                                                        const uint32_t* __restrict__ n_ptr) {
    alignas(16) static float smem[RAZ_EMU_LDS_FLOATS];
    const int rows = n_ptr ? ((int)*n_ptr < n ? (int)*n_ptr : n) : n;
    const int per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    for (int base = lo; base < hi; base += 64) {
        const int mine = base + (int)threadIdx.x;
        unsigned long long m = __ballot(mine < hi && rowflag[(size_t)mine * RAZ_NET_ROWFLAG_WORDS] != 0u);
        for (; m; m &= m - 1) {
            const int i = base + __ffsll((long long)m) - 1;
            if (threadIdx.x == 0) {
                atomicAdd(sticky + 1, 1u);   // rows repaired since the net was loaded
                // a forward with many rows out of range belongs on the exact-f32 matrix-core kernels: tell the caller (raz_net_range_check)
                if (atomicAdd(rowflag + 1, 1u) >= RAZ_NET_REPAIR_ROWS) atomicOr(sticky, 1u);
            }
            const size_t row = list ? list[i] : (size_t)i;
            net_wave_position<true, true>(W, d, own[row], enemy[row], policy + row * 64, value + row, nullptr, 0, smem);
        }
    }
}

}  // namespace

extern "C" size_t raz_net_weight_bytes(int filters, int res_layers, int value_fc) {
    if (filters <= 0 || filters % 16 || res_layers < 0 || value_fc <= 0) return 0;
    return total_floats(filters, res_layers, value_fc) * sizeof(float);
}

static size_t lds_bytes_for(int F, int V, bool lds_act) {
    return ((lds_act ? (size_t)3 * F * 64 : 0) + 192 + (size_t)V) * sizeof(float);
}
static bool use_lds(int F, int V) { return lds_bytes_for(F, V, true) <= 64 * 1024; }

extern "C" size_t raz_net_scratch_bytes(int filters, int value_fc, size_t n) {
    if (raz_net_mfma_supported(filters, value_fc) || use_lds(filters, value_fc)) return 0;
    if (wide_supported(filters)) {  // the larger of the two paths (reserved==1 forces the VALU kernel)
        const size_t a = raz_net_wide_scratch_bytes(filters, n), b = n * 3 * (size_t)filters * 64 * sizeof(float);
        const size_t c = f16x3_supported(filters) ? raz_net_f16x3_scratch_bytes(filters, n) : 0;   // (its repair rows make it the largest for small n)
        return a > b ? (a > c ? a : c) : (b > c ? b : c);
    }
    return n * 3 * (size_t)filters * 64 * sizeof(float);
}

// Re-arrange the canonical blob (agent/model.py to_blob; see include/raz.h) into the device layout
// and copy it to caller-owned device memory.
extern "C" int raz_net_load(raz_net* net, const void* blob, size_t blob_bytes, void* d_weights,
                            size_t d_bytes, raz_stream_t stream) {
    if (!net || !blob || !d_weights) return raz_fail(RAZ_EINVAL, "raz_net_load: NULL argument");
    if (blob_bytes < 32) return raz_fail(RAZ_EINVAL, "raz_net_load: blob too small");
    int32_t h[8];
    memcpy(h, blob, 32);
    if (h[0] != kMagic || h[1] != 1 || h[5] != 3)
        return raz_fail(RAZ_EINVAL, "raz_net_load: not a raznet v1 blob (3x3 filters)");
    const int F = h[2], R = h[3], V = h[4];
    if (F <= 0 || F % 16 || R < 0 || V <= 0)
        return raz_fail(RAZ_EINVAL, "raz_net_load: filters must be a positive multiple of 16");
    const size_t nsrc = ((size_t)F * 18 + F) + (size_t)R * 2 * ((size_t)F * F * 9 + F) + (2 * (size_t)F + 2) +
                        (128 * 64 + 64) + ((size_t)F + 1) + (64 * (size_t)V + V) + ((size_t)V + 1);
    if (blob_bytes != 32 + 4 * nsrc) return raz_fail(RAZ_EINVAL, "raz_net_load: blob size mismatch");
    const size_t need = total_floats(F, R, V) * sizeof(float);
    if (d_bytes < need) return raz_fail(RAZ_ENOMEM, "raz_net_load: device weight buffer too small");
    const float* src = (const float*)((const char*)blob + 32);
    std::vector<float> dst(total_floats(F, R, V));
    for (int l = 0; l < 2 * R + 1; ++l) {
        const int cin = l == 0 ? 2 : F;
        float* w = dst.data() + conv_off(F, l);
        for (int oc = 0; oc < F; ++oc)
            for (int ic = 0; ic < cin; ++ic)
                for (int t = 0; t < 9; ++t)
                    w[(((size_t)(oc / 16) * 9 + t) * cin + ic) * 16 + (oc % 16)] = src[((size_t)oc * cin + ic) * 9 + t];
        memcpy(w + (size_t)F * 9 * cin, src + (size_t)F * cin * 9, F * sizeof(float));
        src += (size_t)F * cin * 9 + F;
    }
    const size_t nheads = wave_floats(F, R, V) - heads_off(F, R);
    memcpy(dst.data() + heads_off(F, R), src, nheads * sizeof(float));
    {   // region 2: B operands of v_mfma_f32_16x16x4_f32, k = tap*Cin + ic (raz_net_layout.h)
        const float* lsrc = (const float*)((const char*)blob + 32);
        for (int l = 0; l < 2 * R + 1; ++l) {
            const int cin = l == 0 ? 2 : F, ks = mfma_ksteps(F, l);
            float* w = dst.data() + mfma_layer_off(F, R, V, l);
            for (int nt = 0; nt < F / 16; ++nt)
                for (int st = 0; st < ks; ++st)
                    for (int ln = 0; ln < 64; ++ln) {
                        const int k = 4 * st + (ln >> 4), oc = nt * 16 + (ln & 15);
                        float v = 0.0f;
                        if (k < 9 * cin) {
                            int t, ic;
                            if (cin <= 16) { t = k / cin; ic = k % cin; }
                            else { const int c = k / 144, r = k % 144; t = r / 16; ic = c * 16 + r % 16; }
                            v = lsrc[((size_t)oc * cin + ic) * 9 + t];
                        }
                        w[((size_t)nt * ks + st) * 64 + ln] = v;
                    }
            lsrc += (size_t)F * cin * 9 + F;
        }
    }
    if (wide_supported(F)) {  // region 3: A operands of v_mfma_f32_32x32x2_f32 per (layer, chunk, 64-channel tile)
        const float* lsrc = (const float*)((const char*)blob + 32) + ((size_t)F * 18 + F);
        for (int l = 1; l < 2 * R + 1; ++l) {
            for (int c = 0; c < F / 16; ++c)
                for (int nt = 0; nt < F / 64; ++nt) {
                    float* w = dst.data() + wide_tile_off(F, R, V, l, c, nt);
                    for (int st = 0; st < 72; ++st)
                        for (int mt = 0; mt < 2; ++mt)
                            for (int ln = 0; ln < 64; ++ln) {
                                const int kk = 2 * st + (ln >> 5), t = kk / 16, ic = c * 16 + kk % 16;
                                const int oc = nt * 64 + mt * 32 + (ln & 31);
                                w[(st * 2 + mt) * 64 + ln] = lsrc[((size_t)oc * F + ic) * 9 + t];
                            }
                }
            lsrc += (size_t)F * F * 9 + F;
        }
    }
    if (f16x3_supported(F)) raz_net_build_f16x3((const float*)((const char*)blob + 32), dst.data(), F, R, V);  // region 4 (+ a cleared range flag)
    RAZ_HIP_TRY(hipMemcpyAsync(d_weights, dst.data(), need, hipMemcpyHostToDevice, (hipStream_t)stream),
                "raz_net_load: hipMemcpyAsync");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_net_load: sync");  // dst is a local
    net->filters = F;
    net->res_layers = R;
    net->value_fc = V;
    net->d_weights = d_weights;
    net->weight_bytes = need;
    return RAZ_OK;
}

// raznet-forward-v2 carries activations as pairs of halfs: an activation beyond the f16 range (>= 60000; none in any
// net we have seen: BatchNorm keeps them O(1)) would become inf.  The kernels flag the ROW instead of failing silently, and
// the forward evaluates flagged rows again on the exact-f32 chains (k_net_wave_repair: a row's answer stays a function of its
// position alone).  Only when more than RAZ_NET_REPAIR_ROWS rows of ONE forward are out of range is the sticky flag in the weight
// image raised: *overflowed = 1 means outputs since the net was loaded cannot be trusted and the net must be run with the
// exact-f32 kernels (raz_net.reserved = 0).  Synchronises `stream`.
extern "C" int raz_net_range_check(const raz_net* net, int* overflowed, raz_stream_t stream) {
    if (!net || !net->d_weights || !overflowed) return raz_fail(RAZ_EINVAL, "raz_net_range_check: NULL argument");
    *overflowed = 0;
    if (!f16x3_supported(net->filters)) return RAZ_OK;
    unsigned v = 0;
    RAZ_HIP_TRY(hipMemcpyAsync(&v, raz_net_f16x3_flag((const float*)net->d_weights, net->filters, net->res_layers, net->value_fc), 4,
                               hipMemcpyDeviceToHost, (hipStream_t)stream), "raz_net_range_check: copy");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_net_range_check: sync");
    *overflowed = v != 0;
    return RAZ_OK;
}

extern "C" int raz_net_range_stats(const raz_net* net, int* overflowed, unsigned long long* rows_repaired, raz_stream_t stream) {
    if (!net || !net->d_weights || !overflowed || !rows_repaired) return raz_fail(RAZ_EINVAL, "raz_net_range_stats: NULL argument");
    *overflowed = 0;
    *rows_repaired = 0;
    if (!f16x3_supported(net->filters)) return RAZ_OK;
    unsigned v[2] = {0, 0};
    RAZ_HIP_TRY(hipMemcpyAsync(v, raz_net_f16x3_flag((const float*)net->d_weights, net->filters, net->res_layers, net->value_fc), 8,
                               hipMemcpyDeviceToHost, (hipStream_t)stream), "raz_net_range_stats: copy");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_net_range_stats: sync");
    *overflowed = v[0] != 0;
    *rows_repaired = v[1];
    return RAZ_OK;
}

int raz_net_repair_rows(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy, float* policy, float* value,
                        size_t n, unsigned* rowflag, unsigned* sticky, const uint32_t* list, const uint32_t* n_ptr, hipStream_t s) {
    NetDims d = {F, R, V};
    const size_t shm = ((size_t)2 * F * 64 + 192 + (size_t)V) * sizeof(float);   // two activation buffers + the heads' scratch
    {   // 130 KB for F = 256: above the default dynamic limit - raise it once per device (two threads racing here both set it: harmless)
        static std::atomic<unsigned long long> attr_devices{0};
        int dev = 0;
        RAZ_HIP_TRY(hipGetDevice(&dev), "raz_net_forward: hipGetDevice");
        if (dev >= 64 || !(attr_devices.load(std::memory_order_acquire) >> dev & 1)) {
            RAZ_HIP_TRY(hipFuncSetAttribute((const void*)k_net_wave_repair, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "raz_net_forward: hipFuncSetAttribute (repair)");
            if (dev < 64) attr_devices.fetch_or(1ull << dev, std::memory_order_release);
        }
    }
    const unsigned grid = n < 256 ? (unsigned)n : 256u;   // one block per CU reads ceil(n / 256) row flags; a flagged row is repaired by the block that found it
    hipLaunchKernelGGL(k_net_wave_repair, dim3(grid), dim3(64), shm, s, W, d, (const raz_bb*)own, (const raz_bb*)enemy, policy, value,
                       (int)n, rowflag, sticky, list, n_ptr);
    return raz_check_launch("raz_net_forward (range repair)");
}

// Engine-internal: raz_net_forward over a compacted batch (raz_leaf_cache.hip).  Only the f16x3 path has the indexed form;
// other nets run the ordinary forward over the rows whose `active` flag the cache left set.
int raz_net_forward_compact(const raz_net* net, const uint64_t* own, const uint64_t* enemy, const uint8_t* active, float* policy,
                            float* value, size_t n, void* scratch, size_t scratch_bytes, hipStream_t stream, const uint32_t* list,
                            const uint32_t* n_ptr) {
    if (n && net && f16x3_supported(net->filters) && net->reserved == 4)
        return raz_net_forward_f16x3((const float*)net->d_weights, net->filters, net->res_layers, net->value_fc, own, enemy, active,
                                     policy, value, n, scratch, scratch_bytes, stream, list, n_ptr);
    return raz_net_forward(net, own, enemy, active, policy, value, n, scratch, scratch_bytes, (raz_stream_t)stream);
}

extern "C" int raz_net_forward(const raz_net* net, const uint64_t* own, const uint64_t* enemy,
                               const uint8_t* active, float* policy, float* value, size_t n,
                               void* scratch, size_t scratch_bytes, raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    if (!net || !net->d_weights || !own || !enemy || !policy || !value)
        return raz_fail(RAZ_EINVAL, "raz_net_forward: NULL argument");
    const int F = net->filters, V = net->value_fc;
    if (net->reserved != 0 && net->reserved != 1 && net->reserved != 2 && net->reserved != 4)
        return raz_fail(RAZ_EINVAL, "raz_net_forward: raz_net.reserved must be 0, 1, 2 or 4 (5 and 6 selected kernels that were removed in ABI 3)");
    // reserved (tests): 1 forces the VALU kernel, 2 the one-wave-per-position MFMA kernel
    if (raz_net_mfma_supported(F, V) && net->reserved != 1)
    {
        // debug: RAZ_NET_PROF=1 and a caller scratch of >= n*64 bytes -> per-position phase ticks
        unsigned long long* prof = nullptr;
        if (scratch && scratch_bytes >= n * 64 && getenv("RAZ_NET_PROF")) prof = (unsigned long long*)scratch;
        return raz_net_forward_mfma((const float*)net->d_weights, F, net->res_layers, V, own, enemy, active, policy,
                                    value, n, (hipStream_t)stream, prof);
    }
    // reserved 4: raznet-forward-v2 - the trunk on the f16 matrix cores with split operands (raz_net_f16x3.hip), within 1e-5
    // of the fp32 graph but not bit-identical to the exact-f32 kernels (0 / 5: raznet-forward-v1)
    if (f16x3_supported(F) && net->reserved == 4)
        return raz_net_forward_f16x3((const float*)net->d_weights, F, net->res_layers, V, own, enemy, active, policy, value, n,
                                     scratch, scratch_bytes, (hipStream_t)stream, nullptr, nullptr);
    if (net->reserved == 4) return raz_fail(RAZ_EINVAL, "raz_net_forward: the f16x3 kernel needs filters % 128 == 0");
    if (wide_supported(F) && net->reserved != 1)
        return raz_net_forward_wide((const float*)net->d_weights, F, net->res_layers, V, own, enemy, active, policy,
                                    value, n, scratch, scratch_bytes, (hipStream_t)stream);
    NetDims d = {F, net->res_layers, V};
    const bool lds = use_lds(F, V);
    if (!lds) {
        if (!scratch || scratch_bytes < raz_net_scratch_bytes(F, V, n))
            return raz_fail(RAZ_ENOMEM, "raz_net_forward: scratch too small (raz_net_scratch_bytes)");
    }
    const size_t shm = lds_bytes_for(F, V, lds);
    if (lds)
        hipLaunchKernelGGL(k_net_wave<true>, dim3((unsigned)n), dim3(64), shm, (hipStream_t)stream,
                           (const float*)net->d_weights, d, (const raz_bb*)own, (const raz_bb*)enemy, active,
                           policy, value, (float*)scratch, (int)n);
    else
        hipLaunchKernelGGL(k_net_wave<false>, dim3((unsigned)n), dim3(64), shm, (hipStream_t)stream,
                           (const float*)net->d_weights, d, (const raz_bb*)own, (const raz_bb*)enemy, active,
                           policy, value, (float*)scratch, (int)n);
    return raz_check_launch("raz_net_forward");
}
