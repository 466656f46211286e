// common.h — shared host/device helpers of libgnnmp (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>
#include "gnnmp.h"

namespace gnnmp {

// ---- error plumbing (thread-local message, negative status codes; nothing throws) -------------
int fail(int status, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define GNNMP_HIP(expr)                                         \
    do {                                                        \
        hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return ::gnnmp::hip_fail(e__, #expr); \
    } while (0)

#define GNNMP_LAUNCH_CHECK(what)                                \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return ::gnnmp::hip_fail(e__, what); \
    } while (0)

// ---- per-device state ------------------------------------------------------------------------------
// One PROCESS may drive several devices (a Julia host calling AMDGPU.device!(d) between calls; torchrun's one
// process per GPU hides this).  Everything the library remembers about "the device" is therefore keyed on the CURRENT device of the
// calling thread: kernel attributes (hipFuncSetAttribute acts on the current device's function object only), the compute-unit count,
// the block pool (pool.hip), graph prep's scratch cache (graphprep.hip).
constexpr int GNNMP_MAX_DEVICES = 32;
// hipGetDevice clamped to [0, GNNMP_MAX_DEVICES); gnnmp_debug_mock_device (tests) overrides it for the calling thread
int current_device();

// run f() once per device (thread-safe); every later call on that device returns what the first one returned
struct DeviceOnce {
    std::mutex m;
    unsigned char done[GNNMP_MAX_DEVICES] = {};
    hipError_t err[GNNMP_MAX_DEVICES] = {};
};
template <class F>
inline hipError_t device_once(DeviceOnce &o, F &&f) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(o.m);     // (uncontended: ~20 ns against a kernel launch's microseconds)
    if (!o.done[dev]) {
        o.err[dev] = f();
        o.done[dev] = 1;
    }
    return o.err[dev];
}

// The opt-in of a kernel to more than 64 KB of dynamic LDS: once per DEVICE and kernel instance, thread-safe (the library promises no
// unsynchronised mutable state besides the knobs, gnnmp.h).  GNNMP_LDS_OPTIN("name", &kernel<args...>)
#define GNNMP_LDS_OPTIN(what, ...)                                                                                            \
    do {                                                                                                                      \
        static ::gnnmp::DeviceOnce once__;                                                                                    \
        const hipError_t err__ = ::gnnmp::device_once(once__, [] {                                                            \
            return hipFuncSetAttribute(reinterpret_cast<const void *>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       160 * 1024);                                                                           \
        });                                                                                                                   \
        if (err__ != hipSuccess) return ::gnnmp::hip_fail(err__, "hipFuncSetAttribute(" what ")");                            \
    } while (0)

// ---- tuning knobs (perf experiments; not part of the drop-in surface) -------------------------
enum Knob {
    KNOB_FORCE_VEC = 0,    // 0 = auto, else 1|2|4
    KNOB_FORCE_LOG2G = 1,  // -1 = auto, else 0..6
    KNOB_UNROLL = 2,       // 0 = auto (8: measured best on both bench shapes), else 2|4|8
    KNOB_XCD_REMAP = 3,    // 0 = off, 1 = auto (default: only when the gathered matrix fits the Infinity Cache), 2 = on
    KNOB_LONG_ROW = 4,     // long-row threshold (default GNNMP_LONG_ROW)
    KNOB_BLOCK_WAVES = 5,  // waves per block in the row kernels: 0 = auto (propagate 4, GAT 1), else 1..4
    KNOB_DENSE_GENERIC = 6,  // 0 = auto (dense_t16 on its shapes, else W-resident 32x32x2, else K-chunked), 1 = force K-chunked,
                             // 2 = skip dense_t16 (round-1 kernels only)
    KNOB_DENSE_PREFETCH = 7,  // W-resident dense kernel scheduling: bit 4 = per-SIMD matrix-pipe token, low 4 bits =
                              // start skew of waves 4-7 in s_sleep(127) units.  Default 17 (token + 1): 0.83 -> 0.72 ms
                              // at 2.4M x 100 => 100.  Bit 5 = turn the cross-tile register prefetch OFF (on by default; its
                              // first version spilled — 270 VGPRs — and was slower: see dense.hip).
    KNOB_GAT_FAST_EXP = 8,   // retired: the one-pass attention kernel always uses v_exp_f32 now (gat_fused.hip, gexp)
    KNOB_GRADW_SLABS = 9,    // ΔW kernel: slabs per CU (0 = auto)
    KNOB_GRADW_RP = 10,      // ΔW kernel: 0 = the 16x16x4 kernel, < 0 = the round-1 32x32x2 kernel (A/B runs)
    KNOB_GRADW_MIN_ROWS = 11,  // ΔW kernel: rows-per-slab floor (0 = auto: ~3 slabs per CU on small inputs, 512 on large)
    KNOB_DENSE_T16_WAVES = 12, // dense_t16_kernel: waves per block (0 = auto, else 1..16)
    KNOB_T16_DEBUG = 13,       // dense_t16_kernel phase ablation (experiments only): 1 = no stores, 2 = no x loads
    KNOB_FUSED_WAVES = 14,     // fused_conv_kernel: 0 = auto (as many waves as LDS holds tiles for, <= 16), > 0 = cap, < 0 = never fuse
    KNOB_ROW_ORDER = 15,       // rows by decreasing length in the row kernels that share a wave between rows: 0 (default) = never,
                               // 1 = when the gathered matrix exceeds the Infinity Cache, 2 = always (use_row_order)
    KNOB_SOFTMAX_ROWS = 16,    // one-pass narrow-row softmax (softmax_rows.hip): 0 = auto, < 0 = the three-step kernels on every row
    KNOB_DENSE_SPLIT = 17,     // split-bf16 dense core (msplit.h, dense_split.hip): 0 = auto (on for its shapes), < 0 = the fp32-MFMA
                               // kernels of rounds 1-2 on every shape
    KNOB_CHAIN = 18,           // fused GraphConv chain kernel (graph_chain.hip): 0 = auto, < 0 = never (layer-by-layer path)
    KNOB_VARIANT = 19,         // A/B switches of round 3 (all variants are correct code): low 2 bits = 1: the wave-pair chain kernel with 8 waves
                               // a block (default 12); 16: dense_split runs its column tiles one after the other; 32: dense_split stores
                               // straight from the accumulator layout (default: through the per-wave LDS stage); 64: never dense_wreg;
                               // 128: split rows folded by a second kernel (csr_combine / gat_fused_combine) as in rounds 1-4 instead of
                               // by the last chunk to arrive inside the row kernel (round 5, use_fold); 512: dense_wreg from 4 096 rows
                               // on (default 32 768: below that its eight-wave blocks are too few) — so that small tests reach it
    KNOB_COUNT = 20
};
int knob(int k);
int device_cus();   // compute units of the current device, queried once (hipDeviceGetAttribute costs microseconds per call)

// ---- device helpers ---------------------------------------------------------------------------
// index element load: idx_bytes in {4, 8}; returns 0-based int64
__device__ __forceinline__ int64_t load_index(const void *p, int64_t k, int idx_bytes, int base) {
    return idx_bytes == 8 ? (reinterpret_cast<const int64_t *>(p)[k] - base)
                          : (static_cast<int64_t>(reinterpret_cast<const int32_t *>(p)[k]) - base);
}
__device__ __forceinline__ void store_index(void *p, int64_t k, int idx_bytes, int64_t v) {
    if (idx_bytes == 8)
        reinterpret_cast<int64_t *>(p)[k] = v;
    else
        reinterpret_cast<int32_t *>(p)[k] = static_cast<int32_t>(v);
}

// Julia Base.max / Base.min on floats: NaN-propagating, max(-0.0, +0.0) = +0.0, min(+0.0,-0.0) = -0.0.
// (NNlib.scatter!(max, ...) applies Base.max element by element.)
// exp of a softmax exponent (x <= 0) on the hardware exponential: v_exp_f32 of x * log2(e).  The product's rounding puts a
// relative error of |x| * 6e-8 on exp(x), i.e. an ABSOLUTE error of at most 0.37 * 6e-8 = 2.2e-8 on a weight in (0, 1] —
// below fp32 epsilon of the weights that matter — for 2 instructions instead of libm's 12.  exp(-inf) = 0, NaN propagates.
// Used by the one-pass attention kernel only: in its pullbacks (one exp per edge, not VALU-bound) the same substitution
// measured 1.6 % SLOWER on the same box, and the reference-order softmax kernels keep expf for parity of the order.
// Attention dropout (GNNlib/src/layers/conv.jl:139: α = dropout(α, l.dropout)): the keep decision of coefficient (edge e, head h) is a
// pure function of (seed, e, h) — e = the edge's position in the caller's edge list, plan-added self loops at E + node like
// add_self_loops appends them — so the forward, both backward passes and any host restatement of these few lines agree
// without any stored mask.  Two rounds of the "lowbias32" integer mixer; kept when the 32 bits are >= floor(p * 2^32).
__host__ __device__ __forceinline__ uint32_t drop_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t drop_bits(uint32_t seed_lo, uint32_t seed_hi, uint32_t e, uint32_t h) {
    return drop_mix32(drop_mix32(e ^ seed_lo) ^ (h * 0x9e3779b9u + seed_hi));
}
struct DropArgs {
    uint32_t thr;        // floor(p * 2^32): keep when drop_bits >= thr
    uint32_t seed_lo, seed_hi;
    float inv;           // 1 / (1 - p)
};
__host__ __forceinline__ DropArgs make_drop(float p, uint64_t seed) {
    DropArgs d;
    const double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    d.inv = 1.0f / (1.0f - p);
    return d;
}
__device__ __forceinline__ float softmax_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// Five instructions each (round 6; the comparison chains they replace were nine — propagate(copy_xj, max) ran 36 % slower than `+` on the
// products shape, VALU-bound): the hardware's v_max_f32 / v_min_f32 already order -0 < +0 the way Julia does and return the OTHER operand
// when one is a NaN, so only the NaN case needs a select — x if x is a NaN (it stays: the first NaN met wins, payload and all), else y.
__device__ __forceinline__ float hw_max(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float hw_min(float x, float y) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float jl_max(float x, float y) {
    const float m = hw_max(x, y);
    const float nan_pick = (x != x) ? x : y;
    return __builtin_isunordered(x, y) ? nan_pick : m;
}
__device__ __forceinline__ float jl_min(float x, float y) {
    const float m = hw_min(x, y);
    const float nan_pick = (x != x) ? x : y;
    return __builtin_isunordered(x, y) ? nan_pick : m;
}

enum { OP_SUM = 0, OP_MAX = 1, OP_MIN = 2 };

template <int OP>
__device__ __forceinline__ float op_identity() {
    return OP == OP_SUM ? 0.0f : (OP == OP_MAX ? -__builtin_inff() : __builtin_inff());
}
template <int OP>
__device__ __forceinline__ float op_apply(float a, float b) {
    if (OP == OP_SUM) return a + b;
    if (OP == OP_MAX) return jl_max(a, b);
    return jl_min(a, b);
}

// vector-of-VEC-floats load/store (16/8/4-byte global accesses)
template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using T = float4;
    static __device__ __forceinline__ void load(const float *p, float v[4]) {
        // (a streaming / `nt` hint on these row gathers was tried: 5.18 -> 6.84 ms on the one-pass attention kernel,
        // 4.75 -> 6.52 ms on the GCN propagate, products shape — the L2 line fills it skips are the coalescing buffer of the
        // four lanes that share a 64-byte sector)
        float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float *p, const float v[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct Vec<2> {
    using T = float2;
    static __device__ __forceinline__ void load(const float *p, float v[2]) {
        float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
    static __device__ __forceinline__ void store(float *p, const float v[2]) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    }
};
template <>
struct Vec<1> {
    using T = float;
    static __device__ __forceinline__ void load(const float *p, float v[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float v[1]) { *p = v[0]; }
};

// Device-coherent accesses WITHOUT cache maintenance — for the few values that cross from one workgroup to another inside a kernel (the
// partials of split rows, csr_reduce.h).  The eight XCDs' L2s are not coherent with each other; a __threadfence() makes ordinary stores
// visible by writing back / invalidating the WHOLE L2 of the issuing XCD (buffer_wbl2 / buffer_inv) — measured: 4 000 of those turned a
// 101 us arxiv-shaped propagate into 270 us.  Relaxed agent-scope atomics instead compile to loads / stores with the sc1 bit: they go
// through to the coherence point themselves and leave every other line of the cache alone.  Ordering is then the program's job:
// coh_publish() waits until this lane's stores have been acknowledged before the arrival counter is bumped, and the consumer reads only
// through coh_load after it saw the count (a control dependency: no load is issued before the counter's value is back).
//
// ARCHITECTURE CONTRACT.  This is outside the HIP memory model (relaxed atomics + a hand-written wait give no release / acquire): it is
// correct on the ISA of gfx942 / gfx950 ONLY, where (i) s_waitcnt vmcnt(0) also counts STORES (gfx10+ track them in vscnt: the last
// arriver could fold stale partials there) and (ii) an sc1 access goes through to the device coherence point.  The library therefore
// refuses to compile its device code for anything else (the Makefile's ARCH is overridable); tests/test_fold_isa.py greps the built
// code object: the FOLD kernels must contain sc1 loads / stores and no buffer_wbl2 / buffer_inv.  Escape hatches: knob 19 bit 7 or the
// environment variable GNNMP_NO_FOLD=1 send split rows through the two-kernel path of rounds 1-4 (csr_combine / gat_fused_combine).
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx942__) || defined(__gfx950__))
#error "libgnnmp's cross-workgroup fold (coh_store / coh_load / coh_publish) relies on gfx942 / gfx950 semantics of s_waitcnt vmcnt and sc1; build with ARCH=gfx950"
#endif
template <int VEC>
__device__ __forceinline__ void coh_store(float *p, const float v[VEC]) {
    if (VEC == 1) {
        __hip_atomic_store(reinterpret_cast<unsigned int *>(p), __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int q = 0; q < VEC; q += 2) {
            const unsigned long long b = (unsigned long long)__float_as_uint(v[q]) | ((unsigned long long)__float_as_uint(v[q + 1 < VEC ? q + 1 : q]) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(p + q), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <int VEC>
__device__ __forceinline__ void coh_load(const float *p, float v[VEC]) {
    if (VEC == 1) {
        v[0] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
#pragma unroll
        for (int q = 0; q < VEC; q += 2) {
            const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p + q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[q] = __uint_as_float((unsigned int)b);
            if (q + 1 < VEC) v[q + 1] = __uint_as_float((unsigned int)(b >> 32));
        }
    }
}
__device__ __forceinline__ void coh_store1(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned int *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coh_load1(const float *p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// every memory operation this lane has issued so far has completed (stores acknowledged by the coherence point)
__device__ __forceinline__ void coh_publish() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Sum over aligned groups of LPH adjacent lanes (LPH a power of two; the lanes that hold one attention head).  Every lane
// of the group ends with the same bits.  Stages 1, 2 are quad permutes, 4 and 8 the half-row / row mirrors (after the
// quad stages the mirror partner holds the other half's sum, so it is as good as xor) — all DPP modifiers folded into
// the v_add, no LDS crossbar traffic and nothing to wait for; only the 16- and 32-lane stages need ds_bpermute.
// LPH = 0: lane count known only at run time (`lph`), plain xor butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LPH>
__device__ __forceinline__ float group_sum(float d, int lph) {
    if (LPH == 0) {
        if (lph & 0x10000) {
            // a head of n lanes, n NOT a power of two (lph_code below): every lane reads its head's n lanes in order —
            // n ds_bpermutes instead of log2(n) DPP steps, the price of odd head widths (e.g. C = 7 classes)
            const int n = lph & 0xff, G = 1 << ((lph >> 8) & 0xff);
            const int lane = (int)__lane_id(), lig = lane & (G - 1);
            const int hb = lane - lig + (lig / n) * n;
            float s = 0.0f;
            for (int k = 0; k < n; ++k) s += __shfl(d, hb + k, 64);
            return s;
        }
        for (int o = 1; o < lph; o <<= 1) d += __shfl_xor(d, o, 64);
        return d;
    }
    if (LPH >= 2) d += dpp_mov<0xB1>(d);    // quad_perm [1,0,3,2]
    if (LPH >= 4) d += dpp_mov<0x4E>(d);    // quad_perm [2,3,0,1]
    if (LPH >= 8) d += dpp_mov<0x141>(d);   // row_half_mirror
    if (LPH >= 16) d += dpp_mov<0x140>(d);  // row_mirror
    if (LPH >= 32) d += __shfl_xor(d, 16, 64);
    if (LPH >= 64) d += __shfl_xor(d, 32, 64);
    return d;
}

// what the kernels get as `lph`: the lane count itself when it is a power of two, otherwise a code group_sum<0> decodes
inline int lph_code(int lph, int log2g) {
    return (lph & (lph - 1)) == 0 ? lph : (0x10000 | (log2g << 8) | lph);
}

// block -> logical chunk remap so that each XCD (block b runs on XCD b % 8) walks a contiguous range of
// destination rows; grid must be launched with 8 * cpx blocks.
__device__ __forceinline__ int xcd_remap(int b, int cpx, int enabled) {
    return enabled ? (b & 7) * cpx + (b >> 3) : b;
}
// The same with the first `nbc` blocks left alone: they hold the chunk virtual rows of the long rows (the heaviest work
// items, listed first) and must stay spread over all XCDs — remapping them too put every hub chunk on XCD 0 (arxiv
// shape: 0.138 vs 0.106 ms).  Blocks b >= nbc: XCD = b & 7 walks logical blocks nbc + xcd * cpx ...; grid = nbc + 8 * cpx.
__device__ __forceinline__ int xcd_remap_after(int b, int nbc, int cpx) {
    return b < nbc ? b : nbc + (b & 7) * cpx + ((b - nbc) >> 3);
}

}  // namespace gnnmp

// ---- the opaque plan ---------------------------------------------------------------------------
struct gnnmp_graph {
    int64_t n_src = 0, n_dst = 0, n_edges = 0, n_total = 0;
    int self_loops = 0;
    uint32_t *rowptr = nullptr;  // [n_dst + 1]  UNSIGNED: a plan may hold 2^31 <= E' < 2^32 - 65536 slots
    int32_t *col = nullptr;     // [n_total] 0-based source of each slot (n_src < 2^31)
    int32_t *eid = nullptr;     // [n_total] 0-based original edge position of each slot, to be read as UNSIGNED 32 bits
    // rows longer than long_thresh (sorted ascending) are cut into n_chunks balanced chunks of at most
    // long_thresh slots; the row kernels process chunks like ordinary (virtual) rows into a partial buffer and a
    // small combine kernel folds the partials of each long row in chunk order.
    int32_t *long_rows = nullptr;   // [n_long]
    int32_t *long_cptr = nullptr;   // [n_long + 1] chunk range of each long row
    int32_t *chunk_row = nullptr;   // [n_chunks]
    uint32_t *chunk_beg = nullptr;   // [n_chunks] (slots: unsigned 32-bit)
    uint32_t *chunk_end = nullptr;   // [n_chunks]
    int n_long = 0;
    int n_chunks = 0;
    int long_thresh = GNNMP_LONG_ROW;
    int64_t max_degree = 0;
    int64_t bytes = 0;
    // plan-owned workspace for the chunk partials (grown on demand; one stream at a time per plan)
    float *ws = nullptr;
    size_t ws_floats = 0;
    // device word(s) for the persistent fused kernel's dynamic tile hand-out (fused_conv.hip), zeroed by a memset node before
    // every launch
    uint32_t *ticket = nullptr;
    // destination rows in order of decreasing length (stable), built on first use (ensure_row_order): the row kernels that
    // put TWO OR MORE rows in a wave walk rows in this order so that the rows sharing a wave are equally long — adjacent
    // rows of a power-law graph are not, and a wave runs as long as its longest row (products shape: 1.43x the instruction
    // issue of perfectly packed rows with adjacent pairing, 1.13x with this order)
    int32_t *row_order = nullptr;
    // fold-in-kernel (round 5): the long row (index into long_rows) of every chunk, and the arrival counters of the long rows — zero
    // between launches (the last arriver resets its counter), grown on demand like ws (ensure_arrive)
    int32_t *chunk_lrow = nullptr;  // [n_chunks]
    uint32_t *arrive = nullptr;
    size_t arrive_n = 0;
    float *spart = nullptr;         // slice partials of the long rows (csr_reduce.h: long_geom)
    size_t spart_floats = 0;
    // plans made by gnnmp_plan_concat / gnnmp_plan_select (plan_batch.hip): rowptr, col, eid (and the member table) live in ONE block taken
    // from the stream-ordered pool (pool.h); gnnmp_plan_release hands it back without a host synchronisation
    void *block = nullptr;
    size_t block_bytes = 0;
    int32_t *status = nullptr;     // (pooled plans) device word: non-zero = the caller's totals did not match the member table
};

namespace gnnmp {
// make sure plan->ws holds at least `floats` floats (hipMalloc on growth; hipFree waits for in-flight work)
int ensure_workspace(gnnmp_graph *p, size_t floats);
// make sure plan->arrive holds at least n zeroed counters and plan->spart at least `floats` floats.
// INVARIANT: every arrival counter of a plan is ZERO whenever no compute call on that plan is in flight — the last arriver of a slice /
// row resets its counter inside the launch that raised it.  It holds as long as (i) a plan is used by one stream at a time (gnnmp.h:
// the plan's workspace rule — two concurrent launches on one plan would also share `ws` / `spart`) and (ii) every FOLD launch runs to
// completion.  A launch that is REFUSED (hipGetLastError at the launch site) never touched the counters; a launch that faults on the
// device poisons the HIP context and with it every later call.  The one remaining way to leave them dirty — a caller breaking (i) — is
// what gnnmp_plan_reset_counters (gnnmp.h) repairs: a stream-ordered memset of the counters, a few KB.
int ensure_arrive(gnnmp_graph *p, size_t n, size_t floats, hipStream_t stream);
// split rows folded inside the row kernels (default) or by the combine kernels of rounds 1-4 (knob 19 bit 7, or GNNMP_NO_FOLD=1 in the
// environment of the process, read once)?
bool fold_disabled_by_env();
inline bool use_fold() { return (knob(KNOB_VARIANT) & 128) == 0 && !fold_disabled_by_env(); }
// allocate + zero plan->ticket on first use
int ensure_ticket(gnnmp_graph *p, hipStream_t stream);
// build plan->row_order on first use
int ensure_row_order(gnnmp_graph *p, hipStream_t stream);
// max_degree, n_long and the chunk tables of the rows longer than plan->long_thresh from plan->rowptr (synchronises the stream)
int plan_build_long_rows(gnnmp_graph *p, hipStream_t stream);
// the long-row threshold gnnmp_plan_create picks for a plan of Etot slots
int plan_long_thresh(int64_t Etot);
}

namespace gnnmp {
// vector width usable for rows of D floats at these base pointers
inline int pick_vec(int64_t D, const void *a, const void *b) {
    uintptr_t m = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    int forced = knob(KNOB_FORCE_VEC);
    int v = 1;
    if (D % 4 == 0 && (m & 15) == 0)
        v = 4;
    else if (D % 2 == 0 && (m & 7) == 0)
        v = 2;
    if (forced == 1 || forced == 2 || forced == 4) {
        if (forced <= v) v = forced;
    }
    return v;
}
// XCD-contiguous block -> row-chunk mapping?  Measured on MI355X: +4 % on the arxiv shape (features resident in the 256 MiB
// Infinity Cache: the per-XCD L2s then hold disjoint destination ranges), -3..5 % on the products shape (980 MB of
// features, no reuse to protect: the remap only concentrates each XCD's streaming on one address range).
inline bool use_xcd_remap(int64_t n_src, int64_t D, int64_t chunks) {
    const int k = knob(KNOB_XCD_REMAP);
    if (k == 0 || chunks < 64) return false;
    if (k == 2) return true;
    return n_src * D * (int64_t)sizeof(float) <= ((int64_t)128 << 20);
}
// Walk rows by decreasing length (gnnmp_graph::row_order)?  OFF by default (knob 15 = 0).  The idea: a wave shares its lanes
// between two or more rows and runs as long as the longest, so pairing rows of equal length removes idle issue slots (products
// shape: 1.43x -> 1.13x of the perfectly packed instruction count).  Measured on the one-pass attention kernel, products shape:
// one box 5.08 -> 4.92 ms, five later boxes 5.17 -> 5.38 ms (same binary, same-box A/B each time; sustained back-to-back
// launches 4.82 vs 4.92 ms at 1 350-1 370 W of the 1 400 W socket limit): the kernel sits at the HBM / power limit and the
// scattered row order costs as much as the saved issue slots give.  With 400-byte rows (D = 100) it is plainly worse (every
// 128-byte line of the output is finished by a different wave: 4.75 -> 5.11 ms), and on cache-resident graphs too (arxiv shape
// 175 -> 184 us; 261 us with the XCD-contiguous block mapping, which puts all heavy rows on one XCD).  knob 15 = 1: on when
// the gathered matrix exceeds the Infinity Cache; 2: always.
inline bool use_row_order(int64_t n_src, int64_t D) {
    const int k = knob(KNOB_ROW_ORDER);
    if (k == 0) return false;
    if (k == 2) return true;
    return n_src * D * (int64_t)sizeof(float) > ((int64_t)128 << 20);
}
// lanes per row group: smallest power of two >= D / vec, clamped to [1, 64]
inline int pick_log2g(int64_t lanes_needed) {
    int forced = knob(KNOB_FORCE_LOG2G);
    if (forced >= 0 && forced <= 6) return forced;
    int l = 0;
    while ((1 << l) < lanes_needed && l < 6) ++l;
    return l;
}
}  // namespace gnnmp
