// arena.hip — a placement-aware device arena for the OUTPUTS of the gather kernels.
//
// Measured on MI355X (tools/experiments/placement_probe.py, placement_map*.py, vmm_probe.py; profiles/README.md round 4): the 288 GB of HBM3E fall into
// THREE placement classes of 96 GiB of physical memory each (the 12-high stacks: three groups of four dies).  A gather kernel — ~27 random row
// reads per row written — whose gathered matrix and whose output lie in the SAME class runs 6 % slower than with the two in different classes
// (one-pass attention kernel 5.15 vs 4.84 ms, fused GCN layer 4.94 vs 4.66 ms on the products shape; same binary, same data, same
// predecessors on the stream): the writes land on the dies the read stream is saturating.  hipMalloc does not say where memory lies, and a
// single allocation is a patchwork of power-of-two blocks of any class, so two ordinary allocations pair up by luck (box-to-box "noise").
//
// The arena takes the luck out: blocks of 2 GiB are allocated one by one (plain hipMalloc), classified where they lie by timing a small probe
// — a propagate(copy_xj, +) over a synthetic random graph, 262 144 rows x 26 sources of 512 bytes out of a reference block, ~550 us, 10 %
// apart between "output in the reference's class" and "not" — at four places of the block; blocks of the three (or two) classes are kept,
// the others freed at the end.  A caller (the host mirror's layers, the Julia extension) allocates a layer's output from a block whose
// class differs from the gathered matrix's class:
//     gnnmp_arena_class_of(arena, x)            -> 0 / 1 (x is in that arena class), 2 (in neither: any range is fine)
//     gnnmp_arena_alloc(arena, 1 - cls, bytes)  -> the output buffer
// The C ABI's rule — the caller allocates — stands: this is an allocator the caller MAY use.  Creation synchronises (graph prep); alloc
// and class_of on arena memory do not; class_of on foreign memory runs the probe (synchronises).
//
// Round 5 — an allocator a caller can leave on.  Creation works inside a BUDGET: at most max_probe_bytes of device memory held at any moment
// (default 32 GiB = 16 blocks) and at most ~0.3 s (GNNMP_ARENA_BUDGET_MS); a block is first screened at ONE window (1 warm-up + 3 timed
// probes, ~2.5 ms) and only the blocks that are kept are checked at all four.  When the budget runs out the arena is returned with the classes
// it found (gnnmp_arena_info: [7] = classes, [9] = 1 "gave up") instead of an error: two classes still separate a gather's source from
// its output, one class means "no placement on this device today" (the caller allocates as usual).  A buffer larger than a block is an
// allocation of its own, classified window by window when it is asked for (gnnmp_arena_alloc).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"

struct gnnmp_arena {
    int64_t block_bytes = 0;
    int n_classes = 2;                                 // classes that exist (2 | 3)
    struct Block { unsigned char *p; int64_t used; };
    std::vector<Block> blocks[3];                      // per class: plain hipMalloc'd blocks whose two halves probed alike
    int64_t cap = 0;                                   // bytes per class
    gnnmp_graph_t *probe_plan = nullptr;               // the synthetic graph of the probe (sources in [0, probe_nsrc))
    int64_t probe_nsrc = 0;
    int dev = 0;
    int64_t created = 0, released = 0;                 // blocks made / given back during classification
    float probe_same_us = 0.0f, probe_other_us = 0.0f; // what the probe measured (info)
    int gave_up = 0;                                   // the budget ran out before every class held its blocks
    int64_t create_us = 0;                             // wall time of gnnmp_arena_create
    // buffers larger than a block: allocations of their own, classified window by window when they are made (gnnmp_arena_alloc)
    struct Big { unsigned char *p; int64_t bytes; int cls; bool used; };
    std::vector<Big> big;
    float thr = 0.0f;                                  // the probe time that separates "same class as the source" from "another"
    const float *ref[2] = {nullptr, nullptr};          // the probe's sources: the first GiB of a block of class 0 / class 1
    std::mutex lock;
};

namespace gnnmp {
namespace {
constexpr int64_t CHUNK = (int64_t)2 << 30;      // a block: one hipMalloc; classified at four places (every 512 MiB)
constexpr int PROBE_ROWS = 262144, PROBE_DEG = 26, PROBE_D = 128;    // 512-byte rows like the attention kernel's; the two cases are 5 % apart at 65 536 rows, 7 % here

__global__ void probe_edges_kernel(int64_t n_src, int64_t *src, int64_t *dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)PROBE_ROWS * PROBE_DEG) return;
    const uint32_t r = drop_mix32(drop_mix32((uint32_t)e ^ 0x9e3779b9u) + 0x7f4a7c15u);
    src[e] = (int64_t)(((uint64_t)r * (uint64_t)n_src) >> 32) + 1;
    dst[e] = e / PROBE_DEG + 1;
}

// median of `reps` (odd, <= 5) timed launches of the probe after `warm` untimed ones: source rows at `src`, output at `out` (PROBE_ROWS x
// PROBE_D floats); microseconds
int probe_us(gnnmp_arena *a, const float *src, float *out, hipStream_t stream, float *us, int warm = 2, int reps = 5) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    GNNMP_HIP(hipEventCreate(&e0));
    GNNMP_HIP(hipEventCreate(&e1));
    float t[5];
    int rc = GNNMP_OK;
    for (int it = -warm; it < reps && rc == GNNMP_OK; ++it) {
        hipError_t e = hipEventRecord(e0, stream);
        if (e == hipSuccess) {
            rc = gnnmp_propagate_f32(a->probe_plan, GNNMP_COPY_XJ, GNNMP_SUM, src, nullptr, nullptr, nullptr, out, PROBE_D, stream);
            if (rc != GNNMP_OK) break;
            e = hipEventRecord(e1, stream);
        }
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) { rc = hip_fail(e, "arena probe"); break; }
        if (it >= 0) t[it] = ms * 1e3f;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != GNNMP_OK) return rc;
    std::sort(t, t + reps);
    *us = t[reps / 2];
    return GNNMP_OK;
}

// a block probed at four places, every 512 MiB (an allocation is a patchwork of whatever physical blocks were free: only blocks whose
// windows agree are used); *lo / *hi = the smallest / largest of the four times
int probe_chunk(gnnmp_arena *a, const float *src, unsigned char *chunk, hipStream_t stream, float *lo, float *hi) {
    *lo = 1e30f;
    *hi = 0.0f;
    for (int k = 0; k < 4; ++k) {
        float t = 0.0f;
        int rc = probe_us(a, src, reinterpret_cast<float *>(chunk + (int64_t)k * (CHUNK / 4)), stream, &t, 1, 3);
        if (rc != GNNMP_OK) return rc;
        *lo = std::min(*lo, t);
        *hi = std::max(*hi, t);
    }
    return GNNMP_OK;
}
// the screening probe: one window (the block's second quarter), 1 warm-up + 3 timed launches
int probe_quick(gnnmp_arena *a, const float *src, unsigned char *chunk, hipStream_t stream, float *us) {
    return probe_us(a, src, reinterpret_cast<float *>(chunk + CHUNK / 4), stream, us, 1, 3);
}

int make_probe_plan(gnnmp_arena *a, int64_t n_src, hipStream_t stream) {
    if (a->probe_plan && a->probe_nsrc == n_src) return GNNMP_OK;
    if (a->probe_plan) { gnnmp_plan_destroy(a->probe_plan); a->probe_plan = nullptr; }
    const int64_t E = (int64_t)PROBE_ROWS * PROBE_DEG;
    int64_t *s = nullptr, *t = nullptr;
    GNNMP_HIP(hipMalloc((void **)&s, sizeof(int64_t) * 2 * (size_t)E));
    t = s + E;
    probe_edges_kernel<<<(unsigned)((E + 255) / 256), 256, 0, stream>>>(n_src, s, t);
    int rc = gnnmp_plan_create(&a->probe_plan, s, t, 8, 1, n_src, PROBE_ROWS, E, 0, 0, stream);
    (void)hipFree(s);
    if (rc == GNNMP_OK) a->probe_nsrc = n_src;
    return rc;
}
}  // namespace
}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_arena_destroy(gnnmp_arena_t *a) {
    if (!a) return GNNMP_OK;
    for (int c = 0; c < 3; ++c)
        for (auto &b : a->blocks[c]) (void)hipFree(b.p);      // (hipFree waits for the device)
    for (auto &b : a->big) (void)hipFree(b.p);
    if (a->probe_plan) gnnmp_plan_destroy(a->probe_plan);
    delete a;
    return GNNMP_OK;
}

int gnnmp_arena_create(gnnmp_arena_t **out, int64_t bytes_per_class, int n_classes, int64_t max_probe_bytes, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out || bytes_per_class <= 0 || (n_classes != 2 && n_classes != 3)) return fail(GNNMP_EINVAL, "arena_create: bad argument");
    *out = nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    gnnmp_arena *a = new gnnmp_arena();
    a->dev = current_device();
    a->block_bytes = CHUNK;
    a->n_classes = n_classes;
    const int need = (int)((bytes_per_class + CHUNK - 1) / CHUNK);
    a->cap = (int64_t)need * CHUNK;
    if (max_probe_bytes <= 0) max_probe_bytes = (int64_t)32 << 30;
    // at any moment the arena holds what it probed so far (a freed block would come straight back from hipMalloc): the budget bounds that
    const int max_blocks = (int)std::max<int64_t>(2, max_probe_bytes / CHUNK);
    double budget_ms = 300.0;
    if (const char *e = getenv("GNNMP_ARENA_BUDGET_MS")) budget_ms = std::max(1.0, atof(e));
    const bool debug = getenv("GNNMP_ARENA_DEBUG") != nullptr;
    // Blocks are plain hipMalloc allocations, each classified where it lies and then kept or freed — no remapping.  (A first version built
    // the ranges out of hipMemCreate chunks moved between addresses with hipMemMap / hipMemUnmap: translations of an unmapped range
    // outlived the unmap — 80 different chunks probed through one address measured alike — and two runs in six ended in a GPU memory
    // fault.)
    struct Pend { unsigned char *p; float t; };
    std::vector<Pend> pending;
    std::vector<unsigned char *> spare;                      // blocks of a class that is full / mixed blocks: freed at the end (freeing
                                                             // one earlier would hand the same physical memory out again)
    int rc = GNNMP_OK;
    float lo = 0.0f, hi = 0.0f, thr = 0.0f;
    const float *ref0 = nullptr, *ref1 = nullptr;
    auto keep = [&](unsigned char *p, int cls) {
        if (cls < a->n_classes && (int)a->blocks[cls].size() < need) a->blocks[cls].push_back({p, 0});
        else spare.push_back(p);
    };
    // (consumes q.p on every path: kept, spare, or — on an error — spare)
    auto classify = [&](const Pend &q) -> int {
        int cls;
        if (q.t > thr) {
            cls = 0;                                         // as slow as block 0's own class
        } else if (!ref1) {
            cls = 1;
            ref1 = reinterpret_cast<const float *>(q.p);
            if (hipMemsetAsync(q.p, 0, (size_t)CHUNK / 2, stream) != hipSuccess) {
                spare.push_back(q.p);
                return fail(GNNMP_ELAUNCH, "arena: memset of the second reference");
            }
        } else {
            float t1 = 0.0f;
            const int r2 = probe_quick(a, ref1, q.p, stream, &t1);
            if (r2 != GNNMP_OK) { spare.push_back(q.p); return r2; }
            cls = t1 > thr ? 1 : 2;
            if (debug) fprintf(stderr, "[arena]   against the class-1 reference %.1f us -> class %d\n", t1, cls);
        }
        keep(q.p, cls);
        return GNNMP_OK;
    };
    auto all_full = [&]() {
        for (int c = 0; c < a->n_classes; ++c)
            if ((int)a->blocks[c].size() < need) return false;
        return true;
    };
    rc = make_probe_plan(a, (CHUNK / 2) / (PROBE_D * 4), stream);      // the probe gathers from the first GiB of a reference block
    if (rc == GNNMP_OK) {
        unsigned char *p0 = nullptr;
        hipError_t e = hipMalloc((void **)&p0, (size_t)CHUNK);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(arena block 0)");
        else {
            ++a->created;
            a->blocks[0].push_back({p0, 0});                 // block 0 is class 0 by definition and the probe's source from now on
            ref0 = reinterpret_cast<const float *>(p0);
            e = hipMemsetAsync(p0, 0, (size_t)CHUNK / 2, stream);      // (the GiB the probe gathers from)
            if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync(arena block 0)");
        }
    }
    while (rc == GNNMP_OK && !all_full()) {
        if (a->created >= max_blocks || elapsed_ms() > budget_ms) {      // out of budget: the caller gets what was found
            a->gave_up = 1;
            break;
        }
        unsigned char *p = nullptr;
        hipError_t e = hipMalloc((void **)&p, (size_t)CHUNK);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            a->gave_up = 1;                                  // device memory exhausted: the same graceful end
            break;
        }
        ++a->created;
        Pend q = {p, 0.0f};
        rc = probe_quick(a, ref0, p, stream, &q.t);
        if (rc != GNNMP_OK) { spare.push_back(p); break; }
        if (debug) fprintf(stderr, "[arena] block %lld @ %p: probe against block 0 %.1f us (%.0f ms)\n", (long long)a->created - 1, (void *)p, q.t, elapsed_ms());
        lo = lo == 0.0f ? q.t : std::min(lo, q.t);
        hi = std::max(hi, q.t);
        if (thr == 0.0f) {
            pending.push_back(q);
            if (hi > 1.06f * lo) {                        // both clusters seen (pure pairs are 10 % apart): decide everything held back
                thr = 0.5f * (lo + hi);
                a->probe_same_us = hi;
                a->probe_other_us = lo;
                size_t i = 0;
                for (; i < pending.size() && rc == GNNMP_OK; ++i) rc = classify(pending[i]);
                for (; i < pending.size(); ++i) spare.push_back(pending[i].p);
                pending.clear();
            }
        } else {
            rc = classify(q);
        }
    }
    // no second cluster seen: everything probed so far lies in block 0's class (keep what class 0 still needs of it)
    for (const Pend &q : pending) keep(q.p, 0);
    pending.clear();
    // the kept blocks at all four windows against the reference: every window of a class-0 block slow, of any other block fast; a block
    // that fails is a patchwork (or was mis-screened) and is dropped
    if (thr > 0.0f) {
        for (int c = 0; c < a->n_classes && rc == GNNMP_OK; ++c)
            for (size_t i = (c == 0 ? 1 : 0); i < a->blocks[c].size() && rc == GNNMP_OK;) {
                float l2 = 0.0f, h2 = 0.0f;
                rc = probe_chunk(a, ref0, a->blocks[c][i].p, stream, &l2, &h2);
                const bool bad = rc == GNNMP_OK && (c == 0 ? l2 <= thr : h2 > thr);
                if (debug) fprintf(stderr, "[arena] class %d block %zu: four windows %.1f .. %.1f us%s\n", c, i, l2, h2, bad ? " -> dropped" : "");
                if (bad && reinterpret_cast<const float *>(a->blocks[c][i].p) != ref1) {
                    spare.push_back(a->blocks[c][i].p);
                    a->blocks[c].erase(a->blocks[c].begin() + (long)i);
                    a->gave_up = 1;
                } else {
                    ++i;
                }
            }
    }
    if (rc == GNNMP_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(GNNMP_ELAUNCH, "arena_create: stream synchronisation failed");
    (void)hipDeviceSynchronize();
    for (unsigned char *p : spare) { (void)hipFree(p); ++a->released; }
    if (rc != GNNMP_OK) {
        gnnmp_arena_destroy(a);
        return rc;
    }
    // classes that exist: the non-empty ones, renumbered 0 .. n - 1 in their order; blocks of a class sorted by address (adjacency: alloc)
    int found = 0;
    for (int c = 0; c < 3; ++c) {
        if (a->blocks[c].empty()) continue;
        std::sort(a->blocks[c].begin(), a->blocks[c].end(), [](const gnnmp_arena::Block &x, const gnnmp_arena::Block &y) { return x.p < y.p; });
        if (found != c) { a->blocks[found] = std::move(a->blocks[c]); a->blocks[c].clear(); }
        ++found;
    }
    a->n_classes = found;
    a->thr = thr;
    a->ref[0] = ref0;                                        // (block 0 is never dropped; ref1's block is kept even if a window of it failed)
    a->ref[1] = ref1;
    a->create_us = (int64_t)(elapsed_ms() * 1e3);
    *out = a;
    return GNNMP_OK;
}

// an arena belongs to the device it was created on: its blocks live there and the probe runs there
static int arena_check_device(const gnnmp_arena *a, const char *who) {
    const int dev = current_device();
    if (dev != a->dev) return fail(GNNMP_EINVAL, "%s: the arena belongs to device %d, the current device is %d", who, a->dev, dev);
    return GNNMP_OK;
}

int gnnmp_arena_alloc(gnnmp_arena_t *a, int cls, int64_t bytes, void **ptr) {
    if (!a || !ptr || bytes < 0 || cls < 0 || cls >= a->n_classes) return fail(GNNMP_EINVAL, "arena_alloc: bad argument");
    if (int rc = arena_check_device(a, "arena_alloc")) return rc;
    std::lock_guard<std::mutex> lk(a->lock);
    auto &bl = a->blocks[cls];
    if (bytes <= a->block_bytes) {
        for (auto &b : bl) {                                  // inside ONE block
            const int64_t at = (b.used + 4095) & ~(int64_t)4095;
            if (at + bytes <= a->block_bytes) {
                *ptr = b.p + at;
                b.used = at + bytes;
                return GNNMP_OK;
            }
        }
    } else {
        // Larger than a block (SAGEConv's 2.5 GB output on the products shape).  Separate hipMallocs are never adjacent in the address space
        // (measured: 2 GiB + 2 MiB apart), so such a buffer is an allocation of its own, classified where it lies — every 512 MiB window
        // probed against the class references, all windows must agree — and kept if it came out in class `cls`; up to three tries, the
        // rejects held until the end (freed earlier they would come straight back).  SYNCHRONISES THE DEVICE and holds the arena's lock
        // for the whole of it (multi-GB hipMallocs + probes: tens of milliseconds) — this is set-up, the buffers are persistent, but a
        // layer that asks for a placed output (gnnmp.placement / sage_conv's buffer_for) pays it on its FIRST call: gnnmp.h says so.
        // GNNMP_EALLOC = not today, the caller allocates as usual.
        for (auto &b : a->big)
            if (!b.used && b.cls == cls && b.bytes >= bytes && b.bytes <= bytes + (bytes >> 2)) { b.used = true; *ptr = b.p; return GNNMP_OK; }
        if (a->thr <= 0.0f || !a->ref[0]) return fail(GNNMP_EALLOC, "arena_alloc: no class references to classify a %lld-byte buffer", (long long)bytes);
        // (gnnmp_arena_class_of on foreign memory may have rebuilt the probe's graph for a smaller source range: the threshold belongs to this one)
        if (int rcp = make_probe_plan(a, (CHUNK / 2) / (PROBE_D * 4), nullptr)) return rcp;
        const int64_t win = CHUNK / 4, nb = (bytes + win - 1) / win * win;
        std::vector<unsigned char *> rejects;
        bool spare_kept = false;
        int rc = GNNMP_EALLOC;
        for (int attempt = 0; attempt < 3 && rc == GNNMP_EALLOC; ++attempt) {
            unsigned char *p = nullptr;
            if (hipMalloc((void **)&p, (size_t)nb) != hipSuccess) { (void)hipGetLastError(); break; }
            int got = -1;                                    // the class all windows agree on; -2 = they do not
            for (int64_t w = 0; w < nb / win && got != -2; ++w) {
                float t0 = 0.0f, t1 = 0.0f;
                int c = -1;
                int r = probe_us(a, a->ref[0], reinterpret_cast<float *>(p + w * win), nullptr, &t0, 1, 3);
                if (r == GNNMP_OK && t0 > a->thr) c = 0;
                else if (r == GNNMP_OK && a->ref[1]) {
                    r = probe_us(a, a->ref[1], reinterpret_cast<float *>(p + w * win), nullptr, &t1, 1, 3);
                    if (r == GNNMP_OK) c = t1 > a->thr ? 1 : 2;
                } else if (r == GNNMP_OK) c = 1;              // not class 0, and no second reference: "the other one"
                if (r != GNNMP_OK) { got = -2; rc = r; break; }
                got = got == -1 ? c : (got == c ? got : -2);
            }
            if (getenv("GNNMP_ARENA_DEBUG")) fprintf(stderr, "[arena] big buffer of %lld bytes, attempt %d: class %d (wanted %d)\n", (long long)nb, attempt, got, cls);
            if (got == cls) {
                a->big.push_back({p, nb, cls, true});
                *ptr = p;
                rc = GNNMP_OK;
            } else if (got >= 0 && got < a->n_classes && !spare_kept) {
                // a pure buffer of another class: somebody may ask for it — but at most ONE unused spare is ever retained per arena (the
                // first version kept every miss until gnnmp_arena_destroy: one 2.5 GB request could pin 7.5 GB)
                bool have_spare = false;
                for (const auto &b : a->big) have_spare |= !b.used;
                if (have_spare) rejects.push_back(p);
                else { a->big.push_back({p, nb, got, false}); spare_kept = true; }
            } else {
                rejects.push_back(p);
            }
        }
        (void)hipDeviceSynchronize();                        // (the probes ran on the null stream; rejects are freed behind them)
        for (unsigned char *p : rejects) (void)hipFree(p);
        if (rc == GNNMP_OK) return GNNMP_OK;
        if (rc != GNNMP_EALLOC) return rc;
    }
    return fail(GNNMP_EALLOC, "arena_alloc: class %d cannot hold %lld bytes (blocks of %lld bytes, %zu of them; a larger buffer is an "
                              "allocation of its own that did not come out in that class)", cls, (long long)bytes, (long long)a->block_bytes,
                bl.size());
}

int gnnmp_arena_reset(gnnmp_arena_t *a) {
    if (!a) return fail(GNNMP_EINVAL, "arena_reset: null arena");
    std::lock_guard<std::mutex> lk(a->lock);
    for (int c = 0; c < 3; ++c)
        for (auto &b : a->blocks[c]) b.used = 0;
    for (auto &b : a->big) b.used = false;
    return GNNMP_OK;
}

int gnnmp_arena_class_of(gnnmp_arena_t *a, const void *ptr, int64_t bytes, int *cls, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || !ptr || !cls) return fail(GNNMP_EINVAL, "arena_class_of: bad argument");
    if (int rc = arena_check_device(a, "arena_class_of")) return rc;
    const unsigned char *p = static_cast<const unsigned char *>(ptr);
    for (int c = 0; c < a->n_classes; ++c)
        for (const auto &b : a->blocks[c])
            if (p >= b.p && p < b.p + a->block_bytes) { *cls = c; return GNNMP_OK; }
    for (const auto &b : a->big)
        if (p >= b.p && p < b.p + b.bytes) { *cls = b.cls; return GNNMP_OK; }
    // foreign memory: the probe with the buffer as the gathered matrix, the output in a spare corner of each arena range
    const int none = a->n_classes;                         // "in none of the ranges' classes / cannot tell"
    const int64_t n_src = std::min<int64_t>(bytes, CHUNK / 2) / (PROBE_D * 4);
    if (n_src < 4 * PROBE_ROWS) { *cls = none; return GNNMP_OK; }      // (under 512 MiB: served by the Infinity Cache, no contrast)
    if ((reinterpret_cast<uintptr_t>(ptr) & 15)) { *cls = none; return GNNMP_OK; }
    std::lock_guard<std::mutex> lk(a->lock);
    int rc = make_probe_plan(a, n_src, stream);
    if (rc != GNNMP_OK) return rc;
    const int64_t out_bytes = (int64_t)PROBE_ROWS * PROBE_D * 4;
    float us[3] = {0.0f, 0.0f, 0.0f};
    for (int c = 0; c < a->n_classes; ++c) {
        const auto &b = a->blocks[c].back();
        if (a->block_bytes - b.used < out_bytes + 4096) { *cls = none; return GNNMP_OK; }      // no room for the probe's output: cannot tell
        float *o = reinterpret_cast<float *>(b.p + a->block_bytes - out_bytes);      // the top of the block: free by the check above
        rc = probe_us(a, static_cast<const float *>(ptr), o, stream, &us[c]);
        if (rc != GNNMP_OK) return rc;
    }
    if (getenv("GNNMP_ARENA_DEBUG"))
        fprintf(stderr, "[arena] foreign buffer %p: probe %.1f / %.1f / %.1f us into ranges 0 / 1 / 2\n", ptr, us[0], us[1], us[2]);
    int slow = 0;
    float lo = us[0];
    for (int c = 1; c < a->n_classes; ++c) {
        if (us[c] > us[slow]) slow = c;
        lo = std::min(lo, us[c]);
    }
    *cls = us[slow] > 1.06f * lo ? slow : none;
    return GNNMP_OK;
}

/* info[0] = bytes per class asked for, [1] = bytes used in range 0, [2] = in range 1, [3] = blocks created while classifying, [4] = released
 * again, [5] = probe microseconds with source and output in one class, [6] = in two, [7] = classes the arena HOLDS (0 .. 3: fewer than asked
 * for when the budget ran out), [8] = bytes used in range 2, [9] = 1 if creation gave up on its budget (bytes or time) before every class
 * held its blocks, [10] / [11] / [12] = blocks held in range 0 / 1 / 2, [13] = microseconds gnnmp_arena_create took */
int gnnmp_arena_info(const gnnmp_arena_t *a, int64_t *info) {
    if (!a || !info) return fail(GNNMP_EINVAL, "arena_info: null argument");
    int64_t used[3] = {0, 0, 0};
    for (int c = 0; c < 3; ++c)
        for (const auto &b : a->blocks[c]) used[c] += b.used;
    info[0] = a->cap; info[1] = used[0]; info[2] = used[1]; info[3] = a->created; info[4] = a->released;
    info[5] = (int64_t)(a->probe_same_us + 0.5f); info[6] = (int64_t)(a->probe_other_us + 0.5f);
    info[7] = a->n_classes; info[8] = used[2];
    info[9] = a->gave_up;
    for (int c = 0; c < 3; ++c) info[10 + c] = (int64_t)a->blocks[c].size();
    info[13] = a->create_us;
    return GNNMP_OK;
}

}  // extern "C"
