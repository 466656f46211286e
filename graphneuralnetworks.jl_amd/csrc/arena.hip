// arena.hip — a placement-aware device arena for the OUTPUTS of the gather kernels.
//
// Measured on MI355X (tools/placement_probe.py, placement_map*.py, vmm_probe.py; profiles/README.md round 4): the 288 GB of HBM3E fall into
// THREE placement classes of 96 GiB of physical memory each (the 12-high stacks: three groups of four dies).  A gather kernel — ~27 random row
// reads per row written — whose gathered matrix and whose output lie in the SAME class runs 6 % slower than with the two in different classes
// (one-pass attention kernel 5.15 vs 4.84 ms, fused GCN layer 4.94 vs 4.66 ms on the products shape; same binary, same data, same
// predecessors on the stream): the writes land on the dies the read stream is saturating.  hipMalloc does not say where memory lies, and a
// single allocation is a patchwork of power-of-two blocks of any class, so two ordinary allocations pair up by luck (box-to-box "noise").
//
// The arena takes the luck out: physical chunks of 2 GiB are created one by one (hipMemCreate), classified by timing a small probe — a
// propagate(copy_xj, +) over a synthetic random graph, 262 144 rows x 26 sources of 512 bytes, ~550 us, 7 % apart between the two cases —
// and the chunks of two different classes are mapped back to back into two address ranges (hipMemMap).  A caller (the host mirror's layers,
// the Julia extension) allocates a layer's output from the range whose class differs from the gathered matrix's class:
//     gnnmp_arena_class_of(arena, x)            -> 0 / 1 (x is in that arena class), 2 (in neither: any range is fine)
//     gnnmp_arena_alloc(arena, 1 - cls, bytes)  -> the output buffer
// The C ABI's rule — the caller allocates — stands: this is an allocator the caller MAY use.  Creation synchronises (graph prep); alloc
// and class_of on arena memory do not; class_of on foreign memory runs the probe (synchronises).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"

struct gnnmp_arena {
    int64_t chunk_bytes = 0;
    int n_classes = 2;                                 // ranges that exist (2 | 3)
    int n_chunks[3] = {0, 0, 0};
    std::vector<hipMemGenericAllocationHandle_t> handles[3];
    unsigned char *base[3] = {nullptr, nullptr, nullptr};      // the mapped ranges
    int64_t cap[3] = {0, 0, 0}, used[3] = {0, 0, 0};
    gnnmp_graph_t *probe_plan = nullptr;               // the synthetic graph of the probe (sources in [0, probe_nsrc))
    int64_t probe_nsrc = 0;
    int dev = 0;
    int64_t created = 0, released = 0;                 // chunks made / given back during classification
    float probe_same_us = 0.0f, probe_other_us = 0.0f; // what the probe measured on the reference pair (info)
    std::mutex lock;
};

namespace gnnmp {
namespace {
constexpr int64_t CHUNK = (int64_t)2 << 30;
constexpr int PROBE_ROWS = 262144, PROBE_DEG = 26, PROBE_D = 128;    // 512-byte rows like the attention kernel's; the two cases are 5 % apart at 65 536 rows, 7 % here

__global__ void probe_edges_kernel(int64_t n_src, int64_t *src, int64_t *dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)PROBE_ROWS * PROBE_DEG) return;
    const uint32_t r = drop_mix32(drop_mix32((uint32_t)e ^ 0x9e3779b9u) + 0x7f4a7c15u);
    src[e] = (int64_t)(((uint64_t)r * (uint64_t)n_src) >> 32) + 1;
    dst[e] = e / PROBE_DEG + 1;
}

hipMemAllocationProp chunk_prop(int dev) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = dev;
    return p;
}
hipError_t map_rw(void *va, size_t size, hipMemGenericAllocationHandle_t h, int dev) {
    hipError_t e = hipMemMap(va, size, 0, h, 0);
    if (e != hipSuccess) return e;
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = dev;
    d.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(va, size, &d, 1);
}

// median of `reps` timed launches of the probe: source rows at `src`, output at `out` (PROBE_ROWS x PROBE_D floats); microseconds
int probe_us(gnnmp_arena *a, const float *src, float *out, hipStream_t stream, float *us) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    GNNMP_HIP(hipEventCreate(&e0));
    GNNMP_HIP(hipEventCreate(&e1));
    float t[5];
    int rc = GNNMP_OK;
    for (int it = -2; it < 5 && rc == GNNMP_OK; ++it) {
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
    std::sort(t, t + 5);
    *us = t[2];
    return GNNMP_OK;
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
    (void)hipDeviceSynchronize();
    for (int c = 0; c < 3; ++c) {
        for (size_t i = 0; i < a->handles[c].size(); ++i) {
            if (a->base[c]) (void)hipMemUnmap(a->base[c] + (int64_t)i * a->chunk_bytes, (size_t)a->chunk_bytes);
            (void)hipMemRelease(a->handles[c][i]);
        }
        if (a->base[c]) (void)hipMemAddressFree(a->base[c], (size_t)a->cap[c]);
    }
    if (a->probe_plan) gnnmp_plan_destroy(a->probe_plan);
    delete a;
    return GNNMP_OK;
}

int gnnmp_arena_create(gnnmp_arena_t **out, int64_t bytes_per_class, int n_classes, int64_t max_probe_bytes, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out || bytes_per_class <= 0 || (n_classes != 2 && n_classes != 3)) return fail(GNNMP_EINVAL, "arena_create: bad argument");
    *out = nullptr;
    gnnmp_arena *a = new gnnmp_arena();
    (void)hipGetDevice(&a->dev);
    a->chunk_bytes = CHUNK;
    a->n_classes = n_classes;
    const int need = (int)((bytes_per_class + CHUNK - 1) / CHUNK);
    if (max_probe_bytes <= 0) max_probe_bytes = (int64_t)160 << 30;
    const int max_chunks = (int)std::max<int64_t>((int64_t)n_classes * need, max_probe_bytes / CHUNK);
    const hipMemAllocationProp prop = chunk_prop(a->dev);
    // a scratch range where chunks are mapped one at a time while they are classified, + the reference chunk of class 0
    unsigned char *scratch = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> spare;      // chunks of a class that is already full (or of the third class): released at the end
    std::vector<unsigned char *> test_mapped;                // test addresses that still hold a chunk
    int rc = GNNMP_OK;
    hipError_t e = hipSuccess;
#define ARENA_HIP(expr) do { e = (expr); if (e != hipSuccess) { rc = hip_fail(e, #expr); goto done; } } while (0)
    for (int c = 0; c < n_classes; ++c) {
        a->cap[c] = (int64_t)need * CHUNK;
        ARENA_HIP(hipMemAddressReserve((void **)&a->base[c], (size_t)a->cap[c], (size_t)2 << 20, nullptr, 0));
    }
    // every chunk under test gets an address of its OWN (slot = its creation number): probing different chunks one after the other through ONE
    // address measured the first chunk every time — 80 chunks, one time (the translation of an unmapped range outlives the unmap)
    ARENA_HIP(hipMemAddressReserve((void **)&scratch, (size_t)max_chunks * (size_t)CHUNK, (size_t)2 << 20, nullptr, 0));
    rc = make_probe_plan(a, CHUNK / (PROBE_D * 4), stream);
    if (rc != GNNMP_OK) goto done;
    {
        // chunk 0 is class 0 by definition and the probe's source from now on
        hipMemGenericAllocationHandle_t h0 = nullptr;
        ARENA_HIP(hipMemCreate(&h0, (size_t)CHUNK, &prop, 0));
        ++a->created;
        a->handles[0].push_back(h0);
        ARENA_HIP(map_rw(a->base[0], (size_t)CHUNK, h0, a->dev));
        ARENA_HIP(hipMemsetAsync(a->base[0], 0, (size_t)CHUNK, stream));
        const float *ref0 = reinterpret_cast<const float *>(a->base[0]);
        const float *ref1 = nullptr;            // the first chunk of class 1, once found
        // Every further chunk is timed as the probe's OUTPUT against chunk 0.  The times fall into two clusters ~5 % apart (run-to-run
        // spread ~1 %): the slow one = chunk 0's class.  Until both clusters have shown up nothing is decided (`pending`).
        struct Pend { hipMemGenericAllocationHandle_t h; float us0; unsigned char *va; };      // va: where the chunk is mapped while under test
        std::vector<Pend> pending;
        float lo = 0.0f, hi = 0.0f, thr = 0.0f;
        auto place = [&](hipMemGenericAllocationHandle_t h, int cls) -> hipError_t {
            if (cls < a->n_classes && (int)a->handles[cls].size() < need) {
                unsigned char *va = a->base[cls] + (int64_t)a->handles[cls].size() * CHUNK;
                hipError_t e2 = map_rw(va, (size_t)CHUNK, h, a->dev);
                if (e2 != hipSuccess) return e2;
                a->handles[cls].push_back(h);
                if (cls == 1 && !ref1) {
                    e2 = hipMemsetAsync(va, 0, (size_t)CHUNK, stream);
                    ref1 = reinterpret_cast<const float *>(va);
                }
                return e2;
            }
            spare.push_back(h);      // held until the end: releasing it now would hand the same physical chunk out again
            return hipSuccess;
        };
        auto classify = [&](const Pend &q) -> int {      // rc; places the chunk
            int cls;
            if (q.us0 > thr) {
                cls = 0;
            } else if (!ref1) {
                cls = 1;
            } else {
                float us1 = 0.0f;
                int r2 = probe_us(a, ref1, reinterpret_cast<float *>(q.va), stream, &us1);
                if (r2 != GNNMP_OK) return r2;
                cls = us1 > thr ? 1 : 2;
                if (getenv("GNNMP_ARENA_DEBUG")) fprintf(stderr, "[arena]   against the class-1 reference %.1f us -> class %d\n", us1, cls);
            }
            {
                hipError_t e2 = hipMemUnmap(q.va, (size_t)CHUNK);       // leaves its test address for good
                if (e2 != hipSuccess) return hip_fail(e2, "arena: unmap of a classified chunk");
                test_mapped.erase(std::find(test_mapped.begin(), test_mapped.end(), q.va));
            }
            hipError_t e3 = place(q.h, cls);
            return e3 == hipSuccess ? GNNMP_OK : hip_fail(e3, "arena: map of a classified chunk");
        };
        auto all_full = [&]() {
            for (int c = 0; c < a->n_classes; ++c)
                if ((int)a->handles[c].size() < need) return false;
            return true;
        };
        while (!all_full()) {
            if (a->created >= max_chunks) {
                rc = fail(GNNMP_EUNSUPPORTED, "arena_create: %lld chunks of 2 GiB probed (probe %.0f..%.0f us), classes hold %zu / %zu / %zu of %d chunks "
                                              "(raise max_probe_bytes or free device memory)", (long long)a->created, lo, hi,
                          a->handles[0].size(), a->handles[1].size(), a->handles[2].size(), need);
                for (const Pend &q : pending) spare.push_back(q.h);
                goto done;
            }
            hipMemGenericAllocationHandle_t h = nullptr;
            e = hipMemCreate(&h, (size_t)CHUNK, &prop, 0);
            if (e != hipSuccess) {
                rc = hip_fail(e, "hipMemCreate(arena chunk): device memory exhausted while looking for two placement classes");
                for (const Pend &q : pending) spare.push_back(q.h);
                goto done;
            }
            ++a->created;
            Pend q = {h, 0.0f, scratch + (a->created - 1) * CHUNK};
            e = map_rw(q.va, (size_t)CHUNK, h, a->dev);
            if (e == hipSuccess) test_mapped.push_back(q.va);
            if (e == hipSuccess) rc = probe_us(a, ref0, reinterpret_cast<float *>(q.va), stream, &q.us0);
            if (e != hipSuccess && rc == GNNMP_OK) rc = hip_fail(e, "arena: map of a chunk under test");
            if (rc != GNNMP_OK) {
                spare.push_back(h);
                for (const Pend &w : pending) spare.push_back(w.h);
                goto done;
            }
            if (getenv("GNNMP_ARENA_DEBUG")) fprintf(stderr, "[arena] chunk %lld: probe against chunk 0 %.1f us\n", (long long)a->created - 1, q.us0);
            lo = lo == 0.0f ? q.us0 : std::min(lo, q.us0);
            hi = std::max(hi, q.us0);
            if (thr == 0.0f) {
                pending.push_back(q);
                if (hi > 1.04f * lo) {                    // both clusters seen: decide everything held back
                    thr = 0.5f * (lo + hi);
                    a->probe_same_us = hi;
                    a->probe_other_us = lo;
                    for (size_t i = 0; i < pending.size() && rc == GNNMP_OK; ++i) {
                        rc = classify(pending[i]);
                        if (rc != GNNMP_OK)
                            for (size_t k2 = i; k2 < pending.size(); ++k2) spare.push_back(pending[k2].h);
                    }
                    pending.clear();
                    if (rc != GNNMP_OK) goto done;
                }
            } else {
                rc = classify(q);
                if (rc != GNNMP_OK) { spare.push_back(h); goto done; }
            }
        }
        for (const Pend &q : pending) spare.push_back(q.h);
    }
    ARENA_HIP(hipStreamSynchronize(stream));
done:
#undef ARENA_HIP
    for (hipMemGenericAllocationHandle_t h : spare) { (void)hipMemRelease(h); ++a->released; }
    (void)hipDeviceSynchronize();
    if (scratch) {
        for (unsigned char *va : test_mapped) (void)hipMemUnmap(va, (size_t)CHUNK);
        (void)hipMemAddressFree(scratch, (size_t)max_chunks * (size_t)CHUNK);
    }
    if (rc != GNNMP_OK) {
        gnnmp_arena_destroy(a);
        return rc;
    }
    for (int c = 0; c < 3; ++c) a->n_chunks[c] = (int)a->handles[c].size();
    *out = a;
    return GNNMP_OK;
}

int gnnmp_arena_alloc(gnnmp_arena_t *a, int cls, int64_t bytes, void **ptr) {
    if (!a || !ptr || bytes < 0 || cls < 0 || cls >= a->n_classes) return fail(GNNMP_EINVAL, "arena_alloc: bad argument");
    std::lock_guard<std::mutex> lk(a->lock);
    const int64_t at = (a->used[cls] + 4095) & ~(int64_t)4095;
    if (at + bytes > a->cap[cls])
        return fail(GNNMP_EALLOC, "arena_alloc: class %d holds %lld of %lld bytes, %lld more do not fit", cls, (long long)a->used[cls],
                    (long long)a->cap[cls], (long long)bytes);
    *ptr = a->base[cls] + at;
    a->used[cls] = at + bytes;
    return GNNMP_OK;
}

int gnnmp_arena_reset(gnnmp_arena_t *a) {
    if (!a) return fail(GNNMP_EINVAL, "arena_reset: null arena");
    std::lock_guard<std::mutex> lk(a->lock);
    a->used[0] = a->used[1] = a->used[2] = 0;
    return GNNMP_OK;
}

int gnnmp_arena_class_of(gnnmp_arena_t *a, const void *ptr, int64_t bytes, int *cls, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || !ptr || !cls) return fail(GNNMP_EINVAL, "arena_class_of: bad argument");
    const unsigned char *p = static_cast<const unsigned char *>(ptr);
    for (int c = 0; c < a->n_classes; ++c)
        if (p >= a->base[c] && p < a->base[c] + a->cap[c]) { *cls = c; return GNNMP_OK; }
    // foreign memory: the probe with the buffer as the gathered matrix, the output in a spare corner of each arena range
    const int none = a->n_classes;                         // "in none of the ranges' classes / cannot tell"
    const int64_t n_src = std::min<int64_t>(bytes, CHUNK) / (PROBE_D * 4);
    if (n_src < PROBE_ROWS) { *cls = none; return GNNMP_OK; }      // (too small to matter)
    if ((reinterpret_cast<uintptr_t>(ptr) & 15)) { *cls = none; return GNNMP_OK; }
    std::lock_guard<std::mutex> lk(a->lock);
    int rc = make_probe_plan(a, n_src, stream);
    if (rc != GNNMP_OK) return rc;
    const int64_t out_bytes = (int64_t)PROBE_ROWS * PROBE_D * 4;
    float us[3] = {0.0f, 0.0f, 0.0f};
    for (int c = 0; c < a->n_classes; ++c) {
        if (a->cap[c] - a->used[c] < out_bytes + 4096) return fail(GNNMP_EALLOC, "arena_class_of: no room for the probe's output in range %d", c);
        float *o = reinterpret_cast<float *>(a->base[c] + a->cap[c] - out_bytes);      // the top of the range: free by the check above
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
    *cls = us[slow] > 1.04f * lo ? slow : none;
    return GNNMP_OK;
}

/* info[0] = bytes per class, [1] = bytes used in range 0, [2] = in range 1, [3] = chunks created while classifying, [4] = released again,
 * [5] = probe microseconds with source and output in one class, [6] = in two, [7] = ranges (2 | 3), [8] = bytes used in range 2 */
int gnnmp_arena_info(const gnnmp_arena_t *a, int64_t *info) {
    if (!a || !info) return fail(GNNMP_EINVAL, "arena_info: null argument");
    info[0] = a->cap[0]; info[1] = a->used[0]; info[2] = a->used[1]; info[3] = a->created; info[4] = a->released;
    info[5] = (int64_t)(a->probe_same_us + 0.5f); info[6] = (int64_t)(a->probe_other_us + 0.5f);
    info[7] = a->n_classes; info[8] = a->used[2];
    return GNNMP_OK;
}

}  // extern "C"
