// plan.hip — graph-side prep: dst-sorted CSR plan of a COO edge index, plus the bit-exact index ops
// (add_self_loops, batch).  Reference: GNNGraphs/src/gnngraph.jl:108-117 (COO container),
// GNNGraphs/src/convert.jl:221-237 (to_sparse, rebuilt per call by the reference fast path),
// GNNGraphs/src/transform.jl:12-28 (add_self_loops), :682-709 (batch).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <vector>


#include "common.h"
#include "pool.h"
#include "sort_scan.h"

namespace gnnmp {

static thread_local char g_err[512] = "";
static int g_knobs[KNOB_COUNT] = {0, -1, 0, 1, 0, 0, 0, 17, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}
int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? GNNMP_EALLOC : GNNMP_ELAUNCH;
}
int knob(int k) { return (k >= 0 && k < KNOB_COUNT) ? g_knobs[k] : 0; }
static thread_local int g_mock_device = -1;
int current_device() {
    if (g_mock_device >= 0) return g_mock_device < GNNMP_MAX_DEVICES ? g_mock_device : GNNMP_MAX_DEVICES - 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < GNNMP_MAX_DEVICES ? dev : GNNMP_MAX_DEVICES - 1;
}
int device_cus() {
    static std::atomic<int> cached[GNNMP_MAX_DEVICES] = {};
    const int dev = current_device();
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

int plan_dispose(gnnmp_graph_t *p, hipStream_t stream, bool stream_known);

int ensure_workspace(gnnmp_graph *p, size_t floats) {
    if (floats <= p->ws_floats) return GNNMP_OK;
    if (p->ws) (void)hipFree(p->ws);  // hipFree waits for work that may still read the old buffer
    p->ws = nullptr;
    p->ws_floats = 0;
    hipError_t e = hipMalloc((void **)&p->ws, sizeof(float) * floats);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(plan workspace)");
    p->ws_floats = floats;
    return GNNMP_OK;
}

bool fold_disabled_by_env() {
    static const bool off = [] {
        const char *e = getenv("GNNMP_NO_FOLD");
        return e && *e && *e != '0';
    }();
    return off;
}

int ensure_arrive(gnnmp_graph *p, size_t n, size_t floats, hipStream_t stream) {
    if (floats > p->spart_floats) {
        if (p->spart) (void)hipFree(p->spart);   // (waits for in-flight work)
        p->spart = nullptr;
        p->spart_floats = 0;
        hipError_t e = hipMalloc((void **)&p->spart, sizeof(float) * floats);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(plan slice partials)");
        p->spart_floats = floats;
    }
    if (n <= p->arrive_n) return GNNMP_OK;
    if (p->arrive) (void)hipFree(p->arrive);   // (waits for in-flight work; counters are zero between launches, nothing to carry over)
    p->arrive = nullptr;
    p->arrive_n = 0;
    hipError_t e = hipMalloc((void **)&p->arrive, sizeof(uint32_t) * n);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(plan arrival counters)");
    e = hipMemsetAsync(p->arrive, 0, sizeof(uint32_t) * n, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(plan arrival counters)");
    p->arrive_n = n;
    return GNNMP_OK;
}

int ensure_ticket(gnnmp_graph *p, hipStream_t stream) {
    if (p->ticket) return GNNMP_OK;
    hipError_t e = hipMalloc((void **)&p->ticket, 2 * sizeof(uint32_t));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(plan ticket)");
    e = hipMemsetAsync(p->ticket, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(plan ticket)");
    return GNNMP_OK;
}

__global__ void plan_degree_keys(const uint32_t *rowptr, int64_t n, uint32_t maxdeg, uint32_t *keys, uint32_t *vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = maxdeg - (rowptr[i + 1] - rowptr[i]);   // ascending key = descending length
    vals[i] = (uint32_t)i;
}

int ensure_row_order(gnnmp_graph *p, hipStream_t stream) {
    if (p->row_order || p->n_dst == 0) return GNNMP_OK;
    const size_t n = (size_t)p->n_dst;
    uint32_t *kin = nullptr, *kout = nullptr, *vin = nullptr;
    int32_t *order = nullptr;
    int rc = GNNMP_OK;
    hipError_t e = hipSuccess;
    unsigned bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) <= p->max_degree) ++bits;   // max_degree <= E' < 2^32
    if ((e = hipMalloc((void **)&kin, 4 * n)) != hipSuccess || (e = hipMalloc((void **)&kout, 4 * n)) != hipSuccess ||
        (e = hipMalloc((void **)&vin, 4 * n)) != hipSuccess || (e = hipMalloc((void **)&order, 4 * n)) != hipSuccess) {
        rc = hip_fail(e, "hipMalloc(row order)");
    } else {
        plan_degree_keys<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(p->rowptr, p->n_dst, (uint32_t)p->max_degree, kin, vin);
        rc = radix_sort_pairs_u32(kin, kout, vin, reinterpret_cast<uint32_t *>(order), n, 0, (int)bits, stream);   // syncs the stream
    }
    if (kin) (void)hipFree(kin);
    if (kout) (void)hipFree(kout);
    if (vin) (void)hipFree(vin);
    if (rc != GNNMP_OK) {
        if (order) (void)hipFree(order);
        return rc;
    }
    p->row_order = order;
    p->bytes += (int64_t)(4 * n);
    return GNNMP_OK;
}

// ---- kernels -----------------------------------------------------------------------------------

// keys[k] = 0-based destination of edge k (self loops k >= E: k - E); vals[k] = k.
// bad[0] is set if any index is outside its range.
__global__ void plan_fill_keys(const void *src, const void *dst, int idx_bytes, int base, int64_t E,
                               int64_t Etot, int64_t n_src, int64_t n_dst, uint32_t *keys,
                               uint32_t *vals, int *bad) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Etot) return;
    int64_t d;
    if (k < E) {
        d = load_index(dst, k, idx_bytes, base);
        int64_t s = load_index(src, k, idx_bytes, base);
        if (d < 0 || d >= n_dst || s < 0 || s >= n_src) {
            *bad = 1;
            d = 0;
        }
    } else {
        d = k - E;
    }
    keys[k] = (uint32_t)d;
    vals[k] = (uint32_t)k;
}

// rowptr[i] = first slot whose key >= i  (binary search in the sorted keys); rowptr[n_dst] = Etot
__global__ void plan_rowptr(const uint32_t *keys, int64_t Etot, int64_t n_dst, uint32_t *rowptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_dst) return;
    int64_t lo = 0, hi = Etot;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[i] = (uint32_t)lo;
}

// col[p] = 0-based source of the edge in slot p
__global__ void plan_col(const void *src, int idx_bytes, int base, int64_t E, int64_t Etot,
                         const uint32_t *eid, int32_t *col) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Etot) return;
    int64_t e = eid[p];
    int64_t s = e < E ? load_index(src, e, idx_bytes, base) : e - E;
    col[p] = (int32_t)s;
}

// ---- a plan straight from a compressed-sparse-column adjacency (gnnmp_plan_from_csc): no sort ---------------------------------------
// rowptr[i] = colptr[i] - base; flags[0] |= 1 if the column pointers do not start at `base`, end at E + base or decrease
__global__ void csc_rowptr_kernel(const void *colptr, int idx_bytes, int base, int64_t n_dst, int64_t E, uint32_t *rowptr, int *bad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_dst) return;
    const int64_t v = load_index(colptr, i, idx_bytes, base);
    bool ok = v >= 0 && v <= E;
    if (i == 0) ok = ok && v == 0;
    if (i == n_dst) ok = ok && v == E;
    if (i < n_dst) ok = ok && load_index(colptr, i + 1, idx_bytes, base) >= v;
    if (!ok) *bad = 1;
    rowptr[i] = (uint32_t)(ok ? v : (i == n_dst ? E : 0));
}
// col[p] = rowval[p] - base, eid[p] = p (findnz order IS the slot order); flags[0] |= 1 for a source outside the graph
__global__ void csc_col_kernel(const void *rowval, int idx_bytes, int base, int64_t E, int64_t n_src, int32_t *col, int32_t *eid, int *bad) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    int64_t s = load_index(rowval, p, idx_bytes, base);
    if (s < 0 || s >= n_src) {
        *bad = 1;
        s = 0;
    }
    col[p] = (int32_t)s;
    eid[p] = (int32_t)(uint32_t)p;
}

// collect rows longer than `thresh` as (row, beg, end) triples; *count = how many (atomic), *maxdeg = max degree
__global__ void plan_long_rows(const uint32_t *rowptr, int64_t n_dst, int thresh, int64_t *list,
                               int cap, int *count, unsigned long long *maxdeg) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst) return;
    const int64_t beg = rowptr[i], end = rowptr[i + 1];
    const int64_t len = end - beg;
    if (len > 0) atomicMax(maxdeg, (unsigned long long)len);
    if (len > thresh) {
        int pos = atomicAdd(count, 1);
        if (pos < cap) {
            list[3 * pos] = i;
            list[3 * pos + 1] = beg;
            list[3 * pos + 2] = end;
        }
    }
}
__global__ void widen_u32_kernel(const uint32_t *in, int64_t n, int64_t *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// out[p] = v[col[p]] (by == 0, node vector) or v[eid[p]] with 1.0 for plan-added self loops (by == 1, edge vector)
__global__ void slot_gather_kernel(const int32_t *idx, const float *v, int64_t Etot, int64_t n_valid,
                                   float *out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Etot) return;
    const int64_t k = (uint32_t)idx[p];   // source ids (< 2^31) or edge positions (< 2^32): unsigned either way
    out[p] = k < n_valid ? v[k] : 1.0f;
}

__global__ void self_loops_kernel(const void *src, const void *dst, int idx_bytes, int base,
                                  int64_t E, int64_t n, void *out_src, void *out_dst,
                                  const float *w, float *out_w) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E + n) return;
    if (k < E) {
        store_index(out_src, k, idx_bytes, load_index(src, k, idx_bytes, 0));
        store_index(out_dst, k, idx_bytes, load_index(dst, k, idx_bytes, 0));
        if (out_w) out_w[k] = w[k];
    } else {
        int64_t v = (k - E) + base;
        store_index(out_src, k, idx_bytes, v);
        store_index(out_dst, k, idx_bytes, v);
        if (out_w) out_w[k] = 1.0f;
    }
}

// one thread per edge: find its graph by binary search in edge_ptr, add the node offset.
__global__ void batch_edges_kernel(const void *src, const void *dst, int idx_bytes,
                                   const int64_t *edge_ptr, const int64_t *node_ptr, int64_t G,
                                   int64_t Etot, void *out_src, void *out_dst) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Etot) return;
    int64_t lo = 0, hi = G;  // largest g with edge_ptr[g] <= k
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (edge_ptr[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    int64_t off = node_ptr[lo];
    store_index(out_src, k, idx_bytes, load_index(src, k, idx_bytes, 0) + off);
    store_index(out_dst, k, idx_bytes, load_index(dst, k, idx_bytes, 0) + off);
}
__global__ void batch_indicator_kernel(const int64_t *node_ptr, int64_t G, int64_t Ntot,
                                       int idx_bytes, int base, void *gi) {
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= Ntot) return;
    int64_t lo = 0, hi = G;
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (node_ptr[mid] <= v)
            lo = mid;
        else
            hi = mid;
    }
    store_index(gi, v, idx_bytes, lo + base);
}

static inline unsigned nblocks(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// Long-row threshold: a row is reduced sequentially by one lane group, ~0.25 us per edge once HBM latency is the
// limit, while the whole kernel moves an edge every ~0.1 ns; rows above ~4e-5 * E' edges would become the tail.
// Clamped to [GNNMP_MIN_LONG_ROW, GNNMP_LONG_ROW]; rows up to this length keep the exact reference edge order.
int plan_long_thresh(int64_t Etot) {
    if (knob(KNOB_LONG_ROW) > 0) return knob(KNOB_LONG_ROW);
    int th = GNNMP_MIN_LONG_ROW;
    while (th < GNNMP_LONG_ROW && (double)th < 4e-5 * (double)Etot) th <<= 1;
    return th;
}

// slots per chunk of a split row (<= the plan's long_thresh); GNNMP_CHUNK_SLOTS in the environment overrides it per plan build (A/B runs)
static int plan_chunk_slots() {
    if (const char *e = getenv("GNNMP_CHUNK_SLOTS")) {
        const int v = atoi(e);
        if (v >= 16) return v;
    }
    return 128;
}

int plan_build_long_rows(gnnmp_graph *p, hipStream_t stream) {
    const int64_t n_dst = p->n_dst, Etot = p->n_total;
    const int BS = 256;
    int *flags = nullptr;       // [0] long count; [2..3] = one unsigned long long: max degree
    int64_t *long_tmp = nullptr;
    int rc = GNNMP_OK;
#define PLAN_HIP(expr)                                  \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #expr);                  \
            goto done;                                  \
        }                                               \
    } while (0)
    {
        // long rows: at most Etot / thresh of them
        int cap = (int)std::min<int64_t>(Etot / std::max(1, p->long_thresh) + 1, n_dst + 1);
        PLAN_HIP(hipMalloc((void **)&flags, sizeof(int) * 4));
        PLAN_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, stream));
        PLAN_HIP(hipMalloc((void **)&long_tmp, sizeof(int64_t) * 3 * (size_t)std::max(cap, 1)));
        if (n_dst > 0) {
            plan_long_rows<<<nblocks(n_dst, BS), BS, 0, stream>>>(p->rowptr, n_dst, p->long_thresh, long_tmp, cap, flags,
                                                                  reinterpret_cast<unsigned long long *>(flags + 2));
            PLAN_HIP(hipGetLastError());
        }
        int meta[4] = {0, 0, 0, 0};
        PLAN_HIP(hipMemcpyAsync(meta, flags, sizeof(int) * 4, hipMemcpyDeviceToHost, stream));
        PLAN_HIP(hipStreamSynchronize(stream));
        {
            unsigned long long md;
            memcpy(&md, meta + 2, sizeof(md));
            p->max_degree = (int64_t)md;
        }
        p->n_long = std::min(meta[0], cap);
        if (p->n_long > 0) {
            struct Tri { int64_t row, beg, end; };
            std::vector<Tri> h((size_t)p->n_long);
            PLAN_HIP(hipMemcpy(h.data(), long_tmp, sizeof(Tri) * h.size(), hipMemcpyDeviceToHost));
            // atomics filled the list in arbitrary order: make it canonical (ascending row)
            std::sort(h.begin(), h.end(), [](const Tri &a, const Tri &b) { return a.row < b.row; });
            std::vector<int32_t> rows, cptr, crow, clrow;
            std::vector<uint32_t> cbeg, cend;
            cptr.push_back(0);
            // Chunk length (round 6): rows are split ABOVE long_thresh (that is what bounds bit-exactness and keeps a sequential row from
            // being the kernel's tail), but the chunks of a row that is split anyway are at most plan_chunk_slots() = 128 slots long: a chunk
            // is a latency-bound chain of len / 8 row-load batches, and the fused layer kernel's pre-pass over the split rows — nothing
            // else runs beside it — took 110 us of a 4.8 ms GCN layer with 512-slot chunks on the products shape.
            const int64_t clen = std::min<int64_t>(p->long_thresh, plan_chunk_slots());
            for (const Tri &t : h) {
                const int64_t len = t.end - t.beg;
                const int64_t nch = (len + clen - 1) / clen;
                const int64_t csz = (len + nch - 1) / nch;  // balanced chunks, each <= clen <= long_thresh slots
                for (int64_t c = 0; c < nch; ++c) {
                    crow.push_back((int32_t)t.row);
                    clrow.push_back((int32_t)rows.size());
                    cbeg.push_back((uint32_t)(t.beg + c * csz));
                    cend.push_back((uint32_t)std::min(t.beg + (c + 1) * csz, t.end));
                }
                rows.push_back((int32_t)t.row);
                cptr.push_back((int32_t)crow.size());
            }
            if (crow.size() >= (size_t)INT32_MAX) {
                rc = fail(GNNMP_EUNSUPPORTED, "plan_create: %zu chunks of split rows exceed the int32 chunk index", crow.size());
                goto done;
            }
            p->n_chunks = (int)crow.size();
            auto upload = [&](int32_t **dst, const std::vector<int32_t> &v) -> hipError_t {
                hipError_t e = hipMalloc((void **)dst, sizeof(int32_t) * v.size());
                if (e != hipSuccess) return e;
                p->bytes += (int64_t)(sizeof(int32_t) * v.size());
                return hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice);
            };
            PLAN_HIP(upload(&p->long_rows, rows));
            PLAN_HIP(upload(&p->long_cptr, cptr));
            PLAN_HIP(upload(&p->chunk_row, crow));
            PLAN_HIP(upload(&p->chunk_lrow, clrow));
            auto upload_u = [&](uint32_t **dst, const std::vector<uint32_t> &v) -> hipError_t {
                hipError_t e = hipMalloc((void **)dst, sizeof(uint32_t) * v.size());
                if (e != hipSuccess) return e;
                p->bytes += (int64_t)(sizeof(uint32_t) * v.size());
                return hipMemcpy(*dst, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice);
            };
            PLAN_HIP(upload_u(&p->chunk_beg, cbeg));
            PLAN_HIP(upload_u(&p->chunk_end, cend));
        }
    }
done:
    if (flags) (void)hipFree(flags);
    if (long_tmp) (void)hipFree(long_tmp);
#undef PLAN_HIP
    return rc;
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_version(void) { return GNNMP_VERSION; }
const char *gnnmp_last_error(void) { return g_err; }

// perf-experiment hook, see common.h Knob (not part of the drop-in surface)
int gnnmp_tune(int k, int value) {
    if (k < 0 || k >= KNOB_COUNT) return fail(GNNMP_EINVAL, "gnnmp_tune: bad knob %d", k);
    g_knobs[k] = value;
    return GNNMP_OK;
}

// test hooks (like gnnmp_tune: exported, not part of the drop-in surface).  gnnmp_debug_mock_device(d >= 0) makes `d` the calling
// thread's "current device" for every per-device table of the library (common.h: current_device), d < 0 restores hipGetDevice.
int gnnmp_debug_mock_device(int dev) {
    g_mock_device = dev;
    return current_device();
}
// Runs the per-device machinery with a counting stand-in for hipFuncSetAttribute and returns how often it ran: the sequence of mocked
// devices devs[0..n) must run it once per DISTINCT device (tests/test_multi_device_cpu.py).  fail_on >= 0: the stand-in fails on that
// device; *n_failed = calls that reported the failure (every call on that device must, not only the first).
int gnnmp_debug_device_once(const int *devs, int n, int fail_on, int *n_failed) {
    DeviceOnce once;
    int ran = 0, failed = 0;
    const int keep = g_mock_device;
    for (int k = 0; k < n; ++k) {
        g_mock_device = devs[k];
        const hipError_t e = device_once(once, [&] {
            ++ran;
            return current_device() == fail_on ? hipErrorInvalidValue : hipSuccess;
        });
        if (e != hipSuccess) ++failed;
    }
    g_mock_device = keep;
    if (n_failed) *n_failed = failed;
    return ran;
}
// the pooled block of a plan made by gnnmp_plan_concat / gnnmp_plan_select (NULL for other plans): lets a test see WHICH block a plan got
void *gnnmp_debug_plan_block(const gnnmp_graph_t *p) { return p ? p->block : nullptr; }
// pool.h's slot choice on host arrays
int gnnmp_debug_pool_pick(const uint64_t *caps, int n, uint64_t bytes) {
    size_t c[64];
    if (n < 0 || n > 64) return -2;
    for (int i = 0; i < n; ++i) c[i] = (size_t)caps[i];
    return pool_pick(c, n, (size_t)bytes);
}

int gnnmp_plan_destroy(gnnmp_graph_t *p) { return plan_dispose(p, nullptr, false); }

/* the arrival counters of the in-kernel fold and the fused layer kernel's ticket back to zero, stream-ordered (gnnmp.h: repair hook) */
int gnnmp_plan_reset_counters(gnnmp_graph_t *p, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_reset_counters: null plan");
    if (p->arrive && p->arrive_n) GNNMP_HIP(hipMemsetAsync(p->arrive, 0, sizeof(uint32_t) * p->arrive_n, stream));
    if (p->ticket) GNNMP_HIP(hipMemsetAsync(p->ticket, 0, 2 * sizeof(uint32_t), stream));
    return GNNMP_OK;
}

/* stream-ordered destroy of a pooled plan (gnnmp_plan_concat / gnnmp_plan_select): see gnnmp.h */
int gnnmp_plan_release(gnnmp_graph_t *p, gnnmp_stream_t stream) { return plan_dispose(p, (hipStream_t)stream, true); }

}  // extern "C"

namespace gnnmp {
int plan_dispose(gnnmp_graph_t *p, hipStream_t stream, bool stream_known) {
    if (!p) return GNNMP_OK;
    if (p->block) {
        // rowptr / col / eid are pieces of the pooled block: back to the pool (with an event on `stream` when the caller names it)
        pool_park(p->block, p->block_bytes, stream, stream_known);
        p->rowptr = nullptr;
        p->col = nullptr;
        p->eid = nullptr;
    }
    if (p->rowptr) (void)hipFree(p->rowptr);
    if (p->col) (void)hipFree(p->col);
    if (p->eid) (void)hipFree(p->eid);
    if (p->long_rows) (void)hipFree(p->long_rows);
    if (p->long_cptr) (void)hipFree(p->long_cptr);
    if (p->chunk_row) (void)hipFree(p->chunk_row);
    if (p->chunk_lrow) (void)hipFree(p->chunk_lrow);
    if (p->arrive) (void)hipFree(p->arrive);
    if (p->spart) (void)hipFree(p->spart);
    if (p->chunk_beg) (void)hipFree(p->chunk_beg);
    if (p->chunk_end) (void)hipFree(p->chunk_end);
    if (p->ws) (void)hipFree(p->ws);
    if (p->ticket) (void)hipFree(p->ticket);
    if (p->row_order) (void)hipFree(p->row_order);
    delete p;
    return GNNMP_OK;
}
}  // namespace gnnmp

extern "C" {

int gnnmp_plan_create(gnnmp_graph_t **out, const void *src, const void *dst, int idx_bytes,
                      int index_base, int64_t n_src, int64_t n_dst, int64_t n_edges,
                      int add_self_loops, int validate, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return fail(GNNMP_EINVAL, "plan_create: out is NULL");
    *out = nullptr;
    if (idx_bytes != 4 && idx_bytes != 8)
        return fail(GNNMP_EINVAL, "plan_create: idx_bytes must be 4 or 8 (got %d)", idx_bytes);
    if (index_base != 0 && index_base != 1)
        return fail(GNNMP_EINVAL, "plan_create: index_base must be 0 or 1 (got %d)", index_base);
    if (n_src < 0 || n_dst < 0 || n_edges < 0)
        return fail(GNNMP_EINVAL, "plan_create: negative size");
    if (n_edges > 0 && (!src || !dst)) return fail(GNNMP_EINVAL, "plan_create: null edge index");
    if (add_self_loops && n_src != n_dst)
        return fail(GNNMP_EINVAL, "plan_create: add_self_loops needs n_src == n_dst (%lld vs %lld)",
                    (long long)n_src, (long long)n_dst);
    const int64_t Etot = n_edges + (add_self_loops ? n_dst : 0);
    // slots and edge positions are UNSIGNED 32-bit, node ids signed 32-bit; 65 536 slots of headroom so that no slot counter of
    // a kernel (base += G, p0 += 4, off += cap ...) can wrap
    if (Etot >= (int64_t)GNNMP_MAX_SLOTS || n_src >= (int64_t)INT32_MAX || n_dst >= (int64_t)INT32_MAX)
        return fail(GNNMP_EUNSUPPORTED, "plan_create: E' = %lld (limit 2^32 - 65536) / N = %lld (limit 2^31 - 2) exceed the plan format",
                    (long long)Etot, (long long)std::max(n_src, n_dst));

    gnnmp_graph_t *p = new gnnmp_graph_t();
    p->n_src = n_src;
    p->n_dst = n_dst;
    p->n_edges = n_edges;
    p->n_total = Etot;
    p->self_loops = add_self_loops ? 1 : 0;
    p->long_thresh = plan_long_thresh(Etot);

    uint32_t *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr;
    int *flags = nullptr;  // [0] bad index
    int rc = GNNMP_OK;
    const int BS = 256;
    const size_t epad = (size_t)std::max<int64_t>(Etot, 1);

#define PLAN_HIP(expr)                                  \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #expr);                  \
            goto done;                                  \
        }                                               \
    } while (0)

    PLAN_HIP(hipMalloc((void **)&p->rowptr, sizeof(uint32_t) * (size_t)(n_dst + 1)));
    PLAN_HIP(hipMalloc((void **)&p->col, sizeof(int32_t) * epad));
    PLAN_HIP(hipMalloc((void **)&p->eid, sizeof(int32_t) * epad));
    PLAN_HIP(hipMalloc((void **)&flags, sizeof(int) * 4));
    PLAN_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, stream));
    p->bytes = (int64_t)(sizeof(int32_t) * ((size_t)(n_dst + 1) + 2 * epad));

    if (Etot > 0) {
        // one allocation for the three key / value arrays of the sort
        const size_t piece = (sizeof(uint32_t) * epad + 255) & ~(size_t)255;
        PLAN_HIP(hipMalloc((void **)&keys_in, 3 * piece));
        keys_out = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(keys_in) + piece);
        vals_in = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(keys_in) + 2 * piece);
        plan_fill_keys<<<nblocks(Etot, BS), BS, 0, stream>>>(src, dst, idx_bytes, index_base,
                                                             n_edges, Etot, n_src, n_dst, keys_in,
                                                             vals_in, flags);
        PLAN_HIP(hipGetLastError());
        if (validate) {
            int bad = 0;
            PLAN_HIP(hipMemcpyAsync(&bad, flags, sizeof(int), hipMemcpyDeviceToHost, stream));
            PLAN_HIP(hipStreamSynchronize(stream));
            if (bad) {
                rc = fail(GNNMP_EBOUNDS, "plan_create: edge index outside 1..N (n_src=%lld n_dst=%lld)",
                          (long long)n_src, (long long)n_dst);
                goto done;
            }
        }
        // stable LSD radix sort of (dst, edge position) on the bits that can be set
        unsigned bits = 1;
        while (bits < 32 && ((int64_t)1 << bits) < n_dst) ++bits;
        rc = radix_sort_pairs_u32(keys_in, keys_out, vals_in, reinterpret_cast<uint32_t *>(p->eid), (size_t)Etot, 0, (int)bits,
                                  stream);
        if (rc != GNNMP_OK) goto done;
        plan_col<<<nblocks(Etot, BS), BS, 0, stream>>>(src, idx_bytes, index_base, n_edges, Etot,
                                                       reinterpret_cast<uint32_t *>(p->eid), p->col);
        PLAN_HIP(hipGetLastError());
    }
    plan_rowptr<<<nblocks(n_dst + 1, BS), BS, 0, stream>>>(keys_out, Etot, n_dst, p->rowptr);
    PLAN_HIP(hipGetLastError());

    rc = plan_build_long_rows(p, stream);

done:
    if (keys_in) (void)hipFree(keys_in);   // keys_out and vals_in live in the same allocation
    if (flags) (void)hipFree(flags);
#undef PLAN_HIP
    if (rc != GNNMP_OK) {
        gnnmp_plan_destroy(p);
        return rc;
    }
    *out = p;
    return GNNMP_OK;
}

int gnnmp_plan_from_csc(gnnmp_graph_t **out, const void *colptr, const void *rowval, int idx_bytes, int index_base, int64_t n_src,
                        int64_t n_dst, int64_t n_edges, int validate, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return fail(GNNMP_EINVAL, "plan_from_csc: out is NULL");
    *out = nullptr;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "plan_from_csc: idx_bytes must be 4 or 8 (got %d)", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "plan_from_csc: index_base must be 0 or 1 (got %d)", index_base);
    if (n_src < 0 || n_dst < 0 || n_edges < 0) return fail(GNNMP_EINVAL, "plan_from_csc: negative size");
    if (!colptr || (n_edges > 0 && !rowval)) return fail(GNNMP_EINVAL, "plan_from_csc: null colptr / rowval");
    if (n_edges >= (int64_t)GNNMP_MAX_SLOTS || n_src >= (int64_t)INT32_MAX || n_dst >= (int64_t)INT32_MAX)
        return fail(GNNMP_EUNSUPPORTED, "plan_from_csc: E = %lld (limit 2^32 - 65536) / N = %lld (limit 2^31 - 2) exceed the plan format",
                    (long long)n_edges, (long long)std::max(n_src, n_dst));
    gnnmp_graph_t *p = new gnnmp_graph_t();
    p->n_src = n_src;
    p->n_dst = n_dst;
    p->n_edges = n_edges;
    p->n_total = n_edges;
    p->long_thresh = plan_long_thresh(n_edges);
    int *flags = nullptr;
    int rc = GNNMP_OK;
    const int BS = 256;
    const size_t epad = (size_t)std::max<int64_t>(n_edges, 1);
#define PLAN_HIP(expr)                                  \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #expr);                  \
            goto done;                                  \
        }                                               \
    } while (0)
    PLAN_HIP(hipMalloc((void **)&p->rowptr, sizeof(uint32_t) * (size_t)(n_dst + 1)));
    PLAN_HIP(hipMalloc((void **)&p->col, sizeof(int32_t) * epad));
    PLAN_HIP(hipMalloc((void **)&p->eid, sizeof(int32_t) * epad));
    PLAN_HIP(hipMalloc((void **)&flags, sizeof(int) * 4));
    PLAN_HIP(hipMemsetAsync(flags, 0, sizeof(int) * 4, stream));
    p->bytes = (int64_t)(sizeof(int32_t) * ((size_t)(n_dst + 1) + 2 * epad));
    csc_rowptr_kernel<<<nblocks(n_dst + 1, BS), BS, 0, stream>>>(colptr, idx_bytes, index_base, n_dst, n_edges, p->rowptr, flags);
    PLAN_HIP(hipGetLastError());
    if (n_edges > 0) {
        csc_col_kernel<<<nblocks(n_edges, BS), BS, 0, stream>>>(rowval, idx_bytes, index_base, n_edges, n_src, p->col, p->eid, flags);
        PLAN_HIP(hipGetLastError());
    }
    (void)validate;
    {   // ALWAYS checked: the build synchronises below anyway (plan_build_long_rows), so skipping the read-back would save nothing, and a
        // column pointer sanitised entry by entry can leave a row with beg > end that no kernel would ever write
        int bad = 0;
        PLAN_HIP(hipMemcpyAsync(&bad, flags, sizeof(int), hipMemcpyDeviceToHost, stream));
        PLAN_HIP(hipStreamSynchronize(stream));
        if (bad) {
            rc = fail(GNNMP_EBOUNDS, "plan_from_csc: colptr is not a monotone 0..E column pointer, or a row index lies outside 1..n_src "
                      "(n_src=%lld n_dst=%lld E=%lld)", (long long)n_src, (long long)n_dst, (long long)n_edges);
            goto done;
        }
    }
    rc = plan_build_long_rows(p, stream);
done:
    if (flags) (void)hipFree(flags);
#undef PLAN_HIP
    if (rc != GNNMP_OK) {
        gnnmp_plan_destroy(p);
        return rc;
    }
    *out = p;
    return GNNMP_OK;
}

int gnnmp_plan_slot_gather_f32(gnnmp_graph_t *p, int by, const float *v, float *out_slot,
                               gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_slot_gather: null plan");
    if (by != 0 && by != 1) return fail(GNNMP_EINVAL, "plan_slot_gather: `by` must be 0 (node) or 1 (edge)");
    if (p->n_total == 0) return GNNMP_OK;
    if (!v || !out_slot) return fail(GNNMP_EINVAL, "plan_slot_gather: null pointer");
    if (by == 0)
        slot_gather_kernel<<<nblocks(p->n_total, 256), 256, 0, stream>>>(p->col, v, p->n_total, p->n_src, out_slot);
    else
        slot_gather_kernel<<<nblocks(p->n_total, 256), 256, 0, stream>>>(p->eid, v, p->n_total, p->n_edges, out_slot);
    GNNMP_LAUNCH_CHECK("slot_gather_kernel");
    return GNNMP_OK;
}

int gnnmp_plan_info(const gnnmp_graph_t *p, int64_t info[8]) {
    if (!p || !info) return fail(GNNMP_EINVAL, "plan_info: null argument");
    info[0] = p->n_src;
    info[1] = p->n_dst;
    info[2] = p->n_edges;
    info[3] = p->n_total;
    info[4] = p->max_degree;
    info[5] = p->n_long;  // rows that are split
    info[6] = p->bytes;
    info[7] = p->long_thresh;
    return GNNMP_OK;
}

int gnnmp_plan_export(const gnnmp_graph_t *p, int32_t *rowptr, int32_t *col, int32_t *eid,
                      gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_export: null plan");
    if (p->n_total > (int64_t)INT32_MAX)
        return fail(GNNMP_EUNSUPPORTED, "plan_export: E' = %lld does not fit int32 rowptr / eid (use gnnmp_plan_export64)",
                    (long long)p->n_total);
    if (rowptr)   // fewer than 2^31 slots: the unsigned values ARE the int32 values
        GNNMP_HIP(hipMemcpyAsync(rowptr, p->rowptr, sizeof(int32_t) * (size_t)(p->n_dst + 1), hipMemcpyDeviceToDevice, stream));
    if (col && p->n_total > 0)
        GNNMP_HIP(hipMemcpyAsync(col, p->col, sizeof(int32_t) * (size_t)p->n_total,
                                 hipMemcpyDeviceToDevice, stream));
    if (eid && p->n_total > 0)
        GNNMP_HIP(hipMemcpyAsync(eid, p->eid, sizeof(int32_t) * (size_t)p->n_total,
                                 hipMemcpyDeviceToDevice, stream));
    return GNNMP_OK;
}

int gnnmp_plan_export64(const gnnmp_graph_t *p, int64_t *rowptr, int32_t *col, uint32_t *eid, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_export64: null plan");
    if (rowptr) {
        widen_u32_kernel<<<nblocks(p->n_dst + 1, 256), 256, 0, stream>>>(p->rowptr, p->n_dst + 1, rowptr);
        GNNMP_LAUNCH_CHECK("widen_u32_kernel");
    }
    if (col && p->n_total > 0)
        GNNMP_HIP(hipMemcpyAsync(col, p->col, sizeof(int32_t) * (size_t)p->n_total, hipMemcpyDeviceToDevice, stream));
    if (eid && p->n_total > 0)
        GNNMP_HIP(hipMemcpyAsync(eid, p->eid, sizeof(uint32_t) * (size_t)p->n_total, hipMemcpyDeviceToDevice, stream));
    return GNNMP_OK;
}

int gnnmp_add_self_loops(const void *src, const void *dst, int idx_bytes, int index_base,
                         int64_t n_edges, int64_t n, void *out_src, void *out_dst, const float *w,
                         float *out_w, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "add_self_loops: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "add_self_loops: index_base %d", index_base);
    if (n_edges < 0 || n < 0) return fail(GNNMP_EINVAL, "add_self_loops: negative size");
    if ((w != nullptr && out_w == nullptr) || (w == nullptr && out_w != nullptr && n_edges > 0))
        return fail(GNNMP_EINVAL, "add_self_loops: w and out_w must both be NULL or both non-NULL");
    if (n_edges + n == 0) return GNNMP_OK;
    if (!out_src || !out_dst || (n_edges > 0 && (!src || !dst)))
        return fail(GNNMP_EINVAL, "add_self_loops: null pointer");
    self_loops_kernel<<<nblocks(n_edges + n, 256), 256, 0, stream>>>(
        src, dst, idx_bytes, index_base, n_edges, n, out_src, out_dst, w, out_w);
    GNNMP_LAUNCH_CHECK("self_loops_kernel");
    return GNNMP_OK;
}

int gnnmp_batch_coo(const void *src, const void *dst, int idx_bytes, int index_base,
                    const int64_t *edge_ptr, const int64_t *node_ptr, int64_t n_graphs,
                    void *out_src, void *out_dst, void *graph_indicator, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "batch_coo: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "batch_coo: index_base %d", index_base);
    if (n_graphs <= 0 || !edge_ptr || !node_ptr) return fail(GNNMP_EINVAL, "batch_coo: bad graph table");
    // totals live on the device; read them back (graph prep, not on the timed path)
    int64_t tot[2] = {0, 0};
    GNNMP_HIP(hipMemcpyAsync(&tot[0], edge_ptr + n_graphs, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    GNNMP_HIP(hipMemcpyAsync(&tot[1], node_ptr + n_graphs, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    GNNMP_HIP(hipStreamSynchronize(stream));
    if (tot[0] > 0) {
        if (!src || !dst || !out_src || !out_dst) return fail(GNNMP_EINVAL, "batch_coo: null edge arrays");
        batch_edges_kernel<<<nblocks(tot[0], 256), 256, 0, stream>>>(
            src, dst, idx_bytes, edge_ptr, node_ptr, n_graphs, tot[0], out_src, out_dst);
        GNNMP_LAUNCH_CHECK("batch_edges_kernel");
    }
    if (tot[1] > 0 && graph_indicator) {
        batch_indicator_kernel<<<nblocks(tot[1], 256), 256, 0, stream>>>(
            node_ptr, n_graphs, tot[1], idx_bytes, index_base, graph_indicator);
        GNNMP_LAUNCH_CHECK("batch_indicator_kernel");
    }
    return GNNMP_OK;
}

}  // extern "C"
