// graphprep.hip — device-side graph preparation on either side of the path.  Integer work,
// bit-exact by construction:
//   sort_edge_index(u, v)       GNNGraphs/src/utils.jl:30-45     lexicographic sort of the (u_k, v_k) pairs
//   is_bidirected(g)            GNNGraphs/src/query.jl:553-558   sort_edge_index(s, t) == sort_edge_index(t, s)
//   has_self_loops(g)           GNNGraphs/src/query.jl:565-569   any(s .== t)
//   sample_neighbors(g, nodes, K; dir, replace)   GNNGraphs/src/sampling.jl:68-119 (the edge-id selection; the reference
//                               builds adjacency lists on the host and calls StatsBase.sample per node)
// The CUDA extension of the reference round-trips sort_edge_index through the CPU (GNNGraphsCUDAExt.jl:24-30).
#include <cstring>
#include <algorithm>


#include "common.h"
#include "sort_scan.h"

namespace gnnmp {

static inline unsigned nb(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// key = (u << 32) | v, both 0-based; bad[0] set if an index does not fit 32 bits or is negative
__global__ void pack_pairs_kernel(const void *u, const void *v, int idx_bytes, int base, int64_t E, uint64_t *keys,
                                  int *bad) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const int64_t a = load_index(u, k, idx_bytes, base), b = load_index(v, k, idx_bytes, base);
    if (a < 0 || b < 0 || a > 0xffffffffLL || b > 0xffffffffLL) {
        *bad = 1;
        keys[k] = 0;
        return;
    }
    keys[k] = ((uint64_t)a << 32) | (uint64_t)b;
}
__global__ void unpack_pairs_kernel(const uint64_t *keys, int idx_bytes, int base, int64_t E, void *u, void *v) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    store_index(u, k, idx_bytes, (int64_t)(keys[k] >> 32) + base);
    store_index(v, k, idx_bytes, (int64_t)(keys[k] & 0xffffffffULL) + base);
}
__global__ void keys_differ_kernel(const uint64_t *a, const uint64_t *b, int64_t E, int *flag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E && a[k] != b[k]) *flag = 1;
}
__global__ void self_loop_kernel(const void *u, const void *v, int idx_bytes, int64_t E, int *flag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < E && load_index(u, k, idx_bytes, 0) == load_index(v, k, idx_bytes, 0)) *flag = 1;
}

#define PREP_HIP(expr)                                  \
    do {                                                \
        hipError_t e__ = (expr);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #expr);                  \
            goto done;                                  \
        }                                               \
    } while (0)
#define PREP_G(expr)                                    \
    do {                                                \
        rc = (expr);                                    \
        if (rc != GNNMP_OK) goto done;                  \
    } while (0)

// sorted keys of the pairs (first, second) into keys_out (device, E entries); scratch freed before returning
static int sorted_pair_keys(const void *first, const void *second, int idx_bytes, int base, int64_t E, uint64_t *keys_out,
                            hipStream_t stream) {
    int rc = GNNMP_OK;
    uint64_t *keys_in = nullptr;
    int *flag = nullptr;
    int hflag = 0;
    PREP_HIP(hipMalloc((void **)&keys_in, sizeof(uint64_t) * (size_t)std::max<int64_t>(E, 1)));
    PREP_HIP(hipMalloc((void **)&flag, sizeof(int)));
    PREP_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream));
    pack_pairs_kernel<<<nb(E), 256, 0, stream>>>(first, second, idx_bytes, base, E, keys_in, flag);
    PREP_HIP(hipGetLastError());
    PREP_G(radix_sort_keys_u64(keys_in, keys_out, (size_t)E, 0, 64, stream));
    PREP_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipStreamSynchronize(stream));
    if (hflag) rc = fail(GNNMP_EBOUNDS, "sort_edge_index: an index is negative or does not fit 32 bits");
done:
    if (keys_in) (void)hipFree(keys_in);
    if (flag) (void)hipFree(flag);
    return rc;
}

// Scratch for the per-mini-batch entry points (sample_neighbors, unique_append, induced_subgraph): hipMalloc / hipFree cost
// ~0.1-1 ms each and a NeighborLoader batch made ~50 of them.  Freed blocks are parked in a small per-thread cache and
// handed out again (every entry point synchronises its stream before returning, so a parked block is idle).
// A block is only handed out on the device it was allocated on (dev = common.h's current_device at allocation; alloc and free of one
// entry point run under the same current device).
struct PrepBlock { void *p; size_t cap; int dev; };
static thread_local PrepBlock g_prep_cache[8] = {};
static hipError_t prep_alloc(void **out, size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    const int dev = current_device();
    int best = -1;
    for (int i = 0; i < 8; ++i)
        if (g_prep_cache[i].p && g_prep_cache[i].dev == dev && g_prep_cache[i].cap >= bytes &&
            (best < 0 || g_prep_cache[i].cap < g_prep_cache[best].cap))
            best = i;
    if (best >= 0 && g_prep_cache[best].cap <= 4 * bytes + (1 << 20)) {
        *out = g_prep_cache[best].p;
        g_prep_cache[best] = PrepBlock{nullptr, 0, 0};
        return hipSuccess;
    }
    return hipMalloc(out, bytes);
}
static void prep_free(void *p, size_t bytes) {
    if (!p) return;
    bytes = std::max<size_t>(bytes, 256);
    const int dev = current_device();
    int slot = -1;
    for (int i = 0; i < 8; ++i)
        if (!g_prep_cache[i].p) { slot = i; break; }
    if (slot < 0) {   // cache full: evict the smallest block
        slot = 0;
        for (int i = 1; i < 8; ++i)
            if (g_prep_cache[i].cap < g_prep_cache[slot].cap) slot = i;
        if (g_prep_cache[slot].cap >= bytes) {
            (void)hipFree(p);
            return;
        }
        (void)hipFree(g_prep_cache[slot].p);
    }
    g_prep_cache[slot] = PrepBlock{p, bytes, dev};
}

// ---- neighbour sampling ---------------------------------------------------------------------------
// counter-based generator: splitmix64 of (seed, seed node, draw index) -> 53 uniform bits
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01(uint64_t seed, uint64_t node, uint64_t draw) {
    const uint64_t r = mix64(mix64(seed ^ (node * 0xd1342543de82ef95ULL)) + draw);
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void sample_counts_kernel(const uint32_t *rowptr, const void *nodes, int idx_bytes, int base, int64_t M,
                                     int64_t n_rows, int64_t K, int replace, int64_t *counts, int *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t v = load_index(nodes, i, idx_bytes, base);
    if (v < 0 || v >= n_rows) {
        *bad = 1;
        counts[i] = 0;
        return;
    }
    const int64_t d = (int64_t)(rowptr[v + 1] - rowptr[v]);
    // sampling.jl:73-78: replace ? (K > 0 ? K : d) : (K > 0 ? min(d, K) : d); nothing can be drawn from an empty list
    // replace = 2: picks with replacement but only min(d, K) of them — NeighborLoader's rand(neighbors, min(K, d)), samplers.jl:60-61
    counts[i] = d == 0 ? 0 : (K > 0 ? (replace == 1 ? K : (d < K ? d : K)) : d);
}

// one thread per seed node.  Without replacement: Knuth's selection sampling (Algorithm S) over the row — every k-subset
// equally likely, chosen edges stay in original edge order.  With replacement: k independent uniform picks.
__global__ void sample_fill_kernel(const uint32_t *rowptr, const int32_t *eid, const void *nodes, int idx_bytes, int base,
                                   int64_t M, int64_t K, int replace, uint64_t seed, const int64_t *offsets,
                                   void *eids_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t v = load_index(nodes, i, idx_bytes, base);
    const int64_t beg = rowptr[v], d = (int64_t)rowptr[v + 1] - beg;
    const int64_t o = offsets[i], k = offsets[i + 1] - o;
    if (k == 0) return;
    if (replace) {
        for (int64_t j = 0; j < k; ++j) {
            const int64_t pick = min(d - 1, (int64_t)(uniform01(seed, (uint64_t)i, (uint64_t)j) * (double)d));
            store_index(eids_out, o + j, idx_bytes, (int64_t)(uint32_t)eid[beg + pick] + base);
        }
        return;
    }
    if (d > 2 * k * k) {
        // Hub seed, few picks: Floyd's algorithm — a uniformly random k-subset of the d slots in k draws (not d): for
        // j = d-k .. d-1 draw r in [0, j]; take r unless it is already chosen, then take j (never chosen before).  The
        // chosen slots are kept sorted in the output (insertion, O(k) each), so the result is in original edge order like
        // the selection sampling below; O(k^2) instead of O(d) keeps a 17 000-edge seed from being the kernel.
        for (int64_t c = 0; c < k; ++c) {
            const int64_t j = d - k + c;
            const int64_t r = min(j, (int64_t)(uniform01(seed, (uint64_t)i, (uint64_t)c) * (double)(j + 1)));
            int64_t lo = 0, hi = c;                    // first position with slot >= r among the c sorted picks
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (load_index(eids_out, o + mid, idx_bytes, 0) < r) lo = mid + 1; else hi = mid;
            }
            const bool taken = lo < c && load_index(eids_out, o + lo, idx_bytes, 0) == r;
            if (taken) {
                store_index(eids_out, o + c, idx_bytes, j);        // j exceeds every earlier pick: append
            } else {
                for (int64_t q = c; q > lo; --q)
                    store_index(eids_out, o + q, idx_bytes, load_index(eids_out, o + q - 1, idx_bytes, 0));
                store_index(eids_out, o + lo, idx_bytes, r);
            }
        }
        for (int64_t c = 0; c < k; ++c) {
            const int64_t p = load_index(eids_out, o + c, idx_bytes, 0);
            store_index(eids_out, o + c, idx_bytes, (int64_t)(uint32_t)eid[beg + p] + base);
        }
        return;
    }
    int64_t chosen = 0;
    for (int64_t p = 0; p < d && chosen < k; ++p) {
        // take slot p with probability (k - chosen) / (d - p)
        if (uniform01(seed, (uint64_t)i, (uint64_t)p) * (double)(d - p) < (double)(k - chosen)) {
            store_index(eids_out, o + chosen, idx_bytes, (int64_t)(uint32_t)eid[beg + p] + base);
            ++chosen;
        }
    }
}

// ---- node sets and induced subgraphs --------------------------------------------------------------------
// first[v] = smallest position k with cand[k] = v, for the candidates that are not in the set yet (map[v] == 0).
// Three passes keep the result independent of thread scheduling: reset, atomicMin, claim.
__global__ void unique_reset_kernel(const void *cand, int idx_bytes, int base, int64_t M, int64_t n_nodes, int32_t *first,
                                    int *bad) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const int64_t v = load_index(cand, k, idx_bytes, base);
    if (v < 0 || v >= n_nodes) {
        *bad = 1;
        return;
    }
    first[v] = 0x7fffffff;
}
__global__ void unique_min_kernel(const void *cand, int idx_bytes, int base, int64_t M, int64_t n_nodes,
                                  const int32_t *map, int32_t *first) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const int64_t v = load_index(cand, k, idx_bytes, base);
    if (v < 0 || v >= n_nodes || map[v] != 0) return;
    atomicMin(&first[v], (int32_t)k);
}
__global__ void unique_flag_kernel(const void *cand, int idx_bytes, int base, int64_t M, int64_t n_nodes,
                                   const int32_t *map, const int32_t *first, int64_t *flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M) return;
    if (k == M) {
        flags[k] = 0;
        return;
    }
    const int64_t v = load_index(cand, k, idx_bytes, base);
    flags[k] = (v >= 0 && v < n_nodes && map[v] == 0 && first[v] == (int32_t)k) ? 1 : 0;
}
__global__ void unique_write_kernel(const void *cand, int idx_bytes, int base, int64_t M, const int64_t *flags,
                                    const int64_t *pos, int64_t set_size, int32_t *map, void *list_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M || !flags[k]) return;
    const int64_t v = load_index(cand, k, idx_bytes, base);
    map[v] = (int32_t)(set_size + pos[k] + 1);
    store_index(list_out, pos[k], idx_bytes, v + base);
}

// induced_subgraph(graph, nodes) — GNNGraphs/src/sampling.jl:173-203: for every listed node (in list order) its incoming
// edges (in edge order) whose source is listed too, relabelled by list position.
__global__ void induced_count_kernel(const uint32_t *rowptr, const int32_t *col, const int32_t *map, const void *nodes,
                                     int idx_bytes, int base, int64_t M, int64_t *counts) {
    // one WAVE per listed node, 64 slots of its row at a time (a thread per node walked a 17 000-edge hub alone: tens of
    // milliseconds per mini-batch); the kept slots are counted with a ballot
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i > M) return;
    if (i == M) {
        if (lane == 0) counts[i] = 0;
        return;
    }
    const int64_t v = load_index(nodes, i, idx_bytes, base);
    const int64_t beg = rowptr[v], end = rowptr[v + 1];
    int64_t c = 0;
    for (int64_t p0 = beg; p0 < end; p0 += 64) {
        const int64_t p = p0 + lane;
        const bool keep = p < end && map[col[min(p, end - 1)]] != 0;
        c += __popcll(__ballot(keep));
    }
    if (lane == 0) counts[i] = c;
}
__global__ void induced_fill_kernel(const uint32_t *rowptr, const int32_t *col, const int32_t *eid, const int32_t *map,
                                    const void *nodes, int idx_bytes, int base, int64_t M, const int64_t *offsets,
                                    void *s_out, void *t_out, void *eid_out) {
    // one wave per listed node; a kept slot's output position = edges kept before it in the row (ballot prefix): in-edge
    // order is preserved
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= M) return;
    const int64_t v = load_index(nodes, i, idx_bytes, base);
    const int64_t beg = rowptr[v], end = rowptr[v + 1];
    int64_t o = offsets[i];
    for (int64_t p0 = beg; p0 < end; p0 += 64) {
        const int64_t p = p0 + lane;
        const int64_t pc = min(p, end - 1);
        const int32_t m = map[col[pc]];
        const bool keep = p < end && m != 0;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int64_t at = o + __popcll(mask & ((1ull << lane) - 1ull));
            store_index(s_out, at, idx_bytes, (int64_t)m - 1 + base);
            store_index(t_out, at, idx_bytes, i + base);
            store_index(eid_out, at, idx_bytes, (int64_t)(uint32_t)eid[pc] + base);
        }
        o += __popcll(mask);
    }
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_sort_edge_index(const void *u, const void *v, int idx_bytes, int index_base, int64_t n_edges, void *u_out,
                          void *v_out, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "sort_edge_index: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "sort_edge_index: index_base %d", index_base);
    if (n_edges < 0) return fail(GNNMP_EINVAL, "sort_edge_index: negative size");
    if (n_edges == 0) return GNNMP_OK;
    if (!u || !v || !u_out || !v_out) return fail(GNNMP_EINVAL, "sort_edge_index: null pointer");
    uint64_t *keys = nullptr;
    if (hipMalloc((void **)&keys, sizeof(uint64_t) * (size_t)n_edges) != hipSuccess)
        return fail(GNNMP_EALLOC, "sort_edge_index: hipMalloc");
    int rc = sorted_pair_keys(u, v, idx_bytes, index_base, n_edges, keys, stream);
    if (rc == GNNMP_OK) {
        unpack_pairs_kernel<<<nb(n_edges), 256, 0, stream>>>(keys, idx_bytes, index_base, n_edges, u_out, v_out);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            rc = fail(GNNMP_ELAUNCH, "sort_edge_index: unpack");
    }
    (void)hipFree(keys);
    return rc;
}

int gnnmp_is_bidirected(const void *s, const void *t, int idx_bytes, int index_base, int64_t n_edges, int *result,
                        gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "is_bidirected: idx_bytes %d", idx_bytes);
    if (!result) return fail(GNNMP_EINVAL, "is_bidirected: null result");
    *result = 1;
    if (n_edges <= 0) return n_edges < 0 ? fail(GNNMP_EINVAL, "is_bidirected: negative size") : GNNMP_OK;
    if (!s || !t) return fail(GNNMP_EINVAL, "is_bidirected: null pointer");
    uint64_t *a = nullptr, *b = nullptr;
    int *flag = nullptr;
    int rc = GNNMP_OK, h = 0;
    if (hipMalloc((void **)&a, sizeof(uint64_t) * (size_t)n_edges) != hipSuccess ||
        hipMalloc((void **)&b, sizeof(uint64_t) * (size_t)n_edges) != hipSuccess ||
        hipMalloc((void **)&flag, sizeof(int)) != hipSuccess) {
        rc = fail(GNNMP_EALLOC, "is_bidirected: hipMalloc");
    } else {
        rc = sorted_pair_keys(s, t, idx_bytes, index_base, n_edges, a, stream);
        if (rc == GNNMP_OK) rc = sorted_pair_keys(t, s, idx_bytes, index_base, n_edges, b, stream);
        if (rc == GNNMP_OK) {
            (void)hipMemsetAsync(flag, 0, sizeof(int), stream);
            keys_differ_kernel<<<nb(n_edges), 256, 0, stream>>>(a, b, n_edges, flag);
            if (hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                rc = fail(GNNMP_ELAUNCH, "is_bidirected: compare");
            else
                *result = h ? 0 : 1;
        }
    }
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (flag) (void)hipFree(flag);
    return rc;
}

int gnnmp_has_self_loops(const void *s, const void *t, int idx_bytes, int64_t n_edges, int *result,
                         gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "has_self_loops: idx_bytes %d", idx_bytes);
    if (!result) return fail(GNNMP_EINVAL, "has_self_loops: null result");
    *result = 0;
    if (n_edges <= 0) return n_edges < 0 ? fail(GNNMP_EINVAL, "has_self_loops: negative size") : GNNMP_OK;
    if (!s || !t) return fail(GNNMP_EINVAL, "has_self_loops: null pointer");
    int *flag = nullptr;
    int h = 0;
    if (hipMalloc((void **)&flag, sizeof(int)) != hipSuccess) return fail(GNNMP_EALLOC, "has_self_loops: hipMalloc");
    (void)hipMemsetAsync(flag, 0, sizeof(int), stream);
    self_loop_kernel<<<nb(n_edges), 256, 0, stream>>>(s, t, idx_bytes, n_edges, flag);
    int rc = GNNMP_OK;
    if (hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        rc = fail(GNNMP_ELAUNCH, "has_self_loops");
    *result = h;
    (void)hipFree(flag);
    return rc;
}

int gnnmp_sample_neighbors(gnnmp_graph_t *plan, const void *nodes, int idx_bytes, int index_base, int64_t n_nodes,
                           int64_t K, int replace, uint64_t seed, int64_t *offsets, void *eids_out, int64_t capacity,
                           int64_t *total, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan) return fail(GNNMP_EINVAL, "sample_neighbors: null plan");
    if (plan->self_loops) return fail(GNNMP_EINVAL, "sample_neighbors: the plan must not add self loops");
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "sample_neighbors: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "sample_neighbors: index_base %d", index_base);
    if (n_nodes < 0 || capacity < 0) return fail(GNNMP_EINVAL, "sample_neighbors: negative size");
    if (!total || !offsets) return fail(GNNMP_EINVAL, "sample_neighbors: null offsets/total");
    *total = 0;
    if (n_nodes == 0) {
        if (hipMemsetAsync(offsets, 0, sizeof(int64_t), stream) != hipSuccess) return fail(GNNMP_ELAUNCH, "sample_neighbors");
        return GNNMP_OK;
    }
    if (!nodes) return fail(GNNMP_EINVAL, "sample_neighbors: null nodes");
    int rc = GNNMP_OK;
    int64_t *counts = nullptr;
    int *flag = nullptr;
    int hflag = 0;
    int64_t tot = 0;
    // (+ the scan's block sums behind the counts: one pooled block, no allocation or synchronisation inside the scan)
    const size_t counts_elems = (size_t)(n_nodes + 1) + exclusive_scan_workspace((size_t)(n_nodes + 1));
    PREP_HIP(prep_alloc((void **)&counts, sizeof(int64_t) * counts_elems));
    PREP_HIP(prep_alloc((void **)&flag, sizeof(int)));
    PREP_HIP(hipMemsetAsync(flag, 0, sizeof(int), stream));
    PREP_HIP(hipMemsetAsync(counts + n_nodes, 0, sizeof(int64_t), stream));
    sample_counts_kernel<<<nb(n_nodes), 256, 0, stream>>>(plan->rowptr, nodes, idx_bytes, index_base, n_nodes, plan->n_dst,
                                                          K, replace, counts, flag);
    PREP_HIP(hipGetLastError());
    PREP_G(exclusive_scan_i64(counts, offsets, (size_t)(n_nodes + 1), stream, counts + n_nodes + 1));
    PREP_HIP(hipMemcpyAsync(&tot, offsets + n_nodes, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipStreamSynchronize(stream));
    if (hflag) {
        rc = fail(GNNMP_EBOUNDS, "sample_neighbors: a seed node is outside 1..%lld", (long long)plan->n_dst);
        goto done;
    }
    *total = tot;
    if (tot > capacity) {
        rc = fail(GNNMP_EINVAL, "sample_neighbors: %lld edge ids do not fit the capacity %lld (offsets are valid: retry)",
                  (long long)tot, (long long)capacity);
        goto done;
    }
    if (tot > 0) {
        if (!eids_out) {
            rc = fail(GNNMP_EINVAL, "sample_neighbors: null eids_out");
            goto done;
        }
        sample_fill_kernel<<<nb(n_nodes), 256, 0, stream>>>(plan->rowptr, plan->eid, nodes, idx_bytes, index_base, n_nodes, K,
                                                            replace ? 1 : 0, seed, offsets, eids_out);
        PREP_HIP(hipGetLastError());
    }
done:
    prep_free(counts, sizeof(int64_t) * ((size_t)(n_nodes + 1) + exclusive_scan_workspace((size_t)(n_nodes + 1))));
    prep_free(flag, sizeof(int));
    return rc;
}

int gnnmp_unique_append(int32_t *map, int32_t *first, int64_t n_nodes, const void *cand, int idx_bytes, int index_base,
                        int64_t n_cand, int64_t set_size, void *list_out, int64_t *n_new, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "unique_append: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "unique_append: index_base %d", index_base);
    if (n_cand < 0 || n_nodes < 0 || set_size < 0 || n_nodes > 0x7fffffff || n_cand > 0x7fffffff)
        return fail(GNNMP_EINVAL, "unique_append: bad size");
    if (!n_new) return fail(GNNMP_EINVAL, "unique_append: null n_new");
    *n_new = 0;
    if (n_cand == 0) return GNNMP_OK;
    if (!map || !first || !cand || !list_out) return fail(GNNMP_EINVAL, "unique_append: null pointer");
    int rc = GNNMP_OK;
    int64_t *flags = nullptr, *pos = nullptr;
    int *bad = nullptr;
    int hbad = 0;
    int64_t tot = 0;
    PREP_HIP(prep_alloc((void **)&flags, sizeof(int64_t) * ((size_t)(n_cand + 1) + exclusive_scan_workspace((size_t)(n_cand + 1)))));
    PREP_HIP(prep_alloc((void **)&pos, sizeof(int64_t) * (size_t)(n_cand + 1)));
    PREP_HIP(prep_alloc((void **)&bad, sizeof(int)));
    PREP_HIP(hipMemsetAsync(bad, 0, sizeof(int), stream));
    unique_reset_kernel<<<nb(n_cand), 256, 0, stream>>>(cand, idx_bytes, index_base, n_cand, n_nodes, first, bad);
    unique_min_kernel<<<nb(n_cand), 256, 0, stream>>>(cand, idx_bytes, index_base, n_cand, n_nodes, map, first);
    unique_flag_kernel<<<nb(n_cand + 1), 256, 0, stream>>>(cand, idx_bytes, index_base, n_cand, n_nodes, map, first, flags);
    PREP_HIP(hipGetLastError());
    PREP_G(exclusive_scan_i64(flags, pos, (size_t)(n_cand + 1), stream, flags + n_cand + 1));
    unique_write_kernel<<<nb(n_cand), 256, 0, stream>>>(cand, idx_bytes, index_base, n_cand, flags, pos, set_size, map,
                                                        list_out);
    PREP_HIP(hipGetLastError());
    PREP_HIP(hipMemcpyAsync(&tot, pos + n_cand, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipStreamSynchronize(stream));
    if (hbad) rc = fail(GNNMP_EBOUNDS, "unique_append: a node is outside 1..%lld", (long long)n_nodes);
    *n_new = tot;
done:
    prep_free(flags, sizeof(int64_t) * ((size_t)(n_cand + 1) + exclusive_scan_workspace((size_t)(n_cand + 1))));
    prep_free(pos, sizeof(int64_t) * (size_t)(n_cand + 1));
    prep_free(bad, sizeof(int));
    return rc;
}

int gnnmp_induced_subgraph(gnnmp_graph_t *plan, const int32_t *map, const void *nodes, int idx_bytes, int index_base,
                           int64_t n_nodes, int64_t *offsets, void *s_out, void *t_out, void *eid_out, int64_t capacity,
                           int64_t *total, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!plan) return fail(GNNMP_EINVAL, "induced_subgraph: null plan");
    if (plan->self_loops) return fail(GNNMP_EINVAL, "induced_subgraph: the plan must not add self loops");
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "induced_subgraph: idx_bytes %d", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "induced_subgraph: index_base %d", index_base);
    if (n_nodes < 0 || capacity < 0) return fail(GNNMP_EINVAL, "induced_subgraph: negative size");
    if (!total || !offsets) return fail(GNNMP_EINVAL, "induced_subgraph: null offsets/total");
    *total = 0;
    if (n_nodes == 0) return GNNMP_OK;
    if (!map || !nodes) return fail(GNNMP_EINVAL, "induced_subgraph: null pointer");
    int rc = GNNMP_OK;
    int64_t *counts = nullptr;
    int64_t tot = 0;
    // (+ the scan's block sums behind the counts: one pooled block, no allocation or synchronisation inside the scan)
    const size_t counts_elems = (size_t)(n_nodes + 1) + exclusive_scan_workspace((size_t)(n_nodes + 1));
    PREP_HIP(prep_alloc((void **)&counts, sizeof(int64_t) * counts_elems));
    induced_count_kernel<<<nb((n_nodes + 1) * 64), 256, 0, stream>>>(plan->rowptr, plan->col, map, nodes, idx_bytes, index_base,
                                                              n_nodes, counts);
    PREP_HIP(hipGetLastError());
    PREP_G(exclusive_scan_i64(counts, offsets, (size_t)(n_nodes + 1), stream, counts + n_nodes + 1));
    PREP_HIP(hipMemcpyAsync(&tot, offsets + n_nodes, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    PREP_HIP(hipStreamSynchronize(stream));
    *total = tot;
    if (capacity == 0 && !s_out && !t_out && !eid_out) goto done;   // count-only call: offsets and *total are the result
    if (tot > capacity) {
        rc = fail(GNNMP_EINVAL, "induced_subgraph: %lld edges do not fit the capacity %lld (offsets are valid: retry)",
                  (long long)tot, (long long)capacity);
        goto done;
    }
    if (tot > 0) {
        if (!s_out || !t_out || !eid_out) {
            rc = fail(GNNMP_EINVAL, "induced_subgraph: null output");
            goto done;
        }
        induced_fill_kernel<<<nb(n_nodes * 64), 256, 0, stream>>>(plan->rowptr, plan->col, plan->eid, map, nodes, idx_bytes,
                                                             index_base, n_nodes, offsets, s_out, t_out, eid_out);
        PREP_HIP(hipGetLastError());
    }
done:
    prep_free(counts, sizeof(int64_t) * ((size_t)(n_nodes + 1) + exclusive_scan_workspace((size_t)(n_nodes + 1))));
    return rc;
}

}  // extern "C"
