// plan_batch.hip — the plan of a BATCH of member graphs without a sort.
//   MLUtils.batch(gs) (GNNGraphs/src/transform.jl:682-709) only offsets the members' node ids and concatenates their edge lists, member
//   after member; the dst-sorted stable CSR of the result is therefore the CONCATENATION of the members' CSRs: rowptr pieces shifted by
//   the slots before them, col by the nodes before them, edge positions by the edges before them (plan-added self loops stay at
//   E_batch + node).  The reference's graph-classification loop makes a new batch every step
//   (examples/graph_classification_tudataset.jl:70-71 DataLoader(...; shuffle = true, collate = true), :97-104), so this is per-step work:
//     gnnmp_plan_concat   members given as plan handles (what a `batch(gs)` override holds: one cached plan per member graph)
//     gnnmp_plan_select   members given as graph ids into ONE resident plan of the whole dataset batched once (the MI355X way to hold a
//                         dataset: 288 GB of HBM, nothing crosses PCIe per step) — also what getobs / getgraph of a batched graph needs
//                         (GNNGraphs/src/gnngraph.jl:311, transform.jl:827-876)
//   Both are two launches and no host synchronisation; the index arrays come from the stream-ordered block pool (pool.h) and go back to
//   it with gnnmp_plan_release.  gnnmp_plan_edge_index gives the batch's COO back (s, t in original edge order) for callers that want it.
#include <algorithm>
#include <vector>

#include "common.h"
#include "pool.h"

namespace gnnmp {

int plan_dispose(gnnmp_graph_t *p, hipStream_t stream, bool stream_known);   // plan.hip

// where a member's piece of its source plan lives
struct MemberDesc {
    const uint32_t *rowptr;   // source rowptr at the member's first row
    const int32_t *col;       // source col at the member's first slot
    const int32_t *eid;       // source eid at the member's first slot
    uint32_t slot_base;       // value of rowptr[0] above (subtracted)
    int32_t node_base;        // source id of the member's first node (subtracted from col)
    uint32_t edge_base;       // source edge position of the member's first edge
    uint32_t src_edges;       // n_edges of the SOURCE plan: eid >= src_edges is a plan-added self loop of source node eid - src_edges
};
static_assert(sizeof(MemberDesc) == 40, "MemberDesc layout");

struct BatchTab {             // device arrays inside the plan's block
    int64_t *row_off;         // [k + 1] first row of each member in the batch (= the batch's node_ptr / seg_ptr)
    uint32_t *slot_off;       // [k + 1] first slot of each member
    MemberDesc *desc;         // [k]
    int32_t *status;          // [0] != 0: the table does not describe what the caller announced (bit 0: graph id out of range,
                              //     bit 1: totals differ) — every index array is then left empty (rowptr = 0)
};

constexpr int SEL_THREADS = 256;

// ---- the member table of gnnmp_plan_select.  A member per thread over MANY blocks: the three dependent loads of a member (id -> node
// offsets -> slot offsets) touch a different cache line each — one block alone spends 79 us on the 25 000 line requests of 8 192 members
// (measured), the whole chip a few.  Sizes go to row_off / slot_off in place; the block that finishes LAST (a ticket in status[1]) turns
// them into offsets with one in-place exclusive scan (integer sums: the same bits whichever block it is) -----------------------------------
__global__ void __launch_bounds__(SEL_THREADS) select_table_kernel(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                                   const int32_t *__restrict__ eid, int self_loops, int64_t src_edges,
                                                                   const int64_t *__restrict__ node_ptr, int64_t n_graphs,
                                                                   const void *__restrict__ ids, int idx_bytes, int base, int64_t k,
                                                                   int64_t n_rows, int64_t n_slots, BatchTab tab, int64_t *seg_ptr_out) {
    __shared__ int64_t wave_rows[SEL_THREADS / 64];
    __shared__ int64_t wave_slots[SEL_THREADS / 64];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int64_t m = (int64_t)blockIdx.x * SEL_THREADS + tid;
        if (m < k) {
            int64_t g = load_index(ids, m, idx_bytes, base);
            if (g < 0 || g >= n_graphs) { atomicOr(tab.status, 1); g = 0; }
            const int64_t nb = node_ptr[g], ne = node_ptr[g + 1];
            const uint32_t sb = rowptr[nb], se = rowptr[ne];
            tab.row_off[m] = ne - nb;                  // sizes for now
            tab.slot_off[m] = se - sb;
            MemberDesc d;
            d.rowptr = rowptr + nb;
            d.col = col + sb;
            d.eid = eid + sb;
            d.slot_base = sb;
            d.node_base = (int32_t)nb;
            d.edge_base = sb - (self_loops ? (uint32_t)nb : 0u);   // the dataset's edges are member-major like its rows
            d.src_edges = (uint32_t)src_edges;
            tab.desc[m] = d;
        }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(tab.status + 1, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // the last block: rounds of SEL_THREADS x 8 members, thread t owns 8 CONSECUTIVE ones (their loads in flight together: the round trips
    // of a sequential walk were 20 us of this kernel), one block-wide scan of the threads' sums per round
    constexpr int PER = 8;
    int64_t tot_r = 0, tot_s = 0;          // members before this round (uniform)
    for (int64_t q0 = 0; q0 < k; q0 += (int64_t)SEL_THREADS * PER) {
        const int64_t mt = q0 + (int64_t)tid * PER;
        int64_t rr[PER];
        uint32_t ss[PER];
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const bool v = mt + r < k;
            rr[r] = v ? tab.row_off[mt + r] : 0;
            ss[r] = v ? tab.slot_off[mt + r] : 0u;
        }
        int64_t trow = 0, tslot = 0;
#pragma unroll
        for (int r = 0; r < PER; ++r) { trow += rr[r]; tslot += ss[r]; }
        int64_t ir = trow, is = tslot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t ur = __shfl_up(ir, o, 64), us = __shfl_up(is, o, 64);
            if (lane >= o) { ir += ur; is += us; }
        }
        __syncthreads();
        if (lane == 63) { wave_rows[wave] = ir; wave_slots[wave] = is; }
        __syncthreads();
        int64_t run_r = tot_r + ir - trow, run_s = tot_s + is - tslot;
#pragma unroll
        for (int w = 0; w < SEL_THREADS / 64; ++w) {
            const int64_t wr = wave_rows[w], ws = wave_slots[w];
            if (w < wave) { run_r += wr; run_s += ws; }
            tot_r += wr; tot_s += ws;
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            if (mt + r < k) {
                tab.row_off[mt + r] = run_r;
                tab.slot_off[mt + r] = (uint32_t)run_s;
                if (seg_ptr_out) seg_ptr_out[mt + r] = run_r;
            }
            run_r += rr[r];
            run_s += ss[r];
        }
    }
    if (tid == 0) {
        tab.row_off[k] = tot_r;
        tab.slot_off[k] = (uint32_t)tot_s;
        if (seg_ptr_out) seg_ptr_out[k] = tot_r;
        if (tot_r != n_rows || tot_s != n_slots) atomicOr(tab.status, 2);
    }
}

// largest m in [lo, hi) with off[m] <= x  (off[hi] > x)
template <class T>
__device__ __forceinline__ int member_of(const T *__restrict__ off, int lo, int hi, T x) {
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- the index arrays: blocks [0, row_blocks) write rowptr (+ node map, graph indicator), the rest col and eid --------------------------
__global__ void __launch_bounds__(256) batch_fill_kernel(BatchTab tab, int k, int64_t n_rows, int64_t n_slots, int self_loops,
                                                         unsigned row_blocks, uint32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                         int32_t *__restrict__ eid, int32_t *__restrict__ node_map,
                                                         void *__restrict__ indicator, int ind_bytes, int ind_base) {
    const bool ok = tab.status[0] == 0;
    // the members this block's 256 consecutive rows / slots can belong to: two 13-step searches per BLOCK (first and last item), then
    // every thread searches only that handful of members (a 20..40-node member graph holds ~30 rows / ~120 slots)
    __shared__ int span[2];
    const bool rows = blockIdx.x < row_blocks;
    if (ok && threadIdx.x < 2 && k > 0) {
        if (rows) {
            const int64_t i = min((int64_t)blockIdx.x * 256 + (threadIdx.x ? 255 : 0), n_rows > 0 ? n_rows - 1 : 0);
            span[threadIdx.x] = member_of<int64_t>(tab.row_off, 0, k, i);
        } else {
            const int64_t p = min((int64_t)(blockIdx.x - row_blocks) * 256 + (threadIdx.x ? 255 : 0), n_slots > 0 ? n_slots - 1 : 0);
            span[threadIdx.x] = member_of<uint32_t>(tab.slot_off, 0, k, (uint32_t)p);
        }
    }
    __syncthreads();
    const int mlo = span[0], mhi = span[1] + 1;
    if (rows) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i > n_rows) return;
        if (!ok) { rowptr[i] = 0; return; }
        if (i == n_rows) { rowptr[i] = (uint32_t)n_slots; return; }
        const int m = member_of<int64_t>(tab.row_off, mlo, mhi, i);
        const MemberDesc d = tab.desc[m];
        const int64_t local = i - tab.row_off[m];
        rowptr[i] = tab.slot_off[m] + (d.rowptr[local] - d.slot_base);
        if (node_map) node_map[i] = d.node_base + (int32_t)local;
        if (indicator) store_index(indicator, i, ind_bytes, (int64_t)m + ind_base);
    } else {
        const int64_t p = (int64_t)(blockIdx.x - row_blocks) * 256 + threadIdx.x;
        if (p >= n_slots || !ok) return;
        const int m = member_of<uint32_t>(tab.slot_off, mlo, mhi, (uint32_t)p);
        const MemberDesc d = tab.desc[m];
        const uint32_t so = tab.slot_off[m];
        const int64_t ro = tab.row_off[m];
        const uint32_t q = (uint32_t)p - so;
        col[p] = d.col[q] - d.node_base + (int32_t)ro;
        const uint32_t e = (uint32_t)d.eid[q];
        const uint32_t edge_off = so - (self_loops ? (uint32_t)ro : 0u);
        const uint32_t e_batch = (uint32_t)(n_slots - (self_loops ? n_rows : 0));
        eid[p] = (int32_t)(e < d.src_edges ? e - d.edge_base + edge_off
                                           : e_batch + (uint32_t)ro + (e - d.src_edges - (uint32_t)d.node_base));
    }
}

// s, t of the plan's graph in ORIGINAL edge order: slot p of row i holds edge eid[p] = (col[p] -> i); plan-added self loops are skipped
__global__ void __launch_bounds__(256) plan_edge_index_kernel(const uint32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                              const int32_t *__restrict__ eid, int64_t n_dst, int64_t n_total,
                                                              int64_t n_edges, int idx_bytes, int base, void *out_src, void *out_dst) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_total) return;
    const int64_t e = (uint32_t)eid[p];
    if (e >= n_edges) return;
    int64_t lo = 0, hi = n_dst;               // largest row with rowptr[row] <= p
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rowptr[mid] <= p) lo = mid; else hi = mid;
    }
    store_index(out_src, e, idx_bytes, (int64_t)col[p] + base);
    store_index(out_dst, e, idx_bytes, lo + base);
}

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// the plan object + its pooled block; the arrays of `tab` point into the block
static int batch_plan_alloc(gnnmp_graph_t **out, BatchTab *tab, int64_t k, int64_t n_rows, int64_t n_slots, int self_loops,
                            hipStream_t stream) {
    const size_t o_rowptr = 0;
    const size_t o_col = o_rowptr + up256(sizeof(uint32_t) * (size_t)(n_rows + 1));
    const size_t o_eid = o_col + up256(sizeof(int32_t) * (size_t)std::max<int64_t>(n_slots, 1));
    const size_t o_rowoff = o_eid + up256(sizeof(int32_t) * (size_t)std::max<int64_t>(n_slots, 1));
    const size_t o_slotoff = o_rowoff + up256(sizeof(int64_t) * (size_t)(k + 1));
    const size_t o_desc = o_slotoff + up256(sizeof(uint32_t) * (size_t)(k + 1));
    const size_t o_status = o_desc + up256(sizeof(MemberDesc) * (size_t)std::max<int64_t>(k, 1));
    const size_t total = o_status + 256;
    gnnmp_graph_t *p = new gnnmp_graph_t();
    if (!pool_take(&p->block, &p->block_bytes, total, stream)) {
        delete p;
        return fail(GNNMP_EALLOC, "batch plan: hipMalloc of %zu bytes failed", total);
    }
    unsigned char *b = static_cast<unsigned char *>(p->block);
    p->rowptr = reinterpret_cast<uint32_t *>(b + o_rowptr);
    p->col = reinterpret_cast<int32_t *>(b + o_col);
    p->eid = reinterpret_cast<int32_t *>(b + o_eid);
    p->status = reinterpret_cast<int32_t *>(b + o_status);
    tab->row_off = reinterpret_cast<int64_t *>(b + o_rowoff);
    tab->slot_off = reinterpret_cast<uint32_t *>(b + o_slotoff);
    tab->desc = reinterpret_cast<MemberDesc *>(b + o_desc);
    tab->status = p->status;
    p->n_src = p->n_dst = n_rows;
    p->n_total = n_slots;
    p->n_edges = n_slots - (self_loops ? n_rows : 0);
    p->self_loops = self_loops;
    p->long_thresh = plan_long_thresh(n_slots);
    p->bytes = (int64_t)total;
    *out = p;
    return GNNMP_OK;
}

static int batch_plan_fill(gnnmp_graph_t *p, const BatchTab &tab, int64_t k, int32_t *node_map, void *indicator, int ind_bytes,
                           int ind_base, int64_t max_degree_bound, hipStream_t stream) {
    const unsigned row_blocks = (unsigned)((p->n_dst + 1 + 255) / 256);
    const unsigned slot_blocks = (unsigned)((p->n_total + 255) / 256);
    batch_fill_kernel<<<row_blocks + slot_blocks, 256, 0, stream>>>(tab, (int)k, p->n_dst, p->n_total, p->self_loops, row_blocks, p->rowptr,
                                                                   p->col, p->eid, node_map, indicator, ind_bytes, ind_base);
    GNNMP_LAUNCH_CHECK("batch_fill_kernel");
    p->max_degree = max_degree_bound;
    if (max_degree_bound > p->long_thresh) return plan_build_long_rows(p, stream);   // (hub rows: the chunk tables, with a synchronisation)
    return GNNMP_OK;
}

}  // namespace gnnmp

using namespace gnnmp;

extern "C" {

int gnnmp_plan_select(gnnmp_graph_t **out, const gnnmp_graph_t *ds, const int64_t *node_ptr, int64_t n_graphs, const void *ids,
                      int idx_bytes, int index_base, int64_t k, int64_t n_rows, int64_t n_slots, int64_t *seg_ptr_out,
                      int32_t *node_map_out, void *graph_indicator_out, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return fail(GNNMP_EINVAL, "plan_select: out is NULL");
    *out = nullptr;
    if (!ds || !node_ptr || n_graphs < 0 || k < 0 || (k > 0 && !ids) || n_rows < 0 || n_slots < 0)
        return fail(GNNMP_EINVAL, "plan_select: bad argument");
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "plan_select: idx_bytes must be 4 or 8 (got %d)", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "plan_select: index_base must be 0 or 1 (got %d)", index_base);
    if (ds->n_src != ds->n_dst) return fail(GNNMP_EINVAL, "plan_select: the dataset plan is not a square graph");
    if (k >= (int64_t)INT32_MAX || n_rows >= (int64_t)INT32_MAX || n_slots >= (int64_t)GNNMP_MAX_SLOTS)
        return fail(GNNMP_EUNSUPPORTED, "plan_select: the batch exceeds the plan format");
    gnnmp_graph_t *p = nullptr;
    BatchTab tab = {};
    int rc = batch_plan_alloc(&p, &tab, k, n_rows, n_slots, ds->self_loops, stream);
    if (rc != GNNMP_OK) return rc;
    hipError_t e = hipMemsetAsync(tab.status, 0, 4 * sizeof(int32_t), stream);        // [0] status, [1] the last-block ticket
    if (e == hipSuccess) {
        select_table_kernel<<<(unsigned)std::max<int64_t>(1, (k + SEL_THREADS - 1) / SEL_THREADS), SEL_THREADS, 0, stream>>>(
            ds->rowptr, ds->col, ds->eid, ds->self_loops, ds->n_edges, node_ptr, n_graphs, ids, idx_bytes, index_base, k, n_rows, n_slots, tab,
            seg_ptr_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        rc = batch_plan_fill(p, tab, k, node_map_out, graph_indicator_out, idx_bytes, index_base, ds->max_degree, stream);
    else
        rc = hip_fail(e, "select_table_kernel");
    if (rc != GNNMP_OK) {
        plan_dispose(p, stream, true);
        return rc;
    }
    *out = p;
    return GNNMP_OK;
}

int gnnmp_plan_concat(gnnmp_graph_t **out, const gnnmp_graph_t *const *members, int64_t k, int64_t *seg_ptr_out,
                      void *graph_indicator_out, int idx_bytes, int index_base, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return fail(GNNMP_EINVAL, "plan_concat: out is NULL");
    *out = nullptr;
    if (k < 0 || (k > 0 && !members)) return fail(GNNMP_EINVAL, "plan_concat: bad argument");
    if (graph_indicator_out && idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "plan_concat: idx_bytes must be 4 or 8");
    if (graph_indicator_out && index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "plan_concat: index_base must be 0 or 1");
    if (k >= (int64_t)INT32_MAX) return fail(GNNMP_EUNSUPPORTED, "plan_concat: too many members");
    // the member table on the host: offsets + where each member's arrays live
    std::vector<int64_t> row_off((size_t)k + 1, 0);
    std::vector<uint32_t> slot_off((size_t)k + 1, 0);
    std::vector<MemberDesc> desc((size_t)k);
    int64_t rows = 0, slots = 0, maxdeg = 0;
    int loops = k > 0 && members[0] ? members[0]->self_loops : 0;
    for (int64_t m = 0; m < k; ++m) {
        const gnnmp_graph_t *q = members[m];
        if (!q) return fail(GNNMP_EINVAL, "plan_concat: member %lld is NULL", (long long)m);
        if (q->n_src != q->n_dst) return fail(GNNMP_EINVAL, "plan_concat: member %lld is not a square graph", (long long)m);
        if (q->self_loops != loops) return fail(GNNMP_EINVAL, "plan_concat: members with and without added self loops");
        row_off[(size_t)m] = rows;
        slot_off[(size_t)m] = (uint32_t)slots;
        MemberDesc &d = desc[(size_t)m];
        d.rowptr = q->rowptr; d.col = q->col; d.eid = q->eid;
        d.slot_base = 0; d.node_base = 0; d.edge_base = 0;
        d.src_edges = (uint32_t)q->n_edges;
        rows += q->n_dst;
        slots += q->n_total;
        maxdeg = std::max(maxdeg, q->max_degree);
        if (rows >= (int64_t)INT32_MAX || slots >= (int64_t)GNNMP_MAX_SLOTS)
            return fail(GNNMP_EUNSUPPORTED, "plan_concat: the batch exceeds the plan format (N < 2^31 - 1, E' < 2^32 - 65536)");
    }
    row_off[(size_t)k] = rows;
    slot_off[(size_t)k] = (uint32_t)slots;
    gnnmp_graph_t *p = nullptr;
    BatchTab tab = {};
    int rc = batch_plan_alloc(&p, &tab, k, rows, slots, loops, stream);
    if (rc != GNNMP_OK) return rc;
    hipError_t e = hipMemcpyAsync(tab.row_off, row_off.data(), sizeof(int64_t) * row_off.size(), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tab.slot_off, slot_off.data(), sizeof(uint32_t) * slot_off.size(), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess && k > 0) e = hipMemcpyAsync(tab.desc, desc.data(), sizeof(MemberDesc) * desc.size(), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemsetAsync(tab.status, 0, 4 * sizeof(int32_t), stream);
    if (e == hipSuccess && seg_ptr_out)
        e = hipMemcpyAsync(seg_ptr_out, row_off.data(), sizeof(int64_t) * row_off.size(), hipMemcpyHostToDevice, stream);
    // (the host vectors die with this call: the copies above are from pageable memory, which hipMemcpyAsync stages before it returns)
    if (e != hipSuccess) rc = hip_fail(e, "plan_concat: upload of the member table");
    if (rc == GNNMP_OK) rc = batch_plan_fill(p, tab, k, nullptr, graph_indicator_out, idx_bytes, index_base, maxdeg, stream);
    if (rc != GNNMP_OK) {
        plan_dispose(p, stream, true);
        return rc;
    }
    *out = p;
    return GNNMP_OK;
}

int gnnmp_plan_edge_index(const gnnmp_graph_t *p, int idx_bytes, int index_base, void *out_src, void *out_dst, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_edge_index: null plan");
    if (idx_bytes != 4 && idx_bytes != 8) return fail(GNNMP_EINVAL, "plan_edge_index: idx_bytes must be 4 or 8 (got %d)", idx_bytes);
    if (index_base != 0 && index_base != 1) return fail(GNNMP_EINVAL, "plan_edge_index: index_base must be 0 or 1 (got %d)", index_base);
    if (p->n_edges == 0) return GNNMP_OK;
    if (!out_src || !out_dst) return fail(GNNMP_EINVAL, "plan_edge_index: null output");
    plan_edge_index_kernel<<<(unsigned)((p->n_total + 255) / 256), 256, 0, stream>>>(p->rowptr, p->col, p->eid, p->n_dst, p->n_total,
                                                                                    p->n_edges, idx_bytes, index_base, out_src, out_dst);
    GNNMP_LAUNCH_CHECK("plan_edge_index_kernel");
    return GNNMP_OK;
}

/* 0 = the plan's member table matched the totals its caller announced; synchronises the stream (tests, debugging) */
int gnnmp_plan_status(const gnnmp_graph_t *p, gnnmp_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(GNNMP_EINVAL, "plan_status: null plan");
    if (!p->status) return GNNMP_OK;
    int32_t st = 0;
    GNNMP_HIP(hipMemcpyAsync(&st, p->status, sizeof(st), hipMemcpyDeviceToHost, stream));
    GNNMP_HIP(hipStreamSynchronize(stream));
    if (st & 1) return fail(GNNMP_EBOUNDS, "plan_select: a graph id is outside 1..num_graphs");
    if (st & 2) return fail(GNNMP_EINVAL, "plan_select: n_rows / n_slots differ from the selected members' totals");
    return GNNMP_OK;
}

}  // extern "C"
