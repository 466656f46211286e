/*
 * gnnmp.h — C ABI of libgnnmp.so, the MI355X (gfx950) message-passing engine.
 *
 * This is the drop-in boundary for the GNNlib.jl hot path
 *   propagate / apply_edges / aggregate_neighbors  (GNNlib/src/msgpass.jl:71-79,121-129,145-149)
 * and the leaf ops it is built from
 *   _gather / _scatter                             (GNNGraphs/src/gatherscatter.jl:4,12-18)
 * The reference has no FFI: its device seam is the Julia package extension
 * GNNlib/ext/GNNlibAMDGPUExt.jl:13-32 (methods of GNNlib.propagate on AnyROCMatrix).  A replacement
 * extension `@ccall`s the symbols below (see INTEGRATION.md for the exact stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer, borrowed for the duration of the call only;
 *     the caller allocates inputs AND outputs.  The only long-lived object is the opaque plan.
 *   - feature arrays are Julia column-major (D, N)  ==  C row-major [N][D]: one node's / edge's
 *     feature vector is contiguous.  fp32 (every entry point) or fp64 (the *_f64 entry points: propagate, _gather, _scatter).
 *   - index arrays are passed exactly as the reference holds them: Int64 or Int32 (idx_bytes = 8 | 4),
 *     1-based (index_base = 1, Julia) or 0-based (index_base = 0).
 *   - `stream` is a hipStream_t (NULL = the null stream).  Compute entry points launch on it and
 *     return; they never call hipDeviceSynchronize and never allocate (exception: the first use of a
 *     plan with a row longer than GNNMP_LONG_ROW at a larger D than before grows plan-owned scratch
 *     with hipMalloc: the workspace of the split rows' partials and, since round 5, their slice
 *     partials and arrival counters — the split rows are folded inside the row kernels by the last
 *     chunk to arrive, not by a second launch).  gnnmp_plan_create synchronises `stream` (it is graph prep, done
 *     once per graph, outside the timed path).
 *   - a plan carries scratch of its own (the partials of split rows, the tile ticket of the fused layer kernel, cached
 *     orderings): calls that take the SAME plan must be ordered on ONE stream (or by events) — two streams may run
 *     different plans, or read-only queries of one plan, concurrently, but not two compute calls on one plan.  The
 *     reference has the same shape: one GNNGraph, one task.  Host threads: besides the tuning knobs (gnnmp_tune, experiments
 *     only) and the thread-local error string, the library's only process-wide state is (i) the per-DEVICE tables behind a mutex —
 *     which kernels have been opted into large LDS on which device, and the stream-ordered block pool that backs
 *     gnnmp_plan_concat / gnnmp_plan_select / gnnmp_chain_jobs_pack (one table per device: a block, its event and its stream are
 *     only ever matched against takers of the SAME device) — and (ii) a per-thread scratch cache of graph prep, keyed on the device.
 *   - several devices in one process (a Julia host calling AMDGPU.device!(d) between calls): supported.  Every entry point acts on
 *     the calling thread's CURRENT device (hipGetDevice) — the one `stream` and every pointer of the call must belong to.  Plans,
 *     job handles and arenas belong to the device they were created on; gnnmp_arena_* refuse a foreign current device with
 *     GNNMP_EINVAL.  Release a pooled object (gnnmp_plan_release, gnnmp_chain_jobs_release) with a stream of ITS device.
 *   - return value: 0 = ok, negative = gnnmp_status; nothing throws, nothing aborts.
 *     gnnmp_last_error() gives a thread-local message for the last failing call.
 *   - determinism: no entry point except *_atomic_* uses floating-point atomics.  Per-destination
 *     reductions run in the ORIGINAL COO edge order (the order NNlib.scatter uses on the CPU) for
 *     rows of at most GNNMP_LONG_ROW edges — bit-identical to the reference CPU gather->scatter
 *     path; longer rows are split into fixed chunks combined in fixed order (run-to-run
 *     deterministic, within 1e-5 rel of the sequential sum).
 */
#ifndef GNNMP_H
#define GNNMP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNMP_VERSION 100 /* 0.1.0 */

typedef void *gnnmp_stream_t;           /* hipStream_t */
typedef struct gnnmp_graph gnnmp_graph_t; /* opaque: dst-sorted CSR plan of one (s, t) edge index */

typedef enum {
    GNNMP_OK = 0,
    GNNMP_EINVAL = -1,       /* bad argument (null pointer, bad enum, bad size)            */
    GNNMP_EBOUNDS = -2,      /* index outside 1..N (validate != 0 only)                    */
    GNNMP_EALLOC = -3,       /* hipMalloc failed                                           */
    GNNMP_ELAUNCH = -4,      /* kernel launch / HIP runtime error                          */
    GNNMP_EUNSUPPORTED = -5  /* valid request outside the build's envelope (e.g. E' >= 2^32 - 65536) */
} gnnmp_status;

/* aggregation operator — the `aggr` argument of propagate / aggregate_neighbors / reduce_nodes
 * (GNNGraphs/src/gatherscatter.jl:12-18 -> NNlib.scatter).  Empty destinations keep the identity:
 * 0 for SUM and MEAN, -Inf for MAX, +Inf for MIN.  MEAN = sum / count (true division). */
typedef enum { GNNMP_SUM = 0, GNNMP_MEAN = 1, GNNMP_MAX = 2, GNNMP_MIN = 3 } gnnmp_aggr;

/* built-in message functions with a fused path (GNNlib/src/msgpass.jl:162,191-208).
 * E_MUL_XJ with a vector `e` is W_MUL_XJ with w = e (msgpass.jl:223-228). */
typedef enum { GNNMP_COPY_XJ = 0, GNNMP_W_MUL_XJ = 1 } gnnmp_msg;

/* activation fused into gnnmp_dense_f32's epilogue (the layers' `σ`): IDENTITY and RELU everywhere an `act` is taken;
 * SOFTPLUS (NNlib.softplus = log1p(exp(-|x|)) + relu(x)) and TANH also as the second factor of gnnmp_propagate_cg_f32 and
 * in gnnmp_bias_act_f32; SWISH (x * sigmoid(x), egnn_conv's Dense layers) in gnnmp_bias_act_f32 only. */
typedef enum {
    GNNMP_ACT_IDENTITY = 0, GNNMP_ACT_RELU = 1, GNNMP_ACT_SOFTPLUS = 2, GNNMP_ACT_TANH = 3, GNNMP_ACT_SWISH = 4
} gnnmp_act;
/* per-edge attention logit of gnnmp_attn_conv_f32 (Q_i = row i of the target array, K_j = row j of the source array) */
typedef enum {
    GNNMP_ATTN_GAT = 0,   /* leakyrelu(a[h][0:C] . Q_i + a[h][C:2C] . K_j)      gat_message   conv.jl:152-167 */
    GNNMP_ATTN_GATV2 = 1, /* a[h] . leakyrelu(Q_i + K_j)                        gatv2_message conv.jl:202-214 */
    GNNMP_ATTN_DOT = 2,   /* (Q_i . K_j) / scale, values from a third array     transformer_message_uij conv.jl:609-616 */
    GNNMP_ATTN_COS = 3    /* scale * cos(Q_i, K_j), single head                 agnn_conv     conv.jl:337-352 */
} gnnmp_attn;

/* Destinations with more edges than the plan's threshold are split into balanced chunks (see determinism note
 * above).  The threshold is chosen per plan in [GNNMP_MIN_LONG_ROW, GNNMP_LONG_ROW] from the graph size
 * (~4e-5 * E', so that no single sequential row can become the kernel's tail) and reported by gnnmp_plan_info
 * (info[7]); rows of at most GNNMP_MIN_LONG_ROW edges are never split. */
#define GNNMP_LONG_ROW 512
/* a plan holds fewer than this many slots (edges + added self loops): slots and edge positions are unsigned 32-bit values, with
 * headroom so that no slot counter inside a kernel can wrap */
#define GNNMP_MAX_SLOTS 4294901760LL /* 2^32 - 65536 */
#define GNNMP_MIN_LONG_ROW 64

int gnnmp_version(void);
const char *gnnmp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Plan: the graph side of GNNGraph{COO_T} (GNNGraphs/src/gnngraph.jl:108-117) as the kernels want it.
 * Built once per (s, t) pair; replaces the per-call `sparse(s,t,...)` rebuild of the reference fast path
 * (GNNlib/src/msgpass.jl:215-218 -> GNNGraphs/src/convert.jl:221-237) and the per-call index
 * concatenation of add_self_loops (GNNGraphs/src/transform.jl:12-28).
 *
 *   src, dst     : s and t, n_edges entries each (edge k goes src[k] -> dst[k]).
 *   n_src, n_dst : number of source / destination nodes (equal for GNNGraph; different for the bipartite
 *                  case of GNNlib/src/utils.jl:123-125 and for reduce_nodes-as-scatter).
 *   add_self_loops != 0 : append the edges (i, i), i = 1..n (requires n_src == n_dst), AFTER the given
 *                  edges, exactly like transform.jl:12-28 (never de-duplicates).  E' = n_edges + n.
 *   validate != 0: check 1 <= s <= n_src, 1 <= t <= n_dst on the device (GNNGraphs/src/convert.jl:47-54)
 *                  and return GNNMP_EBOUNDS instead of building.
 * The plan stores: rowptr[n_dst+1] (unsigned 32-bit), col[E'] (0-based source of each slot, int32), eid[E'] (0-based original
 * edge position of each slot, unsigned 32-bit; self loops are n_edges + i).  Slots of one destination keep the original
 * edge order (stable sort).  Limits: n_src, n_dst < 2^31 - 1 and E' < GNNMP_MAX_SLOTS = 2^32 - 65536 (COO_T admits any
 * Integer, GNNGraphs/src/abstracttypes.jl:1: the INDEX type may be Int64 at any size; the COUNT of edges of one plan is
 * bounded by the 32-bit slots); beyond that GNNMP_EUNSUPPORTED.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_plan_create(gnnmp_graph_t **out, const void *src, const void *dst, int idx_bytes,
                      int index_base, int64_t n_src, int64_t n_dst, int64_t n_edges,
                      int add_self_loops, int validate, gnnmp_stream_t stream);
/* The plan of a GNNGraph{SPARSE_T} (GNNGraphs/src/abstracttypes.jl:5, gnngraph.jl:108): the graph IS a compressed-sparse-column
 * adjacency A with A[s, t] != 0 for an edge s -> t, and the reference's device seam takes such graphs next to COO ones
 * (GNNlib/ext/GNNlibAMDGPUExt.jl:13-32 dispatches on Union{COO_T, SPARSE_T}).  Column t of A lists the sources of t's incoming edges
 * in ascending order, and edge_index(g) = findnz(A) walks the columns in order (GNNGraphs/src/query.jl:14, convert.jl:62-73): the
 * edges of a sparse graph are ALREADY destination-sorted and the plan is the CSC structure itself — no sort:
 *      rowptr = colptr - index_base      col = rowval - index_base      eid[p] = p   (edge k of edge_index(g) is slot k)
 * so that edge data in findnz order (get_edge_weight(g) = nzval, query.jl:18) is what `w` / `e` of the propagate entry points take.
 *   colptr : n_dst + 1 column pointers (SparseMatrixCSC.colptr / ROCSparseMatrixCSC.colPtr), rowval : n_edges row indices.
 *   colptr[1] = 1, colptr[end] = n_edges + 1, non-decreasing, 1 <= rowval <= n_src are ALWAYS checked on the device (GNNMP_EBOUNDS) —
 *   the build synchronises the stream anyway; `validate` is accepted for symmetry with gnnmp_plan_create and ignored.
 * Stored entries whose VALUE is zero are edges: a sparse graph's num_edges is nnz(A) (convert.jl:199) and its edge_index is
 * to_coo(A::SPARSE_T) = findnz(A) (convert.jl:62-73), which walks the STORED entries, explicit zeros included — the `findall(!=(0), A)` of
 * convert.jl:75-79 is the dense-matrix method.  Same limits and the same stream synchronisation as gnnmp_plan_create. */
int gnnmp_plan_from_csc(gnnmp_graph_t **out, const void *colptr, const void *rowval, int idx_bytes, int index_base, int64_t n_src,
                        int64_t n_dst, int64_t n_edges, int validate, gnnmp_stream_t stream);
int gnnmp_plan_destroy(gnnmp_graph_t *plan);
/* info[0]=n_src info[1]=n_dst info[2]=n_edges (as given) info[3]=E' (with self loops)
 * info[4]=max in-degree info[5]=number of split (long) rows info[6]=bytes of device memory held */
int gnnmp_plan_info(const gnnmp_graph_t *plan, int64_t info[8]);
/* copy the plan's index arrays into caller device buffers (any may be NULL): int32 rowptr[n_dst+1],
 * col[E'], eid[E'] — bit-exact index outputs, used by the parity tests.  GNNMP_EUNSUPPORTED when E' >= 2^31 (the values do
 * not fit int32): gnnmp_plan_export64 hands out rowptr widened to int64 and eid as the unsigned values they are, at any size. */
int gnnmp_plan_export(const gnnmp_graph_t *plan, int32_t *rowptr, int32_t *col, int32_t *eid,
                      gnnmp_stream_t stream);
int gnnmp_plan_export64(const gnnmp_graph_t *plan, int64_t *rowptr, int32_t *col, uint32_t *eid,
                        gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The plan of a BATCH without a sort (csrc/plan_batch.hip).  MLUtils.batch(gs) (GNNGraphs/src/transform.jl:682-709) offsets the
 * members' node ids by the nodes before them and concatenates their edge lists member after member, so the batch's dst-sorted stable
 * CSR is the concatenation of the members' CSRs (rowptr pieces shifted by the slots before them, col by the nodes, edge positions by the
 * edges; plan-added self loops stay at E_batch + node).  The reference's graph-classification loop makes a new batch EVERY step
 * (GraphNeuralNetworks/examples/graph_classification_tudataset.jl:70-71 DataLoader(...; shuffle = true, collate = true), :97-104): these
 * entry points are per-step work — two launches, NO host synchronisation, index arrays from a stream-ordered pool of device blocks — and
 * the result is bit-identical to gnnmp_plan_create on the batched COO (tests/test_plan_batch.py).
 *
 *   gnnmp_plan_concat  members as plan handles, in batch order (what a `batch(gs)` override holds: one cached plan per member graph;
 *                      all square, all with or all without added self loops).  The member table (48 bytes per member) is built on the
 *                      host and uploaded with the call.
 *   gnnmp_plan_select  members as graph ids (ids[k], idx_bytes / index_base like every index array) into ONE resident plan of the whole
 *                      dataset batched once — `dataset` = the plan of MLUtils.batch(all graphs), node_ptr[n_graphs + 1] (device, int64)
 *                      its node offsets — in the order of `ids`: the plan of batch(gs[ids]), which is also what getobs / getgraph of a
 *                      batched graph returns for ascending ids (GNNGraphs/src/gnngraph.jl:311, transform.jl:827-876).  Nothing crosses
 *                      PCIe per step.  The dataset's edge list must be member-major (what batch produces).  n_rows / n_slots: the
 *                      batch's node count and its E' (edges + added self loops if the dataset plan has them), which the caller knows
 *                      on the host (every GNNGraph carries num_nodes / num_edges as host integers); they size the allocation.  If they
 *                      disagree with the selected members the plan comes out EMPTY (rowptr = 0) and gnnmp_plan_status reports it.
 *   Optional outputs (device, NULL to skip): seg_ptr_out int64[k + 1] the batch's node offsets (the seg_ptr of gnnmp_graphconv_chain_f32
 *   / gnnmp_segment_pool_ptr_f32), node_map_out int32[n_rows] the dataset row of every batch row (0-based: the `nmap` of getgraph; feed
 *   it to gnnmp_gather_f32 to collate node features), graph_indicator_out [n_rows] in the index type of `ids` (concat: idx_bytes /
 *   index_base arguments).
 * Rows longer than the new plan's long-row threshold (hubs) need the chunk tables: only then the call synchronises the stream.
 * gnnmp_plan_info's max in-degree of a selected plan is the dataset's (an upper bound).
 *
 *   gnnmp_plan_release  stream-ordered destroy: the plan's block returns to the pool behind the work enqueued on `stream` so far; the next
 *                      pooled object created on ANY stream waits for that work on the device (no host synchronisation).  The caller
 *                      promises that no later work uses the plan.  gnnmp_plan_destroy works on these plans too (the block is then handed
 *                      out again only after a device-wide synchronisation).
 *   gnnmp_plan_edge_index  s, t of the plan's graph in original edge order (plan-added self loops skipped): the batch's COO for callers
 *                      that want it (out_src / out_dst: n_edges entries of idx_bytes each).
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_plan_concat(gnnmp_graph_t **out, const gnnmp_graph_t *const *members, int64_t k, int64_t *seg_ptr_out,
                      void *graph_indicator_out, int idx_bytes, int index_base, gnnmp_stream_t stream);
int gnnmp_plan_select(gnnmp_graph_t **out, const gnnmp_graph_t *dataset, const int64_t *node_ptr, int64_t n_graphs, const void *ids,
                      int idx_bytes, int index_base, int64_t k, int64_t n_rows, int64_t n_slots, int64_t *seg_ptr_out,
                      int32_t *node_map_out, void *graph_indicator_out, gnnmp_stream_t stream);
int gnnmp_plan_release(gnnmp_graph_t *plan, gnnmp_stream_t stream);
int gnnmp_plan_status(const gnnmp_graph_t *plan, gnnmp_stream_t stream);   /* synchronises `stream`; 0 = the member table matched */
int gnnmp_plan_edge_index(const gnnmp_graph_t *plan, int idx_bytes, int index_base, void *out_src, void *out_dst,
                          gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * A placement-aware arena for the OUTPUTS of the gather kernels (csrc/arena.hip) — optional: the rule "the caller allocates" stands,
 * this is an allocator the caller may use.  Measured on MI355X (profiles/README.md, round 4): device memory falls into three placement
 * classes of 96 GiB of physical HBM each; a kernel that gathers ~27 random rows per row it writes (propagate, the fused GCN layer, the
 * one-pass attention) runs 6 % slower when the gathered matrix and the output lie in the SAME class than when they lie in two — and
 * hipMalloc pairs buffers by luck.  The arena allocates 2 GiB blocks one by one (plain hipMalloc), classifies each where it lies with a
 * 0.6 ms probe (a propagate over a synthetic random graph, timed with the block as its output at four places: 10 % apart between the two
 * cases) and keeps blocks of n_classes = 2 or 3 different classes (three let a pipeline keep the PREVIOUS kernel's output — whose dirty
 * lines are still being written back — out of the class the next kernel gathers from); the other blocks are freed:
 *   gnnmp_arena_create(&a, bytes_per_class, n_classes, max_probe_bytes, stream)   bytes_per_class rounded up to 2 GiB blocks.  Creation works
 *                        inside a BUDGET: at most max_probe_bytes (<= 0: 32 GiB) of blocks held at any moment while the classes are being
 *                        looked for (they come in runs of GiB to tens of GiB) and ~0.3 s (env GNNMP_ARENA_BUDGET_MS); a block is screened at
 *                        one window, the kept ones checked at four.  Out of budget is NOT an error: the arena comes back with the classes
 *                        it found — info[7] = 0 .. n_classes of them, info[9] = 1 — and the caller places what it can (two classes still
 *                        separate a gather's source from its output; fewer: allocate as usual).  Synchronises.
 *   gnnmp_arena_class_of(a, ptr, bytes, &cls, stream)   arena memory: its range (0 .. classes - 1), no launch.  Foreign memory (bytes >= 128 MiB):
 *                        the probe with `ptr` as the gathered matrix and the output in every range — c = it shares range c's class,
 *                        classes = none of them / mixed / too small to tell / no room left for the probe's output; synchronises
 *   gnnmp_arena_alloc(a, cls, bytes, &ptr)   bump allocation (4 KiB aligned) inside one block of class cls; a buffer LARGER than a block
 *                        (SAGEConv's 2.5 GB output on the products shape) is an allocation of its own, classified where it lies when it
 *                        is asked for (every 512 MiB window probed, all must agree; up to three tries; SYNCHRONISES THE DEVICE and holds the
 *                        arena's lock meanwhile — set-up work for a persistent buffer, tens of milliseconds: the arena is NOT a compute
 *                        entry point and the "never synchronise" rule above does not cover it; a layer that asks for a placed output pays
 *                        this on its first call only.  At most one unused buffer of another class is retained as a spare, the other
 *                        misses are freed before the call returns); GNNMP_EALLOC when the class cannot serve it (the caller then
 *                        allocates as usual)
 *   gnnmp_arena_reset(a)  forget every allocation (the caller knows nothing uses them any more)
 *   gnnmp_arena_info      info[14]: [0] bytes per class asked for [1], [2], [8] bytes used in range 0 / 1 / 2 [3] blocks created while classifying
 *                         [4] released again [5], [6] the probe's microseconds with source and output in one class / in two [7] classes held
 *                         [9] 1 = gave up on the budget [10], [11], [12] blocks held per range [13] microseconds creation took
 * ALIASING: the arena is a bump allocator — memory it hands out stays handed out until gnnmp_arena_reset.  A caller that keeps a layer's
 * output buffer in the arena and passes the SAME buffer to the layer's next call (gnnmp/placement.py's persistent outputs, the static
 * outputs of a captured graph) overwrites the previous result; a caller that needs both alive allocates both.
 * Usage: cls = class_of(gathered matrix); out = alloc(any range != cls).  Results do not depend on where buffers lie.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gnnmp_arena gnnmp_arena_t;
int gnnmp_arena_create(gnnmp_arena_t **out, int64_t bytes_per_class, int n_classes, int64_t max_probe_bytes, gnnmp_stream_t stream);
int gnnmp_arena_destroy(gnnmp_arena_t *arena);
int gnnmp_arena_alloc(gnnmp_arena_t *arena, int cls, int64_t bytes, void **ptr);
int gnnmp_arena_reset(gnnmp_arena_t *arena);
int gnnmp_arena_class_of(gnnmp_arena_t *arena, const void *ptr, int64_t bytes, int *cls, gnnmp_stream_t stream);
int gnnmp_arena_info(const gnnmp_arena_t *arena, int64_t *info);

/* ------------------------------------------------------------------------------------------------
 * Index ops (bit-exact)
 * ---------------------------------------------------------------------------------------------- */
/* add_self_loops(g::GNNGraph{COO}) — GNNGraphs/src/transform.jl:12-28.
 * out_src/out_dst have n_edges + n entries of the same width/base as the inputs; w / out_w may both be
 * NULL (unweighted) or both non-NULL (appended weights are 1). */
int gnnmp_add_self_loops(const void *src, const void *dst, int idx_bytes, int index_base,
                         int64_t n_edges, int64_t n, void *out_src, void *out_dst, const float *w,
                         float *out_w, gnnmp_stream_t stream);

/* MLUtils.batch(::Vector{GNNGraph{COO}}) index part — GNNGraphs/src/transform.jl:682-709.
 * src/dst hold the member graphs' edge indices concatenated (local numbering); edge_ptr[G+1] and
 * node_ptr[G+1] are int64 device arrays of exclusive prefix sums of num_edges / num_nodes.
 * Writes the offset indices (same width/base) and graph_indicator[N] (same width; values
 * index_base .. index_base+G-1, i.e. 1..G for Julia). */
int gnnmp_batch_coo(const void *src, const void *dst, int idx_bytes, int index_base,
                    const int64_t *edge_ptr, const int64_t *node_ptr, int64_t n_graphs,
                    void *out_src, void *out_dst, void *graph_indicator, gnnmp_stream_t stream);

/* sort_edge_index(u, v) — GNNGraphs/src/utils.jl:30-45: the pairs (u_k, v_k) sorted lexicographically (by u, then v),
 * one common permutation.  64-bit radix sort of the packed pairs on the device (the reference's CUDA extension copies
 * the index to the CPU for this, GNNGraphsCUDAExt.jl:24-30).  Indices must fit 32 bits (GNNMP_EBOUNDS otherwise).
 * Synchronises the stream (graph prep). */
int gnnmp_sort_edge_index(const void *u, const void *v, int idx_bytes, int index_base, int64_t n_edges,
                          void *u_out, void *v_out, gnnmp_stream_t stream);
/* is_bidirected(g) — GNNGraphs/src/query.jl:553-558: sort_edge_index(s, t) == sort_edge_index(t, s); *result = 0 | 1 (host). */
int gnnmp_is_bidirected(const void *s, const void *t, int idx_bytes, int index_base, int64_t n_edges,
                        int *result, gnnmp_stream_t stream);
/* has_self_loops(g) — GNNGraphs/src/query.jl:565-569: any(s .== t); *result = 0 | 1 (host). */
int gnnmp_has_self_loops(const void *s, const void *t, int idx_bytes, int64_t n_edges, int *result,
                         gnnmp_stream_t stream);
/* The edge selection of sample_neighbors(g, nodes, K; dir, replace) — GNNGraphs/src/sampling.jl:68-83.  For seed i
 * (nodes[i], in the plan's destination numbering) with d incoming edges, draws k_i = (K > 0 ? (replace ? K : min(d, K))
 * : d) of them (0 if d = 0): without replacement every k-subset is equally likely (selection sampling; the chosen edges
 * keep their original order), with replacement k independent uniform picks (replace = 2: min(d, K) picks with
 * replacement, what NeighborLoader's `rand(neighbors, min(K, d))` draws, samplers.jl:56-62).  dir = :in uses the graph's plan, dir = :out
 * a plan of the reversed edge index.  Outputs: offsets[n_nodes + 1] (device, int64, exclusive prefix sums of k_i),
 * eids_out[offsets[n_nodes]] = original edge positions (index width / base of `nodes`), *total (host).  capacity =
 * entries available in eids_out; if too small GNNMP_EINVAL is returned with *total and offsets valid.  The generator is
 * counter based (seed, seed-node position, draw index): the same call returns the same sample; it is NOT Julia's RNG
 * stream, so parity with the reference is distributional (tests check membership, counts, uniqueness, uniformity).
 * Synchronises the stream. */
int gnnmp_sample_neighbors(gnnmp_graph_t *plan, const void *nodes, int idx_bytes, int index_base, int64_t n_nodes,
                           int64_t K, int replace, uint64_t seed, int64_t *offsets, void *eids_out,
                           int64_t capacity, int64_t *total, gnnmp_stream_t stream);

/* Ordered node sets on the device (the `Set` / `Dict(node => i)` bookkeeping of NeighborLoader, samplers.jl:78-99, and of
 * induced_subgraph, sampling.jl:178).  map[n_nodes] (int32, zero-initialised by the caller) holds 0 for "absent" and the
 * 1-based list position otherwise; first[n_nodes] is int32 scratch.  Appends to the set the candidates that are not in
 * it yet, each once, in order of first occurrence: writes them to list_out[0 .. *n_new) (index width / base of `cand`)
 * and map[v] = set_size + position + 1.  Deterministic (no dependence on thread scheduling).  Synchronises the stream. */
int gnnmp_unique_append(int32_t *map, int32_t *first, int64_t n_nodes, const void *cand, int idx_bytes,
                        int index_base, int64_t n_cand, int64_t set_size, void *list_out, int64_t *n_new,
                        gnnmp_stream_t stream);
/* induced_subgraph(graph, nodes) — GNNGraphs/src/sampling.jl:173-203: for every listed node i (in list order) its incoming
 * edges (in edge order) whose source is listed too, relabelled by list position: s_out = map[source], t_out = position of
 * i, eid_out = the edge's position in g.  `map` as built by gnnmp_unique_append from the same list (nodes must be valid
 * and distinct).  offsets[n_nodes + 1] (device int64) = first output edge of every listed node; *total (host) = edge
 * count; capacity as in gnnmp_sample_neighbors (capacity = 0 with NULL outputs = count-only call).  NB the reference records `findfirst` of the (source, target) pair as the
 * edge index, i.e. the FIRST parallel edge for all copies of a multi-edge; this returns each copy's own position. */
int gnnmp_induced_subgraph(gnnmp_graph_t *plan, const int32_t *map, const void *nodes, int idx_bytes,
                           int index_base, int64_t n_nodes, int64_t *offsets, void *s_out, void *t_out,
                           void *eid_out, int64_t capacity, int64_t *total, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Leaf ops: _gather / _scatter  (GNNGraphs/src/gatherscatter.jl:4,12-18)
 * ---------------------------------------------------------------------------------------------- */
/* out[k][:] = x[idx[k]][:]  for k in [0,K)  — NNlib.gather; pure copy, bit-exact. */
int gnnmp_gather_f32(const float *x, const void *idx, int idx_bytes, int index_base, int64_t K,
                     float *out, int64_t D, gnnmp_stream_t stream);
/* out[k][:] = xi[t_k][:] - xj[s_k][:]  (xj_minus_xi = 0: xi_sub_xj) or xj[s_k][:] - xi[t_k][:] (1: xj_sub_xi) for every
 * edge k — apply_edges with those message functions (GNNlib/src/msgpass.jl:177-185) without materialising the two
 * gathered (D, E) arrays.  (xi_dot_xj, :172, is gnnmp_edge_dot_f32 below.) */
int gnnmp_edge_sub_f32(const float *xi, const float *xj, const void *s, const void *t, int idx_bytes, int index_base,
                       int64_t K, int xj_minus_xi, float *out, int64_t D, gnnmp_stream_t stream);
/* out[i][:] = aggr_{k : t_k = i, in edge order} m[k][:]   — NNlib.scatter(aggr, m, t; dstsize=(D, n_dst))
 * with t = the plan's dst.  m is [E'][D] in ORIGINAL edge order (self-loop rows last). */
int gnnmp_scatter_f32(gnnmp_graph_t *plan, int aggr, const float *m, float *out, int64_t D,
                      gnnmp_stream_t stream);
/* What NNlib's KernelAbstractions scatter does on GPU arrays today: one atomic per element, fp32 '+'
 * order not defined.  Kept as the measured comparator and for callers that have no plan.
 * `out` must be pre-filled with the identity by the caller.  aggr in {SUM, MAX, MIN}. */
int gnnmp_scatter_atomic_f32(int aggr, const float *m, const void *idx, int idx_bytes,
                             int index_base, int64_t K, float *out, int64_t D,
                             gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused propagate  (GNNlib/src/msgpass.jl:71-79; fast-path specialisations :215-238)
 *
 *   out[i][:] = scale_dst[i] * aggr_{p in row i} ( w[eid_p] * ( scale_src[col_p] * xj[col_p][:] ) )
 *
 * msg = COPY_XJ ignores w.  w has n_edges entries in ORIGINAL edge order (self loops added by the plan
 * have weight 1, as GNNlib/src/layers/conv.jl:28-33 and transform.jl:22-24 pad).  scale_src / scale_dst
 * (nullable) are the GCN normalisation vectors `cout`, `cin` of GNNlib/src/layers/conv.jl:57-67; every
 * product is rounded separately (no FMA contraction) so the result equals the reference's materialised
 * `xj .* cout'` -> propagate -> `x .* cin'` sequence bit for bit on rows <= GNNMP_LONG_ROW.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_propagate_f32(gnnmp_graph_t *plan, int msg, int aggr, const float *xj, const float *w,
                        const float *scale_src, const float *scale_dst, float *out, int64_t D,
                        gnnmp_stream_t stream);

/* propagate(e_mul_xj, g, aggr; xj, e) with a MATRIX e of size (D, E) — GNNlib/src/msgpass.jl:187-191 (`e .* xj`): every
 * edge carries its own row of D factors, e[k][:] in original edge order.  out[i] = aggr_{k: t_k = i} e[k] .* xj[s_k];
 * edges the plan added as self loops weigh 1.  Two rows per edge are fetched (the factor row by edge id, the feature row
 * by source id); the (D, E') product is never written. */
int gnnmp_propagate_emul_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *e, float *out, int64_t D,
                             gnnmp_stream_t stream);

/* The gated message of res_gated_graph_conv (GNNlib/src/layers/conv.jl:287-300):
 *   out[i] = aggr_{k: t_k = i} sigmoid.(gate_i[i] .+ Bx[s_k]) .* Vx[s_k]
 * gate_i = A*x [n_dst][D]; bv_j [n_src][2D] holds every source's Bx (first D) and Vx (last D) side by side — one dense
 * call with the stacked weight [B; V] produces it, and one contiguous 8D-byte gather per edge reads it.  sigmoid is
 * NNlib's formulation (exp(-|x|) based).  The (D, E) gate and message arrays are never written. */
int gnnmp_propagate_gated_f32(gnnmp_graph_t *plan, int aggr, const float *gate_i, const float *bv_j, float *out,
                              int64_t D, gnnmp_stream_t stream);

/* The same fused propagate with the per-edge factors already laid out in the plan's slot order:
 *   w_slot[p]   = w[eid_p]          (1 for plan-added self loops)      — gnnmp_plan_slot_gather_f32(plan, 1, w, ...)
 *   ss_slot[p]  = scale_src[col_p]                                    — gnnmp_plan_slot_gather_f32(plan, 0, scale_src, ...)
 * (either may be NULL).  The products and their order are identical to gnnmp_propagate_f32, hence so are the bits;
 * what changes is the traffic: a coalesced 4 B read per edge instead of a random 4 B gather (one 64 B sector) per
 * edge.  The factors depend only on the graph (GCN's 1/sqrt(deg), the graph's own weights), so a caller computes them
 * once per graph where the reference recomputes `xj .* cout'` on every call (GNNlib/src/layers/conv.jl:57-59). */
/* cg_conv's propagate (GNNlib/src/layers/conv.jl:304-333): out[i] = Σ_{j -> i} sigmoid(f_ij) .* act(s_ij) with
 *   z = vcat(x_i, x_j, e_ij),  f = l.dense_f's W z + b,  s = l.dense_s's W z + b  (act = dense_s's σ, any gnnmp_act)
 * split by the blocks of the two weight matrices: fs_i [n_dst][2D] = the x_i share of (f | s) with the biases folded in,
 * fs_j [n_src][2D] the x_j share, fs_e [n_edges][2D] the edge share in original edge order (NULL without edge features).
 * One pass over the edges; the (2D, E) pre-activations of the reference are never built. */
int gnnmp_propagate_cg_f32(gnnmp_graph_t *plan, const float *fs_i, const float *fs_j, const float *fs_e, int act,
                           float *out, int64_t D, gnnmp_stream_t stream);
/* nn_conv's propagate (GNNlib/src/layers/conv.jl:260-273): out[i] = aggr_{k: j -> i} W_k x_j with one (Dout, Din) matrix per
 * edge, W_k = reshape(l.nn(e)[:, k], Dout, Din) — `we` is nn(e) as the reference holds it, [n_edges][Dout * Din] in
 * original edge order, element (o, c) of edge k at we[k][o + Dout * c].  The (Dout, E) message array is never built. */
int gnnmp_propagate_nn_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *we, float *out, int64_t Din,
                           int64_t Dout, gnnmp_stream_t stream);
int gnnmp_propagate_slots_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *w_slot,
                              const float *ss_slot, const float *scale_dst, float *out, int64_t D,
                              gnnmp_stream_t stream);
/* the same with the layer epilogue of gcn_conv's W-first branch in the row kernel: out = act.(A .+ bias), A = what
 * gnnmp_propagate_slots_f32 returns — `x = l.weight * x` before the convolution when Dout < Din, then `σ.(x .+ l.bias)` after it
 * (GNNlib/src/layers/conv.jl:36-40,71).  bias: [D] or NULL; act: GNNMP_ACT_IDENTITY | GNNMP_ACT_RELU.  Bit-identical to
 * gnnmp_propagate_slots_f32 followed by gnnmp_bias_act_f32, one pass over (N, D) less. */
int gnnmp_propagate_slots_act_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *w_slot, const float *ss_slot,
                                  const float *scale_dst, const float *bias, int act, float *out, int64_t D,
                                  gnnmp_stream_t stream);
/* out_slot[p] = v[col_p] (by = 0: v is a node vector of n_src entries) or v[eid_p] (by = 1: v is an edge vector of
 * n_edges entries in original order; plan-added self loops get 1.0).  out_slot has E' entries. */
int gnnmp_plan_slot_gather_f32(gnnmp_graph_t *plan, int by, const float *v, float *out_slot,
                               gnnmp_stream_t stream);

/* degree(g, Float32; dir = :in, edge_weight) — GNNGraphs/src/query.jl:314-331,355-369.
 * w == NULL: counts (exact integers).  w != NULL: n_edges weights in original order, summed in edge
 * order; plan-added self loops count 1. */
int gnnmp_degree_f32(gnnmp_graph_t *plan, const float *w, float *deg, gnnmp_stream_t stream);

/* out[i] = 1/sqrt(deg[i])  — the default norm_fn of GCNConv (GraphNeuralNetworks/src/layers/conv.jl:99);
 * evaluated as the correctly rounded sqrt then the correctly rounded division, like Julia's broadcast. */
int gnnmp_inv_sqrt_f32(const float *deg, float *out, int64_t n, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * softmax_edge_neighbors(g, e) — GNNlib/src/utils.jl:84-97.  logits and alpha are [E'][H] in ORIGINAL
 * edge order.  Per destination and channel: max, exp(e - max), sum in edge order, true division.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_edge_softmax_f32(gnnmp_graph_t *plan, const float *logits, float *alpha, int64_t H,
                           gnnmp_stream_t stream);
/* softmax_nodes(g, x) / softmax_edges(g, e) — GNNlib/src/utils.jl:49-72: the same three steps over the segments of a
 * graph indicator.  plan = a plan whose destination index is the indicator (src = 1..K, dst = indicator, n_dst =
 * num_graphs); x, out [K][D] in original order.  den_add is added to the denominator when non-zero: softmax_edges
 * divides by `den .+ eps(T)` (utils.jl:71), softmax_nodes by `den` (utils.jl:57). */
int gnnmp_segment_softmax_f32(gnnmp_graph_t *plan, const float *x, float *out, int64_t D, float den_add,
                              gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GATConv attention path — GNNlib/src/layers/conv.jl:112-167 from `apply_edges` to
 * `aggregate_neighbors`, fused.
 *   Wx       : [N][H][C]  (Julia reshape(dense_x(x), C, H, N))
 *   a        : [H][2C]    (Julia l.a of size (2C, H)); a[h][0:C] pairs with the TARGET Wx_i,
 *                          a[h][C:2C] with the SOURCE Wx_j (order of vcat(Wxi, Wxj), conv.jl:157)
 * gnnmp_gat_node_scores_f32: score_dst[n][h] = sum_c a[h][c]*Wx[n][h][c],
 *                            score_src[n][h] = sum_c a[h][C+c]*Wx[n][h][c]   (either out may be NULL)
 * gnnmp_gat_aggregate_f32  : logit_p[h] = leakyrelu(score_dst[i][h] + score_src[col_p][h], slope)
 *                            alpha = softmax over row i (as gnnmp_edge_softmax_f32)
 *                            out[i][h][:] = sum_p alpha_p[h] * Wx_src[col_p][h][:]   (edge order)
 *   alpha_out (nullable): [E'][H] attention coefficients in original edge order.
 *   bias (nullable, [H*C]) and act: the layer's `σ.(x .+ bias)` (conv.jl:147) fused into the store for the
 *   concat = true case.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_gat_node_scores_f32(const float *Wx, const float *a, float *score_dst, float *score_src,
                              int64_t N, int64_t H, int64_t C, gnnmp_stream_t stream);
int gnnmp_gat_aggregate_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *score_dst,
                            const float *score_src, float negative_slope, const float *bias, int act,
                            float *out, float *alpha_out, int64_t H, int64_t C,
                            gnnmp_stream_t stream);

/* The production entry point for the same GATConv path: ONE pass over the edges, no node pre-pass, no score arrays.
 * The source half of the logit is formed in registers from the source row that the weighted sum fetches anyway, the
 * target half from row i of Wx_dst (NULL = Wx_src), and the neighbourhood softmax is computed online (running max and
 * denominator).  Same mathematical function as gnnmp_gat_node_scores_f32 + gnnmp_gat_aggregate_f32; rounding differs
 * at the 1e-6 level (north_star's bar for fp32 aggregation is 1e-5).  Head widths whose lane count is not a power of
 * two fall back to those two kernels internally (scores in the plan's workspace). */
int gnnmp_gat_conv_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                       float negative_slope, const float *bias, int act, float *out, int64_t H,
                       int64_t C, gnnmp_stream_t stream);

/* gat_conv with edge features (l.dense_e, conv.jl:152-167: `Wxx = vcat(Wxi, Wxj, We)`): the edge's share of the logit,
 * edge_score[k][h] = a[2C:3C, h] . We_k[:, h] with We = dense_e(e), is a per-edge scalar per head — compute it with
 * gnnmp_gat_node_scores_f32 on the (E, H*C) matrix We — and is added inside the one-pass kernel, fetched by original edge
 * position.  `a` is the node part [H][2C].  The plan must not add self loops (conv.jl:119-121 forbids the combination). */
int gnnmp_gat_conv_edge_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                            const float *edge_score, float negative_slope, const float *bias, int act, float *out,
                            int64_t H, int64_t C, gnnmp_stream_t stream);

/* Training forward of the same path: as gnnmp_gat_conv_f32, and additionally saves the neighbourhood-softmax statistics
 * stats[i][h] = (m_i, den_i) — the running maximum of the logits and Σ_j exp(l_ij - m_i) — 8 bytes per destination and
 * head instead of the reference's (H, E') α array that Zygote keeps alive for the pullback.  The feature row must fit one
 * wave (H*C / v lanes <= 64 with v = 4, 2 or 1 floats per lane as C's divisibility allows; GNNMP_EUNSUPPORTED otherwise).
 * Head widths whose lane count is a power of two reduce with DPP butterflies; any other width (C = 7 classes, ...) is
 * supported too, summing the head's lanes one by one. */
int gnnmp_gat_conv_stats_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                             float negative_slope, const float *bias, int act, float *out, float *stats,
                             int64_t H, int64_t C, gnnmp_stream_t stream);

/* Dropout on the attention coefficients — GNNlib/src/layers/conv.jl:139 `α = dropout(α, l.dropout)` (NNlib.dropout: every
 * coefficient kept with probability 1 - p and scaled by 1 / (1 - p); the reference applies it whenever l.dropout > 0).  Inside the
 * one-pass kernel: out_i = Σ_j keep_ij / (1 - p) α_ij Wx_j with α the UNdropped softmax, so the saved statistics (stats, nullable) are
 * those of gnnmp_gat_conv_stats_f32.  No mask is stored: keep_ij[h] is a pure function of (seed, e, h) that the forward, both
 * backward passes and a host restatement share —
 *     mix(x):  x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16     (32-bit, wrapping)
 *     keep  =  mix( mix(e ^ lo32(seed)) ^ (h * 0x9e3779b9 + hi32(seed)) )  >=  floor(p * 2^32)
 * with e = the edge's 0-based position in the caller's edge list (plan-added self loops: E + node, the order add_self_loops appends
 * them in) and h the 0-based head.  0 <= p < 1; p = 0 is gnnmp_gat_conv_stats_f32 / gnnmp_gat_conv_f32 exactly.  The feature row must
 * fit one wave and there are no edge features (GNNMP_EUNSUPPORTED otherwise).  A fresh seed per call gives the reference's behaviour;
 * the same seed must be handed to gnnmp_gat_conv_grad_drop_f32.
 * gnnmp_dropout_keep_u8 writes that mask, keep[e][h] in {0, 1}, for n_edges edge positions (tests, or a caller that wants α .* mask). */
/* The TRAINING forward of GATConv and its pullback with ONE edge pass (round 4).  With o_i = Σ_j α_ij Wx_j the softmax rrule's
 * D_i = Σ_j α_ij (Δ_i . Wx_j) is Δ_i . o_i, and dsd_i = Σ_j α_ij (g_ij - D_i) lrelu'(z_ij) = (1 - slope) (Δ_i . o+_i - D_i P_i) with
 * o+_i / P_i the same sums restricted to the edges whose logit z_ij is positive — so the forward also writes oplus [n_dst][H*C] and pplus
 * [n_dst][H] (4 bytes per feature and node more), and the destination side of the pullback becomes a node kernel on (dout, out, oplus,
 * pplus): the gather of every Wx_j by destination (5.7 ms on the products shape) is gone; the source-side pass, dWx, dss, da are those of
 * gnnmp_gat_conv_grad_f32.  `out` / `bias` of grad2 are the forward's output act(o + bias) and its bias (act = identity | relu): o is read
 * back as out - bias, which is exact wherever dout is not zero when dout = dL/d(o + bias) (relu's switched-off entries carry dout = 0).
 * No dropout, no edge features (those keep gnnmp_gat_conv_stats_f32 / gnnmp_gat_conv_grad_f32). */
int gnnmp_gat_conv_train_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a, float negative_slope,
                             const float *bias, int act, float *out, float *stats, float *oplus, float *pplus, int64_t H, int64_t C,
                             gnnmp_stream_t stream);
int gnnmp_gat_conv_grad2_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src, const float *Wx_dst, const float *a,
                             float negative_slope, const float *stats, const float *out, const float *bias, const float *oplus,
                             const float *pplus, const float *dout, float *line, float *dsd, float *dss, float *dWx_src, float *dWx_dst,
                             float *da, int64_t H, int64_t C, gnnmp_stream_t stream);
int gnnmp_gat_conv_drop_f32(gnnmp_graph_t *plan, const float *Wx_src, const float *Wx_dst, const float *a,
                            float negative_slope, float p, uint64_t seed, const float *bias, int act, float *out,
                            float *stats, int64_t H, int64_t C, gnnmp_stream_t stream);
int gnnmp_dropout_keep_u8(uint64_t seed, float p, int64_t n_edges, int64_t H, uint8_t *keep, gnnmp_stream_t stream);

/* The same one-pass kernel for the other attention layers that share the path: GATv2Conv
 * (gatv2_conv, conv.jl:171-214), TransformerConv's attention core (transformer_conv, conv.jl:553-616) and AGNNConv
 * (agnn_conv, conv.jl:337-352).  out[i][h][:] = Σ_j softmax_{j in N(i)}(l_ij) V_j[h][:], then + bias and act.
 *   Q [n_dst][H*C] (NULL = K), K [n_src][H*C], V [n_src][H*C] (NULL = K; a separate V only with GNNMP_ATTN_DOT)
 *   a: GAT [H][2C], GATV2 [H][C] (Julia (C, H) as stored), otherwise ignored
 *   scale: DOT divides the dot product by it (l.sqrt_out); COS multiplies the cosine by it (l.β); else ignored
 *   stats: optional [n_dst][H][2] softmax statistics as in gnnmp_gat_conv_stats_f32
 * The feature row must fit one wave (see gnnmp_gat_conv_stats_f32); GNNMP_EUNSUPPORTED otherwise (mode GAT without stats
 * falls back to the three-pass kernels instead). */
int gnnmp_attn_conv_f32(gnnmp_graph_t *plan, int mode, const float *Q, const float *K, const float *V,
                        const float *a, float negative_slope, float scale, const float *bias, int act,
                        float *out, float *stats, int64_t H, int64_t C, gnnmp_stream_t stream);

/* Pullback of the attention path (what Zygote composes from the rrules of gather / leakyrelu / softmax_edge_neighbors /
 * scatter(+) for conv.jl:136-141,152-167).  dout = Δ w.r.t. the aggregated (pre-bias, pre-σ) output [n_dst][H*C];
 * plan_t = plan of the reversed edge index (same self-loop flag); stats from gnnmp_gat_conv_stats_f32.
 * Scratch/outputs supplied by the caller: line [n_dst][H][4] (16-byte aligned), dsd [n_dst][H] (Δ of the target logit
 * half), dss [n_src][H] (Δ of the source half).  Results: dWx_src [n_src][H*C]; dWx_dst [n_dst][H*C] for a bipartite
 * layer (Wx_dst != Wx_src) — with Wx_dst NULL / == Wx_src pass dWx_dst = NULL and dWx_src holds the whole ΔWx;
 * da [H][2C] (may be NULL).  Two passes over the edges, no atomics, run-to-run identical. */
int gnnmp_gat_conv_grad_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src, const float *Wx_dst,
                            const float *a, float negative_slope, const float *stats, const float *dout,
                            float *line, float *dsd, float *dss, float *dWx_src, float *dWx_dst, float *da,
                            int64_t H, int64_t C, gnnmp_stream_t stream);

/* The same pullback through the dropped coefficients of gnnmp_gat_conv_drop_f32 (same p and seed as the forward): with
 * k_ij = keep_ij / (1 - p), g_ij = k_ij (Δ_i . Wx_j) replaces Δ_i . Wx_j and Σ_i k_ij α_ij Δ_i replaces Σ_i α_ij Δ_i; everything else,
 * including the caller-supplied buffers, is as in gnnmp_gat_conv_grad_f32. */
int gnnmp_gat_conv_grad_drop_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, const float *Wx_src, const float *Wx_dst,
                                 const float *a, float negative_slope, float p, uint64_t seed, const float *stats,
                                 const float *dout, float *line, float *dsd, float *dss, float *dWx_src, float *dWx_dst,
                                 float *da, int64_t H, int64_t C, gnnmp_stream_t stream);

/* gnnmp_attn_conv_f32 with dropout on the attention coefficients — gatv2_conv's `α = dropout(α, l.dropout)` (conv.jl:191); same mask
 * function, same rules as gnnmp_gat_conv_drop_f32 (mode GNNMP_ATTN_GAT or GNNMP_ATTN_GATV2; GNNMP_EUNSUPPORTED for the others, which
 * have no dropout in the reference), and its pullback for the GATV2 logit (same p and seed; arguments as gnnmp_attn_conv_grad_f32). */
int gnnmp_attn_conv_drop_f32(gnnmp_graph_t *plan, int mode, const float *Q, const float *K, const float *V, const float *a,
                             float negative_slope, float scale, float p, uint64_t seed, const float *bias, int act, float *out,
                             float *stats, int64_t H, int64_t C, gnnmp_stream_t stream);
int gnnmp_attn_conv_grad_drop_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, int mode, const float *Q, const float *K,
                                  const float *V, const float *a, float negative_slope, float scale, float p, uint64_t seed,
                                  const float *stats, const float *dout, float *line, float *dQ, float *dK, float *dV, float *dA,
                                  float *da, int64_t H, int64_t C, gnnmp_stream_t stream);

/* Pullback of gnnmp_attn_conv_f32 for the GATV2 and DOT logits (the GAT logit has the cheaper dedicated entry above; the
 * cosine logit has none yet: GNNMP_EUNSUPPORTED).  dout = Δ w.r.t. the aggregated (pre-bias, pre-σ) output; stats from the
 * forward; plan_t = plan of the reversed edge index.  Caller-supplied: line [n_dst][H][4] (16-byte aligned scratch),
 * dQ [n_dst][H*C], dK [n_src][H*C]; DOT also dV [n_src][H*C]; GATV2 also dA [n_dst][H*C] (scratch: per-destination terms
 * of Δa) and da [H][C] (may be NULL).  With GATV2, V is K and dK holds the whole gradient of K.  Two edge passes, no
 * atomics, run-to-run identical. */
int gnnmp_attn_conv_grad_f32(gnnmp_graph_t *plan, gnnmp_graph_t *plan_t, int mode, const float *Q, const float *K,
                             const float *V, const float *a, float negative_slope, float scale, const float *stats,
                             const float *dout, float *line, float *dQ, float *dK, float *dV, float *dA, float *da,
                             int64_t H, int64_t C, gnnmp_stream_t stream);

/* out[n][c] = act( mean_h y[n][h][c] + bias[c] ) — the concat = false tail of gat_conv (`mean(x, dims = 2)`,
 * GNNlib/src/layers/conv.jl:143-147): heads added in order, one division by H, then σ.(x .+ bias). */
int gnnmp_head_mean_f32(const float *y, const float *bias, int act, float *out, int64_t N, int64_t H,
                        int64_t C, gnnmp_stream_t stream);
/* its pullback: dy[n][h][c] = dz[n][c] / H (the heads' gradient of `mean(x, dims = 2)`) */
int gnnmp_head_mean_grad_f32(const float *dz, float *dy, int64_t N, int64_t H, int64_t C, gnnmp_stream_t stream);
/* gmm_conv's mixture weights (GNNlib/src/layers/conv.jl:379-385): out[e][k * C + c] = exp(Σ_d ((e[e][d] - mu[k][d])^2 / 2) *
 * sigma_inv[k][d]^2) for every edge, kernel k and (repeated) output channel c — the factor array of the layer's
 * `propagate(e_mul_xj, g, mean; xj = reshape(dense_x(x), out, K, :), e = w)`, ready for gnnmp_propagate_emul_f32.
 * mu, sigma_inv: Julia (ein, K) column-major = [K][ein]. */
int gnnmp_gmm_weights_f32(const float *e, const float *mu, const float *sigma_inv, float *out, int64_t E, int64_t ein,
                          int64_t K, int64_t C, gnnmp_stream_t stream);
/* egnn_conv's radial features (GNNlib/src/layers/conv.jl:465-467): sq[k] = sum(x_diff[k][:] .^ 2) and
 * xn[k][:] = x_diff[k][:] / (sqrt(sq[k]) + eps) for every row k (eps = 1f-6 in the reference). */
int gnnmp_row_sqnorm_normalize_f32(const float *x, float *sq, float *xn, float eps, int64_t N, int64_t D,
                                   gnnmp_stream_t stream);
/* Flux.LSTMCell's pointwise part — the cell of set2set_pool (GNNlib/src/layers/pool.jl:31-44): gx = Wi x and gh = Wh h are
 * [N][4D] (gates input, forget, cell, output), b [4D] or NULL:
 *   c_out = σ(forget) .* c + σ(input) .* tanh(cell),  h_out = σ(output) .* tanh(c_out) */
int gnnmp_lstm_pointwise_f32(const float *gx, const float *gh, const float *b, const float *c, float *h_out, float *c_out,
                             int64_t N, int64_t D, gnnmp_stream_t stream);
/* out[n] = sum(a[n][:] .* b[n][:]) — `sum(qn .* x, dims = 1)` of set2set_pool (pool.jl:39) */
int gnnmp_rowdot_f32(const float *a, const float *b, float *out, int64_t N, int64_t D, gnnmp_stream_t stream);
/* Flux.GRUCell's pointwise part — the cell of gated_graph_conv (GNNlib/src/layers/conv.jl:228-232): gx = Wi m and
 * gh = Wh h are [N][3D] (gates r, z, candidate), b [3D] or NULL:
 *   r = σ(gx_r + gh_r + b_r), z = σ(gx_z + gh_z + b_z), h~ = tanh(gx_n + r .* gh_n + b_n), out = (1 - z) .* h~ + z .* h */
int gnnmp_gru_pointwise_f32(const float *gx, const float *gh, const float *b, const float *h, float *out, int64_t N,
                            int64_t D, gnnmp_stream_t stream);
/* xn[n][:] = x[n][:] / sqrt(sum(x[n][:]^2)) — `xn = x ./ sqrt.(sum(x .^ 2, dims = 1))` of agnn_conv
 * (GNNlib/src/layers/conv.jl:341-342); rnorm (optional) keeps the norms for the pullback.  A zero row gives NaN like the
 * reference's 0 / 0. */
int gnnmp_row_normalize_f32(const float *x, float *xn, float *rnorm, int64_t N, int64_t D, gnnmp_stream_t stream);
/* its pullback for the two uses of xn in the attention (query and key cotangents dq, dk; either may be NULL):
 *   dn = dq + dk;  dx = base + (dn - xn (xn . dn)) / rnorm   (base optional: another cotangent of x, e.g. the value's);
 *   qdot[n] = qscale (xn[n] . dq[n]) (optional): with logit = β (xn_i . xn_j) and qscale = 1 / β, Σ_n qdot[n] = dL/dβ. */
int gnnmp_row_normalize_grad_f32(const float *dq, const float *dk, const float *xn, const float *rnorm,
                                 const float *base, float *dx, float *qdot, float qscale, int64_t N, int64_t D,
                                 gnnmp_stream_t stream);
/* out = a + b (n floats) — degree(g; dir = :both) = out-degree + in-degree (GNNGraphs/src/query.jl:362-367). */
int gnnmp_add_f32(const float *a, const float *b, float *out, int64_t n, gnnmp_stream_t stream);
/* out[n][:] = a[n][:] .* b[n][:] with a of 1 channel (broadcast) or D channels — `α .* l.ffeat(x)` of
 * global_attention_pool, GNNlib/src/layers/pool.jl:6-10 (out may alias b) */
int gnnmp_mul_rows_f32(const float *a, int64_t Da, const float *b, float *out, int64_t N, int64_t D,
                       gnnmp_stream_t stream);
/* out = alpha .* x .+ y (out may alias x or y) — `(1 .+ ϵ) .* xi .+ m` of gin_conv, GNNlib/src/layers/conv.jl:250-256 */
int gnnmp_axpy_f32(float alpha, const float *x, const float *y, float *out, int64_t n, gnnmp_stream_t stream);
/* *result_host = 1 iff idx[0..n) is non-decreasing (is a graph_indicator one that `batch` could have built?).
 * Synchronises the stream (graph prep). */
int gnnmp_is_sorted(const void *idx, int idx_bytes, int64_t n, int *result_host, gnnmp_stream_t stream);

/* out[n][:] = act(x[n][:] + bias[:])  — the `σ.(x .+ bias)` tail of a layer body when it is not fused into
 * the producing kernel (GNNlib/src/layers/conv.jl:71,147).  bias nullable; out may alias x. */
int gnnmp_bias_act_f32(const float *x, const float *bias, int act, float *out, int64_t N, int64_t D,
                       gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * reduce_nodes(aggr, g, x) / global_pool — GNNlib/src/utils.jl:12-16, GNNlib/src/layers/pool.jl:3-5,
 * for a SORTED graph_indicator (what MLUtils.batch builds, transform.jl:691-699): contiguous segments.
 * seg_ids[N] has values index_base .. index_base+G-1, non-decreasing.  out is [G][D].
 * (An unsorted indicator goes through gnnmp_plan_create(src = 1..N, dst = indicator) + gnnmp_scatter_f32.)
 * One lane group reduces one graph: right for a batch of many small graphs.  For FEW LARGE graphs (a whole-graph
 * readout is G = 1) use the same plan route — it cuts large segments into balanced chunks (N = 2.4 M, G = 1: 283 ms
 * here against 0.3 ms through the plan); the host mirror switches at N > 256 * G.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_segment_pool_f32(int aggr, const float *x, const void *seg_ids, int idx_bytes,
                           int index_base, float *out, int64_t D, int64_t N, int64_t G,
                           gnnmp_stream_t stream);
/* The same with the segment boundaries precomputed — a constant of the batched graph, like its plan:
 *   gnnmp_segment_bounds   seg_ptr[k] = first node of graph k (k = 0..G; seg_ptr[G] = N), from the sorted indicator, one pass
 *   gnnmp_segment_pool_ptr_f32   the pooling itself: no per-call binary search of the indicator (two 18-step chains of
 *                          dependent loads per graph at N = 245 k: half the kernel's time on the batched config) */
int gnnmp_segment_bounds(const void *seg_ids, int idx_bytes, int index_base, int64_t N, int64_t G, int64_t *seg_ptr,
                         gnnmp_stream_t stream);
int gnnmp_segment_pool_ptr_f32(int aggr, const float *x, const int64_t *seg_ptr, float *out, int64_t D, int64_t N, int64_t G,
                               gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense feature contraction of the layer bodies (the only MFMA work on the path):
 *   out[n][:] = act( W1 * x1[n][:]  (+ W2 * x2[n][:])  (+ bias) )
 * W1 is [Dout][D1], W2 [Dout][D2] row-major (Julia weight' is NOT taken: Julia's (Dout, Din) column-major
 * weight is C row-major [Din][Dout]; pass w_layout = 1 for that, 0 for C row-major [Dout][Din]).
 * Covers `weight * x` (conv.jl:39,69), `weight1*xi .+ weight2*m` (conv.jl:106),
 * `weight * vcat(xi, m)` (conv.jl:281; W1 = first Din columns, W2 = last Din, given by ldw) and
 * `dense_x` (conv.jl:127).  fp32 in, fp32 out.  Round 3: every operand is split EXACTLY into three bf16 planes and the six
 * significant plane products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (csrc/msplit.h: error against float64 in the
 * class of an fp32 fma chain, measured lower; non-finite operands are recomputed with fp32 fma loops); shapes whose weight planes
 * do not fit LDS, K % 4 != 0 or Dout % 32 far from full take the fp32 MFMA kernels of rounds 1-2 (v_mfma_f32_16x16x4_f32 /
 * 32x32x2_f32, an exact fp32 fma chain).
 * ldw1/ldw2: leading dimension (elements between consecutive rows of the stored matrix).
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_dense_f32(const float *x1, const float *W1, int64_t D1, int64_t ldw1, const float *x2,
                    const float *W2, int64_t D2, int64_t ldw2, int w_layout, const float *bias,
                    int act, float *out, int64_t N, int64_t Dout, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Graph-parallel step for batched graphs (north_star: "shard by graph across up to 8 GPUs of one node with an
 * RCCL-over-xGMI all-gather of per-shard logits").  A batched GNNGraph is block-diagonal (MLUtils.batch,
 * GNNGraphs/src/transform.jl:682-709): member graphs are independent units.  One process per GPU:
 *   1. gnnmp_shard_by_size (HOST arrays): deal the member graphs to the ranks — by decreasing size in a boustrophedon: counts +-1,
 *      node totals within a fraction of a per cent; rank_of[g] = owner, gather_index[g] = row of graph g in the all-gathered block
 *      (rank * gmax + position among the rank's graphs in ascending graph order), *gmax = rows per rank in the collective.
 *   2. every rank batches ITS members (MLUtils.batch), uploads, builds its plan and runs the forward (e.g. gnnmp_graphconv_chain_f32),
 *      writing its (G_r, nout) logits at the start of a (gmax, nout) send buffer (the padding rows are never read back).
 *   3. gnnmp_allgather_f32: ONE ncclAllGather of gmax * nout floats per rank on the caller's communicator and stream (RCCL resolved
 *      with dlopen at the first call; a few KB per rank: latency-bound, no bucketing), then ONE gnnmp_gather_f32 with gather_index
 *      (uploaded once) puts the rows back in graph order.  Weights are replicated; there is no other collective on the path.
 * The host mirror of the same design is gnnmp/parallel.py (torch.distributed, "nccl" = RCCL on ROCm).
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_shard_by_size(const int64_t *sizes, int64_t G, int world, int32_t *rank_of, int64_t *gather_index, int64_t *gmax);
int gnnmp_allgather_f32(void *nccl_comm, const float *send, float *recv, int64_t count, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * A whole graph-classification forward in ONE kernel (BASELINE.json config 5):
 *   GNNChain(GraphConv(d0 => d1, σ1; aggr), ..., GraphConv(d_{L-1} => d_L, σL; aggr), GlobalPool(pool_aggr), Dense(d_L => nout))
 *   model: examples/graph_classification_tudataset.jl:79-82; layer body σ.(W1*x_i .+ W2*aggr_j x_j .+ b) GNNlib/src/layers/conv.jl:102-108;
 *   pooling reduce_nodes(aggr, g, x) GNNlib/src/layers/pool.jl:3-5, utils.jl:12-16; head Flux.Dense.
 * on a batched GNNGraph (MLUtils.batch, GNNGraphs/src/transform.jl:682-709): member graphs are contiguous row ranges and no edge
 * crosses them, so a workgroup owns a run of whole graphs and walks the chain without any inter-workgroup traffic; layer outputs
 * live in `scratch` (L2-resident, never read by another workgroup), the last layer is folded straight into per-row head outputs,
 * and only the (G, nout) result is written for the caller.  Contractions run on the split-bf16 MFMA core (fp32-class accuracy).
 *   plan      : plan of the batched graph's (s, t) WITHOUT self loops
 *   seg_ptr   : DEVICE int64 [G + 1], seg_ptr[k] = first node of member graph k (gnnmp_segment_bounds)
 *   dims      : HOST int64 [n_layers + 1] = d0 .. dL;  W_root / W_agg / bias : HOST arrays of n_layers DEVICE pointers
 *               (weight1 / weight2; bias entries may be NULL, `bias` itself may be NULL);  act : HOST int [n_layers] (gnnmp_act:
 *               IDENTITY | RELU)
 *   w_layout  : like gnnmp_dense_f32 — 0: every weight is C row-major [out][in]; 1: Julia's (out, in) column-major matrices as stored
 *               (weight1, weight2 and the head's weight alike)
 *   aggr, pool_aggr : GNNMP_SUM | GNNMP_MEAN;  W_head (nout, dL) in the same layout, b_head [nout] or NULL
 *   scratch   : DEVICE, at least gnnmp_graphconv_chain_scratch_floats(N, n_layers, dims, nout) floats, 16-byte aligned
 *   out       : DEVICE [G][nout]
 * Envelope: n_layers <= 4, every d a multiple of 4, d1..dL <= 128, nout <= 8; outside it (or with max / min aggregation) the call
 * returns GNNMP_EUNSUPPORTED without touching `out` and the caller runs the chain layer by layer (gnnmp_propagate_f32 +
 * gnnmp_dense_f32 + gnnmp_segment_pool_ptr_f32).  The pooled-then-Dense order of the reference and the Dense-then-pooled order used
 * here agree to rounding (both are linear); the per-row aggregates have the bits of gnnmp_propagate_f32's.
 * ---------------------------------------------------------------------------------------------- */
int64_t gnnmp_graphconv_chain_scratch_floats(int64_t N, int n_layers, const int64_t *dims, int64_t nout);
/* Optional per-batch constant (like the plan): the member graphs packed into jobs of at most 64 rows (best fit decreasing), which lets
 * the chain 16 => 128 => 128 run with a PAIR OF WAVES per group of whole member graphs and no layer output in memory at all
 * (csrc/graph_chain2.hip: layer 1's accumulators are layer 2's operands, the neighbour sums of layer 2 run after its product through
 * a 4 KB LDS stage the two waves share, the two 64-column halves of layer 2 go to different workgroups; each row's W_head * h2 — nout
 * floats per half — is the only intermediate written, and a second small launch pools it per member graph in node order: no
 * floating-point atomic, no memset, run-to-run identical).  The handle owns that intermediate and the list of jobs set aside for the
 * exact fp32 path (non-finite operands), so — like a plan's workspace — it serves ONE stream at a time (and one host thread at a time:
 * the chain call picks the parity of its set-aside counters from a per-handle call count).
 * Built once per batched graph from the DEVICE seg_ptr; synchronises `stream` (graph prep).  A batch with a member graph of more than
 * 64 nodes, or without any, yields a handle without jobs: the chain then runs on the general kernel.  info[0] = jobs, [1] = member
 * graphs, [2] = rows, [3] = largest member graph, [4] = per-mille of the MFMA tiles' rows that are real rows. */
typedef struct gnnmp_chain_jobs gnnmp_chain_jobs_t;
int gnnmp_chain_jobs_create(gnnmp_chain_jobs_t **out, const int64_t *seg_ptr, int64_t G, gnnmp_stream_t stream);
int gnnmp_chain_jobs_destroy(gnnmp_chain_jobs_t *jobs);
int gnnmp_chain_jobs_info(const gnnmp_chain_jobs_t *jobs, int64_t *info);
/* The same packing ON THE DEVICE (csrc/chain_pack.h: best fit decreasing on the histogram of the jobs' free room, one thread block): one
 * launch, no copy of the sizes to the host, no synchronisation — the per-step form for a loop that makes a new batch every step
 * (examples/graph_classification_tudataset.jl:97-104).  The caller passes what it holds on the host (num_nodes of every member graph is a
 * host integer, GNNGraphs/src/gnngraph.jl:108-117): n_rows = the batch's nodes, max_graph = its largest member, has_empty = some member has
 * no node.  max_graph > 64 or has_empty gives a handle without jobs, as above.  If the device finds sizes that contradict the announcement
 * the handle is poisoned: the chain writes NaN logits and gnnmp_chain_jobs_info fails.  gnnmp_chain_jobs_info on such a handle
 * synchronises the device.  A jobs handle serves ONE stream at a time and one host thread at a time (its z rows and set-aside list are
 * scratch of the running call).
 *   gnnmp_chain_jobs_release  stream-ordered destroy (see gnnmp_plan_release)
 *   gnnmp_chain_jobs_export   the job table as the kernel reads it: tab_out[cap_rows][64] int32 (global row of each slot, -1 = empty),
 *                             hdr_out[32] int32 ([0] jobs [1] 32-row tiles [2] poisoned [3] largest member seen [4] empty members,
 *                             [16..29] seven 64-bit phase stamps of the packing kernel, 100 MHz): tests and tools */
int gnnmp_chain_jobs_pack(gnnmp_chain_jobs_t **out, const int64_t *seg_ptr, int64_t G, int64_t n_rows, int64_t max_graph, int has_empty,
                          gnnmp_stream_t stream);
int gnnmp_chain_jobs_release(gnnmp_chain_jobs_t *jobs, gnnmp_stream_t stream);
int gnnmp_chain_jobs_export(const gnnmp_chain_jobs_t *jobs, int32_t *tab_out, int64_t cap_rows, int32_t *hdr_out, gnnmp_stream_t stream);
/* jobs: NULL or the handle of THIS batch (gnnmp_chain_jobs_create on the same seg_ptr) */
int gnnmp_graphconv_chain_f32(gnnmp_graph_t *plan, const gnnmp_chain_jobs_t *jobs, const int64_t *seg_ptr, int64_t G, const float *x, int n_layers,
                              const int64_t *dims, const float *const *W_root, const float *const *W_agg,
                              const float *const *bias, const int *act, int w_layout, int aggr, int pool_aggr, const float *W_head,
                              const float *b_head, int64_t nout, float *scratch, float *out, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Aggregate-then-transform in one kernel — the layer bodies whose dense product FOLLOWS the aggregation:
 *   gcn_conv with Dout >= Din   σ.(W * (cin .* Σ_j w_j cout_j x_j) .+ b)      GNNlib/src/layers/conv.jl:59-71
 *   graph_conv                  σ.(W1 * x_i .+ W2 * aggr_j x_j .+ b)           conv.jl:102-108
 *   sage_conv                   σ.(W * vcat(x_i, aggr_j x_j) .+ b)             conv.jl:277-283
 *     out[i][:] = act( W_root * xi[i][:]  +  W_agg * A[i][:]  + bias ),   A = gnnmp_propagate(_slots)_f32's result
 * with A never written to HBM (it is BIT-IDENTICAL to what gnnmp_propagate_slots_f32 would return; pass agg_out to get it
 * as well, e.g. to test that).  The aggregation arguments mean what they mean in gnnmp_propagate_f32 (w: per-edge weights in
 * original order, scale_src / scale_dst per node) and gnnmp_propagate_slots_f32 (w_slot / ss_slot in plan slot order; they
 * take precedence); scalings need aggr = SUM | MEAN.  xi = NULL / D1 = 0: no root term (gcn_conv).  W_root [Dout][D1] and
 * W_agg [Dout][D] as in gnnmp_dense_f32 (w_layout, ldw).  Shapes taken: D, D1 multiples of 4 and <= 128, Dout a multiple of
 * 4 and <= 128, 16-byte aligned arrays, >= 16 destinations; anything else returns GNNMP_EUNSUPPORTED and the caller runs
 * gnnmp_propagate_f32 + gnnmp_dense_f32.  One stream at a time per plan (plan-owned workspace and tile ticket).
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_fused_conv_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *w, const float *scale_src,
                         const float *w_slot, const float *ss_slot, const float *scale_dst, int64_t D, const float *xi,
                         int64_t D1, const float *W_root, int64_t ldw_root, const float *W_agg, int64_t ldw_agg,
                         int w_layout, const float *bias, int act, float *out, int64_t Dout, float *agg_out,
                         gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Adjoints.  The reference trains through NNlib's rrules: ∇gather = scatter(+),
 * ∇scatter(+) = gather, ∇scatter(mean) = gather ./ count, ∇scatter(max|min) = (src .== gather(dst)) .* gather(Δ);
 * the CPU fast path through the adjacency_matrix rrule (GNNGraphs/src/query.jl:244-278).
 *
 * The adjoint of the fused propagate w.r.t. xj needs no entry point of its own: build a plan of the TRANSPOSED edge
 * index (gnnmp_plan_create(..., src = t, dst = s, ...)) and call
 *   gnnmp_propagate_f32(plan_T, msg, GNNMP_SUM, Δ, w, scale_src = scale_dst_fwd (./ count for MEAN), scale_dst = scale_src_fwd, Δx, D)
 * (same per-source edge order as NNlib's scatter(+, gather(Δ, t) .* w, s)).  The two that do:
 * ---------------------------------------------------------------------------------------------- */
/* out[k] = Σ_d a_dst[dst_k][d] * b_src[src_k][d] for every edge k (original order) — the adjoint w.r.t. the edge
 * weights of w_mul_xj / e_mul_xj:  Δw = sum(Δm .* xj_gathered, dims = 1)  with Δm = gather(Δ, t). */
int gnnmp_edge_dot_f32(const float *a_dst, const float *b_src, const void *src, const void *dst,
                       int idx_bytes, int index_base, int64_t n_edges, int64_t D, float *out,
                       gnnmp_stream_t stream);
/* The same per-edge dot products walked in the plan's destination-sorted order (the destination row stays in registers
 * for all of its edges: half the traffic); out[n_edges] is written in ORIGINAL edge order, plan-added self loops
 * produce no output.  D * 4 bytes must fit one wave of 16-byte lanes (D <= 256), else GNNMP_EUNSUPPORTED. */
int gnnmp_edge_dot_plan_f32(gnnmp_graph_t *plan, const float *a_dst, const float *b_src, float *out,
                            int64_t D, gnnmp_stream_t stream);
/* Adjoint of propagate(copy_xj, g, max|min) w.r.t. xj on the transposed plan:
 *   Δx_j[d] = Σ_{k: s_k = j, edge order} (x_j[d] == y_{t_k}[d]) ? Δ_{t_k}[d] : 0      (ties all receive Δ, like NNlib)
 * x: forward input [n][D], y: forward output, dy: incoming gradient, dx: output. */
int gnnmp_propagate_maxmin_grad_f32(gnnmp_graph_t *plan_transposed, const float *x, const float *y,
                                    const float *dy, float *dx, int64_t D, gnnmp_stream_t stream);

/* Adjoints of the dense part  y = σ.(W * x .+ b):
 *   gnnmp_act_grad_f32      Δz = Δy .* σ'(z)  (relu: Δy where y > 0, else 0; identity: copy).  dz may alias dy.
 *   gnnmp_dense_grad_w_f32  ΔW[o][k] = Σ_n Δz[n][o] x[n][k]  ([Dout][K] row-major) and/or Δb[o] = Σ_n Δz[n][o]
 *                           (either output may be NULL); fp32 MFMA, operands read straight from HBM, per-slab partials in
 *                           `workspace` (gnnmp_dense_grad_workspace(N, Dout, K) floats) folded in slab order: no atomics.
 *   Δx = W' * Δz is the forward kernel: gnnmp_dense_f32(Δz, W, D1 = Dout, ldw1 = K, ..., w_layout = 1, ..., Dout = K). */
int gnnmp_act_grad_f32(const float *dy, const float *y, int act, float *dz, int64_t n, gnnmp_stream_t stream);
/* The training step of the graph-classification chain (examples/graph_classification_tudataset.jl:79-82,97-104) with fewer passes over
 * the (N, D) arrays — the same arithmetic as the entry points above, fused where two of them touched one array (round 5):
 *   gnnmp_dense_grad_w2_f32   ΔW1 = Δz' x1, ΔW2 = Δz' x2 and Δb = colsum(Δz) from ONE read of Δz — graph_conv's / sage_conv's two weight
 *                             gradients (conv.jl:102-108: x1 = x_i, x2 = the aggregate).  out = [ΔW1 (Dout x K1) | ΔW2 (Dout x K2) | Δb
 *                             (Dout)], contiguous; K1 a multiple of 16 (else GNNMP_EUNSUPPORTED: call gnnmp_dense_grad_w_f32 twice);
 *                             workspace: gnnmp_dense_grad_w2_workspace floats.  Same slabs, same fold order: the three results are
 *                             bit-identical to the separate calls.
 *   gnnmp_pool_grad_act_f32   dz[i][:] = act'(y[i][:]) .* dpool[g(i)][:] (.* inv_count[g(i)] if given): the pullback of
 *                             reduce_nodes(+ | mean) (utils.jl:12-16, NNlib's ∇scatter = gather) and of the activation of the layer
 *                             that fed the pool, one pass instead of mul_rows + gather + act_grad.
 *   gnnmp_propagate_add_mask_f32   out[i][:] = act'(mask_y[i][:]) .* (addend[i][:] + Σ_{j -> i} xj[j][:]) — the x-pullback of graph_conv,
 *                             Δx = Δz W_root + Aᵀ(Δz W_agg), with the sum and the relu' of the layer below in the row kernel's
 *                             epilogue (plan = the plan of the REVERSED edges; aggr + or mean with scale_dst = 1 / count as in
 *                             gnnmp_propagate_f32; addend / mask_y optional). */
int64_t gnnmp_dense_grad_w2_workspace(int64_t N, int64_t Dout, int64_t K1, int64_t K2);
int gnnmp_dense_grad_w2_f32(const float *dz, const float *x1, int64_t K1, const float *x2, int64_t K2, int64_t N, int64_t Dout,
                            float *out, float *workspace, int64_t workspace_floats, gnnmp_stream_t stream);
int gnnmp_pool_grad_act_f32(const float *dpool, const void *graph_indicator, int idx_bytes, int index_base, const float *inv_count,
                            const float *y, int act, float *dz, int64_t N, int64_t G, int64_t D, gnnmp_stream_t stream);
int gnnmp_propagate_add_mask_f32(gnnmp_graph_t *plan, int aggr, const float *xj, const float *scale_dst, const float *addend,
                                 const float *mask_y, float *out, int64_t D, gnnmp_stream_t stream);
int64_t gnnmp_dense_grad_workspace(int64_t N, int64_t Dout, int64_t K);
int gnnmp_dense_grad_w_f32(const float *dz, const float *x, int64_t N, int64_t Dout, int64_t K, float *dW,
                           float *db, float *workspace, int64_t workspace_floats, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Float64 features (round 6).  The reference's message passing is eltype-generic, and its own micro-benchmark runs in Float64
 * (GraphNeuralNetworks/perf/bench_gnn.jl:9-10: `B = rand(100, n)`; it asserts isequal(propagate(e_mul_xj, g, +; xj = B, e), B * A)).  The
 * seam's methods and their leaves exist for `double` too — the same plan, the same walk, the same order of operations as the `_f32` entry
 * points (adds in ORIGINAL edge order, products rounded separately: bit-identical to NNlib's CPU loop on every row the plan does not split;
 * split rows: chunk partials folded in chunk order, deterministic):
 *   gnnmp_propagate_f64   propagate(copy_xj | w_mul_xj | e_mul_xj with a vector e, g, + | mean | max | min)   msgpass.jl:71-79, 215-238;
 *                         arguments as gnnmp_propagate_f32 (w: n_edges doubles in original edge order; scale_src / scale_dst nullable)
 *   gnnmp_gather_f64      _gather   (gatherscatter.jl:4)
 *   gnnmp_scatter_f64     _scatter  (gatherscatter.jl:12-18) with idx = the plan's targets, m in original edge order
 * The layer bodies' dense products stay with the caller in Float64 (`l.weight * x` is rocBLAS in Julia): the fused layer kernels, the
 * attention kernels and the adjoints are Float32 only, and so is everything `bench.py` measures.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_propagate_f64(gnnmp_graph_t *plan, int msg, int aggr, const double *xj, const double *w, const double *scale_src,
                        const double *scale_dst, double *out, int64_t D, gnnmp_stream_t stream);
int gnnmp_gather_f64(const double *x, const void *idx, int idx_bytes, int index_base, int64_t K, double *out, int64_t D,
                     gnnmp_stream_t stream);
int gnnmp_scatter_f64(gnnmp_graph_t *plan, int aggr, const double *m, double *out, int64_t D, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Repair hook.  The split rows of a plan are folded inside the row kernels through per-plan arrival counters that every launch leaves at
 * zero (csrc/common.h: ensure_arrive states the invariant and when it holds).  A caller that broke the one-stream-per-plan rule above
 * — two compute calls on one plan in flight at once — may leave them dirty, and every later result on that plan's split rows would be
 * wrong.  gnnmp_plan_reset_counters enqueues a memset of the counters (and of the fused layer kernel's tile ticket) on `stream`; a few
 * KB, no synchronisation, a no-op for a plan without split rows.  Never needed by a caller that keeps the rule.
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_plan_reset_counters(gnnmp_graph_t *plan, gnnmp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GNNMP_INTERNAL — exported, NOT part of the drop-in surface: experiment and test hooks.  Declared here so that C callers (tests/c_harness)
 * do not declare them by hand; a Julia / C host has no reason to call them, and their meaning may change between builds.
 *   gnnmp_tune(knob, value)        process-global tuning knobs of the perf experiments (csrc/common.h: enum Knob; every value selects
 *                                  correct code).  Not thread-safe against concurrent compute calls.
 *   gnnmp_debug_mock_device(d)     d >= 0: the calling thread's "current device" for the library's per-device tables; d < 0: hipGetDevice
 *   gnnmp_debug_device_once(...)   runs the once-per-device machinery with a counting stand-in (tests/test_multi_device_cpu.py)
 *   gnnmp_debug_plan_block(plan)   the pooled block of a gnnmp_plan_concat / gnnmp_plan_select plan (NULL otherwise)
 *   gnnmp_debug_pool_pick(...)     the block pool's slot choice on host arrays
 * ---------------------------------------------------------------------------------------------- */
int gnnmp_tune(int knob, int value);
int gnnmp_debug_mock_device(int dev);
int gnnmp_debug_device_once(const int *devs, int n, int fail_on, int *n_failed);
void *gnnmp_debug_plan_block(const gnnmp_graph_t *plan);
int gnnmp_debug_pool_pick(const uint64_t *caps, int n, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GNNMP_H */
