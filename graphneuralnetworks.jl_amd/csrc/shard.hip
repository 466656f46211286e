// shard.hip — the host side of the graph-parallel step behind the C ABI (north_star: "batched graphs shard by graph
// across up to 8 GPUs of one node with an RCCL-over-xGMI all-gather of per-shard logits"):
//   gnnmp_shard_by_size     who owns which member graph of a batch (MLUtils.batch, GNNGraphs/src/transform.jl:682-709: member graphs
//                           are independent units), and where each graph's logits land in the all-gathered block
//   gnnmp_allgather_f32     the ONE collective of the step, on the caller's RCCL communicator and stream
// The reference has no distributed code; the host mirror of the same design is gnnmp/parallel.py (torch.distributed).
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <mutex>
#include <numeric>
#include <vector>

#include "common.h"

using namespace gnnmp;

// Deterministic balanced deal: member graphs by decreasing size (ties by index) in a boustrophedon over the ranks — every rank gets
// the same count +-1 and nearly the same number of nodes.  All arrays are HOST arrays.
//   rank_of[g]      owner of member graph g
//   gather_index[g] row of graph g's logits in the all-gathered block of world * gmax rows (rank * gmax + position of g among its
//                   rank's graphs in ascending graph order: a rank batches its members in that order), nullable
//   *gmax_out       rows every rank contributes to the collective (the largest shard; shorter ones are padded), nullable
extern "C" int gnnmp_shard_by_size(const int64_t *sizes, int64_t G, int world, int32_t *rank_of, int64_t *gather_index,
                                   int64_t *gmax_out) {
    if (G < 0 || world < 1 || (G > 0 && (!sizes || !rank_of))) return fail(GNNMP_EINVAL, "shard_by_size: bad argument");
    std::vector<int64_t> order((size_t)G);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t p, int64_t q) { return sizes[p] > sizes[q]; });
    std::vector<int64_t> count((size_t)world, 0);
    for (int64_t pos = 0; pos < G; ++pos) {
        const int64_t rnd = pos / world, k = pos % world;
        const int r = (int)((rnd & 1) ? world - 1 - k : k);
        rank_of[order[(size_t)pos]] = r;
        ++count[(size_t)r];
    }
    const int64_t gmax = G > 0 ? *std::max_element(count.begin(), count.end()) : 0;
    if (gmax_out) *gmax_out = gmax;
    if (gather_index) {
        std::vector<int64_t> seen((size_t)world, 0);
        for (int64_t g = 0; g < G; ++g) {          // ascending graph order inside every rank
            const int r = rank_of[g];
            gather_index[g] = (int64_t)r * gmax + seen[(size_t)r]++;
        }
    }
    return GNNMP_OK;
}

// ncclAllGather(send, recv, count, ncclFloat32, comm, stream) on the caller's communicator: every rank contributes `count` floats
// (gmax * nout, its logits padded to the largest shard), recv holds world * count.  librccl.so is resolved at the first call
// (dlopen): the library itself does not link RCCL, a single-GPU user never loads it.
extern "C" int gnnmp_allgather_f32(void *nccl_comm, const float *send, float *recv, int64_t count, gnnmp_stream_t stream) {
    if (!nccl_comm || !send || !recv || count < 0) return fail(GNNMP_EINVAL, "allgather: bad argument");
    typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
    // The communicator was made by an RCCL that is ALREADY in the process (e.g. torch's bundled librccl.so, SONAME librccl.so.1): its
    // ncclAllGather is the only one the handle may be given to.  So: the process's global symbols first, then the loaded copies by SONAME
    // (RTLD_NOLOAD: never map a second build next to the caller's), and only when no RCCL is loaded at all a fresh dlopen.
    static allgather_fn fn = nullptr;
    static std::once_flag once;
    static char why[256] = "";
    std::call_once(once, [] {
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !sym; ++pass)
            for (const char *nm : names) {
                void *h = dlopen(nm, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_GLOBAL));
                if (h && (sym = dlsym(h, "ncclAllGather"))) break;
            }
        if (!sym) {
            const char *e = dlerror();
            snprintf(why, sizeof(why), "%s", e ? e : "ncclAllGather not found");
        }
        fn = reinterpret_cast<allgather_fn>(sym);
    });
    if (!fn) return fail(GNNMP_EUNSUPPORTED, "allgather: no RCCL with ncclAllGather in this process or on the library path (%s)", why);
    const int rc = fn(send, recv, (size_t)count, 7 /* ncclFloat32 */, nccl_comm, (hipStream_t)stream);
    if (rc != 0) return fail(GNNMP_ELAUNCH, "allgather: ncclAllGather returned %d", rc);
    return GNNMP_OK;
}
