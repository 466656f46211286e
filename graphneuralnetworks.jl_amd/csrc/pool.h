// pool.h — stream-ordered pool of device blocks for the PER-BATCH objects of a training loop (the plan of a selected / concatenated batch,
// its chain jobs).  examples/graph_classification_tudataset.jl:70-71,97-104 makes a new batch every step; hipMalloc + hipFree per batch
// would cost two device-wide synchronisations per object.  A released block is parked together with an event recorded on the stream its
// last work was enqueued on; the next taker waits for that event ON ITS STREAM (hipStreamWaitEvent: no host synchronisation), or not at
// all when it is the same stream.  Blocks parked without a stream (plain destroy) are handed out again only after a device-wide
// synchronisation — what hipFree would have cost.
// The pool is one table PER DEVICE (common.h: current_device): a taker sees only blocks allocated on its current device, events are
// created and recycled on their own device, and stream handles are compared inside one device only — a process that drives eight
// devices gets eight independent pools.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gnnmp {
// out/cap: the block and its real capacity (>= bytes).  false: allocation failed.
bool pool_take(void **out, size_t *cap, size_t bytes, hipStream_t stream);
// stream_known: work that touches the block was last enqueued on `stream` (an event is recorded there); otherwise unknown streams.
void pool_park(void *p, size_t cap, hipStream_t stream, bool stream_known);
// the slot pool_take hands out for a request of `bytes` among blocks of capacities caps[0..n) (0 = empty slot): the smallest block that
// fits, unless it is more than twice the request (+ 1 MiB); -1 = none.  Pure host logic (tests/test_multi_device_cpu.py).
int pool_pick(const size_t *caps, int n, size_t bytes);
// free every parked block of every device (tests / process teardown)
void pool_trim();
}  // namespace gnnmp
