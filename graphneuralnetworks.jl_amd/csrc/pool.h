// pool.h — stream-ordered pool of device blocks for the PER-BATCH objects of a training loop (the plan of a selected / concatenated batch,
// its chain jobs).  examples/graph_classification_tudataset.jl:70-71,97-104 makes a new batch every step; hipMalloc + hipFree per batch
// would cost two device-wide synchronisations per object.  A released block is parked together with an event recorded on the stream its
// last work was enqueued on; the next taker waits for that event ON ITS STREAM (hipStreamWaitEvent: no host synchronisation), or not at
// all when it is the same stream.  Blocks parked without a stream (plain destroy) are handed out again only after a device-wide
// synchronisation — what hipFree would have cost.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace gnnmp {
// out/cap: the block and its real capacity (>= bytes).  false: allocation failed.
bool pool_take(void **out, size_t *cap, size_t bytes, hipStream_t stream);
// stream_known: work that touches the block was last enqueued on `stream` (an event is recorded there); otherwise unknown streams.
void pool_park(void *p, size_t cap, hipStream_t stream, bool stream_known);
// free every parked block (tests / process teardown)
void pool_trim();
}  // namespace gnnmp
