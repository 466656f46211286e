// pool.hip — see pool.h
#include <mutex>

#include "pool.h"
#include "common.h"

namespace gnnmp {
namespace {
struct Parked {
    void *p = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;      // recorded on `stream` after the block's last use (nullptr: unknown streams)
};
constexpr int POOL_SLOTS = 8;
// One table PER DEVICE: a block lives in the memory of the device it was allocated on, an event belongs to the device it was created on,
// and stream handles are only comparable inside one device (the null stream is nullptr on every device).  A taker only ever sees the
// blocks and events of ITS current device (common.h: current_device).
struct DevicePool {
    Parked slots[POOL_SLOTS];
    hipEvent_t free_events[POOL_SLOTS] = {};
    int n_free_events = 0;
};
DevicePool g_pools[GNNMP_MAX_DEVICES];
std::mutex g_lock;

hipEvent_t event_get(DevicePool &dp) {
    if (dp.n_free_events > 0) return dp.free_events[--dp.n_free_events];
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
void event_put(DevicePool &dp, hipEvent_t e) {
    if (!e) return;
    if (dp.n_free_events < POOL_SLOTS) dp.free_events[dp.n_free_events++] = e;
    else (void)hipEventDestroy(e);
}
// "work enqueued on `a` is ordered before work enqueued later on `b`" without an event: the same explicit stream (or the null stream)
// of one device.  hipStreamPerThread is one handle for a different stream in every host thread: never "the same".
bool same_stream(hipStream_t a, hipStream_t b) { return a == b && a != hipStreamPerThread; }
}  // namespace

int pool_pick(const size_t *caps, int n, size_t bytes) {
    int best = -1;
    for (int i = 0; i < n; ++i)
        if (caps[i] && caps[i] >= bytes && (best < 0 || caps[i] < caps[best])) best = i;
    if (best >= 0 && caps[best] <= 2 * bytes + (1 << 20)) return best;
    return -1;
}

bool pool_take(void **out, size_t *cap, size_t bytes, hipStream_t stream) {
    DevicePool &dp = g_pools[current_device()];
    Parked got;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        size_t caps[POOL_SLOTS];
        for (int i = 0; i < POOL_SLOTS; ++i) caps[i] = dp.slots[i].p ? dp.slots[i].cap : 0;
        const int best = pool_pick(caps, POOL_SLOTS, bytes);
        if (best >= 0) {
            got = dp.slots[best];
            dp.slots[best] = Parked();
        }
    }
    if (got.p) {
        bool ok = true;
        if (!got.ev) {
            ok = hipDeviceSynchronize() == hipSuccess;                     // parked by a plain destroy: streams unknown (same device)
        } else {
            if (!same_stream(got.stream, stream)) ok = hipStreamWaitEvent(stream, got.ev, 0) == hipSuccess;
            std::lock_guard<std::mutex> lk(g_lock);
            event_put(dp, got.ev);
        }
        if (ok) {
            *out = got.p;
            *cap = got.cap;
            return true;
        }
        (void)hipFree(got.p);
    }
    *cap = bytes;
    return hipMalloc(out, bytes) == hipSuccess;
}

void pool_park(void *p, size_t cap, hipStream_t stream, bool stream_known) {
    if (!p) return;
    // the block's own device, not the caller's current one: a plan may be released after the host switched devices
    int dev = current_device();
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device >= 0 && at.device < GNNMP_MAX_DEVICES) {
            if (at.device != dev) stream_known = false;    // `stream` cannot be a stream of the block's device: unknown
            dev = at.device;
        } else {
            (void)hipGetLastError();
        }
    }
    DevicePool &dp = g_pools[dev];
    Parked in;
    in.p = p;
    in.cap = cap;
    Parked evict;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        if (stream_known) {
            in.stream = stream;
            in.ev = event_get(dp);
            if (in.ev && hipEventRecord(in.ev, stream) != hipSuccess) {
                (void)hipGetLastError();
                event_put(dp, in.ev);
                in.ev = nullptr;
            }
        }
        int slot = -1;
        for (int i = 0; i < POOL_SLOTS; ++i)
            if (!dp.slots[i].p) { slot = i; break; }
        if (slot < 0) {                      // full: the smallest block goes
            slot = 0;
            for (int i = 1; i < POOL_SLOTS; ++i)
                if (dp.slots[i].cap < dp.slots[slot].cap) slot = i;
            if (dp.slots[slot].cap >= cap) {
                evict = in;
                in = Parked();
            } else {
                evict = dp.slots[slot];
            }
        }
        if (in.p) dp.slots[slot] = in;
        if (evict.ev) { event_put(dp, evict.ev); evict.ev = nullptr; }
    }
    if (evict.p) (void)hipFree(evict.p);      // (hipFree waits for the device: safe whatever still uses the block)
}

void pool_trim() {
    for (int d = 0; d < GNNMP_MAX_DEVICES; ++d) {
        Parked all[POOL_SLOTS];
        bool any = false;
        {
            std::lock_guard<std::mutex> lk(g_lock);
            DevicePool &dp = g_pools[d];
            for (int i = 0; i < POOL_SLOTS; ++i) {
                all[i] = dp.slots[i];
                dp.slots[i] = Parked();
                if (all[i].ev) event_put(dp, all[i].ev);
                any = any || all[i].p;
            }
        }
        if (!any) continue;
        for (int i = 0; i < POOL_SLOTS; ++i)
            if (all[i].p) (void)hipFree(all[i].p);
    }
}
}  // namespace gnnmp
