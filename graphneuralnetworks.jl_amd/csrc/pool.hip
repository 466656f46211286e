// pool.hip — see pool.h
#include <mutex>

#include "pool.h"

namespace gnnmp {
namespace {
struct Parked {
    void *p = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;      // recorded on `stream` after the block's last use (nullptr: unknown streams)
};
constexpr int POOL_SLOTS = 8;
Parked g_slots[POOL_SLOTS];
hipEvent_t g_free_events[POOL_SLOTS] = {};
int g_n_free_events = 0;
std::mutex g_lock;

hipEvent_t event_get() {
    if (g_n_free_events > 0) return g_free_events[--g_n_free_events];
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
void event_put(hipEvent_t e) {
    if (!e) return;
    if (g_n_free_events < POOL_SLOTS) g_free_events[g_n_free_events++] = e;
    else (void)hipEventDestroy(e);
}
}  // namespace

bool pool_take(void **out, size_t *cap, size_t bytes, hipStream_t stream) {
    Parked got;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        int best = -1;
        for (int i = 0; i < POOL_SLOTS; ++i)
            if (g_slots[i].p && g_slots[i].cap >= bytes && (best < 0 || g_slots[i].cap < g_slots[best].cap)) best = i;
        if (best >= 0 && g_slots[best].cap <= 2 * bytes + (1 << 20)) {
            got = g_slots[best];
            g_slots[best] = Parked();
        }
    }
    if (got.p) {
        bool ok = true;
        if (!got.ev) {
            ok = hipDeviceSynchronize() == hipSuccess;                     // parked by a plain destroy: streams unknown
        } else {
            if (got.stream != stream) ok = hipStreamWaitEvent(stream, got.ev, 0) == hipSuccess;   // same stream: already ordered
            std::lock_guard<std::mutex> lk(g_lock);
            event_put(got.ev);
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
    Parked in;
    in.p = p;
    in.cap = cap;
    Parked evict;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        if (stream_known) {
            in.stream = stream;
            in.ev = event_get();
            if (in.ev && hipEventRecord(in.ev, stream) != hipSuccess) {
                event_put(in.ev);
                in.ev = nullptr;
            }
        }
        int slot = -1;
        for (int i = 0; i < POOL_SLOTS; ++i)
            if (!g_slots[i].p) { slot = i; break; }
        if (slot < 0) {                      // full: the smallest block goes
            slot = 0;
            for (int i = 1; i < POOL_SLOTS; ++i)
                if (g_slots[i].cap < g_slots[slot].cap) slot = i;
            if (g_slots[slot].cap >= cap) {
                evict = in;
                in = Parked();
            } else {
                evict = g_slots[slot];
            }
        }
        if (in.p) g_slots[slot] = in;
        if (evict.ev) { event_put(evict.ev); evict.ev = nullptr; }
    }
    if (evict.p) (void)hipFree(evict.p);      // (hipFree waits for the device: safe whatever still uses the block)
}

void pool_trim() {
    Parked all[POOL_SLOTS];
    {
        std::lock_guard<std::mutex> lk(g_lock);
        for (int i = 0; i < POOL_SLOTS; ++i) {
            all[i] = g_slots[i];
            g_slots[i] = Parked();
            if (all[i].ev) event_put(all[i].ev);
        }
    }
    for (int i = 0; i < POOL_SLOTS; ++i)
        if (all[i].p) (void)hipFree(all[i].p);
}
}  // namespace gnnmp
