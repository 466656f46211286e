// cumask.hip — helper for the CU-mask experiments (tools/sage_cumask.py): streams restricted to a subset of the compute units
// (hipExtStreamCreateWithCUMask), events and waits on raw stream handles.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

extern "C" {
// bits [lo, hi) of the CU mask set.  On a multi-XCD part the driver deals mask bit i to XCD i % n_xcd, so a contiguous bit range is
// spread evenly over the XCDs.
void *cm_stream_create(int lo, int hi, int total) {
    uint32_t mask[16] = {0};
    const int words = (total + 31) / 32;
    for (int i = lo; i < hi && i < total; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { fprintf(stderr, "hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return nullptr; }
    return s;
}
void cm_stream_destroy(void *s) { (void)hipStreamDestroy((hipStream_t)s); }
void *cm_event_create() { hipEvent_t e = nullptr; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); return e; }
int cm_event_record(void *e, void *s) { return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)s); }
int cm_stream_wait(void *s, void *e) { return (int)hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0); }
int cm_stream_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
int cm_get_mask(void *s, uint32_t *out, int words) { return (int)hipExtStreamGetCUMask((hipStream_t)s, (uint32_t)words, out); }
}
