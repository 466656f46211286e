// vmm.hip — HIP virtual-memory-management helper for the placement experiments (tools/vmm_probe.py): physical chunks created one by one
// (hipMemCreate), mapped wherever the caller wants inside a reserved address range.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

static hipMemAllocationProp make_prop(int dev) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = dev;
    return p;
}
#define CK(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { fprintf(stderr, "vmm: %s -> %s\n", #e, hipGetErrorString(e__)); return 0; } } while (0)

extern "C" {
uint64_t vmm_granularity(int dev) {
    size_t g = 0;
    hipMemAllocationProp p = make_prop(dev);
    CK(hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended));
    return g;
}
uint64_t vmm_chunk_create(uint64_t size, int dev) {
    hipMemGenericAllocationHandle_t h = nullptr;
    hipMemAllocationProp p = make_prop(dev);
    CK(hipMemCreate(&h, size, &p, 0));
    return (uint64_t)(uintptr_t)h;
}
int vmm_chunk_release(uint64_t h) { return (int)hipMemRelease((hipMemGenericAllocationHandle_t)(uintptr_t)h); }
uint64_t vmm_reserve(uint64_t size, uint64_t align) {
    void *p = nullptr;
    CK(hipMemAddressReserve(&p, size, align, nullptr, 0));
    return (uint64_t)(uintptr_t)p;
}
int vmm_address_free(uint64_t p, uint64_t size) { return (int)hipMemAddressFree((void *)(uintptr_t)p, size); }
int vmm_map(uint64_t va, uint64_t size, uint64_t h, int dev) {
    hipError_t e = hipMemMap((void *)(uintptr_t)va, size, 0, (hipMemGenericAllocationHandle_t)(uintptr_t)h, 0);
    if (e != hipSuccess) { fprintf(stderr, "vmm: hipMemMap -> %s\n", hipGetErrorString(e)); return (int)e; }
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = dev;
    d.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess((void *)(uintptr_t)va, size, &d, 1);
    if (e != hipSuccess) fprintf(stderr, "vmm: hipMemSetAccess -> %s\n", hipGetErrorString(e));
    return (int)e;
}
int vmm_unmap(uint64_t va, uint64_t size) { return (int)hipMemUnmap((void *)(uintptr_t)va, size); }
}
