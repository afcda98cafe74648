// probe_vmm.hip -- diagnostic only (tools/probe_levels_vmm.py): thin C wrappers of HIP's virtual-memory management calls,
// so that ONE physical allocation can be mapped at several virtual addresses and several physical allocations at one
// virtual address -- which of the two carries the gather's speed level?  Not part of libvoxproj.so.
//
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libprobe_vmm.so tools/probe_vmm.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

extern "C" {

long long vmm_granularity(int device)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return -1;
    return (long long)g;
}

// physical allocation of `bytes` (a multiple of the granularity) on `device`; returns 0 and the handle
int vmm_create(int device, long long bytes, unsigned long long *handle)
{
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemGenericAllocationHandle_t h;
    const hipError_t e = hipMemCreate(&h, (size_t)bytes, &prop, 0);
    if (e != hipSuccess) return (int)e;
    *handle = (unsigned long long)(uintptr_t)h;
    return 0;
}

int vmm_release(unsigned long long handle)
{
    return (int)hipMemRelease((hipMemGenericAllocationHandle_t)(uintptr_t)handle);
}

// reserve `bytes` of virtual address space aligned to `align` (0 = default); returns 0 and the address
int vmm_reserve(long long bytes, long long align, unsigned long long *ptr)
{
    void *p = nullptr;
    const hipError_t e = hipMemAddressReserve(&p, (size_t)bytes, (size_t)align, nullptr, 0);
    if (e != hipSuccess) return (int)e;
    *ptr = (unsigned long long)(uintptr_t)p;
    return 0;
}

int vmm_free(unsigned long long ptr, long long bytes)
{
    return (int)hipMemAddressFree((void *)(uintptr_t)ptr, (size_t)bytes);
}

int vmm_map(int device, unsigned long long ptr, long long bytes, unsigned long long handle)
{
    hipError_t e = hipMemMap((void *)(uintptr_t)ptr, (size_t)bytes, 0, (hipMemGenericAllocationHandle_t)(uintptr_t)handle, 0);
    if (e != hipSuccess) return (int)e;
    hipMemAccessDesc d;
    memset(&d, 0, sizeof(d));
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = device;
    d.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess((void *)(uintptr_t)ptr, (size_t)bytes, &d, 1);
    return (int)e;
}

int vmm_unmap(unsigned long long ptr, long long bytes)
{
    return (int)hipMemUnmap((void *)(uintptr_t)ptr, (size_t)bytes);
}

}  // extern "C"
