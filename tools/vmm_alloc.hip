// A virtually contiguous device buffer backed by physical chunks of a chosen size (HIP virtual memory management):
// experiment helper for tools/beyond_vmm.py - how does the physical chunking of the state matrix change what HBM delivers?
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/_variants/libvmm.so tools/vmm_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "vmm: %s: %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

struct Vmm { void* va; size_t total; std::vector<hipMemGenericAllocationHandle_t> handles; std::vector<size_t> sizes; };

extern "C" size_t vmm_granularity(int device)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
    return g;
}

// total bytes in chunks of `chunk` bytes (the last one shorter); order: 0 = create and map in address order, 1 = create all chunks
// in reverse order (the driver hands physical memory out in creation order), 2 = pseudo-random creation order (seed);
// spacer_mib: a throw-away allocation of that size made between chunk creations (moves the next chunk elsewhere)
extern "C" int vmm_alloc(int device, size_t total, size_t chunk, int order, unsigned seed, int spacer_mib, void** out_va, void** out_handle)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t g = 0;
    CK(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum));
    total = (total + g - 1) / g * g; chunk = (chunk + g - 1) / g * g;
    Vmm* v = new Vmm{nullptr, total, {}, {}};
    CK(hipMemAddressReserve(&v->va, total, 0, nullptr, 0));
    const size_t nchunks = (total + chunk - 1) / chunk;
    std::vector<size_t> idx(nchunks);
    for (size_t k = 0; k < nchunks; ++k) idx[k] = k;
    if (order == 1) std::reverse(idx.begin(), idx.end());
    if (order == 2) { uint64_t s = seed * 2654435761ull + 1; for (size_t k = nchunks; k > 1; --k) { s = s * 6364136223846793005ull + 1442695040888963407ull; std::swap(idx[k - 1], idx[(s >> 33) % k]); } }
    v->handles.resize(nchunks); v->sizes.resize(nchunks);
    std::vector<void*> spacers;
    for (size_t j = 0; j < nchunks; ++j) {
        const size_t k = idx[j];
        const size_t sz = std::min(chunk, total - k * chunk);
        if (spacer_mib > 0) { void* sp = nullptr; CK(hipMalloc(&sp, (size_t)spacer_mib << 20)); spacers.push_back(sp); }
        CK(hipMemCreate(&v->handles[k], sz, &prop, 0));
        v->sizes[k] = sz;
    }
    for (size_t k = 0; k < nchunks; ++k) CK(hipMemMap((char*)v->va + k * chunk, v->sizes[k], 0, v->handles[k], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = device; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v->va, total, &acc, 1));
    for (void* sp : spacers) CK(hipFree(sp));
    *out_va = v->va; *out_handle = v;
    return 0;
}

extern "C" int vmm_free(void* handle)
{
    Vmm* v = (Vmm*)handle;
    CK(hipMemUnmap(v->va, v->total));
    for (auto h : v->handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(v->va, v->total));
    delete v;
    return 0;
}
