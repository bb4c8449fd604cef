// libdfx, memory placement for MI355X: physical chunks of device memory mapped into contiguous virtual ranges in
// an order the CALLER chooses (HIP virtual-memory API).  C ABI in include/dfx.h (dfx_arena_*).
//
// Why it exists (DESIGN.md "Placement of large vectors", profiles/r03_placement_*): the 288 GiB of an MI355X fall
// into three classes of 96 GiB; two 5-D vectors swept in lock-step (x and p; x and exp_avg_sq) stream at 6.4-6.5 TB/s
// when, at equal offsets, they lie in DIFFERENT classes and at 5.4-5.5 TB/s when they share one.  Which physical
// range hipMalloc hands out is the driver's choice and depends on the allocation history; the class of a physical
// chunk can be measured (time a pass over two chunks); and with hipMemCreate / hipMemMap the chunks can be laid out
// behind contiguous virtual addresses in any order.  So: create chunks, classify them (datafusion_amd/placement.py,
// with the real sweep kernel), map vector k's i-th chunk so that the vectors' classes differ at every i.
//
// Nothing here is on the timed path; these entry points allocate and synchronise by nature (they are the only ones
// in the library that do) and are called at set-up time only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/dfx.h"

namespace {

thread_local char g_arena_err[256] = "";

int afail(int code, const char *msg) {
    snprintf(g_arena_err, sizeof(g_arena_err), "%s", msg);
    return code;
}
int ahip(hipError_t e, const char *where) {
    snprintf(g_arena_err, sizeof(g_arena_err), "%s: %s", where, hipGetErrorString(e));
    (void)hipGetLastError();
    return (int)e;
}

struct View {
    void *base;
    size_t bytes;
    std::vector<int> chunks;
};

}  // namespace

struct dfx_arena {
    int device;
    size_t chunk_bytes;
    hipMemAllocationProp prop;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<char> live;      // chunk still owned (not released)
    std::vector<int> mapped;     // how many views map the chunk (0 or 1: no aliasing)
    std::vector<View> views;
    std::mutex mu;
};

extern "C" {

const char *dfx_arena_last_error(void) { return g_arena_err; }

int dfx_arena_create(int64_t chunk_bytes, int64_t n_chunks, dfx_arena **out) {
    if (!out || chunk_bytes <= 0 || n_chunks <= 0 || n_chunks > (1 << 20))
        return afail(DFX_EINVAL, "dfx_arena_create: bad argument");
    *out = nullptr;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ahip(e, "dfx_arena_create: hipGetDevice");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleTypes = hipMemHandleTypeNone;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0) return ahip(e, "dfx_arena_create: hipMemGetAllocationGranularity");
    const size_t cb = ((size_t)chunk_bytes + gran - 1) / gran * gran;
    dfx_arena *a = new (std::nothrow) dfx_arena();
    if (!a) return afail(DFX_EINVAL, "dfx_arena_create: out of host memory");
    a->device = dev;
    a->chunk_bytes = cb;
    a->prop = prop;
    if (int rc = dfx_arena_grow(a, n_chunks)) {
        delete a;
        return rc;
    }
    *out = a;
    return 0;
}

int dfx_arena_grow(dfx_arena *a, int64_t n_more) {
    if (!a || n_more <= 0 || n_more > (1 << 20)) return afail(DFX_EINVAL, "dfx_arena_grow: bad argument");
    std::lock_guard<std::mutex> lk(a->mu);
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return ahip(e, "dfx_arena_grow: hipGetDevice");
    if (cur != a->device) return afail(DFX_EINVAL, "dfx_arena_grow: the arena's device is not the current device");
    const size_t before = a->handles.size();
    for (int64_t i = 0; i < n_more; ++i) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, a->chunk_bytes, &a->prop, 0);
        if (e != hipSuccess) {       // all or nothing: give back what this call created
            int rc = ahip(e, "dfx_arena_grow: hipMemCreate");
            while (a->handles.size() > before) {
                (void)hipMemRelease(a->handles.back());
                a->handles.pop_back();
            }
            return rc;
        }
        a->handles.push_back(h);
    }
    a->live.resize(a->handles.size(), 1);
    a->mapped.resize(a->handles.size(), 0);
    return 0;
}

int dfx_arena_info(const dfx_arena *a, int64_t *chunk_bytes, int64_t *n_chunks, int32_t *device) {
    if (!a) return afail(DFX_EINVAL, "dfx_arena_info: arena is NULL");
    if (chunk_bytes) *chunk_bytes = (int64_t)a->chunk_bytes;
    if (n_chunks) *n_chunks = (int64_t)a->handles.size();
    if (device) *device = a->device;
    return 0;
}

int dfx_arena_map(dfx_arena *a, const int32_t *chunks, int64_t n, void **out_ptr) {
    if (!a || !chunks || n <= 0 || !out_ptr) return afail(DFX_EINVAL, "dfx_arena_map: bad argument");
    std::lock_guard<std::mutex> lk(a->mu);
    for (int64_t i = 0; i < n; ++i) {
        const int c = chunks[i];
        if (c < 0 || c >= (int)a->handles.size() || !a->live[c])
            return afail(DFX_EINVAL, "dfx_arena_map: chunk index out of range or released");
        if (a->mapped[c]) return afail(DFX_EINVAL, "dfx_arena_map: chunk is already mapped in another view");
        for (int64_t j = 0; j < i; ++j)
            if (chunks[j] == c) return afail(DFX_EINVAL, "dfx_arena_map: chunk listed twice");
    }
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return ahip(e, "dfx_arena_map: hipGetDevice");
    if (cur != a->device) return afail(DFX_EINVAL, "dfx_arena_map: the arena's device is not the current device");
    const size_t bytes = (size_t)n * a->chunk_bytes;
    void *base = nullptr;
    e = hipMemAddressReserve(&base, bytes, 0, nullptr, 0);
    if (e != hipSuccess || !base) return ahip(e, "dfx_arena_map: hipMemAddressReserve");
    int64_t done = 0;
    for (; done < n; ++done) {
        e = hipMemMap((char *)base + (size_t)done * a->chunk_bytes, a->chunk_bytes, 0, a->handles[chunks[done]], 0);
        if (e != hipSuccess) break;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc desc = {};
        desc.location.type = hipMemLocationTypeDevice;
        desc.location.id = a->device;
        desc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(base, bytes, &desc, 1);
    }
    if (e != hipSuccess) {
        int rc = ahip(e, "dfx_arena_map: hipMemMap / hipMemSetAccess");
        for (int64_t i = 0; i < done; ++i) (void)hipMemUnmap((char *)base + (size_t)i * a->chunk_bytes, a->chunk_bytes);
        (void)hipMemAddressFree(base, bytes);
        return rc;
    }
    View v;
    v.base = base;
    v.bytes = bytes;
    v.chunks.assign(chunks, chunks + n);
    for (int c : v.chunks) a->mapped[c] = 1;
    a->views.push_back(v);
    *out_ptr = base;
    return 0;
}

int dfx_arena_unmap(dfx_arena *a, void *ptr) {
    if (!a || !ptr) return afail(DFX_EINVAL, "dfx_arena_unmap: bad argument");
    std::lock_guard<std::mutex> lk(a->mu);
    for (size_t k = 0; k < a->views.size(); ++k) {
        if (a->views[k].base != ptr) continue;
        View v = a->views[k];
        hipError_t e = hipDeviceSynchronize();   // nothing may still be running on the range
        if (e != hipSuccess) return ahip(e, "dfx_arena_unmap: hipDeviceSynchronize");
        e = hipMemUnmap(v.base, v.bytes);
        if (e != hipSuccess) return ahip(e, "dfx_arena_unmap: hipMemUnmap");
        e = hipMemAddressFree(v.base, v.bytes);
        if (e != hipSuccess) return ahip(e, "dfx_arena_unmap: hipMemAddressFree");
        for (int c : v.chunks) a->mapped[c] = 0;
        a->views.erase(a->views.begin() + (long)k);
        return 0;
    }
    return afail(DFX_EINVAL, "dfx_arena_unmap: not the base address of a view of this arena");
}

int dfx_arena_release_chunks(dfx_arena *a, const int32_t *chunks, int64_t n) {
    if (!a || (!chunks && n > 0) || n < 0) return afail(DFX_EINVAL, "dfx_arena_release_chunks: bad argument");
    std::lock_guard<std::mutex> lk(a->mu);
    for (int64_t i = 0; i < n; ++i) {
        const int c = chunks[i];
        if (c < 0 || c >= (int)a->handles.size()) return afail(DFX_EINVAL, "dfx_arena_release_chunks: index out of range");
        if (!a->live[c]) continue;
        if (a->mapped[c]) return afail(DFX_EINVAL, "dfx_arena_release_chunks: chunk is mapped; unmap its view first");
        hipError_t e = hipMemRelease(a->handles[c]);
        if (e != hipSuccess) return ahip(e, "dfx_arena_release_chunks: hipMemRelease");
        a->live[c] = 0;
    }
    return 0;
}

int dfx_arena_destroy(dfx_arena *a) {
    if (!a) return 0;
    int rc = 0;
    {
        std::lock_guard<std::mutex> lk(a->mu);
        if (!a->views.empty()) (void)hipDeviceSynchronize();
        for (const View &v : a->views) {
            hipError_t e = hipMemUnmap(v.base, v.bytes);
            if (e == hipSuccess) e = hipMemAddressFree(v.base, v.bytes);
            if (e != hipSuccess && !rc) rc = ahip(e, "dfx_arena_destroy: unmap");
        }
        a->views.clear();
        for (size_t c = 0; c < a->handles.size(); ++c) {
            if (!a->live[c]) continue;
            hipError_t e = hipMemRelease(a->handles[c]);
            if (e != hipSuccess && !rc) rc = ahip(e, "dfx_arena_destroy: hipMemRelease");
            a->live[c] = 0;
        }
    }
    delete a;
    return rc;
}

}  // extern "C"
