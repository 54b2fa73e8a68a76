// Multi-GPU part of the gpurt C-ABI: which image rows a device holds, and the one exchange of a frame - finished rows
// gathered to a root device over RCCL (xGMI between the GPUs of a node) and put back in image order.
//
// The reference renders on one device and has no counterpart. The sharding is by rows dealt in turn (rank r of N holds
// rows r, r + N, ...): neighbouring rows cost the same, so every device gets the same load, and all rays of a pixel stay
// on one device. Scene and hierarchy are replicated and rebuilt per device, so a frame needs no other collective.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first rt_comm_unique_id / rt_comm_init): single-GPU users of
// the library neither link nor load it.

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include <rccl/rccl.h>

#include "rt_internal.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // First the RCCL that lies beside the HIP runtime this library is bound to. A process can hold two ROCm stacks - the
    // system's and the one a Python package brings along, under the same sonames -, and a plain dlopen("librccl.so.1")
    // returns whichever copy is loaded already: one bound to the other stack's HIP runtime finds no device.
    Dl_info hip;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &hip) && hip.dli_fname) {
        std::string dir(hip.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1u);
            for (const char* n : {"librccl.so.1", "librccl.so"}) {
                g_rccl.handle = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (g_rccl.handle) break;
            }
        }
    }
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (g_rccl.handle) break;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!g_rccl.handle) return;
    void* h = g_rccl.handle;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
    g_rccl.Send = (decltype(g_rccl.Send))dlsym(h, "ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))dlsym(h, "ncclRecv");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.Send &&
                g_rccl.Recv && g_rccl.GetErrorString;
}

const RcclApi* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

inline int rccl_status(ncclResult_t r) { return r == ncclSuccess ? RT_OK : RT_ERR_RCCL_BASE + (int)r; }

#define RT_RCCL(call)                                   \
    do {                                                \
        const int s_ = rccl_status(call);               \
        if (s_) return s_;                              \
    } while (0)

// rows of rank r when `height` rows are dealt to `world` ranks in turn
inline uint32_t rows_of(uint32_t height, uint32_t world, uint32_t rank) { return rank < height ? (height - rank + world - 1u) / world : 0u; }

// gathered: the ranks' rows one rank after the other (rank r: rows r, r + world, ...); image: rows in image order.
// One thread per float4 (pixel) of the image.
__global__ __launch_bounds__(256) void k_rows_interleave(const float4* __restrict__ gathered, uint32_t width, uint32_t height, uint32_t world,
                                                         float4* __restrict__ image) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t rank = y % world, k = y / world;
    // rows held by the ranks before this one: rank q < rank holds ceil((height - q) / world) rows
    size_t before = 0;
    for (uint32_t q = 0; q < rank; q++) before += q < height ? (height - q + world - 1u) / world : 0u;
    image[i] = gathered[(before + k) * width + x];
}

}  // namespace

struct rt_comm {
    rt_device* dev;
    ncclComm_t comm;
    int rank, world;
    float4* staging;           // root only: the gathered rows before they are put in image order; grows on demand
    size_t staging_bytes;
};

const char* rt_comm_error_string(int code) {
    const RcclApi* api = rccl();
    if (!api) return "RCCL error (librccl.so.1 could not be loaded)";
    return api->GetErrorString((ncclResult_t)(code - RT_ERR_RCCL_BASE));
}

extern "C" {

int rt_rows_for_rank(uint32_t height, uint32_t world, uint32_t rank, rt_image_rows* rows) {
    if (!rows || !world || rank >= world) return RT_ERR_INVALID_ARGUMENT;
    rows->first_row = rank < height ? rank : height;     // a rank beyond the image holds no row
    rows->num_rows = rows_of(height, world, rank);
    rows->row_stride = world;
    return RT_OK;
}

int rt_rows_interleave(rt_device* dev, const rt_buffer* gathered, uint32_t width, uint32_t height, uint32_t world, rt_buffer* image) {
    if (!dev || !gathered || !image || !world || !width) return RT_ERR_INVALID_ARGUMENT;
    const size_t pixels = (size_t)width * height;
    if (gathered->bytes < pixels * 16u || image->bytes < pixels * 16u) return RT_ERR_BUFFER_TOO_SMALL;
    if (gathered->ptr == image->ptr) return RT_ERR_INVALID_ARGUMENT;       // not in place
    if (!pixels) return RT_OK;
    if ((pixels + 255u) / 256u > 0x7fffffffull) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rows_interleave, dim3((uint32_t)((pixels + 255u) / 256u)), dim3(256), 0, dev->stream, (const float4*)gathered->ptr,
                       width, height, world, (float4*)image->ptr);
    return rt_launch_status();
}

int rt_comm_unique_id(void* unique_id) {
    if (!unique_id) return RT_ERR_INVALID_ARGUMENT;
    const RcclApi* api = rccl();
    if (!api) return RT_ERR_RCCL_UNAVAILABLE;
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "the id is handed around as RT_COMM_ID_BYTES opaque bytes");
    return rccl_status(api->GetUniqueId((ncclUniqueId*)unique_id));
}

int rt_comm_init(rt_device* dev, int rank, int world, const void* unique_id, rt_comm** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!dev || !unique_id || world < 1 || rank < 0 || rank >= world) return RT_ERR_INVALID_ARGUMENT;
    const RcclApi* api = rccl();
    if (!api) return RT_ERR_RCCL_UNAVAILABLE;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rt_comm* c = new (std::nothrow) rt_comm();
    if (!c) return RT_ERR_OUT_OF_MEMORY;
    c->dev = dev; c->comm = nullptr; c->rank = rank; c->world = world; c->staging = nullptr; c->staging_bytes = 0;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    rc = rccl_status(api->CommInitRank(&c->comm, world, id, rank));       // collective: returns once every rank has called
    if (rc) { delete c; return rc; }
    *out = c;
    return RT_OK;
}

int rt_comm_destroy(rt_comm* c) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    (void)rt_set_device(c->dev);
    (void)hipStreamSynchronize(c->dev->stream);
    const RcclApi* api = rccl();
    if (api && c->comm) (void)api->CommDestroy(c->comm);
    if (c->staging) (void)hipFree(c->staging);
    delete c;
    return RT_OK;
}

int rt_comm_rank(const rt_comm* c) { return c ? c->rank : -1; }
int rt_comm_world(const rt_comm* c) { return c ? c->world : 0; }

int rt_gather_rows(rt_comm* c, const rt_buffer* rows, uint32_t width, uint32_t height, int root, rt_buffer* image) {
    if (!c || !rows || !width || root < 0 || root >= c->world) return RT_ERR_INVALID_ARGUMENT;
    const RcclApi* api = rccl();
    if (!api) return RT_ERR_RCCL_UNAVAILABLE;
    const uint32_t world = (uint32_t)c->world, rank = (uint32_t)c->rank;
    const size_t mine = (size_t)rows_of(height, world, rank) * width;           // pixels this rank sends
    const size_t pixels = (size_t)width * height;
    if (rows->bytes < mine * 16u) return RT_ERR_BUFFER_TOO_SMALL;
    const bool is_root = c->rank == root;
    if (is_root && (!image || image->bytes < pixels * 16u)) return image ? RT_ERR_BUFFER_TOO_SMALL : RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(c->dev);
    if (rc) return rc;
    hipStream_t s = c->dev->stream;
    if (is_root && c->staging_bytes < pixels * 16u) {
        RT_HIP(hipStreamSynchronize(s));
        if (c->staging) { (void)hipFree(c->staging); c->staging = nullptr; c->staging_bytes = 0; }
        hipError_t e = hipMalloc((void**)&c->staging, pixels * 16u);
        if (e != hipSuccess) return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
        c->staging_bytes = pixels * 16u;
    }
    // one group: every rank sends its rows to the root (the root to itself, which RCCL turns into a copy on the device);
    // the root receives rank after rank into the staging area. 16 bytes per pixel, counted in floats.
    RT_RCCL(api->GroupStart());
    int first_error = RT_OK;
    if (mine) first_error = rccl_status(api->Send(rows->ptr, mine * 4u, ncclFloat, root, c->comm, s));
    if (is_root) {
        size_t at = 0;
        for (uint32_t q = 0; q < world; q++) {
            const size_t n = (size_t)rows_of(height, world, q) * width;
            if (n && !first_error) first_error = rccl_status(api->Recv(c->staging + at, n * 4u, ncclFloat, (int)q, c->comm, s));
            at += n;
        }
    }
    const int end = rccl_status(api->GroupEnd());
    if (first_error) return first_error;
    if (end) return end;
    if (!is_root || !pixels) return RT_OK;
    hipLaunchKernelGGL(k_rows_interleave, dim3((uint32_t)((pixels + 255u) / 256u)), dim3(256), 0, s, (const float4*)c->staging, width, height,
                       world, (float4*)image->ptr);
    return rt_launch_status();
}

}  // extern "C"
