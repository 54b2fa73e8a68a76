// Device, stream, event and buffer management of the gpurt C-ABI.
// Stands in for the reference's DX12Device (src/GPUDevice.cpp:40-91), GPUScheduler
// (src/GPUScheduler.cpp:130-166) and the buffer wrappers of src/ShaderResources.h.

#include <cstdlib>
#include <cstring>
#include <new>

#include "rt_internal.h"

extern "C" {

int rt_device_create(int gpu_index, rt_device** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    RT_HIP(hipGetDeviceCount(&count));
    if (gpu_index < 0 || gpu_index >= count) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP(hipSetDevice(gpu_index));
    rt_device* d = new (std::nothrow) rt_device();
    if (!d) return RT_ERR_OUT_OF_MEMORY;
    d->gpu = gpu_index;
    d->stream = nullptr;
    d->owns_stream = true;
    d->sort_table = nullptr;
    d->sort_table_words = 0;
    d->sort_totals = nullptr;
    d->trace_stats = nullptr;
    d->hit_records = nullptr;
    d->hit_ci = nullptr;
    d->hit_capacity = 0;
    d->queue_counter = nullptr;
    d->bounds_partials = nullptr;
    d->ray_keys_a = nullptr;
    d->ray_keys_b = nullptr;
    d->listed_rays = nullptr;
    d->split_cand = nullptr; d->split_count = nullptr; d->listed_again = nullptr;
    d->listed_seen_host = nullptr; d->listed_seen_dev = nullptr;
    d->listed_split = 0;
    d->walk_stats = nullptr;
    d->walk_stats_hist = false;
    d->conv_pairs = nullptr;
    d->conv_tris = nullptr;
    d->conv_quads = nullptr;
    d->conv_capacity = 0;
    {   // initial choice of the traversal kernel: GPURT_TRACE_KERNEL = spatial (default) | quad | wave | simple
        const char* v = std::getenv("GPURT_TRACE_KERNEL");
        d->trace_kernel = (v && !std::strcmp(v, "quad"))     ? RT_TRACE_KERNEL_QUAD
                          : (v && !std::strcmp(v, "wave"))   ? RT_TRACE_KERNEL_WAVE
                          : (v && !std::strcmp(v, "simple")) ? RT_TRACE_KERNEL_SIMPLE
                                                             : RT_TRACE_KERNEL_SPATIAL;
    }
    d->slab_division = false;
    d->drop_missed = false;
    d->missed = nullptr;
    d->missed_valid = false;
    d->keys_ready = false;
    d->missed_rays = nullptr;
    d->missed_count = 0;
    d->listed_valid = false;
    d->walk_entry_margin = 1.00390625f;      // 1 + 2^-8
    d->walk_sliver_pad = 1.0f / 32.0f;
    {   // tuning of the traversal kernels: defaults, or the environment's values (read here, once per device)
        auto env = [](const char* name, uint32_t fallback) -> uint32_t {
            const char* v = std::getenv(name);
            return (v && *v) ? (uint32_t)std::strtoul(v, nullptr, 10) : fallback;
        };
        rt_trace_tuning t;
        t.leaf_threshold = env("GPURT_LEAF_THRESHOLD", 0u);
        t.refill_threshold = env("GPURT_REFILL_THRESHOLD", 0u);
        t.blocks_per_cu = env("GPURT_BLOCKS_PER_CU", 4u);
        t.sort_rays = env("GPURT_SORT_RAYS", 2u);
        t.deal_shift = env("GPURT_DEAL_SHIFT", 6u);
        t.cost_bits = env("GPURT_COST_BITS", 6u);
        t.cost_depth = env("GPURT_COST_DEPTH", 0u);
        t.cost_stride = env("GPURT_COST_STRIDE", 0u);
        t.cost_chord = env("GPURT_COST_CHORD", 2u);
        t.key_passes = env("GPURT_KEY_PASSES", 0u);
        t.stuck_threshold = env("GPURT_STUCK_THRESHOLD", 0u);
        t.retire_us = env("GPURT_RETIRE_US", 40u);
        d->tuning = rt_clamp_tuning(t);
        const uint32_t in_parts = env("GPURT_LISTED_SPLIT", 0u);      // rt_device_set_listed_split
        d->listed_split = in_parts <= 2u ? (int)in_parts : 0;
    }
    {
        hipDeviceProp_t prop;
        hipError_t pe = hipGetDeviceProperties(&prop, gpu_index);
        if (pe != hipSuccess) { delete d; return RT_ERR_HIP_BASE + (int)pe; }
        d->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    }
    d->time_traverse = false;
    d->timing_events = nullptr;
    d->timing_count = d->timing_capacity = 0;
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete d; return RT_ERR_HIP_BASE + (int)e; }
    e = hipMalloc((void**)&d->sort_totals, 512 * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipStreamDestroy(d->stream); delete d; return RT_ERR_HIP_BASE + (int)e; }
    const int src = rt_static_scratch_init(gpu_index);      // this GPU is current: the address of its static scratch
    if (src) { (void)hipFree(d->sort_totals); (void)hipStreamDestroy(d->stream); delete d; return src; }
    *out = d;
    return RT_OK;
}

int rt_device_destroy(rt_device* dev) {
    if (!dev) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    (void)hipStreamSynchronize(dev->stream);
    if (dev->sort_table) (void)hipFree(dev->sort_table);
    if (dev->sort_totals) (void)hipFree(dev->sort_totals);
    if (dev->hit_records) (void)hipFree(dev->hit_records);
    if (dev->hit_ci) (void)hipFree(dev->hit_ci);
    if (dev->queue_counter) (void)hipFree(dev->queue_counter);
    if (dev->bounds_partials) (void)hipFree(dev->bounds_partials);
    if (dev->ray_keys_a) (void)hipFree(dev->ray_keys_a);
    if (dev->ray_keys_b) (void)hipFree(dev->ray_keys_b);
    if (dev->listed_rays) (void)hipFree(dev->listed_rays);
    if (dev->split_cand) (void)hipFree(dev->split_cand);
    if (dev->split_count) (void)hipFree(dev->split_count);
    if (dev->listed_again) (void)hipFree(dev->listed_again);
    if (dev->listed_seen_host) (void)hipHostFree(dev->listed_seen_host);
    if (dev->missed) (void)hipFree(dev->missed);
    if (dev->conv_pairs) (void)hipFree(dev->conv_pairs);
    if (dev->conv_tris) (void)hipFree(dev->conv_tris);
    if (dev->conv_quads) (void)hipFree(dev->conv_quads);
    for (uint32_t i = 0; i < 2u * dev->timing_capacity; i++) (void)hipEventDestroy(dev->timing_events[i]);
    delete[] dev->timing_events;
    if (dev->owns_stream && dev->stream) (void)hipStreamDestroy(dev->stream);
    delete dev;
    return RT_OK;
}

int rt_device_sync(rt_device* dev) {
    int rc = rt_set_device(dev);
    if (rc) return rc;
    RT_HIP(hipStreamSynchronize(dev->stream));
    return RT_OK;
}

void* rt_device_stream(rt_device* dev) { return dev ? (void*)dev->stream : nullptr; }

int rt_device_set_stream(rt_device* dev, void* hip_stream) {
    int rc = rt_set_device(dev);
    if (rc) return rc;
    RT_HIP(hipStreamSynchronize(dev->stream));
    if (dev->owns_stream && dev->stream) (void)hipStreamDestroy(dev->stream);
    dev->stream = (hipStream_t)hip_stream;
    dev->owns_stream = false;
    return RT_OK;
}

int rt_device_index(rt_device* dev) { return dev ? dev->gpu : -1; }

int rt_device_count(int* count) {
    if (!count) return RT_ERR_INVALID_ARGUMENT;
    *count = 0;
    RT_HIP(hipGetDeviceCount(count));
    return RT_OK;
}

const char* rt_error_string(int code) {
    switch (code) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case RT_ERR_TOO_FEW_PRIMITIVES: return "fewer than 3 primitives";
        case RT_ERR_BUFFER_TOO_SMALL: return "buffer too small";
        case RT_ERR_OUT_OF_MEMORY: return "out of memory";
        case RT_ERR_LIMIT_EXCEEDED: return "size beyond a fixed limit (of an entry point that allocates nothing, or of the second hierarchy)";
        default: break;
    }
    if (code == RT_ERR_RCCL_UNAVAILABLE) return "librccl.so.1 could not be loaded";
    if (code >= RT_ERR_RCCL_BASE) return rt_comm_error_string(code);
    if (code >= RT_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(code - RT_ERR_HIP_BASE));
    return "unknown error";
}

int rt_event_create(rt_device* dev, rt_event** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rt_event* e = new (std::nothrow) rt_event();
    if (!e) return RT_ERR_OUT_OF_MEMORY;
    e->dev = dev;
    hipError_t he = hipEventCreate(&e->ev);
    if (he != hipSuccess) { delete e; return RT_ERR_HIP_BASE + (int)he; }
    *out = e;
    return RT_OK;
}

int rt_event_record(rt_event* ev) {
    if (!ev) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(ev->dev);
    if (rc) return rc;
    RT_HIP(hipEventRecord(ev->ev, ev->dev->stream));
    return RT_OK;
}

int rt_event_elapsed_ms(rt_event* start, rt_event* stop, float* ms) {
    if (!start || !stop || !ms) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(stop->dev);
    if (rc) return rc;
    RT_HIP(hipEventSynchronize(stop->ev));
    RT_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return RT_OK;
}

int rt_event_destroy(rt_event* ev) {
    if (!ev) return RT_ERR_INVALID_ARGUMENT;
    (void)rt_set_device(ev->dev);
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return RT_OK;
}

int rt_buffer_alloc(rt_device* dev, size_t bytes, rt_buffer** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rt_buffer* b = new (std::nothrow) rt_buffer();
    if (!b) return RT_ERR_OUT_OF_MEMORY;
    b->dev = dev;
    b->bytes = bytes;
    b->owned = true;
    b->ptr = nullptr;
    hipError_t e = hipMalloc(&b->ptr, bytes ? bytes : 16);
    if (e != hipSuccess) { delete b; return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e; }
    e = hipMemsetAsync(b->ptr, 0, bytes ? bytes : 16, dev->stream);
    if (e != hipSuccess) { (void)hipFree(b->ptr); delete b; return RT_ERR_HIP_BASE + (int)e; }
    *out = b;
    return RT_OK;
}

int rt_buffer_wrap(rt_device* dev, void* device_ptr, size_t bytes, rt_buffer** out) {
    if (!out || !dev || (!device_ptr && bytes)) return RT_ERR_INVALID_ARGUMENT;
    rt_buffer* b = new (std::nothrow) rt_buffer();
    if (!b) return RT_ERR_OUT_OF_MEMORY;
    b->dev = dev;
    b->ptr = device_ptr;
    b->bytes = bytes;
    b->owned = false;
    *out = b;
    return RT_OK;
}

int rt_buffer_upload(rt_buffer* buf, const void* src, size_t bytes, size_t dst_offset) {
    if (!buf || (!src && bytes)) return RT_ERR_INVALID_ARGUMENT;
    if (dst_offset > buf->bytes || bytes > buf->bytes - dst_offset) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(buf->dev);
    if (rc) return rc;
    if (!bytes) return RT_OK;
    // the copy is complete on return, so the caller may reuse src at once
    RT_HIP(hipMemcpyAsync((char*)buf->ptr + dst_offset, src, bytes, hipMemcpyHostToDevice, buf->dev->stream));
    RT_HIP(hipStreamSynchronize(buf->dev->stream));
    return RT_OK;
}

int rt_buffer_readback(rt_buffer* buf, void* dst, size_t bytes, size_t src_offset) {
    if (!buf || (!dst && bytes)) return RT_ERR_INVALID_ARGUMENT;
    if (src_offset > buf->bytes || bytes > buf->bytes - src_offset) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(buf->dev);
    if (rc) return rc;
    if (!bytes) return RT_OK;
    RT_HIP(hipMemcpyAsync(dst, (const char*)buf->ptr + src_offset, bytes, hipMemcpyDeviceToHost, buf->dev->stream));
    RT_HIP(hipStreamSynchronize(buf->dev->stream));
    return RT_OK;
}

int rt_buffer_clear(rt_buffer* buf) {
    if (!buf) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(buf->dev);
    if (rc) return rc;
    if (buf->bytes) RT_HIP(hipMemsetAsync(buf->ptr, 0, buf->bytes, buf->dev->stream));
    return RT_OK;
}

int rt_buffer_free(rt_buffer* buf) {
    if (!buf) return RT_ERR_INVALID_ARGUMENT;
    if (buf->owned) {
        (void)rt_set_device(buf->dev);
        (void)hipStreamSynchronize(buf->dev->stream);
        (void)hipFree(buf->ptr);
    }
    delete buf;
    return RT_OK;
}

void* rt_buffer_ptr(rt_buffer* buf) { return buf ? buf->ptr : nullptr; }
size_t rt_buffer_size(rt_buffer* buf) { return buf ? buf->bytes : 0; }

}  // extern "C"

int rt_sort_workspace(rt_device* dev, uint32_t num_elements) {
    size_t tiles = ((size_t)num_elements + RT_SORT_TILE - 1) / RT_SORT_TILE;
    size_t words = 256 * (tiles ? tiles : 1);
    if (words <= dev->sort_table_words) return RT_OK;
    RT_HIP(hipStreamSynchronize(dev->stream));
    if (dev->sort_table) { (void)hipFree(dev->sort_table); dev->sort_table = nullptr; dev->sort_table_words = 0; }
    hipError_t e = hipMalloc((void**)&dev->sort_table, words * sizeof(uint32_t));
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
    dev->sort_table_words = words;
    return RT_OK;
}
