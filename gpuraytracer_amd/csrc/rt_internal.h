// Internal definitions shared by the gpurt translation units (not part of the C-ABI).
#ifndef GPURT_INTERNAL_H
#define GPURT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpurt.h"

struct rt_device {
    int gpu;
    hipStream_t stream;
    bool owns_stream;
    // workspace of the radix sort's count/scan/scatter triple (per-tile digit counts)
    uint32_t* sort_table;      // [256][tiles], digit-major
    size_t sort_table_words;
    uint32_t* sort_totals;     // [256] digit totals + [256] exclusive digit bases
    unsigned long long* trace_stats;   // optional traversal counters (rt_device_set_trace_stats)
    // workspace of the wavefront trace: closest-hit records between the traversal and shading kernels
    float4* hit_records;       // [hit_capacity] (t, u, v, facing)
    uint32_t* hit_ci;          // [hit_capacity] first index slot of the hit triangle
    size_t hit_capacity;
    uint32_t* queue_counter;   // ray queue heads of the persistent traversal kernel (256 bytes)
    uint2* ray_keys_a;         // [hit_capacity] (locality key, ray index) pairs: the order rays are handed out in
    uint2* ray_keys_b;
};

struct rt_buffer {
    rt_device* dev;
    void* ptr;
    size_t bytes;
    bool owned;
};

struct rt_event {
    rt_device* dev;
    hipEvent_t ev;
};

struct rt_accel {
    rt_device* dev;
    uint32_t max_prims;
    uint32_t num_prims;        // set by rt_accel_pack
    float4* tris;              // [3*max_prims] packed positions, original order
    float4* sorted_tris;       // [3*max_prims] packed positions, Morton order (w of vertex 0 = 3*tri)
    rt_aabb* pairs;            // private copy of the hierarchy as interleaved sibling pairs (rt_levels::aoff placement)
    uint2* keys_a;             // [max_prims] (code, 3*tri)
    uint2* keys_b;             // [max_prims]
};

#define RT_HIP(call)                                              \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return RT_ERR_HIP_BASE + (int)e_;   \
    } while (0)

static inline int rt_set_device(const rt_device* dev) {
    if (!dev) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP(hipSetDevice(dev->gpu));
    return RT_OK;
}

static inline int rt_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RT_OK : RT_ERR_HIP_BASE + (int)e;
}

// radix sort tiling: 256 threads x RT_SORT_ITEMS keys per workgroup
#define RT_SORT_THREADS 256
#define RT_SORT_ITEMS 8
#define RT_SORT_TILE (RT_SORT_THREADS * RT_SORT_ITEMS)

int rt_sort_workspace(rt_device* dev, uint32_t num_elements);   // grows dev->sort_table as needed
// Stable LSD radix sort of (key, value) pairs by the low 8 * passes key bits, queued on the device's stream.
// a and b hold n pairs each and are both overwritten; *sorted receives whichever of them holds the result.
int rt_sort_pairs(rt_device* dev, uint2* a, uint2* b, uint32_t n, uint32_t passes, uint2** sorted);

// level schedule of the balanced hierarchy (src/RaytracerPipeline.cpp:465-467, 497-559):
// level 0 = leaves at node 1; level k has ceil(n_{k-1} / 2) nodes appended after level k-1;
// the level holding a single node is the root and is stored at node 0.
//
// aoff is a second, private placement of the same nodes used by the accel's own copy: every level starts at
// an even 32-byte slot, so the sibling pair (2j, 2j+1) of any level shares one 64-byte line (slots aoff + 2j and
// aoff + 2j + 1), which holds the two boxes with their components interleaved.
struct rt_levels {
    uint32_t count;            // number of levels including the root level
    uint32_t n[33];            // nodes per level
    uint32_t off[33];          // first node index of each level (root level: 0)
    uint32_t aoff[33];         // first slot of each level in the pair-aligned copy (root level: 0)
};
static inline bool rt_make_levels(uint32_t num_prims, rt_levels* lv) {
    if (num_prims < 3u) return false;
    uint32_t n = (num_prims + 1u) / 2u, off = 1u, aoff = 2u, k = 0;
    for (;;) {
        lv->n[k] = n;
        lv->off[k] = (n == 1u) ? 0u : off;
        lv->aoff[k] = (n == 1u) ? 0u : aoff;
        k++;
        if (n == 1u) break;
        off += n;
        aoff += (n + 1u) & ~1u;
        n = (n + 1u) / 2u;
    }
    lv->count = k;
    return true;
}
// slots the pair-aligned copy needs
static inline size_t rt_aligned_node_slots(const rt_levels* lv) {
    size_t slots = 2;
    for (uint32_t k = 0; k + 1u < lv->count; k++) slots += (lv->n[k] + 1u) & ~1u;
    return slots;
}

#endif
