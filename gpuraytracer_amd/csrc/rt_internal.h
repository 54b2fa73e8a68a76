// Internal definitions shared by the gpurt translation units (not part of the C-ABI).
#ifndef GPURT_INTERNAL_H
#define GPURT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpurt.h"
#include "gpurt_debug.h"

struct rt_device {
    int gpu;
    hipStream_t stream;
    bool owns_stream;
    // workspace of the radix sort's count/scan/scatter triple (per-tile digit counts)
    uint32_t* sort_table;      // [256][tiles], digit-major
    size_t sort_table_words;
    uint32_t* sort_totals;     // [256] digit totals + [256] exclusive digit bases
    unsigned long long* trace_stats;   // optional traversal counters (rt_device_set_trace_stats)
    bool walk_stats_hist;              // the walk_stats buffer holds 20 counters (steps by depth behind the first four)
    unsigned long long* walk_stats;    // optional counters of the walk on the library's own hierarchy (rt_device_set_walk_stats)
    // workspace of the wavefront trace: closest-hit records between the traversal and shading kernels
    float4* hit_records;       // [hit_capacity] (t, u, v, facing)
    uint32_t* hit_ci;          // [hit_capacity] first index slot of the hit triangle
    size_t hit_capacity;
    uint32_t* queue_counter;   // ray queue heads of the two traversal launches (256 bytes each), length of the list between them
    uint2* listed_rays;        // [hit_capacity] (0, ray index): rays the second traversal launch walks again
    uint2* ray_keys_a;         // [hit_capacity] (locality key, ray index) pairs: the order rays are handed out in
    uint2* ray_keys_b;
    // private copy of the hierarchy for trace calls that come without an accel (rebuilt by every such call)
    rt_aabb* conv_pairs;
    float4* conv_tris;
    float4* conv_quads;
    uint32_t conv_capacity;    // primitives the two buffers are sized for
    float* bounds_partials;    // [kBoundsBlocks][6] per-workgroup centroid bounds of the scene-bounds reduction
    // optional event pairs around the traversal kernel (rt_device_set_traverse_timing)
    int trace_kernel;          // RT_TRACE_KERNEL_*: which traversal kernel rt_trace_rays launches
    bool slab_division;        // rt_device_set_slab_division: the box test divides by the direction, as the reference writes it
    float walk_entry_margin;   // rt_device_set_walk_margins: SPATIAL kernel enters boxes up to reach * this
    float walk_sliver_pad;     // ... and the builds widen own leaf boxes of slivers by this part of the longest edge
    rt_trace_tuning tuning;    // rt_device_set_trace_tuning (initial values: defaults or GPURT_* environment variables)
    int compute_units;         // of this GPU: the persistent kernels launch a multiple of it
    bool drop_missed;          // rt_device_set_drop_missed_rays
    uint8_t* missed;           // [hit_capacity] one mark per ray: it missed in an earlier trace call of the current rays
    bool keys_ready;           // ray_keys_a holds the keys of the rays the last trace call left behind (made by its shading kernel)
    const void* keys_rays;     // ... of this ray buffer,
    const void* keys_accel;    // ... for this accel's own hierarchy (any build of it clears keys_ready),
    uint32_t keys_count, keys_cost_bits, keys_bits;     // this many rays, with this key layout
    bool missed_valid;         // the marks belong to the rays of the last trace call (false: cleared before the next one)
    const void* missed_rays;   // ... whose ray buffer and ray count these were
    uint32_t missed_count;
    bool listed_valid;         // the last trace call ran the SPATIAL kernel, so queue_counter[128] is its list length
    // A short list is walked in parts (one subtree of the reference's hierarchy per lane, see k_traverse_quad<true>):
    float4* split_cand;        // [RT_SPLIT_RAYS << 6][RT_SPLIT_CANDIDATES][2] what the parts found, in the order found
    uint32_t* split_count;     // [RT_SPLIT_RAYS << 6] how many (more than RT_SPLIT_CANDIDATES: the part ran out of room)
    uint2* listed_again;       // [RT_SPLIT_RAYS] rays with such a part: walked whole after all
    uint32_t* listed_seen_host;    // host-mapped word: the list length a recent trace call reported (what the next call expects)
    uint32_t* listed_seen_dev;     // ... the address kernels write it through
    int listed_split;          // rt_device_set_listed_split: 0 = by expectation, 1 = always in parts, 2 = never
    bool time_traverse;
    hipEvent_t* timing_events;      // [2 * timing_capacity]: start, stop, start, stop, ...
    uint32_t timing_count, timing_capacity;
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

// the ranges rt_device_set_trace_tuning accepts
static inline rt_trace_tuning rt_clamp_tuning(rt_trace_tuning t) {
    if (t.leaf_threshold > 64u) t.leaf_threshold = 64u;
    if (t.refill_threshold > 64u) t.refill_threshold = 64u;
    if (t.blocks_per_cu < 1u) t.blocks_per_cu = 1u;
    if (t.blocks_per_cu > 8u) t.blocks_per_cu = 8u;
    if (t.sort_rays > 2u) t.sort_rays = 1u;
    if (t.deal_shift < 6u) t.deal_shift = 6u;
    if (t.deal_shift > 20u) t.deal_shift = 20u;
    if (t.cost_bits < 6u) t.cost_bits = 6u;
    if (t.cost_bits > 12u) t.cost_bits = 12u;
    if (t.cost_depth == 1u) t.cost_depth = 2u;
    if (t.cost_depth > 30u) t.cost_depth = 30u;
    if (t.cost_chord > 2u) t.cost_chord = 2u;
    if (t.key_passes > 3u) t.key_passes = 3u;
    if (t.stuck_threshold > 64u) t.stuck_threshold = 64u;
    if (t.retire_us > 1000u) t.retire_us = 1000u;
    return t;
}

static inline int rt_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RT_OK : RT_ERR_HIP_BASE + (int)e;
}

// rows first_row, first_row + row_stride, ... (num_rows of them) lie inside an image of the given height
static inline bool rt_rows_fit(uint32_t first_row, uint32_t num_rows, uint32_t row_stride, uint32_t height) {
    if (!row_stride || first_row > height) return false;
    if (!num_rows) return true;
    return (uint64_t)first_row + (uint64_t)(num_rows - 1u) * row_stride < height;
}

// listed rays walked in parts: lists up to this length, and the candidates a part can hold
#define RT_SPLIT_RAYS 512u
#define RT_SPLIT_CANDIDATES 6u

// radix sort tiling: 256 threads x RT_SORT_ITEMS keys per workgroup
#define RT_SORT_THREADS 256
#ifndef RT_SORT_ITEMS
#define RT_SORT_ITEMS 8
#endif
#define RT_SORT_TILE (RT_SORT_THREADS * RT_SORT_ITEMS)

int rt_sort_workspace(rt_device* dev, uint32_t num_elements);   // grows dev->sort_table as needed
// Looks up, once per GPU and with that GPU current, the device address of the static scratch array behind
// rt_scene_bounds_static (defined beside the array in rt_build.hip); rt_device_create calls it.
int rt_static_scratch_init(int gpu);
// text of an RT_ERR_RCCL_BASE + ncclResult_t code (rt_comm.hip; loads RCCL if it is not loaded yet)
const char* rt_comm_error_string(int code);
// Stable LSD radix sort of (key, value) pairs by the low 8 * passes key bits, queued on the device's stream.
// a and b hold n pairs each and are both overwritten; *sorted receives whichever of them holds the result.
// first_shift: the lowest key bit that counts.
int rt_sort_pairs(rt_device* dev, uint2* a, uint2* b, uint32_t n, uint32_t passes, uint2** sorted, bool first_counted = false,
                  uint32_t first_shift = 0u);
// Writes the private copy (interleaved sibling pairs, Morton-ordered packed triangles: see rt_levels and
// rt_accel) of a hierarchy held in the reference's layout. pairs: rt_aligned_node_slots() slots;
// sorted_tris: 5 float4 per leaf.
// The sibling-pair records of an accel's hierarchy (own = the library's own one), made from its nodes on first use after
// a build: only the one-level kernel, the counting variants and the ray-order probe read them.
int rt_accel_pairs(rt_accel* accel, bool own, const rt_aabb* nodes, const float4** pairs);
int rt_private_copy_from_reference(rt_device* dev, uint32_t num_prims, const rt_aabb* nodes, const uint32_t* indices,
                                   const rt_vertex* verts, rt_aabb* pairs, float4* sorted_tris, float4* quads);

// level schedule of the balanced hierarchy (src/RaytracerPipeline.cpp:465-467, 497-559):
// level 0 = leaves at node 1; level k has ceil(n_{k-1} / 2) nodes appended after level k-1;
// the level holding a single node is the root and is stored at node 0.
//
// aoff is a second, private placement of the same nodes used by the accel's own copy: node i of level k sits in
// 32-byte slot 2^(root level - k) + i, the numbering of a complete binary tree (the node in slot h has its
// children in slots 2h and 2h + 1; a level never holds more nodes than such a tree has room for). The two slots
// 2h, 2h + 1 form one 64-byte record that holds both children's boxes with their components interleaved, so a
// walk only needs h to find the record of a node's children. A child that does not exist (the last node of a
// level with an odd count has one child) is stored as a box no ray can hit.
struct rt_levels {
    uint32_t count;            // number of levels including the root level
    uint32_t n[33];            // nodes per level
    uint32_t off[33];          // first node index of each level (root level: 0)
    uint32_t aoff[33];         // first slot of each level in the private copy: 2^(count - 1 - k)
};
static inline bool rt_make_levels(uint32_t num_prims, rt_levels* lv) {
    if (num_prims < 3u) return false;
    uint32_t n = (num_prims + 1u) / 2u, off = 1u, k = 0;
    for (;;) {
        lv->n[k] = n;
        lv->off[k] = (n == 1u) ? 0u : off;
        k++;
        if (n == 1u) break;
        off += n;
        n = (n + 1u) / 2u;
    }
    lv->count = k;
    for (uint32_t j = 0; j < k; j++) lv->aoff[j] = 1u << (k - 1u - j);
    return true;
}
// Leaf slots of the library's own hierarchy for a scene of num_prims triangles: a power of two, so that every node
// covers an aligned run of slots, and twice the leaves while that stays within RT_SPATIAL_SLOTS_MAX - the room lets
// whole Morton cells start on aligned slots (see k_cell_choose). The maximum is what k_traverse_spatial addresses:
// its 64-byte records of four grandchildren by 32-bit byte offsets (2^25 records), its leaves by 28-bit numbers.
constexpr uint32_t RT_SPATIAL_SLOTS_MAX = 1u << 26;
static inline uint32_t rt_spatial_slots(uint32_t num_prims) {
    const uint32_t leaves = (num_prims + 1u) / 2u;
    uint32_t p = 2u;
    while (p < leaves) p <<= 1;
    if (p <= RT_SPATIAL_SLOTS_MAX / 2u) p <<= 1;
    return p;
}
constexpr uint32_t RT_CELL_BITS_MAX = 21u;     // finest Morton cells leaf slots are aligned to: 2^21
// A leaf's two triangles take 80 bytes; leaves are laid out one per 128-byte line, because past the L2 a read costs
// per line touched, not per byte (measured: random 128-byte records are fetched at the rate of 64-byte ones),
// and an 80-byte stride makes every second leaf straddle two lines.
#define RT_LEAF_STRIDE 8u      // in float4
// The traversal's second copy: the node in slot h has a 64-byte record at quads[RT_QUAD_STRIDE * h] with boxes of its
// four grandchildren (slots 4h .. 4h + 3), so that one read takes the walk two levels down. The boxes are in binary16
// and CONTAIN the true ones (minima rounded down, maxima rounded up): the walk may enter a node the reference would
// not, never the other way round, and every leaf it reaches is tested against its exact box again (recomputed from
// the leaf's triangles) before its triangles are. Only nodes an even number of levels above the leaves have records
// (slots of the other depths are left unused); grandchildren that do not exist are NaNs.
#define RT_QUAD_STRIDE 4u      // in float4
// number of records: one per slot below 2^(root level - 1)
static inline size_t rt_quad_records(const rt_levels* lv) { return lv->count >= 3u ? (size_t)1u << (lv->count - 2u) : 1u; }
// 32-byte slots the private copy needs
static inline size_t rt_aligned_node_slots(const rt_levels* lv) { return (size_t)2u << (lv->count - 1u); }

struct rt_accel {
    rt_device* dev;
    uint32_t max_prims;
    uint32_t num_prims;        // set by rt_accel_pack
    float4* tris;              // [3*max_prims] packed positions, original order
    float4* sorted_tris;       // [RT_LEAF_STRIDE * leaves] the two triangles of every leaf in Morton order, interleaved
    float4* quads;             // second private copy of the hierarchy, four grandchildren per 64-byte record (see rt_quad_records)
    rt_aabb* pairs;            // private copy of the hierarchy as interleaved sibling pairs (rt_levels::aoff placement)
    uint2* keys_a;             // [max_prims] (code, 3*tri)
    uint2* keys_b;             // [max_prims]
    // The library's own hierarchy over the same triangles (same kernels, same shape, codes whose bits do not
    // overlap), for the traversal that does not follow the reference's order: nodes, sibling pairs, records of
    // four grandchildren, leaves (whose records also hold their triangles' places in the reference's order).
    rt_aabb* sp_nodes;
    rt_aabb* sp_pairs;
    float4* sp_quads;
    float4* sp_tris;
    bool pairs_valid, sp_pairs_valid;      // pairs / sp_pairs hold the records of the last build (made on demand, see rt_accel_pairs)
    uint32_t* cells;           // Morton-cell bookkeeping of the last build: see the offsets below
    bool want_spatial;         // rt_accel_set_second_hierarchy: builds make the library's own hierarchy
    uint32_t cell_bits_cap;    // ... with leaf slots aligned to Morton cells of at most this many bits
    bool sp_built;             // the last build made it
    bool sp_nodes_full;        // ... and every level of it is in sp_nodes (builds store only the levels something reads)
    uint32_t built_prims;      // triangles of the last build (0: none yet) - what rt_accel_refit may refit
    uint2* sp_sorted;          // keys_a or keys_b: the own hierarchy's sorted (key, place) pairs of the last build
    rt_levels sp_lv;           // shape of that hierarchy at the last build: rt_spatial_slots(num_prims) leaves, all levels full
    float4* outlier_leaves;    // [RT_LEAF_STRIDE * RT_OUTLIERS_MAX / 2] the outliers (see RT_CELLS_OUTLIERS) as leaf records, two per record
};
// rt_accel::cells, in uint32 (everything before RT_CELLS_START is cleared by every build): [0..7] layout {cell bits T, largest cell at T = 21, 18, ... 3}; then the triangles per cell
// of the finest level, the cells of 15 bits summed from them, and the first rank of every cell at the chosen level
constexpr uint32_t RT_CELLS_HIST = 8u;
constexpr uint32_t RT_CELLS_HIST12 = RT_CELLS_HIST + (1u << RT_CELL_BITS_MAX);          // cells of RT_CELL_BITS_MAX - 9 bits, then the fullest cells of - 6 bits
constexpr uint32_t RT_CELLS_START = RT_CELLS_HIST12 + (1u << (RT_CELL_BITS_MAX - 6u));
// ... and, per axis, how many centroids fall into each of 4096 equal steps of the scene bounds, turned into the
// 10-bit coordinate the keys are made of (k_axis_map)
constexpr uint32_t RT_AXIS_STEPS = 4096u;
constexpr uint32_t RT_CELLS_AXIS = RT_CELLS_START + (1u << RT_CELL_BITS_MAX) + 1u;
// ... and the outliers of the last build: triangles whose box is a large part of the scene's (a wall, a floor). In a
// hierarchy they make every node above them as large as they are, and every ray meets those nodes; the library's own
// hierarchy leaves them out, and k_traverse_spatial tests them from a list before a ray's walk starts.
// [RT_CELLS_OUTLIERS] = how many were found (more than RT_OUTLIERS_MAX: the others stay in the hierarchy), then their
// places in the reference's order, in the order found.
constexpr uint32_t RT_OUTLIERS_MAX = 64u;
constexpr uint32_t RT_CELLS_OUTLIERS = RT_CELLS_AXIS + 3u * RT_AXIS_STEPS;
constexpr uint32_t RT_CELLS_WORDS = RT_CELLS_OUTLIERS + 1u + RT_OUTLIERS_MAX;

#endif
