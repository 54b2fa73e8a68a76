/* gpurt_debug -- instrumentation and test-only switches of the gpurt library.
 *
 * None of these has a counterpart in the reference, and none is part of what a maintainer binds in place of the
 * reference's Direct3D 12 layers (that is gpurt.h): counters and timers that benchmarks and profiles read, and two
 * margins that exist so that tests can show what they are for. Same conventions as gpurt.h.
 */
#ifndef GPURT_DEBUG_H
#define GPURT_DEBUG_H

#include "gpurt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Instrumentation (no counterpart in the reference): while a 32-byte buffer is set, every trace launch adds
 * {box tests, triangle tests, rays that hit, rays} as four uint64 to it (a slower kernel variant); NULL turns
 * it off. The counts fix the algorithmic bytes per ray of a workload. */
int rt_device_set_trace_stats(rt_device* dev, rt_buffer* stats);
/* Diagnostics (no counterpart in the reference): the two safety margins of the SPATIAL kernel. entry_margin: boxes are
 * entered up to reach * entry_margin (default 1 + 2^-8; 1 = no margin); sliver_pad: own leaf boxes of triangles whose
 * altitude is below 1/64 of their longest edge are widened by sliver_pad * that edge on every side at the next build
 * (default 1/32; 0 = no room). Smaller values make the kernel miss hits the reference finds - they exist so that tests
 * can show what the margins are for. Values outside [1, 2] and [0, 1] are refused. */
int rt_device_set_walk_margins(rt_device* dev, float entry_margin, float sliver_pad);
/* Instrumentation (no counterpart in the reference): how many rays of the device's last trace call the SPATIAL
 * kernel handed to the walk in the reference's order (0 after a call with another kernel). Waits for the stream. */
int rt_device_listed_rays(rt_device* dev, uint32_t* count);
/* ... how many of them were listed while they were walked (a closer hit whose leaf box in the reference's hierarchy
 * lets the ray in only beyond the reach of the closest hit so far) rather than at the end (another hit between the
 * closest hit and the entry of its leaf's box), and - where the list was walked in parts (rt_device_set_listed_split) -
 * how many had a part that ran out of room for its candidates and were walked whole after all. */
int rt_device_listed_rays_detail(rt_device* dev, uint32_t* total, uint32_t* during_walk, uint32_t* whole_after_parts);
/* Test switch (no counterpart in the reference): how the listed rays are walked in the reference's order. A short list
 * (up to 512 rays) can be dealt out in parts - one subtree of the reference's hierarchy per lane, put together afterwards,
 * same result - which takes two more kernel launches; 0 (default): only after a recent trace call on this device has
 * listed rays, 1: always, 2: never (one lane per listed ray). */
int rt_device_set_listed_split(rt_device* dev, int mode);
/* Instrumentation (no counterpart in the reference): while a 32-byte buffer is set, every launch of the SPATIAL kernel
 * adds {steps (one 64-byte record read and four boxes tested each), steps that hit none of the four, leaves whose
 * triangles were tested, turns of the waves' walking loops | triangle rounds of the waves << 32} as four uint64 to
 * it; NULL turns it off. The kernel's own algorithmic bytes come from the first three; steps / (64 * turns) and
 * leaves / (64 * rounds) are the shares of a wave's 64 lanes at work in a walking turn and in a triangle round
 * (the two halves of the fourth word hold 2^32 each: clear the buffer before they fill). A buffer of 160 bytes or more
 * also receives, as sixteen more uint64, the steps by depth of the entered node (entry d: nodes 2d levels below the
 * trunk). Triangles tested from the list of outliers are not counted. */
int rt_device_set_walk_stats(rt_device* dev, rt_buffer* stats);
/* Instrumentation (no counterpart in the reference): while enabled, every trace launch brackets its traversal
 * kernel with a pair of events on the device's stream. rt_device_traverse_timing waits for the pairs recorded
 * so far, returns the sum of their durations and their number, and forgets them. */
int rt_device_set_traverse_timing(rt_device* dev, int enable);
int rt_device_traverse_timing(rt_device* dev, double* total_ms, uint32_t* launches);
/* Instrumentation (no counterpart in the reference): the outliers of the accel's last build - triangles whose box has a
 * sixteenth or more of the surface of the box around all centroids. found: how many the build saw; listed: how many of
 * them (at most 64) left the library's own hierarchy for the list the SPATIAL kernel tests before a ray's walk. Both 0
 * after a build without the second hierarchy. Waits for the stream. */
int rt_accel_outliers(rt_accel* accel, uint32_t* found, uint32_t* listed);

#ifdef __cplusplus
}
#endif
#endif
