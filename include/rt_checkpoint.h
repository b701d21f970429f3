/* rt_checkpoint.h -- file format of a progressive render's checkpoint.
 *
 * The reference keeps the frame in memory and dumps it as output.ppm from its ctrl-c handler
 * (desktop/src/main.rs:970-974); a render cannot be continued.  Here a render is a sequence of absolute sample
 * ranges [spp_begin, spp_done) added into fp32 radiance SUMS (rt_render's accumulator), so the sums plus this
 * header are a complete, resumable state: Philox counters are keyed on absolute (pixel, sample, bounce), hence
 * continuing at spp_done renders exactly the samples an uninterrupted run would have rendered next.
 * One file per rank (each rank owns a disjoint sample range); little endian; header, then width*height*3 floats
 * in the accumulator's order (row 0 = bottom row, RGB interleaved).
 * Written by raytracing_amd/host/desktop_main.cpp (rt_desktop --checkpoint/--resume) and
 * raytracing_amd/progressive.py; both write to "<path>.tmp" and rename, so a kill never leaves half a file. */
#ifndef RT_CHECKPOINT_H
#define RT_CHECKPOINT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RT_CHECKPOINT_MAGIC "RTCKPT1"   /* 7 characters + NUL */

typedef struct rt_checkpoint_header {
    char magic[8];
    uint32_t width, height;
    uint32_t spp_begin;      /* first absolute sample index of this rank's range */
    uint32_t spp_done;       /* samples [spp_begin, spp_done) are in the sums */
    uint32_t spp_end;        /* end of this rank's range (the render is complete when spp_done == spp_end) */
    int32_t max_depth;
    int32_t integrator;      /* RT_INT_* */
    uint32_t rank, world;    /* informational: which shard of an spp-sharded render */
    uint32_t chunk;          /* samples added per fp32 addition into the sums (0 = not recorded): a resume with another chunk size is refused,
                              * because it would add the same samples in another grouping and the sums would differ in the last bits */
    uint64_t seed;
    uint64_t scene_id;       /* (blob length << 32) | CRC-32 (IEEE, zlib's) of the RTSB blob: resuming on another scene is refused */
} rt_checkpoint_header;      /* 64 bytes */

#ifdef __cplusplus
}
#endif
#endif
