/* rt_abi.h -- C ABI of librt_hip.so, the MI355X (gfx950) implementation of the reference's
 * `ray_colour` hot path.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * The reference (a Rust workspace, paths relative to /root/reference) has NO FFI/plugin interface:
 * the seam is the trait call `Integrator::ray_colour(&self, Ray, &Scene, i32) -> DVec3`
 * (renderer/src/integrator.rs:6-8) made once per camera sample from the tile loop
 * (desktop/src/main.rs:978-1019), plus `Hittable::hit` for `Scene` (renderer/src/scene.rs:31-38).
 * A per-ray FFI call is useless for a GPU, so the boundary is batched:
 *   rt_trace_closest  == Scene::hit            (scene.rs:31-38 -> bvh.rs:91-106 / hittable.rs:41-53) for n rays
 *   rt_render         == the spp loop of main.rs:991-1007 calling ray_colour (integrator.rs:31-397)
 *                        for every pixel over a range of sample indices, summed per pixel
 *   rt_resolve        == main.rs:1009-1013 (mean, aces_tonemapping main.rs:913-921) + Image::put NaN scrub (image.rs:43-59)
 *   rt_scene_upload   == handing `&Scene` (scene.rs:11-17) to the integrator, as one flattened blob
 * INTEGRATION.md shows the Rust `extern "C"` block that binds these.
 *
 * Conventions: 0 = ok, negative = error class; message via rt_last_error() (thread-local).  Every
 * pointer argument is caller-owned and only borrowed for the duration of the call.  Handles are
 * library-owned and freed only by their _destroy.  A context is bound to one device and is not
 * thread-safe: one host thread (or process) per GPU.  Pointers documented "device" must be device
 * memory of the context's GPU (e.g. a torch CUDA tensor's data_ptr); all others are host memory.
 */
#ifndef RT_ABI_H
#define RT_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_BAD_ARG (-1)
#define RT_ERR_HIP (-2)
#define RT_ERR_OOM (-3)
#define RT_ERR_UNSUPPORTED (-4)
#define RT_ERR_BAD_BLOB (-5)

#define RT_NONE 0xFFFFFFFFu

/* integrator ids (renderer/src/integrator.rs) */
#define RT_INT_MIS 0      /* MultipleImportanceSampleIntegrator   integrator.rs:29-171  */
#define RT_INT_LIGHT_IS 1 /* ImportanceSampleLightIntegrator      integrator.rs:172-260 */
#define RT_INT_BRDF 2     /* BRDFSampledPathIntegrator            integrator.rs:262-327 */
#define RT_INT_UNIFORM 3  /* UniformSampledPathIntegrator         integrator.rs:329-397 */

/* background ids: `fn(Ray)->DVec3` cannot cross a C ABI, so the two functions the reference
 * defines become an enum (desktop/src/main.rs:37-48) */
#define RT_BG_BLACK 0 /* no_light */
#define RT_BG_SKY 1   /* skybox   */

/* ---------------------------------------------------------------------------------------------
 * Flattened scene blob ("RTSB" v1): one contiguous, position-independent byte range -> one
 * hipMemcpy, one broadcast.  Produced host-side by librt_host.so (Scene::flatten); layout below.
 * All offsets are bytes from the start of the blob and 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
#define RT_BLOB_MAGIC 0x42535452u
#define RT_BLOB_VERSION 2u

/* Threaded BVH node, 64 B, pre-order (DFS, left first): the reference's fixed left-then-right
 * recursion (bvh.rs:91-106) needs no per-lane stack once every node stores where to go when its
 * box test fails (`skip` = index of the next node in pre-order outside this subtree). */
typedef struct rt_node {
    double bmin[3], bmax[3];
    uint32_t skip;   /* next node index when the box test fails (== n_nodes of the sub-array at the end) */
    uint32_t flags;  /* RT_NODE_NO_BOX: linear-list item (hittable.rs:41-53), no box test */
    uint32_t prim0;  /* leaf node: first primitive (bvh.rs left), RT_NONE for an inner node */
    uint32_t prim1;  /* leaf node: second primitive (bvh.rs right), RT_NONE if absent or == prim0 */
} rt_node;
#define RT_NODE_NO_BOX 1u
#define RT_NODE_DUP_RIGHT 2u /* 1-object leaf (bvh.rs:37-43): the reference tests prim0 a second time as `right`;
                                the kernel skips the re-test (same record) but counts it in the test counters */

#define RT_PRIM_TRIANGLE 0u
#define RT_PRIM_SPHERE 1u
#define RT_PRIM_MOVING_SPHERE 2u
#define RT_PRIM_AARECT 3u
#define RT_PRIM_INSTANCE 4u /* BVHNode object, optionally under a Transformed (main.rs:161-188) */

typedef struct rt_prim {
    uint32_t kind;
    uint32_t index;     /* TRIANGLE: rt_tri_attrs index; INSTANCE: first node of the sub-BVH; else 0 */
    uint32_t xform;     /* rt_xform index of the enclosing Transformed, or RT_NONE */
    uint32_t material;  /* rt_material index (RT_NONE for INSTANCE) */
    uint32_t object_id; /* id of the primitive object in the scene description */
    uint32_t inst_id;   /* id of the enclosing Transformed object, or RT_NONE */
    uint32_t sub_end;   /* INSTANCE: one past the last node of the sub-BVH */
    uint32_t pad;
} rt_prim;

/* Intersection record, 80 B, PARALLEL to the prims table (geom[p] belongs to prims[p]): everything the
 * primitive test reads, fetched with one wave-uniform scalar load.  d[] by kind:
 *   TRIANGLE       v0(3), e1 = v1 - v0 (3), e2 = v2 - v0 (3): the edges are the first two statements of
 *                  mesh.rs:55-56, evaluated once on the host in f64 (same IEEE subtraction, same bits)
 *   SPHERE         center(3), radius
 *   MOVING_SPHERE  center_0(3), center_1(3), time_0, time_1, radius
 *   AARECT         x0, x1, y0, y1, z
 *   INSTANCE       first node, one-past-last node of the sub-BVH (exact small integers) */
typedef struct rt_geom { double d[9]; uint32_t kind; uint32_t xform; } rt_geom;
typedef struct rt_tri_attrs { double n0[3], n1[3], n2[3]; double uv0[2], uv1[2], uv2[2]; } rt_tri_attrs; /* 120 B, shading only */
typedef struct rt_xform { double t_inv[16]; } rt_xform; /* column-major inverse of the Transformed matrix (transform.rs:14-20) */

#define RT_MAT_LAMBERTIAN 0u
#define RT_MAT_METAL 1u
#define RT_MAT_DIELECTRIC 2u
#define RT_MAT_DIFFUSE_LIGHT 3u
typedef struct rt_material {
    uint32_t kind;
    uint32_t texture;  /* albedo (LAMBERTIAN) / emit_colour (DIFFUSE_LIGHT) */
    double albedo[3];  /* METAL */
    double fuzz;       /* METAL, already clamped to <= 1 (material.rs:55-62) */
    double ior;        /* DIELECTRIC */
} rt_material;

#define RT_TEX_SOLID 0u
#define RT_TEX_CHECKER 1u
#define RT_TEX_IMAGE 2u
typedef struct rt_texture {
    uint32_t kind;
    uint32_t a;  /* CHECKER: odd texture  | IMAGE: byte offset of the first texel in the texel pool */
    uint32_t b;  /* CHECKER: even texture | IMAGE: width  */
    uint32_t c;  /*                         IMAGE: height */
    double colour[3]; /* SOLID */
} rt_texture;
/* texel pool: RGB8 bytes; the device computes (byte/255)^2 in f64 per fetch, bit-identical to the
 * reference's load_texture table (main.rs:900-903) at 1/8 of its footprint */

typedef struct rt_light { /* scene.lights entry: only Sphere implements SampleableLight (shape.rs:36-53) */
    double center[3];
    double radius;
    uint32_t material;
    uint32_t object_id;
} rt_light;

typedef struct rt_camera { /* renderer/src/camera.rs:9-19 */
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius, time0, time1;
} rt_camera;

typedef struct rt_blob_header {
    uint32_t magic, version;
    uint64_t total_bytes;
    uint32_t n_nodes, n_prims, n_tris, n_spheres, n_moving_spheres, n_rects, n_xforms, n_materials, n_textures, n_lights; /* n_tris..n_rects: counts by kind (n_tris = rt_tri_attrs entries) */
    uint32_t n_texel_bytes;
    uint32_t root_node; /* first node of the scene-level traversal (BVH root or first list item) */
    uint32_t root_end;  /* one past the last node of the scene-level traversal */
    uint32_t background;
    uint64_t off_nodes, off_prims, off_geom, off_tri_attrs, reserved0, reserved1, reserved2, off_xforms,
        off_materials, off_textures, off_texels, off_lights;
    rt_camera camera;
    uint64_t bvh_seed;
} rt_blob_header;

/* ---------------------------------------------------------------------------------------------
 * Entry points
 * ------------------------------------------------------------------------------------------- */
typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

typedef struct rt_render_params {
    uint32_t width, height;
    uint32_t spp_begin, spp_end; /* absolute sample indices [begin, end): rank r of R renders its own slice */
    int32_t max_depth;           /* main.rs:944 */
    int32_t integrator;          /* RT_INT_* */
    uint64_t seed;               /* Philox key; counter = (pixel, sample, level, block) */
    uint32_t samples_per_pass;   /* 0 = library default, a pure function of (width*height, max_depth, spp_end - spp_begin) and the scene's pass
                                  * budget (rt_scene_set_pass_budget, 80 GiB of path state per lane): budget, 128M paths, half the call's samples,
                                  * whichever is smallest -- pass boundaries fix the order of the fp32 sums.  If the buffers do not fit, the budget
                                  * is halved (at most five times, and it stays lowered for the scene) and the call renders in smaller passes;
                                  * rt_stats.samples_per_pass reports the size used.  Paths in flight = width*height*samples_per_pass per lane; a
                                  * call of two or more passes runs two passes at a time on two streams */
    uint32_t flags;              /* RT_RENDER_* */
} rt_render_params;
#define RT_RENDER_COUNT_TESTS 1u /* also count box/primitive tests (slower kernel variant) */
#define RT_RENDER_NO_SORT 2u     /* do not sort the ray queues by coherence key (octant | origin cell | direction bin | foot point of the line); debug / A-B */
#define RT_RENDER_LANE_WALK 8u   /* A-B: trace secondary and shadow rays with the per-lane walk (k_walk: own cursor and vector loads per lane, finished
                                  * lanes take the next ray) in scenes that qualify (a pruned space, no instances); same frame, measured slower than the
                                  * default wave-coherent walk */
#define RT_RENDER_COUNT_STEPS 32u   /* fill rt_stats.trace_wave_steps (record steps of the traversal waves: what bench.py's byte model needs).  One
                                        same-address atomic per wave: measured 5 % of the frame, so a plain render leaves the field 0.
                                        (RT_RENDER_COUNT_TESTS implies it.) */
#define RT_RENDER_MATERIAL_BINS 16u /* opt-in: before every shade launch regroup the queue's slots by the material class of their hits
                                        (stable, atomic-free partition: k_bin_count / k_bin_scan / k_bin_scatter), so that a wave shades
                                        one class.  Frames are bit-identical either way.  Measured r02: +1.5 % on `random` (three
                                        scattering kinds), -5 % on `earth`: k_shade is bound by its f64 path-state traffic, not by
                                        divergent material branches, so the default shades in queue order.  Ignored in scenes with
                                        one kind of scattering material. */
#define RT_RENDER_SERIAL_SHADOW 64u /* trace the MIS shadow rays on the pass's own stream instead of a side stream beside the next bounce's path rays
                                        (A-B and attribution: with everything on one stream a kernel's elapsed time is its own; same frame) */
#define RT_RENDER_ONE_LANE 4u    /* run the passes of a call one after the other instead of two at a time (A-B; same frame) */

typedef struct rt_stats {
    uint64_t rays;       /* Scene::hit-equivalent closest-hit queries: camera + bounce + MIS shadow rays */
    uint64_t paths;      /* camera samples = width*height*(spp_end-spp_begin) */
    uint64_t box_tests;  /* only with RT_RENDER_COUNT_TESTS / rt_trace_closest counters */
    uint64_t tri_tests;
    uint64_t sphere_tests;
    uint64_t launches;   /* kernel launches issued (a radix sort of a queue counts as one) */
    double ms_total;     /* first ray-gen launch -> accumulators final (HIP events on the context stream) */
    double ms_trace;     /* sum over trace_closest launches (HIP events on the launching stream; launches of the two lanes
                          * overlap, so the sums of both lanes can exceed ms_total) */
    double ms_shade;     /* sum over shade launches */
    double ms_other;     /* raygen + unwind/accumulate + queue upkeep */
    uint64_t trace_launches;
    uint64_t trace_wave_steps;  /* stream records stepped through, summed over the waves of every traversal launch (one record = one box or leaf test for up to 64 rays) */
    uint64_t filter_violations; /* only with RT_RENDER_COUNT_TESTS: fp32 filter decisions contradicted by the exact f64 test; 0 by construction */
    double ms_trace_shadow;          /* the part of ms_trace spent in launches over MIS shadow rays (integrator.rs:82) */
    uint64_t trace_shadow_launches;  /* the part of trace_launches that traced shadow rays */
    uint64_t shadow_rays;            /* the part of `rays` that were shadow rays */
    uint32_t samples_per_pass;       /* pass size the call used (samples_per_pass == 0: a pure function of the parameters and the scene's pass budget) */
    uint32_t lanes;                  /* render lanes the call used (1 or 2) */
    uint64_t walk_run_steps;         /* per-lane walk (k_walk): records visited, summed over rays (one 64-byte fetch each) */
    uint64_t walk_tri_steps;         /* ... triangle-leaf visits (144 bytes each) */
    uint64_t walk_slow_steps;        /* ... visits that needed f64 arithmetic (exact box / triangle / sphere tests) */
    uint64_t walk_iterations;        /* ... block executions summed over waves (each runs ONE of fetch / record / triangle-leaf / f64 for its waiting lanes) */
    double ms_tail;                  /* bounces over SMALL queues (fewer rays than a launch needs to outlast a host round trip; the 10-spp frame of configs[0] is
                                      * 45 of them): traversal, shading and visibility launches timed as ONE span per pass -- per-launch events would cost
                                      * more than the launches -- and counted in neither ms_trace / ms_shade nor trace_launches */
    uint64_t tail_bounces;           /* bounces timed under ms_tail */
} rt_stats;

/* number of visible HIP devices (>= 0) or negative error */
int rt_device_count(void);
/* Creates a context on `device` with its own stream.  `stream` may be a caller hipStream_t (cast
 * to void*, e.g. torch.cuda.current_stream().cuda_stream) or NULL for a library-owned stream.
 * Also takes the device address of this GPU's fixed scratch array (rt_trace_closest_device_noalloc). */
int rt_context_create(int device, void* stream, rt_context** out);
int rt_context_destroy(rt_context* ctx); /* RT_ERR_BAD_ARG while scenes uploaded through it are alive: destroy those first */
int rt_context_synchronize(rt_context* ctx);

/* Validates and copies the flattened blob to the device (caller keeps `blob`). */
int rt_scene_upload(rt_context* ctx, const void* blob, uint64_t nbytes, rt_scene** out);
int rt_scene_destroy(rt_scene* scene);
/* Host only (no HIP call; works without a GPU): runs the validation of rt_scene_upload and writes the trace stream
 * rt_scene_upload derives from the blob (pre-order records, 32 B header = fp32 box |
 * skip offset | flags/size/node word, leaf blocks behind it) to stream_out (may be NULL to query the size).
 * *stream_bytes = size including the zero padding the kernel's window over-reads; *root_off = offset of the first
 * record of the scene-level array; node_off (may be NULL) = n_nodes + 1 record offsets, by node index.
 * No counterpart in the reference (the stream replaces the pointer tree of bvh.rs:15-23 on the device): it
 * exists so that tests without a GPU can check what the kernel will walk. */
int rt_stream_build_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes,
                         uint32_t* root_off, uint32_t* node_off);
/* Same for the PRODUCTION stream the kernel walks: the reference's leaves in the reference's order under a hierarchy
 * of this library's own (re-bracketed by surface area when the reference's boxes nest; inner nodes most waves enter anyway have no record).
 * node_off = n_nodes + 1 offsets by the BLOB's node index: meaningful for leaves and instance sub-arrays (what the certificate and the
 * instance walk look up); the node word of a record indexes the array rt_stream_nodes_prod_host returns. */
int rt_stream_build_prod_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes,
                              uint32_t* root_off, uint32_t* node_off);
/* The node array the production stream's records index: the blob's nodes, followed -- when the scene-level hierarchy was rebuilt -- by the
 * new scene-level range [*range_begin, *range_end) in pre-order (copies of the leaves in the reference's order under new inner nodes whose
 * boxes are the unions of their leaves').  nodes_out may be NULL to query *n_nodes_out; elided_out (may be NULL) = one byte per node, 1 = no
 * record in the production stream.  Host only; no counterpart in the reference. */
int rt_stream_nodes_prod_host(const void* blob, uint64_t nbytes, rt_node* nodes_out, uint32_t capacity_nodes, uint32_t* n_nodes_out,
                              uint32_t* range_begin, uint32_t* range_end, uint8_t* elided_out);
/* Host only, for tests without a GPU: what rt_scene_upload derives for strict pruning.
 * info[12] = enabled, pruned transform id, triangles covered, inner records with a local box, distinct normal
 * clusters, direction-grid entries, minimum shape factor, tau_coef, delta_coef, radius, local box |max|, inner nodes
 * without a record in the production stream.
 * rays (may be NULL with n_rays = 0): n_rays x 6 doubles (origin, direction) ALREADY in the pruned space;
 * certified[i] = 1 iff the device would walk that ray with pruning.  No counterpart in the reference. */
int rt_prune_info_host(const void* blob, uint64_t nbytes, double* info, uint32_t n_rays, const double* rays, uint8_t* certified);

/* Scene::hit for n rays (scene.rs:31-38).  rays: n x 7 doubles (ox,oy,oz,dx,dy,dz,time), host.
 * Outputs (host): prim_id = object id of the closest primitive or RT_NONE; inst_id = enclosing
 * Transformed object id or RT_NONE (may be NULL); t = ray parameter (0 on miss).
 * counters (host, may be NULL): 8 x u64 = box, triangle, sphere tests summed over the batch (reference order: the
 * counting kernel walks the reference's full visited set), then the number of decisions of the production kernel's
 * shortcuts the exact walk contradicted (fp32 filter verdicts, accepts inside a subtree strict pruning would skip:
 * 0 by construction), then the number of rays that would be walked without a pruning certificate; 3 reserved. */
int rt_trace_closest(rt_scene* scene, uint32_t n, const double* rays, double t_min, double t_max,
                     uint32_t* prim_id, uint32_t* inst_id, double* t, uint64_t* counters);
/* Same, everything already resident: rays_dev n x 7 doubles, outputs device arrays; asynchronous on
 * the context stream.  Device conventions differ from the host variant: prim_id_dev holds the INDEX into the blob's prims
 * table (RT_NONE on a miss), and t_dev is NaN (not 0) on a miss. */
int rt_trace_closest_device(rt_scene* scene, uint32_t n, const double* rays_dev, double t_min, double t_max,
                            uint32_t* prim_id_dev, double* t_dev);
/* rt_trace_closest_device for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  NO counterpart in the reference.  Same arguments, same outputs (prim_id_dev: index into the
 * blob's prims table or RT_NONE; t_dev: NaN on a miss), asynchronous on the context stream, but it calls no hipMalloc /
 * hipFree: the 64-byte ray blocks the kernel reads and its counters live in a fixed array inside the library
 * (16 MiB + 256 B per GPU, held until the process ends).  n > 262144 -> RT_ERR_BAD_ARG.  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (two contexts / streams) race; calls on one context's stream are ordered. */
#define RT_NOALLOC_MAX_RAYS_ABI 262144u
int rt_trace_closest_device_noalloc(rt_scene* scene, uint32_t n, const double* rays_dev, double t_min, double t_max,
                                    uint32_t* prim_id_dev, double* t_dev);

/* Camera rays (main.rs:996-999 + camera.rs:83-101) of one sample index for every pixel, written
 * to host `rays` (width*height x 7 doubles, row 0 = bottom row). */
int rt_primary_rays(rt_scene* scene, uint32_t width, uint32_t height, uint32_t sample, uint64_t seed, double* rays);

/* Renders samples [spp_begin, spp_end) of every pixel and ADDS the per-pixel radiance SUM (not the
 * mean) into accum_dev: device, width*height*3 fp32, row 0 = bottom row.  The caller zeroes it
 * (rt_accum_zero) before the first call; calls may be chained over disjoint sample ranges; ranks
 * all-reduce(sum) their accumulators before rt_resolve.  Synchronous: returns when accum is final. */
int rt_render(rt_scene* scene, const rt_render_params* params, float* accum_dev, rt_stats* stats);
int rt_accum_zero(rt_context* ctx, float* accum_dev, uint32_t width, uint32_t height);
/* Bytes of path state per render lane that a call with samples_per_pass == 0 may use (0 restores the default, 80 GiB: two lanes,
 * the scene and its certificate tables fit 288 GB).  A renderer that shares its GPU sets it once, before the first render, so that
 * every call -- also resumed and sharded ones -- cuts the same passes.  No counterpart in the reference (it renders pixel by pixel). */
int rt_scene_set_pass_budget(rt_scene* scene, uint64_t bytes_per_lane);
/* rt_render split in two, so that consecutive calls overlap on the GPU (the thin deep bounces of one call leave the machine part-idle;
 * the next call's kernels fill it -- what the two lanes do INSIDE one rt_render, without halving the pass size).  rt_render_begin
 * queues the call on one of the scene's two lanes (own stream, own path buffers, a host thread for the per-bounce readbacks) and
 * returns; rt_render_end waits for it, fills stats and frees the job.  At most two jobs per scene are in flight: begin, begin,
 * end(first), begin, ...  Jobs add their sums to the accumulator in the order they were begun (pass size = the default of rt_render
 * without the "half the call" rule, or samples_per_pass), so a sequence of jobs gives one well-defined frame.  While jobs are in flight
 * the scene accepts no other call.  RT_RENDER_COUNT_TESTS is not available here.  No counterpart in the reference. */
typedef struct rt_job rt_job;
int rt_render_begin(rt_scene* scene, const rt_render_params* params, float* accum_dev, rt_job** job);
int rt_render_end(rt_job* job, rt_stats* stats);
/* Host-memory convenience for callers without device allocations (the `desktop` app): allocates a
 * device accumulator internally, renders, copies the sums to host accum (width*height*3 fp32). */
int rt_render_host(rt_scene* scene, const rt_render_params* params, float* accum_host, rt_stats* stats);

/* Test hook, no counterpart in the reference: runs the render loop for ONE sample index of every pixel (spp_end == spp_begin + 1;
 * the kernels, queues and sorts of rt_render) up to the traversal launch of `bounce` and copies out every stride-th ray of that
 * launch's queue (shadow = 0: the path rays; 1: the MIS shadow rays, integrator.rs:82), at most max_n.  Host outputs: rays n x 7;
 * for path rays prim_id / inst_id / t = the closest hit exactly as the render loop's own launch recorded it (object ids as in
 * rt_trace_closest, t = NaN on a miss); for shadow rays t = what the visibility walk reported (it stops at the first accepted hit
 * that already decides "not visible", so only `|t - t_light| <= 1e-4` is meaningful) and t_light = light.hit(ray).t.
 * *queue_len (may be NULL) = rays in that queue.  Parity tests hand these rays of the real frame to the oracle. */
int rt_capture_rays(rt_scene* scene, const rt_render_params* params, uint32_t bounce, int shadow, uint32_t stride, uint32_t max_n,
                    uint32_t* n_out, uint32_t* queue_len, double* rays, uint32_t* prim_id, uint32_t* inst_id, double* t, double* t_light);

/* accum sums -> displayable pixels: /spp, optional ACES (main.rs:913-921), NaN channel -> 0
 * (image.rs:48-56).  Device in, device out (width*height*3 fp32; may alias). */
int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t width, uint32_t height, uint32_t spp, int tonemap,
               float* out_dev);
int rt_resolve_host(const float* accum_host, uint32_t width, uint32_t height, uint32_t spp, int tonemap, float* out_host);

/* Radiance of individual camera samples, f64, no accumulation: out_rgb (host, n x 3 doubles) for
 * samples (pixel_index[i], sample_index[i]).  Used by the parity tests to compare single paths. */
int rt_sample_radiance(rt_scene* scene, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index,
                       const uint32_t* sample_index, int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb);

/* Same, plus the SIGNATURE of every path (out_sig: n x u64, host): an order-free hash over every discrete decision the path took
 * (object hit and side per level, checker branch / texel, kind of scatter, reflect or refract, light picked and whether its sample
 * counted, visible or not, the .001 weight gate; csrc/rt_math.h).  Paths whose signature equals the oracle's followed the same
 * branches and must agree to rounding; a differing signature explains a deviation.  Test hook, no counterpart in the reference. */
int rt_sample_radiance_sig(rt_scene* scene, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index,
                           const uint32_t* sample_index, int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb, uint64_t* out_sig);

/* PDF::value / PDF::generate (renderer/src/pdf.rs:9-15) for n independent items on the device, for parity tests of the
 * PDF set by itself.  Item i is a list of n_comp PDFs pdfs[i*n_comp ..]: n_comp == 1 evaluates that PDF, n_comp in 2..8
 * evaluates MixturePDF with MixtureMethod::Uniform over the list (pdf.rs:185-224: value = sum of value_k / n, generate =
 * one uniformly chosen member's sample, normalised).  The random stream of item i is Philox (seed; i, 0, 0).
 * Fields by kind (the reference's constructors, pdf.rs):  COSINE / UNIFORM_HEMI: a = normal;  UNIFORM_SPHERE: a = centre,
 * s = radius;  FUZZY: a = direction as stored (already normalised, pdf.rs:87-91), s = fuzz;  DIELECTRIC: a = reflect_dir,
 * b = refract_dir, s = reflectance, cannot_refract;  CONE: a = axis, s = cos_theta_max.  All host pointers. */
#define RT_PDF_COSINE 0u
#define RT_PDF_UNIFORM_HEMI 1u
#define RT_PDF_UNIFORM_SPHERE 2u
#define RT_PDF_FUZZY 3u
#define RT_PDF_DIELECTRIC 4u
#define RT_PDF_CONE 5u
typedef struct rt_pdf_desc { uint32_t kind; uint32_t cannot_refract; double a[3], b[3], s; } rt_pdf_desc; /* 64 B */
int rt_pdf_eval(rt_context* ctx, uint32_t n, uint32_t n_comp, const rt_pdf_desc* pdfs, const double* dirs, uint64_t seed,
                double* value, double* generated);

/* IEEE check helper: q[i] = a[i]/b[i], s[i] = sqrt(a[i]) computed on the device with the same
 * compiler flags as the kernels (host arrays). */
int rt_ieee_div_sqrt(rt_context* ctx, uint32_t n, const double* a, const double* b, double* q, double* s);

const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
