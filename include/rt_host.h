/* rt_host.h -- C entry points of librt_host.so: host-side scene construction and flattening
 * (no GPU code, no intersection code).  This is the C face of the C++ mirror of the reference's
 * `renderer` scene API (renderer/src/scene.rs:57-114, bvh.rs:27-87) and `desktop` scene library
 * (desktop/src/main.rs:35-889), for callers that cannot include raytracing_amd/host/renderer.hpp
 * (Python ctypes, a Rust `extern "C"` block).  0 / non-NULL = ok; message via rth_last_error().
 */
#ifndef RT_HOST_H
#define RT_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* Seed of the BVH split-axis stream (replaces thread_rng at bvh.rs:53) for scenes built afterwards. */
void rth_set_bvh_axis_seed(uint64_t seed);
/* Where a BVH build takes its split axes from, for scenes built afterwards.  bvh.rs:53 draws them from an UNSEEDED thread_rng, so every
 * assignment of axes is a tree the reference can build.  0 (default): the seeded stream above.  1: at every node the axis, of the three,
 * whose median split (bvh.rs:55-77 unchanged) has the smallest area(left)*|left| + area(right)*|right| (host/renderer.hpp).  Returns 0,
 * -1 for an unknown mode.  The scene description records the mode (`bvh_axis_mode 1`), so the oracle builds the same tree. */
#define RTH_BVH_AXIS_SEEDED 0u
#define RTH_BVH_AXIS_MIN_AREA 1u
int rth_set_bvh_axis_mode(uint32_t mode);

/* One of the reference scene functions by name: "random" (create_random_scene(false), main.rs:365),
 * "random_blur", "mesh" (mesh_scene, main.rs:789), "simple" (create_simple_scene, main.rs:281),
 * "simple_triangle" (main.rs:547), "cornell" (main.rs:92), "sphere", "cube", "two_spheres",
 * "simple_light", "single_sphere_light".  assets_dir holds pagoda/, textures/, cube.obj.
 * legacy_bvh: build a scene BVH for the legacy (SceneDescription) scenes. */
rth_scene* rth_scene_builtin(const char* name, const char* assets_dir, int legacy_bvh);
/* Scene from an "rtscene 1" text description (grammar in INTEGRATION.md). */
rth_scene* rth_scene_from_description(const char* text, uint64_t len);
void rth_scene_free(rth_scene* s);

/* Borrowed views, valid until rth_scene_free: canonical description text and the flattened
 * "RTSB" blob that rt_scene_upload (rt_abi.h) takes. */
const char* rth_scene_describe(rth_scene* s, uint64_t* len);
const void* rth_scene_flatten(rth_scene* s, uint64_t* nbytes);
uint32_t rth_scene_object_count(rth_scene* s);

/* exporters/ppm.rs:6-38: rgb = width*height*3 doubles (row 0 = bottom), already resolved. */
int rth_write_ppm(const char* path, const double* rgb, uint32_t width, uint32_t height);
/* No counterpart in the reference (output for diffing): the same frame as little-endian PFM (linear
 * float32, rows bottom-up) and as 8-bit RGB PNG holding exactly the values rth_write_ppm prints. */
int rth_write_pfm(const char* path, const double* rgb, uint32_t width, uint32_t height);
int rth_write_png(const char* path, const double* rgb, uint32_t width, uint32_t height);
/* image.rs:61-70 / 19-26 (tile helpers of the framebuffer API) */
void rth_image_get_tile(uint32_t img_w, uint32_t img_h, uint32_t ox, uint32_t oy, uint32_t tw, uint32_t th, uint32_t* out4);
void rth_tile_get_xy(uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, uint32_t index, uint32_t* xy);
/* image.rs:72-78 + 43-59: merge a tile of pixels into an image buffer (NaN channels -> 0) */
int rth_image_merge_tile(double* img_rgb, uint32_t img_w, uint32_t img_h, uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, const double* tile_rgb);
/* glam DMat4::inverse restatement used by Transformed::new (transform.rs:17), column-major */
void rth_mat4_inverse(const double* m16, double* out16);
/* Sphere::transform (shape.rs:23-34): in center(3)+radius, out center(3)+radius */
void rth_sphere_transform(const double* sphere4, const double* m16, double* out4);
/* Pcg64::seed_from_u64(seed) stream (main.rs:376): n next_u64 outputs */
void rth_pcg64_stream(uint64_t seed, uint32_t n, uint64_t* out);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
