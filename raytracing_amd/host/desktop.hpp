// Host-side mirror of the reference `desktop` crate's scene library and I/O helpers
// (desktop/src/main.rs:35-921, importers/obj.rs, exporters/ppm.rs).  Scene functions keep their
// reference names and build the same object graphs through the mirrored `renderer` API.
#pragma once
#include "renderer.hpp"

namespace desktop {
using namespace renderer;

// Legacy scene style of main.rs:35 -- (objects, camera, background) -- adapted into a Scene by
// `adapt`: empty light list, optional scene BVH.
struct SceneDescription { HittableList world; Camera camera; Background background; };
Scene adapt(SceneDescription d, bool build_bvh);

struct Assets {
    std::string root;  // directory holding pagoda/model_triangulated.obj, textures/earthmap.ppm, cube.obj
};

std::optional<Image> load_texture(const std::string& filename);                                             // main.rs:891-910 (PPM P6 input)
std::optional<std::vector<std::vector<Triangle>>> load_obj(const std::string& file_name, Arc<Material> material);  // importers/obj.rs:6-58
void write_image(const Image& img, const std::string& location);
void write_pfm(const Image& img, const std::string& location);   // linear float32 dump for diffing (no counterpart in the reference)
void write_png(const Image& img, const std::string& location);   // the bytes of write_image as an 8-bit RGB PNG (no counterpart)
//                                            // exporters/ppm.rs:6-38
DVec3 aces_tonemapping(DVec3 pixel);                                                                        // main.rs:913-921 (host copy for tools; the render path tonemaps on the GPU)

SceneDescription create_cube_scene();                              // main.rs:51-91
SceneDescription create_cornell_scene(const Assets& a);            // main.rs:92-211
SceneDescription create_sphere_scene(const Assets& a);             // main.rs:212-278
SceneDescription create_simple_scene(const Assets& a);             // main.rs:281-362
Scene create_random_scene(bool motion_blur);                       // main.rs:365-494
SceneDescription two_spheres();                                    // main.rs:497-545
SceneDescription simple_triangle_scene();                          // main.rs:547-633
SceneDescription simple_light(const Assets& a);                    // main.rs:636-685
Scene single_sphere_light_scene();                                 // main.rs:687-787
Scene mesh_scene(const Assets& a, bool textured);                  // main.rs:789-889

// name -> Scene ("random", "random_blur", "mesh", "simple", "simple_triangle", "cornell", "sphere",
// "cube", "two_spheres", "simple_light", "single_sphere_light"); legacy scenes adapted with a BVH
// when `legacy_bvh` is set.
Scene scene_by_name(const std::string& name, const Assets& a, bool legacy_bvh);

// rand_pcg::Pcg64 (Lcg128Xsl64) + rand 0.8 float conversions, restated from the crates' published
// algorithms (sources not in /root/reference: parity unpinned) -- fixes the sphere layout of
// create_random_scene (main.rs:376).
struct Pcg64 {
    unsigned __int128 state, increment;
    static Pcg64 seed_from_u64(uint64_t seed);
    uint64_t next_u64();
    double gen_f64();                              // Standard: (u64 >> 11) * 2^-53
    double gen_range(double low, double high);     // UniformFloat::sample_single
};

}  // namespace desktop
