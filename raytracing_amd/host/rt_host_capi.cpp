// C entry points of librt_host.so (include/rt_host.h).
#include "../../include/rt_host.h"
#include "desktop.hpp"

using namespace renderer;

struct rth_scene {
    Scene scene;
    std::string text; bool have_text = false;
    std::vector<uint8_t> blob; bool have_blob = false;
};

static thread_local std::string g_err;
#define RTH_TRY try {
#define RTH_CATCH(ret) } catch (const std::exception& e) { g_err = e.what(); return ret; }

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }
void rth_set_bvh_axis_seed(uint64_t seed) { bvh_axis_seed() = seed; }
int rth_set_bvh_axis_mode(uint32_t mode) { if (mode > BVH_AXIS_MIN_AREA) return -1; bvh_axis_mode() = mode; return 0; }

rth_scene* rth_scene_builtin(const char* name, const char* assets_dir, int legacy_bvh) {
    RTH_TRY
    auto s = std::make_unique<rth_scene>();
    desktop::Assets a{assets_dir ? assets_dir : "assets"};
    s->scene = desktop::scene_by_name(name, a, legacy_bvh != 0);
    return s.release();
    RTH_CATCH(nullptr)
}
rth_scene* rth_scene_from_description(const char* text, uint64_t len) {
    RTH_TRY
    auto s = std::make_unique<rth_scene>();
    s->scene = scene_from_description(text, (size_t)len);
    return s.release();
    RTH_CATCH(nullptr)
}
void rth_scene_free(rth_scene* s) { delete s; }

const char* rth_scene_describe(rth_scene* s, uint64_t* len) {
    RTH_TRY
    if (!s->have_text) { s->text = s->scene.describe(); s->have_text = true; }
    if (len) *len = s->text.size();
    return s->text.c_str();
    RTH_CATCH(nullptr)
}
const void* rth_scene_flatten(rth_scene* s, uint64_t* nbytes) {
    RTH_TRY
    if (!s->have_blob) { s->blob = s->scene.flatten(); s->have_blob = true; }
    if (nbytes) *nbytes = s->blob.size();
    return s->blob.data();
    RTH_CATCH(nullptr)
}
uint32_t rth_scene_object_count(rth_scene* s) { return (uint32_t)s->scene.objects.size(); }

int rth_write_ppm(const char* path, const double* rgb, uint32_t width, uint32_t height) {
    RTH_TRY
    Image img = Image::new_({width, height});
    for (size_t i = 0; i < (size_t)width * height; ++i) img.data[i] = DVec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    desktop::write_image(img, path);
    return 0;
    RTH_CATCH(-1)
}
static Image image_of(const double* rgb, uint32_t width, uint32_t height) {
    Image img = Image::new_({width, height});
    for (size_t i = 0; i < (size_t)width * height; ++i) img.data[i] = DVec3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    return img;
}
int rth_write_pfm(const char* path, const double* rgb, uint32_t width, uint32_t height) {
    RTH_TRY
    desktop::write_pfm(image_of(rgb, width, height), path);
    return 0;
    RTH_CATCH(-1)
}
int rth_write_png(const char* path, const double* rgb, uint32_t width, uint32_t height) {
    RTH_TRY
    desktop::write_png(image_of(rgb, width, height), path);
    return 0;
    RTH_CATCH(-1)
}
void rth_image_get_tile(uint32_t img_w, uint32_t img_h, uint32_t ox, uint32_t oy, uint32_t tw, uint32_t th, uint32_t* out4) {
    Image img; img.size = {img_w, img_h};
    ImageTile t = img.get_tile({ox, oy}, {tw, th});
    out4[0] = t.origin.first; out4[1] = t.origin.second; out4[2] = t.size.first; out4[3] = t.size.second;
}
void rth_tile_get_xy(uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, uint32_t index, uint32_t* xy) {
    ImageTile t; t.origin = {ox, oy}; t.size = {w, h};
    auto p = t.get_xy(index); xy[0] = p.first; xy[1] = p.second;
}
int rth_image_merge_tile(double* img_rgb, uint32_t img_w, uint32_t img_h, uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, const double* tile_rgb) {
    RTH_TRY
    Image img = Image::new_({img_w, img_h});
    for (size_t i = 0; i < img.data.size(); ++i) img.data[i] = DVec3(img_rgb[3 * i], img_rgb[3 * i + 1], img_rgb[3 * i + 2]);
    ImageTile t; t.origin = {ox, oy}; t.size = {w, h}; t.pixels.resize((size_t)w * h);
    for (size_t i = 0; i < t.pixels.size(); ++i) t.pixels[i] = DVec3(tile_rgb[3 * i], tile_rgb[3 * i + 1], tile_rgb[3 * i + 2]);
    img.merge_tile(t);
    for (size_t i = 0; i < img.data.size(); ++i) { img_rgb[3 * i] = img.data[i].x; img_rgb[3 * i + 1] = img.data[i].y; img_rgb[3 * i + 2] = img.data[i].z; }
    return 0;
    RTH_CATCH(-1)
}
void rth_mat4_inverse(const double* m, double* out) {
    DMat4 x = DMat4::from_cols(DVec4(m[0], m[1], m[2], m[3]), DVec4(m[4], m[5], m[6], m[7]), DVec4(m[8], m[9], m[10], m[11]), DVec4(m[12], m[13], m[14], m[15]));
    DMat4 r = x.inverse();
    const DVec4* c[4] = {&r.x_axis, &r.y_axis, &r.z_axis, &r.w_axis};
    for (int i = 0; i < 4; ++i) { out[4 * i] = c[i]->x; out[4 * i + 1] = c[i]->y; out[4 * i + 2] = c[i]->z; out[4 * i + 3] = c[i]->w; }
}
void rth_sphere_transform(const double* s, const double* m, double* out) {
    DMat4 x = DMat4::from_cols(DVec4(m[0], m[1], m[2], m[3]), DVec4(m[4], m[5], m[6], m[7]), DVec4(m[8], m[9], m[10], m[11]), DVec4(m[12], m[13], m[14], m[15]));
    Sphere r = Sphere(DVec3(s[0], s[1], s[2]), s[3], nullptr).transform(x);
    out[0] = r.center.x; out[1] = r.center.y; out[2] = r.center.z; out[3] = r.radius;
}
void rth_pcg64_stream(uint64_t seed, uint32_t n, uint64_t* out) {
    desktop::Pcg64 p = desktop::Pcg64::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = p.next_u64();
}

}  // extern "C"
