// Host-side mirror of the reference `renderer` crate's public surface (renderer/src/lib.rs:1-55):
// same type names, constructor arguments and builder flow, so the scene functions of
// desktop/src/main.rs carry over statement by statement.  What differs, by design:
//   * there is NO CPU intersection / shading code here: `hit`, `ray_colour`, PDFs and materials
//     execute only on the GPU (librt_hip.so).  The host builds the scene graph, builds BVHs
//     (bvh.rs:36-87 is host-side work), and flattens everything into one "RTSB" blob (rt_abi.h).
//   * trait objects become small class hierarchies with a `describe`/`flatten` hook
//     (the additive API SURVEY.md section 7 calls for); `fn(Ray)->DVec3` backgrounds become an enum.
//   * the BVH split axis (thread_rng in bvh.rs:53) comes from a seeded stream so trees are reproducible.
#pragma once
#include "glam.hpp"
#include "../../include/rt_abi.h"
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace renderer {
using glam::DMat4;
using glam::DVec2;
using glam::DVec3;
using glam::DVec4;

template <class T> using Arc = std::shared_ptr<T>;

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

struct Ray {  // ray.rs:3-12
    DVec3 origin, dir; double time = 0;
    DVec3 at(double t) const { return origin + dir * t; }
};

// ---- bounding_box.rs (host needs boxes to build the BVH; the slab test runs on the GPU) -------
struct AABB {
    DVec3 min, max;
    bool operator==(const AABB& o) const { return min.x == o.min.x && min.y == o.min.y && min.z == o.min.z && max.x == o.max.x && max.y == o.max.y && max.z == o.max.z; }
    static AABB surrounding_box(const AABB& box_1, const AABB& box_2) {  // bounding_box.rs:11-25
        return AABB{DVec3(std::fmin(box_1.min.x, box_2.min.x), std::fmin(box_1.min.y, box_2.min.y), std::fmin(box_1.min.z, box_2.min.z)),
                    DVec3(std::fmax(box_1.max.x, box_2.max.x), std::fmax(box_1.max.y, box_2.max.y), std::fmax(box_1.max.z, box_2.max.z))};
    }
};

// ---- image.rs ---------------------------------------------------------------------------------
struct ImageTile {  // image.rs:13-26
    std::pair<uint32_t, uint32_t> origin, size;
    std::vector<DVec3> pixels;
    std::pair<uint32_t, uint32_t> get_xy(uint32_t index) const { return {(index % size.first) + origin.first, (index / size.first) + origin.second}; }
};

struct Texture {
    virtual ~Texture() {}
};
struct SolidColour : Texture {  // texture.rs:10-18
    DVec3 colour;
    explicit SolidColour(DVec3 c) : colour(c) {}
};
struct CheckerTexture : Texture {  // texture.rs:21-33
    Arc<Texture> odd, even;
    CheckerTexture(Arc<Texture> o, Arc<Texture> e) : odd(std::move(o)), even(std::move(e)) {}
    static CheckerTexture new_(DVec3 c1, DVec3 c2) { return CheckerTexture(std::make_shared<SolidColour>(c1), std::make_shared<SolidColour>(c2)); }
};
struct Image : Texture {  // image.rs:7-79 (framebuffer and image texture storage)
    std::pair<uint32_t, uint32_t> size;
    std::vector<DVec3> data;
    std::vector<uint8_t> rgb8;  // source bytes when made by load_texture (main.rs:891-910); what the GPU texel pool stores
    static Image new_(std::pair<uint32_t, uint32_t> size) { Image i; i.size = size; i.data.assign((size_t)size.first * size.second, DVec3()); return i; }
    const DVec3* get(uint32_t x, uint32_t y) const {
        size_t index = (size_t)(y * size.first) + x;
        return index < data.size() ? &data[index] : nullptr;
    }
    void put(uint32_t x, uint32_t y, const DVec3& colour_in) {  // image.rs:43-59 (NaN channel -> 0)
        size_t index = (size_t)(y * size.first) + x;
        DVec3 colour = colour_in;
        if (std::isnan(colour.x)) colour.x = 0.0;
        if (std::isnan(colour.y)) colour.y = 0.0;
        if (std::isnan(colour.z)) colour.z = 0.0;
        data.at(index) = colour;
    }
    ImageTile get_tile(std::pair<uint32_t, uint32_t> origin, std::pair<uint32_t, uint32_t> tsize) const {  // image.rs:61-70
        uint32_t aw = std::min(origin.first + tsize.first, size.first) - origin.first;
        uint32_t ah = std::min(origin.second + tsize.second, size.second) - origin.second;
        ImageTile t; t.origin = origin; t.size = {aw, ah}; t.pixels.assign((size_t)aw * ah, DVec3());
        return t;
    }
    void merge_tile(const ImageTile& tile) {  // image.rs:72-78
        uint32_t n = tile.size.first * tile.size.second;
        for (uint32_t i = 0; i < n; ++i) { auto xy = tile.get_xy(i); put(xy.first, xy.second, tile.pixels[i]); }
    }
};

// ---- material.rs ------------------------------------------------------------------------------
struct Material { virtual ~Material() {} };
struct Lambertian : Material {  // material.rs:19-29
    Arc<Texture> albedo;
    explicit Lambertian(Arc<Texture> a) : albedo(std::move(a)) {}
    static Lambertian new_(DVec3 colour) { return Lambertian(std::make_shared<SolidColour>(colour)); }
};
struct Metal : Material {  // material.rs:50-62
    static Metal new_(DVec3 albedo, double fuzz) { Metal m; m.albedo_ = albedo; m.fuzz_ = fuzz > 1.0 ? 1.0 : fuzz; return m; }
    DVec3 albedo() const { return albedo_; }
    double fuzz() const { return fuzz_; }
  private:
    DVec3 albedo_; double fuzz_ = 0;
};
struct Dielectric : Material { double ior; explicit Dielectric(double i) : ior(i) {} };               // material.rs:82-84
struct DiffuseLight : Material { Arc<Texture> emit_colour; explicit DiffuseLight(Arc<Texture> e) : emit_colour(std::move(e)) {} };  // material.rs:120-122

// ---- hittable.rs ------------------------------------------------------------------------------
struct Hittable {
    virtual ~Hittable() {}
    virtual std::optional<AABB> bounding_box(double time_0, double time_1) const = 0;
    // `hit` exists only on the device: see rt_trace_closest / rt_render in rt_abi.h.
};
using HittableList = std::vector<Arc<Hittable>>;

inline std::optional<AABB> bounding_box(const Arc<Hittable>* list, size_t n, double time_0, double time_1) {  // hittable.rs:93-107
    if (n == 0) return std::nullopt;
    auto acc = list[0]->bounding_box(time_0, time_1);
    if (!acc) return std::nullopt;
    for (size_t i = 1; i < n; ++i) {
        auto b = list[i]->bounding_box(time_0, time_1);
        if (!b) return std::nullopt;
        acc = AABB::surrounding_box(*acc, *b);
    }
    return acc;
}

// ---- shape.rs ---------------------------------------------------------------------------------
struct Sphere : Hittable {  // shape.rs:17-21 (also the only SampleableLight, shape.rs:36-53)
    DVec3 center; double radius; Arc<Material> material;
    Sphere(DVec3 c, double r, Arc<Material> m) : center(c), radius(r), material(std::move(m)) {}
    Sphere transform(const DMat4& transform) const {  // Transformable, shape.rs:23-34
        DVec4 o = DVec4::from(center, 1.0);
        DVec4 r(o.x, o.y, o.z + radius, 0.0);
        return Sphere(transform.mul_vec4(o).xyz(), transform.mul_vec4(r).length(), material);
    }
    std::optional<AABB> bounding_box(double, double) const override {  // shape.rs:134-139
        return AABB{center - DVec3(radius, radius, radius), center + DVec3(radius, radius, radius)};
    }
};
using SampleableLight = Sphere;

struct MovingSphere : Hittable {  // shape.rs:202-247
    DVec3 center_0, center_1; double time_0, time_1, radius; Arc<Material> material;
    MovingSphere(DVec3 c0, DVec3 c1, double t0, double t1, double r, Arc<Material> m) : center_0(c0), center_1(c1), time_0(t0), time_1(t1), radius(r), material(std::move(m)) {}
    DVec3 center(double time) const { return center_0 + (center_1 - center_0) * ((time - time_0) / (time_1 - time_0)); }
    std::optional<AABB> bounding_box(double t0, double t1) const override {
        DVec3 c0 = center(t0), c1 = center(t1), rv(radius, radius, radius);
        return AABB::surrounding_box(AABB{c0 - rv, c0 + rv}, AABB{c1 - rv, c1 + rv});
    }
};

struct AARect : Hittable {  // shape.rs:258-305
    std::pair<double, double> x_range, y_range; double z; Arc<Material> material;
    AARect(std::pair<double, double> xr, std::pair<double, double> yr, double z_, Arc<Material> m) : x_range(xr), y_range(yr), z(z_), material(std::move(m)) {}
    std::optional<AABB> bounding_box(double, double) const override {
        return AABB{DVec3(x_range.first, y_range.first, z - 0.0001), DVec3(x_range.second, y_range.second, z + 0.0001)};
    }
};

// ---- mesh.rs ----------------------------------------------------------------------------------
struct Mesh {  // mesh.rs:8-13
    std::vector<DVec3> vertices, normals; std::vector<DVec2> uv; Arc<Material> material;
};
struct Triangle : Hittable {  // mesh.rs:42-45,109-120
    uint32_t indices[3]; Arc<Mesh> data;
    std::optional<AABB> bounding_box(double, double) const override {
        DVec3 mn = DVec3::splat(1.7976931348623157e308), mx = DVec3::splat(-1.7976931348623157e308);
        for (uint32_t i : indices) { const DVec3& v = data->vertices.at(i); mn = mn.min(v); mx = mx.max(v); }
        return AABB{mn, mx};
    }
};
inline std::vector<Triangle> create_mesh(std::vector<DVec3> vertices, std::vector<DVec3> normals, std::vector<DVec2> uv,
                                         const std::vector<std::array<uint32_t, 3>>& indices, Arc<Material> material) {  // mesh.rs:15-39
    auto mesh = std::make_shared<Mesh>();
    mesh->vertices = std::move(vertices); mesh->normals = std::move(normals); mesh->uv = std::move(uv); mesh->material = std::move(material);
    std::vector<Triangle> triangles;
    triangles.reserve(indices.size());
    for (const auto& ix : indices) { Triangle t; t.indices[0] = ix[0]; t.indices[1] = ix[1]; t.indices[2] = ix[2]; t.data = mesh; triangles.push_back(std::move(t)); }
    return triangles;
}

// ---- transform.rs -----------------------------------------------------------------------------
struct Transformed : Hittable {
    DMat4 t, t_inv; Arc<Hittable> hittable;
    static Transformed new_(const DMat4& transformation, Arc<Hittable> h) {  // transform.rs:14-20
        Transformed x; x.t = transformation; x.t_inv = transformation.inverse(); x.hittable = std::move(h); return x;
    }
    std::optional<AABB> bounding_box(double time_0, double time_1) const override {  // transform.rs:51-61: min & max corners only
        auto bb = hittable->bounding_box(time_0, time_1);
        if (!bb) return std::nullopt;
        DVec3 mx = t.mul_vec4(DVec4::from(bb->max, 1.0)).xyz();
        DVec3 mn = t.mul_vec4(DVec4::from(bb->min, 1.0)).xyz();
        return AABB{mn, mx};
    }
};

// ---- bvh.rs -----------------------------------------------------------------------------------
// Split-axis source.  bvh.rs:53 draws `thread_rng().gen_range(0..=2)`; here every BVH build starts a
// fresh splitmix64 stream from the process-wide seed below, axis = (next >> 33) % 3, one draw per
// node with >= 3 objects, in construction (pre-) order.
//
// Because that RNG is unseeded, EVERY assignment of axes to nodes is a tree the reference can build (each with probability 3^-nodes),
// and its `hit` results on any of them are "the reference's".  Axis mode 1 (BVH_AXIS_MIN_AREA) uses that freedom: at each node it takes,
// of the three axes bvh.rs:53 can draw, the one whose median split (the reference's own: stable sort by box minimum, cut at len/2)
// has the smallest  area(left box)*|left| + area(right box)*|right|,  lowest axis on ties, no draws.  Everything else -- the sort key,
// the cut, the boxes, the traversal -- stays bvh.rs.  The pagoda walks 25 % fewer records per ray on such a tree.
inline uint64_t& bvh_axis_seed() { static uint64_t s = 0; return s; }
enum : uint32_t { BVH_AXIS_SEEDED = 0, BVH_AXIS_MIN_AREA = 1 };
inline uint32_t& bvh_axis_mode() { static uint32_t m = BVH_AXIS_SEEDED; return m; }
struct AxisStream {
    uint64_t state; uint32_t mode;
    explicit AxisStream(uint64_t seed, uint32_t mode_ = BVH_AXIS_SEEDED) : state(seed), mode(mode_) {}
    uint32_t gen_axis() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (uint32_t)((z >> 33) % 3ull);
    }
};

struct BVHNode : Hittable {
    Arc<Hittable> left, right; AABB bbox;
    bool inner = false;          // children are BVHNodes made by this build
    HittableList source;         // root only: the slice handed to new_ (kept for describe())
    double time_0 = 0, time_1 = 0;
    uint64_t axis_seed = 0;      // root only
    uint32_t axis_mode = 0;      // root only

    static BVHNode from_mesh(std::vector<Triangle> triangles, double time_0, double time_1) {  // bvh.rs:27-34
        HittableList h;
        h.reserve(triangles.size());
        for (auto& t : triangles) h.push_back(std::make_shared<Triangle>(std::move(t)));
        return new_(h, time_0, time_1);
    }
    static BVHNode new_(const HittableList& hittables, double time_0, double time_1) {  // bvh.rs:36-87
        AxisStream axes(bvh_axis_seed(), bvh_axis_mode());
        BVHNode n = build(hittables.data(), hittables.size(), time_0, time_1, axes);
        n.source = hittables; n.time_0 = time_0; n.time_1 = time_1; n.axis_seed = bvh_axis_seed(); n.axis_mode = bvh_axis_mode();
        return n;
    }
    std::optional<AABB> bounding_box(double, double) const override { return bbox; }

  private:
    static BVHNode build(const Arc<Hittable>* h, size_t len, double time_0, double time_1, AxisStream& axes) {
        if (len == 0) throw std::runtime_error("BVHNode::new: empty slice");
        BVHNode node;
        if (len <= 2) {
            node.left = h[0]; node.right = h[len - 1];
            auto b = renderer::bounding_box(h, len, time_0, time_1);
            if (!b) throw std::runtime_error("BVHNode::new: object without a bounding box (bvh.rs:41 unwrap)");
            node.bbox = *b;
            return node;
        }
        std::vector<AABB> boxes(len);
        for (size_t i = 0; i < len; ++i) {
            auto b = h[i]->bounding_box(time_0, time_1);
            if (!b) throw std::runtime_error("BVHNode::new: object without a bounding box");
            boxes[i] = *b;
        }
        std::vector<std::pair<double, size_t>> order(len);
        auto sort_by = [&](uint32_t axis) {
            for (size_t i = 0; i < len; ++i) order[i] = {boxes[i].min[axis], i};
            std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });  // sort_by is stable
        };
        uint32_t sort_axis = 0;
        if (axes.mode == BVH_AXIS_SEEDED) sort_axis = axes.gen_axis();
        else {   // BVH_AXIS_MIN_AREA: the cheapest of the three median splits the reference could draw
            auto half_cost = [&](size_t b, size_t e) {
                DVec3 lo = boxes[order[b].second].min, hi = boxes[order[b].second].max;
                for (size_t i = b + 1; i < e; ++i) { lo = lo.min(boxes[order[i].second].min); hi = hi.max(boxes[order[i].second].max); }
                const DVec3 d = hi - lo;
                return (d.x * d.y + d.y * d.z + d.z * d.x) * (double)(e - b);
            };
            double best = 0.0;
            for (uint32_t axis = 0; axis < 3; ++axis) {
                sort_by(axis);
                const double c = half_cost(0, len / 2) + half_cost(len / 2, len);
                if (axis == 0 || c < best) { best = c; sort_axis = axis; }
            }
        }
        sort_by(sort_axis);
        HittableList objects(len);
        for (size_t i = 0; i < len; ++i) objects[i] = h[order[i].second];
        const size_t midpoint = len / 2;
        auto l = std::make_shared<BVHNode>(build(objects.data(), midpoint, time_0, time_1, axes));
        auto r = std::make_shared<BVHNode>(build(objects.data() + midpoint, len - midpoint, time_0, time_1, axes));
        node.bbox = AABB::surrounding_box(l->bbox, r->bbox);
        node.left = l; node.right = r; node.inner = true;
        return node;
    }
};

// ---- camera.rs --------------------------------------------------------------------------------
struct Camera {
    // constructor arguments (kept for describe())
    DVec3 look_from, look_at, up; double vfov = 0, aspect_ratio = 1, aperture = 0, focus_dist = 1, time_0 = 0, time_1 = 0;
    // derived fields, camera.rs:9-19
    DVec3 origin, lower_left_corner, horizontal, vertical, u, v; double lens_radius = 0;
    static Camera new_(DVec3 look_from, DVec3 look_at, DVec3 up, double vfov, double aspect_ratio, double aperture, double focus_dist,
                       double time_0, double time_1) {  // camera.rs:22-59
        Camera c;
        c.look_from = look_from; c.look_at = look_at; c.up = up; c.vfov = vfov; c.aspect_ratio = aspect_ratio; c.aperture = aperture;
        c.focus_dist = focus_dist; c.time_0 = time_0; c.time_1 = time_1;
        const double theta = vfov * (3.14159265358979323846 / 180.0);
        const double h = std::tan(theta / 2.0);
        const double viewport_height = 2.0 * h;
        const double viewport_width = aspect_ratio * viewport_height;
        const DVec3 w = (look_from - look_at).normalize();
        c.u = up.cross(w).normalize();
        c.v = w.cross(c.u);
        c.origin = look_from;
        c.horizontal = c.u * focus_dist * viewport_width;
        c.vertical = c.v * focus_dist * viewport_height;
        c.lower_left_corner = c.origin - (c.horizontal / 2.0) - (c.vertical / 2.0) - (w * focus_dist);
        c.lens_radius = aperture / 2.0;
        return c;
    }
    static Camera new_instant(DVec3 look_from, DVec3 look_at, DVec3 up, double vfov, double aspect_ratio, double aperture, double focus_dist) {  // camera.rs:61-81
        return new_(look_from, look_at, up, vfov, aspect_ratio, aperture, focus_dist, 0.0, 0.0);
    }
};

// ---- integrator.rs: the four integrators are selected by id; ray_colour itself runs on the GPU --
struct Integrator { virtual ~Integrator() {} virtual int abi_id() const = 0; virtual const char* name() const = 0; };
struct MultipleImportanceSampleIntegrator : Integrator { int abi_id() const override { return RT_INT_MIS; } const char* name() const override { return "MultipleImportanceSampleIntegrator"; } };
struct ImportanceSampleLightIntegrator : Integrator { int abi_id() const override { return RT_INT_LIGHT_IS; } const char* name() const override { return "ImportanceSampleLightIntegrator"; } };
struct BRDFSampledPathIntegrator : Integrator { int abi_id() const override { return RT_INT_BRDF; } const char* name() const override { return "BRDFSampledPathIntegrator"; } };
struct UniformSampledPathIntegrator : Integrator { int abi_id() const override { return RT_INT_UNIFORM; } const char* name() const override { return "UniformSampledPathIntegrator"; } };

// ---- scene.rs ---------------------------------------------------------------------------------
enum class Background { NoLight = RT_BG_BLACK, Skybox = RT_BG_SKY };  // main.rs:37-48

struct SceneBuilder;
struct Scene {  // scene.rs:11-17
    HittableList objects;
    std::vector<Arc<SampleableLight>> lights;
    Camera camera;
    Background background = Background::NoLight;
    std::optional<BVHNode> bvh;
    static SceneBuilder build();

    std::string describe() const;          // "rtscene 1" text (grammar: oracle/orc_scene_desc.hpp header)
    std::vector<uint8_t> flatten() const;  // "RTSB" blob for rt_scene_upload
};

struct SceneBuilder {  // scene.rs:63-114
    Scene scene; bool build_bvh_ = false;
    SceneBuilder objects(HittableList objs) && { scene.objects = std::move(objs); return std::move(*this); }
    SceneBuilder add_objects(const HittableList& objs) && { for (auto& o : objs) scene.objects.push_back(o); return std::move(*this); }
    void add_object(Arc<Hittable> obj) { scene.objects.push_back(std::move(obj)); }
    SceneBuilder background(Background b) && { scene.background = b; return std::move(*this); }
    SceneBuilder lights(std::vector<Arc<SampleableLight>> l) && { scene.lights = std::move(l); return std::move(*this); }
    SceneBuilder camera(Camera c) && { scene.camera = c; return std::move(*this); }
    SceneBuilder build_bvh() && { build_bvh_ = true; return std::move(*this); }
    Scene build() && {  // scene.rs:106-113
        if (build_bvh_) scene.bvh = BVHNode::new_(scene.objects, 0.0, 0.0);
        return std::move(scene);
    }
};
inline SceneBuilder Scene::build() { return SceneBuilder(); }

// Parses an "rtscene 1" description back into a Scene (used for ad-hoc scenes from tests/tools).
Scene scene_from_description(const char* text, size_t len);

}  // namespace renderer
