// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  Never used by the product path.
//
// CPU restatement of the reference `renderer` crate hot path: pointer-style scene graph, recursive
// BVH, recursive integrators.  Every function cites the reference file:line it follows
// (paths relative to /root/reference).  Parity status: geometry pinned by the reference's own
// known-answer tests (tests/test_oracle_kat.py); everything radiometric is "parity unpinned"
// (the reference holds no test or recorded output for it) and follows the source text literally.
#pragma once
#include "orc_math.hpp"
#include <atomic>
#include <chrono>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace orc {

struct Ray {  // renderer/src/ray.rs:3-12
    V3 origin, dir;
    double time;
    V3 at(double t) const { return origin + dir * t; }
};

struct Counters { uint64_t rays = 0, box = 0, tri = 0, sph = 0, rect = 0, paths = 0; };
inline Counters& counters() { static thread_local Counters c; return c; }

// ---------------------------------------------------------------------------------------------
// renderer/src/bounding_box.rs
// ---------------------------------------------------------------------------------------------
struct AABB {
    V3 min, max;
    static AABB surrounding_box(const AABB& a, const AABB& b) {  // bounding_box.rs:11-25
        AABB r;
        r.min = V3(std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z));
        r.max = V3(std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z));
        return r;
    }
    // bounding_box.rs:27-46.  NOTE the `let` at lines 37-38 shadows t_min/t_max INSIDE the loop
    // body: every axis is tested against the ORIGINAL interval (three independent 1-D tests).
    bool hit(const Ray& ray, double t_min, double t_max) const {
        counters().box++;
        for (int a = 0; a < 3; ++a) {
            double inv_d = 1.0 / ray.dir[a];
            double t0 = (min[a] - ray.origin[a]) * inv_d;
            double t1 = (max[a] - ray.origin[a]) * inv_d;
            if (t1 < t0) std::swap(t0, t1);
            double lo = t0 > t_min ? t0 : t_min;
            double hi = t1 < t_max ? t1 : t_max;
            if (hi < lo) return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/texture.rs, renderer/src/image.rs:81-94, desktop/src/main.rs:891-910
// ---------------------------------------------------------------------------------------------
struct Texture {
    enum Kind { SOLID, CHECKER, IMAGE } kind = SOLID;
    V3 colour;                      // SOLID
    const Texture* odd = nullptr;   // CHECKER
    const Texture* even = nullptr;
    uint32_t w = 0, h = 0;          // IMAGE
    std::vector<V3> data;
    V3 sample(double u, double v, V3 p) const {
        switch (kind) {
            case SOLID: return colour;  // texture.rs:14-18
            case CHECKER: {             // texture.rs:35-45
                double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
                return sines < 0.0 ? odd->sample(u, v, p) : even->sample(u, v, p);
            }
            default: {                  // image.rs:81-94
                double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);        // f64::clamp (NaN passes through)
                double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                double vv = 1.0 - vc;
                uint32_t x = sat_u32((double)w * uu), y = sat_u32((double)h * vv);
                x = std::min(x, w - 1); y = std::min(y, h - 1);
                return data[(size_t)y * w + x];
            }
        }
    }
    // signature of sample()'s discrete choices: one bit per checker node on the way down (leading 1), and the texel picked
    uint64_t sig(double u, double v, V3 p) const {
        const Texture* t = this;
        uint64_t bits = 1;
        for (int depth = 0; depth < 8 && t->kind == CHECKER; ++depth) {
            double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
            bits = (bits << 1) | (sines < 0.0 ? 1u : 0u);
            t = sines < 0.0 ? t->odd : t->even;
        }
        if (t->kind == IMAGE) {
            double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
            double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            double vv = 1.0 - vc;
            uint32_t x = sat_u32((double)t->w * uu), y = sat_u32((double)t->h * vv);
            x = std::min(x, t->w - 1); y = std::min(y, t->h - 1);
            bits ^= ((uint64_t)y * t->w + x + 1) << 16;
        }
        return bits;
    }
    static uint32_t sat_u32(double d) {  // Rust `as u32`: saturating, NaN -> 0
        if (!(d > 0.0)) return 0;
        if (d >= 4294967295.0) return 4294967295u;
        return (uint32_t)d;
    }
    // main.rs:891-910: texel = (byte/255)^2 per channel
    void load_rgb8(uint32_t w_, uint32_t h_, const uint8_t* rgb) {
        kind = IMAGE; w = w_; h = h_; data.resize((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            double r = (double)rgb[3 * i] / 255.0, g = (double)rgb[3 * i + 1] / 255.0, b = (double)rgb[3 * i + 2] / 255.0;
            data[i] = V3(r * r, g * g, b * b);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/pdf.rs  (Box<dyn PDF> restated as a tagged value)
// ---------------------------------------------------------------------------------------------
inline double reflectance(double cosine, double ref_idx) {  // material.rs:87-92 (powi(5) = x*((x*x)*(x*x)))
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosine;
    double x2 = x * x;
    double p5 = x * (x2 * x2);
    return r0 + ((1.0 - r0) * p5);
}

struct Pdf {
    enum Kind { COSINE, UNIFORM_HEMI, UNIFORM_SPHERE, FUZZY_DELTA, DIELECTRIC, CONE } kind = COSINE;
    OrthoNormalBasis basis;          // COSINE / UNIFORM_HEMI / CONE
    double cos_theta_max = 0;        // CONE
    V3 center; double radius = 0;    // UNIFORM_SPHERE
    V3 dir; double fuzz = 0;         // FUZZY_DELTA (dir normalised at construction, pdf.rs:87-91)
    V3 reflect_dir, refract_dir; double refl = 0; bool cannot_refract = false;  // DIELECTRIC

    static Pdf cosine(V3 n) { Pdf p; p.kind = COSINE; p.basis = OrthoNormalBasis::from_w(n); return p; }            // pdf.rs:21-27
    static Pdf uniform_hemi(V3 n) { Pdf p; p.kind = UNIFORM_HEMI; p.basis = OrthoNormalBasis::from_w(n); return p; }  // pdf.rs:49-55
    static Pdf uniform_sphere(V3 c, double r) { Pdf p; p.kind = UNIFORM_SPHERE; p.center = c; p.radius = r; return p; }
    static Pdf fuzzy(V3 d, double f) { Pdf p; p.kind = FUZZY_DELTA; p.dir = normalize(d); p.fuzz = f; return p; }     // pdf.rs:112-119
    static Pdf cone(V3 n, double c) { Pdf p; p.kind = CONE; p.basis = OrthoNormalBasis::from_w(n); p.cos_theta_max = c; return p; }  // pdf.rs:231-238
    static Pdf dielectric(const Ray& ray_in, V3 normal, double ior) {  // pdf.rs:142-160
        Pdf p; p.kind = DIELECTRIC;
        V3 unit_direction = normalize(ray_in.dir);
        double cos_theta = std::fmin(dot(-unit_direction, normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        p.cannot_refract = (ior * sin_theta) > 1.0;
        p.reflect_dir = normalize(reflect(unit_direction, normal));
        p.refract_dir = normalize(refract(unit_direction, normal, ior));
        p.refl = reflectance(cos_theta, ior);
        return p;
    }
    bool is_delta() const { return kind == FUZZY_DELTA || kind == DIELECTRIC; }
    double value(V3 direction) const {
        switch (kind) {
            case COSINE: {  // pdf.rs:30-38
                double cosine = dot(normalize(direction), basis.w);
                return cosine <= 0.0 ? 0.0 : cosine / PI;
            }
            case UNIFORM_HEMI: return 0.5 * FRAC_1_PI;                                  // pdf.rs:58-60
            case UNIFORM_SPHERE: return 1.0 / (4.0 * PI * (radius * radius));           // pdf.rs:73-75
            case CONE: return 1.0 / (2.0 * PI * (1.0 - cos_theta_max));                 // pdf.rs:241-250 (no in-cone test)
            default: return 0.0;                                                        // pdf.rs:95-97,163-165
        }
    }
    V3 generate(Rng& rng) const {
        switch (kind) {
            case COSINE: return normalize(basis.local(rand_cosine_hemisphere(rng)));   // pdf.rs:40-42
            case UNIFORM_HEMI: return normalize(basis.local(rand_hemisphere(rng)));    // pdf.rs:62-64
            case UNIFORM_SPHERE: return (normalize(rand_in_unit_sphere(rng)) * radius) + center;  // pdf.rs:77-79 (a POINT)
            case FUZZY_DELTA: {                                                         // pdf.rs:126-128
                V3 s = rand_in_unit_sphere(rng);
                return normalize(dir + (s * fuzz));
            }
            case DIELECTRIC: {                                                          // pdf.rs:167-173
                if (cannot_refract) return reflect_dir;   // `||` short-circuits: no draw consumed
                return refl > rng.next_f64() ? reflect_dir : refract_dir;
            }
            default: {                                                                  // pdf.rs:252-262
                double r1 = rng.next_f64();
                double cos_theta = (1.0 - r1) + (r1 * cos_theta_max);
                double sin_theta = std::sqrt(1.0 - (cos_theta * cos_theta));
                double phi = rng.next_f64() * 2.0 * PI;
                return basis.local(V3(std::cos(phi) * sin_theta, std::sin(phi) * sin_theta, cos_theta));
            }
        }
    }
};

// MixturePDF with MixtureMethod::Uniform (pdf.rs:185-224).  The PowerHeuristic arm is todo!() in the reference, and no
// integrator instantiates MixturePDF; restated because it is part of the PDF set.
struct MixturePdf {
    std::vector<Pdf> pdfs;
    double value(V3 direction) const {   // pdf.rs:199-210: weight = 1 / len; map(value * weight).sum()
        const double weight = 1.0 / (double)pdfs.size();
        double sum = 0.0;
        for (const Pdf& p : pdfs) sum = sum + (p.value(direction) * weight);
        return sum;
    }
    V3 generate(Rng& rng) const {        // pdf.rs:212-222: choose(rng).generate(rng).normalize()
        const Pdf& p = pdfs[rng.index((uint32_t)pdfs.size())];
        return normalize(p.generate(rng));
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/hit.rs, renderer/src/material.rs
// ---------------------------------------------------------------------------------------------
struct Material;
struct HitRecord {
    V3 point, normal;
    const Material* material = nullptr;
    double t = 0, u = 0, v = 0;
    bool front_face = true;
    uint32_t prim_id = 0xFFFFFFFFu;  // oracle bookkeeping: innermost primitive's object id
    uint32_t inst_id = 0xFFFFFFFFu;  // oracle bookkeeping: outermost Transformed wrapper's object id
    // hit.rs:19-47
    static HitRecord make(const Ray& ray, V3 point, V3 outward_normal, const Material* m, double t, double u, double v) {
        HitRecord hr; hr.point = point; hr.material = m; hr.t = t; hr.u = u; hr.v = v;
        hr.front_face = dot(ray.dir, outward_normal) < 0.0;
        hr.normal = hr.front_face ? outward_normal : -outward_normal;
        return hr;
    }
};

struct Material {
    enum Kind { LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT } kind = LAMBERTIAN;
    const Texture* tex = nullptr;  // albedo (LAMBERTIAN) / emit_colour (DIFFUSE_LIGHT)
    V3 albedo; double fuzz = 0;    // METAL (fuzz clamped to <= 1 at construction, material.rs:55-62)
    double ior = 1;                // DIELECTRIC
    V3 emitted(double u, double v, V3 p) const {  // material.rs:10-12,125-127
        return kind == DIFFUSE_LIGHT ? tex->sample(u, v, p) : V3(0.0, 0.0, 0.0);
    }
    bool scattering_pdf(const Ray& ray_in, const HitRecord& hr, Pdf& out) const {
        switch (kind) {
            case LAMBERTIAN: out = Pdf::cosine(hr.normal); return true;                               // material.rs:32-36
            case METAL: out = Pdf::fuzzy(reflect(normalize(ray_in.dir), hr.normal), fuzz); return true;  // material.rs:71-74
            case DIELECTRIC: out = Pdf::dielectric(ray_in, hr.normal, hr.front_face ? 1.0 / ior : ior); return true;  // material.rs:95-106
            default: return false;                                                                    // material.rs:133-135
        }
    }
    V3 brdf(const Ray&, const HitRecord& hr, const Ray& ray_out) const {
        switch (kind) {
            case LAMBERTIAN: return tex->sample(hr.u, hr.v, hr.point) / PI;           // material.rs:38-42
            case METAL: return albedo / dot(ray_out.dir, hr.normal);                  // material.rs:65-69
            case DIELECTRIC: return V3(1.0, 1.0, 1.0) / dot(ray_out.dir, hr.normal);  // material.rs:108-112
            default: return V3(0.0, 0.0, 0.0);                                        // material.rs:129-131
        }
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/hittable.rs
// ---------------------------------------------------------------------------------------------
struct Hittable {
    uint32_t id = 0xFFFFFFFFu;
    virtual ~Hittable() {}
    virtual bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const = 0;
    virtual bool bounding_box(double time_0, double time_1, AABB& out) const = 0;
};

// hittable.rs:41-53 / 80-92: linear closest-hit fold (later object replaces on hit with t_max = prev.t)
inline bool list_hit(Hittable* const* objs, size_t n, const Ray& ray, double t_min, double t_max, HitRecord& out) {
    bool have = false;
    for (size_t i = 0; i < n; ++i) {
        HitRecord h;
        if (objs[i]->hit(ray, t_min, have ? out.t : t_max, h)) { out = h; have = true; }
    }
    return have;
}
// hittable.rs:55-69 / 93-107
inline bool list_bounding_box(Hittable* const* objs, size_t n, double t0, double t1, AABB& out) {
    if (n == 0) return false;
    if (!objs[0]->bounding_box(t0, t1, out)) return false;
    for (size_t i = 1; i < n; ++i) {
        AABB b;
        if (!objs[i]->bounding_box(t0, t1, b)) return false;
        out = AABB::surrounding_box(out, b);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// renderer/src/shape.rs
// ---------------------------------------------------------------------------------------------
inline void spherical_uv(V3 p, double& u, double& v) {  // shape.rs:73-91
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}
// shape.rs:93-128
inline bool spherical_hit(V3 center, double radius, const Material* m, const Ray& ray, double t_min, double t_max, HitRecord& out) {
    counters().sph++;
    V3 oc = ray.origin - center;
    double a = length_squared(ray.dir);
    double half_b = dot(oc, ray.dir);
    double c = length_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrt_discriminant = std::sqrt(discriminant);
    double root = (-half_b - sqrt_discriminant) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrt_discriminant) / a;
        if (root < t_min || root > t_max) return false;
    }
    double t = root;
    V3 point = ray.at(t);
    V3 normal = (point - center) / radius;
    double u, v;
    spherical_uv(normal, u, v);
    out = HitRecord::make(ray, point, normal, m, t, u, v);
    return true;
}

struct Sphere : Hittable {  // shape.rs:17-21,130-139
    V3 center; double radius = 0; const Material* material = nullptr;
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {
        if (!spherical_hit(center, radius, material, ray, t_min, t_max, out)) return false;
        out.prim_id = id; out.inst_id = 0xFFFFFFFFu; return true;
    }
    bool bounding_box(double, double, AABB& out) const override {
        out.min = center - V3(radius, radius, radius);
        out.max = center + V3(radius, radius, radius);
        return true;
    }
    // SampleableLight::pdf_for_point, shape.rs:36-53
    Pdf pdf_for_point(V3 point) const {
        if (length_squared(point - center) <= radius * radius) return Pdf::uniform_sphere(center, radius);
        double sin_theta_max_2 = (radius * radius) / distance_squared(point, center);
        double cos_theta_max = std::sqrt(std::fmax(0.0, 1.0 - sin_theta_max_2));
        return Pdf::cone(normalize(center - point), cos_theta_max);
    }
};

struct MovingSphere : Hittable {  // shape.rs:202-247
    V3 center_0, center_1; double time_0 = 0, time_1 = 1, radius = 0; const Material* material = nullptr;
    V3 center(double time) const {  // shape.rs:214-217
        return center_0 + (center_1 - center_0) * ((time - time_0) / (time_1 - time_0));
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {
        if (!spherical_hit(center(ray.time), radius, material, ray, t_min, t_max, out)) return false;
        out.prim_id = id; out.inst_id = 0xFFFFFFFFu; return true;
    }
    bool bounding_box(double t0, double t1, AABB& out) const override {
        V3 c0 = center(t0), c1 = center(t1), rv(radius, radius, radius);
        AABB a{c0 - rv, c0 + rv}, b{c1 - rv, c1 + rv};
        out = AABB::surrounding_box(a, b);
        return true;
    }
};

struct AARect : Hittable {  // shape.rs:258-305
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0, z = 0; const Material* material = nullptr;
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {
        counters().rect++;
        double t = (z - ray.origin.z) / ray.dir.z;
        if (t < t_min || t > t_max) return false;
        V3 p = ray.at(t);
        if (p.x < x0 || p.x > x1 || p.y < y0 || p.y > y1) return false;
        double width = x1 - x0, height = y1 - y0;
        double u = (p.x - x0) / width, v = (p.y - y0) / height;
        out = HitRecord::make(ray, p, V3(0.0, 0.0, 1.0), material, t, u, v);
        out.prim_id = id; out.inst_id = 0xFFFFFFFFu;
        return true;
    }
    bool bounding_box(double, double, AABB& out) const override {
        out.min = V3(x0, y0, z - 0.0001); out.max = V3(x1, y1, z + 0.0001); return true;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/mesh.rs
// ---------------------------------------------------------------------------------------------
struct Mesh {  // mesh.rs:8-13
    std::vector<V3> vertices, normals;
    std::vector<double> uv;  // 2 per vertex
    const Material* material = nullptr;
};
struct Triangle : Hittable {  // mesh.rs:42-120
    uint32_t idx[3] = {0, 0, 0};
    const Mesh* data = nullptr;
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {
        counters().tri++;
        V3 v0 = data->vertices[idx[0]], v1 = data->vertices[idx[1]], v2 = data->vertices[idx[2]];
        V3 edge1 = v1 - v0, edge2 = v2 - v0;
        V3 h = cross(ray.dir, edge2);
        double det = dot(edge1, h);
        if (det > -F64_EPSILON && det < F64_EPSILON) return false;
        double det_inv = 1.0 / det;
        V3 s = ray.origin - v0;
        double u = det_inv * dot(s, h);
        if (!(u >= 0.0 && u <= 1.0)) return false;   // !(0.0..=1.0).contains(&u): NaN rejects
        V3 q = cross(s, edge1);
        double v = det_inv * dot(ray.dir, q);
        if (v < 0.0 || (u + v) > 1.0) return false;
        double t = det_inv * dot(edge2, q);
        if (t <= F64_EPSILON || t > t_max || t <= t_min) return false;
        V3 n0 = data->normals[idx[0]], n1 = data->normals[idx[1]], n2 = data->normals[idx[2]];
        double w = 1.0 - (u + v);
        V3 n = ((u * n0) + (v * n1)) + (w * n2);     // mesh.rs:95 (u weights vertex 0: reference quirk)
        const double* uv = data->uv.data();
        double tu = ((u * uv[2 * idx[0]]) + (v * uv[2 * idx[1]])) + (w * uv[2 * idx[2]]);
        double tv = ((u * uv[2 * idx[0] + 1]) + (v * uv[2 * idx[1] + 1])) + (w * uv[2 * idx[2] + 1]);
        out.point = ray.at(t); out.material = data->material; out.normal = n; out.u = tu; out.v = tv; out.t = t;
        out.front_face = dot(ray.dir, n) > 0.0;      // mesh.rs:105 (opposite sense to hit.rs:41)
        out.prim_id = id; out.inst_id = 0xFFFFFFFFu;
        return true;
    }
    bool bounding_box(double, double, AABB& out) const override {  // mesh.rs:109-120
        V3 mn = V3::splat(1.7976931348623157e308), mx = V3::splat(-1.7976931348623157e308);
        for (int k = 0; k < 3; ++k) { V3 p = data->vertices[idx[k]]; mn = vmin(mn, p); mx = vmax(mx, p); }
        out.min = mn; out.max = mx; return true;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/transform.rs
// ---------------------------------------------------------------------------------------------
struct Transformed : Hittable {
    M4 t, t_inv; const Hittable* hittable = nullptr;
    void set(const M4& m, const Hittable* h) { t = m; t_inv = m.inverse(); hittable = h; }  // transform.rs:14-20
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {   // transform.rs:24-49
        V4 new_origin = t_inv.mul(V4(ray.origin, 1.0));
        V4 new_dir = t_inv.mul(V4(ray.dir, 0.0));
        Ray tr{new_origin.xyz(), new_dir.xyz(), ray.time};
        M4 inverse_transpose = t_inv.transpose();
        HitRecord hr;
        if (!hittable->hit(tr, t_min, t_max, hr)) return false;
        out = hr;
        out.point = ray.at(hr.t);
        out.normal = normalize(inverse_transpose.mul(V4(hr.normal, 0.0)).xyz());
        out.inst_id = id;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB& out) const override {  // transform.rs:51-61 (min/max corners only)
        AABB bb;
        if (!hittable->bounding_box(t0, t1, bb)) return false;
        out.max = t.mul(V4(bb.max, 1.0)).xyz();
        out.min = t.mul(V4(bb.min, 1.0)).xyz();
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/bvh.rs
// ---------------------------------------------------------------------------------------------
// The reference draws the split axis from thread_rng (bvh.rs:53), so its tree differs per run.
// The oracle injects the axis sequence: splitmix64 stream, axis = (next >> 33) % 3, drawn at node
// entry (pre-order), only for slices of length >= 3 -- exactly where bvh.rs:53 draws.
// Any axis sequence is a tree the reference can build.  mode 1 ("min area"): instead of drawing, a node takes the axis whose median
// split -- the one bvh.rs:55-77 would make for that axis -- minimises  sum over the two halves of  half_area(box of the half) * count,
// half_area = dx*dy + dy*dz + dz*dx; axis 0 unless a later axis is strictly cheaper.
struct AxisRng {
    uint64_t s; uint32_t mode;
    explicit AxisRng(uint64_t seed, uint32_t mode_ = 0) : s(seed), mode(mode_) {}
    uint32_t next_axis() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (uint32_t)((z >> 33) % 3ull);
    }
};
// Every BVH build (scene BVH and each BVHNode object) starts a fresh stream from the scene's bvh_seed.

struct BVHNode : Hittable {
    const Hittable* left = nullptr; const Hittable* right = nullptr; AABB bbox;
    std::vector<std::unique_ptr<BVHNode>> owned;  // children created by build()
    // bvh.rs:36-87
    void build(Hittable* const* hittables, size_t n, double time_0, double time_1, AxisRng& rng) {
        if (n == 0) throw std::runtime_error("BVHNode::new on empty slice");
        if (n == 1) {
            left = hittables[0]; right = hittables[0];
            if (!list_bounding_box(hittables, n, time_0, time_1, bbox)) throw std::runtime_error("unbounded object in BVH");
            return;
        }
        if (n == 2) {
            left = hittables[0]; right = hittables[1];
            if (!list_bounding_box(hittables, n, time_0, time_1, bbox)) throw std::runtime_error("unbounded object in BVH");
            return;
        }
        struct Key { double k; Hittable* h; AABB box; };
        std::vector<Key> keys(n);
        auto sorted_on = [&](int axis) {   // bvh.rs:55-73 for one axis
            std::vector<Key> out(n);
            for (size_t i = 0; i < n; ++i) {
                AABB b;
                if (!hittables[i]->bounding_box(time_0, time_1, b)) throw std::runtime_error("unbounded object in BVH");
                out[i] = Key{b.min[axis], hittables[i], b};
            }
            std::stable_sort(out.begin(), out.end(), [](const Key& a, const Key& b) { return a.k < b.k; });
            return out;
        };
        if (rng.mode == 0) keys = sorted_on((int)rng.next_axis());
        else {
            double cheapest = 0.0;
            for (int axis = 0; axis < 3; ++axis) {
                std::vector<Key> cand = sorted_on(axis);
                double total = 0.0;
                const size_t cut[3] = {0, n / 2, n};
                for (int half = 0; half < 2; ++half) {
                    double lo[3], hi[3];
                    for (int k = 0; k < 3; ++k) { lo[k] = cand[cut[half]].box.min[k]; hi[k] = cand[cut[half]].box.max[k]; }
                    for (size_t i = cut[half] + 1; i < cut[half + 1]; ++i)
                        for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], cand[i].box.min[k]); hi[k] = std::fmax(hi[k], cand[i].box.max[k]); }
                    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                    const double part = (dx * dy + dy * dz + dz * dx) * (double)(cut[half + 1] - cut[half]);
                    total = half == 0 ? part : total + part;
                }
                if (axis == 0 || total < cheapest) { cheapest = total; keys = std::move(cand); }
            }
        }
        std::vector<Hittable*> objects(n);
        for (size_t i = 0; i < n; ++i) objects[i] = keys[i].h;
        size_t midpoint = n / 2;
        auto l = std::make_unique<BVHNode>(); auto r = std::make_unique<BVHNode>();
        l->build(objects.data(), midpoint, time_0, time_1, rng);
        r->build(objects.data() + midpoint, n - midpoint, time_0, time_1, rng);
        bbox = AABB::surrounding_box(l->bbox, r->bbox);
        left = l.get(); right = r.get();
        owned.push_back(std::move(l)); owned.push_back(std::move(r));
    }
    // bvh.rs:91-106
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const override {
        if (!bbox.hit(ray, t_min, t_max)) return false;
        HitRecord lh, rh;
        bool l = left->hit(ray, t_min, t_max, lh);
        double tm = l ? lh.t : t_max;
        bool r = right->hit(ray, t_min, tm, rh);
        if (r) { out = rh; return true; }
        if (l) { out = lh; return true; }
        return false;
    }
    bool bounding_box(double, double, AABB& out) const override { out = bbox; return true; }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/camera.rs
// ---------------------------------------------------------------------------------------------
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, u, v;
    double lens_radius = 0, time_0 = 0, time_1 = 0;
    void set(V3 look_from, V3 look_at, V3 up, double vfov, double aspect_ratio, double aperture, double focus_dist,
             double t0, double t1) {  // camera.rs:22-59
        double theta = vfov * (PI / 180.0);
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        V3 w = normalize(look_from - look_at);
        u = normalize(cross(up, w));
        v = cross(w, u);
        origin = look_from;
        horizontal = u * focus_dist * viewport_width;
        vertical = v * focus_dist * viewport_height;
        lower_left_corner = origin - (horizontal / 2.0) - (vertical / 2.0) - (w * focus_dist);
        lens_radius = aperture / 2.0; time_0 = t0; time_1 = t1;
    }
    Ray get_ray(double s, double t, Rng& rng) const {  // camera.rs:83-101
        V3 random_in_lens = rand_in_unit_sphere(rng) * lens_radius;
        V3 offset = (u * random_in_lens.x) + (v * random_in_lens.y);
        double time = (time_1 - time_0) > 0.000001 ? rng.range(time_0, time_1) : 0.0;
        Ray r;
        r.origin = origin + offset;
        r.dir = lower_left_corner + (horizontal * s) + (vertical * t) - origin - offset;
        r.time = time;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/scene.rs + desktop/src/main.rs:37-48 (backgrounds)
// ---------------------------------------------------------------------------------------------
enum Background { BG_BLACK = 0, BG_SKY = 1 };
inline V3 background_colour(int bg, const Ray& ray) {
    if (bg == BG_SKY) {  // main.rs:37-44   (t = 0.5*y + 1.0 as written)
        V3 unit_dir = normalize(ray.dir);
        double t = 0.5 * unit_dir.y + 1.0;
        return V3(1.0, 1.0, 1.0) * (1.0 - t) + V3(0.5, 0.7, 1.0) * t;
    }
    return V3(0.0, 0.0, 0.0);  // main.rs:46-48
}

struct Scene {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Mesh>> meshes;
    std::vector<M4> xforms;
    std::vector<std::unique_ptr<Hittable>> pool;  // every object of the description, by id
    std::vector<Hittable*> objects;               // scene.rs:12
    std::vector<const Sphere*> lights;            // scene.rs:13 (only Sphere implements SampleableLight)
    Camera camera;
    int background = BG_BLACK;
    std::unique_ptr<BVHNode> bvh;                 // scene.rs:16
    uint64_t bvh_seed = 0;
    uint32_t bvh_axis_mode = 0;   // 0: seeded axis stream, 1: min-area axis per node (AxisRng)

    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& out) const {  // scene.rs:31-38
        counters().rays++;
        if (bvh) return bvh->hit(ray, t_min, t_max, out);
        return list_hit(objects.data(), objects.size(), ray, t_min, t_max, out);
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/integrator.rs
// ---------------------------------------------------------------------------------------------
enum Integrator { INT_MIS = 0, INT_LIGHT_IS = 1, INT_BRDF = 2, INT_UNIFORM = 3 };

struct SampleCtx { uint64_t seed; uint32_t pixel, sample; int max_depth; };

inline double power_heuristic(uint32_t num_f, double pdf_f, uint32_t num_g, double pdf_g) {  // integrator.rs:10-15
    double f = (double)num_f * pdf_f, g = (double)num_g * pdf_g;
    return (f * f) / ((f * f) + (g * g));
}
inline double balance_heuristic(double pdf_f, double pdf_g) { return pdf_f / (pdf_f + pdf_g); }  // integrator.rs:17-19

inline bool light_hit(const Sphere* light, const Ray& ray, double t_min, double t_max, HitRecord& out) {
    uint64_t save = counters().sph;  // single-light re-intersections are not counted (BASELINE.md "Ray definition")
    bool r = light->hit(ray, t_min, t_max, out);
    counters().sph = save;
    return r;
}

V3 ray_colour(int integrator, const Ray& ray, const Scene& scene, int depth, const SampleCtx& ctx);

// path-signature events common to the four integrators (tests only; no effect on the arithmetic)
inline void sig_hit_events(uint32_t level, bool hit, const HitRecord& hr) {
    if (!g_sig) return;
    sig_event(level, SIG_HIT, hit ? ((uint64_t)hr.prim_id | ((uint64_t)hr.inst_id << 32)) : 0xFFFFFFFFFFFFFFFFull);
    if (!hit) return;
    sig_event(level, SIG_FRONT, hr.front_face ? 1 : 0);
    if (hr.material->kind == Material::LAMBERTIAN || hr.material->kind == Material::DIFFUSE_LIGHT) sig_event(level, SIG_TEX, hr.material->tex->sig(hr.u, hr.v, hr.point));
}
inline void sig_delta_event(uint32_t level, const Pdf& pdf, V3 generated) {   // reflected (1) or refracted (0), pdf.rs:167-173
    if (g_sig && pdf.kind == Pdf::DIELECTRIC)
        sig_event(level, SIG_DELTA, (generated.x == pdf.reflect_dir.x && generated.y == pdf.reflect_dir.y && generated.z == pdf.reflect_dir.z) ? 1 : 0);
}

// integrator.rs:31-171
inline V3 ray_colour_mis(const Ray& ray, const Scene& scene, int depth, const SampleCtx& ctx) {
    if (depth <= 0) return V3(0.0, 0.0, 0.0);
    Rng rng(ctx.seed, ctx.pixel, ctx.sample, (uint32_t)(ctx.max_depth - depth + 1));
    const uint32_t level = (uint32_t)(ctx.max_depth - depth);
    HitRecord hr;
    const bool any_hit = scene.hit(ray, 0.001, 100000.0, hr);
    sig_hit_events(level, any_hit, hr);
    if (!any_hit) return background_colour(scene.background, ray);
    V3 emitted = hr.material->emitted(hr.u, hr.v, hr.point);
    Pdf material_pdf;
    if (!hr.material->scattering_pdf(ray, hr, material_pdf)) { sig_event(level, SIG_KIND, 0); return emitted; }
    sig_event(level, SIG_KIND, material_pdf.is_delta() ? 1 : 2);
    if (material_pdf.is_delta()) {  // integrator.rs:54-66
        const V3 generated = material_pdf.generate(rng);
        sig_delta_event(level, material_pdf, generated);
        Ray ray_out{hr.point, normalize(generated), ray.time};
        double cos_theta = dot(ray_out.dir, hr.normal);
        return emitted + (hr.material->brdf(ray, hr, ray_out) * cos_theta * ray_colour_mis(ray_out, scene, depth - 1, ctx));
    }
    // integrator.rs:68-88
    const Sphere* light = nullptr; Pdf light_pdf; Ray light_ray{};
    if (!scene.lights.empty()) {
        const uint32_t li = rng.index((uint32_t)scene.lights.size());
        const Sphere* cand = scene.lights[li];
        Pdf lp = cand->pdf_for_point(hr.point);
        V3 dir = normalize(lp.generate(rng));
        Ray visibility_ray{hr.point, dir, ray.time};
        bool sample_counts = false;   // signature only: the light sample faces the surface and reaches the light's own sphere
        if (!(dot(hr.normal, dir) < 0.0)) {
            HitRecord vis, lh;
            if (g_sig) { HitRecord tmp; sample_counts = light_hit(cand, visibility_ray, 0.001, 10000.0, tmp); }
            if (scene.hit(visibility_ray, 0.001, 10000.0, vis) && light_hit(cand, visibility_ray, 0.001, 10000.0, lh) &&
                !(std::fabs(vis.t - lh.t) > 0.0001)) {
                light = cand; light_pdf = lp; light_ray = visibility_ray;
            }
        }
        sig_event(level, SIG_LIGHT, 2ull * li + (sample_counts ? 1 : 0));
        if (sample_counts) sig_event(level, SIG_VIS, light ? 1 : 0);
    }
    if (light) {  // integrator.rs:90-154
        Ray material_out{hr.point, normalize(material_pdf.generate(rng)), ray.time};
        HitRecord light_h;
        if (!light_hit(light, light_ray, 0.001, 10000.0, light_h)) throw std::runtime_error("Should hit");
        double light_pdf_value = light_pdf.value(light_ray.dir);
        double material_cos_theta = dot(material_out.dir, hr.normal);
        double material_out_pdf = material_pdf.value(material_out.dir);
        V3 light_emit = light_h.material->emitted(light_h.u, light_h.v, hr.point);
        HitRecord tmp;
        bool material_ray_hit_light = light_hit(light, material_out, 0.001, 10000.0, tmp);
        V3 material_ray_colour = ray_colour_mis(material_out, scene, depth - 1, ctx);
        bool material_ray_has_light = material_ray_colour.x >= 0.001 && material_ray_colour.y >= 0.001 && material_ray_colour.z >= 0.001;
        sig_event(level, SIG_GATE, material_ray_has_light ? 1 : 0);
        double material_weight = material_ray_has_light
            ? power_heuristic(1, material_out_pdf, 1, material_ray_hit_light ? light_pdf.value(material_out.dir) : 0.0)
            : 0.0;
        V3 material_contribution = hr.material->brdf(ray, hr, material_out) * material_cos_theta * material_ray_colour * material_weight / material_out_pdf;
        double light_weight = power_heuristic(1, light_pdf_value, 1, material_pdf.value(light_ray.dir));
        double light_cos_theta = std::fmax(dot(hr.normal, light_ray.dir), 0.0);
        V3 light_contribution = hr.material->brdf(ray, hr, light_ray) * light_cos_theta * light_emit * light_weight / light_pdf_value;
        return emitted + material_contribution + light_contribution;
    }
    // integrator.rs:155-169
    Ray ray_out{hr.point, normalize(material_pdf.generate(rng)), ray.time};
    double cos_theta = dot(ray_out.dir, hr.normal);
    double pdf = material_pdf.value(ray_out.dir);
    return emitted + (hr.material->brdf(ray, hr, ray_out) * cos_theta * ray_colour_mis(ray_out, scene, depth - 1, ctx)) / pdf;
}

// integrator.rs:174-260
inline V3 ray_colour_light_is(const Ray& ray, const Scene& scene, int depth, const SampleCtx& ctx) {
    if (depth <= 0) return V3(0.0, 0.0, 0.0);
    Rng rng(ctx.seed, ctx.pixel, ctx.sample, (uint32_t)(ctx.max_depth - depth + 1));
    const uint32_t level = (uint32_t)(ctx.max_depth - depth);
    HitRecord hr;
    const bool any_hit = scene.hit(ray, 0.001, 100000.0, hr);
    sig_hit_events(level, any_hit, hr);
    if (!any_hit) return background_colour(scene.background, ray);
    V3 emitted = hr.material->emitted(hr.u, hr.v, hr.point);
    Pdf material_pdf;
    const bool scatters = hr.material->scattering_pdf(ray, hr, material_pdf);
    if (!scatters) sig_event(level, SIG_KIND, 0);
    if (length(emitted) > 0.0) return emitted;
    if (!scatters) return emitted;
    sig_event(level, SIG_KIND, material_pdf.is_delta() ? 1 : 2);
    Pdf scatter_pdf = material_pdf;
    if (!material_pdf.is_delta() && !scene.lights.empty()) {
        const uint32_t li = rng.index((uint32_t)scene.lights.size());
        scatter_pdf = scene.lights[li]->pdf_for_point(hr.point);
        sig_event(level, SIG_LIGHT, 2ull * li);
    }
    V3 out_dir = scatter_pdf.generate(rng);
    sig_delta_event(level, material_pdf, out_dir);
    Ray ray_out{hr.point, out_dir, ray.time};
    double cos_theta = dot(out_dir, hr.normal);
    sig_event(level, SIG_CUT, cos_theta < 0.0 ? 1 : 0);
    if (cos_theta < 0.0) return emitted;  // integrator.rs:206-215 (the regenerated sample is discarded)
    double pdf = scatter_pdf.is_delta() ? 1.0 : scatter_pdf.value(out_dir);
    V3 brdf = hr.material->brdf(ray, hr, ray_out);
    return emitted + (brdf * cos_theta * ray_colour_light_is(ray_out, scene, depth - 1, ctx)) / pdf;
}

// integrator.rs:263-327 (uniform = false) and 330-397 (uniform = true)
inline V3 ray_colour_path(bool uniform, const Ray& ray, const Scene& scene, int depth, const SampleCtx& ctx) {
    if (depth <= 0) return V3(0.0, 0.0, 0.0);
    Rng rng(ctx.seed, ctx.pixel, ctx.sample, (uint32_t)(ctx.max_depth - depth + 1));
    const uint32_t level = (uint32_t)(ctx.max_depth - depth);
    HitRecord hr;
    const bool any_hit = scene.hit(ray, 0.001, 100000.0, hr);
    sig_hit_events(level, any_hit, hr);
    if (!any_hit) return background_colour(scene.background, ray);
    V3 emitted = hr.material->emitted(hr.u, hr.v, hr.point);
    Pdf material_pdf;
    if (!hr.material->scattering_pdf(ray, hr, material_pdf)) { sig_event(level, SIG_KIND, 0); return emitted; }
    sig_event(level, SIG_KIND, material_pdf.is_delta() ? 1 : 2);
    Pdf scatter_pdf = (uniform && !material_pdf.is_delta()) ? Pdf::uniform_hemi(hr.normal) : material_pdf;
    V3 out_dir = scatter_pdf.generate(rng);
    sig_delta_event(level, material_pdf, out_dir);
    Ray ray_out{hr.point, out_dir, ray.time};
    double cos_theta = dot(out_dir, hr.normal);
    double pdf = scatter_pdf.is_delta() ? 1.0 : scatter_pdf.value(out_dir);
    V3 brdf = hr.material->brdf(ray, hr, ray_out);
    return emitted + (brdf * cos_theta * ray_colour_path(uniform, ray_out, scene, depth - 1, ctx)) / pdf;
}

inline V3 ray_colour(int integrator, const Ray& ray, const Scene& scene, int depth, const SampleCtx& ctx) {
    switch (integrator) {
        case INT_MIS: return ray_colour_mis(ray, scene, depth, ctx);
        case INT_LIGHT_IS: return ray_colour_light_is(ray, scene, depth, ctx);
        case INT_BRDF: return ray_colour_path(false, ray, scene, depth, ctx);
        default: return ray_colour_path(true, ray, scene, depth, ctx);
    }
}

// ---------------------------------------------------------------------------------------------
// desktop/src/main.rs:955-1019 (render loop) + 913-921 (ACES) + renderer/src/image.rs
// ---------------------------------------------------------------------------------------------
inline Ray primary_ray(const Scene& scene, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t sample, uint64_t seed) {
    Rng rng(seed, y * width + x, sample, 0);
    double u = ((double)x + rng.next_f64()) / (double)(width - 1);   // main.rs:996
    double v = ((double)y + rng.next_f64()) / (double)(height - 1);  // main.rs:997
    return scene.camera.get_ray(u, v, rng);                          // main.rs:999
}

inline V3 aces_tonemapping(V3 pixel) {  // main.rs:913-921
    const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
    V3 num = pixel * (a * pixel + V3::splat(b));
    V3 den = pixel * (c * pixel + V3::splat(d)) + V3::splat(e);
    V3 r = num / den;
    auto cl = [](double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); };
    return V3(cl(r.x), cl(r.y), cl(r.z));
}

struct ImageTile {  // image.rs:13-26
    uint32_t ox, oy, w, h;
    void get_xy(uint32_t index, uint32_t& x, uint32_t& y) const { x = (index % w) + ox; y = (index / w) + oy; }
};
inline ImageTile get_tile(uint32_t img_w, uint32_t img_h, uint32_t ox, uint32_t oy, uint32_t tw, uint32_t th) {  // image.rs:61-70
    ImageTile t; t.ox = ox; t.oy = oy;
    t.w = std::min(ox + tw, img_w) - ox; t.h = std::min(oy + th, img_h) - oy;
    return t;
}

struct RenderStats { uint64_t rays = 0, paths = 0, box = 0, tri = 0, sph = 0; double seconds = 0; };

// Sum (not mean) of radiance over samples [spp_begin, spp_end) per pixel; row 0 = bottom row
// (ppm.rs:15 writes rows reversed).  16x16 tiles handed out dynamically to `nthreads` workers --
// what rayon's into_par_iter at main.rs:979 does when enabled; nthreads = 1 is what the surveyed
// commit does (main.rs:980).
inline void render_sum(const Scene& scene, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_end,
                       int max_depth, int integrator, uint64_t seed, int nthreads, double* sum_rgb, RenderStats* stats) {
    const uint32_t tile_size = 16;
    const uint32_t tiles_x = (width + tile_size - 1) / tile_size, tiles_y = (height + tile_size - 1) / tile_size;
    std::atomic<uint32_t> next_tile{0};
    std::vector<Counters> per_thread((size_t)std::max(1, nthreads));
    auto worker = [&](int tid) {
        counters() = Counters();
        for (;;) {
            uint32_t tile_index = next_tile.fetch_add(1);
            if (tile_index >= tiles_x * tiles_y) break;
            ImageTile tile = get_tile(width, height, (tile_index % tiles_x) * tile_size, (tile_index / tiles_x) * tile_size, tile_size, tile_size);
            for (uint32_t index = 0; index < tile.w * tile.h; ++index) {
                uint32_t x, y; tile.get_xy(index, x, y);
                V3 colour(0.0, 0.0, 0.0);
                for (uint32_t s = spp_begin; s < spp_end; ++s) {
                    Ray ray = primary_ray(scene, width, height, x, y, s, seed);
                    SampleCtx ctx{seed, y * width + x, s, max_depth};
                    colour = colour + ray_colour(integrator, ray, scene, max_depth, ctx);
                    counters().paths++;
                }
                size_t o = ((size_t)y * width + x) * 3;
                sum_rgb[o] = colour.x; sum_rgb[o + 1] = colour.y; sum_rgb[o + 2] = colour.z;
            }
        }
        per_thread[(size_t)tid] = counters();
    };
    auto t0 = std::chrono::steady_clock::now();
    if (nthreads <= 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        *stats = RenderStats();
        for (auto& c : per_thread) { stats->rays += c.rays; stats->paths += c.paths; stats->box += c.box; stats->tri += c.tri; stats->sph += c.sph; }
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
    }
}

// main.rs:1009-1013 + image.rs:43-59: mean -> ACES (optional) -> NaN channel -> 0
inline void resolve(const double* sum_rgb, uint32_t width, uint32_t height, uint32_t spp, int tonemap, double* out_rgb) {
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        V3 c(sum_rgb[3 * i], sum_rgb[3 * i + 1], sum_rgb[3 * i + 2]);
        c = c / (double)spp;
        if (tonemap) c = aces_tonemapping(c);
        if (std::isnan(c.x)) c.x = 0.0;
        if (std::isnan(c.y)) c.y = 0.0;
        if (std::isnan(c.z)) c.z = 0.0;
        out_rgb[3 * i] = c.x; out_rgb[3 * i + 1] = c.y; out_rgb[3 * i + 2] = c.z;
    }
}

// desktop/src/exporters/ppm.rs:27-31: sqrt gamma via powf(0.5), clamp [0, .999], *256 truncate; rows bottom-up.
inline void quantize_ppm(const double* rgb, uint32_t width, uint32_t height, uint32_t* out /* w*h*3, top row first */) {
    size_t k = 0;
    for (uint32_t yy = height; yy-- > 0;)
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) {
                double p = std::pow(rgb[((size_t)yy * width + x) * 3 + c], 0.5);
                double cl = p < 0.0 ? 0.0 : (p > 0.999 ? 0.999 : p);  // NaN passes through clamp; `as u32` maps NaN to 0
                double q = cl * 256.0;
                out[k++] = (q != q) ? 0u : (uint32_t)q;
            }
}

}  // namespace orc
