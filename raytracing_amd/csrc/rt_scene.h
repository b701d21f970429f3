// Device view of the flattened scene + primitive intersection / hit-record / material / PDF
// device functions.  Each function cites the reference lines it implements (paths relative to
// /root/reference).  f64 throughout: on CDNA4 a wave64 f64 VALU op issues at the same rate as an
// unpacked f32 one, so the reference's f64 semantics are kept rather than down-converted.
#pragma once
#include "../../include/rt_abi.h"
#include "rt_math.h"

namespace rt {

// Pointers into the resident blob.  Node / primitive / geometry tables are read through the
// constant address space from wave-uniform indices, which makes hipcc emit scalar loads
// (s_load_dwordx*): one fetch per wave instead of 64 per-lane fetches.
#define RT_CONST __attribute__((address_space(4)))
template <class T> using cptr = const RT_CONST T*;
template <class T> RT_DEV cptr<T> as_const(const T* p) { return (cptr<T>)(uintptr_t)p; }

// Strict pruning of a whole BVHNode instance (rt_trace.hip.h, inst_line_misses): per INSTANCE primitive, the box of the triangles of its
// sub-array in their own (object) space, the error constants of DESIGN.md section 5 for that space, and the (unit normal, plane offset)
// of every triangle -- few enough for the certificate to be a plain loop.  ok == 0: the instance is always walked.
struct InstCull { double lo[3], hi[3]; double tau_coef, delta_coef, radius; uint32_t ok, n_planes, first_plane, pad; };
struct InstPlane { double n[3], q; };
#define RT_INST_CULL_MAX_TRIS 64u

struct SceneDev {
    const InstCull* inst_cull;      // [n_prims] (entries of INSTANCE primitives are meaningful), or null: scene without instances
    const InstPlane* inst_planes;
    const rt_node* nodes;
    const rt_prim* prims;
    const rt_geom* geom;  // parallel to prims
    const rt_tri_attrs* tri_attrs;
    const rt_xform* xforms;
    const rt_material* materials;
    const rt_texture* textures;
    const uint8_t* texels;
    const rt_light* lights;
    const uint8_t* mat_uses_uv;
    const uint8_t* stream;  // trace stream (rt_trace.hip.h): fp32 filter records in pre-order; production form (always-entered inner nodes have no record)
    const uint32_t* node_off;  // [n_nodes + 1] byte offset of each node's record in `stream` (a node without record: its first child's)
    const uint8_t* stream_full;     // the same with a record for every node: counting kernel, NaN-capable waves
    const uint32_t* node_off_full;
    uint32_t root_off_full;
    double box_abs_max;        // max |coordinate| over the finite boxes (box filter error bound)
    // strict pruning (rt_trace.hip.h, StrictFilter / ray_certified): one transform space, its direction grid and error constants
    const uint32_t* cert_cells;    // [3 K K + 1] CSR offsets into cert_entries
    const void* cert_entries;      // CertEntry[]: fp32 normal + index into cert_normals, lists padded to multiples of 4
    const double* cert_normals;    // 3 per distinct unit normal of the pruned space
    const double* cert_planes;       // per normal cluster: sorted plane offsets v0 . n^ of its triangles
    const uint32_t* cert_plane_off;  // [n_normals + 1] CSR offsets into cert_planes
    const uint32_t* cert_plane_leaf; // parallel to cert_planes: node index of the triangle's leaf
    double prune_tau_coef, prune_delta_coef, prune_radius, prune_box_abs_max;
    double prune_time0, prune_time1;  // covered moving spheres are boxed over this shutter; rays outside it are not pruned
    uint32_t prune_on, prune_xf, cert_n_normals;
    uint32_t root_off;  // first scene-level record in `stream` (the range is closed by an END record)
    uint32_t end_off, end_off_full;  // that END record's offset in `stream` / `stream_full`
    rt_camera camera;
    uint32_t n_lights, root_node, root_end, background;
    uint32_t n_nodes, boxes_finite;
    uint32_t translate_xf;  // a transform whose inverse is exactly identity + a translation with three finite non-zero components (RT_NONE: none):
                            // for rays without zero / non-finite components apply_xform<true> adds the translation instead of multiplying the matrix out
    uint32_t oracle_best;   // experiment builds only (-DRT_ORACLE_BEST): start every walk from the hit a previous launch left in out_hit (launch_trace's copy)
    uint32_t count_steps;   // per launch (launch_trace's copy): k_trace adds its waves' record steps to stats[15] -- one same-address atomic per wave, 5 % of the frame, so only on request
    float sort_min[3], sort_scale[3];  // origin quantisation grid of the ray-sorting key
};

// ---- bounding_box.rs:27-46, literal: three independent per-axis tests against the ORIGINAL
// [t_min, t_max] (the `let` at lines 37-38 shadows inside the loop body).  inv = 1/dir is hoisted
// per ray: the same IEEE division the reference performs per test.
RT_DEV bool box_axis(double bmin, double bmax, double o, double inv, double t_min, double t_max) {
    double t0 = (bmin - o) * inv;
    double t1 = (bmax - o) * inv;
    bool sw = t1 < t0;
    double a = sw ? t1 : t0, b = sw ? t0 : t1;
    double lo = a > t_min ? a : t_min;
    double hi = b < t_max ? b : t_max;
    return !(hi < lo);
}

// Fast form, 8 VALU ops per axis.  With a = min(t0,t1), b = max(t0,t1):
//   literal: lo = max(a, t_min), hi = min(b, t_max), pass iff !(hi < lo)
//        <=> a <= t_max  &&  b >= t_min            (given t_min <= t_max and a <= b)
// Bit-for-bit the same decision as box_axis whenever t0, t1 are not NaN (min/max reproduce the swap;
// comparisons ignore the sign of zero).  t0/t1 cannot be NaN when the box, the origin and 1/dir are
// finite and 1/dir != 0; t_max (the running best) never drops below t_min because every primitive
// test rejects t < t_min.  The kernel checks these preconditions per wave and otherwise uses box_axis.
// v_min_f64 / v_max_f64 are issued directly: fmin()/fmax() would add a canonicalising v_max x,x each.
RT_DEV double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV bool box_axis_fast(double bmin, double bmax, double o, double inv, double t_min, double t_max) {
    double t0 = (bmin - o) * inv;
    double t1 = (bmax - o) * inv;
    return (vmin_f64(t0, t1) <= t_max) & (vmax_f64(t0, t1) >= t_min);
}

// ---- primitive tests: return true and t when the reference's `hit` returns Some --------------
// mesh.rs:48-81 (Moller-Trumbore, two-sided, exact statement order)
// edge1 = v1 - v0 and edge2 = v2 - v0 (mesh.rs:55-56) come precomputed from the host (rt_geom)
RT_DEV bool tri_hit(d3 v0, d3 edge1, d3 edge2, d3 o, d3 d, double t_min, double t_max, double& t_out, double& u_out, double& v_out) {
    d3 h = cross(d, edge2);
    double det = dot(edge1, h);
    if (det > -F64_EPS && det < F64_EPS) return false;
    double det_inv = 1.0 / det;
    d3 s = o - v0;
    double u = det_inv * dot(s, h);
    if (!(u >= 0.0 && u <= 1.0)) return false;
    d3 q = cross(s, edge1);
    double v = det_inv * dot(d, q);
    if (v < 0.0 || (u + v) > 1.0) return false;
    double t = det_inv * dot(edge2, q);
    if (t <= F64_EPS || t > t_max || t <= t_min) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}
// Wave-friendly staging of the same test.  Stage A computes the reference's own h, det, s and s.h
// (same operations, same order) and applies a CONSERVATIVE rejection implied by the reference's
// tests, so the 1/det division and the rest run only for lanes that may still hit:
//   (E)  !(|det| >= EPS): the reference returns None at mesh.rs:59-61, or det is NaN and its `u` test fails;
//   for EPS <= |det| <= 1e100, det_inv = RN(1/det) is a normal number with relative error <= 2^-53, and
//   (R1) RN(s.h * det) < -1e-36  =>  s.h and det have opposite signs and |s.h / det| > 1e-36 / det^2 >= 1e-236, so
//        u = RN(det_inv * s.h) is a negative NORMAL number and `u >= 0` fails (a negative u that
//        underflowed to -0 would pass `u >= 0`, hence the magnitude floor);
//   (R2) |s.h| > RN(|det| * (1 + 2^-40))  =>  |det_inv * s.h| > (1 + 2^-41)(1 - 2^-53) > 1  =>  u < 0 or u > 1.
// Everything not rejected here is decided by the literal code of tri_hit_tail.
// Only det and s.h stay live between the stages (4 VGPRs per triangle); the rare tail recomputes s = o - v0.
struct TriStage { double det, sh; };
RT_DEV bool tri_stage_a(d3 v0, d3 edge1, d3 edge2, d3 o, d3 d, TriStage& st) {
    const d3 h = cross(d, edge2);
    st.det = dot(edge1, h);
    const d3 s = o - v0;
    st.sh = dot(s, h);
    const double ad = fabs(st.det);
    const bool e_ok = ad >= F64_EPS;
    const bool in_range = ad <= 1e100;
    const bool r1 = st.sh * st.det < -1e-36;
    const bool r2 = fabs(st.sh) > ad * 1.0000000000009094947;  // 1 + 2^-40
    return e_ok & !(in_range & (r1 | r2));
}
RT_DEV bool tri_hit_tail(const TriStage& st, d3 v0, d3 edge1, d3 edge2, d3 o, d3 d, double t_min, double t_max, double& t_out) {
    double det_inv = 1.0 / st.det;
    double u = det_inv * st.sh;
    if (!(u >= 0.0 && u <= 1.0)) return false;
    d3 q = cross(o - v0, edge1);
    double v = det_inv * dot(d, q);
    if (v < 0.0 || (u + v) > 1.0) return false;
    double t = det_inv * dot(edge2, q);
    if (t <= F64_EPS || t > t_max || t <= t_min) return false;
    t_out = t;
    return true;
}
// shape.rs:93-121
RT_DEV bool sphere_hit(d3 center, double radius, d3 o, d3 d, double t_min, double t_max, double& t_out) {
    d3 oc = o - center;
    double a = length_squared(d);
    double half_b = dot(oc, d);
    double c = length_squared(oc) - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrt_discriminant = __builtin_sqrt(discriminant);
    double root = (-half_b - sqrt_discriminant) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrt_discriminant) / a;
        if (root < t_min || root > t_max) return false;
    }
    t_out = root;
    return true;
}
RT_DEV d3 moving_center(d3 c0, d3 c1, double time0, double time1, double time) {  // shape.rs:214-217
    return c0 + (c1 - c0) * ((time - time0) / (time1 - time0));
}
// shape.rs:266-278
RT_DEV bool rect_hit(double x0, double x1, double y0, double y1, double z, d3 o, d3 d, double t_min, double t_max, double& t_out) {
    double t = (z - o.z) / d.z;
    if (t < t_min || t > t_max) return false;
    double x = o.x + d.x * t, y = o.y + d.y * t;
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    t_out = t;
    return true;
}

// ---- full hit record (hit.rs, mesh.rs:83-106, shape.rs:121-127, transform.rs:36-48) -------------
struct HitRecord {
    d3 point, normal;
    double t, u, v;
    uint32_t material;
    bool front_face;
};
RT_DEV void spherical_uv(d3 p, double& u, double& v) {  // shape.rs:73-91
    double theta = acos(-p.y);
    double phi = atan2(-p.z, p.x) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}
RT_DEV void set_face_normal(HitRecord& hr, d3 ray_dir, d3 outward_normal) {  // hit.rs:40-47
    hr.front_face = dot(ray_dir, outward_normal) < 0.0;
    hr.normal = hr.front_face ? outward_normal : -outward_normal;
}
RT_DEV bool material_uses_uv(const SceneDev& sc, uint32_t material);

// Rebuilds the reference HitRecord for the winning primitive.  `prim` / `inst` are indices into
// the prims table (inst = enclosing INSTANCE primitive or RT_NONE).  The primitive test is re-run
// on the same (possibly transformed) ray with t_max = t, which reproduces the same root / (u,v).
RT_DEV bool full_hit(const SceneDev& sc, uint32_t prim, uint32_t inst, const Ray& ray, double t_min, double t_hit, HitRecord& hr) {
    rt_prim p = sc.prims[prim];
    const rt_geom& g = sc.geom[prim];
    uint32_t xf = p.xform;
    if (inst != RT_NONE) xf = sc.prims[inst].xform;
    Ray r = ray;
    const double* m = nullptr;
    if (xf != RT_NONE) {  // transform.rs:25-32
        m = sc.xforms[xf].t_inv;
        r.o = mat_mul_xyz(m, ray.o, 1.0);
        r.d = mat_mul_xyz(m, ray.d, 0.0);
    }
    hr.material = p.material;
    double t = 0.0;
    if (p.kind == RT_PRIM_TRIANGLE) {
        double u, v;
        if (!tri_hit(ld3(g.d), ld3(g.d + 3), ld3(g.d + 6), r.o, r.d, t_min, t_hit, t, u, v)) return false;
        const rt_tri_attrs& ta = sc.tri_attrs[p.index];
        double w = 1.0 - (u + v);
        hr.normal = ((u * ld3(ta.n0)) + (v * ld3(ta.n1))) + (w * ld3(ta.n2));  // mesh.rs:95: u weights vertex 0 (reference quirk)
        hr.u = ((u * ta.uv0[0]) + (v * ta.uv1[0])) + (w * ta.uv2[0]);
        hr.v = ((u * ta.uv0[1]) + (v * ta.uv1[1])) + (w * ta.uv2[1]);
        hr.point = ray_at(r, t);
        hr.front_face = dot(r.d, hr.normal) > 0.0;  // mesh.rs:105
    } else if (p.kind == RT_PRIM_SPHERE || p.kind == RT_PRIM_MOVING_SPHERE) {
        d3 center; double radius;
        if (p.kind == RT_PRIM_SPHERE) { center = ld3(g.d); radius = g.d[3]; }
        else { center = moving_center(ld3(g.d), ld3(g.d + 3), g.d[6], g.d[7], r.time); radius = g.d[8]; }
        if (!sphere_hit(center, radius, r.o, r.d, t_min, t_hit, t)) return false;
        hr.point = ray_at(r, t);
        d3 normal = (hr.point - center) / radius;
        hr.u = 0.0; hr.v = 0.0;
        if (material_uses_uv(sc, p.material)) spherical_uv(normal, hr.u, hr.v);  // (u,v) only feed image textures
        set_face_normal(hr, r.d, normal);
    } else {  // RT_PRIM_AARECT, shape.rs:280-295
        const double x0 = g.d[0], x1 = g.d[1], y0 = g.d[2], y1 = g.d[3];
        if (!rect_hit(x0, x1, y0, y1, g.d[4], r.o, r.d, t_min, t_hit, t)) return false;
        hr.point = ray_at(r, t);
        hr.u = (hr.point.x - x0) / (x1 - x0);
        hr.v = (hr.point.y - y0) / (y1 - y0);
        set_face_normal(hr, r.d, mk(0.0, 0.0, 1.0));
    }
    hr.t = t;
    if (m) {  // transform.rs:38-48
        hr.point = ray_at(ray, t);
        hr.normal = normalize(mat_t_mul_xyz0(m, hr.normal));
    }
    return true;
}

// ---- texture.rs / image.rs:81-94 ------------------------------------------------------------------
RT_DEV uint32_t sat_u32(double d) {  // Rust `as u32`
    if (!(d > 0.0)) return 0u;
    if (d >= 4294967295.0) return 4294967295u;
    return (uint32_t)d;
}
RT_DEV d3 tex_sample(const SceneDev& sc, uint32_t tex, double u, double v, d3 p) {
    rt_texture t = sc.textures[tex];
    for (int depth = 0; depth < 8 && t.kind == RT_TEX_CHECKER; ++depth) {  // texture.rs:35-45
        double sines = sin(10.0 * p.x) * sin(10.0 * p.y) * sin(10.0 * p.z);
        t = sc.textures[sines < 0.0 ? t.a : t.b];
    }
    if (t.kind == RT_TEX_IMAGE) {
        double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
        double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        double vv = 1.0 - vc;
        uint32_t x = sat_u32((double)t.b * uu), y = sat_u32((double)t.c * vv);
        x = x < t.b - 1 ? x : t.b - 1; y = y < t.c - 1 ? y : t.c - 1;
        const uint8_t* px = sc.texels + t.a + ((size_t)y * t.b + x) * 3;
        double r = (double)px[0] / 255.0, g = (double)px[1] / 255.0, b = (double)px[2] / 255.0;  // main.rs:900-903
        return mk(r * r, g * g, b * b);
    }
    return ld3(t.colour);
}
// Signature of tex_sample's discrete choices: one bit per checker node on the way down (leading 1), and the texel picked
RT_DEV unsigned long long tex_sig(const SceneDev& sc, uint32_t tex, double u, double v, d3 p) {
    rt_texture t = sc.textures[tex];
    unsigned long long bits = 1ull;
    for (int depth = 0; depth < 8 && t.kind == RT_TEX_CHECKER; ++depth) {
        double sines = sin(10.0 * p.x) * sin(10.0 * p.y) * sin(10.0 * p.z);
        bits = (bits << 1) | (sines < 0.0 ? 1ull : 0ull);
        t = sc.textures[sines < 0.0 ? t.a : t.b];
    }
    if (t.kind == RT_TEX_IMAGE) {
        double uu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
        double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        double vv = 1.0 - vc;
        uint32_t x = sat_u32((double)t.b * uu), y = sat_u32((double)t.c * vv);
        x = x < t.b - 1 ? x : t.b - 1; y = y < t.c - 1 ? y : t.c - 1;
        bits ^= ((unsigned long long)y * t.b + x + 1ull) << 16;
    }
    return bits;
}
// mat_uses_uv[m] != 0 iff material m's texture tree contains an image texture (computed at upload)
RT_DEV bool material_uses_uv(const SceneDev& sc, uint32_t material) { return sc.mat_uses_uv[material] != 0; }

RT_DEV d3 material_emitted(const SceneDev& sc, uint32_t material, double u, double v, d3 p) {  // material.rs:10-12,125-127
    const rt_material& m = sc.materials[material];
    return m.kind == RT_MAT_DIFFUSE_LIGHT ? tex_sample(sc, m.texture, u, v, p) : mk(0.0, 0.0, 0.0);
}

// ---- pdf.rs as a tagged value -----------------------------------------------------------------------
enum PdfKind { PDF_COSINE, PDF_UNIFORM_HEMI, PDF_UNIFORM_SPHERE, PDF_FUZZY, PDF_DIELECTRIC, PDF_CONE };
struct Pdf {
    int kind;
    d3 a;      // COSINE/HEMI/CONE: basis w | SPHERE: center | FUZZY: dir | DIELECTRIC: reflect_dir
    d3 b;      // DIELECTRIC: refract_dir
    double s;  // CONE: cos_theta_max | SPHERE: radius | FUZZY: fuzz | DIELECTRIC: reflectance
    bool cannot_refract;
};
RT_DEV double reflectance(double cosine, double ref_idx) {  // material.rs:87-92, powi(5) = x*((x*x)*(x*x))
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    double x = 1.0 - cosine;
    double x2 = x * x;
    return r0 + ((1.0 - r0) * (x * (x2 * x2)));
}
RT_DEV bool pdf_is_delta(const Pdf& p) { return p.kind == PDF_FUZZY || p.kind == PDF_DIELECTRIC; }
RT_DEV double pdf_value(const Pdf& p, d3 direction) {
    switch (p.kind) {
        case PDF_COSINE: { double cosine = dot(normalize(direction), p.a); return cosine <= 0.0 ? 0.0 : cosine / PI; }  // pdf.rs:30-38
        case PDF_UNIFORM_HEMI: return 0.5 * FRAC_1_PI;                       // pdf.rs:58-60
        case PDF_UNIFORM_SPHERE: return 1.0 / (4.0 * PI * (p.s * p.s));      // pdf.rs:73-75
        case PDF_CONE: return 1.0 / (2.0 * PI * (1.0 - p.s));                // pdf.rs:241-250
        default: return 0.0;
    }
}
RT_DEV d3 pdf_generate(const Pdf& p, Rng& rng) {
    switch (p.kind) {
        case PDF_COSINE: { Onb b = onb_from_w(p.a); return normalize(onb_local(b, rand_cosine_hemisphere(rng))); }  // pdf.rs:40-42
        case PDF_UNIFORM_HEMI: { Onb b = onb_from_w(p.a); return normalize(onb_local(b, rand_hemisphere(rng))); }   // pdf.rs:62-64
        case PDF_UNIFORM_SPHERE: return (normalize(rand_in_unit_sphere(rng)) * p.s) + p.a;                          // pdf.rs:77-79
        case PDF_FUZZY: { d3 s = rand_in_unit_sphere(rng); return normalize(p.a + (s * p.s)); }                      // pdf.rs:126-128
        case PDF_DIELECTRIC: {                                                                                       // pdf.rs:167-173
            if (p.cannot_refract) return p.a;
            return p.s > rng.next_f64() ? p.a : p.b;
        }
        default: {                                                                                                   // pdf.rs:252-262
            Onb b = onb_from_w(p.a);
            double r1 = rng.next_f64();
            double cos_theta = (1.0 - r1) + (r1 * p.s);
            double sin_theta = __builtin_sqrt(1.0 - (cos_theta * cos_theta));
            double phi = rng.next_f64() * 2.0 * PI;
            double sn, cs;
            sincos(phi, &sn, &cs);
            return onb_local(b, mk(cs * sin_theta, sn * sin_theta, cos_theta));
        }
    }
}
// MixturePDF, MixtureMethod::Uniform (pdf.rs:185-224; the PowerHeuristic arm is todo!() in the reference): value = sum of
// value_i * (1 / n) in list order (Iterator::sum folds from 0.0), generate = one uniformly chosen member's direction, normalised.
// No integrator of the reference instantiates it; it is part of the PDF set and checked through rt_pdf_eval.
RT_DEV double pdf_mixture_value(const Pdf* pdfs, uint32_t n, d3 direction) {
    const double weight = 1.0 / (double)n;
    double sum = 0.0;
    for (uint32_t i = 0; i < n; ++i) sum = sum + (pdf_value(pdfs[i], direction) * weight);
    return sum;
}
RT_DEV d3 pdf_mixture_generate(const Pdf* pdfs, uint32_t n, Rng& rng) {
    const uint32_t pick = rng.index(n);  // SliceRandom::choose (pdf.rs:214-217) on the shared Philox stream
    return normalize(pdf_generate(pdfs[pick], rng));
}
// material.rs scattering_pdf: false for DiffuseLight
RT_DEV bool scattering_pdf(const rt_material& m, const Ray& ray_in, const HitRecord& hr, Pdf& out) {
    out.cannot_refract = false; out.b = mk(0, 0, 0); out.s = 0.0;
    if (m.kind == RT_MAT_LAMBERTIAN) { out.kind = PDF_COSINE; out.a = hr.normal; return true; }              // material.rs:32-36
    if (m.kind == RT_MAT_METAL) {                                                                              // material.rs:71-74, pdf.rs:87-91
        out.kind = PDF_FUZZY; out.a = normalize(reflect(normalize(ray_in.d), hr.normal)); out.s = m.fuzz; return true;
    }
    if (m.kind == RT_MAT_DIELECTRIC) {                                                                         // material.rs:95-106, pdf.rs:142-160
        double ior = hr.front_face ? 1.0 / m.ior : m.ior;
        d3 unit_direction = normalize(ray_in.d);
        double cos_theta = fmin(dot(-unit_direction, hr.normal), 1.0);
        double sin_theta = __builtin_sqrt(1.0 - cos_theta * cos_theta);
        out.kind = PDF_DIELECTRIC;
        out.cannot_refract = (ior * sin_theta) > 1.0;
        out.a = normalize(reflect(unit_direction, hr.normal));
        out.b = normalize(refract(unit_direction, hr.normal, ior));
        out.s = reflectance(cos_theta, ior);
        return true;
    }
    return false;
}
RT_DEV d3 material_brdf(const SceneDev& sc, const rt_material& m, const HitRecord& hr, d3 out_dir) {
    if (m.kind == RT_MAT_LAMBERTIAN) return tex_sample(sc, m.texture, hr.u, hr.v, hr.point) / PI;  // material.rs:38-42
    if (m.kind == RT_MAT_METAL) return ld3(m.albedo) / dot(out_dir, hr.normal);                   // material.rs:65-69
    if (m.kind == RT_MAT_DIELECTRIC) return mk(1.0, 1.0, 1.0) / dot(out_dir, hr.normal);          // material.rs:108-112
    return mk(0.0, 0.0, 0.0);
}
// Sphere::pdf_for_point, shape.rs:36-53
RT_DEV Pdf light_pdf_for_point(const rt_light& l, d3 point) {
    Pdf p; p.cannot_refract = false; p.b = mk(0, 0, 0);
    d3 c = ld3(l.center);
    if (length_squared(point - c) <= l.radius * l.radius) { p.kind = PDF_UNIFORM_SPHERE; p.a = c; p.s = l.radius; return p; }
    double sin_theta_max_2 = (l.radius * l.radius) / length_squared(point - c);
    p.kind = PDF_CONE; p.a = normalize(c - point); p.s = __builtin_sqrt(fmax(0.0, 1.0 - sin_theta_max_2));
    return p;
}
RT_DEV double power_heuristic(double pdf_f, double pdf_g) {  // integrator.rs:10-15 with num_f = num_g = 1
    double f = 1.0 * pdf_f, g = 1.0 * pdf_g;
    return (f * f) / ((f * f) + (g * g));
}
RT_DEV d3 background_colour(uint32_t bg, const Ray& ray) {  // main.rs:37-48
    if (bg == RT_BG_SKY) {
        d3 unit_dir = normalize(ray.d);
        double t = 0.5 * unit_dir.y + 1.0;
        return mk(1.0, 1.0, 1.0) * (1.0 - t) + mk(0.5, 0.7, 1.0) * t;
    }
    return mk(0.0, 0.0, 0.0);
}

}  // namespace rt
