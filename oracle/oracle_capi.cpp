// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  C entry points for ctypes
// (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  Build: oracle/Makefile.
#include "orc_scene_desc.hpp"
#include <cstdio>

using namespace orc;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH(ret) } catch (const std::exception& e) { g_err = e.what(); return ret; }

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_scene_load(const char* text, uint64_t len) {
    ORC_TRY
    return load_scene_desc(text, (size_t)len).release();
    ORC_CATCH(nullptr)
}
void orc_scene_free(void* s) { delete (Scene*)s; }

// rays: n x 7 doubles (ox,oy,oz,dx,dy,dz,time).  prim/inst = 0xFFFFFFFF and t = NaN-free 0 on miss.
// counters (optional, 4 x u64): box, tri, sph, rect test counts summed over the batch.
int orc_trace_closest(void* scene, uint32_t n, const double* rays, double t_min, double t_max,
                      uint32_t* prim, uint32_t* inst, double* t, uint64_t* cnt) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    counters() = Counters();
    for (uint32_t i = 0; i < n; ++i) {
        const double* r = rays + 7 * (size_t)i;
        Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6]};
        HitRecord hr;
        if (sc.hit(ray, t_min, t_max, hr)) { prim[i] = hr.prim_id; if (inst) inst[i] = hr.inst_id; t[i] = hr.t; }
        else { prim[i] = 0xFFFFFFFFu; if (inst) inst[i] = 0xFFFFFFFFu; t[i] = 0.0; }
    }
    if (cnt) { cnt[0] = counters().box; cnt[1] = counters().tri; cnt[2] = counters().sph; cnt[3] = counters().rect; }
    return 0;
    ORC_CATCH(-1)
}

// Full hit record for one ray: out[0..11] = point(3) normal(3) t u v front_face prim inst ; returns 1 on hit.
int orc_hit_record(void* scene, const double* r, double t_min, double t_max, double* out) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6]};
    HitRecord hr;
    if (!sc.hit(ray, t_min, t_max, hr)) return 0;
    out[0] = hr.point.x; out[1] = hr.point.y; out[2] = hr.point.z;
    out[3] = hr.normal.x; out[4] = hr.normal.y; out[5] = hr.normal.z;
    out[6] = hr.t; out[7] = hr.u; out[8] = hr.v; out[9] = hr.front_face ? 1.0 : 0.0;
    out[10] = (double)hr.prim_id; out[11] = (double)hr.inst_id;
    return 1;
    ORC_CATCH(-1)
}

// Camera rays of sample index `sample` for every pixel (row 0 = bottom), n = w*h rays x 7 doubles.
int orc_primary_rays(void* scene, uint32_t w, uint32_t h, uint32_t sample, uint64_t seed, double* rays) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            Ray r = primary_ray(sc, w, h, x, y, sample, seed);
            double* o = rays + 7 * ((size_t)y * w + x);
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z; o[6] = r.time;
        }
    return 0;
    ORC_CATCH(-1)
}

// stats (optional): 5 x u64 = rays, paths, box, tri, sph ; seconds (optional) = tile-loop wall time.
int orc_render(void* scene, uint32_t w, uint32_t h, uint32_t spp_begin, uint32_t spp_end, int max_depth, int integrator,
               uint64_t seed, int nthreads, double* sum_rgb, uint64_t* stats, double* seconds) {
    ORC_TRY
    RenderStats st;
    render_sum(*(Scene*)scene, w, h, spp_begin, spp_end, max_depth, integrator, seed, nthreads, sum_rgb, &st);
    if (stats) { stats[0] = st.rays; stats[1] = st.paths; stats[2] = st.box; stats[3] = st.tri; stats[4] = st.sph; }
    if (seconds) *seconds = st.seconds;
    return 0;
    ORC_CATCH(-1)
}

// Radiance of ONE camera sample (pixel x,y; sample s): the unit the GPU path is compared on.
int orc_sample_radiance(void* scene, uint32_t w, uint32_t h, uint32_t x, uint32_t y, uint32_t s, int max_depth, int integrator,
                        uint64_t seed, double* rgb) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    Ray ray = primary_ray(sc, w, h, x, y, s, seed);
    SampleCtx ctx{seed, y * w + x, s, max_depth};
    V3 c = ray_colour(integrator, ray, sc, max_depth, ctx);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return 0;
    ORC_CATCH(-1)
}

// Same, plus the path's signature (orc_math.hpp): the hash of every discrete decision the sample's path took
int orc_sample_radiance_sig(void* scene, uint32_t w, uint32_t h, uint32_t x, uint32_t y, uint32_t s, int max_depth, int integrator,
                            uint64_t seed, double* rgb, uint64_t* sig) {
    uint64_t acc = 0;
    g_sig = &acc;
    int rc = orc_sample_radiance(scene, w, h, x, y, s, max_depth, integrator, seed, rgb);
    g_sig = nullptr;
    *sig = acc;
    return rc;
}

int orc_resolve(const double* sum_rgb, uint32_t w, uint32_t h, uint32_t spp, int tonemap, double* out_rgb) {
    resolve(sum_rgb, w, h, spp, tonemap, out_rgb);
    return 0;
}
int orc_quantize_ppm(const double* rgb, uint32_t w, uint32_t h, uint32_t* out) { quantize_ppm(rgb, w, h, out); return 0; }

// ---- structure dumps used to compare the product's host-side BVH build / flattening ----------
// Pre-order walk of the scene BVH (or of object `obj_id` if it is a BVHNode; obj_id = 0xFFFFFFFF
// selects the scene BVH).  Per node: 6 doubles (min, max) and 2 u32 = object ids of (left,right)
// when they are NOT BVHNodes, else 0xFFFFFFFF.  Returns node count (or -1); pass null to count.
static int64_t dump_node(const BVHNode* n, double* boxes, uint32_t* kids, int64_t at) {
    if (boxes) { double* b = boxes + 6 * at; b[0] = n->bbox.min.x; b[1] = n->bbox.min.y; b[2] = n->bbox.min.z; b[3] = n->bbox.max.x; b[4] = n->bbox.max.y; b[5] = n->bbox.max.z; }
    const BVHNode* l = n->owned.empty() ? nullptr : n->owned[0].get();
    const BVHNode* r = n->owned.empty() ? nullptr : n->owned[1].get();
    if (kids) { kids[2 * at] = l ? 0xFFFFFFFFu : n->left->id; kids[2 * at + 1] = r ? 0xFFFFFFFFu : n->right->id; }
    int64_t next = at + 1;
    if (l) next = dump_node(l, boxes, kids, next);
    if (r) next = dump_node(r, boxes, kids, next);
    return next;
}
int64_t orc_bvh_dump(void* scene, uint32_t obj_id, double* boxes, uint32_t* kids) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    const BVHNode* root = nullptr;
    if (obj_id == 0xFFFFFFFFu) root = sc.bvh.get();
    else if (obj_id < sc.pool.size()) root = dynamic_cast<const BVHNode*>(sc.pool[obj_id].get());
    if (!root) { g_err = "no such BVH"; return -1; }
    return dump_node(root, boxes, kids, 0);
    ORC_CATCH(-1)
}

int orc_object_bbox(void* scene, uint32_t obj_id, double t0, double t1, double* out6) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    if (obj_id >= sc.pool.size()) return -1;
    AABB b;
    if (!sc.pool[obj_id]->bounding_box(t0, t1, b)) return 0;
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z; out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
    return 1;
    ORC_CATCH(-1)
}
// hittable.rs:93-107 on an explicit id list
int orc_list_bbox(void* scene, uint32_t n, const uint32_t* ids, double t0, double t1, double* out6) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    std::vector<Hittable*> v;
    for (uint32_t i = 0; i < n; ++i) v.push_back(sc.pool.at(ids[i]).get());
    AABB b;
    if (!list_bounding_box(v.data(), v.size(), t0, t1, b)) return 0;
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z; out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
    return 1;
    ORC_CATCH(-1)
}
// Single object hit (KATs that call .hit on one primitive / wrapper): out as orc_hit_record.
int orc_object_hit(void* scene, uint32_t obj_id, const double* r, double t_min, double t_max, double* out) {
    ORC_TRY
    const Scene& sc = *(Scene*)scene;
    Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6]};
    HitRecord hr;
    if (!sc.pool.at(obj_id)->hit(ray, t_min, t_max, hr)) return 0;
    out[0] = hr.point.x; out[1] = hr.point.y; out[2] = hr.point.z;
    out[3] = hr.normal.x; out[4] = hr.normal.y; out[5] = hr.normal.z;
    out[6] = hr.t; out[7] = hr.u; out[8] = hr.v; out[9] = hr.front_face ? 1.0 : 0.0;
    out[10] = (double)hr.prim_id; out[11] = (double)hr.inst_id;
    return 1;
    ORC_CATCH(-1)
}
int orc_aabb_hit(const double* box6, const double* r, double t_min, double t_max) {
    AABB b{V3(box6[0], box6[1], box6[2]), V3(box6[3], box6[4], box6[5])};
    Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6]};
    return b.hit(ray, t_min, t_max) ? 1 : 0;
}
void orc_philox(uint32_t* ctr4, uint32_t k0, uint32_t k1) { philox4x32_10(ctr4, k0, k1); }
// n consecutive draws of the (seed, pixel, sample, level) stream
void orc_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t level, uint32_t n, double* out) {
    Rng rng(seed, pixel, sample, level);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next_f64();
}
// image.rs tile helpers (KAT-8..10)
void orc_get_tile(uint32_t iw, uint32_t ih, uint32_t ox, uint32_t oy, uint32_t tw, uint32_t th, uint32_t* out4) {
    ImageTile t = get_tile(iw, ih, ox, oy, tw, th);
    out4[0] = t.ox; out4[1] = t.oy; out4[2] = t.w; out4[3] = t.h;
}
void orc_tile_xy(uint32_t ox, uint32_t oy, uint32_t w, uint32_t h, uint32_t index, uint32_t* xy) {
    ImageTile t{ox, oy, w, h};
    t.get_xy(index, xy[0], xy[1]);
}
// 4x4 inverse (column-major in/out) for checking the host side's DMat4::inverse restatement
void orc_mat4_inverse(const double* m, double* out) {
    M4 x; x.c0 = V4(m[0], m[1], m[2], m[3]); x.c1 = V4(m[4], m[5], m[6], m[7]); x.c2 = V4(m[8], m[9], m[10], m[11]); x.c3 = V4(m[12], m[13], m[14], m[15]);
    M4 r = x.inverse();
    const V4* c[4] = {&r.c0, &r.c1, &r.c2, &r.c3};
    for (int i = 0; i < 4; ++i) { out[4 * i] = c[i]->x; out[4 * i + 1] = c[i]->y; out[4 * i + 2] = c[i]->z; out[4 * i + 3] = c[i]->w; }
}
// PDF::value / PDF::generate for n items of n_comp PDFs each (n_comp > 1: MixturePdf), same argument meaning as rt_pdf_eval
// (desc = kind u32, cannot_refract u32, a[3], b[3], s: 64 B); `skip` draws of item i's stream are discarded first (tests only).
int orc_pdf_eval(uint32_t n, uint32_t n_comp, const void* descs, const double* dirs, uint64_t seed, uint32_t skip, double* value, double* generated) {
    struct Desc { uint32_t kind, cannot_refract; double a[3], b[3], s; };
    static_assert(sizeof(Desc) == 64, "rt_pdf_desc layout");
    const Desc* d = (const Desc*)descs;
    for (uint32_t i = 0; i < n; ++i) {
        MixturePdf mix;
        for (uint32_t c = 0; c < n_comp; ++c) {
            const Desc& e = d[(size_t)i * n_comp + c];
            Pdf p; const V3 a(e.a[0], e.a[1], e.a[2]);
            switch (e.kind) {
                case 0: p = Pdf::cosine(a); break;
                case 1: p = Pdf::uniform_hemi(a); break;
                case 2: p = Pdf::uniform_sphere(a, e.s); break;
                case 3: p.kind = Pdf::FUZZY_DELTA; p.dir = a; p.fuzz = e.s; break;   // direction as stored
                case 4: p.kind = Pdf::DIELECTRIC; p.reflect_dir = a; p.refract_dir = V3(e.b[0], e.b[1], e.b[2]); p.refl = e.s; p.cannot_refract = e.cannot_refract != 0; break;
                case 5: p = Pdf::cone(a, e.s); break;
                default: g_err = "orc_pdf_eval: unknown kind"; return -1;
            }
            mix.pdfs.push_back(p);
        }
        const V3 dir(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]);
        Rng rng(seed, i, 0, 0);
        for (uint32_t k = 0; k < skip; ++k) (void)rng.next_f64();
        value[i] = n_comp == 1 ? mix.pdfs[0].value(dir) : mix.value(dir);
        const V3 g = n_comp == 1 ? mix.pdfs[0].generate(rng) : mix.generate(rng);
        generated[3 * (size_t)i] = g.x; generated[3 * (size_t)i + 1] = g.y; generated[3 * (size_t)i + 2] = g.z;
    }
    return 0;
}
// f64 division / sqrt reference values for the device IEEE check (tests/test_gpu_ieee.py)
void orc_div_sqrt(uint32_t n, const double* a, const double* b, double* q, double* s) {
    for (uint32_t i = 0; i < n; ++i) { q[i] = a[i] / b[i]; s[i] = std::sqrt(a[i]); }
}

}  // extern "C"
