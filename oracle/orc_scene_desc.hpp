// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// Reader for the neutral "rtscene 1" text description (the INPUT shared by the oracle and the
// product: a list of textures, materials, meshes, transforms and objects exactly as a reference
// scene function of desktop/src/main.rs would construct them, BEFORE any BVH is built).  Grammar
// (whitespace-separated tokens, ids are 0-based in order of appearance per table):
//   rtscene 1
//   texture solid R G B | texture checker ODD_TEX EVEN_TEX | texture image W H HEX(W*H*3 bytes)
//   material lambertian TEX | material metal R G B FUZZ | material dielectric IOR | material diffuse_light TEX
//   mesh NV NT MAT  then NV x "X Y Z NX NY NZ U V"  then NT x "I J K"
//   xform 16 doubles (column-major)
//   object sphere CX CY CZ R MAT | object moving_sphere C0x3 C1x3 T0 T1 R MAT | object aarect X0 X1 Y0 Y1 Z MAT
//   object triangle MESH TRI | object transformed XFORM CHILD | object bvh T0 T1 N CHILD...
//   scene_objects N ID... | lights N ID... | camera FROMx3 ATx3 UPx3 VFOV ASPECT APERTURE FOCUS T0 T1
//   background black|sky | build_bvh 0|1 | bvh_seed S | bvh_axis_mode 0|1 (after bvh_seed, before the first bvh object) | end
#pragma once
#include "orc_render.hpp"
#include <cstdlib>

namespace orc {

struct Tok {
    const char* p; const char* end;
    Tok(const char* s, size_t n) : p(s), end(s + n) {}
    bool next(std::string& out) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        if (p >= end) return false;
        const char* b = p;
        while (p < end && !(*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
        out.assign(b, p);
        return true;
    }
    std::string word() { std::string s; if (!next(s)) throw std::runtime_error("rtscene: unexpected end"); return s; }
    double f() { std::string s = word(); char* e; double d = std::strtod(s.c_str(), &e); if (*e) throw std::runtime_error("rtscene: bad number " + s); return d; }
    uint64_t u() { std::string s = word(); char* e; unsigned long long d = std::strtoull(s.c_str(), &e, 10); if (*e) throw std::runtime_error("rtscene: bad integer " + s); return d; }
    V3 v3() { double a = f(), b = f(), c = f(); return V3(a, b, c); }
};

inline int hexval(char c) { return c <= '9' ? c - '0' : (c <= 'F' ? c - 'A' + 10 : c - 'a' + 10); }

inline std::unique_ptr<Scene> load_scene_desc(const char* text, size_t len) {
    Tok tk(text, len);
    auto sc = std::make_unique<Scene>();
    if (tk.word() != "rtscene" || tk.u() != 1) throw std::runtime_error("rtscene: bad header");
    struct TriRef { uint32_t mesh, tri; };
    std::vector<std::vector<uint32_t>> mesh_indices;
    bool build_bvh = false;
    struct PendingBvh { BVHNode* node; std::vector<uint32_t> kids; double t0, t1; uint32_t id; };
    std::string w;
    auto mat = [&](uint64_t i) -> const Material* { if (i >= sc->materials.size()) throw std::runtime_error("rtscene: bad material id"); return sc->materials[i].get(); };
    auto obj = [&](uint64_t i) -> Hittable* { if (i >= sc->pool.size()) throw std::runtime_error("rtscene: bad object id"); return sc->pool[i].get(); };
    while (tk.next(w)) {
        if (w == "end") break;
        if (w == "texture") {
            auto t = std::make_unique<Texture>();
            std::string k = tk.word();
            if (k == "solid") { t->kind = Texture::SOLID; t->colour = tk.v3(); }
            else if (k == "checker") {
                t->kind = Texture::CHECKER;
                uint64_t o = tk.u(), e = tk.u();
                if (o >= sc->textures.size() || e >= sc->textures.size()) throw std::runtime_error("rtscene: bad texture id");
                t->odd = sc->textures[o].get(); t->even = sc->textures[e].get();
            } else if (k == "image") {
                uint32_t iw = (uint32_t)tk.u(), ih = (uint32_t)tk.u();
                std::string hex = tk.word();
                if (hex.size() != (size_t)iw * ih * 6) throw std::runtime_error("rtscene: image payload size");
                std::vector<uint8_t> rgb((size_t)iw * ih * 3);
                for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)(hexval(hex[2 * i]) * 16 + hexval(hex[2 * i + 1]));
                t->load_rgb8(iw, ih, rgb.data());
            } else throw std::runtime_error("rtscene: texture kind " + k);
            sc->textures.push_back(std::move(t));
        } else if (w == "material") {
            auto m = std::make_unique<Material>();
            std::string k = tk.word();
            auto tex = [&](uint64_t i) -> const Texture* { if (i >= sc->textures.size()) throw std::runtime_error("rtscene: bad texture id"); return sc->textures[i].get(); };
            if (k == "lambertian") { m->kind = Material::LAMBERTIAN; m->tex = tex(tk.u()); }
            else if (k == "metal") { m->kind = Material::METAL; m->albedo = tk.v3(); double fz = tk.f(); m->fuzz = fz > 1.0 ? 1.0 : fz; }  // material.rs:56-61
            else if (k == "dielectric") { m->kind = Material::DIELECTRIC; m->ior = tk.f(); }
            else if (k == "diffuse_light") { m->kind = Material::DIFFUSE_LIGHT; m->tex = tex(tk.u()); }
            else throw std::runtime_error("rtscene: material kind " + k);
            sc->materials.push_back(std::move(m));
        } else if (w == "mesh") {
            auto me = std::make_unique<Mesh>();
            uint64_t nv = tk.u(), nt = tk.u();
            me->material = mat(tk.u());
            me->vertices.resize(nv); me->normals.resize(nv); me->uv.resize(2 * nv);
            for (uint64_t i = 0; i < nv; ++i) { me->vertices[i] = tk.v3(); me->normals[i] = tk.v3(); me->uv[2 * i] = tk.f(); me->uv[2 * i + 1] = tk.f(); }
            std::vector<uint32_t> idx(3 * nt);
            for (uint64_t i = 0; i < 3 * nt; ++i) { idx[i] = (uint32_t)tk.u(); if (idx[i] >= nv) throw std::runtime_error("rtscene: bad vertex index"); }
            mesh_indices.push_back(std::move(idx));
            sc->meshes.push_back(std::move(me));
        } else if (w == "xform") {
            double m[16];
            for (int i = 0; i < 16; ++i) m[i] = tk.f();
            M4 x; x.c0 = V4(m[0], m[1], m[2], m[3]); x.c1 = V4(m[4], m[5], m[6], m[7]); x.c2 = V4(m[8], m[9], m[10], m[11]); x.c3 = V4(m[12], m[13], m[14], m[15]);
            sc->xforms.push_back(x);
        } else if (w == "object") {
            std::string k = tk.word();
            uint32_t id = (uint32_t)sc->pool.size();
            std::unique_ptr<Hittable> h;
            if (k == "sphere") { auto s = std::make_unique<Sphere>(); s->center = tk.v3(); s->radius = tk.f(); s->material = mat(tk.u()); h = std::move(s); }
            else if (k == "moving_sphere") {
                auto s = std::make_unique<MovingSphere>(); s->center_0 = tk.v3(); s->center_1 = tk.v3(); s->time_0 = tk.f(); s->time_1 = tk.f();
                s->radius = tk.f(); s->material = mat(tk.u()); h = std::move(s);
            } else if (k == "aarect") {
                auto r = std::make_unique<AARect>(); r->x0 = tk.f(); r->x1 = tk.f(); r->y0 = tk.f(); r->y1 = tk.f(); r->z = tk.f(); r->material = mat(tk.u()); h = std::move(r);
            } else if (k == "triangle") {
                auto t = std::make_unique<Triangle>();
                uint64_t mi = tk.u(), ti = tk.u();
                if (mi >= sc->meshes.size() || 3 * ti + 2 >= mesh_indices[mi].size()) throw std::runtime_error("rtscene: bad triangle ref");
                t->data = sc->meshes[mi].get();
                for (int j = 0; j < 3; ++j) t->idx[j] = mesh_indices[mi][3 * ti + j];
                h = std::move(t);
            } else if (k == "transformed") {
                auto t = std::make_unique<Transformed>();
                uint64_t xi = tk.u(), ci = tk.u();
                if (xi >= sc->xforms.size()) throw std::runtime_error("rtscene: bad xform id");
                t->set(sc->xforms[xi], obj(ci));
                h = std::move(t);
            } else if (k == "bvh") {  // BVHNode::new / from_mesh (bvh.rs:27-87), built immediately like the reference
                auto b = std::make_unique<BVHNode>();
                double t0 = tk.f(), t1 = tk.f();
                uint64_t n = tk.u();
                std::vector<Hittable*> kids(n);
                for (uint64_t i = 0; i < n; ++i) kids[i] = obj(tk.u());
                AxisRng ar(sc->bvh_seed, sc->bvh_axis_mode);
                b->build(kids.data(), kids.size(), t0, t1, ar);
                h = std::move(b);
            } else throw std::runtime_error("rtscene: object kind " + k);
            h->id = id;
            sc->pool.push_back(std::move(h));
        } else if (w == "scene_objects") {
            uint64_t n = tk.u();
            for (uint64_t i = 0; i < n; ++i) sc->objects.push_back(obj(tk.u()));
        } else if (w == "lights") {
            uint64_t n = tk.u();
            for (uint64_t i = 0; i < n; ++i) {
                const Sphere* s = dynamic_cast<const Sphere*>(obj(tk.u()));
                if (!s) throw std::runtime_error("rtscene: light must be a sphere (only Sphere implements SampleableLight, shape.rs:36)");
                sc->lights.push_back(s);
            }
        } else if (w == "camera") {
            V3 from = tk.v3(), at = tk.v3(), up = tk.v3();
            double vfov = tk.f(), aspect = tk.f(), aperture = tk.f(), focus = tk.f(), t0 = tk.f(), t1 = tk.f();
            sc->camera.set(from, at, up, vfov, aspect, aperture, focus, t0, t1);
        } else if (w == "background") {
            std::string k = tk.word();
            sc->background = (k == "sky") ? BG_SKY : BG_BLACK;
        } else if (w == "build_bvh") { build_bvh = tk.u() != 0; }
        else if (w == "bvh_seed") { sc->bvh_seed = tk.u(); }
        else if (w == "bvh_axis_mode") { uint64_t m = tk.u(); if (m > 1) throw std::runtime_error("rtscene: unknown bvh_axis_mode"); sc->bvh_axis_mode = (uint32_t)m; }
        else throw std::runtime_error("rtscene: unknown record " + w);
    }
    if (build_bvh) {  // scene.rs:106-113
        sc->bvh = std::make_unique<BVHNode>();
        AxisRng ar(sc->bvh_seed, sc->bvh_axis_mode);
        sc->bvh->build(sc->objects.data(), sc->objects.size(), 0.0, 0.0, ar);
    }
    return sc;
}

}  // namespace orc
