// Scene graph -> (a) "rtscene 1" text description, (b) "RTSB" flattened blob (include/rt_abi.h).
//
// Flattening turns the reference's pointer tree (bvh.rs:10-14, Arc<dyn Hittable> children) into a
// THREADED pre-order node array: the reference always recurses left then right (bvh.rs:96-103), so
// "where to go when a box test fails" is a single precomputed index (`skip`) and traversal needs
// no stack.  A leaf BVHNode (slice of 1 or 2 objects, bvh.rs:37-51) carries its primitives inline;
// the duplicate right child of a 1-object node is dropped (re-testing the same primitive with
// t_max = its own t returns the same record).  BVHNode *objects* (from_mesh under a Transformed,
// main.rs:161-188) become INSTANCE primitives whose sub-arrays follow the scene-level array.
#include "renderer.hpp"
#include <array>
#include <cinttypes>

namespace renderer {
namespace {

struct Tables {
    std::vector<const Texture*> textures;   std::unordered_map<const Texture*, uint32_t> tex_id;
    std::vector<const Material*> materials; std::unordered_map<const Material*, uint32_t> mat_id;
    std::vector<const Mesh*> meshes;        std::unordered_map<const Mesh*, uint32_t> mesh_id;
    std::vector<std::vector<std::array<uint32_t, 3>>> mesh_faces;
    std::vector<DMat4> xforms;              std::map<std::array<double, 16>, uint32_t> xform_id;
    std::vector<const Hittable*> objects;   std::unordered_map<const Hittable*, uint32_t> obj_id;
    std::unordered_map<const Triangle*, uint32_t> tri_face;
    uint64_t seed = 0; uint32_t mode = 0; bool have_seed = false;

    uint32_t texture(const Texture* t) {
        auto it = tex_id.find(t);
        if (it != tex_id.end()) return it->second;
        if (auto c = dynamic_cast<const CheckerTexture*>(t)) { texture(c->odd.get()); texture(c->even.get()); }
        else if (auto im = dynamic_cast<const Image*>(t)) {
            if (im->rgb8.size() != (size_t)im->size.first * im->size.second * 3)
                throw Unsupported("image texture without RGB8 source bytes (only load_texture images can be uploaded)");
        } else if (!dynamic_cast<const SolidColour*>(t)) throw Unsupported("unknown Texture implementation");
        uint32_t id = (uint32_t)textures.size();
        textures.push_back(t); tex_id[t] = id;
        return id;
    }
    uint32_t material(const Material* m) {
        auto it = mat_id.find(m);
        if (it != mat_id.end()) return it->second;
        if (auto l = dynamic_cast<const Lambertian*>(m)) texture(l->albedo.get());
        else if (auto d = dynamic_cast<const DiffuseLight*>(m)) texture(d->emit_colour.get());
        else if (!dynamic_cast<const Metal*>(m) && !dynamic_cast<const Dielectric*>(m)) throw Unsupported("unknown Material implementation");
        uint32_t id = (uint32_t)materials.size();
        materials.push_back(m); mat_id[m] = id;
        return id;
    }
    uint32_t xform(const DMat4& m) {
        std::array<double, 16> k = {m.x_axis.x, m.x_axis.y, m.x_axis.z, m.x_axis.w, m.y_axis.x, m.y_axis.y, m.y_axis.z, m.y_axis.w,
                                    m.z_axis.x, m.z_axis.y, m.z_axis.z, m.z_axis.w, m.w_axis.x, m.w_axis.y, m.w_axis.z, m.w_axis.w};
        auto it = xform_id.find(k);
        if (it != xform_id.end()) return it->second;
        uint32_t id = (uint32_t)xforms.size();
        xforms.push_back(m); xform_id[k] = id;
        return id;
    }
    void note_seed(uint64_t s, uint32_t m) {
        if (have_seed && (s != seed || m != mode)) throw Unsupported("BVHs of one scene built with different axis seeds or axis modes");
        seed = s; mode = m; have_seed = true;
    }
    // post-order: children get ids before their parents
    uint32_t object(const Hittable* h) {
        auto it = obj_id.find(h);
        if (it != obj_id.end()) return it->second;
        if (auto s = dynamic_cast<const Sphere*>(h)) material(s->material.get());
        else if (auto ms = dynamic_cast<const MovingSphere*>(h)) material(ms->material.get());
        else if (auto r = dynamic_cast<const AARect*>(h)) material(r->material.get());
        else if (auto t = dynamic_cast<const Triangle*>(h)) {
            const Mesh* me = t->data.get();
            auto mi = mesh_id.find(me);
            uint32_t m;
            if (mi == mesh_id.end()) { m = (uint32_t)meshes.size(); meshes.push_back(me); mesh_id[me] = m; mesh_faces.emplace_back(); material(me->material.get()); }
            else m = mi->second;
            tri_face[t] = (uint32_t)mesh_faces[m].size();
            mesh_faces[m].push_back({t->indices[0], t->indices[1], t->indices[2]});
        } else if (auto x = dynamic_cast<const Transformed*>(h)) { xform(x->t); object(x->hittable.get()); }
        else if (auto b = dynamic_cast<const BVHNode*>(h)) {
            if (b->source.empty()) throw Unsupported("BVHNode object that was not made by BVHNode::new_/from_mesh");
            note_seed(b->axis_seed, b->axis_mode);
            for (auto& c : b->source) object(c.get());
        } else throw Unsupported("unknown Hittable implementation");
        uint32_t id = (uint32_t)objects.size();
        objects.push_back(h); obj_id[h] = id;
        return id;
    }
    void collect(const Scene& sc) {
        for (auto& o : sc.objects) object(o.get());
        for (auto& l : sc.lights) object(l.get());
        if (sc.bvh) note_seed(sc.bvh->axis_seed, sc.bvh->axis_mode);
        if (!have_seed) { seed = bvh_axis_seed(); mode = bvh_axis_mode(); }
    }
};

void put_f(std::string& s, double d) { char b[40]; snprintf(b, sizeof b, " %.17g", d); s += b; }
void put_u(std::string& s, uint64_t u) { char b[32]; snprintf(b, sizeof b, " %" PRIu64, u); s += b; }
void put_v3(std::string& s, const DVec3& v) { put_f(s, v.x); put_f(s, v.y); put_f(s, v.z); }

}  // namespace

std::string Scene::describe() const {
    Tables tb; tb.collect(*this);
    std::string s = "rtscene 1\nbvh_seed";
    put_u(s, tb.seed); s += "\n";
    if (tb.mode != BVH_AXIS_SEEDED) { s += "bvh_axis_mode"; put_u(s, tb.mode); s += "\n"; }   // absent = the seeded stream
    for (const Texture* t : tb.textures) {
        if (auto sc = dynamic_cast<const SolidColour*>(t)) { s += "texture solid"; put_v3(s, sc->colour); }
        else if (auto c = dynamic_cast<const CheckerTexture*>(t)) { s += "texture checker"; put_u(s, tb.tex_id[c->odd.get()]); put_u(s, tb.tex_id[c->even.get()]); }
        else {
            auto im = static_cast<const Image*>(t);
            s += "texture image"; put_u(s, im->size.first); put_u(s, im->size.second); s += " ";
            static const char* hx = "0123456789abcdef";
            size_t at = s.size(); s.resize(at + im->rgb8.size() * 2);
            for (size_t i = 0; i < im->rgb8.size(); ++i) { s[at + 2 * i] = hx[im->rgb8[i] >> 4]; s[at + 2 * i + 1] = hx[im->rgb8[i] & 15]; }
        }
        s += "\n";
    }
    for (const Material* m : tb.materials) {
        if (auto l = dynamic_cast<const Lambertian*>(m)) { s += "material lambertian"; put_u(s, tb.tex_id[l->albedo.get()]); }
        else if (auto me = dynamic_cast<const Metal*>(m)) { s += "material metal"; put_v3(s, me->albedo()); put_f(s, me->fuzz()); }
        else if (auto d = dynamic_cast<const Dielectric*>(m)) { s += "material dielectric"; put_f(s, d->ior); }
        else { s += "material diffuse_light"; put_u(s, tb.tex_id[static_cast<const DiffuseLight*>(m)->emit_colour.get()]); }
        s += "\n";
    }
    for (size_t m = 0; m < tb.meshes.size(); ++m) {
        const Mesh* me = tb.meshes[m];
        s += "mesh"; put_u(s, me->vertices.size()); put_u(s, tb.mesh_faces[m].size()); put_u(s, tb.mat_id[me->material.get()]); s += "\n";
        for (size_t i = 0; i < me->vertices.size(); ++i) {
            put_v3(s, me->vertices[i]);
            put_v3(s, i < me->normals.size() ? me->normals[i] : DVec3());
            put_f(s, i < me->uv.size() ? me->uv[i].x : 0.0); put_f(s, i < me->uv.size() ? me->uv[i].y : 0.0);
            s += "\n";
        }
        for (auto& f : tb.mesh_faces[m]) { put_u(s, f[0]); put_u(s, f[1]); put_u(s, f[2]); s += "\n"; }
    }
    for (const DMat4& x : tb.xforms) {
        s += "xform";
        for (const DVec4* c : {&x.x_axis, &x.y_axis, &x.z_axis, &x.w_axis}) { put_f(s, c->x); put_f(s, c->y); put_f(s, c->z); put_f(s, c->w); }
        s += "\n";
    }
    for (const Hittable* h : tb.objects) {
        if (auto sp = dynamic_cast<const Sphere*>(h)) { s += "object sphere"; put_v3(s, sp->center); put_f(s, sp->radius); put_u(s, tb.mat_id[sp->material.get()]); }
        else if (auto ms = dynamic_cast<const MovingSphere*>(h)) {
            s += "object moving_sphere"; put_v3(s, ms->center_0); put_v3(s, ms->center_1); put_f(s, ms->time_0); put_f(s, ms->time_1); put_f(s, ms->radius);
            put_u(s, tb.mat_id[ms->material.get()]);
        } else if (auto r = dynamic_cast<const AARect*>(h)) {
            s += "object aarect"; put_f(s, r->x_range.first); put_f(s, r->x_range.second); put_f(s, r->y_range.first); put_f(s, r->y_range.second); put_f(s, r->z);
            put_u(s, tb.mat_id[r->material.get()]);
        } else if (auto t = dynamic_cast<const Triangle*>(h)) { s += "object triangle"; put_u(s, tb.mesh_id[t->data.get()]); put_u(s, tb.tri_face[t]); }
        else if (auto x = dynamic_cast<const Transformed*>(h)) { s += "object transformed"; put_u(s, tb.xform(x->t)); put_u(s, tb.obj_id[x->hittable.get()]); }
        else {
            auto b = static_cast<const BVHNode*>(h);
            s += "object bvh"; put_f(s, b->time_0); put_f(s, b->time_1); put_u(s, b->source.size());
            for (auto& c : b->source) put_u(s, tb.obj_id[c.get()]);
        }
        s += "\n";
    }
    s += "scene_objects"; put_u(s, objects.size());
    for (auto& o : objects) put_u(s, tb.obj_id[o.get()]);
    s += "\nlights"; put_u(s, lights.size());
    for (auto& l : lights) put_u(s, tb.obj_id[l.get()]);
    s += "\ncamera"; put_v3(s, camera.look_from); put_v3(s, camera.look_at); put_v3(s, camera.up);
    put_f(s, camera.vfov); put_f(s, camera.aspect_ratio); put_f(s, camera.aperture); put_f(s, camera.focus_dist); put_f(s, camera.time_0); put_f(s, camera.time_1);
    s += background == Background::Skybox ? "\nbackground sky" : "\nbackground black";
    s += bvh ? "\nbuild_bvh 1" : "\nbuild_bvh 0";
    s += "\nend\n";
    return s;
}

// ---------------------------------------------------------------------------------------------
namespace {

struct Flattener {
    Tables tb;
    std::vector<rt_node> nodes; std::vector<rt_prim> prims;
    std::vector<rt_geom> geom; std::vector<rt_tri_attrs> tri_attrs;
    uint32_t n_spheres = 0, n_moving = 0, n_rects = 0;
    std::vector<uint8_t> texels; std::vector<uint32_t> tex_offset;
    struct PendingInstance { const BVHNode* bvh; std::vector<uint32_t> prim_slots; };
    std::vector<PendingInstance> pending; std::unordered_map<const BVHNode*, size_t> pending_of;
    bool inside_instance = false;

    static void v3(double* d, const DVec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

    uint32_t make_prim(const Hittable* h, uint32_t xf, uint32_t inst, bool under_instance_xform) {
        rt_prim p; memset(&p, 0, sizeof p);
        p.xform = xf; p.inst_id = inst; p.material = RT_NONE; p.sub_end = 0;
        if (auto x = dynamic_cast<const Transformed*>(h)) {
            if (xf != RT_NONE) throw Unsupported("Transformed nested directly inside Transformed");
            if (under_instance_xform) throw Unsupported("Transformed primitive inside a transformed BVH instance");
            return make_prim(x->hittable.get(), tb.xform(x->t), tb.obj_id.at(h), under_instance_xform);
        }
        p.object_id = tb.obj_id.at(h);
        rt_geom g; memset(&g, 0, sizeof g);
        if (auto s = dynamic_cast<const Sphere*>(h)) {
            p.kind = RT_PRIM_SPHERE; p.material = tb.mat_id.at(s->material.get()); ++n_spheres;
            v3(g.d, s->center); g.d[3] = s->radius;
        } else if (auto ms = dynamic_cast<const MovingSphere*>(h)) {
            p.kind = RT_PRIM_MOVING_SPHERE; p.material = tb.mat_id.at(ms->material.get()); ++n_moving;
            v3(g.d, ms->center_0); v3(g.d + 3, ms->center_1); g.d[6] = ms->time_0; g.d[7] = ms->time_1; g.d[8] = ms->radius;
        } else if (auto a = dynamic_cast<const AARect*>(h)) {
            p.kind = RT_PRIM_AARECT; p.material = tb.mat_id.at(a->material.get()); ++n_rects;
            g.d[0] = a->x_range.first; g.d[1] = a->x_range.second; g.d[2] = a->y_range.first; g.d[3] = a->y_range.second; g.d[4] = a->z;
        } else if (auto t = dynamic_cast<const Triangle*>(h)) {
            const Mesh* me = t->data.get();
            p.kind = RT_PRIM_TRIANGLE; p.index = (uint32_t)tri_attrs.size(); p.material = tb.mat_id.at(me->material.get());
            rt_tri_attrs ta; memset(&ta, 0, sizeof ta);
            double* nn[3] = {ta.n0, ta.n1, ta.n2}; double* uu[3] = {ta.uv0, ta.uv1, ta.uv2};
            DVec3 vv[3];
            for (int k = 0; k < 3; ++k) {
                uint32_t i = t->indices[k];
                vv[k] = me->vertices.at(i);
                if (i < me->normals.size()) v3(nn[k], me->normals[i]);
                if (i < me->uv.size()) { uu[k][0] = me->uv[i].x; uu[k][1] = me->uv[i].y; }
            }
            v3(g.d, vv[0]); v3(g.d + 3, vv[1] - vv[0]); v3(g.d + 6, vv[2] - vv[0]);  // edge1, edge2 of mesh.rs:55-56
            tri_attrs.push_back(ta);
        } else if (auto b = dynamic_cast<const BVHNode*>(h)) {
            if (inside_instance) throw Unsupported("BVHNode object nested inside another BVHNode object");
            p.kind = RT_PRIM_INSTANCE;
            auto it = pending_of.find(b);
            size_t k;
            if (it == pending_of.end()) { k = pending.size(); pending.push_back({b, {}}); pending_of[b] = k; } else k = it->second;
            pending[k].prim_slots.push_back((uint32_t)prims.size());
        } else throw Unsupported("unknown Hittable implementation");
        g.kind = p.kind; g.xform = p.xform;
        geom.push_back(g);
        prims.push_back(p);
        return (uint32_t)prims.size() - 1;
    }

    void emit_bvh(const BVHNode* n, bool under_instance_xform) {
        uint32_t idx = (uint32_t)nodes.size();
        rt_node nd; memset(&nd, 0, sizeof nd);
        v3(nd.bmin, n->bbox.min); v3(nd.bmax, n->bbox.max);
        nd.prim0 = nd.prim1 = RT_NONE;
        nodes.push_back(nd);
        if (n->inner) {
            emit_bvh(static_cast<const BVHNode*>(n->left.get()), under_instance_xform);
            emit_bvh(static_cast<const BVHNode*>(n->right.get()), under_instance_xform);
        } else {
            uint32_t p0 = make_prim(n->left.get(), RT_NONE, RT_NONE, under_instance_xform);
            uint32_t p1 = n->right.get() == n->left.get() ? RT_NONE : make_prim(n->right.get(), RT_NONE, RT_NONE, under_instance_xform);
            nodes[idx].prim0 = p0; nodes[idx].prim1 = p1;
            if (n->right.get() == n->left.get()) nodes[idx].flags |= RT_NODE_DUP_RIGHT;
        }
        nodes[idx].skip = (uint32_t)nodes.size();
    }

    template <class T> static uint64_t append(std::vector<uint8_t>& out, const std::vector<T>& v) {
        while (out.size() % 16) out.push_back(0);
        uint64_t off = out.size();
        const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data());
        out.insert(out.end(), p, p + v.size() * sizeof(T));
        return off;
    }

    std::vector<uint8_t> run(const Scene& sc) {
        tb.collect(sc);
        // scene-level array
        uint32_t root = 0;
        if (sc.bvh) emit_bvh(&*sc.bvh, false);
        else
            for (auto& o : sc.objects) {  // hittable.rs:41-53: tested in list order, no boxes
                rt_node nd; memset(&nd, 0, sizeof nd);
                nd.flags = RT_NODE_NO_BOX; nd.prim1 = RT_NONE;
                uint32_t at = (uint32_t)nodes.size();
                nodes.push_back(nd);
                uint32_t p0 = make_prim(o.get(), RT_NONE, RT_NONE, false);
                nodes[at].prim0 = p0; nodes[at].skip = at + 1;
            }
        uint32_t root_end = (uint32_t)nodes.size();
        // sub-BVH arrays of INSTANCE primitives (each distinct BVHNode object once)
        inside_instance = true;
        for (size_t k = 0; k < pending.size(); ++k) {
            bool any_xf = false;
            for (uint32_t slot : pending[k].prim_slots) any_xf = any_xf || prims[slot].xform != RT_NONE;
            uint32_t first = (uint32_t)nodes.size();
            emit_bvh(pending[k].bvh, any_xf);
            uint32_t end = (uint32_t)nodes.size();
            for (uint32_t slot : pending[k].prim_slots) { prims[slot].index = first; prims[slot].sub_end = end; geom[slot].d[0] = (double)first; geom[slot].d[1] = (double)end; }
        }
        // materials / textures / lights
        std::vector<rt_texture> tex(tb.textures.size());
        for (size_t i = 0; i < tb.textures.size(); ++i) {
            rt_texture t; memset(&t, 0, sizeof t);
            if (auto s = dynamic_cast<const SolidColour*>(tb.textures[i])) { t.kind = RT_TEX_SOLID; v3(t.colour, s->colour); }
            else if (auto c = dynamic_cast<const CheckerTexture*>(tb.textures[i])) { t.kind = RT_TEX_CHECKER; t.a = tb.tex_id.at(c->odd.get()); t.b = tb.tex_id.at(c->even.get()); }
            else {
                auto im = static_cast<const Image*>(tb.textures[i]);
                t.kind = RT_TEX_IMAGE; t.a = (uint32_t)texels.size(); t.b = im->size.first; t.c = im->size.second;
                texels.insert(texels.end(), im->rgb8.begin(), im->rgb8.end());
            }
            tex[i] = t;
        }
        std::vector<rt_material> mats(tb.materials.size());
        for (size_t i = 0; i < tb.materials.size(); ++i) {
            rt_material m; memset(&m, 0, sizeof m); m.texture = RT_NONE;
            if (auto l = dynamic_cast<const Lambertian*>(tb.materials[i])) { m.kind = RT_MAT_LAMBERTIAN; m.texture = tb.tex_id.at(l->albedo.get()); }
            else if (auto me = dynamic_cast<const Metal*>(tb.materials[i])) { m.kind = RT_MAT_METAL; v3(m.albedo, me->albedo()); m.fuzz = me->fuzz(); }
            else if (auto d = dynamic_cast<const Dielectric*>(tb.materials[i])) { m.kind = RT_MAT_DIELECTRIC; m.ior = d->ior; }
            else { m.kind = RT_MAT_DIFFUSE_LIGHT; m.texture = tb.tex_id.at(static_cast<const DiffuseLight*>(tb.materials[i])->emit_colour.get()); }
            mats[i] = m;
        }
        std::vector<rt_light> lights(sc.lights.size());
        for (size_t i = 0; i < sc.lights.size(); ++i) {
            rt_light l; memset(&l, 0, sizeof l);
            v3(l.center, sc.lights[i]->center); l.radius = sc.lights[i]->radius;
            l.material = tb.mat_id.at(sc.lights[i]->material.get()); l.object_id = tb.obj_id.at(sc.lights[i].get());
            lights[i] = l;
        }
        std::vector<rt_xform> xf(tb.xforms.size());
        for (size_t i = 0; i < tb.xforms.size(); ++i) {
            DMat4 inv = tb.xforms[i].inverse();  // transform.rs:17
            const DVec4* c[4] = {&inv.x_axis, &inv.y_axis, &inv.z_axis, &inv.w_axis};
            for (int k = 0; k < 4; ++k) { xf[i].t_inv[4 * k] = c[k]->x; xf[i].t_inv[4 * k + 1] = c[k]->y; xf[i].t_inv[4 * k + 2] = c[k]->z; xf[i].t_inv[4 * k + 3] = c[k]->w; }
        }
        rt_blob_header h; memset(&h, 0, sizeof h);
        h.magic = RT_BLOB_MAGIC; h.version = RT_BLOB_VERSION;
        h.n_nodes = (uint32_t)nodes.size(); h.n_prims = (uint32_t)prims.size(); h.n_tris = (uint32_t)tri_attrs.size();
        h.n_spheres = n_spheres; h.n_moving_spheres = n_moving; h.n_rects = n_rects;
        h.n_xforms = (uint32_t)xf.size(); h.n_materials = (uint32_t)mats.size(); h.n_textures = (uint32_t)tex.size(); h.n_lights = (uint32_t)lights.size();
        h.n_texel_bytes = (uint32_t)texels.size();
        h.root_node = root; h.root_end = root_end; h.background = (uint32_t)sc.background; h.bvh_seed = tb.seed;
        const Camera& c = sc.camera;
        v3(h.camera.origin, c.origin); v3(h.camera.lower_left_corner, c.lower_left_corner); v3(h.camera.horizontal, c.horizontal);
        v3(h.camera.vertical, c.vertical); v3(h.camera.u, c.u); v3(h.camera.v, c.v);
        h.camera.lens_radius = c.lens_radius; h.camera.time0 = c.time_0; h.camera.time1 = c.time_1;
        std::vector<uint8_t> out(sizeof h, 0);
        h.off_nodes = append(out, nodes); h.off_prims = append(out, prims); h.off_geom = append(out, geom); h.off_tri_attrs = append(out, tri_attrs);
        h.off_xforms = append(out, xf);
        h.off_materials = append(out, mats); h.off_textures = append(out, tex); h.off_texels = append(out, texels); h.off_lights = append(out, lights);
        while (out.size() % 16) out.push_back(0);
        h.total_bytes = out.size();
        memcpy(out.data(), &h, sizeof h);
        return out;
    }
};

}  // namespace

std::vector<uint8_t> Scene::flatten() const { Flattener f; return f.run(*this); }

// ---------------------------------------------------------------------------------------------
namespace {
struct Cursor {
    const char* p; const char* e;
    bool tok(std::string& o) {
        while (p < e && isspace((unsigned char)*p)) ++p;
        if (p >= e) return false;
        const char* b = p;
        while (p < e && !isspace((unsigned char)*p)) ++p;
        o.assign(b, p);
        return true;
    }
    std::string w() { std::string s; if (!tok(s)) throw std::runtime_error("rtscene: truncated"); return s; }
    double f() { std::string s = w(); char* q; double d = strtod(s.c_str(), &q); if (*q) throw std::runtime_error("rtscene: number expected, got " + s); return d; }
    uint64_t u() { std::string s = w(); char* q; uint64_t d = strtoull(s.c_str(), &q, 10); if (*q) throw std::runtime_error("rtscene: integer expected, got " + s); return d; }
    DVec3 v() { double a = f(), b = f(), c = f(); return DVec3(a, b, c); }
};
}  // namespace

Scene scene_from_description(const char* text, size_t len) {
    // the description's bvh_seed / bvh_axis_mode records hold for THIS scene only: whatever the caller had set for its own builds
    // (rth_set_bvh_axis_seed / _mode) is back in place on every way out, so a later rth_scene_builtin builds the tree it asked for
    struct RestoreAxes {
        uint64_t seed = bvh_axis_seed(); uint32_t mode = bvh_axis_mode();
        ~RestoreAxes() { bvh_axis_seed() = seed; bvh_axis_mode() = mode; }
    } restore_axes;
    Cursor c{text, text + len};
    if (c.w() != "rtscene" || c.u() != 1) throw std::runtime_error("rtscene: bad header");
    std::vector<Arc<Texture>> tex; std::vector<Arc<Material>> mat; std::vector<Arc<Mesh>> mesh; std::vector<std::vector<std::array<uint32_t, 3>>> faces;
    std::vector<DMat4> xf; HittableList obj;
    Scene sc; bool want_bvh = false;
    auto at = [](auto& v, uint64_t i, const char* what) -> decltype(v[0])& { if (i >= v.size()) throw std::runtime_error(std::string("rtscene: bad ") + what + " id"); return v[i]; };
    std::string k;
    while (c.tok(k)) {
        if (k == "end") break;
        if (k == "bvh_seed") { bvh_axis_seed() = c.u(); bvh_axis_mode() = BVH_AXIS_SEEDED; }   // a bvh_axis_mode record follows when it is not the seeded stream
        else if (k == "bvh_axis_mode") { uint64_t m = c.u(); if (m > BVH_AXIS_MIN_AREA) throw std::runtime_error("rtscene: unknown bvh_axis_mode"); bvh_axis_mode() = (uint32_t)m; }
        else if (k == "texture") {
            std::string t = c.w();
            if (t == "solid") tex.push_back(std::make_shared<SolidColour>(c.v()));
            else if (t == "checker") { uint64_t o = c.u(), e = c.u(); tex.push_back(std::make_shared<CheckerTexture>(at(tex, o, "texture"), at(tex, e, "texture"))); }
            else if (t == "image") {
                uint32_t w = (uint32_t)c.u(), h = (uint32_t)c.u(); std::string hex = c.w();
                if (hex.size() != (size_t)w * h * 6) throw std::runtime_error("rtscene: image payload size");
                auto im = std::make_shared<Image>(Image::new_({w, h}));
                im->rgb8.resize((size_t)w * h * 3);
                auto hv = [](char ch) { return ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10; };
                for (size_t i = 0; i < im->rgb8.size(); ++i) im->rgb8[i] = (uint8_t)(hv(hex[2 * i]) * 16 + hv(hex[2 * i + 1]));
                for (size_t i = 0; i < (size_t)w * h; ++i) {  // main.rs:897-904
                    double r = im->rgb8[3 * i] / 255.0, g = im->rgb8[3 * i + 1] / 255.0, b = im->rgb8[3 * i + 2] / 255.0;
                    im->data[i] = DVec3(r * r, g * g, b * b);
                }
                tex.push_back(im);
            } else throw std::runtime_error("rtscene: texture kind " + t);
        } else if (k == "material") {
            std::string t = c.w();
            if (t == "lambertian") mat.push_back(std::make_shared<Lambertian>(at(tex, c.u(), "texture")));
            else if (t == "metal") { DVec3 a = c.v(); double fz = c.f(); mat.push_back(std::make_shared<Metal>(Metal::new_(a, fz))); }
            else if (t == "dielectric") mat.push_back(std::make_shared<Dielectric>(c.f()));
            else if (t == "diffuse_light") mat.push_back(std::make_shared<DiffuseLight>(at(tex, c.u(), "texture")));
            else throw std::runtime_error("rtscene: material kind " + t);
        } else if (k == "mesh") {
            uint64_t nv = c.u(), nt = c.u();
            auto me = std::make_shared<Mesh>();
            me->material = at(mat, c.u(), "material");
            me->vertices.resize(nv); me->normals.resize(nv); me->uv.resize(nv);
            for (uint64_t i = 0; i < nv; ++i) { me->vertices[i] = c.v(); me->normals[i] = c.v(); double a = c.f(), b = c.f(); me->uv[i] = DVec2(a, b); }
            std::vector<std::array<uint32_t, 3>> fc(nt);
            for (auto& f : fc) for (int j = 0; j < 3; ++j) { f[j] = (uint32_t)c.u(); if (f[j] >= nv) throw std::runtime_error("rtscene: bad vertex index"); }
            mesh.push_back(me); faces.push_back(std::move(fc));
        } else if (k == "xform") {
            double m[16]; for (double& d : m) d = c.f();
            xf.push_back(DMat4::from_cols(DVec4(m[0], m[1], m[2], m[3]), DVec4(m[4], m[5], m[6], m[7]), DVec4(m[8], m[9], m[10], m[11]), DVec4(m[12], m[13], m[14], m[15])));
        } else if (k == "object") {
            std::string t = c.w();
            if (t == "sphere") { DVec3 ce = c.v(); double r = c.f(); obj.push_back(std::make_shared<Sphere>(ce, r, at(mat, c.u(), "material"))); }
            else if (t == "moving_sphere") { DVec3 c0 = c.v(), c1 = c.v(); double t0 = c.f(), t1 = c.f(), r = c.f(); obj.push_back(std::make_shared<MovingSphere>(c0, c1, t0, t1, r, at(mat, c.u(), "material"))); }
            else if (t == "aarect") { double x0 = c.f(), x1 = c.f(), y0 = c.f(), y1 = c.f(), z = c.f(); obj.push_back(std::make_shared<AARect>(std::make_pair(x0, x1), std::make_pair(y0, y1), z, at(mat, c.u(), "material"))); }
            else if (t == "triangle") {
                uint64_t mi = c.u(), ti = c.u();
                auto& me = at(mesh, mi, "mesh"); auto& fc = at(faces[mi], ti, "face");
                auto tr = std::make_shared<Triangle>(); tr->data = me; tr->indices[0] = fc[0]; tr->indices[1] = fc[1]; tr->indices[2] = fc[2];
                obj.push_back(tr);
            } else if (t == "transformed") { uint64_t xi = c.u(), ci = c.u(); obj.push_back(std::make_shared<Transformed>(Transformed::new_(at(xf, xi, "xform"), at(obj, ci, "object")))); }
            else if (t == "bvh") {
                double t0 = c.f(), t1 = c.f(); uint64_t n = c.u(); HittableList kids;
                for (uint64_t i = 0; i < n; ++i) kids.push_back(at(obj, c.u(), "object"));
                obj.push_back(std::make_shared<BVHNode>(BVHNode::new_(kids, t0, t1)));
            } else throw std::runtime_error("rtscene: object kind " + t);
        } else if (k == "scene_objects") { uint64_t n = c.u(); for (uint64_t i = 0; i < n; ++i) sc.objects.push_back(at(obj, c.u(), "object")); }
        else if (k == "lights") {
            uint64_t n = c.u();
            for (uint64_t i = 0; i < n; ++i) {
                auto s = std::dynamic_pointer_cast<Sphere>(at(obj, c.u(), "object"));
                if (!s) throw std::runtime_error("rtscene: light must be a sphere (shape.rs:36: only Sphere is a SampleableLight)");
                sc.lights.push_back(s);
            }
        } else if (k == "camera") {
            DVec3 from = c.v(), to = c.v(), up = c.v(); double vfov = c.f(), asp = c.f(), ap = c.f(), fo = c.f(), t0 = c.f(), t1 = c.f();
            sc.camera = Camera::new_(from, to, up, vfov, asp, ap, fo, t0, t1);
        } else if (k == "background") sc.background = c.w() == "sky" ? Background::Skybox : Background::NoLight;
        else if (k == "build_bvh") want_bvh = c.u() != 0;
        else throw std::runtime_error("rtscene: unknown record " + k);
    }
    if (want_bvh) sc.bvh = BVHNode::new_(sc.objects, 0.0, 0.0);
    return sc;
}

}  // namespace renderer
