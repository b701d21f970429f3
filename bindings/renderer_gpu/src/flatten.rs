//! Scene graph of the `renderer` crate -> "RTSB" blob (include/rt_abi.h), the Rust counterpart of
//! raytracing_amd/host/flatten.cpp.
//!
//! The reference's objects are opaque trait objects (`Arc<dyn Hittable>`, `Arc<dyn Material>`, `Arc<dyn Texture>`) and some
//! of their fields are private, so flattening needs ONE additive hook per trait, implemented next to each type (where its
//! private fields are visible).  Nothing else in the crate changes; scene-building code stays as it is.
//!
//!   renderer/src/hittable.rs:7-20   trait Hittable  { ...; fn gpu_describe(&self) -> GpuShape<'_> { GpuShape::Unsupported } }
//!   renderer/src/material.rs:9-16   trait Material  { ...; fn gpu_describe(&self) -> GpuMaterial<'_>; }
//!   renderer/src/texture.rs:5-7     trait Texture   { ...; fn gpu_describe(&self) -> GpuTexture<'_>; }
//!   renderer/src/shape.rs           impl for Sphere (17-21), MovingSphere (202-211), AARect (258-263)
//!   renderer/src/mesh.rs:42-45      impl for Triangle          renderer/src/transform.rs:7-11   impl for Transformed
//!   renderer/src/bvh.rs:10-14       impl for BVHNode (children + whether `left` and `right` are the same Arc, bvh.rs:37-43)
//!   renderer/src/scene.rs:11-17     pub fn gpu_parts(&self) -> (&[Arc<dyn Hittable>], Option<&BVHNode>)   (objects and bvh are private)
//!   renderer/src/camera.rs:9-19     pub fn gpu_parts(&self) -> [DVec3; 6] + (lens_radius, time_0, time_1) (all fields are private)
//!
//! `background: fn(Ray) -> DVec3` (scene.rs:15) cannot be introspected: `classify_background` probes it against the two
//! functions the reference defines (desktop/src/main.rs:37-48).
use crate::abi::*;
use glam::{DMat4, DVec2, DVec3};
use std::collections::HashMap;
use std::sync::Arc;

/// What a `Hittable` is, as far as the GPU path needs to know.  Returned by the additive `Hittable::gpu_describe`.
pub enum GpuShape<'a> {
    Sphere { center: DVec3, radius: f64, material: &'a Arc<dyn renderer::Material> },
    MovingSphere { center_0: DVec3, center_1: DVec3, time_0: f64, time_1: f64, radius: f64, material: &'a Arc<dyn renderer::Material> },
    AARect { x_range: (f64, f64), y_range: (f64, f64), z: f64, material: &'a Arc<dyn renderer::Material> },
    Triangle { indices: [u32; 3], mesh: &'a Arc<renderer::Mesh> },
    Transformed { t: DMat4, child: &'a Arc<dyn renderer::Hittable> },
    /// `same_child`: `left` and `right` are one object (the 1-element slice of bvh.rs:37-43); `bbox` = the node's own box
    Bvh { left: &'a Arc<dyn renderer::Hittable>, right: &'a Arc<dyn renderer::Hittable>, same_child: bool, bmin: DVec3, bmax: DVec3 },
    Unsupported,
}
pub enum GpuMaterial<'a> {
    Lambertian(&'a Arc<dyn renderer::Texture>),
    Metal { albedo: DVec3, fuzz: f64 },
    Dielectric { ior: f64 },
    DiffuseLight(&'a Arc<dyn renderer::Texture>),
}
pub enum GpuTexture<'a> {
    Solid(DVec3),
    Checker { odd: &'a Arc<dyn renderer::Texture>, even: &'a Arc<dyn renderer::Texture> },
    /// texels as the reference stores them: (byte / 255)^2 per channel (desktop/src/main.rs:900-903), row-major
    Image { size: (u32, u32), data: &'a [DVec3] },
}

#[derive(Debug)]
pub enum FlattenError {
    Unsupported(&'static str),
    BackgroundNotRecognised,
}

fn v3(d: &mut [f64], v: DVec3) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
fn key<T: ?Sized>(a: &Arc<T>) -> usize {
    Arc::as_ptr(a) as *const u8 as usize
}

/// `no_light` / `skybox` of desktop/src/main.rs:37-48 by their values on three probe rays.
pub fn classify_background(f: fn(renderer::Ray) -> DVec3) -> Result<u32, FlattenError> {
    let probe = |d: DVec3| f(renderer::Ray { origin: DVec3::ZERO, dir: d, time: 0.0 });
    let dirs = [DVec3::new(0.0, 1.0, 0.0), DVec3::new(1.0, -2.0, 0.5), DVec3::new(-0.3, 0.2, 1.0)];
    if dirs.iter().all(|d| probe(*d) == DVec3::ZERO) {
        return Ok(RT_BG_BLACK);
    }
    let sky = |d: DVec3| {
        let t = 0.5 * d.normalize().y + 1.0; // the reference's own formula (not 0.5 * (y + 1))
        DVec3::ONE * (1.0 - t) + DVec3::new(0.5, 0.7, 1.0) * t
    };
    if dirs.iter().all(|d| probe(*d) == sky(*d)) {
        return Ok(RT_BG_SKY);
    }
    Err(FlattenError::BackgroundNotRecognised)
}

#[derive(Default)]
pub struct Flattener {
    nodes: Vec<rt_node>,
    prims: Vec<rt_prim>,
    geom: Vec<rt_geom>,
    tri_attrs: Vec<rt_tri_attrs>,
    xforms: Vec<DMat4>,
    materials: Vec<rt_material>,
    textures: Vec<rt_texture>,
    texels: Vec<u8>,
    mat_id: HashMap<usize, u32>,
    tex_id: HashMap<usize, u32>,
    obj_id: HashMap<usize, u32>,
    next_obj: u32,
    n_spheres: u32,
    n_moving: u32,
    n_rects: u32,
    /// BVHNode objects met as primitives (main.rs:161-188): (root, primitive slots that instance it)
    pending: Vec<(Arc<dyn renderer::Hittable>, Vec<u32>)>,
    inside_instance: bool,
}

impl Flattener {
    fn texture(&mut self, t: &Arc<dyn renderer::Texture>) -> Result<u32, FlattenError> {
        if let Some(id) = self.tex_id.get(&key(t)) {
            return Ok(*id);
        }
        let mut r = rt_texture::default();
        match t.gpu_describe() {
            GpuTexture::Solid(c) => {
                r.kind = RT_TEX_SOLID;
                v3(&mut r.colour, c);
            }
            GpuTexture::Checker { odd, even } => {
                r.kind = RT_TEX_CHECKER;
                r.a = self.texture(odd)?;
                r.b = self.texture(even)?;
            }
            GpuTexture::Image { size, data } => {
                // the blob keeps RGB8 and the device evaluates (byte / 255)^2 in f64 per fetch: bit-identical to the table
                // load_texture builds, at 1/8 of its footprint.  sqrt(v) * 255 recovers the byte exactly for v = (b / 255)^2.
                r.kind = RT_TEX_IMAGE;
                r.a = self.texels.len() as u32;
                r.b = size.0;
                r.c = size.1;
                for px in data {
                    for c in [px.x, px.y, px.z] {
                        self.texels.push((c.sqrt() * 255.0).round() as u8);
                    }
                }
            }
        }
        let id = self.textures.len() as u32;
        self.textures.push(r);
        self.tex_id.insert(key(t), id);
        Ok(id)
    }

    fn material(&mut self, m: &Arc<dyn renderer::Material>) -> Result<u32, FlattenError> {
        if let Some(id) = self.mat_id.get(&key(m)) {
            return Ok(*id);
        }
        let mut r = rt_material::default();
        r.texture = RT_NONE;
        match m.gpu_describe() {
            GpuMaterial::Lambertian(t) => {
                r.kind = RT_MAT_LAMBERTIAN;
                r.texture = self.texture(t)?;
            }
            GpuMaterial::Metal { albedo, fuzz } => {
                r.kind = RT_MAT_METAL;
                v3(&mut r.albedo, albedo);
                r.fuzz = fuzz; // already clamped to <= 1 by Metal::new (material.rs:55-62)
            }
            GpuMaterial::Dielectric { ior } => {
                r.kind = RT_MAT_DIELECTRIC;
                r.ior = ior;
            }
            GpuMaterial::DiffuseLight(t) => {
                r.kind = RT_MAT_DIFFUSE_LIGHT;
                r.texture = self.texture(t)?;
            }
        }
        let id = self.materials.len() as u32;
        self.materials.push(r);
        self.mat_id.insert(key(m), id);
        Ok(id)
    }

    fn object_id(&mut self, h: &Arc<dyn renderer::Hittable>) -> u32 {
        let k = key(h);
        if let Some(id) = self.obj_id.get(&k) {
            return *id;
        }
        let id = self.next_obj;
        self.next_obj += 1;
        self.obj_id.insert(k, id);
        id
    }

    fn xform(&mut self, m: DMat4) -> u32 {
        if let Some(i) = self.xforms.iter().position(|x| x.to_cols_array() == m.to_cols_array()) {
            return i as u32;
        }
        self.xforms.push(m);
        (self.xforms.len() - 1) as u32
    }

    /// One primitive slot (rt_prim + the parallel rt_geom) for `h`; `xf` / `inst` come from an enclosing Transformed.
    fn make_prim(&mut self, h: &Arc<dyn renderer::Hittable>, xf: u32, inst: u32, under_instance_xform: bool) -> Result<u32, FlattenError> {
        let mut p = rt_prim::default();
        p.xform = xf;
        p.inst_id = inst;
        p.material = RT_NONE;
        let mut g = rt_geom::default();
        match h.gpu_describe() {
            GpuShape::Transformed { t, child } => {
                if xf != RT_NONE {
                    return Err(FlattenError::Unsupported("Transformed nested directly inside Transformed"));
                }
                if under_instance_xform {
                    return Err(FlattenError::Unsupported("Transformed primitive inside a transformed BVH instance"));
                }
                let id = self.object_id(h);
                let x = self.xform(t);
                return self.make_prim(child, x, id, under_instance_xform);
            }
            GpuShape::Sphere { center, radius, material } => {
                p.kind = RT_PRIM_SPHERE;
                p.material = self.material(material)?;
                self.n_spheres += 1;
                v3(&mut g.d, center);
                g.d[3] = radius;
            }
            GpuShape::MovingSphere { center_0, center_1, time_0, time_1, radius, material } => {
                p.kind = RT_PRIM_MOVING_SPHERE;
                p.material = self.material(material)?;
                self.n_moving += 1;
                v3(&mut g.d[0..3], center_0);
                v3(&mut g.d[3..6], center_1);
                g.d[6] = time_0;
                g.d[7] = time_1;
                g.d[8] = radius;
            }
            GpuShape::AARect { x_range, y_range, z, material } => {
                p.kind = RT_PRIM_AARECT;
                p.material = self.material(material)?;
                self.n_rects += 1;
                g.d[0] = x_range.0;
                g.d[1] = x_range.1;
                g.d[2] = y_range.0;
                g.d[3] = y_range.1;
                g.d[4] = z;
            }
            GpuShape::Triangle { indices, mesh } => {
                p.kind = RT_PRIM_TRIANGLE;
                p.index = self.tri_attrs.len() as u32;
                p.material = self.material(&mesh.material)?;
                let mut ta = rt_tri_attrs::default();
                let v: Vec<DVec3> = indices.iter().map(|i| mesh.vertices[*i as usize]).collect();
                let n = |i: u32| mesh.normals.get(i as usize).copied().unwrap_or(DVec3::ZERO);
                let uv = |i: u32| mesh.uv.get(i as usize).copied().unwrap_or(DVec2::ZERO);
                v3(&mut ta.n0, n(indices[0]));
                v3(&mut ta.n1, n(indices[1]));
                v3(&mut ta.n2, n(indices[2]));
                ta.uv0 = [uv(indices[0]).x, uv(indices[0]).y];
                ta.uv1 = [uv(indices[1]).x, uv(indices[1]).y];
                ta.uv2 = [uv(indices[2]).x, uv(indices[2]).y];
                v3(&mut g.d[0..3], v[0]);
                v3(&mut g.d[3..6], v[1] - v[0]); // edge1, edge2: the first two statements of mesh.rs:55-56
                v3(&mut g.d[6..9], v[2] - v[0]);
                self.tri_attrs.push(ta);
            }
            GpuShape::Bvh { .. } => {
                if self.inside_instance {
                    return Err(FlattenError::Unsupported("BVHNode object nested inside another BVHNode object"));
                }
                p.kind = RT_PRIM_INSTANCE;
                let slot = self.prims.len() as u32;
                match self.pending.iter_mut().find(|(b, _)| key(b) == key(h)) {
                    Some((_, slots)) => slots.push(slot),
                    None => self.pending.push((h.clone(), vec![slot])),
                }
            }
            GpuShape::Unsupported => return Err(FlattenError::Unsupported("Hittable without gpu_describe")),
        }
        p.object_id = self.object_id(h);
        g.kind = p.kind;
        g.xform = p.xform;
        self.geom.push(g);
        self.prims.push(p);
        Ok((self.prims.len() - 1) as u32)
    }

    /// Threaded pre-order: the reference always recurses left then right (bvh.rs:96-103), so `skip` = the next node in
    /// pre-order outside this subtree replaces the recursion; a leaf BVHNode carries its one or two objects inline.
    fn emit_bvh(&mut self, n: &Arc<dyn renderer::Hittable>, under_instance_xform: bool) -> Result<(), FlattenError> {
        let (left, right, same_child, bmin, bmax) = match n.gpu_describe() {
            GpuShape::Bvh { left, right, same_child, bmin, bmax } => (left.clone(), right.clone(), same_child, bmin, bmax),
            _ => return Err(FlattenError::Unsupported("emit_bvh on a non-BVH object")),
        };
        let idx = self.nodes.len();
        let mut nd = rt_node::default();
        v3(&mut nd.bmin, bmin);
        v3(&mut nd.bmax, bmax);
        nd.prim0 = RT_NONE;
        nd.prim1 = RT_NONE;
        self.nodes.push(nd);
        let inner = matches!(left.gpu_describe(), GpuShape::Bvh { .. }) && !same_child; // BVHNode::new recurses on both halves (bvh.rs:64-76)
        if inner {
            self.emit_bvh(&left, under_instance_xform)?;
            self.emit_bvh(&right, under_instance_xform)?;
        } else {
            let p0 = self.make_prim(&left, RT_NONE, RT_NONE, under_instance_xform)?;
            let p1 = if same_child { RT_NONE } else { self.make_prim(&right, RT_NONE, RT_NONE, under_instance_xform)? };
            self.nodes[idx].prim0 = p0;
            self.nodes[idx].prim1 = p1;
            if same_child {
                self.nodes[idx].flags |= RT_NODE_DUP_RIGHT;
            }
        }
        self.nodes[idx].skip = self.nodes.len() as u32;
        Ok(())
    }

    fn append<T: Copy>(out: &mut Vec<u8>, v: &[T]) -> u64 {
        while out.len() % 16 != 0 {
            out.push(0);
        }
        let off = out.len() as u64;
        let bytes = unsafe { std::slice::from_raw_parts(v.as_ptr() as *const u8, std::mem::size_of_val(v)) };
        out.extend_from_slice(bytes);
        off
    }

    /// `Scene` -> blob.  `bvh_seed` is informational (the reference draws its split axes from thread_rng, bvh.rs:53).
    pub fn run(mut self, scene: &renderer::Scene, bvh_seed: u64) -> Result<Vec<u8>, FlattenError> {
        let (objects, bvh) = scene.gpu_parts();
        for o in objects {
            self.object_id(o);
        }
        match bvh {
            Some(root) => self.emit_bvh(&root, false)?,
            None => {
                for o in objects {
                    // hittable.rs:41-53: tested in list order, no boxes
                    let at = self.nodes.len();
                    let mut nd = rt_node::default();
                    nd.flags = RT_NODE_NO_BOX;
                    nd.prim1 = RT_NONE;
                    self.nodes.push(nd);
                    let p0 = self.make_prim(o, RT_NONE, RT_NONE, false)?;
                    self.nodes[at].prim0 = p0;
                    self.nodes[at].skip = (at + 1) as u32;
                }
            }
        }
        let root_end = self.nodes.len() as u32;
        self.inside_instance = true;
        let pending = std::mem::take(&mut self.pending);
        for (root, slots) in &pending {
            let any_xf = slots.iter().any(|s| self.prims[*s as usize].xform != RT_NONE);
            let first = self.nodes.len() as u32;
            self.emit_bvh(root, any_xf)?;
            let end = self.nodes.len() as u32;
            for s in slots {
                self.prims[*s as usize].index = first;
                self.prims[*s as usize].sub_end = end;
                self.geom[*s as usize].d[0] = first as f64;
                self.geom[*s as usize].d[1] = end as f64;
            }
        }
        let mut lights = Vec::new();
        for l in &scene.lights {
            // only Sphere implements SampleableLight (shape.rs:36-53)
            let (center, radius, material, id) = l.gpu_light_parts();
            let mut r = rt_light::default();
            v3(&mut r.center, center);
            r.radius = radius;
            r.material = self.material(material)?;
            r.object_id = id.map(|h| self.object_id(h)).unwrap_or(RT_NONE);
            lights.push(r);
        }
        let xf: Vec<rt_xform> = self.xforms.iter().map(|m| rt_xform { t_inv: m.inverse().to_cols_array() }).collect(); // transform.rs:17
        let mut h = rt_blob_header::default();
        h.magic = RT_BLOB_MAGIC;
        h.version = RT_BLOB_VERSION;
        h.n_nodes = self.nodes.len() as u32;
        h.n_prims = self.prims.len() as u32;
        h.n_tris = self.tri_attrs.len() as u32;
        h.n_spheres = self.n_spheres;
        h.n_moving_spheres = self.n_moving;
        h.n_rects = self.n_rects;
        h.n_xforms = xf.len() as u32;
        h.n_materials = self.materials.len() as u32;
        h.n_textures = self.textures.len() as u32;
        h.n_lights = lights.len() as u32;
        h.n_texel_bytes = self.texels.len() as u32;
        h.root_node = 0;
        h.root_end = root_end;
        h.background = classify_background(scene.background)?;
        h.bvh_seed = bvh_seed;
        let (cam, lens_radius, time_0, time_1) = scene.camera.gpu_parts();
        v3(&mut h.camera.origin, cam[0]);
        v3(&mut h.camera.lower_left_corner, cam[1]);
        v3(&mut h.camera.horizontal, cam[2]);
        v3(&mut h.camera.vertical, cam[3]);
        v3(&mut h.camera.u, cam[4]);
        v3(&mut h.camera.v, cam[5]);
        h.camera.lens_radius = lens_radius;
        h.camera.time0 = time_0;
        h.camera.time1 = time_1;
        let mut out = vec![0u8; std::mem::size_of::<rt_blob_header>()];
        h.off_nodes = Self::append(&mut out, &self.nodes);
        h.off_prims = Self::append(&mut out, &self.prims);
        h.off_geom = Self::append(&mut out, &self.geom);
        h.off_tri_attrs = Self::append(&mut out, &self.tri_attrs);
        h.off_xforms = Self::append(&mut out, &xf);
        h.off_materials = Self::append(&mut out, &self.materials);
        h.off_textures = Self::append(&mut out, &self.textures);
        h.off_texels = Self::append(&mut out, &self.texels);
        h.off_lights = Self::append(&mut out, &lights);
        while out.len() % 16 != 0 {
            out.push(0);
        }
        h.total_bytes = out.len() as u64;
        let hb = unsafe { std::slice::from_raw_parts(&h as *const rt_blob_header as *const u8, std::mem::size_of::<rt_blob_header>()) };
        out[..hb.len()].copy_from_slice(hb);
        Ok(out)
    }
}

/// `scene.flatten()` for callers.
pub fn flatten(scene: &renderer::Scene) -> Result<Vec<u8>, FlattenError> {
    Flattener::default().run(scene, 0)
}
