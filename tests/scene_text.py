"""Tiny writer for "rtscene 1" descriptions used by the tests (ad-hoc scenes such as the reference's
unit-test fixtures).  Ids are returned in canonical post-order so oracle ids == host ids."""


class SceneText:
    def __init__(self, bvh_seed=0):
        self.lines = ["rtscene 1", f"bvh_seed {bvh_seed}"]
        self.n = dict(texture=0, material=0, mesh=0, xform=0, object=0)
        self.tail = []

    def _add(self, table, line):
        self.lines.append(line)
        i = self.n[table]
        self.n[table] += 1
        return i

    @staticmethod
    def _f(*v):
        return " ".join(repr(float(x)) for x in v)

    def solid(self, r, g, b): return self._add("texture", f"texture solid {self._f(r, g, b)}")
    def checker(self, odd, even): return self._add("texture", f"texture checker {odd} {even}")
    def image(self, w, h, rgb_bytes): return self._add("texture", f"texture image {w} {h} {bytes(rgb_bytes).hex()}")
    def lambertian(self, tex): return self._add("material", f"material lambertian {tex}")
    def lambertian_rgb(self, r, g, b): return self.lambertian(self.solid(r, g, b))
    def metal(self, r, g, b, fuzz): return self._add("material", f"material metal {self._f(r, g, b, fuzz)}")
    def dielectric(self, ior): return self._add("material", f"material dielectric {self._f(ior)}")
    def diffuse_light(self, tex): return self._add("material", f"material diffuse_light {tex}")

    def mesh(self, verts, normals, uvs, faces, mat):
        body = [f"mesh {len(verts)} {len(faces)} {mat}"]
        for v, n, t in zip(verts, normals, uvs):
            body.append(self._f(*v, *n, *t))
        for f in faces:
            body.append(f"{f[0]} {f[1]} {f[2]}")
        return self._add("mesh", "\n".join(body))

    def xform(self, m16): return self._add("xform", "xform " + self._f(*m16))
    def sphere(self, c, r, mat): return self._add("object", f"object sphere {self._f(*c, r)} {mat}")
    def moving_sphere(self, c0, c1, t0, t1, r, mat): return self._add("object", f"object moving_sphere {self._f(*c0, *c1, t0, t1, r)} {mat}")
    def aarect(self, x0, x1, y0, y1, z, mat): return self._add("object", f"object aarect {self._f(x0, x1, y0, y1, z)} {mat}")
    def triangle(self, mesh, face): return self._add("object", f"object triangle {mesh} {face}")
    def transformed(self, xf, child): return self._add("object", f"object transformed {xf} {child}")
    def bvh(self, kids, t0=0.0, t1=0.0): return self._add("object", f"object bvh {self._f(t0, t1)} {len(kids)} " + " ".join(map(str, kids)))

    def finish(self, objects, lights=(), camera=None, background="black", build_bvh=False):
        cam = camera or dict(look_from=(0, 0, 5), look_at=(0, 0, 0), up=(0, 1, 0), vfov=40.0, aspect=16.0 / 9.0, aperture=0.0, focus=5.0, t0=0.0, t1=0.0)
        out = list(self.lines)
        out.append(f"scene_objects {len(objects)} " + " ".join(map(str, objects)))
        out.append(f"lights {len(lights)} " + " ".join(map(str, lights)))
        out.append("camera " + self._f(*cam["look_from"], *cam["look_at"], *cam["up"], cam["vfov"], cam["aspect"], cam["aperture"], cam["focus"], cam["t0"], cam["t1"]))
        out.append(f"background {background}")
        out.append(f"build_bvh {1 if build_bvh else 0}")
        out.append("end")
        return "\n".join(out) + "\n"


def scale(sx, sy, sz): return [sx, 0, 0, 0, 0, sy, 0, 0, 0, 0, sz, 0, 0, 0, 0, 1]
def translate(tx, ty, tz): return [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, tx, ty, tz, 1]
