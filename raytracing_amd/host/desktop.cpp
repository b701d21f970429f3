// See desktop.hpp.  Scene functions are restatements of the scene constructors of
// desktop/src/main.rs against the mirrored host API; line references are to that file.
#include "desktop.hpp"
#include <array>
#include <fstream>
#include <sstream>

namespace desktop {

static const double FRAC_PI_4 = 0.785398163397448309615660845819875721;

// ---------------------------------------------------------------------------------------------
// Pcg64 = Lcg128Xsl64 (rand_pcg 0.3) seeded through rand_core's default seed_from_u64 (PCG32 fill)
// ---------------------------------------------------------------------------------------------
static const unsigned __int128 PCG_MULT = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;

Pcg64 Pcg64::seed_from_u64(uint64_t seed) {
    uint8_t bytes[32];
    uint64_t st = seed;
    for (int chunk = 0; chunk < 8; ++chunk) {
        st = st * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xorshifted = (uint32_t)(((st >> 18) ^ st) >> 27);
        uint32_t rot = (uint32_t)(st >> 59);
        uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        for (int b = 0; b < 4; ++b) bytes[4 * chunk + b] = (uint8_t)(x >> (8 * b));
    }
    uint64_t w[4];
    for (int i = 0; i < 4; ++i) { w[i] = 0; for (int b = 7; b >= 0; --b) w[i] = (w[i] << 8) | bytes[8 * i + b]; }
    unsigned __int128 state = (unsigned __int128)w[0] | ((unsigned __int128)w[1] << 64);
    unsigned __int128 incr = (unsigned __int128)w[2] | ((unsigned __int128)w[3] << 64);
    Pcg64 p;
    p.increment = (incr << 1) | 1;
    p.state = state + p.increment;
    p.state = p.state * PCG_MULT + p.increment;
    return p;
}
uint64_t Pcg64::next_u64() {
    state = state * PCG_MULT + increment;
    uint32_t rot = (uint32_t)(state >> 122);
    uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
    return (xsl >> rot) | (xsl << ((64 - rot) & 63));
}
double Pcg64::gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
double Pcg64::gen_range(double low, double high) {
    double scale = high - low;
    for (;;) {
        uint64_t bits = (next_u64() >> 12) | (1023ull << 52);
        double value1_2; memcpy(&value1_2, &bits, 8);
        double res = (value1_2 - 1.0) * scale + low;
        if (res < high) return res;
        uint64_t sb; memcpy(&sb, &scale, 8); sb -= 1; memcpy(&scale, &sb, 8);  // shrink scale by one ulp and retry
    }
}
static DVec3 random(Pcg64& rng) { double a = rng.gen_f64(), b = rng.gen_f64(), c = rng.gen_f64(); return DVec3(a, b, c); }            // math.rs:70-72
static DVec3 rand_in_range(Pcg64& rng, double lo, double hi) { double a = rng.gen_range(lo, hi), b = rng.gen_range(lo, hi), c = rng.gen_range(lo, hi); return DVec3(a, b, c); }  // math.rs:4-10

// ---------------------------------------------------------------------------------------------
// I/O
// ---------------------------------------------------------------------------------------------
std::optional<Image> load_texture(const std::string& filename) {
    // The reference decodes any format through the `image` crate (main.rs:892); the GPU box has no
    // decoder library, so textures are shipped pre-decoded as binary PPM (P6, maxval 255).
    std::ifstream f(filename, std::ios::binary);
    if (!f) return std::nullopt;
    std::string magic; f >> magic;
    if (magic != "P6") return std::nullopt;
    auto next_int = [&]() -> long {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string l; std::getline(f, l); }
            else if (isspace(c)) f.get();
            else break;
        }
        long v; f >> v; return v;
    };
    long w = next_int(), h = next_int(), mx = next_int();
    f.get();
    if (!f || w <= 0 || h <= 0 || mx != 255) return std::nullopt;
    Image img = Image::new_({(uint32_t)w, (uint32_t)h});
    img.rgb8.resize((size_t)w * h * 3);
    f.read((char*)img.rgb8.data(), (std::streamsize)img.rgb8.size());
    if (!f) return std::nullopt;
    for (uint32_t y = 0; y < (uint32_t)h; ++y)
        for (uint32_t x = 0; x < (uint32_t)w; ++x) {
            const uint8_t* p = &img.rgb8[((size_t)y * w + x) * 3];
            double r = (double)p[0] / 255.0, g = (double)p[1] / 255.0, b = (double)p[2] / 255.0;
            img.put(x, y, DVec3(r * r, g * g, b * b));  // powi(2): gamma approximation, main.rs:901-903
        }
    return img;
}

std::optional<std::vector<std::vector<Triangle>>> load_obj(const std::string& file_name, Arc<Material> material) {
    std::ifstream f(file_name);
    if (!f) return std::nullopt;
    std::vector<std::array<float, 3>> position, normal; std::vector<std::array<float, 2>> texture;
    struct Corner { long v, t, n; };
    std::vector<std::vector<std::array<Corner, 3>>> objects;  // polys per `o`, file order across groups
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string tag; ls >> tag;
        if (tag == "v") { std::array<float, 3> p{}; ls >> p[0] >> p[1] >> p[2]; position.push_back(p); }   // obj crate stores f32
        else if (tag == "vn") { std::array<float, 3> p{}; ls >> p[0] >> p[1] >> p[2]; normal.push_back(p); }
        else if (tag == "vt") { std::array<float, 2> p{}; ls >> p[0] >> p[1]; texture.push_back(p); }
        else if (tag == "o") objects.emplace_back();
        else if (tag == "f") {
            if (objects.empty()) objects.emplace_back();
            std::vector<Corner> cs; std::string tok;
            while (ls >> tok) {
                Corner c{0, 0, 0};
                size_t s1 = tok.find('/');
                c.v = std::stol(tok.substr(0, s1));
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) c.t = std::stol(t);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = std::stol(tok.substr(s2 + 1));
                }
                if (c.v < 0) c.v = (long)position.size() + 1 + c.v;
                if (c.t < 0) c.t = (long)texture.size() + 1 + c.t;
                if (c.n < 0) c.n = (long)normal.size() + 1 + c.n;
                cs.push_back(c);
            }
            if (cs.size() != 3) throw std::runtime_error("load_obj: non-triangle face (obj.rs:21 assert)");
            objects.back().push_back({cs[0], cs[1], cs[2]});
        }
    }
    std::vector<std::vector<Triangle>> meshes;
    for (auto& polys : objects) {  // obj.rs:11-55: un-indexed, 3 fresh vertices per face
        std::vector<DVec3> vertices, normals; std::vector<DVec2> uvs; std::vector<std::array<uint32_t, 3>> indices;
        uint32_t index = 0;
        for (auto& poly : polys) {
            for (auto& c : poly) {
                if (c.v < 1 || (size_t)c.v > position.size()) throw std::runtime_error("load_obj: position index out of range");
                auto& p = position[c.v - 1];
                vertices.push_back(DVec3((double)p[0], (double)p[1], (double)p[2]));
                if (c.t >= 1 && (size_t)c.t <= texture.size()) uvs.push_back(DVec2((double)texture[c.t - 1][0], (double)texture[c.t - 1][1]));
                else uvs.push_back(DVec2(0.0, 0.0));
                if (c.n >= 1 && (size_t)c.n <= normal.size()) normals.push_back(DVec3((double)normal[c.n - 1][0], (double)normal[c.n - 1][1], (double)normal[c.n - 1][2]));
                else normals.push_back(DVec3::X());
            }
            indices.push_back({index, index + 1, index + 2});
            index += 3;
        }
        meshes.push_back(create_mesh(std::move(vertices), std::move(normals), std::move(uvs), indices, material));
    }
    return meshes;
}

void write_image(const Image& img, const std::string& location) {
    if ((size_t)img.size.first * img.size.second != img.data.size()) throw std::runtime_error("write_image: size mismatch");
    FILE* f = fopen(location.c_str(), "w");
    if (!f) throw std::runtime_error("write_image: cannot create " + location);
    fprintf(f, "P3\n%u %u\n255\n", img.size.first, img.size.second);
    auto q = [](double c) -> unsigned {
        double p = std::pow(c, 0.5);
        double cl = p < 0.0 ? 0.0 : (p > 0.999 ? 0.999 : p);
        double s = cl * 256.0;
        return s != s ? 0u : (unsigned)s;
    };
    for (uint32_t y = img.size.second; y-- > 0;)  // rows bottom-up, ppm.rs:15
        for (uint32_t x = 0; x < img.size.first; ++x) {
            const DVec3& p = *img.get(x, y);
            fprintf(f, "%u %u %u\n", q(p.x), q(p.y), q(p.z));
        }
    fclose(f);
}

// The two writers below have no counterpart in the reference ("PNG/PFM output for diffing"): same frame, same
// orientation (rows bottom-up in memory), PFM keeps the linear doubles as float32, PNG holds exactly the bytes write_image prints.
void write_pfm(const Image& img, const std::string& location) {
    if ((size_t)img.size.first * img.size.second != img.data.size()) throw std::runtime_error("write_pfm: size mismatch");
    FILE* f = fopen(location.c_str(), "wb");
    if (!f) throw std::runtime_error("write_pfm: cannot create " + location);
    fprintf(f, "PF\n%u %u\n-1.0\n", img.size.first, img.size.second);   // negative scale = little endian; PFM rows run bottom to top, like Image
    std::vector<float> row((size_t)img.size.first * 3);
    for (uint32_t y = 0; y < img.size.second; ++y) {
        for (uint32_t x = 0; x < img.size.first; ++x) { const DVec3& p = *img.get(x, y); row[3 * x] = (float)p.x; row[3 * x + 1] = (float)p.y; row[3 * x + 2] = (float)p.z; }
        if (fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { fclose(f); throw std::runtime_error("write_pfm: short write"); }
    }
    fclose(f);
}

void write_png(const Image& img, const std::string& location) {
    if ((size_t)img.size.first * img.size.second != img.data.size()) throw std::runtime_error("write_png: size mismatch");
    const uint32_t w = img.size.first, h = img.size.second;
    auto q = [](double c) -> uint8_t {   // the quantisation of write_image (ppm.rs:20-33)
        double p = std::pow(c, 0.5);
        double cl = p < 0.0 ? 0.0 : (p > 0.999 ? 0.999 : p);
        double s = cl * 256.0;
        return s != s ? (uint8_t)0 : (uint8_t)(unsigned)s;
    };
    std::vector<uint8_t> raw;  // scanlines top-down, filter byte 0 each
    raw.reserve((size_t)h * (1 + 3 * (size_t)w));
    for (uint32_t y = h; y-- > 0;) {
        raw.push_back(0);
        for (uint32_t x = 0; x < w; ++x) { const DVec3& p = *img.get(x, y); raw.push_back(q(p.x)); raw.push_back(q(p.y)); raw.push_back(q(p.z)); }
    }
    static uint32_t table[256];
    if (!table[1]) for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); };
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        be32(out, (uint32_t)data.size());
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = start; i < out.size(); ++i) c = table[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
        be32(out, c ^ 0xFFFFFFFFu);
    };
    std::vector<uint8_t> ihdr; be32(ihdr, w); be32(ihdr, h); ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});   // 8-bit RGB
    chunk("IHDR", ihdr);
    std::vector<uint8_t> z = {0x78, 0x01};   // zlib stream of stored (uncompressed) deflate blocks
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
    be32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    FILE* f = fopen(location.c_str(), "wb");
    if (!f) throw std::runtime_error("write_png: cannot create " + location);
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) throw std::runtime_error("write_png: short write");
}

DVec3 aces_tonemapping(DVec3 pixel) {
    const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;
    auto ch = [&](double x) { double v = (x * (a * x + b)) / (x * (c * x + d) + e); return std::fmin(std::fmax(v, 0.0), 1.0); };
    return DVec3(ch(pixel.x), ch(pixel.y), ch(pixel.z));
}

// ---------------------------------------------------------------------------------------------
// Scenes
// ---------------------------------------------------------------------------------------------
Scene adapt(SceneDescription d, bool build_bvh) {
    auto b = Scene::build().objects(std::move(d.world)).camera(d.camera).background(d.background);
    if (build_bvh) return std::move(b).build_bvh().build();
    return std::move(b).build();
}

static Arc<Material> checker_ground() {  // the ground material every legacy scene starts with
    return std::make_shared<Lambertian>(std::make_shared<CheckerTexture>(CheckerTexture::new_(DVec3(0.2, 0.3, 0.1), DVec3(0.9, 0.9, 0.9))));
}
static Arc<Material> solid_light(DVec3 c) { return std::make_shared<DiffuseLight>(std::make_shared<SolidColour>(c)); }
static const double ASPECT = 16.0 / 9.0;

static std::vector<Triangle> cube_mesh(const Assets& a, Arc<Material> m) {
    // main.rs:118 loads "F:\Models\cube.obj", which the reference checkout does not contain
    // (.MISSING_LARGE_BLOBS); assets/cube.obj is a generated +-1 cube (documented in DESIGN.md).
    auto meshes = load_obj(a.root + "/cube.obj", std::move(m));
    if (!meshes || meshes->empty()) throw std::runtime_error("cannot load " + a.root + "/cube.obj");
    std::vector<Triangle> last = std::move(meshes->back());  // .pop().unwrap()
    return last;
}

SceneDescription create_cube_scene() {  // main.rs:51-91
    HittableList world{std::make_shared<Sphere>(DVec3(-0.0, -100.5, -1.0), 100.0, checker_ground())};
    DVec3 look_from(1.0, 3.0, 4.0), look_at(0.0, 0.0, -1.0), up(0.0, 1.0, 0.0);
    double dist_to_focus = (look_at - look_from).length();
    return {world, Camera::new_(look_from, look_at, up, 30.0, ASPECT, 0.1, dist_to_focus, 0.0, 1.0), Background::Skybox};
}

SceneDescription create_cornell_scene(const Assets& a) {  // main.rs:92-211
    HittableList world{std::make_shared<Sphere>(DVec3(-0.0, -101.0, -1.0), 100.0, checker_ground())};
    auto grey_cube = std::make_shared<BVHNode>(BVHNode::from_mesh(cube_mesh(a, std::make_shared<Lambertian>(Lambertian::new_(DVec3::splat(0.1)))), 0.0, 0.0));
    auto lit_cube = std::make_shared<BVHNode>(BVHNode::from_mesh(cube_mesh(a, solid_light(DVec3::splat(1.0))), 0.0, 0.0));
    auto red_cube = std::make_shared<BVHNode>(BVHNode::from_mesh(cube_mesh(a, std::make_shared<Lambertian>(Lambertian::new_(DVec3(1.0, 0.0, 0.0)))), 0.0, 0.0));
    auto green_cube = std::make_shared<BVHNode>(BVHNode::from_mesh(cube_mesh(a, std::make_shared<Lambertian>(Lambertian::new_(DVec3(0.0, 1.0, 0.0)))), 0.0, 0.0));
    auto place = [&](DVec3 t, DVec3 s, Arc<Hittable> h) {
        world.push_back(std::make_shared<Transformed>(Transformed::new_(DMat4::from_translation(t) * DMat4::from_scale(s), std::move(h))));
    };
    place(DVec3(0.0, 0.0, -1.5), DVec3(1.5, 1.5, 0.01), lit_cube);      // back wall
    place(DVec3(1.5, 0.0, 0.0), DVec3(0.001, 1.5, 1.5), red_cube);      // right wall
    place(DVec3(-1.5, 0.0, 0.0), DVec3(0.001, 1.5, 1.5), green_cube);   // left wall
    place(DVec3(0.0, -1.0, 0.0), DVec3(1.5, 0.1, 2.0), grey_cube);      // floor
    place(DVec3(0.0, 1.0, 0.0), DVec3(1.5, 0.1, 2.0), grey_cube);       // ceiling
    world.push_back(std::make_shared<Transformed>(Transformed::new_(DMat4::from_translation(DVec3(0.0, 0.0, 0.0)), grey_cube)));  // box
    DVec3 look_from(0.0, 0.0, 5.0), look_at(0.0, 0.0, 0.0), up(0.0, 1.0, 0.0);
    double dist_to_focus = (look_at - look_from).length();
    return {world, Camera::new_(look_from, look_at, up, 30.0, ASPECT, 0.1, dist_to_focus, 0.0, 1.0), Background::NoLight};
}

SceneDescription create_sphere_scene(const Assets& a) {  // main.rs:212-278
    HittableList world{std::make_shared<Sphere>(DVec3(-0.0, -101.0, -1.0), 100.0, checker_ground())};
    auto sphere_material = std::make_shared<Metal>(Metal::new_(DVec3(0.8, 0.6, 0.2), 0.2));
    auto bvh_cube = std::make_shared<BVHNode>(BVHNode::from_mesh(cube_mesh(a, sphere_material), 0.0, 1.0));
    DMat4 transformation = DMat4::from_scale(DVec3(2.0, 1.0, 1.0)) * DMat4::from_rotation_y(FRAC_PI_4) * DMat4::from_translation(DVec3(1.0, 0.0, 0.0));
    world.push_back(std::make_shared<Transformed>(Transformed::new_(transformation, bvh_cube)));
    DVec3 look_from(0.0, 0.0, 5.0), look_at(0.0, 0.0, 0.0), up(0.0, 1.0, 0.0);
    double dist_to_focus = (look_at - look_from).length();
    return {world, Camera::new_(look_from, look_at, up, 30.0, ASPECT, 0.1, dist_to_focus, 0.0, 1.0), Background::Skybox};
}

SceneDescription create_simple_scene(const Assets& a) {  // main.rs:281-362
    auto earth = load_texture(a.root + "/textures/earthmap.ppm");
    if (!earth) throw std::runtime_error("cannot load " + a.root + "/textures/earthmap.ppm");
    Arc<Material> center_material = std::make_shared<Lambertian>(std::make_shared<Image>(std::move(*earth)));
    auto shiny_metal_material = std::make_shared<Metal>(Metal::new_(DVec3(0.1, 0.1, 0.1), 0.0));
    auto left_material = std::make_shared<Dielectric>(1.5);
    Arc<Material> right_material = std::make_shared<Metal>(Metal::new_(DVec3(0.8, 0.6, 0.2), 1.0));
    HittableList world{
        std::make_shared<Transformed>(Transformed::new_(DMat4::from_scale(DVec3::splat(1.4)), std::make_shared<Sphere>(DVec3(0.0, 0.0, -1.0), 0.5, center_material))),
        std::make_shared<Sphere>(DVec3(-1.0, 0.0, -1.0), 0.5, left_material),
        std::make_shared<Sphere>(DVec3(-1.0, 0.0, -1.0), -0.49, left_material),
        std::make_shared<Sphere>(DVec3(1.0, 0.0, -1.0), 0.5, right_material),
        std::make_shared<Sphere>(Sphere(DVec3(0.0, 0.0, 0.0), 1.0, shiny_metal_material)
                                     .transform(DMat4::from_scale(DVec3(0.5, 0.5, 0.5)) * DMat4::from_translation(DVec3(0.0, 2.0, 0.0)))),
        std::make_shared<Sphere>(DVec3(-0.0, -100.5, -1.0), 100.0, checker_ground()),
    };
    DVec3 look_from(-5.0, 0.8, -3.5), look_at(0.0, 0.0, -1.0), up(0.0, 1.0, 0.0);
    double dist_to_focus = (look_at - look_from).length();
    return {world, Camera::new_(look_from, look_at, up, 30.0, ASPECT, 0.1, dist_to_focus, 0.0, 1.0), Background::Skybox};
}

Scene create_random_scene(bool motion_blur) {  // main.rs:365-494
    auto scene_builder = Scene::build();
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(0.0, -1000.0, 0.0), 1000.0, std::make_shared<Lambertian>(Lambertian::new_(DVec3(0.5, 0.5, 0.5)))));
    Pcg64 rng = Pcg64::seed_from_u64(10);
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double material_choice = rng.gen_f64();
            double cx = (double)a + (0.9 * rng.gen_f64());
            double cz = (double)b + (0.9 * rng.gen_f64());
            DVec3 center(cx, 0.2, cz);
            if ((center - DVec3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (material_choice < 0.8) {  // diffuse, moving
                    DVec3 r1 = random(rng), r2 = random(rng);
                    auto material = std::make_shared<Lambertian>(Lambertian::new_(r1 * r2));
                    DVec3 center_2 = center + DVec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
                    scene_builder.add_object(std::make_shared<MovingSphere>(center, center_2, 0.0, 1.0, 0.2, material));
                } else if (material_choice < 0.95) {  // metal
                    DVec3 albedo = rand_in_range(rng, 0.5, 1.0);
                    double fuzz = rng.gen_range(0.0, 0.5);
                    scene_builder.add_object(std::make_shared<Sphere>(center, 0.2, std::make_shared<Metal>(Metal::new_(albedo, fuzz))));
                } else {  // glass
                    scene_builder.add_object(std::make_shared<Sphere>(center, 0.2, std::make_shared<Dielectric>(1.5)));
                }
            }
        }
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(0.0, 1.0, 0.0), 1.0, std::make_shared<Dielectric>(1.5)));
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(-4.0, 1.0, 0.0), 1.0, std::make_shared<Lambertian>(Lambertian::new_(DVec3(0.4, 0.2, 0.1)))));
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(4.0, 1.0, 0.0), 1.0, std::make_shared<Metal>(Metal::new_(DVec3(0.7, 0.6, 0.5), 0.0))));
    auto light = std::make_shared<Sphere>(DVec3(-10.0, 5.0, 10.0), 1.0, solid_light(DVec3(114.0, 114.0, 114.0)));
    scene_builder.add_object(light);
    auto light2 = std::make_shared<Sphere>(DVec3(8.0, 5.0, 0.0), 2.0, solid_light(DVec3(10.0, 10.0, 10.0)));
    scene_builder.add_object(light2);
    DVec3 look_from(8.0, 2.0, 10.0), look_at(0.0, 0.0, 0.0), up(0.0, 1.0, 0.0);
    return std::move(scene_builder)
        .lights({light, light2})
        .camera(Camera::new_(look_from, look_at, up, 20.0, ASPECT, 0.01, 10.0, 0.0, motion_blur ? 1.0 : 0.0))
        .background(Background::NoLight)
        .build_bvh()
        .build();
}

SceneDescription two_spheres() {  // main.rs:497-545
    auto material = checker_ground();
    DVec3 look_from(13.0, 2.0, 3.0), look_at(0.0, 0.0, 0.0), up(0.0, 1.0, 0.0);
    double focus_dist = (look_from - look_at).length();
    HittableList world{
        std::make_shared<Sphere>(DVec3(0.0, -10.0, 0.0), 10.0, material),
        std::make_shared<Sphere>(DVec3(0.0, 10.0, 0.0), 10.0, material),
        std::make_shared<Sphere>(DVec3(10.0, 2.0, 0.5), 1.0, solid_light(DVec3(4.0, 4.0, 4.0))),
    };
    return {world, Camera::new_instant(look_from, look_at, up, 20.0, ASPECT, 0.1, focus_dist), Background::NoLight};
}

SceneDescription simple_triangle_scene() {  // main.rs:547-633 (the "broken cube": 10 of 12 faces)
    auto triangle_mat = std::make_shared<Lambertian>(Lambertian::new_(DVec3::splat(0.5)));
    HittableList world{std::make_shared<Sphere>(DVec3(-0.0, -100.5, -1.0), 100.0, checker_ground())};
    auto tris = create_mesh(
        {DVec3(-0.5, -0.5, 0.5), DVec3(0.5, 0.5, 0.5), DVec3(-0.5, 0.5, 0.5), DVec3(0.5, -0.5, 0.5),
         DVec3(0.5, -0.5, -0.5), DVec3(0.5, 0.5, -0.5), DVec3(-0.5, 0.5, -0.5), DVec3(-0.5, -0.5, -0.5)},
        {DVec3(0.0, 0.0, 1.0), DVec3(0.0, 0.0, 1.0), DVec3(0.0, 0.0, 1.0), DVec3(0.0, 0.0, 1.0),
         DVec3(1.0, 0.0, 0.0), DVec3(1.0, 0.0, 0.0), DVec3(0.0, 1.0, 0.0), DVec3(0.0, -1.0, 0.0)},
        {DVec2(0.0, 0.0), DVec2(1.0, 1.0), DVec2(0.0, 1.0), DVec2(1.0, 0.0), DVec2(1.0, 0.0), DVec2(1.0, 1.0), DVec2(1.0, 0.0), DVec2(0.0, 0.0)},
        {{0, 1, 2}, {0, 3, 1}, {3, 4, 1}, {4, 5, 1}, {2, 1, 6}, {1, 5, 6}, {7, 0, 2}, {7, 2, 6}, {6, 5, 7}, {7, 5, 4}},
        triangle_mat);
    for (auto& t : tris) world.push_back(std::make_shared<Triangle>(std::move(t)));
    DVec3 look_from(0.0, 0.5, -2.0), look_at(0.0, 0.0, 0.0), up(0.0, 1.0, 0.0);
    double dist_to_focus = (look_at - look_from).length();
    return {world, Camera::new_(look_from, look_at, up, 90.0, ASPECT, 0.1, dist_to_focus, 0.0, 1.0), Background::Skybox};
}

SceneDescription simple_light(const Assets& a) {  // main.rs:636-685
    auto earth = load_texture(a.root + "/textures/earthmap.ppm");
    if (!earth) throw std::runtime_error("cannot load " + a.root + "/textures/earthmap.ppm");
    HittableList world{
        std::make_shared<Sphere>(DVec3(0.0, -1000.0, 0.0), 1000.0, std::make_shared<Lambertian>(Lambertian::new_(DVec3(0.9, 0.9, 0.9)))),
        std::make_shared<Sphere>(DVec3(0.0, 2.0, 0.0), 2.0, std::make_shared<Lambertian>(std::make_shared<Image>(std::move(*earth)))),
        std::make_shared<Sphere>(DVec3(0.0, 6.5, 0.0), 2.0, solid_light(DVec3(4.0, 4.0, 4.0))),
        std::make_shared<AARect>(std::make_pair(3.0, 5.0), std::make_pair(1.0, 3.0), -2.0, solid_light(DVec3(4.0, 4.0, 4.0))),
    };
    DVec3 look_from(26.0, 3.0, 6.0), look_at(0.0, 2.0, 0.0);
    return {world, Camera::new_instant(look_from, look_at, DVec3(0.0, 1.0, 0.0), 20.0, ASPECT, 0.1, (look_at - look_from).length()), Background::NoLight};
}

Scene single_sphere_light_scene() {  // main.rs:687-787
    auto scene_builder = Scene::build();
    auto light = std::make_shared<Sphere>(DVec3(10.0, 5.0, 0.0), 1.0, solid_light(DVec3(114.0, 114.0, 114.0)));
    scene_builder.add_object(light);
    auto light2 = std::make_shared<Sphere>(DVec3(-10.0, 5.0, -10.0), 1.0, solid_light(DVec3(114.0, 114.0, 114.0)));
    scene_builder.add_object(light2);
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(0.0, -1000.0, 0.0), 1000.0, checker_ground()));
    DVec3 look_from(0.0, 10.0, 20.0), look_at(0.0, 0.0, 0.0);
    Camera camera = Camera::new_instant(look_from, look_at, DVec3(0.0, 1.0, 0.0), 80.0, ASPECT, 0.1, (look_at - look_from).length());
    return std::move(scene_builder).lights({light, light2}).camera(camera).background(Background::NoLight).build_bvh().build();
}

Scene mesh_scene(const Assets& a, bool textured) {  // main.rs:789-889
    // main.rs:791 wants models/pagoda/textures/albedo.png, absent from the reference checkout
    // (.MISSING_LARGE_BLOBS): the default is the alternative the reference keeps commented at
    // main.rs:794, Lambertian(0.5); `textured` uses assets/pagoda/albedo.ppm if a user supplies one.
    Arc<Material> triangle_mat;
    if (textured) {
        auto tex = load_texture(a.root + "/pagoda/albedo.ppm");
        if (!tex) throw std::runtime_error("cannot load " + a.root + "/pagoda/albedo.ppm");
        triangle_mat = std::make_shared<Lambertian>(std::make_shared<Image>(std::move(*tex)));
    } else triangle_mat = std::make_shared<Lambertian>(Lambertian::new_(DVec3::splat(0.5)));
    auto test_mesh = load_obj(a.root + "/pagoda/model_triangulated.obj", triangle_mat);
    if (!test_mesh) throw std::runtime_error("cannot load " + a.root + "/pagoda/model_triangulated.obj");
    auto scene_builder = Scene::build();
    for (auto& mesh : *test_mesh)
        for (auto& triangle : mesh)
            scene_builder.add_object(std::make_shared<Transformed>(Transformed::new_(DMat4::from_translation(DVec3::splat(5.0)), std::make_shared<Triangle>(std::move(triangle)))));
    auto light = std::make_shared<Sphere>(DVec3(40.0, 40.0, 20.0), 25.0, solid_light(DVec3(4.0, 4.0, 4.0)));
    scene_builder.add_object(light);
    auto light2 = std::make_shared<Sphere>(DVec3(-40.0, 40.0, -20.0), 25.0, solid_light(DVec3(4.0, 4.0, 4.0)));
    scene_builder.add_object(light2);
    scene_builder.add_object(std::make_shared<Sphere>(DVec3(0.0, -1000.0, 0.0), 1000.0, checker_ground()));
    DVec3 look_from(-2.0, 30.0, 30.0), look_at(0.0, 15.0, 0.0);
    Camera camera = Camera::new_instant(look_from, look_at, DVec3(0.0, 1.0, 0.0), 80.0, ASPECT, 0.1, (look_at - look_from).length());
    return std::move(scene_builder).lights({light, light2}).camera(camera).background(Background::NoLight).build_bvh().build();
}

Scene scene_by_name(const std::string& name, const Assets& a, bool legacy_bvh) {
    if (name == "random") return create_random_scene(false);
    if (name == "random_blur") return create_random_scene(true);
    if (name == "mesh") return mesh_scene(a, false);
    if (name == "mesh_textured") return mesh_scene(a, true);
    if (name == "single_sphere_light") return single_sphere_light_scene();
    if (name == "simple") return adapt(create_simple_scene(a), legacy_bvh);
    if (name == "simple_triangle") return adapt(simple_triangle_scene(), legacy_bvh);
    if (name == "cornell") return adapt(create_cornell_scene(a), legacy_bvh);
    if (name == "sphere") return adapt(create_sphere_scene(a), legacy_bvh);
    if (name == "cube") return adapt(create_cube_scene(), legacy_bvh);
    if (name == "two_spheres") return adapt(two_spheres(), legacy_bvh);
    if (name == "simple_light") return adapt(simple_light(a), legacy_bvh);
    throw std::runtime_error("unknown scene name: " + name);
}

}  // namespace desktop
