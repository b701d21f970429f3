// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (raytracing_amd/, librt_hip.so, librt_host.so).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement (C++17, f64, no FMA contraction) of the vector/matrix arithmetic the reference
// takes from glam ^0.17.3 (renderer/Cargo.toml:11) and the sampling helpers of
// renderer/src/math.rs + renderer/src/onb.rs.  glam's source is NOT in /root/reference, so the
// operation order below restates glam 0.17's published scalar f64 implementation:
//   dot        = (x*x' + y*y') + z*z'           (left-assoc sum)
//   cross      = (y*z' - y'*z, z*x' - z'*x, x*y' - x'*y)
//   normalize  = v * (1.0 / sqrt(dot(v,v)))     (mul by length_recip, NOT a divide)
//   v / s      = component-wise divide
//   M * v4     = ((c0*x + c1*y) + c2*z) + c3*w
//   M.inverse  = cofactor expansion (GLM form) scaled by 1/det
// Rounding-level parity with real glam is pinned only by the reference's exactly-representable
// known-answer tests (bvh.rs:129-173, transform.rs:81-101, mesh.rs:175-239): "parity unpinned"
// at the ulp level, stated in DESIGN.md.
//
// RNG: the reference draws from rand::thread_rng() (OS-seeded, not reproducible:
// integrator.rs:36,179,268,335; camera.rs:84; main.rs:990).  The oracle substitutes the
// counter-based Philox4x32-10 generator keyed on (seed) with counter (pixel, sample, level, block),
// level 0 = camera sample, level k>=1 = k-th ray_colour call along the path.  Draw ORDER inside a
// level follows the reference statement order exactly.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace orc {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double FRAC_1_PI = 0.318309886183790671537767526745028724;
constexpr double F64_EPSILON = 2.2204460492503131e-16;

struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    static V3 splat(double s) { return V3(s, s, s); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline double length_squared(V3 a) { return dot(a, a); }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / length(a)); }
inline V3 vmin(V3 a, V3 b) { return V3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return V3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
inline double distance_squared(V3 a, V3 b) { return length_squared(a - b); }

struct V4 {
    double x, y, z, w;
    V4() : x(0), y(0), z(0), w(0) {}
    V4(double a, double b, double c, double d) : x(a), y(b), z(c), w(d) {}
    V4(V3 v, double d) : x(v.x), y(v.y), z(v.z), w(d) {}
    V3 xyz() const { return V3(x, y, z); }
};
inline V4 operator+(V4 a, V4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline V4 operator-(V4 a, V4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline V4 operator*(V4 a, V4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 operator*(V4 a, double s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline double length(V4 a) { return std::sqrt((((a.x * a.x) + (a.y * a.y)) + (a.z * a.z)) + (a.w * a.w)); }

// Column-major 4x4 (glam DMat4: x_axis..w_axis are columns).
struct M4 {
    V4 c0, c1, c2, c3;
    static M4 identity() {
        M4 m; m.c0 = V4(1, 0, 0, 0); m.c1 = V4(0, 1, 0, 0); m.c2 = V4(0, 0, 1, 0); m.c3 = V4(0, 0, 0, 1); return m;
    }
    static M4 from_scale(V3 s) {
        M4 m; m.c0 = V4(s.x, 0, 0, 0); m.c1 = V4(0, s.y, 0, 0); m.c2 = V4(0, 0, s.z, 0); m.c3 = V4(0, 0, 0, 1); return m;
    }
    static M4 from_translation(V3 t) {
        M4 m = identity(); m.c3 = V4(t.x, t.y, t.z, 1); return m;
    }
    static M4 from_rotation_y(double angle) {
        double s = std::sin(angle), c = std::cos(angle);
        M4 m; m.c0 = V4(c, 0, -s, 0); m.c1 = V4(0, 1, 0, 0); m.c2 = V4(s, 0, c, 0); m.c3 = V4(0, 0, 0, 1); return m;
    }
    V4 mul(V4 v) const {
        V4 r = c0 * v.x;
        r = r + c1 * v.y;
        r = r + c2 * v.z;
        r = r + c3 * v.w;
        return r;
    }
    M4 mul(const M4& o) const {
        M4 m; m.c0 = mul(o.c0); m.c1 = mul(o.c1); m.c2 = mul(o.c2); m.c3 = mul(o.c3); return m;
    }
    M4 transpose() const {
        M4 m;
        m.c0 = V4(c0.x, c1.x, c2.x, c3.x); m.c1 = V4(c0.y, c1.y, c2.y, c3.y);
        m.c2 = V4(c0.z, c1.z, c2.z, c3.z); m.c3 = V4(c0.w, c1.w, c2.w, c3.w);
        return m;
    }
    // glam 0.17 DMat4::inverse (GLM cofactor form).
    M4 inverse() const {
        double m00 = c0.x, m01 = c0.y, m02 = c0.z, m03 = c0.w;
        double m10 = c1.x, m11 = c1.y, m12 = c1.z, m13 = c1.w;
        double m20 = c2.x, m21 = c2.y, m22 = c2.z, m23 = c2.w;
        double m30 = c3.x, m31 = c3.y, m32 = c3.z, m33 = c3.w;
        double coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
        double coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
        double coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
        double coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
        double coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
        double coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
        V4 fac0(coef00, coef00, coef02, coef03), fac1(coef04, coef04, coef06, coef07), fac2(coef08, coef08, coef10, coef11);
        V4 fac3(coef12, coef12, coef14, coef15), fac4(coef16, coef16, coef18, coef19), fac5(coef20, coef20, coef22, coef23);
        V4 vec0(m10, m00, m00, m00), vec1(m11, m01, m01, m01), vec2(m12, m02, m02, m02), vec3(m13, m03, m03, m03);
        V4 inv0 = (vec1 * fac0 - vec2 * fac1) + vec3 * fac2;
        V4 inv1 = (vec0 * fac0 - vec2 * fac3) + vec3 * fac4;
        V4 inv2 = (vec0 * fac1 - vec1 * fac3) + vec3 * fac5;
        V4 inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5;
        V4 sign_a(1.0, -1.0, 1.0, -1.0), sign_b(-1.0, 1.0, -1.0, 1.0);
        M4 inv; inv.c0 = inv0 * sign_a; inv.c1 = inv1 * sign_b; inv.c2 = inv2 * sign_a; inv.c3 = inv3 * sign_b;
        V4 col0(inv.c0.x, inv.c1.x, inv.c2.x, inv.c3.x);
        V4 dot0 = c0 * col0;
        double dot1 = ((dot0.x + dot0.y) + dot0.z) + dot0.w;
        double rcp_det = 1.0 / dot1;
        M4 r; r.c0 = inv.c0 * rcp_det; r.c1 = inv.c1 * rcp_det; r.c2 = inv.c2 * rcp_det; r.c3 = inv.c3 * rcp_det;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), restated from the published algorithm.
// ---------------------------------------------------------------------------------------------
inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

// Stream of 53-bit uniform doubles in [0,1): counter = (pixel, sample, level, block); each block
// yields two doubles ((w1:w0)>>11, (w3:w2)>>11) * 2^-53  (rand 0.8 `Standard` f64 conversion).
// ---- path signatures (tests only; the device computes the same hash, raytracing_amd/csrc/rt_math.h) ---------------------
// Sum of hashes of (level, tag, value) over every discrete decision of a path; see rt_math.h for the list and the purpose.
enum { SIG_HIT = 1, SIG_FRONT = 2, SIG_TEX = 3, SIG_KIND = 4, SIG_DELTA = 5, SIG_LIGHT = 6, SIG_VIS = 7, SIG_GATE = 8, SIG_CUT = 9 };
inline thread_local uint64_t* g_sig = nullptr;   // where the current sample's signature accumulates (null: off)
inline uint64_t sig_hash(uint32_t level, uint32_t tag, uint64_t value) {
    uint64_t x = ((uint64_t)level << 56) ^ ((uint64_t)tag << 48) ^ value;
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline void sig_event(uint32_t level, uint32_t tag, uint64_t value) { if (g_sig) *g_sig += sig_hash(level, tag, value); }

struct Rng {
    uint32_t k0, k1, pixel, sample, level, draw;
    uint32_t w[4];
    Rng(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t level_)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), pixel(pixel_), sample(sample_), level(level_), draw(0) {}
    double next_f64() {
        if ((draw & 1u) == 0) {
            w[0] = pixel; w[1] = sample; w[2] = level; w[3] = draw >> 1;
            philox4x32_10(w, k0, k1);
        }
        uint64_t bits = (draw & 1u) ? (((uint64_t)w[3] << 32) | w[2]) : (((uint64_t)w[1] << 32) | w[0]);
        ++draw;
        return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    }
    // rand `gen_range(lo..hi)` for f64 restated as (hi-lo)*xi + lo.
    double range(double lo, double hi) { return (hi - lo) * next_f64() + lo; }
    // rand `SliceRandom::choose` restated as floor(xi*n) clamped.
    uint32_t index(uint32_t n) {
        uint32_t i = (uint32_t)(next_f64() * (double)n);
        return i < n ? i : n - 1;
    }
};

// ---------------------------------------------------------------------------------------------
// renderer/src/math.rs
// ---------------------------------------------------------------------------------------------
inline V3 rand_in_range(Rng& rng, double lo, double hi) {  // math.rs:4-10
    double a = rng.range(lo, hi), b = rng.range(lo, hi), c = rng.range(lo, hi);
    return V3(a, b, c);
}
inline V3 rand_in_unit_sphere(Rng& rng) {  // math.rs:12-19
    for (;;) {
        V3 p = rand_in_range(rng, -1.0, 1.0);
        if (length(p) < 1.0) return p;
    }
}
inline V3 rand_cosine_hemisphere(Rng& rng) {  // math.rs:25-36
    double r1 = rng.next_f64();
    double r2 = rng.next_f64();
    double z = std::sqrt(1.0 - r2);
    double phi = 2.0 * PI * r1;
    double sqrt_r2 = std::sqrt(r2);
    double x = std::cos(phi) * sqrt_r2;
    double y = std::sin(phi) * sqrt_r2;
    return V3(x, y, z);
}
inline V3 rand_hemisphere(Rng& rng) {  // math.rs:39-45
    double z = rng.next_f64();
    double r = std::sqrt(std::fmax(0.0, 1.0 - z * z));
    double phi = 2.0 * PI * rng.next_f64();
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}
inline V3 reflect(V3 a, V3 b) { return a - (b * dot(a, b) * 2.0); }  // math.rs:51-53
inline V3 refract(V3 v, V3 n, double etai_over_etat) {               // math.rs:62-68
    double cos_theta = std::fmin(dot(-v, n), 1.0);
    V3 r_out_perp = (v + (n * cos_theta)) * etai_over_etat;
    V3 r_out_parallel = n * (-std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))));
    return r_out_perp + r_out_parallel;
}
inline V3 random_v3(Rng& rng) { double a = rng.next_f64(), b = rng.next_f64(), c = rng.next_f64(); return V3(a, b, c); }

// renderer/src/onb.rs:9-26
struct OrthoNormalBasis {
    V3 u, v, w;
    static OrthoNormalBasis from_w(V3 w) {
        V3 a = std::fabs(w.x) > 0.9 ? V3(0.0, 1.0, 0.0) : V3(1.0, 0.0, 0.0);
        OrthoNormalBasis b;
        b.v = normalize(cross(w, a));
        b.u = cross(w, b.v);
        b.w = w;
        return b;
    }
    V3 local(V3 vec) const { return ((vec.x * u) + (vec.y * v)) + (vec.z * w); }
};

}  // namespace orc
