// Device-side f64 vector math, Philox RNG and sampling helpers for the gfx950 kernels.
// Operation ORDER is part of the contract: the closest-hit result must be bit-exact against the
// reference's f64 arithmetic (glam ^0.17 semantics: left-assoc dot, normalize = v * (1/len)),
// so this file is compiled with -ffp-contract=off and every expression is written in the
// association the reference uses (renderer/src/{math,onb,ray}.rs, glam DVec3/DMat4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rt {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double FRAC_1_PI = 0.318309886183790671537767526745028724;
constexpr double F64_EPS = 2.2204460492503131e-16;

struct d3 {
    double x, y, z;
};
RT_DEV d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV d3 operator-(d3 a) { return mk(-a.x, -a.y, -a.z); }
RT_DEV d3 operator*(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV d3 operator*(d3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_DEV d3 operator*(double s, d3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RT_DEV d3 operator/(d3 a, double s) { return mk(a.x / s, a.y / s, a.z / s); }
RT_DEV double dot(d3 a, d3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
RT_DEV d3 cross(d3 a, d3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
RT_DEV double length_squared(d3 a) { return dot(a, a); }
RT_DEV double length(d3 a) { return __builtin_sqrt(dot(a, a)); }
RT_DEV d3 normalize(d3 a) { return a * (1.0 / length(a)); }
RT_DEV d3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }

struct Ray {
    d3 o, d;
    double time;
};
RT_DEV d3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }  // ray.rs:10-12

// column-major 4x4 times (v, w): ((c0*x + c1*y) + c2*z) + c3*w, xyz only (glam DMat4 * DVec4)
RT_DEV d3 mat_mul_xyz(const double* m, d3 v, double w) {
    return mk(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * w,
              ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * w,
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * w);
}
// transpose(m) * (v, 0), xyz only (transform.rs:34,40)
RT_DEV d3 mat_t_mul_xyz0(const double* m, d3 v) {
    return mk(((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3] * 0.0,
              ((m[4] * v.x + m[5] * v.y) + m[6] * v.z) + m[7] * 0.0,
              ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11] * 0.0);
}

// ---- Philox4x32-10, counter = (pixel, sample, level, block), key = seed ------------------------
RT_DEV void mul_wide_u32(uint32_t k, uint32_t c, uint32_t& hi, uint32_t& lo) {
#ifdef RT_PHILOX_SPLIT_MUL
    hi = __umulhi(k, c); lo = k * c;
#else
    unsigned long long p, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=&v"(p), "=s"(carry) : "s"(k), "v"(c));
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
#endif
}
RT_DEV void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;   // both halves of a product from ONE v_mad_u64_u32 (left to itself the compiler issues v_mul_hi_u32 + v_mul_lo_u32, each as slow)
        mul_wide_u32(0xD2511F53u, c0, hi0, lo0);
        mul_wide_u32(0xCD9E8D57u, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct Rng {
    uint32_t k0, k1, pixel, sample, level, draw;
    uint32_t w2, w3;
    RT_DEV Rng(uint64_t seed, uint32_t px, uint32_t s, uint32_t lv) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), pixel(px), sample(s), level(lv), draw(0), w2(0), w3(0) {}
    RT_DEV double next_f64() {
        uint64_t bits;
        if ((draw & 1u) == 0) {
            uint32_t a = pixel, b = sample, c = level, d = draw >> 1;
            philox4x32_10(a, b, c, d, k0, k1);
            bits = ((uint64_t)b << 32) | a;
            w2 = c; w3 = d;
        } else bits = ((uint64_t)w3 << 32) | w2;
        ++draw;
        return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    }
    RT_DEV double range(double lo, double hi) { return (hi - lo) * next_f64() + lo; }
    RT_DEV uint32_t index(uint32_t n) {
        uint32_t i = (uint32_t)(next_f64() * (double)n);
        return i < n ? i : n - 1;
    }
};

// ---- path signatures (parity tests only: rt_sample_radiance_sig) ----------------------------------
// Every DISCRETE decision a path takes -- which object a ray hit, which side, which checker branch or texel, which kind of
// scatter, reflect or refract, which light and whether its sample counted, visible or not, the .001 gate -- is an event
// (level, tag, value); a path's signature is the sum of the events' hashes (order-free, so the wavefront kernels and the
// oracle's recursion may meet the events in different orders).  Two paths with equal signatures followed the same branches,
// so their radiance is a smooth function of the same inputs and must agree to rounding; differing signatures explain a
// deviation (a direction moved by an ulp of libm across a discontinuity).  The oracle computes the same hash (orc_math.hpp).
enum { SIG_HIT = 1, SIG_FRONT = 2, SIG_TEX = 3, SIG_KIND = 4, SIG_DELTA = 5, SIG_LIGHT = 6, SIG_VIS = 7, SIG_GATE = 8, SIG_CUT = 9 };
RT_DEV unsigned long long sig_hash(uint32_t level, uint32_t tag, unsigned long long value) {
    unsigned long long x = ((unsigned long long)level << 56) ^ ((unsigned long long)tag << 48) ^ value;
    x += 0x9E3779B97F4A7C15ull;   // splitmix64 finaliser
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ---- math.rs ------------------------------------------------------------------------------------
RT_DEV d3 rand_in_unit_sphere(Rng& rng) {  // math.rs:12-19 (rejection; each lane loops until accepted)
    for (;;) {
        double a = rng.range(-1.0, 1.0), b = rng.range(-1.0, 1.0), c = rng.range(-1.0, 1.0);
        d3 p = mk(a, b, c);
        // length(p) < 1.0 (math.rs:16), decided without the square root: sqrt is monotone and correctly rounded, sqrt(1 - 2^-53) rounds to
        // 1 - 2^-53 and sqrt(1) = 1, so RN(sqrt(s)) < 1 <=> s < 1 for the same rounded s = (a a + b b) + c c (NaN: false both ways)
        if (dot(p, p) < 1.0) return p;
    }
}
RT_DEV d3 rand_cosine_hemisphere(Rng& rng) {  // math.rs:25-36
    double r1 = rng.next_f64();
    double r2 = rng.next_f64();
    double z = __builtin_sqrt(1.0 - r2);
    double phi = 2.0 * PI * r1;
    double sqrt_r2 = __builtin_sqrt(r2);
    double s, c;
    sincos(phi, &s, &c);
    return mk(c * sqrt_r2, s * sqrt_r2, z);
}
RT_DEV d3 rand_hemisphere(Rng& rng) {  // math.rs:39-45
    double z = rng.next_f64();
    double r = __builtin_sqrt(fmax(0.0, 1.0 - z * z));
    double phi = 2.0 * PI * rng.next_f64();
    double s, c;
    sincos(phi, &s, &c);
    return mk(r * c, r * s, z);
}
RT_DEV d3 reflect(d3 a, d3 b) { return a - (b * dot(a, b) * 2.0); }  // math.rs:51-53
RT_DEV d3 refract(d3 v, d3 n, double etai_over_etat) {              // math.rs:62-68
    double cos_theta = fmin(dot(-v, n), 1.0);
    d3 r_out_perp = (v + (n * cos_theta)) * etai_over_etat;
    d3 r_out_parallel = n * (-__builtin_sqrt(fabs(1.0 - length_squared(r_out_perp))));
    return r_out_perp + r_out_parallel;
}

struct Onb {  // onb.rs:9-26
    d3 u, v, w;
};
RT_DEV Onb onb_from_w(d3 w) {
    d3 a = fabs(w.x) > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
    Onb b;
    b.v = normalize(cross(w, a));
    b.u = cross(w, b.v);
    b.w = w;
    return b;
}
RT_DEV d3 onb_local(const Onb& b, d3 vec) { return ((vec.x * b.u) + (vec.y * b.v)) + (vec.z * b.w); }

}  // namespace rt
