// Host-side f64 vector/matrix types with the names and operation order of the `glam` crate
// (^0.17.3, renderer/Cargo.toml:11) as the reference uses them: DVec2, DVec3, DVec4, DMat4.
// Only what scene construction and BVH building need -- no intersection code lives on the host.
#pragma once
#include <cmath>
#include <cstdint>

namespace glam {

struct DVec2 {
    double x = 0, y = 0;
    DVec2() {}
    DVec2(double x_, double y_) : x(x_), y(y_) {}
    static DVec2 new_(double x, double y) { return DVec2(x, y); }
};

struct DVec3 {
    double x = 0, y = 0, z = 0;
    DVec3() {}
    DVec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static DVec3 new_(double x, double y, double z) { return DVec3(x, y, z); }
    static DVec3 splat(double v) { return DVec3(v, v, v); }
    static DVec3 ZERO() { return DVec3(0, 0, 0); }
    static DVec3 ONE() { return DVec3(1, 1, 1); }
    static DVec3 X() { return DVec3(1, 0, 0); }
    double operator[](size_t i) const { return i == 0 ? x : i == 1 ? y : z; }
    DVec3 operator+(const DVec3& o) const { return DVec3(x + o.x, y + o.y, z + o.z); }
    DVec3 operator-(const DVec3& o) const { return DVec3(x - o.x, y - o.y, z - o.z); }
    DVec3 operator-() const { return DVec3(-x, -y, -z); }
    DVec3 operator*(const DVec3& o) const { return DVec3(x * o.x, y * o.y, z * o.z); }
    DVec3 operator*(double s) const { return DVec3(x * s, y * s, z * s); }
    DVec3 operator/(double s) const { return DVec3(x / s, y / s, z / s); }
    double dot(const DVec3& o) const { return (x * o.x) + (y * o.y) + (z * o.z); }
    DVec3 cross(const DVec3& o) const { return DVec3(y * o.z - o.y * z, z * o.x - o.z * x, x * o.y - o.x * y); }
    double length_squared() const { return dot(*this); }
    double length() const { return std::sqrt(dot(*this)); }
    double length_recip() const { return 1.0 / length(); }
    DVec3 normalize() const { return *this * length_recip(); }
    DVec3 min(const DVec3& o) const { return DVec3(std::fmin(x, o.x), std::fmin(y, o.y), std::fmin(z, o.z)); }
    DVec3 max(const DVec3& o) const { return DVec3(std::fmax(x, o.x), std::fmax(y, o.y), std::fmax(z, o.z)); }
};
inline DVec3 operator*(double s, const DVec3& v) { return DVec3(s * v.x, s * v.y, s * v.z); }

struct DVec4 {
    double x = 0, y = 0, z = 0, w = 0;
    DVec4() {}
    DVec4(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
    static DVec4 from(const DVec3& v, double w) { return DVec4(v.x, v.y, v.z, w); }
    DVec3 xyz() const { return DVec3(x, y, z); }
    DVec4 operator+(const DVec4& o) const { return DVec4(x + o.x, y + o.y, z + o.z, w + o.w); }
    DVec4 operator-(const DVec4& o) const { return DVec4(x - o.x, y - o.y, z - o.z, w - o.w); }
    DVec4 operator*(const DVec4& o) const { return DVec4(x * o.x, y * o.y, z * o.z, w * o.w); }
    DVec4 operator*(double s) const { return DVec4(x * s, y * s, z * s, w * s); }
    double length() const { return std::sqrt((((x * x) + (y * y)) + (z * z)) + (w * w)); }
};

// Column-major: x_axis..w_axis are the columns.
struct DMat4 {
    DVec4 x_axis, y_axis, z_axis, w_axis;
    static DMat4 from_cols(DVec4 a, DVec4 b, DVec4 c, DVec4 d) { DMat4 m; m.x_axis = a; m.y_axis = b; m.z_axis = c; m.w_axis = d; return m; }
    static DMat4 IDENTITY() { return from_cols(DVec4(1, 0, 0, 0), DVec4(0, 1, 0, 0), DVec4(0, 0, 1, 0), DVec4(0, 0, 0, 1)); }
    static DMat4 from_scale(const DVec3& s) { return from_cols(DVec4(s.x, 0, 0, 0), DVec4(0, s.y, 0, 0), DVec4(0, 0, s.z, 0), DVec4(0, 0, 0, 1)); }
    static DMat4 from_translation(const DVec3& t) { return from_cols(DVec4(1, 0, 0, 0), DVec4(0, 1, 0, 0), DVec4(0, 0, 1, 0), DVec4(t.x, t.y, t.z, 1)); }
    static DMat4 from_rotation_y(double angle) {
        const double s = std::sin(angle), c = std::cos(angle);
        return from_cols(DVec4(c, 0, -s, 0), DVec4(0, 1, 0, 0), DVec4(s, 0, c, 0), DVec4(0, 0, 0, 1));
    }
    DVec4 mul_vec4(const DVec4& v) const {
        DVec4 res = x_axis * v.x;
        res = res + y_axis * v.y;
        res = res + z_axis * v.z;
        res = res + w_axis * v.w;
        return res;
    }
    DVec4 operator*(const DVec4& v) const { return mul_vec4(v); }
    DMat4 operator*(const DMat4& o) const { return from_cols(mul_vec4(o.x_axis), mul_vec4(o.y_axis), mul_vec4(o.z_axis), mul_vec4(o.w_axis)); }
    DMat4 operator*(double s) const { return from_cols(x_axis * s, y_axis * s, z_axis * s, w_axis * s); }
    DMat4 transpose() const {
        return from_cols(DVec4(x_axis.x, y_axis.x, z_axis.x, w_axis.x), DVec4(x_axis.y, y_axis.y, z_axis.y, w_axis.y),
                         DVec4(x_axis.z, y_axis.z, z_axis.z, w_axis.z), DVec4(x_axis.w, y_axis.w, z_axis.w, w_axis.w));
    }
    // General inverse by 2x2 sub-determinant cofactors, scaled by the reciprocal determinant
    // (the form glam 0.17 ships for its scalar DMat4).
    DMat4 inverse() const {
        const double a00 = x_axis.x, a01 = x_axis.y, a02 = x_axis.z, a03 = x_axis.w;
        const double a10 = y_axis.x, a11 = y_axis.y, a12 = y_axis.z, a13 = y_axis.w;
        const double a20 = z_axis.x, a21 = z_axis.y, a22 = z_axis.z, a23 = z_axis.w;
        const double a30 = w_axis.x, a31 = w_axis.y, a32 = w_axis.z, a33 = w_axis.w;
        const double k00 = a22 * a33 - a32 * a23, k02 = a12 * a33 - a32 * a13, k03 = a12 * a23 - a22 * a13;
        const double k04 = a21 * a33 - a31 * a23, k06 = a11 * a33 - a31 * a13, k07 = a11 * a23 - a21 * a13;
        const double k08 = a21 * a32 - a31 * a22, k10 = a11 * a32 - a31 * a12, k11 = a11 * a22 - a21 * a12;
        const double k12 = a20 * a33 - a30 * a23, k14 = a10 * a33 - a30 * a13, k15 = a10 * a23 - a20 * a13;
        const double k16 = a20 * a32 - a30 * a22, k18 = a10 * a32 - a30 * a12, k19 = a10 * a22 - a20 * a12;
        const double k20 = a20 * a31 - a30 * a21, k22 = a10 * a31 - a30 * a11, k23 = a10 * a21 - a20 * a11;
        const DVec4 f0(k00, k00, k02, k03), f1(k04, k04, k06, k07), f2(k08, k08, k10, k11);
        const DVec4 f3(k12, k12, k14, k15), f4(k16, k16, k18, k19), f5(k20, k20, k22, k23);
        const DVec4 v0(a10, a00, a00, a00), v1(a11, a01, a01, a01), v2(a12, a02, a02, a02), v3(a13, a03, a03, a03);
        const DVec4 i0 = (v1 * f0 - v2 * f1) + v3 * f2;
        const DVec4 i1 = (v0 * f0 - v2 * f3) + v3 * f4;
        const DVec4 i2 = (v0 * f1 - v1 * f3) + v3 * f5;
        const DVec4 i3 = (v0 * f2 - v1 * f4) + v2 * f5;
        const DVec4 sa(1.0, -1.0, 1.0, -1.0), sb(-1.0, 1.0, -1.0, 1.0);
        const DMat4 adj = from_cols(i0 * sa, i1 * sb, i2 * sa, i3 * sb);
        const DVec4 row0(adj.x_axis.x, adj.y_axis.x, adj.z_axis.x, adj.w_axis.x);
        const DVec4 d = x_axis * row0;
        const double det = ((d.x + d.y) + d.z) + d.w;
        return adj * (1.0 / det);
    }
};

}  // namespace glam
