// Math.hpp -- the slice of the reference's CUDATracer::Math the callers of the tracer use
// (reference: CUDATracer/Math/Vec.hpp, Math/Matrix.hpp:116-146,288-346, Math/Quaternion.hpp:79-81).
// Host-only, fp32, plain multiply/add in the order the reference writes them.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace CUDATracer {
namespace Math {

template <typename T> struct vec2_t {
    T x{}, y{};
    vec2_t() = default;
    vec2_t(T s) : x(s), y(s) {}
    vec2_t(T x_, T y_) : x(x_), y(y_) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};
using vec2f = vec2_t<float>;
using vec2d = vec2_t<double>;
using vec2i = vec2_t<int32_t>;
using vec2ui = vec2_t<uint32_t>;

struct vec3f {
    float x = 0, y = 0, z = 0;
    vec3f() = default;
    vec3f(float s) : x(s), y(s), z(s) {}
    vec3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
inline vec3f operator+(const vec3f& a, const vec3f& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3f operator-(const vec3f& a, const vec3f& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3f operator*(const vec3f& a, const vec3f& b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3f operator*(const vec3f& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3f operator*(float s, const vec3f& a) { return { s * a.x, s * a.y, s * a.z }; }
inline vec3f operator/(const vec3f& a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline vec3f operator-(const vec3f& a) { return { -a.x, -a.y, -a.z }; }
inline vec3f& operator+=(vec3f& a, const vec3f& b) { a = a + b; return a; }
inline vec3f cross(const vec3f& a, const vec3f& b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline float dot(const vec3f& a, const vec3f& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(const vec3f& a) { return std::sqrt(dot(a, a)); }
inline vec3f normalize(const vec3f& a) { return (a * 1.f) / std::sqrt(dot(a, a)); }

// row-major 4x4
struct mat4x4f {
    float data[16];
    mat4x4f() { std::memset(data, 0, sizeof data); }
    explicit mat4x4f(float diag) { std::memset(data, 0, sizeof data); data[0] = data[5] = data[10] = data[15] = diag; }
    mat4x4f(std::initializer_list<float> d) { int i = 0; for (float v : d) if (i < 16) data[i++] = v; for (; i < 16; i++) data[i] = 0; }
    float& at(unsigned r, unsigned c) { return data[r * 4 + c]; }
    float at(unsigned r, unsigned c) const { return data[r * 4 + c]; }
};
inline mat4x4f MatMul(const mat4x4f& M, const mat4x4f& N) {
    mat4x4f r;
    for (unsigned i = 0; i < 4; i++) for (unsigned j = 0; j < 4; j++) {
        float u = 0; for (unsigned t = 0; t < 4; t++) u += M.at(i, t) * N.at(t, j);
        r.at(i, j) = u;
    }
    return r;
}
inline vec3f MatMul(const mat4x4f& M, const vec3f& v, float w) {
    return { M.data[0] * v.x + M.data[1] * v.y + M.data[2] * v.z + M.data[3] * w,
             M.data[4] * v.x + M.data[5] * v.y + M.data[6] * v.z + M.data[7] * w,
             M.data[8] * v.x + M.data[9] * v.y + M.data[10] * v.z + M.data[11] * w };
}
inline mat4x4f MakeTranslate(const vec3f& t) { mat4x4f m(1.f); m.data[3] = t.x; m.data[7] = t.y; m.data[11] = t.z; return m; }
inline mat4x4f MakeScale(const vec3f& s) { mat4x4f m(1.f); m.data[0] = s.x; m.data[5] = s.y; m.data[10] = s.z; return m; }

struct Quaternion3f {
    float r = 1, i = 0, j = 0, k = 0;
    Quaternion3f() = default;
    Quaternion3f(float r_, const vec3f& v) : r(r_), i(v.x), j(v.y), k(v.z) {}
    static Quaternion3f rotate(const vec3f& u, float rad) { return Quaternion3f(std::cos(0.5f * rad), std::sin(0.5f * rad) * normalize(u)); }
};
inline mat4x4f MakeRotation(const Quaternion3f& q) {
    mat4x4f m;
    float qxx = q.i * q.i, qyy = q.j * q.j, qzz = q.k * q.k, qxz = q.i * q.k, qxy = q.i * q.j, qyz = q.j * q.k;
    float qwx = q.r * q.i, qwy = q.r * q.j, qwz = q.r * q.k;
    m.at(0, 0) = 1.f - 2.f * (qyy + qzz); m.at(1, 0) = 2.f * (qxy + qwz); m.at(2, 0) = 2.f * (qxz - qwy);
    m.at(0, 1) = 2.f * (qxy - qwz); m.at(1, 1) = 1.f - 2.f * (qxx + qzz); m.at(2, 1) = 2.f * (qyz + qwx);
    m.at(0, 2) = 2.f * (qxz + qwy); m.at(1, 2) = 2.f * (qyz - qwx); m.at(2, 2) = 1.f - 2.f * (qxx + qyy);
    m.at(3, 3) = 1.f;
    return m;
}

}  // namespace Math
}  // namespace CUDATracer
