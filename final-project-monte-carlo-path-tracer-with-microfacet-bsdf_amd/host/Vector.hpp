// Minimal float vectors for the host side (the reference uses Eigen's Vector3f/Vector2f/Matrix3f for the same job).
// Only what scene assembly needs: storage, +, -, scalar *, dot/cross/normalized with Eigen's evaluation order.
#pragma once
#include <cmath>

struct Vector2f {
    float v[2] = {0.f, 0.f};
    Vector2f() = default;
    Vector2f(float x, float y) : v{x, y} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float operator[](int i) const { return v[i]; }
};

struct Vector3f {
    float v[3] = {0.f, 0.f, 0.f};
    Vector3f() = default;
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    static Vector3f Zero() { return Vector3f(0.f, 0.f, 0.f); }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float operator[](int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    const float *data() const { return v; }
    Vector3f operator+(const Vector3f &o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3f operator-(const Vector3f &o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3f operator*(float s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    float dot(const Vector3f &o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
    Vector3f cross(const Vector3f &o) const {
        return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
    }
    float norm() const { return std::sqrt(dot(*this)); }
    Vector3f normalized() const {
        const float z = dot(*this);
        if (z > 0.f) {
            const float s = std::sqrt(z);
            return {v[0] / s, v[1] / s, v[2] / s};
        }
        return *this;
    }
};
inline Vector3f operator*(float s, const Vector3f &a) { return a * s; }

struct Matrix3f {  // row-major storage, column accessors as the reference uses them (Camera.hpp:21-23)
    float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    static Matrix3f Identity() { return Matrix3f(); }
    void setCol(int j, const Vector3f &c) {
        for (int i = 0; i < 3; ++i) m[3 * i + j] = c[i];
    }
    float operator()(int i, int j) const { return m[3 * i + j]; }
};
