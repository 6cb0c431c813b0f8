// glsl_shim.hpp -- TEST TOOLING ("Oracle-A").
//
// A minimal GLSL vocabulary in C++ so that the reference's shader TEXT (read from
// /root/reference at generation time, mechanically edited by oracle/gen_oracle_a.py, never
// copied into this repo) can be executed on the CPU to produce known-answer vectors.
// Everything is IEEE fp32: float overloads call sinf/cosf/..., no silent widening.
// Texture lookups are NOT the reference's (it relies on the Vulkan driver): they are routed
// to callbacks the driver program installs (analytic environment, or the oracle's sampler).
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

typedef unsigned int uint;

#define ARITH(T) typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0

struct uvec2 { uint x, y; };
struct ivec2 { int x, y; ivec2() : x(0), y(0) {} ivec2(int a, int b) : x(a), y(b) {} explicit ivec2(uvec2 u) : x(int(u.x)), y(int(u.y)) {} };
struct vec2 {
    union { float x; float r; float s; };
    union { float y; float g; float t; };
    vec2() : x(0), y(0) {}
    template <class A, ARITH(A)> explicit vec2(A a) : x(float(a)), y(float(a)) {}
    template <class A, class B, ARITH(A), ARITH(B)> vec2(A a, B b) : x(float(a)), y(float(b)) {}
    explicit vec2(uvec2 u) : x(float(u.x)), y(float(u.y)) {}
    explicit vec2(ivec2 u) : x(float(u.x)), y(float(u.y)) {}
    vec2 xy() const { return *this; }
};
struct vec3 {
    union { float x; float r; };
    union { float y; float g; };
    union { float z; float b; };
    vec3() : x(0), y(0), z(0) {}
    template <class A, ARITH(A)> explicit vec3(A a) : x(float(a)), y(float(a)), z(float(a)) {}
    template <class A, class B, class C, ARITH(A), ARITH(B), ARITH(C)> vec3(A a, B b_, C c) : x(float(a)), y(float(b_)), z(float(c)) {}
    template <class C, ARITH(C)> vec3(vec2 v, C c) : x(v.x), y(v.y), z(float(c)) {}
    vec3 xyz() const { return *this; }
    vec3 rgb() const { return *this; }
    vec2 xy() const { return vec2(x, y); }
};
struct vec4 {
    union { float x; float r; };
    union { float y; float g; };
    union { float z; float b; };
    union { float w; float a; };
    vec4() : x(0), y(0), z(0), w(0) {}
    template <class A, ARITH(A)> explicit vec4(A a_) : x(float(a_)), y(float(a_)), z(float(a_)), w(float(a_)) {}
    template <class A, class B, class C, class D, ARITH(A), ARITH(B), ARITH(C), ARITH(D)>
    vec4(A a_, B b_, C c, D d) : x(float(a_)), y(float(b_)), z(float(c)), w(float(d)) {}
    template <class D, ARITH(D)> vec4(vec3 v, D d) : x(v.x), y(v.y), z(v.z), w(float(d)) {}
    template <class C, class D, ARITH(C), ARITH(D)> vec4(vec2 v, C c, D d) : x(v.x), y(v.y), z(float(c)), w(float(d)) {}
    vec3 xyz() const { return vec3(x, y, z); }
    vec3 rgb() const { return vec3(x, y, z); }
    vec2 xy() const { return vec2(x, y); }
};
struct uvec3 { uint x, y, z; uvec2 xy() const { return uvec2{x, y}; } uvec2 yz() const { return uvec2{y, z}; } };
struct ivec3 {
    int x, y, z;
    ivec3() : x(0), y(0), z(0) {}
    ivec3(int a, int b, int c) : x(a), y(b), z(c) {}
    ivec3(int a, uvec2 bc) : x(a), y(int(bc.x)), z(int(bc.y)) {}
    explicit ivec3(uvec3 u) : x(int(u.x)), y(int(u.y)), z(int(u.z)) {}
    ivec3 zxy() const { return ivec3(z, x, y); }
    ivec3 yzx() const { return ivec3(y, z, x); }
};
inline ivec3 operator+(ivec3 a, ivec3 b) { return ivec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline ivec3 operator*(int s, ivec3 a) { return ivec3(s * a.x, s * a.y, s * a.z); }

// ---- operators --------------------------------------------------------------------------
#define VOPS2(OP) \
    inline vec2 operator OP(vec2 a, vec2 b) { return vec2(a.x OP b.x, a.y OP b.y); } \
    template <class S, ARITH(S)> inline vec2 operator OP(vec2 a, S s) { return vec2(a.x OP float(s), a.y OP float(s)); } \
    template <class S, ARITH(S)> inline vec2 operator OP(S s, vec2 a) { return vec2(float(s) OP a.x, float(s) OP a.y); }
#define VOPS3(OP) \
    inline vec3 operator OP(vec3 a, vec3 b) { return vec3(a.x OP b.x, a.y OP b.y, a.z OP b.z); } \
    template <class S, ARITH(S)> inline vec3 operator OP(vec3 a, S s) { return vec3(a.x OP float(s), a.y OP float(s), a.z OP float(s)); } \
    template <class S, ARITH(S)> inline vec3 operator OP(S s, vec3 a) { return vec3(float(s) OP a.x, float(s) OP a.y, float(s) OP a.z); }
#define VOPS4(OP) \
    inline vec4 operator OP(vec4 a, vec4 b) { return vec4(a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w); } \
    template <class S, ARITH(S)> inline vec4 operator OP(vec4 a, S s) { return vec4(a.x OP float(s), a.y OP float(s), a.z OP float(s), a.w OP float(s)); } \
    template <class S, ARITH(S)> inline vec4 operator OP(S s, vec4 a) { return vec4(float(s) OP a.x, float(s) OP a.y, float(s) OP a.z, float(s) OP a.w); }
VOPS2(+) VOPS2(-) VOPS2(*) VOPS2(/)
VOPS3(+) VOPS3(-) VOPS3(*) VOPS3(/)
VOPS4(+) VOPS4(-) VOPS4(*) VOPS4(/)
inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec4 operator-(vec4 a) { return vec4(-a.x, -a.y, -a.z, -a.w); }
#define VASSIGN(T, OP) \
    inline T& operator OP##=(T& a, T b) { a = a OP b; return a; } \
    template <class S, ARITH(S)> inline T& operator OP##=(T& a, S s) { a = a OP s; return a; }
VASSIGN(vec2, +) VASSIGN(vec2, -) VASSIGN(vec2, *) VASSIGN(vec2, /)
VASSIGN(vec3, +) VASSIGN(vec3, -) VASSIGN(vec3, *) VASSIGN(vec3, /)
VASSIGN(vec4, +) VASSIGN(vec4, -) VASSIGN(vec4, *) VASSIGN(vec4, /)
inline bool operator==(vec2 a, vec2 b) { return a.x == b.x && a.y == b.y; }
inline bool operator!=(vec2 a, vec2 b) { return !(a == b); }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(vec3 a, vec3 b) { return !(a == b); }

// ---- scalar builtins (fp32 only) ----------------------------------------------------------
#if defined(SHIM_DET_TRIG)   // voxel-GI variant: the fixed fp32 polynomials shared with the oracle and the kernel (pbr_oracle.h, N4)
extern "C" float orc_sinf_det(float), orc_cosf_det(float), orc_acosf_det(float);
inline float sin(float x) { return orc_sinf_det(x); }
inline float cos(float x) { return orc_cosf_det(x); }
#else
inline float sin(float x) { return sinf(x); }
inline float cos(float x) { return cosf(x); }
#endif
inline float tan(float x) { return tanf(x); }
#if defined(SHIM_DET_TRIG)
inline float acos(float x) { return orc_acosf_det(x); }
#else
inline float acos(float x) { return acosf(x); }
#endif
inline float exp(float x) { return expf(x); }
inline float sqrt(float x) { return sqrtf(x); }
inline float floor(float x) { return floorf(x); }
inline float ceil(float x) { return ceilf(x); }
inline float abs(float x) { return fabsf(x); }
inline float fract(float x) { return x - floorf(x); }
template <class B, ARITH(B)> inline float pow(float a, B b) { return powf(a, float(b)); }
template <class A, class B, ARITH(A), ARITH(B)> inline float min(A a, B b) { return fminf(float(a), float(b)); }
template <class A, class B, ARITH(A), ARITH(B)> inline float max(A a, B b) { return fmaxf(float(a), float(b)); }
template <class B, class C, ARITH(B), ARITH(C)> inline float clamp(float x, B lo, C hi) { return fminf(fmaxf(x, float(lo)), float(hi)); }
template <class T, ARITH(T)> inline float mix(float a, float b, T t) { return a * (1.0f - float(t)) + b * float(t); }

// ---- vector builtins ----------------------------------------------------------------------
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(vec2 a) { return sqrtf(dot(a, a)); }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline float length(vec4 a) { return sqrtf(dot(a, a)); }
inline vec2 normalize(vec2 a) { return a / length(a); }
inline vec3 normalize(vec3 a) { return a / length(a); }
inline vec2 min(vec2 a, vec2 b) { return vec2(fminf(a.x, b.x), fminf(a.y, b.y)); }
inline vec2 max(vec2 a, vec2 b) { return vec2(fmaxf(a.x, b.x), fmaxf(a.y, b.y)); }
inline vec2 clamp(vec2 v, vec2 lo, vec2 hi) { return min(max(v, lo), hi); }
inline vec3 min(vec3 a, vec3 b) { return vec3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline vec3 max(vec3 a, vec3 b) { return vec3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
inline vec4 max(vec4 a, vec4 b) { return vec4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
inline vec3 clamp(vec3 v, vec3 lo, vec3 hi) { return min(max(v, lo), hi); }
template <class B, class C, ARITH(B), ARITH(C)> inline vec3 clamp(vec3 v, B lo, C hi) { return clamp(v, vec3(float(lo)), vec3(float(hi))); }
inline vec4 clamp(vec4 v, vec4 lo, vec4 hi) { return vec4(clamp(v.x, lo.x, hi.x), clamp(v.y, lo.y, hi.y), clamp(v.z, lo.z, hi.z), clamp(v.w, lo.w, hi.w)); }
template <class T, ARITH(T)> inline vec3 mix(vec3 a, vec3 b, T t) { return a * (1.0f - float(t)) + b * float(t); }
template <class T, ARITH(T)> inline vec4 mix(vec4 a, vec4 b, T t) { return a * (1.0f - float(t)) + b * float(t); }
inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
inline vec3 sqrt(vec3 a) { return vec3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
inline vec3 abs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
inline vec2 fract(vec2 a) { return vec2(fract(a.x), fract(a.y)); }
inline vec3 fract(vec3 a) { return vec3(fract(a.x), fract(a.y), fract(a.z)); }
inline vec2 floor(vec2 a) { return vec2(floorf(a.x), floorf(a.y)); }
inline vec3 floor(vec3 a) { return vec3(floorf(a.x), floorf(a.y), floorf(a.z)); }

struct bvec2 { bool x, y; };
inline bvec2 notEqual(vec2 a, vec2 b) { return bvec2{a.x != b.x, a.y != b.y}; }
inline bool any(bvec2 b) { return b.x || b.y; }
inline vec3 pow(vec3 a, vec3 b) { return vec3(powf(a.x, b.x), powf(a.y, b.y), powf(a.z, b.z)); }
template <class B, ARITH(B)> inline vec3 max(vec3 a, B b) { return max(a, vec3(float(b))); }
template <class B, class C, ARITH(B), ARITH(C)> inline vec2 clamp(vec2 v, B lo, C hi) { return clamp(v, vec2(float(lo)), vec2(float(hi))); }

// ---- matrices (column-major, like GLSL) ---------------------------------------------------
struct mat4 {
    float m[4][4]; // m[col][row]
    mat4() { for (int c = 0; c < 4; c++) for (int r_ = 0; r_ < 4; r_++) m[c][r_] = 0; }
    mat4(vec4 c0, vec4 c1, vec4 c2, vec4 c3) {
        vec4 cs[4] = {c0, c1, c2, c3};
        for (int c = 0; c < 4; c++) { m[c][0] = cs[c].x; m[c][1] = cs[c].y; m[c][2] = cs[c].z; m[c][3] = cs[c].w; }
    }
};
inline vec4 operator*(const mat4& M, vec4 v) {
    vec4 o;
    o.x = ((M.m[0][0] * v.x + M.m[1][0] * v.y) + M.m[2][0] * v.z) + M.m[3][0] * v.w;
    o.y = ((M.m[0][1] * v.x + M.m[1][1] * v.y) + M.m[2][1] * v.z) + M.m[3][1] * v.w;
    o.z = ((M.m[0][2] * v.x + M.m[1][2] * v.y) + M.m[2][2] * v.z) + M.m[3][2] * v.w;
    o.w = ((M.m[0][3] * v.x + M.m[1][3] * v.y) + M.m[2][3] * v.z) + M.m[3][3] * v.w;
    return o;
}
struct mat3 {
    vec3 c[3];
    mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
};
inline vec3 operator*(const mat3& M, vec3 v) { return (M.c[0] * v.x + M.c[1] * v.y) + M.c[2] * v.z; }

// ---- resources ----------------------------------------------------------------------------
// Each texture is an id; combined-sampler constructors return a typed handle carrying the id.
struct sampler { int id; };
struct samplerShadow { int id; };
struct textureCube { int id; };
struct texture2D { int id; };
struct texture3D { int id; };
struct imageCube { int size; float* data; };   // data: [6][size][size][4]
struct image2D { int size; float* data; };     // data: [size][size][4]
struct image3D { int w, h, d; uint16_t* data; };   // RGBA16F bit patterns, data: [d][h][w][4]
struct H_cube { int id; };
struct H_2d { int id; };
struct H_3d { int id; };
struct H_shadow { int id; };
inline H_cube samplerCube(textureCube t, sampler) { return H_cube{t.id}; }
#if defined(SHIM_SAMPLER_IDS)   // sampler id 101 = SAMPLER_NEAREST_CLAMP: reported to the callback as texture id + 1000
inline H_2d sampler2D(texture2D t, sampler s) { return H_2d{t.id + (s.id == 101 ? 1000 : 0)}; }
#else
inline H_2d sampler2D(texture2D t, sampler) { return H_2d{t.id}; }
#endif
inline H_3d sampler3D(texture3D t, sampler) { return H_3d{t.id}; }
inline H_shadow sampler2DShadow(texture2D t, samplerShadow) { return H_shadow{t.id}; }

// callbacks installed by the driver program
extern vec4 (*shim_cube_lookup)(int id, vec3 dir, float lod);
extern vec4 (*shim_tex2d_lookup)(int id, vec2 uv, float lod);
inline vec4 textureLod(H_cube h, vec3 d, float lod) { return shim_cube_lookup(h.id, d, lod); }
inline vec4 textureLod(H_2d h, vec2 uv, float lod) { return shim_tex2d_lookup(h.id, uv, lod); }
inline vec4 texture(H_2d h, vec2 uv) { return shim_tex2d_lookup(h.id, uv, 0.0f); }
extern ivec2 (*shim_tex2d_size)(int id);
inline ivec2 textureSize(H_2d h, int) { return shim_tex2d_size(h.id); }
extern vec4 (*shim_tex3d_lookup)(int id, vec3 p);               // NULL: LIGHTGRID == 0
inline vec4 texture(H_3d h, vec3 p) { return shim_tex3d_lookup ? shim_tex3d_lookup(h.id, p) : vec4(0.0f); }
extern float (*shim_shadow_lookup)(int id, vec3 uv_ref);       // NULL: shadow map == fully lit
inline float texture(H_shadow h, vec3 p) { return shim_shadow_lookup ? shim_shadow_lookup(h.id, p) : 1.0f; }

inline ivec2 imageSize(const imageCube& im) { return ivec2(im.size, im.size); }
inline ivec2 imageSize(const image2D& im) { return ivec2(im.size, im.size); }
inline void imageStore(imageCube& im, ivec3 p, vec4 v) {
    float* o = im.data + ((size_t(p.z) * im.size + p.y) * im.size + p.x) * 4;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
inline void imageStore(image2D& im, ivec2 p, vec4 v) {
    float* o = im.data + (size_t(p.y) * im.size + p.x) * 4;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// RGBA16F storage image: loads widen, stores round to nearest-even (F16C hardware conversion, so this
// is independent of the oracle's own fp16 code; the Vulkan driver's store rounding is not pinned by the reference)
#if defined(__F16C__)
typedef float shim_v4sf __attribute__((vector_size(16)));
typedef short shim_v8hi __attribute__((vector_size(16)));
inline float shim_h2f(uint16_t h) { shim_v8hi v = {short(h), 0, 0, 0, 0, 0, 0, 0}; return __builtin_ia32_vcvtph2ps(v)[0]; }
inline uint16_t shim_f2h(float f) { shim_v4sf v = {f, 0, 0, 0}; return uint16_t(__builtin_ia32_vcvtps2ph(v, 0)[0]); }   // imm 0 = round to nearest even
inline vec4 imageLoad(const image3D& im, ivec3 p) {
    const uint16_t* v = im.data + ((size_t(p.z) * im.h + p.y) * im.w + p.x) * 4;
    return vec4(shim_h2f(v[0]), shim_h2f(v[1]), shim_h2f(v[2]), shim_h2f(v[3]));
}
inline void imageStore(image3D& im, ivec3 p, vec4 v) {
    uint16_t* o = im.data + ((size_t(p.z) * im.h + p.y) * im.w + p.x) * 4;
    o[0] = shim_f2h(v.x); o[1] = shim_f2h(v.y); o[2] = shim_f2h(v.z); o[3] = shim_f2h(v.w);
}
#endif

static uvec3 gl_GlobalInvocationID;
static vec4 gl_FragCoord;
