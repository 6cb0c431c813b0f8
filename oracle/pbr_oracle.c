/*
 * pbr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See pbr_oracle.h.
 *
 * Scalar fp32 restatement of the reference's hot-path arithmetic.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference).  Build with
 *   gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp
 * so that every operation is a separately rounded IEEE fp32 op in source order (GLSL evaluates
 * left to right; literals are fp32).
 *
 * Parity status: Monte-Carlo maths pinned by tests/golden/oracle_a_* (the reference shader text
 * executed on the CPU); texture filtering "parity unpinned" (Vulkan leaves it to the driver,
 * the reference has no tests) and defined here: exact fp32 weights, seamless cube edges, corner
 * = mean of the three existing texels, 2:1 blit = ((a+b)+c+d)*0.25.
 */
#include "pbr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846f            /* shaders: #define PI 3.14159265358979323846 */
#define ORC_GOLDEN_RATIO 1.61803398875f           /* shaders: #define GOLDEN_RATIO 1.61803398875 */

static int g_threads = 0;

void orc_set_threads(int n) { g_threads = n; }
int orc_get_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
#ifdef _OPENMP
#define ORC_NT() (g_threads > 0 ? g_threads : omp_get_max_threads())
#else
#define ORC_NT() 1
#endif

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers with GLSL semantics                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* GLSL cross(x,y) = (x1*y2 - y1*x2, x2*y0 - y2*x0, x0*y1 - y0*x1) */
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 v3_normalize(v3 a) {
    float len = sqrtf(v3_dot(a, a));
    return V3(a.x / len, a.y / len, a.z / len);
}
static inline float fractf_(float x) { return x - floorf(x); }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; } /* GLSL mix */
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }       /* texture filter */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Rotate(): gen_prefiltered_env_map.glsl:68-71 (same text in gen_irradiance_map.glsl:68-71,
 * gen_brdf_integration_map.glsl:61-64):  cos*(v - dot(v,n)*n) + sin*cross(n,v) + dot(v,n)*n   */
static inline v3 rotate_cs(v3 v, v3 n, float c, float s) {
    float d = v3_dot(v, n);
    v3 a = v3_scale(v3_sub(v, v3_scale(n, d)), c);
    v3 b = v3_scale(v3_cross(n, v), s);
    v3 cc = v3_scale(n, d);
    return v3_add(v3_add(a, b), cc);
}

/* ------------------------------------------------------------------------------------------ */
/* fp16                                                                                        */
/* ------------------------------------------------------------------------------------------ */
uint16_t orc_f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                          /* < 2^-25 (+1ulp) -> 0 */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; } else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t h;
    if (hexp == 0) h = q;                       /* subnormal (q may carry into exponent 1: correct) */
    else h = ((hexp - 1u) << 10) + q;           /* q has the implicit bit at 0x400: adds 1 to exponent */
    return (uint16_t)(sign | h);
}

float orc_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            float v = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
            memcpy(&x, &v, 4); x |= sign;
        }
    } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

/* ------------------------------------------------------------------------------------------ */
/* A1: Radiance RGBE decode.  Follows third_party/stb_image.h:7130-7155 (stbi__hdr_convert)    */
/* and :7157-7286 (stbi__hdr_load) as called by asset_import.cpp:19 with req_comp = 4.         */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t* p; size_t n, pos; } rd_t;
static int rd_get8(rd_t* r) { return r->pos < r->n ? r->p[r->pos++] : 0; }
static int rd_eof(const rd_t* r) { return r->pos >= r->n; }

static void rd_token(rd_t* r, char* buf, int cap) {
    int len = 0;
    char c = (char)rd_get8(r);
    while (!rd_eof(r) && c != '\n') {
        buf[len++] = c;
        if (len == cap - 1) {
            while (!rd_eof(r) && rd_get8(r) != '\n') {}
            break;
        }
        c = (char)rd_get8(r);
    }
    buf[len] = 0;
}

static void rgbe_convert(float* out, const uint8_t* in) {
    if (in[3] != 0) {
        float f1 = (float)ldexp(1.0f, (int)in[3] - (int)(128 + 8));
        out[0] = in[0] * f1; out[1] = in[1] * f1; out[2] = in[2] * f1; out[3] = 1.0f;
    } else {
        out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f;
    }
}

int orc_rgbe_decode(const uint8_t* bytes, size_t n, int* w, int* h, float* out) {
    rd_t r = {bytes, n, 0};
    char buf[1024];
    rd_token(&r, buf, sizeof buf);
    if (strcmp(buf, "#?RADIANCE") != 0 && strcmp(buf, "#?RGBE") != 0) return 1;
    int valid = 0;
    for (;;) {
        rd_token(&r, buf, sizeof buf);
        if (buf[0] == 0) break;
        if (strcmp(buf, "FORMAT=32-bit_rle_rgbe") == 0) valid = 1;
    }
    if (!valid) return 2;
    rd_token(&r, buf, sizeof buf);
    char* t = buf;
    if (strncmp(t, "-Y ", 3)) return 3;
    t += 3;
    int height = (int)strtol(t, &t, 10);
    while (*t == ' ') ++t;
    if (strncmp(t, "+X ", 3)) return 3;
    t += 3;
    int width = (int)strtol(t, NULL, 10);
    if (width <= 0 || height <= 0) return 4;
    *w = width; *h = height;
    if (!out) return 0;

    int flat = (width < 8 || width >= 32768);
    int i = 0, j = 0;
    if (!flat) {
        uint8_t* scan = (uint8_t*)malloc((size_t)width * 4);
        for (j = 0; j < height; ++j) {
            int c1 = rd_get8(&r), c2 = rd_get8(&r), len = rd_get8(&r);
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                /* not RLE: this is pixel (0,0); the rest of the file is flat (stb restarts at j=0,i=1) */
                uint8_t px[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)rd_get8(&r)};
                rgbe_convert(out, px);
                flat = 1; i = 1; j = 0;
                break;
            }
            len = (len << 8) | rd_get8(&r);
            if (len != width) { free(scan); return 5; }
            for (int k = 0; k < 4; ++k) {
                int nleft; i = 0;
                while ((nleft = width - i) > 0) {
                    int count = rd_get8(&r);
                    if (count > 128) {
                        int value = rd_get8(&r); count -= 128;
                        if (count == 0 || count > nleft) { free(scan); return 6; }
                        for (int z = 0; z < count; ++z) scan[i++ * 4 + k] = (uint8_t)value;
                    } else {
                        if (count == 0 || count > nleft) { free(scan); return 6; }
                        for (int z = 0; z < count; ++z) scan[i++ * 4 + k] = (uint8_t)rd_get8(&r);
                    }
                }
            }
            for (i = 0; i < width; ++i) rgbe_convert(out + ((size_t)j * width + i) * 4, scan + i * 4);
        }
        free(scan);
        if (!flat) return 0;
    }
    for (; j < height; ++j) {
        for (; i < width; ++i) {
            uint8_t px[4];
            px[0] = (uint8_t)rd_get8(&r); px[1] = (uint8_t)rd_get8(&r);
            px[2] = (uint8_t)rd_get8(&r); px[3] = (uint8_t)rd_get8(&r);
            rgbe_convert(out + ((size_t)j * width + i) * 4, px);
        }
        i = 0;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* A2: mip pyramid.  Count: src/gpu/gpu_vulkan.c:1344-1351.  Chain: :1458-1483 issues one      */
/* linear vkCmdBlitImage per (level, face) from the whole level l-1 (n x n) into the whole     */
/* level l (floor(n/2) x floor(n/2)), :2786-2826.  Exact 2:1 (n even) => 2x2 box mean per face.*/
/* n odd (faces that are not a power of two: asset_import.cpp:21 only asserts y == 6x): the    */
/* blit is a genuine linear resample with scale n / floor(n/2).  The reference leaves its      */
/* arithmetic to the driver (UNPINNED, like the box rule); defined here after the Vulkan spec  */
/* (vkCmdBlitImage, unnormalised linear filtering, clamp to edge), in fp32, every op rounded:  */
/*   u = (x + 0.5) * (ns / nd);  t = u - 0.5;  i0 = floor(t);  a = t - i0;  taps i0, i0 + 1    */
/*   clamped to [0, ns - 1];  same along y with weight b;                                       */
/*   out = (t00 (1 - a) + t10 a) (1 - b) + (t01 (1 - a) + t11 a) b                              */
/* ------------------------------------------------------------------------------------------ */
int orc_mip_count(int w, int h) {
    int s = w < h ? w : h, c = 1;
    while (s > 1) { s /= 2; c++; }
    return c;
}
static inline int level_size(int W, int l) { int n = W >> l; return n < 1 ? 1 : n; }
size_t orc_level_offset(int W, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) { size_t n = (size_t)level_size(W, l); off += 6 * n * n * 4; }
    return off;
}
size_t orc_pyramid_floats(int W) { return orc_level_offset(W, orc_mip_count(W, W)); }

/* one linear blit of all layers: src [layers][ns_h][ns_w][4] -> dst [layers][nd_h][nd_w][4] (whole subresources) */
void orc_blit_linear(const float* src, int ns_w, int ns_h, float* dst, int nd_w, int nd_h, int layers) {
    const float sx = (float)ns_w / (float)nd_w, sy = (float)ns_h / (float)nd_h;
    #pragma omp parallel for collapse(2) num_threads(ORC_NT()) if ((long)nd_w * nd_h >= 4096)
    for (int f = 0; f < layers; ++f)
        for (int y = 0; y < nd_h; ++y) {
            const float tv = ((float)y + 0.5f) * sy - 0.5f;
            const float fv = floorf(tv), b = tv - fv;
            int j0 = (int)fv, j1 = j0 + 1;
            j0 = j0 < 0 ? 0 : (j0 > ns_h - 1 ? ns_h - 1 : j0); j1 = j1 < 0 ? 0 : (j1 > ns_h - 1 ? ns_h - 1 : j1);
            for (int x = 0; x < nd_w; ++x) {
                const float tu = ((float)x + 0.5f) * sx - 0.5f;
                const float fu = floorf(tu), a = tu - fu;
                int i0 = (int)fu, i1 = i0 + 1;
                i0 = i0 < 0 ? 0 : (i0 > ns_w - 1 ? ns_w - 1 : i0); i1 = i1 < 0 ? 0 : (i1 > ns_w - 1 ? ns_w - 1 : i1);
                const float* t00 = src + (((size_t)f * ns_h + j0) * ns_w + i0) * 4;
                const float* t10 = src + (((size_t)f * ns_h + j0) * ns_w + i1) * 4;
                const float* t01 = src + (((size_t)f * ns_h + j1) * ns_w + i0) * 4;
                const float* t11 = src + (((size_t)f * ns_h + j1) * ns_w + i1) * 4;
                float* o = dst + (((size_t)f * nd_h + y) * nd_w + x) * 4;
                for (int k = 0; k < 4; ++k) {
                    const float top = t00[k] * (1.0f - a) + t10[k] * a;
                    const float bot = t01[k] * (1.0f - a) + t11[k] * a;
                    o[k] = top * (1.0f - b) + bot * b;
                }
            }
        }
}

void orc_build_pyramid(float* pyr, int W) {
    int levels = orc_mip_count(W, W);
    for (int l = 1; l < levels; ++l) {
        int ns = level_size(W, l - 1), nd = level_size(W, l);
        const float* src = pyr + orc_level_offset(W, l - 1);
        float* dst = pyr + orc_level_offset(W, l);
        if (ns != 2 * nd) { orc_blit_linear(src, ns, ns, dst, nd, nd, 6); continue; }      /* odd level: genuine linear resample */
        #pragma omp parallel for collapse(2) num_threads(ORC_NT()) if (nd >= 64)
        for (int f = 0; f < 6; ++f)
            for (int y = 0; y < nd; ++y)
                for (int x = 0; x < nd; ++x) {
                    const float* a = src + (((size_t)f * ns + 2 * y) * ns + 2 * x) * 4;
                    const float* b = a + 4;
                    const float* c = a + (size_t)ns * 4;
                    const float* d = c + 4;
                    float* o = dst + (((size_t)f * nd + y) * nd + x) * 4;
                    for (int k = 0; k < 4; ++k) o[k] = (((a[k] + b[k]) + c[k]) + d[k]) * 0.25f;
                }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* A3: CubemapSampleDirFromFaceUV  gen_prefiltered_env_map.glsl:11-66                          */
/* ------------------------------------------------------------------------------------------ */
static inline v3 face_dir(int face, float u, float v) {
    float sc = 2 * (u - 0.5f);
    float tc = 2 * (v - 0.5f);
    v3 r = V3(0, 0, 0);
    switch (face) {
    case 0: r.z = -sc; r.y = -tc; r.x = +1; break;
    case 1: r.z = sc;  r.y = -tc; r.x = -1; break;
    case 2: r.x = sc;  r.z = tc;  r.y = +1; break;
    case 3: r.x = sc;  r.z = -tc; r.y = -1; break;
    case 4: r.x = sc;  r.y = -tc; r.z = +1; break;
    case 5: r.x = -sc; r.y = -tc; r.z = -1; break;
    }
    return v3_normalize(r);
}
void orc_face_dir(int face, float u, float v, float out[3]) {
    v3 r = face_dir(face, u, v); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------------ */
/* Cube sampling.  The reference delegates to the Vulkan driver through a linear/clamp sampler  */
/* (gpu_vulkan.c:613-634, 935-943); the face-selection table is the Vulkan one quoted in        */
/* gen_prefiltered_env_map.glsl:12-23.  Definition used here: major axis = largest  */
/* |component| with ties z > y > x; s = 0.5*sc/|rc| + 0.5; u = s*n; taps floor(u-.5), +1 with   */
/* weight frac(u-.5); out-of-face taps come from the adjacent face; corner = mean of 3.         */
/* ------------------------------------------------------------------------------------------ */
static inline void dir_to_face(v3 d, int* face, float* sc, float* tc, float* ma) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (az >= ax && az >= ay) {
        if (d.z < 0) { *face = 5; *sc = -d.x; *tc = -d.y; } else { *face = 4; *sc = d.x; *tc = -d.y; }
        *ma = az;
    } else if (ay >= ax) {
        if (d.y < 0) { *face = 3; *sc = d.x; *tc = -d.z; } else { *face = 2; *sc = d.x; *tc = d.z; }
        *ma = ay;
    } else {
        if (d.x < 0) { *face = 1; *sc = d.z; *tc = -d.y; } else { *face = 0; *sc = -d.z; *tc = -d.y; }
        *ma = ax;
    }
}

/* Integer edge fold.  Texel (i,j) of an n*n face sits at sc = (2i+1-n)/n, tc = (2j+1-n)/n; a
 * tap one texel outside an edge lies at |coord| = (n+1)/n.  Folding the cube plane over that
 * edge puts it on the adjacent face, one half-texel inside it, same along-edge coordinate. */
int orc_cube_neighbor(int face, int n, int i, int j, int* ni, int* nj) {
    int sc = 2 * i + 1 - n, tc = 2 * j + 1 - n;   /* units of 1/n */
    int p[3];
    switch (face) {
    case 0: p[0] = n;   p[1] = -tc; p[2] = -sc; break;
    case 1: p[0] = -n;  p[1] = -tc; p[2] = sc;  break;
    case 2: p[0] = sc;  p[1] = n;   p[2] = tc;  break;
    case 3: p[0] = sc;  p[1] = -n;  p[2] = -tc; break;
    case 4: p[0] = sc;  p[1] = -tc; p[2] = n;   break;
    default: p[0] = -sc; p[1] = -tc; p[2] = -n;  break;
    }
    int major = face >> 1;
    int over = -1;
    for (int k = 0; k < 3; ++k) if (k != major && abs(p[k]) > n) over = k;
    if (over < 0) { *ni = i; *nj = j; return face; }
    p[over] = p[over] > 0 ? n : -n;
    p[major] = p[major] > 0 ? n - 1 : -(n - 1);
    int nf = over * 2 + (p[over] < 0 ? 1 : 0);
    int nsc, ntc;
    switch (nf) {
    case 0: nsc = -p[2]; ntc = -p[1]; break;
    case 1: nsc = p[2];  ntc = -p[1]; break;
    case 2: nsc = p[0];  ntc = p[2];  break;
    case 3: nsc = p[0];  ntc = -p[2]; break;
    case 4: nsc = p[0];  ntc = -p[1]; break;
    default: nsc = -p[0]; ntc = -p[1]; break;
    }
    *ni = (nsc + n - 1) / 2;
    *nj = (ntc + n - 1) / 2;
    return nf;
}

static inline const float* texel_ptr(const float* lvl, int n, int f, int i, int j) {
    return lvl + (((size_t)f * n + j) * n + i) * 4;
}

static void fetch_tap(const float* lvl, int n, int f, int i, int j, float out[4]) {
    int oi = (i < 0 || i >= n), oj = (j < 0 || j >= n);
    if (!oi && !oj) { memcpy(out, texel_ptr(lvl, n, f, i, j), 16); return; }
    if (oi != oj) {
        int ni, nj, nf = orc_cube_neighbor(f, n, i, j, &ni, &nj);
        memcpy(out, texel_ptr(lvl, n, nf, ni, nj), 16);
        return;
    }
    /* corner: the missing texel is replaced by the mean of the three that exist */
    int ci = i < 0 ? 0 : n - 1, cj = j < 0 ? 0 : n - 1;
    float a[4], b[4], c[4];
    memcpy(a, texel_ptr(lvl, n, f, ci, cj), 16);
    fetch_tap(lvl, n, f, i, cj, b);
    fetch_tap(lvl, n, f, ci, j, c);
    for (int k = 0; k < 4; ++k) out[k] = ((a[k] + b[k]) + c[k]) / 3.0f;
}

/* Sampler-coordinate convention (UNPINNED: the reference defers to the driver, gpu_vulkan.c:613-634).  Default: exact fp32 tap
 * weights.  orc_set_cube_sampler_snap(1): texel coordinates and the LOD fraction are first snapped to 1/256 -- the 8 sub-texel /
 * mip-fraction bits real texture units resolve (Vulkan subTexelPrecisionBits / mipmapPrecisionBits), the convention the 2-D / 3-D
 * samplers of the widened passes already use (snap_split).  Exists to MEASURE how far the outputs move between two conventions
 * a conformant implementation may choose (tests/test_gpu_parity.py, bench.py --check). */
static int g_cube_snap = 0;
void orc_set_cube_sampler_snap(int on) { g_cube_snap = on != 0; }
int orc_get_cube_sampler_snap(void) { return g_cube_snap; }
static inline float snap256(float x) { return floorf(x * 256.0f + 0.5f) * (1.0f / 256.0f); }

static void sample_level(const float* pyr, int W, int l, v3 d, float out[4]) {
    int n = level_size(W, l);
    const float* lvl = pyr + orc_level_offset(W, l);
    int f; float sc, tc, ma;
    dir_to_face(d, &f, &sc, &tc, &ma);
    float s = 0.5f * sc / ma + 0.5f;
    float t = 0.5f * tc / ma + 0.5f;
    float u = s * (float)n - 0.5f;
    float v = t * (float)n - 0.5f;
    if (g_cube_snap) { u = snap256(u); v = snap256(v); }
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    float t00[4], t10[4], t01[4], t11[4];
    fetch_tap(lvl, n, f, i0, j0, t00);
    fetch_tap(lvl, n, f, i0 + 1, j0, t10);
    fetch_tap(lvl, n, f, i0, j0 + 1, t01);
    fetch_tap(lvl, n, f, i0 + 1, j0 + 1, t11);
    for (int k = 0; k < 4; ++k) {
        float top = lerpf(t00[k], t10[k], a);
        float bot = lerpf(t01[k], t11[k], a);
        out[k] = lerpf(top, bot, b);
    }
}

void orc_env_analytic(const float dir[3], float out[4]) {
    v3 d = v3_normalize(V3(dir[0], dir[1], dir[2]));
    out[0] = 1.0f + 0.5f * d.x;
    out[1] = 1.0f + 0.5f * d.y * d.y;
    out[2] = 1.0f + 0.5f * d.z * d.x;
    out[3] = 1.0f;
}

static void cube_sample(const float* pyr, int W, int levels, v3 d, float lod, float out[4]) {
    if (!pyr) { float dd[3] = {d.x, d.y, d.z}; orc_env_analytic(dd, out); return; }
    float maxl = (float)(levels - 1);
    if (lod < 0) lod = 0;
    if (lod > maxl) lod = maxl;
    if (g_cube_snap) lod = snap256(lod);
    float fl = floorf(lod);
    int l0 = (int)fl;
    float w = lod - fl;
    sample_level(pyr, W, l0, d, out);
    if (w > 0.0f) {
        int l1 = l0 + 1 < levels ? l0 + 1 : levels - 1;
        float o1[4];
        sample_level(pyr, W, l1, d, o1);
        for (int k = 0; k < 4; ++k) out[k] = lerpf(out[k], o1[k], w);
    }
}
void orc_cube_sample(const float* pyr, int W, int levels, const float dir[3], float lod, float out[4]) {
    cube_sample(pyr, W, levels, V3(dir[0], dir[1], dir[2]), lod, out);
}

/* ------------------------------------------------------------------------------------------ */
/* Extension (no reference counterpart: the reference only loads cube strips):    */
/* equirectangular RGBA32F panorama -> cube level 0.  Z-up; u = atan2(y,x)/2pi + .5 (wraps),     */
/* v = acos(z/|d|)/pi (clamps); bilinear with exact fp32 weights; the two angles in fp64.        */
/* ------------------------------------------------------------------------------------------ */
void orc_equirect_to_cube(const float* eq, int w, int h, int size, float* out) {
    const double inv_2pi = 0.15915494309189535, inv_pi = 0.3183098861837907;
    #pragma omp parallel for collapse(2) num_threads(ORC_NT())
    for (int f = 0; f < 6; ++f)
        for (int y = 0; y < size; ++y)
            for (int x = 0; x < size; ++x) {
                v3 d = face_dir(f, ((float)x + 0.5f) / (float)size, ((float)y + 0.5f) / (float)size);
                double len = sqrt((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z);
                double cz = (double)d.z / len; if (cz > 1.0) cz = 1.0; if (cz < -1.0) cz = -1.0;
                float u = (float)(atan2((double)d.y, (double)d.x) * inv_2pi + 0.5);
                float v = (float)(acos(cz) * inv_pi);
                float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
                float flx = floorf(fx), fly = floorf(fy);
                float a = fx - flx, b = fy - fly;
                int i0 = (int)flx, j0 = (int)fly, i1 = i0 + 1, j1 = j0 + 1;
                i0 = ((i0 % w) + w) % w; i1 = ((i1 % w) + w) % w;
                j0 = clampi(j0, 0, h - 1); j1 = clampi(j1, 0, h - 1);
                float* o = out + (((size_t)f * size + y) * size + x) * 4;
                for (int k = 0; k < 4; ++k) {
                    float t00 = eq[((size_t)j0 * w + i0) * 4 + k], t10 = eq[((size_t)j0 * w + i1) * 4 + k];
                    float t01 = eq[((size_t)j1 * w + i0) * 4 + k], t11 = eq[((size_t)j1 * w + i1) * 4 + k];
                    o[k] = lerpf(lerpf(t00, t10, a), lerpf(t01, t11, a), b);
                }
            }
}

/* ------------------------------------------------------------------------------------------ */
/* A4: Fibonacci-spiral hemisphere sample angles.  gen_prefiltered_env_map.glsl:125-128 (same   */
/* text gen_irradiance_map.glsl:85-88, gen_brdf_integration_map.glsl:171-174):                  */
/*   x = float(i)/float(N); y = float(i)/GOLDEN_RATIO; pitch = PI - acos(x - 1.); yaw = 2.*PI*y  */
/* These depend only on i; hoisting them out of the texel loop changes no bits.                 */
/* ------------------------------------------------------------------------------------------ */
static inline void sample_pitch_yaw(int i, int N, float* pitch, float* yaw) {
    float x = (float)i / (float)N;
    float y = (float)i / ORC_GOLDEN_RATIO;
    *pitch = ORC_PI - acosf(x - 1.0f);
    *yaw = (2.0f * ORC_PI) * y;
}
void orc_sample_angles(int N, float* cs) {
    for (int i = 0; i < N; ++i) {
        float pitch, yaw;
        sample_pitch_yaw(i, N, &pitch, &yaw);
        cs[i * 4 + 0] = cosf(pitch); cs[i * 4 + 1] = sinf(pitch);
        cs[i * 4 + 2] = cosf(yaw);   cs[i * 4 + 3] = sinf(yaw);
    }
}

/* DistributionBeckmann  gen_prefiltered_env_map.glsl:86-91 (= gen_brdf_integration_map.glsl:34-39) */
static inline float beckmann(float NdotH, float m) {
    float m2 = m * m;
    float a = tanf(acosf(NdotH));
    float NdotH2 = NdotH * NdotH;
    return expf(-(a * a) / m2) / (ORC_PI * m2 * NdotH2 * NdotH2);
}
void orc_prefilter_D(int N, float roughness, float* D) {
    for (int i = 0; i < N; ++i) {
        float pitch, yaw;
        sample_pitch_yaw(i, N, &pitch, &yaw);
        D[i] = beckmann(cosf(pitch * 0.5f), roughness);   /* :141 */
    }
}

static const float SOME_VECTOR[3] = {12.123825810901f, 6.11831989512f, -5.12039214121f}; /* :108 */

/* ------------------------------------------------------------------------------------------ */
/* A5: specular prefilter  gen_prefiltered_env_map.glsl:103-152                                 */
/* ------------------------------------------------------------------------------------------ */
void orc_prefilter_mip(const float* pyr, int W, int levels, int out_size, int mip, float roughness,
                       float src_lod, int N, int literal,
                       int face0, int face1, int y0, int y1, float* out) {
    float* cs = NULL; float* D = NULL;
    if (mip != 0 && !literal) {
        cs = (float*)malloc(sizeof(float) * 4 * (size_t)N);
        D = (float*)malloc(sizeof(float) * (size_t)N);
        orc_sample_angles(N, cs);
        orc_prefilter_D(N, roughness, D);
    }
    float dw = (2.0f * ORC_PI) / (float)N;                                  /* :122 */
    int rows = y1 - y0;
    #pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(ORC_NT())
    for (int f = face0; f < face1; ++f)
        for (int yy = 0; yy < rows; ++yy) {
            int y = y0 + yy;
            for (int x = 0; x < out_size; ++x) {
                float u = ((float)x + 0.5f) / (float)out_size;              /* :105 */
                float v = ((float)y + 0.5f) / (float)out_size;
                v3 R = face_dir(f, u, v);                                   /* :107 */
                v3 tangent = v3_normalize(v3_cross(R, V3(SOME_VECTOR[0], SOME_VECTOR[1], SOME_VECTOR[2])));
                float sum[4] = {0, 0, 0, 0};
                if (mip == 0) {
                    cube_sample(pyr, W, levels, R, src_lod, sum);           /* :113 (lod 1.) */
                } else {
                    for (int i = 0; i < N; ++i) {
                        float cp, sp, cy, sy, Di;
                        if (literal) {
                            float pitch, yaw;
                            sample_pitch_yaw(i, N, &pitch, &yaw);
                            cp = cosf(pitch); sp = sinf(pitch); cy = cosf(yaw); sy = sinf(yaw);
                            Di = beckmann(cosf(pitch * 0.5f), roughness);
                        } else {
                            cp = cs[i * 4]; sp = cs[i * 4 + 1]; cy = cs[i * 4 + 2]; sy = cs[i * 4 + 3];
                            Di = D[i];
                        }
                        v3 L = rotate_cs(R, tangent, cp, sp);               /* :132 */
                        L = rotate_cs(L, R, cy, sy);                        /* :133 */
                        float rad[4];
                        cube_sample(pyr, W, levels, L, src_lod, rad);       /* :138 (lod 3+mip) */
                        /* :143  sum += D * vec4(L_radiance, 1.) * cos(pitch) * dw */
                        sum[0] += Di * rad[0] * cp * dw;
                        sum[1] += Di * rad[1] * cp * dw;
                        sum[2] += Di * rad[2] * cp * dw;
                        sum[3] += Di * 1.0f * cp * dw;
                    }
                    for (int k = 0; k < 4; ++k) sum[k] /= ORC_PI;           /* :145 */
                }
                float* o = out + (((size_t)f * out_size + y) * out_size + x) * 4;
                o[0] = sum[0]; o[1] = sum[1]; o[2] = sum[2]; o[3] = sum[3];  /* :151 */
            }
        }
    free(cs); free(D);
}

/* ------------------------------------------------------------------------------------------ */
/* A6: diffuse irradiance  gen_irradiance_map.glsl:73-102                                       */
/* (the literal divisor 32. at :74 is the output size; generalised)                             */
/* ------------------------------------------------------------------------------------------ */
void orc_irradiance(const float* pyr, int W, int levels, int out_size, float src_lod, int N,
                    int literal, int face0, int face1, int y0, int y1, float* out) {
    float* cs = NULL;
    if (!literal) { cs = (float*)malloc(sizeof(float) * 4 * (size_t)N); orc_sample_angles(N, cs); }
    int rows = y1 - y0;
    #pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(ORC_NT())
    for (int f = face0; f < face1; ++f)
        for (int yy = 0; yy < rows; ++yy) {
            int y = y0 + yy;
            for (int x = 0; x < out_size; ++x) {
                float u = ((float)x + 0.5f) / (float)out_size;              /* :74 */
                float v = ((float)y + 0.5f) / (float)out_size;
                v3 Nn = face_dir(f, u, v);                                  /* :78 */
                v3 tangent = v3_normalize(v3_cross(Nn, V3(SOME_VECTOR[0], SOME_VECTOR[1], SOME_VECTOR[2])));
                float sum[3] = {0, 0, 0};
                for (int i = 0; i < N; ++i) {
                    float cp, sp, cy, sy;
                    if (literal) {
                        float pitch, yaw;
                        sample_pitch_yaw(i, N, &pitch, &yaw);
                        cp = cosf(pitch); sp = sinf(pitch); cy = cosf(yaw); sy = sinf(yaw);
                    } else {
                        cp = cs[i * 4]; sp = cs[i * 4 + 1]; cy = cs[i * 4 + 2]; sy = cs[i * 4 + 3];
                    }
                    v3 d = rotate_cs(Nn, tangent, cp, sp);                  /* :91 */
                    d = rotate_cs(d, Nn, cy, sy);                           /* :92 */
                    float val[4];
                    cube_sample(pyr, W, levels, d, src_lod, val);           /* :94 (lod 6.) */
                    sum[0] += cp * val[0]; sum[1] += cp * val[1]; sum[2] += cp * val[2]; /* :95 */
                }
                float* o = out + (((size_t)f * out_size + y) * out_size + x) * 4;
                o[0] = sum[0] / (float)N; o[1] = sum[1] / (float)N; o[2] = sum[2] / (float)N; /* :97 */
                o[3] = 0.0f / (float)N;
            }
        }
    free(cs);
}

/* ------------------------------------------------------------------------------------------ */
/* A7: split-sum BRDF LUT  gen_brdf_integration_map.glsl:142-210                                */
/* GeometryMikkelsen :57-59                                                                     */
/* ------------------------------------------------------------------------------------------ */
static inline float geometry_mikkelsen(float NdotH, float VdotN, float LdotN, float VdotH) {
    float a = 2.0f * NdotH * VdotN / VdotH;
    float b = 2.0f * NdotH * LdotN / VdotH;
    return fminf(1.0f, fminf(a, b));
}

void orc_brdf_lut(int size, int N, int y0, int y1, float* out) {
    float* cs = (float*)malloc(sizeof(float) * 4 * (size_t)N);
    orc_sample_angles(N, cs);
    float dw = 2 * ORC_PI / (float)N;                                        /* :168 */
    #pragma omp parallel for schedule(dynamic, 1) num_threads(ORC_NT())
    for (int y = y0; y < y1; ++y)
        for (int x = 0; x < size; ++x) {
            float NdotV = ((float)x + 0.5f) / (float)size;                   /* :143,:154 */
            float rough = ((float)y + 0.5f) / (float)size;                   /* :155 */
            v3 Nn = V3(0, 0, 1);
            float th = acosf(NdotV);
            v3 V = rotate_cs(Nn, V3(1, 0, 0), cosf(th), sinf(th));           /* :160 */
            float scale = 0.0f, bias = 0.0f;
            for (int i = 0; i < N; ++i) {
                float cp = cs[i * 4], sp = cs[i * 4 + 1], cy = cs[i * 4 + 2], sy = cs[i * 4 + 3];
                v3 L = rotate_cs(Nn, V3(1, 0, 0), cp, sp);                   /* :177 */
                L = rotate_cs(L, Nn, cy, sy);                                /* :178 */
                v3 H = v3_normalize(v3_add(L, V));                           /* :179 */
                float NdotL = v3_dot(Nn, L);
                float NdotH = v3_dot(Nn, H);
                float VdotH = v3_dot(V, H);
                float Dv = beckmann(NdotH, rough);                           /* :192 */
                float G = geometry_mikkelsen(NdotH, NdotV, NdotL, VdotH);    /* :193 */
                float Fc = powf(1.0f - VdotH, 5.0f);                         /* :196 */
                scale += Dv * G * (1 - Fc) * dw / (4.0f * NdotV);            /* :198 */
                bias  += Dv * G * (0 + Fc) * dw / (4.0f * NdotV);            /* :199 */
            }
            out[((size_t)y * size + x) * 2 + 0] = scale;
            out[((size_t)y * size + x) * 2 + 1] = bias;
        }
    free(cs);
}

/* ------------------------------------------------------------------------------------------ */
/* A8: shade pass  lighting_pass.glsl:432-716 (in-scope sub-blocks)           */
/* ------------------------------------------------------------------------------------------ */
static inline void mat4_mul_v4(const float m[16], const float v[4], float o[4]) {
    for (int r = 0; r < 4; ++r)
        o[r] = ((m[0 * 4 + r] * v[0] + m[1 * 4 + r] * v[1]) + m[2 * 4 + r] * v[2]) + m[3 * 4 + r] * v[3];
}
/* InterleavedGradientNoise :119-121 */
static inline float ign(float px, float py) {
    return fractf_(52.9829189f * fractf_(0.06711056f * px + 0.00583715f * py));
}
/* DistributionGGX :21-31 */
static inline float distribution_ggx(float NdotH, float roughness) {
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH2 = NdotH * NdotH;
    float nom = a2;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    denom = ORC_PI * denom * denom;
    return nom / denom;
}
/* FresnelSchlick :76-79 */
static inline v3 fresnel_schlick(float cosTheta, v3 F0) {
    float p = powf(1.0f - cosTheta, 5.0f);
    return V3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}

static void lut_sample(const OrcShadeInputs* in, int flags, float u, float v, float out[2]) {
    if (flags & ORC_SHADE_ANALYTIC) { out[0] = 0.9f - 0.5f * v; out[1] = 0.02f + 0.1f * (1.0f - u); return; }
    int S = in->lut_size;
    float fx = u * (float)S - 0.5f, fy = v * (float)S - 0.5f;
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)flx, j0 = (int)fly, i1 = i0 + 1, j1 = j0 + 1;
    i0 = clampi(i0, 0, S - 1); i1 = clampi(i1, 0, S - 1);
    j0 = clampi(j0, 0, S - 1); j1 = clampi(j1, 0, S - 1);
    for (int k = 0; k < 2; ++k) {
        float t00 = orc_f16_to_f32(in->lut[((size_t)j0 * S + i0) * 2 + k]);
        float t10 = orc_f16_to_f32(in->lut[((size_t)j0 * S + i1) * 2 + k]);
        float t01 = orc_f16_to_f32(in->lut[((size_t)j1 * S + i0) * 2 + k]);
        float t11 = orc_f16_to_f32(in->lut[((size_t)j1 * S + i1) * 2 + k]);
        out[k] = lerpf(lerpf(t00, t10, a), lerpf(t01, t11, a), b);
    }
}

void orc_lut_sample(const uint16_t* lut, int size, float u, float v, float out[2]) {
    OrcShadeInputs in; memset(&in, 0, sizeof in); in.lut = lut; in.lut_size = size;
    lut_sample(&in, 0, u, v, out);
}

static void irradiance_sample(const OrcShadeInputs* in, int flags, v3 d, float out[4]) {
    if (flags & ORC_SHADE_ANALYTIC) {
        float dd[3] = {d.x, d.y, d.z}; orc_env_analytic(dd, out);
        for (int k = 0; k < 4; ++k) out[k] = 0.5f * out[k];
        return;
    }
    cube_sample(in->irradiance, in->irradiance_size, 1, d, 0.0f, out);
}
static void prefiltered_sample(const OrcShadeInputs* in, int flags, v3 d, float lod, float out[4]) {
    if (flags & ORC_SHADE_ANALYTIC) {
        /* SURVEY 8c stand-ins: live-shader KATs use prefiltered(d,lod) = env(d); the
         * "Oracle-A'" (IBL lines un-commented) KATs use env(d)*(1 - 0.1*lod). */
        float dd[3] = {d.x, d.y, d.z}; orc_env_analytic(dd, out);
        if (flags & ORC_SHADE_IBL) {
            float s = 1.0f - 0.1f * lod;
            for (int k = 0; k < 4; ++k) out[k] = out[k] * s;
        }
        return;
    }
    cube_sample(in->prefiltered, in->prefiltered_size, in->prefiltered_levels, d, lod, out);
}

/* ---- N4: deterministic trig, samplers, voxel / screen-space trace ---- */
static void sincos_det(float x, float* sn, float* cs) {
    /* k = nearest multiple of pi/2, r = x - k*pi/2 (two-constant Cody-Waite), then the [-pi/4, pi/4] polynomials */
    float kf = floorf(x * 0.63661977236758134f + 0.5f);
    float r = fmaf(-kf, 1.5707962512969971f, x);
    r = fmaf(-kf, 7.5497894158615964e-08f, r);
    float z = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    int k = (int)kf & 3;
    *sn = (k == 0) ? ps : (k == 1) ? pc : (k == 2) ? -ps : -pc;
    *cs = (k == 0) ? pc : (k == 1) ? -ps : (k == 2) ? -pc : ps;
}
float orc_sinf_det(float x) { float s, c; sincos_det(x, &s, &c); return s; }
float orc_cosf_det(float x) { float s, c; sincos_det(x, &s, &c); return c; }
static float asin_poly_det(float x) {       /* |x| <= 0.5 */
    float z = x * x;
    float p = fmaf(fmaf(fmaf(fmaf(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z, 7.4953002686e-2f), z, 1.6666752422e-1f);
    return fmaf(p * z, x, x);
}
float orc_acosf_det(float x) {              /* x in [0, 1] */
    if (x > 0.5f) return 2.0f * asin_poly_det(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - asin_poly_det(x);
}

static void snap_split(float coord01, int extent, int* i0, int* i1, float* a) {
    float f = coord01 * (float)extent - 0.5f;
    f = floorf(f * 256.0f + 0.5f) * (1.0f / 256.0f);
    float fl = floorf(f);
    *a = f - fl;
    int i = (int)fminf(fmaxf(fl, -1.0f), (float)extent);       /* clamp before converting: rays may have marched to 1e30 */
    *i0 = clampi(i, 0, extent - 1); *i1 = clampi(i + 1, 0, extent - 1);
}
void orc_tex3d_sample(const uint16_t* grid, int n, const float p[3], float out[4]) {
    int i0, i1, j0, j1, k0, k1; float a, b, c;
    snap_split(p[0], n, &i0, &i1, &a); snap_split(p[1], n, &j0, &j1, &b); snap_split(p[2], n, &k0, &k1, &c);
    const int ii[2] = {i0, i1}, jj[2] = {j0, j1}, kk[2] = {k0, k1};
    for (int ch = 0; ch < 4; ++ch) {
        float t[2][2][2];
        for (int z = 0; z < 2; ++z) for (int y = 0; y < 2; ++y) for (int x = 0; x < 2; ++x)
            t[z][y][x] = orc_f16_to_f32(grid[(((size_t)kk[z] * n + jj[y]) * n + ii[x]) * 4 + ch]);
        float z0 = lerpf(lerpf(t[0][0][0], t[0][0][1], a), lerpf(t[0][1][0], t[0][1][1], a), b);
        float z1 = lerpf(lerpf(t[1][0][0], t[1][0][1], a), lerpf(t[1][1][0], t[1][1][1], a), b);
        out[ch] = lerpf(z0, z1, c);
    }
}
float orc_tex2d_nearest_r32f(const float* d, int w, int h, float u, float v) {
    int i = clampi((int)fminf(fmaxf(floorf(u * (float)w), 0.0f), (float)w), 0, w - 1), j = clampi((int)fminf(fmaxf(floorf(v * (float)h), 0.0f), (float)h), 0, h - 1);
    return d[(size_t)j * w + i];
}
void orc_tex2d_sample_lod(const OrcTex2D* levels, int nlevels, float u, float v, float lod, float out[4]) {
    lod = fminf(fmaxf(lod, 0.0f), (float)(nlevels - 1));
    float fl = floorf(lod);
    int l0 = (int)fl;
    float w = lod - fl;
    orc_tex2d_sample(&levels[l0], u, v, out);
    if (w > 0.0f) {
        float o1[4]; orc_tex2d_sample(&levels[l0 + 1 < nlevels ? l0 + 1 : nlevels - 1], u, v, o1);
        for (int k = 0; k < 4; ++k) out[k] = lerpf(out[k], o1[k], w);
    }
}

static void grid_at(const OrcShadeInputs* in, v3 ro, float out[4]) {          /* texture(LIGHTGRID, ro*0.5 + 0.5) */
    float p[3] = {ro.x * 0.5f + 0.5f, ro.y * 0.5f + 0.5f, ro.z * 0.5f + 0.5f};
    orc_tex3d_sample(in->lightgrid, in->lightgrid_size, p, out);
}
static v3 luminance_normalise(const float sum[4]) {                          /* :267-269, :314-316, :420-422 */
    float luminance = 0.299f * sum[0] + 0.587f * sum[1] + 0.114f * sum[2];
    float k = sqrtf(luminance) / fmaxf(luminance, 0.0001f);
    return V3(sum[0] * k, sum[1] * k, sum[2] * k);
}

/* exit statistics of the trace (test infrastructure: lets a test assert that a fixture reaches every exit) */
static uint64_t g_gi_exits[4];      /* 0 off-screen fallback, 1 screen-space hit, 2 no open point, 3 voxel march */
void orc_gi_exit_counts(uint64_t out[4], int reset) {
    for (int k = 0; k < 4; ++k) { out[k] = g_gi_exits[k]; if (reset) g_gi_exits[k] = 0; }
}
#define GI_EXIT(k) do { _Pragma("omp atomic") g_gi_exits[k]++; } while (0)

/* lighting_pass.glsl:273-424 SampleRadianceWithScreenSpaceTrace */
static v3 sample_radiance_ss(const OrcGlobals* g, const OrcShadeInputs* in, v3 V, const float p0_vs[4], v3 ray_origin, v3 ray_direction,
                             int num_steps, float step_scale, float noise_01, float foggyness, float ss_intensity) {
    const float voxel_scale = 2.0f / 128.0f;                                             /* :274 */
    const float ls = g->lightgrid_scale;
    v3 rd = v3_scale(ray_direction, voxel_scale);
    v3 ro = v3_scale(ray_origin, ls);
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0001f};
    for (int i = 0; i < 4; ++i) {                                                        /* :281-288 skip the initial blockage */
        ro = v3_add(ro, rd);
        float rad[4]; grid_at(in, ro, rad);
        if (rad[3] < 0.3f) { sum[0] += rad[0]; sum[1] += rad[1]; sum[2] += rad[2]; sum[3] += 1.0f; break; }
    }
    float opw[4] = {ro.x / ls, ro.y / ls, ro.z / ls, 1.0f}, open_vs[4];
    mat4_mul_v4(g->view_space_from_world, opw, open_vs);                                 /* :290 */
    float d4[4] = {open_vs[0] - p0_vs[0], open_vs[1] - p0_vs[1], open_vs[2] - p0_vs[2], open_vs[3] - p0_vs[3]};   /* :298 */
    float step_length = fmaxf(p0_vs[2], 1.0f) * (1.0f + noise_01) / 100.0f;              /* :300 */
    float lxy = sqrtf(d4[0] * d4[0] + d4[1] * d4[1]);
    v3 ssray_dir = V3(d4[0] / lxy, d4[1] / lxy, d4[2] / lxy);                            /* :301 */
    v3 ssray_step = v3_scale(ssray_dir, step_length);
    v3 pos = V3(p0_vs[0], p0_vs[1], p0_vs[2]);                                           /* :304 */
    float dist_to_travel = sqrtf(d4[0] * d4[0] + d4[1] * d4[1] + d4[2] * d4[2]);         /* :308 */
    float dist_travelled = 0.0f;
    for (int guard = 0; guard < 512; ++guard) {                                          /* :315 for (;;): steps grow >= 1.2x, see header */
        pos = v3_add(pos, ssray_step);
        dist_travelled += step_length;
        float pv[4] = {pos.x, pos.y, pos.z, 1.0f}, ndc[4];
        mat4_mul_v4(g->clip_space_from_view, pv, ndc);                                   /* :319-320 */
        float nw = ndc[3];
        ndc[0] = ndc[0] / nw; ndc[1] = ndc[1] / nw; ndc[2] = ndc[2] / nw; ndc[3] = ndc[3] / nw;
        float cx = fminf(fmaxf(ndc[0], -1.0f), 1.0f), cy = fminf(fmaxf(ndc[1], -1.0f), 1.0f);
        if (cx != ndc[0] || cy != ndc[1]) {                                              /* :322-330 left the screen: fallback */
            v3 fp = V3(ray_origin.x * ls + 2.5f * V.x * voxel_scale, ray_origin.y * ls + 2.5f * V.y * voxel_scale, ray_origin.z * ls + 2.5f * V.z * voxel_scale);
            float s4[4]; grid_at(in, fp, s4);
            GI_EXIT(0);
            return luminance_normalise(s4);
        }
        ssray_step = v3_scale(ssray_step, 1.2f); step_length *= 1.2f;                    /* :332-333 */
        float depth_ndc = orc_tex2d_nearest_r32f(in->depth, in->width, in->height, ndc[0] * 0.5f + 0.5f, ndc[1] * 0.5f + 0.5f);   /* :335 */
        float sn[4] = {ndc[0], ndc[1], depth_ndc, 1.0f}, surf[4];
        mat4_mul_v4(g->view_space_from_clip, sn, surf);                                  /* :338-339 */
        float sw = surf[3];
        surf[0] = surf[0] / sw; surf[1] = surf[1] / sw; surf[2] = surf[2] / sw; surf[3] = surf[3] / sw;
        float ls_surf = sqrtf(surf[0] * surf[0] + surf[1] * surf[1] + surf[2] * surf[2]);
        float ls_pos = sqrtf(pos.x * pos.x + pos.y * pos.y + pos.z * pos.z);
        if (ls_surf < ls_pos) {                                                          /* :343 the ray is behind the visible surface */
            float ts[4], te[4], pe[4] = {pos.x, pos.y, pos.z, 1.0f};
            mat4_mul_v4(g->world_space_from_view, surf, ts);                             /* :348-349 */
            mat4_mul_v4(g->world_space_from_view, pe, te);
            for (int k = 0; k < 4; ++k) { ts[k] = ts[k] * ls * 0.5f + 0.5f; te[k] = te[k] * ls * 0.5f + 0.5f; }
            float noise_offset = noise_01 * 0.2f;                                        /* :351 */
            float alpha = 0.0f;
            const float tt[3] = {noise_offset + 0.2f, noise_offset + 0.4f, noise_offset + 0.6f};
            for (int k = 0; k < 3; ++k) {                                                /* :352-355 */
                float mp[3] = {mixf(ts[0], te[0], tt[k]), mixf(ts[1], te[1], tt[k]), mixf(ts[2], te[2], tt[k])}, r4[4];
                orc_tex3d_sample(in->lightgrid, in->lightgrid_size, mp, r4);
                alpha = (k == 0) ? r4[3] : alpha + r4[3];
            }
            if (alpha < 1.5f) {                                                          /* :357-361 thin: keep marching, faster */
                float f = 2.0f + noise_01;
                ssray_step = v3_scale(ssray_step, f); step_length *= f;
                continue;
            }
            float rad[4];                                                                /* :372-382 solid: last frame's radiance on screen */
            orc_tex2d_sample_lod(in->prev_frame, in->prev_frame_levels, ndc[0] * 0.5f + 0.5f, ndc[1] * 0.5f + 0.5f, fminf(step_length * 5.0f, 5.0f), rad);
            GI_EXIT(1);
            return V3(rad[0] * ss_intensity, rad[1] * ss_intensity, rad[2] * ss_intensity);
        }
        if (dist_travelled > dist_to_travel) break;                                      /* :393 */
    }
    if (sum[3] < 0.5f) { GI_EXIT(2); return V3(0, 0, 0); }                               /* :400-403 */
    rd = v3_scale(rd, step_scale);                                                       /* :407-408 */
    ro = V3(ro.x + rd.x * noise_01, ro.y + rd.y * noise_01, ro.z + rd.z * noise_01);
    for (int i = 0; i < num_steps; ++i) {                                                /* :411-419 continue until hitting a voxel */
        ro = V3(ro.x + 0.5f * rd.x, ro.y + 0.5f * rd.y, ro.z + 0.5f * rd.z);
        float rad[4]; grid_at(in, ro, rad);
        if (rad[3] > 0.3f) break;
        sum[0] = sum[0] * foggyness + rad[0]; sum[1] = sum[1] * foggyness + rad[1]; sum[2] = sum[2] * foggyness + rad[2];
        sum[3] = sum[3] * foggyness + 1.0f;
    }
    float sw = sum[3];
    sum[0] = sum[0] / sw; sum[1] = sum[1] / sw; sum[2] = sum[2] / sw; sum[3] = sum[3] / sw;   /* :421 */
    GI_EXIT(3);
    return luminance_normalise(sum);
}

void orc_shade(const OrcGlobals* g, const OrcShadeInputs* in, int flags,
               int x0, int x1, int y0, int y1, float* out_rgba) {
    const int W = in->width, H = in->height;
    #pragma omp parallel for schedule(static) num_threads(ORC_NT())
    for (int py = y0; py < y1; ++py)
        for (int px = x0; px < x1; ++px) {
            size_t pi = (size_t)py * W + px;
            const float inv255 = 1.0f / 255.0f; (void)inv255;
            /* :433-442  G-buffer decode (unorm8 -> float = b/255; point fetch at pixel centre) */
            v3 base = V3(in->base_color[pi * 4] / 255.0f, in->base_color[pi * 4 + 1] / 255.0f, in->base_color[pi * 4 + 2] / 255.0f);
            v3 Nn = V3(in->normal[pi * 4] / 255.0f, in->normal[pi * 4 + 1] / 255.0f, in->normal[pi * 4 + 2] / 255.0f);
            Nn = V3(Nn.x * 2.0f - 1.0f, Nn.y * 2.0f - 1.0f, Nn.z * 2.0f - 1.0f);           /* :435 */
            v3 orm = V3(in->orm[pi * 4] / 255.0f, in->orm[pi * 4 + 1] / 255.0f, in->orm[pi * 4 + 2] / 255.0f);
            v3 emissive = V3(in->emissive[pi * 4] / 255.0f * 10.0f, in->emissive[pi * 4 + 1] / 255.0f * 10.0f,
                             in->emissive[pi * 4 + 2] / 255.0f * 10.0f);                  /* :440 */
            float roughness = orm.y, metallic = orm.z;                                    /* :441-442 */

            /* :444-451 position reconstruction */
            float fs_u = ((float)px + 0.5f) / (float)W, fs_v = ((float)py + 0.5f) / (float)H;
            float p0_ndc[4] = {fs_u * 2.0f - 1.0f, fs_v * 2.0f - 1.0f, in->depth[pi], 1.0f};
            float pw[4];
            mat4_mul_v4(g->world_space_from_clip, p0_ndc, pw);
            v3 p0_world = V3(pw[0] / pw[3], pw[1] / pw[3], pw[2] / pw[3]);

            /* :456-459 noise */
            float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
            float noise_offset = (1000 * 1.61803398875f) * g->frame_idx_mod_59;
            float noise_1 = fractf_(ign(fcx, fcy) + noise_offset);
            float noise_2 = fractf_(ign(fcx + 90.0f, fcy + 20.0f) + noise_offset);
            float noise_3 = fractf_(ign(fcx + 522.0f, fcy + 55.0f) + noise_offset);

            /* :594-608 sun shadow: 4 PCF taps around a noise-jittered position (shadow = 1 without ORC_SHADE_SHADOWS) */
            float shadow = 1.0f;
            float p0s[4] = {0, 0, 0, 0};                                                  /* p0_sun_space */
            if (flags & (ORC_SHADE_SHADOWS | ORC_SHADE_SHAFTS)) {
                float sp4[4] = {p0_world.x + Nn.x * 0.1f, p0_world.y + Nn.y * 0.1f, p0_world.z + Nn.z * 0.1f, 1.0f};   /* :596 */
                mat4_mul_v4(g->sun_space_from_world, sp4, p0s);                          /* :597 */
            }
            if (flags & ORC_SHADE_SHADOWS) {
                const float px_size = 1.0f / 2048.0f;                                    /* :594 */
                float sx = p0s[0] * 0.5f + 0.5f, sy = p0s[1] * 0.5f + 0.5f, sz = p0s[2]; /* :598 */
                sx = sx + (2.0f * (noise_2 - 0.5f)) * px_size;                           /* :600 */
                sy = sy + (2.0f * (noise_1 - 0.5f)) * px_size;
                static const float ox[4] = {0.75f, -0.25f, 0.25f, -0.75f}, oy[4] = {0.25f, 0.75f, -0.75f, -0.25f};
                float acc = 0.0f;
                for (int k = 0; k < 4; ++k)                                              /* :604-608 */
                    acc = acc + orc_shadow_sample(&in->sun_depth_map, sx + ox[k] * px_size, sy + oy[k] * px_size, sz + 0.0f * px_size);
                shadow = acc * 0.25f;
            }

            /* :612-613 */
            v3 cam = V3(g->camera_pos[0], g->camera_pos[1], g->camera_pos[2]);
            v3 V = v3_normalize(v3_sub(cam, p0_world));
            float VdotN = fmaxf(v3_dot(V, Nn), 0.0f);

            v3 sun_emission = V3(25.0f * 1.0f, 25.0f * 0.9f, 25.0f * 0.7f);               /* :616 */
            v3 outgoing = V3(0, 0, 0);

            /* :622-651 light shafts; visibility from the sun depth map with ORC_SHADE_SHADOWS, else 1 */
            if (flags & ORC_SHADE_SHAFTS) {
                const float intensity = 0.001f;
                float c4[4] = {cam.x, cam.y, cam.z, 1.0f};
                float rp[4]; mat4_mul_v4(g->sun_space_from_world, c4, rp);               /* :627 */
                v3 pos = V3(rp[0], rp[1], rp[2]);
                v3 delta = V3(p0s[0] - rp[0], p0s[1] - rp[1], p0s[2] - rp[2]);            /* :630 */
                float dist = sqrtf(v3_dot(delta, delta));
                float travelled = 0.0f;
                const float step = 1.0f / 16.0f;
                v3 stepv = V3(step * (delta.x / dist), step * (delta.y / dist), step * (delta.z / dist));   /* :635 */
                pos = V3(pos.x + stepv.x * noise_1, pos.y + stepv.y * noise_1, pos.z + stepv.z * noise_1);   /* :637 */
                travelled += step * noise_1;                                             /* :638 */
                for (int guard = 0; guard < 65536; ++guard) {                            /* :640-650 */
                    pos = v3_add(pos, stepv);
                    travelled += step;
                    if (travelled > dist) break;
                    float visibility = 1.0f;
                    if (flags & ORC_SHADE_SHADOWS)
                        visibility = orc_shadow_sample(&in->sun_depth_map, pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z);   /* :644-646 */
                    outgoing.x += intensity * visibility * sun_emission.x;
                    outgoing.y += intensity * visibility * sun_emission.y;
                    outgoing.z += intensity * visibility * sun_emission.z;
                }
            }

            /* :657-661 */
            v3 F0 = V3(mixf(0.04f, base.x, metallic), mixf(0.04f, base.y, metallic), mixf(0.04f, base.z, metallic));
            v3 kS = fresnel_schlick(fmaxf(v3_dot(Nn, V), 0.0f), F0);
            v3 kD = V3((1.0f - kS.x) * (1.0f - metallic), (1.0f - kS.y) * (1.0f - metallic), (1.0f - kS.z) * (1.0f - metallic));

            /* :664-679 sun */
            {
                v3 L = V3(-g->sun_direction[0], -g->sun_direction[1], -g->sun_direction[2]);
                v3 Hh = v3_normalize(v3_add(L, V));
                float NdotL = fmaxf(v3_dot(Nn, L), 0.0f);
                if (NdotL > 0.0f) {
                    float VdotH = fmaxf(v3_dot(V, Hh), 0.0f);
                    float NdotH = fmaxf(v3_dot(Nn, Hh), 0.0f);
                    float Dv = distribution_ggx(NdotH, roughness);
                    float G = geometry_mikkelsen(NdotH, VdotN, NdotL, VdotH);
                    v3 F = fresnel_schlick(VdotH, F0);
                    float den = fmaxf(4.0f * NdotL * VdotN, 0.0001f);
                    v3 brdf = V3(F.x * G * Dv / den, F.y * G * Dv / den, F.z * G * Dv / den);
                    outgoing.x += shadow * (kD.x * base.x / ORC_PI + brdf.x) * sun_emission.x * NdotL;
                    outgoing.y += shadow * (kD.y * base.y / ORC_PI + brdf.y) * sun_emission.y * NdotL;
                    outgoing.z += shadow * (kD.z * base.z / ORC_PI + brdf.z) * sun_emission.z * NdotL;
                }
            }

            /* :681 LUT fetch */
            float sb[2];
            lut_sample(in, flags, VdotN, fmaxf(roughness, 0.05f), sb);

            /* :683-687 ambient: voxel GI is out of scope (== 0 with an empty light grid); IBL mode
             * uses the commented line :690  irradiance = textureLod(TEX_IRRADIANCE_MAP, N, 0.) */
            v3 ambient = V3(0, 0, 0);
            if (flags & ORC_SHADE_IBL) {
                float ir[4]; irradiance_sample(in, flags, Nn, ir);
                ambient = V3(ir[0], ir[1], ir[2]);
            }
            /* the sky test of :708 decides early here: sky pixels overwrite everything, so their traces are not evaluated */
            const int sky_px = (fminf(fmaxf(p0_world.x, -99.0f), 99.0f) != p0_world.x) ||
                               (fminf(fmaxf(p0_world.y, -99.0f), 99.0f) != p0_world.y) ||
                               (fminf(fmaxf(p0_world.z, -99.0f), 99.0f) != p0_world.z);
            float p0_view[4] = {0, 0, 0, 1};
            if ((flags & ORC_SHADE_GI) && !sky_px) {
                float pv[4]; mat4_mul_v4(g->view_space_from_clip, p0_ndc, pv);              /* :446-447 */
                p0_view[0] = pv[0] / pv[3]; p0_view[1] = pv[1] / pv[3]; p0_view[2] = pv[2] / pv[3]; p0_view[3] = pv[3] / pv[3];
                /* :546-577 random direction in the hemisphere around N (cosine-distributed pitch) */
                v3 some_vector = v3_normalize(V3(0.7128864983f, 0.8217892113f, 0.948912748f));
                v3 tangent = v3_normalize(v3_cross(some_vector, Nn));
                v3 bitangent = v3_cross(Nn, tangent);
                float pitch = orc_acosf_det(sqrtf(1.0f - noise_1));
                float yaw = (2.0f * ORC_PI) * noise_3;
                float sp = orc_sinf_det(pitch), cp = orc_cosf_det(pitch), cyw = orc_cosf_det(yaw), syw = orc_sinf_det(yaw);
                float lx = sp * cyw, ly = sp * syw, lz = cp;
                v3 bent = V3((tangent.x * lx + bitangent.x * ly) + Nn.x * lz, (tangent.y * lx + bitangent.y * ly) + Nn.y * lz,
                             (tangent.z * lx + bitangent.z * ly) + Nn.z * lz);
                ambient = sample_radiance_ss(g, in, V, p0_view, p0_world, bent, 12, 1.0f, noise_3, 0.5f, 0.75f);   /* :685 */
            }
            outgoing.x += kD.x * ambient.x * base.x;
            outgoing.y += kD.y * ambient.y * base.y;
            outgoing.z += kD.z * ambient.z * base.z;

            /* :693-697 reflection vector */
            v3 I = V3(-V.x, -V.y, -V.z);
            float dNI = v3_dot(Nn, I);
            v3 R = V3(I.x - 2.0f * dNI * Nn.x, I.y - 2.0f * dNI * Nn.y, I.z - 2.0f * dNI * Nn.z);
            float jr = 0.6f * roughness;
            R = v3_normalize(V3(R.x + jr * (noise_1 - 0.5f), R.y + jr * (noise_2 - 0.5f), R.z + jr * (noise_3 - 0.5f)));
            float r2 = roughness * roughness;
            float r4 = r2 * r2;
            R = V3(mixf(R.x, Nn.x, r4), mixf(R.y, Nn.y, r4), mixf(R.z, Nn.z, r4));

            /* :699-702 specular: IBL mode = commented line :699 */
            v3 spec = V3(0, 0, 0);
            if (flags & ORC_SHADE_IBL) {
                float pc[4]; prefiltered_sample(in, flags, R, roughness * 4.0f, pc);
                spec = V3(pc[0], pc[1], pc[2]);
            }
            if ((flags & ORC_SHADE_GI) && !sky_px)
                spec = sample_radiance_ss(g, in, V, p0_view, p0_world, R, 16, 2.0f, noise_3, roughness, 0.9f);   /* :701 */
            outgoing.x += spec.x * (F0.x * sb[0] + sb[1]);
            outgoing.y += spec.y * (F0.y * sb[0] + sb[1]);
            outgoing.z += spec.z * (F0.z * sb[0] + sb[1]);

            outgoing = v3_add(outgoing, emissive);                                        /* :706 */

            /* :708-710 sky */
            int sky = (fminf(fmaxf(p0_world.x, -99.0f), 99.0f) != p0_world.x) ||
                      (fminf(fmaxf(p0_world.y, -99.0f), 99.0f) != p0_world.y) ||
                      (fminf(fmaxf(p0_world.z, -99.0f), 99.0f) != p0_world.z);
            if (sky) {
                float pc[4]; prefiltered_sample(in, flags, V3(-V.x, -V.y, -V.z), 1.0f, pc);
                outgoing = V3(pc[0], pc[1], pc[2]);
            }

            float* o = out_rgba + pi * 4;                                                 /* :712-713 */
            o[0] = fmaxf(outgoing.x, 0.0f); o[1] = fmaxf(outgoing.y, 0.0f); o[2] = fmaxf(outgoing.z, 0.0f);
            o[3] = 1.0f;
        }
}

/* ------------------------------------------------------------------------------------------
 * N2: light-grid sweep (lightgrid_sweep.glsl:9-75; dispatched by render.cpp:1064-1072 as (1,16,16)
 * groups of 1x8x8 over a 128^3 RGBA16F image, X_direction cycling 0,1,2 per frame)
 * ------------------------------------------------------------------------------------------ */
void orc_lightgrid_sweep(uint16_t* img, int w, int h, int d, int direction, int ny, int nz) {
    (void)d;
#pragma omp parallel for collapse(2) schedule(static)
    for (int iz = 0; iz < nz; ++iz)
        for (int iy = 0; iy < ny; ++iy) {
            /* :10-22 base coordinate and step */
            int bx = 0, by = iy, bz = iz, sx = 1, sy = 0, sz = 0;
            if (direction == 0) {}
            else if (direction == 1) { int tx = bz, ty = bx, tz = by; bx = tx; by = ty; bz = tz; sx = 0; sy = 1; }
            else { int tx = by, ty = bz, tz = bx; bx = tx; by = ty; bz = tz; sx = 0; sz = 1; }
            const float SKY[3] = {1.0f, 1.2f, 2.0f};                                     /* :24 */
            float old_v[ORC_SWEEP_LEN][4], val[ORC_SWEEP_LEN][4];
            uint16_t* px[ORC_SWEEP_LEN];
            for (int x = 0; x < ORC_SWEEP_LEN; ++x) {                                    /* :28-31 */
                px[x] = img + (((size_t)(bz + x * sz) * h + (by + x * sy)) * w + (bx + x * sx)) * 4;
                for (int c = 0; c < 4; ++c) old_v[x][c] = val[x][c] = orc_f16_to_f32(px[x][c]);
            }
            const float move_ratio = 0.5f;                                               /* :33 */
            float m[3] = {SKY[0], SKY[1], SKY[2]};                                       /* :36 */
            for (int x = 0; x < ORC_SWEEP_LEN; ++x) {                                    /* :37-48 */
                if (old_v[x][3] > 0.5f) { m[0] = old_v[x][0]; m[1] = old_v[x][1]; m[2] = old_v[x][2]; }
                else for (int c = 0; c < 3; ++c) {
                    val[x][c] = val[x][c] + m[c];
                    m[c] = move_ratio * val[x][c];
                    val[x][c] = val[x][c] - m[c];
                }
            }
            for (int c = 0; c < 3; ++c) val[ORC_SWEEP_LEN - 1][c] = val[ORC_SWEEP_LEN - 1][c] + m[c];   /* :49 */
            m[0] = SKY[0]; m[1] = SKY[1]; m[2] = SKY[2];                                 /* :52 */
            for (int x = ORC_SWEEP_LEN - 1; x >= 0; --x) {                               /* :53-66 */
                if (old_v[x][3] > 0.5f) { m[0] = old_v[x][0]; m[1] = old_v[x][1]; m[2] = old_v[x][2]; }
                else for (int c = 0; c < 3; ++c) {
                    val[x][c] = val[x][c] + m[c];
                    m[c] = move_ratio * val[x][c];
                    val[x][c] = val[x][c] - m[c];
                }
            }
            for (int c = 0; c < 3; ++c) val[0][c] = val[0][c] + m[c];                    /* :67 */
            for (int x = 0; x < ORC_SWEEP_LEN; ++x) {                                    /* :70-75 */
                if (old_v[x][3] < 0.5f)
                    for (int c = 0; c < 4; ++c)
                        px[x][c] = orc_f32_to_f16(old_v[x][c] * (1.0f - 0.35f) + val[x][c] * 0.35f);   /* mix(): x*(1-a)+y*a */
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * N3: post-process tail
 * ------------------------------------------------------------------------------------------ */
static void tex2d_texel(const OrcTex2D* t, int i, int j, float o[4]) {
    size_t idx = (size_t)j * t->width + i;
    o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f;
    switch (t->format) {
    case ORC_TEX_RGBA16F: { const uint16_t* p = (const uint16_t*)t->data + idx * 4; for (int k = 0; k < 4; ++k) o[k] = orc_f16_to_f32(p[k]); break; }
    case ORC_TEX_RG16F:   { const uint16_t* p = (const uint16_t*)t->data + idx * 2; o[0] = orc_f16_to_f32(p[0]); o[1] = orc_f16_to_f32(p[1]); break; }
    case ORC_TEX_R32F:    { o[0] = ((const float*)t->data)[idx]; break; }
    default:              { const float* p = (const float*)t->data + idx * 4; for (int k = 0; k < 4; ++k) o[k] = p[k]; break; }
    }
}

void orc_tex2d_sample(const OrcTex2D* t, float u, float v, float out[4]) {
    float fx = u * (float)t->width - 0.5f, fy = v * (float)t->height - 0.5f;
    fx = floorf(fx * 256.0f + 0.5f) * (1.0f / 256.0f);           /* 8 sub-texel bits */
    fy = floorf(fy * 256.0f + 0.5f) * (1.0f / 256.0f);
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)fminf(fmaxf(flx, -1.0f), (float)t->width), j0 = (int)fminf(fmaxf(fly, -1.0f), (float)t->height), i1 = i0 + 1, j1 = j0 + 1;
    i0 = clampi(i0, 0, t->width - 1); i1 = clampi(i1, 0, t->width - 1);
    j0 = clampi(j0, 0, t->height - 1); j1 = clampi(j1, 0, t->height - 1);
    float t00[4], t10[4], t01[4], t11[4];
    tex2d_texel(t, i0, j0, t00); tex2d_texel(t, i1, j0, t10); tex2d_texel(t, i0, j1, t01); tex2d_texel(t, i1, j1, t11);
    for (int k = 0; k < 4; ++k) out[k] = lerpf(lerpf(t00[k], t10[k], a), lerpf(t01[k], t11[k], a), b);
}

float orc_shadow_sample(const OrcTex2D* t, float u, float v, float ref) {
    float fx = u * (float)t->width - 0.5f, fy = v * (float)t->height - 0.5f;
    fx = floorf(fx * 256.0f + 0.5f) * (1.0f / 256.0f);
    fy = floorf(fy * 256.0f + 0.5f) * (1.0f / 256.0f);
    float flx = floorf(fx), fly = floorf(fy);
    float a = fx - flx, b = fy - fly;
    int i0 = (int)fminf(fmaxf(flx, -1.0f), (float)t->width), j0 = (int)fminf(fmaxf(fly, -1.0f), (float)t->height), i1 = i0 + 1, j1 = j0 + 1;
    i0 = clampi(i0, 0, t->width - 1); i1 = clampi(i1, 0, t->width - 1);
    j0 = clampi(j0, 0, t->height - 1); j1 = clampi(j1, 0, t->height - 1);
    const float* d = (const float*)t->data;
    float c00 = ref < d[(size_t)j0 * t->width + i0] ? 1.0f : 0.0f, c10 = ref < d[(size_t)j0 * t->width + i1] ? 1.0f : 0.0f;
    float c01 = ref < d[(size_t)j1 * t->width + i0] ? 1.0f : 0.0f, c11 = ref < d[(size_t)j1 * t->width + i1] ? 1.0f : 0.0f;
    return lerpf(lerpf(c00, c10, a), lerpf(c01, c11, a), b);
}

static float mitchell_netravali(float x) {                        /* taa_resolve.glsl:13-26 */
    float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
    float ax = fabsf(x);
    if (ax < 1.0f)
        return ((12.0f - 9.0f * B - 6.0f * C) * ax * ax * ax + (-18.0f + 12.0f * B + 6.0f * C) * ax * ax + (6.0f - 2.0f * B)) / 6.0f;
    else if (ax >= 1.0f && ax < 2.0f)
        return ((-B - 6.0f * C) * ax * ax * ax + (6.0f * B + 30.0f * C) * ax * ax + (-12.0f * B - 48.0f * C) * ax + (8.0f * B + 24.0f * C)) / 6.0f;
    return 0.0f;
}

/* taa_resolve.glsl:133-178: Catmull-Rom history filter with 9 bilinear taps */
static void history_catmull_rom(const OrcTex2D* hist, float uvx, float uvy, float tsx, float tsy, float out[4]) {
    float sp[2] = {uvx * tsx, uvy * tsy}, ts[2] = {tsx, tsy};
    float tp0[2], tp3[2], tp12[2], w0[2], w3[2], w12[2];
    for (int k = 0; k < 2; ++k) {
        float tp1 = floorf(sp[k] - 0.5f) + 0.5f;                                        /* :141 */
        float f = sp[k] - tp1;                                                          /* :145 */
        w0[k] = f * (-0.5f + f * (1.0f - 0.5f * f));                                    /* :150-153 */
        float w1 = 1.0f + f * f * (-2.5f + 1.5f * f);
        float w2 = f * (0.5f + f * (2.0f - 1.5f * f));
        w3[k] = f * f * (-0.5f + 0.5f * f);
        w12[k] = w1 + w2;                                                               /* :157-158 */
        float offset12 = w2 / (w1 + w2);
        tp0[k] = (tp1 - 1.0f) / ts[k];                                                  /* :161-167 */
        tp3[k] = (tp1 + 2.0f) / ts[k];
        tp12[k] = (tp1 + offset12) / ts[k];
    }
    const float* px[3] = {tp0, tp12, tp3}; const float* wx[3] = {w0, w12, w3};
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    for (int row = 0; row < 3; ++row)                                                   /* :170-180, row-major order */
        for (int col = 0; col < 3; ++col) {
            float s[4]; orc_tex2d_sample(hist, px[col][0], px[row][1], s);
            for (int k = 0; k < 4; ++k) out[k] = out[k] + (s[k] * wx[col][0]) * wx[row][1];
        }
}

void orc_taa_resolve(const OrcTaaInputs* in, int width, int height, int y0, int y1, float* out_rgba) {
    (void)height;
    #pragma omp parallel for schedule(static) num_threads(ORC_NT())
    for (int py = y0; py < y1; ++py)
        for (int px = 0; px < width; ++px) {
            float tsx = (float)in->lighting_result.width, tsy = (float)in->lighting_result.height;   /* :189 */
            float psx = 1.0f / tsx, psy = 1.0f / tsy;                                   /* :190 */
            float uvx = ((float)px + 0.5f) * psx, uvy = ((float)py + 0.5f) * psy;       /* :192 */
            float total[3] = {0, 0, 0}, wsum = 0.0f;
            float nmin[3] = {10000, 10000, 10000}, nmax[3] = {-10000, -10000, -10000};
            float m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
            float closest_depth = 10000.0f, cdu = 0.0f, cdv = 0.0f;
            for (int x = -1; x <= 1; ++x)                                               /* :205-227 */
                for (int y = -1; y <= 1; ++y) {
                    float su = uvx + (float)x * psx, sv = uvy + (float)y * psy;
                    float nb[4]; orc_tex2d_sample(&in->lighting_result, su, sv, nb);
                    float w = mitchell_netravali(sqrtf((float)x * (float)x + (float)y * (float)y));
                    for (int k = 0; k < 3; ++k) {
                        total[k] = total[k] + nb[k] * w;
                        nmin[k] = fminf(nmin[k], nb[k]); nmax[k] = fmaxf(nmax[k], nb[k]);
                        m1[k] = m1[k] + nb[k];
                        m2[k] = m2[k] + nb[k] * nb[k];
                    }
                    wsum = wsum + w;
                    float d[4]; orc_tex2d_sample(&in->gbuffer_depth, uvx, uvy, d);      /* :221 samples uv, not sample_uv */
                    if (d[0] < closest_depth) { closest_depth = d[0]; cdu = su; cdv = sv; }
                }
            float src[3] = {total[0] / wsum, total[1] / wsum, total[2] / wsum};         /* :228 */
            float vel[4]; orc_tex2d_sample(&in->gbuffer_velocity, cdu, cdv, vel);       /* :230 */
            float ru = uvx - vel[0] * 0.5f, rv = uvy - vel[1] * 0.5f;                   /* :231 */
            float pvel[4]; orc_tex2d_sample(&in->gbuffer_velocity_prev, ru, rv, pvel);  /* :232 */
            float prev[4]; history_catmull_rom(&in->prev_frame_result, ru, rv, tsx, tsy, prev);   /* :234 */
            const float inv9 = 1.0f / 9.0f, gamma = 1.0f;                               /* :237-238 */
            for (int k = 0; k < 3; ++k) {                                               /* :239-244 */
                float avg = m1[k] * inv9;
                float sigma = sqrtf(fabsf(m2[k] * inv9 - avg * avg));
                float minc = avg - gamma * sigma, maxc = avg + gamma * sigma;
                prev[k] = fminf(fmaxf(prev[k], minc), maxc);
            }
            float wB = 0.05f, wA = 1.0f - wB;                                           /* :252-253 */
            float dvx = pvel[0] - vel[0], dvy = pvel[1] - vel[1];
            float velocity_diff = 1000.0f * sqrtf(dvx * dvx + dvy * dvy);               /* :269 */
            wB = wB + velocity_diff;                                                    /* :270 */
            float cu = fminf(fmaxf(ru, 0.0f), 1.0f), cv = fminf(fmaxf(rv, 0.0f), 1.0f);
            if (ru != cu || rv != cv) { wA = 0.0f; wB = 1.0f; }                         /* :272-275 */
            float* o = out_rgba + ((size_t)py * width + px) * 4;
            float den = fmaxf(wB + wA, 0.00001f);
            for (int k = 0; k < 3; ++k) o[k] = (src[k] * wB + prev[k] * wA) / den;      /* :277 */
            o[3] = 1.0f;                                                                /* :290 */
            (void)nmin; (void)nmax;
        }
}

static float aces_approx(float v) {                                /* final_post_process.glsl:2-10 */
    v = v * 0.6f;
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(fmaxf((v * (a * v + b)) / (v * (c * v + d) + e), 0.0f), 1.0f);
}

void orc_final_post_process(const OrcTex2D* bloom_result, int width, int height, int y0, int y1, float* out_rgba) {
    #pragma omp parallel for schedule(static) num_threads(ORC_NT())
    for (int py = y0; py < y1; ++py)
        for (int px = 0; px < width; ++px) {
            float s[4]; orc_tex2d_sample(bloom_result, ((float)px + 0.5f) / (float)width, ((float)py + 0.5f) / (float)height, s);
            float* o = out_rgba + ((size_t)py * width + px) * 4;
            for (int k = 0; k < 3; ++k) o[k] = powf(aces_approx(2.0f * s[k]), 1.0f / 2.2f);   /* :32-33 */
            o[3] = 1.0f;
        }
}

uint8_t orc_unorm8(float v) {
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return (uint8_t)lrintf(v * 255.0f);                            /* round to nearest even (default rounding mode) */
}

/* ---- bloom chain (post-process, between TAA and the final pass) ---- */
void orc_bloom_downsample(const OrcTex2D* src, int dw, int dh, int dst_mip_level, float* out_rgba) {
    #pragma omp parallel for schedule(static) num_threads(ORC_NT())
    for (int py = 0; py < dh; ++py)
        for (int px = 0; px < dw; ++px) {
            float u = ((float)px + 0.5f) / (float)dw, v = ((float)py + 0.5f) / (float)dh;
            float x = 1.0f / (float)src->width, y = 1.0f / (float)src->height;          /* :40-42 */
            static const float ox[13] = {-2, 0, 2, -2, 0, 2, -2, 0, 2, -1, 1, -1, 1};     /* a b c d e f g h i j k l m (:50-64) */
            static const float oy[13] = {-2, -2, -2, 0, 0, 0, 2, 2, 2, -1, -1, 1, 1};
            float t[13][4];
            for (int k = 0; k < 13; ++k) orc_tex2d_sample(src, u + ox[k] * x, v + oy[k] * y, t[k]);
            float* o = out_rgba + ((size_t)py * dw + px) * 4;
            for (int c = 0; c < 3; ++c) {
                float sum = t[4][c] * 0.125f;                                                   /* :90 */
                sum = sum + (((t[0][c] + t[2][c]) + t[6][c]) + t[8][c]) * 0.03125f;             /* :91 (a+c+g+i) */
                sum = sum + (((t[1][c] + t[3][c]) + t[5][c]) + t[7][c]) * 0.0625f;              /* :92 (b+d+f+h) */
                sum = sum + (((t[9][c] + t[10][c]) + t[11][c]) + t[12][c]) * 0.125f;            /* :93 (j+k+l+m) */
                if (dst_mip_level == 1) sum = fminf(sum, 1.0f);                                 /* :94-97 */
                o[c] = sum;
            }
            o[3] = 1.0f;
        }
}

void orc_bloom_upsample(const OrcTex2D* src, int dw, int dh, int dst_mip_level, float* out_rgba) {
    #pragma omp parallel for schedule(static) num_threads(ORC_NT())
    for (int py = 0; py < dh; ++py)
        for (int px = 0; px < dw; ++px) {
            float u = ((float)px + 0.5f) / (float)dw, v = ((float)py + 0.5f) / (float)dh;
            const float radius = 1.5f;                                                          /* :26-28 */
            float x = radius / (float)src->width, y = radius / (float)src->height;
            float factor = 1.0f;
            if (dst_mip_level == 0) factor = 0.06f;                                             /* :37 */
            static const float ox[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}, oy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1};   /* a..i (:43-51) */
            float t[9][4];
            for (int k = 0; k < 9; ++k) orc_tex2d_sample(src, u + ox[k] * x, v + oy[k] * y, t[k]);
            float* o = out_rgba + ((size_t)py * dw + px) * 4;
            for (int c = 0; c < 3; ++c) {
                float sum = t[4][c] * 4.0f;                                                     /* :55 */
                sum = sum + (((t[1][c] + t[3][c]) + t[5][c]) + t[7][c]) * 2.0f;                 /* :56 */
                sum = sum + (((t[0][c] + t[2][c]) + t[6][c]) + t[8][c]);                        /* :57 */
                o[c] = sum * factor / 16.0f;                                                    /* :58 */
            }
            o[3] = 1.0f;
        }
}
