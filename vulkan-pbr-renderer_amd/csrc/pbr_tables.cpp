// pbr_tables.cpp -- host-side sample/weight tables and pyramid layout arithmetic.
//
// The per-sample quantities of the reference's Monte-Carlo loops depend only on the sample index
// (SURVEY.md Appendix A): they are evaluated once here with the host libm, in the shader's own
// operation order, and uploaded.  The kernels' inner loops then contain no transcendentals and the
// GPU sees exactly the sample directions a CPU evaluation of the shader would use.
// Build: -ffp-contract=off (separately rounded fp32 ops, as GLSL source order implies).
#include "pbr_kernels.h"

#include <math.h>

#define T_PI 3.14159265358979323846f     // shaders: #define PI
#define T_GOLDEN 1.61803398875f          // shaders: #define GOLDEN_RATIO

extern "C" {

int pbrk_mip_count(int w, int h) {
    int s = w < h ? w : h, c = 1;
    while (s > 1) { s /= 2; c++; }
    return c;
}
static inline size_t lvl_n(int W, int l) { int n = W >> l; return (size_t)(n < 1 ? 1 : n); }
size_t pbrk_level_offset(int W, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += 6 * lvl_n(W, l) * lvl_n(W, l);
    return off;
}
size_t pbrk_pyramid_texels(int W, int levels) { return pbrk_level_offset(W, levels); }
size_t pbrk_bordered_level_offset(int W, int level) {
    size_t off = 0;
    for (int l = 0; l < level; ++l) off += 6 * (lvl_n(W, l) + 2) * (lvl_n(W, l) + 2);
    return off;
}
size_t pbrk_bordered_pyramid_texels(int W, int levels) { return pbrk_bordered_level_offset(W, levels); }

// gen_prefiltered_env_map.glsl:125-128
static inline void pitch_yaw(int i, int n, float* pitch, float* yaw) {
    float x = (float)i / (float)n;
    float y = (float)i / T_GOLDEN;
    *pitch = T_PI - acosf(x - 1.0f);
    *yaw = (2.0f * T_PI) * y;
}

void pbrk_host_sample_angles(int n, float* a) {
    for (int i = 0; i < n; ++i) {
        float pitch, yaw;
        pitch_yaw(i, n, &pitch, &yaw);
        a[4 * i + 0] = cosf(pitch); a[4 * i + 1] = sinf(pitch);
        a[4 * i + 2] = cosf(yaw);   a[4 * i + 3] = sinf(yaw);
    }
}

// DistributionBeckmann, gen_prefiltered_env_map.glsl:86-91
static inline float beckmann(float ndoth, float m) {
    float m2 = m * m;
    float a = tanf(acosf(ndoth));
    float n2 = ndoth * ndoth;
    return expf(-(a * a) / m2) / (T_PI * m2 * n2 * n2);
}

int pbrk_host_prefilter_table(int n, float roughness, float* t, float* alpha) {
    float dw = (2.0f * T_PI) / (float)n;                         // :122
    float asum = 0.0f;
    int count = 0;
    for (int i = 0; i < n; ++i) {
        float pitch, yaw;
        pitch_yaw(i, n, &pitch, &yaw);
        float cp = cosf(pitch), sp = sinf(pitch), cy = cosf(yaw), sy = sinf(yaw);
        float D = beckmann(cosf(pitch * 0.5f), roughness);       // :141
        asum += D * 1.0f * cp * dw;                              // alpha lane of :143
        float w = D * cp * dw;
        if (w != 0.0f) {
            // L = cp*R + sp*(cy*(T x R) + sy*T): the closed form of the two Rotate() calls (:132-133)
            t[4 * count + 0] = sp * cy;
            t[4 * count + 1] = sp * sy;
            t[4 * count + 2] = cp;
            t[4 * count + 3] = w;
            count++;
        }
    }
    if (alpha) *alpha = asum / T_PI;                             // :145
    return count;
}

int pbrk_host_irradiance_table(int n, float* t) {
    for (int i = 0; i < n; ++i) {
        float pitch, yaw;
        pitch_yaw(i, n, &pitch, &yaw);
        float cp = cosf(pitch), sp = sinf(pitch), cy = cosf(yaw), sy = sinf(yaw);
        t[4 * i + 0] = sp * cy;
        t[4 * i + 1] = sp * sy;
        t[4 * i + 2] = cp;
        t[4 * i + 3] = cp;                                       // gen_irradiance_map.glsl:95
    }
    return n;
}

}  // extern "C"
