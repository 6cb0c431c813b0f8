// k_lut.hip -- K1: split-sum BRDF integration map (gfx950).
//
// Replaces shaders/gen_brdf_integration_map.glsl:142-210 (Beckmann D :34-39, Mikkelsen G :57-59,
// Schlick Fc :196; uniform Fibonacci-hemisphere quadrature :170-179) as dispatched by
// render.cpp:607-613 (whose 6 z-slices all write the same texel; computed once here).
//
// The quadrature is numerically ill-conditioned at low roughness: the Beckmann lobe is a function of
// 1 - N.H, so a one-ulp change of H moves a sample's weight by percents.  Everything that feeds N.H
// therefore keeps the shader's operation order with correctly rounded results (Rotate, normalize; sample
// angles and the per-column view angle come from host tables); what follows N.H is continuous and is
// evaluated in the cheapest accurate form: tan^2(acos n) = (1 - n^2) / n^2 with 1 - n^2 from one FMA
// (a relative error of 2^-24 where the libm route also carries ~2 ulp), v_exp_f32, x^5 by three
// multiplications, reciprocals instead of divisions.
//
// Work decomposition (north_star: wavefront-shuffle reductions).  The sample directions L_i depend on i only, so a 512-thread
// workgroup first builds them in LDS (64 KB per chunk of 4096: the shader's two Rotate() calls per sample, bit for bit).  And of a sample's
// integrand only the Beckmann exponential depends on the row (roughness): H, N.H, V.H, G's two products and the Fresnel
// term are functions of (i, column).  A wave therefore owns one column x 16 rows: its 64 lanes take samples l, l + 64, ..., evaluate
// the column part once per sample (~45 instructions) and only `exp2(a2 * k_row) * rn2^2 * rpm_row * G * kw` plus the two FMAs per
// row (8 instructions) -- every product in the order the one-texel-at-a-time form had, so the results are bit-identical to it
// (round 2a: 60 instructions per sample and texel, 0.50 ms).  The 32 sums are folded with a fixed xor-butterfly (deterministic; a
// row-sharded dispatch equals a full one bit for bit).  512 workgroups for a 256^2 map.
#include "pbr_device.h"
#include "pbr_kernels.h"
#include <hip/hip_fp16.h>

#define LUT_BLOCK 512
#define LUT_ROWS 16            // rows of one column per wave
#define LUT_CHUNK 4096         // samples whose directions sit in LDS at a time (64 KB; a multiple of 64)

// EXACT: Rotate() of gen_brdf_integration_map.glsl:61-64
__device__ __forceinline__ f3 rotate_exact(f3 v, f3 n, float c, float s) {
    float d = dot3(v, n);
    f3 a = scale3(sub3(v, scale3(n, d)), c);
    f3 b = scale3(cross3(n, v), s);
    f3 cc = scale3(n, d);
    return add3(add3(a, b), cc);
}

__global__ __launch_bounds__(LUT_BLOCK) void k_brdf_lut(void* __restrict__ out, int fmt, int size, int nsamples,
                                                        const float4* __restrict__ angles, const float2* __restrict__ view_cs,
                                                        int y0, int y1) {
    extern __shared__ __attribute__((aligned(16))) float4 Ltab[];        // L_i (xyz) of the current chunk of LUT_CHUNK samples
    const f3 N = mk3(0.0f, 0.0f, 1.0f);
    const f3 X = mk3(1.0f, 0.0f, 0.0f);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = (int)blockIdx.x * (LUT_BLOCK / 64) + wave;          // (row block, column), columns fastest
    const int x = item % size, yb = y0 + (item / size) * LUT_ROWS;
    const bool active = yb < y1;                             // wave-uniform; idle waves still fill the table and meet the barriers
    const float NdotV = ((float)x + 0.5f) / (float)size;     // :143, :154
    const float2 vcs = view_cs[x];
    const f3 V = rotate_exact(N, X, vcs.x, vcs.y);           // :160  = (0, -sin, cos)
    const float dw = 2 * PBR_PI / (float)nsamples;           // :168
    const float kw = dw * __builtin_amdgcn_rcpf(4.0f * NdotV);
    float k_exp[LUT_ROWS], rpm[LUT_ROWS], scale[LUT_ROWS], bias[LUT_ROWS];
#pragma unroll
    for (int r = 0; r < LUT_ROWS; ++r) {
        const float rough = ((float)(yb + r) + 0.5f) / (float)size;      // :155
        const float m2 = rough * rough;
        k_exp[r] = -1.4426950408889634f / m2;                // exp(-a2 / m2) = 2^(a2 * k_exp)
        rpm[r] = __builtin_amdgcn_rcpf(PBR_PI * m2);
        scale[r] = 0.0f; bias[r] = 0.0f;
    }
    // The sample table goes through LDS in chunks of LUT_CHUNK (a multiple of 64: a lane's samples l, l + 64, ... keep their order
    // across chunks, so any sample count gives the sums the one-chunk form would); the reference's 4096 samples are one chunk.
    for (int c0 = 0; c0 < nsamples; c0 += LUT_CHUNK) {
        const int cn = min(LUT_CHUNK, nsamples - c0);
        if (c0 > 0) __syncthreads();                         // readers of the previous chunk are done
        for (int i = threadIdx.x; i < cn; i += LUT_BLOCK) {
            float4 a = angles[c0 + i];
            f3 L = rotate_exact(N, X, a.x, a.y);                 // :177
            L = rotate_exact(L, N, a.z, a.w);                    // :178
            Ltab[i] = make_float4(L.x, L.y, L.z, 0.0f);
        }
        __syncthreads();
        if (!active) continue;
        for (int i = lane; i < cn; i += 64) {
            const float4 Lq = Ltab[i];
            const f3 L = mk3(Lq.x, Lq.y, Lq.z);
            const f3 H = normalize3_nr(add3(L, V));              // :179 (correctly rounded: feeds N.H)
            const float NdotL = dot3(N, L);
            const float NdotH = dot3(N, H);
            const float VdotH = dot3(V, H);
            // :34-39 Beckmann: tan^2(acos n) = (1 - n^2) / n^2
            const float n2 = NdotH * NdotH;
            const float rn2 = __builtin_amdgcn_rcpf(n2);
            const float a2 = fmaf(-NdotH, NdotH, 1.0f) * rn2;
            const float rn4 = rn2 * rn2;
            // :57-59 Mikkelsen
            const float t2 = 2.0f * NdotH * __builtin_amdgcn_rcpf(VdotH);
            const float G = fminf(1.0f, fminf(t2 * NdotV, t2 * NdotL));      // :193
            const float q = 1.0f - VdotH, q2 = q * q;
            const float Fc = q2 * q2 * q;                        // :196 pow(1 - VdotH, 5.)
            const float omFc = 1.0f - Fc;
#pragma unroll
            for (int r = 0; r < LUT_ROWS; ++r) {                 // the row part: the products in the order D * G * kw had as one expression
                const float D = __builtin_amdgcn_exp2f(a2 * k_exp[r]) * rn4 * rpm[r];
                const float w = D * G * kw;                      // :198-199
                scale[r] = fmaf(w, omFc, scale[r]);
                bias[r] = fmaf(w, Fc, bias[r]);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < LUT_ROWS; ++r) {
        float sc = scale[r], bi = bias[r];
        // fixed xor-butterfly over the 64 lanes
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sc += __shfl_xor(sc, o);
            bi += __shfl_xor(bi, o);
        }
        if (lane == 0 && yb + r < y1) {
            size_t o = (size_t)(yb + r) * size + x;
            if (fmt == PBRK_FMT_RG16F) ((__half2*)out)[o] = __halves2half2(__float2half_rn(sc), __float2half_rn(bi));
            else if (fmt == PBRK_FMT_RG32F) ((float2*)out)[o] = make_float2(sc, bi);
            else ((float4*)out)[o] = make_float4(sc, bi, 0.0f, 1.0f);   // :209
        }
    }
}

extern "C" int pbrk_brdf_lut(void* out, int out_format, int size, int nsamples, const void* angles4,
                             const void* view_cs, int y0, int y1, void* stream) {
    if (!out || !angles4 || !view_cs || size < 1 || nsamples < 1) return PBRK_E_ARG;
    if (y0 < 0 || y1 > size || y0 >= y1) return PBRK_E_ARG;
    if (out_format != PBRK_FMT_RG16F && out_format != PBRK_FMT_RG32F && out_format != PBRK_FMT_RGBA32F) return PBRK_E_FORMAT;
    size_t lds = (size_t)(nsamples < LUT_CHUNK ? nsamples : LUT_CHUNK) * 16;     // any sample count: the table passes through LDS in chunks
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_brdf_lut, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    const int row_blocks = (y1 - y0 + LUT_ROWS - 1) / LUT_ROWS;
    const int items = row_blocks * size, per_wg = LUT_BLOCK / 64;
    hipLaunchKernelGGL(k_brdf_lut, dim3((unsigned)((items + per_wg - 1) / per_wg)), dim3(LUT_BLOCK), lds, (hipStream_t)stream,
                       out, out_format, size, nsamples, (const float4*)angles4, (const float2*)view_cs, y0, y1);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
