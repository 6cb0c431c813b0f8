// k_mc.hip -- K4b / K3: Monte-Carlo hemisphere filter of an environment level (gfx950).
//
//   out.rgb(texel) = ( sum_i w_i * bilinear(src, l_i.x*(T x R) + l_i.y*T + l_i.z*R) ) / divisor
//
// R = direction through the texel centre, T = normalize(cross(R, some_vector)); (l_i, w_i) is the
// host-built table (pbr_tables.cpp).  This is the closed form of the two Rotate() calls per sample in
// shaders/gen_prefiltered_env_map.glsl:124-144 (weights D*cos*dw, divisor PI) and
// shaders/gen_irradiance_map.glsl:84-97 (weights cos, divisor N).
//
// Work decomposition: a 256-thread workgroup owns TX = 256/S output texels (a TW x TH tile of one
// face) and S interleaved slices of the sample table; lanes of a wave are adjacent texels, so for a
// given sample their source footprints overlap (cache friendly) and table reads are wave-uniform
// (scalar loads) when TX >= 64.  Partial sums are combined by a fixed LDS tree (deterministic).
//
// Inner loop (per sample, per lane): 9 FMA frame transform, v_cube{id,sc,tc,ma}_f32 face selection
// (same table as the Vulkan one quoted in gen_prefiltered_env_map.glsl:12-23, ties z > y > x),
// one v_rcp_f32, 2 FMA projection onto the bordered level, v_fract/v_cvt for tap + weights, 4
// range-checked buffer_load_dwordx3 (32-bit offsets, no clamps needed), 9 lerps, 3 FMA accumulate.
#include "pbr_device.h"
#include "pbr_kernels.h"

#include <stdlib.h>

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

struct McArgs {
    const float4* src; int n_src; unsigned src_bytes;
    const float4* tab; int n_tab;
    float divisor, alpha;
    float4* out; int size;
    int face0, y0, rows, tiles_x, tiles_per_face;
};

__device__ __forceinline__ f3 tap_rgb(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
    return mk3(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
}

// one sample: direction L -> bilinear RGB of the bordered level behind `rs`
__device__ __forceinline__ f3 sample_bordered(__amdgpu_buffer_rsrc_t rs, f3 L, float nf, float off, int nb, int row_bytes) {
    float fid = __builtin_amdgcn_cubeid(L.x, L.y, L.z);
    float sc = __builtin_amdgcn_cubesc(L.x, L.y, L.z);
    float tc = __builtin_amdgcn_cubetc(L.x, L.y, L.z);
    float ma2 = __builtin_amdgcn_cubema(L.x, L.y, L.z);          // 2 * major axis
    float h = __builtin_amdgcn_rcpf(fabsf(ma2)) * nf;            // n / (2 |rc|)
    float u = fmaf(sc, h, off);                                  // s*n - 0.5 + 1 (bordered), in [0.5, n + 0.5]
    float v = fmaf(tc, h, off);
    float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
    int i0 = (int)u, j0 = (int)v, face = (int)fid;
    int texel = (face * nb + j0) * nb + i0;
    int voff = texel << 4;
    f3 t00 = tap_rgb(rs, voff, 0), t10 = tap_rgb(rs, voff + 16, 0);
    f3 t01 = tap_rgb(rs, voff, row_bytes), t11 = tap_rgb(rs, voff + 16, row_bytes);
    f3 r;
    r.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
    r.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
    r.z = lerp_fma(lerp_fma(t00.z, t10.z, a), lerp_fma(t01.z, t11.z, a), b);
    return r;
}

template <int S, int UNROLL>
__global__ __launch_bounds__(256) void k_mc_filter(const McArgs p) {
    constexpr int TX = 256 / S;
    constexpr int TW = TX >= 16 ? 16 : TX;
    constexpr int TH = TX / TW;
    __shared__ float red[S > 1 ? 256 * 3 : 1];

    unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    int face = p.face0 + (int)(tile / (unsigned)p.tiles_per_face);
    int tf = (int)(tile % (unsigned)p.tiles_per_face);
    int ty = tf / p.tiles_x, tx = tf % p.tiles_x;

    int t = threadIdx.x % TX;
    int s = threadIdx.x / TX;
    if (TX >= 64) s = __builtin_amdgcn_readfirstlane(s);
    int x = tx * TW + (t % TW);
    int y = p.y0 + ty * TH + (t / TW);
    bool valid = (x < p.size) && (y < p.y0 + p.rows);
    int xc = min(x, p.size - 1), yc = min(y, p.y0 + p.rows - 1);

    f3 R = face_texel_dir(face, xc, yc, p.size);
    f3 T = tangent_of(R);
    f3 B = cross3(T, R);

    // wave-uniform buffer descriptor over the bordered source level: out-of-range taps read 0, never fault
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
    const int nb = p.n_src + 2;
    const int row_bytes = nb * 16;
    const float nf = (float)p.n_src;
    const float off = 0.5f * nf + 0.5f;
    const float4* __restrict__ tab = p.tab;

    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
#pragma unroll UNROLL
    for (int i = s; i < p.n_tab; i += S) {
        float4 e = tab[i];
        f3 L;
        L.x = fmaf(e.x, B.x, fmaf(e.y, T.x, e.z * R.x));
        L.y = fmaf(e.x, B.y, fmaf(e.y, T.y, e.z * R.y));
        L.z = fmaf(e.x, B.z, fmaf(e.y, T.z, e.z * R.z));
        f3 c = sample_bordered(rs, L, nf, off, nb, row_bytes);
        ar = fmaf(e.w, c.x, ar);
        ag = fmaf(e.w, c.y, ag);
        ab = fmaf(e.w, c.z, ab);
    }

    if (S > 1) {
        red[(s * TX + t) * 3 + 0] = ar;
        red[(s * TX + t) * 3 + 1] = ag;
        red[(s * TX + t) * 3 + 2] = ab;
        __syncthreads();
        for (int stride = S / 2; stride >= 1; stride >>= 1) {
            if (s < stride) {
                int a = (s * TX + t) * 3, b = ((s + stride) * TX + t) * 3;
                red[a + 0] += red[b + 0];
                red[a + 1] += red[b + 1];
                red[a + 2] += red[b + 2];
            }
            __syncthreads();
        }
        ar = red[t * 3 + 0]; ag = red[t * 3 + 1]; ab = red[t * 3 + 2];
    }
    if (valid && s == 0) {
        float4 o;
        o.x = ar / p.divisor; o.y = ag / p.divisor; o.z = ab / p.divisor; o.w = p.alpha;
        p.out[((size_t)face * p.size + y) * p.size + x] = o;
    }
}

template <int S>
static void launch_mc(McArgs a, int nfaces, hipStream_t st) {
    constexpr int TX = 256 / S;
    constexpr int TW = TX >= 16 ? 16 : TX;
    constexpr int TH = TX / TW;
    a.tiles_x = (a.size + TW - 1) / TW;
    int tiles_y = (a.rows + TH - 1) / TH;
    a.tiles_per_face = a.tiles_x * tiles_y;
    static int unroll = -1;
    if (unroll < 0) { const char* e = getenv("PBR_MC_UNROLL"); unroll = e ? atoi(e) : 4; }
    dim3 grid((unsigned)(a.tiles_per_face * nfaces));
    switch (unroll) {
    case 1: hipLaunchKernelGGL((k_mc_filter<S, 1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((k_mc_filter<S, 2>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((k_mc_filter<S, 8>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((k_mc_filter<S, 4>), grid, dim3(256), 0, st, a); break;
    }
}

extern "C" int pbrk_mc_filter(const void* src_bordered_level, int n_src, const void* table4, int n_entries,
                              float divisor, float alpha, void* out, int out_size,
                              int face0, int face1, int y0, int y1, void* stream) {
    if (!src_bordered_level || !table4 || !out || n_src < 1 || out_size < 1 || n_entries < 0) return PBRK_E_ARG;
    if (face0 < 0 || face1 > 6 || face0 >= face1 || y0 < 0 || y1 > out_size || y0 >= y1) return PBRK_E_ARG;
    if (!(divisor != 0.0f)) return PBRK_E_ARG;
    size_t src_bytes = (size_t)6 * (n_src + 2) * (n_src + 2) * 16;
    if (src_bytes > 0x7FFFFFFFu) return PBRK_E_ARG;             // 32-bit buffer offsets (n_src <= 4727)
    McArgs a;
    a.src = (const float4*)src_bordered_level; a.n_src = n_src; a.src_bytes = (unsigned)src_bytes;
    a.tab = (const float4*)table4; a.n_tab = n_entries;
    a.divisor = divisor; a.alpha = alpha;
    a.out = (float4*)out; a.size = out_size;
    a.face0 = face0; a.y0 = y0; a.rows = y1 - y0;
    int nfaces = face1 - face0;
    hipStream_t st = (hipStream_t)stream;
    // Sample-split factor S depends on the LEVEL size only (not on the dispatched sub-range), so that a
    // sharded dispatch sums in exactly the same order as a full one (bit-identical results).
    size_t texels = (size_t)6 * out_size * out_size;
    // enough workgroups to fill 256 CUs several times over: split the sample table when texels are few
    size_t want_blocks = 2048;
    int S = 1;
    while (S < 256 && texels * (size_t)S < want_blocks * 256) S <<= 1;
    switch (S) {
    case 1: launch_mc<1>(a, nfaces, st); break;
    case 2: launch_mc<2>(a, nfaces, st); break;
    case 4: launch_mc<4>(a, nfaces, st); break;
    case 8: launch_mc<8>(a, nfaces, st); break;
    case 16: launch_mc<16>(a, nfaces, st); break;
    case 32: launch_mc<32>(a, nfaces, st); break;
    case 64: launch_mc<64>(a, nfaces, st); break;
    case 128: launch_mc<128>(a, nfaces, st); break;
    default: launch_mc<256>(a, nfaces, st); break;
    }
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
