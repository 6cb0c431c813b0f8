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
// given sample their source footprints overlap (L1/LDS friendly) and table reads are wave-uniform
// (scalar loads) when TX >= 64.  Partial sums are combined by a fixed LDS tree (deterministic).
#include "pbr_device.h"
#include "pbr_kernels.h"

template <int S>
__global__ __launch_bounds__(256) void k_mc_filter(const float4* __restrict__ src, int n_src,
                                                   const float4* __restrict__ tab, int n_tab,
                                                   float divisor, float alpha,
                                                   float4* __restrict__ out, int size,
                                                   int face0, int y0, int rows, int tiles_x, int tiles_per_face) {
    constexpr int TX = 256 / S;
    constexpr int TW = TX >= 16 ? 16 : TX;
    constexpr int TH = TX / TW;
    __shared__ float red[S > 1 ? 256 * 3 : 1];

    unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    int face = face0 + (int)(tile / (unsigned)tiles_per_face);
    int tf = (int)(tile % (unsigned)tiles_per_face);
    int ty = tf / tiles_x, tx = tf % tiles_x;

    int t = threadIdx.x % TX;
    int s = threadIdx.x / TX;
    if (TX >= 64) s = __builtin_amdgcn_readfirstlane(s);
    int x = tx * TW + (t % TW);
    int y = y0 + ty * TH + (t / TW);
    bool valid = (x < size) && (y < y0 + rows);
    int xc = min(x, size - 1), yc = min(y, y0 + rows - 1);

    f3 R = face_texel_dir(face, xc, yc, size);
    f3 T = tangent_of(R);
    f3 B = cross3(T, R);

    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    for (int i = s; i < n_tab; i += S) {
        float4 e = tab[i];
        f3 L;
        L.x = fmaf(e.x, B.x, fmaf(e.y, T.x, e.z * R.x));
        L.y = fmaf(e.x, B.y, fmaf(e.y, T.y, e.z * R.y));
        L.z = fmaf(e.x, B.z, fmaf(e.y, T.z, e.z * R.z));
        f3 c = cube_fetch_rgb<false>(src, n_src, L);
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
        o.x = ar / divisor; o.y = ag / divisor; o.z = ab / divisor; o.w = alpha;
        out[((size_t)face * size + y) * size + x] = o;
    }
}

template <int S>
static void launch_mc(const float4* src, int n_src, const float4* tab, int n_tab, float divisor, float alpha,
                      float4* out, int size, int face0, int nfaces, int y0, int rows, hipStream_t st) {
    constexpr int TX = 256 / S;
    constexpr int TW = TX >= 16 ? 16 : TX;
    constexpr int TH = TX / TW;
    int tiles_x = (size + TW - 1) / TW, tiles_y = (rows + TH - 1) / TH;
    int tiles_per_face = tiles_x * tiles_y;
    hipLaunchKernelGGL(k_mc_filter<S>, dim3((unsigned)(tiles_per_face * nfaces)), dim3(256), 0, st,
                       src, n_src, tab, n_tab, divisor, alpha, out, size, face0, y0, rows, tiles_x, tiles_per_face);
}

extern "C" int pbrk_mc_filter(const void* src_bordered_level, int n_src, const void* table4, int n_entries,
                              float divisor, float alpha, void* out, int out_size,
                              int face0, int face1, int y0, int y1, void* stream) {
    if (!src_bordered_level || !table4 || !out || n_src < 1 || out_size < 1 || n_entries < 0) return PBRK_E_ARG;
    if (face0 < 0 || face1 > 6 || face0 >= face1 || y0 < 0 || y1 > out_size || y0 >= y1) return PBRK_E_ARG;
    if (!(divisor != 0.0f)) return PBRK_E_ARG;
    const float4* src = (const float4*)src_bordered_level;
    const float4* tab = (const float4*)table4;
    float4* o = (float4*)out;
    int nfaces = face1 - face0, rows = y1 - y0;
    hipStream_t st = (hipStream_t)stream;
    // Sample-split factor S depends on the LEVEL size only (not on the dispatched sub-range), so that a
    // sharded dispatch sums in exactly the same order as a full one (bit-identical results).
    size_t texels = (size_t)6 * out_size * out_size;
    // enough workgroups to fill 256 CUs several times over: split the sample table when texels are few
    size_t want_blocks = 2048;
    int S = 1;
    while (S < 256 && texels * (size_t)S < want_blocks * 256) S <<= 1;
    switch (S) {
    case 1: launch_mc<1>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 2: launch_mc<2>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 4: launch_mc<4>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 8: launch_mc<8>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 16: launch_mc<16>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 32: launch_mc<32>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 64: launch_mc<64>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    case 128: launch_mc<128>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    default: launch_mc<256>(src, n_src, tab, n_entries, divisor, alpha, o, out_size, face0, nfaces, y0, rows, st); break;
    }
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
