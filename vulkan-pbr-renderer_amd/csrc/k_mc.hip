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
#include "k_mc_internal.h"

#include <stdlib.h>

// SNAP: the diagnostic cube-sampler convention (tap coordinates snapped to 1/256 texel, pbrk_set_cube_sampler_snap); a separate
// instantiation, so that the default kernel's inner loop is untouched
template <int S, int UNROLL, bool CELLS, bool SNAP = false>
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
    __amdgpu_buffer_rsrc_t rs = CELLS ? __builtin_amdgcn_make_buffer_rsrc((void*)p.cells, 0, (int)p.cells_bytes, 0x00020000)
                                      : __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
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
        f3 c = sample_bordered<CELLS>(rs, L, nf, off, nb, row_bytes, SNAP);
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

// ------------------------------------------------------------------------------------------
// LDS-resident variant: when the whole bordered source level fits in LDS (n_src <= 32: 111 KB of the CU's
// 160 KB), one 1024-thread workgroup per CU copies it in once and serves every tap with ds_read_b128
// (256 B/clk/CU) instead of vector-memory instructions (16 clk each): the kernel becomes VALU-bound.
// ------------------------------------------------------------------------------------------
// 64 VGPRs at most: two of these 16-wave workgroups then share a CU whenever the level leaves room in LDS (8 waves per SIMD;
// at 68 VGPRs only one fitted and the VALU-bound loop issued at 4.4 instead of 3 clk per instruction: mip 4 4.5 -> 4.0 ms)
template <int S>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mc_filter_lds(const McArgs p) {
    constexpr int BLOCK = 1024;
    constexpr int TX = BLOCK / S;
    constexpr int TW = TX >= 32 ? 32 : TX;
    constexpr int TH = TX / TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
    float4* lvl = (float4*)smem_l;
    const int nb = p.n_src + 2;
    const int n_texels = 6 * nb * nb;
    float* red = (float*)(lvl + n_texels);

    for (int i = threadIdx.x; i < n_texels; i += BLOCK) lvl[i] = p.src[i];

    unsigned tile = blockIdx.x;
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
    const float nf = (float)p.n_src, nbf = (float)nb;
    const float off = 0.5f * nf + 0.5f;
    const float4* __restrict__ tab = p.tab;
    __syncthreads();

    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    auto one_sample = [&](float ex, float ey, float ez, float ew) {
        f3 L;
        L.x = fmaf(ex, B.x, fmaf(ey, T.x, ez * R.x));
        L.y = fmaf(ex, B.y, fmaf(ey, T.y, ez * R.y));
        L.z = fmaf(ex, B.z, fmaf(ey, T.z, ez * R.z));
        float fid = __builtin_amdgcn_cubeid(L.x, L.y, L.z);
        float sc = __builtin_amdgcn_cubesc(L.x, L.y, L.z);
        float tc = __builtin_amdgcn_cubetc(L.x, L.y, L.z);
        float h = __builtin_amdgcn_rcpf(fabsf(__builtin_amdgcn_cubema(L.x, L.y, L.z))) * nf;
        float u = fmaf(sc, h, off), v = fmaf(tc, h, off);
        float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
        // texel index in exact fp32 (the level has at most 6*34*34 texels): FMAs instead of conversions, integer clamps and integer
        // multiplies (1.45x-1.7x an FMA each); LDS reads are not range-checked, so the taps are kept inside the level as before
        float i0 = fminf(floorf(u), nf), j0 = fminf(floorf(v), nf);
        const float4* wp = lvl + (int)fmaf(fmaf(fminf(fid, 5.0f), nbf, j0), nbf, i0);
        float4 q00 = wp[0], q10 = wp[1], q01 = wp[nb], q11 = wp[nb + 1];
        float cr = lerp_fma(lerp_fma(q00.x, q10.x, a), lerp_fma(q01.x, q11.x, a), b);
        float cg = lerp_fma(lerp_fma(q00.y, q10.y, a), lerp_fma(q01.y, q11.y, a), b);
        float cb = lerp_fma(lerp_fma(q00.z, q10.z, a), lerp_fma(q01.z, q11.z, a), b);
        ar = fmaf(ew, cr, ar);
        ag = fmaf(ew, cg, ag);
        ab = fmaf(ew, cb, ab);
    };
#pragma unroll 4
    for (int i = s; i < p.n_tab; i += S) {
        float4 e = tab[i];
        one_sample(e.x, e.y, e.z, e.w);
    }

    if (S > 1) {
        red[(s * TX + t) * 3 + 0] = ar;
        red[(s * TX + t) * 3 + 1] = ag;
        red[(s * TX + t) * 3 + 2] = ab;
        __syncthreads();
        for (int stride = S / 2; stride >= 1; stride >>= 1) {
            if (s < stride) {
                int a2 = (s * TX + t) * 3, b2 = ((s + stride) * TX + t) * 3;
                red[a2 + 0] += red[b2 + 0];
                red[a2 + 1] += red[b2 + 1];
                red[a2 + 2] += red[b2 + 2];
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
static void launch_mc_lds_s(McArgs a, int nfaces, size_t lds, hipStream_t st) {
    constexpr int TX = 1024 / S;
    constexpr int TW = TX >= 32 ? 32 : TX;
    constexpr int TH = TX / TW;
    a.tiles_x = (a.size + TW - 1) / TW;
    int tiles_y = (a.rows + TH - 1) / TH;
    a.tiles_per_face = a.tiles_x * tiles_y;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_mc_filter_lds<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_mc_filter_lds<S>, dim3((unsigned)(a.tiles_per_face * nfaces)), dim3(1024), lds, st, a);
}

// returns false when the level does not fit in LDS
static bool launch_mc_lds(McArgs a, int nfaces, hipStream_t st) {
    if (g_mc_lds_mode < 0) { const char* e = getenv("PBR_MC_LDS"); g_mc_lds_mode = e ? atoi(e) : 1; }
    const int mode = g_mc_lds_mode;
    int nb = a.n_src + 2;
    size_t lvl_bytes = (size_t)6 * nb * nb * 16;
    if (!mode || lvl_bytes > 112 * 1024) return false;
    // Measured (C4 / C2 on MI355X): with a 111 KB level only one 1024-thread workgroup fits per CU (16 waves) and the
    // VALU-bound loop runs at ~4.4 clk/instruction; the direct kernel (32 waves/CU, 3 loads/sample) then wins on big
    // outputs.  Small levels (several workgroups per CU) and small outputs (launch-limited) win with LDS.
    if (lvl_bytes > 48 * 1024 && (size_t)6 * a.size * a.size > ((size_t)1 << 19)) return false;
    // S from the level size only (deterministic under sharding): at least 512 workgroups of 1024 threads
    size_t texels = (size_t)6 * a.size * a.size;
    int S = 1;
    while (S < 1024 && texels * (size_t)S < (size_t)512 * 1024) S <<= 1;
    size_t lds = lvl_bytes + (S > 1 ? (size_t)1024 * 3 * 4 : 0);
    switch (S) {
    case 1: launch_mc_lds_s<1>(a, nfaces, lds, st); break;
    case 2: launch_mc_lds_s<2>(a, nfaces, lds, st); break;
    case 4: launch_mc_lds_s<4>(a, nfaces, lds, st); break;
    case 8: launch_mc_lds_s<8>(a, nfaces, lds, st); break;
    case 16: launch_mc_lds_s<16>(a, nfaces, lds, st); break;
    case 32: launch_mc_lds_s<32>(a, nfaces, lds, st); break;
    case 64: launch_mc_lds_s<64>(a, nfaces, lds, st); break;
    case 128: launch_mc_lds_s<128>(a, nfaces, lds, st); break;
    case 256: launch_mc_lds_s<256>(a, nfaces, lds, st); break;
    case 512: launch_mc_lds_s<512>(a, nfaces, lds, st); break;
    default: launch_mc_lds_s<1024>(a, nfaces, lds, st); break;
    }
    return true;
}

template <int S>
static void launch_mc(McArgs a, int nfaces, hipStream_t st) {
    constexpr int TX = 256 / S;
    constexpr int TW = TX >= 16 ? 16 : TX;
    constexpr int TH = TX / TW;
    a.tiles_x = (a.size + TW - 1) / TW;
    int tiles_y = (a.rows + TH - 1) / TH;
    a.tiles_per_face = a.tiles_x * tiles_y;
    dim3 grid((unsigned)(a.tiles_per_face * nfaces));
    if (a.snap) hipLaunchKernelGGL((k_mc_filter<S, 4, false, true>), grid, dim3(256), 0, st, a);
    else if (a.cells) hipLaunchKernelGGL((k_mc_filter<S, 4, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_mc_filter<S, 4, false>), grid, dim3(256), 0, st, a);
}

// ---- cells build: bordered level -> 2x2-footprint cells (pbr_device.h: cells_bilerp) ----
__global__ __launch_bounds__(256) void k_cells_build(const float4* __restrict__ b, float4* __restrict__ cells, int n) {
    int nb = n + 2, nc = n + 1;
    int i = blockIdx.x * 64 + (threadIdx.x & 63);
    int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (i >= nc || j >= nc) return;
    const float4* p = b + ((size_t)f * nb + j) * nb + i;
    float4 t00 = p[0], t10 = p[1], t01 = p[nb], t11 = p[nb + 1];
    float4* o = cells + ((size_t)(f * nc + j) * nc + i) * PBR_CELL_F4;
    // coefficient form {t00, t10 - t00, t01, t11 - t01}: the differences are the (rounded) ones lerp_fma would form per fetch
    o[0] = make_float4(t00.x, t00.y, t00.z, t10.x - t00.x);
    o[1] = make_float4(t10.y - t00.y, t10.z - t00.z, t01.x, t01.y);
    o[2] = make_float4(t01.z, t11.x - t01.x, t11.y - t01.y, t11.z - t01.z);
}

extern "C" size_t pbrk_cells_bytes(int n) { return (size_t)6 * (n + 1) * (n + 1) * PBR_CELL_BYTES; }

extern "C" int pbrk_cells_build(const void* bordered_level, int n, void* cells, void* stream) {
    if (!bordered_level || !cells || n < 1) return PBRK_E_ARG;
    hipLaunchKernelGGL(k_cells_build, dim3((n + 1 + 63) / 64, (n + 1 + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream, (const float4*)bordered_level, (float4*)cells, n);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_mc_filter(const void* src_bordered_level, const void* src_cells, int n_src, const void* table4, int n_entries,
                              float divisor, float alpha, void* out, int out_size,
                              int face0, int face1, int y0, int y1, void* stream) {
    if (!src_bordered_level || !table4 || !out || n_src < 1 || out_size < 1 || n_entries < 0) return PBRK_E_ARG;
    if (face0 < 0 || face1 > 6 || face0 >= face1 || y0 < 0 || y1 > out_size || y0 >= y1) return PBRK_E_ARG;
    if (!(divisor != 0.0f)) return PBRK_E_ARG;
    size_t src_bytes = (size_t)6 * (n_src + 2) * (n_src + 2) * 16;
    if (src_bytes > 0x7FFFFFFFu) return PBRK_E_ARG;             // 32-bit buffer offsets (n_src <= 4727)
    McArgs a;
    a.src = (const float4*)src_bordered_level; a.n_src = n_src; a.src_bytes = (unsigned)src_bytes;
    a.cells = (const float4*)src_cells; a.cells_bytes = (unsigned)pbrk_cells_bytes(n_src);
    if (src_cells && n_src > 512) return PBRK_E_ARG;            // the kernel forms cell offsets in exact fp32 (n + 1 <= 513)
    a.tab = (const float4*)table4; a.n_tab = n_entries;
    a.divisor = divisor; a.alpha = alpha;
    a.out = (float4*)out; a.size = out_size;
    a.face0 = face0; a.y0 = y0; a.rows = y1 - y0;
    int nfaces = face1 - face0;
    hipStream_t st = (hipStream_t)stream;
    a.snap = pbrk_get_cube_sampler_snap();                      // diagnostic convention: the direct kernel only
    // Kernel choice and the sample-split factor S depend on the LEVEL size only
    if (!a.snap && launch_mc_region(a, nfaces, st)) return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    if (!a.snap && launch_mc_lds(a, nfaces, st)) return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    // Sample-split factor S depends on the LEVEL size only (not on the dispatched sub-range), so that a
    // sharded dispatch sums in exactly the same order as a full one (bit-identical results).
    size_t texels = (size_t)6 * out_size * out_size;
    // Enough workgroups to fill 256 CUs several times over: split the sample table when texels are few.  Never fewer than 4
    // slices: a 256-texel workgroup walking all 8192 samples runs for ~5.6 ms, and a share of the level (one rank's tiles)
    // that is only a few such rounds long loses up to a round in its tail; 4 slices of 64 texels (one wave each, table
    // entries still wave-uniform scalar loads) cost nothing and cut that tail by four (8-way share: 18.1 -> 17.3 ms mean).
    size_t want_blocks = 2048;
    int S = 4;
    while (S < 256 && texels * (size_t)S < want_blocks * 256) S <<= 1;
    switch (S) {
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
