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

template <int S, int UNROLL, bool CELLS>
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
        f3 c = sample_bordered<CELLS>(rs, L, nf, off, nb, row_bytes);
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

__device__ __forceinline__ void cube_project(f3 L, float nf, float off, int* face, float* u, float* v) {
    float fid = __builtin_amdgcn_cubeid(L.x, L.y, L.z);
    float sc = __builtin_amdgcn_cubesc(L.x, L.y, L.z);
    float tc = __builtin_amdgcn_cubetc(L.x, L.y, L.z);
    float ma2 = __builtin_amdgcn_cubema(L.x, L.y, L.z);
    float h = __builtin_amdgcn_rcpf(fabsf(ma2)) * nf;
    *u = fmaf(sc, h, off);
    *v = fmaf(tc, h, off);
    *face = (int)fid;
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
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("PBR_MC_LDS"); mode = e ? atoi(e) : 1; }
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

// ==========================================================================================
// Binned variant for large output levels (one 16x16 output tile per workgroup, S = 1).
//
// Measured on MI355X: the direct kernel above is bound by vector-memory INSTRUCTION issue (~16 clk per
// wave-level load whatever its width or lane mask; 4 taps per sample), not by VALU or bandwidth.  A
// 16x16 output tile has nearly one tangent frame, so sample k lands in nearly the same place of the source
// level for all 256 texels.  Each workgroup therefore
//   1. bins the sample table by the 16x16-texel source WINDOW the tile-centre frame sends each sample to
//      (LDS histogram -> scan -> scatter -> per-window insertion sort: deterministic order),
//   2. for every non-empty window stages the window plus a margin of M texels in LDS (one coalesced pass),
//   3. runs the window's samples with all four taps served by ds_read_b128 from LDS.
// Lanes whose own frame puts a sample outside the staged window (frame twist near the pole of
// `some_vector`, cube-edge crossings) take the direct buffer-load path for that sample: same taps, same
// arithmetic, so the result does not depend on which path served it.
// Tiles are anchored at multiples of 16 rows (not at the dispatched row range), so a sharded dispatch
// reproduces a full one bit for bit.
// ==========================================================================================
#define BIN_WC 16          // window core, in tap coordinates of the bordered level

struct BinArgs {
    McArgs a;
    int M;                 // margin (texels) staged around the window core
    int WS;                // staged window edge = BIN_WC + 2*M + 1
    int G;                 // windows per face edge = n_src / 16 + 1
    int NW;                // 6 * G * G
    int NP;                // n_tab rounded up to a power of two
    int region_bytes;      // bytes of the keys / list + window region
    int tile_y0;           // first tile row (multiple of 16) covering a.y0
    unsigned long long* stats;   // optional: [0] += wave-samples served by the fallback path, [1] += all wave-samples
};


__device__ __forceinline__ void taps_lds(const float4* __restrict__ win, int WS, int lx, int ly, f3& t00, f3& t10, f3& t01, f3& t11) {
    const float4* wp = win + (ly * WS + lx);
    float4 q00 = wp[0], q10 = wp[1], q01 = wp[WS], q11 = wp[WS + 1];
    t00 = mk3(q00.x, q00.y, q00.z); t10 = mk3(q10.x, q10.y, q10.z);
    t01 = mk3(q01.x, q01.y, q01.z); t11 = mk3(q11.x, q11.y, q11.z);
}
__device__ __forceinline__ void taps_mem(__amdgpu_buffer_rsrc_t rs, int nb, int row_bytes, int f, int i0, int j0,
                                         f3& t00, f3& t10, f3& t01, f3& t11) {
    int voff = ((f * nb + j0) * nb + i0) << 4;
    t00 = tap_rgb(rs, voff, 0); t10 = tap_rgb(rs, voff + 16, 0);
    t01 = tap_rgb(rs, voff, row_bytes); t11 = tap_rgb(rs, voff + 16, row_bytes);
}

__global__ __launch_bounds__(256) void k_mc_binned(const BinArgs q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const McArgs& p = q.a;
    const int n_tab = p.n_tab;
    const int NP = q.NP;                                   // n_tab rounded up to a power of two (bitonic sort)
    // LDS carve.  Binning phase: keys[NP] (u32: window << 16 | sample).  Main phase, same bytes: list[NP] (u16) at
    // offset 0, window buffer at offset 2*NP.  Then wstart[NW], wcount[NW].
    unsigned* keys = (unsigned*)smem;
    unsigned short* list = (unsigned short*)smem;
    float4* win = (float4*)(smem + 2 * NP);
    unsigned* wstart = (unsigned*)(smem + q.region_bytes);
    unsigned* wcount = wstart + q.NW;

    const int tid = threadIdx.x;
    unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    int face = p.face0 + (int)(tile / (unsigned)p.tiles_per_face);
    int tf = (int)(tile % (unsigned)p.tiles_per_face);
    int ty = tf / p.tiles_x, tx = tf % p.tiles_x;
    int x = tx * 16 + (tid & 15);
    int y = q.tile_y0 + ty * 16 + (tid >> 4);
    bool valid = (x < p.size) && (y >= p.y0) && (y < p.y0 + p.rows);
    int xc = min(x, p.size - 1), yc = min(y, p.size - 1);

    f3 R = face_texel_dir(face, xc, yc, p.size);
    f3 T = tangent_of(R);
    f3 B = cross3(T, R);
    // tile-centre frame (wave-uniform values, evaluated redundantly per lane)
    f3 Rc = face_texel_dir(face, min(tx * 16 + 8, p.size - 1), min(q.tile_y0 + ty * 16 + 8, p.size - 1), p.size);
    f3 Tc = tangent_of(Rc);
    f3 Bc = cross3(Tc, Rc);

    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
    const int nb = p.n_src + 2;
    const int row_bytes = nb * 16;
    const float nf = (float)p.n_src;
    const float off = 0.5f * nf + 0.5f;
    const float4* __restrict__ tab = p.tab;
    const int G = q.G, NW = q.NW, WS = q.WS, M = q.M;

    // ---- 1. bin the samples by window (tile-centre frame): keys + histogram, bitonic sort, scan ----
    for (int w = tid; w < NW; w += 256) wcount[w] = 0;
    __syncthreads();
    for (int k = tid; k < NP; k += 256) {
        unsigned key = 0xFFFFFFFFu;
        if (k < n_tab) {
            float4 e = tab[k];
            f3 L;
            L.x = fmaf(e.x, Bc.x, fmaf(e.y, Tc.x, e.z * Rc.x));
            L.y = fmaf(e.x, Bc.y, fmaf(e.y, Tc.y, e.z * Rc.y));
            L.z = fmaf(e.x, Bc.z, fmaf(e.y, Tc.z, e.z * Rc.z));
            int f; float u, v;
            cube_project(L, nf, off, &f, &u, &v);
            int ci = min(max((int)u, 0) >> 4, G - 1), cj = min(max((int)v, 0) >> 4, G - 1);
            int w = (min(max(f, 0), 5) * G + cj) * G + ci;
            atomicAdd(&wcount[w], 1u);                     // counts do not depend on arrival order
            key = ((unsigned)w << 16) | (unsigned)k;
        }
        keys[k] = key;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= NP; k2 <<= 1) {                 // deterministic order: by window, then by sample index
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += 256) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned ka = keys[i], kb = keys[ixj];
                    bool asc = (i & k2) == 0;
                    if ((ka > kb) == asc) { keys[i] = kb; keys[ixj] = ka; }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        unsigned acc = 0;
        for (int w = 0; w < NW; ++w) { wstart[w] = acc; acc += wcount[w]; }
    }
    {   // compact the sorted keys to 16-bit sample indices in place (frees the upper half for the window buffer)
        unsigned short tmp[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) { int i = tid + r * 256; tmp[r] = i < NP ? (unsigned short)(keys[i] & 0xFFFFu) : 0; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; ++r) { int i = tid + r * 256; if (i < NP) list[i] = tmp[r]; }
    }
    __syncthreads();

    // ---- 2./3. windows ----
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    unsigned long long slow = 0, total = 0;
    for (int w = 0; w < NW; ++w) {
        const unsigned cnt = wcount[w];
        if (cnt == 0) continue;                            // workgroup-uniform
        const unsigned base = wstart[w];
        const int fw = w / (G * G);
        const int cj = (w / G) % G, ci = w % G;
        const int oi = ci * BIN_WC - M, oj = cj * BIN_WC - M;
        __syncthreads();                                   // readers of the previous window are done
        for (int t = tid; t < WS * WS; t += 256) {
            int wy = t / WS, wx = t - wy * WS;
            int si = oi + wx, sj = oj + wy;
            bool in = (si >= 0) & (si < nb) & (sj >= 0) & (sj < nb);
            int voff = in ? (((fw * nb + sj) * nb + si) << 4) : 0x7FFFFFF0;      // out of range -> reads 0
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            win[t] = make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
        }
        __syncthreads();
        for (unsigned s = 0; s < cnt; s += 4) {
            // four samples per trip: table entries by scalar loads, 16 LDS tap reads in flight before the lerps
            float4 e[4]; float a[4], b[4]; int lx[4], ly[4], fj[4], i0[4], j0[4]; bool inw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned sj = min(s + (unsigned)j, cnt - 1u);
                int k = __builtin_amdgcn_readfirstlane((int)list[base + sj]);
                e[j] = tab[k];
                if (s + (unsigned)j >= cnt) e[j].w = 0.0f;          // tail: repeats the last sample with weight 0 (acc + 0*c == acc)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f3 L;
                L.x = fmaf(e[j].x, B.x, fmaf(e[j].y, T.x, e[j].z * R.x));
                L.y = fmaf(e[j].x, B.y, fmaf(e[j].y, T.y, e[j].z * R.y));
                L.z = fmaf(e[j].x, B.z, fmaf(e[j].y, T.z, e[j].z * R.z));
                float u, v;
                cube_project(L, nf, off, &fj[j], &u, &v);
                a[j] = __builtin_amdgcn_fractf(u); b[j] = __builtin_amdgcn_fractf(v);
                i0[j] = (int)u; j0[j] = (int)v;
                lx[j] = i0[j] - oi; ly[j] = j0[j] - oj;
                inw[j] = (fj[j] == fw) & ((unsigned)lx[j] < (unsigned)(WS - 1)) & ((unsigned)ly[j] < (unsigned)(WS - 1));
            }
            f3 t00[4], t10[4], t01[4], t11[4];
            bool all_in = inw[0] & inw[1] & inw[2] & inw[3];
            if (__builtin_amdgcn_ballot_w64(!all_in) == 0) {       // wave-uniform: every tap of the four samples is in LDS
#pragma unroll
                for (int j = 0; j < 4; ++j) taps_lds(win, WS, lx[j], ly[j], t00[j], t10[j], t01[j], t11[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (__builtin_amdgcn_ballot_w64(!inw[j]) == 0) taps_lds(win, WS, lx[j], ly[j], t00[j], t10[j], t01[j], t11[j]);
                    else { taps_mem(rs, nb, row_bytes, fj[j], i0[j], j0[j], t00[j], t10[j], t01[j], t11[j]); ++slow; }
                }
            }
            total += 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float cr = lerp_fma(lerp_fma(t00[j].x, t10[j].x, a[j]), lerp_fma(t01[j].x, t11[j].x, a[j]), b[j]);
                float cg = lerp_fma(lerp_fma(t00[j].y, t10[j].y, a[j]), lerp_fma(t01[j].y, t11[j].y, a[j]), b[j]);
                float cb = lerp_fma(lerp_fma(t00[j].z, t10[j].z, a[j]), lerp_fma(t01[j].z, t11[j].z, a[j]), b[j]);
                ar = fmaf(e[j].w, cr, ar);
                ag = fmaf(e[j].w, cg, ag);
                ab = fmaf(e[j].w, cb, ab);
            }
        }
    }
    if (q.stats && (tid & 63) == 0) { atomicAdd(&q.stats[0], slow); atomicAdd(&q.stats[1], total); }
    if (valid) {
        float4 o;
        o.x = ar / p.divisor; o.y = ag / p.divisor; o.z = ab / p.divisor; o.w = p.alpha;
        p.out[((size_t)face * p.size + y) * p.size + x] = o;
    }
}

static unsigned long long* g_bin_stats = nullptr;      // device counters, enabled by PBR_MC_STATS=1 (tuning aid)

// returns false when the binned kernel does not apply (caller uses the direct kernel)
static bool launch_binned(McArgs a, int nfaces, hipStream_t st) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("PBR_MC_BINNED"); mode = e ? atoi(e) : 0; }   // experimental: opt-in
    if (!mode || a.size < 256 || a.n_tab > 65535) return false;
    BinArgs q;
    q.a = a;
    float ratio = (float)a.size / (float)a.n_src;
    int M = (int)ceilf(0.5f * (16.0f / ratio) * 2.4f) + 1;
    static int m_override = -2;
    if (m_override == -2) { const char* e = getenv("PBR_MC_MARGIN"); m_override = e ? atoi(e) : -1; }
    if (m_override >= 0) M = m_override;
    if (M < 2) M = 2;
    if (M > 18) M = 18;
    q.M = M; q.WS = BIN_WC + 2 * M + 1;
    q.G = a.n_src / BIN_WC + 1; q.NW = 6 * q.G * q.G;
    if (q.NW > 4096) return false;
    q.tile_y0 = (a.y0 / 16) * 16;
    q.a.tiles_x = (a.size + 15) / 16;
    int tiles_y = (a.y0 + a.rows - q.tile_y0 + 15) / 16;
    q.a.tiles_per_face = q.a.tiles_x * tiles_y;
    static int stats_on = -1;
    if (stats_on < 0) {
        const char* e = getenv("PBR_MC_STATS"); stats_on = e ? atoi(e) : 0;
        if (stats_on) { if (hipMalloc(&g_bin_stats, 16) != hipSuccess) g_bin_stats = nullptr; else (void)hipMemset(g_bin_stats, 0, 16); }
    }
    q.stats = g_bin_stats;
    int NP = 256; while (NP < a.n_tab) NP <<= 1;
    if (NP > 8192) return false;
    q.NP = NP;
    int region = 4 * NP;                                   // keys during binning
    int main_bytes = 2 * NP + q.WS * q.WS * 16;            // list + window buffer afterwards
    if (main_bytes > region) region = main_bytes;
    q.region_bytes = (region + 15) & ~15;
    size_t lds = (size_t)q.region_bytes + (size_t)q.NW * 8;
    if (lds > 64 * 1024) return false;
    hipLaunchKernelGGL(k_mc_binned, dim3((unsigned)(q.a.tiles_per_face * nfaces)), dim3(256), lds, st, q);
    return true;
}

extern "C" int pbrk_mc_stats(unsigned long long* out2) {     // tuning aid: {fallback wave-samples, all wave-samples}
    if (!g_bin_stats || !out2) return PBRK_E_ARG;
    if (hipMemcpy(out2, g_bin_stats, 16, hipMemcpyDeviceToHost) != hipSuccess) return PBRK_E_LAUNCH;
    return PBRK_OK;
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
    if (a.cells) hipLaunchKernelGGL((k_mc_filter<S, 4, true>), grid, dim3(256), 0, st, a);
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
    float4* o = cells + ((size_t)(f * nc + j) * nc + i) * 3;
    // coefficient form {t00, t10 - t00, t01, t11 - t01}: the differences are the (rounded) ones lerp_fma would form per fetch
    o[0] = make_float4(t00.x, t00.y, t00.z, t10.x - t00.x);
    o[1] = make_float4(t10.y - t00.y, t10.z - t00.z, t01.x, t01.y);
    o[2] = make_float4(t01.z, t11.x - t01.x, t11.y - t01.y, t11.z - t01.z);
}

extern "C" size_t pbrk_cells_bytes(int n) { return (size_t)6 * (n + 1) * (n + 1) * 48; }

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
    // Kernel choice and the sample-split factor S depend on the LEVEL size only
    if (launch_binned(a, nfaces, st)) return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    if (launch_mc_region(a, nfaces, st)) return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    if (launch_mc_lds(a, nfaces, st)) return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
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
