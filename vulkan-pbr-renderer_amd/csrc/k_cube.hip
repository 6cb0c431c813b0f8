// k_cube.hip -- bandwidth-shaped cube kernels for gfx950:
//   K2   mip chain (2x2 box per face)           reference: src/gpu/gpu_vulkan.c:1458-1483, :2786-2826
//   apron build (seamless-cube bordered layout)  reference: sampler state src/gpu/gpu_vulkan.c:613-634
//   K4a  prefilter mip 0 = bilinear copy         reference: shaders/gen_prefiltered_env_map.glsl:112-114
#include "pbr_device.h"
#include "pbr_kernels.h"

// ------------------------------------------------------------------------------------------
// K2: one thread per destination texel; each reads two 32-byte row segments and writes 16 bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mip_level(const float4* __restrict__ src, float4* __restrict__ dst,
                                                   int ns, int nd, int nfaces) {
    // grid: x = column blocks of 64 texels, y = row blocks of 4 rows, z = layer (no integer division per texel)
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (x >= nd || y >= nd || f >= nfaces) return;
    const float4* p = src + ((size_t)f * ns + 2 * y) * ns + 2 * x;
    float4 a = p[0], b = p[1], c = p[ns], d = p[ns + 1];
    float4 o;
    o.x = (((a.x + b.x) + c.x) + d.x) * 0.25f;
    o.y = (((a.y + b.y) + c.y) + d.y) * 0.25f;
    o.z = (((a.z + b.z) + c.z) + d.z) * 0.25f;
    o.w = (((a.w + b.w) + c.w) + d.w) * 0.25f;
    dst[((size_t)f * nd + y) * nd + x] = o;
}

// Two levels per launch: a thread forms a 2x2 block of level l from a 4x4 block of level l-1 (four 64-byte row segments) and,
// from those four results, the texel of level l+1 they cover -- exactly the values and the operation order the next launch
// would have read back from HBM, so the chain stays bit-identical while level l is never re-read (671 -> 562 MB at W = 2048,
// half the launches).  Lanes of a wave cover 64 adjacent 2x2 blocks of a row pair: 128 contiguous bytes per lane pair on the
// read side, 32-byte stores for level l and 16-byte stores for level l+1.
__device__ __forceinline__ float4 box4(float4 a, float4 b, float4 c, float4 d) {
    float4 o;
    o.x = (((a.x + b.x) + c.x) + d.x) * 0.25f;
    o.y = (((a.y + b.y) + c.y) + d.y) * 0.25f;
    o.z = (((a.z + b.z) + c.z) + d.z) * 0.25f;
    o.w = (((a.w + b.w) + c.w) + d.w) * 0.25f;
    return o;
}
__global__ __launch_bounds__(256) void k_mip_level2(const float4* __restrict__ src, float4* __restrict__ dst1, float4* __restrict__ dst2,
                                                    int ns, int n1, int n2) {
    int x2 = blockIdx.x * 64 + (threadIdx.x & 63);           // texel of level l+1
    int y2 = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (x2 >= n2 || y2 >= n2) return;
    const float4* p = src + ((size_t)f * ns + 4 * y2) * ns + 4 * x2;
    float4 r[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4* q0 = p + (size_t)(2 * j) * ns;
        const float4* q1 = q0 + ns;
        float4 a0 = q0[0], b0 = q0[1], a1 = q0[2], b1 = q0[3];
        float4 c0 = q1[0], d0 = q1[1], c1 = q1[2], d1 = q1[3];
        r[2 * j] = box4(a0, b0, c0, d0);
        r[2 * j + 1] = box4(a1, b1, c1, d1);
    }
    float4* o1 = dst1 + ((size_t)f * n1 + 2 * y2) * n1 + 2 * x2;
    o1[0] = r[0]; o1[1] = r[1]; o1[n1] = r[2]; o1[n1 + 1] = r[3];
    dst2[((size_t)f * n2 + y2) * n2 + x2] = box4(r[0], r[1], r[2], r[3]);
}

// K2 for levels that are not an exact 2:1 of their source (faces that are not a power of two: 125 -> 62, 3 -> 1), and
// GPU_OpBlit between whole RGBA32F subresources of any two sizes: a genuine linear resample, the rule of oracle/pbr_oracle.c A2
// (vkCmdBlitImage, unnormalised linear filtering, clamp to edge; fp32, every operation rounded -- the file is compiled with
// -ffp-contract=off).  The scales ns / nd are formed on the host by a correctly rounded division.
__global__ __launch_bounds__(256) void k_blit_linear(const float4* __restrict__ src, float4* __restrict__ dst,
                                                     int ns_w, int ns_h, int nd_w, int nd_h, float sx, float sy) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (x >= nd_w || y >= nd_h) return;
    const float tu = ((float)x + 0.5f) * sx - 0.5f, tv = ((float)y + 0.5f) * sy - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, 0), ns_w - 1), i1 = min(max((int)fu + 1, 0), ns_w - 1);
    const int j0 = min(max((int)fv, 0), ns_h - 1), j1 = min(max((int)fv + 1, 0), ns_h - 1);
    const float4* p = src + (size_t)f * ns_h * ns_w;
    const float4 t00 = p[(size_t)j0 * ns_w + i0], t10 = p[(size_t)j0 * ns_w + i1], t01 = p[(size_t)j1 * ns_w + i0], t11 = p[(size_t)j1 * ns_w + i1];
    const float ia = 1.0f - a, ib = 1.0f - b;
    float4 o;
    o.x = (t00.x * ia + t10.x * a) * ib + (t01.x * ia + t11.x * a) * b;
    o.y = (t00.y * ia + t10.y * a) * ib + (t01.y * ia + t11.y * a) * b;
    o.z = (t00.z * ia + t10.z * a) * ib + (t01.z * ia + t11.z * a) * b;
    o.w = (t00.w * ia + t10.w * a) * ib + (t01.w * ia + t11.w * a) * b;
    dst[((size_t)f * nd_h + y) * nd_w + x] = o;
}

// ------------------------------------------------------------------------------------------
// Apron build.  Edge adjacency of the Vulkan cube faces (table in gen_prefiltered_env_map.glsl:12-23),
// per face and edge {left i=-1, right i=n, top j=-1, bottom j=n}: neighbour face and how its
// (i', j') follow from the along-edge index k (0: k, 1: n-1-k, 2: 0, 3: n-1).
// ------------------------------------------------------------------------------------------
__constant__ unsigned char kEdge[6][4][3] = {
    /* +X */ {{4, 3, 0}, {5, 2, 0}, {2, 3, 1}, {3, 3, 0}},
    /* -X */ {{5, 3, 0}, {4, 2, 0}, {2, 2, 0}, {3, 2, 1}},
    /* +Y */ {{1, 0, 2}, {0, 1, 2}, {5, 1, 2}, {4, 0, 2}},
    /* -Y */ {{1, 1, 3}, {0, 0, 3}, {4, 0, 3}, {5, 1, 3}},
    /* +Z */ {{1, 3, 0}, {0, 2, 0}, {2, 0, 3}, {3, 0, 2}},
    /* -Z */ {{0, 3, 0}, {1, 2, 0}, {2, 1, 2}, {3, 1, 3}},
};

__device__ __forceinline__ int edge_code(int code, int k, int n) {
    return code == 0 ? k : (code == 1 ? n - 1 - k : (code == 2 ? 0 : n - 1));
}

// texel (i,j) of face f where exactly one of i,j is out of range
__device__ __forceinline__ float4 edge_texel(const float4* __restrict__ lvl, int n, int f, int i, int j) {
    int e, k;
    if (i < 0) { e = 0; k = j; } else if (i >= n) { e = 1; k = j; } else if (j < 0) { e = 2; k = i; } else { e = 3; k = i; }
    int nf = kEdge[f][e][0];
    int ni = edge_code(kEdge[f][e][1], k, n);
    int nj = edge_code(kEdge[f][e][2], k, n);
    return lvl[((size_t)nf * n + nj) * n + ni];
}

__global__ __launch_bounds__(256) void k_border_level(const float4* __restrict__ lvl, float4* __restrict__ out, int n) {
    int nb = n + 2;
    int bi = blockIdx.x * 64 + (threadIdx.x & 63);
    int bj = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = blockIdx.z;
    if (bi >= nb || bj >= nb) return;
    int i = bi - 1, j = bj - 1;
    bool oi = (i < 0) | (i >= n), oj = (j < 0) | (j >= n);
    float4 v;
    if (!oi && !oj) {
        v = lvl[((size_t)f * n + j) * n + i];
    } else if (oi != oj) {
        v = edge_texel(lvl, n, f, i, j);
    } else {
        // cube corner: the missing texel is the mean of the three that exist
        int ci = i < 0 ? 0 : n - 1, cj = j < 0 ? 0 : n - 1;
        float4 a = lvl[((size_t)f * n + cj) * n + ci];
        float4 b = edge_texel(lvl, n, f, i, cj);
        float4 c = edge_texel(lvl, n, f, ci, j);
        v.x = ((a.x + b.x) + c.x) / 3.0f;
        v.y = ((a.y + b.y) + c.y) / 3.0f;
        v.z = ((a.z + b.z) + c.z) / 3.0f;
        v.w = ((a.w + b.w) + c.w) / 3.0f;
    }
    out[((size_t)f * nb + bj) * nb + bi] = v;
}

// ------------------------------------------------------------------------------------------
// K4a: out(face, y, x) = bilinear(src level, R(face, x, y)); 16 B coalesced store per lane.
// ------------------------------------------------------------------------------------------
// The lookup is bit-identical to a scalar CPU evaluation (a tap weight next to a 5e4:1 sun texel leaves no room): the
// direction, its cube projection and the weights keep the shader's operation order with correctly rounded results, through the
// Newton-corrected v_rsq / v_rcp sequences of pbr_device.h; taps come through range-checked buffer loads (32-bit offsets).
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(__amdgpu_buffer_rsrc_t r, int off) {
    u32x4k v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__global__ __launch_bounds__(256) void k_prefilter_copy(const float4* __restrict__ src, int n_src, unsigned src_bytes,
                                                        float4* __restrict__ out, int size,
                                                        int face0, int y0, int rows) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int yr = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = face0 + blockIdx.z;
    if (x >= size || yr >= rows) return;
    int y = y0 + yr;
    // CubemapSampleDirFromFaceUV (gen_prefiltered_env_map.glsl:11-66), power-of-two sizes: x / n == x * (1/n) exactly
    float rn = 1.0f / (float)size;
    float sc0 = 2 * (((float)x + 0.5f) * rn - 0.5f), tc0 = 2 * (((float)y + 0.5f) * rn - 0.5f);
    f3 r;
    switch (f) {
    case 0: r = mk3(1.0f, -tc0, -sc0); break;
    case 1: r = mk3(-1.0f, -tc0, sc0); break;
    case 2: r = mk3(sc0, 1.0f, tc0); break;
    case 3: r = mk3(sc0, -1.0f, -tc0); break;
    case 4: r = mk3(sc0, -tc0, 1.0f); break;
    default: r = mk3(-sc0, -tc0, -1.0f); break;
    }
    const f3 R = normalize3_nr(r);
    // textureLod(env, R, 1.0): face selection in hardware, s = (0.5 sc) / |ma| + 0.5, u = s n - 0.5
    const float fid = __builtin_amdgcn_cubeid(R.x, R.y, R.z);
    const float sc = __builtin_amdgcn_cubesc(R.x, R.y, R.z), tc = __builtin_amdgcn_cubetc(R.x, R.y, R.z);
    SharedRcp rma; rma.d = 0.5f * fabsf(__builtin_amdgcn_cubema(R.x, R.y, R.z)); rma.r = rcp_nr(rma.d);
    const float s = div_by(0.5f * sc, rma) + 0.5f, t = div_by(0.5f * tc, rma) + 0.5f;
    const float nf = (float)n_src;
    const float u = s * nf - 0.5f, v = t * nf - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const float nbf = nf + 2.0f;
    // bordered texel index (face * nb + j0) * nb + i0 with i0 = floor(u) + 1 in [0, n]: exact in fp32 while 6 (n + 2)^2 < 2^24 (n <= 1600 here)
    const int off = (int)(fmaf(fmaf(fid, nbf, fv + 1.0f), nbf, fu + 1.0f) * 16.0f);
    const int row = (n_src + 2) * 16;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    const float4 t00 = ld4(rs, off), t10 = ld4(rs, off + 16), t01 = ld4(rs, off + row), t11 = ld4(rs, off + row + 16);
    float4 o;
    o.x = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
    o.y = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
    o.z = lerp_fma(lerp_fma(t00.z, t10.z, a), lerp_fma(t01.z, t11.z, a), b);
    o.w = lerp_fma(lerp_fma(t00.w, t10.w, a), lerp_fma(t01.w, t11.w, a), b);
    typedef float v4n __attribute__((ext_vector_type(4)));
    v4n ov = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(ov, (v4n*)&out[((size_t)f * size + y) * size + x]);        // written once, read by a later pass: keep it out of L2
}
// general sizes (not a power of two, or a level beyond the fp32 index range): the plain exact path
__global__ __launch_bounds__(256) void k_prefilter_copy_general(const float4* __restrict__ src, int n_src,
                                                                float4* __restrict__ out, int size,
                                                                int face0, int y0, int rows, int snap) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int yr = blockIdx.y * 4 + (threadIdx.x >> 6);
    int f = face0 + blockIdx.z;
    if (x >= size || yr >= rows) return;
    int y = y0 + yr;
    f3 R = face_texel_dir(f, x, y, size);
    float4 v = cube_fetch_rgba<true>(src, n_src, R, snap != 0);
    out[((size_t)f * size + y) * size + x] = v;
}

// Smallest and largest RGB value of one level (non-negative inputs; anything negative or NaN reports a minimum of 0): the dynamic
// range the tolerance-budgeted sample cut of K4b needs (gpu_hip.cpp, GPUX_SetPrefilterTolerance).  out2 = {min bits, max bits},
// initialised by the caller to {+inf, 0}; non-negative floats order like their bit patterns.
__global__ __launch_bounds__(256) void k_level_minmax(const float4* __restrict__ lvl, size_t n, unsigned* __restrict__ out2) {
    float mn = __uint_as_float(0x7F800000u), mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = lvl[i];
        mn = fminf(mn, fmaxf(fminf(fminf(v.x, v.y), v.z), 0.0f));
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), v.z));
        if (!(v.x >= 0.0f) || !(v.y >= 0.0f) || !(v.z >= 0.0f)) mn = 0.0f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out2[0], __float_as_uint(mn)); atomicMax(&out2[1], __float_as_uint(fmaxf(mx, 0.0f))); }
}
extern "C" int pbrk_level_minmax(const void* level, size_t texels, void* out2_device, void* stream) {
    if (!level || !texels || !out2_device) return PBRK_E_ARG;
    size_t blocks = (texels + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_level_minmax, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)level, texels, (unsigned*)out2_device);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// Cube-sampler convention switch: 0 = exact fp32 tap weights (default, every fast kernel), 1 = coordinates and
// LOD fraction snapped to 1/256 texel.  With the switch on, K3 / K4 / K5 run their general kernels (the fast ones implement the
// default convention only): a diagnostic to MEASURE how far the outputs move between two conventions the reference permits.
static int g_cube_snap = 0;
extern "C" void pbrk_set_cube_sampler_snap(int on) { g_cube_snap = on != 0; }
extern "C" int pbrk_get_cube_sampler_snap(void) { return g_cube_snap; }

// ------------------------------------------------------------------------------------------
static inline int grid_for(size_t total, int block, int cap) {
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
static inline int lvl_size(int W, int l) { int n = W >> l; return n < 1 ? 1 : n; }

extern "C" int pbrk_blit_linear(const void* src, int ns_w, int ns_h, void* dst, int nd_w, int nd_h, int nlayers, void* stream) {
    if (!src || !dst || ns_w < 1 || ns_h < 1 || nd_w < 1 || nd_h < 1 || nlayers < 1) return PBRK_E_ARG;
    hipLaunchKernelGGL(k_blit_linear, dim3((nd_w + 63) / 64, (nd_h + 3) / 4, nlayers), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, ns_w, ns_h, nd_w, nd_h, (float)ns_w / (float)nd_w, (float)ns_h / (float)nd_h);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// GPU_OpClearColor* (gpu.h: clear one level or all of them): `bytes` of device memory filled with a texel pattern of 1, 2, 4, 8 or 16
// bytes, one launch whatever the alignment and size.  (hipMemsetAsync splits an all-levels clear of the 1080p bloom target -- 22 MB
// that are not a multiple of 16 -- into a bulk and a 256-thread remainder kernel of 6.4 + 4.9 us, and has no 8- / 16-byte form.)
__global__ __launch_bounds__(256) void k_fill_pattern(unsigned char* __restrict__ p, size_t bytes, uint4 pat) {
    size_t head = (16 - ((size_t)p & 15)) & 15;                   // bytes up to the first 16-byte boundary: a multiple of the pattern size
    if (head > bytes) head = bytes;
    const size_t body = (bytes - head) >> 4, tail = bytes - head - (body << 4);
    uint4* v = (uint4*)(p + head);
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < body; i += stride) v[i] = pat;
    const unsigned char* pb = (const unsigned char*)&pat;
    if (i0 < head) p[i0] = pb[i0];                                 // the pattern starts at p: byte k holds pattern byte k mod size = byte k of its 16-byte repeat
    if (i0 < tail) p[head + (body << 4) + i0] = pb[i0];
}

extern "C" int pbrk_fill_pattern(void* dst, unsigned long long bytes, const void* pattern, int pattern_bytes, void* stream) {
    if (!dst || !pattern || (pattern_bytes != 1 && pattern_bytes != 2 && pattern_bytes != 4 && pattern_bytes != 8 && pattern_bytes != 16)) return PBRK_E_ARG;
    if (bytes % (unsigned long long)pattern_bytes || ((size_t)dst % (size_t)pattern_bytes)) return PBRK_E_ARG;
    if (!bytes) return PBRK_OK;
    unsigned char rep[16];
    for (int k = 0; k < 16; ++k) rep[k] = ((const unsigned char*)pattern)[k % pattern_bytes];
    uint4 pat;
    __builtin_memcpy(&pat, rep, 16);
    const unsigned long long vecs = bytes / 16 + 1;
    unsigned blocks = (unsigned)((vecs + 1023) / 1024);           // four 16-byte stores per thread on a big fill
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill_pattern, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst, (size_t)bytes, pat);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

// Level sizes are the reference's (gpu_vulkan.c:1344-1351, 1458-1483): n_l = max(1, W >> l), as many levels as 1 + floor(log2 W).
// A level that is exactly half its source takes the 2x2 box kernels (two levels per launch while that holds for both); an odd
// source (faces that are not a power of two) takes the linear resample.
extern "C" int pbrk_mip_chain(void* pyramid, int W, int levels, void* stream) {
    if (!pyramid || W <= 0 || levels < 1 || levels > pbrk_mip_count(W, W)) return PBRK_E_ARG;
    float4* base = (float4*)pyramid;
    auto even = [&](int l) { return lvl_size(W, l - 1) == 2 * lvl_size(W, l); };      // level l is an exact 2:1 of level l - 1
    int l = 1;
    while (l < levels) {
        int ns = lvl_size(W, l - 1), n1 = lvl_size(W, l);
        const float4* src = base + pbrk_level_offset(W, l - 1);
        if (l + 1 < levels && even(l) && even(l + 1) && lvl_size(W, l + 1) >= 2) {      // pairs of levels while the second one is at least 2x2
            int n2 = lvl_size(W, l + 1);
            hipLaunchKernelGGL(k_mip_level2, dim3((n2 + 63) / 64, (n2 + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream,
                               src, base + pbrk_level_offset(W, l), base + pbrk_level_offset(W, l + 1), ns, n1, n2);
            l += 2;
        } else if (even(l)) {
            hipLaunchKernelGGL(k_mip_level, dim3((n1 + 63) / 64, (n1 + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream, src, base + pbrk_level_offset(W, l), ns, n1, 6);
            l += 1;
        } else {
            hipLaunchKernelGGL(k_blit_linear, dim3((n1 + 63) / 64, (n1 + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream,
                               src, base + pbrk_level_offset(W, l), ns, ns, n1, n1, (float)ns / (float)n1, (float)ns / (float)n1);
            l += 1;
        }
    }
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_box_downsample(const void* src, int ns, void* dst, int nlayers, void* stream) {
    if (!src || !dst || ns < 2 || (ns & 1) || nlayers < 1) return PBRK_E_ARG;
    int nd = ns / 2;
    hipLaunchKernelGGL(k_mip_level, dim3((nd + 63) / 64, (nd + 3) / 4, nlayers), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, ns, nd, nlayers);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_border_build_range(const void* pyramid, void* bordered, int W, int levels, int level0, int level1, void* stream) {
    if (!pyramid || !bordered || W <= 0 || levels < 1 || levels > pbrk_mip_count(W, W)) return PBRK_E_ARG;
    if (level0 < 0 || level1 > levels || level0 > level1) return PBRK_E_ARG;
    for (int l = level0; l < level1; ++l) {
        int n = lvl_size(W, l);
        const float4* src = (const float4*)pyramid + pbrk_level_offset(W, l);
        float4* dst = (float4*)bordered + pbrk_bordered_level_offset(W, l);
        hipLaunchKernelGGL(k_border_level, dim3((n + 2 + 63) / 64, (n + 2 + 3) / 4, 6), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    }
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

extern "C" int pbrk_border_build(const void* pyramid, void* bordered, int W, int levels, void* stream) {
    return pbrk_border_build_range(pyramid, bordered, W, levels, 0, levels, stream);
}

extern "C" int pbrk_prefilter_copy(const void* src_bordered_level, int n_src, void* out, int out_size,
                                   int face0, int face1, int y0, int y1, void* stream) {
    if (!src_bordered_level || !out || n_src < 1 || out_size < 1) return PBRK_E_ARG;
    if (face0 < 0 || face1 > 6 || face0 >= face1 || y0 < 0 || y1 > out_size || y0 >= y1) return PBRK_E_ARG;
    dim3 grid((out_size + 63) / 64, (y1 - y0 + 3) / 4, face1 - face0);
    size_t src_bytes = (size_t)6 * (n_src + 2) * (n_src + 2) * 16;
    const int snap = pbrk_get_cube_sampler_snap();                 // diagnostic convention: general kernel only
    if (!snap && (out_size & (out_size - 1)) == 0 && n_src <= 1600)         // 6 (n + 2)^2 < 2^24: the texel index is exact in fp32
        hipLaunchKernelGGL(k_prefilter_copy, grid, dim3(256), 0, (hipStream_t)stream,
                           (const float4*)src_bordered_level, n_src, (unsigned)src_bytes, (float4*)out, out_size, face0, y0, y1 - y0);
    else
        hipLaunchKernelGGL(k_prefilter_copy_general, grid, dim3(256), 0, (hipStream_t)stream,
                           (const float4*)src_bordered_level, n_src, (float4*)out, out_size, face0, y0, y1 - y0, snap);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
