// k_sweep.hip -- K7: voxel light-grid sweep, shaders/lightgrid_sweep.glsl:9-75.
//
// The reference runs one invocation per 128-voxel line of a 128^3 RGBA16F image (render.cpp:1064-1072,
// GPU_OpDispatch(1,16,16) of 1x8x8 groups), each walking its line left->right and right->left with a
// "moving light" that is halved into every empty voxel and reset by every occupied one, and stores
// mix(old, new, 0.35) into the empty voxels.  The line axis cycles x, y, z with the frame.
//
// The recurrence is sequential along the line (m' = 0.5*(v + m) in fp32, order matters for bit parity) but the
// three colour channels never mix, so the unit of work here is one (line, channel):
//   * a block owns a tile of 64 lines that are adjacent in memory and stages them through LDS, so that HBM
//     sees whole 512-B rows in every sweep direction (for the x direction the lines themselves are contiguous:
//     the load transposes them; for y and z adjacent lines are adjacent voxels);
//   * LDS holds the raw RGBA16F voxels as tile[x][65] (one voxel of padding per row): both the transposing
//     writes (lanes along x) and the sweep reads (lanes along lines) are then free of bank conflicts and every
//     sweep access is base + immediate (65 KB of dynamic LDS per block);
//   * waves 0..2 each sweep one colour channel of the 64 lines (lane = line); the forward pass keeps its 128
//     results in VGPRs (fully unrolled), the backward pass finishes each voxel and writes the mixed fp16
//     value into the tile; the alpha channel is left as loaded (mix(a, a, .35) rounds back to a in fp16);
//   * the kernel is streamed (see k_lightgrid_sweep): loads are released chunk by chunk to the forward sweep, and wave 3 -- which
//     has no channel -- streams chunks the backward sweep has finished back to HBM; occupied voxels keep their loaded bits,
//     so whole-pair stores rewrite identical data where the shader skips the store (old alpha >= 0.5).
// HBM traffic: 8 B read + 8 B written per voxel; 128^3 -> 32 MiB.
#include "pbr_device.h"
#include "pbr_kernels.h"

#include <hip/hip_fp16.h>

namespace {
constexpr int kLen = PBRK_SWEEP_LEN;     // voxels per line (the shader's array size)
constexpr int kTile = 64;                // lines per block
constexpr int kPitch = kTile + 1;        // LDS row pitch in voxels
constexpr int kLdsBytes = kLen * kPitch * 8 + 16;   // tile + the backward sweep's progress counter

struct SweepGeom {
    long long base;                      // voxel offset of line (f0, s0), step 0
    long long fstride, sstride, xstride; // voxel strides: next line in the tile, next tile row, next step along the line
    int nf;                              // lines in the fast dimension (tiles are cut from it)
    int contiguous_lines;                // 1: xstride == 1 (sweep along x), 0: fstride == 1
    int pair_stores;                     // 1: every (even voxel, next voxel) pair of a tile row is 16-B aligned in memory
};

#ifdef PBRK_SWEEP_PROFILE      // tools/ubench_sweep.hip only: per-block phase timestamps (wave 0) + placement
__device__ unsigned long long* g_sweep_prof;
#define SWEEP_STAMP(k) do { if (threadIdx.x == 0 && g_sweep_prof) { \
    g_sweep_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); \
    g_sweep_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8 + (k)] = clock64(); } } while (0)
#else
#define SWEEP_STAMP(k) do {} while (0)
#endif

// streaming (non-temporal) store: the swept voxels are not read again by this kernel
__device__ __forceinline__ void store_stream(uint2* p, uint2 v) {
    __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x, (unsigned long long*)p);
}

__device__ __forceinline__ void store_stream2(uint2* p, uint2 v0, uint2 v1) {      // two adjacent voxels, 16-B aligned
    typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
    u32x4n q = {v0.x, v0.y, v1.x, v1.y};
    __builtin_nontemporal_store(q, (u32x4n*)p);
}

__device__ __forceinline__ float half_bits_to_float(unsigned h) { return __half2float(__ushort_as_half((unsigned short)h)); }

// Streamed form: the line is cut into 8 chunks of 16 steps.  All loads are issued up front, in chunk order; chunk c is written to
// LDS and released by a barrier as soon as it has landed, so the forward sweep of chunk c runs while chunks c+1.. are still in
// flight.  The backward sweep releases each finished chunk to wave 3 (which has no channel to sweep), and wave 3 streams it
// back to HBM while waves 0-2 continue down the line.  The memory phases (2.4 us + 4 us when run back to back) hide under the
// 7.5 us dependent chain.
template <bool kContig>      // kContig: lines run along x (each line is contiguous); else adjacent lines are adjacent voxels
__global__ __launch_bounds__(256) void k_lightgrid_sweep(uint2* __restrict__ img, SweepGeom g) {
    extern __shared__ uint2 tile[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int f0 = blockIdx.x * kTile;
    const int nvalid = min(kTile, g.nf - f0);
    uint2* base = img + g.base + (long long)blockIdx.y * g.sstride + (long long)f0 * g.fstride;
    SWEEP_STAMP(0);
    constexpr int kChunk = 16, kChunks = kLen / kChunk;

    // ---- loads, chunk-major: 4 voxels per thread and chunk.
    //      lines along x (contiguous): thread -> (line t>>2, 4 consecutive voxels): 2 x 16 B, 128 B per line and chunk;
    //      lines along y / z: thread -> (line = lane, steps wave + 4j): 4 x 8 B, 512 B rows across the lanes.
    //      Loads are unconditional (lines beyond a partial tile re-read its last line; only stores are predicated), so that
    //      nothing but the data dependence orders them: the compiler's vmcnt waits then release chunk after chunk.
    uint2 v[kChunks][4];
    const int ll = min(kContig ? (t >> 2) : lane, nvalid - 1);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (kContig) {
            const uint4* p = (const uint4*)(base + (long long)ll * g.fstride + (c * kChunk + (t & 3) * 4));
            uint4 a = p[0], b = p[1];
            v[c][0] = make_uint2(a.x, a.y); v[c][1] = make_uint2(a.z, a.w); v[c][2] = make_uint2(b.x, b.y); v[c][3] = make_uint2(b.z, b.w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = base[(long long)ll + (long long)(c * kChunk + wave + 4 * j) * g.xstride];
        }
    }

    if (t == 0) *(int*)(tile + kLen * kPitch) = 0;                            // backward-sweep progress counter (visible after the first barrier)
    const bool sweeper = wave < 3 && lane < nvalid;
    const unsigned short* th = (const unsigned short*)tile + lane * 4;
    unsigned short* tw = (unsigned short*)tile + lane * 4;
    const int ch = wave;                                                      // colour channel of this wave (waves 0-2)
    const float sky = ch == 0 ? 1.0f : (ch == 1 ? 1.2f : 2.0f);               // :24 SKYLIGHT
    const float move_ratio = 0.5f;                                            // :33
    // The shader's step for an empty voxel is  t = v + m;  m' = 0.5*t;  v' = t - m'.  Halving is exact in binary fp
    // (no result here is near the fp32 subnormal range unless it is ~1e-30 below anything an fp16 store can see),
    // so m' = fl(0.5*v + 0.5*m) = fmaf(0.5, m, 0.5*v) and v' = t - 0.5*t = m' bit for bit: the dependent chain is one
    // FMA and one select per step, everything else is off the chain.  wv[x] holds 0.5 * (forward result of voxel x).
    float wv[kLen];
    float m = sky;                                                            // :36

    // ---- forward sweep, chunk by chunk as the data lands (:37-48)
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
        if (kContig) {
            const int l = t >> 2, x = c * kChunk + (t & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[(x + j) * kPitch + l] = v[c][j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[(c * kChunk + wave + 4 * j) * kPitch + lane] = v[c][j];
        }
        __syncthreads();
        if (sweeper) {
            unsigned cv[kChunk], ca[kChunk];
#pragma unroll
            for (int j = 0; j < kChunk; ++j) { const int x = c * kChunk + j; cv[j] = th[x * kPitch * 4 + ch]; ca[j] = th[x * kPitch * 4 + 3]; }
#pragma unroll
            for (int j = 0; j < kChunk; ++j) {
                float ov = half_bits_to_float(cv[j]), a = half_bits_to_float(ca[j]);
                float h = fmaf(move_ratio, m, move_ratio * ov);               // = 0.5 * (ov + m)
                m = a > 0.5f ? ov : h;                                        // moving light, and the voxel's forward value
                float w = move_ratio * m;
                asm("" : "+v"(w));                                            // keep one value per step live, nothing else
                wv[c * kChunk + j] = w;
            }
        }
    }
    SWEEP_STAMP(3);
    if (sweeper) wv[kLen - 1] = m;                                            // :49 values[127] += moving_light: (m + m) * 0.5
    m = sky;                                                                  // :52
    const float keep = 1.0f - 0.35f;                                          // mix(x, y, a) = x*(1-a) + y*a

    // ---- backward sweep (:53-66) fused with the mix (:70-75).  Waves 0-2 never wait: after each finished chunk they bump a
    //      counter in LDS; wave 3 (no channel to sweep) polls it and streams every chunk all three have finished back to HBM
    //      (:72 only voxels whose old alpha < 0.5; alpha itself is unchanged).  The counter reaches 3 * kChunks unconditionally,
    //      so wave 3 always leaves its loop.
    int* progress = (int*)(tile + kLen * kPitch);
    if (wave < 3) {
        unsigned cv[kChunk], ca[kChunk], nv[kChunk], na[kChunk];
        if (sweeper) {
#pragma unroll
            for (int j = 0; j < kChunk; ++j) { const int x = kLen - 1 - j; cv[j] = th[x * kPitch * 4 + ch]; ca[j] = th[x * kPitch * 4 + 3]; }
        }
#pragma unroll
        for (int c = kChunks - 1; c >= 0; --c) {
            if (sweeper) {
                if (c > 0) {
#pragma unroll
                    for (int j = 0; j < kChunk; ++j) { const int x = c * kChunk - 1 - j; nv[j] = th[x * kPitch * 4 + ch]; na[j] = th[x * kPitch * 4 + 3]; }
                }
#pragma unroll
                for (int j = 0; j < kChunk; ++j) {
                    const int x = c * kChunk + kChunk - 1 - j;
                    unsigned ab = ca[j];
                    asm("" : "+v"(ab));                                       // opaque: re-test alpha rather than carry 128 lane masks
                    float ov = half_bits_to_float(cv[j]), a = half_bits_to_float(ab);
                    float h = fmaf(move_ratio, m, wv[x]);                     // = 0.5 * (forward value + m) = the voxel's new value
                    m = a > 0.5f ? ov : h;
                    float vv = h;
                    if (x == 0) vv = vv + m;                                  // :67 (m is final here)
                    float mixed = ov * keep + vv * 0.35f;
                    unsigned short nb = __half_as_ushort(__float2half_rn(mixed));
                    tw[x * kPitch * 4 + ch] = a < 0.5f ? nb : (unsigned short)cv[j];   // :72 (other voxels keep their bits)
                }
#pragma unroll
                for (int j = 0; j < kChunk; ++j) { cv[j] = nv[j]; ca[j] = na[j]; }
            }
            // release: this chunk's tile writes are ordered before the counter bump for the compiler as well as the hardware
            // (LDS operations of one wave complete in order; the release keeps the stores from being sunk below the atomic)
            if (lane == 0) __hip_atomic_fetch_add(progress, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
#pragma unroll 1
        for (int done = kChunks - 1; done >= 0; --done) {
            const int need = 3 * (kChunks - done);
            // acquire: the tile loads below may not be hoisted above the poll that licenses them
            while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(2);
            if (g.pair_stores) {
                // 16 B per lane.  Occupied voxels still hold their loaded bits in the tile, so storing a pair whole only rewrites
                // identical data where the shader would have skipped the store (:72); this keeps wave 3 (the only storing
                // wave) at 8 store instructions per chunk and ahead of the sweep.
                if (kContig) {                                                // lane -> (8 lines x 8 voxel pairs)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int l = k * 8 + (lane >> 3), x = done * kChunk + 2 * (lane & 7);
                        uint2 o0 = tile[x * kPitch + l], o1 = tile[(x + 1) * kPitch + l];
                        if (l < nvalid) store_stream2(base + (long long)l * g.fstride + x, o0, o1);
                    }
                } else {                                                      // lane -> (line pair, step parity)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int l = 2 * (lane & 31), x = done * kChunk + 2 * k + (lane >> 5);
                        uint2 o0 = tile[x * kPitch + l], o1 = tile[x * kPitch + l + 1];
                        uint2* dst = base + (long long)l + (long long)x * g.xstride;
                        if (l + 1 < nvalid) store_stream2(dst, o0, o1);
                        else if (l < nvalid) store_stream(dst, o0);
                    }
                }
            } else if (kContig) {                                             // lane -> (4 lines x 16 voxels): 128 B per line
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int l = k * 4 + (lane >> 4), x = done * kChunk + (lane & 15);
                    uint2 o = tile[x * kPitch + l];
                    if (l < nvalid && half_bits_to_float(o.y >> 16) < 0.5f) store_stream(base + (long long)l * g.fstride + x, o);
                }
            } else {                                                          // lane = line: 512-B rows
#pragma unroll
                for (int k = 0; k < kChunk; ++k) {
                    const int x = done * kChunk + k;
                    uint2 o = tile[x * kPitch + lane];
                    if (lane < nvalid && half_bits_to_float(o.y >> 16) < 0.5f) store_stream(base + (long long)lane + (long long)x * g.xstride, o);
                }
            }
        }
    }
    SWEEP_STAMP(6);
#ifdef PBRK_SWEEP_PROFILE
    if (threadIdx.x == 0 && g_sweep_prof) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sweep_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}
}  // namespace

extern "C" int pbrk_lightgrid_sweep(void* image_rgba16f, int w, int h, int d, int direction, int y0, int y1, int z0, int z1, void* stream) {
    if (!image_rgba16f || w < 1 || h < 1 || d < 1 || direction < 0 || direction > 2) return PBRK_E_ARG;
    if (y0 < 0 || z0 < 0 || y0 >= y1 || z0 >= z1) return PBRK_E_ARG;
    // invocation (iy, iz) touches: 0 -> (x, iy, iz); 1 -> (iz, x, iy); 2 -> (iy, iz, x)   (lightgrid_sweep.glsl:10-22)
    long long W = w, H = h;
    SweepGeom g;
    int ns;
    if (direction == 0) {
        if (w < kLen || y1 > h || z1 > d) return PBRK_E_ARG;
        g.fstride = W; g.sstride = W * H; g.xstride = 1; g.contiguous_lines = 1;
        g.base = (long long)z0 * W * H + (long long)y0 * W; g.nf = y1 - y0; ns = z1 - z0;
    } else if (direction == 1) {
        if (h < kLen || z1 > w || y1 > d) return PBRK_E_ARG;
        g.fstride = 1; g.sstride = W * H; g.xstride = W; g.contiguous_lines = 0;
        g.base = (long long)y0 * W * H + z0; g.nf = z1 - z0; ns = y1 - y0;
    } else {
        if (d < kLen || y1 > w || z1 > h) return PBRK_E_ARG;
        g.fstride = 1; g.sstride = W; g.xstride = W * H; g.contiguous_lines = 0;
        g.base = (long long)z0 * W + y0; g.nf = y1 - y0; ns = z1 - z0;
    }
    if (ns > 65535) return PBRK_E_ARG;
    // 16-B stores need every even voxel of a tile row on a 16-B boundary: image base (hipMalloc: yes), offsets and strides even
    g.pair_stores = ((uintptr_t)image_rgba16f % 16 == 0) && (g.base % 2 == 0) && (g.sstride % 2 == 0) &&
                    (g.contiguous_lines ? (g.fstride % 2 == 0) : (g.xstride % 2 == 0));
    static int attr_device = -1;                                                // the attribute is per device
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return PBRK_E_LAUNCH;
    if (attr_device != device) {
        if (hipFuncSetAttribute((const void*)k_lightgrid_sweep<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) return PBRK_E_LAUNCH;
        if (hipFuncSetAttribute((const void*)k_lightgrid_sweep<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) return PBRK_E_LAUNCH;
        attr_device = device;
    }
    if (g.contiguous_lines) hipLaunchKernelGGL(k_lightgrid_sweep<true>, dim3((g.nf + kTile - 1) / kTile, ns), dim3(256), kLdsBytes, (hipStream_t)stream, (uint2*)image_rgba16f, g);
    else hipLaunchKernelGGL(k_lightgrid_sweep<false>, dim3((g.nf + kTile - 1) / kTile, ns), dim3(256), kLdsBytes, (hipStream_t)stream,
                       (uint2*)image_rgba16f, g);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
