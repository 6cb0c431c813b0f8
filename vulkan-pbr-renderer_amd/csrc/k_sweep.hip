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
//   * the write-back stores exactly the voxels the shader stores (old alpha < 0.5).
// HBM traffic: 8 B read per voxel + 8 B written per empty voxel; 128^3 -> 16 MiB + <= 16 MiB.
#include "pbr_device.h"
#include "pbr_kernels.h"

#include <hip/hip_fp16.h>

namespace {
constexpr int kLen = PBRK_SWEEP_LEN;     // voxels per line (the shader's array size)
constexpr int kTile = 64;                // lines per block
constexpr int kPitch = kTile + 1;        // LDS row pitch in voxels
constexpr int kLdsBytes = kLen * kPitch * 8;

struct SweepGeom {
    long long base;                      // voxel offset of line (f0, s0), step 0
    long long fstride, sstride, xstride; // voxel strides: next line in the tile, next tile row, next step along the line
    int nf;                              // lines in the fast dimension (tiles are cut from it)
    int contiguous_lines;                // 1: xstride == 1 (sweep along x), 0: fstride == 1
};

#ifdef PBRK_SWEEP_PROFILE      // tools/ubench_sweep.hip only: per-block phase timestamps (wave 0) + placement
__device__ unsigned long long* g_sweep_prof;
#define SWEEP_STAMP(k) do { if (threadIdx.x == 0 && g_sweep_prof) { \
    g_sweep_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); \
    g_sweep_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8 + (k)] = clock64(); } } while (0)
#else
#define SWEEP_STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ float half_bits_to_float(unsigned h) { return __half2float(__ushort_as_half((unsigned short)h)); }

__global__ __launch_bounds__(256) void k_lightgrid_sweep(uint2* __restrict__ img, SweepGeom g) {
    extern __shared__ uint2 tile[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int f0 = blockIdx.x * kTile;
    const int nvalid = min(kTile, g.nf - f0);
    uint2* base = img + g.base + (long long)blockIdx.y * g.sstride + (long long)f0 * g.fstride;
    SWEEP_STAMP(0);

    // ---- stage the tile: 32 x 8-B loads per thread, 512 B contiguous per wave-level load in every direction
    {
        uint2 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {                                        // all 32 loads in flight: one memory round trip
            int item = wave + 4 * i, l, x;
            if (g.contiguous_lines) { l = item >> 1; x = lane + 64 * (item & 1); }
            else { l = lane; x = item; }
            v[i] = make_uint2(0u, 0u);
            if (l < nvalid) v[i] = base[(long long)l * g.fstride + (long long)x * g.xstride];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            int item = wave + 4 * i, l, x;
            if (g.contiguous_lines) { l = item >> 1; x = lane + 64 * (item & 1); }
            else { l = lane; x = item; }
            tile[x * kPitch + l] = v[i];
        }
    }
    SWEEP_STAMP(1);
    __syncthreads();
    SWEEP_STAMP(2);

    // ---- sweeps: wave c < 3 handles channel c of line `lane`.  LDS reads run one batch of kBatch steps ahead of the
    //      dependent chain (the chain is the critical path: a wave alone on its SIMD cannot hide LDS latency otherwise).
    if (wave < 3 && lane < nvalid) {
        const unsigned short* th = (const unsigned short*)tile + lane * 4;
        unsigned short* tw = (unsigned short*)tile + lane * 4;
        const int c = wave;
        const float sky = c == 0 ? 1.0f : (c == 1 ? 1.2f : 2.0f);            // :24 SKYLIGHT
        const float move_ratio = 0.5f;                                        // :33
        constexpr int kBatch = 16, kBatches = kLen / kBatch;
        // The shader's step for an empty voxel is  t = v + m;  m' = 0.5*t;  v' = t - m'.  Halving is exact in binary fp
        // (no result here is near the fp32 subnormal range unless it is ~1e-30 below anything an fp16 store can see),
        // so m' = fl(0.5*v + 0.5*m) = fmaf(0.5, m, 0.5*v) and v' = t - 0.5*t = m' bit for bit: the dependent chain is one
        // FMA and one select per step, everything else is off the chain.  wv[x] holds 0.5 * (forward result of voxel x).
        float wv[kLen];
        unsigned cv[kBatch], ca[kBatch], nv[kBatch], na[kBatch];
        float m = sky;                                                        // :36
#pragma unroll
        for (int j = 0; j < kBatch; ++j) { cv[j] = th[j * kPitch * 4 + c]; ca[j] = th[j * kPitch * 4 + 3]; }
#pragma unroll
        for (int b = 0; b < kBatches; ++b) {                                  // :37-48
            if (b + 1 < kBatches) {
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    int x = (b + 1) * kBatch + j;
                    nv[j] = th[x * kPitch * 4 + c]; na[j] = th[x * kPitch * 4 + 3];
                }
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                float ov = half_bits_to_float(cv[j]), a = half_bits_to_float(ca[j]);
                float h = fmaf(move_ratio, m, move_ratio * ov);               // = 0.5 * (ov + m)
                m = a > 0.5f ? ov : h;                                        // moving light, and the voxel's forward value
                float w = move_ratio * m;
                asm("" : "+v"(w));                                            // keep one value per step live, nothing else
                wv[b * kBatch + j] = w;
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) { cv[j] = nv[j]; ca[j] = na[j]; }
        }
        wv[kLen - 1] = m;                                                     // :49 values[127] += moving_light: (m + m) * 0.5
        SWEEP_STAMP(3);
        m = sky;                                                              // :52
        const float keep = 1.0f - 0.35f;                                      // mix(x, y, a) = x*(1-a) + y*a
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            int x = kLen - 1 - j;
            cv[j] = th[x * kPitch * 4 + c]; ca[j] = th[x * kPitch * 4 + 3];
        }
#pragma unroll
        for (int b = 0; b < kBatches; ++b) {                                  // :53-66, fused with the store loop :70-75
            if (b + 1 < kBatches) {
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    int x = kLen - 1 - ((b + 1) * kBatch + j);
                    nv[j] = th[x * kPitch * 4 + c]; na[j] = th[x * kPitch * 4 + 3];
                }
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int x = kLen - 1 - (b * kBatch + j);
                unsigned ab = ca[j];
                asm("" : "+v"(ab));                                           // opaque: re-test alpha rather than carry 128 lane masks
                float ov = half_bits_to_float(cv[j]), a = half_bits_to_float(ab);
                float h = fmaf(move_ratio, m, wv[x]);                         // = 0.5 * (forward value + m) = the voxel's new value
                m = a > 0.5f ? ov : h;
                float v = h;
                if (x == 0) v = v + m;                                        // :67 (m is final here)
                float mixed = ov * keep + v * 0.35f;
                unsigned short nb = __half_as_ushort(__float2half_rn(mixed));
                tw[x * kPitch * 4 + c] = a < 0.5f ? nb : (unsigned short)cv[j];   // :72 (other voxels keep their bits)
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) { cv[j] = nv[j]; ca[j] = na[j]; }
        }
    }
    SWEEP_STAMP(4);
    __syncthreads();
    SWEEP_STAMP(5);

    // ---- write back the voxels the shader writes (:72 old alpha < 0.5; alpha itself is unchanged)
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        int item = wave + 4 * i, l, x;
        if (g.contiguous_lines) { l = item >> 1; x = lane + 64 * (item & 1); }
        else { l = lane; x = item; }
        if (l < nvalid) {
            uint2 v = tile[x * kPitch + l];
            if (half_bits_to_float(v.y >> 16) < 0.5f) base[(long long)l * g.fstride + (long long)x * g.xstride] = v;
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
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_lightgrid_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) return PBRK_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_lightgrid_sweep, dim3((g.nf + kTile - 1) / kTile, ns), dim3(256), kLdsBytes, (hipStream_t)stream,
                       (uint2*)image_rgba16f, g);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
