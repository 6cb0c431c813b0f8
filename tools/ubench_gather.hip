// Cost of per-lane scattered 48-byte "cell" fetches through the vector L1 (K5's prefiltered-map taps), development tool:
//   A  48-byte cells, three dwordx4 loads per lane (what k_shade_fast did in round 2a)
//   B  64-byte aligned cells, three dwordx4 loads per lane
//   C  64-byte aligned cells, quad-cooperative: lanes 4k..4k+3 load the four 16-byte pieces of ONE cell (16 cells per instruction,
//      4 instructions per 64 cells), pieces go through wave-private LDS (stride 80 B) back to the lane that wants the cell
//   D  as C, but only the loads (no LDS redistribution; wrong data): isolates the load cost
// Cells are picked at random inside a 21 x 21 window that moves with the wave (K5: +-10 texels of jitter around a smooth centre).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o tools/ubench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define NC 513                      // cells per row of a face level
#define ITER 16

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__device__ __forceinline__ int pick_cell(unsigned wave, int it, int lane, int window) {
    unsigned h = window < 0 ? 12345u : hash(wave * 977u + it * 131071u), g = hash(hash(wave * 977u + it * 131071u) ^ (lane * 2654435761u));
    if (window < 0) window = -window;          // negative: one fixed window for the whole grid (L1-resident: isolates the hit path)
    int cx = 16 + (int)(h % (NC - 32)), cy = 16 + (int)((h >> 12) % (NC - 32));
    int dx = (int)(g % (unsigned)window) - window / 2, dy = (int)((g >> 10) % (unsigned)window) - window / 2;
    return (cy + dy) * NC + cx + dx;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const void* cells, int bytes, float* out, int window) {
    __shared__ __attribute__((aligned(16))) char lds[4][64 * 80];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned wave = blockIdx.x * 4 + wv;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)cells, 0, bytes, 0x00020000);
    float acc = 0.0f;
    for (int it = 0; it < ITER; ++it) {
        int cell = pick_cell(wave, it, lane, window);
        if (MODE == 0 || MODE == 1) {
            int off = cell * (MODE == 0 ? 48 : 64);
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0),
                  c = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, 0);
            acc += __uint_as_float(a.x ^ b.y ^ c.z) + __uint_as_float(a.w ^ b.x ^ c.y);
        } else {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int c = __shfl(cell, 16 * i + (lane >> 2));                    // the cell lane 16 i + (lane / 4) wants
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, c * 64 + (lane & 3) * 16, 0, 0);
            }
            if (MODE == 2) {
                char* base = lds[wv];
#pragma unroll
                for (int i = 0; i < 4; ++i) *(u32x4*)(base + (16 * i + (lane >> 2)) * 80 + (lane & 3) * 16) = v[i];
                __builtin_amdgcn_wave_barrier();
                u32x4 a = *(u32x4*)(base + lane * 80), b = *(u32x4*)(base + lane * 80 + 16), c = *(u32x4*)(base + lane * 80 + 32);
                __builtin_amdgcn_wave_barrier();
                acc += __uint_as_float(a.x ^ b.y ^ c.z) + __uint_as_float(a.w ^ b.x ^ c.y);
            } else {
                acc += __uint_as_float(v[0].x ^ v[1].y ^ v[2].z ^ v[3].w);
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}


// Load-bound variant: the cell of every lane is picked once; the loop only re-issues the fetches (next 64-byte-aligned cell row
// each time, so the compiler cannot hoist them), everything stays L1-resident.  clk per iteration per CU = the memory pipe's cost.
//   0: 3 scattered dwordx4 per lane, 48 B cells     1: 3 scattered dwordx4 per lane, 64 B cells     2: 4 quad-cooperative dwordx4 + LDS
//   3: 3 coherent dwordx4 (lanes adjacent)          4: 1 scattered dwordx4                           5: 1 scattered dword
//   6: 4 quad-cooperative dwordx4, no LDS           7: 4 tri-cooperative dwordx4 on 48 B cells + LDS
template <int MODE>
__global__ __launch_bounds__(256) void k_loads(const void* cells, int bytes, float* out, int window, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[4][64 * 80];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)cells, 0, bytes, 0x00020000);
    unsigned acc = 0;
    int cell = pick_cell(blockIdx.x * 4 + wv, 0, lane, -window);
    if (MODE == 3) cell = (16 + (blockIdx.x * 4 + wv) % 21) * NC + 16 + lane;
    int cq[4];
    for (int i = 0; i < 4; ++i) cq[i] = __shfl(cell, 16 * i + (lane >> 2)) * 64 + (lane & 3) * 16;
    char* base = lds[wv];
    for (int it = 0; it < iters; ++it) {
        int rot = (it & 7) * NC * 64;                   // walk down 8 cell rows and come back
        if (MODE == 0 || MODE == 1 || MODE == 3) {
            int off = cell * (MODE == 0 ? 48 : 64) + (MODE == 0 ? (it & 7) * NC * 48 : rot);
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0),
                  c = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, 0);
            acc += (a.x ^ b.y ^ c.z) + (a.y ^ b.z ^ c.w) + (a.z ^ b.w ^ c.x) + (a.w ^ b.x ^ c.y);
        } else if (MODE == 4) {
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, cell * 64 + rot, 0, 0);
            acc += (a.x ^ a.w) + (a.y ^ a.z);
        } else if (MODE == 5) {
            acc += __builtin_amdgcn_raw_buffer_load_b32(r, cell * 64 + rot, 0, 0);
        } else if (MODE == 11) {                           // SoA cells: piece k of every cell in its own plane (16 B per cell and plane)
            const int plane = NC * NC * 16;
            int off = cell * 16 + (it & 7) * NC * 16;
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(r, off, plane, 0),
                  c = __builtin_amdgcn_raw_buffer_load_b128(r, off, 2 * plane, 0);
            acc += (a.x ^ b.y ^ c.z) + (a.y ^ b.z ^ c.w) + (a.z ^ b.w ^ c.x) + (a.w ^ b.x ^ c.y);
        } else if (MODE == 8) {                            // coalesced: 64 lanes x 4 B = 256 contiguous bytes (a G-buffer plane of 64 pixels)
            acc += __builtin_amdgcn_raw_buffer_load_b32(r, ((blockIdx.x * 4 + wv) % 512) * 4096 + lane * 4 + rot, 0, 0);
        } else if (MODE == 9) {                            // coalesced: 64 lanes x 16 B = 1 KB contiguous (4 pixels of a plane per lane)
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, ((blockIdx.x * 4 + wv) % 512) * 4096 + lane * 16 + rot, 0, 0);
            acc += (a.x ^ a.w) + (a.y ^ a.z);
        } else if (MODE == 10) {                           // coalesced: 64 lanes x 8 B
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, ((blockIdx.x * 4 + wv) % 512) * 4096 + lane * 8 + rot, 0, 0);
            acc += a.x ^ a.y;
        } else if (MODE == 7) {                            // three lanes of a quad fetch one 48-byte cell (16-byte aligned), the fourth sits out
            u32x4 v[4];
            const int piece = (lane & 3) == 3 ? 0x7FFFFFF0 : (lane & 3) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (cq[i] / 64) * 48 + (it & 7) * NC * 48 + piece, 0, 0);
            if ((lane & 3) != 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(u32x4*)(base + (16 * i + (lane >> 2)) * 48 + (lane & 3) * 16) = v[i];
            }
            __builtin_amdgcn_wave_barrier();
            u32x4 a = *(u32x4*)(base + lane * 48), b = *(u32x4*)(base + lane * 48 + 16), c = *(u32x4*)(base + lane * 48 + 32);
            __builtin_amdgcn_wave_barrier();
            acc += (a.x ^ b.y ^ c.z) + (a.y ^ b.z ^ c.w) + (a.z ^ b.w ^ c.x) + (a.w ^ b.x ^ c.y);
        } else {
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, cq[i] + rot, 0, 0);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(u32x4*)(base + (16 * i + (lane >> 2)) * 80 + (lane & 3) * 16) = v[i];
                __builtin_amdgcn_wave_barrier();
                u32x4 a = *(u32x4*)(base + lane * 80), b = *(u32x4*)(base + lane * 80 + 16), c = *(u32x4*)(base + lane * 80 + 32);
                __builtin_amdgcn_wave_barrier();
                acc += (a.x ^ b.y ^ c.z) + (a.y ^ b.z ^ c.w) + (a.z ^ b.w ^ c.x) + (a.w ^ b.x ^ c.y);
            } else {
                for (int i = 0; i < 4; ++i) acc += (v[i].x ^ v[i].y) + (v[i].z ^ v[i].w);
            }
        }
    }
    if (acc == 0x12345678u) out[0] = (float)acc;
}

int main() {
    const int ncell = NC * NC;
    const int bytes = ncell * 64;
    void* d; float* o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8 * 16;            // 16 rounds of 8 blocks per CU
    const char* names[4] = {"A 48B cells, 3 loads/lane", "B 64B cells, 3 loads/lane", "C 64B cells, quad-cooperative + LDS", "D quad-cooperative loads only"};
    for (int window : {21, 5, 101, -21, -15, -5}) {
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, d, bytes, o, window);
                if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, d, bytes, o, window);
                if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, d, bytes, o, window);
                if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, d, bytes, o, window);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            double wave_iters = (double)blocks * 4 * ITER;
            printf("window %3d  %-40s %8.3f ms  %7.1f clk per wave-cell-set per CU (2.4 GHz)\n", window, names[mode], ms,
                   ms * 1e-3 * 2.4e9 * 256 / wave_iters);
        }
    }
    const char* ln[12] = {"3 scattered x4 (48 B cells)", "3 scattered x4 (64 B cells)", "4 quad-coop x4 + LDS", "3 coherent x4", "1 scattered x4", "1 scattered dword", "4 quad-coop x4, no LDS", "4 tri-coop x4 (48 B cells) + LDS",
                          "1 coalesced dword (256 B / wave)", "1 coalesced dwordx4 (1 KB / wave)", "1 coalesced dwordx2 (512 B / wave)", "3 scattered x4, SoA planes"};
    for (int window : {21, 13, 9}) {
        for (int mode = 0; mode < 12; ++mode) {
            float ms = 0; const int iters = 256, lb = 256 * 8 * 2;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k_loads<0>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 1: hipLaunchKernelGGL(k_loads<1>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 2: hipLaunchKernelGGL(k_loads<2>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 3: hipLaunchKernelGGL(k_loads<3>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 4: hipLaunchKernelGGL(k_loads<4>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 5: hipLaunchKernelGGL(k_loads<5>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 7: hipLaunchKernelGGL(k_loads<7>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 8: hipLaunchKernelGGL(k_loads<8>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 9: hipLaunchKernelGGL(k_loads<9>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 10: hipLaunchKernelGGL(k_loads<10>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                case 11: hipLaunchKernelGGL(k_loads<11>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                default: hipLaunchKernelGGL(k_loads<6>, dim3(lb), dim3(256), 0, 0, d, bytes, o, window, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("load-bound, window %2d  %-32s %8.3f ms  %7.1f clk per wave-iteration per CU (2.4 GHz)\n", window, ln[mode], ms,
                   ms * 1e-3 * 2.4e9 * 256 / ((double)lb * 4 * iters));
        }
    }
    return 0;
}
