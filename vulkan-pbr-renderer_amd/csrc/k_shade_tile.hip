// k_shade_tile.hip -- K5, tiled instantiation of the deferred shade pass (lighting_pass.glsl:432-716) for the modes without sun
// shadows and voxel GI: same arithmetic as k_shade_fast.hip (see there for the two classes of arithmetic and k_shade.hip for the
// block-by-block citations), different data movement (round 3):
//
//  * LDS-staged environment windows.  On a large frame a 64 x 16 pixel tile subtends ~0.02 rad, so the reflection vectors of its
//    pixels fall into a small window of each of the two prefiltered levels they interpolate -- what scatters them is the shader's
//    own jitter of +-0.3 * roughness (lighting_pass.glsl:695), which made every lane's three 16-byte cell loads a separate trip
//    through the texture path (TA busy 93 % of the 8K launch in round 2).  Here the tile's centre pixel proposes, for each of its two
//    levels, a window of WIN_A x WIN_A / WIN_B x WIN_B cells around its own tap; the 16 waves stage both windows with coalesced
//    direct-to-LDS loads (global_load_lds_dwordx4: whole rows of the cells twin, no VGPRs), and every lane whose tap position
//    falls inside a window reads its cell from LDS.  Lanes outside (other face, other level, a far-off normal) take the
//    range-checked buffer loads of the fast kernel: the result never depends on where a cell came from, only the time does.
//  * Per-column / per-row constants from host tables.  (x + .5) / width * 2 - 1 and the three interleaved-gradient-noise products
//    of a column (and of a row) depend on the pixel coordinate alone: two float4 tables, computed on the host with the shader's
//    operation order in correctly rounded fp32, replace 24 instructions per pixel by one coalesced 16-byte load and one scalar load.
#include "k_shade_internal.h"

#ifndef TILE_H
#define TILE_H 4           // rows (= waves) per workgroup
#endif
#define TILE_W 64          // one wave per tile row: the row index is wave-uniform (scalar row table)
#ifndef WIN_A
#define WIN_A 16           // window of the centre pixel's lower level, in cells per side (16 * 16 * 48 B = 12 KB)
#endif
#ifndef WIN_B
#define WIN_B 10           // ... of its upper level (half the footprint): 4.7 KB
#endif

typedef unsigned int u32x4t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bl4t(__amdgpu_buffer_rsrc_t r, int off) {
    u32x4t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#define GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

// Stage a WW x WW window of one level's cells twin.  The window is WW rows of WW * 3 contiguous float4 ("pieces") in the twin and
// one flat array of WW * WW * 3 pieces in LDS; a wave-instruction moves 64 consecutive pieces (lane-linear LDS destination, per-lane
// source address: the tail of one row and the head of the next travel together), waves take the 64-piece chunks round-robin.
template <int WW>
__device__ __forceinline__ void stage_window(float4* win, const float4* __restrict__ cells_level, int face, int oi, int oj, int nc, int wave, int lane) {
    constexpr int ROWLEN = WW * 3, PIECES = WW * WW * 3;
    const float4* origin = cells_level + (size_t)((face * nc + oj) * nc + oi) * 3;
    for (int e0 = wave * 64; e0 < PIECES; e0 += TILE_H * 64) {
        const int e = e0 + lane;
        if (e < PIECES) {
            const int row = e / ROWLEN, c = e - row * ROWLEN;          // compile-time divisor
            GLDS16(origin + (size_t)row * nc * 3 + c, win + e0);
        }
    }
}

template <bool kIBL, bool kShafts>
__global__ __launch_bounds__(TILE_H * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shade_tile(const ShadeParams p) {
    __shared__ float4 winA[WIN_A * WIN_A * 3];
    __shared__ float4 winB[WIN_B * WIN_B * 3];
    __shared__ int keys[8];           // per window: level (-1: none), face, first cell column, first cell row
    __shared__ int lv_off[16];        // float4 offset of level l inside the prefiltered cells twin
#ifdef PBR_K5_DEBUG
    const int dbg = p.dbg;
#else
    const int dbg = 0;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wrow = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 16) lv_off[tid] = cells_level_off(p.pre_size, 0, min(tid, p.pre_levels - 1));
    const int lx = blockIdx.x * TILE_W + lane, ly = blockIdx.y * TILE_H + wrow;
    const bool valid = lx < p.w && ly < p.h;
    const int px = p.x0 + lx, py = p.y0 + ly;                         // py is wave-uniform
    // the tile's centre pixel (clamped into the image): its taps place the windows
    const int leader = min(TILE_H / 2, p.h - 1 - (int)blockIdx.y * TILE_H) * 64 + min(TILE_W / 2, p.w - 1 - (int)blockIdx.x * TILE_W);
    const int pi4 = (py * p.width + px) * 4;                          // all five G-buffer planes hold 4 bytes per pixel
    const int plane_bytes = p.width * p.height * 4;
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.base, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)p.normal, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.orm, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)p.emissive, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.depth, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rcol = __builtin_amdgcn_make_buffer_rsrc((void*)p.col_tab, 0, p.width * 16, 0x00020000);
    const unsigned nn = __builtin_amdgcn_raw_buffer_load_b32(rn, pi4, 0, 0), oo = __builtin_amdgcn_raw_buffer_load_b32(ro, pi4, 0, 0);
    const float depth = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, pi4, 0, 0));
    const float4 ct = bl4t(rcol, px * 16);                            // { xn, .06711056 (x + .5), .06711056 (x + 90.5), .06711056 (x + 522.5) }
    const unsigned bb = __builtin_amdgcn_raw_buffer_load_b32(rb, pi4, 0, 0), ee = __builtin_amdgcn_raw_buffer_load_b32(re, pi4, 0, 0);
    // scalar load (constant address space + wave-uniform index): { yn, .00583715 (y + .5), .00583715 (y + 20.5), .00583715 (y + 55.5) }
    typedef float v4ft __attribute__((ext_vector_type(4)));
    const v4ft rtv = ((const __attribute__((address_space(4))) v4ft*)p.row_tab)[min(py, p.height - 1)];
    const float4 rt = make_float4(rtv.x, rtv.y, rtv.z, rtv.w);

    // :433-442.  N and roughness feed the exact chain (exact b/255); the rest is continuous (b * fl(1/255), within one ulp)
    const float k255 = 1.0f / 255.0f;
    const f3 N = mk3(fmaf(unorm8(nn & 255u), 2.0f, -1.0f), fmaf(unorm8((nn >> 8) & 255u), 2.0f, -1.0f), fmaf(unorm8((nn >> 16) & 255u), 2.0f, -1.0f));
    const float roughness = unorm8((oo >> 8) & 255u);

    // :690 irradiance(N) (IBL mode): depends on the G-buffer alone; its three loads are issued ahead of the long exact chain
    f3 amb = mk3(0.0f, 0.0f, 0.0f);
    if (kIBL) {
        const float nf = p.irr_nf;
        float fid = __builtin_amdgcn_cubeid(N.x, N.y, N.z);
        float sc = __builtin_amdgcn_cubesc(N.x, N.y, N.z), tc = __builtin_amdgcn_cubetc(N.x, N.y, N.z);
        float h = __builtin_amdgcn_rcpf(fabsf(__builtin_amdgcn_cubema(N.x, N.y, N.z))) * nf;
        float off1 = p.irr_off1;                                      // 0.5 nf + 0.5
        float u = fmaf(sc, h, off1), v = fmaf(tc, h, off1);
        float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
        float ncf = nf + 1.0f;
        int off = (int)(fmaf(fmaf(fid, ncf, v - b), ncf, u - a) * (float)PBR_CELL_BYTES);       // (face * nc + j0) * nc + i0, exact in fp32
        __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.irr_cells, 0, 6 * (p.irr_size + 1) * (p.irr_size + 1) * PBR_CELL_BYTES, 0x00020000);
        amb = cells_bilerp(bl4t(ri, off), bl4t(ri, off + 16), bl4t(ri, off + 32), a, b);
    }

    // :444-451 (exact: the shader's operation order; xn, yn from the tables)
    float pw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r] = ((p.wfc[r] * ct.x + p.wfc[4 + r] * rt.x) + p.wfc[8 + r] * depth) + p.wfc[12 + r];
    SharedRcp rpw; rpw.d = pw[3]; rpw.r = rcp_nr(pw[3]);
    const f3 P = mk3(div_by(pw[0], rpw), div_by(pw[1], rpw), div_by(pw[2], rpw));

    // :456-459 (exact; x - floor(x) == v_fract for the non-negative arguments here)
    const float noise_offset = p.noise_offset;                         // (1000 * 1.61803398875f) * frame_idx_mod_59, rounded on the host as the shader rounds it
    const float noise_1 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.y + rt.y)) + noise_offset);
    const float noise_2 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.z + rt.z)) + noise_offset);
    const float noise_3 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.w + rt.w)) + noise_offset);

    const f3 cam = mk3(p.cam[0], p.cam[1], p.cam[2]);
    const f3 V = normalize3_nr(sub3(cam, P));                                          // :612 (exact)
    const bool sky = !(fabsf(P.x) <= 99.0f) || !(fabsf(P.y) <= 99.0f) || !(fabsf(P.z) <= 99.0f);   // :708 (== clamp(x) != x, NaN included)
    const float dNV = dot3(N, V);                                                      // exact: -dNV is dot(N, I) of :694

    // ---- the one lookup of the prefiltered map: along -V at lod 1 (sky, :708-710) or along the jittered reflection vector at
    //      lod 4 * roughness (:693-699).  Direction and lod here, tap positions below, taps after the windows are staged.
    f3 d = mk3(-V.x, -V.y, -V.z);
    float lod = 1.0f;
    const bool fetch = valid && (sky || kIBL);
    if (kIBL && !sky) {
        // :693-697 (exact: feeds the taps of the prefiltered fetch)
        const float dNI2 = 2.0f * -dNV;
        f3 R = mk3(-V.x - dNI2 * N.x, -V.y - dNI2 * N.y, -V.z - dNI2 * N.z);
        const float jr = 0.6f * roughness;
        R = normalize3_nr(mk3(R.x + jr * (noise_1 - 0.5f), R.y + jr * (noise_2 - 0.5f), R.z + jr * (noise_3 - 0.5f)));
        const float r2 = roughness * roughness, r4 = r2 * r2;
        d = mk3(mix_(R.x, N.x, r4), mix_(R.y, N.y, r4), mix_(R.z, N.z, r4));
        lod = roughness * 4.0f;
    }
    // sampler coordinates of the direction (exact): s = (0.5 sc) / |ma| + 0.5; tap positions and weights on both levels
    const float fid = __builtin_amdgcn_cubeid(d.x, d.y, d.z);
    float wl;                                                          // weight of the upper level
    int l0, l1, ci0, cj0, ci1, cj1;                                    // levels and bordered tap coordinates (cell indices in [0, n])
    float a0, b0, a1, b1;
    {
        float sc = __builtin_amdgcn_cubesc(d.x, d.y, d.z), tc = __builtin_amdgcn_cubetc(d.x, d.y, d.z);
        SharedRcp rma; rma.d = 0.5f * fabsf(__builtin_amdgcn_cubema(d.x, d.y, d.z)); rma.r = rcp_nr(rma.d);
        float s = div_by(0.5f * sc, rma) + 0.5f, t = div_by(0.5f * tc, rma) + 0.5f;
        lod = fminf(fmaxf(lod, 0.0f), p.pre_maxl);
        float fl = floorf(lod);
        wl = lod - fl;
        l0 = (int)fl; l1 = min(l0 + 1, p.pre_levels - 1);
        // level sizes are powers of two: n_l = W * 2^-l exactly, so s * n is exact and u = (s * n) - 0.5 as the sampler states it
        float n0 = ldexpf(p.pre_wf, -l0), n1 = ldexpf(p.pre_wf, -l1);
        float u0 = fmaf(s, n0, -0.5f), v0 = fmaf(t, n0, -0.5f), u1 = fmaf(s, n1, -0.5f), v1 = fmaf(t, n1, -0.5f);
        float fu0 = floorf(u0), fv0 = floorf(v0), fu1 = floorf(u1), fv1 = floorf(v1);
        a0 = u0 - fu0; b0 = v0 - fv0; a1 = u1 - fu1; b1 = v1 - fv1;
        ci0 = (int)fu0 + 1; cj0 = (int)fv0 + 1; ci1 = (int)fu1 + 1; cj1 = (int)fv1 + 1;
    }
    const int face = (int)fid;

    if (tid == leader) {                                               // always a valid pixel
        // Window placement does not touch the result (a cell is the same 48 bytes wherever it is read from), so it may use
        // relaxed arithmetic: the windows are centred on the tap of the UNJITTERED direction -- the jitter of +-0.3 roughness
        // per component (:695) is what spreads the tile's taps, symmetrically around it.
        int wi0 = ci0, wj0 = cj0, wi1 = ci1, wj1 = cj1;
        if (kIBL && !sky) {
            const float dNI2 = 2.0f * -dNV;
            f3 R0 = mk3(-V.x - dNI2 * N.x, -V.y - dNI2 * N.y, -V.z - dNI2 * N.z);
            const float rl = __builtin_amdgcn_rsqf(dot3(R0, R0));
            const float r2 = roughness * roughness, r4 = r2 * r2;
            const f3 d0 = mk3(mix_(R0.x * rl, N.x, r4), mix_(R0.y * rl, N.y, r4), mix_(R0.z * rl, N.z, r4));
            if (__builtin_amdgcn_cubeid(d0.x, d0.y, d0.z) == fid) {                      // same face as the leader's own tap: else keep that
                const float h = 0.5f * __builtin_amdgcn_rcpf(0.5f * fabsf(__builtin_amdgcn_cubema(d0.x, d0.y, d0.z)));
                const float s0 = fmaf(__builtin_amdgcn_cubesc(d0.x, d0.y, d0.z), h, 0.5f), t0 = fmaf(__builtin_amdgcn_cubetc(d0.x, d0.y, d0.z), h, 0.5f);
                const float n0 = ldexpf(p.pre_wf, -l0), n1 = ldexpf(p.pre_wf, -l1);
                wi0 = (int)floorf(fmaf(s0, n0, 0.5f)); wj0 = (int)floorf(fmaf(t0, n0, 0.5f));
                wi1 = (int)floorf(fmaf(s0, n1, 0.5f)); wj1 = (int)floorf(fmaf(t0, n1, 0.5f));
            }
        }
        const int nc0 = (p.pre_size >> l0) + 1, nc1 = (p.pre_size >> l1) + 1;
        keys[0] = (fetch && nc0 >= WIN_A && !(dbg & 1)) ? l0 : -1; keys[1] = face;
        keys[2] = min(max(wi0 - WIN_A / 2, 0), nc0 - WIN_A); keys[3] = min(max(wj0 - WIN_A / 2, 0), nc0 - WIN_A);
        keys[4] = (fetch && l1 != l0 && nc1 >= WIN_B && !(dbg & 1)) ? l1 : -1; keys[5] = face;
        keys[6] = min(max(wi1 - WIN_B / 2, 0), nc1 - WIN_B); keys[7] = min(max(wj1 - WIN_B / 2, 0), nc1 - WIN_B);
    }
    __syncthreads();
    const int kA_l = __builtin_amdgcn_readfirstlane(keys[0]), kA_f = __builtin_amdgcn_readfirstlane(keys[1]);
    const int kA_oi = __builtin_amdgcn_readfirstlane(keys[2]), kA_oj = __builtin_amdgcn_readfirstlane(keys[3]);
    const int kB_l = __builtin_amdgcn_readfirstlane(keys[4]), kB_f = __builtin_amdgcn_readfirstlane(keys[5]);
    const int kB_oi = __builtin_amdgcn_readfirstlane(keys[6]), kB_oj = __builtin_amdgcn_readfirstlane(keys[7]);
    constexpr int wwA = WIN_A, wwB = WIN_B;
    if (kA_l >= 0) stage_window<WIN_A>(winA, p.pre_cells + __builtin_amdgcn_readfirstlane(lv_off[kA_l]), kA_f, kA_oi, kA_oj, (p.pre_size >> kA_l) + 1, wrow, lane);
    if (kB_l >= 0) stage_window<WIN_B>(winB, p.pre_cells + __builtin_amdgcn_readfirstlane(lv_off[kB_l]), kB_f, kB_oi, kB_oj, (p.pre_size >> kB_l) + 1, wrow, lane);

    // ---- everything that does not need the taps runs while the windows are in flight
    f3 outl = mk3(0.0f, 0.0f, 0.0f);
    f3 specw = mk3(1.0f, 1.0f, 1.0f);                                  // what the prefiltered colour is multiplied with (sky: 1)
    if (!sky) {
        const float metallic = (float)((oo >> 16) & 255u) * k255;
        const f3 base = mk3((float)(bb & 255u) * k255, (float)((bb >> 8) & 255u) * k255, (float)((bb >> 16) & 255u) * k255);
        const float VdotN = fmaxf(dNV, 0.0f);                                           // :613
        if (kShafts) {                                                                  // :622-651 with visibility == 1
            float sp[4], cp4[4];
            mat_mul(p.ssw, P.x + N.x * 0.1f, P.y + N.y * 0.1f, P.z + N.z * 0.1f, 1.0f, sp);
            mat_mul(p.ssw, cam.x, cam.y, cam.z, 1.0f, cp4);
            f3 delta = mk3(sp[0] - cp4[0], sp[1] - cp4[1], sp[2] - cp4[2]);
            float dist = sqrtf(dot3(delta, delta));
            const float step = 1.0f / 16.0f;
            float travelled = step * noise_1;
            for (int it = 0; it < 4096; ++it) {                                         // bounded: non-sky pixels lie within +-99 world units
                travelled += step;
                if (travelled > dist) break;
                outl.x += 0.001f * 1.0f * (25.0f * 1.0f); outl.y += 0.001f * 1.0f * (25.0f * 0.9f); outl.z += 0.001f * 1.0f * (25.0f * 0.7f);
            }
        }
        // :657-661 (continuous)
        const f3 F0 = mk3(fmaf(metallic, base.x - 0.04f, 0.04f), fmaf(metallic, base.y - 0.04f, 0.04f), fmaf(metallic, base.z - 0.04f, 0.04f));
        const float omm = 1.0f - metallic;
        const float p5v = pow5(1.0f - VdotN);
        const f3 kD = mk3((1.0f - fmaf(1.0f - F0.x, p5v, F0.x)) * omm, (1.0f - fmaf(1.0f - F0.y, p5v, F0.y)) * omm, (1.0f - fmaf(1.0f - F0.z, p5v, F0.z)) * omm);
        const f3 kdb = mk3(kD.x * base.x, kD.y * base.y, kD.z * base.z);
        {   // :664-679
            const f3 Ls = mk3(-p.sun[0], -p.sun[1], -p.sun[2]);
            const float NdotL = fmaxf(dot3(N, Ls), 0.0f);                               // exact: decides the branch
            if (NdotL > 0.0f) {
                const f3 H = normalize3_nr(add3(Ls, V));                                // exact (N.H below)
                const float NdotH = fmaxf(dot3(N, H), 0.0f);
                const float VdotH = fmaxf(fmaf(V.x, H.x, fmaf(V.y, H.y, V.z * H.z)), 0.0f);
                const float a = roughness * roughness, a2 = a * a;
                float denom = NdotH * NdotH * (a2 - 1.0f) + 1.0f;                       // the cancellation the exact chain exists for: shader order
                const float D = a2 * __builtin_amdgcn_rcpf(PBR_PI * denom * denom);
                const float t2 = 2.0f * NdotH * __builtin_amdgcn_rcpf(VdotH);
                const float G = fminf(1.0f, fminf(t2 * VdotN, t2 * NdotL));
                const float p5h = pow5(1.0f - VdotH);
                const float gd = G * D * __builtin_amdgcn_rcpf(fmaxf(4.0f * NdotL * VdotN, 0.0001f));
                const float rpi = 1.0f / PBR_PI;
                const float e = 25.0f * NdotL;
                outl.x = fmaf(fmaf(fmaf(1.0f - F0.x, p5h, F0.x), gd, kdb.x * rpi), e, outl.x);
                outl.y = fmaf(fmaf(fmaf(1.0f - F0.y, p5h, F0.y), gd, kdb.y * rpi), e * 0.9f, outl.y);
                outl.z = fmaf(fmaf(fmaf(1.0f - F0.z, p5h, F0.z), gd, kdb.z * rpi), e * 0.7f, outl.z);
            }
        }
        if (kIBL) {
            // :681 LUT fetch, one 16-byte cell (continuous).  v = max(roughness, .05) lies in [.05, 1] and u = N.V is >= 0, but N is
            // whatever the G-buffer holds (|N| > 1 for many byte triples): u may exceed 1, so the column clamps like the sampler
            float sbx, sby;
            {
                const float S = p.lut_sf;
                float fx = fmaf(VdotN, S, -0.5f), fy = fmaf(fmaxf(roughness, 0.05f), S, -0.5f);
                float flx = floorf(fx), fly = floorf(fy);
                float a = fx - flx, b = fy - fly;
                int off = (int)(fmaf(fly + 1.0f, S + 1.0f, fminf(flx + 1.0f, S)) * 16.0f);
                __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)p.lut_cells, 0, (p.lut_size + 1) * (p.lut_size + 1) * 16, 0x00020000);
                u32x4t c = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 0);
                unsigned c0 = c.x, c1 = c.y, c2 = c.z, c3 = c.w;
                float2 t00 = __half22float2(*reinterpret_cast<__half2*>(&c0)), t10 = __half22float2(*reinterpret_cast<__half2*>(&c1));
                float2 t01 = __half22float2(*reinterpret_cast<__half2*>(&c2)), t11 = __half22float2(*reinterpret_cast<__half2*>(&c3));
                sbx = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
                sby = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
            }
            outl.x = fmaf(kdb.x, amb.x, outl.x); outl.y = fmaf(kdb.y, amb.y, outl.y); outl.z = fmaf(kdb.z, amb.z, outl.z);   // :687
            specw = mk3(fmaf(F0.x, sbx, sby), fmaf(F0.y, sbx, sby), fmaf(F0.z, sbx, sby));                                     // :702
        }
    }
    __syncthreads();                                                   // the staged windows have landed (the barrier's fence waits for the LDS-DMA)

    if (fetch) {
        __amdgpu_buffer_rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_cells, 0, p.pre_cells_bytes, 0x00020000);
        f3 c0, c1;
        bool inA = l0 == kA_l && face == kA_f && (unsigned)(ci0 - kA_oi) < (unsigned)wwA && (unsigned)(cj0 - kA_oj) < (unsigned)wwA;
        bool inB = l1 == kB_l && face == kB_f && (unsigned)(ci1 - kB_oi) < (unsigned)wwB && (unsigned)(cj1 - kB_oj) < (unsigned)wwB;
#ifdef PBR_K5_DEBUG
        if (dbg & 4) {
            unsigned long long* st = (unsigned long long*)p.dbg_stats;
            unsigned long long mf = __ballot(1), ma = __ballot(inA), mb = __ballot(inB);
            if (lane == __builtin_ctzll(mf)) { atomicAdd(st, (unsigned long long)__popcll(mf)); atomicAdd(st + 1, (unsigned long long)__popcll(ma)); atomicAdd(st + 2, (unsigned long long)__popcll(mb));
                                              atomicAdd(st + 3, 1ull); atomicAdd(st + 4, (unsigned long long)(ma == mf)); atomicAdd(st + 5, (unsigned long long)(mb == mf)); }
        }
        if (dbg & 2) {                                                 // timing only: every lane reads LDS (wrong picture)
            if (!inA) { ci0 = kA_oi + ((unsigned)ci0 % (unsigned)wwA); cj0 = kA_oj + ((unsigned)cj0 % (unsigned)wwA); inA = true; }
            if (!inB) { ci1 = kB_oi + ((unsigned)ci1 % (unsigned)wwB); cj1 = kB_oj + ((unsigned)cj1 % (unsigned)wwB); inB = true; }
        }
#endif
        if (inA) {
            const float4* q = winA + ((cj0 - kA_oj) * wwA + (ci0 - kA_oi)) * 3;
            c0 = cells_bilerp(q[0], q[1], q[2], a0, b0);
        } else {
            const int nc = (p.pre_size >> l0) + 1;
            const int off = (((face * nc + cj0) * nc + ci0) * 3 + lv_off[l0]) * 16;
            c0 = cells_bilerp(bl4t(rpre, off), bl4t(rpre, off + 16), bl4t(rpre, off + 32), a0, b0);
        }
        if (inB) {
            const float4* q = winB + ((cj1 - kB_oj) * wwB + (ci1 - kB_oi)) * 3;
            c1 = cells_bilerp(q[0], q[1], q[2], a1, b1);
        } else {
            const int nc = (p.pre_size >> l1) + 1;
            const int off = (((face * nc + cj1) * nc + ci1) * 3 + lv_off[l1]) * 16;
            c1 = cells_bilerp(bl4t(rpre, off), bl4t(rpre, off + 16), bl4t(rpre, off + 32), a1, b1);      // wl == 0 leaves c0 untouched: no branch
        }
        const f3 spec = mk3(fmaf(wl, c1.x - c0.x, c0.x), fmaf(wl, c1.y - c0.y, c0.y), fmaf(wl, c1.z - c0.z, c0.z));
        if (sky) outl = spec;
        else { outl.x = fmaf(spec.x, specw.x, outl.x); outl.y = fmaf(spec.y, specw.y, outl.y); outl.z = fmaf(spec.z, specw.z, outl.z); }
    }
    if (!valid) return;
    if (!sky) {                                                                         // :706 (after the specular term, as in the shader: same sums as k_shade_fast, bit for bit)
        const float k10 = 10.0f / 255.0f;
        outl = add3(outl, mk3((float)(ee & 255u) * k10, (float)((ee >> 8) & 255u) * k10, (float)((ee >> 16) & 255u) * k10));
    }
    outl = mk3(fmaxf(outl.x, 0.0f), fmaxf(outl.y, 0.0f), fmaxf(outl.z, 0.0f));          // :712
    const size_t pi = (size_t)py * p.width + px;
    if (p.out_fmt == PBRK_FMT_RGBA16F) {
        __half2 lo = __halves2half2(__float2half_rn(outl.x), __float2half_rn(outl.y));
        __half2 hi = __halves2half2(__float2half_rn(outl.z), __float2half_rn(1.0f));
        uint2 packed;
        packed.x = *reinterpret_cast<unsigned*>(&lo);
        packed.y = *reinterpret_cast<unsigned*>(&hi);
        ((uint2*)p.out)[pi] = packed;
    } else {
        ((float4*)p.out)[pi] = make_float4(outl.x, outl.y, outl.z, 1.0f);
    }
}

int launch_shade_tile(const ShadeParams& p, bool ibl, bool shafts, hipStream_t stream) {
    dim3 grid((p.w + TILE_W - 1) / TILE_W, (p.h + TILE_H - 1) / TILE_H);
    if (ibl && shafts) hipLaunchKernelGGL((k_shade_tile<true, true>), grid, dim3(TILE_H * 64), 0, stream, p);
    else if (ibl) hipLaunchKernelGGL((k_shade_tile<true, false>), grid, dim3(TILE_H * 64), 0, stream, p);
    else if (shafts) hipLaunchKernelGGL((k_shade_tile<false, true>), grid, dim3(TILE_H * 64), 0, stream, p);
    else hipLaunchKernelGGL((k_shade_tile<false, false>), grid, dim3(TILE_H * 64), 0, stream, p);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
