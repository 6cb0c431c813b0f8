// k_shade_fast.hip -- K5, fast instantiation of the deferred shade pass (lighting_pass.glsl:432-716) for the modes without sun
// shadows and voxel GI; see k_shade.hip for the general kernel and the block-by-block citations.
#include "k_shade_internal.h"

// ==========================================================================================
// Fast instantiation for the modes without sun shadows and voxel GI (IBL mode, the reference's default frame): 489 VALU
// instructions per surface pixel instead of ~870.
//
// Two classes of arithmetic.  (1) The chain G-buffer -> P -> V -> {H, R} -> {N.H, cube taps of the prefiltered map} is
// ill-conditioned at the 1e-4 tolerance: GGX amplifies an error in N.H by 1/a^2 (1300x at roughness 1/6), and one ulp of a
// reflection direction moves a bilinear weight next to a 1e4:1 sun texel by more than the whole tolerance.  It is therefore
// evaluated in the shader's operation order with correctly rounded results -- but through short sequences: v_rcp / v_rsq plus
// one Newton step give the correctly rounded reciprocal / square root unless the exact value lies within ~1e-14 (relative) of a
// rounding boundary (2e-7 of all operands; the deviation is then one ulp), and quotients by a shared divisor take Markstein's
// three instructions (pbr_device.h).  (2) Everything downstream of those (Fresnel, G, the D quotient, kD, LUT and irradiance
// fetches, composition) is continuous and uses FMAs, 1-ulp reciprocals and the fast unorm8 decode.
// All texture reads go through range-checked buffer loads with 32-bit offsets (no 64-bit address arithmetic, no clamps).
// ==========================================================================================
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bl4(__amdgpu_buffer_rsrc_t r, int off) {
    u32x4s v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// one level of the prefiltered cells twin: exact tap selection and weights (s, t are the sampler coordinates, exact)
__device__ __forceinline__ f3 pre_level_fetch(__amdgpu_buffer_rsrc_t rc, float fid, float s, float t, float nf, int level_bytes) {
    float u = fmaf(s, nf, -0.5f), v = fmaf(t, nf, -0.5f);          // n is a power of two: s*n is exact, so this is (s*n) - 0.5 as the sampler states it
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    float ncf = nf + 1.0f;
    float cellf = fmaf(fmaf(fid, ncf, fv + 1.0f), ncf, fu + 1.0f);  // (face * nc + j0) * nc + i0, bordered tap coordinates, exact in fp32
    int off = (int)(cellf * (float)PBR_CELL_BYTES) + level_bytes;
    return cells_bilerp(bl4(rc, off), bl4(rc, off + 16), bl4(rc, off + 32), a, b);
}

template <bool kIBL, bool kShafts, bool kTab = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shade_fast(const ShadeParams p) {
    __shared__ int lv_off[16];        // byte offset of level l inside the prefiltered cells twin
    if (threadIdx.x < 16) lv_off[threadIdx.x] = cells_level_off(p.pre_size, 0, min((int)threadIdx.x, p.pre_levels - 1)) * 16;
    __syncthreads();
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ly = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (lx >= p.w || ly >= p.h) return;
    const int px = p.x0 + lx, py = p.y0 + ly;
    const int pi4 = (py * p.width + px) * 4;                          // all five G-buffer planes hold 4 bytes per pixel
    const int plane_bytes = p.width * p.height * 4;
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.base, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)p.normal, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.orm, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)p.emissive, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.depth, 0, plane_bytes, 0x00020000);
    const unsigned bb = __builtin_amdgcn_raw_buffer_load_b32(rb, pi4, 0, 0), nn = __builtin_amdgcn_raw_buffer_load_b32(rn, pi4, 0, 0);
    const unsigned oo = __builtin_amdgcn_raw_buffer_load_b32(ro, pi4, 0, 0), ee = __builtin_amdgcn_raw_buffer_load_b32(re, pi4, 0, 0);
    const float depth = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, pi4, 0, 0));

    // :433-442.  N and roughness feed the exact chain (exact b/255); the rest is continuous (b * fl(1/255), within one ulp)
    const float k255 = 1.0f / 255.0f;
    const f3 N = mk3(fmaf(unorm8(nn & 255u), 2.0f, -1.0f), fmaf(unorm8((nn >> 8) & 255u), 2.0f, -1.0f), fmaf(unorm8((nn >> 16) & 255u), 2.0f, -1.0f));
    const float roughness = unorm8((oo >> 8) & 255u);
    const float metallic = (float)((oo >> 16) & 255u) * k255;
    const f3 base = mk3((float)(bb & 255u) * k255, (float)((bb >> 8) & 255u) * k255, (float)((bb >> 16) & 255u) * k255);
    const float k10 = 10.0f / 255.0f;
    const f3 emissive = mk3((float)(ee & 255u) * k10, (float)((ee >> 8) & 255u) * k10, (float)((ee >> 16) & 255u) * k10);

    // :444-451 (exact)
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    float xn, yn, noise_1, noise_2, noise_3;
    const float noise_offset = (1000 * 1.61803398875f) * p.frame_idx_mod_59;
    if (kTab) {
        // per-column / per-row constants from the host tables (k_shade.hip shade_tables: the same operations, correctly rounded):
        // col[x] = { xn, .06711056 fcx, .06711056 (fcx + 90), .06711056 (fcx + 522) }, row[y] likewise with .00583715 and 20 / 55
        __amdgpu_buffer_rsrc_t rcol = __builtin_amdgcn_make_buffer_rsrc((void*)p.col_tab, 0, p.width * 16, 0x00020000);
        const float4 ct = bl4(rcol, px * 16);
        typedef float v4ft __attribute__((ext_vector_type(4)));
        const v4ft rt = ((const __attribute__((address_space(4))) v4ft*)p.row_tab)[__builtin_amdgcn_readfirstlane(py)];     // a wave is one row: scalar load
        xn = ct.x; yn = rt.x;
        noise_1 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.y + rt.y)) + noise_offset);
        noise_2 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.z + rt.z)) + noise_offset);
        noise_3 = __builtin_amdgcn_fractf(__builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(ct.w + rt.w)) + noise_offset);
    } else {
        SharedRcp rw, rh;
        rw.d = (float)p.width; rw.r = p.rcp_width; rh.d = (float)p.height; rh.r = p.rcp_height;
        xn = fmaf(div_by(fcx, rw), 2.0f, -1.0f); yn = fmaf(div_by(fcy, rh), 2.0f, -1.0f);     // 2u is exact: (u*2) - 1
        // :456-459 (exact; x - floor(x) == v_fract for the non-negative arguments here)
        auto ignf = [](float x, float y) { return __builtin_amdgcn_fractf(52.9829189f * __builtin_amdgcn_fractf(0.06711056f * x + 0.00583715f * y)); };
        noise_1 = __builtin_amdgcn_fractf(ignf(fcx, fcy) + noise_offset);
        noise_2 = __builtin_amdgcn_fractf(ignf(fcx + 90.0f, fcy + 20.0f) + noise_offset);
        noise_3 = __builtin_amdgcn_fractf(ignf(fcx + 522.0f, fcy + 55.0f) + noise_offset);
    }
    float pw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[r] = ((p.wfc[r] * xn + p.wfc[4 + r] * yn) + p.wfc[8 + r] * depth) + p.wfc[12 + r];
    SharedRcp rpw; rpw.d = pw[3]; rpw.r = rcp_nr(pw[3]);
    const f3 P = mk3(div_by(pw[0], rpw), div_by(pw[1], rpw), div_by(pw[2], rpw));

    const f3 cam = mk3(p.cam[0], p.cam[1], p.cam[2]);
    const f3 V = normalize3_nr(sub3(cam, P));                                          // :612 (exact)
    const bool sky = !(fabsf(P.x) <= 99.0f) || !(fabsf(P.y) <= 99.0f) || !(fabsf(P.z) <= 99.0f);   // :708 (== clamp(x) != x, NaN included)

    // :690 irradiance(N) (IBL mode): depends on the G-buffer alone; its three loads are issued as soon as the wave knows that it holds a
    // surface pixel at all (an all-sky wave -- most waves of a frame with sky -- skips them) and are consumed ~250 instructions later.
    // A smooth 32^2 map: one v_rcp projection.
    f3 amb = mk3(0.0f, 0.0f, 0.0f);
    if (kIBL && __builtin_amdgcn_ballot_w64(!sky) != 0ull) {
        const float nf = (float)p.irr_size;
        float fid = __builtin_amdgcn_cubeid(N.x, N.y, N.z);
        float sc = __builtin_amdgcn_cubesc(N.x, N.y, N.z), tc = __builtin_amdgcn_cubetc(N.x, N.y, N.z);
        float h = __builtin_amdgcn_rcpf(fabsf(__builtin_amdgcn_cubema(N.x, N.y, N.z))) * nf;
        float off1 = 0.5f * nf + 0.5f;
        float u = fmaf(sc, h, off1), v = fmaf(tc, h, off1);
        float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
        float ncf = nf + 1.0f;
        int off = (int)(fmaf(fmaf(fid, ncf, v - b), ncf, u - a) * (float)PBR_CELL_BYTES);       // (face * nc + j0) * nc + i0, exact in fp32
        __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.irr_cells, 0, 6 * (p.irr_size + 1) * (p.irr_size + 1) * PBR_CELL_BYTES, 0x00020000);
        amb = cells_bilerp(bl4(ri, off), bl4(ri, off + 16), bl4(ri, off + 32), a, b);
    }

    __amdgpu_buffer_rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_cells, 0, p.pre_cells_bytes, 0x00020000);
    const float maxl = (float)(p.pre_levels - 1);
    const float wf = (float)p.pre_size;
    f3 outl = mk3(0.0f, 0.0f, 0.0f);
    // sampler coordinates of a direction (exact): s = (0.5 sc) / |ma| + 0.5
    auto pre_fetch = [&](f3 d, float lod) {
        float fid = __builtin_amdgcn_cubeid(d.x, d.y, d.z);
        float sc = __builtin_amdgcn_cubesc(d.x, d.y, d.z), tc = __builtin_amdgcn_cubetc(d.x, d.y, d.z);
        SharedRcp rma; rma.d = 0.5f * fabsf(__builtin_amdgcn_cubema(d.x, d.y, d.z)); rma.r = rcp_nr(rma.d);
        float s = div_by(0.5f * sc, rma) + 0.5f, t = div_by(0.5f * tc, rma) + 0.5f;
        lod = fminf(fmaxf(lod, 0.0f), maxl);
        float fl = floorf(lod), w = lod - fl;
        int l0 = (int)fl, l1 = min(l0 + 1, p.pre_levels - 1);
        // level sizes are powers of two: n_l = W * 2^-l exactly
        f3 c0 = pre_level_fetch(rpre, fid, s, t, ldexpf(wf, -l0), lv_off[l0]);
        // The upper level only where some lane of the wave blends it in: a sky pixel asks for lod 1.0 exactly (w == 0), and most waves of a
        // frame with sky are all-sky -- their three upper-level loads are half of what such a wave fetches.  fma(0, c1 - c0, c0) == c0.
        if (__builtin_amdgcn_ballot_w64(w != 0.0f) == 0ull) return c0;
        f3 c1 = pre_level_fetch(rpre, fid, s, t, ldexpf(wf, -l1), lv_off[l1]);          // a lane with w == 0 keeps c0: no per-lane branch
        return mk3(fmaf(w, c1.x - c0.x, c0.x), fmaf(w, c1.y - c0.y, c0.y), fmaf(w, c1.z - c0.z, c0.z));
    };

    f3 fdir = mk3(-V.x, -V.y, -V.z);                                                    // :708-710: a sky pixel shows level 1 along the view ray
    float flod = 1.0f;
    f3 fscale = mk3(1.0f, 1.0f, 1.0f);
    if (!sky) {
        const float dNV = dot3(N, V);                                                   // exact: -dNV is dot(N, I) of :694
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
                const float S = (float)p.lut_size;
                float fx = fmaf(VdotN, S, -0.5f), fy = fmaf(fmaxf(roughness, 0.05f), S, -0.5f);
                float flx = floorf(fx), fly = floorf(fy);
                float a = fx - flx, b = fy - fly;
                int off = (int)(fmaf(fly + 1.0f, S + 1.0f, fminf(flx + 1.0f, S)) * 16.0f);
                __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)p.lut_cells, 0, (p.lut_size + 1) * (p.lut_size + 1) * 16, 0x00020000);
                u32x4s c = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 0);
                unsigned c0 = c.x, c1 = c.y, c2 = c.z, c3 = c.w;
                float2 t00 = __half22float2(*reinterpret_cast<__half2*>(&c0)), t10 = __half22float2(*reinterpret_cast<__half2*>(&c1));
                float2 t01 = __half22float2(*reinterpret_cast<__half2*>(&c2)), t11 = __half22float2(*reinterpret_cast<__half2*>(&c3));
                sbx = lerp_fma(lerp_fma(t00.x, t10.x, a), lerp_fma(t01.x, t11.x, a), b);
                sby = lerp_fma(lerp_fma(t00.y, t10.y, a), lerp_fma(t01.y, t11.y, a), b);
            }
            outl.x = fmaf(kdb.x, amb.x, outl.x); outl.y = fmaf(kdb.y, amb.y, outl.y); outl.z = fmaf(kdb.z, amb.z, outl.z);   // :687
            // :693-697 (exact: feeds the taps of the prefiltered fetch)
            const float dNI2 = 2.0f * -dNV;
            f3 R = mk3(-V.x - dNI2 * N.x, -V.y - dNI2 * N.y, -V.z - dNI2 * N.z);
            const float jr = 0.6f * roughness;
            R = normalize3_nr(mk3(R.x + jr * (noise_1 - 0.5f), R.y + jr * (noise_2 - 0.5f), R.z + jr * (noise_3 - 0.5f)));
            const float r2 = roughness * roughness, r4 = r2 * r2;
            R = mk3(mix_(R.x, N.x, r4), mix_(R.y, N.y, r4), mix_(R.z, N.z, r4));
            fdir = R; flod = roughness * 4.0f;                                           // :699
            fscale = mk3(fmaf(F0.x, sbx, sby), fmaf(F0.y, sbx, sby), fmaf(F0.z, sbx, sby));  // :702
        }
    }
    if (kIBL || sky) {
        // one fetch sequence for sky and surface lanes of a wave (the sky's level-1 lookup along the view ray and the surface's
        // jittered reflection differ in their operands only)
        const f3 spec = pre_fetch(fdir, flod);
        if (sky) outl = spec;
        else { outl.x = fmaf(spec.x, fscale.x, outl.x); outl.y = fmaf(spec.y, fscale.y, outl.y); outl.z = fmaf(spec.z, fscale.z, outl.z); }
    }
    if (!sky) outl = add3(outl, emissive);                                              // :706
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

int launch_shade_fast(const ShadeParams& p, bool ibl, bool shafts, hipStream_t stream) {
    // kTab: measured SLOWER than recomputing (8K frame 564-576 vs 540-544 us, 1080p 24.8 vs 23.3 us: 24 fewer VALU instructions do not
    // pay for one more vector-memory instruction per wave -- the kernel is bound by its memory instructions); opt-in
    static int tab = -1;
    if (tab < 0) { const char* e = getenv("PBR_SHADE_TABLES"); tab = e ? atoi(e) : 0; }
    if (tab && p.col_tab && p.row_tab) {
        dim3 grid((p.w + 63) / 64, (p.h + 3) / 4);
        if (ibl && shafts) hipLaunchKernelGGL((k_shade_fast<true, true, true>), grid, dim3(256), 0, stream, p);
        else if (ibl) hipLaunchKernelGGL((k_shade_fast<true, false, true>), grid, dim3(256), 0, stream, p);
        else if (shafts) hipLaunchKernelGGL((k_shade_fast<false, true, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((k_shade_fast<false, false, true>), grid, dim3(256), 0, stream, p);
        return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
    }
    dim3 grid((p.w + 63) / 64, (p.h + 3) / 4);
    if (ibl && shafts) hipLaunchKernelGGL((k_shade_fast<true, true>), grid, dim3(256), 0, stream, p);
    else if (ibl) hipLaunchKernelGGL((k_shade_fast<true, false>), grid, dim3(256), 0, stream, p);
    else if (shafts) hipLaunchKernelGGL((k_shade_fast<false, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((k_shade_fast<false, false>), grid, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
