// k_mc_region.hip -- K4b / K3 for source levels too big for LDS as a whole: taps served from LDS-staged REGIONS.
//
// Same sum as k_mc.hip (gen_prefiltered_env_map.glsl:115-146, gen_irradiance_map.glsl:81-97):
//   out.rgb(texel) = ( sum_i w_i * bilinear(src, frame(texel) * l_i) ) / divisor
//
// Why: the direct kernel moves 48 B per (texel, sample) through the vector L1 (64 B/clk/CU): 48 clk per wave-sample per
// CU whatever the hit rate, above its ~40 clk of VALU work.  LDS delivers 256 B/clk/CU, but only levels up to 32^2 fit as a
// whole.  Here the bordered source level is cut into regions of at most 66 x 66 texels (70 KB: a whole face at n = 64, a
// quarter face at n = 128), and a 1024-thread workgroup (16 x 16 output texels x 4 slices of the sample table, two workgroups
// per CU) walks the regions one after the other:
//   1. binning (once per tile): every sample direction is pushed through the tile-centre frame; a rigorous bound on how far
//      any texel of the tile can move it (|M_texel - M_centre|_F) yields the regions it can reach; one bit per (region,
//      sample) in LDS.  ~1 % of the work.
//   2. per flagged region: stage it in LDS, then every wave runs the flagged samples of its slice.  The face is known per
//      pass, so the cube projection is a static signed permutation folded into the frame (no v_cube*), each lane tests
//      exactly whether ITS direction falls into this region's cells (the hardware's tie rule: z >= y >= x), and lanes that
//      do not are masked (weight 0): they meet the sample again in the pass of their own region.
//   3. every wave counts the (lane, sample) pairs it has accumulated (scalar popcount of the exec mask).  A total short of
//      64 x the slice's sample count would mean the bound of step 1 missed a region; the wave then recomputes its slice with
//      direct loads (never observed; counter in stats[0]).
//   2b. (round 3, quarter-face levels) binning also PROVES, for 94 % of the samples, that every texel of the tile taps the SAME
//      region of the same face (the bound of step 1 read as "certainly" instead of "possibly"): those run a body without the
//      in-face / in-region tests, their six ballots, the exec masking and the pair count -- 37 instead of 45 instructions.
//      Measured (one box, A/B): C4 mip 1 36.7 -> 35.1 ms.  The same on the whole-face levels (39 -> 37 instructions) LOSES
//      3-5 % -- the wave-uniform choice of body per sample costs more than two comparisons -- so it is compiled for SUB only;
//      two separate loops (proved samples, then the rest) were slower on every level.
// Each (texel, sample) pair is accumulated exactly once, in an order (region, then sample index) that depends on the texel
// only, not on the tile: a row-sharded dispatch equals a full one bit for bit.
#include "pbr_device.h"
#include "pbr_kernels.h"
#include "k_mc_internal.h"

#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));
// The sample table is read-only for the whole launch: a pointer into the constant address space makes every wave-uniform
// read of it a scalar load (behind the kernel's barriers / LDS atomics the compiler no longer proves that for a global pointer
// and falls back to 64-lane vector loads of one address).
typedef const __attribute__((address_space(4))) v4f* ctab_t;
typedef const __attribute__((address_space(3))) v4f* lds_v4f_p;

// A 1024-thread workgroup owns a TILE x TILE block of output texels and 1024 / TILE^2 slices of the sample table:
//   TILE 16 (the only instantiation): 256 texels x 4 slices (waves 4s .. 4s+3 own slice s).
// The smaller tile halves the frame spread delta the region flags are built from (fewer samples flagged for two regions) and
// pays four times the binning / staging per texel: it wins where regions are small next to that spread (n_src <= 32).
#define REG_MAX_S 16

struct RegArgs {
    McArgs a;
    int G;                      // regions per face edge
    int RC;                     // tap positions (cells) per region edge; the last region of a row may hold fewer
    int NR;                     // 6 * G * G
    int NW;                     // mask words per region = ceil(n_tab / 32)
    int expect[REG_MAX_S];      // samples per slice
    int tile_y0;                // MFMA variant: first tile row (multiple of 16) covering a.y0
    int pole_row[2];            // faces +X / -X: tile row (relative to the dispatch's first row, clamped) nearest to the pole of the tangent frame
    unsigned long long* stats;  // optional: [0] += healed wave-slices, [1] += all wave-slices, [2] += (region, sample) flags, [3] += samples per tile, [4] += regions visited
};

// direction -> (sc, tc, ma) of face f: the table of v_cubesc / v_cubetc / v_cubema (gen_prefiltered_env_map.glsl:12-23)
__device__ __forceinline__ void face_coords(int f, float x, float y, float z, float& sc, float& tc, float& ma) {
    switch (f) {
    case 0: sc = -z; tc = -y; ma = x; break;
    case 1: sc = z; tc = -y; ma = -x; break;
    case 2: sc = x; tc = z; ma = y; break;
    case 3: sc = x; tc = -z; ma = -y; break;
    case 4: sc = x; tc = -y; ma = z; break;
    default: sc = -x; tc = -y; ma = -z; break;
    }
}

// One sample of a pass: accumulate it in the lanes whose direction falls into the staged region.
// CLS = major axis of the region's face (0: x, 1: y, 2: z): the hardware's tie rule (z >= y >= x) in two comparisons.
template <int RS, bool SUB, int CLS>
__device__ __forceinline__ void region_sample(const v4f e, unsigned lds_base, f3 Pb, f3 Pt, f3 Pr, float half_n, float off,
                                              float ulo, float uhi, float vlo, float vhi,
                                              float& ar, float& ag, float& ab, unsigned& cnt) {
    // (sc, tc, ma) = permuted frame * local direction, same FMA order as the direct kernel's L
    const float sc = fmaf(e.x, Pb.x, fmaf(e.y, Pt.x, e.z * Pr.x));
    const float tc = fmaf(e.x, Pb.y, fmaf(e.y, Pt.y, e.z * Pr.y));
    const float ma = fmaf(e.x, Pb.z, fmaf(e.y, Pt.z, e.z * Pr.z));
    // in-face test with the hardware's tie rule; each comparison's lane mask is taken by its own ballot (the compiler folds
    // a ballot of ONE comparison into the v_cmp's SGPR result, not a ballot of their conjunction)
    bool c1, c2;
    if (CLS == 0) { c1 = ma > fabsf(sc); c2 = ma > fabsf(tc); }
    else if (CLS == 1) { c1 = ma >= fabsf(sc); c2 = ma > fabsf(tc); }
    else { c1 = ma >= fabsf(sc); c2 = ma >= fabsf(tc); }
    unsigned long long inm = __builtin_amdgcn_ballot_w64(c1) & __builtin_amdgcn_ballot_w64(c2);
    // Lanes whose direction is not on this face (or, SUB, not in this region's cells) sit the sample out under the exec
    // mask; a wave none of whose lanes is on the face skips the rest.  The (lane, sample) pairs taken are counted per wave
    // with scalar instructions.
    if (inm == 0) return;
    bool in = c1 && c2;
    const float h = __builtin_amdgcn_rcpf(ma) * half_n;
    const float u = fmaf(sc, h, off), v = fmaf(tc, h, off);                   // bordered tap coordinates, [0.5, n + 0.5]
    const int il = (int)u, jl = (int)v;
    if (SUB) {
        // floor(u) in [ox, ox + rcx) <=> u in [ox, ox + rcx) (integer bounds): four float comparisons, no integer arithmetic
        const bool c3 = u >= ulo, c4 = u < uhi, c5 = v >= vlo, c6 = v < vhi;
        inm &= __builtin_amdgcn_ballot_w64(c3) & __builtin_amdgcn_ballot_w64(c4) & __builtin_amdgcn_ballot_w64(c5) & __builtin_amdgcn_ballot_w64(c6);
        in = in && c3 && c4 && c5 && c6;
    }
    cnt += (unsigned)__builtin_popcountll(inm);
    if (in) {
        const float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
        // whole 16-byte texels: ds_read_b128 runs at 256 B/clk/CU, the 12-byte form the compiler would pick at 96 (the
        // empty asm keeps the fourth component alive); LDS byte address = jl * row + (il << 4) + base (the region's origin
        // is folded into the base) in one shift-add and one 24-bit multiply-add
        unsigned t16, addr;
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(t16) : "v"(il), "s"(lds_base));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(jl), "s"(RS * 16), "v"(t16));
        lds_v4f_p tp = (lds_v4f_p)(unsigned long long)addr;
        v4f q00 = tp[0], q10 = tp[1], q01 = tp[RS], q11 = tp[RS + 1];
        asm("" : "+v"(q00)); asm("" : "+v"(q10)); asm("" : "+v"(q01)); asm("" : "+v"(q11));
        // weights of the four taps with the sample weight folded in
        const float wgt = e.w;
        const float wa = wgt * a;
        const float w11 = wa * b;
        const float w10 = wa - w11;
        const float wt = wgt - wa;
        const float w01 = wt * b;
        const float w00 = wt - w01;
        ar = fmaf(w11, q11.x, fmaf(w01, q01.x, fmaf(w10, q10.x, fmaf(w00, q00.x, ar))));
        ag = fmaf(w11, q11.y, fmaf(w01, q01.y, fmaf(w10, q10.y, fmaf(w00, q00.y, ag))));
        ab = fmaf(w11, q11.z, fmaf(w01, q01.z, fmaf(w10, q10.z, fmaf(w00, q00.z, ab))));
    }
}

// The same sample when binning has proved that EVERY texel of the tile taps this region of this face: all 64 lanes are in, so
// the tests, ballots and the exec mask fall away; u, v, taps, weights and the FMA order are those of region_sample, bit for bit.
template <int RS>
__device__ __forceinline__ void certain_sample(const v4f e, unsigned lds_base, f3 Pb, f3 Pt, f3 Pr, float half_n, float off,
                                               float& ar, float& ag, float& ab) {
    const float sc = fmaf(e.x, Pb.x, fmaf(e.y, Pt.x, e.z * Pr.x));
    const float tc = fmaf(e.x, Pb.y, fmaf(e.y, Pt.y, e.z * Pr.y));
    const float ma = fmaf(e.x, Pb.z, fmaf(e.y, Pt.z, e.z * Pr.z));
    const float h = __builtin_amdgcn_rcpf(ma) * half_n;
    const float u = fmaf(sc, h, off), v = fmaf(tc, h, off);
    const int il = (int)u, jl = (int)v;
    const float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
    unsigned t16, addr;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(t16) : "v"(il), "s"(lds_base));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(jl), "s"(RS * 16), "v"(t16));
    lds_v4f_p tp = (lds_v4f_p)(unsigned long long)addr;
    v4f q00 = tp[0], q10 = tp[1], q01 = tp[RS], q11 = tp[RS + 1];
    asm("" : "+v"(q00)); asm("" : "+v"(q10)); asm("" : "+v"(q01)); asm("" : "+v"(q11));
    const float wgt = e.w;
    const float wa = wgt * a;
    const float w11 = wa * b;
    const float w10 = wa - w11;
    const float wt = wgt - wa;
    const float w01 = wt * b;
    const float w00 = wt - w01;
    ar = fmaf(w11, q11.x, fmaf(w01, q01.x, fmaf(w10, q10.x, fmaf(w00, q00.x, ar))));
    ag = fmaf(w11, q11.y, fmaf(w01, q01.y, fmaf(w10, q10.y, fmaf(w00, q00.y, ag))));
    ab = fmaf(w11, q11.z, fmaf(w01, q01.z, fmaf(w10, q10.z, fmaf(w00, q00.z, ab))));
}

// One pass over the flagged samples of this wave's slice for the staged region.  Samples are taken two at a time so that the
// second table entry's scalar load is in flight while the first sample computes.
// CERT: cwords holds, per mask word, the samples proved to be in this region for the whole tile (certain_sample).
template <int RS, bool SUB, int CLS, int REG_S, bool CERT>
__device__ __forceinline__ void region_pass(unsigned lds_base, const unsigned* __restrict__ mwords, const unsigned* __restrict__ cwords, int NW, int s,
                                            ctab_t tab, f3 Pb, f3 Pt, f3 Pr, float half_n, float off,
                                            float ulo, float uhi, float vlo, float vhi,
                                            float& ar, float& ag, float& ab, unsigned& cnt) {
    unsigned mnext = s < NW ? (unsigned)__builtin_amdgcn_readfirstlane((int)mwords[s]) : 0u;
    unsigned cnext = (CERT && s < NW) ? (unsigned)__builtin_amdgcn_readfirstlane((int)cwords[s]) : 0u;
    for (int w = s; w < NW; w += REG_S) {
        unsigned m = mnext;
        const unsigned c = cnext;
        mnext = w + REG_S < NW ? (unsigned)__builtin_amdgcn_readfirstlane((int)mwords[w + REG_S]) : 0u;
        if (CERT) {
            cnext = w + REG_S < NW ? (unsigned)__builtin_amdgcn_readfirstlane((int)cwords[w + REG_S]) : 0u;
            cnt += 64u * (unsigned)__builtin_popcount(m & c);              // every lane takes every proved sample
        }
        ctab_t tw = tab + (w << 5);
        while (m) {
            const int i0 = __builtin_ctz(m);
            m &= m - 1u;
            const bool two = m != 0u;
            const int i1 = two ? __builtin_ctz(m) : i0;
            m &= m - 1u;                                                   // no-op when m is already 0
            const v4f e0 = tw[i0];
            const v4f e1 = tw[i1];
            // samples in index order whichever body they take: the order of a texel's sum stays (region, sample index)
            if (CERT && ((c >> i0) & 1u)) certain_sample<RS>(e0, lds_base, Pb, Pt, Pr, half_n, off, ar, ag, ab);
            else region_sample<RS, SUB, CLS>(e0, lds_base, Pb, Pt, Pr, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
            if (two) {
                if (CERT && ((c >> i1) & 1u)) certain_sample<RS>(e1, lds_base, Pb, Pt, Pr, half_n, off, ar, ag, ab);
                else region_sample<RS, SUB, CLS>(e1, lds_base, Pb, Pt, Pr, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
            }
        }
    }
}

// Binning (once per tile): which regions of the source level can the samples reach from ANY texel of the tile?  Every sample
// direction is pushed through the tile-centre frame; a rigorous bound on how far a texel's own frame can move it yields the
// regions; one bit per (region, sample) in `masks`, any[r] != 0 when region r has a bit.  Leaves with a barrier pending: callers
// synchronise before reading the masks.
// cmask (optional, [NW], directly behind dmax and zeroed here with the rest): bit i set when sample i is PROVED to tap one region
// of one face from every texel of the tile -- certainly on the face (ma' - |sc'| >= (ma - |sc|) - sqrt(2) delta > 0), its tap
// bounds inside the face and inside one region's cells.  Such a sample has exactly one region flag.
__device__ __forceinline__ void region_bin(unsigned* masks, unsigned* any, unsigned* dmax, int NR, int NW, int G, int RC, int n,
                                           f3 R, f3 T, f3 B, f3 Rc, f3 Tc, f3 Bc, ctab_t tab, int n_tab, int tid, unsigned* cmask = nullptr) {
    const float nf = (float)n;
    const float half_n = 0.5f * nf;
    const float off = 0.5f * nf + 0.5f;
    for (int k = tid; k < NR * NW + NR + 1 + (cmask ? NW : 0); k += 1024) masks[k] = 0u;
    __syncthreads();
    {
        f3 dR = sub3(R, Rc), dT = sub3(T, Tc), dB = sub3(B, Bc);
        float d2 = dot3(dR, dR) + dot3(dT, dT) + dot3(dB, dB);
        atomicMax(dmax, __float_as_uint(sqrtf(d2)));                 // non-negative floats order like their bit patterns
    }
    __syncthreads();
    // |L_texel - L_centre| <= |M_texel - M_centre|_2 for a unit local direction.  Both frames are orthonormal, so M_t - M_c =
    // (Rot - I) M_c with singular values {0, 2 sin(theta/2), 2 sin(theta/2)}: the spectral norm is the Frobenius norm / sqrt(2).
    // Inflated for the frames' own rounding (1e-7) and that of both evaluations.
    const float delta = __uint_as_float(*dmax) * 0.70710678f * 1.001f + 4e-6f;
    for (int i = tid; i < n_tab; i += 1024) {
        const v4f e = tab[i];
        const float Lx = fmaf(e.x, Bc.x, fmaf(e.y, Tc.x, e.z * Rc.x));
        const float Ly = fmaf(e.x, Bc.y, fmaf(e.y, Tc.y, e.z * Rc.y));
        const float Lz = fmaf(e.x, Bc.z, fmaf(e.y, Tc.z, e.z * Rc.z));
        const unsigned bit = 1u << (i & 31);
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            float sc, tc, ma;
            face_coords(f, Lx, Ly, Lz, sc, tc, ma);
            // A direction L' with |L' - L|_2 <= delta has ma' - |sc'| <= (ma - |sc|) + sqrt(2) delta: face f needs that >= 0 (same for tc)
            const float d2 = 1.41421357f * delta;
            if (!(ma + d2 >= fabsf(sc)) || !(ma + d2 >= fabsf(tc))) continue;
            int lo_u = 0, hi_u = n, lo_v = 0, hi_v = n;
            const float mlo = ma - delta;
            bool certain = false;
            if (mlo > 0.2f) {
                // g(L) = sc / ma has |grad g| = sqrt(1 + g^2) / ma; along the segment L -> L' (ma >= ma - delta, |g| <= (|sc| + delta) /
                // (ma - delta)) that is bounded, so |g(L') - g(L)| <= delta sqrt(1 + gmax^2) / (ma - delta); + 0.05 texel for rcp / fma rounding
                const float rm = 1.0f / ma, rl = 1.0f / mlo;
                const float ru = sc * rm, rv = tc * rm;
                const float gu = (fabsf(sc) + delta) * rl, gv = (fabsf(tc) + delta) * rl;
                const float mu = delta * sqrtf(fmaf(gu, gu, 1.0f)) * rl * half_n * 1.0001f + 0.05f;
                const float mv = delta * sqrtf(fmaf(gv, gv, 1.0f)) * rl * half_n * 1.0001f + 0.05f;
                const float uc = fmaf(ru, half_n, off), vc = fmaf(rv, half_n, off);
                const float ul = floorf(uc - mu), uh = floorf(uc + mu), vl = floorf(vc - mv), vh = floorf(vc + mv);
                if (uh < 0.0f || ul > nf || vh < 0.0f || vl > nf) continue;      // cannot be on this face at all
                lo_u = (int)fmaxf(ul, 0.0f); hi_u = (int)fminf(uh, nf);
                lo_v = (int)fmaxf(vl, 0.0f); hi_v = (int)fminf(vh, nf);
                certain = cmask && ma - d2 > fabsf(sc) && ma - d2 > fabsf(tc) && ul >= 0.0f && uh <= nf && vl >= 0.0f && vh <= nf;
            }
            const int gx0 = lo_u / RC, gx1 = hi_u / RC, gy0 = lo_v / RC, gy1 = hi_v / RC;
            if (certain && gx0 == gx1 && gy0 == gy1) atomicOr(&cmask[i >> 5], bit);
            for (int gy = gy0; gy <= gy1; ++gy)
                for (int gx = gx0; gx <= gx1; ++gx) {
                    const int r = (f * G + gy) * G + gx;
                    atomicOr(&masks[r * NW + (i >> 5)], bit);
                    any[r] = 1u;
                }
        }
    }
}

template <int RS, bool SUB, int TILE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mc_region(const RegArgs q) {
    constexpr int REG_TX = TILE * TILE, REG_S = 1024 / REG_TX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    float4* region = (float4*)smem_r;
    const unsigned lds_base = (unsigned)(unsigned long long)smem_r;      // LDS byte offset of the staged region (low half of the flat address)
    unsigned* masks = (unsigned*)(smem_r + RS * RS * 16);
    unsigned* any = masks + q.NR * q.NW;
    unsigned* dmax = any + q.NR;
    constexpr bool CERT = SUB;                                           // see the header, 2b
    unsigned* cmask = dmax + 1;                                          // [NW] (CERT) samples proved to tap one region from the whole tile
    const McArgs& p = q.a;
    const int tid = threadIdx.x;
    const int s = __builtin_amdgcn_readfirstlane(tid / REG_TX);
    const int t = tid % REG_TX;

    // Tiles in plain block order: consecutive tiles go round-robin over the 8 XCDs.  An XCD-contiguous remap (one eighth of the
    // grid per XCD) put all tiles around the pole of the tangent frame -- where every sample is flagged for several regions and a tile
    // takes up to 3x as long -- on ONE XCD, and the launch waited for it (a single-face dispatch of a +-X face: 10.2 vs 7.6 ms).
    // The whole level fits every XCD's L2, so locality has nothing to lose.
    unsigned tile = blockIdx.x;
    const int face = p.face0 + (int)(tile / (unsigned)p.tiles_per_face);
    const int tf = (int)(tile % (unsigned)p.tiles_per_face);
    int ty = tf / p.tiles_x;
    const int tx = tf % p.tiles_x;
    // Longest tiles first: around the pole of the tangent frame (tangent_of: inside the +X face, its antipode inside -X) the frames
    // of a tile twist against each other, a sample lands in several regions and a tile takes up to 3x as long.  In row order those
    // tiles came last on the -X face (pole at 3/4 of its height) and the launch ended in their tail; here the rows of these two
    // faces are dealt outwards from the pole row, so the long tiles start first and the short ones fill in behind them.
    if (face < 2) {
        const int ny = p.tiles_per_face / p.tiles_x, pr = q.pole_row[face];
        const int a = min(pr, ny - 1 - pr);
        if (ty <= 2 * a) { const int h = (ty + 1) >> 1; ty = (ty & 1) ? pr + h : pr - h; }
        else { const int rest = ty - 2 * a; ty = (pr > ny - 1 - pr) ? pr - a - rest : pr + a + rest; }
    }
    // a wave covers an 8 x 8 quadrant of the tile (not 16 x 4): the smaller its extent, the fewer samples its lanes spread over
    // two regions (PBR_MC_WAVE_SHAPE experiment: see DESIGN.md)
    const int q8 = t >> 6, l8 = t & 63;
    const int x = tx * TILE + (q8 & 1) * 8 + (l8 & 7);
    const int y = p.y0 + ty * TILE + (q8 >> 1) * 8 + (l8 >> 3);
    const int xc = min(x, p.size - 1), yc = min(y, p.y0 + p.rows - 1);

    const f3 R = face_texel_dir(face, xc, yc, p.size);
    const f3 T = tangent_of(R);
    const f3 B = cross3(T, R);
    // tile-centre frame (evaluated redundantly per lane: wave-uniform values)
    const f3 Rc = face_texel_dir(face, min(tx * TILE + TILE / 2, p.size - 1), min(p.y0 + ty * TILE + TILE / 2, p.y0 + p.rows - 1), p.size);
    const f3 Tc = tangent_of(Rc);
    const f3 Bc = cross3(Tc, Rc);

    const int n = p.n_src, nb = n + 2;
    const float nf = (float)n;
    const float half_n = 0.5f * nf;
    const float off = 0.5f * nf + 0.5f;
    ctab_t tab = (ctab_t)(unsigned long long)p.tab;
    const int NW = q.NW, NR = q.NR, G = q.G, RC = q.RC;

    // ---- 1. binning ----
    region_bin(masks, any, dmax, NR, NW, G, RC, n, R, T, B, Rc, Tc, Bc, tab, p.n_tab, tid, CERT ? cmask : nullptr);

    // ---- 2. region passes ----
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    unsigned cnt = 0;
    if (q.stats) {                                                 // diagnostics: (region, sample) flags of this tile, samples, regions visited
        __syncthreads();
        unsigned fl = 0;
        for (int k = tid; k < NR * NW; k += 1024) fl += __popc(masks[k]);
        if (fl) atomicAdd(&q.stats[2], (unsigned long long)fl);
        if (tid == 0) { atomicAdd(&q.stats[3], (unsigned long long)p.n_tab); unsigned v = 0; for (int r = 0; r < NR; ++r) v += any[r] != 0u; atomicAdd(&q.stats[4], (unsigned long long)v);
                        if (CERT) { unsigned c = 0; for (int k = 0; k < NW; ++k) c += __popc(cmask[k]); atomicAdd(&q.stats[5], (unsigned long long)c); } }
    }
    for (int r = 0; r < NR; ++r) {
        __syncthreads();                                           // binning done / readers of the previous region done
        if (any[r] == 0u) continue;                                // workgroup-uniform
        const int f = r / (G * G);
        const int gy = (r / G) % G, gx = r % G;
        const int ox = gx * RC, oy = gy * RC;
        const int rcx = min(RC, n + 1 - ox), rcy = min(RC, n + 1 - oy);      // cells of this region; texels: one more
        const float4* __restrict__ fsrc = p.src + ((size_t)f * nb + oy) * nb + ox;
        for (int k = tid; k < RS * RS; k += 1024) {
            const int ry = k / RS, rx = k - ry * RS;
            if (rx <= rcx && ry <= rcy) region[k] = fsrc[ry * nb + rx];
        }
        __syncthreads();
        // signed permutation of the frame for this face: rows give (sc, tc, ma) directly
        f3 Pb, Pt, Pr;
        face_coords(f, B.x, B.y, B.z, Pb.x, Pb.y, Pb.z);
        face_coords(f, T.x, T.y, T.z, Pt.x, Pt.y, Pt.z);
        face_coords(f, R.x, R.y, R.z, Pr.x, Pr.y, Pr.z);
        const unsigned* mw = masks + r * NW;
        const unsigned pass_base = lds_base - (unsigned)(oy * RS + ox) * 16u;      // taps are addressed with face coordinates
        switch (f >> 1) {
        case 0: region_pass<RS, SUB, 0, REG_S, CERT>(pass_base, mw, cmask, NW, s, tab, Pb, Pt, Pr, half_n, off, (float)ox, (float)(ox + rcx), (float)oy, (float)(oy + rcy), ar, ag, ab, cnt); break;
        case 1: region_pass<RS, SUB, 1, REG_S, CERT>(pass_base, mw, cmask, NW, s, tab, Pb, Pt, Pr, half_n, off, (float)ox, (float)(ox + rcx), (float)oy, (float)(oy + rcy), ar, ag, ab, cnt); break;
        default: region_pass<RS, SUB, 2, REG_S, CERT>(pass_base, mw, cmask, NW, s, tab, Pb, Pt, Pr, half_n, off, (float)ox, (float)(ox + rcx), (float)oy, (float)(oy + rcy), ar, ag, ab, cnt); break;
        }
    }

    // ---- 3. completeness check; a wave that missed a sample recomputes its slice with direct loads ----
    // cnt is a wave total (scalar): no (texel, sample) pair can be taken twice -- the in-region tests partition the tap positions
    // exactly -- so the total is right exactly when no lane missed a sample
    const unsigned expect = 64u * (unsigned)q.expect[s];
    const bool healed = cnt != expect;
    if (healed) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
        ar = 0.0f; ag = 0.0f; ab = 0.0f;
        for (int w = s; w < NW; w += REG_S) {
            const int i1 = min((w << 5) + 32, p.n_tab);
            for (int i = w << 5; i < i1; ++i) {
                const v4f e = tab[i];
                f3 L;
                L.x = fmaf(e.x, B.x, fmaf(e.y, T.x, e.z * R.x));
                L.y = fmaf(e.x, B.y, fmaf(e.y, T.y, e.z * R.y));
                L.z = fmaf(e.x, B.z, fmaf(e.y, T.z, e.z * R.z));
                f3 c = sample_bordered<false>(rs, L, nf, off, nb, nb * 16);
                ar = fmaf(e.w, c.x, ar); ag = fmaf(e.w, c.y, ag); ab = fmaf(e.w, c.z, ab);
            }
        }
    }
    if (q.stats && (tid & 63) == 0) {
        if (healed) atomicAdd(&q.stats[0], 1ull);
        atomicAdd(&q.stats[1], 1ull);
    }

    // ---- 4. combine the slices (fixed tree) and store ----
    __syncthreads();                                               // everybody is done with the staged region
    float* red = (float*)smem_r;
    red[(s * REG_TX + t) * 3 + 0] = ar;
    red[(s * REG_TX + t) * 3 + 1] = ag;
    red[(s * REG_TX + t) * 3 + 2] = ab;
    __syncthreads();
    for (int stride = REG_S / 2; stride >= 1; stride >>= 1) {
        if (s < stride) {
            const int a2 = (s * REG_TX + t) * 3, b2 = ((s + stride) * REG_TX + t) * 3;
            red[a2 + 0] += red[b2 + 0];
            red[a2 + 1] += red[b2 + 1];
            red[a2 + 2] += red[b2 + 2];
        }
        __syncthreads();
    }
    // The texel's coordinates are formed AGAIN from the thread index (opaque to the optimiser) instead of being kept alive across
    // the passes: at the 64-VGPR budget the quarter-face instantiation otherwise spills them (4 VGPRs, 16 bytes of scratch per lane
    // and tile: harmless in time, but rocprofv3's WRITE_SIZE of C4 mip 1 read 853 MB against 403 MB of output).
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    const int t_o = tid_o % REG_TX, q8_o = t_o >> 6, l8_o = t_o & 63;
    const int x_o = tx * TILE + (q8_o & 1) * 8 + (l8_o & 7);
    const int y_o = p.y0 + ty * TILE + (q8_o >> 1) * 8 + (l8_o >> 3);
    if (x_o < p.size && y_o < p.y0 + p.rows && s == 0) {
        float4 o;
        o.x = red[t_o * 3 + 0] / p.divisor; o.y = red[t_o * 3 + 1] / p.divisor; o.z = red[t_o * 3 + 2] / p.divisor; o.w = p.alpha;
        p.out[((size_t)face * p.size + y_o) * p.size + x_o] = o;
    }
}

// ==========================================================================================
// MFMA variant: the frame transform on the matrix pipe.
//
// L = e.x B + e.y T + e.z R is a K = 3 contraction: (samples x 3) . (3 x texels) per cube-coordinate.  v_mfma_f32_16x16x4_f32
// (exact fp32, the k-ordered fmaf chain of the scalar code bit for bit) computes it for 16 samples x 16 texels at a time on the
// matrix pipe, which runs beside the VALU; K = 4's spare slot carries the sample weight (B row (0,0,0,1)).  That fixes the lane
// layout: a wave owns 16 texels (a 4 x 4 block of the tile; lane & 15) and its four lane groups (lane >> 4) take the sample rows
// 4g .. 4g+3 of every 16-sample block; the 16 waves of a workgroup cover the 16 x 16 tile and all walk the same sample list.
//   * the per-(region, sample) flags are compacted once per pass into a list of sample indices in LDS (chunks of LIST_CAP);
//   * per block: one 2-byte LDS read + one 4-byte table gather per lane (the A operand: lane (row, k) holds component k of
//     sample `row` in the order (z, y, x, w)), four MFMAs (sc, tc, ma, weight; B operand = the lane's own column of the
//     permuted frame), then the four samples of the lane run the in-region test / LDS taps / accumulate exactly as above;
//   * the four groups' sums are folded with a fixed xor-butterfly at the end.
// 30 VALU instructions per sample and lane instead of 39.  The order of a texel's sum (region, list position, row) depends on
// the tile's flags, so tiles are anchored at multiples of 16 rows of the LEVEL: a row-sharded dispatch runs the same tiles.
//
// MEASURED (MI355X, C4 mip 2): correct (bit-for-bit the same taps and weights, zero recomputed waves), 16.5 % fewer VALU and
// 30 % fewer SALU instructions per wave (SQ_INSTS_VALU 92.6k -> 77.3k) -- and SLOWER: 48.1 ms against 43.9 ms.  The fp32-input
// MFMA runs at the fp32 vector rate because it runs ON the vector multipliers: its four 32-cycle issues per block hold the
// SIMD for as long as the 36 FMAs they replace, plus a 40-cycle dependency and a gather per block.  There is no idle matrix
// pipe to offload fp32 work to on gfx950.  The variant stays as an opt-in (PBR_MC_MFMA=1) record of that result.
// ==========================================================================================
#define LIST_CAP 2048
typedef const __attribute__((address_space(3))) unsigned short* lds_u16_p;

template <int RS, bool SUB, int CLS>
__device__ __forceinline__ void mfma_sample(const float sc, const float tc, const float ma, const float wgt, const unsigned long long vmask,
                                            unsigned lds_base, float half_n, float off, float ulo, float uhi, float vlo, float vhi,
                                            float& ar, float& ag, float& ab, unsigned& cnt) {
    bool c1, c2;
    if (CLS == 0) { c1 = ma > fabsf(sc); c2 = ma > fabsf(tc); }
    else if (CLS == 1) { c1 = ma >= fabsf(sc); c2 = ma > fabsf(tc); }
    else { c1 = ma >= fabsf(sc); c2 = ma >= fabsf(tc); }
    unsigned long long inm = __builtin_amdgcn_ballot_w64(c1) & __builtin_amdgcn_ballot_w64(c2);
    if (inm == 0) return;                                           // none of the 64 (texel, sample) pairs is on this face
    bool in = c1 && c2;
    const float h = __builtin_amdgcn_rcpf(ma) * half_n;
    const float u = fmaf(sc, h, off), v = fmaf(tc, h, off);
    const int il = (int)u, jl = (int)v;
    if (SUB) {
        const bool c3 = u >= ulo, c4 = u < uhi, c5 = v >= vlo, c6 = v < vhi;
        inm &= __builtin_amdgcn_ballot_w64(c3) & __builtin_amdgcn_ballot_w64(c4) & __builtin_amdgcn_ballot_w64(c5) & __builtin_amdgcn_ballot_w64(c6);
        in = in && c3 && c4 && c5 && c6;
    }
    cnt += (unsigned)__builtin_popcountll(inm & vmask);             // padding rows of a list's last block do not count
    if (in) {
        const float a = __builtin_amdgcn_fractf(u), b = __builtin_amdgcn_fractf(v);
        unsigned t16, addr;
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(t16) : "v"(il), "s"(lds_base));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(jl), "s"(RS * 16), "v"(t16));
        lds_v4f_p tp = (lds_v4f_p)(unsigned long long)addr;
        v4f q00 = tp[0], q10 = tp[1], q01 = tp[RS], q11 = tp[RS + 1];
        asm("" : "+v"(q00)); asm("" : "+v"(q10)); asm("" : "+v"(q01)); asm("" : "+v"(q11));
        const float wa = wgt * a;
        const float w11 = wa * b;
        const float w10 = wa - w11;
        const float wt = wgt - wa;
        const float w01 = wt * b;
        const float w00 = wt - w01;
        ar = fmaf(w11, q11.x, fmaf(w01, q01.x, fmaf(w10, q10.x, fmaf(w00, q00.x, ar))));
        ag = fmaf(w11, q11.y, fmaf(w01, q01.y, fmaf(w10, q10.y, fmaf(w00, q00.y, ag))));
        ab = fmaf(w11, q11.z, fmaf(w01, q01.z, fmaf(w10, q10.z, fmaf(w00, q00.z, ab))));
    }
}

// all blocks of the list chunk [0, count) for this wave's 16 texels
template <int RS, bool SUB, int CLS>
__device__ __forceinline__ void mfma_chunk(lds_u16_p list, int count, const float* __restrict__ tabf, int comp, int row, int g,
                                           float bsc, float btc, float bma, float bw,
                                           unsigned lds_base, float half_n, float off, float ulo, float uhi, float vlo, float vhi,
                                           float& ar, float& ag, float& ab, unsigned& cnt) {
    const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int pos = 0; pos < count; pos += 16) {
        const int nvalid = min(16, count - pos);                    // wave-uniform
        const int idx = (int)list[pos + min(row, nvalid - 1)];
        float aval = tabf[idx * 4 + comp];
        unsigned long long vm0 = ~0ull, vm1 = ~0ull, vm2 = ~0ull, vm3 = ~0ull;
        if (nvalid < 16) {                                          // last block of the list: rows >= nvalid repeat the last sample with weight 0
            if (g == 3 && row >= nvalid) aval = 0.0f;
            vm0 = __builtin_amdgcn_ballot_w64(4 * g + 0 < nvalid); vm1 = __builtin_amdgcn_ballot_w64(4 * g + 1 < nvalid);
            vm2 = __builtin_amdgcn_ballot_w64(4 * g + 2 < nvalid); vm3 = __builtin_amdgcn_ballot_w64(4 * g + 3 < nvalid);
        }
        // D[row 4g + i][texel] in element i: (sc, tc, ma) = e.z R' + e.y T' + e.x B' in that order (k = 0, 1, 2), weight through k = 3
        const v4f dsc = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bsc, zero, 0, 0, 0);
        const v4f dtc = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, btc, zero, 0, 0, 0);
        const v4f dma = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bma, zero, 0, 0, 0);
        const v4f dwv = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bw, zero, 0, 0, 0);
        mfma_sample<RS, SUB, CLS>(dsc.x, dtc.x, dma.x, dwv.x, vm0, lds_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
        mfma_sample<RS, SUB, CLS>(dsc.y, dtc.y, dma.y, dwv.y, vm1, lds_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
        mfma_sample<RS, SUB, CLS>(dsc.z, dtc.z, dma.z, dwv.z, vm2, lds_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
        mfma_sample<RS, SUB, CLS>(dsc.w, dtc.w, dma.w, dwv.w, vm3, lds_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt);
    }
}

template <int RS, bool SUB>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mc_region_mfma(const RegArgs q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    float4* region = (float4*)smem_r;
    const unsigned lds_base = (unsigned)(unsigned long long)smem_r;
    unsigned* masks = (unsigned*)(smem_r + RS * RS * 16);
    unsigned* any = masks + q.NR * q.NW;
    unsigned* dmax = any + q.NR;
    unsigned* offs = dmax + 1;                                     // [NW + 1] exclusive prefix of the mask words' bit counts (current region)
    unsigned short* list = (unsigned short*)(offs + q.NW + 1);     // [LIST_CAP] sample indices of the current chunk
    const McArgs& p = q.a;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, row = lane & 15, g = lane >> 4;

    unsigned tile = blockIdx.x;                                    // plain order: see k_mc_region
    const int face = p.face0 + (int)(tile / (unsigned)p.tiles_per_face);
    const int tf = (int)(tile % (unsigned)p.tiles_per_face);
    const int ty = tf / p.tiles_x, tx = tf % p.tiles_x;
    const int x = tx * 16 + (wave & 3) * 4 + (row & 3);
    const int y = q.tile_y0 + ty * 16 + (wave >> 2) * 4 + (row >> 2);
    const bool valid = (x < p.size) && (y >= p.y0) && (y < p.y0 + p.rows);
    const int xc = min(x, p.size - 1), yc = min(y, p.size - 1);    // clamped to the LEVEL: a tile's frames do not depend on the dispatched rows

    const int n = p.n_src, nb = n + 2;
    const float nf = (float)n;
    const float half_n = 0.5f * nf;
    const float off = 0.5f * nf + 0.5f;
    ctab_t tab = (ctab_t)(unsigned long long)p.tab;
    const int NW = q.NW, NR = q.NR, G = q.G, RC = q.RC;

    f3 Fg;                                                          // this lane's column of the frame: group 0 R, 1 T, 2 B, 3 none (weight slot)
    {
        const f3 R = face_texel_dir(face, xc, yc, p.size);
        const f3 T = tangent_of(R);
        const f3 B = cross3(T, R);
        const f3 Rc = face_texel_dir(face, min(tx * 16 + 8, p.size - 1), min(q.tile_y0 + ty * 16 + 8, p.size - 1), p.size);
        const f3 Tc = tangent_of(Rc);
        const f3 Bc = cross3(Tc, Rc);
        for (int k = tid; k < NR * NW + NR + 1; k += 1024) masks[k] = 0u;
        __syncthreads();
        region_bin(masks, any, dmax, NR, NW, G, RC, n, R, T, B, Rc, Tc, Bc, tab, p.n_tab, tid);
        Fg = g == 0 ? R : (g == 1 ? T : (g == 2 ? B : mk3(0.0f, 0.0f, 0.0f)));
    }
    const float bw = g == 3 ? 1.0f : 0.0f;
    const int comp = g == 0 ? 2 : (g == 1 ? 1 : (g == 2 ? 0 : 3));    // table component of this lane's k: (z, y, x, w)
    const float* __restrict__ tabf = (const float*)p.tab;

    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    unsigned cnt = 0;
    for (int r = 0; r < NR; ++r) {
        __syncthreads();                                           // binning done / readers of the previous region and list done
        if (any[r] == 0u) continue;                                // workgroup-uniform
        const int f = r / (G * G);
        const int gy = (r / G) % G, gx = r % G;
        const int ox = gx * RC, oy = gy * RC;
        const int rcx = min(RC, n + 1 - ox), rcy = min(RC, n + 1 - oy);
        const float4* __restrict__ fsrc = p.src + ((size_t)f * nb + oy) * nb + ox;
        for (int k = tid; k < RS * RS; k += 1024) {
            const int ry = k / RS, rx = k - ry * RS;
            if (rx <= rcx && ry <= rcy) region[k] = fsrc[ry * nb + rx];
        }
        const unsigned* mw = masks + r * NW;
        if (wave == 0) {                                           // exclusive prefix of the words' bit counts: 4 words per lane + wave scan
            unsigned c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int w = lane * 4 + j; c[j] = w < NW ? (unsigned)__builtin_popcount(mw[w]) : 0u; sum += c[j]; }
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { unsigned v = (unsigned)__shfl_up((int)incl, o); if (lane >= o) incl += v; }
            unsigned run = incl - sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int w = lane * 4 + j; if (w < NW) offs[w] = run; run += c[j]; }
            if (lane == 63) offs[NW] = incl;
        }
        __syncthreads();
        const int total = (int)offs[NW];
        float bsc, btc, bma;
        face_coords(f, Fg.x, Fg.y, Fg.z, bsc, btc, bma);
        const unsigned pass_base = lds_base - (unsigned)(oy * RS + ox) * 16u;
        const float ulo = (float)ox, uhi = (float)(ox + rcx), vlo = (float)oy, vhi = (float)(oy + rcy);
        for (int c0 = 0; c0 < total; c0 += LIST_CAP) {
            if (c0 > 0) __syncthreads();                           // readers of the previous chunk are done
            if (tid < NW) {                                        // thread t writes the set bits of word t that fall into this chunk
                unsigned m = mw[tid];
                int pidx = (int)offs[tid] - c0;
                while (m) {
                    const int bpos = __builtin_ctz(m);
                    m &= m - 1u;
                    if ((unsigned)pidx < (unsigned)LIST_CAP) list[pidx] = (unsigned short)((tid << 5) + bpos);
                    ++pidx;
                }
            }
            __syncthreads();
            const int count = min(LIST_CAP, total - c0);
            lds_u16_p lp = (lds_u16_p)(unsigned long long)(unsigned)(unsigned long long)list;
            switch (f >> 1) {
            case 0: mfma_chunk<RS, SUB, 0>(lp, count, tabf, comp, row, g, bsc, btc, bma, bw, pass_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt); break;
            case 1: mfma_chunk<RS, SUB, 1>(lp, count, tabf, comp, row, g, bsc, btc, bma, bw, pass_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt); break;
            default: mfma_chunk<RS, SUB, 2>(lp, count, tabf, comp, row, g, bsc, btc, bma, bw, pass_base, half_n, off, ulo, uhi, vlo, vhi, ar, ag, ab, cnt); break;
            }
        }
    }

    // completeness: every (texel, sample) pair of this wave exactly once; else the wave recomputes its 16 texels with direct loads
    const bool healed = cnt != 16u * (unsigned)p.n_tab;
    if (healed) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
        const f3 R = face_texel_dir(face, xc, yc, p.size);
        const f3 T = tangent_of(R);
        const f3 B = cross3(T, R);
        ar = 0.0f; ag = 0.0f; ab = 0.0f;
        for (int i = g; i < p.n_tab; i += 4) {
            const float4 e = p.tab[i];
            f3 L;
            L.x = fmaf(e.x, B.x, fmaf(e.y, T.x, e.z * R.x));
            L.y = fmaf(e.x, B.y, fmaf(e.y, T.y, e.z * R.y));
            L.z = fmaf(e.x, B.z, fmaf(e.y, T.z, e.z * R.z));
            f3 c = sample_bordered<false>(rs, L, nf, off, nb, nb * 16);
            ar = fmaf(e.w, c.x, ar); ag = fmaf(e.w, c.y, ag); ab = fmaf(e.w, c.z, ab);
        }
    }
    if (q.stats && lane == 0) {
        if (healed) atomicAdd(&q.stats[0], 1ull);
        atomicAdd(&q.stats[1], 1ull);
    }
    // fold the four lane groups: (g0 + g1) + (g2 + g3)
    ar += __shfl_xor(ar, 16); ag += __shfl_xor(ag, 16); ab += __shfl_xor(ab, 16);
    ar += __shfl_xor(ar, 32); ag += __shfl_xor(ag, 32); ab += __shfl_xor(ab, 32);
    if (valid && g == 0) {
        float4 o;
        o.x = ar / p.divisor; o.y = ag / p.divisor; o.z = ab / p.divisor; o.w = p.alpha;
        p.out[((size_t)face * p.size + y) * p.size + x] = o;
    }
}

static unsigned long long* g_reg_stats = nullptr;      // device counters, enabled by PBR_MC_STATS=1

extern "C" int pbrk_mc_region_stats(unsigned long long* out2, int reset) {
    if (!g_reg_stats || !out2) return PBRK_E_ARG;
    if (hipMemcpy(out2, g_reg_stats, 16, hipMemcpyDeviceToHost) != hipSuccess) return PBRK_E_LAUNCH;
    if (reset && hipMemset(g_reg_stats, 0, 64) != hipSuccess) return PBRK_E_LAUNCH;
    return PBRK_OK;
}
extern "C" int pbrk_mc_region_window_stats(unsigned long long* out1) {      // samples run through the test-free body (sum over tiles)
    if (!g_reg_stats || !out1) return PBRK_E_ARG;
    return hipMemcpy(out1, g_reg_stats + 5, 8, hipMemcpyDeviceToHost) == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}
extern "C" int pbrk_mc_region_flag_stats(unsigned long long* out3) {
    if (!g_reg_stats || !out3) return PBRK_E_ARG;
    return hipMemcpy(out3, g_reg_stats + 2, 24, hipMemcpyDeviceToHost) == hipSuccess ? PBRK_OK : PBRK_E_LAUNCH;
}

template <int RS, bool SUB, int TILE>
static void launch_region_t(const RegArgs& q, unsigned grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_mc_region<RS, SUB, TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL((k_mc_region<RS, SUB, TILE>), dim3(grid), dim3(1024), lds, st, q);
}

template <int RS, bool SUB>
static void launch_mfma_t(const RegArgs& q, unsigned grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_mc_region_mfma<RS, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL((k_mc_region_mfma<RS, SUB>), dim3(grid), dim3(1024), lds, st, q);
}

// Which kernel serves a level (tests / A-B runs; -1 = the environment variable PBR_MC_REGION / PBR_MC_LDS, else the default 1)
int g_mc_region_mode = -1, g_mc_lds_mode = -1;
extern "C" void pbrk_mc_set_kernels(int region, int lds) { g_mc_region_mode = region; g_mc_lds_mode = lds; }

bool launch_mc_region(McArgs a, int nfaces, hipStream_t st) {
    static int stats_on = -1;
    if (g_mc_region_mode < 0) { const char* e = getenv("PBR_MC_REGION"); g_mc_region_mode = e ? atoi(e) : 1; }
    if (!g_mc_region_mode) return false;
    // shape conditions (level only): source too big for LDS as a whole, enough 16x16 tiles to fill the chip twice over
    if (a.n_src < 16 || a.size < 256 || a.n_tab < 1 || a.n_tab > 8192) return false;
    RegArgs q;
    q.a = a;
    int RS;
    if (a.n_src <= 16) { RS = 18; q.G = 1; q.RC = a.n_src + 1; }
    else if (a.n_src <= 32) { RS = 34; q.G = 1; q.RC = a.n_src + 1; }
    else if (a.n_src <= 64) { RS = 66; q.G = 1; q.RC = a.n_src + 1; }
    else { RS = 66; q.RC = 65; q.G = (a.n_src + 1 + 64) / 65; }
    q.NR = 6 * q.G * q.G;
    q.NW = (a.n_tab + 31) / 32;
    size_t lds = (size_t)RS * RS * 16 + ((size_t)q.NR * q.NW + q.NR + 4 + (q.G > 1 ? q.NW : 0)) * 4;      // region, flags, any[], dmax, (quarter-face levels) proved-sample flags
    if (lds < (size_t)1024 * 3 * 4) lds = (size_t)1024 * 3 * 4;      // the slices' partial sums (REG_S * REG_TX = 1024 texel-slices)
    if (lds > 80 * 1024) return false;                             // two workgroups per CU or not at all
    if (stats_on < 0) {
        const char* e = getenv("PBR_MC_STATS"); stats_on = e ? atoi(e) : 0;
        if (stats_on) { if (hipMalloc(&g_reg_stats, 64) != hipSuccess) g_reg_stats = nullptr; else (void)hipMemset(g_reg_stats, 0, 64); }
    }
    q.stats = g_reg_stats;
    // Tile size: 16 x 16 output texels x 4 slices of the sample table everywhere.  Measured and dropped: 8 x 8 tiles x 16
    // slices (halve the frame spread, pay four times the per-tile work: C4 mip 2 44.0 -> 53.5 ms), 32 x 16 and 32 x 32 tiles (no gain
    // on the big levels, losses on the small ones).
    static int mfma_mode = -1;
    if (mfma_mode < 0) { const char* e = getenv("PBR_MC_MFMA"); mfma_mode = e ? atoi(e) : 0; }      // opt-in: measured slower (see above)
    if (mfma_mode && q.NW <= 256) {
        size_t lds_m = (size_t)RS * RS * 16 + ((size_t)q.NR * q.NW + q.NR + 1 + q.NW + 1) * 4 + (size_t)LIST_CAP * 2;
        lds_m = (lds_m + 15) & ~(size_t)15;
        if (lds_m <= 80 * 1024) {
            q.tile_y0 = (a.y0 / 16) * 16;
            q.a.tiles_x = (a.size + 15) / 16;
            q.a.tiles_per_face = q.a.tiles_x * ((a.y0 + a.rows - q.tile_y0 + 15) / 16);
            unsigned grid = (unsigned)(q.a.tiles_per_face * nfaces);
            if (RS == 18) launch_mfma_t<18, false>(q, grid, lds_m, st);
            else if (RS == 34) launch_mfma_t<34, false>(q, grid, lds_m, st);
            else if (q.G == 1) launch_mfma_t<66, false>(q, grid, lds_m, st);
            else launch_mfma_t<66, true>(q, grid, lds_m, st);
            return true;
        }
    }
    const int tile = 16, nslices = 4;
    for (int s = 0; s < REG_MAX_S; ++s) q.expect[s] = 0;
    for (int w = 0; w < q.NW; ++w) { int c = a.n_tab - w * 32; q.expect[w % nslices] += c > 32 ? 32 : c; }
    q.a.tiles_x = (a.size + tile - 1) / tile;
    const int tiles_y = (a.rows + tile - 1) / tile;
    q.a.tiles_per_face = q.a.tiles_x * tiles_y;                            // tiles start at the dispatch's first row, whatever it is
    {   // output row of the tangent frame's pole on faces +X (t = (1 - vy/vx)/2) and -X (t = (1 + vy/vx)/2), v = the vector of tangent_of
        const double vy_vx = 6.11831989512 / 12.123825810901;
        for (int f = 0; f < 2; ++f) {
            int row = (int)((f == 0 ? 0.5 * (1.0 - vy_vx) : 0.5 * (1.0 + vy_vx)) * a.size);
            int tr = (row - a.y0) / tile;
            if (row < a.y0) tr = 0;
            q.pole_row[f] = tr < 0 ? 0 : (tr > tiles_y - 1 ? tiles_y - 1 : tr);
        }
    }
    const unsigned grid = (unsigned)(q.a.tiles_per_face * nfaces);
    if (RS == 18) launch_region_t<18, false, 16>(q, grid, lds, st);
    else if (RS == 34) launch_region_t<34, false, 16>(q, grid, lds, st);
    else if (q.G == 1) launch_region_t<66, false, 16>(q, grid, lds, st);
    else launch_region_t<66, true, 16>(q, grid, lds, st);
    return true;
}
