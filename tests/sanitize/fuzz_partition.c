// sanitizer harness for the multi-GPU host logic (run on CPU only): PBR_PartitionIBL / PBR_SelectUnits / PBR_UnitByteRange /
// PBR_GatherPlan over many (sizes, world) combinations -- every texel of every level is owned by exactly one rank, the byte ranges
// of a rank's units lie inside their texture and do not overlap, the two phases of the overlapped exchange are a partition of the
// one-shot plan, and root's receives mirror the peers' sends.  The texture queries are answered from the tight [mip][face][y][x] layout.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "pbr_host.h"

static uint32_t rs = 2463534242u;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }

/* ---- the few backend queries the host logic makes, on plain structs ---- */
static uint64_t mip_bytes(const GPU_Texture* t, uint32_t m) { uint64_t w = t->width >> m, h = t->height >> m; if (!w) w = 1; if (!h) h = 1; return w * h * 16u * t->layer_count; }
uint64_t GPUX_TextureMipBytes(const GPU_Texture* t, uint32_t m) { return mip_bytes(t, m); }
uint64_t GPUX_TextureMipOffset(const GPU_Texture* t, uint32_t m) { uint64_t o = 0; for (uint32_t l = 0; l < m; ++l) o += mip_bytes(t, l); return o; }
uint64_t GPUX_TextureTotalBytes(const GPU_Texture* t) { return GPUX_TextureMipOffset(t, t->mip_level_count); }
void* GPUX_TextureDevicePtr(GPU_Texture* t, uint32_t m) { (void)m; return (void*)(uintptr_t)(t->layer_count == 6 && t->mip_level_count > 1 ? 0x10000000u : 0x70000000u); }
void GPUX_InvalidateTexture(GPU_Texture* t) { (void)t; }
#define CAP 4096
static int cmp_range(const void* a, const void* b) {
    const PBR_XferRange* x = (const PBR_XferRange*)a; const PBR_XferRange* y = (const PBR_XferRange*)b;
    if (x->ptr != y->ptr) return (uintptr_t)x->ptr < (uintptr_t)y->ptr ? -1 : 1;
    return x->peer - y->peer;
}

int main(void) {
    static PBR_WorkUnit units[CAP], sel[CAP];
    static PBR_XferRange a[CAP], e[CAP], l[CAP], root[CAP], peers[CAP];
    long cases = 0;
    for (int iter = 0; iter < 1500; ++iter) {
        const uint32_t spec = 16u << (rnd() % 9), irr = (rnd() % 4) ? (8u << (rnd() % 5)) : 0u;      /* 16 .. 4096; 0 or 8 .. 128 */
        uint32_t min_size = 1u << (rnd() % 6); if (min_size > spec) min_size = spec;
        const int world = 1 + (int)(rnd() % 9);
        const uint32_t env = spec / 2 ? spec / 2 : 1;
        uint32_t mips = 0; for (uint32_t s = spec; s; s >>= 1) ++mips;
        GPU_Texture st; memset(&st, 0, sizeof st); st.width = st.height = spec; st.depth = 1; st.layer_count = 6; st.mip_level_count = mips;
        GPU_Texture it; memset(&it, 0, sizeof it); it.width = it.height = irr ? irr : 1; it.depth = 1; it.layer_count = 6; it.mip_level_count = 1;
        PBR_IBLMaps maps; memset(&maps, 0, sizeof maps); maps.tex_specular_env_map = &st; maps.irradiance_map = irr ? &it : NULL;
        /* ownership: one counter per (level, face, row) */
        static uint8_t own[14][6][4096]; static uint8_t iown[6][128];
        memset(own, 0, sizeof own); memset(iown, 0, sizeof iown);
        for (int r = 0; r < world; ++r) {
            const uint32_t n = PBR_PartitionIBL(spec, min_size, irr, env, world, r, units, CAP);
            if (n > CAP) return 2;
            for (uint32_t k = 0; k < n; ++k) {
                const PBR_WorkUnit* u = &units[k];
                GPU_Texture* t; uint64_t off, bytes;
                if (PBR_UnitByteRange(&maps, u, &t, &off, &bytes) != PBR_OK) { fprintf(stderr, "byte range rejected a partition unit\n"); return 3; }
                if (off + bytes > GPUX_TextureTotalBytes(t)) { fprintf(stderr, "unit outside its texture\n"); return 4; }
                for (uint32_t f = u->face0; f < u->face1; ++f) for (uint32_t y = u->row0; y < u->row1; ++y) {
                    if (u->kind == PBR_Unit_Irradiance) iown[f][y]++; else own[u->mip][f][y]++;
                }
            }
            const uint32_t ne = PBR_SelectUnits(units, n, 0x2u, sel), nl = PBR_SelectUnits(units, n, ~0x2u, sel + ne);
            if (ne + nl != n) { fprintf(stderr, "phases do not add up\n"); return 5; }
        }
        for (uint32_t m = 0; m < mips; ++m) {
            const uint32_t size = spec >> m;
            if (size < min_size) break;
            for (int f = 0; f < 6; ++f) for (uint32_t y = 0; y < size; ++y) if (own[m][f][y] != 1) { fprintf(stderr, "spec %u world %d: mip %u face %d row %u owned %d times\n", spec, world, m, f, y, own[m][f][y]); return 6; }
        }
        if (irr) for (int f = 0; f < 6; ++f) for (uint32_t y = 0; y < irr; ++y) if (iown[f][y] != 1) { fprintf(stderr, "irradiance row owned %d times\n", iown[f][y]); return 7; }
        /* exchange plans */
        if (world > 1) {
            for (uint32_t mask_i = 0; mask_i < 3; ++mask_i) {
                const uint32_t mask = mask_i == 0 ? 0xFFFFFFFFu : (mask_i == 1 ? 0x2u : ~0x2u);
                const int64_t nr = PBR_GatherPlan(0, world, 0, &maps, min_size, env, mask, root, CAP);
                if (nr < 0) return 8;
                int64_t np = 0;
                for (int r = 1; r < world; ++r) {
                    const int64_t k = PBR_GatherPlan(0, world, r, &maps, min_size, env, mask, peers + np, (uint32_t)(CAP - np));
                    if (k < 0) return 9;
                    for (int64_t i = 0; i < k; ++i) { if (peers[np + i].peer != 0) return 10; peers[np + i].peer = r; }
                    np += k;
                }
                if (np != nr) { fprintf(stderr, "root expects %lld transfers, peers send %lld\n", (long long)nr, (long long)np); return 11; }
                qsort(root, (size_t)nr, sizeof root[0], cmp_range); qsort(peers, (size_t)np, sizeof peers[0], cmp_range);
                for (int64_t i = 0; i < nr; ++i) {
                    if (root[i].ptr != peers[i].ptr || root[i].bytes != peers[i].bytes || root[i].peer != peers[i].peer) { fprintf(stderr, "root / peer plans differ\n"); return 12; }
                    if (i && root[i - 1].ptr == root[i].ptr) { fprintf(stderr, "two transfers to one address\n"); return 13; }
                    if (i && (uintptr_t)root[i - 1].ptr + root[i - 1].bytes > (uintptr_t)root[i].ptr && ((uintptr_t)root[i].ptr >> 28) == ((uintptr_t)root[i - 1].ptr >> 28)) { fprintf(stderr, "overlapping receives\n"); return 14; }
                }
            }
            const int r = 1 + (int)(rnd() % (uint32_t)(world - 1));
            const int64_t na = PBR_GatherPlan(0, world, r, &maps, min_size, env, 0xFFFFFFFFu, a, CAP);
            const int64_t ne = PBR_GatherPlan(0, world, r, &maps, min_size, env, 0x2u, e, CAP), nl = PBR_GatherPlan(0, world, r, &maps, min_size, env, ~0x2u, l, CAP);
            if (na < 0 || ne < 0 || nl < 0 || ne + nl != na) { fprintf(stderr, "phase plans do not add up\n"); return 15; }
        }
        /* bad arguments are rejected, not dereferenced */
        if (PBR_GatherPlan(0, world, world, &maps, min_size, env, 1, a, CAP) != PBR_E_BADARG) return 16;
        if (PBR_GatherPlan(world, world, 0, &maps, min_size, env, 1, a, CAP) != PBR_E_BADARG) return 17;
        if (world > 1 && PBR_GatherUnits(NULL, NULL, 0, world, 0, &maps, min_size, env) != PBR_E_BADARG) return 18;
        ++cases;
    }
    printf("ok: %ld partition / exchange-plan cases\n", cases);
    return 0;
}
