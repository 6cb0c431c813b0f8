/*
 * pbr_gather.c -- the one exchange step of the multi-GPU job, in the C host layer: RCCL grouped
 * point-to-point transfers of the finished work units (precompute) or screen bands (shade pass) to the root rank.
 *
 * The reference has no collectives at all (one process, one Vulkan queue: src/gpu/gpu_vulkan.c:1040-1106); this is the
 * new host component its multi-GPU form needs.  The communicator is the caller's: an `ncclComm_t` passed as an opaque
 * pointer, created by whatever bootstrap the application has (MPI, torch.distributed, a file) -- the library never
 * initialises or destroys one.  Every transfer is a contiguous byte range: a unit is rows [row0,row1) of one face, or a
 * run of whole faces, of one level in the [mip][face][y][x] layout (PBR_UnitByteRange), a band is rows of a 2-D frame.
 * All ranks derive the same unit lists from PBR_PartitionIBL, so nothing but pixels crosses xGMI.
 *
 * RCCL is bound at FIRST USE, not at link time (round 3): a single-GPU consumer of libgpu_hip.so neither loads nor needs
 * librccl, and a multi-GPU one gets the copy of librccl its own process already holds -- the one that created the
 * communicator it passes in (an ncclComm_t is only meaningful to the library copy that made it).  Search order:
 *   1. PBR_SetRcclLibrary(path) / environment PBR_RCCL_LIB           (explicit; also how the CPU tests plug in their stub)
 *   2. dlopen("librccl.so.1", RTLD_NOLOAD)                           (already mapped: torch's bundled copy, or ROCm's)
 *   3. the process's global scope (an application linked against RCCL or exporting the entry points itself)
 *   4. dlopen("librccl.so.1"), then "librccl.so", then "/opt/rocm/lib/librccl.so.1"
 */
#define _GNU_SOURCE 1
#include "pbr_host.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the five entry points of rccl.h that the exchange uses (+ three queries), by their C ABI: ncclResult_t is an int
 * (ncclSuccess = 0), ncclDataType_t an int (ncclInt8 = 0), ncclComm_t and hipStream_t are pointers */
typedef struct PBR_Rccl {
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void* buf, size_t count, int datatype, int peer, void* comm, void* stream);
    int (*Recv)(void* buf, size_t count, int datatype, int peer, void* comm, void* stream);
    const char* (*GetErrorString)(int);
    int (*GetVersion)(int*);                  /* optional */
    int (*CommCount)(void* comm, int*);       /* optional */
    int (*CommUserRank)(void* comm, int*);    /* optional */
    void* handle;
    char  path[512];
    int   resolved;                           /* 0 = not tried, 1 = bound, -1 = failed */
} PBR_Rccl;
static PBR_Rccl g_rccl;
static char g_rccl_override[512];
enum { NCCL_INT8 = 0 };

static int rccl_bind_from(void* h) {
    PBR_Rccl* R = &g_rccl;
    *(void**)&R->GroupStart = dlsym(h, "ncclGroupStart");
    *(void**)&R->GroupEnd = dlsym(h, "ncclGroupEnd");
    *(void**)&R->Send = dlsym(h, "ncclSend");
    *(void**)&R->Recv = dlsym(h, "ncclRecv");
    *(void**)&R->GetErrorString = dlsym(h, "ncclGetErrorString");
    *(void**)&R->GetVersion = dlsym(h, "ncclGetVersion");
    *(void**)&R->CommCount = dlsym(h, "ncclCommCount");
    *(void**)&R->CommUserRank = dlsym(h, "ncclCommUserRank");
    if (!R->GroupStart || !R->GroupEnd || !R->Send || !R->Recv || !R->GetErrorString) return 0;
    Dl_info info;
    R->path[0] = 0;
    if (dladdr(*(void**)&R->Send, &info) && info.dli_fname) snprintf(R->path, sizeof R->path, "%s", info.dli_fname);
    return 1;
}

static int rccl_resolve(void) {
    PBR_Rccl* R = &g_rccl;
    if (R->resolved) return R->resolved > 0;
    const char* explicit_path = g_rccl_override[0] ? g_rccl_override : getenv("PBR_RCCL_LIB");
    void* h = NULL;
    if (explicit_path && explicit_path[0]) {
        h = dlopen(explicit_path, RTLD_NOW | RTLD_LOCAL);
        if (!h) fprintf(stderr, "GPU-ERROR: RCCL library \"%s\" cannot be loaded: %s\n", explicit_path, dlerror());
        else if (!rccl_bind_from(h)) { fprintf(stderr, "GPU-ERROR: \"%s\" lacks ncclGroupStart/End, ncclSend/Recv\n", explicit_path); dlclose(h); h = NULL; }
        R->handle = h; R->resolved = h ? 1 : -1;
        return h != NULL;
    }
    if ((h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD)) && rccl_bind_from(h)) { R->handle = h; R->resolved = 1; return 1; }
    if (rccl_bind_from(RTLD_DEFAULT)) { R->handle = NULL; R->resolved = 1; return 1; }
    static const char* const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (unsigned i = 0; i < sizeof names / sizeof names[0]; ++i)
        if ((h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL)) && rccl_bind_from(h)) { R->handle = h; R->resolved = 1; return 1; }
    fprintf(stderr, "GPU-ERROR: the multi-GPU exchange needs librccl.so.1 and none could be loaded (%s); set PBR_RCCL_LIB\n", dlerror());
    R->resolved = -1;
    return 0;
}

int PBR_SetRcclLibrary(const char* path) {
    if (g_rccl.resolved > 0 && g_rccl.handle) dlclose(g_rccl.handle);
    memset(&g_rccl, 0, sizeof g_rccl);
    g_rccl_override[0] = 0;
    if (path && path[0]) {
        if (strlen(path) >= sizeof g_rccl_override) return PBR_E_BADARG;
        strcpy(g_rccl_override, path);
        return rccl_resolve() ? PBR_OK : PBR_E_COMM;
    }
    return PBR_OK;
}

int PBR_RcclInfo(int* version, const char** path) {
    if (!rccl_resolve()) return PBR_E_COMM;
    if (version) { *version = 0; if (g_rccl.GetVersion) g_rccl.GetVersion(version); }
    if (path) *path = g_rccl.path;
    return PBR_OK;
}

int PBR_CommInfo(void* nccl_comm, int* count, int* user_rank) {
    if (!nccl_comm) return PBR_E_BADARG;
    if (!rccl_resolve()) return PBR_E_COMM;
    if (count) { *count = -1; if (g_rccl.CommCount && g_rccl.CommCount(nccl_comm, count) != 0) return PBR_E_COMM; }
    if (user_rank) { *user_rank = -1; if (g_rccl.CommUserRank && g_rccl.CommUserRank(nccl_comm, user_rank) != 0) return PBR_E_COMM; }
    return PBR_OK;
}

int PBR_UnitByteRange(const PBR_IBLMaps* maps, const PBR_WorkUnit* u, GPU_Texture** tex, uint64_t* offset, uint64_t* bytes) {
    if (!maps || !u || !tex || !offset || !bytes) return PBR_E_BADARG;
    GPU_Texture* t; uint32_t mip = 0;
    if (u->kind == PBR_Unit_Irradiance) t = maps->irradiance_map;
    else if (u->kind == PBR_Unit_Prefilter) { t = maps->tex_specular_env_map; mip = u->mip; }
    else if (u->kind == PBR_Unit_BrdfLut) t = maps->brdf_lut;
    else return PBR_E_BADARG;
    if (!t || mip >= t->mip_level_count) return PBR_E_BADARG;
    uint64_t w = t->width >> mip; if (w < 1) w = 1;
    uint64_t h = t->height >> mip; if (h < 1) h = 1;
    uint64_t layer_bytes = GPUX_TextureMipBytes(t, mip) / t->layer_count;
    uint64_t row_bytes = layer_bytes / h;
    if (u->face0 >= u->face1 || u->face1 > t->layer_count || u->row0 >= u->row1 || u->row1 > h) return PBR_E_BADARG;
    /* contiguous only as rows of ONE face or as whole faces */
    if (u->face1 != u->face0 + 1 && !(u->row0 == 0 && u->row1 == h)) return PBR_E_BADARG;
    (void)w;
    *tex = t;
    *offset = GPUX_TextureMipOffset(t, mip) + u->face0 * layer_bytes + u->row0 * row_bytes;
    *bytes = (u->face1 - 1 - u->face0) * layer_bytes + (u->row1 - u->row0) * row_bytes;
    return PBR_OK;
}

int PBR_ExchangeRanges(void* nccl_comm, void* stream, const PBR_XferRange* sends, uint32_t n_sends,
                       const PBR_XferRange* recvs, uint32_t n_recvs) {
    if (!nccl_comm || (n_sends && !sends) || (n_recvs && !recvs)) return PBR_E_BADARG;
    for (uint32_t i = 0; i < n_sends; ++i) if (!sends[i].ptr || !sends[i].bytes || sends[i].peer < 0) return PBR_E_BADARG;
    for (uint32_t i = 0; i < n_recvs; ++i) if (!recvs[i].ptr || !recvs[i].bytes || recvs[i].peer < 0) return PBR_E_BADARG;
    if (n_sends + n_recvs == 0) return PBR_OK;
    if (!rccl_resolve()) return PBR_E_COMM;
    const PBR_Rccl* R = &g_rccl;
    /* one group: RCCL schedules every transfer of the step together (all peers' links at once).  A call that fails inside
     * the group must not leave the group open -- every later RCCL call of this thread would be queued into it and hang --
     * so the error is remembered, the group is closed (its own result no longer matters) and PBR_E_COMM returned. */
    int r = R->GroupStart();
    if (r != 0) { fprintf(stderr, "GPU-ERROR: ncclGroupStart failed: %s\n", R->GetErrorString(r)); return PBR_E_COMM; }
    int failed = 0; const char* what = "";
    for (uint32_t i = 0; i < n_recvs && !failed; ++i)
        if ((r = R->Recv(recvs[i].ptr, recvs[i].bytes, NCCL_INT8, recvs[i].peer, nccl_comm, stream)) != 0) { failed = r; what = "ncclRecv"; }
    for (uint32_t i = 0; i < n_sends && !failed; ++i)
        if ((r = R->Send(sends[i].ptr, sends[i].bytes, NCCL_INT8, sends[i].peer, nccl_comm, stream)) != 0) { failed = r; what = "ncclSend"; }
    r = R->GroupEnd();
    if (failed) { fprintf(stderr, "GPU-ERROR: %s failed: %s (group closed; treat the communicator as dead)\n", what, R->GetErrorString(failed)); return PBR_E_COMM; }
    if (r != 0) { fprintf(stderr, "GPU-ERROR: ncclGroupEnd failed: %s\n", R->GetErrorString(r)); return PBR_E_COMM; }
    return PBR_OK;
}

/* unit lists are small (tens of units per rank) */
#define MAX_UNITS 512

/* bit of a unit in a level mask: prefilter mip l = bit l, irradiance = PBR_LEVEL_IRRADIANCE */
static uint32_t unit_level_bit(const PBR_WorkUnit* u) {
    if (u->kind == PBR_Unit_Irradiance) return PBR_LEVEL_IRRADIANCE;
    if (u->kind == PBR_Unit_Prefilter) return u->mip < 31 ? (1u << u->mip) : 0u;
    return 0u;
}

uint32_t PBR_SelectUnits(const PBR_WorkUnit* units, uint32_t n, uint32_t level_mask, PBR_WorkUnit* out) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i) if (unit_level_bit(&units[i]) & level_mask) out[k++] = units[i];
    return k;
}

int64_t PBR_GatherPlan(int root, int world, int rank, const PBR_IBLMaps* maps, uint32_t min_size, uint32_t env_size,
                       uint32_t level_mask, PBR_XferRange* out, uint32_t capacity) {
    if (!maps || !maps->tex_specular_env_map || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return PBR_E_BADARG;
    if (world == 1) return 0;
    const uint32_t spec = maps->tex_specular_env_map->width;
    const uint32_t irr = maps->irradiance_map ? maps->irradiance_map->width : 0;
    PBR_WorkUnit* units = (PBR_WorkUnit*)malloc(sizeof(PBR_WorkUnit) * MAX_UNITS);
    if (!units) return PBR_E_BADARG;
    uint32_t n_xf = 0;
    int rc = PBR_OK;
    for (int r = 0; r < world && rc == PBR_OK; ++r) {
        if (r == root || (rank != root && r != rank)) continue;           /* the root's own units are already in place */
        uint32_t n = PBR_PartitionIBL(spec, min_size, irr, env_size, world, r, units, MAX_UNITS);
        for (uint32_t k = 0; k < n; ++k) {
            if (!(unit_level_bit(&units[k]) & level_mask)) continue;      /* other phase; the BRDF LUT is computed redundantly on every rank */
            GPU_Texture* t; uint64_t off, bytes;
            rc = PBR_UnitByteRange(maps, &units[k], &t, &off, &bytes);
            if (rc != PBR_OK) break;
            if (out) {
                if (n_xf >= capacity) { rc = PBR_E_BADARG; break; }
                out[n_xf].ptr = (char*)GPUX_TextureDevicePtr(t, 0) + off;
                out[n_xf].bytes = bytes;
                out[n_xf].peer = rank == root ? r : root;
            }
            ++n_xf;
        }
    }
    free(units);
    return rc == PBR_OK ? (int64_t)n_xf : rc;
}

int64_t PBR_GatherUnitsMasked(void* nccl_comm, void* stream, int root, int world, int rank, const PBR_IBLMaps* maps,
                              uint32_t min_size, uint32_t env_size, uint32_t level_mask) {
    if (!maps || !maps->tex_specular_env_map || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return PBR_E_BADARG;
    if (world == 1) return 0;
    if (!nccl_comm) return PBR_E_BADARG;
    const uint32_t cap = MAX_UNITS * (uint32_t)(rank == root ? world : 1);
    PBR_XferRange* xf = (PBR_XferRange*)malloc(sizeof(PBR_XferRange) * (size_t)cap);
    if (!xf) return PBR_E_BADARG;
    int64_t n = PBR_GatherPlan(root, world, rank, maps, min_size, env_size, level_mask, xf, cap);
    int64_t total = 0;
    int rc = n < 0 ? (int)n : PBR_OK;
    if (rc == PBR_OK) {
        for (int64_t i = 0; i < n; ++i) total += (int64_t)xf[i].bytes;
        rc = rank == root ? PBR_ExchangeRanges(nccl_comm, stream, NULL, 0, xf, (uint32_t)n) : PBR_ExchangeRanges(nccl_comm, stream, xf, (uint32_t)n, NULL, 0);
    }
    if (rc == PBR_OK && rank == root && n > 0) {
        /* the received bytes bypass the backend's op recording: drop the sampler twins built from the old contents */
        GPUX_InvalidateTexture(maps->tex_specular_env_map);
        if (maps->irradiance_map) GPUX_InvalidateTexture(maps->irradiance_map);
    }
    free(xf);
    return rc == PBR_OK ? total : rc;
}

int64_t PBR_GatherUnits(void* nccl_comm, void* stream, int root, int world, int rank, const PBR_IBLMaps* maps,
                        uint32_t min_size, uint32_t env_size) {
    return PBR_GatherUnitsMasked(nccl_comm, stream, root, world, rank, maps, min_size, env_size, 0xFFFFFFFFu);
}

/* The whole partitioned job of one rank with the exchange overlapped (see pbr_host.h).  Two graphs, two exchanges:
 *   g_early: [whatever the caller recorded: the source's mip chain] + the units of `early_mask`   -> submit
 *   g_late : the remaining units                                                                 -> submit (ordered after g_early's kernels)
 *   exchange of the early units on g_early's stream: starts when g_early's kernels are done, runs beside g_late's kernels
 *   exchange of the late units on g_late's stream
 * Root receives into ranges no kernel of its own writes (units are disjoint), so the transfers need no further fencing. */
int64_t PBR_RunPartitionedIBL(PBR_IBLPipelines* p, GPU_Graph* g_early, GPU_Graph* g_late, GPU_DescriptorArena* arena,
                              GPU_Texture* tex_env_cube, const PBR_IBLMaps* maps, void* nccl_comm, int root, int world, int rank,
                              uint32_t min_size, uint32_t early_mask) {
    if (!p || !g_early || !g_late || g_early == g_late || !arena || !tex_env_cube || !maps || !maps->tex_specular_env_map ||
        world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || (world > 1 && !nccl_comm)) return PBR_E_BADARG;
    const uint32_t spec = maps->tex_specular_env_map->width;
    const uint32_t irr = maps->irradiance_map ? maps->irradiance_map->width : 0;
    const uint32_t env_size = tex_env_cube->width;
    PBR_WorkUnit* units = (PBR_WorkUnit*)malloc(sizeof(PBR_WorkUnit) * MAX_UNITS * 2);
    if (!units) return PBR_E_BADARG;
    PBR_WorkUnit* part = units + MAX_UNITS;
    uint32_t n = PBR_PartitionIBL(spec, min_size, irr, env_size, world, rank, units, MAX_UNITS);
    uint32_t ne = PBR_SelectUnits(units, n, early_mask, part);
    if (ne) PBR_RecordUnits(p, g_early, arena, tex_env_cube, maps, part, ne);
    GPU_GraphSubmit(g_early);
    uint32_t nl = PBR_SelectUnits(units, n, ~early_mask, part);
    if (nl) PBR_RecordUnits(p, g_late, arena, tex_env_cube, maps, part, nl);
    GPU_GraphSubmit(g_late);
    free(units);
    /* both graphs are in flight from here on: whatever is returned, the caller must GPU_GraphWait both (pbr_host.h) */
    int64_t b0 = PBR_GatherUnitsMasked(nccl_comm, GPUX_GraphStream(g_early), root, world, rank, maps, min_size, env_size, early_mask);
    if (b0 < 0) return b0;                                   /* the late exchange is not issued: the communicator is dead */
    int64_t b1 = PBR_GatherUnitsMasked(nccl_comm, GPUX_GraphStream(g_late), root, world, rank, maps, min_size, env_size, ~early_mask);
    if (b1 < 0) return b1;
    return b0 + b1;
}

void PBR_BandRows(uint32_t height, int world, int rank, uint32_t* row0, uint32_t* row1) {
    *row0 = (uint32_t)((uint64_t)height * (uint64_t)rank / (uint64_t)world);
    *row1 = (uint32_t)((uint64_t)height * (uint64_t)(rank + 1) / (uint64_t)world);
}

int64_t PBR_GatherBands(void* nccl_comm, void* stream, int root, int world, int rank, GPU_Texture* frame) {
    if (!frame || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || frame->layer_count != 1 || frame->depth != 1) return PBR_E_BADARG;
    if (world == 1) return 0;
    if (!nccl_comm) return PBR_E_BADARG;
    const uint64_t row_bytes = GPUX_TextureMipBytes(frame, 0) / frame->height;
    char* base = (char*)GPUX_TextureDevicePtr(frame, 0);
    PBR_XferRange* xf = (PBR_XferRange*)malloc(sizeof(PBR_XferRange) * (size_t)world);
    if (!xf) return PBR_E_BADARG;
    uint32_t n = 0;
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        if (r == root || (rank != root && r != rank)) continue;
        uint32_t r0, r1;
        PBR_BandRows(frame->height, world, r, &r0, &r1);
        if (r1 <= r0) continue;
        xf[n].ptr = base + r0 * row_bytes; xf[n].bytes = (uint64_t)(r1 - r0) * row_bytes; xf[n].peer = rank == root ? r : root;
        total += (int64_t)xf[n].bytes;
        ++n;
    }
    int rc = rank == root ? PBR_ExchangeRanges(nccl_comm, stream, NULL, 0, xf, n) : PBR_ExchangeRanges(nccl_comm, stream, xf, n, NULL, 0);
    if (rc == PBR_OK && rank == root) GPUX_InvalidateTexture(frame);
    free(xf);
    return rc == PBR_OK ? total : rc;
}
