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
 */
#define __HIP_PLATFORM_AMD__ 1
#include <rccl/rccl.h>

#include "pbr_host.h"

#include <stdio.h>
#include <stdlib.h>

#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
        fprintf(stderr, "GPU-ERROR: %s failed: %s\n", #call, ncclGetErrorString(r_)); return PBR_E_COMM; } } while (0)

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
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)stream;
    /* one group: RCCL schedules every transfer of the step together (all peers' links at once) */
    NCCL_OK(ncclGroupStart());
    for (uint32_t i = 0; i < n_recvs; ++i) NCCL_OK(ncclRecv(recvs[i].ptr, recvs[i].bytes, ncclInt8, recvs[i].peer, comm, st));
    for (uint32_t i = 0; i < n_sends; ++i) NCCL_OK(ncclSend(sends[i].ptr, sends[i].bytes, ncclInt8, sends[i].peer, comm, st));
    NCCL_OK(ncclGroupEnd());
    return PBR_OK;
}

/* unit lists are small (tens of units per rank) */
#define MAX_UNITS 512

int64_t PBR_GatherUnits(void* nccl_comm, void* stream, int root, int world, int rank, const PBR_IBLMaps* maps,
                        uint32_t min_size, uint32_t env_size) {
    if (!maps || !maps->tex_specular_env_map || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return PBR_E_BADARG;
    if (world == 1) return 0;
    if (!nccl_comm) return PBR_E_BADARG;
    const uint32_t spec = maps->tex_specular_env_map->width;
    const uint32_t irr = maps->irradiance_map ? maps->irradiance_map->width : 0;
    PBR_WorkUnit* units = (PBR_WorkUnit*)malloc(sizeof(PBR_WorkUnit) * MAX_UNITS);
    PBR_XferRange* xf = (PBR_XferRange*)malloc(sizeof(PBR_XferRange) * MAX_UNITS * (size_t)(rank == root ? world : 1));
    if (!units || !xf) { free(units); free(xf); return PBR_E_BADARG; }
    uint32_t n_xf = 0;
    int64_t total = 0;
    int rc = PBR_OK;
    for (int r = 0; r < world && rc == PBR_OK; ++r) {
        if (r == root || (rank != root && r != rank)) continue;           /* the root's own units are already in place */
        uint32_t n = PBR_PartitionIBL(spec, min_size, irr, env_size, world, r, units, MAX_UNITS);
        for (uint32_t k = 0; k < n; ++k) {
            if (units[k].kind == PBR_Unit_BrdfLut) continue;              /* computed redundantly on every rank */
            GPU_Texture* t; uint64_t off, bytes;
            rc = PBR_UnitByteRange(maps, &units[k], &t, &off, &bytes);
            if (rc != PBR_OK) break;
            xf[n_xf].ptr = (char*)GPUX_TextureDevicePtr(t, 0) + off;
            xf[n_xf].bytes = bytes;
            xf[n_xf].peer = rank == root ? r : root;
            total += (int64_t)bytes;
            ++n_xf;
        }
    }
    if (rc == PBR_OK)
        rc = rank == root ? PBR_ExchangeRanges(nccl_comm, stream, NULL, 0, xf, n_xf) : PBR_ExchangeRanges(nccl_comm, stream, xf, n_xf, NULL, 0);
    if (rc == PBR_OK && rank == root) {
        /* the received bytes bypass the backend's op recording: drop the sampler twins built from the old contents */
        GPUX_InvalidateTexture(maps->tex_specular_env_map);
        if (maps->irradiance_map) GPUX_InvalidateTexture(maps->irradiance_map);
    }
    free(units); free(xf);
    return rc == PBR_OK ? total : rc;
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
