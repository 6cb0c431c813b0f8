// CPU harness for the multi-GPU host layer: runs the REAL host/pbr_gather.c and host/pbr_ibl.c with world = N
// processes -- PBR_PartitionIBL -> PBR_RecordUnits -> PBR_GatherUnits, the two overlapped phases of PBR_RunPartitionedIBL, and
// PBR_GatherBands -- against (i) a fake GPU_* backend in this file and (ii) the test-only RCCL stand-in of nccl_stub.c,
// which pbr_gather.c binds through PBR_SetRcclLibrary() exactly as it binds librccl.so.1 in production.
//
// The fake backend keeps "device" memory on the host and executes a recorded dispatch at GPU_GraphSubmit by writing
// value(texture kind, mip, face, y, x) into the rows the dispatch covers; everything else stays poisoned.  Root then checks
// that the gathered maps equal that function EVERYWHERE (every texel computed by exactly the rank that owns it and delivered to
// the right bytes), peers check that nothing but their own rows changed.  Compiled with ASan + UBSan by tests/test_host_cpu.py.
//
//   gather_world <libnccl_stub.so> <world> <specular_size> <irradiance_size> <min_size> <mode>
//   mode: units | overlap | bands | failsend
#define _GNU_SOURCE 1
#include <dlfcn.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include "pbr_host.h"

/* ================= fake GPU_* backend ================= */
typedef struct FTex { GPU_Texture pub; char* mem; uint64_t total; int id; } FTex;
typedef struct FSet { FTex* out; uint32_t out_mip; } FSet;
typedef struct FOp { FTex* out; uint32_t mip, f0, f1, r0, r1; } FOp;
typedef struct FGraph { FOp ops[256]; int n; FSet* set; int submitted; } FGraph;
static int g_tex_ids;

static uint32_t fmt_bytes(GPU_Format f) { return f == GPU_Format_RGBA32F ? 16u : f == GPU_Format_RGBA16F ? 8u : f == GPU_Format_RG16F ? 4u : 4u; }
static uint64_t mip_bytes(const GPU_Texture* t, uint32_t m) {
    uint64_t w = t->width >> m, h = t->height >> m; if (!w) w = 1; if (!h) h = 1;
    return w * h * fmt_bytes(t->format) * t->layer_count;
}
uint64_t GPUX_TextureMipBytes(const GPU_Texture* t, uint32_t m) { return mip_bytes(t, m); }
uint64_t GPUX_TextureMipOffset(const GPU_Texture* t, uint32_t m) { uint64_t o = 0; for (uint32_t l = 0; l < m; ++l) o += mip_bytes(t, l); return o; }
uint64_t GPUX_TextureTotalBytes(const GPU_Texture* t) { return GPUX_TextureMipOffset(t, t->mip_level_count); }
void* GPUX_TextureDevicePtr(GPU_Texture* t, uint32_t m) { return ((FTex*)t)->mem + GPUX_TextureMipOffset(t, m); }
static int g_invalidations;
void GPUX_InvalidateTexture(GPU_Texture* t) { (void)t; ++g_invalidations; }

GPU_Texture* GPU_MakeTexture(GPU_Format f, uint32_t w, uint32_t h, uint32_t d, GPU_TextureFlags fl, const void* p) {
    (void)p;
    FTex* t = (FTex*)calloc(1, sizeof *t);
    t->pub.width = w; t->pub.height = h; t->pub.depth = d; t->pub.format = f; t->pub.flags = fl;
    t->pub.layer_count = (fl & GPU_TextureFlag_Cubemap) ? 6 : 1;
    t->pub.mip_level_count = 1;
    if (fl & GPU_TextureFlag_HasMipmaps) { uint32_t s = w < h ? w : h; t->pub.mip_level_count = 0; while (s) { ++t->pub.mip_level_count; s >>= 1; } }
    t->total = GPUX_TextureTotalBytes(&t->pub);
    t->mem = (char*)malloc(t->total);
    memset(t->mem, 0xCD, t->total);
    t->id = ++g_tex_ids;
    return &t->pub;
}
void GPU_DestroyTexture(GPU_Texture* t) { if (t) { free(((FTex*)t)->mem); free(t); } }

static int g_dummy[8];
GPU_Sampler* GPU_SamplerLinearClamp(void) { return (GPU_Sampler*)&g_dummy[0]; }
GPU_PipelineLayout* GPU_InitPipelineLayout(void) { return (GPU_PipelineLayout*)calloc(1, 16); }
uint32_t GPU_SamplerBinding(GPU_PipelineLayout* l, const char* n) { (void)l; (void)n; return 0; }
uint32_t GPU_TextureBinding(GPU_PipelineLayout* l, const char* n) { (void)l; (void)n; return 1; }
uint32_t GPU_BufferBinding(GPU_PipelineLayout* l, const char* n) { (void)l; (void)n; return 3; }
uint32_t GPU_StorageImageBinding(GPU_PipelineLayout* l, const char* n, GPU_Format f) { (void)l; (void)n; (void)f; return 2; }
void GPU_FinalizePipelineLayout(GPU_PipelineLayout* l) { (void)l; }
void GPU_DestroyPipelineLayout(GPU_PipelineLayout* l) { free(l); }
GPU_String GPU_SPIRVFromGLSL(DS_Arena* a, GPU_ShaderStage s, GPU_PipelineLayout* l, const GPU_ShaderDesc* d, GPU_GLSLErrorArray* e) {
    (void)a; (void)s; (void)l; (void)d; (void)e; GPU_String r = { "ok", 2 }; return r;
}
GPU_String GPU_JoinGLSLErrorString(DS_Arena* a, GPU_GLSLErrorArray e) { (void)a; (void)e; GPU_String r = { "", 0 }; return r; }
GPU_ComputePipeline* GPU_MakeComputePipeline(GPU_PipelineLayout* l, const GPU_ShaderDesc* d) { (void)l; (void)d; return (GPU_ComputePipeline*)calloc(1, 16); }
void GPU_DestroyComputePipeline(GPU_ComputePipeline* p) { free(p); }
GPU_DescriptorArena* GPU_MakeDescriptorArena(void) { return (GPU_DescriptorArena*)calloc(1, 16); }
void GPU_ResetDescriptorArena(GPU_DescriptorArena* a) { (void)a; }
void GPU_DestroyDescriptorArena(GPU_DescriptorArena* a) { free(a); }
/* descriptor sets of an arena die with the test process */
GPU_DescriptorSet* GPU_InitDescriptorSet(GPU_DescriptorArena* a, GPU_PipelineLayout* l) { (void)a; (void)l; return (GPU_DescriptorSet*)calloc(1, sizeof(FSet)); }
void GPU_DestroyDescriptorSet(GPU_DescriptorSet* s) { free(s); }
void GPU_SetSamplerBinding(GPU_DescriptorSet* s, uint32_t b, GPU_Sampler* v) { (void)s; (void)b; (void)v; }
void GPU_SetTextureBinding(GPU_DescriptorSet* s, uint32_t b, GPU_Texture* v) { (void)s; (void)b; (void)v; }
void GPU_SetStorageImageBinding(GPU_DescriptorSet* s, uint32_t b, GPU_Texture* v, uint32_t mip) { (void)b; ((FSet*)s)->out = (FTex*)v; ((FSet*)s)->out_mip = mip; }
void GPU_FinalizeDescriptorSet(GPU_DescriptorSet* s) { (void)s; }
GPU_Graph* GPU_MakeGraph(void) { return (GPU_Graph*)calloc(1, sizeof(FGraph)); }
void GPU_DestroyGraph(GPU_Graph* g) { free(g); }
void GPU_OpBindComputePipeline(GPU_Graph* g, GPU_ComputePipeline* p) { (void)g; (void)p; }
void GPU_OpBindComputeDescriptorSet(GPU_Graph* g, GPU_DescriptorSet* s) { ((FGraph*)g)->set = (FSet*)s; }
void GPU_OpPushComputeConstants(GPU_Graph* g, GPU_PipelineLayout* l, void* d, uint32_t n) { (void)g; (void)l; (void)d; (void)n; }
void GPU_OpDispatch(GPU_Graph* g, uint32_t x, uint32_t y, uint32_t z) { (void)g; (void)x; (void)y; (void)z; fprintf(stderr, "harness: unexpected GPU_OpDispatch\n"); exit(40); }
void GPUX_OpDispatchRows(GPU_Graph* g_, uint32_t f0, uint32_t f1, uint32_t r0, uint32_t r1) {
    FGraph* g = (FGraph*)g_;
    if (!g->set || !g->set->out || g->n >= 256) { fprintf(stderr, "harness: dispatch without an output binding\n"); exit(41); }
    FOp* o = &g->ops[g->n++];
    o->out = g->set->out; o->mip = g->set->out_mip; o->f0 = f0; o->f1 = f1; o->r0 = r0; o->r1 = r1;
}
void* GPUX_GraphStream(GPU_Graph* g) { return g; }

static uint32_t value_of(int tex_id, uint32_t mip, uint32_t f, uint32_t y, uint32_t x, uint32_t c) {
    uint32_t h = (uint32_t)tex_id * 0x9E3779B1u ^ (mip + 1) * 0x85EBCA77u ^ (f + 1) * 0xC2B2AE3Du ^ (y + 1) * 0x27D4EB2Fu ^ (x + 1) * 0x165667B1u ^ (c + 1) * 0x9E3779B9u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h | 1u;                                            /* never the poison pattern */
}
static long g_written_rows;
void GPU_GraphSubmit(GPU_Graph* g_) {
    FGraph* g = (FGraph*)g_;
    for (int i = 0; i < g->n; ++i) {
        FOp* o = &g->ops[i];
        const GPU_Texture* t = &o->out->pub;
        uint32_t w = t->width >> o->mip, h = t->height >> o->mip; if (!w) w = 1; if (!h) h = 1;
        if (o->f1 > t->layer_count || o->r1 > h || o->f0 >= o->f1 || o->r0 >= o->r1) { fprintf(stderr, "harness: dispatch outside its level\n"); exit(42); }
        uint32_t* base = (uint32_t*)(o->out->mem + GPUX_TextureMipOffset(t, o->mip));
        for (uint32_t f = o->f0; f < o->f1; ++f) for (uint32_t y = o->r0; y < o->r1; ++y) {
            for (uint32_t x = 0; x < w; ++x) for (uint32_t c = 0; c < 4; ++c)
                base[(((uint64_t)f * h + y) * w + x) * 4 + c] = value_of(o->out->id, o->mip, f, y, x, c);
            ++g_written_rows;
        }
    }
    g->n = 0; g->set = NULL; g->submitted = 1;
}
void GPU_GraphWait(GPU_Graph* g) { ((FGraph*)g)->submitted = 0; }

/* ================= the ranks ================= */
typedef void* (*comm_create_fn)(int, int, const int*);
typedef void (*counters_fn)(long*, long*, long*, long*, int*);
typedef void (*fail_at_fn)(long);

static int check_level(FTex* t, uint32_t mip, int expect_all, const uint8_t* own_rows /* [6][h] or NULL */, const char* who) {
    const GPU_Texture* p = &t->pub;
    uint32_t w = p->width >> mip, h = p->height >> mip; if (!w) w = 1; if (!h) h = 1;
    const uint32_t* base = (const uint32_t*)(t->mem + GPUX_TextureMipOffset(p, mip));
    for (uint32_t f = 0; f < p->layer_count; ++f) for (uint32_t y = 0; y < h; ++y) {
        int want = expect_all || (own_rows && own_rows[f * 4096 + y]);
        for (uint32_t x = 0; x < w; ++x) for (uint32_t c = 0; c < 4; ++c) {
            uint32_t got = base[(((uint64_t)f * h + y) * w + x) * 4 + c];
            uint32_t exp = want ? value_of(t->id, mip, f, y, x, c) : 0xCDCDCDCDu;
            if (got != exp) { fprintf(stderr, "%s: tex %d mip %u face %u row %u x %u: %08x, expected %08x\n", who, t->id, mip, f, y, x, got, exp); return 0; }
        }
    }
    return 1;
}

static int run_rank(const char* stub, int world, int rank, const int* fds, uint32_t spec, uint32_t irr, uint32_t min_size, const char* mode) {
    if (PBR_SetRcclLibrary(stub) != PBR_OK) return 10;
    void* h = dlopen(stub, RTLD_NOW | RTLD_NOLOAD);            /* the copy pbr_gather.c bound */
    if (!h) return 11;
    comm_create_fn mk; counters_fn counters; fail_at_fn fail_send;
    *(void**)&mk = dlsym(h, "stub_comm_create"); *(void**)&counters = dlsym(h, "stub_counters"); *(void**)&fail_send = dlsym(h, "stub_fail_send_at");
    void* comm = mk(world, rank, fds);
    int cn = 0, cr = -1;
    if (PBR_CommInfo(comm, &cn, &cr) != PBR_OK || cn != world || cr != rank) return 12;
    char who[64]; snprintf(who, sizeof who, "rank %d/%d (%s)", rank, world, mode);
    const uint32_t env_size = spec / 2 ? spec / 2 : 1;

    if (!strcmp(mode, "bands")) {
        /* a frame whose height does not divide by the world size */
        const uint32_t W = 96, H = spec + 7;
        FTex* frame = (FTex*)GPU_MakeTexture(GPU_Format_RGBA16F, W, H, 1, GPU_TextureFlag_RenderTarget, NULL);
        uint32_t r0, r1; PBR_BandRows(H, world, rank, &r0, &r1);
        uint16_t* px = (uint16_t*)frame->mem;
        for (uint32_t y = r0; y < r1; ++y) for (uint32_t x = 0; x < W * 4; ++x) px[(uint64_t)y * W * 4 + x] = (uint16_t)(value_of(7, 0, 0, y, x, 0) | 1u);
        int64_t moved = PBR_GatherBands(comm, NULL, 0, world, rank, &frame->pub);
        if (moved < 0) return 20;
        for (uint32_t y = 0; y < H; ++y) for (uint32_t x = 0; x < W * 4; ++x) {
            int mine = y >= r0 && y < r1;
            uint16_t exp = (rank == 0 || mine) ? (uint16_t)(value_of(7, 0, 0, y, x, 0) | 1u) : (uint16_t)0xCDCD;
            if (px[(uint64_t)y * W * 4 + x] != exp) { fprintf(stderr, "%s: band row %u wrong\n", who, y); return 21; }
        }
        const int64_t want = rank == 0 ? (int64_t)(H - (r1 - r0)) * W * 8 : (int64_t)(r1 - r0) * W * 8;
        if (moved != want) { fprintf(stderr, "%s: moved %lld bytes, expected %lld\n", who, (long long)moved, (long long)want); return 22; }
        GPU_DestroyTexture(&frame->pub);
        return 0;
    }

    if (!strcmp(mode, "failsend")) {
        /* one process, transfers to self: the second ncclSend fails inside the group -- the group must be closed again */
        FTex* t = (FTex*)GPU_MakeTexture(GPU_Format_RGBA32F, 16, 16, 1, 0, NULL);
        PBR_XferRange s[3], r[3];
        for (int i = 0; i < 3; ++i) { s[i].ptr = t->mem + 256 * i; s[i].bytes = 128; s[i].peer = rank; r[i].ptr = t->mem + 2048 + 256 * i; r[i].bytes = 128; r[i].peer = rank; }
        fail_send(1);
        int rc = PBR_ExchangeRanges(comm, NULL, s, 3, r, 3);
        long st, en; int depth; counters(&st, &en, NULL, NULL, &depth);
        if (rc != PBR_E_COMM) { fprintf(stderr, "%s: failing ncclSend returned %d\n", who, rc); return 30; }
        if (st != en || depth != 0) { fprintf(stderr, "%s: %ld group starts, %ld ends, depth %d after a failed send\n", who, st, en, depth); return 31; }
        fail_send(-1);
        /* the thread is usable again: the next exchange is not swallowed by a dangling group */
        for (int i = 0; i < 3; ++i) memset(s[i].ptr, 0x11 * (i + 1), 128);
        rc = PBR_ExchangeRanges(comm, NULL, s, 3, r, 3);
        if (rc != PBR_OK) return 32;
        for (int i = 0; i < 3; ++i) if (memcmp(s[i].ptr, r[i].ptr, 128)) { fprintf(stderr, "%s: exchange after the failure did not run\n", who); return 33; }
        GPU_DestroyTexture(&t->pub);
        return 0;
    }

    PBR_IBLMaps maps;
    PBR_MakeIBLMaps(&maps, irr ? irr : 8, 16, spec);
    if (!irr) { GPU_DestroyTexture(maps.irradiance_map); maps.irradiance_map = NULL; }
    GPU_Texture* env = GPU_MakeTexture(GPU_Format_RGBA32F, env_size, env_size, 1, GPU_TextureFlag_Cubemap | GPU_TextureFlag_HasMipmaps, NULL);
    PBR_IBLPipelines* pipes = PBR_MakeIBLPipelines();
    GPU_DescriptorArena* arena = GPU_MakeDescriptorArena();
    GPU_Graph* g = GPU_MakeGraph();
    GPU_Graph* g2 = GPU_MakeGraph();
    static PBR_WorkUnit units[4096];
    const uint32_t n = PBR_PartitionIBL(spec, min_size, irr, env_size, world, rank, units, 4096);
    /* rows this rank owns */
    static uint8_t own[16][6 * 4096]; static uint8_t iown[6 * 4096];
    memset(own, 0, sizeof own); memset(iown, 0, sizeof iown);
    int64_t own_bytes = 0;
    for (uint32_t k = 0; k < n; ++k) for (uint32_t f = units[k].face0; f < units[k].face1; ++f) for (uint32_t y = units[k].row0; y < units[k].row1; ++y) {
        if (units[k].kind == PBR_Unit_Irradiance) { iown[f * 4096 + y] = 1; own_bytes += (int64_t)irr * 16; }
        else { own[units[k].mip][f * 4096 + y] = 1; uint32_t s = spec >> units[k].mip; own_bytes += (int64_t)(s ? s : 1) * 16; }
    }
    int64_t moved;
    if (!strcmp(mode, "overlap")) {
        moved = PBR_RunPartitionedIBL(pipes, g, g2, arena, env, &maps, comm, 0, world, rank, min_size, 0x2u);
        GPU_GraphWait(g2); GPU_GraphWait(g);
    } else {
        PBR_RecordUnits(pipes, g, arena, env, &maps, units, n);
        GPU_GraphSubmit(g);
        moved = PBR_GatherUnits(comm, GPUX_GraphStream(g), 0, world, rank, &maps, min_size, env_size);
        GPU_GraphWait(g);
    }
    if (moved < 0) { fprintf(stderr, "%s: gather returned %lld\n", who, (long long)moved); return 50; }
    /* what must have crossed: a peer sends exactly the bytes of its units; root receives everything it does not own */
    int64_t total_bytes = 0; uint32_t mips = 0;
    for (uint32_t s = spec, m = 0; s >= min_size && s >= 1; s /= 2, ++m) { total_bytes += 6ll * s * s * 16; mips = m + 1; if (s == 1) break; }
    if (irr) total_bytes += 6ll * irr * irr * 16;
    const int64_t want = world == 1 ? 0 : (rank == 0 ? total_bytes - own_bytes : own_bytes);
    if (moved != want) { fprintf(stderr, "%s: moved %lld bytes, expected %lld\n", who, (long long)moved, (long long)want); return 51; }
    for (uint32_t m = 0; m < mips; ++m) if (!check_level((FTex*)maps.tex_specular_env_map, m, rank == 0, own[m], who)) return 52;
    if (irr && !check_level((FTex*)maps.irradiance_map, 0, rank == 0, iown, who)) return 53;
    if (rank == 0 && world > 1 && g_invalidations == 0) { fprintf(stderr, "%s: root's sampler twins were not invalidated\n", who); return 54; }
    long st, en; int depth; counters(&st, &en, NULL, NULL, &depth);
    if (st != en || depth != 0) return 55;
    GPU_DestroyGraph(g); GPU_DestroyGraph(g2); GPU_DestroyDescriptorArena(arena); PBR_DestroyIBLPipelines(pipes);
    GPU_DestroyTexture(env); PBR_DestroyIBLMaps(&maps);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: gather_world <stub.so> <world> <spec> <irr> <min_size> <units|overlap|bands|failsend>\n"); return 2; }
    const char* stub = argv[1];
    const int world = atoi(argv[2]);
    const uint32_t spec = (uint32_t)atoi(argv[3]), irr = (uint32_t)atoi(argv[4]), min_size = (uint32_t)atoi(argv[5]);
    const char* mode = argv[6];
    if (world < 1 || world > 16 || spec > 4096) return 2;
    /* one socket pair per pair of ranks, created before the fork */
    static int fd[16][16];
    for (int i = 0; i < world; ++i) for (int j = 0; j < world; ++j) fd[i][j] = -1;
    for (int i = 0; i < world; ++i) for (int j = i + 1; j < world; ++j) {
        int sv[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 3;
        fcntl(sv[0], F_SETFL, fcntl(sv[0], F_GETFL) | O_NONBLOCK); fcntl(sv[1], F_SETFL, fcntl(sv[1], F_GETFL) | O_NONBLOCK);
        fd[i][j] = sv[0]; fd[j][i] = sv[1];
    }
    pid_t pids[16];
    for (int r = 0; r < world; ++r) {
        pids[r] = fork();
        if (pids[r] < 0) return 4;
        if (pids[r] == 0) {
            for (int i = 0; i < world; ++i) for (int j = 0; j < world; ++j) if (i != r && fd[i][j] >= 0) close(fd[i][j]);
            int rc = run_rank(stub, world, r, fd[r], spec, irr, min_size, mode);
            fflush(stderr);
            _exit(rc);
        }
    }
    for (int i = 0; i < world; ++i) for (int j = 0; j < world; ++j) if (fd[i][j] >= 0) close(fd[i][j]);
    int bad = 0;
    for (int r = 0; r < world; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %d: exit status %d (signal %d)\n", r, WIFEXITED(st) ? WEXITSTATUS(st) : -1, WIFSIGNALED(st) ? WTERMSIG(st) : 0); bad = 1; }
    }
    if (!bad) printf("ok: world %d spec %u irr %u min %u mode %s\n", world, spec, irr, min_size, mode);
    return bad;
}
