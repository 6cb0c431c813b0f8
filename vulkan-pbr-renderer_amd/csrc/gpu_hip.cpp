// gpu_hip.cpp -- HIP / gfx950 implementation of the GPU_* boundary declared in include/gpu_hip.h.
//
// Takes the place of src/gpu/gpu_vulkan.c for the IBL-precompute and shade hot path:
//   * textures/buffers are linear HBM allocations (cube = [mip][face][y][x], tight);
//   * sampled cubes carry a lazily rebuilt "bordered" twin (seamless-filter apron, k_cube.hip);
//   * descriptor sets are plain slot arrays resolved by binding NAME at launch;
//   * compute / graphics pipelines resolve to built-in kernels by shader identity;
//   * a graph is a HIP stream plus a recorded op list, launched at GPU_GraphSubmit.
// Error convention follows gpu_vulkan.c:6-8,387-392: print "GPU-ERROR: ..." and trap, unless a
// handler was installed through GPUX_SetErrorHandler.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "gpux.h"
#include "pbr_kernels.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static GPUX_ErrorHandler g_err_handler = nullptr;
static void* g_err_user = nullptr;

static void gpu_fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g_err_handler) { g_err_handler(buf, g_err_user); return; }
    fprintf(stderr, "GPU-ERROR: %s\n", buf);
    fflush(stderr);
    abort();
}
#define GPU_REQUIRE(cond, ret, ...) do { if (!(cond)) { gpu_fail(__VA_ARGS__); return ret; } } while (0)
#define GPU_REQUIRE_V(cond, ...) do { if (!(cond)) { gpu_fail(__VA_ARGS__); return; } } while (0)
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { gpu_fail("%s: HIP call failed: %s (%s)", __func__, #call, hipGetErrorString(e_)); } } while (0)

// ------------------------------------------------------------------------------------------
// object model
// ------------------------------------------------------------------------------------------
enum BindKind { Bind_Texture, Bind_Sampler, Bind_Buffer, Bind_StorageImage };
enum KernelId { Kernel_None = 0, Kernel_BrdfLut, Kernel_Irradiance, Kernel_Prefilter, Kernel_Lighting, Kernel_LightgridSweep, Kernel_TaaResolve, Kernel_FinalPost, Kernel_BloomDown, Kernel_BloomUp };

struct GPU_Sampler { GPU_SamplerDesc desc; bool shared; };

struct LayoutBinding { std::string name; BindKind kind; GPU_Format format; };
struct GPU_PipelineLayout { std::vector<LayoutBinding> bindings; bool finalized = false; };

struct TextureImpl {
    GPU_Texture base;                 // public part first (gpu_vulkan.c:182-205 does the same)
    void* dev = nullptr;              // [mip][layer][z][y][x]
    size_t bytes = 0;
    std::vector<size_t> mip_offset;   // bytes
    uint32_t texel_bytes = 0;
    void* bordered = nullptr;         // bordered pyramid twin (RGBA32F cubes only), built on demand
    size_t bordered_bytes = 0;
    bool bordered_valid = false;                   // apron twin valid for levels >= bordered_from
    int bordered_from = 0;
    bool owns_memory = true;          // false: GPUX_MakeTextureExternal (caller-owned HBM, e.g. a torch tensor)
    // 2x2-footprint "cells" twin of the levels with n <= 512 (levels cells_first.., back to back), built per level on demand
    void* cells = nullptr; int cells_first = 0; std::vector<size_t> cells_off; std::vector<char> cells_valid;
    void* lut_cells = nullptr; bool lut_cells_valid = false;      // RG16F 2-D textures sampled by the shade pass
    // {min, max} of a level's RGB values (GPUX_SetPrefilterTolerance), measured on demand; dropped with the cells twin
    std::vector<char> range_valid; std::vector<float> range_min, range_max; void* range_dev = nullptr;
};
struct BufferImpl {
    GPU_Buffer base;
    void* dev = nullptr;              // device-visible pointer (== base.data for CPU buffers)
    bool pinned_host = false;
};

struct Slot { BindKind kind; bool set = false; TextureImpl* tex = nullptr; uint32_t mip = 0; bool whole = true;
              GPU_Sampler* sampler = nullptr; BufferImpl* buf = nullptr; };
struct GPU_DescriptorSet { GPU_PipelineLayout* layout; std::vector<Slot> slots; bool finalized = false; GPU_DescriptorArena* arena = nullptr; };
struct GPU_DescriptorArena { std::vector<GPU_DescriptorSet*> sets; };

struct GPU_ComputePipeline { GPU_PipelineLayout* layout; KernelId kernel; };
struct GPU_RenderPass { GPU_RenderPassDesc desc; std::vector<GPU_TextureView> targets; };
struct GPU_GraphicsPipeline { GPU_PipelineLayout* layout; GPU_RenderPass* pass; KernelId kernel; int shade_flags; bool blend_additive = false; };

enum OpKind { Op_Dispatch, Op_Shade, Op_MipGen, Op_CopyB2T, Op_CopyT2B, Op_CopyB2B, Op_Blit, Op_Clear };
struct Op {
    OpKind kind;
    std::string name;
    // dispatch
    GPU_ComputePipeline* cpipe = nullptr;
    GPU_DescriptorSet* set = nullptr;
    uint8_t push[128]; uint32_t push_size = 0;
    uint32_t face0 = 0, face1 = 6, row0 = 0, row1 = 0;
    bool rows_explicit = false;
    uint32_t gx = 0, gy = 0, gz = 0;
    // shade
    GPU_GraphicsPipeline* gpipe = nullptr;
    GPU_RenderPass* pass = nullptr;
    // transfers
    TextureImpl* tex = nullptr; TextureImpl* tex2 = nullptr;
    BufferImpl* buf = nullptr; BufferImpl* buf2 = nullptr;
    uint32_t mip = 0, mip2 = 0, layer0 = 0, layer_count = 0, layer2 = 0;
    uint64_t off_a = 0, off_b = 0, size = 0;
    float clear[4]; uint32_t cleari[4]; int clear_mode = 0;
    // submit-time folding (fold_blits): a 1:1 blit whose copy is consumed only by an additive bloom draw onto its target is not
    // executed; that draw takes the blend operand from the blit's source instead
    bool folded = false;
    TextureImpl* blend_tex = nullptr; uint32_t blend_mip = 0;
    bool skip_level0 = false;                      // Op_Clear of all levels whose level 0 the next op overwrites entirely
};
struct DrawParams { GPU_GraphicsPipeline* pipeline; GPU_DescriptorSet* set; };
struct GPU_Graph {
    hipStream_t stream = nullptr;
    hipStream_t cur = nullptr;                     // stream the op being executed launches on (stream, or a side stream)
    std::vector<hipStream_t> side;                 // side streams for overlapping independent tile dispatches (created on first use)
    std::vector<hipEvent_t> sync_ev;               // untimed fork/join events
    hipEvent_t order_ev = nullptr;                 // recorded on this graph's stream when the NEXT submitted graph must follow it
    // head / tail split of the last submission (GPU_GraphSubmit): `mid_ev` is recorded in front of the first bloom draw; the next graph's
    // leading ops that touch none of the textures in tail_reads / tail_writes wait for it instead of for the whole graph
    hipEvent_t mid_ev = nullptr;
    bool mid_recorded = false;
    std::vector<TextureImpl*> tail_reads, tail_writes;
    size_t sync_used = 0;
    std::vector<Op> ops;
    bool submitted = false;
    GPU_ComputePipeline* bound_cpipe = nullptr;
    GPU_DescriptorSet* bound_cset = nullptr;
    uint8_t push[128]; uint32_t push_size = 0;
    GPU_RenderPass* preparing = nullptr; GPU_RenderPass* in_pass = nullptr;
    std::vector<DrawParams> draw_params;
    int bound_draw = -1;
    // hipGraph replay (GPUX_SetGraphReplay): the executable graph of the previous submission, updated in place when the
    // next one has the same shape
    hipGraphExec_t exec = nullptr;
    bool replay_broken = false;                    // a capture failed on this graph: it stays on plain launches
    uint64_t replay_launches = 0, replay_updates = 0, replay_instantiations = 0;
    // timing of the last waited submission
    std::vector<hipEvent_t> ev;
    std::vector<std::string> timed_names;
    std::vector<float> timed_ms;
    // one event pair on the graph's main stream around the whole submission (first op start .. last join): the busy span of the
    // graph's kernels even when its dispatches overlap on side streams (the sum of per-op times would count the overlap twice)
    hipEvent_t span_a = nullptr, span_b = nullptr;
    bool span_recorded = false;
    float span_ms = 0.0f;
};

struct DeviceTable { void* dev = nullptr; int count = 0; float alpha = 0.0f; std::vector<float> weights; /* prefilter tables: w_i of the kept entries, in order */ };
struct TableKey {
    int kind, n, aux; uint32_t rbits;
    bool operator<(const TableKey& o) const {
        if (kind != o.kind) return kind < o.kind;
        if (n != o.n) return n < o.n;
        if (aux != o.aux) return aux < o.aux;
        return rbits < o.rbits;
    }
};

static struct {
    GPU_Graph* last_submitted = nullptr;            // submission order between graphs (the reference has one queue: gpu_vulkan.c:2481)
    int tile_streams = -1;                          // side streams for small independent precompute dispatches (-1: PBR_TILE_STREAMS or 4)
    bool init = false;
    int device = -1;
    GPU_Sampler samplers[6];
    std::map<TableKey, DeviceTable> tables;
    bool timing = false;
    float prefilter_tol = 0.0f;                     // GPUX_SetPrefilterTolerance: 0 = exact sums (default)
    int kept_samples[32] = {0};                     // per output mip: samples kept by the last prefilter dispatch
    int replay = -1;                                // GPUX_SetGraphReplay: submissions go through an instantiated hipGraph (-1: PBR_GRAPH_REPLAY or 0)
} G;

static const char kTokenLut[] = "HIPK1:gen_brdf_integration_map";
static const char kTokenIrr[] = "HIPK3:gen_irradiance_map";
static const char kTokenPre[] = "HIPK4:gen_prefiltered_env_map";
static const char kTokenLit[] = "HIPK5:lighting_pass";
static const char kTokenSweep[] = "HIPK7:lightgrid_sweep";
static const char kTokenTaa[] = "HIPK8:taa_resolve";
static const char kTokenFinal[] = "HIPK9:final_post_process";
static const char kTokenBloomDown[] = "HIPK10:bloom_downsample";
static const char kTokenBloomUp[] = "HIPK11:bloom_upsample";

// ------------------------------------------------------------------------------------------
// formats  [gpu.h:99-144]
// ------------------------------------------------------------------------------------------
GPU_API GPU_FormatInfo GPUX_GetFormatInfo(GPU_Format f) {
    struct Row { GPU_Format f; uint32_t ext, size; bool s, v, c, d, st, i; const char* glsl; };
    static const Row rows[] = {
        {GPU_Format_R8UN, 1, 1, 1, 1, 1, 0, 0, 0, "r8"}, {GPU_Format_RG8UN, 1, 2, 1, 1, 1, 0, 0, 0, "rg8"},
        {GPU_Format_RGBA8UN, 1, 4, 1, 1, 1, 0, 0, 0, "rgba8"}, {GPU_Format_BGRA8UN, 1, 4, 1, 1, 1, 0, 0, 0, nullptr},
        {GPU_Format_R16F, 1, 2, 1, 1, 1, 0, 0, 0, "r16f"}, {GPU_Format_RG16F, 1, 4, 1, 1, 1, 0, 0, 0, "rg16f"},
        {GPU_Format_RGB16F, 1, 6, 0, 1, 0, 0, 0, 0, nullptr}, {GPU_Format_RGBA16F, 1, 8, 1, 1, 1, 0, 0, 0, "rgba16f"},
        {GPU_Format_R32F, 1, 4, 1, 1, 1, 0, 0, 0, "r32f"}, {GPU_Format_RG32F, 1, 8, 1, 1, 1, 0, 0, 0, "rg32f"},
        {GPU_Format_RGB32F, 1, 12, 0, 1, 0, 0, 0, 0, nullptr}, {GPU_Format_RGBA32F, 1, 16, 1, 1, 1, 0, 0, 0, "rgba32f"},
        {GPU_Format_R8I, 1, 1, 1, 1, 1, 0, 0, 1, "r8ui"}, {GPU_Format_R16I, 1, 2, 1, 1, 1, 0, 0, 1, "r16ui"},
        {GPU_Format_RG16I, 1, 4, 1, 1, 1, 0, 0, 1, "rg16ui"}, {GPU_Format_RGBA16I, 1, 8, 1, 1, 1, 0, 0, 1, "rgba16ui"},
        {GPU_Format_R32I, 1, 4, 1, 1, 1, 0, 0, 1, "r32ui"}, {GPU_Format_RG32I, 1, 8, 1, 1, 1, 0, 0, 1, "rg32ui"},
        {GPU_Format_RGB32I, 1, 12, 1, 1, 0, 0, 0, 1, nullptr}, {GPU_Format_RGBA32I, 1, 16, 1, 1, 1, 0, 0, 1, "rgba32ui"},
        {GPU_Format_R64I, 1, 8, 0, 0, 0, 0, 0, 1, "r64ui"}, {GPU_Format_D16UN, 1, 2, 1, 0, 0, 1, 0, 0, nullptr},
        {GPU_Format_D32F_Or_X8D24UN, 1, 4, 1, 0, 0, 1, 0, 0, nullptr}, {GPU_Format_D32FS8I_Or_D24UNS8I, 1, 5, 0, 0, 0, 1, 0, 0, nullptr},
        {GPU_Format_D24UNS8I_Or_D32FS8I, 1, 4, 0, 0, 0, 1, 0, 0, nullptr},
        {GPU_Format_BC1_RGB_UN, 4, 8, 1, 0, 0, 0, 0, 0, nullptr}, {GPU_Format_BC1_RGBA_UN, 4, 8, 1, 0, 0, 0, 0, 0, nullptr},
        {GPU_Format_BC3_RGBA_UN, 4, 16, 1, 0, 0, 0, 0, 0, nullptr}, {GPU_Format_BC5_UN, 4, 16, 1, 0, 0, 0, 0, 0, nullptr},
    };
    for (const Row& r : rows)
        if (r.f == f) { GPU_FormatInfo o = {r.ext, r.size, r.s, r.v, r.c, r.d, r.st, r.i, r.glsl}; return o; }
    GPU_FormatInfo z = {0, 0, false, false, false, false, false, false, nullptr};
    return z;
}

// ------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------
GPU_API void GPUX_SetDevice(int idx) { G.device = idx; }
GPU_API int GPUX_GetDevice(void) { return G.device; }
GPU_API void GPUX_SetErrorHandler(GPUX_ErrorHandler h, void* user) { g_err_handler = h; g_err_user = user; }
GPU_API const char* GPUX_BackendName(void) { return "hip-gfx950"; }

static void make_shared_sampler(GPU_Sampler* s, GPU_Filter f, GPU_AddressMode m) {
    memset(&s->desc, 0, sizeof s->desc);
    s->desc.min_filter = s->desc.mag_filter = s->desc.mipmap_mode = f;     // gpu_vulkan.c:935-943
    s->desc.address_modes[0] = s->desc.address_modes[1] = s->desc.address_modes[2] = m;
    s->desc.max_lod = 1000.0f;
    s->shared = true;
}

GPU_API void GPU_Init(GPU_WindowHandle window) {
    (void)window;   // headless
    GPU_REQUIRE_V(!G.init, "GPU_Init: already initialised");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    GPU_REQUIRE_V(e == hipSuccess && count > 0, "GPU_Init: no HIP device (%s)", hipGetErrorString(e));
    if (G.device < 0) {
        const char* lr = getenv("LOCAL_RANK");
        G.device = lr ? atoi(lr) % count : 0;
    }
    GPU_REQUIRE_V(G.device < count, "GPU_Init: device %d out of range (%d devices)", G.device, count);
    HIP_OK(hipSetDevice(G.device));
    make_shared_sampler(&G.samplers[0], GPU_Filter_Linear, GPU_AddressMode_Wrap);
    make_shared_sampler(&G.samplers[1], GPU_Filter_Linear, GPU_AddressMode_Clamp);
    make_shared_sampler(&G.samplers[2], GPU_Filter_Linear, GPU_AddressMode_Mirror);
    make_shared_sampler(&G.samplers[3], GPU_Filter_Nearest, GPU_AddressMode_Clamp);
    make_shared_sampler(&G.samplers[4], GPU_Filter_Nearest, GPU_AddressMode_Wrap);
    make_shared_sampler(&G.samplers[5], GPU_Filter_Nearest, GPU_AddressMode_Mirror);
    G.init = true;
}

GPU_API void GPU_Deinit(void) {
    if (!G.init) return;
    (void)hipDeviceSynchronize();
    for (auto& kv : G.tables) (void)hipFree(kv.second.dev);
    G.tables.clear();
    G.init = false;
}

GPU_API void GPU_WaitUntilIdle(void) { if (G.init) HIP_OK(hipDeviceSynchronize()); }

// ------------------------------------------------------------------------------------------
// samplers
// ------------------------------------------------------------------------------------------
GPU_API GPU_Sampler* GPU_SamplerLinearWrap(void) { return &G.samplers[0]; }
GPU_API GPU_Sampler* GPU_SamplerLinearClamp(void) { return &G.samplers[1]; }
GPU_API GPU_Sampler* GPU_SamplerLinearMirror(void) { return &G.samplers[2]; }
GPU_API GPU_Sampler* GPU_SamplerNearestClamp(void) { return &G.samplers[3]; }
GPU_API GPU_Sampler* GPU_SamplerNearestWrap(void) { return &G.samplers[4]; }
GPU_API GPU_Sampler* GPU_SamplerNearestMirror(void) { return &G.samplers[5]; }
GPU_API GPU_Sampler* GPU_MakeSampler(const GPU_SamplerDesc* desc) {
    GPU_REQUIRE(desc, nullptr, "GPU_MakeSampler: desc is NULL");
    GPU_Sampler* s = new GPU_Sampler();
    s->desc = *desc; s->shared = false;
    return s;
}
GPU_API void GPU_DestroySampler(GPU_Sampler* s) {
    if (!s) return;
    GPU_REQUIRE_V(!s->shared, "GPU_DestroySampler: shared sampler shortcuts must not be destroyed");
    delete s;
}

// ------------------------------------------------------------------------------------------
// pipeline layouts
// ------------------------------------------------------------------------------------------
GPU_API GPU_PipelineLayout* GPU_InitPipelineLayout(void) { return new GPU_PipelineLayout(); }
static GPU_Binding add_binding(GPU_PipelineLayout* l, const char* name, BindKind k, GPU_Format f, const char* fn) {
    GPU_REQUIRE(l && name, 0, "%s: NULL argument", fn);
    GPU_REQUIRE(!l->finalized, 0, "%s: layout already finalised", fn);
    l->bindings.push_back({name, k, f});
    return (GPU_Binding)(l->bindings.size() - 1);
}
GPU_API GPU_Binding GPU_TextureBinding(GPU_PipelineLayout* l, const char* name) { return add_binding(l, name, Bind_Texture, GPU_Format_Invalid, __func__); }
GPU_API GPU_Binding GPU_SamplerBinding(GPU_PipelineLayout* l, const char* name) { return add_binding(l, name, Bind_Sampler, GPU_Format_Invalid, __func__); }
GPU_API GPU_Binding GPU_BufferBinding(GPU_PipelineLayout* l, const char* name) { return add_binding(l, name, Bind_Buffer, GPU_Format_Invalid, __func__); }
GPU_API GPU_Binding GPU_StorageImageBinding(GPU_PipelineLayout* l, const char* name, GPU_Format f) { return add_binding(l, name, Bind_StorageImage, f, __func__); }
GPU_API void GPU_FinalizePipelineLayout(GPU_PipelineLayout* l) {
    GPU_REQUIRE_V(l, "GPU_FinalizePipelineLayout: NULL layout");
    l->finalized = true;
}
GPU_API void GPU_DestroyPipelineLayout(GPU_PipelineLayout* l) { delete l; }

static int find_binding(const GPU_PipelineLayout* l, const char* name) {
    for (size_t i = 0; i < l->bindings.size(); ++i) if (l->bindings[i].name == name) return (int)i;
    return -1;
}

// ------------------------------------------------------------------------------------------
// descriptor sets
// ------------------------------------------------------------------------------------------
GPU_API GPU_DescriptorArena* GPU_MakeDescriptorArena(void) { return new GPU_DescriptorArena(); }
GPU_API void GPU_ResetDescriptorArena(GPU_DescriptorArena* a) {
    GPU_REQUIRE_V(a, "GPU_ResetDescriptorArena: NULL arena");
    for (GPU_DescriptorSet* s : a->sets) delete s;
    a->sets.clear();
}
GPU_API void GPU_DestroyDescriptorArena(GPU_DescriptorArena* a) {
    if (!a) return;
    for (GPU_DescriptorSet* s : a->sets) delete s;
    delete a;
}
GPU_API GPU_DescriptorSet* GPU_InitDescriptorSet(GPU_DescriptorArena* arena, GPU_PipelineLayout* layout) {
    GPU_REQUIRE(layout && layout->finalized, nullptr, "GPU_InitDescriptorSet: layout is NULL or not finalised");
    GPU_DescriptorSet* s = new GPU_DescriptorSet();
    s->layout = layout;
    s->slots.resize(layout->bindings.size());
    for (size_t i = 0; i < s->slots.size(); ++i) s->slots[i].kind = layout->bindings[i].kind;
    s->arena = arena;
    if (arena) arena->sets.push_back(s);
    return s;
}
static Slot* slot_for(GPU_DescriptorSet* set, GPU_Binding b, BindKind kind, const char* fn) {
    GPU_REQUIRE(set, nullptr, "%s: NULL descriptor set", fn);
    GPU_REQUIRE(!set->finalized, nullptr, "%s: descriptor set already finalised", fn);
    GPU_REQUIRE(b < set->slots.size(), nullptr, "%s: binding %u out of range", fn, b);
    GPU_REQUIRE(set->slots[b].kind == kind, nullptr, "%s: binding %u (\"%s\") has a different kind", fn, b, set->layout->bindings[b].name.c_str());
    return &set->slots[b];
}
GPU_API void GPU_SetTextureBinding(GPU_DescriptorSet* set, GPU_Binding b, GPU_Texture* v) {
    Slot* s = slot_for(set, b, Bind_Texture, __func__); if (!s) return;
    GPU_REQUIRE_V(v, "GPU_SetTextureBinding: NULL texture");
    s->tex = (TextureImpl*)v; s->whole = true; s->mip = 0; s->set = true;
}
GPU_API void GPU_SetTextureMipBinding(GPU_DescriptorSet* set, GPU_Binding b, GPU_Texture* v, uint32_t mip) {
    Slot* s = slot_for(set, b, Bind_Texture, __func__); if (!s) return;
    GPU_REQUIRE_V(v && mip < v->mip_level_count, "GPU_SetTextureMipBinding: bad texture / mip");
    GPU_REQUIRE_V(v->flags & GPU_TextureFlag_PerMipBinding, "GPU_SetTextureMipBinding: texture lacks GPU_TextureFlag_PerMipBinding");
    s->tex = (TextureImpl*)v; s->whole = false; s->mip = mip; s->set = true;
}
GPU_API void GPU_SetSamplerBinding(GPU_DescriptorSet* set, GPU_Binding b, GPU_Sampler* v) {
    Slot* s = slot_for(set, b, Bind_Sampler, __func__); if (!s) return;
    GPU_REQUIRE_V(v, "GPU_SetSamplerBinding: NULL sampler");
    s->sampler = v; s->set = true;
}
GPU_API void GPU_SetBufferBinding(GPU_DescriptorSet* set, GPU_Binding b, GPU_Buffer* v) {
    Slot* s = slot_for(set, b, Bind_Buffer, __func__); if (!s) return;
    GPU_REQUIRE_V(v, "GPU_SetBufferBinding: NULL buffer");
    s->buf = (BufferImpl*)v; s->set = true;
}
GPU_API void GPU_SetStorageImageBinding(GPU_DescriptorSet* set, GPU_Binding b, GPU_Texture* v, uint32_t mip) {
    Slot* s = slot_for(set, b, Bind_StorageImage, __func__); if (!s) return;
    GPU_REQUIRE_V(v && mip < v->mip_level_count, "GPU_SetStorageImageBinding: bad texture / mip level %u", mip);
    GPU_REQUIRE_V(v->flags & GPU_TextureFlag_StorageImage, "GPU_SetStorageImageBinding: texture lacks GPU_TextureFlag_StorageImage");
    s->tex = (TextureImpl*)v; s->whole = false; s->mip = mip; s->set = true;
}
GPU_API void GPU_FinalizeDescriptorSet(GPU_DescriptorSet* set) {
    GPU_REQUIRE_V(set, "GPU_FinalizeDescriptorSet: NULL set");
    for (size_t i = 0; i < set->slots.size(); ++i)
        GPU_REQUIRE_V(set->slots[i].set, "GPU_FinalizeDescriptorSet: binding %zu (\"%s\") was never set", i, set->layout->bindings[i].name.c_str());
    set->finalized = true;
}
GPU_API void GPU_DestroyDescriptorSet(GPU_DescriptorSet* set) {
    if (!set) return;
    GPU_REQUIRE_V(!set->arena, "GPU_DestroyDescriptorSet: set belongs to an arena");
    delete set;
}

// ------------------------------------------------------------------------------------------
// resources
// ------------------------------------------------------------------------------------------
static inline uint32_t mip_dim(uint32_t d, uint32_t m) { uint32_t v = d >> m; return v < 1 ? 1 : v; }

GPU_API uint64_t GPUX_TextureMipBytes(const GPU_Texture* t, uint32_t mip) {
    if (!t || mip >= t->mip_level_count) return 0;
    GPU_FormatInfo fi = GPUX_GetFormatInfo(t->format);
    uint64_t w = mip_dim(t->width, mip), h = mip_dim(t->height, mip), d = mip_dim(t->depth, mip);
    uint64_t bw = (w + fi.block_extent - 1) / fi.block_extent, bh = (h + fi.block_extent - 1) / fi.block_extent;
    return bw * bh * d * fi.block_size * t->layer_count;
}

static bool is_f4_cube(const TextureImpl* t) {
    return t->base.format == GPU_Format_RGBA32F && (t->base.flags & GPU_TextureFlag_Cubemap) && t->base.width == t->base.height && t->base.depth == 1;
}

static GPU_Texture* make_texture_impl(GPU_Format format, uint32_t width, uint32_t height, uint32_t depth, GPU_TextureFlags flags,
                                      const void* data, void* external, uint64_t external_bytes, const char* fn) {
    GPU_REQUIRE(G.init, nullptr, "%s: GPU_Init has not been called", fn);
    GPU_REQUIRE(width > 0 && height > 0 && depth > 0, nullptr, "%s: zero extent", fn);   // gpu_vulkan.c:1339
    GPU_FormatInfo fi = GPUX_GetFormatInfo(format);
    GPU_REQUIRE(fi.block_size > 0, nullptr, "%s: invalid format %d", fn, (int)format);
    TextureImpl* t = new TextureImpl();
    t->base.width = width; t->base.height = height; t->base.depth = depth;
    t->base.layer_count = (flags & GPU_TextureFlag_Cubemap) ? 6 : 1;
    uint32_t mips = 1;
    if (flags & GPU_TextureFlag_HasMipmaps) {                                  // gpu_vulkan.c:1344-1351
        uint32_t s = width < height ? width : height;
        while (s > 1) { s /= 2; mips++; }
    }
    t->base.mip_level_count = mips;
    t->base.format = format; t->base.flags = flags;
    t->texel_bytes = fi.block_size;
    size_t off = 0;
    for (uint32_t m = 0; m < mips; ++m) { t->mip_offset.push_back(off); off += (size_t)GPUX_TextureMipBytes(&t->base, m); }
    t->bytes = off;
    if (external) {
        if (external_bytes < t->bytes) { gpu_fail("%s: external allocation of %llu bytes is smaller than the %zu the texture needs", fn, (unsigned long long)external_bytes, t->bytes); delete t; return nullptr; }
        t->dev = external; t->owns_memory = false;
    } else {
        hipError_t e = hipMalloc(&t->dev, t->bytes);
        if (e != hipSuccess) { gpu_fail("%s: hipMalloc(%zu) failed: %s", fn, t->bytes, hipGetErrorString(e)); delete t; return nullptr; }
        HIP_OK(hipMemset(t->dev, 0, t->bytes));
    }
    if (data) {
        HIP_OK(hipMemcpy(t->dev, data, (size_t)GPUX_TextureMipBytes(&t->base, 0), hipMemcpyHostToDevice));
        if (mips > 1) {                                                        // gpu_vulkan.c:1444-1446
            if (!is_f4_cube(t)) {
                gpu_fail("%s: mip generation is implemented for RGBA32F cubemaps only", fn);
            } else {
                int rc = pbrk_mip_chain(t->dev, (int)width, (int)mips, nullptr);
                if (rc != PBRK_OK) gpu_fail("%s: mip chain kernel failed (%d)", fn, rc);
                HIP_OK(hipStreamSynchronize(nullptr));                          // :1448-1449 blocking
            }
        }
    }
    return &t->base;
}

GPU_API GPU_Texture* GPU_MakeTexture(GPU_Format format, uint32_t width, uint32_t height, uint32_t depth, GPU_TextureFlags flags, const void* data) {
    return make_texture_impl(format, width, height, depth, flags, data, nullptr, 0, __func__);
}
GPU_API GPU_Texture* GPUX_MakeTextureExternal(GPU_Format format, uint32_t width, uint32_t height, uint32_t depth, GPU_TextureFlags flags,
                                              void* device_memory, uint64_t device_bytes) {
    GPU_REQUIRE(device_memory, nullptr, "GPUX_MakeTextureExternal: NULL device memory");
    return make_texture_impl(format, width, height, depth, flags, nullptr, device_memory, device_bytes, __func__);
}
GPU_API GPU_Texture* GPUX_MakeCubemapFromEquirect(const void* rgba32f, uint32_t width, uint32_t height, uint32_t face_size, GPU_TextureFlags extra_flags) {
    GPU_REQUIRE(rgba32f && width > 0 && height > 0 && face_size > 0, nullptr, "GPUX_MakeCubemapFromEquirect: bad arguments");
    GPU_Texture* tex = make_texture_impl(GPU_Format_RGBA32F, face_size, face_size, 1, GPU_TextureFlag_Cubemap | GPU_TextureFlag_HasMipmaps | extra_flags,
                                         nullptr, nullptr, 0, __func__);
    if (!tex) return nullptr;
    TextureImpl* t = (TextureImpl*)tex;
    void* pano = nullptr;
    size_t bytes = (size_t)width * height * 16;
    hipError_t e = hipMalloc(&pano, bytes);
    if (e != hipSuccess) { gpu_fail("GPUX_MakeCubemapFromEquirect: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); GPU_DestroyTexture(tex); return nullptr; }
    HIP_OK(hipMemcpy(pano, rgba32f, bytes, hipMemcpyHostToDevice));
    int rc = pbrk_equirect_to_cube(pano, (int)width, (int)height, t->dev, (int)face_size, nullptr);
    if (rc == PBRK_OK) rc = pbrk_mip_chain(t->dev, (int)face_size, (int)tex->mip_level_count, nullptr);
    HIP_OK(hipStreamSynchronize(nullptr));
    (void)hipFree(pano);
    if (rc != PBRK_OK) { gpu_fail("GPUX_MakeCubemapFromEquirect: conversion kernels failed (%d)", rc); GPU_DestroyTexture(tex); return nullptr; }
    return tex;
}
GPU_API uint64_t GPUX_TextureTotalBytes(const GPU_Texture* t) { return t ? ((const TextureImpl*)t)->bytes : 0; }
GPU_API uint64_t GPUX_TextureMipOffset(const GPU_Texture* t, uint32_t mip) {
    return (t && mip < t->mip_level_count) ? ((const TextureImpl*)t)->mip_offset[mip] : 0;
}

GPU_API void GPU_DestroyTexture(GPU_Texture* tex) {
    if (!tex) return;
    TextureImpl* t = (TextureImpl*)tex;
    if (t->owns_memory) (void)hipFree(t->dev);
    if (t->bordered) (void)hipFree(t->bordered);
    if (t->cells) (void)hipFree(t->cells);
    if (t->range_dev) (void)hipFree(t->range_dev);
    if (t->lut_cells) (void)hipFree(t->lut_cells);
    delete t;
}

GPU_API GPU_Buffer* GPU_MakeBuffer(uint32_t size, GPU_BufferFlags flags, const void* data) {
    GPU_REQUIRE(G.init, nullptr, "GPU_MakeBuffer: GPU_Init has not been called");
    GPU_REQUIRE(size > 0, nullptr, "GPU_MakeBuffer: zero size");
    BufferImpl* b = new BufferImpl();
    b->base.flags = flags; b->base.size = size; b->base.data = nullptr;
    hipError_t e;
    if (flags & GPU_BufferFlag_CPU) {                                          // persistently mapped (gpu_vulkan.c:1248-1250)
        e = hipHostMalloc(&b->dev, size, hipHostMallocDefault);
        b->pinned_host = true;
        b->base.data = b->dev;
    } else {
        e = hipMalloc(&b->dev, size);
    }
    if (e != hipSuccess) { gpu_fail("GPU_MakeBuffer: allocation of %u bytes failed: %s", size, hipGetErrorString(e)); delete b; return nullptr; }
    if (data) {
        if (b->pinned_host) memcpy(b->dev, data, size);
        else HIP_OK(hipMemcpy(b->dev, data, size, hipMemcpyHostToDevice));
    }
    return &b->base;
}

GPU_API void GPU_DestroyBuffer(GPU_Buffer* buf) {
    if (!buf) return;
    BufferImpl* b = (BufferImpl*)buf;
    if (b->pinned_host) (void)hipHostFree(b->dev); else (void)hipFree(b->dev);
    delete b;
}

GPU_API void* GPUX_TextureDevicePtr(GPU_Texture* tex, uint32_t mip) {
    GPU_REQUIRE(tex && mip < tex->mip_level_count, nullptr, "GPUX_TextureDevicePtr: bad texture / mip");
    TextureImpl* t = (TextureImpl*)tex;
    return (char*)t->dev + t->mip_offset[mip];
}
GPU_API void GPUX_InvalidateTexture(GPU_Texture* tex) {
    GPU_REQUIRE_V(tex, "GPUX_InvalidateTexture: NULL texture");
    TextureImpl* t = (TextureImpl*)tex;
    t->bordered_valid = false; t->lut_cells_valid = false;          // ensure_bordered() drops the cells twin with the apron
    for (char& v : t->cells_valid) v = 0;
}
GPU_API void* GPUX_BufferDevicePtr(GPU_Buffer* buf) { return buf ? ((BufferImpl*)buf)->dev : nullptr; }

// ------------------------------------------------------------------------------------------
// pipelines: shader identity -> built-in kernel
// ------------------------------------------------------------------------------------------
static std::string basename_of(GPU_String path) {
    std::string s(path.data ? path.data : "", path.data ? path.length : 0);
    size_t p = s.find_last_of("/\\");
    return p == std::string::npos ? s : s.substr(p + 1);
}
static bool glsl_contains(GPU_String glsl, const char* needle) {
    if (!glsl.data || glsl.length == 0) return true;        // no text given: trust the file name
    std::string s(glsl.data, glsl.length);
    return s.find(needle) != std::string::npos;
}
static KernelId identify_shader(const GPU_ShaderDesc* d) {
    if (d->spirv.data && d->spirv.length) {
        std::string t(d->spirv.data, d->spirv.length);
        if (t == kTokenLut) return Kernel_BrdfLut;
        if (t == kTokenIrr) return Kernel_Irradiance;
        if (t == kTokenPre) return Kernel_Prefilter;
        if (t == kTokenLit) return Kernel_Lighting;
        if (t == kTokenSweep) return Kernel_LightgridSweep;
        if (t == kTokenTaa) return Kernel_TaaResolve;
        if (t == kTokenFinal) return Kernel_FinalPost;
        if (t == kTokenBloomDown) return Kernel_BloomDown;
        if (t == kTokenBloomUp) return Kernel_BloomUp;
        return Kernel_None;
    }
    std::string b = basename_of(d->glsl_debug_filepath);
    // file name first; the GLSL text (when present) must carry the entry's signature symbols
    if (b == "gen_brdf_integration_map.glsl" && glsl_contains(d->glsl, "GeometryMikkelsen") && glsl_contains(d->glsl, "image2D OUTPUT")) return Kernel_BrdfLut;
    if (b == "gen_irradiance_map.glsl" && glsl_contains(d->glsl, "CubemapSampleDirFromFaceUV") && glsl_contains(d->glsl, "TEX_ENV_CUBE")) return Kernel_Irradiance;
    if (b == "gen_prefiltered_env_map.glsl" && glsl_contains(d->glsl, "DistributionBeckmann") && glsl_contains(d->glsl, "mip_level")) return Kernel_Prefilter;
    if (b == "lighting_pass.glsl" && glsl_contains(d->glsl, "BRDF_INTEGRATION_MAP") && glsl_contains(d->glsl, "GBUFFER_DEPTH")) return Kernel_Lighting;
    if (b == "taa_resolve.glsl" && glsl_contains(d->glsl, "SampleHistoryTextureCatmullRom") && glsl_contains(d->glsl, "GBUFFER_VELOCITY_PREV")) return Kernel_TaaResolve;
    if (b == "final_post_process.glsl" && glsl_contains(d->glsl, "aces_approx") && glsl_contains(d->glsl, "BLOOM_RESULT")) return Kernel_FinalPost;
    if (b == "bloom_downsample.glsl" && glsl_contains(d->glsl, "BLOOM_INPUT") && glsl_contains(d->glsl, "dst_mip_level")) return Kernel_BloomDown;
    if (b == "bloom_upsample.glsl" && glsl_contains(d->glsl, "BLOOM_INPUT") && glsl_contains(d->glsl, "radius")) return Kernel_BloomUp;
    if (b == "lightgrid_sweep.glsl" && glsl_contains(d->glsl, "LIGHTMAP_IMG") && glsl_contains(d->glsl, "X_direction")) return Kernel_LightgridSweep;
    return Kernel_None;
}
static GPU_String token_for(KernelId k) {
    switch (k) {
    case Kernel_BrdfLut: return GPU_String{kTokenLut, sizeof kTokenLut - 1};
    case Kernel_Irradiance: return GPU_String{kTokenIrr, sizeof kTokenIrr - 1};
    case Kernel_Prefilter: return GPU_String{kTokenPre, sizeof kTokenPre - 1};
    case Kernel_Lighting: return GPU_String{kTokenLit, sizeof kTokenLit - 1};
    case Kernel_LightgridSweep: return GPU_String{kTokenSweep, sizeof kTokenSweep - 1};
    case Kernel_TaaResolve: return GPU_String{kTokenTaa, sizeof kTokenTaa - 1};
    case Kernel_FinalPost: return GPU_String{kTokenFinal, sizeof kTokenFinal - 1};
    case Kernel_BloomDown: return GPU_String{kTokenBloomDown, sizeof kTokenBloomDown - 1};
    case Kernel_BloomUp: return GPU_String{kTokenBloomUp, sizeof kTokenBloomUp - 1};
    default: return GPU_String{nullptr, 0};
    }
}

static GPU_GLSLError g_last_error;
static char g_last_error_text[512];

GPU_API GPU_String GPU_SPIRVFromGLSL(DS_Arena* arena, GPU_ShaderStage stage, GPU_PipelineLayout* layout, const GPU_ShaderDesc* desc, GPU_GLSLErrorArray* out_errors) {
    (void)arena; (void)layout;
    GPU_String empty = {nullptr, 0};
    GPU_REQUIRE(desc, empty, "GPU_SPIRVFromGLSL: NULL shader desc");
    GPU_ShaderDesc probe = *desc;
    probe.spirv = empty;
    KernelId k = identify_shader(&probe);
    bool stage_ok = (k == Kernel_Lighting || k == Kernel_TaaResolve || k == Kernel_FinalPost || k == Kernel_BloomDown || k == Kernel_BloomUp) ? (stage == GPU_ShaderStage_Vertex || stage == GPU_ShaderStage_Fragment)
                                           : (stage == GPU_ShaderStage_Compute);
    if (k != Kernel_None && stage_ok) {
        if (out_errors) { out_errors->data = nullptr; out_errors->length = 0; }
        return token_for(k);
    }
    snprintf(g_last_error_text, sizeof g_last_error_text,
             "the HIP backend has no built-in kernel for shader \"%s\" (stage %d); supported: gen_brdf_integration_map.glsl, "
             "gen_irradiance_map.glsl, gen_prefiltered_env_map.glsl, lightgrid_sweep.glsl (compute), lighting_pass.glsl, taa_resolve.glsl, bloom_downsample.glsl, bloom_upsample.glsl, final_post_process.glsl (full-screen)",
             basename_of(desc->glsl_debug_filepath).c_str(), (int)stage);
    if (!out_errors) { gpu_fail("GPU_SPIRVFromGLSL: %s", g_last_error_text); return empty; }
    g_last_error.shader_stage = stage; g_last_error.line = 0;
    g_last_error.error_message = GPU_String{g_last_error_text, strlen(g_last_error_text)};
    out_errors->data = &g_last_error; out_errors->length = 1;
    return empty;
}

GPU_API GPU_String GPU_JoinGLSLErrorString(DS_Arena* arena, GPU_GLSLErrorArray errors) {
    (void)arena;
    static std::string joined;
    joined.clear();
    for (uint32_t i = 0; i < errors.length; ++i) {
        char head[64];
        snprintf(head, sizeof head, "line %u: ", errors.data[i].line);
        joined += head;
        joined.append(errors.data[i].error_message.data, errors.data[i].error_message.length);
        joined += "\n";
    }
    return GPU_String{joined.c_str(), joined.size()};
}

GPU_API GPU_ComputePipeline* GPU_MakeComputePipeline(GPU_PipelineLayout* layout, const GPU_ShaderDesc* cs) {
    GPU_REQUIRE(layout && layout->finalized && cs, nullptr, "GPU_MakeComputePipeline: NULL / unfinalised argument");
    KernelId k = identify_shader(cs);
    GPU_REQUIRE(k == Kernel_BrdfLut || k == Kernel_Irradiance || k == Kernel_Prefilter || k == Kernel_LightgridSweep, nullptr,
                "GPU_MakeComputePipeline: shader \"%s\" has no built-in HIP kernel", basename_of(cs->glsl_debug_filepath).c_str());
    if (k == Kernel_LightgridSweep) {                                          // lightgrid_sweep.glsl:3 GPU_BINDING(IMG0) image3D
        int b = find_binding(layout, "IMG0");
        GPU_REQUIRE(b >= 0 && layout->bindings[b].kind == Bind_StorageImage, nullptr, "GPU_MakeComputePipeline: layout has no \"IMG0\" storage image");
        GPU_ComputePipeline* p = new GPU_ComputePipeline();
        p->layout = layout; p->kernel = k;
        return p;
    }
    GPU_REQUIRE(find_binding(layout, "OUTPUT") >= 0, nullptr, "GPU_MakeComputePipeline: layout has no \"OUTPUT\" storage image");
    if (k != Kernel_BrdfLut)
        GPU_REQUIRE(find_binding(layout, "TEX_ENV_CUBE") >= 0, nullptr, "GPU_MakeComputePipeline: layout has no \"TEX_ENV_CUBE\" texture");
    GPU_ComputePipeline* p = new GPU_ComputePipeline();
    p->layout = layout; p->kernel = k;
    return p;
}
GPU_API void GPU_DestroyComputePipeline(GPU_ComputePipeline* p) { delete p; }

GPU_API GPU_RenderPass* GPU_MakeRenderPass(const GPU_RenderPassDesc* desc) {
    GPU_REQUIRE(desc, nullptr, "GPU_MakeRenderPass: NULL desc");
    GPU_REQUIRE(desc->color_targets != GPU_SWAPCHAIN_COLOR_TARGET, nullptr, "GPU_MakeRenderPass: swapchain targets are unsupported (headless backend)");
    GPU_REQUIRE(desc->msaa_color_resolve_targets == nullptr, nullptr, "GPU_MakeRenderPass: MSAA is unsupported (raster)");
    GPU_RenderPass* rp = new GPU_RenderPass();
    rp->desc = *desc;
    for (uint32_t i = 0; i < desc->color_targets_count; ++i) rp->targets.push_back(desc->color_targets[i]);
    rp->desc.color_targets = rp->targets.data();
    if ((rp->desc.width == 0 || rp->desc.height == 0) && !rp->targets.empty() && rp->targets[0].texture) {   // size left to the target (the swapchain pass, render.cpp:782-785)
        rp->desc.width = mip_dim(rp->targets[0].texture->width, rp->targets[0].mip_level);
        rp->desc.height = mip_dim(rp->targets[0].texture->height, rp->targets[0].mip_level);
    }
    return rp;
}
GPU_API void GPU_DestroyRenderPass(GPU_RenderPass* rp) { delete rp; }

GPU_API GPU_GraphicsPipeline* GPU_MakeGraphicsPipeline(const GPU_GraphicsPipelineDesc* desc) {
    GPU_REQUIRE(desc && desc->layout && desc->render_pass, nullptr, "GPU_MakeGraphicsPipeline: NULL argument");
    KernelId k = identify_shader(&desc->fs);
    GPU_REQUIRE(k == Kernel_Lighting || k == Kernel_TaaResolve || k == Kernel_FinalPost || k == Kernel_BloomDown || k == Kernel_BloomUp, nullptr,
                "GPU_MakeGraphicsPipeline: unsupported (raster): only the full-screen lighting_pass / taa_resolve / final_post_process pipelines have HIP kernels (got \"%s\")",
                basename_of(desc->fs.glsl_debug_filepath).c_str());
    GPU_REQUIRE(desc->vertex_input_formats_count == 0, nullptr, "GPU_MakeGraphicsPipeline: unsupported (raster): vertex inputs");
    GPU_REQUIRE(desc->render_pass->desc.color_targets_count == 1, nullptr, "GPU_MakeGraphicsPipeline: full-screen passes have exactly one colour target");
    GPU_GraphicsPipeline* p = new GPU_GraphicsPipeline();
    p->layout = desc->layout; p->pass = desc->render_pass; p->kernel = k; p->shade_flags = GPUX_Shade_IBL;
    if (desc->enable_blending) {                                              // gpu_vulkan.c:1828-1842
        if (!(desc->blending_mode_additive && (k == Kernel_BloomDown || k == Kernel_BloomUp))) {
            gpu_fail("GPU_MakeGraphicsPipeline: blending is implemented for the additive bloom passes only"); delete p; return nullptr;
        }
        p->blend_additive = true;
    }
    return p;
}
GPU_API void GPU_DestroyGraphicsPipeline(GPU_GraphicsPipeline* p) { delete p; }
GPU_API void GPUX_SetShadeFlags(GPU_GraphicsPipeline* p, int flags) {
    GPU_REQUIRE_V(p, "GPUX_SetShadeFlags: NULL pipeline");
    p->shade_flags = flags;
}

// ------------------------------------------------------------------------------------------
// host tables -> device (cached)
// ------------------------------------------------------------------------------------------
static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static DeviceTable* get_table(int kind, int n, float roughness, int aux) {
    TableKey key = {kind, n, aux, fbits(roughness)};
    auto it = G.tables.find(key);
    if (it != G.tables.end()) return &it->second;
    DeviceTable t;
    std::vector<float> host;
    if (kind == 0) {            // prefilter (lx,ly,lz,w)
        host.resize((size_t)n * 4);
        t.count = pbrk_host_prefilter_table(n, roughness, host.data(), &t.alpha);
        t.weights.resize((size_t)(t.count > 0 ? t.count : 0));
        for (int i = 0; i < t.count; ++i) t.weights[(size_t)i] = host[(size_t)i * 4 + 3];
    } else if (kind == 1) {     // irradiance
        host.resize((size_t)n * 4);
        t.count = pbrk_host_irradiance_table(n, host.data());
    } else if (kind == 2) {     // LUT sample angles
        host.resize((size_t)n * 4);
        pbrk_host_sample_angles(n, host.data());
        t.count = n;
    } else {                    // LUT per-column view angle: V = Rotate(N, X, acos(NdotV)), gen_brdf_integration_map.glsl:154-160
        host.resize((size_t)n * 2);
        for (int x = 0; x < n; ++x) {
            float ndv = ((float)x + 0.5f) / (float)n;
            float th = acosf(ndv);
            host[2 * x] = cosf(th); host[2 * x + 1] = sinf(th);
        }
        t.count = n;
    }
    size_t bytes = host.size() * sizeof(float);
    if (bytes < 16) bytes = 16;
    hipError_t e = hipMalloc(&t.dev, bytes);
    if (e != hipSuccess) { gpu_fail("table allocation failed: %s", hipGetErrorString(e)); return nullptr; }
    HIP_OK(hipMemcpy(t.dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    auto ins = G.tables.insert({key, t});
    return &ins.first->second;
}

// ------------------------------------------------------------------------------------------
// graphs
// ------------------------------------------------------------------------------------------
GPU_API GPU_Graph* GPU_MakeGraph(void) {
    GPU_REQUIRE(G.init, nullptr, "GPU_MakeGraph: GPU_Init has not been called");
    GPU_Graph* g = new GPU_Graph();
    HIP_OK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    return g;
}
GPU_API void GPU_MakeSwapchainGraphs(uint32_t count, GPU_Graph** out) {
    for (uint32_t i = 0; i < count; ++i) out[i] = GPU_MakeGraph();
}
GPU_API GPU_Texture* GPU_GetBackbuffer(GPU_Graph* g) { (void)g; return nullptr; }
GPU_API void* GPUX_GraphStream(GPU_Graph* g) { return g ? (void*)g->stream : nullptr; }

static void reset_graph(GPU_Graph* g) {
    g->ops.clear();
    g->submitted = false;
    g->bound_cpipe = nullptr; g->bound_cset = nullptr; g->push_size = 0;
    g->preparing = nullptr; g->in_pass = nullptr; g->draw_params.clear(); g->bound_draw = -1;
}
GPU_API void GPU_DestroyGraph(GPU_Graph* g) {
    GPU_REQUIRE_V(g, "GPU_DestroyGraph: NULL graph");            // the reference does not accept NULL here (gpu_vulkan.c:2393-2404)
    (void)hipStreamSynchronize(g->stream);
    for (hipStream_t s : g->side) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->sync_ev) (void)hipEventDestroy(e);
    if (g->order_ev) (void)hipEventDestroy(g->order_ev);
    if (g->mid_ev) (void)hipEventDestroy(g->mid_ev);
    if (g->span_a) (void)hipEventDestroy(g->span_a);
    if (g->span_b) (void)hipEventDestroy(g->span_b);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (G.last_submitted == g) G.last_submitted = nullptr;        // idle by contract (gpu.h:453): nothing left to order against
    (void)hipStreamDestroy(g->stream);
    delete g;
}

#define REC_GUARD(g) GPU_REQUIRE_V((g) && !(g)->submitted, "%s: graph is NULL or already submitted (call GPU_GraphWait first)", __func__)

GPU_API void GPU_OpBindComputePipeline(GPU_Graph* g, GPU_ComputePipeline* p) { REC_GUARD(g); g->bound_cpipe = p; }
GPU_API void GPU_OpBindComputeDescriptorSet(GPU_Graph* g, GPU_DescriptorSet* s) {
    REC_GUARD(g);
    GPU_REQUIRE_V(s && s->finalized, "GPU_OpBindComputeDescriptorSet: set is NULL or not finalised");
    g->bound_cset = s;
}
static void push_constants(GPU_Graph* g, void* data, uint32_t size, const char* fn) {
    GPU_REQUIRE_V(g && !g->submitted, "%s: graph is NULL or already submitted", fn);
    GPU_REQUIRE_V(size <= 128 && (data || size == 0), "%s: at most 128 bytes of push constants (gpu_vulkan.c:710)", fn);
    memcpy(g->push, data, size);
    g->push_size = size;
}
GPU_API void GPU_OpPushComputeConstants(GPU_Graph* g, GPU_PipelineLayout* l, void* data, uint32_t size) { (void)l; push_constants(g, data, size, __func__); }
GPU_API void GPU_OpPushGraphicsConstants(GPU_Graph* g, GPU_PipelineLayout* l, void* data, uint32_t size) { (void)l; push_constants(g, data, size, __func__); }

static Slot* named_slot(GPU_DescriptorSet* set, const char* name) {
    int b = find_binding(set->layout, name);
    return b < 0 ? nullptr : &set->slots[b];
}

// Validates a dispatch against the bound pipeline/set at record time, so that nothing can fault at launch.
static bool record_dispatch(GPU_Graph* g, Op& op, const char* fn) {
    GPU_REQUIRE(g->bound_cpipe, false, "%s: no compute pipeline bound", fn);
    GPU_REQUIRE(g->bound_cset, false, "%s: no compute descriptor set bound", fn);
    GPU_REQUIRE(g->bound_cset->layout == g->bound_cpipe->layout, false, "%s: descriptor set and pipeline use different layouts", fn);
    op.kind = Op_Dispatch;
    op.cpipe = g->bound_cpipe; op.set = g->bound_cset;
    memcpy(op.push, g->push, g->push_size); op.push_size = g->push_size;
    if (op.cpipe->kernel == Kernel_LightgridSweep) {
        Slot* img = named_slot(op.set, "IMG0");
        GPU_REQUIRE(img && img->tex, false, "%s: IMG0 is not bound", fn);
        const GPU_Texture* t = &img->tex->base;
        GPU_REQUIRE(t->format == GPU_Format_RGBA16F && t->layer_count == 1 && img->mip == 0, false, "%s: IMG0 must be mip 0 of an RGBA16F 3-D image", fn);
        GPU_REQUIRE(op.push_size >= 4, false, "%s: the sweep needs its `int X_direction` push constant (lightgrid_sweep.glsl:5-7)", fn);
        return true;
    }
    Slot* out = named_slot(op.set, "OUTPUT");
    GPU_REQUIRE(out && out->tex, false, "%s: OUTPUT is not bound", fn);
    TextureImpl* ot = out->tex;
    if (op.cpipe->kernel == Kernel_BrdfLut) {
        GPU_REQUIRE(ot->base.layer_count == 1 && ot->base.width == ot->base.height, false, "%s: BRDF LUT output must be a square 2D image", fn);
        GPU_REQUIRE(ot->base.format == GPU_Format_RG16F || ot->base.format == GPU_Format_RG32F || ot->base.format == GPU_Format_RGBA32F, false,
                    "%s: BRDF LUT output format must be RG16F, RG32F or RGBA32F", fn);
    } else {
        GPU_REQUIRE(is_f4_cube(ot), false, "%s: OUTPUT must be a square RGBA32F cubemap", fn);
        Slot* env = named_slot(op.set, "TEX_ENV_CUBE");
        GPU_REQUIRE(env && env->tex && is_f4_cube(env->tex), false, "%s: TEX_ENV_CUBE must be a square RGBA32F cubemap", fn);
        GPU_REQUIRE(env->tex != ot, false, "%s: TEX_ENV_CUBE and OUTPUT alias", fn);
    }
    return true;
}

// Sweep invocations (iy, iz) in [y0,y1) x [z0,z1): checks that every voxel they touch lies inside the image
// (direction 0 -> (x, iy, iz); 1 -> (iz, x, iy); 2 -> (iy, iz, x) with x < 128; lightgrid_sweep.glsl:10-22).
// The op keeps the ranges in row0/row1 (iy) and face0/face1 (iz).
static bool record_sweep_lines(GPU_Graph* g, Op& op, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, const char* fn) {
    (void)g;
    const GPU_Texture* t = &named_slot(op.set, "IMG0")->tex->base;
    int32_t dir; memcpy(&dir, op.push, 4);
    if (dir < 0 || dir > 2) dir = 2;                                          // the shader's final `else` (:19)
    uint32_t line_extent = dir == 0 ? t->width : (dir == 1 ? t->height : t->depth);
    uint32_t y_extent = dir == 0 ? t->height : (dir == 1 ? t->depth : t->width);
    uint32_t z_extent = dir == 0 ? t->depth : (dir == 1 ? t->width : t->height);
    GPU_REQUIRE(y0 < y1 && z0 < z1, false, "%s: empty line range", fn);
    GPU_REQUIRE(line_extent >= PBRK_SWEEP_LEN && y1 <= y_extent && z1 <= z_extent, false,
                "%s: sweep direction %d over lines [%u,%u) x [%u,%u) leaves the %ux%ux%u image", fn, (int)dir, y0, y1, z0, z1, t->width, t->height, t->depth);
    int32_t d32 = dir; memcpy(op.push, &d32, 4);
    op.row0 = y0; op.row1 = y1; op.face0 = z0; op.face1 = z1;
    return true;
}

GPU_API void GPUX_OpDispatchLines(GPU_Graph* g, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPUX_OpDispatchLines: inside a render pass");
    Op op;
    if (!record_dispatch(g, op, __func__)) return;
    GPU_REQUIRE_V(op.cpipe->kernel == Kernel_LightgridSweep, "GPUX_OpDispatchLines: the bound pipeline is not the light-grid sweep");
    if (!record_sweep_lines(g, op, y0, y1, z0, z1, __func__)) return;
    op.rows_explicit = true;
    g->ops.push_back(op);
}

GPU_API void GPU_OpDispatch(GPU_Graph* g, uint32_t gx, uint32_t gy, uint32_t gz) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPU_OpDispatch: inside a render pass");
    GPU_REQUIRE_V(gx > 0 && gy > 0 && gz > 0, "GPU_OpDispatch: zero group count");
    Op op;
    if (!record_dispatch(g, op, __func__)) return;
    op.gx = gx; op.gy = gy; op.gz = gz;
    if (op.cpipe->kernel == Kernel_LightgridSweep) {
        // local size 1x8x8 (lightgrid_sweep.glsl:1); invocation x > 0 would redo line x = 0 (base_coord.x is the constant 0, :11)
        GPU_REQUIRE_V(gx == 1, "GPU_OpDispatch: the sweep is dispatched as (1, gy, gz) (render.cpp:1072)");
        if (!record_sweep_lines(g, op, 0, gy * 8, 0, gz * 8, __func__)) return;
        g->ops.push_back(op);
        return;
    }
    Slot* out = named_slot(op.set, "OUTPUT");
    uint32_t size = mip_dim(out->tex->base.width, out->mip);
    // local size is 8x8x6 (shader line 1): the kernels work on whole rows, so the x extent must cover the image
    GPU_REQUIRE_V(gx * 8 >= size, "GPU_OpDispatch: partial-width dispatch (%u groups over %u texels) is unsupported", gx, size);
    op.face0 = 0; op.face1 = out->tex->base.layer_count;
    op.row0 = 0; op.row1 = gy * 8 < size ? gy * 8 : size;
    g->ops.push_back(op);
}

GPU_API void GPUX_OpDispatchRows(GPU_Graph* g, uint32_t face0, uint32_t face1, uint32_t row0, uint32_t row1) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPUX_OpDispatchRows: inside a render pass");
    Op op;
    if (!record_dispatch(g, op, __func__)) return;
    GPU_REQUIRE_V(op.cpipe->kernel != Kernel_LightgridSweep, "GPUX_OpDispatchRows: use GPUX_OpDispatchLines for the light-grid sweep");
    Slot* out = named_slot(op.set, "OUTPUT");
    uint32_t size = mip_dim(out->tex->base.width, out->mip);
    GPU_REQUIRE_V(face0 < face1 && face1 <= out->tex->base.layer_count && row0 < row1 && row1 <= size,
                  "GPUX_OpDispatchRows: range faces [%u,%u) rows [%u,%u) outside a %u-layer %ux%u image", face0, face1, row0, row1,
                  out->tex->base.layer_count, size, size);
    op.face0 = face0; op.face1 = face1; op.row0 = row0; op.row1 = row1; op.rows_explicit = true;
    g->ops.push_back(op);
}

// ---- render-pass vocabulary (lighting pass only) ----
GPU_API void GPU_OpPrepareRenderPass(GPU_Graph* g, GPU_RenderPass* rp) {
    REC_GUARD(g);
    GPU_REQUIRE_V(rp && g->preparing == nullptr && g->in_pass == nullptr, "GPU_OpPrepareRenderPass: bad state");   // gpu_vulkan.c:2957
    g->preparing = rp;
    g->draw_params.clear();
}
GPU_API uint32_t GPU_OpPrepareDrawParams(GPU_Graph* g, GPU_GraphicsPipeline* p, GPU_DescriptorSet* s) {
    GPU_REQUIRE(g && g->preparing, 0, "GPU_OpPrepareDrawParams: no render pass is being prepared");              // :2964
    GPU_REQUIRE(p && s && s->finalized, 0, "GPU_OpPrepareDrawParams: NULL pipeline / unfinalised set");
    g->draw_params.push_back({p, s});
    return (uint32_t)g->draw_params.size() - 1;
}
GPU_API void GPU_OpBeginRenderPass(GPU_Graph* g) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->preparing, "GPU_OpBeginRenderPass: GPU_OpPrepareRenderPass was not called");
    g->in_pass = g->preparing; g->preparing = nullptr; g->bound_draw = -1;
}
GPU_API void GPU_OpEndRenderPass(GPU_Graph* g) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass, "GPU_OpEndRenderPass: not inside a render pass");
    g->in_pass = nullptr; g->bound_draw = -1;
}
GPU_API void GPU_OpBindDrawParams(GPU_Graph* g, uint32_t idx) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass && idx < g->draw_params.size(), "GPU_OpBindDrawParams: bad state / index");
    g->bound_draw = (int)idx;
}

static bool check_plane(Slot* s, GPU_Format f, uint32_t w, uint32_t h, const char* name) {
    GPU_REQUIRE(s && s->tex, false, "lighting pass: \"%s\" is not bound", name);
    GPU_REQUIRE(s->tex->base.format == f && s->tex->base.width == w && s->tex->base.height == h && s->tex->base.layer_count == 1, false,
                "lighting pass: \"%s\" must be a %ux%u 2D texture of format %d", name, w, h, (int)f);
    return true;
}

static bool check_post_plane(Slot* s, GPU_Format f, uint32_t w, uint32_t h, const char* name, const char* fn) {
    GPU_REQUIRE(s && s->tex, false, "%s: \"%s\" is not bound", fn, name);
    GPU_REQUIRE(s->tex->base.format == f && s->tex->base.layer_count == 1 && s->tex->base.depth == 1 && (w == 0 || (s->tex->base.width == w && s->tex->base.height == h)), false,
                "%s: \"%s\" must be a 2D texture of format %d%s", fn, name, (int)f, w ? " with the pass's size" : "");
    return true;
}

// taa_resolve.glsl / final_post_process.glsl drawn as the full-screen triangle (render.cpp:1131-1137, 1181-1187)
static void record_post(GPU_Graph* g, const DrawParams& dp, uint32_t row0, uint32_t row1, bool explicit_rows, const char* fn) {
    GPU_RenderPass* rp = g->in_pass;
    uint32_t W = rp->desc.width, H = rp->desc.height;
    GPU_REQUIRE_V(rp->targets.size() == 1 && rp->targets[0].texture && rp->targets[0].mip_level < rp->targets[0].texture->mip_level_count, "%s: the pass needs one colour target", fn);
    TextureImpl* target = (TextureImpl*)rp->targets[0].texture;
    const uint32_t tmip = rp->targets[0].mip_level;
    const bool bloom = dp.pipeline->kernel == Kernel_BloomDown || dp.pipeline->kernel == Kernel_BloomUp;
    GPU_REQUIRE_V(bloom || tmip == 0, "%s: only the bloom passes render into mip levels", fn);
    GPU_REQUIRE_V(mip_dim(target->base.width, tmip) == W && mip_dim(target->base.height, tmip) == H && target->base.layer_count == 1 && target->base.depth == 1,
                  "%s: colour target (mip %u) must be %ux%u", fn, tmip, W, H);
    GPU_DescriptorSet* s = dp.set;
    if (bloom) {                                                              // bloom_*.glsl: TEX0 = BLOOM_INPUT (a texture or one of its mips), int dst_mip_level pushed
        GPU_REQUIRE_V(target->base.format == GPU_Format_RGBA16F, "%s: bloom targets are RGBA16F (render.cpp:742-746)", fn);
        Slot* in = named_slot(s, "TEX0");
        GPU_REQUIRE_V(in && in->tex && in->tex->base.format == GPU_Format_RGBA16F && in->tex->base.layer_count == 1 && in->tex->base.depth == 1, "%s: \"TEX0\" must be an RGBA16F 2D texture", fn);
        uint32_t smip = in->whole ? 0 : in->mip;
        GPU_REQUIRE_V(!(in->tex == target && smip == tmip), "%s: bloom pass reads the level it writes", fn);
        GPU_REQUIRE_V(g->push_size >= 4, "%s: the bloom shaders need their `int dst_mip_level` push constant", fn);
        Op op;
        op.kind = Op_Shade; op.gpipe = dp.pipeline; op.set = s; op.pass = rp;
        memcpy(op.push, g->push, g->push_size); op.push_size = g->push_size;
        op.mip = smip; op.mip2 = tmip;
        op.row0 = explicit_rows ? row0 : 0; op.row1 = explicit_rows ? row1 : H;
        GPU_REQUIRE_V(op.row0 < op.row1 && op.row1 <= H, "%s: rows [%u,%u) outside the %u-row pass", fn, op.row0, op.row1, H);
        g->ops.push_back(op);
        return;
    }
    if (dp.pipeline->kernel == Kernel_TaaResolve) {
        GPU_REQUIRE_V(target->base.format == GPU_Format_RGBA16F || target->base.format == GPU_Format_RGBA32F, "%s: TAA target must be RGBA16F/RGBA32F", fn);
        if (!check_post_plane(named_slot(s, "LIGHTING_RESULT"), GPU_Format_RGBA16F, W, H, "LIGHTING_RESULT", fn)) return;
        if (!check_post_plane(named_slot(s, "GBUFFER_DEPTH"), GPU_Format_D32F_Or_X8D24UN, W, H, "GBUFFER_DEPTH", fn)) return;
        if (!check_post_plane(named_slot(s, "GBUFFER_VELOCITY"), GPU_Format_RG16F, W, H, "GBUFFER_VELOCITY", fn)) return;
        if (!check_post_plane(named_slot(s, "GBUFFER_VELOCITY_PREV"), GPU_Format_RG16F, W, H, "GBUFFER_VELOCITY_PREV", fn)) return;
        if (!check_post_plane(named_slot(s, "PREV_FRAME_RESULT"), GPU_Format_RGBA16F, 0, 0, "PREV_FRAME_RESULT", fn)) return;
        GPU_REQUIRE_V(named_slot(s, "PREV_FRAME_RESULT")->tex != target && named_slot(s, "LIGHTING_RESULT")->tex != target,
                      "%s: the TAA target aliases one of its inputs (the reference ping-pongs taa_output_rt, render.cpp:695-697)", fn);
    } else {
        GPU_REQUIRE_V(target->base.format == GPU_Format_RGBA8UN || target->base.format == GPU_Format_BGRA8UN ||
                      target->base.format == GPU_Format_RGBA16F || target->base.format == GPU_Format_RGBA32F, "%s: unsupported target format %d", fn, (int)target->base.format);
        if (!check_post_plane(named_slot(s, "TEX0"), GPU_Format_RGBA16F, 0, 0, "TEX0", fn)) return;     // final_post_process.glsl:28 BLOOM_RESULT
        GPU_REQUIRE_V(named_slot(s, "TEX0")->tex != target, "%s: target aliases TEX0", fn);
    }
    Op op;
    op.kind = Op_Shade;
    op.gpipe = dp.pipeline; op.set = s; op.pass = rp;
    op.row0 = explicit_rows ? row0 : 0; op.row1 = explicit_rows ? row1 : H;
    GPU_REQUIRE_V(op.row0 < op.row1 && op.row1 <= H, "%s: rows [%u,%u) outside the %u-row pass", fn, op.row0, op.row1, H);
    g->ops.push_back(op);
}

static void record_shade(GPU_Graph* g, uint32_t row0, uint32_t row1, bool explicit_rows, const char* fn) {
    GPU_REQUIRE_V(g->in_pass && g->bound_draw >= 0, "%s: no draw params bound inside a render pass", fn);
    DrawParams dp = g->draw_params[g->bound_draw];
    GPU_REQUIRE_V(dp.pipeline->pass == g->in_pass, "%s: pipeline was created for a different render pass", fn);
    GPU_REQUIRE_V(dp.set->layout == dp.pipeline->layout, "%s: descriptor set and pipeline use different layouts", fn);
    GPU_RenderPass* rp = g->in_pass;
    uint32_t W = rp->desc.width, H = rp->desc.height;
    if (dp.pipeline->kernel == Kernel_TaaResolve || dp.pipeline->kernel == Kernel_FinalPost || dp.pipeline->kernel == Kernel_BloomDown || dp.pipeline->kernel == Kernel_BloomUp) {
        record_post(g, dp, row0, row1, explicit_rows, fn);
        return;
    }
    GPU_REQUIRE_V(dp.pipeline->kernel == Kernel_Lighting, "%s: unsupported (raster)", fn);
    GPU_REQUIRE_V(rp->targets.size() == 1 && rp->targets[0].texture, "%s: lighting pass needs one colour target", fn);
    TextureImpl* target = (TextureImpl*)rp->targets[0].texture;
    GPU_REQUIRE_V((target->base.format == GPU_Format_RGBA16F || target->base.format == GPU_Format_RGBA32F) &&
                  target->base.width == W && target->base.height == H, "%s: colour target must be %ux%u RGBA16F/RGBA32F", fn, W, H);
    GPU_DescriptorSet* s = dp.set;
    if (!check_plane(named_slot(s, "GBUFFER_BASE_COLOR"), GPU_Format_RGBA8UN, W, H, "GBUFFER_BASE_COLOR")) return;
    if (!check_plane(named_slot(s, "GBUFFER_NORMAL"), GPU_Format_RGBA8UN, W, H, "GBUFFER_NORMAL")) return;
    if (!check_plane(named_slot(s, "GBUFFER_ORM"), GPU_Format_RGBA8UN, W, H, "GBUFFER_ORM")) return;
    if (!check_plane(named_slot(s, "GBUFFER_EMISSIVE"), GPU_Format_RGBA8UN, W, H, "GBUFFER_EMISSIVE")) return;
    if (!check_plane(named_slot(s, "GBUFFER_DEPTH"), GPU_Format_D32F_Or_X8D24UN, W, H, "GBUFFER_DEPTH")) return;
    Slot* gl = named_slot(s, "GLOBALS");
    GPU_REQUIRE_V(gl && gl->buf && gl->buf->base.size >= 552, "%s: \"GLOBALS\" must be a buffer of at least 552 bytes (render.h:122-136)", fn);
    Slot* pre = named_slot(s, "PREFILTERED_ENV_MAP");
    GPU_REQUIRE_V(pre && pre->tex && is_f4_cube(pre->tex), "%s: \"PREFILTERED_ENV_MAP\" must be a square RGBA32F cubemap", fn);
    if (dp.pipeline->shade_flags & GPUX_Shade_SunShadows) {
        Slot* sun = named_slot(s, "SUN_DEPTH_MAP");
        GPU_REQUIRE_V(sun && sun->tex && sun->tex->base.format == GPU_Format_D32F_Or_X8D24UN && sun->tex->base.layer_count == 1 && sun->tex->base.depth == 1,
                      "%s: \"SUN_DEPTH_MAP\" must be a 2D D32F texture (render.cpp:676)", fn);
    }
    if (dp.pipeline->shade_flags & GPUX_Shade_VoxelGI) {
        Slot* grid = named_slot(s, "LIGHTGRID");
        GPU_REQUIRE_V(grid && grid->tex && grid->tex->base.format == GPU_Format_RGBA16F && grid->tex->base.layer_count == 1 &&
                      grid->tex->base.width == grid->tex->base.height && grid->tex->base.width == grid->tex->base.depth && grid->tex->base.width <= 1024,
                      "%s: \"LIGHTGRID\" must be a cubic RGBA16F 3-D texture (render.cpp:678)", fn);
        Slot* prev = named_slot(s, "PREV_FRAME_RESULT");
        GPU_REQUIRE_V(prev && prev->tex && prev->tex->base.format == GPU_Format_RGBA16F && prev->tex->base.layer_count == 1 && prev->tex->base.depth == 1,
                      "%s: \"PREV_FRAME_RESULT\" must be a 2D RGBA16F texture (render.cpp:862 binds bloom_downscale_rt)", fn);
        GPU_REQUIRE_V(prev->tex != target, "%s: PREV_FRAME_RESULT aliases the colour target", fn);
        Slot* lut = named_slot(s, "BRDF_INTEGRATION_MAP");
        GPU_REQUIRE_V(lut && lut->tex && lut->tex->base.format == GPU_Format_RG16F && lut->tex->base.width == lut->tex->base.height,
                      "%s: \"BRDF_INTEGRATION_MAP\" must be a square RG16F texture", fn);
    }
    if (dp.pipeline->shade_flags & GPUX_Shade_IBL) {
        Slot* irr = named_slot(s, "TEX_IRRADIANCE_MAP");
        GPU_REQUIRE_V(irr && irr->tex && is_f4_cube(irr->tex), "%s: \"TEX_IRRADIANCE_MAP\" must be a square RGBA32F cubemap", fn);
        Slot* lut = named_slot(s, "BRDF_INTEGRATION_MAP");
        GPU_REQUIRE_V(lut && lut->tex && lut->tex->base.format == GPU_Format_RG16F && lut->tex->base.width == lut->tex->base.height,
                      "%s: \"BRDF_INTEGRATION_MAP\" must be a square RG16F texture", fn);
    }
    Op op;
    op.kind = Op_Shade;
    op.gpipe = dp.pipeline; op.set = s; op.pass = rp;
    op.row0 = explicit_rows ? row0 : 0; op.row1 = explicit_rows ? row1 : H;
    GPU_REQUIRE_V(op.row0 < op.row1 && op.row1 <= H, "%s: rows [%u,%u) outside the %u-row pass", fn, op.row0, op.row1, H);
    g->ops.push_back(op);
}

GPU_API void GPU_OpDraw(GPU_Graph* g, uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance) {
    REC_GUARD(g);
    GPU_REQUIRE_V(vertex_count == 3 && instance_count == 1 && first_vertex == 0 && first_instance == 0,
                  "GPU_OpDraw: unsupported (raster): only the full-screen triangle GPU_OpDraw(3,1,0,0) of the lighting pass is implemented");
    record_shade(g, 0, 0, false, __func__);
}
GPU_API void GPUX_OpDrawRows(GPU_Graph* g, uint32_t row0, uint32_t row1) { REC_GUARD(g); record_shade(g, row0, row1, true, __func__); }
GPU_API void GPU_OpDrawIndexed(GPU_Graph* g, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { (void)g; gpu_fail("GPU_OpDrawIndexed: unsupported (raster)"); }
GPU_API void GPU_OpBindVertexBuffer(GPU_Graph* g, GPU_Buffer* b) { (void)g; (void)b; gpu_fail("GPU_OpBindVertexBuffer: unsupported (raster)"); }
GPU_API void GPU_OpBindIndexBuffer(GPU_Graph* g, GPU_Buffer* b) { (void)g; (void)b; gpu_fail("GPU_OpBindIndexBuffer: unsupported (raster)"); }

// ---- transfers ----
GPU_API void GPU_OpCopyBufferToBuffer(GPU_Graph* g, GPU_Buffer* src, GPU_Buffer* dst, uint32_t dst_offset, uint32_t src_offset, uint32_t size) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPU_OpCopyBufferToBuffer: inside a render pass");
    GPU_REQUIRE_V(src && dst && (uint64_t)src_offset + size <= src->size && (uint64_t)dst_offset + size <= dst->size, "GPU_OpCopyBufferToBuffer: range out of bounds");
    Op op; op.kind = Op_CopyB2B; op.name = "copy.buffer_to_buffer";
    op.buf = (BufferImpl*)src; op.buf2 = (BufferImpl*)dst; op.off_a = src_offset; op.off_b = dst_offset; op.size = size;
    g->ops.push_back(op);
}
GPU_API void GPU_OpCopyBufferToTexture(GPU_Graph* g, GPU_Buffer* src, GPU_Texture* dst, uint32_t first_layer, uint32_t layer_count, uint32_t mip) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPU_OpCopyBufferToTexture: inside a render pass");
    GPU_REQUIRE_V(src && dst && mip < dst->mip_level_count && layer_count > 0 && first_layer + layer_count <= dst->layer_count, "GPU_OpCopyBufferToTexture: bad subresource");
    uint64_t per_layer = GPUX_TextureMipBytes(dst, mip) / dst->layer_count;
    GPU_REQUIRE_V(per_layer * layer_count <= src->size, "GPU_OpCopyBufferToTexture: buffer too small (%u < %llu)", src->size, (unsigned long long)(per_layer * layer_count));
    Op op; op.kind = Op_CopyB2T; op.name = "copy.buffer_to_texture";
    op.buf = (BufferImpl*)src; op.tex = (TextureImpl*)dst; op.mip = mip; op.layer0 = first_layer; op.layer_count = layer_count; op.off_a = 0;
    g->ops.push_back(op);
}
GPU_API void GPUX_OpCopyBufferToTextureMip(GPU_Graph* g, GPU_Buffer* src, uint32_t src_offset, GPU_Texture* dst, uint32_t mip) {
    REC_GUARD(g);
    GPU_REQUIRE_V(src && dst && mip < dst->mip_level_count, "GPUX_OpCopyBufferToTextureMip: bad arguments");
    GPU_REQUIRE_V((uint64_t)src_offset + GPUX_TextureMipBytes(dst, mip) <= src->size, "GPUX_OpCopyBufferToTextureMip: buffer too small");
    Op op; op.kind = Op_CopyB2T; op.name = "copy.buffer_to_texture";
    op.buf = (BufferImpl*)src; op.tex = (TextureImpl*)dst; op.mip = mip; op.layer0 = 0; op.layer_count = dst->layer_count; op.off_a = src_offset;
    g->ops.push_back(op);
}
static void record_t2b(GPU_Graph* g, GPU_Texture* src, uint32_t mip, GPU_Buffer* dst, uint32_t dst_offset, const char* fn) {
    GPU_REQUIRE_V(g->in_pass == nullptr, "%s: inside a render pass", fn);
    GPU_REQUIRE_V(src && dst && mip < src->mip_level_count, "%s: bad arguments", fn);
    GPU_REQUIRE_V((uint64_t)dst_offset + GPUX_TextureMipBytes(src, mip) <= dst->size, "%s: buffer too small (%u bytes for %llu)", fn, dst->size,
                  (unsigned long long)GPUX_TextureMipBytes(src, mip));
    Op op; op.kind = Op_CopyT2B; op.name = "copy.texture_to_buffer";
    op.tex = (TextureImpl*)src; op.mip = mip; op.buf = (BufferImpl*)dst; op.off_b = dst_offset;
    g->ops.push_back(op);
}
GPU_API void GPU_OpCopyTextureToBuffer(GPU_Graph* g, GPU_Texture* src, GPU_Buffer* dst) { REC_GUARD(g); record_t2b(g, src, 0, dst, 0, __func__); }   // mip 0 only (gpu_vulkan.c:2948)
GPU_API void GPUX_OpCopyTextureMipToBuffer(GPU_Graph* g, GPU_Texture* src, uint32_t mip, GPU_Buffer* dst, uint32_t off) { REC_GUARD(g); record_t2b(g, src, mip, dst, off, __func__); }

GPU_API void GPU_OpGenerateMipmaps(GPU_Graph* g, GPU_Texture* tex) {
    REC_GUARD(g);
    GPU_REQUIRE_V(tex, "GPU_OpGenerateMipmaps: NULL texture");
    TextureImpl* t = (TextureImpl*)tex;
    GPU_REQUIRE_V(is_f4_cube(t), "GPU_OpGenerateMipmaps: implemented for RGBA32F cubemaps only");
    Op op; op.kind = Op_MipGen; op.name = "K2.mip_chain"; op.tex = t;
    g->ops.push_back(op);
}
GPU_API void GPU_OpBlit(GPU_Graph* g, const GPU_OpBlitInfo* info) {
    REC_GUARD(g);
    GPU_REQUIRE_V(g->in_pass == nullptr, "GPU_OpBlit: inside a render pass");                                        // gpu_vulkan.c:2787
    GPU_REQUIRE_V(info && info->src_texture && info->dst_texture, "GPU_OpBlit: NULL argument");
    const GPU_Texture* s = info->src_texture; const GPU_Texture* d = info->dst_texture;
    if (s == d) GPU_REQUIRE_V(info->dst_mip_level != info->src_mip_level || info->src_layer != info->dst_layer, "GPU_OpBlit: blit of a subresource onto itself");   // :2792
    GPU_REQUIRE_V(info->src_mip_level < s->mip_level_count && info->dst_mip_level < d->mip_level_count && info->src_layer < s->layer_count && info->dst_layer < d->layer_count,
                  "GPU_OpBlit: bad subresource");
    {   // whole-subresource 1:1 blit between equal formats (render.cpp:1158-1163: TAA result -> bloom_upscale_rt mip 0): a copy under either filter
        uint32_t sw1 = mip_dim(s->width, info->src_mip_level), sh1 = mip_dim(s->height, info->src_mip_level);
        uint32_t dw1 = mip_dim(d->width, info->dst_mip_level), dh1 = mip_dim(d->height, info->dst_mip_level);
        bool whole = info->src_area[0].x == 0 && info->src_area[0].y == 0 && info->dst_area[0].x == 0 && info->dst_area[0].y == 0 &&
                     (uint32_t)info->src_area[1].x == sw1 && (uint32_t)info->src_area[1].y == sh1 && (uint32_t)info->dst_area[1].x == dw1 && (uint32_t)info->dst_area[1].y == dh1;
        if (whole && sw1 == dw1 && sh1 == dh1 && s->format == d->format && s->depth == 1 && d->depth == 1) {
            Op op; op.kind = Op_Blit; op.name = "blit_1to1";
            op.tex = (TextureImpl*)s; op.tex2 = (TextureImpl*)d; op.mip = info->src_mip_level; op.mip2 = info->dst_mip_level;
            op.layer0 = info->src_layer; op.layer2 = info->dst_layer;
            op.size = (uint64_t)sw1 * sh1 * ((TextureImpl*)s)->texel_bytes;
            g->ops.push_back(op);
            return;
        }
    }
    GPU_REQUIRE_V(s->format == GPU_Format_RGBA32F && d->format == GPU_Format_RGBA32F && info->filter == GPU_Filter_Linear, "GPU_OpBlit: RGBA32F linear blits only");
    GPU_REQUIRE_V(info->src_mip_level < s->mip_level_count && info->dst_mip_level < d->mip_level_count && info->src_layer < s->layer_count && info->dst_layer < d->layer_count,
                  "GPU_OpBlit: bad subresource");
    uint32_t sw = mip_dim(s->width, info->src_mip_level), sh = mip_dim(s->height, info->src_mip_level);
    uint32_t dw = mip_dim(d->width, info->dst_mip_level), dh = mip_dim(d->height, info->dst_mip_level);
    bool full = info->src_area[0].x == 0 && info->src_area[0].y == 0 && info->dst_area[0].x == 0 && info->dst_area[0].y == 0 &&
                (uint32_t)info->src_area[1].x == sw && (uint32_t)info->src_area[1].y == sh && (uint32_t)info->dst_area[1].x == dw && (uint32_t)info->dst_area[1].y == dh;
    GPU_REQUIRE_V(full && s->depth == 1 && d->depth == 1, "GPU_OpBlit: only whole-subresource 2-D blits are implemented");
    Op op; op.kind = Op_Blit;
    // exact 2:1 square blits (every level of a power-of-two mip chain, gpu_vulkan.c:1458-1483) are the 2x2 box; anything else is the
    // linear resample of oracle/pbr_oracle.c A2 (odd levels of faces that are not a power of two: 125 -> 62)
    op.name = (sw == sh && dw == dh && sw == 2 * dw) ? "K2.blit_2to1" : "K2.blit_linear";
    op.tex = (TextureImpl*)s; op.tex2 = (TextureImpl*)d; op.mip = info->src_mip_level; op.mip2 = info->dst_mip_level;
    op.layer0 = info->src_layer; op.layer2 = info->dst_layer;
    g->ops.push_back(op);
}
static void record_clear(GPU_Graph* g, GPU_Texture* dst, uint32_t mip, int mode, const float* f, const uint32_t* u, const char* fn) {
    GPU_REQUIRE_V(g->in_pass == nullptr, "%s: inside a render pass", fn);
    GPU_REQUIRE_V(dst && (mip == GPU_MIP_LEVEL_ALL || mip < dst->mip_level_count), "%s: bad texture / mip", fn);
    Op op; op.kind = Op_Clear; op.name = "clear"; op.tex = (TextureImpl*)dst; op.mip = mip; op.clear_mode = mode;
    for (int i = 0; i < 4; ++i) { op.clear[i] = f ? f[i] : 0.0f; op.cleari[i] = u ? u[i] : 0; }
    g->ops.push_back(op);
}
GPU_API void GPU_OpClearColorF(GPU_Graph* g, GPU_Texture* dst, uint32_t mip, float r, float gg, float b, float a) {
    REC_GUARD(g); float f[4] = {r, gg, b, a}; record_clear(g, dst, mip, 0, f, nullptr, __func__);
}
GPU_API void GPU_OpClearColorI(GPU_Graph* g, GPU_Texture* dst, uint32_t mip, uint32_t r, uint32_t gg, uint32_t b, uint32_t a) {
    REC_GUARD(g); uint32_t u[4] = {r, gg, b, a}; record_clear(g, dst, mip, 1, nullptr, u, __func__);
}
GPU_API void GPU_OpClearDepthStencil(GPU_Graph* g, GPU_Texture* dst, uint32_t mip) {
    REC_GUARD(g);
    GPU_REQUIRE_V(dst && GPUX_GetFormatInfo(dst->format).depth_target, "GPU_OpClearDepthStencil: not a depth format");   // gpu_vulkan.c:2875
    float f[4] = {1.0f, 0, 0, 0}; record_clear(g, dst, mip, 2, f, nullptr, __func__);                                    // GPU_REVERSE_DEPTH false -> far = 1
}

// ------------------------------------------------------------------------------------------
// execution
// ------------------------------------------------------------------------------------------
static bool ensure_bordered(TextureImpl* t, hipStream_t st, int first_level = 0) {
    int W = (int)t->base.width, levels = (int)t->base.mip_level_count;
    if (first_level < 0) first_level = 0;
    if (first_level > levels - 1) first_level = levels - 1;
    if (t->bordered_valid && t->bordered_from <= first_level) return true;
    if (!t->bordered) {
        t->bordered_bytes = pbrk_bordered_pyramid_texels(W, levels) * 16;
        hipError_t e = hipMalloc(&t->bordered, t->bordered_bytes);
        if (e != hipSuccess) { gpu_fail("bordered twin allocation (%zu bytes) failed: %s", t->bordered_bytes, hipGetErrorString(e)); return false; }
    }
    // only the levels somebody samples get an apron: the precompute never touches level 0 of its source (400 MB at 2048^2)
    int end = t->bordered_valid ? t->bordered_from : levels;
    if (!t->bordered_valid) { for (char& v : t->cells_valid) v = 0; for (char& v : t->range_valid) v = 0; }
    int rc = pbrk_border_build_range(t->dev, t->bordered, W, levels, first_level, end, st);
    if (rc != PBRK_OK) { gpu_fail("border build failed (%d)", rc); return false; }
    t->bordered_valid = true; t->bordered_from = first_level;
    return true;
}

// {min, max} of the RGB values of level l (blocking: one reduction + an 8-byte read-back per level and content change)
static bool level_range(TextureImpl* t, int l, hipStream_t st, float* mn, float* mx) {
    const int levels = (int)t->base.mip_level_count;
    if (t->range_valid.empty()) { t->range_valid.assign((size_t)levels, 0); t->range_min.assign((size_t)levels, 0.0f); t->range_max.assign((size_t)levels, 0.0f); }
    if (!t->range_valid[(size_t)l]) {
        if (!t->range_dev && hipMalloc(&t->range_dev, 8) != hipSuccess) { t->range_dev = nullptr; return false; }
        const unsigned init[2] = {0x7F800000u, 0u};
        unsigned got[2] = {0, 0};
        HIP_OK(hipMemcpyAsync(t->range_dev, init, 8, hipMemcpyHostToDevice, st));
        const size_t n = mip_dim(t->base.width, (uint32_t)l);
        if (pbrk_level_minmax((const char*)t->dev + t->mip_offset[(size_t)l], 6 * n * n, t->range_dev, st) != PBRK_OK) return false;
        HIP_OK(hipMemcpyAsync(got, t->range_dev, 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        memcpy(&t->range_min[(size_t)l], &got[0], 4); memcpy(&t->range_max[(size_t)l], &got[1], 4);
        t->range_valid[(size_t)l] = 1;
    }
    *mn = t->range_min[(size_t)l]; *mx = t->range_max[(size_t)l];
    return true;
}
// Samples a prefilter dispatch keeps under GPUX_SetPrefilterTolerance(rel): the smallest prefix K of the weight table (weights fall
// with the sample index) with  tail(K) * max <= rel * head(K) * min.  Every tap is a convex combination of the level's texels, so a
// texel's dropped part is at most tail * max and what it keeps at least head * min: `rel` bounds the relative error of every
// texel and channel.  Sums in double, from the tail so that the small terms are not lost.
static int bounded_sample_count(const DeviceTable* tab, float mn, float mx, float rel) {
    const int n = tab->count;
    if (!(rel > 0.0f) || n < 2 || !(mn > 0.0f) || !(mx >= mn) || !isfinite(mx) || (int)tab->weights.size() != n) return n;
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += (double)tab->weights[(size_t)i];
    double tail = 0.0;
    int K = n;
    for (int i = n - 1; i >= 1; --i) {                      // try K = i: drop samples i .. n-1
        const double t2 = tail + (double)tab->weights[(size_t)i];
        const double head = total - t2;
        if (!(t2 * (double)mx <= (double)rel * head * (double)mn)) break;
        tail = t2; K = i;
    }
    return K;
}

// cells twin of one level (needs a valid bordered twin); NULL when the level is too big to be worth it (n > 512)
static void* ensure_cells(TextureImpl* t, int level, hipStream_t st) {
    int W = (int)t->base.width, levels = (int)t->base.mip_level_count;
    if (t->cells_off.empty()) {
        int first = 0;
        while (first < levels && (W >> first) > 512) ++first;
        t->cells_first = first;
        size_t off = 0;
        for (int l = first; l < levels; ++l) { t->cells_off.push_back(off); int n = W >> l; if (n < 1) n = 1; off += pbrk_cells_bytes(n); }
        t->cells_off.push_back(off);
        t->cells_valid.assign(levels > first ? levels - first : 0, 0);
        if (off) {
            hipError_t e = hipMalloc(&t->cells, off);
            if (e != hipSuccess) { t->cells = nullptr; gpu_fail("cells twin allocation (%zu bytes) failed: %s", off, hipGetErrorString(e)); }
        }
    }
    if (!t->cells || level < t->cells_first || level >= levels) return nullptr;
    int li = level - t->cells_first;
    void* c = (char*)t->cells + t->cells_off[li];
    if (t->cells_valid[li]) return c;
    int n = W >> level; if (n < 1) n = 1;
    const void* src = (const char*)t->bordered + pbrk_bordered_level_offset(W, level) * 16;
    if (pbrk_cells_build(src, n, c, st) != PBRK_OK) return nullptr;
    t->cells_valid[li] = 1;
    return c;
}
static bool cells_ready(TextureImpl* t, int level) {
    return t->cells && level >= t->cells_first && level - t->cells_first < (int)t->cells_valid.size() && t->cells_valid[level - t->cells_first];
}

static float reference_roughness(int mip) {                      // gen_prefiltered_env_map.glsl:117 + SURVEY 8d extension
    static const float tab[5] = {0.0f, 0.03f, 0.15f, 0.4f, 0.6f};
    if (mip < 5) return tab[mip < 0 ? 0 : mip];
    float r = 0.6f + 0.08f * (float)(mip - 4);
    return r > 1.0f ? 1.0f : r;
}

static hipEvent_t next_event(GPU_Graph* g, size_t& used) {
    if (used == g->ev.size()) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); g->ev.push_back(e); }
    return g->ev[used++];
}

// Runs `launch` bracketed by events when timing is on.
template <class F>
static void timed(GPU_Graph* g, const std::string& name, size_t& ev_used, F launch) {
    if (G.timing) {
        hipEvent_t a = next_event(g, ev_used);
        HIP_OK(hipEventRecord(a, g->cur));
        launch();
        hipEvent_t b = next_event(g, ev_used);
        HIP_OK(hipEventRecord(b, g->cur));
        g->timed_names.push_back(name);
    } else {
        launch();
    }
}

static hipEvent_t next_sync_event(GPU_Graph* g) {
    if (g->sync_used == g->sync_ev.size()) { hipEvent_t e; HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->sync_ev.push_back(e); }
    return g->sync_ev[g->sync_used++];
}

// Lazily built twins (apron, cells) enqueued on a side stream must be visible to the launches that follow on every other stream.
static void publish_side_work(GPU_Graph* g) {
    if (g->cur == g->stream) return;
    hipEvent_t e = next_sync_event(g);
    HIP_OK(hipEventRecord(e, g->cur));
    HIP_OK(hipStreamWaitEvent(g->stream, e, 0));
    for (hipStream_t s : g->side) if (s != g->cur) HIP_OK(hipStreamWaitEvent(s, e, 0));
}

static void exec_op(GPU_Graph* g, Op& op, size_t& ev_used) {
    hipStream_t st = g->cur;
    switch (op.kind) {
    case Op_Dispatch: {
        if (op.cpipe->kernel == Kernel_LightgridSweep) {
            TextureImpl* it = named_slot(op.set, "IMG0")->tex;
            int32_t dir; memcpy(&dir, op.push, 4);
            timed(g, dir == 0 ? "K7.sweep_x" : (dir == 1 ? "K7.sweep_y" : "K7.sweep_z"), ev_used, [&] {
                int rc = pbrk_lightgrid_sweep(it->dev, (int)it->base.width, (int)it->base.height, (int)it->base.depth, dir,
                                              (int)op.row0, (int)op.row1, (int)op.face0, (int)op.face1, st);
                if (rc != PBRK_OK) gpu_fail("K7 launch failed (%d)", rc);
            });
            return;
        }
        Slot* out = named_slot(op.set, "OUTPUT");
        TextureImpl* ot = out->tex;
        uint32_t size = mip_dim(ot->base.width, out->mip);
        void* out_ptr = (char*)ot->dev + ot->mip_offset[out->mip];
        GPUX_IBLConstants c = {0, 0.0f, 0.0f, 0};
        bool explicit_c = false;
        if (op.push_size == sizeof(GPUX_IBLConstants)) { memcpy(&c, op.push, sizeof c); explicit_c = true; }
        else if (op.push_size >= 4) memcpy(&c.mip_level, op.push, 4);
        char nm[96];
        if (op.cpipe->kernel == Kernel_BrdfLut) {
            int n = (explicit_c && c.sample_count > 0) ? c.sample_count : 4096;
            DeviceTable* ang = get_table(2, n, 0.0f, 0);
            DeviceTable* vcs = get_table(3, (int)size, 0.0f, 0);
            if (!ang || !vcs) return;
            int fmt = ot->base.format == GPU_Format_RG16F ? PBRK_FMT_RG16F : (ot->base.format == GPU_Format_RG32F ? PBRK_FMT_RG32F : PBRK_FMT_RGBA32F);
            timed(g, "K1.brdf_lut", ev_used, [&] {
                int rc = pbrk_brdf_lut(out_ptr, fmt, (int)size, n, ang->dev, vcs->dev, (int)op.row0, (int)op.row1, st);
                if (rc != PBRK_OK) gpu_fail("K1 launch failed (%d)", rc);
            });
            ot->bordered_valid = false; ot->lut_cells_valid = false;
            return;
        }
        Slot* env = named_slot(op.set, "TEX_ENV_CUBE");
        TextureImpl* et = env->tex;
        int W = (int)et->base.width, levels = (int)et->base.mip_level_count;
        float lod; int n; float divisor, alpha = 0.0f; DeviceTable* tab = nullptr;
        bool copy = false;
        if (op.cpipe->kernel == Kernel_Prefilter) {
            int mip = c.mip_level;
            if (mip == 0) { copy = true; lod = explicit_c ? c.src_lod : 1.0f; }
            else {
                lod = explicit_c ? c.src_lod : 3.0f + (float)mip;
                n = (explicit_c && c.sample_count > 0) ? c.sample_count : 8192;
                float rough = explicit_c ? c.roughness : reference_roughness(mip);
                if (!(rough > 0.0f)) { gpu_fail("prefilter: roughness must be > 0 for Monte-Carlo mips (got %g)", rough); return; }
                tab = get_table(0, n, rough, 0);
                if (!tab) return;
                divisor = 3.14159265358979323846f; alpha = tab->alpha;
            }
            snprintf(nm, sizeof nm, copy ? "K4a.prefilter_copy.mip%d" : "K4b.prefilter_mc.mip%d", mip);
        } else {
            lod = explicit_c ? c.src_lod : 6.0f;                 // gen_irradiance_map.glsl:94
            n = (explicit_c && c.sample_count > 0) ? c.sample_count : 1024;
            tab = get_table(1, n, 0.0f, 0);
            if (!tab) return;
            divisor = (float)n; alpha = 0.0f;
            snprintf(nm, sizeof nm, "K3.irradiance");
        }
        int l = (int)floorf(lod);
        if (l < 0) l = 0;
        if (l > levels - 1) l = levels - 1;                     // sampler clamps LOD to the chain
        if ((float)l != lod && lod < (float)(levels - 1)) { gpu_fail("%s: fractional source LOD %g is not supported by the precompute kernels", nm, lod); return; }
        if (!et->bordered_valid || et->bordered_from > l) {
            timed(g, "apron.env", ev_used, [&] { ensure_bordered(et, st, l); });
            if (!et->bordered_valid || et->bordered_from > l) return;
            publish_side_work(g);
        }
        int n_src = W >> l; if (n_src < 1) n_src = 1;
        const void* src = (const char*)et->bordered + pbrk_bordered_level_offset(W, l) * 16;
        const void* cells = nullptr;
        if (!copy) {
            if (cells_ready(et, l)) cells = ensure_cells(et, l, st);
            else { timed(g, "cells.env", ev_used, [&] { cells = ensure_cells(et, l, st); }); publish_side_work(g); }
        }
        int n_keep = tab ? tab->count : 0;
        if (!copy && op.cpipe->kernel == Kernel_Prefilter && G.prefilter_tol > 0.0f) {         // opt-in: tolerance-budgeted sample cut (gpux.h)
            float mn = 0.0f, mx = 0.0f;
            if (level_range(et, l, st, &mn, &mx)) n_keep = bounded_sample_count(tab, mn, mx, G.prefilter_tol);
        }
        if (!copy && op.cpipe->kernel == Kernel_Prefilter && c.mip_level >= 0 && c.mip_level < 32) G.kept_samples[c.mip_level] = n_keep;
        timed(g, nm, ev_used, [&] {
            int rc = copy ? pbrk_prefilter_copy(src, n_src, out_ptr, (int)size, (int)op.face0, (int)op.face1, (int)op.row0, (int)op.row1, st)
                          : pbrk_mc_filter(src, cells, n_src, tab->dev, n_keep, divisor, alpha, out_ptr, (int)size,
                                           (int)op.face0, (int)op.face1, (int)op.row0, (int)op.row1, st);
            if (rc != PBRK_OK) gpu_fail("%s launch failed (%d)", nm, rc);
        });
        ot->bordered_valid = false;
        return;
    }
    case Op_Shade: {
        GPU_DescriptorSet* s = op.set;
        GPU_RenderPass* rp = op.pass;
        TextureImpl* target = (TextureImpl*)rp->targets[0].texture;
        if (op.gpipe->kernel == Kernel_TaaResolve || op.gpipe->kernel == Kernel_FinalPost || op.gpipe->kernel == Kernel_BloomDown || op.gpipe->kernel == Kernel_BloomUp) {
            auto tex2d = [](TextureImpl* t, int fmt) { PbrkTex2D r; r.data = t->dev; r.format = fmt; r.width = (int)t->base.width; r.height = (int)t->base.height; return r; };
            auto out_fmt = [](GPU_Format f) {
                return f == GPU_Format_RGBA16F ? PBRK_FMT_RGBA16F : (f == GPU_Format_RGBA32F ? PBRK_FMT_RGBA32F : (f == GPU_Format_RGBA8UN ? PBRK_FMT_RGBA8UN : PBRK_FMT_BGRA8UN));
            };
            if (op.gpipe->kernel == Kernel_BloomDown || op.gpipe->kernel == Kernel_BloomUp) {
                TextureImpl* in = named_slot(s, "TEX0")->tex;
                PbrkBloomArgs a;
                a.src.data = (char*)in->dev + in->mip_offset[op.mip]; a.src.format = PBRK_FMT_RGBA16F;
                a.src.width = (int)mip_dim(in->base.width, op.mip); a.src.height = (int)mip_dim(in->base.height, op.mip);
                a.dst = (char*)target->dev + target->mip_offset[op.mip2];
                a.dst_width = (int)rp->desc.width; a.dst_height = (int)rp->desc.height;
                int32_t lvl; memcpy(&lvl, op.push, 4); a.dst_mip_level = lvl;
                a.upsample = op.gpipe->kernel == Kernel_BloomUp; a.blend_additive = op.gpipe->blend_additive;
                a.blend_src = op.blend_tex ? (const void*)((const char*)op.blend_tex->dev + op.blend_tex->mip_offset[op.blend_mip]) : nullptr;
                a.y0 = (int)op.row0; a.y1 = (int)op.row1;
                timed(g, a.upsample ? "K11.bloom_upsample" : "K10.bloom_downsample", ev_used, [&] {
                    int rc = pbrk_bloom_pass(&a, st);
                    if (rc != PBRK_OK) gpu_fail("bloom launch failed (%d)", rc);
                });
                return;
            }
            if (op.gpipe->kernel == Kernel_TaaResolve) {
                PbrkTaaArgs a;
                a.lighting_result = tex2d(named_slot(s, "LIGHTING_RESULT")->tex, PBRK_FMT_RGBA16F);
                a.gbuffer_depth = tex2d(named_slot(s, "GBUFFER_DEPTH")->tex, PBRK_FMT_R32F);
                a.gbuffer_velocity = tex2d(named_slot(s, "GBUFFER_VELOCITY")->tex, PBRK_FMT_RG16F);
                a.gbuffer_velocity_prev = tex2d(named_slot(s, "GBUFFER_VELOCITY_PREV")->tex, PBRK_FMT_RG16F);
                a.prev_frame_result = tex2d(named_slot(s, "PREV_FRAME_RESULT")->tex, PBRK_FMT_RGBA16F);
                a.out = target->dev; a.out_format = out_fmt(target->base.format);
                a.width = (int)rp->desc.width; a.height = (int)rp->desc.height; a.y0 = (int)op.row0; a.y1 = (int)op.row1;
                timed(g, "K8.taa_resolve", ev_used, [&] {
                    int rc = pbrk_taa_resolve(&a, st);
                    if (rc != PBRK_OK) gpu_fail("K8 launch failed (%d)", rc);
                });
            } else {
                PbrkFinalArgs a;
                a.src = tex2d(named_slot(s, "TEX0")->tex, PBRK_FMT_RGBA16F);
                a.out = target->dev; a.out_format = out_fmt(target->base.format);
                a.width = (int)rp->desc.width; a.height = (int)rp->desc.height; a.y0 = (int)op.row0; a.y1 = (int)op.row1;
                timed(g, "K9.final_post_process", ev_used, [&] {
                    int rc = pbrk_final_post_process(&a, st);
                    if (rc != PBRK_OK) gpu_fail("K9 launch failed (%d)", rc);
                });
            }
            return;
        }
        PbrkShadeArgs a;
        memset(&a, 0, sizeof a);
        a.width = (int)rp->desc.width; a.height = (int)rp->desc.height;
        a.x0 = 0; a.x1 = a.width; a.y0 = (int)op.row0; a.y1 = (int)op.row1;
        a.base_color = named_slot(s, "GBUFFER_BASE_COLOR")->tex->dev;
        a.normal = named_slot(s, "GBUFFER_NORMAL")->tex->dev;
        a.orm = named_slot(s, "GBUFFER_ORM")->tex->dev;
        a.emissive = named_slot(s, "GBUFFER_EMISSIVE")->tex->dev;
        a.depth = named_slot(s, "GBUFFER_DEPTH")->tex->dev;
        TextureImpl* pre = named_slot(s, "PREFILTERED_ENV_MAP")->tex;
        if (!pre->bordered_valid || pre->bordered_from > 0) { timed(g, "apron.prefiltered", ev_used, [&] { ensure_bordered(pre, g->stream); }); if (!pre->bordered_valid) return; }
        a.prefiltered_bordered = pre->bordered; a.prefiltered_size = (int)pre->base.width; a.prefiltered_levels = (int)pre->base.mip_level_count;
        {
            bool all = true;
            for (int l = 0; l < (int)pre->base.mip_level_count; ++l) if (((int)pre->base.width >> l) <= 512 && !cells_ready(pre, l)) all = false;
            auto build = [&] { for (int l = 0; l < (int)pre->base.mip_level_count; ++l) if (((int)pre->base.width >> l) <= 512) ensure_cells(pre, l, g->stream); };
            if (all) build(); else timed(g, "cells.prefiltered", ev_used, build);
            a.prefiltered_cells = pre->cells; a.prefiltered_cells_first = pre->cells_first;
        }
        a.flags = 0;
        if (op.gpipe->shade_flags & GPUX_Shade_IBL) {
            a.flags |= PBRK_SHADE_IBL;
            TextureImpl* irr = named_slot(s, "TEX_IRRADIANCE_MAP")->tex;
            if (!irr->bordered_valid || irr->bordered_from > 0) { timed(g, "apron.irradiance", ev_used, [&] { ensure_bordered(irr, g->stream); }); if (!irr->bordered_valid) return; }
            a.irradiance_bordered = irr->bordered; a.irradiance_size = (int)irr->base.width;
            a.irradiance_cells = ensure_cells(irr, 0, g->stream);
            TextureImpl* lut = named_slot(s, "BRDF_INTEGRATION_MAP")->tex;
            a.lut = lut->dev; a.lut_size = (int)lut->base.width;
            if (!lut->lut_cells_valid) {
                size_t bytes = (size_t)(a.lut_size + 1) * (a.lut_size + 1) * 16;
                if (!lut->lut_cells && hipMalloc(&lut->lut_cells, bytes) != hipSuccess) lut->lut_cells = nullptr;
                if (lut->lut_cells && pbrk_lut_cells_build(lut->dev, a.lut_size, lut->lut_cells, g->stream) == PBRK_OK) lut->lut_cells_valid = true;
            }
            a.lut_cells = lut->lut_cells_valid ? lut->lut_cells : nullptr;
        }
        if (op.gpipe->shade_flags & GPUX_Shade_VoxelGI) {
            a.flags |= PBRK_SHADE_GI;
            TextureImpl* grid = named_slot(s, "LIGHTGRID")->tex;
            a.lightgrid = grid->dev; a.lightgrid_size = (int)grid->base.width;
            TextureImpl* prev = named_slot(s, "PREV_FRAME_RESULT")->tex;
            a.prev_frame_levels = (int)(prev->base.mip_level_count < 8 ? prev->base.mip_level_count : 8);
            a.prev_frame_w = (int)prev->base.width; a.prev_frame_h = (int)prev->base.height;
            for (int l = 0; l < a.prev_frame_levels; ++l) a.prev_frame[l] = (char*)prev->dev + prev->mip_offset[l];
            if (!(op.gpipe->shade_flags & GPUX_Shade_IBL)) {                   // the LUT fetch of :681 feeds the GI specular term too
                TextureImpl* lut = named_slot(s, "BRDF_INTEGRATION_MAP")->tex;
                a.lut = lut->dev; a.lut_size = (int)lut->base.width;
            }
        }
        if (op.gpipe->shade_flags & GPUX_Shade_LightShafts) a.flags |= PBRK_SHADE_SHAFTS;
        if (op.gpipe->shade_flags & GPUX_Shade_SunShadows) {
            TextureImpl* sun = named_slot(s, "SUN_DEPTH_MAP")->tex;
            a.flags |= PBRK_SHADE_SHADOWS;
            a.sun_depth = sun->dev; a.sun_depth_w = (int)sun->base.width; a.sun_depth_h = (int)sun->base.height;
        }
        a.out = target->dev;
        a.out_format = target->base.format == GPU_Format_RGBA16F ? PBRK_FMT_RGBA16F : PBRK_FMT_RGBA32F;
        BufferImpl* gb = named_slot(s, "GLOBALS")->buf;
        // Globals snapshot at submit time: the caller fills the persistently mapped buffer before GPU_GraphSubmit (render.cpp:991)
        if (gb->pinned_host) memcpy(a.globals, gb->dev, 552);
        else HIP_OK(hipMemcpy(a.globals, gb->dev, 552, hipMemcpyDeviceToHost));
        timed(g, "K5.shade", ev_used, [&] {
            int rc = pbrk_shade(&a, st);
            if (rc != PBRK_OK) gpu_fail("K5 launch failed (%d)", rc);
        });
        target->bordered_valid = false;
        return;
    }
    case Op_MipGen: {
        timed(g, op.name, ev_used, [&] {
            int rc = pbrk_mip_chain(op.tex->dev, (int)op.tex->base.width, (int)op.tex->base.mip_level_count, st);
            if (rc != PBRK_OK) gpu_fail("K2 launch failed (%d)", rc);
        });
        op.tex->bordered_valid = false;
        return;
    }
    case Op_Blit: {
        if (op.folded) { op.tex2->bordered_valid = false; op.tex2->lut_cells_valid = false; return; }   // its consumer reads the source (fold_blits)
        if (op.size) {                                                         // 1:1 copy of one layer
            const void* src = (const char*)op.tex->dev + op.tex->mip_offset[op.mip] + op.size * op.layer0;
            void* dst = (char*)op.tex2->dev + op.tex2->mip_offset[op.mip2] + op.size * op.layer2;
            timed(g, op.name, ev_used, [&] { HIP_OK(hipMemcpyAsync(dst, src, op.size, hipMemcpyDeviceToDevice, st)); });
            op.tex2->bordered_valid = false; op.tex2->lut_cells_valid = false;
            return;
        }
        uint32_t nsw = mip_dim(op.tex->base.width, op.mip), nsh = mip_dim(op.tex->base.height, op.mip);
        uint32_t ndw = mip_dim(op.tex2->base.width, op.mip2), ndh = mip_dim(op.tex2->base.height, op.mip2);
        size_t layer_bytes_s = (size_t)nsw * nsh * 16, layer_bytes_d = (size_t)ndw * ndh * 16;
        const void* src = (const char*)op.tex->dev + op.tex->mip_offset[op.mip] + layer_bytes_s * op.layer0;
        void* dst = (char*)op.tex2->dev + op.tex2->mip_offset[op.mip2] + layer_bytes_d * op.layer2;
        timed(g, op.name, ev_used, [&] {
            int rc = (nsw == nsh && ndw == ndh && nsw == 2 * ndw) ? pbrk_box_downsample(src, (int)nsw, dst, 1, st)
                                                                  : pbrk_blit_linear(src, (int)nsw, (int)nsh, dst, (int)ndw, (int)ndh, 1, st);
            if (rc != PBRK_OK) gpu_fail("blit launch failed (%d)", rc);
        });
        op.tex2->bordered_valid = false;
        return;
    }
    case Op_CopyB2B:
        timed(g, op.name, ev_used, [&] { HIP_OK(hipMemcpyAsync((char*)op.buf2->dev + op.off_b, (const char*)op.buf->dev + op.off_a, op.size, hipMemcpyDefault, st)); });
        return;
    case Op_CopyB2T: {
        uint64_t per_layer = GPUX_TextureMipBytes(&op.tex->base, op.mip) / op.tex->base.layer_count;
        timed(g, op.name, ev_used, [&] {
            HIP_OK(hipMemcpyAsync((char*)op.tex->dev + op.tex->mip_offset[op.mip] + per_layer * op.layer0, (const char*)op.buf->dev + op.off_a,
                                  per_layer * op.layer_count, hipMemcpyDefault, st));
        });
        op.tex->bordered_valid = false; op.tex->lut_cells_valid = false;
        return;
    }
    case Op_CopyT2B:
        timed(g, op.name, ev_used, [&] {
            HIP_OK(hipMemcpyAsync((char*)op.buf->dev + op.off_b, (const char*)op.tex->dev + op.tex->mip_offset[op.mip],
                                  GPUX_TextureMipBytes(&op.tex->base, op.mip), hipMemcpyDefault, st));
        });
        return;
    case Op_Clear: {
        TextureImpl* t = op.tex;
        uint32_t m0 = op.mip == GPU_MIP_LEVEL_ALL ? 0 : op.mip, m1 = op.mip == GPU_MIP_LEVEL_ALL ? t->base.mip_level_count : op.mip + 1;
        // Levels are back to back in the allocation ([mip][layer][y][x], tight), so GPU_MIP_LEVEL_ALL is ONE fill of the whole
        // texture instead of one per level: the reference clears bloom_upscale_rt this way every frame (render.cpp:1156; 11 levels at
        // 1080p = 11 fill kernels of ~5 us each in rocprofv3's trace, a quarter of the post-process tail, before round 3).
        const bool whole = op.mip == GPU_MIP_LEVEL_ALL;
        for (uint32_t m = m0; m < (whole ? m0 + 1 : m1); ++m) {
            size_t bytes = whole ? t->bytes : (size_t)GPUX_TextureMipBytes(&t->base, m);
            void* p = (char*)t->dev + t->mip_offset[m];
            if (whole && op.skip_level0) { p = (char*)t->dev + t->mip_offset[1]; bytes = t->bytes - t->mip_offset[1]; }   // fold_blits
            // build one texel pattern on the host and replicate it (clears are rare: upload a staging row)
            uint32_t tb = t->texel_bytes;
            std::vector<uint8_t> texel(tb, 0);
            if (op.clear_mode == 2 || t->base.format == GPU_Format_R32F) { memcpy(texel.data(), &op.clear[0], 4 < tb ? 4 : tb); }
            else if (op.clear_mode == 1) { for (uint32_t k = 0; k * 4 < tb && k < 4; ++k) memcpy(texel.data() + 4 * k, &op.cleari[k], 4); }
            else if (t->base.format == GPU_Format_RGBA32F || t->base.format == GPU_Format_RG32F) { memcpy(texel.data(), op.clear, tb); }
            else if (t->base.format == GPU_Format_RGBA8UN || t->base.format == GPU_Format_BGRA8UN) {
                int order[4] = {0, 1, 2, 3};
                if (t->base.format == GPU_Format_BGRA8UN) { order[0] = 2; order[2] = 0; }
                for (int k = 0; k < 4; ++k) { float v = op.clear[order[k]]; v = v < 0 ? 0 : (v > 1 ? 1 : v); texel[k] = (uint8_t)(v * 255.0f + 0.5f); }
            } else {
                bool zero = op.clear[0] == 0 && op.clear[1] == 0 && op.clear[2] == 0 && op.clear[3] == 0;
                if (!zero) { gpu_fail("GPU_OpClearColorF: non-zero clears of format %d are not implemented", (int)t->base.format); return; }
            }
            // one fill launch for every texel size and alignment (k_cube.hip)
            if (tb == 1 || tb == 2 || tb == 4 || tb == 8 || tb == 16) {
                int rc = pbrk_fill_pattern(p, bytes, texel.data(), (int)tb, st);
                if (rc != PBRK_OK) gpu_fail("clear launch failed (%d)", rc);
            } else {
                std::vector<uint8_t> host(bytes);
                for (size_t o = 0; o < bytes; o += tb) memcpy(host.data() + o, texel.data(), tb);
                HIP_OK(hipMemcpyAsync(p, host.data(), bytes, hipMemcpyHostToDevice, st));
                HIP_OK(hipStreamSynchronize(st));      // the staging vector dies at scope exit
            }
        }
        t->bordered_valid = false; t->lut_cells_valid = false;
        return;
    }
    }
}

// sample evaluations per output texel of a precompute dispatch (the same decoding of the push constants as exec_op)
static double dispatch_samples_per_texel(const Op& op) {
    GPUX_IBLConstants c = {0, 0.0f, 0.0f, 0};
    bool explicit_c = op.push_size == sizeof(GPUX_IBLConstants);
    if (explicit_c) memcpy(&c, op.push, sizeof c);
    else if (op.push_size >= 4) memcpy(&c.mip_level, op.push, 4);
    if (op.cpipe->kernel == Kernel_Irradiance) return (explicit_c && c.sample_count > 0) ? c.sample_count : 1024;
    if (c.mip_level == 0) return 4.0;                                       // the copy level: one bilinear fetch, HBM-bound
    int n = (explicit_c && c.sample_count > 0) ? c.sample_count : 8192;
    float rough = explicit_c ? c.roughness : reference_roughness(c.mip_level);
    if (!(rough > 0.0f)) return n;
    DeviceTable* tab = get_table(0, n, rough, 0);                           // cached; only entries with non-zero weight are kept
    return tab ? (double)tab->count : (double)n;
}

// ---- hipGraph replay of the per-frame chain --------------------------------------------------------------------------
// A frame of the reference records ~25 dependent small launches (light-grid sweep, shade, TAA resolve, 12 bloom passes, tone map:
// render.cpp:1061-1187) into one GPU_Graph and submits it; on a HIP stream each dependent launch costs ~2.7 us of dispatch latency,
// inside an instantiated hipGraph ~2.0 us (tools/ubench_graph.hip).  With replay on, GPU_GraphSubmit captures the op loop of an
// eligible graph into a hipGraph, updates the executable graph kept from the previous submission of the same GPU_Graph in place
// (hipGraphExecUpdate: same topology, new kernel arguments -- the Globals snapshot, the ping-pong targets) or instantiates a new one,
// and launches that.  Eligible = every op is a launch-only op whose lazily built inputs (aprons, cells twins, tables) already
// exist: nothing inside a capture may allocate or synchronise.  Anything else -- the precompute with its side streams, the first
// frame that still builds twins, per-op timing -- takes the plain path.
static bool op_replayable(const Op& op) {
    switch (op.kind) {
    case Op_Dispatch: return op.cpipe->kernel == Kernel_LightgridSweep;
    case Op_MipGen: return true;
    case Op_Blit: return true;
    case Op_CopyB2B: case Op_CopyB2T: case Op_CopyT2B: return false;          // host pointers may be involved: keep them out of captures
    case Op_Clear: {
        const uint32_t tb = op.tex->texel_bytes;                              // one fill launch; the staged form of other texel sizes synchronises
        return tb == 1 || tb == 2 || tb == 4 || tb == 8 || tb == 16;
    }
    case Op_Shade: {
        KernelId k = op.gpipe->kernel;
        if (k == Kernel_TaaResolve || k == Kernel_FinalPost || k == Kernel_BloomDown || k == Kernel_BloomUp) return true;
        GPU_DescriptorSet* s = op.set;
        TextureImpl* pre = named_slot(s, "PREFILTERED_ENV_MAP")->tex;
        if (!pre->bordered_valid || pre->bordered_from > 0) return false;
        for (int l = 0; l < (int)pre->base.mip_level_count; ++l) if (((int)pre->base.width >> l) <= 512 && !cells_ready(pre, l)) return false;
        if (op.gpipe->shade_flags & GPUX_Shade_IBL) {
            TextureImpl* irr = named_slot(s, "TEX_IRRADIANCE_MAP")->tex;
            if (!irr->bordered_valid || irr->bordered_from > 0 || !cells_ready(irr, 0)) return false;
            if (!named_slot(s, "BRDF_INTEGRATION_MAP")->tex->lut_cells_valid) return false;
        }
        if (op.pass && pbrk_shade_needs_tables((int)op.pass->desc.width, (int)op.pass->desc.height)) return false;    // first tiled launch of a frame size builds its tables
        return named_slot(s, "GLOBALS")->buf->pinned_host;
    }
    }
    return false;
}

// Graph-level eligibility.  op_replayable() looks at the sampler twins as they are NOW; an earlier op of the same graph that writes
// one of the maps a later shade op samples (Op_Clear / Op_MipGen / Op_Blit / a storage-image dispatch drop the apron, cells and LUT
// twins in exec_op) would have that shade op rebuild its twins -- allocation + synchronisation -- inside the capture.
// Such a graph takes the plain path: the textures written by the ops seen so far are tracked and compared with what each shade
// op samples.
static bool graph_replayable(GPU_Graph* g) {
    std::vector<const TextureImpl*> written;
    auto is_written = [&](const TextureImpl* t) { for (const TextureImpl* w : written) if (w == t) return true; return false; };
    for (const Op& op : g->ops) {
        if (!op_replayable(op)) return false;
        switch (op.kind) {
        case Op_Clear: case Op_MipGen: written.push_back(op.tex); break;
        case Op_Blit: if (op.tex2) written.push_back(op.tex2); break;          // tex = source, tex2 = destination
        case Op_Dispatch: { Slot* o = named_slot(op.set, "IMG0"); if (o && o->tex) written.push_back(o->tex); break; }
        case Op_Shade: {
            KernelId k = op.gpipe->kernel;
            if (k == Kernel_TaaResolve || k == Kernel_FinalPost || k == Kernel_BloomDown || k == Kernel_BloomUp) break;
            for (const char* name : {"PREFILTERED_ENV_MAP", "TEX_IRRADIANCE_MAP", "BRDF_INTEGRATION_MAP"}) {
                Slot* sl = named_slot(op.set, name);
                if (sl && sl->tex && is_written(sl->tex)) return false;
            }
            break;
        }
        default: break;
        }
    }
    return true;
}

GPU_API void GPUX_SetGraphReplay(int enable) { G.replay = enable < 0 ? -1 : (enable != 0); }
GPU_API void GPUX_GraphReplayStats(GPU_Graph* g, uint64_t* launches, uint64_t* updates, uint64_t* instantiations) {
    if (!g) return;
    if (launches) *launches = g->replay_launches;
    if (updates) *updates = g->replay_updates;
    if (instantiations) *instantiations = g->replay_instantiations;
}

// The reference's bloom chain copies the TAA result into level 0 of bloom_upscale_rt (a 1:1 GPU_OpBlit, render.cpp:1158-1163) and, five
// passes later, ADDS the last upsample onto that copy (additive blending, render.cpp:1165-1176).  Nothing else reads the copy: the
// upsample can add onto the blit's SOURCE and write the sum to the target, and the copy (33 MB of traffic, one launch per 1080p frame)
// need not exist.  Folded only when the ops between the two provably touch neither subresource: bloom draws with other sources and
// targets; anything else in between keeps the blit.  Same values, same additions: the target's bits do not change.
static uint64_t g_folded_blits = 0;
GPU_API uint64_t GPUX_FoldedBlitCount(void) { return g_folded_blits; }
static void fold_blits(GPU_Graph* g) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PBR_GRAPH_FOLD"); on = e ? atoi(e) : 1; }
    for (Op& op : g->ops) { op.folded = false; op.blend_tex = nullptr; op.skip_level0 = false; }
    if (!on) return;
    for (size_t i = 0; i < g->ops.size(); ++i) {
        Op& b = g->ops[i];
        // a clear of ALL levels directly followed by a whole-level copy onto level 0 (render.cpp:1156-1163): level 0 -- three quarters of
        // the bytes -- is overwritten before anything can read the zeros
        if (b.kind == Op_Blit && b.size && i > 0 && b.mip2 == 0 && b.tex2->base.layer_count == 1 && b.tex2->base.mip_level_count > 1) {
            Op& c = g->ops[i - 1];
            if (c.kind == Op_Clear && c.tex == b.tex2 && c.mip == GPU_MIP_LEVEL_ALL && b.tex != b.tex2) c.skip_level0 = true;
        }
        if (b.kind != Op_Blit || !b.size || b.layer0 != 0 || b.layer2 != 0 || b.tex->base.layer_count != 1 || b.tex2->base.layer_count != 1 ||
            b.tex->base.format != GPU_Format_RGBA16F || b.tex == b.tex2) continue;
        for (size_t j = i + 1; j < g->ops.size(); ++j) {
            Op& d = g->ops[j];
            if (d.kind != Op_Shade || !(d.gpipe->kernel == Kernel_BloomDown || d.gpipe->kernel == Kernel_BloomUp)) break;
            TextureImpl* target = (TextureImpl*)d.pass->targets[0].texture;
            Slot* in = named_slot(d.set, "TEX0");
            if (!in || !in->tex) break;
            const bool reads_copy = in->tex == b.tex2 && d.mip == b.mip2, writes_src = target == b.tex && d.mip2 == b.mip;
            if (reads_copy || writes_src) break;
            if (target == b.tex2 && d.mip2 == b.mip2) {
                const uint32_t H = mip_dim(target->base.height, d.mip2);
                if (d.gpipe->kernel == Kernel_BloomUp && d.gpipe->blend_additive && d.row0 == 0 && d.row1 == H && !(in->tex == b.tex && d.mip == b.mip)) {
                    b.folded = true; d.blend_tex = b.tex; d.blend_mip = b.mip; ++g_folded_blits;
                }
                break;
            }
        }
    }
}

// Textures an op reads / writes, at whole-texture granularity; false when the op touches memory this analysis does not follow (buffers,
// host pointers).  Draws and dispatches: every texture bound in their descriptor set counts as read, a dispatch's also as written
// (storage images), a draw's render targets as written.
static bool op_access(const Op& op, std::vector<TextureImpl*>& r, std::vector<TextureImpl*>& w) {
    switch (op.kind) {
    case Op_Dispatch: case Op_Shade:
        if (!op.set) return false;
        for (const Slot& sl : op.set->slots) {
            if (sl.buf) return false;
            if (sl.tex) { r.push_back(sl.tex); if (op.kind == Op_Dispatch) w.push_back(sl.tex); }
        }
        if (op.kind == Op_Shade) {
            if (!op.pass) return false;
            for (const GPU_TextureView& tv : op.pass->targets) w.push_back((TextureImpl*)tv.texture);
            if (op.blend_tex) r.push_back(op.blend_tex);
        }
        return true;
    case Op_MipGen: r.push_back(op.tex); w.push_back(op.tex); return true;
    case Op_Blit: r.push_back(op.tex); w.push_back(op.tex2); return true;
    case Op_Clear: w.push_back(op.tex); return true;
    default: return false;
    }
}
static int g_overlap_on = -1;                   // -1: PBR_GRAPH_OVERLAP or the default (1)
static uint64_t g_overlapped_submits = 0;
GPU_API void GPUX_SetGraphOverlap(int on) { g_overlap_on = on; }
GPU_API uint64_t GPUX_OverlappedSubmitCount(void) { return g_overlapped_submits; }
static bool contains(const std::vector<TextureImpl*>& v, const TextureImpl* t) { for (const TextureImpl* x : v) if (x == t) return true; return false; }

GPU_API void GPU_GraphSubmit(GPU_Graph* g) {
    GPU_REQUIRE_V(g && !g->submitted, "GPU_GraphSubmit: graph is NULL or already submitted");
    GPU_REQUIRE_V(g->in_pass == nullptr && g->preparing == nullptr, "GPU_GraphSubmit: render pass still open");
    g->timed_names.clear(); g->timed_ms.clear();
    size_t ev_used = 0;
    g->sync_used = 0;
    g->cur = g->stream;
    // Submission order is execution order, as on the reference's single queue (vkQueueSubmit, gpu_vulkan.c:2481-2530): a graph
    // submitted while another one is still in flight (main.cpp:49-51, 91-99 keeps two) starts after everything enqueued so far on
    // the previous graph's stream -- its kernels, lazily built sampler twins, and any exchange the caller appended through
    // GPUX_GraphStream.  Side streams fork from g->stream, so they inherit the dependency.
    static int order_on = -1;                                     // PBR_GRAPH_ORDER=0: diagnostic only (shows that the ordering test can fail)
    if (order_on < 0) { const char* e = getenv("PBR_GRAPH_ORDER"); order_on = e ? atoi(e) : 1; }
    fold_blits(g);
    // Frames in flight (main.cpp:49-51, 91-99 keeps two graphs): the tail of a frame's graph -- its bloom chain, a dozen small dependent
    // launches that leave most of the chip idle -- touches only the bloom targets, the TAA result it reads and the backbuffer.  The next
    // graph's leading ops (light-grid sweep, shade, TAA resolve) touch none of those: they wait for the event in front of the previous
    // graph's first bloom draw instead of for its end, and run beside that tail; the first op that shares a texture with the tail (and
    // everything behind it) waits for the end as before.  Whole-texture granularity, every texture bound to an op counts; an op this
    // analysis does not follow ends the overlap.  PBR_GRAPH_OVERLAP=0: every graph waits for all of its predecessor.
    if (g_overlap_on < 0) { const char* e = getenv("PBR_GRAPH_OVERLAP"); g_overlap_on = e ? atoi(e) : 1; }
    const int overlap_on = g_overlap_on;
    if (G.replay < 0) { const char* e = getenv("PBR_GRAPH_REPLAY"); G.replay = e ? (atoi(e) != 0) : 0; }
    size_t order_before = 0;                                      // index of the first op that must wait for the previous graph's end
    GPU_Graph* order_prev = nullptr;
    if (order_on && G.last_submitted && G.last_submitted != g) {
        GPU_Graph* prev = G.last_submitted;
        if (!prev->order_ev) HIP_OK(hipEventCreateWithFlags(&prev->order_ev, hipEventDisableTiming));
        HIP_OK(hipEventRecord(prev->order_ev, prev->stream));
        if (overlap_on && prev->mid_recorded && G.replay != 1) {
            std::vector<TextureImpl*> r, w;
            for (; order_before < g->ops.size(); ++order_before) {
                r.clear(); w.clear();
                if (!op_access(g->ops[order_before], r, w)) break;
                bool clash = false;
                for (TextureImpl* t : w) clash |= contains(prev->tail_reads, t) || contains(prev->tail_writes, t);
                for (TextureImpl* t : r) clash |= contains(prev->tail_writes, t);
                if (clash) break;
            }
        }
        if (order_before > 0) { HIP_OK(hipStreamWaitEvent(g->stream, prev->mid_ev, 0)); order_prev = prev; ++g_overlapped_submits; }
        else HIP_OK(hipStreamWaitEvent(g->stream, prev->order_ev, 0));
    }
    G.last_submitted = g;
    // this graph's own head / tail split: the tail starts at the first bloom draw
    size_t mid_idx = g->ops.size();
    for (size_t i = 0; i < g->ops.size(); ++i)
        if (g->ops[i].kind == Op_Shade && (g->ops[i].gpipe->kernel == Kernel_BloomDown || g->ops[i].gpipe->kernel == Kernel_BloomUp)) { mid_idx = i; break; }
    g->mid_recorded = false; g->tail_reads.clear(); g->tail_writes.clear();
    if (overlap_on && mid_idx > 0 && mid_idx < g->ops.size()) {
        bool known = true;
        for (size_t i = mid_idx; i < g->ops.size() && known; ++i) known = op_access(g->ops[i], g->tail_reads, g->tail_writes);
        if (!known) { mid_idx = g->ops.size(); g->tail_reads.clear(); g->tail_writes.clear(); }
    } else mid_idx = g->ops.size();
    // the chain property the next submission relies on: this graph's mid event (or, without one, its end) implies the END of the graph
    // before it -- so the wait for that end never comes later than the mid event
    if (order_before > mid_idx) order_before = mid_idx;
    size_t op_index = 0;
    if (G.replay < 0) { const char* e = getenv("PBR_GRAPH_REPLAY"); G.replay = e ? (atoi(e) != 0) : 0; }
    bool capture = G.replay == 1 && !G.timing && !g->replay_broken && !g->ops.empty();
    if (capture) capture = graph_replayable(g);
    g->span_recorded = false;
    if (G.timing) {
        if (!g->span_a) { HIP_OK(hipEventCreate(&g->span_a)); HIP_OK(hipEventCreate(&g->span_b)); }
        HIP_OK(hipEventRecord(g->span_a, g->stream));
    }
    if (capture && hipStreamBeginCapture(g->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); capture = false; g->replay_broken = true; }
    // Row-ranged precompute dispatches (the work units of a partitioned job) are too small to keep 256 CUs x 8 waves busy one
    // at a time: consecutive ones whose outputs are disjoint and which do not read each other's output go round-robin onto
    // side streams, fenced by a fork event before the first and join events after the last.  Everything else stays in order.
    if (G.tile_streams < 0) { const char* e = getenv("PBR_TILE_STREAMS"); G.tile_streams = e ? atoi(e) : 4; }
    if (G.tile_streams > 16) G.tile_streams = 16;
    const int n_side = G.tile_streams;
    struct Wr { TextureImpl* t; uint32_t mip, f0, f1, r0, r1; };
    std::vector<Wr> writes; std::vector<TextureImpl*> reads;
    bool open = false; size_t rr = 0;
    auto close_region = [&] {
        if (!open) return;
        for (hipStream_t s : g->side) { hipEvent_t e = next_sync_event(g); HIP_OK(hipEventRecord(e, s)); HIP_OK(hipStreamWaitEvent(g->stream, e, 0)); }
        writes.clear(); reads.clear(); open = false; g->cur = g->stream;
    };
    for (Op& op : g->ops) {
        if (order_prev && op_index == order_before) {                  // from here on: after everything the previous graph enqueued
            close_region();
            HIP_OK(hipStreamWaitEvent(g->stream, order_prev->order_ev, 0));
            order_prev = nullptr;
        }
        if (op_index == mid_idx && !capture) {                         // head done: the next graph's independent ops may start
            close_region();
            if (!g->mid_ev) HIP_OK(hipEventCreateWithFlags(&g->mid_ev, hipEventDisableTiming));
            HIP_OK(hipEventRecord(g->mid_ev, g->stream));
            g->mid_recorded = true;
        }
        ++op_index;
        bool tile = n_side >= 2 && op.kind == Op_Dispatch && op.rows_explicit &&
                    (op.cpipe->kernel == Kernel_Prefilter || op.cpipe->kernel == Kernel_Irradiance);
        Slot* out = tile ? named_slot(op.set, "OUTPUT") : nullptr;
        Slot* env = tile ? named_slot(op.set, "TEX_ENV_CUBE") : nullptr;
        // a launch of several full-chip rounds (2048 resident workgroups x 256 texels x 8192 samples each) gains nothing from
        // company and keeps its own event timing clean: only smaller ones are overlapped
        // (a whole level fills the chip by itself from ~1e9 evaluations on; shares of a level -- a rank's tiles -- come in
        // groups with ragged tails and gain up to the bound above)
        if (tile && out && out->tex) {
            uint32_t size = mip_dim(out->tex->base.width, out->mip);
            double evals = (double)(op.face1 - op.face0) * (op.row1 - op.row0) * size * dispatch_samples_per_texel(op);
            bool whole = op.face0 == 0 && op.face1 == out->tex->base.layer_count && op.row0 == 0 && op.row1 == size;
            if (evals >= (whole ? 1.0e9 : 2.0e6 * 8192.0)) tile = false;
            // fork + join cost ~0.25 ms of cross-stream signalling: whole small levels alone (the tail of a single-GPU job) do not
            // repay it -- they join a region that shares of a level have opened, but do not open one
            if (whole && !open) tile = false;
        }
        if (!tile || !out || !env || !out->tex || !env->tex) { close_region(); exec_op(g, op, ev_used); continue; }
        Wr w = {out->tex, out->mip, op.face0, op.face1, op.row0, op.row1};
        bool clash = w.t == env->tex;
        for (const Wr& o : writes) {
            if (o.t == env->tex) clash = true;
            if (o.t == w.t && o.mip == w.mip && o.f0 < w.f1 && w.f0 < o.f1 && o.r0 < w.r1 && w.r0 < o.r1) clash = true;
        }
        for (TextureImpl* r : reads) if (r == w.t) clash = true;
        if (clash) close_region();
        if (w.t == env->tex) { exec_op(g, op, ev_used); continue; }        // reads what it writes: never overlapped
        if (!open) {
            while ((int)g->side.size() > n_side) { (void)hipStreamSynchronize(g->side.back()); (void)hipStreamDestroy(g->side.back()); g->side.pop_back(); }
            while ((int)g->side.size() < n_side) { hipStream_t s; HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); g->side.push_back(s); }
            hipEvent_t e = next_sync_event(g);
            HIP_OK(hipEventRecord(e, g->stream));
            for (hipStream_t s : g->side) HIP_OK(hipStreamWaitEvent(s, e, 0));
            open = true;
        }
        writes.push_back(w); reads.push_back(env->tex);
        g->cur = g->side[rr++ % g->side.size()];
        exec_op(g, op, ev_used);
        g->cur = g->stream;
    }
    close_region();
    if (order_prev) { HIP_OK(hipStreamWaitEvent(g->stream, order_prev->order_ev, 0)); order_prev = nullptr; }   // every op was independent: still end after it
    if (G.timing) { HIP_OK(hipEventRecord(g->span_b, g->stream)); g->span_recorded = true; }
    if (capture) {
        hipGraph_t cg = nullptr;
        bool ok = hipStreamEndCapture(g->stream, &cg) == hipSuccess && cg;
        if (ok && g->exec) {
            hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult res;
            if (hipGraphExecUpdate(g->exec, cg, &bad, &res) == hipSuccess) ++g->replay_updates;
            else { (void)hipGetLastError(); (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
        }
        if (ok && !g->exec) {
            if (hipGraphInstantiate(&g->exec, cg, nullptr, nullptr, 0) == hipSuccess) ++g->replay_instantiations;
            else { (void)hipGetLastError(); g->exec = nullptr; ok = false; }
        }
        if (ok && hipGraphLaunch(g->exec, g->stream) == hipSuccess) ++g->replay_launches; else ok = false;
        if (cg) (void)hipGraphDestroy(cg);
        if (!ok) {
            // nothing of the captured work has run: take the plain path for this and every later submission of this graph
            (void)hipGetLastError();
            g->replay_broken = true;
            if (g->exec) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
            for (Op& op : g->ops) exec_op(g, op, ev_used);
        }
    }
    g->submitted = true;
}

GPU_API void GPU_GraphWait(GPU_Graph* g) {
    GPU_REQUIRE_V(g, "GPU_GraphWait: NULL graph");
    HIP_OK(hipStreamSynchronize(g->stream));
    g->timed_ms.clear();
    for (size_t i = 0; i < g->timed_names.size(); ++i) {
        float ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&ms, g->ev[2 * i], g->ev[2 * i + 1]));
        g->timed_ms.push_back(ms);
    }
    g->span_ms = 0.0f;
    if (g->span_recorded) { HIP_OK(hipEventElapsedTime(&g->span_ms, g->span_a, g->span_b)); g->span_recorded = false; }
    reset_graph(g);                      // Wait also resets the graph [gpu.h:452]
}

GPU_API void GPUX_EnableOpTiming(int enable) { G.timing = enable != 0; }
GPU_API void GPUX_SetTileStreams(int count) { G.tile_streams = count < 0 ? -1 : count; }
GPU_API uint32_t GPUX_GraphTimedOpCount(GPU_Graph* g) { return g ? (uint32_t)g->timed_ms.size() : 0; }
GPU_API const char* GPUX_GraphTimedOpName(GPU_Graph* g, uint32_t i) { return (g && i < g->timed_names.size()) ? g->timed_names[i].c_str() : ""; }
GPU_API float GPUX_GraphTimedOpMs(GPU_Graph* g, uint32_t i) { return (g && i < g->timed_ms.size()) ? g->timed_ms[i] : 0.0f; }
GPU_API float GPUX_GraphSpanMs(GPU_Graph* g) { return g ? g->span_ms : 0.0f; }
GPU_API void GPUX_SetPrefilterTolerance(float rel) { G.prefilter_tol = rel > 0.0f ? rel : 0.0f; }
GPU_API int GPUX_PrefilterKeptSamples(uint32_t mip) { return mip < 32 ? G.kept_samples[mip] : 0; }
