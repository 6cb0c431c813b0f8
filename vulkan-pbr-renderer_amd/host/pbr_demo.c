/*
 * pbr_demo.c -- headless C driver of the hot path through the GPU_* boundary (what main.cpp:35-51 +
 * HotreloadShaders + BuildRenderCommands do for this path, without window / mesh import / raster passes).
 *
 *   pbr_demo <cube_strip.hdr> [irradiance_size lut_size specular_size min_size [width height [frame.ppm]]]
 *
 * Loads a vertical-strip HDR cube (asset_import.cpp:17-27), runs the IBL precompute (render.cpp:505-619),
 * shades a flat synthetic G-buffer (a metallic floor under the sky), then runs three frames of the per-frame chain
 * lighting -> TAA resolve -> bloom -> tone map (render.cpp:1119-1187) plus the light-grid sweep (render.cpp:1061-1072),
 * and prints fp64 checksums of every map / frame plus HIP-event timings per kernel, one "key value" pair per line, so
 * that a test can compare it with the same sequence driven from another language.  With a last argument the 8-bit
 * frame is written as a binary PPM.
 */
#include "pbr_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double checksum_texture_mip(GPU_Texture* tex, uint32_t mip, int channels_f32) {
    uint64_t bytes = GPUX_TextureMipBytes(tex, mip);
    GPU_Buffer* buf = GPU_MakeBuffer((uint32_t)bytes, GPU_BufferFlag_CPU, NULL);
    GPU_Graph* g = GPU_MakeGraph();
    GPUX_OpCopyTextureMipToBuffer(g, tex, mip, buf, 0);
    GPU_GraphSubmit(g);
    GPU_GraphWait(g);
    double sum = 0.0;
    if (channels_f32) {
        const float* f = (const float*)buf->data;
        for (uint64_t i = 0; i < bytes / 4; ++i) sum += (double)f[i];
    } else {
        const uint16_t* h = (const uint16_t*)buf->data;           /* fp16 payloads: checksum of the raw bit patterns */
        for (uint64_t i = 0; i < bytes / 2; ++i) sum += (double)h[i];
    }
    GPU_DestroyGraph(g);
    GPU_DestroyBuffer(buf);
    return sum;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s cube_strip.hdr [irr lut spec min_size [width height]]\n", argv[0]); return 2; }
    uint32_t irr = argc > 2 ? (uint32_t)atoi(argv[2]) : 32, lut = argc > 3 ? (uint32_t)atoi(argv[3]) : 256;
    uint32_t spec = argc > 4 ? (uint32_t)atoi(argv[4]) : 256, min_size = argc > 5 ? (uint32_t)atoi(argv[5]) : 16;
    uint32_t width = argc > 6 ? (uint32_t)atoi(argv[6]) : 320, height = argc > 7 ? (uint32_t)atoi(argv[7]) : 180;

    GPU_Init(NULL);                                                          /* main.cpp:35 */
    GPUX_EnableOpTiming(1);
    GPU_Texture* tex_env_cube = PBR_MakeTextureFromHDRIFile(argv[1]);        /* main.cpp:47 */
    if (!tex_env_cube) return 1;
    printf("env_size %u\nenv_mips %u\n", tex_env_cube->width, tex_env_cube->mip_level_count);

    PBR_IBLMaps maps;
    PBR_MakeIBLMaps(&maps, irr, lut, spec);                                  /* render.cpp:794-796 */
    PBR_GenIrradianceMap(tex_env_cube, maps.irradiance_map);                 /* render.cpp:505-540 */
    PBR_GenPrefilteredEnvMap(tex_env_cube, maps.tex_specular_env_map, min_size);   /* render.cpp:542-589 */
    PBR_GenBRDFIntegrationMap(maps.brdf_lut);                                /* render.cpp:591-619 */

    printf("env_mip0_sum %.9e\n", checksum_texture_mip(tex_env_cube, 0, 1));
    printf("env_last_mip_sum %.9e\n", checksum_texture_mip(tex_env_cube, tex_env_cube->mip_level_count - 1, 1));
    printf("irradiance_sum %.9e\n", checksum_texture_mip(maps.irradiance_map, 0, 1));
    uint32_t size = spec;
    for (uint32_t m = 0; m < maps.tex_specular_env_map->mip_level_count && size >= min_size; ++m, size /= 2)
        printf("specular_mip%u_sum %.9e\n", m, checksum_texture_mip(maps.tex_specular_env_map, m, 1));
    printf("lut_bits_sum %.9e\n", checksum_texture_mip(maps.brdf_lut, 0, 0));

    /* a flat synthetic G-buffer: lower half = rough gold floor facing +Z at NDC depth .998, upper half = sky */
    PBR_GBuffer gb;
    PBR_MakeGBuffer(&gb, width, height, GPU_Format_RGBA16F);                 /* render.cpp:680-693 */
    {
        size_t n = (size_t)width * height;
        uint8_t* rgba = (uint8_t*)malloc(n * 4);
        float* depth = (float*)malloc(n * 4);
        GPU_Graph* g = GPU_MakeGraph();
        struct { GPU_Texture* t; uint8_t v[4]; } planes[4] = {
            {gb.base_color, {255, 195, 86, 255}}, {gb.normal, {128, 128, 255, 255}}, {gb.orm, {255, 90, 255, 255}}, {gb.emissive, {0, 0, 0, 255}}};
        for (int p = 0; p < 4; ++p) {
            for (size_t i = 0; i < n; ++i) memcpy(rgba + 4 * i, planes[p].v, 4);
            GPU_Buffer* b = GPU_MakeBuffer((uint32_t)(n * 4), GPU_BufferFlag_CPU, rgba);
            GPU_OpCopyBufferToTexture(g, b, planes[p].t, 0, 1, 0);
            GPU_GraphSubmit(g); GPU_GraphWait(g);
            GPU_DestroyBuffer(b);
        }
        for (size_t i = 0; i < n; ++i) depth[i] = (i / width) >= height / 2 ? 0.998f : 1.0f;
        GPU_Buffer* b = GPU_MakeBuffer((uint32_t)(n * 4), GPU_BufferFlag_CPU, depth);
        GPU_OpCopyBufferToTexture(g, b, gb.depth, 0, 1, 0);
        GPU_GraphSubmit(g); GPU_GraphWait(g);
        GPU_DestroyBuffer(b); GPU_DestroyGraph(g);
        free(rgba); free(depth);
    }
    PBR_LightingPass* lp = PBR_MakeLightingPass(&gb, &maps, width, height);  /* render.cpp:716-723, 829-871 */
    PBR_Globals globals;
    float pos[3] = {0.f, 0.f, 5.f};                                          /* main.cpp:18 */
    PBR_FillGlobals(&globals, pos, NULL, 75.f, (float)width / (float)height, 0.02f, 10000.f, 56.5f, 97.f, 0);   /* main.cpp:21,85-88 */
    GPU_Graph* graph = GPU_MakeGraph();
    PBR_RecordLightingPass(lp, graph, &globals, 0, 0);                       /* render.cpp:1119-1127 */
    GPU_GraphSubmit(graph);
    GPU_GraphWait(graph);
    for (uint32_t i = 0; i < GPUX_GraphTimedOpCount(graph); ++i)
        printf("time_ms %s %.6f\n", GPUX_GraphTimedOpName(graph, i), GPUX_GraphTimedOpMs(graph, i));
    printf("lit_bits_sum %.9e\n", checksum_texture_mip(gb.lighting_result, 0, 0));

    /* ---- per-frame chain: lighting -> TAA -> bloom -> final, three frames (velocity buffers stay zero: a still camera) ---- */
    PBR_PostProcess* pp = PBR_MakePostProcess(&gb, width, height, GPU_Format_RGBA8UN);
    for (uint32_t frame = 0; frame < 3; ++frame) {
        PBR_FillGlobals(&globals, pos, NULL, 75.f, (float)width / (float)height, 0.02f, 10000.f, 56.5f, 97.f, frame);
        PBR_RecordLightingPass(lp, graph, &globals, 0, 0);                   /* render.cpp:1119-1127 */
        PBR_RecordTaaResolve(pp, graph, frame);                              /* render.cpp:1131-1137 */
        PBR_RecordBloom(pp, graph, frame);                                   /* render.cpp:1139-1176 */
        PBR_RecordFinalPostProcessBloom(pp, graph, frame);                   /* render.cpp:1181-1187 */
        GPU_GraphSubmit(graph);
        GPU_GraphWait(graph);
    }
    for (uint32_t i = 0; i < GPUX_GraphTimedOpCount(graph); ++i)
        printf("time_ms %s %.6f\n", GPUX_GraphTimedOpName(graph, i), GPUX_GraphTimedOpMs(graph, i));
    printf("taa_bits_sum %.9e\n", checksum_texture_mip(PBR_PostTaaOutput(pp, 0), 0, 0));
    printf("bloom_bits_sum %.9e\n", checksum_texture_mip(PBR_PostBloomUpscale(pp), 0, 0));
    {
        GPU_Texture* bb = PBR_PostBackbuffer(pp);
        uint32_t bytes = (uint32_t)GPUX_TextureMipBytes(bb, 0);
        GPU_Buffer* buf = GPU_MakeBuffer(bytes, GPU_BufferFlag_CPU, NULL);
        GPU_Graph* g = GPU_MakeGraph();
        GPU_OpCopyTextureToBuffer(g, bb, buf);
        GPU_GraphSubmit(g); GPU_GraphWait(g);
        const uint8_t* px = (const uint8_t*)buf->data;
        double sum = 0.0;
        for (uint32_t i = 0; i < bytes; ++i) sum += (double)px[i];
        printf("frame_bytes_sum %.9e\n", sum);
        if (argc > 8) {                                                      /* binary PPM, rows top-down as rendered */
            FILE* f = fopen(argv[8], "wb");
            if (!f) { perror(argv[8]); return 1; }
            fprintf(f, "P6\n%u %u\n255\n", width, height);
            for (uint32_t i = 0; i < width * height; ++i) fwrite(px + 4 * i, 1, 3, f);
            fclose(f);
        }
        GPU_DestroyGraph(g); GPU_DestroyBuffer(buf);
    }

    /* ---- voxel light grid: clear, a ground slab of lit voxels, three sweeps (directions y, z, x) ---- */
    {
        PBR_Lightgrid* lg = PBR_MakeLightgrid(128);                          /* render.cpp:678 */
        GPU_Texture* grid = PBR_LightgridTexture(lg);
        uint32_t bytes = (uint32_t)GPUX_TextureMipBytes(grid, 0);
        uint16_t* vox = (uint16_t*)calloc(bytes, 1);
        for (uint32_t z = 0; z < 4; ++z)                                     /* RGBA16F: (0.5, 0.25, 0.125, 1) in the four lowest layers */
            for (uint32_t i = 0; i < 128 * 128; ++i) {
                uint16_t* v = vox + ((size_t)z * 128 * 128 + i) * 4;
                v[0] = 0x3800; v[1] = 0x3400; v[2] = 0x3000; v[3] = 0x3c00;
            }
        GPU_Buffer* up = GPU_MakeBuffer(bytes, GPU_BufferFlag_CPU, vox);
        GPU_Graph* g = GPU_MakeGraph();
        PBR_RecordLightgridClear(lg, g);                                     /* render.cpp:1028 */
        GPUX_OpCopyBufferToTextureMip(g, up, 0, grid, 0);
        for (int k = 0; k < 3; ++k) PBR_RecordLightgridSweep(lg, g);         /* render.cpp:1061-1072 */
        GPU_GraphSubmit(g); GPU_GraphWait(g);
        for (uint32_t i = 0; i < GPUX_GraphTimedOpCount(g); ++i)
            printf("time_ms %s %.6f\n", GPUX_GraphTimedOpName(g, i), GPUX_GraphTimedOpMs(g, i));
        printf("lightgrid_bits_sum %.9e\n", checksum_texture_mip(grid, 0, 0));
        GPU_DestroyGraph(g); GPU_DestroyBuffer(up); free(vox);
        PBR_DestroyLightgrid(lg);
    }
    PBR_DestroyPostProcess(pp);

    GPU_DestroyGraph(graph);
    PBR_DestroyLightingPass(lp);
    PBR_DestroyGBuffer(&gb);
    PBR_DestroyIBLMaps(&maps);
    GPU_DestroyTexture(tex_env_cube);
    GPU_WaitUntilIdle();
    GPU_Deinit();                                                            /* main.cpp:113 */
    printf("ok 1\n");
    return 0;
}
