/*
 * pbr_hdrio.c -- file-format extensions around the hot path (the reference has only the strip loader
 * of asset_import.cpp:17-27): equirectangular .hdr -> cubemap, and a Radiance RGBE writer so that the computed
 * cubemaps can be written in the very format MakeTextureFromHDRIFile reads.
 */
#include "pbr_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

GPU_Texture* PBR_MakeTextureFromEquirectHDRIMemory(const void* bytes, size_t size, uint32_t face_size) {
    int w = 0, h = 0;
    const char* err = NULL;
    float* data = PBR_DecodeHDR(bytes, size, &w, &h, &err);
    if (!data) { fprintf(stderr, "GPU-ERROR: PBR_MakeTextureFromEquirectHDRI: %s\n", err ? err : "decode failed"); return NULL; }
    GPU_Texture* t = GPUX_MakeCubemapFromEquirect(data, (uint32_t)w, (uint32_t)h, face_size, 0);
    free(data);
    return t;
}

GPU_Texture* PBR_MakeTextureFromEquirectHDRIFile(const char* filepath, uint32_t face_size) {
    FILE* f = fopen(filepath, "rb");
    if (!f) { fprintf(stderr, "GPU-ERROR: PBR_MakeTextureFromEquirectHDRIFile: cannot open %s\n", filepath); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* buf = malloc((size_t)n);
    GPU_Texture* t = NULL;
    if (buf && fread(buf, 1, (size_t)n, f) == (size_t)n) t = PBR_MakeTextureFromEquirectHDRIMemory(buf, (size_t)n, face_size);
    free(buf);
    fclose(f);
    return t;
}

/* Radiance RGBE: shared exponent of the largest component, 8-bit mantissas truncated (value = m * 2^(e-136)) */
static void float_to_rgbe(const float* rgb, unsigned char out[4]) {
    float v = rgb[0] > rgb[1] ? rgb[0] : rgb[1];
    if (rgb[2] > v) v = rgb[2];
    if (!(v > 1e-32f)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    int e;
    float m = frexpf(v, &e);                     /* v = m * 2^e, m in [0.5, 1) */
    float scale = m * 256.0f / v;
    for (int k = 0; k < 3; ++k) {
        float x = rgb[k] > 0.0f ? rgb[k] * scale : 0.0f;
        int q = (int)x;
        out[k] = (unsigned char)(q > 255 ? 255 : q);
    }
    out[3] = (unsigned char)(e + 128);
}

void* PBR_EncodeHDR(const float* rgba, int w, int h, size_t* out_size) {
    if (!rgba || w <= 0 || h <= 0 || !out_size) return NULL;
    char head[128];
    int hl = snprintf(head, sizeof head, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    /* Widths 8..32767 use the new-style scanline container (marker 2,2,hi,lo + four channel planes) with literal
     * runs only, so that no pixel can be mistaken for a marker; other widths are flat (stb_image.h:7216 reads them so). */
    int planar = (w >= 8 && w < 32768);
    size_t per_row = planar ? 4 + 4 * ((size_t)w + ((size_t)w + 127) / 128) : (size_t)w * 4;
    size_t n = (size_t)hl + per_row * (size_t)h;
    unsigned char* buf = (unsigned char*)malloc(n);
    unsigned char* row = (unsigned char*)malloc((size_t)w * 4);
    if (!buf || !row) { free(buf); free(row); return NULL; }
    memcpy(buf, head, (size_t)hl);
    unsigned char* p = buf + hl;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) float_to_rgbe(rgba + 4 * ((size_t)y * w + x), row + 4 * x);
        if (!planar) { memcpy(p, row, (size_t)w * 4); p += (size_t)w * 4; continue; }
        *p++ = 2; *p++ = 2; *p++ = (unsigned char)(w >> 8); *p++ = (unsigned char)(w & 255);
        for (int ch = 0; ch < 4; ++ch)
            for (int x = 0; x < w; x += 128) {
                int cnt = w - x < 128 ? w - x : 128;
                *p++ = (unsigned char)cnt;                       /* literal run (<= 128) */
                for (int k = 0; k < cnt; ++k) *p++ = row[4 * (x + k) + ch];
            }
    }
    free(row);
    *out_size = (size_t)(p - buf);
    return buf;
}

int PBR_WriteHDRFile(const char* filepath, const float* rgba, int w, int h) {
    size_t n = 0;
    void* bytes = PBR_EncodeHDR(rgba, w, h, &n);
    if (!bytes) return 1;
    FILE* f = fopen(filepath, "wb");
    int rc = 2;
    if (f) { rc = fwrite(bytes, 1, n, f) == n ? 0 : 3; fclose(f); }
    free(bytes);
    return rc;
}

int PBR_WriteCubeStripHDR(const char* filepath, GPU_Texture* cube, uint32_t mip_level) {
    if (!cube || cube->format != GPU_Format_RGBA32F || cube->layer_count != 6 || mip_level >= cube->mip_level_count) return 1;
    uint64_t bytes = GPUX_TextureMipBytes(cube, mip_level);
    uint32_t size = cube->width >> mip_level; if (size < 1) size = 1;
    GPU_Buffer* buf = GPU_MakeBuffer((uint32_t)bytes, GPU_BufferFlag_CPU, NULL);
    if (!buf) return 2;
    GPU_Graph* g = GPU_MakeGraph();
    GPUX_OpCopyTextureMipToBuffer(g, cube, mip_level, buf, 0);
    GPU_GraphSubmit(g);
    GPU_GraphWait(g);
    int rc = PBR_WriteHDRFile(filepath, (const float*)buf->data, (int)size, (int)size * 6);   /* faces stacked top to bottom */
    GPU_DestroyGraph(g);
    GPU_DestroyBuffer(buf);
    return rc;
}
