/*
 * pbr_rgbe.c -- Radiance .hdr decoder + HDRI strip -> cubemap texture (host side, C11).
 *
 * Produces what the reference's input stage hands to GPU_MakeTexture:
 *   asset_import.cpp:17-27  MakeTextureFromHDRIFile: stbi_loadf(path, &x, &y, &comp, 4); assert(y == x*6);
 *                           GPU_MakeTexture(RGBA32F, x, x, 1, Cubemap|HasMipmaps, data)
 *   third_party/stb_image.h:7157-7286 (header + flat / new-RLE scanlines), :7130-7155 (RGBE -> float)
 * Written from the Radiance format description; stb_image is not used or copied.
 */
#include "pbr_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct Cursor { const unsigned char* p; size_t n, at; } Cursor;

static int next_byte(Cursor* c) { return c->at < c->n ? c->p[c->at++] : -1; }

/* reads one '\n'-terminated header line (without the newline) */
static int read_line(Cursor* c, char* out, size_t cap) {
    size_t len = 0;
    for (;;) {
        int b = next_byte(c);
        if (b < 0) { if (len == 0) return 0; break; }
        if (b == '\n') break;
        if (len + 1 < cap) out[len++] = (char)b;
    }
    out[len] = 0;
    return 1;
}

static void rgbe_to_float(const unsigned char px[4], float* out) {
    if (px[3]) {
        float scale = ldexpf(1.0f, (int)px[3] - 136);        /* 2^(e-128) / 256 */
        out[0] = (float)px[0] * scale; out[1] = (float)px[1] * scale; out[2] = (float)px[2] * scale;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
    out[3] = 1.0f;
}

float* PBR_DecodeHDR(const void* bytes, size_t size, int* w, int* h, const char** err) {
    static const char* dummy;
    if (!err) err = &dummy;
    *err = NULL;
    Cursor c = {(const unsigned char*)bytes, size, 0};
    char line[1024];
    if (!read_line(&c, line, sizeof line) || (strcmp(line, "#?RADIANCE") && strcmp(line, "#?RGBE"))) { *err = "not a Radiance HDR file"; return NULL; }
    int have_format = 0;
    for (;;) {
        if (!read_line(&c, line, sizeof line)) { *err = "truncated header"; return NULL; }
        if (line[0] == 0) break;
        if (!strcmp(line, "FORMAT=32-bit_rle_rgbe")) have_format = 1;
    }
    if (!have_format) { *err = "unsupported HDR format (need 32-bit_rle_rgbe)"; return NULL; }
    if (!read_line(&c, line, sizeof line)) { *err = "missing resolution line"; return NULL; }
    int width = 0, height = 0;
    char* q = line;
    if (strncmp(q, "-Y ", 3)) { *err = "unsupported HDR orientation (need -Y h +X w)"; return NULL; }
    height = (int)strtol(q + 3, &q, 10);
    while (*q == ' ') ++q;
    if (strncmp(q, "+X ", 3)) { *err = "unsupported HDR orientation (need -Y h +X w)"; return NULL; }
    width = (int)strtol(q + 3, NULL, 10);
    if (width <= 0 || height <= 0 || width > (1 << 24) || height > (1 << 24)) { *err = "bad HDR dimensions"; return NULL; }

    float* img = (float*)malloc((size_t)width * (size_t)height * 4 * sizeof(float));
    if (!img) { *err = "out of memory"; return NULL; }
    size_t pixel = 0, total = (size_t)width * (size_t)height;
    int rle = !(width < 8 || width >= 32768);
    unsigned char* row = rle ? (unsigned char*)malloc((size_t)width * 4) : NULL;

    while (rle && pixel < total) {
        int b0 = next_byte(&c), b1 = next_byte(&c), b2 = next_byte(&c);
        if (b0 != 2 || b1 != 2 || (b2 & 0x80)) {
            /* old-style file: these three bytes + one more are the first pixel; everything after is flat */
            unsigned char px[4] = {(unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)next_byte(&c)};
            rgbe_to_float(px, img);
            pixel = 1;
            rle = 0;
            break;
        }
        int len = (b2 << 8) | next_byte(&c);
        if (len != width) { *err = "corrupt HDR: scanline length mismatch"; free(row); free(img); return NULL; }
        for (int ch = 0; ch < 4; ++ch) {
            int x = 0;
            while (x < width) {
                int count = next_byte(&c);
                if (count < 0) { *err = "corrupt HDR: truncated scanline"; free(row); free(img); return NULL; }
                if (count > 128) {
                    int value = next_byte(&c);
                    count -= 128;
                    if (count > width - x) { *err = "corrupt HDR: bad run"; free(row); free(img); return NULL; }
                    while (count--) row[4 * x++ + ch] = (unsigned char)value;
                } else {
                    if (count == 0 || count > width - x) { *err = "corrupt HDR: bad literal run"; free(row); free(img); return NULL; }
                    while (count--) row[4 * x++ + ch] = (unsigned char)next_byte(&c);
                }
            }
        }
        for (int x = 0; x < width; ++x) rgbe_to_float(row + 4 * x, img + (pixel + (size_t)x) * 4);
        pixel += (size_t)width;
    }
    free(row);
    for (; pixel < total; ++pixel) {
        unsigned char px[4];
        for (int k = 0; k < 4; ++k) { int b = next_byte(&c); px[k] = (unsigned char)(b < 0 ? 0 : b); }
        rgbe_to_float(px, img + pixel * 4);
    }
    *w = width; *h = height;
    return img;
}

GPU_Texture* PBR_MakeTextureFromHDRIMemory(const void* bytes, size_t size) {
    int x = 0, y = 0;
    const char* err = NULL;
    float* data = PBR_DecodeHDR(bytes, size, &x, &y, &err);
    if (!data) { fprintf(stderr, "GPU-ERROR: PBR_MakeTextureFromHDRI: %s\n", err ? err : "decode failed"); return NULL; }
    if (y != x * 6) {                                       /* asset_import.cpp:21 */
        fprintf(stderr, "GPU-ERROR: PBR_MakeTextureFromHDRI: expected a vertical strip of 6 square faces (got %dx%d)\n", x, y);
        free(data);
        return NULL;
    }
    GPU_Texture* t = GPU_MakeTexture(GPU_Format_RGBA32F, (uint32_t)x, (uint32_t)x, 1, GPU_TextureFlag_Cubemap | GPU_TextureFlag_HasMipmaps, data);
    free(data);
    return t;
}

GPU_Texture* PBR_MakeTextureFromHDRIFile(const char* filepath) {
    FILE* f = fopen(filepath, "rb");
    if (!f) { fprintf(stderr, "GPU-ERROR: PBR_MakeTextureFromHDRIFile: cannot open %s\n", filepath); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* buf = malloc((size_t)n);
    GPU_Texture* t = NULL;
    if (buf && fread(buf, 1, (size_t)n, f) == (size_t)n) t = PBR_MakeTextureFromHDRIMemory(buf, (size_t)n);
    free(buf);
    fclose(f);
    return t;
}
