// sanitizer harness for the file-facing host code (run on CPU only): random / truncated / bit-flipped .hdr inputs through PBR_DecodeHDR,
// and encode -> decode round trips
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
float* PBR_DecodeHDR(const void* bytes, size_t size, int* w, int* h, const char** err);
void* PBR_EncodeHDR(const float* rgba, int w, int h, size_t* out_size);
static uint32_t rs = 12345;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
int main(void) {
    long decoded = 0, rejected = 0;
    for (int iter = 0; iter < 20000; ++iter) {
        int w = 1 + rnd() % 40, h = 1 + rnd() % 12;
        if (iter % 7 == 0) w = 8 + rnd() % 300;           /* RLE-eligible widths */
        float* img = (float*)malloc((size_t)w * h * 16);
        for (int i = 0; i < w * h * 4; ++i) img[i] = (i % 4 == 3) ? 1.0f : ldexpf((float)(rnd() % 1000) / 1000.0f, (int)(rnd() % 30) - 15);
        size_t n = 0;
        uint8_t* file = (uint8_t*)PBR_EncodeHDR(img, w, h, &n);
        if (!file) return 2;
        int dw, dh; const char* err = NULL;
        float* back = PBR_DecodeHDR(file, n, &dw, &dh, &err);
        if (!back || dw != w || dh != h) { fprintf(stderr, "round trip failed: %s\n", err ? err : "?"); return 3; }
        for (int i = 0; i < w * h; ++i) for (int c = 0; c < 3; ++c) {
            float a = img[i * 4 + c], b = back[i * 4 + c], m = fmaxf(fmaxf(img[i * 4], img[i * 4 + 1]), img[i * 4 + 2]);
            if (fabsf(a - b) > m / 128.0f + 1e-30f) { fprintf(stderr, "value %g -> %g (max %g)\n", a, b, m); return 4; }
        }
        free(back);
        /* corrupt: truncate, flip bytes, or splice garbage */
        uint8_t* bad = (uint8_t*)malloc(n + 64);
        memcpy(bad, file, n);
        size_t bn = n;
        switch (rnd() % 4) {
        case 0: bn = rnd() % (n + 1); break;
        case 1: for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) bad[rnd() % n] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 2: for (size_t k = rnd() % n; k < n; ++k) bad[k] = (uint8_t)rnd(); break;
        default: { size_t p = rnd() % n; memmove(bad + p + 16, bad + p, n - p); for (int k = 0; k < 16; ++k) bad[p + k] = (uint8_t)rnd(); bn = n + 16; }
        }
        uint8_t* exact = (uint8_t*)malloc(bn ? bn : 1);     /* exact-size heap copy: ASan sees any over-read */
        memcpy(exact, bad, bn);
        float* r = PBR_DecodeHDR(exact, bn, &dw, &dh, &err);
        if (r) { decoded++; free(r); } else rejected++;
        free(exact); free(bad); free(file); free(img);
    }
    printf("ok: %ld corrupt inputs decoded, %ld rejected\n", decoded, rejected);
    return 0;
}
