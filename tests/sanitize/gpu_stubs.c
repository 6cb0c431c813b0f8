#include <stddef.h>
#include <stdint.h>
/* link stubs: the harness never reaches the GPU entry points */
void* GPU_MakeTexture(int f, uint32_t w, uint32_t h, uint32_t d, int fl, const void* p) { (void)f; (void)w; (void)h; (void)d; (void)fl; (void)p; return 0; }
void* GPUX_MakeCubemapFromEquirect(const void* a, uint32_t w, uint32_t h, uint32_t s, int f) { (void)a; (void)w; (void)h; (void)s; (void)f; return 0; }
uint64_t GPUX_TextureMipBytes(const void* t, uint32_t m) { (void)t; (void)m; return 0; }
void* GPU_MakeBuffer(uint32_t s, int f, const void* d) { (void)s; (void)f; (void)d; return 0; }
void GPU_DestroyBuffer(void* b) { (void)b; }
void* GPU_MakeGraph(void) { return 0; }
void GPU_DestroyGraph(void* g) { (void)g; }
void GPU_GraphSubmit(void* g) { (void)g; }
void GPU_GraphWait(void* g) { (void)g; }
void GPUX_OpCopyTextureMipToBuffer(void* g, void* t, uint32_t m, void* b, uint32_t o) { (void)g; (void)t; (void)m; (void)b; (void)o; }
