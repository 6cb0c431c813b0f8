/*
 * ref_thirdparty.c -- builds oracle/_ref/libref_thirdparty.so (TEST INFRASTRUCTURE).
 *
 * Compiles the reference's own vendored single-header C libraries WHERE THEY LIE under
 * /root/reference/third_party (include path given by oracle/Makefile; nothing is copied):
 *   - stb_image.h v2.28   : Radiance .hdr decode used by asset_import.cpp:19 (stbi_loadf, 4 comps)
 *   - HandmadeMath.h v2.0 : the matrix maths behind utils/camera.h:95-120 and render.cpp:962-991
 * and exports thin entry points used by oracle/gen_golden.py to emit golden vectors
 * (tests/golden/) for our own RGBE decoder and our own camera/Globals code.
 * The few lines below that call HMM_* restate the call sequence of utils/camera.h:103-120
 * (world_from_view, view_from_world, LH_ZO perspective, inverses) and render.cpp:962-991.
 */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_ONLY_HDR
#define STBI_NO_STDIO
#include "stb_image.h"
#include "HandmadeMath.h"

#include <string.h>
#include <stdint.h>

/* RGBE decode through stbi_loadf_from_memory with req_comp = 4. Returns 0 on success. */
int ref_hdr_decode(const unsigned char* bytes, int len, int* w, int* h, float* out_rgba, int out_capacity_floats) {
    int comp = 0;
    float* data = stbi_loadf_from_memory(bytes, len, w, h, &comp, 4);
    if (!data) return 1;
    int n = (*w) * (*h) * 4;
    if (out_rgba && n <= out_capacity_floats) memcpy(out_rgba, data, sizeof(float) * (size_t)n);
    stbi_image_free(data);
    return 0;
}

typedef struct RefGlobals {  /* render.h:122-136 */
    HMM_Mat4 clip_space_from_world;
    HMM_Mat4 clip_space_from_view;
    HMM_Mat4 world_space_from_clip;
    HMM_Mat4 view_space_from_clip;
    HMM_Mat4 view_space_from_world;
    HMM_Mat4 world_space_from_view;
    HMM_Mat4 sun_space_from_world;
    HMM_Mat4 old_clip_space_from_world;
    HMM_Vec4 sun_direction;
    HMM_Vec3 camera_pos;
    float frame_idx_mod_59;
    float lightgrid_scale;
    uint32_t visualize_lightgrid;
} RefGlobals;

int ref_globals_size(void) { return (int)sizeof(RefGlobals); }

/* default_ori != 0: ori = HMM_QFromAxisAngle_RH((1,0,0), -PI/2) as in utils/camera.h:45
 * (CAMERA_VIEW_SPACE_IS_POSITIVE_Y_DOWN is defined in common.h:6); else ori = quat xyzw given. */
void ref_fill_globals(const float pos[3], const float ori_xyzw[4], int default_ori,
                      float fov_deg, float aspect, float z_near, float z_far,
                      float sun_angle_x_deg, float sun_angle_y_deg, uint32_t frame_idx, void* out552) {
    HMM_Vec3 p = HMM_V3(pos[0], pos[1], pos[2]);
    HMM_Quat ori = default_ori ? HMM_QFromAxisAngle_RH(HMM_V3(1, 0, 0), -HMM_PI32 / 2.f)
                               : HMM_Q(ori_xyzw[0], ori_xyzw[1], ori_xyzw[2], ori_xyzw[3]);
    /* camera.h:103-120 with lazy_pos/lazy_ori converged to pos/ori */
    HMM_Mat4 world_from_view = HMM_MulM4(HMM_Translate(p), HMM_QToM4(ori));
    HMM_Mat4 view_from_world = HMM_MulM4(HMM_QToM4(HMM_InvQ(ori)), HMM_Translate(HMM_MulV3F(p, -1.f)));
    HMM_Mat4 clip_from_view = HMM_Perspective_LH_ZO(HMM_AngleDeg(fov_deg), aspect, z_near, z_far);
    HMM_Mat4 view_from_clip = HMM_InvGeneralM4(clip_from_view);
    HMM_Mat4 clip_from_world = HMM_MulM4(clip_from_view, view_from_world);
    HMM_Mat4 world_from_clip = HMM_InvGeneralM4(clip_from_world);

    /* render.cpp:959-971 */
    const float sun_half_size = 40.f;
    const float lightgrid_extent = 40.f;
    HMM_Mat4 sun_ori = HMM_Rotate_RH(HMM_AngleDeg(sun_angle_x_deg),
                                     HMM_V3(cosf(HMM_AngleDeg(sun_angle_y_deg)), sinf(HMM_AngleDeg(sun_angle_y_deg)), 0.f));
    HMM_Mat4 sun_space_from_world = HMM_InvGeneralM4(sun_ori);
    sun_space_from_world = HMM_MulM4(HMM_Orthographic_RH_ZO(-sun_half_size, sun_half_size, -sun_half_size, sun_half_size,
                                                            -sun_half_size, sun_half_size), sun_space_from_world);
    HMM_Vec3 sun_dir = HMM_MulM4V4(sun_ori, HMM_V4(0, 0, -1, 0)).XYZ;

    RefGlobals g;
    memset(&g, 0, sizeof g);
    g.clip_space_from_world = clip_from_world;
    g.clip_space_from_view = clip_from_view;
    g.world_space_from_clip = world_from_clip;
    g.view_space_from_clip = view_from_clip;
    g.view_space_from_world = view_from_world;
    g.world_space_from_view = world_from_view;
    g.sun_space_from_world = sun_space_from_world;
    g.old_clip_space_from_world = clip_from_world;       /* frame_idx == 0 branch, render.cpp:985 */
    g.sun_direction.XYZ = sun_dir;
    g.camera_pos = p;
    g.frame_idx_mod_59 = (float)(frame_idx % 59);
    g.lightgrid_scale = 1.f / lightgrid_extent;
    g.visualize_lightgrid = 0;
    memcpy(out552, &g, sizeof g);
}
