/*
 * ur_host.h — host-side math of the callers of the hot path: the code that fills the constant blocks the four
 * passes receive. Plain fp32 C, no DirectXMath (the reference takes these from DirectXMath, which is outside the
 * repository: SURVEY.md §8c). Row-major matrices, row-vector convention.
 */
#ifndef UR_HOST_H
#define UR_HOST_H

#include <stdint.h>

#include "ur_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

/* XMMatrixLookToLH as used by FCamera::GetViewMatrix (Source/Scene/Camera.cpp:23-31). */
void ur_host_look_to_lh(const float eye[3], const float dir[3], const float up[3], float out[16]);
/* XMMatrixLookAtLH (RendererUtils.cpp:1129). */
void ur_host_look_at_lh(const float eye[3], const float at[3], const float up[3], float out[16]);
/* Reverse-Z infinite projection, FCamera::GetProjectionMatrix (Source/Scene/Camera.cpp:33-47). */
void ur_host_reverse_z_projection(float fov_y, float aspect, float near_clip, float out[16]);
/* XMMatrixOrthographicLH (RendererUtils.cpp:1133). */
void ur_host_orthographic_lh(float w, float h, float zn, float zf, float out[16]);
void ur_host_mat_mul(const float a[16], const float b[16], float out[16]);
/* General 4x4 inverse (XMMatrixInverse); returns 0 when singular. */
int ur_host_mat_inverse(const float m[16], float out[16]);

/* BuildFrustumPlanesFromMatrix (RendererUtils.cpp:1151-1190): L, R, B, T, plane 4 = column 3 alone, plane 5 =
 * column 4 - column 3; each divided by |xyz| with no zero guard, so the reverse-Z infinite projection makes plane 4
 * (NaN,NaN,NaN,+inf) — kept, it never rejects. */
void ur_host_frustum_planes(const float view_proj[16], float planes[24]);
/* IsAabbInCameraFrustum (RendererUtils.cpp:1192-1218). */
int ur_host_is_aabb_in_frustum(const float planes[24], const float bmin[3], const float bmax[3]);
/* BuildDirectionalLightViewProjection (RendererUtils.cpp:1117-1137). */
void ur_host_light_view_projection(const float center[3], float radius, const float light_dir[3], float out[16]);

/* The 46 root constants of FRenderer::DispatchGpuCulling (Renderer.cpp:411-429). */
void ur_host_pack_culling_constants(const float view[16], const float proj[16], uint32_t model_count, uint32_t hzb_enabled,
                                    uint32_t hzb_mip_count, uint32_t hzb_width, uint32_t hzb_height, uint32_t debug_print,
                                    uint32_t out[UR_CULL_CONSTANT_DWORDS]);

/* RendererUtils::UpdateSceneConstants (RendererUtils.cpp:1029-1088) for the fields the lighting pass reads; the
 * material fields keep FSceneConstants' defaults (RendererUtils.h:41-79). */
void ur_host_fill_scene_constants(const float view[16], const float proj[16], const float camera_pos[3], float light_intensity,
                                  const float light_dir[3], const float light_color[3], const float light_view_proj[16],
                                  float shadow_strength, float shadow_bias, float shadow_w, float shadow_h, float env_mip_count,
                                  ur_scene_constants* out);
/* FDeferredRenderer::UpdateSkyConstants + RendererUtils::UpdateSkyConstants (DeferredRenderer.cpp:3789-3801,
 * RendererUtils.cpp:1090-1115): World = scale(radius) * translate(camera). */
void ur_host_fill_sky_constants(const float view[16], const float proj[16], const float camera_pos[3], float sky_radius,
                                const float light_dir[3], const float light_color[3], ur_sky_constants* out);

/* Scene JSON conventions: BuildDirectionFromEulerDegrees (Scene/SceneJsonLoader.cpp:257-269); camera forward from
 * (pitch, yaw) degrees via RotationRollPitchYaw (Core/Application.cpp:896-902); and the light vector the renderer
 * ends up with after the app's asin/atan2 round trip (Core/Application.cpp:236-242,1225-1230), i.e. (d.x,-d.y,d.z). */
void ur_host_direction_from_euler_degrees(float pitch_deg, float yaw_deg, float out[3]);
void ur_host_camera_forward_from_euler_degrees(float pitch_deg, float yaw_deg, float out[3]);
void ur_host_light_direction_roundtrip(const float json_dir[3], float out[3]);

#ifdef __cplusplus
}
#endif
#endif /* UR_HOST_H */
