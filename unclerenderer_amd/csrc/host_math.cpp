// Host-side math of the hot path's callers (include/ur_host.h). Plain fp32, written against the published
// definitions of the DirectXMath functions the reference calls; no DirectXMath here.
// Built with -ffp-contract=off so results do not depend on FMA availability.

#include "../../include/ur_host.h"

#include <cmath>
#include <cstring>

namespace {

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a)
{
    const float l = std::sqrt(dot(a, a));
    return {a.x / l, a.y / l, a.z / l};
}
inline V3 load(const float* p) { return {p[0], p[1], p[2]}; }

void plane_normalize(float a, float b, float c, float d, float* out)
{
    const float l = std::sqrt((a * a + b * b) + c * c); // XMPlaneNormalize: no zero guard
    out[0] = a / l; out[1] = b / l; out[2] = c / l; out[3] = d / l;
}

} // namespace

extern "C" {

void ur_host_look_to_lh(const float eye[3], const float dir[3], const float up[3], float out[16])
{
    const V3 r2 = normalize(load(dir));
    const V3 r0 = normalize(cross(load(up), r2));
    const V3 r1 = cross(r2, r0);
    const V3 ne = {-eye[0], -eye[1], -eye[2]};
    const float d0 = dot(r0, ne), d1 = dot(r1, ne), d2 = dot(r2, ne);
    const float m[16] = {r0.x, r1.x, r2.x, 0.0f, r0.y, r1.y, r2.y, 0.0f, r0.z, r1.z, r2.z, 0.0f, d0, d1, d2, 1.0f};
    std::memcpy(out, m, sizeof(m));
}

void ur_host_look_at_lh(const float eye[3], const float at[3], const float up[3], float out[16])
{
    const float dir[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
    ur_host_look_to_lh(eye, dir, up, out);
}

void ur_host_reverse_z_projection(float fov_y, float aspect, float near_clip, float out[16])
{
    const float ys = 1.0f / std::tan(fov_y * 0.5f);
    const float xs = ys / aspect;
    const float m[16] = {xs, 0, 0, 0, 0, ys, 0, 0, 0, 0, 0, 1.0f, 0, 0, near_clip, 0};
    std::memcpy(out, m, sizeof(m));
}

void ur_host_orthographic_lh(float w, float h, float zn, float zf, float out[16])
{
    const float range = 1.0f / (zf - zn);
    const float m[16] = {2.0f / w, 0, 0, 0, 0, 2.0f / h, 0, 0, 0, 0, range, 0, 0, 0, -range * zn, 1.0f};
    std::memcpy(out, m, sizeof(m));
}

void ur_host_mat_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[i * 4 + j] = ((a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j]) + a[i * 4 + 2] * b[8 + j]) + a[i * 4 + 3] * b[12 + j];
    std::memcpy(out, r, sizeof(r));
}

int ur_host_mat_inverse(const float m[16], float out[16])
{
    // cofactor expansion in double, rounded once
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return 0;
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] / det);
    return 1;
}

void ur_host_frustum_planes(const float m[16], float planes[24])
{
    // _rc is m[(r-1)*4 + (c-1)]
    plane_normalize(m[3] + m[0], m[7] + m[4], m[11] + m[8], m[15] + m[12], planes + 0);
    plane_normalize(m[3] - m[0], m[7] - m[4], m[11] - m[8], m[15] - m[12], planes + 4);
    plane_normalize(m[3] + m[1], m[7] + m[5], m[11] + m[9], m[15] + m[13], planes + 8);
    plane_normalize(m[3] - m[1], m[7] - m[5], m[11] - m[9], m[15] - m[13], planes + 12);
    plane_normalize(m[2], m[6], m[10], m[14], planes + 16);
    plane_normalize(m[3] - m[2], m[7] - m[6], m[11] - m[10], m[15] - m[14], planes + 20);
}

int ur_host_is_aabb_in_frustum(const float planes[24], const float bmin[3], const float bmax[3])
{
    for (int i = 0; i < 6; ++i) {
        const float* p = planes + i * 4;
        const float x = p[0] >= 0.0f ? bmax[0] : bmin[0];
        const float y = p[1] >= 0.0f ? bmax[1] : bmin[1];
        const float z = p[2] >= 0.0f ? bmax[2] : bmin[2];
        if (((p[0] * x + p[1] * y) + p[2] * z) + p[3] * 1.0f < 0.0f) return 0;
    }
    return 1;
}

void ur_host_light_view_projection(const float center[3], float radius, const float light_dir[3], float out[16])
{
    const V3 d = normalize(load(light_dir));
    const float dist = radius * 2.5f;
    const float eye[3] = {center[0] + d.x * dist, center[1] + d.y * dist, center[2] + d.z * dist};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float view[16], proj[16];
    ur_host_look_at_lh(eye, center, up, view);
    const float ortho = radius * 2.0f;
    ur_host_orthographic_lh(ortho, ortho, 0.1f, radius * 5.0f, proj);
    ur_host_mat_mul(view, proj, out);
}

void ur_host_pack_culling_constants(const float view[16], const float proj[16], uint32_t model_count, uint32_t hzb_enabled,
                                    uint32_t hzb_mip_count, uint32_t hzb_width, uint32_t hzb_height, uint32_t debug_print,
                                    uint32_t out[UR_CULL_CONSTANT_DWORDS])
{
    float vp[16], planes[24];
    ur_host_mat_mul(view, proj, vp);
    ur_host_frustum_planes(vp, planes);
    std::memcpy(out, planes, sizeof(planes));
    std::memcpy(out + 24, vp, sizeof(vp));
    out[40] = model_count;
    out[41] = hzb_enabled ? 1u : 0u;
    out[42] = hzb_mip_count;
    out[43] = hzb_width;
    out[44] = hzb_height;
    out[45] = debug_print ? 1u : 0u;
}

void ur_host_fill_scene_constants(const float view[16], const float proj[16], const float camera_pos[3], float light_intensity,
                                  const float light_dir[3], const float light_color[3], const float light_view_proj[16],
                                  float shadow_strength, float shadow_bias, float shadow_w, float shadow_h, float env_mip_count,
                                  ur_scene_constants* out)
{
    ur_scene_constants c;
    std::memset(&c, 0, sizeof(c));
    const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(c.World, identity, sizeof(identity));
    std::memcpy(c.View, view, sizeof(c.View));
    if (!ur_host_mat_inverse(view, c.ViewInverse)) std::memcpy(c.ViewInverse, identity, sizeof(identity));
    std::memcpy(c.Projection, proj, sizeof(c.Projection));
    c.BaseColor[0] = c.BaseColor[1] = c.BaseColor[2] = 1.0f;
    c.LightIntensity = light_intensity;
    const V3 ld = normalize(load(light_dir));
    c.LightDirection[0] = ld.x; c.LightDirection[1] = ld.y; c.LightDirection[2] = ld.z;
    std::memcpy(c.CameraPosition, camera_pos, 12);
    std::memcpy(c.LightColor, light_color, 12);
    std::memcpy(c.LightViewProjection, light_view_proj, sizeof(c.LightViewProjection));
    c.ShadowStrength = shadow_strength;
    c.ShadowBias = shadow_bias;
    c.ShadowMapSize[0] = shadow_w; c.ShadowMapSize[1] = shadow_h;
    c.MetallicFactor = 1.0f; c.RoughnessFactor = 1.0f; c.BaseColorAlpha = 1.0f; c.AlphaCutoff = 0.5f;
    float* xf[4] = {c.BaseColorTransformOffsetScale, c.MetallicRoughnessTransformOffsetScale, c.NormalTransformOffsetScale, c.EmissiveTransformOffsetScale};
    float* xr[4] = {c.BaseColorTransformRotation, c.MetallicRoughnessTransformRotation, c.NormalTransformRotation, c.EmissiveTransformRotation};
    for (int i = 0; i < 4; ++i) {
        xf[i][2] = 1.0f; xf[i][3] = 1.0f;
        xr[i][0] = 1.0f;
    }
    c.EnvMapMipCount = env_mip_count;
    *out = c;
}

void ur_host_fill_sky_constants(const float view[16], const float proj[16], const float camera_pos[3], float sky_radius,
                                const float light_dir[3], const float light_color[3], ur_sky_constants* out)
{
    ur_sky_constants c;
    std::memset(&c, 0, sizeof(c));
    const float world[16] = {sky_radius, 0, 0, 0, 0, sky_radius, 0, 0, 0, 0, sky_radius, 0, camera_pos[0], camera_pos[1], camera_pos[2], 1.0f};
    std::memcpy(c.World, world, sizeof(world));
    std::memcpy(c.View, view, sizeof(c.View));
    std::memcpy(c.Projection, proj, sizeof(c.Projection));
    std::memcpy(c.CameraPosition, camera_pos, 12);
    const V3 ld = normalize(load(light_dir));
    c.LightDirection[0] = ld.x; c.LightDirection[1] = ld.y; c.LightDirection[2] = ld.z;
    std::memcpy(c.LightColor, light_color, 12);
    *out = c;
}

void ur_host_direction_from_euler_degrees(float pitch_deg, float yaw_deg, float out[3])
{
    const float DegToRad = 3.14159265f / 180.0f;
    const float p = pitch_deg * DegToRad, y = yaw_deg * DegToRad;
    out[0] = std::cos(p) * std::sin(y);
    out[1] = std::sin(p);
    out[2] = std::cos(p) * std::cos(y);
}

void ur_host_camera_forward_from_euler_degrees(float pitch_deg, float yaw_deg, float out[3])
{
    // (0,0,1) * RotationRollPitchYaw(pitch, yaw, 0) = (cosP sinY, -sinP, cosP cosY), then normalised
    const float DegToRad = 3.14159265358979f / 180.0f; // XMConvertToRadians
    const float p = pitch_deg * DegToRad, y = yaw_deg * DegToRad;
    const V3 f = normalize(V3{std::cos(p) * std::sin(y), -std::sin(p), std::cos(p) * std::cos(y)});
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}

void ur_host_light_direction_roundtrip(const float json_dir[3], float out[3])
{
    const V3 d = normalize(load(json_dir));
    const float pitch = std::asin(d.y), yaw = std::atan2(d.x, d.z);
    const V3 f = normalize(V3{std::cos(pitch) * std::sin(yaw), -std::sin(pitch), std::cos(pitch) * std::cos(yaw)});
    out[0] = f.x; out[1] = f.y; out[2] = f.z;
}

} // extern "C"
