// Scene -> instance AABB extraction: the caller side of the cull pass. Produces the ModelBounds array
// (world AABB per draw command) and the scene centre / radius that size the shadow frustum and the sky sphere.
//
// Reference: RendererUtils::CreateSceneModelsFromJson (Source/Render/RendererUtils.cpp:298-543: per-mesh bounds
// :46-82, World = NodeWorld * Scale * RotationRollPitchYaw * Translation :402-410, 8-corner world AABB :418-440, one
// model resource per primitive section sharing the MESH-level box :460-522, scene bounds :277-286,533-540),
// FGltfLoader (Scene/GltfLoader.cpp:407-505 TRS / quaternion matrices, :498-502 left-handed conversion
// MirrorZ * M * MirrorZ, :557-593 node walk from scenes[scene].nodes, :823 vertex z flip), FSceneJsonLoader
// (Scene/SceneJsonLoader.cpp:421-429) and the command order of CreateGpuDrivenResources
// (DeferredRenderer.cpp:28-36,3301-3366: sort by pipeline key, then texture descriptor).
//
// Differences, stated: (1) mesh bounds come from the POSITION accessors' min/max in the glTF JSON (required by the
// glTF spec and equal to the vertex extrema the reference scans), so no .bin is read — Sponza's 10.8 MB blob is absent
// from the reference checkout; (2) the reference's second sort key is a D3D12 descriptor address (load-order dependent)
// under an unstable std::sort; here the order is a STABLE sort by (pipeline key, glTF material index, original
// order) — deterministic, same grouping.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ur_scene.h"

namespace {

// ---- minimal JSON DOM -----------------------------------------------------------------------------------------------------
struct JValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue* find(const char* key) const
    {
        if (type != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const JValue* at(size_t i) const { return (type == Array && i < arr.size()) ? &arr[i] : nullptr; }
};

struct JParser {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    bool lit(const char* s)
    {
        const size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) { p += n; return true; }
        return false;
    }
    std::string string()
    {
        std::string out;
        ++p; // opening quote
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': out += '?'; p += (end - p > 4) ? 4 : 0; break; // code points are irrelevant to this loader
                default: out += *p; break;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p < end) ++p; else ok = false;
        return out;
    }
    JValue value(int depth = 0)
    {
        JValue v;
        ws();
        if (p >= end || depth > 256) { ok = false; return v; }
        if (*p == '{') {
            v.type = JValue::Object;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return v; }
            while (ok && p < end) {
                ws();
                if (p >= end || *p != '"') { ok = false; break; }
                std::string k = string();
                ws();
                if (p >= end || *p != ':') { ok = false; break; }
                ++p;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.type = JValue::Array;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return v; }
            while (ok && p < end) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') {
            v.type = JValue::String;
            v.str = string();
        } else if (lit("true")) { v.type = JValue::Bool; v.b = true;
        } else if (lit("false")) { v.type = JValue::Bool;
        } else if (lit("null")) {
        } else {
            char* e = nullptr;
            v.num = std::strtod(p, &e);
            if (e == p) ok = false;
            v.type = JValue::Number;
            p = e;
        }
        return v;
    }
};

bool parse_json(const char* text, JValue& out)
{
    if (!text) return false;
    JParser ps{text, text + std::strlen(text)};
    out = ps.value();
    return ps.ok;
}

bool vec3(const JValue* v, float out[3])
{
    if (!v || v->type != JValue::Array || v->arr.size() != 3) return false;
    for (int i = 0; i < 3; ++i) out[i] = (float)v->arr[i].num;
    return true;
}

// ---- 4x4 matrices, glTF convention: column-major storage, column vectors (M[col*4+row]) -----------------------------------
struct M4 { float m[16]; };
M4 identity() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}; }
M4 mul(const M4& A, const M4& B) // A * B
{
    M4 R{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += A.m[k * 4 + r] * B.m[c * 4 + k];
            R.m[c * 4 + r] = s;
        }
    return R;
}
M4 from_trs(const JValue& node) // GltfLoader.cpp:426-495
{
    if (const JValue* mv = node.find("matrix"); mv && mv->type == JValue::Array && mv->arr.size() == 16) {
        M4 M{};
        for (int i = 0; i < 16; ++i) M.m[i] = (float)mv->arr[i].num;
        return M;
    }
    float t[3] = {0, 0, 0}, s[3] = {1, 1, 1}, q[4] = {0, 0, 0, 1};
    vec3(node.find("translation"), t);
    vec3(node.find("scale"), s);
    if (const JValue* r = node.find("rotation"); r && r->type == JValue::Array && r->arr.size() == 4)
        for (int i = 0; i < 4; ++i) q[i] = (float)r->arr[i].num;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const M4 R = {{1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y + w * z), 2.0f * (x * z - w * y), 0.0f, 2.0f * (x * y - w * z), 1.0f - 2.0f * (x * x + z * z),
                   2.0f * (y * z + w * x), 0.0f, 2.0f * (x * z + w * y), 2.0f * (y * z - w * x), 1.0f - 2.0f * (x * x + y * y), 0.0f, 0.0f, 0.0f, 0.0f, 1.0f}};
    const M4 T = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, t[0], t[1], t[2], 1}};
    const M4 S = {{s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1}};
    return mul(mul(T, R), S);
}
M4 to_left_handed(const M4& M) // MirrorZ * M * MirrorZ
{
    const M4 Z = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1}};
    return mul(Z, mul(M, Z));
}

struct Node { int mesh; M4 world; };
void walk(const JValue& nodes, int64_t index, const M4& parent, size_t mesh_count, std::vector<Node>& out, int depth = 0)
{
    const JValue* n = nodes.at((size_t)index);
    if (!n || n->type != JValue::Object || depth > 512) return;
    const M4 world = mul(parent, to_left_handed(from_trs(*n)));
    const JValue* mi = n->find("mesh");
    if (mi && mi->type == JValue::Number && mi->num >= 0 && (size_t)mi->num < mesh_count) out.push_back({(int)mi->num, world});
    if (const JValue* ch = n->find("children"); ch && ch->type == JValue::Array)
        for (const JValue& c : ch->arr) walk(nodes, (int64_t)c.num, world, mesh_count, out, depth + 1);
}

// Row-vector product used from here on (DirectXMath side): XMFLOAT4X4.m[r][c] = M[r*4+c] of the column-major array.
struct RM { float m[4][4]; };
RM rm_mul(const RM& A, const RM& B)
{
    RM R{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) R.m[i][j] = ((A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j]) + A.m[i][2] * B.m[2][j]) + A.m[i][3] * B.m[3][j];
    return R;
}
void transform_coord(const RM& M, const float p[3], float out[3])
{
    float r[4];
    for (int j = 0; j < 4; ++j) r[j] = ((p[0] * M.m[0][j] + p[1] * M.m[1][j]) + p[2] * M.m[2][j]) + M.m[3][j];
    for (int j = 0; j < 3; ++j) out[j] = r[j] / r[3];
}
RM rotation_roll_pitch_yaw(float pitch, float yaw, float roll) // XMMatrixRotationRollPitchYaw
{
    const float cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw), cr = std::cos(roll), sr = std::sin(roll);
    RM M{};
    M.m[0][0] = cr * cy + sr * sp * sy; M.m[0][1] = sr * cp; M.m[0][2] = sr * sp * cy - cr * sy;
    M.m[1][0] = cr * sp * sy - sr * cy; M.m[1][1] = cr * cp; M.m[1][2] = sr * sy + cr * sp * cy;
    M.m[2][0] = cp * sy; M.m[2][1] = -sp; M.m[2][2] = cp * cy;
    M.m[3][3] = 1.0f;
    return M;
}

struct MeshBounds { float mn[3], mx[3], center[3], radius; std::vector<int> materials; };

uint32_t pipeline_key(const JValue* materials, int material) // BuildPipelineKey, DeferredRenderer.cpp:28-36
{
    if (!materials || material < 0) return 0;
    const JValue* m = materials->at((size_t)material);
    if (!m) return 0;
    const JValue* pbr = m->find("pbrMetallicRoughness");
    const uint32_t normal = m->find("normalTexture") ? 1u : 0u;
    const uint32_t mr = (pbr && pbr->find("metallicRoughnessTexture")) ? 1u : 0u;
    const uint32_t base = (pbr && pbr->find("baseColorTexture")) ? 1u : 0u;
    const uint32_t emissive = m->find("emissiveTexture") ? 1u : 0u;
    const JValue* am = m->find("alphaMode");
    const uint32_t mask = (am && am->type == JValue::String && am->str == "MASK") ? 1u : 0u;
    return normal | (mr << 1) | (base << 2) | (emissive << 3) | (mask << 4);
}

} // namespace

extern "C" {

int ur_scene_model_count(const char* scene_json)
{
    JValue root;
    if (!parse_json(scene_json, root)) return -1;
    const JValue* models = root.find("models");
    return (models && models->type == JValue::Array) ? (int)models->arr.size() : 0;
}

int ur_scene_model_path(const char* scene_json, uint32_t index, char* buf, uint32_t cap)
{
    JValue root;
    if (!parse_json(scene_json, root) || !buf || cap == 0) return -1;
    const JValue* models = root.find("models");
    const JValue* m = models ? models->at(index) : nullptr;
    const JValue* path = m ? m->find("path") : nullptr;
    if (!path || path->type != JValue::String) return -1;
    std::snprintf(buf, cap, "%s", path->str.c_str());
    return (int)path->str.size();
}

int ur_scene_extract(const char* scene_json, const char* const* gltf_json, uint32_t gltf_count, ur_scene_model* out, uint32_t capacity,
                     ur_scene_summary* summary)
{
    JValue scene;
    if (!parse_json(scene_json, scene)) return UR_SCENE_EJSON;
    const JValue* models = scene.find("models");
    if (!models || models->type != JValue::Array || models->arr.empty()) return UR_SCENE_EEMPTY;
    if (gltf_count != models->arr.size()) return UR_SCENE_EINVAL;

    std::vector<ur_scene_model> all;
    float smin[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, smax[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    for (size_t mi = 0; mi < models->arr.size(); ++mi) {
        const JValue& model = models->arr[mi];
        float position[3] = {0, 0, 0}, euler[3] = {0, 0, 0}, scale[3] = {1, 1, 1};
        vec3(model.find("translate"), position);
        vec3(model.find("rotate_euler"), euler);
        vec3(model.find("scale"), scale);
        JValue gltf;
        if (!parse_json(gltf_json[mi], gltf)) return UR_SCENE_EJSON;
        const JValue* meshes = gltf.find("meshes");
        const JValue* accessors = gltf.find("accessors");
        if (!meshes || meshes->type != JValue::Array || !accessors) continue;

        // mesh-level bounds = union over the primitives' POSITION min/max, z negated (GltfLoader.cpp:823)
        std::vector<MeshBounds> mb(meshes->arr.size());
        for (size_t k = 0; k < meshes->arr.size(); ++k) {
            MeshBounds& B = mb[k];
            for (int a = 0; a < 3; ++a) { B.mn[a] = 3.402823466e38f; B.mx[a] = -3.402823466e38f; }
            const JValue* prims = meshes->arr[k].find("primitives");
            if (!prims || prims->type != JValue::Array) continue;
            for (const JValue& pr : prims->arr) {
                const JValue* attrs = pr.find("attributes");
                const JValue* pos = attrs ? attrs->find("POSITION") : nullptr;
                const JValue* acc = pos ? accessors->at((size_t)pos->num) : nullptr;
                float lo[3], hi[3];
                if (!acc || !vec3(acc->find("min"), lo) || !vec3(acc->find("max"), hi)) return UR_SCENE_ENOBOUNDS;
                const float zlo = -hi[2], zhi = -lo[2];
                lo[2] = zlo; hi[2] = zhi;
                for (int a = 0; a < 3; ++a) { B.mn[a] = std::min(B.mn[a], lo[a]); B.mx[a] = std::max(B.mx[a], hi[a]); }
                const JValue* mat = pr.find("material");
                B.materials.push_back(mat ? (int)mat->num : -1);
            }
            float ext2 = 0.0f;
            for (int a = 0; a < 3; ++a) { B.center[a] = 0.5f * (B.mn[a] + B.mx[a]); ext2 += (B.mx[a] - B.mn[a]) * (B.mx[a] - B.mn[a]); }
            B.radius = std::max(std::sqrt(ext2) * 0.5f, 1.0f);
        }

        std::vector<Node> nodes;
        const JValue* gnodes = gltf.find("nodes");
        const JValue* scenes = gltf.find("scenes");
        const JValue* sidx = gltf.find("scene");
        if (gnodes && gnodes->type == JValue::Array && scenes && scenes->type == JValue::Array) {
            const JValue* sc = scenes->at(sidx ? (size_t)sidx->num : 0);
            const JValue* roots = sc ? sc->find("nodes") : nullptr;
            if (roots && roots->type == JValue::Array)
                for (const JValue& r : roots->arr) walk(*gnodes, (int64_t)r.num, identity(), mb.size(), nodes);
        }
        if (nodes.empty())
            for (size_t k = 0; k < mb.size(); ++k) nodes.push_back({(int)k, identity()});

        const float deg = 3.14159265358979f / 180.0f;
        RM S{}, T{};
        S.m[0][0] = scale[0]; S.m[1][1] = scale[1]; S.m[2][2] = scale[2]; S.m[3][3] = 1.0f;
        for (int i = 0; i < 4; ++i) T.m[i][i] = 1.0f;
        T.m[3][0] = position[0]; T.m[3][1] = position[1]; T.m[3][2] = position[2];
        const RM R = rotation_roll_pitch_yaw(euler[0] * deg, euler[1] * deg, euler[2] * deg);
        const float max_scale = std::max(std::fabs(scale[0]), std::max(std::fabs(scale[1]), std::fabs(scale[2])));
        const JValue* materials = gltf.find("materials");

        for (size_t ni = 0; ni < nodes.size(); ++ni) {
            const Node& nd = nodes[ni];
            const MeshBounds& B = mb[(size_t)nd.mesh];
            RM NW; // ToFloat4x4: m[r][c] = M[r*4+c]
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) NW.m[r][c] = nd.world.m[r * 4 + c];
            const RM World = rm_mul(rm_mul(rm_mul(NW, S), R), T);
            float bmin[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, bmax[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
            for (int corner = 0; corner < 8; ++corner) {
                const float p[3] = {(corner & 1) ? B.mx[0] : B.mn[0], (corner & 2) ? B.mx[1] : B.mn[1], (corner & 4) ? B.mx[2] : B.mn[2]};
                float w[3];
                transform_coord(World, p, w);
                for (int a = 0; a < 3; ++a) { bmin[a] = std::min(bmin[a], w[a]); bmax[a] = std::max(bmax[a], w[a]); }
            }
            float node_scale = 0.0f; // ComputeMaxScale over the node matrix columns (RendererUtils.cpp:288-295)
            for (int c = 0; c < 3; ++c) node_scale = std::max(node_scale, std::sqrt(NW.m[0][c] * NW.m[0][c] + NW.m[1][c] * NW.m[1][c] + NW.m[2][c] * NW.m[2][c]));
            float center[3];
            transform_coord(World, B.center, center);
            const float radius = B.radius * max_scale * node_scale;
            const size_t sections = std::max<size_t>(1, B.materials.size());
            for (size_t s = 0; s < sections; ++s) {
                ur_scene_model M{};
                for (int a = 0; a < 3; ++a) { M.bounds_min[a] = bmin[a]; M.bounds_max[a] = bmax[a]; M.center[a] = center[a]; }
                M.radius = radius;
                M.material_index = s < B.materials.size() ? (uint32_t)B.materials[s] : 0xFFFFFFFFu;
                M.pipeline_key = pipeline_key(materials, s < B.materials.size() ? B.materials[s] : -1);
                M.model_index = (uint32_t)mi; M.node_order = (uint32_t)ni; M.mesh_index = (uint32_t)nd.mesh; M.primitive_index = (uint32_t)s;
                for (int a = 0; a < 3; ++a) { smin[a] = std::min(smin[a], center[a] - radius); smax[a] = std::max(smax[a], center[a] + radius); }
                all.push_back(M);
            }
        }
    }
    if (all.empty()) return UR_SCENE_EEMPTY;
    std::stable_sort(all.begin(), all.end(), [](const ur_scene_model& a, const ur_scene_model& b) {
        if (a.pipeline_key != b.pipeline_key) return a.pipeline_key < b.pipeline_key;
        return a.material_index < b.material_index;
    });
    if (summary) {
        summary->model_count = (uint32_t)all.size();
        float e2 = 0.0f;
        for (int a = 0; a < 3; ++a) { summary->scene_center[a] = 0.5f * (smin[a] + smax[a]); e2 += (smax[a] - smin[a]) * (smax[a] - smin[a]); }
        summary->scene_radius = std::max(std::sqrt(e2) * 0.5f, 1.0f);
    }
    if (out) {
        if (capacity < all.size()) return UR_SCENE_ECAPACITY;
        std::memcpy(out, all.data(), all.size() * sizeof(ur_scene_model));
    }
    return UR_SCENE_OK;
}

} // extern "C"
