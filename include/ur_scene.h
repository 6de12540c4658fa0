/*
 * ur_scene.h — scene -> instance AABB extraction, host only: scene JSON + glTF JSON in, the cull
 * pass's ModelBounds order and boxes out. See csrc/scene.cpp for the reference citations and the two stated differences
 * (accessor min/max instead of scanning the .bin; deterministic order inside a pipeline-key group).
 */
#ifndef UR_SCENE_H
#define UR_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UR_SCENE_OK 0
#define UR_SCENE_EINVAL (-1)
#define UR_SCENE_EJSON (-2)     /* malformed JSON */
#define UR_SCENE_EEMPTY (-3)    /* no models / no renderable meshes */
#define UR_SCENE_ENOBOUNDS (-4) /* a POSITION accessor lacks min/max */
#define UR_SCENE_ECAPACITY (-5)

/* One draw command's worth of FSceneModelResource (RendererUtils.h:113-…): what the cull pass and the light set-up use. */
typedef struct ur_scene_model {
    float bounds_min[3];
    float bounds_max[3];
    float center[3];        /* mesh-box centre through World */
    float radius;           /* mesh radius x max model scale x max node scale */
    uint32_t pipeline_key;  /* BuildPipelineKey, DeferredRenderer.cpp:28-36 */
    uint32_t material_index;
    uint32_t model_index, node_order, mesh_index, primitive_index;
} ur_scene_model;

typedef struct ur_scene_summary {
    uint32_t model_count;
    float scene_center[3];
    float scene_radius;
} ur_scene_summary;

/* Number of entries of the scene JSON's "models" array (-1 on malformed JSON). */
int ur_scene_model_count(const char* scene_json);
/* The i-th model's "path" (relative to the assets root), NUL-terminated into buf; returns its length or -1. */
int ur_scene_model_path(const char* scene_json, uint32_t index, char* buf, uint32_t cap);
/* scene_json: text of Assets/Scenes/<x>.json. gltf_json[i]: text of the i-th model's .gltf (JSON form).
 * out (nullable, capacity entries): models in command order. summary (nullable): count + scene centre/radius. */
int ur_scene_extract(const char* scene_json, const char* const* gltf_json, uint32_t gltf_count, ur_scene_model* out, uint32_t capacity,
                     ur_scene_summary* summary);

#ifdef __cplusplus
}
#endif
#endif /* UR_SCENE_H */
