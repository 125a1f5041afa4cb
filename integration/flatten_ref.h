// Reference-side adapter, part 1: flattens the reference's own Scene + KdTree (scene.h:19-43,
// kdtree.h:11-32) into the mrt_scene_desc of mrt.h.  Written against the reference's headers; it
// is meant to be dropped into the reference tree as src/rendering/flatten_ref.{h,cpp} and built
// with it (it uses glm and the reference's types, nothing of this repository but mrt.h).
//
//   mrtFlattenGeometry   objects, transforms, materials, checker and image textures, kd-tree, camera
//                        (needs scene.h, kdtree.h, Camera.h, Object.h only)
//   mrtFlattenLights     Scene::lights / envLights (flatten_ref_lights.cpp: needs Light.h)
#pragma once
#include "../kdtree.h"
#include "../scene.h"

#include <mrt.h>

#include <cstdint>
#include <vector>

struct MrtFlat {
    mrt_scene_desc desc {};
    std::vector<uint8_t> obj_kind;
    std::vector<uint32_t> obj_material, obj_transform, refs, out_of_tree;
    std::vector<float> tri_p, tri_n, tri_uv, sphere, texels, env_rgb;
    std::vector<mrt_transform> transforms;
    std::vector<Transform> transform_values; // transforms[i] as the reference's Transform
    std::vector<mrt_material> materials;
    std::vector<mrt_texture> textures;
    std::vector<mrt_kdnode> nodes;
    std::vector<mrt_light> lights;
    uint32_t tree_depth = 0;

    uint32_t internTransform( const Transform& t );
    void bind(); // points desc at the vectors (call after the last change to any of them)
};

// Throws std::runtime_error on a scene the backend cannot represent (cube objects, materials with a
// texture kind other than checker_texture and image_texture, a camera that is not a SimpleCamera).
void mrtFlattenGeometry( Scene* scene, KdTree* tree, MrtFlat& out );
void mrtFlattenLights( Scene* scene, MrtFlat& out );
