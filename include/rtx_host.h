/*
 * rtx_host.h — C-ABI of the host-side scene container that feeds rtx_create().
 *
 * These are the callers / data formats on the INPUT side of the hot path
 * (SURVEY.md §8f ranks 1-2): they fill the same five arrays that the reference's
 * Configurator + serializers + BVHTree leave in RenderingContext
 * (src/RenderingContext.h:31-40), so a renderer can be built without the
 * reference's GL-dependent translation units. Inside the reference itself none
 * of this is needed: HIPShader hands context->objects.data() etc. to rtx_create
 * directly.
 *
 * Plain C, host memory only, no HIP.
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtxHostScene RtxHostScene;

/* Replaces Configurator::Configurator + Initialize + the resolution part of
 * ParseArgs (src/Configurator.cpp:3-48,144-149): width/height are rounded to a
 * multiple of 32 (min 32), camera at (500,500,-900) looking +z, aspect = w/h,
 * depth 480, default material 0 and the 1x1 white texture 0.
 * Material 0's fields the reference leaves uninitialised
 * (src/Configurator.cpp:24-36) are defined as 0 here. */
RtxHostScene* rtxh_scene_new(uint32_t requestedWidth, uint32_t requestedHeight);
void          rtxh_scene_free(RtxHostScene* s);
/* Set width/height exactly (no rounding to 32) and aspect = w/h. Only before any
 * geometry is loaded: mesh placement depends on it (src/MeshSerializer.cpp:98-100). */
int           rtxh_scene_set_resolution(RtxHostScene* s, uint32_t width, uint32_t height);

/* Replaces SceneSerializer::LoadFromFile (src/SceneSerializer.cpp:240-255) and
 * everything it pulls in: MaterialSerializer (.mtl), MeshSerializer (.obj),
 * ObjectBuilder (plane/disk/cube tessellation), BitmapReader (BITMAPV5 .bmp). */
int rtxh_scene_load_scn(RtxHostScene* s, const char* path);
/* Pieces of the above, usable on their own. */
int rtxh_scene_load_mtl(RtxHostScene* s, const char* path);
int rtxh_scene_load_obj(RtxHostScene* s, const char* path);
int rtxh_scene_material_id(RtxHostScene* s, const char* name);      /* 0 if unknown (DEFAULT_MATERIAL_ID) */
/* Append a texture (uint32 texels packed R|G<<8|B<<16|255<<24); returns its id. */
int rtxh_scene_add_texture(RtxHostScene* s, const uint32_t* texels, int32_t width, int32_t height);
int rtxh_scene_add_material(RtxHostScene* s, const RtxMaterial* m, const char* name);
/* ObjectBuilder<PLANE/DISK/CUBE>::Build (src/ObjectBuilder.h) and sphere. */
int rtxh_scene_add_plane(RtxHostScene* s, const float position[3], const float normal[3], const float scales[3], int materialID);
int rtxh_scene_add_disk(RtxHostScene* s, const float position[3], const float normal[3], float radius, int materialID);
int rtxh_scene_add_cube(RtxHostScene* s, const float position[3], const float scales[3], int materialID);
int rtxh_scene_add_sphere(RtxHostScene* s, const float position[3], float radius, int materialID);
/* Append triangles given in OBJ space; placed exactly like
 * MeshSerializer::BuildTriangles (src/MeshSerializer.cpp:96-137):
 * v*100*aspect + (w/2,h/2,depth/2), position = centroid. vertices: nv*3 floats,
 * normals: nv*3 floats (per vertex), indices: ntri*3, materialIDs: ntri (or NULL -> materialID0). */
int rtxh_scene_add_mesh(RtxHostScene* s, const float* vertices, const float* normals, uint64_t nv,
                        const uint32_t* indices, const int32_t* materialIDs, uint64_t ntri, int materialID0);
/* Synthetic height-field mesh ("bumpy wall") with exactly ntri triangles, generated
 * in OBJ space on a extentU x extentV rectangle centred at (cx,cy,cz), leaning
 * back by tiltDeg about the x axis, then placed like rtxh_scene_add_mesh.
 * Seeded (std::mt19937 raw outputs mapped by explicit arithmetic; no
 * implementation-defined distributions); materials vary per 64x64-cell patch. */
int rtxh_scene_add_heightfield(RtxHostScene* s, uint64_t ntri, uint32_t seed,
                               float extentU, float extentV, float amplitude, float tiltDeg,
                               float cx, float cy, float cz,
                               const int32_t* materialIDs, int32_t numMaterialIDs);
/* The same mesh written as Wavefront OBJ text ("v", "vn", "usemtl <name>", "f a//a b//b c//c"; the materials must
 * have names in `s`): the route the reference itself takes for meshes (src/MeshSerializer.cpp:141-243). Loading the
 * file with rtxh_scene_load_obj into a scene of the same resolution gives byte-identical objects. */
int rtxh_write_heightfield_obj(RtxHostScene* s, const char* path, uint64_t ntri, uint32_t seed,
                               float extentU, float extentV, float amplitude, float tiltDeg,
                               float cx, float cy, float cz,
                               const int32_t* materialIDs, int32_t numMaterialIDs);

/* Replaces BVHTree::BuildBVH (src/BVHTree.cpp:42-60,148-195): same tree (median
 * split by std::nth_element on position[depth%3], one object per leaf, 2N-1
 * nodes in pre-order, left subtree first), built in place and in parallel. */
int rtxh_scene_build_bvh(RtxHostScene* s, int nthreads);

/* Deep copy of caller-owned arrays (reference layouts) into a new scene. */
RtxHostScene* rtxh_scene_from_arrays(const RtxSceneDesc* arrays, const RtxCamera* camera, uint32_t depth);

/* Borrowed view for rtx_create(); valid until the scene is modified or freed. */
int rtxh_scene_desc(RtxHostScene* s, RtxSceneDesc* out);
int rtxh_scene_camera(RtxHostScene* s, RtxCamera* out);
int rtxh_scene_set_camera(RtxHostScene* s, const RtxCamera* in);
/* Camera::LookAt (src/Camera.h:51-55) and the orbit step of src/Main.cpp:68-82. */
void rtxh_camera_look_at(RtxCamera* cam, const float target[3]);
void rtxh_camera_orbit(RtxCamera* cam, float yawDeg, float pitchDeg, float radius);
/* Camera::Rotate (src/Camera.h:26-43): yaw / pitch += offset * rotationSpeed, pitch clamped to +-89 degrees, basis rebuilt. */
void rtxh_camera_rotate(RtxCamera* cam, double offsetX, double offsetY);

/* Binary scene blob (the wire format between ingest and the C-ABI). */
int rtxh_scene_save_blob(RtxHostScene* s, const char* path);
RtxHostScene* rtxh_scene_load_blob(const char* path);

const char* rtxh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
