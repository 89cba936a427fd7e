/*
 * rtx.h — C-ABI of the MI355X-native HIP render backend for RayTracer's
 * frame-render hot path (IFrameRender::Render(Color*)).
 *
 * Every entry point is plain C: pointers, sizes, PODs. No torch / C++ types.
 * File:line citations are relative to the reference tree (DavidSolz/RayTracer).
 *
 * The POD structs below restate the reference's host layouts byte for byte
 * (sizes and offsets are asserted in tests/test_abi.py and, against the reference's own
 * structs — Object, Material, BoundingBox, Texture, Camera —, by the static_asserts of
 * oracle/ref_parts.cpp and tests/test_oracle.py)
 * so that a HIPShader : ComputeShader inside the reference can pass
 * context->objects.data() etc. straight through.
 *
 * Return convention: 0 = ok, non-zero = error (message via rtx_last_error()).
 * The reference's own convention is void + log + exit(-1)
 * (src/ComputeEnvironment.cpp:129-132); the C++ wrapper maps non-zero to that.
 * All functions are to be called from one thread per renderer (the reference
 * calls Render strictly sequentially from the main thread,
 * src/WindowManager.cpp:154-158).
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference POD layouts ------------------------------------------------ */

/* src/Vector3.h:7-201 and src/Color.h:9-145 — four floats, 16-byte aligned. */
typedef struct RtxVec4 { float x, y, z, w; } __attribute__((aligned(16))) RtxVec4;

/* src/SpatialType.h:7-15 */
enum { RTX_SPHERE = 0, RTX_PLANE = 1, RTX_DISK = 2, RTX_CUBE = 3, RTX_TRIANGLE = 4, RTX_INVALID = 5 };

/* src/Object.h:11-19 — 256 bytes: type@0 radius@4 position@16 normals@32
 * vertices@80 uvs@128 materialID@176. */
typedef struct RtxObject {
    int32_t type;
    float   radius;
    float   _pad0[2];
    RtxVec4 position;
    RtxVec4 normals[3];
    RtxVec4 vertices[3];
    RtxVec4 uvs[3];
    int32_t materialID;
    int32_t _pad1[19];
} __attribute__((aligned(256))) RtxObject;

/* src/Material.h:7-25 — 128 bytes. */
typedef struct RtxMaterial {
    RtxVec4 albedo;              /*   0 */
    RtxVec4 tint;                /*  16 */
    RtxVec4 specular;            /*  32 */
    RtxVec4 transmissionFilter;  /*  48 */
    float   specularIntensity;   /*  64 */
    float   transparency;        /*  68 */
    float   indexOfRefraction;   /*  72 */
    float   roughness;           /*  76 */
    float   metallic;            /*  80 */
    float   sheen;               /*  84 */
    float   tintRoughness;       /*  88 */
    float   clearcoatThickness;  /*  92 */
    float   clearcoatRoughness;  /*  96 */
    float   emmissionIntensity;  /* 100 (sic) */
    float   anisotropy;          /* 104 */
    float   anisotropyRotation;  /* 108 */
    int32_t textureID;           /* 112 */
    int32_t _pad[3];
} __attribute__((aligned(128))) RtxMaterial;

/* src/BoundingBox.h:7-48 — 64 bytes; leaf <=> objectID != -1. */
typedef struct RtxBoundingBox {
    int32_t objectID;
    int32_t parentID;
    int32_t leftID;
    int32_t rightID;
    RtxVec4 minimalPosition;
    RtxVec4 maximalPosition;
    int32_t _pad[4];
} __attribute__((aligned(64))) RtxBoundingBox;

/* src/Texture.h:4-9 */
typedef struct RtxTexture {
    int32_t  width;
    int32_t  height;
    int32_t  offset;    /* first texel, in uint32 words, inside textureData */
    uint32_t checksum;
} __attribute__((aligned(16))) RtxTexture;

/* src/Camera.h:9-65 — 96 bytes. */
typedef struct RtxCamera {
    RtxVec4 front;          /*  0 */
    RtxVec4 up;             /* 16 */
    RtxVec4 right;          /* 32 */
    RtxVec4 position;       /* 48 */
    float   movementSpeed;  /* 64 */
    float   rotationSpeed;  /* 68 */
    float   aspectRatio;    /* 72 */
    float   nearView;       /* 76 */
    float   farView;        /* 80 */
    float   fov;            /* 84 */
    float   pitch;          /* 88 */
    float   yaw;            /* 92 */
} RtxCamera;

/* ---- scene hand-over ------------------------------------------------------- */

/* Borrowed pointers to the five host arrays of src/RenderingContext.h:31-40 in
 * their reference layouts, read once by rtx_create (precedent: the OpenCL
 * backend uploads them once in its constructor, src/CLShader.cpp:20-49). */
typedef struct RtxSceneDesc {
    const RtxObject*      objects;      uint64_t numObjects;
    const RtxMaterial*    materials;    uint64_t numMaterials;
    const RtxBoundingBox* boxes;        uint64_t numBoxes;     /* 2N-1, pre-order; may be 0 */
    const RtxTexture*     textureInfo;  uint64_t numTextures;
    const uint32_t*       textureData;  uint64_t numTexels;
    uint32_t width;                     /* context->width  */
    uint32_t height;                    /* context->height */
    int32_t  useBVH;                    /* context->bvhAcceleration (src/ThreadedShader.cpp:15-19) */
    int32_t  _pad;
} RtxSceneDesc;

enum { RTX_TRAVERSE_WIDE   = 0,   /* DEFAULT. Own 4-wide SAH tree over the reference's leaves, NO distance culling: tests
                                     exactly the leaves the reference tests (a leaf is tested iff its own box passes,
                                     DESIGN.md 3.1), same t arithmetic, same tie-break: the reference's result by
                                     construction, for every input */
       RTX_TRAVERSE_EXACT  = 1,   /* the reference's own tree and visit order: no distance culling, no reordering */
       RTX_TRAVERSE_CULLED = 2 }; /* opt-in. WIDE's tree walked nearest child first with distance culling (per-scene
                                     slack, none on sliver geometry): same result unless a grazing triangle's computed t
                                     undercuts its own box's computed entry distance by more than the slack — a bet, not a
                                     proof (none found in 140 M rays / 3800 randomised scenes) */

typedef struct RtxConfig {
    int32_t device;        /* HIP device ordinal */
    int32_t rank;          /* this process' share of the screen tiles ...            */
    int32_t worldSize;     /* ... tile t is owned by rank (t % worldSize)            */
    int32_t tileW;         /* screen tile size in pixels; 0 -> 64                    */
    int32_t tileH;         /*                             0 -> 32                    */
    int32_t traversal;     /* RTX_TRAVERSE_*                                         */
    int32_t countStats;    /* 1: count node/leaf visits per frame (slower kernels)   */
    int32_t framesInFlight;/* frames that may overlap on the device when rtx_render is called with
                              pixels_rgba == NULL (each has its own stream and path state; the running
                              mean is still updated in frame order); 1 = strictly serial; 0 -> as many as the
                              process has hardware queues for: 3 with HIP's default GPU_MAX_HW_QUEUES=4,
                              6 with GPU_MAX_HW_QUEUES=8, 14 with 16 (an environment variable HIP reads at
                              start-up), but no more than 3 batches when frames are rendered in batches of 8 or more
                              (framesPerBatch); at most 16. Each batch in flight holds its own ray queues and path
                              state: about 125 B per pixel of this rank's tiles and frame of the batch */
    int32_t numDevices;    /* > 1: a DEVICE GROUP in this one process — devices[0..numDevices) each render the tiles   */
    int32_t devices[8];    /* of rank i of numDevices (scene flattened once, replicated), and every call that delivers
                              a frame (rtx_render with pixels, rtx_download_pixels) first gathers the members' tiles
                              into devices[0]'s framebuffer with one peer copy each (xGMI). The returned handle stands
                              for the whole group, so IFrameRender::Render(Color*) drives up to 8 GPUs with the
                              reference's Main.cpp untouched. rank/worldSize must be 0/1 then. The same ordinal may
                              appear twice (rehearsal on one GPU). 0 or 1: one device, `device`. */
    int32_t lookAhead;     /* rtx_render WITH pixels hands the frame back before it returns, so calls cannot overlap —
                              unless the renderer starts the next frames before they are asked for. After two calls in
                              a row for (camera, n) and (camera, n + 1) — the reference's loop, src/WindowManager.cpp:
                              154-158 — it starts up to `lookAhead` SETS of following frames (everything but their update of
                              the running mean) before it copies the current one out; a set is one set of launches for 1, 2,
                              4, then 8 consecutive frames (it grows while the loop lasts, so that a short loop or an early
                              wrong guess wastes little), applied to the running mean one frame per call. The next call
                              either asks for the first frame started ahead or they are all dropped unused — and cut short:
                              the library overwrites their work counters from a control stream, so their launches run out
                              at once (RtxInfo.setsCutShort). Images and work counters are the same either way.
                              0 -> up to 3, at most framesInFlight - 1 (2 with HIP's default 4 hardware queues); -1 -> never.
                              A device group looks ahead on all its members. */
    int32_t framesPerBatch;/* frames queued with pixels_rgba == NULL are collected and rendered `framesPerBatch` at a time by
                              one set of launches (longer queues: the thin ends of the launches are shared by the frames of a
                              batch); any call that needs results renders what has been collected first, so results do not
                              depend on it. 0 -> as many as fit 35 % of the free device memory with three batches in flight, at most 32 (32 at
                              3840x2160 on a 288-GB device); 1 -> every frame is launched by the call that queues it. rtx_create allocates
                              queues for 4 frames per batch in flight (the look-ahead of the synchronous loop grows them to its sets
                              of 8); the first frame that is queued grows them to whole batches (RtxInfo.deviceBytes follows).
                              Each frame of a batch in flight holds about 125 B per pixel of this rank's tiles. */
} RtxConfig;

typedef struct RtxCounters {
    uint64_t frames;          /* rtx_render calls since create / reset                     */
    uint64_t rays;            /* closest-hit queries issued (primary + bounce)             */
    uint64_t raysPerBounce[4];
    uint64_t hitsPerBounce[4];
    uint64_t innerVisits[4];  /* node steps (4-wide nodes on the culled path, binary nodes otherwise), per bounce (countStats only) */
    uint64_t leafTests[4];    /* leaf primitives intersected, per bounce (countStats only) */
    double   lastFrameMs;     /* device time of the last rtx_render call (HIP events)      */
    double   lastTraceMs;     /* ... of which traversal kernels                            */
    double   lastShadeMs;     /* ... of which shading kernels                              */
    double   totalMs;         /* device time of `frames`, overlap of frames in flight counted once */
    /* per launch of the frame pipeline, summed over `frames` (HIP events on the
     * renderer's stream): [0] primary, [1] shade0, [2] trace1, [3] shade1, ... [7] shade3 */
    double   kernelMs[8];
    uint64_t wavePasses[4];   /* traversal-loop passes of whole 64-lane waves, per bounce (countStats, culled bounce
                                 launches only): (innerVisits + leafTests) / (64 * wavePasses) = work items per lane-pass
                                 (a lane can do a node step and a leaf test in one pass) */
} RtxCounters;

/* What rtx_create decided (no reference counterpart; diagnostics for callers and benchmarks). */
typedef struct RtxInfo {
    int32_t  traversal;        /* RTX_TRAVERSE_* actually in effect (EXACT when the caller's boxes break the own tree's
                                  preconditions, or without a BVH: then `linear` is set) */
    int32_t  linear;           /* 1: no BVH, the list is walked in object order (src/ThreadedShader.cpp:15-19) */
    int32_t  bigNodes;         /* 1: 64-bit node addressing (more than 2^25 wide nodes) */
    float    cullSlack;        /* relative slack of distance culling; +inf = no culling */
    int32_t  framesInFlight;   /* frame contexts (streams) the renderer holds */
    int32_t  framesPerBatch;   /* queued frames rendered by one set of launches */
    int32_t  hwQueues;         /* GPU_MAX_HW_QUEUES as seen at create (HIP's default 4 when unset) */
    int32_t  refTreeDepth;     /* depth of the caller's tree */
    int32_t  wideTreeDepth;    /* depth of the own 4-wide tree (0 when absent) */
    int32_t  stackEntries;     /* traversal stack entries a lane can need */
    uint64_t leaves, innerNodes, wideNodes;
    double   sliverFraction;   /* triangles with longest edge^2 / (2 area) > 32 */
    uint64_t deviceBytes;      /* device memory held by this renderer */
    double   createSeconds[4]; /* validate + walk boxes, own-tree build, record emission, upload + allocation */
    int32_t  devices;          /* members of the device group this handle stands for (1: a single renderer) */
    int32_t  lookAhead;        /* frames rtx_render(pixels) starts ahead (RtxConfig.lookAhead as resolved) */
    uint64_t setsCutShort;     /* sets of frames started ahead and never asked for whose launches were made to run out at
                                  once (their work counters overwritten from a control stream); a running count, read at the
                                  time of rtx_get_info, summed over the members of a device group. Results never depend on it. */
} RtxInfo;

typedef struct RtxRenderer RtxRenderer;

/* ---- the hot path ----------------------------------------------------------- */

/* Replaces: ThreadedShader::ThreadedShader (src/ThreadedShader.cpp:4-21) /
 * CLShader::CLShader (src/CLShader.cpp:6-69): bind to a RenderingContext's
 * arrays. Flattens boxes+objects into the SoA device layout, uploads materials
 * and textures, allocates the resident framebuffer (zero-filled like
 * src/WindowManager.cpp:26-27) and the ray queues. */
int rtx_create(const RtxSceneDesc* scene, const RtxConfig* cfg, RtxRenderer** out);

/* Replaces: IFrameRender::Render(Color* _pixels) (src/IFrameRender.h:8,
 * src/ThreadedShader.cpp:186-207). Reads camera + frameCounter per call like
 * src/CLShader.cpp:160-166. pixels_rgba: host, width*height*4 floats, row-major
 * y*width+x; may be NULL to skip the device-to-host copy (the running mean
 * stays resident on the device either way): the call then only queues the frame
 * (camera and frameCounter are copied): queued frames are rendered
 * RtxConfig.framesPerBatch at a time by one set of launches, up to
 * RtxConfig.framesInFlight batches overlap on the device; every entry point that
 * returns results (a later rtx_render with pixels, rtx_download_pixels,
 * rtx_get_counters, rtx_pack_tiles ...) completes all queued frames first, in
 * call order. With pixels the call returns when the frame has been copied; in
 * the loop "same camera, frameCounter + 1, one delivery per frame" the library
 * starts the following frames before they are asked for (RtxConfig.lookAhead;
 * results and counters are not affected). With worldSize > 1 only this rank's
 * tiles are written; other pixels of pixels_rgba are left untouched. */
int rtx_render(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, float* pixels_rgba);

/* Parity probe for ThreadedShader::BVHTraverse / LinearTraverse on primary rays
 * (src/ThreadedShader.cpp:141-149,158): objectIDs[y*width+x] = sample.objectID
 * of the first traverse() of frame `frameCounter` (-1 = miss). Host buffer. */
int rtx_primary_hits(RtxRenderer* r, const RtxCamera* camera, uint32_t frameCounter, int32_t* objectIDs);

/* Parity probe for traverse() on caller-supplied rays (bounce-like, incoherent):
 * origins/directions are n*4 floats (x,y,z,ignored). Outputs (host, each may be
 * NULL): objectIDs[n], points[n*4] (sample.point), normals[n*4]. */
int rtx_trace_rays(RtxRenderer* r, const float* origins, const float* directions, uint64_t n,
                   int32_t* objectIDs, float* points, float* normals);

/* rtx_trace_rays for callers that cannot allocate. NO COUNTERPART IN THE REFERENCE: it exists for tight serving loops
 * and harnesses that forbid hipMalloc / hipFree (it is a synchronous call on the renderer's own stream, so it cannot itself
 * be recorded into a HIP graph). Same kernels, results and work counters (bounce-1 slots) as rtx_trace_rays, but
 * origins / directions / outputs are DEVICE pointers (n*4 floats, n ints, n*4, n*4; each output may be NULL) and the
 * intermediate buffers rtx_trace_rays allocates per call (work counters, hit distances, hit leaves, side list, staged
 * directions) live in one fixed file-scope device array of 14 MiB per GPU, held until the process ends. The call
 * issues no hipMalloc / hipFree (device or pinned) and returns after the results are complete.
 * Limits: n <= 524288 rays per call (a larger n is rejected with an error; split the batch). Every call on a GPU
 * shares that GPU's array, so calls that overlap on one GPU — from several threads or renderers — race: serialise them. */
int rtx_trace_rays_noalloc(RtxRenderer* r, const float* d_origins, const float* d_directions, uint64_t n,
                           int32_t* d_objectIDs, float* d_points, float* d_normals);

/* Overwrite / read the resident running-mean framebuffer (the reference's
 * `pixels`, src/WindowManager.cpp:26-27). */
int rtx_upload_pixels(RtxRenderer* r, const float* pixels_rgba);
int rtx_download_pixels(RtxRenderer* r, float* pixels_rgba);
int rtx_clear_pixels(RtxRenderer* r);

/* Present path. Replaces the pixel hand-over of WindowManager::UpdateWindow (src/WindowManager.cpp:160-166:
 * glTexSubImage2D(..., GL_RGBA, GL_FLOAT, pixels) into a GL_RGBA texture, :18 — GL clamps to [0,1] and rounds to 8
 * bits): the resident running mean is converted on the device — clamp, optional display gamma (context->gamma,
 * src/RenderingContext.h:48; <= 0: none; alpha is never gamma-corrected), round(x * 255) — and delivered as
 * width*height RGBA8 words (R in the low byte), 4 bytes per pixel instead of 16 over PCIe. The caller uploads it with
 * GL_RGBA / GL_UNSIGNED_BYTE, or writes it out as the screenshot of WindowManager::DumpContent (:212-259). Completes
 * all queued frames first (and gathers a device group), like rtx_download_pixels. */
int rtx_present_rgba8(RtxRenderer* r, float gamma, uint8_t* rgba8);

/* Page-locked host memory for the caller's pixel array (hipHostMalloc / hipHostFree behind the C ABI, so that a caller needs
 * no HIP headers). No reference counterpart: the reference's `pixels` is `new Color[w * h]` (src/WindowManager.cpp:26); a
 * maintainer may allocate it here instead. rtx_render / rtx_download_pixels / rtx_present_rgba8 recognise a page-locked
 * destination by themselves (however it was obtained: this call, hipHostMalloc, a pinned tensor) and copy the frame into it
 * directly — one DMA transfer, no staging through the library's own buffer (4K synchronous loop: 5.09 instead of 5.30 ms
 * per frame). A pageable destination works as before. rtx_alloc_host returns NULL on failure (rtx_last_error says why);
 * the memory is not zeroed. */
void* rtx_alloc_host(uint64_t bytes);
void  rtx_free_host(void* p);

int  rtx_get_info(RtxRenderer* r, RtxInfo* out);
int  rtx_get_counters(RtxRenderer* r, RtxCounters* out);
int  rtx_reset_counters(RtxRenderer* r);
void rtx_destroy(RtxRenderer* r);
const char* rtx_last_error(void);

/* ---- multi-GPU tile plumbing (no reference counterpart: new work) */

/* Device pointer of the resident framebuffer (width*height float4, row-major). */
void*    rtx_framebuffer_device_ptr(RtxRenderer* r);
/* The HIP stream every kernel of this renderer is launched on (hipStream_t). */
void*    rtx_stream(RtxRenderer* r);
/* Number of pixels this rank owns / number of tiles it owns. */
uint64_t rtx_local_pixel_count(RtxRenderer* r);
uint32_t rtx_local_tile_count(RtxRenderer* r);
/* Floats needed by a packed tile buffer of `rank` (tileCount*tileW*tileH*4). */
uint64_t rtx_packed_floats(RtxRenderer* r, int32_t rank);
/* Pack this rank's tiles from the resident framebuffer into d_packed (device,
 * tile-major: tile k of this rank at k*tileW*tileH float4, row-major inside). */
int rtx_pack_tiles(RtxRenderer* r, void* d_packed);
/* Scatter a packed tile buffer that belongs to `srcRank` into the resident
 * framebuffer of this renderer (used on the gathering rank). */
int rtx_unpack_tiles(RtxRenderer* r, const void* d_packed, int32_t srcRank);

/* ---- build info -------------------------------------------------------------- */
const char* rtx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
