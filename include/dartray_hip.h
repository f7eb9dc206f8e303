/* dartray_hip.h -- C ABI of the MI355X-native DartRay hot path.
 *
 * The reference (brendan-duncan/dartray, pure Dart) has NO native boundary; the
 * coarse Dart-level seam this library sits behind is
 *
 *     abstract class Renderer { Future<OutputImage> render(Scene scene); ... }
 *                                              (lib/core/renderer.dart:27-35)
 *
 * called once per task from DartRay.worldEnd (lib/dartray/dartray.dart:574).
 * A Dart `HipSamplerRenderer extends Renderer` flattens the
 * Scene (BVHAccel.nodes/primitives are public fields, bvh_accel.dart:486-487,
 * :533-538) into the POD structs below and calls these entry points through
 * dart:ffi.  Every entry point cites the reference interface it replaces.
 *
 * Conventions: plain C, POD structs, little endian.  The caller owns every
 * host buffer for the duration of the call; the library owns device memory.
 * All functions return DR_OK (0) or a negative error code; dr_last_error()
 * gives the message (the Dart shim turns it into LogSevere -> Exception ->
 * completeError, log.dart:42-47, dartray.dart:573-583).  Calls are blocking
 * unless they take a stream.  One DrScene per GPU / rank, one calling thread
 * per DrScene.
 */
#ifndef DARTRAY_HIP_H
#define DARTRAY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_OK 0
#define DR_ERR_INVALID (-1)
#define DR_ERR_HIP (-2)
#define DR_ERR_NO_DEVICE (-3)
#define DR_ERR_UNSUPPORTED (-4)

/* _LinearBVHNode (lib/accelerators/bvh_accel.dart:533-538): bounds, offset
 * (first primitive for a leaf, second-child index for an interior node; the
 * first child is index+1, :432), nPrimitives (0 => interior), split axis. */
typedef struct DrBvhNode {
  float bmin[3];
  float bmax[3];
  uint32_t offset;
  uint16_t nprims;
  uint8_t axis;
  uint8_t pad;
} DrBvhNode; /* 32 bytes */

/* MatteMaterial with constant textures (lib/materials/matte_material.dart:41-65). */
#define DR_MATERIAL_MATTE 0  /* MatteMaterial (lib/materials/matte_material.dart): kd, sigma */
#define DR_MATERIAL_MIRROR 1 /* MirrorMaterial (mirror_material.dart:38-55): kr */
#define DR_MATERIAL_GLASS 2  /* GlassMaterial (glass_material.dart:44-69): kr, kt, index */
#define DR_MATERIAL_PLASTIC 3 /* PlasticMaterial (plastic_material.dart:43-70): kd, ks in kr, roughness in index */
typedef struct DrMaterial {
  int32_t type; /* DR_MATERIAL_*; specular materials are traced by the PathIntegrator only */
  float kd[3];
  float kr[3];
  float kt[3];
  double sigma; /* matte: 0 => Lambertian, else OrenNayar(Kd, sigma) (matte_material.dart:54-61); a Dart double */
  double index; /* glass: the constant 'index' texture's value (a Dart double) */
} DrMaterial;

#define DR_LIGHT_DIFFUSE_AREA 0 /* DiffuseAreaLight (lib/lights/diffuse_area_light.dart) */
#define DR_LIGHT_INFINITE 1     /* InfiniteAreaLight (lib/lights/infinite_area_light.dart) */
#define DR_LIGHT_POINT 2        /* PointLight (lib/lights/point_light.dart): a delta light */
#define DR_LIGHT_SPOT 3         /* SpotLight (lib/lights/spot_light.dart): a delta light */
#define DR_LIGHT_DISTANT 4      /* DistantLight (lib/lights/distant_light.dart): a delta light */
#define DR_LIGHT_SPOT_COS 5     /* DR_LIGHT_SPOT whose cone_width / cone_falloff_start hold the two COSINES a constructed SpotLight
                                 * keeps (cosTotalWidth, cosFalloffStart: spot_light.dart:46-47) instead of the constructor's
                                 * degrees -- what a host that only sees the Light object can marshal bit for bit */

/* One entry of Scene.lights.  kind DR_LIGHT_DIFFUSE_AREA: DiffuseAreaLight + its
 * ShapeSet (diffuse_area_light.dart:36-43, lib/core/light/shape_set.dart:24-51);
 * the triangle list is in ShapeSet order (i.e. after the LIFO refine
 * reversal).  kind DR_LIGHT_INFINITE: InfiniteAreaLight; L is the factor
 * _radiance() applies (infinite_area_light.dart:180-182), env_index selects the
 * radiance map. */
typedef struct DrAreaLight {
  float L[3];
  int32_t nsamples;
  uint32_t first_tri; /* into light_tris */
  uint32_t ntris;
  uint32_t kind;
  uint32_t env_index; /* into env_maps */
  float position[3];  /* DR_LIGHT_POINT / _SPOT: lightPos = lightToWorld(0,0,0) (point_light.dart:36-39); L = intensity.
                       * DR_LIGHT_DISTANT: lightDir = normalize(lightToWorld(dir)) (distant_light.dart:38-42); L = radiance */
  float pad;
  float world_to_light[16];  /* DR_LIGHT_SPOT: Light.worldToLight (falloff is evaluated in light space, spot_light.dart:54-70) */
  double cone_width, cone_falloff_start;  /* DR_LIGHT_SPOT: total width and falloff start, degrees (spot_light.dart:42-48) */
} DrAreaLight;

/* InfiniteAreaLight's radiance map (f32 RGB texels, TEXTURE_REPEAT) and Light.lightToWorld / worldToLight
 * (lib/core/light.dart:28-34).  texels: either level 0 of the light's MIPMap (MIPMap.pyramid[0], lib/core/mipmap.dart:139 -- always
 * a power-of-two size) or the image as MIPMap.texture receives it (already multiplied by L, infinite_area_light.dart:44-49): a width
 * or height that is no power of two is resampled up to the next one exactly as that constructor does (mipmap.dart:71-138: four-tap
 * Lanczos, s then t, clamped at 0).  The Distribution2D over luminance x sin(theta) (infinite_area_light.dart:283-307) is rebuilt by
 * the library.  At most 16384 texels a side. */
typedef struct DrEnvMap {
  const float* texels; /* [height][width][3] */
  int32_t width, height;
  float light_to_world[16];
  float world_to_light[16];
} DrEnvMap;

typedef struct DrLightTri {
  uint32_t v[3]; /* vertex indices; v[0] == DR_PRIM_QUADRIC: v[1] indexes DrSceneDesc.quadrics */
  uint32_t reverse_orientation; /* bit 0: Shape.reverseOrientation; bit 1: the mesh has uvs (vert_uvs) */
} DrLightTri;

/* objectToWorld of a TriangleMesh that carries per-vertex normals / tangents: they stay in object space and are
 * transformed at shading time (lib/shapes/triangle.dart:303-317). */
typedef struct DrMeshXform {
  float object_to_world[16];
  float world_to_object[16];
} DrMeshXform;
#define DR_SHADING_N 1u  /* tri_shading bits: the primitive's mesh has 'N' */
#define DR_SHADING_S 2u  /* ... 'S' */
#define DR_SHADING_UV 4u /* ... 'uv' / 'st' */

/* Quadric shapes (lib/shapes/sphere.dart:23-38, lib/shapes/disk.dart:23-29).  Unlike triangle meshes they keep
 * their objectToWorld Transform and transform the RAY per test (transform.dart:180-196).  A primitive whose
 * tri_idx[3*i] is DR_PRIM_QUADRIC is the quadric tri_idx[3*i+1]; its material / light / orientation come from
 * the same per-primitive tables as a triangle's. */
#define DR_PRIM_QUADRIC 0xFFFFFFFFu
#define DR_QUADRIC_SPHERE 1
#define DR_QUADRIC_DISK 2
typedef struct DrQuadric {
  int32_t kind; /* DR_QUADRIC_* */
  int32_t pad;
  float object_to_world[16]; /* Shape.objectToWorld.m, row-major */
  float world_to_object[16]; /* Shape.objectToWorld.mInv == worldToObject.m (dartray.dart:383-384) */
  /* constructor arguments as Dart doubles -- sphere: radius, z0, z1, phimax [deg] (sphere.dart:313-321);
   * disk: height, radius, innerradius, phimax [deg] (disk.dart:157-165) */
  double params[4];
} DrQuadric;

/* Flattened Scene (lib/core/scene.dart:26-45): aggregate + lights. */
typedef struct DrSceneDesc {
  const DrBvhNode* nodes; /* BVHAccel.nodes, depth-first (bvh_accel.dart:419-437) */
  uint64_t nnodes;
  const float* verts; /* world-space f32 xyz (lib/shapes/triangle_mesh.dart:29-36) */
  uint64_t nverts;
  const uint32_t* tri_idx; /* ntris*3, BVHAccel.primitives order */
  uint64_t ntris;
  const uint32_t* tri_material; /* per primitive: index into materials */
  const int32_t* tri_light;     /* per primitive: area-light index or -1 (geometric_primitive.dart:63-65) */
  const uint8_t* tri_reverse;   /* per primitive: Shape.reverseOrientation (shape.dart:29) */
  const DrMaterial* materials;
  uint32_t nmaterials;
  const DrAreaLight* lights; /* Scene.lights order */
  uint32_t nlights;
  const DrLightTri* light_tris;
  uint32_t nlight_tris;
  uint32_t bvh_depth; /* max depth of the tree; 0 = unknown: dr_scene_create measures it */
  const DrEnvMap* env_maps; /* at most one infinite light is supported */
  uint32_t nenv_maps;
  const DrQuadric* quadrics; /* spheres / disks referenced from tri_idx and light_tris */
  uint32_t nquadrics;
  /* optional per-vertex shading data (triangle_mesh.dart:195-203; Triangle.getShadingGeometry triangle.dart:271-364,
   * Triangle.getUVs :247-263): arrays indexed like verts (entries of meshes without the attribute are ignored),
   * and per primitive which attributes its mesh has / which transform it uses.  All NULL: no mesh has any. */
  const float* vert_normals;   /* nverts*3, OBJECT space */
  const float* vert_tangents;  /* nverts*3, OBJECT space */
  const float* vert_uvs;       /* nverts*2 */
  const uint8_t* tri_shading;  /* ntris: DR_SHADING_* bits */
  const uint32_t* tri_xform;   /* ntris: index into mesh_xforms (read when DR_SHADING_N or _S is set) */
  const DrMeshXform* mesh_xforms;
  uint32_t nmesh_xforms;
} DrSceneDesc;

typedef struct DrScene DrScene;

/* Ray (lib/core/ray.dart:27-47) and the hit record of
 * GeometricPrimitive.intersect (geometric_primitive.dart:47-61). */
typedef struct DrRay {
  float o[3];
  float d[3];
  double tmin;
  double tmax;
} DrRay;

typedef struct DrHit {
  int32_t prim; /* index in BVH primitive order; -1 = miss.  any-hit: 0 = occluded, -1 = free */
  int32_t pad;
  double t;
  double b1;
  double b2;
} DrHit;

/* Camera state (lib/core/projective_camera.dart:27-32): PerspectiveCamera (cameras/perspective_camera.dart:93-132),
 * OrthographicCamera (cameras/orthographic_camera.dart:52-80) -- both ProjectiveCameras with their own
 * raster_to_camera -- and EnvironmentCamera (cameras/environment_camera.dart:42-52), which only uses camera_to_world
 * and the film resolution. */
#define DR_CAMERA_PERSPECTIVE 0
#define DR_CAMERA_ORTHOGRAPHIC 1
#define DR_CAMERA_ENVIRONMENT 2
typedef struct DrCamera {
  float raster_to_camera[16]; /* row-major, Matrix4x4.data order (matrix4x4.dart:170-176) */
  float camera_to_world[16];
  double lens_radius;    /* ProjectiveCamera.lensRadius / focalDistance and Camera.shutterOpen / shutterClose are Dart */
  double focal_distance; /* doubles (projective_camera.dart:31-32, camera.dart): a lens radius of 0.8 is 0.8, not its */
  double shutter_open;   /* f32 neighbour                                                                          */
  double shutter_close;
  int32_t type; /* DR_CAMERA_* */
  int32_t pad;
} DrCamera;

/* ImageFilm + Filter (lib/film/image_film.dart:51-97). */
typedef struct DrFilm {
  int32_t xres, yres;
  double crop[4];              /* cropWindow: Dart doubles (image_film.dart:61-65 rounds xres * crop up) */
  double filter_xw, filter_yw; /* Filter.xWidth / yWidth (filter.dart:33-37), Dart doubles */
  float filter_table[256];     /* 16x16, image_film.dart:74-82 */
} DrFilm;

#define DR_INTEGRATOR_DIRECT_ALL 0 /* DirectLightingIntegrator, strategy "all" (direct_lighting_integrator.dart) */
#define DR_INTEGRATOR_PATH 1       /* PathIntegrator (path_integrator.dart) */
#define DR_INTEGRATOR_DIRECT_ONE 2 /* DirectLightingIntegrator, strategy "one" (direct_lighting_integrator.dart:51-55,82-87): ONE light per
                                    * vertex, picked by its own 1-D sample slot (lightNumOffset), the estimate scaled by the light count */

#define DR_SAMPLER_HOST_BUFFER 0 /* caller supplies sample vectors (+ the in-Li RNG draws) */
#define DR_SAMPLER_COUNTER 1     /* on-device LD sampler, keyed per (pixel, block) / (pixel, sample) */

/* Everything SamplerRenderer.render needs besides the Scene
 * (lib/renderers/sampler_renderer.dart:29-31,36-65). */
typedef struct DrRenderDesc {
  DrCamera camera;
  DrFilm film;
  int32_t integrator;
  int32_t max_depth; /* PathIntegrator.maxDepth / DirectLightingIntegrator.maxDepth (default 5) */
  int32_t spp;       /* LowDiscrepancySampler.nPixelSamples, power of two */
  int32_t sampler_mode;
  int64_t seed; /* DR_SAMPLER_COUNTER */
  /* Work split.  task_*: the reference's GetSubWindow rectangle of the sampler
   * window (lib/core/common.dart:52-73, dartray.dart:1009-1023).  tile_*: 32x32
   * tiles (TilePixelSampler.tileSize, tile_pixel_sampler.dart:37) dealt
   * round-robin over ranks; tile_count <= 1 disables it. */
  int32_t task_num, task_count;
  int32_t tile_rank, tile_count, tile_size;
  /* DR_SAMPLER_HOST_BUFFER: nsamples camera samples, in reference order
   * (pixel-major, all spp of a pixel adjacent). */
  int64_t nsamples;
  const int32_t* pixel_xy;  /* [nsamples/spp][2] raster pixel of each group of spp samples */
  const float* sample_vec;  /* [nsamples][sample_stride]: imageU, imageV, lensU, lensV, time, oneD..., twoD... */
  int32_t sample_stride;
  const double* tail;       /* [nsamples][max_tail] RNG.randomFloat() values drawn inside Li, or NULL */
  int32_t max_tail;
  /* Optional packed form of `tail` (round 5): tail_offsets[nsamples + 1], non-decreasing; sample s owns
   * tail[tail_offsets[s] .. tail_offsets[s + 1]) -- the values it actually drew, in draw order (position p of the fixed
   * form = element p of the run; a read past the run yields 0.0 like the zero fill of the fixed form); max_tail stays the
   * bound on a run's length.  NULL: the fixed [nsamples][max_tail] form.  A path draws nothing before its fourth vertex
   * and most paths end early, so a recorded C2 stream holds ~5 of its 40 slots per sample: the replay moves 0.4 x the
   * bytes over PCIe. */
  const uint64_t* tail_offsets;
} DrRenderDesc;

/* Counters and timings accumulated over the dr_render* calls on a scene since
 * the last dr_reset_stats (the "Stats" probes of lib/core/stats.dart that the
 * hot loops call, e.g. bvh_accel.dart:106-163). */
typedef struct DrRenderStats {
  uint64_t camera_samples; /* samples traced (incl. the sampler's dead border, image_film.dart:247-252) */
  uint64_t film_samples;   /* film pixels in this rank's window x spp: the throughput numerator */
  uint64_t closest_rays, any_rays;
  uint64_t closest_nodes, any_nodes; /* iterations of the loops at bvh_accel.dart:122 / :185 */
  uint64_t closest_tris, any_tris;   /* primitive tests at bvh_accel.dart:131 / :193 */
  uint64_t trace_launches;           /* number of traversal-kernel launches (closest + any) */
  double trace_ms;                   /* summed device time of the traversal launches (HIP events) */
  double total_ms;                   /* device time of the render calls (HIP events) */
  uint64_t batches;
  uint64_t closest_launches, any_launches; /* per kernel: k_trace<0> (closest hit), k_trace<1> (any hit) */
  double closest_ms, any_ms; /* a stage's any-hit launch runs beside its closest-hit launch (second stream; DARTRAY_OVERLAP_ANY=0
                              * serialises them): any_ms counts its time AFTER the closest-hit launch ended */
  double shade_ms;  /* k_shade_path / k_shade_direct */
  double gen_ms;    /* k_gen_samples (+ host-buffer transpose) and k_raygen */
  double film_ms;   /* k_film */
  /* shading-stage work items, summed over the stages of every batch: entries of the stage's active list
   * (shade_items), of which path vertices that were set up (a surface hit within maxDepth: shade_vertices), and
   * the rays they queued (continuation, MIS and shadow rays) */
  uint64_t shade_items, shade_vertices, shade_cont, shade_mis, shade_shadow;
  /* device time of the traversal-kernel pilot the FIRST big render of a big scene runs (dr_render_device); it is
   * part of that call's total_ms */
  double pilot_ms;
} DrRenderStats;

/* Select the GPU.  Must precede everything else. */
int dr_init(int device);

/* Host-side BVHAccel constructor (bvh_accel.dart:41-91,228-437; SAH, 12
 * buckets): used by the standalone host; a Dart caller marshals its own
 * BVHAccel.nodes instead.  tri_idx is in *refined* primitive order.
 * nodes_out holds 2*ntris-1 nodes; order_out[i] = input triangle placed at
 * BVH primitive slot i. */
int dr_bvh_build(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris, int32_t max_prims_in_node,
                 DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out, uint32_t* depth_out);

/* Same with quadric primitives in the list: a row tri_idx[3*i] == DR_PRIM_QUADRIC takes its world bound
 * (Shape.worldBound, shape.dart:37-39) from quadric_bounds[6 * tri_idx[3*i+1]] = (pMin xyz, pMax xyz). */
int dr_bvh_build_mixed(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris,
                       const float* quadric_bounds, uint64_t nquadrics, int32_t max_prims_in_node,
                       DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out, uint32_t* depth_out);

/* The same constructor ON THE GPU (dr_bvh_device.hip): identical arguments (host
 * pointers) and byte-identical nodes_out / order_out -- the bounds, the centroid bounds and the 12 SAH buckets are
 * device-wide reductions, the reference's two-pointer `partition` (common.dart:256-287) is the swap of the k-th
 * misplaced item of the left part with the k-th from the end (ranks from one prefix sum), sub-trees of at most 64
 * items are finished by one thread each with the reference's recursion as it stands.  Needs dr_init.  The host
 * builder above stays as the fallback (no GPU) and as this one's checker. */
int dr_bvh_build_device(const float* verts, uint64_t nverts, const uint32_t* tri_idx, uint64_t ntris,
                        const float* quadric_bounds, uint64_t nquadrics, int32_t max_prims_in_node,
                        DrBvhNode* nodes_out, uint64_t* nnodes_out, uint32_t* order_out, uint32_t* depth_out);

/* Scene upload (replaces the construction of lib/core/scene.dart Scene). */
int dr_scene_create(const DrSceneDesc* desc, DrScene** out);
void dr_scene_destroy(DrScene* scene);

/* Which traversal kernel each ray kind ([0] closest hit, [1] any hit) uses on this scene: 0 = not decided yet (the
 * first big render of a big scene measures the candidates on calibration batches of its own samples, rendered into the
 * film like any other: DrRenderStats.pilot_ms), 2 = one node per step, 3 = sibling pairs, 5 (closest-hit rays only) =
 * sibling pairs with the ray's cold state in LDS, six workgroups per CU.  All are bit-exact; only speed depends on the choice.  A host that renders the same
 * scene again (another frame, another process) can store the measured choice and hand it back: the pilot is then
 * skipped.  Setting 0 makes the next big render measure again. */
int dr_scene_get_trace_kernels(const DrScene* scene, uint32_t kernels_out[2]);
int dr_scene_set_trace_kernels(DrScene* scene, const uint32_t kernels[2]);
/* What the pilot measured, in ms per algorithmic GB (32 B per node visit + 48 B per triangle test of the calibration
 * batch's own counters): out[0..2] closest-hit rays with kernel 2 / 3 / 5, out[3..4] any-hit rays with kernel 2 / 3,
 * out[5] reserved (0).  0 = not measured (no pilot has run, the kernels were set by the host, or that candidate was
 * skipped: kernel 5 is not timed where kernel 3 lost to kernel 2 by more than 5 %).  Diagnostics: how close a choice was. */
int dr_scene_get_pilot(const DrScene* scene, float ms_per_gb_out[6]);
/* Diagnostics: what the scene's LAST dr_render_device call actually ran with, switches (dr_set_option / environment)
 * included -- out[0] path-state layout (64 / 4), out[1] / out[2] the traversal kernel of its closest-hit / any-hit launches
 * (2, 3, 5 as above), out[3] reserved (-1), out[4] the
 * calibration batches it ran (0: no pilot), out[5] its batches, out[6] workgroups per CU of a persistent traversal
 * launch, out[7] bit 0: a stage's any-hit launch ran beside its closest-hit launch, bit 1: the camera rays went through the
 * wave-coherent kernel (k_trace_pk), bit 3: the device sampler generated bounce b's blocks only for the pixel
 * groups alive at bounce b (DARTRAY_LAZY_GEN).  All 0 / -1 before the first render. */
int dr_scene_last_render_info(const DrScene* scene, int32_t info_out[8]);
/* Diagnostics: the part of DrRenderStats' closest-hit totals (closest_rays / _nodes / _tris / _launches / _ms, accumulated since the last
 * dr_reset_stats) that the wave-coherent kernel k_trace_pk traced -- the camera rays -- as out[0] rays, out[1] node visits, out[2]
 * triangle tests, out[3] launches, out[4] ms.  Waits for the renders in flight like dr_get_stats.  What is left after subtracting them
 * belongs to the per-lane closest-hit kernel (k_trace<0> / k_trace3<0> / k_trace3c): how bench.py prices the two separately. */
int dr_scene_get_coherent_stats(DrScene* scene, double out[5]);
/* Diagnostics: the device sampler's work since the last dr_reset_stats, in (pixel, LD block) pairs -- one pair = one shuffled run of
 * spp indices.  out[0]: pairs generated; out[1]: pairs the integrator's reads name (every block a path of maximal length would read,
 * for every pixel).  Equal unless generation is lazy (DARTRAY_LAZY_GEN, the default for path renders at >= 64 spp) and paths end
 * early: bounce b's blocks are shuffled only for 64-pixel groups with a path alive at bounce b.  Both 0 for the host-buffer sampler
 * and the full-float sample form.  Waits for the renders in flight like dr_get_stats. */
int dr_scene_get_sampler_stats(DrScene* scene, double out[2]);
/* Device memory the scene's path-state workspace holds right now (tiles, queues, sample storage; it is allocated by the first render,
 * grows when a later render needs more and is freed with the scene): what a host budgets next to its own allocations. */
int dr_scene_workspace_bytes(const DrScene* scene, uint64_t* bytes_out);
/* The path-state layout of this scene's path-traced renders, next to the traversal kernels and measured the same way: the
 * first big render's first pilot batch counts how many of its slots are still alive at the second bounce (density_out;
 * -1 before); below one half the renders use four-slot, line-grouped sub-tiles (layout 4: stage lists that thin out touch
 * few lines per surviving slot), otherwise 64-slot runs (layout 64).  Results never depend on it.  A host can store the
 * choice with the scene and hand it back (0 = measure again at the next big render). */
int dr_scene_get_state_layout(const DrScene* scene, int32_t* layout_out, float* density_out);
/* Diagnostics: what dr_scene_create derived from the marshalled tree -- the sibling-pair records of the v3 traversal kernels
 * (64 bytes each: the two child nodes of an interior node side by side, a child's `offset` naming its own pair record; 0 records
 * when the tree cannot use them), how many of them form the breadth-first top of the tree, and the height it measured.  out may
 * be NULL to query the counts. */
int dr_scene_get_pairs(const DrScene* scene, void* out, uint64_t cap_bytes, uint64_t* npairs_out, uint32_t* top_pairs_out, uint32_t* depth_out);
int dr_scene_set_state_layout(DrScene* scene, int32_t layout);

/* Aggregate.intersect / Aggregate.intersectP (lib/core/primitive.dart:33-55 ->
 * bvh_accel.dart:101-226) on a batch of rays; host buffers. */
int dr_intersect(DrScene* scene, const DrRay* rays, int64_t n, DrHit* out, int32_t any_hit);

/* Floats per camera-sample vector for an integrator (Sample layout,
 * lib/core/sample.dart:23-79). */
int32_t dr_sample_floats(int32_t integrator, uint32_t nlights);
/* Same for an uploaded scene: DirectLighting requests roundSize(light.nSamples) entries per light slot
 * (direct_lighting_integrator.dart:70-87; low_discrepancy_sampler.dart:43-49), so the vector length depends on
 * the lights; dr_sample_floats assumes nsamples == 1 everywhere. */
int32_t dr_scene_sample_floats(const DrScene* scene, int32_t integrator);

/* Renderer.render(Scene) (lib/core/renderer.dart:28;
 * sampler_renderer.dart:36-65): traces this task's window and returns the
 * film.  film_out: [height*width*4] f32 (X, Y, Z, weightSum) -- ImageFilm's
 * _Lxyz/_weightSum (image_film.dart:69-71); rgb_out (optional):
 * [height*width*3] = OutputImage.rgb after ImageFilm.writeImage (:268-299). */
int dr_render(DrScene* scene, const DrRenderDesc* desc, float* film_out, float* rgb_out);

/* Same, with the film left in device memory (accumulated into film_dev, which
 * the caller zero-initialises) on the given hipStream_t; used by bench.py and
 * by the multi-GPU path, which reduces film_dev over RCCL before resolving.
 * Everything the call enqueues is ordered behind earlier work of hip_stream and
 * in front of later work of it; internally a stage's any-hit launch runs on a
 * stream of the scene's own, beside the closest-hit launch, tied to hip_stream
 * by events (DARTRAY_OVERLAP_ANY=0: everything on hip_stream).
 * The call returns with its kernels enqueued, except in three cases where it waits for hip_stream itself: the
 * first big render of a big scene (the traversal-kernel pilot reads its counters back between its three batches),
 * DR_SAMPLER_HOST_BUFFER (the host sample buffers of a batch are staged before the next batch reuses the area), and
 * DirectLighting over mirror / glass materials (one count is read back per round of the specular recursion). */
int dr_render_device(DrScene* scene, const DrRenderDesc* desc, void* film_dev, void* hip_stream);

/* The raster pixels a DR_SAMPLER_COUNTER render of `desc` traces, in trace
 * order (GetSubWindow rectangle, common.dart:52-73, intersected with this
 * rank's round-robin tiles).  Host-only; out_xy may be NULL to query the count. */
int dr_enumerate_pixels(const DrRenderDesc* desc, int32_t* out_xy, uint64_t cap, uint64_t* n_out);

/* ImageFilm.writeImage on a device film: XYZ -> RGB, divide by weightSum. */
int dr_film_resolve_device(const void* film_dev, int64_t npixels, void* rgb_dev, void* hip_stream);

/* Accumulated stats of this scene (synchronises with the last render's stream work). */
int dr_get_stats(DrScene* scene, DrRenderStats* out);
int dr_reset_stats(DrScene* scene);

/* Device float4 copy kernel: the measured HBM-bandwidth denominator. Returns GB/s. */
int dr_copy_bandwidth(uint64_t bytes, int32_t iters, double* gbps_out);

/* ---- multi-GPU: one process per GPU, the film merged over RCCL -------------------------------------------
 * The reference fans a render out over isolates, one sub-window each, and merges their rectangles in the
 * host (lib/dartray_web/render_manager.dart:100-141; GetSubWindow lib/core/common.dart:52-73).  Here every
 * rank renders its tile share (DrRenderDesc.tile_*) into a zero-initialised full-frame device film with
 * dr_render_device, and ONE ncclReduce(sum, f32) over xGMI merges the films on `root`.  The host only has to
 * carry DR_COMM_ID_BYTES from rank 0 to the other ranks (any channel: a file, a socket, MPI, torchrun's
 * store).  librccl is loaded on the first dr_comm_* call (DARTRAY_RCCL_LIB overrides the search), so
 * single-GPU hosts need not have it.  One communicator per process. */
#define DR_COMM_ID_BYTES 128 /* == NCCL_UNIQUE_ID_BYTES */
/* every rank, local: binds librccl and checks its version against the ABI subset this library declares -- DR_OK or the
 * reason, without talking to any other rank (so a host can agree on "every rank can" BEFORE anyone blocks in dr_comm_init) */
int dr_comm_available(void);
/* rank 0: ncclGetUniqueId into id_out[DR_COMM_ID_BYTES] */
int dr_comm_unique_id(void* id_out, uint64_t cap);
/* every rank, after dr_init: ncclCommInitRank(world, id, rank); blocks until all ranks have called it */
int dr_comm_init(int32_t rank, int32_t world, const void* unique_id, uint64_t id_bytes);
/* film_dev: [npixels][4] f32 (X, Y, Z, weightSum) on every rank; summed in place into rank `root`'s buffer on
 * the given hipStream_t (asynchronous like dr_render_device; other ranks' buffers are left as they were).
 * world == 1: no-op through the same code path (an in-place single-rank ncclReduce). */
int dr_film_reduce(void* film_dev, int64_t npixels, int32_t root, void* hip_stream);
/* in-place ncclAllReduce(max / sum) of n doubles on the stream: the barrier + max-over-ranks of a timed region */
int dr_comm_allreduce_f64(void* buf_dev, int64_t n, int32_t op_max, void* hip_stream);
/* One rank's part of a sharded render and the merge, in one call (the fan-out of RenderManager,
 * lib/dartray_web/render_manager.dart:100-141, with the rectangle copies replaced by ONE reduce): renders desc's share
 * (tile_rank / tile_count / tile_size, or task_num / task_count) into a zero-initialised full-frame device film,
 * sums the ranks' films on `root` through the communicator of dr_comm_init (skipped when none exists or its world
 * is 1), and on the root rank resolves the film and copies it out.  film_out / rgb_out ([h*w*4] / [h*w*3] f32) may be
 * NULL; on ranks other than root nothing is written.  A foreign host needs no device-memory calls of its own. */
int dr_render_sharded(DrScene* scene, const DrRenderDesc* desc, int32_t root, float* film_out, float* rgb_out);
int dr_comm_rank(void);  /* -1 before dr_comm_init */
int dr_comm_world(void); /* 0 before dr_comm_init */
int dr_comm_destroy(void);

const char* dr_last_error(void);
const char* dr_version(void);
/* The layout version of this header's structs and the meaning of its entry points.  A host compares dr_abi_version() of the library it
 * loaded with the DR_ABI_VERSION it was built against BEFORE it passes a struct: the structs carry no size field, so a host built against
 * an older header would hand the library a shorter object than it reads (version 5 -> 6: DrRenderDesc grew by tail_offsets, 1344 -> 1352
 * bytes; version 6 -> 7: DR_INTEGRATOR_DIRECT_ONE, dr_scene_workspace_bytes, the switch list of dr_set_option). */
#define DR_ABI_VERSION 7
int32_t dr_abi_version(void);

/* Tuning / diagnostic switches.  Every switch is also an environment variable of the same name (DARTRAY_<NAME>); a
 * value set here takes precedence, is read at every use (nothing is latched at first use: the next render sees it) and
 * needs no setenv in a long-lived foreign host.  name: with or without the DARTRAY_ prefix, any case; value NULL: back
 * to the environment's value; "": unset for this process.  Unknown names are DR_ERR_INVALID.  The film never depends on a
 * switch, with one stated exception: the switches pick between bit-exact variants of a kernel, a layout or a schedule, or
 * print diagnostics.  The exception is the sampler: LAZY_GEN and the GEN_* switches choose when and how the device LD
 * sampler draws the same keyed streams -- bit-exact by test (tests/test_gpu_render.py, test_gpu_options.py), but they are
 * variants of the sampler, not of a schedule.  The fifteen switches (round 6; the A/B switches of measured negatives left the library with their
 * code: experiments/r06_*.diff):
 *   TRACE_IMPL 2|3|5      traversal kernels for both ray kinds: 2 = k_trace<0/1>, 3 = the sibling-pair kernels k_trace3<0> / k_trace3a,
 *                         5 = 3 with the closest-hit rays' cold state in LDS (k_trace3c); default: the scene's measured choice
 *   TRACE_WG_PER_CU n     workgroups of a persistent traversal launch per CU (occupancy sweeps)
 *   STATE_LAYOUT 64|4     path-state layout (default: picked per scene from the pilot's stage-list densities)
 *   BATCH_BITS b          at most 2^b camera samples per batch (16..28; default 27 for a scene's first big render, 28 afterwards)
 *   OVERLAP_ANY 0         a stage's any-hit launch after its closest-hit launch instead of beside it
 *   PILOT 0|force         no calibration batches / calibration batches also on renders too small to need them (tests)
 *   COHERENT_CAMERA 0     the camera rays through the per-lane traversal kernels like every other ray (default: the wave-coherent k_trace_pk)
 *   LAZY_GEN 0            the device sampler shuffles every LD block for every pixel up front (default: bounce b's blocks only for the
 *                         64-pixel groups that still have a path alive at bounce b)
 *   GEN_ALL_BLOCKS 1      ... and also the blocks no kernel reads;  GEN_SLOW_DRAWS 1   every generator step through Random.nextInt's retry loop
 *   SCENE_PREP host       dr_scene_create's tree checks and pair records by the serial host loops (the reference the device code is tested against)
 *   BUILD_THREADS n       threads of the host BVH builder;  RCCL_LIB path   the librccl to bind
 *   STAGE_COUNTS 1|2      per stage: list lengths and kernel times (2: also the node visits / triangle tests of each stage's traversals,
 *                         waiting for the device after every stage);  VERBOSE 1|2   what the library decided (2: + the BVH builders' timings) */
int dr_set_option(const char* name, const char* value);


/* ---- layout checks: a foreign host (dart:ffi Struct classes, ctypes, a C program) must see exactly these
 * sizes and offsets (LP64, little endian, natural alignment) ---- */
#ifdef __cplusplus
#define DR_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define DR_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif
#define DR_ABI_SIZE(T, n) DR_ABI_ASSERT(sizeof(T) == (n), "sizeof(" #T ") != " #n)
#define DR_ABI_OFFSET(T, f, n) DR_ABI_ASSERT(offsetof(T, f) == (n), "offsetof(" #T ", " #f ") != " #n)
DR_ABI_SIZE(DrBvhNode, 32);
DR_ABI_OFFSET(DrBvhNode, offset, 24);
DR_ABI_OFFSET(DrBvhNode, nprims, 28);
DR_ABI_OFFSET(DrBvhNode, axis, 30);
DR_ABI_SIZE(DrMaterial, 56);
DR_ABI_OFFSET(DrMaterial, kd, 4);
DR_ABI_OFFSET(DrMaterial, sigma, 40);
DR_ABI_OFFSET(DrMaterial, index, 48);
DR_ABI_SIZE(DrAreaLight, 128);
DR_ABI_OFFSET(DrAreaLight, first_tri, 16);
DR_ABI_OFFSET(DrAreaLight, position, 32);
DR_ABI_OFFSET(DrAreaLight, world_to_light, 48);
DR_ABI_OFFSET(DrAreaLight, cone_width, 112);
DR_ABI_SIZE(DrEnvMap, 144);
DR_ABI_OFFSET(DrEnvMap, width, 8);
DR_ABI_OFFSET(DrEnvMap, light_to_world, 16);
DR_ABI_SIZE(DrLightTri, 16);
DR_ABI_SIZE(DrMeshXform, 128);
DR_ABI_SIZE(DrQuadric, 168);
DR_ABI_OFFSET(DrQuadric, object_to_world, 8);
DR_ABI_OFFSET(DrQuadric, params, 136);
DR_ABI_SIZE(DrSceneDesc, 208);
DR_ABI_OFFSET(DrSceneDesc, nnodes, 8);
DR_ABI_OFFSET(DrSceneDesc, verts, 16);
DR_ABI_OFFSET(DrSceneDesc, tri_idx, 32);
DR_ABI_OFFSET(DrSceneDesc, tri_material, 48);
DR_ABI_OFFSET(DrSceneDesc, materials, 72);
DR_ABI_OFFSET(DrSceneDesc, nmaterials, 80);
DR_ABI_OFFSET(DrSceneDesc, lights, 88);
DR_ABI_OFFSET(DrSceneDesc, nlights, 96);
DR_ABI_OFFSET(DrSceneDesc, light_tris, 104);
DR_ABI_OFFSET(DrSceneDesc, nlight_tris, 112);
DR_ABI_OFFSET(DrSceneDesc, bvh_depth, 116);
DR_ABI_OFFSET(DrSceneDesc, env_maps, 120);
DR_ABI_OFFSET(DrSceneDesc, quadrics, 136);
DR_ABI_OFFSET(DrSceneDesc, vert_normals, 152);
DR_ABI_OFFSET(DrSceneDesc, mesh_xforms, 192);
DR_ABI_OFFSET(DrSceneDesc, nmesh_xforms, 200);
DR_ABI_SIZE(DrRay, 40);
DR_ABI_OFFSET(DrRay, tmin, 24);
DR_ABI_SIZE(DrHit, 32);
DR_ABI_OFFSET(DrHit, t, 8);
DR_ABI_SIZE(DrCamera, 168);
DR_ABI_OFFSET(DrCamera, lens_radius, 128);
DR_ABI_OFFSET(DrCamera, focal_distance, 136);
DR_ABI_OFFSET(DrCamera, type, 160);
DR_ABI_SIZE(DrFilm, 1080);
DR_ABI_OFFSET(DrFilm, crop, 8);
DR_ABI_OFFSET(DrFilm, filter_xw, 40);
DR_ABI_OFFSET(DrFilm, filter_table, 56);
DR_ABI_SIZE(DrRenderDesc, 1352);
DR_ABI_OFFSET(DrRenderDesc, film, 168);
DR_ABI_OFFSET(DrRenderDesc, integrator, 1248);
DR_ABI_OFFSET(DrRenderDesc, seed, 1264);
DR_ABI_OFFSET(DrRenderDesc, task_num, 1272);
DR_ABI_OFFSET(DrRenderDesc, tile_rank, 1280);
DR_ABI_OFFSET(DrRenderDesc, nsamples, 1296);
DR_ABI_OFFSET(DrRenderDesc, pixel_xy, 1304);
DR_ABI_OFFSET(DrRenderDesc, sample_vec, 1312);
DR_ABI_OFFSET(DrRenderDesc, sample_stride, 1320);
DR_ABI_OFFSET(DrRenderDesc, tail, 1328);
DR_ABI_OFFSET(DrRenderDesc, max_tail, 1336);
DR_ABI_OFFSET(DrRenderDesc, tail_offsets, 1344);
DR_ABI_SIZE(DrRenderStats, 200);
DR_ABI_OFFSET(DrRenderStats, trace_ms, 72);
DR_ABI_OFFSET(DrRenderStats, film_ms, 144);
DR_ABI_OFFSET(DrRenderStats, shade_items, 152);
DR_ABI_OFFSET(DrRenderStats, pilot_ms, 192);

#ifdef __cplusplus
}
#endif
#endif /* DARTRAY_HIP_H */
