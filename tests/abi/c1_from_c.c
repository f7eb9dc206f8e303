/* c1_from_c.c -- BASELINE config C1 rendered by a plain C host through include/dartray_hip.h ALONE (no Python, no
 * C++, no HIP headers): the proof that the header is sufficient for a foreign host such as the Dart shim of
 * integration/hip_sampler_renderer.dart.
 *
 * The scene is built by hand exactly as DartRay's API would flatten it (Cornell floor quad + quad emitter:
 * web/scenes semantics, dartray_amd/scenes.py:cornell_c1_prims): two TriangleMesh shapes refined to triangles in
 * LIFO order (primitive.dart:71-84), BVHAccel (dr_bvh_build stands in for the Dart constructor), one
 * DiffuseAreaLight whose ShapeSet holds the emitter's triangles in refine order (shape_set.dart:25-35).
 * Camera, film and the recorded serial sample stream (what the Dart objects hold / the Dart sampler produces) come
 * from a blob written by the test driver from tests/golden/c1_serial.npz:
 *
 *   int32 npix, spp, stride, width, height | DrCamera | DrFilm | int32 pixel_xy[npix][2] | float sample_vec[npix*spp][stride]
 *
 * Output blob: float film[height*width*4] | float rgb[height*width*3].
 *
 *   gcc -std=c11 -Wall -I include tests/abi/c1_from_c.c -o c1_from_c dartray_amd/libdartray_hip.so
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dartray_hip.h"

#define CHECK(call)                                                         \
  do {                                                                      \
    int rc_ = (call);                                                       \
    if (rc_ != DR_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dr_last_error());      \
      return 2;                                                             \
    }                                                                       \
  } while (0)

static int read_exact(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s input.blob output.blob\n", argv[0]);
    return 1;
  }
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 1;
  int32_t hdr[5];
  DrRenderDesc rd;
  memset(&rd, 0, sizeof(rd));
  if (read_exact(in, hdr, sizeof(hdr)) || read_exact(in, &rd.camera, sizeof(DrCamera)) || read_exact(in, &rd.film, sizeof(DrFilm))) return 1;
  const int32_t npix = hdr[0], spp = hdr[1], stride = hdr[2], width = hdr[3], height = hdr[4];
  int32_t* pixel_xy = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)npix);
  float* sample_vec = (float*)malloc(sizeof(float) * (size_t)npix * spp * stride);
  if (read_exact(in, pixel_xy, sizeof(int32_t) * 2 * (size_t)npix) || read_exact(in, sample_vec, sizeof(float) * (size_t)npix * spp * stride)) return 1;
  fclose(in);

  /* ---- the flattened Scene ---- */
  static const float verts[8][3] = {
      {-10, -10, -10}, {10, -10, -10}, {10, -10, 10}, {-10, -10, 10},            /* floor quad */
      {-3, 9.9f, -3},  {3, 9.9f, -3},  {3, 9.9f, 3},  {-3, 9.9f, 3}};            /* emitter quad, facing down */
  /* refined primitive order: each mesh's triangles come out of the LIFO todo list reversed */
  static const uint32_t refined[4][3] = {{0, 2, 3}, {0, 1, 2}, {4, 6, 7}, {4, 5, 6}};
  static const uint32_t mat_of[4] = {0, 0, 1, 1};
  static const int32_t light_of[4] = {-1, -1, 0, 0};
  DrBvhNode nodes[7];
  uint64_t nnodes = 0;
  uint32_t order[4], depth = 0;
  if (dr_abi_version() != DR_ABI_VERSION) {  /* a host checks the layout version before it passes a struct */
    fprintf(stderr, "libdartray_hip.so has ABI version %d, this host was built against %d\n", (int)dr_abi_version(), DR_ABI_VERSION);
    return 2;
  }
  CHECK(dr_init(0));
  CHECK(dr_bvh_build(&verts[0][0], 8, &refined[0][0], 4, 4, nodes, &nnodes, order, &depth));
  uint32_t tri_idx[4][3], tri_material[4];
  int32_t tri_light[4];
  uint8_t tri_reverse[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { /* BVHAccel.primitives = orderedPrims (bvh_accel.dart:69-76) */
    memcpy(tri_idx[i], refined[order[i]], sizeof(tri_idx[i]));
    tri_material[i] = mat_of[order[i]];
    tri_light[i] = light_of[order[i]];
  }
  DrMaterial mats[2];
  memset(mats, 0, sizeof(mats));
  mats[0].type = DR_MATERIAL_MATTE;
  mats[0].kd[0] = mats[0].kd[1] = mats[0].kd[2] = 0.75f;
  mats[1].type = DR_MATERIAL_MATTE;
  mats[1].kd[0] = mats[1].kd[1] = mats[1].kd[2] = 0.5f;
  DrAreaLight light;
  memset(&light, 0, sizeof(light));
  light.kind = DR_LIGHT_DIFFUSE_AREA;
  light.L[0] = light.L[1] = light.L[2] = 36.0f;
  light.nsamples = 1;
  light.first_tri = 0;
  light.ntris = 2;
  DrLightTri ltris[2] = {{{4, 6, 7}, 0}, {{4, 5, 6}, 0}};
  DrSceneDesc sd;
  memset(&sd, 0, sizeof(sd));
  sd.nodes = nodes;
  sd.nnodes = nnodes;
  sd.verts = &verts[0][0];
  sd.nverts = 8;
  sd.tri_idx = &tri_idx[0][0];
  sd.ntris = 4;
  sd.tri_material = tri_material;
  sd.tri_light = tri_light;
  sd.tri_reverse = tri_reverse;
  sd.materials = mats;
  sd.nmaterials = 2;
  sd.lights = &light;
  sd.nlights = 1;
  sd.light_tris = ltris;
  sd.nlight_tris = 2;
  sd.bvh_depth = 0; /* unknown: the library measures it (a Dart BVHAccel does not record its depth) */
  DrScene* scene = NULL;
  CHECK(dr_scene_create(&sd, &scene));

  /* ---- SamplerRenderer.render: DirectLighting, the recorded serial sample stream ---- */
  rd.integrator = DR_INTEGRATOR_DIRECT_ALL;
  rd.max_depth = 5;
  rd.spp = spp;
  rd.sampler_mode = DR_SAMPLER_HOST_BUFFER;
  rd.task_num = 0;
  rd.task_count = 1;
  rd.tile_count = 1;
  rd.nsamples = (int64_t)npix * spp;
  rd.pixel_xy = pixel_xy;
  rd.sample_vec = sample_vec;
  rd.sample_stride = stride;
  if (stride < dr_scene_sample_floats(scene, rd.integrator)) {
    fprintf(stderr, "sample vectors too short: %d < %d\n", stride, dr_scene_sample_floats(scene, rd.integrator));
    return 3;
  }
  float* film = (float*)calloc((size_t)width * height * 4, sizeof(float));
  float* rgb = (float*)calloc((size_t)width * height * 3, sizeof(float));
  CHECK(dr_render(scene, &rd, film, rgb));
  DrRenderStats st;
  CHECK(dr_get_stats(scene, &st));
  printf("c1_from_c: %llu camera samples, %llu closest rays, %llu shadow rays, depth %u, %s\n",
         (unsigned long long)st.camera_samples, (unsigned long long)st.closest_rays, (unsigned long long)st.any_rays, depth, dr_version());
  dr_scene_destroy(scene);
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 1;
  fwrite(film, sizeof(float), (size_t)width * height * 4, out);
  fwrite(rgb, sizeof(float), (size_t)width * height * 3, out);
  fclose(out);
  return 0;
}
