/* mt_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's per-pixel ray-trace path
 * (/root/reference/VerStarting, cited per function in mt_oracle.c).  Only
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
 * load it, and only as the checker.  The product path (mythtracer_amd/,
 * include/) never links, imports or executes anything from oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * bit-for-bit against golden vectors produced by the reference itself
 * (oracle/_ref, built from the reference's own sources by oracle/Makefile;
 * generator: tests/golden/make_golden.py).  One function is NOT pinned:
 * orc_tex_color_at (Texture::GetColorAt, texture.cc:11-58), because
 * texture.cc needs SDL2 and cannot be built here — "parity unpinned" for that
 * function only.
 */
#ifndef MT_ORACLE_H_
#define MT_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* counters filled by orc_render_chunk / orc_intersect_rays */
enum {
  ORC_CNT_RAYS_PRIMARY = 0, /* OctTree::IntersectRay calls from level-0 TraceRayWorker */
  ORC_CNT_RAYS_SECONDARY,   /* ... from level>0 TraceRayWorker (reflection/refraction) */
  ORC_CNT_RAYS_SHADOW,      /* ... from the shadow loop */
  ORC_CNT_BOX_TESTS,        /* Node::NodeIntersectRay calls (root included) */
  ORC_CNT_NODE_VISITS,      /* Node::PrimitiveIntersectRay calls */
  ORC_CNT_TRI_TESTS,        /* Triangle::IntersectRay calls */
  ORC_CNT_MT_TESTS,         /* ... that passed the AABB pre-filter */
  ORC_CNT_SHADED_HITS,      /* TraceRayWorker calls that found a primitive */
  ORC_CNT_COUNT
};

orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);

/* ObjFileReader::ReadObjFile.  Returns 1 on success, 0 on failure (bool). */
int orc_load_obj(orc_scene *s, const char *path);

/* Programmatic construction (what octtree_test.cc does by hand). */
int orc_add_material(orc_scene *s, const char *name, const double ka[3],
                     const double kd[3], const double ks[3], double ns,
                     double refl, double tr, const double tf[3], double ni);
int orc_add_texture(orc_scene *s, const char *name, int w, int h,
                    const double *rgb /* w*h*3 */);
int orc_material_set_texture(orc_scene *s, int mtl, int tex);
/* mtl = -1 for "no material".  Returns the triangle index. */
int orc_add_triangle(orc_scene *s, const double v[9], const double n[9],
                     const double uvw[9], int mtl, int line_no);

/* OctTree::Finalize (idempotent here). Returns 1 ok, 0 on depth overflow. */
int orc_finalize(orc_scene *s);
/* scene.lights = n x {position, ambient, diffuse, specular} (12 doubles). */
void orc_set_lights(orc_scene *s, const double *lights, int n);

int orc_num_triangles(const orc_scene *s);
int orc_num_materials(const orc_scene *s);
void orc_root_aabb(const orc_scene *s, double out[6]);
/* per triangle: vertex(9) normal(9) uvw(9) aabb(6) doubles, mtl, line_no */
void orc_get_triangle(const orc_scene *s, int i, double out33[33], int *mtl,
                      int *line_no);
/* ambient(3) diffuse(3) specular(3) ns refl tr tf(3) ni = 16 doubles + tex */
void orc_get_material(const orc_scene *s, int i, double out16[16], int *tex,
                      char name[128]);

/* Octree dump in BFS order (children of a node are 8 consecutive entries).
 * orc_tree_info: node count, depth (root = 1).  orc_tree_dump fills
 * aabb[n*6], center[n*3], first_child[n] (0 = none), prim_begin[n],
 * prim_count[n] and prim_ids[num_triangles] (triangle indices, node after
 * node, in stored order). */
void orc_tree_info(const orc_scene *s, int *n_nodes, int *depth);
void orc_tree_dump(const orc_scene *s, double *aabb, double *center,
                   int32_t *first_child, int32_t *prim_begin,
                   int32_t *prim_count, int32_t *prim_ids);

/* Camera::GetSensor + Sensor::Reset.  cam = origin(3) pitch yaw roll aov.
 * out = origin(3) start_point(3) delta_scanline(3) delta_pixel(3). */
void orc_sensor(const double cam[7], int width, int height, double out[12]);
/* Sensor::GetRay direction for pixel (x, y). */
void orc_sensor_ray(const double sensor[12], int x, int y, double dir[3]);

/* MythTracer::RayTrace(WorkChunk*).  max_level = MAX_RECURSION_LEVEL (5 in
 * the reference).  dbg_line / dbg_point / counters may be NULL.  nthreads<=0
 * means omp_get_max_threads().  Returns wall seconds of the pixel loop via
 * *seconds if non-NULL. */
int orc_render_chunk(orc_scene *s, const double cam[7], int image_w,
                     int image_h, int chunk_x, int chunk_y, int chunk_w,
                     int chunk_h, int max_level, uint8_t *rgb,
                     int32_t *dbg_line, double *dbg_point,
                     uint64_t counters[ORC_CNT_COUNT], int nthreads,
                     double *seconds);

/* OctTree::IntersectRay on n rays (6 doubles each).  tri[i] = -1 on miss. */
void orc_intersect_rays(orc_scene *s, int n, const double *rays, int32_t *tri,
                        int32_t *line, double *t, double *point,
                        double *normal, double *uvw,
                        uint64_t counters[ORC_CNT_COUNT]);

/* Texture::GetColorAt (UNPINNED, see header comment). */
void orc_tex_color_at(const orc_scene *s, int tex, double u, double v,
                      double out[3]);
/* MythTracer::V3DtoRGB. */
void orc_v3d_to_rgb(const double v[3], uint8_t rgb[3]);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MT_ORACLE_H_ */
