/* mythtracer_hip.h — C ABI of libmythtracer_hip.so (MI355X / gfx950).
 *
 * The reference (gynvael/MythTracer, /root/reference/VerStarting) has no FFI or
 * plugin layer: its seam is the C++ class raytracer::MythTracer
 * (mythtracer.h:55-66).  This C ABI sits directly UNDER that seam.  Each entry
 * point below names the reference code it replaces; the host-side C++ facade
 * (mythtracer_amd/host, same class/field names as the reference) and any other
 * language binding call these and nothing else.  See INTEGRATION.md.
 *
 * Conventions: plain C, POD only, no exceptions.  Functions returning int
 * return MT_OK (0) or a negative MT_ERR_*; mt_last_error() gives the text for
 * the calling thread.  All pointers in mt_scene_desc / mt_render_* arguments
 * are HOST pointers unless the name starts with d_ (device pointer on the
 * scene's GPU).  One render may be in flight per mt_scene at a time.  The
 * library never falls back to a CPU path: without a usable HIP device every
 * call fails with MT_ERR_HIP.
 */
#ifndef MYTHTRACER_HIP_H_
#define MYTHTRACER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MT_ABI_VERSION 4

enum {
  MT_OK = 0,
  MT_ERR_ARG = -1,         /* bad argument / inconsistent description */
  MT_ERR_HIP = -2,         /* HIP runtime error (text in mt_last_error) */
  MT_ERR_UNSUPPORTED = -3, /* e.g. octree deeper than MT_MAX_TREE_DEPTH */
  MT_ERR_NOMEM = -4,
  MT_ERR_INTERNAL = -5     /* a device-side loop bound tripped: kernel logic error */
};

/* Deepest octree (root = depth 1) the traversal stack can hold.  The
 * reference has no limit (octtree.cc:52-135 recurses while a node holds >= 16
 * primitives); we refuse deeper trees instead of overflowing. */
#define MT_MAX_TREE_DEPTH 64
/* Deepest reflection/refraction recursion (reference: compile-time
 * MAX_RECURSION_LEVEL = 5, mythtracer.h:11). */
#define MT_MAX_RECURSION 16

/* raytracer::Material (material.h:12-48), tex = index into textures or -1. */
typedef struct mt_material {
  double ambient[3], diffuse[3], specular[3];
  double specular_exp, reflectance, transparency;
  double transmission_filter[3];
  double refraction_index;
  int32_t tex;
  int32_t reserved;
} mt_material;

/* raytracer::Light (light.h:8-14). */
typedef struct mt_light {
  double position[3], ambient[3], diffuse[3], specular[3];
} mt_light;

enum { MT_TEX_RGB8 = 0, MT_TEX_F64 = 1 };

/* raytracer::Texture (texture.h:13-23).  RGB8: 3 bytes per texel, colour =
 * byte / 255.0 (exactly what texture.cc:100-104 stores); F64: 3 doubles. */
typedef struct mt_texture {
  int32_t width, height;
  int32_t format;
  int32_t reserved;
  const void *texels;
} mt_texture;

/* Camera::Sensor after Sensor::Reset (camera.cc:27-63) plus the camera
 * origin: what Sensor::GetRay (camera.cc:65-69) needs.  Computed on the host
 * (sin/cos stay glibc's). */
typedef struct mt_sensor {
  double origin[3];
  double start_point[3];
  double delta_scanline[3];
  double delta_pixel[3];
} mt_sensor;

/* PerPixelDebugInfo (mythtracer.h:13-16). */
typedef struct mt_debug_px {
  int32_t line_no; /* -1 = no hit */
  int32_t reserved;
  double point[3];
} mt_debug_px;

/* Work counters of one call (sums over all pixels / rays). */
typedef struct mt_stats {
  uint64_t rays_primary;   /* OctTree::IntersectRay from level-0 TraceRayWorker */
  uint64_t rays_secondary; /* ... from level>0 (reflection / refraction) */
  uint64_t rays_shadow;    /* ... from the shadow loop (mythtracer.cc:94-156) */
  uint64_t box_tests;      /* Node::NodeIntersectRay evaluations (root included) */
  uint64_t node_visits;    /* Node::PrimitiveIntersectRay evaluations */
  uint64_t tri_tests;      /* Triangle::IntersectRay evaluations */
  uint64_t mt_tests;       /* ... that passed the AABB pre-filter */
  uint64_t shaded_hits;    /* TraceRayWorker calls that hit a primitive */
  uint64_t wave_node_steps;/* wave-level node scans executed (GPU only) */
  uint64_t wave_tri_steps; /* wave-level triangle slab evaluations (GPU only) */
  double kernel_ms;        /* device time of the kernel(s), HIP events */
  double total_ms;         /* wall time of the call incl. copies */
  /* Bytes the kernels REQUESTED (counted at the load/store instructions of the
   * path, whatever cache served them): by wave-uniform scalar loads -- one box
   * or node record fetched once for all rays of a wave -- and by per-lane
   * vector accesses (boxes, vertices, node records, shading inputs, per-ray
   * state), summed over lanes.  bench.py's roofline numerator. */
  uint64_t bytes_scalar;
  uint64_t bytes_vector;
} mt_stats;

/* Flattened scene: what Scene{tree, materials, textures} (scene.h:9-15) holds
 * after OctTree::Finalize (octtree.cc:16-24).
 *
 * Nodes are in breadth-first order, root = node 0; the 8 children of a split
 * node are the consecutive nodes first_child .. first_child+7 in the
 * reference's child order (octtree.cc:61-100: bit0 = x high, bit1 = z high,
 * bit2 = y high); first_child = 0 means "no children".
 *
 * Triangles are in NODE-STREAM order: node after node (same order as the node
 * array), inside a node in the order of Node::primitives.  prim_begin/count
 * index into that stream.  tri_id gives the AddPrimitive order index. */
typedef struct mt_scene_desc {
  uint32_t struct_size; /* sizeof(mt_scene_desc) */
  uint32_t abi_version; /* MT_ABI_VERSION */
  int32_t device;       /* HIP device ordinal */
  int32_t n_nodes, n_tris, n_materials, n_textures;
  int32_t tree_depth;   /* root = 1 */
  const double *node_aabb;        /* n_nodes x 6: min xyz, max xyz */
  const double *node_center;      /* n_nodes x 3 (Node::CalcCenter) */
  const int32_t *node_first_child;
  const int32_t *node_prim_begin;
  const int32_t *node_prim_count;
  const double *tri_vertex;       /* n_tris x 9 */
  const double *tri_normal;       /* n_tris x 9 */
  const double *tri_uvw;          /* n_tris x 9 */
  const double *tri_aabb;         /* n_tris x 6 (Triangle::cached_aabb) */
  const int32_t *tri_material;    /* -1 = mtl == nullptr */
  const int32_t *tri_line_no;     /* Primitive::debug_line_no */
  const int32_t *tri_id;
  const mt_material *materials;
  const mt_texture *textures;
} mt_scene_desc;

typedef struct mt_scene mt_scene;

const char *mt_last_error(void);
int mt_abi_version(void);
/* Number of visible HIP devices (<0 on error). */
int mt_device_count(void);

/* Uploads the flattened scene to HBM.  Replaces: the in-memory Scene the
 * reference keeps after LoadObj + Finalize (mythtracer.cc:247-256,281-285). */
mt_scene *mt_scene_create(const mt_scene_desc *desc);
void mt_scene_destroy(mt_scene *scene);

/* scene.lights (scene.h:14) — read at render time (mythtracer.cc:78), mutated
 * by the caller between frames (main_local.cc:79-110). */
int mt_scene_set_lights(mt_scene *scene, const mt_light *lights, int n);

/* MythTracer::RayTrace(WorkChunk*) (mythtracer.cc:280-312): renders the chunk
 * [chunk_x, chunk_x+chunk_w) x [chunk_y, chunk_y+chunk_h) of an image_w x
 * image_h image.  out_rgb: chunk_w*chunk_h*3 bytes, chunk-local row-major
 * RGB8 (WorkChunk::output_bitmap).  out_debug (nullable): chunk_w*chunk_h
 * entries (WorkChunk::output_debug).  max_depth = MAX_RECURSION_LEVEL (5). */
int mt_render_chunk(mt_scene *scene, const mt_sensor *sensor, int image_w,
                    int image_h, int chunk_x, int chunk_y, int chunk_w,
                    int chunk_h, int max_depth, uint8_t *out_rgb,
                    mt_debug_px *out_debug, mt_stats *stats);

/* Same, asynchronous, output left in HBM: d_rgb / d_debug are device pointers
 * on the scene's GPU, stream is a hipStream_t (NULL = default stream).  Work
 * counters accumulate in the scene and are fetched with mt_scene_read_stats
 * after the stream has been synchronised. */
int mt_render_chunk_device(mt_scene *scene, const mt_sensor *sensor,
                           int image_w, int image_h, int chunk_x, int chunk_y,
                           int chunk_w, int chunk_h, int max_depth,
                           void *d_rgb, void *d_debug, void *stream);

/* The master's work split (main_net_master.cc:195-221 GenerateWork, 128x128
 * WorkChunks) for a multi-GPU frame: the image is cut into tile_w x tile_h
 * tiles in row-major order; this call renders tiles first_tile,
 * first_tile+tile_stride, ... (n_tiles of them) in ONE launch.  Tile number j
 * of the call is written to d_rgb + j*tile_w*tile_h*3 as a chunk-local
 * row-major bitmap of its actual (edge-clipped) width x height, i.e. the PXLS
 * payload of that WorkChunk.  mt_blit_tiles_device is BlitWorkChunk
 * (main_net_master.cc:223-236) for such a buffer. */
int mt_render_tiles_device(mt_scene *scene, const mt_sensor *sensor,
                           int image_w, int image_h, int tile_w, int tile_h,
                           int first_tile, int tile_stride, int n_tiles,
                           int max_depth, void *d_rgb, void *stream);
int mt_blit_tiles_device(mt_scene *scene, int image_w, int image_h,
                         int tile_w, int tile_h, int first_tile,
                         int tile_stride, int n_tiles, const void *d_tiles,
                         void *d_image, void *stream);

/* Cost-balanced ownership of a multi-GPU frame's tiles.  The reference's master hands its WorkChunks out dynamically:
 * a worker asks for the next chunk when it is done with one (main_net_master.cc:62-80 the queue, :82-169
 * WorkerHandler), so no worker idles while chunks wait.  Ranks that render at the same time cannot pull from one
 * queue without a collective per tile; instead every rank computes THE SAME balanced assignment by itself from the
 * frame-wide cost map all ranks hold after their exchange (mt_scene_export_costs_device, all-reduce MAX): the tiles
 * sorted by the summed cost of their blocks, most expensive first (ties: lower tile number first), are dealt out in
 * rounds of alternating direction -- 0 1 .. N-1, N-1 .. 1 0, ... -- so that every rank holds one tile of every round:
 * the tile counts stay what the modular assignment gave (buffer sizes and the gather do not change) and the cost sums
 * differ by less than one tile of a round.
 *   mt_order_tiles_device: d_order[p] (int32, tiles_x * tiles_y entries) = tile at position p of that order, computed
 *     from d_cost_map (uint32 [map_h][map_w], one word per 8x8 block of the image) on the scene's GPU.  An all-zero map
 *     orders the tiles by number.
 *   mt_dealt_tile_count: how many tiles rank `rank` of `world` holds (host arithmetic, no device access).
 *   mt_deal_tiles_device: d_list[q] (int32) = the rank's tile of round q (d_order == NULL: the order by tile number);
 *     returns the count.
 *   mt_render_tile_list_device / mt_blit_tile_list_device: as the modular forms above with tile d_list[j] in slot j.
 *     list_id: launches with the same non-zero list_id promise the same list -- the scene then keeps its per-block
 *     cost history, running means included, as for any repeated launch; a launch with another list_id (or 0) takes its
 *     forecast from the cost map imported since the previous launch (mt_scene_import_costs_device; all ranks' costs by
 *     image position) and is a first frame without one.  The list is copied: the caller may reuse d_list at once.
 * Nothing computed for a pixel depends on the assignment: the gathered frame is byte-identical. */
int mt_order_tiles_device(mt_scene *scene, const void *d_cost_map, int map_w, int map_h,
                          int image_w, int image_h, int tile_w, int tile_h, void *d_order,
                          void *stream);
int mt_dealt_tile_count(int image_w, int image_h, int tile_w, int tile_h, int world, int rank);
int mt_deal_tiles_device(mt_scene *scene, const void *d_order, int image_w, int image_h,
                         int tile_w, int tile_h, int world, int rank, void *d_list, void *stream);
int mt_render_tile_list_device(mt_scene *scene, const mt_sensor *sensor, int image_w,
                               int image_h, int tile_w, int tile_h, const void *d_list,
                               int n_tiles, uint64_t list_id, int max_depth, void *d_rgb,
                               void *stream);
int mt_blit_tile_list_device(mt_scene *scene, int image_w, int image_h, int tile_w, int tile_h,
                             const void *d_list, int n_tiles, const void *d_tiles, void *d_image,
                             void *stream);

/* One frame on SEVERAL GPUs of this process -- the master/worker farm of the
 * reference (main_net_master.cc:195-236: GenerateWork cuts the frame into
 * WorkChunks, every worker renders chunks with the full-image sensor from its
 * own copy of the scene, main_net_worker.cc:29-32,148-150, BlitWorkChunk puts
 * them into the frame) inside one host process: scenes[r] is a replica of the
 * scene on its own HIP device (mt_scene_desc.device; several replicas may share
 * a device), the tiles of the tile_w x tile_h grid are dealt out by cost as
 * described above (first frame of a geometry: by tile number; a camera at rest
 * keeps its assignment from the second frame on), all
 * replicas render at the same time, the tile buffers travel to scenes[0]'s
 * device (peer copies over xGMI, 3 bytes per pixel in total), are blitted there
 * and the frame is copied to out_rgb (image_w*image_h*3 bytes, row-major, top
 * row first -- what RayTrace(int,int,Camera*,vector*) returns).  Lights must
 * have been set on every replica.  stats (nullable): n_scenes entries, the work
 * counters, kernel_ms = that replica's frame kernels, total_ms = wall time of
 * the whole call; stats[0].total_ms - max kernel_ms ~ exchange + blit + D2H.
 * The result is byte-identical to mt_render_chunk of the whole frame.
 * STATE OF TESTING: on one-GPU boxes only.  Replicas that share a device are
 * covered by the GPU tests, including the gather-buffer offsets and the blit of
 * the cross-device branch (forced through hipMemcpyPeerAsync on one device by
 * MT_TUNE_MULTI_FORCE_PEER_COPY); peer access BETWEEN devices
 * (hipDeviceCanAccessPeer / EnablePeerAccess, cross-device events) has not run
 * on hardware yet, and no multi-GPU scaling figure has been measured. */
int mt_render_frame_multi(mt_scene *const *scenes, int n_scenes,
                          const mt_sensor *sensor, int image_w, int image_h,
                          int tile_w, int tile_h, int max_depth,
                          uint8_t *out_rgb, mt_stats *stats);

/* Multi-GPU frames with a MOVING camera (the reference's loop turns it every
 * frame, main_local.cc:51-76).  A launch orders its work by the block costs of
 * the previous frame, re-projected through the camera change -- but a rank
 * measured only its own tiles, and the old-image position of a block mostly
 * lies in another rank's tile.  So the ranks exchange their costs, 4 bytes per
 * 8x8 block of the frame (the path's second, tiny exchange step; the reference's
 * master hands chunks out dynamically instead, main_net_master.cc:62-80):
 * export writes the costs of this scene's LAST launch, on a common scale, into a
 * frame-wide map d_map[map_h][map_w] of uint32 (map_w >= ceil(image_w / 8),
 * map_h >= ceil(image_h / 8); only the blocks of that launch's tiles are
 * written: zero the map first); the ranks combine their maps with an
 * element-wise MAX (torch.distributed.all_reduce / RCCL); import hands the
 * result to the scene, whose NEXT launch reads it wherever a re-projected
 * forecast needs a cost (later launches fall back to the scene's own costs
 * unless a new map is imported).  Nothing computed for a pixel depends on it.
 * mt_render_frame_multi does the same between its replicas by itself. */
int mt_scene_export_costs_device(mt_scene *scene, void *d_map, int map_w,
                                 int map_h, void *stream);
int mt_scene_import_costs_device(mt_scene *scene, const void *d_map, int map_w,
                                 int map_h, void *stream);

/* Fetches and clears the accumulated counters (kernel_ms/total_ms = 0). */
int mt_scene_read_stats(mt_scene *scene, mt_stats *stats);

/* Work counters of the *_device calls: 1 (default) = the kernels count rays,
 * node visits, box tests, requested bytes ... as they go (about 7 % of a
 * frame's time: one LDS atomic per counter and node visit); 0 = kernels built
 * without the counters (mt_scene_read_stats then returns zeros).  The image is
 * the same either way.  mt_render_chunk with a stats pointer always counts. */
int mt_scene_set_stats(mt_scene *scene, int enabled);

/* Work scheduling.  use_cost_history = 1 (default): a launch with the same
 * geometry as the previous one (image, region, tiling, recursion depth, light
 * count) hands out its 8x8-pixel blocks in the order of the costs measured in
 * that previous launch, longest first, the few longest as four quarters with
 * four lanes per pixel; everything then runs in ONE kernel.  0: every launch
 * classifies its blocks by material first (two kernels).  Either way each
 * pixel is computed by the same arithmetic: the output does not depend on it.
 * The call also forgets the recorded costs.  (No reference counterpart: the
 * reference hands rows to a thread pool, mythtracer.cc:244-290.) */
int mt_scene_set_scheduling(mt_scene *scene, int use_cost_history);

/* Frame engine.  Two implementations of TraceRayWorker's control flow exist;
 * every pixel goes through the same operations in the same order in both, so
 * the output does not depend on the choice.  1 = throughput engine: one lane
 * per pixel runs the recursion as a state machine (lowest cost per ray).  2 =
 * latency engine: the recursion of every pixel is unrolled into a per-wave pool
 * of rays, so that a call's shadow loops and child calls are traced side by
 * side (shortest chain of dependent passes per pixel; wins when a launch has
 * few blocks per wave, e.g. one rank's share of a multi-GPU frame).  3 =
 * hybrid: ONE kernel in which every wave first works through the launch's
 * longest blocks, cut into pieces, as a ray-pool wave and then through
 * everything else as a state-machine wave (needs measured block costs and has no
 * debug-buffer path: a launch without the one or with the other is rendered by
 * engine 2).  0 = automatic (default): 2 for launches without measured block
 * costs; with them, 3 for launches with fewer than 9 blocks per resident wave,
 * else 1.  Also forgets the recorded costs.  Limits of engine 2: at most 254 lights (a pool entry holds the
 * light in 8 bits), and its scratch -- per resident wave `capacity` records of
 * 160 + 80 n_lights bytes, capacity <= 1024 -- must fit a budget (4 GiB; the
 * capacity shrinks to fit, down to the ~280 records its depth-first throttle
 * needs).  The automatic mode never fails on either limit: launches the pool
 * cannot hold comfortably are rendered by engine 1, which has no such limits;
 * only an EXPLICIT engine 2 beyond them returns MT_ERR_UNSUPPORTED.
 * (No reference counterpart.) */
int mt_scene_set_engine(mt_scene *scene, int engine);
/* Process-wide default of mt_scene_set_engine for scenes created afterwards
 * (also those the C++ facade creates); 0 initially. */
int mt_set_default_engine(int engine);

/* Tuning constants of the work order and of the engine choice (defaults = what
 * the sweeps in DESIGN.md settled on).  None of them changes a pixel; tests and
 * experiment scripts use them instead of environment variables.  The call also
 * forgets the recorded costs. */
enum {
  MT_TUNE_POOL_BELOW = 0,     /* automatic engine: ray pool below this many blocks per resident wave (9) */
  MT_TUNE_POOL_CAP,           /* records per wave of the ray pool, 0 = default (tests: force the throttle) */
  MT_TUNE_PACKED_STACK,       /* 1 (default): 16-byte traversal stack frames when indices fit; 0: 20-byte */
  MT_TUNE_BLOCKS_PER_CU,      /* 0 = as many workgroups per CU as fit */
  MT_TUNE_FORECAST_RADIUS,    /* blocks; < 0 = 1, or 2 when the camera origin moved */
  MT_TUNE_BLEND,              /* damping of a repeated frame's cost forecast (0.9) */
  MT_TUNE_FORMS,              /* 1 (default): per-block ratio of the two measured cost forms */
  MT_TUNE_POOL_CUT_SHARE,     /* < 0 = 1.0 with history, 0.3 without */
  MT_TUNE_POOL_PIECE_TIME1, MT_TUNE_POOL_PIECE_TIME2,
  MT_TUNE_POOL_PIECE_WORK1, MT_TUNE_POOL_PIECE_WORK2,
  MT_TUNE_POOL_CELL_FACTOR,
  MT_TUNE_QUAD_SHARE, MT_TUNE_QUAD_SHARE_MOVING, MT_TUNE_QUAD_KEEP,
  MT_TUNE_QUAD_WORK, MT_TUNE_QUAD_WORK_MOVING,
  MT_TUNE_POOL_SCRATCH_MB,    /* scratch budget of the ray pool (4096) */
  MT_TUNE_HYBRID_POOL_SHARE,  /* engine 3: blocks above this share of an even split go to the ray pool in pieces (1.3) */
  MT_TUNE_HYBRID_QUAD_SHARE,  /* ... above this one to the state machine as quarters, four lanes per pixel (1.0 = none) */
  MT_TUNE_HYBRID_WORK1, MT_TUNE_HYBRID_WORK2, /* pool quarters / cells: summed cost over the whole block's (1.3, 3.3) */
  MT_TUNE_FORECAST_STEP,      /* pixels between the positions a re-projected forecast takes its maximum over (8) */
  MT_TUNE_HYBRID_STARTER_SHARE, /* engine 3: state-machine units above this share of an even split start with the launch,
                                   on waves that skip the pool's part (0.33; a value above every unit = none) */
  MT_TUNE_DEEP_LAYOUT,        /* 1 (default): octrees of 12 .. 16 levels keep only the first ten levels' traversal frames in LDS
                                 (the rest in global memory: 8 waves per CU instead of 7 .. 5); 0: everything in LDS */
  MT_TUNE_MULTI_FORCE_PEER_COPY, /* tests: mt_render_frame_multi copies every replica's tiles into the gather buffer with
                                    hipMemcpyPeerAsync even when it shares the first replica's device (0) */
  MT_TUNE_MULTI_BALANCE,      /* mt_render_frame_multi: 1 (default) = tiles dealt out by cost, 0 = by tile number */
  MT_TUNE_XCD_QUEUES,         /* state-machine launches ordered by cost history: 0 = one work order for the chip; 1 = eight
                                 orders, one per XCD, each over a stripe of the picture with an eighth of the forecast cost
                                 (an XCD's L2 then holds its stripe's part of the tree; an XCD that runs dry takes units
                                 from the fullest other queue); 2 = a 4 x 2 grid of regions instead of stripes */
  MT_TUNE_ORDER_GROUPS,       /* workgroups of the three kernels that make a launch's work order (forecast per block,
                                 counting sort, units longest first): 64; 1 .. 256 */
  MT_TUNE_SM_CELL_SHARE,      /* state machine: a block goes out as sixteen 2x2 cells (four lanes per pixel) when a QUARTER
                                 of it is expected above this multiple of the quarters' cutting threshold */
  MT_TUNE_SM_CELL_TIME, MT_TUNE_SM_CELL_WORK, /* a cell's expected time / the cells' summed cost, over the block as one unit */
  MT_TUNE_HYBRID_CELL_FACTOR, /* engine 3: the pool's pieces of a block are 2x2 cells when a quarter is expected above this
                                 multiple of the pool's threshold (0.85; the ray pool by itself: MT_TUNE_POOL_CELL_FACTOR) */
  MT_TUNE_COUNT
};
int mt_scene_set_tuning(mt_scene *scene, int knob, double value);

/* Device durations of the launches made since the previous call (at most the
 * last 64, oldest first; at most max_n): primary_ms[i] = the kernels that
 * prepare the work order (mt::order_kernel with cost history, else
 * mt::primary_kernel; for the latency engine's first frame mt::probe_kernel +
 * mt::order_kernel), render_ms[i] = the frame kernel
 * (mt::render_kernel or mt::pool_kernel) of launch i, from HIP events recorded on
 * the launch's own stream.  Waits for those launches.  Returns the number of
 * entries written, or a negative MT_ERR_*.  (No reference counterpart: the
 * reference times a frame with wall clocks, main_local.cc:86-101.) */
int mt_scene_kernel_times(mt_scene *scene, int max_n, double *primary_ms,
                          double *render_ms);

/* OctTree::IntersectRay (octtree.cc:26-40) for a batch: rays = n x 6 doubles
 * (origin, direction).  Outputs (each nullable): tri = stream-order triangle
 * index or -1, line_no, t, point (3 per ray; untouched = NaN on miss). */
int mt_intersect_rays(mt_scene *scene, int n, const double *rays,
                      int32_t *tri, int32_t *line_no, double *t,
                      double *point, mt_stats *stats);

/* Test hook: 0 = automatic (default: regular rays take the hit-set walk --
 * every node once per wave, children in any order, the reference's choice among
 * them by its (entry distance, index) rule; DESIGN.md section 3.1 -- and the
 * ordered descent of the modes below serves the rest), 1 = always use the exact
 * std::min/std::max comparison path, 2 = allow min/max instructions but not
 * the octant-uniform path, 3 = automatic but never the triangle-parallel
 * (transposed) node scan, 4 = automatic but without the fp32 conservative
 * pre-filter, 5 = automatic but every node through a wave step (no
 * lane-parallel scan of small nodes), 6 = automatic but without the block
 * boxes that skip runs of triangles, 7 = automatic but without the subtree
 * boxes that skip children.  Results are identical in every mode; the
 * counters box_tests / node_visits / tri_tests / mt_tests equal the
 * reference's traversal in modes 1, 2, 4 and 7 (no subtree is skipped there);
 * in the others the first three count only the nodes actually visited, and in
 * mode 0 all four are the walk's own work (it may look at a node that lies
 * behind the reference's early exit: mt_tests is then not smaller than the
 * reference's, never the other way round). */
int mt_scene_set_traversal_mode(mt_scene *scene, int mode);

#ifdef __cplusplus
}
#endif
#endif /* MYTHTRACER_HIP_H_ */
