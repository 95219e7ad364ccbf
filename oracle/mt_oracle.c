/* mt_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 * See mt_oracle.h for the rules and the parity status (pinned against the
 * compiled reference; Texture::GetColorAt alone is unpinned).
 *
 * Plain C restatement of /root/reference/VerStarting (all file:line
 * citations below are relative to that directory).  Every floating point
 * expression keeps the reference's operand order and association; build with
 * -ffp-contract=off and no -march so that no FMA is formed (the reference is
 * built that way, Makefile:1-5).
 */
#define _GNU_SOURCE
#include "mt_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_MAX_TREE_DEPTH 1024

typedef struct { double v[3]; } v3;

typedef struct {
  v3 min, max;
} box3;

typedef struct {
  char name[128];
  v3 ka, kd, ks;
  int tex; /* -1 = none */
  double ns, refl, tr;
  v3 tf;
  double ni;
} material;

typedef struct {
  char name[256];
  int w, h;
  double *rgb; /* w*h*3 doubles */
} texture;

typedef struct {
  v3 vertex[3], normal[3], uvw[3];
  box3 aabb;
  int mtl; /* -1 = nullptr */
  int line_no;
} triangle;

typedef struct {
  v3 pos, amb, dif, spec;
} light;

typedef struct node {
  int *prims;
  int n_prims;
  struct node *kids; /* NULL or 8 */
  v3 center;
  box3 aabb;
} node;

struct orc_scene {
  triangle *tris;
  int n_tris, cap_tris;
  material *mtls;
  int n_mtls, cap_mtls;
  texture *texs;
  int n_texs, cap_texs;
  light *lights;
  int n_lights;
  node root;
  int finalized;
};

static __thread char g_err[512];
static char g_err_shared[512];

static void set_err(const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  snprintf(g_err_shared, sizeof g_err_shared, "%s", msg);
}

const char *orc_last_error(void) { return g_err_shared; }

/* ------------------------------------------------------------------ math3d.h */

static inline v3 v3_make(double x, double y, double z) {
  v3 r = {{x, y, z}};
  return r;
}
static inline v3 v3_add(v3 a, v3 b) { /* math3d.h:31-33 */
  return v3_make(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]);
}
static inline v3 v3_sub(v3 a, v3 b) { /* math3d.h:35-37 */
  return v3_make(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]);
}
static inline v3 v3_neg(v3 a) { /* math3d.h:39-41 */
  return v3_make(-a.v[0], -a.v[1], -a.v[2]);
}
static inline v3 v3_mul(v3 a, v3 b) { /* math3d.h:59-61 */
  return v3_make(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2]);
}
static inline v3 v3_scale(v3 a, double n) { /* math3d.h:88-90 */
  return v3_make(a.v[0] * n, a.v[1] * n, a.v[2] * n);
}
static inline v3 v3_divs(v3 a, double n) { /* math3d.h:92-94 */
  return v3_make(a.v[0] / n, a.v[1] / n, a.v[2] / n);
}
static inline double v3_sqrlen(v3 a) { /* math3d.h:97-99 */
  return a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2];
}
static inline double v3_sqrdist(v3 self, v3 a) { /* math3d.h:105-110 */
  const double dx = a.v[0] - self.v[0];
  const double dy = a.v[1] - self.v[1];
  const double dz = a.v[2] - self.v[2];
  return dx * dx + dy * dy + dz * dz;
}
static inline double v3_dist(v3 self, v3 a) { /* math3d.h:112-114 */
  return sqrt(v3_sqrdist(self, a));
}
static inline double v3_dot(v3 self, v3 a) { /* math3d.h:116-118 */
  return a.v[0] * self.v[0] + a.v[1] * self.v[1] + a.v[2] * self.v[2];
}
static inline v3 v3_cross(v3 s, v3 a) { /* math3d.h:120-126 */
  return v3_make(s.v[1] * a.v[2] - s.v[2] * a.v[1],
                 s.v[2] * a.v[0] - s.v[0] * a.v[2],
                 s.v[0] * a.v[1] - s.v[1] * a.v[0]);
}
static inline v3 v3_norm(v3 a) { /* math3d.h:128-131 */
  const double l = sqrt(v3_sqrlen(a));
  return v3_make(a.v[0] / l, a.v[1] / l, a.v[2] / l);
}
/* std::min / std::max exactly as libstdc++ defines them (NaN behaviour). */
static inline double std_min(double a, double b) { return (b < a) ? b : a; }
static inline double std_max(double a, double b) { return (a < b) ? b : a; }
static inline double std_min3(double a, double b, double c) {
  double r = a; /* std::min(initializer_list): keep first unless later is < */
  if (b < r) r = b;
  if (c < r) r = c;
  return r;
}
static inline double std_max3(double a, double b, double c) {
  double r = a; /* std::max(initializer_list): replace when r < later */
  if (r < b) r = b;
  if (r < c) r = c;
  return r;
}

typedef struct { double m[4][4]; } m4;

static m4 m4_mul(const m4 *s, const m4 *a) { /* math3d.h:187-199 */
  m4 res;
  for (int j = 0; j < 4; j++) {
    for (int i = 0; i < 4; i++) {
      res.m[j][i] = s->m[j][0] * a->m[0][i] + s->m[j][1] * a->m[1][i] +
                    s->m[j][2] * a->m[2][i] + s->m[j][3] * a->m[3][i];
    }
  }
  return res;
}
static v3 m4_apply(const m4 *s, v3 a) { /* math3d.h:208-215 (m[0][3] x3, sic) */
  return v3_make(
      s->m[0][0] * a.v[0] + s->m[0][1] * a.v[1] + s->m[0][2] * a.v[2] + s->m[0][3],
      s->m[1][0] * a.v[0] + s->m[1][1] * a.v[1] + s->m[1][2] * a.v[2] + s->m[0][3],
      s->m[2][0] * a.v[0] + s->m[2][1] * a.v[1] + s->m[2][2] * a.v[2] + s->m[0][3]);
}
static double deg2rad(double angle) { /* math3d.h:178-180 */
  return (angle * M_PI) / 180.0;
}
static m4 rot_x_deg(double deg) { /* math3d.h:225-232,273-277 */
  const double a = deg2rad(deg);
  m4 r = {{{1.0, 0.0, 0.0, 0.0},
           {0.0, cos(a), -sin(a), 0.0},
           {0.0, sin(a), cos(a), 0.0},
           {0.0, 0.0, 0.0, 1.0}}};
  return r;
}
static m4 rot_y_deg(double deg) { /* math3d.h:234-241,279-283 */
  const double a = deg2rad(deg);
  m4 r = {{{cos(a), 0.0, sin(a), 0.0},
           {0.0, 1.0, 0.0, 0.0},
           {-sin(a), 0.0, cos(a), 0.0},
           {0.0, 0.0, 0.0, 1.0}}};
  return r;
}
static m4 rot_z_deg(double deg) { /* math3d.h:243-250,285-289 */
  const double a = deg2rad(deg);
  m4 r = {{{cos(a), -sin(a), 0.0, 0.0},
           {sin(a), cos(a), 0.0, 0.0},
           {0.0, 0.0, 1.0, 0.0},
           {0.0, 0.0, 0.0, 1.0}}};
  return r;
}

/* ------------------------------------------------------------------ aabb.cc */

static int box_contains_point(const box3 *b, v3 p) { /* aabb.cc:29-33 */
  return p.v[0] >= b->min.v[0] && p.v[0] <= b->max.v[0] &&
         p.v[1] >= b->min.v[1] && p.v[1] <= b->max.v[1] &&
         p.v[2] >= b->min.v[2] && p.v[2] <= b->max.v[2];
}
static int box_fully_contains(const box3 *b, const box3 *o) { /* aabb.cc:5-7 */
  return box_contains_point(b, o->min) && box_contains_point(b, o->max);
}
static void box_extend_point(box3 *b, v3 p) { /* aabb.cc:42-47 */
  for (int i = 0; i < 3; i++) {
    b->min.v[i] = std_min(b->min.v[i], p.v[i]);
    b->max.v[i] = std_max(b->max.v[i], p.v[i]);
  }
}
static void box_extend_box(box3 *b, const box3 *o) { /* aabb.cc:35-40 */
  for (int i = 0; i < 3; i++) {
    b->min.v[i] = std_min(b->min.v[i], o->min.v[i]);
    b->max.v[i] = std_max(b->max.v[i], o->max.v[i]);
  }
}

/* ------------------------------------------------------------------ scene */

orc_scene *orc_scene_new(void) {
  orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
  return s; /* root.aabb = {0,0,0}-{0,0,0}: aabb.h:16 + math3d.h:141 */
}

static void free_node(node *n) {
  if (n->kids) {
    for (int i = 0; i < 8; i++) free_node(&n->kids[i]);
    free(n->kids);
  }
  free(n->prims);
}

void orc_scene_free(orc_scene *s) {
  if (!s) return;
  free_node(&s->root);
  for (int i = 0; i < s->n_texs; i++) free(s->texs[i].rgb);
  free(s->texs);
  free(s->mtls);
  free(s->tris);
  free(s->lights);
  free(s);
}

static int find_material(const orc_scene *s, const char *name) {
  for (int i = 0; i < s->n_mtls; i++) {
    if (strcmp(s->mtls[i].name, name) == 0) return i;
  }
  return -1;
}

static int push_material(orc_scene *s, const material *m) {
  /* scene->materials[name] = ...  (objreader.cc:279): same name replaces. */
  int at = find_material(s, m->name);
  if (at >= 0) {
    s->mtls[at] = *m;
    return at;
  }
  if (s->n_mtls == s->cap_mtls) {
    s->cap_mtls = s->cap_mtls ? s->cap_mtls * 2 : 16;
    s->mtls = (material *)realloc(s->mtls, sizeof(material) * s->cap_mtls);
  }
  s->mtls[s->n_mtls] = *m;
  return s->n_mtls++;
}

int orc_add_material(orc_scene *s, const char *name, const double ka[3],
                     const double kd[3], const double ks[3], double ns,
                     double refl, double tr, const double tf[3], double ni) {
  material m;
  memset(&m, 0, sizeof m);
  snprintf(m.name, sizeof m.name, "%s", name);
  memcpy(m.ka.v, ka, 24);
  memcpy(m.kd.v, kd, 24);
  memcpy(m.ks.v, ks, 24);
  memcpy(m.tf.v, tf, 24);
  m.ns = ns;
  m.refl = refl;
  m.tr = tr;
  m.ni = ni;
  m.tex = -1;
  return push_material(s, &m);
}

static int find_texture(const orc_scene *s, const char *name) {
  for (int i = 0; i < s->n_texs; i++) {
    if (strcmp(s->texs[i].name, name) == 0) return i;
  }
  return -1;
}

int orc_add_texture(orc_scene *s, const char *name, int w, int h,
                    const double *rgb) {
  if (s->n_texs == s->cap_texs) {
    s->cap_texs = s->cap_texs ? s->cap_texs * 2 : 8;
    s->texs = (texture *)realloc(s->texs, sizeof(texture) * s->cap_texs);
  }
  texture *t = &s->texs[s->n_texs];
  snprintf(t->name, sizeof t->name, "%s", name);
  t->w = w;
  t->h = h;
  t->rgb = (double *)malloc(sizeof(double) * 3 * (size_t)w * h);
  memcpy(t->rgb, rgb, sizeof(double) * 3 * (size_t)w * h);
  return s->n_texs++;
}

int orc_material_set_texture(orc_scene *s, int mtl, int tex) {
  if (mtl < 0 || mtl >= s->n_mtls || tex < -1 || tex >= s->n_texs) return 0;
  s->mtls[mtl].tex = tex;
  return 1;
}

/* Triangle::CacheAABB, primitive_triangle.cc:18-24 */
static void cache_aabb(triangle *t) {
  box3 b = {t->vertex[0], t->vertex[0]};
  box_extend_point(&b, t->vertex[1]);
  box_extend_point(&b, t->vertex[2]);
  t->aabb = b;
}

/* OctTree::AddPrimitive, octtree.cc:8-14 */
static int add_triangle(orc_scene *s, const triangle *t) {
  if (s->n_tris == s->cap_tris) {
    s->cap_tris = s->cap_tris ? s->cap_tris * 2 : 1024;
    s->tris = (triangle *)realloc(s->tris, sizeof(triangle) * s->cap_tris);
  }
  s->tris[s->n_tris] = *t;
  box_extend_box(&s->root.aabb, &t->aabb);
  return s->n_tris++;
}

int orc_add_triangle(orc_scene *s, const double v[9], const double n[9],
                     const double uvw[9], int mtl, int line_no) {
  triangle t;
  memset(&t, 0, sizeof t);
  memcpy(t.vertex, v, 72);
  if (n) memcpy(t.normal, n, 72);
  if (uvw) memcpy(t.uvw, uvw, 72);
  t.mtl = mtl;
  t.line_no = line_no;
  cache_aabb(&t);
  return add_triangle(s, &t);
}

void orc_set_lights(orc_scene *s, const double *l, int n) {
  free(s->lights);
  s->lights = (light *)malloc(sizeof(light) * (n > 0 ? n : 1));
  s->n_lights = n;
  for (int i = 0; i < n; i++) memcpy(&s->lights[i], l + 12 * i, 96);
}

int orc_num_triangles(const orc_scene *s) { return s->n_tris; }
int orc_num_materials(const orc_scene *s) { return s->n_mtls; }
void orc_root_aabb(const orc_scene *s, double out[6]) {
  memcpy(out, &s->root.aabb, 48);
}
void orc_get_triangle(const orc_scene *s, int i, double out33[33], int *mtl,
                      int *line_no) {
  const triangle *t = &s->tris[i];
  memcpy(out33, t->vertex, 72);
  memcpy(out33 + 9, t->normal, 72);
  memcpy(out33 + 18, t->uvw, 72);
  memcpy(out33 + 27, &t->aabb, 48);
  *mtl = t->mtl;
  *line_no = t->line_no;
}
void orc_get_material(const orc_scene *s, int i, double o[16], int *tex,
                      char name[128]) {
  const material *m = &s->mtls[i];
  memcpy(o, m->ka.v, 24);
  memcpy(o + 3, m->kd.v, 24);
  memcpy(o + 6, m->ks.v, 24);
  o[9] = m->ns;
  o[10] = m->refl;
  o[11] = m->tr;
  memcpy(o + 12, m->tf.v, 24);
  o[15] = m->ni;
  *tex = m->tex;
  memcpy(name, m->name, 128);
}

/* ------------------------------------------------------------------ objreader.cc */

static void directory_part(const char *path, char *out, size_t n) {
  /* GetDirectoryPart, objreader.cc:192-199 */
  const char *last = NULL;
  for (const char *p = path; *p; p++) {
    if (*p == '/' || *p == '\\') last = p;
  }
  if (!last) {
    out[0] = 0;
    return;
  }
  size_t len = (size_t)(last - path);
  if (len >= n) len = n - 1;
  memcpy(out, path, len);
  out[len] = 0;
}

static void strip_eol(char *line) { /* objreader.cc:239-247 */
  char *cp = strrchr(line, '\r');
  if (cp) *cp = 0;
  cp = strrchr(line, '\n');
  if (cp) *cp = 0;
}

/* Dependency-free stand-in for Texture::LoadFromFile (texture.cc:60-109):
 * binary PPM (P6, maxval 255) only; colour = byte / 255.0 as texture.cc:100-104. */
static int load_texture_file(orc_scene *s, const char *path, const char *key) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  int w = 0, h = 0, maxv = 0;
  char magic[3] = {0};
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) {
    fclose(f);
    return -1;
  }
  int vals[3], got = 0;
  while (got < 3) {
    int c = fgetc(f);
    if (c == EOF) break;
    if (c == '#') {
      while (c != '\n' && c != EOF) c = fgetc(f);
    } else if (c >= '0' && c <= '9') {
      ungetc(c, f);
      if (fscanf(f, "%d", &vals[got]) != 1) break;
      got++;
    }
  }
  if (got != 3) {
    fclose(f);
    return -1;
  }
  w = vals[0];
  h = vals[1];
  maxv = vals[2];
  fgetc(f); /* single whitespace after maxval */
  if (w <= 0 || h <= 0 || w > 30000 || h > 30000 || maxv != 255) {
    fclose(f);
    return -1;
  }
  size_t n = (size_t)w * h * 3;
  unsigned char *px = (unsigned char *)malloc(n);
  if (fread(px, 1, n, f) != n) {
    free(px);
    fclose(f);
    return -1;
  }
  fclose(f);
  double *rgb = (double *)malloc(sizeof(double) * n);
  for (size_t i = 0; i < n; i++) rgb[i] = (double)px[i] / 255.0;
  int idx = orc_add_texture(s, key, w, h, rgb);
  free(rgb);
  free(px);
  return idx;
}

/* MtlFileReader::ReadMtlFile, objreader.cc:472-549 with its handlers
 * :289-470.  Returns 1/0. */
static int read_mtl_file(orc_scene *s, const char *path) {
  char base[1024];
  directory_part(path, base, sizeof base);
  FILE *f = fopen(path, "r");
  if (!f) {
    set_err("mtl file not found");
    return 0;
  }
  material cur;
  int have = 0;
  int ok = 1;
  char line[128];
  while (fgets(line, sizeof line, f)) {
    strip_eol(line);
    char token[16] = {0};
    if (sscanf(line, "%15s", token) != 1) continue;
    if (token[0] == '#') continue;
    double a, b, c;
    if (strcmp(token, "newmtl") == 0) { /* :289-303 */
      if (have) push_material(s, &cur);
      have = 0;
      char name[128];
      if (sscanf(line, "newmtl %127s", name) != 1) {
        ok = 0;
        break;
      }
      memset(&cur, 0, sizeof cur);
      cur.tex = -1;
      snprintf(cur.name, sizeof cur.name, "%s", name);
      have = 1;
    } else if (strcmp(token, "Ka") == 0) { /* :305-319 */
      if (!have || sscanf(line, " Ka %lf %lf %lf", &a, &b, &c) != 3) { ok = 0; break; }
      cur.ka = v3_make(a, b, c);
    } else if (strcmp(token, "Kd") == 0) { /* :321-335 */
      if (!have || sscanf(line, " Kd %lf %lf %lf", &a, &b, &c) != 3) { ok = 0; break; }
      cur.kd = v3_make(a, b, c);
    } else if (strcmp(token, "Ks") == 0) { /* :337-351 */
      if (!have || sscanf(line, " Ks %lf %lf %lf", &a, &b, &c) != 3) { ok = 0; break; }
      cur.ks = v3_make(a, b, c);
    } else if (strcmp(token, "Ns") == 0) { /* :353-367 */
      if (!have || sscanf(line, " Ns %lf", &a) != 1) { ok = 0; break; }
      cur.ns = a;
    } else if (strcmp(token, "Refl") == 0) { /* :369-383 */
      if (!have || sscanf(line, " Refl %lf", &a) != 1) { ok = 0; break; }
      cur.refl = a;
    } else if (strcmp(token, "Tr") == 0) { /* :385-399 */
      if (!have || sscanf(line, " Tr %lf", &a) != 1) { ok = 0; break; }
      cur.tr = a;
    } else if (strcmp(token, "Ni") == 0) { /* :401-415 */
      if (!have || sscanf(line, " Ni %lf", &a) != 1) { ok = 0; break; }
      cur.ni = a;
    } else if (strcmp(token, "Tf") == 0) { /* :417-431 */
      if (!have || sscanf(line, " Tf %lf %lf %lf", &a, &b, &c) != 3) { ok = 0; break; }
      cur.tf = v3_make(a, b, c);
    } else if (strcmp(token, "map_Ka") == 0) { /* :455-470, :433-453 */
      char fname[256];
      if (!have || sscanf(line, " map_Ka %255[^\n]", fname) != 1) { ok = 0; break; }
      int t = find_texture(s, fname);
      if (t < 0) {
        char tpath[1400];
        if (base[0]) snprintf(tpath, sizeof tpath, "%s/%s", base, fname);
        else snprintf(tpath, sizeof tpath, "%s", fname);
        t = load_texture_file(s, tpath, fname);
      }
      if (t < 0) {
        set_err("cannot load texture");
        ok = 0;
        break;
      }
      cur.tex = t;
    } else if (strcmp(token, "d") == 0 || strcmp(token, "illum") == 0 ||
               strcmp(token, "Ke") == 0 || strcmp(token, "map_Kd") == 0) {
      /* ReadNotImplemented, :285-287 */
    } else {
      /* unknown MTL feature: warning, keep going (:533-536) */
    }
  }
  fclose(f);
  if (!ok) {
    if (!g_err[0]) set_err("mtl parse error");
    return 0;
  }
  if (have) push_material(s, &cur); /* :546 */
  return 1;
}

typedef struct {
  v3 *p;
  int n, cap;
} v3list;

static void v3list_push(v3list *l, v3 x) {
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 1024;
    l->p = (v3 *)realloc(l->p, sizeof(v3) * l->cap);
  }
  l->p[l->n++] = x;
}

/* ObjFileReader::ReadFace, objreader.cc:96-190 */
static int read_face(orc_scene *s, const char *line, const v3list *verts,
                     const v3list *texc, const v3list *norms, int sel_mtl,
                     int line_no) {
  int vi[5], ni[5], ti[5];
  int count = 0;
  const char *p = line;
  /* `s >> token` for the leading "f": skip ws, eat non-ws. */
  while (*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r' || *p == '\n') p++;
  while (*p && !(*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r' || *p == '\n')) p++;
  /* If the stream hit EOF while reading the first token it is no longer
   * good() and the loop body never runs (:111). */
  int good = (*p != 0);
  while (good) {
    while (*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r' || *p == '\n') p++;
    if (*p == 0) break; /* extraction failed at EOF -> eof() -> break */
    char token[128];
    size_t tl = 0;
    while (*p && !(*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r' || *p == '\n')) {
      if (tl + 1 < sizeof token) token[tl++] = *p;
      p++;
    }
    token[tl] = 0;
    if (*p == 0) break; /* token ran into EOF: eofbit set -> dropped (:113-115) */
    int v = 0, vt = 0, vn = 0;
    if (sscanf(token, "%i/%i/%i", &v, &vt, &vn) != 3 &&
        sscanf(token, "%i//%i", &v, &vn) != 2 &&
        sscanf(token, "%i/%i", &v, &vt) != 2 &&
        sscanf(token, "%i", &v) != 1) {
      set_err("unsupported face format");
      return 0;
    }
    v -= 1;
    vt -= 1;
    vn -= 1;
    if (count >= 5) { /* more than 4 -> "unsupported face count" anyway */
      count++;
      continue;
    }
    vi[count] = v;
    ni[count] = vn;
    ti[count] = vt;
    count++;
  }
  if (count != 3 && count != 4) {
    set_err("unsupported face count");
    return 0;
  }
  if (count == 4) { /* :141-146 */
    vi[4] = vi[0];
    ni[4] = ni[0];
    ti[4] = ti[0];
    count = 5;
  }
  for (int i = 3; i <= count; i += 2) { /* :151 */
    triangle t;
    memset(&t, 0, sizeof t);
    for (int j = 0; j < 3; j++) {
      int idx = vi[i - 3 + j];
      if (idx < 0 || idx >= verts->n) {
        /* reference: unchecked vector index = undefined behaviour */
        set_err("face vertex index out of range (reference UB)");
        return 0;
      }
      t.vertex[j] = verts->p[idx];
    }
    if (ni[i - 3] != -1 && ni[i - 2] != -1 && ni[i - 1] != -1) {
      for (int j = 0; j < 3; j++) {
        int idx = ni[i - 3 + j];
        if (idx < 0 || idx >= norms->n) {
          set_err("face normal index out of range (reference UB)");
          return 0;
        }
        t.normal[j] = norms->p[idx];
      }
    }
    if (ti[i - 3] != -1 && ti[i - 2] != -1 && ti[i - 1] != -1) {
      for (int j = 0; j < 3; j++) {
        int idx = ti[i - 3 + j];
        if (idx < 0 || idx >= texc->n) {
          set_err("face texcoord index out of range (reference UB)");
          return 0;
        }
        t.uvw[j] = texc->p[idx];
      }
    }
    t.mtl = sel_mtl;
    t.line_no = line_no;
    cache_aabb(&t);
    add_triangle(s, &t);
  }
  return 1;
}

/* ObjFileReader::ReadObjFile, objreader.cc:201-274 */
int orc_load_obj(orc_scene *s, const char *path) {
  g_err[0] = 0;
  char base[1024];
  directory_part(path, base, sizeof base);
  FILE *f = fopen(path, "r");
  if (!f) {
    set_err("obj file not found");
    return 0;
  }
  v3list verts = {0}, texc = {0}, norms = {0};
  int sel_mtl = -1;
  int ok = 1;
  char line[128];
  for (int line_no = 0;; line_no++) {
    if (!fgets(line, sizeof line, f)) break;
    strip_eol(line);
    char token[16] = {0};
    if (sscanf(line, "%15s", token) != 1) continue;
    if (token[0] == '#') continue;
    double x, y, z;
    if (strcmp(token, "v") == 0) { /* :43-52 */
      if (sscanf(line, "v %lf %lf %lf", &x, &y, &z) != 3) { ok = 0; break; }
      v3list_push(&verts, v3_make(x, y, z));
    } else if (strcmp(token, "vn") == 0) { /* :65-74 */
      if (sscanf(line, "vn %lf %lf %lf", &x, &y, &z) != 3) { ok = 0; break; }
      v3list_push(&norms, v3_make(x, y, z));
    } else if (strcmp(token, "vt") == 0) { /* :54-63 */
      z = 0.0;
      if (sscanf(line, "vt %lf %lf %lf", &x, &y, &z) < 2) { ok = 0; break; }
      v3list_push(&texc, v3_make(x, y, z));
    } else if (strcmp(token, "f") == 0) {
      if (!read_face(s, line, &verts, &texc, &norms, sel_mtl, line_no)) { ok = 0; break; }
    } else if (strcmp(token, "mtllib") == 0) { /* :28-41 */
      char fname[256];
      if (sscanf(line, "mtllib %255[^\n]", fname) != 1) { ok = 0; break; }
      char mpath[1400];
      if (base[0]) snprintf(mpath, sizeof mpath, "%s/%s", base, fname);
      else snprintf(mpath, sizeof mpath, "%s", fname);
      if (!read_mtl_file(s, mpath)) { ok = 0; break; }
    } else if (strcmp(token, "usemtl") == 0) { /* :76-94 */
      char name[128];
      if (sscanf(line, "usemtl %127s", name) != 1) { ok = 0; break; }
      sel_mtl = find_material(s, name); /* -1: warning, keep parsing */
    } else if (strcmp(token, "s") == 0 || strcmp(token, "g") == 0 ||
               strcmp(token, "o") == 0) {
      /* ReadNotImplemented */
    } else {
      /* unknown OBJ feature: warning, continue (:262-266) */
    }
  }
  fclose(f);
  free(verts.p);
  free(texc.p);
  free(norms.p);
  if (!ok && !g_err[0]) set_err("obj parse error");
  return ok;
}

/* ------------------------------------------------------------------ octtree.cc build */

static int attempt_split(orc_scene *s, node *n, int depth) {
  /* OctTree::Node::AttemptSplit, octtree.cc:52-135; SPLIT_BOUNDARY = 16 */
  if (n->n_prims < 16) return 1;
  if (depth >= ORC_MAX_TREE_DEPTH) {
    set_err("octree deeper than ORC_MAX_TREE_DEPTH (reference would keep recursing)");
    return 0;
  }
  /* CalcCenter, :46-50 */
  for (int i = 0; i < 3; i++) {
    n->center.v[i] = n->aabb.min.v[i] + (n->aabb.max.v[i] - n->aabb.min.v[i]) / 2.0;
  }
  n->kids = (node *)calloc(8, sizeof(node));
  const v3 lo = n->aabb.min, hi = n->aabb.max, c = n->center;
  /* child boxes :61-100 — index bit0 = x high, bit1 = z high, bit2 = y high */
  for (int k = 0; k < 8; k++) {
    const int xh = k & 1, zh = (k >> 1) & 1, yh = (k >> 2) & 1;
    n->kids[k].aabb.min = v3_make(xh ? c.v[0] : lo.v[0], yh ? c.v[1] : lo.v[1], zh ? c.v[2] : lo.v[2]);
    n->kids[k].aabb.max = v3_make(xh ? hi.v[0] : c.v[0], yh ? hi.v[1] : c.v[1], zh ? hi.v[2] : c.v[2]);
  }
  int *remaining = (int *)malloc(sizeof(int) * n->n_prims);
  int n_rem = 0;
  int counts[8] = {0};
  int *where = (int *)malloc(sizeof(int) * n->n_prims);
  for (int i = 0; i < n->n_prims; i++) { /* :106-122 first-fit */
    const box3 *pb = &s->tris[n->prims[i]].aabb;
    int found = -1;
    for (int k = 0; k < 8; k++) {
      if (box_fully_contains(&n->kids[k].aabb, pb)) {
        found = k;
        break;
      }
    }
    where[i] = found;
    if (found >= 0) counts[found]++;
  }
  for (int k = 0; k < 8; k++) {
    n->kids[k].prims = (int *)malloc(sizeof(int) * (counts[k] ? counts[k] : 1));
    n->kids[k].n_prims = 0;
  }
  for (int i = 0; i < n->n_prims; i++) {
    if (where[i] >= 0) {
      node *k = &n->kids[where[i]];
      k->prims[k->n_prims++] = n->prims[i];
    } else {
      remaining[n_rem++] = n->prims[i];
    }
  }
  free(where);
  free(n->prims);
  n->prims = remaining;
  n->n_prims = n_rem;
  for (int k = 0; k < 8; k++) {
    if (!attempt_split(s, &n->kids[k], depth + 1)) return 0;
  }
  return 1;
}

int orc_finalize(orc_scene *s) { /* OctTree::Finalize, octtree.cc:16-24 */
  if (s->finalized) return 1;
  s->root.prims = (int *)malloc(sizeof(int) * (s->n_tris ? s->n_tris : 1));
  s->root.n_prims = s->n_tris;
  for (int i = 0; i < s->n_tris; i++) s->root.prims[i] = i;
  s->finalized = 1;
  return attempt_split(s, &s->root, 1);
}

static void tree_walk_info(const node *n, int depth, int *count, int *maxd) {
  (*count)++;
  if (depth > *maxd) *maxd = depth;
  if (n->kids) {
    for (int k = 0; k < 8; k++) tree_walk_info(&n->kids[k], depth + 1, count, maxd);
  }
}

void orc_tree_info(const orc_scene *s, int *n_nodes, int *depth) {
  int c = 0, d = 0;
  tree_walk_info(&s->root, 1, &c, &d);
  *n_nodes = c;
  *depth = d;
}

void orc_tree_dump(const orc_scene *s, double *aabb, double *center,
                   int32_t *first_child, int32_t *prim_begin,
                   int32_t *prim_count, int32_t *prim_ids) {
  int n_nodes, depth;
  orc_tree_info(s, &n_nodes, &depth);
  const node **queue = (const node **)malloc(sizeof(node *) * n_nodes);
  int head = 0, tail = 0, pcur = 0;
  queue[tail++] = &s->root;
  while (head < tail) {
    const node *n = queue[head];
    memcpy(aabb + 6 * head, &n->aabb, 48);
    memcpy(center + 3 * head, &n->center, 24);
    prim_begin[head] = pcur;
    prim_count[head] = n->n_prims;
    for (int i = 0; i < n->n_prims; i++) prim_ids[pcur++] = n->prims[i];
    if (n->kids) {
      first_child[head] = tail;
      for (int k = 0; k < 8; k++) queue[tail++] = &n->kids[k];
    } else {
      first_child[head] = 0;
    }
    head++;
  }
  free(queue);
}

/* ------------------------------------------------------------------ traversal */

typedef struct {
  v3 origin, direction, inv_direction; /* ray.h:12-24 */
} ray_t;

/* The slab test shared by Node::NodeIntersectRay (octtree.cc:138-167) and
 * the pre-filter of Triangle::IntersectRay (primitive_triangle.cc:83-108). */
static inline int slab_test(const box3 *b, const ray_t *r, double *dist) {
  const double t1 = (b->min.v[0] - r->origin.v[0]) * r->inv_direction.v[0];
  const double t2 = (b->max.v[0] - r->origin.v[0]) * r->inv_direction.v[0];
  const double t3 = (b->min.v[1] - r->origin.v[1]) * r->inv_direction.v[1];
  const double t4 = (b->max.v[1] - r->origin.v[1]) * r->inv_direction.v[1];
  const double t5 = (b->min.v[2] - r->origin.v[2]) * r->inv_direction.v[2];
  const double t6 = (b->max.v[2] - r->origin.v[2]) * r->inv_direction.v[2];
  const double tmax = std_min3(std_max(t1, t2), std_max(t3, t4), std_max(t5, t6));
  if (tmax < 0.0) return 0;
  const double tmin = std_max3(std_min(t1, t2), std_min(t3, t4), std_min(t5, t6));
  if (tmin > tmax) return 0;
  *dist = tmin;
  return 1;
}

/* Triangle::IntersectRay, primitive_triangle.cc:81-143 */
static inline int tri_intersect(const triangle *t, const ray_t *r, v3 *point,
                                double *distance, uint64_t *cnt) {
  double unused;
  cnt[ORC_CNT_TRI_TESTS]++;
  if (!slab_test(&t->aabb, r, &unused)) return 0;
  cnt[ORC_CNT_MT_TESTS]++;
  const v3 e1 = v3_sub(t->vertex[1], t->vertex[0]);
  const v3 e2 = v3_sub(t->vertex[2], t->vertex[0]);
  const v3 pvec = v3_cross(r->direction, e2);
  const double det = v3_dot(e1, pvec);
  if (det >= -0.00000001 && det < 0.00000001) return 0;
  const double inv_det = 1.0 / det;
  const v3 tvec = v3_sub(r->origin, t->vertex[0]);
  const double u = v3_dot(tvec, pvec) * inv_det;
  if (u < 0.0 || u > 1.0) return 0;
  const v3 qvec = v3_cross(tvec, e1);
  const double v = v3_dot(r->direction, qvec) * inv_det;
  if (v < 0.0 || u + v > 1.0) return 0;
  const double final_distance = v3_dot(e2, qvec) * inv_det;
  if (final_distance < 0.0) return 0;
  *distance = final_distance;
  *point = v3_add(r->origin, v3_scale(r->direction, *distance));
  return 1;
}

typedef struct {
  const node *n;
  double d;
} child_hit;

/* std::sort on <= 8 elements == libstdc++ __insertion_sort (bits/stl_algo.h),
 * restated literally so that NaN keys behave the same. */
static void sort_children(child_hit *a, int n) {
  if (n == 0) return;
  for (int i = 1; i < n; i++) {
    if (a[i].d < a[0].d) {
      child_hit val = a[i];
      memmove(a + 1, a, sizeof(child_hit) * (size_t)i);
      a[0] = val;
    } else {
      child_hit val = a[i];
      int last = i, next = i - 1;
      while (val.d < a[next].d) {
        a[last] = a[next];
        last = next;
        next--;
      }
      a[last] = val;
    }
  }
}

/* OctTree::Node::PrimitiveIntersectRay, octtree.cc:169-257.  Returns triangle
 * index or -1. */
#ifdef ORC_F2_PROBE
/* Diagnostic build only (scripts/f2_bound.py; never part of liboracle.so): what would the PROVABLE early-out of a
 * shadow-loop iteration (mythtracer.cc:94-156) save?  The iteration's outcome is "in shadow" as soon as
 * a node visited in the reference's order has an own-list hit that is opaque and not farther than the light, and no
 * transparent triangle lives in the subtree below it: whatever the children return replaces that hit only by a closer
 * one, which is opaque too.  The probe lets the traversal run on and counts the work below such nodes.
 * g_f2: [0] node visits of shadow rays, [1] of them below a deciding node, [2] / [3] the same for triangle tests,
 * [4] deciding nodes, [5] shadow rays, [6] shadow rays that end on an opaque hit within the light distance. */
static __thread double g_ld = -1.0;
static __thread int g_inskip = 0;
uint64_t g_f2[8];
static int subtree_transparent(const orc_scene *s, const node *n) {
  for (int i = 0; i < n->n_prims; i++) {
    const int m = s->tris[n->prims[i]].mtl;
    if (m >= 0 && s->mtls[m].tr != 0.0) return 1;
  }
  if (n->kids) {
    for (int k = 0; k < 8; k++) {
      if (subtree_transparent(s, &n->kids[k])) return 1;
    }
  }
  return 0;
}
#endif

static int node_intersect(const orc_scene *s, const node *n, const ray_t *r,
                          v3 *point, double *distance, uint64_t *cnt) {
  cnt[ORC_CNT_NODE_VISITS]++;
#ifdef ORC_F2_PROBE
  if (g_ld >= 0.0) {
    __sync_fetch_and_add(&g_f2[0], 1);
    if (g_inskip) __sync_fetch_and_add(&g_f2[1], 1);
  }
  const uint64_t f2_tri0 = cnt[ORC_CNT_TRI_TESTS];
#endif
  int closest = -1;
  double closest_d = 0.0;
  v3 closest_p = v3_make(0, 0, 0);
  for (int i = 0; i < n->n_prims; i++) { /* :177-196 */
    v3 p;
    double d;
    if (!tri_intersect(&s->tris[n->prims[i]], r, &p, &d, cnt)) continue;
    if (closest != -1 && d > closest_d) continue;
    closest = n->prims[i];
    closest_d = d;
    closest_p = p;
  }
#ifdef ORC_F2_PROBE
  int f2_mine = 0;
  if (g_ld >= 0.0) {
    __sync_fetch_and_add(&g_f2[2], cnt[ORC_CNT_TRI_TESTS] - f2_tri0);
    if (g_inskip) __sync_fetch_and_add(&g_f2[3], cnt[ORC_CNT_TRI_TESTS] - f2_tri0);
    if (!g_inskip && closest != -1 && closest_d <= g_ld && n->kids) {
      const int m = s->tris[closest].mtl;
      if ((m < 0 || s->mtls[m].tr == 0.0) && !subtree_transparent(s, n)) {
        f2_mine = g_inskip = 1;
        __sync_fetch_and_add(&g_f2[4], 1);
      }
    }
  }
#endif
  child_hit cons[8];
  int n_cons = 0;
  if (n->kids) { /* :204-211 */
    for (int k = 0; k < 8; k++) {
      double d;
      cnt[ORC_CNT_BOX_TESTS]++;
      if (!slab_test(&n->kids[k].aabb, r, &d)) continue;
      cons[n_cons].n = &n->kids[k];
      cons[n_cons].d = d;
      n_cons++;
    }
  }
  sort_children(cons, n_cons); /* :213-216 */
  for (int i = 0; i < n_cons; i++) { /* :219-247 */
    v3 p;
    double d;
    int hit = node_intersect(s, cons[i].n, r, &p, &d, cnt);
    if (hit == -1) continue;
    if (closest != -1 && d > closest_d) continue;
    closest = hit;
    closest_d = d;
    closest_p = p;
    break;
  }
#ifdef ORC_F2_PROBE
  if (f2_mine) g_inskip = 0;
#endif
  if (closest == -1) return -1;
  *point = closest_p;
  *distance = closest_d;
  return closest;
}

/* OctTree::IntersectRay, octtree.cc:26-40 */
static int tree_intersect(const orc_scene *s, v3 origin, v3 direction,
                          v3 *point, double *distance, uint64_t *cnt) {
  ray_t r;
  r.origin = origin;
  r.direction = direction;
  r.inv_direction.v[0] = 1.0 / direction.v[0];
  r.inv_direction.v[1] = 1.0 / direction.v[1];
  r.inv_direction.v[2] = 1.0 / direction.v[2];
  double dist;
  cnt[ORC_CNT_BOX_TESTS]++;
  if (!slab_test(&s->root.aabb, &r, &dist)) return -1;
  return node_intersect(s, &s->root, &r, point, distance, cnt);
}

/* ------------------------------------------------------------------ primitive_triangle.cc shading helpers */

static double area_of_triangle(double a, double b, double c) { /* :27-40 */
  double p = (a + b + c) / 2.0;
  double area_sqr = p * (p - a) * (p - b) * (p - c);
  if (area_sqr < 0.0) return 0.0;
  return sqrt(area_sqr);
}

static v3 tri_interpolate(const triangle *t, const v3 attr[3], v3 point) {
  /* Triangle::GetNormal :43-61 and Triangle::GetUVW :63-79 */
  double a = v3_dist(t->vertex[0], t->vertex[1]);
  double b = v3_dist(t->vertex[1], t->vertex[2]);
  double c = v3_dist(t->vertex[2], t->vertex[0]);
  double p0 = v3_dist(point, t->vertex[0]);
  double p1 = v3_dist(point, t->vertex[1]);
  double p2 = v3_dist(point, t->vertex[2]);
  double n0 = area_of_triangle(b, p2, p1);
  double n1 = area_of_triangle(c, p0, p2);
  double n2 = area_of_triangle(a, p1, p0);
  double n = n0 + n1 + n2;
  return v3_divs(v3_add(v3_add(v3_scale(attr[0], n0), v3_scale(attr[1], n1)),
                        v3_scale(attr[2], n2)),
                 n);
}

/* Texture::GetColorAt, texture.cc:11-58.  UNPINNED (texture.cc not buildable). */
static v3 tex_color_at(const texture *t, double u, double v) {
  if (u != u || v != v) { /* (size_t)NaN is undefined; colors.at() throws in the reference */
    return v3_make(NAN, NAN, NAN);
  }
  u = fmod(u, 1.0);
  v = fmod(v, 1.0);
  if (u < 0.0) u += 1.0;
  if (v < 0.0) v += 1.0;
  v = 1.0 - v;
  double x = u * (double)(t->w - 1);
  double y = v * (double)(t->h - 1);
  size_t bx = (size_t)x;
  size_t by = (size_t)y;
  size_t w = (size_t)t->w, h = (size_t)t->h;
  size_t x1 = (bx + 1 == w) ? bx : bx + 1;
  size_t y1 = (by + 1 == h) ? by : by + 1;
  size_t idx[4] = {bx + by * w, x1 + by * w, bx + y1 * w, x1 + y1 * w};
  v3 c[4];
  for (int i = 0; i < 4; i++) {
    if (idx[i] >= w * h) { /* colors.at() would throw */
      return v3_make(NAN, NAN, NAN);
    }
    memcpy(c[i].v, t->rgb + 3 * idx[i], 24);
  }
  double dx = fmod(x, 1.0);
  double dy = fmod(y, 1.0);
  double area[4] = {(1.0 - dx) * (1.0 - dy), dx * (1.0 - dy), (1.0 - dx) * dy, dx * dy};
  return v3_add(v3_add(v3_add(v3_scale(c[0], area[0]), v3_scale(c[1], area[1])),
                       v3_scale(c[2], area[2])),
                v3_scale(c[3], area[3]));
}

void orc_tex_color_at(const orc_scene *s, int tex, double u, double v,
                      double out[3]) {
  v3 c = tex_color_at(&s->texs[tex], u, v);
  memcpy(out, c.v, 24);
}

/* ------------------------------------------------------------------ mythtracer.cc */

/* MythTracer::TraceRayWorker, mythtracer.cc:13-228 */
static v3 trace_ray_worker(const orc_scene *s, v3 origin, v3 direction,
                           int level, int in_object, double coef,
                           int max_level, int32_t *dbg_line, double *dbg_point,
                           uint64_t *cnt) {
  v3 P;
  double t_hit;
  cnt[level == 0 ? ORC_CNT_RAYS_PRIMARY : ORC_CNT_RAYS_SECONDARY]++;
  int prim = tree_intersect(s, origin, direction, &P, &t_hit, cnt);
  if (prim == -1) { /* :23-31 */
    if (dbg_line) {
      *dbg_line = -1;
      dbg_point[0] = dbg_point[1] = dbg_point[2] = NAN;
    }
    return v3_make(0.0, 0.0, 0.0);
  }
  const triangle *tri = &s->tris[prim];
  if (dbg_line) { /* :33-36 */
    *dbg_line = tri->line_no;
    memcpy(dbg_point, P.v, 24);
  }
  cnt[ORC_CNT_SHADED_HITS]++;
  v3 normal = tri_interpolate(tri, tri->normal, P); /* :38 */
  v3 towards_camera = v3_neg(direction);
  double normal_ray_dot = v3_dot(normal, towards_camera);
  if (normal_ray_dot < 0.0) { /* :42-45 */
    normal = v3_neg(normal);
    normal_ray_dot = v3_dot(normal, towards_camera);
  }
  if (tri->mtl < 0) { /* :49-52 */
    normal_ray_dot = (normal_ray_dot + 1.0) * 0.5;
    return v3_make(normal_ray_dot, normal_ray_dot, normal_ray_dot);
  }
  const material *mtl = &s->mtls[tri->mtl];
  v3 surface_color = mtl->ka; /* :58 */
  if (mtl->tex >= 0) {         /* :59-64 */
    v3 uvw = tri_interpolate(tri, tri->uvw, P);
    v3 tc = tex_color_at(&s->texs[mtl->tex], uvw.v[0], uvw.v[1]);
    surface_color = v3_mul(surface_color, tc);
  }
  /* :68-74 */
  v3 reflected_direction =
      v3_sub(direction, v3_scale(normal, 2 * v3_dot(direction, normal)));
  v3 reflected_origin = v3_add(P, v3_scale(reflected_direction, 0.0001));

  v3 color = v3_make(0.0, 0.0, 0.0);
  for (int li = 0; li < s->n_lights; li++) { /* :78-178 */
    const light *lt = &s->lights[li];
    v3 light_direction = v3_norm(v3_sub(lt->pos, P));
    color = v3_add(color, v3_mul(lt->amb, surface_color)); /* :83-84 */
    v3 light_power = v3_make(1.0, 1.0, 1.0);
    int in_shadow = 0;
    int traversing = 0;
    v3 start_point = P;
    for (;;) { /* :94-156 */
      v3 so = v3_add(start_point, v3_scale(light_direction, 0.00001));
      double light_distance = v3_dist(start_point, lt->pos);
      v3 sp;
      double sd;
      cnt[ORC_CNT_RAYS_SHADOW]++;
#ifdef ORC_F2_PROBE
      g_ld = light_distance;
      __sync_fetch_and_add(&g_f2[5], 1);
#endif
      int sprim = tree_intersect(s, so, light_direction, &sp, &sd, cnt);
#ifdef ORC_F2_PROBE
      g_ld = -1.0;
      if (sprim != -1 && sd <= light_distance && (s->tris[sprim].mtl < 0 || s->mtls[s->tris[sprim].mtl].tr == 0.0)) __sync_fetch_and_add(&g_f2[6], 1);
#endif
      if (sprim == -1) break;
      if (sd > light_distance) break;
      /* :121 dereferences shadow_primitive->mtl unconditionally (segfault on
       * a material-less occluder).  We define that case as opaque. */
      int sm = s->tris[sprim].mtl;
      double s_tr = sm >= 0 ? s->mtls[sm].tr : 0.0;
      if (s_tr == 0.0) {
        light_power = v3_make(0.0, 0.0, 0.0);
        in_shadow = 1;
        break;
      }
      if (!traversing) { /* :129-132 */
        light_power = v3_mul(light_power, v3_scale(s->mtls[sm].tf, s_tr));
      }
      traversing = !traversing;
      start_point = v3_add(sp, v3_scale(light_direction, 0.0000001)); /* :137 */
      if (v3_sqrdist(P, start_point) > v3_sqrdist(P, lt->pos)) break; /* :141-145 */
      if (light_power.v[0] <= 0.001 && light_power.v[1] <= 0.001 &&
          light_power.v[2] <= 0.001) { /* :149-155 */
        light_power = v3_make(0.0, 0.0, 0.0);
        in_shadow = 1;
        break;
      }
    }
    light_power.v[0] = std_max(light_power.v[0], lt->amb.v[0]); /* :159-161 */
    light_power.v[1] = std_max(light_power.v[1], lt->amb.v[1]);
    light_power.v[2] = std_max(light_power.v[2], lt->amb.v[2]);
    /* :163-167 */
    color = v3_add(
        color,
        v3_mul(v3_mul(v3_scale(v3_mul(mtl->kd, surface_color),
                               v3_dot(light_direction, normal)),
                      lt->dif),
               light_power));
    if (!in_shadow) { /* :169-177 */
      double refl_dot = v3_dot(reflected_direction, towards_camera);
      if (refl_dot > 0) {
        color = v3_add(color,
                       v3_mul(v3_scale(v3_mul(mtl->ks, surface_color),
                                       pow(refl_dot, mtl->ns)),
                              lt->spec));
      }
    }
  }
  if (level < max_level && mtl->refl > 0.0 && coef > 0.01 && !in_object) { /* :181-189 */
    v3 c = trace_ray_worker(s, reflected_origin, reflected_direction, level + 1,
                            in_object, coef * mtl->refl, max_level, NULL, NULL, cnt);
    color = v3_add(color, v3_scale(c, mtl->refl));
  }
  if (level < max_level && mtl->tr > 0.0) { /* :192-225 */
    v3 refracted_direction = v3_norm(direction); /* :208-212 */
    v3 ro = v3_add(P, v3_scale(refracted_direction, 0.00001));
    v3 c = trace_ray_worker(s, ro, refracted_direction, level + 1, !in_object,
                            coef, max_level, NULL, NULL, cnt);
    color = v3_add(color, v3_scale(v3_mul(c, mtl->tf), mtl->tr));
  }
  return color;
}

void orc_v3d_to_rgb(const double v[3], uint8_t rgb[3]) { /* mythtracer.cc:235-241 */
  for (int i = 0; i < 3; i++) {
    /* NaN falls through to the cast, which is UB in C/C++; x86-64 yields 0
     * (cvttsd2si -> 0x80000000 -> low byte). */
    rgb[i] = v[i] > 1.0 ? 255 : v[i] < 0.0 ? 0 : (v[i] != v[i] ? 0 : (uint8_t)(v[i] * 255));
  }
}

void orc_sensor(const double cam[7], int width, int height, double out[12]) {
  /* Camera::GetSensor + Sensor::Reset, camera.cc:17-63 */
  const double pitch = cam[3], yaw = cam[4], roll = cam[5], aov = cam[6];
  double aov_vertical = ((double)height / (double)width) * aov;
  m4 rot_left = rot_y_deg(aov / 2.0);
  m4 rot_right = rot_y_deg(-aov / 2.0);
  m4 rot_top = rot_z_deg(aov_vertical / 2.0);
  m4 rot_bottom = rot_z_deg(-aov_vertical / 2.0);
  m4 rot_left_top = m4_mul(&rot_top, &rot_left);
  m4 rot_right_top = m4_mul(&rot_bottom, &rot_right); /* sic, :38 */
  m4 rot_left_bottom = m4_mul(&rot_bottom, &rot_left);
  v3 dir = v3_make(0.0, 0.0, 1.0);
  v3 tl = m4_apply(&rot_left_top, dir);
  v3 tr = m4_apply(&rot_right_top, dir);
  v3 bl = m4_apply(&rot_left_bottom, dir);
  m4 ry = rot_y_deg(yaw), rx = rot_x_deg(pitch), rz = rot_z_deg(roll);
  m4 ryx = m4_mul(&ry, &rx);
  m4 fr = m4_mul(&ryx, &rz);
  tl = m4_apply(&fr, tl);
  tr = m4_apply(&fr, tr);
  bl = m4_apply(&fr, bl);
  v3 ds = v3_divs(v3_sub(bl, tl), (double)height);
  v3 dp = v3_divs(v3_sub(tr, tl), (double)width);
  memcpy(out, cam, 24);
  memcpy(out + 3, tl.v, 24);
  memcpy(out + 6, ds.v, 24);
  memcpy(out + 9, dp.v, 24);
}

void orc_sensor_ray(const double sensor[12], int x, int y, double dir[3]) {
  /* Sensor::GetRay, camera.cc:65-69 */
  v3 start, ds, dp;
  memcpy(start.v, sensor + 3, 24);
  memcpy(ds.v, sensor + 6, 24);
  memcpy(dp.v, sensor + 9, 24);
  v3 d = v3_add(v3_add(start, v3_scale(ds, (double)y)), v3_scale(dp, (double)x));
  d = v3_norm(d);
  memcpy(dir, d.v, 24);
}

int orc_render_chunk(orc_scene *s, const double cam[7], int image_w,
                     int image_h, int chunk_x, int chunk_y, int chunk_w,
                     int chunk_h, int max_level, uint8_t *rgb,
                     int32_t *dbg_line, double *dbg_point,
                     uint64_t counters[ORC_CNT_COUNT], int nthreads,
                     double *seconds) {
  /* MythTracer::RayTrace(WorkChunk*), mythtracer.cc:280-312 */
  if (!s->finalized && !orc_finalize(s)) return 0;
  double sensor[12];
  orc_sensor(cam, image_w, image_h, sensor);
  v3 origin;
  memcpy(origin.v, cam, 24);
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  uint64_t total[ORC_CNT_COUNT] = {0};
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel num_threads(nthreads)
  {
    uint64_t cnt[ORC_CNT_COUNT] = {0};
#pragma omp for
    for (int j = 0; j < chunk_h; j++) {
      for (int i = 0; i < chunk_w; i++) {
        double d[3];
        orc_sensor_ray(sensor, chunk_x + i, chunk_y + j, d);
        size_t px = (size_t)j * chunk_w + i;
        v3 c = trace_ray_worker(s, origin, v3_make(d[0], d[1], d[2]), 0, 0, 1.0,
                                max_level, dbg_line ? &dbg_line[px] : NULL,
                                dbg_point ? &dbg_point[3 * px] : NULL, cnt);
        orc_v3d_to_rgb(c.v, &rgb[px * 3]);
      }
    }
#pragma omp critical
    for (int k = 0; k < ORC_CNT_COUNT; k++) total[k] += cnt[k];
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (counters) memcpy(counters, total, sizeof total);
  return 1;
}

void orc_intersect_rays(orc_scene *s, int n, const double *rays, int32_t *tri,
                        int32_t *line, double *t, double *point,
                        double *normal, double *uvw,
                        uint64_t counters[ORC_CNT_COUNT]) {
  if (!s->finalized) orc_finalize(s);
  uint64_t cnt[ORC_CNT_COUNT] = {0};
  for (int i = 0; i < n; i++) {
    v3 o = v3_make(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    v3 d = v3_make(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    v3 p = v3_make(NAN, NAN, NAN);
    double dist = NAN;
    int hit = tree_intersect(s, o, d, &p, &dist, cnt);
    tri[i] = hit;
    if (line) line[i] = hit >= 0 ? s->tris[hit].line_no : -1;
    if (t) t[i] = dist;
    if (point) memcpy(point + 3 * i, p.v, 24);
    v3 nn = v3_make(NAN, NAN, NAN), uu = nn;
    if (hit >= 0) {
      nn = tri_interpolate(&s->tris[hit], s->tris[hit].normal, p);
      uu = tri_interpolate(&s->tris[hit], s->tris[hit].uvw, p);
    }
    if (normal) memcpy(normal + 3 * i, nn.v, 24);
    if (uvw) memcpy(uvw + 3 * i, uu.v, 24);
  }
  if (counters) memcpy(counters, cnt, sizeof cnt);
}
