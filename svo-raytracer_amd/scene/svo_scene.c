/*
 * svo_scene.c -- deterministic, integer-only procedural SVO scene generator.
 *
 * Produces a sparse-voxel-octree byte pool in EXACTLY the layout the reference's
 * host code produces and its shader consumes:
 *   interior   (tag 0) 7 B : value u8 | child pointer i32 BE, relative to this node | leafMask u16 BE
 *   surface    (tag 1) 3 B : value u8 | packed normal u16 LE
 *   subdivid.  (tag 2) 7 B : value u8 | 6 zero bytes
 *   nonsurface (tag 3) 1 B : value u8
 *   (encoders: reference Octree.java:119-176; tag bits: Octree.java:589-602)
 * A parent's 8 children are contiguous, order n = x + 2y + 4z (Octree.java:42-51).
 *
 * The tree-construction RULES follow the reference builder (own code, not a copy):
 *   - classification / node-type decisions      Octree.java:527-599 (constructInnerOctree)
 *   - surface normal = sum of offsets to empty in-chunk 26-neighbours, /2 + 5 per axis,
 *     packed nx + 10 ny + 100 nz                Octree.java:620-649 (genSurfaceNormal)
 *   - "big node exposed" looks only at the 27 corner-region voxels, in-chunk
 *                                               Octree.java:651-670 (checkBigNodeExposed)
 *   - pool prefix = root + all-interior levels down to 1024^3 chunks, then per chunk
 *     8 interior 512^3 nodes followed by their sub-octrees
 *                                               Octree.java:232-353, 481-502
 *   - voxel rule: solid iff y <= h(x,z); top 5 layers take the surface material,
 *     below that material 1                     chunkgen-heightmap.comp:16-28
 * What is NOT the reference's: the height field itself (the reference reads PNG
 * height/material maps that are not shipped).  Here h(x,z) is fixed-point value-noise
 * fBm from a 32-bit integer hash, so every language twin produces the same bytes, and
 * the tree is built top-down from a min/max pyramid of h -- no dense voxel grid
 * (8192^3 dense would be 512 GiB).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC (see __graft_entry__.build()).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NODE_SIZE 7
#define LEAF_SIZE 3
#define NS_LEAF_SIZE 1
#define TASK_SIZE 512   /* OctreeThread.java:20-23 builds 512^3 sub-octrees */
#define CHUNK_SIZE 1024 /* Octree.java:39 */

typedef struct {
  uint64_t bytes;
  uint64_t interior, surface_leaf, nonsurface_leaf, subdiv_leaf;
  int32_t depth; /* log2(N) */
  int32_t hmin, hmax;
} svo_scene_stats;

typedef struct {
  uint8_t *d;
  size_t len, cap;
} buf_t;

/* family 1 ("caves"): levels of hashed balls over the terrain (below).  A level = a grid of cells of edge C; a cell
   holds at most one ball, wholly inside the cell. */
#define MAX_BALL_LEVELS 5   /* four levels of balls next to surfaces + the dust level (family 2) */
typedef struct {
  int C, G;         /* cell edge in voxels, cells per axis (N / C) */
  uint8_t *val;     /* per cell: 0xff = no ball, 0 = the ball carves (air), 1..3 = the ball is solid, of that material */
  uint8_t *pres;    /* a bit per cell: a ball is present (2 MB for the finest level of an 8192^3 world: stays in the caches) */
  uint16_t *cx, *cy, *cz, *r;
  int nocc;         /* occupancy pyramid: occ[j][cell of edge C << j] = a ball is present somewhere inside */
  uint8_t **occ;
} ball_level_t;

typedef struct {
  int N, chunk, nlev;
  uint32_t seed;
  uint16_t *h;     /* N*N heights, index z*N + x */
  uint16_t **pmin; /* pyramid: level l covers cells of size 8<<l */
  uint16_t **pmax;
  int nball;       /* 0 = family 0, the height-field terrain alone */
  ball_level_t ball[MAX_BALL_LEVELS];
} scene_t;

typedef struct {
  uint64_t interior, surface_leaf, nonsurface_leaf, subdiv_leaf;
} counts_t;

/* ---------------------------------------------------------------- height field */

static inline uint32_t mix32(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352dU;
  a ^= a >> 15; a *= 0x846ca68bU;
  a ^= a >> 16;
  return a;
}
static inline int64_t lattice16(uint32_t ix, uint32_t iz, uint32_t seed, uint32_t oct) {
  return (int64_t)(mix32(ix * 0x9E3779B1U ^ mix32(iz * 0x85EBCA77U ^ mix32(seed + oct * 0x27d4eb2fU))) & 0xFFFFU);
}
/* smooth value noise in [0,65535], all integer */
static int64_t noise16(int x, int z, int cell, uint32_t seed, uint32_t oct) {
  int ix = x / cell, iz = z / cell;
  int64_t tx = ((int64_t)(x - ix * cell) << 16) / cell;
  int64_t tz = ((int64_t)(z - iz * cell) << 16) / cell;
  int64_t sx = (((tx * tx) >> 16) * (3 * 65536 - 2 * tx)) >> 16;
  int64_t sz = (((tz * tz) >> 16) * (3 * 65536 - 2 * tz)) >> 16;
  int64_t v00 = lattice16((uint32_t)ix, (uint32_t)iz, seed, oct);
  int64_t v10 = lattice16((uint32_t)ix + 1, (uint32_t)iz, seed, oct);
  int64_t v01 = lattice16((uint32_t)ix, (uint32_t)iz + 1, seed, oct);
  int64_t v11 = lattice16((uint32_t)ix + 1, (uint32_t)iz + 1, seed, oct);
  int64_t a = v00 + (((v10 - v00) * sx) >> 16);
  int64_t b = v01 + (((v11 - v01) * sx) >> 16);
  return a + (((b - a) * sz) >> 16);
}

/* h(x,z) in [0,N): base 0.30 N plus 4 octaves with cell N/4, N/16, N/64, N/256 and
   peak-to-peak amplitude (amp_num/16) * cell.  Scale-free in N. */
int svo_scene_height(int N, uint32_t seed, int amp_num, int x, int z) {
  int64_t h = (int64_t)N * 30 / 100;
  int cell = N / 4;
  for (uint32_t o = 0; o < 4 && cell >= 4; o++, cell /= 4) {
    int64_t amp = (int64_t)cell * amp_num / 16;
    h += (amp * (noise16(x, z, cell, seed, o) - 32768)) >> 16;
  }
  if (h < 0) h = 0;
  if (h > N - 1) h = N - 1;
  return (int)h;
}

static inline uint8_t band_material_of(uint32_t seed, int x, int z) {
  return (uint8_t)(2 + (mix32((uint32_t)(x >> 5) * 0x9E3779B1U ^ mix32((uint32_t)(z >> 5) + seed * 0x61C88647U)) & 1U));
}
static inline uint8_t band_material(const scene_t *s, int x, int z) { return band_material_of(s->seed, x, z); }

/* The scene as the two maps the reference's world generator starts from (Octree.java:208-231: a height map and a
   surface-material map; chunkgen-heightmap.comp:16-28 turns them into voxels): height[z*N + x] in voxels, the
   column is solid for y <= height; material[z*N + x] = value of its top five layers.  Input of the GPU builder
   (svo_build_from_heightmap), which must then produce the bytes svo_scene_build produces. */
int svo_scene_maps(int N, uint32_t seed, int amp_num, uint16_t *height, uint8_t *material) {
  if (N < 8 || N > 8192 || (N & (N - 1)) || !height || !material) return 1;
#pragma omp parallel for schedule(static)
  for (int z = 0; z < N; z++)
    for (int x = 0; x < N; x++) {
      height[(size_t)z * N + x] = (uint16_t)svo_scene_height(N, seed, amp_num, x, z);
      material[(size_t)z * N + x] = band_material_of(seed, x, z);
    }
  return 0;
}
static inline int H(const scene_t *s, int x, int z) { return s->h[(size_t)z * s->N + x]; }
/* voxel rule of chunkgen-heightmap.comp:16-28 */
static inline uint8_t terrain_voxel(const scene_t *s, int x, int y, int z) {
  int h = H(s, x, z);
  if (y > h) return 0;
  if (h - y <= 4) return band_material(s, x, z);
  return 1;
}

/* ---------------------------------------------------------------- family 1: balls over the terrain
 *
 * The reference's own 3-D generator (chunkgen.comp:228-233: Perlin + simplex + Worley noise, thresholded) makes
 * caves and overhangs; its float noise cannot be bounded over a cube, so it cannot be built at 8192^3 without the
 * dense grid.  This family keeps the Worley part -- "solid / empty within r of a hashed feature point" -- in
 * integers, where a cube can be tested against a ball exactly:
 *
 *   state_-1 = the height-field terrain above;
 *   level k = 0..: a grid of cells of edge C_k = N >> (2 + 2k) (while C_k >= 8, at most 4 levels).  A cell may hold
 *   one ball (radius C/8 .. 3C/8, centre hashed so that the ball stays inside its cell).  The ball INVERTS what
 *   it finds: if its centre is solid in state_(k-1) it carves (value 0: craters, cave mouths under overhanging
 *   rims, holes through coarser balls), else it is solid, of a hashed material 1..3 (boulders, arches where it is
 *   half embedded, floating debris, rubble inside coarser cavities).  It exists only next to a surface of
 *   state_(k-1) -- one of six probes along the axes, at r from a solid centre (a carving ball must breach) or at
 *   2 r from an empty one, differs from the centre -- and then with probability dens / 256;
 *   state_k(v) = the ball's value if v is inside the level-k ball of its cell, else state_(k-1)(v).
 *
 * Every quantity is scale-free in N, as the terrain is. */

static inline uint32_t cell_hash(uint32_t seed, int k, int ix, int iy, int iz, uint32_t salt) {
  return mix32((uint32_t)ix * 0x9E3779B1U ^
               mix32((uint32_t)iy * 0x85EBCA77U ^ mix32((uint32_t)iz * 0xC2B2AE3DU ^ mix32(seed * 0x27d4eb2fU + (uint32_t)k * 0x165667B1U + salt))));
}

static inline int in_ball(const ball_level_t *b, size_t ci, int x, int y, int z) {
  int64_t dx = x - (int)b->cx[ci], dy = y - (int)b->cy[ci], dz = z - (int)b->cz[ci], r = b->r[ci];
  return dx * dx + dy * dy + dz * dz <= r * r;
}

/* state_(upto-1): the terrain overridden by the balls of levels 0 .. upto-1, the finest one that contains the voxel
   deciding.  Outside the world: air. */
static inline uint8_t state_upto(const scene_t *s, int upto, int x, int y, int z) {
  if ((unsigned)x >= (unsigned)s->N || (unsigned)y >= (unsigned)s->N || (unsigned)z >= (unsigned)s->N) return 0;
  for (int k = upto - 1; k >= 0; k--) {
    const ball_level_t *b = &s->ball[k];
    size_t ci = ((size_t)(z / b->C) * b->G + (size_t)(y / b->C)) * b->G + (size_t)(x / b->C);
    if ((b->pres[ci >> 3] >> (ci & 7) & 1) && in_ball(b, ci, x, y, z)) return b->val[ci];
  }
  return terrain_voxel(s, x, y, z);
}

static inline uint8_t voxel(const scene_t *s, int x, int y, int z) {
  return s->nball ? state_upto(s, s->nball, x, y, z) : terrain_voxel(s, x, y, z);
}

/* Does any ball share a voxel with the box [x0, x1] x [y0, y1] x [z0, z1] (inclusive, any alignment)?  Exact for boxes that span
   few cells of a level; a box over many cells of a level answers from that level's presence bits alone (1 = "maybe": the
   callers then take the general path, which is exact either way).  Where the answer is 0 the scene IS the terrain inside the
   box, and the height-field shortcuts of family 0 apply: nearly every voxel of an 8192^3 world. */
static int balls_touch(const scene_t *s, int x0, int y0, int z0, int x1, int y1, int z1) {
  const int N = s->N;
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (z0 < 0) z0 = 0;
  if (x1 > N - 1) x1 = N - 1;
  if (y1 > N - 1) y1 = N - 1;
  if (z1 > N - 1) z1 = N - 1;
  if (x0 > x1 || y0 > y1 || z0 > z1) return 0;
  for (int k = 0; k < s->nball; k++) {
    const ball_level_t *b = &s->ball[k];
    const int cx0 = x0 / b->C, cx1 = x1 / b->C, cy0 = y0 / b->C, cy1 = y1 / b->C, cz0 = z0 / b->C, cz1 = z1 / b->C;
    const long ncell = (long)(cx1 - cx0 + 1) * (cy1 - cy0 + 1) * (cz1 - cz0 + 1);
    for (int iz = cz0; iz <= cz1; iz++)
      for (int iy = cy0; iy <= cy1; iy++)
        for (int ix = cx0; ix <= cx1; ix++) {
          const size_t ci = ((size_t)iz * b->G + (size_t)iy) * b->G + (size_t)ix;
          if (!(b->pres[ci >> 3] >> (ci & 7) & 1)) continue;
          if (ncell > 27) return 1;
          int c0[3] = {x0, y0, z0}, c1[3] = {x1, y1, z1}, bc[3] = {b->cx[ci], b->cy[ci], b->cz[ci]};
          int64_t r2 = (int64_t)b->r[ci] * b->r[ci], dn = 0;
          for (int a = 0; a < 3; a++) {
            const int64_t near = bc[a] < c0[a] ? c0[a] - bc[a] : bc[a] > c1[a] ? bc[a] - c1[a] : 0;
            dn += near * near;
          }
          if (dn <= r2) return 1;
        }
  }
  return 0;
}

static void free_balls(scene_t *s) {
  for (int k = 0; k < s->nball; k++) {
    ball_level_t *b = &s->ball[k];
    free(b->val); free(b->pres); free(b->cx); free(b->cy); free(b->cz); free(b->r);
    if (b->occ) for (int j = 0; j < b->nocc; j++) free(b->occ[j]);
    free(b->occ);
  }
  s->nball = 0;
}

/* fill the level tables, coarse to fine (a level's balls look at the state the coarser levels left) */
/* `dens`: low 16 bits = probability / 256 of a ball in a cell next to a surface (family 1, above); bits 16.. = `dust`, the
   probability / 256 that a cell of the DUST level holds a particle (family 2): cells of edge max(8, N / 256), a particle =
   a ball of radius 1 or 2 floating in the air -- its centre and the six points r + 1 along the axes are empty -- of a hashed
   material.  Dust is the hostile case of an octree walk: a ray does not hit the particles, it descends into every coarse cell
   that holds one and steps through its fine cells (iterations per ray go up several-fold, sky rays are no longer cheap). */
static int make_balls(scene_t *s, int dens) {
  int N = s->N;
  const int dust = dens >> 16;
  dens &= 0xffff;
  int levels = 0;
  while (dens > 0 && levels < MAX_BALL_LEVELS - 1 && (N >> (2 + 2 * levels)) >= 8) levels++;
  for (int k = 0; k < levels + (dust > 0 ? 1 : 0); k++) {
    const int is_dust = k == levels;
    int C = is_dust ? (N / 256 > 8 ? N / 256 : 8) : N >> (2 + 2 * k);
    ball_level_t *b = &s->ball[k];
    memset(b, 0, sizeof *b);
    b->C = C; b->G = N / C;
    size_t n = (size_t)b->G * b->G * b->G;
    b->val = (uint8_t *)malloc(n);
    b->cx = (uint16_t *)malloc(n * 2); b->cy = (uint16_t *)malloc(n * 2);
    b->cz = (uint16_t *)malloc(n * 2); b->r = (uint16_t *)malloc(n * 2);
    if (!b->val || !b->cx || !b->cy || !b->cz || !b->r) { s->nball = k + 1; return 2; }
    int G = b->G;
#pragma omp parallel for schedule(static)
    for (int iz = 0; iz < G; iz++)
      for (int iy = 0; iy < G; iy++)
        for (int ix = 0; ix < G; ix++) {
          size_t ci = ((size_t)iz * G + iy) * G + ix;
          uint32_t hr = cell_hash(s->seed, k, ix, iy, iz, 1), hc = cell_hash(s->seed, k, ix, iy, iz, 2);
          uint32_t hp = cell_hash(s->seed, k, ix, iy, iz, 3);
          int r = is_dust ? 1 + (int)(hr & 1u) : C / 8 + (int)(hr % (uint32_t)(C / 4 + 1));
          int span = C - 2 * r; /* centre in [cell + r, cell + C - 1 - r] */
          int cx = ix * C + r + (int)((hc & 0x3ff) * (uint32_t)span >> 10);
          int cy = iy * C + r + (int)(((hc >> 10) & 0x3ff) * (uint32_t)span >> 10);
          int cz = iz * C + r + (int)(((hc >> 20) & 0x3ff) * (uint32_t)span >> 10);
          b->cx[ci] = (uint16_t)cx; b->cy[ci] = (uint16_t)cy; b->cz[ci] = (uint16_t)cz; b->r[ci] = (uint16_t)r;
          b->val[ci] = 0xff;
          if ((int)(hp & 0xff) >= (is_dust ? dust : dens)) continue;
          uint8_t sc = state_upto(s, k, cx, cy, cz);
          if (is_dust) { /* a particle floats: nothing solid at its centre nor just outside it along the axes */
            int clear = sc == 0;
            for (int a = 0; a < 6 && clear; a++) {
              static const int dx[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
              clear = state_upto(s, k, cx + (r + 1) * dx[a][0], cy + (r + 1) * dx[a][1], cz + (r + 1) * dx[a][2]) == 0;
            }
            if (clear) b->val[ci] = (uint8_t)(1 + ((hp >> 8) % 3));
            continue;
          }
          int D = sc ? r : 2 * r, near = 0;
          static const int ax[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
          for (int a = 0; a < 6 && !near; a++)
            near = (state_upto(s, k, cx + D * ax[a][0], cy + D * ax[a][1], cz + D * ax[a][2]) != 0) != (sc != 0);
          if (!near) continue;
          b->val[ci] = sc ? 0 : (uint8_t)(1 + ((hp >> 8) % 3));
        }
    b->pres = (uint8_t *)calloc((n + 7) / 8, 1);
    for (size_t i = 0; i < n; i++)
      if (b->val[i] != 0xff) b->pres[i >> 3] |= (uint8_t)(1u << (i & 7));
    /* occupancy pyramid */
    int nocc = 1;
    while ((G >> (nocc - 1)) > 1) nocc++;
    b->nocc = nocc;
    b->occ = (uint8_t **)calloc((size_t)nocc, sizeof(uint8_t *));
    b->occ[0] = (uint8_t *)malloc(n);
    for (size_t i = 0; i < n; i++) b->occ[0][i] = b->val[i] != 0xff;
    for (int j = 1; j < nocc; j++) {
      int g = G >> j, g2 = g * 2;
      b->occ[j] = (uint8_t *)malloc((size_t)g * g * g);
      for (int z = 0; z < g; z++)
        for (int y = 0; y < g; y++)
          for (int x = 0; x < g; x++) {
            uint8_t o = 0;
            for (int d = 0; d < 8; d++)
              o |= b->occ[j - 1][((size_t)(2 * z + (d >> 2)) * g2 + (2 * y + ((d >> 1) & 1))) * g2 + (2 * x + (d & 1))];
            b->occ[j][((size_t)z * g + y) * g + x] = o;
          }
    }
    s->nball = k + 1; /* the next level sees this one */
  }
  return 0;
}

/* ---------------------------------------------------------------- byte pool */

static void buf_reserve(buf_t *b, size_t extra) {
  if (b->len + extra <= b->cap) return;
  size_t nc = b->cap ? b->cap * 2 : 4096;
  while (nc < b->len + extra) nc *= 2;
  b->d = (uint8_t *)realloc(b->d, nc);
  if (!b->d) { fprintf(stderr, "svo_scene: out of memory\n"); abort(); }
  b->cap = nc;
}
static size_t put_node7(buf_t *b, uint8_t val) {
  buf_reserve(b, NODE_SIZE);
  size_t p = b->len;
  b->d[p] = val;
  memset(b->d + p + 1, 0, 6);
  b->len += NODE_SIZE;
  return p;
}
static size_t put_surface_leaf(buf_t *b, uint8_t val, uint16_t normal) {
  buf_reserve(b, LEAF_SIZE);
  size_t p = b->len;
  b->d[p] = val;
  b->d[p + 1] = (uint8_t)(normal & 0xff); /* little-endian, Octree.java:150-151 */
  b->d[p + 2] = (uint8_t)(normal >> 8);
  b->len += LEAF_SIZE;
  return p;
}
static size_t put_ns_leaf(buf_t *b, uint8_t val) {
  buf_reserve(b, NS_LEAF_SIZE);
  size_t p = b->len;
  b->d[p] = val;
  b->len += 1;
  return p;
}
static void set_cp(uint8_t *d, size_t parent, int64_t rel) { /* int32 big-endian, Octree.java:162-164 */
  uint32_t u = (uint32_t)(int32_t)rel;
  d[parent + 1] = (uint8_t)(u >> 24);
  d[parent + 2] = (uint8_t)(u >> 16);
  d[parent + 3] = (uint8_t)(u >> 8);
  d[parent + 4] = (uint8_t)u;
}
static void set_mask(uint8_t *d, size_t parent, uint16_t m) { /* u16 big-endian, Octree.java:170-172 */
  d[parent + 5] = (uint8_t)(m >> 8);
  d[parent + 6] = (uint8_t)m;
}

/* ---------------------------------------------------------------- classification */

enum { K_EMPTY = 0, K_SOLID = 1, K_MIXED = 2, K_UNKNOWN = 3 };

static inline int in_chunk(const scene_t *s, int g, int c) {
  int o = (c / s->chunk) * s->chunk;
  return g >= o && g < o + s->chunk;
}

/* ---- family 1: exact classification of a cube without its voxels ----
   prove3() decides what can be decided from the height pyramids and the cube's relation to the balls (each test exact
   in integers); what it cannot decide is resolved by recursion down to cubes of 4, which are enumerated.  The result is
   what the reference's scan over the dense voxels gives (tests/test_scene3.py: against the brute force over the grid). */

/* 0 = the cube [c, c + cs) shares no voxel with the ball, 1 = all its voxels are inside, 2 = some are */
static inline int ball_relation(const ball_level_t *b, size_t ci, int cx, int cy, int cz, int cs) {
  int c0[3] = {cx, cy, cz}, bc[3] = {b->cx[ci], b->cy[ci], b->cz[ci]};
  int64_t r2 = (int64_t)b->r[ci] * b->r[ci], dn = 0, df = 0;
  for (int a = 0; a < 3; a++) {
    int lo = c0[a], hi = c0[a] + cs - 1;
    int64_t near = bc[a] < lo ? lo - bc[a] : bc[a] > hi ? bc[a] - hi : 0;
    int64_t far = (bc[a] - lo) > (hi - bc[a]) ? (bc[a] - lo) : (hi - bc[a]);
    dn += near * near; df += far * far;
  }
  if (dn > r2) return 0;
  return df <= r2 ? 1 : 2;
}

/* the terrain alone over a cube of 8 or more: exact (see classify() below for the argument) */
static inline int terrain_kind(const scene_t *s, int cx, int cy, int cz, int cs, uint8_t *value) {
  int l = 0;
  while ((8 << l) < cs) l++;
  int w = s->N / cs;
  size_t pi = (size_t)(cz / cs) * w + (cx / cs);
  int mn = s->pmin[l][pi], mx = s->pmax[l][pi];
  if (mx < cy) { *value = 0; return K_EMPTY; }
  if (cy + cs - 1 <= mn - 5) { *value = 1; return K_SOLID; }
  *value = 0;
  return K_MIXED;
}

static int prove3(const scene_t *s, int cx, int cy, int cz, int cs, uint8_t *value) {
  uint8_t v;
  int kind = terrain_kind(s, cx, cy, cz, cs, &v);
  for (int k = 0; k < s->nball; k++) {
    const ball_level_t *b = &s->ball[k];
    if (cs > b->C) { /* the cube spans several cells of this (and of every finer) level: their balls lie wholly inside it */
      int j = 0;
      while ((b->C << j) < cs) j++;
      int g = b->G >> j;
      if (b->occ[j][((size_t)(cz / cs) * g + (size_t)(cy / cs)) * g + (size_t)(cx / cs)]) return K_UNKNOWN;
      continue;
    }
    size_t ci = ((size_t)(cz / b->C) * b->G + (size_t)(cy / b->C)) * b->G + (size_t)(cx / b->C);
    if (b->val[ci] == 0xff) continue;
    int rel = ball_relation(b, ci, cx, cy, cz, cs);
    if (rel == 0) continue;
    if (rel == 1) { v = b->val[ci]; kind = v ? K_SOLID : K_EMPTY; continue; }
    if ((kind == K_EMPTY || kind == K_SOLID) && v == b->val[ci]) continue; /* the ball changes nothing here */
    kind = K_UNKNOWN;
  }
  *value = v;
  return kind;
}

/* all voxels of the cube equal?  K_EMPTY / K_SOLID with the value, or K_MIXED */
static int homog3(const scene_t *s, int cx, int cy, int cz, int cs, uint8_t *hv) {
  if (cs < 8) {
    uint8_t first = state_upto(s, s->nball, cx, cy, cz);
    for (int z = cz; z < cz + cs; z++)
      for (int y = cy; y < cy + cs; y++)
        for (int x = cx; x < cx + cs; x++)
          if (state_upto(s, s->nball, x, y, z) != first) return K_MIXED;
    *hv = first;
    return first ? K_SOLID : K_EMPTY;
  }
  int k = prove3(s, cx, cy, cz, cs, hv);
  if (k != K_UNKNOWN) return k;
  int h = cs / 2;
  uint8_t v0 = 0;
  for (int n = 0; n < 8; n++) {
    uint8_t vn;
    if (homog3(s, cx + (n & 1) * h, cy + ((n >> 1) & 1) * h, cz + ((n >> 2) & 1) * h, h, &vn) == K_MIXED) return K_MIXED;
    if (n == 0) v0 = vn;
    else if (vn != v0) return K_MIXED;
  }
  *hv = v0;
  return v0 ? K_SOLID : K_EMPTY;
}

/* the first non-zero voxel of the cube in the reference's scan order (z outer, y, x inner; Octree.java:535-552):
   smallest (z, y, x) -- *best is the smallest key found so far */
static void first_nonzero3(const scene_t *s, int cx, int cy, int cz, int cs, uint64_t *best, uint8_t *bval) {
  uint64_t lowkey = (uint64_t)cz << 32 | (uint64_t)cy << 16 | (uint64_t)cx;
  if (lowkey >= *best) return; /* nothing in here comes earlier */
  if (cs < 8) {
    for (int z = cz; z < cz + cs; z++)
      for (int y = cy; y < cy + cs; y++)
        for (int x = cx; x < cx + cs; x++) {
          uint8_t v = state_upto(s, s->nball, x, y, z);
          if (!v) continue;
          uint64_t key = (uint64_t)z << 32 | (uint64_t)y << 16 | (uint64_t)x;
          if (key < *best) { *best = key; *bval = v; }
          return; /* later voxels of this cube come later */
        }
    return;
  }
  uint8_t v;
  int k = prove3(s, cx, cy, cz, cs, &v);
  if (k == K_EMPTY) return;
  v = state_upto(s, s->nball, cx, cy, cz);
  if (v) { *best = lowkey; *bval = v; return; }
  int h = cs / 2;
  for (int n = 0; n < 8; n++)
    first_nonzero3(s, cx + (n & 1) * h, cy + ((n >> 1) & 1) * h, cz + ((n >> 2) & 1) * h, h, best, bval);
}

/* classify region [cx,cx+cs) x [cy,cy+cs) x [cz,cz+cs); value per Octree.java:528-555 */
static int classify(const scene_t *s, int cx, int cy, int cz, int cs, uint8_t *value) {
  const int touched = s->nball && balls_touch(s, cx, cy, cz, cx + cs - 1, cy + cs - 1, cz + cs - 1);
  if (cs >= 8 && touched) {
    uint8_t hv;
    int k = homog3(s, cx, cy, cz, cs, &hv);
    if (k != K_MIXED) { *value = hv; return k; }
    uint8_t first = voxel(s, cx, cy, cz);
    if (!first) {
      uint64_t best = ~(uint64_t)0;
      first_nonzero3(s, cx, cy, cz, cs, &best, &first);
    }
    *value = first;
    return K_MIXED;
  }
  if (cs >= 8) {
    int l = 0;
    while ((8 << l) < cs) l++;
    int w = s->N / cs;
    size_t pi = (size_t)(cz / cs) * w + (cx / cs);
    int mn = s->pmin[l][pi], mx = s->pmax[l][pi];
    if (mx < cy) { *value = 0; return K_EMPTY; }
    if (cy + cs - 1 <= mn - 5) { *value = 1; return K_SOLID; }
    /* mixed (an all-solid region of edge >= 6 touching the 5-layer
       material band always also contains material 1) */
    uint8_t v0 = terrain_voxel(s, cx, cy, cz);   /* (no ball touches this cube: the scene is the terrain here) */
    if (v0) { *value = v0; return K_MIXED; }
    /* first non-zero sample in the reference's z, y, x scan order: solid voxels of a
       column form a prefix in y, so the first one is found on the y = cy plane */
    for (int z = cz; z < cz + cs; z++) {
      const uint16_t *row = s->h + (size_t)z * s->N;
      for (int x = cx; x < cx + cs; x++)
        if (row[x] >= cy) { *value = terrain_voxel(s, x, cy, z); return K_MIXED; }
    }
    *value = 0; /* unreachable: mx >= cy */
    return K_EMPTY;
  }
#define VOX(x, y, z) (touched ? voxel(s, x, y, z) : terrain_voxel(s, x, y, z))
  uint8_t first = VOX(cx, cy, cz), val = first;
  if (cs == 1) { *value = first; return first ? K_SOLID : K_EMPTY; }
  for (int z = cz; z < cz + cs; z++)
    for (int y = cy; y < cy + cs; y++)
      for (int x = cx; x < cx + cs; x++) {
        uint8_t smp = VOX(x, y, z);
        if (smp) val = smp;
        if (smp != first) {
          if (first == 0) first = smp;
          *value = first;
          return K_MIXED;
        }
      }
  *value = val;
  return val ? K_SOLID : K_EMPTY;
#undef VOX
}

/* Octree.java:620-649 */
static int surface_normal(const scene_t *s, int cx, int cy, int cz, uint16_t *packed) {
  int exposed = 0, nx = 0, ny = 0, nz = 0;
  const int touched = s->nball && balls_touch(s, cx - 1, cy - 1, cz - 1, cx + 1, cy + 1, cz + 1);
  for (int i = cx - 1; i <= cx + 1; i++) {
    if (i < 0 || i >= s->N || !in_chunk(s, i, cx)) continue;
    for (int k = cz - 1; k <= cz + 1; k++) {
      if (k < 0 || k >= s->N || !in_chunk(s, k, cz)) continue;
      int h = H(s, i, k);
      for (int j = cy - 1; j <= cy + 1; j++) {
        if (j < 0 || j >= s->N || !in_chunk(s, j, cy)) continue;
        if (touched ? voxel(s, i, j, k) == 0 : j > h) { exposed = 1; nx += i - cx; ny += j - cy; nz += k - cz; }
      }
    }
  }
  nx = nx / 2 + 5; ny = ny / 2 + 5; nz = nz / 2 + 5; /* C and Java both truncate toward zero */
  *packed = (uint16_t)(nx + ny * 10 + nz * 100);
  return exposed;
}

/* Octree.java:651-670: only coordinates {c-1, c+cs, c+cs+1} on every axis are looked at */
static int big_node_exposed(const scene_t *s, int cx, int cy, int cz, int cs) {
  const int touched = s->nball && balls_touch(s, cx - 1, cy - 1, cz - 1, cx + cs + 1, cy + cs + 1, cz + cs + 1);
  int ax[3][3] = {{cx - 1, cx + cs, cx + cs + 1}, {cy - 1, cy + cs, cy + cs + 1}, {cz - 1, cz + cs, cz + cs + 1}};
  int c0[3] = {cx, cy, cz};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++)
      for (int c = 0; c < 3; c++) {
        int x = ax[0][a], y = ax[1][b], z = ax[2][c];
        if (x < 0 || x >= s->N || !in_chunk(s, x, c0[0])) continue;
        if (y < 0 || y >= s->N || !in_chunk(s, y, c0[1])) continue;
        if (z < 0 || z >= s->N || !in_chunk(s, z, c0[2])) continue;
        if ((touched ? voxel(s, x, y, z) : terrain_voxel(s, x, y, z)) == 0) return 1;
      }
  return 0;
}

enum { T_INTERIOR = 0, T_SURFACE = 1, T_SUBDIV = 2, T_NONSURFACE = 3 };

/* restates the decision structure of Octree.java:511-608 for the node at parent_off */
static void build_children(const scene_t *s, buf_t *b, counts_t *cnt, size_t parent_off, int px, int py, int pz,
                           int size) {
  int cs = size / 2;
  if (cs == 0) return;
  /* family 1: a subtree no ball comes near (two voxels of margin: the normals look one voxel out, the exposure test two) IS the
     terrain's -- walk it as family 0, without asking again further down */
  scene_t plain;
  if (s->nball && !balls_touch(s, px - 2, py - 2, pz - 2, px + size + 1, py + size + 1, pz + size + 1)) {
    plain = *s;
    plain.nball = 0;
    s = &plain;
  }
  size_t child_off[8];
  uint8_t child_val[8];
  int child_type[8];
  uint16_t mask = 0;
  for (int n = 0; n < 8; n++) {
    int cx = px + (n & 1) * cs, cy = py + ((n >> 1) & 1) * cs, cz = pz + ((n >> 2) & 1) * cs;
    uint8_t val;
    int kind = classify(s, cx, cy, cz, cs, &val);
    int type;
    if (kind == K_SOLID) {
      if (cs == 1) {
        uint16_t nrm;
        if (surface_normal(s, cx, cy, cz, &nrm)) {
          child_off[n] = put_surface_leaf(b, val, nrm); type = T_SURFACE; cnt->surface_leaf++;
        } else {
          child_off[n] = put_ns_leaf(b, val); type = T_NONSURFACE; cnt->nonsurface_leaf++;
        }
      } else if (big_node_exposed(s, cx, cy, cz, cs)) {
        child_off[n] = put_node7(b, val); type = T_INTERIOR; cnt->interior++;
      } else {
        child_off[n] = put_node7(b, val); type = T_SUBDIV; cnt->subdiv_leaf++;
      }
    } else if (kind == K_EMPTY) {
      if (cs == 1) {
        child_off[n] = put_ns_leaf(b, val); type = T_NONSURFACE; cnt->nonsurface_leaf++;
      } else {
        child_off[n] = put_node7(b, val); type = T_SUBDIV; cnt->subdiv_leaf++;
      }
    } else {
      child_off[n] = put_node7(b, val); type = T_INTERIOR; cnt->interior++;
    }
    child_val[n] = val;
    child_type[n] = type;
    mask |= (uint16_t)(type << (n << 1));
  }
  set_cp(b->d, parent_off, (int64_t)child_off[0] - (int64_t)parent_off);
  set_mask(b->d, parent_off, mask);
  for (int n = 0; n < 8; n++) {
    if (child_val[n] != 0 && child_type[n] == T_INTERIOR) {
      int cx = px + (n & 1) * cs, cy = py + ((n >> 1) & 1) * cs, cz = pz + ((n >> 2) & 1) * cs;
      build_children(s, b, cnt, child_off[n], cx, cy, cz, cs);
    }
  }
}

/* ---------------------------------------------------------------- top of the tree */

typedef struct {
  size_t node_off;
  int x, y, z;
} chunk_t;

/* Octree.java:481-502 (fillEmptyChildren): all-interior value-1 levels down to the chunks */
static void fill_top(buf_t *b, counts_t *cnt, size_t parent, int levels, int x, int y, int z, chunk_t *chunks,
                     size_t *nchunks) {
  if (levels == 0) {
    chunks[*nchunks].node_off = parent;
    chunks[*nchunks].x = x; chunks[*nchunks].y = y; chunks[*nchunks].z = z;
    (*nchunks)++;
    return;
  }
  int cs = CHUNK_SIZE << (levels - 1);
  size_t ch[8];
  for (int i = 0; i < 8; i++) { ch[i] = put_node7(b, 1); cnt->interior++; }
  for (int i = 0; i < 8; i++)
    fill_top(b, cnt, ch[i], levels - 1, x + (i & 1) * cs, y + ((i >> 1) & 1) * cs, z + ((i >> 2) & 1) * cs, chunks,
             nchunks);
  set_cp(b->d, parent, (int64_t)ch[0] - (int64_t)parent);
}

static void free_scene(scene_t *s) {
  free_balls(s);
  if (s->pmin) for (int l = 0; l < s->nlev; l++) { free(s->pmin[l]); free(s->pmax[l]); }
  free(s->pmin); free(s->pmax); free(s->h);
}

/*
 * Build the scene. N must be a power of two in [8, 8192].  amp_num/16 = octave
 * peak-to-peak amplitude in units of the octave's cell size (8 = default terrain).
 * Returns 0 on success; *out_pool is malloc'ed (free with svo_scene_free).
 */
static int scene_setup(scene_t *s, int N, uint32_t seed, int amp_num, int dens);
static int scene_build(int N, uint32_t seed, int amp_num, int dens, uint8_t **out_pool, uint64_t *out_len, svo_scene_stats *st);

int svo_scene_build(int N, uint32_t seed, int amp_num, uint8_t **out_pool, uint64_t *out_len, svo_scene_stats *st) {
  return scene_build(N, seed, amp_num, -1, out_pool, out_len, st);
}

/* family 1: the terrain with levels of hashed balls over it (caves, overhangs, boulders, floating debris; see "family 1"
   above).  dens / 256 = the probability that a cell next to a surface holds a ball, 0 .. 256. */
int svo_scene_build3(int N, uint32_t seed, int amp_num, int dens, uint8_t **out_pool, uint64_t *out_len, svo_scene_stats *st) {
  if (dens < 0 || (dens & 0xffff) > 256 || (dens >> 16) > 256) return 1;   /* (bits 16..: the dust level of family 2, make_balls) */
  return scene_build(N, seed, amp_num, dens, out_pool, out_len, st);
}

/* the dense voxels of a family-1 scene, grid[z][y][x] (tests: the brute-force builders start from these) */
int svo_scene3_voxels(int N, uint32_t seed, int amp_num, int dens, uint8_t *grid) {
  if (N < 8 || N > 1024 || (N & (N - 1)) || dens < 0 || (dens & 0xffff) > 256 || (dens >> 16) > 256 || !grid) return 1;
  scene_t s;
  int rc = scene_setup(&s, N, seed, amp_num, dens);
  if (rc) { free_scene(&s); return rc; }
#pragma omp parallel for schedule(static)
  for (int z = 0; z < N; z++)
    for (int y = 0; y < N; y++)
      for (int x = 0; x < N; x++) grid[((size_t)z * N + y) * N + x] = voxel(&s, x, y, z);
  free_scene(&s);
  return 0;
}

/* number of balls per level and kind (tests, bench line): out[level][0] = carving, out[level][1] = solid */
int svo_scene3_ball_counts(int N, uint32_t seed, int amp_num, int dens, uint64_t out[MAX_BALL_LEVELS][2]) {
  scene_t s;
  int rc = scene_setup(&s, N, seed, amp_num, dens);
  memset(out, 0, sizeof(uint64_t) * MAX_BALL_LEVELS * 2);
  if (!rc)
    for (int k = 0; k < s.nball; k++) {
      size_t n = (size_t)s.ball[k].G * s.ball[k].G * s.ball[k].G;
      for (size_t i = 0; i < n; i++)
        if (s.ball[k].val[i] != 0xff) out[k][s.ball[k].val[i] != 0]++;
    }
  free_scene(&s);
  return rc;
}

/* the balls of a family-1 scene: out[i] = {level, cx, cy, cz, r, value (0 = carving)}, at most `max` of them, coarse levels
   first; returns how many there are (tests, and the camera bench.py places inside a cave: --camera CAVE) */
int svo_scene3_balls(int N, uint32_t seed, int amp_num, int dens, int32_t *out, int max) {
  scene_t s;
  int n = 0;
  if (scene_setup(&s, N, seed, amp_num, dens) == 0)
    for (int k = 0; k < s.nball; k++) {
      const ball_level_t *b = &s.ball[k];
      const size_t cells = (size_t)b->G * b->G * b->G;
      for (size_t i = 0; i < cells; i++) {
        if (b->val[i] == 0xff) continue;
        if (n < max && out) {
          int32_t *o = out + 6 * (size_t)n;
          o[0] = k; o[1] = b->cx[i]; o[2] = b->cy[i]; o[3] = b->cz[i]; o[4] = b->r[i]; o[5] = b->val[i];
        }
        n++;
      }
    }
  free_scene(&s);
  return n;
}

static int scene_build(int N, uint32_t seed, int amp_num, int dens, uint8_t **out_pool, uint64_t *out_len, svo_scene_stats *st) {
  if (N < 8 || N > 8192 || (N & (N - 1))) return 1;
  scene_t s;
  int rc0 = scene_setup(&s, N, seed, amp_num, dens);
  if (rc0) { free_scene(&s); return rc0; }
  int nlev = s.nlev;

  counts_t cnt;
  memset(&cnt, 0, sizeof cnt);
  buf_t pool = {0};
  size_t root = put_node7(&pool, 1); /* createDummyHead / root, value 1 */
  cnt.interior++;

  if (N <= TASK_SIZE) {
    build_children(&s, &pool, &cnt, root, 0, 0, 0, N);
  } else {
    int levels = 0;
    while ((CHUNK_SIZE << levels) < N) levels++;
    size_t nchunks_max = (size_t)1 << (3 * levels), nchunks = 0;
    chunk_t *chunks = (chunk_t *)malloc(nchunks_max * sizeof(chunk_t));
    fill_top(&pool, &cnt, root, levels, 0, 0, 0, chunks, &nchunks);
    size_t ntasks = nchunks * 8;
    buf_t *tb = (buf_t *)calloc(ntasks, sizeof(buf_t));
    counts_t *tc = (counts_t *)calloc(ntasks, sizeof(counts_t));
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < (long)ntasks; t++) {
      const chunk_t *c = &chunks[t / 8];
      int i = (int)(t % 8);
      int x = c->x + (i & 1) * TASK_SIZE, y = c->y + ((i >> 1) & 1) * TASK_SIZE, z = c->z + ((i >> 2) & 1) * TASK_SIZE;
      size_t head = put_node7(&tb[t], 1); /* dummy head, OctreeThread.java:21 */
      build_children(&s, &tb[t], &tc[t], head, x, y, z, TASK_SIZE);
    }
    /* splice, Octree.java:317-343 */
    size_t total = pool.len;
    for (size_t t = 0; t < ntasks; t++) total += tb[t].len - NODE_SIZE;
    total += nchunks * 8 * NODE_SIZE;
    if (total > 0x7fffffffULL) {
      for (size_t t = 0; t < ntasks; t++) free(tb[t].d);
      free(tb); free(tc); free(chunks); free(pool.d); free_scene(&s);
      if (out_len) *out_len = total;
      return 3; /* child pointers are signed 32-bit: pool must stay below 2^31 bytes */
    }
    buf_reserve(&pool, total - pool.len);
    for (size_t c = 0; c < nchunks; c++) {
      size_t ch[8];
      for (int i = 0; i < 8; i++) { ch[i] = put_node7(&pool, 1); cnt.interior++; }
      set_cp(pool.d, chunks[c].node_off, (int64_t)ch[0] - (int64_t)chunks[c].node_off);
      for (int i = 0; i < 8; i++) {
        buf_t *b = &tb[c * 8 + i];
        set_cp(pool.d, ch[i], (int64_t)pool.len - (int64_t)ch[i]);
        pool.d[ch[i] + 5] = b->d[5]; /* leaf mask of the dummy head */
        pool.d[ch[i] + 6] = b->d[6];
        memcpy(pool.d + pool.len, b->d + NODE_SIZE, b->len - NODE_SIZE);
        pool.len += b->len - NODE_SIZE;
        cnt.interior += tc[c * 8 + i].interior;
        cnt.surface_leaf += tc[c * 8 + i].surface_leaf;
        cnt.nonsurface_leaf += tc[c * 8 + i].nonsurface_leaf;
        cnt.subdiv_leaf += tc[c * 8 + i].subdiv_leaf;
        free(b->d);
      }
    }
    free(tb); free(tc); free(chunks);
  }

  if (st) {
    st->bytes = pool.len;
    st->interior = cnt.interior; st->surface_leaf = cnt.surface_leaf;
    st->nonsurface_leaf = cnt.nonsurface_leaf; st->subdiv_leaf = cnt.subdiv_leaf;
    int d = 0;
    while ((1 << d) < N) d++;
    st->depth = d;
    st->hmin = s.pmin[nlev - 1][0]; st->hmax = s.pmax[nlev - 1][0];
  }
  free_scene(&s);
  *out_pool = pool.d;
  *out_len = pool.len;
  return 0;
}

static int scene_setup(scene_t *s, int N, uint32_t seed, int amp_num, int dens) {
  memset(s, 0, sizeof *s);
  s->N = N; s->seed = seed; s->chunk = N < CHUNK_SIZE ? N : CHUNK_SIZE;
  s->h = (uint16_t *)malloc((size_t)N * N * sizeof(uint16_t));
  if (!s->h) return 2;
#pragma omp parallel for schedule(static)
  for (int z = 0; z < N; z++)
    for (int x = 0; x < N; x++) s->h[(size_t)z * N + x] = (uint16_t)svo_scene_height(N, seed, amp_num, x, z);
  /* min/max pyramid, level l = cells of 8<<l */
  int nlev = 0;
  while ((8 << nlev) <= N) nlev++;
  s->nlev = nlev;
  s->pmin = (uint16_t **)calloc((size_t)nlev, sizeof(uint16_t *));
  s->pmax = (uint16_t **)calloc((size_t)nlev, sizeof(uint16_t *));
  for (int l = 0; l < nlev; l++) {
    int cs = 8 << l, w = N / cs;
    s->pmin[l] = (uint16_t *)malloc((size_t)w * w * 2);
    s->pmax[l] = (uint16_t *)malloc((size_t)w * w * 2);
    if (l == 0) {
#pragma omp parallel for schedule(static)
      for (int cz = 0; cz < w; cz++)
        for (int cx = 0; cx < w; cx++) {
          int mn = 65535, mx = 0;
          for (int z = cz * 8; z < cz * 8 + 8; z++)
            for (int x = cx * 8; x < cx * 8 + 8; x++) {
              int h = s->h[(size_t)z * N + x];
              if (h < mn) mn = h;
              if (h > mx) mx = h;
            }
          s->pmin[0][(size_t)cz * w + cx] = (uint16_t)mn;
          s->pmax[0][(size_t)cz * w + cx] = (uint16_t)mx;
        }
    } else {
      int w2 = w * 2;
      for (int cz = 0; cz < w; cz++)
        for (int cx = 0; cx < w; cx++) {
          int mn = 65535, mx = 0;
          for (int dz = 0; dz < 2; dz++)
            for (int dx = 0; dx < 2; dx++) {
              size_t i = (size_t)(cz * 2 + dz) * w2 + (cx * 2 + dx);
              if (s->pmin[l - 1][i] < mn) mn = s->pmin[l - 1][i];
              if (s->pmax[l - 1][i] > mx) mx = s->pmax[l - 1][i];
            }
          s->pmin[l][(size_t)cz * w + cx] = (uint16_t)mn;
          s->pmax[l][(size_t)cz * w + cx] = (uint16_t)mx;
        }
    }
  }

  if (dens >= 0) return make_balls(s, dens);
  return 0;
}

void svo_scene_free(uint8_t *p) { free(p); }

/* ---------------------------------------------------------------- validator */

static inline int tag_size(int tag) { return tag == 1 ? LEAF_SIZE : tag == 3 ? NS_LEAF_SIZE : NODE_SIZE; }

/*
 * Walk every reachable node; check that child blocks lie inside the pool and count
 * node types.  Returns 0 if consistent, else a negative error code; *max_depth gets
 * the deepest level reached (root = 0).
 */
int svo_pool_validate(const uint8_t *pool, uint64_t len, svo_scene_stats *st, int *max_depth) {
  typedef struct { uint64_t off; int depth; } item_t;
  size_t cap = 1024, sp = 0;
  item_t *stack = (item_t *)malloc(cap * sizeof(item_t));
  counts_t cnt;
  memset(&cnt, 0, sizeof cnt);
  int maxd = 0, rc = 0;
  if (len < NODE_SIZE) { free(stack); return -1; }
  stack[sp].off = 0; stack[sp].depth = 0; sp++;
  cnt.interior = 1;
  while (sp) {
    item_t it = stack[--sp];
    const uint8_t *p = pool + it.off;
    int32_t cp = (int32_t)((uint32_t)p[1] << 24 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 8 | p[4]);
    uint16_t mask = (uint16_t)(p[5] << 8 | p[6]);
    if (cp == 0) continue;
    int64_t c = (int64_t)it.off + cp;
    if (it.depth + 1 > maxd) maxd = it.depth + 1;
    for (int n = 0; n < 8; n++) {
      int tag = (mask >> (2 * n)) & 3, sz = tag_size(tag);
      if (c < 0 || (uint64_t)c + (uint64_t)sz > len) { rc = -2; goto done; }
      if (tag == 0) {
        cnt.interior++;
        if (sp + 1 >= cap) { cap *= 2; stack = (item_t *)realloc(stack, cap * sizeof(item_t)); }
        if (pool[c] != 0 || 1) { stack[sp].off = (uint64_t)c; stack[sp].depth = it.depth + 1; sp++; }
      } else if (tag == 1) cnt.surface_leaf++;
      else if (tag == 2) cnt.subdiv_leaf++;
      else cnt.nonsurface_leaf++;
      c += sz;
    }
  }
done:
  free(stack);
  if (st) {
    st->bytes = len; st->interior = cnt.interior; st->surface_leaf = cnt.surface_leaf;
    st->nonsurface_leaf = cnt.nonsurface_leaf; st->subdiv_leaf = cnt.subdiv_leaf;
  }
  if (max_depth) *max_depth = maxd;
  return rc;
}
