/*
 * svo_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the reference's hot path
 * (the compute shader /root/reference/src/shaders/svotrace.comp), single-threaded,
 * plain C.  It is the CHECKER for the HIP path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path never calls it.
 *
 * Parity pin: this file is validated bit-for-bit (rgba8, depth bits, hit pointer,
 * value, raw normal, level, iteration count) against the reference shader itself,
 * executed unmodified by Mesa llvmpipe through oracle/llvmpipe_ref.c; the resulting
 * golden vectors are committed under tests/golden/ (tests/golden/make_golden.py).
 *
 * It deliberately keeps the reference's STRUCTURE (one byte fetch at a time, the
 * while-loop child offset walk, the 24-entry stack) so that it shares no code and no
 * shortcuts with the optimised device kernels it checks.
 *
 * Float semantics GLSL leaves open are pinned to what the llvmpipe run does
 * (SURVEY.md Appendix B/C): IEEE binary32, no contraction (build with
 * -ffp-contract=off), mix = x + t*(y-x), dot3 = x*a + (y*b + z*c),
 * normalize = v * (1/sqrt(dot)), min/max ignore NaN (fminf/fmaxf), sign(NaN)=0,
 * transcendental algorithms of Appendix C with explicit fmaf where they fuse.
 *
 * Reference map (file:line in /root/reference/src/shaders/svotrace.comp):
 *   get_byte            :75-79      extract_*          :88-130
 *   extract_child       :132-157    intersect_octree   :211-432
 *   glsl_rand           :26-29      trace              :435-646
 *   pixel (main)        :649-729
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define EPSILON 3.552713678800501e-15f
#define PI_F 3.14159265359f
#define SQRT3_F 1.73205080757f
#define NODE_SIZE 7u
#define LEAF_SIZE 3u
#define NON_SURFACE_LEAF_SIZE 1u
#define MAX_SCALE 23
#define MAX_DEPTH 13
#define MAX_RAYCAST_ITERATIONS 1500u
#define BEAM_BLOCK 4 /* Main.java:41 beamSquareSize */
#define MAX_FRAME_ITER 100
/* pinned by tests/golden/accum_golden.npz (the reference shader with :712-719 switched on, under llvmpipe): imageLoad of
   an rgba8 texel is byte * (1/255) (not byte / 255: 115 pixels of the goldens tell them apart), the division by
   frameNumber + 1 is a true IEEE division (a multiply by the reciprocal misses 48) */
#define UNORM8_TO_FLOAT(b) ((float)(b) * (1.0f / 255.0f))
#define PROG_DIV(a, b) ((a) / (b))

typedef struct svo_hit {
  uint32_t pointer;    /* byte offset of the hit node; 0 = miss */
  uint16_t raw_normal; /* leafMask field of the hit node (packed normal for tag-1 leaves) */
  uint8_t value;
  uint8_t level;       /* MAX_SCALE - scale */
  uint32_t iter;
  float t;
} svo_hit;

typedef struct svo_oracle_params {
  int32_t width, height;
  float cam[15]; /* pos, l1, l2, r1, r2 (Camera.java:142-151) */
  int32_t frame_number, render_mode, buffer_end, use_beam;
  int32_t bounces;      /* path segments in mode 0; reference live value 2 (svotrace.comp:444) */
  uint32_t mirror_mask; /* bit v set: material v reflects specularly (dormant svotrace.comp:500-504) */
  int32_t spp;          /* samples per pixel (dormant SAMPLES loop, svotrace.comp:668-670); live value 1 */
  int32_t progressive;  /* cross-frame accumulation (dormant, commented out at svotrace.comp:712-719); live value 0.
                           When set, `rgba` is read as the previous frame's image before it is overwritten. */
} svo_oracle_params;

typedef struct svo_oracle_stats {
  uint64_t pixels;
  uint64_t rays;       /* intersectOctree casts, all-NaN rays excluded */
  uint64_t nan_rays;   /* casts whose origin or direction is entirely NaN (quirk Q7) */
  uint64_t iterations; /* loop iterations of counted rays */
  uint64_t alg_bytes;  /* 7 per cast (root record) + size of every fetched child record */
  uint64_t max_iter;   /* largest iteration count of a counted ray */
  uint64_t descends, advances, pops; /* iteration mix of counted rays (diagnostic) */
  uint64_t push_by_scale[24], pop_by_scale[24]; /* stack traffic per level (diagnostic) */
  uint64_t cold_pops;  /* pops inside the octree that read a stack level their ray never pushed: the zero-initialised
                          entry {node 0, t_max 0} is what the walk goes on with (diagnostic; finds cases for the tests of
                          the HIP stack's zeroed column) */
} svo_oracle_stats;

typedef struct { float x, y, z; } vec3;

typedef struct {
  const uint8_t *pool;
  uint64_t len;
  svo_oracle_stats *st;
  int cur_nan;
} ctx_t;

/* ------------------------------------------------------------------ GLSL helpers */

static inline float gmin(float a, float b) { return fminf(a, b); }
static inline float gmax(float a, float b) { return fmaxf(a, b); }
static inline float gsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float gmix(float x, float y, float t) { return x + t * (y - x); }
static inline float dot3(vec3 a, vec3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 vscale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vmul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 gnormalize(vec3 v) {
  float s = dot3(v, v);
  float r = 1.0f / sqrtf(s);
  return vscale(v, r);
}
static inline vec3 gcross(vec3 a, vec3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline int find_msb(uint32_t v) { return v ? 31 - __builtin_clz(v) : -1; }

/* llvmpipe's sin/cos: Cephes-style, 3-step Cody-Waite with fma */
static float sincos_core(float x, int want_cos) {
  float ax = fabsf(x);
  float y = ax * 1.27323954473516f;
  /* the conversion is x86's cvttps2dq: out of range (|x| >= 1.69e9) and NaN give 0x80000000 */
  uint32_t ju = (y >= 2147483648.0f || y != y) ? 0x80000000u : (uint32_t)(int)y;
  ju = (ju + 1u) & ~1u;
  int j = (int)ju;
  y = (float)j;
  float r = fmaf(y, -0.78515625f, ax);
  r = fmaf(y, -2.4187564849853515625e-4f, r);
  r = fmaf(y, -3.77489497744594108e-8f, r);
  int neg, use_cos_poly;
  if (want_cos) {
    int j2 = j - 2;
    neg = ((~j2) & 4) != 0;
    use_cos_poly = (j2 & 2) != 0;
  } else {
    neg = (((j & 4) != 0) ^ (signbit(x) != 0));
    use_cos_poly = (j & 2) != 0;
  }
  float z = r * r, v;
  if (use_cos_poly) {
    float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = fmaf(p, z, 4.166664568298827e-2f);
    p = p * z;
    p = p * z;
    p = fmaf(z, -0.5f, p);
    v = p + 1.0f;
  } else {
    float q = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    q = fmaf(q, z, -1.6666654611e-1f);
    q = q * z;
    v = fmaf(q, r, r);
  }
  v = neg ? -v : v;
  /* llvmpipe clamps the result to [-1, 1] (visible once the reduced argument is no longer small: |x| > ~6e7);
     round 2, tools/probes/sin_probe.comp */
  if (v > 1.0f) v = 1.0f;
  if (v < -1.0f) v = -1.0f;
  return v;
}
static float gsin(float x) { return sincos_core(x, 0); }
static float gcos(float x) { return sincos_core(x, 1); }
static float gacos(float x) {
  float ax = fabsf(x);
  float t = ax * (-0.02363318f) + 0.08132463f;
  t = ax * t + (0.785398163f - 1.0f); /* (float)(pi/4) - 1 = -0.21460181 (round 1 had mistyped it as -0.2145988: a 3e-6
                                         error that only the < 0.4 test of svotrace.comp:540 ever saw; tools/probes/fn_probe.comp) */
  t = ax * t + 1.5707964f;
  return 1.5707964f - gsign(x) * (1.5707964f - sqrtf(1.0f - ax) * t);
}
static float gexp2(float y) {
  if (y != y) return y;   /* NaN in, NaN out (the clamps below would otherwise swallow it) */
  y = gmin(y, 128.0f);
  y = gmax(y, -126.99999f);
  float ip = floorf(y);
  float fp = y - ip;
  float t2 = fp * fp;
  float e = fmaf(t2, 0.00898934009049466391101f, 0.240153617044375388211f);
  e = fmaf(t2, e, 1.0f);
  float o = fmaf(t2, 0.00187757667519147912699f, 0.0558263180532956664775f);
  o = fmaf(t2, o, 0.693153073200168932794f);
  /* 2^ip is built in the exponent field, (ip + 127) << 23: ip = -127 gives 0.0 (not the denormal 2^-127), ip = 128 +inf */
  return u2f((uint32_t)((int)ip + 127) << 23) * fmaf(o, fp, e);
}
/* svotrace.comp:26-29 */
static float rand_from_dot(float d) {
  float s = gsin(d);
  float v = s * 43758.5453f;
  return v - floorf(v);
}
static float glsl_rand(float cx, float cy) { return rand_from_dot(cx * 12.9898f + cy * 78.233f); }
/* rand(vec2(a, seed2 * k)) as the reference's compiler evaluates it: the two constant
   factors of (seed2 * k) * 78.233 are folded into one float constant first (Mesa NIR
   algebraic rule fmul(fmul(a, #b), #c) -> fmul(a, #b * #c)); observable from frame 7 on */
static float glsl_rand_scaled(float a, float seed2, float k) {
  float kc = k * 78.233f;
  return rand_from_dot(a * 12.9898f + seed2 * kc);
}

/* ------------------------------------------------------------------ node decode */

typedef struct {
  int value;
  int cp;
  int leafMask;
  uint32_t descriptor;
} Node;

/* :75-79 -- the pool is viewed as little-endian dwords, so byte p is pool[p];
   reads past the uploaded bytes return 0 (the reference buffer is over-allocated
   and zero-filled, Octree.java:63-67) */
static inline int get_byte(const ctx_t *c, uint32_t p) { return p < c->len ? (int)c->pool[p] : 0; }

static Node extract_node(const ctx_t *c, uint32_t p) { /* :88-101 */
  Node n;
  n.descriptor = p;
  n.value = get_byte(c, p);
  n.cp = (int)(((uint32_t)get_byte(c, p + 1) << 24) | ((uint32_t)get_byte(c, p + 2) << 16) |
               ((uint32_t)get_byte(c, p + 3) << 8) | (uint32_t)get_byte(c, p + 4));
  n.leafMask = (get_byte(c, p + 5) << 8) | get_byte(c, p + 6);
  return n;
}
static Node extract_leaf(const ctx_t *c, uint32_t p) { /* :103-108 */
  Node n = {get_byte(c, p), 0, 0, p};
  n.leafMask = get_byte(c, p + 1) | (get_byte(c, p + 2) << 8);
  return n;
}
static Node extract_non_surface_leaf(const ctx_t *c, uint32_t p) { /* :110-114 */
  Node n = {get_byte(c, p), 0, 0, p};
  return n;
}
static Node extract_subdividable_leaf(const ctx_t *c, uint32_t p) { /* :116-130 */
  Node n = extract_node(c, p);
  n.cp = 0;
  return n;
}
/* :132-157 */
static Node extract_child(const ctx_t *c, uint32_t parentPointer, uint32_t childPointer, uint32_t child, int leafMask,
                          uint32_t *endPointer, uint32_t *rec_size) {
  uint32_t i = 0;
  uint32_t pointer = childPointer + parentPointer;
  while (i < child) {
    int localMask = (leafMask & (0x0003 << (i << 1))) >> (i << 1);
    if (localMask == 0 || localMask == 2) pointer += NODE_SIZE;
    else if (localMask == 1) pointer += LEAF_SIZE;
    else pointer += NON_SURFACE_LEAF_SIZE;
    i++;
  }
  *endPointer = pointer;
  int localMask = (leafMask & (0x0003 << (child << 1))) >> (child << 1);
  if (localMask == 0) { *rec_size = NODE_SIZE; return extract_node(c, pointer); }
  if (localMask == 1) { *rec_size = LEAF_SIZE; return extract_leaf(c, pointer); }
  if (localMask == 2) { *rec_size = NODE_SIZE; return extract_subdividable_leaf(c, pointer); }
  *rec_size = NON_SURFACE_LEAF_SIZE;
  return extract_non_surface_leaf(c, pointer);
}

/* ------------------------------------------------------------------ traversal */

typedef struct {
  uint32_t value, pointer, iter;
  float t;
  vec3 hitPos;
  float scale;
  vec3 debugColor;
  vec3 normal;
  vec3 voxelPos;
  uint32_t depth;
  int leafMask; /* not in the reference struct: raw leafMask field of the hit node, for hit records */
} castResult;

typedef struct {
  Node node;
  float tmax;
} stackEntry;

static inline int all_nan3(vec3 v) { return isnan(v.x) && isnan(v.y) && isnan(v.z); }

/* :211-432.  `invdir` of the reference is unused and omitted. */
/* t_start: 0 everywhere in the reference.  With the beam pre-pass (use_beam, see svo_oracle_beam below) the primary
   ray of a pixel starts its walk at the conservative distance of its 4x4 block: same origin, same coefficients,
   only t_min is raised -- every later t is derived from cell corners, so the hit record keeps its bits. */
static int intersect_octree(ctx_t *c, vec3 origin, vec3 dir, castResult *res, int maxDepth, int coneTrace, float t_start) {
  stackEntry octstack[MAX_SCALE + 1];
  memset(octstack, 0, sizeof octstack);
  uint32_t pushed = 0; /* diagnostic only (cold_pops) */
  res->debugColor = v3(0.3f, 0.3f, 0.6f);
  Node parent = extract_node(c, 0);
  uint32_t iter = 0;
  uint64_t bytes = NODE_SIZE;
  int is_nan_ray = all_nan3(origin) || all_nan3(dir);

  if (fabsf(dir.x) < EPSILON) dir.x = EPSILON * gsign(dir.x);
  if (fabsf(dir.y) < EPSILON) dir.y = EPSILON * gsign(dir.y);
  if (fabsf(dir.z) < EPSILON) dir.z = EPSILON * gsign(dir.z);

  float tx_coef = 1.0f / -fabsf(dir.x);
  float ty_coef = 1.0f / -fabsf(dir.y);
  float tz_coef = 1.0f / -fabsf(dir.z);

  float tx_bias = tx_coef * origin.x;
  float ty_bias = ty_coef * origin.y;
  float tz_bias = tz_coef * origin.z;

  uint32_t octant_mask = 0;
  if (dir.x > 0.0f) { octant_mask ^= 1u; tx_bias = 3.0f * tx_coef - tx_bias; }
  if (dir.y > 0.0f) { octant_mask ^= 2u; ty_bias = 3.0f * ty_coef - ty_bias; }
  if (dir.z > 0.0f) { octant_mask ^= 4u; tz_bias = 3.0f * tz_coef - tz_bias; }

  float t_min = gmax(gmax(2.0f * tx_coef - tx_bias, 2.0f * ty_coef - ty_bias), 2.0f * tz_coef - tz_bias);
  float t_max = gmin(gmin(tx_coef - tx_bias, ty_coef - ty_bias), tz_coef - tz_bias);
  t_min = gmax(t_min, 0.0f);
  t_min = gmax(t_min, t_start);
  float h = t_max;

  uint32_t idx = 0;
  vec3 pos = v3(1.0f, 1.0f, 1.0f);
  int scale = MAX_SCALE - 1;
  float scale_exp2 = 0.5f;
  int child_descriptor = 0;

  if (1.5f * tx_coef - tx_bias > t_min) { idx ^= 1u; pos.x = 1.5f; }
  if (1.5f * ty_coef - ty_bias > t_min) { idx ^= 2u; pos.y = 1.5f; }
  if (1.5f * tz_coef - tz_bias > t_min) { idx ^= 4u; pos.z = 1.5f; }
  uint32_t child_shift = 0;
  int ok = 0, capped = 0;

  while (scale < MAX_SCALE) {
    iter++;
    if (iter > MAX_RAYCAST_ITERATIONS) { capped = 1; break; }
    if (child_descriptor == 0) child_descriptor = parent.cp;
    if (t_min > 0.05f && coneTrace) maxDepth = 11;

    float tx_corner = pos.x * tx_coef - tx_bias;
    float ty_corner = pos.y * ty_coef - ty_bias;
    float tz_corner = pos.z * tz_coef - tz_bias;
    float tc_max = gmin(gmin(tx_corner, ty_corner), tz_corner);

    child_shift = idx ^ octant_mask;
    uint32_t rec;
    Node child = extract_child(c, parent.descriptor, (uint32_t)child_descriptor, child_shift, parent.leafMask,
                               &res->pointer, &rec);
    bytes += rec;
    if (child.value != 0 && t_min <= t_max) {
      if (MAX_SCALE - scale == maxDepth) { ok = 1; break; }
      float tv_max = gmin(t_max, tc_max);
      float one_half = scale_exp2 * 0.5f;
      float tx_center = one_half * tx_coef + tx_corner;
      float ty_center = one_half * ty_coef + ty_corner;
      float tz_center = one_half * tz_coef + tz_corner;
      if (t_min <= tv_max) {
        if (child.cp == 0) { ok = 1; break; }
        if (tc_max < h && scale >= 0 && scale <= MAX_SCALE) { /* guard: index is always 11..22 for pools <= 13 levels deep */
          if (!is_nan_ray) c->st->push_by_scale[scale]++;
          octstack[scale].node = parent;
          octstack[scale].tmax = t_max;
          pushed |= 1u << scale;
        }
        h = tc_max;
        if (!is_nan_ray) c->st->descends++;
        parent = child;
        idx = 0u;
        --scale;
        scale_exp2 = one_half;
        if (tx_center > t_min) { idx ^= 1u; pos.x += scale_exp2; }
        if (ty_center > t_min) { idx ^= 2u; pos.y += scale_exp2; }
        if (tz_center > t_min) { idx ^= 4u; pos.z += scale_exp2; }
        t_max = tv_max;
        child_descriptor = 0;
        continue;
      }
    }
    /* ADVANCE */
    if (!is_nan_ray) c->st->advances++;
    uint32_t step_mask = 0u;
    if (tx_corner <= tc_max) { step_mask ^= 1u; pos.x -= scale_exp2; }
    if (ty_corner <= tc_max) { step_mask ^= 2u; pos.y -= scale_exp2; }
    if (tz_corner <= tc_max) { step_mask ^= 4u; pos.z -= scale_exp2; }
    t_min = tc_max;
    idx ^= step_mask;
    /* POP */
    if ((idx & step_mask) != 0) {
      if (!is_nan_ray) c->st->pops++;
      uint32_t differing_bits = 0;
      if (step_mask & 1u) differing_bits |= f2u(pos.x) ^ f2u(pos.x + scale_exp2);
      if (step_mask & 2u) differing_bits |= f2u(pos.y) ^ f2u(pos.y + scale_exp2);
      if (step_mask & 4u) differing_bits |= f2u(pos.z) ^ f2u(pos.z + scale_exp2);
      scale = find_msb(differing_bits);
      scale_exp2 = u2f(((uint32_t)scale - (uint32_t)MAX_SCALE + 127u) << 23u);
      if (scale >= 0 && scale <= MAX_SCALE) { /* pin P6: the reference reads out of bounds here; value is dead */
        if (!is_nan_ray) c->st->pop_by_scale[scale]++;
        if (!is_nan_ray && scale < MAX_SCALE && !((pushed >> scale) & 1u)) c->st->cold_pops++;
        parent = octstack[scale].node;
        t_max = octstack[scale].tmax;
      }
      uint32_t sh = (uint32_t)scale & 31u;
      uint32_t shx = f2u(pos.x) >> sh, shy = f2u(pos.y) >> sh, shz = f2u(pos.z) >> sh;
      pos.x = u2f(shx << sh);
      pos.y = u2f(shy << sh);
      pos.z = u2f(shz << sh);
      idx = (shx & 1u) | ((shy & 1u) << 1u) | ((shz & 1u) << 2u);
      h = 0.0f;
      child_descriptor = 0;
    }
  }

  if (is_nan_ray) {
    c->st->nan_rays++;
  } else {
    c->st->rays++;
    c->st->iterations += iter > MAX_RAYCAST_ITERATIONS ? MAX_RAYCAST_ITERATIONS : iter;
    c->st->alg_bytes += bytes;
    if (iter > c->st->max_iter) c->st->max_iter = iter;
  }

  /* pin P7: fields the reference leaves unwritten on its early returns behave as if
     assigned from the cast's own state */
  res->iter = iter;
  res->t = t_min;
  res->scale = scale_exp2;
  if (capped) return 0; /* :263-266 */
  if (!ok) {            /* :371-377 */
    res->debugColor = v3(0.01f * (float)iter, 0.01f * (float)iter, 0.01f * (float)iter);
    return 0;
  }

  vec3 norm = v3(0.0f, 0.0f, 0.0f);
  uint32_t rec;
  Node target = extract_child(c, parent.descriptor, (uint32_t)child_descriptor, child_shift, parent.leafMask,
                              &res->pointer, &rec);
  if (target.leafMask != 0) {
    int raw = target.leafMask;
    float normX = (float)((raw % 10) - 5);
    float normY = (float)((((raw % 100) - (raw % 10)) / 10) - 5);
    float normZ = (float)(((raw - (raw % 100)) / 100) - 5);
    norm = gnormalize(v3(normX, normY, normZ));
  }
  res->t = t_min;
  res->value = (uint32_t)target.value;
  res->leafMask = target.leafMask;
  res->iter = iter;
  res->normal = norm;
  res->scale = scale_exp2;
  res->depth = (uint32_t)(MAX_SCALE - scale);
  /* res.hitPos = origin + t_min * dir + norm * scale_exp2*2  (value never read) */
  res->hitPos = vadd(vadd(origin, vscale(dir, t_min)), vscale(vscale(norm, scale_exp2), 2.0f));

  vec3 vp = pos;
  if (dir.x > 0) vp.x = 3.0f - vp.x - scale_exp2;
  if (dir.y > 0) vp.y = 3.0f - vp.y - scale_exp2;
  if (dir.z > 0) vp.z = 3.0f - vp.z - scale_exp2;
  /* vp += norm * scale_exp2 * 2 * 1.74  -- left-associative, float throughout */
  vp.x += ((norm.x * scale_exp2) * 2.0f) * 1.74f;
  vp.y += ((norm.y * scale_exp2) * 2.0f) * 1.74f;
  vp.z += ((norm.z * scale_exp2) * 2.0f) * 1.74f;
  res->voxelPos = vp;
  res->debugColor = v3(0.005f * (float)iter, 0.005f * (float)iter, 0.005f * (float)iter);
  return scale < MAX_SCALE && t_min <= t_max;
}

/* ------------------------------------------------------------------ shading */

static void record_hit(svo_hit *hit, const castResult *res, int intersect) {
  if (!hit) return;
  if (intersect) {
    hit->pointer = res->pointer;
    hit->raw_normal = (uint16_t)res->leafMask;
    hit->value = (uint8_t)res->value;
    hit->level = (uint8_t)res->depth;
    hit->iter = res->iter;
    hit->t = res->t;
  } else {
    hit->pointer = 0;
    hit->raw_normal = 0;
    hit->value = 0;
    hit->level = 0;
    hit->iter = res->iter;
    hit->t = 0.0f;
  }
}

/* :435-646 */
static vec3 trace(ctx_t *c, const svo_oracle_params *prm, float beamDist, float t_start, vec3 origin, vec3 dir, float seed0,
                  float seed1, float seed2, float *depth, svo_hit *hit) {
  castResult res;
  memset(&res, 0, sizeof res);
  res.t = 2.0f;
  origin = vadd(origin, vscale(dir, beamDist));
  int intersect = 1;
  vec3 accum = v3(0, 0, 0), mask = v3(1, 1, 1), normal;
  int mode = prm->render_mode;
  if (mode == 0) {
    for (int i = 0; i < prm->bounces; i++) {
      int coneTrace = i != 0;
      intersect = intersect_octree(c, origin, dir, &res, MAX_DEPTH, coneTrace, i == 0 ? t_start : 0.0f);
      if (i == 0) record_hit(hit, &res, intersect);
      if (!intersect && i == 0) {
        vec3 sky = v3(0.6725f, 0.8784f, 1.0f);
        accum = vadd(accum, v3(sky.x - dir.y * 0.4f, sky.y - dir.y * 0.4f, sky.z - dir.y * 0.25f));
        break;
      }
      normal = res.normal;
      vec3 hitpoint = res.voxelPos;
      float r = glsl_rand(seed0 + glsl_rand_scaled(seed0, seed2, 0.1f), seed1 + glsl_rand_scaled(seed1, seed2, 0.02f));
      float rand1 = (2.0f * PI_F) * r;
      vec3 w = normal;
      vec3 axis = fabsf(w.x) > 0.1f ? v3(0, 1, 0) : v3(1, 0, 0);
      vec3 u = gnormalize(gcross(axis, w));
      vec3 v = gcross(w, u);
      vec3 newdir;
      if ((prm->mirror_mask >> (res.value & 31u)) & 1u) {
        /* dormant :503  newdir = dir - 2 * dot(dir, normal) * normal */
        float k = 2.0f * dot3(dir, normal);
        newdir = v3(dir.x - k * normal.x, dir.y - k * normal.y, dir.z - k * normal.z);
      } else {
        float cs = gcos(rand1), sn = gsin(rand1), om = 1.0f - r;
        newdir = gnormalize(vadd(vadd(vscale(u, cs), vscale(v, sn)), vscale(w, om)));
      }
      origin = hitpoint;
      dir = newdir;
      vec3 matcolor = v3(hitpoint.x - 1.0f, hitpoint.y - 1.0f, hitpoint.z - 1.0f);
      if (res.value == 1) matcolor = v3(0.84f, 0.86f, 0.78f);
      if (res.value == 2) matcolor = v3(0.57f, 0.5f, 0.31f);
      if (res.value == 3) matcolor = v3(0.37f, 0.43f, 0.27f);
      if (intersect) {
        *depth = res.t;
        accum = vadd(accum, vmul(mask, v3(0, 0, 0)));
        mask = vmul(mask, matcolor);
        mask = vscale(mask, dot3(newdir, normal));
      } else {
        vec3 sun_dir = gnormalize(v3(1.0f, 1.0f, 1.0f));
        float diff = gacos(dot3(dir, sun_dir));
        if (diff < 0.4f) accum = vadd(accum, vscale(mask, 7.0f));
        accum = vadd(accum, vscale(mask, 1.0f));
        *depth = 0.0f;
        break;
      }
    }
    return accum;
  } else if (mode == 1) {
    intersect = intersect_octree(c, origin, dir, &res, MAX_DEPTH, 0, t_start);
    record_hit(hit, &res, intersect);
    *depth = intersect ? res.t : 0.0f;
    return res.debugColor;
  } else if (mode == 2) {
    intersect = intersect_octree(c, origin, dir, &res, MAX_DEPTH, 0, t_start);
    record_hit(hit, &res, intersect);
    if (intersect) {
      *depth = res.t;
      /* pin P8: the shader leaves matcolor uninitialised for other values (svotrace.comp:577-586); llvmpipe resolves the
         undefined value to material 1's colour (tests/golden/fuzz_golden.npz: values 4 and 127) */
      vec3 matcolor = v3(0.84f, 0.86f, 0.78f);
      if (res.value == 1) matcolor = v3(0.84f, 0.86f, 0.78f);
      if (res.value == 2) matcolor = v3(0.57f, 0.5f, 0.31f);
      if (res.value == 3) matcolor = v3(0.37f, 0.43f, 0.27f);
      vec3 sun_dir = gnormalize(v3(0.5f, 0.5f, 0.5f));
      float ph = dot3(res.normal, sun_dir) * 0.1f;
      if (res.depth >= 10) {
        matcolor = v3(matcolor.x + ph, matcolor.y + ph, matcolor.z + ph);
      } else {
        float k = dot3(v3(0, 1.0f, 0), sun_dir) * 0.1f;
        matcolor = v3(matcolor.x + k, matcolor.y + k, matcolor.z + k);
      }
      float trueDist = res.t + beamDist;
      /* exp(-0.5*d*K) = exp2(d * (-0.5*K*log2e)), SURVEY Appendix C */
      float lambdag = gexp2(trueDist * (-0.5f * 2.0f * 1.44269504f));
      float lambdab = gexp2(trueDist * (-0.5f * 4.0f * 1.44269504f));
      float lambdar = gexp2(trueDist * (-0.5f * 1.0f * 1.44269504f));
      matcolor.x = lambdar * matcolor.x + (1.0f - lambdar) * 1.0f;
      matcolor.y = lambdag * matcolor.y + (1.0f - lambdag) * 1.0f;
      matcolor.z = lambdab * matcolor.z + (1.0f - lambdab) * 1.0f;
      /* shadow ray; `res` is reused exactly as in the reference (:607) */
      vec3 so = res.voxelPos;
      int sh = intersect_octree(c, so, sun_dir, &res, MAX_DEPTH, 0, 0.0f);
      if (sh && res.t > res.scale * SQRT3_F) {
        matcolor = v3(matcolor.x - 0.2f, matcolor.y - 0.2f, matcolor.z - 0.2f);
      } else if (res.iter > 260) {
        float pen = (0.05f * (float)res.iter) / 100.0f;
        matcolor = v3(matcolor.x - pen, matcolor.y - pen, matcolor.z - pen);
      }
      return matcolor;
    } else {
      *depth = 0.0f;
      return v3(0.6725f - dir.y * 0.4f, 0.8784f - dir.y * 0.4f, 1.0f - dir.y * 0.25f);
    }
  } else if (mode == 3) {
    intersect = intersect_octree(c, origin, dir, &res, MAX_DEPTH, 0, t_start);
    record_hit(hit, &res, intersect);
    if (intersect) {
      *depth = res.t;
      return v3(res.normal.x * 0.5f + 0.5f, res.normal.y * 0.5f + 0.5f, res.normal.z * 0.5f + 0.5f);
    }
    *depth = 0.0f;
    return v3(0, 0, 0);
  }
  /* mode 4 (:643-645) returns the unset res.voxelPos; modes >= 5 fall off the end: zero (pin P8) */
  if (hit) memset(hit, 0, sizeof *hit);
  return v3(0, 0, 0);
}

/* imageStore to rgba8 (pin P9): clamp, round-half-even; NaN -> 255 */
static uint8_t unorm8(float x) {
  if (isnan(x)) return 255;
  if (x <= 0.0f) return 0;
  if (x >= 1.0f) return 255;
  return (uint8_t)rintf(x * 255.0f);
}

/*
 * Render pixels (x, y) with y in [y0, y1), x in [0, width), taking every xstep-th /
 * ystep-th pixel (others are left untouched).  Output layouts match the reference's
 * GL images: rgba8 row-major, row 0 = p.y = 0 (svotrace.comp:662-664, 726-727).
 * Any of rgba / depth / hits may be NULL.
 */
int svo_oracle_render_beam(const uint8_t *pool, uint64_t pool_len, const svo_oracle_params *prm, int y0, int y1, int xstep,
                           int ystep, uint8_t *rgba, float *depth_out, svo_hit *hits, svo_oracle_stats *stats,
                           const float *beam) {
  svo_oracle_stats local;
  memset(&local, 0, sizeof local);
  ctx_t c = {pool, pool_len, stats ? stats : &local, 0};
  if (stats) memset(stats, 0, sizeof *stats);
  if (pool_len < NODE_SIZE || prm->width <= 0 || prm->height <= 0) return 1;
  if (xstep < 1) xstep = 1;
  if (ystep < 1) ystep = 1;
  int W = prm->width, Hh = prm->height;
  vec3 camPos = v3(prm->cam[0], prm->cam[1], prm->cam[2]);
  vec3 l1 = v3(prm->cam[3], prm->cam[4], prm->cam[5]), l2 = v3(prm->cam[6], prm->cam[7], prm->cam[8]);
  vec3 r1 = v3(prm->cam[9], prm->cam[10], prm->cam[11]), r2 = v3(prm->cam[12], prm->cam[13], prm->cam[14]);
  uint32_t dword0 = (uint32_t)get_byte(&c, 0) | ((uint32_t)get_byte(&c, 1) << 8) | ((uint32_t)get_byte(&c, 2) << 16) |
                    ((uint32_t)get_byte(&c, 3) << 24);
  int spp = prm->spp < 1 ? 1 : prm->spp;
  for (int py = y0; py < y1 && py < Hh; py += ystep) {
    for (int px = 0; px < W; px += xstep) {
      float p_x = ((float)px + 0.5f) / (float)W;
      float p_y = ((float)py + 0.5f) / (float)Hh;
      vec3 a = v3(gmix(l1.x, l2.x, p_y), gmix(l1.y, l2.y, p_y), gmix(l1.z, l2.z, p_y));
      vec3 b = v3(gmix(r1.x, r2.x, p_y), gmix(r1.y, r2.y, p_y), gmix(r1.z, r2.z, p_y));
      vec3 dir = v3(gmix(a.x, b.x, p_x), gmix(a.y, b.y, p_x), gmix(a.z, b.z, p_x));
      vec3 normdir = gnormalize(dir);
      /* `float depth = -1.0f` at :672 is lost: a mode-0 primary miss stores 0.0 (pin P7) */
      float depth = 0.0f;
      vec3 fin = v3(0, 0, 0);
      svo_hit hit;
      memset(&hit, 0, sizeof hit);
      for (int s = 0; s < spp; s++) {
        float d_s = 0.0f;
        const float t_start = beam ? beam[(size_t)(py / BEAM_BLOCK) * (size_t)((W + BEAM_BLOCK - 1) / BEAM_BLOCK) + (size_t)(px / BEAM_BLOCK)] : 0.0f;
        vec3 col = trace(&c, prm, 0.0f, t_start, camPos, normdir, (float)px, (float)py, (float)(prm->frame_number + s), &d_s,
                         s == 0 ? &hit : NULL);
        if (s == 0) depth = d_s;
        fin = vadd(fin, col);
      }
      if (spp > 1) fin = vscale(fin, 1.0f / (float)spp);
      vec3 debugColor = v3(1, 1, 1);
      if (dword0 == 0) debugColor = v3(1.0f, 0.0f, 0.0f);
      if (px < 10 && py < 10) fin = debugColor;
      size_t o = (size_t)py * (size_t)W + (size_t)px;
      if (prm->progressive && rgba && prm->frame_number > 1) { /* :712-719, pinned by tests/golden/accum_golden.npz */
        /* imageLoad of an rgba8 image: unorm8 -> float */
        vec3 last = v3(UNORM8_TO_FLOAT(rgba[o * 4 + 0]), UNORM8_TO_FLOAT(rgba[o * 4 + 1]), UNORM8_TO_FLOAT(rgba[o * 4 + 2]));
        if (prm->frame_number < MAX_FRAME_ITER) {
          const float fn = (float)prm->frame_number, fd = (float)(prm->frame_number + 1);
          fin = v3(PROG_DIV(fn * last.x + fin.x, fd), PROG_DIV(fn * last.y + fin.y, fd), PROG_DIV(fn * last.z + fin.z, fd));
        } else {
          fin = last;
        }
      }
      if (rgba) {
        rgba[o * 4 + 0] = unorm8(fin.x);
        rgba[o * 4 + 1] = unorm8(fin.y);
        rgba[o * 4 + 2] = unorm8(fin.z);
        rgba[o * 4 + 3] = 255;
      }
      if (depth_out) depth_out[o] = depth;
      if (hits) hits[o] = hit;
      c.st->pixels++;
    }
  }
  return 0;
}

int svo_oracle_render(const uint8_t *pool, uint64_t pool_len, const svo_oracle_params *prm, int y0, int y1, int xstep,
                      int ystep, uint8_t *rgba, float *depth_out, svo_hit *hits, svo_oracle_stats *stats) {
  return svo_oracle_render_beam(pool, pool_len, prm, y0, y1, xstep, ystep, rgba, depth_out, hits, stats, NULL);
}

/* The same frame on `nthreads` host threads (OpenMP; rows are independent, every thread runs the single-threaded function
 * above on its rows with its own counters, summed at the end): the all-core CPU figure bench.py prints next to the
 * single-thread baseline.  Identical bytes. */
int svo_oracle_render_mt(const uint8_t *pool, uint64_t pool_len, const svo_oracle_params *prm, int y0, int y1, int xstep,
                         int ystep, uint8_t *rgba, float *depth_out, svo_hit *hits, svo_oracle_stats *stats, int nthreads) {
  if (pool_len < NODE_SIZE || prm->width <= 0 || prm->height <= 0) return 1;
  if (ystep < 1) ystep = 1;
  if (y1 > prm->height) y1 = prm->height;
  const int nrows = y1 > y0 ? (y1 - y0 + ystep - 1) / ystep : 0;
  svo_oracle_stats total;
  memset(&total, 0, sizeof total);
  int rc = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
  {
    svo_oracle_stats mine, row;
    memset(&mine, 0, sizeof mine);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int i = 0; i < nrows; i++) {
      const int py = y0 + i * ystep;
      if (svo_oracle_render_beam(pool, pool_len, prm, py, py + 1, xstep, 1, rgba, depth_out, hits, &row, NULL)) continue;
      mine.pixels += row.pixels; mine.rays += row.rays; mine.nan_rays += row.nan_rays; mine.iterations += row.iterations;
      mine.alg_bytes += row.alg_bytes; mine.descends += row.descends; mine.advances += row.advances; mine.pops += row.pops;
      mine.cold_pops += row.cold_pops;
      if (row.max_iter > mine.max_iter) mine.max_iter = row.max_iter;
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      total.pixels += mine.pixels; total.rays += mine.rays; total.nan_rays += mine.nan_rays; total.iterations += mine.iterations;
      total.alg_bytes += mine.alg_bytes; total.descends += mine.descends; total.advances += mine.advances; total.pops += mine.pops;
      total.cold_pops += mine.cold_pops;
      if (mine.max_iter > total.max_iter) total.max_iter = mine.max_iter;
    }
  }
  if (stats) *stats = total;
  return rc;
}

/* ------------------------------------------------------------------ beam pre-pass (use_beam)
 *
 * NOT a restatement of the reference: its beam pass (svobeam.comp:617-637, Main.java:257-266, svotrace.comp:438,
 * 656-658) is dormant and inconsistent -- one full-depth ray through the corner pixel of every 4x4 block with an
 * un-normalised direction, whose t then moves the origin of rays with normalised directions.  This is the CPU
 * statement of the replacement the HIP library implements (csrc/svo_beam.hip.h), kept here as its checker:
 *
 *   for every 4x4 pixel block (beamSquareSize, Main.java:41) the pyramid of its rays -- the pixel footprints
 *   widened by half a pixel -- is intersected with the octree itself: depth-first over the non-empty child
 *   cubes that are not entirely outside one of the pyramid's four side planes, pruned by distance; a cube ends the
 *   descent when it is a leaf, at MAX_DEPTH, or no larger than the pyramid is wide at its distance.  The block's
 *   value is the smallest distance from the camera to such a cube, scaled by 1 - 2^-10: no ray of the block can
 *   meet a non-empty voxel before it (rays have unit directions, so t is the distance travelled).
 *
 * Parity statement for use_beam = 1: hit pointer / value / normal / level / t and the colour and depth images equal
 * those of use_beam = 0 (modes 0, 2, 3); iteration counts drop (mode 1 shows them, so its colours change).
 */
typedef struct {
  uint32_t base;   /* offset of the first child record */
  int mask;        /* child tags */
  float x, y, z;   /* cube origin */
  int next;        /* next child slot to look at */
  uint32_t path;   /* the cube's octant path from the root, 3 bits per level (x | y << 1 | z << 2) */
} beam_frame;

/* The top BEAM_TOP levels of a pool may carry no emptiness information: the reference's world builder flags every node of
 * its all-interior levels (fillEmptyChildren, the chunk nodes, the 512^3 task heads; Octree.java:317-343, 481-502) with
 * value 1 whether anything lies below or not, and the coarse walk would have to open each of them.  So the walk first
 * marks, for every node of those levels by its octant path, whether a cast could end in it or below it:
 *   H(node) = value != 0 and (the cast treats it as a leaf  or  H of one of its children),
 * children BEAM_TOP + 1 levels down counting by their value alone; the walk then skips cubes of those levels with H = 0.
 * (A cast cannot end inside such a cube either: it only descends through non-empty nodes.) */
#define BEAM_TOP 4
static const int beam_live_off[BEAM_TOP + 2] = {0, 0, 8, 72, 584, 4680};   /* level d starts at (8^d - 8) / 7 */

static int beam_mark(const ctx_t *c, uint32_t base, int mask, int depth, uint32_t path, uint8_t *live) {
  int any = 0;
  uint32_t ptr = base;
  for (int n = 0; n < 8; n++) {
    const int tag = (mask >> (2 * n)) & 3;
    const int value = get_byte(c, ptr);
    int h;
    if (depth == BEAM_TOP + 1) {
      h = value != 0;
    } else {
      h = 0;
      if (value != 0) {
        if (tag != 0) h = 1;
        else {
          const Node ch = extract_node(c, ptr);
          h = ch.cp == 0 ? 1 : beam_mark(c, ptr + (uint32_t)ch.cp, ch.leafMask, depth + 1, path * 8u + (uint32_t)n, live);
        }
      }
      live[beam_live_off[depth] + path * 8u + (uint32_t)n] = (uint8_t)h;
    }
    any |= h;
    ptr += tag == 1 ? LEAF_SIZE : (tag == 3 ? NON_SURFACE_LEAF_SIZE : NODE_SIZE);
  }
  return any;
}

static inline vec3 beam_dir(const svo_oracle_params *prm, float u, float v) {
  const float *c = prm->cam;
  vec3 a = v3(gmix(c[3], c[6], v), gmix(c[4], c[7], v), gmix(c[5], c[8], v));
  vec3 b = v3(gmix(c[9], c[12], v), gmix(c[10], c[13], v), gmix(c[11], c[14], v));
  return v3(gmix(a.x, b.x, u), gmix(a.y, b.y, u), gmix(a.z, b.z, u));
}

/* 1 if the camera is a finite planar image rectangle (r2 = r1 + l2 - l1 up to rounding): the pyramid of a block is
   then bounded by the planes through its corner rays */
static int beam_camera_ok(const svo_oracle_params *prm) {
  const float *c = prm->cam;
  for (int i = 0; i < 15; i++) if (!(fabsf(c[i]) < 1.0e30f)) return 0;
  float ex = c[12] - (c[9] + (c[6] - c[3])), ey = c[13] - (c[10] + (c[7] - c[4])), ez = c[14] - (c[11] + (c[8] - c[5]));
  float dx = c[12] - c[3], dy = c[13] - c[4], dz = c[14] - c[5];
  float e2 = ex * ex + (ey * ey + ez * ez), d2 = dx * dx + (dy * dy + dz * dz);
  return d2 > 0.0f && e2 <= 1.0e-8f * d2;
}

/* `visits` (may be NULL): number of child cubes looked at, summed over the blocks (diagnostic) */
int svo_oracle_beam(const uint8_t *pool, uint64_t pool_len, const svo_oracle_params *prm, float *tbeam, uint64_t *visits) {
  uint64_t nvisit = 0;
  svo_oracle_stats dummy;
  memset(&dummy, 0, sizeof dummy);
  ctx_t c = {pool, pool_len, &dummy, 0};
  if (pool_len < NODE_SIZE || prm->width <= 0 || prm->height <= 0) return 1;
  const int W = prm->width, Hh = prm->height;
  const int bw = (W + BEAM_BLOCK - 1) / BEAM_BLOCK, bh = (Hh + BEAM_BLOCK - 1) / BEAM_BLOCK;
  const int cam_ok = beam_camera_ok(prm);
  const vec3 o = v3(prm->cam[0], prm->cam[1], prm->cam[2]);
  const Node root = extract_node(&c, 0);
  static uint8_t live[4680];
  memset(live, 0, sizeof live);
  beam_mark(&c, (uint32_t)root.cp, root.leafMask, 1, 0u, live);
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      float *out = &tbeam[(size_t)by * bw + bx];
      *out = 0.0f;
      if (!cam_ok) continue;
      const float u0 = ((float)(bx * BEAM_BLOCK) - 0.5f) / (float)W, u1 = ((float)(bx * BEAM_BLOCK + BEAM_BLOCK) + 0.5f) / (float)W;
      const float v0 = ((float)(by * BEAM_BLOCK) - 0.5f) / (float)Hh, v1 = ((float)(by * BEAM_BLOCK + BEAM_BLOCK) + 0.5f) / (float)Hh;
      const vec3 d00 = beam_dir(prm, u0, v0), d10 = beam_dir(prm, u1, v0), d01 = beam_dir(prm, u0, v1), d11 = beam_dir(prm, u1, v1);
      const vec3 dc = beam_dir(prm, 0.5f * (u0 + u1), 0.5f * (v0 + v1));
      vec3 n[4] = {gcross(d00, d01), gcross(d11, d10), gcross(d10, d00), gcross(d01, d11)};
      int planes_ok = 1;
      for (int k = 0; k < 4; k++) {
        const float s = dot3(n[k], dc);
        if (s < 0.0f) n[k] = v3(-n[k].x, -n[k].y, -n[k].z);
        else if (!(s > 0.0f)) planes_ok = 0;
      }
      if (!planes_ok) continue;
      /* width of the pyramid per unit distance, squared (the wider of its two image-plane edges over the centre ray) */
      const vec3 eu = v3(d10.x - d00.x, d10.y - d00.y, d10.z - d00.z), ev = v3(d01.x - d00.x, d01.y - d00.y, d01.z - d00.z);
      const float spread2 = gmax(dot3(eu, eu), dot3(ev, ev)) / dot3(dc, dc);
      float best2 = INFINITY;
      {   /* like the trace pass, the walk starts at the root's children whatever the root record's value is */
        beam_frame st[MAX_DEPTH + 1];
        int sp = 0;
        st[0].base = (uint32_t)root.cp; st[0].mask = root.leafMask; st[0].x = 1.0f; st[0].y = 1.0f; st[0].z = 1.0f; st[0].next = 0;
        st[0].path = 0u;
        while (sp >= 0) {
          beam_frame *f = &st[sp];
          if (f->next == 8) { sp--; continue; }
          const float size = u2f((uint32_t)(127 - (sp + 1)) << 23);   /* edge of a child cube at depth sp + 1 */
          /* nearest octant first (the minimum does not depend on the order; the pruning does): slot k looks at child
             k ^ near, near = the octant of this cube that holds / faces the camera */
          const int near = (o.x >= f->x + size ? 1 : 0) | (o.y >= f->y + size ? 2 : 0) | (o.z >= f->z + size ? 4 : 0);
          const int nch = (f->next++) ^ near;
          nvisit++;
          uint32_t ptr = f->base;                                      /* walk to child nch (svotrace.comp:135-145) */
          for (int i = 0; i < nch; i++) {
            const int tg = (f->mask >> (2 * i)) & 3;
            ptr += tg == 1 ? LEAF_SIZE : (tg == 3 ? NON_SURFACE_LEAF_SIZE : NODE_SIZE);
          }
          const int tag = (f->mask >> (2 * nch)) & 3;
          const uint32_t cpath = f->path * 8u + (uint32_t)nch;
          if (sp + 1 <= BEAM_TOP ? !live[beam_live_off[sp + 1] + cpath] : get_byte(&c, ptr) == 0) continue;   /* nothing to meet in it */
          const float lx = f->x + (float)(nch & 1) * size, ly = f->y + (float)((nch >> 1) & 1) * size,
                      lz = f->z + (float)((nch >> 2) & 1) * size;
          const float half = 0.5f * size;
          const float mx = (lx + half) - o.x, my = (ly + half) - o.y, mz = (lz + half) - o.z;   /* cube centre, camera-relative */
          int outside = 0;
          for (int k = 0; k < 4 && !outside; k++) {
            const float far = dot3(n[k], v3(mx, my, mz)) + (fabsf(n[k].x) + (fabsf(n[k].y) + fabsf(n[k].z))) * half;
            if (far < 0.0f) outside = 1;                                /* every corner is behind this side plane */
          }
          if (outside) continue;
          const float ddx = gmax(gmax(lx - o.x, o.x - (lx + size)), 0.0f), ddy = gmax(gmax(ly - o.y, o.y - (ly + size)), 0.0f),
                      ddz = gmax(gmax(lz - o.z, o.z - (lz + size)), 0.0f);
          const float dist2 = ddx * ddx + (ddy * ddy + ddz * ddz);
          if (!(dist2 < best2)) continue;                               /* cannot improve the bound */
          int cp = 0, cmask = 0;
          if (tag == 0) {
            const Node ch = extract_node(&c, ptr);
            cp = ch.cp; cmask = ch.leafMask;
          }
          const int terminal = tag != 0 || cp == 0 || sp + 1 >= MAX_DEPTH || size * size <= dist2 * spread2;
          if (terminal) { best2 = dist2; continue; }
          sp++;
          st[sp].base = ptr + (uint32_t)cp; st[sp].mask = cmask; st[sp].x = lx; st[sp].y = ly; st[sp].z = lz; st[sp].next = 0;
          st[sp].path = cpath;
        }
      }
      *out = sqrtf(best2) * 0.9990234375f;
    }
  if (visits) *visits = nvisit;
  return 0;
}

/* exported for tests of the pinned transcendentals */
float svo_oracle_sin(float x) { return gsin(x); }
float svo_oracle_cos(float x) { return gcos(x); }
float svo_oracle_acos(float x) { return gacos(x); }
float svo_oracle_exp2(float x) { return gexp2(x); }
float svo_oracle_rand(float a, float b) { return glsl_rand(a, b); }
