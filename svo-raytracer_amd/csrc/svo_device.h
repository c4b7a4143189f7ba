// svo_device.h -- gfx950 device-side building blocks of the SVO hot path.
//
// What the reference computes here: src/shaders/svotrace.comp
//   node decode           :75-157      traversal (Laine-Karras stack walk)  :211-432
//   rand / shading        :26-29, :435-646
// How it is computed here is NOT the shader's way:
//   * a child record is fetched with ONE unaligned 8-byte load (the shader issues one
//     dependent dword load per byte, up to 7 per record);
//   * the child's byte offset inside its sibling block is a closed form on the parent's
//     tag mask (two popcounts) instead of a data-dependent loop;
//   * the traversal stack lives in LDS, laid out [level][lane] so that a wave's 64
//     lanes hit 64 distinct banks whatever level each lane is on; an entry is
//     {child-block base, tag mask, t_max} = 12 B instead of the shader's 20 B
//     {Node, t_max} in scratch memory; only the 12 levels that can ever be pushed
//     (scale 11..22) exist;
//   * a per-lane "written" bitmask replaces clearing the stack between rays.
// Float semantics are pinned to the reference-under-llvmpipe run (parity):
// build with -ffp-contract=off; every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svo {

constexpr int kMaxScale = 23;
constexpr int kMaxDepth = 13;
constexpr uint32_t kMaxIter = 1500u;
constexpr int kStackLevels = 12;   // scale 11..22 (a PUSH needs 23 - scale < maxDepth <= 13)
constexpr int kStackBase = 11;
constexpr float kEpsilon = 3.552713678800501e-15f;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ float fmin_g(float a, float b) { return __builtin_fminf(a, b); }  // NaN-ignoring, like GLSL min on llvmpipe
__device__ __forceinline__ float fmax_g(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float sign_g(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ V3 normalize3(V3 v) {
  float s = dot3(v, v);
  float r = 1.0f / __builtin_sqrtf(s);
  return mk(v.x * r, v.y * r, v.z * r);
}
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float mix_g(float x, float y, float t) { return x + t * (y - x); }
__device__ __forceinline__ bool all_nan(V3 v) { return (v.x != v.x) && (v.y != v.y) && (v.z != v.z); }

// ---- transcendental functions as the reference's GLSL implementation evaluates them -----------
// range reduction by multiples of pi/4 in three fused steps, then a degree-3/4 minimax
// polynomial in r^2; quadrant logic differs between sin and cos.
template <bool kCos>
__device__ __forceinline__ float sincos_pinned(float x) {
  const float ax = __builtin_fabsf(x);
  const float yf = ax * 1.27323954473516f;
  // the conversion is x86's cvttps2dq (llvmpipe's host): out of range (|x| >= 1.69e9) and NaN give 0x80000000, where
  // v_cvt_i32_f32 would saturate; reachable through frameNumber (svotrace.comp:486 multiplies it by 7.8)
  uint32_t ju = (yf >= 2147483648.0f || yf != yf) ? 0x80000000u : (uint32_t)(int)yf;
  ju = (ju + 1u) & ~1u;
  const int j = (int)ju;
  const float y = (float)j;
  float r = __builtin_fmaf(y, -0.78515625f, ax);
  r = __builtin_fmaf(y, -2.4187564849853515625e-4f, r);
  r = __builtin_fmaf(y, -3.77489497744594108e-8f, r);
  const int q = kCos ? j - 2 : j;
  const bool cos_poly = (q & 2) != 0;
  bool neg;
  if (kCos) neg = ((~q) & 4) != 0;
  else neg = ((q & 4) != 0) != (__builtin_signbit(x) != 0);
  const float z = r * r;
  float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  pc = pc * z;
  pc = pc * z;
  pc = __builtin_fmaf(z, -0.5f, pc);
  pc = pc + 1.0f;
  float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  ps = ps * z;
  ps = __builtin_fmaf(ps, r, r);
  float v = cos_poly ? pc : ps;
  v = neg ? -v : v;
  // llvmpipe clamps the result to [-1, 1]: visible once the reduced argument stops being small (|x| > ~6e7, i.e.
  // frameNumber beyond 2^23; tools/probes/sin_probe.comp)
  v = v > 1.0f ? 1.0f : v;
  v = v < -1.0f ? -1.0f : v;
  return v;
}
__device__ __forceinline__ float acos_pinned(float x) {
  const float ax = __builtin_fabsf(x);
  float t = ax * (-0.02363318f) + 0.08132463f;
  t = ax * t + (0.785398163f - 1.0f);   // (float)(pi/4) - 1 = -0.21460181 (mistyped as -0.2145988 in round 1; tools/probes/fn_probe.comp)
  t = ax * t + 1.5707964f;
  return 1.5707964f - sign_g(x) * (1.5707964f - __builtin_sqrtf(1.0f - ax) * t);
}
__device__ __forceinline__ float exp2_pinned(float y) {
  if (y != y) return y;   // NaN in, NaN out
  y = fmin_g(y, 128.0f);
  y = fmax_g(y, -126.99999f);
  const float ip = __builtin_floorf(y);
  const float fp = y - ip;
  const float t2 = fp * fp;
  float e = __builtin_fmaf(t2, 0.00898934009049466391101f, 0.240153617044375388211f);
  e = __builtin_fmaf(t2, e, 1.0f);
  float o = __builtin_fmaf(t2, 0.00187757667519147912699f, 0.0558263180532956664775f);
  o = __builtin_fmaf(t2, o, 0.693153073200168932794f);
  // 2^ip is built in the exponent field: ip = -127 gives 0.0 (not the denormal 2^-127), ip = 128 +inf
  return __uint_as_float((uint32_t)((int)ip + 127) << 23) * __builtin_fmaf(o, fp, e);
}
// svotrace.comp:26-29 on the already-formed dot product
__device__ __forceinline__ float rand_of_dot(float d) {
  const float v = sincos_pinned<false>(d) * 43758.5453f;
  return v - __builtin_floorf(v);
}
// The per-pixel random number of svotrace.comp:486.  seed2 * 0.1 * 78.233 and
// seed2 * 0.02 * 78.233 are evaluated with the two constants folded first (that is what
// the reference's compiler does; visible from frameNumber 7 on).
__device__ __forceinline__ float pixel_rand(float seed0, float seed1, float seed2) {
  const float k0 = 0.1f * 78.233f, k1 = 0.02f * 78.233f;
  const float ra = rand_of_dot(seed0 * 12.9898f + seed2 * k0);
  const float rb = rand_of_dot(seed1 * 12.9898f + seed2 * k1);
  return rand_of_dot((seed0 + ra) * 12.9898f + (seed1 + rb) * 78.233f);
}

// ---- pool access -----------------------------------------------------------------------------
struct Pool {
  const uint8_t *base;   // device pointer, zero-padded by >= 16 bytes past `len`
  uint32_t len;          // meaningful bytes (memOffset)
};

// 8 bytes at byte offset p (little-endian view); bytes at or beyond len read as 0.
__device__ __forceinline__ uint64_t load_record(const Pool &pool, uint32_t p) {
  uint64_t v;
  const bool inside = p < pool.len;
  __builtin_memcpy(&v, pool.base + (inside ? p : 0u), 8);
  return inside ? v : 0ull;
}
__device__ __forceinline__ uint32_t rec_value(uint64_t r) { return (uint32_t)r & 0xffu; }
// child pointer: int32 big-endian in bytes 1..4 (Octree.java:162-164)
__device__ __forceinline__ uint32_t rec_cp(uint64_t r) { return __builtin_bswap32((uint32_t)(r >> 8)); }
// tag mask: u16 big-endian in bytes 5..6 (Octree.java:170-172)
__device__ __forceinline__ uint32_t rec_mask_be(uint64_t r) {
  return (((uint32_t)(r >> 40) & 0xffu) << 8) | ((uint32_t)(r >> 48) & 0xffu);
}
// packed normal of a surface leaf: u16 little-endian in bytes 1..2 (Octree.java:150-151)
__device__ __forceinline__ uint32_t rec_normal_le(uint64_t r) { return (uint32_t)(r >> 8) & 0xffffu; }

// byte offset of child c inside the sibling block of a parent with tag mask m:
// record sizes are 7/3/7/1 for tags 0/1/2/3 = 7 - 4*lo - 2*(lo & hi)
__device__ __forceinline__ uint32_t child_offset(uint32_t m, uint32_t c) {
  const uint32_t below = (1u << (2u * c)) - 1u;
  const uint32_t lo = m & 0x5555u & below;
  const uint32_t both = lo & (m >> 1);
  return 7u * c - 4u * (uint32_t)__builtin_popcount(lo) - 2u * (uint32_t)__builtin_popcount(both);
}
__device__ __forceinline__ uint32_t tag_size(uint32_t tag) { return tag == 1u ? 3u : (tag == 3u ? 1u : 7u); }

// ---- traversal ---------------------------------------------------------------------------------
struct Cast {
  bool hit;
  bool capped;        // left through the iteration cap (svotrace.comp:263-266)
  uint32_t pointer;   // byte offset of the hit node
  uint32_t value;
  uint32_t raw;       // leafMask field of the hit node
  uint32_t iter;
  uint32_t level;     // kMaxScale - scale
  float t;            // t_min at exit
  float scale_exp2;
  V3 normal;
  V3 voxel_pos;
};

struct Counters {
  uint32_t rays, nan_rays, iters, bytes, max_iter;
};

// LDS stack of ONE wave: [level][lane].  pm = {child-block base, t_max bits}, mk = tag mask.
struct WaveStack {
  uint2 pm[kStackLevels * 64];
  uint16_t mk[kStackLevels * 64];
};

// t_start: 0 for every ray of the reference; the primary ray of a pixel starts at its block's beam distance when the
// beam pre-pass is on (svo_beam.hip.h) -- same origin and coefficients, only t_min is raised
template <bool kCount>
__device__ __forceinline__ Cast cast_ray(const Pool &pool, WaveStack &stk, const uint32_t lane, V3 o, V3 d,
                                         int max_depth, const bool cone, Counters &cnt, const float t_start = 0.0f) {
  Cast res;
  res.hit = false; res.capped = false; res.pointer = 0; res.value = 0; res.raw = 0; res.level = 0;
  res.normal = mk(0.f, 0.f, 0.f); res.voxel_pos = mk(0.f, 0.f, 0.f);

  // A ray whose origin or direction is NaN on all three axes can never advance: the
  // reference spins to the iteration cap and returns a miss with iter = 1501 (quirk Q7).
  if (all_nan(o) || all_nan(d)) {
    res.capped = true; res.iter = kMaxIter + 1u; res.t = 0.0f; res.scale_exp2 = 0.5f;
    if (kCount) cnt.nan_rays++;
    return res;
  }

  if (__builtin_fabsf(d.x) < kEpsilon) d.x = kEpsilon * sign_g(d.x);
  if (__builtin_fabsf(d.y) < kEpsilon) d.y = kEpsilon * sign_g(d.y);
  if (__builtin_fabsf(d.z) < kEpsilon) d.z = kEpsilon * sign_g(d.z);
  const float cx = 1.0f / -__builtin_fabsf(d.x);
  const float cy = 1.0f / -__builtin_fabsf(d.y);
  const float cz = 1.0f / -__builtin_fabsf(d.z);
  float bx = cx * o.x, by = cy * o.y, bz = cz * o.z;
  uint32_t octant = 0;
  if (d.x > 0.0f) { octant ^= 1u; bx = 3.0f * cx - bx; }
  if (d.y > 0.0f) { octant ^= 2u; by = 3.0f * cy - by; }
  if (d.z > 0.0f) { octant ^= 4u; bz = 3.0f * cz - bz; }
  float t_min = fmax_g(fmax_g(2.0f * cx - bx, 2.0f * cy - by), 2.0f * cz - bz);
  float t_max = fmin_g(fmin_g(cx - bx, cy - by), cz - bz);
  t_min = fmax_g(t_min, 0.0f);
  t_min = fmax_g(t_min, t_start);
  float h = t_max;

  uint32_t idx = 0;
  float px = 1.0f, py = 1.0f, pz = 1.0f;
  int scale = kMaxScale - 1;
  float sexp = 0.5f;
  if (1.5f * cx - bx > t_min) { idx ^= 1u; px = 1.5f; }
  if (1.5f * cy - by > t_min) { idx ^= 2u; py = 1.5f; }
  if (1.5f * cz - bz > t_min) { idx ^= 4u; pz = 1.5f; }

  // root record (svotrace.comp:222)
  const uint64_t root = load_record(pool, 0u);
  uint32_t pbase = rec_cp(root);          // child-block base = node offset (0) + cp
  uint32_t pmask = rec_mask_be(root);
  uint32_t written = 0;                    // which stack levels this ray has pushed
  uint32_t iter = 0, bytes = 7;
  uint32_t cptr = 0, tag = 0;
  uint64_t rec = 0;
  bool hit = false, capped = false;

  while (scale < kMaxScale) {
    iter++;
    if (iter > kMaxIter) { capped = true; break; }
    if (cone && t_min > 0.05f) max_depth = 11;

    const float tcx = px * cx - bx;
    const float tcy = py * cy - by;
    const float tcz = pz * cz - bz;
    const float tc_max = fmin_g(fmin_g(tcx, tcy), tcz);

    const uint32_t cs = idx ^ octant;
    tag = (pmask >> (2u * cs)) & 3u;
    cptr = pbase + child_offset(pmask, cs);
    rec = load_record(pool, cptr);
    if (kCount) bytes += tag_size(tag);

    if (rec_value(rec) != 0u && t_min <= t_max) {
      if (kMaxScale - scale == max_depth) { hit = true; break; }
      const float tv_max = fmin_g(t_max, tc_max);
      const float half = sexp * 0.5f;
      const float tmx = half * cx + tcx;
      const float tmy = half * cy + tcy;
      const float tmz = half * cz + tcz;
      if (t_min <= tv_max) {
        const uint32_t ccp = tag == 0u ? rec_cp(rec) : 0u;
        if (ccp == 0u) { hit = true; break; }
        if (tc_max < h) {  // PUSH.  scale is 11..22 for pools of up to 13 levels (the supported depth, = the
          // reference's MAX_DEPTH); the clamp only keeps LDS accesses in range on deeper pools, and is the same in
          // all three pipelines (trav_step, trav_loop) so that they keep producing identical bytes there too
          const uint32_t lvu = (uint32_t)(scale - kStackBase);
          const uint32_t lv = lvu < (uint32_t)(kStackLevels - 1) ? lvu : (uint32_t)(kStackLevels - 1);
          stk.pm[lv * 64 + lane] = make_uint2(pbase, __float_as_uint(t_max));
          stk.mk[lv * 64 + lane] = (uint16_t)pmask;
          written |= 1u << lv;
        }
        h = tc_max;
        pbase = cptr + ccp;
        pmask = rec_mask_be(rec);
        idx = 0u;
        --scale;
        sexp = half;
        if (tmx > t_min) { idx ^= 1u; px += sexp; }
        if (tmy > t_min) { idx ^= 2u; py += sexp; }
        if (tmz > t_min) { idx ^= 4u; pz += sexp; }
        t_max = tv_max;
        continue;
      }
    }
    // ADVANCE
    uint32_t step = 0u;
    if (tcx <= tc_max) { step ^= 1u; px -= sexp; }
    if (tcy <= tc_max) { step ^= 2u; py -= sexp; }
    if (tcz <= tc_max) { step ^= 4u; pz -= sexp; }
    t_min = tc_max;
    idx ^= step;
    // POP
    if ((idx & step) != 0u) {
      uint32_t diff = 0u;
      if (step & 1u) diff |= __float_as_uint(px) ^ __float_as_uint(px + sexp);
      if (step & 2u) diff |= __float_as_uint(py) ^ __float_as_uint(py + sexp);
      if (step & 4u) diff |= __float_as_uint(pz) ^ __float_as_uint(pz + sexp);
      scale = diff != 0u ? 31 - __builtin_clz(diff) : -1;
      sexp = __uint_as_float(((uint32_t)scale - (uint32_t)kMaxScale + 127u) << 23);
      if (scale < kMaxScale) {
        const int lv = scale - kStackBase;
        if (lv >= 0 && lv < kStackLevels && ((written >> lv) & 1u)) {
          const uint2 e = stk.pm[lv * 64 + lane];
          pbase = e.x;
          t_max = __uint_as_float(e.y);
          pmask = stk.mk[lv * 64 + lane];
        } else if (scale >= 0) {
          pbase = 0u; pmask = 0u; t_max = 0.0f;  // never-pushed level: the reference stack holds zeros
        }
      }
      const uint32_t sh = (uint32_t)scale & 31u;
      const uint32_t sx = __float_as_uint(px) >> sh, sy = __float_as_uint(py) >> sh, sz = __float_as_uint(pz) >> sh;
      px = __uint_as_float(sx << sh);
      py = __uint_as_float(sy << sh);
      pz = __uint_as_float(sz << sh);
      idx = (sx & 1u) | ((sy & 1u) << 1) | ((sz & 1u) << 2);
      h = 0.0f;
    }
  }

  if (kCount) {
    cnt.rays++;
    const uint32_t it = iter > kMaxIter ? kMaxIter : iter;
    cnt.iters += it;
    cnt.bytes += bytes;
    cnt.max_iter = it > cnt.max_iter ? it : cnt.max_iter;
  }
  res.iter = iter;
  res.t = t_min;
  res.scale_exp2 = sexp;
  res.capped = capped;
  if (!hit) return res;

  // hit: decode the node we stopped on (svotrace.comp:380-431)
  uint32_t raw = 0u;
  if (tag == 1u) raw = rec_normal_le(rec);
  else if (tag != 3u) raw = rec_mask_be(rec);
  V3 n = mk(0.f, 0.f, 0.f);
  if (raw != 0u) {
    const int r = (int)raw;
    const float nx = (float)((r % 10) - 5);
    const float ny = (float)((((r % 100) - (r % 10)) / 10) - 5);
    const float nz = (float)(((r - (r % 100)) / 100) - 5);
    n = normalize3(mk(nx, ny, nz));
  }
  res.hit = true;
  res.pointer = cptr;
  res.value = rec_value(rec);
  res.raw = raw;
  res.level = (uint32_t)(kMaxScale - scale);
  res.normal = n;
  float vx = px, vy = py, vz = pz;
  if (d.x > 0.0f) vx = 3.0f - vx - sexp;
  if (d.y > 0.0f) vy = 3.0f - vy - sexp;
  if (d.z > 0.0f) vz = 3.0f - vz - sexp;
  vx += ((n.x * sexp) * 2.0f) * 1.74f;
  vy += ((n.y * sexp) * 2.0f) * 1.74f;
  vz += ((n.z * sexp) * 2.0f) * 1.74f;
  res.voxel_pos = mk(vx, vy, vz);
  return res;
}

// material colour table of svotrace.comp:514-522 / :577-586
// renderMode 2 leaves its colour variable unset for values other than 1..3 (svotrace.comp:577-586); the reference under
// llvmpipe resolves the undefined value to material 1's colour (tests/golden/fuzz_golden.npz: values 4 and 127)
#define kMode2OtherMaterial mk(0.84f, 0.86f, 0.78f)
__device__ __forceinline__ V3 material_colour(uint32_t value, V3 other) {
  if (value == 1u) return mk(0.84f, 0.86f, 0.78f);
  if (value == 2u) return mk(0.57f, 0.5f, 0.31f);
  if (value == 3u) return mk(0.37f, 0.43f, 0.27f);
  return other;
}

// imageStore to rgba8: clamp, round half to even; NaN stores 255
__device__ __forceinline__ uint32_t unorm8(float x) {
  if (x != x) return 255u;
  if (x <= 0.0f) return 0u;
  if (x >= 1.0f) return 255u;
  return (uint32_t)__builtin_rintf(x * 255.0f);
}

}  // namespace svo
