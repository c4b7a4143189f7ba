// svo_trav.h -- the cast of svo_device.h cut into init / step / result pieces so that a
// persistent wave can keep 64 traversals in flight and swap finished rays for new ones
// between steps.  Same arithmetic, same order, as cast_ray() (svotrace.comp:211-432);
// the instruction stream is trimmed for the issue-bound inner loop:
//   * records come through a buffer descriptor (SGPR base + 32-bit lane offset, hardware
//     range check: a pointer past the pool reads zeros, which is the reference's
//     zero-filled over-allocation) instead of compare / select / 64-bit address math;
//   * min / max are the bare v_min_f32 / v_max_f32 / v_min3_f32 (IEEE mode: a quiet NaN
//     operand is ignored = GLSL-on-llvmpipe semantics); the builtins make the compiler
//     canonicalise both inputs first (3 instructions per min);
//   * the POP reads its stack slot unconditionally from a clamped level and selects.
#pragma once
#include "svo_device.h"

namespace svo {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The pool behind a buffer descriptor.  num_records covers the zero padding the
// library keeps behind the pool, so an 8-byte read that starts inside never straddles.
struct BufPool {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ BufPool make_bufpool(const uint8_t *base, uint32_t len) {
  BufPool p;
  p.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(len + 16u), 0x00020000);
  return p;
}
__device__ __forceinline__ uint64_t load_record(const BufPool &pool, uint32_t p) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(pool.rsrc, (int)p, 0, 0);
  return ((uint64_t)v.y << 32) | (uint64_t)v.x;
}
__device__ __forceinline__ u32x2 load_record2(const BufPool &pool, uint32_t p) {
  return __builtin_amdgcn_raw_buffer_load_b64(pool.rsrc, (int)p, 0, 0);
}
// field decoders on the two dwords of a record: one v_perm_b32 each
// (selector bytes 0..3 address the second operand, 4..7 the first, 0x0c = constant zero)
__device__ __forceinline__ uint32_t rec2_cp(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x01020304u); }
__device__ __forceinline__ uint32_t rec2_mask_be(uint32_t hi) { return __builtin_amdgcn_perm(hi, hi, 0x0c0c0102u); }

// Traversal state of one ray, held in registers across refill rounds.
struct Trav {
  float cx, cy, cz, bx, by, bz;
  float px, py, pz;
  float t_min, t_max, h, sexp;
  int scale, lod_scale;   // lod_scale = kMaxScale - maxDepth: a non-empty child met at this scale is a hit
  float cone_t;   // t_min beyond which a cone (secondary) ray drops to LOD 11; +inf for other rays
  uint32_t idx, octant, pbase, pmask, written, iter;
  uint32_t cptr, tag;
  uint32_t rlo, rhi;   // the record at cptr (bytes 0..3, 4..7)
};

enum : int { ST_SKIP = -1, ST_IDLE = 0, ST_ACTIVE = 1, ST_HIT = 2, ST_MISS = 3, ST_CAPPED = 4 };

// set-up part of the cast (svotrace.comp:221-260)
// `root` = the root record (svotrace.comp:222), fetched once per wave by the caller
__device__ __forceinline__ int trav_init(const uint64_t root, Trav &t, V3 o, V3 d, const bool cone, const float t_start = 0.0f) {
  t.cone_t = cone ? 0.05f : __builtin_inff();
  t.iter = 0; t.cptr = 0; t.tag = 0; t.rlo = 0; t.rhi = 0; t.written = 0; t.lod_scale = kMaxScale - kMaxDepth;
  t.scale = kMaxScale - 1; t.sexp = 0.5f;
  if (all_nan(o) || all_nan(d)) {  // quirk Q7: the reference spins to the cap, iter = 1501
    t.iter = kMaxIter + 1u; t.t_min = 0.0f; t.t_max = 0.0f; t.h = 0.0f; t.octant = 0; t.idx = 0;
    t.cx = t.cy = t.cz = t.bx = t.by = t.bz = 0.0f; t.px = t.py = t.pz = 1.0f; t.pbase = 0; t.pmask = 0;
    return ST_CAPPED;
  }
  if (__builtin_fabsf(d.x) < kEpsilon) d.x = kEpsilon * sign_g(d.x);
  if (__builtin_fabsf(d.y) < kEpsilon) d.y = kEpsilon * sign_g(d.y);
  if (__builtin_fabsf(d.z) < kEpsilon) d.z = kEpsilon * sign_g(d.z);
  t.cx = 1.0f / -__builtin_fabsf(d.x);
  t.cy = 1.0f / -__builtin_fabsf(d.y);
  t.cz = 1.0f / -__builtin_fabsf(d.z);
  t.bx = t.cx * o.x; t.by = t.cy * o.y; t.bz = t.cz * o.z;
  t.octant = 0;
  if (d.x > 0.0f) { t.octant ^= 1u; t.bx = 3.0f * t.cx - t.bx; }
  if (d.y > 0.0f) { t.octant ^= 2u; t.by = 3.0f * t.cy - t.by; }
  if (d.z > 0.0f) { t.octant ^= 4u; t.bz = 3.0f * t.cz - t.bz; }
  t.t_min = vmax3(2.0f * t.cx - t.bx, 2.0f * t.cy - t.by, 2.0f * t.cz - t.bz);
  t.t_max = vmin3(t.cx - t.bx, t.cy - t.by, t.cz - t.bz);
  t.t_min = vmax(t.t_min, 0.0f);
  t.t_min = vmax(t.t_min, t_start);   // beam pre-pass: the walk starts further along the same ray
  t.h = t.t_max;
  t.idx = 0; t.px = 1.0f; t.py = 1.0f; t.pz = 1.0f;
  if (1.5f * t.cx - t.bx > t.t_min) { t.idx ^= 1u; t.px = 1.5f; }
  if (1.5f * t.cy - t.by > t.t_min) { t.idx ^= 2u; t.py = 1.5f; }
  if (1.5f * t.cz - t.bz > t.t_min) { t.idx ^= 4u; t.pz = 1.5f; }
  t.pbase = rec_cp(root);
  t.pmask = rec_mask_be(root);
  return ST_ACTIVE;
}

// one iteration of the loop at svotrace.comp:262-369
#ifdef SVO_STAMPS
#define SVO_STAMP_ARG , unsigned long long &st_load
#else
#define SVO_STAMP_ARG
#endif
__device__ __forceinline__ int trav_step(const BufPool &pool, WaveStack &stk, const uint32_t lane, Trav &t SVO_STAMP_ARG) {
  t.iter++;
  if (t.iter > kMaxIter) return ST_CAPPED;
  if (t.t_min > t.cone_t) t.lod_scale = kMaxScale - 11;
  const float tcx = t.px * t.cx - t.bx;
  const float tcy = t.py * t.cy - t.by;
  const float tcz = t.pz * t.cz - t.bz;
  const float tc_max = vmin3(tcx, tcy, tcz);
  const uint32_t cs = t.idx ^ t.octant;
  t.tag = (t.pmask >> (2u * cs)) & 3u;
  {  // child_offset(): 7 * cs - 4 * popcount(lo) - 2 * popcount(both), kept in 32-bit full-rate instructions
    const uint32_t below = (1u << (2u * cs)) - 1u;
    const uint32_t lo = t.pmask & 0x5555u & below;
    const uint32_t both = lo & (t.pmask >> 1);
    const uint32_t w = (uint32_t)__builtin_popcount(both) + 2u * (uint32_t)__builtin_popcount(lo);
    t.cptr = (__umul24(cs, 7u) + t.pbase) - 2u * w;
  }

#ifdef SVO_STAMPS
  const unsigned long long l0 = __builtin_readcyclecounter();
#endif
  { const u32x2 rr = load_record2(pool, t.cptr); t.rlo = rr.x; t.rhi = rr.y; }
#ifdef SVO_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st_load += __builtin_readcyclecounter() - l0;
#endif
  if ((t.rlo & 0xffu) != 0u && t.t_min <= t.t_max) {
    if (t.scale == t.lod_scale) return ST_HIT;
    const float tv_max = vmin(t.t_max, tc_max);
    if (t.t_min <= tv_max) {
      const uint32_t ccp = t.tag == 0u ? rec2_cp(t.rlo, t.rhi) : 0u;
      if (ccp == 0u) return ST_HIT;
      const float half = t.sexp * 0.5f;
      const float tmx = half * t.cx + tcx;
      const float tmy = half * t.cy + tcy;
      const float tmz = half * t.cz + tcz;
      if (tc_max < t.h) {  // PUSH (scale is 11..22 for pools up to 13 levels; clamp keeps LDS accesses in range)
        const uint32_t lvu = (uint32_t)(t.scale - kStackBase);
        const uint32_t lv = lvu < (uint32_t)(kStackLevels - 1) ? lvu : (uint32_t)(kStackLevels - 1);
        const uint32_t slot = (lv << 6) | lane;
        *(uint2 *)((char *)stk.pm + (slot << 3)) = make_uint2(t.pbase, __float_as_uint(t.t_max));
        *(uint16_t *)((char *)stk.mk + (slot << 1)) = (uint16_t)t.pmask;
        t.written |= 1u << lv;
      }
      t.h = tc_max;
      t.pbase = t.cptr + ccp;
      t.pmask = rec2_mask_be(t.rhi);
      t.idx = 0u;
      --t.scale;
      t.sexp = half;
      if (tmx > t.t_min) { t.idx ^= 1u; t.px += half; }
      if (tmy > t.t_min) { t.idx ^= 2u; t.py += half; }
      if (tmz > t.t_min) { t.idx ^= 4u; t.pz += half; }
      t.t_max = tv_max;
      return ST_ACTIVE;
    }
  }
  uint32_t step = 0u;
  const float opx = t.px, opy = t.py, opz = t.pz;
  if (tcx <= tc_max) { step ^= 1u; t.px -= t.sexp; }
  if (tcy <= tc_max) { step ^= 2u; t.py -= t.sexp; }
  if (tcz <= tc_max) { step ^= 4u; t.pz -= t.sexp; }
  t.t_min = tc_max;
  t.idx ^= step;
  if ((t.idx & step) != 0u) {  // POP
    // the reference XORs pos with pos + cell size on every stepped axis; pos + cell size is exactly the
    // position before the step (all values are multiples of the cell size), and an axis that did not step
    // contributes zero, so the differing bits are simply old ^ new on all three axes
    const uint32_t diff = (__float_as_uint(t.px) ^ __float_as_uint(opx)) | (__float_as_uint(t.py) ^ __float_as_uint(opy)) |
                          (__float_as_uint(t.pz) ^ __float_as_uint(opz));
    // diff != 0 always (a stepped axis changed); |1 only keeps clz defined
    t.scale = 31 - __builtin_clz(diff | 1u);
    t.sexp = __uint_as_float(((uint32_t)t.scale - (uint32_t)kMaxScale + 127u) << 23);
    // restore {child-block base, t_max, tag mask} of that level; a level this ray never pushed reads as the
    // reference's zero-initialised stack entry
    // (scale is 0..30 here; `written` only ever has bits 0..11 set, so the shift by (scale - 11) mod 32 is
    // also the range check)
    const uint32_t lvu = (uint32_t)(t.scale - kStackBase);
    const uint32_t lv = lvu < (uint32_t)(kStackLevels - 1) ? lvu : (uint32_t)(kStackLevels - 1);
    const bool have = ((t.written >> (lvu & 31u)) & 1u) != 0u;
    const uint32_t slot = (lv << 6) | lane;
    const uint2 e = *(const uint2 *)((const char *)stk.pm + (slot << 3));
    const uint32_t m = *(const uint16_t *)((const char *)stk.mk + (slot << 1));
    t.pbase = have ? e.x : 0u;
    t.t_max = have ? __uint_as_float(e.y) : 0.0f;
    t.pmask = have ? m : 0u;
    const uint32_t sh = (uint32_t)t.scale & 31u;
    const uint32_t keep = ~0u << sh;
    const uint32_t bx = __float_as_uint(t.px) & keep, by = __float_as_uint(t.py) & keep, bz = __float_as_uint(t.pz) & keep;
    t.px = __uint_as_float(bx);
    t.py = __uint_as_float(by);
    t.pz = __uint_as_float(bz);
    t.idx = __builtin_amdgcn_ubfe(bx, sh, 1u) | (__builtin_amdgcn_ubfe(by, sh, 1u) << 1) | (__builtin_amdgcn_ubfe(bz, sh, 1u) << 2);
    t.h = 0.0f;
    if (t.scale >= kMaxScale) return ST_MISS;
  }
  return ST_ACTIVE;
}

// result part of the cast (svotrace.comp:371-431)
__device__ __forceinline__ Cast trav_result(const Trav &t, int status) {
  Cast res;
  res.hit = status == ST_HIT;
  res.capped = status == ST_CAPPED;
  res.pointer = 0; res.value = 0; res.raw = 0; res.level = 0;
  res.normal = mk(0.f, 0.f, 0.f); res.voxel_pos = mk(0.f, 0.f, 0.f);
  res.iter = t.iter;
  res.t = t.t_min;
  res.scale_exp2 = t.sexp;
  if (!res.hit) return res;
  uint32_t raw = 0u;
  if (t.tag == 1u) raw = (t.rlo >> 8) & 0xffffu;   // packed normal, u16 little-endian in bytes 1..2
  else if (t.tag != 3u) raw = rec2_mask_be(t.rhi);
  V3 n = mk(0.f, 0.f, 0.f);
  if (raw != 0u) {
    const int r = (int)raw;
    const float nx = (float)((r % 10) - 5);
    const float ny = (float)((((r % 100) - (r % 10)) / 10) - 5);
    const float nz = (float)(((r - (r % 100)) / 100) - 5);
    n = normalize3(mk(nx, ny, nz));
  }
  res.pointer = t.cptr;
  res.value = t.rlo & 0xffu;
  res.raw = raw;
  res.level = (uint32_t)(kMaxScale - t.scale);
  res.normal = n;
  float vx = t.px, vy = t.py, vz = t.pz;
  if (t.octant & 1u) vx = 3.0f - vx - t.sexp;
  if (t.octant & 2u) vy = 3.0f - vy - t.sexp;
  if (t.octant & 4u) vz = 3.0f - vz - t.sexp;
  vx += ((n.x * t.sexp) * 2.0f) * 1.74f;
  vy += ((n.y * t.sexp) * 2.0f) * 1.74f;
  vz += ((n.z * t.sexp) * 2.0f) * 1.74f;
  res.voxel_pos = mk(vx, vy, vz);
  return res;
}

}  // namespace svo
