// svo_trav2.h -- the cast over the interior-descriptor table (svo_derive.hip.h) instead of the pool's records.
//
// Same loop as svo_trav.h / svotrace.comp:262-369, same arithmetic in the same order; what changes is where the three
// facts about the child come from: "empty" (svotrace.comp:295), "has a child block" (:311) are a nibble of the PARENT's
// descriptor, already in registers; only a DESCEND (or a POP, which re-reads its ancestor's descriptor) loads -- one
// aligned 8-byte descriptor.  The hit node's record (value, normal, pointer) is fetched once, after the loop.
//   trav_step2()  the readable statement (SVO_ASM_LOOP=0 builds)
//   trav_loop2()  the same trips in gfx950 assembly
#pragma once
#include "svo_trav.h"

namespace svo {

struct DescTab {
  __amdgpu_buffer_rsrc_t rsrc;   // the descriptors, 8 bytes each, addressed by byte offset
  const uint2 *aux;              // {child-block base, tag mask} of every descriptor
  const float4 *ntab;            // the unit normal of every 16-bit normal code (normal_table_kernel), or null: decode in place
};
__device__ __forceinline__ DescTab make_desctab(const uint2 *desc, const uint2 *aux, uint32_t count, const float4 *ntab = nullptr) {
  DescTab t;
  t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)desc, 0, (int)(count * 8u), 0x00020000);
  t.aux = aux;
  t.ntab = ntab;
  return t;
}
// the unit normal a node's 16-bit code stands for (svotrace.comp:405-421: three decimal digits - 5, normalised; 0 = no normal)
__device__ __forceinline__ V3 decode_normal(const uint32_t raw) {
  V3 n = mk(0.f, 0.f, 0.f);
  if (raw != 0u) {
    const int r = (int)raw;
    const float nx = (float)((r % 10) - 5);
    const float ny = (float)((((r % 100) - (r % 10)) / 10) - 5);
    const float nz = (float)(((r - (r % 100)) / 100) - 5);
    n = normalize3(mk(nx, ny, nz));
  }
  return n;
}
// all 65 536 of them, once per context: the same function on every code, so a look-up returns the bits the decode would
__global__ void normal_table_kernel(float4 *t) {
  const uint32_t raw = blockIdx.x * blockDim.x + threadIdx.x;
  if (raw >= 65536u) return;
  const V3 n = decode_normal(raw);
  t[raw] = make_float4(n.x, n.y, n.z, 0.0f);
}
constexpr uint32_t kDescPhantom = 0u, kDescRoot = 8u;   // byte offsets of descriptors 0 and 1

// LDS stack of one wave: [level][lane] of {descriptor offset of the parent, t_max}
struct WaveStack2 {
  uint2 pm[kStackLevels * 64];
};

struct Trav2 {
  float cx, cy, cz, bx, by, bz;
  float px, py, pz;
  float t_min, t_max, h, sexp;
  int scale, lod_scale;
  float cone_t;
  uint32_t idx, octant, self, dlo, dhi, written, iter;
};

// set-up part of the cast (svotrace.comp:221-260); `rootd` = the root's descriptor, fetched once per wave
__device__ __forceinline__ int trav_init2(const uint2 rootd, Trav2 &t, V3 o, V3 d, const bool cone, const float t_start = 0.0f) {
  t.cone_t = cone ? 0.05f : __builtin_inff();
  t.iter = 0; t.written = 0; t.lod_scale = kMaxScale - kMaxDepth;
  t.scale = kMaxScale - 1; t.sexp = 0.5f;
  t.self = kDescRoot; t.dlo = rootd.x; t.dhi = rootd.y;
  if (all_nan(o) || all_nan(d)) {  // quirk Q7: the reference spins to the cap, iter = 1501
    t.iter = kMaxIter + 1u; t.t_min = 0.0f; t.t_max = 0.0f; t.h = 0.0f; t.octant = 0; t.idx = 0;
    t.cx = t.cy = t.cz = t.bx = t.by = t.bz = 0.0f; t.px = t.py = t.pz = 1.0f;
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
  t.t_min = vmax(t.t_min, t_start);
  t.h = t.t_max;
  t.idx = 0; t.px = 1.0f; t.py = 1.0f; t.pz = 1.0f;
  if (1.5f * t.cx - t.bx > t.t_min) { t.idx ^= 1u; t.px = 1.5f; }
  if (1.5f * t.cy - t.by > t.t_min) { t.idx ^= 2u; t.py = 1.5f; }
  if (1.5f * t.cz - t.bz > t.t_min) { t.idx ^= 4u; t.pz = 1.5f; }
  return ST_ACTIVE;
}

// one iteration of the loop at svotrace.comp:262-369
__device__ __forceinline__ int trav_step2(const DescTab &tab, WaveStack2 &stk, const uint32_t lane, Trav2 &t) {
  t.iter++;
  if (t.iter > kMaxIter) return ST_CAPPED;
  if (t.t_min > t.cone_t) t.lod_scale = kMaxScale - 11;
  const float tcx = t.px * t.cx - t.bx;
  const float tcy = t.py * t.cy - t.by;
  const float tcz = t.pz * t.cz - t.bz;
  const float tc_max = vmin3(tcx, tcy, tcz);
  const uint32_t cs = t.idx ^ t.octant;
  const uint32_t nib = (t.dhi >> (4u * cs)) & 15u;   // 0 empty, 1 not empty without a child block, 8 | rank: descend (svo_derive.hip.h)
  const bool ne = nib != 0u, has = nib >= 8u;
  if (ne && t.t_min <= t.t_max) {
    if (t.scale == t.lod_scale) return ST_HIT;
    const float tv_max = vmin(t.t_max, tc_max);
    if (t.t_min <= tv_max) {
      if (!has) return ST_HIT;
      const float half = t.sexp * 0.5f;
      const float tmx = half * t.cx + tcx;
      const float tmy = half * t.cy + tcy;
      const float tmz = half * t.cz + tcz;
      if (tc_max < t.h) {  // PUSH
        const uint32_t lvu = (uint32_t)(t.scale - kStackBase);
        const uint32_t lv = lvu < (uint32_t)(kStackLevels - 1) ? lvu : (uint32_t)(kStackLevels - 1);
        stk.pm[(lv << 6) | lane] = make_uint2(t.self, __float_as_uint(t.t_max));
        t.written |= 1u << lv;
      }
      t.h = tc_max;
      t.self = t.dlo + 8u * nib;   // (dlo = the first child descriptor's offset - 64)
      { const u32x2 dd = __builtin_amdgcn_raw_buffer_load_b64(tab.rsrc, (int)t.self, 0, 0); t.dlo = dd.x; t.dhi = dd.y; }
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
    const uint32_t diff = (__float_as_uint(t.px) ^ __float_as_uint(opx)) | (__float_as_uint(t.py) ^ __float_as_uint(opy)) |
                          (__float_as_uint(t.pz) ^ __float_as_uint(opz));
    t.scale = 31 - __builtin_clz(diff | 1u);
    t.sexp = __uint_as_float(((uint32_t)t.scale - (uint32_t)kMaxScale + 127u) << 23);
    const uint32_t lvu = (uint32_t)(t.scale - kStackBase);
    const uint32_t lv = lvu < (uint32_t)(kStackLevels - 1) ? lvu : (uint32_t)(kStackLevels - 1);
    const bool have = ((t.written >> (lvu & 31u)) & 1u) != 0u;
    const uint2 e = stk.pm[(lv << 6) | lane];
    // a level this ray never pushed reads as the reference's zero-initialised stack entry: the state (0, 0) = descriptor 0
    t.self = have ? e.x : kDescPhantom;
    t.t_max = have ? __uint_as_float(e.y) : 0.0f;
    { const u32x2 dd = __builtin_amdgcn_raw_buffer_load_b64(tab.rsrc, (int)t.self, 0, 0); t.dlo = dd.x; t.dhi = dd.y; }
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

// result part of the cast (svotrace.comp:371-431): `self` is the parent state the ray stopped in, `cs` its child slot
__device__ __forceinline__ Cast cast_result2(const BufPool &pool, const DescTab &tab, int status, uint32_t self, uint32_t cs,
                                             uint32_t octant, uint32_t iter, float t_min, float sexp, int scale, float px,
                                             float py, float pz) {
  Cast res;
  res.hit = status == ST_HIT;
  res.capped = status == ST_CAPPED;
  res.pointer = 0; res.value = 0; res.raw = 0; res.level = 0;
  res.normal = mk(0.f, 0.f, 0.f); res.voxel_pos = mk(0.f, 0.f, 0.f);
  res.iter = iter;
  res.t = t_min;
  res.scale_exp2 = sexp;
  if (!res.hit) return res;
  const uint2 a = tab.aux[self >> 3];
  const uint32_t M = a.y & 0xffffu, tag = (M >> (2u * cs)) & 3u;
  const uint32_t cptr = a.x + child_offset(M, cs);
  const u32x2 rr = load_record2(pool, cptr);
  uint32_t raw = 0u;
  if (tag == 1u) raw = (rr.x >> 8) & 0xffffu;   // packed normal, u16 little-endian in bytes 1..2
  else if (tag != 3u) raw = rec2_mask_be(rr.y);
  V3 n;
  if (tab.ntab) { const float4 q = tab.ntab[raw]; n = mk(q.x, q.y, q.z); }
  else n = decode_normal(raw);
  res.pointer = cptr;
  res.value = rr.x & 0xffu;
  res.raw = raw;
  res.level = (uint32_t)(kMaxScale - scale);
  res.normal = n;
  float vx = px, vy = py, vz = pz;
  if (octant & 1u) vx = 3.0f - vx - sexp;
  if (octant & 2u) vy = 3.0f - vy - sexp;
  if (octant & 4u) vz = 3.0f - vz - sexp;
  vx += ((n.x * sexp) * 2.0f) * 1.74f;
  vy += ((n.y * sexp) * 2.0f) * 1.74f;
  vz += ((n.z * sexp) * 2.0f) * 1.74f;
  res.voxel_pos = mk(vx, vy, vz);
  return res;
}
__device__ __forceinline__ Cast trav_result2(const BufPool &pool, const DescTab &tab, const Trav2 &t, int status) {
  return cast_result2(pool, tab, status, t.self, t.idx ^ t.octant, t.octant, t.iter, t.t_min, t.sexp, t.scale, t.px, t.py, t.pz);
}

}  // namespace svo
