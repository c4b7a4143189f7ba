// svo_travloop.h -- the traversal trips of a persistent wave, written in gfx950 assembly.
//
// trav_loop() runs trav_step() (svo_trav.h, the readable statement of the same arithmetic) on
// the wave's active lanes until no more than `threshold` of them are still traversing.  The
// instruction stream is the kernel's critical resource: a wave issues one instruction every ~5
// cycles, a SIMD retires a wave64 VALU instruction every ~2.5 (tools/calib_valu.hip), and every
// trip executes the descend, advance and pop sections one after the other for whichever lanes
// need them.  hipcc's version of the loop is 123 vector instructions per trip (phi copies on the
// back edge, status bookkeeping in a VGPR, hazard nops); this one is 96 (+ 30 scalar / branch):
//   * lane sets (active / hit / descend / advance / pop / cone rays) live in SGPR pairs and are combined
//     on the scalar unit, the record-independent part before the wait; the status VGPR is only
//     written when a lane stops; the rare exits (iteration cap, leaving the octree) are out of line;
//   * the child index is not carried from trip to trip: a cell's origin is a multiple of its size, so the
//     index is bit `scale` of the three position components (3 bit-field extracts + 2 shift-ors at the top
//     of a trip; the descend and pop sections need not rebuild it, the advance section tests
//     step & ~index instead of updating it: -4 instructions per trip, +1.0 % frames/s);
//   * the three per-axis comparisons of a step select increments (0 or the cell size) that update the
//     position in place -- no old/new copies of the position; a carry chain (v_addc_co_u32) builds the
//     step mask;
//   * the advance step of every active lane is computed while the record is in flight;
//   * the record comes in two dword loads (34 cycles each in the texture path for a divergent
//     wave, any alignment) instead of one misaligned dwordx2 (96 cycles, tools/calib_td.hip);
//   * the pushed {child-block base, t_max} pair goes to LDS with one ds_write2_b32 from the two
//     registers where they live.
// Tried and dropped here: asking for the first line of the child block as soon as a descend is decided (a third
// buffer_load_dword per trip into a register nobody reads): -9 % -- the texture path is the busier unit.
// Arithmetic, operand order and rounding are those of trav_step(); the parity tests run both.
//
// Hazards observed (gfx950): a VALU write of an SGPR pair / VCC needs two other instructions
// before a VALU reads it as a mask (v_cndmask, v_addc); SALU reads are interlocked.  Loads are
// counted here (s_waitcnt vmcnt(0) / lgkmcnt(0) inside the block).
//
// Most operands are allocated by the compiler; the asm needs both the whole and the halves of the
// y/z pairs, so those are pinned: v[68:69] py,pz; v72 cell size (v[72:73] is the broadcast source of a
// packed multiply, v73 is scratch); temporaries v[86:87] tcy,tcz, v[92:93]; v[88:89] the record of the
// last visited child (lanes are masked once they stop, so a hit lane's record stays put).
#pragma once
#include "svo_trav.h"

namespace svo {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The second dword of a child record (low byte of the child pointer, tag mask) only matters to a lane that can
// still descend into the child.  SVO_LOAD2_NARROWED=1 asks for it on those lanes only (a ray that stops on an
// interior or tag-2 record then fetches the dword once, after the loop, in trav_result_regs): measured 2.4 % SLOWER
// than both dwords for every active lane right behind the address (tools/history/r03_ab_ld2.sh) -- the texture path's cost
// is per wave-level load, not per lane, and the narrowed load leaves ~15 instructions later.  Kept as an A/B switch.
#ifndef SVO_LOAD2_NARROWED
#define SVO_LOAD2_NARROWED 0
#endif
// SVO_LOAD_X3=1: one dword-aligned dwordx3 (it always covers the 7 bytes) + two v_alignbyte_b32, instead of two
// dword loads of any alignment: one tag lookup per record instead of two, and no second request that finds the
// line of the first one still on its way from L2.
#ifndef SVO_LOAD_X3
#define SVO_LOAD_X3 0
#endif
#if SVO_LOAD_X3
#define SVO_RECORD_LOAD                                                      \
  "v_and_b32 %[t3], -4, %[cptr]\n\t"                                         \
  "buffer_load_dwordx3 v[88:90], %[t3], %[rs], 0 offen\n\t"
#define SVO_RECORD_ALIGN                                                     \
  "v_alignbyte_b32 v88, v89, v88, %[cptr]\n\t"                               \
  "v_alignbyte_b32 v89, v90, v89, %[cptr]\n\t"
#define SVO_LOAD2_NARROW
#elif SVO_LOAD2_NARROWED
#define SVO_RECORD_ALIGN
#define SVO_RECORD_LOAD "buffer_load_dword v88, %[cptr], %[rs], 0 offen\n\t"
#define SVO_LOAD2_NARROW                                                     \
  "s_mov_b64 exec, %[sd]\n\t"                                                \
  "buffer_load_dword v89, %[cptr], %[rs], 0 offen offset:4\n\t"              \
  "s_mov_b64 exec, %[act]\n\t"
#else
#define SVO_RECORD_ALIGN
#define SVO_RECORD_LOAD                                                      \
  "buffer_load_dword v88, %[cptr], %[rs], 0 offen\n\t"                       \
  "buffer_load_dword v89, %[cptr], %[rs], 0 offen offset:4\n\t"
#define SVO_LOAD2_NARROW
#endif


// per-ray constants and state in the register layout of trav_loop()
struct TravRegs {
  float cx, bx;
  f32x2 cyz, byz;
  uint32_t octant;
  float px;
  float py, pz;
  float t_min, t_max, sexp, h;
  int scale;
  uint32_t idx, pbase, pmask, written, iter;
  int lod_scale;
  uint32_t cptr, tag, rlo, rhi;
};

// set-up part of the cast (svotrace.comp:221-260)
// `root` = the root record (svotrace.comp:222), fetched once per wave by the caller
__device__ __forceinline__ int trav_init_regs(const uint64_t root, TravRegs &t, V3 o, V3 d, const bool cone,
                                              const float t_start = 0.0f) {
  (void)cone;   // which lanes carry cone (secondary) rays is a lane set the caller passes to trav_loop
  t.iter = 0; t.cptr = 0; t.tag = 0; t.rlo = 0; t.rhi = 0; t.written = 0; t.lod_scale = kMaxScale - kMaxDepth;
  t.scale = kMaxScale - 1; t.sexp = 0.5f;
  if (all_nan(o) || all_nan(d)) {  // quirk Q7: the reference spins to the cap, iter = 1501
    t.iter = kMaxIter + 1u; t.t_min = 0.0f; t.t_max = 0.0f; t.h = 0.0f; t.octant = 0; t.idx = 0;
    t.cx = t.cyz.x = t.cyz.y = t.bx = t.byz.x = t.byz.y = 0.0f; t.px = t.py = t.pz = 1.0f; t.pbase = 0; t.pmask = 0;
    return ST_CAPPED;
  }
  if (__builtin_fabsf(d.x) < kEpsilon) d.x = kEpsilon * sign_g(d.x);
  if (__builtin_fabsf(d.y) < kEpsilon) d.y = kEpsilon * sign_g(d.y);
  if (__builtin_fabsf(d.z) < kEpsilon) d.z = kEpsilon * sign_g(d.z);
  t.cx = 1.0f / -__builtin_fabsf(d.x);
  t.cyz.x = 1.0f / -__builtin_fabsf(d.y);
  t.cyz.y = 1.0f / -__builtin_fabsf(d.z);
  t.bx = t.cx * o.x; t.byz.x = t.cyz.x * o.y; t.byz.y = t.cyz.y * o.z;
  t.octant = 0;
  if (d.x > 0.0f) { t.octant ^= 1u; t.bx = 3.0f * t.cx - t.bx; }
  if (d.y > 0.0f) { t.octant ^= 2u; t.byz.x = 3.0f * t.cyz.x - t.byz.x; }
  if (d.z > 0.0f) { t.octant ^= 4u; t.byz.y = 3.0f * t.cyz.y - t.byz.y; }
  t.t_min = vmax3(2.0f * t.cx - t.bx, 2.0f * t.cyz.x - t.byz.x, 2.0f * t.cyz.y - t.byz.y);
  t.t_max = vmin3(t.cx - t.bx, t.cyz.x - t.byz.x, t.cyz.y - t.byz.y);
  t.t_min = vmax(t.t_min, 0.0f);
  t.t_min = vmax(t.t_min, t_start);   // beam pre-pass: the walk starts further along the same ray
  t.h = t.t_max;
  t.idx = 0; t.px = 1.0f; t.py = 1.0f; t.pz = 1.0f;
  if (1.5f * t.cx - t.bx > t.t_min) { t.idx ^= 1u; t.px = 1.5f; }
  if (1.5f * t.cyz.x - t.byz.x > t.t_min) { t.idx ^= 2u; t.py = 1.5f; }
  if (1.5f * t.cyz.y - t.byz.y > t.t_min) { t.idx ^= 4u; t.pz = 1.5f; }
  t.pbase = rec_cp(root);
  t.pmask = rec_mask_be(root);
  return ST_ACTIVE;
}


// result part of the cast (svotrace.comp:371-431)
__device__ __forceinline__ Cast trav_result_regs(const BufPool &pool, const TravRegs &t, int status) {
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
  else if (t.tag != 3u) {
#if SVO_LOAD2_NARROWED && !SVO_LOAD_X3
    raw = rec2_mask_be((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pool.rsrc, (int)(t.cptr + 4u), 0, 0));
#else
    raw = rec2_mask_be(t.rhi);
#endif
  }
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


// LDS byte offset of a __shared__ object
template <typename T>
__device__ __forceinline__ uint32_t lds_offset(T *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) T *)p;
}

// SVO_STAMPS builds count, per wave: trips, the trips in which the descend / advance / pop sections ran, and the lanes
// that were active in the trip and in each section (eight scalar counters, added up by the kernel at its end)
#ifdef SVO_STAMPS
#define SVO_COUNT(slot, lanes, mask)                     \
  "s_bcnt1_i32_b64 %[cnt], " mask "\n\t"                \
  "s_add_u32 %[" slot "], %[" slot "], 1\n\t"             \
  "s_add_u32 %[" lanes "], %[" lanes "], %[cnt]\n\t"
#else
#define SVO_COUNT(slot, lanes, mask)
#endif

// Run trips until at most `threshold` lanes of `act` (the lanes with status == ST_ACTIVE) are still traversing.
// Lanes that stop get their status (ST_HIT / ST_MISS / ST_CAPPED); r.rlo/r.rhi are then the hit record.
// `cone_lanes`: the lanes whose ray is a cone (secondary) ray -- they drop to LOD 11 once t_min > 0.05 (svotrace.comp:275-277).
__device__ __forceinline__ void trav_loop(const BufPool &pool, WaveStack &stk, const uint32_t lane, TravRegs &r,
                                          int &status, unsigned long long act, const int threshold,
                                          const unsigned long long cone_lanes, uint32_t *mix = nullptr) {
  const uint32_t lds8 = lds_offset(&stk.pm[lane]);
  const uint32_t lds2 = lds_offset(&stk.mk[lane]);
  unsigned long long sv, sa, sb, sc, sd, se, sf, sg, sh;
  int cnt;
#ifdef SVO_STAMPS
#define SVO_RFL(i) (uint32_t) __builtin_amdgcn_readfirstlane((int)mix[i])
  uint32_t c0 = SVO_RFL(0), c1 = SVO_RFL(1), c2 = SVO_RFL(2), c3 = SVO_RFL(3), c4 = SVO_RFL(4), c5 = SVO_RFL(5), c6 = SVO_RFL(6), c7 = SVO_RFL(7);
#undef SVO_RFL
#else
  (void)mix;
#endif
  uint32_t t0, t1, t2, t3;
  float tcx, tcm;
  asm volatile(
      "s_mov_b64 %[sv], exec\n"
      "Ltrip%=:\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      SVO_COUNT("c0", "c1", "exec")
      // ---- child slot, iteration cap (svotrace.comp:263-266)
      // the child index is not carried: it is the bit `scale` of the three position components (a cell's origin is a
      // multiple of its size), so the descend / pop sections need not rebuild it
      "v_bfe_u32 %[t0], %[px], %[scale], 1\n\t"
      "v_bfe_u32 %[t1], v68, %[scale], 1\n\t"
      "v_bfe_u32 %[t2], v69, %[scale], 1\n\t"
      "v_lshl_or_b32 %[t0], %[t1], 1, %[t0]\n\t"
      "v_lshl_or_b32 %[idx], %[t2], 2, %[t0]\n\t"                // idx = x | y << 1 | z << 2
      "v_xor_b32 %[t0], %[idx], %[oct]\n\t"                       // cs = idx ^ octant
      "v_add_u32 %[iter], 1, %[iter]\n\t"                         // iter++
      "v_lshlrev_b32 %[t1], 1, %[t0]\n\t"                     // 2 cs
      "v_cmp_lt_u32 vcc, 0x5dc, %[iter]\n\t"                  // iter > 1500
      "v_lshrrev_b32 %[t2], 1, %[pmask]\n\t"
      "v_bfe_u32 %[tag], %[pmask], %[t1], 2\n\t"                    // tag of the child
      "v_lshlrev_b32_e64 %[t1], %[t1], -1\n\t"                // ~(children below cs)
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cbranch_scc1 Lcap%=\n"                          // rare, out of line
      "Lnocap%=:\n\t"
      // byte offset of child cs in its sibling block: 7 cs - 4 popcount(lo) - 2 popcount(both)
      "v_bitop3_b32 %[t1], %[pmask], %[k5555], %[t1] bitop3:0x40\n\t"   // lo = pmask & 0x5555 & below
      "v_and_b32 %[t2], %[t1], %[t2]\n\t"                       // both = lo & (pmask >> 1)
      "v_bcnt_u32_b32 %[t2], %[t2], 0\n\t"
      "v_bcnt_u32_b32 %[t2], %[t1], %[t2]\n\t"
      "v_bcnt_u32_b32 %[t2], %[t1], %[t2]\n\t"                  // popcount(both) + 2 popcount(lo)
      "v_mad_u32_u24 %[cptr], %[t0], 7, %[pbase]\n\t"
      "v_mad_i32_i24 %[cptr], %[t2], -2, %[cptr]\n\t"               // cptr
      // two dword loads: the texture path takes 96 cycles per CU for a wave of misaligned, divergent dwordx2 and
      // 34 for a dword of any alignment (tools/calib_td.hip)
      SVO_RECORD_LOAD
      // ---- exit distances of the current cell (svotrace.comp:268-269)
      "v_mul_f32 %[tcx], %[px], %[cx]\n\t"
      "v_pk_mul_f32 v[86:87], v[68:69], %[cyz]\n\t"
      "v_cmp_lt_f32 vcc, %[k005], %[tmin]\n\t"                    // t_min > 0.05 ...
      "v_sub_f32 %[tcx], %[tcx], %[bx]\n\t"
      "v_pk_add_f32 v[86:87], v[86:87], %[byz] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "s_and_b64 vcc, vcc, %[conem]\n\t"                          // ... on a cone (secondary) ray: LOD 11 from here on (sticky)
      "v_cmp_le_f32_e64 %[sa], %[tmin], %[tmax]\n\t"              // t_min <= t_max
      "v_min3_f32 %[tcm], %[tcx], v86, v87\n\t"                 // tc_max
      "v_cndmask_b32_e64 %[lod], %[lod], 12, vcc\n\t"
      "v_min_f32 %[t3], %[tmax], %[tcm]\n\t"                       // tv_max
      "v_cmp_eq_u32_e64 %[sb], %[scale], %[lod]\n\t"              // at the LOD scale
      "v_cmp_le_f32_e64 %[sc], %[tmin], %[t3]\n\t"              // t_min <= tv_max
      "v_cmp_eq_u32_e64 %[sd], 0, %[tag]\n\t"                // interior tag
      // lane sets, the part that does not need the record (scalar unit, overlaps the load)
      "s_or_b64 %[se], %[sb], %[sc]\n\t"
      "s_and_b64 %[se], %[se], %[sa]\n\t"                 // in range & (at LOD | inside): hits or descends if not empty
      "s_and_b64 %[sd], %[sd], %[sc]\n\t"
      "s_andn2_b64 %[sd], %[sd], %[sb]\n\t"
      "s_and_b64 %[sd], %[sd], %[sa]\n\t"                 // in range & !at LOD & inside & interior tag: descends if it has a child block
      SVO_LOAD2_NARROW
      // the ADVANCE step of every active lane, computed while the record is in flight (it does not depend on the
      // record); lanes that turn out to hit or descend simply do not commit it
      "v_cmp_le_f32 vcc, %[tcx], %[tcm]\n\t"
      "v_cmp_le_f32_e64 %[sg], v86, %[tcm]\n\t"
      "v_cmp_le_f32_e64 %[sh], v87, %[tcm]\n\t"
      "v_cndmask_b32_e64 %[t0], 0, v72, vcc\n\t"            // per-axis decrement: the cell size or 0
      "v_cndmask_b32_e64 v92, 0, v72, %[sg]\n\t"
      "v_cndmask_b32_e64 v93, 0, v72, %[sh]\n\t"
      "v_cndmask_b32_e64 %[t2], 0, 1, %[sh]\n\t"
      "v_addc_co_u32_e64 %[t2], %[sf], %[t2], %[t2], %[sg]\n\t"
      "v_addc_co_u32_e64 %[t2], %[sf], %[t2], %[t2], vcc\n\t"   // step mask
      "s_waitcnt vmcnt(0)\n\t"
      SVO_RECORD_ALIGN
      "v_cmp_ne_u32_sdwa %[sa], v88, %[zero] src0_sel:BYTE_0 src1_sel:DWORD\n\t"   // value != 0
      "v_perm_b32 %[t1], v89, v88, %[selcp]\n\t"            // child pointer (big-endian bytes 1..4)
      "v_cmp_ne_u32_e64 vcc, 0, %[t1]\n\t"
      // lane sets
      "s_and_b64 %[sd], %[sd], vcc\n\t"
      "s_and_b64 %[sd], %[sd], %[sa]\n\t"                 // DESCEND = not empty & in range & !at LOD & inside & child block
      "s_and_b64 %[se], %[se], %[sa]\n\t"                 // not empty & in range & (at LOD | inside)
      "s_andn2_b64 %[sa], exec, %[se]\n\t"                // ADVANCE = the rest
      "s_andn2_b64 %[se], %[se], %[sd]\n\t"               // HIT = not empty & in range & (at LOD | (inside & no child block))
      "s_mov_b64 exec, %[se]\n\t"
      "v_mov_b32 %[st], 2\n\t"                              // ST_HIT
      "s_andn2_b64 %[act], %[act], %[se]\n\t"
      // ---- DESCEND (svotrace.comp:291-327)
      "s_mov_b64 exec, %[sd]\n\t"
      "s_cbranch_execz LnoD%=\n\t"
      SVO_COUNT("c2", "c3", "exec")
      "v_cmp_lt_f32 vcc, %[tcm], %[h]\n\t"                    // tc_max < h: PUSH
      "v_mul_f32 v72, 0.5, v72\n\t"                       // half
      "v_add_u32 %[t0], -11, %[scale]\n\t"
      "v_min_u32 %[t0], 11, %[t0]\n\t"                        // stack level
      "s_and_saveexec_b64 %[sb], vcc\n\t"
      "v_lshl_add_u32 v93, %[t0], 9, %[lds8]\n\t"
      "v_lshl_add_u32 v92, %[t0], 7, %[lds2]\n\t"
      "ds_write2_b32 v93, %[pbase], %[tmax] offset1:1\n\t"         // {child-block base, t_max}
      "ds_write_b16 v92, %[pmask]\n\t"                         // tag mask
      "v_lshl_or_b32 %[wr], 1, %[t0], %[wr]\n\t"
      "s_mov_b64 exec, %[sd]\n\t"
      "v_mul_f32 %[t0], %[cx], v72\n\t"
      "v_pk_mul_f32 v[92:93], %[cyz], v[72:73] op_sel_hi:[1,0]\n\t"
      "v_add_f32 %[t0], %[t0], %[tcx]\n\t"                       // centre distances
      "v_pk_add_f32 v[92:93], v[92:93], v[86:87]\n\t"
      "v_cmp_gt_f32 vcc, %[t0], %[tmin]\n\t"
      "v_cmp_gt_f32_e64 %[sb], v92, %[tmin]\n\t"
      "v_cmp_gt_f32_e64 %[sc], v93, %[tmin]\n\t"
      "v_add_u32 %[pbase], %[t1], %[cptr]\n\t"                       // child-block base of the child
      "v_perm_b32 %[pmask], v89, v89, %[selmask]\n\t"          // its tag mask (big-endian bytes 5..6)
      "v_cndmask_b32_e64 %[t0], 0, v72, vcc\n\t"
      "v_cndmask_b32_e64 v92, 0, v72, %[sb]\n\t"
      "v_cndmask_b32_e64 v93, 0, v72, %[sc]\n\t"
      "v_add_f32 %[px], %[px], %[t0]\n\t"
      "v_pk_add_f32 v[68:69], v[68:69], v[92:93]\n\t"
      "v_add_u32 %[scale], -1, %[scale]\n\t"
      "v_mov_b32 %[h], %[tcm]\n\t"                            // h = tc_max
      "v_mov_b32 %[tmax], %[t3]\n"                              // t_max = tv_max
      "LnoD%=:\n\t"
      // ---- ADVANCE (svotrace.comp:329-339)
      "s_mov_b64 exec, %[sa]\n\t"
      "s_cbranch_execz LnoA%=\n\t"
      SVO_COUNT("c4", "c5", "exec")
      "v_mov_b32 %[tmin], %[tcm]\n\t"                            // t_min = tc_max
      "v_sub_f32 %[px], %[px], %[t0]\n\t"
      "v_pk_add_f32 v[68:69], v[68:69], v[92:93] neg_lo:[0,1] neg_hi:[0,1]\n\t"
      "v_bitop3_b32 %[t2], %[t2], %[idx], %[idx] bitop3:0x30\n\t"   // step & ~idx: an axis stepped out of the lower half
      "v_cmp_ne_u32 vcc, 0, %[t2]\n\t"                      // left the parent: POP
      "s_mov_b64 exec, vcc\n\t"
      "s_cbranch_execz LnoA%=\n\t"
      SVO_COUNT("c6", "c7", "exec")
      // ---- POP (svotrace.comp:341-366)
      "v_add_f32 %[t0], %[px], %[t0]\n\t"                       // position before the step (exact)
      "v_pk_add_f32 v[92:93], v[68:69], v[92:93]\n\t"
      "v_xor_b32 %[t0], %[t0], %[px]\n\t"
      "v_xor_b32 %[t1], v92, v68\n\t"
      "v_bitop3_b32 %[t0], %[t0], v93, v69 bitop3:0xf6\n\t"   // a | (b ^ c)
      "v_or3_b32 %[t0], %[t0], %[t1], 1\n\t"                    // differing bits (| 1 keeps ffbh defined)
      "v_ffbh_u32 %[t0], %[t0]\n\t"
      "v_sub_u32 %[t2], 20, %[t0]\n\t"                        // scale - 11
      "v_xor_b32 %[scale], 31, %[t0]\n\t"                        // scale = 31 - leading zeros
      "v_min_u32 %[t1], 11, %[t2]\n\t"
      "v_lshl_add_u32 v72, %[scale], 23, %[kexp]\n\t"        // cell size = 2^(scale - 23)
      "v_lshl_add_u32 %[t0], %[t1], 9, %[lds8]\n\t"
      "v_lshl_add_u32 %[t1], %[t1], 7, %[lds2]\n\t"
      "ds_read2_b32 v[92:93], %[t0] offset1:1\n\t"
      "ds_read_u16 %[t1], %[t1]\n\t"
      "v_bfe_i32 %[t2], %[wr], %[t2], 1\n\t"                    // all ones if this ray pushed that level
      "v_lshlrev_b32_e64 %[t3], %[scale], -1\n\t"
      "v_mov_b32 %[h], 0\n\t"                              // h = 0
      "v_and_b32 %[px], %[px], %[t3]\n\t"                       // round the position to the cell
      "v_and_b32 v68, v68, %[t3]\n\t"
      "v_and_b32 v69, v69, %[t3]\n\t"
      "v_cmp_le_u32 vcc, 23, %[scale]\n\t"                     // left the octree: MISS
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_and_b32 %[pbase], %[t2], v92\n\t"
      "v_and_b32 %[tmax], %[t2], v93\n\t"
      "v_and_b32 %[pmask], %[t2], %[t1]\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cbranch_scc1 Lmiss%=\n"                         // out of line
      "LnoA%=:\n\t"
      "s_bcnt1_i32_b64 %[cnt], %[act]\n\t"
      "s_cmp_gt_i32 %[cnt], %[thresh]\n\t"
      "s_cbranch_scc1 Ltrip%=\n\t"
      "s_branch Lend%=\n"
      "Lcap%=:\n\t"                                       // iteration cap: status = ST_CAPPED, lane out of the loop
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 4\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "s_branch Lnocap%=\n"
      "Lmiss%=:\n\t"                                      // left the octree: status = ST_MISS
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 3\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_branch LnoA%=\n"
      "Lend%=:\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [px] "+v"(r.px), "+{v68}"(r.py), "+{v69}"(r.pz), [tmin] "+v"(r.t_min), [tmax] "+v"(r.t_max), "+{v72}"(r.sexp), [h] "+v"(r.h),
        [scale] "+v"(r.scale), [idx] "+v"(r.idx), [pbase] "+v"(r.pbase), [pmask] "+v"(r.pmask), [wr] "+v"(r.written),
        [iter] "+v"(r.iter), [lod] "+v"(r.lod_scale), [st] "+v"(status), [cptr] "+v"(r.cptr), [tag] "+v"(r.tag),
        "=&{v88}"(r.rlo), "=&{v89}"(r.rhi), [tcx] "=&v"(tcx), [tcm] "=&v"(tcm), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
        [t3] "=&v"(t3), [act] "+s"(act), [sv] "=&s"(sv), [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc), [sd] "=&s"(sd),
        [se] "=&s"(se), [sf] "=&s"(sf), [sg] "=&s"(sg), [sh] "=&s"(sh), [cnt] "=&s"(cnt)
#ifdef SVO_STAMPS
        , [c0] "+s"(c0), [c1] "+s"(c1), [c2] "+s"(c2), [c3] "+s"(c3), [c4] "+s"(c4), [c5] "+s"(c5), [c6] "+s"(c6), [c7] "+s"(c7)
#endif
      : [cx] "v"(r.cx), [bx] "v"(r.bx), [cyz] "v"(r.cyz), [byz] "v"(r.byz), [oct] "v"(r.octant), [k005] "s"(0.05f), [conem] "s"(cone_lanes),
        [lds8] "v"(lds8), [lds2] "v"(lds2), [rs] "s"(pool.rsrc), [k5555] "s"(0x5555u), [selcp] "s"(0x01020304u),
        [selmask] "s"(0x0c0c0102u), [zero] "s"(0u), [kexp] "s"(0x34000000u), [thresh] "s"(threshold)
      : "vcc", "scc", "memory", "v73", "v86", "v87", "v90", "v92", "v93");
#ifdef SVO_STAMPS
  mix[0] = c0; mix[1] = c1; mix[2] = c2; mix[3] = c3; mix[4] = c4; mix[5] = c5; mix[6] = c6; mix[7] = c7;
#endif
}

}  // namespace svo
