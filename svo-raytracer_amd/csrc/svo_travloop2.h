// svo_travloop2.h -- the traversal trips over the interior-descriptor table, in gfx950 assembly.
//
// trav_loop2() runs trav_step2() (svo_trav2.h) on the wave's active lanes until no more than `threshold` of them are
// still traversing.  Against the byte walk's loop (svo_travloop.h) a trip loses: the child-offset arithmetic (two
// shifts, a bit-op, an and, three popcounts, two mads: it is only needed for the hit pointer, once per cast, after the
// loop), the two record loads of EVERY trip (whether the child is empty and whether it has a child block is a nibble of
// the parent's descriptor: one v_bfe and two v_cmp on a register), the cp / tag-mask extraction, the 16-bit tag-mask plane of
// the stack and the pushed-levels mask (a PUSH is one ds_write2_b32 of {descriptor offset, t_max}; a POP two ds_read_b32
// straight into the state registers: the caller zeroes a lane's stack column when it sets up a ray, so a level the ray
// never pushed reads as the reference's zero-initialised entry).  What it gains is one shift-add: the child's descriptor offset from
// that nibble (v_lshl_add_u32; until round 5's nibbles the rank among the parent's `has` bits: v_bfm, v_and, v_bcnt, v_lshl_add).
//   trips load only for lanes that DESCEND or POP: one aligned 8-byte descriptor (49 cycles of the texture path per
//   wave-level load against 2 x 34 for the two unaligned dwords of a record, tools/calib_td.hip), issued at the end
//   of the trip for both lane sets together, waited for at the top of the next trip behind everything that does not
//   need it (child slot, exit distances, the advance step).
// Arithmetic, operand order and rounding are those of trav_step2() = trav_step() = svotrace.comp:262-369.
//
// Pinned registers: v[56:57] py,pz; v58 cell size; v[60:61] tcy,tcz; v[62:63] temporaries; v[64:65] the parent's
// descriptor {first child descriptor - 64, a nibble per child} (svo_derive.hip.h).
//
// Which instructions: gfx950 issues a wave64 v_add / v_sub / v_mul / v_fma / logic op / v_mov / right shift / v_bitop3
// every ~2.35 cycles per SIMD and every compare, v_cndmask, min / max, bit-field, shift-and-add, count or packed-f32
// instruction every ~4.3 (tools/calib_valu2.hip), with the scalar unit running beside; the kernel is bound by exactly
// that.  Hence: per-axis position updates as v_add / v_sub under an EXEC mask (the lane set of the axis' compare)
// instead of v_cndmask-selected increments; plain instead of packed multiplies; the iteration cap from the carry of a
// biased counter; ST_HIT written once behind the loop; no pushed-levels mask.
#pragma once
#include "svo_trav2.h"
#include "svo_travloop.h"

namespace svo {

// SVO_DESC_LOAD_EARLY=0: one load per trip, at its end, for the lanes that descended or popped; 1: the descending lanes'
// load leaves as soon as their child's descriptor offset is known (20 instructions earlier), the popping lanes' at the end
#ifndef SVO_DESC_LOAD_EARLY
#define SVO_DESC_LOAD_EARLY 0
#endif
#if SVO_DESC_LOAD_EARLY
#define SVO_DESC_LOAD_D "buffer_load_dwordx2 v[64:65], %[self], %[rsd], 0 offen\n\t"
#define SVO_DESC_LOAD_M                                                      \
  "s_mov_b64 exec, %[sp]\n\t"                                                \
  "buffer_load_dwordx2 v[64:65], %[self], %[rsd], 0 offen\n\t"
#else
#define SVO_DESC_LOAD_D
#define SVO_DESC_LOAD_M                                                      \
  "s_or_b64 exec, %[sd], %[sp]\n\t"                                          \
  "buffer_load_dwordx2 v[64:65], %[self], %[rsd], 0 offen\n\t"
#endif

// The child slot times four: the scale register holds scale - 2 and a position has no bits below `scale`, so three bits from
// scale - 2 on are the component's bit << 2.  (Right shifts by scale - 2 / - 3 / - 4 merged with two v_bitop3 bit selects -- 8
// fast-class instructions instead of 5 slow + 1 -- measured +0.25 %, inside the spread: this part of the trip waits for the
// descriptor anyway.  profiles/round5_experiments.txt)
#define SVO_CHILD_SLOT                                                        \
  "v_bfe_u32 %[t0], %[px], %[scale], 3\n\t"                                   \
  "v_bfe_u32 %[t1], v56, %[scale], 3\n\t"                                     \
  "v_bfe_u32 %[t2], v57, %[scale], 3\n\t"                                     \
  "v_lshl_or_b32 %[t0], %[t1], 1, %[t0]\n\t"                                  \
  "v_lshl_or_b32 %[t0], %[t2], 2, %[t0]\n\t" /* 4 * idx */                    \
  "v_xor_b32 %[cs], %[t0], %[oct]\n\t"       /* 4 * (idx ^ octant): where the child's nibble starts */
// the child's nibble of the parent's descriptor: one bit-field extract (slow class), or shift + and (two fast-class instructions:
// -0.35 %, inside the spread)
#ifndef SVO_NIB_BFE
#define SVO_NIB_BFE 1
#endif
#if SVO_NIB_BFE
#define SVO_NIBBLE "v_bfe_u32 %[bit], v65, %[cs], 4\n\t"
#else
#define SVO_NIBBLE                            \
  "v_lshrrev_b32 %[bit], %[cs], v65\n\t"      \
  "v_and_b32 %[bit], 15, %[bit]\n\t"
#endif

#ifdef SVO_STAMPS
#define SVO_HIST                                        \
  "s_mov_b64 exec, -1\n\t"                             \
  "v_cmp_eq_u32 vcc, %[cnt], %[lane]\n\t"              \
  "v_addc_co_u32 %[hist], vcc, 0, %[hist], vcc\n\t"    \
  "s_mov_b64 exec, %[act]\n\t"
// ... and a second histogram: lanes in the POP section of the trips that run it (round 6: sizing a deferred POP)
#define SVO_HIST_POP                                    \
  "s_mov_b64 exec, -1\n\t"                             \
  "v_cmp_eq_u32 vcc, %[cnt], %[lane]\n\t"              \
  "v_addc_co_u32 %[hist2], vcc, 0, %[hist2], vcc\n\t"  \
  "s_mov_b64 exec, %[sp]\n\t"
#else
#define SVO_HIST
#define SVO_HIST_POP
#endif

// Where a level's stack entry lives.  The descriptor table only exists for pools of at most 13 levels (svo_derive.hip.h), so
// this loop only ever pushes at scale 11..22 and the clamp of the record walk's loop (a deeper pool's pushes land on the last
// level, identically in all pipelines) is dead here: the entry of scale s is at lds8 + (s - 11) * 512 = s * 512 + (lds8 - 11 * 512),
// one shift-add on the scale instead of subtract, clamp, shift-add.  A POP that leaves the octree (scale 23: MISS) reads one row
// past the column -- a row of the same wave's array or past it, where LDS reads return zero -- and drops what it read.
// A PUSH below scale 11 (only from the phantom state behind a POP to a never-pushed level: no input produces one)
// forms an address below the lane's column = below LDS offset 0: the hardware drops the write, a POP there reads zero.  That
// needs the stack to be the kernel's ONLY __shared__ object, at offset 0: tests/test_kernel_isa.py checks the code object's LDS
// size (6 144 bytes, not a byte more) for every descriptor-walk kernel.
// SVO_STACK_CLAMP=1 builds the clamped form (A/B).
#ifndef SVO_STACK_CLAMP
#define SVO_STACK_CLAMP 0
#endif
#if SVO_STACK_CLAMP
#define SVO_PUSH_ADDR                                   \
  "v_add_u32 %[t1], -9, %[scale]\n\t"                   \
  "v_min_u32 %[t1], 11, %[t1]\n\t"                      \
  "s_and_saveexec_b64 %[sb], vcc\n\t"                   \
  "v_lshl_add_u32 v63, %[t1], 9, %[lds8]\n\t"
#define SVO_POP_ADDR                                    \
  "v_sub_u32 %[t2], 20, %[t0]\n\t"                      \
  "v_sub_u32 %[scale], 29, %[t0]\n\t"                   \
  "v_min_u32 %[t1], 11, %[t2]\n\t"                      \
  "v_lshl_add_u32 v58, %[scale], 23, %[kexp]\n\t"       \
  "v_lshl_add_u32 %[t0], %[t1], 9, %[lds8]\n\t"
#else
#define SVO_PUSH_ADDR                                   \
  "s_and_saveexec_b64 %[sb], vcc\n\t"                   \
  "v_lshl_add_u32 v63, %[scale], 9, %[ldsb]\n\t"
#define SVO_POP_ADDR                                    \
  "v_sub_u32 %[scale], 29, %[t0]\n\t"                   \
  "v_lshl_add_u32 v58, %[scale], 23, %[kexp]\n\t"       \
  "v_lshl_add_u32 %[t0], %[scale], 9, %[ldsb]\n\t"
#endif

// per-ray constants and state in the register layout of trav_loop2()
struct TravRegs2 {
  float cx, bx;
  f32x2 cyz, byz;
  uint32_t octant;    // the mirror mask of svotrace.comp:239-252, times four (see cs)
  float px;
  float py, pz;   // (two floats, not a pair: a vector member tied to v[56:57] kept px / py / pz in scratch memory around every loop)
  float t_min, t_max, sexp, h;
  int scale;          // scale - 2: three bits of a position component from there are its child-slot bit times four
  uint32_t cs;        // 4 * child slot of the last trip (index ^ octant) = where the child's nibble starts in the parent's
                      // descriptor: the slot a stopped lane stopped on
  uint32_t self;      // byte offset of the parent state's descriptor
  uint32_t dlo, dhi;  // that descriptor
  uint32_t written, iter;
  int lod_scale;      // minus two, like scale
};
constexpr int kScaleBias = 2;

// set-up part of the cast (svotrace.comp:221-260); `rootd` = the root's descriptor, fetched once per wave
__device__ __forceinline__ int trav_init_regs2(const uint2 rootd, TravRegs2 &t, V3 o, V3 d, const bool cone,
                                               const float t_start = 0.0f) {
  (void)cone;   // which lanes carry cone (secondary) rays is a lane set the caller passes to trav_loop2
  t.iter = 0; t.cs = 0; t.written = 0; t.lod_scale = kMaxScale - kMaxDepth - kScaleBias;
  t.scale = kMaxScale - 1 - kScaleBias; t.sexp = 0.5f;
  t.self = kDescRoot; t.dlo = rootd.x; t.dhi = rootd.y;
  if (all_nan(o) || all_nan(d)) {  // quirk Q7: the reference spins to the cap, iter = 1501
    t.iter = kMaxIter + 1u; t.t_min = 0.0f; t.t_max = 0.0f; t.h = 0.0f; t.octant = 0;
    t.cx = t.cyz.x = t.cyz.y = t.bx = t.byz.x = t.byz.y = 0.0f; t.px = t.py = t.pz = 1.0f;
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
  if (d.x > 0.0f) { t.octant ^= 4u; t.bx = 3.0f * t.cx - t.bx; }
  if (d.y > 0.0f) { t.octant ^= 8u; t.byz.x = 3.0f * t.cyz.x - t.byz.x; }
  if (d.z > 0.0f) { t.octant ^= 16u; t.byz.y = 3.0f * t.cyz.y - t.byz.y; }
  t.t_min = vmax3(2.0f * t.cx - t.bx, 2.0f * t.cyz.x - t.byz.x, 2.0f * t.cyz.y - t.byz.y);
  t.t_max = vmin3(t.cx - t.bx, t.cyz.x - t.byz.x, t.cyz.y - t.byz.y);
  t.t_min = vmax(t.t_min, 0.0f);
  t.t_min = vmax(t.t_min, t_start);   // beam pre-pass: the walk starts further along the same ray
  t.h = t.t_max;
  t.px = 1.0f; t.py = 1.0f; t.pz = 1.0f;
  if (1.5f * t.cx - t.bx > t.t_min) t.px = 1.5f;
  if (1.5f * t.cyz.x - t.byz.x > t.t_min) t.py = 1.5f;
  if (1.5f * t.cyz.y - t.byz.y > t.t_min) t.pz = 1.5f;
  return ST_ACTIVE;
}

__device__ __forceinline__ Cast trav_result_regs2(const BufPool &pool, const DescTab &tab, const TravRegs2 &t, int status) {
  return cast_result2(pool, tab, status, t.self, t.cs >> 2, t.octant >> 2, t.iter, t.t_min, t.sexp, t.scale + kScaleBias, t.px, t.py, t.pz);
}

// Run trips until at most `threshold` lanes of `act` (the lanes with status == ST_ACTIVE) are still traversing.
// Lanes that stop get their status (ST_HIT / ST_MISS / ST_CAPPED); r.self / r.cs then name the parent state and the
// child slot they stopped on.  `cone_lanes`: the lanes whose ray is a cone (secondary) ray -- they drop to LOD 11 once
// t_min > 0.05 (svotrace.comp:275-277).
__device__ __forceinline__ void trav_loop2(const DescTab &tab, WaveStack2 &stk, const uint32_t lane, TravRegs2 &r, int &status,
                                           unsigned long long act, const int threshold, const unsigned long long cone_lanes,
                                           uint32_t *mix = nullptr) {
  const uint32_t lds8 = lds_offset(&stk.pm[lane]);
  const uint32_t ldsb = lds8 - ((uint32_t)kStackBase - 2u) * 512u;   // + (scale - 2) * 512 = the entry of that scale
  unsigned long long sv, sa, sb, sc, sd, se, sf, sg, sh, sp, sm, sx;
  int cnt;
#ifdef SVO_STAMPS
#define SVO_RFL(i) (uint32_t) __builtin_amdgcn_readfirstlane((int)mix[i])
  uint32_t c0 = SVO_RFL(0), c1 = SVO_RFL(1), c2 = SVO_RFL(2), c3 = SVO_RFL(3), c4 = SVO_RFL(4), c5 = SVO_RFL(5), c6 = SVO_RFL(6), c7 = SVO_RFL(7);
#undef SVO_RFL
  uint32_t hist = mix[8];   // lane L: trips of this wave that ran with exactly L lanes traversing
  uint32_t hist2 = mix[9];  // lane L: trips of this wave whose POP section ran for exactly L lanes
#else
  (void)mix;
#endif
  uint32_t t0, t1, t2, t3, bit;
  float tcx, tcm;
  r.iter += 0u - (kMaxIter + 1u);
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_mov_b64 %[sm], %[act]\n"
      "Ltrip%=:\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      SVO_COUNT("c0", "c1", "exec")
      SVO_HIST
      // ---- child slot (bit `scale` of the three position components), iteration cap (svotrace.comp:263-266)
      SVO_CHILD_SLOT
      "v_add_co_u32 %[iter], vcc, 1, %[iter]\n\t"                 // iter++ on a counter biased by 2^32 - 1501: the carry is "iter > 1500"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cbranch_scc1 Lcap%=\n"                                   // rare, out of line
      "Lnocap%=:\n\t"
      // ---- exit distances of the current cell (svotrace.comp:268-269)
      "v_mul_f32 %[tcx], %[px], %[cx]\n\t"
      "v_mul_f32 v60, v56, %[cy]\n\t"
      "v_mul_f32 v61, v57, %[cz]\n\t"
      "v_cmp_lt_f32 vcc, %[k005], %[tmin]\n\t"                    // t_min > 0.05 ...
      "v_sub_f32 %[tcx], %[tcx], %[bx]\n\t"
      "v_sub_f32 v60, v60, %[by]\n\t"
      "v_sub_f32 v61, v61, %[bz]\n\t"
      "s_and_b64 vcc, vcc, %[conem]\n\t"                          // ... on a cone (secondary) ray: LOD 11 from here on (sticky)
      "v_cmp_le_f32_e64 %[sa], %[tmin], %[tmax]\n\t"              // t_min <= t_max
      "v_min3_f32 %[tcm], %[tcx], v60, v61\n\t"                   // tc_max
      "v_cndmask_b32_e64 %[lod], %[lod], 10, vcc\n\t"                // (12 - 2)
      "v_min_f32 %[t3], %[tmax], %[tcm]\n\t"                      // tv_max
      "v_cmp_eq_u32_e64 %[sb], %[scale], %[lod]\n\t"              // at the LOD scale
      "v_cmp_le_f32_e64 %[sc], %[tmin], %[t3]\n\t"                // t_min <= tv_max
      "s_or_b64 %[se], %[sb], %[sc]\n\t"
      "s_and_b64 %[se], %[se], %[sa]\n\t"                         // in range & (at LOD | inside): hits or descends if not empty
      "s_andn2_b64 %[sd], %[sc], %[sb]\n\t"
      "s_and_b64 %[sd], %[sd], %[sa]\n\t"                         // in range & !at LOD & inside: descends if it has a child block
      // the ADVANCE step of every active lane, while the descriptor of lanes that descended / popped is in flight
      "v_cmp_le_f32_e64 %[sx], %[tcx], %[tcm]\n\t"                // the axes whose exit distance is the cell's
      "v_cmp_le_f32_e64 %[sg], v60, %[tcm]\n\t"
      "v_cmp_le_f32_e64 %[sh], v61, %[tcm]\n\t"
      // an axis that steps out of the lower half leaves the parent: POP (svotrace.comp:341; idx & step after the flip =
      // step & ~idx before it) -- on lane sets, no step mask in a register
      "s_waitcnt vmcnt(0)\n\t"
      SVO_NIBBLE                                          // the child's nibble: 0 empty, 1 not empty, 8 | rank with a child block
      "v_cmp_ne_u32_e64 %[sa], 0, %[bit]\n\t"           // child not empty
      "v_cmp_lt_u32 vcc, 7, %[bit]\n\t"                 // child has a child block
      // lane sets
      "s_and_b64 %[sd], %[sd], vcc\n\t"
      "s_and_b64 %[sd], %[sd], %[sa]\n\t"                 // DESCEND = not empty & in range & !at LOD & inside & child block
      "s_and_b64 %[se], %[se], %[sa]\n\t"                 // not empty & in range & (at LOD | inside)
      "s_andn2_b64 %[sa], exec, %[se]\n\t"                // ADVANCE = the rest
      "s_mov_b64 %[sp], 0\n\t"                            // (no advancing lane: no POP)
      "s_andn2_b64 %[se], %[se], %[sd]\n\t"               // HIT = not empty & in range & (at LOD | (inside & no child block))
      "s_andn2_b64 %[act], %[act], %[se]\n\t"
      // ---- DESCEND (svotrace.comp:291-327)
      "s_mov_b64 exec, %[sd]\n\t"
      "s_cbranch_execz LnoD%=\n\t"
      SVO_COUNT("c2", "c3", "exec")
      "v_cmp_lt_f32 vcc, %[tcm], %[h]\n\t"                // tc_max < h: PUSH
      "v_mul_f32 v58, 0.5, v58\n\t"                       // half
      SVO_PUSH_ADDR
      "ds_write2_b32 v63, %[self], %[tmax] offset1:1\n\t" // {parent state, t_max}
      "s_mov_b64 exec, %[sd]\n\t"
      "v_lshl_add_u32 %[self], %[bit], 3, v64\n\t"        // the child's descriptor: (first - 64) + 8 * (8 | rank)
      SVO_DESC_LOAD_D
      "v_mul_f32 %[t0], %[cx], v58\n\t"
      "v_mul_f32 v62, %[cy], v58\n\t"
      "v_mul_f32 v63, %[cz], v58\n\t"
      "v_add_f32 %[t0], %[t0], %[tcx]\n\t"                // centre distances
      "v_add_f32 v62, v62, v60\n\t"
      "v_add_f32 v63, v63, v61\n\t"
      "v_cmp_gt_f32 vcc, %[t0], %[tmin]\n\t"
      "v_cmp_gt_f32_e64 %[sb], v62, %[tmin]\n\t"
      "v_cmp_gt_f32_e64 %[sc], v63, %[tmin]\n\t"
      "v_add_u32 %[scale], -1, %[scale]\n\t"
      "v_mov_b32 %[h], %[tcm]\n\t"                        // h = tc_max
      "v_mov_b32 %[tmax], %[t3]\n\t"                     // t_max = tv_max
      "s_and_b64 exec, %[sd], vcc\n\t"                   // the upper half on an axis: position += half, under the compare's lane set
      "v_add_f32 %[px], %[px], v58\n\t"
      "s_and_b64 exec, %[sd], %[sb]\n\t"
      "v_add_f32 v56, v56, v58\n\t"
      "s_and_b64 exec, %[sd], %[sc]\n\t"
      "v_add_f32 v57, v57, v58\n"
      "LnoD%=:\n\t"
      // ---- ADVANCE (svotrace.comp:329-339)
      "s_mov_b64 exec, %[sa]\n\t"
      "s_cbranch_execz LnoA%=\n\t"
      SVO_COUNT("c4", "c5", "exec")
      "v_mov_b32 %[tmin], %[tcm]\n\t"                     // t_min = tc_max
      "v_mov_b32 %[t0], %[px]\n\t"                        // the position before the step, for the POP's differing bits
      "v_mov_b32 %[t1], v56\n\t"
      "v_mov_b32 %[t2], v57\n\t"
      "s_and_b64 exec, %[sa], %[sx]\n\t"                  // step: position -= cell size on the axes that leave the cell
      "v_sub_f32 %[px], %[px], v58\n\t"
      "s_and_b64 exec, %[sa], %[sg]\n\t"
      "v_sub_f32 v56, v56, v58\n\t"
      "s_and_b64 exec, %[sa], %[sh]\n\t"
      "v_sub_f32 v57, v57, v58\n\t"
      "s_mov_b64 exec, %[sa]\n\t"
      // POP detection (svotrace.comp:341) from the differing bits of the position before and after the step: a step out of the
      // upper half clears bit `scale` of the coordinate and nothing else, a step out of the lower half borrows into the bits
      // above -- with d = OR over the axes of (old ^ new): no step 0, step inside the parent d >> scale == 1, POP d >> scale > 1
      // (d is what the POP section needs anyway; round 3 compared "upper half on this axis" per axis in the common part and
      // combined lane sets with five scalar operations: same speed, eight instructions more)
      "v_xor_b32 %[t0], %[t0], %[px]\n\t"
      "v_bitop3_b32 %[t0], %[t0], %[t1], v56 bitop3:0xf6\n\t"   // a | (b ^ c)
      "v_bitop3_b32 %[t0], %[t0], %[t2], v57 bitop3:0xf6\n\t"
      "v_lshrrev_b32 %[t1], %[scale], %[t0]\n\t"        // (by scale - 2)
      "v_cmp_lt_u32_e64 %[sp], 7, %[t1]\n\t"
      "s_mov_b64 exec, %[sp]\n\t"                         // left the parent: POP
      "s_cbranch_execz LnoA%=\n\t"
      SVO_COUNT("c6", "c7", "exec")
      SVO_HIST_POP
      // ---- POP (svotrace.comp:341-366)
      "v_ffbh_u32 %[t0], %[t0]\n\t"                       // (the differing bits of a POP lane are never zero: d >> scale > 1)
      SVO_POP_ADDR
      "ds_read_b32 %[self], %[t0]\n\t"                    // a level this ray never pushed holds the zeros it started on:
      "ds_read_b32 %[tmax], %[t0] offset:4\n\t"           // state (0, 0) = descriptor 0, t_max 0
      "v_lshlrev_b32_e64 %[t3], %[scale], -4\n\t"       // ~0 << scale
      "v_mov_b32 %[h], 0\n\t"                             // h = 0
      "v_and_b32 %[px], %[px], %[t3]\n\t"                 // round the position to the cell
      "v_and_b32 v56, v56, %[t3]\n\t"
      "v_and_b32 v57, v57, %[t3]\n\t"
      "v_cmp_le_u32 vcc, 21, %[scale]\n\t"                // left the octree (scale 23): MISS
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cbranch_scc1 Lmiss%=\n"                          // out of line
      "LnoA%=:\n\t"
      // ---- the descriptors of the lanes that changed their parent state
      SVO_DESC_LOAD_M
      "s_bcnt1_i32_b64 %[cnt], %[act]\n\t"
      "s_cmp_gt_i32 %[cnt], %[thresh]\n\t"
      "s_cbranch_scc1 Ltrip%=\n\t"
      "s_branch Lend%=\n"
      "Lcap%=:\n\t"                                       // iteration cap: status = ST_CAPPED, lane out of the loop
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 4\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_andn2_b64 %[sm], %[sm], vcc\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "s_branch Lnocap%=\n"
      "Lmiss%=:\n\t"                                      // left the octree: status = ST_MISS
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 3\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_andn2_b64 %[sm], %[sm], vcc\n\t"
      "s_andn2_b64 %[sp], %[sp], vcc\n\t"
      "s_branch LnoA%=\n"
      "Lend%=:\n\t"
      "s_andn2_b64 exec, %[sm], %[act]\n\t"             // the lanes that stopped without leaving the octree or the budget: ST_HIT
      "v_mov_b32 %[st], 2\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [px] "+v"(r.px), "+{v56}"(r.py), "+{v57}"(r.pz), [tmin] "+v"(r.t_min), [tmax] "+v"(r.t_max), "+{v58}"(r.sexp), [h] "+v"(r.h),
        [scale] "+v"(r.scale), [cs] "+v"(r.cs), [self] "+v"(r.self), "+{v64}"(r.dlo), "+{v65}"(r.dhi),
        [iter] "+v"(r.iter), [lod] "+v"(r.lod_scale), [st] "+v"(status), [tcx] "=&v"(tcx), [tcm] "=&v"(tcm), [t0] "=&v"(t0),
        [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [bit] "=&v"(bit), [act] "+s"(act), [sv] "=&s"(sv), [sa] "=&s"(sa),
        [sb] "=&s"(sb), [sc] "=&s"(sc), [sd] "=&s"(sd), [se] "=&s"(se), [sf] "=&s"(sf), [sg] "=&s"(sg), [sh] "=&s"(sh),
        [sp] "=&s"(sp), [sm] "=&s"(sm), [sx] "=&s"(sx), [cnt] "=&s"(cnt)
#ifdef SVO_STAMPS
        , [c0] "+s"(c0), [c1] "+s"(c1), [c2] "+s"(c2), [c3] "+s"(c3), [c4] "+s"(c4), [c5] "+s"(c5), [c6] "+s"(c6), [c7] "+s"(c7), [hist] "+v"(hist), [hist2] "+v"(hist2)
#endif
      : [cx] "v"(r.cx), [bx] "v"(r.bx),
        [cy] "v"(r.cyz.x), [cz] "v"(r.cyz.y), [by] "v"(r.byz.x), [bz] "v"(r.byz.y), [oct] "v"(r.octant), [k005] "s"(0.05f), [conem] "s"(cone_lanes),
        [lds8] "v"(lds8), [ldsb] "v"(ldsb), [rsd] "s"(tab.rsrc), [kexp] "s"(0x35000000u), [thresh] "s"(threshold)
#ifdef SVO_STAMPS
        , [lane] "v"(lane)
#endif
      : "vcc", "scc", "memory", "v60", "v61", "v62", "v63");
  r.iter += kMaxIter + 1u;
#ifdef SVO_STAMPS
  mix[0] = c0; mix[1] = c1; mix[2] = c2; mix[3] = c3; mix[4] = c4; mix[5] = c5; mix[6] = c6; mix[7] = c7; mix[8] = hist; mix[9] = hist2;
#endif
}

}  // namespace svo
