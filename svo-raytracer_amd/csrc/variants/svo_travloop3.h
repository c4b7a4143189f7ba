// svo_travloop3.h -- the descriptor walk's trips (svo_travloop2.h) with a SPARE RAY per lane swapped in inside the loop.
//
// What trav_loop2 leaves on the table (profiles/round5_experiments.txt, SVO_STAMPS histogram): between two rounds a lane
// whose ray has stopped waits -- 24.4 % of all lane-trips are such waits (mean 48.4 of 64 lanes traversing).  Here every
// lane carries, next to the ray it traverses, a slot of nine registers that holds either a spare ray, ready to start
// (its per-ray constants, made in the last round: the divisions and the origin products are the round's work), or the
// parked result of a ray that has stopped.  When SVO_SWAP_BATCH lanes have stopped whose slot holds a spare, the loop
// exchanges them in place: result -> slot, spare -> traversal registers, stack column zeroed, lane back in the active
// set; a round is only needed once enough lanes have run out of spares.  The trips themselves are trav_loop2's,
// instruction for instruction: the exchange costs nothing in a trip that does not need it (the per-trip exit test
// compares with a moving trigger instead of the constant threshold).
//
// Slot layouts (nine 32-bit words):
//   spare : cx cy cz bx by bz t_min t_max (octant | upper-half bits of the start position << 3)
//   parked: descriptor offset `self`, iter (biased by -(kMaxIter + 1), as inside the loop), scale | cs << 8 | status << 16,
//           px py pz, t_min, -, octant
// Lane sets (scalar, owned by the caller): spare / parked = what the slot holds (neither: empty); cpath = which of the
// lane's two path records belongs to the ray being traversed (the slot's ray owns the other); conem / scone = the
// traversing / the slot's ray is a cone (secondary) ray.
#pragma once
#include "../svo_travloop2.h"

namespace svo {

#ifndef SVO_SWAP_BATCH
#define SVO_SWAP_BATCH 6
#endif
#define SVO_STR2(x) #x
#define SVO_STR(x) SVO_STR2(x)

constexpr uint32_t kIterBias = 0u - (kMaxIter + 1u);

struct Slot {
  uint32_t w0, w1, w2, w3, w4, w5, w6, w7, w8;
};

// per-ray constants and state in the register layout of trav_loop3(): TravRegs2 with the y / z constants as scalars of
// their own (the exchange writes them)
struct TravRegs3 {
  float cx, cy, cz, bx, by, bz;
  uint32_t octant;
  float px, py, pz;
  float t_min, t_max, sexp, h;
  int scale;
  uint32_t cs, self, dlo, dhi, iter;
  int lod_scale;
};

// set-up part of the cast (svotrace.comp:221-260) into a slot: a spare ray (ST_ACTIVE), or -- for the all-NaN ray that the
// reference spins to the iteration cap (quirk Q7, trav_init_regs2) -- straight the parked result (ST_CAPPED, iter 1501)
__device__ __forceinline__ int slot_init(Slot &s, V3 o, V3 d, const float t_start) {
  if (all_nan(o) || all_nan(d)) {
    s.w0 = kDescRoot; s.w1 = (kMaxIter + 1u) + kIterBias; s.w2 = (uint32_t)(kMaxScale - 1) | ((uint32_t)ST_CAPPED << 16);
    s.w3 = s.w4 = s.w5 = __float_as_uint(1.0f); s.w6 = 0u; s.w7 = 0u; s.w8 = 0u;
    return ST_CAPPED;
  }
  if (__builtin_fabsf(d.x) < kEpsilon) d.x = kEpsilon * sign_g(d.x);
  if (__builtin_fabsf(d.y) < kEpsilon) d.y = kEpsilon * sign_g(d.y);
  if (__builtin_fabsf(d.z) < kEpsilon) d.z = kEpsilon * sign_g(d.z);
  const float cx = 1.0f / -__builtin_fabsf(d.x);
  const float cy = 1.0f / -__builtin_fabsf(d.y);
  const float cz = 1.0f / -__builtin_fabsf(d.z);
  float bx = cx * o.x, by = cy * o.y, bz = cz * o.z;
  uint32_t oct = 0;
  if (d.x > 0.0f) { oct ^= 1u; bx = 3.0f * cx - bx; }
  if (d.y > 0.0f) { oct ^= 2u; by = 3.0f * cy - by; }
  if (d.z > 0.0f) { oct ^= 4u; bz = 3.0f * cz - bz; }
  float t_min = vmax3(2.0f * cx - bx, 2.0f * cy - by, 2.0f * cz - bz);
  const float t_max = vmin3(cx - bx, cy - by, cz - bz);
  t_min = vmax(t_min, 0.0f);
  t_min = vmax(t_min, t_start);   // beam pre-pass: the walk starts further along the same ray
  if (1.5f * cx - bx > t_min) oct |= 8u;
  if (1.5f * cy - by > t_min) oct |= 16u;
  if (1.5f * cz - bz > t_min) oct |= 32u;
  s.w0 = __float_as_uint(cx); s.w1 = __float_as_uint(cy); s.w2 = __float_as_uint(cz);
  s.w3 = __float_as_uint(bx); s.w4 = __float_as_uint(by); s.w5 = __float_as_uint(bz);
  s.w6 = __float_as_uint(t_min); s.w7 = __float_as_uint(t_max); s.w8 = oct;
  return ST_ACTIVE;
}

// the parked result of the ray in `t` (t.iter not biased: outside the loop)
__device__ __forceinline__ void slot_park(Slot &s, const TravRegs3 &t, const int status) {
  s.w0 = t.self; s.w1 = t.iter + kIterBias; s.w2 = ((uint32_t)t.scale & 0xffu) | (t.cs << 8) | ((uint32_t)status << 16);
  s.w3 = __float_as_uint(t.px); s.w4 = __float_as_uint(t.py); s.w5 = __float_as_uint(t.pz);
  s.w6 = __float_as_uint(t.t_min); s.w7 = 0u; s.w8 = t.octant;
}

// the spare ray of a slot into the traversal registers (what the loop's exchange does in assembly)
__device__ __forceinline__ void slot_start(const Slot &s, TravRegs3 &t, const uint2 rootd) {
  t.cx = __uint_as_float(s.w0); t.cy = __uint_as_float(s.w1); t.cz = __uint_as_float(s.w2);
  t.bx = __uint_as_float(s.w3); t.by = __uint_as_float(s.w4); t.bz = __uint_as_float(s.w5);
  t.t_min = __uint_as_float(s.w6); t.t_max = __uint_as_float(s.w7); t.h = t.t_max;
  t.octant = s.w8 & 7u;
  t.px = (s.w8 & 8u) ? 1.5f : 1.0f; t.py = (s.w8 & 16u) ? 1.5f : 1.0f; t.pz = (s.w8 & 32u) ? 1.5f : 1.0f;
  t.scale = kMaxScale - 1; t.sexp = 0.5f;
  t.self = kDescRoot; t.dlo = rootd.x; t.dhi = rootd.y;
  t.iter = 0; t.cs = 0; t.lod_scale = kMaxScale - kMaxDepth;
}

// result part of the cast (svotrace.comp:371-431) from a parked slot
__device__ __forceinline__ Cast slot_result(const BufPool &pool, const DescTab &tab, const Slot &s) {
  const int scale = (int)(s.w2 & 0xffu);
  const float sexp = __uint_as_float((uint32_t)(scale - kMaxScale + 127) << 23);
  return cast_result2(pool, tab, (int)(s.w2 >> 16), s.w0, (s.w2 >> 8) & 0xffu, s.w8 & 7u, s.w1 - kIterBias, __uint_as_float(s.w6), sexp,
                      scale, __uint_as_float(s.w3), __uint_as_float(s.w4), __uint_as_float(s.w5));
}

// Run trips until at most `threshold` lanes are still traversing and no stopped lane has a spare left.
// act: the lanes with status == ST_ACTIVE.  Lanes that stop with a spare in their slot go on with it (their result is then
// parked in the slot, `parked` / `spare` / `cpath` / `conem` / `scone` follow); lanes that stop without one get their status
// (ST_HIT / ST_MISS / ST_CAPPED) and wait, r.self / r.cs naming the parent state and child slot they stopped on.
__device__ __forceinline__ void trav_loop3(const DescTab &tab, WaveStack2 &stk, const uint32_t lane, TravRegs3 &r, int &status, Slot &s,
                                           unsigned long long act, const int threshold, unsigned long long &conem,
                                           unsigned long long &scone, unsigned long long &spare, unsigned long long &parked,
                                           unsigned long long &cpath, const uint2 rootd, uint32_t *mix = nullptr) {
  const uint32_t lds8 = lds_offset(&stk.pm[lane]);
  const uint32_t ldsb = lds8 - (uint32_t)kStackBase * 512u;   // + scale * 512 = the entry of that scale (svo_travloop2.h)
  unsigned long long sv, sa, sb, sc, sd, se, sf, sg, sh, sp, sm, sx, sq, sw;
  int cnt, trig;
#ifdef SVO_STAMPS
#define SVO_RFL(i) (uint32_t) __builtin_amdgcn_readfirstlane((int)mix[i])
  uint32_t c0 = SVO_RFL(0), c1 = SVO_RFL(1), c2 = SVO_RFL(2), c3 = SVO_RFL(3), c4 = SVO_RFL(4), c5 = SVO_RFL(5), c6 = SVO_RFL(6), c7 = SVO_RFL(7);
#undef SVO_RFL
  uint32_t hist = mix[8];   // lane L: trips of this wave that ran with exactly L lanes traversing
#else
  (void)mix;
#endif
  uint32_t t0, t1, t2, t3, bit;
  float tcx, tcm;
  r.iter += kIterBias;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_mov_b64 %[sm], %[act]\n\t"
      "s_mov_b64 %[sq], 0\n\t"
      "s_bcnt1_i32_b64 %[cnt], %[act]\n\t"
      "s_sub_i32 %[trig], %[cnt], " SVO_STR(SVO_SWAP_BATCH) "\n\t"
      "s_max_i32 %[trig], %[trig], %[thresh]\n"
      "Ltrip%=:\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      SVO_COUNT("c0", "c1", "exec")
      SVO_HIST
      // ---- child slot (bit `scale` of the three position components), iteration cap (svotrace.comp:263-266)
      "v_bfe_u32 %[t0], %[px], %[scale], 1\n\t"
      "v_bfe_u32 %[t1], v56, %[scale], 1\n\t"
      "v_bfe_u32 %[t2], v57, %[scale], 1\n\t"
      "v_lshl_or_b32 %[t0], %[t1], 1, %[t0]\n\t"
      "v_lshl_or_b32 %[t0], %[t2], 2, %[t0]\n\t"                  // idx = x | y << 1 | z << 2
      "v_xor_b32 %[cs], %[t0], %[oct]\n\t"                        // cs = idx ^ octant
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
      "v_cndmask_b32_e64 %[lod], %[lod], 12, vcc\n\t"
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
      "s_waitcnt vmcnt(0)\n\t"
      // (this loop keeps scale and child slot unbiased, unlike trav_loop2: one shift more per trip)
      "v_lshlrev_b32_e64 %[bit], 2, %[cs]\n\t"
      "v_lshrrev_b32 %[bit], %[bit], v65\n\t"          // the child's nibble: 0 empty, 1 not empty, 8 | rank with a child block
      "v_and_b32 %[bit], 15, %[bit]\n\t"
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
      "s_and_saveexec_b64 %[sb], vcc\n\t"
      "v_lshl_add_u32 v63, %[scale], 9, %[ldsb]\n\t"
      "ds_write2_b32 v63, %[self], %[tmax] offset1:1\n\t" // {parent state, t_max}
      "s_mov_b64 exec, %[sd]\n\t"
      "v_lshl_add_u32 %[self], %[bit], 3, v64\n\t"        // the child's descriptor: (first - 64) + 8 * (8 | rank)
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
      // POP detection (svotrace.comp:341) from the differing bits of the position before and after the step (svo_travloop2.h)
      "v_xor_b32 %[t0], %[t0], %[px]\n\t"
      "v_bitop3_b32 %[t0], %[t0], %[t1], v56 bitop3:0xf6\n\t"   // a | (b ^ c)
      "v_bitop3_b32 %[t0], %[t0], %[t2], v57 bitop3:0xf6\n\t"
      "v_lshrrev_b32 %[t1], %[scale], %[t0]\n\t"
      "v_cmp_lt_u32_e64 %[sp], 1, %[t1]\n\t"
      "s_mov_b64 exec, %[sp]\n\t"                         // left the parent: POP
      "s_cbranch_execz LnoA%=\n\t"
      SVO_COUNT("c6", "c7", "exec")
      // ---- POP (svotrace.comp:341-366)
      "v_ffbh_u32 %[t0], %[t0]\n\t"                       // (the differing bits of a POP lane are never zero: d >> scale > 1)
      "v_xor_b32 %[scale], 31, %[t0]\n\t"
      "v_lshl_add_u32 v58, %[scale], 23, %[kexp]\n\t"
      "v_lshl_add_u32 %[t0], %[scale], 9, %[ldsb]\n\t"
      "ds_read_b32 %[self], %[t0]\n\t"                    // a level this ray never pushed holds the zeros it started on:
      "ds_read_b32 %[tmax], %[t0] offset:4\n\t"           // state (0, 0) = descriptor 0, t_max 0
      "v_lshlrev_b32_e64 %[t3], %[scale], -1\n\t"
      "v_mov_b32 %[h], 0\n\t"                             // h = 0
      "v_and_b32 %[px], %[px], %[t3]\n\t"                 // round the position to the cell
      "v_and_b32 v56, v56, %[t3]\n\t"
      "v_and_b32 v57, v57, %[t3]\n\t"
      "v_cmp_le_u32 vcc, 23, %[scale]\n\t"                // left the octree: MISS
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_lg_u64 vcc, 0\n\t"
      "s_cbranch_scc1 Lmiss%=\n"                          // out of line
      "LnoA%=:\n\t"
      // ---- the descriptors of the lanes that changed their parent state
      "s_or_b64 exec, %[sd], %[sp]\n\t"
      "buffer_load_dwordx2 v[64:65], %[self], %[rsd], 0 offen\n\t"
      // ---- go on while more lanes traverse than the moving trigger says (SVO_SWAP_BATCH lanes below the count after the last exchange)
      "s_bcnt1_i32_b64 %[cnt], %[act]\n\t"
      "s_cmp_gt_i32 %[cnt], %[trig]\n\t"
      "s_cbranch_scc1 Ltrip%=\n\t"
      // ---- some lanes have stopped: those with a spare in their slot go on with it
      "s_andn2_b64 %[sf], %[sm], %[act]\n\t"              // stopped on a HIT since they (re)started ...
      "s_or_b64 %[sx], %[sf], %[sq]\n\t"                  // ... or left the octree / the budget
      "s_and_b64 %[sw], %[sx], %[spare]\n\t"
      "s_cmp_lg_u64 %[sw], 0\n\t"
      "s_cbranch_scc0 Lafter%=\n\t"
      "s_waitcnt vmcnt(0)\n\t"                            // (v[64:65] are written below, for other lanes than the load's)
      "s_and_b64 exec, %[sf], %[sw]\n\t"
      "v_mov_b32 %[st], 2\n\t"                            // ST_HIT of the lanes that stopped on one (the others have theirs)
      "s_mov_b64 exec, %[sw]\n\t"
      "v_swap_b32 %[cx], %[s0]\n\t"                       // the spare's constants in, the stopped ray's out (dead)
      "v_swap_b32 %[cy], %[s1]\n\t"
      "v_swap_b32 %[cz], %[s2]\n\t"
      "v_swap_b32 %[bx], %[s3]\n\t"
      "v_swap_b32 %[by], %[s4]\n\t"
      "v_swap_b32 %[bz], %[s5]\n\t"
      "v_mov_b32 %[s0], %[self]\n\t"                      // the result parked: parent state, iter, scale | cs << 8 | status << 16,
      "v_mov_b32 %[s1], %[iter]\n\t"
      "v_lshl_or_b32 %[t0], %[cs], 8, %[scale]\n\t"
      "v_lshl_or_b32 %[s2], %[st], 16, %[t0]\n\t"
      "v_mov_b32 %[s3], %[px]\n\t"                        // position, t_min, octant
      "v_mov_b32 %[s4], v56\n\t"
      "v_mov_b32 %[s5], v57\n\t"
      "v_swap_b32 %[tmin], %[s6]\n\t"
      "v_mov_b32 %[tmax], %[s7]\n\t"
      "v_swap_b32 %[oct], %[s8]\n\t"
      "v_mov_b32 %[h], %[tmax]\n\t"                       // the fresh ray's state (svotrace.comp:245-260): h = t_max, root cell, ...
      "v_mov_b32 %[scale], 22\n\t"
      "v_mov_b32 v58, 0.5\n\t"
      "v_mov_b32 %[self], 8\n\t"
      "v_mov_b32 v64, %[rootlo]\n\t"
      "v_mov_b32 v65, %[roothi]\n\t"
      "v_mov_b32 %[iter], %[bias]\n\t"
      "v_mov_b32 %[lod], 10\n\t"
      "v_mov_b32 %[st], 1\n\t"                            // ST_ACTIVE
      "v_bfe_u32 %[t0], %[oct], 3, 1\n\t"                 // position 1.0 / 1.5 per axis: bit 22 of the float
      "v_bfe_u32 %[t1], %[oct], 4, 1\n\t"
      "v_bfe_u32 %[t2], %[oct], 5, 1\n\t"
      "v_lshl_or_b32 %[px], %[t0], 22, %[kone]\n\t"
      "v_lshl_or_b32 v56, %[t1], 22, %[kone]\n\t"
      "v_lshl_or_b32 v57, %[t2], 22, %[kone]\n\t"
      "v_and_b32 %[oct], 7, %[oct]\n\t"
      "v_mov_b32 v62, 0\n\t"                              // a new ray starts on a zeroed stack column (svotrace.comp:227)
      "v_mov_b32 v63, 0\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:0 offset1:1\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:2 offset1:3\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:4 offset1:5\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:6 offset1:7\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:8 offset1:9\n\t"
      "ds_write2st64_b64 %[lds8], v[62:63], v[62:63] offset0:10 offset1:11\n\t"
      "s_andn2_b64 %[spare], %[spare], %[sw]\n\t"         // lane sets: the slot holds a result, the records change hands,
      "s_or_b64 %[parked], %[parked], %[sw]\n\t"
      "s_xor_b64 %[cpath], %[cpath], %[sw]\n\t"
      "s_andn2_b64 %[conem], %[conem], %[sw]\n\t"         // the cone flag travels with the ray,
      "s_and_b64 %[sf], %[scone], %[sw]\n\t"
      "s_or_b64 %[conem], %[conem], %[sf]\n\t"
      "s_andn2_b64 %[scone], %[scone], %[sw]\n\t"
      "s_or_b64 %[act], %[act], %[sw]\n\t"                // and the lane traverses again
      "s_or_b64 %[sm], %[sm], %[sw]\n\t"
      "s_andn2_b64 %[sq], %[sq], %[sw]\n"
      "Lafter%=:\n\t"
      "s_bcnt1_i32_b64 %[cnt], %[act]\n\t"
      "s_sub_i32 %[trig], %[cnt], " SVO_STR(SVO_SWAP_BATCH) "\n\t"
      "s_max_i32 %[trig], %[trig], %[thresh]\n\t"
      "s_cmp_gt_i32 %[cnt], %[thresh]\n\t"
      "s_cbranch_scc1 Ltrip%=\n\t"
      "s_branch Lend%=\n"
      "Lcap%=:\n\t"                                       // iteration cap: status = ST_CAPPED, lane out of the loop
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 4\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_andn2_b64 %[sm], %[sm], vcc\n\t"
      "s_or_b64 %[sq], %[sq], vcc\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "s_branch Lnocap%=\n"
      "Lmiss%=:\n\t"                                      // left the octree: status = ST_MISS
      "s_mov_b64 exec, vcc\n\t"
      "v_mov_b32 %[st], 3\n\t"
      "s_andn2_b64 %[act], %[act], vcc\n\t"
      "s_andn2_b64 %[sm], %[sm], vcc\n\t"
      "s_or_b64 %[sq], %[sq], vcc\n\t"
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
        [sp] "=&s"(sp), [sm] "=&s"(sm), [sx] "=&s"(sx), [sq] "=&s"(sq), [sw] "=&s"(sw), [cnt] "=&s"(cnt), [trig] "=&s"(trig),
        [cx] "+v"(r.cx), [cy] "+v"(r.cy), [cz] "+v"(r.cz), [bx] "+v"(r.bx), [by] "+v"(r.by), [bz] "+v"(r.bz), [oct] "+v"(r.octant),
        [s0] "+v"(s.w0), [s1] "+v"(s.w1), [s2] "+v"(s.w2), [s3] "+v"(s.w3), [s4] "+v"(s.w4), [s5] "+v"(s.w5), [s6] "+v"(s.w6),
        [s7] "+v"(s.w7), [s8] "+v"(s.w8),
        [conem] "+s"(conem), [scone] "+s"(scone), [spare] "+s"(spare), [parked] "+s"(parked), [cpath] "+s"(cpath)
#ifdef SVO_STAMPS
        , [c0] "+s"(c0), [c1] "+s"(c1), [c2] "+s"(c2), [c3] "+s"(c3), [c4] "+s"(c4), [c5] "+s"(c5), [c6] "+s"(c6), [c7] "+s"(c7), [hist] "+v"(hist)
#endif
      : [k005] "s"(0.05f), [lds8] "v"(lds8), [ldsb] "v"(ldsb), [rsd] "s"(tab.rsrc), [kexp] "s"(0x34000000u),
        [thresh] "s"(threshold), [rootlo] "s"(rootd.x), [roothi] "s"(rootd.y), [bias] "s"(kIterBias), [kone] "s"(0x3f800000u)
#ifdef SVO_STAMPS
        , [lane] "v"(lane)
#endif
      : "vcc", "scc", "memory", "v60", "v61", "v62", "v63");
  r.iter -= kIterBias;
#ifdef SVO_STAMPS
  mix[0] = c0; mix[1] = c1; mix[2] = c2; mix[3] = c3; mix[4] = c4; mix[5] = c5; mix[6] = c6; mix[7] = c7; mix[8] = hist;
#endif
}

}  // namespace svo
