// svo_persistent.hip.h -- pipeline 1: persistent waves, whole paths, lane refill.
//
// Each wave keeps 64 traversals in flight.  A lane owns one PATH at a time: its primary
// ray, then -- regenerated in place from the hit -- its bounce / shadow ray(s); when the
// path ends the lane stores the pixel and takes the next pixel of its XCD's screen band.
// Finished lanes are handled in rounds: once at most 9/16 of the lanes that were traversing
// at the start of a burst of trips are still traversing (the default; svo_set_tuning), the
// stopped lanes shade together (ballot), regenerate or retire, and the
// freed lanes are refilled with one atomic per wave (ballot + prefix count).  Compared
// with one-thread-per-pixel (pipeline 0 = the reference's decomposition) no lane waits for
// the slowest ray of its tile; compared with stage-per-kernel wavefront tracing
// (pipeline 2) the bounce ray starts where its primary just warmed the caches, no path
// state travels through HBM and the frame has one tail instead of one per stage.
//
// Work distribution: the frame is cut into 8 strips of whole tile rows, one per XCD (each XCD
// has its own 4 MiB L2; neighbouring tiles walk the same subtrees), and a strip is walked column
// by column so that the pixels an XCD has in flight form a compact block.  A wave reads its XCD
// id from the hardware register and draws from that band's counter, then steals from the
// other bands.  Consecutive launches walk the columns in opposite directions (a frame starts
// where the previous one is finishing).  Placement is only a locality hint: any wave may render
// any pixel.
#pragma once
#include "svo_fused.hip.h"
#include "svo_trav.h"
#include "svo_travloop.h"
#include "svo_travloop2.h"

#include <algorithm>
#include <cstdlib>

namespace svo {

// One 128-byte line per band counter.  With the 8 counters in one line every refill atomic of every CU
// serialised on it and -- vmcnt being in-order on gfx950 -- stalled the record load behind it: in-kernel
// stamps (SVO_STAMPS build) showed the load wait growing from 420 to 2000 cycles per iteration as rounds
// became more frequent; with separate lines it stays below 500.
constexpr int kHeadStride = 32;
#ifdef SVO_STAMPS
constexpr int kHeadWords = 8 * kHeadStride + 32 + 64 + 64 + 16;  // + diagnostics words + the histograms of lanes traversing / popping per trip
                                                                   // + 8 x u64: cycles of a round by part
#else
constexpr int kHeadWords = 8 * kHeadStride + 32;  // + diagnostics words
#endif

struct PersistArgs {
  const uint8_t *pool;
  Frame f;
  uint32_t *color;
  float *depth;
  uint4 *hits;
  float *facc;       // spp > 1: colour sums (3 planes of npix)
  size_t npix;
  uint32_t *heads;   // 8 band counters (pixel slots drawn so far)
  int tiles_per_band;
  int rows_per_band;   // tile rows per XCD band (SVO_BAND_COLMAJOR)
  int reverse;         // walk the columns right to left (every other launch)
  int sample;        // sample index of this launch (one launch per sample), 0 when the launch carries all samples
  int fold;          // samples per pixel carried by this launch (see persist_launch); 1 = one launch per sample
  int group;         // fold > 1: tiles per group of a band's walk (sample by sample inside a group)
  int thresh_num;    // a round starts once active lanes <= thresh_num/16 of those active at its start
  // reciprocals (udiv_magic) of a band's tile rows and of its tile slots per frame, for a full band and for the last one
  uint32_t mg_rows[2], mg_tpf[2];
  // interior-descriptor table (svo_derive.hip.h); only read by the kDerived kernels
  const uint2 *desc;
  const uint2 *aux;
  uint32_t desc_count;
  // per-frame cameras and frame numbers of a batch (svo_ring_submit_cams), f.batch entries; only read by the kCams kernels
  const FrameVar *fvar;
  // path records of the spare-ray kernel (variants/svo_persist2.hip.h): kRecWaveWords words per persistent wave of the launch
  uint32_t *prec;
  int spare;   // 1 = launch the spare-ray kernel
  // row / column tables of the launch's frames (rc_table_kernel; only read by the kTab kernels): per frame rc_stride floats =
  // W column records {ra, u} then H row records {a.x, a.y, a.z, rb, b.x, b.y, b.z, 0}
  const float *rc;
  uint32_t rc_stride;
  const float4 *ntab;   // unit normals by 16-bit code (svo_trav2.h), or null
};

// n / d by multiply-high with m = ceil(2^32 / d) = (2^32 + e) / d, 0 <= e < d (made on the host, udiv_magic): with
// n = q d + r the product is q 2^32 + q e + r (2^32 + e) / d, so the high word is q exactly while e (n + d) < 2^32.
// The host checks that against the largest n the launch can form (tile slots of a band x frames x samples: a 4K batch of
// 17 frames, or 1080p with 16 samples x 2 frames, is already past it) and passes m = 0 otherwise = a real division.
__device__ __forceinline__ uint32_t udiv_by(uint32_t n, uint32_t d, uint32_t m) {
  return d <= 1u ? n : (m == 0u ? n / d : __umulhi(n, m));
}
inline uint32_t udiv_magic(uint32_t d, uint64_t n_max) {
  if (d <= 1u) return 0u;
  const uint32_t m = 0xffffffffu / d + 1u;
  const uint64_t e = (uint64_t)m * d - (1ull << 32);
  return e * (n_max + d) < (1ull << 32) ? m : 0u;
}

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

// `sk`: sample index (bits 0..15) and frame of the batch (bits 16..23) of this path when the launch carries all samples of
// its pixels (a.fold > 1), else 0
__device__ __forceinline__ void persist_emit(const PersistArgs &a, uint32_t pix, uint32_t sk, int px, int py, V3 col, float depth) {
  const uint32_t smp = sk & 0xffffu;
#ifdef SVO_NO_STORES  // timing experiment only: keep the values alive, skip the stores
  asm volatile("" ::"v"(col.x), "v"(col.y), "v"(col.z), "v"(depth), "v"(pix));
  return;
#endif
  if (a.f.spp <= 1 && !a.f.progressive) {   // the live shader's case: straight to rgba8
    if (px < 10 && py < 10) col = a.f.dword0 == 0u ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
    a.color[pix] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xff000000u;
  } else if (a.fold > 1) {
    // every sample in a slot of its own, [frame][tile][sample][channel][pixel of the tile]: the 64 values of a tile's
    // sample and channel share two cache lines, and the kernel that adds the samples up in order reads them coalesced
    const int ry = py - a.f.y0;
    const uint32_t ty = a.f.row_step == 1 ? (uint32_t)(ry >> 3) : (uint32_t)(ry >> 3) / (uint32_t)a.f.row_step;
    const uint32_t tile = ty * (uint32_t)a.f.tiles_x + (uint32_t)(px >> 3), l = (uint32_t)(((ry & 7) << 3) | (px & 7));
    float *p = a.facc + (((size_t)(sk >> 16) * (size_t)a.f.ntiles + tile) * (size_t)a.fold + smp) * 192 + l;
    p[0] = col.x; p[64] = col.y; p[128] = col.z;
  } else {   // several samples and / or cross-frame accumulation: float sums, finished by persist_resolve_kernel
    float *fx = a.facc + pix, *fy = a.facc + a.npix + pix, *fz = a.facc + 2 * a.npix + pix;
    if (a.sample == 0) { *fx = 0.0f + col.x; *fy = 0.0f + col.y; *fz = 0.0f + col.z; }
    else { *fx = *fx + col.x; *fy = *fy + col.y; *fz = *fz + col.z; }
  }
  // the depth image is the first sample's; of a progressive sequence (every "sample" a frame of its own) the last frame's
  if (a.sample == 0 && smp == (a.f.seq > 1 ? (uint32_t)(a.fold - 1) : 0u)) a.depth[pix] = depth;
}

// SVO_ASM_LOOP=1 (default): the trips run in trav_loop() / trav_loop2() (gfx950 assembly);
// SVO_ASM_LOOP=0: hipcc's translation of trav_step() / trav_step2() -- same results, kept for A/B runs and as the
// readable form.
#ifndef SVO_ASM_LOOP
#define SVO_ASM_LOOP 1
#endif

// How a cast reads the octree.  ByteWalk: the reference's records, one fetched per iteration (svo_trav.h).
// DescWalk: the interior-descriptor table derived from them (svo_derive.hip.h, svo_trav2.h); the pool itself is only
// read for the record of the node a cast ends on.
struct ByteWalk {
#if SVO_ASM_LOOP
  typedef TravRegs State;
#else
  typedef Trav State;
#endif
  typedef WaveStack Stack;
  BufPool pool;
  uint64_t root;
  __device__ __forceinline__ void setup(const PersistArgs &a) {
    pool = make_bufpool(a.pool, a.f.pool_len);
    root = load_record(pool, 0u);
  }
  __device__ __forceinline__ int init(State &t, V3 o, V3 d, bool cone, float t_start = 0.0f) const {
#if SVO_ASM_LOOP
    return trav_init_regs(root, t, o, d, cone, t_start);
#else
    return trav_init(root, t, o, d, cone, t_start);
#endif
  }
  __device__ __forceinline__ void fresh_stack(Stack &, uint32_t) const {}   // the record walk keeps a "pushed" mask per ray
  __device__ __forceinline__ Cast result(const State &t, int status) const {
#if SVO_ASM_LOOP
    return trav_result_regs(pool, t, status);
#else
    return trav_result(t, status);
#endif
  }
  // trips until at most `threshold` of the lanes in `act` are still traversing
  __device__ __forceinline__ void run(Stack &stk, uint32_t lane, State &t, int &status, unsigned long long act, int threshold,
                                      unsigned long long cone_lanes, uint32_t *mix) const {
#if SVO_ASM_LOOP
    trav_loop(pool, stk, lane, t, status, act, threshold, cone_lanes, mix);
#else
    (void)act; (void)cone_lanes; (void)mix;
    for (;;) {
#ifdef SVO_STAMPS
      unsigned long long st_load = 0;
      if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t, st_load);
#else
      if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t);
#endif
      if (__builtin_popcountll(__ballot(status == ST_ACTIVE)) <= threshold) break;
    }
#endif
  }
};

struct DescWalk {
#if SVO_ASM_LOOP
  typedef TravRegs2 State;
#else
  typedef Trav2 State;
#endif
  typedef WaveStack2 Stack;
  BufPool pool;
  DescTab tab;
  uint2 rootd;
  __device__ __forceinline__ void setup(const PersistArgs &a) {
    pool = make_bufpool(a.pool, a.f.pool_len);
    tab = make_desctab(a.desc, a.aux, a.desc_count, a.ntab);
    const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(tab.rsrc, (int)kDescRoot, 0, 0);
    rootd = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)r.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)r.y));
  }
  __device__ __forceinline__ int init(State &t, V3 o, V3 d, bool cone, float t_start = 0.0f) const {
#if SVO_ASM_LOOP
    return trav_init_regs2(rootd, t, o, d, cone, t_start);
#else
    return trav_init2(rootd, t, o, d, cone, t_start);
#endif
  }
  // a new ray starts on a zeroed stack column, like the reference's zero-initialised stack[] (svotrace.comp:227): a pop to a
  // level the ray never pushed then reads {descriptor 0, t_max 0} by itself, and the loop keeps no "pushed" mask
  __device__ __forceinline__ void fresh_stack(Stack &stk, uint32_t lane) const {
#if SVO_ASM_LOOP && !defined(SVO_TIMING_ONLY_NO_STACK_CLEAR)
    // (always in the assembly build: trav_loop2's POP reads its entry without a pushed-levels mask.  Round 3's closing
    // commit lost the define that guarded these stores and shipped a kernel without them; the switch above exists to
    // price them in an A/B and builds a WRONG kernel; tests/test_kernel_isa.py checks the shipped one.)
#pragma unroll
    for (int lv = 0; lv < kStackLevels; ++lv) stk.pm[lane + 64u * (uint32_t)lv] = make_uint2(0u, 0u);
#else
    (void)stk; (void)lane;
#endif
  }
  __device__ __forceinline__ Cast result(const State &t, int status) const {
#if SVO_ASM_LOOP
    return trav_result_regs2(pool, tab, t, status);
#else
    return trav_result2(pool, tab, t, status);
#endif
  }
  __device__ __forceinline__ void run(Stack &stk, uint32_t lane, State &t, int &status, unsigned long long act, int threshold,
                                      unsigned long long cone_lanes, uint32_t *mix) const {
#if SVO_ASM_LOOP
    trav_loop2(tab, stk, lane, t, status, act, threshold, cone_lanes, mix);
#else
    (void)act; (void)cone_lanes; (void)mix;
    for (;;) {
      if (status == ST_ACTIVE) status = trav_step2(tab, stk, lane, t);
      if (__builtin_popcountll(__ballot(status == ST_ACTIVE)) <= threshold) break;
    }
#endif
  }
};

#ifndef SVO_BAND_COLMAJOR
#define SVO_BAND_COLMAJOR 1
#endif
#ifndef SVO_SERPENTINE
#define SVO_SERPENTINE 1
#endif
#ifndef SVO_STEAL_NOW
#define SVO_STEAL_NOW 0
#endif
#ifndef SVO_DRAIN_NUM
#define SVO_DRAIN_NUM 12
#endif
#ifndef SVO_PERSIST_WAVES_PER_SIMD
#define SVO_PERSIST_WAVES_PER_SIMD 5
#endif
#ifndef SVO_DERIVED_WAVES_PER_SIMD
#define SVO_DERIVED_WAVES_PER_SIMD 6
#endif
template <class Walk> struct WalkWaves { static constexpr int value = SVO_PERSIST_WAVES_PER_SIMD; };
template <> struct WalkWaves<DescWalk> { static constexpr int value = SVO_DERIVED_WAVES_PER_SIMD; };

// kCams: the frames of the batch carry their own camera and frameNumber (a.fvar) -- a kernel of its own, so that the
// static-camera kernel keeps the camera in SGPRs from its kernel arguments
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
// kTab: the row and column parts of a new pixel's pure functions come from tables made once per launch (rc_table_kernel) instead
// of being evaluated in every round for a third of the lanes -- of the primary direction (svotrace.comp:662-675) the two
// divisions of the pixel centre and the six mixes along the left and right edge vectors (functions of the row) and the
// horizontal fraction (of the column); of the pixel's random number (:26-29, :486) the two inner sin / fract chains (one a
// function of the column and frameNumber, one of the row and frameNumber).  What stays in the round: three mixes, the
// normalisation, the outer chain -- the same operations on the same values, so the same bits.
template <int kMode, class Walk, bool kCams = false, bool kTab = false>
__global__ __launch_bounds__(64, WalkWaves<Walk>::value) void persist_kernel(const PersistArgs a) {
  __shared__ typename Walk::Stack stk;
  const uint32_t lane = threadIdx.x;
  const Frame &f = a.f;
  Walk walk;
  walk.setup(a);
  const V3 cam_o = mk(f.cam[0], f.cam[1], f.cam[2]);
  const V3 sun2 = normalize3(mk(0.5f, 0.5f, 0.5f));

  typename Walk::State t;
  int status = ST_IDLE;
  uint32_t pix = 0, seg = 0;   // seg: path segment in the low byte; a.fold > 1: sample index in bits 8..23, frame of the batch above
  int px = 0, py = 0;
  // path state that outlives a cast
  V3 d = mk(0.f, 0.f, 0.f), mask = mk(1.f, 1.f, 1.f), accum = mk(0.f, 0.f, 0.f), normal = mk(0.f, 0.f, 0.f);
  float r = 0.0f, depth = 0.0f;
  uint32_t value = 0;

  uint32_t band = xcc_id();
  int bands_left = 8;

#ifdef SVO_STAMPS
  unsigned long long st_round = 0, st_trav = 0, st_nround = 0, st_ntrip = 0, st_shade = 0, st_load = 0, st_t0 = __builtin_readcyclecounter();
  const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole device
  unsigned long long st_dry = 0;
  uint32_t st_mix[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // trips / lanes, by section (trav_loop); [8], [9]: per-lane histogram words (trav_loop2)
  // a round by part (round 6): the cast's result (hit-record fetch + decode), the shading arithmetic, the pixel store, the refill
  // (counter draw + the new pixels' primary direction / random number), the ray set-up (three IEEE divisions + the stack column)
  unsigned long long st_part[5] = {0, 0, 0, 0, 0}, st_lanes_shaded = 0, st_lanes_refilled = 0, st_lanes_init = 0;
#endif
  for (;;) {
    // ---------------- finished lanes: shade, then regenerate the next ray in place or retire
    // A lane leaves this block with a pixel to store (emit) and / or a new ray to set up (ninit); the store and the
    // set-up run once per round, behind the refill, for all the lanes that need them together -- not once per branch
    // (three copies of the store, two of the set-up with its three IEEE divisions, each for a handful of lanes).
    bool emit = false, ninit = false, icone = false;
    V3 ecol = mk(0.f, 0.f, 0.f), io = mk(0.f, 0.f, 0.f);
    float edepth = 0.0f, its = 0.0f;
#ifdef SVO_STAMPS
    unsigned long long st_r1 = 0, st_r2 = 0;
    st_lanes_shaded += (unsigned long long)__builtin_popcountll(__ballot(status >= ST_HIT));
#endif
    if (status >= ST_HIT) {
      const Cast c = walk.result(t, status);
#ifdef SVO_STAMPS
      asm volatile("" ::"v"(c.t), "v"(c.pointer), "v"(c.normal.x), "v"(c.voxel_pos.x), "v"(c.value) : "memory");   // the record is here
      st_r1 = __builtin_readcyclecounter();
#endif
      status = ST_IDLE;
      const uint32_t segn = seg & 0xffu, smp = seg >> 8;   // smp: sample | frame of the batch << 16
      if (segn == 0u && (smp & 0xffffu) == 0u && f.write_hits && a.sample == 0) {
        uint4 h;
        h.x = c.hit ? c.pointer : 0u;
        h.y = c.hit ? ((c.raw & 0xffffu) | ((c.value & 0xffu) << 16) | ((c.level & 0xffu) << 24)) : 0u;
        h.z = c.iter;
        h.w = c.hit ? __float_as_uint(c.t) : 0u;
        if (kMode == 4) h = make_uint4(0u, 0u, 0u, 0u);   // trace() casts nothing in modes >= 4 (svotrace.comp:643-646)
        a.hits[pix] = h;
      }
      emit = true;
      if (kMode == 0) {
        if (segn == 0u && !c.hit) {
          const V3 s = sky_colour(d);
          ecol = mk(0.0f + s.x, 0.0f + s.y, 0.0f + s.z);
        } else {
          V3 vpos = mk(0.f, 0.f, 0.f);
          if (c.hit) { normal = c.normal; value = c.value; vpos = c.voxel_pos; }
          const V3 nd = scatter(d, normal, r, ((f.mirror_mask >> (value & 31u)) & 1u) != 0u);
          if (c.hit) {
            const V3 mc = material_colour(value, mk(vpos.x - 1.0f, vpos.y - 1.0f, vpos.z - 1.0f));
            depth = c.t;
            accum = mk(accum.x + mask.x * 0.0f, accum.y + mask.y * 0.0f, accum.z + mask.z * 0.0f);
            mask = mk(mask.x * mc.x, mask.y * mc.y, mask.z * mc.z);
            const float k = dot3(nd, normal);
            mask = mk(mask.x * k, mask.y * k, mask.z * k);
            if ((int)segn + 1 >= f.bounces) {
              ecol = accum; edepth = depth;
            } else {
              d = nd;
              seg++;
              emit = false; ninit = true; icone = true; io = vpos;
              status = ST_ACTIVE;   // not idle: keeps its pixel (the set-up behind the refill gives the real status)
            }
          } else {
            const V3 sun = normalize3(mk(1.0f, 1.0f, 1.0f));
            const float diff = acos_pinned(dot3(nd, sun));
            if (diff < 0.4f) accum = mk(accum.x + mask.x * 7.0f, accum.y + mask.y * 7.0f, accum.z + mask.z * 7.0f);
            accum = mk(accum.x + mask.x * 1.0f, accum.y + mask.y * 1.0f, accum.z + mask.z * 1.0f);
            ecol = accum;
          }
        }
      } else if (kMode == 1) {
        if (c.hit) { const float g = 0.005f * (float)c.iter; ecol = mk(g, g, g); }
        else if (c.capped) ecol = mk(0.3f, 0.3f, 0.6f);
        else { const float g = 0.01f * (float)c.iter; ecol = mk(g, g, g); }
        edepth = c.hit ? c.t : 0.0f;
      } else if (kMode == 2) {
        if (segn == 0u) {
          if (c.hit) {
            V3 mc = material_colour(c.value, kMode2OtherMaterial);
            const float k = (c.level >= 10u ? dot3(c.normal, sun2) : dot3(mk(0.f, 1.0f, 0.f), sun2)) * 0.1f;
            mc = mk(mc.x + k, mc.y + k, mc.z + k);
            const float dist = c.t + 0.0f;
            const float lg = exp2_pinned(dist * (-0.5f * 2.0f * 1.44269504f));
            const float lb = exp2_pinned(dist * (-0.5f * 4.0f * 1.44269504f));
            const float lr = exp2_pinned(dist * (-0.5f * 1.0f * 1.44269504f));
            mc.x = lr * mc.x + (1.0f - lr) * 1.0f;
            mc.y = lg * mc.y + (1.0f - lg) * 1.0f;
            mc.z = lb * mc.z + (1.0f - lb) * 1.0f;
            mask = mc;
            depth = c.t;
            seg = (seg & ~0xffu) | 1u;
            d = sun2;   // (the primary direction is not needed any more)
            emit = false; ninit = true; icone = false; io = c.voxel_pos;
            status = ST_ACTIVE;
          } else {
            ecol = sky_colour(d);
          }
        } else {
          V3 mc = mask;
          if (c.hit && c.t > c.scale_exp2 * 1.73205080757f) {
            mc = mk(mc.x - 0.2f, mc.y - 0.2f, mc.z - 0.2f);
          } else if (c.iter > 260u) {
            const float pen = (0.05f * (float)c.iter) / 100.0f;
            mc = mk(mc.x - pen, mc.y - pen, mc.z - pen);
          }
          ecol = mc; edepth = depth;
        }
      } else if (kMode == 3) {
        if (c.hit) { ecol = mk(c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f, c.normal.z * 0.5f + 0.5f); edepth = c.t; }
      }
    }
#ifdef SVO_STAMPS
    asm volatile("" ::"v"(ecol.x), "v"(ecol.y), "v"(ecol.z), "v"(edepth), "v"(io.x), "v"(d.x) : "memory");
    st_r2 = __builtin_readcyclecounter();
    {
      // st_r1 is per lane (taken under the branch): the wave's figure = the largest (the lanes that took the branch share one clock read)
      unsigned long long m = st_r1;
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(m, o, 64); m = x > m ? x : m; }
      if (m != 0) { st_part[0] += m - st_t0; st_part[1] += st_r2 - m; }
    }
#endif
    if (emit) persist_emit(a, pix, seg >> 8, px, py, ecol, edepth);

#ifdef SVO_STAMPS
    { const unsigned long long now = __builtin_readcyclecounter(); st_shade += now - st_t0; st_part[2] += now - st_r2; st_r2 = now; }
#endif
    // ---------------- refill idle lanes: ballot + prefix count, one atomic per wave
    // (a wave whose band is used up tries the next band in its next round; SVO_STEAL_NOW=1 tries it in the same round:
    // 1-4 % slower -- tools/history/r03_ab_drain.sh)
    uint32_t cam_frame = 0u; int cam_sample = 0; bool cam_fresh = false;   // kCams only
    while (bands_left > 0) {
      const unsigned long long idle = __ballot(status == ST_IDLE);
      if (idle == 0ull) break;
      {
        const uint32_t n = (uint32_t)__builtin_popcountll(idle);
        const int leader = __builtin_ctzll(idle);
#if SVO_BAND_COLMAJOR
        // a band = a strip of whole tile rows, walked column by column: the pixels in flight on an XCD form a compact
        // block of the screen (strip height x a few dozen tile columns) instead of two or three full-width tile rows
        const int first_row = (int)band * a.rows_per_band;
        int band_rows = f.tiles_y - first_row;
        band_rows = band_rows < 0 ? 0 : (band_rows > a.rows_per_band ? a.rows_per_band : band_rows);
        const uint32_t band_frame = (uint32_t)(band_rows * f.tiles_x) * 64u;   // pixel slots of the band in one frame
        const uint32_t band_total = band_frame * (uint32_t)(f.batch * a.fold);   // ... and over the launch's frames x samples
#else
        const int first_tile = (int)band * a.tiles_per_band;
        int band_tiles = f.ntiles - first_tile;
        band_tiles = band_tiles < 0 ? 0 : (band_tiles > a.tiles_per_band ? a.tiles_per_band : band_tiles);
        const uint32_t band_total = (uint32_t)band_tiles * 64u;
#endif
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(a.heads + band * kHeadStride, n);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);   // wave-uniform, and the compiler knows it
        const uint32_t slot =
            base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (status == ST_IDLE && slot < band_total) {
          const uint32_t l = slot & 63u;
#if SVO_BAND_COLMAJOR
          // frames of a batch follow one another inside the band: a wave that runs out of frame k goes on with k + 1
          // ; the samples of a pixel (a.fold > 1) follow one another tile by tile: tile 0 sample 0, 1, ..., tile 1 sample 0,
          // ... -- what is in flight at any time are a few tiles' samples, whose primary rays are the same and whose
          // slots in the sample buffer are neighbours
          // (the two divisions of every refill -- by the band's tile slots per frame and by its tile rows -- are
          // multiply-highs by reciprocals made on the host)
          const uint32_t per_frame = band_frame * (uint32_t)a.fold;
          const uint32_t tiles_pf = per_frame >> 6;   // tile slots of the band per frame
          const int full_band = band_rows == a.rows_per_band ? 0 : 1;
          const uint32_t fi = f.batch > 1 ? udiv_by(slot >> 6, tiles_pf, a.mg_tpf[full_band]) : 0u;
          const uint32_t q = (slot - fi * per_frame) >> 6;
          uint32_t si = 0u;
          int j = (int)q;
          if (a.fold > 1) {   // groups of a.group tiles: sample 0 of the group's tiles, sample 1 of them, ...
            const uint32_t tiles = band_frame >> 6, g = q / ((uint32_t)a.group * (uint32_t)a.fold);
            const uint32_t first = g * (uint32_t)a.group;
            const uint32_t gsize = tiles - first < (uint32_t)a.group ? tiles - first : (uint32_t)a.group;
            const uint32_t r = q - first * (uint32_t)a.fold;
            si = r / gsize;
            j = (int)(first + r % gsize);
          }
          int tile_x = (int)udiv_by((uint32_t)j, (uint32_t)band_rows, a.mg_rows[full_band]);
          const int tile_y = first_row + (j - tile_x * band_rows);
          if (a.reverse) tile_x = f.tiles_x - 1 - tile_x;   // serpentine: this frame ends where the next one starts
#else
          const int tile = first_tile + (int)(slot >> 6);
          const int tile_x = tile % f.tiles_x, tile_y = tile / f.tiles_x;
#endif
          px = tile_x * 8 + (int)(l & 7u);
          py = frame_gy(f, tile_y, (int)(l >> 3));
          if (px < f.width && py < f.y1 && py < f.height) {
#if SVO_BAND_COLMAJOR
            pix = fi * f.frame_stride + (uint32_t)frame_oy(f, tile_y, (int)(l >> 3)) * (uint32_t)f.width + (uint32_t)px;
#else
            pix = (uint32_t)frame_oy(f, tile_y, (int)(l >> 3)) * (uint32_t)f.width + (uint32_t)px;
#endif
            if (kTab) {
              const float *fr = a.rc + (size_t)fi * a.rc_stride;
              const float2 col = *(const float2 *)(fr + 2 * px);
              const float *rowp = fr + ((2 * f.width + 3) & ~3) + 8 * py;   // (row records start on a 16-byte boundary)
              const float4 ra4 = *(const float4 *)rowp, rb4 = *(const float4 *)(rowp + 4);
              d = normalize3(mk(mix_g(ra4.x, rb4.x, col.y), mix_g(ra4.y, rb4.y, col.y), mix_g(ra4.z, rb4.z, col.y)));
              if (kMode == 0) r = rand_of_dot(((float)px + col.x) * 12.9898f + ((float)py + ra4.w) * 78.233f);
            } else if (!kCams) d = primary_direction(f, px, py);
#if SVO_BAND_COLMAJOR
            seg = a.fold > 1 ? (si << 8) | (fi << 24) : 0u;
#else
            seg = 0u;
#endif
            mask = mk(1.f, 1.f, 1.f);
            accum = mk(0.f, 0.f, 0.f);
            normal = mk(0.f, 0.f, 0.f);
            value = 0u;
            depth = 0.0f;
#if SVO_BAND_COLMAJOR
            if (kMode == 0 && !kCams && !kTab) r = pixel_rand((float)px, (float)py, (float)(f.frame_number + (int)fi + a.sample + (int)si));
            if (kCams) { cam_frame = fi; cam_sample = a.sample + (int)si; cam_fresh = true; }
#else
            if (kMode == 0) r = pixel_rand((float)px, (float)py, (float)(f.frame_number + a.sample));
#endif
            ninit = true; icone = false; io = cam_o; its = beam_start(f, px, py);
            status = ST_ACTIVE;   // taken (the set-up below gives the real status)
          }
        }
        if (kCams) {
          // the camera of every refilled lane's frame: the slots of one draw are consecutive, so nearly always all lanes
          // are in one frame -- one 64-byte scalar load per distinct frame, the lanes of that frame take their ray from it
          unsigned long long todo = __ballot(cam_fresh);
          while (todo != 0ull) {
            const uint32_t fu = (uint32_t)__builtin_amdgcn_readlane((int)cam_frame, __builtin_ctzll(todo));
            const FrameVar *vp = a.fvar + fu;
            u32x16 raw;   // = *vp: cam[0..14], frame_number
            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(vp) : "memory");
            const bool mine = cam_fresh && cam_frame == fu;
            if (mine) {
              float cam[15];
#pragma unroll
              for (int i = 0; i < 15; i++) cam[i] = __uint_as_float(raw[i]);
              if (!kTab) d = primary_direction_cam(cam, f.width, f.height, px, py);
              io = mk(cam[0], cam[1], cam[2]);
              if (kMode == 0 && !kTab) r = pixel_rand((float)px, (float)py, (float)((int)raw[15] + cam_sample));
              cam_fresh = false;
            }
            todo &= ~__ballot(mine);
          }
        }
        if (base + n >= band_total) {  // this band is used up: move on (work stealing)
          band = (band + 1u) & 7u;
          bands_left--;
#ifdef SVO_STAMPS
          if (bands_left == 0) st_dry = __builtin_amdgcn_s_memrealtime();
#endif
#if !SVO_STEAL_NOW
          break;
#endif
        } else {
          break;
        }
      }
    }
#ifdef SVO_STAMPS
    asm volatile("" ::"v"(d.x), "v"(r), "v"(pix) : "memory");
    { const unsigned long long now = __builtin_readcyclecounter(); st_part[3] += now - st_r2; st_r2 = now; }
    st_lanes_init += (unsigned long long)__builtin_popcountll(__ballot(ninit));
    st_lanes_refilled += (unsigned long long)__builtin_popcountll(__ballot(ninit && !icone && (seg & 0xffu) == 0u));
#endif
    // ---------------- set up the new rays: regenerated bounce / shadow rays and refilled primaries together
    if (ninit) {
      status = walk.init(t, io, d, icone, its);
      walk.fresh_stack(stk, lane);
      if (kMode == 4) status = ST_MISS;   // no cast: straight to the (black) pixel
    }
    if (__ballot(status != ST_IDLE) == 0ull) {
      if (bands_left > 0) continue;
      break;
    }

#ifdef SVO_STAMPS
    { const unsigned long long now = __builtin_readcyclecounter(); st_part[4] += now - st_r2; st_round += now - st_t0; st_t0 = now; st_nround++; }
#endif
    // ---------------- traverse until enough lanes have stopped to make a round worthwhile
    const int active0 = __builtin_popcountll(__ballot(status == ST_ACTIVE));
    // (a fixed "N lanes free" trigger was tried instead of the proportional one: 3 % slower at its best setting)
    // once every band is used up a round only serves the lanes that go on with another segment of their path: they should
    // not wait for the longest cast of the wave (SVO_DRAIN_NUM/16 of the active lanes may still be traversing; 0 = wait for all)
    const int drained = (active0 * SVO_DRAIN_NUM) / 16 < active0 - 1 ? (active0 * SVO_DRAIN_NUM) / 16 : (active0 > 0 ? active0 - 1 : 0);
    const int threshold = __builtin_amdgcn_readfirstlane(bands_left > 0 ? (active0 * a.thresh_num) / 16 : drained);
    {
      const unsigned long long act = __ballot(status == ST_ACTIVE);
      // cone rays: the secondary segments of a GI path (svotrace.comp:446: coneTrace = i != 0)
#ifdef SVO_STAMPS
      uint32_t *const mixp = st_mix;
#else
      uint32_t *const mixp = nullptr;
#endif
      walk.run(stk, lane, t, status, act, __builtin_amdgcn_readfirstlane(threshold),
               kMode == 0 ? __ballot((seg & 0xffu) != 0u) : 0ull, mixp);
    }
#ifdef SVO_STAMPS
    { const unsigned long long now = __builtin_readcyclecounter(); st_trav += now - st_t0; st_t0 = now; }
#endif
  }
#ifdef SVO_STAMPS
  if (lane == 0u) {
    unsigned long long *dbg = (unsigned long long *)(a.heads + 8 * kHeadStride);  // spare words behind the 8 band counters
    const unsigned long long st_end = __builtin_amdgcn_s_memrealtime();
    atomicMax(dbg + 6, ~st_begin); atomicMax(dbg + 7, ~st_dry); atomicMax(dbg + 8, st_dry); atomicMax(dbg + 9, st_end);
    atomicAdd(dbg + 10, st_end - st_begin); atomicAdd(dbg + 11, st_dry - st_begin);
    for (int i = 0; i < 8; i += 2)   // dbg[12..15]: lanes << 32 | trips, for the whole trip / descend / advance / pop
      atomicAdd(dbg + 12 + i / 2, ((unsigned long long)st_mix[i + 1] << 32) + st_mix[i]);

  }
  atomicAdd(a.heads + 8 * kHeadStride + 32 + lane, st_mix[8]);
  atomicAdd(a.heads + 8 * kHeadStride + 32 + 64 + lane, st_mix[9]);
  if (lane == 0u) {
    unsigned long long *parts = (unsigned long long *)(a.heads + 8 * kHeadStride + 32 + 64 + 64);
    for (int i = 0; i < 5; i++) atomicAdd(parts + i, st_part[i]);
    atomicAdd(parts + 5, st_lanes_shaded); atomicAdd(parts + 6, st_lanes_init); atomicAdd(parts + 7, st_lanes_refilled);
  }
  if (lane == 0u) {
    unsigned long long *dbg = (unsigned long long *)(a.heads + 8 * kHeadStride);
    atomicAdd(dbg + 0, st_round); atomicAdd(dbg + 1, st_trav); atomicAdd(dbg + 2, st_nround); atomicAdd(dbg + 3, st_ntrip); atomicAdd(dbg + 4, st_shade); atomicAdd(dbg + 5, st_load);
  }
#endif
}

// a.fold > 1: tiles per group of a band's walk.  1 / 8 / 64 / 512 / all: 4.47 / 4.49 / 4.47 / 4.41 / 4.20 Grays/s at 64
// samples per pixel (tools/history/r03_fold2.sh): a group's samples should be in flight together, not a whole band's
constexpr int kFoldGroup = 8;
// Persistent waves per CU and launch when several launches share the GPU (a ring of more than one slot): the next launch's
// waves take the CUs the previous launch's tail frees.  Swept in rounds 2-4 (profiles/round4_experiments.txt: 10 waves with
// a round once at most 9/16 of the lanes are still traversing is the shape every headline number was measured on; 16 waves: -6 %).
constexpr int kRingWavesPerCu = 10;
// ... and when svo_dispatch_async takes turns on n {stream, image} sets (the reference's loop: one frame per launch, up to n
// launches in flight): waves per CU and launch by n -- about 32 / n, the best of tools/loop_shape.py's sweeps
// (profiles/round6_experiments.txt: 2 sets 12, 3 sets 10, 4 sets 8, 5 and more 6)
inline int overlap_waves_per_cu(int sets) { return sets <= 2 ? 12 : sets == 3 ? 10 : sets == 4 ? 8 : 6; }
constexpr int kHeadSets = 8;   // counter sets: one per frame in flight (its sample launches follow one another on one
                               // stream and share it), reused round-robin
constexpr int kFaccSets = 4;   // colour-sum buffers (spp > 1): one per frame, reused round-robin

struct PersistBuffers {
  uint32_t *heads = nullptr;
  // a counter set / colour-sum buffer may still be in use by a frame in flight on another stream when the ring
  // comes round: the launch that re-uses it first waits (on the GPU, hipStreamWaitEvent) for the event its
  // previous user recorded
  hipEvent_t head_done[kHeadSets] = {};
  bool head_used[kHeadSets] = {};
  float *facc[kFaccSets] = {};
  hipEvent_t facc_done[kFaccSets] = {};
  bool facc_used[kFaccSets] = {};
  size_t facc_floats = 0;   // floats in each colour-sum buffer
  int blocks = 0;
  int thresh_num = 0;   // sixteenths; 0 = the running kernel's default: 9 for persist_kernel (8 / 9 / 10 / 11: 4.28 / 4.38 / 4.33 /
                        // 4.14 Grays/s, tools/history/sweep8.sh), SVO_SPARE_THRESH for the spare-ray kernel
  int spare_mode = 0;   // 1 = the spare-ray kernel (variants/svo_persist2.hip.h) on walkable pools (environment SVO_SPARE=1: round 5's
                        // experiment, measured slower than persist_kernel with launches in flight); 0 = persist_kernel everywhere
  const float4 *ntab = nullptr;   // the context's normal table (made with the context; survives persist_free)
  int ntab_mode = 1;              // environment SVO_NORMAL_TABLE=0: decode in place
  int table_mode = 1;   // 1 = row / column tables + the kTab kernels where they apply (environment SVO_RC_TABLE=0: all in the rounds)
  float *rc[8] = {};           // row / column tables, one buffer per counter set (a set's launches are ordered by its event)
  size_t rc_floats[8] = {};
  uint32_t *prec[8] = {};      // path records of the spare-ray kernel, one buffer per counter set
  size_t prec_words = 0;
  int last_thresh = 9;
  int waves_per_cu = 0;      // 0 = automatic: as many as fit (occupancy query) for one launch at a time, kRingWavesPerCu for
                             // the submissions of a ring with more than one slot (in_ring, set around the launch by ring_submit)
  bool in_ring = false;
  bool in_overlap = false;   // set around the launch by svo_dispatch_async while it takes turns on several image sets ...
  int overlap_sets = 4;      // ... and on how many
  int last_blocks = 0, last_per_cu = 0;   // shape of the last launch (svo_launch_info)
  int max_per_cu = 16, max_per_cu_desc = 16, cus = 256;   // resident waves per CU: byte walk / descriptor walk
  int max_per_cu_spare = 0;                               // ... / the spare-ray kernel (0 = not asked yet)
  int cus_reserved = 0;   // CUs the launching stream may not use (svo_set_reserved_cus): fewer persistent waves
  unsigned launches = 0, frames = 0;
};

inline void persist_free(PersistBuffers &b) {
  if (b.heads) (void)hipFree(b.heads);
  for (auto &p : b.prec) if (p) (void)hipFree(p);
  for (auto &p : b.rc) if (p) (void)hipFree(p);
  for (auto &f : b.facc) if (f) (void)hipFree(f);
  for (auto &e : b.head_done) if (e) (void)hipEventDestroy(e);
  for (auto &e : b.facc_done) if (e) (void)hipEventDestroy(e);
  const int wpc = b.waves_per_cu, th = b.thresh_num, sm = b.spare_mode, tm = b.table_mode;   // tuning survives a resize
  const float4 *nt = b.ntab;
  const int nm = b.ntab_mode;
  b = PersistBuffers();
  b.waves_per_cu = wpc; b.thresh_num = th; b.spare_mode = sm; b.table_mode = tm; b.ntab = nt; b.ntab_mode = nm;
}

// The row / column tables of a launch: per frame of the batch W + H threads.  A handful of workgroups: unlike a kernel of one
// thread per pixel (tried and dropped, profiles/round5_experiments.txt) they find their slots next to the persistent waves of the
// launches in flight as soon as a few of those retire.  The same device functions primary_direction_cam / pixel_rand call, on the
// same values, in the same order.
// Workgroups of ONE wave: a workgroup of four needs four free wave slots on one CU at the same moment, which CUs packed with
// persistent waves offer much later than the single slot a lone wave needs.
#ifndef SVO_RC_BLOCK
#define SVO_RC_BLOCK 64
#endif
#ifndef SVO_RC_WAVES
#define SVO_RC_WAVES 4    // workgroups per frame of the launch
#endif
// the counter set of a launch without tables: zeroed by a one-wave kernel (hipMemsetAsync is a fill kernel of wider workgroups here)
__global__ __launch_bounds__(64) void zero_words_kernel(uint32_t *p, const uint32_t n) {
  for (uint32_t w = threadIdx.x; w < n; w += 64u) p[w] = 0u;
}
constexpr int kCamPack = 8;   // per-frame cameras of a launch that travel in the table kernel's arguments (no copy in front of it)
struct FrameVarPack { FrameVar v[kCamPack]; };
__global__ __launch_bounds__(SVO_RC_BLOCK) void rc_table_kernel(const Frame f, FrameVar *fvar, float *rc, const uint32_t stride, const int sample,
                                                                uint32_t *heads, const FrameVarPack pack, const int npack) {
  // (the launch's counter set is zeroed here too: one small kernel in front of the launch instead of a fill kernel and this one;
  // and with npack > 0 the launch's per-frame cameras arrive in `pack` and are written to the slot's table `fvar`, which the
  // persistent kernel reads with a scalar load per frame: no host-to-device copy -- a blit kernel on this runtime -- in front either)
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    for (uint32_t w = threadIdx.x; w < (uint32_t)kHeadWords; w += (uint32_t)SVO_RC_BLOCK) heads[w] = 0u;
    for (uint32_t w = threadIdx.x; w < (uint32_t)npack * 16u; w += (uint32_t)SVO_RC_BLOCK)
      ((uint32_t *)fvar)[w] = ((const uint32_t *)pack.v)[w];
  }
  const int k = (int)blockIdx.y;
  float *fr = rc + (size_t)k * stride;
  const float *cam = npack > 0 ? pack.v[k].cam : (fvar ? fvar[k].cam : f.cam);
  const float seed2 = (float)((npack > 0 ? pack.v[k].frame_number : (fvar ? fvar[k].frame_number : f.frame_number + k)) + sample);
  const float k0 = 0.1f * 78.233f, k1 = 0.02f * 78.233f;   // (pixel_rand's folded constants)
  // a few waves per frame, each over a strided share of the W + H entries: the fewer workgroups, the sooner all of them have a slot
  for (int i = (int)(blockIdx.x * (uint32_t)SVO_RC_BLOCK + threadIdx.x); i < f.width + f.height; i += (int)(gridDim.x * (uint32_t)SVO_RC_BLOCK)) {
    if (i < f.width) {
      const float x = (float)i;
      fr[2 * i] = rand_of_dot(x * 12.9898f + seed2 * k0);
      fr[2 * i + 1] = (x + 0.5f) / (float)f.width;
    } else {
      const int y = i - f.width;
      const float v = ((float)y + 0.5f) / (float)f.height;
      float *row = fr + ((2 * f.width + 3) & ~3) + 8 * y;
      row[0] = mix_g(cam[3], cam[6], v); row[1] = mix_g(cam[4], cam[7], v); row[2] = mix_g(cam[5], cam[8], v);
      row[3] = rand_of_dot((float)y * 12.9898f + seed2 * k1);
      row[4] = mix_g(cam[9], cam[12], v); row[5] = mix_g(cam[10], cam[13], v); row[6] = mix_g(cam[11], cam[14], v);
      row[7] = 0.0f;
    }
  }
}

// the spare-ray kernel (variants/svo_persist2.hip.h, included behind this file by svo_hip.hip in SVO_VARIANTS builds)
#ifndef SVO_VARIANTS
#define SVO_VARIANTS 0
#endif
#define SVO_HAVE_SPARE (SVO_ASM_LOOP && SVO_VARIANTS)
#if SVO_HAVE_SPARE
template <int kMode>
inline void persist2_launch_mode(const PersistArgs &a, int blocks, hipStream_t stream);
inline hipError_t persist2_occupancy(int *per_cu);
#endif
#ifndef SVO_SPARE_THRESH
#define SVO_SPARE_THRESH 13
#endif
// SVO_SPARE_RECORDS: 0 = the two path records of a lane live in registers (18 each; the kernel then runs 4 waves per SIMD),
// 1 = in global memory (6 waves per SIMD; measured: the records' traffic -- 144 bytes per ray written and read back, more
// than the caches hold with six launches in flight -- costs more than the lanes gained: profiles/round5_experiments.txt)
#ifndef SVO_SPARE_RECORDS
#define SVO_SPARE_RECORDS 0
#endif

template <int kMode>
inline void persist_launch_mode(const PersistArgs &a, int blocks, hipStream_t stream) {
#if SVO_HAVE_SPARE
  if (a.spare) { persist2_launch_mode<kMode>(a, blocks, stream); return; }
#endif
  if (a.rc && a.desc) {   // (the tables are only made for launches that walk the descriptor table)
    if (a.fvar) hipLaunchKernelGGL((persist_kernel<kMode, DescWalk, true, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((persist_kernel<kMode, DescWalk, false, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
    return;
  }
  if (a.fvar) {
    if (a.desc) hipLaunchKernelGGL((persist_kernel<kMode, DescWalk, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((persist_kernel<kMode, ByteWalk, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  } else if (a.desc) hipLaunchKernelGGL((persist_kernel<kMode, DescWalk>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((persist_kernel<kMode, ByteWalk>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
}

__global__ void persist_resolve_kernel(const Frame f, const float *facc, size_t npix, uint32_t *color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = frame_gy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7);
  if (x >= f.width || y >= f.y1 || y >= f.height) return;
  const size_t pix = (size_t)blockIdx.z * f.frame_stride +      // frame blockIdx.z of a batch
                     (size_t)frame_oy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7) * f.width + x;
  const float inv = 1.0f / (float)f.spp;
  const V3 sum = mk(facc[pix], facc[npix + pix], facc[2 * npix + pix]);
  color[pix] = final_rgba8(f, x, y, mk(sum.x * inv, sum.y * inv, sum.z * inv), color + pix);
}

// The same for a launch that left every sample in a slot of its own (persist_emit, a.fold > 1): one workgroup per tile;
// the samples are added in sample order, the order in which one launch per sample adds them: ((0 + s0) + s1) + ...
__global__ __launch_bounds__(64) void persist_resolve_tiles_kernel(const Frame f, const float *facc, int fold, uint32_t *color) {
  const uint32_t l = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y, k = blockIdx.z;
  const float *p = facc + (((size_t)k * (size_t)f.ntiles + (size_t)ty * f.tiles_x + tx) * (size_t)fold) * 192 + l;
  V3 sum = mk(0.0f + p[0], 0.0f + p[64], 0.0f + p[128]);
  for (int i = 1; i < fold; i++) sum = mk(sum.x + p[192 * i], sum.y + p[192 * i + 64], sum.z + p[192 * i + 128]);
  const int x = (int)(tx * 8u + (l & 7u)), y = frame_gy(f, (int)ty, (int)(l >> 3));
  if (x >= f.width || y >= f.y1 || y >= f.height) return;
  const size_t pix = (size_t)k * f.frame_stride + (size_t)frame_oy(f, (int)ty, (int)(l >> 3)) * f.width + x;
  const float inv = 1.0f / (float)f.spp;
  color[pix] = final_rgba8(f, x, y, mk(sum.x * inv, sum.y * inv, sum.z * inv), color + pix);
}

// A progressive sequence (f.seq frames of the cross-frame accumulation in one launch, every frame's colour in a slot of its
// own like a sample): the reference's recurrence in frame order -- blend with the texel the previous frame left, quantise to
// rgba8 exactly as imageStore / imageLoad do (svotrace.comp:712-719, pins in svo_fused.hip.h::final_rgba8_v), next frame.
// What f.seq dispatches of one frame each would leave in the image, bit for bit; `color` holds the image the sequence starts on.
__global__ __launch_bounds__(64) void persist_resolve_sequence_kernel(const Frame f, const float *facc, int fold, uint32_t *color) {
  const uint32_t l = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y;
  const float *p = facc + (((size_t)ty * f.tiles_x + tx) * (size_t)fold) * 192 + l;
  const int x = (int)(tx * 8u + (l & 7u)), y = frame_gy(f, (int)ty, (int)(l >> 3));
  if (x >= f.width || y >= f.y1 || y >= f.height) return;
  const size_t pix = (size_t)frame_oy(f, (int)ty, (int)(l >> 3)) * f.width + x;
  uint32_t last = color[pix];
  for (int i = 0; i < fold; i++)
    last = final_rgba8_v(f, f.frame_number + i, x, y, mk(0.0f + p[192 * i], 0.0f + p[192 * i + 64], 0.0f + p[192 * i + 128]), last);
  color[pix] = last;
}

// can `n` samples (or frames of a progressive sequence) per pixel be carried by one launch of `f`?
inline bool persist_can_fold(const Frame &f, int n) {
  const unsigned long long nb = (unsigned long long)(f.batch > 1 ? f.batch : 1);
  const unsigned long long bytes = 192ull * (unsigned long long)n * (unsigned long long)f.ntiles * nb * sizeof(float);
#if SVO_VARIANTS
  static const unsigned long long budget = getenv("SVO_FOLD_BYTES") ? strtoull(getenv("SVO_FOLD_BYTES"), nullptr, 10) : (4ull << 30);
#else
  const unsigned long long budget = 4ull << 30;
#endif
  return SVO_BAND_COLMAJOR && n > 1 && f.bounces <= 255 && bytes <= budget && f.batch <= 255 && n < 65536 &&
         (long long)f.ntiles * (long long)nb * n < (1ll << 25);
}

// `out_npix` = elements of the output images: W*H, or more when packed stripes overhang the frame (caller-owned
// gather buffers); the colour-sum planes are indexed like the outputs.
// `desc` / `aux` / `desc_count`: the interior-descriptor table of the pool (svo_derive.hip.h), or null = walk the records
inline int persist_launch(PersistBuffers &b, const uint8_t *pool, const Frame &f, uint32_t *color, float *depth,
                          uint4 *hits, size_t out_npix, hipStream_t stream, const uint2 *desc = nullptr,
                          const uint2 *aux = nullptr, uint32_t desc_count = 0, FrameVar *fvar = nullptr,
                          const FrameVar *fvar_host = nullptr, int (*copy_cams)(void *) = nullptr, void *copy_arg = nullptr) {
  // fvar: where the launch's per-frame cameras live on the device (null: one camera, f.cam); fvar_host: the same on the host.  They
  // reach the device inside the table kernel's arguments when the launch has one and they fit (kCamPack), else through copy_cams
  // (the caller's staged copy on `stream`), called here in front of the first kernel that reads them.
  // the colour-sum planes are indexed like the outputs: a batch needs room for all its frames
  const size_t npix = f.batch > 1 ? std::max(out_npix, (size_t)f.batch * (size_t)f.frame_stride) : out_npix;
  hipError_t e;
  if (!b.heads) {
    if ((e = hipMalloc((void **)&b.heads, kHeadSets * kHeadWords * sizeof(uint32_t))) != hipSuccess) return (int)e;
    for (auto &ev : b.head_done)
      if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return (int)e;
    for (auto &ev : b.facc_done)
      if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return (int)e;
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel<0, ByteWalk>, 64, 0) != hipSuccess || per_cu < 1)
      per_cu = 16;
    b.max_per_cu = per_cu;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel<0, DescWalk>, 64, 0) != hipSuccess || per_cu < 1)
      per_cu = 16;
    b.max_per_cu_desc = per_cu;
    b.cus = cus;
#if SVO_VARIANTS
    // experiment knobs (override svo_set_tuning): libsvohip_variants.so only
    if (const char *e1 = getenv("SVO_PERSIST_WAVES_PER_CU")) b.waves_per_cu = atoi(e1);
    if (const char *e2 = getenv("SVO_PERSIST_THRESH")) b.thresh_num = atoi(e2);
    if (const char *e3 = getenv("SVO_SPARE")) b.spare_mode = atoi(e3) != 0 ? 1 : 0;
    if (const char *e4 = getenv("SVO_RC_TABLE")) b.table_mode = atoi(e4) != 0 ? 1 : 0;
    if (const char *e5 = getenv("SVO_NORMAL_TABLE")) b.ntab_mode = atoi(e5) != 0 ? 1 : 0;
#endif
  }
  // the spare-ray kernel walks the descriptor table in assembly: pools the table cannot state, and builds with hipcc's
  // translation of the loop (SVO_ASM_LOOP=0), run persist_kernel
  const bool spare_kernel = SVO_HAVE_SPARE && b.spare_mode != 0 && desc != nullptr;
#if SVO_HAVE_SPARE
  if (spare_kernel && b.max_per_cu_spare == 0) {
    int per_cu = 0;
    if (persist2_occupancy(&per_cu) != hipSuccess || per_cu < 1) per_cu = 16;
    b.max_per_cu_spare = per_cu;
  }
#endif
  if (spare_kernel && SVO_SPARE_RECORDS) {
    const size_t need = (size_t)b.cus * (size_t)b.max_per_cu_desc * (size_t)(2 * 18 * 64);   // = kRecWaveWords per wave that fits
    if (b.prec_words < need) {
      if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
      for (auto &p : b.prec) { if (p) (void)hipFree(p); p = nullptr; }
      b.prec_words = 0;
      for (auto &p : b.prec)
        if ((e = hipMalloc((void **)&p, need * sizeof(uint32_t))) != hipSuccess) return (int)e;
      b.prec_words = need;
    }
  }
  {
    const int fill = spare_kernel ? b.max_per_cu_spare : (desc ? b.max_per_cu_desc : b.max_per_cu);
    int per_cu = b.waves_per_cu > 0 ? b.waves_per_cu : (b.in_ring ? std::min(kRingWavesPerCu, fill) : (b.in_overlap ? std::min(overlap_waves_per_cu(b.overlap_sets), fill) : fill));
    b.blocks = (b.cus - b.cus_reserved) * per_cu;
    b.last_per_cu = per_cu;
  }
  const int spp = f.spp < 1 ? 1 : f.spp;
  const bool resolve = spp > 1 || f.progressive;   // colours go through the float planes and the resolve kernel
  float *facc = nullptr;
  const unsigned frame_no = b.frames++;
  int fset = 0;
  // All samples of a frame in ONE launch when their slots fit (3 floats per pixel and sample, kFaccSets times): the
  // launch's bands hold the frame's samples one after the other like the frames of a batch, a wave goes from sample to
  // sample without a tail, and the sums cost one store per sample instead of a read-modify-write.  Otherwise one launch
  // per sample.  Same bytes either way (tests/test_gpu_inflight.py).
  int fold = 1;
  unsigned long long slots = 0;
  const bool sequence = f.progressive && f.seq > 1;   // the caller has checked persist_can_fold(f, f.seq) and spp == 1
  const int per_pixel = sequence ? f.seq : spp;
  if (persist_can_fold(f, per_pixel)) {
    fold = per_pixel;
    slots = 192ull * (unsigned long long)fold * (unsigned long long)f.ntiles * (f.batch > 1 ? f.batch : 1);
  }
  if (sequence && (fold != f.seq || spp != 1)) return (int)hipErrorInvalidValue;
  if (resolve) {
    const size_t need = fold > 1 ? (size_t)slots : 3 * npix;   // floats per buffer
    if (b.facc_floats < need) {   // grow: nothing may still be summing into the old buffers
      if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
      for (int i = 0; i < kFaccSets; i++) {
        if (b.facc[i]) (void)hipFree(b.facc[i]);
        b.facc[i] = nullptr;
        b.facc_used[i] = false;
      }
      b.facc_floats = 0;
      for (int i = 0; i < kFaccSets; i++)
        if ((e = hipMalloc((void **)&b.facc[i], need * sizeof(float))) != hipSuccess) return (int)e;
      b.facc_floats = need;
    }
    fset = (int)(frame_no % kFaccSets);
    facc = b.facc[fset];
    if (b.facc_used[fset] && (e = hipStreamWaitEvent(stream, b.facc_done[fset], 0)) != hipSuccess) return (int)e;
  }
  PersistArgs a;
  a.pool = pool; a.f = f; a.color = color; a.depth = depth; a.hits = hits; a.facc = facc; a.npix = npix;
  a.tiles_per_band = (f.ntiles + 7) / 8;
  a.rows_per_band = (f.tiles_y + 7) / 8;
  a.thresh_num = b.thresh_num > 0 ? b.thresh_num : (spare_kernel ? SVO_SPARE_THRESH : 9);
  b.last_thresh = a.thresh_num;
  a.fold = fold;
  a.group = kFoldGroup;
  a.desc = desc; a.aux = aux; a.desc_count = desc_count;
  a.fvar = fvar;
  {
    const int last_rows = a.rows_per_band > 0 ? f.tiles_y % a.rows_per_band : 0;   // the one band that is not full (if any)
    const uint32_t tpf[2] = {(uint32_t)(a.rows_per_band * f.tiles_x * fold), (uint32_t)(last_rows * f.tiles_x * fold)};
    const uint64_t nb = (uint64_t)(f.batch > 1 ? f.batch : 1);
    a.mg_rows[0] = udiv_magic((uint32_t)a.rows_per_band, (uint64_t)a.rows_per_band * (uint64_t)f.tiles_x);
    a.mg_rows[1] = udiv_magic((uint32_t)last_rows, (uint64_t)last_rows * (uint64_t)f.tiles_x);
    a.mg_tpf[0] = udiv_magic(tpf[0], nb * tpf[0]);   // n = slot >> 6 < frames x tile slots per frame
    a.mg_tpf[1] = udiv_magic(tpf[1], nb * tpf[1]);
  }
  const long long work = (long long)f.ntiles * (f.batch > 1 ? f.batch : 1) * fold;
  const int blocks = work < (long long)b.blocks ? (int)work : b.blocks;
  b.last_blocks = blocks;
  // a ring of counter sets: frames may be in flight on different streams at the same time.  A frame's sample launches
  // are ordered by its stream, so they share the frame's set; only another frame's re-use waits (for the event)
  const int hset = (int)(frame_no % kHeadSets);
  a.heads = b.heads + (size_t)hset * kHeadWords;
  a.prec = (spare_kernel && SVO_SPARE_RECORDS) ? b.prec[hset] : nullptr;   // (a counter set's launches are ordered by the set's event: so are its records)
  a.spare = spare_kernel ? 1 : 0;
  // the row / column tables: launches that carry one sample per pixel (samples / sequences folded into a launch seed every
  // sample of a pixel differently and keep computing in the round), on the descriptor walk, not the spare-ray kernel
  a.rc = nullptr; a.rc_stride = 0;
  a.ntab = b.ntab_mode != 0 ? b.ntab : nullptr;
  if (b.table_mode != 0 && desc != nullptr && !spare_kernel && fold == 1) {
    const size_t stride = (size_t)((2 * f.width + 3) & ~3) + 8 * (size_t)f.height, need = stride * (size_t)(f.batch > 1 ? f.batch : 1);
    if (b.rc_floats[hset] < need) {   // grow all sets at once (one wait for the device, not one per set inside a run)
      if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;   // (launches in flight may still read the old tables)
      for (int i = 0; i < kHeadSets; i++) {
        if (b.rc[i]) (void)hipFree(b.rc[i]);
        b.rc[i] = nullptr; b.rc_floats[i] = 0;
      }
      for (int i = 0; i < kHeadSets; i++) {
        if ((e = hipMalloc((void **)&b.rc[i], need * sizeof(float))) != hipSuccess) return (int)e;
        b.rc_floats[i] = need;
      }
    }
    a.rc = b.rc[hset]; a.rc_stride = (uint32_t)stride;
  }
  if (b.head_used[hset] && (e = hipStreamWaitEvent(stream, b.head_done[hset], 0)) != hipSuccess) return (int)e;
  for (int s = 0; s < (fold > 1 ? 1 : spp); s++) {
    a.reverse = SVO_SERPENTINE ? (int)(b.launches++ & 1u) : 0;
    a.sample = s;
    const int nb = f.batch > 1 ? f.batch : 1;
    const bool packed = a.rc && fvar && fvar_host && nb <= kCamPack;
    if (fvar && !packed && s == 0 && copy_cams) { const int rcc = copy_cams(copy_arg); if (rcc) return rcc; }
    if (a.rc) {
      FrameVarPack pack;
      if (packed) memcpy(pack.v, fvar_host, (size_t)nb * sizeof(FrameVar));
      else memset(&pack, 0, sizeof pack);
      const unsigned per_frame = (unsigned)((f.width + f.height + SVO_RC_BLOCK - 1) / SVO_RC_BLOCK);
      const dim3 tgrid(per_frame < (unsigned)SVO_RC_WAVES ? per_frame : (unsigned)SVO_RC_WAVES, (unsigned)(f.batch > 1 ? f.batch : 1));
      hipLaunchKernelGGL(rc_table_kernel, tgrid, dim3(SVO_RC_BLOCK), 0, stream, f, fvar, b.rc[hset], a.rc_stride, s, a.heads, pack, packed ? nb : 0);
      if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    } else {
      hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, stream, a.heads, (uint32_t)kHeadWords);
      if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
    switch (f.render_mode) {
      case 0: persist_launch_mode<0>(a, blocks, stream); break;
      case 1: persist_launch_mode<1>(a, blocks, stream); break;
      case 2: persist_launch_mode<2>(a, blocks, stream); break;
      case 3: persist_launch_mode<3>(a, blocks, stream); break;
      default: persist_launch_mode<4>(a, blocks, stream); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  if ((e = hipEventRecord(b.head_done[hset], stream)) != hipSuccess) return (int)e;
  b.head_used[hset] = true;
  if (resolve) {
    if (sequence) {
      hipLaunchKernelGGL(persist_resolve_sequence_kernel, dim3((unsigned)f.tiles_x, (unsigned)f.tiles_y), dim3(64), 0, stream, f, facc, fold, color);
    } else if (fold > 1) {
      dim3 grid((unsigned)f.tiles_x, (unsigned)f.tiles_y, (unsigned)(f.batch > 1 ? f.batch : 1));
      hipLaunchKernelGGL(persist_resolve_tiles_kernel, grid, dim3(64), 0, stream, f, facc, fold, color);
    } else {
      dim3 grid((unsigned)((f.width + 255) / 256), (unsigned)(f.tiles_y * 8), (unsigned)(f.batch > 1 ? f.batch : 1));
      hipLaunchKernelGGL(persist_resolve_kernel, grid, dim3(256), 0, stream, f, facc, npix, color);
    }
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if ((e = hipEventRecord(b.facc_done[fset], stream)) != hipSuccess) return (int)e;
    b.facc_used[fset] = true;
  }
  return 0;
}

}  // namespace svo
