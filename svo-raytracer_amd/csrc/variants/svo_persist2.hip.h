// svo_persist2.hip.h -- pipeline 1 on the descriptor table, with a spare ray per lane.  Round 5's experiment on the lanes that wait
// for a round; OPT-IN (environment SVO_SPARE=1): it fills the lanes and saves a tenth of the vector instructions, but its state costs
// two waves per SIMD and with launches in flight that costs more (-15 %: DESIGN.md section 4, profiles/round5_experiments.txt).
//
// persist_kernel (svo_persistent.hip.h) keeps 64 paths in flight per wave, one per lane; a lane whose ray has stopped waits
// for the next round, and a round only pays once 7/16 of the lanes wait: 24.4 % of all lane-trips of the traversal loop are
// such waits (SVO_STAMPS histogram: mean 48.4 of 64 lanes traversing, profiles/round5_experiments.txt).  Here a lane carries
// TWO paths: the one whose ray it traverses, and one whose ray waits in a nine-register slot -- as a spare, ready to start
// (set up in the last round), or as the parked result of a ray that has stopped and awaits shading.  The traversal loop
// (svo_travloop3.h) exchanges the two in place whenever a few lanes have stopped with a spare at hand, so lanes keep
// traversing between rounds; a round shades every parked result (one per lane), turns each into the path's next ray or a
// stored pixel plus a fresh primary, and leaves it in the slot as the new spare.
//
// What outlives a cast of a path (direction, throughput mask, radiance, last normal / value, depth, pixel, segment, random
// number: 18 words) is only needed when the path is shaded.  Two records per lane; SVO_SPARE_RECORDS picks where they live: in
// registers (default: 128 VGPRs, 4 waves per SIMD, the record of the slot's path picked by the lane set `cpath`), or in global
// memory ([wave][record][field][lane] words, coalesced 256-byte rows; 80 VGPRs and 6 waves per SIMD, but 144 bytes per ray written
// and read back: measured -26 %).
//
// Same arithmetic as persist_kernel statement for statement (the shading block below is its block, reading the cast from
// a parked slot instead of the traversal registers): same bytes, checked by the whole parity suite on this kernel (it is
// what pipeline 1 ran while it was the default) and against persist_kernel itself (tests/test_gpu_spare.py).
#pragma once
#include "../svo_persistent.hip.h"
#include "svo_travloop3.h"

namespace svo {

constexpr int kRecWords = 18;                       // words of a path record
constexpr int kRecWaveWords = 2 * kRecWords * 64;   // two records per lane

// SVO_SPARE_THRESH (svo_persistent.hip.h), sixteenths: a round starts once that share of the lanes is all that still traverses
// -- the others wait without a spare

// SVO_SPARE_RECORDS (svo_persistent.hip.h): where the two path records of a lane live
#ifndef SVO_SPARE_WAVES_PER_SIMD
#define SVO_SPARE_WAVES_PER_SIMD (SVO_SPARE_RECORDS ? SVO_DERIVED_WAVES_PER_SIMD : 4)
#endif
struct PathRec {
  uint32_t pix, seg, pxy;
  V3 d, mask, accum, normal;
  float r, depth;
  uint32_t value;
};
__device__ __forceinline__ uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
__device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ V3 sel(bool c, V3 a, V3 b) { return mk(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ PathRec sel(bool c, const PathRec &a, const PathRec &b) {
  PathRec o;
  o.pix = sel(c, a.pix, b.pix); o.seg = sel(c, a.seg, b.seg); o.pxy = sel(c, a.pxy, b.pxy);
  o.d = sel(c, a.d, b.d); o.mask = sel(c, a.mask, b.mask); o.accum = sel(c, a.accum, b.accum); o.normal = sel(c, a.normal, b.normal);
  o.r = sel(c, a.r, b.r); o.depth = sel(c, a.depth, b.depth); o.value = sel(c, a.value, b.value);
  return o;
}

template <int kMode, bool kCams>
__global__ __launch_bounds__(64, SVO_SPARE_WAVES_PER_SIMD) void persist2_kernel(const PersistArgs a) {
  __shared__ WaveStack2 stk;
  const uint32_t lane = threadIdx.x;
  const Frame &f = a.f;
  DescWalk walk;
  walk.setup(a);
  const V3 cam_o = mk(f.cam[0], f.cam[1], f.cam[2]);
  const V3 sun2 = normalize3(mk(0.5f, 0.5f, 0.5f));
#if SVO_SPARE_RECORDS
  uint32_t *const recs = a.prec + (size_t)blockIdx.x * (size_t)kRecWaveWords + lane;
#else
  PathRec P0, P1;
  P0.pix = P0.seg = P0.pxy = P0.value = 0u; P0.d = P0.mask = P0.accum = P0.normal = mk(0.f, 0.f, 0.f); P0.r = P0.depth = 0.0f;
  P1 = P0;
#endif
// is this lane in the (wave-uniform) lane set?  -- a scalar mask used as a predicate costs no vector register
#define SVO_IN(m) __builtin_amdgcn_inverse_ballot_w64(m)

  TravRegs3 t;
  t.cx = t.cy = t.cz = t.bx = t.by = t.bz = 0.0f; t.octant = 0; t.px = t.py = t.pz = 1.0f;
  t.t_min = t.t_max = t.h = 0.0f; t.sexp = 0.5f; t.scale = kMaxScale - 1; t.cs = 0; t.self = kDescRoot; t.dlo = t.dhi = 0; t.iter = 0;
  t.lod_scale = kMaxScale - kMaxDepth;
  int status = ST_IDLE;
  Slot s;
  s.w0 = s.w1 = s.w2 = s.w3 = s.w4 = s.w5 = s.w6 = s.w7 = s.w8 = 0u;
  // what every lane's slot holds (neither set: nothing), which record belongs to the traversing ray, the cone (secondary) rays
  unsigned long long spare = 0ull, parked = 0ull, cpath = 0ull, ccone = 0ull, scone = 0ull;

  uint32_t band = xcc_id();
  int bands_left = 8;

#ifdef SVO_STAMPS
  unsigned long long st_round = 0, st_trav = 0, st_nround = 0, st_ntrip = 0, st_shade = 0, st_load = 0, st_t0 = __builtin_readcyclecounter();
  const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_dry = 0;
  uint32_t st_mix[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
#ifdef SVO_STAMPS
    const unsigned long long st_s0 = __builtin_readcyclecounter();
#endif
    // ---------------- a stopped ray whose lane has nothing in its slot parks its result there (the path's record follows)
    {
      const bool mv = status >= ST_HIT && !SVO_IN(spare | parked);
      if (mv) { slot_park(s, t, status); status = ST_IDLE; }
      const unsigned long long m = __ballot(mv);
      parked |= m; cpath ^= m;
    }
    // ---------------- parked results: shade, then regenerate the next ray in place or retire (persist_kernel's block)
    const bool shade = SVO_IN(parked);
    // the slot's path owns the record the traversing ray does not
#if SVO_SPARE_RECORDS
    uint32_t *const rec = recs + (SVO_IN(cpath) ? 0 : kRecWords * 64);
#else
    const bool rec0 = SVO_IN(cpath);
#endif
    uint32_t pix = 0, seg = 0, value = 0;
    int px = 0, py = 0;
    V3 d = mk(0.f, 0.f, 0.f), mask = mk(1.f, 1.f, 1.f), accum = mk(0.f, 0.f, 0.f), normal = mk(0.f, 0.f, 0.f);
    float r = 0.0f, depth = 0.0f;
    bool emit = false, ninit = false, icone = false;
    V3 ecol = mk(0.f, 0.f, 0.f), io = mk(0.f, 0.f, 0.f);
    float edepth = 0.0f, its = 0.0f;
    if (shade) {
#if SVO_SPARE_RECORDS
      pix = rec[0 * 64]; seg = rec[1 * 64];
      { const uint32_t pxy = rec[2 * 64]; px = (int)(pxy & 0xffffu); py = (int)(pxy >> 16); }
      d = mk(__uint_as_float(rec[3 * 64]), __uint_as_float(rec[4 * 64]), __uint_as_float(rec[5 * 64]));
      mask = mk(__uint_as_float(rec[6 * 64]), __uint_as_float(rec[7 * 64]), __uint_as_float(rec[8 * 64]));
      accum = mk(__uint_as_float(rec[9 * 64]), __uint_as_float(rec[10 * 64]), __uint_as_float(rec[11 * 64]));
      normal = mk(__uint_as_float(rec[12 * 64]), __uint_as_float(rec[13 * 64]), __uint_as_float(rec[14 * 64]));
      r = __uint_as_float(rec[15 * 64]); depth = __uint_as_float(rec[16 * 64]); value = rec[17 * 64];
#else
      {
        const PathRec P = sel(rec0, P0, P1);
        pix = P.pix; seg = P.seg; px = (int)(P.pxy & 0xffffu); py = (int)(P.pxy >> 16);
        d = P.d; mask = P.mask; accum = P.accum; normal = P.normal; r = P.r; depth = P.depth; value = P.value;
      }
#endif
      const Cast c = slot_result(walk.pool, walk.tab, s);
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
          const V3 sk = sky_colour(d);
          ecol = mk(0.0f + sk.x, 0.0f + sk.y, 0.0f + sk.z);
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
    parked = 0ull;   // every parked result has been taken; its slot holds nothing until a ray is set up in it below
    if (emit) persist_emit(a, pix, seg >> 8, px, py, ecol, edepth);

#ifdef SVO_STAMPS
    st_shade += __builtin_readcyclecounter() - st_s0;
#endif
    // ---------------- refill the empty slots: ballot + prefix count, one atomic per wave (persist_kernel's refill)
    uint32_t cam_frame = 0u; int cam_sample = 0; bool cam_fresh = false;   // kCams only
    while (bands_left > 0) {
      const unsigned long long idle = ~spare & ~__ballot(ninit) & __ballot(true);
      if (idle == 0ull) break;
      {
        const bool want = SVO_IN(idle);
        const uint32_t n = (uint32_t)__builtin_popcountll(idle);
        const int leader = __builtin_ctzll(idle);
        // a band = a strip of whole tile rows, walked column by column (svo_persistent.hip.h)
        const int first_row = (int)band * a.rows_per_band;
        int band_rows = f.tiles_y - first_row;
        band_rows = band_rows < 0 ? 0 : (band_rows > a.rows_per_band ? a.rows_per_band : band_rows);
        const uint32_t band_frame = (uint32_t)(band_rows * f.tiles_x) * 64u;   // pixel slots of the band in one frame
        const uint32_t band_total = band_frame * (uint32_t)(f.batch * a.fold);   // ... and over the launch's frames x samples
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(a.heads + band * kHeadStride, n);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);   // wave-uniform, and the compiler knows it
        const uint32_t slot =
            base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (want && slot < band_total) {
          const uint32_t l = slot & 63u;
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
            const uint32_t rr = q - first * (uint32_t)a.fold;
            si = rr / gsize;
            j = (int)(first + rr % gsize);
          }
          int tile_x = (int)udiv_by((uint32_t)j, (uint32_t)band_rows, a.mg_rows[full_band]);
          const int tile_y = first_row + (j - tile_x * band_rows);
          if (a.reverse) tile_x = f.tiles_x - 1 - tile_x;   // serpentine: this frame ends where the next one starts
          px = tile_x * 8 + (int)(l & 7u);
          py = frame_gy(f, tile_y, (int)(l >> 3));
          if (px < f.width && py < f.y1 && py < f.height) {
            pix = fi * f.frame_stride + (uint32_t)frame_oy(f, tile_y, (int)(l >> 3)) * (uint32_t)f.width + (uint32_t)px;
            if (!kCams) d = primary_direction(f, px, py);
            seg = a.fold > 1 ? (si << 8) | (fi << 24) : 0u;
            mask = mk(1.f, 1.f, 1.f);
            accum = mk(0.f, 0.f, 0.f);
            normal = mk(0.f, 0.f, 0.f);
            value = 0u;
            depth = 0.0f;
            if (kMode == 0 && !kCams) r = pixel_rand((float)px, (float)py, (float)(f.frame_number + (int)fi + a.sample + (int)si));
            if (kCams) { cam_frame = fi; cam_sample = a.sample + (int)si; cam_fresh = true; }
            ninit = true; icone = false; io = cam_o; its = beam_start(f, px, py);
          }
        }
        if (kCams) {
          // the camera of every refilled lane's frame: one 64-byte scalar load per distinct frame (svo_persistent.hip.h)
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
              d = primary_direction_cam(cam, f.width, f.height, px, py);
              io = mk(cam[0], cam[1], cam[2]);
              if (kMode == 0) r = pixel_rand((float)px, (float)py, (float)((int)raw[15] + cam_sample));
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
          break;
        } else {
          break;
        }
      }
    }
    // ---------------- set up the new rays in their slots -- regenerated bounce / shadow rays and refilled primaries together --
    // and write their paths' records
    {
      int nst = ST_IDLE;
      if (ninit) {
        nst = slot_init(s, io, d, its);
        if (kMode == 4) {   // no cast: straight to the (black) pixel
          s.w0 = kDescRoot; s.w1 = kIterBias; s.w2 = (uint32_t)(kMaxScale - 1) | ((uint32_t)ST_MISS << 16);
          s.w3 = s.w4 = s.w5 = __float_as_uint(1.0f); s.w6 = 0u; s.w7 = 0u; s.w8 = 0u;
          nst = ST_MISS;
        }
#if SVO_SPARE_RECORDS
        rec[0 * 64] = pix; rec[1 * 64] = seg; rec[2 * 64] = (uint32_t)px | ((uint32_t)py << 16);
        rec[3 * 64] = __float_as_uint(d.x); rec[4 * 64] = __float_as_uint(d.y); rec[5 * 64] = __float_as_uint(d.z);
        rec[6 * 64] = __float_as_uint(mask.x); rec[7 * 64] = __float_as_uint(mask.y); rec[8 * 64] = __float_as_uint(mask.z);
        rec[9 * 64] = __float_as_uint(accum.x); rec[10 * 64] = __float_as_uint(accum.y); rec[11 * 64] = __float_as_uint(accum.z);
        rec[12 * 64] = __float_as_uint(normal.x); rec[13 * 64] = __float_as_uint(normal.y); rec[14 * 64] = __float_as_uint(normal.z);
        rec[15 * 64] = __float_as_uint(r); rec[16 * 64] = __float_as_uint(depth); rec[17 * 64] = value;
#else
        PathRec P;
        P.pix = pix; P.seg = seg; P.pxy = (uint32_t)px | ((uint32_t)py << 16);
        P.d = d; P.mask = mask; P.accum = accum; P.normal = normal; P.r = r; P.depth = depth; P.value = value;
        if (rec0) P0 = P; else P1 = P;
#endif
      }
      const unsigned long long nin = __ballot(ninit), ok = __ballot(ninit && nst == ST_ACTIVE);
      spare |= ok;
      parked |= nin & ~ok;   // a ray that ends where it starts (all-NaN, mode 4): its result waits for the next round
      scone = (scone & ~nin) | __ballot(ninit && icone);
    }
    // ---------------- lanes that are not traversing take the spare of their slot; a result they hold moves into the slot
    {
      const bool sw = status != ST_ACTIVE && SVO_IN(spare);
      const bool had = sw && status >= ST_HIT;
      if (sw) {
        Slot old = s;
        if (had) slot_park(old, t, status);
        slot_start(s, t, walk.rootd);
        walk.fresh_stack(stk, lane);
        if (had) s = old;
        status = ST_ACTIVE;
      }
      const unsigned long long m = __ballot(sw);
      spare &= ~m; parked |= __ballot(had); cpath ^= m;
      ccone = (ccone & ~m) | (scone & m); scone &= ~m;
    }
    const unsigned long long actm = __ballot(status == ST_ACTIVE);
    if (actm == 0ull) {
      if (parked != 0ull || __ballot(status >= ST_HIT) != 0ull || bands_left > 0) continue;
      break;
    }
    // (the first rounds of a wave, and its last ones: slots that are still empty while there is work left are filled before
    // the trips start)
    if (bands_left > 0 && __builtin_popcountll(~(spare | parked)) >= 16) continue;

#ifdef SVO_STAMPS
    { const unsigned long long now = __builtin_readcyclecounter(); st_round += now - st_t0; st_t0 = now; st_nround++; }
#endif
    // ---------------- traverse until enough lanes wait without a spare to make a round worthwhile
    const int active0 = __builtin_popcountll(actm);
    const int drained = (active0 * SVO_DRAIN_NUM) / 16 < active0 - 1 ? (active0 * SVO_DRAIN_NUM) / 16 : (active0 > 0 ? active0 - 1 : 0);
    const int threshold = __builtin_amdgcn_readfirstlane(bands_left > 0 ? (active0 * a.thresh_num) / 16 : drained);
    {
#ifdef SVO_STAMPS
      uint32_t *const mixp = st_mix;
#else
      uint32_t *const mixp = nullptr;
#endif
      unsigned long long cone = kMode == 0 ? ccone : 0ull;
      trav_loop3(walk.tab, stk, lane, t, status, s, actm, threshold, cone, scone, spare, parked, cpath, walk.rootd, mixp);
      ccone = cone;
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
  if (lane == 0u) {
    unsigned long long *dbg = (unsigned long long *)(a.heads + 8 * kHeadStride);
    atomicAdd(dbg + 0, st_round); atomicAdd(dbg + 1, st_trav); atomicAdd(dbg + 2, st_nround); atomicAdd(dbg + 3, st_ntrip); atomicAdd(dbg + 4, st_shade); atomicAdd(dbg + 5, st_load);
  }
#endif
}

inline hipError_t persist2_occupancy(int *per_cu) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, persist2_kernel<0, false>, 64, 0);
}

template <int kMode>
inline void persist2_launch_mode(const PersistArgs &a, int blocks, hipStream_t stream) {
  if (a.fvar) hipLaunchKernelGGL((persist2_kernel<kMode, true>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((persist2_kernel<kMode, false>), dim3((unsigned)blocks), dim3(64), 0, stream, a);
}

}  // namespace svo
