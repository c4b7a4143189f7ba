// svo_wavefront.hip.h -- pipeline 2: wavefront tracing with a LEAN persistent traversal kernel.
//
// In-kernel stamps on pipeline 1 showed that one traversal iteration takes ~1700 cycles of a wave's
// life even when the wave is alone on its SIMD (a serial chain of ~140 dependent instructions, one
// dependent load, branches), so throughput grows with the number of resident waves -- and the
// shading code, not the traversal, is what inflates the register count.  Here the two are split:
//   gen      one thread per pixel slot: primary ray records (tile order, 8 XCD bands)
//   trace    persistent waves, traversal ONLY: ~52 VGPRs.  A finished lane writes a 32-byte raw result
//            and takes the next ray record of its XCD's band queue (ballot + prefix count, one atomic per
//            wave round on a per-band counter line).  Rounds cost ~100 instructions instead of ~1000, so
//            lanes are refilled early and few lanes idle.
//   shade    one thread per traced ray at full lane utilisation: decodes the raw result, shades, stores the
//            pixel or appends the path's next ray to the band's queue for the following stage
//            (wave-aggregated ballot/prefix append).
// Stages alternate trace / shade; per-pixel path state (mask, accum, normal, r, value) lives in SoA planes.
// A stage boundary drains the GPU, which is why the host keeps several frames in flight on separate streams
// (each frame uses its own buffer set from a small ring): another frame's kernels fill the gap.
// Results are bit-identical to pipelines 0 / 1 and to the oracle: same cast and shading arithmetic per ray.
#pragma once
#include "../svo_device.h"
#include "../svo_fused.hip.h"
#include "../svo_kernels.h"
#include "../svo_trav.h"

#include <cstdlib>

namespace svo {

constexpr int kWfSets = 4;          // frames that may be in flight at once
constexpr int kWfMaxStages = 16;    // trace stages per sample (1 + secondary segments)
constexpr int kWfStride = 32;       // counters on separate 128-byte lines
constexpr int kStatePlanes = 11;    // mask(3) accum(3) normal(3) r value
enum { P_MX = 0, P_MY, P_MZ, P_AX, P_AY, P_AZ, P_NX, P_NY, P_NZ, P_R, P_VAL };

struct WfSet {
  uint4 *rays[2] = {nullptr, nullptr};  // ray records, 2 x uint4 per slot: {pix, ox, oy, oz} {dx, dy, dz, flags}
  uint4 *results = nullptr;             // raw results, 2 x uint4 per slot
  float *state = nullptr;               // kStatePlanes planes of npix floats
  uint32_t *counters = nullptr;         // per stage: 8 heads + 8 counts, kWfStride apart
  float *facc = nullptr;                // spp > 1 colour sums
  hipEvent_t done = nullptr;            // recorded behind the set's last kernel; the frame that re-uses the set waits for it
  bool used = false;
};
struct WavefrontBuffers {
  WfSet set[kWfSets];
  size_t npix = 0, slots = 0;
  int blocks = 0;
  unsigned launches = 0;
  int waves_per_cu = 0, thresh_num = 12;   // sixteenths
  int max_per_cu = 16, cus = 256;
};

inline void wavefront_free(WavefrontBuffers &b) {
  for (auto &s : b.set) {
    if (s.rays[0]) (void)hipFree(s.rays[0]);
    if (s.rays[1]) (void)hipFree(s.rays[1]);
    if (s.results) (void)hipFree(s.results);
    if (s.state) (void)hipFree(s.state);
    if (s.counters) (void)hipFree(s.counters);
    if (s.facc) (void)hipFree(s.facc);
    if (s.done) (void)hipEventDestroy(s.done);
    s = WfSet();
  }
  b.npix = 0; b.slots = 0;
}

struct WfArgs {
  const uint8_t *pool;
  Frame f;
  uint32_t *color;
  float *depth;
  uint4 *hits;
  float *state, *facc;
  size_t npix;
  const uint4 *rays_in;
  uint4 *rays_out;
  uint4 *results;
  uint32_t *heads;          // this stage's 8 work counters
  const uint32_t *count_in; // this stage's 8 queue sizes (secondary stages)
  uint32_t *count_out;      // next stage's 8 queue sizes
  int tiles_per_band;       // band capacity in tiles; a band's queue starts at band * tiles_per_band * 64
  int segment, sample, thresh_num;
};

__device__ __forceinline__ uint32_t wf_xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}
__device__ __forceinline__ uint32_t band_tiles_of(const Frame &f, int tiles_per_band, uint32_t band) {
  int bt = f.ntiles - (int)band * tiles_per_band;
  bt = bt < 0 ? 0 : (bt > tiles_per_band ? tiles_per_band : bt);
  return (uint32_t)bt;
}

// ---------------------------------------------------------------------------------------------------
// gen: primary ray records + per-pixel random number, one thread per slot
__global__ __launch_bounds__(256) void wf_gen_kernel(const WfArgs a) {
  const Frame &f = a.f;
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  const uint32_t cap = (uint32_t)a.tiles_per_band * 64u;
  const uint32_t band = slot / cap;
  const uint32_t idx = slot - band * cap;
  if (band >= 8u || idx >= band_tiles_of(f, a.tiles_per_band, band) * 64u) return;
  const int tile = (int)band * a.tiles_per_band + (int)(idx >> 6);
  const uint32_t l = idx & 63u;
  const int px = (tile % f.tiles_x) * 8 + (int)(l & 7u);
  const int ty = tile / f.tiles_x;
  const int py = frame_gy(f, ty, (int)(l >> 3));
  uint4 q0, q1;
  if (px < f.width && py < f.y1 && py < f.height) {
    const uint32_t pix = (uint32_t)frame_oy(f, ty, (int)(l >> 3)) * (uint32_t)f.width + (uint32_t)px;
    const V3 d = primary_direction(f, px, py);
    q0 = make_uint4(pix, __float_as_uint(f.cam[0]), __float_as_uint(f.cam[1]), __float_as_uint(f.cam[2]));
    q1 = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), 0u);
    if (f.render_mode == 0)
      a.state[(size_t)P_R * a.npix + pix] = pixel_rand((float)px, (float)py, (float)(f.frame_number + a.sample));
  } else {
    q0 = make_uint4(0xffffffffu, 0u, 0u, 0u);  // slot outside the image
    q1 = make_uint4(0u, 0u, 0u, 0u);
  }
  a.rays_out[2 * (size_t)slot] = q0;
  a.rays_out[2 * (size_t)slot + 1] = q1;
}

// ---------------------------------------------------------------------------------------------------
// trace: persistent waves, traversal only
#ifndef SVO_TRACE_WAVES_PER_SIMD
#define SVO_TRACE_WAVES_PER_SIMD 5
#endif
template <bool kPrimary>
__global__ __launch_bounds__(64, SVO_TRACE_WAVES_PER_SIMD) void wf_trace_kernel(const WfArgs a) {
  __shared__ WaveStack stk;
  const uint32_t lane = threadIdx.x;
  const Frame &f = a.f;
  const BufPool pool = make_bufpool(a.pool, f.pool_len);
  const uint64_t root = load_record(pool, 0u);
  const uint32_t cap = (uint32_t)a.tiles_per_band * 64u;

  Trav t;
  int status = ST_IDLE;
  uint32_t slot = 0;
  uint32_t band = wf_xcc_id();
  int bands_left = 8;

  for (;;) {
    // ---- finished lanes: store the raw result
    if (status >= ST_HIT) {
      uint4 r0, r1;
      r0.x = t.cptr; r0.y = t.rlo; r0.z = t.rhi; r0.w = __float_as_uint(t.t_min);
      r1.x = __float_as_uint(t.px); r1.y = __float_as_uint(t.py); r1.z = __float_as_uint(t.pz);
      r1.w = (t.iter & 0x7ffu) | (((uint32_t)t.scale & 0x3fu) << 11) | ((t.octant & 7u) << 17) | ((t.tag & 3u) << 20) |
             ((uint32_t)status << 22);
      a.results[2 * (size_t)slot] = r0;
      a.results[2 * (size_t)slot + 1] = r1;
      status = ST_IDLE;
    }
    // ---- refill idle lanes from this XCD's band queue (ballot + prefix count, one atomic per wave)
    if (bands_left > 0) {
      const unsigned long long idle = __ballot(status == ST_IDLE);
      if (idle != 0ull) {
        const uint32_t n = (uint32_t)__builtin_popcountll(idle);
        const int leader = __builtin_ctzll(idle);
        const uint32_t band_total = kPrimary ? band_tiles_of(f, a.tiles_per_band, band) * 64u : a.count_in[band * kWfStride];
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(a.heads + band * kWfStride, n);
        base = (uint32_t)__shfl((int)base, leader);
        const uint32_t idx =
            base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (status == ST_IDLE && idx < band_total) {
          slot = band * cap + idx;
          const uint4 q0 = a.rays_in[2 * (size_t)slot];
          const uint4 q1 = a.rays_in[2 * (size_t)slot + 1];
          if (q0.x != 0xffffffffu) {
            float t_start = 0.0f;
            if (kPrimary && f.use_beam) {   // pixel coordinates from the output index, as the shade kernel does
              const int oyl = (int)(q0.x / (uint32_t)f.width) - f.out_y0;
              t_start = beam_start(f, (int)(q0.x % (uint32_t)f.width), frame_gy(f, oyl >> 3, oyl & 7));
            }
            status = trav_init(root, t, mk(__uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)),
                               mk(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)), (q1.w & 1u) != 0u,
                               t_start);
          }
        }
        if (base + n >= band_total) {  // band used up: steal from the next one
          band = (band + 1u) & 7u;
          bands_left--;
        }
      }
    }
    if (__ballot(status != ST_IDLE) == 0ull) {
      if (bands_left > 0) continue;
      break;
    }
    const int active0 = __builtin_popcountll(__ballot(status == ST_ACTIVE));
    const int threshold = bands_left > 0 ? (active0 * a.thresh_num) / 16 : 0;
    for (;;) {
#ifdef SVO_STAMPS
      { unsigned long long dummy = 0; if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t, dummy); }
#else
      if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t);
#endif
      if (__builtin_popcountll(__ballot(status == ST_ACTIVE)) <= threshold) break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// shade: one thread per traced ray
__device__ __forceinline__ void wf_emit(const WfArgs &a, uint32_t pix, int px, int py, V3 col, bool set_depth, float depth) {
  if (a.f.spp <= 1 && !a.f.progressive) {
    if (px < 10 && py < 10) col = a.f.dword0 == 0u ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
    a.color[pix] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xff000000u;
  } else {
    float *fx = a.facc + pix, *fy = a.facc + a.npix + pix, *fz = a.facc + 2 * a.npix + pix;
    if (a.sample == 0) { *fx = 0.0f + col.x; *fy = 0.0f + col.y; *fz = 0.0f + col.z; }
    else { *fx = *fx + col.x; *fy = *fy + col.y; *fz = *fz + col.z; }
  }
  if (set_depth && a.sample == 0) a.depth[pix] = depth;
}

// append (q0, q1) of every lane with `want` to band `band`'s queue for the next stage
__device__ __forceinline__ void wf_append(const WfArgs &a, bool want, uint32_t band, uint32_t cap, uint4 q0, uint4 q1) {
  // lanes of a wave may belong to two bands at a band boundary: handle each band present in the wave
  unsigned long long todo = __ballot(want);
  while (todo != 0ull) {
    const int first = __builtin_ctzll(todo);
    const uint32_t b = (uint32_t)__shfl((int)band, first);
    const unsigned long long m = __ballot(want && band == b);
    const uint32_t n = (uint32_t)__builtin_popcountll(m);
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63u) == first) base = atomicAdd(a.count_out + b * kWfStride, n);
    base = (uint32_t)__shfl((int)base, first);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (want && band == b) {
      const size_t s = (size_t)b * cap + base + rank;
      a.rays_out[2 * s] = q0;
      a.rays_out[2 * s + 1] = q1;
    }
    todo &= ~m;
  }
}

template <int kMode, bool kPrimary>
__global__ __launch_bounds__(256) void wf_shade_kernel(const WfArgs a) {
  const Frame &f = a.f;
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  const uint32_t cap = (uint32_t)a.tiles_per_band * 64u;
  const uint32_t band = slot / cap < 8u ? slot / cap : 7u;
  const uint32_t idx = slot - band * cap;
  const uint32_t total = slot / cap >= 8u ? 0u
                         : (kPrimary ? band_tiles_of(f, a.tiles_per_band, band) * 64u : a.count_in[band * kWfStride]);
  bool valid = idx < total;
  uint4 q0 = make_uint4(0xffffffffu, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
  if (valid) {
    q0 = a.rays_in[2 * (size_t)slot];
    q1 = a.rays_in[2 * (size_t)slot + 1];
    valid = q0.x != 0xffffffffu;
  }
  bool cont = false;
  uint4 n0 = q0, n1 = q1;
  if (valid) {
    const uint32_t pix = q0.x;
    const uint4 r0 = a.results[2 * (size_t)slot], r1 = a.results[2 * (size_t)slot + 1];
    // rebuild what trav_result needs
    Trav t;
    t.cptr = r0.x; t.rlo = r0.y; t.rhi = r0.z; t.t_min = __uint_as_float(r0.w);
    t.px = __uint_as_float(r1.x); t.py = __uint_as_float(r1.y); t.pz = __uint_as_float(r1.z);
    t.iter = r1.w & 0x7ffu;
    t.scale = (int)((r1.w >> 11) & 0x3fu);
    if (t.scale >= 32) t.scale -= 64;  // 6-bit two's complement (-1 .. 30)
    t.octant = (r1.w >> 17) & 7u; t.tag = (r1.w >> 20) & 3u;
    const int status = (int)(r1.w >> 22);
    t.sexp = __uint_as_float(((uint32_t)t.scale - (uint32_t)kMaxScale + 127u) << 23);
    const Cast c = trav_result(t, status);
    // pixel coordinates from the output index
    const int oyl = (int)(pix / (uint32_t)f.width) - f.out_y0;
    const int px = (int)(pix % (uint32_t)f.width);
    const int py = frame_gy(f, oyl >> 3, oyl & 7);
    const V3 d = mk(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
    const size_t n = a.npix;
    float *st = a.state + pix;
    if (kPrimary && f.write_hits && a.sample == 0) {
      uint4 h;
      h.x = c.hit ? c.pointer : 0u;
      h.y = c.hit ? ((c.raw & 0xffffu) | ((c.value & 0xffu) << 16) | ((c.level & 0xffu) << 24)) : 0u;
      h.z = c.iter;
      h.w = c.hit ? __float_as_uint(c.t) : 0u;
      if (kMode == 4) h = make_uint4(0u, 0u, 0u, 0u);   // trace() casts nothing in modes >= 4 (svotrace.comp:643-646)
      a.hits[pix] = h;
    }
    if (kMode == 0) {
      if (kPrimary && !c.hit) {
        const V3 s = sky_colour(d);
        wf_emit(a, pix, px, py, mk(0.0f + s.x, 0.0f + s.y, 0.0f + s.z), true, 0.0f);
      } else {
        V3 mask = mk(1.f, 1.f, 1.f), accum = mk(0.f, 0.f, 0.f), normal = mk(0.f, 0.f, 0.f), vpos = mk(0.f, 0.f, 0.f);
        uint32_t value = 0;
        const float r = st[P_R * n];
        if (!kPrimary) {
          mask = mk(st[P_MX * n], st[P_MY * n], st[P_MZ * n]);
          accum = mk(st[P_AX * n], st[P_AY * n], st[P_AZ * n]);
          normal = mk(st[P_NX * n], st[P_NY * n], st[P_NZ * n]);
          value = __float_as_uint(st[P_VAL * n]);
        }
        if (c.hit) { normal = c.normal; value = c.value; vpos = c.voxel_pos; }
        const V3 nd = scatter(d, normal, r, ((f.mirror_mask >> (value & 31u)) & 1u) != 0u);
        if (c.hit) {
          const V3 mc = material_colour(value, mk(vpos.x - 1.0f, vpos.y - 1.0f, vpos.z - 1.0f));
          accum = mk(accum.x + mask.x * 0.0f, accum.y + mask.y * 0.0f, accum.z + mask.z * 0.0f);
          mask = mk(mask.x * mc.x, mask.y * mc.y, mask.z * mc.z);
          const float k = dot3(nd, normal);
          mask = mk(mask.x * k, mask.y * k, mask.z * k);
          if (a.segment + 1 >= f.bounces) {
            wf_emit(a, pix, px, py, accum, true, c.t);
          } else {
            if (a.sample == 0) a.depth[pix] = c.t;
            st[P_MX * n] = mask.x; st[P_MY * n] = mask.y; st[P_MZ * n] = mask.z;
            st[P_AX * n] = accum.x; st[P_AY * n] = accum.y; st[P_AZ * n] = accum.z;
            st[P_NX * n] = normal.x; st[P_NY * n] = normal.y; st[P_NZ * n] = normal.z;
            st[P_VAL * n] = __uint_as_float(value);
            n0 = make_uint4(pix, __float_as_uint(vpos.x), __float_as_uint(vpos.y), __float_as_uint(vpos.z));
            n1 = make_uint4(__float_as_uint(nd.x), __float_as_uint(nd.y), __float_as_uint(nd.z), 1u);  // cone ray
            cont = true;
          }
        } else {
          const V3 sun = normalize3(mk(1.0f, 1.0f, 1.0f));
          const float diff = acos_pinned(dot3(nd, sun));
          if (diff < 0.4f) accum = mk(accum.x + mask.x * 7.0f, accum.y + mask.y * 7.0f, accum.z + mask.z * 7.0f);
          accum = mk(accum.x + mask.x * 1.0f, accum.y + mask.y * 1.0f, accum.z + mask.z * 1.0f);
          wf_emit(a, pix, px, py, accum, true, 0.0f);
        }
      }
    } else if (kMode == 1) {
      V3 col;
      if (c.hit) { const float g = 0.005f * (float)c.iter; col = mk(g, g, g); }
      else if (c.capped) col = mk(0.3f, 0.3f, 0.6f);
      else { const float g = 0.01f * (float)c.iter; col = mk(g, g, g); }
      wf_emit(a, pix, px, py, col, true, c.hit ? c.t : 0.0f);
    } else if (kMode == 2) {
      const V3 sun2 = normalize3(mk(0.5f, 0.5f, 0.5f));
      if (kPrimary) {
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
          if (a.sample == 0) a.depth[pix] = c.t;
          st[P_MX * n] = mc.x; st[P_MY * n] = mc.y; st[P_MZ * n] = mc.z;
          n0 = make_uint4(pix, __float_as_uint(c.voxel_pos.x), __float_as_uint(c.voxel_pos.y), __float_as_uint(c.voxel_pos.z));
          n1 = make_uint4(__float_as_uint(sun2.x), __float_as_uint(sun2.y), __float_as_uint(sun2.z), 0u);
          cont = true;
        } else {
          wf_emit(a, pix, px, py, sky_colour(d), true, 0.0f);
        }
      } else {
        V3 mc = mk(st[P_MX * n], st[P_MY * n], st[P_MZ * n]);
        if (c.hit && c.t > c.scale_exp2 * 1.73205080757f) {
          mc = mk(mc.x - 0.2f, mc.y - 0.2f, mc.z - 0.2f);
        } else if (c.iter > 260u) {
          const float pen = (0.05f * (float)c.iter) / 100.0f;
          mc = mk(mc.x - pen, mc.y - pen, mc.z - pen);
        }
        wf_emit(a, pix, px, py, mc, false, 0.0f);
      }
    } else if (kMode == 3) {
      if (c.hit) wf_emit(a, pix, px, py, mk(c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f, c.normal.z * 0.5f + 0.5f), true, c.t);
      else wf_emit(a, pix, px, py, mk(0.f, 0.f, 0.f), true, 0.0f);
    } else {
      wf_emit(a, pix, px, py, mk(0.f, 0.f, 0.f), true, 0.0f);
    }
  }
  wf_append(a, cont, band, cap, n0, n1);
}

// spp > 1: colour sums -> rgba8
__global__ void resolve_kernel(const Frame f, const float *facc, size_t npix, uint32_t *color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = frame_gy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7);
  if (x >= f.width || y >= f.y1 || y >= f.height) return;
  const size_t pix = (size_t)frame_oy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7) * f.width + x;
  const float inv = 1.0f / (float)f.spp;
  const V3 col = mk(facc[pix] * inv, facc[npix + pix] * inv, facc[2 * npix + pix] * inv);
  color[pix] = final_rgba8(f, x, y, col, color + pix);
}

inline int wavefront_prepare(WavefrontBuffers &b, const Frame &f, size_t out_npix) {
  const size_t npix = out_npix;
  const size_t tiles_per_band = (size_t)(f.ntiles + 7) / 8;
  const size_t slots = tiles_per_band * 8 * 64;
  if (b.npix < npix || b.slots < slots || !b.set[0].state) {
    hipError_t e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;   // frames in flight may still use the old buffers
    wavefront_free(b);
    for (auto &s : b.set) {
      if ((e = hipMalloc((void **)&s.rays[0], slots * 32)) != hipSuccess) return (int)e;
      if ((e = hipMalloc((void **)&s.rays[1], slots * 32)) != hipSuccess) return (int)e;
      if ((e = hipMalloc((void **)&s.results, slots * 32)) != hipSuccess) return (int)e;
      if ((e = hipMalloc((void **)&s.state, npix * kStatePlanes * sizeof(float))) != hipSuccess) return (int)e;
      if ((e = hipMalloc((void **)&s.counters, (size_t)(kWfMaxStages + 1) * 16 * kWfStride * 4)) != hipSuccess) return (int)e;
      if ((e = hipMalloc((void **)&s.facc, npix * 3 * sizeof(float))) != hipSuccess) return (int)e;
      if ((e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return (int)e;
    }
    b.npix = npix; b.slots = slots;
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wf_trace_kernel<true>, 64, 0) != hipSuccess || per_cu < 1)
      per_cu = 16;
    b.max_per_cu = per_cu; b.cus = cus;
    if (const char *e1 = getenv("SVO_WF_WAVES_PER_CU")) b.waves_per_cu = atoi(e1);
    if (const char *e2 = getenv("SVO_WF_THRESH")) b.thresh_num = atoi(e2);
  }
  b.blocks = b.cus * (b.waves_per_cu > 0 ? b.waves_per_cu : b.max_per_cu);
  return 0;
}

template <int kMode>
inline void wf_launch_shade(bool primary, const WfArgs &a, unsigned grid, hipStream_t stream) {
  if (primary) hipLaunchKernelGGL((wf_shade_kernel<kMode, true>), dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((wf_shade_kernel<kMode, false>), dim3(grid), dim3(256), 0, stream, a);
}

// One frame: gen, then trace / shade per path segment, repeated per sample.  Everything is enqueued on
// `stream`; queue sizes never visit the host.  `out_npix` = elements of the output images (state planes).
inline int wavefront_launch(WavefrontBuffers &b, const uint8_t *pool, const Frame &f, uint32_t *color, float *depth,
                            uint4 *hits, size_t out_npix, hipStream_t stream) {
  int rc = wavefront_prepare(b, f, out_npix);
  if (rc) return rc;
  WfSet &S = b.set[b.launches++ % kWfSets];
  const int spp = f.spp < 1 ? 1 : f.spp;
  int nsec = 0;
  if (f.render_mode == 0) nsec = f.bounces - 1;
  else if (f.render_mode == 2) nsec = 1;
  if (nsec + 1 > kWfMaxStages) return (int)hipErrorInvalidValue;
  const int tiles_per_band = (f.ntiles + 7) / 8;
  const unsigned slot_blocks = (unsigned)(((size_t)tiles_per_band * 8 * 64 + 255) / 256);
  hipError_t e;
  if (S.used && (e = hipStreamWaitEvent(stream, S.done, 0)) != hipSuccess) return (int)e;
  for (int s = 0; s < spp; s++) {
    if ((e = hipMemsetAsync(S.counters, 0, (size_t)(kWfMaxStages + 1) * 16 * kWfStride * 4, stream)) != hipSuccess)
      return (int)e;
    WfArgs a;
    a.pool = pool; a.f = f; a.color = color; a.depth = depth; a.hits = hits;
    a.state = S.state; a.facc = S.facc; a.npix = b.npix; a.results = S.results;
    a.tiles_per_band = tiles_per_band; a.sample = s; a.thresh_num = b.thresh_num;
    // gen -> rays[0]
    a.rays_in = nullptr; a.rays_out = S.rays[0]; a.heads = nullptr; a.count_in = nullptr; a.count_out = nullptr; a.segment = 0;
    hipLaunchKernelGGL(wf_gen_kernel, dim3(slot_blocks), dim3(256), 0, stream, a);
    for (int k = 0; k <= nsec; k++) {
      a.segment = k;
      a.rays_in = S.rays[k & 1];
      a.rays_out = S.rays[(k + 1) & 1];
      a.heads = S.counters + (size_t)k * 16 * kWfStride;
      a.count_in = S.counters + (size_t)k * 16 * kWfStride + 8 * kWfStride;
      a.count_out = S.counters + (size_t)(k + 1) * 16 * kWfStride + 8 * kWfStride;
      const int blocks = f.ntiles < b.blocks ? f.ntiles : b.blocks;
      if (k == 0) hipLaunchKernelGGL(wf_trace_kernel<true>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
      else hipLaunchKernelGGL(wf_trace_kernel<false>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
      switch (f.render_mode) {
        case 0: wf_launch_shade<0>(k == 0, a, slot_blocks, stream); break;
        case 1: wf_launch_shade<1>(k == 0, a, slot_blocks, stream); break;
        case 2: wf_launch_shade<2>(k == 0, a, slot_blocks, stream); break;
        case 3: wf_launch_shade<3>(k == 0, a, slot_blocks, stream); break;
        default: wf_launch_shade<4>(k == 0, a, slot_blocks, stream); break;
      }
      if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    }
  }
  if (spp > 1 || f.progressive) {
    dim3 grid((unsigned)((f.width + 255) / 256), (unsigned)(f.tiles_y * 8));
    hipLaunchKernelGGL(resolve_kernel, grid, dim3(256), 0, stream, f, S.facc, b.npix, color);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  if ((e = hipEventRecord(S.done, stream)) != hipSuccess) return (int)e;
  S.used = true;
  return 0;
}

}  // namespace svo
