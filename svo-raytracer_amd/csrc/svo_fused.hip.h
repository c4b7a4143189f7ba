// svo_fused.hip.h -- pipeline 0: one wavefront per 8x8 pixel tile, whole path per lane.
//
// Same work decomposition as the reference dispatch (local_size 8x8 = 64 threads = exactly
// one CDNA wavefront, svotrace.comp:648; grid ceil(W/8) x ceil(H/8), Main.java:108-109,285),
// but tiles are handed to workgroups so that each XCD (own L2) walks one contiguous
// band of the screen.
#pragma once
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

struct PathState {
  V3 colour;
  float depth;
  // first cast of the pixel
  bool hit; uint32_t pointer, value, raw, level, iter; float t;
};

// start distance of pixel (px, py)'s primary ray: its block's entry of the beam image, or 0
__device__ __forceinline__ float beam_start(const Frame &f, int px, int py) {
  return f.use_beam ? f.beam[(size_t)(py >> 2) * (size_t)f.beam_w + (size_t)(px >> 2)] : 0.0f;
}

// What main() does with a pixel's colour before imageStore (svotrace.comp:696-726): the debug square, then -- when the
// dormant cross-frame accumulation (:712-719) is switched on -- the running mean with the image the previous frame left
// in the colour buffer: (frameNumber * last + colour) / (frameNumber + 1), frozen from MAX_FRAME_ITER = 100 on.
// imageLoad of rgba8 = byte * (1/255); true division (both pinned by tests/golden/accum_golden.npz).
// `last`: the texel the previous frame left (only looked at when the accumulation is on); frame_number: this frame's
__device__ __forceinline__ uint32_t final_rgba8_v(const Frame &f, int frame_number, int px, int py, V3 col, uint32_t last) {
  if (px < 10 && py < 10) col = f.dword0 == 0u ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
  if (f.progressive && frame_number > 1) {
    const V3 lc = mk((float)(last & 0xffu) * (1.0f / 255.0f), (float)((last >> 8) & 0xffu) * (1.0f / 255.0f),
                     (float)((last >> 16) & 0xffu) * (1.0f / 255.0f));
    if (frame_number < 100) {
      const float fn = (float)frame_number, fd = (float)(frame_number + 1);
      col = mk((fn * lc.x + col.x) / fd, (fn * lc.y + col.y) / fd, (fn * lc.z + col.z) / fd);
    } else {
      col = lc;
    }
  }
  return unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xff000000u;
}
__device__ __forceinline__ uint32_t final_rgba8(const Frame &f, int px, int py, V3 col, const uint32_t *dst) {
  return final_rgba8_v(f, f.frame_number, px, py, col, (f.progressive && f.frame_number > 1) ? *dst : 0u);
}

__device__ __forceinline__ V3 sky_colour(V3 d) {
  return mk(0.6725f - d.y * 0.4f, 0.8784f - d.y * 0.4f, 1.0f - d.y * 0.25f);
}

__device__ __forceinline__ void record_first(PathState &ps, const Cast &c) {
  ps.hit = c.hit;
  ps.pointer = c.hit ? c.pointer : 0u;
  ps.value = c.hit ? c.value : 0u;
  ps.raw = c.hit ? c.raw : 0u;
  ps.level = c.hit ? c.level : 0u;
  ps.iter = c.iter;
  ps.t = c.hit ? c.t : 0.0f;
}

// Diffuse / mirror continuation of svotrace.comp:476-509 given the (possibly stale) normal.
__device__ __forceinline__ V3 scatter(V3 d, V3 normal, float r, bool mirror) {
  if (mirror) {
    const float k = 2.0f * dot3(d, normal);
    return mk(d.x - k * normal.x, d.y - k * normal.y, d.z - k * normal.z);
  }
  const float rand1 = (2.0f * 3.14159265359f) * r;
  const V3 w = normal;
  const V3 axis = __builtin_fabsf(w.x) > 0.1f ? mk(0.f, 1.f, 0.f) : mk(1.f, 0.f, 0.f);
  const V3 u = normalize3(cross3(axis, w));
  const V3 v = cross3(w, u);
  const float cs = sincos_pinned<true>(rand1), sn = sincos_pinned<false>(rand1), om = 1.0f - r;
  return normalize3(mk((u.x * cs + v.x * sn) + w.x * om, (u.y * cs + v.y * sn) + w.y * om,
                       (u.z * cs + v.z * sn) + w.z * om));
}

// trace() of svotrace.comp:435-646 for one sample.
template <bool kCount>
__device__ __forceinline__ void trace_sample(const Pool &pool, WaveStack &stk, uint32_t lane, const Frame &f, V3 o,
                                             V3 d, float seed0, float seed1, float seed2, PathState &ps, bool first,
                                             V3 &out_colour, float &out_depth, Counters &cnt, const float t_start) {
  const int mode = f.render_mode;
  V3 colour = mk(0.f, 0.f, 0.f);
  float depth = 0.0f;
  if (mode == 0) {
    V3 accum = mk(0.f, 0.f, 0.f), mask = mk(1.f, 1.f, 1.f);
    // fields of the reference's `res` that survive a missed cast
    V3 normal = mk(0.f, 0.f, 0.f), vpos = mk(0.f, 0.f, 0.f);
    uint32_t value = 0;
    const float r = pixel_rand(seed0, seed1, seed2);
    for (int i = 0; i < f.bounces; i++) {
      const Cast c = cast_ray<kCount>(pool, stk, lane, o, d, kMaxDepth, i != 0, cnt, i == 0 ? t_start : 0.0f);
      if (i == 0 && first) record_first(ps, c);
      if (!c.hit && i == 0) {
        const V3 s = sky_colour(d);
        accum = mk(accum.x + s.x, accum.y + s.y, accum.z + s.z);
        break;
      }
      if (c.hit) { normal = c.normal; vpos = c.voxel_pos; value = c.value; }
      const V3 nd = scatter(d, normal, r, ((f.mirror_mask >> (value & 31u)) & 1u) != 0u);
      o = vpos;
      d = nd;
      const V3 mc = material_colour(value, mk(vpos.x - 1.0f, vpos.y - 1.0f, vpos.z - 1.0f));
      if (c.hit) {
        depth = c.t;
        accum = mk(accum.x + mask.x * 0.0f, accum.y + mask.y * 0.0f, accum.z + mask.z * 0.0f);
        mask = mk(mask.x * mc.x, mask.y * mc.y, mask.z * mc.z);
        const float k = dot3(nd, normal);
        mask = mk(mask.x * k, mask.y * k, mask.z * k);
      } else {
        const V3 sun = normalize3(mk(1.0f, 1.0f, 1.0f));
        const float diff = acos_pinned(dot3(d, sun));
        if (diff < 0.4f) accum = mk(accum.x + mask.x * 7.0f, accum.y + mask.y * 7.0f, accum.z + mask.z * 7.0f);
        accum = mk(accum.x + mask.x * 1.0f, accum.y + mask.y * 1.0f, accum.z + mask.z * 1.0f);
        depth = 0.0f;
        break;
      }
    }
    colour = accum;
  } else if (mode == 1) {
    const Cast c = cast_ray<kCount>(pool, stk, lane, o, d, kMaxDepth, false, cnt, t_start);
    if (first) record_first(ps, c);
    depth = c.hit ? c.t : 0.0f;
    if (c.hit) { const float g = 0.005f * (float)c.iter; colour = mk(g, g, g); }
    else if (c.capped) colour = mk(0.3f, 0.3f, 0.6f);
    else { const float g = 0.01f * (float)c.iter; colour = mk(g, g, g); }
  } else if (mode == 2) {
    const Cast c = cast_ray<kCount>(pool, stk, lane, o, d, kMaxDepth, false, cnt, t_start);
    if (first) record_first(ps, c);
    if (c.hit) {
      depth = c.t;
      V3 mc = material_colour(c.value, kMode2OtherMaterial);
      const V3 sun = normalize3(mk(0.5f, 0.5f, 0.5f));
      const float k = (c.level >= 10u ? dot3(c.normal, sun) : dot3(mk(0.f, 1.0f, 0.f), sun)) * 0.1f;
      mc = mk(mc.x + k, mc.y + k, mc.z + k);
      const float dist = c.t + 0.0f;
      const float lg = exp2_pinned(dist * (-0.5f * 2.0f * 1.44269504f));
      const float lb = exp2_pinned(dist * (-0.5f * 4.0f * 1.44269504f));
      const float lr = exp2_pinned(dist * (-0.5f * 1.0f * 1.44269504f));
      mc.x = lr * mc.x + (1.0f - lr) * 1.0f;
      mc.y = lg * mc.y + (1.0f - lg) * 1.0f;
      mc.z = lb * mc.z + (1.0f - lb) * 1.0f;
      const Cast s = cast_ray<kCount>(pool, stk, lane, c.voxel_pos, sun, kMaxDepth, false, cnt);
      if (s.hit && s.t > s.scale_exp2 * 1.73205080757f) {
        mc = mk(mc.x - 0.2f, mc.y - 0.2f, mc.z - 0.2f);
      } else if (s.iter > 260u) {
        const float pen = (0.05f * (float)s.iter) / 100.0f;
        mc = mk(mc.x - pen, mc.y - pen, mc.z - pen);
      }
      colour = mc;
    } else {
      depth = 0.0f;
      colour = sky_colour(d);
    }
  } else if (mode == 3) {
    const Cast c = cast_ray<kCount>(pool, stk, lane, o, d, kMaxDepth, false, cnt, t_start);
    if (first) record_first(ps, c);
    if (c.hit) {
      depth = c.t;
      colour = mk(c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f, c.normal.z * 0.5f + 0.5f);
    }
  }
  out_colour = colour;
  out_depth = depth;
}

// c: the 15 camera floats (pos, l1, l2, r1, r2), any address space
template <class CamPtr>
__device__ __forceinline__ V3 primary_direction_cam(CamPtr c, int width, int height, int px, int py) {
  const float u = ((float)px + 0.5f) / (float)width;
  const float v = ((float)py + 0.5f) / (float)height;
  const V3 a = mk(mix_g(c[3], c[6], v), mix_g(c[4], c[7], v), mix_g(c[5], c[8], v));
  const V3 b = mk(mix_g(c[9], c[12], v), mix_g(c[10], c[13], v), mix_g(c[11], c[14], v));
  return normalize3(mk(mix_g(a.x, b.x, u), mix_g(a.y, b.y, u), mix_g(a.z, b.z, u)));
}
__device__ __forceinline__ V3 primary_direction(const Frame &f, int px, int py) {
  return primary_direction_cam(f.cam, f.width, f.height, px, py);
}

// blockIdx -> tile so that the 8 XCDs (blocks are dealt to them round-robin) each take a
// contiguous band of tiles: neighbouring tiles share subtrees, and each XCD has its own L2.
__device__ __forceinline__ int xcd_tile(int b, int ntiles) {
  const int per = (ntiles + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

__device__ __forceinline__ void store_pixel(const Frame &f, int px, int py, int oy, V3 fin, float depth,
                                            const PathState &ps, uint32_t *color, float *depthbuf, uint4 *hits) {
  const size_t o = (size_t)oy * (size_t)f.width + (size_t)px;
  color[o] = final_rgba8(f, px, py, fin, color + o);
  depthbuf[o] = depth;
  if (f.write_hits) {
    uint4 h;
    h.x = ps.pointer;
    h.y = (ps.raw & 0xffffu) | ((ps.value & 0xffu) << 16) | ((ps.level & 0xffu) << 24);
    h.z = ps.iter;
    h.w = __float_as_uint(ps.t);
    hits[o] = h;
  }
}

template <bool kCount>
__global__ __launch_bounds__(64) void trace_fused_kernel(const uint8_t *__restrict__ pool_base, const Frame f,
                                                         uint32_t *__restrict__ color, float *__restrict__ depthbuf,
                                                         uint4 *__restrict__ hits, DeviceCounters *counters) {
  __shared__ WaveStack stk;
  const int tile = xcd_tile((int)blockIdx.x, f.ntiles);
  if (tile >= f.ntiles) return;
  const uint32_t lane = threadIdx.x;
  const int tx = tile % f.tiles_x, ty = tile / f.tiles_x;
  const int px = tx * 8 + (int)(lane & 7u);
  const int py = frame_gy(f, ty, (int)(lane >> 3));
  const int oy = frame_oy(f, ty, (int)(lane >> 3));
  Counters cnt = {0, 0, 0, 0, 0};
  const bool live = px < f.width && py < f.y1 && py < f.height;
  if (live) {
    Pool pool;
    pool.base = pool_base;
    pool.len = f.pool_len;
    const V3 o = mk(f.cam[0], f.cam[1], f.cam[2]);
    const V3 d = primary_direction(f, px, py);
    PathState ps;
    ps.hit = false; ps.pointer = 0; ps.value = 0; ps.raw = 0; ps.level = 0; ps.iter = 0; ps.t = 0.0f;
    V3 fin = mk(0.f, 0.f, 0.f);
    float depth = 0.0f;
    const int spp = f.spp < 1 ? 1 : f.spp;
    const float t_start = beam_start(f, px, py);
    for (int s = 0; s < spp; s++) {
      V3 col;
      float dep;
      trace_sample<kCount>(pool, stk, lane, f, o, d, (float)px, (float)py, (float)(f.frame_number + s), ps, s == 0, col,
                           dep, cnt, t_start);
      if (s == 0) depth = dep;
      fin = mk(fin.x + col.x, fin.y + col.y, fin.z + col.z);
    }
    if (spp > 1) {
      const float inv = 1.0f / (float)spp;
      fin = mk(fin.x * inv, fin.y * inv, fin.z * inv);
    }
    store_pixel(f, px, py, oy, fin, depth, ps, color, depthbuf, hits);
  }
  if (kCount) {
    // wave-level reduction, one atomic per counter per wave
    unsigned long long rays = cnt.rays, nans = cnt.nan_rays, its = cnt.iters, by = cnt.bytes, pix = live ? 1ull : 0ull;
    unsigned int mx = cnt.max_iter;
    for (int off = 32; off > 0; off >>= 1) {
      rays += __shfl_down(rays, off);
      nans += __shfl_down(nans, off);
      its += __shfl_down(its, off);
      by += __shfl_down(by, off);
      pix += __shfl_down(pix, off);
      const unsigned int m2 = __shfl_down(mx, off);
      mx = m2 > mx ? m2 : mx;
    }
    if (lane == 0) {
      atomicAdd(&counters->pixels, pix);
      atomicAdd(&counters->rays, rays);
      atomicAdd(&counters->nan_rays, nans);
      atomicAdd(&counters->iterations, its);
      atomicAdd(&counters->alg_bytes, by);
      atomicMax(&counters->max_iter, mx);
    }
  }
}

// The pick pixel of a frame (svo_set_pick; the crosshair read-back of Main.updateEarly, Main.java:132-146), by a launch of its
// own: ONE lane walks the pixel's whole path -- the same device functions, on the same values, as every pipeline's kernels, so the
// same bits (tests/test_gpu_pick.py) -- and writes {rgba8, depth, hit record}, then the dispatch's sequence number, to host memory
// the device can write.  svo_read_pixel polls that word instead of waiting for the frame.  A launch of one wave on a stream of
// its own: it needs ONE free wave slot, not the frame's turn on the GPU, and the frame's kernels carry nothing for it (a pick
// inside persist_kernel -- its tile drawn first, the lane that stores the pixel writing the mail -- was built first and cost the
// throughput configuration 1.4 % through eleven more spilled SGPRs: profiles/round6_experiments.txt).
constexpr int kPickWords = 8;   // seq, rgba8, depth bits, 0, hit[4]
constexpr int kPickSlots = 8;   // mail slots, one per dispatch, re-used round-robin
__global__ __launch_bounds__(64) void pick_kernel(const uint8_t *__restrict__ pool_base, const Frame f, const int px, const int py,
                                                   uint32_t *mail, const uint32_t seq) {
  __shared__ WaveStack stk;
  const uint32_t lane = threadIdx.x;
  if (lane != 0u) return;
  Pool pool;
  pool.base = pool_base;
  pool.len = f.pool_len;
  Counters cnt = {0, 0, 0, 0, 0};
  PathState ps;
  ps.hit = false; ps.pointer = 0; ps.value = 0; ps.raw = 0; ps.level = 0; ps.iter = 0; ps.t = 0.0f;
  V3 col;
  float depth;
  trace_sample<false>(pool, stk, lane, f, mk(f.cam[0], f.cam[1], f.cam[2]), primary_direction(f, px, py), (float)px, (float)py,
                      (float)f.frame_number, ps, true, col, depth, cnt, 0.0f);
  mail[1] = final_rgba8(f, px, py, col, nullptr);   // (one sample, no accumulation: the live shader's store, debug square included)
  mail[2] = __float_as_uint(depth);
  mail[3] = 0u;
  mail[4] = ps.pointer;
  mail[5] = (ps.raw & 0xffffu) | ((ps.value & 0xffu) << 16) | ((ps.level & 0xffu) << 24);
  mail[6] = ps.iter;
  mail[7] = __float_as_uint(ps.t);
  __threadfence_system();
  __hip_atomic_store(mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace svo
