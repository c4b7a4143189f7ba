// svo_wavefront.hip.h -- pipeline 1: wavefront path tracing with persistent waves.
//
// The reference runs one thread per pixel through primary cast -> shading -> secondary
// cast (svotrace.comp:435-646, 649-729).  A 64-lane wavefront then idles on its slowest
// ray twice per pixel (iteration counts: mean ~70, tail > 300 at 8192^3).  Here the frame is
// processed as STAGES of rays instead:
//   stage 0  primary rays of every pixel, drawn in 8x8-tile order from an implicit queue
//   stage k  the surviving secondary rays (diffuse / mirror bounce, or shadow ray),
//            drawn from a queue that stage k-1 compacted with wave ballots
// Every stage is one persistent kernel: a wave keeps 64 traversals in flight; when enough
// lanes have finished, those lanes shade their hit together (one SIMD-efficient pass),
// append continuing paths to the next queue (ballot + prefix count, one atomic per wave)
// and pull fresh rays (same ballot/prefix scheme on the queue head).  Path state that
// outlives a stage lives in per-pixel SoA arrays in HBM.
// Results are bit-identical to pipeline 0 and to the oracle: the same cast / shading
// arithmetic is executed per ray, only the scheduling differs.
#pragma once
#include "svo_device.h"
#include "svo_fused.hip.h"
#include "svo_kernels.h"
#include "svo_trav.h"

namespace svo {

constexpr int kStateFloats = 17;  // ox oy oz dx dy dz mask(3) accum(3) normal(3) r value
enum { S_OX = 0, S_OY, S_OZ, S_DX, S_DY, S_DZ, S_MX, S_MY, S_MZ, S_AX, S_AY, S_AZ, S_NX, S_NY, S_NZ, S_R, S_VAL };

struct WavefrontBuffers {
  float *state = nullptr;       // kStateFloats arrays of npix floats
  uint32_t *queue[2] = {nullptr, nullptr};
  uint32_t *counters = nullptr; // [0] stage head, [1] queue A count, [2] queue B count, [3] spare; x stages
  float *facc = nullptr;        // spp > 1: per-pixel colour sums (3 arrays)
  size_t npix = 0;
  int blocks = 0;
};

inline void wavefront_free(WavefrontBuffers &b) {
  if (b.state) (void)hipFree(b.state);
  if (b.queue[0]) (void)hipFree(b.queue[0]);
  if (b.queue[1]) (void)hipFree(b.queue[1]);
  if (b.counters) (void)hipFree(b.counters);
  if (b.facc) (void)hipFree(b.facc);
  b = WavefrontBuffers();
}

struct StageArgs {
  const uint8_t *pool;
  Frame f;
  uint32_t *color;
  float *depth;
  uint4 *hits;
  float *state;
  float *facc;
  size_t npix;
  const uint32_t *queue_in;   // stage >= 1
  const uint32_t *count_in;   // stage >= 1: number of entries of queue_in
  uint32_t *queue_out;
  uint32_t *count_out;
  uint32_t *head;             // work counter of this stage
  int segment;                // 0 = primary, k = k-th secondary ray of the path
  int sample;                 // spp index
};

__device__ __forceinline__ void emit_pixel(const StageArgs &a, uint32_t pix, int px, int py, V3 col, bool set_depth,
                                           float depth) {
  if (a.f.spp <= 1) {
    if (px < 10 && py < 10) col = a.f.dword0 == 0u ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
    a.color[pix] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xff000000u;
  } else {
    float *fx = a.facc + pix, *fy = a.facc + a.npix + pix, *fz = a.facc + 2 * a.npix + pix;
    if (a.sample == 0) { *fx = 0.0f + col.x; *fy = 0.0f + col.y; *fz = 0.0f + col.z; }
    else { *fx = *fx + col.x; *fy = *fy + col.y; *fz = *fz + col.z; }
  }
  if (set_depth && a.sample == 0) a.depth[pix] = depth;
}

// append `pix` of every lane with `want` to the output queue: ballot + prefix, one atomic per wave
__device__ __forceinline__ void wave_enqueue(bool want, uint32_t pix, uint32_t *queue, uint32_t *count) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return;
  const uint32_t n = (uint32_t)__builtin_popcountll(m);
  const int leader = __builtin_ctzll(m);
  uint32_t base = 0;
  if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(count, n);
  base = (uint32_t)__shfl((int)base, leader);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (want) queue[base + rank] = pix;
}

template <bool kPrimary>
__global__ __launch_bounds__(64) void stage_kernel(const StageArgs a) {
  __shared__ WaveStack stk;
  const uint32_t lane = threadIdx.x;
  const Frame &f = a.f;
  const BufPool pool = make_bufpool(a.pool, f.pool_len);
  const uint64_t root = load_record(pool, 0u);
  const uint32_t total = kPrimary ? (uint32_t)f.ntiles * 64u : *a.count_in;
  const bool cone = !kPrimary && f.render_mode == 0;
  const V3 sun2 = normalize3(mk(0.5f, 0.5f, 0.5f));

  Trav t;
  int status = ST_IDLE;
  uint32_t pix = 0;
  int px = 0, py = 0;
  bool exhausted = false;

  for (;;) {
    // ---------------- shade every finished lane (together), hand continuing paths on
    const bool done = status >= ST_HIT;
    bool cont = false;
    if (done) {
      const Cast c = trav_result(t, status);
      if (kPrimary) {
        const V3 d = primary_direction(f, px, py);
        if (f.write_hits && a.sample == 0) {
          uint4 h;
          h.x = c.hit ? c.pointer : 0u;
          h.y = c.hit ? ((c.raw & 0xffffu) | ((c.value & 0xffu) << 16) | ((c.level & 0xffu) << 24)) : 0u;
          h.z = c.iter;
          h.w = c.hit ? __float_as_uint(c.t) : 0u;
          a.hits[pix] = h;
        }
        const int mode = f.render_mode;
        if (mode == 0) {
          if (!c.hit) {
            emit_pixel(a, pix, px, py, mk(0.0f + (0.6725f - d.y * 0.4f), 0.0f + (0.8784f - d.y * 0.4f),
                                          0.0f + (1.0f - d.y * 0.25f)), true, 0.0f);
          } else {
            const float r = pixel_rand((float)px, (float)py, (float)(f.frame_number + a.sample));
            const V3 nd = scatter(d, c.normal, r, ((f.mirror_mask >> (c.value & 31u)) & 1u) != 0u);
            const V3 mc = material_colour(c.value, mk(c.voxel_pos.x - 1.0f, c.voxel_pos.y - 1.0f, c.voxel_pos.z - 1.0f));
            const V3 accum = mk(0.0f + 1.0f * 0.0f, 0.0f + 1.0f * 0.0f, 0.0f + 1.0f * 0.0f);
            V3 mask = mk(1.0f * mc.x, 1.0f * mc.y, 1.0f * mc.z);
            const float k = dot3(nd, c.normal);
            mask = mk(mask.x * k, mask.y * k, mask.z * k);
            if (f.bounces <= 1) {
              emit_pixel(a, pix, px, py, accum, true, c.t);
            } else {
              if (a.sample == 0) a.depth[pix] = c.t;
              float *s = a.state + pix;
              const size_t n = a.npix;
              s[S_OX * n] = c.voxel_pos.x; s[S_OY * n] = c.voxel_pos.y; s[S_OZ * n] = c.voxel_pos.z;
              s[S_DX * n] = nd.x; s[S_DY * n] = nd.y; s[S_DZ * n] = nd.z;
              s[S_MX * n] = mask.x; s[S_MY * n] = mask.y; s[S_MZ * n] = mask.z;
              s[S_AX * n] = accum.x; s[S_AY * n] = accum.y; s[S_AZ * n] = accum.z;
              s[S_NX * n] = c.normal.x; s[S_NY * n] = c.normal.y; s[S_NZ * n] = c.normal.z;
              s[S_R * n] = r; s[S_VAL * n] = __uint_as_float(c.value);
              cont = true;
            }
          }
        } else if (mode == 1) {
          V3 col;
          if (c.hit) { const float g = 0.005f * (float)c.iter; col = mk(g, g, g); }
          else if (c.capped) col = mk(0.3f, 0.3f, 0.6f);
          else { const float g = 0.01f * (float)c.iter; col = mk(g, g, g); }
          emit_pixel(a, pix, px, py, col, true, c.hit ? c.t : 0.0f);
        } else if (mode == 2) {
          if (c.hit) {
            V3 mc = material_colour(c.value, mk(0.f, 0.f, 0.f));
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
            float *s = a.state + pix;
            const size_t n = a.npix;
            s[S_OX * n] = c.voxel_pos.x; s[S_OY * n] = c.voxel_pos.y; s[S_OZ * n] = c.voxel_pos.z;
            s[S_DX * n] = sun2.x; s[S_DY * n] = sun2.y; s[S_DZ * n] = sun2.z;
            s[S_MX * n] = mc.x; s[S_MY * n] = mc.y; s[S_MZ * n] = mc.z;
            cont = true;
          } else {
            emit_pixel(a, pix, px, py, sky_colour(d), true, 0.0f);
          }
        } else if (mode == 3) {
          if (c.hit) emit_pixel(a, pix, px, py, mk(c.normal.x * 0.5f + 0.5f, c.normal.y * 0.5f + 0.5f,
                                                   c.normal.z * 0.5f + 0.5f), true, c.t);
          else emit_pixel(a, pix, px, py, mk(0.f, 0.f, 0.f), true, 0.0f);
        } else {
          emit_pixel(a, pix, px, py, mk(0.f, 0.f, 0.f), true, 0.0f);
        }
      } else {
        // secondary ray finished
        float *s = a.state + pix;
        const size_t n = a.npix;
        if (f.render_mode == 2) {
          V3 mc = mk(s[S_MX * n], s[S_MY * n], s[S_MZ * n]);
          if (c.hit && c.t > c.scale_exp2 * 1.73205080757f) {
            mc = mk(mc.x - 0.2f, mc.y - 0.2f, mc.z - 0.2f);
          } else if (c.iter > 260u) {
            const float pen = (0.05f * (float)c.iter) / 100.0f;
            mc = mk(mc.x - pen, mc.y - pen, mc.z - pen);
          }
          emit_pixel(a, pix, px, py, mc, false, 0.0f);
        } else {
          const V3 d = mk(s[S_DX * n], s[S_DY * n], s[S_DZ * n]);
          V3 mask = mk(s[S_MX * n], s[S_MY * n], s[S_MZ * n]);
          V3 accum = mk(s[S_AX * n], s[S_AY * n], s[S_AZ * n]);
          V3 normal = mk(s[S_NX * n], s[S_NY * n], s[S_NZ * n]);
          V3 vpos = mk(s[S_OX * n], s[S_OY * n], s[S_OZ * n]);
          uint32_t value = __float_as_uint(s[S_VAL * n]);
          const float r = s[S_R * n];
          if (c.hit) { normal = c.normal; vpos = c.voxel_pos; value = c.value; }
          const V3 nd = scatter(d, normal, r, ((f.mirror_mask >> (value & 31u)) & 1u) != 0u);
          const V3 mc = material_colour(value, mk(vpos.x - 1.0f, vpos.y - 1.0f, vpos.z - 1.0f));
          if (c.hit) {
            accum = mk(accum.x + mask.x * 0.0f, accum.y + mask.y * 0.0f, accum.z + mask.z * 0.0f);
            mask = mk(mask.x * mc.x, mask.y * mc.y, mask.z * mc.z);
            const float k = dot3(nd, normal);
            mask = mk(mask.x * k, mask.y * k, mask.z * k);
            if (a.segment + 1 >= f.bounces) {
              emit_pixel(a, pix, px, py, accum, true, c.t);
            } else {
              if (a.sample == 0) a.depth[pix] = c.t;
              s[S_OX * n] = vpos.x; s[S_OY * n] = vpos.y; s[S_OZ * n] = vpos.z;
              s[S_DX * n] = nd.x; s[S_DY * n] = nd.y; s[S_DZ * n] = nd.z;
              s[S_MX * n] = mask.x; s[S_MY * n] = mask.y; s[S_MZ * n] = mask.z;
              s[S_AX * n] = accum.x; s[S_AY * n] = accum.y; s[S_AZ * n] = accum.z;
              s[S_NX * n] = normal.x; s[S_NY * n] = normal.y; s[S_NZ * n] = normal.z;
              s[S_VAL * n] = __uint_as_float(value);
              cont = true;
            }
          } else {
            const V3 sun = normalize3(mk(1.0f, 1.0f, 1.0f));
            const float diff = acos_pinned(dot3(nd, sun));
            if (diff < 0.4f) accum = mk(accum.x + mask.x * 7.0f, accum.y + mask.y * 7.0f, accum.z + mask.z * 7.0f);
            accum = mk(accum.x + mask.x * 1.0f, accum.y + mask.y * 1.0f, accum.z + mask.z * 1.0f);
            emit_pixel(a, pix, px, py, accum, true, 0.0f);
          }
        }
      }
      status = ST_IDLE;
    }
    wave_enqueue(cont, pix, a.queue_out, a.count_out);

    // ---------------- refill idle lanes from the stage's work queue (ballot + prefix)
    if (!exhausted) {
      const unsigned long long idle = __ballot(status == ST_IDLE);
      if (idle != 0ull) {
        const uint32_t n = (uint32_t)__builtin_popcountll(idle);
        const int leader = __builtin_ctzll(idle);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(a.head, n);
        base = (uint32_t)__shfl((int)base, leader);
        if (base + n >= total) exhausted = true;
        const uint32_t slot =
            base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (status == ST_IDLE && slot < total) {
          V3 o, d;
          bool live = true;
          if (kPrimary) {
            const int tile = (int)(slot >> 6);
            const uint32_t l = slot & 63u;
            px = (tile % f.tiles_x) * 8 + (int)(l & 7u);
            py = frame_gy(f, tile / f.tiles_x, (int)(l >> 3));
            live = px < f.width && py < f.y1 && py < f.height;
            pix = (uint32_t)frame_oy(f, tile / f.tiles_x, (int)(l >> 3)) * (uint32_t)f.width + (uint32_t)px;
            o = mk(f.cam[0], f.cam[1], f.cam[2]);
            d = live ? primary_direction(f, px, py) : mk(0.f, 0.f, 1.f);
          } else {
            pix = a.queue_in[slot];
            px = (int)(pix % (uint32_t)f.width);
            {
              const int oyl = (int)(pix / (uint32_t)f.width) - f.out_y0;
              py = frame_gy(f, oyl >> 3, oyl & 7);
            }
            const float *s = a.state + pix;
            const size_t n2 = a.npix;
            o = mk(s[S_OX * n2], s[S_OY * n2], s[S_OZ * n2]);
            d = mk(s[S_DX * n2], s[S_DY * n2], s[S_DZ * n2]);
          }
          if (live) status = trav_init(root, t, o, d, cone);
        }
      }
    }
    const unsigned long long busy = __ballot(status != ST_IDLE);
    if (busy == 0ull) break;

    // ---------------- traverse until enough lanes have finished to make a round worthwhile
    const int active0 = __builtin_popcountll(__ballot(status == ST_ACTIVE));
    const int threshold = exhausted ? 0 : (active0 * 5) / 8;  // refill once fewer than 5/8 of them remain
    for (;;) {
#ifdef SVO_STAMPS
      { unsigned long long dummy = 0; if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t, dummy); }
#else
      if (status == ST_ACTIVE) status = trav_step(pool, stk, lane, t);
#endif
      const int active = __builtin_popcountll(__ballot(status == ST_ACTIVE));
      if (active <= threshold) break;
    }
  }
}

// spp > 1: colour sums -> rgba8
__global__ void resolve_kernel(const Frame f, const float *facc, size_t npix, uint32_t *color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = frame_gy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7);
  if (x >= f.width || y >= f.y1 || y >= f.height) return;
  const size_t pix = (size_t)frame_oy(f, (int)blockIdx.y >> 3, (int)blockIdx.y & 7) * f.width + x;
  const float inv = 1.0f / (float)f.spp;
  V3 col = mk(facc[pix] * inv, facc[npix + pix] * inv, facc[2 * npix + pix] * inv);
  if (x < 10 && y < 10) col = f.dword0 == 0u ? mk(1.f, 0.f, 0.f) : mk(1.f, 1.f, 1.f);
  color[pix] = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xff000000u;
}

inline int wavefront_prepare(WavefrontBuffers &b, const Frame &f, hipStream_t stream) {
  const size_t npix = (size_t)f.width * (size_t)f.height;
  if (b.npix != npix || !b.state) {
    wavefront_free(b);
    hipError_t e;
    if ((e = hipMalloc((void **)&b.state, npix * kStateFloats * sizeof(float))) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&b.queue[0], npix * 4)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&b.queue[1], npix * 4)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&b.counters, 4096)) != hipSuccess) return (int)e;
    if ((e = hipMalloc((void **)&b.facc, npix * 3 * sizeof(float))) != hipSuccess) return (int)e;
    b.npix = npix;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stage_kernel<true>, 64, 0) != hipSuccess || per_cu < 1)
      per_cu = 16;
    b.blocks = cus * per_cu;
  }
  (void)stream;
  return 0;
}

// One frame: stage 0 + (bounces - 1) secondary stages (mode 0) or one shadow stage (mode 2),
// repeated per sample.  Everything is enqueued on `stream`; queue sizes never visit the host.
inline int wavefront_launch(WavefrontBuffers &b, const uint8_t *pool, const Frame &f, uint32_t *color, float *depth,
                            uint4 *hits, hipStream_t stream) {
  int rc = wavefront_prepare(b, f, stream);
  if (rc) return rc;
  const int spp = f.spp < 1 ? 1 : f.spp;
  int nsec = 0;
  if (f.render_mode == 0) nsec = f.bounces - 1;
  else if (f.render_mode == 2) nsec = 1;
  for (int s = 0; s < spp; s++) {
    // counters: per stage k a head at [4k] and an output count at [4k+1]
    hipError_t e = hipMemsetAsync(b.counters, 0, 4096, stream);
    if (e != hipSuccess) return (int)e;
    if (nsec + 1 > 250) return (int)hipErrorInvalidValue;
    for (int k = 0; k <= nsec; k++) {
      StageArgs a;
      a.pool = pool; a.f = f; a.color = color; a.depth = depth; a.hits = hits;
      a.state = b.state; a.facc = b.facc; a.npix = b.npix;
      a.queue_in = k > 0 ? b.queue[(k - 1) & 1] : nullptr;
      a.count_in = k > 0 ? b.counters + 4 * (k - 1) + 1 : nullptr;
      a.queue_out = b.queue[k & 1];
      a.count_out = b.counters + 4 * k + 1;
      a.head = b.counters + 4 * k;
      a.segment = k;
      a.sample = s;
      const int work_blocks = k == 0 ? f.ntiles : b.blocks;
      const int blocks = work_blocks < b.blocks ? work_blocks : b.blocks;
      if (k == 0) hipLaunchKernelGGL(stage_kernel<true>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
      else hipLaunchKernelGGL(stage_kernel<false>, dim3((unsigned)blocks), dim3(64), 0, stream, a);
      e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
    }
  }
  if (spp > 1) {
    dim3 grid((unsigned)((f.width + 255) / 256), (unsigned)(f.tiles_y * 8));
    hipLaunchKernelGGL(resolve_kernel, grid, dim3(256), 0, stream, f, b.facc, b.npix, color);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // namespace svo
