// svo_beam.hip.h -- the beam pre-pass (useBeamOptimization): a conservative start distance per 4x4 pixel block.
//
// Reference: Main.java:257-266 dispatches svobeam.comp (:617-637) at 1/4 resolution before the trace pass, which
// reads beamDist = beam[px / 4] (svotrace.comp:656-658) and moves the ray origin by it (:438).  That pass is
// dormant and not usable as written: one full-depth ray through the corner pixel of each block, cast with an
// un-normalised direction, so its t is neither conservative for the other 15 pixels nor in the units of the trace
// pass's normalised rays; the beam image only exists when the flag is set before start-up.
//
// What is built instead is the pass that flag asks for, made exact:
//   * the coarse pass walks the OCTREE against the block's ray pyramid (the pixel footprints widened by half a
//     pixel, four side planes through the camera): depth-first over non-empty child cubes that are not entirely
//     outside a side plane, pruned by distance; a cube ends the descent when it is a leaf, at MAX_DEPTH, or no
//     larger than the pyramid is wide at its distance.  The block's value = the smallest camera-to-cube distance,
//     times 1 - 2^-10: no ray of the block can meet a non-empty voxel before it;
//   * the trace pass does NOT move the origin: the primary ray keeps its coefficients and only starts its walk at
//     t_min = max(t_min, beam).  Every t the walk reports is derived from cell corners, so hit pointer, value,
//     normal, level, t, colour and depth keep the bits they have without the pre-pass; only the iteration count
//     drops (and renderMode 1, which displays it, changes).
// The CPU statement of the same pass is oracle/svo_oracle.c::svo_oracle_beam (the checker; bit-equal floats).
#pragma once
#include "svo_device.h"
#include "svo_kernels.h"
#include "svo_trav.h"

namespace svo {

constexpr int kBeamBlock = 4;   // Main.java:41 beamSquareSize

// The top kBeamTop levels of a pool may carry no emptiness information: the reference's world builder flags every node
// of its all-interior levels (fillEmptyChildren, the chunk nodes, the 512^3 task heads; Octree.java:317-343, 481-502) with
// value 1 whether anything lies below or not, and the coarse walk would have to open each of them (80 % of its visits
// on the 8192^3 scene).  So, once per pool, every node of those levels is marked by its octant path:
//   H(node) = value != 0 and (the cast treats it as a leaf  or  H of one of its children),
// children kBeamTop + 1 levels down counting by their value alone; the walk skips cubes of those levels with H = 0 (a
// cast cannot end inside them either: it only descends through non-empty nodes).  4 680 bytes.
constexpr int kBeamTop = 4;
constexpr int kBeamLiveBytes = 4680;
__host__ __device__ inline int beam_live_off(int depth) { return depth == 1 ? 0 : depth == 2 ? 8 : depth == 3 ? 72 : 584; }

// level `depth` (kBeamTop first, then upwards): one thread per octant path
__global__ void beam_live_kernel(const uint8_t *pool_base, uint32_t pool_len, int depth, uint8_t *live) {
  const uint32_t path = blockIdx.x * blockDim.x + threadIdx.x;
  if (path >= (1u << (3 * depth))) return;
  Pool pool;
  pool.base = pool_base;
  pool.len = pool_len;
  uint64_t rec = load_record(pool, 0u);
  uint32_t base = rec_cp(rec), mask = rec_mask_be(rec), ptr = 0, tag = 0;   // the walk starts at the root's children
  bool reach = true;
  for (int l = 1; l <= depth && reach; l++) {
    const uint32_t n = (path >> (3 * (depth - l))) & 7u;
    ptr = base + child_offset(mask, n);
    tag = (mask >> (2u * n)) & 3u;
    rec = load_record(pool, ptr);
    if (l < depth) {   // an ancestor: the cast only goes on through a non-empty interior node with a child block
      reach = rec_value(rec) != 0u && tag == 0u && rec_cp(rec) != 0u;
      base = ptr + rec_cp(rec);
      mask = rec_mask_be(rec);
    }
  }
  uint32_t h = 0u;
  if (reach && rec_value(rec) != 0u) {
    if (tag != 0u || rec_cp(rec) == 0u) {
      h = 1u;
    } else if (depth == kBeamTop) {
      const uint32_t cb = ptr + rec_cp(rec), cm = rec_mask_be(rec);
      for (uint32_t n = 0; n < 8u; n++) h |= rec_value(load_record(pool, cb + child_offset(cm, n))) != 0u ? 1u : 0u;
    } else {
      for (uint32_t n = 0; n < 8u; n++) h |= live[beam_live_off(depth + 1) + path * 8u + n];
    }
  }
  live[beam_live_off(depth) + path] = (uint8_t)h;
}

struct BeamArgs {
  const uint8_t *pool;
  const uint8_t *live;   // H of the top levels, by octant path (beam_live_kernel)
  Frame f;
  float *beam;    // [beam_h][beam_w], whole-frame indexing; only the block rows of this launch's tile rows are written
  int beam_w, beam_h;
  int cam_ok;     // the camera is a finite planar rectangle (host check); otherwise every block gets 0
};

__device__ __forceinline__ V3 beam_dir(const Frame &f, float u, float v) {
  const float *c = f.cam;
  const V3 a = mk(mix_g(c[3], c[6], v), mix_g(c[4], c[7], v), mix_g(c[5], c[8], v));
  const V3 b = mk(mix_g(c[9], c[12], v), mix_g(c[10], c[13], v), mix_g(c[11], c[14], v));
  return mk(mix_g(a.x, b.x, u), mix_g(a.y, b.y, u), mix_g(a.z, b.z, u));
}

// Eight lanes per block, one per child octant: a wave walks 8 neighbouring blocks.  A step of a block's walk takes the
// nearest pending cube off the block's stack (LDS) and looks at its 8 children at once -- each lane fetches one child
// record, tests it against the pyramid's side planes and measures its distance; the lanes then agree (cross-lane
// min / ranks inside the group of 8) on the new bound and push the children that have to be opened, farthest
// first, so that the nearest is opened next.  One dependent memory round trip per opened cube instead of eight.
constexpr int kBeamStack = 96;   // <= 12 levels x 7 pending siblings + 8

// min over the 8 lanes of a group with three DPP moves (no LDS crossbar): neighbours in the quad, the other pair of the
// quad, then the mirrored lane of the half row (which lies in the other quad)
__device__ __forceinline__ float group8_min(float v) {
  int i = __float_as_int(v);
  v = fmin_g(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
  i = __float_as_int(v);
  v = fmin_g(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
  i = __float_as_int(v);
  return fmin_g(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0x141, 0xf, 0xf, false))); // row_half_mirror
}

__global__ __launch_bounds__(64) void beam_kernel(const BeamArgs a) {
  __shared__ uint32_t stack[3][8][kBeamStack];   // three planes of dwords: 9 KB per wave
  const Frame &f = a.f;
  const uint32_t lane = threadIdx.x, grp = lane >> 3, ch = lane & 7u;
  const int bx = (int)(blockIdx.x * 8u + grp);
  // blockIdx.y enumerates the block rows under this launch's tile rows (two per 8-pixel tile row)
  const int ty = (int)blockIdx.y >> 1;
  const int by = (frame_gy(f, ty, 0) >> 2) + ((int)blockIdx.y & 1);
  const bool live = bx < a.beam_w && by < a.beam_h;
  const BufPool pool = make_bufpool(a.pool, f.pool_len);   // hardware range check: reads past the pool give 0
  const float W = (float)f.width, H = (float)f.height;
  const float u0 = ((float)(bx * kBeamBlock) - 0.5f) / W, u1 = ((float)(bx * kBeamBlock + kBeamBlock) + 0.5f) / W;
  const float v0 = ((float)(by * kBeamBlock) - 0.5f) / H, v1 = ((float)(by * kBeamBlock + kBeamBlock) + 0.5f) / H;
  const V3 d00 = beam_dir(f, u0, v0), d10 = beam_dir(f, u1, v0), d01 = beam_dir(f, u0, v1), d11 = beam_dir(f, u1, v1);
  const V3 dc = beam_dir(f, 0.5f * (u0 + u1), 0.5f * (v0 + v1));
  V3 n[4] = {cross3(d00, d01), cross3(d11, d10), cross3(d10, d00), cross3(d01, d11)};
  bool planes_ok = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float s = dot3(n[k], dc);
    if (s < 0.0f) n[k] = mk(-n[k].x, -n[k].y, -n[k].z);
    else if (!(s > 0.0f)) planes_ok = false;
  }
  float an[4];
#pragma unroll
  for (int k = 0; k < 4; k++) an[k] = __builtin_fabsf(n[k].x) + (__builtin_fabsf(n[k].y) + __builtin_fabsf(n[k].z));
  const V3 eu = mk(d10.x - d00.x, d10.y - d00.y, d10.z - d00.z), ev = mk(d01.x - d00.x, d01.y - d00.y, d01.z - d00.z);
  const float spread2 = fmax_g(dot3(eu, eu), dot3(ev, ev)) / dot3(dc, dc);
  const V3 o = mk(f.cam[0], f.cam[1], f.cam[2]);
  const bool walk = live && a.cam_ok && planes_ok;

  float best2 = __builtin_inff();
  // a pending cube, 3 dwords = {offset of its first child record,
  //                            its child tags | origin x << 16,   origin y | origin z << 14 | depth of its children << 28};
  // the origin in units of 2^-13 (exact: no cube is smaller), 14 bits per axis
  int sp = 0;
  if (walk) {
    const uint64_t root = load_record(pool, 0u);
    if (ch == 0u) { stack[0][grp][0] = rec_cp(root); stack[1][grp][0] = rec_mask_be(root); stack[2][grp][0] = 1u << 28; }
    sp = 1;
  }
  while (__ballot(sp > 0) != 0ull) {
    if (sp > 0) {
      sp--;
      const uint32_t e0 = stack[0][grp][sp], e1 = stack[1][grp][sp], e2 = stack[2][grp][sp];
      const uint32_t mask = e1 & 0xffffu, depth = e2 >> 28;                  // depth of the children, 1..13
      const float size = __uint_as_float((127u - depth) << 23);             // their edge
      // parent origin from integer coordinates (units of 2^-13, exact)
      const float x = 1.0f + (float)(e1 >> 16) * 0.0001220703125f, y = 1.0f + (float)(e2 & 0x3fffu) * 0.0001220703125f,
                  z = 1.0f + (float)((e2 >> 14) & 0x3fffu) * 0.0001220703125f;
      {   // the bound may have passed this cube since it was pushed: its own distance again (edge 2 * size)
        const float ps = size * 2.0f;
        const float px = fmax_g(fmax_g(x - o.x, o.x - (x + ps)), 0.0f), py = fmax_g(fmax_g(y - o.y, o.y - (y + ps)), 0.0f),
                    pz = fmax_g(fmax_g(z - o.z, o.z - (z + ps)), 0.0f);
        if (!(px * px + (py * py + pz * pz) < best2)) continue;
      }
      const uint32_t ptr = e0 + child_offset(mask, ch);
      const uint32_t tag = (mask >> (2u * ch)) & 3u;
      const uint64_t rec = load_record(pool, ptr);
      const float lx = x + (float)(ch & 1u) * size, ly = y + (float)((ch >> 1) & 1u) * size, lz = z + (float)((ch >> 2) & 1u) * size;
      const float half = 0.5f * size;
      const V3 m = mk((lx + half) - o.x, (ly + half) - o.y, (lz + half) - o.z);   // cube centre, camera-relative
      bool cand = rec_value(rec) != 0u;                                             // not empty
      if (depth <= (uint32_t)kBeamTop) {   // top levels: a cast could end in it or below it (octant path from the origin)
        const uint32_t sh = 13u - depth;
        const uint32_t xi = (uint32_t)((lx - 1.0f) * 8192.0f) >> sh, yi = (uint32_t)((ly - 1.0f) * 8192.0f) >> sh,
                       zi = (uint32_t)((lz - 1.0f) * 8192.0f) >> sh;
        uint32_t path = 0u;
        for (int b = (int)depth - 1; b >= 0; b--) path = path * 8u + (((xi >> b) & 1u) | (((yi >> b) & 1u) << 1) | (((zi >> b) & 1u) << 2));
        cand = a.live[beam_live_off((int)depth) + path] != 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) cand = cand && !(dot3(n[k], m) + an[k] * half < 0.0f);   // not entirely behind a side plane
      const float ddx = fmax_g(fmax_g(lx - o.x, o.x - (lx + size)), 0.0f), ddy = fmax_g(fmax_g(ly - o.y, o.y - (ly + size)), 0.0f),
                  ddz = fmax_g(fmax_g(lz - o.z, o.z - (lz + size)), 0.0f);
      const float dist2 = ddx * ddx + (ddy * ddy + ddz * ddz);
      cand = cand && dist2 < best2;
      const uint32_t cp = tag == 0u ? rec_cp(rec) : 0u;
      const bool terminal = tag != 0u || cp == 0u || depth >= (uint32_t)kMaxDepth || size * size <= dist2 * spread2;
      best2 = fmin_g(best2, group8_min(cand && terminal ? dist2 : __builtin_inff()));
      const bool open = cand && !terminal && dist2 < best2;
      // push the cubes to open so that the octant nearest the camera is opened first: octants are ordered by
      // k = child ^ near (near = the octant of the parent cube that holds / faces the camera; k = 0 nearest, 7 farthest);
      // the group's open set comes from one ballot, its bits are permuted into k order, and a lane's stack slot is
      // the number of open octants with a larger k
      const uint32_t near = (o.x >= x + size ? 1u : 0u) | (o.y >= y + size ? 2u : 0u) | (o.z >= z + size ? 4u : 0u);
      uint32_t om = (uint32_t)(__ballot(open) >> (lane & ~7u)) & 0xffu;
      const uint32_t nopen = (uint32_t)__builtin_popcount(om);
      if (near & 1u) om = ((om & 0x55u) << 1) | ((om & 0xaau) >> 1);
      if (near & 2u) om = ((om & 0x33u) << 2) | ((om & 0xccu) >> 2);
      if (near & 4u) om = ((om & 0x0fu) << 4) | ((om & 0xf0u) >> 4);
      const uint32_t farther = (uint32_t)__builtin_popcount(om >> ((ch ^ near) + 1u));
      if (open) {
        const uint32_t ix = (uint32_t)((lx - 1.0f) * 8192.0f), iy = (uint32_t)((ly - 1.0f) * 8192.0f), iz = (uint32_t)((lz - 1.0f) * 8192.0f);
        const int at = sp + (int)farther;
        stack[0][grp][at] = ptr + cp;
        stack[1][grp][at] = rec_mask_be(rec) | (ix << 16);
        stack[2][grp][at] = iy | (iz << 14) | ((depth + 1u) << 28);
      }
      sp += (int)nopen;
    }
    __builtin_amdgcn_wave_barrier();   // the groups of the wave run in lockstep; keep the LDS traffic in program order
  }
  if (live && ch == 0u) a.beam[(size_t)by * (size_t)a.beam_w + (size_t)bx] = walk ? __builtin_sqrtf(best2) * 0.9990234375f : 0.0f;
}

// host-side check that the four corner rays span a finite planar rectangle (r2 = r1 + l2 - l1 up to rounding)
inline int beam_camera_ok(const float *c) {
  for (int i = 0; i < 15; i++)
    if (!(__builtin_fabsf(c[i]) < 1.0e30f)) return 0;
  const float ex = c[12] - (c[9] + (c[6] - c[3])), ey = c[13] - (c[10] + (c[7] - c[4])), ez = c[14] - (c[11] + (c[8] - c[5]));
  const float dx = c[12] - c[3], dy = c[13] - c[4], dz = c[14] - c[5];
  const float e2 = ex * ex + (ey * ey + ez * ez), d2 = dx * dx + (dy * dy + dz * dz);
  return d2 > 0.0f && e2 <= 1.0e-8f * d2;
}

}  // namespace svo
