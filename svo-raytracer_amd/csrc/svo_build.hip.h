// svo_build.hip.h -- heightmap -> SVO pool builder on the GPU.
//
// What the reference does (Octree.constructCompleteOctree, Octree.java:192-353): per 1024^3 chunk a compute
// shader fills a dense voxel array from a height map and a material map (chunkgen-heightmap.comp:16-28), the array
// is read back, eight Java threads run the recursive constructInnerOctree over 512^3 sub-cubes (:511-670,
// OctreeThread.java:20-23), and the eight sub-pools are spliced behind the chunk node (:317-343).
//
// Here the pool is produced on the GPU straight from the two maps, level by level, without a voxel grid (8192^3
// dense would be 512 GiB) and without recursion:
//   1. min / max pyramid of the heights (cells of 8, 16, ... N columns);
//   2. top-down, one level at a time: a thread per (node, child octant) classifies the child cube as empty / solid /
//      mixed -- from the pyramid when the cube is 8 voxels or more on a side, by enumeration below that -- and
//      decides its record type by the reference's rules (26-neighbour normal for unit voxels, the 27 corner samples
//      for bigger solid cubes, neighbours outside the 1024^3 chunk ignored).  The 8 lanes of a node combine tags,
//      block size and the set of children that get a block of their own with cross-lane reductions; an exclusive
//      scan over the level places those children -- in order, so every level stays sorted in the reference's
//      depth-first order;
//   3. bottom-up: bytes of every subtree; top-down: byte offset of every sibling block (a block is followed by the
//      subtrees of its children in order -- the recursion's emission order as a prefix sum);
//   4. a thread per record writes its 1 / 3 / 7 bytes, child pointers relative and big-endian (Octree.java:162-172).
// The bytes equal those of the restated constructInnerOctree over the dense grid (tests/test_gpu_builder.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace svo {
namespace build {

constexpr int kChunk = 1024;   // Octree.java:39
constexpr int kTask = 512;     // OctreeThread.java:22
enum : uint32_t { T_INTERIOR = 0, T_SURFACE = 1, T_SUBDIV = 2, T_NONSURFACE = 3 };   // Octree.java:589-599

// voxel source 1: a height map and a material map (the reference's world generator, chunkgen-heightmap.comp)
struct Maps {
  const uint16_t *h;     // N*N heights, index z*N + x: the column is solid for y <= h
  const uint8_t *mat;    // N*N surface materials (the top five layers of a column)
  const uint16_t *pmin[12];   // pyramid level l: cells of 8 << l columns
  const uint16_t *pmax[12];
  const int16_t *pdeep[12];   // min over the cell of the deepest y up to which a column is uniformly value 1:
                              // h when its surface material is 1 as well, else h - 5 (below the material band)
  int n, chunk;
  __device__ __forceinline__ bool empty_at(int x, int y, int z) const { return y > (int)h[(size_t)z * n + x]; }
};

// voxel source 2: a dense chunk of voxels, the input of Octree.constructInnerOctree itself (index x | y << shift |
// z << 2 shift, Octree.java:110-112), with a summary per cube of 2, 4, ... n voxels: whether all its voxels are equal,
// and its first non-empty voxel in the reference's z, y, x scan order (position key z << 20 | y << 10 | x, and value)
struct Grid {
  const uint8_t *vox;
  const uint32_t *fpos[11];   // level k: cubes of 2^k; first non-empty voxel's key, 0xffffffff if the cube is empty
  const uint16_t *fval[11];   // its value | 0x100 if the cube is homogeneous
  int n, shift, chunk;
  __device__ __forceinline__ uint32_t at(int x, int y, int z) const {
    return vox[(size_t)x | ((size_t)y << shift) | ((size_t)z << (2 * shift))];
  }
  __device__ __forceinline__ bool empty_at(int x, int y, int z) const { return at(x, y, z) == 0u; }
};

// one level of the tree: the nodes that own a block of 8 children, in depth-first order
struct Level {
  uint64_t *pos = nullptr;      // x | y << 16 | z << 32 of the node's cube
  uint16_t *mask = nullptr;     // child tags (the node's leafMask)
  uint8_t *vals = nullptr;      // 8 child values
  uint16_t *normals = nullptr;  // 8 packed normals (only the level whose children are unit voxels)
  uint8_t *blk = nullptr;       // bytes of the children block
  uint8_t *expmask = nullptr;   // children that own a block themselves
  uint32_t *first = nullptr;    // index of the first such child in the next level
  uint32_t *sub = nullptr;      // bytes of the subtree below the node: its block + its children's subtrees
  uint32_t *start = nullptr;    // byte offset of the block in the pool
  uint32_t count = 0;
  int size = 0;                 // edge of the nodes' cubes
};

__device__ __forceinline__ uint32_t voxel_at(const Maps &m, int x, int y, int z) {   // chunkgen-heightmap.comp:16-28
  const int h = m.h[(size_t)z * m.n + x];
  if (y > h) return 0u;
  if (h - y <= 4) return m.mat[(size_t)z * m.n + x];
  return 1u;
}
template <class Src>
__device__ __forceinline__ bool usable(const Src &m, int g, int c) {   // inside the world and inside c's chunk
  const int o = (c / m.chunk) * m.chunk;
  return g >= 0 && g < m.n && g >= o && g < o + m.chunk;
}

// kind: 0 empty, 1 homogeneous solid, 2 mixed; value as Octree.java:528-555 leaves it
__device__ __forceinline__ int classify(const Maps &m, int cx, int cy, int cz, int cs, uint32_t &value) {
  if (cs >= 8) {
    const int l = 31 - __builtin_clz((unsigned)cs) - 3;
    const int w = m.n / cs;
    const size_t pi = (size_t)(cz / cs) * w + (cx / cs);
    const int mx = m.pmax[l][pi];
    if (mx < cy) { value = 0u; return 0; }
    // a cube of 8 or more is taller than the 5-layer material band, so it can only be homogeneous in value 1:
    // every column solid up to the cube's top, and no band voxel of another material inside
    if (cy + cs - 1 <= (int)m.pdeep[l][pi]) { value = 1u; return 1; }
    // mixed: the value is the first non-empty sample in z, y, x order.  Solid voxels form a prefix of their column,
    // so that sample lies on the plane y = cy, in the first row that has a column reaching cy
    const uint32_t v0 = voxel_at(m, cx, cy, cz);
    if (v0) { value = v0; return 2; }
    const int w0 = m.n >> 3;
    for (int z = cz; z < cz + cs; z++) {
      const uint16_t *row = m.h + (size_t)z * m.n;
      const uint16_t *cellmax = m.pmax[0] + (size_t)(z >> 3) * w0;
      for (int x8 = cx; x8 < cx + cs; x8 += 8) {
        if ((int)cellmax[x8 >> 3] < cy) continue;   // no column of this 8 x 8 cell reaches cy
        for (int x = x8; x < x8 + 8; x++)
          if ((int)row[x] >= cy) { value = voxel_at(m, x, cy, z); return 2; }
      }
    }
    value = 0u;
    return 0;
  }
  uint32_t first = voxel_at(m, cx, cy, cz), val = first;
  if (cs == 1) { value = first; return first ? 1 : 0; }
  for (int z = cz; z < cz + cs; z++)
    for (int y = cy; y < cy + cs; y++)
      for (int x = cx; x < cx + cs; x++) {
        const uint32_t smp = voxel_at(m, x, y, z);
        if (smp) val = smp;
        if (smp != first) {
          if (first == 0u) first = smp;
          value = first;
          return 2;
        }
      }
  value = val;
  return val ? 1 : 0;
}

__device__ __forceinline__ int classify(const Grid &m, int cx, int cy, int cz, int cs, uint32_t &value) {
  if (cs == 1) { value = m.at(cx, cy, cz); return value ? 1 : 0; }
  const int k = 31 - __builtin_clz((unsigned)cs);
  const size_t w = (size_t)(m.n >> k);
  const uint32_t fv = m.fval[k][(((size_t)(cz >> k)) * w + (size_t)(cy >> k)) * w + (size_t)(cx >> k)];
  value = fv & 0xffu;
  if (fv & 0x100u) return value ? 1 : 0;
  return 2;
}

// genSurfaceNormal (Octree.java:620-649): offsets to the empty usable 26-neighbours, summed, halved toward zero
template <class Src>
__device__ __forceinline__ bool surface_normal(const Src &m, int cx, int cy, int cz, uint32_t &packed) {
  bool exposed = false;
  int nx = 0, ny = 0, nz = 0;
  for (int i = cx - 1; i <= cx + 1; i++) {
    if (!usable(m, i, cx)) continue;
    for (int k = cz - 1; k <= cz + 1; k++) {
      if (!usable(m, k, cz)) continue;
      for (int j = cy - 1; j <= cy + 1; j++) {
        if (!usable(m, j, cy)) continue;
        if (m.empty_at(i, j, k)) { exposed = true; nx += i - cx; ny += j - cy; nz += k - cz; }
      }
    }
  }
  packed = (uint32_t)((nx / 2 + 5) + (ny / 2 + 5) * 10 + (nz / 2 + 5) * 100);
  return exposed;
}

// checkBigNodeExposed (Octree.java:651-670): only coordinates {c - 1, c + cs, c + cs + 1} are looked at on every axis
template <class Src>
__device__ __forceinline__ bool big_node_exposed(const Src &m, int cx, int cy, int cz, int cs) {
  const int dx[3] = {-1, cs, cs + 1};
  for (int a = 0; a < 3; a++) {
    const int x = cx + dx[a];
    if (!usable(m, x, cx)) continue;
    for (int c = 0; c < 3; c++) {
      const int z = cz + dx[c];
      if (!usable(m, z, cz)) continue;
      for (int b = 0; b < 3; b++) {
        const int y = cy + dx[b];
        if (!usable(m, y, cy)) continue;
        if (m.empty_at(x, y, z)) return true;
      }
    }
  }
  return false;
}

__device__ __forceinline__ uint32_t tag_bytes(uint32_t tag) { return tag == 1u ? 3u : (tag == 3u ? 1u : 7u); }

// ---- pyramid ----------------------------------------------------------------------------------------------
__global__ void pyramid_base_kernel(const uint16_t *h, const uint8_t *mat, int n, uint16_t *pmin, uint16_t *pmax,
                                    int16_t *pdeep) {
  const int w = n >> 3;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= w * w) return;
  const int cx = c % w, cz = c / w;
  int mn = 65535, mx = 0, dp = 32767;
  for (int z = cz * 8; z < cz * 8 + 8; z++) {
    const uint4 v = *(const uint4 *)(h + (size_t)z * n + cx * 8);    // 8 heights
    const uint2 mm = *(const uint2 *)(mat + (size_t)z * n + cx * 8); // 8 materials
    const uint32_t q[4] = {v.x, v.y, v.z, v.w};
    const uint32_t mq[2] = {mm.x, mm.y};
    for (int i = 0; i < 8; i++) {
      const int a = (int)((q[i >> 1] >> (16 * (i & 1))) & 0xffffu);
      const int mt = (int)((mq[i >> 2] >> (8 * (i & 3))) & 0xffu);
      mn = min(mn, a);
      mx = max(mx, a);
      dp = min(dp, mt == 1 ? a : a - 5);
    }
  }
  pmin[c] = (uint16_t)mn;
  pmax[c] = (uint16_t)mx;
  pdeep[c] = (int16_t)dp;
}
__global__ void pyramid_up_kernel(const uint16_t *imin, const uint16_t *imax, const int16_t *ideep, int w, uint16_t *pmin,
                                  uint16_t *pmax, int16_t *pdeep) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= w * w) return;
  const int cx = c % w, cz = c / w, w2 = w * 2;
  int mn = 65535, mx = 0, dp = 32767;
  for (int dz = 0; dz < 2; dz++)
    for (int dxx = 0; dxx < 2; dxx++) {
      const size_t i = (size_t)(cz * 2 + dz) * w2 + (cx * 2 + dxx);
      mn = min(mn, (int)imin[i]);
      mx = max(mx, (int)imax[i]);
      dp = min(dp, (int)ideep[i]);
    }
  pmin[c] = (uint16_t)mn;
  pmax[c] = (uint16_t)mx;
  pdeep[c] = (int16_t)dp;
}

// ---- one level: classify the 8 children of every node ---------------------------------------------------------
// thread = (node, child octant); the 8 lanes of a node sit in one wave
template <class Src, bool kForced>
__global__ __launch_bounds__(256) void classify_kernel(const Src m, const Level lv) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t i = t >> 3, n = t & 7u;
  const bool live = i < lv.count;
  uint32_t type = T_SUBDIV, val = 0u, nrm = 0u;
  bool expand = false;
  if (live) {
    const uint64_t p = lv.pos[i];
    const int cs = lv.size >> 1;
    const int cx = (int)(p & 0xffffu) + (int)(n & 1u) * cs, cy = (int)((p >> 16) & 0xffffu) + (int)((n >> 1) & 1u) * cs,
              cz = (int)((p >> 32) & 0xffffu) + (int)((n >> 2) & 1u) * cs;
    if (kForced) {   // the eight 512^3 sub-cubes of a chunk: interior nodes of value 1 whatever they hold (Octree.java:317-343)
      type = T_INTERIOR; val = 1u; expand = true;
    } else {
      const int kind = classify(m, cx, cy, cz, cs, val);
      if (kind == 1) {
        if (cs == 1) type = surface_normal(m, cx, cy, cz, nrm) ? T_SURFACE : T_NONSURFACE;
        else type = big_node_exposed(m, cx, cy, cz, cs) ? T_INTERIOR : T_SUBDIV;
      } else if (kind == 0) {
        type = cs == 1 ? T_NONSURFACE : T_SUBDIV;
      } else {
        type = T_INTERIOR;
      }
      expand = type == T_INTERIOR && val != 0u && cs >= 2;
    }
  }
  // combine the node's 8 lanes
  uint32_t mk = live ? (type << (2u * n)) : 0u, bytes = live ? tag_bytes(type) : 0u;
  for (int off = 1; off < 8; off <<= 1) {
    mk |= (uint32_t)__shfl_xor((int)mk, off);
    bytes += (uint32_t)__shfl_xor((int)bytes, off);
  }
  const unsigned long long ex = __ballot(expand);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t em = (uint32_t)(ex >> (lane & ~7u)) & 0xffu;
  if (live) {
    lv.vals[(size_t)i * 8 + n] = (uint8_t)val;
    if (lv.normals) lv.normals[(size_t)i * 8 + n] = (uint16_t)nrm;
    if (n == 0u) {
      lv.mask[i] = (uint16_t)mk;
      lv.blk[i] = (uint8_t)bytes;
      lv.expmask[i] = (uint8_t)em;
    }
  }
}

// children that own a block become the nodes of the next level, in order
__global__ __launch_bounds__(256) void expand_kernel(const Level lv, uint64_t *next_pos) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t i = t >> 3, n = t & 7u;
  if (i >= lv.count) return;
  const uint32_t em = lv.expmask[i];
  if (!((em >> n) & 1u)) return;
  const uint64_t p = lv.pos[i];
  const uint64_t cs = (uint64_t)(lv.size >> 1);
  const uint64_t x = (p & 0xffffu) + (n & 1u) * cs, y = ((p >> 16) & 0xffffu) + ((n >> 1) & 1u) * cs,
                 z = ((p >> 32) & 0xffffu) + ((n >> 2) & 1u) * cs;
  next_pos[lv.first[i] + (uint32_t)__builtin_popcount(em & ((1u << n) - 1u))] = x | (y << 16) | (z << 32);
}

// ---- exclusive scan of per-node counts (popcount of expmask) ---------------------------------------------------
constexpr int kScanBlock = 256, kScanItems = 4, kScanTile = kScanBlock * kScanItems;

template <typename Tin, bool kPopcount>
__global__ __launch_bounds__(kScanBlock) void scan_tile_kernel(const Tin *in, uint32_t *out, uint32_t *tile_sums, uint32_t n) {
  __shared__ uint32_t part[kScanBlock];
  const uint32_t base = blockIdx.x * (uint32_t)kScanTile + threadIdx.x * (uint32_t)kScanItems;
  uint32_t v[kScanItems], sum = 0;
  for (int k = 0; k < kScanItems; k++) {
    uint32_t x = base + k < n ? (uint32_t)in[base + k] : 0u;
    if (kPopcount) x = (uint32_t)__builtin_popcount(x);
    v[k] = sum;
    sum += x;
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const uint32_t add = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  const uint32_t before = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int k = 0; k < kScanItems; k++)
    if (base + k < n) out[base + k] = before + v[k];
  if (threadIdx.x == kScanBlock - 1) tile_sums[blockIdx.x] = part[threadIdx.x];
}
__global__ void scan_add_kernel(uint32_t *out, const uint32_t *tile_offsets, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += tile_offsets[i / (uint32_t)kScanTile];
}

// ---- subtree bytes (bottom-up) and block offsets (top-down) ------------------------------------------------------
__global__ void subtree_kernel(const Level lv, const uint32_t *next_sub) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lv.count) return;
  // saturating: a dense chunk's subtree sums pass 2^32 (a 60 %-filled 1024^3 noise chunk is ~4.5 GB of records), and a
  // wrapped sum would look like a small pool to the size check -- 0xffffffff says "too large" whatever is added to it
  uint64_t s = lv.blk[i];
  if (next_sub) {
    const uint32_t k = (uint32_t)__builtin_popcount(lv.expmask[i]), f = lv.first[i];
    for (uint32_t j = 0; j < k; j++) s += next_sub[f + j];
  }
  lv.sub[i] = s > 0xffffffffull ? 0xffffffffu : (uint32_t)s;
}
__global__ void place_kernel(const Level lv, const uint32_t *next_sub, uint32_t *next_start) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lv.count) return;
  const uint32_t k = (uint32_t)__builtin_popcount(lv.expmask[i]), f = lv.first[i];
  uint32_t at = lv.start[i] + lv.blk[i];    // a block is followed by the subtrees of its children, in order
  for (uint32_t j = 0; j < k; j++) {
    next_start[f + j] = at;
    at += next_sub[f + j];
  }
}
// level 0: the chunks follow one another behind the pool's prefix
__global__ void place_top_kernel(const Level lv, uint32_t prefix_len) {
  if (blockIdx.x || threadIdx.x) return;
  uint32_t at = prefix_len;
  for (uint32_t i = 0; i < lv.count; i++) {
    lv.start[i] = at;
    at += lv.sub[i];
  }
}

// ---- emission: one thread per record -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_kernel(const Level lv, const uint16_t *next_mask, const uint32_t *next_start,
                                                   uint8_t *pool) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t i = t >> 3, n = t & 7u;
  if (i >= lv.count) return;
  const uint32_t mk = lv.mask[i];
  const uint32_t tag = (mk >> (2u * n)) & 3u;
  // byte offset of child n in the block: 7 n - 4 popcount(lo) - 2 popcount(lo & hi) over the children below it
  const uint32_t below = (1u << (2u * n)) - 1u, lo = mk & 0x5555u & below, both = lo & (mk >> 1);
  const uint32_t off = lv.start[i] + 7u * n - 4u * (uint32_t)__builtin_popcount(lo) - 2u * (uint32_t)__builtin_popcount(both);
  uint8_t *p = pool + off;
  const uint32_t val = lv.vals[(size_t)i * 8 + n];
  p[0] = (uint8_t)val;
  if (tag == T_NONSURFACE) return;
  if (tag == T_SURFACE) {   // packed normal, little-endian (Octree.java:150-151)
    const uint32_t nr = lv.normals ? lv.normals[(size_t)i * 8 + n] : 0u;
    p[1] = (uint8_t)nr;
    p[2] = (uint8_t)(nr >> 8);
    return;
  }
  uint32_t cp = 0u, cm = 0u;
  const uint32_t em = lv.expmask[i];
  if (tag == T_INTERIOR && ((em >> n) & 1u)) {
    const uint32_t j = lv.first[i] + (uint32_t)__builtin_popcount(em & ((1u << n) - 1u));
    cp = next_start[j] - off;     // relative to the node itself (Octree.java:162-164)
    cm = next_mask[j];
  }
  p[1] = (uint8_t)(cp >> 24); p[2] = (uint8_t)(cp >> 16); p[3] = (uint8_t)(cp >> 8); p[4] = (uint8_t)cp;
  p[5] = (uint8_t)(cm >> 8); p[6] = (uint8_t)cm;
}

// ---- host side --------------------------------------------------------------------------------------------------
struct Builder {
  std::vector<void *> allocs;
  hipStream_t stream = nullptr;
  hipError_t err = hipSuccess;

  template <typename T>
  T *alloc(size_t n) {
    void *p = nullptr;
    if (err == hipSuccess) err = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
    if (err != hipSuccess) return nullptr;
    allocs.push_back(p);
    return (T *)p;
  }
  void release() {
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
  }
  bool ok() {
    if (err == hipSuccess) err = hipGetLastError();
    return err == hipSuccess;
  }

  // out[i] = sum of count(in[j]) for j < i; returns the total through *total (device -> host)
  template <typename Tin, bool kPopcount>
  void exclusive_scan(const Tin *in, uint32_t *out, uint32_t n, uint32_t *total) {
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    uint32_t *sums = alloc<uint32_t>(tiles + 1);
    if (!sums) return;
    hipLaunchKernelGGL((scan_tile_kernel<Tin, kPopcount>), dim3(tiles), dim3(kScanBlock), 0, stream, in, out, sums, n);
    if (tiles > 1) {
      uint32_t *offs = alloc<uint32_t>(tiles);
      if (!offs) return;
      exclusive_scan<uint32_t, false>(sums, offs, tiles, total);
      hipLaunchKernelGGL(scan_add_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, offs, n);
    } else if (total) {
      if (err == hipSuccess) err = hipMemcpyAsync(total, sums, 4, hipMemcpyDeviceToHost, stream);
      if (err == hipSuccess) err = hipStreamSynchronize(stream);
    }
  }
};

// Octree.fillEmptyChildren (Octree.java:481-502): the all-interior value-1 levels between the root and the chunks,
// depth-first by blocks; records the byte offset of every chunk node in visiting order
inline void prefix_levels(std::vector<uint8_t> &pre, size_t parent, int levels, std::vector<uint32_t> &chunk_nodes) {
  if (levels == 0) { chunk_nodes.push_back((uint32_t)parent); return; }
  size_t ch[8];
  for (int i = 0; i < 8; i++) {
    ch[i] = pre.size();
    pre.push_back(1);
    pre.insert(pre.end(), 6, 0);
  }
  for (int i = 0; i < 8; i++) prefix_levels(pre, ch[i], levels - 1, chunk_nodes);
  const uint32_t rel = (uint32_t)(ch[0] - parent);
  pre[parent + 1] = (uint8_t)(rel >> 24); pre[parent + 2] = (uint8_t)(rel >> 16); pre[parent + 3] = (uint8_t)(rel >> 8);
  pre[parent + 4] = (uint8_t)rel;
}
inline void chunk_positions(std::vector<uint64_t> &pos, int levels, uint64_t x, uint64_t y, uint64_t z) {
  if (levels == 0) { pos.push_back(x | (y << 16) | (z << 32)); return; }
  const uint64_t cs = (uint64_t)kChunk << (levels - 1);
  for (uint64_t i = 0; i < 8; i++) chunk_positions(pos, levels - 1, x + (i & 1) * cs, y + ((i >> 1) & 1) * cs, z + ((i >> 2) & 1) * cs);
}

struct Result {
  uint8_t *pool = nullptr;    // device, zero-padded by `pad` bytes; caller owns it
  uint64_t len = 0, cap = 0;
  uint64_t nodes = 0;
  int levels = 0;
};

// d_h / d_mat: device copies of the maps.  Returns hipSuccess, or hipErrorInvalidValue for a bad size,
// hipErrorOutOfMemory / others from the runtime; *too_large is set when the pool would reach 2^31 bytes.
template <class Src>
inline hipError_t build_levels(Builder &B, const Src &m, int n, uint64_t pad, hipStream_t stream, Result &res, bool *too_large);

inline hipError_t build_pool(const uint16_t *d_h, const uint8_t *d_mat, int n, uint64_t pad, hipStream_t stream, Result &res,
                             bool *too_large) {
  *too_large = false;
  if (n < 8 || n > 8192 || (n & (n - 1))) return hipErrorInvalidValue;
  Builder B;
  B.stream = stream;
  Maps m;
  m.h = d_h; m.mat = d_mat; m.n = n; m.chunk = n < kChunk ? n : kChunk;
  for (int l = 0; l < 12; l++) { m.pmin[l] = nullptr; m.pmax[l] = nullptr; m.pdeep[l] = nullptr; }
  int nlev = 0;
  while ((8 << nlev) <= n) nlev++;
  for (int l = 0; l < nlev; l++) {
    const int w = n / (8 << l);
    uint16_t *mn = B.alloc<uint16_t>((size_t)w * w), *mx = B.alloc<uint16_t>((size_t)w * w);
    int16_t *dp = B.alloc<int16_t>((size_t)w * w);
    if (!mn || !mx || !dp) { B.release(); return B.err; }
    if (l == 0)
      hipLaunchKernelGGL(pyramid_base_kernel, dim3((w * w + 255) / 256), dim3(256), 0, stream, d_h, d_mat, n, mn, mx, dp);
    else
      hipLaunchKernelGGL(pyramid_up_kernel, dim3((w * w + 255) / 256), dim3(256), 0, stream, m.pmin[l - 1], m.pmax[l - 1],
                         m.pdeep[l - 1], w, mn, mx, dp);
    m.pmin[l] = mn; m.pmax[l] = mx; m.pdeep[l] = dp;
  }
  return build_levels(B, m, n, pad, stream, res, too_large);
}

// ---- voxel source 2: summaries of a dense chunk ------------------------------------------------------------------
// level 1 from the voxels: one thread per cube of 2
__global__ void grid_base_kernel(const uint8_t *vox, int n, int shift, uint32_t *fpos, uint16_t *fval) {
  const size_t w = (size_t)(n >> 1);
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= w * w * w) return;
  const int cx = (int)(c % w) * 2, cy = (int)((c / w) % w) * 2, cz = (int)(c / (w * w)) * 2;
  uint32_t first_key = 0xffffffffu, first_val = 0u;
  const uint32_t v0 = vox[(size_t)cx | ((size_t)cy << shift) | ((size_t)cz << (2 * shift))];
  bool same = true;
  for (int z = cz; z < cz + 2; z++)        // the reference's scan order: z, then y, then x (Octree.java:535-537)
    for (int y = cy; y < cy + 2; y++)
      for (int x = cx; x < cx + 2; x++) {
        const uint32_t v = vox[(size_t)x | ((size_t)y << shift) | ((size_t)z << (2 * shift))];
        same = same && v == v0;
        if (v != 0u && first_key == 0xffffffffu) { first_key = ((uint32_t)z << 20) | ((uint32_t)y << 10) | (uint32_t)x; first_val = v; }
      }
  fpos[c] = first_key;
  fval[c] = (uint16_t)(first_val | (same ? 0x100u : 0u));
}
// level k from level k - 1: a cube is homogeneous when its 8 parts are and agree; its first non-empty voxel is the one
// with the smallest (z, y, x) among its parts' first voxels
__global__ void grid_up_kernel(const uint32_t *ipos, const uint16_t *ival, int w, uint32_t *fpos, uint16_t *fval) {
  const size_t W = (size_t)w, c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= W * W * W) return;
  const size_t cx = c % W, cy = (c / W) % W, cz = c / (W * W), w2 = W * 2;
  uint32_t best = 0xffffffffu, bval = 0u, v0 = 0u;
  bool same = true;
  for (int i = 0; i < 8; i++) {
    const size_t j = ((cz * 2 + (size_t)((i >> 2) & 1)) * w2 + (cy * 2 + (size_t)((i >> 1) & 1))) * w2 + (cx * 2 + (size_t)(i & 1));
    const uint32_t pk = ipos[j], pv = ival[j];
    if (i == 0) v0 = pv;
    same = same && (pv & 0x100u) != 0u && pv == v0;
    if (pk < best) { best = pk; bval = pv & 0xffu; }
  }
  fpos[c] = best;
  fval[c] = (uint16_t)(bval | (same ? 0x100u : 0u));
}

// d_vox: device copy of an n^3 chunk (n a power of two, 2..1024), indexed x | y << log2 n | z << 2 log2 n.
inline hipError_t build_pool_from_voxels(const uint8_t *d_vox, int n, uint64_t pad, hipStream_t stream, Result &res, bool *too_large) {
  *too_large = false;
  if (n < 2 || n > kChunk || (n & (n - 1))) return hipErrorInvalidValue;
  Builder B;
  B.stream = stream;
  Grid g;
  g.vox = d_vox; g.n = n; g.chunk = n; g.shift = 0;
  while ((1 << g.shift) < n) g.shift++;
  for (int k = 0; k < 11; k++) { g.fpos[k] = nullptr; g.fval[k] = nullptr; }
  for (int k = 1; k <= g.shift; k++) {
    const size_t w = (size_t)(n >> k), cells = w * w * w;
    uint32_t *fp = B.alloc<uint32_t>(cells);
    uint16_t *fv = B.alloc<uint16_t>(cells);
    if (!fp || !fv) { B.release(); return B.err; }
    const unsigned grid = (unsigned)((cells + 255) / 256);
    if (k == 1) hipLaunchKernelGGL(grid_base_kernel, dim3(grid), dim3(256), 0, stream, d_vox, n, g.shift, fp, fv);
    else hipLaunchKernelGGL(grid_up_kernel, dim3(grid), dim3(256), 0, stream, g.fpos[k - 1], g.fval[k - 1], (int)w, fp, fv);
    g.fpos[k] = fp; g.fval[k] = fv;
  }
  return build_levels(B, g, n, pad, stream, res, too_large);
}

// Levels, offsets and emission for any voxel source (the source's summaries are already built; B owns the scratch).
template <class Src>
inline hipError_t build_levels(Builder &B, const Src &m, int n, uint64_t pad, hipStream_t stream, Result &res, bool *too_large) {
  // the pool's prefix and the first level: the chunk nodes (or the root itself for worlds of one 512^3 task)
  std::vector<uint8_t> pre = {1, 0, 0, 0, 0, 0, 0};   // createDummyHead / the root: interior, value 1
  std::vector<uint32_t> chunk_nodes;
  std::vector<uint64_t> top_pos;
  const bool chunked = n > kTask;
  int top_levels = 0;
  if (chunked) {
    while ((kChunk << top_levels) < n) top_levels++;
    prefix_levels(pre, 0, top_levels, chunk_nodes);
    chunk_positions(top_pos, top_levels, 0, 0, 0);
  } else {
    chunk_nodes.push_back(0);
    top_pos.push_back(0);
  }

  std::vector<Level> lv;
  {
    Level L;
    L.count = (uint32_t)top_pos.size();
    L.size = chunked ? kChunk : n;
    L.pos = B.alloc<uint64_t>(L.count);
    if (!L.pos) { B.release(); return B.err; }
    B.err = hipMemcpyAsync(L.pos, top_pos.data(), top_pos.size() * 8, hipMemcpyHostToDevice, stream);
    lv.push_back(L);
  }
  uint64_t nodes = 0;
  for (size_t d = 0; B.ok(); d++) {
    Level &L = lv[d];
    const size_t c = L.count;
    nodes += c;
    L.mask = B.alloc<uint16_t>(c); L.vals = B.alloc<uint8_t>(c * 8); L.blk = B.alloc<uint8_t>(c);
    L.expmask = B.alloc<uint8_t>(c); L.first = B.alloc<uint32_t>(c); L.sub = B.alloc<uint32_t>(c); L.start = B.alloc<uint32_t>(c);
    if (L.size == 2) L.normals = B.alloc<uint16_t>(c * 8);
    if (!B.ok()) break;
    const unsigned grid = (unsigned)((c * 8 + 255) / 256);
    if (chunked && d == 0) hipLaunchKernelGGL((classify_kernel<Src, true>), dim3(grid), dim3(256), 0, stream, m, L);
    else hipLaunchKernelGGL((classify_kernel<Src, false>), dim3(grid), dim3(256), 0, stream, m, L);
    uint32_t total = 0;
    B.exclusive_scan<uint8_t, true>(L.expmask, L.first, (uint32_t)c, &total);
    if (!B.ok() || total == 0 || L.size == 2) break;
    Level nx;
    nx.count = total;
    nx.size = L.size >> 1;
    nx.pos = B.alloc<uint64_t>(total);
    if (!nx.pos) break;
    hipLaunchKernelGGL(expand_kernel, dim3(grid), dim3(256), 0, stream, L, nx.pos);
    lv.push_back(nx);   // (invalidates L)
  }
  if (!B.ok()) { B.release(); return B.err; }

  // subtree bytes, bottom-up; then the pool size
  for (int d = (int)lv.size() - 1; d >= 0; d--) {
    const uint32_t *next_sub = d + 1 < (int)lv.size() ? lv[(size_t)d + 1].sub : nullptr;
    hipLaunchKernelGGL(subtree_kernel, dim3((lv[(size_t)d].count + 255) / 256), dim3(256), 0, stream, lv[(size_t)d], next_sub);
  }
  std::vector<uint32_t> top_sub(lv[0].count);
  if (B.ok()) B.err = hipMemcpyAsync(top_sub.data(), lv[0].sub, top_sub.size() * 4, hipMemcpyDeviceToHost, stream);
  if (B.ok()) B.err = hipStreamSynchronize(stream);
  if (!B.ok()) { B.release(); return B.err; }
  uint64_t total = pre.size();
  for (uint32_t s : top_sub) total += s;   // (64-bit: the saturated per-node sums cannot wrap here)
  if (total > 0x7fffffffull) {   // child pointers are signed 32-bit (Octree.java:162-168)
    B.release();
    res.len = total;
    *too_large = true;
    return hipSuccess;
  }
  // block offsets, top-down
  hipLaunchKernelGGL(place_top_kernel, dim3(1), dim3(1), 0, stream, lv[0], (uint32_t)pre.size());
  for (size_t d = 0; d + 1 < lv.size(); d++)
    hipLaunchKernelGGL(place_kernel, dim3((lv[d].count + 255) / 256), dim3(256), 0, stream, lv[d], lv[d + 1].sub, lv[d + 1].start);

  // the pool
  uint8_t *pool = nullptr;
  if (B.ok()) B.err = hipMalloc((void **)&pool, total + pad);
  if (!B.ok()) { B.release(); return B.err; }
  (void)hipMemsetAsync(pool + total, 0, pad, stream);
  for (size_t d = 0; d < lv.size(); d++) {
    const bool last = d + 1 == lv.size();
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)(((size_t)lv[d].count * 8 + 255) / 256)), dim3(256), 0, stream, lv[d],
                       last ? nullptr : lv[d + 1].mask, last ? nullptr : lv[d + 1].start, pool);
  }
  // prefix: chunk nodes point at their blocks (the root carries its own tag mask when it is the only node above the blocks)
  std::vector<uint32_t> top_start(lv[0].count);
  std::vector<uint16_t> top_mask(lv[0].count);
  if (B.ok()) B.err = hipMemcpyAsync(top_start.data(), lv[0].start, top_start.size() * 4, hipMemcpyDeviceToHost, stream);
  if (B.ok()) B.err = hipMemcpyAsync(top_mask.data(), lv[0].mask, top_mask.size() * 2, hipMemcpyDeviceToHost, stream);
  if (B.ok()) B.err = hipStreamSynchronize(stream);
  if (B.ok()) {
    for (size_t c = 0; c < chunk_nodes.size(); c++) {
      const uint32_t at = chunk_nodes[c], rel = top_start[c] - at;
      pre[at + 1] = (uint8_t)(rel >> 24); pre[at + 2] = (uint8_t)(rel >> 16); pre[at + 3] = (uint8_t)(rel >> 8); pre[at + 4] = (uint8_t)rel;
      if (!chunked) { pre[at + 5] = (uint8_t)(top_mask[c] >> 8); pre[at + 6] = (uint8_t)top_mask[c]; }
    }
    B.err = hipMemcpyAsync(pool, pre.data(), pre.size(), hipMemcpyHostToDevice, stream);
  }
  if (B.ok()) B.err = hipStreamSynchronize(stream);
  B.release();
  if (B.err != hipSuccess) {
    if (pool) (void)hipFree(pool);
    return B.err;
  }
  res.pool = pool; res.len = total; res.cap = total + pad; res.nodes = nodes; res.levels = (int)lv.size();
  return hipSuccess;
}

}  // namespace build
}  // namespace svo
