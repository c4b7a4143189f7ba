// svo_derive.hip.h -- the interior-descriptor table: a derived acceleration copy of the pool.
//
// What the reference does per iteration (svotrace.comp:286-313): fetch the child record, then look at its value
// byte (empty?), its tag (from the parent's leafMask) and its child pointer (leaf?).  48 % of the iterations are
// ADVANCEs that only needed "is this child empty", and every HIT needs nothing but "non-empty, no child block" -- the
// reference's layout keeps those two bits inside the child, so a cast pays a dependent, unaligned 7-byte fetch per
// iteration.  SURVEY 8(b) allows "a derived acceleration copy inside the library, results must not change": this is it.
//
// One 8-byte descriptor per PARENT STATE of the traversal, i.e. per pair (B = child-block base, M = tag mask) the
// walk can hold (svotrace.comp:291: extractChild(parent.descriptor, parent.childPointer, ..., parent.leafMask) is a
// function of exactly that pair):
//   desc[i] = { byte offset of the descriptor of its first child the walk can descend into, minus 64,
//               a nibble per child slot c }    0: child c's value byte == 0 -- empty               (svotrace.comp:295)
//                                              1: not empty, nothing below it: a HIT               (svotrace.comp:311, extractNode)
//                                              8 | rank: not empty, tag 0 with a non-zero cp -- the walk descends, into the
//                                                        rank-th of this state's child descriptors
//   aux[i]  = { B, M }               only read after the loop: hit pointer = B + offset(c, M) (svotrace.comp:381)
// Children the walk can descend into get consecutive descriptors, so DESCEND is desc.x + 8 * nibble -- one shift-add on the
// nibble the trip has looked at anyway (rounds 2-4 kept the two bit masks ne | has << 8 and counted the has bits below c:
// three more vector instructions in 93 % of the trips); ADVANCE and HIT touch no memory at all; POP re-reads the ancestor's
// descriptor (the LDS stack entry shrinks to {descriptor offset, t_max}).  An EMPTY child with a child block (fuzzed pools)
// is never descended into (svotrace.comp:295 comes first) and gets no descriptor.
//
// The table is an unrolling of the pool from the root pair, level by level (13 levels: a 13-level pool is the
// reference's limit, MAX_DEPTH), so it states exactly what the byte walk would read -- also for pools no builder
// produces (children that overlap, point backwards, lie past the end: reads there give 0 through the same buffer
// descriptor the byte walk uses).  Descriptor 0 is the pair (0, 0) that a POP to a never-pushed stack level restores
// (the reference's zero-initialised octstack): its children are the records at bytes 0, 7, .., 49 read as interior
// nodes; where those are the root / the root's children their descriptors are shared, otherwise unrolled as well.
// A pool that is still interior at level 13, or whose unrolling outgrows the budget (cycles), is left to the byte walk
// (Table::ok = false): every result stays the reference's either way.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "svo_build.hip.h"
#include "svo_trav.h"
#include "svo_descword.h"

namespace svo {
namespace derive {

constexpr uint32_t kPhantom = 0u, kRoot = 1u;
constexpr int kLevels = kMaxDepth;   // parent states at depth 0..12
constexpr size_t kHeadroom = 1u << 16;

struct Table {
  uint2 *desc = nullptr;
  uint2 *aux = nullptr;
  uint32_t count = 0;      // descriptors
  size_t cap = 0;          // allocated descriptors
  bool ok = false;         // the persistent pipeline may walk it
  int levels = 0;
  float build_ms = 0.0f;
  // incremental refresh after svo_pool_update (refresh_table): scratch that lives as long as the table + what the last one did
  uint32_t *list = nullptr;   // indices of the states a changed byte range touches
  uint32_t *ctr = nullptr;    // device counters: [0] length of the list, [1] end of the table, [2] failure flags
  uint32_t refreshes = 0, refresh_states = 0, refresh_added = 0;
  float refresh_ms = 0.0f;
};

inline void free_table(Table &t) {
  if (t.desc) (void)hipFree(t.desc);
  if (t.aux) (void)hipFree(t.aux);
  if (t.list) (void)hipFree(t.list);
  if (t.ctr) (void)hipFree(t.ctr);
  t = Table();
}

// child c of the parent state (B, M): is it non-empty, can the walk descend into it (non-empty, with a child block), and which
// state is it as a parent
__device__ __forceinline__ void child_of(const BufPool &pool, uint32_t B, uint32_t M, uint32_t c, bool &ne, bool &has, uint2 &key) {
  const uint32_t tag = (M >> (2u * c)) & 3u;
  const uint32_t ptr = B + child_offset(M, c);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pool.rsrc, (int)ptr, 0, 0);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pool.rsrc, (int)(ptr + 4u), 0, 0);
  ne = (lo & 0xffu) != 0u;
  const uint32_t cp = tag == 0u ? rec2_cp(lo, hi) : 0u;
  has = ne && cp != 0u;
  key = make_uint2(ptr + cp, rec2_mask_be(hi));
}

// pass 1 over the parent states [lo, lo + n): the ne / has bits of their eight children
__global__ __launch_bounds__(256) void masks_kernel(const uint8_t *pool_base, uint32_t pool_len, const uint2 *aux, uint32_t lo,
                                                    uint32_t n, uint16_t *masks, uint8_t *hasmask) {
  const BufPool pool = make_bufpool(pool_base, pool_len);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 3, c = t & 7u;
  bool ne = false, has = false;
  if (i < n) {
    const uint2 k = aux[lo + i];
    uint2 ck;
    child_of(pool, k.x, k.y & 0xffffu, c, ne, has, ck);
  }
  const unsigned long long bn = __ballot(ne), bh = __ballot(has);
  if (i < n && c == 0u) {
    const uint32_t sh = threadIdx.x & 56u;
    const uint32_t m_ne = (uint32_t)(bn >> sh) & 0xffu, m_has = (uint32_t)(bh >> sh) & 0xffu;
    masks[i] = (uint16_t)(m_ne | (m_has << 8));
    hasmask[i] = (uint8_t)m_has;
  }
}

// pass 2: descriptors of [lo, lo + n); the states of their children with a child block are appended from `next` on
__global__ __launch_bounds__(256) void place_kernel(const uint8_t *pool_base, uint32_t pool_len, uint2 *aux, uint2 *desc,
                                                    uint32_t lo, uint32_t n, const uint16_t *masks, const uint32_t *first,
                                                    uint32_t next, uint32_t cap, int last_level, uint32_t depth) {
  const BufPool pool = make_bufpool(pool_base, pool_len);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 3, c = t & 7u;
  if (i >= n) return;
  const uint32_t m = masks[i], has = m >> 8;
  const uint32_t base = next + first[i];
  if (c == 0u) desc[lo + i] = make_uint2(last_level ? 0u : desc_base(base), desc_word(m & 0xffu, has));
  if (last_level || !((has >> c) & 1u)) return;
  const uint32_t j = base + (uint32_t)__builtin_popcount(has & ((1u << c) - 1u));
  if (j >= cap) return;
  const uint2 k = aux[lo + i];
  bool ne, hs;
  uint2 ck;
  child_of(pool, k.x, k.y & 0xffffu, c, ne, hs, ck);
  aux[j] = make_uint2(ck.x, ck.y | ((depth + 1u) << 16));   // bits 16..19: the state's depth (refresh_table checks the 13-level limit with it)
}

// Expand the states [lo, lo + n): returns how many child states were appended at `next` (0xffffffff on overflow of the
// budget); with last_level set nothing is appended and the return value is the number of children that WOULD have
// been (non-zero = the pool is deeper than the table).
inline uint32_t expand(build::Builder &B, Table &t, const uint8_t *pool, uint32_t pool_len, uint32_t lo, uint32_t n, uint32_t next,
                       bool last_level, uint32_t depth) {
  if (n == 0) return 0;
  uint16_t *masks = B.alloc<uint16_t>(n);
  uint8_t *hasmask = B.alloc<uint8_t>(n);
  uint32_t *first = B.alloc<uint32_t>(n);
  if (!B.ok()) return 0xffffffffu;
  const unsigned grid = (unsigned)(((size_t)n * 8 + 255) / 256);
  hipLaunchKernelGGL(masks_kernel, dim3(grid), dim3(256), 0, B.stream, pool, pool_len, t.aux, lo, n, masks, hasmask);
  uint32_t total = 0;
  B.exclusive_scan<uint8_t, true>(hasmask, first, n, &total);
  if (!B.ok()) return 0xffffffffu;
  if (!last_level && (uint64_t)next + total > t.cap) return 0xffffffffu;
  hipLaunchKernelGGL(place_kernel, dim3(grid), dim3(256), 0, B.stream, pool, pool_len, t.aux, t.desc, lo, n, masks, first, next,
                     (uint32_t)t.cap, last_level ? 1 : 0, depth);
  return total;
}

// One attempt with room for `want` states; *overflow is set when the unrolling needed more.
inline hipError_t build_table_sized(Table &t, const uint8_t *d_pool, uint32_t pool_len, size_t want, hipStream_t stream, bool *overflow);

// (Re)build the table for the pool.  hipSuccess with t.ok = false means "not derivable, use the byte walk".
// Budget: a proper tree has one state per interior node with a child block, i.e. per >= 7 + 8 bytes of pool -- the first
// attempt allocates for that (8192^3 bench scene: 67.7 M states of 96 M); pools whose phantom / overlapping children
// unroll into more get a second attempt with one state per 8 bytes of pool; beyond that the records are walked.
inline hipError_t build_table(Table &t, const uint8_t *d_pool, uint64_t pool_len64, hipStream_t stream) {
  const uint32_t pool_len = (uint32_t)pool_len64;
  bool overflow = false;
  // (+ kHeadroom: what refresh_table may append before a table of a small pool has to be rebuilt)
  hipError_t e = build_table_sized(t, d_pool, pool_len, (size_t)pool_len / 15 + 4096 + kHeadroom, stream, &overflow);
  if (e != hipSuccess || !overflow) return e;
  return build_table_sized(t, d_pool, pool_len, (size_t)pool_len / 8 + 4096 + kHeadroom, stream, &overflow);
}

inline hipError_t build_table_sized(Table &t, const uint8_t *d_pool, uint32_t pool_len, size_t want, hipStream_t stream, bool *overflow) {
  *overflow = false;
  t.ok = false; t.count = 0; t.levels = 0;
  t.refresh_states = 0; t.refresh_added = 0; t.refresh_ms = 0.0f;   // (t.refreshes counts over the table's life)
  if (want >= (1u << 28)) return hipSuccess;   // descriptor byte offsets must fit 31 bits
  hipError_t e;
  if (t.cap < want || t.cap > 2 * want) {
    if (t.desc) (void)hipFree(t.desc);
    if (t.aux) (void)hipFree(t.aux);
    t.desc = nullptr; t.aux = nullptr; t.cap = 0;
    if ((e = hipMalloc((void **)&t.desc, want * sizeof(uint2))) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&t.aux, want * sizeof(uint2))) != hipSuccess) return e;
    t.cap = want;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, stream);
  build::Builder B;
  B.stream = stream;
  struct Cleanup {   // whichever way the function is left: scratch and events go
    build::Builder &b; hipEvent_t &a, &z;
    ~Cleanup() { b.release(); if (a) (void)hipEventDestroy(a); if (z) (void)hipEventDestroy(z); }
  } cleanup{B, e0, e1};
  // the first 64 bytes of the pool on the host: the root record and the records the phantom state sees
  uint8_t head[64];
  if ((e = hipMemcpyAsync(head, d_pool, 64, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  for (uint32_t i = 0; i < 64; i++) if (i >= pool_len) head[i] = 0;
  auto rec_key = [&](uint32_t at, bool &ne, bool &has) {   // the record at byte `at`, read as an interior node
    ne = head[at] != 0;
    const uint32_t cp = ((uint32_t)head[at + 1] << 24) | ((uint32_t)head[at + 2] << 16) | ((uint32_t)head[at + 3] << 8) | head[at + 4];
    has = ne && cp != 0;
    return make_uint2(at + cp, ((uint32_t)head[at + 5] << 8) | head[at + 6]);
  };
  bool ne0, has0;
  const uint2 seeds[2] = {make_uint2(0u, 0u), rec_key(0, ne0, has0)};   // phantom, root (svotrace.comp:222)
  if ((e = hipMemcpyAsync(t.aux, seeds, sizeof seeds, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;

  bool ok = true;
  uint32_t lo = kRoot, n = 1, end = 2;
  int depth = 0;
  for (; depth < kLevels && n > 0; depth++) {
    const bool last = depth == kLevels - 1;
    const uint32_t got = expand(B, t, d_pool, pool_len, lo, n, end, last, (uint32_t)depth);
    if (got == 0xffffffffu) { ok = false; *overflow = B.err == hipSuccess; break; }
    if (last) { if (got) ok = false; break; }
    lo = end; n = got; end += got;
  }
  t.levels = depth + (n > 0 ? 1 : 0);
  if (B.ok() && ok) {
    // the phantom state (0, 0): children = the records at 0, 7, .., 49 read as interior nodes
    uint2 rootd, rootkids_aux[8], rootkids_desc[8];
    if ((e = hipMemcpyAsync(&rootd, t.desc + kRoot, sizeof rootd, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    const uint32_t nkids = (uint32_t)__builtin_popcount(desc_has(rootd.y)), kid0 = desc_first(rootd.x);
    if (nkids) {
      if ((e = hipMemcpyAsync(rootkids_aux, t.aux + kid0, nkids * sizeof(uint2), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
      if ((e = hipMemcpyAsync(rootkids_desc, t.desc + kid0, nkids * sizeof(uint2), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
      if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    }
    uint32_t m_ne = 0, m_has = 0, nph = 0;
    uint2 ph_aux[8], ph_desc[8];
    bool ph_known[8];
    for (uint32_t c = 0; c < 8; c++) {
      bool ne, has;
      const uint2 key = rec_key(7u * c, ne, has);
      if (ne) m_ne |= 1u << c;
      if (!has) continue;
      m_has |= 1u << c;
      ph_aux[nph] = make_uint2(key.x, key.y | (1u << 16)); ph_known[nph] = false;   // (depth 1, like the root's children)
      if (key.x == seeds[1].x && key.y == seeds[1].y) { ph_desc[nph] = rootd; ph_known[nph] = true; }
      for (uint32_t k = 0; k < nkids && !ph_known[nph]; k++)
        if (key.x == rootkids_aux[k].x && key.y == (rootkids_aux[k].y & 0xffffu)) { ph_desc[nph] = rootkids_desc[k]; ph_known[nph] = true; }
      if (!ph_known[nph]) ph_desc[nph] = make_uint2(0u, 0u);
      nph++;
    }
    if ((uint64_t)end + nph > t.cap) { ok = false; *overflow = true; }
    if (ok) {
      const uint2 phd = make_uint2(desc_base(end), desc_word(m_ne, m_has));
      if ((e = hipMemcpyAsync(t.desc + kPhantom, &phd, sizeof phd, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
      if (nph) {
        if ((e = hipMemcpyAsync(t.aux + end, ph_aux, nph * sizeof(uint2), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(t.desc + end, ph_desc, nph * sizeof(uint2), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
      }
      const uint32_t ph0 = end;
      end += nph;
      // children of the phantom that are neither the root nor one of its children: unrolled like the root
      uint32_t lo2 = end;
      for (uint32_t k = 0; k < nph && ok; k++) {
        if (ph_known[k]) continue;
        const uint32_t got = expand(B, t, d_pool, pool_len, ph0 + k, 1, end, false, 1u);
        if (got == 0xffffffffu) { ok = false; *overflow = B.err == hipSuccess; break; }
        end += got;
      }
      uint32_t n2 = end - lo2;
      for (int d = 2; d < kLevels && n2 > 0 && ok; d++) {
        const bool last = d == kLevels - 1;
        const uint32_t got = expand(B, t, d_pool, pool_len, lo2, n2, end, last, (uint32_t)d);
        if (got == 0xffffffffu) { ok = false; *overflow = B.err == hipSuccess; break; }
        if (last) { if (got) ok = false; break; }
        lo2 = end; n2 = got; end += got;
      }
    }
  }
  (void)hipEventRecord(e1, stream);
  e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = B.err;
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) (void)hipEventElapsedTime(&t.build_ms, e0, e1);
  if (e != hipSuccess) return e;
  t.count = end;
  t.ok = ok;
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------
// Incremental refresh after svo_pool_update(start, end) (Octree.useSDFBrush -> Renderer.updateSSBO, Main.java:349-350):
// an edit rewrites a few records and appends new child blocks; rebuilding 67.7 M states for it costs 6.7 ms at 8192^3.
// What a state (B, M) states depends on the bytes of its child block [B, B + size(M)) only, so:
//   1. every state whose child block overlaps [start, end) is listed (one pass over aux: 0.5 GB at 8192^3);
//   2. its masks are recomputed from the pool.  Same children with a child block, same (B', M') each: the masks are
//      rewritten in place.  Otherwise the state gets a NEW group of consecutive child descriptors at the table's end:
//      children whose (B', M') did not change are copied there with their descriptor (the copy shares the subtree),
//      the others are new states (desc.x = kNew).  The old group stays behind as garbage until the next full build.
//      A copy may itself overlap the range, and the copy may have been taken before or after its original was
//      rewritten in the same launch: the appended states are listed and recomputed again until nothing is listed;
//   3. the new states are unrolled level by level like build_table does, but with a counter instead of a scan
//      (groups need to be consecutive, levels need not be in any order) and the depth taken from aux bits 16..19.
// Any doubt -- the root record in the range, list or table full, a state at depth 12 that has a child block, more than 13 levels -- gives up:
// the caller drops the table and the next dispatch rebuilds it whole.
constexpr uint32_t kNew = 0xffffffffu;
enum : uint32_t { kCtrList = 0, kCtrEnd = 1, kCtrFlags = 2 };
enum : uint32_t { kFlagListFull = 1u, kFlagTableFull = 2u, kFlagTooDeep = 4u };
constexpr uint32_t kListCap = 1u << 22;

__global__ __launch_bounds__(256) void affected_kernel(const uint2 *aux, const uint2 *desc, uint32_t lo, uint32_t n, uint32_t start,
                                                       uint32_t end, uint32_t *list, uint32_t *ctr, int may_be_new) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= n) return;
  const uint32_t i = lo + t;
  if (may_be_new && desc[i].x == kNew) return;   // not unrolled yet: step 3 reads the edited pool anyway
  const uint2 a = aux[i];
  const uint64_t last = (uint64_t)a.x + child_offset(a.y & 0xffffu, 8u);   // one past the child block
  if (a.x >= end || last <= start) return;
  const uint32_t k = atomicAdd(ctr + kCtrList, 1u);
  if (k < kListCap) list[k] = i;
  else atomicOr(ctr + kCtrFlags, kFlagListFull);
}

__global__ __launch_bounds__(256) void recompute_kernel(const uint8_t *pool_base, uint32_t pool_len, uint2 *aux, uint2 *desc,
                                                        const uint32_t *list, uint32_t n, uint32_t cap, uint32_t *ctr) {
  const BufPool pool = make_bufpool(pool_base, pool_len);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, k = t >> 3, c = t & 7u;
  const bool live = k < n;
  bool ne = false, has = false, same = true;
  uint2 ck = make_uint2(0u, 0u), me = make_uint2(0u, 0u), od = make_uint2(0u, 0u);
  uint32_t i = 0, jold = 0;
  if (live) {
    i = list[k]; me = aux[i]; od = desc[i];
    child_of(pool, me.x, me.y & 0xffffu, c, ne, has, ck);
    const uint32_t old_has = desc_has(od.y);
    const bool old_c = ((old_has >> c) & 1u) != 0u;
    if (has != old_c) same = false;
    else if (has) {
      jold = desc_first(od.x) + (uint32_t)__builtin_popcount(old_has & ((1u << c) - 1u));
      const uint2 oa = aux[jold];
      same = oa.x == ck.x && (oa.y & 0xffffu) == ck.y;
    }
  }
  const unsigned long long bn = __ballot(ne), bh = __ballot(has), bs = __ballot(same);
  if (!live) return;
  const uint32_t sh = threadIdx.x & 56u;
  const uint32_t m_ne = (uint32_t)(bn >> sh) & 0xffu, m_has = (uint32_t)(bh >> sh) & 0xffu;
  const uint32_t masks = desc_word(m_ne, m_has);
  if (((uint32_t)(bs >> sh) & 0xffu) == 0xffu) {   // same children with a child block, same states: the bits only
    if (c == 0u) desc[i] = make_uint2(od.x, masks);
    return;
  }
  const uint32_t d = (me.y >> 16) & 15u, cnt = (uint32_t)__builtin_popcount(m_has);
  if (d + 1u >= (uint32_t)kLevels && cnt) { if (c == 0u) atomicOr(ctr + kCtrFlags, kFlagTooDeep); return; }
  uint32_t base = 0;
  if (c == 0u) base = atomicAdd(ctr + kCtrEnd, cnt);
  base = (uint32_t)__shfl((int)base, (int)(threadIdx.x & 56u));
  if ((uint64_t)base + cnt > cap) { if (c == 0u) atomicOr(ctr + kCtrFlags, kFlagTableFull); return; }
  if (c == 0u) desc[i] = make_uint2(desc_base(base), masks);
  if (!has) return;
  const uint32_t j = base + (uint32_t)__builtin_popcount(m_has & ((1u << c) - 1u));
  if (same) { desc[j] = desc[jold]; aux[j] = aux[jold]; }   // (same && has: it was there before, with this (B', M'))
  else { aux[j] = make_uint2(ck.x, ck.y | ((d + 1u) << 16)); desc[j] = make_uint2(kNew, 0u); }
}

__global__ __launch_bounds__(256) void unroll_new_kernel(const uint8_t *pool_base, uint32_t pool_len, uint2 *aux, uint2 *desc,
                                                         uint32_t lo, uint32_t n, uint32_t cap, uint32_t *ctr) {
  const BufPool pool = make_bufpool(pool_base, pool_len);
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, k = t >> 3, c = t & 7u, i = lo + k;
  const bool live = k < n && desc[i].x == kNew;
  bool ne = false, has = false;
  uint2 ck = make_uint2(0u, 0u), me = make_uint2(0u, 0u);
  if (live) { me = aux[i]; child_of(pool, me.x, me.y & 0xffffu, c, ne, has, ck); }
  const unsigned long long bn = __ballot(ne), bh = __ballot(has);
  if (!live) return;
  const uint32_t sh = threadIdx.x & 56u;
  const uint32_t m_ne = (uint32_t)(bn >> sh) & 0xffu, m_has = (uint32_t)(bh >> sh) & 0xffu;
  const uint32_t masks = desc_word(m_ne, m_has);
  const uint32_t d = (me.y >> 16) & 15u, cnt = (uint32_t)__builtin_popcount(m_has);
  if (d + 1u >= (uint32_t)kLevels) {   // the last level the walk can stand on: no child blocks below it
    if (c == 0u) { desc[i] = make_uint2(0u, masks); if (cnt) atomicOr(ctr + kCtrFlags, kFlagTooDeep); }
    return;
  }
  uint32_t base = 0;
  if (c == 0u) base = atomicAdd(ctr + kCtrEnd, cnt);
  base = (uint32_t)__shfl((int)base, (int)(threadIdx.x & 56u));
  if ((uint64_t)base + cnt > cap) { if (c == 0u) atomicOr(ctr + kCtrFlags, kFlagTableFull); return; }
  if (c == 0u) desc[i] = make_uint2(desc_base(base), masks);
  if (!has) return;
  const uint32_t j = base + (uint32_t)__builtin_popcount(m_has & ((1u << c) - 1u));
  aux[j] = make_uint2(ck.x, ck.y | ((d + 1u) << 16));
  desc[j] = make_uint2(kNew, 0u);
}

// Bring the table up to date with the pool after the bytes [start, end) changed.  *refreshed = false (and a table that may
// be half rewritten: drop it) when the refresh gave up.
inline hipError_t refresh_table(Table &t, const uint8_t *d_pool, uint64_t pool_len64, uint64_t start64, uint64_t end64,
                                hipStream_t stream, bool *refreshed) {
  *refreshed = false;
#if defined(SVO_VARIANTS) && SVO_VARIANTS
  static const bool log = []() { const char *e = getenv("SVO_DERIVED_REFRESH_LOG"); return e && e[0] == '1'; }();
#else
  const bool log = false;
#endif
  auto gave_up = [&](const char *why, uint32_t flags) {
    if (log) fprintf(stderr, "[svo] table refresh [%llu, %llu) gave up: %s (flags %u, %u of %zu descriptors)\n", (unsigned long long)start64,
                     (unsigned long long)end64, why, flags, t.count, t.cap);
    return hipSuccess;
  };
  if (!t.ok || !t.desc || t.count < 2 || end64 > 0xffffffffull || start64 >= end64) return hipSuccess;
  // the root state is not stated by any other state but by the record at byte 0 (svotrace.comp:222): when that record
  // may have changed, the table is built anew
  if (start64 < 7) {
    uint8_t head[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint2 root;
    hipError_t r = hipMemcpyAsync(head, d_pool, std::min<uint64_t>(7, pool_len64), hipMemcpyDeviceToHost, stream);
    if (r == hipSuccess) r = hipMemcpyAsync(&root, t.aux + kRoot, sizeof root, hipMemcpyDeviceToHost, stream);
    if (r == hipSuccess) r = hipStreamSynchronize(stream);
    if (r != hipSuccess) return r;
    const uint32_t cp = ((uint32_t)head[1] << 24) | ((uint32_t)head[2] << 16) | ((uint32_t)head[3] << 8) | head[4];
    if (root.x != cp || (root.y & 0xffffu) != (((uint32_t)head[5] << 8) | head[6])) return gave_up("the root record changed", 0);
  }
  const uint32_t pool_len = (uint32_t)pool_len64, start = (uint32_t)start64, end = (uint32_t)end64;
  hipError_t e;
  if (!t.list && (e = hipMalloc((void **)&t.list, (size_t)kListCap * 4)) != hipSuccess) return e;
  if (!t.ctr && (e = hipMalloc((void **)&t.ctr, 8 * 4)) != hipSuccess) return e;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct Cleanup { hipEvent_t &a, &z; ~Cleanup() { if (a) (void)hipEventDestroy(a); if (z) (void)hipEventDestroy(z); } } cleanup{e0, e1};
  (void)hipEventRecord(e0, stream);
  uint32_t h[3] = {0u, t.count, 0u};
  auto put = [&]() { return hipMemcpyAsync(t.ctr, h, sizeof h, hipMemcpyHostToDevice, stream); };
  auto get = [&]() {
    hipError_t r = hipMemcpyAsync(h, t.ctr, sizeof h, hipMemcpyDeviceToHost, stream);
    return r != hipSuccess ? r : hipStreamSynchronize(stream);
  };
  const uint32_t cap = (uint32_t)std::min<size_t>(t.cap, 0xffffffffu);
  uint32_t listed = 0, scan_lo = 0, scan_hi = t.count;
  bool settled = false;
  for (int pass = 0; pass < kLevels + 2; pass++) {   // step 1 + 2, again over what step 2 appended
    h[kCtrList] = 0;
    if ((e = put()) != hipSuccess) return e;
    const uint32_t n = scan_hi - scan_lo;
    hipLaunchKernelGGL(affected_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, t.aux, t.desc, scan_lo, n, start, end, t.list, t.ctr,
                       pass > 0 ? 1 : 0);
    if ((e = get()) != hipSuccess) return e;
    if (h[kCtrFlags]) return gave_up("list full", h[kCtrFlags]);
    const uint32_t naff = h[kCtrList];
    if (naff == 0) { settled = true; break; }
    listed += naff;
    hipLaunchKernelGGL(recompute_kernel, dim3((unsigned)(((size_t)naff * 8 + 255) / 256)), dim3(256), 0, stream, d_pool, pool_len, t.aux,
                       t.desc, t.list, naff, cap, t.ctr);
    if ((e = get()) != hipSuccess) return e;
    if (h[kCtrFlags]) return gave_up("recompute", h[kCtrFlags]);
    scan_lo = scan_hi; scan_hi = h[kCtrEnd];
    if (scan_lo == scan_hi) { settled = true; break; }
  }
  if (!settled) return gave_up("copies kept changing", 0);
  uint32_t lo = t.count, hi = h[kCtrEnd];   // step 3: [lo, hi) holds copies and new states, what follows new states only
  for (int level = 0; lo < hi; level++) {
    if (level > kLevels) return gave_up("more than 13 levels", 0);
    const uint32_t n = hi - lo;
    hipLaunchKernelGGL(unroll_new_kernel, dim3((unsigned)(((size_t)n * 8 + 255) / 256)), dim3(256), 0, stream, d_pool, pool_len, t.aux, t.desc,
                       lo, n, cap, t.ctr);
    if ((e = get()) != hipSuccess) return e;
    if (h[kCtrFlags]) return gave_up("unroll", h[kCtrFlags]);
    lo = hi; hi = h[kCtrEnd];
  }
  (void)hipEventRecord(e1, stream);
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  if ((e = hipGetLastError()) != hipSuccess) return e;
  (void)hipEventElapsedTime(&t.refresh_ms, e0, e1);
  t.refreshes++; t.refresh_states = listed; t.refresh_added = hi - t.count;
  if (log) fprintf(stderr, "[svo] table refresh [%llu, %llu): %u states recomputed, %u descriptors appended, %.3f ms\n",
                   (unsigned long long)start64, (unsigned long long)end64, listed, hi - t.count, t.refresh_ms);
  t.count = hi;
  *refreshed = true;
  return hipSuccess;
}

}  // namespace derive
}  // namespace svo
