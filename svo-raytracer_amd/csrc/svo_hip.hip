// svo_hip.hip -- C-ABI implementation of include/svo_hip.h (libsvohip.so).
// The reference-side interface each entry point replaces is cited in the header.
#include "../../include/svo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// SVO_VARIANTS=0 (libsvohip.so, what a host loads): pipeline 1 -- persistent waves over the descriptor table, with the record
// walk for pools the table cannot state -- and pipeline 0 (the reference's one-thread-per-pixel decomposition: its kernel is also
// the counting pass of svo_count_frame), the beam pass, the builder, the ring, the group.
// SVO_VARIANTS=1 (libsvohip_variants.so, `make variants`; loaded by the tests that compare against them, like cxxloop): the same
// plus the comparators -- pipeline 2 (staged wavefront tracing), the spare-ray kernel of round 5 (SVO_SPARE=1) -- and the
// environment switches of the A/B runs (SVO_DERIVED, SVO_RC_TABLE, SVO_NORMAL_TABLE, SVO_FORCE_CAMS, SVO_PERSIST_*, ...).
#ifndef SVO_VARIANTS
#define SVO_VARIANTS 0
#endif
#include "svo_fused.hip.h"
#include "svo_persistent.hip.h"
#if SVO_ASM_LOOP && SVO_VARIANTS
#include "variants/svo_persist2.hip.h"
#endif
#if SVO_VARIANTS
#include "variants/svo_wavefront.hip.h"
#endif
#include "svo_build.hip.h"
#include "svo_beam.hip.h"
#include "svo_derive.hip.h"

using namespace svo;

static_assert(sizeof(svo_hit) == 16, "svo_hit must be 16 bytes");

struct svo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // the stream dispatches go to: own_stream, or the caller's (svo_set_stream)
  hipStream_t own_stream = nullptr;   // created with the context, kept for its lifetime
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // pool
  uint8_t *d_pool = nullptr;
  uint64_t pool_len = 0, pool_cap = 0;
  uint32_t dword0 = 0;
  // frame state
  float cam[15] = {1.5f, 1.5f, 2.0f, -1.6f, -0.9f, -1.f, -1.6f, 0.9f, -1.f, 1.6f, -0.9f, -1.f, 1.6f, 0.9f, -1.f};
  int width = 0, height = 0, y0 = 0, y1 = 0;
  bool rows_set = false;
  int row_step = 1, out_y0 = 0, n_tile_rows = -1;  // stripe mode (svo_set_stripes); n_tile_rows < 0 = band mode
  int frame_number = 2, render_mode = 2, buffer_end = 0, use_beam = 0, bounces = 2, spp = 1, progressive = 0;
  int seq = 1, seq_fresh = 0;    // progressive: frames of the accumulation per dispatch, on a zeroed image or not (svo_set_sequence)
  const FrameVar *batch_cams = nullptr;     // host copy of the cameras / frame numbers of the batch being submitted
  FrameVar *batch_cams_dev = nullptr;       // (svo_ring_submit_cams), and where the slot keeps them on the device
  void *batch_cams_slot = nullptr;          // the ring slot being submitted (its staging copy: ring_copy_cams)
  int batch = 1;                 // frames per dispatch (svo_set_batch)
  uint64_t frame_stride = 0;     // elements between consecutive frames of a batch in each output
  uint32_t mirror_mask = 0;
  int pipeline = 1;        // persistent waves on the descriptor table; 0 / 2: svo_set_pipeline
  int write_hits = 1;
  // outputs
  uint32_t *d_color = nullptr;
  float *d_depth = nullptr;
  uint4 *d_hits = nullptr;
  bool external_outputs = false;
  uint32_t *own_color = nullptr;
  float *own_depth = nullptr;
  uint4 *own_hits = nullptr;
  // svo_dispatch_async takes turns on `overlap` sets {stream, images} while the library owns them (the reference's loop, Main.java:132-146,
  // 257-289: frame N + 1 starts in frame N's tail): set 0 = own_stream + own_*, sets 1 .. overlap - 1 = alt_* (made on first use).
  // c->stream / c->d_* always name ONE set, the one of the last dispatch, so every read-back sees the last dispatched frame.
  static constexpr int kMaxSets = 8;
  hipStream_t alt_stream[kMaxSets] = {};     // [0] unused: set 0 is own_stream / own_*
  uint32_t *alt_color[kMaxSets] = {};
  float *alt_depth[kMaxSets] = {};
  uint4 *alt_hits[kMaxSets] = {};
  hipEvent_t set_done[kMaxSets] = {};        // behind the last overlapped dispatch into each set (incl. set 0): a set is rendered
  bool set_used[kMaxSets] = {};              // into again only once that frame is complete -- at most `overlap` frames are queued
  int cur_set = 0;
  int overlap = 4;             // image sets svo_dispatch_async takes turns on (svo_set_overlap; 1 = no alternation)
  bool alt_inflight = false;   // a set that is not current may still have a frame in flight
  hipStream_t set_stream(int k) const { return k == 0 ? own_stream : alt_stream[k]; }
  bool is_own_stream(hipStream_t st) const {
    if (st == own_stream) return true;
    for (int k = 1; k < kMaxSets; k++) if (alt_stream[k] && st == alt_stream[k]) return true;
    return false;
  }
  // the pick pixel (svo_set_pick): answered from pinned host memory by the lane that stores it
  int pick_x = -1, pick_y = -1;
  bool pick_default = true;    // follows the image centre (the crosshair, Main.java:139-141) until svo_set_pick names a pixel
  uint32_t *pick_mail = nullptr;   // kPickSlots x kPickWords words of host memory the device writes
  hipStream_t pick_stream = nullptr;   // the pick launches' own stream (greatest priority), made on first use
  uint32_t pick_seq = 0;       // sequence number of the last dispatch that carried the pick
  uint64_t pick_from_mail = 0, pick_waited = 0;   // svo_read_pixel calls answered from the mail / by waiting for the frame (svo_pick_info)
  bool pick_live = false;      // the current set's last dispatch carried it (cleared by everything else that renders into the set)
  int pick_live_x = -1, pick_live_y = -1;
  DeviceCounters *d_counters = nullptr;
  float4 *d_ntab = nullptr;   // unit normal of every 16-bit normal code (svo_trav2.h::normal_table_kernel), made with the context
  // beam images: one per frame in flight, re-used round-robin behind the event of the frame that read it last
  static constexpr int kBeamSets = 8;
  float *d_beam[kBeamSets] = {};
  hipEvent_t beam_done[kBeamSets] = {};
  bool beam_used[kBeamSets] = {};
  size_t beam_cap = 0;
  unsigned beam_frames = 0;
  uint8_t *d_beam_live = nullptr;   // which nodes of the pool's top levels a cast can end in or below (svo_beam.hip.h)
  bool beam_live_valid = false;     // cleared by everything that changes the pool
  // interior-descriptor table of the pool (svo_derive.hip.h): rebuilt lazily after every pool change, walked by the
  // persistent pipeline when the pool is derivable
  derive::Table dt;
  bool derived_valid = false;     // cleared by everything that changes the pool
  int derived_mode = 1;           // 0 = always walk the records, 1 = walk the table when there is one
#if SVO_VARIANTS
  WavefrontBuffers wf;
#endif
  PersistBuffers pb;
  // frames in flight behind the boundary (svo_ring_*): per slot a stream of its own, output buffers for up to
  // ring_frames consecutive frames, and the events around its last submission
  struct RingSlot {
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEvent_t e2 = nullptr;      // behind the forward copy of the last submission (svo_ring_forward_slot), made on demand
    uint32_t *color = nullptr; float *depth = nullptr; uint4 *hits = nullptr;      // library-owned
    void *xcolor = nullptr, *xdepth = nullptr, *xhits = nullptr; uint64_t xstride = 0;   // caller-owned (svo_ring_bind_slot)
    int first_frame = 0, nframes = 0;
    bool used = false;
    FrameVar *d_fvar = nullptr;   // per-frame cameras of the slot's last submission (svo_ring_submit_cams) ...
    FrameVar *h_fvar = nullptr;   // ... their pinned staging copy, and the event behind the copy that last read it
    hipEvent_t fvar_copied = nullptr;
    // svo_ring_forward_slot: after every submission, src -> dst (a peer's memory) and the submission's number -> *fwd_flag
    const void *fwd_src = nullptr; void *fwd_dst = nullptr; uint64_t fwd_bytes = 0; void *fwd_flag = nullptr;
    int fwd_dst_device = -1;      // >= 0: dst lives on that device of THIS process (svo_group_*): a peer copy
    uint32_t *seq_word = nullptr;
  };
  std::vector<void *> ipc_opened;
  std::vector<void *> dev_allocs;   // svo_dev_alloc, freed with the context at the latest
  std::vector<RingSlot> ring;
  int reserved_cus = 0;        // CUs per XCD the ring's streams leave free (svo_set_reserved_cus)
  int ring_frames = 0;
  uint64_t ring_stride = 0;    // elements between consecutive frames of a library-owned slot
  unsigned ring_next = 0;
  svo_stats stats{};
  std::string err;
};

static int fail(svo_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(ctx, SVO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
  } while (0)

static constexpr uint64_t kPad = 64;  // zero bytes kept behind the pool (8-byte record loads may overhang)

extern "C" {

int svo_create(int device, svo_ctx **out) {
  if (!out) return SVO_E_INVALID;
  *out = nullptr;
  // Frames in flight (svo_ring_*) want a hardware queue per stream; HIP maps streams onto GPU_MAX_HW_QUEUES queues
  // (default 4: two frame streams on one queue serialise their launches).  Read when the runtime starts, so this only
  // has an effect when the library makes the process's first HIP call (a JVM host); it never overrides the caller's value.
  (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return SVO_E_NODEVICE;
  svo_ctx *c = new svo_ctx();
  c->device = device;
  c->stats.device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipMalloc((void **)&c->d_counters, sizeof(DeviceCounters)) != hipSuccess) {
    delete c;
    return SVO_E_HIP;
  }
  c->stream = c->own_stream;
  // the normal table: 65 536 x 16 bytes, a function of nothing but the code
  if (hipMalloc((void **)&c->d_ntab, 65536 * sizeof(float4)) == hipSuccess) {
    hipLaunchKernelGGL(normal_table_kernel, dim3(256), dim3(256), 0, c->own_stream, c->d_ntab);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->own_stream) != hipSuccess) { (void)hipFree(c->d_ntab); c->d_ntab = nullptr; }
  }
  c->pb.ntab = c->d_ntab;   // (null: the kernels decode in place)
  *out = c;
  return SVO_OK;
}

static void ring_free(svo_ctx *c) {
  for (auto &s : c->ring) {
    if (s.color) (void)hipFree(s.color);
    if (s.depth) (void)hipFree(s.depth);
    if (s.hits) (void)hipFree(s.hits);
    if (s.e0) (void)hipEventDestroy(s.e0);
    if (s.e1) (void)hipEventDestroy(s.e1);
    if (s.e2) (void)hipEventDestroy(s.e2);
    if (s.seq_word) (void)hipFree(s.seq_word);
    if (s.d_fvar) (void)hipFree(s.d_fvar);
    if (s.h_fvar) (void)hipHostFree(s.h_fvar);
    if (s.fvar_copied) (void)hipEventDestroy(s.fvar_copied);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  c->ring.clear();
  c->pb.cus_reserved = 0;
  c->ring_frames = 0; c->ring_stride = 0; c->ring_next = 0;
}

static void free_outputs(svo_ctx *c) {
  ring_free(c);   // the ring's images have the size of the frame
  if (c->own_color) (void)hipFree(c->own_color);
  if (c->own_depth) (void)hipFree(c->own_depth);
  if (c->own_hits) (void)hipFree(c->own_hits);
  c->own_color = nullptr; c->own_depth = nullptr; c->own_hits = nullptr;
  for (int k = 1; k < svo_ctx::kMaxSets; k++) {
    if (c->alt_color[k]) (void)hipFree(c->alt_color[k]);
    if (c->alt_depth[k]) (void)hipFree(c->alt_depth[k]);
    if (c->alt_hits[k]) (void)hipFree(c->alt_hits[k]);
    c->alt_color[k] = nullptr; c->alt_depth[k] = nullptr; c->alt_hits[k] = nullptr;
  }
  if (c->cur_set != 0 && (c->is_own_stream(c->stream) || c->stream == nullptr)) c->stream = c->own_stream;
  c->cur_set = 0; c->alt_inflight = false; c->pick_live = false;
  for (int k = 0; k < svo_ctx::kMaxSets; k++) c->set_used[k] = false;
  if (!c->external_outputs) { c->d_color = nullptr; c->d_depth = nullptr; c->d_hits = nullptr; }
#if SVO_VARIANTS
  wavefront_free(c->wf);
#endif
  persist_free(c->pb);
  for (int i = 0; i < svo_ctx::kBeamSets; i++) {
    if (c->d_beam[i]) (void)hipFree(c->d_beam[i]);
    c->d_beam[i] = nullptr;
    c->beam_used[i] = false;
  }
  c->beam_cap = 0;
}

int svo_destroy(svo_ctx *c) {
  if (!c) return SVO_E_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  free_outputs(c);
  if (c->d_pool) (void)hipFree(c->d_pool);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_ntab) (void)hipFree(c->d_ntab);
  for (auto &e : c->beam_done) if (e) (void)hipEventDestroy(e);
  if (c->d_beam_live) (void)hipFree(c->d_beam_live);
  derive::free_table(c->dt);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  ring_free(c);
  for (void *p : c->ipc_opened) if (p) (void)hipIpcCloseMemHandle(p);
  for (void *p : c->dev_allocs) if (p) (void)hipFree(p);
  for (int k = 1; k < svo_ctx::kMaxSets; k++) if (c->alt_stream[k]) (void)hipStreamDestroy(c->alt_stream[k]);
  for (int k = 0; k < svo_ctx::kMaxSets; k++) if (c->set_done[k]) (void)hipEventDestroy(c->set_done[k]);
  if (c->pick_stream) (void)hipStreamDestroy(c->pick_stream);
  if (c->pick_mail) (void)hipHostFree(c->pick_mail);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return SVO_OK;
}

const char *svo_last_error(const svo_ctx *c) { return c ? c->err.c_str() : "null context"; }

int svo_build_info(void) { return (SVO_VARIANTS ? 1 : 0) | (SVO_ASM_LOOP ? 2 : 0); }

// ---------------------------------------------------------------- pool
static int ensure_pool_capacity(svo_ctx *c, uint64_t need_len, bool keep) {
  if (need_len >= 0x80000000ull) return fail(c, SVO_E_TOOLARGE, "pool must stay below 2^31 bytes");
  const uint64_t need = need_len + kPad;
  if (need <= c->pool_cap) return SVO_OK;
  uint64_t ncap = keep ? std::max<uint64_t>(need, c->pool_cap + c->pool_cap / 4) : need;
  uint8_t *np = nullptr;
  HIPCHK(c, hipMalloc((void **)&np, ncap));
  HIPCHK(c, hipMemsetAsync(np, 0, ncap, c->own_stream));
  if (keep && c->d_pool && c->pool_len)
    HIPCHK(c, hipMemcpyAsync(np, c->d_pool, c->pool_len, hipMemcpyDeviceToDevice, c->own_stream));
  HIPCHK(c, hipStreamSynchronize(c->own_stream));
  if (c->d_pool) HIPCHK(c, hipFree(c->d_pool));   // callers have synchronised the device: no frame still reads it
  c->d_pool = np;
  c->pool_cap = ncap;
  return SVO_OK;
}

static int refresh_dword0(svo_ctx *c) {
  c->dword0 = 0;
  if (c->pool_len >= 4) HIPCHK(c, hipMemcpy(&c->dword0, c->d_pool, 4, hipMemcpyDeviceToHost));
  else if (c->pool_len > 0) HIPCHK(c, hipMemcpy(&c->dword0, c->d_pool, c->pool_len, hipMemcpyDeviceToHost));
  return SVO_OK;
}

int svo_pool_reserve(svo_ctx *c, uint64_t nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  int rc = ensure_pool_capacity(c, nbytes, false);
  if (rc) return rc;
  HIPCHK(c, hipMemsetAsync(c->d_pool, 0, c->pool_cap, c->own_stream));
  HIPCHK(c, hipStreamSynchronize(c->own_stream));
  c->pool_len = nbytes;
  return SVO_OK;
}

int svo_pool_upload_device(svo_ctx *c, const void *dptr, uint64_t nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || (!dptr && nbytes)) return fail(c, SVO_E_INVALID, "svo_pool_upload_device: null buffer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  int rc = ensure_pool_capacity(c, nbytes, false);
  if (rc) return rc;
  if (nbytes) HIPCHK(c, hipMemcpy(c->d_pool, dptr, nbytes, hipMemcpyDeviceToDevice));
  HIPCHK(c, hipMemset(c->d_pool + nbytes, 0, c->pool_cap - nbytes));
  c->pool_len = nbytes;
  return refresh_dword0(c);
}

int svo_pool_upload(svo_ctx *c, const void *host, uint64_t nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || (!host && nbytes)) return fail(c, SVO_E_INVALID, "svo_pool_upload: null buffer");
  HIPCHK(c, hipSetDevice(c->device));
  // frames may be in flight on any stream the caller has handed over (svo_set_stream): the pool changes only
  // when the whole device is idle, so a frame sees either the old bytes or the new ones, never a mix
  HIPCHK(c, hipDeviceSynchronize());
  int rc = ensure_pool_capacity(c, nbytes, false);
  if (rc) return rc;
  if (nbytes) HIPCHK(c, hipMemcpy(c->d_pool, host, nbytes, hipMemcpyHostToDevice));
  // bytes behind the new end must read as zero
  HIPCHK(c, hipMemset(c->d_pool + nbytes, 0, c->pool_cap - nbytes));
  c->pool_len = nbytes;
  return refresh_dword0(c);
}

int svo_pool_update(svo_ctx *c, const void *host_base, uint64_t start, uint64_t end) {
  // a walkable descriptor table follows a ranged update instead of being rebuilt (derive::refresh_table): an SDF brush
  // stroke is two such updates (Main.java:349-350), the table of the 8192^3 scene takes 6.7 ms to build
#if SVO_VARIANTS
  static const bool follow = []() { const char *e = getenv("SVO_DERIVED_REFRESH"); return !(e && e[0] == '0'); }();
#else
  const bool follow = true;
#endif
  const bool had_table = c && follow && c->derived_valid && c->dt.ok;
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || !host_base) return fail(c, SVO_E_INVALID, "svo_pool_update: null buffer");
  if (start >= end) return fail(c, SVO_E_INVALID, "Update SSBO error: Invalid parameters.");
  if (!c->d_pool) return fail(c, SVO_E_NOPOOL, "svo_pool_update before svo_pool_upload");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());   // see svo_pool_upload
  if (end > c->pool_len) {
    int rc = ensure_pool_capacity(c, end, true);
    if (rc) return rc;
  }
  HIPCHK(c, hipMemcpy(c->d_pool + start, (const uint8_t *)host_base + start, end - start, hipMemcpyHostToDevice));
  if (end > c->pool_len) c->pool_len = end;
  if (start < 4) { int rc = refresh_dword0(c); if (rc) return rc; }
  if (had_table) {
    bool refreshed = false;
    hipError_t e = derive::refresh_table(c->dt, c->d_pool, c->pool_len, start, end, c->own_stream, &refreshed);
    if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string("descriptor table refresh: ") + hipGetErrorString(e));
    c->derived_valid = refreshed;   // false: the next dispatch builds it anew
  }
  return SVO_OK;
}

int svo_pool_download(svo_ctx *c, void *host, uint64_t nbytes) {
  if (!c || !host) return fail(c, SVO_E_INVALID, "svo_pool_download: null buffer");
  if (!c->d_pool) return fail(c, SVO_E_NOPOOL, "no pool");
  if (nbytes > c->pool_len) nbytes = c->pool_len;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(host, c->d_pool, nbytes, hipMemcpyDeviceToHost));
  return SVO_OK;
}

int svo_pool_device_ptr(svo_ctx *c, void **dptr, uint64_t *nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || !dptr) return SVO_E_INVALID;
  *dptr = c->d_pool;
  if (nbytes) *nbytes = c->pool_len;
  // the caller may have written the pool through this pointer (RCCL broadcast): re-read dword 0
  if (c->d_pool) {
    HIPCHK(c, hipSetDevice(c->device));
    return refresh_dword0(c);
  }
  return SVO_OK;
}

int svo_pool_commit(svo_ctx *c) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }
  if (!c) return SVO_E_INVALID;
  if (!c->d_pool) return fail(c, SVO_E_NOPOOL, "svo_pool_commit: no pool");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());   // the caller's writes (any stream) and every frame in flight
  return refresh_dword0(c);
}

// ---------------------------------------------------------------- world generation
__global__ void count_zero_bytes_kernel(const uint8_t *p, size_t n, unsigned int *zeros) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int z = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) z += p[i] == 0 ? 1u : 0u;
  if (z) atomicAdd(zeros, z);
}

int svo_build_from_heightmap(svo_ctx *c, const uint16_t *height, const uint8_t *material, int n, uint64_t *out_nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || !height || !material) return fail(c, SVO_E_INVALID, "svo_build_from_heightmap: null map");
  if (n < 8 || n > 8192 || (n & (n - 1))) return fail(c, SVO_E_INVALID, "svo_build_from_heightmap: n must be a power of two in 8..8192");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  const size_t cells = (size_t)n * (size_t)n;
  uint16_t *d_h = nullptr;
  uint8_t *d_m = nullptr;
  unsigned int *d_z = nullptr;
  auto cleanup = [&]() { if (d_h) (void)hipFree(d_h); if (d_m) (void)hipFree(d_m); if (d_z) (void)hipFree(d_z); };
  hipError_t e = hipMalloc((void **)&d_h, cells * 2);
  if (e == hipSuccess) e = hipMalloc((void **)&d_m, cells);
  if (e == hipSuccess) e = hipMalloc((void **)&d_z, 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_h, height, cells * 2, hipMemcpyHostToDevice, c->own_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_m, material, cells, hipMemcpyHostToDevice, c->own_stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_z, 0, 4, c->own_stream);
  unsigned int zeros = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(count_zero_bytes_kernel, dim3(1024), dim3(256), 0, c->own_stream, d_m, cells, d_z);
    e = hipMemcpyAsync(&zeros, d_z, 4, hipMemcpyDeviceToHost, c->own_stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
  if (e != hipSuccess) { cleanup(); return fail(c, SVO_E_HIP, std::string("svo_build_from_heightmap: ") + hipGetErrorString(e)); }
  if (zeros) { cleanup(); return fail(c, SVO_E_INVALID, "svo_build_from_heightmap: material 0 in the material map (0 is the empty voxel)"); }
  build::Result r;
  bool too_large = false;
  e = build::build_pool(d_h, d_m, n, kPad, c->own_stream, r, &too_large);
  cleanup();
  if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string("svo_build_from_heightmap: ") + hipGetErrorString(e));
  if (too_large) {
    if (out_nbytes) *out_nbytes = r.len;
    return fail(c, SVO_E_TOOLARGE, "pool must stay below 2^31 bytes");
  }
  if (c->d_pool) (void)hipFree(c->d_pool);
  c->d_pool = r.pool; c->pool_len = r.len; c->pool_cap = r.cap;
  if (out_nbytes) *out_nbytes = r.len;
  return refresh_dword0(c);
}

int svo_build_from_heightmap16(svo_ctx *c, const uint16_t *raw16, const uint8_t *material, int n, uint64_t *out_nbytes) {
  if (!c || !raw16 || !material) return fail(c, SVO_E_INVALID, "svo_build_from_heightmap16: null map");
  if (n < 8 || n > 8192 || (n & (n - 1))) return fail(c, SVO_E_INVALID, "svo_build_from_heightmap16: n must be a power of two in 8..8192");
  // chunkgen-heightmap.comp:16-19: heightSample = int(r / 65536.0 * 2048) -- exact in float (two powers of two): r >> 5
  std::vector<uint16_t> h((size_t)n * (size_t)n);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(raw16[i] >> 5);
  return svo_build_from_heightmap(c, h.data(), material, n, out_nbytes);
}

int svo_build_from_voxels(svo_ctx *c, const uint8_t *voxels, int n, uint64_t *out_nbytes) {
  if (c) { c->beam_live_valid = false; c->derived_valid = false; }   // the pool is about to change (or to be handed out for writing)
  if (!c || !voxels) return fail(c, SVO_E_INVALID, "svo_build_from_voxels: null grid");
  if (n < 2 || n > 1024 || (n & (n - 1))) return fail(c, SVO_E_INVALID, "svo_build_from_voxels: n must be a power of two in 2..1024");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  const size_t cells = (size_t)n * (size_t)n * (size_t)n;
  uint8_t *d_v = nullptr;
  hipError_t e = hipMalloc((void **)&d_v, cells);
  if (e == hipSuccess) e = hipMemcpyAsync(d_v, voxels, cells, hipMemcpyHostToDevice, c->own_stream);
  build::Result r;
  bool too_large = false;
  if (e == hipSuccess) e = build::build_pool_from_voxels(d_v, n, kPad, c->own_stream, r, &too_large);
  if (d_v) (void)hipFree(d_v);
  if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string("svo_build_from_voxels: ") + hipGetErrorString(e));
  if (too_large) {
    if (out_nbytes) *out_nbytes = r.len;
    return fail(c, SVO_E_TOOLARGE, "pool must stay below 2^31 bytes");
  }
  if (c->d_pool) (void)hipFree(c->d_pool);
  c->d_pool = r.pool; c->pool_len = r.len; c->pool_cap = r.cap;
  if (out_nbytes) *out_nbytes = r.len;
  return refresh_dword0(c);
}

// ---------------------------------------------------------------- frame state
int svo_set_camera(svo_ctx *c, const float pos[3], const float l1[3], const float l2[3], const float r1[3],
                   const float r2[3]) {
  if (!c || !pos || !l1 || !l2 || !r1 || !r2) return fail(c, SVO_E_INVALID, "svo_set_camera: null vector");
  memcpy(c->cam + 0, pos, 12); memcpy(c->cam + 3, l1, 12); memcpy(c->cam + 6, l2, 12);
  memcpy(c->cam + 9, r1, 12); memcpy(c->cam + 12, r2, 12);
  return SVO_OK;
}

int svo_set_params(svo_ctx *c, int frame_number, int render_mode, int buffer_end, int use_beam, int bounces,
                   uint32_t mirror_mask, int spp) {
  if (!c) return SVO_E_INVALID;
  if (bounces < 1 || spp < 1) return fail(c, SVO_E_INVALID, "bounces and spp must be >= 1");
  c->frame_number = frame_number; c->render_mode = render_mode; c->buffer_end = buffer_end; c->use_beam = use_beam;
  c->bounces = bounces; c->mirror_mask = mirror_mask; c->spp = spp;
  return SVO_OK;
}

int svo_resize(svo_ctx *c, int width, int height) {
  if (!c || width <= 0 || height <= 0) return fail(c, SVO_E_INVALID, "svo_resize: bad size");
  HIPCHK(c, hipSetDevice(c->device));
  if (width == c->width && height == c->height && c->own_color) return SVO_OK;
  HIPCHK(c, hipDeviceSynchronize());   // frames in flight on other streams still write the old images
  free_outputs(c);
  const size_t n = (size_t)width * (size_t)height;
  HIPCHK(c, hipMalloc((void **)&c->own_color, n * 4));
  HIPCHK(c, hipMalloc((void **)&c->own_depth, n * 4));
  HIPCHK(c, hipMalloc((void **)&c->own_hits, n * 16));
  HIPCHK(c, hipMemset(c->own_color, 0, n * 4));
  HIPCHK(c, hipMemset(c->own_depth, 0, n * 4));
  HIPCHK(c, hipMemset(c->own_hits, 0, n * 16));
  // hipMemset is asynchronous and the dispatch streams do not synchronise with the null stream: without this wait the
  // zeroing can land after the first frame's stores (seen as all-zero hit records of sky pixels)
  HIPCHK(c, hipDeviceSynchronize());
  if (!c->external_outputs) { c->d_color = c->own_color; c->d_depth = c->own_depth; c->d_hits = c->own_hits; }
  c->width = width; c->height = height;
  if (c->pick_default) { c->pick_x = width / 2; c->pick_y = height / 2; }   // the crosshair: Main.java:139-141 reads (WIDTH / 2, HEIGHT / 2)
  else if (c->pick_x >= width || c->pick_y >= height) { c->pick_x = -1; c->pick_y = -1; }
  c->y0 = 0; c->y1 = height; c->rows_set = false;  // a new image size resets the row band to the whole frame
  c->row_step = 1; c->out_y0 = 0; c->n_tile_rows = -1;
  return SVO_OK;
}

int svo_bind_outputs(svo_ctx *c, void *color, void *depth, void *hits) {
  if (!c) return SVO_E_INVALID;
  c->pick_live = false;
  if (!color) {
    c->external_outputs = false;
    const int k = (c->cur_set != 0 && c->alt_color[c->cur_set]) ? c->cur_set : 0;
    c->d_color = k ? c->alt_color[k] : c->own_color; c->d_depth = k ? c->alt_depth[k] : c->own_depth; c->d_hits = k ? c->alt_hits[k] : c->own_hits;
    return SVO_OK;
  }
  if (!depth) return fail(c, SVO_E_INVALID, "svo_bind_outputs: depth buffer required");
  c->external_outputs = true;
  c->d_color = (uint32_t *)color; c->d_depth = (float *)depth; c->d_hits = (uint4 *)hits;
  return SVO_OK;
}

int svo_set_rows(svo_ctx *c, int y0, int y1) {
  if (!c) return SVO_E_INVALID;
  if (y0 < 0 || y1 < y0 || (y0 & 7)) return fail(c, SVO_E_INVALID, "svo_set_rows: need 0 <= y0 <= y1, y0 % 8 == 0");
  c->y0 = y0; c->y1 = y1; c->rows_set = true;
  c->row_step = 1; c->out_y0 = y0; c->n_tile_rows = -1;
  return SVO_OK;
}

int svo_set_stripes(svo_ctx *c, int first_tile_row, int tile_row_step, int n_tile_rows, int out_row0) {
  if (!c) return SVO_E_INVALID;
  if (first_tile_row < 0 || tile_row_step < 1 || n_tile_rows < 0 || out_row0 < 0)
    return fail(c, SVO_E_INVALID, "svo_set_stripes: bad values");
  c->y0 = first_tile_row * 8; c->y1 = 0x7fffffff; c->rows_set = true;
  c->row_step = tile_row_step; c->out_y0 = out_row0; c->n_tile_rows = n_tile_rows;
  return SVO_OK;
}

int svo_set_pipeline(svo_ctx *c, int pipeline) {
  if (!c || pipeline < 0 || pipeline > 2) return fail(c, SVO_E_INVALID, "pipeline must be 0, 1 or 2");
#if !SVO_VARIANTS
  if (pipeline == 2)
    return fail(c, SVO_E_INVALID, "pipeline 2 (staged wavefront tracing, a comparator) is built into libsvohip_variants.so, not into this library");
#endif
  c->pipeline = pipeline;
  return SVO_OK;
}

int svo_set_tuning(svo_ctx *c, int waves_per_cu, int round_threshold_sixteenths) {
  if (!c || waves_per_cu < 0 || round_threshold_sixteenths < 0 || round_threshold_sixteenths > 15)
    return fail(c, SVO_E_INVALID, "svo_set_tuning: bad values");
  c->pb.waves_per_cu = waves_per_cu;
  c->pb.thresh_num = round_threshold_sixteenths;   // 0 = the running kernel's own default
#if SVO_VARIANTS
  c->wf.waves_per_cu = waves_per_cu;
  c->wf.thresh_num = round_threshold_sixteenths ? round_threshold_sixteenths : 12;
#endif
  return SVO_OK;
}

int svo_launch_info(svo_ctx *c, int *waves, int *waves_per_cu, int *round_threshold_sixteenths) {
  if (!c) return SVO_E_INVALID;
  if (waves) *waves = c->pb.last_blocks;
  if (waves_per_cu) *waves_per_cu = c->pb.last_per_cu;
  if (round_threshold_sixteenths) *round_threshold_sixteenths = c->pb.last_thresh;
  return SVO_OK;
}

static int ensure_derived(svo_ctx *c);

int svo_set_derived(svo_ctx *c, int mode) {
  if (!c || mode < 0 || mode > 1) return fail(c, SVO_E_INVALID, "svo_set_derived: 0 (records) or 1 (descriptor table)");
  c->derived_mode = mode;
  return SVO_OK;
}

int svo_derived_info(svo_ctx *c, uint64_t *descriptors, uint64_t *bytes, int *walkable, float *build_ms) {
  if (!c) return SVO_E_INVALID;
  if (!c->d_pool || c->pool_len < 7) return fail(c, SVO_E_NOPOOL, "svo_derived_info: no pool uploaded");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = ensure_derived(c);
  if (rc) return rc;
  if (descriptors) *descriptors = c->dt.count;
  if (bytes) *bytes = (uint64_t)c->dt.cap * 2 * sizeof(uint2);
  if (walkable) *walkable = c->dt.ok ? 1 : 0;
  if (build_ms) *build_ms = c->dt.build_ms;
  return SVO_OK;
}

int svo_derived_refresh_info(svo_ctx *c, uint64_t *refreshes, uint64_t *states, uint64_t *added, float *gpu_ms) {
  if (!c) return SVO_E_INVALID;
  if (refreshes) *refreshes = c->dt.refreshes;
  if (states) *states = c->dt.refresh_states;
  if (added) *added = c->dt.refresh_added;
  if (gpu_ms) *gpu_ms = c->dt.refresh_ms;
  return SVO_OK;
}

int svo_set_batch(svo_ctx *c, int nframes, uint64_t frame_stride) {
  if (!c || nframes < 1 || nframes > 64) return fail(c, SVO_E_INVALID, "svo_set_batch: 1..64 frames");
  if (nframes > 1 && frame_stride == 0) return fail(c, SVO_E_INVALID, "svo_set_batch: frame_stride must cover one frame's outputs");
  if (frame_stride * (uint64_t)nframes >= (1ull << 32)) return fail(c, SVO_E_INVALID, "svo_set_batch: batch too large for 32-bit output indices");
  c->batch = nframes;
  c->frame_stride = frame_stride;
  return SVO_OK;
}

int svo_set_progressive(svo_ctx *c, int enabled) {
  if (!c) return SVO_E_INVALID;
  c->progressive = enabled ? 1 : 0;
  return SVO_OK;
}

int svo_set_sequence(svo_ctx *c, int nframes, int fresh) {
  if (!c || nframes < 1 || nframes > 4096) return fail(c, SVO_E_INVALID, "svo_set_sequence: 1..4096 frames");
  c->seq = nframes;
  c->seq_fresh = fresh ? 1 : 0;
  return SVO_OK;
}

int svo_set_hit_records(svo_ctx *c, int enabled) {
  if (!c) return SVO_E_INVALID;
  c->write_hits = enabled ? 1 : 0;
  return SVO_OK;
}

int svo_set_stream(svo_ctx *c, void *hip_stream) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  // no synchronisation here: a caller may alternate streams to keep two frames in flight
  // (different output buffers); ordering between streams is the caller's business.  The library's own stream lives as
  // long as the context: NULL returns to it.
  if (hip_stream) {
    // leaving the library's own streams: what they still have in flight writes the library's images, which dispatches on the
    // caller's stream may render into next
    if (c->alt_inflight || c->is_own_stream(c->stream)) {
      HIPCHK(c, hipStreamSynchronize(c->own_stream));
      for (int k = 1; k < svo_ctx::kMaxSets; k++) if (c->alt_stream[k]) HIPCHK(c, hipStreamSynchronize(c->alt_stream[k]));
      c->alt_inflight = false;
    }
    c->stream = (hipStream_t)hip_stream;
  } else {
    c->stream = (c->cur_set != 0 && c->alt_stream[c->cur_set]) ? c->alt_stream[c->cur_set] : c->own_stream;   // the current set's own
  }
  c->pick_live = false;
  return SVO_OK;
}

int svo_set_overlap(svo_ctx *c, int sets) {
  if (!c) return SVO_E_INVALID;
  // 0: one set (no alternation); 1: the default number of sets; 2 .. 4: that many
  if (sets < 0 || sets > svo_ctx::kMaxSets) return fail(c, SVO_E_INVALID, "svo_set_overlap: 0 (off), 1 (the default number of sets) or 2 .. 8 sets");
  c->overlap = sets == 0 ? 1 : (sets == 1 ? 4 : sets);
  return SVO_OK;
}

int svo_pick_info(svo_ctx *c, int *x, int *y, uint64_t *from_mail, uint64_t *waited) {
  if (!c) return SVO_E_INVALID;
  if (x) *x = c->pick_x;
  if (y) *y = c->pick_y;
  if (from_mail) *from_mail = c->pick_from_mail;
  if (waited) *waited = c->pick_waited;
  return SVO_OK;
}

int svo_set_pick(svo_ctx *c, int x, int y) {
  if (!c) return SVO_E_INVALID;
  if (x < 0 || y < 0) { c->pick_x = -1; c->pick_y = -1; c->pick_default = false; return SVO_OK; }   // no pick: read-backs wait
  if (c->width > 0 && (x >= c->width || y >= c->height)) return fail(c, SVO_E_INVALID, "svo_set_pick: outside the image");
  c->pick_x = x; c->pick_y = y; c->pick_default = false;
  return SVO_OK;
}

// ---------------------------------------------------------------- dispatch
static int make_frame(svo_ctx *c, Frame &f) {
  if (!c->d_pool || c->pool_len < 7) return fail(c, SVO_E_NOPOOL, "svo_dispatch: no pool uploaded");
  if (!c->d_color) return fail(c, SVO_E_INVALID, "svo_dispatch: svo_resize not called");
  memcpy(f.cam, c->cam, sizeof f.cam);
  f.width = c->width; f.height = c->height;
  f.y0 = std::min(c->y0, c->height);
  f.y1 = std::min(c->y1, c->height);
  f.row_step = c->row_step;
  f.out_y0 = c->n_tile_rows < 0 ? f.y0 : c->out_y0;
  f.frame_number = c->frame_number; f.render_mode = c->render_mode;
  f.bounces = c->bounces; f.spp = c->spp; f.mirror_mask = c->mirror_mask;
  f.pool_len = (uint32_t)c->pool_len;
  f.dword0 = c->dword0;
  f.tiles_x = (c->width + 7) / 8;
  f.tiles_y = c->n_tile_rows < 0 ? (f.y1 - f.y0 + 7) / 8 : c->n_tile_rows;
  f.ntiles = f.tiles_x * f.tiles_y;
  f.write_hits = (c->write_hits && c->d_hits) ? 1 : 0;
  f.use_beam = 0; f.beam_w = 0; f.beam = nullptr;
  f.progressive = c->progressive;
  f.batch = 1; f.frame_stride = 0; f.seq = 1;
  if (!c->external_outputs && c->n_tile_rows > 0) {
    // packed stripes land at output rows out_y0 + 8 j + ly: they must stay inside the library's W x H images
    // (caller-owned gather buffers are the caller's to size, see svo_bind_outputs)
    int last = -1;
    for (int j = c->n_tile_rows - 1; j >= 0; j--)
      if ((long long)f.y0 + (long long)j * 8 * f.row_step < (long long)f.height) { last = j; break; }
    if (last >= 0) {
      const int gy = f.y0 + last * 8 * f.row_step;
      const int rows = std::min(8, f.height - gy);
      if (f.out_y0 + last * 8 + rows > f.height)
        return fail(c, SVO_E_INVALID, "svo_set_stripes: packed stripes do not fit the library-owned images; bind "
                                      "caller-owned outputs that cover out_row0 + 8 * n_tile_rows rows");
    }
  }
  return SVO_OK;
}

// elements per output plane that a launch of `f` may index
static size_t out_elems(const svo_ctx *c, const Frame &f) {
  return std::max((size_t)c->width * (size_t)c->height, (size_t)(f.out_y0 + f.tiles_y * 8) * (size_t)c->width);
}

// the coarse pass of useBeamOptimization (Main.java:257-266), enqueued in front of the frame's trace kernels
static int launch_beam(svo_ctx *c, Frame &f, int &set) {
  const int bw = (f.width + kBeamBlock - 1) / kBeamBlock, bh = (f.height + kBeamBlock - 1) / kBeamBlock;
  const size_t need = (size_t)bw * (size_t)bh;
  if (c->beam_cap < need) {
    HIPCHK(c, hipDeviceSynchronize());
    for (int i = 0; i < svo_ctx::kBeamSets; i++) {
      if (c->d_beam[i]) (void)hipFree(c->d_beam[i]);
      c->d_beam[i] = nullptr;
      c->beam_used[i] = false;
      HIPCHK(c, hipMalloc((void **)&c->d_beam[i], need * sizeof(float)));
      if (!c->beam_done[i]) HIPCHK(c, hipEventCreateWithFlags(&c->beam_done[i], hipEventDisableTiming));
    }
    c->beam_cap = need;
  }
  if (!c->beam_live_valid) {
    // once per pool: mark the live nodes of the top levels, bottom-up.  The pool only changes while the device is idle
    // (the mutators synchronise it), so no frame in flight reads the table now; the marking is complete before this call
    // returns, whichever stream the next dispatch uses.
    if (!c->d_beam_live) HIPCHK(c, hipMalloc((void **)&c->d_beam_live, kBeamLiveBytes));
    for (int d = kBeamTop; d >= 1; d--) {
      const unsigned n = 1u << (3 * d);
      hipLaunchKernelGGL(beam_live_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_pool, (uint32_t)c->pool_len, d,
                         c->d_beam_live);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->beam_live_valid = true;
  }
  set = (int)(c->beam_frames++ % svo_ctx::kBeamSets);
  if (c->beam_used[set]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->beam_done[set], 0));
  BeamArgs a;
  a.pool = c->d_pool; a.live = c->d_beam_live; a.f = f; a.beam = c->d_beam[set]; a.beam_w = bw; a.beam_h = bh;
  a.cam_ok = beam_camera_ok(f.cam);
  hipLaunchKernelGGL(beam_kernel, dim3((unsigned)((bw + 7) / 8), (unsigned)(f.tiles_y * 2)), dim3(64), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  f.use_beam = 1; f.beam_w = bw; f.beam = c->d_beam[set];
  return SVO_OK;
}

static int launch_frame_kernels(svo_ctx *c, Frame &f, bool count, uint32_t *color, float *depth, uint4 *hits);

// The descriptor table follows the pool: (re)built at the first dispatch after a pool change.  Every pool mutator has
// synchronised the device, except a caller writing through svo_pool_device_ptr -- so wait for the device here too: no
// frame in flight may still read the table that is replaced.
static int ensure_derived(svo_ctx *c) {
  if (c->derived_valid) return SVO_OK;
  HIPCHK(c, hipDeviceSynchronize());
  hipError_t e = derive::build_table(c->dt, c->d_pool, c->pool_len, c->own_stream);
  if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string("descriptor table: ") + hipGetErrorString(e));
  c->derived_valid = true;
  return SVO_OK;
}

static int launch_frame(svo_ctx *c, bool count) {
  Frame f;
  int rc = make_frame(c, f);
  if (rc) return rc;
  if (f.ntiles <= 0) return SVO_OK;
  int beam_set = -1;
  if (c->use_beam) {
    rc = launch_beam(c, f, beam_set);
    if (rc) return rc;
  }
  const int nb = count ? 1 : c->batch;
  if (nb > 1 && !c->external_outputs)
    return fail(c, SVO_E_INVALID, "svo_set_batch: a batch renders into caller-owned outputs (svo_bind_outputs)");
  if (nb > 1) {
    // frame k lands frame_stride elements behind frame k - 1: the stride must cover the rows one frame writes, or the
    // frames of the batch overwrite one another
    int last_row = 0;
    for (int j = f.tiles_y - 1; j >= 0; j--) {
      const long long gy = (long long)f.y0 + (long long)j * 8 * f.row_step;
      if (gy < (long long)std::min(f.height, f.y1)) { last_row = f.out_y0 + j * 8 + (int)std::min<long long>(8, std::min(f.height, f.y1) - gy); break; }
    }
    if (c->frame_stride < (uint64_t)last_row * (uint64_t)f.width)
      return fail(c, SVO_E_INVALID, "svo_set_batch: frame_stride is smaller than the rows one frame writes");
  }
  if (nb > 1 && f.progressive)
    return fail(c, SVO_E_INVALID, "svo_set_batch: cross-frame accumulation blends into ONE image frame after frame; "
                                  "a batch writes every frame to its own");
  if (nb > 1 && c->batch_cams && c->use_beam)
    return fail(c, SVO_E_INVALID, "svo_ring_submit_cams: the beam pre-pass belongs to one camera; submit such frames one by one");
  if (f.progressive && (c->seq > 1 || c->seq_fresh) && !count) {
    // a progressive sequence: c->seq frames of the accumulation, frameNumber, frameNumber + 1, ..., into the one image
    if (c->seq_fresh) {   // ... starting on a zeroed image (the application's first frames; the image after glTexStorage2D)
      int last_row = f.out_y0;
      for (int j = f.tiles_y - 1; j >= 0; j--) {
        const long long gy = (long long)f.y0 + (long long)j * 8 * f.row_step;
        if (gy < (long long)std::min(f.height, f.y1)) { last_row = f.out_y0 + j * 8 + (int)std::min<long long>(8, std::min(f.height, f.y1) - gy); break; }
      }
      if (last_row > f.out_y0)
        HIPCHK(c, hipMemsetAsync(c->d_color + (size_t)f.out_y0 * (size_t)f.width, 0, (size_t)(last_row - f.out_y0) * (size_t)f.width * 4, c->stream));
    }
    Frame probe = f;
    if (c->pipeline == 1 && f.spp <= 1 && persist_can_fold(probe, c->seq)) {
      // the persistent pipeline carries the whole sequence in one launch and blends it in frame order afterwards
      f.seq = c->seq;
      rc = launch_frame_kernels(c, f, count, c->d_color, c->d_depth, c->d_hits);
    } else {
      for (int k = 0; k < c->seq && rc == SVO_OK; k++) {
        Frame g = f;
        g.frame_number = f.frame_number + k;
        rc = launch_frame_kernels(c, g, count, c->d_color, c->d_depth, c->d_hits);
      }
    }
  } else if (nb > 1 && c->pipeline == 1) {
    // the persistent pipeline takes the whole batch as one launch: its waves go from frame to frame without a tail
    f.batch = nb;
    f.frame_stride = (uint32_t)c->frame_stride;
    rc = launch_frame_kernels(c, f, count, c->d_color, c->d_depth, c->d_hits);
  } else {
    for (int k = 0; k < nb && rc == SVO_OK; k++) {
      Frame g = f;
      g.frame_number = f.frame_number + k;
      if (c->batch_cams) { memcpy(g.cam, c->batch_cams[k].cam, sizeof g.cam); g.frame_number = c->batch_cams[k].frame_number; }
      const size_t o = (size_t)k * (size_t)c->frame_stride;
      rc = launch_frame_kernels(c, g, count, c->d_color + o, c->d_depth + o, c->d_hits ? c->d_hits + o : nullptr);
    }
  }
  if (rc == SVO_OK && beam_set >= 0) {
    HIPCHK(c, hipEventRecord(c->beam_done[beam_set], c->stream));
    c->beam_used[beam_set] = true;
  }
  return rc;
}

static int ring_copy_cams(void *ctx);

static int launch_frame_kernels(svo_ctx *c, Frame &f, bool count, uint32_t *color, float *depth, uint4 *hits) {
  int rc = SVO_OK;
  if (count) HIPCHK(c, hipMemsetAsync(c->d_counters, 0, sizeof(DeviceCounters), c->stream));
  if (c->pipeline == 1 && !count) {
#if SVO_VARIANTS
    static const int env_mode = getenv("SVO_DERIVED") ? atoi(getenv("SVO_DERIVED")) : -1;   // A/B override
#else
    const int env_mode = -1;
#endif
    const int mode = env_mode >= 0 ? env_mode : c->derived_mode;
    if (mode != 0 && (rc = ensure_derived(c)) != SVO_OK) return rc;
    const bool walk_table = mode != 0 && c->dt.ok;   // not derivable (deeper than 13 levels, cyclic): the records are walked
    rc = persist_launch(c->pb, c->d_pool, f, color, depth, hits, out_elems(c, f), c->stream, walk_table ? c->dt.desc : nullptr,
                        walk_table ? c->dt.aux : nullptr, walk_table ? c->dt.count : 0u, c->batch_cams ? c->batch_cams_dev : nullptr,
                        c->batch_cams, ring_copy_cams, c);
    if (rc) return fail(c, SVO_E_HIP, std::string("persistent pipeline: ") + hipGetErrorString((hipError_t)rc));
    return SVO_OK;
  }
#if SVO_VARIANTS
  if (c->pipeline == 2 && !count) {
    rc = wavefront_launch(c->wf, c->d_pool, f, color, depth, hits, out_elems(c, f), c->stream);
    if (rc) return fail(c, SVO_E_HIP, std::string("wavefront pipeline: ") + hipGetErrorString((hipError_t)rc));
    return SVO_OK;
  }
#endif
  const int per = (f.ntiles + 7) / 8;
  dim3 grid((unsigned)(per * 8)), block(64);
  if (count)
    hipLaunchKernelGGL(trace_fused_kernel<true>, grid, block, 0, c->stream, c->d_pool, f, color, depth, hits, c->d_counters);
  else
    hipLaunchKernelGGL(trace_fused_kernel<false>, grid, block, 0, c->stream, c->d_pool, f, color, depth, hits, c->d_counters);
  HIPCHK(c, hipGetLastError());
  return SVO_OK;
}

// The frame's pick: a launch of ONE wave (pick_kernel, svo_fused.hip.h) on a high-priority stream of its own, in front of the
// frame's kernels; a mail slot of its own per dispatch (kPickSlots of them, round-robin).  Frames the live shader's store
// applies to: one sample per pixel, no cross-frame accumulation, no batch, no beam pre-pass (its start distances change the
// iteration count of the hit record), the whole frame (a rank's stripes / a row band may not even contain the pixel).
static int launch_pick(svo_ctx *c) {
  c->pick_live = false;
  if (c->pick_x < 0 || c->batch != 1 || c->progressive || c->spp != 1 || c->use_beam || c->rows_set || c->n_tile_rows >= 0) return SVO_OK;
  Frame f;
  int rc = make_frame(c, f);
  if (rc) return rc;
  if (f.ntiles <= 0 || c->pick_x >= f.width || c->pick_y >= f.height) return SVO_OK;
  // (the pick is a convenience: a runtime that cannot give it its mail or its stream leaves every read-back on the waiting path)
  if (!c->pick_mail) {
    if (hipHostMalloc((void **)&c->pick_mail, (size_t)kPickSlots * kPickWords * sizeof(uint32_t), hipHostMallocCoherent) != hipSuccess) {
      c->pick_mail = nullptr;
      (void)hipGetLastError();
      return SVO_OK;
    }
    memset(c->pick_mail, 0, (size_t)kPickSlots * kPickWords * sizeof(uint32_t));
  }
  if (!c->pick_stream) {
    int lo = 0, hi = 0;   // (hi = the greatest priority = the numerically lowest value)
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) hi = 0;
    if (hipStreamCreateWithPriority(&c->pick_stream, hipStreamNonBlocking, hi) != hipSuccess &&
        hipStreamCreateWithFlags(&c->pick_stream, hipStreamNonBlocking) != hipSuccess) {
      c->pick_stream = nullptr;
      (void)hipGetLastError();
      return SVO_OK;
    }
  }
  const uint32_t seq = c->pick_seq + 1u == 0u ? 1u : c->pick_seq + 1u;   // (0 = "nothing written yet")
  uint32_t *mail = c->pick_mail + (size_t)(seq % (uint32_t)kPickSlots) * kPickWords;
  hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(64), 0, c->pick_stream, c->d_pool, f, c->pick_x, c->pick_y, mail, seq);
  HIPCHK(c, hipGetLastError());
  c->pick_seq = seq;
  c->pick_live = true; c->pick_live_x = c->pick_x; c->pick_live_y = c->pick_y;
  return SVO_OK;
}

int svo_dispatch_async(svo_ctx *c) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  // Two sets {stream, images} in turn while the library owns both and nothing ties a frame to the image before it: the next
  // frame's persistent waves take the CUs this frame's tail frees (glDispatchCompute only enqueues as well: Renderer.java:118-121).
  const bool own = !c->external_outputs && c->own_color && c->is_own_stream(c->stream);
  if (c->overlap > 1 && own && c->pipeline == 1 && !c->progressive && c->batch == 1) {
    const int k = (c->cur_set + 1) % c->overlap;
    if (k != 0) {
      if (!c->alt_stream[k]) HIPCHK(c, hipStreamCreateWithFlags(&c->alt_stream[k], hipStreamNonBlocking));
      if (!c->alt_color[k]) {
        const size_t n = (size_t)c->width * (size_t)c->height;
        HIPCHK(c, hipMalloc((void **)&c->alt_color[k], n * 4));
        HIPCHK(c, hipMalloc((void **)&c->alt_depth[k], n * 4));
        HIPCHK(c, hipMalloc((void **)&c->alt_hits[k], n * 16));
        HIPCHK(c, hipMemsetAsync(c->alt_color[k], 0, n * 4, c->alt_stream[k]));
        HIPCHK(c, hipMemsetAsync(c->alt_depth[k], 0, n * 4, c->alt_stream[k]));
        HIPCHK(c, hipMemsetAsync(c->alt_hits[k], 0, n * 16, c->alt_stream[k]));
      }
    }
    // the host runs ahead of the GPU by at most `overlap` frames (a swap chain's depth): the set's previous frame must be complete
    // (the pick launches do not throttle it any more: they need a wave slot, not the frame's turn)
    if (!c->set_done[k]) HIPCHK(c, hipEventCreateWithFlags(&c->set_done[k], hipEventDisableTiming));
    if (c->set_used[k]) HIPCHK(c, hipEventSynchronize(c->set_done[k]));
    c->cur_set = k;
    c->stream = c->set_stream(k);
    c->d_color = k ? c->alt_color[k] : c->own_color; c->d_depth = k ? c->alt_depth[k] : c->own_depth; c->d_hits = k ? c->alt_hits[k] : c->own_hits;
    c->alt_inflight = true;
    c->pb.in_overlap = true;   // the launch shape of overlapping one-frame launches unless svo_set_tuning named one
    c->pb.overlap_sets = c->overlap;
  }
  int rc = launch_pick(c);
  if (rc == SVO_OK) {
    rc = launch_frame(c, false);
    if (rc) c->pick_live = false;
  }
  const bool overlapped = c->pb.in_overlap;
  c->pb.in_overlap = false;
  if (overlapped && rc == SVO_OK) {
    HIPCHK(c, hipEventRecord(c->set_done[c->cur_set], c->stream));
    c->set_used[c->cur_set] = true;
  }
  return rc;
}

int svo_sync(svo_ctx *c) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->alt_inflight) {   // the other sets' frames too
    HIPCHK(c, hipStreamSynchronize(c->own_stream));
    for (int k = 1; k < svo_ctx::kMaxSets; k++) if (c->alt_stream[k]) HIPCHK(c, hipStreamSynchronize(c->alt_stream[k]));
    c->alt_inflight = false;
  }
  return SVO_OK;
}

int svo_dispatch(svo_ctx *c) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  int rc = launch_pick(c);
  if (rc) return rc;
  rc = launch_frame(c, false);
  if (rc) { c->pick_live = false; return rc; }
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->stats.last_dispatch_ms = ms;
  return SVO_OK;
}

int svo_count_frame(svo_ctx *c, svo_stats *out) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  c->pick_live = false;   // (the counting pass renders into the current images)
  int rc = launch_frame(c, true);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  DeviceCounters h;
  HIPCHK(c, hipMemcpy(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost));
  c->stats.pixels = h.pixels; c->stats.rays = h.rays; c->stats.nan_rays = h.nan_rays;
  c->stats.iterations = h.iterations; c->stats.alg_bytes = h.alg_bytes; c->stats.max_iter = h.max_iter;
  if (out) *out = c->stats;
  return SVO_OK;
}

int svo_get_stats(svo_ctx *c, svo_stats *out) {
  if (!c || !out) return SVO_E_INVALID;
  *out = c->stats;
  return SVO_OK;
}

int svo_time_frames(svo_ctx *c, int warmup, int iters, float *ms) {
  if (!c || iters <= 0 || !ms) return fail(c, SVO_E_INVALID, "svo_time_frames: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  c->pick_live = false;
  std::vector<hipEvent_t> ev((size_t)iters + 1, nullptr);
  int rc = SVO_OK;
  hipError_t e = hipSuccess;
  const char *what = "";
  auto ok = [&](hipError_t r, const char *w) { if (r != hipSuccess && e == hipSuccess) { e = r; what = w; } return r == hipSuccess; };
  for (int i = 0; i < warmup && rc == SVO_OK; i++) rc = launch_frame(c, false);
  for (size_t i = 0; i < ev.size() && rc == SVO_OK && e == hipSuccess; i++) ok(hipEventCreate(&ev[i]), "hipEventCreate");
  if (rc == SVO_OK && e == hipSuccess) ok(hipEventRecord(ev[0], c->stream), "hipEventRecord");
  for (int i = 0; i < iters && rc == SVO_OK && e == hipSuccess; i++) {
    rc = launch_frame(c, false);
    if (rc == SVO_OK) ok(hipEventRecord(ev[(size_t)i + 1], c->stream), "hipEventRecord");
  }
  (void)hipStreamSynchronize(c->stream);   // also on the error paths: the events must be idle before they go
  for (int i = 0; i < iters && rc == SVO_OK && e == hipSuccess; i++)
    ok(hipEventElapsedTime(&ms[i], ev[(size_t)i], ev[(size_t)i + 1]), "hipEventElapsedTime");
  for (auto &x : ev) if (x) (void)hipEventDestroy(x);
  if (rc != SVO_OK) return rc;
  if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
  c->stats.last_dispatch_ms = ms[iters - 1];
  return SVO_OK;
}


// ---------------------------------------------------------------- frames in flight behind the boundary
static int ring_create_impl(svo_ctx *c, int slots, int frames_per_slot, int want_hits, bool own_images);
int svo_ring_create(svo_ctx *c, int slots, int frames_per_slot, int want_hits) {
  return ring_create_impl(c, slots, frames_per_slot, want_hits, true);
}
// own_images = false: every slot will be bound to caller-owned buffers before its first submission (svo_group_*)
static int ring_create_impl(svo_ctx *c, int slots, int frames_per_slot, int want_hits, bool own_images) {
  if (!c || slots < 1 || slots > 8 || frames_per_slot < 1 || frames_per_slot > 64)
    return fail(c, SVO_E_INVALID, "svo_ring_create: 1..8 slots of 1..64 frames");
  if (c->width <= 0 || c->height <= 0) return fail(c, SVO_E_INVALID, "svo_ring_create: svo_resize not called");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  ring_free(c);
  const uint64_t stride = (uint64_t)c->width * (uint64_t)c->height;
  if (stride * (uint64_t)frames_per_slot >= (1ull << 32)) return fail(c, SVO_E_INVALID, "svo_ring_create: slot too large for 32-bit output indices");
  c->ring.resize((size_t)slots);
  const size_t n = (size_t)stride * (size_t)frames_per_slot;
  // CU mask of the slots' streams: bit i = CU i / 8 of XCD i % 8 (tools/cumask_probe.hip); the top 8 r bits cleared
  // keep r CUs of every XCD free of persistent waves
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
  std::vector<uint32_t> cumask((size_t)(ncu + 31) / 32, 0xffffffffu);
  const int reserve = std::min(c->reserved_cus * 8, ncu - 8);
  for (int b = ncu - reserve; b < ncu && reserve > 0; b++) cumask[(size_t)b >> 5] &= ~(1u << (b & 31));
  if (ncu & 31) cumask.back() &= (1u << (ncu & 31)) - 1u;
  c->pb.cus_reserved = reserve > 0 ? reserve : 0;
  for (auto &s : c->ring) {
    hipError_t e = reserve > 0 ? hipExtStreamCreateWithCUMask(&s.stream, (uint32_t)cumask.size(), cumask.data())
                               : hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&s.e0);
    if (e == hipSuccess) e = hipEventCreate(&s.e1);
    if (e == hipSuccess && own_images) e = hipMalloc((void **)&s.color, n * 4);
    if (e == hipSuccess && own_images) e = hipMalloc((void **)&s.depth, n * 4);
    if (e == hipSuccess && own_images && want_hits) e = hipMalloc((void **)&s.hits, n * 16);
    if (e == hipSuccess && own_images) e = hipMemset(s.color, 0, n * 4);
    if (e == hipSuccess && own_images) e = hipMemset(s.depth, 0, n * 4);
    if (e == hipSuccess && own_images && want_hits) e = hipMemset(s.hits, 0, n * 16);
    if (e != hipSuccess) {
      ring_free(c);
      return fail(c, SVO_E_HIP, std::string("svo_ring_create: ") + hipGetErrorString(e));
    }
  }
  HIPCHK(c, hipDeviceSynchronize());   // the zeroing above is asynchronous; the slots' streams do not wait for the null stream
  c->ring_frames = frames_per_slot;
  c->ring_stride = stride;
  return SVO_OK;
}

int svo_set_reserved_cus(svo_ctx *c, int per_xcd) {
  if (!c || per_xcd < 0 || per_xcd > 16) return fail(c, SVO_E_INVALID, "svo_set_reserved_cus: 0..16 CUs per XCD");
  c->reserved_cus = per_xcd;
  return SVO_OK;
}

int svo_ring_destroy(svo_ctx *c) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  ring_free(c);
  return SVO_OK;
}

static svo_ctx::RingSlot *ring_slot(svo_ctx *c, int slot, const char *who) {
  if (!c) return nullptr;
  if (slot < 0 || slot >= (int)c->ring.size()) {
    (void)fail(c, SVO_E_INVALID, std::string(who) + ": no such slot (svo_ring_create first)");
    return nullptr;
  }
  return &c->ring[(size_t)slot];
}

int svo_ring_bind_slot(svo_ctx *c, int slot, void *color, void *depth, void *hits, uint64_t frame_stride) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, "svo_ring_bind_slot");
  if (!s) return SVO_E_INVALID;
  if (color && (!depth || frame_stride == 0)) return fail(c, SVO_E_INVALID, "svo_ring_bind_slot: depth buffer and frame stride required");
  if (color && frame_stride * (uint64_t)c->ring_frames >= (1ull << 32)) return fail(c, SVO_E_INVALID, "svo_ring_bind_slot: slot too large for 32-bit output indices");
  s->xcolor = color; s->xdepth = color ? depth : nullptr; s->xhits = color ? hits : nullptr; s->xstride = color ? frame_stride : 0;
  return SVO_OK;
}

// The staged host-to-device copy of a submission's per-frame cameras, on the slot's stream, out of a pinned copy that is
// rewritten only once the copy that last read it has run.  Called by persist_launch for launches whose cameras do not travel in
// the table kernel's arguments.  Returns a hipError_t as int.
static int ring_copy_cams(void *ctx) {
  svo_ctx *c = (svo_ctx *)ctx;
  svo_ctx::RingSlot *s = (svo_ctx::RingSlot *)c->batch_cams_slot;
  if (!s || !c->batch_cams || !s->d_fvar) return (int)hipErrorInvalidValue;
  hipError_t e = hipEventSynchronize(s->fvar_copied);
  if (e != hipSuccess) return (int)e;
  memcpy(s->h_fvar, c->batch_cams, (size_t)c->batch * sizeof(FrameVar));
  e = hipMemcpyAsync(s->d_fvar, s->h_fvar, (size_t)c->batch * sizeof(FrameVar), hipMemcpyHostToDevice, s->stream);
  if (e == hipSuccess) e = hipEventRecord(s->fvar_copied, s->stream);
  return (int)e;
}

// `cams`: null = nframes consecutive frames of the context's camera starting at frame_number (svo_ring_submit); else nframes
// entries, every frame with its own camera and frameNumber (svo_ring_submit_cams)
static int ring_submit(svo_ctx *c, int frame_number, int nframes, const FrameVar *cams, int *slot, const char *who) {
  if (!c) return SVO_E_INVALID;
  if (c->ring.empty()) return fail(c, SVO_E_INVALID, std::string(who) + ": svo_ring_create first");
  if (nframes < 1 || nframes > c->ring_frames) return fail(c, SVO_E_INVALID, std::string(who) + ": 1..frames_per_slot frames");
  // the cross-frame accumulation blends with what the previous dispatch left in the SAME image (svotrace.comp:712-719); a
  // ring of several slots hands every submission another image, so a continued accumulation would blend with the frame of
  // `slots` submissions ago.  Sequences that start on a fresh image (svo_set_sequence(n, 1)) are whole in one slot.
  if (c->progressive && !c->seq_fresh && c->ring.size() > 1)
    return fail(c, SVO_E_INVALID, std::string(who) + ": a progressive accumulation that continues on the previous image needs a ring of "
                                  "ONE slot (or svo_dispatch); with several slots start every submission fresh: svo_set_sequence(n, 1)");
  HIPCHK(c, hipSetDevice(c->device));
  const int si = (int)(c->ring_next % (unsigned)c->ring.size());
  svo_ctx::RingSlot &s = c->ring[(size_t)si];
  if (!s.xcolor && !s.color) return fail(c, SVO_E_INVALID, std::string(who) + ": the slot has no images (bind it first)");
  // the dispatch state of the context, with this slot's stream, images and frame range swapped in
  struct Saved {
    hipStream_t stream; uint32_t *col; float *dep; uint4 *hit; bool ext; int batch; uint64_t stride; int frame; float cam[15];
  } sv = {c->stream, c->d_color, c->d_depth, c->d_hits, c->external_outputs, c->batch, c->frame_stride, c->frame_number, {}};
  memcpy(sv.cam, c->cam, sizeof sv.cam);
  c->stream = s.stream;
  c->external_outputs = true;
  if (s.xcolor) { c->d_color = (uint32_t *)s.xcolor; c->d_depth = (float *)s.xdepth; c->d_hits = (uint4 *)s.xhits; c->frame_stride = s.xstride; }
  else { c->d_color = s.color; c->d_depth = s.depth; c->d_hits = s.hits; c->frame_stride = c->ring_stride; }
  c->batch = nframes;
  c->frame_number = frame_number;
  int rc = SVO_OK;
  hipError_t e = hipSuccess;
  if (cams && nframes == 1) {           // one frame: simply this frame's camera (beam pre-pass and all)
    memcpy(c->cam, cams[0].cam, sizeof c->cam);
    c->frame_number = cams[0].frame_number;
  } else if (cams) {
    // the batch's cameras live in the slot's table on the device.  Launches with row / column tables carry them there inside the
    // table kernel's arguments (up to kCamPack frames); the others get the staged copy of ring_copy_cams, enqueued by the
    // launch itself in front of the first kernel that reads the table.
    if (!s.d_fvar) {
      e = hipMalloc((void **)&s.d_fvar, (size_t)c->ring_frames * sizeof(FrameVar));
      if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_fvar, (size_t)c->ring_frames * sizeof(FrameVar), hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s.fvar_copied, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(s.fvar_copied, s.stream);   // (so that the first wait below has something to wait for)
    }
    c->batch_cams = cams; c->batch_cams_dev = s.d_fvar; c->batch_cams_slot = &s;
    c->frame_number = cams[0].frame_number;
  }
  if (e == hipSuccess) e = hipEventRecord(s.e0, s.stream);
  // the launch shape a ring of several slots gets unless svo_set_tuning named one: kRingWavesPerCu persistent waves per CU
  c->pb.in_ring = c->ring.size() > 1;
  if (e == hipSuccess) rc = launch_frame(c, false);
  c->pb.in_ring = false;
  if (e == hipSuccess && rc == SVO_OK) e = hipEventRecord(s.e1, s.stream);
  if (e == hipSuccess && rc == SVO_OK && s.fwd_dst) {
    // the slot's frames travel to the frame owner behind the launch, on the same stream: a device-to-device copy (SDMA
    // between GPUs: no CU slot needed next to the persistent waves), then the submission's number into the owner's flag
    if (s.fwd_dst_device >= 0 && s.fwd_dst_device != c->device)
      e = hipMemcpyPeerAsync(s.fwd_dst, s.fwd_dst_device, s.fwd_src, c->device, s.fwd_bytes, s.stream);
    else
      e = hipMemcpyAsync(s.fwd_dst, s.fwd_src, s.fwd_bytes, hipMemcpyDeviceToDevice, s.stream);
    if (e == hipSuccess && s.fwd_flag) {
      e = hipMemsetD32Async((hipDeviceptr_t)s.seq_word, (int)(c->ring_next + 1u), 1, s.stream);
      if (e == hipSuccess) e = hipMemcpyAsync(s.fwd_flag, s.seq_word, 4, hipMemcpyDeviceToDevice, s.stream);
    }
    if (e == hipSuccess && s.e2) e = hipEventRecord(s.e2, s.stream);
  }
  c->stream = sv.stream; c->d_color = sv.col; c->d_depth = sv.dep; c->d_hits = sv.hit; c->external_outputs = sv.ext;
  c->batch = sv.batch; c->frame_stride = sv.stride; c->frame_number = sv.frame;
  memcpy(c->cam, sv.cam, sizeof c->cam);
  c->batch_cams = nullptr; c->batch_cams_dev = nullptr; c->batch_cams_slot = nullptr;
  if (e != hipSuccess) return fail(c, SVO_E_HIP, std::string(who) + ": " + hipGetErrorString(e));
  if (rc) return rc;
  s.first_frame = cams ? cams[0].frame_number : frame_number; s.nframes = nframes; s.used = true;
  c->ring_next++;
  if (slot) *slot = si;
  return SVO_OK;
}

int svo_ring_submit(svo_ctx *c, int frame_number, int nframes, int *slot) {
#if SVO_VARIANTS
  static const bool force_cams = getenv("SVO_FORCE_CAMS") && atoi(getenv("SVO_FORCE_CAMS")) != 0;   // A/B: the kCams kernel on a static camera
#else
  const bool force_cams = false;
#endif
  if (force_cams && c && nframes > 1 && nframes <= 64 && !c->use_beam && !c->progressive) {
    FrameVar v[64];
    for (int k = 0; k < nframes; k++) { memcpy(v[k].cam, c->cam, sizeof v[k].cam); v[k].frame_number = frame_number + k; }
    return ring_submit(c, frame_number, nframes, v, slot, "svo_ring_submit");
  }
  return ring_submit(c, frame_number, nframes, nullptr, slot, "svo_ring_submit");
}

int svo_ring_submit_cams(svo_ctx *c, int nframes, const float *cams, const int *frame_numbers, int *slot) {
  if (!c || !cams || !frame_numbers) return fail(c, SVO_E_INVALID, "svo_ring_submit_cams: null array");
  if (nframes < 1 || nframes > 64) return fail(c, SVO_E_INVALID, "svo_ring_submit_cams: 1..frames_per_slot frames");
  FrameVar v[64];
  for (int k = 0; k < nframes; k++) {
    memcpy(v[k].cam, cams + 15 * (size_t)k, sizeof v[k].cam);
    v[k].frame_number = frame_numbers[k];
  }
  return ring_submit(c, frame_numbers[0], nframes, v, slot, "svo_ring_submit_cams");
}

int svo_ring_wait(svo_ctx *c, int slot) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, "svo_ring_wait");
  if (!s) return SVO_E_INVALID;
  if (!s->used) return SVO_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(s->e1));
  return SVO_OK;
}

int svo_ring_query(svo_ctx *c, int slot, int *done, int *first_frame, int *nframes, float *gpu_ms) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, "svo_ring_query");
  if (!s) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  bool fin = true;
  if (s->used) {
    const hipError_t q = hipEventQuery(s->e1);
    if (q == hipErrorNotReady) fin = false;
    else if (q != hipSuccess) return fail(c, SVO_E_HIP, std::string("svo_ring_query: ") + hipGetErrorString(q));
  }
  if (done) *done = fin ? 1 : 0;
  if (first_frame) *first_frame = s->first_frame;
  if (nframes) *nframes = s->used ? s->nframes : 0;
  if (gpu_ms) {
    *gpu_ms = 0.0f;
    if (s->used && fin) HIPCHK(c, hipEventElapsedTime(gpu_ms, s->e0, s->e1));
  }
  return SVO_OK;
}

// frame k of a slot: base pointers of its three images
static int ring_frame(svo_ctx *c, int slot, int k, const char *who, const uint32_t **col, const float **dep, const uint4 **hit,
                      size_t *elems = nullptr) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, who);
  if (!s) return SVO_E_INVALID;
  if (!s->used || k < 0 || k >= s->nframes) return fail(c, SVO_E_INVALID, std::string(who) + ": the slot does not hold that frame");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(s->e1));
  const uint64_t stride = s->xcolor ? s->xstride : c->ring_stride;
  const uint64_t o = (uint64_t)k * stride;
  // a caller-owned slot may hold less than W x H elements per frame (packed stripes of one rank)
  if (elems) *elems = (size_t)std::min<uint64_t>((uint64_t)c->width * (uint64_t)c->height, stride);
  *col = (s->xcolor ? (const uint32_t *)s->xcolor : s->color) + o;
  *dep = (s->xcolor ? (const float *)s->xdepth : s->depth) + o;
  const uint4 *h = s->xcolor ? (const uint4 *)s->xhits : s->hits;
  *hit = h ? h + o : nullptr;
  return SVO_OK;
}

int svo_ring_read_color(svo_ctx *c, int slot, int k, void *rgba8) {
  const uint32_t *col; const float *dep; const uint4 *hit;
  if (!rgba8) return fail(c, SVO_E_INVALID, "readback: null buffer");
  size_t n = 0;
  int rc = ring_frame(c, slot, k, "svo_ring_read_color", &col, &dep, &hit, &n);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(rgba8, col, n * 4, hipMemcpyDeviceToHost));
  return SVO_OK;
}
int svo_ring_read_depth(svo_ctx *c, int slot, int k, float *depth) {
  const uint32_t *col; const float *dep; const uint4 *hit;
  if (!depth) return fail(c, SVO_E_INVALID, "readback: null buffer");
  size_t n = 0;
  int rc = ring_frame(c, slot, k, "svo_ring_read_depth", &col, &dep, &hit, &n);
  if (rc) return rc;
  HIPCHK(c, hipMemcpy(depth, dep, n * 4, hipMemcpyDeviceToHost));
  return SVO_OK;
}
int svo_ring_read_hits(svo_ctx *c, int slot, int k, svo_hit *hits) {
  const uint32_t *col; const float *dep; const uint4 *hit;
  if (!hits) return fail(c, SVO_E_INVALID, "readback: null buffer");
  size_t n = 0;
  int rc = ring_frame(c, slot, k, "svo_ring_read_hits", &col, &dep, &hit, &n);
  if (rc) return rc;
  if (!hit) return fail(c, SVO_E_INVALID, "svo_ring_read_hits: the ring was created without hit records");
  HIPCHK(c, hipMemcpy(hits, hit, n * 16, hipMemcpyDeviceToHost));
  return SVO_OK;
}
int svo_ring_read_pixel(svo_ctx *c, int slot, int k, int x, int y, void *rgba8, float *depth, svo_hit *hit) {
  const uint32_t *col; const float *dep; const uint4 *hp;
  if (!c) return SVO_E_INVALID;
  if (x < 0 || y < 0 || x >= c->width || y >= c->height) return fail(c, SVO_E_INVALID, "svo_ring_read_pixel: outside the image");
  int rc = ring_frame(c, slot, k, "svo_ring_read_pixel", &col, &dep, &hp);
  if (rc) return rc;
  const size_t o = (size_t)y * (size_t)c->width + (size_t)x;
  if (rgba8) HIPCHK(c, hipMemcpy(rgba8, col + o, 4, hipMemcpyDeviceToHost));
  if (depth) HIPCHK(c, hipMemcpy(depth, dep + o, 4, hipMemcpyDeviceToHost));
  if (hit) {
    if (!hp) return fail(c, SVO_E_INVALID, "svo_ring_read_pixel: the ring was created without hit records");
    HIPCHK(c, hipMemcpy(hit, hp + o, 16, hipMemcpyDeviceToHost));
  }
  return SVO_OK;
}

int svo_ring_forward_slot(svo_ctx *c, int slot, const void *src, void *dst, uint64_t nbytes, void *flag) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, "svo_ring_forward_slot");
  if (!s) return SVO_E_INVALID;
  if (dst && (!src || nbytes == 0)) return fail(c, SVO_E_INVALID, "svo_ring_forward_slot: source and size required");
  HIPCHK(c, hipSetDevice(c->device));
  if (dst && !s->seq_word) HIPCHK(c, hipMalloc((void **)&s->seq_word, 4));
  if (dst && !s->e2) HIPCHK(c, hipEventCreateWithFlags(&s->e2, hipEventDisableTiming));
  s->fwd_dst_device = -1;
  s->fwd_src = dst ? src : nullptr; s->fwd_dst = dst; s->fwd_bytes = dst ? nbytes : 0; s->fwd_flag = dst ? flag : nullptr;
  return SVO_OK;
}

// ---------------------------------------------------------------- device memory shared between the ranks of one node
int svo_dev_alloc(svo_ctx *c, uint64_t nbytes, void **dptr) {
  if (!c || !dptr || nbytes == 0) return fail(c, SVO_E_INVALID, "svo_dev_alloc: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(dptr, nbytes));
  c->dev_allocs.push_back(*dptr);
  HIPCHK(c, hipMemset(*dptr, 0, nbytes));
  HIPCHK(c, hipDeviceSynchronize());
  return SVO_OK;
}
int svo_dev_free(svo_ctx *c, void *dptr) {
  if (!c) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  for (auto &p : c->dev_allocs)
    if (p && p == dptr) { p = nullptr; HIPCHK(c, hipFree(dptr)); return SVO_OK; }
  return dptr ? fail(c, SVO_E_INVALID, "svo_dev_free: not allocated by this context") : SVO_OK;
}
int svo_dev_read(svo_ctx *c, const void *dptr, void *host, uint64_t nbytes) {
  if (!c || !dptr || !host) return fail(c, SVO_E_INVALID, "svo_dev_read: null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy(host, dptr, nbytes, hipMemcpyDeviceToHost));
  return SVO_OK;
}
int svo_ipc_export(svo_ctx *c, void *dptr, void *handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!c || !dptr || !handle64) return fail(c, SVO_E_INVALID, "svo_ipc_export: null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  HIPCHK(c, hipIpcGetMemHandle(&h, dptr));
  memcpy(handle64, &h, sizeof h);
  return SVO_OK;
}
int svo_ipc_open(svo_ctx *c, const void *handle64, void **dptr) {
  if (!c || !dptr || !handle64) return fail(c, SVO_E_INVALID, "svo_ipc_open: null pointer");
  HIPCHK(c, hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof h);
  HIPCHK(c, hipIpcOpenMemHandle(dptr, h, hipIpcMemLazyEnablePeerAccess));
  c->ipc_opened.push_back(*dptr);
  return SVO_OK;
}
int svo_ipc_close(svo_ctx *c, void *dptr) {
  if (!c || !dptr) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  for (auto &p : c->ipc_opened)
    if (p == dptr) { p = nullptr; HIPCHK(c, hipIpcCloseMemHandle(dptr)); return SVO_OK; }
  return fail(c, SVO_E_INVALID, "svo_ipc_close: not opened by this context");
}

int svo_ring_device_ptrs(svo_ctx *c, int slot, void **color, void **depth, void **hits, uint64_t *frame_stride, void **stream) {
  svo_ctx::RingSlot *s = ring_slot(c, slot, "svo_ring_device_ptrs");
  if (!s) return SVO_E_INVALID;
  if (color) *color = s->xcolor ? s->xcolor : (void *)s->color;
  if (depth) *depth = s->xcolor ? s->xdepth : (void *)s->depth;
  if (hits) *hits = s->xcolor ? s->xhits : (void *)s->hits;
  if (frame_stride) *frame_stride = s->xcolor ? s->xstride : c->ring_stride;
  if (stream) *stream = (void *)s->stream;
  return SVO_OK;
}

// ---------------------------------------------------------------- readback
static int read_rows(svo_ctx *c, void *dst, const void *src, size_t elem) {
  if (!dst) return fail(c, SVO_E_INVALID, "readback: null buffer");
  if (!src) return fail(c, SVO_E_INVALID, "readback before svo_resize");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(dst, src, (size_t)c->width * (size_t)c->height * elem, hipMemcpyDeviceToHost));
  return SVO_OK;
}
int svo_read_color(svo_ctx *c, void *rgba8) { return c ? read_rows(c, rgba8, c->d_color, 4) : SVO_E_INVALID; }
int svo_read_depth(svo_ctx *c, float *depth) { return c ? read_rows(c, depth, c->d_depth, 4) : SVO_E_INVALID; }
int svo_read_hits(svo_ctx *c, svo_hit *hits) { return c ? read_rows(c, hits, c->d_hits, 16) : SVO_E_INVALID; }

int svo_read_beam(svo_ctx *c, float *beam) {
  if (!c || !beam) return fail(c, SVO_E_INVALID, "svo_read_beam: null buffer");
  if (!c->beam_frames || !c->beam_cap) return fail(c, SVO_E_INVALID, "svo_read_beam: no frame was dispatched with use_beam");
  HIPCHK(c, hipSetDevice(c->device));
  const int set = (int)((c->beam_frames - 1) % svo_ctx::kBeamSets);
  // the frame that made this image may have run on another stream than the current one (frames in flight)
  if (c->beam_used[set]) HIPCHK(c, hipEventSynchronize(c->beam_done[set]));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)((c->width + kBeamBlock - 1) / kBeamBlock) * (size_t)((c->height + kBeamBlock - 1) / kBeamBlock);
  HIPCHK(c, hipMemcpy(beam, c->d_beam[set], std::min(n, c->beam_cap) * sizeof(float), hipMemcpyDeviceToHost));
  return SVO_OK;
}

int svo_read_pixel(svo_ctx *c, int x, int y, void *rgba8, float *depth, svo_hit *hit) {
  if (!c) return SVO_E_INVALID;
  if (x < 0 || y < 0 || x >= c->width || y >= c->height) return fail(c, SVO_E_INVALID, "svo_read_pixel: outside the image");
  if (!c->d_color) return fail(c, SVO_E_INVALID, "readback before svo_resize");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->pick_live && x == c->pick_live_x && y == c->pick_live_y && c->pick_mail && (!hit || (c->write_hits && c->d_hits))) {
    // the pick pixel of the last dispatch: its pick launch has written (or will write) the mail slot -- no wait for the frame,
    // no copy.  Should the pick stream run dry without the word, the waiting path answers.
    volatile uint32_t *m = c->pick_mail + (size_t)(c->pick_seq % (uint32_t)kPickSlots) * kPickWords;
    bool got = false;
    for (unsigned spin = 0;; spin++) {
      if (__atomic_load_n((const uint32_t *)m, __ATOMIC_ACQUIRE) == c->pick_seq) { got = true; break; }
      if ((spin & 255u) == 255u) {
        const hipError_t q = hipStreamQuery(c->pick_stream);
        if (q == hipSuccess) { got = __atomic_load_n((const uint32_t *)m, __ATOMIC_ACQUIRE) == c->pick_seq; break; }
        if (q != hipErrorNotReady) return fail(c, SVO_E_HIP, std::string("svo_read_pixel: ") + hipGetErrorString(q));
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (got) {
      c->pick_from_mail++;
      if (rgba8) { const uint32_t v = m[1]; memcpy(rgba8, &v, 4); }
      if (depth) { const uint32_t v = m[2]; memcpy(depth, &v, 4); }
      if (hit) { uint32_t h[4] = {m[4], m[5], m[6], m[7]}; memcpy(hit, h, 16); }
      return SVO_OK;
    }
  }
  c->pick_waited++;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t o = (size_t)y * (size_t)c->width + (size_t)x;
  if (rgba8) HIPCHK(c, hipMemcpy(rgba8, c->d_color + o, 4, hipMemcpyDeviceToHost));
  if (depth) HIPCHK(c, hipMemcpy(depth, c->d_depth + o, 4, hipMemcpyDeviceToHost));
  if (hit) {
    if (!c->d_hits) return fail(c, SVO_E_INVALID, "svo_read_pixel: no hit image bound");
    HIPCHK(c, hipMemcpy(hit, c->d_hits + o, 16, hipMemcpyDeviceToHost));
  }
  return SVO_OK;
}

int svo_output_device_ptrs(svo_ctx *c, void **color, void **depth, void **hits) {
  if (!c) return SVO_E_INVALID;
  if (color) *color = c->d_color;
  if (depth) *depth = c->d_depth;
  if (hits) *hits = c->d_hits;
  return SVO_OK;
}

}  // extern "C"
#include "svo_group.hip.h"
extern "C" {

#ifdef SVO_STAMPS
// diagnostic builds only: the diagnostics words of the persistent pipeline's last counter set (16 x u64, then the
// 64-word histograms of lanes traversing per trip and of lanes in the POP section, 8 x u64 of a round's parts): 704 bytes
int svo_debug_heads(svo_ctx *c, void *out) {
  if (!c || !out || !c->pb.heads) return SVO_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(out, c->pb.heads + (size_t)((c->pb.frames - 1) % kHeadSets) * kHeadWords + 8 * kHeadStride, 704, hipMemcpyDeviceToHost));
  return SVO_OK;
}
#endif

}  // extern "C"
