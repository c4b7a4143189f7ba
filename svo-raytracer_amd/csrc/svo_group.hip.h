// svo_group.hip.h -- N GPUs behind the C ABI (include/svo_hip.h, svo_group_*): one process, one host thread, n member
// contexts.  Included by svo_hip.hip (it uses the contexts' internals).
//
// The path shards by screen tile: member r renders tile rows r, r + n, r + 2n, ... of every frame
// (svo_set_stripes) packed into its chunk; the chunks of a submission travel to the frame owner (member 0) behind the
// launch -- a peer copy on the member's slot stream (SDMA over xGMI: no CU slot next to the persistent waves;
// svo_ring_forward_slot), or one RCCL send / receive pair per member inside a group call -- and the owner hands out whole
// frames in frame order.  The pool is replicated: one upload from the host, n - 1 peer copies.
//
// Owner's gather buffer of slot b:  [n members][planes][frames_per_slot][rows_per_member][W] 32-bit words, planes = colour,
// depth [+ 4 words of hit record, pixel-major]; member r > 0 renders into a chunk of the same shape on its own device,
// member 0 straight into chunk 0.  Same layout as the torch driver's (svo-raytracer_amd/framering.py).
#pragma once

#include <dlfcn.h>
// RCCL: types and prototypes only -- the library is dlopen()ed on demand, never linked.  From the installed header when
// there is one (every call below is then type-checked against it); a ROCm install without RCCL's headers still builds the
// library (the single-GPU path and the peer-copy exchange need none of this) from the declarations below, which state
// the seven point-to-point entry points as RCCL's public API documents them.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

struct svo_group {
  int n = 0;
  std::vector<svo_ctx *> m;
  std::vector<int> dev;
  int width = 0, height = 0, rpr = 0;        // rows_per_member: packed rows of one member's chunk of one frame
  int slots = 0, frames = 0, planes = 2;
  bool want_hits = false;
  int exchange = 0;                          // 0: peer copies, 1: RCCL send / receive
  uint64_t chunk_bytes = 0;
  std::vector<uint8_t *> gather;             // [slots], on the owner's device
  std::vector<std::vector<uint8_t *>> local; // [member][slot]: member r's chunk on its device (member 0: inside gather)
  std::vector<int> slot_first, slot_n;
  std::vector<char> slot_used;
  unsigned next = 0;
  // RCCL (loaded on demand: librccl is not a link-time dependency of the library)
  void *rccl = nullptr;
  std::vector<ncclComm_t> comm;              // one communicator per member
  std::vector<hipStream_t> recv_stream;      // owner: one receive stream per slot
  std::vector<hipEvent_t> recv_done;
  // (declared from rccl.h's own prototypes, so that every call below is type-checked against the installed header)
  decltype(&ncclCommInitAll) nccl_init_all = nullptr;
  decltype(&ncclCommDestroy) nccl_destroy = nullptr;
  decltype(&ncclSend) nccl_send = nullptr;
  decltype(&ncclRecv) nccl_recv = nullptr;
  decltype(&ncclGroupStart) nccl_group_start = nullptr;
  decltype(&ncclGroupEnd) nccl_group_end = nullptr;
  decltype(&ncclGetErrorString) nccl_error = nullptr;
  std::string err;
};

static int gfail(svo_group *g, int code, const std::string &msg) {
  if (g) g->err = msg;
  return code;
}
// a member's call failed: its message becomes the group's
static int gmember(svo_group *g, int r, int rc) {
  if (rc != SVO_OK) g->err = "member " + std::to_string(r) + " (device " + std::to_string(g->dev[(size_t)r]) + "): " + g->m[(size_t)r]->err;
  return rc;
}
#define GHIP(g, call)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) return gfail(g, SVO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GALL(g, expr)                                                \
  do {                                                               \
    for (int r_ = 0; r_ < (g)->n; r_++) {                            \
      svo_ctx *c = (g)->m[(size_t)r_];                               \
      const int rc_ = gmember(g, r_, (expr));                        \
      if (rc_) return rc_;                                           \
    }                                                                \
    return SVO_OK;                                                   \
  } while (0)

static void group_ring_free(svo_group *g) {
  for (int r = 0; r < g->n; r++) {
    svo_ctx *c = g->m[(size_t)r];
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    ring_free(c);
    if (r < (int)g->local.size())
      for (uint8_t *p : g->local[(size_t)r])
        if (p && r > 0) (void)svo_dev_free(c, p);
  }
  if (g->n > 0) {
    svo_ctx *o = g->m[0];
    (void)hipSetDevice(o->device);
    for (uint8_t *p : g->gather) if (p) (void)svo_dev_free(o, p);
    for (auto &s : g->recv_stream) if (s) (void)hipStreamDestroy(s);
    for (auto &e : g->recv_done) if (e) (void)hipEventDestroy(e);
  }
  g->gather.clear(); g->local.clear(); g->recv_stream.clear(); g->recv_done.clear();
  g->slot_first.clear(); g->slot_n.clear(); g->slot_used.clear();
  g->slots = 0; g->frames = 0; g->next = 0;
}

static int group_load_rccl(svo_group *g) {
  if (g->rccl) return SVO_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return gfail(g, SVO_E_INVALID, std::string("svo_group_ring_create: RCCL exchange asked for, but librccl cannot be loaded: ") + dlerror());
  g->nccl_init_all = (decltype(g->nccl_init_all))dlsym(h, "ncclCommInitAll");
  g->nccl_destroy = (decltype(g->nccl_destroy))dlsym(h, "ncclCommDestroy");
  g->nccl_send = (decltype(g->nccl_send))dlsym(h, "ncclSend");
  g->nccl_recv = (decltype(g->nccl_recv))dlsym(h, "ncclRecv");
  g->nccl_group_start = (decltype(g->nccl_group_start))dlsym(h, "ncclGroupStart");
  g->nccl_group_end = (decltype(g->nccl_group_end))dlsym(h, "ncclGroupEnd");
  g->nccl_error = (decltype(g->nccl_error))dlsym(h, "ncclGetErrorString");
  if (!g->nccl_init_all || !g->nccl_destroy || !g->nccl_send || !g->nccl_recv || !g->nccl_group_start || !g->nccl_group_end) {
    dlclose(h);
    return gfail(g, SVO_E_INVALID, "svo_group_ring_create: librccl lacks the point-to-point entry points");
  }
  g->rccl = h;
  return SVO_OK;
}
static int gnccl(svo_group *g, ncclResult_t rc, const char *what) {
  if (rc == ncclSuccess) return SVO_OK;
  return gfail(g, SVO_E_HIP, std::string(what) + ": " + (g->nccl_error ? g->nccl_error(rc) : "RCCL error"));
}

extern "C" {

int svo_group_create(const int *devices, int n, svo_group **out) {
  if (!out) return SVO_E_INVALID;
  *out = nullptr;
  if (!devices || n < 1 || n > 64) return SVO_E_INVALID;
  svo_group *g = new svo_group();
  for (int r = 0; r < n; r++) {
    svo_ctx *c = nullptr;
    const int rc = svo_create(devices[r], &c);
    if (rc != SVO_OK) {
      for (svo_ctx *x : g->m) (void)svo_destroy(x);
      delete g;
      return rc;
    }
    g->m.push_back(c);
    g->dev.push_back(devices[r]);
  }
  g->n = n;
  // peer access between the owner and every other device, both ways (pool replication reads the owner's pool, the tile
  // exchange writes the owner's gather buffer); where the hardware offers none, hipMemcpyPeer stages through the host
  for (int r = 1; r < n; r++) {
    if (devices[r] == devices[0]) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can) {
      (void)hipSetDevice(devices[r]);
      (void)hipDeviceEnablePeerAccess(devices[0], 0);
      (void)hipSetDevice(devices[0]);
      (void)hipDeviceEnablePeerAccess(devices[r], 0);
    }
    (void)hipGetLastError();   // "already enabled" is not an error
  }
  *out = g;
  return SVO_OK;
}

int svo_group_destroy(svo_group *g) {
  if (!g) return SVO_E_INVALID;
  group_ring_free(g);
  if (g->rccl) {
    for (ncclComm_t c : g->comm) if (c) (void)g->nccl_destroy(c);
    // (librccl stays loaded: unloading a library that owns GPU state at exit is not worth the risk)
  }
  for (svo_ctx *c : g->m) (void)svo_destroy(c);
  delete g;
  return SVO_OK;
}

const char *svo_group_last_error(const svo_group *g) { return g ? g->err.c_str() : "null group"; }
int svo_group_size(const svo_group *g) { return g ? g->n : 0; }
svo_ctx *svo_group_member(svo_group *g, int i) { return (g && i >= 0 && i < g->n) ? g->m[(size_t)i] : nullptr; }

// ---- pool: one copy from the host, n - 1 copies from the owner
static int group_replicate_pool(svo_group *g) {
  svo_ctx *o = g->m[0];
  for (int r = 1; r < g->n; r++) {
    svo_ctx *c = g->m[(size_t)r];
    int rc = gmember(g, r, svo_pool_reserve(c, o->pool_len));   // waits for the member's frames, zeroes its buffer
    if (rc) return rc;
    if (o->pool_len) {
      GHIP(g, hipSetDevice(c->device));
      if (c->device == o->device) GHIP(g, hipMemcpy(c->d_pool, o->d_pool, o->pool_len, hipMemcpyDeviceToDevice));
      else GHIP(g, hipMemcpyPeer(c->d_pool, c->device, o->d_pool, o->device, o->pool_len));
    }
    c->dword0 = o->dword0;
  }
  return SVO_OK;
}

int svo_group_pool_upload(svo_group *g, const void *host, uint64_t nbytes) {
  if (!g) return SVO_E_INVALID;
  int rc = gmember(g, 0, svo_pool_upload(g->m[0], host, nbytes));
  return rc ? rc : group_replicate_pool(g);
}
int svo_group_pool_update(svo_group *g, const void *host_base, uint64_t start, uint64_t end) {
  if (!g) return SVO_E_INVALID;
  GALL(g, svo_pool_update(c, host_base, start, end));   // two small byte ranges per brush stroke: from the host to every member
}
int svo_group_build_from_heightmap(svo_group *g, const uint16_t *height, const uint8_t *material, int n, uint64_t *out_nbytes) {
  if (!g) return SVO_E_INVALID;
  int rc = gmember(g, 0, svo_build_from_heightmap(g->m[0], height, material, n, out_nbytes));
  return rc ? rc : group_replicate_pool(g);
}
int svo_group_pool_download(svo_group *g, void *host, uint64_t nbytes) {
  return g ? gmember(g, 0, svo_pool_download(g->m[0], host, nbytes)) : SVO_E_INVALID;
}

// ---- frame state: the same on every member
int svo_group_set_camera(svo_group *g, const float pos[3], const float l1[3], const float l2[3], const float r1[3], const float r2[3]) {
  if (!g) return SVO_E_INVALID;
  GALL(g, svo_set_camera(c, pos, l1, l2, r1, r2));
}
int svo_group_set_params(svo_group *g, int frame_number, int render_mode, int buffer_end, int use_beam, int bounces,
                         uint32_t mirror_mask, int spp) {
  if (!g) return SVO_E_INVALID;
  GALL(g, svo_set_params(c, frame_number, render_mode, buffer_end, use_beam, bounces, mirror_mask, spp));
}
int svo_group_set_pipeline(svo_group *g, int pipeline) { if (!g) return SVO_E_INVALID; GALL(g, svo_set_pipeline(c, pipeline)); }
int svo_group_set_tuning(svo_group *g, int waves_per_cu, int thresh) { if (!g) return SVO_E_INVALID; GALL(g, svo_set_tuning(c, waves_per_cu, thresh)); }
int svo_group_set_progressive(svo_group *g, int enabled) { if (!g) return SVO_E_INVALID; GALL(g, svo_set_progressive(c, enabled)); }
int svo_group_set_sequence(svo_group *g, int nframes, int fresh) { if (!g) return SVO_E_INVALID; GALL(g, svo_set_sequence(c, nframes, fresh)); }

int svo_group_resize(svo_group *g, int width, int height) {
  if (!g || width <= 0 || height <= 0) return gfail(g, SVO_E_INVALID, "svo_group_resize: bad size");
  group_ring_free(g);   // the ring's buffers have the size of the frame
  const int tile_rows = (height + 7) / 8, per = (tile_rows + g->n - 1) / g->n;
  for (int r = 0; r < g->n; r++) {
    svo_ctx *c = g->m[(size_t)r];
    int rc = gmember(g, r, svo_resize(c, width, height));
    if (rc) return rc;
    const int cnt = tile_rows > r ? (tile_rows - r + g->n - 1) / g->n : 0;
    rc = gmember(g, r, svo_set_stripes(c, r, g->n, cnt, 0));   // tile rows r, r + n, ... packed from row 0 of the member's chunk
    if (rc) return rc;
  }
  g->width = width; g->height = height; g->rpr = per * 8;
  return SVO_OK;
}

// ---- frames in flight
int svo_group_ring_create(svo_group *g, int slots, int frames_per_slot, int want_hits, int exchange) {
  if (!g) return SVO_E_INVALID;
  if (g->width <= 0) return gfail(g, SVO_E_INVALID, "svo_group_ring_create: svo_group_resize first");
  if (slots < 1 || slots > 8 || frames_per_slot < 1 || frames_per_slot > 64 || exchange < 0 || exchange > 1)
    return gfail(g, SVO_E_INVALID, "svo_group_ring_create: 1..8 slots of 1..64 frames, exchange 0 (peer copies) or 1 (RCCL)");
  group_ring_free(g);
  g->want_hits = want_hits != 0;
  g->planes = 2 + (g->want_hits ? 4 : 0);
  g->exchange = exchange;
  const uint64_t plane_words = (uint64_t)frames_per_slot * (uint64_t)g->rpr * (uint64_t)g->width;
  if (plane_words >= (1ull << 32)) return gfail(g, SVO_E_INVALID, "svo_group_ring_create: slot too large for 32-bit output indices");
  g->chunk_bytes = (uint64_t)g->planes * plane_words * 4;
  if (exchange == 1 && g->n > 1) {
    int rc = group_load_rccl(g);
    if (rc) return rc;
    if (g->comm.empty()) {
      g->comm.assign((size_t)g->n, nullptr);
      rc = gnccl(g, g->nccl_init_all(g->comm.data(), g->n, g->dev.data()), "ncclCommInitAll");
      if (rc) { g->comm.clear(); return rc; }
    }
  }
  svo_ctx *o = g->m[0];
  g->gather.assign((size_t)slots, nullptr);
  g->local.assign((size_t)g->n, std::vector<uint8_t *>((size_t)slots, nullptr));
  for (int b = 0; b < slots; b++) {
    void *p = nullptr;
    int rc = gmember(g, 0, svo_dev_alloc(o, g->chunk_bytes * (uint64_t)g->n, &p));
    if (rc) return rc;
    g->gather[(size_t)b] = (uint8_t *)p;
  }
  for (int r = 0; r < g->n; r++) {
    svo_ctx *c = g->m[(size_t)r];
    // RCCL's send / receive kernels need CU slots next to persistent waves that hold theirs for a whole launch: with the
    // RCCL exchange every member's slot streams leave at least one CU per XCD free (what the torch driver does for its
    // gather, bench.py --comm-cus); the peer-copy exchange runs on the SDMA engines and needs none
    if (exchange == 1 && g->n > 1 && c->reserved_cus < 1) c->reserved_cus = 1;
    int rc = gmember(g, r, ring_create_impl(c, slots, frames_per_slot, want_hits, false));
    if (rc) return rc;
    for (int b = 0; b < slots; b++) {
      uint8_t *base = g->gather[(size_t)b];
      if (r > 0) {
        void *p = nullptr;
        rc = gmember(g, r, svo_dev_alloc(c, g->chunk_bytes, &p));
        if (rc) return rc;
        base = (uint8_t *)p;
      }
      g->local[(size_t)r][(size_t)b] = base;
      const uint64_t plane = plane_words * 4;
      rc = gmember(g, r, svo_ring_bind_slot(c, b, base, base + plane, g->want_hits ? base + 2 * plane : nullptr,
                                            (uint64_t)g->rpr * (uint64_t)g->width));
      if (rc) return rc;
      if (r > 0 && exchange == 0) {
        rc = gmember(g, r, svo_ring_forward_slot(c, b, base, g->gather[(size_t)b] + (uint64_t)r * g->chunk_bytes, g->chunk_bytes, nullptr));
        if (rc) return rc;
        c->ring[(size_t)b].fwd_dst_device = o->device;
      }
    }
  }
  if (exchange == 1 && g->n > 1) {
    GHIP(g, hipSetDevice(o->device));
    g->recv_stream.assign((size_t)slots, nullptr);
    g->recv_done.assign((size_t)slots, nullptr);
    for (int b = 0; b < slots; b++) {
      GHIP(g, hipStreamCreateWithFlags(&g->recv_stream[(size_t)b], hipStreamNonBlocking));
      GHIP(g, hipEventCreateWithFlags(&g->recv_done[(size_t)b], hipEventDisableTiming));
    }
  }
  g->slots = slots; g->frames = frames_per_slot;
  g->slot_first.assign((size_t)slots, 0); g->slot_n.assign((size_t)slots, 0); g->slot_used.assign((size_t)slots, 0);
  g->next = 0;
  return SVO_OK;
}

int svo_group_ring_destroy(svo_group *g) {
  if (!g) return SVO_E_INVALID;
  group_ring_free(g);
  return SVO_OK;
}

static int group_submit(svo_group *g, int frame_number, int nframes, const FrameVar *cams, int *slot, const char *who) {
  if (!g) return SVO_E_INVALID;
  if (g->slots == 0) return gfail(g, SVO_E_INVALID, std::string(who) + ": svo_group_ring_create first");
  const int b = (int)(g->next % (unsigned)g->slots);
  for (int r = 0; r < g->n; r++) {
    int s = -1;
    const int rc = gmember(g, r, ring_submit(g->m[(size_t)r], frame_number, nframes, cams, &s, who));
    if (rc) return rc;   // (a member that fails leaves the group's members out of step: destroy the ring)
    if (s != b) return gfail(g, SVO_E_INVALID, std::string(who) + ": members out of step");
  }
  if (g->exchange == 1 && g->n > 1) {
    // one send / receive pair per member, fused in a group call (one host thread drives every device); a member's send is
    // ordered behind its launch by its slot's stream, the owner's receives run on the slot's receive stream
    svo_ctx *o = g->m[0];
    int rc = gnccl(g, g->nccl_group_start(), "ncclGroupStart");
    for (int r = 1; r < g->n && rc == SVO_OK; r++) {
      rc = gnccl(g, g->nccl_recv(g->gather[(size_t)b] + (uint64_t)r * g->chunk_bytes, (size_t)g->chunk_bytes, ncclUint8, r, g->comm[0],
                                 g->recv_stream[(size_t)b]), "ncclRecv");
      if (rc == SVO_OK)
        rc = gnccl(g, g->nccl_send(g->local[(size_t)r][(size_t)b], (size_t)g->chunk_bytes, ncclUint8, 0, g->comm[(size_t)r],
                                   g->m[(size_t)r]->ring[(size_t)b].stream), "ncclSend");
    }
    const int rc2 = gnccl(g, g->nccl_group_end(), "ncclGroupEnd");
    if (rc) return rc;
    if (rc2) return rc2;
    GHIP(g, hipSetDevice(o->device));
    GHIP(g, hipEventRecord(g->recv_done[(size_t)b], g->recv_stream[(size_t)b]));
  }
  g->slot_first[(size_t)b] = cams ? cams[0].frame_number : frame_number;
  g->slot_n[(size_t)b] = nframes;
  g->slot_used[(size_t)b] = 1;
  g->next++;
  if (slot) *slot = b;
  return SVO_OK;
}

int svo_group_ring_submit(svo_group *g, int frame_number, int nframes, int *slot) {
  return group_submit(g, frame_number, nframes, nullptr, slot, "svo_group_ring_submit");
}
int svo_group_ring_submit_cams(svo_group *g, int nframes, const float *cams, const int *frame_numbers, int *slot) {
  if (!g || !cams || !frame_numbers) return gfail(g, SVO_E_INVALID, "svo_group_ring_submit_cams: null array");
  if (nframes < 1 || nframes > 64) return gfail(g, SVO_E_INVALID, "svo_group_ring_submit_cams: 1..frames_per_slot frames");
  FrameVar v[64];
  for (int k = 0; k < nframes; k++) {
    memcpy(v[k].cam, cams + 15 * (size_t)k, sizeof v[k].cam);
    v[k].frame_number = frame_numbers[k];
  }
  return group_submit(g, frame_numbers[0], nframes, v, slot, "svo_group_ring_submit_cams");
}

// every member's frames of the slot are complete AND have reached the owner
int svo_group_ring_wait(svo_group *g, int slot) {
  if (!g || slot < 0 || slot >= g->slots) return gfail(g, SVO_E_INVALID, "svo_group_ring_wait: no such slot");
  if (!g->slot_used[(size_t)slot]) return SVO_OK;
  for (int r = 0; r < g->n; r++) {
    svo_ctx *c = g->m[(size_t)r];
    svo_ctx::RingSlot &s = c->ring[(size_t)slot];
    GHIP(g, hipSetDevice(c->device));
    GHIP(g, hipEventSynchronize(s.e1));
    if (s.fwd_dst && s.e2) GHIP(g, hipEventSynchronize(s.e2));
    if (g->exchange == 1 && g->n > 1 && r > 0) GHIP(g, hipStreamSynchronize(s.stream));   // its send
  }
  if (g->exchange == 1 && g->n > 1) {
    GHIP(g, hipSetDevice(g->m[0]->device));
    GHIP(g, hipEventSynchronize(g->recv_done[(size_t)slot]));
  }
  return SVO_OK;
}

int svo_group_ring_query(svo_group *g, int slot, int *done, int *first_frame, int *nframes, float *gpu_ms) {
  if (!g || slot < 0 || slot >= g->slots) return gfail(g, SVO_E_INVALID, "svo_group_ring_query: no such slot");
  bool fin = true;
  float ms_max = 0.0f;
  for (int r = 0; r < g->n && g->slot_used[(size_t)slot]; r++) {
    svo_ctx *c = g->m[(size_t)r];
    svo_ctx::RingSlot &s = c->ring[(size_t)slot];
    GHIP(g, hipSetDevice(c->device));
    hipError_t q = hipEventQuery((s.fwd_dst && s.e2) ? s.e2 : s.e1);
    if (q == hipErrorNotReady) { fin = false; continue; }
    if (q != hipSuccess) return gfail(g, SVO_E_HIP, std::string("svo_group_ring_query: ") + hipGetErrorString(q));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.e0, s.e1) == hipSuccess && ms > ms_max) ms_max = ms;
  }
  if (fin && g->exchange == 1 && g->n > 1 && g->slot_used[(size_t)slot]) {
    GHIP(g, hipSetDevice(g->m[0]->device));
    if (hipEventQuery(g->recv_done[(size_t)slot]) == hipErrorNotReady) fin = false;
  }
  if (done) *done = fin ? 1 : 0;
  if (first_frame) *first_frame = g->slot_first[(size_t)slot];
  if (nframes) *nframes = g->slot_used[(size_t)slot] ? g->slot_n[(size_t)slot] : 0;
  if (gpu_ms) *gpu_ms = fin ? ms_max : 0.0f;   // the slowest member's launch
  return SVO_OK;
}

// frame k of a slot in frame order: plane 0 colour / 1 depth (4 bytes per pixel) or the hit records (16 bytes per pixel)
static int group_read_plane(svo_group *g, int slot, int k, int plane, size_t px_bytes, void *dst, const char *who) {
  if (!g || !dst) return gfail(g, SVO_E_INVALID, std::string(who) + ": null buffer");
  if (slot < 0 || slot >= g->slots || !g->slot_used[(size_t)slot] || k < 0 || k >= g->slot_n[(size_t)slot])
    return gfail(g, SVO_E_INVALID, std::string(who) + ": the slot does not hold that frame");
  int rc = svo_group_ring_wait(g, slot);
  if (rc) return rc;
  svo_ctx *o = g->m[0];
  GHIP(g, hipSetDevice(o->device));
  const size_t W = (size_t)g->width, rows = (size_t)g->rpr;
  const uint64_t plane_bytes = (uint64_t)g->frames * rows * W * 4;   // one 4-byte plane of a chunk, all frames
  std::vector<uint8_t> tmp(rows * W * px_bytes);
  for (int r = 0; r < g->n; r++) {
    const uint8_t *src = g->gather[(size_t)slot] + (uint64_t)r * g->chunk_bytes + (uint64_t)plane * plane_bytes +
                         (uint64_t)k * rows * W * px_bytes;
    GHIP(g, hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
    for (size_t j = 0; j * 8 < rows; j++) {                   // tile row j of member r = tile row r + j n of the frame
      const size_t y0 = ((size_t)r + j * (size_t)g->n) * 8;
      if (y0 >= (size_t)g->height) break;
      const size_t nrows = std::min<size_t>(8, (size_t)g->height - y0);
      memcpy((uint8_t *)dst + y0 * W * px_bytes, tmp.data() + j * 8 * W * px_bytes, nrows * W * px_bytes);
    }
  }
  return SVO_OK;
}
int svo_group_ring_read_color(svo_group *g, int slot, int k, void *rgba8) { return group_read_plane(g, slot, k, 0, 4, rgba8, "svo_group_ring_read_color"); }
int svo_group_ring_read_depth(svo_group *g, int slot, int k, float *depth) { return group_read_plane(g, slot, k, 1, 4, depth, "svo_group_ring_read_depth"); }
int svo_group_ring_read_hits(svo_group *g, int slot, int k, svo_hit *hits) {
  if (g && !g->want_hits) return gfail(g, SVO_E_INVALID, "svo_group_ring_read_hits: the ring was created without hit records");
  return group_read_plane(g, slot, k, 2, 16, hits, "svo_group_ring_read_hits");
}
// the crosshair pick of Main.java:132-146: 4 (+ 4 + 16) bytes from the member chunk that holds the pixel
int svo_group_ring_read_pixel(svo_group *g, int slot, int k, int x, int y, void *rgba8, float *depth, svo_hit *hit) {
  if (!g) return SVO_E_INVALID;
  if (x < 0 || y < 0 || x >= g->width || y >= g->height) return gfail(g, SVO_E_INVALID, "svo_group_ring_read_pixel: outside the image");
  if (slot < 0 || slot >= g->slots || !g->slot_used[(size_t)slot] || k < 0 || k >= g->slot_n[(size_t)slot])
    return gfail(g, SVO_E_INVALID, "svo_group_ring_read_pixel: the slot does not hold that frame");
  if (hit && !g->want_hits) return gfail(g, SVO_E_INVALID, "svo_group_ring_read_pixel: the ring was created without hit records");
  int rc = svo_group_ring_wait(g, slot);
  if (rc) return rc;
  GHIP(g, hipSetDevice(g->m[0]->device));
  const int ty = y >> 3, r = ty % g->n, j = ty / g->n;
  const size_t W = (size_t)g->width, rows = (size_t)g->rpr;
  const uint64_t plane_bytes = (uint64_t)g->frames * rows * W * 4;
  const uint64_t px = ((uint64_t)k * rows + (uint64_t)(j * 8 + (y & 7))) * W + (uint64_t)x;
  const uint8_t *chunk = g->gather[(size_t)slot] + (uint64_t)r * g->chunk_bytes;
  if (rgba8) GHIP(g, hipMemcpy(rgba8, chunk + px * 4, 4, hipMemcpyDeviceToHost));
  if (depth) GHIP(g, hipMemcpy(depth, chunk + plane_bytes + px * 4, 4, hipMemcpyDeviceToHost));
  if (hit) GHIP(g, hipMemcpy(hit, chunk + 2 * plane_bytes + px * 16, 16, hipMemcpyDeviceToHost));
  return SVO_OK;
}

}  // extern "C"
