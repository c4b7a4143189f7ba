/*
 * svo_hip.h -- C ABI of libsvohip.so: the MI355X-native replacement for the reference's
 * GPU compute path (src/shaders/svotrace.comp dispatched through LWJGL OpenGL).
 *
 * Every entry point replaces a piece of the reference's host <-> GPU interface; the
 * citation after each declaration is the reference code it stands in for
 * (paths relative to /root/reference).  Plain pointers and sizes only: this is what a
 * JNI / FFI stub binds (see include/svo_hip_jni.h and INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 on success, a negative SVO_E_* code otherwise;
 *     svo_last_error() gives the text.  Nothing throws, nothing prints (the reference's
 *     GL path polls glGetError and prints, Renderer.java:160-165; the Java twin keeps that).
 *   - the library copies on upload / update and writes into caller memory on readback;
 *     it never retains a caller pointer across calls.
 *   - a context is bound to one GPU and is not re-entrant (the reference drives GL from
 *     the single thread that owns the context, Window.java:40).
 *   - ordering: svo_dispatch_async only enqueues.  A caller that alternates streams (svo_set_stream)
 *     to keep frames in flight orders its own output buffers; the library orders everything it owns:
 *     per-frame work counters and sample accumulators are re-used only after the frame that used
 *     them has finished (GPU-side event waits), and every call that changes or moves the pool or the
 *     library's images (svo_pool_upload / _update / _reserve / _upload_device, svo_resize,
 *     svo_destroy) first waits for the whole device, so a frame in flight on any stream sees the
 *     pool either before or after the edit, never torn.
 *   - pools of up to 13 levels (the reference's MAX_DEPTH, "up to 8192^3") are supported; on deeper
 *     pools rays that descend below level 13 re-use the deepest stack slot.
 *   - the pool bytes are taken exactly as Octree.getByteBuffer() holds them
 *     (Octree.java:68-176); no re-encoding happens at the boundary.
 */
#ifndef SVO_HIP_H
#define SVO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_OK 0
#define SVO_E_INVALID (-1)   /* bad argument (null, bad range, size <= 0) */
#define SVO_E_NOPOOL (-2)    /* dispatch before a pool upload */
#define SVO_E_HIP (-3)       /* a HIP runtime call failed */
#define SVO_E_NODEVICE (-4)  /* no such GPU */
#define SVO_E_TOOLARGE (-5)  /* pool >= 2^31 bytes: child pointers are signed 32-bit (Octree.java:162-168) */

typedef struct svo_ctx svo_ctx;

/* Per-pixel record of the FIRST cast of the pixel (the primary ray).  `pointer` is the
 * byte offset of the hit node in the pool -- the "hit voxel ID"; 0 = miss (the
 * convention of svobeam.comp:538,553; the live trace shader's store is commented out,
 * svotrace.comp:728). */
typedef struct svo_hit {
  uint32_t pointer;
  uint16_t raw_normal; /* leafMask field of the hit node = packed normal for surface leaves (svotrace.comp:382-388) */
  uint8_t value;       /* material (svotrace.comp:404) */
  uint8_t level;       /* MAX_SCALE - scale (svotrace.comp:408) */
  uint32_t iter;       /* traversal iterations of that cast (svotrace.comp:405) */
  float t;             /* res.t (svotrace.comp:403); 0 on a miss */
} svo_hit;

typedef struct svo_stats {
  uint64_t pixels;
  uint64_t rays;        /* intersectOctree-equivalent casts of the last counted frame (all-NaN rays excluded) */
  uint64_t nan_rays;    /* casts whose origin or direction is entirely NaN: exited at once, reported iter = 1501 */
  uint64_t iterations;  /* traversal loop iterations of the counted rays */
  uint64_t alg_bytes;   /* algorithmic bytes: 7 per cast + size (7/3/7/1) of every fetched child record */
  uint64_t max_iter;
  float last_dispatch_ms; /* GPU time of the last svo_dispatch*, HIP events on the library's stream */
  int32_t device;
} svo_stats;

/* ---- lifetime ------------------------------------------------------------------ */
/* replaces: GL context + Renderer singleton creation (Window.java:24-50, Renderer.java:14-36) */
/* which build of the library this is: bit 0 = the comparators and A/B switches are built in (libsvohip_variants.so: pipeline 2,
 * the spare-ray kernel, the SVO_* environment switches; the product library libsvohip.so reports 0 there), bit 1 = the traversal
 * trips are the hand-written gfx950 assembly (0: hipcc's translation of the same statement, libsvohip_cxxloop.so) */
int svo_build_info(void);
int svo_create(int device, svo_ctx **out);
int svo_destroy(svo_ctx *ctx);
const char *svo_last_error(const svo_ctx *ctx); /* replaces Renderer.printGLErrors, Renderer.java:160-165 */

/* ---- SVO pool (SSBO binding 7) --------------------------------------------------- */
/* replaces Renderer.addSSBO(7, buf) / updateSSBO(7, buf): glBufferData of the whole pool
 * (Renderer.java:123-134, Main.java:122).  nbytes = bytes that are meaningful (memOffset);
 * reads past them return 0 like the reference's zero-filled over-allocation. */
int svo_pool_upload(svo_ctx *ctx, const void *host, uint64_t nbytes);
/* replaces Renderer.updateSSBO(7, buf, start, end): glBufferSubData of [start, end)
 * (Renderer.java:136-146, Main.java:349-350).  host_base points at byte 0 of the pool.
 * start >= end is rejected like the reference does (it prints and returns). */
int svo_pool_update(svo_ctx *ctx, const void *host_base, uint64_t start, uint64_t end);
/* replaces Renderer.getSSBO: glGetBufferSubData (Renderer.java:148-150) */
int svo_pool_download(svo_ctx *ctx, void *host, uint64_t nbytes);
/* device-side view of the pool, for a multi-GPU broadcast straight into it (RCCL);
 * svo_pool_reserve allocates without uploading. */
int svo_pool_reserve(svo_ctx *ctx, uint64_t nbytes);
/* same as svo_pool_upload but the source is device memory on the context's GPU (e.g. the
 * receive buffer of an RCCL broadcast owned by torch) */
int svo_pool_upload_device(svo_ctx *ctx, const void *dptr, uint64_t nbytes);
int svo_pool_device_ptr(svo_ctx *ctx, void **dptr, uint64_t *nbytes);
/* the caller has finished writing the pool through svo_pool_device_ptr (e.g. an RCCL broadcast into it has completed
 * or been enqueued): waits for the whole device, then re-reads what the library derives from the pool (dword 0 for
 * the debug square, the beam pass's liveness table, the interior-descriptor table).  Call it before the next dispatch;
 * a dispatch without it sees the derived data of the pool as it was when svo_pool_device_ptr was called. */
int svo_pool_commit(svo_ctx *ctx);

/* ---- world generation -------------------------------------------------------------------- */
/* replaces Octree.constructCompleteOctree(chunkGenShader, voxelTexture, heightmapTexture, materialTexture)
 * (Octree.java:192-353): the chunkgen-heightmap.comp voxel rule (:16-28: a column is solid for y <= height, its top
 * five layers take the material map's value, the rest value 1), OctreeThread / constructInnerOctree over 512^3
 * sub-cubes of 1024^3 chunks (Octree.java:511-670, OctreeThread.java:20-23) and the chunk splice (:317-343) -- done on
 * the GPU from the two maps, without the dense voxel grid.  height / material: host arrays of n*n elements indexed
 * [z*n + x], heights already in voxels (the reference scales its 16-bit PNG sample by 2048/65536 in the shader),
 * materials 1..255.  n = world edge, a power of two in 8..8192.  The result becomes the context's pool (as after
 * svo_pool_upload; read it with svo_pool_download); *out_nbytes = its size.  SVO_E_TOOLARGE if it would not fit
 * signed 32-bit child pointers. */
int svo_build_from_heightmap(svo_ctx *ctx, const uint16_t *height, const uint8_t *material, int n, uint64_t *out_nbytes);
/* the same from the maps as Octree.constructCompleteOctree feeds them to the shader: raw 16-bit samples of the height PNG
 * (stbi_load_16 -> r16ui image, Octree.java:208-216), which the shader scales itself -- heightSample = int(r / 65536.0 *
 * 2048) (chunkgen-heightmap.comp:16-19) = r >> 5, pinned with the rest of the voxel rule by the reference shader's own runs
 * (tests/golden/chunkgen_golden.npz: bytes >= 128 of the signed r8i material image survive as they are). */
int svo_build_from_heightmap16(svo_ctx *ctx, const uint16_t *raw16, const uint8_t *material, int n, uint64_t *out_nbytes);
/* replaces OctreeThread.run / Octree.constructInnerOctree (OctreeThread.java:20-23, Octree.java:511-670) for a dense
 * chunk of voxels, whatever produced them (the reference fills chunks from height maps or from 3-D noise,
 * chunkgen*.comp): voxels[x | y << log2 n | z << 2 log2 n] (Octree.java:110-112), 0 = empty, n a power of two in
 * 2..1024 (one chunk).  The result becomes the context's pool, as above. */
int svo_build_from_voxels(svo_ctx *ctx, const uint8_t *voxels, int n, uint64_t *out_nbytes);

/* ---- per-frame state (the shader's uniforms) ------------------------------------- */
/* replaces glUniform3fv(8,pos), (1..4, l1,l2,r1,r2) (Main.java:269-273); values as
 * Camera.getUniform() returns them (Camera.java:142-151) */
int svo_set_camera(svo_ctx *ctx, const float pos[3], const float l1[3], const float l2[3], const float r1[3],
                   const float r2[3]);
/* replaces glUniform1i(5 frameNumber | 6 renderMode | 9 bufferEnd | 11 useBeam)
 * (Main.java:275-283).  bounces / mirror_mask / spp expose the shader's dormant
 * features (svotrace.comp:444, 500-504, 668-670); the live behaviour is 2 / 0 / 1.
 * use_beam != 0 = useBeamOptimization (Main.java:51, 257-266, 280-283; default off): every dispatch is preceded by a
 * coarse pass that finds, per 4x4 pixel block, a distance before which none of its rays can meet a voxel, and primary
 * rays start their walk there.  The reference's own coarse pass (svobeam.comp) is dormant and inconsistent; here the
 * feature is exact: colour, depth and the hit records' pointer / value / normal / level / t are the same bytes as with
 * use_beam = 0, only the records' iteration counts drop (renderMode 1 displays them, so its colours change).
 * spp > 1 on pipeline 1: one launch renders every sample of the frame into slots of its own (768 B per pixel at 64 spp,
 * four such buffers for frames in flight) and a second kernel adds them up in sample order; when a buffer would exceed
 * 4 GB (environment SVO_FOLD_BYTES overrides) the samples are one launch each with running sums.  Same bytes. */
int svo_set_params(svo_ctx *ctx, int frame_number, int render_mode, int buffer_end, int use_beam, int bounces,
                   uint32_t mirror_mask, int spp);
/* replaces the image allocations: rgba8 WxH on unit 0, r32f WxH on unit 1 (Main.java:66-78) */
int svo_resize(svo_ctx *ctx, int width, int height);
/* multi-GPU screen-tile split: render only pixel rows [y0, y1) (multiples of 8 except
 * the last); default = whole frame; svo_resize to a different size resets it.  No reference
 * equivalent (single GPU). */
int svo_set_rows(svo_ctx *ctx, int y0, int y1);
/* interleaved variant for load balance: render tile rows first + j*step, j = 0..n-1 (8 pixel rows each,
 * rows past the image height are skipped) and store them PACKED: tile row j lands at output rows
 * out_row0 + 8j .. out_row0 + 8j + 7.  Rank r of N uses (r, N, ceil((tile_rows - r) / N), r * band_rows). */
int svo_set_stripes(svo_ctx *ctx, int first_tile_row, int tile_row_step, int n_tile_rows, int out_row0);
/* 0 = one thread per pixel (the reference's decomposition); 1 = persistent waves with lane
 * refill and in-place bounce regeneration; 2 = stage-per-kernel wavefront tracing with
 * compacted ray queues (a comparator: built into libsvohip_variants.so only; libsvohip.so refuses it with SVO_E_INVALID).
 * All three produce identical bytes; a new context runs pipeline 1 (the fast one). */
int svo_set_pipeline(svo_ctx *ctx, int pipeline);
/* pipeline-1 launch shape: persistent waves per CU and the round threshold in sixteenths (0 = default 9: a round starts once
 * at most 9/16 of the lanes that were traversing at the start of the burst are still traversing).  waves_per_cu = 0 (what a
 * new context has) = automatic: a waiting dispatch on the context's stream fills the GPU (24 per CU: right for one launch at a
 * time); svo_dispatch_async while it takes turns on n image sets takes about 32 / n (4 sets: 8); the
 * submissions of a ring with more than one slot (svo_ring_create / svo_group_ring_create) take 10 per launch, so that the next
 * launch's waves find CU slots while the previous launch drains -- the shape bench.py's headline is measured on, without any
 * call.  A positive value is used as given everywhere (a caller that alternates its own streams with svo_set_stream: ~10).
 * The threshold 9 is the optimum of the bench scenes (8: -0.6 % on the default cell, -1.4 % on the worst terrain cell); scenes
 * whose rays take hundreds of iterations -- the "dust" cells of profiles/round6_matrix.md -- prefer 8 by 1.4 ... 3.5 %
 * (profiles/round6_sweep_hostile_cell.md): a host that renders such scenes calls svo_set_tuning(ctx, 0, 8). */
int svo_set_tuning(svo_ctx *ctx, int waves_per_cu, int round_threshold_sixteenths);
/* shape of the last pipeline-1 launch of the context: persistent waves launched, waves per CU they were sized by, round
 * threshold in sixteenths (any pointer may be NULL).  Diagnostic: what svo_set_tuning's automatic choice resolved to. */
int svo_launch_info(svo_ctx *ctx, int *waves, int *waves_per_cu, int *round_threshold_sixteenths);
/* the reference's dormant cross-frame accumulation (commented out at svotrace.comp:712-719; MAX_FRAME_ITER :43):
 * when enabled and frameNumber > 1, a pixel's colour becomes (frameNumber * last + colour) / (frameNumber + 1), `last`
 * being what the colour image holds from the previous dispatch (rgba8, as imageLoad returns it), and stays `last` from
 * frameNumber 100 on.  Default off = the live shader.  The caller keeps rendering into the same colour image and
 * resets frameNumber when the camera moves, as Main.java does (:16, :275). */
int svo_set_progressive(svo_ctx *ctx, int enabled);
/* BASELINE config 5 in the reference's terms ("64 spp accumulated GI" = the accumulation above over 64 frames): with
 * svo_set_progressive(1), every dispatch renders nframes consecutive frames of the accumulation -- frameNumber,
 * frameNumber + 1, ... frameNumber + nframes - 1 -- into the ONE colour image, exactly what nframes dispatches with
 * frameNumber advancing leave there (Main.java:275 + svotrace.comp:712-719; depth and hit image: the last frame's).
 * fresh != 0: the sequence starts on a zeroed image (the application's first frames: frameNumber 2 on the image
 * glTexStorage2D left); 0: on the image as the previous dispatch left it.  On the persistent pipeline the whole sequence
 * is ONE launch (every frame's colour kept in a slot of its own, 12 bytes per pixel and frame) followed by a pass that
 * applies the reference's recurrence in frame order, quantising to rgba8 between frames as imageStore / imageLoad do: the
 * same bytes as one dispatch per frame, without their tails.  Sequences that do not fit 4 GB of slots, more than one
 * sample per pixel, and the other pipelines fall back to one launch per frame.  nframes = 1 (default): one frame per
 * dispatch.  A sequence is one frame of a batch: svo_set_batch must be 1.
 * On a ring of MORE than one slot (svo_ring_* / svo_group_ring_*) every submission lands in another image, so a progressive
 * submission must start fresh (fresh != 0): with fresh = 0 -- or plain svo_set_progressive(1) without a sequence -- "the
 * image the previous dispatch left" would be the frame of `slots` submissions ago, and the submission is refused
 * (SVO_E_INVALID).  Continue an accumulation with svo_dispatch, or on a ring of one slot. */
int svo_set_sequence(svo_ctx *ctx, int nframes, int fresh);
/* Throughput mode: every dispatch renders `nframes` consecutive frames of the current camera -- frameNumber,
 * frameNumber + 1, ... exactly what nframes turns of Main.updateEarly with a static camera render (Main.java:275 only
 * increments frameNumber) -- frame k into the bound (caller-owned) outputs at element offset k * frame_stride (the same
 * element count for colour, depth and hit records).  On the persistent pipeline the batch is ONE launch whose waves run
 * from frame to frame, so the launch's tail (its longest paths) is paid once per batch instead of once per frame: +5..8 %
 * on a full 1080p frame, 1.5x on the 1/8 frame of an 8-GPU split.  The bytes of every frame are those of a dispatch of
 * its own.  nframes = 1 (default) = the reference's one dispatch per frame.  svo_count_frame counts the first frame. */
int svo_set_batch(svo_ctx *ctx, int nframes, uint64_t frame_stride);
/* The interior-descriptor table: a derived acceleration copy of the pool inside the library ("a derived
 * acceleration copy inside the library is allowed, results must not change").  The shader fetches a child record in
 * every iteration only to learn "empty?" (svotrace.comp:295) and "leaf?" (:311); the table keeps those two answers for all
 * eight children (a nibble each: empty / leaf / "descend, into the n-th child descriptor") in an 8-byte descriptor of the
 * PARENT, so that only a descend loads (one aligned descriptor) and the
 * pool's records are read once per cast, for the node it ends on.  Built on the GPU at the first dispatch after a pool
 * change (svo_pool_upload / _update / builders), walked by pipeline 1.  mode 1 (default): use it when the pool can be
 * derived (up to 13 levels, unrolling within budget -- anything a builder produces); otherwise, and with mode 0, the
 * records are walked as the shader does.  Same bytes either way. */
int svo_set_derived(svo_ctx *ctx, int mode);
/* builds the table if the pool changed; *descriptors = its entries, *bytes = device memory it holds, *walkable = 1 if
 * pipeline 1 walks it (0 = the pool is not derivable), *build_ms = GPU time of the last build.  Any may be NULL. */
int svo_derived_info(svo_ctx *ctx, uint64_t *descriptors, uint64_t *bytes, int *walkable, float *build_ms);
/* svo_pool_update (Renderer.updateSSBO of an SDF brush stroke's two byte ranges, Main.java:349-350) does not drop a
 * walkable table: the states whose child block the range touches are recomputed, changed sibling groups and new
 * subtrees are appended at the table's end (SVO_DERIVED_REFRESH=0 in the environment: rebuild instead).  *refreshes =
 * updates followed that way so far, and of the last one: *states recomputed, *added descriptors, GPU
 * time.  Any may be NULL. */
int svo_derived_refresh_info(svo_ctx *ctx, uint64_t *refreshes, uint64_t *states, uint64_t *added, float *gpu_ms);
/* record per-pixel svo_hit (costs 16 B/pixel of stores); default on */
int svo_set_hit_records(svo_ctx *ctx, int enabled);

/* ---- dispatch ------------------------------------------------------------------- */
/* replaces Renderer.useProgram + dispatchCompute(traceShader, W/8, H/8, 1):
 * glDispatchCompute + glMemoryBarrier (Renderer.java:114-121, Main.java:267,285) -- and waits for the frame
 * (last_dispatch_ms of svo_stats is its GPU time). */
int svo_dispatch(svo_ctx *ctx);
/* the same with GL's own semantics: both GL calls return at once and the reference's wait is the next frame's glGetTexImage
 * (Main.java:132-146), so its quad draw / ImGui / input overlap the trace.  Only enqueues on the context's stream; every
 * svo_read_* waits for that stream first, every pool mutator for the whole device.  What Renderer.dispatchCompute of the
 * host mirrors (HipRenderer.java, host/svo_host.hpp) calls. */
int svo_dispatch_async(svo_ctx *ctx);
/* The reference's loop needs ONE pixel of frame N before it dispatches frame N + 1 (Main.updateEarly: the crosshair depth,
 * Main.java:132-146 -- there a glGetTexImage of the whole 8.3 MB depth image that waits for the frame; SURVEY T11 "where the
 * time goes today").  Here the two do not wait for each other:
 *   - svo_dispatch_async takes turns on several sets {stream, colour / depth / hit images} while the library owns them (no
 *     svo_set_stream / svo_bind_outputs, pipeline 1, no cross-frame accumulation, no batch): frame N + 1's persistent waves take
 *     the CUs frame N's tail frees.  svo_read_color / _depth / _hits / _pixel and svo_output_device_ptrs always name the LAST
 *     dispatched frame (GL's semantics: a read-back sees the last dispatch) and wait for it alone; svo_sync waits for all.
 *     svo_set_overlap(ctx, n): 0 = no alternation (one stream, one image set, as before round 6), 1 = the default (4 sets: up
 *     to four frames in flight, like a swap chain), 2 .. 8 = that many sets.  The host never runs further ahead than that:
 *     before a set is rendered into again svo_dispatch_async waits for the set's previous frame to be complete.
 *   - the pick pixel -- svo_set_pick(x, y); default the image centre, Main.java:139-141 -- is answered without waiting for its
 *     frame: in front of the frame's kernels a launch of ONE wave on a high-priority stream of its own walks that pixel's path
 *     (the same device functions on the same values: the same bits) and writes {rgba8, depth, hit record} and the dispatch's
 *     sequence number to pinned host memory; svo_read_pixel at that position polls the word (no stream synchronisation, no copy).
 *     The frame's own kernels carry nothing for it.  Any other position, a frame that gets no pick launch (stripes / row bands,
 *     batches, accumulation, several samples per pixel, the beam pre-pass) or a negative x (= no pick) take the waiting path:
 *     same values either way (tests/test_gpu_pick.py). */
int svo_set_pick(svo_ctx *ctx, int x, int y);
int svo_set_overlap(svo_ctx *ctx, int sets);   /* 0 = one set, 1 = the default (4), 2 .. 8 = that many */
/* the pick position in force (-1, -1: none) and how many svo_read_pixel calls were answered from the mail / by waiting for
 * their frame since the context was made.  Any pointer may be NULL. */
int svo_pick_info(svo_ctx *ctx, int *x, int *y, uint64_t *from_mail, uint64_t *waited);
int svo_sync(svo_ctx *ctx);
/* run the frame once more with counters on and fill svo_stats (untimed diagnostic pass) */
int svo_count_frame(svo_ctx *ctx, svo_stats *out);
int svo_get_stats(svo_ctx *ctx, svo_stats *out);
/* use a caller-owned hipStream_t (e.g. torch's current stream) instead of the library's.  May be
 * called between dispatches to alternate streams: with separate output buffers (svo_bind_outputs)
 * several frames can then be in flight, the next one filling the GPU while the previous one drains its longest paths.
 * The library rotates its per-frame work counters / queues / accumulators / beam images over small rings and
 * orders their re-use with events, so any number of dispatches may be outstanding (more than 8 serialise).
 * HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): two frame streams on one queue serialise. */
int svo_set_stream(svo_ctx *ctx, void *hip_stream);   /* NULL: back to the library's own stream (kept for the context's lifetime) */
/* time `iters` back-to-back frames with HIP events on the dispatch stream after `warmup`
 * untimed ones; per-frame milliseconds into ms[iters] */
int svo_time_frames(svo_ctx *ctx, int warmup, int iters, float *ms);

/* ---- frames in flight behind the boundary --------------------------------------------- */
/* The reference's loop renders one frame and reads the crosshair back (Main.java:132-146, 257-289): svo_dispatch +
 * svo_read_pixel.  The throughput mode bench.py times -- several launches in flight, several consecutive frames of the
 * camera per launch, so that a launch's tail (its longest paths) overlaps the next launch -- needs streams and output
 * buffers a host like Java cannot own.  The ring owns them: `slots` output sets (colour, depth [, hit records]) of
 * `frames_per_slot` frames each, every slot with a HIP stream of its own.  svo_resize (a new image size) destroys the
 * ring; pool changes wait for every frame in flight as always.  The process should run with GPU_MAX_HW_QUEUES >= slots
 * (svo_create sets 8 when it makes the process's first HIP call and the variable is unset). */
int svo_ring_create(svo_ctx *ctx, int slots, int frames_per_slot, int want_hits);
/* Persistent waves hold their CU slots for a whole launch; a collective's kernels (RCCL send / receive of the tile
 * gather) queued next to four such launches only start when a launch drains.  per_xcd > 0 makes the NEXT
 * svo_ring_create build its streams with a CU mask that leaves that many CUs of each of the 8 XCDs free
 * (hipExtStreamCreateWithCUMask), so that a collective on another stream always finds room; 0 (default) = all CUs. */
/* NOTE: HIP offers CU-masked streams only as default (blocking) streams: unlike the ring's unreserved streams
 * (hipStreamNonBlocking) they synchronise implicitly with the legacy null stream.  While a ring with reserved CUs has
 * frames in flight, keep work off the null stream (hipMemcpy / hipMemset without a stream, a framework's default
 * stream): each such call serialises against every slot.  The library's own null-stream calls are all in functions that
 * wait for the frames anyway (svo_ring_read_*, svo_pool_update, svo_resize). */
int svo_set_reserved_cus(svo_ctx *ctx, int per_xcd);
int svo_ring_destroy(svo_ctx *ctx);
/* enqueue frames frame_number .. frame_number + nframes - 1 (what nframes turns of Main.updateEarly with a static
 * camera render, Main.java:275) with the context's current camera / params / stripes / tuning into the next slot
 * (round robin; a slot's re-use is ordered behind its previous frames by its stream); returns at once. */
int svo_ring_submit(svo_ctx *ctx, int frame_number, int nframes, int *slot);
/* the same for a camera that moves: frame k of the submission carries its own camera -- cams[15 k .. 15 k + 14] = pos, l1,
 * l2, r1, r2 as Camera.getUniform yields them (Camera.java:142-151) -- and its own frameNumber (Main.updateEarly moves the
 * camera and resets frameNumber to 0 on any motion, pre-incremented to 1 before the dispatch: Main.java:161-236, 275).
 * One persistent launch carries all nframes frames; every frame's bytes are those of svo_set_camera + svo_set_params +
 * svo_dispatch of its own.  The context's camera and frameNumber are left as they were.  The arrays are copied before
 * the call returns.  With nframes > 1 the beam pre-pass (one camera's) is refused. */
int svo_ring_submit_cams(svo_ctx *ctx, int nframes, const float *cams, const int *frame_numbers, int *slot);
/* host waits until the slot's last submission is complete */
int svo_ring_wait(svo_ctx *ctx, int slot);
/* non-blocking: *done = 1 when complete; the first frameNumber and the number of frames it holds; GPU milliseconds
 * between the start and the end of its launches (HIP events on the slot's stream; 0 while running).  Any may be NULL. */
int svo_ring_query(svo_ctx *ctx, int slot, int *done, int *first_frame, int *nframes, float *gpu_ms);
/* readback of frame k of a slot (waits for the slot): glGetTexImage of images 0 / 1 (Main.java:132-146) */
int svo_ring_read_color(svo_ctx *ctx, int slot, int k, void *rgba8);
int svo_ring_read_depth(svo_ctx *ctx, int slot, int k, float *depth);
int svo_ring_read_hits(svo_ctx *ctx, int slot, int k, svo_hit *hits);
int svo_ring_read_pixel(svo_ctx *ctx, int slot, int k, int x, int y, void *rgba8, float *depth, svo_hit *hit);
/* multi-GPU: a slot may render into caller-owned device buffers instead (a rank's chunk of an RCCL gather buffer):
 * frame k at element offset k * frame_stride of each; color == NULL returns the slot to its own images */
int svo_ring_bind_slot(svo_ctx *ctx, int slot, void *color, void *depth, void *hits, uint64_t frame_stride);
/* device pointers of a slot's images, their frame stride in elements, and the slot's hipStream_t (to order a
 * collective that reads the slot behind its frames).  Any may be NULL. */
int svo_ring_device_ptrs(svo_ctx *ctx, int slot, void **color, void **depth, void **hits, uint64_t *frame_stride, void **stream);

/* The tile exchange without a collective's kernels (one process per GPU): a rank's slot may forward what it
 * rendered straight into the frame owner's memory.  After every svo_ring_submit into `slot` the library enqueues, on the
 * slot's stream behind the launch, a device-to-device copy of nbytes from src (this rank's chunk, e.g. what
 * svo_ring_bind_slot bound) to dst (the owner's gather buffer at this rank's chunk, opened with svo_ipc_open), then the
 * number of the submission (1, 2, ... over the ring's lifetime) into the 32-bit word *flag (owner's memory as well,
 * NULL = none): the owner knows a rank's frames have landed when the word has reached the submission it waits for.
 * Between GPUs such copies run on the SDMA engines: they need no CU slot next to the persistent waves, which an RCCL
 * send / receive does.  dst == NULL switches forwarding off.
 * LOCKSTEP: the word carries the sender's submission count, so owner and senders must submit the same sequence of
 * dispatches from ring creation on, and the owner must have read (or given up) a slot's previous frames before any rank
 * submits into that slot again -- there is no back-pressure from the owner to the senders (a rank that ran a whole ring
 * ahead would overwrite frames not yet read, and the word would still compare >=).  bench.py and svo_group_* (one host
 * thread submits for every member) satisfy that by construction. */
int svo_ring_forward_slot(svo_ctx *ctx, int slot, const void *src, void *dst, uint64_t nbytes, void *flag);
/* device memory that can be shared with the other ranks of the node: plain allocations (zeroed), their 64-byte IPC
 * handles, and a peer's allocation opened from its handle (hipIpcGetMemHandle / hipIpcOpenMemHandle) */
int svo_dev_alloc(svo_ctx *ctx, uint64_t nbytes, void **dptr);
int svo_dev_free(svo_ctx *ctx, void *dptr);
int svo_dev_read(svo_ctx *ctx, const void *dptr, void *host, uint64_t nbytes);
int svo_ipc_export(svo_ctx *ctx, void *dptr, void *handle64);
int svo_ipc_open(svo_ctx *ctx, const void *handle64, void **dptr);
int svo_ipc_close(svo_ctx *ctx, void *dptr);

/* ---- N GPUs of one node behind the boundary ----------------------------------------------------------------------
 * SURVEY 8(b) Threading: "a context is single-threaded; one context per GPU; the multi-GPU driver owns 8 contexts" -- this
 * is that driver, inside the library, for a host that is ONE process with one render thread (the reference's Java host).
 * The path shards by screen tile: member r of n renders tile rows r, r + n, r + 2n, ... of every frame;
 * behind each launch its packed stripes travel to member 0 (the frame owner) -- exchange 0: a peer copy enqueued on the
 * member's own stream (SDMA over xGMI, no CU slots needed next to the persistent waves); exchange 1: one RCCL send /
 * receive pair per member inside ncclGroupStart / End (librccl is loaded on demand) -- and the owner hands out whole
 * frames in frame order.  The pool is replicated: one upload from the host, n - 1 peer copies.  Every call is made from
 * the one host thread; results are byte-identical to a single context's.  `devices` may name a device more than once
 * (how the tests run n = 2, 3, 8 on one GPU; RCCL refuses that, peer copies do not).
 * Replaces, for n GPUs: Renderer.addSSBO / updateSSBO (pool), the uniforms of Main.java:267-285, Renderer.dispatchCompute
 * (Main.java:285) and the glGetTexImage readbacks (Main.java:132-146). */
typedef struct svo_group svo_group;
int svo_group_create(const int *devices, int n, svo_group **out);
int svo_group_destroy(svo_group *g);
const char *svo_group_last_error(const svo_group *g);
int svo_group_size(const svo_group *g);
/* member i's context, for what is per GPU (svo_get_stats, svo_derived_info, svo_set_derived, ...); do not resize, bind,
 * or submit through it */
svo_ctx *svo_group_member(svo_group *g, int i);
int svo_group_pool_upload(svo_group *g, const void *host, uint64_t nbytes);
int svo_group_pool_update(svo_group *g, const void *host_base, uint64_t start, uint64_t end);
int svo_group_pool_download(svo_group *g, void *host, uint64_t nbytes);
int svo_group_build_from_heightmap(svo_group *g, const uint16_t *height, const uint8_t *material, int n, uint64_t *out_nbytes);
int svo_group_set_camera(svo_group *g, const float pos[3], const float l1[3], const float l2[3], const float r1[3], const float r2[3]);
int svo_group_set_params(svo_group *g, int frame_number, int render_mode, int buffer_end, int use_beam, int bounces,
                         uint32_t mirror_mask, int spp);
int svo_group_set_pipeline(svo_group *g, int pipeline);
int svo_group_set_tuning(svo_group *g, int waves_per_cu, int round_threshold_sixteenths);
int svo_group_set_progressive(svo_group *g, int enabled);
int svo_group_set_sequence(svo_group *g, int nframes, int fresh);
/* the size of the whole frame; every member renders its stripes of it */
int svo_group_resize(svo_group *g, int width, int height);
/* as svo_ring_*: `slots` submissions in flight of up to frames_per_slot frames each, on every member at once.
 * exchange 0 (recommended): every member's slot forwards its chunk to the owner with a peer copy on its own stream behind the
 * launch (SDMA over xGMI; no CU slots needed).  exchange 1 (EXPERIMENTAL: has only ever run with one member -- no
 * multi-GPU node was available to the build): one RCCL send / receive pair per member inside ncclGroupStart / End; the
 * members' slot streams are then created with at least one CU per XCD left free for RCCL's kernels (svo_set_reserved_cus),
 * because persistent waves hold their CU slots for a whole launch. */
int svo_group_ring_create(svo_group *g, int slots, int frames_per_slot, int want_hits, int exchange);
int svo_group_ring_destroy(svo_group *g);
int svo_group_ring_submit(svo_group *g, int frame_number, int nframes, int *slot);
int svo_group_ring_submit_cams(svo_group *g, int nframes, const float *cams, const int *frame_numbers, int *slot);
/* the slot's frames are complete on every member AND have reached the owner */
int svo_group_ring_wait(svo_group *g, int slot);
/* gpu_ms: the slowest member's launch */
int svo_group_ring_query(svo_group *g, int slot, int *done, int *first_frame, int *nframes, float *gpu_ms);
/* whole frames in frame order (the stripes de-interleaved on the way out) */
int svo_group_ring_read_color(svo_group *g, int slot, int k, void *rgba8);
int svo_group_ring_read_depth(svo_group *g, int slot, int k, float *depth);
int svo_group_ring_read_hits(svo_group *g, int slot, int k, svo_hit *hits);
int svo_group_ring_read_pixel(svo_group *g, int slot, int k, int x, int y, void *rgba8, float *depth, svo_hit *hit);

/* ---- readback ------------------------------------------------------------------- */
/* replaces glGetTexImage of image 0 (rgba8; row 0 = p.y = 0, bytes R,G,B,A) and
 * image 1 (r32f depth) (Main.java:132-146, svotrace.comp:726-727) */
int svo_read_color(svo_ctx *ctx, void *rgba8);
int svo_read_depth(svo_ctx *ctx, float *depth);
int svo_read_hits(svo_ctx *ctx, svo_hit *hits);
/* the beam image of the last frame dispatched with use_beam (image unit 2 of the reference, Main.java:79-86):
 * ceil(H/4) rows of ceil(W/4) floats; only the block rows under the rows that frame rendered are defined */
int svo_read_beam(svo_ctx *ctx, float *beam);
/* one pixel of the three images: what Main.updateEarly actually needs from its full-frame glGetTexImage
 * (Main.java:132-146 reads the 8.3 MB depth image to pick depth[540][960], the crosshair).  Any of
 * rgba8 (4 bytes) / depth / hit may be NULL.  Library-owned or bound outputs; waits for the context's stream. */
int svo_read_pixel(svo_ctx *ctx, int x, int y, void *rgba8, float *depth, svo_hit *hit);
/* render into caller-owned device buffers (e.g. torch tensors that an RCCL all-gather then
 * reads in place): color = u32 rgba8 [rows][W], depth = f32, hits = svo_hit (may be NULL ->
 * hit records off).  Pixel (x, y) lands at element y*W + x, so the buffers must cover every
 * row this context renders.  Passing color == NULL returns to library-owned images. */
int svo_bind_outputs(svo_ctx *ctx, void *color, void *depth, void *hits);
/* device pointers of the three output images (W*H elements each), for an RCCL gather */
int svo_output_device_ptrs(svo_ctx *ctx, void **color, void **depth, void **hits);

#ifdef __cplusplus
}
#endif
#endif /* SVO_HIP_H */
