// svo_jni.cpp -- JNI-typed shim (include/svo_hip_jni.h): forwards to the C ABI, nothing else.
#include "../../include/svo_hip_jni.h"
#include "../../include/svo_hip.h"

extern "C" {

jlong Java_src_engine_HipRenderer_nCreate(void *, void *, jint device) {
  svo_ctx *c = nullptr;
  if (svo_create(device, &c) != SVO_OK) return 0;
  return (jlong)(intptr_t)c;
}
jint Java_src_engine_HipRenderer_nDestroy(void *, void *, jlong ctx) { return svo_destroy((svo_ctx *)(intptr_t)ctx); }
jlong Java_src_engine_HipRenderer_nLastError(void *, void *, jlong ctx) {
  return (jlong)(intptr_t)svo_last_error((svo_ctx *)(intptr_t)ctx);
}
jint Java_src_engine_HipRenderer_nPoolUpload(void *, void *, jlong ctx, jlong addr, jlong nbytes) {
  return svo_pool_upload((svo_ctx *)(intptr_t)ctx, (const void *)(intptr_t)addr, (uint64_t)nbytes);
}
jint Java_src_engine_HipRenderer_nPoolUpdate(void *, void *, jlong ctx, jlong base, jlong start, jlong end) {
  if (start < 0 || end < 0) return SVO_E_INVALID;
  return svo_pool_update((svo_ctx *)(intptr_t)ctx, (const void *)(intptr_t)base, (uint64_t)start, (uint64_t)end);
}
jint Java_src_engine_HipRenderer_nPoolDownload(void *, void *, jlong ctx, jlong addr, jlong nbytes) {
  return svo_pool_download((svo_ctx *)(intptr_t)ctx, (void *)(intptr_t)addr, (uint64_t)nbytes);
}
jint Java_src_engine_HipRenderer_nSetCamera(void *, void *, jlong ctx, jfloat px, jfloat py, jfloat pz, jfloat l1x,
                                            jfloat l1y, jfloat l1z, jfloat l2x, jfloat l2y, jfloat l2z, jfloat r1x,
                                            jfloat r1y, jfloat r1z, jfloat r2x, jfloat r2y, jfloat r2z) {
  const float p[3] = {px, py, pz}, l1[3] = {l1x, l1y, l1z}, l2[3] = {l2x, l2y, l2z}, r1[3] = {r1x, r1y, r1z},
              r2[3] = {r2x, r2y, r2z};
  return svo_set_camera((svo_ctx *)(intptr_t)ctx, p, l1, l2, r1, r2);
}
jint Java_src_engine_HipRenderer_nSetParams(void *, void *, jlong ctx, jint frame_number, jint render_mode,
                                            jint buffer_end, jint use_beam, jint bounces, jint mirror_mask, jint spp) {
  return svo_set_params((svo_ctx *)(intptr_t)ctx, frame_number, render_mode, buffer_end, use_beam, bounces,
                        (uint32_t)mirror_mask, spp);
}
jint Java_src_engine_HipRenderer_nResize(void *, void *, jlong ctx, jint w, jint h) {
  return svo_resize((svo_ctx *)(intptr_t)ctx, w, h);
}
jint Java_src_engine_HipRenderer_nDispatch(void *, void *, jlong ctx) { return svo_dispatch((svo_ctx *)(intptr_t)ctx); }
jint Java_src_engine_HipRenderer_nReadColor(void *, void *, jlong ctx, jlong addr) {
  return svo_read_color((svo_ctx *)(intptr_t)ctx, (void *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nReadDepth(void *, void *, jlong ctx, jlong addr) {
  return svo_read_depth((svo_ctx *)(intptr_t)ctx, (float *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nReadHits(void *, void *, jlong ctx, jlong addr) {
  return svo_read_hits((svo_ctx *)(intptr_t)ctx, (svo_hit *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nReadPixel(void *, void *, jlong ctx, jint x, jint y, jlong rgba_addr, jlong depth_addr,
                                            jlong hit_addr) {
  return svo_read_pixel((svo_ctx *)(intptr_t)ctx, x, y, (void *)(intptr_t)rgba_addr, (float *)(intptr_t)depth_addr,
                        (svo_hit *)(intptr_t)hit_addr);
}
jint Java_src_engine_HipRenderer_nReadBeam(void *, void *, jlong ctx, jlong addr) {
  return svo_read_beam((svo_ctx *)(intptr_t)ctx, (float *)(intptr_t)addr);
}
jlong Java_src_engine_HipRenderer_nBuildFromHeightmap(void *, void *, jlong ctx, jlong height_addr, jlong material_addr, jint n) {
  uint64_t nbytes = 0;
  const int rc = svo_build_from_heightmap((svo_ctx *)(intptr_t)ctx, (const uint16_t *)(intptr_t)height_addr,
                                          (const uint8_t *)(intptr_t)material_addr, n, &nbytes);
  return rc == SVO_OK ? (jlong)nbytes : (jlong)rc;   // the pool's size (memOffset), or a negative status
}
jlong Java_src_engine_HipRenderer_nBuildFromVoxels(void *, void *, jlong ctx, jlong voxels_addr, jint n) {
  uint64_t nbytes = 0;
  const int rc = svo_build_from_voxels((svo_ctx *)(intptr_t)ctx, (const uint8_t *)(intptr_t)voxels_addr, n, &nbytes);
  return rc == SVO_OK ? (jlong)nbytes : (jlong)rc;
}
jint Java_src_engine_HipRenderer_nSetBatch(void *, void *, jlong ctx, jint nframes, jlong frame_stride) {
  if (frame_stride < 0) return SVO_E_INVALID;
  return svo_set_batch((svo_ctx *)(intptr_t)ctx, nframes, (uint64_t)frame_stride);
}
jint Java_src_engine_HipRenderer_nBindOutputs(void *, void *, jlong ctx, jlong color_dptr, jlong depth_dptr, jlong hits_dptr) {
  return svo_bind_outputs((svo_ctx *)(intptr_t)ctx, (void *)(intptr_t)color_dptr, (void *)(intptr_t)depth_dptr,
                          (void *)(intptr_t)hits_dptr);
}
jint Java_src_engine_HipRenderer_nSetProgressive(void *, void *, jlong ctx, jint enabled) {
  return svo_set_progressive((svo_ctx *)(intptr_t)ctx, enabled);
}


#define CTX(x) ((svo_ctx *)(intptr_t)(x))
jint Java_src_engine_HipRenderer_nDispatchAsync(void *, void *, jlong ctx) { return svo_dispatch_async(CTX(ctx)); }
jint Java_src_engine_HipRenderer_nSync(void *, void *, jlong ctx) { return svo_sync(CTX(ctx)); }
jint Java_src_engine_HipRenderer_nSetPick(void *, void *, jlong ctx, jint x, jint y) { return svo_set_pick(CTX(ctx), x, y); }
jint Java_src_engine_HipRenderer_nSetOverlap(void *, void *, jlong ctx, jint sets) { return svo_set_overlap(CTX(ctx), sets); }
jlong Java_src_engine_HipRenderer_nPickInfo(void *, void *, jlong ctx, jlong xy_addr, jlong waited_addr) {
  int xy[2] = {-1, -1};
  uint64_t from_mail = 0, waited = 0;
  const int rc = svo_pick_info(CTX(ctx), &xy[0], &xy[1], &from_mail, &waited);
  if (rc != SVO_OK) return (jlong)rc;
  if (xy_addr) { ((jint *)(intptr_t)xy_addr)[0] = xy[0]; ((jint *)(intptr_t)xy_addr)[1] = xy[1]; }
  if (waited_addr) *(jlong *)(intptr_t)waited_addr = (jlong)waited;
  return (jlong)from_mail;
}
jint Java_src_engine_HipRenderer_nSetStream(void *, void *, jlong ctx, jlong hip_stream) {
  return svo_set_stream(CTX(ctx), (void *)(intptr_t)hip_stream);
}
jint Java_src_engine_HipRenderer_nSetPipeline(void *, void *, jlong ctx, jint pipeline) { return svo_set_pipeline(CTX(ctx), pipeline); }
jint Java_src_engine_HipRenderer_nSetTuning(void *, void *, jlong ctx, jint waves_per_cu, jint thresh) {
  return svo_set_tuning(CTX(ctx), waves_per_cu, thresh);
}
jint Java_src_engine_HipRenderer_nLaunchInfo(void *, void *, jlong ctx, jlong waves_per_cu_addr) {
  int waves = 0;
  const int rc = svo_launch_info(CTX(ctx), &waves, (int *)(intptr_t)waves_per_cu_addr, nullptr);
  return rc == SVO_OK ? (jint)waves : (jint)rc;
}
jint Java_src_engine_HipRenderer_nSetDerived(void *, void *, jlong ctx, jint mode) { return svo_set_derived(CTX(ctx), mode); }
jint Java_src_engine_HipRenderer_nSetHitRecords(void *, void *, jlong ctx, jint enabled) { return svo_set_hit_records(CTX(ctx), enabled); }
jint Java_src_engine_HipRenderer_nSetRows(void *, void *, jlong ctx, jint y0, jint y1) { return svo_set_rows(CTX(ctx), y0, y1); }
jint Java_src_engine_HipRenderer_nSetStripes(void *, void *, jlong ctx, jint first_tile_row, jint tile_row_step, jint n_tile_rows,
                                             jint out_row0) {
  return svo_set_stripes(CTX(ctx), first_tile_row, tile_row_step, n_tile_rows, out_row0);
}
jint Java_src_engine_HipRenderer_nCountFrame(void *, void *, jlong ctx, jlong stats_addr) {
  return svo_count_frame(CTX(ctx), (svo_stats *)(intptr_t)stats_addr);
}
jint Java_src_engine_HipRenderer_nGetStats(void *, void *, jlong ctx, jlong stats_addr) {
  return svo_get_stats(CTX(ctx), (svo_stats *)(intptr_t)stats_addr);
}
jlong Java_src_engine_HipRenderer_nDerivedInfo(void *, void *, jlong ctx, jlong walkable_addr) {
  uint64_t n = 0;
  const int rc = svo_derived_info(CTX(ctx), &n, nullptr, (int *)(intptr_t)walkable_addr, nullptr);
  return rc == SVO_OK ? (jlong)n : (jlong)rc;
}
jlong Java_src_engine_HipRenderer_nDerivedRefreshInfo(void *, void *, jlong ctx, jlong states_addr, jlong added_addr) {
  uint64_t n = 0;
  const int rc = svo_derived_refresh_info(CTX(ctx), &n, (uint64_t *)(intptr_t)states_addr, (uint64_t *)(intptr_t)added_addr, nullptr);
  return rc == SVO_OK ? (jlong)n : (jlong)rc;
}
jint Java_src_engine_HipRenderer_nRingCreate(void *, void *, jlong ctx, jint slots, jint frames_per_slot, jint want_hits) {
  return svo_ring_create(CTX(ctx), slots, frames_per_slot, want_hits);
}
jint Java_src_engine_HipRenderer_nRingDestroy(void *, void *, jlong ctx) { return svo_ring_destroy(CTX(ctx)); }
jint Java_src_engine_HipRenderer_nRingSubmit(void *, void *, jlong ctx, jint frame_number, jint nframes) {
  int slot = -1;
  const int rc = svo_ring_submit(CTX(ctx), frame_number, nframes, &slot);
  return rc == SVO_OK ? slot : rc;
}
jint Java_src_engine_HipRenderer_nRingSubmitCams(void *, void *, jlong ctx, jint nframes, jlong cams_addr, jlong frame_numbers_addr) {
  int slot = -1;
  const int rc = svo_ring_submit_cams(CTX(ctx), nframes, (const float *)(intptr_t)cams_addr, (const int *)(intptr_t)frame_numbers_addr, &slot);
  return rc == SVO_OK ? slot : rc;
}
jint Java_src_engine_HipRenderer_nSetSequence(void *, void *, jlong ctx, jint nframes, jint fresh) {
  return svo_set_sequence(CTX(ctx), nframes, fresh);
}
jint Java_src_engine_HipRenderer_nRingWait(void *, void *, jlong ctx, jint slot) { return svo_ring_wait(CTX(ctx), slot); }
jint Java_src_engine_HipRenderer_nRingDone(void *, void *, jlong ctx, jint slot, jlong ms_addr) {
  int done = 0;
  const int rc = svo_ring_query(CTX(ctx), slot, &done, nullptr, nullptr, (float *)(intptr_t)ms_addr);
  return rc == SVO_OK ? done : rc;
}
jint Java_src_engine_HipRenderer_nRingReadColor(void *, void *, jlong ctx, jint slot, jint k, jlong addr) {
  return svo_ring_read_color(CTX(ctx), slot, k, (void *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nRingReadDepth(void *, void *, jlong ctx, jint slot, jint k, jlong addr) {
  return svo_ring_read_depth(CTX(ctx), slot, k, (float *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nRingReadHits(void *, void *, jlong ctx, jint slot, jint k, jlong addr) {
  return svo_ring_read_hits(CTX(ctx), slot, k, (svo_hit *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nRingReadPixel(void *, void *, jlong ctx, jint slot, jint k, jint x, jint y, jlong rgba_addr,
                                                jlong depth_addr, jlong hit_addr) {
  return svo_ring_read_pixel(CTX(ctx), slot, k, x, y, (void *)(intptr_t)rgba_addr, (float *)(intptr_t)depth_addr,
                             (svo_hit *)(intptr_t)hit_addr);
}
jint Java_src_engine_HipRenderer_nRingBindSlot(void *, void *, jlong ctx, jint slot, jlong color_dptr, jlong depth_dptr,
                                               jlong hits_dptr, jlong frame_stride) {
  if (frame_stride < 0) return SVO_E_INVALID;
  return svo_ring_bind_slot(CTX(ctx), slot, (void *)(intptr_t)color_dptr, (void *)(intptr_t)depth_dptr, (void *)(intptr_t)hits_dptr,
                            (uint64_t)frame_stride);
}
#undef CTX

// ---- N GPUs behind the boundary (svo_group_*): the handle is the group's address
#define GRP(x) ((svo_group *)(intptr_t)(x))
jlong Java_src_engine_HipRenderer_nGroupCreate(void *, void *, jlong devices_addr, jint n) {
  svo_group *g = nullptr;
  const int rc = svo_group_create((const int *)(intptr_t)devices_addr, n, &g);
  return rc == SVO_OK ? (jlong)(intptr_t)g : (jlong)rc;   // the handle, or a negative status
}
jint Java_src_engine_HipRenderer_nGroupDestroy(void *, void *, jlong g) { return svo_group_destroy(GRP(g)); }
jlong Java_src_engine_HipRenderer_nGroupLastError(void *, void *, jlong g) { return (jlong)(intptr_t)svo_group_last_error(GRP(g)); }
jlong Java_src_engine_HipRenderer_nGroupMember(void *, void *, jlong g, jint i) { return (jlong)(intptr_t)svo_group_member(GRP(g), i); }
jint Java_src_engine_HipRenderer_nGroupPoolUpload(void *, void *, jlong g, jlong addr, jlong nbytes) {
  return nbytes < 0 ? SVO_E_INVALID : svo_group_pool_upload(GRP(g), (const void *)(intptr_t)addr, (uint64_t)nbytes);
}
jint Java_src_engine_HipRenderer_nGroupPoolUpdate(void *, void *, jlong g, jlong base_addr, jlong start, jlong end) {
  return (start < 0 || end < 0) ? SVO_E_INVALID : svo_group_pool_update(GRP(g), (const void *)(intptr_t)base_addr, (uint64_t)start, (uint64_t)end);
}
jint Java_src_engine_HipRenderer_nGroupSetCamera(void *, void *, jlong g, jlong cam15_addr) {
  const float *c = (const float *)(intptr_t)cam15_addr;
  return c ? svo_group_set_camera(GRP(g), c, c + 3, c + 6, c + 9, c + 12) : SVO_E_INVALID;
}
jint Java_src_engine_HipRenderer_nGroupSetParams(void *, void *, jlong g, jint frame_number, jint render_mode, jint buffer_end,
                                                 jint use_beam, jint bounces, jint mirror_mask, jint spp) {
  return svo_group_set_params(GRP(g), frame_number, render_mode, buffer_end, use_beam, bounces, (uint32_t)mirror_mask, spp);
}
jint Java_src_engine_HipRenderer_nGroupSetTuning(void *, void *, jlong g, jint waves_per_cu, jint thresh) {
  return svo_group_set_tuning(GRP(g), waves_per_cu, thresh);
}
jint Java_src_engine_HipRenderer_nGroupSetProgressive(void *, void *, jlong g, jint enabled) { return svo_group_set_progressive(GRP(g), enabled); }
jint Java_src_engine_HipRenderer_nGroupSetSequence(void *, void *, jlong g, jint nframes, jint fresh) { return svo_group_set_sequence(GRP(g), nframes, fresh); }
jint Java_src_engine_HipRenderer_nGroupResize(void *, void *, jlong g, jint w, jint h) { return svo_group_resize(GRP(g), w, h); }
jint Java_src_engine_HipRenderer_nGroupRingCreate(void *, void *, jlong g, jint slots, jint frames_per_slot, jint want_hits, jint exchange) {
  return svo_group_ring_create(GRP(g), slots, frames_per_slot, want_hits, exchange);
}
jint Java_src_engine_HipRenderer_nGroupRingDestroy(void *, void *, jlong g) { return svo_group_ring_destroy(GRP(g)); }
jint Java_src_engine_HipRenderer_nGroupRingSubmit(void *, void *, jlong g, jint frame_number, jint nframes) {
  int slot = -1;
  const int rc = svo_group_ring_submit(GRP(g), frame_number, nframes, &slot);
  return rc == SVO_OK ? slot : rc;
}
jint Java_src_engine_HipRenderer_nGroupRingSubmitCams(void *, void *, jlong g, jint nframes, jlong cams_addr, jlong frame_numbers_addr) {
  int slot = -1;
  const int rc = svo_group_ring_submit_cams(GRP(g), nframes, (const float *)(intptr_t)cams_addr, (const int *)(intptr_t)frame_numbers_addr, &slot);
  return rc == SVO_OK ? slot : rc;
}
jint Java_src_engine_HipRenderer_nGroupRingWait(void *, void *, jlong g, jint slot) { return svo_group_ring_wait(GRP(g), slot); }
jint Java_src_engine_HipRenderer_nGroupRingDone(void *, void *, jlong g, jint slot, jlong ms_addr) {
  int done = 0;
  const int rc = svo_group_ring_query(GRP(g), slot, &done, nullptr, nullptr, (float *)(intptr_t)ms_addr);
  return rc == SVO_OK ? done : rc;
}
jint Java_src_engine_HipRenderer_nGroupRingReadColor(void *, void *, jlong g, jint slot, jint k, jlong addr) {
  return svo_group_ring_read_color(GRP(g), slot, k, (void *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nGroupRingReadDepth(void *, void *, jlong g, jint slot, jint k, jlong addr) {
  return svo_group_ring_read_depth(GRP(g), slot, k, (float *)(intptr_t)addr);
}
jint Java_src_engine_HipRenderer_nGroupRingReadPixel(void *, void *, jlong g, jint slot, jint k, jint x, jint y, jlong rgba_addr,
                                                     jlong depth_addr, jlong hit_addr) {
  return svo_group_ring_read_pixel(GRP(g), slot, k, x, y, (void *)(intptr_t)rgba_addr, (float *)(intptr_t)depth_addr,
                                   (svo_hit *)(intptr_t)hit_addr);
}
#undef GRP

}  // extern "C"
