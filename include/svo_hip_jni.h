/*
 * svo_hip_jni.h -- JNI-typed shim over include/svo_hip.h, exported by libsvohip.so.
 *
 * The reference reaches its GPU through LWJGL, whose natives take only primitives and
 * `long` addresses of direct ByteBuffers (e.g. GL15C.glBufferData -> nglBufferData(target,
 * size, memAddress(data), usage)).  The shim keeps that style, so no function here needs
 * a JNIEnv call -- which is also why it compiles without jni.h (absent in the build image):
 * the first two parameters (JNIEnv*, jclass) are opaque pointers that are never touched.
 *
 * Java side: integration/java/src/engine/HipRenderer.java (static native methods with
 * these exact names; class src.engine.HipRenderer mirrors src.engine.Renderer,
 * /root/reference/src/engine/Renderer.java:43-165).
 */
#ifndef SVO_HIP_JNI_H
#define SVO_HIP_JNI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;

/* Renderer.getInstance() / GL context  ->  svo_create; returns the context handle, 0 on failure */
jlong Java_src_engine_HipRenderer_nCreate(void *env, void *cls, jint device);
jint Java_src_engine_HipRenderer_nDestroy(void *env, void *cls, jlong ctx);
/* Renderer.printGLErrors (Renderer.java:160-165): address of a NUL-terminated ASCII string */
jlong Java_src_engine_HipRenderer_nLastError(void *env, void *cls, jlong ctx);
/* Renderer.addSSBO / updateSSBO(full) (Renderer.java:123-134): addr = memAddress(buffer) */
jint Java_src_engine_HipRenderer_nPoolUpload(void *env, void *cls, jlong ctx, jlong addr, jlong nbytes);
/* Renderer.updateSSBO(bind, data, start, end) (Renderer.java:136-146) */
jint Java_src_engine_HipRenderer_nPoolUpdate(void *env, void *cls, jlong ctx, jlong base_addr, jlong start, jlong end);
/* Renderer.getSSBO (Renderer.java:148-150) */
jint Java_src_engine_HipRenderer_nPoolDownload(void *env, void *cls, jlong ctx, jlong addr, jlong nbytes);
/* glUniform3fv(8 | 1..4) (Main.java:269-273) */
jint Java_src_engine_HipRenderer_nSetCamera(void *env, void *cls, jlong ctx, jfloat px, jfloat py, jfloat pz,
                                            jfloat l1x, jfloat l1y, jfloat l1z, jfloat l2x, jfloat l2y, jfloat l2z,
                                            jfloat r1x, jfloat r1y, jfloat r1z, jfloat r2x, jfloat r2y, jfloat r2z);
/* glUniform1i(5 | 6 | 9 | 11) (Main.java:275-283) + dormant path options */
jint Java_src_engine_HipRenderer_nSetParams(void *env, void *cls, jlong ctx, jint frame_number, jint render_mode,
                                            jint buffer_end, jint use_beam, jint bounces, jint mirror_mask, jint spp);
/* texture allocation (Main.java:66-78) */
jint Java_src_engine_HipRenderer_nResize(void *env, void *cls, jlong ctx, jint width, jint height);
/* Renderer.dispatchCompute (Renderer.java:118-121) */
jint Java_src_engine_HipRenderer_nDispatch(void *env, void *cls, jlong ctx);
/* glGetTexImage (Main.java:132-146): addr = memAddress(direct buffer of W*H*4 / W*H*4 / W*H*16 bytes) */
jint Java_src_engine_HipRenderer_nReadColor(void *env, void *cls, jlong ctx, jlong addr);
jint Java_src_engine_HipRenderer_nReadDepth(void *env, void *cls, jlong ctx, jlong addr);
jint Java_src_engine_HipRenderer_nReadHits(void *env, void *cls, jlong ctx, jlong addr);
/* the crosshair pick of Main.java:132-146 without the full-frame readback: addresses of 4 / 4 / 16 bytes, 0 = skip */
jint Java_src_engine_HipRenderer_nReadPixel(void *env, void *cls, jlong ctx, jint x, jint y, jlong rgba_addr,
                                            jlong depth_addr, jlong hit_addr);
/* image unit 2 of the reference (Main.java:79-86): ceil(H/4) x ceil(W/4) floats */
jint Java_src_engine_HipRenderer_nReadBeam(void *env, void *cls, jlong ctx, jlong addr);
/* Octree.constructCompleteOctree (Octree.java:192-353) on the GPU: addresses of n*n u16 heights (voxels) and n*n u8
 * materials; returns the new pool's size in bytes (what Octree.memOffset would be), or a negative status */
jlong Java_src_engine_HipRenderer_nBuildFromHeightmap(void *env, void *cls, jlong ctx, jlong height_addr,
                                                      jlong material_addr, jint n);
/* OctreeThread.run / constructInnerOctree over a dense chunk (OctreeThread.java:20-23): address of n^3 voxel bytes,
 * the layout of the reference's voxelBuffer (x | y << log2 n | z << 2 log2 n); returns the pool size or a negative status */
jlong Java_src_engine_HipRenderer_nBuildFromVoxels(void *env, void *cls, jlong ctx, jlong voxels_addr, jint n);
/* throughput mode: n consecutive frames of the camera per dispatch into caller-owned device buffers (svo_set_batch,
 * svo_bind_outputs; device addresses as longs) */
jint Java_src_engine_HipRenderer_nSetBatch(void *env, void *cls, jlong ctx, jint nframes, jlong frame_stride);
jint Java_src_engine_HipRenderer_nBindOutputs(void *env, void *cls, jlong ctx, jlong color_dptr, jlong depth_dptr, jlong hits_dptr);
/* the commented-out cross-frame accumulation of svotrace.comp:712-719 */
jint Java_src_engine_HipRenderer_nSetProgressive(void *env, void *cls, jlong ctx, jint enabled);


/* ---- the rest of the C ABI's dispatch surface (include/svo_hip.h), same LWJGL style ---- */
jint Java_src_engine_HipRenderer_nDispatchAsync(void *env, void *cls, jlong ctx);
jint Java_src_engine_HipRenderer_nSync(void *env, void *cls, jlong ctx);
jint Java_src_engine_HipRenderer_nSetPick(void *env, void *cls, jlong ctx, jint x, jint y);       /* svo_set_pick */
jint Java_src_engine_HipRenderer_nSetOverlap(void *env, void *cls, jlong ctx, jint sets);        /* svo_set_overlap: 0, 1, 2 .. 8 */
/* svo_pick_info: read-backs answered from the mail (or a negative status); two ints (x, y of the pick in force) at xy_addr and
 * the read-backs that waited for their frame at waited_addr (a long) unless 0 */
jlong Java_src_engine_HipRenderer_nPickInfo(void *env, void *cls, jlong ctx, jlong xy_addr, jlong waited_addr);
/* hipStream_t as a long; 0 = the library's own stream */
jint Java_src_engine_HipRenderer_nSetStream(void *env, void *cls, jlong ctx, jlong hip_stream);
jint Java_src_engine_HipRenderer_nSetPipeline(void *env, void *cls, jlong ctx, jint pipeline);
jint Java_src_engine_HipRenderer_nSetTuning(void *env, void *cls, jlong ctx, jint waves_per_cu, jint round_threshold_sixteenths);
/* svo_launch_info: persistent waves of the last pipeline-1 launch (what the automatic launch shape resolved to); waves per
 * CU through the address (0 = not wanted) */
jint Java_src_engine_HipRenderer_nLaunchInfo(void *env, void *cls, jlong ctx, jlong waves_per_cu_addr);
jint Java_src_engine_HipRenderer_nSetDerived(void *env, void *cls, jlong ctx, jint mode);
jint Java_src_engine_HipRenderer_nSetHitRecords(void *env, void *cls, jlong ctx, jint enabled);
jint Java_src_engine_HipRenderer_nSetRows(void *env, void *cls, jlong ctx, jint y0, jint y1);
jint Java_src_engine_HipRenderer_nSetStripes(void *env, void *cls, jlong ctx, jint first_tile_row, jint tile_row_step,
                                             jint n_tile_rows, jint out_row0);
/* svo_count_frame / svo_get_stats: address of a 64-byte svo_stats (six u64, f32 last_dispatch_ms, i32 device) */
jint Java_src_engine_HipRenderer_nCountFrame(void *env, void *cls, jlong ctx, jlong stats_addr);
jint Java_src_engine_HipRenderer_nGetStats(void *env, void *cls, jlong ctx, jlong stats_addr);
/* svo_derived_info: descriptors of the table, or a negative status; *walkable_addr (4 bytes, may be 0) = 1 if walked */
jlong Java_src_engine_HipRenderer_nDerivedInfo(void *env, void *cls, jlong ctx, jlong walkable_addr);
/* svo_derived_refresh_info: ranged updates (Renderer.updateSSBO) the descriptor table followed without a rebuild so far;
 * states recomputed / descriptors appended by the last one through the two addresses (0 = not wanted) */
jlong Java_src_engine_HipRenderer_nDerivedRefreshInfo(void *env, void *cls, jlong ctx, jlong states_addr, jlong added_addr);
/* ---- frames in flight (svo_ring_*): what a Java host cannot own -- streams, device buffers -- stays in the library */
jint Java_src_engine_HipRenderer_nRingCreate(void *env, void *cls, jlong ctx, jint slots, jint frames_per_slot, jint want_hits);
jint Java_src_engine_HipRenderer_nRingDestroy(void *env, void *cls, jlong ctx);
/* returns the slot (>= 0) the frames went to, or a negative status */
jint Java_src_engine_HipRenderer_nRingSubmit(void *env, void *cls, jlong ctx, jint frame_number, jint nframes);
/* a submission whose frames carry their own camera (15 floats each at cams_addr) and frameNumber (ints at
 * frame_numbers_addr): svo_ring_submit_cams; returns the slot or a negative status */
jint Java_src_engine_HipRenderer_nRingSubmitCams(void *env, void *cls, jlong ctx, jint nframes, jlong cams_addr, jlong frame_numbers_addr);
/* svo_set_sequence: nframes frames of the cross-frame accumulation per dispatch (BASELINE config 5 = 64) */
jint Java_src_engine_HipRenderer_nSetSequence(void *env, void *cls, jlong ctx, jint nframes, jint fresh);
jint Java_src_engine_HipRenderer_nRingWait(void *env, void *cls, jlong ctx, jint slot);
/* 1 = complete, 0 = still running, negative = status; *ms_addr (4 bytes, may be 0) = GPU milliseconds of the slot */
jint Java_src_engine_HipRenderer_nRingDone(void *env, void *cls, jlong ctx, jint slot, jlong ms_addr);
jint Java_src_engine_HipRenderer_nRingReadColor(void *env, void *cls, jlong ctx, jint slot, jint k, jlong addr);
jint Java_src_engine_HipRenderer_nRingReadDepth(void *env, void *cls, jlong ctx, jint slot, jint k, jlong addr);
jint Java_src_engine_HipRenderer_nRingReadHits(void *env, void *cls, jlong ctx, jint slot, jint k, jlong addr);
jint Java_src_engine_HipRenderer_nRingReadPixel(void *env, void *cls, jlong ctx, jint slot, jint k, jint x, jint y,
                                                jlong rgba_addr, jlong depth_addr, jlong hit_addr);
jint Java_src_engine_HipRenderer_nRingBindSlot(void *env, void *cls, jlong ctx, jint slot, jlong color_dptr, jlong depth_dptr,
                                               jlong hits_dptr, jlong frame_stride);


/* ---- N GPUs behind the boundary: svo_group_* (include/svo_hip.h).  nGroupCreate returns the handle (the group's address) or
 * a negative status; devices_addr = n ints; nGroupSetCamera takes the 15 floats of Camera.getUniform(); nGroupLastError
 * returns the address of a NUL-terminated string (MemoryUtil.memUTF8); nGroupMember a context handle for the per-GPU natives. */
jlong Java_src_engine_HipRenderer_nGroupCreate(void *env, void *cls, jlong devices_addr, jint n);
jint Java_src_engine_HipRenderer_nGroupDestroy(void *env, void *cls, jlong g);
jlong Java_src_engine_HipRenderer_nGroupLastError(void *env, void *cls, jlong g);
jlong Java_src_engine_HipRenderer_nGroupMember(void *env, void *cls, jlong g, jint i);
jint Java_src_engine_HipRenderer_nGroupPoolUpload(void *env, void *cls, jlong g, jlong addr, jlong nbytes);
jint Java_src_engine_HipRenderer_nGroupPoolUpdate(void *env, void *cls, jlong g, jlong base_addr, jlong start, jlong end);
jint Java_src_engine_HipRenderer_nGroupSetCamera(void *env, void *cls, jlong g, jlong cam15_addr);
jint Java_src_engine_HipRenderer_nGroupSetParams(void *env, void *cls, jlong g, jint frame_number, jint render_mode, jint buffer_end, jint use_beam, jint bounces, jint mirror_mask, jint spp);
jint Java_src_engine_HipRenderer_nGroupSetTuning(void *env, void *cls, jlong g, jint waves_per_cu, jint thresh);
jint Java_src_engine_HipRenderer_nGroupSetProgressive(void *env, void *cls, jlong g, jint enabled);
jint Java_src_engine_HipRenderer_nGroupSetSequence(void *env, void *cls, jlong g, jint nframes, jint fresh);
jint Java_src_engine_HipRenderer_nGroupResize(void *env, void *cls, jlong g, jint w, jint h);
jint Java_src_engine_HipRenderer_nGroupRingCreate(void *env, void *cls, jlong g, jint slots, jint frames_per_slot, jint want_hits, jint exchange);
jint Java_src_engine_HipRenderer_nGroupRingDestroy(void *env, void *cls, jlong g);
jint Java_src_engine_HipRenderer_nGroupRingSubmit(void *env, void *cls, jlong g, jint frame_number, jint nframes);
jint Java_src_engine_HipRenderer_nGroupRingSubmitCams(void *env, void *cls, jlong g, jint nframes, jlong cams_addr, jlong frame_numbers_addr);
jint Java_src_engine_HipRenderer_nGroupRingWait(void *env, void *cls, jlong g, jint slot);
jint Java_src_engine_HipRenderer_nGroupRingDone(void *env, void *cls, jlong g, jint slot, jlong ms_addr);
jint Java_src_engine_HipRenderer_nGroupRingReadColor(void *env, void *cls, jlong g, jint slot, jint k, jlong addr);
jint Java_src_engine_HipRenderer_nGroupRingReadDepth(void *env, void *cls, jlong g, jint slot, jint k, jlong addr);
jint Java_src_engine_HipRenderer_nGroupRingReadPixel(void *env, void *cls, jlong g, jint slot, jint k, jint x, jint y, jlong rgba_addr, jlong depth_addr, jlong hit_addr);

#ifdef __cplusplus
}
#endif
#endif
