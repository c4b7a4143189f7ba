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

#ifdef __cplusplus
}
#endif
#endif
