package src.engine;

import java.nio.ByteBuffer;
import java.util.ArrayList;

import org.lwjgl.system.MemoryUtil;

/**
 * Drop-in twin of {@link Renderer} for the SVO trace path: same method names and argument
 * meaning, the work goes to libsvohip.so (MI355X) instead of the OpenGL compute shader.
 *
 * NOT COMPILED in the build image (no JDK there); shipped as the binding a maintainer adds.
 * Natives are LWJGL-style: primitives and memAddress(...) longs only, so the C side
 * (include/svo_hip_jni.h) needs no JNIEnv calls.
 *
 * Usage in Main.java: replace `Renderer.getInstance()` by `HipRenderer.getInstance()`, replace
 * the raw glUniform* calls at Main.java:269-283 by setCamera(...) / setUniformInteger(...),
 * and read the images back with readFramebuffer / readDepth instead of glGetTexImage.
 */
public class HipRenderer {

  static {
    System.loadLibrary("svohip");
  }

  private static final HipRenderer instance = new HipRenderer();
  private final ArrayList<Shader> shaders = new ArrayList<Shader>();
  private long ctx;
  private int width, height;
  private int frameNumber = 1, renderMode = 2, bufferEnd = 0, useBeam = 0;
  private int bounces = 2, mirrorMask = 0, spp = 1;

  public class Shader {
    String name;
    int computeProgram;
    int computeProgramShader;

    Shader(String name, int computeProgram, int computeProgramShader) {
      this.name = name;
      this.computeProgram = computeProgram;
      this.computeProgramShader = computeProgramShader;
    }
  }

  private HipRenderer() {
    ctx = nCreate(0);
    if (ctx == 0)
      System.out.println("HIP ERR: no MI355X visible");
  }

  public static HipRenderer getInstance() {
    return instance;
  }

  /** Renderer.addShader: "svotrace" binds to the precompiled HIP path; other shaders are inert. */
  public Shader addShader(String name, String path) {
    int id = (path.contains("svotrace") || name.equals("svotrace")) ? 1 : 0;
    Shader shader = new Shader(name, id, id);
    shaders.add(shader);
    return shader;
  }

  /** glUniform1i: 5 frameNumber, 6 renderMode, 9 bufferEnd, 11 useBeamOptimization. */
  public void setUniformInteger(int location, int value) {
    if (location == 5) frameNumber = value;
    else if (location == 6) renderMode = value;
    else if (location == 9) bufferEnd = value;
    else if (location == 11) useBeam = value;
  }

  /** glUniform3fv(8, pos), (1..4, l1, l2, r1, r2): pass Camera.getUniform(). */
  public void setCamera(float[][] u) {
    check(nSetCamera(ctx, u[0][0], u[0][1], u[0][2], u[1][0], u[1][1], u[1][2], u[2][0], u[2][1], u[2][2],
        u[3][0], u[3][1], u[3][2], u[4][0], u[4][1], u[4][2]));
  }

  public void setImageSize(int w, int h) {
    width = w;
    height = h;
  }

  public void useProgram(Shader shader) {
  }

  /**
   * glDispatchCompute + glMemoryBarrier (Renderer.java:118-121): both GL calls return at once, and so does this -- the frame
   * is enqueued (svo_dispatch_async).  The wait is where the reference has it: the next read-back (readDepthPixel /
   * readFramebuffer / readDepth / readHits = next frame's glGetTexImage, Main.java:132-146) waits for the frame, as do
   * addSSBO / updateSSBO / getSSBO; the quad draw, ImGui and input of Main's loop overlap the trace as they do under GL.
   */
  public void dispatchCompute(Shader shader, int numGroupsX, int numGroupsY, int numGroupsZ) {
    if (prepareDispatch(shader, numGroupsX, numGroupsY))
      check(nDispatchAsync(ctx));
  }

  /** The same, but returns when the frame is complete (svo_dispatch); the stats' last_dispatch_ms is then its GPU time. */
  public void dispatchComputeAndWait(Shader shader, int numGroupsX, int numGroupsY, int numGroupsZ) {
    if (prepareDispatch(shader, numGroupsX, numGroupsY))
      check(nDispatch(ctx));
  }

  private boolean prepareDispatch(Shader shader, int numGroupsX, int numGroupsY) {
    if (shader == null || shader.computeProgram != 1)
      return false;
    if (width == 0) {
      width = numGroupsX * Constants.COMPUTE_GROUP_SIZE;
      height = numGroupsY * Constants.COMPUTE_GROUP_SIZE;
    }
    check(nResize(ctx, width, height));
    check(nSetParams(ctx, frameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    return true;
  }

  public void addSSBO(int bindIndex, ByteBuffer data) {
    if (bindIndex != 7)
      return;
    check(nPoolUpload(ctx, MemoryUtil.memAddress(data), data.remaining()));
  }

  public void updateSSBO(int bindIndex, ByteBuffer data) {
    addSSBO(bindIndex, data);
  }

  public void updateSSBO(int bindIndex, ByteBuffer data, int start, int end) {
    if (start >= end) {
      System.out.println("Update SSBO error: Invalid parameters.");
      return;
    }
    check(nPoolUpdate(ctx, MemoryUtil.memAddress0(data), start, end));
  }

  public void getSSBO(ByteBuffer buffer) {
    check(nPoolDownload(ctx, MemoryUtil.memAddress(buffer), buffer.remaining()));
  }

  public Shader getShaderByName(String name) {
    for (Shader shader : shaders) {
      if (shader.name.equals(name))
        return shader;
    }
    return null;
  }

  public void printGLErrors() {
    long s = nLastError(ctx);
    if (s != 0) {
      String msg = MemoryUtil.memASCII(s);
      if (!msg.isEmpty())
        System.out.println("HIP ERR: " + msg);
    }
  }

  /** glGetTexImage of image 0: W*H*4 bytes, row 0 = bottom of the screen, RGBA. */
  public void readFramebuffer(ByteBuffer rgba8) {
    check(nReadColor(ctx, MemoryUtil.memAddress(rgba8)));
  }

  /** glGetTexImage of image 1: W*H floats. */
  public void readDepth(ByteBuffer depth) {
    check(nReadDepth(ctx, MemoryUtil.memAddress(depth)));
  }

  /** 16 bytes per pixel: u32 pointer, u16 rawNormal, u8 value, u8 level, u32 iter, f32 t. */
  public void readHits(ByteBuffer hits) {
    check(nReadHits(ctx, MemoryUtil.memAddress(hits)));
  }

  /**
   * The crosshair pick of Main.updateEarly (Main.java:132-146) without the full-frame readback:
   * depth of one pixel.  Main reads depth[540][960] of the previous frame every frame.
   */
  public float readDepthPixel(int x, int y) {
    ByteBuffer one = MemoryUtil.memAlloc(4);
    check(nReadPixel(ctx, x, y, 0L, MemoryUtil.memAddress(one), 0L));
    float d = one.getFloat(0);
    MemoryUtil.memFree(one);
    return d;
  }

  /**
   * Octree.constructCompleteOctree on the GPU: heights (u16, voxels) and surface materials (u8) of an n x n world,
   * indexed [z * n + x].  The built pool becomes the bound SSBO; returns its size (the octree's memOffset) or a
   * negative status.  getSSBO(buf) copies it into the Java-side ByteBuffer if the host needs it (SDF edits).
   */
  public long buildFromHeightmap(ByteBuffer heights, ByteBuffer materials, int n) {
    long r = nBuildFromHeightmap(ctx, MemoryUtil.memAddress(heights), MemoryUtil.memAddress(materials), n);
    if (r < 0)
      printGLErrors();
    return r;
  }

  /** OctreeThread / constructInnerOctree over one dense chunk (the reference's voxelBuffer layout), on the GPU. */
  public long buildFromVoxels(ByteBuffer voxels, int n) {
    long r = nBuildFromVoxels(ctx, MemoryUtil.memAddress(voxels), n);
    if (r < 0)
      printGLErrors();
    return r;
  }

  /** The commented-out cross-frame accumulation of svotrace.comp:712-719 (default off = the live shader). */
  public void setProgressive(boolean on) {
    check(nSetProgressive(ctx, on ? 1 : 0));
  }

  /** Dormant shader features (svotrace.comp:444, 500-504, 668-670); defaults = live behaviour. */
  public void setPathOptions(int bounces, int mirrorMask, int spp) {
    this.bounces = bounces;
    this.mirrorMask = mirrorMask;
    this.spp = spp;
  }

  // ---- frames in flight (svo_ring_*): the throughput mode bench.py times, from Java ---------------------------
  // Main.updateEarly renders a frame and reads the crosshair back before the next one (Main.java:132-146, 257-289).
  // A host that does not need every frame's result at once (offline renders, progressive GI with a static camera,
  // several viewports) can keep `slots` launches of `framesPerSlot` consecutive frames in flight instead: the
  // library owns the streams and the device images.

  /**
   * Create (or re-create) the ring; call after the first dispatchCompute / setImageSize has fixed the image size.  With more
   * than one slot the submissions run in the launch shape bench.py's headline is measured on (10 persistent waves per CU and
   * launch, a round once at most 9/16 of the lanes are still traversing) by themselves -- no setTuning call needed; setTuning with a positive wave count overrides it.
   */
  public void createFrameRing(int slots, int framesPerSlot, boolean wantHits) {
    check(nResize(ctx, width, height));
    check(nRingCreate(ctx, slots, framesPerSlot, wantHits ? 1 : 0));
  }

  public void destroyFrameRing() {
    check(nRingDestroy(ctx));
  }

  /**
   * Enqueue frames firstFrameNumber .. firstFrameNumber + nframes - 1 of the current camera (what nframes turns of
   * Main.updateEarly with a static camera render: Main.java:275 only increments frameNumber).  Returns the slot the
   * frames went to (pass it to awaitFrames / readFrame), or a negative status.  Does not wait.
   */
  public int submitFrames(int firstFrameNumber, int nframes) {
    check(nSetParams(ctx, firstFrameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    int slot = nRingSubmit(ctx, firstFrameNumber, nframes);
    if (slot < 0)
      printGLErrors();
    else
      frameNumber = firstFrameNumber + nframes - 1;
    return slot;
  }

  /**
   * The same for a camera that moves: frame k carries cams[k] (float[15]: pos, l1, l2, r1, r2 = Camera.getUniform(),
   * Camera.java:142-151) and frameNumbers[k] (Main.updateEarly resets frameNumber on motion: Main.java:161-236, 275).
   * One launch carries all the frames.  Returns the slot, or a negative status.  Does not wait.
   */
  public int submitFrames(float[][] cams, int[] frameNumbers) {
    int n = frameNumbers.length;
    check(nSetParams(ctx, frameNumbers[0], renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    ByteBuffer c = MemoryUtil.memAlloc(60 * n), f = MemoryUtil.memAlloc(4 * n);
    for (int k = 0; k < n; k++) {
      for (int i = 0; i < 15; i++)
        c.putFloat(60 * k + 4 * i, cams[k][i]);
      f.putInt(4 * k, frameNumbers[k]);
    }
    int slot = nRingSubmitCams(ctx, n, MemoryUtil.memAddress(c), MemoryUtil.memAddress(f));
    MemoryUtil.memFree(c);
    MemoryUtil.memFree(f);
    if (slot < 0)
      printGLErrors();
    else
      frameNumber = frameNumbers[n - 1];
    return slot;
  }

  /**
   * BASELINE config 5 in the reference's terms: with setProgressive(true), every dispatch / submitFrames(f, 1) renders
   * nframes consecutive frames of the accumulation of svotrace.comp:712-719 into the one image (svo_set_sequence).
   */
  public void setSequence(int nframes, boolean fresh) {
    check(nSetSequence(ctx, nframes, fresh ? 1 : 0));
  }

  /** Block until the frames of a slot are complete. */
  public void awaitFrames(int slot) {
    check(nRingWait(ctx, slot));
  }

  /** Non-blocking: true once the slot's frames are complete. */
  public boolean framesDone(int slot) {
    return nRingDone(ctx, slot, 0L) == 1;
  }

  /** glGetTexImage of image 0 / image 1 for frame k of a slot (waits for the slot). */
  public void readFrame(int slot, int k, ByteBuffer rgba8, ByteBuffer depth) {
    if (rgba8 != null)
      check(nRingReadColor(ctx, slot, k, MemoryUtil.memAddress(rgba8)));
    if (depth != null)
      check(nRingReadDepth(ctx, slot, k, MemoryUtil.memAddress(depth)));
  }

  /** The crosshair pick of Main.java:132-146 on frame k of a slot. */
  public float readFrameDepthPixel(int slot, int k, int x, int y) {
    ByteBuffer one = MemoryUtil.memAlloc(4);
    check(nRingReadPixel(ctx, slot, k, x, y, 0L, MemoryUtil.memAddress(one), 0L));
    float d = one.getFloat(0);
    MemoryUtil.memFree(one);
    return d;
  }

  /** Launch shape of the persistent pipeline: 0 waves (default) = automatic -- fill the GPU for dispatchCompute, 10 per CU for
   *  the submissions of a ring with more than one slot; a positive count is used as given. */
  public void setTuning(int wavesPerCu, int roundThresholdSixteenths) {
    check(nSetTuning(ctx, wavesPerCu, roundThresholdSixteenths));
  }

  /** 0 one thread per pixel (the shader's decomposition), 1 persistent waves (fastest; what a new context runs), 2 staged wavefront. */
  public void setPipeline(int pipeline) {
    check(nSetPipeline(ctx, pipeline));
  }

  /** How many updateSSBO ranges the library's descriptor table has followed in place (an SDF stroke is two; a stroke
   *  that rewrites the root record makes the next dispatch rebuild the table instead). */
  public long tableRefreshes() {
    return nDerivedRefreshInfo(ctx, 0L, 0L);
  }

  /** Multi-GPU split by interleaved 8-pixel tile rows (one HipRenderer context per GPU). */
  public void setStripes(int firstTileRow, int tileRowStep, int nTileRows, int outRow0) {
    check(nSetStripes(ctx, firstTileRow, tileRowStep, nTileRows, outRow0));
  }

  // ---- the rest of the C ABI, one thin method per native (what a host that drives the library directly needs) ---------
  /** svo_destroy: frees every device resource of the context (the JVM's exit does it otherwise). */
  public void destroy() {
    check(nDestroy(ctx));
    ctx = 0;
  }

  /** svo_dispatch_async / svo_sync: enqueue a frame and return; wait for it. */
  public void dispatchAsync() {
    check(nSetParams(ctx, frameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    check(nDispatchAsync(ctx));
  }

  public void sync() {
    check(nSync(ctx));
  }

  /**
   * svo_set_pick: the pixel readDepthPixel answers without waiting for its frame (default: the image centre, the crosshair of
   * Main.java:139-141).  The frame's pick tile is drawn first and its lane writes the value to pinned host memory; a negative x
   * switches the pick off (every read-back then waits for the frame).
   */
  public void setPick(int x, int y) {
    check(nSetPick(ctx, x, y));
  }

  /** svo_pick_info: how many readDepthPixel calls were answered without waiting for their frame. */
  public long picksAnsweredEarly() {
    return nPickInfo(ctx, 0L, 0L);
  }

  /**
   * svo_set_overlap: dispatchCompute takes turns on several {stream, image} sets so that a frame starts in the tails of the frames
   * before it (default: four sets, up to four frames in flight); read-backs always see the last dispatched frame.  false = one
   * stream, one image set; setOverlap(int sets) names the number (2 .. 8).
   */
  public void setOverlap(boolean enabled) {
    check(nSetOverlap(ctx, enabled ? 1 : 0));
  }

  public void setOverlap(int sets) {
    check(nSetOverlap(ctx, sets));
  }

  /** svo_set_stream: a caller-owned hipStream_t (0 = the library's own). */
  public void setStream(long hipStream) {
    check(nSetStream(ctx, hipStream));
  }

  /** svo_bind_outputs / svo_set_batch: caller-owned device images (device pointers), frames per dispatch. */
  public void bindOutputs(long colorDevicePtr, long depthDevicePtr, long hitsDevicePtr) {
    check(nBindOutputs(ctx, colorDevicePtr, depthDevicePtr, hitsDevicePtr));
  }

  public void setBatch(int nframes, long frameStride) {
    check(nSetBatch(ctx, nframes, frameStride));
  }

  /** svo_ring_bind_slot: a ring slot renders into caller-owned device buffers (a rank's chunk of a gather buffer). */
  public void bindFrameSlot(int slot, long colorDevicePtr, long depthDevicePtr, long hitsDevicePtr, long frameStride) {
    check(nRingBindSlot(ctx, slot, colorDevicePtr, depthDevicePtr, hitsDevicePtr, frameStride));
  }

  /** hit records (16 bytes per pixel: pointer, normal | value | level, iterations, t) of frame k of a slot */
  public void readFrameHits(int slot, int k, ByteBuffer hits) {
    check(nRingReadHits(ctx, slot, k, MemoryUtil.memAddress(hits)));
  }

  /** the beam pre-pass image of the last frame dispatched with useBeamOptimization (ceil(H/4) x ceil(W/4) floats) */
  public void readBeam(ByteBuffer beam) {
    check(nReadBeam(ctx, MemoryUtil.memAddress(beam)));
  }

  public void setRows(int y0, int y1) {
    check(nSetRows(ctx, y0, y1));
  }

  public void setHitRecords(boolean on) {
    check(nSetHitRecords(ctx, on ? 1 : 0));
  }

  /** Persistent waves of the last launch (what the automatic launch shape resolved to); wavesPerCu[0] = waves per CU. */
  public int lastLaunchWaves(int[] wavesPerCu) {
    ByteBuffer w = MemoryUtil.memAlloc(4);
    int n = nLaunchInfo(ctx, MemoryUtil.memAddress(w));
    if (wavesPerCu != null && wavesPerCu.length > 0)
      wavesPerCu[0] = w.getInt(0);
    MemoryUtil.memFree(w);
    return n;
  }

  /** 0 = walk the pool's records as the shader does, 1 (default) = the interior-descriptor table when the pool allows */
  public void setDerived(int mode) {
    check(nSetDerived(ctx, mode));
  }

  /** descriptors of the interior-descriptor table (built if need be), or a negative status; walkable[0] = 1 if it is used */
  public long derivedInfo(int[] walkable) {
    ByteBuffer w = MemoryUtil.memAlloc(4);
    long n = nDerivedInfo(ctx, MemoryUtil.memAddress(w));
    if (walkable != null && walkable.length > 0)
      walkable[0] = w.getInt(0);
    MemoryUtil.memFree(w);
    return n;
  }

  /** svo_count_frame / svo_get_stats into a 56-byte svo_stats (pixels, rays, nan rays, iterations, algorithmic bytes, ...) */
  public void countFrame(ByteBuffer stats56) {
    check(nSetParams(ctx, frameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    check(nCountFrame(ctx, MemoryUtil.memAddress(stats56)));
  }

  public void getStats(ByteBuffer stats56) {
    check(nGetStats(ctx, MemoryUtil.memAddress(stats56)));
  }

  // ---- N GPUs of one node (svo_group_*): the same calls, the work split by screen tile across the devices -------------
  /**
   * createGroup(new int[] {0, 1, ..., 7}): one process, this one render thread, n MI355X.  GPU r renders every n-th 8-pixel
   * tile row of every frame; its stripes travel to GPU 0 over xGMI behind the launch (peer copies on the SDMA engines, or
   * exchange = 1: RCCL send / receive); readFrame hands out whole frames.  The pool is uploaded once and replicated device
   * to device.  Same results, byte for byte, as the single-GPU renderer.
   */
  public static Group createGroup(int[] devices) {
    ByteBuffer d = MemoryUtil.memAlloc(4 * devices.length);
    for (int i = 0; i < devices.length; i++)
      d.putInt(4 * i, devices[i]);
    long h = nGroupCreate(MemoryUtil.memAddress(d), devices.length);
    MemoryUtil.memFree(d);
    if (h <= 0) {
      System.out.println("HIP ERR: svo_group_create status " + h);
      return null;
    }
    return new Group(h, devices.length);
  }

  public static class Group {
    private long g;
    private final int size;
    private int renderMode = 2, bufferEnd = 0, useBeam = 0, bounces = 2, mirrorMask = 0, spp = 1;

    Group(long handle, int n) {
      g = handle;
      size = n;
    }

    private void check(int rc) {
      if (rc != 0)
        System.out.println("HIP ERR: " + MemoryUtil.memUTF8(nGroupLastError(g)));   // print and go on, as Renderer does
    }

    public int size() {
      return size;
    }

    /** Renderer.addSSBO: the whole pool, once from the host, then GPU to GPU. */
    public void addSSBO(int bindIndex, ByteBuffer data) {
      check(nGroupPoolUpload(g, MemoryUtil.memAddress(data), data.remaining()));
    }

    /** Renderer.updateSSBO(bindIndex, data, start, end): a brush stroke's byte range, to every GPU. */
    public void updateSSBO(int bindIndex, ByteBuffer data, int start, int end) {
      if (start >= end) {
        System.out.println("Update SSBO error: Invalid parameters.");   // Renderer.java:137-140
        return;
      }
      check(nGroupPoolUpdate(g, MemoryUtil.memAddress0(data), start, end));
    }

    public void setImageSize(int width, int height) {
      check(nGroupResize(g, width, height));
    }

    /** Camera.getUniform(): pos, l1, l2, r1, r2 (Main.java:269-273). */
    public void setCamera(float[][] u) {
      ByteBuffer c = MemoryUtil.memAlloc(60);
      for (int v = 0; v < 5; v++)
        for (int i = 0; i < 3; i++)
          c.putFloat(4 * (3 * v + i), u[v][i]);
      check(nGroupSetCamera(g, MemoryUtil.memAddress(c)));
      MemoryUtil.memFree(c);
    }

    public void setFrameParams(int renderMode, int bufferEnd, boolean useBeam) {
      this.renderMode = renderMode;
      this.bufferEnd = bufferEnd;
      this.useBeam = useBeam ? 1 : 0;
    }

    public void setPathOptions(int bounces, int mirrorMask, int spp) {
      this.bounces = bounces;
      this.mirrorMask = mirrorMask;
      this.spp = spp;
    }

    public void setTuning(int wavesPerCu, int roundThresholdSixteenths) {
      check(nGroupSetTuning(g, wavesPerCu, roundThresholdSixteenths));
    }

    public void setProgressive(boolean on, int framesPerDispatch, boolean fresh) {
      check(nGroupSetProgressive(g, on ? 1 : 0));
      check(nGroupSetSequence(g, framesPerDispatch, fresh ? 1 : 0));
    }

    public void createFrameRing(int slots, int framesPerSlot, boolean wantHits, int exchange) {
      check(nGroupRingCreate(g, slots, framesPerSlot, wantHits ? 1 : 0, exchange));
    }

    public void destroyFrameRing() {
      check(nGroupRingDestroy(g));
    }

    /** frames firstFrameNumber .. + nframes - 1 of the current camera on every GPU; returns the slot or a negative status */
    public int submitFrames(int firstFrameNumber, int nframes) {
      check(nGroupSetParams(g, firstFrameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
      return nGroupRingSubmit(g, firstFrameNumber, nframes);
    }

    /** frames with their own cameras (float[15] each) and frame numbers */
    public int submitFrames(float[][] cams, int[] frameNumbers) {
      int n = frameNumbers.length;
      check(nGroupSetParams(g, frameNumbers[0], renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
      ByteBuffer c = MemoryUtil.memAlloc(60 * n), f = MemoryUtil.memAlloc(4 * n);
      for (int k = 0; k < n; k++) {
        for (int i = 0; i < 15; i++)
          c.putFloat(60 * k + 4 * i, cams[k][i]);
        f.putInt(4 * k, frameNumbers[k]);
      }
      int slot = nGroupRingSubmitCams(g, n, MemoryUtil.memAddress(c), MemoryUtil.memAddress(f));
      MemoryUtil.memFree(c);
      MemoryUtil.memFree(f);
      return slot;
    }

    public void awaitFrames(int slot) {
      check(nGroupRingWait(g, slot));
    }

    public boolean framesDone(int slot) {
      return nGroupRingDone(g, slot, 0L) == 1;
    }

    /** glGetTexImage of image 0 / image 1 for frame k of a slot: whole frames, rows in frame order. */
    public void readFrame(int slot, int k, ByteBuffer rgba8, ByteBuffer depth) {
      if (rgba8 != null)
        check(nGroupRingReadColor(g, slot, k, MemoryUtil.memAddress(rgba8)));
      if (depth != null)
        check(nGroupRingReadDepth(g, slot, k, MemoryUtil.memAddress(depth)));
    }

    /** The crosshair pick of Main.java:132-146. */
    public float readFrameDepthPixel(int slot, int k, int x, int y) {
      ByteBuffer one = MemoryUtil.memAlloc(4);
      check(nGroupRingReadPixel(g, slot, k, x, y, 0L, MemoryUtil.memAddress(one), 0L));
      float d = one.getFloat(0);
      MemoryUtil.memFree(one);
      return d;
    }

    /** member i's context handle (svo_group_member), for the per-GPU natives: statistics, descriptor-table information */
    public long memberContext(int i) {
      return nGroupMember(g, i);
    }

    public void destroy() {
      check(nGroupDestroy(g));
      g = 0;
    }
  }

  private void check(int rc) {
    if (rc != 0)
      printGLErrors();
  }

  private static native long nCreate(int device);
  private static native int nDestroy(long ctx);
  private static native long nLastError(long ctx);
  private static native int nPoolUpload(long ctx, long addr, long nbytes);
  private static native int nPoolUpdate(long ctx, long baseAddr, long start, long end);
  private static native int nPoolDownload(long ctx, long addr, long nbytes);
  private static native int nSetCamera(long ctx, float px, float py, float pz, float l1x, float l1y, float l1z,
      float l2x, float l2y, float l2z, float r1x, float r1y, float r1z, float r2x, float r2y, float r2z);
  private static native int nSetParams(long ctx, int frameNumber, int renderMode, int bufferEnd, int useBeam,
      int bounces, int mirrorMask, int spp);
  private static native int nResize(long ctx, int width, int height);
  private static native int nDispatch(long ctx);
  private static native int nReadColor(long ctx, long addr);
  private static native int nReadDepth(long ctx, long addr);
  private static native int nReadHits(long ctx, long addr);
  private static native long nBuildFromVoxels(long ctx, long voxelsAddr, int n);
  private static native int nSetProgressive(long ctx, int enabled);
  private static native int nSetBatch(long ctx, int nframes, long frameStride);
  private static native int nBindOutputs(long ctx, long colorDevicePtr, long depthDevicePtr, long hitsDevicePtr);
  private static native int nReadBeam(long ctx, long addr);
  private static native long nBuildFromHeightmap(long ctx, long heightAddr, long materialAddr, int n);
  private static native int nReadPixel(long ctx, int x, int y, long rgbaAddr, long depthAddr, long hitAddr);
  private static native int nDispatchAsync(long ctx);
  private static native int nSync(long ctx);
  private static native int nSetPick(long ctx, int x, int y);
  private static native int nSetOverlap(long ctx, int sets);
  private static native long nPickInfo(long ctx, long xyAddr, long waitedAddr);
  private static native int nSetStream(long ctx, long hipStream);
  private static native int nSetPipeline(long ctx, int pipeline);
  private static native int nSetTuning(long ctx, int wavesPerCu, int roundThresholdSixteenths);
  private static native int nLaunchInfo(long ctx, long wavesPerCuAddr);
  private static native int nSetDerived(long ctx, int mode);
  private static native int nSetHitRecords(long ctx, int enabled);
  private static native int nSetRows(long ctx, int y0, int y1);
  private static native int nSetStripes(long ctx, int firstTileRow, int tileRowStep, int nTileRows, int outRow0);
  private static native int nCountFrame(long ctx, long statsAddr);
  private static native int nGetStats(long ctx, long statsAddr);
  private static native long nDerivedInfo(long ctx, long walkableAddr);
  private static native long nDerivedRefreshInfo(long ctx, long statesAddr, long addedAddr);
  private static native int nRingCreate(long ctx, int slots, int framesPerSlot, int wantHits);
  private static native int nRingDestroy(long ctx);
  private static native int nRingSubmit(long ctx, int frameNumber, int nframes);
  private static native int nRingSubmitCams(long ctx, int nframes, long camsAddr, long frameNumbersAddr);
  private static native int nSetSequence(long ctx, int nframes, int fresh);
  private static native int nRingWait(long ctx, int slot);
  private static native int nRingDone(long ctx, int slot, long msAddr);
  private static native int nRingReadColor(long ctx, int slot, int k, long addr);
  private static native int nRingReadDepth(long ctx, int slot, int k, long addr);
  private static native int nRingReadHits(long ctx, int slot, int k, long addr);
  private static native int nRingReadPixel(long ctx, int slot, int k, int x, int y, long rgbaAddr, long depthAddr, long hitAddr);
  private static native int nRingBindSlot(long ctx, int slot, long colorDevicePtr, long depthDevicePtr, long hitsDevicePtr,
      long frameStride);
  private static native long nGroupCreate(long devicesAddr, int n);
  private static native int nGroupDestroy(long g);
  private static native long nGroupLastError(long g);
  private static native long nGroupMember(long g, int i);
  private static native int nGroupPoolUpload(long g, long addr, long nbytes);
  private static native int nGroupPoolUpdate(long g, long baseAddr, long start, long end);
  private static native int nGroupSetCamera(long g, long cam15Addr);
  private static native int nGroupSetParams(long g, int frameNumber, int renderMode, int bufferEnd, int useBeam, int bounces,
      int mirrorMask, int spp);
  private static native int nGroupSetTuning(long g, int wavesPerCu, int roundThresholdSixteenths);
  private static native int nGroupSetProgressive(long g, int enabled);
  private static native int nGroupSetSequence(long g, int nframes, int fresh);
  private static native int nGroupResize(long g, int width, int height);
  private static native int nGroupRingCreate(long g, int slots, int framesPerSlot, int wantHits, int exchange);
  private static native int nGroupRingDestroy(long g);
  private static native int nGroupRingSubmit(long g, int frameNumber, int nframes);
  private static native int nGroupRingSubmitCams(long g, int nframes, long camsAddr, long frameNumbersAddr);
  private static native int nGroupRingWait(long g, int slot);
  private static native int nGroupRingDone(long g, int slot, long msAddr);
  private static native int nGroupRingReadColor(long g, int slot, int k, long addr);
  private static native int nGroupRingReadDepth(long g, int slot, int k, long addr);
  private static native int nGroupRingReadPixel(long g, int slot, int k, int x, int y, long rgbaAddr, long depthAddr, long hitAddr);
}
