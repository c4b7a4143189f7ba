package src.engine;

import java.io.FileOutputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.channels.FileChannel;

import src.engine.sdf.Box;
import src.engine.sdf.SignedDistanceField;
import src.engine.sdf.Sphere;

/**
 * One-command pins for everything on the Java side of the boundary: runs the REFERENCE's own
 * Octree.constructInnerOctree / genSurfaceNormal / checkBigNodeExposed (Octree.java:511-670),
 * Octree.useSDFBrush / subdivideNode (Octree.java:672-885, sdf/Sphere, sdf/Box) and Camera.rotate
 * (Camera.java:76-140, JOML) on seeded inputs and writes inputs + outputs to tests/golden/java_*.bin.
 * tests/test_java_fixtures.py picks the files up when present and compares the GPU builder
 * (svo_build_from_voxels), the restatement (oracle/octree_restatement.cpp), the numpy builder
 * (tests/poolbuilder.py) and the C++ camera mirror (host/svo_host.hpp) with them, byte for byte.
 *
 * NOT RUNNABLE in the build image (no JDK); never compiled there.  With a JDK, from the reference checkout:
 *
 *   javac -cp "lib/*" -d out src/engine/*.java src/engine/sdf/*.java \
 *         /path/to/repo/integration/java/tools/src/engine/DumpFixtures.java
 *   java -Xmx6g -XX:MaxDirectMemorySize=4g -cp "out:lib/*" src.engine.DumpFixtures /path/to/repo/tests/golden
 *
 * (the class sits in package src.engine because Octree.buffer / memOffset and Camera.dir / rot are
 * package-private; no window, no GL context: only Octree, Camera, Util, Constants, sdf/* and
 * org.lwjgl.BufferUtils / JOML are touched.)
 *
 * Voxel layout: Octree.getVoxel reads voxelData.get(x | y << 10 | z << 20) whatever the cube's size, so
 * every chunk here lives in the corner of a zero-filled 1024^3 buffer.  Bounds tests in genSurfaceNormal /
 * checkBigNodeExposed are against CHUNK_SIZE = 1024 -- for a cube of n < 1024 the voxels just outside it
 * are in bounds and EMPTY, where a builder that takes the n^3 cube as the whole chunk ignores them.  The
 * generators below therefore keep the three high faces (x, y or z = n - 1) empty: no solid voxel has a
 * neighbour outside the cube, and the two readings agree.
 *
 * All files little-endian.
 *   java_build.bin : "SVOJBLD1", u32 count, then per case  u32 n, u32 kind, u32 seed, u32 poolLen,
 *                    n^3 voxel bytes (index x + n * (y + n * z)), poolLen pool bytes
 *                    (pool = createDummyHead() + constructInnerOctree(n, 0, log2 n, {0,0,0}, 0, voxels))
 *   java_brush.bin : "SVOJBRS1", u32 count, then per case  u32 n, u32 kind, u32 seed, u32 nStrokes, u32 baseLen,
 *                    base pool bytes, then per stroke  i32 type (0 sphere, 1 box), i32 ox, oy, oz, i32 a, b, c
 *                    (radius, 0, 0 | width, height, depth), i32 value, i32 start0, end0, start1, end1 (ChangeBounds),
 *                    u32 poolLen, poolLen pool bytes (the whole pool after the stroke)
 *   java_camera.bin: "SVOJCAM1", u32 nSequences, then per sequence  u32 nSteps, f32 pos[3] (setPos), then per step
 *                    f32 x, y, z (the rotate call), f32 uniform[15] (pos, l1, l2, r1, r2), f32 dir[3], f32 rot[3]
 */
public class DumpFixtures {

  // ---- deterministic integer noise (no java.util.Random: the streams are part of no contract) ----------------
  static int hash(int x, int y, int z, int seed) {
    int h = seed * 0x9E3779B1 + x * 0x85EBCA6B + y * 0xC2B2AE35 + z * 0x27D4EB2F;
    h ^= h >>> 15;
    h *= 0x2C1B3C6D;
    h ^= h >>> 12;
    h *= 0x297A2D39;
    h ^= h >>> 15;
    return h;
  }

  /** value noise in [0, 256) on a lattice of `cell` voxels, trilinear in integers */
  static int noise3(int x, int y, int z, int cell, int seed) {
    int x0 = x / cell, y0 = y / cell, z0 = z / cell;
    int fx = (x % cell) * 256 / cell, fy = (y % cell) * 256 / cell, fz = (z % cell) * 256 / cell;
    int acc = 0;
    for (int c = 0; c < 8; c++) {
      int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
      int v = hash(x0 + dx, y0 + dy, z0 + dz, seed) & 255;
      int wx = dx == 1 ? fx : 256 - fx, wy = dy == 1 ? fy : 256 - fy, wz = dz == 1 ? fz : 256 - fz;
      acc += (int) (((long) v * wx * wy * wz) >> 24);
    }
    return acc;
  }

  /** voxel (0 = empty) of the seeded chunk of the given kind; the high faces of the cube stay empty */
  static byte voxel(int kind, int n, int seed, int x, int y, int z) {
    if (x >= n - 1 || y >= n - 1 || z >= n - 1)
      return 0;
    int cell = Math.max(2, n / 4);
    switch (kind) {
      case 0: { // terrain: solid up to a height, the top layers take a hashed material (chunkgen-heightmap.comp:16-28)
        int h = n / 4 + noise3(x, 0, z, cell, seed) * (n / 2) / 256;
        if (y > h)
          return 0;
        if (h - y <= 4)
          return (byte) (2 + (hash(x, 1, z, seed) & 1));
        return 1;
      }
      case 1: { // caves and overhangs: thresholded 3-D noise, two materials
        int v = noise3(x, y, z, cell, seed);
        if (v < 120)
          return 0;
        return (byte) (v > 180 ? 3 : 1);
      }
      case 2: // dust: isolated voxels (packed-555 normals, fully exposed leaves)
        return (hash(x, y, z, seed) & 15) == 0 ? (byte) (1 + ((hash(x, y, z, seed + 1) >>> 4) % 3)) : 0;
      case 3: { // a solid block with drilled holes: homogeneous big nodes, exposed or not (checkBigNodeExposed)
        boolean hole = ((x / 3) % 4 == 1 && (z / 3) % 4 == 1) || (hash(x / 4, y / 4, z / 4, seed) & 31) == 0;
        return hole ? 0 : (byte) 1;
      }
      default: // islands: blobs of one material floating in air
        return noise3(x, y, z, Math.max(2, n / 8), seed) > 170 ? (byte) (1 + (seed % 3)) : 0;
    }
  }

  static ByteBuffer chunk; // 1 GiB, zero-filled, indexed x | y << 10 | z << 20 (Octree.getVoxel)

  static byte[] fillChunk(int kind, int n, int seed) {
    byte[] vox = new byte[n * n * n];
    for (int z = 0; z < n; z++)
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
          byte v = voxel(kind, n, seed, x, y, z);
          vox[x + n * (y + n * z)] = v;
          chunk.put(x | (y << 10) | (z << 20), v);
        }
    return vox;
  }

  static void clearChunk(int n) {
    for (int z = 0; z < n; z++)
      for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
          chunk.put(x | (y << 10) | (z << 20), (byte) 0);
  }

  static int log2(int n) {
    int l = 0;
    while ((1 << l) < n)
      l++;
    return l;
  }

  /** OctreeThread.run on a cube of n voxels at the chunk's origin (OctreeThread.java:20-23 with 512 -> n) */
  static Octree build(int n) {
    Octree o = new Octree(Math.max(8192, n * n * n / 16));   // KB; brush strokes append to it
    o.createDummyHead();
    o.constructInnerOctree(n, 0, log2(n), new int[] { 0, 0, 0 }, 0, chunk);
    return o;
  }

  static byte[] poolBytes(Octree o) {
    byte[] b = new byte[o.memOffset];
    for (int i = 0; i < o.memOffset; i++)
      b[i] = o.buffer.get(i);
    return b;
  }

  // ---- little-endian output ------------------------------------------------------------------------------------
  static FileChannel out;
  static ByteBuffer le = ByteBuffer.allocate(1 << 16).order(ByteOrder.LITTLE_ENDIAN);

  static void flush() throws IOException {
    le.flip();
    while (le.hasRemaining())
      out.write(le);
    le.clear();
  }

  static void i32(int v) throws IOException {
    if (le.remaining() < 4)
      flush();
    le.putInt(v);
  }

  static void f32(float v) throws IOException {
    if (le.remaining() < 4)
      flush();
    le.putFloat(v);
  }

  static void bytes(byte[] b) throws IOException {
    flush();
    ByteBuffer w = ByteBuffer.wrap(b);
    while (w.hasRemaining())
      out.write(w);
  }

  static void open(String path, String magic) throws IOException {
    out = new FileOutputStream(path).getChannel();
    bytes(magic.getBytes("US-ASCII"));
  }

  static void close() throws IOException {
    flush();
    out.close();
  }

  // ---- the three fixture files ----------------------------------------------------------------------------------
  static final int[][] BUILD_CASES = { // n, kind, seed
      { 2, 2, 1 }, { 2, 3, 2 }, { 4, 0, 3 }, { 4, 1, 4 }, { 4, 3, 5 }, { 8, 0, 6 }, { 8, 1, 7 }, { 8, 2, 8 }, { 8, 3, 9 },
      { 16, 0, 10 }, { 16, 1, 11 }, { 16, 2, 12 }, { 16, 3, 13 }, { 16, 4, 14 }, { 32, 0, 15 }, { 32, 1, 16 }, { 32, 3, 17 },
      { 32, 4, 18 }, { 64, 0, 19 }, { 64, 1, 20 }, { 64, 2, 21 }, { 64, 3, 22 }, { 128, 0, 23 }, { 128, 1, 24 }, { 128, 4, 25 } };

  static void dumpBuilds(String dir) throws IOException {
    open(dir + "/java_build.bin", "SVOJBLD1");
    i32(BUILD_CASES.length);
    for (int[] c : BUILD_CASES) {
      int n = c[0], kind = c[1], seed = c[2];
      byte[] vox = fillChunk(kind, n, seed);
      Octree o = build(n);
      i32(n); i32(kind); i32(seed); i32(o.memOffset);
      bytes(vox);
      bytes(poolBytes(o));
      clearChunk(n);
      System.out.println("build n=" + n + " kind=" + kind + " seed=" + seed + " -> " + o.memOffset + " bytes");
    }
    close();
  }

  /**
   * Brush strokes on pools the reference's own builder made.  useSDFBrush walks the octree as the cube of
   * Constants.WORLD_SIZE (8196 [sic]) voxels whatever built it, down to LOD 13, so a pool of an n^3 chunk stands for a
   * world whose leaves are (8196 / n)-voxel blocks; the strokes below are placed on its surface and in its air, fill
   * (value 1..3) and delete (Constants.DELETE_VALUE).
   */
  static final int[][] BRUSH_BASES = { { 16, 0, 31 }, { 32, 1, 32 }, { 64, 0, 33 }, { 32, 3, 34 } }; // n, kind, seed

  static void dumpBrushes(String dir) throws IOException {
    open(dir + "/java_brush.bin", "SVOJBRS1");
    i32(BRUSH_BASES.length);
    for (int[] c : BRUSH_BASES) {
      int n = c[0], kind = c[1], seed = c[2];
      fillChunk(kind, n, seed);
      Octree o = build(n);
      clearChunk(n);
      int[][] strokes = new int[6][];
      int w = Constants.WORLD_SIZE;
      for (int s = 0; s < strokes.length; s++) {
        int hx = hash(s, 0, 0, seed), hy = hash(s, 1, 0, seed), hz = hash(s, 2, 0, seed);
        int ox = w / 8 + (hx >>> 8) % (w / 2), oy = w / 8 + (hy >>> 8) % (w / 2), oz = w / 8 + (hz >>> 8) % (w / 2);
        byte value = (s % 3 == 2) ? Constants.DELETE_VALUE : (byte) (1 + s % 3);
        if (s % 2 == 0)
          strokes[s] = new int[] { 0, ox, oy, oz, 20 + (hx & 63), 0, 0, value };
        else
          strokes[s] = new int[] { 1, ox, oy, oz, 16 + (hx & 31), 16 + (hy & 31), 16 + (hz & 31), value };
      }
      i32(n); i32(kind); i32(seed); i32(strokes.length); i32(o.memOffset);
      bytes(poolBytes(o));
      for (int[] s : strokes) {
        SignedDistanceField sdf = s[0] == 0 ? new Sphere(new int[] { s[1], s[2], s[3] }, s[4])
            : new Box(new int[] { s[1], s[2], s[3] }, s[4], s[5], s[6]);
        Octree.ChangeBounds cb = o.useSDFBrush(sdf, (byte) s[7]);
        for (int v : s)
          i32(v);
        i32(cb.start0); i32(cb.end0); i32(cb.start1); i32(cb.end1);
        i32(o.memOffset);
        bytes(poolBytes(o));
        System.out.println("brush base n=" + n + " stroke type " + s[0] + " -> " + o.memOffset + " bytes, bounds " + cb.start0
            + ".." + cb.end0 + ", " + cb.start1 + ".." + cb.end1);
      }
    }
    close();
  }

  static final float[][][] CAMERA_SEQS = { // sequences of rotate(x, y, z) calls, as Main feeds them (mouse deltas * sensitivity)
      { { 0.0f, 0.3f, 0.0f }, { -0.2f, 0.0f, 0.0f }, { 0.05f, -0.7f, 0.0f } },
      { { -0.5f, 0.0f, 0.0f }, { 0.0f, 0.7f, 0.0f }, { 0.0f, 0.7f, 0.0f }, { 0.0f, 5.0f, 0.0f }, { 0.1f, -9.0f, 0.0f } },
      { { 2.0f, 0.0f, 0.0f }, { -4.0f, 0.0f, 0.0f }, { 1.0f, 1.0f, 0.0f } }, // the pitch clamp, both ways
      { { 0.002f, 0.004f, 0.0f }, { 0.002f, 0.004f, 0.0f }, { -0.006f, 0.002f, 0.0f }, { 0.0f, -0.01f, 0.0f },
        { 0.002f, 0.004f, 0.0f }, { 0.002f, 0.004f, 0.0f }, { 0.002f, 0.004f, 0.0f }, { 0.002f, 0.004f, 0.0f } } };

  static void dumpCameras(String dir) throws IOException {
    open(dir + "/java_camera.bin", "SVOJCAM1");
    i32(CAMERA_SEQS.length);
    for (float[][] seq : CAMERA_SEQS) {
      Camera cam = new Camera();
      cam.setPos(1.5f, 1.5f, 2.0f);   // Main.java:120
      i32(seq.length);
      f32(1.5f); f32(1.5f); f32(2.0f);
      for (float[] r : seq) {
        cam.rotate(r[0], r[1], r[2]);
        f32(r[0]); f32(r[1]); f32(r[2]);
        float[][] u = cam.getUniform();
        for (int i = 0; i < 5; i++)
          for (int k = 0; k < 3; k++)
            f32(u[i][k]);
        for (int k = 0; k < 3; k++)
          f32(cam.dir[k]);
        for (int k = 0; k < 3; k++)
          f32(cam.rot[k]);
      }
    }
    close();
  }

  public static void main(String[] args) throws IOException {
    String dir = args.length > 0 ? args[0] : ".";
    chunk = ByteBuffer.allocateDirect(1 << 30);
    dumpBuilds(dir);
    dumpBrushes(dir);
    dumpCameras(dir);
    System.out.println("wrote java_build.bin, java_brush.bin, java_camera.bin to " + dir);
  }
}
