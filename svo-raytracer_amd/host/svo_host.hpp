// svo_host.hpp -- C++ host-side mirror of the reference's Java interface for the hot path.
//
// The reference host is Java (no JDK in the build image), so the host side above the C ABI
// is written in C++ with the reference's own names, argument meaning and error behaviour:
//   svo::host::Renderer  <->  src/engine/Renderer.java   (singleton, shaders, SSBO, dispatch)
//   svo::host::Camera    <->  src/engine/Camera.java     (pos + 4 corner rays, rotate/strafe)
//   svo::host::Octree    <->  src/engine/Octree.java     (byte pool, node encoders, .svo IO)
// A user of the reference swaps `Renderer` for this class and keeps the frame loop of
// Main.updateEarly (Main.java:257-289) -- see INTEGRATION.md for the Java twin.
// Like the reference, nothing here throws: failures are printed and the call returns.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/svo_hip.h"

namespace svo {
namespace host {

// ---------------------------------------------------------------------------------------------
// Constants.java
struct Constants {
  static constexpr int WINDOW_WIDTH = 1920;            // Constants.java:4
  static constexpr int WINDOW_HEIGHT = 1080;           // :5
  static constexpr float CAMERA_LOWER_LIMIT = -1.570f; // :9
  static constexpr float CAMERA_UPPER_LIMIT = 1.570f;  // :10
  static constexpr int OCTREE_MEMORY_SIZE_KB = 2000000; // :11
  static constexpr int COMPUTE_GROUP_SIZE = 8;         // :13
  static constexpr int WORLD_SIZE = 8196;              // :30 (sic)
  static constexpr float PI = 3.14159265359f;          // :35
};


// ---------------------------------------------------------------------------------------------
// Octree.java -- only what the hot path consumes: the byte pool and its node encoders.
class Octree {
 public:
  static constexpr int NODE_SIZE = 7, LEAF_SIZE = 3, NON_SURFACE_LEAF_SIZE = 1;  // Octree.java:36-38
  std::vector<uint8_t> buffer;  // Octree.java:27 (direct ByteBuffer, BIG_ENDIAN for multi-byte puts)
  int memOffset = 0;            // :28
  int bufferSize = 0;           // :29
  long surfaceLeafNodes = 0, nonSurfaceLeafNodes = 0, interiorNodes = 0, subdividableLeafNodes = 0;  // :31-34

  explicit Octree(int memSizeKB) : buffer((size_t)memSizeKB * 1024, 0), bufferSize(memSizeKB * 1024) {}  // :63-67

  void createDummyHead() { createInteriorNode(1); }                    // :97-100
  uint8_t getValue(int node) const { return buffer[(size_t)node]; }     // :102-104
  void setValue(int node, uint8_t v) { buffer[(size_t)node] = v; }      // :106-108
  void setNormal(int surfaceLeafNode, uint16_t normal) {                // :114-117 (little-endian)
    buffer[(size_t)surfaceLeafNode + 1] = (uint8_t)normal;
    buffer[(size_t)surfaceLeafNode + 2] = (uint8_t)(normal >> 8);
  }
  // node encoders (private in the reference, Octree.java:119-176; public here for tools)
  int createInteriorNode(uint8_t val) { interiorNodes++; return put7(val); }
  int createSubdividableLeafNode(uint8_t val) { subdividableLeafNodes++; return put7(val); }
  int createSurfaceLeafNode(uint8_t val, uint16_t normal) {
    surfaceLeafNodes++;
    int p = memOffset;
    buffer[(size_t)memOffset++] = val;
    buffer[(size_t)memOffset++] = (uint8_t)normal;
    buffer[(size_t)memOffset++] = (uint8_t)(normal >> 8);
    return p;
  }
  int createNonSurfaceLeafNode(uint8_t val) {
    nonSurfaceLeafNodes++;
    int p = memOffset;
    buffer[(size_t)memOffset++] = val;
    return p;
  }
  void setChildPointer(int parent, int child) { putIntBE(parent + 1, child - parent); }   // :162-164
  int getChildPointer(int parent) const { return getIntBE(parent + 1) + parent; }          // :166-168
  void setLeafMask(int parent, uint16_t m) {                                                // :170-172
    buffer[(size_t)parent + 5] = (uint8_t)(m >> 8);
    buffer[(size_t)parent + 6] = (uint8_t)m;
  }
  uint16_t getLeafMask(int parent) const {                                                  // :174-176
    return (uint16_t)((buffer[(size_t)parent + 5] << 8) | buffer[(size_t)parent + 6]);
  }
  uint8_t *getByteBuffer() { return buffer.data(); }                                        // :958-960
  void editLeafNodeValue(int pointer, uint8_t val) { buffer[(size_t)pointer] = val; }       // :1014-1016

  // .svo = 4-byte big-endian memOffset + the pool bytes (Octree.java:974-1012).  The reference
  // prefixes Constants.MAP_DIR; here the caller passes the full path.
  void writeBufferToFile(const std::string &path) const {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { printf("Error while writing buffer to %s: \n", path.c_str()); return; }
    const uint8_t hdr[4] = {(uint8_t)(memOffset >> 24), (uint8_t)(memOffset >> 16), (uint8_t)(memOffset >> 8),
                            (uint8_t)memOffset};
    fwrite(hdr, 1, 4, f);
    fwrite(buffer.data(), 1, (size_t)memOffset, f);
    fclose(f);
  }
  void readBufferFromFile(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { printf("Error while reading file %s: \n", path.c_str()); return; }
    uint8_t hdr[4] = {0, 0, 0, 0};
    if (fread(hdr, 1, 4, f) != 4) { printf("Error while reading file %s: \n", path.c_str()); fclose(f); return; }
    size_t n = fread(buffer.data(), 1, buffer.size(), f);  // the reference reads as much as the buffer holds
    (void)n;
    fclose(f);
    memOffset = (int)(((uint32_t)hdr[0] << 24) | ((uint32_t)hdr[1] << 16) | ((uint32_t)hdr[2] << 8) | hdr[3]);
  }
  void printNodeCounts() const {  // :1018-1026
    printf("Surface Leaves: %ld\nNon-Surface Leaves: %ld\nSubdividable Leaves: %ld\nInterior Nodes: %ld\nTotal: %ld\n",
           surfaceLeafNodes, nonSurfaceLeafNodes, subdividableLeafNodes, interiorNodes,
           surfaceLeafNodes + nonSurfaceLeafNodes + interiorNodes + subdividableLeafNodes);
  }


  // adopt a pool produced elsewhere (procedural scene generator)
  void adopt(const uint8_t *pool, size_t n) {
    if (n > buffer.size()) buffer.resize(n);
    memcpy(buffer.data(), pool, n);
    memOffset = (int)n;
  }

 private:
  int put7(uint8_t val) {
    int p = memOffset;
    buffer[(size_t)memOffset++] = val;
    for (int i = 0; i < 6; i++) buffer[(size_t)memOffset++] = 0;
    return p;
  }
  void putIntBE(int at, int v) {
    uint32_t u = (uint32_t)v;
    buffer[(size_t)at] = (uint8_t)(u >> 24); buffer[(size_t)at + 1] = (uint8_t)(u >> 16);
    buffer[(size_t)at + 2] = (uint8_t)(u >> 8); buffer[(size_t)at + 3] = (uint8_t)u;
  }
  int getIntBE(int at) const {
    return (int)(((uint32_t)buffer[(size_t)at] << 24) | ((uint32_t)buffer[(size_t)at + 1] << 16) |
                 ((uint32_t)buffer[(size_t)at + 2] << 8) | (uint32_t)buffer[(size_t)at + 3]);
  }
};

// ---------------------------------------------------------------------------------------------
// Camera.java.  rotate() multiplies the four corner rays by Ry(y) * R_right(x) every call, with
// JOML 1.10.5 float semantics: sin = (float)sin((double)a), cos = cosFromSin (sqrt(1 - sin^2), sign
// from the angle), plain (unfused) multiply-adds.
class Camera {
 public:
  float pos[3] = {0, 0, 0};
  float speed = 0.005f;
  float scaleX = 0.9f, scaleY = 1.6f;                      // Camera.java:13-14
  float l1[3] = {-1.6f, -0.9f, -1}, l2[3] = {-1.6f, 0.9f, -1};  // :15-16
  float r1[3] = {1.6f, -0.9f, -1}, r2[3] = {1.6f, 0.9f, -1};    // :17-18
  float rot[3] = {0, 0, 0}, dir[3] = {0, 0, 1}, right[3] = {1, 0, 0};  // :19-21

  void setSpeed(float s) { speed = s; }
  void setPos(float x, float y, float z) { pos[0] = x; pos[1] = y; pos[2] = z; }   // :40-44
  const float *getPos() const { return pos; }
  void strafe(float forwardSpeed, float sideSpeed) {                                // :46-50
    for (int i = 0; i < 3; i++) pos[i] += -dir[i] * speed * forwardSpeed + right[i] * speed * sideSpeed;
  }
  void rotate(float x, float y, float z) {                                          // :76-140
    if (rot[0] + x < Constants::CAMERA_LOWER_LIMIT || rot[0] + x > Constants::CAMERA_UPPER_LIMIT) {
      rot[0] = x > 0 ? Constants::CAMERA_UPPER_LIMIT : Constants::CAMERA_LOWER_LIMIT;
      x = 0;
    } else {
      rot[0] += x;
    }
    rot[1] += y;
    rot[1] = std::fmod(rot[1], Constants::PI * 2);  // Java float % float == fmodf
    rot[2] += z;
    rot[2] = std::fmod(rot[2], Constants::PI * 2);
    // convertToCameraXAxis(1, 0, 0): (float)(x*cos(alpha) + z*sin(alpha)) in double  (:58-67)
    const double alpha = rot[1];
    right[0] = (float)(1.0 * std::cos(alpha) + 0.0 * std::sin(alpha));
    right[1] = 0.0f;
    right[2] = (float)(0.0 * std::cos(alpha) - 1.0 * std::sin(alpha));
    float m[9];
    rotationY(y, m);                                 // matrix.rotate(y, 0, 1, 0) on the identity
    float ra[9], mm[9];
    rotationAxis(x, right[0], right[1], right[2], ra);
    mul3(m, ra, mm);                                 // matrix.rotate(x, right) post-multiplies
    // updateDirection: (0,0,1).rotateX(rot0).rotateY(rot1).rotateZ(rot2)  (:69-74)
    float dx = 0, dy = 0, dz = 1, s, c, t0, t1;
    s = jsin(rot[0]); c = jcos(s, rot[0]); t0 = dy * c - dz * s; t1 = dy * s + dz * c; dy = t0; dz = t1;
    s = jsin(rot[1]); c = jcos(s, rot[1]); t0 = dx * c + dz * s; t1 = -dx * s + dz * c; dx = t0; dz = t1;
    s = jsin(rot[2]); c = jcos(s, rot[2]); t0 = dx * c - dy * s; t1 = dx * s + dy * c; dx = t0; dy = t1;
    dir[0] = dx; dir[1] = dy; dir[2] = dz;
    apply(mm, l1); apply(mm, l2); apply(mm, r1); apply(mm, r2);
  }
  // {pos, l1, l2, r1, r2}  (:142-151)
  void getUniform(float out[15]) const {
    memcpy(out, pos, 12); memcpy(out + 3, l1, 12); memcpy(out + 6, l2, 12); memcpy(out + 9, r1, 12);
    memcpy(out + 12, r2, 12);
  }
  // Util.toVoxelSpace(invert(dir) * depth + pos)  (Camera.java:31-34, Util.java:11-18)
  void getRayPickLocation(float depth, int out[3]) const {
    for (int i = 0; i < 3; i++) out[i] = (int)(((-dir[i]) * depth + pos[i] - 1) * Constants::WORLD_SIZE);
  }

 private:
  // m is row-major 3x3: m[r*3+c]
  static float jsin(float a) { return (float)std::sin((double)a); }
  static float jcos(float sin, float angle) {  // org.joml.Math.cosFromSin (non-fast path)
    const float PI_f = (float)M_PI, PI2 = PI_f * 2.0f, PIHalf = PI_f * 0.5f;
    float cos = std::sqrt(1.0f - sin * sin);
    float a = angle + PIHalf;
    float b = a - (int)(a / PI2) * PI2;
    if (b < 0.0f) b = PI2 + b;
    if (b >= PI_f) return -cos;
    return cos;
  }
  static void rotationY(float ang, float *m) {
    float s = jsin(ang), c = jcos(s, ang);
    const float r[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    memcpy(m, r, sizeof r);
  }
  static void rotationAxis(float ang, float x, float y, float z, float *m) {  // Matrix4f.rotation(angle, x, y, z)
    float s = jsin(ang), c = jcos(s, ang), C = 1.0f - c;
    float xy = x * y, xz = x * z, yz = y * z;
    const float r[9] = {c + x * x * C, xy * C - z * s, xz * C + y * s,
                        xy * C + z * s, c + y * y * C, yz * C - x * s,
                        xz * C - y * s, yz * C + x * s, c + z * z * C};
    memcpy(m, r, sizeof r);
  }
  static void mul3(const float *a, const float *b, float *o) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
  }
  static void apply(const float *m, float *v) {
    const float x = v[0], y = v[1], z = v[2];
    v[0] = m[0] * x + m[1] * y + m[2] * z;
    v[1] = m[3] * x + m[4] * y + m[5] * z;
    v[2] = m[6] * x + m[7] * y + m[8] * z;
  }
};

// ---------------------------------------------------------------------------------------------
// Renderer.java on top of libsvohip.so.  The GL objects become fields of one svo_ctx:
//   SSBO binding 7      -> the pool          uniforms 1..4, 8 -> camera vectors
//   uniforms 5, 6, 9, 11 -> frame parameters  images 0 / 1     -> colour / depth
class Renderer {
 public:
  struct Shader {        // Renderer.java:18-28
    std::string name;
    int computeProgram;  // 1 = the SVO trace path; other shaders of the reference are out of scope
    int computeProgramShader;
  };
  static Renderer &getInstance() { static Renderer r; return r; }   // :34-36

  // The reference compiles GLSL here and never checks the status (Renderer.java:43-54).  The HIP path is
  // compiled ahead of time: "svotrace" binds to it, anything else yields an inert program.
  Shader *addShader(const std::string &name, const std::string &path) {
    ensure();
    const bool trace = path.find("svotrace") != std::string::npos || name == "svotrace";
    shaders.push_back(Shader{name, trace ? 1 : 0, trace ? 1 : 0});
    return &shaders.back();
  }
  void setUniformInteger(int location, int value) {   // :56-58 (glUniform1i)
    switch (location) {
      case 5: frameNumber = value; break;     // svotrace.comp:10
      case 6: renderMode = value; break;      // :11
      case 9: bufferEnd = value; break;       // :17
      case 11: useBeam = value; break;        // :18
      default: break;                         // GL ignores unknown locations silently (error polled later)
    }
  }
  void setUniform3fv(int location, const float *v) {  // raw glUniform3fv in Main.java:269-273
    switch (location) {
      case 8: memcpy(cam + 0, v, 12); break;
      case 1: memcpy(cam + 3, v, 12); break;
      case 2: memcpy(cam + 6, v, 12); break;
      case 3: memcpy(cam + 9, v, 12); break;
      case 4: memcpy(cam + 12, v, 12); break;
      default: break;
    }
  }
  void useProgram(Shader *s) { current = s; }          // :114-116
  // glDispatchCompute(gx, gy, gz) + glMemoryBarrier: the image is gx*8 x gy*8 unless setImageSize was called.  Both GL calls
  // return at once (Renderer.java:118-121) and so does this: the frame is enqueued (svo_dispatch_async); the wait is where the
  // reference has it -- the next read-back (readFramebuffer / readDepth / readHits / readDepthPixel = next frame's
  // glGetTexImage, Main.java:132-146) -- and in every SSBO call.
  void dispatchCompute(Shader *s, int numGroupsX, int numGroupsY, int numGroupsZ) {   // :118-121
    (void)numGroupsZ;
    if (prepareDispatch(s, numGroupsX, numGroupsY)) check(svo_dispatch_async(ctx));
  }
  // the same, returning when the frame is complete (svo_dispatch; svo_get_stats().last_dispatch_ms = its GPU time)
  void dispatchComputeAndWait(Shader *s, int numGroupsX, int numGroupsY, int numGroupsZ) {
    (void)numGroupsZ;
    if (prepareDispatch(s, numGroupsX, numGroupsY)) check(svo_dispatch(ctx));
  }
  void setImageSize(int w, int h) { width = w; height = h; }   // glTexStorage2D in Main.java:69,76
  void addSSBO(int bindIndex, const uint8_t *data, size_t nbytes) {   // :123-129
    if (!ensure() || bindIndex != 7) return;
    check(svo_pool_upload(ctx, data, nbytes));
  }
  void updateSSBO(int bindIndex, const uint8_t *data, size_t nbytes) { addSSBO(bindIndex, data, nbytes); }  // :131-134
  void updateSSBO(int bindIndex, const uint8_t *data, int start, int end) {   // :136-146
    if (start >= end) { printf("Update SSBO error: Invalid parameters.\n"); return; }
    if (!ensure() || bindIndex != 7) return;
    check(svo_pool_update(ctx, data, (uint64_t)start, (uint64_t)end));
  }
  void getSSBO(uint8_t *out, size_t nbytes) { if (ensure()) check(svo_pool_download(ctx, out, nbytes)); }  // :148-150
  Shader *getShaderByName(const std::string &name) {   // :152-158
    for (auto &s : shaders) if (s.name == name) return &s;
    return nullptr;
  }
  void printGLErrors() {   // :160-165
    if (!lastErr.empty()) { printf("HIP ERR: %s\n", lastErr.c_str()); lastErr.clear(); }
  }
  // glGetTexImage of image 0 / 1 (Main.java:132-146)
  void readFramebuffer(void *rgba8) { if (ensure()) check(svo_read_color(ctx, rgba8)); }
  void readDepth(float *d) { if (ensure()) check(svo_read_depth(ctx, d)); }
  void readHits(svo_hit *h) { if (ensure()) check(svo_read_hits(ctx, h)); }
  // the crosshair pick of Main.updateEarly (Main.java:132-146) without the full-frame read-back.  At the pick position
  // (setPick; default the image centre = Main's crosshair) the value comes from pinned host memory as soon as the lane that
  // renders the pixel has stored it -- the frame need not have ended (round 6); anywhere else it waits for the frame.
  float readDepthPixel(int x, int y) { float d = 0.0f; if (ensure()) check(svo_read_pixel(ctx, x, y, nullptr, &d, nullptr)); return d; }
  void setPick(int x, int y) { if (ensure()) check(svo_set_pick(ctx, x, y)); }          // x < 0: no pick, every read-back waits
  void setOverlap(int sets) { if (ensure()) check(svo_set_overlap(ctx, sets)); }   // {stream, image} sets dispatchCompute takes turns on: 0 = one, 1 = default (4), 2 .. 8
  // dormant shader features (svotrace.comp:444, 500-504, 668-670); defaults = the live behaviour
  void setPathOptions(int bounces_, uint32_t mirrorMask_, int spp_) { bounces = bounces_; mirrorMask = mirrorMask_; spp = spp_; }
  svo_ctx *context() { ensure(); return ctx; }
  bool hasError() const { return !lastErr.empty(); }
  ~Renderer() { if (ctx) svo_destroy(ctx); }

 private:
  Renderer() { shaders.reserve(16); }
  bool prepareDispatch(Shader *s, int numGroupsX, int numGroupsY) {
    if (!ensure() || !s || s->computeProgram != 1) return false;
    if (width == 0) { width = numGroupsX * Constants::COMPUTE_GROUP_SIZE; height = numGroupsY * Constants::COMPUTE_GROUP_SIZE; }
    check(svo_resize(ctx, width, height));
    check(svo_set_camera(ctx, cam, cam + 3, cam + 6, cam + 9, cam + 12));
    check(svo_set_params(ctx, frameNumber, renderMode, bufferEnd, useBeam, bounces, mirrorMask, spp));
    return true;
  }
  bool ensure() {
    if (ctx) return true;
    if (svo_create(0, &ctx) != SVO_OK) { lastErr = "svo_create failed (no MI355X visible)"; ctx = nullptr; return false; }
    return true;
  }
  void check(int rc) { if (rc != SVO_OK) lastErr = svo_last_error(ctx); }
  std::vector<Shader> shaders;
  Shader *current = nullptr;
  svo_ctx *ctx = nullptr;
  std::string lastErr;
  float cam[15] = {1.5f, 1.5f, 2.0f, -1.6f, -0.9f, -1, -1.6f, 0.9f, -1, 1.6f, -0.9f, -1, 1.6f, 0.9f, -1};
  int frameNumber = 1, renderMode = 2, bufferEnd = 0, useBeam = 0, bounces = 2, spp = 1;
  uint32_t mirrorMask = 0;
  int width = 0, height = 0;
};

}  // namespace host
}  // namespace svo
