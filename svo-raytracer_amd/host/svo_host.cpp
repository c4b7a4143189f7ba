// svo_host.cpp -- C entry points over the C++ host mirror (svo_host.hpp), so that tests and
// tools written in Python can drive Renderer / Camera / Octree the way Main.java does.
#include "svo_host.hpp"

using namespace svo::host;

extern "C" {

void *svoh_camera_new() { return new Camera(); }
void svoh_camera_free(void *c) { delete (Camera *)c; }
void svoh_camera_set_pos(void *c, float x, float y, float z) { ((Camera *)c)->setPos(x, y, z); }
void svoh_camera_set_speed(void *c, float s) { ((Camera *)c)->setSpeed(s); }
void svoh_camera_rotate(void *c, float x, float y, float z) { ((Camera *)c)->rotate(x, y, z); }
void svoh_camera_strafe(void *c, float f, float s) { ((Camera *)c)->strafe(f, s); }
void svoh_camera_get_uniform(void *c, float *out15) { ((Camera *)c)->getUniform(out15); }
void svoh_camera_get_dir(void *c, float *out3) { memcpy(out3, ((Camera *)c)->dir, 12); }
void svoh_camera_get_rot(void *c, float *out3) { memcpy(out3, ((Camera *)c)->rot, 12); }
void svoh_camera_pick(void *c, float depth, int *out3) { ((Camera *)c)->getRayPickLocation(depth, out3); }

void *svoh_octree_new(int mem_size_kb) { return new Octree(mem_size_kb); }
void svoh_octree_free(void *o) { delete (Octree *)o; }
void svoh_octree_adopt(void *o, const uint8_t *pool, uint64_t n) { ((Octree *)o)->adopt(pool, (size_t)n); }
int svoh_octree_mem_offset(void *o) { return ((Octree *)o)->memOffset; }
uint8_t *svoh_octree_buffer(void *o) { return ((Octree *)o)->getByteBuffer(); }
void svoh_octree_write(void *o, const char *path) { ((Octree *)o)->writeBufferToFile(path); }
void svoh_octree_read(void *o, const char *path) { ((Octree *)o)->readBufferFromFile(path); }
int svoh_octree_create_interior(void *o, int v) { return ((Octree *)o)->createInteriorNode((uint8_t)v); }
int svoh_octree_create_surface_leaf(void *o, int v, int n) { return ((Octree *)o)->createSurfaceLeafNode((uint8_t)v, (uint16_t)n); }
int svoh_octree_create_nonsurface_leaf(void *o, int v) { return ((Octree *)o)->createNonSurfaceLeafNode((uint8_t)v); }
int svoh_octree_create_subdividable_leaf(void *o, int v) { return ((Octree *)o)->createSubdividableLeafNode((uint8_t)v); }
void svoh_octree_set_child_pointer(void *o, int p, int c) { ((Octree *)o)->setChildPointer(p, c); }
int svoh_octree_get_child_pointer(void *o, int p) { return ((Octree *)o)->getChildPointer(p); }
void svoh_octree_set_leaf_mask(void *o, int p, int m) { ((Octree *)o)->setLeafMask(p, (uint16_t)m); }
int svoh_octree_get_leaf_mask(void *o, int p) { return ((Octree *)o)->getLeafMask(p); }

// One frame the way the reference drives it: Main.preRun (:102-125) + Main.updateEarly (:267-285).
// Returns 0 when the frame was rendered, 1 if the renderer reported an error.
int svoh_render_frame(void *octree, void *camera, int width, int height, int frame_number, int render_mode,
                      uint8_t *rgba, float *depth) {
  Renderer &renderer = Renderer::getInstance();
  Octree *oct = (Octree *)octree;
  Camera *cam = (Camera *)camera;
  Renderer::Shader *traceShader = renderer.getShaderByName("svotrace");
  if (!traceShader) traceShader = renderer.addShader("svotrace", "src/shaders/svotrace.comp");
  renderer.setImageSize(width, height);
  renderer.addSSBO(7, oct->getByteBuffer(), (size_t)oct->memOffset);
  renderer.useProgram(traceShader);
  renderer.setUniform3fv(8, cam->pos);
  renderer.setUniform3fv(1, cam->l1);
  renderer.setUniform3fv(2, cam->l2);
  renderer.setUniform3fv(3, cam->r1);
  renderer.setUniform3fv(4, cam->r2);
  renderer.setUniformInteger(5, frame_number);
  renderer.setUniformInteger(6, render_mode);
  renderer.setUniformInteger(9, oct->memOffset);
  renderer.setUniformInteger(11, 0);
  const int gx = (width + 7) / 8, gy = (height + 7) / 8;
  renderer.dispatchCompute(traceShader, gx, gy, 1);
  if (renderer.hasError()) { renderer.printGLErrors(); return 1; }
  if (rgba) renderer.readFramebuffer(rgba);
  if (depth) renderer.readDepth(depth);
  if (renderer.hasError()) { renderer.printGLErrors(); return 1; }
  return 0;
}

// Main.updateEarly for n frames, as the reference's loop runs it (Main.java:130-289): per frame the crosshair depth of the frame
// before (readDepthPixel at the image centre -- Main reads it at the top of updateEarly, :132-146), the camera's motion
// (rotate + strafe when `move`; any motion resets frameNumber to 0, :225-233), frameNumber++ (:275), the uniforms, the dispatch.
// picks[i] = the crosshair depth of frame i (read at the top of frame i + 1's turn, the last one after the loop); cams = the
// 15 uniform floats of every frame (n x 15) and frame_numbers what each frame was rendered with, for the checker.
// Returns 0, or 1 if the renderer reported an error.
int svoh_render_loop(void *octree, void *camera, int width, int height, int nframes, int render_mode, int move, float *picks,
                     float *cams, int *frame_numbers, uint8_t *last_rgba, float *last_depth) {
  Renderer &renderer = Renderer::getInstance();
  Octree *oct = (Octree *)octree;
  Camera *cam = (Camera *)camera;
  Renderer::Shader *traceShader = renderer.getShaderByName("svotrace");
  if (!traceShader) traceShader = renderer.addShader("svotrace", "src/shaders/svotrace.comp");
  renderer.setImageSize(width, height);
  renderer.addSSBO(7, oct->getByteBuffer(), (size_t)oct->memOffset);
  int frameNumber = 1;                                        // Main.java:16
  const int gx = (width + 7) / 8, gy = (height + 7) / 8;
  for (int i = 0; i < nframes; i++) {
    if (i > 0) picks[i - 1] = renderer.readDepthPixel(width / 2, height / 2);   // :132-146, the frame dispatched last turn
    if (move) { cam->rotate(0.0f, 0.01f, 0.0f); cam->strafe(0.002f, 0.001f); frameNumber = 0; }   // :161-236
    renderer.useProgram(traceShader);
    renderer.setUniform3fv(8, cam->pos);
    renderer.setUniform3fv(1, cam->l1);
    renderer.setUniform3fv(2, cam->l2);
    renderer.setUniform3fv(3, cam->r1);
    renderer.setUniform3fv(4, cam->r2);
    frameNumber++;
    renderer.setUniformInteger(5, frameNumber);
    renderer.setUniformInteger(6, render_mode);
    renderer.setUniformInteger(9, oct->memOffset);
    renderer.setUniformInteger(11, 0);
    renderer.dispatchCompute(traceShader, gx, gy, 1);
    if (cams) cam->getUniform(cams + 15 * (size_t)i);
    if (frame_numbers) frame_numbers[i] = frameNumber;
    if (renderer.hasError()) { renderer.printGLErrors(); return 1; }
  }
  if (nframes > 0) picks[nframes - 1] = renderer.readDepthPixel(width / 2, height / 2);
  if (last_rgba) renderer.readFramebuffer(last_rgba);
  if (last_depth) renderer.readDepth(last_depth);
  if (renderer.hasError()) { renderer.printGLErrors(); return 1; }
  return 0;
}

}  // extern "C"
