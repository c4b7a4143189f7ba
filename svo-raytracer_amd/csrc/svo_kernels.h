// svo_kernels.h -- frame description shared by the kernels and the C-ABI host code.
#pragma once
#include <stdint.h>

namespace svo {

// Everything the reference passes as uniforms (svotrace.comp:5-18, Main.java:269-283)
// plus the image size and the row range this GPU renders.
struct Frame {
  float cam[15];        // pos, l1, l2, r1, r2
  int32_t width, height;
  int32_t y0, y1;       // first pixel row of this launch's first tile row; rows >= y1 are not rendered
  int32_t row_step;     // distance, in tile rows, between consecutive tile rows of this launch (1 = a band)
  int32_t out_y0;       // output row of the launch's first tile row (outputs of a launch are packed)
  int32_t frame_number, render_mode;
  int32_t bounces, spp;
  uint32_t mirror_mask;
  uint32_t pool_len;
  uint32_t dword0;      // first dword of the pool (debug square colour, svotrace.comp:696-698)
  int32_t tiles_x, tiles_y, ntiles;
  int32_t write_hits;
  // beam pre-pass (useBeamOptimization, Main.java:257-283): start distance per 4x4 pixel block, or off
  int32_t use_beam, beam_w;
  const float *beam;
  // cross-frame accumulation (commented out in the reference, svotrace.comp:712-719): blend with the image the
  // previous frame left in the colour buffer
  int32_t progressive;
  // several consecutive frames of one camera in one launch (svo_set_batch): frame k renders frameNumber + k into the
  // outputs at element offset k * frame_stride.  1 = the reference's one dispatch per frame.
  int32_t batch;
  uint32_t frame_stride;
  // progressive && seq > 1: this launch carries `seq` consecutive frames of the cross-frame accumulation -- frameNumber,
  // frameNumber + 1, ... blended into ONE image in frame order, as seq dispatches of the reference's loop would
  // (Main.java:275 + svotrace.comp:712-719).  1 = one frame per dispatch.
  int32_t seq;
};

// What changes from frame to frame of a batch whose frames carry their own camera (svo_ring_submit_cams): the five camera
// vectors and frameNumber (Main.updateEarly moves the camera and resets frameNumber on motion, Main.java:161-236, 275).
struct FrameVar {
  float cam[15];
  int32_t frame_number;
};
static_assert(sizeof(FrameVar) == 64, "FrameVar is one 64-byte scalar load");

// device-side counters of a counted frame
struct DeviceCounters {
  unsigned long long pixels, rays, nan_rays, iterations, alg_bytes;
  unsigned int max_iter, pad;
};

// tile row `ty` of the launch, row `ly` inside the tile -> pixel row in the frame / row in the output images
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int frame_gy(const Frame &f, int ty, int ly) { return f.y0 + ty * 8 * f.row_step + ly; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int frame_oy(const Frame &f, int ty, int ly) { return f.out_y0 + ty * 8 + ly; }

}  // namespace svo
