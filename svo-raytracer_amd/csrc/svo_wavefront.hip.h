// svo_wavefront.hip.h -- pipeline 1 (placeholder until the wavefront kernels land)
#pragma once
#include "svo_device.h"
#include "svo_kernels.h"
namespace svo {
struct WavefrontBuffers { void *queue = nullptr; };
inline void wavefront_free(WavefrontBuffers &) {}
inline int wavefront_launch(WavefrontBuffers &, const uint8_t *, const Frame &, uint32_t *, float *, uint4 *, hipStream_t) {
  return (int)hipErrorNotSupported;
}
}  // namespace svo
