// svo_descword.h -- the two words of an interior descriptor (svo_derive.hip.h), as plain functions: the table builder and its
// refresh use them on the device, tests/test_descword.py compiles them for the host and checks every combination of children.
#pragma once
#include <cstdint>
#ifndef __host__
#define __host__
#define __device__
#define SVO_DESCWORD_PLAIN 1
#endif

namespace svo {
namespace derive {

constexpr uint32_t kGroupBias = 64u;   // desc.x = 8 * (index of the first child descriptor) - 64: + 8 * (8 | rank) is the child's

// the second word of a descriptor from the masks of its children (has is a subset of ne), and back
__host__ __device__ inline uint32_t desc_word(uint32_t m_ne, uint32_t m_has) {
  uint32_t w = 0, rank = 0;
  for (uint32_t c = 0; c < 8; c++) {
    if ((m_has >> c) & 1u) w |= (8u | rank++) << (4u * c);
    else if ((m_ne >> c) & 1u) w |= 1u << (4u * c);
  }
  return w;
}
__host__ __device__ inline uint32_t desc_has(uint32_t w) {
  uint32_t m = 0;
  for (uint32_t c = 0; c < 8; c++) m |= ((w >> (4u * c + 3u)) & 1u) << c;
  return m;
}
__host__ __device__ inline uint32_t desc_first(uint32_t x) { return (x + kGroupBias) >> 3; }   // index of the first child descriptor
__host__ __device__ inline uint32_t desc_base(uint32_t first) { return first * 8u - kGroupBias; }

}  // namespace derive
}  // namespace svo
#ifdef SVO_DESCWORD_PLAIN
#undef __host__
#undef __device__
#undef SVO_DESCWORD_PLAIN
#endif
