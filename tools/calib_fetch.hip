// calib_fetch.hip -- calibrate rocprofv3's FETCH_SIZE for THIS path's access pattern:
// one unaligned 8-byte buffer load per lane at scattered addresses (like a child-record
// fetch that misses every cache).  Every load touches a distinct 4 KiB page of a buffer far
// larger than L2 + Infinity Cache, so the bytes HBM must deliver are known:
//   loads x (fetch granule).  Comparing with the counter tells the granule the counter
// implies (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void scattered_8b(const uint8_t* base, unsigned long long nbytes, unsigned nloads, unsigned* sink) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nloads) return;
  // bit-reversal-ish scramble so that neighbouring lanes are far apart
  unsigned j = (i * 2654435761u) % nloads;
  unsigned long long off = (unsigned long long)j * 4096ull + 13ull;
  unsigned long long v;
  __builtin_memcpy(&v, base + off, 8);
  if (v == 0x123456789abcdefull) sink[0] = 1;
}
// same but 4 loads per lane inside one 128-byte line -> counts per-line behaviour
__global__ void scattered_line(const uint8_t* base, unsigned nloads, unsigned* sink) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nloads) return;
  unsigned j = (i * 2654435761u) % nloads;
  unsigned long long off = (unsigned long long)j * 4096ull;
  unsigned long long a, b;
  __builtin_memcpy(&a, base + off + 5, 8);
  __builtin_memcpy(&b, base + off + 64 + 5, 8);
  if ((a ^ b) == 0x123456789abcdefull) sink[0] = 1;
}
int main() {
  const unsigned nloads = 2u << 20;                 // 2 Mi loads
  const unsigned long long nbytes = (unsigned long long)nloads * 4096ull;  // 8 GiB
  uint8_t* buf; unsigned* sink;
  if (hipMalloc(&buf, nbytes + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, nbytes + 4096);
  hipMemset(sink, 0, 4);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(scattered_8b, dim3(nloads / 256), dim3(256), 0, 0, buf, nbytes, nloads, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(scattered_line, dim3(nloads / 256), dim3(256), 0, 0, buf, nloads, sink);
    hipDeviceSynchronize();
  }
  printf("loads per launch: %u ; if every load fetched 32 / 64 / 128 B: %.1f / %.1f / %.1f MB\n", nloads,
         nloads * 32.0 / 1e6, nloads * 64.0 / 1e6, nloads * 128.0 / 1e6);
  return 0;
}
