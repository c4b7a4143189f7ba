// calib_td.hip -- cost of one wave-level divergent buffer load in the texture path (TA/TD), by width and alignment.
// Every lane reads its own cache line from a small (L1/L2-resident) table; loads are independent, so the loop
// measures issue/return throughput, not latency.   hipcc --offload-arch=gfx950 -O3 -o calib_td calib_td.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int kKind>
__global__ __launch_bounds__(64) void k(const unsigned char *base, unsigned len, unsigned mask, unsigned misalign, int iters, unsigned *out) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)len, 0x00020000);
  unsigned lane = threadIdx.x;
  unsigned a = (lane * 2654435761u + blockIdx.x * 40503u);
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    a = a * 1664525u + 1013904223u;
    unsigned off = ((a >> 8) & mask & ~127u) + ((lane * 4u) & 124u) + misalign * (1u + (lane & 2u));   // own 128-B line, dword slot by lane
    if (kKind == 0) acc += __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    if (kKind == 1) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); acc += v.x ^ v.y; }
    if (kKind == 2) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off, 0, 0); acc += v.x ^ v.y ^ v.z; }
    if (kKind == 3) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (kKind == 4) acc += __builtin_amdgcn_raw_buffer_load_b8(rs, (int)off, 0, 0);
    if (kKind == 5) acc += __builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int kKind>
static void run(const char *name, const unsigned char *d, unsigned len, unsigned mask, unsigned misalign, unsigned *out) {
  const int blocks = 256 * 16, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, len, mask, misalign, 50, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<kKind>, dim3(blocks), dim3(64), 0, 0, d, len, mask, misalign, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wl = (double)blocks * iters;            // wave-level loads
  double cyc_per_cu = ms * 1e-3 * 2.4e9 / (wl / 256.0);
  printf("%-10s footprint %8u B misalign %u : %.3f ms, %.1f cycles per wave-load per CU (at 2.4 GHz)\n", name, mask + 1, misalign, ms, cyc_per_cu);
}

int main() {
  const unsigned len = 256u << 20;
  unsigned char *d; unsigned *out;
  hipMalloc(&d, len + 64); hipMemset(d, 1, len + 64); hipMalloc(&out, 256 * 16 * 64 * 4);
  for (unsigned fp : {8192u, 1u << 20, 64u << 20}) {
    for (unsigned mis : {0u, 1u}) {
      run<4>("ubyte", d, len, fp - 1, mis, out);
      run<5>("ushort", d, len, fp - 1, mis, out);
      run<0>("dword", d, len, fp - 1, mis, out);
      run<1>("dwordx2", d, len, fp - 1, mis, out);
      run<2>("dwordx3", d, len, fp - 1, mis, out);
      run<3>("dwordx4", d, len, fp - 1, mis, out);
    }
  }
  return 0;
}
