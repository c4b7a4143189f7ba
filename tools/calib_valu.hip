// calib_valu.hip -- VALU issue cost of a wave64 instruction on gfx950, by EXEC pattern.
// hipcc --offload-arch=gfx950 -O3 -o calib_valu calib_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long mask, int iters) {
  float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
  unsigned long long m = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
  unsigned long long sv;
  for (int i = 0; i < iters; i++) {
    asm volatile(
        "s_mov_b64 %[sv], exec\n\ts_mov_b64 exec, %[m]\n\t"
        ".rept 64\n\t"
        "v_add_f32 %[a], %[a], %[b]\n\tv_add_f32 %[c], %[c], %[b]\n\tv_xor_b32 %[d], %[d], %[b]\n\tv_mul_f32 %[b], 1.0, %[b]\n\t"
        ".endr\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c), [d] "+v"(d), [sv] "=&s"(sv) : [m] "s"(m));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
int main() {
  float *out; hipMalloc(&out, 256 * 32 * 64 * 4);
  const int iters = 400;
  for (int wpc : {4, 8, 20}) {
    for (unsigned long long mask : {~0ull, 0xffffffffull, 0xffffull, 0x1ull, 0x0001000100010001ull, 0x5555555555555555ull, 0ull}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, out, mask, 10);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, out, mask, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_simd = (double)wpc / 4.0 * iters * 256.0;
      printf("waves/CU %2d exec %016llx : %.3f ms, %.2f cycles per VALU instruction per SIMD (2.4 GHz)\n", wpc, mask, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
  }
  return 0;
}
