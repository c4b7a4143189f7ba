// queue_probe.hip -- which HIP streams share a hardware queue?  (hipcc --offload-arch=gfx950 -O2 -o tools/queue_probe tools/queue_probe.hip)
// Creates N non-blocking streams in a row (as svo_ring_create does), then for every pair launches one single-workgroup
// spinning kernel on each and times the pair: two streams on one hardware queue run their kernels one after the other
// (2 x the spin), streams on queues of their own run them side by side (1 x).  GPU_MAX_HW_QUEUES is read by the runtime at start.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(unsigned long long cycles, unsigned long long *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
  if (out) *out = t0;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  const int pre = argc > 2 ? atoi(argv[2]) : 1;     // streams created (and kept) in front, like a context's own stream
  std::vector<hipStream_t> front(pre), s(n);
  for (auto &x : front) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  const unsigned long long spin_ticks = 200000;   // 2 ms
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], 1000ull, nullptr);
  hipDeviceSynchronize();
  printf("GPU_MAX_HW_QUEUES=%s, %d stream(s) in front, %d ring streams: pairs that serialise\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", pre, n);
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], spin_ticks, nullptr);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[j], spin_ticks, nullptr);
      hipDeviceSynchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > 3.0) printf("  (%d, %d): %.2f ms\n", i, j, ms);
    }
  printf("done\n");
  return 0;
}
