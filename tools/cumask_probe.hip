// cumask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (gfx950: 8 XCDs x 32 CUs)
// hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(unsigned *out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the wave alive for a while so that the grid spreads over every CU the stream may use
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 7u; out[2 * blockIdx.x + 1] = hwid; }
}
static void run(const char *tag, hipStream_t s) {
  const int blocks = 8192;
  unsigned *d; hipMalloc(&d, blocks * 8);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  int cnt[8][64]; memset(cnt, 0, sizeof cnt);
  for (int i = 0; i < blocks; i++) {
    const unsigned x = h[2 * i], id = h[2 * i + 1];
    const unsigned cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;   // HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
    cnt[x][(se << 4 | cu) & 63]++;
    (void)sh;
  }
  printf("%s: CUs in use per XCD:", tag);
  int total = 0;
  for (int x = 0; x < 8; x++) { int n = 0; for (int c = 0; c < 64; c++) n += cnt[x][c] > 0; printf(" %d", n); total += n; }
  printf("  (total %d)\n", total);
  hipFree(d);
}
int main() {
  hipStream_t s0; hipStreamCreate(&s0); run("unmasked", s0);
  for (int off : {8, 16, 32}) {
    uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    for (int b = 256 - off; b < 256; b++) mask[b >> 5] &= ~(1u << (b & 31));   // clear the top `off` bits
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    char tag[64]; snprintf(tag, sizeof tag, "top %d bits cleared (%s)", off, hipGetErrorString(e));
    if (e == hipSuccess) run(tag, s);
    else printf("%s\n", tag);
  }
  {   // bits 0..7 cleared
    uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    mask[0] &= ~0xffu;
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e == hipSuccess) run("bits 0..7 cleared", s);
  }
  {   // every 8th bit cleared in the first 64
    uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    mask[0] = 0xfefefefeu; mask[1] = 0xfefefefeu;
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e == hipSuccess) run("bits 0,8,..,56 cleared", s);
  }
  return 0;
}
