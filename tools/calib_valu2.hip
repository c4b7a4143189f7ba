// calib_valu2.hip -- issue cost of the instruction kinds the traversal loop is made of: gfx950, 6 waves per SIMD (the
// kernel's occupancy), eight independent instructions of ONE kind per body, cycles per instruction per SIMD at 2.4 GHz.
// hipcc --offload-arch=gfx950 -O3 -o calib_valu2 calib_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(x0, x1, x2, x3, x4, x5, x6, x7) x0 "\n\t" x1 "\n\t" x2 "\n\t" x3 "\n\t" x4 "\n\t" x5 "\n\t" x6 "\n\t" x7 "\n\t"
#define KERNEL(NAME, BODY)                                                                                          \
  __global__ __launch_bounds__(64) void NAME(float *out, int iters) {                                               \
    float a0 = threadIdx.x, a1 = 1.5f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, b = 1.25f, c = 0.5f;  \
    unsigned long long s0 = 1, s1 = 2, s2 = 3, s3 = 4;                                                              \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                           \
    f2 p0 = {a0, b}, p1 = {a1, b}, p2 = {a2, b}, p3 = {a3, c}, q = {b, c};                                          \
    for (int i = 0; i < iters; i++) {                                                                               \
      asm volatile(".rept 32\n\t" BODY ".endr\n\t"                                                                  \
                   : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5),      \
                     [a6] "+v"(a6), [a7] "+v"(a7), [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3),      \
                     [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3)                                     \
                   : [b] "v"(b), [c] "v"(c), [q] "v"(q)                                                             \
                   : "vcc", "scc");                                                                                 \
    }                                                                                                               \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(s0 + s1 + s2 + s3);  \
  }
#define EIGHT(fmt) OPS(fmt("%[a0]"), fmt("%[a1]"), fmt("%[a2]"), fmt("%[a3]"), fmt("%[a4]"), fmt("%[a5]"), fmt("%[a6]"), fmt("%[a7]"))
#define F_ADD32(r) "v_add_f32 " r ", " r ", %[b]"
#define F_ADD64(r) "v_add_f32_e64 " r ", " r ", %[b]"
#define F_XOR(r) "v_xor_b32 " r ", " r ", %[b]"
#define F_MOV(r) "v_mov_b32 " r ", %[b]"
#define F_MIN(r) "v_min_f32 " r ", " r ", %[b]"
#define F_MIN3(r) "v_min3_f32 " r ", " r ", %[b], %[c]"
#define F_FMA(r) "v_fma_f32 " r ", " r ", %[b], %[c]"
#define F_BFE(r) "v_bfe_u32 " r ", " r ", %[b], 1"
#define F_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 1, %[b]"
#define F_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 3, %[b]"
#define F_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %[b]"
#define F_FFBH(r) "v_ffbh_u32 " r ", " r
#define F_BITOP3(r) "v_bitop3_b32 " r ", " r ", %[b], %[c] bitop3:0x90"
#define F_BFM(r) "v_bfm_b32 " r ", " r ", 8"
#define F_CND32(r) "v_cndmask_b32 " r ", " r ", %[b], vcc"
#define F_CND64(r) "v_cndmask_b32_e64 " r ", " r ", %[b], %[s0]"
#define F_CNDC(r) "v_cndmask_b32_e64 " r ", 0, %[b], %[s1]"
#define F_CND64V(r) "v_cndmask_b32_e64 " r ", " r ", %[b], vcc"
#define F_LSHR(r) "v_lshrrev_b32 " r ", 1, " r
#define F_LSHLE64(r) "v_lshlrev_b32_e64 " r ", " r ", -1"
#define F_AND(r) "v_and_b32 " r ", " r ", %[b]"
#define F_OR3(r) "v_or3_b32 " r ", " r ", %[b], 1"
#define F_SUBF(r) "v_sub_f32 " r ", " r ", %[b]"
#define F_MINU(r) "v_min_u32 " r ", 11, " r
#define F_LSHRV(r) "v_lshrrev_b32 " r ", %[b], " r
#define F_LSHLV(r) "v_lshlrev_b32 " r ", %[b], " r
#define F_MAD24(r) "v_mad_u32_u24 " r ", " r ", 8, %[b]"
#define F_MUL24(r) "v_mul_u32_u24 " r ", 7, " r
#define F_BFI(r) "v_bfi_b32 " r ", " r ", %[b], %[c]"
#define F_ANDOR(r) "v_and_or_b32 " r ", " r ", %[b], %[c]"
#define F_ADD3(r) "v_add3_u32 " r ", " r ", %[b], %[c]"
#define F_PERM(r) "v_perm_b32 " r ", " r ", %[b], %[c]"
#define F_MAXF(r) "v_max_f32 " r ", " r ", %[b]"
#define F_MED3(r) "v_med3_f32 " r ", " r ", %[b], %[c]"
#define F_SUBU(r) "v_sub_u32 " r ", 20, " r
#define F_OR(r) "v_or_b32 " r ", " r ", %[b]"
#define F_ADDU(r) "v_add_u32 " r ", 1, " r
#define F_MULF(r) "v_mul_f32 " r ", 0.5, " r
KERNEL(k_add32, EIGHT(F_ADD32)) KERNEL(k_add64, EIGHT(F_ADD64)) KERNEL(k_xor, EIGHT(F_XOR)) KERNEL(k_mov, EIGHT(F_MOV))
KERNEL(k_min, EIGHT(F_MIN)) KERNEL(k_min3, EIGHT(F_MIN3)) KERNEL(k_fma, EIGHT(F_FMA)) KERNEL(k_bfe, EIGHT(F_BFE))
KERNEL(k_lshlor, EIGHT(F_LSHLOR)) KERNEL(k_lshladd, EIGHT(F_LSHLADD)) KERNEL(k_bcnt, EIGHT(F_BCNT)) KERNEL(k_ffbh, EIGHT(F_FFBH))
KERNEL(k_bitop3, EIGHT(F_BITOP3)) KERNEL(k_bfm, EIGHT(F_BFM)) KERNEL(k_cnd32, EIGHT(F_CND32)) KERNEL(k_cnd64, EIGHT(F_CND64))
KERNEL(k_cndc, EIGHT(F_CNDC)) KERNEL(k_cnd64v, EIGHT(F_CND64V)) KERNEL(k_lshr, EIGHT(F_LSHR)) KERNEL(k_lshle64, EIGHT(F_LSHLE64)) KERNEL(k_and, EIGHT(F_AND)) KERNEL(k_or3, EIGHT(F_OR3)) KERNEL(k_subf, EIGHT(F_SUBF)) KERNEL(k_minu, EIGHT(F_MINU)) KERNEL(k_addu, EIGHT(F_ADDU)) KERNEL(k_mulf, EIGHT(F_MULF))
KERNEL(k_cmpvcc, OPS("v_cmp_lt_f32 vcc, %[a0], %[b]", "v_cmp_lt_f32 vcc, %[a1], %[b]", "v_cmp_lt_f32 vcc, %[a2], %[b]", "v_cmp_lt_f32 vcc, %[a3], %[b]",
                     "v_cmp_lt_f32 vcc, %[a4], %[b]", "v_cmp_lt_f32 vcc, %[a5], %[b]", "v_cmp_lt_f32 vcc, %[a6], %[b]", "v_cmp_lt_f32 vcc, %[a7], %[b]"))
KERNEL(k_cmpsgpr, OPS("v_cmp_lt_f32_e64 %[s0], %[a0], %[b]", "v_cmp_lt_f32_e64 %[s1], %[a1], %[b]", "v_cmp_lt_f32_e64 %[s2], %[a2], %[b]",
                      "v_cmp_lt_f32_e64 %[s3], %[a3], %[b]", "v_cmp_lt_f32_e64 %[s0], %[a4], %[b]", "v_cmp_lt_f32_e64 %[s1], %[a5], %[b]",
                      "v_cmp_lt_f32_e64 %[s2], %[a6], %[b]", "v_cmp_lt_f32_e64 %[s3], %[a7], %[b]"))
KERNEL(k_cmpsdwa, OPS("v_cmp_ne_u32_sdwa %[s0], %[a0], %[b] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ne_u32_sdwa %[s1], %[a1], %[b] src0_sel:BYTE_1 src1_sel:DWORD",
                      "v_cmp_ne_u32_sdwa %[s2], %[a2], %[b] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ne_u32_sdwa %[s3], %[a3], %[b] src0_sel:BYTE_1 src1_sel:DWORD",
                      "v_cmp_ne_u32_sdwa %[s0], %[a4], %[b] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ne_u32_sdwa %[s1], %[a5], %[b] src0_sel:BYTE_1 src1_sel:DWORD",
                      "v_cmp_ne_u32_sdwa %[s2], %[a6], %[b] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ne_u32_sdwa %[s3], %[a7], %[b] src0_sel:BYTE_1 src1_sel:DWORD"))
KERNEL(k_pkmul, OPS("v_pk_mul_f32 %[p0], %[p0], %[q]", "v_pk_mul_f32 %[p1], %[p1], %[q]", "v_pk_mul_f32 %[p2], %[p2], %[q]", "v_pk_mul_f32 %[p3], %[p3], %[q]",
                    "v_pk_add_f32 %[p0], %[p0], %[q]", "v_pk_add_f32 %[p1], %[p1], %[q]", "v_pk_add_f32 %[p2], %[p2], %[q]", "v_pk_add_f32 %[p3], %[p3], %[q]"))
KERNEL(k_lshrv, EIGHT(F_LSHRV)) KERNEL(k_lshlv, EIGHT(F_LSHLV)) KERNEL(k_mad24, EIGHT(F_MAD24)) KERNEL(k_mul24, EIGHT(F_MUL24)) KERNEL(k_bfi, EIGHT(F_BFI))
KERNEL(k_andor, EIGHT(F_ANDOR)) KERNEL(k_add3, EIGHT(F_ADD3)) KERNEL(k_perm, EIGHT(F_PERM)) KERNEL(k_maxf, EIGHT(F_MAXF)) KERNEL(k_med3, EIGHT(F_MED3))
KERNEL(k_subu, EIGHT(F_SUBU)) KERNEL(k_or, EIGHT(F_OR))
KERNEL(k_h4f4, OPS(F_BFE("%[a0]"), F_ADD32("%[a1]"), F_BFE("%[a2]"), F_ADD32("%[a3]"), F_BFE("%[a4]"), F_ADD32("%[a5]"), F_BFE("%[a6]"), F_ADD32("%[a7]")))
KERNEL(k_h2f6, OPS(F_BFE("%[a0]"), F_ADD32("%[a1]"), F_XOR("%[a2]"), F_ADD32("%[a3]"), F_BFE("%[a4]"), F_ADD32("%[a5]"), F_XOR("%[a6]"), F_ADD32("%[a7]")))
KERNEL(k_h6f2, OPS(F_BFE("%[a0]"), F_BFE("%[a1]"), F_BFE("%[a2]"), F_ADD32("%[a3]"), F_BFE("%[a4]"), F_BFE("%[a5]"), F_BFE("%[a6]"), F_ADD32("%[a7]")))
KERNEL(k_c4f4, OPS("v_cmp_lt_f32_e64 %[s0], %[a0], %[b]", F_XOR("%[a1]"), "v_cmp_lt_f32_e64 %[s1], %[a2], %[b]", F_XOR("%[a3]"),
                   "v_cmp_lt_f32_e64 %[s2], %[a4], %[b]", F_XOR("%[a5]"), "v_cmp_lt_f32_e64 %[s3], %[a6], %[b]", F_XOR("%[a7]")))
KERNEL(k_p4f4, OPS("v_pk_mul_f32 %[p0], %[p0], %[q]", F_ADD32("%[a1]"), "v_pk_add_f32 %[p1], %[p1], %[q]", F_ADD32("%[a3]"),
                   "v_pk_mul_f32 %[p2], %[p2], %[q]", F_ADD32("%[a5]"), "v_pk_add_f32 %[p3], %[p3], %[q]", F_ADD32("%[a7]")))
KERNEL(k_hh4, OPS(F_BFE("%[a0]"), F_MIN("%[a1]"), F_BFE("%[a2]"), F_MIN("%[a3]"), F_BFE("%[a4]"), F_MIN("%[a5]"), F_BFE("%[a6]"), F_MIN("%[a7]")))
KERNEL(k_cmpcnd, OPS("v_cmp_lt_f32 vcc, %[a0], %[b]", "v_add_f32 %[a1], %[a1], %[b]", "v_add_f32 %[a2], %[a2], %[b]", "v_cndmask_b32_e64 %[a3], %[a3], %[b], vcc",
                     "v_cmp_lt_f32 vcc, %[a4], %[b]", "v_add_f32 %[a5], %[a5], %[b]", "v_add_f32 %[a6], %[a6], %[b]", "v_cndmask_b32_e64 %[a7], %[a7], %[b], vcc"))
KERNEL(k_execsub, OPS("s_and_b64 exec, %[s0], %[s1]", "v_sub_f32 %[a0], %[a0], %[b]", "s_and_b64 exec, %[s0], %[s2]", "v_sub_f32 %[a1], %[a1], %[b]",
                      "s_and_b64 exec, %[s0], %[s3]", "v_sub_f32 %[a2], %[a2], %[b]", "s_mov_b64 exec, -1", "v_add_f32 %[a3], %[a3], %[b]"))
KERNEL(k_cmpcnd32, OPS("v_cmp_lt_f32 vcc, %[a0], %[b]", "v_cndmask_b32 %[a1], %[a1], %[b], vcc", "v_cmp_lt_f32 vcc, %[a2], %[b]", "v_cndmask_b32 %[a3], %[a3], %[b], vcc",
                       "v_cmp_lt_f32 vcc, %[a4], %[b]", "v_cndmask_b32 %[a5], %[a5], %[b], vcc", "v_cmp_lt_f32 vcc, %[a6], %[b]", "v_cndmask_b32 %[a7], %[a7], %[b], vcc"))
KERNEL(k_cmpcnd32b, OPS("v_cmp_lt_f32 vcc, %[a0], %[b]", "v_cndmask_b32 %[a1], %[a1], %[b], vcc", "v_cndmask_b32 %[a2], %[a2], %[b], vcc", "v_cndmask_b32 %[a3], %[a3], %[b], vcc",
                        "v_cmp_lt_f32 vcc, %[a4], %[b]", "v_cndmask_b32 %[a5], %[a5], %[b], vcc", "v_cndmask_b32 %[a6], %[a6], %[b], vcc", "v_cndmask_b32 %[a7], %[a7], %[b], vcc"))
KERNEL(k_salu, OPS("s_and_b64 %[s0], %[s0], %[s1]", "s_andn2_b64 %[s1], %[s1], %[s2]", "s_or_b64 %[s2], %[s2], %[s3]", "s_and_b64 %[s3], %[s3], %[s0]",
                   "s_or_b64 %[s0], %[s0], %[s2]", "s_andn2_b64 %[s1], %[s1], %[s3]", "s_or_b64 %[s2], %[s2], %[s0]", "s_and_b64 %[s3], %[s3], %[s1]"))
KERNEL(k_mix, OPS("v_add_f32 %[a0], %[a0], %[b]", "s_and_b64 %[s0], %[s0], %[s1]", "v_add_f32 %[a1], %[a1], %[b]", "s_or_b64 %[s2], %[s2], %[s3]",
                  "v_add_f32 %[a2], %[a2], %[b]", "s_andn2_b64 %[s1], %[s1], %[s3]", "v_add_f32 %[a3], %[a3], %[b]", "s_or_b64 %[s3], %[s3], %[s0]"))
template <class K> static void run(const char *name, K k, float *out) {
  const int iters = 300, wpc = 24;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256 * wpc), dim3(64), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)wpc / 4.0 * iters * 256.0;
  printf("%-28s %.2f cycles per instruction per SIMD\n", name, ms * 1e-3 * 2.4e9 / instr_per_simd); fflush(stdout);
}
int main() {
  float *out; hipMalloc(&out, 256 * 32 * 64 * 4);
  run("v_add_f32 (VOP2)", k_add32, out); run("v_add_f32_e64 (VOP3, 2 src)", k_add64, out); run("v_xor_b32", k_xor, out); run("v_mov_b32", k_mov, out);
  run("v_add_u32", k_addu, out); run("v_mul_f32", k_mulf, out); run("v_min_f32", k_min, out); run("v_min3_f32", k_min3, out); run("v_fma_f32", k_fma, out);
  run("v_bfe_u32", k_bfe, out); run("v_lshl_or_b32", k_lshlor, out); run("v_lshl_add_u32", k_lshladd, out); run("v_bcnt_u32_b32", k_bcnt, out);
  run("v_ffbh_u32", k_ffbh, out); run("v_bitop3_b32", k_bitop3, out); run("v_bfm_b32", k_bfm, out);
  run("v_cndmask_b32 (vcc)", k_cnd32, out); run("v_cndmask_b32_e64 (sgpr)", k_cnd64, out); run("v_cndmask_e64 0, v, sgpr", k_cndc, out);
  run("v_cndmask_e64 v, v, vcc", k_cnd64v, out); run("cmp vcc; 2 adds; cndmask_e64 vcc", k_cmpcnd, out); run("3 x (s_and exec; v_sub) + mov exec + add", k_execsub, out);
  run("4 x (cmp vcc; cndmask VOP2 vcc)", k_cmpcnd32, out); run("2 x (cmp vcc; 3 cndmask VOP2 vcc)", k_cmpcnd32b, out);
  run("v_lshrrev_b32", k_lshr, out); run("v_lshlrev_b32_e64", k_lshle64, out); run("v_and_b32", k_and, out); run("v_or3_b32", k_or3, out); run("v_sub_f32", k_subf, out); run("v_min_u32", k_minu, out);
  run("v_lshrrev_b32 by vgpr", k_lshrv, out); run("v_lshlrev_b32 by vgpr", k_lshlv, out); run("v_mad_u32_u24", k_mad24, out); run("v_mul_u32_u24", k_mul24, out);
  run("v_bfi_b32", k_bfi, out); run("v_and_or_b32", k_andor, out); run("v_add3_u32", k_add3, out); run("v_perm_b32", k_perm, out); run("v_max_f32", k_maxf, out);
  run("v_med3_f32", k_med3, out); run("v_sub_u32", k_subu, out); run("v_or_b32", k_or, out);
  run("4 bfe + 4 add interleaved", k_h4f4, out); run("2 bfe + 6 add/xor", k_h2f6, out); run("6 bfe + 2 add", k_h6f2, out); run("4 cmp->sgpr + 4 xor", k_c4f4, out);
  run("4 pk + 4 add", k_p4f4, out); run("4 bfe + 4 min", k_hh4, out);
  run("v_cmp_lt_f32 vcc", k_cmpvcc, out); run("v_cmp_lt_f32_e64 sgpr", k_cmpsgpr, out); run("v_cmp_ne_u32_sdwa sgpr", k_cmpsdwa, out);
  run("v_pk_mul/add_f32", k_pkmul, out); run("s_and/or/andn2_b64", k_salu, out); run("v_add_f32 + s_and_b64 mix", k_mix, out);
  return 0;
}
