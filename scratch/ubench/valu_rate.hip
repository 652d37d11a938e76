// microbenchmark: issue cost of the integer VALU instructions the tile kernels are made of, 8 waves per SIMD, independent
// chains (4 accumulators per lane): cycles per wave-instruction per SIMD = elapsed * clock * SIMDs / (instr * waves)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int ITER = 4096;
template <int OP>
__global__ __launch_bounds__(512, 8) void k(uint32_t *out, uint32_t s0, uint32_t s1) {
  uint32_t a = threadIdx.x * 2654435761u + s0, b = a ^ s1, c = a + 77u, d = b + 99u;
  uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d;
  unsigned long long sel = __ballot((threadIdx.x & 3) == (s0 & 3));
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      if (OP == 0) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 1) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 2) { asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 3) { asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a)); asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(b)); asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(c)); asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(d)); }
      if (OP == 4) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(b) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d) : "v"(a)); }
      if (OP == 19) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(c) : "v"(d), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(b) : "v"(c), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(d) : "v"(a), "s"(sel)); }
      if (OP == 20) { asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(a) : "v"(b), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, -1, %1, %2" : "=v"(c) : "v"(d), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(b) : "v"(c), "s"(sel)); asm volatile("v_cndmask_b32_e64 %0, -1, %1, %2" : "=v"(d) : "v"(a), "s"(sel)); }
      if (OP == 21) { asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 22) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_or_b32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 23) { asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(c) : "v"(d)); asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(b) : "v"(c)); asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(d) : "v"(a)); }
      if (OP == 24) { asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(d)); asm volatile("v_not_b32 %0, %1" : "=v"(b) : "v"(c)); asm volatile("v_not_b32 %0, %1" : "=v"(d) : "v"(a)); }
      if (OP == 25) { asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(sel) : "v"(a), "v"(b)); asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(sel) : "v"(c), "v"(d)); asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(sel) : "v"(b), "v"(c)); asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(sel) : "v"(d), "v"(a)); }
      if (OP == 26) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_max_u32 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 27) { asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d), "v"(a) : "vcc"); }
      if (OP == 28) { asm volatile("v_cmp_gt_u32_e64 %3, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(a) : "v"(b), "v"(c), "s"(sel)); asm volatile("v_cmp_gt_u32_e64 %3, %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(c) : "v"(d), "v"(a), "s"(sel)); }
      if (OP == 29) { asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc" : "+v"(a), "+v"(d) : "v"(b), "v"(c) : "vcc"); }
      if (OP == 6) { asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(a) : "v"(b)); asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(c) : "v"(d)); asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(b) : "v"(c)); asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(d) : "v"(a)); }
      if (OP == 7) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 8) { asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(r)); asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q)); asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(r)); }
      if (OP == 9) { asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a)); asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(b)); asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c)); asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d)); }
      if (OP == 10) { asm volatile("v_cmp_ne_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_ne_u32 vcc, %0, %1" :: "v"(c), "v"(d) : "vcc"); asm volatile("v_cmp_ne_u32 vcc, %0, %1" :: "v"(b), "v"(c) : "vcc"); asm volatile("v_cmp_ne_u32 vcc, %0, %1" :: "v"(d), "v"(a) : "vcc"); }
      if (OP == 11) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 12) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d)); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); }
      if (OP == 13) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c) : "v"(d)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(b) : "v"(c)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d) : "v"(a)); }
      if (OP == 14) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 15) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b) : "v"(c)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d) : "v"(a)); }
      if (OP == 16) { asm volatile("v_ffbl_b32 %0, %1" : "=v"(a) : "v"(b)); asm volatile("v_ffbl_b32 %0, %1" : "=v"(c) : "v"(d)); asm volatile("v_ffbl_b32 %0, %1" : "=v"(b) : "v"(c)); asm volatile("v_ffbl_b32 %0, %1" : "=v"(d) : "v"(a)); }
      if (OP == 17) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "s"(s0)); asm volatile("v_and_b32 %0, 0x55555555, %0" : "+v"(c)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(b) : "s"(s1)); asm volatile("v_and_b32 %0, 0x33333333, %0" : "+v"(d)); }
      if (OP == 18) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(r)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(q)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(r)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(q)); }
    }
  }
  if ((a ^ b ^ c ^ d ^ (uint32_t)q ^ (uint32_t)r ^ (uint32_t)sel) == 0x12345678u) out[0] = a;
}
template <int OP> void run(const char *name, uint32_t *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 workgroups of 512 threads per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 1u, 2u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 1u, 2u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)ITER * 16 * 8;   // 16 instructions per iteration, 8 waves per SIMD
  printf("%-22s %8.3f ms  %6.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / instr_per_simd);
}
int main() {
  uint32_t *d; hipMalloc(&d, 64);
  run<11>("v_add_f32", d); run<0>("v_and_b32", d); run<17>("v_and_b32 sgpr/literal", d); run<1>("v_perm_b32", d); run<2>("v_bitop3_b32", d); run<3>("v_lshrrev_b32", d);
  run<4>("v_add3_u32", d); run<5>("v_cndmask_b32", d); run<6>("v_sad_u8", d); run<7>("v_dot4_u32_u8", d); run<8>("v_lshlrev_b64", d);
  run<9>("v_add_u32_dpp", d); run<10>("v_cmp_ne_u32", d); run<12>("v_and_or_b32", d); run<13>("v_bcnt_u32_b32", d); run<14>("v_mul_u32_u24", d);
  run<15>("v_mul_lo_u32", d); run<16>("v_ffbl_b32", d); run<18>("v_lshl_add_u64", d); run<19>("v_cndmask_e64 sgpr", d); run<20>("v_cndmask_e64 const", d); run<21>("v_bfi_b32", d); run<22>("xor/or/add/sub", d); run<23>("v_lsh*_b32 variable", d); run<24>("v_mov/v_not", d); run<25>("v_cmp_e64 -> sgpr", d); run<26>("v_min/max_u32", d); run<27>("cmp+cndmask vcc (x2, /16)", d); run<28>("cmp+cndmask sgpr (x2, /16)", d); run<29>("cmp+2cndmask vcc (x1, /16)", d);
  return 0;
}
