// microbenchmark: LDS atomic throughput on gfx950 (cycles per wave-instruction per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters) {
  __shared__ uint32_t lds[16384];
  __shared__ unsigned long long lds64[4096];
  for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = 0;
  for (int i = threadIdx.x; i < 4096; i += 1024) lds64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *p = lds + wave * 1024 + lane;            // distinct banks within a wave, distinct regions per wave
  unsigned long long *p64 = lds64 + wave * 256 + lane;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) atomicAdd(p + u * 64, 1u);                                   // all lanes
      if (MODE == 1) { if ((lane & 3) == 0) atomicAdd(p + u * 64, 1u); }          // 16 of 64 lanes
      if (MODE == 2) { if (lane < 16) atomicAdd(p + u * 64, 1u); }                // first 16 lanes
      if (MODE == 3) atomicAdd(p64 + u * 16, 1ull);                               // 64-bit
      if (MODE == 4) p[u * 64] = it;                                              // plain store
      if (MODE == 5) acc += atomicAdd(p + u * 64, 1u);                            // returning atomic
      if (MODE == 6) atomicAdd(lds + wave * 1024 + (lane * 4 + u) % 1024, 1u);    // stride-4 dwords: 4-way bank conflict
      if (MODE == 7) { uint32_t v = p[u * 64]; p[u * 64] = v + 1; }              // non-atomic read-modify-write
      if (MODE == 8) { if (lane < 32) atomicAdd(p + u * 64, 1u); }                // half wave
      if (MODE == 9) atomicAdd(p + u * 64, 0u);                                   // add zero
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + acc + (uint32_t)lds64[0];
}
template <int MODE> void run(const char *name, uint32_t *d) {
  const int iters = 20000, nblk = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(1024), 0, 0, d, 100);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(1024), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = (double)iters * 8 * 16;   // wave-instructions per CU (16 waves, 1 block per CU)
  printf("%-44s %8.3f ms  %6.2f ns per wave-instr per CU  (~%5.1f cycles @2.4GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}
int main() {
  uint32_t *d; hipMalloc(&d, 4096);
  run<0>("ds_add_u32 all 64 lanes", d);
  run<1>("ds_add_u32 16 lanes (every 4th)", d);
  run<2>("ds_add_u32 16 lanes (first 16)", d);
  run<8>("ds_add_u32 32 lanes (first half)", d);
  run<9>("ds_add_u32 all lanes, value 0", d);
  run<3>("ds_add_u64 all 64 lanes", d);
  run<4>("ds_write_b32 all lanes", d);
  run<5>("ds_add_rtn_u32 all lanes", d);
  run<6>("ds_add_u32 4-way bank conflict", d);
  run<7>("ds_read+ds_write (non-atomic RMW)", d);
  return 0;
}
