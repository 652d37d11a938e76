#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned __int128 u128;
__device__ __forceinline__ int popc128(u128 x) { return __popcll((unsigned long long)x) + __popcll((unsigned long long)(x >> 64)); }
__device__ __forceinline__ int ctz128(u128 x) { const unsigned long long lo = (unsigned long long)x; return lo ? __ffsll(lo) - 1 : 64 + __ffsll((unsigned long long)(x >> 64)) - 1; }
__global__ void k(const unsigned long long *in, unsigned long long *out, int sh) {
  const int i = threadIdx.x;
  u128 a = ((u128)in[2 * i + 1] << 64) | in[2 * i], m = ((u128)in[2 * i + 65] << 64) | in[2 * i + 64];
  u128 f = ((((a + m) ^ m) & m) | a);
  u128 low = a & (~a + 1);
  u128 run = ((a + low) ^ a) & a;
  u128 s = (a << sh) | (m >> (sh + 3));
  out[i] = (unsigned long long)f ^ (unsigned long long)(f >> 64) ^ popc128(run) ^ ctz128(low | 1) ^ (unsigned long long)(s >> 64) ^ (unsigned long long)s;
}
