// probe: unaligned raw buffer loads on gfx950 -- do they return the bytes at any byte offset, and what comes back
// for an offset past num_records (whole load past it / straddling it)?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t *xm, int nrec, const int *offs, u4 *out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)xm, (short)0, nrec, 0x00020000);
  out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b128(r, offs[threadIdx.x], 0, 0);
}
int main() {
  const int N = 4096, NREC = 1024;
  std::vector<uint8_t> h(N);
  for (int i = 0; i < N; i++) h[i] = (uint8_t)(i * 7 + 3);
  std::vector<int> offs;
  for (int i = 0; i < 40; i++) offs.push_back(i);                  // all alignments
  for (int i = NREC - 24; i < NREC + 8; i++) offs.push_back(i);    // straddling and past num_records
  offs.push_back((int)0x80000000u); offs.push_back((int)0x80000040u); offs.push_back(-5); offs.push_back((int)0xFFFFFF00u);
  while (offs.size() < 128) offs.push_back(0);
  uint8_t *d; int *doff; u4 *dout;
  hipMalloc(&d, N); hipMalloc(&doff, 128 * 4); hipMalloc(&dout, 128 * 16);
  hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice); hipMemcpy(doff, offs.data(), 128 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, NREC, doff, dout);
  std::vector<uint8_t> o(128 * 16);
  hipMemcpy(o.data(), dout, 128 * 16, hipMemcpyDeviceToHost);
  int bad_in = 0;
  for (int t = 0; t < 76; t++) {
    const long long off = (unsigned)offs[t];
    char line[128]; int n = 0; bool all_ok = true;
    for (int b = 0; b < 16; b++) {
      const long long p = off + b;
      const uint8_t want = p < NREC ? h[p] : 0;
      const uint8_t got = o[t * 16 + b];
      line[n++] = got == want ? '.' : (got == 0 ? '0' : (p < N && got == h[p] ? 'm' : 'X'));
      if (got != want) all_ok = false;
    }
    line[n] = 0;
    if (off + 16 <= NREC && !all_ok) bad_in++;
    if (!all_ok || t >= 72) printf("off %lld: %s\n", off, line);
  }
  printf("in-range loads wrong: %d\n", bad_in);
  return 0;
}
