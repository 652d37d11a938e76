// lut16_xor_form (csrc/common.hpp) against a host model of v_perm_b32's selector rules (CDNA3 ISA guide, V_PERM_B32:
// selector 0-7 a byte of {S0, S1}; 8-11 the sign of byte 1 / 3 / 5 / 7 replicated; 12 0x00; >= 13 0xFF): for every table
// the XOR of the two lookups must be the table's entry for all 16 codes, with and without the lower-casing OR of 8.
// (That the hardware follows these rules is what the GPU parity tests pin: all 256 byte values, every context string.)
#include "common.hpp"
#include <stdio.h>
#include <random>

static uint8_t perm_byte(uint32_t s0, uint32_t s1, uint8_t sel) {
  uint8_t in[8];
  for (int i = 0; i < 4; i++) { in[i] = (uint8_t)(s1 >> (8 * i)); in[4 + i] = (uint8_t)(s0 >> (8 * i)); }
  if (sel >= 13) return 0xFF;
  if (sel == 12) return 0x00;
  if (sel >= 8) return (in[2 * (sel - 8) + 1] & 0x80) ? 0xFF : 0x00;
  return in[sel];
}
static uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) r |= (uint32_t)perm_byte(s0, s1, (uint8_t)(sel >> (8 * i))) << (8 * i);
  return r;
}
static uint32_t lookup(uint32_t w, const epi::ClassLut &x, uint32_t low8) {   // cx2_lut / mhlf_lut4
  const uint32_t sel = (w & 0x0F0F0F0Fu) | low8;
  return perm(x.lo1, x.lo0, sel) ^ perm(x.hi1, x.hi0, sel ^ 0x08080808u);
}

int main() {
  std::mt19937 rng(12345);
  for (int it = 0; it < 20000; it++) {
    epi::ClassLut d;
    d.lo0 = rng(); d.lo1 = rng(); d.hi0 = rng(); d.hi1 = rng();
    if (it % 3 == 0) { d.lo0 &= 0x7F7F7F7Fu; d.lo1 &= 0x7F7F7F7Fu; d.hi0 &= 0x7F7F7F7Fu; d.hi1 &= 0x7F7F7F7Fu; }   // the kernels' tables: entries below 128
    const epi::ClassLut x = epi::lut16_xor_form(d);
    const uint32_t tab[4] = {d.lo0, d.lo1, d.hi0, d.hi1};
    for (uint32_t c = 0; c < 16; c++) {
      const uint32_t want = (tab[c >> 2] >> (8 * (c & 3))) & 255u, up = c | 8u, want_up = (tab[up >> 2] >> (8 * (up & 3))) & 255u;
      const uint32_t w = c | ((rng() & 0xF0u)) | (((c + 5) & 15u) << 8) | 0xAB000000u;      // other bytes and high nibbles: any
      const uint32_t got = lookup(w, x, 0u) & 255u, got_up = lookup(w, x, 0x08080808u) & 255u;
      if (got != want || got_up != want_up) {
        printf("table %d code %u: got %02x want %02x (lower-cased: %02x / %02x)\n", it, c, got, want, got_up, want_up);
        return 1;
      }
    }
  }
  printf("lut16 ok\n");
  return 0;
}
