#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include "old_renamed.h"
#include "new_numerics.h"
static uint32_t b(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
  long bad = 0, n = 0;
  uint32_t specials[] = {0x00000000u, 0x80000000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00001u, 0x7f800001u, 0x3f000000u, 0xbf000000u,
                         0x3f000001u, 0x3effffffu, 0x42b00000u, 0x42b00001u, 0xc2ae0000u, 0xc2ae0001u, 0x00000001u, 0x80000001u, 0x7f7fffffu, 0xff7fffffu};
  for (unsigned i = 0; i < sizeof(specials) / 4; ++i) {
    float x; memcpy(&x, &specials[i], 4);
    if (b(old_expf(x)) != b(uis_expf(x))) { ++bad; printf("exp %08x: %08x %08x\n", specials[i], b(old_expf(x)), b(uis_expf(x))); }
    if (b(old_tanhf(x)) != b(uis_tanhf(x))) { ++bad; printf("tanh %08x: %08x %08x\n", specials[i], b(old_tanhf(x)), b(uis_tanhf(x))); }
    if (b(old_sigmoidf(x)) != b(uis_sigmoidf(x))) { ++bad; printf("sig %08x\n", specials[i]); }
    ++n;
  }
  /* every 257th float bit pattern: 16.7 M values over the whole range incl. NaNs and denormals */
  for (uint64_t u = 0; u < 0x100000000ull; u += 257) {
    uint32_t v = (uint32_t)u; float x; memcpy(&x, &v, 4);
    if (b(old_expf(x)) != b(uis_expf(x)) || b(old_tanhf(x)) != b(uis_tanhf(x)) || b(old_sigmoidf(x)) != b(uis_sigmoidf(x))) {
      if (bad < 10) printf("diff at %08x: exp %08x/%08x tanh %08x/%08x\n", v, b(old_expf(x)), b(uis_expf(x)), b(old_tanhf(x)), b(uis_tanhf(x)));
      ++bad;
    }
    ++n;
  }
  srand(5);
  for (int i = 0; i < 4000000; ++i) {
    float a[7];
    for (int k = 0; k < 7; ++k) a[k] = ((float)rand() / RAND_MAX * 2.0f - 1.0f) * (k == 6 ? 1.0f : 4.0f);
    if (b(old_gru_unit(a[0], a[1], a[2], a[3], a[4], a[5], a[6])) != b(uis_gru_unit(a[0], a[1], a[2], a[3], a[4], a[5], a[6]))) ++bad;
    ++n;
  }
  printf("compared %ld values, %ld differing\n", n, bad);
  return bad != 0;
}
