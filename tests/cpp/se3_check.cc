// Prints se3_exp / se3_log / se3_mul / se3_inverse / se3_matrix3x4 of badslam_amd/csrc/se3_device.h (the host compilation of
// the functions the pose kernels use) for tangents read from stdin, as hexadecimal binary32.  tests/test_cpu_se3.py compares
// them with the oracle's restatement of Sophus (oracle_core.c).
#include <cstdio>
#include <cstring>

#include "libvis_min.h"

static void Print(const float* v, int n) {
  for (int i = 0; i < n; ++i) { unsigned bits; memcpy(&bits, &v[i], 4); printf("%08x ", bits); }
}

int main() {
  float a[6], b[6];
  while (scanf("%f %f %f %f %f %f %f %f %f %f %f %f", &a[0], &a[1], &a[2], &a[3], &a[4], &a[5], &b[0], &b[1], &b[2], &b[3], &b[4], &b[5]) == 12) {
    float Ta[7], Tb[7], prod[7], inv[7], lg[6], m[12];
    bahip::se3_exp(a, Ta);
    bahip::se3_exp(b, Tb);
    bahip::se3_mul(Ta, Tb, prod);
    bahip::se3_inverse(Ta, inv);
    bahip::se3_log(prod, lg);
    bahip::se3_matrix3x4(Ta, m);
    Print(Ta, 7); Print(prod, 7); Print(inv, 7); Print(lg, 6); Print(m, 12);
    printf("\n");
  }
  return 0;
}
