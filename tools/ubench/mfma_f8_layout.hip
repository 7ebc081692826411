// Which lane/byte holds which (row, k) element of the A / B operands of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3)?
// "Other dtypes: check the map with exact integer data before relying on it" (cdna_hip_programming.md §3).
// Hypotheses for lane l, byte j (0..31) of the 8-VGPR operand:
//   H1: k = 32*(l>>4) + j                      (one contiguous run of 32 per lane group)
//   H2: k = 64*(j>>4) + 16*(l>>4) + (j&15)     (two K=64 halves, 16 per lane group each)
// A[i][k] and B[k][n] are small integers (exact in e4m3); D = A.B is compared with the host product under each hypothesis.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int hyp) {   // A [16][128], B^T [16][128] (B[k][n] stored as Bt[n][k])
  const int l = threadIdx.x;
  uint8_t a[32], b[32];
  for (int j = 0; j < 32; ++j) {
    const int kk = hyp == 1 ? 32 * (l >> 4) + j : 64 * (j >> 4) + 16 * (l >> 4) + (j & 15);
    a[j] = A[(l & 15) * 128 + kk];
    b[j] = B[(l & 15) * 128 + kk];
  }
  v8i av, bv;
  __builtin_memcpy(&av, a, 32);
  __builtin_memcpy(&bv, b, 32);
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // C/D: col = lane & 15, row = (lane >> 4) * 4 + r
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

static uint8_t e4m3_of_int(int v) {   // exact for 0..16
  if (v == 0) return 0;
  int e = 0; while ((1 << (e + 1)) <= v) ++e;          // v = 1.m * 2^e
  const int m = ((v << 3) >> e) & 7;
  return (uint8_t)(((e + 7) << 3) | m);
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128]; int iA[16 * 128], iB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9; iB[i] = rand() % 5; hA[i] = e4m3_of_int(iA[i]); hB[i] = e4m3_of_int(iB[i]); }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int hyp = 1; hyp <= 2; ++hyp) {
    k<<<1, 64>>>(dA, dB, dD, hyp);
    float hD[256]; hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) {
      long s = 0; for (int kk = 0; kk < 128; ++kk) s += (long)iA[i * 128 + kk] * iB[n * 128 + kk];
      bad += hD[i * 16 + n] != (float)s;       // D[row = A row][col = B col]
      badT += hD[n * 16 + i] != (float)s;
    }
    printf("hypothesis %d: %d / 256 wrong (D[Arow][Bcol]), %d / 256 wrong (transposed)   sample D[1][2] = %.0f\n", hyp, bad, badT, hD[18]);
  }
  return 0;
}
