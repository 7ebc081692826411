// Micro-benchmark: how fast can 256 persistent workgroups write a [M x N] bf16 matrix tile by tile (256 x 256 tiles,
// 8 waves, each wave owning 128 rows x 64 cols) with different shapes of the per-instruction store?
//   mode 0: wave-instruction = 8 rows x 128 B  (16 B per lane)     -- the GEMM epilogue's current shape
//   mode 1: wave-instruction = 4 rows x 256 B
//   mode 2: wave-instruction = 2 rows x 512 B
//   mode 3: wave-instruction = 1 row  x 1024 B
// (modes 1-3 re-assign which wave writes which bytes of the tile; bytes written are identical.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(uint4* out, long M, int N, int mode, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gn = N / 256; const long gm = M / 256; const long ntiles = gm * gn;
  const long rowstride = (long)N * 2 / 16;  // in uint4
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // fake K loop
    float acc = lane;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    const uint4 v = make_uint4(__float_as_uint(acc), lane, wave, (unsigned)t);
    const long m0 = (t / gn) * 256; const int n0 = (int)(t % gn) * 256;   // tile origin; tile row = 512 B = 32 uint4
    uint4* base = out + m0 * rowstride + n0 * 2 / 16;
    // the tile is 256 rows x 32 uint4; 8 waves x 64 lanes x 16 instr = 8192 uint4
    for (int it = 0; it < 16; ++it) {
      int row, c;
      if (mode == 0) { // wave w: rows (w>>2)*128 + it*8 + lane/8 ; cols (w&3)*8 + lane%8
        row = (wave >> 2) * 128 + it * 8 + (lane >> 3); c = (wave & 3) * 8 + (lane & 7);
      } else if (mode == 1) { // 4 rows x 16 uint4
        int blk = wave * 16 + it;            // 128 blocks of (4 rows x 16 uint4): 64 row-groups x 2 col halves
        row = (blk >> 1) * 4 + (lane >> 4); c = (blk & 1) * 16 + (lane & 15);
      } else if (mode == 2) { // 2 rows x 32 uint4
        int blk = wave * 16 + it;            // 128 blocks of 2 rows
        row = blk * 2 + (lane >> 5); c = lane & 31;
      } else { // 1 row x 64 uint4?? tile row has only 32 uint4 -> two half... use 2 rows x 32 in different order (same as mode 2 but rows interleaved across waves)
        int blk = it * 8 + wave;
        row = blk * 2 + (lane >> 5); c = lane & 31;
      }
      base[row * rowstride + c] = v;
    }
  }
}
int main() {
  const long M = 1204224; const int N = 1024;
  uint4* d; hipMalloc(&d, (size_t)M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int spin : {0, 20000}) for (int mode = 0; mode < 4; ++mode) {
    k<<<256, 512>>>(d, M, N, mode, spin); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) k<<<256, 512>>>(d, M, N, mode, spin); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("spin %5d mode %d: %8.1f us  %.2f TB/s\n", spin, mode, ms * 1e3, (double)M * N * 2 / ms / 1e9);
  }
  return 0;
}
