// Micro-benchmark: the memory side of the expansion-conv epilogue, out[M,N] = relu(acc + res[M,N]) in bf16 with N = 1024,
// as a function of the TILE SHAPE a workgroup owns and of how its waves walk the tile -- no MFMA, no K loop: only the
// residual read + output write (+ optionally the A-operand read), i.e. the floor the real kernel's epilogue can reach.
//   shape 0: 256 x 256 tile, wave = 128 rows x 64 cols, instruction = 8 rows x 128 B      (the v3 kernel today)
//   shape 1: 256 x 256 tile, wave = 32 rows x 256 cols, instruction = 2 rows x 512 B
//   shape 2: 128 x 512 tile, wave = 16 rows x 512 cols, instruction = 1 row  x 1 KiB
//   shape 3:  64 x 1024 tile, wave = 8 rows x 1024 cols, instruction = 1/2 row x 1 KiB (rows fully contiguous: 2 KiB)
//   shape 4:  64 x 1024 tile, wave = 64 rows x 128 cols, instruction = 4 rows x 256 B; all 8 waves on the same 4 rows
//             (what a 64 x 1024 MFMA tile with waves side by side along N would store)
// 512 threads, one workgroup per CU, XCD-contiguous tile order; D = strips of loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint4 addrelu(uint4 a, uint4 b) {
  uint4 r;
  r.x = (a.x & 0x7fff7fffu) + (b.x & 0x00010001u); r.y = (a.y & 0x7fff7fffu) + (b.y & 0x00010001u);
  r.z = (a.z & 0x7fff7fffu) + (b.z & 0x00010001u); r.w = (a.w & 0x7fff7fffu) + (b.w & 0x00010001u);
  return r;
}

// A wave's tile walk: returns the uint4 index (inside the [M][N/8] uint4 matrix) of instruction `it` (0..15) for this lane.
template <int SHAPE>
__device__ __forceinline__ long where(long tile, int wave, int lane, int it, int n16 /* uint4 per row */) {
  if (SHAPE == 0) {           // tiles of 256 x 32 uint4; 4 column tiles
    const long tm = tile >> 2; const int tn = (int)(tile & 3);
    const int row = (wave >> 2) * 128 + it * 8 + (lane >> 3), c = (wave & 3) * 8 + (lane & 7);
    return (tm * 256 + row) * n16 + tn * 32 + c;
  } else if (SHAPE == 1) {
    const long tm = tile >> 2; const int tn = (int)(tile & 3);
    const int row = wave * 32 + it * 2 + (lane >> 5), c = lane & 31;
    return (tm * 256 + row) * n16 + tn * 32 + c;
  } else if (SHAPE == 2) {    // tiles of 128 x 64 uint4; 2 column tiles
    const long tm = tile >> 1; const int tn = (int)(tile & 1);
    const int row = wave * 16 + it, c = lane;
    return (tm * 128 + row) * n16 + tn * 64 + c;
  } else if (SHAPE == 3) {    // tiles of 64 x 128 uint4
    const int row = wave * 8 + (it >> 1), c = (it & 1) * 64 + lane;
    return (tile * 64 + row) * n16 + c;
  } else {                    // SHAPE 4
    const int row = it * 4 + (lane >> 4), c = wave * 16 + (lane & 15);
    return (tile * 64 + row) * n16 + c;
  }
}

template <int SHAPE, int D, bool NT>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ res, uint4* __restrict__ out, long ntiles, int n16) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = gridDim.x;
  int vb = blockIdx.x;
  { const int xcd = vb & 7, q = G >> 3, r = G & 7; vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3); }
  for (long t = vb; t < ntiles; t += G) {
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += D) {
      uint4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = res[where<SHAPE>(t, wave, lane, i0 + d, n16)];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const uint4 o = addrelu(v[d], v[d]);
        uint4* p = out + where<SHAPE>(t, wave, lane, i0 + d, n16);
        if (NT) { u32x4 q = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(p)); }
        else *p = o;
      }
    }
  }
}

template <int SHAPE, int D, bool NT>
void run(const uint4* res, uint4* out, long M, int N) {
  const int n16 = N / 8;
  const long ntiles = M * (long)N / 65536;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, D, NT>), dim3(256), dim3(512), 0, 0, res, out, ntiles, n16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, D, NT>), dim3(256), dim3(512), 0, 0, res, out, ntiles, n16);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("shape %d  in-flight %2d  nt=%d : %8.1f us  %5.2f TB/s\n", SHAPE, D, (int)NT, ms * 1e3, 2.0 * M * N * 2 / ms / 1e9);
  fflush(stdout);
}

template <int SHAPE>
void sweep(const uint4* res, uint4* out, long M, int N) {
  run<SHAPE, 2, false>(res, out, M, N);
  run<SHAPE, 4, false>(res, out, M, N);
  run<SHAPE, 8, false>(res, out, M, N);
  run<SHAPE, 16, false>(res, out, M, N);
  run<SHAPE, 8, true>(res, out, M, N);
}

int main() {
  const long M = 1204224; const int N = 1024;      // layer3 at batch 6144: 2.47 GB per tensor
  uint4 *res, *out;
  if (hipMalloc(&res, (size_t)M * N * 2) != hipSuccess || hipMalloc(&out, (size_t)M * N * 2) != hipSuccess) return 1;
  hipMemset(res, 1, (size_t)M * N * 2);
  sweep<0>(res, out, M, N);
  sweep<1>(res, out, M, N);
  sweep<2>(res, out, M, N);
  sweep<3>(res, out, M, N);
  sweep<4>(res, out, M, N);
  return 0;
}
