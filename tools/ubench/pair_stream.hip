// Micro-benchmark: the memory side of the fused pair kernel (csrc/pair.hip) -- identity read + Z write of [M, 1024] bf16, no arithmetic --
// as a function of the walk: 256 threads, one workgroup per CU, a wave owns 48 rows of a 192-row tile and walks the 1024 columns in
// chunks; per chunk and 16-row fragment it loads / stores PIECES.
//   mode 0: chunk = 64 columns, instruction = 16 rows x 64 B  (lane (r, g) <- row r, 16 bytes at column 8 g: the MFMA accumulator layout), two per fragment
//   mode 1: chunk = 128 columns, four such instructions per fragment
//   mode 2: chunk = 256 columns, eight per fragment
//   mode 3: chunk = 64 columns, but the four waves take four DIFFERENT chunks of the same 48 rows at a time (512 contiguous bytes per row per visit)
//   mode 4: chunk = 64 columns, instruction = 8 rows x 128 B (lane <- row l / 8, 16 bytes at column 8 (l % 8)): what a transposed-through-LDS store would do
// DEPTH = chunks of identity in flight.  build: hipcc --offload-arch=gfx950 -O3 pair_stream.hip -o pair_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ constexpr int chunk_cols() { return MODE == 1 ? 128 : (MODE == 2 ? 256 : 64); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ res, uint4* __restrict__ out, long M) {
  constexpr int CC = chunk_cols<MODE>(), NCH = 1024 / CC, PPF = CC / 32;       // pieces (instructions) per fragment and chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long ntiles = M / (MODE == 3 ? 48 : 192);
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long row0 = MODE == 3 ? t * 48 : t * 192 + wave * 48;
    auto at = [&](int c, int f, int pc) -> long {       // uint4 index of piece pc of fragment f of chunk c
      if (MODE == 4) return (row0 + 16 * f + 8 * pc + (lane >> 3)) * 128 + c * 8 + (lane & 7);
      const int cc = MODE == 3 ? (c * 4 + wave) : c;
      return (row0 + 16 * f + r) * 128 + cc * (CC / 8) + pc * 4 + g;
    };
    constexpr int NC = MODE == 3 ? NCH / 4 : NCH;
    uint4 v[DEPTH][3][PPF];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int pc = 0; pc < PPF; ++pc) v[d][f][pc] = res[at(d, f, pc)];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int pc = 0; pc < PPF; ++pc) {
          uint4 o = v[c % DEPTH][f][pc];
          o.x = (o.x & 0x7fff7fffu) + 1; o.y += 2; o.z ^= 5; o.w += o.x;
          out[at(c, f, pc)] = o;
          if (c + DEPTH < NC) v[c % DEPTH][f][pc] = res[at(c + DEPTH, f, pc)];
        }
    }
  }
}

template <int MODE, int DEPTH>
void run(const uint4* res, uint4* out, long M) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, res, out, M);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(256), 0, 0, res, out, M);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("mode %d  chunks in flight %d : %8.1f us  %5.2f TB/s\n", MODE, DEPTH, ms * 1e3, 2.0 * M * 1024 * 2 / ms / 1e9);
}

int main() {
  const long M = 6144L * 196;
  uint4 *res, *out;
  hipMalloc(&res, M * 2048); hipMalloc(&out, M * 2048);
  hipMemset(res, 1, M * 2048);
  run<0, 1>(res, out, M); run<0, 2>(res, out, M); run<0, 4>(res, out, M);
  run<1, 1>(res, out, M); run<1, 2>(res, out, M);
  run<2, 1>(res, out, M); run<2, 2>(res, out, M);
  run<3, 2>(res, out, M); run<3, 4>(res, out, M);
  run<4, 2>(res, out, M); run<4, 4>(res, out, M);
  return 0;
}
