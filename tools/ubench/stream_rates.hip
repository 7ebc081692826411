// Micro-benchmark: what HBM rate do the access patterns of the HBM-bound kernels reach, and which knobs move it?
//   op 0: y = relu(x*s + b) in place          (bn_apply: 1 read + 1 write stream)
//   op 1: y = relu(a + r)   out of place      (expansion-conv epilogue: 2 read + 1 write streams)
//   op 2: sum(x)                              (pure read)
//   op 3: y = c                               (pure write)
// knobs: workgroups per CU (grid = 256 * wpc, 256 threads), 16-byte accesses in flight per lane (U), nontemporal
// loads/stores, and the mapping (0: grid-stride per instruction, consecutive waves touch consecutive KiB;
// 1: each workgroup walks contiguous 64 KiB chunks).
// Buffers are 2.4 GB (the layer3 1024-channel activation at batch 6144), far beyond the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__device__ __forceinline__ void ld(uint4 (&v)[U], const uint4* p, long stride) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (NT) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + u * stride)); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
    else v[u] = p[u * stride];
  }
}
template <int U, bool NT>
__device__ __forceinline__ void st(const uint4 (&v)[U], uint4* p, long stride) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (NT) { u32x4 t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p + u * stride)); }
    else p[u * stride] = v[u];
  }
}
__device__ __forceinline__ uint4 f(uint4 a, uint4 b) {   // a little per-element work (bf16 unpack/relu-ish), like the real kernels
  uint4 r;
  r.x = (a.x & 0x7fff7fffu) + (b.x & 0x00010001u); r.y = (a.y & 0x7fff7fffu) + (b.y & 0x00010001u);
  r.z = (a.z & 0x7fff7fffu) + (b.z & 0x00010001u); r.w = (a.w & 0x7fff7fffu) + (b.w & 0x00010001u);
  return r;
}

template <int OP, int U, bool NT, int MAP>
__global__ __launch_bounds__(256) void k(uint4* __restrict__ x, const uint4* __restrict__ r, uint4* __restrict__ y, long n16, unsigned* sink) {
  // n16 = number of 16-byte elements (multiple of 256*U*gridDim)
  const long nthreads = (long)gridDim.x * 256;
  unsigned acc = 0;
  if (MAP == 0) {
    // iteration i covers [i*nthreads*U, (i+1)*nthreads*U): instruction u of a wave covers a contiguous 1 KiB
    for (long base = ((long)blockIdx.x * 256 + threadIdx.x); base < n16; base += nthreads * U) {
      uint4 a[U], b[U];
      if (OP != 3) ld<U, NT>(a, x + base, nthreads);
      if (OP == 1) ld<U, NT>(b, r + base, nthreads);
      if (OP == 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = OP == 3 ? make_uint4(1, 2, 3, 4) : f(a[u], OP == 1 ? b[u] : a[u]);
        st<U, NT>(a, (OP == 0 ? x : y) + base, nthreads);
      }
    }
  } else {
    // workgroup walks contiguous chunks of 256*U 16-byte elements (4 KiB * U)
    const long chunk = 256L * U;
    for (long c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
      const long base = c * chunk + threadIdx.x;
      uint4 a[U], b[U];
      if (OP != 3) ld<U, NT>(a, x + base, 256);
      if (OP == 1) ld<U, NT>(b, r + base, 256);
      if (OP == 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = OP == 3 ? make_uint4(1, 2, 3, 4) : f(a[u], OP == 1 ? b[u] : a[u]);
        st<U, NT>(a, (OP == 0 ? x : y) + base, 256);
      }
    }
  }
  if (OP == 2 && acc == 0x12345678u) *sink = acc;
}

template <int OP, int U, bool NT, int MAP>
void run(uint4* x, uint4* r, uint4* y, long n16, unsigned* sink, int wpc) {
  static const char* names[] = {"inplace 1R1W", "add 2R1W", "read", "write"};
  const double streams[] = {2, 3, 1, 1};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wpc;
  hipLaunchKernelGGL((k<OP, U, NT, MAP>), dim3(grid), dim3(256), 0, 0, x, r, y, n16, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<OP, U, NT, MAP>), dim3(grid), dim3(256), 0, 0, x, r, y, n16, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%-13s U=%d nt=%d map=%d wg/cu=%2d : %8.1f us  %6.2f TB/s\n", names[OP], U, (int)NT, MAP, wpc, ms * 1e3,
         streams[OP] * n16 * 16.0 / ms / 1e9);
  fflush(stdout);
}

template <int OP>
void sweep(uint4* x, uint4* r, uint4* y, long n16, unsigned* sink) {
  for (int wpc : {4, 8, 16}) {
    run<OP, 1, false, 0>(x, r, y, n16, sink, wpc);
    run<OP, 2, false, 0>(x, r, y, n16, sink, wpc);
    run<OP, 4, false, 0>(x, r, y, n16, sink, wpc);
    run<OP, 8, false, 0>(x, r, y, n16, sink, wpc);
    run<OP, 4, true, 0>(x, r, y, n16, sink, wpc);
    run<OP, 4, false, 1>(x, r, y, n16, sink, wpc);
    run<OP, 8, false, 1>(x, r, y, n16, sink, wpc);
    run<OP, 8, true, 1>(x, r, y, n16, sink, wpc);
  }
}

int main() {
  const long n16 = 18L * 8388608;              // 2.4 GB per buffer; a multiple of 256 threads * 8 accesses * 4096 workgroups
  const long bytes = n16 * 16;
  uint4 *x, *r, *y; unsigned* sink;
  if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&r, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) return 1;
  hipMalloc(&sink, 4);
  hipMemset(x, 1, bytes); hipMemset(r, 2, bytes); hipMemset(y, 3, bytes);
  sweep<0>(x, r, y, n16, sink);
  sweep<1>(x, r, y, n16, sink);
  sweep<2>(x, r, y, n16, sink);
  sweep<3>(x, r, y, n16, sink);
  return 0;
}
