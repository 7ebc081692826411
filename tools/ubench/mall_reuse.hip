// Micro-benchmark: can a consumer kernel read what its producer just wrote out of the 256 MiB Infinity Cache?
//   test A (direction): the producer sweeps a 2.4 GB tensor forward (y = f(r): one read + one write stream); the consumer then
//          reads y forward (nothing of its start is still cached) or BACKWARD (starts where the producer ended).
//   test B (chunk pipeline): producer and consumer alternate over chunks of S bytes (producer: read r_chunk, write y_chunk;
//          consumer: read y_chunk), against the same launches with the consumer reading a far-away chunk.
// 1024 workgroups x 256 threads, contiguous 32 KiB per workgroup pass (the fast mapping of stream_rates.hip).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void produce(const uint4* __restrict__ r, uint4* __restrict__ y, long n16) {
  const long chunk = 256L * 8;
  for (long c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
    const long base = c * chunk + threadIdx.x;
    uint4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = r[base + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u].x += 1; y[base + u * 256] = a[u]; }
  }
}
__global__ __launch_bounds__(256) void consume(const uint4* __restrict__ y, long n16, int backward, unsigned* sink) {
  const long chunk = 256L * 8, nch = n16 / chunk;
  unsigned acc = 0;
  for (long c = blockIdx.x; c < nch; c += gridDim.x) {
    const long cc = backward ? nch - 1 - c : c;
    const long base = cc * chunk + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const uint4 v = y[base + u * 256]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) *sink = acc;
}

static float timed(void (*fn)(void*), void* ctx, int rep) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fn(ctx); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < rep; ++i) fn(ctx);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / rep;
}

struct Ctx { uint4 *r, *y; long n16; int backward; long chunk16; int far; unsigned* sink; };

int main() {
  const long n16 = 18L * 8388608;          // 2.4 GB
  Ctx c{};
  c.n16 = n16;
  if (hipMalloc(&c.r, n16 * 16) != hipSuccess || hipMalloc(&c.y, n16 * 16) != hipSuccess) return 1;
  hipMalloc(&c.sink, 4);
  hipMemset(c.r, 1, n16 * 16);
  // ---- test A
  for (int bw = 0; bw < 2; ++bw) {
    c.backward = bw;
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float tp = 0, tc = 0;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(e0);
      produce<<<1024, 256>>>(c.r, c.y, n16);
      hipEventRecord(e1);
      consume<<<1024, 256>>>(c.y, n16, bw, c.sink);
      hipEventRecord(e2); hipEventSynchronize(e2);
      float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
      if (it) { tp += a; tc += b; }
    }
    printf("A: consumer %s : producer %7.1f us (%.2f TB/s), consumer %7.1f us (%.2f TB/s)\n", bw ? "BACKWARD" : "forward ",
           tp / 3 * 1e3, 2.0 * n16 * 16 / (tp / 3) / 1e9, tc / 3 * 1e3, 1.0 * n16 * 16 / (tc / 3) / 1e9);
  }
  // ---- test B
  for (long mb : {16L, 32L, 48L, 64L, 96L, 128L, 192L, 256L}) {
    const long ch16 = mb * 1024 * 1024 / 16;
    const long nch = n16 / ch16;
    for (int far = 0; far < 2; ++far) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9;
      for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        for (long k = 0; k < nch; ++k) {
          produce<<<1024, 256>>>(c.r + k * ch16, c.y + k * ch16, ch16);
          const long kc = far ? (k + nch / 2) % nch : k;
          consume<<<1024, 256>>>(c.y + kc * ch16, ch16, 0, c.sink);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
      }
      printf("B: chunk %4ld MB %s : %8.1f us for 3 x 2.4 GB = %.2f TB/s algorithmic\n", mb, far ? "consumer reads a FAR chunk " : "consumer reads the NEW chunk",
             best * 1e3, 3.0 * nch * ch16 * 16 / best / 1e9);
    }
  }
  return 0;
}
