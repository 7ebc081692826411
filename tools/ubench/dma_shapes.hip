// Micro-benchmark: LDS-DMA (global_load_lds_dwordx4) throughput per CU for different shapes of one wave-instruction's
// 1 KiB piece, with and without MFMAs running beside it.  256 persistent workgroups x 8 waves; per step every wave issues
// 4 pieces (32 KiB per workgroup and step) into a 4-slot LDS ring, waits for the pieces of two steps ago and meets
// the others at a barrier -- the K-loop skeleton of the GEMM kernels.
//   shape 0: 16 rows x  64 B   (K-step of 32 bf16: the v3 GEMM's pieces)
//   shape 1:  8 rows x 128 B   (K-step of 64 bf16)
//   shape 2:  4 rows x 256 B
// Source: a [rows][ld] bf16 matrix; each workgroup sweeps its own 512 rows (A-like, 256 rows) + 256 rows shared by all
// (W-like); `ld` bytes per row; the K offset advances every step and wraps inside the row, so everything stays in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SHAPE, int MFMA, int READS, int SALU = 0>
__global__ __launch_bounds__(512, 1) void k(const char* src, long ld, int steps, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RB = SHAPE == 0 ? 64 : (SHAPE == 1 ? 128 : 256);     // bytes per row piece
  constexpr int RPI = 1024 / RB;                                     // rows per instruction
  constexpr int LPR = RB / 16;                                       // lanes per row
  const int prow = lane / LPR, pch = lane % LPR;
  // per step: 32 KiB = 32 pieces; wave w issues pieces w, w+8, w+16, w+24; pieces 0-15: A rows, 16-31: W rows
  const char* base[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = wave + q * 8;
    const bool isw = piece >= 16;
    const long row = (isw ? 0 : 256 + (long)blockIdx.x * 256) + (long)(piece & 15) * RPI * (256 / (16 * RPI)) * 0 + (long)(piece & 15) * RPI + prow;
    base[q] = src + row * ld + pch * 16;
  }
  f32x4_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  bf16x8_t fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(float)(lane + e); fb[e] = (__bf16)(float)(wave - e); }
  auto issue = [&](int st) {
    const long koff = ((long)st * RB) % ld;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base[q] + koff),
                                       (__attribute__((address_space(3))) void*)(smem + (st & 3) * 32768 + (wave + q * 8) * 1024), 16, 0, 0);
  };
  issue(0); issue(1); issue(2);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (READS == 3) {
    // ping-pong: waves 4-7 run one barrier behind waves 0-3; L = reads + DMA issue + wait for the reads, M = 32 MFMAs
    const int grp = wave >> 2;
    wait_vm<8>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int s = 0; s < steps; ++s) {
      const char* slot = smem + (s & 3) * 32768;
      const int off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 8) >> 2)) << 4);
      bf16x8_t a[8], b[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(slot + (wave >> 2) * 8192 + i * 1024 + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(slot + 16384 + (wave & 3) * 4096 + j * 1024 + off);
      issue(s + 3);
      if (grp == 1) wait_vm<8>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * j], a[i], acc[i], 0, 0, 0);
          acc[8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * j + 1], a[i], acc[8 + i], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) wait_vm<8>();
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else
  for (int s = 0; s < steps; ++s) {
    wait_vm<8>();
    __builtin_amdgcn_s_barrier();
    if (SALU) {   // a dependent chain of scalar bookkeeping in front of the step's issue, as a generic persistent loop has
      unsigned x = (unsigned)s;
#pragma unroll
      for (int i = 0; i < SALU; ++i) asm volatile("s_add_u32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xffff" : "+s"(x));
      if (x == 0x12345678u) sink[1] = 1.f;
    }
    issue(s + 3);
    if (READS) {
      // the real kernels' fragment traffic: 8 + 4 ds_read_b128 per wave and step (64-byte rows, conflict-free swizzle), consumed by the MFMAs
      const char* slot = smem + (s & 3) * 32768;
      const int off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 8) >> 2)) << 4);
      bf16x8_t a[8], b[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(slot + (wave >> 2) * 8192 + i * 1024 + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(slot + 16384 + (wave & 3) * 4096 + j * 1024 + off);
      if (READS == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * j], a[i], acc[i], 0, 0, 0);
          acc[8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2 * j + 1], a[i], acc[8 + i], 0, 0, 0);
        }
    } else if (MFMA) {
#pragma unroll
      for (int r = 0; r < MFMA; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    }
  }
  wait_vm<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 12345.f) sink[0] = s + smem[lane];
}

template <int SHAPE, int MFMA, int READS = 0, int SALU = 0>
void run(const char* src, long ld, const char* name) {
  unsigned long long* cyc; float* sink;
  hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, MFMA, READS, SALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int steps = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, MFMA, READS, SALU><<<256, 512, 131072>>>(src, ld, steps, cyc, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, MFMA, READS, SALU><<<256, 512, 131072>>>(src, ld, steps, cyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  printf("%-16s ld %5ld B  salu %2d reads %d mfma/step/wave %2d : %7.0f cycles/step  %6.1f GB/s/CU  %5.2f TB/s chip  (%.2f GHz)\n", name, ld, 2 * SALU, READS, MFMA * 16,
         avg / steps, 32768.0 * steps / (ms * 1e-3) / 1e9, 256 * 32768.0 * steps / (ms * 1e-3) / 1e12, avg / (ms * 1e-3) / 1e9);
  hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
  char* src; const size_t bytes = (size_t)(256 + 256 * 256) * 2048;
  hipMalloc(&src, bytes); hipMemset(src, 0x3c, bytes);
  if (argc > 1) {   // random bf16 values in [-2, 2): operand toggling as in a real GEMM (the clock and any throttling follow the data)
    unsigned short* h = (unsigned short*)malloc(bytes);
    unsigned x = 12345u;
    for (size_t i = 0; i < bytes / 2; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x807f) | (0x3f00 + ((x >> 8) & 0x80))); }
    hipMemcpy(src, h, bytes, hipMemcpyHostToDevice); free(h);
    printf("random operands\n");
  }
  for (long ld : {512L}) {
    run<0, 0>(src, ld, "16 rows x  64 B"); run<1, 0>(src, ld, " 8 rows x 128 B"); run<2, 0>(src, ld, " 4 rows x 256 B");
    run<0, 2>(src, ld, "16 rows x  64 B"); run<1, 2>(src, ld, " 8 rows x 128 B"); run<2, 2>(src, ld, " 4 rows x 256 B");
    run<0, 2, 1>(src, ld, "16 rows x  64 B"); run<1, 2, 1>(src, ld, " 8 rows x 128 B");
    run<0, 2, 2>(src, ld, "16 rows x  64 B"); run<1, 2, 2>(src, ld, " 8 rows x 128 B");
    run<0, 2, 3>(src, ld, "16x64 ping-pong"); run<1, 2, 3>(src, ld, "8x128 ping-pong");
    run<0, 2, 1, 15>(src, ld, "16 rows x  64 B"); run<0, 2, 1, 30>(src, ld, "16 rows x  64 B"); run<0, 2, 1, 60>(src, ld, "16 rows x  64 B");
  }
  return 0;
}
