#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const char* src, unsigned* out, int nrec) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
  int voff = lane * 16;
  if (lane >= 32 && lane < 40) voff = 0x80000000;      // masked lanes
  if (lane >= 48) voff = lane * 16 + 4096;              // beyond num_records via plain offset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, 64 /*soffset*/, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  char* d; unsigned* o; hipMalloc(&d, 1 << 20); hipMalloc(&o, 1024);
  unsigned h[1 << 18]; for (int i = 0; i < (1 << 18); ++i) h[i] = i;
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(d, o, 2048);
  unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %08x %08x %08x %08x\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
  return 0;
}
