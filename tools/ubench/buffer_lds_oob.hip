// What does `buffer_load_dwordx4 ... lds` write for lanes whose address is out of the descriptor's range?  Every kernel that
// zero-fills padding through the loader (gemm.hip rows past M / N and padding taps, c3d.hip / c3ds.hip pad pixels and the rows
// above / below an image, gram.hip rows past a slice) relies on the answer being "sixteen zero bytes, for every such lane, also when
// the WHOLE wave is out of range and when num_records is 0".  Three launches over an LDS image pre-filled with 0xdeadbeef:
//   mixed : lanes 0-31 in range, 32-39 marker offset 0x80000000, 40-47 in range, 48-63 beyond num_records by a plain offset
//   allOOB: every lane carries the marker offset (c3d.hip: pieces 0..6 of an image's first tile -- the row above the image)
//   empty : num_records = 0, every lane a valid-looking offset (c3d.hip: `issue(tile + G, ..., valid = false)` past the last tile)
// Prints one PASS / FAIL line per case (exit code 1 on any FAIL).   hipcc --offload-arch=gfx950 -O2 buffer_lds_oob.hip -o buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const char* src, unsigned* out, int nrec, int mode) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
  int voff = lane * 16;
  if (mode == 0) {
    if (lane >= 32 && lane < 40) voff = 0x80000000;      // masked lanes
    if (lane >= 48) voff = lane * 16 + 4096;             // beyond num_records via plain offset
  } else if (mode == 1) {
    voff = 0x80000000;
  }
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, 64 /*soffset*/, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  char* d; unsigned* o; hipMalloc(&d, 1 << 20); hipMalloc(&o, 1024);
  static unsigned h[1 << 18]; for (int i = 0; i < (1 << 18); ++i) h[i] = i + 1;   // (no zero word anywhere in the source)
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  const char* names[3] = {"mixed", "allOOB", "empty"};
  int bad_total = 0;
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64, 4096>>>(d, o, mode == 2 ? 0 : 2048, mode);
    unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const bool in_range = mode == 0 && (l < 32 || (l >= 40 && l < 48));
      for (int w = 0; w < 4; ++w) {
        const unsigned want = in_range ? h[(64 + l * 16) / 4 + w] : 0u;
        if (r[l * 4 + w] != want) { if (!bad) printf("  %s lane %d word %d: %08x, expected %08x\n", names[mode], l, w, r[l * 4 + w], want); ++bad; }
      }
    }
    printf("%-6s %s (%d words differ)\n", names[mode], bad ? "FAIL" : "PASS: out-of-range lanes wrote zeros, in-range lanes their data", bad);
    bad_total += bad;
  }
  return bad_total ? 1 : 0;
}
