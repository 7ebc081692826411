// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet layer1 conv2; call site reference model.py:35) as
// a DIRECT convolution for gfx950.
//
// As an implicit GEMM (gemm.hip, 256x64 tiles) this layer pulls every input pixel through LDS-DMA nine times -- 22 GB of L2 -> LDS
// traffic per launch at batch 6144 for a 2.5 GB input -- and ran at 2.98 ms (477 TFLOP/s) against a 0.9 ms HBM floor.  Here:
//   * a workgroup's tile is EIGHT FULL IMAGE ROWS (448 consecutive output pixels of a 56-wide image = 28 fragments of 16 pixels):
//     its input patch is ten consecutive image rows, one contiguous 70 KiB run of the NHWC tensor, staged ONCE by LDS-DMA into a
//     zero-padded LDS image (58 pixels x 128 B per row; the pad pixels and the rows above / below the image are out-of-range
//     buffer loads, i.e. zeros), double buffered across tiles; the nine taps of a fragment are nine shifted reads of that image;
//   * the weights (64 x 576 bf16 = 72 MFMA fragments) never leave the register file: FOUR waves per workgroup, one per SIMD with
//     the whole 512-register budget, each holding ALL of W -- 36 fragments in AccVGPRs and 36 in ArchVGPRs (the MFMA reads its
//     first operand from either file) next to the 7 x 4 accumulator fragments of its 112 pixels -- so a K-step is 7 ds_read_b128
//     and 28 MFMAs, with no barrier and no weight traffic inside a tile;
//   * LDS bank conflicts: 16-byte chunk c of patch pixel q is stored at chunk position c ^ (q & 7) (applied on the DMA's source
//     address and on the fragment read): 4.6 LDS cycles per ds_read_b128 over all fragments and taps (4 = conflict free; enumerated
//     with the instruction's real lane groups);
//   * epilogue: BatchNorm partial sums (train mode) reduced per tile into a per-wave LDS row, or bias + ReLU (eval mode, folded
//     BatchNorm); bf16 through a per-wave staging strip; a fragment's 16 pixels are one contiguous 2 KiB run of the output.
// Same interface as the generic path (sr_conv2d); the partial-statistics row count comes from sr_conv_stats_rows.
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct C3Args {
  const bf16_t* x;          // [B, H, W, 64]
  const bf16_t* w;          // [64][9][64]   (K index = tap * 64 + channel)
  bf16_t* y;                // [B, H, W, 64]
  const float* bias;        // [64] or null
  float* stats;             // [grid * 4][2][64] or null
  int B, H, relu, no_store, tiles_h;
  const float* in_scale; const float* in_shift;   // [64] or null: the convolution runs on relu(x*in_scale + in_shift) (IN kernels)
};

constexpr int C3_W = 56, C3_TH = 8, C3_PW = C3_W + 2, C3_PR = C3_TH + 2;
constexpr int C3_TM = C3_TH * C3_W;                       // 448 output pixels per tile
constexpr int C3_FPW = C3_TM / 16 / 4;                    // 7 fragments per wave
constexpr int C3_PBYTES = C3_PR * C3_PW * 128;            // 74 240
constexpr int C3_NP = (C3_PBYTES + 1023) / 1024;          // 73 LDS-DMA pieces
constexpr int C3_NPW = (C3_NP + 3) / 4;                   // 19 per wave
constexpr int C3_PBUF = C3_NP * 1024;                     // 74 752 (pieces past the patch are not issued)
constexpr int C3_STG = 2 * C3_PBUF, C3_STAT = C3_STG + 4 * 2048, C3_BIAS = C3_STAT + 4 * 512, C3_INAFF = C3_BIAS + 256, C3_LDS = C3_INAFF + 512;
constexpr int C3_OOB = (int)0x80000000;
static_assert(C3_TM % 64 == 0 && C3_LDS <= 160 * 1024, "tile / LDS budget");

template <int N> __device__ __forceinline__ void c3wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc (AccVGPRs) += W fragment (AccVGPRs or ArchVGPRs) x activation fragment (ArchVGPRs)
__device__ __forceinline__ void c3mma_a(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(a));
}
__device__ __forceinline__ void c3mma_a0(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {   // first K-step: C = 0
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "a"(w), "v"(a));
}
__device__ __forceinline__ void c3mma_v(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}

__device__ __forceinline__ float c3row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// AFF: bias (+ ReLU) in the epilogue (eval mode: folded BatchNorm).  ST: BatchNorm partial statistics (train mode).
// IN: the input is the RAW output of the preceding convolution; its BatchNorm + ReLU (in_scale, in_shift) is applied to the patch in LDS.
template <bool AFF, bool ST, bool IN = false>
__device__ __forceinline__ void c3_body(const C3Args& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 patch buffers | 4 staging strips | 4 statistics rows | bias
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const long ntiles = (long)p.B * p.tiles_h;
  const int G = gridDim.x;

  // ---- weights -> registers: fragment (j, ks): lane holds W[j*16 + (lane & 15)][ks*32 + 8*(lane >> 4) .. +7]; ks = tap*2 + half
  // K-steps 0 .. NWA-1 in AccVGPRs (next to the 112 accumulator registers: 16 of the 256 stay free for the allocator), the rest in ArchVGPRs
  constexpr int NWA = 8, NWV = 18 - NWA;
  bf16x8_t wa[4][NWA], wv[4][NWV];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int ks = 0; ks < NWA; ++ks) wa[j][ks] = *reinterpret_cast<const bf16x8_t*>(p.w + (j * 16 + frow) * 576 + ks * 32 + fgrp * 8);
#pragma unroll
    for (int ks = 0; ks < NWV; ++ks) wv[j][ks] = *reinterpret_cast<const bf16x8_t*>(p.w + (j * 16 + frow) * 576 + (NWA + ks) * 32 + fgrp * 8);
  }
  float* const lstat = reinterpret_cast<float*>(smem + C3_STAT) + wave * 128;      // this wave's [2][64] running sums
  float* const lbias = reinterpret_cast<float*>(smem + C3_BIAS);
  lstat[lane] = 0.f; lstat[64 + lane] = 0.f;
  if (threadIdx.x < 64) lbias[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
  float* const inaff = reinterpret_cast<float*>(smem + C3_INAFF);
  if (IN && threadIdx.x < 64) {              // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
    const int ch = (threadIdx.x & ~7) | sr_pair_order(threadIdx.x & 7);
    inaff[threadIdx.x] = p.in_scale[ch]; inaff[64 + threadIdx.x] = p.in_shift[ch];
  }

  // ---- patch loader.  Piece q = i*4 + wave lands at LDS bytes q*1024 + lane*16 of the buffer: patch pixel pq = q*8 + lane/8, chunk
  // position lane%8, which holds data chunk (lane%8) ^ (pq & 7) of that pixel.  The per-lane source offsets are relative to
  // image row y0 - 1 (the tile's buffer descriptor starts there and ends with the image, so the row below the last image row is
  // out of range by itself); pad pixels and pieces past the patch carry the out-of-range marker; the row above the image (first
  // tile of an image: patch row 0 = pieces 0..7) is masked per tile.  19 loop-invariant values per lane.
  int vrel[C3_NPW];
#pragma unroll
  for (int i = 0; i < C3_NPW; ++i) {
    const int pq = (i * 4 + wave) * 8 + (lane >> 3);
    const int pr = pq / C3_PW, pc = pq - pr * C3_PW;
    // (bits 0..19: the offset; bits 20..23: the patch row, for the IN kernels' own validity test)
    vrel[i] = (pr < C3_PR && pc >= 1 && pc <= C3_W) ? (((pr * C3_W + pc - 1) * 128 + (((lane & 7) ^ (pq & 7)) << 4)) | (pr << 20)) : C3_OOB;
  }
  auto issue = [&](long tile, int buf, bool valid) {
    const unsigned ut = (unsigned)tile;
    const long b = ut / (unsigned)p.tiles_h;
    const int y0 = (int)(ut - (unsigned)b * (unsigned)p.tiles_h) * C3_TH;
    const long left = (long)(p.H - y0 + 1) * C3_W * 128;                      // bytes from row y0 - 1 to the end of the image
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((uintptr_t)p.x + ((b * p.H + y0 - 1) * (long)C3_W) * 128), 0, valid ? (int)left : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < C3_NPW; ++i) {
      if (i * 4 + wave >= C3_NP) continue;          // (wave-uniform; the waits below count back from the youngest operations only)
      int vo = vrel[i] < 0 ? C3_OOB : (vrel[i] & 0xfffff);
      if (i < 2) vo = (y0 == 0 && (i * 4 + wave) * 8 + (lane >> 3) < C3_PW) ? C3_OOB : vo;   // (pieces 0..7 hold patch row 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * C3_PBUF + (i * 4 + wave) * 1024), 16, vo, 0,
                                               0, 0);
    }
  };

  // ---- fragment geometry.  A wave's 7 fragments are two image rows (112 pixels): fragment i = tile pixels wave*112 + 16 i + frow.
  // With qA = patch pixel of fragment 0 (tap 0,0) the fragments 0..2 sit at qA + 16 i, and -- two pad pixels later -- the
  // fragments 4..6 at qB + 16 i with qB = qA + 2; fragment 3 straddles the rows (lanes < 8: qA + 48, lanes >= 8: qB + 48).  16 pixels
  // = 2 KiB and 16 = 0 mod 8 (the swizzle period), so per K-step TWO addresses are computed and the seven reads are immediate
  // offsets i * 2048 from one of them.
  const int qA = wave * 2 * C3_PW + frow;
  char* const stg = smem + C3_STG + wave * 2048;
  const int srow = lane >> 3, sq = lane & 7;

  long tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, true);
  c3wait_vm<0>();
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += G) {
    // my pieces of this tile's patch have landed: everything but the 14 stores of the previous tile's epilogue
    c3wait_vm<2 * C3_FPW>();
    if constexpr (IN) {
      // BatchNorm + ReLU of the layer in front, applied to the 16-byte chunks THIS lane loaded (always the same 8 channels: chunk
      // (lane & 7) ^ (lane >> 3 & 7)), in place, between the lane's own wait for them and the tile's barrier.  Pad pixels and rows
      // outside the image were zero-filled by the loader and must stay zero: the convolution pads the NORMALISED tensor.
      const unsigned ut0 = (unsigned)tile;
      const int y00 = (int)(ut0 - (ut0 / (unsigned)p.tiles_h) * (unsigned)p.tiles_h) * C3_TH;
      const int c8 = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
      const sr_f32x4 s0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), s1_ = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
      const sr_f32x4 h0 = *reinterpret_cast<const sr_f32x4*>(inaff + 64 + c8), h1 = *reinterpret_cast<const sr_f32x4*>(inaff + 64 + c8 + 4);
#pragma unroll
      for (int i = 0; i < C3_NPW; ++i) {
        if (i * 4 + wave >= C3_NP) continue;
        const int row = y00 - 1 + ((vrel[i] >> 20) & 15);
        if (vrel[i] < 0 || row < 0 || row >= p.H) continue;
        char* const at = smem + buf * C3_PBUF + (i * 4 + wave) * 1024 + lane * 16;
        const sr_u32x4 nv = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(at), s0, s1_, h0, h1);
        // (written through inline asm: in front of an LDS store the compiler sees, it drains ALL vector-memory operations -- LDS-DMA
        //  may alias -- which here would be the previous tile's 14 output stores, a full HBM write latency per tile)
        asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)at), "v"(nv) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                     // ... everybody's have, and everybody has finished reading the other buffer
    asm volatile("" ::: "memory");
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    issue(tile + G, buf ^ 1, tile + G < ntiles);
    const char* pb = smem + buf * C3_PBUF;
    buf ^= 1;

    f32x4_t acc[C3_FPW][4];                          // (written, not accumulated into, by the MFMAs of the first K-step)
    // K-step ks = (tap, channel half): 7 fragment reads, 28 MFMAs, software pipelined by hand: fragment i of step ks+1 is read
    // into the registers of fragment i of step ks right behind the four MFMAs that consume them -- six MFMA groups (~400 cycles)
    // ahead of its own use.  Scheduling barriers keep that order (left alone, the compiler hoists all 126 reads to the top, or
    // sinks each step's reads below its MFMAs and exposes their latency).
    bf16x8_t a[C3_FPW];
    int adA = 0, adB = 0, adM = 0;
    auto addr_step = [&](int ks) {                  // byte addresses of the step's row-part bases inside the patch buffer
      const int tap = ks >> 1;
      if ((ks & 1) == 0) {
        const int q = qA + z + (tap / 3) * C3_PW + (tap % 3);
        adA = (q << 7) | ((fgrp ^ (q & 7)) << 4);
        adB = ((q + 2) << 7) | ((fgrp ^ ((q + 2) & 7)) << 4);
        adM = frow < 8 ? adA : adB;
      } else {                                        // the upper 32 channels: chunk ^ 4
        adA ^= 64; adB ^= 64; adM ^= 64;
      }
    };
    auto read_frag = [&](int i) {
      a[i] = *reinterpret_cast<const bf16x8_t*>(pb + (i < 3 ? adA : (i == 3 ? adM : adB)) + i * 2048);
    };
    addr_step(0);
#pragma unroll
    for (int i = 0; i < C3_FPW; ++i) read_frag(i);
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      if (ks + 1 < 18) { __builtin_amdgcn_sched_barrier(0); addr_step(ks + 1); }
#pragma unroll
      for (int i = 0; i < C3_FPW; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ks == 0) c3mma_a0(acc[i][j], wa[j][0], a[i]);
          else if (ks < NWA) c3mma_a(acc[i][j], wa[j][ks < NWA ? ks : 0], a[i]);
          else c3mma_v(acc[i][j], wv[j][ks < NWA ? 0 : ks - NWA], a[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < 18) read_frag(i);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (H is a multiple of 8: every pixel of the tile exists)
    const unsigned ut = (unsigned)tile;
    const long b = ut / (unsigned)p.tiles_h;
    const int y0 = (int)(ut - (unsigned)b * (unsigned)p.tiles_h) * C3_TH;
    // BatchNorm partial sums.  32 running sums per lane do not fit beside the weights (160 ArchVGPRs): the channel fragments j = 0, 1
    // are summed in the store pass, j = 2, 3 in a second pass over the accumulators, and each pair is folded per tile (16-lane
    // DPP sums, then LDS float adds by four lanes) into the wave's LDS row.  Packed f32 math: two channels per instruction.
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t s1[2][2], s2[2][2];
    auto flush = [&](int j0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a_ = c3row16_sum(s1[j][r >> 1][r & 1]), c_ = c3row16_sum(s2[j][r >> 1][r & 1]);
          if (frow == 0) {
            // (inline asm, like the strip stores below: hipcc guards a visible LDS atomic with a wait for the next tile's patch)
            const unsigned at = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lstat + (j0 + j) * 16 + fgrp * 4 + r);
            asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:256" ::"v"(at), "v"(a_), "v"(c_) : "memory");
          }
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) s1[j][h] = s2[j][h] = f32x2_t{0.f, 0.f};
    };
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) s1[j][h] = s2[j][h] = f32x2_t{0.f, 0.f};
    const bf16_t* const ybase = p.y + ((b * p.H + y0) * (long)C3_W) * 64;
#pragma unroll
    for (int i = 0; i < C3_FPW; ++i) {
      __builtin_amdgcn_sched_barrier(0);             // one fragment's 16 accumulator registers leave the AccVGPRs at a time
      const int t0 = (wave * C3_FPW + i) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2_t v01 = {acc[i][j][0], acc[i][j][1]}, v23 = {acc[i][j][2], acc[i][j][3]};
        if constexpr (AFF) {
          const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(lbias + j * 16 + fgrp * 4);
          v01 += f32x2_t{bv[0], bv[1]}; v23 += f32x2_t{bv[2], bv[3]};
        }
        if constexpr (ST) {
          if (j < 2) {
            s1[j][0] += v01; s1[j][1] += v23;
            s2[j][0] = __builtin_elementwise_fma(v01, v01, s2[j][0]);
            s2[j][1] = __builtin_elementwise_fma(v23, v23, s2[j][1]);
          }
        }
        float v[4] = {v01[0], v01[1], v23[0], v23[1]};
        if constexpr (AFF) {
          if (p.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm for the same reason as the normalisation's store above: the visible form got `s_waitcnt vmcnt(0)` in front of
        //  it, i.e. every epilogue waited for the next tile's patch; the strip is wave-private and a wave's LDS operations are in order)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(stg + frow * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const sr_u32x2*>(pk))
                     : "memory");
      }
      // 16 pixels x 128 B = one contiguous 2 KiB run of the NHWC output: two 16-byte stores per lane (the descriptor's range is
      // empty for statistics-only launches: every store is ISSUED, so that the wait at the top of the loop can count them)
      const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(ybase + (long)t0 * 64), 0, p.no_store ? 0 : 2048, 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int px = h * 8 + srow;
        const u32x4_t val = *reinterpret_cast<const u32x4_t*>(stg + px * 128 + ((sq ^ (px & 7)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(val, srd_o, px * 128 + sq * 16, 0, 0);
      }
    }
    if constexpr (ST) {
      flush(0);
#pragma unroll
      for (int i = 0; i < C3_FPW; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 2; j < 4; ++j) {
          asm volatile("" : "+a"(acc[i][j]));      // (opaque: otherwise the store pass's values are kept live for this pass -- 56 registers)
          f32x2_t v01 = {acc[i][j][0], acc[i][j][1]}, v23 = {acc[i][j][2], acc[i][j][3]};
          if constexpr (AFF) {
            const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(lbias + j * 16 + fgrp * 4);
            v01 += f32x2_t{bv[0], bv[1]}; v23 += f32x2_t{bv[2], bv[3]};
          }
          s1[j - 2][0] += v01; s1[j - 2][1] += v23;
          s2[j - 2][0] = __builtin_elementwise_fma(v01, v01, s2[j - 2][0]);
          s2[j - 2][1] = __builtin_elementwise_fma(v23, v23, s2[j - 2][1]);
        }
      }
      flush(2);
    }
  }
  c3wait_vm<0>();
  if constexpr (ST) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* const row = p.stats + (long)(blockIdx.x * 4 + wave) * 128;
    row[lane] = lstat[lane];
    row[64 + lane] = lstat[64 + lane];
  }
}

template <bool AFF, bool ST, bool IN = false>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(const C3Args p) { c3_body<AFF, ST, IN>(p); }
template <bool AFF, bool ST, bool IN = false> struct C3Tag {};

template <bool AFF, bool ST, bool IN = false>
int c3_launch(const C3Args& s, unsigned grid, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<C3Tag<AFF, ST, IN>>(reinterpret_cast<const void*>(&conv3x3_c64_kernel<AFF, ST, IN>), C3_LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_c64_kernel<AFF, ST, IN>), dim3(grid), dim3(256), C3_LDS, st, s);
  return SR_OK;
}

inline bool c3_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_C3_DIRECT"); return e && e[0] == '1'; }();
  return !off;
}
inline bool c3_serves(const sr_conv_args* a) {
  return c3_enabled() && !a->stem && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Cin == 64 && a->Cout == 64 && a->W == C3_W &&
         !a->res && !a->escale && a->H > 0 && a->H % C3_TH == 0 && (long)a->H * C3_W * 128 < 0x7fffffffL;
}
inline unsigned c3_grid(long ntiles) {
  const long cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

}  // namespace

// Internal hand-over from sr_conv2d / sr_conv_stats_rows (gemm.hip): SR_ERR_UNSUPPORTED when the launch is not this layer shape
// (the caller then uses the generic kernel).
int srx_c3d_rows(const sr_conv_args* a) {
  if (!c3_serves(a)) return SR_ERR_UNSUPPORTED;
  return (int)c3_grid((long)a->B * ((a->H + C3_TH - 1) / C3_TH)) * 4;
}

// Does the direct kernel serve this launch WITH an input affine?  (the train-mode form: raw output + statistics, no bias / ReLU)
bool srx_c3d_in_affine_ok(const sr_conv_args* a) {
  return c3_serves(a) && a->act == SR_ACT_NONE && !a->bias && a->stats != nullptr;
}

int srx_c3d_conv(const sr_conv_args* a, void* stream) {
  if (!c3_serves(a) || (a->act != SR_ACT_NONE && a->act != SR_ACT_RELU)) return SR_ERR_UNSUPPORTED;
  if ((a->in_scale || a->in_shift) && (!a->in_scale || !a->in_shift || !srx_c3d_in_affine_ok(a))) return SR_ERR_UNSUPPORTED;
  C3Args s;
  s.x = (const bf16_t*)a->x; s.w = (const bf16_t*)a->w; s.y = (bf16_t*)a->y; s.bias = a->bias; s.stats = a->stats;
  s.B = a->B; s.H = a->H; s.relu = a->act == SR_ACT_RELU; s.no_store = a->no_store;
  s.tiles_h = (a->H + C3_TH - 1) / C3_TH;
  s.in_scale = a->in_scale; s.in_shift = a->in_shift;
  const long ntiles = (long)s.B * s.tiles_h;
  if (ntiles > 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  SR_ROUTE(SR_ROUTE_C3D);
  if (a->in_scale) {
    const int rc = c3_launch<false, true, true>(s, c3_grid(ntiles), (hipStream_t)stream);
    if (rc != SR_OK) return rc;
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  const bool aff = a->bias != nullptr || s.relu, st = a->stats != nullptr;
  const unsigned grid = c3_grid(ntiles);
  const int rc = aff ? (st ? c3_launch<true, true>(s, grid, (hipStream_t)stream) : c3_launch<true, false>(s, grid, (hipStream_t)stream))
                     : (st ? c3_launch<false, true>(s, grid, (hipStream_t)stream) : c3_launch<false, false>(s, grid, (hipStream_t)stream));
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}
