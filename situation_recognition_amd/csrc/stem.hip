// The 7x7 / stride 2 stem convolution (torchvision ResNet conv1; call site reference model.py:35) as a DIRECT convolution for gfx950.
//
// The generic implicit-GEMM kernel treats the stem as a GEMM with K = 8 filter rows x 32 (8 pixels x 4 channels of the padded
// NHWC4 image): every output pixel pulls its own 8 x 64-byte row slices through LDS-DMA -- 39 GB of L2 -> LDS traffic per
// launch at batch 6144 for a 2.6 GB input, and the kernel ran at 7.6 ms (190 TFLOP/s).  Neighbouring output pixels share
// almost all of their input, so here a workgroup stages the input PATCH of a 16 x 16 output tile once (37 rows x 38 pixels x 8 B
// = 11 KiB, double buffered, LDS-DMA) and every MFMA operand is read straight out of it:
//   * fragment of output row `ho`, filter row r: lane (m = lane & 15, kc = lane >> 4) needs pixels 2*wo+2*kc, +1 of input row
//     2*ho + r -- 16 contiguous, 16-byte-aligned bytes of the patch, at a 16-byte stride from lane to lane: conflict free, and no
//     im2col copy exists anywhere;
//   * the weights (64 x 7 x 32 bf16, the 8th pixel of each row is a zero tap) live in REGISTERS for the whole kernel
//     (4 x 7 fragments = 112 registers per lane): no weight traffic after the first microsecond;
//   * 8 waves, each 2 output rows x 64 channels = 2 x 4 accumulator fragments, 56 MFMAs (v_mfma_f32_16x16x32_bf16) per tile;
//   * epilogue: BatchNorm partial sums kept RUNNING per lane over all tiles of the workgroup (train mode) or bias + ReLU
//     (eval mode, folded BatchNorm), bf16 through a per-wave staging strip, 16-byte coalesced stores (an output row of 16 pixels
//     x 64 channels is 2 KiB contiguous in NHWC).
// Same interface as the generic path (sr_conv2d with stem != 0); the partial-statistics row count comes from sr_conv_stats_rows.
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

struct StemArgs {
  const bf16_t* x;          // [B, Hp, Wp, 4]
  const bf16_t* w;          // [64][8][32]
  bf16_t* y;                // [B, Ho, Wo, 64]
  const float* bias;        // [64] or null
  float* stats;             // [grid * 8][2][64] or null
  int B, Hp, Wp, Ho, Wo, relu, no_store;
  int tiles_h, tiles_w;
};

constexpr int PROW = 304;                 // patch row: 38 pixels x 8 B
constexpr int PBUF = 12288;               // one patch buffer (37 x 304 = 11 248 B = 11 LDS-DMA pieces; 12 are issued: waves 0-3 two, waves 4-7 one)
constexpr int NPB = 3;                    // patch buffers: the patch of tile t + 2 is in flight while tile t is computed (round 3)
constexpr int SOOB = (int)0x80000000;

template <int N> __device__ __forceinline__ void swait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float srow16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// NOSTORE: the statistics-only launch (train mode, first of the two launches): nothing is converted, staged or stored -- with K = 147 the
// epilogue's vector instructions outweigh the tile's 56 MFMAs (the kernel is bound by VALU + MFMA issue per SIMD, 16 waves per CU), so
// the launch that only needs the sums must not pay for the store path.
template <bool NOSTORE>
__device__ __forceinline__ void stem_body(const StemArgs& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 patch buffers, then 8 x 2 KiB staging strips
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;

  const long ntiles = (long)p.B * p.tiles_h * p.tiles_w;
  const int G = gridDim.x;

  // ---- weights -> registers: fragment (j, r): lane holds W[j*16 + (lane & 15)][r][8*(lane >> 4) .. +7]
  bf16x8_t wf[4][7];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 7; ++r) wf[j][r] = *reinterpret_cast<const bf16x8_t*>(p.w + (j * 16 + frow) * 256 + r * 32 + fgrp * 8);
  float* const vec = reinterpret_cast<float*>(smem + NPB * PBUF + 8 * 2048);        // bias in LDS (read per tile: 16 registers saved)
  if (threadIdx.x < 64) vec[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;

  // ---- patch loader: 16 pieces of 1 KiB (64 lanes x 16 B) cover the 703 16-byte chunks of a patch; wave w issues pieces w and w+8
  int pvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (wave + i * 8) * 64 + lane;                      // chunk index: row q / 19, 16-byte column q % 19
    const int row = q / 19, c = q - row * 19;
    pvo[i] = row < 37 ? (row * p.Wp * 8 + c * 16) : SOOB;
  }
  auto issue = [&](long tile, int buf, bool valid) {
    const unsigned ut = (unsigned)tile, t2 = ut / (unsigned)p.tiles_w;      // (32-bit: the host checks the tile count)
    const int tw = (int)(ut - t2 * (unsigned)p.tiles_w);
    const long b = t2 / (unsigned)p.tiles_h;
    const int th = (int)(t2 - (unsigned)b * (unsigned)p.tiles_h);
    const bf16_t* base = p.x + ((b * p.Hp + th * 32) * (long)p.Wp + tw * 32) * 4;
    // the range ends with the image batch: rows of a bottom-edge tile past the last image read as zeros
    const long left = ((long)p.B * p.Hp * p.Wp * 4 - (base - p.x)) * 2;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, valid ? (int)(left < 0x7ffff000L ? left : 0x7ffff000L) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i == 0 || wave < 4)          // (12 pieces: wave-uniform, the waits below count per wave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * PBUF + (wave + i * 8) * 1024), 16, pvo[i], 0, 0, 0);
  };

  f32x4_t acc[2][4];
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;

  char* const stg = smem + NPB * PBUF + wave * 2048;
  const int a_off = (lane & 15) * 16 + fgrp * 16;                  // + (2*lr + r) * PROW
  const int srow = lane >> 3, sq = lane & 7;                       // staging read: pixel lane/8 (+8), 16-byte chunk lane%8

  // Round 3: the patch of tile t + 2 is in flight while tile t is computed (three buffers; it was t + 1, and the compiler drained even
  // that mid-tile: in front of the epilogue's LDS stores hipcc waits vmcnt(0) -- an LDS-DMA may alias -- so every tile waited for a
  // full HBM latency: 2.2 us per 16 x 16 tile against 0.45 us of MFMAs).  The epilogue's LDS stores are inline asm now.
  long tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, true);
  if (tile + G < ntiles) issue(tile + G, 1, true); else issue(tile, 1, false);
  swait_vm<0>();
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += G) {
    // my pieces of this tile's patch (issued two tiles ago): everything but the 4 + 4 stores of the two epilogues since and the
    // pieces of the patch in between (2 for waves 0-3, 1 for waves 4-7)
    if (wave < 4) swait_vm<(NOSTORE ? 0 : 8) + 2>(); else swait_vm<(NOSTORE ? 0 : 8) + 1>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(tile + 2 * (long)G, buf == 0 ? 2 : buf - 1, tile + 2 * (long)G < ntiles);
    const char* pb = smem + buf * PBUF;
    buf = buf == NPB - 1 ? 0 : buf + 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(pb + (2 * (wave * 2 + i) + r) * PROW + a_off);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][r], fa, acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue
    const unsigned ut = (unsigned)tile, t2 = ut / (unsigned)p.tiles_w;      // (32-bit: the host checks the tile count)
    const int tw = (int)(ut - t2 * (unsigned)p.tiles_w);
    const long b = t2 / (unsigned)p.tiles_h;
    const int th = (int)(t2 - (unsigned)b * (unsigned)p.tiles_h);
    const int wo0 = tw * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ho = th * 16 + wave * 2 + i;
      const bool ok = ho < p.Ho && wo0 + frow < p.Wo;              // this lane's pixel (row ho, column wo0 + frow) exists
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(vec + j * 16 + fgrp * 4);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bv[r];
          const float m = ok ? v[r] : 0.f;
          s1[j][r] += m;
          s2[j][r] = __builtin_fmaf(m, m, s2[j][r]);
          if (p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr (NOSTORE) continue;
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm: see above; the strip is private to the wave and a wave's LDS operations execute in order)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + frow * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
      if constexpr (NOSTORE) continue;
      // 16 pixels x 128 B = one contiguous 2 KiB run of the NHWC output: two 16-byte stores per lane.  The descriptor's range
      // ends with the row's last valid pixel (and is empty for rows past Ho): every store is ISSUED.
      const long row0 = ((b * p.Ho + ho) * (long)p.Wo + wo0) * 64;
      const int npx = ho < p.Ho ? (p.Wo - wo0 < 16 ? p.Wo - wo0 : 16) : 0;
      const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (npx > 0 ? row0 : 0)), 0, npx * 128, 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int px = h * 8 + srow;
        const u32x4_t val = *reinterpret_cast<const u32x4_t*>(stg + px * 128 + ((sq ^ (px & 7)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(val, srd_o, px * 128 + sq * 16, 0, 0);
      }
    }
  }
  swait_vm<0>();
  if (p.stats) {
    float* const row = p.stats + (long)(blockIdx.x * 8 + wave) * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a[4], c[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = srow16_sum(s1[j][r]); c[r] = srow16_sum(s2[j][r]); }
      if (frow == 0) {
        *reinterpret_cast<float4*>(row + j * 16 + fgrp * 4) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(row + 64 + j * 16 + fgrp * 4) = make_float4(c[0], c[1], c[2], c[3]);
      }
    }
  }
}

// =====================================================================================================
// Stem + BatchNorm + ReLU + 3x3/2 max-pool in one launch (train mode, second launch of a two-launch scheme; eval mode with the
// folded affine): the 9.9 GB raw stem tensor of batch 6144 is never written nor read back.
//   launch 1: stem_conv_kernel with no_store (statistics only), sr_bn_finalize -> scale / shift;
//   launch 2: this kernel: convolution again, y = relu(conv * scale + shift) into an LDS tile, max over the 3x3 windows, store
//             the pooled pixels.
// A workgroup's 16 x 16 conv tile starts at conv row / column 14*t - 1, so that the 7 x 7 pooled pixels 7t .. 7t+6 find all of
// their 3x3 windows (conv rows 14t-1 .. 14t+13) inside the tile: tiles overlap by two conv rows (31 % more convolution work, on
// a kernel that is bound by its 2.5 GB of output).  Conv positions outside the image (row / column -1, or past Ho / Wo) enter
// the pool as 0: every window holds at least one real post-ReLU value >= 0, so this equals the -inf padding of max_pool2d.
// =====================================================================================================
struct StemPoolArgs {
  const bf16_t* x; const bf16_t* w; bf16_t* y;        // y: [B, Po, Qo, 64] pooled
  const float* scale; const float* shift;              // [64]
  int B, Hp, Wp, Ho, Wo, Po, Qo, tiles_h, tiles_w;
};

__device__ __forceinline__ void stem_pool_body(const StemPoolArgs& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 patch buffers, then the 16 x 16 x 64 bf16 conv tile (32 KiB)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const long ntiles = (long)p.B * p.tiles_h * p.tiles_w;
  const int G = gridDim.x;

  bf16x8_t wf[4][7];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 7; ++r) wf[j][r] = *reinterpret_cast<const bf16x8_t*>(p.w + (j * 16 + frow) * 256 + r * 32 + fgrp * 8);
  // scale / shift live in LDS (behind the conv tile) and are read per tile: 32 registers per lane more for fragment prefetch
  float* const vec = reinterpret_cast<float*>(smem + NPB * PBUF + 32768);
  if (threadIdx.x < 64) { vec[threadIdx.x] = p.scale[threadIdx.x]; vec[64 + threadIdx.x] = p.shift[threadIdx.x]; }

  // patch chunk q = piece * 64 + lane: row q / 19, 16-byte column q % 19 (2 pixels)
  int prow_[2], pcol_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (wave + i * 8) * 64 + lane;
    prow_[i] = q / 19; pcol_[i] = q - prow_[i] * 19;
  }
  auto issue = [&](long tile, int buf, bool valid) {
    const unsigned ut = (unsigned)tile, t2 = ut / (unsigned)p.tiles_w;      // (32-bit: the host checks the tile count)
    const int tw = (int)(ut - t2 * (unsigned)p.tiles_w);
    const long b = t2 / (unsigned)p.tiles_h;
    const int th = (int)(t2 - (unsigned)b * (unsigned)p.tiles_h);
    // patch origin in the padded image: row 2*(14 th - 1) = 28 th - 2, pixel 28 tw - 2 (negative for the first tile row / column:
    // those chunks are not fetched at all -- their conv positions are forced to 0 below)
    const int r0 = 28 * th - 2, c0 = 28 * tw - 2;
    const bf16_t* img = p.x + b * (long)p.Hp * p.Wp * 4;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, valid ? p.Hp * p.Wp * 8 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + prow_[i], px = c0 + 2 * pcol_[i];
      const bool ok = prow_[i] < 37 && row >= 0 && row < p.Hp && px >= 0 && px + 1 < p.Wp;
      const int vo = ok ? (row * p.Wp + px) * 8 : SOOB;
      if (i == 0 || wave < 4)          // (12 pieces: wave-uniform, the waits below count per wave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * PBUF + (wave + i * 8) * 1024), 16, vo, 0, 0, 0);
    }
  };

  char* const tilebuf = smem + NPB * PBUF;          // [16 conv rows][16 conv cols][64 ch] bf16: 128 B per pixel
  const int a_off = (lane & 15) * 16 + fgrp * 16;
  long tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, true);
  if (tile + G < ntiles) issue(tile + G, 1, true); else issue(tile, 1, false);
  swait_vm<0>();
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += G) {
    // my pieces of this tile's patch (issued two tiles ago): everything but the two (one-store) epilogues since and the pieces of the
    // patch in between (2 for waves 0-3, 1 for waves 4-7)
    if (wave < 4) swait_vm<4>(); else swait_vm<3>();
    __builtin_amdgcn_s_barrier();                    // ... and every wave has finished pooling the previous tile
    asm volatile("" ::: "memory");
    issue(tile + 2 * (long)G, buf == 0 ? 2 : buf - 1, tile + 2 * (long)G < ntiles);
    const char* pb = smem + buf * PBUF;
    buf = buf == NPB - 1 ? 0 : buf + 1;
    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(pb + (2 * (wave * 2 + i) + r) * PROW + a_off);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][r], fa, acc[i][j], 0, 0, 0);
      }
    }
    const unsigned ut = (unsigned)tile, t2 = ut / (unsigned)p.tiles_w;      // (32-bit: the host checks the tile count)
    const int tw = (int)(ut - t2 * (unsigned)p.tiles_w);
    const long b = t2 / (unsigned)p.tiles_h;
    const int th = (int)(t2 - (unsigned)b * (unsigned)p.tiles_h);
    // ---- BatchNorm + ReLU, conv tile -> LDS (pixel-major, 128 B per pixel; 8 bytes per (lane, j))
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = wave * 2 + i;                   // local conv row; lane's local conv column = frow
      const int ho = 14 * th - 1 + lr, wo = 14 * tw - 1 + frow;
      const bool ok = ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(vec + j * 16 + fgrp * 4), sh = *reinterpret_cast<const f32x4_t*>(vec + 64 + j * 16 + fgrp * 4);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ok ? fmaxf(__builtin_fmaf(acc[i][j][r], sc[r], sh[r]), 0.f) : 0.f;
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm: a visible LDS store would make hipcc drain the patches in flight; the lgkmcnt(0) + barrier below cover it)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(tilebuf + (lr * 16 + frow) * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- 3x3/2 max over the tile: work item = (pooled pixel 0..48, 16-byte channel chunk 0..7); 392 of the 512 lanes work
    const int item = threadIdx.x;
    const int pp = item >> 3, ch = item & 7;
    const int ph = pp / 7, pw = pp - ph * 7;
    u32x4_t best = {0, 0, 0, 0};
    const bool act = pp < 49;
    if (act) {
      // every candidate is a post-ReLU bf16 >= +0 (positions outside the image were written as +0): for such values the order of
      // the bf16 numbers is the order of their 16-bit patterns, so the 3x3 maximum is nine packed UNSIGNED 16-bit maxima per dword
      // (v_pk_max_u16: 36 instructions per lane against ~150 for unpack + v_max_f32 + repack)
      u16x8_t m = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int lr = 2 * ph + dh, lc = 2 * pw + dw;           // local conv position (tile origin = conv row 14 th - 1)
          const u16x8_t t = *reinterpret_cast<const u16x8_t*>(tilebuf + (lr * 16 + lc) * 128 + ((ch ^ (lc & 7)) << 4));
          m = __builtin_elementwise_max(m, t);
        }
      best = __builtin_bit_cast(u32x4_t, m);
    }
    // one store per lane and tile (issued by every lane: lanes without a pooled pixel, and pixels past the pooled image, are
    // dropped by the descriptor's range check)
    const int po = 7 * th + ph, qo = 7 * tw + pw;
    const bool in = act && po < p.Po && qo < p.Qo;
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + b * (long)p.Po * p.Qo * 64), 0, p.Po * p.Qo * 128, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(best, srd_o, in ? (po * p.Qo + qo) * 128 + ch * 16 : SOOB, 0, 0);
  }
  swait_vm<0>();
}

__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const StemPoolArgs p) { stem_pool_body(p); }
struct StemPoolTag {};

template <bool NOSTORE>
__global__ __launch_bounds__(512, 2) void stem_conv_kernel(const StemArgs p) { stem_body<NOSTORE>(p); }

template <bool NOSTORE> struct StemTag {};

inline bool stem_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_STEM_DIRECT"); return e && e[0] == '1'; }();
  return !off;
}

inline unsigned stem_grid(long ntiles) {
  const long cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

}  // namespace

// Internal hand-over from sr_conv2d / sr_conv_stats_rows (gemm.hip).  Returns SR_ERR_UNSUPPORTED when the launch is not a bf16 stem
// with 64 output channels (the caller then uses the generic kernel).
int srx_stem_rows(const sr_conv_args* a) {
  if (!stem_enabled() || !a->stem || a->Cout != 64) return SR_ERR_UNSUPPORTED;
  const int Ho = (a->H + 6 - 7) / 2 + 1, Wo = (a->W + 6 - 7) / 2 + 1;
  const long ntiles = (long)a->B * ((Ho + 15) / 16) * ((Wo + 15) / 16);
  return (int)stem_grid(ntiles) * 8;
}

extern "C" int sr_stem_bn_relu_maxpool(const void* xp, const void* w, const float* scale, const float* shift, void* y, int B, int H, int W,
                                      int dtype, void* stream) {
  if (!xp || !w || !scale || !shift || !y || B <= 0 || H <= 0 || W <= 0) return SR_ERR_ARG;
  if (dtype != SR_BF16) return SR_ERR_UNSUPPORTED;
  StemPoolArgs s;
  s.x = (const bf16_t*)xp; s.w = (const bf16_t*)w; s.y = (bf16_t*)y; s.scale = scale; s.shift = shift;
  s.B = B; s.Hp = (H + 6 + 1) & ~1; s.Wp = (W + 6 + 1) & ~1;
  s.Ho = (H + 6 - 7) / 2 + 1; s.Wo = (W + 6 - 7) / 2 + 1;
  s.Po = (s.Ho - 1) / 2 + 1; s.Qo = (s.Wo - 1) / 2 + 1;
  s.tiles_h = (s.Po + 6) / 7; s.tiles_w = (s.Qo + 6) / 7;
  if ((long)s.Hp * s.Wp * 8 >= 0x7fffffffL || (long)s.Po * s.Qo * 128 >= 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  const long ntiles = (long)B * s.tiles_h * s.tiles_w;
  if (ntiles > 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  constexpr int LDS = NPB * PBUF + 32768 + 512;
  if (!sr_set_dynamic_lds_tagged<StemPoolTag>(reinterpret_cast<const void*>(&stem_pool_kernel), LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL(stem_pool_kernel, dim3(stem_grid(ntiles)), dim3(512), LDS, (hipStream_t)stream, s);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

int srx_stem_conv(const sr_conv_args* a, void* stream) {
  if (!stem_enabled() || !a->stem || a->Cout != 64 || a->res || a->escale) return SR_ERR_UNSUPPORTED;
  const int Hp = (a->H + 6 + 1) & ~1, Wp = (a->W + 6 + 1) & ~1;
  const int Ho = (a->H + 6 - 7) / 2 + 1, Wo = (a->W + 6 - 7) / 2 + 1;
  if ((long)37 * Wp * 8 + 19 * 16 >= 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  StemArgs s;
  s.x = (const bf16_t*)a->x; s.w = (const bf16_t*)a->w; s.y = (bf16_t*)a->y; s.bias = a->bias; s.stats = a->stats;
  s.B = a->B; s.Hp = Hp; s.Wp = Wp; s.Ho = Ho; s.Wo = Wo; s.relu = a->act == SR_ACT_RELU; s.no_store = a->no_store;
  s.tiles_h = (Ho + 15) / 16; s.tiles_w = (Wo + 15) / 16;
  const long ntiles = (long)s.B * s.tiles_h * s.tiles_w;
  if (ntiles > 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  SR_ROUTE(SR_ROUTE_STEM);
  constexpr int LDS = NPB * PBUF + 8 * 2048 + 256;
  if (s.no_store) {
    if (!sr_set_dynamic_lds_tagged<StemTag<true>>(reinterpret_cast<const void*>(&stem_conv_kernel<true>), LDS)) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL(stem_conv_kernel<true>, dim3(stem_grid(ntiles)), dim3(512), LDS, (hipStream_t)stream, s);
  } else {
    if (!sr_set_dynamic_lds_tagged<StemTag<false>>(reinterpret_cast<const void*>(&stem_conv_kernel<false>), LDS)) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL(stem_conv_kernel<false>, dim3(stem_grid(ntiles)), dim3(512), LDS, (hipStream_t)stream, s);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}
