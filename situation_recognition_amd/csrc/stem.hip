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
constexpr int PBUF = 16384;               // one patch buffer (37 x 304 = 11 248 B, rounded up to 16 LDS-DMA pieces)
constexpr int SOOB = (int)0x80000000;

template <int N> __device__ __forceinline__ void swait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float srow16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

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
  float bv[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = p.bias ? p.bias[j * 16 + fgrp * 4 + r] : 0.f;

  // ---- patch loader: 16 pieces of 1 KiB (64 lanes x 16 B) cover the 703 16-byte chunks of a patch; wave w issues pieces w and w+8
  int pvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (wave + i * 8) * 64 + lane;                      // chunk index: row q / 19, 16-byte column q % 19
    const int row = q / 19, c = q - row * 19;
    pvo[i] = row < 37 ? (row * p.Wp * 8 + c * 16) : SOOB;
  }
  auto issue = [&](long tile, int buf, bool valid) {
    const int tw = (int)(tile % p.tiles_w);
    const long t2 = tile / p.tiles_w;
    const int th = (int)(t2 % p.tiles_h);
    const long b = t2 / p.tiles_h;
    const bf16_t* base = p.x + ((b * p.Hp + th * 32) * (long)p.Wp + tw * 32) * 4;
    // the range ends with the image batch: rows of a bottom-edge tile past the last image read as zeros
    const long left = ((long)p.B * p.Hp * p.Wp * 4 - (base - p.x)) * 2;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, valid ? (int)(left < 0x7ffff000L ? left : 0x7ffff000L) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * PBUF + (wave + i * 8) * 1024), 16, pvo[i], 0, 0, 0);
  };

  f32x4_t acc[2][4];
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;

  char* const stg = smem + 2 * PBUF + wave * 2048;
  const int a_off = (lane & 15) * 16 + fgrp * 16;                  // + (2*lr + r) * PROW
  const int srow = lane >> 3, sq = lane & 7;                       // staging read: pixel lane/8 (+8), 16-byte chunk lane%8

  long tile = blockIdx.x;
  if (tile < ntiles) issue(tile, 0, true);
  swait_vm<0>();
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += G) {
    // my pieces of this tile's patch: everything but the 4 stores of the previous tile's epilogue
    swait_vm<4>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(tile + G, buf ^ 1, tile + G < ntiles);
    const char* pb = smem + buf * PBUF;
    buf ^= 1;
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
    const int tw = (int)(tile % p.tiles_w);
    const long t2 = tile / p.tiles_w;
    const int th = (int)(t2 % p.tiles_h);
    const long b = t2 / p.tiles_h;
    const int wo0 = tw * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ho = th * 16 + wave * 2 + i;
      const bool ok = ho < p.Ho && wo0 + frow < p.Wo;              // this lane's pixel (row ho, column wo0 + frow) exists
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bv[j][r];
          const float m = ok ? v[r] : 0.f;
          s1[j][r] += m;
          s2[j][r] = __builtin_fmaf(m, m, s2[j][r]);
          if (p.relu) v[r] = fmaxf(v[r], 0.f);
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<uint2*>(stg + frow * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8) = *reinterpret_cast<const uint2*>(pk);
      }
      // 16 pixels x 128 B = one contiguous 2 KiB run of the NHWC output: two 16-byte stores per lane.  The descriptor's range
      // ends with the row's last valid pixel (and is empty for rows past Ho / statistics-only launches): every store is ISSUED.
      const long row0 = ((b * p.Ho + ho) * (long)p.Wo + wo0) * 64;
      const int npx = (ho < p.Ho && !p.no_store) ? (p.Wo - wo0 < 16 ? p.Wo - wo0 : 16) : 0;
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

__global__ __launch_bounds__(512, 2) void stem_conv_kernel(const StemArgs p) { stem_body(p); }

struct StemTag {};

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
  constexpr int LDS = 2 * PBUF + 8 * 2048;
  if (!sr_set_dynamic_lds_tagged<StemTag>(reinterpret_cast<const void*>(&stem_conv_kernel), LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL(stem_conv_kernel, dim3(stem_grid(ntiles)), dim3(512), LDS, (hipStream_t)stream, s);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
