// 3x3 / stride 1 / pad 1 convolution with 128 input and 128 output channels on 28 x 28 images (ResNet layer2 conv2; call site reference
// model.py:35) as a DIRECT convolution for gfx950.
//
// On the generic implicit-GEMM kernel (gemm.hip, 256x128 tiles) this layer ran at 1.6 ms per launch at batch 6144 (883 TFLOP/s): every
// K-step refills 16 KiB of activations AND 8 KiB of weights through LDS-DMA, each input pixel nine times, and its input had to be
// normalised by a separate sweep (bn1 -> relu: 0.46 ms of pure HBM traffic per block) because an in-LDS normalisation would run nine
// times per element.  Here (the 64-channel layer's design, c3d.hip, with the weights streamed instead of register resident):
//   * a workgroup's tile is FOUR FULL IMAGE ROWS (112 consecutive output pixels = 7 fragments of 16); its input patch -- six image
//     rows, 30 padded pixels each, 256 B per pixel = 45 KiB -- is staged ONCE by LDS-DMA (pad pixels and the rows outside the image are
//     out-of-range buffer loads: zeros), double buffered across tiles; the nine taps of a fragment are nine shifted reads of it;
//   * FOUR waves, one per SIMD, split the OUTPUT CHANNELS: wave w owns couts 32w .. 32w+31 (2 weight fragments) for all 7 pixel
//     fragments = 14 MFMAs (v_mfma_f32_16x16x32_bf16) per K-step against 7 + 2 fragment reads;
//   * the weights (128 x 1152 bf16 = 288 KiB, L2 resident) are streamed, but each wave only ever reads its OWN 32 rows: every wave
//     runs a PRIVATE 6-slot LDS ring (2 KiB per K-step), filled by its own LDS-DMA and retired by its own counted vmcnt -- the 36
//     K-steps of a tile contain no barrier at all (one per tile, for the patch);
//   * BatchNorm + ReLU of the layer in front (train mode: in_scale / in_shift) is applied to the NEXT tile's patch, one LDS-DMA piece
//     per K-step, in the shadow of the current tile's MFMAs: the normalised tensor is never written and the sweep is gone;
//   * LDS bank conflicts: a pixel is 256 B = all 64 banks, and a ds_read_b128 lane group is 8 pixels at k-chunk c plus 8 OTHER pixels at
//     chunk c + 1 (lanes 0-3, 12-15 | 20-27).  Patch chunk c of the pixel in patch row pr, column pc sits at chunk position
//     2 (((c >> 1) ^ key) & 7) + (c & 1) with key = (28 pr + pc) & 7: the parity of the chunk keeps the two halves of a lane group apart, and
//     the key -- the pixel's index at the IMAGE's row pitch, which is consecutive over the 16 output pixels of a fragment for every tap,
//     also where a fragment runs over the end of an image row (the patch's own pitch of 30 is not) -- spreads each half's 8 pixels over
//     the 8 chunk pairs: 4.0 LDS cycles per read for all 7 x 9 x 4 fragment reads (enumerated with the instruction's lane groups; a
//     plain c ^ (pixel & 15) costs 6.5).  Weight-row chunk c of row n sits at c ^ (-(n >> 2) & 3) (4.0 as well).  Both are applied
//     on the DMA's source address and on the fragment read;
//   * epilogue: running BatchNorm partial sums per lane over ALL tiles of the workgroup (one reduction per kernel), or bias + ReLU
//     (eval mode); bf16 through a per-wave staging strip, one 16-byte store per lane and pixel fragment.
// Same interface as the generic path (sr_conv2d); the partial-statistics row count comes from sr_conv_stats_rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct K8Args {
  const bf16_t* x;          // [B, 28, 28, 128]
  const bf16_t* w;          // [128][9][128]   (K index = tap * 128 + channel)
  bf16_t* y;                // [B, 28, 28, 128]
  const float* bias;        // [128] or null
  float* stats;             // [grid][2][128] or null
  int B, relu, no_store;
  const float* in_scale; const float* in_shift;   // [128] or null: the convolution runs on relu(x*in_scale + in_shift)
};

constexpr int K8_W = 28, K8_H = 28, K8_TH = 4, K8_PW = K8_W + 2, K8_PR = K8_TH + 2, K8_TILES_H = K8_H / K8_TH;
constexpr int K8_FP = K8_TH * K8_W / 16;                  // 7 pixel fragments per tile
constexpr int K8_NP = K8_PR * K8_PW * 256 / 1024;         // 45 LDS-DMA pieces per patch
constexpr int K8_NPW = 12;                                // ... 12 per wave (pieces 45..47 are out of range: zeros behind the patch)
constexpr int K8_PBUF = 4 * K8_NPW * 1024;                // 48 KiB per patch buffer
constexpr int K8_D = 6;                                   // depth of a wave's weight ring (divides the 36 K-steps: the slot of a step is static)
constexpr int K8_NK = 36;
constexpr int K8_WRING = 2 * K8_PBUF, K8_STG = K8_WRING + 4 * K8_D * 2048, K8_VEC = K8_STG + 4 * 2048, K8_LDS = K8_VEC + 512 + 1024;
constexpr int K8_OOB = (int)0x80000000;
static_assert(K8_NP <= 4 * K8_NPW && K8_NK % K8_D == 0 && K8_LDS <= 160 * 1024, "tile / LDS budget");

template <int N> __device__ __forceinline__ void k8wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float k8row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// vector-memory operations a wave has issued AFTER the weight pieces of K-step k (k in 0..35) at the moment it waits for them -- in step
// k - 1, right behind that step's own weight issue: the pieces of the D - 1 steps that follow, and -- for the first D - 1 steps of a
// tile, whose pieces were issued during the previous tile -- that tile's 7 output stores and this tile's 12 patch pieces (the very
// first tile of a workgroup has no stores in front of it).
template <int K, bool FIRST> constexpr int k8_younger() { return 2 * (K8_D - 1) + ((K >= 1 && K < K8_D) ? (FIRST ? K8_NPW : K8_NPW + K8_FP) : 0); }

// AFF: bias (+ ReLU) in the epilogue (eval mode: folded BatchNorm).  ST: BatchNorm partial statistics (train mode).
// IN: the input is the RAW output of the preceding convolution; its BatchNorm + ReLU (in_scale, in_shift) is applied to the patch in LDS.
template <bool AFF, bool ST, bool IN>
__device__ __forceinline__ void k8_body(const K8Args& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 patch buffers | 4 weight rings | 4 staging strips | bias | in-affine
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const long ntiles = (long)p.B * K8_TILES_H;
  const int G = gridDim.x;

  float* const lbias = reinterpret_cast<float*>(smem + K8_VEC);
  float* const inaff = reinterpret_cast<float*>(smem + K8_VEC + 512);
  if (threadIdx.x < 128) lbias[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
  if (IN && threadIdx.x < 128) {             // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
    const int ch = (threadIdx.x & ~7) | sr_pair_order(threadIdx.x & 7);
    inaff[threadIdx.x] = p.in_scale[ch]; inaff[128 + threadIdx.x] = p.in_shift[ch];
  }

  // ---- patch loader.  Piece q = i*4 + wave lands at LDS bytes q*1024 + lane*16 of the buffer: patch pixel pq = 4q + lane/16, chunk
  // position lane%16, which holds data chunk (lane%16) ^ (pq & 15) of that pixel.  Source offsets are relative to image row y0 - 1
  // (the tile's descriptor starts there and ends with the image, so rows below the image are out of range by themselves); pad pixels
  // and pieces past the patch carry the out-of-range marker; the row above the image (first tile: patch row 0) is masked per tile.
  int vrel[K8_NPW];
#pragma unroll
  for (int i = 0; i < K8_NPW; ++i) {
    const int q = i * 4 + wave, pq = q * 4 + (lane >> 4);
    const int pr = pq / K8_PW, pc = pq - pr * K8_PW;
    const int key = (pr * K8_W + pc) & 7, cp = lane & 15;
    const int cdat = ((((cp >> 1) ^ key) & 7) << 1) | (cp & 1);            // the data chunk that lives at chunk position cp of this pixel
    // (bits 0..19: the offset; bits 20..23: the patch row, for the row-above-the-image mask and the IN kernels' validity test;
    //  bits 24..27: the data chunk, for the IN kernels' scale / shift lookup)
    vrel[i] = (q < K8_NP && pc >= 1 && pc <= K8_W) ? (((pr * K8_W + pc - 1) * 256 + (cdat << 4)) | (pr << 20) | (cdat << 24)) : K8_OOB;
  }
  auto tile_y0 = [&](long tile) { const unsigned ut = (unsigned)tile; return (int)(ut - (ut / (unsigned)K8_TILES_H) * (unsigned)K8_TILES_H) * K8_TH; };
  auto issue_patch = [&](long tile, int buf, bool valid) {
    const long b = (unsigned)tile / (unsigned)K8_TILES_H;
    const int y0 = tile_y0(tile);
    const long left = (long)(K8_H - y0 + 1) * K8_W * 256;                     // bytes from row y0 - 1 to the end of the image
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((uintptr_t)p.x + ((b * K8_H + y0 - 1) * (long)K8_W) * 256), 0, valid ? (int)left : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < K8_NPW; ++i) {
      int vo = vrel[i] < 0 ? K8_OOB : (vrel[i] & 0xfffff);
      if (i < 2) vo = (y0 == 0 && ((vrel[i] >> 20) & 15) == 0) ? K8_OOB : vo;          // (patch row 0 = pixels 0..29: pieces 0..7)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * K8_PBUF + (i * 4 + wave) * 1024), 16, vo, 0, 0,
                                               0);
    }
  };
  // IN: BatchNorm + ReLU of the layer in front on the 16-byte chunks THIS lane loaded (their 8 channels: bits 24..27 of vrel; scale and
  // shift come from the LDS table).  Pad pixels and rows outside the image were zero-filled by the loader and must stay zero: the
  // convolution pads the NORMALISED tensor.
  auto normalise_piece = [&](int i, int buf, int y0) {
    const int row = y0 - 1 + ((vrel[i] >> 20) & 15);
    if (vrel[i] < 0 || row < 0 || row >= K8_H) return;
    const int c8 = ((vrel[i] >> 24) & 15) * 8;
    const sr_f32x4 ns0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), ns1 = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
    const sr_f32x4 nh0 = *reinterpret_cast<const sr_f32x4*>(inaff + 128 + c8), nh1 = *reinterpret_cast<const sr_f32x4*>(inaff + 128 + c8 + 4);
    char* const at = smem + buf * K8_PBUF + (i * 4 + wave) * 1024 + lane * 16;
    const sr_u32x4 nv = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(at), ns0, ns1, nh0, nh1);
    // (inline asm: in front of an LDS store it can see, hipcc drains every vector-memory operation -- LDS-DMA may alias)
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)at), "v"(nv) : "memory");
  };

  // ---- weight ring of this wave: K-step k (slot k % D) = rows 32 wave .. +31 of W, 64 bytes each at K offset 64 k: two pieces (one per
  // 16-row fragment); lane l of a piece -> row l/4, chunk position l%4, which holds data chunk (l%4) ^ (-(row >> 2) & 3)
  char* const wring = smem + K8_WRING + wave * (K8_D * 2048);
  int wvo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = lane >> 2, cd = (lane & 3) ^ ((0 - (n >> 2)) & 3);
    wvo[j] = ((32 * wave + 16 * j + n) * 1152 + cd * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 128 * 1152 * 2, 0x00020000);
  auto issue_w = [&](int k) {              // k = K-step 0..35 (compile time at every call site)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(wring + (k % K8_D) * 2048 + j * 1024), 16, wvo[j], k * 64, 0, 0);
  };
  const int boff = frow * 64 + ((fgrp ^ ((0 - (frow >> 2)) & 3)) << 4);
  auto read_b = [&](int k, bf16x8_t (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(wring + (k % K8_D) * 2048 + j * 1024 + boff);
  };

  // ---- fragment geometry.  Output pixel t = 16 i + frow of the tile sits in tile row t / 28; its patch pixel for tap (r, q) is
  // t + 2 (t / 28) + 30 r + q.  d = 2 (t / 28) takes the values 0, 2, 4, 6; fragments 1, 3 and 5 straddle two rows (per-lane select).
  // Per tap four base addresses are computed (one per d) and the seven reads are immediate offsets i * 4096 from them; the three
  // further 32-channel slices of a tap are XORs of the chunk bits.
  const bool hi1 = frow >= 12, hi3 = frow >= 8, hi5 = frow >= 4;
  char* const stg = smem + K8_STG + wave * 2048;       // two 1 KiB strips per wave (fragment i uses strip i & 1)

  // statistics / bias of this lane's 2 x 4 output channels (couts 32 wave + 16 j + 4 fgrp + r)
  float s1[2][4], s2[2][4], bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; bv[j][r] = 0.f; }

  long tile = blockIdx.x;
  if (tile < ntiles) issue_patch(tile, 0, true);
  k8wait_vm<0>();
  __syncthreads();                                    // bias / in-affine tables; my patch pieces have landed
  if (AFF) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = lbias[32 * wave + 16 * j + 4 * fgrp + r];
  }
  if (IN) {
    if (tile < ntiles) {
      const int y00 = tile_y0(tile);
#pragma unroll
      for (int i = 0; i < K8_NPW; ++i) normalise_piece(i, 0, y00);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int k = 0; k < K8_D; ++k) issue_w(k);
  bf16x8_t bb[2][2];                                  // weight fragments of the current / the next K-step (step ks uses bb[ks & 1]; 36 is even)
  k8wait_vm<2 * (K8_D - 1)>();                         // the pieces of step 0
  read_b(0, bb[0]);
  int buf = 0;

  // (Tried and removed: two accumulator sets, the previous tile's epilogue drained one pixel fragment per K-step inside the next
  //  tile's K loop.  With ONE wave per SIMD the loop is bound by instruction issue and LDS latency, not by the matrix pipe -- 14 MFMAs
  //  against 9 fragment reads, 2 DMA pieces and ~13 address instructions per step -- so every instruction moved into it lengthened it:
  //  1744 us against 1666 us for the epilogue between the K loops, same box.)
  const int px = lane >> 2, cq = lane & 3;
  bool first = true;
  for (; tile < ntiles; tile += G) {
    const bool fst = first;
    first = false;
    __builtin_amdgcn_s_barrier();                     // everybody's pieces of this patch are in place (and normalised); the other buffer is free
    asm volatile("" ::: "memory");
    const long tnext = tile + G;
    const int y0n = tile_y0(tnext < ntiles ? tnext : tile);
    issue_patch(tnext, buf ^ 1, tnext < ntiles);
    const char* pb = smem + buf * K8_PBUF;

    f32x4_t acc[K8_FP][2];
    bf16x8_t a[K8_FP];
    int z;                                            // an opaque 0, new per tile: without it the compiler computes the addresses of all 36 steps
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));        // once per kernel and keeps ~250 registers of them alive across the tile loop (spills)
    int ad[4] = {0, 0, 0, 0}, ad1 = 0, ad3 = 0, ad5 = 0;
    auto addr_step = [&](int ks) {
      const int tap = ks >> 2, s = ks & 3;
      if (s == 0) {
        const int off = (tap / 3) * K8_PW + (tap % 3);
        const int key = (frow + z + (tap / 3) * K8_W + (tap % 3)) & 7;    // (the same for every fragment: 16 i = 0 mod 8, and it ignores the patch pitch)
        const int pos = ((((fgrp >> 1) ^ key) & 7) << 5) | ((fgrp & 1) << 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) ad[d] = ((frow + z + 2 * d + off) << 8) | pos;
      } else {                                        // chunk index 4 s + fgrp: the chunk-pair bits change by 2, 6, 2 (x 32 bytes)
        const int x = (s == 2 ? 12 : 4) << 4;
#pragma unroll
        for (int d = 0; d < 4; ++d) ad[d] ^= x;
      }
      ad1 = hi1 ? ad[1] : ad[0]; ad3 = hi3 ? ad[2] : ad[1]; ad5 = hi5 ? ad[3] : ad[2];
    };
    auto read_frag = [&](int i) {
      const int base = i == 0 ? ad[0] : (i == 1 ? ad1 : (i == 2 ? ad[1] : (i == 3 ? ad3 : (i == 4 ? ad[2] : (i == 5 ? ad5 : ad[3])))));
      a[i] = *reinterpret_cast<const bf16x8_t*>(pb + base + i * 4096);
    };
    addr_step(0);
#pragma unroll
    for (int i = 0; i < K8_FP; ++i) read_frag(i);

    // K-step ks: 14 MFMAs (step 0 starts the accumulators from 0); behind fragment 0's pair the weight pieces of step ks + D go out
    // (slot ks % D: its fragments are in registers by then), then the wave waits for ITS pieces of step ks + 1 and reads their two
    // fragments; fragment i of step ks + 1 is read into a[i] right behind the MFMAs that consumed it.
    auto kstep = [&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      constexpr int kn = (ks + 1) % K8_NK;
      if (ks + 1 < K8_NK) { __builtin_amdgcn_sched_barrier(0); addr_step(ks + 1); }
#pragma unroll
      for (int i = 0; i < K8_FP; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[ks & 1][j], a[i], ks == 0 ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) {
          issue_w((ks + K8_D) % K8_NK);
          if (k8_younger<kn, true>() == k8_younger<kn, false>()) k8wait_vm<k8_younger<kn, false>()>();
          else if (fst) k8wait_vm<k8_younger<kn, true>()>();        // (wave-uniform: the first tile of a workgroup has no stores in front of it)
          else k8wait_vm<k8_younger<kn, false>()>();
          read_b(kn, bb[(ks + 1) & 1]);
        }
        if (ks + 1 < K8_NK) read_frag(i);
        if (IN && i == 3 && ks >= K8_D && ks < K8_D + K8_NPW) normalise_piece(ks - K8_D, buf ^ 1, y0n);   // (landed: older than every piece waited for since)
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#define K8S(n) kstep(std::integral_constant<int, n>{});
    K8S(0) K8S(1) K8S(2) K8S(3) K8S(4) K8S(5) K8S(6) K8S(7) K8S(8) K8S(9) K8S(10) K8S(11) K8S(12) K8S(13) K8S(14) K8S(15) K8S(16) K8S(17)
    K8S(18) K8S(19) K8S(20) K8S(21) K8S(22) K8S(23) K8S(24) K8S(25) K8S(26) K8S(27) K8S(28) K8S(29) K8S(30) K8S(31) K8S(32) K8S(33) K8S(34) K8S(35)
#undef K8S
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: the tile's 112 pixels are one contiguous run of the NHWC output (4 full image rows)
    const long b = (unsigned)tile / (unsigned)K8_TILES_H;
    const int y0 = tile_y0(tile);
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + ((b * K8_H + y0) * (long)K8_W) * 128), 0,
                                                                           p.no_store ? 0 : K8_TH * K8_W * 256, 0x00020000);
    // fragment i: accumulators -> (bias, statistics, ReLU) -> bf16 -> strip i & 1; its strip read is issued BEFORE fragment i + 1 is
    // converted and written (other strip), its store behind that: the LDS round trip of one fragment hides under the arithmetic of
    // the next (one wave per SIMD: nothing else would cover it)
    auto stage_frag = [&](int i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r];
          if constexpr (AFF) v[r] += bv[j][r];
          if constexpr (ST) { s1[j][r] += v[r]; s2[j][r] = fmaf(v[r], v[r], s2[j][r]); }
          if constexpr (AFF) v[r] = p.relu ? fmaxf(v[r], 0.f) : v[r];
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm, like the normalisation's store: a visible LDS store would make hipcc drain the weight ring and the next patch;
        //  the strip is private to the wave and a wave's LDS operations execute in order, so the read needs no wait for the write)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + (i & 1) * 1024 + frow * 64 + (((j * 2 + (fgrp >> 1)) ^ (frow >> 2)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
    };
    stage_frag(0);
#pragma unroll
    for (int i = 0; i < K8_FP; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      const u32x4_t val = *reinterpret_cast<const u32x4_t*>(stg + (i & 1) * 1024 + px * 64 + ((cq ^ (px >> 2)) << 4));
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < K8_FP) stage_frag(i + 1);
      __builtin_amdgcn_sched_barrier(0);
      // (every store is ISSUED, statistics-only launches too -- their descriptor's range is empty --, so that the waits can count them)
      __builtin_amdgcn_raw_buffer_store_b128(val, srd_o, (16 * i + px) * 256 + wave * 64 + cq * 16, 0, 0);
    }
    if (IN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my normalised chunks of the next patch are written
    buf ^= 1;
  }
  k8wait_vm<0>();
  if constexpr (ST) {
    float* const row = p.stats + (long)blockIdx.x * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = k8row16_sum(s1[j][r]), t2 = k8row16_sum(s2[j][r]);
        if (frow == 0) {
          row[32 * wave + 16 * j + 4 * fgrp + r] = t1;
          row[128 + 32 * wave + 16 * j + 4 * fgrp + r] = t2;
        }
      }
  }
}

// Tried and removed (round 3): a SPLIT-K form -- the four waves as two pairs, a pair owning 64 output channels and its two waves taking the
// K-steps of one parity each (28 MFMAs against 7 + 4 fragment reads per step: 0.39 KiB of LDS reads per MFMA instead of 0.64, half the
// per-step overhead), partial sums exchanged through the dead patch buffer at the end of a tile.  Correct (same tests), and no faster:
// 1722 us against 1620-1650 us.  Neither is the depth of the weight rings a lever (4 slots: 1650 us; 6: 1620-1650).  Three designs --
// the generic 256x128 tiles, this kernel, the split-K form -- land within 5 % of each other: at the clock the part holds on this data
// (1.4-1.55 GHz inside the 3x3 kernels) the tile's 504 MFMAs per wave are ~70 % of the K loop's time, and the
// rest is the epilogue a single wave per SIMD cannot hide.

template <bool AFF, bool ST, bool IN = false>
__global__ __launch_bounds__(256, 1) void conv3x3_c128_kernel(const K8Args p) { k8_body<AFF, ST, IN>(p); }
template <bool AFF, bool ST, bool IN = false> struct K8Tag {};

template <bool AFF, bool ST, bool IN = false>
int k8_launch(const K8Args& s, unsigned grid, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<K8Tag<AFF, ST, IN>>(reinterpret_cast<const void*>(&conv3x3_c128_kernel<AFF, ST, IN>), K8_LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_c128_kernel<AFF, ST, IN>), dim3(grid), dim3(256), K8_LDS, st, s);
  return SR_OK;
}

inline bool k8_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_C3_128"); return e && e[0] == '1'; }();
  return !off;
}
inline bool k8_serves(const sr_conv_args* a) {
  return k8_enabled() && !a->stem && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Cin == 128 && a->Cout == 128 && a->W == K8_W &&
         a->H == K8_H && !a->res && !a->escale && a->B > 0;
}
inline unsigned k8_grid(long ntiles) {
  const long cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

}  // namespace

// Internal hand-over from sr_conv2d / sr_conv_stats_rows (gemm.hip): SR_ERR_UNSUPPORTED when the launch is not this layer shape
int srx_c3d128_rows(const sr_conv_args* a) {
  if (!k8_serves(a)) return SR_ERR_UNSUPPORTED;
  return (int)k8_grid((long)a->B * K8_TILES_H);
}

// Does the direct kernel serve this launch WITH an input affine?  (the train-mode form: raw output + statistics, no bias / ReLU)
bool srx_c3d128_in_affine_ok(const sr_conv_args* a) {
  return k8_serves(a) && a->act == SR_ACT_NONE && !a->bias && a->stats != nullptr;
}

int srx_c3d128_conv(const sr_conv_args* a, void* stream) {
  if (!k8_serves(a) || (a->act != SR_ACT_NONE && a->act != SR_ACT_RELU)) return SR_ERR_UNSUPPORTED;
  if ((a->in_scale || a->in_shift) && (!a->in_scale || !a->in_shift || !srx_c3d128_in_affine_ok(a))) return SR_ERR_UNSUPPORTED;
  K8Args s;
  s.x = (const bf16_t*)a->x; s.w = (const bf16_t*)a->w; s.y = (bf16_t*)a->y; s.bias = a->bias; s.stats = a->stats;
  s.B = a->B; s.relu = a->act == SR_ACT_RELU; s.no_store = a->no_store;
  s.in_scale = a->in_scale; s.in_shift = a->in_shift;
  const long ntiles = (long)s.B * K8_TILES_H;
  if (ntiles > 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  SR_ROUTE(SR_ROUTE_C3D128);
  const unsigned grid = k8_grid(ntiles);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (a->in_scale) rc = k8_launch<false, true, true>(s, grid, st);
  else {
    const bool aff = a->bias != nullptr || s.relu, stt = a->stats != nullptr;
    rc = aff ? (stt ? k8_launch<true, true>(s, grid, st) : k8_launch<true, false>(s, grid, st))
             : (stt ? k8_launch<false, true>(s, grid, st) : k8_launch<false, false>(s, grid, st));
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}
