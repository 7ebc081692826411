// 3x3 / stride 1 / pad 1 convolution with 256 input and 256 output channels on 14 x 14 images (ResNet layer3 conv2: 35 launches per
// backbone pass, the largest single item of the training step; call site reference model.py:35) as a DIRECT convolution for gfx950.
//
// On the generic implicit-GEMM kernel (gemm.hip, 256x256 tiles) this layer pulls every input pixel through LDS-DMA nine times (16 KiB of
// activations AND 16 KiB of weights per K-step) and its input has to be normalised by a separate sweep (bn1 -> relu: `bn_apply`, 0.24 ms
// of pure HBM traffic per block, 16.5 ms per training step) because an in-LDS normalisation would run nine times per element.  A
// 256-channel patch of a whole image does not fit the LDS next to a weight ring (16 x 16 padded pixels x 512 B = 128 KiB: DESIGN.md,
// round 3) -- but a 32-CHANNEL SLICE of it does.  Here (the design of c3d128.hip with the input cut into channel chunks):
//   * a workgroup's tile is ONE IMAGE: 196 output pixels = 13 fragments of 16 (the last one carries 4 pixels; its other lanes read a
//     pad pixels only -- zeros -- and are dropped by the output descriptor's range);
//   * K order = (channel chunk, tap): the K loop walks 8 chunks of 32 input channels; a chunk's patch -- 16 rows x 18 padded pixels x
//     64 B = 18 KiB -- is staged ONCE by LDS-DMA (pad pixels are out-of-range buffer loads: zeros) and serves all nine taps = nine
//     K-steps; two chunk buffers alternate, chunk c + 1 lands while chunk c is multiplied (one workgroup barrier per chunk, one step
//     before the chunk's end, so that the next chunk's first fragments are read ahead like any other step's);
//   * FOUR waves, one per SIMD, split the OUTPUT CHANNELS: wave w owns couts 64w .. 64w+63 (4 weight fragments) for all 13 pixel
//     fragments = 52 MFMAs (v_mfma_f32_16x16x32_bf16) per K-step against 13 + 4 fragment reads;
//   * the weights (256 x 2304 bf16 = 1.18 MB, L2 resident) are streamed, each wave only ever reading its OWN 64 rows through a PRIVATE
//     6-slot LDS ring (4 KiB per K-step), filled by its own LDS-DMA and retired by its own counted vmcnt: no barrier inside a chunk;
//   * BatchNorm + ReLU of the layer in front (train mode: in_scale / in_shift) is applied to the NEXT chunk's patch in LDS, by the lanes
//     that loaded it, in the shadow of the current chunk's MFMAs -- once per element: the normalised tensor is never written and the
//     `bn_apply` sweep in front of layer3's 3x3 is gone;
//   * LDS bank conflicts: patch pixel P = 18 pr + pc is 64 B; its 16-byte chunk c sits at chunk position c ^ key, key = ((14 pr + pc) >> 1)
//     & 3 -- the pixel's index at the IMAGE's pitch, halved: the same for all 13 fragments of a tap (16 i = 0 mod 8) -- 4.2 LDS cycles
//     per ds_read_b128 over all fragments and taps (4 = conflict free; enumerated with the instruction's lane groups; with a patch pitch
//     of 16 no key of this family gets below 7.2).  Weight-row chunk c of row n sits at c ^ (-(n >> 2) & 3), as in c3d128.hip;
//   * epilogue: BatchNorm partial sums folded per tile into a per-wave LDS row (one row of partial statistics per workgroup), or bias +
//     ReLU (eval mode); bf16 through a per-wave staging strip, two 16-byte stores per lane and pixel fragment.
// Same interface as the generic path (sr_conv2d); the partial-statistics row count comes from sr_conv_stats_rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct K6Args {
  const bf16_t* x;          // [B, 14, 14, 256]
  const bf16_t* w;          // [256][9][256]   (K index = tap * 256 + channel)
  bf16_t* y;                // [B, 14, 14, 256]
  const float* bias;        // [256] or null
  float* stats;             // [grid][2][256] or null
  int B, relu, no_store;
  const float* in_scale; const float* in_shift;   // [256] or null: the convolution runs on relu(x*in_scale + in_shift)
};

constexpr int K6_W = 14, K6_PW = 18, K6_PR = 16, K6_C = 256;
constexpr int K6_NPIX = K6_W * K6_W;                      // 196 output pixels per tile (= image)
constexpr int K6_FP = (K6_NPIX + 15) / 16;                // 13 pixel fragments
constexpr int K6_NCH = 8, K6_CHB = 64;                    // 8 chunks of 32 channels = 64 B per pixel
constexpr int K6_NP = K6_PR * K6_PW * K6_CHB / 1024;      // 18 LDS-DMA pieces per chunk patch
constexpr int K6_NPW = 5;                                 // ... 5 per wave (pieces 18, 19: out of range, zeros behind the patch)
constexpr int K6_PBUF = 4 * K6_NPW * 1024;                // 20 KiB per chunk buffer
constexpr int K6_D = 6;                                   // depth of a wave's weight ring (divides the 18 K-steps of a chunk pair: static slots)
constexpr int K6_NKC = 9;                                 // K-steps per chunk (taps)
constexpr int K6_WSTEP = 4096;                            // ring bytes per K-step and wave: 64 rows x 64 B
constexpr int K6_WRING = 2 * K6_PBUF, K6_STG = K6_WRING + 4 * K6_D * K6_WSTEP, K6_VEC = K6_STG + 4 * 4096, K6_STAT = K6_VEC + 1024 + 2048;
constexpr int K6_LDS = K6_STAT + 4 * 512;               // ... | per-wave [2][64] running BatchNorm sums
constexpr int K6_NST = 2 * K6_FP;                         // output stores per wave and tile
constexpr int K6_OOB = (int)0x80000000;
static_assert(K6_NP <= 4 * K6_NPW && (2 * K6_NKC) % K6_D == 0 && K6_LDS <= 160 * 1024, "tile / LDS budget");
static_assert(4 * (K6_D - 1) + K6_NPW + K6_NST < 64, "vmcnt is a 6-bit counter");

template <int N> __device__ __forceinline__ void k6wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc (AccVGPRs, updated IN PLACE) += W fragment x activation fragment.  Inline asm pins the 52 accumulator fragments to exactly 208
// AccVGPRs: left to itself hipcc renames the destinations (64 quads = all 256 AccVGPRs), and the ArchVGPR side -- 13 + 8 operand
// fragments, addresses, 32 running BatchNorm sums -- then has nowhere cheap to spill to and goes to scratch, whose traffic would break
// the counted vmcnt waits.
__device__ __forceinline__ void k6mma(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void k6mma0(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {   // a tile's first K-step: C = 0
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a));
}

__device__ __forceinline__ float k6row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// Vector-memory operations a wave has issued AFTER the weight pieces of the K-step it waits for.  The wait sits in step s (local index
// sl = s % 9 inside its chunk), right behind that step's own weight issue (the pieces of step s + D), for the pieces of step s + 1,
// which went out in step s + 1 - D:
//   always                      the 4 pieces of each of the D - 2 steps s + 2 .. s + D - 1 (the pieces of step s + D go out BEHIND the wait,
//                               one per three pixel fragments: four LDS-DMA issues in a row stall the matrix pipe -- 1177 -> 1152 us;
//                               spreading the five patch pieces of the barrier step the same way gained nothing);
//   PATCH (sl = 8, 0, 1, 2, 3)  the 5 patch pieces issued at the start of the latest step with sl = 8 (behind the chunk barrier, in front
//                               of that step's weight issue) -- except in the first chunk of a workgroup's first tile, whose patches
//                               went out in the prologue, in front of every weight piece;
//   STORES (sl = 0 .. 4 of a tile's first chunk, not the workgroup's first tile)  the previous tile's 26 output stores.
template <bool PATCH, bool STORES> constexpr int k6_younger() { return 4 * (K6_D - 2) + (PATCH ? K6_NPW : 0) + (STORES ? K6_NST : 0); }

// AFF: bias (+ ReLU) in the epilogue (eval mode: folded BatchNorm).  ST: BatchNorm partial statistics (train mode).
// IN: the input is the RAW output of the preceding convolution; its BatchNorm + ReLU (in_scale, in_shift) is applied to the patch in LDS.
template <bool AFF, bool ST, bool IN>
__device__ __forceinline__ void k6_body(const K6Args& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 chunk buffers | 4 weight rings | 4 x 2 staging strips | bias | in-affine
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const long ntiles = p.B;
  const int G = gridDim.x;

  float* const lbias = reinterpret_cast<float*>(smem + K6_VEC);
  float* const inaff = reinterpret_cast<float*>(smem + K6_VEC + 1024);
  lbias[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
  if (IN) {                                    // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
    const int ch = (threadIdx.x & ~7) | sr_pair_order(threadIdx.x & 7);
    inaff[threadIdx.x] = p.in_scale[ch]; inaff[K6_C + threadIdx.x] = p.in_shift[ch];
  }

  // ---- patch loader.  Piece q = i*4 + wave lands at LDS bytes q*1024 + lane*16 of the chunk buffer: patch pixel P = 16 q + lane/4
  // (P = 18 pr + pc), chunk position lane%4, which holds data chunk (lane%4) ^ key(P) of that pixel's 64-byte channel slice.  Source
  // offsets are relative to the image's first pixel, the chunk's channel offset is the instruction's scalar offset; pad pixels and
  // pieces past the patch carry the out-of-range marker (zeros).
  int vrel[K6_NPW];
#pragma unroll
  for (int i = 0; i < K6_NPW; ++i) {
    const int q = i * 4 + wave, P = q * 16 + (lane >> 2);
    const int pr = P / K6_PW, pc = P - pr * K6_PW;
    const int key = ((K6_W * pr + pc) >> 1) & 3, cdat = (lane & 3) ^ key;
    // (bits 0..19: the offset; bits 24..25: the data chunk, for the IN kernels' scale / shift lookup)
    vrel[i] = (pr >= 1 && pr <= K6_W && pc >= 1 && pc <= K6_W) ? ((((pr - 1) * K6_W + pc - 1) * (K6_C * 2) + (cdat << 4)) | (cdat << 24)) : K6_OOB;
  }
  auto issue_patch = [&](long tile, int chunk, int buf, bool valid) {
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)((uintptr_t)p.x + (valid ? tile : 0) * (long)(K6_NPIX * K6_C * 2)), 0,
                                                                         valid ? K6_NPIX * K6_C * 2 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < K6_NPW; ++i) {
      const int vo = vrel[i] < 0 ? K6_OOB : (vrel[i] & 0xfffff);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * K6_PBUF + (i * 4 + wave) * 1024), 16, vo,
                                               chunk * K6_CHB, 0, 0);
    }
  };
  // IN: BatchNorm + ReLU of the layer in front on the 16-byte chunks THIS lane loaded (their 8 channels: chunk * 32 + 8 * bits 24..25 of
  // vrel; scale and shift come from the LDS table).  Pad pixels were zero-filled by the loader and must stay zero: the convolution pads
  // the NORMALISED tensor.
  auto normalise_piece = [&](int i, int chunk, int buf) {
    if (vrel[i] < 0) return;
    const int c8 = chunk * 32 + ((vrel[i] >> 24) & 3) * 8;
    const sr_f32x4 ns0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), ns1 = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
    const sr_f32x4 nh0 = *reinterpret_cast<const sr_f32x4*>(inaff + K6_C + c8), nh1 = *reinterpret_cast<const sr_f32x4*>(inaff + K6_C + c8 + 4);
    char* const at = smem + buf * K6_PBUF + (i * 4 + wave) * 1024 + lane * 16;
    const sr_u32x4 nv = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(at), ns0, ns1, nh0, nh1);
    // (inline asm: in front of an LDS store it can see, hipcc drains every vector-memory operation -- LDS-DMA may alias)
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)at), "v"(nv) : "memory");
  };

  // ---- weight ring of this wave: a K-step = rows 64 wave .. +63 of W, 64 bytes each at byte offset tap * 512 + chunk * 64 of the row:
  // four pieces (one per 16-row fragment); lane l of a piece -> row l/4, chunk position l%4, which holds data chunk (l%4) ^ (-(row >> 2) & 3)
  char* const wring = smem + K6_WRING + wave * (K6_D * K6_WSTEP);
  int wvo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = lane >> 2, cd = (lane & 3) ^ ((0 - (n >> 2)) & 3);
    wvo[j] = ((64 * wave + 16 * j + n) * (9 * K6_C) + cd * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, K6_C * 9 * K6_C * 2, 0x00020000);
  auto issue_w1 = [&](int slot, int j, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(wring + slot * K6_WSTEP + j * 1024), 16, wvo[j], soff, 0, 0);
  };
  auto issue_w = [&](int slot, int soff) {   // slot = compile time at every call site; soff = tap * 512 + chunk * 64 (scalar)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(wring + slot * K6_WSTEP + j * 1024), 16, wvo[j], soff, 0, 0);
  };
  const int boff = frow * 64 + ((fgrp ^ ((0 - (frow >> 2)) & 3)) << 4);
  auto read_b = [&](int slot, bf16x8_t (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(wring + slot * K6_WSTEP + j * 1024 + boff);
  };

  char* const stg = smem + K6_STG + wave * 4096;        // two 2 KiB strips per wave (fragment i uses strip i & 1)

  // bias of this lane's 4 x 4 output channels (couts 64 wave + 16 j + 4 fgrp + r).  The BatchNorm partial sums do NOT live in registers
  // across the K loop (32 of them pushed the train-mode kernels into scratch, whose traffic the counted vmcnt waits cannot see): every
  // tile's sums are folded (16-lane DPP sums, then LDS float adds by four lanes) into the wave's LDS row, written out once per kernel.
  float bv[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
  float* const lstat = reinterpret_cast<float*>(smem + K6_STAT) + wave * 128;      // this wave's [2][64] running sums
  lstat[lane] = 0.f; lstat[64 + lane] = 0.f;

  long tile = blockIdx.x;
  issue_patch(tile, 0, 0, tile < ntiles);
  issue_patch(tile, 1, 1, tile < ntiles);
  k6wait_vm<0>();
  __syncthreads();                                      // bias / in-affine tables; my patch pieces have landed
  if (AFF) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = lbias[64 * wave + 16 * j + 4 * fgrp + r];
  }
  if (IN) {                                             // chunk 0 of the first tile (chunk 1 follows inside the K loop, as every later chunk)
#pragma unroll
    for (int i = 0; i < K6_NPW; ++i) normalise_piece(i, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // the weight pieces of the first D K-steps (chunk 0, taps 0 .. 5)
#pragma unroll
  for (int k = 0; k < K6_D; ++k) issue_w(k, k * 512);
  bf16x8_t bb[2][4];                                    // weight fragments of the current / the next K-step (step s18 uses bb[s18 & 1]; 18 is even)
  k6wait_vm<4 * (K6_D - 1)>();                          // the pieces of step 0
  read_b(0, bb[0]);

  const int px8 = lane >> 3, cq8 = lane & 7;
  bool first = true;
  for (; tile < ntiles; tile += G) {
    const bool fst = first;
    first = false;
    const long tnext = tile + G;

    f32x4_t acc[K6_FP][4];                              // (written, not accumulated into, by the MFMAs of the tile's first K-step)
    bf16x8_t a[K6_FP];
    int z;                                              // an opaque 0, new per tile: without it the compiler computes every step's addresses once
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));          // per kernel and keeps them alive across the tile loop (spills: c3d128.hip)
    // byte address of fragment i's pixel for tap (0, 0) inside a chunk buffer: output pixel t = 16 i + frow sits at patch pixel
    // (t / 14) * 18 + t % 14 = t + 4 (t / 14); lanes past the image (fragment 12: t >= 196) start at patch pixel (15, 0): its nine taps
    // are pixels 270..272 (the pad row below the image) and 288..290, 306..308 (pieces 18 and 19: out-of-range loads) -- all zeros, so
    // those lanes' accumulators stay exactly 0 and add nothing to the BatchNorm sums
    int base[K6_FP];
#pragma unroll
    for (int i = 0; i < K6_FP; ++i) {
      const int t = 16 * i + frow + z;
      base[i] = t < K6_NPIX ? (t + 4 * (t / K6_W)) * K6_CHB : (K6_PR - 1) * K6_PW * K6_CHB;
    }

    for (int cp = 0; cp < K6_NCH / 2; ++cp) {           // two chunks (18 K-steps, fully unrolled) per trip: ring slots and buffers are static
      const int soff_cp = cp * 128, soff_nx = ((cp + 1) & 3) * 128;
      int zc;                                           // (an opaque 0 per trip, as `z` per tile: the 18 steps' chunk positions are invariant over
      asm volatile("v_mov_b32 %0, 0" : "=v"(zc));       //  the trips and would be hoisted out of the loop -- 36 live registers, spills)

      auto kstep = [&](auto S18) {
        constexpr int s18 = decltype(S18)::value;       // 0 .. 17 inside the chunk pair
        constexpr int half = s18 / K6_NKC, sl = s18 % K6_NKC, tap = sl;
        constexpr int buf = half;
        const char* const pb = smem + buf * K6_PBUF;
        // fragment addresses of this tap: chunk position fgrp ^ key, key = ((t + 14 r + q) >> 1) & 3 (16 i = 0 mod 8: one key per lane and tap)
        constexpr int tr = tap / 3, tq = tap % 3;
        if constexpr (s18 == 0) {                       // a trip's first step: nothing is read ahead across the loop's back edge (it would keep
          const int pos = ((fgrp ^ (((frow + zc) >> 1) & 3)) << 4);   // 52 fragment registers alive across it and across the epilogue: spills)
#pragma unroll
          for (int i = 0; i < K6_FP; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(pb + base[i] + pos);
        }
        if constexpr (sl == K6_NKC - 1) {
          // ---- the chunk barrier, ONE STEP BEFORE the chunk ends: my reads of this buffer have returned (the last tap's fragments were
          // read a step ago) and my normalised chunks of the other buffer are written; behind the barrier that holds for everybody, so
          // (1) this buffer may be refilled -- with the chunk after next --, and (2) the next chunk's first fragments can be read AHEAD,
          // behind this step's MFMAs, instead of in the open at the start of the next chunk (13 reads of exposed LDS latency per chunk).
          // Everybody's pieces of the next chunk have landed: each wave's wait in step sl = 4 (for weight pieces issued behind them)
          // covered its own.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          const int c2 = 2 * cp + half + 2;             // the chunk after next (8, 9 = the next tile's chunks 0, 1)
          if (c2 < K6_NCH) issue_patch(tile, c2, buf, true);
          else issue_patch(tnext, c2 - K6_NCH, buf, tnext < ntiles);
        }
        // the tap that follows: its fragments are read into a[i] right behind the MFMAs that consumed a[i] -- within the chunk from this
        // buffer, in the first chunk's last step from the OTHER buffer (the second chunk's tap 0)
        constexpr int ntap = (tap + 1) % K6_NKC, ntr = ntap / 3, ntq = ntap % 3;
        const char* const pbn = smem + (sl + 1 < K6_NKC ? buf : buf ^ 1) * K6_PBUF;
        const int npos = ((fgrp ^ (((frow + zc + K6_W * ntr + ntq) >> 1) & 3)) << 4) + (ntr * K6_PW + ntq) * K6_CHB;
        constexpr bool ahead = s18 != 17;
#pragma unroll
        for (int i = 0; i < K6_FP; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          if (s18 == 0 && cp == 0) {                    // (wave-uniform; only step 0 of a trip carries both forms)
#pragma unroll
            for (int j = 0; j < 4; ++j) k6mma0(acc[i][j], bb[s18 & 1][j], a[i]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k6mma(acc[i][j], bb[s18 & 1][j], a[i]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i == 1 || i == 4 || i == 7 || i == 10) {
            // one weight piece of step s + D (slot (s18 + D) % D = s18 % D: its fragments are in registers)
            constexpr int k18 = s18 + K6_D;
            constexpr int kk = k18 % 18;
            issue_w1(s18 % K6_D, (i - 1) / 3, (kk % K6_NKC) * 512 + (kk / K6_NKC) * 64 + (k18 < 18 ? soff_cp : soff_nx));
          }
          if (i == 0) {
            // the wave waits for ITS weight pieces of step s + 1 and reads their four fragments
            if constexpr (sl == K6_NKC - 1) {
              k6wait_vm<k6_younger<true, false>()>();
            } else if constexpr (sl <= 4) {
              constexpr bool patch = sl <= 3;
              if (half == 0 && cp == 0) {               // (wave-uniform) a tile's first chunk
                if (fst) k6wait_vm<k6_younger<false, false>()>(); else k6wait_vm<k6_younger<patch, true>()>();
              } else {
                k6wait_vm<k6_younger<patch, false>()>();
              }
            } else {
              k6wait_vm<k6_younger<false, false>()>();
            }
            read_b((s18 + 1) % K6_D, bb[(s18 + 1) & 1]);
          }
          if constexpr (ahead) a[i] = *reinterpret_cast<const bf16x8_t*>(pbn + base[i] + npos);
          // the next chunk's patch has landed (it is older than the weight pieces waited for in step sl = 4): its pieces are normalised
          // in steps 5, 6, 7 -- in front of the barrier of step 8
          if constexpr (IN && sl >= 5 && sl <= 7) {
            const int nchunk = (2 * cp + half + 1) & 7;
            if (i == 3) normalise_piece(sl == 5 ? 0 : (sl == 6 ? 2 : 4), nchunk, buf ^ 1);
            if (i == 8 && sl < 7) normalise_piece(sl == 5 ? 1 : 3, nchunk, buf ^ 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#define K6S(n) kstep(std::integral_constant<int, n>{});
      K6S(0) K6S(1) K6S(2) K6S(3) K6S(4) K6S(5) K6S(6) K6S(7) K6S(8) K6S(9) K6S(10) K6S(11) K6S(12) K6S(13) K6S(14) K6S(15) K6S(16) K6S(17)
#undef K6S
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: the tile's 196 pixels are one contiguous run of the NHWC output (the whole image); this wave writes its 64 channels
    // (128 B) of every pixel; pixels past the image fall outside the descriptor
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + tile * (long)(K6_NPIX * K6_C)), 0,
                                                                           p.no_store ? 0 : K6_NPIX * K6_C * 2, 0x00020000);
    // fragment i: accumulators -> (bias, statistics, ReLU) -> bf16 -> strip i & 1; its strip reads are issued BEFORE fragment i + 1 is
    // converted and written (other strip), its stores behind that (one wave per SIMD: nothing else would cover the LDS round trip)
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
    auto stage_frag = [&](int i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r];
          if constexpr (AFF) v[r] += bv[j][r];
          if constexpr (ST) {
            // (the last fragment's lanes past the image: exact zeros without a bias -- their patch pixels are pads -- but conv + bias
            //  with one, which must not enter the sums)
            const float vs = (AFF && i == K6_FP - 1 && frow >= K6_NPIX - 16 * (K6_FP - 1)) ? 0.f : v[r];
            s1[j][r] += vs; s2[j][r] = fmaf(vs, vs, s2[j][r]);
          }
          if constexpr (AFF) v[r] = p.relu ? fmaxf(v[r], 0.f) : v[r];
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm, like the normalisation's store: a visible LDS store would make hipcc drain the weight ring and the next patch;
        //  the strip is private to the wave and a wave's LDS operations execute in order, so the read needs no wait for the write)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + (i & 1) * 2048 + frow * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
    };
    stage_frag(0);
#pragma unroll
    for (int i = 0; i < K6_FP; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      u32x4_t val[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pxl = h * 8 + px8;
        val[h] = *reinterpret_cast<const u32x4_t*>(stg + (i & 1) * 2048 + pxl * 128 + ((cq8 ^ (pxl & 7)) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < K6_FP) stage_frag(i + 1);
      __builtin_amdgcn_sched_barrier(0);
      // (every store is ISSUED, statistics-only launches too -- their descriptor's range is empty --, so that the waits can count them)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_buffer_store_b128(val[h], srd_o, (16 * i + h * 8 + px8) * (K6_C * 2) + wave * 128 + cq8 * 16, 0, 0);
    }
    if constexpr (ST) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t1 = k6row16_sum(s1[j][r]), t2 = k6row16_sum(s2[j][r]);
          if (frow == 0) {
            // (inline asm, like the strip stores: hipcc guards a visible LDS atomic with a wait for every vector-memory operation in flight)
            const unsigned at = (unsigned)(uintptr_t)(lstat + 16 * j + 4 * fgrp + r);
            asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:256" ::"v"(at), "v"(t1), "v"(t2) : "memory");
          }
        }
    }
  }
  k6wait_vm<0>();
  if constexpr (ST) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* const row = p.stats + (long)blockIdx.x * (2 * K6_C);
    row[64 * wave + lane] = lstat[lane];
    row[K6_C + 64 * wave + lane] = lstat[64 + lane];
  }
}

template <bool AFF, bool ST, bool IN = false>
__global__ __launch_bounds__(256, 1) void conv3x3_c256_kernel(const K6Args p) { k6_body<AFF, ST, IN>(p); }
template <bool AFF, bool ST, bool IN = false> struct K6Tag {};

template <bool AFF, bool ST, bool IN = false>
int k6_launch(const K6Args& s, unsigned grid, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<K6Tag<AFF, ST, IN>>(reinterpret_cast<const void*>(&conv3x3_c256_kernel<AFF, ST, IN>), K6_LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_c256_kernel<AFF, ST, IN>), dim3(grid), dim3(256), K6_LDS, st, s);
  return SR_OK;
}

inline bool k6_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_C3_256"); return e && e[0] == '1'; }();
  return !off;
}
inline bool k6_serves(const sr_conv_args* a) {
  return k6_enabled() && !a->stem && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Cin == K6_C && a->Cout == K6_C && a->W == K6_W &&
         a->H == K6_W && !a->res && !a->escale && a->B > 0;
}
inline unsigned k6_grid(long ntiles) {
  const long cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

}  // namespace

// Internal hand-over from sr_conv2d / sr_conv_stats_rows (gemm.hip): SR_ERR_UNSUPPORTED when the launch is not this layer shape
int srx_c3d256_rows(const sr_conv_args* a) {
  if (!k6_serves(a)) return SR_ERR_UNSUPPORTED;
  return (int)k6_grid((long)a->B);
}

// Does the direct kernel serve this launch WITH an input affine?  (the train-mode form: raw output + statistics, no bias / ReLU)
bool srx_c3d256_in_affine_ok(const sr_conv_args* a) {
  return k6_serves(a) && a->act == SR_ACT_NONE && !a->bias && a->stats != nullptr;
}

int srx_c3d256_conv(const sr_conv_args* a, void* stream) {
  if (!k6_serves(a) || (a->act != SR_ACT_NONE && a->act != SR_ACT_RELU)) return SR_ERR_UNSUPPORTED;
  if ((a->in_scale || a->in_shift) && (!a->in_scale || !a->in_shift || !srx_c3d256_in_affine_ok(a))) return SR_ERR_UNSUPPORTED;
  K6Args s;
  s.x = (const bf16_t*)a->x; s.w = (const bf16_t*)a->w; s.y = (bf16_t*)a->y; s.bias = a->bias; s.stats = a->stats;
  s.B = a->B; s.relu = a->act == SR_ACT_RELU; s.no_store = a->no_store;
  s.in_scale = a->in_scale; s.in_shift = a->in_shift;
  SR_ROUTE(SR_ROUTE_C3D256);
  const unsigned grid = k6_grid((long)s.B);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (a->in_scale) rc = k6_launch<false, true, true>(s, grid, st);
  else {
    const bool aff = a->bias != nullptr || s.relu, stt = a->stats != nullptr;
    rc = aff ? (stt ? k6_launch<true, true>(s, grid, st) : k6_launch<true, false>(s, grid, st))
             : (stt ? k6_launch<false, true>(s, grid, st) : k6_launch<false, false>(s, grid, st));
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}
