// Direct 3x3 / stride 1 / pad 1 convolutions over CHANNEL SLICES of the input, for gfx950: ResNet layer3's conv2 (256 channels, 14 x 14:
// 35 launches per backbone pass, the largest single item of the training step) and layer2's (128 channels, 28 x 28: 7 launches).  Call
// site: reference model.py:35 (torchvision Bottleneck.conv2).
//
// On the generic implicit-GEMM kernel (gemm.hip, 256x256 tiles) layer3's 3x3 pulls every input pixel through LDS-DMA nine times (16 KiB
// of activations AND 16 KiB of weights per K-step) and its input has to be normalised by a separate sweep (bn1 -> relu: `bn_apply`,
// 0.24 ms of pure HBM traffic per block, 16.5 ms per training step) because an in-LDS normalisation would run nine times per element.
// A 256-channel patch of a whole image does not fit the LDS next to a weight ring (16 x 16 padded pixels x 512 B = 128 KiB: DESIGN.md,
// round 3) -- but a 32-CHANNEL SLICE of it does.  The design (round 3's 128-channel kernel -- whole patches of four image rows, per-wave private weight rings: c3d128.hip, removed in round 4 --
// with the input cut into channel chunks), layer3 numbers:
//   * a workgroup's tile is ONE IMAGE: 196 output pixels = 13 fragments of 16 (the last one carries 4 pixels; its other lanes read
//     whatever pixel 0's taps hold, are masked out of the BatchNorm sums and dropped by the output descriptor's range);
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
//   * LDS bank conflicts: patch pixel P = PW pr + pc is 64 B; its 16-byte chunk c sits at chunk position c ^ key, key = ((W pr + pc) >> 1)
//     & 3 -- the pixel's index at the IMAGE's pitch, halved: the same for all fragments of a tap (16 i = 0 mod 8) -- with a patch pitch of
//     PW = W + 4 pixels: 4.2 LDS cycles per ds_read_b128 over all fragments and taps for both image widths (4 = conflict free; enumerated
//     with the instruction's lane groups; with the natural pitch W + 2 no key of this family gets below 5.2 / 7.2).  Weight-row chunk c
//     of row n sits at c ^ (-(n >> 2) & 3) (4.0);
//   * epilogue: BatchNorm partial sums folded per tile into a per-wave LDS row (one row of partial statistics per workgroup), or bias +
//     ReLU (eval mode); bf16 through a per-wave staging strip, 16-byte stores.
// The same body, instantiated for LAYER2 (`KsL2`: 128 channels, 28 x 28): a tile is HALF an image (14 rows = 392 output pixels = 25
// fragments; the patch 16 rows x 32 padded pixels x 64 B = 32 KiB per slice; the rows above / below the tile are real image rows except
// at the image's top and bottom), four chunks, each wave 32 output channels = 2 weight fragments: 50 MFMAs per K-step against 25 + 2
// fragment reads -- where the round-3 kernel for this layer (whole 128-channel patches of four rows) ran 14 against 9: 1600 -> 1338 us.  (The
// 100 registers of its 25 pixel fragments fit only because a fragment's address is not a register of its own: see `base` below.)
// Same interface as the generic path (sr_conv2d); the partial-statistics row count comes from sr_conv_stats_rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct KsArgs {
  const bf16_t* x;          // [B, H, W, C]
  const bf16_t* w;          // [C][9][C]   (K index = tap * C + channel)
  bf16_t* y;                // [B, H, W, C]
  const float* bias;        // [C] or null
  float* stats;             // [grid][2][C] or null
  int B, relu, no_store;
  const float* in_scale; const float* in_shift;   // [C] or null: the convolution runs on relu(x*in_scale + in_shift)
};

// W x W images of C channels (Cin = Cout), TR image rows per tile, JW 16-channel weight fragments per wave (4 waves x 16 JW = C)
template <int W_, int C_, int TR_, int JW_>
struct KsCfg {
  static constexpr int W = W_, C = C_, TR = TR_, JW = JW_;
  static constexpr int PW = W + 4, PR = TR + 2;                 // patch pitch (pixels) and rows
  static constexpr int TPI = W / TR;                            // tiles per image
  static constexpr int NPIX = TR * W;                           // output pixels per tile
  static constexpr int FP = (NPIX + 15) / 16;                   // pixel fragments
  static constexpr int NCH = C / 32, CHB = 64;                  // chunks of 32 channels = 64 B per pixel
  static constexpr int NP = (PR * PW * CHB + 1023) / 1024;      // LDS-DMA pieces per chunk patch
  static constexpr int NPW = (NP + 3) / 4;                      // ... per wave (pieces past the patch: out of range, zeros)
  static constexpr int PBUF = 4 * NPW * 1024;
  static constexpr int D = 6;                                   // depth of a wave's weight ring (divides the 18 K-steps of a chunk pair: static slots)
  static constexpr int NKC = 9;                                 // K-steps per chunk (taps)
  static constexpr int WSTEP = JW * 1024;                       // ring bytes per K-step and wave: 16 JW rows x 64 B
  static constexpr int STRIP = 16 * 32 * JW;                    // staging strip: 16 pixels x 16 JW channels, bf16
  static constexpr int CPR = 2 * JW;                            // 16-byte chunks per strip row
  static constexpr int SPF = (16 * CPR) / 64 > 0 ? (16 * CPR) / 64 : 1;   // store instructions per fragment
  static constexpr int WRING = 2 * PBUF, STG = WRING + 4 * D * WSTEP, VEC = STG + 4 * 2 * STRIP, STAT = VEC + 4 * C + 8 * C;
  static constexpr int LDS = STAT + 4 * 2 * 16 * JW * 4;
  static constexpr int NST = SPF * FP;                          // output stores per wave and tile
  static constexpr int LIVE_LAST = NPIX - 16 * (FP - 1);        // valid lanes (frow) of the last fragment
  static_assert(W % TR == 0 && C % 64 == 0 && 64 * JW == C && NCH % 2 == 0 && (2 * NKC) % D == 0 && LDS <= 160 * 1024, "tile / LDS budget");
  static_assert(JW * (D - 2) + NPW + NST < 64, "vmcnt is a 6-bit counter");
};
typedef KsCfg<14, 256, 14, 4> KsL3;     // layer3: one image per tile
typedef KsCfg<28, 128, 14, 2> KsL2;     // layer2: half an image per tile

constexpr int KS_OOB = (int)0x80000000;

template <int N> __device__ __forceinline__ void kswait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc (AccVGPRs, updated IN PLACE) += W fragment x activation fragment.  Inline asm pins the accumulator fragments to exactly 4 FP JW
// AccVGPRs: left to itself hipcc renames the destinations (64 quads = all 256 AccVGPRs), and the ArchVGPR side -- operand fragments,
// addresses -- then has nowhere cheap to spill to and goes to scratch, whose traffic would break the counted vmcnt waits.
__device__ __forceinline__ void ksmma(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void ksmma0(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& a) {   // a tile's first K-step: C = 0
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a));
}

__device__ __forceinline__ float ksrow16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// Vector-memory operations a wave has issued AFTER the weight pieces of the K-step it waits for.  The wait sits in step s (local index
// sl = s % 9 inside its chunk), behind the MFMAs of the step's first pixel fragment, for the pieces of step s + 1, which went out in step
// s + 1 - D:
//   always                      the JW pieces of each of the D - 2 steps s + 2 .. s + D - 1 (the pieces of step s + D go out BEHIND the wait,
//                               spread over the pixel fragments: LDS-DMA issues in a row stall the matrix pipe -- 1177 -> 1152 us on
//                               layer3; spreading the patch pieces of the barrier step the same way gained nothing);
//   PATCH (sl = 8, 0, 1, 2, 3)  the NPW patch pieces issued at the start of the latest step with sl = 8 (behind the chunk barrier, in front
//                               of that step's wait) -- except in the first chunk of a workgroup's first tile, whose patches went out in
//                               the prologue, in front of every weight piece;
//   STORES (sl = 0 .. 4 of a tile's first chunk, not the workgroup's first tile)  the previous tile's output stores.
template <typename CF, bool PATCH, bool STORES> constexpr int ks_younger() { return CF::JW * (CF::D - 2) + (PATCH ? CF::NPW : 0) + (STORES ? CF::NST : 0); }

// AFF: bias (+ ReLU) in the epilogue (eval mode: folded BatchNorm).  ST: BatchNorm partial statistics (train mode).
// IN: the input is the RAW output of the preceding convolution; its BatchNorm + ReLU (in_scale, in_shift) is applied to the patch in LDS.
template <typename CF, bool AFF, bool ST, bool IN>
__device__ __forceinline__ void ks_body(const KsArgs& p) {
  constexpr int W = CF::W, C = CF::C, JW = CF::JW, FP = CF::FP, PW = CF::PW, NPW = CF::NPW, D = CF::D, NKC = CF::NKC, CHB = CF::CHB;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 chunk buffers | 4 weight rings | 4 x 2 staging strips | bias | in-affine | statistics rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fgrp = lane >> 4;
  const long ntiles = (long)p.B * CF::TPI;
  const int G = gridDim.x;

  float* const lbias = reinterpret_cast<float*>(smem + CF::VEC);
  float* const inaff = reinterpret_cast<float*>(smem + CF::VEC + 4 * C);
  for (int k = threadIdx.x; k < C; k += 256) {
    lbias[k] = p.bias ? p.bias[k] : 0.f;
    if (IN) {                                  // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
      const int ch = (k & ~7) | sr_pair_order(k & 7);
      inaff[k] = p.in_scale[ch]; inaff[C + k] = p.in_shift[ch];
    }
  }

  // ---- patch loader.  Piece q = i*4 + wave lands at LDS bytes q*1024 + lane*16 of the chunk buffer: patch pixel P = 16 q + lane/4
  // (P = PW pr + pc), chunk position lane%4, which holds data chunk (lane%4) ^ key(P) of that pixel's 64-byte channel slice.  Source
  // offsets are relative to image row y0 - 1 (the tile's descriptor starts there and ends with the image, so rows below the image are
  // out of range by themselves), the chunk's channel offset is the instruction's scalar offset; pad pixels and pieces past the patch
  // carry the out-of-range marker (zeros); the row above the image (an image's first tile: patch row 0) is masked per tile.
  int vrel[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = i * 4 + wave, P = q * 16 + (lane >> 2);
    const int pr = P / PW, pc = P - pr * PW;
    const int key = ((W * pr + pc) >> 1) & 3, cdat = (lane & 3) ^ key;
    // (bits 0..19: the offset; bits 20..24: the patch row; bits 26..27: the data chunk, for the IN kernels' scale / shift lookup)
    vrel[i] = (pr < CF::PR && pc >= 1 && pc <= W) ? (((pr * W + pc - 1) * (C * 2) + (cdat << 4)) | (pr << 20) | (cdat << 26)) : KS_OOB;
  }
  auto tile_y0 = [&](long tile) { const unsigned ut = (unsigned)tile; return (int)(ut - (ut / (unsigned)CF::TPI) * (unsigned)CF::TPI) * CF::TR; };
  auto issue_patch = [&](long tile, int chunk, int buf, bool valid) {
    const long b = (unsigned)tile / (unsigned)CF::TPI;
    const int y0 = tile_y0(tile);
    const long left = (long)(W - y0 + 1) * W * (C * 2);                      // bytes from row y0 - 1 to the end of the image
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((uintptr_t)p.x + (valid ? ((b * W + y0 - 1) * (long)W) * (C * 2) : 0)), 0, valid ? (int)left : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int vo = vrel[i] < 0 ? KS_OOB : (vrel[i] & 0xfffff);
      if (i * 64 < PW) vo = (y0 == 0 && ((vrel[i] >> 20) & 31) == 0) ? KS_OOB : vo;       // (patch row 0 lies in the first ceil(PW / 64) pieces of a wave)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + buf * CF::PBUF + (i * 4 + wave) * 1024), 16, vo,
                                               chunk * CHB, 0, 0);
    }
  };
  // IN: BatchNorm + ReLU of the layer in front on the 16-byte chunks THIS lane loaded (their 8 channels: chunk * 32 + 8 * bits 26..27 of
  // vrel; scale and shift come from the LDS table).  Pad pixels and rows outside the image were zero-filled by the loader and must stay
  // zero: the convolution pads the NORMALISED tensor.
  auto normalise_piece = [&](int i, int chunk, int buf, int y0) {
    const int row = y0 - 1 + ((vrel[i] >> 20) & 31);
    if (vrel[i] < 0 || row < 0 || row >= W) return;
    const int c8 = chunk * 32 + ((vrel[i] >> 26) & 3) * 8;
    const sr_f32x4 ns0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), ns1 = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
    const sr_f32x4 nh0 = *reinterpret_cast<const sr_f32x4*>(inaff + C + c8), nh1 = *reinterpret_cast<const sr_f32x4*>(inaff + C + c8 + 4);
    char* const at = smem + buf * CF::PBUF + (i * 4 + wave) * 1024 + lane * 16;
    const sr_u32x4 nv = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(at), ns0, ns1, nh0, nh1);
    // (inline asm: in front of an LDS store it can see, hipcc drains every vector-memory operation -- LDS-DMA may alias)
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)at), "v"(nv) : "memory");
  };

  // ---- weight ring of this wave: a K-step = rows 16 JW wave .. of W, 64 bytes each at byte offset tap * 2 C + chunk * 64 of the row:
  // JW pieces (one per 16-row fragment); lane l of a piece -> row l/4, chunk position l%4, which holds data chunk (l%4) ^ (-(row >> 2) & 3)
  char* const wring = smem + CF::WRING + wave * (D * CF::WSTEP);
  int wvo[JW];
#pragma unroll
  for (int j = 0; j < JW; ++j) {
    const int n = lane >> 2, cd = (lane & 3) ^ ((0 - (n >> 2)) & 3);
    wvo[j] = ((16 * JW * wave + 16 * j + n) * (9 * C) + cd * 8) * 2;
  }
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, C * 9 * C * 2, 0x00020000);
  auto issue_w1 = [&](int slot, int j, int soff) {   // slot, j = compile time at every call site; soff = tap * 2 C + chunk * 64 (scalar)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(wring + slot * CF::WSTEP + j * 1024), 16, wvo[j], soff, 0, 0);
  };
  const int boff = frow * 64 + ((fgrp ^ ((0 - (frow >> 2)) & 3)) << 4);
  auto read_b = [&](int slot, bf16x8_t (&b)[JW]) {
#pragma unroll
    for (int j = 0; j < JW; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(wring + slot * CF::WSTEP + j * 1024 + boff);
  };

  char* const stg = smem + CF::STG + wave * (2 * CF::STRIP);     // two strips per wave (fragment i uses strip i & 1)

  // bias of this lane's JW x 4 output channels (couts 16 JW wave + 16 j + 4 fgrp + r).  The BatchNorm partial sums do NOT live in
  // registers across the K loop (they pushed the train-mode kernels into scratch, whose traffic the counted vmcnt waits cannot see):
  // every tile's sums are folded (16-lane DPP sums, then LDS float adds by four lanes) into the wave's LDS row, written out once per kernel.
  float bv[JW][4];
#pragma unroll
  for (int j = 0; j < JW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
  float* const lstat = reinterpret_cast<float*>(smem + CF::STAT) + wave * (2 * 16 * JW);      // this wave's [2][16 JW] running sums
  if (lane < 16 * JW) { lstat[lane] = 0.f; lstat[16 * JW + lane] = 0.f; }

  long tile = blockIdx.x;
  issue_patch(tile, 0, 0, tile < ntiles);
  issue_patch(tile, 1, 1, tile < ntiles);
  kswait_vm<0>();
  __syncthreads();                                      // bias / in-affine tables; my patch pieces have landed
  if (AFF) {
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = lbias[16 * JW * wave + 16 * j + 4 * fgrp + r];
  }
  if (IN) {                                             // chunk 0 of the first tile (chunk 1 follows inside the K loop, as every later chunk)
    const int y00 = tile_y0(tile < ntiles ? tile : 0);
#pragma unroll
    for (int i = 0; i < NPW; ++i) normalise_piece(i, 0, 0, y00);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // the weight pieces of the first D K-steps (chunk 0, taps 0 .. 5)
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int j = 0; j < JW; ++j) issue_w1(k, j, k * (2 * C));
  bf16x8_t bb[2][JW];                                   // weight fragments of the current / the next K-step (step s18 uses bb[s18 & 1]; 18 is even)
  kswait_vm<JW * (D - 1)>();                            // the pieces of step 0
  read_b(0, bb[0]);

  bool first = true;
  for (; tile < ntiles; tile += G) {
    const bool fst = first;
    first = false;
    const long tnext = tile + G;
    const int y0n = tile_y0(tnext < ntiles ? tnext : tile);     // (a patch past the last tile is all zeros: normalising it is harmless)
    const int y0c = tile_y0(tile);

    f32x4_t acc[FP][JW];                                // (written, not accumulated into, by the MFMAs of the tile's first K-step)
    bf16x8_t a[FP];
    int z;                                              // an opaque 0, new per tile: without it the compiler computes every step's addresses once
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));          // per kernel and keeps them alive across the tile loop (spills)
    // byte address of fragment i's pixel for tap (0, 0) inside a chunk buffer: output pixel t = 16 i + frow of the tile sits at patch
    // pixel (t / W) * PW + t % W = t + (PW - W) (t / W); lanes past the tile (the last fragment) read the pixels behind it: whatever they
    // hold, those lanes are masked out of the BatchNorm sums and their stores fall outside the output descriptor
    // As registers: fragment i starts at tile pixel 16 i, in tile row i0 = 16 i / W; its lanes from frow = (i0 + 1) W - 16 i on sit one
    // row further down (at most one row boundary per fragment: W >= 14 and 16 i is even).  So the address is a per-lane value that depends
    // on that threshold only -- frow * 64 plus one row's pitch excess for the lanes behind it: one register per DISTINCT threshold
    // (three for W = 28, seven for W = 14) -- plus a compile-time constant per fragment that goes into the read's immediate offset.
    const int f64 = (frow + z) * CHB;
    int bthr[8];                                        // bthr[k]: threshold 2 (k + 1)   (only the entries a fragment uses stay alive)
#pragma unroll
    for (int k = 0; k < 8; ++k) bthr[k] = f64 + ((frow + z) >= 2 * (k + 1) ? (PW - W) * CHB : 0);
    auto base = [&](int i) -> int {                     // (i is a compile-time constant at every call site: the loops over i are unrolled)
      const int t0 = 16 * i, i0 = t0 / W, thr = (i0 + 1) * W - t0;
      const int imm = (t0 + (PW - W) * i0) * CHB;
      return thr >= 16 ? f64 + imm : bthr[thr / 2 - 1] + imm;
    };

    for (int cp = 0; cp < CF::NCH / 2; ++cp) {          // two chunks (18 K-steps, fully unrolled) per trip: ring slots and buffers are static
      const int soff_cp = cp * 128, soff_nx = (cp + 1 == CF::NCH / 2 ? 0 : cp + 1) * 128;
      int zc;                                           // (an opaque 0 per trip, as `z` per tile: the 18 steps' chunk positions are invariant over
      asm volatile("v_mov_b32 %0, 0" : "=v"(zc));       //  the trips and would be hoisted out of the loop -- 36 live registers, spills)

      auto kstep = [&](auto S18) {
        constexpr int s18 = decltype(S18)::value;       // 0 .. 17 inside the chunk pair
        constexpr int half = s18 / NKC, sl = s18 % NKC, tap = sl;
        constexpr int buf = half;
        const char* const pb = smem + buf * CF::PBUF;
        // fragment addresses of a tap: chunk position fgrp ^ key, key = ((t + W r + q) >> 1) & 3 (16 i = 0 mod 8: one key per lane and tap)
        if constexpr (s18 == 0) {                       // a trip's first step: nothing is read ahead across the loop's back edge (it would keep
          const int pos = ((fgrp ^ (((frow + zc) >> 1) & 3)) << 4);   // the fragment registers alive across it and across the epilogue: spills)
#pragma unroll
          for (int i = 0; i < FP; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(pb + base(i) + pos);
        }
        if constexpr (sl == NKC - 1) {
          // ---- the chunk barrier, ONE STEP BEFORE the chunk ends: my reads of this buffer have returned (the last tap's fragments were
          // read a step ago) and my normalised chunks of the other buffer are written; behind the barrier that holds for everybody, so
          // (1) this buffer may be refilled -- with the chunk after next --, and (2) the next chunk's first fragments can be read AHEAD,
          // behind this step's MFMAs, instead of in the open at the start of the next chunk.  Everybody's pieces of the next chunk have
          // landed: each wave's wait in step sl = 4 (for weight pieces issued behind them) covered its own.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          const int c2 = 2 * cp + half + 2;             // the chunk after next (NCH, NCH + 1 = the next tile's chunks 0, 1)
          if (c2 < CF::NCH) issue_patch(tile, c2, buf, true);
          else issue_patch(tnext, c2 - CF::NCH, buf, tnext < ntiles);
        }
        // the tap that follows: its fragments are read into a[i] right behind the MFMAs that consumed a[i] -- within the chunk from this
        // buffer, in the first chunk's last step from the OTHER buffer (the second chunk's tap 0)
        constexpr int ntap = (tap + 1) % NKC, ntr = ntap / 3, ntq = ntap % 3;
        const char* const pbn = smem + (sl + 1 < NKC ? buf : buf ^ 1) * CF::PBUF;
        const int npos = ((fgrp ^ (((frow + zc + W * ntr + ntq) >> 1) & 3)) << 4) + (ntr * PW + ntq) * CHB;
        constexpr bool ahead = s18 != 17;
        // where the JW weight pieces of step s + D and the NPW normalisations of the next chunk's patch go: spread over the pixel fragments
        constexpr int WGAP = FP / JW, NGAP = FP / 3;
#pragma unroll
        for (int i = 0; i < FP; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          if (s18 == 0 && cp == 0) {                    // (wave-uniform; only step 0 of a trip carries both forms)
#pragma unroll
            for (int j = 0; j < JW; ++j) ksmma0(acc[i][j], bb[s18 & 1][j], a[i]);
          } else {
#pragma unroll
            for (int j = 0; j < JW; ++j) ksmma(acc[i][j], bb[s18 & 1][j], a[i]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (i >= 1 && (i - 1) % WGAP == 0 && (i - 1) / WGAP < JW) {
            // one weight piece of step s + D (slot (s18 + D) % D = s18 % D: its fragments are in registers)
            constexpr int k18 = s18 + D;
            constexpr int kk = k18 % 18;
            issue_w1(s18 % D, (i - 1) / WGAP, (kk % NKC) * (2 * C) + (kk / NKC) * 64 + (k18 < 18 ? soff_cp : soff_nx));
          }
          if (i == 0) {
            // the wave waits for ITS weight pieces of step s + 1 and reads their fragments
            if constexpr (sl == NKC - 1) {
              kswait_vm<ks_younger<CF, true, false>()>();
            } else if constexpr (sl <= 4) {
              constexpr bool patch = sl <= 3;
              if (half == 0 && cp == 0) {               // (wave-uniform) a tile's first chunk
                if (fst) kswait_vm<ks_younger<CF, false, false>()>(); else kswait_vm<ks_younger<CF, patch, true>()>();
              } else {
                kswait_vm<ks_younger<CF, patch, false>()>();
              }
            } else {
              kswait_vm<ks_younger<CF, false, false>()>();
            }
            read_b((s18 + 1) % D, bb[(s18 + 1) & 1]);
          }
          if constexpr (ahead) a[i] = *reinterpret_cast<const bf16x8_t*>(pbn + base(i) + npos);
          // the next chunk's patch has landed (it is older than the weight pieces waited for in step sl = 4): its pieces are normalised
          // in steps 5, 6, 7 -- in front of the barrier of step 8 -- up to three per step, each in a fragment gap of its own
          if constexpr (IN && sl >= 5 && sl <= 7) {
            const int nchunk = 2 * cp + half + 1 == CF::NCH ? 0 : 2 * cp + half + 1;
            const int y0x = 2 * cp + half + 1 == CF::NCH ? y0n : y0c;
            if (i >= 2 && (i - 2) % NGAP == 0 && (i - 2) / NGAP < 3) {
              const int pi = (sl - 5) * 3 + (i - 2) / NGAP;
              if (pi < NPW) normalise_piece(pi, nchunk, buf ^ 1, y0x);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#define KSS(n) kstep(std::integral_constant<int, n>{});
      KSS(0) KSS(1) KSS(2) KSS(3) KSS(4) KSS(5) KSS(6) KSS(7) KSS(8) KSS(9) KSS(10) KSS(11) KSS(12) KSS(13) KSS(14) KSS(15) KSS(16) KSS(17)
#undef KSS
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: the tile's pixels are one contiguous run of the NHWC output (TR full image rows); this wave writes its 16 JW
    // channels of every pixel; pixels past the tile fall outside the descriptor
    const long b = (unsigned)tile / (unsigned)CF::TPI;
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + ((b * W + y0c) * (long)W) * C), 0,
                                                                           p.no_store ? 0 : CF::NPIX * C * 2, 0x00020000);
    // fragment i: accumulators -> (bias, statistics, ReLU) -> bf16 -> strip i & 1; its strip reads are issued BEFORE fragment i + 1 is
    // converted and written (other strip), its stores behind that (one wave per SIMD: nothing else would cover the LDS round trip)
    float s1[JW][4], s2[JW][4];
#pragma unroll
    for (int j = 0; j < JW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
    constexpr int CPR = CF::CPR;                        // 16-byte chunks per strip row (= per pixel of this wave's channels)
    auto stage_frag = [&](int i) {
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r];
          if constexpr (AFF) v[r] += bv[j][r];
          if constexpr (ST) {
            // (the last fragment's lanes past the tile hold whatever pixel 0's taps gave them: not part of the sums)
            const float vs = (CF::LIVE_LAST < 16 && i == FP - 1 && frow >= CF::LIVE_LAST) ? 0.f : v[r];
            s1[j][r] += vs; s2[j][r] = fmaf(vs, vs, s2[j][r]);
          }
          if constexpr (AFF) v[r] = p.relu ? fmaxf(v[r], 0.f) : v[r];
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm, like the normalisation's store: a visible LDS store would make hipcc drain the weight ring and the next patch;
        //  the strip is private to the wave and a wave's LDS operations execute in order, so the read needs no wait for the write)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + (i & 1) * CF::STRIP + frow * (CPR * 16) + (((j * 2 + (fgrp >> 1)) ^ (frow & (CPR - 1))) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
    };
    stage_frag(0);
#pragma unroll
    for (int i = 0; i < FP; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      u32x4_t val[CF::SPF];
#pragma unroll
      for (int h = 0; h < CF::SPF; ++h) {
        const int pxl = (h * 64 + lane) / CPR, cq = lane % CPR;
        val[h] = *reinterpret_cast<const u32x4_t*>(stg + (i & 1) * CF::STRIP + pxl * (CPR * 16) + ((cq ^ (pxl & (CPR - 1))) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < FP) stage_frag(i + 1);
      __builtin_amdgcn_sched_barrier(0);
      // (every store is ISSUED, statistics-only launches too -- their descriptor's range is empty --, so that the waits can count them)
#pragma unroll
      for (int h = 0; h < CF::SPF; ++h) {
        const int pxl = (h * 64 + lane) / CPR, cq = lane % CPR;
        __builtin_amdgcn_raw_buffer_store_b128(val[h], srd_o, (16 * i + pxl) * (C * 2) + wave * (32 * JW) + cq * 16, 0, 0);
      }
    }
    if constexpr (ST) {
#pragma unroll
      for (int j = 0; j < JW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t1 = ksrow16_sum(s1[j][r]), t2 = ksrow16_sum(s2[j][r]);
          if (frow == 0) {
            // (inline asm, like the strip stores: hipcc guards a visible LDS atomic with a wait for every vector-memory operation in flight)
            const unsigned at = (unsigned)(uintptr_t)(lstat + 16 * j + 4 * fgrp + r);
            asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:%3" ::"v"(at), "v"(t1), "v"(t2), "n"(16 * JW * 4) : "memory");
          }
        }
    }
  }
  kswait_vm<0>();
  if constexpr (ST) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* const row = p.stats + (long)blockIdx.x * (2 * C);
    if (lane < 16 * JW) {
      row[16 * JW * wave + lane] = lstat[lane];
      row[C + 16 * JW * wave + lane] = lstat[16 * JW + lane];
    }
  }
}

template <typename CF, bool AFF, bool ST, bool IN = false>
__global__ __launch_bounds__(256, 1) void conv3x3_slices_kernel(const KsArgs p) { ks_body<CF, AFF, ST, IN>(p); }
template <typename CF, bool AFF, bool ST, bool IN = false> struct KsTag {};

template <typename CF, bool AFF, bool ST, bool IN = false>
int ks_launch(const KsArgs& s, unsigned grid, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<KsTag<CF, AFF, ST, IN>>(reinterpret_cast<const void*>(&conv3x3_slices_kernel<CF, AFF, ST, IN>), CF::LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_slices_kernel<CF, AFF, ST, IN>), dim3(grid), dim3(256), CF::LDS, st, s);
  return SR_OK;
}

inline bool ks_env_off(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }
template <typename CF> inline bool ks_enabled();
template <> inline bool ks_enabled<KsL3>() { static const bool off = ks_env_off("SR_NO_C3_256"); return !off; }
template <> inline bool ks_enabled<KsL2>() { static const bool off = ks_env_off("SR_NO_C3_128S"); return !off; }

template <typename CF>
inline bool ks_serves(const sr_conv_args* a) {
  return ks_enabled<CF>() && !a->stem && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Cin == CF::C && a->Cout == CF::C && a->W == CF::W &&
         a->H == CF::W && !a->res && !a->escale && a->B > 0;
}
template <typename CF>
inline unsigned ks_grid(long ntiles) {
  const long cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}
template <typename CF>
inline bool ks_in_affine_ok(const sr_conv_args* a) {
  return ks_serves<CF>(a) && a->act == SR_ACT_NONE && !a->bias && a->stats != nullptr;
}

template <typename CF>
int ks_conv(const sr_conv_args* a, void* stream, int route) {
  if (!ks_serves<CF>(a) || (a->act != SR_ACT_NONE && a->act != SR_ACT_RELU)) return SR_ERR_UNSUPPORTED;
  if ((a->in_scale || a->in_shift) && (!a->in_scale || !a->in_shift || !ks_in_affine_ok<CF>(a))) return SR_ERR_UNSUPPORTED;
  KsArgs s;
  s.x = (const bf16_t*)a->x; s.w = (const bf16_t*)a->w; s.y = (bf16_t*)a->y; s.bias = a->bias; s.stats = a->stats;
  s.B = a->B; s.relu = a->act == SR_ACT_RELU; s.no_store = a->no_store;
  s.in_scale = a->in_scale; s.in_shift = a->in_shift;
  const long ntiles = (long)s.B * CF::TPI;
  if (ntiles > 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  SR_ROUTE(route);
  const unsigned grid = ks_grid<CF>(ntiles);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (a->in_scale) rc = ks_launch<CF, false, true, true>(s, grid, st);
  else {
    const bool aff = a->bias != nullptr || s.relu, stt = a->stats != nullptr;
    rc = aff ? (stt ? ks_launch<CF, true, true>(s, grid, st) : ks_launch<CF, true, false>(s, grid, st))
             : (stt ? ks_launch<CF, false, true>(s, grid, st) : ks_launch<CF, false, false>(s, grid, st));
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

}  // namespace

// Internal hand-over from sr_conv2d / sr_conv_stats_rows (gemm.hip): SR_ERR_UNSUPPORTED when the launch is not this layer shape
int srx_c3d256_rows(const sr_conv_args* a) {
  if (!ks_serves<KsL3>(a)) return SR_ERR_UNSUPPORTED;
  return (int)ks_grid<KsL3>((long)a->B * KsL3::TPI);
}
bool srx_c3d256_in_affine_ok(const sr_conv_args* a) { return ks_in_affine_ok<KsL3>(a); }
int srx_c3d256_conv(const sr_conv_args* a, void* stream) { return ks_conv<KsL3>(a, stream, SR_ROUTE_C3D256); }

int srx_c3d128s_rows(const sr_conv_args* a) {
  if (!ks_serves<KsL2>(a)) return SR_ERR_UNSUPPORTED;
  return (int)ks_grid<KsL2>((long)a->B * KsL2::TPI);
}
bool srx_c3d128s_in_affine_ok(const sr_conv_args* a) { return ks_in_affine_ok<KsL2>(a); }
int srx_c3d128s_conv(const sr_conv_args* a, void* stream) { return ks_conv<KsL2>(a, stream, SR_ROUTE_C3D128); }
