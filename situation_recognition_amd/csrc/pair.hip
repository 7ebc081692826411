// A bottleneck's expansion conv FUSED with the next block's reduce conv, for gfx950 (train mode, bf16):
//
//     Z[M, 4C] = relu( (relu(X*in_scale + in_shift) . W3[4C, C]^T) * escale + eshift + R[M, 4C] )          block output (written)
//     Y[M, CR] = Z . W1[CR, 4C]^T                                                                          next block's conv1, RAW (written)
//     stats    = per-workgroup column sums / sums of squares of Y's fp32 accumulators                       (bn1 of the next block)
//
// torchvision chain replaced: conv3 -> bn3 -> (+identity) -> relu -> next.conv1 (call site: reference model.py:35); CR = C inside a layer, 2 C
// where the next block opens the next layer.  Unfused, the block output Z (4 units of M x C x 2 B) is written by the expansion conv and read
// back by the reduce conv: of the pair's 14 units of HBM traffic (X 1, R 4, Z 4 | Z 4, Y 1) the 4-unit re-read goes -- 8.63 -> 6.17 GB per
// layer3 pair at batch 6144 (PMC: 6.38 GB) -- and the reduce conv's MFMAs run in the same pass.
//
// Design.  Both GEMMs are split by ROWS over the four waves of a workgroup (one wave per SIMD, up to 512 registers): a wave owns RW row
// fragments (48 rows of a 192-row tile at C = 256, 64 of 256 at C = 128 / 64) for ALL columns, so the chain X -> Z -> Y never leaves the wave:
//   * MFMA operands are arranged so that every product comes out TRANSPOSED (weights as the A operand, activations as B): the
//     accumulator of v_mfma_f32_16x16x32_bf16 then holds, per lane, 4 consecutive COLUMNS of one tile row.  The expansion's weight
//     rows are permuted inside a 32-column group (pair_sigma) so that two accumulator fragments, rounded to bf16, are exactly the
//     16 bytes a lane needs (a) for a row-major global store of Z, (b) for the residual it adds, and (c) as the B operand of the
//     reduce GEMM's K-step (k = 8 (lane / 16) .. + 7: the natural operand layout).  Z goes from accumulators to operand registers
//     without touching LDS, and the same permutation on W1's rows makes Y's accumulators 16-byte row-major stores as well.
//   * X (the tile's rows x C channels per wave) is loaded ONCE per tile into registers in B-operand layout and normalised there
//     (BatchNorm + ReLU of the 3x3 in front, `sr_affine_relu_chunk`: the same function the unfused expansion kernel applies).
//   * Z's columns are walked in chunks of 64: E phase (K = C; at C = 256: 8 K-steps x 4 weight fragments x 3 row fragments = 96 MFMAs into
//     48 accumulator registers of the VECTOR file), epilogue (scale / shift, + identity, ReLU on the packed pair, bf16, store), R phase (K = 64; 2 K-steps x
//     CR / 16 weight fragments x RW MFMAs into Y's accumulators -- 192 AccVGPRs at C = CR = 256 -- which live across all chunks of the tile).
//   * Only the WEIGHTS go through LDS: both matrices, pre-packed in exactly the order the phases consume them (sr_conv_pair_pack: one
//     linear stream per tile -- 1 MiB at C = 256 --, a fragment = 1 KiB in lane order, so reads are conflict-free by construction and the
//     LDS-DMA copies 1 KiB runs), through a ring of four slots (one slot = one phase's fragments), filled by all four waves three phases
//     ahead.  One workgroup barrier per phase: behind it everybody's pieces of this phase have landed (own counted vmcnt in front) and
//     everybody has finished the previous phase's slot, which is refilled during this phase.  One ds_read_b128 per RW MFMAs.
//   * The identity is prefetched two chunks ahead into registers, X of the next tile behind the tile's last R phase.
// Where the memory instructions sit was measured, not assumed (profiles/r05/exp_pair/README.txt, same-box A/B of source variants): LDS-DMA
// pieces (L2 hits, no register data) are spread over the MFMAs of both phases (issued together behind the barrier they cost 13 %); Z's
// stores sit between the epilogue's vector instructions and the identity loads follow in a row (stores or loads between MFMAs: + 14 %;
// stores grouped, loads per unit, pieces in the epilogue, everything spread evenly: + 2 ... 8 %); a software-pipelined form with the
// epilogue's arithmetic between the R phase's MFMAs was 8 % slower.  At C = 128 / 64 the kernel runs at 5.0-5.4 TB/s (HBM-bound); at C = 256
// it is bound by instruction ISSUE at one wave per SIMD (matrix pipe 39 % busy, waves stalled at issue 49 % of their cycles, waiting 13 %).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// Diagnostic builds only (tools/pair_ablation.sh; results are garbage, the TIME of what is left is the measurement): bit 0 no MFMAs,
// bit 1 no HBM traffic (X / identity loads, Z / Y stores), bit 2 no weight stream (LDS-DMA), bit 3 no Z epilogue arithmetic, bit 4 no
// statistics reduction, bit 11 Z stores / identity loads as 8 rows x 128 B per instruction (misplaced data, same bytes).
#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct PairArgs {
  const bf16_t* x;          // [M, C]   raw output of the 3x3 (in_scale != null) or the normalised tensor
  const bf16_t* wpack;      // packed W3 | W1 stream (sr_conv_pair_pack)
  const bf16_t* res;        // [M, 4C]  identity
  bf16_t* z;                // [M, 4C]  block output
  bf16_t* y;                // [M, CR]  raw output of the next block's reduce conv
  const float* escale; const float* eshift;      // [4C] bn3 scale / shift
  const float* in_scale; const float* in_shift;  // [C] bn2 scale / shift or null
  float* stats;             // [grid][2][CR]   (train mode)
  const float* ybias;       // [CR] eval mode (EV): y = relu?(y + ybias), no statistics
  int yrelu;
  long M;
};

// C mid channels (K of the expansion), RW row fragments per wave, CR output channels of the reduce conv: C inside a layer, 2 C where the
// next block opens the next layer (its conv1 is 1x1 / stride 1 on the block output too, torchvision v1.5)
template <int C_, int RW_, int CR_ = C_>
struct PairCfg {
  static constexpr int C = C_, CX = 4 * C_, RW = RW_, CR = CR_;
  static constexpr int TM = 4 * RW * 16;                 // rows per tile
  static constexpr int NCH = CX / 64;                    // chunks of 64 Z columns
  static constexpr int KT = C / 32;                      // K-steps of the expansion
  static constexpr int QF = CR / 16;                     // column fragments of Y
  static constexpr int EF = KT * 4, RF = 2 * QF;         // weight fragments per E / R phase
  static constexpr int HALF = (EF > RF ? EF : RF) * 1024;   // a ring slot holds one phase's fragments (the larger of the two kinds)
  static constexpr int PWE = EF / 4, PWR = RF / 4;       // LDS-DMA pieces per wave of an E / R phase
  static constexpr int CHB = (EF + RF) * 1024;           // bytes of the weight stream per chunk: E phase, then R phase
  // LDS: the small tables FIRST (their reads then are one per-lane base register + an immediate offset: behind the 128 KiB ring the
  // offsets do not fit the instruction's 16 bits and hipcc keeps one address register per table access alive across the tile loop)
  static constexpr int TAB = 0, INAFF = TAB + 2 * CX * 4, STAT = INAFF + 2 * C * 4, RING = STAT + 4 * 2 * CR * 4, LDS = RING + 4 * HALF;
  static constexpr int ST = 2 * RW, LD = 2 * RW;         // Z stores / identity loads per chunk and wave
  static constexpr int YST = RW * (CR / 32);             // Y stores per tile and wave
  // vector-memory operations a wave has issued BEHIND the one it waits for (steady state; at a tile boundary there are more -- the next
  // tile's X loads, Y's stores -- and a smaller count only waits for more than it must).  The pieces of phase p + 3 go out during phase
  // p: an E phase issues the PWR pieces of an R phase, an R phase the PWE pieces of an E phase; the epilogue between them ST + LD.
  static constexpr int NE = PWR + ST + LD + PWE;         // E(c) start: its pieces went out during R(c - 2); E(c - 1), its epilogue, R(c - 1) since
  static constexpr int NR = ST + LD + PWE + PWR + ST + LD; // R(c) start: its pieces went out during E(c - 1); epilogue, R(c - 1), E(c), epilogue since
  static constexpr int NI = 2 * PWE + 2 * PWR + ST + LD; // epilogue: the identity loads issued at the end of the epilogue two chunks ago
  static constexpr int NA = YST;                         // tile start: X, loaded behind the last R phase, in front of Y's stores
  static_assert(EF % 4 == 0 && RF % 4 == 0 && NCH % 2 == 0 && EF % PWR == 0 && RF % PWE == 0 && LDS <= 160 * 1024 && NE < 64 && NR < 64 && NI < 64 && NA < 64,
                "pair kernel budget");
};
typedef PairCfg<256, 3> PairL3;        // layer3: 256 -> 1024 -> 256 on 14 x 14 images (192-row tiles)
typedef PairCfg<128, 4> PairL2;        // layer2: 128 ->  512 -> 128 on 28 x 28 images (256-row tiles; 128 + 64 accumulator registers)
typedef PairCfg<64, 4> PairL1;         // layer1:  64 ->  256 ->  64 on 56 x 56 images
typedef PairCfg<128, 3, 256> PairL23;  // layer2's last block -> layer3.0.conv1: 128 -> 512 -> 256 on 28 x 28 images
typedef PairCfg<64, 4, 128> PairL12;   // layer1's last block -> layer2.0.conv1:  64 -> 256 -> 128 on 56 x 56 images

#ifndef PAIR_NB
#define PAIR_NB 4
#endif
constexpr int NB = PAIR_NB;             // weight fragments read ahead of their MFMAs (4: measured against 2, 3, 6, 8)

// s_waitcnt vmcnt(N) as the BUILTIN (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt / lgkmcnt left at their maxima): hipcc's own
// wait insertion sees it and knows which of the loads it tracks (X, the identity) have landed.  Behind an inline-asm wait it cannot see
// it re-waits for those registers at their first use with counts derived from the shortest path through the loops -- vmcnt(15..30) in
// every chunk, which drains the weight ring and the identity prefetch the counted waits are there to keep in flight.
template <int N> __device__ __forceinline__ void prwait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

// accumulators pinned to AccVGPRs (see c3ds.hip: left alone hipcc renames MFMA destinations over the whole accumulator file)
__device__ __forceinline__ void prmma(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& b) {
#if PAIR_ABL & 1
  asm("; %0 %1 %2" : "+a"(acc) : "v"(w), "v"(b));
#else
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(b));
#endif
}
__device__ __forceinline__ void prmma0(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& b) {
#if PAIR_ABL & 1
  asm("; %0 %1 %2" : "=a"(acc) : "v"(w), "v"(b));
#else
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(b));
#endif
}

// the E phase's accumulators live in the VECTOR file (gfx90a+: an MFMA's C / D may be either file): the epilogue reads every element once,
// and with one wave per SIMD the 768 v_accvgpr_read per tile were plain issue time (round 5: 231 VGPR + 192 AccVGPR, - 1.3 % at layer3)
__device__ __forceinline__ void prmmav(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}
__device__ __forceinline__ void prmmav0(f32x4_t& acc, const bf16x8_t& w, const bf16x8_t& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(b));
}
// element i of an accumulator fragment as a vector register (explicit: asked for the values in plain C++, hipcc moves whole fragments
// from the accumulator file to the vector file THROUGH SCRATCH once both files are full)
__device__ __forceinline__ float pracc(const f32x4_t& acc, int i) {
  float x;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i]));
  return x;
}

__device__ __forceinline__ float prrow16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// Row of a weight matrix that sits at row m (0..15) of fragment j inside a group of 32 output columns starting at `base`: with the
// weights as the MFMA's A operand, lane (r, g) of the accumulator holds rows 4 g + i of the fragment, i.e. output columns
// base + 8 g + 4 (j & 1) + i: fragments j = 2 t and 2 t + 1 together give the lane columns base + 8 g .. + 7.
__host__ __device__ constexpr int pair_sigma(int j, int m) { return 32 * (j >> 1) + 8 * (m >> 2) + 4 * (j & 1) + (m & 3); }

// IN: X is the RAW output of the 3x3 in front, its BatchNorm + ReLU applied on load (train mode).  EV: eval mode -- Y gets its folded
// BatchNorm's bias (+ ReLU) instead of partial statistics.
template <typename CF, bool IN, bool EV>
__device__ __forceinline__ void pair_body(const PairArgs& p) {
  constexpr int C = CF::C, CX = CF::CX, CR = CF::CR, RW = CF::RW, TM = CF::TM, NCH = CF::NCH, KT = CF::KT, QF = CF::QF, HALF = CF::HALF, PWE = CF::PWE, PWR = CF::PWR;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // escale, eshift | in-affine | 4 statistics rows | weight ring
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const long ntiles = (p.M + TM - 1) / TM;
  const int G = gridDim.x;

  float* const tab = reinterpret_cast<float*>(smem + CF::TAB);
  float* const inaff = reinterpret_cast<float*>(smem + CF::INAFF);
  float* const lstat = reinterpret_cast<float*>(smem + CF::STAT) + wave * (2 * CR);
  for (int k = threadIdx.x; k < CX; k += 256) { tab[k] = p.escale[k]; tab[CX + k] = p.eshift[k]; }
  if (IN) {
    for (int k = threadIdx.x; k < C; k += 256) {          // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
      const int ch = (k & ~7) | sr_pair_order(k & 7);
      inaff[k] = p.in_scale[ch]; inaff[C + k] = p.in_shift[ch];
    }
  }
  for (int k = lane; k < 2 * CR; k += 64) lstat[k] = EV ? (wave == 0 && k < CR ? p.ybias[k] : 0.f) : 0.f;   // (EV: wave 0's row holds Y's bias)
  const float* const ybias = reinterpret_cast<const float*>(smem + CF::STAT);

  // ---- weight stream: per chunk EF + RF fragments of 1 KiB (E phase, then R phase), cyclic over the NCH chunks of a tile; of a phase
  // with NP pieces per wave, wave w copies pieces w NP .. w NP + NP - 1
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpack, 0, NCH * CF::CHB, 0x00020000);
  const int w_lane = lane * 16;
  // piece i of the E phase (KIND 0) / R phase (KIND 1) of chunk c (0 .. NCH-1) into ring slot `slot` (slot, kind, i: compile time at every call site)
  auto issue_w1 = [&](int slot, int kind, int c, int i) {
    if (PAIR_ABL & 4) return;
    const int np = kind ? PWR : PWE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(smem + CF::RING + slot * HALF + (wave * np + i) * 1024), 16, w_lane,
                                             c * CF::CHB + (kind ? CF::EF * 1024 : 0) + (wave * np + i) * 1024, 0, 0);
  };
  auto issue_w = [&](int slot, int kind, int c) {
    if (kind) {
#pragma unroll
      for (int i = 0; i < PWR; ++i) issue_w1(slot, 1, c, i);
    } else {
#pragma unroll
      for (int i = 0; i < PWE; ++i) issue_w1(slot, 0, c, i);
    }
  };

  // ---- per-tile descriptors of this wave's 16 RW rows (rows past M: outside the range -- loads give zeros, stores are dropped)
  auto wave_rows = [&](long tile) -> long {
    if (tile >= ntiles) return 0;
    const long left = p.M - (tile * TM + wave * (16 * RW));
    return left < 0 ? 0 : (left > 16 * RW ? 16 * RW : left);
  };
  auto srd_of = [&](const bf16_t* base, long tile, int ld) {
    const long rows = wave_rows(tile);
    const long m0 = rows > 0 ? tile * TM + wave * (16 * RW) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + m0 * ld), 0, (PAIR_ABL & 2) ? 0 : (int)(rows * ld * 2), 0x00020000);
  };
  const int xo = r * (C * 2) + g * 16;                     // + rho * 16 * C * 2 + kk * 64
#if PAIR_ABL & 2048
  // (diagnostic: every store / load instruction covers 8 rows x 128 contiguous bytes instead of 16 rows x 64 -- t selects the row half)
  const int zo = (r & 7) * (CX * 2) + (r >> 3) * 64 + g * 16;
#define PR_ZOFF(rho, c, t) ((rho) * (16 * CX * 2) + (t) * (8 * CX * 2) + (64 * (c)) * 2)
#else
  const int zo = r * (CX * 2) + g * 16;                    // + rho * 16 * CX * 2 + (64 c + 32 t) * 2
#define PR_ZOFF(rho, c, t) ((rho) * (16 * CX * 2) + (64 * (c) + 32 * (t)) * 2)
#endif

  u32x4_t a[RW][KT];                                       // X in B-operand layout: lane (r, g) <- row 16 rho + r, channels 32 kk + 8 g .. + 7
  auto load_a = [&](long tile) {
    const __amdgpu_buffer_rsrc_t s = srd_of(p.x, tile, C);
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
#pragma unroll
      for (int rho = 0; rho < RW; ++rho) a[rho][kk] = __builtin_amdgcn_raw_buffer_load_b128(s, xo, rho * (16 * C * 2) + kk * 64, 0);
  };
  u32x4_t idn[2][RW][2];                                   // identity of two chunks: lane <- row 16 rho + r, columns 64 c + 32 t + 8 g .. + 7
  auto load_idn = [&](auto BUF, const __amdgpu_buffer_rsrc_t s, int c) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int rho = 0; rho < RW; ++rho)
#pragma unroll
      for (int t = 0; t < 2; ++t) idn[buf][rho][t] = __builtin_amdgcn_raw_buffer_load_b128(s, zo, PR_ZOFF(rho, c, t), 0);
  };

  f32x4_t yacc[RW][QF];                                    // Y^T fragments: lane (r, g), register i <- row 16 rho + r, column 32 (q / 2) + 8 g + 4 (q % 2) + i

  long tile = blockIdx.x;
  issue_w(0, 0, 0);                                        // E(0), R(0), E(1): ring slot = tile phase & 3
  issue_w(1, 1, 0);
  issue_w(2, 0, 1);
  load_a(tile);
  {
    const __amdgpu_buffer_rsrc_t s = srd_of(p.res, tile, CX);
    load_idn(std::integral_constant<int, 0>{}, s, 0);
    load_idn(std::integral_constant<int, 1>{}, s, 1);
  }
  prwait_vm<0>();
  __syncthreads();

  for (; tile < ntiles; tile += G) {
    const long tnext = tile + G;
    const __amdgpu_buffer_rsrc_t srd_z = srd_of(p.z, tile, CX);
    const __amdgpu_buffer_rsrc_t srd_rc = srd_of(p.res, tile, CX), srd_rn = srd_of(p.res, tnext, CX);
    // X of this tile (loaded in the prologue / behind the previous tile's last epilogue), normalised in registers
    prwait_vm<CF::NA>();
    if (IN) {
      int z;                                               // an opaque 0, new per tile: without it the table addresses of all K-steps are
      asm volatile("v_mov_b32 %0, 0" : "=v"(z));           // computed once per kernel and kept alive across the tile loop (spills)
#pragma unroll
      for (int kk = 0; kk < KT; ++kk) {
        const int c8 = kk * 32 + (g + z) * 8;
        const sr_f32x4 s0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), s1 = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
        const sr_f32x4 h0 = *reinterpret_cast<const sr_f32x4*>(inaff + C + c8), h1 = *reinterpret_cast<const sr_f32x4*>(inaff + C + c8 + 4);
#pragma unroll
        for (int rho = 0; rho < RW; ++rho) a[rho][kk] = sr_affine_relu_chunk(a[rho][kk], s0, s1, h0, h1);
      }
    }

    auto chunk = [&](auto PARC, auto FIRSTC, const int c) {
      constexpr int PAR = decltype(PARC)::value;           // c & 1: ring slots 2 PAR (E) and 2 PAR + 1 (R), identity buffer PAR
      constexpr bool FIRST = decltype(FIRSTC)::value;      // chunk 0 of a tile: Y's accumulators are written, not added to, by its first K-step
      constexpr int SE = 2 * PAR, SR = 2 * PAR + 1;
      // ---------------- E phase: zacc[rho][j] = W3 fragment (kk, j) x X[rho][kk]
      prwait_vm<CF::NE>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // the pieces of phase p + 3 go into the slot everybody has just left, SPREAD over this phase's MFMAs (one per DGE / DGR fragments): the
      // four waves' pieces issued in a row right behind the barrier queue up in the CU's one address unit, and every wave sits in its
      // issue stall with an idle matrix pipe (measured: 345 of 1745 us)
      const int c_e = c + 1 >= NCH ? c + 1 - NCH : c + 1;  // this phase issues the pieces of R(c + 1) (tile phase p + 3)
      constexpr int DGE = CF::EF / PWR;
      f32x4_t zacc[RW][4];
      const char* const es = smem + CF::RING + SE * HALF + lane * 16;
      // (fragment f is read NB fragments ahead of its MFMAs into a rotating set of registers; the scheduling barriers keep hipcc from
      //  hoisting all 32 reads -- 128 registers -- to the top of the phase)
      bf16x8_t wf[NB];
#pragma unroll
      for (int f = 0; f < NB; ++f) wf[f] = *reinterpret_cast<const bf16x8_t*>(es + f * 1024);
#pragma unroll
      for (int f = 0; f < CF::EF; ++f) {
        const int kk = f / 4, j = f % 4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rho = 0; rho < RW; ++rho) {
          if (kk == 0) prmmav0(zacc[rho][j], wf[f % NB], __builtin_bit_cast(bf16x8_t, a[rho][kk]));
          else prmmav(zacc[rho][j], wf[f % NB], __builtin_bit_cast(bf16x8_t, a[rho][kk]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (f + NB < CF::EF) wf[f % NB] = *reinterpret_cast<const bf16x8_t*>(es + (f + NB) * 1024);
        if (f % DGE == DGE / 2) issue_w1((SE + 3) & 3, 1, c_e, f / DGE);
      }
      __builtin_amdgcn_sched_barrier(0);
      // (inline-asm MFMAs are invisible to the compiler's hazard recogniser: the accumulators are read by vector instructions below)
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      // ---------------- epilogue: scale / shift, + identity, ReLU, bf16; stored and kept as the R phase's B operand
      prwait_vm<CF::NI>();
      bf16x8_t zb[RW][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int col = 64 * c + 32 * t + 8 * g;
        const sr_f32x4 s0 = *reinterpret_cast<const sr_f32x4*>(tab + col), s1 = *reinterpret_cast<const sr_f32x4*>(tab + col + 4);
        const sr_f32x4 h0 = *reinterpret_cast<const sr_f32x4*>(tab + CX + col), h1 = *reinterpret_cast<const sr_f32x4*>(tab + CX + col + 4);
#pragma unroll
        for (int rho = 0; rho < RW; ++rho) {
          const u32x4_t iv = idn[PAR][rho][t];
          float v[8];
          if (PAIR_ABL & 8) {
            u32x4_t keep = a[rho][t] ^ iv;                 // (one read per accumulator fragment keeps the E phase alive)
            keep[0] ^= __float_as_uint(zacc[rho][2 * t][0]) ^ __float_as_uint(zacc[rho][2 * t + 1][0]);
            zb[rho][t] = __builtin_bit_cast(bf16x8_t, keep);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, zb[rho][t]), srd_z, zo, PR_ZOFF(rho, c, t), 0);
            continue;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = __builtin_fmaf(zacc[rho][2 * t][i], s0[i], h0[i]);
            v[4 + i] = __builtin_fmaf(zacc[rho][2 * t + 1][i], s1[i], h1[i]);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += __uint_as_float(iv[k] << 16);
            v[2 * k + 1] += __uint_as_float(iv[k] & 0xffff0000u);
          }
          sr_u32x4 pk;
#pragma unroll
          for (int k = 0; k < 4; ++k) {                    // round to bf16, then ReLU on the packed pair (a negative bf16 is a negative int16)
            pk[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(sr_f32x2{v[2 * k], v[2 * k + 1]}, sr_bf16x2));
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(pk[k]) : "v"(pk[k]));
          }
          zb[rho][t] = __builtin_bit_cast(bf16x8_t, pk);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, zb[rho][t]), srd_z, zo, PR_ZOFF(rho, c, t), 0);
        }
      }
      // identity of the chunk after next (the first two chunks of the next tile at a tile's end) into the buffer just consumed
      if (c + 2 < NCH) load_idn(PARC, srd_rc, c + 2);
      else load_idn(PARC, srd_rn, c + 2 - NCH);
      // ---------------- R phase: yacc[rho][q] += W1 fragment (t, q) x Z[rho][t]
      prwait_vm<CF::NR>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int c_r = c + 2 >= NCH ? c + 2 - NCH : c + 2;  // this phase issues the pieces of E(c + 2)
      constexpr int DGR = CF::RF / PWE;
      const char* const rs = smem + CF::RING + SR * HALF + lane * 16;
#pragma unroll
      for (int f = 0; f < NB; ++f) wf[f] = *reinterpret_cast<const bf16x8_t*>(rs + f * 1024);
#pragma unroll
      for (int f = 0; f < CF::RF; ++f) {
        const int t = f / QF, q = f % QF;
        __builtin_amdgcn_sched_barrier(0);
        if (FIRST && t == 0) {
#pragma unroll
          for (int rho = 0; rho < RW; ++rho) prmma0(yacc[rho][q], wf[f % NB], zb[rho][t]);
        } else {
#pragma unroll
          for (int rho = 0; rho < RW; ++rho) prmma(yacc[rho][q], wf[f % NB], zb[rho][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (f + NB < CF::RF) wf[f % NB] = *reinterpret_cast<const bf16x8_t*>(rs + (f + NB) * 1024);
        if (f % DGR == DGR / 2) issue_w1((SR + 3) & 3, 0, c_r, f / DGR);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    chunk(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    chunk(std::integral_constant<int, 1>{}, std::false_type{}, 1);
    for (int cp = 1; cp < NCH / 2; ++cp) {
      chunk(std::integral_constant<int, 0>{}, std::false_type{}, 2 * cp);
      chunk(std::integral_constant<int, 1>{}, std::false_type{}, 2 * cp + 1);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    // X is dead: the next tile's lands during Y's epilogue.  (Not inside the chunk loop, behind the last E phase: a load into `a` on the
    // loop's back edge makes hipcc wait for it in front of every chunk's first MFMAs.)
    load_a(tnext);

    // ---------------- Y epilogue: statistics from the fp32 accumulators (rows past M masked), bf16 stores
    const __amdgpu_buffer_rsrc_t srd_y = srd_of(p.y, tile, CR);
    const long row0 = tile * TM + wave * (16 * RW) + r;
    float mask[RW];
#pragma unroll
    for (int rho = 0; rho < RW; ++rho) mask[rho] = row0 + 16 * rho < p.M ? 1.f : 0.f;
    unsigned stat_at;                                      // this lane's statistics columns 8 g .. (an opaque copy per tile: see `z` above)
    asm volatile("v_mov_b32 %0, %1" : "=v"(stat_at) : "v"((unsigned)(uintptr_t)(lstat + 8 * g)));
    const int yo = r * (CR * 2) + g * 16;                  // + rho * 16 * CR * 2 + u * 64
#pragma unroll
    for (int u = 0; u < CR / 32; ++u) {
      float s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
      for (int rho = 0; rho < RW; ++rho) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = pracc(yacc[rho][2 * u], i); v[4 + i] = pracc(yacc[rho][2 * u + 1], i); }
        if constexpr (EV) {
          const sr_f32x4 b0 = *reinterpret_cast<const sr_f32x4*>(ybias + 32 * u + 8 * g), b1 = *reinterpret_cast<const sr_f32x4*>(ybias + 32 * u + 8 * g + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { v[i] += b0[i]; v[4 + i] += b1[i]; }
          if (p.yrelu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm("v_max_f32 %0, 0, %1" : "=v"(v[i]) : "v"(v[i]));
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float vm = v[i] * mask[rho];
            s1[i] += vm;
            s2[i] = __builtin_fmaf(vm, v[i], s2[i]);
          }
        }
        bf16_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = (bf16_t)v[i];
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(pk), srd_y, yo, rho * (16 * CR * 2) + u * 64, 0);
      }
      if (EV || (PAIR_ABL & 16)) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] = prrow16_sum(s1[i]); s2[i] = prrow16_sum(s2[i]); }
      if (r == 0) {
        // (inline asm: hipcc guards a visible LDS atomic with a wait for every vector-memory operation in flight; ONE address register
        //  and immediate offsets: with an address per sum hipcc keeps all 64 of them alive across the tile loop)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("ds_add_f32 %0, %1 offset:%3\n\tds_add_f32 %0, %2 offset:%4" ::"v"(stat_at), "v"(s1[i]), "v"(s2[i]), "n"((32 * u + i) * 4), "n"(CR * 4 + (32 * u + i) * 4) : "memory");
      }
    }
  }
  prwait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (!EV) {
    const float* const all = reinterpret_cast<const float*>(smem + CF::STAT);
    float* const row = p.stats + (long)blockIdx.x * (2 * CR);
    for (int k = threadIdx.x; k < 2 * CR; k += 256) row[k] = (all[k] + all[2 * CR + k]) + (all[4 * CR + k] + all[6 * CR + k]);
  }
}

template <typename CF, bool IN, bool EV>
__global__ __launch_bounds__(256, 1) void conv1x1_pair_kernel(const PairArgs p) { pair_body<CF, IN, EV>(p); }
template <typename CF, bool IN, bool EV> struct PairTag {};

// ---- weight stream: [chunk][EF + RF fragments][lane][8 bf16]: chunk c = the expansion's weights of Z columns 64 c .. + 63 (fragment kk * 4 + j),
// then K-slice c of the reduce conv (fragment t * QF + q)
template <typename CF>
__global__ void pair_pack_kernel(const bf16_t* __restrict__ w3, const bf16_t* __restrict__ w1, uint4* __restrict__ out) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece each
  constexpr int FPC = CF::EF + CF::RF;
  if (id >= CF::NCH * FPC * 64) return;
  const int lane = id & 63, f = (id >> 6) % FPC, c = id / (64 * FPC);
  const int m = lane & 15, g = lane >> 4;
  const bf16_t* src;
  if (f < CF::EF) {
    const int kk = f / 4, j = f % 4;
    src = w3 + (long)(64 * c + pair_sigma(j, m)) * CF::C + 32 * kk + 8 * g;
  } else {
    const int t = (f - CF::EF) / CF::QF, q = (f - CF::EF) % CF::QF;
    src = w1 + (long)pair_sigma(q, m) * CF::CX + 64 * c + 32 * t + 8 * g;
  }
  out[id] = *reinterpret_cast<const uint4*>(src);
}

inline bool pair_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_PAIR"); return e && e[0] == '1'; }();
  return !off;
}
// (Cmid, Cexp, Cred): inside a layer Cred = Cmid (256 / 128 / 64); across a layer boundary Cred = 2 Cmid (128 / 64)
inline bool pair_shape_ok(long M, int C, int CX, int CR) {
  return pair_enabled() && CX == 4 * C && M >= 1 && M <= 0x7fffffffL &&
         ((CR == C && (C == 256 || C == 128 || C == 64)) || (CR == 2 * C && (C == 128 || C == 64)));
}
template <typename CF>
inline unsigned pair_grid(long M) {
  const long ntiles = (M + CF::TM - 1) / CF::TM, cus = sr_num_cus();
  return (unsigned)(ntiles < cus ? ntiles : cus);
}
template <typename CF>
int pair_pack_launch(const void* w_exp, const void* w_red, void* out, hipStream_t st) {
  const int n = CF::NCH * (CF::EF + CF::RF) * 64;
  hipLaunchKernelGGL(pair_pack_kernel<CF>, dim3((n + 255) / 256), dim3(256), 0, st, (const bf16_t*)w_exp, (const bf16_t*)w_red, (uint4*)out);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
template <typename CF, bool IN, bool EV>
int pair_launch_v(const PairArgs& s, unsigned grid, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<PairTag<CF, IN, EV>>(reinterpret_cast<const void*>(&conv1x1_pair_kernel<CF, IN, EV>), CF::LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv1x1_pair_kernel<CF, IN, EV>), dim3(grid), dim3(256), CF::LDS, st, s);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
template <typename CF>
int pair_launch(const PairArgs& s, hipStream_t st) {
  const unsigned grid = pair_grid<CF>(s.M);
  if (s.ybias) return pair_launch_v<CF, false, true>(s, grid, st);        // eval mode: X is already normalised (the 3x3's own epilogue)
  return s.in_scale ? pair_launch_v<CF, true, false>(s, grid, st) : pair_launch_v<CF, false, false>(s, grid, st);
}
// one call per configuration: f(PairCfgTag<Cfg>{})
template <typename CF> struct PairCfgTag { typedef CF type; };
template <typename F>
inline auto pair_dispatch(int C, int CR, F&& f) {
  if (CR == C) return C == 256 ? f(PairCfgTag<PairL3>{}) : (C == 128 ? f(PairCfgTag<PairL2>{}) : f(PairCfgTag<PairL1>{}));
  return C == 128 ? f(PairCfgTag<PairL23>{}) : f(PairCfgTag<PairL12>{});
}

}  // namespace

extern "C" int sr_conv_pair_supported(int64_t M, int Cmid, int Cexp, int Cred, int dtype) {
  return dtype == SR_BF16 && pair_shape_ok(M, Cmid, Cexp, Cred) ? 1 : 0;
}
extern "C" int sr_conv_pair_pack_bytes(int Cmid, int Cexp, int Cred) {
  if (!pair_shape_ok(1, Cmid, Cexp, Cred)) return SR_ERR_UNSUPPORTED;
  return (Cmid + Cred) * Cexp * 2;
}
extern "C" int sr_conv_pair_stats_rows(int64_t M, int Cmid, int Cexp, int Cred) {
  if (!pair_shape_ok(M, Cmid, Cexp, Cred)) return SR_ERR_UNSUPPORTED;
  return (int)pair_dispatch(Cmid, Cred, [&](auto tag) { return pair_grid<typename decltype(tag)::type>(M); });
}
extern "C" int sr_conv_pair_pack(const void* w_exp, const void* w_red, void* out, int Cmid, int Cexp, int Cred, int dtype, void* stream) {
  if (!w_exp || !w_red || !out) return SR_ERR_ARG;
  if (dtype != SR_BF16) return SR_ERR_DTYPE;
  if (!pair_shape_ok(1, Cmid, Cexp, Cred)) return SR_ERR_UNSUPPORTED;
  if (((uintptr_t)w_exp | (uintptr_t)w_red | (uintptr_t)out) & 15) return SR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  return pair_dispatch(Cmid, Cred, [&](auto tag) { return pair_pack_launch<typename decltype(tag)::type>(w_exp, w_red, out, st); });
}
extern "C" int sr_conv_pair(const sr_pair_args* a, int dtype, void* stream) {
  if (!a || !a->x || !a->wpack || !a->res || !a->z || !a->y || !a->escale || !a->eshift || a->M <= 0) return SR_ERR_ARG;
  if ((a->stats == nullptr) == (a->ybias == nullptr)) return SR_ERR_ARG;        // train mode: statistics; eval mode: Y's bias
  if (dtype != SR_BF16) return SR_ERR_DTYPE;
  if ((a->in_scale == nullptr) != (a->in_shift == nullptr)) return SR_ERR_ARG;
  if (a->ybias && a->in_scale) return SR_ERR_UNSUPPORTED;
  if (((uintptr_t)a->x | (uintptr_t)a->wpack | (uintptr_t)a->res | (uintptr_t)a->z | (uintptr_t)a->y) & 15) return SR_ERR_ARG;
  const int cred = a->Cred ? a->Cred : a->Cmid;
  if (!pair_shape_ok(a->M, a->Cmid, a->Cexp, cred)) return SR_ERR_UNSUPPORTED;
  PairArgs s;
  s.x = (const bf16_t*)a->x; s.wpack = (const bf16_t*)a->wpack; s.res = (const bf16_t*)a->res; s.z = (bf16_t*)a->z; s.y = (bf16_t*)a->y;
  s.escale = a->escale; s.eshift = a->eshift; s.in_scale = a->in_scale; s.in_shift = a->in_shift; s.stats = a->stats; s.ybias = a->ybias; s.yrelu = a->yrelu; s.M = a->M;
  hipStream_t st = (hipStream_t)stream;
  return pair_dispatch(a->Cmid, cred, [&](auto tag) { return pair_launch<typename decltype(tag)::type>(s, st); });
}
