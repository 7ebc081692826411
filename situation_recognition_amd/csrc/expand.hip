// Output-heavy 1x1 convolution (bottleneck expansion / eval-mode folded expansion) for gfx950:
//
//     out[M,N] = relu?( (A[M,K] . W[N,K]^T) * escale[n] + bias[n] (+ res[M,N]) )        bf16 in / out, fp32 accumulate
//
// with K in {64,128,256} and N = a multiple of 256 (ResNet: N = 4K).  Per output element the kernel moves 2 B of output,
// 2 B of residual and 0.5 B of A through HBM against 2K FLOPs: it is HBM-bound (layer3: 4.9 GB of residual + output per launch),
// and the generic 256x256 kernel of gemm.hip runs its K loop (matrix cores busy, HBM idle) and its epilogue (HBM busy, matrix
// cores idle) one after the other: measured 1500 us = 600 (K loops) + 900 (the epilogue's memory floor).
//
// Here the two overlap inside one workgroup:
//   * tile 128 x 256, 8 waves as 2 x 4, each wave 64 x 64 = 4 x 4 fragments of v_mfma_f32_16x16x32_bf16: 64 accumulator
//     registers -- so TWO accumulator sets fit where the 256x256 tile had one;
//   * while the K loop of tile t accumulates into one set, the epilogue of tile t-1 drains the other, cut into 8 half-strips
//     (8 rows x 64 columns per wave) that are spread over the K-steps of tile t: fragments -> per-wave fp32 staging strip in
//     LDS (XOR swizzle) -> rows of 8 consecutive columns per lane -> scale/shift FMA, residual add, ReLU, one 16-byte store;
//   * a workgroup walks all N/256 column tiles of one 128-row block before it moves on (A re-reads hit L2, its stores cover whole
//     output rows within a few microseconds);
//   * ONE instruction stream for every tile, including a dummy epilogue in front of the first tile and a dummy K loop behind the
//     last one (their buffer descriptors have num_records = 0: loads return zeros, stores are dropped, but every operation is
//     issued), so the number of vector-memory operations between any two points of the stream is a compile-time constant and
//     every wait is a COUNTED vmcnt: LDS-DMA pieces, residual loads and output stores all retire in issue order (CDNA4) and
//     none of them ever drains the queue.
//
// LDS (N <= 1024): 5 ring slots x 24 KiB (A 128 rows + W 256 rows of 64 bytes per K-step) + 8 x 4 KiB staging + 8 KiB of
// escale/bias = exactly 160 KiB.  Row swizzle / fragment reads as in gemm.hip's v3 kernels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct ExpArgs {
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  const bf16_t* res; long ldres;
  bf16_t* out; long ldc;
  const float* escale; const float* bias;
  int M, N, K, relu;
  const float* in_scale; const float* in_shift;   // weight-stationary kernel only: A <- relu(A * in_scale[k] + in_shift[k]) on load
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {       // f(std::integral_constant<int, I>) for I .. N-1: indices usable as template arguments
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int N> __device__ __forceinline__ void xwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int XSLOT = 24576;      // one K-step: A 128 x 64 B, then W 256 x 64 B
constexpr int XSTG = 4096;        // per-wave staging strip: 16 rows x 64 columns fp32
constexpr int XOOB = (int)0x80000000;
constexpr int XNREC = 0x7ffff000;

// half-strips of the previous tile's epilogue that ride on K-step `u` of the current tile (8 per tile)
template <int NKT> constexpr int hs_count(int u) { return NKT == 16 ? (u & 1) : 8 / NKT; }
template <int NKT> constexpr int hs_first(int u) { return NKT == 16 ? u / 2 : u * (8 / NKT); }
// vector-memory operations issued after the LDS-DMA pieces of K-step s (issued D steps earlier) and before the wait for them
template <int NKT, int D, int OPS> constexpr int younger_than_dma(int s) {
  int n = (D - 1) * 3;
  for (int u = s - D; u < s; ++u) n += OPS * hs_count<NKT>(((u % NKT) + NKT) % NKT);
  return n;
}

#ifdef XSTAMPS
// Diagnostic build only (build.py --stamps -> libsrhip_stamps.so): per-wave sums of s_memtime deltas over the sections of a K-step.
//   0 wait vmcnt + barrier, 1 LDS-DMA issue, 2 fragment reads (+ wait), 3 MFMAs, 4 staging write/read (+ wait), 5 epilogue
//   arithmetic, 6 store + residual load issue, 7 K-steps
__device__ unsigned long long g_xstamps[256 * 8 * 8];
#define XSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#define XACC(k) do { unsigned long long t_; XSTAMP(t_); xs[k] += t_ - xt; xt = t_; } while (0)
#else
#define XACC(k) do {} while (0)
#endif

template <int NKT, int NSLOT, bool RES, bool RELU, bool PP>
__device__ __forceinline__ void expand_body(const ExpArgs& p) {
  constexpr int D = NSLOT - 1;
  static_assert(D >= 1 && D <= NKT, "the loader runs at most one tile ahead");
  constexpr int OPS = RES ? 2 : 1;                 // vector-memory operations per half-strip: one store (+ one residual load)
  constexpr int PERIOD = 3 * NKT + 8 * OPS;        // ... per tile
  static_assert(PERIOD - 2 < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stg_base = smem + NSLOT * XSLOT;
  float* const vec = reinterpret_cast<float*>(stg_base + 8 * XSTG);   // [2][N]: escale, bias

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int fsw = ((lane >> 4) ^ ((lane & 8) >> 2)) << 4;   // fragment read: byte offset of this lane's 16-byte k-chunk in a 64-B row

  const int gn = p.N >> 8;
  const int nblk = (p.M + 127) >> 7;
  const int G = gridDim.x;
  int vb = blockIdx.x;
  {
    const int xcd = vb & 7, q = G >> 3, r = G & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  }
  const int my_blocks = vb < nblk ? (nblk - vb + G - 1) / G : 0;
  const int tiles_mine = my_blocks * gn;

  // per-column vectors of the whole launch -> LDS (read back per tile with ds_read: no vector-memory operation in the stream)
  for (int i = threadIdx.x; i < p.N; i += 512) {
    vec[i] = p.escale ? p.escale[i] : 1.f;
    vec[p.N + i] = p.bias ? p.bias[i] : 0.f;
  }

  // ---------------- loader ----------------
  // piece = 16 rows x 64 B per wave-instruction (lane -> row l>>2, 16-byte chunk l&3 of the swizzled row image)
  const int prow = lane >> 2;
  const int ec = ((lane & 3) ^ ((lane >> 5) << 1)) * 8;                 // source element offset of the chunk this lane stores
  const int a_row = wave * 16 + prow;                                    // A piece `wave`: rows wave*16 .. +15 of the 128
  const int a_vo_ok = (int)((a_row * p.lda + ec) * 2);
  int w_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_vo[i] = (int)((((wave + i * 8) * 16 + prow) * p.ldw + ec) * 2);
  char* const a_dst = smem + wave * 1024;
  char* const w_dst0 = smem + 8192 + wave * 1024;
  char* const w_dst1 = smem + 8192 + (wave + 8) * 1024;

  struct Desc { __amdgpu_buffer_rsrc_t a, w; int a_vo; };
  auto make_desc = [&](int blk, int tn, bool valid) {
    Desc d;
    const long m0 = (long)blk * 128;
    d.a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * p.lda), 0, valid ? XNREC : 0, 0x00020000);
    d.w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long)tn * 256 * p.ldw), 0, valid ? XNREC : 0, 0x00020000);
    d.a_vo = (m0 + a_row < p.M) ? a_vo_ok : XOOB;
    return d;
  };
  auto issue = [&](const Desc& d, int kstep, int slot) {
    const int so = kstep * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.a, (__attribute__((address_space(3))) void*)(a_dst + slot * XSLOT), 16, d.a_vo, so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.w, (__attribute__((address_space(3))) void*)(w_dst0 + slot * XSLOT), 16, w_vo[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.w, (__attribute__((address_space(3))) void*)(w_dst1 + slot * XSLOT), 16, w_vo[1], so, 0, 0);
  };

  // ---------------- epilogue addressing (row layout: lane -> row lane/8 of a half-strip, columns (lane%8)*8 .. +7) ----------------
  const int rrow = lane >> 3, rq8 = lane & 7;
  const int o_vo = (int)((((wm * 64 + rrow) * p.ldc) + wn * 64 + rq8 * 8) * 2);
  const int r_vo = (int)((((wm * 64 + rrow) * p.ldres) + wn * 64 + rq8 * 8) * 2);
  const int o_step = (int)(8 * p.ldc * 2), r_step = (int)(8 * p.ldres * 2);      // bytes from one half-strip to the next
  char* const stg = stg_base + wave * XSTG;
  // staging strip: row r (256 B), 16-byte chunk c stored at chunk position c ^ r
  const int stg_w = frow * 256;                                  // + ((j*4 + fgrp) ^ frow) * 16
  struct OutDesc { __amdgpu_buffer_rsrc_t o, r; int n0; };
  auto make_out = [&](int blk, int tn, bool valid) {
    OutDesc d;
    const long m0 = (long)blk * 128;
    const long rows = valid ? ((long)p.M - m0 < 128 ? (long)p.M - m0 : 128) : 0;
    const int n0 = tn * 256;
    // a row at or beyond `rows` starts at or beyond num_records (ldc >= 256): its store is dropped, its residual reads as zero
    d.o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + m0 * p.ldc + n0), 0, rows > 0 ? (int)(((rows - 1) * p.ldc + 256) * 2) : 0, 0x00020000);
    d.r = __builtin_amdgcn_make_buffer_rsrc((void*)((RES ? p.res : p.out) + m0 * p.ldres + n0), 0,
                                            (RES && rows > 0) ? (int)(((rows - 1) * p.ldres + 256) * 2) : 0, 0x00020000);
    d.n0 = n0;
    return d;
  };

  f32x4_t acc[2][4][4];   // [set][n-fragment j][m-fragment i]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[s][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 rq[8];            // residual chunks of the tile whose K loop is running, one per half-strip
#pragma unroll
  for (int e = 0; e < 8; ++e) rq[e] = make_uint4(0, 0, 0, 0);

  const int a_off = (wm * 64 + frow) * 64 + fsw;
  const int b_off = 8192 + (wn * 64 + frow) * 64 + fsw;

  // tile ordinal q -> (block, column tile)
  int blk_c = vb, tn_c = 0;                 // tile whose K loop runs in the current iteration
  Desc ld_c = make_desc(blk_c, tn_c, tiles_mine > 0);
  OutDesc out_c = make_out(blk_c, tn_c, tiles_mine > 0);
  OutDesc out_p = make_out(0, 0, false);    // tile being drained: none yet
  // prologue: the first D K-steps of tile 0; drained completely once, so that the counted waits below start from a known state
#pragma unroll
  for (int s = 0; s < D; ++s) issue(ld_c, s, s);
  xwait_vm<0>();
  __syncthreads();
  // The two wave groups (wm = 0 / 1: one wave of each on every SIMD) run HALF A STEP apart: while one group is in its L section
  // (LDS-DMA issue, fragment reads, staging round trip) the other is in its M section (MFMAs, epilogue arithmetic, store), each
  // section closed by a barrier.  In lock-step all eight waves hit the LDS at once and then all hit the matrix pipe at once:
  // in-kernel stamps showed 700 cycles per step for the LDS burst alone and 2200-2800 cycles per step in all.
  //   DMA-after-read : L(s) refills the slot of K-step s-1, whose fragments both groups had in registers (lgkmcnt 0) before a
  //                    barrier both have passed (group 0: B1(s-1); group 1: its B1(s-1) = group 0's B2(s-1)).
  //   read-after-DMA : group 0 reads K-step s+1 in the section after its M(s), group 1 one section later; every wave has waited
  //                    for its own pieces of K-step s+1 before the barrier in front of group 0's read (group 0 at the end of
  //                    M(s), group 1 at the end of its L(s), the same time slot).
  // PP (SR_EXPAND_PP) is off by default: interleaved A/B runs at batch 6144 (3 x 8 launches each) gave, lock-step / ping-pong /
  // generic 256x256 kernel: K=256 1486 / 1585 / 1686 us, K=128 2063 / 2075 / 2426 us, K=64 4160 / 4250 / 4214 us -- with four
  // waves in every section the LDS burst halves, but the LDS-DMA issue backs up behind the other group's stores.
  if (PP && wm == 1) __builtin_amdgcn_s_barrier();

  int slot_c = 0, slot_i = D % NSLOT;       // ring slot read by the next K-step / filled by the next issue
#ifdef XSTAMPS
  unsigned long long xs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xt;
  XSTAMP(xt);
#endif

  auto body = [&](auto CURC, int it) {
    constexpr int CUR = decltype(CURC)::value, PRV = 1 - CUR;
    // descriptors of the next tile (the loader crosses into it D steps before this tile's K loop ends)
    int blk_n = blk_c, tn_n = tn_c + 1;
    if (tn_n == gn) { tn_n = 0; blk_n += G; }
    const bool valid_n = it + 1 < tiles_mine;
    const Desc ld_n = make_desc(blk_n, tn_n, valid_n);
    // per-column vectors of the tile being drained
    float esc[8], bia[8];
    {
      const float* e0 = vec + out_p.n0 + wn * 64 + rq8 * 8;
      const float4 a = *reinterpret_cast<const float4*>(e0), b = *reinterpret_cast<const float4*>(e0 + 4);
      const float4 c = *reinterpret_cast<const float4*>(e0 + p.N), d = *reinterpret_cast<const float4*>(e0 + p.N + 4);
      esc[0] = a.x; esc[1] = a.y; esc[2] = a.z; esc[3] = a.w; esc[4] = b.x; esc[5] = b.y; esc[6] = b.z; esc[7] = b.w;
      bia[0] = c.x; bia[1] = c.y; bia[2] = c.z; bia[3] = c.w; bia[4] = d.x; bia[5] = d.y; bia[6] = d.z; bia[7] = d.w;
    }
    static_for<0, NKT>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      if (!PP) {
        xwait_vm<younger_than_dma<NKT, D, OPS>(s)>();          // lock-step form: my pieces of K-step s have landed
        __builtin_amdgcn_s_barrier();                          // everybody's have; everybody has read K-step s-1
        asm volatile("" ::: "memory");
        XACC(0);
      }
      // ---- L section (ping-pong form: this wave group loads while the other one multiplies): LDS-DMA of K-step s+D, every LDS access of the step
      if (s + D < NKT) issue(ld_c, s + D, slot_i); else issue(ld_n, s + D - NKT, slot_i);
      slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
      XACC(1);
      // ---- every LDS operation of the step in one burst: the K-step's fragments, then the staging round trip of the FIRST
      //      half-strip that rides on it.  LDS operations of a wave complete in order, so the MFMAs below wait (counted
      //      lgkmcnt) for the fragments only, and the staging reads come back underneath them; measured in-kernel before this
      //      reordering: 770 cycles per step for the fragment reads and 400 more for a separate staging round trip.
      bf16x8_t fa[4], fb[4];
      const char* sl = smem + slot_c * XSLOT;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(sl + b_off + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sl + a_off + i * 1024);
      slot_c = slot_c + 1 == NSLOT ? 0 : slot_c + 1;
      constexpr int NH = hs_count<NKT>(s);
      f32x4_t x0, x1;
      auto stage = [&](auto EC) {                   // fragments of strip i -> staging (first half only), rows of half-strip e back
        constexpr int e = decltype(EC)::value, i = e >> 1, h = e & 1;
        if (h == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4_t*>(stg + stg_w + (((j * 4 + fgrp) ^ frow) << 4)) = acc[PRV][j][i];
        }
        const int r16 = h * 8 + rrow;
        x0 = *reinterpret_cast<const f32x4_t*>(stg + r16 * 256 + (((2 * rq8) ^ r16) << 4));
        x1 = *reinterpret_cast<const f32x4_t*>(stg + r16 * 256 + (((2 * rq8 + 1) ^ r16) << 4));
      };
      if constexpr (NH > 0) stage(std::integral_constant<int, hs_first<NKT>(s)>{});
      // group 1 publishes its pieces of K-step s+1 here (group 0 reads them in the next section), group 0 at the end of M
      if (PP) {
        if (wm == 1) xwait_vm<younger_than_dma<NKT, D, OPS>(s + 1) - OPS * NH>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // fragments and staged rows are in registers: the slot may be refilled
      }
      XACC(2);
#ifdef XSTAMPS
      xs[7] += 1;
#endif
      if (PP) {
        __builtin_amdgcn_s_barrier();                          // B1
        asm volatile("" ::: "memory");
        XACC(0);
      }
      // ---- M section: 16 MFMAs (at raised priority in the ping-pong form), then the arithmetic and the store of the staged half-strip
      if (PP) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[CUR][j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[CUR][j][i], 0, 0, 0);
      if (PP) __builtin_amdgcn_s_setprio(0);
      XACC(3);
      // ---- arithmetic + store of the half-strips that ride on this K-step (issued behind the MFMAs, executing beside them)
      static_for<0, NH>([&](auto KC) {
        constexpr int e = hs_first<NKT>(s) + decltype(KC)::value;
        if constexpr (decltype(KC)::value > 0) stage(std::integral_constant<int, e>{});
        XACC(4);
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = __builtin_fmaf(x0[c], esc[c], bia[c]);
          v[4 + c] = __builtin_fmaf(x1[c], esc[4 + c], bia[4 + c]);
        }
        if (RES) {
          xwait_vm<PERIOD - 2>();                              // the chunk requested one tile ago (everything older has retired)
          const unsigned* pr = reinterpret_cast<const unsigned*>(&rq[e]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[2 * c] += __uint_as_float(pr[c] << 16);
            v[2 * c + 1] += __uint_as_float(pr[c] & 0xffff0000u);
          }
        }
        if (RELU) {
#pragma unroll
          for (int c = 0; c < 8; ++c) asm("v_max_f32 %0, 0, %1" : "=v"(v[c]) : "v"(v[c]));
        }
        bf16_t pk[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pk[c] = (bf16_t)v[c];
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifdef XSTAMPS
        asm volatile("" :: "v"(*reinterpret_cast<const u32x4_t*>(pk)));
        XACC(5);
#endif
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(pk), out_p.o, o_vo, e * o_step, 0);
        if (RES) {
          const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(out_c.r, r_vo, e * r_step, 0);
          rq[e] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        XACC(6);
      });
      if (PP) {
        if (wm == 0) xwait_vm<younger_than_dma<NKT, D, OPS>(s + 1)>();
        __builtin_amdgcn_s_barrier();                          // B2
        asm volatile("" ::: "memory");
        XACC(0);
      }
    });
    // the drained set starts the next tile from zero
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[PRV][j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    out_p = out_c;
    blk_c = blk_n; tn_c = tn_n;
    ld_c = ld_n;
    out_c = make_out(blk_c, tn_c, valid_n);
  };

  // iteration `it`: K loop of tile it (dummy when it == tiles_mine), epilogue of tile it-1 (dummy when it == 0)
  for (int it = 0; it <= tiles_mine; it += 2) {
    body(std::integral_constant<int, 0>{}, it);
    if (it + 1 <= tiles_mine) body(std::integral_constant<int, 1>{}, it + 1);
  }
  if (PP && wm == 0) __builtin_amdgcn_s_barrier();   // (group 1 finishes its last section)
  xwait_vm<0>();
#ifdef XSTAMPS
  if (blockIdx.x < 256 && lane == 0) {
    unsigned long long* o = g_xstamps + (blockIdx.x * 8 + wave) * 8;
    for (int k = 0; k < 8; ++k) o[k] = xs[k];
  }
#endif
}

// Tried and removed: a variant with TWO workgroups per CU (one accumulator set, <= 128 registers, K loop and epilogue in plain
// order, the overlap left to the second workgroup).  End to end it was 0.7 % slower than this kernel (the residual prefetch
// shrank to two chunks per wave to stay under 128 registers) and its race screen was not clean (tools/race_expand.py).

// =====================================================================================================
// Weight-stationary form (K <= 256, N in {256, 512, 1024}).
//
// In-kernel stamps and SQ counters of the kernel above put it on the LDS: per K-step every wave re-reads 4 weight + 4 activation
// fragments for 16 MFMAs (64 KiB per step and CU), the ring takes 24 KiB of LDS-DMA per step (two thirds of it WEIGHTS that every
// tile fetches again), and the fp32 staging adds 32 KiB -- 30 LDS bytes per output element against a memory floor that leaves
// ~12 K cycles per 128 x 256 tile.  But the weights of a wave's 64 columns are only K/32 x 4 fragments = 128 registers at
// K = 256: here they are loaded ONCE per kernel and stay in registers.  A workgroup owns a fixed block of 64 x WN columns
// (8 waves x 64 = 512 columns; N = 1024 is two such blocks), walks 32-row tiles of it, and only the ACTIVATIONS go through
// LDS: one 16 KiB tile per barrier (not per K-step), read by all eight waves.  17 LDS bytes per output, no weight traffic,
// one barrier per tile, the whole K loop of a tile (64 MFMAs per wave) back to back.
// =====================================================================================================
template <int NKT, int WN, bool RES, bool RELU, bool INAFF = false>
__device__ __forceinline__ void ws_body(const ExpArgs& p) {
  constexpr int WM = 8 / WN;                     // wave rows: every wave row works on its own 32 rows of the tile
  constexpr int TM = 32 * WM;                    // rows per tile
  constexpr int KB = NKT * 64;                   // bytes of K per row
  constexpr int ASLOT = TM * KB;                 // one activation tile in LDS: [k-step][row][64 B]
  constexpr int NSLOT = ASLOT <= 8192 ? 8 : (ASLOT <= 16384 ? 6 : 4), D = NSLOT - 2;
  constexpr int PIECES = ASLOT / 1024, PPW = (PIECES + 7) / 8;     // LDS-DMA pieces per tile / per wave
  constexpr int OPS = PPW + 4 + (RES ? 4 : 0);   // vector-memory operations per tile and wave: pieces, 4 stores, 4 residual loads
  static_assert((D - 1) * OPS + OPS - PPW < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stg = smem + NSLOT * ASLOT + (threadIdx.x >> 6) * XSTG;
  float* const inaff = reinterpret_cast<float*>(smem + NSLOT * ASLOT + 8 * XSTG);     // INAFF: [2][K] scale | shift of the input channels

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wn = wave % WN, wmi = wave / WN;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int fsw = ((lane >> 4) ^ ((lane & 8) >> 2)) << 4;

  const int NWG = WN * 64;                       // columns per workgroup
  const int nh = p.N / NWG;                      // column blocks
  const int G = gridDim.x / nh;                  // row walkers per column block (the host makes the grid a multiple of nh)
  const int half = blockIdx.x % nh, walker = blockIdx.x / nh;
  const int ntile = (p.M + TM - 1) / TM;
  const int my_tiles = walker < ntile ? (ntile - walker + G - 1) / G : 0;
  const int n0 = half * NWG + wn * 64;           // this wave's first column

  // ---- weights -> registers: fragment (j, ks): lane holds W[n0 + j*16 + (lane & 15)][ks*32 + 8*(lane >> 4) .. +7]
  bf16x8_t wf[4][NKT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ks = 0; ks < NKT; ++ks)
      wf[j][ks] = *reinterpret_cast<const bf16x8_t*>(p.W + (long)(n0 + j * 16 + frow) * p.ldw + ks * 32 + fgrp * 8);
  // per-column vectors in row layout (lane -> columns n0 + (lane % 8) * 8 .. +7): fixed for the whole kernel
  const int rrow = lane >> 3, rq8 = lane & 7;
  float esc[8], bia[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    esc[c] = p.escale ? p.escale[n0 + rq8 * 8 + c] : 1.f;
    bia[c] = p.bias ? p.bias[n0 + rq8 * 8 + c] : 0.f;
  }

  // ---- activation loader: piece q = (k-step q / (TM/16), 16-row group q % (TM/16)); lane -> row l>>2, 16-byte chunk l&3
  const int prow = lane >> 2;
  const int ec = ((lane & 3) ^ ((lane >> 5) << 1)) * 8;
  int a_vo[PPW], a_row[PPW], a_dst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = wave + i * 8;
    const int ks = q / (TM / 16), rg = q % (TM / 16);
    a_row[i] = rg * 16 + prow;
    a_vo[i] = q < PIECES ? (int)((a_row[i] * p.lda + ks * 32 + ec) * 2) : XOOB;
    a_dst[i] = (q < PIECES ? q : 0) * 1024;
  }
  auto issue = [&](int t, int slot) {            // tile ordinal t of this walker
    const long m0 = ((long)walker + (long)t * G) * TM;
    const bool valid = t < my_tiles;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (valid ? m0 : 0) * p.lda), 0, valid ? XNREC : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int vo = (m0 + a_row[i] < p.M) ? a_vo[i] : XOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + slot * ASLOT + a_dst[i]), 16, vo, 0, 0, 0);
    }
  };

  const int o_vo = (int)(((wmi * 32 + rrow) * p.ldc + n0 + rq8 * 8) * 2);
  const int r_vo = (int)(((wmi * 32 + rrow) * p.ldres + n0 + rq8 * 8) * 2);
  const int o_step = (int)(8 * p.ldc * 2), r_step = (int)(8 * p.ldres * 2);
  auto out_srd = [&](int t, const bf16_t* base, long ld, bool on) {
    const long m0 = ((long)walker + (long)t * G) * TM;
    const long rows = (on && t < my_tiles) ? ((long)p.M - m0 < TM ? (long)p.M - m0 : TM) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (rows > 0 ? m0 : 0) * ld), 0, rows > 0 ? (int)(((rows - 1) * ld + p.N) * 2) : 0, 0x00020000);
  };

  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t rq[4];
  {
    const __amdgpu_buffer_rsrc_t r0 = out_srd(0, RES ? p.res : p.out, p.ldres, RES);
#pragma unroll
    for (int e = 0; e < 4; ++e) rq[e] = RES ? __builtin_amdgcn_raw_buffer_load_b128(r0, r_vo, e * r_step, 0) : u32x4_t{0, 0, 0, 0};
  }
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s, s);
  if (INAFF) {
    for (int k = threadIdx.x; k < p.K; k += 512) {       // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
      const int ch = (k & ~7) | sr_pair_order(k & 7);
      inaff[k] = p.in_scale[ch]; inaff[p.K + k] = p.in_shift[ch];
    }
  }
  xwait_vm<0>();
  __syncthreads();

  // INAFF: the activation tile holds the RAW output of the preceding convolution; every lane applies that convolution's BatchNorm
  // + ReLU to the 16-byte chunks it loaded itself (they hold the same 8 channels of every tile), in LDS, between its own wait
  // for them and the tile's barrier -- the normalised tensor is never written to memory (it was one more read + write of the
  // tensor in front of this kernel).  Rows past M are zero-filled and become relu(shift): their output rows are dropped anyway.
  auto in_affine = [&](int slot) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int q = wave + i * 8;
      if (q >= PIECES) continue;
      const int c8 = (q / (TM / 16)) * 32 + ec;               // first of the chunk's 8 channels
      char* const at = smem + slot * ASLOT + a_dst[i] + lane * 16;
      const sr_f32x4 s0 = *reinterpret_cast<const sr_f32x4*>(inaff + c8), s1 = *reinterpret_cast<const sr_f32x4*>(inaff + c8 + 4);
      const sr_f32x4 h0 = *reinterpret_cast<const sr_f32x4*>(inaff + p.K + c8), h1 = *reinterpret_cast<const sr_f32x4*>(inaff + p.K + c8 + 4);
      *reinterpret_cast<sr_u32x4*>(at) = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(at), s0, s1, h0, h1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  const int a_off = (wmi * 32 + frow) * 64 + fsw;          // + ks * TM * 64 + i * 1024
  int slot_c = 0, slot_i = D;
  for (int t = 0; t < my_tiles; ++t) {
    // my pieces of tile t: issued D tiles ago, right after that tile's barrier
    xwait_vm<(D - 1) * OPS + OPS - PPW>();
    if (INAFF) in_affine(slot_c);
    __builtin_amdgcn_s_barrier();                 // the tile is complete; every wave has finished reading tile t-1
    asm volatile("" ::: "memory");
    issue(t + D, slot_i);                         // (refills the slot of tile t-2: NSLOT = D + 2)
    slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
    const char* sl = smem + slot_c * ASLOT + a_off;
    slot_c = slot_c + 1 == NSLOT ? 0 : slot_c + 1;
    f32x4_t acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKT; ++ks) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(sl + ks * (TM * 64) + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], fa, acc[j][i], 0, 0, 0);
      }
    }
    // ---- epilogue: two 16-row strips = four half-strips
    const __amdgpu_buffer_rsrc_t so = out_srd(t, p.out, p.ldc, true);
    const __amdgpu_buffer_rsrc_t sr = out_srd(t + 1, RES ? p.res : p.out, p.ldres, RES);
    static_for<0, 4>([&](auto EC) {
      constexpr int e = decltype(EC)::value, i = e >> 1, h = e & 1;
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4_t*>(stg + frow * 256 + (((j * 4 + fgrp) ^ frow) << 4)) = acc[j][i];
      }
      const int r16 = h * 8 + rrow;
      const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(stg + r16 * 256 + (((2 * rq8) ^ r16) << 4));
      const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(stg + r16 * 256 + (((2 * rq8 + 1) ^ r16) << 4));
      float v[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[c] = __builtin_fmaf(x0[c], esc[c], bia[c]);
        v[4 + c] = __builtin_fmaf(x1[c], esc[4 + c], bia[4 + c]);
      }
      if (RES) {
        xwait_vm<OPS - 2>();                      // the chunk requested one tile ago
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[2 * c] += __uint_as_float(rq[e][c] << 16);
          v[2 * c + 1] += __uint_as_float(rq[e][c] & 0xffff0000u);
        }
      }
      if (RELU) {
#pragma unroll
        for (int c = 0; c < 8; ++c) asm("v_max_f32 %0, 0, %1" : "=v"(v[c]) : "v"(v[c]));
      }
      bf16_t pk[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) pk[c] = (bf16_t)v[c];
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(pk), so, o_vo, e * o_step, 0);
      if (RES) rq[e] = __builtin_amdgcn_raw_buffer_load_b128(sr, r_vo, e * r_step, 0);
    });
  }
  xwait_vm<0>();
}

template <int NKT, int WN, bool RES, bool RELU, bool INAFF = false>
__global__ __launch_bounds__(512, 2) void conv1x1_ws_kernel(const ExpArgs p) { ws_body<NKT, WN, RES, RELU, INAFF>(p); }

// (thin kernel around a __device__ body: with the generic lambdas inside the __global__ function itself hipcc's HOST pass
//  silently drops the kernel's launch stub and the library no longer links)
template <int NKT, int NSLOT, bool RES, bool RELU, bool PP>
__global__ __launch_bounds__(512, 2) void conv1x1_expand_kernel(const ExpArgs p) { expand_body<NKT, NSLOT, RES, RELU, PP>(p); }

inline bool expand_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_EXPAND"); return e && e[0] == '1'; }();
  return !off;
}

template <int NKT, int NSLOT, bool RES, bool RELU, bool PP> struct XTag {};
template <int NKT, int WN, bool RES, bool RELU, bool INAFF = false> struct WTag {};

inline bool ws_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_WS"); return e && e[0] == '1'; }();
  return !off;
}

template <int NKT, int WN, bool RES, bool RELU, bool INAFF = false>
int launch_ws_v(const ExpArgs& a, hipStream_t st) {
  constexpr int TM = 32 * (8 / WN), ASLOT = TM * NKT * 64, NSLOT = ASLOT <= 8192 ? 8 : (ASLOT <= 16384 ? 6 : 4);
  const size_t lds = (size_t)NSLOT * ASLOT + 8 * XSTG + (INAFF ? NKT * 32 * 8 : 0);
  const int nh = a.N / (WN * 64);
  const long ntile = ((long)a.M + TM - 1) / TM;
  long walkers = sr_num_cus() / nh;
  if (walkers > ntile) walkers = ntile;
  if (walkers < 1) walkers = 1;
  if (!sr_set_dynamic_lds_tagged<WTag<NKT, WN, RES, RELU, INAFF>>(reinterpret_cast<const void*>(&conv1x1_ws_kernel<NKT, WN, RES, RELU, INAFF>), (int)lds)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv1x1_ws_kernel<NKT, WN, RES, RELU, INAFF>), dim3((unsigned)(walkers * nh)), dim3(512), lds, st, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
template <int NKT, int WN>
int launch_ws(const ExpArgs& a, hipStream_t st) {
  if (a.in_scale) {                              // (train-mode expansion conv: identity + ReLU; the caller has checked that form)
    return launch_ws_v<NKT, WN, true, true, true>(a, st);
  }
  if (a.res) return a.relu ? launch_ws_v<NKT, WN, true, true>(a, st) : launch_ws_v<NKT, WN, true, false>(a, st);
  return a.relu ? launch_ws_v<NKT, WN, false, true>(a, st) : launch_ws_v<NKT, WN, false, false>(a, st);
}

// which K variants run the half-step ping-pong form: bit 0 K=64, bit 1 K=128, bit 2 K=256 (SR_EXPAND_PP overrides, for A/B runs)
inline int expand_pp_mask() {
  static const int m = [] { const char* e = getenv("SR_EXPAND_PP"); return e ? atoi(e) : 0; }();
  return m;
}
template <int NKT, int NSLOT, bool RES, bool RELU, bool PP>
int launch_expand_p(const ExpArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  if (!sr_set_dynamic_lds_tagged<XTag<NKT, NSLOT, RES, RELU, PP>>(reinterpret_cast<const void*>(&conv1x1_expand_kernel<NKT, NSLOT, RES, RELU, PP>), (int)lds))
    return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((conv1x1_expand_kernel<NKT, NSLOT, RES, RELU, PP>), dim3(grid), dim3(512), lds, st, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
template <int NKT, int NSLOT, bool RES, bool RELU>
int launch_expand_v(const ExpArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  const int bit = NKT == 2 ? 1 : (NKT == 4 ? 2 : 4);
  return (expand_pp_mask() & bit) ? launch_expand_p<NKT, NSLOT, RES, RELU, true>(a, grid, lds, st)
                                  : launch_expand_p<NKT, NSLOT, RES, RELU, false>(a, grid, lds, st);
}

template <int NKT, int NSLOT>
int launch_expand(const ExpArgs& a, hipStream_t st) {
  const size_t lds = (size_t)NSLOT * XSLOT + 8 * XSTG + (size_t)a.N * 8;
  const int nblk = (a.M + 127) / 128, cus = sr_num_cus();
  const unsigned grid = (unsigned)(nblk < cus ? nblk : cus);
  if (a.res) return a.relu ? launch_expand_v<NKT, NSLOT, true, true>(a, grid, lds, st) : launch_expand_v<NKT, NSLOT, true, false>(a, grid, lds, st);
  return a.relu ? launch_expand_v<NKT, NSLOT, false, true>(a, grid, lds, st) : launch_expand_v<NKT, NSLOT, false, false>(a, grid, lds, st);
}

}  // namespace

#ifdef XSTAMPS
extern "C" int srx_expand_stamps(unsigned long long* host_out) {
  if (hipDeviceSynchronize() != hipSuccess) return SR_ERR_LAUNCH;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xstamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? SR_OK : SR_ERR_LAUNCH;
}
#endif

// Internal (not part of include/srhip.h): sr_conv2d hands over the launches this kernel serves.  Returns SR_ERR_UNSUPPORTED
// when the shape is not one of them (the caller then uses the generic kernel).
// Does the weight-stationary kernel serve this launch WITH an input affine (a->in_scale / in_shift)?  (sr_conv_in_affine_supported)
bool srx_conv1x1_in_affine_ok(const sr_conv_args* a, long M) {
  return expand_enabled() && ws_enabled() && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0 && !a->stem && !a->stats && !a->no_store &&
         (a->Cout == 256 || a->Cout == 512 || a->Cout == 1024) && (a->Cin == 64 || a->Cin == 128 || a->Cin == 256) && M >= 128 * 256 &&
         M <= 0x7fffffffL && a->res != nullptr && a->act == SR_ACT_RELU;
}

int srx_conv1x1_expand(const sr_conv_args* a, long M, void* stream) {
  if (a->in_scale || a->in_shift) {
    if (!a->in_scale || !a->in_shift || !srx_conv1x1_in_affine_ok(a, M)) return SR_ERR_UNSUPPORTED;
  }
  if (!expand_enabled()) return SR_ERR_UNSUPPORTED;
  if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->stem || a->stats || a->no_store) return SR_ERR_UNSUPPORTED;
  if (a->Cout % 256 || a->Cout > 1024 || (a->Cin != 64 && a->Cin != 128 && a->Cin != 256)) return SR_ERR_UNSUPPORTED;
  if (M < 128 * 256 || M > 0x7fffffffL) return SR_ERR_UNSUPPORTED;      // (small launches: the generic kernel's narrow tiles fill the chip better)
  if ((long)128 * a->Cout * 2 >= 0x7fffffffL) return SR_ERR_UNSUPPORTED;
  ExpArgs x;
  x.A = (const bf16_t*)a->x; x.lda = a->Cin;
  x.W = (const bf16_t*)a->w; x.ldw = a->Cin;
  x.res = (const bf16_t*)a->res; x.ldres = a->Cout;
  x.out = (bf16_t*)a->y; x.ldc = a->Cout;
  x.escale = a->escale; x.bias = a->bias;
  x.M = (int)M; x.N = a->Cout; x.K = a->Cin; x.relu = a->act == SR_ACT_RELU;
  x.in_scale = a->in_scale; x.in_shift = a->in_shift;
  hipStream_t st = (hipStream_t)stream;
  if (ws_enabled() && (a->Cout == 256 || a->Cout == 512 || a->Cout == 1024)) {     // weight-stationary form
    if (a->Cout == 256) {
      switch (a->Cin) { case 64: return launch_ws<2, 4>(x, st); case 128: return launch_ws<4, 4>(x, st); default: return launch_ws<8, 4>(x, st); }
    }
    switch (a->Cin) { case 64: return launch_ws<2, 8>(x, st); case 128: return launch_ws<4, 8>(x, st); default: return launch_ws<8, 8>(x, st); }
  }
  switch (a->Cin) {
    case 64: return launch_expand<2, 3>(x, st);
    case 128: return launch_expand<4, 5>(x, st);
    default: return launch_expand<8, 5>(x, st);
  }
}
