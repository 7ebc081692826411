// Output-heavy 1x1 convolution (bottleneck expansion / eval-mode folded expansion) for gfx950:
//
//     out[M,N] = relu?( (A[M,K] . W[N,K]^T) * escale[n] + bias[n] (+ res[M,N]) )        bf16 in / out, fp32 accumulate
//
// with K in {64,128,256} and N in {256,512,1024} (ResNet: N = 4K).  Per output element the kernel moves 2 B of output,
// 2 B of residual and 0.5 B of A through HBM against 2K FLOPs: it is HBM-bound (layer3: 4.9 GB of residual + output per launch),
// and the generic 256x256 kernel of gemm.hip runs its K loop (matrix cores busy, HBM idle) and its epilogue (HBM busy, matrix
// cores idle) one after the other: measured 1500 us = 600 (K loops) + 900 (the epilogue's memory floor).
//
// History: a first design -- 128 x 256 tiles with TWO accumulator sets, the previous tile's epilogue riding
// on the current tile's K-steps (`conv1x1_expand_kernel`, rounds 1-2) -- reached 1517 us on layer3 and was bound by its LDS traffic
// (every tile re-fetched the weights); it only ever served shapes no ResNet has (N not in {256, 512, 1024}) once the
// weight-stationary kernel below existed, and was removed in round 3.
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
  int plain_map;                                  // SR_WS_PLAIN_MAP=1 (A/B measurements): the round-3 block -> (column block, walker) map
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {       // f(std::integral_constant<int, I>) for I .. N-1: indices usable as template arguments
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int N> __device__ __forceinline__ void xwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int XSTG = 4096;        // per-wave staging strip: 16 rows x 64 columns fp32
constexpr int XOOB = (int)0x80000000;
constexpr int XNREC = 0x7ffff000;




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
  // Two column blocks (N = 1024: layer3's expansion, 36 launches per pass) walk the SAME rows, i.e. both read every activation tile.
  // With the plain map (half = block % 2) the two readers of a tile are neighbouring block ids, which the dispatcher deals to
  // different XCDs (round robin over 8): the tile crossed the fabric twice -- the family's measured 1.11x of its algorithmic bytes
  // (profiles/r03/conv_traffic_b6144.json).  Blocks b and b + 8 share an XCD, so within every group of 16 blocks the two halves of
  // walker 8 g + x are the blocks 16 g + x and 16 g + 8 + x: the second read of a tile is served by that XCD's L2 (both walk the same
  // tile sequence at the same pace, 16 KiB apart at most a few tiles).  Placement is a speed matter only: any map is correct.
  int half = blockIdx.x % nh, walker = blockIdx.x / nh;
  if (nh == 2 && (gridDim.x & 15) == 0 && !p.plain_map) {
    half = (blockIdx.x >> 3) & 1;
    walker = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
  }
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

inline bool expand_enabled() {
  static const bool off = [] { const char* e = getenv("SR_NO_EXPAND"); return e && e[0] == '1'; }();
  return !off;
}

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


}  // namespace


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
  static const int plain_map = [] { const char* e = getenv("SR_WS_PLAIN_MAP"); return e && e[0] == '1' ? 1 : 0; }();
  x.plain_map = plain_map;
  hipStream_t st = (hipStream_t)stream;
  if (ws_enabled() && (a->Cout == 256 || a->Cout == 512 || a->Cout == 1024)) {     // weight-stationary form
    SR_ROUTE(SR_ROUTE_WS);
    if (a->Cout == 256) {
      switch (a->Cin) { case 64: return launch_ws<2, 4>(x, st); case 128: return launch_ws<4, 4>(x, st); default: return launch_ws<8, 4>(x, st); }
    }
    switch (a->Cin) { case 64: return launch_ws<2, 8>(x, st); case 128: return launch_ws<4, 8>(x, st); default: return launch_ws<8, 8>(x, st); }
  }
  return SR_ERR_UNSUPPORTED;
}
