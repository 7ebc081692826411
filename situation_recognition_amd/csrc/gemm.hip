// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( sum_p A_p[M,K_p] . W_p[N,K_p]^T )
//
// One kernel family serves nn.Linear-shaped GEMMs (GGNN gates, classifiers, their backward GEMMs) and NHWC convolutions
// (the A rows are gathered on the fly: row m is output pixel (b,ho,wo), K runs over (tap, channel)).
//
//   v3 (default): persistent workgroups; every K-step is a 64-byte row slice in an LDS ring filled by LDS-DMA
//                 (`buffer_load_dwordx4 ... lds` with a scalar K offset; rows past M / N and padding taps carry an offset beyond
//                 num_records, so the load writes zeros -- no zero page, no branch); counted vmcnt, never 0 in the loop.
//                 256x256 tile: 8 waves, FIVE ring slots of 32 KiB (the whole 160 KiB; the epilogue's staging strips live in the
//                 slot the tile's last step was read from), four steps in flight, the two wave groups HALF A STEP apart
//                 (one loads while the other multiplies).  256x128 / 256x64 tiles: 4 waves, 3 slots, two workgroups per CU,
//                 lock-step.  Straight-line step: no rolling fragment prefetch (measured slower), epilogue kind a template
//                 parameter, 16-bit outputs leave through per-wave LDS staging strips as 16-byte coalesced stores.
//   v2 (fallback for N <= 128 with a non-linear epilogue, ragged 16-bit N, and SR_GEMM_NO_V3=1): 128-byte K-steps, 3-slot
//                 ring filled by LDS-DMA with per-lane pointers (padding taps and row tails read a zero page), 64x64 per wave.
//   Launches that other files serve better are handed over before dispatch: the output-heavy 1x1 convolutions (expand.hip) and the
//   7x7 stem (stem.hip); the fp8 3x3 convolutions have their own entry point (fp8.hip).
//
// Common to v2 and v3: MFMA roles are swapped (weights = MFMA "A", activations = MFMA "B") so each lane ends up with 4
// CONSECUTIVE output columns of one output row; the LDS image is lane-linear for the DMA and the bank-conflict swizzle is applied
// on the per-lane SOURCE address and again on the fragment ds_read_b128 (cdna_hip_programming.md rule 21); the
// workgroup -> tile mapping is XCD-contiguous.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

constexpr int SR_STATS_FLUSH = 32;   // tiles a workgroup accumulates BatchNorm partial sums over before reducing + storing them

struct ConvGeom {
  int on, H, Wd, Ho, Wo, stride, pad, KW, lgCseg, cpix;
};

struct KArgs {
  sr_kpair kp[3];
  int nk[3];  // K-tiles per pair
  int npairs, M, N, act;
  void* C; long ldc;
  void* C2;
  const float* bias; const float* bias2; float bias_scale;
  const void* res; long ldres;
  const void* aux1; const void* aux2;
  float* stats;
  ConvGeom cv;
  const float* escale;    // optional per-column multiplier applied to the accumulator before the bias (v3 kernels only)
  int no_store;           // statistics-only launch: the tile is not written (v3 kernels only)
  int stats_nflush;       // partial-statistics rows each workgroup writes (v3 kernels; see SR_STATS_FLUSH)
  const void* zero_page;  // 256 zero bytes (device address of g_zero_page, resolved once on the host)
  void* trash_page;       // sink for out-of-range lanes' stores
  int debug;  // SR_GEMM_DEBUG bits (diagnostics, tools/ only): v2: 1 = skip MFMA, 2 = skip loads after the prologue; v3: 4 = in-kernel
              // stamps (SR_STAMPS builds), 32 = lock-step loop instead of ping-pong
};

template <typename T> struct Frag;  // one 16-byte MFMA operand fragment
template <> struct Frag<bf16_t> { bf16x8_t v; };
template <> struct Frag<float> { f32x4_t v; };

template <typename T>
__device__ __forceinline__ void mma(const Frag<T>& w, const Frag<T>& a, f32x4_t& acc);
template <>
__device__ __forceinline__ void mma<bf16_t>(const Frag<bf16_t>& w, const Frag<bf16_t>& a, f32x4_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const Frag<float>& w, const Frag<float>& a, f32x4_t& acc) {
  // a 16-byte chunk holds k = 4g..4g+3 for lane group g; MFMA t consumes element t of every
  // group, i.e. the k-set {t, 4+t, 8+t, 12+t}: the same permutation on both operands.
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.v[t], a.v[t], acc, 0, 0, 0);
}

template <typename TO> __device__ __forceinline__ void load4(const TO* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename TO> __device__ __forceinline__ void store4(TO* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  bf16_t t[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(t);
}

// max(x, 0) as ONE instruction: fmaxf() makes hipcc canonicalise its operand first (a second v_max_f32 per element)
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// =====================================================================================================
// v2: persistent workgroups + 3-stage LDS-DMA ring (default).
//
// Each workgroup walks tiles  vb, vb+G, vb+2G, ...  (vb = XCD-contiguous virtual id) and treats all the K-tiles of
// all its output tiles as ONE stream of steps.  Step s lives in ring slot s % 3:
//
//     top of iteration s :  s_waitcnt vmcnt(N)   -- my DMAs of step s have landed (N = ops younger than them)
//                           s_barrier            -- everybody's have, and everybody finished computing step s-1
//                           issue DMAs of step s+2 into slot (s+2)%3 == (s-1)%3  (just freed)
//                           16 ds_read_b128 + 32 MFMA from slot s%3
//                           last K-tile of a tile: epilogue (stores), accumulators cleared
//
// so two steps of loads are always in flight across the barrier (counted vmcnt, never 0 in the loop;
// cdna_hip_programming.md "Pipelining across barriers") and the next tile's first K-tiles are fetched
// underneath the current tile's epilogue.  vmcnt also counts stores (CDNA4), so the epilogue issues a FIXED
// number of store instructions per lane (out-of-range lanes store to a trash page) and the wait after an
// epilogue allows for exactly those.
// =====================================================================================================
__device__ __attribute__((aligned(256))) unsigned char g_trash_page[64 * 16 * 2];

// In-kernel cycle stamps (diagnostic runs only: SR_GEMM_DEBUG bit 2 = value 4).  [block][wave][segment] sums of
// s_memtime deltas: 0 vmcnt wait, 1 barrier, 2 DMA issue, 3 MFMA + fragment reads, 4 epilogue, 5 steps.
__device__ unsigned long long g_stamps[256 * 8 * 8];
#define SR_STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")

// Lane id recomputed on the spot (2 VALU ops).  Used on the per-tile paths so that lane-constant values are not
// kept live (and spilled) across the K loop: a spill reload would bring a compiler-inserted vmcnt(0) that drains
// the LDS-DMA ring (cdna_hip_programming.md, attention pitfalls).
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15); every lane of the row receives the total.
// Four v_add_f32 with row_ror modifiers instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <typename T, typename TO, int WAVES_M, int WAVES_N, bool CONV, int EPI>
__device__ __forceinline__ void gemm_body_v2(const KArgs& p) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64, NW = WAVES_M * WAVES_N;
  constexpr int EPC = 16 / (int)sizeof(T), BK = 8 * EPC;
  constexpr int A_PER_WAVE = (BM / 8) / NW, B_PER_WAVE = (BN / 8) / NW, L = A_PER_WAVE + B_PER_WAVE;
  constexpr int STAGE = (BM + BN) * 128, NSTAGE = 3;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over waves");
  static_assert(L + 32 < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int frow = lane & 15, fgrp = lane >> 4;
  const int lrow = lane >> 3, csrc = (lane & 7) ^ lrow, ecol = csrc * EPC;

  const int gn = (p.N + BN - 1) / BN, gm = (p.M + BM - 1) / BM;
  const int ntiles = gm * gn, G = gridDim.x;
  int vb = blockIdx.x;
  {
    const int xcd = vb & 7, q = G >> 3, r = G & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  }
  if (vb >= ntiles) return;
  const int my_tiles = (ntiles - vb + G - 1) / G;
  int nkt = p.nk[0];
  if (p.npairs > 1) nkt += p.nk[1];
  if (p.npairs > 2) nkt += p.nk[2];
  const int total = my_tiles * nkt;

  // ---------------- loader state (for the tile whose K-tiles are being issued) ----------------
  int ld_tile = vb, ld_kt = 0;
  long a_row[A_PER_WAVE];
  unsigned a_mask[A_PER_WAVE];
  int w_row[B_PER_WAVE];
  auto setup_rows = [&](int tile) {
    const int tm = tile / gn, tn = tile - tm * gn;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      const long m = m0 + (wave + i * NW) * 8 + lrow;
      if (!CONV) {
        a_row[i] = m < p.M ? m : (long)p.M - 1;
        a_mask[i] = 0xffffffffu;
      } else if (m >= p.M) {
        a_row[i] = 0;
        a_mask[i] = 0;
      } else {
        const unsigned hw = (unsigned)(p.cv.Ho * p.cv.Wo), um = (unsigned)m;
        const long b = um / hw;
        const int rem = (int)(um - (unsigned)b * hw);
        const int ho = rem / p.cv.Wo, wo = rem - ho * p.cv.Wo;
        const int hi0 = ho * p.cv.stride - p.cv.pad, wi0 = wo * p.cv.stride - p.cv.pad;
        a_row[i] = ((b * p.cv.H + hi0) * (long)p.cv.Wd + wi0) * p.cv.cpix;
        const int ntap = p.kp[0].K >> p.cv.lgCseg;
        unsigned mk = 0;
        for (int t = 0; t < ntap; ++t) {
          const int dh = p.cv.KW == 1 ? t : (t * 11) >> 5, dw = t - dh * p.cv.KW;
          const int hi = hi0 + dh, wi = wi0 + dw;
          if (hi >= 0 && hi < p.cv.H && wi >= 0 && wi < p.cv.Wd) mk |= 1u << t;
        }
        a_mask[i] = mk;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      const int n = n0 + (wave + i * NW) * 8 + lrow;
      w_row[i] = n < p.N ? n : p.N - 1;
    }
  };
  setup_rows(ld_tile);

  auto issue = [&](int slot) {
    int pr = 0, kl = ld_kt;
    if (p.npairs > 1 && kl >= p.nk[0]) { kl -= p.nk[0]; pr = 1; }
    if (p.npairs > 2 && pr == 1 && kl >= p.nk[1]) { kl -= p.nk[1]; pr = 2; }
    const sr_kpair& kp = p.kp[pr];
    const int k = kl * BK + ecol;
    char* sA = smem + slot * STAGE;
    char* sB = sA + BM * 128;
    long tapoff = 0;
    int tap = 0;
    if (CONV) {
      tap = k >> p.cv.lgCseg;
      const int cc = k & ((1 << p.cv.lgCseg) - 1);
      const int dh = p.cv.KW == 1 ? tap : (tap * 11) >> 5, dw = tap - dh * p.cv.KW;
      tapoff = ((long)dh * p.cv.Wd + dw) * p.cv.cpix + cc;
    }
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      const T* src;
      if (!CONV) src = (const T*)kp.A + a_row[i] * kp.lda + k;
      else src = ((a_mask[i] >> tap) & 1) ? (const T*)kp.A + a_row[i] + tapoff : (const T*)p.zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sA + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      const T* src = (const T*)kp.W + (long)w_row[i] * kp.ldw + k;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (wave + i * NW) * 1024), 16, 0, 0);
    }
    if (++ld_kt == nkt) {
      ld_kt = 0;
      ld_tile += G;
      if (ld_tile < ntiles) setup_rows(ld_tile);
    }
  };

  f32x4_t acc[4][4];
  auto clear_acc = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  clear_acc();

  auto compute = [&](int slot) {
    const char* sA = smem + slot * STAGE + (wm * 64 + frow) * 128;
    const char* sB = smem + slot * STAGE + BM * 128 + (wn * 64 + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int sw = ((ks * 4 + fgrp) ^ (lane & 7)) << 4;
      Frag<T> a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const Frag<T>*>(sA + i * 16 * 128 + sw);
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const Frag<T>*>(sB + j * 16 * 128 + sw);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) mma<T>(w[j], a[i], acc[j][i]);
    }
  };

  constexpr bool two = (EPI == SR_ACT_SIGMOID_MUL || EPI == SR_ACT_TANH_BLEND);
  const bool relu = (p.act == SR_ACT_RELU);
  const int Nv = (p.N + 3) & ~3;   // columns that may be written (pad columns up to a multiple of 4 belong to the row)
  TO* const trash = reinterpret_cast<TO*>((char*)p.trash_page + lane * 16);
  const TO* const zeros = reinterpret_cast<const TO*>(p.zero_page);

  // Epilogue of one finished tile: exactly 16 (32 with a second output) store instructions per lane.
  auto epilogue = [&](int tile, int slot) {
    const int tm = tile / gn, tn = tile - tm * gn;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const bool want_stats = p.stats != nullptr;
    float s1[4][4], s2[4][4];
    if (want_stats) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fgrp * 4;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r < p.N ? n + r : p.N - 1;
        bv[r] = (p.bias ? p.bias_scale * p.bias[nn] : 0.f) + (p.bias2 ? p.bias2[nn] : 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long m = m0 + wm * 64 + i * 16 + frow;
        const bool ok = (m < p.M) && (n < Nv);
        float v[4], o2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] + bv[r];
        if (want_stats && ok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
        if (p.res) {
          float rv[4];
          load4<TO>(ok ? (const TO*)p.res + m * p.ldres + n : zeros, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        if constexpr (EPI == 0) {
          if (relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
        } else if constexpr (EPI == SR_ACT_SIGMOID) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
        } else if constexpr (EPI == SR_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
        } else if constexpr (EPI == SR_ACT_SIGMOID_MUL) {
          float h[4];
          load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = sigmoidf_(v[r]); o2[r] = v[r] * h[r]; }
        } else if constexpr (EPI == SR_ACT_TANH_BLEND) {
          float h[4], z[4];
          load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
          load4<TO>(ok ? (const TO*)p.aux2 + m * p.ldc + n : zeros, z);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float c = tanhf_(v[r]);
            o2[r] = c;
            v[r] = (1.f - z[r]) * h[r] + z[r] * c;
          }
        }
        store4<TO>(ok ? (TO*)p.C + m * p.ldc + n : trash, v);
        if (two) store4<TO>(ok ? (TO*)p.C2 + m * p.ldc + n : trash, o2);
      }
    }
    if (want_stats) {
      float* red = reinterpret_cast<float*>(smem + slot * STAGE);  // the slot just consumed: no DMA targets it
      lds_barrier();                                               // ... once every wave is done reading it
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s1[j][r], b = s2[j][r];
          a = row16_sum(a); b = row16_sum(b);
          if (frow == 0) {
            const int col = wn * 64 + j * 16 + fgrp * 4 + r;
            red[(0 * WAVES_M + wm) * BN + col] = a;
            red[(1 * WAVES_M + wm) * BN + col] = b;
          }
        }
      lds_barrier();
      for (int t = threadIdx.x; t < 2 * BN; t += NW * 64) {
        const int which = t / BN, col = t - which * BN;
        if (n0 + col < p.N) {
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES_M; ++w) s += red[(which * WAVES_M + w) * BN + col];
          p.stats[((long)tm * 2 + which) * p.N + n0 + col] = s;
        }
      }
    }
  };

  // ---------------- the stream ----------------
  int issued = 0;
  issue(0); ++issued;
  if (total > 1) { issue(1); ++issued; }
  int c_tile = vb, c_kt = 0;
  bool stored = false;
  for (int s = 0; s < total; ++s) {
    const int ahead = issued - s - 1;  // 0 or 1 later steps already in flight
    if (p.debug & 2) {
      wait_vm<0>();
    } else if (!stored) {
      if (ahead) wait_vm<L>(); else wait_vm<0>();
    } else if (!two) {
      if (ahead) wait_vm<L + 16>(); else wait_vm<16>();
    } else {
      if (ahead) wait_vm<L + 32>(); else wait_vm<32>();
    }
    stored = false;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < total) {
      if (!(p.debug & 2)) issue(issued % NSTAGE);
      ++issued;
    }
    const int slot = s % NSTAGE;
    if (!(p.debug & 1)) compute(slot);
    if (++c_kt == nkt) {
      epilogue(c_tile, slot);
      stored = true;
      c_kt = 0;
      c_tile += G;
      clear_acc();
    }
  }
}

template <typename T, typename TO, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_nt_v2_kernel(const KArgs p) {
  gemm_body_v2<T, TO, WAVES_M, WAVES_N, false, EPI>(p);
}
template <typename T, typename TO, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_igemm_v2_kernel(const KArgs p) {
  gemm_body_v2<T, TO, WAVES_M, WAVES_N, true, 0>(p);
}

// =====================================================================================================
// v3: 256x256 tile, 8 waves (2 x 4, each 128 x 64 = 8 x 4 MFMA fragments), for N > 128.
//
// Same persistent stream of K-steps as v2, but
//   * K-step = 64-byte rows (32 bf16 / 16 f32); the 256x256 ring has FIVE 32 KiB slots, FOUR steps (128 KiB) in flight per CU;
//     a 256x256 tile needs 1.5x fewer LDS-DMA bytes per FLOP than 256x128;
//   * every DMA piece is a `buffer_load_dwordx4 ... lds` with a scalar K offset (no vector address arithmetic per step);
//   * the K loop is straight-line: wait, barrier, 4 DMA pieces, 12 ds_read_b128, 32 MFMAs, and a scalar countdown to the
//     next tap / pair / tile change.  Measured against it on this kernel (tools/bench_gemm.py, same run):
//       - "rolling" fragment prefetch (step s+1's fragments read between the MFMAs of step s): 10-15 % slower on the convs:
//         the conditional reloads chop the MFMA stream into ~30 basic blocks per step;
//       - two wave groups half a step apart (one loads while the other multiplies, two barriers per step): equal to
//         rolling; its load section was dominated by scalar bookkeeping, not by LDS or DMA latency;
//       - in-kernel stamps + a skeleton of this loop (tools/ubench/dma_shapes.hip) put the step at ~1700 cycles for
//         1024 MFMA-pipe cycles; at that point the chip is power-limited on random operands (the clock drops from 2.3 to
//         ~2.0 GHz), i.e. ~1.2 PFLOP/s is the practical ceiling of this tile shape, not 2.5.
// 64-byte rows: 16-byte chunk c of row r is stored at chunk position c ^ ((r & 8) >> 2)  (conflict-free for the
// ds_read_b128 lane groups, checked by enumeration); one DMA piece = 16 rows x 64 B.
// =====================================================================================================
// WAVES_N = 4: 256x256 tile, 8 waves, 5-slot ring (160 KiB), one workgroup per CU   -- compute-heavy shapes
// WAVES_N = 2: 256x128 tile, 4 waves, 3-slot ring ( 72 KiB), TWO workgroups per CU  -- output-heavy shapes (small K, wide N):
//              one workgroup's epilogue (stores) overlaps the other's K loop instead of idling the matrix cores.
// EPI (compile time, keeps the fully unrolled epilogue small enough for the instruction cache):
//   0 linear (+bias, +residual, optional ReLU), 2 sigmoid, 3 tanh, 4 sigmoid & r*h, 5 tanh & GRU blend  (= SR_ACT_* codes; 1 = ReLU folds into 0)
// WAVES_N = 1: 256x64 tile, 4 waves stacked along M (each 64x64 = 4x4 fragments), 3-slot ring (60 KiB), two workgroups per CU -- N <= 64
// CFG = 8: 256x256 tile, FOUR waves (2x2, each 128x128 = 8x8 fragments), one wave per SIMD with the whole 512-register
//          file, 64 MFMAs per wave between barriers, 16 fragment reads per 64 MFMAs, 4-slot ring.
//   EPIX = 1: linear WITH BatchNorm partial statistics (32 more live registers for the running sums); EPIX = 0: linear
//   without them -- the registers go to a deeper residual prefetch instead.
template <typename T, typename TO, bool CONV, int CFG, int EPIX>
__device__ __forceinline__ void gemm_body_v3(const KArgs& p) {
  // EPIX: 0 linear, 1 linear + BatchNorm statistics, 6 linear + row residual (16-bit outputs), 7 RAW output + statistics (no bias,
  // multiplier, residual or activation: every train-mode convolution of the backbone), else the SR_ACT_* code of a fused epilogue
  constexpr int EPI = (EPIX == 1 || EPIX == 6 || EPIX == 7) ? 0 : EPIX;
  constexpr bool ST = EPIX == 1 || EPIX == 7;
  constexpr bool PLAIN = EPIX == 7;
  constexpr int WAVES_N = CFG == 8 ? 2 : CFG, FN = CFG == 8 ? 8 : 4;   // FN: 16-column fragments per wave along N
  constexpr int WAVES_M = CFG == 1 ? 4 : 2, FM = 16 / WAVES_M;          // FM: 16-row fragments per wave along M
  constexpr int BM = 256, BN = 16 * FN * WAVES_N, NW = WAVES_M * WAVES_N;
  constexpr int EPC = 16 / (int)sizeof(T), BK = 4 * EPC;
  // ring slots.  256x256: FIVE slots of 32 KiB = the whole 160 KiB; the epilogue's staging strips live in the slot the
  // tile's last step was read from (free until the next step's DMA, which waits behind a barrier after the epilogue).
  constexpr int NSLOT = CFG == 4 ? 5 : (CFG == 8 ? 4 : 3);
  constexpr int CPR = 2 * FN;          // 16-byte chunks per row of the per-wave output strip
  constexpr int RPI = 64 / CPR;        // strip rows moved by one wave-instruction
  constexpr int NH = 16 / RPI;         // instructions per 16-row strip
  constexpr int SLOT = (BM + BN) * 64, STG_OFF = CFG == 4 ? 0 : NSLOT * SLOT;
  int stg_off = STG_OFF;
  constexpr int A_PER = (BM / 16) / NW, B_PER = (BN / 16) / NW;
  constexpr int L = A_PER + B_PER;                      // DMA instructions per lane per step
  constexpr bool STAGED = sizeof(TO) == 2;              // 16-bit outputs leave through a per-wave LDS staging strip
  constexpr int S = STAGED ? NH * FM : (FN * FM <= 32 ? FN * FM : 0);   // store instructions per lane per epilogue (0: wait for them)
  static_assert(3 * L + S < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int fsw = ((lane >> 4) ^ ((lane & 8) >> 2)) << 4;         // fragment read: byte offset inside the 64-B row

  const int gn = (p.N + BN - 1) / BN, gm = (p.M + BM - 1) / BM;
  const int ntiles = gm * gn, G = gridDim.x;
  int vb = blockIdx.x;
  {
    const int xcd = vb & 7, q = G >> 3, r = G & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  }
  if (vb >= ntiles) return;
  const int my_tiles = (ntiles - vb + G - 1) / G;
  int nkt = p.nk[0];
  if (p.npairs > 1) nkt += p.nk[1];
  if (p.npairs > 2) nkt += p.nk[2];
  const int total = my_tiles * nkt;

  // ---------------- loader: per-lane base pointers, rebuilt only when the tile or the operand pair changes ----------------
  // Every piece is one `buffer_load_dwordx4 ... lds`: the tile's buffer descriptor (scalar), a loop-invariant 32-bit
  // per-lane row offset and a per-step SCALAR K offset.  No vector instruction computes an address inside the K loop
  // (an LDS-DMA with 64-bit per-lane pointers cost 6-8 VALU instructions per piece for the conv's tap offset and zero-page
  // select, issued in the matrix cores' shadow but competing for the same issue port).  Rows past M / N and padding taps
  // are given the offset 0x80000000, beyond num_records: the buffer load then writes zeros to LDS.
  int ld_tile = vb, ld_kt = 0, ld_pr = 0, ld_k0 = 0;   // ld_k0: first K-step of the current pair
  constexpr int OOB = (int)0x80000000;
  constexpr int NREC = 0x7ffff000;
  int a_vo[A_PER], w_vo[B_PER];
  unsigned a_mask[A_PER];
  __amdgpu_buffer_rsrc_t srd_a, srd_w;
  const bool taps = CONV && (p.kp[0].K >> p.cv.lgCseg) > 1;   // per-tap validity masks are needed
  auto setup_ptrs = [&](int tile, int pr) {
    const int tm = tile / gn, tn = tile - tm * gn;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const int fl = fresh_lane();
    const int prow = fl >> 2;
    const int ec = ((fl & 3) ^ ((fl >> 5) << 1)) * EPC;
    // field-wise selects: indexing the by-value argument struct with a run-time index makes hipcc copy it to scratch
    sr_kpair kp;
    kp.A = pr == 0 ? p.kp[0].A : (pr == 1 ? p.kp[1].A : p.kp[2].A);
    kp.W = pr == 0 ? p.kp[0].W : (pr == 1 ? p.kp[1].W : p.kp[2].W);
    kp.lda = pr == 0 ? p.kp[0].lda : (pr == 1 ? p.kp[1].lda : p.kp[2].lda);
    kp.ldw = pr == 0 ? p.kp[0].ldw : (pr == 1 ? p.kp[1].ldw : p.kp[2].ldw);
    kp.K = pr == 0 ? p.kp[0].K : (pr == 1 ? p.kp[1].K : p.kp[2].K);
    // input pixel (tap 0) of output row m, relative to the tensor start; increasing in m
    auto pixel = [&](long m) -> long {
      const unsigned hw = (unsigned)(p.cv.Ho * p.cv.Wo), um = (unsigned)m;
      const long b = um / hw;
      const int rem = (int)(um - (unsigned)b * hw);
      const int ho = rem / p.cv.Wo, wo = rem - ho * p.cv.Wo;
      return (b * p.cv.H + (ho * p.cv.stride - p.cv.pad)) * (long)p.cv.Wd + (wo * p.cv.stride - p.cv.pad);
    };
    const long pix0 = CONV ? pixel(m0) : 0;              // wave-uniform
    const long abase = CONV ? pix0 * p.cv.cpix : m0 * kp.lda;
    srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)kp.A + abase), 0, NREC, 0x00020000);
    srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)kp.W + (long)n0 * kp.ldw), 0, NREC, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int r = (wave + i * NW) * 16 + prow;
      const long m = m0 + r;
      if (m >= p.M) {
        a_vo[i] = OOB;
        a_mask[i] = 0;
      } else if (!CONV) {
        a_vo[i] = (int)((r * kp.lda + ec) * (long)sizeof(T));
        a_mask[i] = 0xffffffffu;
      } else {
        const unsigned hw = (unsigned)(p.cv.Ho * p.cv.Wo), um = (unsigned)m;
        const long b = um / hw;
        const int rem = (int)(um - (unsigned)b * hw);
        const int ho = rem / p.cv.Wo, wo = rem - ho * p.cv.Wo;
        const int hi0 = ho * p.cv.stride - p.cv.pad, wi0 = wo * p.cv.stride - p.cv.pad;
        const long pix = (b * p.cv.H + hi0) * (long)p.cv.Wd + wi0;
        a_vo[i] = (int)(((pix - pix0) * p.cv.cpix + ec) * (long)sizeof(T));
        const int ntap = kp.K >> p.cv.lgCseg;
        unsigned mk = 0;
        for (int t = 0; t < ntap; ++t) {
          const int dh = p.cv.KW == 1 ? t : (t * 11) >> 5, dw = t - dh * p.cv.KW;
          const int hi = hi0 + dh, wi = wi0 + dw;
          if (hi >= 0 && hi < p.cv.H && wi >= 0 && wi < p.cv.Wd) mk |= 1u << t;
        }
        a_mask[i] = mk;
        if (ntap == 1 && mk == 0) a_vo[i] = OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int r = (wave + i * NW) * 16 + prow;
      w_vo[i] = n0 + r < p.N ? (int)((r * kp.ldw + ec) * (long)sizeof(T)) : OOB;
    }
  };
  setup_ptrs(ld_tile, 0);

  // scalar (wave-uniform) description of the step being issued; computed once per step
  int st_aoff = 0, st_woff = 0;                // byte offsets (scalar)
  int st_tap = 0, st_slot = 0;
  auto issue_piece = [&](int q) {              // q < A_PER: A pieces; then B pieces
    char* dst = smem + st_slot * SLOT + (q >= A_PER ? BM * 64 + (wave + (q - A_PER) * NW) * 1024 : (wave + q * NW) * 1024);
    if (q < A_PER) {
      int vo = a_vo[q < A_PER ? q : 0];
      if (taps) vo = ((a_mask[q < A_PER ? q : 0] >> st_tap) & 1) ? vo : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (__attribute__((address_space(3))) void*)dst, 16, vo, st_aoff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)dst, 16, w_vo[q >= A_PER ? q - A_PER : 0],
                                               st_woff, 0, 0);
    }
  };
  f32x4_t acc[FN][FM];  // [n-fragment j][m-fragment i]
  auto clear_acc = [&]() {
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  clear_acc();

  const int a_off = (wm * FM * 16 + (lane & 15)) * 64 + fsw;
  const int b_off = BM * 64 + (wn * FN * 16 + (lane & 15)) * 64 + fsw;
  auto rdA = [&](int slot, int i) { return *reinterpret_cast<const Frag<T>*>(smem + slot * SLOT + a_off + i * 16 * 64); };
  auto rdB = [&](int slot, int j) { return *reinterpret_cast<const Frag<T>*>(smem + slot * SLOT + b_off + j * 16 * 64); };

  constexpr bool two = (EPI == SR_ACT_SIGMOID_MUL || EPI == SR_ACT_TANH_BLEND);
  const bool relu = (p.act == SR_ACT_RELU);
  const int Nv = (p.N + 3) & ~3;
  const TO* const zeros = reinterpret_cast<const TO*>(p.zero_page);

  // BatchNorm partial sums: every tile of this workgroup covers the same columns (the host makes the grid a multiple of
  // the column-tile count), so each lane keeps RUNNING per-column sums over its tiles and the 16-lane reduction + store
  // happens once per SR_STATS_FLUSH tiles, not once per tile (it was 4-5 K cycles of DPP chains per tile; the partial
  // buffer shrinks by the same factor: the stem's was 616 MB).  Row of flush f: ((vb / gn) * nflush + f) * WAVES_M + wm.
  constexpr bool STATS_OK = ST;
  constexpr int RSN = STATS_OK ? FN : 1;
  float rs1[RSN][4], rs2[RSN][4];
#pragma unroll
  for (int j = 0; j < RSN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs1[j][r] = rs2[j][r] = 0.f;
  int st_cnt = 0, st_flush = 0;
  auto stats_flush = [&](bool zero) {
    const int el = fresh_lane();
    const int frow = el & 15, fgrp = el >> 4;
    const int n0 = (vb % gn) * BN;
    float* row = p.stats + ((long)((vb / gn) * p.stats_nflush + st_flush) * WAVES_M + wm) * 2 * p.N;
#pragma unroll
    for (int j = 0; j < RSN; ++j) {
      const int nj = n0 + wn * FN * 16 + j * 16 + fgrp * 4;
      float s1[4], s2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[r] = zero ? 0.f : row16_sum(rs1[j][r]);
        s2[r] = zero ? 0.f : row16_sum(rs2[j][r]);
        rs1[j][r] = 0.f; rs2[j][r] = 0.f;
      }
      if (frow == 0) {
        if (nj + 3 < p.N && (p.N & 3) == 0) {
          *reinterpret_cast<float4*>(row + nj) = make_float4(s1[0], s1[1], s1[2], s1[3]);
          *reinterpret_cast<float4*>(row + p.N + nj) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nj + r < p.N) { row[nj + r] = s1[r]; row[p.N + nj + r] = s2[r]; }
        }
      }
    }
    ++st_flush;
    st_cnt = 0;
  };
#ifdef SR_STAMPS
  unsigned long long te_prep = 0, te_stats = 0, te_store = 0, te0 = 0, te1 = 0;
  const bool estamp = (p.debug & 4) != 0;
#endif
  auto epilogue = [&](int tile) {
#ifdef SR_STAMPS
    if (estamp) SR_STAMP(te0);
#endif
    const int tm = tile / gn, tn = tile - tm * gn;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const bool want_stats = STATS_OK && p.stats != nullptr;
    const int el = fresh_lane();
    const int frow = el & 15, fgrp = el >> 4;
    TO* const trash = reinterpret_cast<TO*>((char*)p.trash_page + el * 16);
    // per-column vectors: a lane's four columns of a fragment are consecutive, so interior tiles fetch them as one
    // 16-byte load per fragment column block (4 loads instead of 16 per vector, all issued before the first is used)
    const bool nvec = (n0 + BN <= p.N) && ((p.N & 3) == 0);
    float bv[FN][4];
    auto colvec = [&](const float* v, float (&out)[FN][4]) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nj = n0 + wn * FN * 16 + j * 16 + fgrp * 4;
        if (nvec) {
          const float4 q = *reinterpret_cast<const float4*>(v + nj);
          out[j][0] = q.x; out[j][1] = q.y; out[j][2] = q.z; out[j][3] = q.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[j][r] = v[nj + r < p.N ? nj + r : p.N - 1];
        }
      }
    };
    {
      float b1[FN][4], b2[FN][4];
      if (p.bias) colvec(p.bias, b1);
      if (p.bias2) colvec(p.bias2, b2);
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = (p.bias ? p.bias_scale * b1[j][r] : 0.f) + (p.bias2 ? b2[j][r] : 0.f);
    }
    // per-column multiplier: applied as one FMA with the bias in the store pass (a statistics pass, which needs the
    // scaled values themselves, folds it into the accumulators first)
    const bool scaled = p.escale != nullptr;
    float ev[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ev[j][r] = 1.f;
    if (scaled) {
      colvec(p.escale, ev);
      if (ST || want_stats) {       // (statistics kernels never keep the multipliers live: their registers hold the running sums)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < FM; ++i) acc[j][i][r] *= ev[j][r];
            ev[j][r] = 1.f;
          }
      }
    }
#ifdef SR_STAMPS
    if (estamp) { SR_STAMP(te1); te_prep += te1 - te0; te0 = te1; }
#endif
    if (want_stats) {
      // pass 1 (column-major over the fragments, 8 live sums): per-channel sum / sum of squares of acc + bias over this
      // wave's 128 rows -> its own partial row [2*tm + wm] of `stats` (no LDS, no barrier: the two wave groups run
      // half a step apart and must not meet at a barrier here)
      // Interior tiles without a bias (every train-mode convolution except at the edges) take a mask-free path on packed
      // f32 math (v_pk_add_f32 / v_pk_fma_f32: two columns per instruction) -- this pass is pure VALU time, per tile.
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const bool interior = (m0 + BM <= p.M) && (n0 + BN <= Nv) && !p.bias && !p.bias2;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float s1[4], s2[4];
        const int nj = n0 + wn * FN * 16 + j * 16 + fgrp * 4;
        if (interior) {
          f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f}, q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const f32x2_t v01 = {acc[j][i][0], acc[j][i][1]}, v23 = {acc[j][i][2], acc[j][i][3]};
            a01 += v01; a23 += v23;
            q01 = __builtin_elementwise_fma(v01, v01, q01);
            q23 = __builtin_elementwise_fma(v23, v23, q23);
          }
          s1[0] = a01[0]; s1[1] = a01[1]; s1[2] = a23[0]; s1[3] = a23[1];
          s2[0] = q01[0]; s2[1] = q01[1]; s2[2] = q23[0]; s2[3] = q23[1];
        } else {
          const bool nok = nj < Nv;
#pragma unroll
          for (int r = 0; r < 4; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const bool ok = nok && (m0 + wm * FM * 16 + i * 16 + frow < p.M);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = ok ? acc[j][i][r] + bv[j][r] : 0.f;
              s1[r] += v;
              s2[r] += v * v;
            }
          }
        }
        if constexpr (STATS_OK) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { rs1[j][r] += s1[r]; rs2[j][r] += s2[r]; }
        }
      }
      if constexpr (STATS_OK) {
        if (++st_cnt == SR_STATS_FLUSH || tile + G >= ntiles) stats_flush(false);
      }
    }
#ifdef SR_STAMPS
    if (estamp) { SR_STAMP(te1); te_stats += te1 - te0; te0 = te1; }
#endif
    // pass 2, i-major: one 16-row x 64-column strip of the wave's tile at a time
    if (!p.no_store) {
    // 16-bit outputs with a residual: the residual is NOT fetched in fragment layout (4 scattered columns per lane) but as
    // whole 16-byte row chunks, coalesced, one strip ahead, and added (+ReLU) after the LDS transpose, on the output rows.
    // A kernel of its own (EPIX 6; the host picks it whenever p.res is set and the output is 16-bit), not a run-time flag of the
    // linear kernel: with `rowres` tested at the fetch and again at the use, hipcc's wait-count pass (path-insensitive) took the
    // residual loads for possibly still pending on the path WITHOUT a residual and put `s_waitcnt vmcnt(0)` in front of the K
    // loop's fragment reads, which reuse those registers -- executed in every step, with or without a residual: the LDS-DMA ring
    // never held more than the step being waited for.  (Statistics kernels: a residual takes the fragment-layout path.)
    constexpr bool ROWRES = STAGED && EPIX == 6;
    constexpr bool rowres = ROWRES;
    // The residual is requested RD strips ahead (RD*NH 16-byte loads per lane in flight, in the registers the K loop's
    // fragments no longer need): one strip ahead left every strip waiting a full HBM latency for its residual; all FM
    // strips at once spills.
    constexpr int RD = FM < 3 ? FM : 3;   // (6 strips ahead in the kernels without statistics registers: 6 % slower)
    uint4 rall[RD][NH];
    auto fetch_res = [&](int i, uint4 (&dst)[NH]) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int r16 = h * RPI + el / CPR, c8 = (el % CPR) * 8;
        const long mm = m0 + wm * FM * 16 + i * 16 + r16;
        const int nn = n0 + wn * FN * 16 + c8;
        const bool okk = (mm < p.M) && (nn + 8 <= Nv);
        dst[h] = *reinterpret_cast<const uint4*>(okk ? (const char*)((const TO*)p.res + mm * p.ldres + nn) : (const char*)p.zero_page);
      }
    };
    if (rowres) {
#pragma unroll
      for (int i = 0; i < RD; ++i) fetch_res(i, rall[i]);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      uint4 rcur[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) rcur[h] = rowres ? rall[i % RD][h] : make_uint4(0, 0, 0, 0);
      if (rowres && i + RD < FM) fetch_res(i + RD, rall[i % RD]);
      const long m = m0 + wm * FM * 16 + i * 16 + frow;
      char* stg = smem + stg_off + wave * (16 * CPR * 16);  // per-wave [16 rows][16*FN cols] 16-bit strip; 16-B chunk c of row r at c ^ (r & (CPR-1))
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * FN * 16 + j * 16 + fgrp * 4;
        const bool ok = (m < p.M) && (n < Nv);
        float v[4], o2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ST ? acc[j][i][r] + bv[j][r] : __builtin_fmaf(acc[j][i][r], ev[j][r], bv[j][r]);
        if (p.res && !rowres) {
          float rv[4];
          load4<TO>(ok ? (const TO*)p.res + m * p.ldres + n : zeros, rv);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rv[r];
        }
        if constexpr (EPI == 0) {
          if (relu && !rowres) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = relu1(v[r]);
          }
        } else if constexpr (EPI == SR_ACT_SIGMOID) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
        } else if constexpr (EPI == SR_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
        } else if constexpr (EPI == SR_ACT_SIGMOID_MUL) {
          float h[4];
          load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = sigmoidf_(v[r]); o2[r] = v[r] * h[r]; }
        } else if constexpr (EPI == SR_ACT_TANH_BLEND) {
          float h[4], z[4];
          load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
          load4<TO>(ok ? (const TO*)p.aux2 + m * p.ldc + n : zeros, z);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float c = tanhf_(v[r]);
            o2[r] = c;
            v[r] = (1.f - z[r]) * h[r] + z[r] * c;
          }
        }
        if (STAGED && !two) {
          store4<TO>(reinterpret_cast<TO*>(stg + frow * (CPR * 16) + (((j * 2 + (fgrp >> 1)) ^ (frow & (CPR - 1))) << 4) + (fgrp & 1) * 8), v);
        } else {
          store4<TO>(ok ? (TO*)p.C + m * p.ldc + n : trash, v);
          if (two) store4<TO>(ok ? (TO*)p.C2 + m * p.ldc + n : trash, o2);
        }
      }
      if (STAGED && !two) {
        // the strip is wave-private and a wave's LDS instructions execute in order: no wait between the fragment writes
        // and the row reads (nor before the next strip's writes) -- only the compiler's own wait for the read data.
        // Every lane moves 16 contiguous bytes (8 lanes = one 128-byte row segment) -> 2 store instructions per strip,
        // full-line coalescing
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int r16 = h * RPI + el / CPR, c8 = (el % CPR) * 8;
          const long mm = m0 + wm * FM * 16 + i * 16 + r16;
          const int nn = n0 + wn * FN * 16 + c8;
          uint4 val = *reinterpret_cast<const uint4*>(stg + r16 * (CPR * 16) + ((((el % CPR)) ^ (r16 & (CPR - 1))) << 4));
          if constexpr (ROWRES) {
            if (rowres) {
              const unsigned* pv_ = reinterpret_cast<const unsigned*>(&val);
              const unsigned* pr_ = reinterpret_cast<const unsigned*>(&rcur[h]);
              unsigned ow[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float lo = __uint_as_float(pv_[q] << 16) + __uint_as_float(pr_[q] << 16);
                float hi = __uint_as_float(pv_[q] & 0xffff0000u) + __uint_as_float(pr_[q] & 0xffff0000u);
                if (relu) { lo = relu1(lo); hi = relu1(hi); }
                bf16_t pk[2] = {(bf16_t)lo, (bf16_t)hi};
                ow[q] = *reinterpret_cast<const unsigned*>(pk);
              }
              val = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
          }
          // (16-bit outputs whose N is not a multiple of 8 never reach this kernel -- `launch` sends them to the v2 kernels --
          //  so a 16-byte chunk is either wholly inside the matrix or wholly outside: an element-wise ragged-edge path here
          //  put ~70 predicated instructions into every strip of every tile)
          const bool okk = (mm < p.M) && (nn + 8 <= Nv);
          *reinterpret_cast<uint4*>(okk ? (char*)((TO*)p.C + mm * p.ldc + nn) : (char*)p.trash_page + el * 16) = val;   // (nontemporal: no difference)
        }
      }
    }
    }
#ifdef SR_STAMPS
    if (estamp) { SR_STAMP(te1); te_store += te1 - te0; }
#endif
    // Every register a global load of this epilogue targeted is USED here, on all paths: a per-column vector loaded under one
    // condition and consumed under another (a `no_store` launch never reads its multipliers) stays "possibly pending" for hipcc's
    // path-insensitive wait-count pass, which then guards the K loop's fragment reads (same registers) with `s_waitcnt vmcnt(0)`
    // in every step.  Where the values were consumed this costs nothing; elsewhere it is one wait per tile instead of one per step.
    if constexpr (!ST) {
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(ev[j][r]), "v"(bv[j][r]));
    }
  };

  // ---------------- EPIX 7: raw output + statistics, one pass over the accumulators ----------------
  // The general epilogue above walks the accumulators twice (statistics column-major, then the strips) and its strip loop is
  // cut into basic blocks by the run-time options (fragment-layout residual, ReLU, edge masks), so every strip's LDS round trip
  // -- 4 fragment writes, 2 row reads, the wait for them -- lies bare: in-kernel stamps gave 2.0 K cycles for the statistics and
  // 6.8 K for the strips of a 256x256 tile (11.6 K with the rest; a K = 1024 tile's loop is 53 K).  Without options the pass is
  // straight-line: fragment (i, j) is read from the accumulators ONCE, added to the lane's running sums (rows past M and columns past
  // N are exact zeros -- their operand rows were zero-filled by the loader and there is no bias -- so no masks), converted and
  // written to strip i & 1; the row reads of strip i are issued BEFORE strip i + 1 is converted and its stores after, so the LDS
  // latency hides under the next strip's arithmetic (256x256 tiles: two strips per wave fill the 32 KiB slot; the narrow tiles'
  // staging area holds one strip per wave: same order, no overlap).
  auto epilogue_plain = [&](int tile) {
   if constexpr (PLAIN) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int NSTG = CFG == 4 ? 2 : 1;
    static_assert(!(CFG == 4) || NW * NSTG * 16 * CPR * 16 <= SLOT, "two strips per wave fit the ring slot");
    const int tm = tile / gn, tn = tile - tm * gn;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const int el = fresh_lane();
    const int frow = el & 15, fgrp = el >> 4;
    char* const stg0 = smem + stg_off + wave * (NSTG * 16 * CPR * 16);
    const bool store = !p.no_store;
    auto put = [&](int i) {
      char* const stg = stg0 + (i % NSTG) * (16 * CPR * 16);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x2_t v01 = {acc[j][i][0], acc[j][i][1]}, v23 = {acc[j][i][2], acc[j][i][3]};
        f32x2_t a01 = {rs1[j][0], rs1[j][1]}, a23 = {rs1[j][2], rs1[j][3]}, q01 = {rs2[j][0], rs2[j][1]}, q23 = {rs2[j][2], rs2[j][3]};
        a01 += v01; a23 += v23;
        q01 = __builtin_elementwise_fma(v01, v01, q01);
        q23 = __builtin_elementwise_fma(v23, v23, q23);
        rs1[j][0] = a01[0]; rs1[j][1] = a01[1]; rs1[j][2] = a23[0]; rs1[j][3] = a23[1];
        rs2[j][0] = q01[0]; rs2[j][1] = q01[1]; rs2[j][2] = q23[0]; rs2[j][3] = q23[1];
        if (store) {
          const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
          if constexpr (STAGED) {
            // (inline asm: in this straight-line form hipcc puts `s_waitcnt vmcnt(0)` in front of a visible LDS store -- the
            //  next tile's LDS-DMA may alias -- and drains the ring; the strip is wave-private and a wave's LDS operations are in order)
            bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + frow * (CPR * 16) + (((j * 2 + (fgrp >> 1)) ^ (frow & (CPR - 1))) << 4) + (fgrp & 1) * 8)),
                         "v"(*reinterpret_cast<const sr_u32x2*>(pk))
                         : "memory");
          } else {
            const long m = m0 + wm * FM * 16 + i * 16 + frow;
            const int n = n0 + wn * FN * 16 + j * 16 + fgrp * 4;
            store4<TO>((m < p.M && n < Nv) ? (TO*)p.C + m * p.ldc + n : reinterpret_cast<TO*>((char*)p.trash_page + el * 16), v);
          }
        }
      }
    };
    put(0);
    if (STAGED && store) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const char* const stg = stg0 + (i % NSTG) * (16 * CPR * 16);
        sr_u32x4 val[NH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int r16 = h * RPI + el / CPR;
          val[h] = *reinterpret_cast<const sr_u32x4*>(stg + r16 * (CPR * 16) + ((((el % CPR)) ^ (r16 & (CPR - 1))) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < FM) put(i + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int r16 = h * RPI + el / CPR, c8 = (el % CPR) * 8;
          const long mm = m0 + wm * FM * 16 + i * 16 + r16;
          const int nn = n0 + wn * FN * 16 + c8;
          const bool okk = (mm < p.M) && (nn + 8 <= Nv);
          *reinterpret_cast<sr_u32x4*>(okk ? (char*)((TO*)p.C + mm * p.ldc + nn) : (char*)p.trash_page + el * 16) = val[h];
        }
      }
    } else {
#pragma unroll
      for (int i = 1; i < FM; ++i) put(i);
    }
    if (++st_cnt == SR_STATS_FLUSH || tile + G >= ntiles) stats_flush(false);
   }
  };

  // (A start offset of half a tile period between the two workgroups of a CU -- so that one drains its stores while the
  //  other multiplies -- was worth a few percent with the first K loops and costs 2-4 % with this one: removed.)
  // ---------------- K loop ----------------
  // One barrier per step, everything else straight-line:
  //     wait (my pieces of step s) ; barrier ; issue the pieces of step s+D ; read the 12 fragments of step s ; 32 MFMAs
  // D = NSLOT-1 steps of LDS-DMA stay in flight.  The slot refilled at step s is the one step s-1 was read from: every
  // wave has issued the MFMAs that consumed those reads before it reaches this step's barrier.  The loader's state is
  // scalar and incremental -- two byte offsets that advance by one K-step, and a countdown to the next "segment" change
  // (conv tap, operand pair, tile) where they are recomputed -- because scalar bookkeeping in front of the matrix
  // instructions is not free: 60 dependent SALU instructions per step cost ~340 cycles of a ~1700-cycle step
  // (tools/ubench/dma_shapes.hip), and the fully general per-step form of this loop used about that many.
  {
    constexpr int D = NSLOT - 1;
    static_assert((D - 1) * L + S < 64, "vmcnt is a 6-bit counter");
    constexpr int STEPB = BK * (int)sizeof(T);
    int seg_left = 0;
    auto refresh = [&]() {                       // offsets and countdown for the step (ld_tile, ld_pr, ld_kt)
      const int kl = ld_kt - ld_k0;
      const int k = kl * BK;
      const int plen = ld_pr == 0 ? p.nk[0] : (ld_pr == 1 ? p.nk[1] : p.nk[2]);
      seg_left = plen - kl;
      st_woff = k * (int)sizeof(T);
      if (CONV) {
        st_tap = k >> p.cv.lgCseg;
        const int cc = k & ((1 << p.cv.lgCseg) - 1);
        const int dh = p.cv.KW == 1 ? st_tap : (st_tap * 11) >> 5, dw = st_tap - dh * p.cv.KW;
        st_aoff = ((dh * p.cv.Wd + dw) * p.cv.cpix + cc) * (int)sizeof(T);
        const int tleft = ((1 << p.cv.lgCseg) - cc) / BK;
        seg_left = tleft < seg_left ? tleft : seg_left;
      } else {
        st_aoff = st_woff;
      }
    };
    auto advance_tail = [&]() {                  // the segment ended: next tap / operand pair / tile
      if (ld_kt == nkt) {
        ld_kt = 0; ld_pr = 0; ld_k0 = 0;
        ld_tile += G;
        if (ld_tile < ntiles) setup_ptrs(ld_tile, 0);
      } else if (ld_kt - ld_k0 == (ld_pr == 0 ? p.nk[0] : (ld_pr == 1 ? p.nk[1] : p.nk[2]))) {
        ld_k0 = ld_kt;
        ++ld_pr;
        setup_ptrs(ld_tile, ld_pr);
      }
      refresh();
    };
    auto advance = [&]() {
      ++ld_kt;
      if (--seg_left > 0) { st_aoff += STEPB; st_woff += STEPB; return; }
      advance_tail();
    };
    auto issue_step = [&](int slot) {
      st_slot = slot;
#pragma unroll
      for (int q = 0; q < L; ++q) issue_piece(q);
      advance();
    };
    refresh();
    int issued = 0;
    for (; issued < D && issued < total; ++issued) issue_step(issued);
    int c_tile = vb, c_left = nkt, since_epi = 0;
    // (the +S stores of an epilogue are younger than the pieces of the next D steps: their waits may leave them in flight)
    auto slow_wait = [&](int later) {
      const bool st = since_epi > 0 && !two && S > 0;
      if (since_epi > 0) --since_epi;
      if (st) {
        if (later >= 2) wait_vm<2 * L + S>(); else if (later == 1) wait_vm<L + S>(); else wait_vm<S>();
      } else {
        if (later >= 2) wait_vm<2 * L>(); else if (later == 1) wait_vm<L>(); else wait_vm<0>();
      }
    };
#ifdef SR_STAMPS
    const bool stamp = (p.debug & 4) != 0;
    unsigned long long tw = 0, tb = 0, tm_ = 0, te = 0, t0 = 0, t1 = 0;
#endif
    Frag<T> a[FM], b[FN];
    int slot = 0, islot = D % NSLOT;
    if (CFG == 4 && !(p.debug & 32)) {
      // ---- 256x256 tile, 8 waves: the two wave groups (wm = 0 / 1: one wave of each per SIMD) run HALF A STEP apart.
      // A step is an L section (12 fragment reads, the 4 DMA pieces of step s+3, wait for the reads) and an M section
      // (32 back-to-back MFMAs at raised priority, no memory instruction), each closed by a barrier; group 1 starts one
      // barrier late, so while one wave of a SIMD multiplies the other one loads.  Skeleton of this schedule
      // (tools/ubench/dma_shapes.hip): 1265 cycles per step against 1735 for the lock-step form.
      // Barrier instances pair as G0.B1(s) = G1.B2(s-1), G0.B2(s) = G1.B1(s).
      //   read-after-DMA : step s is read in L(s); every wave has waited for its own pieces of step s before the last
      //                    instance both groups pass ahead of that (G0: end of M(s-1), before B2(s-1); G1: in L(s-1), before B1(s-1)).
      //   DMA-after-read : the pieces of step s+3 overwrite the slot of step s-1, whose reads both groups drained
      //                    (lgkmcnt 0) before their B1(s-1) -- an instance both have passed when L(s) starts.
      // At a tile end group 0 takes one extra barrier (the groups re-align and run the epilogue together; otherwise the
      // two epilogues would serialise), then group 1 takes one to fall half a step behind again.
      static_assert((D - 1) * L + S < 64 && D <= 4, "vmcnt is a 6-bit counter");
      auto wait_later = [&](int later, bool st) {   // all but the `later` youngest DMA groups (+ an epilogue's S stores)
        if (st) {
          if (later >= 3) wait_vm<3 * L + S>(); else if (later == 2) wait_vm<2 * L + S>(); else if (later == 1) wait_vm<L + S>(); else wait_vm<S>();
        } else {
          if (later >= 3) wait_vm<3 * L>(); else if (later == 2) wait_vm<2 * L>(); else if (later == 1) wait_vm<L>(); else wait_vm<0>();
        }
      };
      wait_later(issued - 1, false);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (wm == 1) __builtin_amdgcn_s_barrier();
      auto wait_next = [&](int s) {          // my pieces of step s+1 (called after the pieces of step s+D were issued)
        const bool st = since_epi > 0 && !two && S > 0;
        if (since_epi > 0) --since_epi;
        if (s + 1 >= total) return;
        wait_later(issued - s - 2, st);
      };
      // Runs of steps are an inner loop that never redefines the loader's registers: a segment change (next tap /
      // operand pair / tile: setup_ptrs) happens BETWEEN runs, as does the epilogue.  With that slow path inside the step,
      // hipcc copied ~25 loop-carried registers per step; two copies of the step body (steady / generic) made it spill.
      // FIRST (a tile's first step; PLAIN kernels): the MFMAs take a zero C operand instead of accumulators that 128
      // `v_accvgpr_write` per wave cleared after the epilogue
      auto pp_step = [&](auto FIRST_, bool steady, int s) {
        constexpr bool FIRST = decltype(FIRST_)::value;
#ifdef SR_STAMPS
        if (stamp) SR_STAMP(t0);
#endif
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = rdB(slot, j);
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = rdA(slot, i);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        if (issued < total) {
          st_slot = islot;
#pragma unroll
          for (int q = 0; q < L; ++q) issue_piece(q);
          ++ld_kt; --seg_left; st_aoff += STEPB; st_woff += STEPB;
          ++issued;
          islot = islot + 1 == NSLOT ? 0 : islot + 1;
        }
        if (wm == 1) { if (steady) wait_vm<(D - 1) * L>(); else wait_next(s); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t1); tw += t1 - t0; }
#endif
        __builtin_amdgcn_s_barrier();                    // B1
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t0); tb += t0 - t1; }
#endif
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < FN / 2; ++k) {
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            if constexpr (FIRST) {
              acc[2 * k][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
              acc[2 * k + 1][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            mma<T>(b[2 * k], a[i], acc[2 * k][i]);
            mma<T>(b[2 * k + 1], a[i], acc[2 * k + 1][i]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        if (wm == 0) { if (steady) wait_vm<(D - 1) * L>(); else wait_next(s); }
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t1); tm_ += t1 - t0; }
#endif
        __builtin_amdgcn_s_barrier();                    // B2
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t0); tb += t0 - t1; }
#endif
      };
      int s = 0;
      const bool zfirst = PLAIN && !(p.debug & 64);    // (SR_GEMM_DEBUG bit 64: clear the accumulators instead -- A/B measurements)
      while (s < total) {
        const bool steady = since_epi == 0 && issued - s == D && issued < total;   // (n >= 1 below: all three terms are)
        int n = 1;
        if (steady) {
          n = total - issued;
          n = c_left < n ? c_left : n;
          n = seg_left < n ? seg_left : n;
        }
        if (zfirst && c_left == nkt && s > 0) {        // a tile's first step behind an epilogue: one step in the general (counted-wait) form
          n = 1;
          pp_step(std::true_type{}, false, s);
        } else {
          for (int i = 0; i < n; ++i) pp_step(std::false_type{}, steady, s + i);
        }
        s += n;
        c_left -= n;
        if (seg_left == 0) advance_tail();
        if (c_left == 0) {
          if (wm == 0) __builtin_amdgcn_s_barrier();     // re-align
          stg_off = (slot == 0 ? NSLOT - 1 : slot - 1) * SLOT;   // the slot of the step just consumed: free until the next DMA
          if constexpr (PLAIN) epilogue_plain(c_tile); else epilogue(c_tile);
          since_epi = p.no_store ? 0 : D - 1;
          c_left = nkt;
          c_tile += G;
          if (!zfirst) clear_acc();
          if (s < total) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // every wave is done with its staging strip: the slot may be refilled
            if (wm == 1) __builtin_amdgcn_s_barrier();   // fall half a step behind again
          }
#ifdef SR_STAMPS
          if (stamp) { SR_STAMP(t1); te += t1 - t0; }
#endif
        }
      }
    } else {
      // ---- 4-wave tiles (two workgroups per CU): lock-step form of the same step; runs of steps as above
      auto plain_step = [&](bool steady, int s) {
#ifdef SR_STAMPS
        if (stamp) SR_STAMP(t0);
#endif
        if (steady) { if (D == 3) wait_vm<2 * L>(); else wait_vm<L>(); }
        else slow_wait(issued - s - 1);
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t1); tw += t1 - t0; }
#endif
        __builtin_amdgcn_s_barrier();
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t0); tb += t0 - t1; }
#endif
        if (issued < total) {
          st_slot = islot;
#pragma unroll
          for (int q = 0; q < L; ++q) issue_piece(q);
          ++ld_kt; --seg_left; st_aoff += STEPB; st_woff += STEPB;
          ++issued;
          islot = islot + 1 == NSLOT ? 0 : islot + 1;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) b[j] = rdB(slot, j);
#pragma unroll
        for (int i = 0; i < FM; ++i) a[i] = rdA(slot, i);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k < FN / 2; ++k) {
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            mma<T>(b[2 * k], a[i], acc[2 * k][i]);
            mma<T>(b[2 * k + 1], a[i], acc[2 * k + 1][i]);
          }
        }
#ifdef SR_STAMPS
        if (stamp) { SR_STAMP(t1); tm_ += t1 - t0; t0 = t1; }
#endif
      };
      int s = 0;
      while (s < total) {
        const bool steady = since_epi == 0 && issued - s == D && issued < total;   // (n >= 1 below: all three terms are)
        int n = 1;
        if (steady) {
          n = total - issued;
          n = c_left < n ? c_left : n;
          n = seg_left < n ? seg_left : n;
        }
        for (int i = 0; i < n; ++i) plain_step(steady, s + i);
        s += n;
        c_left -= n;
        if (seg_left == 0) advance_tail();
        if (c_left == 0) {
          if constexpr (PLAIN) epilogue_plain(c_tile); else epilogue(c_tile);
          since_epi = p.no_store ? 0 : D;
          c_left = nkt;
          c_tile += G;
          clear_acc();
#ifdef SR_STAMPS
          if (stamp) { SR_STAMP(t1); te += t1 - t0; }
#endif
        }
      }
    }
    if constexpr (STATS_OK) {
      if (p.stats) {
        while (st_flush < p.stats_nflush) stats_flush(true);   // the rows of flushes this workgroup never reached: zeros
      }
    }
#ifdef SR_STAMPS
    if (stamp && blockIdx.x < 256 && (threadIdx.x & 63) == 0) {
      unsigned long long* o = g_stamps + ((blockIdx.x * 8 + wave) & 2047) * 8;
      o[0] = tw; o[1] = tb; o[3] = tm_; o[4] = te; o[5] = (unsigned long long)total;
      o[2] = te_prep; o[6] = te_stats; o[7] = te_store;
    }
#endif
    return;
  }

}

constexpr int v3_threads(int CFG) { return (CFG == 1 || CFG == 8) ? 256 : 128 * CFG; }
template <typename T, typename TO, int CFG, int EPI>
__global__ __launch_bounds__(v3_threads(CFG), CFG == 8 ? 1 : 2) void gemm_nt_v3_kernel(const KArgs p) { gemm_body_v3<T, TO, false, CFG, EPI>(p); }
template <typename T, typename TO, int CFG, int EPIX>
__global__ __launch_bounds__(v3_threads(CFG), CFG == 8 ? 1 : 2) void conv_igemm_v3_kernel(const KArgs p) { gemm_body_v3<T, TO, true, CFG, EPIX>(p); }

inline int num_cus() { return sr_num_cus(); }

template <typename T, typename TO, int WM, int WN, int EPI>
int launch_v2e(const KArgs& k, unsigned grid, size_t lds, hipStream_t st) {
  if (!sr_set_dynamic_lds<&gemm_nt_v2_kernel<T, TO, WM, WN, EPI>>((int)lds)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_nt_v2_kernel<T, TO, WM, WN, EPI>), dim3(grid), dim3(WM * WN * 64), lds, st, k);
  return SR_OK;
}

template <typename T, typename TO, int WM, int WN>
int launch_v2(const KArgs& k, hipStream_t st) {
  constexpr int BM = WM * 64, BN = WN * 64;
  const long gm = ((long)k.M + BM - 1) / BM, gn = (k.N + BN - 1) / BN;
  if (gm * gn > 0x7fffffffL) return SR_ERR_ARG;
  const size_t lds = 3 * (BM + BN) * 128;
  const long ntiles = gm * gn;
  const unsigned grid = (unsigned)(ntiles < num_cus() ? ntiles : num_cus());
  int rc = SR_OK;
  if (k.cv.on) {
    if (!sr_set_dynamic_lds<&conv_igemm_v2_kernel<T, TO, WM, WN>>((int)lds)) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_igemm_v2_kernel<T, TO, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, st, k);
  } else {
    switch (k.act) {
      case SR_ACT_SIGMOID: rc = launch_v2e<T, TO, WM, WN, SR_ACT_SIGMOID>(k, grid, lds, st); break;
      case SR_ACT_TANH: rc = launch_v2e<T, TO, WM, WN, SR_ACT_TANH>(k, grid, lds, st); break;
      case SR_ACT_SIGMOID_MUL: rc = launch_v2e<T, TO, WM, WN, SR_ACT_SIGMOID_MUL>(k, grid, lds, st); break;
      case SR_ACT_TANH_BLEND: rc = launch_v2e<T, TO, WM, WN, SR_ACT_TANH_BLEND>(k, grid, lds, st); break;
      default: rc = launch_v2e<T, TO, WM, WN, 0>(k, grid, lds, st); break;
    }
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

inline bool no_v3_n64() {
  static const bool off = [] { const char* e = getenv("SR_GEMM_NO_V3_N64"); return e && e[0] == '1'; }();
  return off;
}
inline bool use_v3() {
  static const bool off = [] { const char* e = getenv("SR_GEMM_NO_V3"); return e && e[0] == '1'; }();
  return !off;
}

template <typename T, typename TO, int WN, int EPI>
int launch_v3e(const KArgs& k, unsigned grid, size_t lds, hipStream_t st) {
  constexpr int NTHR = v3_threads(WN);
  if (!sr_set_dynamic_lds<&gemm_nt_v3_kernel<T, TO, WN, EPI>>((int)lds)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_nt_v3_kernel<T, TO, WN, EPI>), dim3(grid), dim3(NTHR), lds, st, k);
  return SR_OK;
}

// Grid and partial-statistics layout of a v3 launch (shared by the launcher and sr_gemm_stats_tiles): with statistics
// the grid is a multiple of the column-tile count gn, so that a workgroup's tiles all cover the same columns.
struct V3Plan { long grid, gn, nflush, rows; };
inline V3Plan v3_plan(long M, int N, int cfg, bool stats) {
  const int BN = 64 * cfg, wg_per_cu = cfg == 4 ? 1 : 2, waves_m = cfg == 1 ? 4 : 2;
  // SR_GEMM_HALF=1 (experiment: two streams sharing every CU): one workgroup per CU also for the 4-wave tiles, so that a
  // kernel of another stream finds half of every CU's registers and LDS free
  static const bool half = [] { const char* e = getenv("SR_GEMM_HALF"); return e && e[0] == '1'; }();
  const long gm = (M + 255) / 256, gn = (N + BN - 1) / BN, ntiles = gm * gn, cap = (long)num_cus() * (half ? 1 : wg_per_cu);
  V3Plan pl{ntiles < cap ? ntiles : cap, gn, 0, 0};
  if (stats) {
    if (pl.grid >= gn) pl.grid -= pl.grid % gn; else pl.grid = gn;
    const long groups = pl.grid / gn, per_wg = (gm + groups - 1) / groups;
    pl.nflush = (per_wg + SR_STATS_FLUSH - 1) / SR_STATS_FLUSH;
    pl.rows = groups * pl.nflush * waves_m;
  }
  return pl;
}

template <typename T, typename TO, int WN>
int launch_v3(const KArgs& k_in, hipStream_t st) {
  KArgs k = k_in;
  for (int i = 0; i < 3; ++i) k.nk[i] *= 2;  // host counts 128-byte K-tiles; v3 steps are 64 bytes
  constexpr int BN = WN == 8 ? 256 : 64 * WN, NSLOT = (WN == 4 || WN == 8) ? 4 : 3, NTHR = v3_threads(WN);
  const long gm = ((long)k.M + 255) / 256, gn = (k.N + BN - 1) / BN;
  if (gm * gn > 0x7fffffffL) return SR_ERR_ARG;
  const size_t lds = WN == 4 ? (size_t)5 * (256 + BN) * 64 : NSLOT * (256 + BN) * 64 + (NTHR / 64) * (WN == 8 ? 4096 : 2048);
  const V3Plan pl = v3_plan(k.M, k.N, WN, k.stats != nullptr);
  k.stats_nflush = (int)pl.nflush;
  const unsigned grid = (unsigned)pl.grid;
  int rc = SR_OK;
  // raw output + statistics and nothing else (every train-mode convolution of the backbone): the one-pass epilogue, EPIX 7
  const bool plain = sizeof(T) == 2 && sizeof(TO) == 2 && k.stats && !k.bias && !k.bias2 && !k.res && !k.escale && k.act == SR_ACT_NONE;
  if (k.cv.on) {
    if (plain) {
      constexpr int E = (sizeof(T) == 2 && sizeof(TO) == 2) ? 7 : 1;
      if (!sr_set_dynamic_lds<&conv_igemm_v3_kernel<T, TO, WN, E>>((int)lds)) return SR_ERR_LAUNCH;
      hipLaunchKernelGGL((conv_igemm_v3_kernel<T, TO, WN, E>), dim3(grid), dim3(NTHR), lds, st, k);
    } else if (k.stats) {
      if (!sr_set_dynamic_lds<&conv_igemm_v3_kernel<T, TO, WN, 1>>((int)lds)) return SR_ERR_LAUNCH;
      hipLaunchKernelGGL((conv_igemm_v3_kernel<T, TO, WN, 1>), dim3(grid), dim3(NTHR), lds, st, k);
    } else if (sizeof(TO) == 2 && k.res) {
      constexpr int E = sizeof(TO) == 2 ? 6 : 0;
      if (!sr_set_dynamic_lds<&conv_igemm_v3_kernel<T, TO, WN, E>>((int)lds)) return SR_ERR_LAUNCH;
      hipLaunchKernelGGL((conv_igemm_v3_kernel<T, TO, WN, E>), dim3(grid), dim3(NTHR), lds, st, k);
    } else {
      if (!sr_set_dynamic_lds<&conv_igemm_v3_kernel<T, TO, WN, 0>>((int)lds)) return SR_ERR_LAUNCH;
      hipLaunchKernelGGL((conv_igemm_v3_kernel<T, TO, WN, 0>), dim3(grid), dim3(NTHR), lds, st, k);
    }
  } else if constexpr (WN == 1) {
    // 256x64 tiles: linear epilogue only (caller guarantees)
    rc = k.stats ? launch_v3e<T, TO, WN, 1>(k, grid, lds, st)
                 : (sizeof(TO) == 2 && k.res ? launch_v3e<T, TO, WN, (sizeof(TO) == 2 ? 6 : 0)>(k, grid, lds, st) : launch_v3e<T, TO, WN, 0>(k, grid, lds, st));
  } else {
    switch (k.act) {
      case SR_ACT_SIGMOID: rc = launch_v3e<T, TO, WN, SR_ACT_SIGMOID>(k, grid, lds, st); break;
      case SR_ACT_TANH: rc = launch_v3e<T, TO, WN, SR_ACT_TANH>(k, grid, lds, st); break;
      case SR_ACT_SIGMOID_MUL: rc = launch_v3e<T, TO, WN, SR_ACT_SIGMOID_MUL>(k, grid, lds, st); break;
      case SR_ACT_TANH_BLEND: rc = launch_v3e<T, TO, WN, SR_ACT_TANH_BLEND>(k, grid, lds, st); break;
      default:
        rc = k.stats ? launch_v3e<T, TO, WN, 1>(k, grid, lds, st)
                     : (sizeof(TO) == 2 && k.res ? launch_v3e<T, TO, WN, (sizeof(TO) == 2 ? 6 : 0)>(k, grid, lds, st) : launch_v3e<T, TO, WN, 0>(k, grid, lds, st));
        break;
    }
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

// Which v3 tile shape serves (M, N)?  4: 256x256 (8 waves), 2: 256x128, 1: 256x64 (both 4 waves, two workgroups per CU), 0: not v3.
// The 256x64 shape is instantiated for linear epilogues only; the GRU gate epilogues have 256x128 besides 256x256 (round 4: the
// small-batch launches of the GGNN).  sr_gemm_stats_tiles() must agree with this, so it depends on M, N and the calling
// thread's CU share (sr_set_cu_share: thread-local, so a query and the launch it sizes see the same value) alone.
inline int v3_cfg(long M, int N, bool linear) {
  if (!use_v3()) return 0;
  static const int force = [] { const char* e = getenv("SR_GEMM_NARROW"); return e ? atoi(e) : -1; }();
  const long cus = num_cus();
  const long gm = (M + 255) / 256;
  if (linear) {
    if (N <= 64) return no_v3_n64() ? 0 : 1;
    if (N <= 128 || force > 0) return 2;
    // Few tiles (small batch, single-image inference): a 256x256 tile would leave most CUs idle and run its whole K loop on
    // one of them; narrower tiles multiply the workgroups.
    const long tw = gm * ((N + 255) / 256);
    if (tw * 8 <= cus) return 1;
    if (tw * 4 <= cus) return 2;
    // Wave quantisation at medium sizes: two workgroups per CU (512 slots) halve the granule.
    if (tw < 6 * cus) {
      const long tn = gm * ((N + 127) / 128);
      const double ew = (double)tw / (double)(((tw + cus - 1) / cus) * cus);
      const double en = (double)tn / (double)(((tn + 2 * cus - 1) / (2 * cus)) * (2 * cus));
      if (en > ew + 0.08) return 2;
    }
    return 4;
  }
  if (N <= 128) return 0;
  // GRU gate epilogues: 256x128 tiles when every one of them still gets a CU of its own (the verb path of an 8-GPU share: 768 rows are 24
  // tiles of 256 x 256 on 256 CUs).  Between half a round and a full one the narrow tiles do not pay: two of them sharing a CU take as
  // long as the one wide tile they replace (SR_GEMM_GATE_NARROW=2 forces them there for the A/B, =0 switches them off).
  static const int gate_narrow = [] { const char* e = getenv("SR_GEMM_GATE_NARROW"); return e ? atoi(e) : 1; }();
  const long tw = gm * ((N + 255) / 256);
  return (gate_narrow == 2 ? tw < cus : (gate_narrow == 1 && 2 * tw <= cus)) ? 2 : 4;
}


template <typename T, typename TO>
int launch(const KArgs& k, hipStream_t st) {
  const bool linear = k.act == SR_ACT_NONE || k.act == SR_ACT_RELU;
  // N not a multiple of 8: 16-bit outputs (see the staged store of the v3 epilogue) and launches with statistics
  // (sr_gemm_stats_tiles does not know the output type) take the v2 kernels
  const bool ragged16 = (k.N & 7) != 0 && (sizeof(TO) == 2 || k.stats != nullptr);
  SR_ROUTE(ragged16 ? 0 : v3_cfg(k.M, k.N, linear));
  switch (ragged16 ? 0 : v3_cfg(k.M, k.N, linear)) {
    case 4: return launch_v3<T, TO, 4>(k, st);
    case 2: return launch_v3<T, TO, 2>(k, st);
    case 1: return launch_v3<T, TO, 1>(k, st);
    default: break;
  }
  return k.N <= 64 ? launch_v2<T, TO, 4, 1>(k, st) : launch_v2<T, TO, 4, 2>(k, st);
}

inline int debug_flags() {
  static const int f = [] { const char* e = getenv("SR_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
  return f;
}


int dispatch(const KArgs& k_in, int dtype, int out_f32, hipStream_t st) {
  KArgs k = k_in;
  k.debug = debug_flags();
  if (!sr_route_probe) {
    k.zero_page = SR_DEVICE_SYMBOL(g_zero_page);       // (per device: see common.h)
    k.trash_page = SR_DEVICE_SYMBOL(g_trash_page);
    if (!k.zero_page || !k.trash_page) return SR_ERR_LAUNCH;
  }
  if (dtype == SR_F32) return launch<float, float>(k, st);
  if (dtype == SR_BF16) return out_f32 ? launch<bf16_t, float>(k, st) : launch<bf16_t, bf16_t>(k, st);
  return SR_ERR_DTYPE;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int sr_debug_stamps(unsigned long long* host_out, int count) {
  if (!host_out || count <= 0 || count > 256 * 8 * 8) return SR_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return SR_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * count) != hipSuccess) return SR_ERR_LAUNCH;
  return SR_OK;
}

extern "C" int sr_gemm_stats_tiles(int M, int N) {
  const int gm = (M + 255) / 256;
  const int cfg = (N & 7) ? 0 : v3_cfg(M, N, true);    // statistics are only produced with a linear epilogue
  if (cfg == 0) return gm;               // v2: one row per 256-row tile
  // v3: one row per (workgroup column-tile group, flush, wave group): each workgroup keeps running sums over up to
  // SR_STATS_FLUSH of its tiles (all of which cover the same columns)
  return (int)v3_plan(M, N, cfg, true).rows;
}

extern "C" int sr_conv_stats_rows(const sr_conv_args* a, int dtype) {
  if (!a || a->B <= 0) return SR_ERR_ARG;
  int Ho, Wo;
  if (a->stem) {
    Ho = (a->H + 6 - 7) / 2 + 1; Wo = (a->W + 6 - 7) / 2 + 1;
    if (dtype == SR_BF16 && use_v3() && !a->res && !a->escale) {
      const int r = srx_stem_rows(a);
      if (r != SR_ERR_UNSUPPORTED) return r;
    }
  } else {
    if (a->stride <= 0) return SR_ERR_ARG;
    Ho = (a->H + 2 * a->pad - a->KH) / a->stride + 1;
    Wo = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
    if (dtype == SR_BF16 && use_v3()) {
      int r = srx_c3d_rows(a);
      if (r != SR_ERR_UNSUPPORTED) return r;
      r = srx_c3d128s_rows(a);
      if (r != SR_ERR_UNSUPPORTED) return r;
      r = srx_c3d256_rows(a);
      if (r != SR_ERR_UNSUPPORTED) return r;
    }
  }
  const long M = (long)a->B * Ho * Wo;
  if (M <= 0 || M > 0x7fffffffL) return SR_ERR_ARG;
  return sr_gemm_stats_tiles((int)M, a->Cout);
}

extern "C" int sr_conv_in_affine_supported(const sr_conv_args* a, int dtype) {
  if (!a || a->B <= 0 || a->stride <= 0 || dtype != SR_BF16 || !use_v3() || a->stem) return 0;
  const long Ho = (a->H + 2 * a->pad - a->KH) / a->stride + 1, Wo = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
  return (srx_conv1x1_in_affine_ok(a, (long)a->B * Ho * Wo) || srx_c3d_in_affine_ok(a) || srx_c3d128s_in_affine_ok(a) || srx_c3d256_in_affine_ok(a)) ? 1 : 0;
}

thread_local int sr_route_probe = 0;
thread_local int sr_route_code = -1;

extern "C" int sr_conv_route(const sr_conv_args* a, int dtype) {
  sr_route_probe = 1;
  sr_route_code = -1;
  const int rc = sr_conv2d(a, dtype, nullptr);
  sr_route_probe = 0;
  if (rc != SR_OK) return rc;
  return sr_route_code >= 0 ? sr_route_code : SR_ERR_LAUNCH;
}

extern "C" int sr_gemm_tile_cfg(int M, int N, int linear, int out_16bit) {
  if (M <= 0 || N <= 0) return SR_ERR_ARG;
  if ((N & 7) != 0 && out_16bit) return 0;     // (see `launch`: ragged 16-bit outputs take the v2 kernels)
  return v3_cfg(M, N, linear != 0);
}

extern "C" int sr_gemm(const sr_gemm_args* a, int dtype, void* stream) {
  if (!a || a->npairs < 1 || a->npairs > 3 || a->M <= 0 || a->N <= 0 || !a->C) return SR_ERR_ARG;
  if (dtype != SR_F32 && dtype != SR_BF16) return SR_ERR_DTYPE;
  const int es = dtype == SR_F32 ? 4 : 2, bk = 128 / es;
  const int os = (dtype == SR_F32 || a->out_f32) ? 4 : 2;
  KArgs k{};
  for (int i = 0; i < a->npairs; ++i) {
    const sr_kpair& kp = a->kp[i];
    if (!kp.A || !kp.W || kp.K <= 0 || kp.K % bk) return SR_ERR_ARG;
    if (!aligned16(kp.A) || !aligned16(kp.W) || (kp.lda * es) % 16 || (kp.ldw * es) % 16) return SR_ERR_ARG;
    k.kp[i] = kp;
    k.nk[i] = kp.K / bk;
  }
  // vector epilogue accesses need 4-element alignment of every row
  if (a->ldc % 4 || (reinterpret_cast<uintptr_t>(a->C) % (4 * os))) return SR_ERR_ARG;
  if (a->res && (a->ldres % 4 || reinterpret_cast<uintptr_t>(a->res) % (4 * os))) return SR_ERR_ARG;
  if (a->act == SR_ACT_SIGMOID_MUL && (!a->aux1 || !a->C2)) return SR_ERR_ARG;
  if (a->act == SR_ACT_TANH_BLEND && (!a->aux1 || !a->aux2 || !a->C2)) return SR_ERR_ARG;
  if (a->act < 0 || a->act > SR_ACT_TANH_BLEND) return SR_ERR_ARG;
  k.npairs = a->npairs; k.M = a->M; k.N = a->N; k.act = a->act;
  k.C = a->C; k.ldc = a->ldc; k.C2 = a->C2;
  k.bias = a->bias; k.bias2 = a->bias2; k.bias_scale = a->bias_scale;
  k.res = a->res; k.ldres = a->ldres; k.aux1 = a->aux1; k.aux2 = a->aux2;
  k.stats = a->stats;
  k.cv.on = 0;
  return dispatch(k, dtype, a->out_f32, (hipStream_t)stream);
}

extern "C" int sr_conv2d(const sr_conv_args* a, int dtype, void* stream) {
  if (!a || !a->x || !a->w || !a->y || a->B <= 0) return SR_ERR_ARG;
  if (dtype != SR_F32 && dtype != SR_BF16) return SR_ERR_DTYPE;
  const int es = dtype == SR_F32 ? 4 : 2, bk = 128 / es;
  if (!aligned16(a->x) || !aligned16(a->w) || !aligned16(a->y) || (a->res && !aligned16(a->res))) return SR_ERR_ARG;
  if (a->Cout % 4) return SR_ERR_ARG;
  KArgs k{};
  int Ho, Wo;
  if (a->stem) {
    // x: [B, H+6, Wp, 4] already zero padded; one "tap" = one filter row = 8 pixels x 4 channels
    if (a->KH != 7 || a->KW != 7 || a->stride != 2 || a->pad != 3 || a->Cin != 3) return SR_ERR_ARG;
    const int Hp = (a->H + 6 + 1) & ~1, Wp = (a->W + 6 + 1) & ~1;
    Ho = (a->H + 6 - 7) / 2 + 1; Wo = (a->W + 6 - 7) / 2 + 1;
    k.cv = ConvGeom{1, Hp, Wp, Ho, Wo, 2, 0, 1, 5, 4};
    k.kp[0].K = 8 * 32;
    // the fake 8th row/8th pixel must stay inside the padded image
    if ((Ho - 1) * 2 + 7 >= Hp || (Wo - 1) * 2 + 7 >= Wp) return SR_ERR_ARG;
    if (dtype == SR_BF16 && use_v3() && (a->act == SR_ACT_NONE || a->act == SR_ACT_RELU)) {   // direct stem convolution (stem.hip)
      const int rc = srx_stem_conv(a, stream);
      if (rc != SR_ERR_UNSUPPORTED) return rc;
    }
  } else {
    if (a->Cin % bk || (a->Cin & (a->Cin - 1))) return SR_ERR_ARG;  // power of two, >= one K-tile
    if (a->KH != a->KW || (a->KH != 1 && a->KH != 3) || a->KH * a->KW > 31) return SR_ERR_ARG;
    Ho = (a->H + 2 * a->pad - a->KH) / a->stride + 1;
    Wo = (a->W + 2 * a->pad - a->KW) / a->stride + 1;
    int lg = 0;
    while ((1 << lg) < a->Cin) ++lg;
    k.cv = ConvGeom{1, a->H, a->W, Ho, Wo, a->stride, a->pad, a->KW, lg, a->Cin};
    k.kp[0].K = a->KH * a->KW * a->Cin;
  }
  const long M = (long)a->B * Ho * Wo;
  if (M <= 0 || M > 0x7fffffffL) return SR_ERR_ARG;
  if (a->act != SR_ACT_NONE && a->act != SR_ACT_RELU) return SR_ERR_ARG;
  if (a->in_scale || a->in_shift) {       // input affine: only the weight-stationary expansion kernel applies it (never silently dropped)
    if (dtype != SR_BF16 || !use_v3() || a->stem) return SR_ERR_UNSUPPORTED;
    int rc = srx_conv1x1_expand(a, M, stream);
    if (rc != SR_ERR_UNSUPPORTED) return rc;
    rc = srx_c3d_conv(a, stream);
    if (rc != SR_ERR_UNSUPPORTED) return rc;
    rc = srx_c3d128s_conv(a, stream);
    return rc != SR_ERR_UNSUPPORTED ? rc : srx_c3d256_conv(a, stream);
  }
  if (dtype == SR_BF16 && use_v3()) {     // output-heavy 1x1 convolutions: the kernel that overlaps K loop and epilogue (expand.hip)
    int rc = srx_conv1x1_expand(a, M, stream);
    if (rc != SR_ERR_UNSUPPORTED) return rc;
    rc = srx_c3d_conv(a, stream);          // the 64-channel 3x3 layer: direct convolution, weights in registers (c3d.hip)
    if (rc != SR_ERR_UNSUPPORTED) return rc;
    rc = srx_c3d128s_conv(a, stream);      // the 128-channel 3x3 layer on 28 x 28 images: direct convolution over 32-channel patch slices (c3ds.hip)
    if (rc != SR_ERR_UNSUPPORTED) return rc;
    rc = srx_c3d256_conv(a, stream);       // the 256-channel 3x3 layer on 14 x 14 images: direct convolution over 32-channel patch slices (c3ds.hip)
    if (rc != SR_ERR_UNSUPPORTED) return rc;
  }
  k.kp[0].A = a->x; k.kp[0].W = a->w; k.kp[0].lda = 0; k.kp[0].ldw = k.kp[0].K;
  k.nk[0] = k.kp[0].K / bk;
  k.npairs = 1; k.M = (int)M; k.N = a->Cout; k.act = a->act;
  k.C = a->y; k.ldc = a->Cout; k.bias = a->bias; k.bias_scale = 1.f;
  k.res = a->res; k.ldres = a->Cout; k.stats = a->stats;
  k.escale = a->escale; k.no_store = a->no_store;
  if ((a->escale || a->no_store) && !use_v3()) return SR_ERR_UNSUPPORTED;
  if (a->no_store && !a->stats) return SR_ERR_ARG;
  return dispatch(k, dtype, 0, (hipStream_t)stream);
}
