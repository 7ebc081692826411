// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( sum_p A_p[M,K_p] . W_p[N,K_p]^T )
//
// One kernel serves nn.Linear-shaped GEMMs (GGNN gates, classifiers, their backward
// GEMMs) and NHWC convolutions (the A rows are gathered on the fly: row m is output
// pixel (b,ho,wo), K runs over (tap, channel)).
//
// Structure (cdna_hip_programming.md 5, "minimum 2-phase" loop):
//   * block tile (64*WAVES_M) x (64*WAVES_N), each wave owns a 64x64 output tile
//     = 4x4 fragments of v_mfma_f32_16x16x32_bf16 (or 16x16x4_f32 for fp32 storage);
//   * K-tile = 128 bytes per row (64 bf16 / 32 f32); both operands are staged
//     global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 rows x 128 B per
//     wave-instruction), double-buffered;
//   * the LDS image is kept lane-linear for the DMA; the bank-conflict swizzle
//     (16-B chunk c of row r lives at chunk position c ^ (r & 7)) is applied to the
//     per-lane SOURCE address and again on the fragment ds_read_b128 (rule 21);
//   * MFMA roles are swapped (weights = MFMA "A", activations = MFMA "B") so each lane
//     ends up with 4 CONSECUTIVE output columns of one output row -> 8/16-byte stores;
//   * convolution padding and row tails read from a zero page instead of branching;
//   * workgroup -> tile mapping is XCD-aware (each XCD walks a contiguous range of row
//     tiles so the activation panel is reused out of its own L2).
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

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
  int debug;  // SR_GEMM_DEBUG bits (diagnostic builds of bench scripts only): 1 = skip MFMA, 2 = skip loads after the prologue
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

// CONV selects the implicit-GEMM row gather; it is a separate kernel symbol (conv_igemm_kernel vs
// gemm_nt_kernel in profiles) so the backbone convolutions can be told apart from the head's GEMMs.
template <typename T, typename TO, int WAVES_M, int WAVES_N, bool CONV>
__device__ __forceinline__ void gemm_body(const KArgs& p) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64, NW = WAVES_M * WAVES_N;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;               // elements per K-tile (128 B)
  constexpr int A_PER_WAVE = (BM / 8) / NW, B_PER_WAVE = (BN / 8) / NW;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must divide over waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware tile mapping (bijective form, cdna_hip_programming.md 5) ----
  const int gn = (p.N + BN - 1) / BN;
  const int ntiles = gridDim.x;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = ntiles >> 3, r = ntiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / gn, tile_n = bid - tile_m * gn;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-lane loader geometry ----
  const int lrow = lane >> 3;                 // row inside a 1-KiB piece
  const int csrc = (lane & 7) ^ lrow;         // source chunk that lands at chunk position lane&7
  const int ecol = csrc * EPC;                // element offset of that chunk in the K-tile

  long a_row[A_PER_WAVE];      // plain: clamped row index; conv: element offset of x[b, ho*s-p, wo*s-p, 0]
  unsigned a_mask[A_PER_WAVE]; // conv: bit t set <=> tap t is inside the image
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) {
    long m = m0 + (wave + i * NW) * 8 + lrow;
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
        const int dh = p.cv.KW == 1 ? t : (t * 11) >> 5, dw = t - dh * p.cv.KW;  // KW in {1,3}
        const int hi = hi0 + dh, wi = wi0 + dw;
        if (hi >= 0 && hi < p.cv.H && wi >= 0 && wi < p.cv.Wd) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  }
  int w_row[B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) {
    int n = n0 + (wave + i * NW) * 8 + lrow;
    w_row[i] = n < p.N ? n : p.N - 1;
  }

  auto stage = [&](int buf, int kt) {
    int pr = 0, kl = kt;
    if (p.npairs > 1 && kl >= p.nk[0]) { kl -= p.nk[0]; pr = 1; }
    if (p.npairs > 2 && pr == 1 && kl >= p.nk[1]) { kl -= p.nk[1]; pr = 2; }
    const sr_kpair& kp = p.kp[pr];
    const int k = kl * BK + ecol;
    char* sA = smem + buf * STAGE;
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
      if (!CONV) {
        src = (const T*)kp.A + a_row[i] * kp.lda + k;
      } else {
        src = ((a_mask[i] >> tap) & 1) ? (const T*)kp.A + a_row[i] + tapoff : (const T*)g_zero_page;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sA + (wave + i * NW) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      const T* src = (const T*)kp.W + (long)w_row[i] * kp.ldw + k;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];  // [n-fragment j][m-fragment i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fgrp = lane >> 4;
  auto compute = [&](int buf) {
    const char* sA = smem + buf * STAGE + (wm * 64 + frow) * 128;
    const char* sB = smem + buf * STAGE + BM * 128 + (wn * 64 + frow) * 128;
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

  int nkt = p.nk[0];
  if (p.npairs > 1) nkt += p.nk[1];
  if (p.npairs > 2) nkt += p.nk[2];

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt - 1; ++kt) {
    stage(cur ^ 1, kt + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // ---- epilogue: lane holds rows m = .. + i*16 + frow, columns n = .. + j*16 + fgrp*4 + {0..3}
  const bool want_stats = p.stats != nullptr;
  float s1[4][4], s2[4][4];
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
  }
  TO* C = (TO*)p.C;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fgrp * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.N) {
        if (p.bias) bv[r] = p.bias_scale * p.bias[n + r];
        if (p.bias2) bv[r] += p.bias2[n + r];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + wm * 64 + i * 16 + frow;
      if (m >= p.M || n >= p.N) continue;
      const bool full = (n + 3 < p.N);
      float v[4], o2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r] + bv[r];
      if (want_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
      }
      if (p.res) {
        const TO* rp = (const TO*)p.res + m * p.ldres + n;
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) load4<TO>(rp, rv);
        else
          for (int r = 0; r < 4 && n + r < p.N; ++r) rv[r] = to_f<TO>(rp[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      }
      bool two = false;
      switch (p.act) {
        case SR_ACT_RELU:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          break;
        case SR_ACT_SIGMOID:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
          break;
        case SR_ACT_TANH:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
          break;
        case SR_ACT_SIGMOID_MUL: {
          float h[4] = {0.f, 0.f, 0.f, 0.f};
          const TO* hp = (const TO*)p.aux1 + m * p.ldc + n;
          if (full) load4<TO>(hp, h);
          else
            for (int r = 0; r < 4 && n + r < p.N; ++r) h[r] = to_f<TO>(hp[r]);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v[r] = sigmoidf_(v[r]); o2[r] = v[r] * h[r]; }
          two = true;
        } break;
        case SR_ACT_TANH_BLEND: {
          float h[4] = {0.f, 0.f, 0.f, 0.f}, z[4] = {0.f, 0.f, 0.f, 0.f};
          const TO* hp = (const TO*)p.aux1 + m * p.ldc + n;
          const TO* zp = (const TO*)p.aux2 + m * p.ldc + n;
          if (full) { load4<TO>(hp, h); load4<TO>(zp, z); }
          else
            for (int r = 0; r < 4 && n + r < p.N; ++r) { h[r] = to_f<TO>(hp[r]); z[r] = to_f<TO>(zp[r]); }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float c = tanhf_(v[r]);
            o2[r] = c;
            v[r] = (1.f - z[r]) * h[r] + z[r] * c;
          }
          two = true;
        } break;
        default: break;
      }
      TO* cp = C + m * p.ldc + n;
      if (full) {
        store4<TO>(cp, v);
        if (two) store4<TO>((TO*)p.C2 + m * p.ldc + n, o2);
      } else {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          cp[r] = from_f<TO>(v[r]);
          if (two) ((TO*)p.C2)[m * p.ldc + n + r] = from_f<TO>(o2[r]);
        }
      }
    }
  }

  if (want_stats) {
    // columns are shared by the 16 lanes of a lane group and by the WAVES_M waves of a wave column
    float* red = reinterpret_cast<float*>(smem);  // [2][WAVES_M][BN]
    __syncthreads();                              // staging buffers are dead from here on
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[j][r], b = s2[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (frow == 0) {
          const int col = wn * 64 + j * 16 + fgrp * 4 + r;
          red[(0 * WAVES_M + wm) * BN + col] = a;
          red[(1 * WAVES_M + wm) * BN + col] = b;
        }
      }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * BN; t += NW * 64) {
      const int which = t / BN, col = t - which * BN;
      if (n0 + col < p.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) s += red[(which * WAVES_M + w) * BN + col];
        p.stats[((long)tile_m * 2 + which) * p.N + n0 + col] = s;
      }
    }
  }
}

template <typename T, typename TO, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_nt_kernel(const KArgs p) {
  gemm_body<T, TO, WAVES_M, WAVES_N, false>(p);
}
template <typename T, typename TO, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_igemm_kernel(const KArgs p) {
  gemm_body<T, TO, WAVES_M, WAVES_N, true>(p);
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

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <typename T, typename TO, int WAVES_M, int WAVES_N, bool CONV>
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
      else src = ((a_mask[i] >> tap) & 1) ? (const T*)kp.A + a_row[i] + tapoff : (const T*)g_zero_page;
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

  const bool two = (p.act == SR_ACT_SIGMOID_MUL || p.act == SR_ACT_TANH_BLEND);
  const int Nv = (p.N + 3) & ~3;   // columns that may be written (pad columns up to a multiple of 4 belong to the row)
  TO* const trash = reinterpret_cast<TO*>(g_trash_page + lane * 16);
  const TO* const zeros = reinterpret_cast<const TO*>(g_zero_page);

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
        switch (p.act) {
          case SR_ACT_RELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            break;
          case SR_ACT_SIGMOID:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sigmoidf_(v[r]);
            break;
          case SR_ACT_TANH:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
            break;
          case SR_ACT_SIGMOID_MUL: {
            float h[4];
            load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = sigmoidf_(v[r]); o2[r] = v[r] * h[r]; }
          } break;
          case SR_ACT_TANH_BLEND: {
            float h[4], z[4];
            load4<TO>(ok ? (const TO*)p.aux1 + m * p.ldc + n : zeros, h);
            load4<TO>(ok ? (const TO*)p.aux2 + m * p.ldc + n : zeros, z);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float c = tanhf_(v[r]);
              o2[r] = c;
              v[r] = (1.f - z[r]) * h[r] + z[r] * c;
            }
          } break;
          default: break;
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
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
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

template <typename T, typename TO, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_nt_v2_kernel(const KArgs p) {
  gemm_body_v2<T, TO, WAVES_M, WAVES_N, false>(p);
}
template <typename T, typename TO, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_igemm_v2_kernel(const KArgs p) {
  gemm_body_v2<T, TO, WAVES_M, WAVES_N, true>(p);
}

inline bool use_v1() {
  static const bool v1 = [] { const char* e = getenv("SR_GEMM_V1"); return e && e[0] == '1'; }();
  return v1;
}
inline int num_cus() {
  static const int n = [] {
    int dev = 0, cu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 256;
    return cu > 0 ? cu : 256;
  }();
  return n;
}

template <typename T, typename TO, int WM, int WN>
int launch_v2(const KArgs& k, hipStream_t st) {
  constexpr int BM = WM * 64, BN = WN * 64;
  const long gm = ((long)k.M + BM - 1) / BM, gn = (k.N + BN - 1) / BN;
  if (gm * gn > 0x7fffffffL) return SR_ERR_ARG;
  const size_t lds = 3 * (BM + BN) * 128;
  const long ntiles = gm * gn;
  const unsigned grid = (unsigned)(ntiles < num_cus() ? ntiles : num_cus());
  if (k.cv.on) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_v2_kernel<T, TO, WM, WN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_igemm_v2_kernel<T, TO, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, st, k);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v2_kernel<T, TO, WM, WN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_v2_kernel<T, TO, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, st, k);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

inline int tile_m_for(int N) { return use_v1() ? (N <= 64 ? 256 : 128) : 256; }

template <typename T, typename TO, int WM, int WN>
int launch_cfg(const KArgs& k, hipStream_t st) {
  constexpr int BM = WM * 64, BN = WN * 64;
  const long gm = ((long)k.M + BM - 1) / BM, gn = (k.N + BN - 1) / BN;
  if (gm * gn > 0x7fffffffL) return SR_ERR_ARG;
  const size_t lds = 2 * (BM + BN) * 128;
  if (k.cv.on) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, TO, WM, WN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL((conv_igemm_kernel<T, TO, WM, WN>), dim3((unsigned)(gm * gn)), dim3(WM * WN * 64), lds, st, k);
  } else {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<T, TO, WM, WN>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return SR_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO, WM, WN>), dim3((unsigned)(gm * gn)), dim3(WM * WN * 64), lds, st, k);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

template <typename T, typename TO>
int launch(const KArgs& k, hipStream_t st) {
  if (use_v1()) return k.N <= 64 ? launch_cfg<T, TO, 4, 1>(k, st) : launch_cfg<T, TO, 2, 2>(k, st);
  return k.N <= 64 ? launch_v2<T, TO, 4, 1>(k, st) : launch_v2<T, TO, 4, 2>(k, st);
}

inline int debug_flags() {
  static const int f = [] { const char* e = getenv("SR_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
  return f;
}

int dispatch(const KArgs& k_in, int dtype, int out_f32, hipStream_t st) {
  KArgs k = k_in;
  k.debug = debug_flags();
  if (dtype == SR_F32) return launch<float, float>(k, st);
  if (dtype == SR_BF16) return out_f32 ? launch<bf16_t, float>(k, st) : launch<bf16_t, bf16_t>(k, st);
  return SR_ERR_DTYPE;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int sr_gemm_stats_tiles(int M, int N) {
  const int bm = tile_m_for(N);
  return (M + bm - 1) / bm;
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
  k.kp[0].A = a->x; k.kp[0].W = a->w; k.kp[0].lda = 0; k.kp[0].ldw = k.kp[0].K;
  k.nk[0] = k.kp[0].K / bk;
  k.npairs = 1; k.M = (int)M; k.N = a->Cout; k.act = a->act;
  k.C = a->y; k.ldc = a->Cout; k.bias = a->bias; k.bias_scale = 1.f;
  k.res = a->res; k.ldres = a->Cout; k.stats = a->stats;
  return dispatch(k, dtype, 0, (hipStream_t)stream);
}
