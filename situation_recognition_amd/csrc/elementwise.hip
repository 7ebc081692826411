// HBM-bound kernels of the backbone and the bookkeeping around the GEMMs:
// stem layout prep, train-mode BatchNorm finalize/apply, pooling, transpose (+ column
// sums), cast, dropout.  All of them stream 16 bytes per lane (cdna_hip_programming.md
// Guideline 13) with grid-stride loops capped at 256 CUs x 8 workgroups.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;
inline unsigned grid_for(long work_items) {
  long g = (work_items + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (unsigned)g;
}

// ------------------------------------------------------------------ stem prep
// in: fp32 NCHW [B,3,H,W]; out: T [B, H+6, Wp, 4], zero border of 3 and zero 4th channel.
template <typename T>
__global__ void stem_prep_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int Hp, int Wp) {
  const long total = (long)B * Hp * Wp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int wp = (int)(idx % Wp);
    const long t = idx / Wp;
    const int hp = (int)(t % Hp);
    const long b = t / Hp;
    const int h = hp - 3, w = wp - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (h >= 0 && h < H && w >= 0 && w < W) {
      const float* src = img + ((b * 3) * H + h) * (long)W + w;
      v[0] = src[0]; v[1] = src[(long)H * W]; v[2] = src[2L * H * W];
    }
    T* dst = out + idx * 4;
    if constexpr (sizeof(T) == 2) {
      bf16_t tmp[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)0.f};
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(tmp);
    } else {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], 0.f);
    }
  }
}

// ------------------------------------------------------------------ uint8 input pipeline
// in: uint8 NHWC [B,H0,W0,3] (decoded, resized images); out: T [B,Hp,Wp,4] = the stem's padded NHWC4 input of an HxW crop:
// out[b,3+h,3+w,c] = (in[b, y0_b+h, x0_b+(flip_b ? W-1-w : w), c]/255 - mean[c]) / std[c]; zero border / 4th channel.
// Fuses RandomCrop/CenterCrop + RandomHorizontalFlip + ToTensor + Normalize (reference imsitu_encoder.py:21-36) with sr_stem_prep.
template <typename T>
__global__ void image_prep_u8_kernel(const uint8_t* __restrict__ img, T* __restrict__ out, int B, int H0, int W0, int H, int W,
                                     int Hp, int Wp, const int32_t* __restrict__ crop, const uint8_t* __restrict__ flip,
                                     float m0, float m1, float m2, float i0, float i1, float i2) {
  const long total = (long)B * Hp * Wp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int wp = (int)(idx % Wp);
    const long t = idx / Wp;
    const int hp = (int)(t % Hp);
    const long b = t / Hp;
    const int h = hp - 3, w = wp - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (h >= 0 && h < H && w >= 0 && w < W) {
      const int y = (crop ? crop[2 * b] : 0) + h;
      const int x = (crop ? crop[2 * b + 1] : 0) + ((flip && flip[b]) ? W - 1 - w : w);
      const uint8_t* src = img + ((b * H0 + y) * (long)W0 + x) * 3;
      v[0] = ((float)src[0] * (1.f / 255.f) - m0) * i0;
      v[1] = ((float)src[1] * (1.f / 255.f) - m1) * i1;
      v[2] = ((float)src[2] * (1.f / 255.f) - m2) * i2;
    }
    T* dst = out + idx * 4;
    if constexpr (sizeof(T) == 2) {
      bf16_t tmp[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)0.f};
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(tmp);
    } else {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], 0.f);
    }
  }
}

// ------------------------------------------------------------------ BN finalize
// Stage A: grid (C/64, chunks): each workgroup folds its slice of the per-tile partial sums into one
// fp64 partial per channel (deterministic order).  Stage B: one thread per channel folds the <= 256 chunk partials.
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ stats, int tiles, int C, int tiles_per_chunk,
                                                        double* __restrict__ scratch) {
  __shared__ double red[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int t0 = blockIdx.y * tiles_per_chunk;
  const int t1 = min(tiles, t0 + tiles_per_chunk);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int t = t0 + ty; t < t1; t += 4) {
      s1 += (double)stats[((long)t * 2 + 0) * C + c];
      s2 += (double)stats[((long)t * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = s1;
  red[1][ty][tx] = s2;
  __syncthreads();
  if (ty == 0 && c < C) {
    scratch[((long)blockIdx.y * 2 + 0) * C + c] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
    scratch[((long)blockIdx.y * 2 + 1) * C + c] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
  }
}

// Stage B: 16 channels x 64 row groups per workgroup (C/16 workgroups: 64-byte row segments, <= 64 rows per thread with four
// loads in flight); fixed summation order (deterministic).  TI = float: the partial rows themselves (few rows -- the conv
// kernels keep running sums -- so stage A is skipped).  [It was 64 channels x 16 groups: 4 workgroups for C = 256 and 32
// dependent iterations per thread, 16 us per launch; ~220 launches per training step.]
template <typename TI>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const TI* __restrict__ scratch, int chunks, int C, double inv_count,
                                                           double unbias, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float momentum, float eps,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ rmean2, float* __restrict__ rvar2, float momentum2) {
  __shared__ double red[2][64][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C) {
    int t = ty;
    for (; t + 192 < chunks; t += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a1[u] += (double)scratch[((long)(t + 64 * u) * 2 + 0) * C + c];
        a2[u] += (double)scratch[((long)(t + 64 * u) * 2 + 1) * C + c];
      }
    }
    for (; t < chunks; t += 64) {
      a1[0] += (double)scratch[((long)t * 2 + 0) * C + c];
      a2[0] += (double)scratch[((long)t * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  red[1][ty][tx] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  __syncthreads();
  if (threadIdx.x < 128) {                    // 4 row-group quarters x 2 sums x 16 channels
    const int q = threadIdx.x >> 5, w = (threadIdx.x >> 4) & 1;
    double s = 0.0;
    for (int t = 0; t < 16; ++t) s += red[w][q * 16 + t][tx];
    red[w][q * 16][tx] = s;                   // (each thread overwrites only the first row of its own quarter)
  }
  __syncthreads();
  if (ty != 0 || c >= C) return;
  const double s1 = (red[0][0][tx] + red[0][16][tx]) + (red[0][32][tx] + red[0][48][tx]);
  const double s2 = (red[1][0][tx] + red[1][16][tx]) + (red[1][32][tx] + red[1][48][tx]);
  const double mean = s1 * inv_count;
  double var = s2 * inv_count - mean * mean;  // biased (what normalisation uses)
  if (var < 0.0) var = 0.0;
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);
  if (rmean2) rmean2[c] = (1.f - momentum2) * rmean2[c] + momentum2 * (float)mean;      // a second BatchNorm fed the same batch
  if (rvar2) rvar2[c] = (1.f - momentum2) * rvar2[c] + momentum2 * (float)(var * unbias);
}

// ------------------------------------------------------------------ BN apply
// 2-D mapping: a thread owns one 16-byte channel group (its scale/shift live in registers) and walks rows.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       T* __restrict__ y, long rows, int C, int relu) {
  constexpr int N = Vec16<T>::N;
  const int cv = C / N;                       // 16-byte groups per row
  const int tpr = cv < 256 ? cv : 256;        // threads across a row
  const int rpb = 256 / tpr;                  // rows per workgroup pass
  const int cg0 = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  if (rsub >= rpb) return;
  for (int cg = cg0; cg < cv; cg += tpr) {
    float sc[N], sh[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { sc[k] = scale[cg * N + k]; sh[k] = shift[cg * N + k]; }
    for (long r = (long)blockIdx.x * rpb + rsub; r < rows; r += (long)gridDim.x * rpb) {
      const long e = r * C + (long)cg * N;
      Vec16<T> v = ld16<T>(x + e), o;
      if (res) {
        Vec16<T> rr = ld16<T>(res + e);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          float f = v.get(k) * sc[k] + sh[k] + rr.get(k);
          o.set(k, relu ? fmaxf(f, 0.f) : f);
        }
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          float f = v.get(k) * sc[k] + sh[k];
          o.set(k, relu ? fmaxf(f, 0.f) : f);
        }
      }
      st16<T>(y + e, o);
    }
  }
}

// Flat form for channel counts whose 16-byte groups per row divide 256 (every ResNet width in bf16): the tensor is one
// contiguous run of 16-byte elements; a workgroup walks contiguous chunks of 256 x U elements (32 KiB at U = 8), every
// thread's U accesses are 4 KiB apart and all belong to the same channel group, loads and stores are nontemporal.
// Measured on 2.4 GB (tools/ubench/stream_rates.hip): 5.8 TB/s for this shape against 4.8-5.2 for a row-strided walk.
typedef unsigned sr_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ Vec16<T> ld16_nt(const T* p) {
  Vec16<T> v;
  const sr_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const sr_u32x4*>(p));
  __builtin_memcpy(&v.raw, &t, 16);
  return v;
}
template <typename T> __device__ __forceinline__ void st16_nt(T* p, const Vec16<T>& v) {
  sr_u32x4 t;
  __builtin_memcpy(&t, &v.raw, 16);
  __builtin_nontemporal_store(t, reinterpret_cast<sr_u32x4*>(p));
}
template <typename T, int U>
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const T* __restrict__ res,
                                                            T* __restrict__ y, long n16, int cv, int relu) {
  constexpr int N = Vec16<T>::N;
  const int cg = threadIdx.x % cv;
  float sc[N], sh[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { sc[k] = scale[cg * N + k]; sh[k] = shift[cg * N + k]; }
  const long chunk = 256L * U;
  for (long c0 = (long)blockIdx.x * chunk; c0 < n16; c0 += (long)gridDim.x * chunk) {
    const long base = c0 + threadIdx.x;
    Vec16<T> v[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n16) v[u] = ld16_nt<T>(x + (base + u * 256) * N);
    if (res) {
#pragma unroll
      for (int u = 0; u < U; ++u) if (base + u * 256 < n16) r[u] = ld16_nt<T>(res + (base + u * 256) * N);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 < n16) {
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          float f = v[u].get(k) * sc[k] + sh[k];
          if (res) f += r[u].get(k);
          o.set(k, relu ? fmaxf(f, 0.f) : f);
        }
        st16_nt<T>(y + (base + u * 256) * N, o);
      }
    }
  }
}

// ------------------------------------------------------------------ maxpool 3x3 / 2, pad 1
// Branch-free window: the nine 16-byte loads of a thread are issued together (clamped coordinates, out-of-image taps masked
// afterwards) -- with a `continue` per tap the loads went out one at a time and the kernel ran at 3.0 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo,
                                                      const float* __restrict__ scale, const float* __restrict__ shift) {
  constexpr int N = Vec16<T>::N;
  const int cv = C / N;
  const long total = (long)B * Ho * Wo * cv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * N;
    long t = idx / cv;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long b = t / Ho;
    float m[N], sc[N], sh[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { m[k] = -INFINITY; sc[k] = scale ? scale[c + k] : 1.f; sh[k] = scale ? shift[c + k] : 0.f; }
    Vec16<T> v[9];
    bool ok[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hi = ho * 2 - 1 + dh, hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int wi = wo * 2 - 1 + dw, wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
        ok[dh * 3 + dw] = hi >= 0 && hi < H && wi >= 0 && wi < W;
        v[dh * 3 + dw] = ld16<T>(x + ((b * H + hc) * (long)W + wc) * C + c);
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        float f = v[q].get(k);
        if (scale) f = fmaxf(f * sc[k] + sh[k], 0.f);
        m[k] = fmaxf(m[k], ok[q] ? f : -INFINITY);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int k = 0; k < N; ++k) o.set(k, m[k]);
    st16<T>(y + idx * N, o);
  }
}

// ------------------------------------------------------------------ global average pool
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int HW, int C) {
  constexpr int N = Vec16<T>::N;
  const int cv = C / N;
  const long total = (long)B * cv;
  const float inv = 1.f / (float)HW;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * N;
    const long b = idx / cv;
    float s[N];
#pragma unroll
    for (int k = 0; k < N; ++k) s[k] = 0.f;
    const T* p = x + b * HW * (long)C + c;
    for (int i = 0; i < HW; ++i) {
      Vec16<T> v = ld16<T>(p + (long)i * C);
#pragma unroll
      for (int k = 0; k < N; ++k) s[k] += v.get(k);
    }
    Vec16<T> o;
#pragma unroll
    for (int k = 0; k < N; ++k) o.set(k, s[k] * inv);
    st16<T>(y + b * C + c, o);
  }
}

// ------------------------------------------------------------------ transpose (+ column sums)
// 64x64 tiles through LDS; reads are row-contiguous, writes are row-contiguous in the output.
template <typename TI, typename TOUT>
__global__ void transpose_kernel(const TI* __restrict__ in, long ld_in, TOUT* __restrict__ out, long R, long C, long ld_out,
                                 float* colsum, float cs) {
  // out[c][r] = in[r][c] for r < R, 0 for R <= r < ld_out (zero padding of the reduction dimension)
  __shared__ float tile[64][65];
  __shared__ float part[4][64];
  const long tr = (long)blockIdx.y * 64, tc = (long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows at a time
  float csum = 0.f;
  for (int r = ty; r < 64; r += 4) {
    float v = 0.f;
    if (tr + r < R && tc + tx < C) v = to_f<TI>(in[(tr + r) * ld_in + tc + tx]);
    tile[r][tx] = v;
    csum += v;
  }
  part[ty][tx] = csum;
  __syncthreads();
  if (colsum && ty == 0 && tc + tx < C) atomicAdd(colsum + tc + tx, cs * (part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]));
  if (out) {
    for (int r = ty; r < 64; r += 4) {
      if (tc + r < C && tr + tx < ld_out) out[(tc + r) * ld_out + tr + tx] = from_f<TOUT>(tile[tx][r]);
    }
  }
}

// out[r][0..Cpad) = cast(in[r][0..C)) with zero fill of [C, Cpad)
template <typename TI, typename TOUT>
__global__ void cast_pad_kernel(const TI* __restrict__ in, long ld_in, TOUT* __restrict__ out, long ld_out, long R, long C,
                                long Cpad) {
  const long total = R * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Cpad, c = i - r * Cpad;
    out[r * ld_out + c] = from_f<TOUT>(c < C ? to_f<TI>(in[r * ld_in + c]) : 0.f);
  }
}

template <typename TI, typename TOUT>
__global__ void cast_kernel(const TI* __restrict__ in, TOUT* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f<TOUT>(to_f<TI>(in[i]));
}

// ------------------------------------------------------------------ dropout p = 0.5
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
template <typename T>
__global__ void dropout_half_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ mask, long nvec, long n,
                                    uint64_t seed) {
  constexpr int N = Vec16<T>::N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long e = i * N;
    // one 64-bit hash covers 64 consecutive elements: element e uses bit (e & 63) of hash(seed, e >> 6)
    const uint64_t bits = mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)((e >> 6) + 1));
    Vec16<T> v = ld16<T>(x + e), o;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const bool keep = (bits >> ((e + k) & 63)) & 1;
      o.set(k, keep ? 2.f * v.get(k) : 0.f);
      if (mask && e + k < n) mask[e + k] = keep;
    }
    st16<T>(y + e, o);
  }
}

}  // namespace

#define DT_SWITCH(dtype, EXPR)                        \
  if ((dtype) == SR_F32) { using T = float; EXPR; }   \
  else if ((dtype) == SR_BF16) { using T = bf16_t; EXPR; } \
  else return SR_ERR_DTYPE;

extern "C" int sr_abi_version(void) { return SR_ABI_VERSION; }

// The share is PER CALLING THREAD (the process default comes from SR_CU_SHARE): a thread that sizes its grids for half the chip
// -- FCGGNN.forward while its two backbone streams are being enqueued -- does not change the grids, tile shapes or partial-buffer
// sizes of launches issued by any other thread (the autograd engine's, a data loader's).
const int sr_cu_share_default = [] { const char* e = getenv("SR_CU_SHARE"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 8 ? v : 1; }();
thread_local int sr_cu_share_tls = 0;
extern "C" int sr_set_cu_share(int share) {
  if (share < 1 || share > 8) return SR_ERR_ARG;
  const int prev = sr_cu_share_tls > 0 ? sr_cu_share_tls : sr_cu_share_default;
  sr_cu_share_tls = share;
  return prev;
}

extern "C" int sr_stem_prep(const float* img, void* out, int B, int H, int W, int dtype, void* stream) {
  if (!img || !out || B <= 0 || H <= 0 || W <= 0) return SR_ERR_ARG;
  const int Hp = (H + 6 + 1) & ~1, Wp = (W + 6 + 1) & ~1;
  const long total = (long)B * Hp * Wp;
  DT_SWITCH(dtype, hipLaunchKernelGGL(stem_prep_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                      img, (T*)out, B, H, W, Hp, Wp));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_image_prep_u8(const uint8_t* img, void* out, int B, int H0, int W0, int H, int W, const int32_t* crop_yx,
                                const uint8_t* flip, const float* mean3, const float* std3, int dtype, void* stream) {
  if (!img || !out || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0 || H0 < H || W0 < W) return SR_ERR_ARG;
  if (!crop_yx && (H0 != H || W0 != W)) return SR_ERR_ARG;
  for (int c = 0; c < 3; ++c)
    if (!(std3[c] > 0.f)) return SR_ERR_ARG;
  const int Hp = (H + 6 + 1) & ~1, Wp = (W + 6 + 1) & ~1;
  const long total = (long)B * Hp * Wp;
  DT_SWITCH(dtype, hipLaunchKernelGGL(image_prep_u8_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, img,
                                      (T*)out, B, H0, W0, H, W, Hp, Wp, crop_yx, flip, mean3[0], mean3[1], mean3[2],
                                      1.f / std3[0], 1.f / std3[1], 1.f / std3[2]));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_bn_finalize(const float* stats, int tiles, int C, int64_t count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* scale,
                              float* shift, double* scratch, int scratch_rows, float* running_mean2, float* running_var2,
                              float momentum2, void* stream) {
  if (!stats || tiles <= 0 || C <= 0 || count <= 0 || !gamma || !beta || !scale || !shift || !scratch || scratch_rows < 1)
    return SR_ERR_ARG;
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  if (tiles <= 4096) {              // few partial rows: one kernel, fp64 sums straight from the fp32 rows (<= 256 per thread)
    hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, stats, tiles, C,
                       1.0 / (double)count, unbias, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                       running_mean2, running_var2, momentum2);
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  int chunks = (tiles + 31) / 32;   // stage A: >= 32 partial rows per workgroup, up to 1024 workgroups per 64 channels
  if (chunks > scratch_rows) chunks = scratch_rows;
  if (chunks > 1024) chunks = 1024;
  const int tpc = (tiles + chunks - 1) / chunks;
  chunks = (tiles + tpc - 1) / tpc;
  hipLaunchKernelGGL(bn_reduce_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, stats, tiles, C, tpc,
                     scratch);
  hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, (const double*)scratch, chunks, C,
                     1.0 / (double)count, unbias, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                     running_mean2, running_var2, momentum2);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_bn_apply(const void* x, const float* scale, const float* shift, const void* res, void* y, int64_t rows,
                           int C, int relu, int dtype, void* stream) {
  if (!x || !scale || !shift || !y || rows <= 0 || C <= 0) return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (C % n) return SR_ERR_ARG;
  const int cv = C / n, tpr = cv < 256 ? cv : 256, rpb = 256 / tpr;
  if (cv <= 256 && 256 % cv == 0 && rows * (long)cv >= 256L * 8 * 64) {     // flat contiguous walk (see bn_apply_flat_kernel)
    const long n16 = rows * (long)cv;
    long gf = (n16 + 2047) / 2048;
    const long cap = (long)sr_num_cus() * 8;
    if (gf > cap) gf = cap;
    DT_SWITCH(dtype, hipLaunchKernelGGL((bn_apply_flat_kernel<T, 8>), dim3((unsigned)gf), dim3(kThreads), 0, (hipStream_t)stream,
                                        (const T*)x, scale, shift, (const T*)res, (T*)y, n16, cv, relu));
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  long g = (rows + rpb - 1) / rpb;
  if (g > 256 * 16) g = 256 * 16;
  DT_SWITCH(dtype, hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)g), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)x, scale, shift, (const T*)res, (T*)y, (long)rows, C, relu));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, const float* scale, const float* shift,
                               int dtype, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((scale == nullptr) != (shift == nullptr))) return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (C % n) return SR_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / n);
  DT_SWITCH(dtype, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)x, (T*)y, B, H, W, C, Ho, Wo, scale, shift));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_avgpool(const void* x, void* y, int B, int HW, int C, int dtype, void* stream) {
  if (!x || !y || B <= 0 || HW <= 0 || C <= 0) return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (C % n) return SR_ERR_ARG;
  const long total = (long)B * (C / n);
  DT_SWITCH(dtype, hipLaunchKernelGGL(avgpool_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)x, (T*)y, B, HW, C));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

template <typename TI>
static int transpose_out(const void* in, long ld_in, void* out, long R, long C, long ld_out, int out_dtype, float* colsum,
                         float cs, hipStream_t st) {
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((ld_out + 63) / 64));
  if (out_dtype == SR_F32)
    hipLaunchKernelGGL((transpose_kernel<TI, float>), grid, dim3(256), 0, st, (const TI*)in, ld_in, (float*)out, R, C, ld_out, colsum, cs);
  else if (out_dtype == SR_BF16)
    hipLaunchKernelGGL((transpose_kernel<TI, bf16_t>), grid, dim3(256), 0, st, (const TI*)in, ld_in, (bf16_t*)out, R, C, ld_out, colsum, cs);
  else
    return SR_ERR_DTYPE;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_transpose(const void* in, int64_t ld_in, void* out, int64_t R, int64_t C, int64_t ld_out, int in_dtype,
                            int out_dtype, float* colsum, float colsum_scale, void* stream) {
  if (!in || (!out && !colsum) || R <= 0 || C <= 0 || ld_in < C || ld_out < R || (ld_out + 63) / 64 > 65535) return SR_ERR_ARG;
  if (in_dtype == SR_F32) return transpose_out<float>(in, ld_in, out, R, C, ld_out, out_dtype, colsum, colsum_scale, (hipStream_t)stream);
  if (in_dtype == SR_BF16) return transpose_out<bf16_t>(in, ld_in, out, R, C, ld_out, out_dtype, colsum, colsum_scale, (hipStream_t)stream);
  return SR_ERR_DTYPE;
}

extern "C" int sr_colsum(const void* in, int64_t ld_in, int64_t R, int64_t C, int dtype, float* colsum, float scale, void* stream) {
  if (!colsum) return SR_ERR_ARG;
  return sr_transpose(in, ld_in, nullptr, R, C, R, dtype, dtype, colsum, scale, stream);
}

extern "C" int sr_cast_pad(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, int64_t Cpad,
                           int in_dtype, int out_dtype, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || Cpad < C || ld_in < C || ld_out < Cpad) return SR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(R * Cpad)), b(kThreads);
  if (in_dtype == SR_F32 && out_dtype == SR_BF16)
    hipLaunchKernelGGL((cast_pad_kernel<float, bf16_t>), g, b, 0, st, (const float*)in, (long)ld_in, (bf16_t*)out, (long)ld_out, (long)R, (long)C, (long)Cpad);
  else if (in_dtype == SR_F32 && out_dtype == SR_F32)
    hipLaunchKernelGGL((cast_pad_kernel<float, float>), g, b, 0, st, (const float*)in, (long)ld_in, (float*)out, (long)ld_out, (long)R, (long)C, (long)Cpad);
  else if (in_dtype == SR_BF16 && out_dtype == SR_BF16)
    hipLaunchKernelGGL((cast_pad_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)in, (long)ld_in, (bf16_t*)out, (long)ld_out, (long)R, (long)C, (long)Cpad);
  else if (in_dtype == SR_BF16 && out_dtype == SR_F32)
    hipLaunchKernelGGL((cast_pad_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)in, (long)ld_in, (float*)out, (long)ld_out, (long)R, (long)C, (long)Cpad);
  else
    return SR_ERR_DTYPE;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream) {
  if (!in || !out || n <= 0) return SR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(n)), b(kThreads);
  if (in_dtype == SR_F32 && out_dtype == SR_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)in, (bf16_t*)out, (long)n);
  else if (in_dtype == SR_BF16 && out_dtype == SR_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)in, (float*)out, (long)n);
  else if (in_dtype == SR_F32 && out_dtype == SR_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)in, (float*)out, (long)n);
  else if (in_dtype == SR_BF16 && out_dtype == SR_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)in, (bf16_t*)out, (long)n);
  else
    return SR_ERR_DTYPE;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_dropout_half(const void* x, void* y, uint8_t* mask_out, int64_t n, uint64_t seed, int dtype, void* stream) {
  if (!x || !y || n <= 0) return SR_ERR_ARG;
  const int nv = dtype == SR_F32 ? 4 : 8;
  if (n % nv) return SR_ERR_ARG;
  const long nvec = n / nv;
  DT_SWITCH(dtype, hipLaunchKernelGGL(dropout_half_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)x, (T*)y, mask_out, nvec, (long)n, seed));
  SR_CHECK_LAUNCH();
  return SR_OK;
}
