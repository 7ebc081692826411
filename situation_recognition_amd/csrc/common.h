// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libsrhip.
// Wave = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srhip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define SR_CHECK_LAUNCH()                                      \
  do {                                                         \
    hipError_t e_ = hipGetLastError();                         \
    if (e_ != hipSuccess) return SR_ERR_LAUNCH;                \
  } while (0)

// ---- per-device host-side caches.  One process normally drives one GPU, but a process that moves on to a second device must
// not hand the first device's symbol addresses to kernels on the other one, nor assume its kernel attributes were set there.
constexpr int SR_MAX_DEV = 32;
inline int sr_cur_dev() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < SR_MAX_DEV) ? d : -1;
}
// device address of a __device__ variable on the CURRENT device (nullptr on failure)
#define SR_DEVICE_SYMBOL(var)                                                                 \
  ([]() -> void* {                                                                            \
    static void* cache_[SR_MAX_DEV] = {};                                                     \
    const int d_ = sr_cur_dev();                                                              \
    if (d_ < 0) return nullptr;                                                               \
    void* p_ = __atomic_load_n(&cache_[d_], __ATOMIC_ACQUIRE);                                \
    if (!p_) {                                                                                \
      if (hipGetSymbolAddress(&p_, HIP_SYMBOL(var)) != hipSuccess) p_ = nullptr;              \
      __atomic_store_n(&cache_[d_], p_, __ATOMIC_RELEASE);                                    \
    }                                                                                         \
    return p_;                                                                                \
  }())
// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (kernel, device)
template <auto Kernel>
inline bool sr_set_dynamic_lds(int bytes) {
  static unsigned done[SR_MAX_DEV] = {};
  const int d = sr_cur_dev();
  if (d < 0) return false;
  if (__atomic_load_n(&done[d], __ATOMIC_ACQUIRE)) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  __atomic_store_n(&done[d], 1u, __ATOMIC_RELEASE);
  return true;
}
// the same, keyed by a tag type (for kernels hipcc cannot take as a non-type template argument on the host side)
template <typename Tag>
inline bool sr_set_dynamic_lds_tagged(const void* kernel, int bytes) {
  static unsigned done[SR_MAX_DEV] = {};
  const int d = sr_cur_dev();
  if (d < 0) return false;
  if (__atomic_load_n(&done[d], __ATOMIC_ACQUIRE)) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  __atomic_store_n(&done[d], 1u, __ATOMIC_RELEASE);
  return true;
}
// sr_set_cu_share (elementwise.hip): the persistent kernels size their grids for 1/share of the chip, so that the launches of
// `share` concurrent streams co-reside on disjoint sets of CUs instead of queueing behind each other's full-chip grids.
// The value is thread-local (0 = the process default): everything derived from it -- grids, v3_cfg's tile choice, partial-statistics
// row counts, Gram slice counts -- is consistent between a query and the launch it sizes as long as both come from one thread.
extern const int sr_cu_share_default;
extern thread_local int sr_cu_share_tls;
inline int sr_num_cus() {
  static int cache[SR_MAX_DEV] = {};
  const int d = sr_cur_dev();
  const int share = sr_cu_share_tls > 0 ? sr_cu_share_tls : sr_cu_share_default;
  if (d < 0) return 256 / share;
  int n = __atomic_load_n(&cache[d], __ATOMIC_ACQUIRE);
  if (n <= 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cache[d], n, __ATOMIC_RELEASE);
  }
  return n / share;
}


// relu(x * scale + shift) on the eight bf16 values of a 16-byte chunk, rounded back to bf16.  `sp` / `hp`: the chunk's eight scales /
// shifts in PAIR ORDER -- channels [0, 2, 1, 3, 4, 6, 5, 7] (sr_pair_order) -- as two 16-byte vectors each, so that the packed f32
// operations (two channels per instruction: the low halves of two adjacent words, then their high halves) take their operands from
// adjacent registers as loaded, and v_cvt_pk_bf16_f32 puts (low, high) straight back into one word: 20 vector instructions per
// chunk (the scalar form compiled to ~60, half of them register moves).
typedef float sr_f32x2 __attribute__((ext_vector_type(2)));
typedef float sr_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sr_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sr_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sr_bf16x2 __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr int sr_pair_order(int i) { return (i & 4) | ((i & 1) << 1) | ((i & 2) >> 1); }   // position i of a chunk holds channel ...
__device__ __forceinline__ sr_u32x4 sr_affine_relu_chunk(sr_u32x4 v, sr_f32x4 s0, sr_f32x4 s1, sr_f32x4 h0, sr_f32x4 h1) {
  sr_u32x4 o;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const sr_f32x4 s = p ? s1 : s0, h = p ? h1 : h0;
    sr_f32x2 lo = {__uint_as_float(v[2 * p] << 16), __uint_as_float(v[2 * p + 1] << 16)};
    sr_f32x2 hi = {__uint_as_float(v[2 * p] & 0xffff0000u), __uint_as_float(v[2 * p + 1] & 0xffff0000u)};
    lo = __builtin_elementwise_fma(lo, sr_f32x2{s[0], s[1]}, sr_f32x2{h[0], h[1]});
    hi = __builtin_elementwise_fma(hi, sr_f32x2{s[2], s[3]}, sr_f32x2{h[2], h[3]});
    // ReLU AFTER the rounding, on the packed pair: a negative bf16 (including -0) is a negative int16, so max(., 0) as int16 is the
    // ReLU -- rounding is monotonic and keeps the sign, so round(relu(x)) == relu(round(x)) bit for bit (round 5: 20 vector
    // instructions per chunk instead of 24; every consumer that normalises on load and the Gram sweep share this one function).
    // (One difference from v_max_f32: a POSITIVE NaN stays a NaN -- as torch.relu keeps it -- where max(x, 0) returned 0.)
    const sr_bf16x2 w0 = __builtin_convertvector(sr_f32x2{lo[0], hi[0]}, sr_bf16x2), w1 = __builtin_convertvector(sr_f32x2{lo[1], hi[1]}, sr_bf16x2);
    unsigned u0 = __builtin_bit_cast(unsigned, w0), u1 = __builtin_bit_cast(unsigned, w1);
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(u0) : "v"(u0));
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(u1) : "v"(u1));
    o[2 * p] = u0;
    o[2 * p + 1] = u1;
  }
  return o;
}

// sr_conv_route (gemm.hip): a dry run of sr_conv2d's dispatch.  While sr_route_probe is set, every launch site records the
// code of the kernel it WOULD launch and returns instead of launching (codes: include/srhip.h, SR_ROUTE_*).
extern thread_local int sr_route_probe;
extern thread_local int sr_route_code;
#define SR_ROUTE(code) do { if (sr_route_probe) { sr_route_code = (code); return SR_OK; } } while (0)

// expand.hip: the output-heavy 1x1 convolutions (internal hand-over from sr_conv2d; SR_ERR_UNSUPPORTED = not one of its shapes)
int srx_conv1x1_expand(const sr_conv_args* a, long M, void* stream);
bool srx_conv1x1_in_affine_ok(const sr_conv_args* a, long M);
// stem.hip: the 7x7/2 stem as a direct convolution (bf16, 64 output channels); rows of partial statistics it writes
int srx_stem_conv(const sr_conv_args* a, void* stream);
int srx_stem_rows(const sr_conv_args* a);
// direct 3x3 convolution of the 64-channel layer (c3d.hip); same convention
int srx_c3d_conv(const sr_conv_args* a, void* stream);
int srx_c3d_rows(const sr_conv_args* a);
bool srx_c3d_in_affine_ok(const sr_conv_args* a);
// direct 3x3 convolutions over 32-channel patch slices (c3ds.hip); same convention
int srx_c3d256_conv(const sr_conv_args* a, void* stream);     // c3ds.hip: direct 3x3 over 32-channel patch slices, 256 channels, 14 x 14 images (layer3)
int srx_c3d256_rows(const sr_conv_args* a);
bool srx_c3d256_in_affine_ok(const sr_conv_args* a);
int srx_c3d128s_conv(const sr_conv_args* a, void* stream);    // c3ds.hip: the same kernel for 128 channels, 28 x 28 images (layer2)
int srx_c3d128s_rows(const sr_conv_args* a);
bool srx_c3d128s_in_affine_ok(const sr_conv_args* a);

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of elements of T (8 bf16 or 4 f32) with float views.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ float get(int i) const { return ((const float*)&raw)[i]; }
  __device__ __forceinline__ void set(int i, float v) { ((float*)&raw)[i] = v; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    uint32_t w = ((const uint32_t*)&raw)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ void set(int i, float v) { ((bf16_t*)&raw)[i] = (bf16_t)v; }
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
  Vec16<T> v;
  v.raw = *reinterpret_cast<const decltype(v.raw)*>(p);
  return v;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<decltype(v.raw)*>(p) = v.raw;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(1+e^{2x}); exact limits at +-inf, abs err ~1e-7
  return 1.0f - 2.0f / (1.0f + __expf(2.0f * x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
