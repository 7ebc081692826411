// FP8 (OCP e4m3) path for the matrix-bound 3x3 convolutions of the backbone (BASELINE config 5: "ResNet-152 fp8-weight MFMA").
//
// The reference's reduced-precision route is torch.cuda.amp.autocast on every model method (reference model.py:33,58,114,157,171);
// its CDNA4 counterpart here: the bottleneck's 3x3 convolution -- the only matrix-bound convolution family of the backbone
// (35 % of its FLOPs at 14x14 alone) -- takes e4m3 activations and e4m3 weights on v_mfma_scale_f32_16x16x128_f8f6f4 (block
// scales fixed at 2^0), which issues K = 128 per instruction at twice the cycles of the bf16 K = 32 form: 2x the bf16 rate.
// The non-scaled fp8 MFMA runs at the bf16 rate, so fp8 WEIGHTS alone would buy nothing.  Everything around it stays bf16.
//
//   * quantisation: activations e4m3(relu(bn(y)) * s_a) written by the BatchNorm-apply pass that precedes the conv
//     (sr_bn_apply_fp8: no extra sweep), weights e4m3(w * s_w[co]) per output channel (host side, once per weight version);
//     the fp32 accumulator is multiplied by dq[co] = 1 / (s_a * s_w[co]) before statistics and store;
//   * products of two e4m3 numbers are exact in fp32 and the accumulation is fp32: the result equals an fp32 convolution of
//     the dequantised operands up to summation order -- that is the oracle (tests/test_fp8_gpu.py);
//   * kernel: implicit GEMM, tile 256 x 128, 8 waves as 4 x 2 (64 x 64 each = 4 x 4 fragments), K-steps of 128 fp8 = 128-byte
//     rows through a 3-slot LDS ring filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, padding taps and row tails through
//     out-of-range offsets that write zeros), 128-byte-row XOR swizzle, counted vmcnt; epilogue: dequantise, running
//     BatchNorm sums per lane over the workgroup's tiles, bf16 staging strip per wave, 16-byte coalesced stores.
// LDS: 3 x 48 KiB ring + 8 x 2 KiB staging = 160 KiB.
#include <stdlib.h>

#include "common.h"

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct F8Args {
  const unsigned char* x;      // [B, H, W, Cin] e4m3
  const unsigned char* w;      // [Cout][3][3][Cin] e4m3
  const float* dq;             // [Cout]
  bf16_t* y;                   // [M, Cout]
  float* stats;                // [rows][2][Cout] or null
  int M, N, Cin, H, W, Ho, Wo, stride, nkt, kpt, lgkpt;   // nkt = 9 * Cin / 128 K-steps per tile, kpt = Cin / 128 steps per tap
};

constexpr int FSLOT = 49152;     // A 256 x 128 B, then W 128 x 128 B
constexpr int FOOB = (int)0x80000000;
constexpr int FNREC = 0x7ffff000;

template <int N> __device__ __forceinline__ void fwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float frow16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

__device__ __forceinline__ void conv3x3_fp8_body(const F8Args& p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fgrp = lane >> 4;

  const int gn = p.N >> 7;
  const int gm = (p.M + 255) >> 8;
  const int G = gridDim.x, groups = G / gn;                 // the host makes G a multiple of gn
  const int tn = blockIdx.x % gn, grp = blockIdx.x / gn;    // this workgroup's column tile is fixed (running statistics)
  const int my_tiles = grp < gm ? (gm - grp + groups - 1) / groups : 0;

  // ---------------- loader ----------------
  const int prow = lane >> 3;                               // row inside an 8-row piece
  const int csrc = ((lane & 7) ^ prow) << 4;                // source byte offset of the 16-byte chunk this lane stores (swizzle)
  int a_vo[4];
  unsigned a_mask[4];
  __amdgpu_buffer_rsrc_t srd_a;
  const __amdgpu_buffer_rsrc_t srd_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long)tn * 128 * 9 * p.Cin), 0, FNREC, 0x00020000);
  int w_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_vo[i] = ((wave + i * 8) * 8 + prow) * 9 * p.Cin + csrc;
  auto setup_a = [&](int tm, bool valid) {
    const long m0 = (long)tm * 256;
    auto pixel = [&](long m) -> long {
      const unsigned hw = (unsigned)(p.Ho * p.Wo), um = (unsigned)m;
      const long b = um / hw;
      const int rem = (int)(um - (unsigned)b * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      return (b * p.H + (ho * p.stride - 1)) * (long)p.W + (wo * p.stride - 1);
    };
    const long pix0 = pixel(m0);
    srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + pix0 * p.Cin), 0, valid ? FNREC : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + (wave + i * 8) * 8 + prow;
      if (m >= p.M) { a_vo[i] = FOOB; a_mask[i] = 0; continue; }
      const unsigned hw = (unsigned)(p.Ho * p.Wo), um = (unsigned)m;
      const long b = um / hw;
      const int rem = (int)(um - (unsigned)b * hw);
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ho * p.stride - 1, wi0 = wo * p.stride - 1;
      const long pix = (b * p.H + hi0) * (long)p.W + wi0;
      a_vo[i] = (int)((pix - pix0) * p.Cin) + csrc;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hi = hi0 + t / 3, wi = wi0 + t % 3;
        if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  };
  auto issue = [&](int kstep, int slot) {
    const int tap = kstep >> p.lgkpt, dh = (tap * 11) >> 5, dw = tap - dh * 3;       // (t*11)>>5 == t/3 for t < 9
    const int aoff = (dh * p.W + dw) * p.Cin + ((kstep & (p.kpt - 1)) << 7);
    const int woff = kstep << 7;
    char* const base = smem + slot * FSLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vo = ((a_mask[i] >> tap) & 1) ? a_vo[i] : FOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (__attribute__((address_space(3))) void*)(base + (wave + i * 8) * 1024), 16, vo, aoff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (__attribute__((address_space(3))) void*)(base + 32768 + (wave + i * 8) * 1024), 16, w_vo[i], woff, 0, 0);
  };

  // ---------------- fragments: lane -> row (lane & 15), k-group (lane >> 4) = 32 bytes = chunks 2g, 2g+1 of the swizzled row ----------------
  const int sw0 = ((2 * fgrp) ^ (lane & 7)) << 4, sw1 = ((2 * fgrp + 1) ^ (lane & 7)) << 4;
  const int a_base = (wm * 64 + frow) * 128, b_base = 32768 + (wn * 64 + frow) * 128;
  auto frag = [&](const char* sl, int rowoff) -> v8i_t {
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(sl + rowoff + sw0);
    const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(sl + rowoff + sw1);
    v8i_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float dq[4][4], s1[4][4], s2[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 d = *reinterpret_cast<const float4*>(p.dq + tn * 128 + wn * 64 + j * 16 + fgrp * 4);
    dq[j][0] = d.x; dq[j][1] = d.y; dq[j][2] = d.z; dq[j][3] = d.w;
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
  }

  char* const stg = smem + 3 * FSLOT + wave * 2048;   // per wave: 16 rows x 64 columns bf16; 16-byte chunk c of row r at c ^ (r & 7)
  const int rrow = lane >> 3, rq8 = lane & 7;

  if (my_tiles > 0) {
    setup_a(grp, true);
    issue(0, 0);
    issue(1, 1);
  }
  fwait_vm<0>();
  __syncthreads();
  int slot_c = 0, slot_i = 2;
  for (int it = 0; it < my_tiles; ++it) {
    const int tm = grp + it * groups;
    // ---------------- K loop: one barrier per step, two steps of LDS-DMA in flight ----------------
    for (int s = 0; s < p.nkt; ++s) {
      // my pieces of step s: everything older than the 6 pieces of step s+1 (and, in the first two steps behind an epilogue, its 8 stores)
      if (it > 0 && s < 2) fwait_vm<14>(); else fwait_vm<6>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + 2 == p.nkt) setup_a(tm + groups, it + 1 < my_tiles);          // the loader crosses into the next tile
      issue(s + 2 < p.nkt ? s + 2 : s + 2 - p.nkt, slot_i);
      slot_i = slot_i == 2 ? 0 : slot_i + 1;
      const char* sl = smem + slot_c * FSLOT;
      v8i_t fw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = frag(sl, b_base + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v8i_t fa = frag(sl, a_base + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[j], fa, acc[j][i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      slot_c = slot_c == 2 ? 0 : slot_c + 1;
    }
    // ---------------- epilogue: dequantise, statistics, bf16 through the staging strip, 2 stores per strip ----------------
    // (stores go through a buffer descriptor whose range ends with the tile's last valid row: rows past M are dropped by the
    //  range check but every store instruction is ISSUED -- the counted waits above rely on 8 per tile)
    const long t0 = (long)tm * 256;
    const long rows_ok = (long)p.M - t0 < 256 ? (long)p.M - t0 : 256;
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + t0 * p.N + tn * 128), 0,
                                                                           (int)(((rows_ok - 1) * p.N + 128) * 2), 0x00020000);
    const int o_vo = ((wm * 64 + rrow) * p.N + wn * 64 + rq8 * 8) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[j][i][r] * dq[j][r];
          s1[j][r] += v[r];
          s2[j][r] = __builtin_fmaf(v[r], v[r], s2[j][r]);
          acc[j][i][r] = 0.f;
        }
        bf16_t pk[4] = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        // (inline asm: a visible LDS store makes hipcc put s_waitcnt vmcnt(0) in front of it while LDS-DMA is in flight, which drains
        //  the two steps already fetched for the next tile; the strip is private to the wave and a wave's LDS operations execute in order)
        asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(uintptr_t)(stg + frow * 128 + (((j * 2 + (fgrp >> 1)) ^ (frow & 7)) << 4) + (fgrp & 1) * 8)),
                     "v"(*reinterpret_cast<const u32x2_t*>(pk))
                     : "memory");
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r16 = h * 8 + rrow;
        const u32x4_t val = *reinterpret_cast<const u32x4_t*>(stg + r16 * 128 + ((rq8 ^ (r16 & 7)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(val, srd_o, o_vo, (i * 16 + h * 8) * p.N * 2, 0);
      }
    }
  }
  fwait_vm<0>();
  if (p.stats) {
    // every (group, wave row) writes its partial row in full for its 128 columns (zeros if it had no tile)
    float* const row = p.stats + ((long)(grp * 4 + wm) * 2) * p.N + tn * 128 + wn * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = frow16_sum(s1[j][r]); b[r] = frow16_sum(s2[j][r]); }
      if (frow == 0) {
        *reinterpret_cast<float4*>(row + j * 16 + fgrp * 4) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(row + p.N + j * 16 + fgrp * 4) = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
  }
}

__global__ __launch_bounds__(512, 2) void conv3x3_fp8_kernel(const F8Args p) { conv3x3_fp8_body(p); }

struct F8Tag {};

// ---------------- elementwise: quantise to e4m3 ----------------
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}

// out[i] = e4m3( f(x[i]) * act_scale ), f = relu(x * scale[c] + shift[c]) when scale != null, identity otherwise.
// Flat walk (as bn_apply_flat_kernel): a lane owns groups of 16 consecutive elements (two 16-byte loads of bf16, one 16-byte store
// of e4m3), a workgroup walks contiguous chunks of 256 x U groups, all of a lane's groups belong to the same 16 channels
// (C / 16 divides 256), whose scale / shift stay in registers.
template <int U>
__global__ __launch_bounds__(256) void quantize_fp8_flat_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, unsigned char* __restrict__ out, long ngroups,
                                                                int cg_per_row, int relu, float act_scale) {
  const int cg = threadIdx.x % cg_per_row;
  float sc[16], sh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { sc[k] = scale ? scale[cg * 16 + k] * act_scale : act_scale; sh[k] = scale ? shift[cg * 16 + k] * act_scale : 0.f; }
  const long chunk = 256L * U;
  for (long c0 = (long)blockIdx.x * chunk; c0 < ngroups; c0 += (long)gridDim.x * chunk) {
    const long base = c0 + threadIdx.x;
    uint4 lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 < ngroups) {
        const uint4* p = reinterpret_cast<const uint4*>(x + (base + u * 256) * 16);
        lo[u] = p[0]; hi[u] = p[1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u * 256 < ngroups) {
        const unsigned wds[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
        float v[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[2 * k] = __builtin_fmaf(__uint_as_float(wds[k] << 16), sc[2 * k], sh[2 * k]);
          v[2 * k + 1] = __builtin_fmaf(__uint_as_float(wds[k] & 0xffff0000u), sc[2 * k + 1], sh[2 * k + 1]);
        }
        if (relu) {
#pragma unroll
          for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        uint4 o;
        o.x = pack4_e4m3(v[0], v[1], v[2], v[3]); o.y = pack4_e4m3(v[4], v[5], v[6], v[7]);
        o.z = pack4_e4m3(v[8], v[9], v[10], v[11]); o.w = pack4_e4m3(v[12], v[13], v[14], v[15]);
        *reinterpret_cast<uint4*>(out + (base + u * 256) * 16) = o;
      }
    }
  }
}

// generic form (any C % 8 == 0, fp32 input): 8 elements per lane
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           unsigned char* __restrict__ out, long n8, int C, int relu, float act_scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long e = i * 8;
    const int c = (int)(e % C);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = to_f<T>(x[e + k]);
      if (scale) f = f * scale[c + k] + shift[c + k];
      if (relu) f = fmaxf(f, 0.f);
      v[k] = f * act_scale;
    }
    uint2 o;
    o.x = pack4_e4m3(v[0], v[1], v[2], v[3]);
    o.y = pack4_e4m3(v[4], v[5], v[6], v[7]);
    *reinterpret_cast<uint2*>(out + e) = o;
  }
}

}  // namespace

extern "C" int sr_conv3x3_fp8_stats_rows(int M, int N) {
  if (M <= 0 || N <= 0 || (N & 127)) return SR_ERR_ARG;
  const long gm = ((long)M + 255) / 256, gn = N / 128, cus = sr_num_cus();
  long G = gm * gn < cus ? gm * gn : cus;
  G -= G % gn;
  if (G < gn) G = gn;
  return (int)(G / gn * 4);
}

extern "C" int sr_conv3x3_fp8(const void* x, const void* w, const float* dq, void* y, float* stats, int B, int H, int W, int Cin, int Cout,
                              int stride, void* stream) {
  if (!x || !w || !dq || !y || B <= 0 || H <= 0 || W <= 0) return SR_ERR_ARG;
  if ((Cin != 128 && Cin != 256 && Cin != 512) || Cout % 128 || Cout <= 0 || (stride != 1 && stride != 2)) return SR_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) || ((uintptr_t)dq & 15)) return SR_ERR_ARG;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const long M = (long)B * Ho * Wo;
  if (M <= 0 || M > 0x7fffffffL) return SR_ERR_ARG;
  F8Args a;
  a.x = (const unsigned char*)x; a.w = (const unsigned char*)w; a.dq = dq; a.y = (bf16_t*)y; a.stats = stats;
  a.M = (int)M; a.N = Cout; a.Cin = Cin; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.stride = stride;
  a.kpt = Cin / 128; a.nkt = 9 * a.kpt; a.lgkpt = a.kpt == 1 ? 0 : (a.kpt == 2 ? 1 : 2);
  const long gm = (M + 255) / 256, gn = Cout / 128, cus = sr_num_cus();
  long G = gm * gn < cus ? gm * gn : cus;
  G -= G % gn;
  if (G < gn) G = gn;
  constexpr int LDS = 3 * FSLOT + 8 * 2048;
  if (!sr_set_dynamic_lds_tagged<F8Tag>(reinterpret_cast<const void*>(&conv3x3_fp8_kernel), LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL(conv3x3_fp8_kernel, dim3((unsigned)G), dim3(512), LDS, (hipStream_t)stream, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_quantize_fp8(const void* x, const float* scale, const float* shift, void* out, int64_t rows, int C, int relu, float act_scale,
                               int dtype, void* stream) {
  if (!x || !out || rows <= 0 || C <= 0 || (C & 7) || ((scale == nullptr) != (shift == nullptr))) return SR_ERR_ARG;
  if (((uintptr_t)out & 7)) return SR_ERR_ARG;
  const int cgpr = C / 16;
  if (dtype == SR_BF16 && C % 16 == 0 && cgpr <= 256 && 256 % cgpr == 0 && !((uintptr_t)x & 15) && !((uintptr_t)out & 15)) {
    const long ng = rows * (long)cgpr;
    long gf = (ng + 1023) / 1024;
    const long capf = (long)sr_num_cus() * 8;
    if (gf > capf) gf = capf;
    hipLaunchKernelGGL((quantize_fp8_flat_kernel<4>), dim3((unsigned)gf), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift,
                       (unsigned char*)out, ng, cgpr, relu, act_scale);
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  const long n8 = rows * (long)C / 8;
  long g = (n8 + 255) / 256;
  const long cap = (long)sr_num_cus() * 16;
  if (g > cap) g = cap;
  if (dtype == SR_BF16)
    hipLaunchKernelGGL(quantize_fp8_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift,
                       (unsigned char*)out, n8, C, relu, act_scale);
  else if (dtype == SR_F32)
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift,
                       (unsigned char*)out, n8, C, relu, act_scale);
  else return SR_ERR_DTYPE;
  SR_CHECK_LAUNCH();
  return SR_OK;
}
