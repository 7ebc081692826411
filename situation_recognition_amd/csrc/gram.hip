// Train-mode BatchNorm statistics of a 1x1 convolution WITHOUT running the convolution.
//
// For y = x W^T (x: [M, C] NHWC pixels, W: [N, C]) the per-output-channel batch sums are
//     sum_m y[m,n]   = W[n,:] . colsum(x)
//     sum_m y[m,n]^2 = W[n,:] G W[n,:]^T        with the C x C Gram matrix  G = x^T x.
// A bottleneck expansion conv has N = 4C, so G costs a quarter of the conv's MFMA work and reads x once; the
// statistics-only conv launch it replaces ("two-launch BatchNorm") recomputed the whole product.
//
//   gram_kernel<P,TWO>  x -> per-(row slice, k-split) fp32 partial Gram blocks + column sums (MFMA, LDS-DMA ring)
//   gram_reduce_kernel  fp32 partials -> fp64 (two deterministic stages, the layout bn_reduce uses)
//   gram_project_kernel fp64 quadratic forms per output channel, then scale/shift/EMA exactly like bn_finalize_kernel
//
// The Gram sum runs over the pixel index m, so both MFMA operands are the SAME transposed fragment of x
// ("8 consecutive pixels of one channel per lane"): it is read straight out of the row-major LDS image with
// ds_read_b64_tr_b16, and because A and B use one k-order any pixel permutation inside a fragment is harmless.
#include <type_traits>

#include "common.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_gram_zero[256];
__device__ __attribute__((aligned(256))) unsigned char g_gram_trash[512 * 16];   // sink for the stores of rows past the slice (fixed store count per stage: vmcnt)

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct GramArgs {
  const bf16_t* x;     // [M, ldx]
  long M, ldx;
  // TN GEMM mode (gram_kernel<256, true, true>): panel B comes from a second matrix, the block grid is nqa x nqb, the
  // output block (a, b) is stored at rows b*256.., columns a*256.. of a [nqb*256][ldo] partial, no column sums
  const bf16_t* xb;
  long ldb, ldo;
  int nqa, nqb;
  // FUSE mode: x holds RAW conv output; every stage is normalised in place (x = relu(x*scale[c] + shift[c]), in LDS and
  // in global memory) before it enters the Gram sums -- bn_apply and gram in one pass over the tensor
  const float* scale; const float* shift;
  void* trash;
  int C;
  float* partials;     // [nslices * KS][C*C + C]
  long pstride;        // C*C + C
  long rows_per_wg;    // multiple of the stage height
  const void* zero;    // 256 zero bytes
};

template <int N> __device__ __forceinline__ void gram_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XOR applied to the 16-byte chunk index of LDS row `row` (even values: a transposed read takes chunk pairs).
// The 8 rows one 32-lane half touches in a ds_read_b64_tr_b16 (row bits 0,1 and 3 vary) land on 8 distinct
// 32-byte bank groups.  64-column panels have 128-byte rows, where row bit 0 already selects the bank half.
template <int P> __device__ __forceinline__ int gram_swz(int row) {
  if (P >= 128) return 2 * ((row & 3) | (((row >> 3) & 1) << 2));
  return 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}

// P: panel width (channels a workgroup's Gram block spans per side).  TWO: C = 2P, the block's row and column
// panels differ and are staged separately.  8 waves = RG row groups (32 channels of the A side each) x KS k-splits
// (32-pixel sub-chunks of a stage); every stage is 16 KB per panel.
template <int P, bool TWO, bool TN = false, bool FUSE = false, bool KEEP = true>
__global__ __launch_bounds__(512, 1) void gram_kernel(const GramArgs p) {
  static_assert(!TN || (TWO && P == 256), "TN GEMM mode: separate 256-column panels");
  static_assert(!FUSE || (!TWO && !TN), "fused BatchNorm apply: single panel only");
  constexpr int SPS = (FUSE && KEEP) ? 2 : 0;     // stores per lane and stage (KEEP: the normalised stage is written back over x)
  constexpr bool SPLIT = FUSE && !KEEP;           // see the L section of the loop
  constexpr int KS = 256 / P, RG = P / 32, FB = P / 16, SR = 32 * KS;
  constexpr int CPRW = P / 8, ROWB = P * 2, STAGE = SR * ROWB;
  constexpr int NPAN = TWO ? 2 : 1, NS = TWO ? 4 : 8, IPS = 2 * NPAN;
  constexpr int SLOT = STAGE * NPAN;
  static_assert(STAGE == 16384, "stage");
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: everything derived from it stays in SGPRs
  // SYM (one panel, G = x^T x): only the 16 x 16 blocks on or below the block diagonal of the stored matrix -- B-side fragment
  // j >= A-side fragment -- are computed (136 of 256 fragment products at P = 256) and stored; the reduction mirrors them.  A
  // wave's two A-side fragments are rg and FA-1-rg (not 2rg, 2rg+1), so that every wave has FA+1 products per stage: the M
  // sections of the ping-pong loop are barrier-synchronised and as long as their slowest wave's.
  constexpr bool SYM = !TWO && !TN;
  constexpr int FA = P / 16;
  const int rg = wave % RG, ks = wave / RG;
  const int fa0 = SYM ? rg : 2 * rg, fa1 = SYM ? FA - 1 - rg : 2 * rg + 1;
  const int npa = TN ? p.nqa : (TWO ? 2 : 1), npb = TN ? p.nqb : npa;
  const int quad = blockIdx.x % (npa * npb);
  const long slice = blockIdx.x / (npa * npb);
  const int colA0 = (quad / npb) * P, colB0 = (quad % npb) * P;
  const bool diag = !TN && colA0 == colB0;
  const long r0 = slice * p.rows_per_wg;
  const long r1 = min(p.M, r0 + p.rows_per_wg);
  const int nst = (int)((r1 - r0 + SR - 1) / SR);

  // ---- loader (round 3: buffer loads with SCALAR stage offsets).  Chunk q = j*512 + tid of a stage goes to LDS position q*16 and
  // holds data chunk (q % CPRW) ^ swz(row) of stage row q / CPRW.  The slice is one buffer descriptor (base = its first row and
  // the panel's first column, num_records = up to the end of its last row's panel segment); a lane's byte offset inside a stage
  // is loop invariant (vo), the stage's offset is a scalar that advances by SR rows.  Rows past the slice and whole stages past
  // its end fall outside num_records: the load then writes ZEROS into LDS (checked on hardware, tools/ubench/buffer_lds_oob.hip),
  // so the issue count per stage is fixed and nothing is selected per lane.  The first form of this loader -- global_load_lds
  // with a 64-bit per-lane pointer, a zero page for the rows past the end, per-fragment exec-mask branches for the SYM skip
  // -- spent 96 scalar + 50 vector instructions per stage on bookkeeping: the L section of the ping-pong loop was 3x its M
  // section and the sweep ran at 3.1-3.8 TB/s with the matrix pipe 30 % busy (profiles/r03/pmc_gram_reduce.txt).
  const int pitchA = (int)(p.ldx * 2), pitchB = TN ? (int)(p.ldb * 2) : pitchA;
  int vo[2], vob[2], src_row[2];
  long src_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = j * 512 + tid, row = q / CPRW, slot = q % CPRW;
    src_row[j] = row;
    src_off[j] = (long)row * p.ldx + ((slot ^ gram_swz<P>(row)) * 8);        // (elements; KEEP's write-back address)
    vo[j] = row * pitchA + ((slot ^ gram_swz<P>(row)) * 16);
    vob[j] = row * pitchB + ((slot ^ gram_swz<P>(row)) * 16);
  }
  const long nrows = r1 - r0;
  const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + r0 * p.ldx + colA0), 0,
                                                                          (int)((nrows - 1) * pitchA + P * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t srdB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(TN ? p.xb + r0 * p.ldb + colB0 : p.x + r0 * p.ldx + colB0), 0, (int)((nrows - 1) * pitchB + P * 2), 0x00020000);
  auto issue = [&](int st) {
    char* dst = smem + (st % NS) * SLOT + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srdA, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, vo[j], st * (SR * pitchA), 0, 0);
      if (TWO)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdB, (__attribute__((address_space(3))) void*)(dst + STAGE + j * 8192), 16, vob[j],
                                                 st * (SR * pitchB), 0, 0);
    }
  };

  // ---- FUSE: a lane normalises the two 16-byte chunks IT loaded of a stage (they hold the same 8 channels: the chunk
  // index of q = j*512 + tid does not depend on j), in LDS and in global memory.  Rows past the slice stay zero.
  // (scale / shift live in LDS behind the ring, 2 x P floats: 16 more loop-carried registers made the 256-wide kernel spill)
  float* const lds_sc = reinterpret_cast<float*>(smem + NS * SLOT);
  const int c8 = ((tid % CPRW) ^ gram_swz<P>(tid / CPRW)) * 8;
  if (FUSE) {
    if (tid < P) {                       // (pair order inside every 8-channel chunk: sr_affine_relu_chunk)
      const int ch = (tid & ~7) | sr_pair_order(tid & 7);
      lds_sc[tid] = p.scale[ch]; lds_sc[P + tid] = p.shift[ch];
    }
    __syncthreads();
  }
  // (scale / shift of this lane's 8 channels: loop-carried registers -- 16 of them; the round-2 form re-read them from LDS in every
  //  call because the kernel was at 214 registers; the scalar loader of round 3 freed ~30)
  sr_f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, h0 = s0, h1 = s0;
  if (FUSE) {
    s0 = *reinterpret_cast<const sr_f32x4*>(lds_sc + c8); s1 = *reinterpret_cast<const sr_f32x4*>(lds_sc + c8 + 4);
    h0 = *reinterpret_cast<const sr_f32x4*>(lds_sc + P + c8); h1 = *reinterpret_cast<const sr_f32x4*>(lds_sc + P + c8 + 4);
  }
  auto normalise_half = [&](int st, int j) {
    char* img = smem + (st % NS) * SLOT + wave * 1024 + lane * 16;
    const long R = r0 + (long)st * SR;
    const bool ok = st * SR + src_row[j] < (int)nrows;                                              // (implies st < nst)
    const sr_u32x4 n = sr_affine_relu_chunk(*reinterpret_cast<const sr_u32x4*>(img + j * 8192), s0, s1, h0, h1);
    const sr_u32x4 vz = ok ? n : sr_u32x4{0u, 0u, 0u, 0u};                                          // rows past the slice stay zero
    // (inline asm: behind a plain LDS store hipcc puts `s_waitcnt vmcnt(0)` -- the store might alias a pending LDS-DMA -- which
    //  drains the ring once per stage, like the transposed reads below; the caller's `s_waitcnt lgkmcnt(0)` covers it)
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)(img + j * 8192)), "v"(vz) : "memory");
    // KEEP: exactly one store per piece (rows past the slice: the trash page), so that the vmcnt arithmetic below holds
    if (KEEP) *reinterpret_cast<uint4*>(ok ? (char*)(const_cast<bf16_t*>(p.x) + R * p.ldx + src_off[j]) : (char*)p.trash + tid * 16) = make_uint4(vz[0], vz[1], vz[2], vz[3]);
  };
  auto normalise = [&](int st) { normalise_half(st, 0); normalise_half(st, 1); };

  // ---- transposed-read addressing: lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) supplies row 8g+q (+4 for the
  // second read), 8-byte piece pp of the block's 32-byte row segment
  const int g = lane >> 4, tq = (lane >> 2) & 3, pp = lane & 3;
  const int row0 = 32 * ks + 8 * g + tq;
  const int fh = gram_swz<P>(row0) >> 1;
  const int rd_base = row0 * ROWB + 16 * (pp >> 1) + 8 * (pp & 1);

  // The transposed reads are INLINE ASM on purpose: behind the builtin (__builtin_amdgcn_ds_read_tr16_b64) hipcc (ROCm 7.2) puts an
  // `s_waitcnt vmcnt(0)` in front of the first read of every stage -- it treats the read as a possible reader of the pending LDS-DMA
  // writes -- which drains the whole 7-stage ring once per stage: the waves spent 40-50 % of their cycles parked there (SQ_WAIT_ANY,
  // profiles/r03/pmc_gram_reduce.txt) and the sweep was latency-bound.  The asm form is invisible to that pass; the ring is retired by
  // the counted gram_wait_vm<> waits alone.  Consequence: the compiler does not count these reads in lgkmcnt either, so every
  // fragment is pinned behind the explicit `s_waitcnt lgkmcnt(0)` of the L section with frag_ready() before an MFMA may use it.
  auto frag = [&](const char* img, int cb) -> bf16x8_t {
    const unsigned a = (unsigned)(uintptr_t)(img + rd_base + 32 * (cb ^ fh));        // (LDS offset: low 32 bits of the flat address)
    s16x4_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(4 * ROWB));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto frag_ready = [&](bf16x8_t& f) { asm volatile("" : "+v"(f)); };

  f32x4_t acc[2][FB], cs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    cs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

  // ---- pipeline: NS-1 stages in flight; issues past the slice's end fall outside the descriptor (zeros) so the count stays fixed.
  // The two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run half a stage apart, as in the GEMM kernel: an L
  // section (all transposed fragment reads of the stage, the DMA of stage it+NS-1, wait for the reads) and an M section
  // (the stage's MFMAs, no memory instruction), each closed by a barrier; group 1 starts one barrier late, so one wave of
  // a SIMD reads LDS while the other multiplies (lock-step, the 36 reads and 34 MFMAs of a stage ran one after the other:
  // 3500 cycles per stage for ~1100 of matrix-pipe time).  Barrier instances pair as G0.B1(it) = G1.B2(it-1),
  // G0.B2(it) = G1.B1(it).  Stage it is read in L(it); every wave has waited for its own pieces of it before the last
  // instance both groups pass ahead of that (G0: end of M(it-1); G1: in L(it-1)); the slot refilled in L(it) is the one
  // stage it-1 was read from, drained (lgkmcnt 0) by both groups before their B1(it-1).
  // SYM: the loop is instantiated once per row group (RGV compile time, selected by a scalar switch), so which fragment products
  // exist is known to the compiler: no predication inside the stage.
  const int grp = wave >> 2;
  auto run = [&](auto RGC) {
    constexpr int RGV = decltype(RGC)::value;
    const int f0 = SYM ? RGV : fa0, f1 = SYM ? FA - 1 - RGV : fa1;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    gram_wait_vm<(NS - 2) * IPS>();        // my pieces of stage 0
    if (FUSE) {
      // a stage is normalised by its loaders right after their own wait for it, at the end of an M section (the fragment
      // registers are dead there): group 0 does stage it+1 there, group 1 -- whose wait sits a barrier earlier -- stage it+2
      normalise(0);                                       // (+2 stores, younger than every DMA so far)
      if (grp == 1) { gram_wait_vm<(NS - 3) * IPS + SPS>(); normalise(1); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nst; ++it) {
      const char* imgA = smem + (it % NS) * SLOT;
      const char* imgB = TWO ? imgA + STAGE : imgA;
      bf16x8_t a[2], b[FB];
      a[0] = frag(imgA, f0);
      a[1] = frag(imgA, f1);
#pragma unroll
      for (int j = 0; j < FB; ++j)
        if (!SYM || (j > f0 && j != f1)) b[j] = frag(imgB, j);      // (one panel: B-side fragments f0, f1 ARE the A-side fragments)
      issue(it + NS - 1);                  // refills the slot stage it-1 used
      if (!FUSE && grp == 1) gram_wait_vm<(NS - 2) * IPS>();      // my pieces of stage it+1
      if (SPLIT) {
        // no write-back (no stores in the vmcnt order): the normalisation of a stage is split over the L and the M section --
        // chunk 0 here, behind the fragment reads (their latency covers it), chunk 1 behind the MFMAs -- so that the two sections
        // the barriers pair (one group's L beside the other's M) are about equally long.  Group 0 works on stage it+1, group 1,
        // whose sections sit half a stage later, on stage it+2; a lane only ever touches the chunks it loaded itself.
        if (grp == 0) { gram_wait_vm<(NS - 2) * IPS>(); normalise_half(it + 1, 0); }
        else { gram_wait_vm<(NS - 3) * IPS>(); normalise_half(it + 2, 0); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      frag_ready(a[0]); frag_ready(a[1]);
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        if (SYM && j == f0) b[j] = a[0];                           // (register copies only AFTER the wait: the compiler does not know the
        else if (SYM && j == f1) b[j] = a[1];                      //  asm reads are pending)
        else if (!SYM || j > f0) frag_ready(b[j]);
      }
      __builtin_amdgcn_s_barrier();        // B1
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        if (!SYM || j >= f0) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[j], acc[0][j], 0, 0, 0);
        if (!SYM || j >= f1) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[j], acc[1][j], 0, 0, 0);
      }
      if (!TN) {
        cs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], ones, cs[0], 0, 0, 0);
        cs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], ones, cs[1], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      if (SPLIT) {
        normalise_half(grp == 0 ? it + 1 : it + 2, 1);           // (landed: waited for in this iteration's L section)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (FUSE) {
        // vmcnt counts the 2 stores of every normalise() too, in issue order with the DMAs (one DMA pair per L section, one
        // store pair per M section): younger than the pieces of stage it+1 are 6 DMA stages + 6 store pairs (group 0),
        // than those of stage it+2 5 + 5 (group 1).  In the first iterations fewer store pairs have been issued yet, so
        // FEWER operations are younger: there the count without any stores is used (stricter, always safe).
        if (grp == 0) {
          if (it >= NS - 3) gram_wait_vm<(NS - 2) * (IPS + SPS)>(); else gram_wait_vm<(NS - 2) * IPS>();
          normalise(it + 1);
        } else {
          if (it >= NS - 5) gram_wait_vm<(NS - 3) * (IPS + SPS)>(); else gram_wait_vm<(NS - 3) * IPS>();
          normalise(it + 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my LDS writes are in place before the barrier
      } else if (grp == 0) {
        gram_wait_vm<(NS - 2) * IPS>();    // my pieces of stage it+1
      }
      __builtin_amdgcn_s_barrier();        // B2
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  };
  if constexpr (SYM) {
    switch (rg) {
      case 0: run(std::integral_constant<int, 0>{}); break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: if constexpr (RG > 2) run(std::integral_constant<int, 2>{}); break;
      case 3: if constexpr (RG > 2) run(std::integral_constant<int, 3>{}); break;
      case 4: if constexpr (RG > 4) run(std::integral_constant<int, 4>{}); break;
      case 5: if constexpr (RG > 4) run(std::integral_constant<int, 5>{}); break;
      case 6: if constexpr (RG > 4) run(std::integral_constant<int, 6>{}); break;
      default: if constexpr (RG > 4) run(std::integral_constant<int, 7>{}); break;
    }
  } else {
    run(std::integral_constant<int, 0>{});
  }
  gram_wait_vm<0>();                     // drain the padding DMAs before the workgroup's LDS goes away

  // ---- partial (slice, ks): lane holds D[a = 4g+r][b = lane&15]; stored transposed (G is used through the symmetric
  // form only), so the four r values are one 16-byte store
  float* out = p.partials + (slice * KS + ks) * p.pstride;
  const int t = lane & 15;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ca = colA0 + 16 * (i == 0 ? fa0 : fa1) + 4 * g;
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      if (SYM && j < (i == 0 ? fa0 : fa1)) continue;       // (not computed: the mirror block holds the value)
      const int cb = colB0 + 16 * j + t;
      *reinterpret_cast<float4*>(out + (long)cb * (TN ? p.ldo : (long)p.C) + ca) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    if (diag && t == 0)
      *reinterpret_cast<float4*>(out + (long)p.C * p.C + ca) = make_float4(cs[i][0], cs[i][1], cs[i][2], cs[i][3]);
  }
}

// Stage A: grid (E/256, chunks) -- fold a chunk of partials into fp64.  Stage B (chunks == 1 rows of scratch): same kernel
// over the fp64 scratch.
template <typename TI>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const TI* __restrict__ in, long stride, int n, int per_chunk, long E,
                                                          double* __restrict__ out, int symC, int mirror) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  // symC > 0: the Gram part is stored block-lower-triangular (16 x 16 blocks, see gram_kernel SYM): entries above the block
  // diagonal were never written and are skipped; the last stage (mirror) also writes each strictly-lower entry to its transpose
  int row = 0, col = 0;
  const bool gpart = symC > 0 && e < (long)symC * symC;
  if (gpart) {
    row = (int)(e / symC); col = (int)(e - (long)row * symC);
    if ((row >> 4) < (col >> 4)) return;
  }
  const int t0 = blockIdx.y * per_chunk, t1 = min(n, t0 + per_chunk);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int t = t0;
  for (; t + 4 <= t1; t += 4) {
    s0 += (double)in[(long)t * stride + e];
    s1 += (double)in[(long)(t + 1) * stride + e];
    s2 += (double)in[(long)(t + 2) * stride + e];
    s3 += (double)in[(long)(t + 3) * stride + e];
  }
  for (; t < t1; ++t) s0 += (double)in[(long)t * stride + e];
  const double v = (s0 + s1) + (s2 + s3);
  out[(long)blockIdx.y * E + e] = v;
  if (mirror && gpart && (row >> 4) > (col >> 4)) out[(long)col * symC + row] = v;
}

// Both stages in one launch (what sr_bn_finalize_gram runs): a workgroup owns 64 consecutive elements; thread (group g of 16,
// element e) folds partials g*per .. (g+1)*per-1 into fp64 (four interleaved sums, as above), the sixteen group sums meet in LDS
// and are added in group order -- a fixed order for every element.  The last stage's mirror write (symC) is included.
__global__ __launch_bounds__(1024) void gram_reduce_all_kernel(const float* __restrict__ in, long stride, int n, int per, long E,
                                                               double* __restrict__ out, int symC) {
  __shared__ double red[16][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long e = (long)blockIdx.x * 64 + el;
  int row = 0, col = 0;
  bool live = e < E;
  const bool gpart = live && symC > 0 && e < (long)symC * symC;
  if (gpart) {
    row = (int)(e / symC); col = (int)(e - (long)row * symC);
    if ((row >> 4) < (col >> 4)) live = false;          // (never written: see gram_reduce_kernel)
  }
  double v = 0.0;
  if (live) {
    const int t0 = g * per, t1 = min(n, t0 + per);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = t0;
    for (; t + 4 <= t1; t += 4) {
      s0 += (double)in[(long)t * stride + e];
      s1 += (double)in[(long)(t + 1) * stride + e];
      s2 += (double)in[(long)(t + 2) * stride + e];
      s3 += (double)in[(long)(t + 3) * stride + e];
    }
    for (; t < t1; ++t) s0 += (double)in[(long)t * stride + e];
    v = (s0 + s1) + (s2 + s3);
  }
  red[g][el] = v;
  __syncthreads();
  if (g != 0 || !live) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q += 4) { a0 += red[q][el]; a1 += red[q + 1][el]; a2 += red[q + 2][el]; a3 += red[q + 3][el]; }
  const double tot = (a0 + a1) + (a2 + a3);
  out[e] = tot;
  if (gpart && (row >> 4) > (col >> 4)) out[(long)col * symC + row] = tot;
}

// One workgroup (512 threads) per 8 output channels: s1 = w.cs, s2 = w G w in fp64, then the BatchNorm finalize of
// bn_finalize_kernel (elementwise.hip) for those channels.  Thread (g, kk): rows l of eighth g of G, columns kk + 64u
// (u < 4) -- the w[l][0..7] a step needs are wave-uniform LDS reads (broadcasts), and those were what bounded the first
// form of this kernel (one column per thread: 4 ds_read_b128 per 8 fp64 FMAs, 27 us per launch); four columns per thread
// quarter them.  A thread's partial sum is multiplied by w[k] on the spot (the form is linear in it), so only the per-channel
// totals are reduced across threads.
__global__ __launch_bounds__(512) void gram_project_kernel(const double* __restrict__ G, const double* __restrict__ cs,
                                                            const bf16_t* __restrict__ W, long ldw, int C, int N, double inv_count,
                                                            double unbias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float momentum, float eps,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ rmean2, float* __restrict__ rvar2, float momentum2) {
  __shared__ double wl[512][8];
  __shared__ double red[2][8][8];
  const int n0 = blockIdx.x * 8, tid = threadIdx.x;
  for (int i = tid; i < C * 8; i += 512) {
    const int l = i >> 3, j = i & 7;
    wl[l][j] = n0 + j < N ? (double)(float)W[(long)(n0 + j) * ldw + l] : 0.0;
  }
  __syncthreads();
  const int g = tid >> 6, kk = tid & 63, lq = C >> 3;      // (512 threads: the 4 x 8 fp64 partial sums + 16 totals need ~130 registers)
  double s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
  for (int kb = 0; kb < C; kb += 256) {
    double a[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[u][j] = 0.0;
    const int kbase = kb + kk;                      // columns kbase + 64u; past C (C = 64, 128): read column kk again, weight 0
    // (rows four at a time: their 16 loads are issued together -- one row per trip left every trip waiting an L2 round trip,
    //  20 us per launch for 134 MFLOP; lq = C / 8 is a multiple of 4 for every C served)
    for (int l = g * lq; l < (g + 1) * lq; l += 4) {
      double gv[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[q][u] = G[(long)(l + q) * C + (kbase + 64 * u < C ? kbase + 64 * u : kk)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double w = wl[l + q][j];
#pragma unroll
          for (int u = 0; u < 4; ++u) a[u][j] = fma(gv[q][u], w, a[u][j]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kbase + 64 * u;
      if (k < C) {
        const double ck = g == 0 ? cs[k] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s2[j] = fma(a[u][j], wl[k][j], s2[j]);
          s1[j] = fma(ck, wl[k][j], s1[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1[j] += __shfl_xor(s1[j], o, 64);
      s2[j] += __shfl_xor(s2[j], o, 64);
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][tid >> 6][j] = s1[j]; red[1][tid >> 6][j] = s2[j]; }
  }
  __syncthreads();
  if (tid >= 8 || n0 + tid >= N) return;
  const int c = n0 + tid;
  double t1 = 0.0, t2 = 0.0;
  for (int w = 0; w < 8; ++w) { t1 += red[0][w][tid]; t2 += red[1][w][tid]; }
  const double mean = t1 * inv_count;
  double var = t2 * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);
  if (rmean2) rmean2[c] = (1.f - momentum2) * rmean2[c] + momentum2 * (float)mean;      // a second BatchNorm fed the same batch
  if (rvar2) rvar2[c] = (1.f - momentum2) * rvar2[c] + momentum2 * (float)(var * unbias);
}

int gram_cus() { return sr_num_cus(); }   // (per device: common.h)

struct GramPlan { int P, KS, npan; long rows_per_wg, nslices, npartials; };

bool gram_plan(int64_t M, int C, GramPlan* g) {
  if (M <= 0 || (C != 64 && C != 128 && C != 256 && C != 512)) return false;
  g->P = C > 256 ? 256 : C;
  g->KS = 256 / g->P;
  g->npan = C / g->P;
  const long SR = 32 * g->KS;
  const long ncu = gram_cus();
  // one row slice per CU, but no fp32 accumulator sums more than 8192 pixels
  const long cap = 8192 * g->KS;
  const long rounds = (M + ncu * cap - 1) / (ncu * cap);
  long rows = (M + ncu * rounds - 1) / (ncu * rounds);
  rows = (rows + SR - 1) / SR * SR;
  g->rows_per_wg = rows;
  g->nslices = (M + rows - 1) / rows;
  g->npartials = g->nslices * g->KS;
  return true;
}

template <int P, bool TWO, bool FUSE = false, bool KEEP = true>
int gram_launch(const GramArgs& a, const GramPlan& g, hipStream_t st) {
  constexpr int LDS = 131072 + (FUSE ? 2 * P * 4 : 0);
  if (!sr_set_dynamic_lds<&gram_kernel<P, TWO, false, FUSE, KEEP>>(LDS)) return SR_ERR_LAUNCH;
  hipLaunchKernelGGL((gram_kernel<P, TWO, false, FUSE, KEEP>), dim3((unsigned)(g.nslices * g.npan * g.npan)), dim3(512), LDS, st, a);
  return SR_OK;
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i]   (fp32; 4 elements per thread)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int nslices, long stride, long n4,
                                                        float* __restrict__ out, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nslices; ++k) {
    const float4 v = reinterpret_cast<const float4*>(part + (long)k * stride)[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(out)[i] = s;
}

}  // namespace

/* out[N1, N2] (+)= A^T B for A [M, lda] (N1 columns), B [M, ldb] (N2 columns), bf16, fp32 out (row stride = N2): the
 * weight-gradient GEMM dW = dY^T X without transposed copies of the operands.  Runs on the Gram kernel (both MFMA operands
 * read through ds_read_b64_tr_b16); rows are cut into `nslices` slices so that (N1/256)*(N2/256)*nslices workgroups fill the
 * chip, each slice writes an fp32 partial, tn_reduce_kernel folds them into out.  scratch: nslices*N1*N2 floats. */
extern "C" int sr_gemm_tn_slices(int64_t M, int N1, int N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0 || (N1 & 255) || (N2 & 255)) return SR_ERR_ARG;
  const long blocks = (long)(N1 / 256) * (N2 / 256), ncu = gram_cus();
  long n = (ncu + blocks - 1) / blocks;                 // one round of workgroups over the chip
  const long max_by_rows = (M + 2047) / 2048;           // no slice shorter than 2048 rows
  if (n > max_by_rows) n = max_by_rows;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int sr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int N1, int N2, int dtype, float* out,
                          int accumulate, float* scratch, int64_t scratch_floats, void* stream) {
  const int ns = sr_gemm_tn_slices(M, N1, N2);
  if (ns < 0 || dtype != SR_BF16 || !A || !B || !out || !scratch || lda < N1 || ldb < N2 || (lda & 7) || (ldb & 7) ||
      ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)out & 15) || ((uintptr_t)scratch & 15) ||
      scratch_floats < (int64_t)ns * N1 * N2)
    return SR_ERR_ARG;
  const void* zero = SR_DEVICE_SYMBOL(g_gram_zero);
  if (!zero) return SR_ERR_LAUNCH;
  constexpr int LDS = 131072;
  if (!sr_set_dynamic_lds<&gram_kernel<256, true, true>>(LDS)) return SR_ERR_LAUNCH;
  // the kernel stores block (a, b) transposed -- rows from panel B, four consecutive columns of panel A per lane -- so the
  // matrix whose columns index out's COLUMNS goes in as panel A:  panel A = B (N2), panel B = A (N1)
  GramArgs a{};
  a.x = (const bf16_t*)B; a.ldx = ldb; a.xb = (const bf16_t*)A; a.ldb = lda; a.M = M;
  a.nqa = N2 / 256; a.nqb = N1 / 256; a.ldo = N2; a.C = 0;
  a.partials = scratch; a.pstride = (long)N1 * N2;
  long rows = (M + ns - 1) / ns;
  rows = (rows + 31) / 32 * 32;
  a.rows_per_wg = rows; a.zero = zero;
  const long nsl = (M + rows - 1) / rows;                // (<= ns)
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((gram_kernel<256, true, true>), dim3((unsigned)(nsl * a.nqa * a.nqb)), dim3(512), LDS, st, a);
  const long n4 = (long)N1 * N2 / 4;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)scratch, (int)nsl,
                     (long)N1 * N2, n4, out, accumulate);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_gram_plan(int64_t M, int C, int64_t* npartials, int64_t* partial_floats) {
  GramPlan g;
  if (!gram_plan(M, C, &g) || !npartials || !partial_floats) return SR_ERR_ARG;
  *npartials = g.npartials;
  *partial_floats = (int64_t)C * C + C;
  return SR_OK;
}

extern "C" int sr_gram(const void* x, int64_t M, int C, int64_t ldx, int dtype, float* partials, int64_t npartials, void* stream) {
  GramPlan g;
  if (dtype != SR_BF16 || !x || !partials || !gram_plan(M, C, &g) || npartials != g.npartials || ldx < C || (ldx & 7) ||
      ((uintptr_t)x & 15) || ((uintptr_t)partials & 15))
    return SR_ERR_ARG;
  const void* zero = SR_DEVICE_SYMBOL(g_gram_zero);
  if (!zero) return SR_ERR_LAUNCH;
  GramArgs a;
  a.x = (const bf16_t*)x; a.M = M; a.ldx = ldx; a.C = C; a.partials = partials; a.pstride = (long)C * C + C;
  a.rows_per_wg = g.rows_per_wg; a.zero = zero;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  switch (C) {
    case 64: rc = gram_launch<64, false>(a, g, st); break;
    case 128: rc = gram_launch<128, false>(a, g, st); break;
    case 256: rc = gram_launch<256, false>(a, g, st); break;
    default: rc = gram_launch<256, true>(a, g, st); break;
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

/* x = relu(x*scale[c] + shift[c]) in place (what sr_bn_apply does after the 3x3 conv of a bottleneck) AND the Gram partials of
 * the result for the expansion conv that follows, in one pass over the tensor (C in {64,128,256}; layout of `partials` as
 * sr_gram). */
static int bn_gram_launch(const void* x, int64_t M, int C, int64_t ldx, int dtype, const float* scale, const float* shift,
                          float* partials, int64_t npartials, int keep, void* stream) {
  GramPlan g;
  if (dtype != SR_BF16 || !x || !partials || !scale || !shift || !gram_plan(M, C, &g) || g.npan != 1 || npartials != g.npartials ||
      ldx < C || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)partials & 15))
    return SR_ERR_ARG;
  const void* zero = SR_DEVICE_SYMBOL(g_gram_zero);
  void* trash = SR_DEVICE_SYMBOL(g_gram_trash);
  if (!zero || !trash) return SR_ERR_LAUNCH;
  GramArgs a{};
  a.x = (const bf16_t*)x; a.M = M; a.ldx = ldx; a.C = C; a.partials = partials; a.pstride = (long)C * C + C;
  a.rows_per_wg = g.rows_per_wg; a.zero = zero; a.scale = scale; a.shift = shift; a.trash = trash;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  switch (C) {
    case 64: rc = keep ? gram_launch<64, false, true>(a, g, st) : gram_launch<64, false, true, false>(a, g, st); break;
    case 128: rc = keep ? gram_launch<128, false, true>(a, g, st) : gram_launch<128, false, true, false>(a, g, st); break;
    default: rc = keep ? gram_launch<256, false, true>(a, g, st) : gram_launch<256, false, true, false>(a, g, st); break;
  }
  if (rc != SR_OK) return rc;
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_bn_apply_gram(void* x, int64_t M, int C, int64_t ldx, int dtype, const float* scale, const float* shift,
                                float* partials, int64_t npartials, void* stream) {
  return bn_gram_launch(x, M, C, ldx, dtype, scale, shift, partials, npartials, 1, stream);
}

/* As sr_bn_apply_gram, but x is left as it is: the Gram partials are those of relu(x*scale + shift), which only ever exists in
 * LDS.  For a consumer that applies the same affine on load (sr_conv2d with in_scale / in_shift). */
extern "C" int sr_bn_gram(const void* x, int64_t M, int C, int64_t ldx, int dtype, const float* scale, const float* shift,
                          float* partials, int64_t npartials, void* stream) {
  return bn_gram_launch(x, M, C, ldx, dtype, scale, shift, partials, npartials, 0, stream);
}

extern "C" int sr_bn_finalize_gram(const float* partials, int64_t npartials, int C, const void* w, int64_t ldw, int N, int dtype,
                                   int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* scale, float* shift, double* scratch, int64_t scratch_elems,
                                   float* running_mean2, float* running_var2, float momentum2, void* stream) {
  if (dtype != SR_BF16 || !partials || npartials <= 0 || !w || N <= 0 || ldw < C || count <= 0 || !gamma || !beta || !scale || !shift ||
      !scratch || (C != 64 && C != 128 && C != 256 && C != 512))
    return SR_ERR_ARG;
  const long E = (long)C * C + C;
  int chunks = (int)((npartials + 15) / 16);
  if (chunks > 64) chunks = 64;
  const int per = (int)((npartials + chunks - 1) / chunks);
  chunks = (int)((npartials + per - 1) / per);
  if (scratch_elems < (long)(chunks + 1) * E) return SR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  double* stageA = scratch + E;   // [chunks][E]; final fp64 G | colsum at scratch[0..E)
  const unsigned gx = (unsigned)((E + 255) / 256);
  const int symC = C <= 256 ? C : 0;       // (C = 512 runs as 2 x 2 panels of 256: all four blocks are computed)
  if (chunks <= 16) {          // (up to 256 partials -- one per CU: every BatchNorm of a ResNet at any batch) both stages in one launch
    const int per16 = (int)((npartials + 15) / 16);
    hipLaunchKernelGGL(gram_reduce_all_kernel, dim3((unsigned)((E + 63) / 64)), dim3(1024), 0, st, partials, E, (int)npartials, per16, E,
                       scratch, symC);
  } else {
    hipLaunchKernelGGL(gram_reduce_kernel<float>, dim3(gx, chunks), dim3(256), 0, st, partials, E, (int)npartials, per, E, stageA, symC, 0);
    hipLaunchKernelGGL(gram_reduce_kernel<double>, dim3(gx, 1), dim3(256), 0, st, (const double*)stageA, E, chunks, chunks, E, scratch, symC, 1);
  }
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  hipLaunchKernelGGL(gram_project_kernel, dim3((N + 7) / 8), dim3(512), 0, st, (const double*)scratch, (const double*)(scratch + (long)C * C),
                     (const bf16_t*)w, (long)ldw, C, N, 1.0 / (double)count, unbias, gamma, beta, running_mean, running_var, momentum, eps,
                     scale, shift, running_mean2, running_var2, momentum2);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
