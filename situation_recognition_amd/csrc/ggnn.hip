// Role-graph kernels of the GGNN head (reference model.py:115-155, 59-86) that are not GEMMs:
// node initialisation (embedding gathers fused with the feature product), the 6x6 adjacency
// message aggregation, and the elementwise halves of the GRU backward.  All HBM-bound:
// each lane owns one 16-byte column strip and streams it once.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxR = 8;  // max roles per verb supported (imSitu: 6)

inline unsigned grid_for(long work_items) {
  long g = (work_items + kThreads - 1) / kThreads;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (unsigned)g;
}

// node[b,r,:] = relu(feat[b,:] * role_emb[role_table[verb_b][r],:] * verb_emb[verb_b,:])
// Packed form (`offs` != NULL, [B+1] prefix sums of the images' role counts): only the rows of REAL roles exist -- image b's role r
// is row offs[b] + r, r < offs[b+1] - offs[b] -- see sr_node_init_fwd in include/srhip.h.
template <typename T>
__global__ void node_init_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ role_emb,
                                     const float* __restrict__ verb_emb, const int64_t* __restrict__ verbs,
                                     const int32_t* __restrict__ role_table, T* __restrict__ node, int B, int R, int D,
                                     const int32_t* __restrict__ offs) {
  constexpr int N = Vec16<T>::N;
  const int dv = D / N;
  const long total = (long)B * dv;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int d = (int)(idx % dv) * N;
    const long b = idx / dv;
    const long v = verbs[b];
    Vec16<T> f = ld16<T>(feat + b * D + d);
    float fx[N], ve[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { fx[k] = f.get(k); ve[k] = verb_emb[v * D + d + k]; }
    const long row0 = offs ? offs[b] : b * R;
    const int nr = offs ? offs[b + 1] - offs[b] : R;
    for (int r = 0; r < nr; ++r) {
      const long rid = role_table[v * R + r];
      Vec16<T> o;
#pragma unroll
      for (int k = 0; k < N; ++k) o.set(k, fmaxf((fx[k] * role_emb[rid * D + d + k]) * ve[k], 0.f));  // model.py:143 order
      st16<T>(node + (row0 + r) * D + d, o);
    }
  }
}

// Backward of node init, deterministic (no atomics).  y = relu(f*ro*ve):  d ro = g*[y>0]*f*ve ; d ve = sum_r g*[y>0]*f*ro.
// Within one verb v the factors ve (and ro of slot r) are constant, so with S[v,r,:] = sum_{b: verb_b = v} g_{b,r}*[y>0]*f_b
//     d_verb[v] = sum_r ro[rid(v,r)] * S[v,r]          d_role[rid] = sum_{(v,r): rid(v,r) = rid} ve[v] * S[v,r].
// Phase 1a: the batch, in the (stable) verb-sorted order the host hands over, is cut into CHUNKS of kChunk positions; one wave per
// (chunk, 64 x 16-byte column strip) walks its positions and sums each verb's run.  A verb whose images all lie inside the chunk is
// finished there (S and d_verb written); a run that reaches the chunk's start or end leaves a partial: PH[chunk] (the run that
// begins at the chunk's first position) or PT[chunk] (the run that begins inside and continues past its end).  Phase 1b: one wave
// per (verb, strip) adds the partials of a verb that spans chunks, in chunk order, and writes zeros for absent verbs.  Phase 2: one
// wave per (role, strip) walks the role's (verb, slot) list -- a static inverted index of the encoder's role table.  Every sum has
// a fixed order: gradients are bit-reproducible (the atomic version was not: ~190 adders per d_role row at batch 6144).
// (Round 3: the first form gave one wave a WHOLE verb.  With most images on one verb -- argmax of an untrained verb head, or any
//  skewed batch -- that wave walked thousands of images alone: 2.3 ms at batch 6144 for 180 MB of traffic.)
constexpr int kChunk = 32;

// largest v with seg[v] <= pos (< seg[v+1]): the verb of sorted position pos (seg is non-decreasing, seg[0] = 0, seg[V] = B > pos)
__device__ __forceinline__ int verb_of_position(const int32_t* __restrict__ seg, int V, int pos) {
  int lo = 0, hi = V;                                  // invariant: seg[lo] <= pos < seg[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename T>
__global__ __launch_bounds__(64) void node_init_bwd1a_kernel(const T* __restrict__ dnode, const T* __restrict__ feat,
                                                             const float* __restrict__ role_emb, const float* __restrict__ verb_emb,
                                                             const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
                                                             const int32_t* __restrict__ role_table, float* __restrict__ S,
                                                             float* __restrict__ PH, float* __restrict__ PT,
                                                             float* __restrict__ d_verb, int B, int R, int D, int V, int NR,
                                                             const int32_t* __restrict__ offs) {
  constexpr int N = Vec16<T>::N;
  const int chunk = blockIdx.x, c0 = chunk * kChunk, c1 = min(B, c0 + kChunk);
  // Lanes past the last column strip stay ALIVE (their loads fall on column 0, their stores are predicated): the walk below reads
  // the chunk's image indices out of the lanes 0 .. 31 with readlane, and an exited lane's registers were never written
  // (D / N < 32 lanes -- bf16 D = 64, or D = 128 with B > 16 -- read garbage indices before this).
  const int d_raw = (blockIdx.y * 64 + threadIdx.x) * N;
  const bool live = d_raw < D;
  const int d = live ? d_raw : 0;
  // the chunk's image indices and first node rows, one per lane (read by lane index below: no dependent loads in the walk)
  const int pos = c0 + (int)(threadIdx.x & (kChunk - 1));
  const int my_b = pos < c1 ? order[pos] : 0;
  const int my_o = offs ? offs[my_b] : my_b * R;
  int v = verb_of_position(seg, V, c0);
  int i = c0;
  while (i < c1) {
    const int s0 = seg[v], s1 = seg[v + 1];
    const int e = s1 < c1 ? s1 : c1;
    float ve[N], ro[kMaxR][N], acc[kMaxR][N];
    int rid[kMaxR];
#pragma unroll
    for (int k = 0; k < N; ++k) ve[k] = verb_emb[(long)v * D + d + k];
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) {
      rid[r] = r < R ? role_table[v * R + r] : NR;
#pragma unroll
      for (int k = 0; k < N; ++k) { ro[r][k] = rid[r] != NR ? role_emb[(long)rid[r] * D + d + k] : 0.f; acc[r][k] = 0.f; }
    }
    const int first = i;
    for (; i < e; ++i) {
      const long b = __builtin_amdgcn_readlane(my_b, i - c0);
      const long row0 = __builtin_amdgcn_readlane(my_o, i - c0);
      Vec16<T> f = ld16<T>(feat + b * D + d);
#pragma unroll
      for (int r = 0; r < kMaxR; ++r) {
        if (r < R && rid[r] != NR) {
          Vec16<T> g = ld16<T>(dnode + (row0 + r) * D + d);                  // (packed: a real role r is always < the image's count)
#pragma unroll
          for (int k = 0; k < N; ++k) {
            const float fx = f.get(k);
            const float pre = fx * ro[r][k] * ve[k];
            acc[r][k] += pre > 0.f ? g.get(k) * fx : 0.f;
          }
        }
      }
    }
    const bool whole = s0 >= c0 && s1 <= c1;
    float* const dst = whole ? S + (long)v * R * D : (first == c0 ? PH : PT) + (long)chunk * R * D;
    float dve[N];
#pragma unroll
    for (int k = 0; k < N; ++k) dve[k] = 0.f;
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) {
      if (r < R) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          dve[k] += acc[r][k] * ro[r][k];
          if (live) dst[(long)r * D + d + k] = acc[r][k];
        }
      }
    }
    if (whole && live) {
#pragma unroll
      for (int k = 0; k < N; ++k) d_verb[(long)v * D + d + k] = dve[k];
    }
    do { ++v; } while (v < V && seg[v + 1] <= i);      // the next verb that has images at or behind position i
  }
}

// Phase 1b: verbs absent from the batch (zeros) and verbs whose run crosses a chunk boundary (partials added in chunk order).
__global__ __launch_bounds__(64) void node_init_bwd1b_kernel(const float* __restrict__ role_emb, const int32_t* __restrict__ seg,
                                                             const int32_t* __restrict__ role_table, float* __restrict__ S,
                                                             const float* __restrict__ PH, const float* __restrict__ PT,
                                                             float* __restrict__ d_verb, int R, int D, int NR) {
  const int v = blockIdx.x;
  const int d = (blockIdx.y * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  const int s = seg[v], e = seg[v + 1];
  const int jf = s / kChunk, jl = e > s ? (e - 1) / kChunk : jf;
  if (e > s && jf == jl) return;                         // finished by phase 1a
  float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < R; ++r) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e > s) {
      a = *reinterpret_cast<const float4*>((s % kChunk == 0 ? PH : PT) + ((long)jf * R + r) * D + d);
      for (int j = jf + 1; j <= jl; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(PH + ((long)j * R + r) * D + d);
        a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
      }
      const int rid = role_table[v * R + r];
      if (rid != NR) {
        const float4 w = *reinterpret_cast<const float4*>(role_emb + (long)rid * D + d);
        dv.x += a.x * w.x; dv.y += a.y * w.y; dv.z += a.z * w.z; dv.w += a.w * w.w;
      }
    }
    *reinterpret_cast<float4*>(S + ((long)v * R + r) * D + d) = a;
  }
  *reinterpret_cast<float4*>(d_verb + (long)v * D + d) = dv;
}

__global__ __launch_bounds__(64) void node_init_bwd2_kernel(const float* __restrict__ S, const float* __restrict__ verb_emb,
                                                            const int32_t* __restrict__ inv_ptr, const int32_t* __restrict__ inv_slot,
                                                            float* __restrict__ d_role, int R, int D, int NR) {
  const int rid = blockIdx.x;
  const int d = (blockIdx.y * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rid < NR) {                                   // (row NR = padding_idx: stays zero)
    for (int e = inv_ptr[rid]; e < inv_ptr[rid + 1]; ++e) {
      const int slot = inv_slot[e];                 // v * R + r
      const float4 s = *reinterpret_cast<const float4*>(S + (long)slot * D + d);
      const float4 w = *reinterpret_cast<const float4*>(verb_emb + (long)(slot / R) * D + d);
      a.x += s.x * w.x; a.y += s.y * w.y; a.z += s.z * w.z; a.w += s.w * w.w;
    }
  }
  *reinterpret_cast<float4*>(d_role + (long)rid * D + d) = a;
}

// out[b,i,:] = sum_j A[i][j] h[b,j,:] (+ add).  One workgroup = one image: the R x R adjacency of
// the image's verb is staged in LDS once, every lane keeps its R-row column strip in registers.
// Packed form (`offs` != NULL): image b owns the rows offs[b] .. offs[b+1]-1 (its real roles; the leading block of its verb's
// adjacency), and "image" B is the single shared row of the padded roles (adjacency = its own diagonal 1: out = h (+ add)).
template <typename T, int RR>
__global__ __launch_bounds__(kThreads) void aggregate_kernel(const T* __restrict__ h, const float* __restrict__ adj,
                                                             const int64_t* __restrict__ verbs, const T* __restrict__ add,
                                                             T* __restrict__ out, int B, int D, int transpose,
                                                             const int32_t* __restrict__ offs) {
  constexpr int N = Vec16<T>::N;
  __shared__ float A[RR * RR];
  const int dv = D / N;
  const long nimg = offs ? (long)B + 1 : B;
  for (long b = blockIdx.x; b < nimg; b += gridDim.x) {
    __syncthreads();
    const bool padrow = b == B;
    if (threadIdx.x < RR * RR) {
      const int i = threadIdx.x / RR, j = threadIdx.x % RR;
      A[threadIdx.x] = padrow ? (i == j ? 1.f : 0.f) : adj[verbs[b] * (RR * RR) + (transpose ? j * RR + i : i * RR + j)];
    }
    __syncthreads();
    const long row0 = offs ? (long)offs[padrow ? B : b] : b * RR;
    const int nr = offs ? (padrow ? 1 : offs[b + 1] - offs[b]) : RR;
    for (int c = threadIdx.x; c < dv; c += kThreads) {
      const long base = row0 * (long)D + c * N;
      Vec16<T> hv[RR];
#pragma unroll
      for (int j = 0; j < RR; ++j)
        if (j < nr) hv[j] = ld16<T>(h + base + (long)j * D);
#pragma unroll
      for (int i = 0; i < RR; ++i) {
        if (i >= nr) break;
        float s[N];
        if (add) {
          Vec16<T> a = ld16<T>(add + base + (long)i * D);
#pragma unroll
          for (int k = 0; k < N; ++k) s[k] = a.get(k);
        } else {
#pragma unroll
          for (int k = 0; k < N; ++k) s[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < RR; ++j) {
          if (j < nr) {
            const float a = A[i * RR + j];
#pragma unroll
            for (int k = 0; k < N; ++k) s[k] += a * hv[j].get(k);
          }
        }
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < N; ++k) o.set(k, s[k]);
        st16<T>(out + base + (long)i * D, o);
      }
    }
  }
}

template <typename T>
__global__ void gru_bwd1_kernel(const T* __restrict__ dh, const T* __restrict__ z, const T* __restrict__ c,
                                const T* __restrict__ h, T* __restrict__ dc_pre, T* __restrict__ dz_pre,
                                T* __restrict__ dh_acc, long nvec) {
  constexpr int N = Vec16<T>::N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long e = i * N;
    Vec16<T> g = ld16<T>(dh + e), zv = ld16<T>(z + e), cv = ld16<T>(c + e), hv = ld16<T>(h + e), o1, o2, o3;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = g.get(k), zz = zv.get(k), cc = cv.get(k), hh = hv.get(k);
      o1.set(k, gg * zz * (1.f - cc * cc));
      o2.set(k, gg * (cc - hh) * zz * (1.f - zz));
      o3.set(k, gg * (1.f - zz));
    }
    st16<T>(dc_pre + e, o1); st16<T>(dz_pre + e, o2); st16<T>(dh_acc + e, o3);
  }
}

template <typename T>
__global__ void gru_bwd2_kernel(const T* __restrict__ drh, const T* __restrict__ r, const T* __restrict__ h,
                                T* __restrict__ dr_pre, T* __restrict__ dh_acc, long nvec) {
  constexpr int N = Vec16<T>::N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long e = i * N;
    Vec16<T> g = ld16<T>(drh + e), rv = ld16<T>(r + e), hv = ld16<T>(h + e), acc = ld16<T>(dh_acc + e), o1, o2;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = g.get(k), rr = rv.get(k), hh = hv.get(k);
      o1.set(k, gg * hh * rr * (1.f - rr));
      o2.set(k, acc.get(k) + gg * rr);
    }
    st16<T>(dr_pre + e, o1); st16<T>(dh_acc + e, o2);
  }
}

template <typename T>
int launch_aggregate(const void* h, const float* adj, const int64_t* verbs, const void* add, void* out, int B, int R, int D,
                     int transpose, const int32_t* offs, hipStream_t st) {
  const unsigned g = (unsigned)(B + 1 < 256 * 16 ? B + 1 : 256 * 16);
#define AGG(RR)                                                                                                   \
  case RR:                                                                                                        \
    hipLaunchKernelGGL((aggregate_kernel<T, RR>), dim3(g), dim3(kThreads), 0, st, (const T*)h, adj, verbs,        \
                       (const T*)add, (T*)out, B, D, transpose, offs);                                            \
    break;
  switch (R) {
    AGG(1) AGG(2) AGG(3) AGG(4) AGG(5) AGG(6) AGG(7) AGG(8)
    default: return SR_ERR_UNSUPPORTED;
  }
#undef AGG
  SR_CHECK_LAUNCH();
  return SR_OK;
}

}  // namespace

#define DT_SWITCH(dtype, EXPR)                        \
  if ((dtype) == SR_F32) { using T = float; EXPR; }   \
  else if ((dtype) == SR_BF16) { using T = bf16_t; EXPR; } \
  else return SR_ERR_DTYPE;

extern "C" int sr_node_init_fwd(const void* feat, const float* role_emb, const float* verb_emb, const int64_t* verbs,
                                const int32_t* role_table, void* node, int B, int R, int D, int dtype, const int32_t* offs,
                                void* stream) {
  if (!feat || !role_emb || !verb_emb || !verbs || !role_table || !node || B <= 0 || R <= 0 || D <= 0) return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (D % n) return SR_ERR_ARG;
  const long total = (long)B * (D / n);
  DT_SWITCH(dtype, hipLaunchKernelGGL(node_init_fwd_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)feat, role_emb, verb_emb, verbs, role_table, (T*)node, B, R, D, offs));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_node_init_bwd(const void* dnode, const void* feat, const float* role_emb, const float* verb_emb,
                                const int32_t* order, const int32_t* seg, const int32_t* role_table, const int32_t* inv_ptr,
                                const int32_t* inv_slot, float* scratch, float* d_role_emb, float* d_verb_emb, int B, int R, int D,
                                int V, int NR, int dtype, const int32_t* offs, void* stream) {
  if (!dnode || !feat || !role_emb || !verb_emb || !order || !seg || !role_table || !inv_ptr || !inv_slot || !scratch ||
      !d_role_emb || !d_verb_emb || B <= 0 || R <= 0 || R > kMaxR || D <= 0 || V <= 0 || NR < 0)
    return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (D % n || D % 4) return SR_ERR_ARG;
  const unsigned gy1 = (unsigned)((D / n + 63) / 64), gy2 = (unsigned)((D / 4 + 63) / 64);
  const long nchunk = ((long)B + kChunk - 1) / kChunk;
  float* const S = scratch;                                  // [V][R][D]
  float* const PH = S + (long)V * R * D;                     // [nchunk][R][D]
  float* const PT = PH + nchunk * R * D;                     // [nchunk][R][D]
  DT_SWITCH(dtype, hipLaunchKernelGGL(node_init_bwd1a_kernel<T>, dim3((unsigned)nchunk, gy1), dim3(64), 0, (hipStream_t)stream,
                                      (const T*)dnode, (const T*)feat, role_emb, verb_emb, order, seg, role_table, S, PH, PT,
                                      d_verb_emb, B, R, D, V, NR, offs));
  SR_CHECK_LAUNCH();
  hipLaunchKernelGGL(node_init_bwd1b_kernel, dim3((unsigned)V, gy2), dim3(64), 0, (hipStream_t)stream, role_emb, seg, role_table, S,
                     (const float*)PH, (const float*)PT, d_verb_emb, R, D, NR);
  SR_CHECK_LAUNCH();
  hipLaunchKernelGGL(node_init_bwd2_kernel, dim3((unsigned)NR + 1, gy2), dim3(64), 0, (hipStream_t)stream, scratch, verb_emb,
                     inv_ptr, inv_slot, d_role_emb, R, D, NR);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_ggnn_aggregate(const void* h, const float* adj_table, const int64_t* verbs, const void* add, void* out,
                                 int B, int R, int D, int transpose, int dtype, const int32_t* offs, void* stream) {
  if (!h || !adj_table || !verbs || !out || B <= 0 || R <= 0 || R > kMaxR || D <= 0) return SR_ERR_ARG;
  const int n = dtype == SR_F32 ? 4 : 8;
  if (D % n) return SR_ERR_ARG;
  if (dtype == SR_F32) return launch_aggregate<float>(h, adj_table, verbs, add, out, B, R, D, transpose, offs, (hipStream_t)stream);
  if (dtype == SR_BF16) return launch_aggregate<bf16_t>(h, adj_table, verbs, add, out, B, R, D, transpose, offs, (hipStream_t)stream);
  return SR_ERR_DTYPE;
}

extern "C" int sr_gru_bwd1(const void* dh, const void* z, const void* c, const void* h, void* dc_pre, void* dz_pre,
                           void* dh_acc, int64_t n, int dtype, void* stream) {
  if (!dh || !z || !c || !h || !dc_pre || !dz_pre || !dh_acc || n <= 0) return SR_ERR_ARG;
  const int nv = dtype == SR_F32 ? 4 : 8;
  if (n % nv) return SR_ERR_ARG;
  const long nvec = n / nv;
  DT_SWITCH(dtype, hipLaunchKernelGGL(gru_bwd1_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)dh, (const T*)z, (const T*)c, (const T*)h, (T*)dc_pre, (T*)dz_pre,
                                      (T*)dh_acc, nvec));
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int sr_gru_bwd2(const void* drh, const void* r, const void* h, void* dr_pre, void* dh_acc, int64_t n, int dtype,
                           void* stream) {
  if (!drh || !r || !h || !dr_pre || !dh_acc || n <= 0) return SR_ERR_ARG;
  const int nv = dtype == SR_F32 ? 4 : 8;
  if (n % nv) return SR_ERR_ARG;
  const long nvec = n / nv;
  DT_SWITCH(dtype, hipLaunchKernelGGL(gru_bwd2_kernel<T>, dim3(grid_for(nvec)), dim3(kThreads), 0, (hipStream_t)stream,
                                      (const T*)drh, (const T*)r, (const T*)h, (T*)dr_pre, (T*)dh_acc, nvec));
  SR_CHECK_LAUNCH();
  return SR_OK;
}
