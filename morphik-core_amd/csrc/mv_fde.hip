// mv_fde.hip -- MUVERA fixed dimensional encoding (FDE) and the coarse scan over FDE vectors.
//
// Replaces the un-vendored C++ extension `fixed_dimensional_encoding` as the reference uses it:
//   config          core/vector_store/fast_multivector_store.py:325-331
//                   (dimension=128, num_repetitions=20, num_simhash_projections=5,
//                    projection_dimension=16, projection_type=AMS_SKETCH)
//   doc encoding    fde.generate_document_encoding(...)   fast_multivector_store.py:447-449  (AVERAGE)
//   query encoding  fde.generate_query_encoding(...)      fast_multivector_store.py:521      (SUM)
//   coarse ranking  TurboPuffer ANN over the FDE vectors, cosine distance
//                   fast_multivector_store.py:497,526-532   -> here an exact scan, dot * 1/|d|.
// The extension's sources are absent from the reference snapshot; the algorithm is restated from
// the MUVERA paper / graph-mining fixed_dimensional_encoding.cc and pinned only against our own
// oracle (oracle/mv_oracle.c, "PARITY UNPINNED").
//
// Arithmetic contract (shared with the oracle so SimHash partitions agree bit for bit):
//   sketch_j = fp32 fmaf chain over k ascending from 0.0f
//   proj_c   = fp32 sum over i ascending of (+-x_i) for h(i)=c, then * scale
// Bucket sums over the rows of a partition use LDS float atomics: order-dependent in the last bits,
// compared with a tolerance (no discontinuity downstream).
#include <algorithm>
#include <cmath>
#include <vector>

#include <type_traits>
#include <utility>

#include "mv_common.h"

namespace mv {
namespace {

constexpr int kRowsPerPass = 64;
constexpr int kMaxNS = 6;
constexpr int kMaxPD = 16;

void philox_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

struct EncArgs {
  const float* x_f32;
  const uint16_t* x_bf16;
  const int64_t* row_offsets;
  const int32_t* n_rows;
  int32_t stride;
  int32_t is_query;
  // tables
  const float* G;        // [R][128][NS]
  const int32_t* order;  // [R][128] input dims sorted by (bucket, dim)
  const float* sgn;      // [R][128] sign of order[m]
  const int32_t* bstart; // [R][PD+1]
  int32_t R, NS, PD;
  float scale;
  int64_t out_dim;
  float* out_f32;
  uint16_t* out_bf16;
  float* out_inv_norm;
};

// One block per page; 4 waves split the repetitions; 64 rows staged per pass, transposed in LDS.
__global__ __launch_bounds__(256) void fde_encode_kernel(EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xT = reinterpret_cast<float*>(smem);                     // [128][64]
  float* acc = xT + kDim * kRowsPerPass;                          // [R][NP][PD]
  const int NP = 1 << a.NS;
  int32_t* cnt = reinterpret_cast<int32_t*>(acc + a.out_dim);     // [R][NP]
  float* red = reinterpret_cast<float*>(cnt + a.R * NP);          // [4]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t page = blockIdx.x;

  int64_t r0 = 0;
  int32_t nr;
  if (a.x_f32) {
    if (a.row_offsets) {
      r0 = a.row_offsets[page];
      nr = (int32_t)(a.row_offsets[page + 1] - r0);
    } else {  // ONE page of `stride` rows (the query): no offset array to upload
      nr = a.stride;
    }
  } else {
    nr = a.n_rows ? a.n_rows[page] : a.stride;
    r0 = page * (int64_t)a.stride;
  }

  for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) acc[i] = 0.0f;
  for (int i = threadIdx.x; i < a.R * NP; i += 256) cnt[i] = 0;

  for (int p0 = 0; p0 < nr; p0 += kRowsPerPass) {
    __syncthreads();  // previous pass done with xT; also orders the zero-fill above
    // stage rows p0..p0+63 transposed: thread -> (row = i / 32, 4 dims)
    for (int i = threadIdx.x; i < kRowsPerPass * (kDim / 4); i += 256) {
      const int row = i >> 5, c4 = (i & 31) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + row < nr) {
        const int64_t e = (r0 + p0 + row) * kDim + c4;
        if (a.x_f32) {
          v = *reinterpret_cast<const float4*>(a.x_f32 + e);
        } else {
          const uint2 h = *reinterpret_cast<const uint2*>(a.x_bf16 + e);
          v.x = bf16_to_f32((uint16_t)(h.x & 0xffff)); v.y = bf16_to_f32((uint16_t)(h.x >> 16));
          v.z = bf16_to_f32((uint16_t)(h.y & 0xffff)); v.w = bf16_to_f32((uint16_t)(h.y >> 16));
        }
      }
      xT[(c4 + 0) * kRowsPerPass + row] = v.x;
      xT[(c4 + 1) * kRowsPerPass + row] = v.y;
      xT[(c4 + 2) * kRowsPerPass + row] = v.z;
      xT[(c4 + 3) * kRowsPerPass + row] = v.w;
    }
    __syncthreads();
    const bool valid = p0 + lane < nr;
    for (int r = blockIdx.y * 4 + wave; r < a.R; r += 4 * gridDim.y) {
      // ---- SimHash: NS fmaf chains over k ascending
      float sk[kMaxNS];
#pragma unroll
      for (int j = 0; j < kMaxNS; ++j) sk[j] = 0.0f;
      const float* Gr = a.G + (size_t)r * kDim * a.NS;
      for (int k = 0; k < kDim; ++k) {
        const float x = xT[k * kRowsPerPass + lane];
#pragma unroll
        for (int j = 0; j < kMaxNS; ++j)
          if (j < a.NS) sk[j] = __builtin_fmaf(x, Gr[k * a.NS + j], sk[j]);
      }
      uint32_t part = 0;
#pragma unroll
      for (int j = 0; j < kMaxNS; ++j)
        if (j < a.NS) part = (part << 1) + ((sk[j] > 0.0f ? 1u : 0u) ^ (part & 1u));  // AppendToGrayCode
      // ---- AMS projection, bucket by bucket (members in ascending input dim)
      float pj[kMaxPD];
      const int32_t* ord = a.order + (size_t)r * kDim;
      const float* sg = a.sgn + (size_t)r * kDim;
      const int32_t* bs = a.bstart + (size_t)r * (a.PD + 1);
#pragma unroll
      for (int c = 0; c < kMaxPD; ++c) {
        pj[c] = 0.0f;
        if (c < a.PD) {
          const int m1 = bs[c + 1];
          for (int m = bs[c]; m < m1; ++m) pj[c] = __builtin_fmaf(sg[m], xT[ord[m] * kRowsPerPass + lane], pj[c]);
          pj[c] *= a.scale;
        }
      }
      if (valid) {
        float* dst = acc + ((size_t)r * NP + part) * a.PD;
#pragma unroll
        for (int c = 0; c < kMaxPD; ++c)
          if (c < a.PD) atomicAdd(&dst[c], pj[c]);
        atomicAdd(&cnt[r * NP + part], 1);
      }
    }
  }
  __syncthreads();
  // ---- finish: AVERAGE for documents, write fp32 / bf16, inverse norm of the bf16 image.
  // With the repetitions split over blockIdx.y (latency form used for the single query page) a block writes only
  // the slices of its own repetitions; the norm is then not produced (queries do not need it).
  float nn = 0.0f;
  const int rep_elems = NP * a.PD;
  for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) {
    if (gridDim.y > 1 && ((i / rep_elems) >> 2) % (int)gridDim.y != (int)blockIdx.y) continue;
    float v = acc[i];
    if (!a.is_query) {
      const int n = cnt[i / a.PD];
      if (n > 1) v = v / (float)n;
    }
    if (a.out_f32) a.out_f32[page * a.out_dim + i] = v;
    const uint16_t h = f32_to_bf16_rne(v);
    if (a.out_bf16) a.out_bf16[page * a.out_dim + i] = h;
    const float vb = bf16_to_f32(h);
    nn += vb * vb;
  }
  if (a.out_inv_norm && gridDim.y == 1) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) nn += __shfl_xor(nn, s);
    if (lane == 0) red[wave] = nn;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = (red[0] + red[1]) + (red[2] + red[3]);
      a.out_inv_norm[page] = t > 0.0f ? 1.0f / sqrtf(t) : 0.0f;
    }
  }
}

// ------------------------------------------------------------------------------ encode on the f32 matrix cores
// The same arithmetic as fde_encode_kernel, bit for bit where the oracle is bit-exact:
//   * v_mfma_f32_16x16x4_f32 IS a k-ordered fp32 fmaf chain (one rounding per product-add, C chained across the
//     K steps), so the SimHash sketches  fmaf(x_k, G[k][h], acc)  over k ascending from 0.0f  are reproduced exactly
//     -> identical sign bits -> identical partitions;
//   * the AMS projection is the same chain against a dense {0, +1, -1} column:  fma(x, 0, acc) = acc  and
//     fma(x, +-1, acc) = round(acc +- x)  -- the oracle's ascending sparse sum, exactly.
// What changes is the cost: the scalar kernel reads x from LDS once per FMA and funnels 64 lanes into 32 bucket
// addresses with LDS atomics (11 us/page); here a wave keeps a 16-row x 128-dim tile in 32 VGPRs as the MFMA A operand
// and streams 27 column tiles (7 SimHash + 20 AMS) past it; bucket sums are 4-way conflicts at worst.
// LDS: G^T [128][112] fp32 (56 KiB) + packed AMS codes (2.5 KiB) + acc (40 KiB) + cnt + per-wave x tile / signs.
struct EncMArgs {
  EncArgs e;
  const int32_t* H;   // [R][128] bucket of each input dim
  const float* S;     // [R][128] sign of each input dim
  int64_t n_pages;
};

constexpr int kXStride = 130;  // floats per staged row: conflict-free A-fragment reads

__global__ __launch_bounds__(256) void fde_encode_mfma_kernel(EncMArgs m) {
  const EncArgs& a = m.e;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NP = 1 << a.NS;
  const int NH = a.R * a.NS;              // hash functions
  const int NHP = ((NH + 15) / 16) * 16;  // padded to column tiles
  const int NT = NHP / 16;
  float* Gs = reinterpret_cast<float*>(smem);                       // [128][NHP]
  float* acc = Gs + kDim * NHP;                                     // [out_dim]
  int32_t* cnt = reinterpret_cast<int32_t*>(acc + a.out_dim);       // [R*NP]
  float* red = reinterpret_cast<float*>(cnt + a.R * NP);            // [4]
  uint8_t* codeT = reinterpret_cast<uint8_t*>(red + 4);             // [R][4][32]: code of dim 4s+k at [rep][k][s]
  float* xs_all = reinterpret_cast<float*>(codeT + ((a.R * 128 + 15) / 16) * 16);  // [4 waves][16][kXStride]
  uint8_t* sg_all = reinterpret_cast<uint8_t*>(xs_all + 4 * 16 * kXStride);         // [4 waves][16][NHP] sign bytes
  uint8_t* pt_all = sg_all + 4 * 16 * NHP;                                          // [4 waves][16][R] partitions

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, k = lane >> 4;
  float* xs = xs_all + wave * 16 * kXStride;
  uint8_t* sg = sg_all + wave * 16 * NHP;
  uint8_t* pt = pt_all + wave * 16 * a.R;

  // tables -> LDS (once per block; blocks are persistent over pages)
  for (int i = threadIdx.x; i < kDim * NHP; i += 256) {
    const int dim = i / NHP, h = i - dim * NHP;
    float v = 0.f;
    if (h < NH) { const int r = h / a.NS, jj = h - r * a.NS; v = a.G[((size_t)r * kDim + dim) * a.NS + jj]; }
    Gs[i] = v;
  }
  for (int i = threadIdx.x; i < a.R * kDim; i += 256) {
    const int r = i >> 7, dim = i & 127;
    const uint8_t code = (uint8_t)(m.H[i] & 15) | (m.S[i] < 0.f ? 0x80u : 0u);
    codeT[r * 128 + (dim & 3) * 32 + (dim >> 2)] = code;
  }

  for (int64_t page = blockIdx.x; page < m.n_pages; page += gridDim.x) {
    int64_t r0 = 0;
    int32_t nr;
    if (a.x_f32) {
      if (a.row_offsets) {
        r0 = a.row_offsets[page];
        nr = (int32_t)(a.row_offsets[page + 1] - r0);
      } else {  // ONE page of `stride` rows (the query)
        nr = a.stride;
      }
    } else {
      nr = a.n_rows ? a.n_rows[page] : a.stride;
      r0 = page * (int64_t)a.stride;
    }
    __syncthreads();  // tables staged / previous page's finish done with acc
    for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < a.R * NP; i += 256) cnt[i] = 0;
    __syncthreads();

    const int ntiles = (nr + 15) >> 4;
    for (int t = wave; t < ntiles; t += 4) {
      const int row0 = t * 16;
      // ---- stage 16 rows x 128 dims as fp32 (rows >= nr are zero)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < nr) {
          const int64_t e = (r0 + row0 + row) * kDim + c4;
          if (a.x_f32) {
            v = *reinterpret_cast<const float4*>(a.x_f32 + e);
          } else {
            const uint2 hh = *reinterpret_cast<const uint2*>(a.x_bf16 + e);
            v.x = bf16_to_f32((uint16_t)(hh.x & 0xffff)); v.y = bf16_to_f32((uint16_t)(hh.x >> 16));
            v.z = bf16_to_f32((uint16_t)(hh.y & 0xffff)); v.w = bf16_to_f32((uint16_t)(hh.y >> 16));
          }
        }
        float* d = xs + row * kXStride + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private LDS: in-order, just make the stores land
      // ---- A fragments: x[row = j][dim = 4s + k]
      float af[32];
#pragma unroll
      for (int s = 0; s < 32; ++s) af[s] = xs[j * kXStride + 4 * s + k];
      // ---- SimHash sketches, one 16-hash column tile at a time; only the signs are kept
      for (int n = 0; n < NT; ++n) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 32; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], Gs[(4 * s + k) * NHP + 16 * n + j], c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) sg[(4 * k + i) * NHP + 16 * n + j] = c[i] > 0.0f ? 1 : 0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- partition ids (Gray code of the NS sign bits) for the 16 x R (row, repetition) pairs
      for (int pi = lane; pi < 16 * a.R; pi += 64) {
        const int row = pi & 15, rep = pi >> 4;
        uint32_t part = 0;
        for (int jj = 0; jj < a.NS; ++jj) part = (part << 1) + ((uint32_t)sg[row * NHP + rep * a.NS + jj] ^ (part & 1u));
        pt[row * a.R + rep] = (uint8_t)part;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- AMS projection per repetition (dense {0,+1,-1} column against the same A fragments) + bucket sums
      for (int rep = 0; rep < a.R; ++rep) {
        const uint4 c0 = *reinterpret_cast<const uint4*>(codeT + rep * 128 + k * 32);
        const uint4 c1 = *reinterpret_cast<const uint4*>(codeT + rep * 128 + k * 32 + 16);
        const uint32_t cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 32; ++s) {
          const uint32_t code = (cw[s >> 2] >> (8 * (s & 3))) & 0xffu;
          const float b = ((int)(code & 15u) == j) ? ((code & 0x80u) ? -1.0f : 1.0f) : 0.0f;
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], b, c, 0, 0, 0);
        }
        if (j < a.PD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = 4 * k + i;
            if (row0 + row < nr) {
              const int part = pt[row * a.R + rep];
              atomicAdd(&acc[((size_t)rep * NP + part) * a.PD + j], c[i] * a.scale);
              if (j == 0) atomicAdd(&cnt[rep * NP + part], 1);
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- finish (as fde_encode_kernel): AVERAGE for documents, write fp32 / bf16, inverse norm of the bf16 image
    float nn = 0.0f;
    for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) {
      float v = acc[i];
      if (!a.is_query) {
        const int n = cnt[i / a.PD];
        if (n > 1) v = v / (float)n;
      }
      if (a.out_f32) a.out_f32[page * a.out_dim + i] = v;
      const uint16_t hb = f32_to_bf16_rne(v);
      if (a.out_bf16) a.out_bf16[page * a.out_dim + i] = hb;
      const float vb = bf16_to_f32(hb);
      nn += vb * vb;
    }
    if (a.out_inv_norm) {
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) nn += __shfl_xor(nn, sft);
      if (lane == 0) red[wave] = nn;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float tt = (red[0] + red[1]) + (red[2] + red[3]);
        a.out_inv_norm[page] = tt > 0.0f ? 1.0f / sqrtf(tt) : 0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------ documents from the bf16 slab: AMS on the bf16 matrix pipe
// The corpus build encodes pages that are ALREADY bf16 (the slab), and the AMS projection is a {0, +1, -1} matrix: every product
// x * (+-1) is exact in bf16 x bf16 -> fp32, so the projection can ride v_mfma_f32_16x16x32_bf16 -- K = 128 in FOUR 16-cycle MFMAs per
// repetition instead of thirty-two 32-cycle v_mfma_f32_16x16x4_f32 (16 x fewer matrix cycles for the part that was 74 % of them).  Only the
// summation order inside a 32-wide K block differs from the oracle's ascending chain: the projections agree to fp32 rounding (~1e-7),
// far inside the bf16 rounding of the slab they are stored in.  The SimHash sketches stay the k-ordered fp32 fmaf chains of the other
// kernels (v_mfma_f32_16x16x4_f32) -> identical sign bits -> identical partitions, bit for bit; what changes there is the feeding: the
// Gaussian columns of all NT column tiles live in VGPRs (NT x 32 registers per lane, loop invariant; one wave per SIMD has 512) instead
// of one ds_read + s_waitcnt in front of every MFMA, and the NT chains are interleaved so no MFMA waits for its predecessor.
// The next tile's rows are requested before the current tile's arithmetic (one wave per SIMD: nobody else hides the latency).
// LDS: AMS operand table [R][4 K steps][64 lanes][8 bf16] (80 KiB at R = 20) + acc (40 KiB) + cnt + per-wave x tile / signs / partitions.
constexpr int kXStrideB = 132;  // bf16 elements per staged row of the document kernel (264 B)

template <int NT>
__global__ __launch_bounds__(256) void fde_encode_doc_kernel(EncMArgs m) {
  const EncArgs& a = m.e;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NP = 1 << a.NS;
  const int NH = a.R * a.NS;
  constexpr int NHP = NT * 16;
  uint16_t* Bt = reinterpret_cast<uint16_t*>(smem);                                  // [R][4][64][8] bf16 AMS operand fragments
  float* acc = reinterpret_cast<float*>(smem + (size_t)a.R * 4096);                  // [out_dim]
  int32_t* cnt = reinterpret_cast<int32_t*>(acc + a.out_dim);                        // [R*NP]
  float* red = reinterpret_cast<float*>(cnt + a.R * NP);                             // [4]
  uint16_t* xs_all = reinterpret_cast<uint16_t*>(red + 4);                           // [4 waves][16][kXStrideB] the tile's rows, bf16 as they come
  uint8_t* sg_all = reinterpret_cast<uint8_t*>(xs_all + 4 * 16 * kXStrideB);         // [4 waves][16][NHP] sign bytes
  uint8_t* pt_all = sg_all + 4 * 16 * NHP;                                           // [4 waves][16][R] partitions

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, k = lane >> 4;
  uint16_t* xs = xs_all + wave * 16 * kXStrideB;
  uint8_t* sg = sg_all + wave * 16 * NHP;
  uint8_t* pt = pt_all + wave * 16 * a.R;

  // ---- once per block: the AMS operand fragments (lane (c = l & 15, g = l >> 4) of K step kk holds column c of dims kk*32 + 8g .. +8)
  for (int i = threadIdx.x; i < a.R * 4 * 64; i += 256) {
    const int l = i & 63, kk = (i >> 6) & 3, r = i >> 8;
    const int c = l & 15, g = l >> 4;
    uint16_t* dst = Bt + (size_t)i * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dim = kk * 32 + g * 8 + e;
      const bool hit = (m.H[r * kDim + dim] & 15) == c && c < a.PD;
      dst[e] = hit ? (m.S[r * kDim + dim] < 0.f ? (uint16_t)0xbf80u : (uint16_t)0x3f80u) : (uint16_t)0;  // -1.0 / +1.0 / 0 in bf16
    }
  }
  // ---- once per wave: the SimHash columns, in registers: greg[n][s] = G[dim 4s + k][hash 16n + j]
  float greg[NT][32];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int h = 16 * n + j;
    const int r = h / a.NS, jj = h - r * a.NS;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) greg[n][s2] = h < NH ? a.G[((size_t)r * kDim + (4 * s2 + k)) * a.NS + jj] : 0.f;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) asm volatile("" : "+v"(greg[n][s2]));  // loop invariant: keep them where they are

  for (int64_t page = blockIdx.x; page < m.n_pages; page += gridDim.x) {
    const int32_t nr = a.n_rows ? a.n_rows[page] : a.stride;
    const uint16_t* pg = a.x_bf16 + (size_t)page * (size_t)a.stride * kDim;
    __syncthreads();  // tables staged / previous page's finish done with acc
    for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < a.R * NP; i += 256) cnt[i] = 0;
    __syncthreads();

    const int ntiles = (nr + 15) >> 4;
    // rows of a tile as the two kernels need them: 8 x (4 bf16) per lane for the fp32 staging, 4 x (8 bf16) A fragments for the bf16 MFMA
    uint2 nx[8];
    u32x4 nab[4];
    auto request = [&](int t) {
      const int row0 = t * 16;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
        nx[it] = row0 + row < nr ? *reinterpret_cast<const uint2*>(pg + (size_t)(row0 + row) * kDim + c4) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        nab[kk] = row0 + j < nr ? *reinterpret_cast<const u32x4*>(pg + (size_t)(row0 + j) * kDim + kk * 32 + k * 8) : u32x4{0u, 0u, 0u, 0u};
    };
    if (wave < ntiles) request(wave);
    for (int t = wave; t < ntiles; t += 4) {
      const int row0 = t * 16;
      // ---- this tile's rows (requested one tile ago) -> fp32 staging + bf16 A fragments; then ask for the next tile's
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
        *reinterpret_cast<uint2*>(xs + row * kXStrideB + c4) = nx[it];  // 264-byte rows: 8-byte aligned, conflict-free column reads
      }
      bf16x8 abf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) abf[kk] = __builtin_bit_cast(bf16x8, nab[kk]);
      if (t + 4 < ntiles) request(t + 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private LDS: in-order, just make the stores land
      float af[32];  // A fragments of the fp32 chain: x[row = j][dim = 4s + k]
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) af[s2] = bf16_to_f32(xs[j * kXStrideB + 4 * s2 + k]);
      // ---- SimHash sketches: NT independent k-ordered chains, interleaved; only the signs are kept
      f32x4 c[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2)
#pragma unroll
        for (int n = 0; n < NT; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s2], greg[n][s2], c[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) sg[(4 * k + i) * NHP + 16 * n + j] = c[n][i] > 0.0f ? 1 : 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- partition ids (Gray code of the NS sign bits) for the 16 x R (row, repetition) pairs, stored [repetition][row]: a lane's
      //      four rows of one repetition are ONE dword
      for (int pi = lane; pi < 16 * a.R; pi += 64) {
        const int row = pi & 15, rep = pi >> 4;
        uint32_t part = 0;
        for (int jj = 0; jj < a.NS; ++jj) part = (part << 1) + ((uint32_t)sg[row * NHP + rep * a.NS + jj] ^ (part & 1u));
        pt[rep * 16 + row] = (uint8_t)part;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- AMS projection per repetition on the bf16 pipe + bucket sums.  One wave per SIMD: nothing hides an LDS round trip, so
      //      the loop is software-pipelined -- the NEXT repetition's four operand fragments and its partition dword are requested
      //      before this repetition's MFMAs and atomics (LDS returns in order: the atomics queue behind the reads and nobody waits
      //      for them).  Measured before: 8 exposed round trips per repetition, waves waiting 78 % of their cycles.
      const bf16x8* bt0 = reinterpret_cast<const bf16x8*>(Bt) + lane;
      const uint32_t* ptw = reinterpret_cast<const uint32_t*>(pt) + k;  // rows 4k .. 4k+3 of repetition r at dword r*4 + k
      bf16x8 nb[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) nb[kk] = bt0[kk * 64];
      uint32_t npw = ptw[0];
      bool rv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[i] = row0 + 4 * k + i < nr;
      for (int rep = 0; rep < a.R; ++rep) {
        bf16x8 cb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) cb[kk] = nb[kk];
        const uint32_t pw = npw;
        if (rep + 1 < a.R) {
          const bf16x8* btn = bt0 + (size_t)(rep + 1) * 256;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) nb[kk] = btn[kk * 64];
          npw = ptw[(rep + 1) * 4];
        }
        f32x4 pj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pj = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abf[kk], cb[kk], pj, 0, 0, 0);
        if (j < a.PD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (rv[i]) {
              const int part = (int)((pw >> (8 * i)) & 0xffu);
              atomicAdd(&acc[((size_t)rep * NP + part) * a.PD + j], pj[i] * a.scale);
              if (j == 0) atomicAdd(&cnt[rep * NP + part], 1);
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- finish (as fde_encode_kernel): AVERAGE for documents, write fp32 / bf16, inverse norm of the bf16 image
    float nn = 0.0f;
    for (int i = threadIdx.x; i < (int)a.out_dim; i += 256) {
      float v = acc[i];
      if (!a.is_query) {
        const int n = cnt[i / a.PD];
        if (n > 1) v = v / (float)n;
      }
      if (a.out_f32) a.out_f32[page * a.out_dim + i] = v;
      const uint16_t hb = f32_to_bf16_rne(v);
      if (a.out_bf16) a.out_bf16[page * a.out_dim + i] = hb;
      const float vb = bf16_to_f32(hb);
      nn += vb * vb;
    }
    if (a.out_inv_norm) {
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) nn += __shfl_xor(nn, sft);
      if (lane == 0) red[wave] = nn;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float tt = (red[0] + red[1]) + (red[2] + red[3]);
        a.out_inv_norm[page] = tt > 0.0f ? 1.0f / sqrtf(tt) : 0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------ documents, round 4: two passes, no LDS atomics
// The one-pass kernel above spends its time behind LDS: 100 float atomics per 16-row tile (the bucket sums), the sign / partition
// bytes bounced through LDS, and ONE wave per SIMD (224 VGPRs of SimHash columns + 128 KiB of LDS tables) to hide none of it
// (profiles/r3/pmc_fde_encode_kernels_r3d.json: matrix pipes 11 % busy, waves in s_waitcnt half their cycles; 1.94 us per page).
// Split by what each half needs:
//   pass 1, fde_hash_kernel      the SimHash sketches (the k-ordered fp32 fmaf chains on v_mfma_f32_16x16x4_f32, Gaussian columns in
//                                VGPRs: bit-identical partitions) -> ONE byte per (row, repetition) in a scratch buffer (20 B per row);
//                                no tables in LDS, no accumulators, no barrier: a wave owns its tiles outright
//   pass 2, fde_project_kernel   a wave owns REPETITIONS (5 of 20), not rows: it walks every tile of the page, projects it with its
//                                repetitions' AMS columns (bf16 MFMA, operands loop-invariant in VGPRs) and adds the 16 projected rows
//                                to their buckets with a ONE-HOT matrix product on the f32 pipe:
//                                    acc[col][part] += sum_row PJ[row][col] * [part(row) == part]      (v_mfma_f32_16x16x4_f32, K = rows)
//                                The C layout of the projection (lane (col, k) holds rows 4k .. 4k+3) IS the A layout of that product
//                                when K step s is taken to mean rows {4k + s}: no lane crossing, and the one-hot B operand of the
//                                same step needs the partition bytes of rows 4k .. 4k+3 -- one dword of the scratch buffer.
//                                Bucket sums live in 40 VGPRs for the whole page (no LDS, no atomics, a fixed summation order:
//                                deterministic, where the atomics were not); row counts ride along as integer adds of the same
//                                compares.  ~190 VGPRs, 16 bytes of LDS: two workgroups per CU.
// Cost per 16-row tile: pass 1 224 f32 MFMAs on one wave; pass 2 (20 bf16 + 40 f32 MFMAs) on each of four waves.  The page is read
// twice from HBM / L2 (once per pass; the four waves of pass 2 share their loads through the L1).
template <int NT>
__global__ __launch_bounds__(256) void fde_hash_kernel(EncMArgs m, uint8_t* parts, int tiles_per_page) {
  const EncArgs& a = m.e;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NHP = NT * 16;
  __shared__ __attribute__((aligned(16))) char smem[4 * 16 * kXStrideB * 2 + 4 * 16 * NHP + 4 * 16 * 32];
  uint16_t* xs_all = reinterpret_cast<uint16_t*>(smem);                       // [4 waves][16][kXStrideB] the tile's rows, bf16 as they come
  uint8_t* sg_all = reinterpret_cast<uint8_t*>(xs_all + 4 * 16 * kXStrideB);  // [4 waves][16][NHP] sign bytes
  uint8_t* pt_all = sg_all + 4 * 16 * NHP;                                    // [4 waves][R <= 32][16] partitions
  const int NH = a.R * a.NS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, k = lane >> 4;
  uint16_t* xs = xs_all + wave * 16 * kXStrideB;
  uint8_t* sg = sg_all + wave * 16 * NHP;
  uint8_t* pt = pt_all + wave * 16 * 32;

  // once per wave: the SimHash columns, in registers: greg[n][s] = G[dim 4s + k][hash 16n + j]
  float greg[NT][32];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int h = 16 * n + j;
    const int r = h / a.NS, jj = h - r * a.NS;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) greg[n][s2] = h < NH ? a.G[((size_t)r * kDim + (4 * s2 + k)) * a.NS + jj] : 0.f;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) asm volatile("" : "+v"(greg[n][s2]));  // loop invariant: keep them where they are

  for (int64_t page = blockIdx.x; page < m.n_pages; page += gridDim.x) {
    const int32_t nr = a.n_rows ? a.n_rows[page] : a.stride;
    const uint16_t* pg = a.x_bf16 + (size_t)page * (size_t)a.stride * kDim;
    uint8_t* pp = parts + (size_t)page * (size_t)tiles_per_page * (size_t)a.R * 16;
    const int ntiles = (nr + 15) >> 4;
    uint2 nx[8];
    auto request = [&](int t) {
      const int row0 = t * 16;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
        nx[it] = row0 + row < nr ? *reinterpret_cast<const uint2*>(pg + (size_t)(row0 + row) * kDim + c4) : make_uint2(0u, 0u);
      }
    };
    if (wave < ntiles) request(wave);
    for (int t = wave; t < ntiles; t += 4) {
      const int row0 = t * 16;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
        *reinterpret_cast<uint2*>(xs + row * kXStrideB + c4) = nx[it];  // 264-byte rows: 8-byte aligned, conflict-free column reads
      }
      if (t + 4 < ntiles) request(t + 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private LDS: in-order, just make the stores land
      float af[32];  // A fragments of the fp32 chain: x[row = j][dim = 4s + k]
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) af[s2] = bf16_to_f32(xs[j * kXStrideB + 4 * s2 + k]);
      f32x4 c[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2)
#pragma unroll
        for (int n = 0; n < NT; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s2], greg[n][s2], c[n], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) sg[(4 * k + i) * NHP + 16 * n + j] = c[n][i] > 0.0f ? 1 : 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // partition ids (Gray code of the NS sign bits), stored [repetition][row]; rows past the page's end get 0xFF (no bucket)
      for (int pi = lane; pi < 16 * a.R; pi += 64) {
        const int row = pi & 15, rep = pi >> 4;
        uint32_t part = 0;
        for (int jj = 0; jj < a.NS; ++jj) part = (part << 1) + ((uint32_t)sg[row * NHP + rep * a.NS + jj] ^ (part & 1u));
        pt[rep * 16 + row] = row0 + row < nr ? (uint8_t)part : (uint8_t)0xFF;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint32_t* dst = reinterpret_cast<uint32_t*>(pp + (size_t)t * a.R * 16);
      for (int d = lane; d < a.R * 4; d += 64) dst[d] = reinterpret_cast<const uint32_t*>(pt)[d];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // pt is rewritten by this wave's next tile
    }
  }
}

template <int RPW>  // repetitions per wave: R <= 4 * RPW
__global__ __launch_bounds__(256, 2) void fde_project_kernel(EncMArgs m, const uint8_t* parts, int tiles_per_page) {
  const EncArgs& a = m.e;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, k = lane >> 4;
  const int NP = 1 << a.NS;
  const int rep0 = wave * RPW;

  // once per wave: the AMS operand fragments of its repetitions (lane (c = j, g = k) of K step kk holds column c of dims kk*32 + 8g .. +8)
  bf16x8 cb[RPW][4];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rep = rep0 + r, dim = kk * 32 + k * 8 + e;
        short x = 0;
        if (rep < a.R && j < a.PD && (m.H[rep * kDim + dim] & 15) == j) x = m.S[rep * kDim + dim] < 0.f ? (short)0xbf80 : (short)0x3f80;  // -1.0 / +1.0 in bf16
        v[e] = x;
      }
      cb[r][kk] = v;
    }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(cb[r][kk]));  // loop invariant

  int parity = 0;
  for (int64_t page = blockIdx.x; page < m.n_pages; page += gridDim.x, parity ^= 1) {
    const int32_t nr = a.n_rows ? a.n_rows[page] : a.stride;
    const uint16_t* pg = a.x_bf16 + (size_t)page * (size_t)a.stride * kDim;
    const uint8_t* pp = parts + (size_t)page * (size_t)tiles_per_page * (size_t)a.R * 16;
    const int ntiles = (nr + 15) >> 4;
    f32x4 acc[RPW][2];
    int32_t cnt[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { acc[r][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; cnt[r][mt] = 0; }
    u32x4 nab[4];
    uint32_t npw[RPW];
    auto request = [&](int t) {
      const int row = t * 16 + j;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        nab[kk] = row < nr ? *reinterpret_cast<const u32x4*>(pg + (size_t)row * kDim + kk * 32 + k * 8) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < RPW; ++r)  // partitions of rows 4k .. 4k+3 of the tile under repetition rep0 + r (0xFF: not a row)
        npw[r] = rep0 + r < a.R ? *reinterpret_cast<const uint32_t*>(pp + ((size_t)t * a.R + rep0 + r) * 16 + 4 * k) : 0xffffffffu;
    };
    if (ntiles > 0) request(0);
    for (int t = 0; t < ntiles; ++t) {
      bf16x8 abf[4];
      uint32_t pw[RPW];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) abf[kk] = __builtin_bit_cast(bf16x8, nab[kk]);
#pragma unroll
      for (int r = 0; r < RPW; ++r) pw[r] = npw[r];
      if (t + 1 < ntiles) request(t + 1);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        f32x4 pj = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pj = __builtin_amdgcn_mfma_f32_16x16x32_bf16(abf[kk], cb[r][kk], pj, 0, 0, 0);
        float as[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) as[s2] = pj[s2] * a.scale;  // lane (col j, k): row 4k + s2
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const uint32_t tgt = (uint32_t)(j + 16 * mt);
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const bool hit = ((pw[r] >> (8 * s2)) & 0xffu) == tgt;  // lane (partition tgt, k): does row 4k + s2 fall into it?
            acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[s2], hit ? 1.0f : 0.0f, acc[r][mt], 0, 0, 0);
            cnt[r][mt] += hit ? 1 : 0;
          }
        }
      }
    }
    // ---- finish: acc[r][mt][i] = bucket sum of (col 4k + i, partition j + 16 mt) under repetition rep0 + r; AVERAGE for documents
    float nn = 0.0f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int rep = rep0 + r;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        int32_t n = cnt[r][mt];
        n += __shfl_xor(n, 16);
        n += __shfl_xor(n, 32);  // rows of the page in partition j + 16 mt (the four k groups each counted their rows)
        const int part = j + 16 * mt;
        if (rep < a.R && part < NP) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[r][mt][i];
            if (!a.is_query && n > 1) v[i] = v[i] / (float)n;
          }
          const int64_t base = (int64_t)page * a.out_dim + ((int64_t)rep * NP + part) * a.PD + 4 * k;
          uint16_t hb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            hb[i] = f32_to_bf16_rne(v[i]);
            if (4 * k + i < a.PD) {
              const float vb = bf16_to_f32(hb[i]);
              nn += vb * vb;
              if (a.out_f32) a.out_f32[base + i] = v[i];
            }
          }
          if (a.out_bf16) {
            if (a.PD == 16) {
              *reinterpret_cast<uint2*>(a.out_bf16 + base) = make_uint2((uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16));
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (4 * k + i < a.PD) a.out_bf16[base + i] = hb[i];
            }
          }
        }
      }
    }
    if (a.out_inv_norm) {
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) nn += __shfl_xor(nn, sft);
      if (lane == 0) red[parity][wave] = nn;
      __syncthreads();  // one barrier per page: the sums of page p + 1 go to the other half of red
      if (threadIdx.x == 0) {
        const float tt = (red[parity][0] + red[parity][1]) + (red[parity][2] + red[parity][3]);
        a.out_inv_norm[page] = tt > 0.0f ? 1.0f / sqrtf(tt) : 0.0f;
      }
    }
  }
}

// ------------------------------------------------------------------------------ the QUERY: latency form
// One query page per request sits in front of every FDE search, so what matters is its latency, not throughput: the
// bulk kernel above is one persistent block that stages all 20 repetitions' tables (56 KiB) and walks 27 dependent
// 32-MFMA chains per row tile (~65 us for a 32-token query, measured).  Here every repetition gets its own block: it
// stages only its own 128 x NS SimHash columns and 128 AMS codes, and a wave runs TWO chains per 16-row tile (SimHash,
// AMS).  Same v_mfma_f32_16x16x4_f32 fmaf chains -> the same sign bits, partitions and products as the other kernels.
// SUM aggregation only (queries are never averaged, have no norm).
__global__ __launch_bounds__(256) void fde_encode_query_kernel(EncMArgs m) {
  const EncArgs& a = m.e;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  __shared__ float Gs[kDim * 16];                 // [128][16]: this repetition's NS hash columns, zero padded
  __shared__ __attribute__((aligned(16))) uint8_t codeT[128];  // code of dim 4s+k at [k][s]
  __shared__ float acc[(1 << kMaxNS) * kMaxPD];   // [NP][PD]
  __shared__ float xs_all[4 * 16 * kXStride];
  __shared__ uint8_t sg_all[4 * 16 * 16];
  const int NP = 1 << a.NS;
  const int rep = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, k = lane >> 4;
  float* xs = xs_all + wave * 16 * kXStride;
  uint8_t* sg = sg_all + wave * 256;
  const int nr = a.stride;  // ONE page of `stride` fp32 rows at x_f32 (blockIdx.y: query of a batch, packed [y][stride][128])
  const float* x_f32 = a.x_f32 + (size_t)blockIdx.y * (size_t)nr * kDim;
  const size_t out_base = (size_t)blockIdx.y * (size_t)a.out_dim;

  for (int i = threadIdx.x; i < kDim * 16; i += 256) {
    const int dim = i >> 4, h = i & 15;
    Gs[i] = h < a.NS ? a.G[((size_t)rep * kDim + dim) * a.NS + h] : 0.f;
  }
  if (threadIdx.x < kDim) {
    const int dim = threadIdx.x;
    codeT[(dim & 3) * 32 + (dim >> 2)] = (uint8_t)(m.H[rep * kDim + dim] & 15) | (m.S[rep * kDim + dim] < 0.f ? 0x80u : 0u);
  }
  for (int i = threadIdx.x; i < NP * a.PD; i += 256) acc[i] = 0.0f;
  __syncthreads();

  const int ntiles = (nr + 15) >> 4;
  for (int t = wave; t < ntiles; t += 4) {
    const int row0 = t * 16;
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // stage 16 rows x 128 dims (rows >= nr are zero)
      const int idx = it * 64 + lane, row = idx >> 5, c4 = (idx & 31) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + row < nr) v = *reinterpret_cast<const float4*>(x_f32 + (size_t)(row0 + row) * kDim + c4);
      float* d = xs + row * kXStride + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float af[32];  // A fragments: x[row = j][dim = 4s + k]
#pragma unroll
    for (int s = 0; s < 32; ++s) af[s] = xs[j * kXStride + 4 * s + k];
    // two independent chains, interleaved: SimHash sketches (hash column j) and the AMS projection (bucket j)
    const uint4 c0 = *reinterpret_cast<const uint4*>(codeT + k * 32);
    const uint4 c1 = *reinterpret_cast<const uint4*>(codeT + k * 32 + 16);
    const uint32_t cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      cs = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], Gs[(4 * s + k) * 16 + j], cs, 0, 0, 0);
      const uint32_t code = (cw[s >> 2] >> (8 * (s & 3))) & 0xffu;
      const float b = ((int)(code & 15u) == j) ? ((code & 0x80u) ? -1.0f : 1.0f) : 0.0f;
      cp = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], b, cp, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sg[(4 * k + i) * 16 + j] = cs[i] > 0.0f ? 1 : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (j < a.PD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * k + i;
        if (row0 + row < nr) {
          uint32_t part = 0;
          for (int jj = 0; jj < a.NS; ++jj) part = (part << 1) + ((uint32_t)sg[row * 16 + jj] ^ (part & 1u));  // AppendToGrayCode
          atomicAdd(&acc[(size_t)part * a.PD + j], cp[i] * a.scale);
        }
      }
    }
  }
  __syncthreads();
  const int rep_elems = NP * a.PD;
  for (int i = threadIdx.x; i < rep_elems; i += 256) {
    const float v = acc[i];
    if (a.out_f32) a.out_f32[out_base + (size_t)rep * rep_elems + i] = v;
    if (a.out_bf16) a.out_bf16[out_base + (size_t)rep * rep_elems + i] = f32_to_bf16_rne(v);
  }
}

// ------------------------------------------------------------------------------ coarse scan
struct ScanArgs {
  const uint16_t* fde;
  const float* inv_norm;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const float* q;
  float* scores;
  int64_t n;
  int32_t out_dim;
  uint32_t* hist0;  // nullable: the selection's first radix histogram, accumulated here (see FdeScanArgs)
  uint32_t* work;   // nullable: {next chunk, waves done}, both 0 between launches (LDS-DMA form: dynamic chunk claiming)
};

// Persistent waves: each lane keeps its slice of the query FDE in registers (ITERS x 8 floats) and
// streams pages; one page = ITERS coalesced 1 KiB wave loads.  out_dim = ITERS * 512.
// The cross-check of the default form (fde_scan_rows_kernel): the SAME arithmetic order -- the row in WPR parts of ITERS / WPR
// chunks, one accumulator and one xor butterfly per part, parts summed pairwise -- on plain nt loads by one wave per page.
template <int ITERS, int WPR>
__global__ __launch_bounds__(256) void fde_scan_kernel(ScanArgs a) {
  constexpr int CPW = ITERS / WPR;
  static_assert(ITERS % WPR == 0 && (WPR == 2 || WPR == 4), "part shape");
  __shared__ uint32_t h0[2048];  // per-block share of the selection's first histogram (persistent blocks: zeroed / flushed once)
  if (a.hist0) {
    for (int i = threadIdx.x; i < 2048; i += 256) h0[i] = 0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float q[ITERS][8];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const float4 lo = *reinterpret_cast<const float4*>(a.q + it * 512 + lane * 8);
    const float4 hi = *reinterpret_cast<const float4*>(a.q + it * 512 + lane * 8 + 4);
    q[it][0] = lo.x; q[it][1] = lo.y; q[it][2] = lo.z; q[it][3] = lo.w;
    q[it][4] = hi.x; q[it][5] = hi.y; q[it][6] = hi.z; q[it][7] = hi.w;
  }
  for (int64_t p = wave; p < a.n; p += nwaves) {
    bool m = false;
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[p];
      m = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
    }
    if (m) {
      if (lane == 0) a.scores[p] = -INFINITY;
      continue;
    }
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const u32x4* row = reinterpret_cast<const u32x4*>(a.fde + p * (int64_t)a.out_dim) + lane;
    float part[WPR];
#pragma unroll
    for (int w = 0; w < WPR; ++w) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int it = w * CPW + c;
        const u32x4 v = __builtin_nontemporal_load(row + it * 64);
        const uint32_t x[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc = __builtin_fmaf(__uint_as_float(x[k] << 16), q[it][2 * k], acc);
          acc = __builtin_fmaf(__uint_as_float(x[k] & 0xffff0000u), q[it][2 * k + 1], acc);
        }
      }
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
      part[w] = acc;
    }
    const float total = WPR == 4 ? (part[0] + part[1]) + (part[2] + part[3]) : part[0] + part[1];
    if (lane == 0) {
      const float sc = a.inv_norm ? total * a.inv_norm[p] : total;
      a.scores[p] = sc;
      if (a.hist0) {
        const float s0 = sc + 0.0f;
        if (s0 == s0 && s0 != -INFINITY) atomicAdd(&h0[topk_ordered_u32(s0) >> 21], 1u);
      }
    }
  }
  if (a.hist0) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
      if (h0[i]) atomicAdd(&a.hist0[i], h0[i]);
  }
}

// ---- round 5 (default): the float scan's transport under the same arithmetic.
// The wave-per-page kernel above reads with plain nt loads into VGPRs and stops at the plain-nt ceiling of the box
// (6.6-6.85 TB/s); the float page scan's nt LDS-DMA ring streams 7.2-7.3 TB/s on the same HBM.  This kernel keeps the
// register kernel's arithmetic -- lane l owns elements [512 it + 8 l, +8) of every page, one accumulator, the same FMA
// order, the same xor butterfly: scores are BIT-IDENTICAL -- and replaces only the transport:
//   * `global_load_lds_dwordx4 ... nt` copies 1 KiB chunks (64 lanes x 16 B, linear) into a wave-private ring of D tiles of
//     CPT chunks; lane l reads back exactly the 16 B it requested (ds_read_b128, conflict-free), so the LDS is a FIFO that
//     holds D-1 tiles (12 KiB at CPT 4, D 4) in flight per wave at no VGPR cost; counted `s_waitcnt vmcnt`, no barrier.
//   * NOTHING but the DMAs touches vector memory inside the stream: a wave walks the corpus in chunks of ppw (<= 64)
//     consecutive pages and at each chunk start -- the ring is empty there anyway -- lane i evaluates the doc filter and
//     loads 1/norm of the chunk's i-th page; the live pages are a 64-bit ballot walked with s_ff1; lane i keeps page i's
//     score and the chunk ends with one store (+ one LDS histogram add) per lane.  A per-page load of inv_norm / doc_ord
//     would make hipcc drain vmcnt(0) -- the whole ring -- once per page.
//   * persistent workgroups (2 per CU), the query FDE in 8 x ITERS VGPRs per lane loaded once; chunks are CLAIMED from a
//     device counter (one atomic per chunk per wave).  Measured (profiles/r5): with a static partition of the pages this
//     kernel -- like the register kernel, like the float scan's persistent variant 14 -- stops at 6.7-6.8 TB/s while the
//     float scan's transport, whose workgroups are handed out by the dispatcher, streams 7.3 TB/s over the same 25.6 GB:
//     the CUs do not all stream at the same rate, and a static split runs at the pace of the slowest.
template <int N>
__device__ __forceinline__ void fde_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int CPT, int D>
__device__ __forceinline__ void fde_wait_left(int left) {  // all tiles issued; `left` (< D - 1) tiles are younger than the one needed
  if (D > 6 && left >= 5) fde_wait_vmcnt<5 * CPT>();
  else if (D > 5 && left == 4) fde_wait_vmcnt<4 * CPT>();
  else if (D > 4 && left == 3) fde_wait_vmcnt<3 * CPT>();
  else if (D > 3 && left == 2) fde_wait_vmcnt<2 * CPT>();
  else if (D > 2 && left == 1) fde_wait_vmcnt<1 * CPT>();
  else fde_wait_vmcnt<0>();
}

// STREAM_ONLY: the same ring, claims and waits without the read-back and the arithmetic (MV_CAL_FDE_SCAN_STREAM: what this
// kernel's transport alone sustains).
template <int ITERS, int CPT, int D, bool STREAM_ONLY = false>  // out_dim = 512 ITERS; tile = CPT chunks of 1 KiB; ITERS % CPT == 0; CPT * (D-1) <= 63
__global__ __launch_bounds__(256) void fde_scan_ldsdma_kernel(ScanArgs a, int ppw) {
  constexpr int TPP = ITERS / CPT;  // tiles per page
  constexpr int TILEB = CPT * 1024;
  static_assert(ITERS % CPT == 0 && CPT * (D - 1) <= 63 && (CPT == 2 || CPT == 4) && D >= 2 && D <= 8, "ring shape");
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[4 * D * TILEB + 8192];
  uint32_t* h0 = reinterpret_cast<uint32_t*>(lds + 4 * D * TILEB);  // per-block share of the selection's first histogram
  if (a.hist0) {
    for (int i = threadIdx.x; i < 2048; i += 256) h0[i] = 0;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = lds + wave * (D * TILEB);
  const int voff = lane * 16;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

  float q[ITERS][8];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const float4 lo = *reinterpret_cast<const float4*>(a.q + it * 512 + lane * 8);
    const float4 hi = *reinterpret_cast<const float4*>(a.q + it * 512 + lane * 8 + 4);
    q[it][0] = lo.x; q[it][1] = lo.y; q[it][2] = lo.z; q[it][3] = lo.w;
    q[it][4] = hi.x; q[it][5] = hi.y; q[it][6] = hi.z; q[it][7] = hi.w;
  }
  // waited for HERE, once: left alone hipcc sinks the wait to the first FMA inside the stream, where its counted vmcnt
  // would drain the DMA ring on every iteration
#pragma unroll
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(q[it][k]));

  const int64_t nchunks = (a.n + ppw - 1) / ppw;  // a chunk = ppw consecutive pages, streamed by ONE wave
  const size_t page_bytes = (size_t)a.out_dim * 2;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int64_t c = (int64_t)blockIdx.x * 4 + wave;  // static order (no work counter): chunks g, g + nwaves, ...
  for (;; c += nwaves) {
    if (a.work) {  // dynamic: the next unclaimed chunk (a CU that streams faster takes more of them -- see the header comment)
      uint32_t t = 0;
      if (lane == 0) t = atomicAdd(a.work, 1u);
      c = (int64_t)__builtin_amdgcn_readfirstlane(t);
    }
    if (c >= nchunks) break;
    // ---- chunk prologue (ring empty): lane i <-> the chunk's i-th page
    const int64_t base = c * (int64_t)ppw;
    const int64_t myp = base + (int64_t)lane;
    const bool valid = lane < ppw && myp < a.n;
    bool masked = false;
    float my_inv = 1.0f;
    if (valid) {
      if (a.doc_ord) {
        const int32_t o = a.doc_ord[myp];
        masked = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
      }
      if (a.inv_norm && !masked) my_inv = a.inv_norm[myp];
    }
    asm volatile("" : "+v"(my_inv));  // loaded and waited for before the stream starts
    float my_score = -INFINITY;
    const uint64_t live = __ballot(valid && !masked);
    uint64_t iss = live, cons = live;
    int to_issue = __builtin_popcountll(live) * TPP;  // tiles not yet requested
    int to_read = to_issue;                             // tiles not yet consumed
    int iss_t = 0, iss_slot = 0, cons_slot = 0;

    auto issue_next = [&]() {
      const int i = __builtin_ctzll(iss);
      const char* tp = reinterpret_cast<const char*>(a.fde) + (size_t)(base + (int64_t)i) * page_bytes + (size_t)iss_t * TILEB;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + iss_slot * TILEB));
      uint32_t keep;
      // M0 = wave-uniform LDS slot; the instruction offset walks BOTH addresses; s_nop 4: SALU-write -> VMEM-read of the
      // base SGPRs and M0-write -> LDS-DMA (as in mv_maxsim.hip)
      if (CPT == 4) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %3 nt\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(slot), "s"(tpu)
            : "memory");
      } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %3 nt\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(slot), "s"(tpu)
            : "memory");
      }
      iss_slot = (iss_slot + 1 == D) ? 0 : iss_slot + 1;
      if (++iss_t == TPP) {
        iss_t = 0;
        iss &= iss - 1;
      }
      --to_issue;
    };

#pragma unroll
    for (int k = 0; k < D - 1; ++k)
      if (to_issue > 0) issue_next();

    while (cons) {
      const int i = __builtin_ctzll(cons);
      cons &= cons - 1;
      float acc = 0.0f;
#pragma unroll
      for (int t = 0; t < TPP; ++t) {
        if (to_issue > 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: the last reads of the slot being refilled
          issue_next();
          fde_wait_vmcnt<CPT * (D - 1)>();
        } else {
          fde_wait_left<CPT, D>(to_read - 1);
        }
        --to_read;
        const char* slot = ring + cons_slot * TILEB + voff;
        cons_slot = (cons_slot + 1 == D) ? 0 : cons_slot + 1;
        if (STREAM_ONLY) continue;
#pragma unroll
        for (int ch = 0; ch < CPT; ++ch) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(slot + ch * 1024);
          const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc = __builtin_fmaf(__uint_as_float(w[k] << 16), q[t * CPT + ch][2 * k], acc);
            acc = __builtin_fmaf(__uint_as_float(w[k] & 0xffff0000u), q[t * CPT + ch][2 * k + 1], acc);
          }
        }
      }
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
      if (lane == i) my_score = a.inv_norm ? acc * my_inv : acc;
    }

    // ---- chunk epilogue: one store and one histogram add per lane
    if (valid) {
      a.scores[myp] = my_score;
      if (a.hist0) {
        const float s0 = my_score + 0.0f;
        if (s0 == s0 && s0 != -INFINITY) atomicAdd(&h0[topk_ordered_u32(s0) >> 21], 1u);
      }
    }
  }
  if (a.work && lane == 0) {  // the last wave to leave re-arms the counters for the next launch on this stream
    const uint32_t done = atomicAdd(a.work + 1, 1u);
    if (done == (uint32_t)nwaves - 1u) {
      __threadfence();
      a.work[0] = 0u;
      a.work[1] = 0u;
    }
  }
  if (a.hist0) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
      if (h0[i]) atomicAdd(&a.hist0[i], h0[i]);
  }
}

// ---- round 5, second form: ONE FRESH WORKGROUP PER ROW.
// The transport probe (csrc/mv_synth.hip: stream_probe_kernel; profiles/r5/stream_structure_*.json) says what shape of work the nt
// LDS-DMA streams fastest over a 25.6 GB FDE slab: fresh workgroups of ONE 20 KiB row each, handed out by the dispatcher in row
// order -- 7.18 TB/s, the float scan's rate -- against 6.75-6.95 for every persistent form (claimed or static, wave- or
// workgroup-owned chunks) and for fresh workgroups over larger units.  So: the row is split over the workgroup's waves (WPR waves
// per row, CPW chunks of 1 KiB each; 4 x 5 at 10 240 dims), every wave needs only ITS slice of the query FDE (8 CPW floats per
// lane: 40 VGPRs, read from L2 -- the price of fresh workgroups, 2 bytes of L2 traffic per byte of HBM), and up to eight such
// workgroups share a CU (5 KiB of LDS and <= 64 VGPRs each).  Per row: DMA issue -> query slice -> filter / norm -> one wait ->
// CPW x (ds_read_b128, 8 FMA) -> xor butterfly -> partial to LDS -> barrier -> ((p0 + p1) + (p2 + p3)) / |d|.
// Arithmetic order == fde_scan_kernel's (lane l: elements [512 c + 8 l, +8) of chunks c of ONE part, sequentially; butterfly per
// part; parts summed pairwise): scores are bit-identical across the two.
template <int CPW, int WPR>  // out_dim = 512 CPW WPR; CPW <= 5, WPR in {2, 4}; rows per workgroup = 4 / WPR
__global__ __launch_bounds__(256) void fde_scan_rows_kernel(ScanArgs a) {
  static_assert(CPW >= 1 && CPW <= 5 && (WPR == 2 || WPR == 4), "row shape");
  constexpr int RPB = 4 / WPR;
  // one __shared__ object only (see fde_scan_ldsdma_kernel)
  __shared__ __attribute__((aligned(16))) char lds[4 * CPW * 1024 + 64];
  float* part_sum = reinterpret_cast<float*>(lds + 4 * CPW * 1024);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int part = wave % WPR;
  const int64_t row = (int64_t)blockIdx.x * RPB + wave / WPR;
  const bool in_range = row < a.n;
  bool masked = !in_range;
  if (in_range && a.doc_ord) {
    const int32_t o = a.doc_ord[row];
    masked = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
  }
  char* slot = lds + wave * (CPW * 1024);
  const int voff = lane * 16;
  if (!masked) {
    const char* tp = reinterpret_cast<const char*>(a.fde) + (size_t)row * (size_t)a.out_dim * 2 + (size_t)part * (CPW * 1024);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t m0a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot);
    uint32_t keep;
    // the instruction offset (12 bits here) walks BOTH addresses; the fifth chunk takes a second M0 and a +4 KiB lane offset
    if (CPW == 5) {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %4 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:3072 nt\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %5, %4 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff), "s"(m0a), "s"(m0a + 4096u), "s"(tpu), "v"(voff + 4096)
          : "memory");
    } else {
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %3 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff + c * 1024), "s"(m0a + (uint32_t)(c * 1024)), "s"(tpu)
            : "memory");
      }
    }
  }
  // this wave's slice of the query FDE (issued behind the DMAs: they are older in the vmcnt order, so the compiler's own counted
  // waits for these loads cover them too)
  float q[CPW][8];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float* qp = a.q + (size_t)(part * CPW + c) * 512 + lane * 8;
    const float4 lo4 = *reinterpret_cast<const float4*>(qp);
    const float4 hi4 = *reinterpret_cast<const float4*>(qp + 4);
    q[c][0] = lo4.x; q[c][1] = lo4.y; q[c][2] = lo4.z; q[c][3] = lo4.w;
    q[c][4] = hi4.x; q[c][5] = hi4.y; q[c][6] = hi4.z; q[c][7] = hi4.w;
  }
  float inv = 1.0f;
  if (!masked && a.inv_norm && part == 0) inv = a.inv_norm[row];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs, query slice, norm: everything this row needs has landed
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  float acc = 0.0f;
  if (!masked) {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(slot + c * 1024 + voff);
      const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc = __builtin_fmaf(__uint_as_float(w[k] << 16), q[c][2 * k], acc);
        acc = __builtin_fmaf(__uint_as_float(w[k] & 0xffff0000u), q[c][2 * k + 1], acc);
      }
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
  }
  if (lane == 0) part_sum[wave] = acc;
  __syncthreads();
  if (part == 0 && lane == 0 && in_range) {
    const float* p = part_sum + wave;
    const float t = WPR == 4 ? (p[0] + p[1]) + (p[2] + p[3]) : p[0] + p[1];
    a.scores[row] = masked ? -INFINITY : (a.inv_norm ? t * inv : t);
  }
}

// v2 coarse scan: the query FDE lives in LDS (40 KiB at 10 240 dims), not in 160 VGPRs per lane, so a wave
// needs only the registers of one page's loads (ITERS x 16 B per lane, all issued before the first use) and
// 16 waves fit a CU: 16 x 20 KiB = 320 KiB of HBM reads in flight per CU.  A 512-thread block stages the
// query once and then streams `pages_per_block` pages, one page per wave at a time.
// LDS image of the query: chunk (it, half) of lane l at ((it*2 + half)*64 + l)*16 B -> the two ds_read_b128
// per load are 16-byte strided across lanes (conflict-free).
template <int ITERS>
__global__ __launch_bounds__(512) void fde_scan_lds_kernel(ScanArgs a, int pages_per_block) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  for (int i = threadIdx.x; i < ITERS * 128; i += 512) {  // float4 index in the source order
    const int it = i >> 7, rem = i & 127, l = rem >> 1, half = rem & 1;
    const float4 v = *reinterpret_cast<const float4*>(a.q + (size_t)i * 4);
    *reinterpret_cast<float4*>(qs + ((size_t)((it * 2 + half) * 64 + l)) * 4) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t page0 = (int64_t)blockIdx.x * pages_per_block;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  for (int j = wave; j < pages_per_block; j += 8) {
    const int64_t p = page0 + j;
    if (p >= a.n) break;
    bool m = false;
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[p];
      m = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
    }
    if (m) {
      if (lane == 0) a.scores[p] = -INFINITY;
      continue;
    }
    const u32x4* row = reinterpret_cast<const u32x4*>(a.fde + p * (int64_t)a.out_dim) + lane;
    u32x4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) v[it] = row[it * 64];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const float4 qlo = *reinterpret_cast<const float4*>(qs + ((it * 2 + 0) * 64 + lane) * 4);
      const float4 qhi = *reinterpret_cast<const float4*>(qs + ((it * 2 + 1) * 64 + lane) * 4);
      const float qq[8] = {qlo.x, qlo.y, qlo.z, qlo.w, qhi.x, qhi.y, qhi.z, qhi.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_fmaf(__uint_as_float(v[it][k] << 16), qq[2 * k], acc[k]);
        acc[k] = __builtin_fmaf(__uint_as_float(v[it][k] & 0xffff0000u), qq[2 * k + 1], acc[k]);
      }
    }
    float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) t += __shfl_xor(t, sft);
    if (lane == 0) a.scores[p] = a.inv_norm ? t * a.inv_norm[p] : t;
  }
}

// v2 coarse scan: a 256-thread workgroup walks pages TOGETHER (wave w owns 1 KiB chunks w, w+4, ... of every page),
// so the number of concurrent DRAM streams is the number of workgroups, not waves, and consecutive workgroups read
// consecutive 20 KiB pages.  Each wave keeps only its quarter of the query FDE in registers (CH x 8 floats), and two
// pages of loads in flight (double buffer).  Per-page partial sums are reduced inside the wave with DPP row rotates +
// readlane, parked in LDS, and combined in a fixed order (deterministic) once per batch of 16 pages.
template <int CTRL>
__device__ __forceinline__ float fde_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int CH>  // out_dim = CH * 2048
__global__ __launch_bounds__(256) void fde_scan_coop_kernel(ScanArgs a) {
  constexpr int NB = 16;
  __shared__ float part[2][NB][4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t G = gridDim.x, b = blockIdx.x;
  if (b >= a.n) return;
  const int64_t n_my = (a.n - b + G - 1) / G;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

  float q[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const float* qp = a.q + (size_t)(wave + 4 * c) * 512 + lane * 8;
    const float4 lo = *reinterpret_cast<const float4*>(qp);
    const float4 hi = *reinterpret_cast<const float4*>(qp + 4);
    q[c][0] = lo.x; q[c][1] = lo.y; q[c][2] = lo.z; q[c][3] = lo.w;
    q[c][4] = hi.x; q[c][5] = hi.y; q[c][6] = hi.z; q[c][7] = hi.w;
  }

  auto load = [&](u32x4 (&v)[CH], int64_t i) {
    const u32x4* row = reinterpret_cast<const u32x4*>(a.fde + (b + i * G) * (int64_t)a.out_dim) + wave * 64 + lane;
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = row[c * 256];
  };
  auto dot = [&](const u32x4 (&v)[CH]) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_fmaf(__uint_as_float(v[c][k] << 16), q[c][2 * k], acc[k]);
        acc[k] = __builtin_fmaf(__uint_as_float(v[c][k] & 0xffff0000u), q[c][2 * k + 1], acc[k]);
      }
    float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    t += fde_dpp<0x128>(t);  // row_ror:8
    t += fde_dpp<0x124>(t);
    t += fde_dpp<0x122>(t);
    t += fde_dpp<0x121>(t);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 48));
    return (r0 + r1) + (r2 + r3);
  };
  auto finalize = [&](int64_t k) {  // wave 0: pages k*NB .. of this workgroup
    const int64_t ii = k * NB + lane;
    if (wave == 0 && lane < NB && ii < n_my) {
      const int64_t p = b + ii * G;
      bool m = false;
      if (a.doc_ord) {
        const int32_t o = a.doc_ord[p];
        m = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
      }
      const float* pp = part[k & 1][lane];
      const float t = (pp[0] + pp[1]) + (pp[2] + pp[3]);
      a.scores[p] = m ? -INFINITY : (a.inv_norm ? t * a.inv_norm[p] : t);
    }
  };

  u32x4 va[CH], vb[CH];
  load(va, 0);
  for (int64_t i = 0; i < n_my; i += 2) {
    if (i + 1 < n_my) load(vb, i + 1);
    const float ta = dot(va);
    if (lane == 0) part[(i / NB) & 1][i % NB][wave] = ta;
    if (i + 2 < n_my) load(va, i + 2);
    if (i + 1 < n_my) {
      const float tb = dot(vb);
      if (lane == 0) part[((i + 1) / NB) & 1][(i + 1) % NB][wave] = tb;
    }
    if ((i + 2) % NB == 0 || i + 2 >= n_my) {
      __syncthreads();
      finalize(i / NB);
    }
  }
}

// generic fallback for out_dim not a multiple of 512 or too large for registers
__global__ __launch_bounds__(256) void fde_scan_generic_kernel(ScanArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (p >= a.n) return;
  if (a.doc_ord) {
    const int32_t o = a.doc_ord[p];
    if (o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u))) {
      if (lane == 0) a.scores[p] = -INFINITY;
      return;
    }
  }
  const uint16_t* row = a.fde + p * (int64_t)a.out_dim;
  float acc = 0.0f;
  for (int i = lane; i < a.out_dim; i += 64) acc = __builtin_fmaf(bf16_to_f32(row[i]), a.q[i], acc);
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) a.scores[p] = a.inv_norm ? acc * a.inv_norm[p] : acc;
}

// ------------------------------------------------------------------------------ batched coarse scan
// Up to 32 queries per pass over the FDE slab: the scan above reads 2 * out_dim bytes per page for ONE dot product, so
// under load the coarse stage is a GEMM  S[pages x 16] = F[pages x out_dim] . Qf^T  that is still HBM-bound (16 queries:
// ~16 flop/byte against a ~300 flop/byte ridge) -- sixteen searches for the slab traffic of one.
//
// Workgroup = 4 waves, page tile = 64 pages, ring slot = 64 pages x 256 dims (512 B per page, 32 KiB), 4 slots.
//   * transport: every wave DMAs a quarter of each slot (global_load_lds_dwordx4 nt, 2 pages x 512 B per instruction),
//     counted s_waitcnt, one barrier per slot -- three slots (96 KiB per CU) are in flight while one is consumed;
//   * arithmetic: v_mfma_f32_16x16x32_bf16, A = 16 pages x 32 dims from LDS (XOR-swizzled 16-byte chunks, conflict-free
//     ds_read_b128), B = 32 dims x 16 queries.  Wave w owns dims [64w, 64w+64) of every slot for all four page tiles
//     (K-split): its query fragments are 4 coalesced 1 KiB loads per slot from a fragment-ordered image in L2, issued
//     three slots ahead into a static 4-set register ring, and counted with the DMAs (all VMEM of the loop is inline
//     asm: the compiler's own vmcnt bookkeeping would drain the ring);
//   * the fp32 query FDE enters as bf16 hi + bf16 lo (two MFMAs per fragment): 16 mantissa bits, so the coarse scores
//     agree with the fp32-query scan above to ~1e-5 relative; the slab is bf16 either way;
//   * tile end: the four waves' partial sums meet in LDS and are added in a fixed order (deterministic), 64 x 16
//     scores leave as 256-byte rows.  Cosine rule / tombstones: in the tile epilogue of the FIN instantiations (metadata through
//     the DMA ring: a plain global load in this loop would make the compiler drain it), else by fde_batch_finish_kernel.
struct ScanBatchArgs {
  const char* fde;     // [n][out_dim] bf16
  const char* qfrag;   // fragment-ordered hi/lo image of the queries (fde_batch_qprep_kernel)
  float* scores;       // [n_queries][score_stride]
  int64_t score_stride;
  int64_t n;
  int32_t out_dim;
  int32_t n_queries;
  int32_t n_tiles;     // ceil(n / 64)
  const float* inv_norm;    // FIN kernels: the cosine rule and the tombstones are applied where the scores are written
  const int32_t* doc_ord;   // nullable (no tombstones)
};

constexpr int kFbPages = 64;
constexpr int kFbSlotBytes = kFbPages * 512;
constexpr int kFbSlots = 4;
constexpr int kFbRedStride = 68;  // floats per (wave, query) row of the tile-end reduction: 64 pages + 4 (16-byte skew)

// [nb][out_dim] fp32 -> image[kc][wave][e][query tile][hi|lo][lane] of 16-byte B fragments: lane (query qt*16 + (l&15),
// group l>>4) holds dims kc*256 + (2*wave + e)*32 + 8*(l>>4) .. +8 of its query; queries >= nb are zero.
__global__ __launch_bounds__(256) void fde_batch_qprep_kernel(const float* q, int nb, int out_dim, int nqt, uint16_t* image) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (kc, wave, e, qt, lane)
  const int lane = t & 63;
  int r = t >> 6;
  const int qt = r % nqt; r /= nqt;
  const int e = r & 1, w = (r >> 1) & 3, kc = r >> 3;
  if (kc * 256 >= out_dim) return;
  const int ql = qt * 16 + (lane & 15), g = lane >> 4;
  const int d0 = kc * 256 + (2 * w + e) * 32 + g * 8;
  uint16_t hi[8], lo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = ql < nb ? q[(size_t)ql * out_dim + d0 + i] : 0.0f;
    hi[i] = f32_to_bf16_rne(x);
    lo[i] = f32_to_bf16_rne(x - bf16_to_f32(hi[i]));
  }
  uint16_t* dst = image + ((size_t)(((kc * 4 + w) * 2 + e) * nqt + qt) * 2) * 512 + lane * 8;  // 512 bf16 = 1 KiB per fragment
#pragma unroll
  for (int i = 0; i < 8; ++i) { dst[i] = hi[i]; dst[512 + i] = lo[i]; }
}

// NQT query tiles of 16 (16 or 32 queries per pass).
template <int NQT, bool LO = true>
__global__ __launch_bounds__(256) void fde_scan_batch_kernel(ScanBatchArgs a) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NF = 4 * NQT;       // query fragments per slot and wave: (e, qt, hi|lo)
  constexpr int OPS = 8 + (LO ? NF : NF / 2);  // VMEM operations per slot and wave
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[kFbSlots * kFbSlotBytes + 4 * 16 * kFbRedStride * 4];
  float* red = reinterpret_cast<float*>(lds + kFbSlots * kFbSlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int KC = a.out_dim >> 8;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles)
  const int total = n_my * KC;                   // ring slots of this workgroup; KC % 4 == 0
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;

  // DMA source offsets: instruction i of this wave fills LDS bytes [(wave*8 + i) KiB, +1 KiB) of the slot = pages
  // pl, pl+1 (pl = wave*16 + 2i); lane -> page pl + (lane>>5), chunk position lane&31, which receives the page's
  // logical chunk (lane&31) ^ (page & 15).  The scalar base is 4 KiB below the tile so the offsets (which also carry
  // -1 KiB per instruction of a group of four, see issue()) stay positive.
  uint32_t src_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
    src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
  }
  const uint32_t q_off = (uint32_t)lane * 16u;

  bf16x8 qf[4][NF];  // [ring set][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

  int i_tile = b, i_kc = 0;  // issue-side position (advanced by issue_q: a slot's DMAs come first, its fragments second)
  auto issue_dma = [&](int slot_idx) {
    const int64_t page0 = (int64_t)i_tile * kFbPages;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * kFbSlotBytes + wave * 8192));
    uint32_t so[8];
    if (page0 + kFbPages > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
        so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %11 nt\n\t"
        "global_load_lds_dwordx4 %2, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %3, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %4, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %5, %11 nt\n\t"
        "global_load_lds_dwordx4 %6, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %7, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %8, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[4]), "v"(so[5]), "v"(so[6]), "v"(so[7]), "s"(slot),
          "s"(slot + 4096u), "s"(tpu)
        : "memory");
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // this wave's query fragments of the slot: NF KiB contiguous in the image, four per statement
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else  // bf16 query FDE (hi term only)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
    if (++i_kc == KC) { i_kc = 0; i_tile += G; }
  };

  // fragment read offsets inside a slot: page tile t, k-step kk = 2*wave + e -> page t*16 + p, logical chunk kk*4 + g
  uint32_t rd_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)p * 512u + (((uint32_t)((2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);

  f32x4 acc[4][NQT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) acc[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Four slots in flight.  Per slot: counted wait + barrier (the slot has landed for all waves) -> every wave pulls
  // its fragments into registers -> barrier (the slot is drained) -> its refill is issued AT ONCE, before the MFMAs:
  // a ring position idles for one LDS read, not for a slot's arithmetic plus the wait for the next slot's data.
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u < total) {
      issue_dma(u);
      issue_q(qf[u]);
    }
  }

  int c_tile = b, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u;
      // OPS VMEM operations per slot and wave (8 DMAs + the fragment loads), completed in issue order
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * OPS) : "memory");
      else if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("" : "+v"(qf[u][j]));  // uses stay behind the wait
      const char* slot = lds + u * kFbSlotBytes;
      bf16x8 af[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (16 * 512) + rd_off[e]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + 4 < total) issue_dma(u);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[u][(e * NQT + qt) * 2 + 0], acc[t][qt], 0, 0, 0);
            if (LO) acc[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[u][(e * NQT + qt) * 2 + 1], acc[t][qt], 0, 0, 0);
          }
      if (s + 4 < total) {
        // the register set is free after its last MFMA was issued; at most 63 VMEM operations may be outstanding
        if (4 * OPS > 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(63 - (OPS - 8)) : "memory");
        issue_q(qf[u]);
      }
      if (++c_kc == KC) {  // tile done: acc[t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p over this wave's dims
        const int pg = threadIdx.x & 63;
        const int64_t page = (int64_t)c_tile * kFbPages + pg;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * kFbRedStride + t * 16 + g * 4) = acc[t][qt];
            acc[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ql = (threadIdx.x >> 6) + 4 * j;
              if (qt * 16 + ql < a.n_queries) {
                const float v = (red[(0 * 16 + ql) * kFbRedStride + pg] + red[(1 * 16 + ql) * kFbRedStride + pg]) +
                                (red[(2 * 16 + ql) * kFbRedStride + pg] + red[(3 * 16 + ql) * kFbRedStride + pg]);
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten KC slots (>= 4 barriers) later
        c_kc = 0;
        c_tile += G;
      }
    }
  }
}

// ---- the same scan with page tiles processed in PAIRS per query fragment (default form)
// A wave's query fragments cost as much L2 -> register traffic per slot as the slot's share of the pages costs HBM traffic
// once 32 queries ride a pass (measured: 16 KiB of fragments per 32 KiB slot: 6.5 TB/s, 32 KiB: 5.7).  Here a workgroup
// walks TWO of its page tiles together, slot order (kc, tile 0), (kc, tile 1), (kc+1, tile 0), ...: the fragments of a
// K chunk are loaded once and used for both tiles -- half the fragment traffic, twice the accumulators.  A workgroup's
// odd tile out runs through the single-tile phase afterwards (same code, T = 1), so the tile -> workgroup map and the
// per-page arithmetic (order of the K chunks, of the four waves' partial sums) are those of the single-tile kernel:
// identical scores.
template <typename F, int... I>
__device__ __forceinline__ void fb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void fb_static_for(F&& f) {
  fb_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// n_groups groups of T tiles: tiles i0 + grp*T + j of this workgroup's list b, b + G, b + 2G, ...
// FIN: the finish of the pass (scores *= 1/|d|, -inf for tombstoned pages) happens where a tile's scores are written, instead of in
// a second pass over the [queries][pages] matrix (fde_batch_finish_kernel: 54 us of a 4.0 ms pass at 1.25 M pages x 32 queries).
// A tile's 64 norms and 64 document ordinals travel like its pages: two global_load_lds_dword per wave into a 512-byte LDS
// record, issued with the tile's slots of every FOURTH K chunk (kc & 3 == 0 is a compile-time property of the unrolled slot, so the
// counted vmcnt waits stay constants; the re-loads bring the same 512 bytes, +0.4 % requests), landed -- in issue order -- before that
// slot's own wait returns, read with ds_read at the tile's end.  Every wave issues them (same data, same place: the per-wave
// counts stay uniform).  Records are double-buffered by group parity: the next group's first slots are issued before this group's
// epilogue runs.  The arithmetic is the finish kernel's (one fp32 multiply of the same sum): identical scores.
// PRIV (MV_OPT_FDE_BATCH_VARIANT = 6): every wave DMAs exactly the bytes IT consumes -- its 64-dim quarter (128 B) of all 64 pages of the
// slot instead of the full 512 B of 16 pages -- into a ring of its own, so a slot needs no workgroup barrier at all: "landed" is the
// wave's own vmcnt, "consumed" its own lgkmcnt.  The waves meet only at a tile's end (the cross-wave sum).  Same fragments, same K
// order, same order of the four partial sums: identical scores.
template <int NQT, bool LO, int T, bool FIN = false, bool PRIV = false>
__device__ __forceinline__ void fb_phase(const ScanBatchArgs& a, char* lds, float* red, const int lane, const int wave, const int i0,
                                         const int n_groups, const uint32_t (&src_off)[8], const uint32_t (&rd_off)[2]) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NF = 4 * NQT;              // query fragments per K chunk and wave: (e, qt, hi|lo)
  constexpr int QOPS = LO ? NF : NF / 2;   // fragment loads per K chunk and wave
  constexpr int MOPS = FIN ? 2 : 0;        // metadata loads per slot of a K chunk with kc & 3 == 0
  char* meta = reinterpret_cast<char*>(red) + 4 * 16 * kFbRedStride * 4;  // FIN: [group parity][tile of the group][64 x 1/|d| | 64 x doc ordinal]
  if constexpr (FIN) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the previous phase's last epilogue has read its records
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int KC = a.out_dim >> 8;
  const int total = n_groups * KC * T;     // ring slots of this phase; a multiple of 4 * T
  if (total == 0) return;
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  const uint32_t q_off = (uint32_t)lane * 16u;

  bf16x8 qf[4][NF];  // [K chunk & 3][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[T][4][NQT];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int i_grp = 0, i_kc = 0, i_j = 0;  // issue-side position: (group, K chunk, tile of the group)
  auto issue_dma = [&](int slot_idx) {
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * kFbPages;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * kFbSlotBytes + wave * 8192));
    uint32_t so[8];
    if (page0 + kFbPages > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (PRIV) {
          const uint32_t pl = (uint32_t)(8 * i + (lane >> 3));
          so[i] = min(pl, last) * row_bytes + (uint32_t)wave * 128u + ((((uint32_t)lane & 7u) ^ ((pl >> 1) & 7u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
        } else {
          const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
          so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %11 nt\n\t"
        "global_load_lds_dwordx4 %2, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %3, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %4, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %5, %11 nt\n\t"
        "global_load_lds_dwordx4 %6, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %7, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %8, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[4]), "v"(so[5]), "v"(so[6]), "v"(so[7]), "s"(slot),
          "s"(slot + 4096u), "s"(tpu)
        : "memory");
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // the fragments of K chunk i_kc
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
  };
  auto issue_meta = [&]() {  // the slot being issued belongs to tile (i_grp, i_j): its norms and ordinals -> meta[i_grp & 1][i_j]
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * kFbPages;
    const float* ip = a.inv_norm + page0;
    const int32_t* op = (a.doc_ord ? a.doc_ord : reinterpret_cast<const int32_t*>(a.inv_norm)) + page0;
    const uint32_t ilo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ip);
    const uint32_t ihi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ip >> 32));
    const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)op);
    const uint32_t ohi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)op >> 32));
    const uint64_t ipu = ((uint64_t)ihi << 32) | ilo, opu = ((uint64_t)ohi << 32) | olo;
    const uint32_t rec = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(meta + ((i_grp & 1) * T + i_j) * 512));
    uint32_t vo = (uint32_t)lane * 4u;
    if (page0 + kFbPages > a.n) vo = min((uint32_t)lane, (uint32_t)(a.n - 1 - page0)) * 4u;  // last tile: lanes past the corpus re-read its last page
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %1, %4\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %1, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(vo), "s"(rec), "s"(rec + 256u), "s"(ipu), "s"(opu)
        : "memory");
  };
  auto advance = [&]() {
    if (++i_j == T) {
      i_j = 0;
      if (++i_kc == KC) { i_kc = 0; ++i_grp; }
    }
  };

  // prologue: slots 0..3 (slot x = K chunk x / T, tile x % T); a chunk's fragments ride with its first slot
  fb_static_for<4>([&](auto UC) {
    constexpr int u = decltype(UC)::value;
    issue_dma(u);
    if constexpr (FIN && ((u / T) & 3) == 0) issue_meta();
    if constexpr (u % T == 0) issue_q(qf[(u / T) & 3]);
    advance();
  });

  int c_grp = 0, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += 4 * T) {
    fb_static_for<4 * T>([&](auto UC) {
      constexpr int u = decltype(UC)::value;
      constexpr int j = u % T;            // tile of the group
      constexpr int kcs = (u / T) & 3;    // K chunk & 3 -> fragment register set
      const int s = s0 + u;
      // VMEM operations of the slots x behind this one (8 DMAs + the fragment loads of a chunk's first slot + the metadata loads of
      // the slots of every fourth chunk), in issue order
      constexpr int o1 = 8 + (((u + 1) % T == 0) ? QOPS : 0) + (((((u + 1) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      constexpr int o2 = 8 + (((u + 2) % T == 0) ? QOPS : 0) + (((((u + 2) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      constexpr int o3 = 8 + (((u + 3) % T == 0) ? QOPS : 0) + (((((u + 3) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2 + o3) : "memory");
      else if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2) : "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (!PRIV) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int x = 0; x < NF; ++x) asm volatile("" : "+v"(qf[kcs][x]));  // uses stay behind the wait
      const char* slot = lds + (u & 3) * kFbSlotBytes;
      bf16x8 af[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (PRIV ? 16 * 128 : 16 * 512) + rd_off[e]);
      if constexpr (PRIV) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's part of slot s is in its registers: refill it
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + 4 < total) {
        issue_dma(u & 3);
        if constexpr (FIN && ((((u + 4) % (4 * T)) / T) & 3) == 0) issue_meta();
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 0], acc[j][t][qt], 0, 0, 0);
            if (LO) acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 1], acc[j][t][qt], 0, 0, 0);
          }
      if (s + 4 < total) {
        if constexpr (j == 0) {  // slot s + 4 opens K chunk (u + 4) / T: its fragments go into that chunk's register set
          // (T = 1: the set the MFMAs above just read; at most 63 VMEM operations may be outstanding)
          constexpr int peak = 4 * 8 + ((T == 1) ? 4 : (T == 2 ? 2 : 1)) * QOPS + MOPS * T;
          if constexpr (peak > 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(63 - QOPS) : "memory");
          issue_q(qf[((u + 4) / T) & 3]);
        }
        advance();
      }
      if (c_kc == KC - 1) {  // tile j of the group is done: acc[j][t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p
        const int pg = threadIdx.x & 63;
        const int64_t tile = (int64_t)b + (int64_t)(i0 + c_grp * T + j) * G;
        const int64_t page = tile * kFbPages + pg;
        float inv = 1.0f;
        bool dead = false;
        if constexpr (FIN) {  // the record of this tile landed KC slots ago
          const char* rec = meta + ((c_grp & 1) * T + j) * 512;
          inv = reinterpret_cast<const float*>(rec)[pg];
          if (a.doc_ord) dead = reinterpret_cast<const int32_t*>(rec + 256)[pg] < 0;
        }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (PRIV || qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * kFbRedStride + t * 16 + g * 4) = acc[j][t][qt];
            acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int ql = (threadIdx.x >> 6) + 4 * x;
              if (qt * 16 + ql < a.n_queries) {
                float v = (red[(0 * 16 + ql) * kFbRedStride + pg] + red[(1 * 16 + ql) * kFbRedStride + pg]) +
                          (red[(2 * 16 + ql) * kFbRedStride + pg] + red[(3 * 16 + ql) * kFbRedStride + pg]);
                if constexpr (FIN) v = dead ? -INFINITY : v * inv;
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten by the next tile end: at least one slot barrier later
      }
      if constexpr (j == T - 1) {
        if (++c_kc == KC) { c_kc = 0; ++c_grp; }
      }
    });
  }
}

template <int NQT, bool LO, bool FIN = false, bool PRIV = false>
__global__ __launch_bounds__(256) void fde_scan_batch2_kernel(ScanBatchArgs a) {
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[kFbSlots * kFbSlotBytes + 4 * 16 * kFbRedStride * 4 + (FIN ? 2 * 2 * 512 : 0)];
  float* red = reinterpret_cast<float*>(lds + kFbSlots * kFbSlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles)
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  uint32_t src_off[8];  // see fde_scan_batch_kernel
  uint32_t rd_off[2];
  if constexpr (PRIV) {
    // DMA i of a slot: lane l fetches 16 B of page 8i + (l >> 3) -- piece (l & 7) ^ swizzle(page) of this wave's 128-byte quarter of the
    // row -- and the LDS write is lane-linear: page P sits at P * 128 of the wave's 8 KiB, its piece c at ((c ^ ((P >> 1) & 7)) << 4).
    // A ds_read_b128 serves 8 lanes a clock: pages p, p + 1 share a 256-byte bank row, the swizzle spreads the four pairs' pieces.
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t pl = (uint32_t)(8 * i + (lane >> 3));
      src_off[i] = pl * row_bytes + (uint32_t)wave * 128u + ((((uint32_t)lane & 7u) ^ ((pl >> 1) & 7u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)wave * 8192u + (uint32_t)p * 128u + (((uint32_t)(e * 4 + g) ^ (((uint32_t)p >> 1) & 7u)) << 4);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
      src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)p * 512u + (((uint32_t)((2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);
  }
  fb_phase<NQT, LO, 2, FIN, PRIV>(a, lds, red, lane, wave, 0, n_my / 2, src_off, rd_off);
  fb_phase<NQT, LO, 1, FIN, PRIV>(a, lds, red, lane, wave, (n_my / 2) * 2, n_my & 1, src_off, rd_off);
}

// ------------------------------------------------------------------------------------------------------------
// Round 3: the same phase with the page-tile height and the depth of the fragment register ring as parameters.
// PT = 16-page sub-tiles per page tile (4 = the 64-page tile above; 2 = 32 pages: a 16 KiB ring slot, so TWO workgroups fit a
// CU's LDS and a late slot stalls half a CU instead of all of it); NSETS = fragment register sets (a K chunk's fragments
// serve T slots, so T = 4 needs only the current and the next set: 64 instead of 128 VGPRs -- what lets two workgroups'
// waves share a SIMD's register file).  Arithmetic, K order and the order of the four waves' partial sums are those of
// fb_phase: identical scores.
// n_groups groups of T tiles: tiles i0 + grp*T + j of this workgroup's list b, b + G, b + 2G, ...
template <int NQT, bool LO, int T, int PT, int NSETS>
__device__ __forceinline__ void fb_phase_g(const ScanBatchArgs& a, char* lds, float* red, const int lane, const int wave, const int i0,
                                         const int n_groups, const uint32_t (&src_off)[2 * PT], const uint32_t (&rd_off)[2]) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NF = 4 * NQT;              // query fragments per K chunk and wave: (e, qt, hi|lo)
  constexpr int QOPS = LO ? NF : NF / 2;   // fragment loads per K chunk and wave
  constexpr int PAGES = 16 * PT;           // pages per tile
  constexpr int SLOTB = PAGES * 512;       // ring slot: PAGES pages x 256 dims
  constexpr int NDMA = 2 * PT;             // DMA instructions per slot and wave (2 pages x 512 B each)
  constexpr int REDS = PAGES + 4;          // floats per (wave, query) row of the tile-end reduction
  static_assert(PT == 2 || PT == 4, "page tile of 32 or 64 pages");
  static_assert(NSETS >= 2 && (4 % NSETS == 0) && (NSETS >= 4 / T + (T == 4 ? 1 : 0) || T == 1), "fragment ring too shallow for T");
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int KC = a.out_dim >> 8;
  const int total = n_groups * KC * T;     // ring slots of this phase; a multiple of 4 * T
  if (total == 0) return;
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  const uint32_t q_off = (uint32_t)lane * 16u;

  bf16x8 qf[NSETS][NF];  // [K chunk % NSETS][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < NSETS; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[T][PT][NQT];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int i_grp = 0, i_kc = 0, i_j = 0;  // issue-side position: (group, K chunk, tile of the group)
  auto issue_dma = [&](int slot_idx) {
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * PAGES;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * SLOTB + wave * (4 * PT * 512)));
    uint32_t so[NDMA];
    if (page0 + PAGES > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const uint32_t pl = (uint32_t)(wave * (4 * PT) + 2 * i + (lane >> 5));
        so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    if constexpr (PT == 4) {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %9\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %11 nt\n\t"
          "global_load_lds_dwordx4 %2, %11 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %3, %11 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %4, %11 offset:3072 nt\n\t"
          "s_mov_b32 m0, %10\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %5, %11 nt\n\t"
          "global_load_lds_dwordx4 %6, %11 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %7, %11 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %8, %11 offset:3072 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[NDMA > 4 ? 4 : 0]), "v"(so[NDMA > 4 ? 5 : 0]), "v"(so[NDMA > 4 ? 6 : 0]),
            "v"(so[NDMA > 4 ? 7 : 0]), "s"(slot), "s"(slot + 4096u), "s"(tpu)
          : "memory");
    } else {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %5\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %6 nt\n\t"
          "global_load_lds_dwordx4 %2, %6 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %3, %6 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %4, %6 offset:3072 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "s"(slot), "s"(tpu)
          : "memory");
    }
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // the fragments of K chunk i_kc
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
  };
  auto advance = [&]() {
    if (++i_j == T) {
      i_j = 0;
      if (++i_kc == KC) { i_kc = 0; ++i_grp; }
    }
  };

  // prologue: slots 0..3 (slot x = K chunk x / T, tile x % T); a chunk's fragments ride with its first slot
  fb_static_for<4>([&](auto UC) {
    constexpr int u = decltype(UC)::value;
    issue_dma(u);
    if constexpr (u % T == 0) issue_q(qf[(u / T) % NSETS]);
    advance();
  });

  int c_grp = 0, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += 4 * T) {
    fb_static_for<4 * T>([&](auto UC) {
      constexpr int u = decltype(UC)::value;
      constexpr int j = u % T;            // tile of the group
      constexpr int kcs = (u / T) % NSETS;  // K chunk -> fragment register set
      const int s = s0 + u;
      // VMEM operations of the slots x behind this one (8 DMAs + the fragment loads of a chunk's first slot), in issue order
      constexpr int o1 = NDMA + (((u + 1) % T == 0) ? QOPS : 0), o2 = NDMA + (((u + 2) % T == 0) ? QOPS : 0), o3 = NDMA + (((u + 3) % T == 0) ? QOPS : 0);
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2 + o3) : "memory");
      else if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2) : "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int x = 0; x < NF; ++x) asm volatile("" : "+v"(qf[kcs][x]));  // uses stay behind the wait
      const char* slot = lds + (u & 3) * SLOTB;
      bf16x8 af[2][PT];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (16 * 512) + rd_off[e]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + 4 < total) issue_dma(u & 3);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 0], acc[j][t][qt], 0, 0, 0);
            if (LO) acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 1], acc[j][t][qt], 0, 0, 0);
          }
      if (s + 4 < total) {
        if constexpr (j == 0) {  // slot s + 4 opens K chunk (u + 4) / T: its fragments go into that chunk's register set
          // (T = 1: the set the MFMAs above just read; at most 63 VMEM operations may be outstanding)
          constexpr int peak = 4 * NDMA + (4 / T) * QOPS;
          if constexpr (peak > 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(63 - QOPS) : "memory");
          issue_q(qf[((u + 4) / T) % NSETS]);
        }
        advance();
      }
      if (c_kc == KC - 1) {  // tile j of the group is done: acc[j][t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p
        const int pg = threadIdx.x & (PAGES - 1);
        const int64_t tile = (int64_t)b + (int64_t)(i0 + c_grp * T + j) * G;
        const int64_t page = tile * PAGES + pg;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * REDS + t * 16 + g * 4) = acc[j][t][qt];
            acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int x = 0; x < PAGES / 16; ++x) {
              const int ql = (int)(threadIdx.x / PAGES) + (256 / PAGES) * x;
              if (qt * 16 + ql < a.n_queries) {
                const float v = (red[(0 * 16 + ql) * REDS + pg] + red[(1 * 16 + ql) * REDS + pg]) +
                                (red[(2 * 16 + ql) * REDS + pg] + red[(3 * 16 + ql) * REDS + pg]);
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten by the next tile end: at least one slot barrier later
      }
      if constexpr (j == T - 1) {
        if (++c_kc == KC) { c_kc = 0; ++c_grp; }
      }
    });
  }
}


// ------------------------------------------------------------------------------------------------------------
// Round 4: the phase with the DEPTH of the DMA ring as a parameter.  The pass is latency-bound (SQ counters: MFMA 15 % busy, half the
// wave cycles waiting; removing the slot barriers altogether -- private rings, variant 6 -- changed nothing), i.e. its rate is the
// bytes a CU keeps in flight over the loaded HBM latency: 3 x 32 KiB with the 4-slot ring of 64-page slots.  Here: 32-page slots
// (16 KiB), ONE workgroup per CU, R slots -- R = 9 keeps 8 x 16 = 128 KiB in flight out of a 144 KiB ring.  The ring position is a
// run-time scalar (R need not divide the unroll period); what stays compile-time is the slot's place in its K chunk, which fixes the
// counted vmcnt waits: the operations of the R - 1 slots behind the one waited for, in issue order.  A K chunk's fragments are
// loaded R slots ahead with its first slot, so (R + T - 1) / T + 1 register sets are live: NSETS = 4 at T = 4, R = 9.
// Arithmetic, K order and the order of the four waves' partial sums are those of fb_phase: identical scores.
template <int NQT, bool LO, int T, int PT, int NSETS, int R>
__device__ __forceinline__ void fb_phase_r(const ScanBatchArgs& a, char* lds, float* red, const int lane, const int wave, const int i0,
                                         const int n_groups, const uint32_t (&src_off)[2 * PT], const uint32_t (&rd_off)[2]) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NF = 4 * NQT;              // query fragments per K chunk and wave: (e, qt, hi|lo)
  constexpr int QOPS = LO ? NF : NF / 2;   // fragment loads per K chunk and wave
  constexpr int PAGES = 16 * PT;           // pages per tile
  constexpr int SLOTB = PAGES * 512;       // ring slot: PAGES pages x 256 dims
  constexpr int NDMA = 2 * PT;             // DMA instructions per slot and wave (2 pages x 512 B each)
  constexpr int REDS = PAGES + 4;          // floats per (wave, query) row of the tile-end reduction
  constexpr int U = T * NSETS;             // unroll period: the slot's tile of the group and its chunk's register set are compile-time
  static_assert(PT == 2, "32-page tiles");
  static_assert((R + T - 1) / T + 1 <= NSETS, "fragment ring too shallow for this DMA ring");
  static_assert(R * NDMA + ((R + T - 1) / T) * QOPS <= 63, "more VMEM operations in flight than vmcnt counts");
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int KC = a.out_dim >> 8;
  const int total = n_groups * KC * T;     // ring slots of this phase; a multiple of U (KC % 4 == 0)
  if (total == 0) return;
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  const uint32_t q_off = (uint32_t)lane * 16u;

  bf16x8 qf[NSETS][NF];  // [K chunk % NSETS][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < NSETS; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[T][PT][NQT];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int i_grp = 0, i_kc = 0, i_j = 0;  // issue-side position: (group, K chunk, tile of the group)
  int i_pos = 0, c_pos = 0;          // ring position of the slot issued next / consumed next
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  auto issue_dma = [&]() {
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * PAGES;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)i_pos * (uint32_t)SLOTB + (uint32_t)wave * (uint32_t)(4 * PT * 512));
    uint32_t so[NDMA];
    if (page0 + PAGES > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const uint32_t pl = (uint32_t)(wave * (4 * PT) + 2 * i + (lane >> 5));
        so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %6 nt\n\t"
        "global_load_lds_dwordx4 %2, %6 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %3, %6 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %4, %6 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "s"(slot), "s"(tpu)
        : "memory");
    if (++i_pos == R) i_pos = 0;
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // the fragments of K chunk i_kc
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
  };
  auto advance = [&]() {
    if (++i_j == T) {
      i_j = 0;
      if (++i_kc == KC) { i_kc = 0; ++i_grp; }
    }
  };

  // prologue: slots 0 .. R-1 (slot x = K chunk x / T, tile x % T); a chunk's fragments ride with its first slot  (total >= U >= R is
  // not guaranteed for tiny phases: total is a multiple of U = 16 and R <= 12)
  static_assert(R <= U, "the prologue assumes one unroll period covers the ring");
  fb_static_for<R>([&](auto UC) {
    constexpr int u = decltype(UC)::value;
    issue_dma();
    if constexpr (u % T == 0) issue_q(qf[(u / T) % NSETS]);
    advance();
  });

  int c_grp = 0, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += U) {
    fb_static_for<U>([&](auto UC) {
      constexpr int u = decltype(UC)::value;
      constexpr int j = u % T;              // tile of the group
      constexpr int kcs = (u / T) % NSETS;  // K chunk -> fragment register set
      const int s = s0 + u;
      // VMEM operations of the R - 1 slots behind this one (NDMA DMAs + the fragment loads of a chunk's first slot), in issue order
      constexpr int behind = [] {
        int n = 0;
        for (int x = 1; x < R; ++x) n += NDMA + (((u + x) % T == 0) ? QOPS : 0);
        return n;
      }();
      if (s + R - 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(behind) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the phase's tail: everything left was issued >= one ring ago
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int x = 0; x < NF; ++x) asm volatile("" : "+v"(qf[kcs][x]));  // uses stay behind the wait
      const char* slot = lds + c_pos * SLOTB;
      bf16x8 af[2][PT];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (16 * 512) + rd_off[e]);
      if (++c_pos == R) c_pos = 0;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + R < total) issue_dma();  // into the slot just read (i_pos trails c_pos by one ring)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 0], acc[j][t][qt], 0, 0, 0);
            if (LO) acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 1], acc[j][t][qt], 0, 0, 0);
          }
      if (s + R < total) {
        if constexpr ((u + R) % T == 0) issue_q(qf[((u + R) / T) % NSETS]);  // slot s + R opens a K chunk: its fragments go into that chunk's set
        advance();
      }
      if (c_kc == KC - 1) {  // tile j of the group is done: acc[j][t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p
        const int pg = threadIdx.x & (PAGES - 1);
        const int64_t tile = (int64_t)b + (int64_t)(i0 + c_grp * T + j) * G;
        const int64_t page = tile * PAGES + pg;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * REDS + t * 16 + g * 4) = acc[j][t][qt];
            acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int x = 0; x < PAGES / 16; ++x) {
              const int ql = (int)(threadIdx.x / PAGES) + (256 / PAGES) * x;
              if (qt * 16 + ql < a.n_queries) {
                const float v = (red[(0 * 16 + ql) * REDS + pg] + red[(1 * 16 + ql) * REDS + pg]) +
                                (red[(2 * 16 + ql) * REDS + pg] + red[(3 * 16 + ql) * REDS + pg]);
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten by the next tile end: at least one slot barrier later
      }
      if constexpr (j == T - 1) {
        if (++c_kc == KC) { c_kc = 0; ++c_grp; }
      }
    });
  }
}

// One workgroup per CU, 32-page tiles, a DMA ring of R slots (MV_OPT_FDE_BATCH_VARIANT = 7: R = 9; 8: R = 4, the control).
template <int NQT, bool LO, int R>
__global__ __launch_bounds__(256) void fde_scan_batch5_kernel(ScanBatchArgs a) {
  constexpr int PT = 2;
  __shared__ __attribute__((aligned(16))) char lds[R * (16 * PT * 512) + 4 * 16 * (16 * PT + 4) * 4];
  float* red = reinterpret_cast<float*>(lds + R * (16 * PT * 512));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles; a.n_tiles counts 32-page tiles here)
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  uint32_t src_off[2 * PT];
#pragma unroll
  for (int i = 0; i < 2 * PT; ++i) {
    const uint32_t pl = (uint32_t)(wave * (4 * PT) + 2 * i + (lane >> 5));
    src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
  }
  uint32_t rd_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)p * 512u + (((uint32_t)((2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);
  const int n4 = n_my / 4, n2 = (n_my - 4 * n4) / 2, n1 = n_my - 4 * n4 - 2 * n2;
  fb_phase_r<NQT, LO, 4, PT, 4, R>(a, lds, red, lane, wave, 0, n4, src_off, rd_off);
  fb_phase_g<NQT, LO, 2, PT, 4>(a, lds, red, lane, wave, 4 * n4, n2, src_off, rd_off);  // the odd tiles out: the 4-slot ring
  fb_phase_g<NQT, LO, 1, PT, 4>(a, lds, red, lane, wave, 4 * n4 + 2 * n2, n1, src_off, rd_off);
}

// Two workgroups per CU, 32-page tiles, fragment ring of two sets at four tiles per K chunk (MV_OPT_FDE_BATCH_VARIANT = 4).
template <int NQT, bool LO>
__global__ __launch_bounds__(256, 2) void fde_scan_batch3_kernel(ScanBatchArgs a) {
  constexpr int PT = 2;
  __shared__ __attribute__((aligned(16))) char lds[4 * (16 * PT * 512) + 4 * 16 * (16 * PT + 4) * 4];
  float* red = reinterpret_cast<float*>(lds + 4 * (16 * PT * 512));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles; a.n_tiles counts 32-page tiles here)
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;
  uint32_t src_off[2 * PT];
#pragma unroll
  for (int i = 0; i < 2 * PT; ++i) {
    const uint32_t pl = (uint32_t)(wave * (4 * PT) + 2 * i + (lane >> 5));
    src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
  }
  uint32_t rd_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)p * 512u + (((uint32_t)((2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);
  const int n4 = n_my / 4, n2 = (n_my - 4 * n4) / 2, n1 = n_my - 4 * n4 - 2 * n2;
  fb_phase_g<NQT, LO, 4, PT, 2>(a, lds, red, lane, wave, 0, n4, src_off, rd_off);
  fb_phase_g<NQT, LO, 2, PT, 4>(a, lds, red, lane, wave, 4 * n4, n2, src_off, rd_off);
  fb_phase_g<NQT, LO, 1, PT, 4>(a, lds, red, lane, wave, 4 * n4 + 2 * n2, n1, src_off, rd_off);
}

// Caller-supplied document FDE vectors (mv_index_import_fde): one block per page rounds the fp32 vector to the slab's bf16 (RNE, as the
// encode kernels do) and stores 1 / |d| of the ROUNDED vector -- the same convention, so cosine scores mean the same thing for both sources.
__global__ __launch_bounds__(256) void fde_import_kernel(const float* src, int64_t out_dim, uint16_t* out, float* inv_norm) {
  __shared__ float red[4];
  const int64_t page = blockIdx.x;
  const float* x = src + page * out_dim;
  uint16_t* o = out + page * out_dim;
  float nn = 0.f;
  for (int64_t i = threadIdx.x; i < out_dim; i += 256) {
    const uint16_t h = f32_to_bf16_rne(x[i]);
    o[i] = h;
    const float vb = bf16_to_f32(h);
    nn += vb * vb;
  }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) nn += __shfl_xor(nn, sft);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nn;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (red[0] + red[1]) + (red[2] + red[3]);
    inv_norm[page] = t > 0.0f ? 1.0f / sqrtf(t) : 0.0f;
  }
}

// Masks and the cosine rule of the batched scan, in place: scores[q][page] *= inv_norm[page]; -inf for tombstones and for
// pages outside query q's doc filter (allow_stride_bits = 0: one bitmap for all queries).
__global__ __launch_bounds__(256) void fde_batch_finish_kernel(float* scores, int64_t score_stride, int64_t n, int nq, const float* inv_norm,
                                                               const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits,
                                                               int64_t allow_stride_bits) {
  const int64_t page = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (page >= n) return;
  const float inv = inv_norm ? inv_norm[page] : 1.0f;
  const int32_t o = doc_ord ? doc_ord[page] : 0;
  for (int q = 0; q < nq; ++q) {
    bool m = false;
    if (doc_ord) {
      m = o < 0;
      if (!m && allow) {
        const uint32_t* ab = allow + (size_t)q * (size_t)(allow_stride_bits >> 5);
        m = (int64_t)o >= n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u;
      }
    }
    float* sp = scores + (size_t)q * score_stride + page;
    if (m) *sp = -INFINITY;
    else if (inv_norm) *sp = *sp * inv;
  }
}

// The same pass, also accumulating the FIRST histogram of every query's selection (order-preserving key bits [31:21], the bins
// radix_hist_kernel's pass 0 would count): the finish touches every score anyway, so the 32-request selection loses one of its
// three passes over the 4 * n * nq-byte score matrix.  Block = a page range x kFinQ queries (one 2048-bin LDS histogram each,
// wave-aggregated increments); page metadata is read once per page and block.
constexpr int kFinQ = 4;
__global__ __launch_bounds__(256) void fde_batch_finish_hist_kernel(float* scores, int64_t score_stride, int64_t n, int nq, const float* inv_norm,
                                                                    const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits,
                                                                    int64_t allow_stride_bits, uint32_t* hist0, int64_t hist0_stride_words) {
  __shared__ uint32_t h[kFinQ][2048];
  const int q0 = blockIdx.y * kFinQ;
  const int nql = min(kFinQ, nq - q0);
  for (int i = threadIdx.x; i < kFinQ * 2048; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int64_t step = (int64_t)gridDim.x * 256;
  const int64_t n_round = ((n + step - 1) / step) * step;  // whole waves walk the loop together (the ballots need every lane)
  for (int64_t page = (int64_t)blockIdx.x * 256 + threadIdx.x; page < n_round; page += step) {
    const bool in = page < n;
    float inv = 1.0f;
    int32_t o = 0;
    if (in) {
      if (inv_norm) inv = inv_norm[page];
      if (doc_ord) o = doc_ord[page];
    }
    float v[kFinQ];
#pragma unroll
    for (int ql = 0; ql < kFinQ; ++ql)
      v[ql] = (in && ql < nql) ? __builtin_nontemporal_load(scores + (size_t)(q0 + ql) * score_stride + page) : -INFINITY;
#pragma unroll
    for (int ql = 0; ql < kFinQ; ++ql) {
      if (ql < nql) {  // block-uniform
        bool valid = false;
        uint32_t bin = 0;
        if (in) {
          bool m = false;
          if (doc_ord) {
            m = o < 0;
            if (!m && allow) {
              const uint32_t* ab = allow + (size_t)(q0 + ql) * (size_t)(allow_stride_bits >> 5);
              m = (int64_t)o >= n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u;
            }
          }
          float* sp = scores + (size_t)(q0 + ql) * score_stride + page;
          float s = v[ql];
          if (m) {
            s = -INFINITY;
            *sp = s;
          } else if (inv_norm) {
            s = s * inv;
            *sp = s;
          }
          const float s0 = s + 0.0f;
          if (s0 == s0 && s0 != -INFINITY) {
            valid = true;
            bin = topk_ordered_u32(s0) >> 21;
          }
        }
        topk_hist_add_wave(h[ql], bin, valid);
      }
    }
  }
  __syncthreads();
  for (int ql = 0; ql < nql; ++ql) {
    uint32_t* dst = hist0 + (size_t)(q0 + ql) * (size_t)hist0_stride_words;
    for (int i = threadIdx.x; i < 2048; i += 256)
      if (h[ql][i]) atomicAdd(&dst[i], h[ql][i]);
  }
}

struct FdeDeviceExtra {  // bucket-sorted projection tables
  int32_t* order = nullptr;
  float* sgn = nullptr;
  int32_t* bstart = nullptr;
};

}  // namespace

// Projection tables; mirrors oracle/mv_oracle.c:orc_fde_matrices (spec in that file's FDE comment).
void fde_host_tables(const mv_fde_config& c, float* G, int32_t* H, float* S) {
  const uint32_t key[2] = {(uint32_t)c.seed, (uint32_t)(c.seed >> 32)};
  for (int32_t r = 0; r < c.num_repetitions; ++r)
    for (int32_t k = 0; k < c.dimension; ++k) {
      for (int32_t j = 0; j < c.num_simhash_projections; ++j) {
        uint32_t ctr[4] = {(uint32_t)k, (uint32_t)j, (uint32_t)r, 0x47u};
        uint32_t w[8];
        philox_host(ctr, key, w);
        ctr[3] = 0x48u;
        philox_host(ctr, key, w + 4);
        int32_t sum = 0;
        for (int t = 0; t < 6; ++t) sum += (int32_t)(w[t] & 0xffff) + (int32_t)(w[t] >> 16);
        G[((size_t)r * c.dimension + k) * c.num_simhash_projections + j] = (float)(sum - 6 * 65535) * (1.0f / 65536.0f);
      }
      uint32_t ctr[4] = {(uint32_t)k, 0u, (uint32_t)r, 0x41u};
      uint32_t w[4];
      philox_host(ctr, key, w);
      H[(size_t)r * c.dimension + k] = (int32_t)(w[0] % (uint32_t)c.projection_dimension);
      S[(size_t)r * c.dimension + k] = (w[1] & 1u) ? 1.0f : -1.0f;
    }
}

int fde_tables_create(const mv_fde_config& c, FdeTables* t) {
  if (c.dimension != kDim || c.num_repetitions < 1 || c.num_simhash_projections < 1 ||
      c.num_simhash_projections > kMaxNS || c.projection_dimension < 1 || c.projection_dimension > kMaxPD) {
    set_error("unsupported FDE config: dimension must be 128, 1<=simhash<=%d, 1<=projection_dimension<=%d", kMaxNS, kMaxPD);
    return MV_ERR_INVALID;
  }
  const int R = c.num_repetitions, D = c.dimension, NS = c.num_simhash_projections, PD = c.projection_dimension;
  t->cfg = c;
  t->out_dim = (int64_t)R * (1 << NS) * PD;
  std::vector<float> G((size_t)R * D * NS), S((size_t)R * D), sg((size_t)R * D);
  std::vector<int32_t> H((size_t)R * D), ord((size_t)R * D), bs((size_t)R * (PD + 1));
  fde_host_tables(c, G.data(), H.data(), S.data());
  for (int r = 0; r < R; ++r) {
    int m = 0;
    for (int b = 0; b < PD; ++b) {
      bs[(size_t)r * (PD + 1) + b] = m;
      for (int i = 0; i < D; ++i)
        if (H[(size_t)r * D + i] == b) {
          ord[(size_t)r * D + m] = i;
          sg[(size_t)r * D + m] = S[(size_t)r * D + i];
          ++m;
        }
    }
    bs[(size_t)r * (PD + 1) + PD] = m;
  }
  // one allocation: G | H | S | order | sgn | bstart
  const size_t bytes = G.size() * 4 + H.size() * 4 + S.size() * 4 + ord.size() * 4 + sg.size() * 4 + bs.size() * 4;
  char* d = nullptr;
  MV_HIP(hipMalloc(&d, bytes));
  size_t off = 0;
  auto put = [&](const void* src, size_t n) -> void* {
    void* p = d + off;
    (void)hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    off += n;
    return p;
  };
  t->G = (float*)put(G.data(), G.size() * 4);
  t->H = (int32_t*)put(H.data(), H.size() * 4);
  t->S = (float*)put(S.data(), S.size() * 4);
  // the bucket-sorted tables live behind S; their offsets are recomputed in launch_fde_encode
  put(ord.data(), ord.size() * 4);
  put(sg.data(), sg.size() * 4);
  put(bs.data(), bs.size() * 4);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

void fde_tables_destroy(FdeTables* t) {
  if (t->scratch) (void)hipFree(t->scratch);
  t->scratch = nullptr;
  t->scratch_bytes = 0;
  if (t->G) (void)hipFree(t->G);
  t->G = nullptr;
  t->H = nullptr;
  t->S = nullptr;
}

int launch_fde_encode(const FdeTables& t, const FdeEncodeArgs& a, hipStream_t s) {
  if (a.n_pages <= 0) return MV_OK;
  const int R = t.cfg.num_repetitions, D = t.cfg.dimension, NS = t.cfg.num_simhash_projections,
            PD = t.cfg.projection_dimension;
  EncArgs k{};
  k.x_f32 = a.x_f32; k.x_bf16 = a.x_bf16; k.row_offsets = a.row_offsets; k.n_rows = a.n_rows;
  k.stride = a.stride; k.is_query = a.is_query;
  k.G = t.G;
  const char* after_S = reinterpret_cast<const char*>(t.S) + (size_t)R * D * 4;
  k.order = reinterpret_cast<const int32_t*>(after_S);
  k.sgn = reinterpret_cast<const float*>(after_S + (size_t)R * D * 4);
  k.bstart = reinterpret_cast<const int32_t*>(after_S + (size_t)R * D * 8);
  k.R = R; k.NS = NS; k.PD = PD;
  k.scale = 1.0f / sqrtf((float)PD);
  k.out_dim = t.out_dim;
  k.out_f32 = a.out_f32; k.out_bf16 = a.out_bf16; k.out_inv_norm = a.out_inv_norm;
  if (a.variant == 2 && a.n_pages <= 64 && a.is_query && a.x_f32 && !a.row_offsets && !a.out_inv_norm && PD <= 16 && NS <= kMaxNS) {
    // one block per repetition; n_pages > 1: a batch of queries of `stride` rows each, packed back to back (grid.y)
    EncMArgs mm{k, t.H, t.S, a.n_pages};
    hipLaunchKernelGGL(fde_encode_query_kernel, dim3((unsigned)R, (unsigned)a.n_pages), dim3(256), 0, s, mm);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if (a.variant == 4 && a.x_bf16 && !a.is_query && PD <= 16 && NS <= 5 && R <= 20 && R * NS > 96 && R * NS <= 112 && a.stride % 16 == 0 && (a.out_bf16 || a.out_f32)) {
    // documents from the bf16 slab, two passes (default for the corpus build): hash pass -> partition bytes in scratch -> projection pass with
    // one-hot MFMA bucket sums.  Chunks of <= 8192 pages share one scratch buffer (stride * R bytes per page); stream order keeps them apart.
    static int ncu4 = 0;
    if (ncu4 == 0) {
      int dev = 0, v = 0;
      ncu4 = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    const int tiles_per_page = a.stride / 16;
    const size_t per_page = (size_t)a.stride * R;
    const int64_t chunk = std::min<int64_t>(a.n_pages, 8192);
    if (t.scratch_bytes < (size_t)chunk * per_page) {
      if (t.scratch) (void)hipFree(t.scratch);
      t.scratch = nullptr; t.scratch_bytes = 0;
      hipError_t e = hipMalloc(&t.scratch, (size_t)chunk * per_page);
      if (e != hipSuccess) { t.scratch = nullptr; set_error("out of device memory for %zu bytes of FDE partition scratch", (size_t)chunk * per_page); return MV_ERR_NOMEM; }
      t.scratch_bytes = (size_t)chunk * per_page;
    }
    for (int64_t p0 = 0; p0 < a.n_pages; p0 += chunk) {
      const int64_t c = std::min<int64_t>(chunk, a.n_pages - p0);
      EncArgs kc = k;
      kc.x_bf16 = a.x_bf16 + (size_t)p0 * a.stride * kDim;
      kc.n_rows = a.n_rows ? a.n_rows + p0 : nullptr;
      kc.out_f32 = a.out_f32 ? a.out_f32 + (size_t)p0 * t.out_dim : nullptr;
      kc.out_bf16 = a.out_bf16 ? a.out_bf16 + (size_t)p0 * t.out_dim : nullptr;
      kc.out_inv_norm = a.out_inv_norm ? a.out_inv_norm + p0 : nullptr;
      EncMArgs mm{kc, t.H, t.S, c};
      hipLaunchKernelGGL((fde_hash_kernel<7>), dim3((unsigned)std::min<int64_t>(c, ncu4)), dim3(256), 0, s, mm, t.scratch, tiles_per_page);
      hipLaunchKernelGGL((fde_project_kernel<5>), dim3((unsigned)std::min<int64_t>(c, 2 * (int64_t)ncu4)), dim3(256), 0, s, mm, (const uint8_t*)t.scratch, tiles_per_page);
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if ((a.variant == 3 || a.variant == 4) && a.x_bf16 && !a.is_query && PD <= 16 && R * NS > 96 && R * NS <= 112) {
    // documents from the bf16 slab (default for the corpus build): AMS on the bf16 pipe, SimHash columns in registers (NT = 7 column tiles)
    const size_t ldsd = (size_t)R * 4096 + (size_t)t.out_dim * 4 + (size_t)R * (1 << NS) * 4 + 16 + (size_t)4 * 16 * kXStrideB * 2 + (size_t)4 * 16 * 112 +
                        (size_t)4 * 16 * R + 64;
    if (ldsd <= 160 * 1024) {
      static int ncu2 = 0;
      if (ncu2 == 0) {
        int dev = 0, v = 0;
        ncu2 = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
      }
      EncMArgs mm{k, t.H, t.S, a.n_pages};
      MV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fde_encode_doc_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd));
      hipLaunchKernelGGL((fde_encode_doc_kernel<7>), dim3((unsigned)std::min<int64_t>(a.n_pages, ncu2)), dim3(256), ldsd, s, mm);
      MV_HIP(hipGetLastError());
      return MV_OK;
    }
  }
  if (a.variant != 0 && PD <= 16 && R * NS <= 128) {
    // f32-MFMA form: persistent blocks, one per CU
    const int NHP = ((R * NS + 15) / 16) * 16;
    const size_t ldsm = (size_t)kDim * NHP * 4 + (size_t)t.out_dim * 4 + (size_t)R * (1 << NS) * 4 + 16 + (size_t)((R * 128 + 15) / 16) * 16 +
                        (size_t)4 * 16 * kXStride * 4 + (size_t)4 * 16 * NHP + (size_t)4 * 16 * R + 64;
    if (ldsm <= 160 * 1024) {
      static int ncu = 0;
      if (ncu == 0) {
        int dev = 0, v = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
      }
      EncMArgs mm{k, t.H, t.S, a.n_pages};
      MV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fde_encode_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
      const int64_t grid = std::min<int64_t>(a.n_pages, ncu);
      hipLaunchKernelGGL(fde_encode_mfma_kernel, dim3((unsigned)grid), dim3(256), ldsm, s, mm);
      MV_HIP(hipGetLastError());
      return MV_OK;
    }
  }
  const size_t lds = (size_t)kDim * kRowsPerPass * 4 + (size_t)t.out_dim * 4 + (size_t)R * (1 << NS) * 4 + 16;
  if (lds > 160 * 1024) {
    set_error("FDE config needs %zu B of LDS (> 160 KiB)", lds);
    return MV_ERR_INVALID;
  }
  MV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fde_encode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds));
  // few pages (the query): split the repetitions over blockIdx.y so each wave runs one repetition (latency form)
  const unsigned ysplit = (a.n_pages <= 8 && !a.out_inv_norm && R >= 8) ? (unsigned)((R + 3) / 4) : 1u;
  hipLaunchKernelGGL(fde_encode_kernel, dim3((unsigned)a.n_pages, ysplit), dim3(256), lds, s, k);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

template <int ITERS>
static int launch_fde_scan_lds(const ScanArgs& k, hipStream_t s) {
  const int ppb = 64;  // pages per block: the 40 KiB query staging is amortised over 1.3 MB of page reads
  const size_t lds = (size_t)ITERS * 512 * 4;
  MV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fde_scan_lds_kernel<ITERS>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((fde_scan_lds_kernel<ITERS>), dim3((unsigned)((k.n + ppb - 1) / ppb)), dim3(512), lds, s, k, ppb);
  return MV_OK;
}

bool fde_scan_prebins(int variant, int64_t out_dim) { return (variant <= 0 || variant == 3 || variant == 4) && (out_dim == 10240 || out_dim == 5120); }

// Shape of the LDS-DMA scan: `grid` persistent workgroups (4 waves each); a chunk is ppw (<= 64) consecutive pages for one wave.
// Dynamic claiming (work counter given): 16 pages per chunk -- 320 KiB, ~90 us of one wave's stream; the ring drains once per
// chunk and the launch ends within one chunk time of the last claim.  Static order: ppw chosen so that every wave gets the
// same number of (nearly full) chunks.
static void fde_ldsdma_shape(int64_t n, bool dynamic, int* grid, int* ppw) {
  static int env_ppw = -1, env_bpc = -1;
  if (env_ppw < 0) {  // tuning hooks (tools/fde_scan_probe.py); unset in production
    const char* e = getenv("MV_FDE_SCAN_PPW");
    env_ppw = e ? atoi(e) : 0;
    e = getenv("MV_FDE_SCAN_BLOCKS_PER_CU");
    env_bpc = e ? atoi(e) : 0;
  }
  const int g = 256 * (env_bpc > 0 ? env_bpc : 2);
  const int64_t waves = (int64_t)g * 4;
  if (env_ppw > 0) {
    *ppw = env_ppw > 64 ? 64 : env_ppw;
  } else if (dynamic) {
    *ppw = 16;
  } else {
    const int64_t rounds = (n + waves * 64 - 1) / (waves * 64);  // chunks per wave at ppw = 64
    const int64_t p = (n + rounds * waves - 1) / (rounds * waves);
    *ppw = (int)(p < 1 ? 1 : (p > 64 ? 64 : p));
  }
  const int64_t nchunks = (n + *ppw - 1) / *ppw;
  const int64_t blocks = (nchunks + 3) / 4;
  *grid = (int)(blocks < g ? blocks : g);
}

int launch_fde_scan(const FdeScanArgs& a, int variant, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  ScanArgs k{a.fde, a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n, (int32_t)a.out_dim, nullptr};
  const int grid = 256 * 2;  // 2 blocks/CU x 4 waves, persistent
  if (variant < 0) variant = 3;  // round 5: nt LDS-DMA ring (0 = the same arithmetic on plain nt loads: 6.8 TB/s)
  if (variant == 5 && (a.out_dim == 10240 || a.out_dim == 5120)) {  // one fresh workgroup per row
    constexpr int64_t kRowsPerLaunch = (int64_t)1 << 23;  // a launch's work-item count must stay below 2^32
    for (int64_t off = 0; off < a.n; off += kRowsPerLaunch) {
      ScanArgs c = k;
      c.n = std::min(kRowsPerLaunch, a.n - off);
      c.fde = k.fde + (size_t)off * (size_t)a.out_dim;
      c.scores = k.scores + off;
      if (k.inv_norm) c.inv_norm = k.inv_norm + off;
      if (k.doc_ord) c.doc_ord = k.doc_ord + off;
      if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_rows_kernel<5, 4>), dim3((unsigned)c.n), dim3(256), 0, s, c);
      else hipLaunchKernelGGL((fde_scan_rows_kernel<5, 2>), dim3((unsigned)((c.n + 1) / 2)), dim3(256), 0, s, c);
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if (variant == 5) variant = 0;
  if (variant == 9 && a.out_dim == 10240) {  // calibration: the LDS-DMA form's transport alone
    int g, ppw;
    k.work = a.work;
    fde_ldsdma_shape(a.n, k.work != nullptr, &g, &ppw);
    hipLaunchKernelGGL((fde_scan_ldsdma_kernel<20, 4, 4, true>), dim3(g), dim3(256), 0, s, k, ppw);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if ((variant == 3 || variant == 4) && (a.out_dim == 10240 || a.out_dim == 5120)) {
    int g, ppw;
    k.work = variant == 3 ? a.work : nullptr;
    fde_ldsdma_shape(a.n, k.work != nullptr, &g, &ppw);
    k.hist0 = a.hist0;
    if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_ldsdma_kernel<20, 4, 4>), dim3(g), dim3(256), 0, s, k, ppw);
    else hipLaunchKernelGGL((fde_scan_ldsdma_kernel<10, 2, 8>), dim3(g), dim3(256), 0, s, k, ppw);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if (variant == 3 || variant == 4) variant = 0;
  if (variant == 2 && a.out_dim % 2048 == 0 && a.out_dim / 2048 <= 5 && a.out_dim >= 2048) {
    const int wg = 256 * 4;  // 4 workgroups per CU, persistent
    switch ((int)(a.out_dim / 2048)) {
      case 1: hipLaunchKernelGGL((fde_scan_coop_kernel<1>), dim3(wg), dim3(256), 0, s, k); break;
      case 2: hipLaunchKernelGGL((fde_scan_coop_kernel<2>), dim3(wg), dim3(256), 0, s, k); break;
      case 3: hipLaunchKernelGGL((fde_scan_coop_kernel<3>), dim3(wg), dim3(256), 0, s, k); break;
      case 4: hipLaunchKernelGGL((fde_scan_coop_kernel<4>), dim3(wg), dim3(256), 0, s, k); break;
      default: hipLaunchKernelGGL((fde_scan_coop_kernel<5>), dim3(wg), dim3(256), 0, s, k); break;
    }
  } else if (a.out_dim == 10240 && variant == 1) {
    int rc = launch_fde_scan_lds<20>(k, s);
    if (rc) return rc;
  } else if (a.out_dim == 5120 && variant == 1) {
    int rc = launch_fde_scan_lds<10>(k, s);
    if (rc) return rc;
  } else if (a.out_dim == 10240) {
    k.hist0 = a.hist0;
    hipLaunchKernelGGL((fde_scan_kernel<20, 4>), dim3(grid), dim3(256), 0, s, k);
  } else if (a.out_dim == 5120) {
    k.hist0 = a.hist0;
    hipLaunchKernelGGL((fde_scan_kernel<10, 2>), dim3(grid), dim3(256), 0, s, k);
  } else {
    if (a.n > ((int64_t)1 << 25)) { set_error("generic FDE scan: more than 2^25 pages per launch is not supported"); return MV_ERR_INVALID; }
    hipLaunchKernelGGL(fde_scan_generic_kernel, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, k);
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_fde_import(const float* d_src, int64_t n, int64_t out_dim, uint16_t* out_bf16, float* out_inv_norm, hipStream_t s) {
  if (n <= 0) return MV_OK;
  hipLaunchKernelGGL(fde_import_kernel, dim3((unsigned)n), dim3(256), 0, s, d_src, out_dim, out_bf16, out_inv_norm);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

bool fde_scan_batch_supported(int64_t out_dim) { return out_dim >= 1024 && out_dim <= 65536 && out_dim % 1024 == 0; }
size_t fde_scan_batch_image_bytes(int64_t out_dim) { return (size_t)(out_dim / 256) * 16384 * 2; }  // two query tiles

int launch_fde_scan_batch(const FdeScanBatchArgs& a, hipStream_t s) {
  if (a.n <= 0 || a.n_queries <= 0) return MV_OK;
  if (a.n_queries > kFdeBatchMaxQueries || !fde_scan_batch_supported(a.out_dim)) { set_error("batched FDE scan: %d queries / out_dim %lld not supported", a.n_queries, (long long)a.out_dim); return MV_ERR_INVALID; }
  if (a.n > ((int64_t)1 << 36)) { set_error("batched FDE scan: too many pages"); return MV_ERR_INVALID; }
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int KC = (int)(a.out_dim / 256);
  const int nqt = a.n_queries > 16 ? 2 : 1;
  hipLaunchKernelGGL(fde_batch_qprep_kernel, dim3((unsigned)(KC * 2 * nqt)), dim3(256), 0, s, a.q, a.n_queries, (int)a.out_dim, nqt, a.image);
  const int64_t n_tiles = (a.n + kFbPages - 1) / kFbPages;
  ScanBatchArgs k{reinterpret_cast<const char*>(a.fde), reinterpret_cast<const char*>(a.image), a.scores, a.score_stride, a.n,
                  (int32_t)a.out_dim, a.n_queries, (int32_t)n_tiles, a.inv_norm, a.doc_ord};
  const bool fin = fde_scan_batch_fuses_finish(a);  // the paired-tile kernel applies the cosine rule and the tombstones itself
  const dim3 grid((unsigned)std::min<int64_t>(n_tiles, ncu));
  if (a.ring_slots) {  // 32-page tiles, one workgroup per CU, a deep DMA ring (MV_OPT_FDE_BATCH_VARIANT = 7 / 8)
    const int64_t n32 = (a.n + 31) / 32;
    k.n_tiles = (int32_t)n32;
    const dim3 grid1((unsigned)std::min<int64_t>(n32, (int64_t)ncu));
    auto go = [&](auto RC) {
      constexpr int R = decltype(RC)::value;
      if (a.hi_only) {
        if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch5_kernel<1, false, R>), grid1, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((fde_scan_batch5_kernel<2, false, R>), grid1, dim3(256), 0, s, k);
      } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch5_kernel<1, true, R>), grid1, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch5_kernel<2, true, R>), grid1, dim3(256), 0, s, k);
    };
    if (a.ring_slots >= 9) go(std::integral_constant<int, 9>{});
    else go(std::integral_constant<int, 4>{});
  } else if (a.half_tiles) {  // 32-page tiles, two workgroups per CU (MV_OPT_FDE_BATCH_VARIANT = 4)
    const int64_t n32 = (a.n + 31) / 32;
    k.n_tiles = (int32_t)n32;
    const dim3 grid2((unsigned)std::min<int64_t>(n32, (int64_t)ncu * 2));
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch3_kernel<1, false>), grid2, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch3_kernel<2, false>), grid2, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch3_kernel<1, true>), grid2, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch3_kernel<2, true>), grid2, dim3(256), 0, s, k);
  } else if (a.single_tile) {  // one page tile per query fragment (MV_OPT_FDE_BATCH_VARIANT = 3: the cross-check of the paired form)
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch_kernel<1, false>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch_kernel<2, false>), grid, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch_kernel<1>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch_kernel<2>), grid, dim3(256), 0, s, k);
  } else if (fin) {
    if (a.private_rings) {
      if (a.hi_only) {
        if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, true, true>), grid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, true, true>), grid, dim3(256), 0, s, k);
      } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, true, true>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, true, true>), grid, dim3(256), 0, s, k);
    } else if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, true>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, true>), grid, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, true>), grid, dim3(256), 0, s, k);
    // per-request doc filters are a dependent lookup (ordinal -> bitmap word): they stay a pass of their own, masks only
    if (a.allow && a.doc_ord)
      hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                         (const float*)nullptr, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
    MV_HIP(hipGetLastError());
    return MV_OK;
  } else if (a.private_rings) {
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, false, true>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, false, true>), grid, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, false, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, false, true>), grid, dim3(256), 0, s, k);
  } else if (a.hi_only) {
    if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false>), grid, dim3(256), 0, s, k);
  } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true>), grid, dim3(256), 0, s, k);
  if (fde_scan_batch_prebins(a)) {
    const int gx = (int)std::min<int64_t>((a.n + 255) / 256, 128);
    hipLaunchKernelGGL(fde_batch_finish_hist_kernel, dim3((unsigned)gx, (unsigned)((a.n_queries + kFinQ - 1) / kFinQ)), dim3(256), 0, s, a.scores,
                       a.score_stride, a.n, a.n_queries, a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits, a.hist0,
                       a.hist0_stride_bytes / 4);
  } else if (a.inv_norm || a.doc_ord)
    hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                       a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
