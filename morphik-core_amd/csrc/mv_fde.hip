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
  const int64_t* x_row_off;  // packed layout (with x_bf16): page i's rows start x_row_off[i] rows into x_bf16; null: i * stride
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
    r0 = a.x_row_off ? a.x_row_off[page] : page * (int64_t)a.stride;
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
      r0 = a.x_row_off ? a.x_row_off[page] : page * (int64_t)a.stride;
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

// (Round 3's one-pass document encoder -- AMS on the bf16 pipe with LDS atomics for the bucket sums -- was replaced by the two-pass
// form below in round 4 (1.9x) and removed in round 5; record: profiles/r3/pmc_fde_encode_kernels_r3d.json.)
constexpr int kXStrideB = 132;  // bf16 elements per staged row of the document kernels (264 B)
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
    const uint16_t* pg = a.x_bf16 + (a.x_row_off ? (size_t)a.x_row_off[page] : (size_t)page * (size_t)a.stride) * kDim;
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
    const uint16_t* pg = a.x_bf16 + (a.x_row_off ? (size_t)a.x_row_off[page] : (size_t)page * (size_t)a.stride) * kDim;
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
    float total;
    if constexpr (WPR == 4) total = (part[0] + part[1]) + (part[2] + part[3]);
    else total = part[0] + part[1];
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

// ---- round 5: the coarse scan on the float scan's transport (nt LDS-DMA into wave-private rings, counted vmcnt, no barrier in
// the stream).  Forms built, measured and removed this round (records: profiles/r5/):
//   * wave-owned streams through the ring -- one wave per row, chunks of consecutive rows per persistent wave, static order or
//     claimed from a device counter: 6.65-6.8 TB/s, no better than the register kernel's plain nt loads (6.7-6.8)
//   * one fresh workgroup per ROW with the row split over its waves: 5.3 TB/s (the query slice reloaded from L2 per 20 KiB)
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

// ---- round 5, the form that won: ROW QUARTERS, one fresh workgroup per unit of consecutive rows.
// The transport probe (csrc/mv_synth.hip: stream_probe_kernel; profiles/r5/stream_structure_*.json) measured what shape of work the
// nt LDS-DMA ring streams fastest over a 25.6 GB FDE slab.  Streams owned by single waves -- the register kernel, the ring kernel
// above, persistent or claimed -- stay 2-4 % under shapes in which a WORKGROUP reads contiguous memory together (the float scan's
// 16 KiB bursts, or whole 20 KiB rows), and fresh workgroups handed out by the dispatcher match or beat persistent ones.  One
// fresh workgroup per ROW would be the fastest stream, but then every workgroup reloads its slice of the query FDE from L2
// (2 bytes per byte of HBM: 3.1 TB/s measured).  So: a workgroup takes `ru` (16) consecutive rows = 320 KiB; wave w streams the
// w-th quarter (CPW KiB) of every row through a private ring of D row slots and keeps only ITS slice of the query FDE (8 CPW
// floats per lane, loaded once per workgroup: 12.5 % extra L2 reads); lane i of every wave keeps the wave's partial sum of row i,
// and ONE barrier per workgroup joins the four partials: score = ((p0 + p1) + (p2 + p3)) / |d|.
// Filter and 1/norm are evaluated for the whole unit before the stream starts (nothing but the DMAs touches vector memory inside it).
// Arithmetic order == fde_scan_kernel's: bit-identical scores.
template <int CPW, int WPR, int D>  // out_dim = 512 CPW WPR; CPW <= 5; WPR in {2, 4}; 4 / WPR units per workgroup
__global__ __launch_bounds__(256) void fde_scan_rowq_kernel(ScanArgs a, int ru) {
  static_assert(CPW >= 1 && CPW <= 5 && (WPR == 2 || WPR == 4) && D >= 2 && D <= 4 && CPW * (D - 1) <= 63, "row shape");
  constexpr int G = 4 / WPR;
  constexpr int SLOT = CPW * 1024;
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[4 * D * SLOT + 4 * 64 * 4];
  float* part_sum = reinterpret_cast<float*>(lds + 4 * D * SLOT);  // [wave][row of the unit]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int group = wave / WPR, part = wave % WPR;
  char* ring = lds + wave * (D * SLOT);
  const int voff = lane * 16;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

  // ---- unit prologue: lane i <-> the unit's i-th row
  // ru > 0: units of ru consecutive rows; ru == 0: one 256 KiB-aligned block of the slab per workgroup (block_unit_rows)
  int64_t base;
  if (ru > 0) base = ((int64_t)blockIdx.x * G + group) * (int64_t)ru;
  else block_unit_rows(a.fde, (int64_t)a.out_dim * 2, (int64_t)blockIdx.x, G, group, &base, &ru);
  const int64_t myrow = base + lane;
  const bool valid = lane < ru && myrow < a.n;
  bool masked = false;
  float my_inv = 1.0f;
  if (valid) {
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[myrow];
      masked = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
    }
    if (a.inv_norm && !masked && part == 0) my_inv = a.inv_norm[myrow];
  }
  const uint64_t live = __ballot(valid && !masked);
  uint64_t iss = live, cons = live;
  int to_issue = __builtin_popcountll(live);
  int to_read = to_issue;
  int iss_slot = 0, cons_slot = 0;
  const char* qbase = reinterpret_cast<const char*>(a.fde) + (size_t)part * SLOT;
  const size_t row_bytes = (size_t)a.out_dim * 2;

  auto issue_next = [&]() {
    const int i = __builtin_ctzll(iss);
    iss &= iss - 1;
    const char* tp = qbase + (size_t)(base + i) * row_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t m0a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + iss_slot * SLOT));
    uint32_t keep;
    // the instruction offset (12 bits) walks BOTH addresses; the fifth chunk takes a second M0 and a +4 KiB lane offset
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
    iss_slot = (iss_slot + 1 == D) ? 0 : iss_slot + 1;
    --to_issue;
  };

#pragma unroll
  for (int k = 0; k < D - 1; ++k)
    if (to_issue > 0) issue_next();

  // this wave's slice of the query FDE, behind the first rows' DMAs; waited for HERE, once (left alone hipcc sinks the wait
  // into the stream, where its counted vmcnt would drain the ring on every row)
  float q[CPW][8];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float* qp = a.q + (size_t)(part * CPW + c) * 512 + lane * 8;
    const float4 lo4 = *reinterpret_cast<const float4*>(qp);
    const float4 hi4 = *reinterpret_cast<const float4*>(qp + 4);
    q[c][0] = lo4.x; q[c][1] = lo4.y; q[c][2] = lo4.z; q[c][3] = lo4.w;
    q[c][4] = hi4.x; q[c][5] = hi4.y; q[c][6] = hi4.z; q[c][7] = hi4.w;
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(q[c][k]));
  asm volatile("" : "+v"(my_inv));

  float my_part = 0.0f;
  while (cons) {
    const int i = __builtin_ctzll(cons);
    cons &= cons - 1;
    if (to_issue > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: the last reads of the slot being refilled
      issue_next();
      fde_wait_vmcnt<CPW * (D - 1)>();
    } else {
      fde_wait_left<CPW, D>(to_read - 1);
    }
    --to_read;
    const char* slot = ring + cons_slot * SLOT + voff;
    cons_slot = (cons_slot + 1 == D) ? 0 : cons_slot + 1;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(slot + c * 1024);
      const uint32_t w[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc = __builtin_fmaf(__uint_as_float(w[k] << 16), q[c][2 * k], acc);
        acc = __builtin_fmaf(__uint_as_float(w[k] & 0xffff0000u), q[c][2 * k + 1], acc);
      }
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
    if (lane == i) my_part = acc;
  }

  // ---- unit epilogue: join the parts, one store per row
  part_sum[wave * 64 + lane] = my_part;
  __syncthreads();
  if (part == 0 && valid) {
    const float* p = part_sum + wave * 64 + lane;  // waves [wave, wave + WPR) are this group's parts
    float t;
    if constexpr (WPR == 4) t = (p[0] + p[64]) + (p[128] + p[192]);
    else t = p[0] + p[64];
    a.scores[myrow] = masked ? -INFINITY : (a.inv_norm ? t * my_inv : t);
  }
}

// (Rounds 1-2 also had the query FDE in LDS -- 16 waves per CU -- and a workgroup-cooperative register form: 6.1 TB/s both.)
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
  k.x_row_off = a.x_bf16 ? a.x_row_off : nullptr;
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
      kc.x_bf16 = a.x_row_off ? a.x_bf16 : a.x_bf16 + (size_t)p0 * a.stride * kDim;  // packed: the offsets are absolute rows of the slab
      kc.x_row_off = a.x_row_off ? a.x_row_off + p0 : nullptr;
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

// Only the register form accumulates the selection's first histogram itself (persistent workgroups: one LDS histogram each, flushed
// once).  The row-quarter form's workgroups live for 16 rows: binning from them means global atomics on the dozen bins FDE scores share
// -- measured (round 5, one atomic per distinct bin per workgroup): the 1.25 M-row scan went from 3.69 to 4.50 ms.  Behind the default
// form the radix selection takes its own first pass over the n scores (4 bytes per page against 20 480).
bool fde_scan_prebins(int variant, int64_t out_dim) { return variant == 0 && (out_dim == 10240 || out_dim == 5120); }

// variant: -1 / 6 = row quarters on the nt LDS-DMA ring, one fresh workgroup per 256 KiB-aligned block of the slab (default at 10 240 /
//          5 120 dims; round 6: +1.2 .. 1.6 % over 5);  5 = the same with a workgroup per 16 consecutive rows (320 KiB);
//          0 = one wave per row on plain nt loads, the query FDE in registers (the same arithmetic order: bit-identical scores);
//          other widths: the generic kernel
int launch_fde_scan(const FdeScanArgs& a, int variant, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  ScanArgs k{a.fde, a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n, (int32_t)a.out_dim, nullptr};
  const bool shaped = a.out_dim == 10240 || a.out_dim == 5120;
  if (variant < 0) variant = 6;
  const bool want_blocks = variant == 6;
  if (variant == 6) variant = 5;
  if (variant != 0 && variant != 5) { set_error("unknown FDE scan variant %d (5 = row quarters / LDS-DMA, 0 = register form)", variant); return MV_ERR_INVALID; }
  if (variant == 5 && shaped) {
    static int env_ru = -1;
    if (env_ru == -1) { const char* e = getenv("MV_FDE_SCAN_RU"); env_ru = e ? atoi(e) : 0; }  // tuning hook (tools/scan_ceiling_probe.py)
    const bool blocks = want_blocks && env_ru <= 0;
    const int ru = blocks ? 0 : (env_ru > 0 ? (env_ru > 64 ? 64 : env_ru) : 16);  // 8 / 16 / 32 rows per workgroup measured equal within the noise
    const int64_t units = blocks ? block_unit_count(a.fde, a.out_dim * 2, a.n) * (a.out_dim == 10240 ? 1 : 2) : (a.n + ru - 1) / ru;  // (5 120 dims: two groups per workgroup)
    if (units > ((int64_t)1 << 23)) { set_error("FDE scan: more than 2^27 rows per launch is not supported"); return MV_ERR_INVALID; }
    if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_rowq_kernel<5, 4, 3>), dim3((unsigned)units), dim3(256), 0, s, k, ru);
    else hipLaunchKernelGGL((fde_scan_rowq_kernel<5, 2, 3>), dim3((unsigned)((units + 1) / 2)), dim3(256), 0, s, k, ru);
  } else if (shaped) {
    const int grid = 256 * 2;  // 2 blocks/CU x 4 waves, persistent
    k.hist0 = a.hist0;
    if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_kernel<20, 4>), dim3(grid), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_kernel<10, 2>), dim3(grid), dim3(256), 0, s, k);
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

}  // namespace mv
