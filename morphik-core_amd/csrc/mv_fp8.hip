// mv_fp8.hip -- float MaxSim over an FP8 (OCP e4m3fn) page slab on the CDNA4 block-scaled matrix cores.
//
// BASELINE.json configs[4]: "fp8 (e4m3) patch embeddings on CDNA4 fp8 MFMA ... recall@10 vs bf16 reference".
// Same scoring rule as mv_maxsim.hip (score_multi_vector, core/vector_store/fast_multivector_store.py:553-555)
// on a slab that costs 128 B per patch row instead of 256 B: the scan is HBM-bound, so halving the bytes
// doubles pages/s.  The reference has no fp8 path; the checker is oracle/mv_oracle.c:orc_maxsim_fp8 (same
// quantised operands, fp32/fp64 arithmetic) and, for quality, recall@10 against the bf16 scores.
//
// Quantisation (bit-exact with the oracle: pure integer / exact power-of-two arithmetic):
//   page   : one power-of-two scale 2^e per page, e = floor(log2(448 / amax(page)));  code = e4m3_rne(x * 2^e)
//   query  : per row a power-of-two scale 2^s (amax -> [224, 448]) and a TWO-TERM split
//              hi = e4m3_rne(x 2^s),  lo = e4m3_rne((x 2^s - hi) * 16)          (|residual| <= ulp/2 <= 16)
//            so the query side carries ~8 significant bits (bf16 class); the MFMA pair
//              acc  = Ahi . B            (block scale 2^0)
//              acc += Alo . B * 2^-4     (E8M0 block scale 123 on the A operand)
//            costs two of the ~18% utilised MFMA slots per tile.
//   score  = 2^-e_page * sum_rows 2^-s_row * max_patch acc[row][patch]          (positive scales commute with max)
//
// MFMA: v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 0: e4m3 x e4m3), K = 128 in one instruction:
// lane (r = l&15, g = l>>4) holds row r's bytes [16g, 16g+16) and [64+16g, 64+16g+16) -- any K-slot bijection is
// valid as long as A and B agree, and this one makes the LDS reads conflict-free (below).
//
// Data movement (as mv_maxsim.hip variant 3): four waves per page, each wave owns every 4th 4 KiB piece (32 rows)
// and moves it with four global_load_lds_dwordx4 (8 whole rows = 1 KiB contiguous per instruction) into a
// wave-private 4-slot LDS ring (non-temporal loads: +8% here, +12% on the bf16 scan); counted vmcnt, no barrier in
// the loop.  LDS image: 16-byte chunk c of row w at chunk position c ^ (w & 7) of that row (swizzle applied to the
// per-lane global source address); the ds_read_b128 fragment reads (chunk g+4h of rows r) then hit 16 distinct
// 16-byte columns per 16-lane group.
// The full scan (no candidate list) runs maxsim_fp8_pair_kernel: the same stream over TWO consecutive pages per fresh
// workgroup, 256 KiB of contiguous slab -- measured +2.3 / +2.8 % over one page per workgroup in alternating processes
// (profiles/r5/fp8_scan_page_pairs_ab_r5.jsonl); candidate lists and per-item queries keep maxsim_fp8_kernel.
#include <algorithm>
#include <cstdlib>

#include "mv_common.h"
#include "mv_e4m3.h"

namespace mv {
namespace {

using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kF8RowBytes = kDim;                    // 128
constexpr int kF8SlotRows = 32;                      // rows per ring slot (2 MFMA tiles)
constexpr int kF8SlotBytes = kF8SlotRows * kF8RowBytes;  // 4 KiB

// ---------------------------------------------------------------- e4m3fn codec (mirrors oracle/mv_oracle.c)
__host__ __device__ __forceinline__ uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
#endif
}
__host__ __device__ __forceinline__ float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}


// ---------------------------------------------------------------- page quantisation (ingest side)
// One block per page.  src = fixed-stride bf16 slab pages; rows >= n_rows become zero codes.
// row_off (packed layout): the page's rows start row_off[page] rows into both slabs and its slot ends at row_off[page + 1].
__global__ __launch_bounds__(256) void quantize_pages_kernel(const uint16_t* src, const int32_t* n_rows, int32_t stride,
                                                             uint8_t* dst, float* inv_scale, const int64_t* row_off) {
  __shared__ uint32_t red[4];
  const int64_t page = blockIdx.x;
  const int nr = n_rows ? n_rows[page] : stride;
  const size_t row0 = row_off ? (size_t)row_off[page] : (size_t)page * stride;
  const int slot_rows = row_off ? (int)(row_off[page + 1] - row_off[page]) : stride;
  const uint16_t* sp = src + row0 * kDim;
  uint8_t* dp = dst + row0 * kDim;
  const int n8 = nr * (kDim / 8);  // 16-byte chunks of valid data
  uint32_t amax = 0;
  for (int i = threadIdx.x; i < n8; i += 256) {
    const uint4 v = reinterpret_cast<const uint4*>(sp)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = w[k] & 0x7fffu, hi = (w[k] >> 16) & 0x7fffu;
      // finite magnitudes order like integers; NaN/inf rows are not expected in embeddings (clamped below)
      amax = max(amax, max(lo, hi));
    }
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, s));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = max(max(red[0], red[1]), max(red[2], red[3]));
  if (amax > 0x7f7fu) amax = 0x7f7fu;  // inf/NaN guard: largest finite bf16
  const int e = pow2_scale_exp(amax << 16);
  const float sc = pow2f(e);
  if (threadIdx.x == 0) inv_scale[page] = pow2f(-e);
  const int tot8 = slot_rows * (kDim / 8);
  for (int i = threadIdx.x; i < tot8; i += 256) {
    uint2 out = make_uint2(0u, 0u);
    if (i < n8) {
      const uint4 v = reinterpret_cast<const uint4*>(sp)[i];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      uint32_t c[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c[2 * k] = e4m3_encode(__uint_as_float(w[k] << 16) * sc);
        c[2 * k + 1] = e4m3_encode(__uint_as_float(w[k] & 0xffff0000u) * sc);
      }
      out.x = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
      out.y = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
    }
    reinterpret_cast<uint2*>(dp)[i] = out;
  }
}

// ---------------------------------------------------------------- query preparation
// q: [n_q][128] fp32.  Writes hi/lo e4m3 rows (zero for the padding rows up to `padded`) and 2^-s per row (0 = padding).
__global__ __launch_bounds__(64) void fp8_query_prep_kernel(const float* q, int n_q, int padded, uint8_t* hi, uint8_t* lo,
                                                            float* row_factor) {
  const int row = blockIdx.x;
  const int lane = threadIdx.x;
  if (row >= padded) return;
  if (row >= n_q) {
    reinterpret_cast<uint16_t*>(hi + (size_t)row * kDim)[lane] = 0;
    reinterpret_cast<uint16_t*>(lo + (size_t)row * kDim)[lane] = 0;
    if (lane == 0) row_factor[row] = 0.0f;
    return;
  }
  const float x0 = q[(size_t)row * kDim + 2 * lane], x1 = q[(size_t)row * kDim + 2 * lane + 1];
  uint32_t amax = max(__float_as_uint(x0) & 0x7fffffffu, __float_as_uint(x1) & 0x7fffffffu);
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, s));
  if (amax > 0x7f7fffffu) amax = 0x7f7fffffu;
  const int e = pow2_scale_exp(amax);
  const float sc = pow2f(e);
  uint32_t ch[2], cl[2];
  const float xs[2] = {x0 * sc, x1 * sc};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ch[k] = e4m3_encode(xs[k]);
    const float res = xs[k] - e4m3_decode(ch[k]);  // exact: both are multiples of the hi ulp's sub-grid within fp32
    cl[k] = e4m3_encode(res * 16.0f);
  }
  reinterpret_cast<uint16_t*>(hi + (size_t)row * kDim)[lane] = (uint16_t)(ch[0] | (ch[1] << 8));
  reinterpret_cast<uint16_t*>(lo + (size_t)row * kDim)[lane] = (uint16_t)(cl[0] | (cl[1] << 8));
  if (lane == 0) row_factor[row] = pow2f(-e);
}

// ---------------------------------------------------------------- the scan
struct F8Args {
  const uint8_t* slab;      // [pages][stride][128] e4m3
  const float* inv_scale;   // [pages]
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const int32_t* cand;
  const uint8_t* qhi;       // [padded][128]
  const uint8_t* qlo;
  const float* qfac;        // [padded] 2^-s_row, 0 for padding rows
  float* scores;
  int64_t n;
  int32_t stride;
  int32_t pad_to;
  int64_t page0;
  int32_t accumulate;
  const int32_t* pad_items;  // per-item pad_to; null -> pad_to
  int32_t items_per_q;       // QITEM kernels: work item i scores against query i / items_per_q, whose rows start
  int32_t q_item_rows;       // (i / items_per_q) * q_item_rows rows into qhi / qlo / qfac (the rerank lists of a batch of queries)
  const int64_t* row_off;    // packed layout: first slab row of every page; null: page * stride
};

template <bool PK>  // PK: packed layout -- a template parameter so the fixed layout's kernels are untouched
__device__ __forceinline__ const char* f8_page_base(const F8Args& a, int64_t page) {
  return reinterpret_cast<const char*>(a.slab) + (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride) * kF8RowBytes;
}

__device__ __forceinline__ bool f8_masked(const F8Args& a, int64_t page) {
  if (!a.doc_ord) return false;
  const int32_t o = a.doc_ord[page];
  if (o < 0) return true;
  if (!a.allow) return false;
  if ((int64_t)o >= a.n_allow_bits) return true;
  return ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u;
}

template <int CTRL>
__device__ __forceinline__ float f8_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float f8_group16_max(float v) {
  v = fmaxf(v, f8_dpp<0x128>(v));
  v = fmaxf(v, f8_dpp<0x124>(v));
  v = fmaxf(v, f8_dpp<0x122>(v));
  v = fmaxf(v, f8_dpp<0x121>(v));
  return v;
}

template <int N>
__device__ __forceinline__ void f8_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MT, int D, bool QITEM = false, bool PK = false>
__global__ __launch_bounds__(256) void maxsim_fp8_kernel(F8Args a) {
  __shared__ __attribute__((aligned(16))) char lds[4 * D * kF8SlotBytes + 1024];
  float* red = reinterpret_cast<float*>(lds + 4 * D * kF8SlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t item = blockIdx.x;
  if (item >= a.n) return;
  const int64_t page = a.cand ? (int64_t)a.cand[item] : a.page0 + item;
  if (page < 0 || f8_masked(a, page)) {  // page < 0: padding entry of a device-built candidate list
    if (threadIdx.x == 0) a.scores[item] = -INFINITY;
    return;
  }
  int64_t pk_row0 = 0;  // packed layout: loaded beside the row count (one s_waitcnt for both)
  if constexpr (PK) pk_row0 = a.row_off[page];
  const int nr = PK ? a.n_rows[page] : (a.n_rows ? a.n_rows[page] : a.stride);
  const int ntiles = (nr + 15) >> 4;
  const int nslots = (nr + kF8SlotRows - 1) / kF8SlotRows;
  const bool clamp = (a.pad_items ? a.pad_items[item] : a.pad_to) > nr;
  const int nsw = (nslots - wave + 3) / 4;  // slots owned by this wave (may be <= 0)
  const char* pbase = PK ? reinterpret_cast<const char*>(a.slab) + (size_t)pk_row0 * kF8RowBytes : f8_page_base<false>(a, page);
  char* ring = lds + wave * (D * kF8SlotBytes);

  // DMA source offsets: instruction i covers rows 8i..8i+7 of the slot; lane -> LDS row 8i + (lane>>3), chunk
  // position lane&7, which must receive logical chunk (lane&7) ^ (row&7).
  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 8 + (lane >> 3);
    src_off[i] = w * kF8RowBytes + ((((lane & 7) ^ (w & 7))) << 4) - i * 1024;
  }
  // fragment read offsets within a 16-row tile: logical chunk g + 4h of row r
  int rd_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) rd_off[h] = r * kF8RowBytes + (((g + 4 * h) ^ (r & 7)) << 4);

  auto issue = [&](int it) {
    const char* tp = pbase + (size_t)(wave + it * 4) * kF8SlotBytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (it % D) * kF8SlotBytes));
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
        : "v"(src_off[0]), "v"(src_off[1]), "v"(src_off[2]), "v"(src_off[3]), "s"(slot), "s"(tpu)
        : "memory");
  };

#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (i < nsw) issue(i);

  // query operands (loop invariant), loaded after the prologue DMAs and pinned (see mv_maxsim.hip)
  const uint8_t* qhi = a.qhi;
  const uint8_t* qlo = a.qlo;
  const float* qfac = a.qfac;
  if (QITEM) {  // this item's query (batched rerank)
    const size_t qrow = (size_t)(item / a.items_per_q) * (size_t)a.q_item_rows;
    qhi += qrow * kF8RowBytes;
    qlo += qrow * kF8RowBytes;
    qfac += qrow;
  }
  i32x8 ah[MT], al[MT];
  float fac[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const size_t ro = (size_t)(m * 16 + r) * kF8RowBytes;
    const i32x4 h0 = *reinterpret_cast<const i32x4*>(qhi + ro + g * 16);
    const i32x4 h1 = *reinterpret_cast<const i32x4*>(qhi + ro + 64 + g * 16);
    const i32x4 l0 = *reinterpret_cast<const i32x4*>(qlo + ro + g * 16);
    const i32x4 l1 = *reinterpret_cast<const i32x4*>(qlo + ro + 64 + g * 16);
    const float4 f = *reinterpret_cast<const float4*>(qfac + m * 16 + g * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[m][i] = h0[i]; ah[m][4 + i] = h1[i];
      al[m][i] = l0[i]; al[m][4 + i] = l1[i];
    }
    fac[m][0] = f.x; fac[m][1] = f.y; fac[m][2] = f.z; fac[m][3] = f.w;
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("" : "+v"(ah[m][i]));
      asm volatile("" : "+v"(al[m][i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fac[m][i]));
  }

  f32x4 mx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mx[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

  for (int it = 0; it < nsw; ++it) {
    if (it + D - 1 < nsw) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: last reads of the slot being refilled
      issue(it + D - 1);
      f8_wait_vmcnt<4 * (D - 1)>();
    } else {
      const int left = nsw - 1 - it;
      if (left >= 2) f8_wait_vmcnt<8>();
      else if (left == 1) f8_wait_vmcnt<4>();
      else f8_wait_vmcnt<0>();
    }
    const char* slot = ring + (it % D) * kF8SlotBytes;
    const int t0 = (wave + it * 4) * 2;  // first 16-row tile of this slot
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = t0 + tt;
      if (t < ntiles) {  // wave-uniform
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(slot + tt * 2048 + rd_off[0]);
        const i32x4 b1 = *reinterpret_cast<const i32x4*>(slot + tt * 2048 + rd_off[1]);
        i32x8 b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { b[i] = b0[i]; b[4 + i] = b1[i]; }
        const bool col_valid = t * 16 + r < nr;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah[m], b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al[m], b, acc, 0, 0, 0, 0x7b7b7b7b /* 2^-4 */, 0, 0x7f7f7f7f);
          if (!col_valid) acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
          for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[i]);
        }
      }
    }
  }

  // cross-wave max, then per-row factors, sum, page scale
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = f8_group16_max(mx[m][i]);
      if (r == 0) red[wave * 64 + m * 16 + g * 4 + i] = v;
    }
  __syncthreads();
  if (wave == 0) {
    float v = 0.f;
    if (lane < MT * 16) {
      v = fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
      if (clamp) v = fmaxf(v, 0.f);
      if (v == -INFINITY) v = 0.f;
      v *= qfac[lane];
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    if (lane == 0) {
      const float part = v * a.inv_scale[page];
      a.scores[item] = a.accumulate ? a.scores[item] + part : part;
    }
  }
}

// Two CONSECUTIVE pages per workgroup (round 5, the full scan only: no candidate list, no per-item queries): wave w
// streams its slots of page 2b and then, without draining the ring, its slots of page 2b+1 -- 256 KiB of contiguous
// slab per fresh workgroup, the unit the float scan reads (128 KiB units deliver ~1.3 % less than 256 KiB
// ones), and one query-fragment load per two pages.  Per page the arithmetic is maxsim_fp8_kernel's: the running maxima
// are handed to LDS at the page boundary, waves 0 / 1 finish pages 0 / 1 after the single barrier.
template <int MT, int D, bool PK = false>
__global__ __launch_bounds__(256) void maxsim_fp8_pair_kernel(F8Args a) {
  __shared__ __attribute__((aligned(16))) char lds[4 * D * kF8SlotBytes + 2048];
  float* red = reinterpret_cast<float*>(lds + 4 * D * kF8SlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t item0 = (int64_t)blockIdx.x * 2;
  if (item0 >= a.n) return;
  int nr[2], nsw[2];
  bool live[2];
  const char* pbase[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int64_t page = a.page0 + item0 + p;
    live[p] = item0 + p < a.n && !f8_masked(a, page);
    nr[p] = live[p] ? (a.n_rows ? a.n_rows[page] : a.stride) : 0;
    const int nslots = (nr[p] + kF8SlotRows - 1) / kF8SlotRows;
    nsw[p] = max(0, (nslots - wave + 3) / 4);
    pbase[p] = (!PK || item0 + p < a.n) ? f8_page_base<PK>(a, page) : reinterpret_cast<const char*>(a.slab);  // (packed: no table entry to read behind the last page)
  }
  const int total = nsw[0] + nsw[1];
  char* ring = lds + wave * (D * kF8SlotBytes);

  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 8 + (lane >> 3);
    src_off[i] = w * kF8RowBytes + ((((lane & 7) ^ (w & 7))) << 4) - i * 1024;
  }
  int rd_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) rd_off[h] = r * kF8RowBytes + (((g + 4 * h) ^ (r & 7)) << 4);

  auto issue = [&](int j) {  // j: position in this wave's stream over both pages
    const char* tp = j < nsw[0] ? pbase[0] + (size_t)(wave + j * 4) * kF8SlotBytes
                                : pbase[1] + (size_t)(wave + (j - nsw[0]) * 4) * kF8SlotBytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (j % D) * kF8SlotBytes));
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
        : "v"(src_off[0]), "v"(src_off[1]), "v"(src_off[2]), "v"(src_off[3]), "s"(slot), "s"(tpu)
        : "memory");
  };

#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (i < total) issue(i);

  i32x8 ah[MT], al[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const size_t ro = (size_t)(m * 16 + r) * kF8RowBytes;
    const i32x4 h0 = *reinterpret_cast<const i32x4*>(a.qhi + ro + g * 16);
    const i32x4 h1 = *reinterpret_cast<const i32x4*>(a.qhi + ro + 64 + g * 16);
    const i32x4 l0 = *reinterpret_cast<const i32x4*>(a.qlo + ro + g * 16);
    const i32x4 l1 = *reinterpret_cast<const i32x4*>(a.qlo + ro + 64 + g * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[m][i] = h0[i]; ah[m][4 + i] = h1[i];
      al[m][i] = l0[i]; al[m][4 + i] = l1[i];
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("" : "+v"(ah[m][i]));
      asm volatile("" : "+v"(al[m][i]));
    }

  int j = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    f32x4 mx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mx[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const int ntiles = (nr[p] + 15) >> 4;
    for (int it = 0; it < nsw[p]; ++it, ++j) {
      if (j + D - 1 < total) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: last reads of the slot being refilled
        issue(j + D - 1);
        f8_wait_vmcnt<4 * (D - 1)>();
      } else {
        const int left = total - 1 - j;
        if (left >= 2) f8_wait_vmcnt<8>();
        else if (left == 1) f8_wait_vmcnt<4>();
        else f8_wait_vmcnt<0>();
      }
      const char* slot = ring + (j % D) * kF8SlotBytes;
      const int t0 = (wave + it * 4) * 2;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = t0 + tt;
        if (t < ntiles) {  // wave-uniform
          const i32x4 b0 = *reinterpret_cast<const i32x4*>(slot + tt * 2048 + rd_off[0]);
          const i32x4 b1 = *reinterpret_cast<const i32x4*>(slot + tt * 2048 + rd_off[1]);
          i32x8 b;
#pragma unroll
          for (int i = 0; i < 4; ++i) { b[i] = b0[i]; b[4 + i] = b1[i]; }
          const bool col_valid = t * 16 + r < nr[p];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ah[m], b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(al[m], b, acc, 0, 0, 0, 0x7b7b7b7b /* 2^-4 */, 0, 0x7f7f7f7f);
            if (!col_valid) acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[i]);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = f8_group16_max(mx[m][i]);
        if (r == 0) red[p * 256 + wave * 64 + m * 16 + g * 4 + i] = v;
      }
  }
  __syncthreads();
  if (wave < 2 && item0 + wave < a.n) {
    const int p = wave;
    const int64_t item = item0 + p;
    const float* rp = red + p * 256;
    const int nr_p = p ? nr[1] : nr[0];
    const bool live_p = p ? live[1] : live[0];
    float v = 0.f;
    if (lane < MT * 16) {
      v = fmaxf(fmaxf(rp[lane], rp[64 + lane]), fmaxf(rp[128 + lane], rp[192 + lane]));
      if ((a.pad_items ? a.pad_items[item] : a.pad_to) > nr_p) v = fmaxf(v, 0.f);
      if (v == -INFINITY) v = 0.f;
      v *= a.qfac[lane];
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    if (lane == 0) {
      if (!live_p) {
        a.scores[item] = -INFINITY;
      } else {
        const float part = v * a.inv_scale[a.page0 + item];
        a.scores[item] = a.accumulate ? a.scores[item] + part : part;
      }
    }
  }
}

// MV_FP8_SCAN_PAIRS=0: one page per workgroup for the full scan too (the measurement switch of the pair kernel).
bool f8_scan_pairs() {
  static const bool on = [] {
    const char* e = getenv("MV_FP8_SCAN_PAIRS");
    return !(e && e[0] == '0');
  }();
  return on;
}

template <int MT>
int launch_f8_mt(const F8Args& k0, hipStream_t s) {
  constexpr int64_t kChunk = (int64_t)1 << 22;  // work-item count per launch stays below 2^32
  for (int64_t off = 0; off < k0.n; off += kChunk) {
    F8Args k = k0;
    k.n = std::min(kChunk, k0.n - off);
    k.scores = k0.scores + off;
    if (k0.cand) k.cand = k0.cand + off;
    else k.page0 = k0.page0 + off;
    if (k0.pad_items) k.pad_items = k0.pad_items + off;
    if (k.items_per_q > 0) {
      if (k0.n > kChunk) { set_error("per-item queries: %lld items in one launch not supported", (long long)k0.n); return MV_ERR_INVALID; }
      if (k.row_off) hipLaunchKernelGGL((maxsim_fp8_kernel<MT, 4, true, true>), dim3((unsigned)k.n), dim3(256), 0, s, k);
      else hipLaunchKernelGGL((maxsim_fp8_kernel<MT, 4, true>), dim3((unsigned)k.n), dim3(256), 0, s, k);
    } else if (!k.cand && f8_scan_pairs()) {
      if (k.row_off) hipLaunchKernelGGL((maxsim_fp8_pair_kernel<MT, 4, true>), dim3((unsigned)((k.n + 1) / 2)), dim3(256), 0, s, k);
      else hipLaunchKernelGGL((maxsim_fp8_pair_kernel<MT, 4>), dim3((unsigned)((k.n + 1) / 2)), dim3(256), 0, s, k);
    } else {
      if (k.row_off) hipLaunchKernelGGL((maxsim_fp8_kernel<MT, 4, false, true>), dim3((unsigned)k.n), dim3(256), 0, s, k);
      else hipLaunchKernelGGL((maxsim_fp8_kernel<MT, 4>), dim3((unsigned)k.n), dim3(256), 0, s, k);
    }
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}


// ------------------------------------------------------------------------------------------------------------
// A BATCH of queries against the e4m3 slab in one pass (round 3): the row-split workgroup of mv_batch.hip on the
// block-scaled matrix path.  The M = B x Q query rows are split over the four waves (MTW 16-row tiles per wave, their
// two-term e4m3 fragments resident in VGPRs), page tiles of 16 patches x 128 B are staged ONCE per workgroup in a ring of
// 16 KiB chunks (128 patch rows; wave w DMAs rows [32w, 32w+32) of every chunk with the swizzled image of the single-query
// scan) and read by all four waves.  MFMA roles transposed as in maxsim_batch_kernel: A = page tile, B = query tile,
// D[patch 4(l>>4)+i][token l&15] -- one running maximum per query tile, fed by v_max3_f32.  K = 128 is ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 per (page tile, query tile) and term: with LO the query keeps its hi + lo*2^-4 split
// (the scores of the single-query scan); without it (MV_OPT_BATCH_VARIANT 7) one term only -- half the matrix work, the
// query rounded to e4m3 like the pages: the coarse pass of a two-tier search.
// Half the page bytes of the bf16 batch and, per term, half its matrix cycles: the HBM-bound small batches gain ~2x.
struct F8BatchArgs {
  const uint8_t* slab;
  const float* inv_scale;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  int64_t allow_stride_bits;  // 0: one bitmap for all queries
  const uint8_t* qhi;         // [4 * MTW * 16][128] e4m3, zero rows behind the last query
  const uint8_t* qlo;
  const float* qfac;          // [4 * MTW * 16]
  float* scores;              // [n_queries][score_stride]
  int64_t n;
  int64_t score_stride;
  int32_t stride;
  int32_t n_queries;
  int32_t rows_per_query;     // multiple of 16
  const int64_t* row_off;     // packed layout
};

constexpr int kF8ChunkRows = 128;
constexpr int kF8ChunkBytes = kF8ChunkRows * kF8RowBytes;  // 16 KiB
constexpr int kF8TileBytes = 16 * kF8RowBytes;             // 2 KiB

template <int CTRL>
__device__ __forceinline__ float f8b_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int MTW, int S, bool LO, bool PK = false>
__global__ __launch_bounds__(256, 2) void maxsim_batch_fp8_kernel(F8BatchArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[S * kF8ChunkBytes + 64 * MTW * 4 + 4 * MTW * 4];
  float* red = reinterpret_cast<float*>(lds + S * kF8ChunkBytes);  // [64 * MTW] factor-scaled token maxima of the current page
  float* part = red + 64 * MTW;                                    // [4 * MTW] sums of 16 rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;

  i32x8 qh[MTW], ql[LO ? MTW : 1];
  float fq[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const size_t row = (size_t)(wave * MTW + m) * 16 + r;
    const i32x4 h0 = *reinterpret_cast<const i32x4*>(a.qhi + row * kF8RowBytes + g * 16);
    const i32x4 h1 = *reinterpret_cast<const i32x4*>(a.qhi + row * kF8RowBytes + 64 + g * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { qh[m][i] = h0[i]; qh[m][4 + i] = h1[i]; }
    if (LO) {
      const i32x4 l0 = *reinterpret_cast<const i32x4*>(a.qlo + row * kF8RowBytes + g * 16);
      const i32x4 l1 = *reinterpret_cast<const i32x4*>(a.qlo + row * kF8RowBytes + 64 + g * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { ql[m][i] = l0[i]; ql[m][4 + i] = l1[i]; }
    }
    fq[m] = a.qfac[row];
  }
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("" : "+v"(qh[m][i]));
      if (LO) asm volatile("" : "+v"(ql[m][i]));
    }
    asm volatile("" : "+v"(fq[m]));
  }

  int src_off[4];  // DMA instruction i covers rows 8i..8i+7 of the wave's 32-row piece (as maxsim_fp8_kernel)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 8 + (lane >> 3);
    src_off[i] = w * kF8RowBytes + ((((lane & 7) ^ (w & 7))) << 4) - i * 1024;
  }
  int rd_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) rd_off[h] = r * kF8RowBytes + (((g + 4 * h) ^ (r & 7)) << 4);

  for (int64_t item = blockIdx.x; item < a.n; item += gridDim.x) {
    const int64_t page = item;
    bool masked = false;
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[page];
      masked = o < 0;
      if (!masked && a.allow && !a.allow_stride_bits) masked = (int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u;
    }
    if (masked) {  // block-uniform
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
      continue;
    }
    const int nr = a.n_rows ? a.n_rows[page] : a.stride;
    if (nr <= 0) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
      continue;
    }
    const int ntiles = (nr + 15) >> 4;
    const int nchunks = (nr + kF8ChunkRows - 1) / kF8ChunkRows;
    const char* pbase = reinterpret_cast<const char*>(a.slab) + (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride) * kF8RowBytes;

    auto issue = [&](int c) {  // wave w moves rows [32w, 32w+32) of chunk c (always issued: the slab is padded)
      const char* tp = pbase + (size_t)(c * 4 + wave) * 4096;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (c % S) * kF8ChunkBytes + wave * 4096));
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
          : "v"(src_off[0]), "v"(src_off[1]), "v"(src_off[2]), "v"(src_off[3]), "s"(slot), "s"(tpu)
          : "memory");
    };
    auto frags = [&](i32x8& b, const char* tile) {
      const i32x4 b0 = *reinterpret_cast<const i32x4*>(tile + rd_off[0]);
      const i32x4 b1 = *reinterpret_cast<const i32x4*>(tile + rd_off[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { b[i] = b0[i]; b[4 + i] = b1[i]; }
    };

#pragma unroll
    for (int c = 0; c < S - 1; ++c)
      if (c < nchunks) issue(c);

    float mx[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) mx[m] = -INFINITY;

    for (int c = 0; c < nchunks; ++c) {
      const int ahead = min(S - 2, nchunks - 1 - c);
      if (ahead >= 2) f8_wait_vmcnt<8>();
      else if (ahead == 1) f8_wait_vmcnt<4>();
      else f8_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk c visible to all four waves; everyone is done reading chunk c-1
      if (c + S - 1 < nchunks) issue(c + S - 1);
      const char* chunk = lds + (c % S) * kF8ChunkBytes;
      i32x8 b[2];
      frags(b[0], chunk);
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int t = c * 8 + tt;
        if (t < ntiles) {  // block-uniform
          if (tt + 1 < 8) frags(b[(tt + 1) & 1], chunk + (tt + 1) * kF8TileBytes);  // next tile's fragments behind this tile's MFMAs
          f32x4 acc[MTW];
#pragma unroll
          for (int m = 0; m < MTW; ++m)
            acc[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[tt & 1], qh[m], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          if (LO) {
#pragma unroll
            for (int m = 0; m < MTW; ++m)
              acc[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[tt & 1], ql[m], acc[m], 0, 0, 0, 0x7f7f7f7f, 0, 0x7b7b7b7b /* 2^-4 */);
          }
          // One row tile per wave: nothing but a uniform branch (full tiles skip the mask block) separates the MFMA from the maxima
          // that read its result, and hipcc's hazard pass leaves that branch-target path short of the wait states an 8-pass MFMA
          // result needs before a VALU read (ISA of this instantiation: s_cbranch straight onto v_max3_f32) -- two queries over
          // 1.25 M pages came back as garbage.  Pinned here; with two or more tiles the other tiles' MFMAs cover the distance.
          if constexpr (MTW == 1) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(acc[0]));
          if ((t + 1) * 16 > nr) {  // partial last tile: mask the patches (rows of D) past n_rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool row_valid = t * 16 + g * 4 + i < nr;
#pragma unroll
              for (int m = 0; m < MTW; ++m)
                if (!row_valid) acc[m][i] = -INFINITY;
            }
          }
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            mx[m] = fmaxf(fmaxf(mx[m], acc[m][0]), acc[m][1]);  // v_max3_f32
            mx[m] = fmaxf(fmaxf(mx[m], acc[m][2]), acc[m][3]);
          }
        }
      }
    }

    // token maxima (the four lane groups hold disjoint patches of the same token), scaled by the token's 2^-s
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      float v = mx[m];
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      if (g == 0) red[(wave * MTW + m) * 16 + r] = (v == -INFINITY ? 0.f : v) * fq[m];
    }
    __syncthreads();
    const int rows_total = a.n_queries * a.rows_per_query;
    for (int t = threadIdx.x; t < rows_total; t += 256) {  // whole 16-row groups (rows_total and 256 are multiples of 16)
      float x = red[t];
      x = f8b_dpp_add<0x128>(x);
      x = f8b_dpp_add<0x124>(x);
      x = f8b_dpp_add<0x122>(x);
      x = f8b_dpp_add<0x121>(x);
      if ((t & 15) == 0) part[t >> 4] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.n_queries) {
      const int g16 = a.rows_per_query >> 4;
      const float* pp = part + (size_t)threadIdx.x * g16;
      float sum = 0.f;
      for (int i = 0; i < g16; ++i) sum += pp[i];
      sum *= a.inv_scale[page];
      if (a.allow && a.allow_stride_bits) {  // this query's own doc_ids filter
        const int32_t o = a.doc_ord[page];
        const uint32_t* ab = a.allow + (size_t)threadIdx.x * (size_t)(a.allow_stride_bits >> 5);
        if ((int64_t)o >= a.n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u) sum = -INFINITY;
      }
      a.scores[(size_t)threadIdx.x * a.score_stride + item] = sum;
    }
    // the next page's first barrier orders these reads of red[] / part[] before their rewrite
  }
}

template <int MTW>
int launch_f8_batch_mtw(const F8BatchArgs& k, int grid, bool lo, hipStream_t s) {
  if (k.row_off) {
    if (lo) hipLaunchKernelGGL((maxsim_batch_fp8_kernel<MTW, 4, true, true>), dim3((unsigned)grid), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((maxsim_batch_fp8_kernel<MTW, 4, false, true>), dim3((unsigned)grid), dim3(256), 0, s, k);
  } else if (lo) hipLaunchKernelGGL((maxsim_batch_fp8_kernel<MTW, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, k);
  else hipLaunchKernelGGL((maxsim_batch_fp8_kernel<MTW, 4, false>), dim3((unsigned)grid), dim3(256), 0, s, k);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace

int launch_quantize_pages_fp8(const uint16_t* d_src_pages, const int32_t* d_n_rows, int32_t stride, int64_t n_pages,
                              uint8_t* d_dst, float* d_inv_scale, hipStream_t s, const int64_t* d_row_off) {
  int64_t done = 0;
  while (done < n_pages) {
    const int64_t c = std::min<int64_t>(n_pages - done, (int64_t)1 << 22);
    if (d_row_off)  // packed: the offsets are absolute rows of both slabs
      hipLaunchKernelGGL(quantize_pages_kernel, dim3((unsigned)c), dim3(256), 0, s, d_src_pages, d_n_rows ? d_n_rows + done : nullptr, stride, d_dst,
                         d_inv_scale + done, d_row_off + done);
    else
      hipLaunchKernelGGL(quantize_pages_kernel, dim3((unsigned)c), dim3(256), 0, s, d_src_pages + (size_t)done * stride * kDim,
                         d_n_rows ? d_n_rows + done : nullptr, stride, d_dst + (size_t)done * stride * kDim, d_inv_scale + done, (const int64_t*)nullptr);
    done += c;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_fp8_query_prep(const float* d_q_f32, int n_q, uint8_t* d_hi, uint8_t* d_lo, float* d_fac, hipStream_t s) {
  const int padded = ((n_q + 15) / 16) * 16;
  hipLaunchKernelGGL(fp8_query_prep_kernel, dim3((unsigned)padded), dim3(64), 0, s, d_q_f32, n_q, padded, d_hi, d_lo, d_fac);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_maxsim_fp8(const Fp8ScanArgs& a, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  if (a.row_off && !a.n_rows) { set_error("fp8 scan: a row-offset table needs the per-page row counts"); return MV_ERR_INVALID; }
  const int padded = ((a.n_q + 15) / 16) * 16;
  if (a.items_per_query > 0 && (padded > 64 || !a.cand)) { set_error("per-item queries need a candidate list and <= 64 query rows"); return MV_ERR_INVALID; }
  for (int q0 = 0, pass = 0; q0 < padded; q0 += 64, ++pass) {
    const int mt = std::min(4, (padded - q0) / 16);
    F8Args k{a.slab, a.inv_scale, a.n_rows, a.doc_ord, a.allow, a.n_allow_bits, a.cand,
             a.qhi + (size_t)q0 * kDim, a.qlo + (size_t)q0 * kDim, a.qfac + q0, a.scores, a.n, a.stride, a.pad_to, 0, pass > 0, a.pad_items, a.items_per_query, padded, a.row_off};
    int rc;
    switch (mt) {
      case 1: rc = launch_f8_mt<1>(k, s); break;
      case 2: rc = launch_f8_mt<2>(k, s); break;
      case 3: rc = launch_f8_mt<3>(k, s); break;
      default: rc = launch_f8_mt<4>(k, s); break;
    }
    if (rc) return rc;
  }
  return MV_OK;
}

int launch_maxsim_batch_fp8(const Fp8BatchArgs& a, hipStream_t s) {
  if (a.n <= 0 || a.n_queries <= 0) return MV_OK;
  if (a.row_off && !a.n_rows) { set_error("fp8 batch scan: a row-offset table needs the per-page row counts"); return MV_ERR_INVALID; }
  if (a.rows_per_query < 16 || a.rows_per_query % 16) { set_error("fp8 batch scan: rows_per_query must be a positive multiple of 16"); return MV_ERR_INVALID; }
  const int rows = a.n_queries * a.rows_per_query;
  if (rows > 512 || a.n_queries > 256) { set_error("fp8 batch scan: %d query rows exceed the 512-row group", rows); return MV_ERR_INVALID; }
  F8BatchArgs k{a.slab, a.inv_scale, a.n_rows, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits, a.qhi, a.qlo, a.qfac, a.scores, a.n,
                a.score_stride, a.stride, a.n_queries, a.rows_per_query, a.row_off};
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    else ncu = 256;
  }
  const int grid = (int)std::min<int64_t>(a.n, (int64_t)ncu * 2);
  const bool lo = a.single_term == 0;
  switch ((rows + 63) / 64) {  // query tiles per wave
    case 1: return launch_f8_batch_mtw<1>(k, grid, lo, s);
    case 2: return launch_f8_batch_mtw<2>(k, grid, lo, s);
    case 3: return launch_f8_batch_mtw<3>(k, grid, lo, s);
    case 4: return launch_f8_batch_mtw<4>(k, grid, lo, s);
    case 5: return launch_f8_batch_mtw<5>(k, grid, lo, s);
    case 6: return launch_f8_batch_mtw<6>(k, grid, lo, s);
    case 7: return launch_f8_batch_mtw<7>(k, grid, lo, s);
    default: return launch_f8_batch_mtw<8>(k, grid, lo, s);
  }
}

}  // namespace mv
