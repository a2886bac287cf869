// mv_batch.hip -- float MaxSim of a BATCH of queries in one pass over the bf16 page slab.
//
// Same scoring rule as mv_maxsim.hip (score_multi_vector, core/vector_store/fast_multivector_store.py:553-555; its
// first argument is already a LIST of queries: einsum("bnd,csd->bcns")), for B queries at once.  One query is
// HBM-bound (32 flop/byte); every extra query re-uses the page bytes already on the CU, so B queries cost one slab
// read and the scan moves towards the bf16 MFMA roof: M = B x Q query rows against each 16-patch tile.  At
// M = 512 (16 queries of 32 tokens) a page is 8192 MFMAs (134 MFLOP) per 262 144 B = 512 flop/byte, past the
// ~315 flop/byte ridge.
//
// Work split (the opposite of the single-query kernel): the M rows are split over the four waves of a workgroup
// (MTW 16-row tiles per wave, A fragments resident in VGPRs for the whole launch), and every wave multiplies EVERY
// patch tile -- so page tiles are staged ONCE per workgroup in LDS and read by all four waves:
//   * ring of S chunks x 16 KiB (4 tiles); wave w DMAs tile w of each chunk (4 x global_load_lds_dwordx4, the
//     XOR-swizzled image of mv_maxsim.hip), non-temporal;
//   * per chunk: counted s_waitcnt vmcnt for the own DMA, ONE s_barrier (chunk landed for everyone + previous chunk
//     released), issue of chunk c+S-1, then 4 tiles x (4 ds_read_b128 + MTW x 4 MFMA + MTW x 4 v_max);
//   * the MFMAs of a tile are issued K-quarter-major over MTW independent accumulators (no dependent back-to-back
//     MFMA on one accumulator).
// Workgroups are persistent (grid = 2 per CU): the query fragments are loaded once, pages are taken round-robin.
// Per page the row maxima go through LDS and thread b sums the rows of query b: scores[b][page].
#include <algorithm>
#include <type_traits>

#include "mv_common.h"

namespace mv {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kChunkTiles = 4;
constexpr int kChunkBytes = kChunkTiles * kTileBytes;  // 16 KiB

struct BKArgs {
  const char* slab;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const uint16_t* q;  // [4 * MTW * 16][128] bf16, zero padded
  float* scores;      // [n_queries][score_stride]
  int64_t n;
  int64_t score_stride;
  int64_t page0;
  int32_t stride;
  int32_t n_queries;
  int32_t rows_per_query;  // multiple of 16
  int64_t allow_stride_bits;  // 0: one bitmap shared by all queries; > 0: query b filters with allow + b * stride/32
};

__device__ __forceinline__ bool bk_masked(const BKArgs& a, int64_t page) {
  if (!a.doc_ord) return false;
  const int32_t o = a.doc_ord[page];
  if (o < 0) return true;
  if (!a.allow || a.allow_stride_bits) return false;  // per-query bitmaps are applied when the scores are written
  if ((int64_t)o >= a.n_allow_bits) return true;
  return ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u;
}

template <int CTRL>
__device__ __forceinline__ float bk_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float bk_group16_max(float v) {
  v = fmaxf(v, bk_dpp<0x128>(v));
  v = fmaxf(v, bk_dpp<0x124>(v));
  v = fmaxf(v, bk_dpp<0x122>(v));
  v = fmaxf(v, bk_dpp<0x121>(v));
  return v;
}

template <int N>
__device__ __forceinline__ void bk_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bk_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Shared epilogue: red[] holds one maximum per query row of the current page (n_rows_total rows, a multiple of 16).
// Thread t sums row t's 16-row group with DPP rotations, the group leaders leave part[t / 16] in LDS and thread b adds the
// rows_per_query / 16 partials of query b (was: one thread walking rows_per_query LDS words serially, ~2 us per page).
// `part` must not alias red[].  Ends with the scores written; the caller orders the next rewrite of red[] / part[].
__device__ __forceinline__ float bk_sum16(float v) {
  v += bk_dpp<0x128>(v);
  v += bk_dpp<0x124>(v);
  v += bk_dpp<0x122>(v);
  v += bk_dpp<0x121>(v);
  return v;
}

__device__ __forceinline__ void bk_write_scores(const BKArgs& a, const float* red, float* part, int n_rows_total, int64_t page, int64_t item) {
  for (int t = threadIdx.x; t < n_rows_total; t += blockDim.x) {  // n_rows_total and blockDim are multiples of 16: whole groups
    float x = red[t];
    if (x == -INFINITY) x = 0.f;  // a page without valid rows for this token
    x = bk_sum16(x);
    if ((t & 15) == 0) part[t >> 4] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < a.n_queries) {
    const int g16 = a.rows_per_query >> 4;
    const float* pp = part + (size_t)threadIdx.x * g16;
    float sum = 0.f;
    for (int i = 0; i < g16; ++i) sum += pp[i];
    if (a.allow && a.allow_stride_bits) {  // this query's own doc_ids filter
      const int32_t o = a.doc_ord[page];
      const uint32_t* ab = a.allow + (size_t)threadIdx.x * (size_t)(a.allow_stride_bits >> 5);
      if ((int64_t)o >= a.n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u) sum = -INFINITY;
    }
    a.scores[(size_t)threadIdx.x * a.score_stride + item] = sum;
  }
}

// Row-split workgroup, TRANSPOSED MFMA roles (round 2): the page tile is the A operand and the query tile the B operand,
// D[row = patch 4(l>>4)+i][col = query token l&15].  A lane's four results are four patches of ONE query token, so the
// running maximum over patches is ONE register per query tile, fed by v_max3_f32 two results at a time: 2 VALU ops per
// 4 MFMAs instead of 4 (the fragment registers are the same either way: lane l holds row l&15, k-slice 8(l>>4)).  The 24
// registers this frees at 8 query tiles per wave pay for a second set of page fragments: tile t+1 is read from LDS while
// tile t multiplies.
template <int MTW, int S>
__global__ __launch_bounds__(256, 2) void maxsim_batch_kernel(BKArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[S * kChunkBytes + 64 * MTW * 4 + 4 * MTW * 4];
  float* red = reinterpret_cast<float*>(lds + S * kChunkBytes);  // [64 * MTW] row maxima of the current page
  float* part = red + 64 * MTW;                                  // [4 * MTW] sums of 16 rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;

  // query fragments of this wave's MTW row tiles (resident for the whole launch)
  bf16x8 qa[MTW][4];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      qa[m][j] = *reinterpret_cast<const bf16x8*>(a.q + ((size_t)(wave * MTW + m) * 16 + r) * kDim + j * 32 + g * 8);
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qa[m][j]));

  // DMA source offsets (as mv_maxsim.hip): instruction i covers rows 4i..4i+3 of the tile
  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  int rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = r * kRowBytes + (((j * 4 + g) ^ r) << 4);

  for (int64_t item = blockIdx.x; item < a.n; item += gridDim.x) {
    const int64_t page = a.page0 + item;
    if (bk_masked(a, page)) {  // block-uniform
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
      continue;
    }
    const int nr = a.n_rows ? a.n_rows[page] : a.stride;
    if (nr <= 0) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
      continue;
    }
    const int ntiles = (nr + kTileRows - 1) / kTileRows;
    const int nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    const char* pbase = a.slab + (size_t)page * (size_t)a.stride * kRowBytes;

    // wave w moves tile w of chunk c (always issued: the slab is padded, rows past n_rows are never consumed)
    auto issue = [&](int c) {
      const char* tp = pbase + (size_t)(c * kChunkTiles + wave) * kTileBytes;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (c % S) * kChunkBytes + wave * kTileBytes));
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
    auto frags = [&](bf16x8 (&b)[4], const char* tile) {
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tile + rd_off[j]);
    };

#pragma unroll
    for (int c = 0; c < S - 1; ++c)
      if (c < nchunks) issue(c);

    float mx[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) mx[m] = -INFINITY;

    for (int c = 0; c < nchunks; ++c) {
      // own DMA of chunk c landed: chunks c+1 .. min(c+S-2, nchunks-1) may stay in flight
      const int ahead = min(S - 2, nchunks - 1 - c);
      if (ahead >= 2) bk_wait_vmcnt<8>();
      else if (ahead == 1) bk_wait_vmcnt<4>();
      else bk_wait_vmcnt<0>();
      bk_barrier();  // chunk c visible to all four waves; everyone is done reading chunk c-1
      if (c + S - 1 < nchunks) issue(c + S - 1);
      const char* chunk = lds + (c % S) * kChunkBytes;
      bf16x8 b[2][4];
      frags(b[0], chunk);
#pragma unroll
      for (int tt = 0; tt < kChunkTiles; ++tt) {
        const int t = c * kChunkTiles + tt;
        if (t < ntiles) {  // block-uniform
          if (tt + 1 < kChunkTiles) frags(b[(tt + 1) & 1], chunk + (tt + 1) * kTileBytes);  // next tile's fragments behind this tile's MFMAs
          f32x4 acc[MTW];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
              const f32x4 cin = j == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[m];
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tt & 1][j], qa[m][j], cin, 0, 0, 0);
            }
          // one row tile per wave: pin the MFMA -> VALU wait states (see maxsim_batch_fp8_kernel; hipcc leaves the branch-target path short)
          if constexpr (MTW == 1) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(acc[0]));
          if ((t + 1) * kTileRows > nr) {  // partial last tile: mask the patches (rows of D) past n_rows
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool row_valid = t * kTileRows + g * 4 + i < nr;
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

    // token maxima: the four lane groups hold disjoint patches of the same token -> two cross-group steps, then LDS
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      float v = mx[m];
      v = fmaxf(v, __shfl_xor(v, 16));
      v = fmaxf(v, __shfl_xor(v, 32));
      if (g == 0) red[(wave * MTW + m) * 16 + r] = v;
    }
    __syncthreads();
    bk_write_scores(a, red, part, a.n_queries * a.rows_per_query, page, item);
    // the next page's first bk_barrier() orders these reads of red[] / part[] before their rewrite
  }
}

// ------------------------------------------------------------------------------------------------------------
// Page-split form for SMALL batches (<= 128 query rows in all, e.g. 4 queries of 32 tokens): HBM-bound territory, so
// the single-query scan's transport is kept as it is -- four waves per page, every wave DMAs its own interleaved 4 KiB
// tiles into a wave-private 4-slot ring, no barrier in the loop (mv_maxsim.hip) -- and every wave holds ALL query
// rows (MT <= 8 tiles).  MFMA roles transposed as above: one running maximum per query tile, v_max3_f32.
template <int MT, int D>
__global__ __launch_bounds__(256, 2) void maxsim_batch_ps_kernel(BKArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[4 * D * kTileBytes + 4 * 128 * 4 + 8 * 4];
  float* red = reinterpret_cast<float*>(lds + 4 * D * kTileBytes);  // [4 waves][128 rows]; row maxima land in red[0..128)
  float* part = red + 4 * 128;                                      // [8] sums of 16 rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t item = blockIdx.x;
  if (item >= a.n) return;
  const int64_t page = a.page0 + item;
  if (bk_masked(a, page)) {
    if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
    return;
  }
  const int nr = a.n_rows ? a.n_rows[page] : a.stride;
  if (nr <= 0) {
    if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
    return;
  }
  const int ntiles = (nr + kTileRows - 1) / kTileRows;
  const int ntw = (ntiles - wave + 3) / 4;  // tiles wave, wave + 4, ... (may be <= 0)
  const char* pbase = a.slab + (size_t)page * (size_t)a.stride * kRowBytes;
  char* ring = lds + wave * (D * kTileBytes);

  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  int rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = r * kRowBytes + (((j * 4 + g) ^ r) << 4);

  auto issue = [&](int it) {
    const char* tp = pbase + (size_t)(wave + it * 4) * kTileBytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (it % D) * kTileBytes));
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
    if (i < ntw) issue(i);

  // query fragments: loaded after the prologue DMAs were issued and pinned (see mv_maxsim.hip)
  bf16x8 qa[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) qa[m][j] = *reinterpret_cast<const bf16x8*>(a.q + ((size_t)m * 16 + r) * kDim + j * 32 + g * 8);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qa[m][j]));

  float mx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mx[m] = -INFINITY;

  for (int it = 0; it < ntw; ++it) {
    if (it + D - 1 < ntw) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: last reads of the slot being refilled
      issue(it + D - 1);
      bk_wait_vmcnt<4 * (D - 1)>();
    } else {
      const int left = ntw - 1 - it;
      if (left >= 2) bk_wait_vmcnt<8>();
      else if (left == 1) bk_wait_vmcnt<4>();
      else bk_wait_vmcnt<0>();
    }
    const char* slot = ring + (it % D) * kTileBytes;
    bf16x8 b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(slot + rd_off[j]);
    const int t = wave + it * 4;
    f32x4 acc[MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 cin = j == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[m];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], qa[m][j], cin, 0, 0, 0);
      }
    if ((t + 1) * kTileRows > nr) {  // partial last tile: mask the patches (rows of D) past n_rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool row_valid = t * kTileRows + g * 4 + i < nr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (!row_valid) acc[m][i] = -INFINITY;
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      mx[m] = fmaxf(fmaxf(mx[m], acc[m][0]), acc[m][1]);
      mx[m] = fmaxf(fmaxf(mx[m], acc[m][2]), acc[m][3]);
    }
  }

  // token maxima of this wave's tiles -> LDS; cross-wave maximum; per-query sums
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float v = mx[m];
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    if (g == 0) red[wave * 128 + m * 16 + r] = v;
  }
  __syncthreads();
  const int rows_total = a.n_queries * a.rows_per_query;  // <= MT * 16 <= 128
  float x = -INFINITY;
  if ((int)threadIdx.x < rows_total) x = fmaxf(fmaxf(red[threadIdx.x], red[128 + threadIdx.x]), fmaxf(red[256 + threadIdx.x], red[384 + threadIdx.x]));
  __syncthreads();
  if ((int)threadIdx.x < rows_total) red[threadIdx.x] = x;
  __syncthreads();
  bk_write_scores(a, red, part, rows_total, page, item);
}

// ------------------------------------------------------------------------------------------------------------
// Two-stage software pipeline of the 16x16x32 form (variant 2): MTW <= 6 row tiles per wave (384 rows per workgroup)
// leave registers for a STATIC ping-pong of both the page fragments and the accumulators:
//     tile t   : MFMAs into acc[t & 1]   (K-quarter-major, MTW independent chains)
//     meanwhile: ds_read of tile t+1's fragments into b[(t + 1) & 1]      (LDS latency off the MFMA path)
//                v_max of tile t-1 out of acc[(t - 1) & 1], a quarter after each K-quarter of MFMAs
// The barrier that publishes chunk c+1 sits in front of the LAST tile of chunk c (its fragments are already in
// registers), which also releases chunk c's slot for chunk c+S: all S slots are in flight.
template <int MTW, int S>
__global__ __launch_bounds__(256, 2) void maxsim_batch_pipe_kernel(BKArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[S * kChunkBytes + 64 * MTW * 4];
  float* red = reinterpret_cast<float*>(lds + S * kChunkBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;

  bf16x8 qa[MTW][4];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      qa[m][j] = *reinterpret_cast<const bf16x8*>(a.q + ((size_t)(wave * MTW + m) * 16 + r) * kDim + j * 32 + g * 8);
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qa[m][j]));

  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  int rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = r * kRowBytes + (((j * 4 + g) ^ r) << 4);

  for (int64_t item = blockIdx.x; item < a.n; item += gridDim.x) {
    const int64_t page = a.page0 + item;
    if (bk_masked(a, page)) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
      continue;
    }
    const int nr = a.n_rows ? a.n_rows[page] : a.stride;
    if (nr <= 0) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
      continue;
    }
    const int ntiles = (nr + kTileRows - 1) / kTileRows;
    const int nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    const int nfull = nr / (kChunkTiles * kTileRows);  // chunks whose 64 rows are all valid
    const char* pbase = a.slab + (size_t)page * (size_t)a.stride * kRowBytes;

    auto issue = [&](int c) {
      const char* tp = pbase + (size_t)(c * kChunkTiles + wave) * kTileBytes;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (c % S) * kChunkBytes + wave * kTileBytes));
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
    auto frags = [&](bf16x8 (&b)[4], int c, int tt) {
      const char* tp = lds + (c % S) * kChunkBytes + tt * kTileBytes;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(tp + rd_off[j]);
    };

#pragma unroll
    for (int c = 0; c < S; ++c)
      if (c < nchunks) issue(c);

    f32x4 mx[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) mx[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

    {
      const int ahead = min(S - 1, nchunks - 1);
      if (ahead >= 3) bk_wait_vmcnt<12>();
      else if (ahead == 2) bk_wait_vmcnt<8>();
      else if (ahead == 1) bk_wait_vmcnt<4>();
      else bk_wait_vmcnt<0>();
      bk_barrier();
    }
    bf16x8 b[2][4];
    f32x4 acc[2][MTW];
    frags(b[0], 0, 0);
#pragma unroll
    for (int m = 0; m < MTW; ++m) acc[1][m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // "tile -1": max-neutral

    for (int c = 0; c < nfull; ++c) {
#pragma unroll
      for (int tt = 0; tt < kChunkTiles; ++tt) {
        constexpr int kQuarter = (MTW + 3) / 4;  // v_max groups of the previous tile per K-quarter
        const int cur = tt & 1, prv = cur ^ 1;
        if (tt < kChunkTiles - 1) {
          frags(b[prv], c, tt + 1);
        } else if (c + 1 < nchunks) {
          const int ahead = min(S - 2, nchunks - 2 - c);
          if (ahead >= 2) bk_wait_vmcnt<8>();
          else if (ahead == 1) bk_wait_vmcnt<4>();
          else bk_wait_vmcnt<0>();
          bk_barrier();
          if (c + S < nchunks) issue(c + S);
          frags(b[prv], c + 1, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            const f32x4 cin = j == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[cur][m];
            acc[cur][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[m][j], b[cur][j], cin, 0, 0, 0);
          }
          // a quarter of the previous tile's running-max work rides behind this K-quarter's MFMAs
#pragma unroll
          for (int m = j * kQuarter; m < (j + 1) * kQuarter && m < MTW; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[prv][m][i]);
        }
        // MTW = 6 only (measured): pin the issue order -- fragment reads first, then one VALU behind every MFMA --
        // which also keeps the register allocation under 256 (it spills without); smaller MTW schedule better freely
        if (MTW == 6) {
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int i = 0; i < MTW * 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
        }
      }
    }
    // drain: the last full tile's accumulators (tile index 4*nfull - 1 used buffer 1; with nfull == 0 it is the neutral init)
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[1][m][i]);
    if (nfull < nchunks) {  // ragged tail chunk (its tile 0 is already in b[0]); every DMA has landed
#pragma unroll
      for (int tt = 0; tt < kChunkTiles; ++tt) {
        const int t = nfull * kChunkTiles + tt;
        if (t < ntiles) {  // block-uniform
          if (tt > 0) frags(b[0], nfull, tt);
          const bool col_valid = t * kTileRows + r < nr;
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[m][j], b[0][j], c4, 0, 0, 0);
            if (!col_valid) c4 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], c4[i]);
          }
        }
      }
    }

#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = bk_group16_max(mx[m][i]);
        if (r == 0) red[(wave * MTW + m) * 16 + g * 4 + i] = v;
      }
    __syncthreads();
    if ((int)threadIdx.x < a.n_queries) {
      const float* rp = red + (size_t)threadIdx.x * a.rows_per_query;
      float sum = 0.f;
      for (int i = 0; i < a.rows_per_query; ++i) sum += rp[i];
      if (a.allow && a.allow_stride_bits) {
        const int32_t o = a.doc_ord[page];
        const uint32_t* ab = a.allow + (size_t)threadIdx.x * (size_t)(a.allow_stride_bits >> 5);
        if ((int64_t)o >= a.n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u) sum = -INFINITY;
      }
      a.scores[(size_t)threadIdx.x * a.score_stride + item] = sum;
    }
    __syncthreads();  // red[] and the ring are rewritten by the next page's prologue / first tiles
  }
}

// ------------------------------------------------------------------------------------------------------------
// 32x32x16 form: eight waves per workgroup (one workgroup per CU), RB blocks of 32 query rows per wave
// (8 x RB x 32 <= 512 rows), page tiles of 32 patches.  Every staged tile is read by eight waves instead of four,
// the ring is one per CU (4 x 32 KiB chunks, 96 KiB in flight), and the larger MFMA shape has the higher ceiling
// (2382 vs 2075 TFLOP/s in the guide's micro-benchmarks).  D layout: col = lane & 31 (patch),
// row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5) for accumulator register t.
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kChunk32Bytes = 8 * kTileBytes;  // 32 KiB = 128 patch rows

template <int RB, int S>
__global__ __launch_bounds__(512, 2) void maxsim_batch32_kernel(BKArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[S * kChunk32Bytes + 8 * RB * 32 * 4];
  float* red = reinterpret_cast<float*>(lds + S * kChunk32Bytes);  // [256 * RB] row maxima of the current page
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n32 = lane & 31, h = lane >> 5;

  bf16x8 qa[RB][8];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      qa[rb][kk] = *reinterpret_cast<const bf16x8*>(a.q + ((size_t)(wave * RB + rb) * 32 + n32) * kDim + kk * 16 + h * 8);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qa[rb][kk]));

  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  // fragment offsets inside a 32-patch tile: patch n32 = 16-row sub-tile (n32 >> 4), row w = n32 & 15, 16-byte chunk 2kk + h
  int rd_off[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) rd_off[kk] = (n32 >> 4) * kTileBytes + (n32 & 15) * kRowBytes + ((((kk * 2 + h) ^ (n32 & 15))) << 4);

  for (int64_t item = blockIdx.x; item < a.n; item += gridDim.x) {
    const int64_t page = a.page0 + item;
    if (bk_masked(a, page)) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
      continue;
    }
    const int nr = a.n_rows ? a.n_rows[page] : a.stride;
    if (nr <= 0) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
      continue;
    }
    const int ntiles32 = (nr + 31) / 32;
    const int nchunks = (nr + 127) / 128;
    const char* pbase = a.slab + (size_t)page * (size_t)a.stride * kRowBytes;

    auto issue = [&](int c) {  // wave w moves 16-row tile w of chunk c
      const char* tp = pbase + (size_t)(c * 8 + wave) * kTileBytes;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (c % S) * kChunk32Bytes + wave * kTileBytes));
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
    for (int c = 0; c < S - 1; ++c)
      if (c < nchunks) issue(c);

    f32x16 mx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int t = 0; t < 16; ++t) mx[rb][t] = -INFINITY;

    for (int c = 0; c < nchunks; ++c) {
      const int ahead = min(S - 2, nchunks - 1 - c);
      if (ahead >= 2) bk_wait_vmcnt<8>();
      else if (ahead == 1) bk_wait_vmcnt<4>();
      else bk_wait_vmcnt<0>();
      bk_barrier();
      if (c + S - 1 < nchunks) issue(c + S - 1);
      const char* chunk = lds + (c % S) * kChunk32Bytes;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int t32 = c * 4 + tt;
        if (t32 < ntiles32) {  // block-uniform
          bf16x8 b[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) b[kk] = *reinterpret_cast<const bf16x8*>(chunk + tt * 2 * kTileBytes + rd_off[kk]);
          f32x16 acc[RB];
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[rb][t] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[rb][kk], b[kk], acc[rb], 0, 0, 0);
          if ((t32 + 1) * 32 > nr) {
            const bool col_valid = t32 * 32 + n32 < nr;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
              for (int t = 0; t < 16; ++t)
                if (!col_valid) acc[rb][t] = -INFINITY;
          }
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int t = 0; t < 16; ++t) mx[rb][t] = fmaxf(mx[rb][t], acc[rb][t]);
        }
      }
    }

    // max over the 32 patch columns (lanes with equal lane >> 5), then one thread per query sums its rows
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float v = bk_group16_max(mx[rb][t]);
        v = fmaxf(v, __shfl_xor(v, 16));
        if (n32 == 0) red[(wave * RB + rb) * 32 + (t & 3) + 8 * (t >> 2) + 4 * h] = v;
      }
    __syncthreads();
    if ((int)threadIdx.x < a.n_queries) {
      const float* rp = red + (size_t)threadIdx.x * a.rows_per_query;
      float sum = 0.f;
      for (int i = 0; i < a.rows_per_query; ++i) sum += rp[i];
      if (a.allow && a.allow_stride_bits) {
        const int32_t o = a.doc_ord[page];
        const uint32_t* ab = a.allow + (size_t)threadIdx.x * (size_t)(a.allow_stride_bits >> 5);
        if ((int64_t)o >= a.n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u) sum = -INFINITY;
      }
      a.scores[(size_t)threadIdx.x * a.score_stride + item] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Round 3: 32x32x16 MFMA with the TRANSPOSED roles of maxsim_batch_kernel.  A = 32 patches x 16 dims (from LDS),
// B = 16 dims x 32 query tokens (registers, resident for the whole launch), D[row = patch][col = token]: lane l holds
// token l & 31 and 16 of the tile's 32 patches (rows (i & 3) + 8 (i >> 2) + 4 (l >> 5) of accumulator register i), so
// the running maximum over patches is ONE register per 32-token tile, fed by 8 v_max3_f32 per 8 MFMAs (one VALU op per
// 32-cycle MFMA), and a page ends with ONE cross-half exchange per token tile.  Against the 16x16x32 form every byte
// read from LDS feeds twice the flops and the matrix pipe reads half the operand registers per flop (the shape the
// guide's micro-benchmark puts ~15 % above 16x16x32).  The round-1 32x32x16 kernel above (variant 1) has the
// untransposed roles: 16 running maxima per tile and a 32-lane reduction per accumulator register.
//
// Work split: RS waves split the query rows (NT tiles of 32 tokens per wave), the other PS = 4 / RS-way split is over
// the two 32-patch tiles of a 16 KiB ring chunk.  RS = 4: 128 rows per wave, two workgroups per CU (<= 256 VGPRs);
// RS = 2: 256 rows per wave at ONE wave per SIMD (512-VGPR budget), every staged tile is read by two waves instead of
// four.  The ring is the row-split kernel's (each wave DMAs one 16-row sub-tile of every chunk, XOR-swizzled image),
// but the barrier runs ONE chunk ahead: the barrier at the top of iteration c publishes chunk c + 1, and a tile's
// fragments are re-filled IN PLACE for the next tile while the last pair of token tiles still multiplies (register
// b[kk] is dead once its last MFMA has issued) -- no LDS latency between tiles and no second fragment set.
template <int NT, int RS, int S>
__global__ __launch_bounds__(256, (RS == 4 && NT <= 3) ? 2 : 1) void maxsim_batch32t_kernel(BKArgs a) {
  constexpr int PS = 4 / RS;
  constexpr int LAG = PS == 1 ? 1 : 0;  // PS = 1: a wave still reads tile 1 of chunk c during iteration c, so the slot freed at its top is chunk c-1's
  constexpr int ROWS = RS * NT * 32;
  constexpr bool PEND = RS == 2;  // one wave per SIMD: the last group's folds ride behind the next tile's first MFMAs
  constexpr int kTile32 = 2 * kTileBytes;  // 8 KiB = 32 patch rows
  __shared__ __attribute__((aligned(16))) char lds[S * kChunkBytes + PS * ROWS * 4 + (ROWS / 16) * 4];
  float* red = reinterpret_cast<float*>(lds + S * kChunkBytes);  // [PS][ROWS] token maxima of the current page
  float* part = red + PS * ROWS;                                 // [ROWS / 16] sums of 16 rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n32 = lane & 31, h = lane >> 5;
  const int rg = wave % RS, pg = wave / RS;

  // query fragments (B operand): token row (rg * NT + m) * 32 + n32, dims kk * 16 + 8 h .. + 7
  bf16x8 qb[NT][8];
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      qb[m][kk] = *reinterpret_cast<const bf16x8*>(a.q + ((size_t)((rg * NT + m) * 32 + n32)) * kDim + kk * 16 + h * 8);
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (RS == 2) asm volatile("" : "+a"(qb[m][kk]));  // one wave per SIMD: the query fragments live in the AGPR half of the 512-entry file (MFMA reads them there)
      else asm volatile("" : "+v"(qb[m][kk]));
    }

  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  // A fragment of MFMA kk inside a 32-patch tile: patch n32 = 16-row sub-tile n32 >> 4, row n32 & 15, 16-byte chunk 2 kk + h
  int rd_off[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) rd_off[kk] = (n32 >> 4) * kTileBytes + (n32 & 15) * kRowBytes + (((kk * 2 + h) ^ (n32 & 15)) << 4);

  for (int64_t item = blockIdx.x; item < a.n; item += gridDim.x) {
    const int64_t page = a.page0 + item;
    if (bk_masked(a, page)) {  // block-uniform
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = -INFINITY;
      continue;
    }
    const int nr = a.n_rows ? a.n_rows[page] : a.stride;
    if (nr <= 0) {
      if ((int)threadIdx.x < a.n_queries) a.scores[(size_t)threadIdx.x * a.score_stride + item] = 0.0f;
      continue;
    }
    const int ntiles = (nr + 31) / 32;
    const int nchunks = (nr + 63) / 64;
    const char* pbase = a.slab + (size_t)page * (size_t)a.stride * kRowBytes;

    auto issue = [&](int c) {  // wave w moves 16-row sub-tile w of chunk c
      const char* tp = pbase + (size_t)(c * kChunkTiles + wave) * kTileBytes;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + (c % S) * kChunkBytes + wave * kTileBytes));
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
    auto wait_chunks = [&](int ahead) {  // at most `ahead` whole chunks (4 DMA instructions each) of this wave may stay in flight
      if (ahead >= 5) bk_wait_vmcnt<20>();
      else if (ahead == 4) bk_wait_vmcnt<16>();
      else if (ahead == 3) bk_wait_vmcnt<12>();
      else if (ahead == 2) bk_wait_vmcnt<8>();
      else if (ahead == 1) bk_wait_vmcnt<4>();
      else bk_wait_vmcnt<0>();
    };

#pragma unroll
    for (int c = 0; c < S - LAG; ++c)
      if (c < nchunks) issue(c);

    float mx[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) mx[m] = -INFINITY;

    wait_chunks(min(S - 1 - LAG, nchunks - 1));
    bk_barrier();  // chunk 0 visible
    // Fragment addresses of tile t (tile base `tb` in LDS).  The last tile of a ragged page holds patches past n_rows:
    // instead of masking 16 accumulator registers per token tile, the lanes of such patches read patch 0 of the tile
    // (always valid) -- a duplicated patch cannot change a maximum -- so the tile body has ONE form and no masks.
    int ra[8];
    auto frag_addr = [&](int t, const char* tb) {
      const int base = (int)(uintptr_t)(__attribute__((address_space(3))) const char*)tb;
      if ((t + 1) * 32 > nr) {  // wave-uniform
        const bool ok = t * 32 + n32 < nr;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = base + (ok ? rd_off[kk] : (kk * 2 + h) << 4);
      } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = base + rd_off[kk];
      }
    };
    auto lds_frag = [&](int addr) {
      typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
      return *(lds_frag_t*)(uintptr_t)(uint32_t)addr;  // a 32-bit LDS address -> ds_read_b128
    };
    bf16x8 b[8];
    frag_addr(pg, lds + pg * kTile32);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) b[kk] = lds_frag(ra[kk]);

    // One 32-patch tile against this wave's NT token tiles, in groups of two token tiles (two independent MFMA chains
    // of 8).  Software pipeline: the v_max3 folds of a group's accumulators ride behind the NEXT group's MFMAs (one VALU
    // op per MFMA; the last group's accumulators stay pending across the tile boundary and are folded behind the first
    // group of the wave's next tile), and b[] is re-filled from ra[] (the wave's next tile) behind the last group's
    // MFMAs.  The body is branch-free: one scheduling region.
    constexpr int NG = (NT + 1) / 2;
    f32x16 acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = -INFINITY;  // "tile -1": max-neutral
    auto fold = [&](int m) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) mx[m] = fmaxf(fmaxf(mx[m], acc[m][i]), acc[m][i + 1]);  // v_max3_f32
    };
    auto tile = [&]() {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int m0 = 2 * g, m1 = 2 * g + 1 < NT ? 2 * g + 1 : 2 * g;
        const bool two = 2 * g + 1 < NT;
        const int pgp = (g + NG - 1) % NG;  // the group whose accumulators are folded behind this group's MFMAs
        const int p0 = 2 * pgp, p1 = 2 * pgp + 1 < NT ? 2 * pgp + 1 : 2 * pgp;
        const bool ptwo = 2 * pgp + 1 < NT;
        f32x16 n0, n1;
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk], qb[m0][kk], kk == 0 ? z : n0, 0, 0, 0);
          if (two) n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk], qb[m1][kk], kk == 0 ? z : n1, 0, 0, 0);
          if (g == NG - 1) b[kk] = lds_frag(ra[kk]);
        }
        if (PEND || g > 0) {  // previous group: of this tile (g > 0) or of the wave's previous tile (g == 0)
          fold(p0);
          if (ptwo) fold(p1);
        }
        acc[m0] = n0;
        if (two) acc[m1] = n1;
        if (!PEND && g == NG - 1) {  // two waves per SIMD: the partner's MFMAs cover this fold; no accumulators live across tiles
          fold(m0);
          if (two) fold(m1);
        }
        // issue order: one VALU op behind every MFMA, the fragment re-fills spread over the last group
#pragma unroll
        for (int i = 0; i < (two ? 16 : 8); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (PEND || g > 0) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          if (g == NG - 1 && (i % (two ? 2 : 1)) == (two ? 1 : 0)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // b[kk] is free after its second MFMA
        }
      }
    };

    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) {  // block-uniform: publish chunk c + 1, release the slot of chunk c - LAG
        wait_chunks(min(c + S - 1 - LAG, nchunks - 1) - (c + 1));
        bk_barrier();
        if (c + S - LAG < nchunks) issue(c + S - LAG);
      }
      const char* cur = lds + (c % S) * kChunkBytes;
      const char* nx = lds + ((c + 1) % S) * kChunkBytes;
      if (PS == 1) {
        const int t0 = 2 * c;
        if (t0 + 1 < ntiles) frag_addr(t0 + 1, cur + kTile32);  // else: ra[] keeps pointing at a landed tile (the re-fill is unused)
        tile();
        if (t0 + 1 < ntiles) {
          if (t0 + 2 < ntiles) frag_addr(t0 + 2, nx);
          tile();
        }
      } else {
        const int t = 2 * c + pg;
        if (t < ntiles) {  // wave-uniform (no barrier inside)
          if (t + 2 < ntiles) frag_addr(t + 2, nx + pg * kTile32);
          tile();
        }
      }
    }
    if (PEND) {  // drain: the last group's accumulators of the wave's last tile
      constexpr int l0 = 2 * (NG - 1);
      fold(l0);
      if (l0 + 1 < NT) fold(l0 + 1);
    }

    // token maxima: lanes l and l ^ 32 hold the two halves of the tile's patches
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      float v = mx[m];
      v = fmaxf(v, __shfl_xor(v, 32));
      if (h == 0) red[pg * ROWS + (rg * NT + m) * 32 + n32] = v;
    }
    __syncthreads();
    const int rows_total = a.n_queries * a.rows_per_query;  // <= ROWS
    if (PS == 2) {
      for (int t = threadIdx.x; t < rows_total; t += 256) red[t] = fmaxf(red[t], red[ROWS + t]);
      __syncthreads();
    }
    bk_write_scores(a, red, part, rows_total, page, item);
    // the next page's first bk_barrier() orders these reads of red[] / part[] before their rewrite; its prologue DMAs
    // only start after the __syncthreads() above, i.e. after every wave's last fragment read
  }
}

template <int MTW>
int launch_batch_mtw(const BKArgs& k, int grid, hipStream_t s) {
  // (Round 2 measured two further forms of this kernel on uniform corpora -- one DMA stream across page boundaries,
  // and that stream WITHOUT the per-chunk barrier as a timing probe: both ran in exactly the time of this kernel
  // (profiles/r2/batched_variants_200k_barrier_probe.json: 18.85 / 19.17 / 19.03 ms at B = 16), i.e. neither page-start
  // bubbles nor barrier skew limit it; see DESIGN.md 3.5 for what does.)
  hipLaunchKernelGGL((maxsim_batch_kernel<MTW, 4>), dim3((unsigned)grid), dim3(256), 0, s, k);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace

int batch_rows_capacity() { return 512; }

int launch_maxsim_batch(const BatchArgs& a, hipStream_t s) {
  if (a.n <= 0 || a.n_queries <= 0) return MV_OK;
  if (a.rows_per_query < 16 || a.rows_per_query % 16) { set_error("batch scan: rows_per_query must be a positive multiple of 16"); return MV_ERR_INVALID; }
  const int rows = a.n_queries * a.rows_per_query;
  if (rows > 512 || a.n_queries > 256) { set_error("batch scan: %d query rows exceed the 512-row group", rows); return MV_ERR_INVALID; }
  BKArgs k{reinterpret_cast<const char*>(a.slab), a.n_rows, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n,
           a.score_stride, 0, a.stride, a.n_queries, a.rows_per_query, a.allow_stride_bits};
  static int ncu = 0;  // CUs of the (single-architecture) node's GPUs, queried once
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    else ncu = 256;
  }
  if (a.variant == 1 && rows > 128) {  // 32x32x16 form: one 512-thread workgroup per CU
    const int grid32 = (int)std::min<int64_t>(a.n, (int64_t)ncu);
    if (rows <= 256) hipLaunchKernelGGL((maxsim_batch32_kernel<1, 4>), dim3((unsigned)grid32), dim3(512), 0, s, k);
    else hipLaunchKernelGGL((maxsim_batch32_kernel<2, 4>), dim3((unsigned)grid32), dim3(512), 0, s, k);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if ((a.variant == 5 || a.variant == 6) && rows > 0) {  // transposed 32x32x16 forms (round 3)
    if (a.variant == 5) {  // four row groups, two workgroups per CU
      const int grid5 = (int)std::min<int64_t>(a.n, (int64_t)ncu * (rows > 384 ? 1 : 2));  // 4 token tiles per wave need > 256 registers: one workgroup per CU
      switch ((rows + 127) / 128) {
        case 1: hipLaunchKernelGGL((maxsim_batch32t_kernel<1, 4, 4>), dim3((unsigned)grid5), dim3(256), 0, s, k); break;
        case 2: hipLaunchKernelGGL((maxsim_batch32t_kernel<2, 4, 4>), dim3((unsigned)grid5), dim3(256), 0, s, k); break;
        case 3: hipLaunchKernelGGL((maxsim_batch32t_kernel<3, 4, 4>), dim3((unsigned)grid5), dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL((maxsim_batch32t_kernel<4, 4, 4>), dim3((unsigned)grid5), dim3(256), 0, s, k); break;
      }
    } else {  // two row groups x two tile groups, one wave per SIMD
      const int grid6 = (int)std::min<int64_t>(a.n, (int64_t)ncu);
      switch ((rows + 63) / 64) {
        case 1: hipLaunchKernelGGL((maxsim_batch32t_kernel<1, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 2: hipLaunchKernelGGL((maxsim_batch32t_kernel<2, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 3: hipLaunchKernelGGL((maxsim_batch32t_kernel<3, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 4: hipLaunchKernelGGL((maxsim_batch32t_kernel<4, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 5: hipLaunchKernelGGL((maxsim_batch32t_kernel<5, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 6: hipLaunchKernelGGL((maxsim_batch32t_kernel<6, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        case 7: hipLaunchKernelGGL((maxsim_batch32t_kernel<7, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
        default: hipLaunchKernelGGL((maxsim_batch32t_kernel<8, 2, 6>), dim3((unsigned)grid6), dim3(256), 0, s, k); break;
      }
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  // variant: -1 / 0 = auto (page-split form up to 128 rows, transposed row-split form above), 3 = row-split form always,
  // 2 = the round-1 two-stage pipeline (<= 384 rows), 1 = 32x32x16 form (handled above)
  if ((a.variant <= 0 || a.variant == 4) && rows <= 128) {
    if (a.n > 0x7fffffffLL) { set_error("batch scan: too many pages for one launch"); return MV_ERR_INVALID; }
    const int mt = (rows + 15) / 16;
    constexpr int64_t kChunk = (int64_t)1 << 22;  // work-item count per launch stays below 2^32
    for (int64_t off = 0; off < a.n; off += kChunk) {
      BKArgs kk = k;
      kk.n = std::min(kChunk, a.n - off);
      kk.page0 = off;
      kk.scores = a.scores + off;
      const dim3 gr((unsigned)kk.n), bl(256);
      switch (mt) {
        case 1: hipLaunchKernelGGL((maxsim_batch_ps_kernel<1, 4>), gr, bl, 0, s, kk); break;
        case 2: hipLaunchKernelGGL((maxsim_batch_ps_kernel<2, 4>), gr, bl, 0, s, kk); break;
        case 3: hipLaunchKernelGGL((maxsim_batch_ps_kernel<3, 4>), gr, bl, 0, s, kk); break;
        case 4: hipLaunchKernelGGL((maxsim_batch_ps_kernel<4, 4>), gr, bl, 0, s, kk); break;
        case 5: hipLaunchKernelGGL((maxsim_batch_ps_kernel<5, 4>), gr, bl, 0, s, kk); break;
        case 6: hipLaunchKernelGGL((maxsim_batch_ps_kernel<6, 4>), gr, bl, 0, s, kk); break;
        case 7: hipLaunchKernelGGL((maxsim_batch_ps_kernel<7, 4>), gr, bl, 0, s, kk); break;
        default: hipLaunchKernelGGL((maxsim_batch_ps_kernel<8, 4>), gr, bl, 0, s, kk); break;
      }
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  const int grid = (int)std::min<int64_t>(a.n, (int64_t)ncu * 2);
  if (a.variant == 2 && rows <= 384) {  // round-1 pipelined 16x16x32 form (kept as a cross-check)
    const int mtw6 = (rows + 63) / 64;
    switch (mtw6) {
      case 1: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<1, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
      case 2: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<2, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
      case 3: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<3, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
      case 4: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<4, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
      case 5: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<5, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
      default: hipLaunchKernelGGL((maxsim_batch_pipe_kernel<6, 4>), dim3((unsigned)grid), dim3(256), 0, s, k); break;
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  const int mtw = (rows + 63) / 64;  // query tiles per wave: no MFMA is spent on more than 63 padding rows
  switch (mtw) {
    case 1: return launch_batch_mtw<1>(k, grid, s);
    case 2: return launch_batch_mtw<2>(k, grid, s);
    case 3: return launch_batch_mtw<3>(k, grid, s);
    case 4: return launch_batch_mtw<4>(k, grid, s);
    case 5: return launch_batch_mtw<5>(k, grid, s);
    case 6: return launch_batch_mtw<6>(k, grid, s);
    case 7: return launch_batch_mtw<7>(k, grid, s);
    default: return launch_batch_mtw<8>(k, grid, s);
  }
}

}  // namespace mv
