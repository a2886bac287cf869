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
  const int64_t* row_off;     // packed layout: first slab row of every page; null: page * stride
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
template <int MTW, int S, bool PK = false>  // PK: packed layout (row-offset table); a template parameter so the fixed layout's code is untouched
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
    const char* pbase = a.slab + (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride) * kRowBytes;

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
template <int MT, int D, bool PK = false>
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
  const char* pbase = a.slab + (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride) * kRowBytes;
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
// Forms that lost by measurement and were removed in round 5 (records: profiles/r1-r3 batched_variants_*.json):
//   variant 2  two-stage software pipeline of the 16x16x32 form (<= 384 rows)      round 1
//   variant 1  32x32x16 MFMA, eight waves per workgroup, untransposed roles        round 1
//   variants 5 / 6  32x32x16 MFMA with the transposed roles (4 / 2 row groups)     round 3: 1.36-1.42 PF against 1.50 for the default

template <int MTW>
int launch_batch_mtw(const BKArgs& k, int grid, hipStream_t s) {
  // (Round 2 measured two further forms of this kernel on uniform corpora -- one DMA stream across page boundaries,
  // and that stream WITHOUT the per-chunk barrier as a timing probe: both ran in exactly the time of this kernel
  // (profiles/r2/batched_variants_200k_barrier_probe.json: 18.85 / 19.17 / 19.03 ms at B = 16), i.e. neither page-start
  // bubbles nor barrier skew limit it; see DESIGN.md 3.5 for what does.)
  if (k.row_off) hipLaunchKernelGGL((maxsim_batch_kernel<MTW, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, k);
  else hipLaunchKernelGGL((maxsim_batch_kernel<MTW, 4>), dim3((unsigned)grid), dim3(256), 0, s, k);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace

int batch_rows_capacity() { return 512; }

int launch_maxsim_batch(const BatchArgs& a, hipStream_t s) {
  if (a.n <= 0 || a.n_queries <= 0) return MV_OK;
  if (a.row_off && !a.n_rows) { set_error("batch scan: a row-offset table needs the per-page row counts"); return MV_ERR_INVALID; }
  if (a.rows_per_query < 16 || a.rows_per_query % 16) { set_error("batch scan: rows_per_query must be a positive multiple of 16"); return MV_ERR_INVALID; }
  const int rows = a.n_queries * a.rows_per_query;
  if (rows > 512 || a.n_queries > 256) { set_error("batch scan: %d query rows exceed the 512-row group", rows); return MV_ERR_INVALID; }
  BKArgs k{reinterpret_cast<const char*>(a.slab), a.n_rows, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n,
           a.score_stride, 0, a.stride, a.n_queries, a.rows_per_query, a.allow_stride_bits, a.row_off};
  static int ncu = 0;  // CUs of the (single-architecture) node's GPUs, queried once
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    else ncu = 256;
  }
  // variant: -1 / 0 = auto (page-split form up to 128 rows, row-split form above), 4 = page-split form, 3 = row-split form always
  if (a.variant != 0 && a.variant != 3 && a.variant != 4 && a.variant >= 0 && a.variant != 7 && a.variant != 8) {  // (7 / 8 name forms of the e4m3 batch scan)
    set_error("unknown batch variant %d (0 = auto, 3 = row-split, 4 = page-split)", a.variant);
    return MV_ERR_INVALID;
  }
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
        case 1: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<1, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<1, 4>), gr, bl, 0, s, kk); break;
        case 2: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<2, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<2, 4>), gr, bl, 0, s, kk); break;
        case 3: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<3, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<3, 4>), gr, bl, 0, s, kk); break;
        case 4: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<4, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<4, 4>), gr, bl, 0, s, kk); break;
        case 5: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<5, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<5, 4>), gr, bl, 0, s, kk); break;
        case 6: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<6, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<6, 4>), gr, bl, 0, s, kk); break;
        case 7: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<7, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<7, 4>), gr, bl, 0, s, kk); break;
        default: if (kk.row_off) hipLaunchKernelGGL((maxsim_batch_ps_kernel<8, 4, true>), gr, bl, 0, s, kk); else hipLaunchKernelGGL((maxsim_batch_ps_kernel<8, 4>), gr, bl, 0, s, kk); break;
      }
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  const int grid = (int)std::min<int64_t>(a.n, (int64_t)ncu * 2);
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
