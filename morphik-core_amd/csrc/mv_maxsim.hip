// mv_maxsim.hip -- fused float MaxSim page scan for gfx950 (MI355X).
//
// Replaces processor.score_multi_vector(...) (colpali_engine v0.3.13, called at
// core/vector_store/fast_multivector_store.py:553-555): per page
//     S[Q x P] = Qm[Q x 128] . Pg^T[128 x P]   (bf16 x bf16 -> fp32 on MFMA)
//     score    = sum_q max_p S[q][p]
// The reference materialises S (einsum), then max, then sum -- three kernels and a (Q x P) fp32
// intermediate per page.  Here one kernel streams each 256-byte patch row from HBM exactly once,
// keeps the running row-max in the MFMA accumulator layout and writes ONE float per page.
//
// The scan is HBM-bound (32 flop/byte at Q=32 against a ~315 flop/byte ridge): the kernel's job is
// to keep >= 32 KiB per CU in flight with perfectly coalesced reads; MFMA time is ~10% of the
// memory time.  Bytes per page = n_rows * 256.
//
// MFMA operand mapping (v_mfma_f32_16x16x32_bf16, D = A.B + C):
//   A = query tile   : lane l holds Q[m*16 + (l&15)][32j + 8(l>>4) .. +8)      (loop invariant, VGPRs)
//   B = page tile^T  : lane l holds P[16t + (l&15)][32j + 8(l>>4) .. +8)       (streamed)
//   D[row = 4(l>>4)+i][col = l&15] -> query row 16m+4(l>>4)+i, patch 16t+(l&15)
// so the max over patches is an elementwise max over tiles t followed by ONE 16-lane butterfly.
//
// Variants (MV_OPT_MAXSIM_VARIANT; measured table in DESIGN.md):
//   6  (default, pages of >= 512 rows) non-temporal LDS-DMA (global_load_lds_dwordx4 nt) into a wave-private 4-slot ring,
//      XOR-swizzled rows, ds_read_b128 fragments, FOUR waves per page (tiles interleaved)
//   7  (default below 512 rows) the same ring, one wave per page
//   0  direct global -> VGPR fragment loads, one wave per page, 3-tile register ring: the independent cross-check
//   13 the default's transport with the arithmetic removed (calibration)
#include <algorithm>

#include "mv_common.h"

namespace mv {
namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMaxQTiles = 8;  // 128 query rows per pass (A fragments: 16 VGPRs per tile)
constexpr int kMaxQTilesLo = 4;  // 64 rows per pass when the query's lo half rides along (32 VGPRs per tile)

struct KArgs {
  const char* slab;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const int32_t* cand;
  const uint16_t* q;
  float* scores;
  int64_t n;
  int32_t stride;
  int32_t pad_to;
  int64_t page0;  // first page of this launch when there is no candidate list
  const int32_t* pad_items;  // per-item pad_to (rerank batches of 128 pad independently); null -> pad_to
  int32_t items_per_q;       // QITEM kernels: work item i scores against the query at q + (i / items_per_q) * q_item_stride
  int32_t q_item_stride;     // (bf16 elements) -- the candidate lists of a batch of queries in ONE launch
  const uint16_t* qlo;       // LO >= 1: lo half of the query rows (bf16(q - bf16(q)); layout of q)
  const char* slab_lo;       // LO == 2: lo half of the page rows (layout of slab)
  const int64_t* row_off;    // packed layout: first slab row of every page; null: page * stride
};

// byte offset of a page's first row in a slab of 256-byte rows.  PK (packed layout) is a template parameter, not a test of a.row_off:
// the fixed layout's kernels are instruction for instruction those of round 5 (the extra kernarg load + branch sat in front of the first
// DMA of every fresh workgroup), and the packed kernels issue the table load beside the n_rows load instead of behind a branch.
template <bool PK>
__device__ __forceinline__ size_t page_byte_off(const KArgs& a, int64_t page) {
  return (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride) * kRowBytes;
}

__device__ __forceinline__ bool page_masked(const KArgs& a, int64_t page) {
  if (!a.doc_ord) return false;
  int32_t o = a.doc_ord[page];
  if (o < 0) return true;
  if (!a.allow) return false;
  if ((int64_t)o >= a.n_allow_bits) return true;
  return ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u;
}

__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 8));
  return v;
}

template <int MT>
__device__ __forceinline__ void load_query(const uint16_t* q, int r, int g, bf16x8 (&a)[MT][4]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[m][j] = *reinterpret_cast<const bf16x8*>(q + (m * 16 + r) * kDim + j * 32 + g * 8);
}

// One 16-patch tile against MT query tiles; running elementwise max in mx.
template <int MT>
__device__ __forceinline__ void tile_mfma(const bf16x8 (&a)[MT][4], const bf16x8 (&b)[4], f32x4 (&mx)[MT], bool mask_cols,
                                          bool col_valid) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][j], b[j], acc, 0, 0, 0);
    if (mask_cols && !col_valid) acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[i]);
  }
}

// Split-bf16 form: the operands are hi + lo pairs of bf16 values (x = hi + lo to 2^-18 |x|), the products accumulate in the
// SAME fp32 accumulator before the max -- the maximum runs over the fp32-faithful sums, as the reference's fp32 einsum -> max does.
//   LO == 1  query hi + lo against bf16 pages:        S = qh.p + ql.p
//   LO == 2  ... against page hi + lo (the lo slab):  S = qh.ph + ql.ph + qh.pl   (ql.pl <= 2^-18 of the product: dropped)
// The hi.hi chain comes first, so operands that ARE bf16 (lo = 0) give the bits of tile_mfma (x + 0 = x).
template <int MT, int LO>
__device__ __forceinline__ void tile_mfma_lo(const bf16x8 (&a)[MT][4], const bf16x8 (&al)[MT][4], const bf16x8 (&b)[4], const bf16x8 (&bl)[4],
                                             f32x4 (&mx)[MT], bool mask_cols, bool col_valid) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][j], b[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m][j], b[j], acc, 0, 0, 0);
    if (LO == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][j], bl[j], acc, 0, 0, 0);
    }
    if (mask_cols && !col_valid) acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < 4; ++i) mx[m][i] = fmaxf(mx[m][i], acc[i]);
  }
}

// Lane-level finish for ONE wave that saw all of the page's tiles: returns the page score in every lane.
template <int MT>
__device__ __forceinline__ float finish_wave(const f32x4 (&mx)[MT], bool clamp) {
  float total = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = group16_max(mx[m][i]);
      if (clamp) v = fmaxf(v, 0.f);
      if (v == -INFINITY) v = 0.f;
      total += v;
    }
  total += __shfl_xor(total, 16);
  total += __shfl_xor(total, 32);
  return total;
}

// Cross-wave finish for four waves that split one page's tiles. red: 4 x 128 floats of LDS.
template <int MT>
__device__ __forceinline__ void finish_block(const f32x4 (&mx)[MT], bool clamp, float* red, int wave, int lane,
                                             float* out) {
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = group16_max(mx[m][i]);
      if (r == 0) red[wave * 128 + m * 16 + g * 4 + i] = v;
    }
  __syncthreads();
  if (wave == 0) {
    float v = 0.f;
#pragma unroll
    for (int h = 0; h < (MT * 16 + 63) / 64; ++h) {  // query rows lane, lane + 64
      const int row = lane + 64 * h;
      if (row < MT * 16) {
        float x = fmaxf(fmaxf(red[row], red[128 + row]), fmaxf(red[256 + row], red[384 + row]));
        if (clamp) x = fmaxf(x, 0.f);
        if (x == -INFINITY) x = 0.f;
        v += x;
      }
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    if (lane == 0) *out = v;
  }
}

// --------------------------------------------------------------------------------------------
// Variants 0/1/4/5: direct global -> VGPR fragment loads (64 contiguous bytes per row per instruction).
template <int MT, int WPP, bool NT, int LO = 0, bool PK = false>
__global__ __launch_bounds__(256) void maxsim_direct_kernel(KArgs a) {
  constexpr int PF = 3;  // register ring depth (tiles in flight per wave = PF-1 .. PF)
  __shared__ float red[512];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t item = (WPP == 1) ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (item >= a.n) return;
  const int64_t page = a.cand ? (int64_t)a.cand[item] : a.page0 + item;
  if (page < 0 || page_masked(a, page)) {  // page < 0: padding entry of a device-built candidate list
    if (lane == 0 && (WPP == 1 || wave == 0)) a.scores[item] = -INFINITY;
    return;
  }
  // packed layout: the table entry is loaded BESIDE the row count (both scalar loads before the one s_waitcnt; a packed index always carries n_rows)
  int64_t pk_row0 = 0;
  if constexpr (PK) pk_row0 = a.row_off[page];
  const int nr = PK ? a.n_rows[page] : (a.n_rows ? a.n_rows[page] : a.stride);
  const int ntiles = (nr + kTileRows - 1) / kTileRows;
  const bool clamp = (a.pad_items ? a.pad_items[item] : a.pad_to) > nr;

  bf16x8 qa[MT][4];
  load_query<MT>(a.q, r, g, qa);
  bf16x8 qal[LO ? MT : 1][4];
  if constexpr (LO > 0) load_query<MT>(a.qlo, r, g, qal);

  f32x4 mx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mx[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

  const size_t poff = (PK ? (size_t)pk_row0 * kRowBytes : page_byte_off<false>(a, page)) + r * kRowBytes + g * 16;
  const char* base = a.slab + poff;
  const char* base_lo = LO == 2 ? a.slab_lo + poff : nullptr;
  const int t0 = (WPP == 1) ? 0 : wave;
  const int ntw = (ntiles - t0 + WPP - 1) / WPP;  // tiles owned by this wave (may be <= 0)

  bf16x8 buf[PF][4];
  bf16x8 bufl[LO == 2 ? PF : 1][4];
  auto load_from = [&](const char* bs, bf16x8(&b)[4], int it) {
    const char* tp = bs + (size_t)(t0 + it * WPP) * kTileBytes;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8* p = reinterpret_cast<const bf16x8*>(tp + j * 64);
      b[j] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  };
  auto load_tile = [&](int slot, int it) {
    load_from(base, buf[slot], it);
    if constexpr (LO == 2) load_from(base_lo, bufl[slot], it);
  };
#pragma unroll
  for (int i = 0; i < PF - 1; ++i)
    if (i < ntw) load_tile(i, i);

  for (int it0 = 0; it0 < ntw; it0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int it = it0 + u;
      if (it < ntw) {
        if (it + PF - 1 < ntw) load_tile((u + PF - 1) % PF, it + PF - 1);
        const int t = t0 + it * WPP;
        const bool partial = (t + 1) * kTileRows > nr;
        if constexpr (LO == 0) tile_mfma<MT>(qa, buf[u], mx, partial, t * kTileRows + r < nr);
        else if constexpr (LO == 1) tile_mfma_lo<MT, 1>(qa, qal, buf[u], buf[u], mx, partial, t * kTileRows + r < nr);
        else tile_mfma_lo<MT, 2>(qa, qal, buf[u], bufl[u], mx, partial, t * kTileRows + r < nr);
      }
    }
  }

  if (WPP == 1) {
    float s = finish_wave<MT>(mx, clamp);
    if (lane == 0) a.scores[item] = s;
  } else {
    finish_block<MT>(mx, clamp, red, wave, lane, &a.scores[item]);
  }
}

// --------------------------------------------------------------------------------------------
// Variants 2/3: LDS-DMA into a wave-private ring.  No barrier: a wave only ever reads what it
// DMA'd itself, ordered by its own counted s_waitcnt vmcnt.
//
// LDS image of a tile (4 KiB): row-major 16 rows x 256 B, 16-byte chunk c of row w stored at chunk
// position c ^ w (XOR swizzle) so that the ds_read_b128 fragment reads (16 rows, same logical chunk)
// hit 16 distinct 16-byte bank slots.  The DMA writes LDS linearly (M0 base + lane*16), so the
// swizzle is applied to the per-lane GLOBAL source address (same involution both sides); every DMA
// instruction still reads 4 whole rows = 1 KiB contiguous from HBM.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// STREAM_ONLY: the same ring, the same waits, no fragment reads and no MFMA -- the transport ceiling of this kernel
// (MV_CAL_READ_LDSDMA: the roofline's measured denominator).
//
// LO (split-bf16 operands, tile_mfma_lo): 1 = the query's lo half rides along in registers; 2 = also the pages' lo slab --
// a ring item is then the PAIR (hi tile, lo tile) of one 16-row tile, 8 KiB, the ring D pairs deep (128 KiB of LDS per
// workgroup at D = 4: one workgroup per CU with the bytes in flight of two hi-only ones).
template <int MT, int WPP, int D, bool NT = false, bool CONTIG = false, bool STREAM_ONLY = false, bool QITEM = false, int LO = 0, bool PK = false>
__global__ __launch_bounds__(256) void maxsim_ldsdma_kernel(KArgs a) {
  constexpr int TPI = LO == 2 ? 2 : 1;               // tiles per ring item
  constexpr int kItemBytes = TPI * kTileBytes;
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[4 * D * kItemBytes + 2048];
  float* red = reinterpret_cast<float*>(lds + 4 * D * kItemBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t item = (WPP == 1) ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (item >= a.n) return;
  const int64_t page = a.cand ? (int64_t)a.cand[item] : a.page0 + item;
  if (page < 0 || page_masked(a, page)) {  // page < 0: padding entry of a device-built candidate list
    if (lane == 0 && (WPP == 1 || wave == 0)) a.scores[item] = -INFINITY;
    return;
  }
  // packed layout: the table entry is loaded BESIDE the row count (both scalar loads before the one s_waitcnt; a packed index always carries n_rows)
  int64_t pk_row0 = 0;
  if constexpr (PK) pk_row0 = a.row_off[page];
  const int nr = PK ? a.n_rows[page] : (a.n_rows ? a.n_rows[page] : a.stride);
  const int ntiles = (nr + kTileRows - 1) / kTileRows;
  const bool clamp = (a.pad_items ? a.pad_items[item] : a.pad_to) > nr;

  f32x4 mx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mx[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

  // tile ownership: interleaved (wave w takes tiles w, w+4, ...: the workgroup reads 16 KiB bursts) or, CONTIG,
  // a contiguous quarter of the page per wave
  const int tq = (ntiles + WPP - 1) / WPP;
  const int t0 = (WPP == 1) ? 0 : (CONTIG ? wave * tq : wave);
  const int tstep = CONTIG ? 1 : WPP;
  const int ntw = CONTIG ? max(0, min(tq, ntiles - t0)) : (ntiles - t0 + WPP - 1) / WPP;
  const size_t pboff = PK ? (size_t)pk_row0 * kRowBytes : page_byte_off<false>(a, page);
  const char* pbase = a.slab + pboff;
  const char* plbase = LO == 2 ? a.slab_lo + pboff : nullptr;
  char* ring = lds + wave * (D * kItemBytes);

  // DMA source offsets: instruction i covers rows 4i..4i+3; lane -> row 4i+(lane>>4), position lane&15.
  int src_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = i * 4 + (lane >> 4);
    src_off[i] = w * kRowBytes + (((lane & 15) ^ w) << 4) - i * 1024;
  }
  // fragment read offsets: row r, logical chunk 4j+g at position (4j+g)^r
  int rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = r * kRowBytes + (((j * 4 + g) ^ r) << 4);

  // The DMA is issued from inline asm: hipcc's waitcnt pass otherwise drains vmcnt(0) in front of
  // every ds_read that may alias an in-flight LDS-DMA, which serialises the ring.  One statement per
  // tile: M0 = wave-uniform LDS slot address; the instruction offset (applied to BOTH the global and
  // the LDS address) walks the four 1 KiB pieces, so src_off[i] carries -i*1024 to compensate.
  // s_nop 4 covers SALU-write -> VMEM-read of the base SGPRs and the M0 write -> LDS-DMA hazard.
  auto issue_tile = [&](const char* tp, char* slot_ptr) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot_ptr);
    uint32_t keep;
    if (NT) {  // non-temporal: the page stream is read once; do not let it displace the query / metadata in L2
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
    } else {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %5\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %6\n\t"
          "global_load_lds_dwordx4 %2, %6 offset:1024\n\t"
          "global_load_lds_dwordx4 %3, %6 offset:2048\n\t"
          "global_load_lds_dwordx4 %4, %6 offset:3072\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src_off[0]), "v"(src_off[1]), "v"(src_off[2]), "v"(src_off[3]), "s"(slot), "s"(tpu)
          : "memory");
    }
  };
  auto issue = [&](int it) {
    const size_t toff = (size_t)(t0 + it * tstep) * kTileBytes;
    char* sp = ring + (it % D) * kItemBytes;
    issue_tile(pbase + toff, sp);
    if constexpr (LO == 2) issue_tile(plbase + toff, sp + kTileBytes);
  };

#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (i < ntw) issue(i);

  // Query fragments are loaded AFTER the prologue DMAs were issued and pinned (made "used") here, so
  // hipcc waits for them once, before the loop.  Left to itself it defers the wait to the first MFMA
  // inside the loop, where its counted vmcnt(7..0) ladder drains our DMA ring on every iteration.
  bf16x8 qa[MT][4];
  if (!STREAM_ONLY) {
    const uint16_t* qp = a.q;
    if (QITEM) qp += (size_t)(item / a.items_per_q) * (size_t)a.q_item_stride;  // this item's query (batched rerank)
    load_query<MT>(qp, r, g, qa);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qa[m][j]));
  }
  bf16x8 qal[LO ? MT : 1][4];
  if constexpr (LO > 0 && !STREAM_ONLY) {
    const uint16_t* qlp = a.qlo;
    if (QITEM) qlp += (size_t)(item / a.items_per_q) * (size_t)a.q_item_stride;
    load_query<MT>(qlp, r, g, qal);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(qal[m][j]));
  }

  for (int it = 0; it < ntw; ++it) {
    if (it + D - 1 < ntw) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: last reads of the slot being refilled
      issue(it + D - 1);
      wait_vmcnt<4 * TPI * (D - 1)>();
    } else {
      const int left = ntw - 1 - it;  // items still allowed in flight
      if (left >= 2) wait_vmcnt<8 * TPI>();
      else if (left == 1) wait_vmcnt<4 * TPI>();
      else wait_vmcnt<0>();
    }
    if (STREAM_ONLY) continue;
    const char* slot = ring + (it % D) * kItemBytes;
    bf16x8 b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(slot + rd_off[j]);
    const int t = t0 + it * tstep;
    const bool partial = (t + 1) * kTileRows > nr;
    if constexpr (LO == 0) {
      tile_mfma<MT>(qa, b, mx, partial, t * kTileRows + r < nr);
    } else if constexpr (LO == 1) {
      tile_mfma_lo<MT, 1>(qa, qal, b, b, mx, partial, t * kTileRows + r < nr);
    } else {
      bf16x8 bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const bf16x8*>(slot + kTileBytes + rd_off[j]);
      tile_mfma_lo<MT, 2>(qa, qal, b, bl, mx, partial, t * kTileRows + r < nr);
    }
  }
  if (STREAM_ONLY) {
    if (threadIdx.x == 0) a.scores[item] = 0.f;
    return;
  }

  if (WPP == 1) {
    float s = finish_wave<MT>(mx, clamp);
    if (lane == 0) a.scores[item] = s;
  } else {
    finish_block<MT>(mx, clamp, red, wave, lane, &a.scores[item]);
  }
}

// --------------------------------------------------------------------------------------------
// Removed in round 5 (they lost by measurement; records in profiles/r1, profiles/r2 and DESIGN.md "Kernel variants"):
//   1 / 4 / 5   direct-to-VGPR loads with four waves per page / non-temporal          6.2-6.4 / 5.9-6.0 TB/s
//   2 / 3       the LDS-DMA ring with default-policy (temporal) loads                    6.37 / 6.53 TB/s
//   8..12       ring-depth and tile-ownership probes of the default (D = 2, 3, 6, 8; contiguous quarters): within 1 % or slower
//   14          persistent workgroups with one continuous DMA stream across pages        6.86 TB/s against 7.33
// What stays: 6 (default from 512 rows per page), 7 (default below), 0 (direct loads, one wave per page: the independent
// cross-check of the ring), 13 (the default's transport without arithmetic: MV_CAL_READ_LDSDMA).

template <int MT, int LO, bool PK>
int launch_mt_pk(const KArgs& k0, int variant, hipStream_t s) {
  if (k0.n <= 0) return MV_OK;
  dim3 block(256);
  // a launch's work-item count must stay below 2^32: at most 2^22 pages per launch (x256 or x64 threads)
  constexpr int64_t kChunk = (int64_t)1 << 22;
  for (int64_t off = 0; off < k0.n; off += kChunk) {
    KArgs k = k0;
    k.n = std::min(kChunk, k0.n - off);
    k.scores = k0.scores + off;
    if (k0.cand) k.cand = k0.cand + off;
    else k.page0 = k0.page0 + off;
    if (k0.pad_items) k.pad_items = k0.pad_items + off;
    const int64_t n = k.n;
    if (k.items_per_q > 0) {  // per-item queries: the two default forms only, one launch
      if (k0.n > kChunk || (variant != 6 && variant != 7)) { set_error("per-item queries: variant %d / %lld items not supported", variant, (long long)k0.n); return MV_ERR_INVALID; }
      if (variant == 6) hipLaunchKernelGGL((maxsim_ldsdma_kernel<MT, 4, 4, true, false, false, true, LO, PK>), dim3((unsigned)n), block, 0, s, k);
      else hipLaunchKernelGGL((maxsim_ldsdma_kernel<MT, 1, 4, true, false, false, true, LO, PK>), dim3((unsigned)((n + 3) / 4)), block, 0, s, k);
      continue;
    }
    switch (variant) {
      case 0: hipLaunchKernelGGL((maxsim_direct_kernel<MT, 1, false, LO, PK>), dim3((unsigned)((n + 3) / 4)), block, 0, s, k); break;
      case 6: hipLaunchKernelGGL((maxsim_ldsdma_kernel<MT, 4, 4, true, false, false, false, LO, PK>), dim3((unsigned)n), block, 0, s, k); break;
      case 7: hipLaunchKernelGGL((maxsim_ldsdma_kernel<MT, 1, 4, true, false, false, false, LO, PK>), dim3((unsigned)((n + 3) / 4)), block, 0, s, k); break;
      case 13:
        if constexpr (LO == 0 && !PK) { hipLaunchKernelGGL((maxsim_ldsdma_kernel<MT, 4, 4, true, false, true>), dim3((unsigned)n), block, 0, s, k); break; }
        [[fallthrough]];
      default: set_error("unknown maxsim variant %d%s", variant, LO ? " (split-bf16 operands: 0, 6, 7)" : ""); return MV_ERR_INVALID;
    }
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

template <int MT, int LO>
int launch_mt(const KArgs& k, int variant, hipStream_t s) {
  return k.row_off ? launch_mt_pk<MT, LO, true>(k, variant, s) : launch_mt_pk<MT, LO, false>(k, variant, s);
}

template <int LO>
int launch_lo(const KArgs& k, int q_tiles, int variant, hipStream_t s) {
  switch (q_tiles) {
    case 1: return launch_mt<1, LO>(k, variant, s);
    case 2: return launch_mt<2, LO>(k, variant, s);
    case 3: return launch_mt<3, LO>(k, variant, s);
    default: return launch_mt<4, LO>(k, variant, s);
  }
}

}  // namespace

// Measured on MI355X (profiles/r1/variants_r1*.json, 100-200k pages x 1024 patches, Q=32): non-temporal LDS-DMA ring
// with four waves per page 7.34 TB/s > the same with one wave per page 7.08 > default-policy LDS-DMA 6.53 / 6.37 >
// direct loads 6.2-6.4 TB/s > non-temporal direct 5.9-6.0 TB/s.  Short pages cannot feed four waves: at 256 rows (64 KiB)
// the four-wave form drops to 5.84 TB/s against 6.92 for one wave per page, at 512 rows it leads 7.26 to 6.88
// (profiles/r1/float_scan_vs_page_size.json), so pages below 512 rows take the wave-per-page form.
int maxsim_default_variant(int stride_rows) { return stride_rows >= 512 ? 6 : 7; }

const char* maxsim_variant_name(int v) {
  switch (v) {
    case 0: return "direct_wpp1";
    case 6: return "ldsdma_wpp4_d4_nt";
    case 7: return "ldsdma_wpp1_d4_nt";
    case 13: return "ldsdma_wpp4_d4_nt_stream_only";
    default: return "?";
  }
}

int launch_maxsim_bf16(const MaxsimArgs& a, int variant, hipStream_t s) {
  if (variant < 0) variant = maxsim_default_variant(a.stride);
  if (a.q_tiles < 1 || a.q_tiles > kMaxQTiles) {
    set_error("q_tiles=%d out of range (1..%d)", a.q_tiles, kMaxQTiles);
    return MV_ERR_INVALID;
  }
  if (a.n > 0x7fffffffLL) {
    set_error("too many work items for one launch: %lld", (long long)a.n);
    return MV_ERR_INVALID;
  }
  KArgs k{reinterpret_cast<const char*>(a.slab), a.n_rows, a.doc_ord, a.allow, a.n_allow_bits, a.cand, a.q, a.scores,
          a.n, a.stride, a.pad_to, 0, a.pad_items, a.items_per_query, a.q_item_stride, a.qlo, reinterpret_cast<const char*>(a.slab_lo), a.row_off};
  if (a.items_per_query < 0 || (a.items_per_query > 0 && !a.cand)) { set_error("items_per_query needs a candidate list"); return MV_ERR_INVALID; }
  if (a.row_off && !a.n_rows) { set_error("a row-offset table (packed layout / re-placed exact tier) needs the per-page row counts"); return MV_ERR_INVALID; }
  if (a.slab_lo || a.qlo) {  // split-bf16 operands: the lo fragments double the query registers -- 64 query rows per pass
    if (!a.qlo) { set_error("the lo slab needs the query's lo half (zeros for a bf16 query)"); return MV_ERR_INVALID; }
    if (a.q_tiles > kMaxQTilesLo) { set_error("q_tiles=%d out of range (1..%d with split-bf16 operands)", a.q_tiles, kMaxQTilesLo); return MV_ERR_INVALID; }
    return a.slab_lo ? launch_lo<2>(k, a.q_tiles, variant, s) : launch_lo<1>(k, a.q_tiles, variant, s);
  }
  switch (a.q_tiles) {
    case 1: return launch_mt<1, 0>(k, variant, s);
    case 2: return launch_mt<2, 0>(k, variant, s);
    case 3: return launch_mt<3, 0>(k, variant, s);
    case 4: return launch_mt<4, 0>(k, variant, s);
    case 5: return launch_mt<5, 0>(k, variant, s);
    case 6: return launch_mt<6, 0>(k, variant, s);
    case 7: return launch_mt<7, 0>(k, variant, s);
    default: return launch_mt<8, 0>(k, variant, s);
  }
}

}  // namespace mv
