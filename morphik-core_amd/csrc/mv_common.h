// mv_common.h -- shared declarations of libmvmaxsim's translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "mvmaxsim.h"

namespace mv {

constexpr int kDim = 128;            // embedding width (ColPali/ColQwen head)
constexpr int kRowBytes = kDim * 2;  // one bf16 patch row
constexpr int kTileRows = 16;        // patch rows per MFMA tile (N of mfma_f32_16x16x32_bf16)
constexpr int kTileBytes = kTileRows * kRowBytes;  // 4 KiB
constexpr int kSignBytes = kDim / 8;               // 16 B packed sign row

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define MV_HIP(expr)                                                         \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::mv::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// ---------------------------------------------------------------- float MaxSim (mv_maxsim.hip)
struct MaxsimArgs {
  const uint16_t* slab;     // [pages][stride][128] bf16
  const int32_t* n_rows;    // valid rows per page (nullable -> stride)
  const int32_t* doc_ord;   // doc ordinal per page, <0 = tombstoned (nullable)
  const uint32_t* allow;    // bitmap over doc ordinals (nullable -> all)
  int64_t n_allow_bits;
  const int32_t* cand;      // candidate page list (nullable -> pages 0..n-1)
  const uint16_t* q;        // [q_rows_padded][128] bf16, zero padded to a multiple of 16 rows
  float* scores;            // [n] one score per work item; -inf for masked pages
  int64_t n;                // work items
  int32_t stride;           // rows per page slot
  int32_t q_tiles;          // q_rows_padded / 16 (1..4)
  int32_t pad_to;           // zero-padding clamp: pages with n_rows < pad_to clamp each token max at 0
  const int32_t* pad_items; // per-item pad_to (device; the reference pads every rerank batch of 128 on its own); null -> pad_to
  int32_t items_per_query;  // > 0: the candidate lists of a batch of queries in one launch -- item i is scored against the
  int32_t q_item_stride;    //      query at q + (i / items_per_query) * q_item_stride (bf16 elements); default variants only
  // split-bf16 operands (fp32-faithful scores on the bf16 matrix pipe; the reference scores fp32 pages with fp32 queries,
  // fast_multivector_store.py:553-555, :736, :774): x = hi + lo, hi = bf16(x), lo = bf16(x - hi), |x - hi - lo| <= 2^-18 |x|
  const uint16_t* qlo;      // nullable: the lo half of the query rows (layout of q, q_item_stride applies): S = qhi.p + qlo.p
  const uint16_t* slab_lo;  // nullable: the lo half of the page rows (layout of slab; needs qlo -- zeros for a bf16 query):
                            //           S = qhi.phi + qlo.phi + qhi.plo (the dropped qlo.plo term is <= 2^-18 of the product)
  const int64_t* row_off;   // nullable: MV_LAYOUT_PACKED -- first slab row of every page (a multiple of 16); null: page * stride
};
// variant: -1 default; see DESIGN.md "Kernel variants".
int launch_maxsim_bf16(const MaxsimArgs& a, int variant, hipStream_t s);
const char* maxsim_variant_name(int variant);
int maxsim_default_variant(int stride_rows);

// ---------------------------------------------------------------- batched float MaxSim (mv_batch.hip)
struct BatchArgs {
  const uint16_t* slab;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const uint16_t* q;        // [512][128] bf16: query b occupies rows [b*rows_per_query, (b+1)*rows_per_query), zero padded
  float* scores;            // [n_queries][score_stride]
  int64_t n;                // pages 0..n-1
  int64_t score_stride;
  int32_t stride;
  int32_t n_queries;
  int32_t rows_per_query;   // multiple of 16; n_queries * rows_per_query <= 512
  int64_t allow_stride_bits;  // 0 = `allow` is shared; else query b uses allow + b * allow_stride_bits / 32 (n_allow_bits each)
  int32_t variant;          // 0 = 16x16x32 MFMA, 4 waves / workgroup; 1 = 32x32x16 MFMA, 8 waves / workgroup (> 128 rows)
  const int64_t* row_off;   // nullable: packed layout (MaxsimArgs::row_off)
};
int launch_maxsim_batch(const BatchArgs& a, hipStream_t s);

// ---------------------------------------------------------------- selection (mv_topk.hip)
// keys: order-preserving 64-bit (score desc, local index asc). ws must hold topk_ws_bytes(n,k).
size_t topk_ws_bytes(int64_t n, int32_t k);
constexpr int kTopkMaxDeviceK = 1024;
// d_ids_map: optional int32 map from work index to local page id (candidate lists); id_base added on output.
// hist0_done: the first radix histogram (key bits [31:21]) of THESE scores was already accumulated into
// topk_radix_hist0(ws) by the kernel that produced them (the FDE scan does, when topk_uses_radix(n, k)).
int launch_topk(const float* d_scores, int64_t n, int32_t k, const int32_t* d_ids_map, int64_t id_base, void* ws,
                float* d_out_scores, int64_t* d_out_ids, hipStream_t s, bool hist0_done = false);
// nb selections in one chain of launches (grid.y = selection): query b reads d_scores + b*score_stride, maps through
// d_ids_map + b*map_stride (when given), uses the workspace at ws + b*ws_stride BYTES (each topk_ws_bytes(n, k), zeroed
// once) and writes row b of the [nb][out_stride] results.
int launch_topk_batch(const float* d_scores, int64_t score_stride, int64_t n, int32_t k, const int32_t* d_ids_map, int64_t map_stride,
                      int64_t id_base, void* ws, size_t ws_stride, float* d_out_scores, int64_t* d_out_ids, int64_t out_stride, int nb,
                      hipStream_t s, bool hist0_done = false);
bool topk_uses_radix(int64_t n, int32_t k);   // the selection of k of n takes the radix-threshold path
uint32_t* topk_radix_hist0(void* ws);          // its first histogram (2048 bins, zero between selections)
// order-preserving key of a score (larger score -> larger key); shared by the selection and the kernels that pre-bin
__device__ __forceinline__ uint32_t topk_ordered_u32(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// LDS histogram increment for one key per lane.  Scores of one scan crowd into a handful of bins of the FIRST pass (the top
// 11 key bits are sign + exponent + 2 mantissa bits: cosine-like FDE scores share them), and 64 lanes incrementing the
// same LDS word serialise (measured: the two histogram passes were 0.14 of the 0.26 ms a 32-request selection took).  So the
// wave first aggregates: up to two rounds of "the first active lane's bin -> ballot of the lanes holding it -> ONE add of
// the population count", then whatever is left (the spread-out tail, and every key of the second pass, whose bins are
// mantissa bits) goes through plain atomics.  Every lane of the wave must make the call (the ballots need all of them).
__device__ __forceinline__ void topk_hist_add_wave(uint32_t* h, uint32_t bin, bool valid) {
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const uint64_t active = __ballot(valid);
    if (!active) return;
    const int leader = __ffsll((unsigned long long)active) - 1;
    const uint32_t lb = (uint32_t)__shfl((int)bin, leader);
    const bool mine = valid && bin == lb;
    const uint64_t same = __ballot(mine);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[lb], (uint32_t)__popcll((unsigned long long)same));
    valid = valid && !mine;
  }
  if (valid) atomicAdd(&h[bin], 1u);
}

// s_stride / i_stride: distance between two ranks' lists in floats / int64s (0 = kk: two dense [world][kk] arrays)
int launch_merge_topk(const float* d_scores, const int64_t* d_ids, int32_t world, int32_t kk, int32_t k, float* d_out_scores,
                      int64_t* d_out_ids, hipStream_t s, int64_t s_stride = 0, int64_t i_stride = 0);

// ---------------------------------------------------------------- synthetic generator (mv_synth.hip)
int launch_synth_rows(uint16_t* d_out, uint64_t seed, uint64_t first_unit, int64_t n_units, int32_t n_rows,
                      int32_t stride_rows, hipStream_t s);
// ragged form: unit i gets its first d_n_rows[i] rows at d_base + (d_row_off ? d_row_off[i] : i * stride_rows) * 128, zero rows up to
// the end of its allotment (d_row_off[i + 1], or the stride slot)
int launch_synth_rows_ragged(uint16_t* d_base, uint64_t seed, uint64_t first_unit, int64_t n_units, const int32_t* d_n_rows,
                             const int64_t* d_row_off, int32_t stride_rows, hipStream_t s);

// ---------------------------------------------------------------- binary path (mv_binary.hip)
int launch_sign_pack_f32(const float* d_x, int64_t n_rows, int32_t d, uint8_t* d_out, hipStream_t s);
// pack bf16 slab rows [n_pages*stride][128] -> [n_pages*stride][16 B]
int launch_sign_pack_bf16_rows(const uint16_t* d_rows, int64_t n_rows, uint8_t* d_out, hipStream_t s);
struct BinaryArgs {
  const uint8_t* bits;     // [pages][stride][16]
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const uint8_t* qbits;    // [n_q][16]; MFMA variant: buffer padded to a multiple of 16 rows (zero-filled by the launch)
  float* qpop_rw;          // [n_q padded to 16] workspace: popc per query row (MFMA variant)
  const float* qpop;       // same buffer, read side
  float* scores;           // Q - sum_q min_d hamming / 128, exact in fp32; -inf masked
  int64_t n;
  int32_t stride;
  int32_t n_q;
  const int32_t* cand;     // optional candidate list (item i = page cand[i], scores[i]); n = number of candidates
  const int64_t* row_off;  // nullable: packed layout (MaxsimArgs::row_off; the sign slab shares the row numbering)
};
// variant: 0 = popcount (VALU), 1 = FP4 MFMA (16 VALU ops/tile), 2 = FP4 MFMA with in-place bit operands (8 VALU
// ops/tile, 8-slot ring), 3 / 4 = variant 2 with a 16- / 4-slot ring (4 = default, -1), 5 = persistent waves with one DMA
// stream across pages (fixed-size unfiltered corpora; measured 6.1 vs 6.7 TB/s for 4)
int launch_maxsim_binary(const BinaryArgs& a, int variant, hipStream_t s);
int launch_hamming_batch(const uint8_t* d_q, const uint8_t* d_c, int64_t n, int32_t n_bytes, int32_t* d_out,
                         hipStream_t s);

// ---------------------------------------------------------------- FDE (mv_fde.hip)
struct FdeTables {           // device copies of the projection tables of one mv_fde_config
  mv_fde_config cfg{};
  float* G = nullptr;        // [rep][dim][nsh]
  int32_t* H = nullptr;      // [rep][dim]
  float* S = nullptr;        // [rep][dim]
  int64_t out_dim = 0;
  // partition ids between the two passes of the document encode (stride * R bytes per page of a chunk); grown on demand by the
  // writer that encodes (writers are serialised), freed with the tables
  mutable uint8_t* scratch = nullptr;
  mutable size_t scratch_bytes = 0;
};
void fde_host_tables(const mv_fde_config& c, float* G, int32_t* H, float* S);
int fde_tables_create(const mv_fde_config& c, FdeTables* t);
void fde_tables_destroy(FdeTables* t);
// Encode pages: x rows are fp32 [sum rows][128] (x_f32) or bf16 slab pages (x_bf16 with stride);
// out_f32 (nullable) [n_pages][out_dim] fp32, out_bf16 (nullable) [n_pages][out_dim] bf16,
// out_inv_norm (nullable) [n_pages] = 1/|bf16(fde)|.
struct FdeEncodeArgs {
  const float* x_f32;        // ragged fp32 rows (row_offsets gives the start row of each page) or null
  const uint16_t* x_bf16;    // fixed-stride bf16 slab pages or null
  const int64_t* row_offsets;// [n_pages+1] for x_f32; null with n_pages == 1: the one page has `stride` rows
  const int32_t* n_rows;     // per page rows (for x_bf16; nullable -> stride)
  int32_t stride;
  int64_t n_pages;
  int32_t is_query;
  float* out_f32;
  uint16_t* out_bf16;
  float* out_inv_norm;
  int32_t variant;           // 0 = scalar kernel (LDS atomics), 1 = f32-MFMA kernel, 2 = query latency kernel, 3 = one-pass document kernel (bf16 AMS,
                             // LDS-atomic bucket sums), 4 = two-pass document form (hash pass + projection pass with one-hot MFMA bucket sums)
  const int64_t* x_row_off;  // nullable, with x_bf16: packed layout -- page i's rows start at x_bf16 + x_row_off[i] * 128 (x_bf16 then is
                             // the slab BASE the offsets count from, x_row_off already points at the first page of the call)
};
int launch_fde_encode(const FdeTables& t, const FdeEncodeArgs& a, hipStream_t s);
// caller-supplied document FDE vectors (fp32 [n][out_dim], device memory) -> the slab's bf16 rows + 1 / |d| of the ROUNDED rows (the encode kernels' rule)
int launch_fde_import(const float* d_src, int64_t n, int64_t out_dim, uint16_t* out_bf16, float* out_inv_norm, hipStream_t s);
struct FdeScanArgs {
  const uint16_t* fde;      // [pages][out_dim] bf16
  const float* inv_norm;    // [pages] (nullable -> dot)
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const float* q;           // [out_dim] fp32
  float* scores;            // [n]
  int64_t n;
  int64_t out_dim;
  uint32_t* hist0;          // nullable: 2048-bin histogram of the scores' key bits [31:21], accumulated by the scan itself
                            // (saves the selection's first pass); default variant at out_dim 10240 / 5120 only
};
// variant: -1 / 5 = row quarters through the nt LDS-DMA ring (default), 0 = query in registers, one wave per row, nt loads (cross-check)
int launch_fde_scan(const FdeScanArgs& a, int variant, hipStream_t s);
// The same stage on the e4m3 copy of the slab (mv_fde8.hip): score = (sum_i q_i * decode(code_i)) * scale[page] (* inv_norm[page]).
struct FdeScan8Args {
  const uint8_t* fde8;      // [pages][out_dim] e4m3fn
  const float* scale;       // [pages]
  const float* inv_norm;    // [pages] (nullable -> dot)
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const float* q;           // [out_dim] fp32
  float* scores;            // [n]
  int64_t n;
  int64_t out_dim;
};
bool fde_scan8_supported(int64_t out_dim);
int launch_fde_scan8(const FdeScan8Args& a, hipStream_t s);
// ... and on the FP4 (e2m1) copy (mv_fde4.hip): a.fde8 = the codes [pages][out_dim / 2], a.scale = the per-page factor (scale, or scale / |d|)
bool fde_scan4_supported(int64_t out_dim);
int launch_fde_scan4(const FdeScan8Args& a, hipStream_t s);
int launch_fde_quantize_fp4(const uint16_t* d_fde_rows, int64_t out_dim, int64_t n, uint8_t* d_codes, float* d_scale, const float* d_inv_norm, float* d_cfac,
                            hipStream_t s);
bool fde_scan_prebins(int variant, int64_t out_dim);  // the form launch_fde_scan would run fills FdeScanArgs::hist0
constexpr int kFdeBatchMaxQueries = 32;
// Up to 32 queries per pass over the FDE slab (bf16 MFMA, fp32 queries as bf16 hi + lo): scores[q][page] at
// scores + q*score_stride.  `image` is a scratch buffer of fde_scan_batch_image_bytes(out_dim).
struct FdeScanBatchArgs {
  const uint16_t* fde;
  const float* inv_norm;      // nullable -> dot
  const int32_t* doc_ord;     // nullable -> no masks
  const uint32_t* allow;      // nullable
  int64_t n_allow_bits;
  int64_t allow_stride_bits;  // 0: one bitmap for every query
  const float* q;             // [n_queries][out_dim] fp32 query FDEs
  uint16_t* image;
  float* scores;
  int64_t score_stride;
  int64_t n;
  int64_t out_dim;
  int32_t n_queries;
  int32_t hi_only;            // 1: bf16 query FDE (one MFMA per fragment, half the query traffic; coarse scores within ~2e-3)
  int32_t single_tile;        // 1: one page tile per query fragment (the first form; default: tiles in pairs)
  // nullable: query b's first selection histogram (topk_radix_hist0 of ITS workspace) at hist0 + b * hist0_stride_bytes, zero on
  // entry; accumulated by the finish pass (which touches every score anyway) so the selection starts at its second pass.
  // Only honoured when a finish pass runs (inv_norm or doc_ord given): fde_scan_batch_prebins().
  uint32_t* hist0;
  int64_t hist0_stride_bytes;
  int32_t separate_finish;    // 1: keep the finish a pass of its own even where the scan kernel could apply it (MV_OPT_FDE_BATCH_VARIANT 5)
  // MV_WITH_FDE_E4M3 (nullable): the slab's e4m3 copy [n][out_dim], its per-page factor (scale, or scale / |d| under the cosine rule) and a
  // 32-float scratch for the queries' scales.  Given (and the default form selected), the pass reads THIS slab: half the bytes.
  const uint8_t* fde8;
  const float* fde8_fac;
  float* qfac;
  int32_t copy_fp4;           // 1: fde8 / fde8_fac are the FP4 copy (MV_WITH_FDE_FP4: [n][out_dim / 2] e2m1 codes) -- both MFMA operands FP4, the queries as two e2m1 terms
};
bool fde_scan_batch8_supported(int64_t out_dim);
bool fde_scan_batch4_supported(int64_t out_dim);
// the default (paired-tile) kernel applies the cosine rule and the tombstones where it writes a tile's scores: no finish pass
inline bool fde_scan_batch_fuses_finish(const FdeScanBatchArgs& a) {
  return a.inv_norm != nullptr && !a.single_tile && !a.separate_finish;
}
// the pass reads the slab's e4m3 copy (its kernel always writes finished scores: nothing is binned on the way)
inline bool fde_scan_batch_uses_e4m3(const FdeScanBatchArgs& a) {
  return a.fde8 != nullptr && !a.single_tile && !a.separate_finish && (a.copy_fp4 ? fde_scan_batch4_supported(a.out_dim) : fde_scan_batch8_supported(a.out_dim));
}
inline bool fde_scan_batch_prebins(const FdeScanBatchArgs& a) {
  return a.hist0 != nullptr && (a.inv_norm != nullptr || a.doc_ord != nullptr) && !fde_scan_batch_fuses_finish(a) && !fde_scan_batch_uses_e4m3(a);
}
bool fde_scan_batch_supported(int64_t out_dim);
size_t fde_scan_batch_image_bytes(int64_t out_dim);
int launch_fde_scan_batch(const FdeScanBatchArgs& a, hipStream_t s);

// ---------------------------------------------------------------- fp8 path (mv_fp8.hip)
// quantise fixed-stride bf16 pages -> e4m3 codes + one power-of-two scale per page (inv_scale = 2^-e)
// d_row_off (nullable, packed layout): page i's rows start d_row_off[i] rows into BOTH d_src_pages and d_dst (the two slabs share the
// row numbering; the pointers then are the bases the offsets count from)
int launch_quantize_pages_fp8(const uint16_t* d_src_pages, const int32_t* d_n_rows, int32_t stride, int64_t n_pages,
                              uint8_t* d_dst, float* d_inv_scale, hipStream_t s, const int64_t* d_row_off = nullptr);
// fp32 query rows -> two-term e4m3 split (hi, lo*16) + 2^-s per row; buffers padded to a multiple of 16 rows
int launch_fp8_query_prep(const float* d_q_f32, int n_q, uint8_t* d_hi, uint8_t* d_lo, float* d_fac, hipStream_t s);
struct Fp8ScanArgs {
  const uint8_t* slab;      // [pages][stride][128] e4m3fn
  const float* inv_scale;   // [pages]
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const int32_t* cand;
  const uint8_t* qhi;
  const uint8_t* qlo;
  const float* qfac;
  int32_t n_q;
  float* scores;
  int64_t n;
  int32_t stride;
  int32_t pad_to;
  const int32_t* pad_items;  // per-item pad_to (device); null -> pad_to
  int32_t items_per_query;   // > 0: rerank lists of a batch of queries in one launch -- item i uses the query whose rows start
                             // (i / items_per_query) * padded(n_q) rows into qhi / qlo / qfac; needs cand, n_q <= 64
  const int64_t* row_off;    // nullable: packed layout (MaxsimArgs::row_off)
};
int launch_maxsim_fp8(const Fp8ScanArgs& a, hipStream_t s);
// A batch of queries (n_queries * rows_per_query <= 512 rows, each query padded to rows_per_query with zero rows) against
// every page of the e4m3 slab in one pass: scores[q][page] at scores + q * score_stride.  qhi / qlo / qfac hold the
// group's rows back to back (launch_fp8_query_prep over all of them), zero rows up to 512.
struct Fp8BatchArgs {
  const uint8_t* slab;
  const float* inv_scale;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  int64_t allow_stride_bits;
  const uint8_t* qhi;
  const uint8_t* qlo;
  const float* qfac;
  float* scores;
  int64_t n;
  int64_t score_stride;
  int32_t stride;
  int32_t n_queries;
  int32_t rows_per_query;
  int32_t single_term;  // 1: hi term only (query rounded to e4m3: half the matrix work; coarse pass of a two-tier search)
  const int64_t* row_off;  // nullable: packed layout
};
int launch_maxsim_batch_fp8(const Fp8BatchArgs& a, hipStream_t s);

// ---------------------------------------------------------------- doc_ids filter compaction (mv_filter.hip)
size_t filter_ws_bytes(int64_t capacity);
int launch_filter_compact(const int32_t* d_doc_ord, const uint32_t* d_allow, int64_t n_allow_bits, int64_t n, int32_t* d_counts,
                          int32_t* d_cand, int64_t* out_n, hipStream_t s);

// ---------------------------------------------------------------- misc device helpers
int launch_f32_to_bf16(const float* d_in, uint16_t* d_out, int64_t n, hipStream_t s);
int launch_read_bw(const void* d_buf, int64_t bytes, float* d_sink, hipStream_t s);
int launch_read_bw_nt(const void* d_buf, int64_t bytes, float* d_sink, hipStream_t s);  // contiguous 16 KiB pieces, nt loads
int launch_stream_probe(const void* d_buf, int64_t bytes, int ct, int own, int sched, int blocks_per_cu, uint32_t* d_work, float* d_sink, hipStream_t s,
                        const float* d_q = nullptr, float* d_unit_out = nullptr);  // see mv_synth.hip
int launch_read_bw_strided(const void* d_buf, int64_t n_rows, int piece, float* d_sink, hipStream_t s);  // [rows][20 480 B], `piece` bytes per row and step
int launch_mfma_peak(int blocks, int iters, int shape, float* d_sink, hipStream_t s);   // per iteration per wave: shape 0 = 8 x 16x16x32, 1 = 4 x 32x32x16 bf16 MFMA
// scatter ragged bf16/f32 rows into the fixed-stride slab (zero-filling the tail of each page slot);
// d_nonfinite (nullable): set to 1 when a row holds a NaN / Inf (in its bf16 image);
// d_slab_lo_pages (nullable): the same slots of the lo slab receive bf16(x - bf16(x)) of fp32 input (zeros for bf16 input);
// d_dst_row_off (nullable, packed layout): page i goes to rows [d_dst_row_off[i], d_dst_row_off[i + 1]) of d_slab_pages / d_slab_lo_pages
int launch_scatter_rows(const void* d_src, int dtype, const int64_t* d_row_offsets, int64_t n_pages, int32_t stride,
                        uint16_t* d_slab_pages, hipStream_t s, int32_t* d_nonfinite = nullptr, uint16_t* d_slab_lo_pages = nullptr,
                        const int64_t* d_dst_row_off = nullptr);


// Units of a row scan that follow the slab's 256 KiB-aligned blocks: unit `blk` = the rows that START in the blk-th aligned
// block the slab touches; its `G` groups split them in halves.  Workgroup b runs on XCD b % 8, so with one block per workgroup every
// XCD's successive workgroups stay 2 MiB apart -- the float scan's 256 KiB pages have that shape by construction.
#if defined(__HIPCC__)
__device__ __forceinline__ void block_unit_rows(const void* slab, int64_t row_bytes, int64_t blk, int groups, int group, int64_t* first, int* count) {
  const int64_t off0 = (int64_t)(reinterpret_cast<uintptr_t>(slab) & 262143u);
  const int64_t lo_b = blk * 262144 - off0, hi_b = lo_b + 262144;
  const int64_t r0 = lo_b <= 0 ? 0 : (lo_b + row_bytes - 1) / row_bytes;
  const int n = (int)((hi_b + row_bytes - 1) / row_bytes - r0);
  const int per = (n + groups - 1) / groups;
  *first = r0 + (int64_t)group * per;
  *count = max(0, min(per, n - group * per));
}
#endif
inline int64_t block_unit_count(const void* slab, int64_t row_bytes, int64_t n_rows) {
  const int64_t span = (int64_t)(reinterpret_cast<uintptr_t>(slab) & 262143u) + n_rows * row_bytes;
  return (span + 262143) / 262144;
}

}  // namespace mv
