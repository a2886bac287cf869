// mv_index_priv.h -- the mv_index object shared by mv_api.hip (single-index C ABI) and mv_comm.hip
// (multi-shard communicator).  Not part of the public ABI.
//
// Concurrency model ("append-only slab + atomic published length"):
//   * `size` is the PUBLISHED page count.  A query snapshots it once (acquire) and never looks past it.
//   * Writers (mv_index_add*, fill_synthetic, remove_*) serialise on `w_mu`, work on `w_stream` with their own
//     staging buffers, fill slab slots [size, size+n) that no query can see yet, and publish the new size
//     (release) only after the device work has completed.  They never take `q_mu`: an ingest does not block queries.
//   * Queries serialise on `q_mu` (there is ONE per-query workspace and one GPU to saturate).
//   * Operations that move or rewrite PUBLISHED pages (compact, replace_page, write_rows, save) take both, in the
//     order w_mu -> q_mu.
//   * Tombstoning (remove_*) rewrites 4-byte doc ordinals of published pages in place; a scan in flight sees the
//     old or the new value of each, i.e. it is ordered before or after the removal page by page.
#pragma once
#include <atomic>
#include <mutex>
#include <vector>

#include "mv_common.h"

constexpr int kMaxQRowsPerPass = 128;  // 8 MFMA row tiles held in VGPRs
constexpr int kMaxCand = 65536;
constexpr int kBatchQRows = 1024;       // bf16 query block of the batched paths (float batch: groups of 512 rows; FDE batch: 32 x 32)
constexpr int kRerankBatch = 128;      // score_multi_vector scores passages in batches of 128, each padded on its own

struct mv_index {
  mv_config cfg{};
  hipStream_t stream = nullptr;    // query stream
  hipStream_t w_stream = nullptr;  // writer stream (ingest runs beside the scans)
  // slabs
  uint16_t* slab = nullptr;
  uint16_t* slab_lo = nullptr;   // MV_WITH_FLOAT_LO: bf16(x - bf16(x)) of every slab element (zeros for pages ingested as bf16) -- with `slab` the
                                 // split-bf16 image of the reference's fp32 pages (fast_multivector_store.py:676-681): the same 4 bytes per element
  uint8_t* bits = nullptr;
  uint16_t* fde = nullptr;
  float* fde_inv_norm = nullptr;
  uint16_t* h_exact = nullptr;   // MV_WITH_HOST_EXACT: pinned host bf16 rows of pages [x_split, capacity) ([..][stride][128] + 32 KiB), the exact tier
  uint16_t* d_exact = nullptr;   // the same memory through the device's address space (hipHostGetDevicePointer)
  uint16_t* slab_x = nullptr;    // MV_WITH_EXACT_SPLIT: the exact rows of pages [0, x_split) in HBM (what was free after the other slabs)
  int64_t x_split = 0;           // pages of the exact tier that live in HBM (0: the whole tier is host memory)
  // PLACEMENT of a split tier (mv_index_exact_tier_rebalance): slot s of the tier is HBM for s < x_split, pinned host memory above.
  // Page p's exact rows live in slot x_loc[p] -- the identity until the first rebalance swaps HOT host-resident pages (counted in
  // d_xhits by every rerank's list split) with cold HBM-resident ones.  Only published pages are ever swapped, so slot p of a page
  // appended later is always free.  d_xoff[p] = x_loc[p] * stride_rows: the row-offset table the rerank kernels take in place of
  // page * stride_rows (the packed layout's mechanism, mv_maxsim.hip KArgs::row_off).
  std::vector<int32_t> x_loc, x_page_at;  // [capacity] page -> slot, slot -> page (empty until the first rebalance: identity)
  int32_t* d_xloc = nullptr;              // device copy of x_loc (split_cand_kernel); null = identity
  int64_t* d_xoff = nullptr;              // [capacity] rows; null = identity
  uint32_t* d_xhits = nullptr;            // [capacity] times a page's exact rows were read by a rerank since the last rebalance (split tiers only)
  int32_t* d_xcand = nullptr;    // [2][kMaxCand] a rerank list split by tier part (lazily allocated)
  float* d_xscores = nullptr;    // [kMaxCand] scores of the second part
  uint8_t* slab8 = nullptr;      // e4m3 page slab [capacity][stride][128]
  float* inv_scale8 = nullptr;   // [capacity] 2^-e per page
  int32_t* d_n_rows = nullptr;
  int32_t* d_doc_ord = nullptr;
  std::vector<int32_t> h_n_rows, h_doc_ord;  // sized to capacity at create: never reallocated
  // MV_LAYOUT_PACKED: pages lie back to back in whole 16-row tiles.  row_off[p] = first slab row of page p (a multiple of 16),
  // row_off[p + 1] - row_off[p] = the rows its slot holds, row_off[size] = rows in use.  The row-indexed slabs (slab, slab_lo,
  // slab8, bits) share the numbering.  Entries of published pages are immutable (compaction rewrites them under both locks).
  bool packed = false;
  int64_t cap_rows = 0;          // rows the row-indexed slabs hold (capacity_pages * stride_rows in the fixed layout)
  int64_t* d_row_off = nullptr;  // [capacity + 1] (packed only; nullptr in the fixed layout: kernels then use page * stride)
  std::vector<int64_t> h_row_off;
  std::atomic<int64_t> size{0};              // published pages
  std::atomic<bool> ragged{false};           // some page has n_rows != stride
  std::atomic<bool> tombstones{false};       // some page is deleted
  std::atomic<int32_t> max_doc_ord{-1};      // largest document ordinal seen (filter selectivity estimate)
  mv::FdeTables fde_t;
  // per-query workspace (q_mu)
  float* d_scores = nullptr;   // [capacity]
  float* d_scores2 = nullptr;  // [capacity] (second accumulator for > 64 query rows)
  void* d_topk_ws = nullptr;
  size_t topk_ws_bytes = 0;
  bool sync_call = false;      // the running query drains the stream before it returns (no staging-buffer event needed)
  uint16_t* d_q = nullptr;     // bf16 query, padded
  uint16_t* d_qlo = nullptr;   // lo half of an fp32 query (bf16(q - bf16(q))), padded like d_q; valid when q_lo_valid
  bool q_has_lo = false;       // the uploaded query is not bf16-representable (some lo element is non-zero)
  bool q_lo_valid = false;     // d_qlo holds the uploaded query's lo rows (zeros for a bf16 query): q_has_lo, or the index keeps a lo slab
  uint16_t* d_bqlo = nullptr;  // the same for the batch block d_bq (lazily allocated)
  bool bq_has_lo = false, bq_lo_valid = false;
  uint16_t* h_qlo = nullptr;   // pinned staging of d_qlo
  float* d_qf32 = nullptr;     // fp32 query rows (FDE encode input)
  uint8_t* d_qbits = nullptr;
  float* d_qpop = nullptr;     // popc per query row (binary MFMA scan)
  uint8_t* d_q8hi = nullptr;   // e4m3 query rows, two-term split (fp8 scan)
  uint8_t* d_q8lo = nullptr;
  float* d_q8fac = nullptr;    // 2^-s per query row
  uint16_t* d_bq = nullptr;    // [kBatchQRows][128] bf16 query block of the batched scans
  float* d_bscores = nullptr;  // [32][bscore_stride] per-query score vectors of the batched scan (lazily allocated)
  std::vector<void*> parked;       // (diagnostic option 1000) workspaces set aside instead of freed, so that the next ones get other memory
  int64_t bscore_stride = 0;   // elements between two requests' score vectors: capacity_pages (+ MV_BSCORE_STRIDE_PAD from the environment)
  // batched FDE pipeline (mv_query_topk_batch in the FDE modes; lazily allocated, up to 32 queries per slab pass)
  float* d_bqf32 = nullptr;        // [kBatchQRows][128] fp32 query rows of the group, [query][rows padded to 16][128], zero rows behind each query
  float* d_bqfde = nullptr;        // [32][out_dim] fp32 query FDEs
  uint16_t* d_bqimage = nullptr;   // fragment-ordered bf16 hi/lo image of the query FDEs
  uint8_t* d_bq8hi = nullptr;      // [kBatchQRows][128] e4m3 two-term split of the group's query rows (fp8 rerank)
  uint8_t* d_bq8lo = nullptr;
  float* d_bq8fac = nullptr;       // [kBatchQRows]
  void* d_btopk_ws = nullptr;      // 32 selection workspaces of topk_ws_bytes each
  float* d_bsel_s = nullptr;       // [32][n_coarse] coarse top-n per query
  int64_t* d_bsel_id = nullptr;
  int32_t* d_bcand = nullptr;      // [32][n_coarse] rerank lists + per-batch pad lengths
  int32_t* d_bcand_pads = nullptr;
  float* d_bcand_scores = nullptr;
  float* d_bout_s = nullptr;       // [32][k] results
  int64_t* d_bout_id = nullptr;
  float* h_bout_s = nullptr;       // pinned
  int64_t* h_bout_id = nullptr;
  int32_t* h_bcand = nullptr;      // pinned: candidate lists read back for the accounting
  int32_t* d_fcand = nullptr;  // [capacity] pages a selective doc filter lets through (lazily allocated)
  int32_t* d_fcounts = nullptr;
  int filter_compact_pct = 25; // compact when the filter allows less than this share of the documents (0 = never)
  float* d_qfde = nullptr;
  int64_t* d_qoff = nullptr;   // [2] row offsets for the query "page"
  uint32_t* d_allow = nullptr;
  int64_t allow_cap_words = 0;
  float* d_out_s = nullptr;    // [kTopkMaxDeviceK]
  int64_t* d_out_id = nullptr;
  int32_t* d_cand = nullptr;   // [kMaxCand] candidate pages of a rerank (-1 = padding entry)
  int32_t* d_cand_pads = nullptr;  // [kMaxCand] pad_to of each candidate (its batch-of-128's longest page)
  float* d_cand_scores = nullptr;
  mv_cand_rec* d_recs = nullptr;   // [kTopkMaxDeviceK] coarse candidates of the sharded two-stage pipeline
  int64_t* d_sel_pos = nullptr;    // [kTopkMaxDeviceK] positions of the global coarse top-n inside the gathered records
  float* d_gscores = nullptr;      // gathered coarse scores (two-stage rerank), grown on demand
  int64_t gscores_cap = 0;
  int q_rows_cap = 0;
  // pinned host staging of the query (fp32 + padded bf16) and of the k results: async copies, no sync on the way in
  float* h_qf32 = nullptr;
  uint16_t* h_qbf16 = nullptr;
  float* h_out_s = nullptr;
  int64_t* h_out_id = nullptr;
  float* hd_out_s = nullptr;    // the same pinned buffers as the DEVICE addresses them (the selection writes host results in place)
  int64_t* hd_out_id = nullptr;
  int32_t* h_cand = nullptr;       // [kTopkMaxDeviceK] pinned: candidate ids read back for the accounting only
  hipEvent_t ev_stage = nullptr;  // recorded behind the H2D copies of the staging buffers
  hipEvent_t ev[6] = {};
  hipEvent_t ev_st[3] = {};       // FDE stage boundaries: after the query encode, after the coarse scan, after the selection
  // second set of the timing events: an enqueue-only query with deferred timings (mv_query_topk_device_async) swaps the sets before
  // it records, so the PREVIOUS deferred query's events survive until mv_query_stats_finish reads them -- one query may be outstanding
  hipEvent_t ev_alt[3] = {};
  hipEvent_t ev_st_alt[3] = {};
  int ev_parity = 0;
  std::mutex q_mu;  // queries + the per-query workspace
  std::mutex w_mu;  // writers
  // writer-side staging (w_mu): grown on demand, never freed while a scan may run (hipFree synchronises the device)
  void* w_stage = nullptr;
  size_t w_stage_bytes = 0;
  void* w_aux = nullptr;       // row offsets / sign rows of the batch being added
  size_t w_aux_bytes = 0;
  void* w_tmp = nullptr;       // fixed-stride bf16 image of the batch when the index keeps no float slab
  size_t w_tmp_bytes = 0;
  int32_t* d_w_flag = nullptr; // set by the ingest's scatter pass when a row holds a NaN / Inf
  // options
  int maxsim_variant = -1;
  int binary_variant = -1;
  int fde_scan_variant = -1;
  int batch_variant = -1;      // -1 = auto: pipelined kernel up to 384 query rows, 512-row kernel above
  int long_query_variant = 1;  // 1 = single queries > 64 rows use the row-split (batched) workgroup; 0 = page-split passes
  int fde_encode_variant = 4;  // 4 = documents from the bf16 slab in two passes (hash -> partitions; projection + one-hot MFMA bucket sums), 3 = the one-pass
                               // form (bf16 AMS, LDS-atomic bucket sums); both fall back to 1 for other inputs / shapes; 1 = f32-MFMA kernel, 0 = scalar kernel
  int fde_batch_variant = 0;   // mv_query_topk_batch in the FDE modes: 0 = batched pipeline (one slab pass per 32 queries), 1 = query by query,
                               // 2 = batched with the query FDE rounded to bf16 (no lo term)
  int fde_query_encode_variant = 2;  // the ONE query page: 2 = latency kernel (one block per repetition, default), 1 = bulk f32-MFMA kernel, 0 = scalar kernel
  int64_t fde_coarse_n = 0;
  int64_t rerank_n = 128;  // MV_MODE_FP8_THEN_FLOAT: candidates re-scored on the exact tier
  int exact_tier = 0;      // 0 = the bf16 slab in HBM when the index has one, else the pinned-host tier; 1 = the host tier when present
  int fde_cosine = 1;
  uint8_t* fde8 = nullptr;         // MV_WITH_FDE_E4M3: [capacity][out_dim] e4m3 copy of the FDE slab (derived after every write of `fde`)
  float* fde8_scale = nullptr;     // [capacity] value = decode(code) * scale (a power of two)
  float* fde8_cfac = nullptr;      // [capacity] scale / |d|: the page factor of the batched pass under the cosine rule (one load per page)
  uint8_t* fde4 = nullptr;         // MV_WITH_FDE_FP4: [capacity][out_dim / 2] e2m1 copy of the FDE slab (derived after every write of `fde`; mv_fde4.hip)
  float* fde4_scale = nullptr;     // [capacity] value = decode(code) * scale (a power of two)
  float* fde4_cfac = nullptr;      // [capacity] scale / |d|
  float* d_bqfac = nullptr;        // [32] the batched pass's per-query scales (lazily, with the batch workspace)
  int fde_coarse_e4m3 = 1;         // MV_OPT_FDE_COARSE_SLAB
  int pad_semantics = -1;  // -1: mode default (reference batch rule for FDE_THEN_FLOAT / candidates, none for full scan)
  int float_lo_scan = 1;   // MV_MODE_FLOAT full scans on an index with a lo slab: 1 = read both halves (fp32-faithful scores, 2 x the bytes),
                           // 0 = the hi half only (the bf16-rounded pages), 2 = hi-only scan -> top max(MV_OPT_RERANK_N, k) -> split-bf16 re-score
};

namespace mv {

// first slab row of a page / rows its slot holds, in either layout (host metadata)
inline int64_t page_row0(const mv_index* ix, int64_t p) { return ix->packed ? ix->h_row_off[(size_t)p] : p * (int64_t)ix->cfg.stride_rows; }
inline int32_t page_slot_rows(const mv_index* ix, int64_t p) {
  return ix->packed ? (int32_t)(ix->h_row_off[(size_t)p + 1] - ix->h_row_off[(size_t)p]) : ix->cfg.stride_rows;
}
inline int64_t rows_in_use(const mv_index* ix, int64_t size) { return ix->packed ? ix->h_row_off[(size_t)size] : size * (int64_t)ix->cfg.stride_rows; }

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (prev >= 0 && cur != prev) (void)hipSetDevice(prev);
  }
};

// Scan result left on the device (mv_api.hip).
struct ScanResult {
  const float* d_scores = nullptr;  // per work item
  int64_t n = 0;
  const int32_t* d_ids_map = nullptr;  // work item -> local page (candidate list) or null
  int launches = 0;
  int64_t pages = 0;
  int64_t bytes = 0;
};

// Internal entry points of mv_api.hip used by mv_comm.hip.  Callers hold ix->q_mu and have the index's device current.
int upload_query(mv_index* ix, const void* q, int q_dtype, int n_q, bool want_bf16, bool want_f32, bool want_bits, bool want_fp8);
int upload_allow(mv_index* ix, const uint32_t* allow_bits, int64_t n_words, const uint32_t** d_allow);
// k_next > 0: a top-k_next selection of these scores follows; when it takes the radix path the scan pre-bins its first
// histogram and *hist0_done tells the caller to pass that on to launch_topk.
int fde_coarse_scan(mv_index* ix, int n_q, const uint32_t* d_allow, int64_t n_words, int64_t n, int* launches, bool stage_events = false,
                    int32_t k_next = 0, bool* hist0_done = nullptr);
// exact rerank of the list in d_cand / d_cand_pads on `tier`
int rerank_scan(mv_index* ix, int n_q, int tier, int64_t n_items, float* d_out, int* launches);
// exact bf16 MaxSim of a candidate list on the bf16 slab (kTierSlab) or on the exact host tier (kTierHost): a tier split between HBM and
// host memory is scored in two launches (each part's candidates against its own base) and merged
// d_q_base / d_qlo_base: the query rows of a batch block (d_bq / d_bqlo + offset) instead of the single-query buffers
int exact_scan(mv_index* ix, int n_q, int tier, const int32_t* d_cand, int64_t n_items, int32_t pad_to, const int32_t* d_pad_items, float* d_out,
               int* launches, const uint16_t* d_q_base = nullptr, const uint16_t* d_qlo_base = nullptr);
// e4m3 scan of pages 0..n_items-1 (or of the candidate list d_cand) with the query uploaded by upload_query(want_fp8)
int fp8_scan(mv_index* ix, int n_q, const uint32_t* d_allow, int64_t n_allow_words, const int32_t* d_cand, int64_t n_items,
             int32_t pad_to, const int32_t* d_pad_items, float* d_out, int* launches, bool no_mask = false);
int64_t coarse_n_for(const mv_index* ix, int k);

// Which copy of the candidates' rows the rerank of a cascade reads (FastMultiVectorStore reranks with exact fp32 MaxSim on fp32
// pages: fast_multivector_store.py:553-556, upcast at load :736,774), and whether an e4m3 stage prunes the list first.
//   tier       the bf16 slab in HBM, or the exact host tier (MV_WITH_HOST_EXACT: pinned host memory mapped into the device; with
//              MV_WITH_EXACT_SPLIT its leading pages sit in HBM), or -- the index keeps no exact copy -- the e4m3 slab (final_fp8)
//   mid        the exact tier is host memory and the list is longer than MV_OPT_RERANK_N: the candidates are first re-scored on
//              the e4m3 slab (HBM) and only the n_mid best of them (ties by list position) are read over PCIe
// The caller's own query FDE vectors (mv_query_topk_fde and friends): for the duration of ONE API call on the calling thread the coarse stages
// take the query's FDE from here -- fp32 [queries][out_dim], host memory -- instead of encoding the query rows on the device.  cur = index of
// the call's query the stage at hand starts with (request-by-request loops and the groups of a batch move it).
struct QueryFdeOverride {
  const float* base = nullptr;
  int64_t cur = 0;
};
extern thread_local QueryFdeOverride g_qfde;
inline const float* query_fde_override(int64_t out_dim, int64_t j = 0) { return g_qfde.base ? g_qfde.base + (g_qfde.cur + j) * out_dim : nullptr; }
struct QueryFdeScope {  // sets the override for a call, restores what was there
  QueryFdeOverride prev;
  explicit QueryFdeScope(const float* base) : prev(g_qfde) { g_qfde = QueryFdeOverride{base, 0}; }
  ~QueryFdeScope() { g_qfde = prev; }
};
struct QueryFdeCursor {  // moves the cursor inside a call (a batch's group / request loop)
  int64_t prev;
  explicit QueryFdeCursor(int64_t cur) : prev(g_qfde.cur) { g_qfde.cur = cur; }
  ~QueryFdeCursor() { g_qfde.cur = prev; }
};
int check_fde_finite(const float* v, size_t n, const char* what);

enum { kTierSlab = 0, kTierHost = 1, kTierFp8 = 2 };  // what a rerank reads: the bf16 slab, the exact host tier (maybe split with HBM), the e4m3 slab
struct RerankPlan {
  int tier = kTierSlab;
  bool host_tier = false;
  bool final_fp8 = false;
  bool mid = false;
  int32_t n_mid = 0;
};
// n_list: candidates entering the rerank; rpq: padded query rows; batched: the batched one-launch rerank (e4m3 form: <= 64 rows)
RerankPlan rerank_plan(const mv_index* ix, int mode, int64_t n_list, int32_t k, int rpq, bool batched);
// Keep only the entries of each candidate list whose POSITION is named in pos ([nb][n_sel], padded with -1): the others become
// -1 (skipped by the rerank kernels).  The list keeps its order, so pad lengths and the tie rule (by list position) are untouched.
int launch_keep_selected(const int64_t* d_pos, int64_t pos_stride, int n_sel, int32_t* d_cand, int64_t cand_stride, int n, int nb, hipStream_t s);
int finish_stats(mv_index* ix, mv_query_stats* st, bool had_topk);
// bits of mv_query_stats::reserved that travel with a DEFERRED record (mv_query_topk_device_async -> mv_query_stats_finish): which of the two
// timing-event sets its query recorded into; the low bits are finish_stats' own stage flags
constexpr int32_t kStatsDeferredTag = 1 << 25, kStatsParityBit = 1 << 26;
// ... and the record of a deferred query that was complete when the call returned (nothing enqueued): mv_query_stats_finish just clears it
constexpr int32_t kStatsDoneTag = 1 << 24;
// user_stream value of mv_internal_query_common meaning "the NULL (default) stream, ordered against -- not `no stream, block`"
static void* const kNullStreamTag = reinterpret_cast<void*>(~(uintptr_t)0);

}  // namespace mv

namespace mv {
uint16_t host_f32_to_bf16(float f);
float host_bf16_to_f32(uint16_t h);
// every element finite (no NaN / Inf; for fp32 input also: inside the bf16 range, i.e. finite once rounded to bf16)
bool host_rows_finite(const void* x, int dtype, size_t n_elems);
// a query with a NaN / Inf row has no defined MaxSim (torch's einsum -> max -> topk would rank NaN first): refused, except in
// MV_MODE_BINARY, whose quantiser defines every input (bit = v > 0: NaN and +-0 give 0, binary_ops.rs:81-136)
int check_query_finite(const void* q, int q_dtype, size_t n_elems, int mode);
}
// Batch workspace and stages of mv_api.hip shared with the batched two-stage communicator (mv_comm.hip).  Callers hold q_mu.
extern "C" int mv_internal_ensure_batch_select_ws(mv_index* ix);
extern "C" int mv_internal_ensure_fde_batch_ws(mv_index* ix);
extern "C" void mv_internal_fde_batch_e4m3_args(mv_index* ix, mv::FdeScanBatchArgs* sa);  // the pass reads the e4m3 copy when the index has one (q_mu held)
extern "C" int mv_internal_ensure_fp8_batch_ws(mv_index* ix);
extern "C" int mv_internal_batch_upload_queries(mv_index* ix, const void* q, int q_dtype, int nb, int n_q_rows, bool want_f32, bool want_bf16, bool want_fp8);
// tier: kTierSlab / kTierHost (bf16 rerank on the slab / the exact host tier) or kTierFp8 (e4m3 rerank).  d_out null -> d_bcand_scores.
extern "C" int mv_internal_batch_rerank_lists(mv_index* ix, int nb, int n_q_rows, int64_t nc, int* launches, int tier, float* d_out);

extern "C" int mv_internal_query_common(mv_index* ix, const void* q, int q_dtype, int32_t n_q, int32_t k, int mode,
                                        const uint32_t* allow_bits, int64_t n_words, float* h_scores, int64_t* h_ids, int32_t* out_n,
                                        float* d_scores_out, int64_t* d_ids_out, void* user_stream, mv_query_stats* st, int defer_stats);
