/*
 * mvmaxsim.h -- C ABI of libmvmaxsim.so, the MI355X (gfx950) late-interaction retrieval engine.
 *
 * Flat C, caller-owned host buffers, library-owned device memory, no C++/torch types.
 * Every entry point returns 0 on success or a negative mv_status; mv_last_error() returns a
 * thread-local message for the last failure on the calling thread.
 *
 * Each function names the reference interface it replaces (paths relative to the
 * morphik-core checkout).  The reference has no FFI for this path today -- scoring lives in a
 * PostgreSQL SQL function, a third-party torch einsum and an un-vendored C++ extension -- so
 * these are the entry points a ctypes binding inside core/vector_store/ would bind
 * (INTEGRATION.md shows that binding).
 */
#ifndef MVMAXSIM_H
#define MVMAXSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_API __attribute__((visibility("default")))

typedef enum {
  MV_OK = 0,
  MV_ERR_INVALID = -1,   /* bad argument */
  MV_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
  MV_ERR_NOMEM = -3,     /* device or host allocation failed */
  MV_ERR_CAPACITY = -4,  /* slab full */
  MV_ERR_STATE = -5,     /* feature not enabled on this index */
  MV_ERR_IO = -6         /* save/load failure */
} mv_status;

typedef enum { MV_F32 = 0, MV_BF16 = 1 } mv_dtype;

/* Scoring modes of mv_query_topk / mv_score_all. */
typedef enum {
  MV_MODE_FLOAT = 0,          /* exact float MaxSim over every page (score_multi_vector semantics) */
  MV_MODE_BINARY = 1,         /* sign-bit MaxSim == SQL max_sim(bit[],bit[]) */
  MV_MODE_FDE_THEN_FLOAT = 2, /* FDE coarse top-n_coarse -> exact float rerank (FastMultiVectorStore).  The rerank reads the index's
                                 exact tier: the bf16 slab, else the pinned-host tier of MV_WITH_HOST_EXACT (a list longer than
                                 MV_OPT_RERANK_N is first re-scored on the e4m3 slab, when there is one, and only its
                                 max(MV_OPT_RERANK_N, k) best entries are read over PCIe).  An index with neither exact tier
                                 reranks on its e4m3 slab: NOT the reference's exact rerank (scores 0.3-1 % off) */
  MV_MODE_FDE_ONLY = 3,       /* coarse FDE scores only (the TurboPuffer ANN stage) */
  MV_MODE_FLOAT_FP8 = 4,      /* exact float MaxSim over the e4m3 slab (BASELINE configs[4]; 128 B / patch row) */
  MV_MODE_FP8_THEN_FLOAT = 5  /* e4m3 scan of every page -> top-n (MV_OPT_RERANK_N, default 128) -> exact bf16 re-score of the n
                                 candidates from the index's exact tier (the bf16 slab in HBM, or the pinned-host tier of
                                 MV_WITH_HOST_EXACT read by the rerank kernel itself over PCIe: n x 256 KiB, no staging copy
                                 and no host round trip) -> top-k.  The fp8 scan's 0.3-1 % score noise reorders near-ties;
                                 the exact tier restores the bf16 order among the candidates. */
} mv_mode;

/* Index feature flags (mv_config.flags). */
enum {
  MV_WITH_FLOAT = 1,  /* bf16 page slab (262 144 B / 1024-patch page) */
  MV_WITH_BINARY = 2, /* sign-bit slab   ( 16 384 B / page)            */
  MV_WITH_FDE = 4,    /* bf16 FDE slab   ( 20 480 B / page)            */
  MV_WITH_FP8 = 8,    /* e4m3 page slab  (131 072 B / page) + one power-of-two scale per page */
  MV_WITH_HOST_EXACT = 16, /* exact bf16 rows kept in PINNED HOST memory (262 144 B / page of host RAM, none of HBM), mapped
                             into the device's address space: the exact tier of MV_MODE_FP8_THEN_FLOAT for a shard whose
                             bf16 slab does not fit HBM beside its fp8 slab ("host-resident exact vectors
                             with a gather of the candidates").  May be combined with MV_WITH_FLOAT (both tiers
                             hold the same rows; MV_OPT_EXACT_TIER picks the one the rerank reads) */
  MV_WITH_EXACT_SPLIT = 32, /* with MV_WITH_HOST_EXACT (and no MV_WITH_FLOAT): split the exact tier -- the exact rows of the FIRST
                             pages go to whatever HBM is free once the other slabs are allocated (minus
                             MV_EXACT_HBM_RESERVE_BYTES, default 12 GiB, for the lazily allocated workspaces and the caller's
                             own device memory), the rest to pinned host memory.  A 1.25 M-page shard (328 GB of exact rows
                             beside 190 GB of FDE + e4m3 slabs) keeps ~105 GB of them in HBM and pins ~223 GB: it fits a
                             288 GiB GPU in a container that may pin 300 GiB, and a third of the rerank reads never cross
                             PCIe.  Same answers as an unsplit tier (mv_index_exact_hbm_pages: the pages the HBM part got) */
  MV_WITH_FLOAT_LO = 64,   /* with MV_WITH_FLOAT: a second bf16 slab holding lo = bf16(x - bf16(x)) of every element (+262 144 B / page).
                             The reference keeps its pages as fp32 `.npy` and scores them in fp32 (fast_multivector_store.py:676-681
                             save, :736 / :774 load, :553-555 score); hi + lo is the same 4 bytes per element, reproduces x to
                             2^-18 |x|, and both halves are bf16 MFMA operands: every candidate scorer (mv_score_candidates*, the
                             rerank stages of MV_MODE_FDE_THEN_FLOAT / MV_MODE_FP8_THEN_FLOAT) and -- MV_OPT_FLOAT_LO_SCAN -- the full
                             scan accumulate qhi.phi + qlo.phi + qhi.plo in fp32 before the max: the reference's fp32 score to
                             ~1e-6 relative on fp32 inputs.  Pages added as MV_BF16 have lo = 0 and score bit-identically to an
                             index without the flag.  (Independently of the flag, an MV_F32 QUERY is always split into hi + lo
                             and both halves are scored -- free where the scan is HBM-bound; a bf16-representable query takes
                             the one-term kernels, bit for bit as before.) */
  MV_LAYOUT_PACKED = 128,  /* PACKED page layout for ragged corpora (the reference's real encoder, ColQwen2.5, emits a different token
                             count per page: core/embedding/colpali_embedding_model.py:47-52): pages lie back to back in whole 16-row
                             tiles instead of one stride_rows slot each.  A table of row offsets (int64 per page, on the device) replaces
                             `page * stride_rows` in every kernel; the row-indexed slabs (bf16, its lo half, e4m3, sign bits) share it.
                             mv_config.capacity_rows sizes those slabs (rows, a multiple of 16; 0 = capacity_pages * stride_rows);
                             stride_rows remains the longest page the index accepts.  Appends publish as before; a page may be replaced
                             in place only by one that fits its tiles; mv_index_compact closes the holes of removed pages.  Same scores,
                             bit for bit, as the fixed-stride layout.  Not combinable with MV_WITH_HOST_EXACT. */
  MV_WITH_FDE_E4M3 = 256,  /* with MV_WITH_FDE: keep an e4m3 COPY of the FDE slab (out_dim bytes per page + one power-of-two scale) and run
                             the COARSE stage of the FDE modes on it -- half the bytes of the pass that is nine tenths of every
                             MV_MODE_FDE_THEN_FLOAT request.  The reference's coarse stage is an ANN index (TurboPuffer,
                             fast_multivector_store.py:527-533): approximate by contract; what the stage owes is the candidates.  Row
                             quantisation as the e4m3 page slab's (amax -> 2^e, RNE, saturating at 448; oracle: orc_quantize_page_fp8 on
                             the row viewed as out_dim / 128 rows of 128); the query FDE stays fp32 (single request) or enters as two e4m3
                             terms (batched).  Coarse scores differ from the bf16 slab's by ~1e-3 relative -- candidate sets by a page or
                             two at the cut (recall unchanged on every corpus of the bench).  MV_OPT_FDE_COARSE_SLAB selects
                             the slab per query.  Needs an FDE width of 10 240 (the reference's), 4 096 or 2 048; out_dim bytes per page more HBM. */
  MV_WITH_FDE_FP4 = 512    /* with MV_WITH_FDE (beside or instead of MV_WITH_FDE_E4M3): a copy in FP4 (e2m1: sign + {0, 0.5, 1, 1.5, 2, 3, 4, 6}) under one
                             power-of-two scale per row (half the scale that covers the row's largest element: those saturate, the bulk gets a finer grid)
                             -- out_dim / 2 bytes per page, a QUARTER of the bf16 slab's -- that the coarse stage reads: a single
                             request through a conversion scan (mv_fde4.hip; the query FDE stays fp32), a batch of requests through the FP4 form of
                             the batched pass (mv_fde_batch.hip: both MFMA operands FP4, the queries as two e2m1 terms; widths 10 240 / 8 192 / 4 096,
                             else the batch reads the bf16 slab).  Candidate lists differ from the bf16 slab's at the cut; recall@10 behind the exact
                             rerank, priced before it was built (tools/fde_4bit_recall_probe.py) and measured by the bench (aux_summary.fde4_75_recall_hard
                             beside fde75_recall_hard): DESIGN 3.23; equal to the bf16 slab's at 1000 candidates.  Quantiser: oracle
                             orc_quantize_fde_fp4, bit for bit.  Same widths as MV_WITH_FDE_E4M3.  MV_OPT_FDE_COARSE_SLAB 2 / 1 / 0 selects the slab per query
                             (default on an index with this flag: 2). */
};

/* fixed_dimensional_encoding.FixedDimensionalEncodingConfig as constructed at
 * core/vector_store/fast_multivector_store.py:325-331 (AMS_SKETCH projection, query=SUM,
 * document=AVERAGE). */
typedef struct {
  int32_t dimension;               /* 128 */
  int32_t num_repetitions;         /* 20  */
  int32_t num_simhash_projections; /* 5   */
  int32_t projection_dimension;    /* 16  */
  uint64_t seed;                   /* 1   */
} mv_fde_config;

typedef struct {
  int32_t dim;            /* embedding width; must be 128 */
  int32_t stride_rows;    /* patch rows reserved per page (multiple of 16), e.g. 1024 or 1040 */
  int64_t capacity_pages; /* pages the slabs are sized for (fixed at create) */
  int32_t device;         /* HIP device ordinal */
  int32_t flags;          /* MV_WITH_* */
  int64_t id_base;        /* global page id of local page 0 (row-sharded corpora: rank r passes r*N/R) */
  mv_fde_config fde;      /* used when flags & MV_WITH_FDE */
  int64_t capacity_rows;  /* MV_LAYOUT_PACKED: rows the row-indexed slabs hold in all (multiple of 16); 0 = capacity_pages * stride_rows */
} mv_config;

/* Per-call timing and accounting, filled when a non-NULL pointer is passed. Times are HIP-event
 * durations on the index's stream (the stream the kernels were launched on). */
typedef struct {
  float score_kernel_ms;  /* the page-scan kernel(s) only */
  float topk_ms;          /* selection kernels */
  float total_device_ms;  /* first launch .. last kernel */
  int32_t score_launches; /* kernel launches that make up score_kernel_ms */
  int32_t reserved;
  int64_t pages_scored;   /* pages whose embeddings were read (masked pages excluded) */
  int64_t bytes_scanned;  /* algorithmic bytes: pages_scored * rows * row_bytes */
  /* Stage split of the FDE modes -- the device-side counterparts of the reference's per-stage log lines
   * (fast_multivector_store.py:521-605: encode_query / ns.query / load_multivectors + rerank_scoring); 0 elsewhere. */
  float encode_ms;        /* fde.generate_query_encoding */
  float coarse_ms;        /* FDE slab scan (the ANN stage) */
  float select_ms;        /* coarse top-n selection + candidate list */
  float rerank_ms;        /* exact MaxSim of the candidates */
} mv_query_stats;

typedef struct mv_index mv_index;

/* Tunables for mv_index_set_option. */
typedef enum {
  MV_OPT_MAXSIM_VARIANT = 1, /* float kernel variant id (14 = persistent-stream form); -1 = default */
  MV_OPT_FDE_COARSE_N = 2,   /* candidates kept by the FDE stage (reference: min(10*k, 75)) ; 0 = reference rule */
  MV_OPT_FDE_COSINE = 3,     /* 1 = rank coarse stage by cosine (TurboPuffer cosine_distance, reference), 0 = dot */
  MV_OPT_PAD_SEMANTICS = 4,  /* 0 = max over a page's own rows only; 1 = reference rerank batch rule
                                (pad_to = longest candidate of the batch of 128 => clamp at 0) */
  MV_OPT_BINARY_VARIANT = 5, /* sign-bit scan: 0 = popcount on the VALU (the independent cross-check), 4 = FP4 MFMA with in-place bit
                                operands and a 4-slot ring (default); the same integers.  (1-3, 5, 6 lost by measurement and were
                                removed in round 5: MV_ERR_INVALID.) */
  MV_OPT_FDE_SCAN_VARIANT = 6, /* FDE coarse scan: 6 = row quarters through the nt LDS-DMA ring, one fresh workgroup per 256 KiB-aligned block
                                * of the slab (default), 5 = the same with a workgroup per 16 consecutive rows,
                                * 0 = one wave per row on plain nt loads (the same arithmetic order in all three: bit-identical scores) */
  MV_OPT_BATCH_VARIANT = 7,   /* batched float scan: -1 / 0 = auto (default): page-split form (wave-private rings, no barrier) up to 128
                                 query rows in the group, row-split workgroup above; both with transposed MFMA roles (one running
                                 max per query tile, v_max3).  3 = row-split form always, 4 = page-split form (<= 128 rows): each is
                                 the other's cross-check.  (1, 2, 5, 6 lost by measurement and were removed in round 5.)
                                 MV_MODE_FLOAT_FP8 batches (the batched block-scaled MFMA scan of the e4m3 slab): 7 = ONE e4m3
                                 term per query row instead of the hi + lo split (half the matrix work), 8 = query by query.
                                 MV_MODE_FP8_THEN_FLOAT batches use ONE term by default (the first stage only nominates candidates,
                                 the exact tier restores their order); 0 = the two-term scores of the single-query scan */
  MV_OPT_FDE_ENCODE_VARIANT = 8, /* FDE encode of corpus pages: 4 = (default) pages that are already bf16 -- the slab -- in two passes: SimHash
                                    partitions (fp32 fmaf chains, columns in registers) to a scratch byte per (row, repetition), then the
                                    AMS projection on the bf16 matrix pipe with the bucket sums as a one-hot f32 matrix product (no LDS
                                    atomics: a fixed summation order); other inputs / FDE shapes run as 1; 1 = f32-MFMA kernel, 0 = scalar
                                    kernel (the cross-check).  The same partitions bit for bit in all of them.  (3, round 3's one-pass
                                    form with LDS-atomic bucket sums, was removed in round 5 and runs as 1.) */
  MV_OPT_FILTER_COMPACT_PCT = 9, /* doc filter allowing < this % of the documents: compact the allowed pages first and scan
                                    only those (default 25; 0 = always mask inside the scan) */
  MV_OPT_FDE_QUERY_ENCODE_VARIANT = 11, /* FDE encode of the query (one page, latency matters): 2 = latency kernel, one block per
                                    repetition (default), 1 = the bulk f32-MFMA kernel, 0 = scalar kernel; same partitions bit for bit */
  MV_OPT_LONG_QUERY_VARIANT = 10, /* single query of > 64 rows over the whole slab: 1 = row-split workgroup of the batched scan
                                    (default), 0 = page-split kernel in passes of 128 rows */
  MV_OPT_RERANK_N = 13,          /* candidates read from the exact tier (1..1024, default 128): the list of MV_MODE_FP8_THEN_FLOAT; in
                                    MV_MODE_FDE_THEN_FLOAT over a pinned-host exact tier the cut of the e4m3 pruning stage (coarse lists
                                    up to this length -- the reference's min(10 k, 75) -- go to the exact tier whole) */
  MV_OPT_EXACT_TIER = 14,        /* reranks on an index holding BOTH exact tiers: 0 = the bf16 slab in HBM (default),
                                    1 = the pinned-host tier (what an index without a bf16 slab always uses), 2 = the e4m3 slab
                                    (MV_MODE_FDE_THEN_FLOAT / mv_score_candidates only: the scores an index WITHOUT an exact tier
                                    returns; the host-driven cross-check of the pruning stage uses it) */
  MV_OPT_FLOAT_LO_SCAN = 15,     /* MV_MODE_FLOAT over every page of an index with MV_WITH_FLOAT_LO: 1 = read hi and lo (default: the
                                    fp32-faithful score of every page, twice the bytes per page), 0 = the hi slab only (scores of the
                                    bf16-rounded pages, the speed of an index without the flag), 2 = hi-only scan -> top
                                    max(MV_OPT_RERANK_N, k) -> split-bf16 re-score of those -> top-k (fp32-faithful scores at the
                                    hi-only scan's speed; a page whose hi-only score misses the candidate cut by rounding alone --
                                    ~1e-4 relative -- would be lost: same caveat as MV_MODE_FP8_THEN_FLOAT, three orders smaller) */
  MV_OPT_FDE_COARSE_SLAB = 16,   /* which slab the coarse stage reads: 0 = the bf16 slab, 1 = the e4m3 copy (default on an index with MV_WITH_FDE_E4M3),
                                    2 = the fp4 copy (default on an index with MV_WITH_FDE_FP4; single requests) */
  MV_OPT_FDE_BATCH_VARIANT = 12  /* mv_query_topk_batch in the FDE modes: 0 = batched pipeline (default): up to 32 queries per pass
                                    over the FDE slab (bf16 MFMA, query FDE as bf16 hi + lo), batched selection, every query's
                                    candidates reranked in one launch, one read-back; 1 = query by query; 2 = as 0 with the query
                                    FDE rounded to bf16 (one MFMA per fragment: -13 % pass time, coarse scores within ~2e-3);
                                    3 = as 0 with one page tile per query fragment (the first form of the coarse kernel; the
                                    default walks a workgroup's tiles in pairs: same scores, half the fragment traffic);
                                    5 = as 0 with the round-2 structure: the cosine rule / tombstones as a finish pass of their
                                    own (the default applies them where the scan kernel writes a tile's scores) and all three
                                    passes of the selection (variant 3 lets its finish pass pre-bin the scores for the
                                    selection's first pass); the same results bit for bit.
                                    (4, 6, 7, 8 -- 32-page tiles, private rings, deeper / shallower rings -- lost by measurement and
                                    were removed in round 5: MV_ERR_INVALID.) */
} mv_option;

/* Bumped whenever a signature, a struct layout or an enum value of this header changes incompatibly.  A binding compares
 * mv_abi_version() of the library it loaded with the MV_ABI_VERSION it was written against and refuses a mismatch (a stale
 * libmvmaxsim.so driven with newer argument lists would corrupt memory silently). */
#define MV_ABI_VERSION 7
MV_API int mv_abi_version(void);

MV_API const char* mv_last_error(void);
MV_API const char* mv_version(void);
MV_API int mv_device_count(void);
/* Bytes of host memory this process can still pin (MV_WITH_HOST_EXACT: 262 144 B per 1024-row page) without running into its memory
 * cgroup limit or the machine's available memory, minus a headroom of max(4 GiB, 5 %); INT64_MAX = unlimited.  mv_index_create
 * refuses an exact tier beyond it with MV_ERR_NOMEM -- a container over its cgroup limit is killed during the allocation, not told.
 * MV_HOST_EXACT_MAX_BYTES in the environment lowers it.  No GPU needed. */
MV_API int64_t mv_host_pin_budget_bytes(void);

/* Lifecycle.  Replaces MultiVectorStore.initialize() (core/vector_store/multi_vector_store.py:186-327:
 * table + max_sim function creation) and FastMultiVectorStore.__init__ (fast_multivector_store.py:296-338). */
MV_API int mv_index_create(const mv_config* cfg, mv_index** out);
MV_API void mv_index_destroy(mv_index* ix);
MV_API int mv_index_set_option(mv_index* ix, int option, int64_t value);
MV_API int64_t mv_index_size(const mv_index* ix);     /* pages appended so far (including tombstoned) */
MV_API int64_t mv_index_capacity(const mv_index* ix);
MV_API int64_t mv_index_rows_used(const mv_index* ix);     /* slab rows taken by the pages appended so far (packed layout: whole 16-row tiles per page;
                                                              fixed layout: pages * stride_rows) */
MV_API int64_t mv_index_capacity_rows(const mv_index* ix); /* rows the row-indexed slabs hold */
MV_API int64_t mv_index_exact_hbm_pages(const mv_index* ix); /* pages of a split exact tier (MV_WITH_EXACT_SPLIT) whose rows live in HBM; 0 otherwise */
/* PLACEMENT of a split exact tier.  Every rerank counts, per page, the reads of its exact rows; mv_index_exact_tier_rebalance swaps the
 * most-read pages that live in pinned host memory with the least-read pages that live in HBM (while a swap raises the HBM part's
 * read count; at most max_moves swaps, 0 = no limit), then clears the counters.  Page ids, answers and scores are unchanged -- only
 * where a page's exact rows are read from: the PCIe share of a rerank falls to the share of COLD candidates.  Exclusive with queries
 * and writers for its duration (2 x 256 KiB over PCIe per swap).  A store calls it after compaction or from a maintenance task.
 * mv_index_exact_tier_hits: reads since the last rebalance that were served from HBM / from host memory.  No-ops without a split. */
MV_API int mv_index_exact_tier_rebalance(mv_index* ix, int64_t max_moves, int64_t* out_moved);
MV_API int mv_index_exact_tier_hits(mv_index* ix, int64_t* out_hbm_reads, int64_t* out_host_reads);
/* PLACEMENT of the FDE slab.  The batched coarse pass (mv_query_topk_batch in the FDE modes) reads the slab in 512-byte pieces at a 2*out_dim-byte
 * stride, and its time depends on WHICH physical memory hipMalloc gave the slab: up to 10 % between allocations of one process, constant for the
 * life of an allocation.  mv_index_fde_placement_trial takes up to `trials` further allocations of the slab's size, one after the
 * other, copies the slab into each, times the pass on it (5 launches) and keeps the faster of the two; the loser is freed.  Peak device memory:
 * three slabs; a candidate that cannot be allocated ends the trial quietly.  Contents, ids, answers and scores are unchanged.  Exclusive with
 * queries and writers for its duration (about 25 ms per trial and 10 GB of slab).  out_before_ms / out_after_ms: the pass's time (32 requests)
 * on the slab it found / the slab it leaves; out_moves: how often a candidate won.  Any pointer may be NULL.  MV_ERR_INVALID without an FDE slab. */
MV_API int mv_index_fde_placement_trial(mv_index* ix, int32_t trials, double* out_before_ms, double* out_after_ms, int32_t* out_moves);

/* NON-FINITE VALUES.  A NaN / +-Inf embedding has no defined MaxSim (the reference's torch einsum -> max -> topk propagates NaN
 * and ranks it FIRST), and the scan kernels are compiled without NaN handling.  So they never enter a float-derived slab:
 *   - mv_index_add / _add_device / _replace_page / _write_rows return MV_ERR_INVALID (nothing is published) when a row holds a NaN,
 *     an Inf, or an fp32 value that rounds to Inf in bf16 -- one flag set by the ingest pass that touches every row anyway;
 *   - queries (mv_query_topk*, mv_score_*, mv_two_stage_*, mv_comm_query_*) with such a row return MV_ERR_INVALID;
 *   - the ONE exception is the sign-bit path, whose quantiser defines every input (bit = v > 0: NaN, +0 and -0 give 0, +Inf 1,
 *     -Inf 0 -- core/utils/fast_ops.py:191-227, morphik_rust/src/binary_ops.rs:81-136): an index with MV_WITH_BINARY only accepts
 *     such pages, MV_MODE_BINARY accepts such queries, mv_sign_pack packs them by that rule.
 * -0.0 is an ordinary value everywhere (it scores like +0.0). */

/* Append pages.  `emb` is a HOST buffer of sum(n_rows) rows x dim, pages back to back (ragged),
 * dtype MV_F32 or MV_BF16.  doc_ordinals[i] >= 0 is the caller's dense document number used by
 * the doc_ids filter.  The call fills every enabled slab (bf16 rows, sign bits, FDE) on the GPU.
 * Replaces store_embeddings' device-independent work: MultiVectorStore._binary_quantize +
 * INSERT (multi_vector_store.py:681-699,963-979) and FastMultiVectorStore's
 * fde.generate_document_encoding + np.save (fast_multivector_store.py:447-449,673-707).
 * Returns the local id of the first appended page in *out_first_page. */
MV_API int mv_index_add(mv_index* ix, const void* emb, int dtype, const int32_t* n_rows, int64_t n_pages,
                        const int32_t* doc_ordinals, int64_t* out_first_page);
/* Same, `emb` is a DEVICE pointer on the index's device (encoder output stays on the GPU). */
MV_API int mv_index_add_device(mv_index* ix, const void* d_emb, int dtype, const int32_t* n_rows, int64_t n_pages,
                               const int32_t* doc_ordinals, int64_t* out_first_page);

/* Append pages given only their packed sign rows (sum(n_rows) x 16 bytes, MSB first) -- the import path for an
 * existing MultiVectorStore table (BIT(128)[] column, multi_vector_store.py:248).  The index must have been created
 * with flags == MV_WITH_BINARY (floats cannot be recovered from sign bits). */
MV_API int mv_index_add_bits(mv_index* ix, const uint8_t* bits, const int32_t* n_rows, int64_t n_pages,
                             const int32_t* doc_ordinals, int64_t* out_first_page);

/* Tombstone every page of a document / one page.  Replaces
 * delete_chunks_by_document_id (multi_vector_store.py:921-951). Returns pages removed in *out_n. */
MV_API int mv_index_remove_doc(mv_index* ix, int32_t doc_ordinal, int64_t* out_n);
MV_API int mv_index_remove_page(mv_index* ix, int64_t page);

/* Reclaim the slots of tombstoned pages: live pages move down IN ORDER to a dense prefix of every slab (in place,
 * through a 256 MiB staging buffer).  out_old_to_new (nullable) has mv_index_size() entries: new local page id or -1.
 * Page ids change: the caller remaps whatever it keyed by page id.  *out_new_size = pages after compaction. */
MV_API int mv_index_compact(mv_index* ix, int64_t* out_old_to_new, int64_t* out_new_size);

/* Read back bf16 rows of pages [page0, page0+n) (stride_rows x dim each) to a host buffer (from the bf16 slab, or from the
 * pinned-host exact tier of an index without one). */
MV_API int mv_index_read_pages(mv_index* ix, int64_t page0, int64_t n_pages, void* out_bf16);
/* The same pages as fp32 (stride_rows x dim floats each): hi + lo of an index with MV_WITH_FLOAT_LO -- the values the split-bf16
 * scorers see, equal to the fp32 input to 2^-18 relative (the reference's `.npy` payload, fast_multivector_store.py:676-681) --
 * else the bf16 rows widened. */
MV_API int mv_index_read_pages_f32(mv_index* ix, int64_t page0, int64_t n_pages, float* out_f32);
/* Overwrite rows [row0,row0+n) of one page with host bf16 data (test/bench: planted neighbours).
 * Only the float slab is touched. */
MV_API int mv_index_write_rows(mv_index* ix, int64_t page, int32_t row0, int32_t n, const void* bf16_rows);
/* Overwrite one whole page from HOST bf16 rows (n_rows x 128) and refresh every enabled slab (bf16, sign bits,
 * FDE, fp8).  The update path of a re-embedded page; also how bench/tests plant neighbours on any slab mix. */
MV_API int mv_index_replace_page(mv_index* ix, int64_t page, const void* bf16_rows, int32_t n_rows);
/* Read back e4m3 codes (stride_rows x 128 bytes per page) and the per-page 2^-e scales. */
MV_API int mv_index_read_fp8(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_inv_scale);
/* Synthetic corpus: append n_pages pages of n_rows rows generated ON THE DEVICE by the
 * counter-based generator (key=seed, counter=(unit,row,chunk)); unit of page i = first_unit + i.
 * Bit-identical to oracle/mv_oracle.c:orc_synth_rows. doc ordinal of page i = (first_unit+i) / pages_per_doc.
 * Without MV_WITH_FLOAT the bf16 image is staged in chunks and only its derivatives (bits / FDE / fp8) are kept. */
MV_API int mv_index_fill_synthetic(mv_index* ix, uint64_t seed, uint64_t first_unit, int64_t n_pages, int32_t n_rows,
                                   int32_t pages_per_doc);
/* The same generator with a DIFFERENT row count per page -- the shape of a ColQwen2.5 corpus (dynamic token counts): page i (unit
 * u = first_unit + i) gets its first n(u) rows, n(u) = min_rows + splitmix64(seed ^ 0x9E3779B97F4A7C15 * (u + 1)) % (max_rows - min_rows + 1)
 * (oracle/oracle.py: synth_ragged_rows restates it).  Both layouts; max_rows <= stride_rows. */
MV_API int mv_index_fill_synthetic_ragged(mv_index* ix, uint64_t seed, uint64_t first_unit, int64_t n_pages, int32_t min_rows, int32_t max_rows,
                                          int32_t pages_per_doc);
/* Host helper: the same generator for queries (n_rows x 128 bf16 to a host buffer, computed on the GPU). */
MV_API int mv_synth_rows(int device, uint64_t seed, uint64_t unit, int32_t n_rows, void* out_bf16);

/* Top-k query.  Replaces the scoring half of MultiVectorStore.query_similar
 * (multi_vector_store.py:721-763: quantise query -> SQL max_sim -> ORDER BY similarity DESC LIMIT k)
 * for MV_MODE_BINARY and of FastMultiVectorStore.query_similar (fast_multivector_store.py:504-556:
 * fde.generate_query_encoding -> ANN -> score_multi_vector -> torch.topk) for the other modes.
 *   q           host buffer, n_q_rows x dim, MV_F32 or MV_BF16
 *   allow_bits  optional bitmap over doc ordinals (bit set = allowed), n_allow_words 32-bit words;
 *               NULL = no doc_ids filter
 *   out_scores  k floats, out_ids k int64 (GLOBAL page ids = id_base + local), *out_n = results (<= k)
 * Order: score descending, ties by ascending id (upstream order under ties is unspecified). */
MV_API int mv_query_topk(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                         const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids,
                         int32_t* out_n, mv_query_stats* stats);
/* Same selection, results left on the device (for the RCCL all-gather of per-shard top-k):
 * d_out_scores/d_out_ids are DEVICE buffers of k entries, padded with (-inf, -1).  stream = the
 * hipStream_t to order against (NULL = index stream: the call returns after the kernels finished;
 * with a stream it only enqueues, unless stats != NULL, which waits for the event timings). */
MV_API int mv_query_topk_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                                const uint32_t* allow_bits, int64_t n_allow_words, float* d_out_scores,
                                int64_t* d_out_ids, void* stream, mv_query_stats* stats);
/* The enqueue-only form WITH timings (one rank's step of a row-sharded search: the scan, then the all-gather and the merge
 * behind it on `stream`, nothing of which should wait for the host): as mv_query_topk_device with a non-NULL stream, but
 * `stats` (non-NULL) receives only the accounting fields now; the HIP-event timings of THIS query are filled in by
 * mv_query_stats_finish(ix, stats) -- which waits for the scan's events.  ONE such record may stay outstanding across the NEXT
 * mv_query_topk_device_async on `ix` (the library keeps two sets of timing events): a caller that collects query i's timings right
 * after enqueueing query i+1 never lets the GPU wait for the host.  (Records of the FDE modes carry stage accounting that reads the
 * candidate list: finish those before the next query, MV_ERR_STATE otherwise; and a query WITH timings through any other entry point
 * in between records into the outstanding record's events.)
 * stream = NULL names the legacy default stream here (the call still only enqueues). */
MV_API int mv_query_topk_device_async(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                                      const uint32_t* allow_bits, int64_t n_allow_words, float* d_out_scores,
                                      int64_t* d_out_ids, void* stream, mv_query_stats* stats);
MV_API int mv_query_stats_finish(mv_index* ix, mv_query_stats* stats);
/* Top-k of a BATCH of queries in one pass over the slab (score_multi_vector takes a list of queries:
 * fast_multivector_store.py:553-555 passes [query_embedding]; a serving process batches concurrent requests).
 *   q          host buffer, n_queries x n_q_rows x dim (every query padded to n_q_rows with zero rows -- a zero
 *              query row contributes exactly 0, the reference's own padding rule; NOT in MV_MODE_BINARY, where SQL
 *              max_sim scores a row of zero bits like any other: there every query must have exactly n_q_rows rows)
 *   allow_bits n_allow_words words shared by all queries (allow_per_query = 0) or n_queries bitmaps of n_allow_words
 *              words back to back (allow_per_query = 1: every request keeps its own doc_ids / auth filter)
 *   out_*      n_queries x k entries (row b = results of query b), out_n[b] = results of query b
 * MV_MODE_FLOAT runs the batched MFMA kernel (up to 512 query rows per slab pass: B x Q flop/byte); on an index with
 * MV_WITH_FLOAT_LO in cascade mode (MV_OPT_FLOAT_LO_SCAN 2) that pass nominates every request's max(MV_OPT_RERANK_N, k) pages
 * and ONE launch re-scores all lists with split-bf16 operands (the single request's answers); with MV_OPT_FLOAT_LO_SCAN 1, or
 * fp32 queries on a plain index, the requests of the call are served one by one by the split-bf16 scan.
 * MV_MODE_FLOAT_FP8 runs its e4m3 form (block-scaled MFMA, K = 128 per instruction): half the page bytes per pass.
 * MV_MODE_FP8_THEN_FLOAT runs that pass, then re-scores every request's fp8 top-n exactly from the exact tier in ONE launch.
 * MV_MODE_FDE_THEN_FLOAT / MV_MODE_FDE_ONLY run the batched FDE pipeline: one pass over the FDE slab per 32 queries
 * (the coarse stage as a bf16-MFMA GEMM, query FDEs as bf16 hi + lo: coarse scores within ~1e-5 of the single-query
 * scan), one batched selection, the exact rerank of every query's candidates, one read-back.  The remaining modes are
 * served query by query.  stats sum over the passes. */
MV_API int mv_query_topk_batch(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k,
                               int mode, const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query,
                               float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats);
/* Bring-your-own FDE.  The reference computes its FDE vectors with a C++ extension that is not part of its source tree
 * (`fde.generate_document_encoding` / `generate_query_encoding`: fast_multivector_store.py:447-449, :521); this library restates the
 * published algorithm, which matches that extension only as far as the publication pins it.  Where the extension is
 * installed -- or document FDEs already exist in a TurboPuffer namespace -- a deployment can keep ITS vectors and use the GPU for the
 * scan and the rerank:
 *   mv_index_import_fde      document FDE vectors (host fp32 [n_pages][mv_fde_output_dim(cfg)]) REPLACE the encodings the library made
 *                            of pages [page0, page0 + n_pages) at ingest (bf16-rounded into the slab, 1/|d| of the rounded vector
 *                            beside it: the encode kernels' own convention).  mv_index_replace_page re-encodes that page: import again.
 *   mv_query_topk_fde,       mv_query_topk / _batch (MV_MODE_FDE_THEN_FLOAT or MV_MODE_FDE_ONLY) with the caller's query FDE vector(s)
 *   mv_query_topk_batch_fde  (host fp32 [n_queries][out_dim]) in place of the encoding of the query rows on the device; the rerank
 *                            still scores the query ROWS (q).  NaN / Inf in either kind of vector: MV_ERR_INVALID. */
MV_API int mv_index_import_fde(mv_index* ix, int64_t page0, int64_t n_pages, const float* fde);
/* the FDE slab's rows of pages [page0, page0 + n_pages) as fp32 (the bf16 values the scan reads; host buffer of n_pages x out_dim floats):
 * the library's own document encodings, or what was imported -- for export to another store and for tests */
MV_API int mv_index_read_fde(mv_index* ix, int64_t page0, int64_t n_pages, float* out);
/* the e4m3 copy of the same rows (MV_WITH_FDE_E4M3): n_pages x out_dim codes and one scale per page -- value = decode(code) * scale */
MV_API int mv_index_read_fde_e4m3(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_scale);
/* the fp4 copy of the same rows (MV_WITH_FDE_FP4): n_pages x out_dim / 2 bytes of e2m1 codes (element 2i in the low nibble) and one scale per page */
MV_API int mv_index_read_fde_fp4(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_scale);
MV_API int mv_query_topk_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t k, int mode,
                             const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n,
                             mv_query_stats* stats);
MV_API int mv_query_topk_batch_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, const float* q_fde,
                                   int32_t k, int mode, const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query,
                                   float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats);
/* Merge of the per-shard top-k lists after the RCCL all-gather (row-sharded corpus).  d_scores / d_ids: DEVICE buffers
 * [world][kk], each row sorted (score desc, id asc), padded with (-inf, -1), rank r owning ids below rank r+1's;
 * world*kk <= 2048.  Writes k entries (padded the same way) on `stream` (enqueue only).  Same tie rule as one index. */
MV_API int mv_merge_topk(int device, const float* d_scores, const int64_t* d_ids, int32_t world, int32_t kk, int32_t k,
                         float* d_out_scores, int64_t* d_out_ids, void* stream);
/* The same merge over the result of ONE all-gather: every rank contributes one block of mv_topk_block_bytes(kk) bytes --
 * {int64 ids[kk], float scores[kk]}, padded to a multiple of 16 -- (mv_query_topk_device writes its two outputs straight into the
 * two halves of the local block), d_blocks = the world gathered blocks in rank order. */
MV_API int64_t mv_topk_block_bytes(int32_t kk);
MV_API int mv_merge_topk_blocks(int device, const void* d_blocks, int32_t world, int32_t kk, int32_t k, float* d_out_scores,
                                int64_t* d_out_ids, void* stream);

/* Score every page (no selection) into a host buffer of out_cap floats; masked pages get -inf.  *out_n = entries
 * written = min(published pages, out_cap) -- the corpus may grow between the caller's mv_index_size() and this call.
 * For MV_MODE_FDE_* this returns the coarse scores. */
MV_API int mv_score_all(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode,
                        const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t out_cap,
                        int64_t* out_n, mv_query_stats* stats);
/* Exact float MaxSim of an explicit candidate list (local page ids), in list order, on the index's exact tier (the bf16 slab,
 * else the pinned-host tier; an index with neither scores on its e4m3 slab).  Replaces
 * processor.score_multi_vector(...) at fast_multivector_store.py:553-555 (colpali_engine: passages are scored in
 * batches of 128, each batch zero-padded to ITS longest page by pad_sequence, so a page shorter than its batch's
 * longest sees zero rows and every query token's maximum is clamped at 0).
 *   pad_to = -1  the reference rule: pad length of candidate i = longest page among candidates [128*(i/128), +128)
 *   pad_to =  0  no padding: the maximum runs over a page's own rows only
 *   pad_to >  0  one explicit pad length for the whole list */
MV_API int mv_score_candidates(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const int32_t* cand,
                               int32_t n_cand, int32_t pad_to, float* out_scores, mv_query_stats* stats);
/* Same with an explicit pad length PER candidate (pads[i] rows): a row-sharded rerank scores only the candidates a
 * shard owns, but each one's pad length is that of its batch in the GLOBAL candidate list. */
MV_API int mv_score_candidates_pads(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const int32_t* cand,
                                    int32_t n_cand, const int32_t* pads, float* out_scores, mv_query_stats* stats);

/* Row counts of the named pages (local page ids) from the index's host-side metadata; no device work.  A sharded
 * FDE_THEN_FLOAT query needs them to apply the reference's pad-to-longest rule (pad_sequence at
 * fast_multivector_store.py:553-555) over a candidate list whose pages live on several ranks. */
MV_API int mv_index_page_rows(mv_index* ix, const int32_t* pages, int64_t n_pages, int32_t* out_rows);

/* Stateless helpers (host in, host out, computed on `device`).
 * mv_sign_pack replaces fast_ops.binary_quantize_packed (core/utils/fast_ops.py:191-227 ->
 * morphik_rust binary_quantize_batch_packed, src/binary_ops.rs:148-222): bit = v > 0, MSB first. */
MV_API int mv_sign_pack(int device, const float* x, int64_t n_rows, int32_t d, uint8_t* out);
/* mv_hamming_batch replaces fast_ops.hamming_distance_batch (fast_ops.py:242-248, binary_ops.rs:267-292). */
MV_API int mv_hamming_batch(int device, const uint8_t* query, const uint8_t* cands, int64_t n_cands, int32_t n_bytes,
                            int32_t* out);
/* mv_fde_encode replaces fde.generate_query_encoding / generate_document_encoding
 * (fast_multivector_store.py:521 / :447-449).  out has mv_fde_output_dim(cfg) floats. */
MV_API int64_t mv_fde_output_dim(const mv_fde_config* cfg);
MV_API int mv_fde_encode(int device, const mv_fde_config* cfg, const float* x, int32_t n_rows, int32_t is_query,
                         float* out);

/* Encoder adapters (A1 / A2, morphik_core_amd/encoder_ops.py): the two elementwise chains of the PaliGemma / Qwen2-VL blocks that the
 * framework runs as strings of kernels over fp32 copies, each as ONE pass over bf16 operands.  DEVICE pointers, the caller's
 * stream (0 = the null stream).  The reference formulation is model(**processor(x)) under bf16 autocast
 * (core/embedding/colpali_embedding_model.py:251-262, 275-305); the arithmetic here is the transformers modules', in their order.
 *   mv_enc_rmsnorm_bf16   rows x dim bf16 (contiguous).  style 0 (GemmaRMSNorm): bf16((x * rsqrt(mean x^2 + eps)) * (weight_offset + w));
 *                         style 1 (Llama / Qwen2 RMSNorm): bf16(w * bf16(x * rsqrt(mean x^2 + eps))).  weight: dim values, MV_F32 or MV_BF16.
 *   mv_enc_gated_act_bf16 out[r][c] = bf16( bf16(act(gate[r][c])) * up[r][c] ), out contiguous rows x cols; gate / up rows start every
 *                         *_row_stride elements (the two halves of one fused gate|up GEMM output).  act: 0 = gelu (tanh form),
 *                         1 = silu, 2 = gelu (erf form).  cols, strides: multiples of 8; pointers 16-byte aligned. */
MV_API int mv_enc_rmsnorm_bf16(int device, const void* d_x, const void* d_weight, int weight_dtype, void* d_out, int64_t rows, int32_t dim,
                               float eps, float weight_offset, int style, void* stream);
MV_API int mv_enc_gated_act_bf16(int device, const void* d_gate, int64_t gate_row_stride, const void* d_up, int64_t up_row_stride, void* d_out,
                                 int64_t rows, int64_t cols, int act, void* stream);

/* Calibration: stream-read `bytes` of device memory `iters` times, returns average GB/s (measured peak
 * for the roofline denominator next to the 8 TB/s datasheet figure). */
MV_API int mv_calibrate_read_bw(int device, int64_t bytes, int32_t iters, double* out_gbps);

/* Measured peaks for the roofline denominators, taken in the same process as the measurement:
 *   MV_CAL_READ_NT    streams `bytes` of device memory `iters` times in contiguous 16 KiB pieces with non-temporal
 *                     loads (the scan kernels' access pattern, no arithmetic)            -> *out in GB/s
 *   MV_CAL_MFMA_BF16  register-only v_mfma_f32_16x16x32_bf16 chains on every CU (pseudo-random operands of
 *                     embedding magnitude, ~4 ms per launch: the sustained clock, not a burst) -> *out in TFLOP/s
 *   MV_CAL_MFMA_BF16_32X32  the same with v_mfma_f32_32x32x16_bf16 (half the operand-register reads per flop) -> TFLOP/s
 *   MV_CAL_READ_LDSDMA the float scan's own transport with the arithmetic removed: non-temporal global_load_lds_dwordx4
 *                     into the 4-slot wave-private ring, four waves per 256 KiB piece     -> *out in GB/s */
enum { MV_CAL_READ_NT = 1, MV_CAL_MFMA_BF16 = 2, MV_CAL_READ_LDSDMA = 3, MV_CAL_MFMA_BF16_32X32 = 4,
       /* the batched FDE coarse pass's access pattern without LDS, barriers or arithmetic: a [rows][20 480 B] matrix read tile by tile,
        * 512 / 1024 / 2048 / 4096 contiguous bytes per row and step (32 KiB per workgroup and step, three steps in flight), or whole rows */
       MV_CAL_READ_STRIDED_512 = 5, MV_CAL_READ_STRIDED_1K = 6, MV_CAL_READ_STRIDED_2K = 7, MV_CAL_READ_STRIDED_4K = 8, MV_CAL_READ_ROWS_20K = 9,
       /* ... and through the pass's own transport (non-temporal global_load_lds_dwordx4 into an LDS ring, nothing read back): 512 B /
        * 1 KiB / 2 KiB per row and step, and 128 B per row with 8 rows per instruction (the private-ring form) */
       MV_CAL_DMA_STRIDED_128 = 10, MV_CAL_DMA_STRIDED_512 = 11, MV_CAL_DMA_STRIDED_1K = 12, MV_CAL_DMA_STRIDED_2K = 13,
       /* the single-query FDE coarse scan itself over bytes / 20 480 synthetic rows, `iters` launches back to back -> GB/s
        * (what the kernel sustains without the host gaps between requests): the register form / the default row-quarter form */
       MV_CAL_FDE_SCAN_REGS = 14, MV_CAL_FDE_SCAN_ROWQ = 15,
       /* the ring transport with the SHAPE of the work taken from MV_PROBE_CT / _OWN / _SCHED / _BPC / _QLOAD (csrc/mv_synth.hip:
        * stream_probe_kernel) -- the experiment behind the row-quarter form */
       MV_CAL_STREAM_PROBE = 19 };
MV_API int mv_calibrate(int device, int what, int64_t bytes, int32_t iters, double* out);

/* ---------------------------------------------------------------------------------------------------------------
 * Row-sharded corpus: the two-stage pipeline (FastMultiVectorStore.query_similar, fast_multivector_store.py:521-556)
 * over R shards with the SAME candidate set, pad lengths and answer as one big index.  Stage entry points for callers
 * that own the collective themselves (one process per GPU over torch.distributed / RCCL: morphik_core_amd/sharded.py);
 * mv_comm below drives them for a single-process multi-GPU store.
 * --------------------------------------------------------------------------------------------------------------- */
/* One coarse candidate: 16 bytes, all-gathered as raw bytes.  Padding entries are (-inf, 0, -1). */
typedef struct {
  float score;   /* FDE coarse score */
  int32_t rows;  /* the page's row count (pad-to-longest needs it on every shard) */
  int64_t id;    /* GLOBAL page id */
} mv_cand_rec;

/* Stage 1 on one shard: coarse scan + local top-n_coarse -> d_out_recs (DEVICE, n_coarse records sorted by (score desc, id asc)).
 *   mode = MV_MODE_FDE_THEN_FLOAT  FDE coarse scan (the reference pipeline)
 *   mode = MV_MODE_FP8_THEN_FLOAT  e4m3 scan of every page (configs[4]); n_coarse = max(MV_OPT_RERANK_N, k)
 * n_coarse <= 1024.  stream as in mv_query_topk_device (NULL = return when finished). */
MV_API int mv_two_stage_coarse_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t n_coarse, int mode,
                                      const uint32_t* allow_bits, int64_t n_allow_words, mv_cand_rec* d_out_recs,
                                      void* stream);
/* The rerank rule of a list of n_list candidates on THIS index (one rule for every entry point, so a request takes the same path
 * whatever carries it): *out_tier = 0 bf16 slab in HBM, 1 pinned-host exact tier, 2 e4m3 slab (no exact copy);
 * *out_n_mid > 0: an e4m3 pruning stage runs first and keeps that many entries (MV_MODE_FDE_THEN_FLOAT over a host tier with
 * n_list > max(MV_OPT_RERANK_N, k)); batched = the one-launch rerank of a request group (its e4m3 form takes <= 64 rows). */
MV_API int mv_index_rerank_plan(mv_index* ix, int mode, int32_t n_list, int32_t k, int32_t n_q_rows, int32_t batched,
                                int32_t* out_n_mid, int32_t* out_tier);
/* Pruning stage on one shard (only when mv_index_rerank_plan names one): d_all_recs as below; leaves in d_out_mid (DEVICE,
 * n_coarse floats) the e4m3 MaxSim of the entries of the GLOBAL coarse list this shard owns, -inf elsewhere.  The caller
 * all-gathers the n_coarse floats of every shard ([world][n_coarse]) and hands them to the rerank stage. */
MV_API int mv_two_stage_mid_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode,
                                   const mv_cand_rec* d_all_recs, int32_t world, int32_t n_coarse, float* d_out_mid, void* stream);
/* Last stage on one shard: d_all_recs = [world][n_coarse] records (the all-gather of stage 1 in shard order, shards
 * owning ascending id ranges).  Derives the GLOBAL coarse top-n_coarse (identical on every shard), keeps the
 * candidates this shard owns (d_all_mid != NULL: only those among the n_mid best of the gathered pruning scores, ties by list
 * position), reranks them on the exact tier (bf16 slab, else pinned-host tier, else e4m3 slab) with the pad length of each
 * candidate's batch of 128 in the GLOBAL list (MV_MODE_FDE_THEN_FLOAT; none for MV_MODE_FP8_THEN_FLOAT), and leaves the local
 * top-k in d_out_scores / d_out_ids (DEVICE, padded (-inf, -1)).
 * No host synchronisation between the steps; world * n_coarse <= 16384. */
MV_API int mv_two_stage_rerank_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode,
                                      const mv_cand_rec* d_all_recs, int32_t world, int32_t n_coarse, const float* d_all_mid,
                                      int32_t n_mid, int32_t k, float* d_out_scores, int64_t* d_out_ids, void* stream);

/* Single-process communicator over R shards (SURVEY.md 8b "mv_comm_init(n_ranks, device_ids)"): the reference builds
 * ONE store object in ONE process (core/services_init.py:141-165) -- a store that owns 8 GPUs needs the top-k exchange
 * behind the C ABI.  Shard i lives on device_ids[i] (several shards may share a device: logical shards) and owns the
 * global ids [id_base_i, id_base_i + size_i), ascending with i.
 *   MV_COMM_RCCL  ncclCommInitAll + one grouped ncclAllGather of the k (score,id) pairs per query (distinct devices only)
 *   MV_COMM_P2P   hipMemcpyPeerAsync of every shard's k pairs into shard 0's device, merged there (mv_merge_topk)
 *   MV_COMM_HOST  per-shard results copied to the host and merged there (the correctness reference)
 *   MV_COMM_AUTO  RCCL when the devices are distinct and librccl loads, else P2P */
typedef struct mv_comm mv_comm;
enum { MV_COMM_AUTO = 0, MV_COMM_RCCL = 1, MV_COMM_P2P = 2, MV_COMM_HOST = 3 };
MV_API int mv_comm_create(int32_t n_shards, const int32_t* device_ids, int32_t transport, mv_comm** out);
MV_API void mv_comm_destroy(mv_comm* c);
MV_API int mv_comm_attach(mv_comm* c, int32_t shard, mv_index* ix);
MV_API int mv_comm_transport(const mv_comm* c); /* the transport in use (MV_COMM_RCCL / P2P / HOST) */
/* Query every shard (scans run concurrently, one stream per shard), exchange, merge: same results, order and tie
 * rule as ONE index holding all pages.  MV_MODE_FDE_THEN_FLOAT and MV_MODE_FP8_THEN_FLOAT run the staged pipeline above (the
 * candidate list is the GLOBAL one; every shard must keep the same slabs and rerank options).
 * stats (nullable): n_shards entries, per-shard device timings of this query. */
MV_API int mv_comm_query_topk(mv_comm* c, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                              const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids,
                              int32_t* out_n, mv_query_stats* stats);

/* A batch of requests against the sharded corpus (the store's request coalescer on a sharded store; every request
 * awaits the same store object: core/services/document_service.py:411-417).  Arguments as mv_query_topk_batch.
 * MV_MODE_FDE_THEN_FLOAT / MV_MODE_FP8_THEN_FLOAT: per group of requests (<= 32; <= 1024 / 512 query rows) ONE pass over every
 * shard's FDE / e4m3 slab, ONE exchange of all the requests' candidate records (+ one of their pruning scores when the rerank
 * plan has that stage), every request's share of its GLOBAL candidate list reranked in one launch per shard -- the same
 * candidate sets, pad lengths and answers as mv_query_topk_batch on one index.  Other modes run request by request through
 * mv_comm_query_topk.  stats (nullable): n_shards entries, device spans summed over the groups. */
MV_API int mv_comm_query_topk_batch(mv_comm* c, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k,
                                    int mode, const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query,
                                    float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats);

/* Persistence ("checkpoint" of the HBM index): raw slabs + metadata in one file (written to <path>.tmp, fsync'ed and
 * renamed: a crash mid-save keeps the previous checkpoint). */
/* ... and with the caller's own query FDE vectors (mv_query_topk_fde above): every shard's coarse stage uses them */
MV_API int mv_comm_query_topk_fde(mv_comm* c, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t k, int mode,
                                  const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n,
                                  mv_query_stats* stats);
MV_API int mv_comm_query_topk_batch_fde(mv_comm* c, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, const float* q_fde,
                                        int32_t k, int mode, const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query,
                                        float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats);
MV_API int mv_two_stage_coarse_device_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t n_coarse,
                                          int mode, const uint32_t* allow_bits, int64_t n_allow_words, mv_cand_rec* d_out_recs, void* stream);

MV_API int mv_index_save(mv_index* ix, const char* path);
MV_API int mv_index_load(const char* path, int32_t device, mv_index** out);

#ifdef __cplusplus
}
#endif
#endif /* MVMAXSIM_H */
