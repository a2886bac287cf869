// mv_comm.hip -- row-sharded corpus behind the C ABI.
//
//  (1) mv_two_stage_coarse_device / mv_two_stage_rerank_device: the stages of FastMultiVectorStore.query_similar
//      (core/vector_store/fast_multivector_store.py:521-556: FDE coarse search -> candidates -> exact rerank -> top-k)
//      on ONE shard of a row-sharded corpus, with every intermediate left on the device, so a caller that owns the
//      collective (one process per GPU over RCCL: morphik_core_amd/sharded.py) strings them together without a host
//      round trip.
//  (2) mv_comm: R shards driven from ONE process -- the shape the reference wires its store in
//      (core/services_init.py:141-165 builds one store object; SURVEY.md 8b proposed mv_comm_init(n_ranks, device_ids)).
//      Scans of all shards are enqueued back to back on per-shard streams and run concurrently on their GPUs; the only
//      exchange is k (score, id) pairs per shard (plus n_coarse 16-byte candidate records for the two-stage mode), moved
//      by one grouped RCCL all-gather over xGMI, by peer copies, or through the host (the correctness reference).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy torch ships resolves to the same soname): processes that
// never create a multi-device communicator do not load it.
#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only; the symbols are resolved with dlsym

#include "mv_index_priv.h"

using namespace mv;

namespace {

// ---------------------------------------------------------------- two-stage device kernels
__global__ __launch_bounds__(256) void recs_build_kernel(const float* s, const int64_t* gid, int n, const int32_t* n_rows,
                                                         int32_t stride, int64_t id_base, mv_cand_rec* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  mv_cand_rec r;
  const int64_t g = gid[i];
  if (g < 0) {
    r.score = -INFINITY; r.rows = 0; r.id = -1;
  } else {
    r.score = s[i];
    r.rows = n_rows ? n_rows[g - id_base] : stride;
    r.id = g;
  }
  out[i] = r;
}

__global__ __launch_bounds__(256) void recs_fill_pad_kernel(mv_cand_rec* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = mv_cand_rec{-INFINITY, 0, -1};
}

__global__ __launch_bounds__(256) void recs_scores_kernel(const mv_cand_rec* recs, int n, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = recs[i].id < 0 ? -INFINITY : recs[i].score;
}

// Global coarse top-n (positions into the gathered records, in coarse rank order) -> this shard's rerank list: local page
// of the candidates it owns, -1 for the others (the rerank kernel skips them), and for every entry the pad length of
// its batch of 128 in the GLOBAL list (longest page of the batch, owned or not).  One block per batch.
__global__ __launch_bounds__(kRerankBatch) void owned_select_kernel(const mv_cand_rec* recs, const int64_t* pos, int n, int64_t lo,
                                                                    int64_t hi, int pad_sem, int32_t* cand, int32_t* pads) {
  __shared__ int32_t wmax[kRerankBatch / 64];
  const int j = blockIdx.x * kRerankBatch + threadIdx.x;
  int32_t c = -1, rows = 0;
  if (j < n) {
    const int64_t p = pos[j];
    if (p >= 0) {
      const mv_cand_rec r = recs[p];
      rows = r.rows;
      if (r.id >= lo && r.id < hi) c = (int32_t)(r.id - lo);
    }
  }
  int32_t m = rows;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) m = max(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(wmax[0], wmax[1]);
  if (j < n) {
    cand[j] = c;
    pads[j] = pad_sem ? m : 0;
  }
}

__global__ void fill_topk_pad_kernel(float* s, int64_t* id, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) { s[i] = -INFINITY; id[i] = -1; }
}

// ---- the same stages for a GROUP of requests (mv_comm_query_topk_batch): grid.y = request
// coarse top-n of request b (scores / GLOBAL ids at b * list_stride) -> records out[b][n]
__global__ __launch_bounds__(256) void recs_build_batch_kernel(const float* s, const int64_t* gid, int n, int64_t list_stride, const int32_t* n_rows,
                                                               int32_t stride, int64_t id_base, mv_cand_rec* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  mv_cand_rec r;
  const int64_t g = gid[(int64_t)b * list_stride + i];
  if (g < 0) {
    r.score = -INFINITY; r.rows = 0; r.id = -1;
  } else {
    r.score = s[(int64_t)b * list_stride + i];
    r.rows = n_rows ? n_rows[g - id_base] : stride;
    r.id = g;
  }
  out[(int64_t)b * n + i] = r;
}

// gathered records [world][nb][n] -> per-request score rows out[b][r * n + i] (shard order: ties by position = by ascending id)
__global__ __launch_bounds__(256) void recs_scores_batch_kernel(const mv_cand_rec* recs, int world, int nb, int n, float* out) {
  const int j = blockIdx.x * 256 + threadIdx.x;  // r * n + i
  const int b = blockIdx.y;
  if (j >= world * n) return;
  const int r = j / n, i = j - r * n;
  const mv_cand_rec x = recs[((int64_t)r * nb + b) * n + i];
  out[(int64_t)b * world * n + j] = x.id < 0 ? -INFINITY : x.score;
}

// request b: positions of its GLOBAL coarse top-n (pos[b][j] = r * n + i) -> this shard's rerank list + per-batch pad lengths
__global__ __launch_bounds__(kRerankBatch) void owned_select_batch_kernel(const mv_cand_rec* recs, const int64_t* pos, int n, int nb, int64_t lo,
                                                                          int64_t hi, int pad_sem, int32_t* cand, int32_t* pads) {
  __shared__ int32_t wmax[kRerankBatch / 64];
  const int j = blockIdx.x * kRerankBatch + threadIdx.x;
  const int b = blockIdx.y;
  int32_t c = -1, rows = 0;
  if (j < n) {
    const int64_t p = pos[(int64_t)b * n + j];
    if (p >= 0) {
      const int r = (int)(p / n), i = (int)(p - (int64_t)r * n);
      const mv_cand_rec x = recs[((int64_t)r * nb + b) * n + i];
      rows = x.rows;
      if (x.id >= lo && x.id < hi) c = (int32_t)(x.id - lo);
    }
  }
  int32_t m = rows;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) m = max(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(wmax[0], wmax[1]);
  if (j < n) {
    cand[(int64_t)b * n + j] = c;
    pads[(int64_t)b * n + j] = pad_sem ? m : 0;
  }
}

int order_behind(mv_index* ix, void* user_stream) {
  if (!user_stream) return MV_OK;
  MV_HIP(hipEventRecord(ix->ev[3], (hipStream_t)user_stream));
  MV_HIP(hipStreamWaitEvent(ix->stream, ix->ev[3], 0));
  return MV_OK;
}
int hand_back(mv_index* ix, void* user_stream) {
  if (user_stream) {
    MV_HIP(hipEventRecord(ix->ev[4], ix->stream));
    MV_HIP(hipStreamWaitEvent((hipStream_t)user_stream, ix->ev[4], 0));
  } else {
    MV_HIP(hipStreamSynchronize(ix->stream));
  }
  return MV_OK;
}

}  // namespace

extern "C" {

// ---- shared pieces of the stages (q_mu held, device current)
static bool two_stage_mode_ok(int mode) { return mode == MV_MODE_FDE_THEN_FLOAT || mode == MV_MODE_FP8_THEN_FLOAT; }
// pad rule of the rerank: the reference's batches of 128 for the FDE pipeline (pad_sequence, fast_multivector_store.py:553-555);
// the candidates of an e4m3 scan are scored like a full scan (a page's own rows only)
static int two_stage_pad_sem(const mv_index* ix, int mode) { return mode == MV_MODE_FDE_THEN_FLOAT ? (ix->pad_semantics < 0 ? 1 : ix->pad_semantics) : 0; }

static int ensure_gscores(mv_index* ix, int64_t need) {
  if (need <= ix->gscores_cap) return MV_OK;
  if (ix->d_gscores) (void)hipFree(ix->d_gscores);
  ix->d_gscores = nullptr; ix->gscores_cap = 0;
  const int64_t cap = need <= 16384 ? 16384 : (int64_t)kFdeBatchMaxQueries * 16384;
  MV_HIP(hipMalloc(&ix->d_gscores, (size_t)cap * 4));
  ix->gscores_cap = cap;
  return MV_OK;
}

// gathered records -> GLOBAL top-n_coarse (positions, identical on every shard) -> this shard's rerank list d_cand / d_cand_pads
static int global_owned_list(mv_index* ix, int mode, const mv_cand_rec* d_all_recs, int world, int n_coarse) {
  const int64_t n = ix->size.load(std::memory_order_acquire);
  const int total = world * n_coarse;
  int rc = ensure_gscores(ix, total);
  if (rc) return rc;
  // the gathered lists are in shard order and each is sorted (score desc, id asc), shards own ascending ids, so
  // "ties by position" is "ties by ascending id" -- the single-index rule
  hipLaunchKernelGGL(recs_scores_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ix->stream, d_all_recs, total, ix->d_gscores);
  rc = launch_topk(ix->d_gscores, total, n_coarse, nullptr, 0, ix->d_topk_ws, ix->d_out_s, ix->d_sel_pos, ix->stream);
  if (rc) return rc;
  hipLaunchKernelGGL(owned_select_kernel, dim3((unsigned)((n_coarse + kRerankBatch - 1) / kRerankBatch)), dim3(kRerankBatch), 0, ix->stream,
                     d_all_recs, (const int64_t*)ix->d_sel_pos, (int)n_coarse, ix->cfg.id_base, ix->cfg.id_base + n, two_stage_pad_sem(ix, mode), ix->d_cand,
                     ix->d_cand_pads);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// out[j] = max over the shards of all_mid[r][j]: every list position is owned by exactly one shard, the others hold -inf
__global__ __launch_bounds__(256) void mid_combine_kernel(const float* all_mid, int world, int64_t n, float* out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float m = -INFINITY;
  for (int r = 0; r < world; ++r) m = fmaxf(m, all_mid[(int64_t)r * n + j]);
  out[j] = m;
}

int mv_two_stage_coarse_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t n_coarse, int mode,
                               const uint32_t* allow_bits, int64_t n_allow_words, mv_cand_rec* d_out_recs, void* stream) {
  if (!ix || !q || !d_out_recs || n_q_rows < 1 || n_coarse < 1 || n_coarse > kTopkMaxDeviceK) { set_error("two_stage_coarse: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (!two_stage_mode_ok(mode)) { set_error("two_stage_coarse: mode %d is not a two-stage mode (MV_MODE_FDE_THEN_FLOAT / MV_MODE_FP8_THEN_FLOAT)", mode); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_q_rows * kDim, mode)) return frc;
  const bool fde = mode == MV_MODE_FDE_THEN_FLOAT;
  if (fde && !(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }
  if (!fde && !(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has no fp8 slab (MV_WITH_FP8)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  const int64_t n = ix->size.load(std::memory_order_acquire);
  const unsigned gb = (unsigned)((n_coarse + 255) / 256);
  if (n == 0) {
    hipLaunchKernelGGL(recs_fill_pad_kernel, dim3(gb), dim3(256), 0, ix->stream, d_out_recs, (int)n_coarse);
    MV_HIP(hipGetLastError());
    return hand_back(ix, stream);
  }
  rc = upload_query(ix, q, q_dtype, n_q_rows, false, fde, false, !fde);
  if (rc) return rc;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, n_allow_words, &d_allow);
  if (rc) return rc;
  int launches = 0;
  bool hist0_done = false;
  if (fde) rc = fde_coarse_scan(ix, n_q_rows, d_allow, n_allow_words, n, &launches, false, n_coarse, &hist0_done);
  else rc = fp8_scan(ix, n_q_rows, d_allow, n_allow_words, nullptr, n, 0, nullptr, ix->d_scores, &launches);
  if (rc) return rc;
  rc = launch_topk(ix->d_scores, n, n_coarse, nullptr, ix->cfg.id_base, ix->d_topk_ws, ix->d_out_s, ix->d_out_id, ix->stream, hist0_done);
  if (rc) return rc;
  hipLaunchKernelGGL(recs_build_kernel, dim3(gb), dim3(256), 0, ix->stream, (const float*)ix->d_out_s, (const int64_t*)ix->d_out_id,
                     (int)n_coarse, ix->ragged.load() ? (const int32_t*)ix->d_n_rows : (const int32_t*)nullptr, ix->cfg.stride_rows,
                     ix->cfg.id_base, d_out_recs);
  MV_HIP(hipGetLastError());
  return hand_back(ix, stream);
}

int mv_index_rerank_plan(mv_index* ix, int mode, int32_t n_list, int32_t k, int32_t n_q_rows, int32_t batched, int32_t* out_n_mid, int32_t* out_tier) {
  if (!ix || n_list < 0 || k < 0 || n_q_rows < 1) { set_error("rerank_plan: bad argument"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  const RerankPlan p = rerank_plan(ix, mode, n_list, k, ((n_q_rows + 15) / 16) * 16, batched != 0);
  if (out_n_mid) *out_n_mid = p.mid ? p.n_mid : 0;
  if (out_tier) *out_tier = p.final_fp8 ? 2 : (p.host_tier ? 1 : 0);
  return MV_OK;
}

int mv_two_stage_mid_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode, const mv_cand_rec* d_all_recs,
                            int32_t world, int32_t n_coarse, float* d_out_mid, void* stream) {
  if (!ix || !q || !d_all_recs || !d_out_mid || n_q_rows < 1 || world < 1 || n_coarse < 1 || n_coarse > kTopkMaxDeviceK ||
      (int64_t)world * n_coarse > 16384) { set_error("two_stage_mid: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (mode != MV_MODE_FDE_THEN_FLOAT) { set_error("two_stage_mid: only MV_MODE_FDE_THEN_FLOAT has a pruning stage"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has no fp8 slab (MV_WITH_FP8)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  rc = upload_query(ix, q, q_dtype, n_q_rows, false, false, false, true);
  if (rc) return rc;
  rc = global_owned_list(ix, mode, d_all_recs, world, n_coarse);
  if (rc) return rc;
  int launches = 0;
  rc = rerank_scan(ix, n_q_rows, kTierFp8, n_coarse, d_out_mid, &launches);  // -inf at the positions other shards own
  if (rc) return rc;
  return hand_back(ix, stream);
}

int mv_two_stage_rerank_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode, const mv_cand_rec* d_all_recs,
                               int32_t world, int32_t n_coarse, const float* d_all_mid, int32_t n_mid, int32_t k, float* d_out_scores,
                               int64_t* d_out_ids, void* stream) {
  if (!ix || !q || !d_all_recs || !d_out_scores || !d_out_ids || n_q_rows < 1 || world < 1 || n_coarse < 1 || n_coarse > kTopkMaxDeviceK ||
      k < 1 || k > kTopkMaxDeviceK || (int64_t)world * n_coarse > 16384 || (d_all_mid && (n_mid < 1 || n_mid > kTopkMaxDeviceK))) { set_error("two_stage_rerank: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (!two_stage_mode_ok(mode)) { set_error("two_stage_rerank: mode %d is not a two-stage mode", mode); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  const RerankPlan plan = rerank_plan(ix, mode, n_coarse, k, ((n_q_rows + 15) / 16) * 16, false);
  const bool use_fp8 = plan.final_fp8;
  if (use_fp8 && mode == MV_MODE_FP8_THEN_FLOAT) { set_error("MV_MODE_FP8_THEN_FLOAT needs an exact tier (MV_WITH_FLOAT or MV_WITH_HOST_EXACT)"); return MV_ERR_STATE; }
  if (use_fp8 && !(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has neither an exact tier nor an fp8 slab"); return MV_ERR_STATE; }
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  rc = upload_query(ix, q, q_dtype, n_q_rows, !use_fp8, false, false, use_fp8);
  if (rc) return rc;
  rc = global_owned_list(ix, mode, d_all_recs, world, n_coarse);
  if (rc) return rc;
  if (d_all_mid) {
    // the e4m3 scores of the GLOBAL list, one owner per position -> its n_mid best positions (ties by list position, as on one
    // index) -> the owned entries outside them leave this shard's rerank list
    rc = ensure_gscores(ix, n_coarse);
    if (rc) return rc;
    hipLaunchKernelGGL(mid_combine_kernel, dim3((unsigned)((n_coarse + 255) / 256)), dim3(256), 0, ix->stream, d_all_mid, (int)world, (int64_t)n_coarse, ix->d_gscores);
    rc = launch_topk(ix->d_gscores, n_coarse, std::min(n_mid, n_coarse), nullptr, 0, ix->d_topk_ws, ix->d_out_s, ix->d_sel_pos, ix->stream);
    if (rc) return rc;
    rc = launch_keep_selected(ix->d_sel_pos, 0, std::min(n_mid, n_coarse), ix->d_cand, 0, n_coarse, 1, ix->stream);
    if (rc) return rc;
  }
  int launches = 0;
  rc = rerank_scan(ix, n_q_rows, plan.tier, n_coarse, ix->d_cand_scores, &launches);
  if (rc) return rc;
  // local top-k of the owned candidates; equal scores resolve by coarse rank (the work index), as on one index
  rc = launch_topk(ix->d_cand_scores, n_coarse, k, ix->d_cand, ix->cfg.id_base, ix->d_topk_ws, d_out_scores, d_out_ids, ix->stream);
  if (rc) return rc;
  return hand_back(ix, stream);
}

// The stages for a group of nb requests (<= 32; FDE: nb * padded rows <= 1024, e4m3 scan: <= 512): ONE pass over the shard's
// FDE (or e4m3) slab for all of them (the batched kernels of mv_query_topk_batch), then every request's share of ITS global
// candidate list reranked in one launch.  d_out_recs: [nb][n_coarse]; d_all_recs: [world][nb][n_coarse]; d_out_*: [nb][k].
int mv_internal_two_stage_batch_coarse(mv_index* ix, const void* q, int q_dtype, int32_t nb, int32_t n_q_rows, int32_t n_coarse, int mode,
                                       const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, mv_cand_rec* d_out_recs,
                                       void* stream) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  const int64_t n = ix->size.load(std::memory_order_acquire);
  if (n == 0) {
    hipLaunchKernelGGL(recs_fill_pad_kernel, dim3((unsigned)((nb * n_coarse + 255) / 256)), dim3(256), 0, ix->stream, d_out_recs, (int)(nb * n_coarse));
    MV_HIP(hipGetLastError());
    return hand_back(ix, stream);
  }
  const bool fde = mode == MV_MODE_FDE_THEN_FLOAT;
  rc = fde ? mv_internal_ensure_fde_batch_ws(ix) : mv_internal_ensure_fp8_batch_ws(ix);
  if (rc) return rc;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  if (!fde) MV_HIP(hipMemsetAsync(ix->d_bqf32, 0, (size_t)512 * kDim * 4, ix->stream));  // zero rows behind the group: the kernel's row tiles run to a multiple of 64
  rc = mv_internal_batch_upload_queries(ix, q, q_dtype, nb, n_q_rows, true, false, false);
  if (rc) return rc;
  const bool per_query = allow_bits && allow_per_query;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, per_query ? n_allow_words * (int64_t)nb : n_allow_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  const int64_t cap = ix->bscore_stride;  // elements between two requests' score vectors
  bool prebinned = false;
  if (fde) {
    FdeEncodeArgs e{};
    e.variant = 2;
    e.x_f32 = ix->d_bqf32; e.row_offsets = nullptr; e.stride = rpq; e.n_pages = nb; e.is_query = 1; e.out_f32 = ix->d_bqfde;
    if (const float* ov = query_fde_override(ix->fde_t.out_dim))  // mv_comm_query_topk_batch_fde: the caller's encodings of this group's queries
      MV_HIP(hipMemcpyAsync(ix->d_bqfde, ov, (size_t)nb * ix->fde_t.out_dim * 4, hipMemcpyHostToDevice, ix->stream));
    else
      rc = launch_fde_encode(ix->fde_t, e, ix->stream);
    if (rc) return rc;
    FdeScanBatchArgs sa{};
    sa.fde = ix->fde; sa.inv_norm = ix->fde_cosine ? ix->fde_inv_norm : nullptr; sa.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    sa.allow = d_allow; sa.n_allow_bits = n_allow_words * 32; sa.allow_stride_bits = per_query ? n_allow_words * 32 : 0;
    sa.q = ix->d_bqfde; sa.image = ix->d_bqimage; sa.scores = ix->d_bscores; sa.score_stride = cap; sa.n = n; sa.out_dim = ix->fde_t.out_dim; sa.n_queries = nb;
    sa.hi_only = ix->fde_batch_variant == 2;
    sa.single_tile = ix->fde_batch_variant == 3;
    sa.separate_finish = ix->fde_batch_variant == 5;
    mv_internal_fde_batch_e4m3_args(ix, &sa);
    if (ix->fde_batch_variant != 5 && topk_uses_radix(n, n_coarse)) {  // the finish pass bins the scores for the selection (mv_api.hip)
      sa.hist0 = topk_radix_hist0(ix->d_btopk_ws);
      sa.hist0_stride_bytes = (int64_t)ix->topk_ws_bytes;
    }
    prebinned = fde_scan_batch_prebins(sa);
    rc = launch_fde_scan_batch(sa, ix->stream);
    if (rc) return rc;
  } else {
    rc = launch_fp8_query_prep(ix->d_bqf32, 512, ix->d_bq8hi, ix->d_bq8lo, ix->d_bq8fac, ix->stream);
    if (rc) return rc;
    Fp8BatchArgs a{};
    a.slab = ix->slab8; a.inv_scale = ix->inv_scale8; a.n_rows = ix->ragged.load() ? ix->d_n_rows : nullptr; a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    a.allow = d_allow; a.n_allow_bits = n_allow_words * 32; a.allow_stride_bits = per_query ? n_allow_words * 32 : 0;
    a.qhi = ix->d_bq8hi; a.qlo = ix->d_bq8lo; a.qfac = ix->d_bq8fac; a.scores = ix->d_bscores; a.n = n; a.score_stride = cap;
    a.stride = ix->cfg.stride_rows; a.n_queries = nb; a.rows_per_query = rpq; a.row_off = ix->d_row_off;
    a.single_term = ix->batch_variant == 0 ? 0 : 1;  // the first stage of a two-tier search: one e4m3 term per query row unless asked otherwise (fp8_batch_query)
    rc = launch_maxsim_batch_fp8(a, ix->stream);
    if (rc) return rc;
  }
  // local coarse top-n of every request, GLOBAL ids, padded with (-inf, -1) when the shard holds fewer pages
  rc = launch_topk_batch(ix->d_bscores, cap, n, n_coarse, nullptr, 0, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s, ix->d_bsel_id,
                         n_coarse, nb, ix->stream, prebinned);
  if (rc) return rc;
  hipLaunchKernelGGL(recs_build_batch_kernel, dim3((unsigned)((n_coarse + 255) / 256), (unsigned)nb), dim3(256), 0, ix->stream, (const float*)ix->d_bsel_s,
                     (const int64_t*)ix->d_bsel_id, (int)n_coarse, (int64_t)n_coarse, ix->ragged.load() ? (const int32_t*)ix->d_n_rows : (const int32_t*)nullptr,
                     ix->cfg.stride_rows, ix->cfg.id_base, d_out_recs);
  MV_HIP(hipGetLastError());
  return hand_back(ix, stream);
}

// gathered records of the group -> every request's GLOBAL top-n -> this shard's lists d_bcand / d_bcand_pads ([nb][n_coarse])
static int global_owned_lists_batch(mv_index* ix, int mode, int nb, const mv_cand_rec* d_all_recs, int world, int n_coarse) {
  const int64_t n = ix->size.load(std::memory_order_acquire);
  const int total = world * n_coarse;
  int rc = ensure_gscores(ix, (int64_t)nb * total);  // [nb][world * n_coarse] gathered coarse scores
  if (rc) return rc;
  hipLaunchKernelGGL(recs_scores_batch_kernel, dim3((unsigned)((total + 255) / 256), (unsigned)nb), dim3(256), 0, ix->stream, d_all_recs, (int)world, (int)nb,
                     (int)n_coarse, ix->d_gscores);
  MV_HIP(hipGetLastError());
  // GLOBAL coarse top-n of every request: positions into its gathered row (ids map = none, id_base 0)
  rc = launch_topk_batch(ix->d_gscores, total, total, n_coarse, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s, ix->d_bsel_id, n_coarse, nb,
                         ix->stream);
  if (rc) return rc;
  hipLaunchKernelGGL(owned_select_batch_kernel, dim3((unsigned)((n_coarse + kRerankBatch - 1) / kRerankBatch), (unsigned)nb), dim3(kRerankBatch), 0, ix->stream,
                     d_all_recs, (const int64_t*)ix->d_bsel_id, (int)n_coarse, (int)nb, ix->cfg.id_base, ix->cfg.id_base + n, two_stage_pad_sem(ix, mode), ix->d_bcand,
                     ix->d_bcand_pads);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// the pruning stage of a group: e4m3 scores of every request's owned candidates -> d_out_mid [nb][n_coarse] (-inf elsewhere)
int mv_internal_two_stage_batch_mid(mv_index* ix, const void* q, int q_dtype, int32_t nb, int32_t n_q_rows, int mode, const mv_cand_rec* d_all_recs,
                                    int32_t world, int32_t n_coarse, float* d_out_mid, void* stream) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  rc = mv_internal_ensure_fde_batch_ws(ix);
  if (rc) return rc;
  rc = mv_internal_batch_upload_queries(ix, q, q_dtype, nb, n_q_rows, false, false, true);
  if (rc) return rc;
  rc = global_owned_lists_batch(ix, mode, nb, d_all_recs, world, n_coarse);
  if (rc) return rc;
  int launches = 0;
  rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, n_coarse, &launches, kTierFp8, d_out_mid);
  if (rc) return rc;
  return hand_back(ix, stream);
}

int mv_internal_two_stage_batch_rerank(mv_index* ix, const void* q, int q_dtype, int32_t nb, int32_t n_q_rows, int mode, const mv_cand_rec* d_all_recs,
                                       int32_t world, int32_t n_coarse, const float* d_all_mid, int32_t n_mid, int32_t k, float* d_out_scores,
                                       int64_t* d_out_ids, void* stream) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const RerankPlan plan = rerank_plan(ix, mode, n_coarse, k, rpq, true);
  const bool use_fp8 = plan.final_fp8;
  if (use_fp8 && mode == MV_MODE_FP8_THEN_FLOAT) { set_error("MV_MODE_FP8_THEN_FLOAT needs an exact tier (MV_WITH_FLOAT or MV_WITH_HOST_EXACT)"); return MV_ERR_STATE; }
  if (use_fp8 && !(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has neither an exact tier nor an fp8 slab"); return MV_ERR_STATE; }
  DeviceGuard g(ix->cfg.device);
  int rc = order_behind(ix, stream);
  if (rc) return rc;
  rc = mode == MV_MODE_FDE_THEN_FLOAT ? mv_internal_ensure_fde_batch_ws(ix) : mv_internal_ensure_fp8_batch_ws(ix);
  if (rc) return rc;
  // the group's queries again (another caller may have used the workspace between the stages)
  rc = mv_internal_batch_upload_queries(ix, q, q_dtype, nb, n_q_rows, false, !use_fp8, use_fp8);
  if (rc) return rc;
  rc = global_owned_lists_batch(ix, mode, nb, d_all_recs, world, n_coarse);
  if (rc) return rc;
  if (d_all_mid) {  // [world][nb][n_coarse] e4m3 scores -> every request's n_mid best list positions -> the rest of its list becomes -1
    const int64_t tot = (int64_t)nb * n_coarse;
    rc = ensure_gscores(ix, tot);
    if (rc) return rc;
    hipLaunchKernelGGL(mid_combine_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ix->stream, d_all_mid, (int)world, tot, ix->d_gscores);
    const int nm = std::min(n_mid, n_coarse);
    rc = launch_topk_batch(ix->d_gscores, n_coarse, n_coarse, nm, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s, ix->d_bsel_id, n_coarse, nb, ix->stream);
    if (rc) return rc;
    rc = launch_keep_selected(ix->d_bsel_id, n_coarse, nm, ix->d_bcand, n_coarse, n_coarse, nb, ix->stream);
    if (rc) return rc;
  }
  int launches = 0;
  rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, n_coarse, &launches, plan.tier, nullptr);
  if (rc) return rc;
  // local top-k of the owned candidates of every request; equal scores resolve by coarse rank, as on one index
  rc = launch_topk_batch(ix->d_bcand_scores, n_coarse, n_coarse, k, ix->d_bcand, n_coarse, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, d_out_scores,
                         d_out_ids, k, nb, ix->stream);
  if (rc) return rc;
  return hand_back(ix, stream);
}

}  // extern "C"

// =================================================================================== communicator
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString;
  }
};

struct Shard {
  int dev = 0;
  mv_index* ix = nullptr;
  hipStream_t cs = nullptr;     // comm stream on the shard's device
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_done = nullptr;
  float* d_ls = nullptr;        // [1024] local top-k scores
  int64_t* d_li = nullptr;      // [1024]
  float* d_gs = nullptr;        // [R][1024] gathered scores (shard 0 always; every shard under RCCL)
  int64_t* d_gi = nullptr;
  mv_cand_rec* d_recs = nullptr;  // [1024] local coarse candidates
  mv_cand_rec* d_all = nullptr;   // [R][1024] gathered coarse candidates
  float* d_mid = nullptr;         // [1024] e4m3 scores of the owned entries of the global list (pruning stage), -inf elsewhere
  float* d_mid_all = nullptr;     // [R][1024]
  // batched two-stage queries (allocated on first use): <= 32 requests per group
  mv_cand_rec* d_brecs = nullptr;  // [32][1024]
  mv_cand_rec* d_ball = nullptr;   // [R][32][1024]
  float* d_bls = nullptr;          // [32][1024] local top-k of every request
  int64_t* d_bli = nullptr;
  float* d_bmid = nullptr;         // [32][1024] pruning-stage scores of a request group
  float* d_bmid_all = nullptr;     // [R][32][1024]
  ncclComm_t nccl = nullptr;
};

}  // namespace

struct mv_comm {
  int n = 0;
  int transport = MV_COMM_HOST;
  std::vector<Shard> sh;
  RcclApi rccl;
  float* d_os = nullptr;   // merged result on shard 0's device
  int64_t* d_oi = nullptr;
  float* h_s = nullptr;    // pinned host: [R][1024] + merged
  int64_t* h_i = nullptr;
  mv_cand_rec* h_recs = nullptr;  // pinned host: [R][1024] (HOST transport of the two-stage mode)
  float* h_bs = nullptr;          // pinned host: [R][32][1024] local top-k lists of a request group (batched queries)
  int64_t* h_bi = nullptr;
  mv_cand_rec* h_brecs = nullptr; // pinned host: [R][32][1024] (HOST transport)
  float* h_mid = nullptr;         // pinned host: [R][32][1024] pruning-stage scores (HOST transport)
  std::mutex mu;
};

namespace {

constexpr int kK = kTopkMaxDeviceK;

#define MV_NCCL(c, expr)                                                                  \
  do {                                                                                    \
    ncclResult_t _r = (expr);                                                             \
    if (_r != ncclSuccess) { set_error("RCCL error %d (%s) in %s", (int)_r, (c)->rccl.GetErrorString(_r), #expr); return MV_ERR_HIP; } \
  } while (0)

int copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t s) {
  if (bytes == 0) return MV_OK;
  if (dst_dev == src_dev) MV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
  else MV_HIP(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
  return MV_OK;
}

// Host merge of R sorted lists (score desc; ties: shard asc, position asc == id asc): the correctness reference.
void host_merge(const float* s, const int64_t* id, int R, int kk, int k, float* os, int64_t* oi, int* out_n) {
  std::vector<int> head((size_t)R, 0);
  int m = 0;
  while (m < k) {
    int best = -1;
    for (int r = 0; r < R; ++r) {
      if (head[r] >= kk) continue;
      const int64_t gid = id[(size_t)r * kk + head[r]];
      if (gid < 0) { head[r] = kk; continue; }
      if (best < 0 || s[(size_t)r * kk + head[r]] > s[(size_t)best * kk + head[best]]) best = r;
    }
    if (best < 0) break;
    os[m] = s[(size_t)best * kk + head[best]];
    oi[m] = id[(size_t)best * kk + head[best]];
    ++head[best];
    ++m;
  }
  *out_n = m;
}

// All-gather of `bytes` per shard from src(i) into dst(j)[i * bytes] for the shards that need the result.
// all = every shard receives everything (two-stage candidates); else only shard 0 must (the final top-k).
int exchange(mv_comm* c, size_t bytes, bool all, void* (*src)(Shard&), void* (*dst)(Shard&)) {
  const int R = c->n;
  if (c->transport == MV_COMM_RCCL) {
    MV_NCCL(c, c->rccl.GroupStart());
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      ncclResult_t r = c->rccl.AllGather(src(s), dst(s), bytes, ncclChar, s.nccl, s.cs);
      if (r != ncclSuccess) { (void)c->rccl.GroupEnd(); set_error("RCCL error %d (%s) in ncclAllGather", (int)r, c->rccl.GetErrorString(r)); return MV_ERR_HIP; }
    }
    MV_NCCL(c, c->rccl.GroupEnd());
    return MV_OK;
  }
  // P2P: every shard pushes its piece into shard 0's buffer on its own stream; shard 0 waits for all of them
  Shard& z = c->sh[0];
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    DeviceGuard g(s.dev);
    int rc = copy_between((char*)dst(z) + (size_t)i * bytes, z.dev, src(s), s.dev, bytes, s.cs);
    if (rc) return rc;
    if (i) MV_HIP(hipEventRecord(s.ev_done, s.cs));
  }
  {
    DeviceGuard g(z.dev);
    for (int i = 1; i < R; ++i) MV_HIP(hipStreamWaitEvent(z.cs, c->sh[i].ev_done, 0));
    if (all && R > 1) MV_HIP(hipEventRecord(z.ev_done, z.cs));
  }
  if (all) {  // ... and shard 0 hands the assembled block to the others
    for (int j = 1; j < R; ++j) {
      Shard& s = c->sh[j];
      DeviceGuard g(s.dev);
      MV_HIP(hipStreamWaitEvent(s.cs, z.ev_done, 0));
      int rc = copy_between(dst(s), s.dev, dst(z), z.dev, bytes * (size_t)R, s.cs);
      if (rc) return rc;
    }
  }
  return MV_OK;
}

void* src_ls(Shard& s) { return s.d_ls; }
void* src_li(Shard& s) { return s.d_li; }
void* dst_gs(Shard& s) { return s.d_gs; }
void* dst_gi(Shard& s) { return s.d_gi; }
void* src_recs(Shard& s) { return s.d_recs; }
void* dst_all(Shard& s) { return s.d_all; }
void* src_brecs(Shard& s) { return s.d_brecs; }
void* dst_ball(Shard& s) { return s.d_ball; }
void* src_mid(Shard& s) { return s.d_mid; }
void* dst_mid_all(Shard& s) { return s.d_mid_all; }
void* src_bmid(Shard& s) { return s.d_bmid; }
void* dst_bmid_all(Shard& s) { return s.d_bmid_all; }

// every shard's `bytes` through the host: D2H into h[i], wait, H2D of the assembled block to every shard (the reference transport)
int exchange_host(mv_comm* c, size_t bytes, void* h, void* (*src)(Shard&), void* (*dst)(Shard&));

int sync_all(mv_comm* c) {
  for (int i = 0; i < c->n; ++i) {
    DeviceGuard g(c->sh[i].dev);
    MV_HIP(hipStreamSynchronize(c->sh[i].cs));
  }
  return MV_OK;
}

int exchange_host(mv_comm* c, size_t bytes, void* h, void* (*src)(Shard&), void* (*dst)(Shard&)) {
  const int R = c->n;
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    DeviceGuard g(s.dev);
    MV_HIP(hipMemcpyAsync((char*)h + (size_t)i * bytes, src(s), bytes, hipMemcpyDeviceToHost, s.cs));
  }
  int rc = sync_all(c);
  if (rc) return rc;
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    DeviceGuard g(s.dev);
    MV_HIP(hipMemcpyAsync(dst(s), h, bytes * (size_t)R, hipMemcpyHostToDevice, s.cs));
  }
  return MV_OK;
}

// The rerank plan of a two-stage query must be the same on every shard (same slabs, same options): the pruning stage needs every
// shard's scores of the global list.
int comm_plan(mv_comm* c, int mode, int n_coarse, int k, int n_q_rows, bool batched, RerankPlan* out) {
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  for (int i = 0; i < c->n; ++i) {
    mv_index* ix = c->sh[i].ix;
    std::lock_guard<std::mutex> ql(ix->q_mu);
    const RerankPlan p = rerank_plan(ix, mode, n_coarse, k, rpq, batched);
    if (i == 0) *out = p;
    else if (p.mid != out->mid || p.n_mid != out->n_mid || p.final_fp8 != out->final_fp8 || p.host_tier != out->host_tier) {
      set_error("two-stage query: shard %d keeps different slabs / rerank options than shard 0 (exact tier, fp8 slab, MV_OPT_RERANK_N, MV_OPT_EXACT_TIER must agree)", i);
      return MV_ERR_STATE;
    }
  }
  return MV_OK;
}

}  // namespace

extern "C" {

void mv_comm_destroy(mv_comm* c) {
  if (!c) return;
  for (Shard& s : c->sh) {
    DeviceGuard g(s.dev);
    if (s.cs) (void)hipStreamSynchronize(s.cs);
    if (s.nccl && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(s.nccl);
    for (void* p : {(void*)s.d_ls, (void*)s.d_li, (void*)s.d_gs, (void*)s.d_gi, (void*)s.d_recs, (void*)s.d_all, (void*)s.d_brecs, (void*)s.d_ball,
                    (void*)s.d_bls, (void*)s.d_bli, (void*)s.d_mid, (void*)s.d_mid_all, (void*)s.d_bmid, (void*)s.d_bmid_all})
      if (p) (void)hipFree(p);
    for (hipEvent_t e : {s.ev_t0, s.ev_t1, s.ev_done})
      if (e) (void)hipEventDestroy(e);
    if (s.cs) (void)hipStreamDestroy(s.cs);
  }
  if (!c->sh.empty()) {
    DeviceGuard g(c->sh[0].dev);
    if (c->d_os) (void)hipFree(c->d_os);
    if (c->d_oi) (void)hipFree(c->d_oi);
  }
  for (void* p : {(void*)c->h_s, (void*)c->h_i, (void*)c->h_recs, (void*)c->h_bs, (void*)c->h_bi, (void*)c->h_brecs, (void*)c->h_mid})
    if (p) (void)hipHostFree(p);
  // the RCCL handle stays loaded for the life of the process (its worker threads outlive communicators)
  delete c;
}

int mv_comm_create(int32_t n_shards, const int32_t* device_ids, int32_t transport, mv_comm** out) {
  if (!out || !device_ids || n_shards < 1 || n_shards > 64) { set_error("mv_comm_create: bad argument"); return MV_ERR_INVALID; }
  *out = nullptr;
  if (transport < MV_COMM_AUTO || transport > MV_COMM_HOST) { set_error("mv_comm_create: unknown transport %d", transport); return MV_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (libmvmaxsim requires an MI355X / gfx950 GPU)"); return MV_ERR_HIP; }
  bool distinct = true;
  for (int i = 0; i < n_shards; ++i) {
    if (device_ids[i] < 0 || device_ids[i] >= ndev) { set_error("device %d out of range (have %d)", device_ids[i], ndev); return MV_ERR_INVALID; }
    for (int j = 0; j < i; ++j) distinct = distinct && device_ids[j] != device_ids[i];
  }
  if (transport == MV_COMM_RCCL && !distinct) { set_error("MV_COMM_RCCL needs one distinct device per shard (logical shards on one device: use MV_COMM_P2P)"); return MV_ERR_INVALID; }
  mv_comm* c = new (std::nothrow) mv_comm();
  if (!c) { set_error("host allocation failed"); return MV_ERR_NOMEM; }
  c->n = n_shards;
  c->sh.resize((size_t)n_shards);
  int rc = MV_OK;
  auto fail = [&](int code) { std::string keep = mv_last_error(); mv_comm_destroy(c); set_error("%s", keep.c_str()); return code; };
  for (int i = 0; i < n_shards && !rc; ++i) {
    Shard& s = c->sh[i];
    s.dev = device_ids[i];
    DeviceGuard g(s.dev);
    const size_t R = (size_t)n_shards;
    if (hipStreamCreateWithFlags(&s.cs, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&s.ev_t0) != hipSuccess ||
        hipEventCreate(&s.ev_t1) != hipSuccess || hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) != hipSuccess) { set_error("mv_comm_create: stream / event creation failed"); rc = MV_ERR_HIP; break; }
    if (hipMalloc(&s.d_ls, kK * 4) != hipSuccess || hipMalloc(&s.d_li, kK * 8) != hipSuccess || hipMalloc(&s.d_gs, R * kK * 4) != hipSuccess ||
        hipMalloc(&s.d_gi, R * kK * 8) != hipSuccess || hipMalloc(&s.d_recs, kK * sizeof(mv_cand_rec)) != hipSuccess ||
        hipMalloc(&s.d_all, R * kK * sizeof(mv_cand_rec)) != hipSuccess || hipMalloc(&s.d_mid, kK * 4) != hipSuccess ||
        hipMalloc(&s.d_mid_all, R * kK * 4) != hipSuccess) { set_error("mv_comm_create: out of device memory"); rc = MV_ERR_NOMEM; break; }
  }
  if (!rc) {
    DeviceGuard g(c->sh[0].dev);
    const size_t R = (size_t)n_shards;
    if (hipMalloc(&c->d_os, kK * 4) != hipSuccess || hipMalloc(&c->d_oi, kK * 8) != hipSuccess ||
        hipHostMalloc((void**)&c->h_s, (R + 1) * kK * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_i, (R + 1) * kK * 8, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_recs, R * kK * sizeof(mv_cand_rec), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_mid, R * (size_t)kFdeBatchMaxQueries * kK * 4, hipHostMallocDefault) != hipSuccess) { set_error("mv_comm_create: out of memory"); rc = MV_ERR_NOMEM; }
  }
  if (rc) return fail(rc);
  c->transport = transport;
  if (transport == MV_COMM_AUTO) c->transport = (distinct && n_shards > 1) ? MV_COMM_RCCL : MV_COMM_P2P;
  if (c->transport == MV_COMM_RCCL) {
    bool ok = c->rccl.load();
    if (ok) {
      std::vector<ncclComm_t> comms((size_t)n_shards, nullptr);
      std::vector<int> devs(device_ids, device_ids + n_shards);
      ncclResult_t r = c->rccl.CommInitAll(comms.data(), n_shards, devs.data());
      if (r == ncclSuccess) {
        for (int i = 0; i < n_shards; ++i) c->sh[i].nccl = comms[i];
      } else {
        set_error("ncclCommInitAll failed: %s", c->rccl.GetErrorString(r));
        ok = false;
      }
    } else {
      set_error("librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "symbol missing");
    }
    if (!ok) {
      if (transport == MV_COMM_RCCL) return fail(MV_ERR_HIP);  // asked for explicitly
      c->transport = MV_COMM_P2P;
    }
  }
  *out = c;
  return MV_OK;
}

int mv_comm_attach(mv_comm* c, int32_t shard, mv_index* ix) {
  if (!c || !ix || shard < 0 || shard >= c->n) { set_error("mv_comm_attach: bad argument"); return MV_ERR_INVALID; }
  if (ix->cfg.device != c->sh[shard].dev) { set_error("mv_comm_attach: shard %d lives on device %d, the index on device %d", shard, c->sh[shard].dev, ix->cfg.device); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(c->mu);
  c->sh[shard].ix = ix;
  return MV_OK;
}

int mv_comm_transport(const mv_comm* c) { return c ? c->transport : MV_ERR_INVALID; }

int mv_comm_query_topk(mv_comm* c, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode, const uint32_t* allow_bits,
                       int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats) {
  if (!c || !q || !out_n || k < 0 || (k > 0 && (!out_scores || !out_ids))) { set_error("mv_comm_query_topk: bad argument"); return MV_ERR_INVALID; }
  if (k > kK) { set_error("mv_comm_query_topk supports k <= %d", kK); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_q_rows * kDim, mode)) return frc;
  std::lock_guard<std::mutex> lk(c->mu);
  *out_n = 0;
  const int R = c->n;
  for (int i = 0; i < R; ++i)
    if (!c->sh[i].ix) { set_error("mv_comm_query_topk: shard %d has no index attached", i); return MV_ERR_STATE; }
  if (stats) memset(stats, 0, sizeof(mv_query_stats) * (size_t)R);
  if (k == 0) return MV_OK;
  const bool two_stage = mode == MV_MODE_FDE_THEN_FLOAT || mode == MV_MODE_FP8_THEN_FLOAT;
  int rc = MV_OK;
  // ---- enqueue the local work of every shard; nothing below waits for a GPU until the final copy
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    DeviceGuard g(s.dev);
    MV_HIP(hipEventRecord(s.ev_t0, s.cs));
  }
  int n_coarse = 0;
  if (two_stage) {
    // the candidate rule of ONE index: FDE coarse top-n (reference: min(10 k, 75)) or the e4m3 scan's top-max(MV_OPT_RERANK_N, k),
    // taken over ALL shards before anything is reranked
    n_coarse = mode == MV_MODE_FDE_THEN_FLOAT ? (int)std::min<int64_t>(coarse_n_for(c->sh[0].ix, k), kK)
                                              : (int)std::min<int64_t>(std::max<int64_t>(c->sh[0].ix->rerank_n, k), kK);
    if ((int64_t)R * n_coarse > 16384) { set_error("two-stage query: %d shards x %d candidates exceed 16384", R, n_coarse); return MV_ERR_INVALID; }
    RerankPlan plan;
    rc = comm_plan(c, mode, n_coarse, k, n_q_rows, false, &plan);
    if (rc) return rc;
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      rc = mv_two_stage_coarse_device(s.ix, q, q_dtype, n_q_rows, n_coarse, mode, allow_bits, n_allow_words, s.d_recs, s.cs);
      if (rc) return rc;
    }
    const size_t rb = (size_t)n_coarse * sizeof(mv_cand_rec);
    if (c->transport == MV_COMM_HOST) rc = exchange_host(c, rb, c->h_recs, src_recs, dst_all);
    else rc = exchange(c, rb, true, src_recs, dst_all);
    if (rc) return rc;
    if (plan.mid) {  // pruning stage: every shard scores its share of the global list on its e4m3 slab; n floats per shard are exchanged
      for (int i = 0; i < R; ++i) {
        Shard& s = c->sh[i];
        rc = mv_two_stage_mid_device(s.ix, q, q_dtype, n_q_rows, mode, s.d_all, R, n_coarse, s.d_mid, s.cs);
        if (rc) return rc;
      }
      const size_t mb = (size_t)n_coarse * 4;
      if (c->transport == MV_COMM_HOST) rc = exchange_host(c, mb, c->h_mid, src_mid, dst_mid_all);
      else rc = exchange(c, mb, true, src_mid, dst_mid_all);
      if (rc) return rc;
    }
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      rc = mv_two_stage_rerank_device(s.ix, q, q_dtype, n_q_rows, mode, s.d_all, R, n_coarse, plan.mid ? s.d_mid_all : nullptr, plan.n_mid, k, s.d_ls, s.d_li, s.cs);
      if (rc) return rc;
    }
  } else {
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      rc = mv_internal_query_common(s.ix, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, nullptr, nullptr, nullptr, s.d_ls, s.d_li,
                                    s.cs, stats ? &stats[i] : nullptr, 1);
      if (rc) return rc;
    }
  }
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    DeviceGuard g(s.dev);
    MV_HIP(hipEventRecord(s.ev_t1, s.cs));
  }
  // ---- exchange of the k (score, id) pairs and merge
  float* hs = c->h_s + (size_t)R * kK;
  int64_t* hi = c->h_i + (size_t)R * kK;
  int n_out = 0;
  if (c->transport == MV_COMM_HOST) {
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      DeviceGuard g(s.dev);
      MV_HIP(hipMemcpyAsync(c->h_s + (size_t)i * k, s.d_ls, (size_t)k * 4, hipMemcpyDeviceToHost, s.cs));
      MV_HIP(hipMemcpyAsync(c->h_i + (size_t)i * k, s.d_li, (size_t)k * 8, hipMemcpyDeviceToHost, s.cs));
    }
    rc = sync_all(c);
    if (rc) return rc;
    host_merge(c->h_s, c->h_i, R, k, k, hs, hi, &n_out);
  } else {
    rc = exchange(c, (size_t)k * 4, false, src_ls, dst_gs);
    if (!rc) rc = exchange(c, (size_t)k * 8, false, src_li, dst_gi);
    if (rc) return rc;
    Shard& z = c->sh[0];
    DeviceGuard g(z.dev);
    if ((int64_t)R * k <= 2048) {
      rc = launch_merge_topk(z.d_gs, z.d_gi, R, k, k, c->d_os, c->d_oi, z.cs);
      if (rc) return rc;
      MV_HIP(hipMemcpyAsync(hs, c->d_os, (size_t)k * 4, hipMemcpyDeviceToHost, z.cs));
      MV_HIP(hipMemcpyAsync(hi, c->d_oi, (size_t)k * 8, hipMemcpyDeviceToHost, z.cs));
      rc = sync_all(c);
      if (rc) return rc;
      while (n_out < k && hi[n_out] >= 0) ++n_out;
    } else {  // beyond the merge kernel's 2048 keys: gather on the device, merge on the host
      MV_HIP(hipMemcpyAsync(c->h_s, z.d_gs, (size_t)R * k * 4, hipMemcpyDeviceToHost, z.cs));
      MV_HIP(hipMemcpyAsync(c->h_i, z.d_gi, (size_t)R * k * 8, hipMemcpyDeviceToHost, z.cs));
      rc = sync_all(c);
      if (rc) return rc;
      host_merge(c->h_s, c->h_i, R, k, k, hs, hi, &n_out);
    }
  }
  memcpy(out_scores, hs, (size_t)n_out * 4);
  memcpy(out_ids, hi, (size_t)n_out * 8);
  *out_n = n_out;
  if (stats) {
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      DeviceGuard g(s.dev);
      float whole = 0.f;
      if (!two_stage) {
        std::lock_guard<std::mutex> ql(s.ix->q_mu);
        rc = finish_stats(s.ix, &stats[i], true);
        if (rc) return rc;
      }
      MV_HIP(hipEventElapsedTime(&whole, s.ev_t0, s.ev_t1));
      stats[i].total_device_ms = whole;  // the shard's whole local span on its comm stream
    }
  }
  return MV_OK;
}


// A batch of requests against the sharded corpus.  MV_MODE_FDE_THEN_FLOAT: per group of <= 32 requests every shard makes ONE
// pass over its FDE slab for all of them (the batched coarse GEMM), ONE exchange carries every request's n_coarse
// candidate records, every shard derives each request's GLOBAL top-n, reranks its share of all lists in one launch and
// leaves the local top-k lists, which meet on the host (they are on their way there anyway).  Same candidate sets, pad
// lengths and answers as mv_query_topk_batch on ONE index holding every page.  Other modes: request by request.
int mv_comm_query_topk_batch(mv_comm* c, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k, int mode,
                             const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores, int64_t* out_ids,
                             int32_t* out_n, mv_query_stats* stats) {
  if (!c || !q || !out_n || n_queries < 1 || n_q_rows < 1 || k < 0 || (k > 0 && (!out_scores || !out_ids))) { set_error("mv_comm_query_topk_batch: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (k > kK) { set_error("mv_comm_query_topk_batch supports k <= %d", kK); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_queries * n_q_rows * kDim, mode)) return frc;
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const int R = c->n;
  const bool fde_mode = mode == MV_MODE_FDE_THEN_FLOAT;
  bool batched = (fde_mode || mode == MV_MODE_FP8_THEN_FLOAT) && n_queries > 1 && k >= 1 && rpq <= (fde_mode ? 512 : kMaxQRowsPerPass);
  int n_coarse = 0;
  RerankPlan plan;
  if (batched) {
    std::lock_guard<std::mutex> lk(c->mu);
    for (int i = 0; i < R && batched; ++i) {
      mv_index* ix = c->sh[i].ix;
      if (!ix) { set_error("mv_comm_query_topk_batch: shard %d has no index attached", i); return MV_ERR_STATE; }
      const bool fp8_rr = !(ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT));
      if (fde_mode)
        batched = (ix->cfg.flags & MV_WITH_FDE) && ix->fde_batch_variant != 1 && fde_scan_batch_supported(ix->fde_t.out_dim) &&
                  ix->fde_t.cfg.projection_dimension <= 16 && (!fp8_rr || ((ix->cfg.flags & MV_WITH_FP8) && rpq <= 64));
      else
        batched = (ix->cfg.flags & MV_WITH_FP8) && !fp8_rr && ix->batch_variant != 8;
    }
    n_coarse = fde_mode ? (int)std::min<int64_t>(coarse_n_for(c->sh[0].ix, k), kK) : (int)std::min<int64_t>(std::max<int64_t>(c->sh[0].ix->rerank_n, k), kK);
    if ((int64_t)R * n_coarse > 16384) batched = false;
    if (batched) {
      int prc = comm_plan(c, mode, n_coarse, k, n_q_rows, true, &plan);
      if (prc) return prc;
    }
  }
  if (!batched) {  // request by request through the single-query communicator (each call takes c->mu itself)
    for (int32_t b = 0; b < n_queries; ++b) {
      const uint32_t* ab = (allow_bits && allow_per_query) ? allow_bits + (size_t)b * n_allow_words : allow_bits;
      QueryFdeCursor cur(g_qfde.cur + b);
      int rc = mv_comm_query_topk(c, (const char*)q + (size_t)b * n_q_rows * kDim * esz, q_dtype, n_q_rows, k, mode, ab, n_allow_words,
                                  out_scores ? out_scores + (size_t)b * k : nullptr, out_ids ? out_ids + (size_t)b * k : nullptr, out_n + b, nullptr);
      if (rc) return rc;
    }
    if (stats) memset(stats, 0, sizeof(mv_query_stats) * (size_t)R);
    return MV_OK;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  if (stats) memset(stats, 0, sizeof(mv_query_stats) * (size_t)R);
  constexpr int kG = kFdeBatchMaxQueries;
  // batch buffers, once
  for (int i = 0; i < R; ++i) {
    Shard& s = c->sh[i];
    if (s.d_brecs) continue;
    DeviceGuard g(s.dev);
    if (hipMalloc(&s.d_brecs, (size_t)kG * kK * sizeof(mv_cand_rec)) != hipSuccess || hipMalloc(&s.d_ball, (size_t)R * kG * kK * sizeof(mv_cand_rec)) != hipSuccess ||
        hipMalloc(&s.d_bls, (size_t)kG * kK * 4) != hipSuccess || hipMalloc(&s.d_bli, (size_t)kG * kK * 8) != hipSuccess ||
        hipMalloc(&s.d_bmid, (size_t)kG * kK * 4) != hipSuccess || hipMalloc(&s.d_bmid_all, (size_t)R * kG * kK * 4) != hipSuccess) { set_error("mv_comm_query_topk_batch: out of device memory"); return MV_ERR_NOMEM; }
  }
  if (!c->h_bs) {
    if (hipHostMalloc((void**)&c->h_bs, (size_t)R * kG * kK * 4, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_bi, (size_t)R * kG * kK * 8, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_brecs, (size_t)R * kG * kK * sizeof(mv_cand_rec), hipHostMallocDefault) != hipSuccess) { set_error("mv_comm_query_topk_batch: out of pinned memory"); return MV_ERR_NOMEM; }
  }
  const int group = std::min((fde_mode ? kBatchQRows : 512) / rpq, kG);  // query rows per pass of the batched FDE / e4m3 scan
  const bool per_query = allow_bits && allow_per_query;
  int rc = MV_OK;
  std::vector<float> ms((size_t)R * k), os((size_t)k);
  std::vector<int64_t> mi((size_t)R * k), oi((size_t)k);
  for (int32_t b0 = 0; b0 < n_queries; b0 += group) {
    const int nb = std::min(group, n_queries - b0);
    const char* qg = (const char*)q + (size_t)b0 * n_q_rows * kDim * esz;
    const uint32_t* ag = per_query ? allow_bits + (size_t)b0 * n_allow_words : allow_bits;
    QueryFdeCursor fde_cur(g_qfde.cur + b0);  // a caller's own query FDEs: this group's start
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      DeviceGuard g(s.dev);
      MV_HIP(hipEventRecord(s.ev_t0, s.cs));
    }
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      rc = mv_internal_two_stage_batch_coarse(s.ix, qg, q_dtype, nb, n_q_rows, n_coarse, mode, ag, n_allow_words, per_query ? 1 : 0, s.d_brecs, s.cs);
      if (rc) return rc;
    }
    const size_t rb = (size_t)nb * n_coarse * sizeof(mv_cand_rec);
    if (c->transport == MV_COMM_HOST) rc = exchange_host(c, rb, c->h_brecs, src_brecs, dst_ball);
    else rc = exchange(c, rb, true, src_brecs, dst_ball);
    if (rc) return rc;
    if (plan.mid) {
      for (int i = 0; i < R; ++i) {
        Shard& s = c->sh[i];
        rc = mv_internal_two_stage_batch_mid(s.ix, qg, q_dtype, nb, n_q_rows, mode, s.d_ball, R, n_coarse, s.d_bmid, s.cs);
        if (rc) return rc;
      }
      const size_t mb = (size_t)nb * n_coarse * 4;
      if (c->transport == MV_COMM_HOST) rc = exchange_host(c, mb, c->h_mid, src_bmid, dst_bmid_all);
      else rc = exchange(c, mb, true, src_bmid, dst_bmid_all);
      if (rc) return rc;
    }
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      rc = mv_internal_two_stage_batch_rerank(s.ix, qg, q_dtype, nb, n_q_rows, mode, s.d_ball, R, n_coarse, plan.mid ? s.d_bmid_all : nullptr, plan.n_mid, k, s.d_bls,
                                              s.d_bli, s.cs);
      if (rc) return rc;
    }
    for (int i = 0; i < R; ++i) {
      Shard& s = c->sh[i];
      DeviceGuard g(s.dev);
      MV_HIP(hipEventRecord(s.ev_t1, s.cs));
      MV_HIP(hipMemcpyAsync(c->h_bs + (size_t)i * nb * k, s.d_bls, (size_t)nb * k * 4, hipMemcpyDeviceToHost, s.cs));
      MV_HIP(hipMemcpyAsync(c->h_bi + (size_t)i * nb * k, s.d_bli, (size_t)nb * k * 8, hipMemcpyDeviceToHost, s.cs));
    }
    rc = sync_all(c);
    if (rc) return rc;
    for (int b = 0; b < nb; ++b) {  // merge of the R local lists of request b (score desc; ties: shard asc = id asc)
      for (int i = 0; i < R; ++i) {
        memcpy(ms.data() + (size_t)i * k, c->h_bs + ((size_t)i * nb + b) * k, (size_t)k * 4);
        memcpy(mi.data() + (size_t)i * k, c->h_bi + ((size_t)i * nb + b) * k, (size_t)k * 8);
      }
      int m = 0;
      host_merge(ms.data(), mi.data(), R, k, k, os.data(), oi.data(), &m);
      memcpy(out_scores + (size_t)(b0 + b) * k, os.data(), (size_t)m * 4);
      memcpy(out_ids + (size_t)(b0 + b) * k, oi.data(), (size_t)m * 8);
      out_n[b0 + b] = m;
    }
    if (stats) {
      for (int i = 0; i < R; ++i) {
        Shard& s = c->sh[i];
        DeviceGuard g(s.dev);
        float whole = 0.f;
        MV_HIP(hipEventElapsedTime(&whole, s.ev_t0, s.ev_t1));
        stats[i].total_device_ms += whole;  // the shard's whole local span of the group on its comm stream
        stats[i].score_launches += 1;
      }
    }
  }
  return MV_OK;
}

// ---- the same entry points with the caller's own query FDE vectors (mv_query_topk_fde, mv_api.hip): every shard's coarse stage takes
// the query's FDE from the caller instead of encoding the query rows on its device
static int comm_fde_check(mv_comm* c, int mode, const float* q_fde, int64_t n_queries, const char* what) {
  if (!c || !q_fde || n_queries < 1) { set_error("%s: bad argument", what); return MV_ERR_INVALID; }
  if (mode != MV_MODE_FDE_THEN_FLOAT && mode != MV_MODE_FDE_ONLY) { set_error("%s: mode %d has no FDE stage", what, mode); return MV_ERR_INVALID; }
  const mv_index* ix0 = c->n > 0 ? c->sh[0].ix : nullptr;
  if (!ix0 || !(ix0->cfg.flags & MV_WITH_FDE)) { set_error("%s: shard 0 has no index with an FDE slab attached", what); return MV_ERR_STATE; }
  return check_fde_finite(q_fde, (size_t)n_queries * (size_t)ix0->fde_t.out_dim, what);
}

int mv_comm_query_topk_fde(mv_comm* c, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t k, int mode, const uint32_t* allow_bits,
                           int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats) {
  if (int rc = comm_fde_check(c, mode, q_fde, 1, "mv_comm_query_topk_fde")) return rc;
  QueryFdeScope sc(q_fde);
  return mv_comm_query_topk(c, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, out_scores, out_ids, out_n, stats);
}

int mv_comm_query_topk_batch_fde(mv_comm* c, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, const float* q_fde, int32_t k, int mode,
                                 const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores, int64_t* out_ids,
                                 int32_t* out_n, mv_query_stats* stats) {
  if (int rc = comm_fde_check(c, mode, q_fde, n_queries, "mv_comm_query_topk_batch_fde")) return rc;
  QueryFdeScope sc(q_fde);
  return mv_comm_query_topk_batch(c, q, q_dtype, n_queries, n_q_rows, k, mode, allow_bits, n_allow_words, allow_per_query, out_scores, out_ids, out_n, stats);
}

int mv_two_stage_coarse_device_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t n_coarse, int mode,
                                   const uint32_t* allow_bits, int64_t n_allow_words, mv_cand_rec* d_out_recs, void* stream) {
  if (!ix || !q_fde) { set_error("two_stage_coarse_fde: null argument"); return MV_ERR_INVALID; }
  if (mode != MV_MODE_FDE_THEN_FLOAT) { set_error("two_stage_coarse_fde: MV_MODE_FDE_THEN_FLOAT only"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }
  if (int rc = check_fde_finite(q_fde, (size_t)ix->fde_t.out_dim, "two_stage_coarse_fde")) return rc;
  QueryFdeScope sc(q_fde);
  return mv_two_stage_coarse_device(ix, q, q_dtype, n_q_rows, n_coarse, mode, allow_bits, n_allow_words, d_out_recs, stream);
}

}  // extern "C"
