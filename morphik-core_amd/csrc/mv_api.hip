// mv_api.hip -- the C ABI of libmvmaxsim.so (include/mvmaxsim.h) over the HIP kernels.
//
// One mv_index = one GPU's shard of the corpus: a page-contiguous bf16 slab
// [capacity][stride_rows][128], optionally a sign-bit slab [capacity][stride_rows][16 B] and an FDE
// slab [capacity][out_dim] bf16, plus per-page metadata (valid rows, document ordinal).  All device
// memory is allocated once at create: at 1M x 1024 x 128 bf16 the slab is 262 GB of the 288 GB HBM,
// so there is no room for growth-by-copy.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <new>
#include <vector>

#include "mv_index_priv.h"

#include <fcntl.h>
#include <unistd.h>

namespace mv {

static thread_local std::string g_err;
static int g_stateless_fde_variant = 1;  // mv_fde_encode (no index): 1 = f32-MFMA kernel; MV_FDE_SCALAR=1 in the environment selects the scalar kernel

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  return e == hipErrorOutOfMemory ? MV_ERR_NOMEM : MV_ERR_HIP;
}

uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float host_bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

bool host_rows_finite(const void* x, int dtype, size_t n_elems) {
  if (dtype == MV_BF16) {
    const uint16_t* h = (const uint16_t*)x;
    for (size_t i = 0; i < n_elems; ++i)
      if ((h[i] & 0x7f80u) == 0x7f80u) return false;
    return true;
  }
  const float* f = (const float*)x;
  for (size_t i = 0; i < n_elems; ++i)
    if ((host_f32_to_bf16(f[i]) & 0x7f80u) == 0x7f80u) return false;  // NaN, +-Inf, or beyond the bf16 range (an Inf once rounded)
  return true;
}

int check_query_finite(const void* q, int q_dtype, size_t n_elems, int mode) {
  if (mode == MV_MODE_BINARY || host_rows_finite(q, q_dtype, n_elems)) return MV_OK;
  set_error("the query holds a NaN / Inf value: MaxSim is undefined for it (only MV_MODE_BINARY defines non-finite inputs: bit = v > 0)");
  return MV_ERR_INVALID;
}

__global__ void add_scores_kernel(float* dst, const float* src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
__global__ void ids64_to_32_kernel(const int64_t* in, int32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

}  // namespace mv

using namespace mv;

namespace mv {

int ensure_query_cap(mv_index* ix, int n_rows) {
  const int padded = ((n_rows + 15) / 16) * 16;
  if (padded <= ix->q_rows_cap) return MV_OK;
  const int cap = std::max(padded, 256);
  if (ix->stream) (void)hipStreamSynchronize(ix->stream);  // nothing may still read the buffers being replaced
  if (ix->d_q) (void)hipFree(ix->d_q);
  if (ix->d_qlo) (void)hipFree(ix->d_qlo);
  if (ix->h_qlo) (void)hipHostFree(ix->h_qlo);
  ix->d_qlo = nullptr; ix->h_qlo = nullptr; ix->q_lo_valid = false;
  if (ix->d_qf32) (void)hipFree(ix->d_qf32);
  if (ix->d_qbits) (void)hipFree(ix->d_qbits);
  if (ix->d_qpop) (void)hipFree(ix->d_qpop);
  if (ix->h_qf32) (void)hipHostFree(ix->h_qf32);
  if (ix->h_qbf16) (void)hipHostFree(ix->h_qbf16);
  ix->h_qf32 = nullptr; ix->h_qbf16 = nullptr;
  if (ix->d_q8hi) (void)hipFree(ix->d_q8hi);
  if (ix->d_q8lo) (void)hipFree(ix->d_q8lo);
  if (ix->d_q8fac) (void)hipFree(ix->d_q8fac);
  ix->d_q = nullptr; ix->d_qf32 = nullptr; ix->d_qbits = nullptr; ix->d_qpop = nullptr;
  ix->d_q8hi = nullptr; ix->d_q8lo = nullptr; ix->d_q8fac = nullptr;
  MV_HIP(hipMalloc(&ix->d_q, (size_t)cap * kDim * 2));
  MV_HIP(hipMalloc(&ix->d_qlo, (size_t)cap * kDim * 2));
  MV_HIP(hipHostMalloc((void**)&ix->h_qlo, (size_t)cap * kDim * 2, hipHostMallocDefault));
  MV_HIP(hipMalloc(&ix->d_qf32, (size_t)cap * kDim * 4));
  MV_HIP(hipMalloc(&ix->d_qbits, (size_t)cap * kSignBytes));
  MV_HIP(hipMalloc(&ix->d_qpop, (size_t)cap * 4));
  MV_HIP(hipMalloc(&ix->d_q8hi, (size_t)cap * kDim));
  MV_HIP(hipMalloc(&ix->d_q8lo, (size_t)cap * kDim));
  MV_HIP(hipMalloc(&ix->d_q8fac, (size_t)cap * 4));
  MV_HIP(hipHostMalloc((void**)&ix->h_qf32, (size_t)cap * kDim * 4, hipHostMallocDefault));
  MV_HIP(hipHostMalloc((void**)&ix->h_qbf16, (size_t)cap * kDim * 2, hipHostMallocDefault));
  ix->q_rows_cap = cap;
  return MV_OK;
}

// Upload the query in every representation the mode needs.  Returns padded row count.
int upload_query(mv_index* ix, const void* q, int q_dtype, int n_q, bool want_bf16, bool want_f32, bool want_bits,
                 bool want_fp8) {
  int rc = ensure_query_cap(ix, n_q);
  if (rc) return rc;
  const int padded = ((n_q + 15) / 16) * 16;
  // the pinned staging buffers are free once the previous query's H2D copies have run
  if (ix->ev_stage) MV_HIP(hipEventSynchronize(ix->ev_stage));
  const size_t ne = (size_t)n_q * kDim;
  float* f = ix->h_qf32;
  const bool need_f32 = want_f32 || want_bits || want_fp8 || (want_bf16 && q_dtype == MV_F32);
  if (need_f32) {
    if (q_dtype == MV_F32) {
      memcpy(f, q, ne * 4);
    } else {
      const uint16_t* h = (const uint16_t*)q;
      for (size_t i = 0; i < ne; ++i) f[i] = host_bf16_to_f32(h[i]);
    }
  }
  if (want_bf16) {
    // An fp32 query is split q = hi + lo (hi = bf16(q), lo = bf16(q - hi): the subtraction is exact in fp32, |q - hi - lo| <= 2^-18 |q|)
    // and BOTH halves are scored (mv_maxsim.hip tile_mfma_lo): the query side of the reference's fp32 product
    // (fast_multivector_store.py:553-555) costs a second MFMA chain in an HBM-bound scan, not a rounding.  A query that IS bf16
    // (MV_BF16, or fp32 values with lo == 0) takes the one-term kernels unchanged.
    uint16_t* b = ix->h_qbf16;
    uint16_t* bl = ix->h_qlo;
    bool any_lo = false;
    if (q_dtype == MV_BF16) memcpy(b, q, ne * 2);
    else for (size_t i = 0; i < ne; ++i) {
      b[i] = host_f32_to_bf16(f[i]);
      bl[i] = host_f32_to_bf16(f[i] - host_bf16_to_f32(b[i]));
      any_lo = any_lo || (bl[i] & 0x7fffu) != 0;
    }
    memset(b + ne, 0, ((size_t)padded * kDim - ne) * 2);
    MV_HIP(hipMemcpyAsync(ix->d_q, b, (size_t)padded * kDim * 2, hipMemcpyHostToDevice, ix->stream));
    ix->q_has_lo = any_lo;
    ix->q_lo_valid = any_lo || ix->slab_lo != nullptr;  // the three-term kernels of a lo slab always take a lo query (zeros for a bf16 one)
    if (ix->q_lo_valid) {
      if (any_lo) memset(bl + ne, 0, ((size_t)padded * kDim - ne) * 2);
      else memset(bl, 0, (size_t)padded * kDim * 2);
      MV_HIP(hipMemcpyAsync(ix->d_qlo, bl, (size_t)padded * kDim * 2, hipMemcpyHostToDevice, ix->stream));
    }
  }
  if (want_f32 || want_bits || want_fp8) {
    MV_HIP(hipMemcpyAsync(ix->d_qf32, f, ne * 4, hipMemcpyHostToDevice, ix->stream));
    if (want_bits) {
      rc = launch_sign_pack_f32(ix->d_qf32, n_q, kDim, ix->d_qbits, ix->stream);
      if (rc) return rc;
    }
    if (want_fp8) {
      rc = launch_fp8_query_prep(ix->d_qf32, n_q, ix->d_q8hi, ix->d_q8lo, ix->d_q8fac, ix->stream);
      if (rc) return rc;
    }
  }
  // (a call that drains the stream before it returns needs no marker for the staging buffers: one event record less in the chain)
  if (!ix->sync_call) MV_HIP(hipEventRecord(ix->ev_stage, ix->stream));
  return MV_OK;
}

int upload_allow(mv_index* ix, const uint32_t* allow_bits, int64_t n_words, const uint32_t** d_allow) {
  *d_allow = nullptr;
  if (!allow_bits) return MV_OK;
  if (n_words < 0) { set_error("negative allow bitmap length"); return MV_ERR_INVALID; }
  const int64_t need = std::max<int64_t>(n_words, 1);
  if (need > ix->allow_cap_words) {
    if (ix->d_allow) (void)hipFree(ix->d_allow);
    ix->d_allow = nullptr;
    MV_HIP(hipMalloc(&ix->d_allow, (size_t)need * 4));
    ix->allow_cap_words = need;
  }
  if (n_words > 0) MV_HIP(hipMemcpyAsync(ix->d_allow, allow_bits, (size_t)n_words * 4, hipMemcpyHostToDevice, ix->stream));
  *d_allow = ix->d_allow;
  return MV_OK;
}

int64_t count_allowed_rows(const mv_index* ix, int64_t n, const uint32_t* allow_bits, int64_t n_words, int64_t* pages_out) {
  // algorithmic accounting on the host metadata (exact): pages that are read and their valid rows
  int64_t rows = 0, pages = 0;
  const bool filt = allow_bits != nullptr;
  if (!filt && !ix->tombstones.load() && !ix->ragged.load()) {
    *pages_out = n;
    return n * (int64_t)ix->cfg.stride_rows;
  }
  for (int64_t p = 0; p < n; ++p) {
    // a writer may tombstone a PUBLISHED page (remove_*, under w_mu) while this accounting loop runs under q_mu: the
    // page counts as read or not -- either is a valid ordering -- but the access itself must not be a data race
    // (ThreadSanitizer, tools/sanitize/): relaxed atomics on both sides
    const int32_t o = __atomic_load_n(&ix->h_doc_ord[p], __ATOMIC_RELAXED);
    if (o < 0) continue;
    if (filt && ((int64_t)o >= n_words * 32 || !((allow_bits[o >> 5] >> (o & 31)) & 1u))) continue;
    ++pages;
    rows += ix->h_n_rows[p];
  }
  *pages_out = pages;
  return rows;
}

// Candidate list of a rerank, built on the device: ids (int64 from the selection kernels, or int32 from the caller)
// -> int32 local pages (-1 kept as padding) + the pad_to of every candidate = the longest page of ITS batch of 128 in
// list order (score_multi_vector pads each passage batch on its own: pad_sequence at fast_multivector_store.py:553-555
// -> colpali_engine score_multi_vector, batch_size = 128).  One block per batch.
__global__ __launch_bounds__(kRerankBatch) void cand_prepare_kernel(const int64_t* ids64, const int32_t* ids32, int n,
                                                                    const int32_t* n_rows, int32_t stride, int pad_sem,
                                                                    int32_t* cand, int32_t* pads, int64_t list_stride) {
  __shared__ int32_t wmax[kRerankBatch / 64];
  // blockIdx.y: list of a batch of queries (every list starts its own batches of 128), list_stride entries apart
  if (ids64) ids64 += (int64_t)blockIdx.y * list_stride;
  if (ids32) ids32 += (int64_t)blockIdx.y * list_stride;
  cand += (int64_t)blockIdx.y * list_stride;
  pads += (int64_t)blockIdx.y * list_stride;
  const int i = blockIdx.x * kRerankBatch + threadIdx.x;
  int32_t c = -1, rows = 0;
  if (i < n) {
    c = ids64 ? (ids64[i] < 0 ? -1 : (int32_t)ids64[i]) : ids32[i];
    if (c >= 0) rows = n_rows ? n_rows[c] : stride;
  }
  int32_t m = rows;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) m = max(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(wmax[0], wmax[1]);
  if (i < n) {
    cand[i] = c;
    pads[i] = pad_sem ? m : 0;
  }
}

int launch_cand_prepare(mv_index* ix, const int64_t* d_ids64, const int32_t* d_ids32, int n, int pad_sem) {
  if (n <= 0) return MV_OK;
  hipLaunchKernelGGL(cand_prepare_kernel, dim3((unsigned)((n + kRerankBatch - 1) / kRerankBatch)), dim3(kRerankBatch), 0, ix->stream,
                     d_ids64, d_ids32, n, (const int32_t*)ix->d_n_rows, ix->cfg.stride_rows, pad_sem, ix->d_cand, ix->d_cand_pads, (int64_t)0);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int float_scan(mv_index* ix, int n_q, const uint32_t* d_allow, int64_t n_allow_words, const int32_t* d_cand, int64_t n_items,
               int32_t pad_to, const int32_t* d_pad_items, float* d_out, int* launches, bool no_mask = false,
               const uint16_t* d_q_base = nullptr, const uint16_t* slab_override = nullptr, const uint16_t* d_qlo_base = nullptr,
               const uint16_t* slab_lo = nullptr, const int64_t* row_off_override = nullptr) {
  if (n_items <= 0) return MV_OK;  // nothing to launch (a grid of 0 blocks is an invalid configuration)
  const uint16_t* qbase = d_q_base ? d_q_base : ix->d_q;  // padded bf16 query rows (a batch keeps its queries in d_bq)
  // split-bf16 operands: the query's lo rows (single query: d_qlo when upload_query filled it; a batch block: the caller's d_qlo_base)
  // and -- slab_lo -- the pages' lo slab (only ever beside the bf16 slab in HBM: the host exact tier keeps hi rows only)
  const uint16_t* qlo_base = d_q_base ? d_qlo_base : (ix->q_lo_valid ? ix->d_qlo : nullptr);
  if (slab_lo && !qlo_base) { set_error("float_scan: the lo slab needs the query's lo rows"); return MV_ERR_STATE; }
  const bool lo_q = qlo_base != nullptr && (slab_lo != nullptr || d_q_base != nullptr || ix->q_has_lo);
  if (!lo_q) qlo_base = nullptr;
  const bool need_meta = !no_mask && (ix->tombstones.load() || d_allow != nullptr);
  const bool ragged = ix->ragged.load();
  const int padded = ((n_q + 15) / 16) * 16;
  int done = 0, pass = 0;
  // A long query (> 64 rows) over the whole slab goes through the row-split workgroup of the batched scan (4 waves
  // share the page tiles in LDS, each holds a quarter of the query rows) as ONE query of up to 512 rows: the
  // page-split kernel below keeps all query rows in every wave and falls off the HBM roof past 64 rows
  // (400 k pages: 128 rows 19.2 -> 16.0 ms, 256 rows 38.3 -> 22.7 ms).
  if (!d_cand && pad_to == 0 && !d_pad_items && padded > 64 && ix->long_query_variant == 1 && !lo_q) {  // (split-bf16 operands: the page-split passes below)
    if (!ix->d_bq) MV_HIP(hipMalloc(&ix->d_bq, (size_t)kBatchQRows * kRowBytes));
    while (done < padded) {
      const int left = padded - done;
      const int rows = left <= 512 ? left : 384;
      MV_HIP(hipMemsetAsync(ix->d_bq, 0, (size_t)512 * kRowBytes, ix->stream));
      MV_HIP(hipMemcpyAsync(ix->d_bq, qbase + (size_t)done * kDim, (size_t)rows * kRowBytes, hipMemcpyDeviceToDevice, ix->stream));
      BatchArgs b{};
      b.slab = ix->slab; b.n_rows = ragged ? ix->d_n_rows : nullptr; b.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
      b.allow = d_allow; b.n_allow_bits = n_allow_words * 32; b.allow_stride_bits = 0;
      b.q = ix->d_bq; b.scores = pass == 0 ? d_out : ix->d_scores2; b.n = n_items; b.score_stride = n_items;
      b.stride = ix->cfg.stride_rows; b.n_queries = 1; b.rows_per_query = rows; b.variant = ix->batch_variant >= 0 ? ix->batch_variant : 0;
      b.row_off = ix->d_row_off;
      int rc = launch_maxsim_batch(b, ix->stream);
      if (rc) return rc;
      ++*launches;
      if (pass > 0)
        hipLaunchKernelGGL(add_scores_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, ix->stream, d_out,
                           (const float*)ix->d_scores2, n_items);
      done += rows;
      ++pass;
    }
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  while (done < padded) {
    const int rows = std::min(padded - done, lo_q ? kMaxQRowsPerPass / 2 : kMaxQRowsPerPass);  // the lo fragments take the registers of 64 rows
    MaxsimArgs a{};
    a.qlo = lo_q ? qlo_base + (size_t)done * kDim : nullptr;
    a.slab_lo = slab_lo;
    a.row_off = row_off_override ? row_off_override : ix->d_row_off;  // packed layout, or the placement table of a rebalanced split exact tier (never both)
    a.slab = slab_override ? slab_override : ix->slab;  // the exact tier of FP8_THEN_FLOAT may be pinned host memory mapped into the device
    a.n_rows = (ragged || a.row_off) ? ix->d_n_rows : nullptr;  // the table-driven kernels read the row count unconditionally (it always exists: stride_rows for a full page)
    a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    a.allow = d_allow;
    a.n_allow_bits = n_allow_words * 32;
    a.cand = d_cand;
    a.q = qbase + (size_t)done * kDim;
    a.scores = pass == 0 ? d_out : ix->d_scores2;
    a.n = n_items;
    a.stride = ix->cfg.stride_rows;
    a.q_tiles = rows / 16;
    a.pad_to = pad_to;
    a.pad_items = d_pad_items;
    int rc = launch_maxsim_bf16(a, ix->maxsim_variant, ix->stream);
    if (rc) return rc;
    ++*launches;
    if (pass > 0) {
      // -inf (masked item / padding entry) + anything stays -inf
      hipLaunchKernelGGL(add_scores_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, ix->stream, d_out,
                         (const float*)ix->d_scores2, n_items);
    }
    done += rows;
    ++pass;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// Exact float MaxSim over the e4m3 slab (all query rows in passes of 64 inside the launcher).
int fp8_scan(mv_index* ix, int n_q, const uint32_t* d_allow, int64_t n_allow_words, const int32_t* d_cand, int64_t n_items,
             int32_t pad_to, const int32_t* d_pad_items, float* d_out, int* launches, bool no_mask) {
  if (n_items <= 0) return MV_OK;
  const bool need_meta = !no_mask && (ix->tombstones.load() || d_allow != nullptr);
  Fp8ScanArgs a{};
  a.slab = ix->slab8; a.inv_scale = ix->inv_scale8;
  a.n_rows = ix->ragged.load() ? ix->d_n_rows : nullptr;
  a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
  a.allow = d_allow; a.n_allow_bits = n_allow_words * 32; a.cand = d_cand;
  a.qhi = ix->d_q8hi; a.qlo = ix->d_q8lo; a.qfac = ix->d_q8fac; a.n_q = n_q;
  a.scores = d_out; a.n = n_items; a.stride = ix->cfg.stride_rows; a.pad_to = pad_to; a.pad_items = d_pad_items;
  a.row_off = ix->d_row_off;
  int rc = launch_maxsim_fp8(a, ix->stream);
  if (rc) return rc;
  *launches += (((n_q + 15) / 16) * 16 + 63) / 64;
  return MV_OK;
}

__global__ void fde8_cfac_kernel(const float* scale, const float* inv_norm, int64_t n, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = scale[i] * inv_norm[i];
}

// MV_WITH_FDE_E4M3: the e4m3 copy of FDE rows [first, first + n), derived from the bf16 rows just written (every writer of ix->fde calls it).
// A row is quantised like a page of the e4m3 slab: out_dim / 128 "rows" of 128 under ONE power-of-two scale (orc_quantize_page_fp8).
static int fde8_requantize(mv_index* ix, int64_t first, int64_t n, hipStream_t st) {
  if (n <= 0) return MV_OK;
  const int64_t od = ix->fde_t.out_dim;
  if (ix->fde4) {  // MV_WITH_FDE_FP4: the e2m1 copy (mv_fde4.hip; oracle orc_quantize_fde_fp4), one workgroup per row
    int rc4 = launch_fde_quantize_fp4(ix->fde + (size_t)first * od, od, n, ix->fde4 + (size_t)first * (od / 2), ix->fde4_scale + first, ix->fde_inv_norm + first,
                                      ix->fde4_cfac + first, st);
    if (rc4) return rc4;
  }
  if (!ix->fde8) return MV_OK;
  int rc = launch_quantize_pages_fp8(ix->fde + (size_t)first * od, nullptr, (int32_t)(od / kDim), n, ix->fde8 + (size_t)first * od, ix->fde8_scale + first, st);
  if (rc) return rc;
  hipLaunchKernelGGL(fde8_cfac_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)(ix->fde8_scale + first), (const float*)(ix->fde_inv_norm + first), n,
                     ix->fde8_cfac + first);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// FDE coarse stage: encode the query already uploaded to d_qf32 (SUM aggregation) and scan the FDE slab -> d_scores[n].
int fde_coarse_scan(mv_index* ix, int n_q, const uint32_t* d_allow, int64_t n_words, int64_t n, int* launches, bool stage_events, int32_t k_next,
                    bool* hist0_done) {
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  FdeEncodeArgs e{};
  e.variant = ix->fde_query_encode_variant;
  // one page of n_q rows: the row count travels as a kernel argument (no 16-byte H2D copy in front of every query)
  e.x_f32 = ix->d_qf32; e.row_offsets = nullptr; e.stride = n_q; e.n_pages = 1; e.is_query = 1; e.out_f32 = ix->d_qfde;
  int rc = MV_OK;
  if (const float* ov = query_fde_override(ix->fde_t.out_dim))  // mv_query_topk_fde: the caller's own encoding of this query
    MV_HIP(hipMemcpyAsync(ix->d_qfde, ov, (size_t)ix->fde_t.out_dim * 4, hipMemcpyHostToDevice, ix->stream));
  else
    rc = launch_fde_encode(ix->fde_t, e, ix->stream);
  if (rc) return rc;
  if (stage_events) MV_HIP(hipEventRecord(ix->ev_st[0], ix->stream));
  FdeScanArgs s{};
  s.fde = ix->fde; s.inv_norm = ix->fde_cosine ? ix->fde_inv_norm : nullptr; s.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
  s.allow = d_allow; s.n_allow_bits = n_words * 32; s.q = ix->d_qfde; s.scores = ix->d_scores; s.n = n;
  s.out_dim = ix->fde_t.out_dim;
  if (ix->fde4 && ix->fde_coarse_e4m3 == 2) {  // MV_WITH_FDE_FP4: the same stage over a quarter of the bytes (the selection bins the scores itself)
    FdeScan8Args s4{};
    s4.fde8 = ix->fde4; s4.scale = ix->fde_cosine ? ix->fde4_cfac : ix->fde4_scale; s4.inv_norm = nullptr; s4.doc_ord = s.doc_ord; s4.allow = s.allow;
    s4.n_allow_bits = s.n_allow_bits; s4.q = s.q; s4.scores = s.scores; s4.n = n; s4.out_dim = s.out_dim;
    if (hist0_done) *hist0_done = false;
    rc = launch_fde_scan4(s4, ix->stream);
    if (rc) return rc;
    if (stage_events) MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
    *launches += 2;
    return MV_OK;
  }
  if (ix->fde8 && ix->fde_coarse_e4m3 == 1) {  // MV_WITH_FDE_E4M3: the same stage over half the bytes (the selection bins the scores itself)
    FdeScan8Args s8{};
    s8.fde8 = ix->fde8; s8.scale = ix->fde8_scale; s8.inv_norm = s.inv_norm; s8.doc_ord = s.doc_ord; s8.allow = s.allow; s8.n_allow_bits = s.n_allow_bits;
    s8.q = s.q; s8.scores = s.scores; s8.n = n; s8.out_dim = s.out_dim;
    if (hist0_done) *hist0_done = false;
    rc = launch_fde_scan8(s8, ix->stream);
    if (rc) return rc;
    if (stage_events) MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
    *launches += 2;
    return MV_OK;
  }
  const bool prebin = k_next > 0 && hist0_done && topk_uses_radix(n, k_next) && fde_scan_prebins(ix->fde_scan_variant, s.out_dim);
  s.hist0 = prebin ? topk_radix_hist0(ix->d_topk_ws) : nullptr;  // zero between selections (cleared by the previous one's last kernel)
  if (hist0_done) *hist0_done = prebin;
  rc = launch_fde_scan(s, ix->fde_scan_variant, ix->stream);
  if (rc) return rc;
  if (stage_events) MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
  *launches += 2;
  return MV_OK;
}

// Exact rerank of the candidate list in d_cand / d_cand_pads (built by launch_cand_prepare or the owned-select kernel).
// The candidates were named explicitly or chosen from live, allowed pages: no mask is applied.
int rerank_scan(mv_index* ix, int n_q, int tier, int64_t n_items, float* d_out, int* launches) {
  if (tier == kTierFp8) return fp8_scan(ix, n_q, nullptr, 0, ix->d_cand, n_items, 0, ix->d_cand_pads, d_out, launches, /*no_mask=*/true);
  return exact_scan(ix, n_q, tier, ix->d_cand, n_items, 0, ix->d_cand_pads, d_out, launches);
}

// ---- the exact host tier, possibly split: pages [0, x_split) in HBM (slab_x), pages [x_split, capacity) in pinned host memory
// loc (nullable): page -> slot of the tier after a rebalance (identity otherwise); a slot below `split` is HBM.  hits (nullable): one
// count per page whose exact rows this rerank reads -- what mv_index_exact_tier_rebalance ranks pages by.
__global__ __launch_bounds__(256) void split_cand_kernel(const int32_t* cand, int64_t n, int32_t split, int32_t* lo, int32_t* hi, const int32_t* loc,
                                                         uint32_t* hits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t c = cand[i];
  const int32_t s = c < 0 ? -1 : (loc ? loc[c] : c);
  lo[i] = (c >= 0 && s < split) ? c : -1;
  hi[i] = (c >= 0 && s >= split) ? c : -1;
  if (hits && c >= 0) atomicAdd(&hits[c], 1u);
}
__global__ __launch_bounds__(256) void max_scores_kernel(float* dst, const float* src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = fmaxf(dst[i], src[i]);  // every entry was scored by exactly one part; the other wrote -inf
}
static inline size_t page_elems(const mv_index* ix) { return (size_t)ix->cfg.stride_rows * kDim; }
// base pointer that makes `base + page * stride * 128` the rows of page >= x_split in the host part
static inline const uint16_t* xt_host_vbase(const mv_index* ix) { return ix->d_exact - (size_t)ix->x_split * page_elems(ix); }

int ensure_split_ws(mv_index* ix) {
  if (!ix->d_xcand) MV_HIP(hipMalloc(&ix->d_xcand, (size_t)2 * kMaxCand * 4));
  if (!ix->d_xscores) MV_HIP(hipMalloc(&ix->d_xscores, (size_t)kMaxCand * 4));
  return MV_OK;
}

int exact_scan(mv_index* ix, int n_q, int tier, const int32_t* d_cand, int64_t n_items, int32_t pad_to, const int32_t* d_pad_items, float* d_out,
               int* launches, const uint16_t* d_q_base, const uint16_t* d_qlo_base) {
  if (n_items <= 0) return MV_OK;
  // the bf16 slab in HBM: with its lo half when the index keeps one (MV_WITH_FLOAT_LO) -- the fp32-faithful rerank
  if (tier == kTierSlab) return float_scan(ix, n_q, nullptr, 0, d_cand, n_items, pad_to, d_pad_items, d_out, launches, /*no_mask=*/true, d_q_base, ix->slab, d_qlo_base, ix->slab_lo);
  const int64_t cap = ix->cfg.capacity_pages;
  if (ix->x_split <= 0) return float_scan(ix, n_q, nullptr, 0, d_cand, n_items, pad_to, d_pad_items, d_out, launches, true, d_q_base, ix->d_exact, d_qlo_base);
  if (ix->x_split >= cap) return float_scan(ix, n_q, nullptr, 0, d_cand, n_items, pad_to, d_pad_items, d_out, launches, true, d_q_base, ix->slab_x, d_qlo_base);
  // split tier: the candidates of each part against that part's base, two launches, merged (an entry is -1 in exactly one of the two lists)
  if (n_items > kMaxCand) { set_error("exact_scan: %lld candidates exceed %d", (long long)n_items, kMaxCand); return MV_ERR_INVALID; }
  int rc = ensure_split_ws(ix);
  if (rc) return rc;
  int32_t* lo = ix->d_xcand;
  int32_t* hi = ix->d_xcand + kMaxCand;
  const unsigned gb = (unsigned)((n_items + 255) / 256);
  hipLaunchKernelGGL(split_cand_kernel, dim3(gb), dim3(256), 0, ix->stream, d_cand, n_items, (int32_t)ix->x_split, lo, hi, (const int32_t*)ix->d_xloc, ix->d_xhits);
  // one row-offset table serves both parts: slot * stride rows from slab_x, or from the host part's virtual base (slot x_split = its row 0)
  rc = float_scan(ix, n_q, nullptr, 0, lo, n_items, pad_to, d_pad_items, d_out, launches, true, d_q_base, ix->slab_x, d_qlo_base, nullptr, ix->d_xoff);
  if (rc) return rc;
  rc = float_scan(ix, n_q, nullptr, 0, hi, n_items, pad_to, d_pad_items, ix->d_xscores, launches, true, d_q_base, xt_host_vbase(ix), d_qlo_base, nullptr, ix->d_xoff);
  if (rc) return rc;
  hipLaunchKernelGGL(max_scores_kernel, dim3(gb), dim3(256), 0, ix->stream, d_out, (const float*)ix->d_xscores, n_items);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

static inline int64_t xt_slot(const mv_index* ix, int64_t page) { return ix->x_loc.empty() ? page : ix->x_loc[(size_t)page]; }

// Store the fixed-stride bf16 image of pages [first, first + n) (device memory `d_src`) into the exact host tier on `st`.
// (New pages: slot == page -- only published pages are ever re-placed, so the slots behind the published prefix are free.)
int xt_store_from_device(mv_index* ix, const uint16_t* d_src, int64_t first, int64_t n, hipStream_t st) {
  if (!ix->h_exact && !ix->slab_x) return MV_OK;
  const size_t pe = page_elems(ix);
  const int64_t n_hbm = std::max<int64_t>(0, std::min<int64_t>(first + n, ix->x_split) - first);  // leading pages of the range that fall into the HBM part
  if (n_hbm > 0) MV_HIP(hipMemcpyAsync(ix->slab_x + (size_t)first * pe, d_src, (size_t)n_hbm * pe * 2, hipMemcpyDeviceToDevice, st));
  if (n > n_hbm) MV_HIP(hipMemcpyAsync(ix->h_exact + (size_t)(first + n_hbm - ix->x_split) * pe, d_src + (size_t)n_hbm * pe, (size_t)(n - n_hbm) * pe * 2, hipMemcpyDeviceToHost, st));
  return MV_OK;
}
// Overwrite rows [row0, row0 + n_rows) of one page from HOST bf16 rows (zero_rest: the page's other rows become zero).
int xt_store_rows_from_host(mv_index* ix, int64_t page, int32_t row0, int32_t n_rows, const void* bf16_rows, bool zero_rest) {
  if (!ix->h_exact && !ix->slab_x) return MV_OK;
  const size_t pe = page_elems(ix);
  page = xt_slot(ix, page);  // where the page's rows live (a rebalanced split tier)
  if (page < ix->x_split) {
    uint16_t* dst = ix->slab_x + (size_t)page * pe;
    if (zero_rest) MV_HIP(hipMemsetAsync(dst, 0, pe * 2, ix->stream));
    if (n_rows > 0) MV_HIP(hipMemcpyAsync(dst + (size_t)row0 * kDim, bf16_rows, (size_t)n_rows * kRowBytes, hipMemcpyHostToDevice, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));  // the caller's buffer may be pageable
  } else {
    uint16_t* dst = ix->h_exact + (size_t)(page - ix->x_split) * pe;
    if (zero_rest) memset(dst, 0, pe * 2);
    if (n_rows > 0) memcpy(dst + (size_t)row0 * kDim, bf16_rows, (size_t)n_rows * kRowBytes);
  }
  return MV_OK;
}
// Copy whole pages [page0, page0 + n) of the tier to a HOST buffer.
int xt_read_pages(mv_index* ix, int64_t page0, int64_t n, void* out) {
  const size_t pe = page_elems(ix);
  if (!ix->x_loc.empty()) {  // rebalanced: page by page through the placement table
    for (int64_t i = 0; i < n; ++i) {
      const int64_t s = ix->x_loc[(size_t)(page0 + i)];
      char* o = (char*)out + (size_t)i * pe * 2;
      if (s < ix->x_split) MV_HIP(hipMemcpy(o, ix->slab_x + (size_t)s * pe, pe * 2, hipMemcpyDeviceToHost));
      else memcpy(o, ix->h_exact + (size_t)(s - ix->x_split) * pe, pe * 2);
    }
    return MV_OK;
  }
  const int64_t n_hbm = std::max<int64_t>(0, std::min<int64_t>(page0 + n, ix->x_split) - page0);
  if (n_hbm > 0) MV_HIP(hipMemcpy(out, ix->slab_x + (size_t)page0 * pe, (size_t)n_hbm * pe * 2, hipMemcpyDeviceToHost));
  if (n > n_hbm) memcpy((char*)out + (size_t)n_hbm * pe * 2, ix->h_exact + (size_t)(page0 + n_hbm - ix->x_split) * pe, (size_t)(n - n_hbm) * pe * 2);
  return MV_OK;
}
// Compaction of the tier: page live[j] moves to slot j for j in [first_moved, m) (live ascending, live[j] >= j: a destination is
// never a page still to be read).  Device destinations are stream-ordered copies (from the HBM part or up from the pinned part),
// then the host part moves with memmove.
static int xt_restore_identity(mv_index* ix);
int xt_compact(mv_index* ix, const std::vector<int64_t>& live, int64_t first_moved, int64_t m) {
  if (!ix->h_exact && !ix->slab_x) return MV_OK;
  if (int rc = xt_restore_identity(ix)) return rc;  // a rebalanced tier goes back to slot == page first (the next rebalance re-places the hot pages)
  const size_t pe = page_elems(ix), pb = pe * 2;
  const int64_t sp = ix->x_split;
  for (int64_t j = first_moved; j < std::min(m, sp); ++j) {
    const int64_t src = live[(size_t)j];
    if (src == j) continue;
    if (src < sp) MV_HIP(hipMemcpyAsync(ix->slab_x + (size_t)j * pe, ix->slab_x + (size_t)src * pe, pb, hipMemcpyDeviceToDevice, ix->stream));
    else MV_HIP(hipMemcpyAsync(ix->slab_x + (size_t)j * pe, ix->h_exact + (size_t)(src - sp) * pe, pb, hipMemcpyHostToDevice, ix->stream));
  }
  MV_HIP(hipStreamSynchronize(ix->stream));  // before the host part below overwrites pages the copies above read
  for (int64_t j = std::max(first_moved, sp); j < m; ++j)
    if (live[(size_t)j] != j) memmove(ix->h_exact + (size_t)(j - sp) * pe, ix->h_exact + (size_t)(live[(size_t)j] - sp) * pe, pb);
  return MV_OK;
}

// ---- placement of a split exact tier: hot pages in the HBM part
static inline uint16_t* xt_slot_ptr(mv_index* ix, int64_t slot) {
  const size_t pe = page_elems(ix);
  return slot < ix->x_split ? ix->slab_x + (size_t)slot * pe : ix->h_exact + (size_t)(slot - ix->x_split) * pe;
}
static inline hipMemcpyKind xt_kind(const mv_index* ix, int64_t dst_slot, int64_t src_slot) {
  const bool dd = dst_slot < ix->x_split, sd = src_slot < ix->x_split;
  return dd ? (sd ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) : (sd ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
}
static int xt_ensure_tables(mv_index* ix) {
  if (!ix->x_loc.empty()) return MV_OK;
  const int64_t cap = ix->cfg.capacity_pages;
  ix->x_loc.resize((size_t)cap);
  ix->x_page_at.resize((size_t)cap);
  for (int64_t p = 0; p < cap; ++p) ix->x_loc[(size_t)p] = ix->x_page_at[(size_t)p] = (int32_t)p;
  if (!ix->d_xloc) MV_HIP(hipMalloc(&ix->d_xloc, (size_t)cap * 4));
  if (!ix->d_xoff) MV_HIP(hipMalloc(&ix->d_xoff, (size_t)cap * 8));
  return MV_OK;
}
static int xt_upload_tables(mv_index* ix) {
  const int64_t cap = ix->cfg.capacity_pages;
  std::vector<int64_t> off((size_t)cap);
  for (int64_t p = 0; p < cap; ++p) off[(size_t)p] = (int64_t)ix->x_loc[(size_t)p] * ix->cfg.stride_rows;
  MV_HIP(hipMemcpy(ix->d_xloc, ix->x_loc.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
  MV_HIP(hipMemcpy(ix->d_xoff, off.data(), (size_t)cap * 8, hipMemcpyHostToDevice));
  return MV_OK;
}
// Exchange the contents of slot pairs (any mix of HBM and pinned-host slots) through a ring of device staging pages, stream-ordered;
// the tables follow.  Exclusive access (both locks) is the caller's.
static int xt_swap_slots(mv_index* ix, const std::vector<std::pair<int64_t, int64_t>>& pairs) {
  if (pairs.empty()) return MV_OK;
  const size_t pb = page_elems(ix) * 2;
  const int ring = (int)std::min<size_t>(pairs.size(), 64);
  char* stg = nullptr;
  if (hipMalloc(&stg, (size_t)ring * pb) != hipSuccess) { set_error("exact tier placement: out of device memory for %zu bytes of staging", (size_t)ring * pb); return MV_ERR_NOMEM; }
  int rc = MV_OK;
  for (size_t i = 0; i < pairs.size() && !rc; ++i) {
    const int64_t a = pairs[i].first, b = pairs[i].second;
    char* t = stg + (i % ring) * pb;
    if (i && i % ring == 0 && hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("exact tier placement: stream error"); rc = MV_ERR_HIP; break; }
    if (a >= ix->x_split && b >= ix->x_split) {
      // both in pinned host memory (only when a composed placement is undone): a host swap, ordered behind the stream's copies
      if (hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("exact tier placement: stream error"); rc = MV_ERR_HIP; break; }
      std::vector<char> tmp(pb);
      memcpy(tmp.data(), xt_slot_ptr(ix, a), pb);
      memcpy(xt_slot_ptr(ix, a), xt_slot_ptr(ix, b), pb);
      memcpy(xt_slot_ptr(ix, b), tmp.data(), pb);
    } else {
      hipError_t e = hipMemcpyAsync(t, xt_slot_ptr(ix, a), pb, a < ix->x_split ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ix->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(xt_slot_ptr(ix, a), xt_slot_ptr(ix, b), pb, xt_kind(ix, a, b), ix->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(xt_slot_ptr(ix, b), t, pb, b < ix->x_split ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ix->stream);
      if (e != hipSuccess) { rc = hip_fail(e, "exact tier placement copy", __FILE__, __LINE__); break; }
    }
    const int32_t pa = ix->x_page_at[(size_t)a], pbg = ix->x_page_at[(size_t)b];
    ix->x_page_at[(size_t)a] = pbg; ix->x_page_at[(size_t)b] = pa;
    ix->x_loc[(size_t)pbg] = (int32_t)a; ix->x_loc[(size_t)pa] = (int32_t)b;
  }
  if (hipStreamSynchronize(ix->stream) != hipSuccess && !rc) { set_error("exact tier placement: stream error"); rc = MV_ERR_HIP; }
  (void)hipFree(stg);
  if (!rc) rc = xt_upload_tables(ix);
  return rc;
}
// Back to slot == page (compaction and the bulk checkpoint paths assume it): every swap puts at least one page home.
static int xt_restore_identity(mv_index* ix) {
  if (ix->x_loc.empty()) return MV_OK;
  const int64_t n = ix->size.load();
  std::vector<int32_t> at(ix->x_page_at.begin(), ix->x_page_at.begin() + n);  // simulate, collect the swaps, apply them in one go
  std::vector<std::pair<int64_t, int64_t>> pairs;
  for (int64_t s0 = 0; s0 < n; ++s0)
    while (at[(size_t)s0] != s0) {
      const int64_t p = at[(size_t)s0];  // the page in slot s0 belongs into slot p
      pairs.emplace_back(s0, p);
      std::swap(at[(size_t)s0], at[(size_t)p]);
    }
  int rc = xt_swap_slots(ix, pairs);
  if (rc) return rc;
  // identity again: drop the tables (the kernels fall back to page * stride)
  ix->x_loc.clear(); ix->x_page_at.clear();
  if (ix->d_xloc) { (void)hipFree(ix->d_xloc); ix->d_xloc = nullptr; }
  if (ix->d_xoff) { (void)hipFree(ix->d_xoff); ix->d_xoff = nullptr; }
  return MV_OK;
}

// The tier a rerank reads and whether an e4m3 stage prunes the list first (mv_index_priv.h).  One rule for every entry point
// (single / batched / sharded), so the same request takes the same path whatever carries it.
RerankPlan rerank_plan(const mv_index* ix, int mode, int64_t n_list, int32_t k, int rpq, bool batched) {
  RerankPlan p;
  const int fl = ix->cfg.flags;
  const bool force_fp8 = ix->exact_tier == 2 && (fl & MV_WITH_FP8) && mode != MV_MODE_FP8_THEN_FLOAT;  // MV_OPT_EXACT_TIER 2: score on the e4m3 slab although an exact tier exists
  const bool host = !force_fp8 && (fl & MV_WITH_HOST_EXACT) && (!(fl & MV_WITH_FLOAT) || ix->exact_tier == 1);
  p.host_tier = host;
  p.final_fp8 = force_fp8 || (!host && !(fl & MV_WITH_FLOAT));
  p.tier = p.final_fp8 ? kTierFp8 : (host ? kTierHost : kTierSlab);
  const int64_t n_mid = std::max<int64_t>(ix->rerank_n, k);
  // the e4m3 scan of MV_MODE_FP8_THEN_FLOAT already was the pruning stage; the one-launch e4m3 rerank of a batch takes <= 64 rows
  p.mid = mode == MV_MODE_FDE_THEN_FLOAT && host && (fl & MV_WITH_FP8) && n_list > n_mid && (!batched || rpq <= 64);
  p.n_mid = p.mid ? (int32_t)n_mid : 0;
  return p;
}

// One block per list (lists are <= kTopkMaxDeviceK = 1024 entries): flag the selected positions in LDS, clear the rest.
__global__ __launch_bounds__(1024) void keep_selected_kernel(const int64_t* pos, int64_t pos_stride, int n_sel, int32_t* cand,
                                                             int64_t cand_stride, int n) {
  __shared__ uint8_t keep[kTopkMaxDeviceK];
  pos += (int64_t)blockIdx.x * pos_stride;
  cand += (int64_t)blockIdx.x * cand_stride;
  const int t = threadIdx.x;
  keep[t] = 0;
  __syncthreads();
  if (t < n_sel) {
    const int64_t p = pos[t];
    if (p >= 0 && p < n) keep[p] = 1;
  }
  __syncthreads();
  if (t < n && !keep[t]) cand[t] = -1;
}

int launch_keep_selected(const int64_t* d_pos, int64_t pos_stride, int n_sel, int32_t* d_cand, int64_t cand_stride, int n, int nb, hipStream_t s) {
  if (n < 1 || nb < 1) return MV_OK;
  if (n > kTopkMaxDeviceK || n_sel > kTopkMaxDeviceK) { set_error("keep_selected: lists of %d / %d entries exceed %d", n, n_sel, kTopkMaxDeviceK); return MV_ERR_INVALID; }
  hipLaunchKernelGGL(keep_selected_kernel, dim3((unsigned)nb), dim3(1024), 0, s, d_pos, pos_stride, n_sel, d_cand, cand_stride, n);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// Core of every query entry point: leaves per-item scores on the device.  Caller holds q_mu.
int run_scan(mv_index* ix, const void* q, int q_dtype, int n_q, int mode, const uint32_t* allow_bits, int64_t n_words,
             int64_t want_coarse, ScanResult* out, mv_query_stats* st, bool want_compact = false, int32_t k_final = 0) {
  if (!q || n_q <= 0) { set_error("query must have at least one row"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_q * kDim, mode)) return frc;
  const bool want_fde = mode == MV_MODE_FDE_THEN_FLOAT || mode == MV_MODE_FDE_ONLY;
  const bool want_bin = mode == MV_MODE_BINARY;
  const bool two_tier = mode == MV_MODE_FP8_THEN_FLOAT;
  // The rerank stage of FDE_THEN_FLOAT reads the index's exact tier (the bf16 slab, or the pinned-host tier: through an e4m3
  // pruning stage when the list is longer than MV_OPT_RERANK_N); an index with neither reranks on the e4m3 slab -- the best
  // copy it holds, NOT the reference's exact fp32 rerank (fast_multivector_store.py:553-556).
  const int64_t n_pub = ix->size.load(std::memory_order_acquire);
  const int64_t nc_fde = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want_coarse, n_pub), kTopkMaxDeviceK));
  const RerankPlan plan = mode == MV_MODE_FDE_THEN_FLOAT ? rerank_plan(ix, mode, nc_fde, k_final, ((n_q + 15) / 16) * 16, false) : RerankPlan{};
  const bool rerank_fp8 = mode == MV_MODE_FDE_THEN_FLOAT && plan.final_fp8;
  const bool want_fp8 = mode == MV_MODE_FLOAT_FP8 || rerank_fp8 || plan.mid || two_tier;
  const bool want_float = mode == MV_MODE_FLOAT || (mode == MV_MODE_FDE_THEN_FLOAT && !rerank_fp8) || two_tier;
  if (!want_float && !want_fde && !want_bin && !want_fp8) { set_error("unknown mode %d", mode); return MV_ERR_INVALID; }
  if (two_tier && !(ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT))) { set_error("MV_MODE_FP8_THEN_FLOAT needs an exact tier (MV_WITH_FLOAT or MV_WITH_HOST_EXACT)"); return MV_ERR_STATE; }
  if (mode == MV_MODE_FLOAT && !(ix->cfg.flags & MV_WITH_FLOAT)) { set_error("index has no float slab (MV_WITH_FLOAT)"); return MV_ERR_STATE; }
  if (want_fp8 && !(ix->cfg.flags & MV_WITH_FP8)) {
    set_error(rerank_fp8 ? "MV_MODE_FDE_THEN_FLOAT needs a copy of the pages to rerank on (MV_WITH_FLOAT, MV_WITH_HOST_EXACT or MV_WITH_FP8)" : "index has no fp8 slab (MV_WITH_FP8)");
    return MV_ERR_STATE;
  }
  if (want_bin && !(ix->cfg.flags & MV_WITH_BINARY)) { set_error("index has no sign-bit slab (MV_WITH_BINARY)"); return MV_ERR_STATE; }
  if (want_fde && !(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }

  // snapshot of the published corpus: pages appended while this query runs are not seen
  const int64_t n = n_pub;
  const bool ragged = ix->ragged.load();
  int rc = upload_query(ix, q, q_dtype, n_q, want_float, want_fde, want_bin, want_fp8);
  if (rc) return rc;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, n_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  int64_t pages = 0, rows = 0;  // accounting only (filled below, once it is known whether the filter was compacted)
  bool rr_lo = false, lo_cascade = false;  // a rerank stage read the candidates' lo rows too / MV_OPT_FLOAT_LO_SCAN 2 ran its rerank

  // Stage events only when somebody will read them: every hipEventRecord between two dependent kernels costs ~5.8 us of device
  // time on this stack (rocprofv3 kernel trace of one request, profiles/r5: gaps of 0.0 us between kernels with no event between
  // them, 5.7-6.1 us with one), five of them on a 75-candidate FDE request.
  if (st) MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
  out->launches = 0;
  // Selective doc filter on a top-k scan: compact the allowed pages (in page order) and scan only those.
  const int32_t* d_scan_cand = nullptr;
  int64_t n_scan = n;
  const int32_t max_ord = ix->max_doc_ord.load();
  if (want_compact && d_allow && (mode == MV_MODE_FLOAT || mode == MV_MODE_FLOAT_FP8 || mode == MV_MODE_BINARY || two_tier) && ix->filter_compact_pct > 0 && max_ord >= 0) {
    int64_t allowed_docs = 0;
    for (int64_t w = 0; w < n_words; ++w) allowed_docs += __builtin_popcount(allow_bits[w]);
    if (allowed_docs * 100 < (int64_t)ix->filter_compact_pct * ((int64_t)max_ord + 1)) {
      if (!ix->d_fcand) {
        MV_HIP(hipMalloc(&ix->d_fcand, (size_t)ix->cfg.capacity_pages * 4));
        MV_HIP(hipMalloc(&ix->d_fcounts, filter_ws_bytes(ix->cfg.capacity_pages)));
      }
      rc = launch_filter_compact(ix->d_doc_ord, d_allow, n_words * 32, n, ix->d_fcounts, ix->d_fcand, &n_scan, ix->stream);
      if (rc) return rc;
      d_scan_cand = ix->d_fcand;
      out->launches += 3;
    }
  }
  if (st) {
    if (d_scan_cand && !ragged) { pages = n_scan; rows = n_scan * (int64_t)ix->cfg.stride_rows; }  // the device already counted
    else rows = count_allowed_rows(ix, n, allow_bits, n_words, &pages);
  }
  if (mode == MV_MODE_FLOAT) {
    // an index with a lo slab (MV_WITH_FLOAT_LO): MV_OPT_FLOAT_LO_SCAN 1 reads both halves of every page (fp32-faithful scores),
    // 0 the hi half only, 2 (with a selection to follow: k_final > 0) the hi half and re-scores the best max(MV_OPT_RERANK_N, k)
    const bool cascade = ix->slab_lo && ix->float_lo_scan == 2 && k_final > 0;
    const uint16_t* scan_lo = (ix->slab_lo && (ix->float_lo_scan == 1 || (ix->float_lo_scan == 2 && !cascade))) ? ix->slab_lo : nullptr;
    if (cascade && st) MV_HIP(hipEventRecord(ix->ev_st[0], ix->stream));
    rc = d_scan_cand ? float_scan(ix, n_q, nullptr, 0, d_scan_cand, n_scan, 0, nullptr, ix->d_scores, &out->launches, true, nullptr, nullptr, nullptr, scan_lo)
                     : float_scan(ix, n_q, d_allow, n_words, nullptr, n, 0 /* full scan: no padding rows exist */, nullptr, ix->d_scores, &out->launches, false, nullptr, nullptr, nullptr, scan_lo);
    if (rc) return rc;
    out->d_scores = ix->d_scores; out->n = n_scan; out->d_ids_map = d_scan_cand;
    out->pages = pages; out->bytes = rows * (int64_t)kRowBytes * (scan_lo ? 2 : 1);
    if (cascade) {
      if (st) MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
      const int64_t nc = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(std::max<int64_t>(ix->rerank_n, k_final), n_scan), kTopkMaxDeviceK));
      if (n_scan > 0) {
        rc = launch_topk(ix->d_scores, n_scan, (int32_t)nc, d_scan_cand, 0, ix->d_topk_ws, ix->d_out_s, ix->d_out_id, ix->stream);
        if (rc) return rc;
        rc = launch_cand_prepare(ix, ix->d_out_id, nullptr, (int)nc, /*pad_sem=*/0);
        if (rc) return rc;
        if (st) MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
        rc = exact_scan(ix, n_q, kTierSlab, ix->d_cand, nc, 0, ix->d_cand_pads, ix->d_cand_scores, &out->launches);
        if (rc) return rc;
        rr_lo = lo_cascade = true;
        out->launches += 2;
        out->n = nc; out->d_ids_map = ix->d_cand; out->d_scores = ix->d_cand_scores;
      } else if (st) MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
    }
  } else if (mode == MV_MODE_FLOAT_FP8) {
    rc = d_scan_cand ? fp8_scan(ix, n_q, nullptr, 0, d_scan_cand, n_scan, 0, nullptr, ix->d_scores, &out->launches, true)
                     : fp8_scan(ix, n_q, d_allow, n_words, nullptr, n, 0, nullptr, ix->d_scores, &out->launches);
    if (rc) return rc;
    out->d_scores = ix->d_scores; out->n = n_scan; out->d_ids_map = d_scan_cand;
    out->pages = pages; out->bytes = rows * (int64_t)kDim;
  } else if (two_tier) {
    // e4m3 scan of every (allowed) page -> top-n -> exact bf16 re-score of the n candidates from the exact tier -> the
    // caller's top-k.  One stream-ordered chain: the candidate ids never leave the device, and with a pinned-host exact
    // tier the rerank kernel's own LDS-DMA reads fetch the n pages over PCIe (n x 256 KiB; no staging copy).
    if (st) MV_HIP(hipEventRecord(ix->ev_st[0], ix->stream));
    rc = d_scan_cand ? fp8_scan(ix, n_q, nullptr, 0, d_scan_cand, n_scan, 0, nullptr, ix->d_scores, &out->launches, true)
                     : fp8_scan(ix, n_q, d_allow, n_words, nullptr, n, 0, nullptr, ix->d_scores, &out->launches);
    if (rc) return rc;
    if (st) MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
    out->pages = pages; out->bytes = rows * (int64_t)kDim;
    const int64_t nc = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want_coarse, n_scan), kTopkMaxDeviceK));  // want_coarse = max(MV_OPT_RERANK_N, k)
    if (n_scan > 0) {
      // local page ids (id_base 0) of the fp8 top-n, through the compacted filter list when there is one
      rc = launch_topk(ix->d_scores, n_scan, (int32_t)nc, d_scan_cand, 0, ix->d_topk_ws, ix->d_out_s, ix->d_out_id, ix->stream);
      if (rc) return rc;
      rc = launch_cand_prepare(ix, ix->d_out_id, nullptr, (int)nc, /*pad_sem=*/0);  // a full-corpus scan has no padding rows
      if (rc) return rc;
      if (st) MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
      const int rr_tier = rerank_plan(ix, mode, nc, k_final, ((n_q + 15) / 16) * 16, false).tier;
      rc = exact_scan(ix, n_q, rr_tier, ix->d_cand, nc, 0, ix->d_cand_pads, ix->d_cand_scores, &out->launches);
      if (rc) return rc;
      rr_lo = rr_tier == kTierSlab && ix->slab_lo != nullptr;
      out->launches += 2;
      out->n = nc; out->d_ids_map = ix->d_cand; out->d_scores = ix->d_cand_scores;
    } else {
      out->n = 0; out->d_ids_map = nullptr; out->d_scores = ix->d_scores;
      if (st) MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
    }
  } else if (mode == MV_MODE_BINARY) {
    BinaryArgs b{};
    b.bits = ix->bits; b.n_rows = ragged ? ix->d_n_rows : nullptr;
    // compacted filter: the list holds only live, allowed pages -- no per-page mask test left to do
    b.doc_ord = (need_meta && !d_scan_cand) ? ix->d_doc_ord : nullptr;
    b.allow = d_scan_cand ? nullptr : d_allow; b.n_allow_bits = n_words * 32;
    b.qbits = ix->d_qbits; b.qpop_rw = ix->d_qpop; b.qpop = ix->d_qpop; b.scores = ix->d_scores;
    b.n = n_scan; b.cand = d_scan_cand;
    b.stride = ix->cfg.stride_rows; b.n_q = n_q; b.row_off = ix->d_row_off;
    if (n_scan > 0) {
      rc = launch_maxsim_binary(b, ix->binary_variant, ix->stream);
      if (rc) return rc;
    }
    out->launches += 1;
    out->d_scores = ix->d_scores; out->n = n_scan; out->d_ids_map = d_scan_cand;
    out->pages = pages; out->bytes = rows * (int64_t)kSignBytes;
  } else {
    // FDE: encode the query (SUM), scan the FDE slab
    const int64_t nc = nc_fde;
    bool hist0_done = false;
    rc = fde_coarse_scan(ix, n_q, d_allow, n_words, n, &out->launches, st != nullptr, mode == MV_MODE_FDE_THEN_FLOAT ? (int32_t)nc : 0, &hist0_done);
    if (rc) return rc;
    out->pages = pages; out->bytes = pages * ix->fde_t.out_dim * 2;
    if (mode == MV_MODE_FDE_ONLY) {
      out->d_scores = ix->d_scores; out->n = n; out->d_ids_map = nullptr;
    } else {
      // coarse top-n -> candidate list -> exact rerank, all in stream order: the selection kernels leave the n local
      // page ids on the device, cand_prepare turns them into the rerank list + per-batch pad lengths, the rerank
      // kernel skips the (-1) padding entries.  No host round trip between the stages.
      rc = launch_topk(ix->d_scores, n, (int32_t)nc, nullptr, 0, ix->d_topk_ws, ix->d_out_s, ix->d_out_id, ix->stream, hist0_done);
      if (rc) return rc;
      // reference rule: pad_sequence over each rerank batch (<= 128 pages): shorter pages see zero rows
      const int pad_sem = ix->pad_semantics < 0 ? 1 : ix->pad_semantics;
      rc = launch_cand_prepare(ix, ix->d_out_id, nullptr, (int)nc, pad_sem);
      if (rc) return rc;
      if (st) MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
      if (plan.mid) {
        // pinned-host exact tier behind a long list: e4m3 scores of all nc candidates (HBM) -> the n_mid best positions ->
        // every other entry of the list becomes -1; the list keeps its order, so the pad lengths (the reference's batches of
        // 128 over the WHOLE coarse list) and the tie rule (coarse rank) are those of the direct rerank
        if (st) MV_HIP(hipMemcpyAsync(ix->h_cand + kTopkMaxDeviceK, ix->d_cand, (size_t)nc * 4, hipMemcpyDeviceToHost, ix->stream));
        rc = rerank_scan(ix, n_q, kTierFp8, nc, ix->d_cand_scores, &out->launches);
        if (rc) return rc;
        rc = launch_topk(ix->d_cand_scores, nc, plan.n_mid, nullptr, 0, ix->d_topk_ws, ix->d_out_s, ix->d_sel_pos, ix->stream);
        if (rc) return rc;
        rc = launch_keep_selected(ix->d_sel_pos, 0, plan.n_mid, ix->d_cand, 0, (int)nc, 1, ix->stream);
        if (rc) return rc;
        out->launches += 2;
      }
      rc = rerank_scan(ix, n_q, plan.tier, nc, ix->d_cand_scores, &out->launches);
      if (rc) return rc;
      rr_lo = plan.tier == kTierSlab && ix->slab_lo != nullptr;
      out->launches += 1;
      out->n = nc;
      out->d_ids_map = ix->d_cand;
      out->d_scores = ix->d_cand_scores;
    }
  }
  if (st) MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
  if (st) {
    memset(st, 0, sizeof(*st));
    st->score_launches = out->launches;
    st->pages_scored = out->pages;
    st->bytes_scanned = out->bytes;
    // FDE_THEN_FLOAT: the candidates' rows are added by finish_stats (read back behind the timed span)
    if (mode == MV_MODE_FDE_THEN_FLOAT || ((two_tier || lo_cascade) && out->n > 0)) st->reserved = (int32_t)out->n | (rerank_fp8 ? (1 << 30) : 0) | (plan.mid ? (1 << 28) : 0) | (rr_lo ? (1 << 27) : 0);
    else if (mode == MV_MODE_FDE_ONLY) st->reserved = 1 << 29;  // stage split without a rerank
  }
  return MV_OK;
}

int finish_stats(mv_index* ix, mv_query_stats* st, bool had_topk) {
  if (!st) return MV_OK;
  // a deferred query that had nothing to enqueue, or whose zero-result early return finished it already (empty shard, no allowed page): the
  // record is complete and no event of it may be read -- every caller (mv_query_stats_finish, mv_comm's per-shard loop) comes through here
  if (st->reserved & kStatsDoneTag) { st->reserved = 0; return MV_OK; }
  st->reserved &= ~(kStatsDeferredTag | kStatsParityBit);  // a deferred record's routing bits are not stage flags
  MV_HIP(hipEventSynchronize(ix->ev[had_topk ? 2 : 1]));
  if (st->reserved) {  // stage split of the FDE modes (events recorded by run_scan)
    MV_HIP(hipEventElapsedTime(&st->encode_ms, ix->ev[0], ix->ev_st[0]));
    MV_HIP(hipEventElapsedTime(&st->coarse_ms, ix->ev_st[0], ix->ev_st[1]));
    if (st->reserved & (1 << 29)) { st->reserved = 0; }
  }
  if (st->reserved) {  // accounting of the rerank stage: rows of the candidates actually read
    MV_HIP(hipEventElapsedTime(&st->select_ms, ix->ev_st[1], ix->ev_st[2]));
    MV_HIP(hipEventElapsedTime(&st->rerank_ms, ix->ev_st[2], ix->ev[1]));
    const int nc = st->reserved & 0xffff;
    const bool f8 = (st->reserved >> 30) & 1;
    const bool mid = (st->reserved >> 28) & 1;  // the list before its e4m3 pruning stage was copied to h_cand[1024..) in stream order
    const bool rr_lo = (st->reserved >> 27) & 1;  // the rerank read the candidates' lo rows too (MV_WITH_FLOAT_LO)
    st->reserved = 0;
    MV_HIP(hipMemcpyAsync(ix->h_cand, ix->d_cand, (size_t)nc * 4, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));
    int64_t cand_rows = 0, mid_rows = 0;
    for (int i = 0; i < nc; ++i) {
      if (ix->h_cand[i] >= 0) cand_rows += ix->h_n_rows[ix->h_cand[i]];
      if (mid && ix->h_cand[kTopkMaxDeviceK + i] >= 0) mid_rows += ix->h_n_rows[ix->h_cand[kTopkMaxDeviceK + i]];
    }
    st->bytes_scanned += cand_rows * (int64_t)(f8 ? kDim : (rr_lo ? 2 * kRowBytes : kRowBytes)) + mid_rows * (int64_t)kDim;
  }
  MV_HIP(hipEventElapsedTime(&st->score_kernel_ms, ix->ev[0], ix->ev[1]));
  if (had_topk) {
    MV_HIP(hipEventElapsedTime(&st->topk_ms, ix->ev[1], ix->ev[2]));
    MV_HIP(hipEventElapsedTime(&st->total_device_ms, ix->ev[0], ix->ev[2]));
  } else {
    st->total_device_ms = st->score_kernel_ms;
  }
  return MV_OK;
}

int64_t coarse_n_for(const mv_index* ix, int k) {
  // reference: top_k = min(10 * k, 75)  (fast_multivector_store.py:529)
  if (ix->fde_coarse_n > 0) return ix->fde_coarse_n;
  return std::min<int64_t>(10LL * k, 75);
}

// Host selection for k beyond the device kernel's limit.
void host_topk(const std::vector<float>& s, const std::vector<int32_t>* ids_map, int64_t id_base, int64_t k,
               std::vector<float>* os, std::vector<int64_t>* oi) {
  std::vector<std::pair<float, int64_t>> v;
  v.reserve(s.size());
  for (size_t i = 0; i < s.size(); ++i)
    if (s[i] == s[i] && s[i] != -INFINITY) v.emplace_back(s[i] + 0.0f, ids_map ? (int64_t)(*ids_map)[i] : (int64_t)i);
  auto cmp = [](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
    return a.first > b.first || (a.first == b.first && a.second < b.second);
  };
  const size_t kk = std::min<size_t>((size_t)k, v.size());
  std::partial_sort(v.begin(), v.begin() + kk, v.end(), cmp);
  os->resize(kk);
  oi->resize(kk);
  for (size_t i = 0; i < kk; ++i) { (*os)[i] = v[i].first; (*oi)[i] = id_base + v[i].second; }
}

// Writer-side scratch (w_mu held): grown on demand and kept -- hipFree synchronises the whole device, which would stall
// the scans an ingest is supposed to run beside, so it only happens when a buffer has to grow.
int w_reserve(void** p, size_t* cap, size_t need) {
  if (need <= *cap) return MV_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = std::max<size_t>(need, 4096);
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { *p = nullptr; set_error("out of device memory for %zu bytes of ingest staging", want); return MV_ERR_NOMEM; }
  *cap = want;
  return MV_OK;
}

// Fill slab slots [first, first + n_pages) from device rows d_src ([total_rows][128] of dtype) on the writer stream.
// The slots are beyond the published size: no query can see them until the caller publishes.  Caller holds w_mu.
int add_pages_common(mv_index* ix, const void* d_src, int dtype, const int32_t* n_rows, int64_t n_pages,
                     const int32_t* doc_ordinals, int64_t total_rows, int64_t first) {
  hipStream_t ws = ix->w_stream;
  const int32_t stride = ix->cfg.stride_rows;
  const bool packed = ix->packed;
  std::vector<int64_t> off((size_t)n_pages + 1);
  off[0] = 0;
  for (int64_t i = 0; i < n_pages; ++i) off[i + 1] = off[i] + n_rows[i];
  // host + device metadata of the new slots first: the derive kernels read the device row counts
  for (int64_t i = 0; i < n_pages; ++i) {
    ix->h_n_rows[first + i] = n_rows[i];
    ix->h_doc_ord[first + i] = doc_ordinals ? doc_ordinals[i] : 0;
    if (packed) ix->h_row_off[(size_t)(first + i + 1)] = ix->h_row_off[(size_t)(first + i)] + ((int64_t)n_rows[i] + 15) / 16 * 16;  // whole 16-row tiles
  }
  MV_HIP(hipMemcpyAsync(ix->d_n_rows + first, ix->h_n_rows.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice, ws));
  MV_HIP(hipMemcpyAsync(ix->d_doc_ord + first, ix->h_doc_ord.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice, ws));
  if (packed) MV_HIP(hipMemcpyAsync(ix->d_row_off + first, ix->h_row_off.data() + first, (size_t)(n_pages + 1) * 8, hipMemcpyHostToDevice, ws));
  // the batch's rows in the row-indexed slabs: [row0, row0 + slot_rows) -- one stride slot per page, or (packed) whole tiles back to back
  const int64_t row0 = page_row0(ix, first);
  const int64_t slot_rows = rows_in_use(ix, first + n_pages) - row0;
  const int64_t* d_ro = packed ? ix->d_row_off + first : nullptr;
  const bool f32_bits = (ix->cfg.flags & MV_WITH_BINARY) && dtype == MV_F32;
  const size_t off_bytes = ((off.size() * 8 + 255) / 256) * 256;
  int rc = w_reserve(&ix->w_aux, &ix->w_aux_bytes, off_bytes + (f32_bits ? (size_t)std::max<int64_t>(total_rows, 1) * kSignBytes : 0));
  if (rc) return rc;
  int64_t* d_off = (int64_t*)ix->w_aux;
  uint8_t* tmp_bits = (uint8_t*)ix->w_aux + off_bytes;
  MV_HIP(hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, ws));
  // `img`: where the kernels below find the batch's fixed-stride / packed bf16 image.  Fixed layout: the address of page `first`'s
  // slot (pages are indexed locally).  Packed layout: the base the row offsets count from -- the slab itself, or, for an index that
  // keeps no float slab, the staging buffer shifted back by row0 rows (the offsets are absolute; only rows >= row0 are touched).
  uint16_t* rows_ptr = nullptr;  // the batch's first row
  if (ix->cfg.flags & MV_WITH_FLOAT) {
    rows_ptr = ix->slab + (size_t)row0 * kDim;
  } else {
    // no float slab kept: still need the bf16 rows as the source of the other slabs
    rc = w_reserve(&ix->w_tmp, &ix->w_tmp_bytes, (size_t)std::max<int64_t>(slot_rows, 1) * kRowBytes);
    if (rc) return rc;
    rows_ptr = (uint16_t*)ix->w_tmp;
  }
  uint16_t* img = packed ? reinterpret_cast<uint16_t*>(reinterpret_cast<uintptr_t>(rows_ptr) - (uintptr_t)row0 * kRowBytes) : rows_ptr;
  uint16_t* lo_img = ix->slab_lo ? (packed ? ix->slab_lo : ix->slab_lo + (size_t)row0 * kDim) : nullptr;  // lo = bf16(x - bf16(x)) of fp32 rows, zeros for bf16 rows
  // NaN / Inf rows: an index with any float-derived slab refuses them (their MaxSim is undefined: torch's einsum -> max -> topk
  // would rank a NaN page first); a sign-bit-only index takes them -- its quantiser defines every input (binary_ops.rs:81-136)
  const bool check_finite = (ix->cfg.flags & ~(MV_WITH_BINARY | MV_LAYOUT_PACKED)) != 0;
  if (check_finite) MV_HIP(hipMemsetAsync(ix->d_w_flag, 0, 4, ws));
  rc = launch_scatter_rows(d_src, dtype, d_off, n_pages, stride, img, ws, check_finite ? ix->d_w_flag : nullptr, lo_img, d_ro);
  if (!rc) rc = xt_store_from_device(ix, rows_ptr, first, n_pages, ws);  // exact host tier (pinned host memory, or split with HBM; fixed layout only): the bf16 image, slot for slot
  if (!rc && (ix->cfg.flags & MV_WITH_BINARY)) {
    // sign bits come from the bf16 image: bf16 RNE preserves sign and zero-ness of every fp32 value
    // that is not an fp32 subnormal rounding to zero; fp32 inputs are packed from the fp32 rows below.
    uint8_t* bdst = ix->bits + (size_t)row0 * kSignBytes;
    if (dtype == MV_BF16) {
      rc = launch_sign_pack_bf16_rows(rows_ptr, slot_rows, bdst, ws);
    } else {
      // exact fp32 rule (v > 0.0f): pack the ragged fp32 rows, then scatter 16-byte rows
      rc = launch_sign_pack_f32((const float*)d_src, total_rows, kDim, tmp_bits, ws);
      if (!rc) {
        (void)hipMemsetAsync(bdst, 0, (size_t)slot_rows * kSignBytes, ws);
        for (int64_t i = 0; i < n_pages && !rc; ++i)
          if (n_rows[i] > 0 && hipMemcpyAsync(bdst + (size_t)(page_row0(ix, first + i) - row0) * kSignBytes, tmp_bits + (size_t)off[i] * kSignBytes,
                                              (size_t)n_rows[i] * kSignBytes, hipMemcpyDeviceToDevice, ws) != hipSuccess) {
            rc = MV_ERR_HIP; set_error("D2D of sign rows failed");
          }
      }
    }
  }
  if (!rc && (ix->cfg.flags & MV_WITH_FDE)) {
    FdeEncodeArgs e{};
    e.variant = ix->fde_encode_variant;
    if (dtype == MV_F32) {
      e.x_f32 = (const float*)d_src; e.row_offsets = d_off;
    } else {
      e.x_bf16 = img; e.x_row_off = d_ro; e.n_rows = ix->d_n_rows + first; e.stride = stride;
    }
    e.n_pages = n_pages; e.is_query = 0;
    e.out_bf16 = ix->fde + (size_t)first * ix->fde_t.out_dim;
    e.out_inv_norm = ix->fde_inv_norm + first;
    rc = launch_fde_encode(ix->fde_t, e, ws);
    if (!rc) rc = fde8_requantize(ix, first, n_pages, ws);
  }
  if (!rc && (ix->cfg.flags & MV_WITH_FP8))
    rc = launch_quantize_pages_fp8(img, ix->d_n_rows + first, stride, n_pages, packed ? ix->slab8 : ix->slab8 + (size_t)row0 * kDim,
                                   ix->inv_scale8 + first, ws, d_ro);
  hipError_t e = hipStreamSynchronize(ws);  // the staging buffers are reused by the next chunk; the caller publishes after this
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "ingest", __FILE__, __LINE__);
  if (check_finite) {
    int32_t bad = 0;
    MV_HIP(hipMemcpy(&bad, ix->d_w_flag, 4, hipMemcpyDeviceToHost));
    if (bad) {  // nothing was published: the slots stay invisible and are overwritten by the next add
      set_error("mv_index_add: an embedding row holds a NaN / Inf (or an fp32 value beyond the bf16 range): MaxSim is undefined for it; nothing was added");
      return MV_ERR_INVALID;
    }
  }
  return MV_OK;
}

// Make pages [first, first + n) visible to queries: flags first, then the size (release).  Caller holds w_mu.
void publish_pages(mv_index* ix, int64_t first, int64_t n) {
  int32_t mo = ix->max_doc_ord.load();
  bool rag = false, tomb = false;
  for (int64_t i = first; i < first + n; ++i) {
    mo = std::max(mo, ix->h_doc_ord[i]);
    rag = rag || ix->h_n_rows[i] != ix->cfg.stride_rows;
    tomb = tomb || ix->h_doc_ord[i] < 0;
  }
  ix->max_doc_ord.store(mo);
  if (rag) ix->ragged.store(true);
  if (tomb) ix->tombstones.store(true);
  ix->size.store(first + n, std::memory_order_release);
}

// Caller holds w_mu (the capacity check must see the size no other writer can move).
int validate_add(mv_index* ix, const void* emb, int dtype, const int32_t* n_rows, int64_t n_pages, int64_t* total_rows) {
  if (!ix || (!emb && n_pages > 0) || (!n_rows && n_pages > 0) || n_pages < 0) { set_error("mv_index_add: null argument"); return MV_ERR_INVALID; }
  if (dtype != MV_F32 && dtype != MV_BF16) { set_error("mv_index_add: bad dtype %d", dtype); return MV_ERR_INVALID; }
  const int64_t size = ix->size.load();
  if (size + n_pages > ix->cfg.capacity_pages) {
    set_error("slab full: %lld + %lld > capacity %lld", (long long)size, (long long)n_pages, (long long)ix->cfg.capacity_pages);
    return MV_ERR_CAPACITY;
  }
  int64_t t = 0, slots = 0;
  for (int64_t i = 0; i < n_pages; ++i) {
    if (n_rows[i] < 0 || n_rows[i] > ix->cfg.stride_rows) {
      set_error("page %lld has %d rows; stride_rows is %d", (long long)i, n_rows[i], ix->cfg.stride_rows);
      return MV_ERR_INVALID;
    }
    t += n_rows[i];
    slots += ((int64_t)n_rows[i] + 15) / 16 * 16;
  }
  if (ix->packed && rows_in_use(ix, size) + slots > ix->cap_rows) {
    set_error("slab full: %lld rows in use + %lld > capacity_rows %lld", (long long)rows_in_use(ix, size), (long long)slots, (long long)ix->cap_rows);
    return MV_ERR_CAPACITY;
  }
  *total_rows = t;
  return MV_OK;
}

// Operations that move or rewrite PUBLISHED pages exclude writers and queries (lock order: w_mu, then q_mu).
struct ExclusiveLock {
  std::lock_guard<std::mutex> w, q;
  explicit ExclusiveLock(mv_index* ix) : w(ix->w_mu), q(ix->q_mu) {}
};

}  // namespace mv

// ---------------------------------------------------------------------------------- host memory the process may still pin
// The exact tier of MV_WITH_HOST_EXACT is pinned host memory (262 144 B per 1024-row page: 328 GB for a 1.25 M-page shard).
// Pinned pages are charged to the process's memory cgroup; a container whose memory.max is below the request is KILLED by the
// kernel half way through hipHostMalloc (measured on the MI355X pool: memory.max = 300 GiB on a 3 TiB host -- the box is lost,
// no error is returned).  So the library reads the limits itself and refuses up front.
namespace mv {

thread_local QueryFdeOverride g_qfde;

int check_fde_finite(const float* v, size_t n, const char* what) {
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(v[i])) { set_error("%s: element %zu of the FDE vectors is NaN / Inf", what, i); return MV_ERR_INVALID; }
  return MV_OK;
}

static int64_t read_i64_file(const char* path) {  // -1: missing / "max" / unparsable
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char buf[64] = {0};
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  if (n == 0 || buf[0] < '0' || buf[0] > '9') return -1;
  return (int64_t)strtoll(buf, nullptr, 10);
}

// Bytes of host memory this process can still pin without running into its cgroup limit or the machine's free memory, minus a
// headroom of max(4 GiB, 5 % of the limit); INT64_MAX when nothing limits it.  MV_HOST_EXACT_MAX_BYTES in the environment caps it.
int64_t host_pin_budget_bytes() {
  int64_t budget = INT64_MAX;
  // cgroup v2: the namespace root and every level down to the process's own group; v1: the memory controller's files
  std::string rel, rel_v1;
  if (FILE* f = fopen("/proc/self/cgroup", "r")) {
    char line[1024];
    while (fgets(line, sizeof(line), f)) {
      std::string ln = line;
      while (!ln.empty() && (ln.back() == '\n' || ln.back() == '/')) ln.pop_back();
      if (ln.compare(0, 3, "0::") == 0) rel = ln.substr(3);
      const size_t m = ln.find(":memory:");
      if (m != std::string::npos) rel_v1 = ln.substr(m + 8);
    }
    fclose(f);
  }
  std::vector<std::string> dirs{"/sys/fs/cgroup"};
  for (size_t pos = 0; pos < rel.size();) {
    const size_t nxt = rel.find('/', pos + 1);
    dirs.push_back("/sys/fs/cgroup" + rel.substr(0, nxt == std::string::npos ? rel.size() : nxt));
    if (nxt == std::string::npos) break;
    pos = nxt;
  }
  for (const std::string& d : dirs) {
    const int64_t lim = read_i64_file((d + "/memory.max").c_str());
    if (lim < 0) continue;
    const int64_t cur = std::max<int64_t>(0, read_i64_file((d + "/memory.current").c_str()));
    budget = std::min(budget, lim - cur - std::max<int64_t>((int64_t)4 << 30, lim / 20));
  }
  for (const std::string& d : {std::string("/sys/fs/cgroup/memory"), "/sys/fs/cgroup/memory" + rel_v1}) {
    const int64_t lim = read_i64_file((d + "/memory.limit_in_bytes").c_str());
    if (lim > 0 && lim < ((int64_t)1 << 60)) {  // v1 reports "no limit" as a huge number
      const int64_t cur = std::max<int64_t>(0, read_i64_file((d + "/memory.usage_in_bytes").c_str()));
      budget = std::min(budget, lim - cur - std::max<int64_t>((int64_t)4 << 30, lim / 20));
    }
  }
  if (FILE* f = fopen("/proc/meminfo", "r")) {
    char line[256];
    while (fgets(line, sizeof(line), f)) {
      long long kb = 0;
      if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) budget = std::min(budget, (int64_t)kb * 1024 - ((int64_t)4 << 30));
    }
    fclose(f);
  }
  if (const char* e = getenv("MV_HOST_EXACT_MAX_BYTES")) {
    const int64_t cap = (int64_t)strtoll(e, nullptr, 10);
    if (cap >= 0) budget = std::min(budget, cap);
  }
  return std::max<int64_t>(budget, 0);
}

}  // namespace mv

// =================================================================================== C ABI
extern "C" {

const char* mv_last_error(void) { return g_err.c_str(); }
const char* mv_version(void) { return "mvmaxsim 0.3 (gfx950)"; }
int mv_abi_version(void) { return MV_ABI_VERSION; }

int64_t mv_host_pin_budget_bytes(void) { return host_pin_budget_bytes(); }

int mv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void mv_index_destroy(mv_index* ix) {
  if (!ix) return;
  DeviceGuard g(ix->cfg.device);
  if (ix->stream) (void)hipStreamSynchronize(ix->stream);
  if (ix->w_stream) (void)hipStreamSynchronize(ix->w_stream);
  void* ptrs[] = {ix->d_xloc, ix->d_xoff, ix->d_xhits, ix->d_row_off, ix->slab_lo, ix->d_qlo, ix->d_bqlo, ix->slab_x, ix->d_xcand, ix->d_xscores, ix->w_stage, ix->w_aux, ix->w_tmp, ix->d_w_flag, ix->d_cand_pads, ix->d_recs, ix->d_sel_pos, ix->d_gscores, ix->slab, ix->bits, ix->fde, ix->fde_inv_norm, ix->fde8, ix->fde8_scale, ix->fde8_cfac, ix->fde4, ix->fde4_scale, ix->fde4_cfac, ix->d_bqfac, ix->slab8, ix->inv_scale8, ix->d_q8hi, ix->d_q8lo, ix->d_q8fac, ix->d_bq, ix->d_bscores, ix->d_fcand, ix->d_fcounts, ix->d_n_rows, ix->d_doc_ord, ix->d_scores, ix->d_scores2,
                  ix->d_topk_ws, ix->d_q, ix->d_qf32, ix->d_qbits, ix->d_qpop, ix->d_qfde, ix->d_qoff, ix->d_allow, ix->d_out_s,
                  ix->d_out_id, ix->d_cand, ix->d_cand_scores, ix->d_bqf32, ix->d_bqfde, ix->d_bqimage, ix->d_btopk_ws, ix->d_bsel_s,
                  ix->d_bsel_id, ix->d_bcand, ix->d_bcand_pads, ix->d_bcand_scores, ix->d_bout_s, ix->d_bout_id, ix->d_bq8hi, ix->d_bq8lo, ix->d_bq8fac};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (void* p : ix->parked) (void)hipFree(p);
  fde_tables_destroy(&ix->fde_t);
  for (auto& e : ix->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ix->ev_st)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ix->ev_alt)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ix->ev_st_alt)
    if (e) (void)hipEventDestroy(e);
  if (ix->ev_stage) (void)hipEventDestroy(ix->ev_stage);
  for (void* hp : {(void*)ix->h_qlo, (void*)ix->h_qf32, (void*)ix->h_qbf16, (void*)ix->h_out_s, (void*)ix->h_out_id, (void*)ix->h_cand, (void*)ix->h_bout_s,
                   (void*)ix->h_bout_id, (void*)ix->h_bcand, (void*)ix->h_exact})  // (slab_x, the HBM part of a split exact tier, went with ptrs[])
    if (hp) (void)hipHostFree(hp);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  if (ix->w_stream) (void)hipStreamDestroy(ix->w_stream);
  delete ix;
}

int mv_index_create(const mv_config* cfg, mv_index** out) {
  if (!cfg || !out) { set_error("mv_index_create: null argument"); return MV_ERR_INVALID; }
  *out = nullptr;
  if (cfg->dim != kDim) { set_error("dim must be 128 (got %d)", cfg->dim); return MV_ERR_INVALID; }
  if (cfg->stride_rows < 16 || cfg->stride_rows % 16) { set_error("stride_rows must be a positive multiple of 16 (got %d)", cfg->stride_rows); return MV_ERR_INVALID; }
  if (cfg->capacity_pages < 1 || cfg->capacity_pages > 0x7fffffffLL) { set_error("capacity_pages out of range"); return MV_ERR_INVALID; }
  if (!(cfg->flags & (MV_WITH_FLOAT | MV_WITH_BINARY | MV_WITH_FDE | MV_WITH_FP8))) { set_error("flags select no slab"); return MV_ERR_INVALID; }
  if (cfg->flags & MV_WITH_FDE_E4M3) {
    if (!(cfg->flags & MV_WITH_FDE)) { set_error("MV_WITH_FDE_E4M3 needs MV_WITH_FDE (it is a copy of that slab)"); return MV_ERR_INVALID; }
    if (!fde_scan8_supported(mv_fde_output_dim(&cfg->fde))) { set_error("MV_WITH_FDE_E4M3 needs an FDE width of 10240, 4096 or 2048 (got %lld)", (long long)mv_fde_output_dim(&cfg->fde)); return MV_ERR_INVALID; }
  }
  if (cfg->flags & MV_WITH_FDE_FP4) {
    if (!(cfg->flags & MV_WITH_FDE)) { set_error("MV_WITH_FDE_FP4 needs MV_WITH_FDE (it is a copy of that slab)"); return MV_ERR_INVALID; }
    if (!fde_scan4_supported(mv_fde_output_dim(&cfg->fde))) { set_error("MV_WITH_FDE_FP4 needs an FDE width of 10240, 4096 or 2048 (got %lld)", (long long)mv_fde_output_dim(&cfg->fde)); return MV_ERR_INVALID; }
  }
  if (cfg->flags & MV_LAYOUT_PACKED) {
    if (cfg->flags & MV_WITH_HOST_EXACT) { set_error("MV_LAYOUT_PACKED cannot be combined with MV_WITH_HOST_EXACT (the host exact tier keeps fixed-stride pages)"); return MV_ERR_INVALID; }
    if (cfg->capacity_rows < 0 || cfg->capacity_rows % 16 || (cfg->capacity_rows > 0 && cfg->capacity_rows < cfg->stride_rows)) {
      set_error("capacity_rows must be 0 or a multiple of 16 >= stride_rows (got %lld)", (long long)cfg->capacity_rows); return MV_ERR_INVALID;
    }
  }
  if ((cfg->flags & MV_WITH_FLOAT_LO) && !(cfg->flags & MV_WITH_FLOAT)) { set_error("MV_WITH_FLOAT_LO is the lo half of the bf16 slab: it needs MV_WITH_FLOAT"); return MV_ERR_INVALID; }
  if ((cfg->flags & MV_WITH_EXACT_SPLIT) && (!(cfg->flags & MV_WITH_HOST_EXACT) || (cfg->flags & MV_WITH_FLOAT))) {
    set_error("MV_WITH_EXACT_SPLIT splits the host exact tier: it needs MV_WITH_HOST_EXACT and no MV_WITH_FLOAT"); return MV_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available (libmvmaxsim requires an MI355X / gfx950 GPU)"); return MV_ERR_HIP; }
  if (cfg->device < 0 || cfg->device >= ndev) { set_error("device %d out of range (have %d)", cfg->device, ndev); return MV_ERR_INVALID; }
  DeviceGuard g(cfg->device);
  mv_index* ix = new (std::nothrow) mv_index();
  if (!ix) { set_error("host allocation failed"); return MV_ERR_NOMEM; }
  ix->cfg = *cfg;
  const int64_t cap = cfg->capacity_pages;
  ix->bscore_stride = cap;
  if (const char* e = getenv("MV_BSCORE_STRIDE_PAD")) ix->bscore_stride = cap + std::max<int64_t>(0, (int64_t)strtoll(e, nullptr, 10));  // placement experiments
  ix->packed = (cfg->flags & MV_LAYOUT_PACKED) != 0;
  ix->cap_rows = (ix->packed && cfg->capacity_rows > 0) ? cfg->capacity_rows : cap * (int64_t)cfg->stride_rows;
  ix->cfg.capacity_rows = ix->cap_rows;
  const size_t rows = (size_t)ix->cap_rows;  // rows of every row-indexed slab
  int rc = MV_OK;
  auto alloc = [&](void** p, size_t bytes, const char* what) {
    if (rc) return;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      set_error("hipMalloc of %zu bytes for %s failed: %s", bytes, what, hipGetErrorString(e));
      rc = MV_ERR_NOMEM;
      *p = nullptr;
    }
  };
  if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ix->w_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); rc = MV_ERR_HIP; }
  for (auto& e : ix->ev)
    if (!rc && hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); rc = MV_ERR_HIP; }
  for (auto& e : ix->ev_st)
    if (!rc && hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); rc = MV_ERR_HIP; }
  for (auto& e : ix->ev_alt)
    if (!rc && hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); rc = MV_ERR_HIP; }
  for (auto& e : ix->ev_st_alt)
    if (!rc && hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); rc = MV_ERR_HIP; }
  if (!rc && hipEventCreateWithFlags(&ix->ev_stage, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); rc = MV_ERR_HIP; }
  if (!rc && (hipHostMalloc((void**)&ix->h_out_s, (size_t)kTopkMaxDeviceK * 4, hipHostMallocDefault) != hipSuccess ||
              hipHostMalloc((void**)&ix->h_out_id, (size_t)kTopkMaxDeviceK * 8, hipHostMallocDefault) != hipSuccess ||
              hipHostMalloc((void**)&ix->h_cand, (size_t)2 * kTopkMaxDeviceK * 4, hipHostMallocDefault) != hipSuccess)) { set_error("hipHostMalloc failed"); rc = MV_ERR_NOMEM; }
  if (!rc) {  // the device's view of the pinned result buffers (MV_DIRECT_HOST_RESULTS=0 keeps the D2H copies)
    const char* e = getenv("MV_DIRECT_HOST_RESULTS");
    void *ds = nullptr, *di = nullptr;
    if (!(e && e[0] == '0') && hipHostGetDevicePointer(&ds, ix->h_out_s, 0) == hipSuccess && hipHostGetDevicePointer(&di, ix->h_out_id, 0) == hipSuccess) {
      ix->hd_out_s = (float*)ds;
      ix->hd_out_id = (int64_t*)di;
    }
  }
  if (cfg->flags & MV_WITH_FLOAT) alloc((void**)&ix->slab, rows * kRowBytes + 32768, "bf16 page slab");  // +32 KiB: the batched scans DMA whole 16 / 32 KiB chunks
  if (cfg->flags & MV_WITH_FLOAT_LO) alloc((void**)&ix->slab_lo, rows * kRowBytes + 32768, "lo half of the bf16 page slab");
  if (cfg->flags & MV_WITH_FP8) {
    alloc((void**)&ix->slab8, rows * kDim + 4096, "fp8 page slab");  // +4 KiB: the scan DMAs whole 4 KiB pieces
    alloc((void**)&ix->inv_scale8, (size_t)cap * 4, "fp8 page scales");
  }
  if (cfg->flags & MV_WITH_BINARY) alloc((void**)&ix->bits, rows * kSignBytes + 4096, "sign-bit slab");  // +4 KiB: the scan DMAs whole 1 KiB pieces
  if (!rc && (cfg->flags & MV_WITH_FDE)) {
    rc = fde_tables_create(cfg->fde, &ix->fde_t);
    alloc((void**)&ix->fde, (size_t)cap * ix->fde_t.out_dim * 2, "FDE slab");
    alloc((void**)&ix->fde_inv_norm, (size_t)cap * 4, "FDE norms");
    if (cfg->flags & MV_WITH_FDE_E4M3) {
      alloc((void**)&ix->fde8, (size_t)cap * ix->fde_t.out_dim, "e4m3 copy of the FDE slab");
      alloc((void**)&ix->fde8_scale, (size_t)cap * 4, "e4m3 FDE scales");
      alloc((void**)&ix->fde8_cfac, (size_t)cap * 4, "e4m3 FDE cosine factors");
    }
    if (cfg->flags & MV_WITH_FDE_FP4) {
      alloc((void**)&ix->fde4, (size_t)cap * (ix->fde_t.out_dim / 2) + 4096, "fp4 copy of the FDE slab");  // +4 KiB: the batched pass pads the chunk count to a multiple of four (reads <= 1 KiB past a row)
      alloc((void**)&ix->fde4_scale, (size_t)cap * 4, "fp4 FDE scales");
      alloc((void**)&ix->fde4_cfac, (size_t)cap * 4, "fp4 FDE cosine factors");
      ix->fde_coarse_e4m3 = 2;  // MV_OPT_FDE_COARSE_SLAB: the copy the index was built with is the one the coarse stage reads
    }
    alloc((void**)&ix->d_qfde, (size_t)std::max<int64_t>(ix->fde_t.out_dim, 1) * 4, "query FDE");
  }
  if (ix->packed) {
    alloc((void**)&ix->d_row_off, (size_t)(cap + 1) * 8, "page row offsets");
    if (!rc && hipMemset(ix->d_row_off, 0, (size_t)(cap + 1) * 8) != hipSuccess) { set_error("hipMemset of the row offsets failed"); rc = MV_ERR_HIP; }
    ix->ragged.store(true);  // the kernels always take the per-page row counts: a slot is whole tiles, not stride_rows
  }
  alloc((void**)&ix->d_n_rows, (size_t)cap * 4, "row counts");
  alloc((void**)&ix->d_doc_ord, (size_t)cap * 4, "doc ordinals");
  alloc((void**)&ix->d_scores, (size_t)cap * 4, "scores");
  alloc((void**)&ix->d_scores2, (size_t)cap * 4, "scores2");
  ix->topk_ws_bytes = topk_ws_bytes(std::max<int64_t>(cap, 16384), kTopkMaxDeviceK);  // >= the gathered list of a two-stage rerank
  alloc(&ix->d_topk_ws, ix->topk_ws_bytes, "top-k workspace");
  if (!rc && hipMemset(ix->d_topk_ws, 0, ix->topk_ws_bytes) != hipSuccess) { set_error("hipMemset of the top-k workspace failed"); rc = MV_ERR_HIP; }  // the radix selection keeps its histograms zeroed between calls
  alloc((void**)&ix->d_out_s, (size_t)kTopkMaxDeviceK * 4, "top-k scores");
  alloc((void**)&ix->d_out_id, (size_t)kTopkMaxDeviceK * 8, "top-k ids");
  alloc((void**)&ix->d_cand, (size_t)kMaxCand * 4, "candidates");
  alloc((void**)&ix->d_cand_scores, (size_t)kMaxCand * 4, "candidate scores");
  alloc((void**)&ix->d_cand_pads, (size_t)kMaxCand * 4, "candidate pad lengths");
  alloc((void**)&ix->d_recs, (size_t)kTopkMaxDeviceK * sizeof(mv_cand_rec), "coarse candidate records");
  alloc((void**)&ix->d_sel_pos, (size_t)kTopkMaxDeviceK * 8, "coarse selection");
  alloc((void**)&ix->d_qoff, 16, "query offsets");
  alloc((void**)&ix->d_w_flag, 4, "ingest flag");
  if (!rc) {
    ix->h_n_rows.assign((size_t)cap, 0);
    ix->h_doc_ord.assign((size_t)cap, -1);
    if (ix->packed) ix->h_row_off.assign((size_t)cap + 1, 0);
    rc = ensure_query_cap(ix, 64);
  }
  if (!rc && (cfg->flags & MV_WITH_HOST_EXACT)) {
    // The exact host tier, allocated LAST.  MV_WITH_EXACT_SPLIT: as many leading pages as the HBM left over by the other slabs
    // holds (minus MV_EXACT_HBM_RESERVE_BYTES, default 12 GiB, for the lazily sized batch workspaces and the caller's own
    // device memory) keep their exact rows on the device; only the rest is pinned.
    const size_t page_b = (size_t)cfg->stride_rows * kRowBytes;
    if (cfg->flags & MV_WITH_EXACT_SPLIT) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { set_error("hipMemGetInfo failed"); rc = MV_ERR_HIP; }
      int64_t reserve = (int64_t)12 << 30;
      if (const char* e = getenv("MV_EXACT_HBM_RESERVE_BYTES")) reserve = std::max<int64_t>(0, (int64_t)strtoll(e, nullptr, 10));
      int64_t split = ((int64_t)free_b - reserve - 32768) / (int64_t)page_b;
      if (const char* e = getenv("MV_EXACT_HBM_MAX_PAGES")) split = std::min<int64_t>(split, (int64_t)strtoll(e, nullptr, 10));  // (tests: force a split on a small index)
      ix->x_split = std::max<int64_t>(0, std::min<int64_t>(cap, split));
      if (ix->x_split > 0) alloc((void**)&ix->slab_x, (size_t)ix->x_split * page_b + 32768, "HBM part of the exact tier");  // +32 KiB: whole DMA chunks
      if (ix->x_split > 0 && ix->x_split < cap) {  // a real split: count the exact reads per page (mv_index_exact_tier_rebalance ranks by them)
        alloc((void**)&ix->d_xhits, (size_t)cap * 4, "exact-tier hit counters");
        if (!rc && hipMemset(ix->d_xhits, 0, (size_t)cap * 4) != hipSuccess) { set_error("hipMemset of the hit counters failed"); rc = MV_ERR_HIP; }
      }
    }
    const size_t host_b = (size_t)(cap - ix->x_split) * page_b;
    if (!rc && host_b) {
      const int64_t budget = host_pin_budget_bytes();
      if ((int64_t)(host_b + 32768) > budget) {
        // refuse BEFORE pinning: a container past its memory cgroup limit is killed, not told
        set_error("the pinned-host exact tier needs %zu bytes (%lld pages x %d rows x 256 B%s) but this process may pin only %lld more "
                  "(memory cgroup limit / MemAvailable minus headroom; mv_host_pin_budget_bytes()): use fewer pages per shard, MV_WITH_EXACT_SPLIT, or raise the container's memory limit",
                  host_b, (long long)(cap - ix->x_split), cfg->stride_rows, ix->x_split ? "; the rest is in HBM" : "", (long long)budget);
        rc = MV_ERR_NOMEM;
      }
    }
    if (!rc && host_b) {
      // pinned + mapped: the rerank kernel reads the candidates' rows straight out of host RAM (+32 KiB: whole DMA chunks)
      hipError_t e = hipHostMalloc((void**)&ix->h_exact, host_b + 32768, hipHostMallocMapped | hipHostMallocPortable);
      if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&ix->d_exact, ix->h_exact, 0);
      if (e != hipSuccess) { set_error("pinned host allocation of %zu bytes for the exact tier failed: %s", host_b, hipGetErrorString(e)); rc = MV_ERR_NOMEM; }
    }
  }
  if (rc) {
    std::string keep = g_err;
    mv_index_destroy(ix);
    g_err = keep;
    return rc;
  }
  *out = ix;
  return MV_OK;
}

int64_t mv_index_exact_hbm_pages(const mv_index* ix) { return ix ? ix->x_split : 0; }

int mv_index_exact_tier_hits(mv_index* ix, int64_t* out_hbm_reads, int64_t* out_host_reads) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  if (out_hbm_reads) *out_hbm_reads = 0;
  if (out_host_reads) *out_host_reads = 0;
  if (!ix->d_xhits) return MV_OK;  // not a split tier
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  const int64_t n = ix->size.load();
  std::vector<uint32_t> h((size_t)std::max<int64_t>(n, 1));
  MV_HIP(hipStreamSynchronize(ix->stream));
  MV_HIP(hipMemcpy(h.data(), ix->d_xhits, (size_t)n * 4, hipMemcpyDeviceToHost));
  int64_t hb = 0, ho = 0;
  for (int64_t p = 0; p < n; ++p) (xt_slot(ix, p) < ix->x_split ? hb : ho) += h[(size_t)p];
  if (out_hbm_reads) *out_hbm_reads = hb;
  if (out_host_reads) *out_host_reads = ho;
  return MV_OK;
}

int mv_index_exact_tier_rebalance(mv_index* ix, int64_t max_moves, int64_t* out_moved) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  if (out_moved) *out_moved = 0;
  if (!ix->d_xhits) return MV_OK;  // no split: nothing to place
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  const int64_t n = ix->size.load();
  if (n <= ix->x_split) return MV_OK;  // every published page already sits in HBM
  std::vector<uint32_t> h((size_t)n);
  MV_HIP(hipStreamSynchronize(ix->stream));
  MV_HIP(hipMemcpy(h.data(), ix->d_xhits, (size_t)n * 4, hipMemcpyDeviceToHost));
  // hot pages now in pinned host memory (most read first) against cold pages now in HBM (least read first): swap while it pays
  std::vector<int32_t> hot, cold;
  for (int64_t p = 0; p < n; ++p) {
    if (ix->h_doc_ord[(size_t)p] < 0) { if (xt_slot(ix, p) < ix->x_split) cold.push_back((int32_t)p); continue; }  // a tombstoned page in HBM is the best victim
    if (xt_slot(ix, p) < ix->x_split) cold.push_back((int32_t)p);
    else if (h[(size_t)p] > 0) hot.push_back((int32_t)p);
  }
  auto heat = [&](int32_t p) { return ix->h_doc_ord[(size_t)p] < 0 ? 0u : h[(size_t)p]; };
  std::sort(hot.begin(), hot.end(), [&](int32_t a, int32_t b) { return heat(a) > heat(b) || (heat(a) == heat(b) && a < b); });
  std::sort(cold.begin(), cold.end(), [&](int32_t a, int32_t b) { return heat(a) < heat(b) || (heat(a) == heat(b) && a > b); });
  size_t k = std::min(hot.size(), cold.size());
  if (max_moves > 0) k = std::min<size_t>(k, (size_t)max_moves);
  size_t m = 0;
  while (m < k && heat(hot[m]) > heat(cold[m])) ++m;
  if (m > 0) {
    int rc = xt_ensure_tables(ix);
    if (rc) return rc;
    std::vector<std::pair<int64_t, int64_t>> pairs;
    pairs.reserve(m);
    for (size_t i = 0; i < m; ++i) pairs.emplace_back(ix->x_loc[(size_t)cold[i]], ix->x_loc[(size_t)hot[i]]);  // (HBM slot, host slot)
    rc = xt_swap_slots(ix, pairs);
    if (rc) return rc;
  }
  MV_HIP(hipMemset(ix->d_xhits, 0, (size_t)ix->cfg.capacity_pages * 4));  // the next window starts empty
  if (out_moved) *out_moved = (int64_t)m;
  return MV_OK;
}

// MV_WITH_FDE_E4M3 + MV_OPT_FDE_COARSE_SLAB 1: the batched pass reads the slab's e4m3 copy (the launcher falls back to the bf16 slab for the
// cross-check forms and widths the e4m3 kernel is not built for).  q_mu held.
void mv_internal_fde_batch_e4m3_args(mv_index* ix, mv::FdeScanBatchArgs* sa) {
  if (ix->fde4 && ix->fde_coarse_e4m3 == 2 && ix->d_bqfac) {  // MV_WITH_FDE_FP4 + MV_OPT_FDE_COARSE_SLAB 2: the FP4 form of the pass (both MFMA operands FP4)
    sa->fde8 = ix->fde4; sa->fde8_fac = ix->fde_cosine ? ix->fde4_cfac : ix->fde4_scale; sa->qfac = ix->d_bqfac; sa->copy_fp4 = 1;
    return;
  }
  if (!ix->fde8 || ix->fde_coarse_e4m3 != 1 || !ix->d_bqfac) return;
  sa->fde8 = ix->fde8; sa->fde8_fac = ix->fde_cosine ? ix->fde8_cfac : ix->fde8_scale; sa->qfac = ix->d_bqfac;
}

// One batched coarse pass (32 requests, a fixed pseudo-random query encoding) over `slab`, median of `reps` timed launches behind `warm` untimed ones.
static int time_fde_batch_pass(mv_index* ix, const void* slab, bool e4m3, int64_t n, int warm, int reps, double* out_ms) {
  FdeScanBatchArgs sa{};
  sa.fde = e4m3 ? ix->fde : (const uint16_t*)slab; sa.inv_norm = ix->fde_cosine ? ix->fde_inv_norm : nullptr; sa.doc_ord = nullptr;
  if (e4m3) { mv_internal_fde_batch_e4m3_args(ix, &sa); sa.fde8 = (const uint8_t*)slab; }
  sa.q = ix->d_bqfde; sa.image = ix->d_bqimage; sa.scores = ix->d_bscores; sa.score_stride = ix->bscore_stride; sa.n = n; sa.out_dim = ix->fde_t.out_dim;
  sa.n_queries = kFdeBatchMaxQueries;
  std::vector<float> ms;
  for (int r = 0; r < reps + warm; ++r) {
    MV_HIP(hipEventRecord(ix->ev_st[0], ix->stream));
    int rc = launch_fde_scan_batch(sa, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
    MV_HIP(hipEventSynchronize(ix->ev_st[1]));
    float t = 0.f;
    MV_HIP(hipEventElapsedTime(&t, ix->ev_st[0], ix->ev_st[1]));
    if (r >= warm) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  *out_ms = ms[ms.size() / 2];
  return MV_OK;
}

int mv_index_fde_placement_trial(mv_index* ix, int32_t trials, double* out_before_ms, double* out_after_ms, int32_t* out_moves) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  if (out_before_ms) *out_before_ms = 0.0;
  if (out_after_ms) *out_after_ms = 0.0;
  if (out_moves) *out_moves = 0;
  if (!(ix->cfg.flags & MV_WITH_FDE)) { set_error("mv_index_fde_placement_trial: the index has no FDE slab"); return MV_ERR_INVALID; }
  if (trials < 0 || trials > 16) { set_error("mv_index_fde_placement_trial: trials must be 0..16"); return MV_ERR_INVALID; }
  if (!fde_scan_batch_supported(ix->fde_t.out_dim)) return MV_OK;  // no batched pass over this encoding width: nothing to place
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  int rc = mv_internal_ensure_fde_batch_ws(ix);
  if (rc) return rc;
  const int64_t cap = ix->cfg.capacity_pages, out_dim = ix->fde_t.out_dim;
  const int64_t n = ix->size.load() > 0 ? ix->size.load() : cap;  // an empty index is timed over its whole slab (the pass's time does not depend on what it reads)
  // the slab the batched pass reads: the e4m3 copy of an index that has one (MV_OPT_FDE_COARSE_SLAB 1), else the bf16 slab
  const bool f4 = ix->fde4 && ix->fde_coarse_e4m3 == 2 && ix->d_bqfac && fde_scan_batch4_supported(out_dim);
  const bool e4 = f4 || (ix->fde8 && ix->fde_coarse_e4m3 == 1 && ix->d_bqfac && fde_scan_batch8_supported(out_dim));
  void** slab_pp = f4 ? (void**)&ix->fde4 : (e4 ? (void**)&ix->fde8 : (void**)&ix->fde);
  const size_t bytes = f4 ? (size_t)cap * (out_dim / 2) + 4096 : (size_t)cap * out_dim * (e4 ? 1 : 2);
  {  // the requests of the timing passes: fixed pseudo-random encodings (the next real batch overwrites them)
    std::vector<float> q((size_t)kFdeBatchMaxQueries * out_dim);
    uint64_t z = 0x9E3779B97F4A7C15ull;
    for (auto& v : q) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (float)((int64_t)(z >> 40) - (1 << 23)) * (1.0f / (1 << 23)); }
    MV_HIP(hipStreamSynchronize(ix->stream));
    MV_HIP(hipMemcpy(ix->d_bqfde, q.data(), q.size() * 4, hipMemcpyHostToDevice));
  }
  double best = 0.0;
  rc = time_fde_batch_pass(ix, *slab_pp, e4, n, 4, 5, &best);
  if (rc) return rc;
  if (out_before_ms) *out_before_ms = best;
  void* held = nullptr;  // the loser of the last comparison stays allocated while the next candidate is taken, so that the candidate is OTHER memory
  int moves = 0;
  for (int t = 0; t < trials; ++t) {
    void* cand = nullptr;
    if (hipMalloc(&cand, bytes) != hipSuccess) { (void)hipGetLastError(); break; }  // no room for another candidate: keep what we have
    if (held) { (void)hipFree(held); held = nullptr; }
    hipError_t e = hipMemcpyAsync(cand, *slab_pp, bytes, hipMemcpyDeviceToDevice, ix->stream);
    // incumbent, candidate, incumbent -- all three behind the copy: the pass is power-bound, and the first launches after a quiet spell (a copy
    // is one) run 1-2 % fast; timing the candidate alone there would favour it
    double ms = 0.0, inc_a = 0.0, inc_b = 0.0;
    if (e == hipSuccess) rc = time_fde_batch_pass(ix, *slab_pp, e4, n, 4, 5, &inc_a);
    if (e == hipSuccess && !rc) rc = time_fde_batch_pass(ix, cand, e4, n, 2, 5, &ms);
    if (e == hipSuccess && !rc) rc = time_fde_batch_pass(ix, *slab_pp, e4, n, 2, 5, &inc_b);
    if (e != hipSuccess || rc) { (void)hipStreamSynchronize(ix->stream); (void)hipFree(cand); if (e != hipSuccess) { set_error("hip: %s", hipGetErrorString(e)); return MV_ERR_HIP; } return rc; }
    const double inc = 0.5 * (inc_a + inc_b);
    if (ms < inc * 0.985) {  // (1.5 %: the measurement scatters by 0.2-0.3 %, a move costs a slab copy -- only a clear win is taken)
      held = *slab_pp; *slab_pp = cand; best = ms; ++moves;
    } else {
      held = cand; best = inc;
    }
  }
  MV_HIP(hipStreamSynchronize(ix->stream));
  if (held) (void)hipFree(held);
  if (out_after_ms) *out_after_ms = best;
  if (out_moves) *out_moves = moves;
  return MV_OK;
}

int mv_index_set_option(mv_index* ix, int option, int64_t value) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  switch (option) {
    case MV_OPT_MAXSIM_VARIANT:
      if (value != -1 && value != 0 && value != 6 && value != 7 && value != 13) { set_error("float kernel variant %lld does not exist (-1 default, 0 direct loads, 6 / 7 the ring with four / one wave per page, 13 transport only; 1-5, 8-12, 14 were removed in round 5)", (long long)value); return MV_ERR_INVALID; }
      ix->maxsim_variant = (int)value; return MV_OK;
    case MV_OPT_FDE_COARSE_N:
      if (value < 0 || value > kTopkMaxDeviceK) { set_error("FDE_COARSE_N must be 0..%d", kTopkMaxDeviceK); return MV_ERR_INVALID; }
      ix->fde_coarse_n = value; return MV_OK;
    case MV_OPT_FDE_COSINE: ix->fde_cosine = value ? 1 : 0; return MV_OK;
    case MV_OPT_PAD_SEMANTICS: ix->pad_semantics = (int)value; return MV_OK;
    case MV_OPT_BINARY_VARIANT:
      if (value != -1 && value != 0 && value != 4) { set_error("sign-bit scan variant %lld does not exist (-1 / 4 FP4 MFMA, 0 popcount; 1-3, 5, 6 were removed in round 5)", (long long)value); return MV_ERR_INVALID; }
      ix->binary_variant = (int)value; return MV_OK;
    case MV_OPT_FDE_SCAN_VARIANT:
      if (value != -1 && value != 0 && value != 5 && value != 6) { set_error("FDE scan variant %lld does not exist (-1 / 6 row quarters on the ring over 256 KiB-aligned blocks, 5 the same over 16-row units, 0 register form; 1-4 were removed in round 5)", (long long)value); return MV_ERR_INVALID; }
      ix->fde_scan_variant = (int)value; return MV_OK;
    case MV_OPT_BATCH_VARIANT: ix->batch_variant = (int)value; return MV_OK;
    case MV_OPT_LONG_QUERY_VARIANT: ix->long_query_variant = (int)value; return MV_OK;
    case MV_OPT_FDE_ENCODE_VARIANT: ix->fde_encode_variant = (int)value; return MV_OK;
    case MV_OPT_FILTER_COMPACT_PCT: ix->filter_compact_pct = (int)value; return MV_OK;
    case MV_OPT_FDE_QUERY_ENCODE_VARIANT: ix->fde_query_encode_variant = (int)value; return MV_OK;
    case MV_OPT_FDE_BATCH_VARIANT:
      if (value == 4 || value > 5) { set_error("FDE batch variant %lld was removed in round 5 (0 default, 1 query by query, 2 bf16 query, 3 single tile, 5 separate finish)", (long long)value); return MV_ERR_INVALID; }
      ix->fde_batch_variant = (int)value; return MV_OK;
    case MV_OPT_EXACT_TIER:
      if (value < 0 || value > 2) { set_error("EXACT_TIER must be 0 (HBM slab), 1 (pinned-host tier) or 2 (e4m3 slab)"); return MV_ERR_INVALID; }
      ix->exact_tier = (int)value; return MV_OK;
    case 1000: {  // (diagnostic, not in the header's option list) drop the lazily allocated batch workspaces: the next batched query allocates
      // them afresh -- tools/fde_batch_*_probe.py use it to tell which allocation the batched FDE pass's time follows.  value: 0 = free them all;
      // bit 0 = set them aside instead (freed with the index: the next ones cannot get the same memory back); bits 1 / 2 / 3 = only the score
      // vectors / only the query image / only the rest.
      DeviceGuard g(ix->cfg.device);
      if (ix->stream) (void)hipStreamSynchronize(ix->stream);
      const bool park = value & 1;
      const int sel = (int)(value >> 1) & 7;
      auto drop = [&](void** w) { if (*w) { if (park) ix->parked.push_back(*w); else (void)hipFree(*w); *w = nullptr; } };
      if (!sel || (sel & 1)) drop((void**)&ix->d_bscores);
      if (!sel || (sel & 2)) drop((void**)&ix->d_bqimage);
      if (ix->h_bcand) { (void)hipHostFree(ix->h_bcand); ix->h_bcand = nullptr; }  // the "workspace complete" mark: the next query re-runs the ensure
      if (!sel || (sel & 4)) {
        void** ws[] = {(void**)&ix->d_bqfde, (void**)&ix->d_bqf32, (void**)&ix->d_btopk_ws, (void**)&ix->d_bsel_s,
                       (void**)&ix->d_bsel_id, (void**)&ix->d_bcand, (void**)&ix->d_bcand_pads, (void**)&ix->d_bcand_scores, (void**)&ix->d_bout_s, (void**)&ix->d_bout_id,
                       (void**)&ix->d_bq, (void**)&ix->d_bqlo, (void**)&ix->d_bqfac, (void**)&ix->d_bq8hi, (void**)&ix->d_bq8lo, (void**)&ix->d_bq8fac};
        for (void** w : ws) drop(w);
        void** hs[] = {(void**)&ix->h_bout_s, (void**)&ix->h_bout_id};
        for (void** h : hs) if (*h) { (void)hipHostFree(*h); *h = nullptr; }
        ix->bq_lo_valid = ix->bq_has_lo = false;
      }
      return MV_OK;
    }
    case MV_OPT_FDE_COARSE_SLAB:
      if (value != 0 && value != 1 && value != 2) { set_error("FDE_COARSE_SLAB must be 0 (bf16 slab), 1 (e4m3 copy) or 2 (fp4 copy)"); return MV_ERR_INVALID; }
      if (value == 1 && !ix->fde8) { set_error("FDE_COARSE_SLAB 1 needs an index with MV_WITH_FDE_E4M3"); return MV_ERR_STATE; }
      if (value == 2 && !ix->fde4) { set_error("FDE_COARSE_SLAB 2 needs an index with MV_WITH_FDE_FP4"); return MV_ERR_STATE; }
      ix->fde_coarse_e4m3 = (int)value; return MV_OK;
    case MV_OPT_FLOAT_LO_SCAN:
      if (value < 0 || value > 2) { set_error("FLOAT_LO_SCAN must be 0 (hi slab only), 1 (hi + lo) or 2 (hi-only scan, split-bf16 re-score of the best)"); return MV_ERR_INVALID; }
      ix->float_lo_scan = (int)value; return MV_OK;
    case MV_OPT_RERANK_N:
      if (value < 1 || value > kTopkMaxDeviceK) { set_error("RERANK_N must be 1..%d", kTopkMaxDeviceK); return MV_ERR_INVALID; }
      ix->rerank_n = value; return MV_OK;
    default: set_error("unknown option %d", option); return MV_ERR_INVALID;
  }
}

int64_t mv_index_size(const mv_index* ix) { return ix ? ix->size.load(std::memory_order_acquire) : 0; }
int64_t mv_index_capacity(const mv_index* ix) { return ix ? ix->cfg.capacity_pages : 0; }
int64_t mv_index_capacity_rows(const mv_index* ix) { return ix ? ix->cap_rows : 0; }
int64_t mv_index_rows_used(const mv_index* ix) {
  if (!ix) return 0;
  std::lock_guard<std::mutex> lk(const_cast<mv_index*>(ix)->w_mu);  // row_off[size] belongs to the writers
  return rows_in_use(ix, ix->size.load());
}

int mv_index_add_device(mv_index* ix, const void* d_emb, int dtype, const int32_t* n_rows, int64_t n_pages,
                        const int32_t* doc_ordinals, int64_t* out_first_page) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->w_mu);  // writers only: queries keep running on the published prefix
  int64_t total = 0;
  int rc = validate_add(ix, d_emb, dtype, n_rows, n_pages, &total);
  if (rc) return rc;
  const int64_t first = ix->size.load();
  if (out_first_page) *out_first_page = first;
  if (n_pages == 0) return MV_OK;
  DeviceGuard g(ix->cfg.device);
  rc = add_pages_common(ix, d_emb, dtype, n_rows, n_pages, doc_ordinals, total, first);
  if (rc) return rc;  // nothing was published: the half-written slots stay invisible and are overwritten by the next add
  publish_pages(ix, first, n_pages);
  return MV_OK;
}

int mv_index_add(mv_index* ix, const void* emb, int dtype, const int32_t* n_rows, int64_t n_pages,
                 const int32_t* doc_ordinals, int64_t* out_first_page) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->w_mu);
  int64_t total = 0;
  int rc = validate_add(ix, emb, dtype, n_rows, n_pages, &total);
  if (rc) return rc;
  const int64_t first = ix->size.load();
  if (out_first_page) *out_first_page = first;
  if (n_pages == 0) return MV_OK;
  DeviceGuard g(ix->cfg.device);
  // stage in chunks so the staging buffer stays <= ~1 GiB; the whole call is published at once (all or nothing)
  const size_t esz = dtype == MV_F32 ? 4 : 2;
  int64_t p = 0;
  int64_t row_base = 0;
  while (p < n_pages) {
    int64_t q = p, rows = 0;
    while (q < n_pages && (q == p || (size_t)(rows + n_rows[q]) * kDim * esz <= ((size_t)1 << 30))) rows += n_rows[q++];
    const size_t bytes = (size_t)rows * kDim * esz;
    rc = w_reserve(&ix->w_stage, &ix->w_stage_bytes, std::max<size_t>(bytes, 16));
    if (rc) return rc;
    if (bytes) {
      hipError_t e = hipMemcpyAsync(ix->w_stage, (const char*)emb + (size_t)row_base * kDim * esz, bytes, hipMemcpyHostToDevice, ix->w_stream);
      if (e != hipSuccess) return hip_fail(e, "H2D of embeddings", __FILE__, __LINE__);
    }
    rc = add_pages_common(ix, ix->w_stage, dtype, n_rows + p, q - p, doc_ordinals ? doc_ordinals + p : nullptr, rows, first + p);
    if (rc) return rc;
    row_base += rows;
    p = q;
  }
  publish_pages(ix, first, n_pages);
  return MV_OK;
}

static inline int64_t page_slot_rows_of(const mv_index* ix, int64_t /*page*/, int32_t n_rows) {
  return ix->packed ? ((int64_t)n_rows + 15) / 16 * 16 : ix->cfg.stride_rows;
}

// Append pages given only their packed sign rows (16 B per row, MSB first): the import path for an existing
// MultiVectorStore table (BIT(128)[] column, core/vector_store/multi_vector_store.py:248) -- floats cannot be
// recovered from it, so the index must carry the sign-bit slab only.
int mv_index_add_bits(mv_index* ix, const uint8_t* bits, const int32_t* n_rows, int64_t n_pages, const int32_t* doc_ordinals,
                      int64_t* out_first_page) {
  if (!ix || (!bits && n_pages > 0) || (!n_rows && n_pages > 0) || n_pages < 0) { set_error("mv_index_add_bits: null argument"); return MV_ERR_INVALID; }
  if ((ix->cfg.flags & ~MV_LAYOUT_PACKED) != MV_WITH_BINARY) { set_error("mv_index_add_bits needs an index with MV_WITH_BINARY only (floats are not recoverable from sign bits)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->w_mu);
  if (ix->size.load() + n_pages > ix->cfg.capacity_pages) { set_error("slab full"); return MV_ERR_CAPACITY; }
  const int32_t stride = ix->cfg.stride_rows;
  for (int64_t i = 0; i < n_pages; ++i)
    if (n_rows[i] < 0 || n_rows[i] > stride) { set_error("page %lld has %d rows; stride_rows is %d", (long long)i, n_rows[i], stride); return MV_ERR_INVALID; }
  DeviceGuard g(ix->cfg.device);
  const int64_t first = ix->size.load();
  if (ix->packed) {
    int64_t slots = 0;
    for (int64_t i = 0; i < n_pages; ++i) slots += ((int64_t)n_rows[i] + 15) / 16 * 16;
    if (rows_in_use(ix, first) + slots > ix->cap_rows) { set_error("slab full (capacity_rows)"); return MV_ERR_CAPACITY; }
    for (int64_t i = 0; i < n_pages; ++i) ix->h_row_off[(size_t)(first + i + 1)] = ix->h_row_off[(size_t)(first + i)] + ((int64_t)n_rows[i] + 15) / 16 * 16;
    MV_HIP(hipMemcpy(ix->d_row_off + first, ix->h_row_off.data() + first, (size_t)(n_pages + 1) * 8, hipMemcpyHostToDevice));
  }
  std::vector<uint8_t> img;
  int64_t src_row = 0;
  for (int64_t p0 = 0; p0 < n_pages; p0 += 4096) {  // the slot image (stride slots / whole tiles back to back) of up to 4096 pages per copy
    const int64_t c = std::min<int64_t>(4096, n_pages - p0);
    const int64_t r0 = page_row0(ix, first + p0), rows = page_row0(ix, first + p0 + c - 1) + page_slot_rows_of(ix, first + p0 + c - 1, n_rows[p0 + c - 1]) - r0;
    img.assign((size_t)rows * kSignBytes, (uint8_t)0);
    for (int64_t i = 0; i < c; ++i) {
      memcpy(img.data() + (size_t)(page_row0(ix, first + p0 + i) - r0) * kSignBytes, bits + (size_t)src_row * kSignBytes, (size_t)n_rows[p0 + i] * kSignBytes);
      src_row += n_rows[p0 + i];
    }
    MV_HIP(hipMemcpy(ix->bits + (size_t)r0 * kSignBytes, img.data(), (size_t)rows * kSignBytes, hipMemcpyHostToDevice));
  }
  for (int64_t i = 0; i < n_pages; ++i) {
    ix->h_n_rows[first + i] = n_rows[i];
    ix->h_doc_ord[first + i] = doc_ordinals ? doc_ordinals[i] : 0;
  }
  if (n_pages) {
    MV_HIP(hipMemcpy(ix->d_n_rows + first, ix->h_n_rows.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice));
    MV_HIP(hipMemcpy(ix->d_doc_ord + first, ix->h_doc_ord.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice));
    publish_pages(ix, first, n_pages);
  }
  if (out_first_page) *out_first_page = first;
  return MV_OK;
}

int mv_index_remove_page(mv_index* ix, int64_t page) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->w_mu);
  if (page < 0 || page >= ix->size.load()) { set_error("page out of range"); return MV_ERR_INVALID; }
  DeviceGuard g(ix->cfg.device);
  __atomic_store_n(&ix->h_doc_ord[page], -1, __ATOMIC_RELAXED);  // queries read it under q_mu only (count_allowed_rows)
  ix->tombstones.store(true);  // before the device write: a scan that sees the tombstone also reads doc_ord
  MV_HIP(hipMemcpyAsync(ix->d_doc_ord + page, &ix->h_doc_ord[page], 4, hipMemcpyHostToDevice, ix->w_stream));
  MV_HIP(hipStreamSynchronize(ix->w_stream));
  return MV_OK;
}

int mv_index_remove_doc(mv_index* ix, int32_t doc_ordinal, int64_t* out_n) {
  if (!ix || doc_ordinal < 0) { set_error("bad doc ordinal"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->w_mu);
  DeviceGuard g(ix->cfg.device);
  int64_t n = 0, lo = -1, hi = -1;
  const int64_t size = ix->size.load();
  for (int64_t p = 0; p < size; ++p)
    if (ix->h_doc_ord[p] == doc_ordinal) {
      __atomic_store_n(&ix->h_doc_ord[p], -1, __ATOMIC_RELAXED);
      if (lo < 0) lo = p;
      hi = p;
      ++n;
    }
  if (n) {
    ix->tombstones.store(true);
    MV_HIP(hipMemcpyAsync(ix->d_doc_ord + lo, ix->h_doc_ord.data() + lo, (size_t)(hi - lo + 1) * 4, hipMemcpyHostToDevice, ix->w_stream));
    MV_HIP(hipStreamSynchronize(ix->w_stream));
  }
  if (out_n) *out_n = n;
  return MV_OK;
}

// Packed layout -> the fixed-stride image the readers hand out ([n][stride_rows] rows of row_bytes, zero rows behind a page's slot):
// one D2H copy of the pages' contiguous rows, then a host scatter.  q_mu held.
static int read_packed_pages(mv_index* ix, const void* slab, size_t row_bytes, int64_t page0, int64_t n_pages, void* out) {
  if (n_pages <= 0) return MV_OK;
  const int64_t r0 = page_row0(ix, page0), r1 = page_row0(ix, page0 + n_pages - 1) + page_slot_rows(ix, page0 + n_pages - 1);
  std::vector<char> tmp((size_t)(r1 - r0) * row_bytes);
  MV_HIP(hipMemcpy(tmp.data(), (const char*)slab + (size_t)r0 * row_bytes, tmp.size(), hipMemcpyDeviceToHost));
  const size_t pb = (size_t)ix->cfg.stride_rows * row_bytes;
  memset(out, 0, (size_t)n_pages * pb);
  for (int64_t i = 0; i < n_pages; ++i)
    memcpy((char*)out + (size_t)i * pb, tmp.data() + (size_t)(page_row0(ix, page0 + i) - r0) * row_bytes, (size_t)page_slot_rows(ix, page0 + i) * row_bytes);
  return MV_OK;
}

int mv_index_read_pages(mv_index* ix, int64_t page0, int64_t n_pages, void* out_bf16) {
  if (!ix || !out_bf16 || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_pages: range"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT))) { set_error("index has no float slab"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  const size_t pb = (size_t)ix->cfg.stride_rows * kRowBytes;
  if (!(ix->cfg.flags & MV_WITH_FLOAT)) return xt_read_pages(ix, page0, n_pages, out_bf16);  // exact host tier
  if (ix->packed) return read_packed_pages(ix, ix->slab, kRowBytes, page0, n_pages, out_bf16);
  MV_HIP(hipMemcpy(out_bf16, (const char*)ix->slab + (size_t)page0 * pb, (size_t)n_pages * pb, hipMemcpyDeviceToHost));
  return MV_OK;
}

int mv_index_read_pages_f32(mv_index* ix, int64_t page0, int64_t n_pages, float* out_f32) {
  if (!ix || !out_f32 || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_pages_f32: range"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT))) { set_error("index has no float slab"); return MV_ERR_STATE; }
  const size_t pe = (size_t)ix->cfg.stride_rows * kDim;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(((size_t)32 << 20) / (pe * 2)));
  std::vector<uint16_t> hi((size_t)std::min<int64_t>(chunk, std::max<int64_t>(n_pages, 1)) * pe), lo;
  for (int64_t done = 0; done < n_pages; done += chunk) {
    const int64_t c = std::min(chunk, n_pages - done);
    if (int rc = mv_index_read_pages(ix, page0 + done, c, hi.data())) return rc;
    float* o = out_f32 + (size_t)done * pe;
    for (size_t i = 0; i < (size_t)c * pe; ++i) o[i] = host_bf16_to_f32(hi[i]);
    if (ix->slab_lo) {
      lo.resize((size_t)c * pe);
      std::lock_guard<std::mutex> lk(ix->q_mu);
      DeviceGuard g(ix->cfg.device);
      if (ix->packed) { if (int rc = read_packed_pages(ix, ix->slab_lo, kRowBytes, page0 + done, c, lo.data())) return rc; }
      else MV_HIP(hipMemcpy(lo.data(), ix->slab_lo + (size_t)(page0 + done) * pe, (size_t)c * pe * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < (size_t)c * pe; ++i) o[i] += host_bf16_to_f32(lo[i]);  // exact: both halves came from ONE fp32 value
    }
  }
  return MV_OK;
}

int mv_index_write_rows(mv_index* ix, int64_t page, int32_t row0, int32_t n, const void* bf16_rows) {
  if (!ix || !bf16_rows || page < 0 || page >= ix->size.load() || row0 < 0 || n < 0 || row0 + n > ix->cfg.stride_rows) { set_error("write_rows: range"); return MV_ERR_INVALID; }
  if (ix->packed && row0 + n > page_slot_rows(ix, page)) { set_error("write_rows: rows %d..%d lie outside the page's %d-row slot (packed layout)", row0, row0 + n, page_slot_rows(ix, page)); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT))) { set_error("index has no float slab"); return MV_ERR_STATE; }
  if (!host_rows_finite(bf16_rows, MV_BF16, (size_t)n * kDim)) { set_error("write_rows: a row holds a NaN / Inf"); return MV_ERR_INVALID; }
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  if (int rc = xt_store_rows_from_host(ix, page, row0, n, bf16_rows, /*zero_rest=*/false)) return rc;
  if (!(ix->cfg.flags & MV_WITH_FLOAT)) return MV_OK;
  char* dst = (char*)ix->slab + ((size_t)page_row0(ix, page) + row0) * kRowBytes;
  MV_HIP(hipMemcpy(dst, bf16_rows, (size_t)n * kRowBytes, hipMemcpyHostToDevice));
  if (ix->slab_lo && n > 0) MV_HIP(hipMemset((char*)ix->slab_lo + ((size_t)page_row0(ix, page) + row0) * kRowBytes, 0, (size_t)n * kRowBytes));  // bf16 rows: lo = 0
  return MV_OK;
}

// Derive every enabled non-float slab of pages [first, first+n) from their bf16 image: `rows_ptr` = the first row of page `first`
// in an image laid out like the slabs (stride slots, or -- packed -- whole tiles back to back): the float slab itself, or a
// staging buffer.  d_nr = device row counts of those pages; their row offsets (packed) are already on the device.
static int derive_slabs_from_bf16(mv_index* ix, const uint16_t* rows_ptr, int64_t first, int64_t n, const int32_t* d_nr, hipStream_t st) {
  const int32_t stride = ix->cfg.stride_rows;
  const int64_t row0 = page_row0(ix, first), slot_rows = rows_in_use(ix, first + n) - row0;
  // packed: the kernels index absolute rows -- hand them the base those offsets count from (see add_pages_common)
  const uint16_t* img = ix->packed ? reinterpret_cast<const uint16_t*>(reinterpret_cast<uintptr_t>(rows_ptr) - (uintptr_t)row0 * kRowBytes) : rows_ptr;
  const int64_t* d_ro = ix->packed ? ix->d_row_off + first : nullptr;
  int rc = MV_OK;
  if (ix->cfg.flags & MV_WITH_BINARY) {
    rc = launch_sign_pack_bf16_rows(rows_ptr, slot_rows, ix->bits + (size_t)row0 * kSignBytes, st);
    if (rc) return rc;
  }
  if (ix->cfg.flags & MV_WITH_FP8) {
    rc = launch_quantize_pages_fp8(img, d_nr, stride, n, ix->packed ? ix->slab8 : ix->slab8 + (size_t)row0 * kDim, ix->inv_scale8 + first, st, d_ro);
    if (rc) return rc;
  }
  if (ix->cfg.flags & MV_WITH_FDE) {
    int64_t done = 0;
    while (done < n && !rc) {  // grid.x limit: chunk launches
      const int64_t c = std::min<int64_t>(n - done, 1 << 20);
      FdeEncodeArgs e{};
      e.variant = ix->fde_encode_variant;
      e.x_bf16 = ix->packed ? img : img + (size_t)done * stride * kDim; e.x_row_off = d_ro ? d_ro + done : nullptr;
      e.n_rows = d_nr + done; e.stride = stride; e.n_pages = c; e.is_query = 0;
      e.out_bf16 = ix->fde + (size_t)(first + done) * ix->fde_t.out_dim;
      e.out_inv_norm = ix->fde_inv_norm + first + done;
      rc = launch_fde_encode(ix->fde_t, e, st);
      done += c;
    }
    if (!rc) rc = fde8_requantize(ix, first, n, st);
  }
  return rc;
}

// n(u): the row count of unit u in a ragged synthetic corpus (mv_index_fill_synthetic_ragged; oracle/oracle.py restates it)
static inline int32_t synth_ragged_rows(uint64_t seed, uint64_t unit, int32_t min_rows, int32_t max_rows) {
  uint64_t z = seed ^ (0x9E3779B97F4A7C15ull * (unit + 1));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return min_rows + (int32_t)(z % (uint64_t)(max_rows - min_rows + 1));
}

// Shared body of the synthetic fills: page i (unit first_unit + i) gets rows_of(i) rows.  Caller checked the arguments.
static int fill_synthetic_common(mv_index* ix, uint64_t seed, uint64_t first_unit, int64_t n_pages, int32_t uniform_rows, int32_t min_rows, int32_t max_rows,
                                 int32_t pages_per_doc) {
  if (pages_per_doc < 1) pages_per_doc = 1;
  std::lock_guard<std::mutex> lk(ix->w_mu);
  if (ix->size.load() + n_pages > ix->cfg.capacity_pages) { set_error("slab full"); return MV_ERR_CAPACITY; }
  if (n_pages == 0) return MV_OK;
  DeviceGuard g(ix->cfg.device);
  hipStream_t ws = ix->w_stream;
  const int64_t first = ix->size.load();
  const int32_t stride = ix->cfg.stride_rows;
  const bool ragged_fill = uniform_rows < 0;
  int64_t slots = 0;
  for (int64_t i = 0; i < n_pages; ++i) {
    const int32_t nr = ragged_fill ? synth_ragged_rows(seed, first_unit + (uint64_t)i, min_rows, max_rows) : uniform_rows;
    ix->h_n_rows[first + i] = nr;
    ix->h_doc_ord[first + i] = (int32_t)((first_unit + (uint64_t)i) / (uint64_t)pages_per_doc);
    slots += ((int64_t)nr + 15) / 16 * 16;
  }
  if (ix->packed) {
    if (rows_in_use(ix, first) + slots > ix->cap_rows) { set_error("slab full: %lld rows in use + %lld > capacity_rows %lld", (long long)rows_in_use(ix, first), (long long)slots, (long long)ix->cap_rows); return MV_ERR_CAPACITY; }
    for (int64_t i = 0; i < n_pages; ++i) ix->h_row_off[(size_t)(first + i + 1)] = ix->h_row_off[(size_t)(first + i)] + ((int64_t)ix->h_n_rows[first + i] + 15) / 16 * 16;
    MV_HIP(hipMemcpyAsync(ix->d_row_off + first, ix->h_row_off.data() + first, (size_t)(n_pages + 1) * 8, hipMemcpyHostToDevice, ws));
  }
  MV_HIP(hipMemcpyAsync(ix->d_n_rows + first, ix->h_n_rows.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice, ws));
  MV_HIP(hipMemcpyAsync(ix->d_doc_ord + first, ix->h_doc_ord.data() + first, (size_t)n_pages * 4, hipMemcpyHostToDevice, ws));
  const bool has_float = (ix->cfg.flags & MV_WITH_FLOAT) != 0;
  // without a float slab the bf16 image is staged chunk by chunk (<= 512 MiB) and only its derivatives are kept
  const int64_t max_slot = ix->packed ? ((int64_t)(ragged_fill ? max_rows : uniform_rows) + 15) / 16 * 16 : stride;
  const int64_t chunk = has_float ? n_pages : std::max<int64_t>(1, ((int64_t)512 << 20) / (std::max<int64_t>(max_slot, 16) * kRowBytes));
  int rc = MV_OK;
  if (!has_float) rc = w_reserve(&ix->w_tmp, &ix->w_tmp_bytes, (size_t)std::min(chunk, n_pages) * std::max<int64_t>(max_slot, 16) * kRowBytes);
  for (int64_t done = 0; done < n_pages && !rc; done += chunk) {
    const int64_t c = std::min(chunk, n_pages - done);
    const int64_t row0 = page_row0(ix, first + done), slot_rows = rows_in_use(ix, first + done + c) - row0;
    uint16_t* rows_ptr = has_float ? ix->slab + (size_t)row0 * kDim : (uint16_t*)ix->w_tmp;
    if (!ragged_fill) {
      // uniform pages: a packed slot is ceil16(n_rows) rows -- a fixed-stride region of that stride
      rc = launch_synth_rows(rows_ptr, seed, first_unit + (uint64_t)done, c, uniform_rows, ix->packed ? (int32_t)max_slot : stride, ws);
    } else {
      uint16_t* img = ix->packed ? reinterpret_cast<uint16_t*>(reinterpret_cast<uintptr_t>(rows_ptr) - (uintptr_t)row0 * kRowBytes) : rows_ptr;
      rc = launch_synth_rows_ragged(img, seed, first_unit + (uint64_t)done, c, ix->d_n_rows + first + done, ix->packed ? ix->d_row_off + first + done : nullptr, stride, ws);
    }
    if (!rc && ix->slab_lo && hipMemsetAsync(ix->slab_lo + (size_t)row0 * kDim, 0, (size_t)slot_rows * kRowBytes, ws) != hipSuccess) {
      set_error("fill_synthetic: clearing the lo slab failed"); rc = MV_ERR_HIP;  // the generator's rows ARE bf16: lo = 0
    }
    if (!rc) rc = xt_store_from_device(ix, rows_ptr, first + done, c, ws);
    if (!rc) rc = derive_slabs_from_bf16(ix, rows_ptr, first + done, c, ix->d_n_rows + first + done, ws);
    if (!has_float && hipStreamSynchronize(ws) != hipSuccess && !rc) { set_error("fill_synthetic: stream error"); rc = MV_ERR_HIP; }
  }
  hipError_t e = hipStreamSynchronize(ws);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "fill_synthetic", __FILE__, __LINE__);
  publish_pages(ix, first, n_pages);
  return MV_OK;
}

int mv_index_fill_synthetic(mv_index* ix, uint64_t seed, uint64_t first_unit, int64_t n_pages, int32_t n_rows,
                            int32_t pages_per_doc) {
  if (!ix || n_pages < 0 || n_rows < 0 || n_rows > ix->cfg.stride_rows) { set_error("fill_synthetic: bad argument"); return MV_ERR_INVALID; }
  return fill_synthetic_common(ix, seed, first_unit, n_pages, n_rows, 0, 0, pages_per_doc);
}

int mv_index_fill_synthetic_ragged(mv_index* ix, uint64_t seed, uint64_t first_unit, int64_t n_pages, int32_t min_rows, int32_t max_rows,
                                   int32_t pages_per_doc) {
  if (!ix || n_pages < 0 || min_rows < 0 || max_rows < min_rows || max_rows > ix->cfg.stride_rows) { set_error("fill_synthetic_ragged: bad argument (0 <= min_rows <= max_rows <= stride_rows)"); return MV_ERR_INVALID; }
  return fill_synthetic_common(ix, seed, first_unit, n_pages, -1, min_rows, max_rows, pages_per_doc);
}

// Overwrite one whole page from host bf16 rows and refresh every slab (bench/test: planted neighbours on any
// combination of slabs; also the update path of a re-embedded page).
int mv_index_replace_page(mv_index* ix, int64_t page, const void* bf16_rows, int32_t n_rows) {
  if (!ix || !bf16_rows || page < 0 || page >= ix->size.load() || n_rows < 0 || n_rows > ix->cfg.stride_rows) { set_error("replace_page: bad argument"); return MV_ERR_INVALID; }
  if ((ix->cfg.flags & ~(MV_WITH_BINARY | MV_LAYOUT_PACKED)) && !host_rows_finite(bf16_rows, MV_BF16, (size_t)n_rows * kDim)) { set_error("replace_page: a row holds a NaN / Inf"); return MV_ERR_INVALID; }
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  const int32_t stride = ix->cfg.stride_rows;
  // packed layout: a page owns the tiles it was appended with -- a longer replacement does not fit (remove the page and append the new one)
  const int32_t slot = page_slot_rows(ix, page);
  if (n_rows > slot) { set_error("replace_page: %d rows do not fit the page's %d-row slot (packed layout: remove the page and append its replacement)", n_rows, slot); return MV_ERR_CAPACITY; }
  const int64_t row0 = page_row0(ix, page);
  const bool has_float = (ix->cfg.flags & MV_WITH_FLOAT) != 0;
  uint16_t* dst = has_float ? ix->slab + (size_t)row0 * kDim : nullptr;
  uint16_t* stage = nullptr;
  if (!has_float) { MV_HIP(hipMalloc(&stage, (size_t)slot * kRowBytes)); dst = stage; }
  int rc = MV_OK;
  hipError_t e = hipMemsetAsync(dst, 0, (size_t)slot * kRowBytes, ix->stream);
  if (e == hipSuccess && ix->slab_lo) e = hipMemsetAsync(ix->slab_lo + (size_t)row0 * kDim, 0, (size_t)slot * kRowBytes, ix->stream);  // bf16 rows: lo = 0
  if (e == hipSuccess && n_rows > 0) e = hipMemcpyAsync(dst, bf16_rows, (size_t)n_rows * kRowBytes, hipMemcpyHostToDevice, ix->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);  // the host buffer may be pageable
  if (e != hipSuccess) rc = hip_fail(e, "replace_page upload", __FILE__, __LINE__);
  if (!rc) {
    ix->h_n_rows[page] = n_rows;
    if (n_rows != stride) ix->ragged.store(true);
    (void)stride;
    e = hipMemcpyAsync(ix->d_n_rows + page, &ix->h_n_rows[page], 4, hipMemcpyHostToDevice, ix->stream);
    if (e != hipSuccess) rc = hip_fail(e, "replace_page metadata", __FILE__, __LINE__);
  }
  if (!rc) rc = xt_store_rows_from_host(ix, page, 0, n_rows, bf16_rows, /*zero_rest=*/true);
  if (!rc) rc = derive_slabs_from_bf16(ix, dst, page, 1, ix->d_n_rows + page, ix->stream);
  (void)hipStreamSynchronize(ix->stream);
  if (stage) (void)hipFree(stage);
  return rc;
}

// ---------------------------------------------------------------------------------- compaction
__global__ __launch_bounds__(256) void gather_pages_kernel(const char* base, size_t page_bytes, const int64_t* src_pages, int64_t n,
                                                           char* dst) {
  // grid.x = pages of the batch, grid.y = 16-byte lanes of a page in units of 256 threads
  const int64_t p = blockIdx.x;
  const size_t off = ((size_t)blockIdx.y * 256 + threadIdx.x) * 16;
  if (p >= n || off >= page_bytes) return;
  *reinterpret_cast<uint4*>(dst + (size_t)p * page_bytes + off) =
      *reinterpret_cast<const uint4*>(base + (size_t)src_pages[p] * page_bytes + off);
}

// Packed layout: page p of the batch = rows [src_row0[p], +rows[p]) of a row-indexed slab -> stage + dst_rel[p] * row_bytes.
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* base, size_t row_bytes, const int64_t* src_row0, const int64_t* dst_rel,
                                                          const int64_t* rows, int64_t n, char* dst) {
  const int64_t p = blockIdx.x;
  const size_t off = ((size_t)blockIdx.y * 256 + threadIdx.x) * 16;
  if (p >= n || off >= (size_t)rows[p] * row_bytes) return;
  *reinterpret_cast<uint4*>(dst + (size_t)dst_rel[p] * row_bytes + off) = *reinterpret_cast<const uint4*>(base + (size_t)src_row0[p] * row_bytes + off);
}

// Reclaim the slots of tombstoned pages: live pages move down, in order, to a dense prefix of every slab.
// old_to_new[old page] = new page or -1 (removed); the caller remaps its ids (store.py does).  In place: batches of
// live pages are gathered into a staging buffer and written back as one contiguous block -- sources are always at or
// above their destination and batches ascend, so a write never lands on a page that is still to be read.
int mv_index_compact(mv_index* ix, int64_t* out_old_to_new, int64_t* out_new_size) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  ExclusiveLock lk(ix);
  DeviceGuard g(ix->cfg.device);
  const int64_t n = ix->size.load();
  std::vector<int64_t> live;
  live.reserve((size_t)n);
  for (int64_t p = 0; p < n; ++p) {
    const bool alive = ix->h_doc_ord[p] >= 0;
    if (out_old_to_new) out_old_to_new[p] = alive ? (int64_t)live.size() : -1;
    if (alive) live.push_back(p);
  }
  const int64_t m = (int64_t)live.size();
  if (out_new_size) *out_new_size = m;
  if (m == n) return MV_OK;  // nothing to reclaim
  int64_t first_moved = 0;
  while (first_moved < m && live[first_moved] == first_moved) ++first_moved;
  struct Slab { char* base; size_t page_bytes; size_t row_bytes; };  // row_bytes > 0: a row-indexed slab (packed layout: ragged slots)
  std::vector<Slab> slabs;
  const size_t stride = (size_t)ix->cfg.stride_rows;
  if (ix->cfg.flags & MV_WITH_FLOAT) slabs.push_back({(char*)ix->slab, stride * kRowBytes, (size_t)kRowBytes});
  if (ix->slab_lo) slabs.push_back({(char*)ix->slab_lo, stride * kRowBytes, (size_t)kRowBytes});
  if (ix->cfg.flags & MV_WITH_FP8) { slabs.push_back({(char*)ix->slab8, stride * kDim, (size_t)kDim}); slabs.push_back({(char*)ix->inv_scale8, 16, 0}); }
  if (ix->cfg.flags & MV_WITH_BINARY) slabs.push_back({(char*)ix->bits, stride * kSignBytes, (size_t)kSignBytes});
  if (ix->cfg.flags & MV_WITH_FDE) { slabs.push_back({(char*)ix->fde, (size_t)ix->fde_t.out_dim * 2, 0}); slabs.push_back({(char*)ix->fde_inv_norm, 16, 0}); }
  if (ix->fde4) { slabs.push_back({(char*)ix->fde4, (size_t)ix->fde_t.out_dim / 2, 0}); slabs.push_back({(char*)ix->fde4_scale, 16, 0}); slabs.push_back({(char*)ix->fde4_cfac, 16, 0}); }
  if (ix->fde8) { slabs.push_back({(char*)ix->fde8, (size_t)ix->fde_t.out_dim, 0}); slabs.push_back({(char*)ix->fde8_scale, 16, 0}); slabs.push_back({(char*)ix->fde8_cfac, 16, 0}); }
  int rc = MV_OK;
  // packed layout: the new row offsets (live pages keep their slots, back to back) and, per moved page, (old first row, new first row, rows)
  std::vector<int64_t> new_off;
  int64_t* d_rowmeta = nullptr;  // [3][moved]: src_row0 | dst_row0 | rows
  const int64_t moved = m - first_moved;
  if (ix->packed) {
    new_off.assign((size_t)m + 1, 0);
    for (int64_t j = 0; j < m; ++j) new_off[(size_t)j + 1] = new_off[(size_t)j] + page_slot_rows(ix, live[(size_t)j]);
    if (moved > 0) {
      std::vector<int64_t> meta((size_t)3 * moved);
      for (int64_t j = first_moved; j < m; ++j) {
        meta[(size_t)(j - first_moved)] = ix->h_row_off[(size_t)live[(size_t)j]];
        meta[(size_t)(moved + j - first_moved)] = new_off[(size_t)j];
        meta[(size_t)(2 * moved + j - first_moved)] = page_slot_rows(ix, live[(size_t)j]);
      }
      if (hipMalloc(&d_rowmeta, meta.size() * 8) != hipSuccess || hipMemcpy(d_rowmeta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("compact: out of device memory for the row tables"); rc = MV_ERR_NOMEM;
      }
    }
  }
  int64_t* d_idx = nullptr;
  char* stage = nullptr;
  const size_t stage_bytes = (size_t)256 << 20;
  if (hipMalloc(&d_idx, (size_t)std::max<int64_t>(m - first_moved, 1) * 8) != hipSuccess || hipMalloc(&stage, stage_bytes) != hipSuccess) {
    set_error("compact: out of device memory for the staging buffer");
    rc = MV_ERR_NOMEM;
  }
  if (!rc && hipMemcpy(d_idx, live.data() + first_moved, (size_t)(m - first_moved) * 8, hipMemcpyHostToDevice) != hipSuccess) { set_error("compact: H2D failed"); rc = MV_ERR_HIP; }
  for (const Slab& sl : slabs) {
    if (rc) break;
    if (sl.page_bytes == 16) {
      // per-page scalars (fp32): 4-byte records -- move on the host side of a small round trip
      std::vector<float> h((size_t)n);
      if (hipMemcpy(h.data(), sl.base, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("compact: D2H failed"); rc = MV_ERR_HIP; break; }
      for (int64_t j = first_moved; j < m; ++j) h[(size_t)j] = h[(size_t)live[j]];
      if (hipMemcpy(sl.base, h.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("compact: H2D failed"); rc = MV_ERR_HIP; }
      continue;
    }
    if (ix->packed && sl.row_bytes) {
      // ragged slots: batches of consecutive live pages whose slots fill the staging buffer; gathered, then written back as ONE block at
      // the batch's new first row (destinations never lie above their sources and batches ascend: nothing still to be read is overwritten)
      for (int64_t j0 = first_moved; j0 < m && !rc;) {
        int64_t j1 = j0;
        const int64_t rbase = new_off[(size_t)j0];
        while (j1 < m && (size_t)(new_off[(size_t)j1 + 1] - rbase) * sl.row_bytes <= stage_bytes) ++j1;
        if (j1 == j0) { set_error("compact: a page slot exceeds the staging buffer"); rc = MV_ERR_INVALID; break; }
        const int64_t c = j1 - j0;
        const unsigned gy = (unsigned)((stride * sl.row_bytes / 16 + 255) / 256);
        // dst_rel = new first row relative to the batch: computed on the fly from the dst_row0 table by offsetting the stage pointer
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)c, gy), dim3(256), 0, ix->stream, (const char*)sl.base, sl.row_bytes,
                           (const int64_t*)(d_rowmeta + (j0 - first_moved)), (const int64_t*)(d_rowmeta + moved + (j0 - first_moved)),
                           (const int64_t*)(d_rowmeta + 2 * moved + (j0 - first_moved)), c, stage - (size_t)rbase * sl.row_bytes);
        if (hipMemcpyAsync(sl.base + (size_t)rbase * sl.row_bytes, stage, (size_t)(new_off[(size_t)j1] - rbase) * sl.row_bytes, hipMemcpyDeviceToDevice, ix->stream) != hipSuccess ||
            hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("compact: device copy failed"); rc = MV_ERR_HIP; }
        j0 = j1;
      }
      continue;
    }
    const int64_t batch = std::max<int64_t>(1, (int64_t)(stage_bytes / sl.page_bytes));
    for (int64_t j0 = first_moved; j0 < m && !rc; j0 += batch) {
      const int64_t c = std::min(batch, m - j0);
      const unsigned gy = (unsigned)((sl.page_bytes / 16 + 255) / 256);
      hipLaunchKernelGGL(gather_pages_kernel, dim3((unsigned)c, gy), dim3(256), 0, ix->stream, (const char*)sl.base, sl.page_bytes,
                         (const int64_t*)(d_idx + (j0 - first_moved)), c, stage);
      if (hipMemcpyAsync(sl.base + (size_t)j0 * sl.page_bytes, stage, (size_t)c * sl.page_bytes, hipMemcpyDeviceToDevice, ix->stream) != hipSuccess ||
          hipStreamSynchronize(ix->stream) != hipSuccess) { set_error("compact: device copy failed"); rc = MV_ERR_HIP; }
    }
  }
  if (d_idx) (void)hipFree(d_idx);
  if (stage) (void)hipFree(stage);
  if (d_rowmeta) (void)hipFree(d_rowmeta);
  if (rc) return rc;
  if (ix->packed) {
    for (int64_t j = 0; j <= m; ++j) ix->h_row_off[(size_t)j] = new_off[(size_t)j];
    for (int64_t j = m + 1; j <= n; ++j) ix->h_row_off[(size_t)j] = new_off[(size_t)m];
    MV_HIP(hipMemcpy(ix->d_row_off, ix->h_row_off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
  }
  if ((rc = xt_compact(ix, live, first_moved, m)) != MV_OK) return rc;  // the exact host tier moves with the pages
  if (ix->d_xhits) MV_HIP(hipMemset(ix->d_xhits, 0, (size_t)ix->cfg.capacity_pages * 4));  // page ids changed: the read counts start over
  for (int64_t j = first_moved; j < m; ++j) {
    ix->h_n_rows[(size_t)j] = ix->h_n_rows[(size_t)live[j]];
    ix->h_doc_ord[(size_t)j] = ix->h_doc_ord[(size_t)live[j]];
  }
  for (int64_t j = m; j < n; ++j) { ix->h_n_rows[(size_t)j] = 0; ix->h_doc_ord[(size_t)j] = -1; }
  ix->size.store(m, std::memory_order_release);
  ix->tombstones.store(false);
  bool rag = false;
  for (int64_t j = 0; j < m && !rag; ++j) rag = ix->h_n_rows[(size_t)j] != ix->cfg.stride_rows;
  ix->ragged.store(rag || ix->packed);
  if (n > 0) {
    MV_HIP(hipMemcpy(ix->d_n_rows, ix->h_n_rows.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    MV_HIP(hipMemcpy(ix->d_doc_ord, ix->h_doc_ord.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  return MV_OK;
}

// Read back the e4m3 codes (stride_rows x 128 bytes per page) and 2^-e scales of pages [page0, page0+n).
int mv_index_read_fp8(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_inv_scale) {
  if (!ix || !out_codes || !out_inv_scale || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_fp8: range"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has no fp8 slab"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  const size_t pb = (size_t)ix->cfg.stride_rows * kDim;
  if (ix->packed) { if (int rc = read_packed_pages(ix, ix->slab8, kDim, page0, n_pages, out_codes)) return rc; }
  else MV_HIP(hipMemcpy(out_codes, ix->slab8 + (size_t)page0 * pb, (size_t)n_pages * pb, hipMemcpyDeviceToHost));
  MV_HIP(hipMemcpy(out_inv_scale, ix->inv_scale8 + page0, (size_t)n_pages * 4, hipMemcpyDeviceToHost));
  return MV_OK;
}

int mv_synth_rows(int device, uint64_t seed, uint64_t unit, int32_t n_rows, void* out_bf16) {
  if (!out_bf16 || n_rows < 1) { set_error("mv_synth_rows: bad argument"); return MV_ERR_INVALID; }
  DeviceGuard g(device);
  uint16_t* d = nullptr;
  MV_HIP(hipMalloc(&d, (size_t)n_rows * kRowBytes));
  int rc = launch_synth_rows(d, seed, unit, 1, n_rows, n_rows, nullptr);
  if (!rc) {
    hipError_t e = hipMemcpy(out_bf16, d, (size_t)n_rows * kRowBytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail(e, "D2H", __FILE__, __LINE__);
  }
  (void)hipFree(d);
  return rc;
}

// Shared body of the top-k entry points (also used by mv_comm.hip).  defer_stats: fill the accounting fields of `st` but
// leave the event timings to a later mv::finish_stats(ix, st, true) -- the call then only enqueues.
static void swap_timing_event_sets(mv_index* ix) {
  for (int i = 0; i < 3; ++i) { std::swap(ix->ev[i], ix->ev_alt[i]); std::swap(ix->ev_st[i], ix->ev_st_alt[i]); }
  ix->ev_parity ^= 1;
}

int mv_internal_query_common(mv_index* ix, const void* q, int q_dtype, int32_t n_q, int32_t k, int mode,
                             const uint32_t* allow_bits, int64_t n_words, float* h_scores, int64_t* h_ids, int32_t* out_n,
                             float* d_scores_out, int64_t* d_ids_out, void* user_stream, mv_query_stats* st, int defer_stats) {
  if (!ix) { set_error("null index"); return MV_ERR_INVALID; }
  if (k < 0) { set_error("k must be >= 0"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  if (out_n) *out_n = 0;
  const bool to_device = d_scores_out != nullptr;
  if (to_device && k > kTopkMaxDeviceK) { set_error("device-output top-k supports k <= %d", kTopkMaxDeviceK); return MV_ERR_INVALID; }
  const bool ordered = user_stream != nullptr;  // kNullStreamTag: ordered against the null stream itself
  if (user_stream == kNullStreamTag) user_stream = nullptr;
  if (st && defer_stats) swap_timing_event_sets(ix);  // the previous deferred query's events stay readable (mv_query_stats_finish)
  if (ordered) {  // order our stream behind the caller's
    MV_HIP(hipEventRecord(ix->ev[3], (hipStream_t)user_stream));
    MV_HIP(hipStreamWaitEvent(ix->stream, ix->ev[3], 0));
  }
  if (k == 0 || ix->size.load(std::memory_order_acquire) == 0) {
    if (to_device) {
      std::vector<float> s((size_t)std::max(k, 1), -INFINITY);
      std::vector<int64_t> id((size_t)std::max(k, 1), -1);
      if (k > 0) {
        MV_HIP(hipMemcpy(d_scores_out, s.data(), (size_t)k * 4, hipMemcpyHostToDevice));
        MV_HIP(hipMemcpy(d_ids_out, id.data(), (size_t)k * 8, hipMemcpyHostToDevice));
      }
    }
    if (st) memset(st, 0, sizeof(*st));
    if (st && defer_stats) {  // nothing was enqueued: no events to read later -- the record is complete (all zeros) as it stands
      swap_timing_event_sets(ix);
      st->reserved = kStatsDoneTag;
    }
    return MV_OK;
  }
  ScanResult r;
  const int64_t want_n = mode == MV_MODE_FP8_THEN_FLOAT ? std::max<int64_t>(ix->rerank_n, k) : coarse_n_for(ix, k);
  ix->sync_call = !ordered;  // every unordered path below ends in a hipStreamSynchronize (q_mu is held)
  int rc = run_scan(ix, q, q_dtype, n_q, mode, allow_bits, n_words, want_n, &r, st, /*want_compact=*/true, k);
  ix->sync_call = false;
  if (rc) return rc;
  const int64_t id_base = ix->cfg.id_base;
  if (r.n == 0) {
    if (!ordered) MV_HIP(hipStreamSynchronize(ix->stream));  // the query's H2D copies: the staging buffers are free when we return
    if (to_device) {
      std::vector<float> s((size_t)k, -INFINITY);
      std::vector<int64_t> id((size_t)k, -1);
      MV_HIP(hipMemcpy(d_scores_out, s.data(), (size_t)k * 4, hipMemcpyHostToDevice));
      MV_HIP(hipMemcpy(d_ids_out, id.data(), (size_t)k * 8, hipMemcpyHostToDevice));
    }
    rc = finish_stats(ix, st, false);
    if (st && defer_stats) st->reserved = kStatsDoneTag;  // finished here (no selection ran): mv_query_stats_finish has nothing left to read
    return rc;
  }
  if (k <= kTopkMaxDeviceK) {
    // host results: the selection's last kernel writes its k pairs straight into the pinned result buffers (device-visible host memory;
    // visible to the host once the stream has drained) -- two D2H blit kernels and an event gap less at the end of every request
    const bool direct = !to_device && ix->hd_out_s != nullptr && ix->hd_out_id != nullptr;
    float* ds = to_device ? d_scores_out : (direct ? ix->hd_out_s : ix->d_out_s);
    int64_t* di = to_device ? d_ids_out : (direct ? ix->hd_out_id : ix->d_out_id);
    rc = launch_topk(r.d_scores, r.n, k, r.d_ids_map, id_base, ix->d_topk_ws, ds, di, ix->stream);
    if (rc) return rc;
    if (st || (to_device && ordered)) MV_HIP(hipEventRecord(ix->ev[2], ix->stream));  // timings / the caller's stream waits on it
    if (to_device) {
      if (ordered) {  // caller's stream waits for our result
        MV_HIP(hipStreamWaitEvent((hipStream_t)user_stream, ix->ev[2], 0));
      } else {
        MV_HIP(hipStreamSynchronize(ix->stream));
      }
      if (st && defer_stats) st->reserved |= kStatsDeferredTag | (ix->ev_parity ? kStatsParityBit : 0);
      return (st && !defer_stats) ? finish_stats(ix, st, true) : MV_OK;
    }
    float* s = ix->h_out_s;
    int64_t* id = ix->h_out_id;
    if (!direct) {
      MV_HIP(hipMemcpyAsync(s, ds, (size_t)k * 4, hipMemcpyDeviceToHost, ix->stream));
      MV_HIP(hipMemcpyAsync(id, di, (size_t)k * 8, hipMemcpyDeviceToHost, ix->stream));
    }
    MV_HIP(hipStreamSynchronize(ix->stream));
    int32_t n = 0;
    while (n < k && id[n] >= 0) ++n;
    memcpy(h_scores, s, (size_t)n * 4);
    memcpy(h_ids, id, (size_t)n * 8);
    if (out_n) *out_n = n;
    return finish_stats(ix, st, true);
  }
  // k beyond the device selection kernel: copy the score vector and select on the host
  std::vector<float> s((size_t)r.n);
  MV_HIP(hipMemcpyAsync(s.data(), r.d_scores, (size_t)r.n * 4, hipMemcpyDeviceToHost, ix->stream));
  std::vector<int32_t> map;
  if (r.d_ids_map) {
    map.resize((size_t)r.n);
    MV_HIP(hipMemcpyAsync(map.data(), r.d_ids_map, (size_t)r.n * 4, hipMemcpyDeviceToHost, ix->stream));
  }
  MV_HIP(hipStreamSynchronize(ix->stream));
  std::vector<float> os;
  std::vector<int64_t> oi;
  host_topk(s, r.d_ids_map ? &map : nullptr, id_base, k, &os, &oi);
  memcpy(h_scores, os.data(), os.size() * 4);
  memcpy(h_ids, oi.data(), oi.size() * 8);
  if (out_n) *out_n = (int32_t)os.size();
  return finish_stats(ix, st, false);
}

int mv_query_topk(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                  const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n,
                  mv_query_stats* stats) {
  if (k > 0 && (!out_scores || !out_ids)) { set_error("null output buffer"); return MV_ERR_INVALID; }
  return mv_internal_query_common(ix, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, out_scores, out_ids, out_n, nullptr,
                                  nullptr, nullptr, stats, 0);
}

int mv_query_topk_device(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                         const uint32_t* allow_bits, int64_t n_allow_words, float* d_out_scores, int64_t* d_out_ids,
                         void* stream, mv_query_stats* stats) {
  if (k < 1 || !d_out_scores || !d_out_ids) { set_error("mv_query_topk_device: k >= 1 and device buffers required"); return MV_ERR_INVALID; }
  return mv_internal_query_common(ix, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, nullptr, nullptr, nullptr, d_out_scores,
                                  d_out_ids, stream, stats, 0);
}

int mv_query_topk_device_async(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int32_t k, int mode,
                               const uint32_t* allow_bits, int64_t n_allow_words, float* d_out_scores, int64_t* d_out_ids,
                               void* stream, mv_query_stats* stats) {
  if (k < 1 || !d_out_scores || !d_out_ids || !stats) { set_error("mv_query_topk_device_async: k >= 1, device buffers and a stats record required"); return MV_ERR_INVALID; }
  if (!stream) stream = kNullStreamTag;  // 0 names the default stream here (torch's default stream is the null stream): order against it, do not block
  return mv_internal_query_common(ix, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, nullptr, nullptr, nullptr, d_out_scores,
                                  d_out_ids, stream, stats, /*defer_stats=*/1);
}

int mv_query_stats_finish(mv_index* ix, mv_query_stats* stats) {
  if (!ix || !stats) { set_error("mv_query_stats_finish: null argument"); return MV_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  if (stats->reserved & kStatsDoneTag) { stats->reserved = 0; return MV_OK; }  // a deferred query that had nothing to enqueue (empty shard, no allowed page)
  const bool tagged = (stats->reserved & kStatsDeferredTag) != 0;
  const int parity = (stats->reserved & kStatsParityBit) ? 1 : 0;
  stats->reserved &= ~(kStatsDeferredTag | kStatsParityBit);
  if (!tagged || parity == ix->ev_parity) return finish_stats(ix, stats, true);  // the most recent query: the current event set
  // ONE query behind: its events are in the other set.  The stage accounting of the FDE modes reads the candidate list, which the newer
  // query has overwritten by now: such a record must be finished before the next query is enqueued
  if (stats->reserved) { set_error("mv_query_stats_finish: the stage accounting of an FDE-mode query must be collected before the next query on this index"); return MV_ERR_STATE; }
  swap_timing_event_sets(ix);
  const int rc = finish_stats(ix, stats, true);
  swap_timing_event_sets(ix);
  return rc;
}

// ---- the caller's own FDE vectors (the reference computes them with its `fde` extension: fast_multivector_store.py:447-449, :521)
static int check_fde_call(const mv_index* ix, int mode, const float* q_fde, int64_t n_queries, const char* what) {
  if (!ix || !q_fde) { set_error("%s: null argument", what); return MV_ERR_INVALID; }
  if (mode != MV_MODE_FDE_THEN_FLOAT && mode != MV_MODE_FDE_ONLY) { set_error("%s: mode %d has no FDE stage (MV_MODE_FDE_THEN_FLOAT / MV_MODE_FDE_ONLY)", what, mode); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }
  return check_fde_finite(q_fde, (size_t)n_queries * (size_t)ix->fde_t.out_dim, what);
}

int mv_query_topk_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const float* q_fde, int32_t k, int mode,
                      const uint32_t* allow_bits, int64_t n_allow_words, float* out_scores, int64_t* out_ids, int32_t* out_n, mv_query_stats* stats) {
  if (int rc = check_fde_call(ix, mode, q_fde, 1, "mv_query_topk_fde")) return rc;
  QueryFdeScope sc(q_fde);
  return mv_query_topk(ix, q, q_dtype, n_q_rows, k, mode, allow_bits, n_allow_words, out_scores, out_ids, out_n, stats);
}

int mv_query_topk_batch_fde(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, const float* q_fde, int32_t k, int mode,
                            const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores, int64_t* out_ids,
                            int32_t* out_n, mv_query_stats* stats) {
  if (n_queries < 1) { set_error("mv_query_topk_batch_fde: bad argument"); return MV_ERR_INVALID; }
  if (int rc = check_fde_call(ix, mode, q_fde, n_queries, "mv_query_topk_batch_fde")) return rc;
  QueryFdeScope sc(q_fde);
  return mv_query_topk_batch(ix, q, q_dtype, n_queries, n_q_rows, k, mode, allow_bits, n_allow_words, allow_per_query, out_scores, out_ids, out_n, stats);
}

int mv_index_import_fde(mv_index* ix, int64_t page0, int64_t n_pages, const float* fde) {
  if (!ix || !fde || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("import_fde: range"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }
  const int64_t out_dim = ix->fde_t.out_dim;
  if (int rc = check_fde_finite(fde, (size_t)n_pages * (size_t)out_dim, "import_fde")) return rc;
  ExclusiveLock lk(ix);  // queries read the FDE slab without the writer lock
  DeviceGuard g(ix->cfg.device);
  const int64_t chunk = std::max<int64_t>(1, ((int64_t)64 << 20) / (out_dim * 4));  // 64 MiB of fp32 vectors per staging round
  int rc = w_reserve(&ix->w_tmp, &ix->w_tmp_bytes, (size_t)std::min(chunk, std::max<int64_t>(n_pages, 1)) * out_dim * 4);
  for (int64_t done = 0; done < n_pages && !rc; done += chunk) {
    const int64_t c = std::min(chunk, n_pages - done);
    MV_HIP(hipMemcpyAsync(ix->w_tmp, fde + (size_t)done * out_dim, (size_t)c * out_dim * 4, hipMemcpyHostToDevice, ix->w_stream));
    rc = launch_fde_import((const float*)ix->w_tmp, c, out_dim, ix->fde + (size_t)(page0 + done) * out_dim, ix->fde_inv_norm + page0 + done, ix->w_stream);
    if (!rc) rc = fde8_requantize(ix, page0 + done, c, ix->w_stream);
    MV_HIP(hipStreamSynchronize(ix->w_stream));  // the staging buffer is reused; the caller's buffer may be pageable
  }
  return rc;
}

int mv_index_read_fde(mv_index* ix, int64_t page0, int64_t n_pages, float* out) {
  if (!ix || !out || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_fde: range"); return MV_ERR_INVALID; }
  if (!(ix->cfg.flags & MV_WITH_FDE)) { set_error("index has no FDE slab (MV_WITH_FDE)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  const size_t n = (size_t)n_pages * (size_t)ix->fde_t.out_dim;
  std::vector<uint16_t> h(n);
  MV_HIP(hipMemcpy(h.data(), ix->fde + (size_t)page0 * ix->fde_t.out_dim, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    const uint32_t u = (uint32_t)h[i] << 16;
    memcpy(&out[i], &u, 4);
  }
  return MV_OK;
}

int mv_index_read_fde_fp4(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_scale) {
  if (!ix || !out_codes || !out_scale || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_fde_fp4: range"); return MV_ERR_INVALID; }
  if (!ix->fde4) { set_error("index has no fp4 copy of the FDE slab (MV_WITH_FDE_FP4)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  const size_t rb = (size_t)ix->fde_t.out_dim / 2;
  MV_HIP(hipMemcpy(out_codes, ix->fde4 + (size_t)page0 * rb, (size_t)n_pages * rb, hipMemcpyDeviceToHost));
  MV_HIP(hipMemcpy(out_scale, ix->fde4_scale + page0, (size_t)n_pages * 4, hipMemcpyDeviceToHost));
  return MV_OK;
}

int mv_index_read_fde_e4m3(mv_index* ix, int64_t page0, int64_t n_pages, void* out_codes, float* out_scale) {
  if (!ix || !out_codes || !out_scale || page0 < 0 || n_pages < 0 || page0 + n_pages > ix->size.load()) { set_error("read_fde_e4m3: range"); return MV_ERR_INVALID; }
  if (!ix->fde8) { set_error("index has no e4m3 copy of the FDE slab (MV_WITH_FDE_E4M3)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  MV_HIP(hipMemcpy(out_codes, ix->fde8 + (size_t)page0 * ix->fde_t.out_dim, (size_t)n_pages * ix->fde_t.out_dim, hipMemcpyDeviceToHost));
  MV_HIP(hipMemcpy(out_scale, ix->fde8_scale + page0, (size_t)n_pages * 4, hipMemcpyDeviceToHost));
  return MV_OK;
}

// Selection workspace of the batched entry point: one top-k workspace, result row and pinned read-back row per query of a
// group (q_mu held).
int mv_internal_ensure_batch_select_ws(mv_index* ix) {
  const size_t lists = (size_t)kFdeBatchMaxQueries * kTopkMaxDeviceK;
  if (!ix->d_btopk_ws) {
    hipError_t e = hipMalloc(&ix->d_btopk_ws, (size_t)kFdeBatchMaxQueries * ix->topk_ws_bytes);
    if (e != hipSuccess) { ix->d_btopk_ws = nullptr; set_error("hipMalloc of the batched selection workspace (%zu B) failed", (size_t)kFdeBatchMaxQueries * ix->topk_ws_bytes); return MV_ERR_NOMEM; }
    MV_HIP(hipMemset(ix->d_btopk_ws, 0, (size_t)kFdeBatchMaxQueries * ix->topk_ws_bytes));  // the radix histograms are kept zero between selections
  }
  if (!ix->d_bout_s) MV_HIP(hipMalloc(&ix->d_bout_s, lists * 4));
  if (!ix->d_bout_id) MV_HIP(hipMalloc(&ix->d_bout_id, lists * 8));
  if (!ix->h_bout_s) MV_HIP(hipHostMalloc((void**)&ix->h_bout_s, lists * 4, hipHostMallocDefault));
  if (!ix->h_bout_id) MV_HIP(hipHostMalloc((void**)&ix->h_bout_id, lists * 8, hipHostMallocDefault));
  return MV_OK;
}

// Workspace of the batched FDE pipeline (q_mu held).
int mv_internal_ensure_fde_batch_ws(mv_index* ix) {
  if (ix->h_bcand) return MV_OK;
  int rc0 = mv_internal_ensure_batch_select_ws(ix);
  if (rc0) return rc0;
  const int64_t out_dim = ix->fde_t.out_dim;
  const size_t lists = (size_t)kFdeBatchMaxQueries * kTopkMaxDeviceK;
  if (!ix->d_bq) MV_HIP(hipMalloc(&ix->d_bq, (size_t)kBatchQRows * kRowBytes));
  if (!ix->d_bscores) {
    hipError_t e = hipMalloc(&ix->d_bscores, (size_t)32 * ix->bscore_stride * 4);
    if (e != hipSuccess) { ix->d_bscores = nullptr; set_error("hipMalloc of the batched score vectors (%lld B) failed", (long long)32 * ix->bscore_stride * 4); return MV_ERR_NOMEM; }
  }
  if (!ix->d_bqf32) MV_HIP(hipMalloc(&ix->d_bqf32, (size_t)kBatchQRows * kDim * 4));
  if (!ix->d_bqfde) MV_HIP(hipMalloc(&ix->d_bqfde, (size_t)kFdeBatchMaxQueries * out_dim * 4));
  if (!ix->d_bqimage) MV_HIP(hipMalloc(&ix->d_bqimage, mv::fde_scan_batch_image_bytes(out_dim)));
  if ((ix->fde8 || ix->fde4) && !ix->d_bqfac) MV_HIP(hipMalloc(&ix->d_bqfac, kFdeBatchMaxQueries * 4));
  if (ix->cfg.flags & MV_WITH_FP8) {
    if (!ix->d_bq8hi) MV_HIP(hipMalloc(&ix->d_bq8hi, (size_t)kBatchQRows * kDim));
    if (!ix->d_bq8lo) MV_HIP(hipMalloc(&ix->d_bq8lo, (size_t)kBatchQRows * kDim));
    if (!ix->d_bq8fac) MV_HIP(hipMalloc(&ix->d_bq8fac, (size_t)kBatchQRows * 4));
  }
  if (!ix->d_bsel_s) MV_HIP(hipMalloc(&ix->d_bsel_s, lists * 4));
  if (!ix->d_bsel_id) MV_HIP(hipMalloc(&ix->d_bsel_id, lists * 8));
  if (!ix->d_bcand) MV_HIP(hipMalloc(&ix->d_bcand, lists * 4));
  if (!ix->d_bcand_pads) MV_HIP(hipMalloc(&ix->d_bcand_pads, lists * 4));
  if (!ix->d_bcand_scores) MV_HIP(hipMalloc(&ix->d_bcand_scores, lists * 4));
  MV_HIP(hipHostMalloc((void**)&ix->h_bcand, lists * 4, hipHostMallocDefault));  // last: marks the workspace complete
  return MV_OK;
}

// Workspace of the batched e4m3 scan and its two-tier rerank (q_mu held).
int mv_internal_ensure_fp8_batch_ws(mv_index* ix) {
  int rc = mv_internal_ensure_batch_select_ws(ix);
  if (rc) return rc;
  if (!ix->d_bscores) {
    hipError_t e = hipMalloc(&ix->d_bscores, (size_t)32 * ix->bscore_stride * 4);
    if (e != hipSuccess) { ix->d_bscores = nullptr; set_error("hipMalloc of the batched score vectors (%lld B) failed", (long long)32 * ix->bscore_stride * 4); return MV_ERR_NOMEM; }
  }
  const size_t lists = (size_t)kFdeBatchMaxQueries * kTopkMaxDeviceK;
  if (!ix->d_bqf32) MV_HIP(hipMalloc(&ix->d_bqf32, (size_t)kBatchQRows * kDim * 4));
  if (!ix->d_bq8hi) MV_HIP(hipMalloc(&ix->d_bq8hi, (size_t)kBatchQRows * kDim));
  if (!ix->d_bq8lo) MV_HIP(hipMalloc(&ix->d_bq8lo, (size_t)kBatchQRows * kDim));
  if (!ix->d_bq8fac) MV_HIP(hipMalloc(&ix->d_bq8fac, (size_t)kBatchQRows * 4));
  if (!ix->d_bq) MV_HIP(hipMalloc(&ix->d_bq, (size_t)kBatchQRows * kRowBytes));
  if (!ix->d_bsel_s) MV_HIP(hipMalloc(&ix->d_bsel_s, lists * 4));
  if (!ix->d_bsel_id) MV_HIP(hipMalloc(&ix->d_bsel_id, lists * 8));
  if (!ix->d_bcand) MV_HIP(hipMalloc(&ix->d_bcand, lists * 4));
  if (!ix->d_bcand_pads) MV_HIP(hipMalloc(&ix->d_bcand_pads, lists * 4));
  if (!ix->d_bcand_scores) MV_HIP(hipMalloc(&ix->d_bcand_scores, lists * 4));
  return MV_OK;
}

// Upload a group of nb queries (host buffer, every query n_q_rows rows) into the batch workspace, each padded to rpq rows
// with zero rows (a zero row adds exactly 0 to the FDE and to MaxSim): fp32 rows for the FDE encode (d_bqf32), bf16 rows
// for the bf16 rerank (d_bq), the two-term e4m3 split for the fp8 rerank (d_bq8*).  q_mu held; workspace allocated.
int mv_internal_batch_upload_queries(mv_index* ix, const void* q, int q_dtype, int nb, int n_q_rows, bool want_f32, bool want_bf16, bool want_fp8) {
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  std::vector<float> hf((size_t)nb * rpq * kDim, 0.0f);
  std::vector<uint16_t> hb(want_bf16 ? (size_t)kBatchQRows * kDim : 0, (uint16_t)0);
  std::vector<uint16_t> hbl(want_bf16 ? (size_t)kBatchQRows * kDim : 0, (uint16_t)0);  // lo halves of fp32 queries (upload_query)
  bool any_lo = false;
  for (int b = 0; b < nb; ++b) {
    const char* src = (const char*)q + (size_t)b * n_q_rows * kDim * esz;
    float* df = hf.data() + (size_t)b * rpq * kDim;
    const size_t ne = (size_t)n_q_rows * kDim;
    if (q_dtype == MV_F32) {
      memcpy(df, src, ne * 4);
      if (want_bf16) {
        uint16_t* db = hb.data() + (size_t)b * rpq * kDim;
        uint16_t* dl = hbl.data() + (size_t)b * rpq * kDim;
        for (size_t i = 0; i < ne; ++i) {
          db[i] = host_f32_to_bf16(df[i]);
          dl[i] = host_f32_to_bf16(df[i] - host_bf16_to_f32(db[i]));
          any_lo = any_lo || (dl[i] & 0x7fffu) != 0;
        }
      }
    } else {
      const uint16_t* sb = (const uint16_t*)src;
      if (want_bf16) memcpy(hb.data() + (size_t)b * rpq * kDim, sb, ne * 2);
      for (size_t i = 0; i < ne; ++i) df[i] = host_bf16_to_f32(sb[i]);
    }
  }
  if (want_f32 || want_fp8) MV_HIP(hipMemcpyAsync(ix->d_bqf32, hf.data(), (size_t)nb * rpq * kDim * 4, hipMemcpyHostToDevice, ix->stream));
  if (want_bf16) {
    MV_HIP(hipMemcpyAsync(ix->d_bq, hb.data(), hb.size() * 2, hipMemcpyHostToDevice, ix->stream));
    ix->bq_has_lo = any_lo;
    ix->bq_lo_valid = any_lo || ix->slab_lo != nullptr;
    if (ix->bq_lo_valid) {
      if (!ix->d_bqlo) MV_HIP(hipMalloc(&ix->d_bqlo, (size_t)kBatchQRows * kRowBytes));
      MV_HIP(hipMemcpyAsync(ix->d_bqlo, hbl.data(), hbl.size() * 2, hipMemcpyHostToDevice, ix->stream));
    }
  }
  if (want_fp8) {
    int rc = launch_fp8_query_prep(ix->d_bqf32, nb * rpq, ix->d_bq8hi, ix->d_bq8lo, ix->d_bq8fac, ix->stream);
    if (rc) return rc;
  }
  MV_HIP(hipStreamSynchronize(ix->stream));  // the pageable staging vectors die with this call
  return MV_OK;
}

// Exact rerank of the group's candidate lists d_bcand / d_bcand_pads ([nb][nc], list b against query b of the uploaded
// group) into d_bcand_scores: ONE launch for all lists where the kernels allow it (fp8 slab; bf16 slab with the default
// kernels and queries of <= 128 rows), else one launch per query.  q_mu held.
int mv_internal_batch_rerank_lists(mv_index* ix, int nb, int n_q_rows, int64_t nc, int* launches, int tier, float* d_out) {
  // tier: the bf16 slab in HBM, the exact host tier (pinned host memory, possibly split with HBM), or the e4m3 slab
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const int64_t L = nc;
  const bool rerank_fp8 = tier == kTierFp8;
  const int64_t cap = ix->cfg.capacity_pages;
  const bool split = tier == kTierHost && ix->x_split > 0 && ix->x_split < cap;
  const uint16_t* exact = tier == kTierSlab ? ix->slab : (ix->x_split >= cap ? ix->slab_x : ix->d_exact);
  float* dst = d_out ? d_out : ix->d_bcand_scores;
  const int rr_variant = ix->maxsim_variant < 0 ? maxsim_default_variant(ix->cfg.stride_rows) : ix->maxsim_variant;
  // split-bf16 operands: the pages' lo slab beside the bf16 slab in HBM, and / or fp32 queries whose lo halves sit in d_bqlo
  const uint16_t* slab_lo = tier == kTierSlab ? ix->slab_lo : nullptr;
  const uint16_t* qlo = (ix->bq_lo_valid && (slab_lo || ix->bq_has_lo)) ? ix->d_bqlo : nullptr;
  if (slab_lo && !qlo) { set_error("batched rerank: the lo slab needs the group's lo query rows (mv_internal_batch_upload_queries with want_bf16)"); return MV_ERR_STATE; }
  const bool rerank_one_launch = rerank_fp8 || (rpq <= (qlo ? kMaxQRowsPerPass / 2 : kMaxQRowsPerPass) && (rr_variant == 6 || rr_variant == 7));
  int rc = MV_OK;
  if (rerank_fp8) {
    if (!(ix->cfg.flags & MV_WITH_FP8) || rpq > 64) { set_error("batched e4m3 rerank needs an fp8 slab and queries of <= 64 rows"); return MV_ERR_STATE; }
    Fp8ScanArgs fa{};
    fa.slab = ix->slab8; fa.inv_scale = ix->inv_scale8; fa.n_rows = ix->ragged.load() ? ix->d_n_rows : nullptr; fa.cand = ix->d_bcand;
    fa.qhi = ix->d_bq8hi; fa.qlo = ix->d_bq8lo; fa.qfac = ix->d_bq8fac; fa.n_q = rpq; fa.scores = dst; fa.n = (int64_t)nb * nc;
    fa.stride = ix->cfg.stride_rows; fa.pad_to = 0; fa.pad_items = ix->d_bcand_pads; fa.items_per_query = (int32_t)nc; fa.row_off = ix->d_row_off;
    rc = launch_maxsim_fp8(fa, ix->stream);
    if (rc) return rc;
    ++*launches;
  } else if (rerank_one_launch) {
    const int64_t n = (int64_t)nb * nc;
    MaxsimArgs ma{};
    ma.slab = exact; ma.n_rows = ix->ragged.load() ? ix->d_n_rows : nullptr; ma.cand = ix->d_bcand; ma.q = ix->d_bq;
    ma.scores = dst; ma.n = n; ma.stride = ix->cfg.stride_rows; ma.q_tiles = rpq / 16; ma.pad_to = 0;
    ma.pad_items = ix->d_bcand_pads; ma.items_per_query = (int32_t)nc; ma.q_item_stride = rpq * kDim;
    ma.qlo = qlo; ma.slab_lo = slab_lo; ma.row_off = ix->d_row_off;
    if (!split) {
      rc = launch_maxsim_bf16(ma, rr_variant, ix->stream);
      if (rc) return rc;
      ++*launches;
    } else {  // the lists of all queries split by tier part: one launch per part, merged
      if (n > kMaxCand) { set_error("batched rerank: %lld list entries exceed %d", (long long)n, kMaxCand); return MV_ERR_INVALID; }
      if ((rc = ensure_split_ws(ix)) != MV_OK) return rc;
      int32_t* lo = ix->d_xcand;
      int32_t* hi = ix->d_xcand + kMaxCand;
      const unsigned gb = (unsigned)((n + 255) / 256);
      hipLaunchKernelGGL(split_cand_kernel, dim3(gb), dim3(256), 0, ix->stream, (const int32_t*)ix->d_bcand, n, (int32_t)ix->x_split, lo, hi, (const int32_t*)ix->d_xloc, ix->d_xhits);
      if (ix->d_xoff) { ma.row_off = ix->d_xoff; ma.n_rows = ix->d_n_rows; }  // a rebalanced tier: page -> slot * stride rows (serves both parts, see exact_scan); the table-driven kernels read n_rows unconditionally
      ma.slab = ix->slab_x; ma.cand = lo;
      if ((rc = launch_maxsim_bf16(ma, rr_variant, ix->stream)) != MV_OK) return rc;
      ma.slab = xt_host_vbase(ix); ma.cand = hi; ma.scores = ix->d_xscores;
      if ((rc = launch_maxsim_bf16(ma, rr_variant, ix->stream)) != MV_OK) return rc;
      hipLaunchKernelGGL(max_scores_kernel, dim3(gb), dim3(256), 0, ix->stream, dst, (const float*)ix->d_xscores, n);
      MV_HIP(hipGetLastError());
      *launches += 2;
    }
  } else {
    for (int b = 0; b < nb; ++b) {  // one launch per query (long queries, non-default kernel variants)
      rc = exact_scan(ix, n_q_rows, tier, ix->d_bcand + (size_t)b * L, nc, 0, ix->d_bcand_pads + (size_t)b * L, dst + (size_t)b * L, launches,
                      ix->d_bq + (size_t)b * rpq * kDim, qlo ? qlo + (size_t)b * rpq * kDim : nullptr);
      if (rc) return rc;
    }
  }
  return MV_OK;
}

// mv_query_topk_batch in the FDE modes: per group of <= 32 queries (<= 1024 query rows)
//   encode (one launch, a block per repetition and query) -> ONE pass over the FDE slab for all of them (bf16 MFMA)
//   -> one batched selection of the coarse top-n -> rerank lists + per-batch pad lengths -> exact MaxSim of every
//   query's candidates against ITS query -> one batched final selection -> one read-back.
// The per-query stages are the kernels of the single-query pipeline (same query FDE, same candidate rule, same rerank
// arithmetic); only the coarse dot products differ, by the bf16 hi+lo split of the query FDE (~1e-5 relative).
static int fde_batch_query(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k, int mode,
                           const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores,
                           int64_t* out_ids, int32_t* out_n, mv_query_stats* stats) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  for (int32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
  mv_query_stats total{};
  const int64_t n = ix->size.load(std::memory_order_acquire);  // snapshot of the published corpus
  if (n == 0) { if (stats) *stats = total; return MV_OK; }
  int rc = mv_internal_ensure_fde_batch_ws(ix);
  if (rc) return rc;
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const int group = std::min(kBatchQRows / rpq, kFdeBatchMaxQueries);
  const bool rerank = mode == MV_MODE_FDE_THEN_FLOAT;
  const bool per_query = allow_bits && allow_per_query;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, per_query ? n_allow_words * (int64_t)n_queries : n_allow_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  const int64_t out_dim = ix->fde_t.out_dim;
  const int64_t cap = ix->bscore_stride;  // elements between two requests' score vectors (the capacity + MV_BSCORE_STRIDE_PAD)
  int64_t nc = std::min<int64_t>(std::min<int64_t>(coarse_n_for(ix, k), n), kTopkMaxDeviceK);
  if (nc < 1) nc = 1;
  const int64_t L = nc;  // the per-query lists lie back to back: [query][nc]
  // the rerank tier and the e4m3 pruning stage: the rule of the single-query pipeline (rerank_plan)
  const RerankPlan plan = rerank ? rerank_plan(ix, mode, nc, k, rpq, true) : RerankPlan{};
  const bool rerank_fp8 = rerank && plan.final_fp8;
  const int pad_sem = ix->pad_semantics < 0 ? 1 : ix->pad_semantics;
  int64_t pages = 0;
  if (stats) (void)count_allowed_rows(ix, n, per_query ? nullptr : allow_bits, n_allow_words, &pages);
  for (int32_t b0 = 0; b0 < n_queries; b0 += group) {
    const int nb = std::min(group, n_queries - b0);
    rc = mv_internal_batch_upload_queries(ix, (const char*)q + (size_t)b0 * n_q_rows * kDim * esz, q_dtype, nb, n_q_rows, /*want_f32=*/true,
                                          /*want_bf16=*/rerank && !rerank_fp8, /*want_fp8=*/rerank_fp8 || plan.mid);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
    int launches = 0;
    FdeEncodeArgs e{};
    e.variant = 2;  // the latency kernel of the single-query path, one grid row per query
    e.x_f32 = ix->d_bqf32; e.row_offsets = nullptr; e.stride = rpq; e.n_pages = nb; e.is_query = 1; e.out_f32 = ix->d_bqfde;
    if (const float* ov = query_fde_override(out_dim, b0))  // mv_query_topk_batch_fde: the caller's own encodings of this group's queries
      MV_HIP(hipMemcpyAsync(ix->d_bqfde, ov, (size_t)nb * out_dim * 4, hipMemcpyHostToDevice, ix->stream));
    else
      rc = launch_fde_encode(ix->fde_t, e, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev_st[0], ix->stream));
    FdeScanBatchArgs sa{};
    sa.fde = ix->fde; sa.inv_norm = ix->fde_cosine ? ix->fde_inv_norm : nullptr; sa.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    sa.allow = per_query ? d_allow + (size_t)b0 * n_allow_words : d_allow; sa.n_allow_bits = n_allow_words * 32;
    sa.allow_stride_bits = per_query ? n_allow_words * 32 : 0;
    sa.q = ix->d_bqfde; sa.image = ix->d_bqimage; sa.scores = ix->d_bscores; sa.score_stride = cap; sa.n = n; sa.out_dim = out_dim; sa.n_queries = nb;
    sa.hi_only = ix->fde_batch_variant == 2;
    sa.single_tile = ix->fde_batch_variant == 3;
    sa.separate_finish = ix->fde_batch_variant == 5;
    mv_internal_fde_batch_e4m3_args(ix, &sa);
    // Default: the scan kernel applies the cosine rule / tombstones itself (no finish pass) and the selection runs its three
    // vectorised passes.  Where a finish pass runs anyway (variants 3 / 4, or a dot-product index with masks) it also bins every
    // request's scores for the selection's first radix pass; variant 5 is the round-2 pipeline (separate finish, three passes).
    const int32_t k_sel = rerank ? (int32_t)nc : k;
    if (ix->fde_batch_variant != 5 && topk_uses_radix(n, k_sel)) {
      sa.hist0 = topk_radix_hist0(ix->d_btopk_ws);
      sa.hist0_stride_bytes = (int64_t)ix->topk_ws_bytes;
    }
    const bool prebinned = fde_scan_batch_prebins(sa);
    rc = launch_fde_scan_batch(sa, ix->stream);
    if (rc) return rc;
    launches += 3;
    MV_HIP(hipEventRecord(ix->ev_st[1], ix->stream));
    if (rerank) {
      rc = launch_topk_batch(ix->d_bscores, cap, n, (int32_t)nc, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s, ix->d_bsel_id, L, nb, ix->stream,
                             prebinned);
      if (rc) return rc;
      hipLaunchKernelGGL(cand_prepare_kernel, dim3((unsigned)((nc + kRerankBatch - 1) / kRerankBatch), (unsigned)nb), dim3(kRerankBatch), 0, ix->stream,
                         (const int64_t*)ix->d_bsel_id, (const int32_t*)nullptr, (int)nc, (const int32_t*)ix->d_n_rows, ix->cfg.stride_rows, pad_sem,
                         ix->d_bcand, ix->d_bcand_pads, L);
      MV_HIP(hipGetLastError());
      MV_HIP(hipEventRecord(ix->ev_st[2], ix->stream));
      if (stats) MV_HIP(hipMemcpyAsync(ix->h_bcand, ix->d_bcand, (size_t)nb * L * 4, hipMemcpyDeviceToHost, ix->stream));  // the lists as selected (accounting)
      if (plan.mid) {  // e4m3 scores of every list -> each list's n_mid best positions -> the rest of the list becomes -1
        rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, nc, &launches, kTierFp8, nullptr);
        if (rc) return rc;
        rc = launch_topk_batch(ix->d_bcand_scores, L, nc, plan.n_mid, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s, ix->d_bsel_id, L, nb, ix->stream);
        if (rc) return rc;
        rc = launch_keep_selected(ix->d_bsel_id, L, plan.n_mid, ix->d_bcand, L, (int)nc, nb, ix->stream);
        if (rc) return rc;
      }
      rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, nc, &launches, plan.tier, nullptr);
      if (rc) return rc;
      MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
      rc = launch_topk_batch(ix->d_bcand_scores, L, nc, k, ix->d_bcand, L, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s, ix->d_bout_id, k,
                             nb, ix->stream);
      if (rc) return rc;
    } else {
      MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
      rc = launch_topk_batch(ix->d_bscores, cap, n, k, nullptr, 0, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s, ix->d_bout_id, k, nb,
                             ix->stream, prebinned);
      if (rc) return rc;
    }
    MV_HIP(hipEventRecord(ix->ev[2], ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_s, ix->d_bout_s, (size_t)nb * k * 4, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_id, ix->d_bout_id, (size_t)nb * k * 8, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));
    for (int b = 0; b < nb; ++b) {
      const float* hs = ix->h_bout_s + (size_t)b * k;
      const int64_t* hi = ix->h_bout_id + (size_t)b * k;
      int32_t m = 0;
      while (m < k && hi[m] >= 0) ++m;
      memcpy(out_scores + (size_t)(b0 + b) * k, hs, (size_t)m * 4);
      memcpy(out_ids + (size_t)(b0 + b) * k, hi, (size_t)m * 8);
      out_n[b0 + b] = m;
    }
    if (stats) {
      float enc = 0, coarse = 0, sel = 0, rr = 0, fin = 0, span = 0;
      MV_HIP(hipEventElapsedTime(&enc, ix->ev[0], ix->ev_st[0]));
      MV_HIP(hipEventElapsedTime(&coarse, ix->ev_st[0], ix->ev_st[1]));
      if (rerank) {
        MV_HIP(hipEventElapsedTime(&sel, ix->ev_st[1], ix->ev_st[2]));
        MV_HIP(hipEventElapsedTime(&rr, ix->ev_st[2], ix->ev[1]));
      }
      MV_HIP(hipEventElapsedTime(&fin, ix->ev[1], ix->ev[2]));
      MV_HIP(hipEventElapsedTime(&span, ix->ev[0], ix->ev[2]));
      total.encode_ms += enc; total.coarse_ms += coarse; total.select_ms += sel; total.rerank_ms += rr;
      total.score_kernel_ms += span - fin; total.topk_ms += fin; total.total_device_ms += span;
      total.score_launches += launches; total.pages_scored += pages * nb;
      total.bytes_scanned += pages * out_dim * 2;  // ONE pass over the FDE slab for the whole group
      if (rerank) {
        int64_t cand_rows = 0;
        for (int b = 0; b < nb; ++b)
          for (int64_t i = 0; i < nc; ++i) {
            const int32_t c = ix->h_bcand[(size_t)b * L + i];
            if (c >= 0) cand_rows += ix->h_n_rows[c];
          }
        // with the pruning stage: all candidates on the e4m3 slab, then n_mid of them (at most stride rows each) on the exact tier
        total.bytes_scanned += plan.mid ? cand_rows * (int64_t)kDim + (int64_t)nb * std::min<int64_t>(plan.n_mid, nc) * ix->cfg.stride_rows * kRowBytes
                                        : cand_rows * (int64_t)(rerank_fp8 ? kDim : kRowBytes);
      }
    }
  }
  if (stats) *stats = total;
  return MV_OK;
}

// mv_query_topk_batch, MV_MODE_FLOAT_FP8: groups of <= 512 query rows, ONE pass over the e4m3 slab per group
// (maxsim_batch_fp8_kernel), the group's selections in one chain of launches, one read-back.
// two_tier (MV_MODE_FP8_THEN_FLOAT): every request's fp8 top-n (MV_OPT_RERANK_N) is re-scored exactly from the exact tier, all
// lists in ONE rerank launch, before the final selection -- the exact scan's answers for a batch at the fp8 slab's cost.
static int fp8_batch_query(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k, bool two_tier,
                           const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores, int64_t* out_ids,
                           int32_t* out_n, mv_query_stats* stats) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  for (int32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
  mv_query_stats total{};
  const int64_t n = ix->size.load(std::memory_order_acquire);
  if (n == 0) { if (stats) *stats = total; return MV_OK; }
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const int group = std::min(512 / rpq, 32);
  int rc = mv_internal_ensure_fp8_batch_ws(ix);
  if (rc) return rc;
  int exact_tier = kTierSlab;
  int64_t nc = 0;
  if (two_tier) {
    exact_tier = rerank_plan(ix, MV_MODE_FP8_THEN_FLOAT, 0, k, rpq, true).tier;
    nc = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(std::max<int64_t>(ix->rerank_n, k), n), kTopkMaxDeviceK));
  }
  const bool per_query = allow_bits && allow_per_query;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, per_query ? n_allow_words * (int64_t)n_queries : n_allow_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  int64_t pages = 0;
  const int64_t rows = stats ? count_allowed_rows(ix, n, per_query ? nullptr : allow_bits, n_allow_words, &pages) : 0;
  for (int32_t b0 = 0; b0 < n_queries; b0 += group) {
    const int nb = std::min(group, n_queries - b0);
    // zero rows behind the last query of the group: the kernel's row tiles run to the next multiple of 64 rows
    MV_HIP(hipMemsetAsync(ix->d_bqf32, 0, (size_t)512 * kDim * 4, ix->stream));
    rc = mv_internal_batch_upload_queries(ix, (const char*)q + (size_t)b0 * n_q_rows * kDim * esz, q_dtype, nb, n_q_rows, true, two_tier, false);
    if (rc) return rc;
    rc = launch_fp8_query_prep(ix->d_bqf32, 512, ix->d_bq8hi, ix->d_bq8lo, ix->d_bq8fac, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
    Fp8BatchArgs a{};
    a.slab = ix->slab8; a.inv_scale = ix->inv_scale8; a.n_rows = ix->ragged.load() ? ix->d_n_rows : nullptr; a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    a.allow = per_query ? d_allow + (size_t)b0 * n_allow_words : d_allow; a.n_allow_bits = n_allow_words * 32;
    a.allow_stride_bits = per_query ? n_allow_words * 32 : 0;
    a.qhi = ix->d_bq8hi; a.qlo = ix->d_bq8lo; a.qfac = ix->d_bq8fac; a.scores = ix->d_bscores; a.n = n; a.score_stride = ix->bscore_stride;
    a.stride = ix->cfg.stride_rows; a.n_queries = nb; a.rows_per_query = rpq; a.row_off = ix->d_row_off;
    // two-tier search: the first stage only NOMINATES candidates and the exact tier restores their order, so by default it runs with ONE
    // e4m3 term per query row (half the matrix work: 9.9 vs 16.3 ms per 16 requests at 200 k pages; recall@10 1.0 on every corpus of
    // bench.py either way); MV_OPT_BATCH_VARIANT 0 asks for the two-term scores (the single-query scan's), 7 for one term in either mode
    a.single_term = two_tier ? (ix->batch_variant == 0 ? 0 : 1) : (ix->batch_variant == 7 ? 1 : 0);
    rc = launch_maxsim_batch_fp8(a, ix->stream);
    if (rc) return rc;
    if (two_tier) {
      // fp8 top-n of every request (local page ids) -> rerank lists -> exact bf16 scores from the exact tier, one launch
      rc = launch_topk_batch(ix->d_bscores, ix->bscore_stride, n, (int32_t)nc, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s,
                             ix->d_bsel_id, nc, nb, ix->stream);
      if (rc) return rc;
      hipLaunchKernelGGL(cand_prepare_kernel, dim3((unsigned)((nc + kRerankBatch - 1) / kRerankBatch), (unsigned)nb), dim3(kRerankBatch), 0, ix->stream,
                         (const int64_t*)ix->d_bsel_id, (const int32_t*)nullptr, (int)nc, (const int32_t*)ix->d_n_rows, ix->cfg.stride_rows, /*pad_sem=*/0,
                         ix->d_bcand, ix->d_bcand_pads, nc);
      MV_HIP(hipGetLastError());
      int launches = 0;
      rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, nc, &launches, exact_tier, nullptr);
      if (rc) return rc;
      MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
      rc = launch_topk_batch(ix->d_bcand_scores, nc, nc, k, ix->d_bcand, nc, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s, ix->d_bout_id,
                             k, nb, ix->stream);
      if (rc) return rc;
    } else {
      MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
      rc = launch_topk_batch(ix->d_bscores, ix->bscore_stride, n, k, nullptr, 0, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s,
                             ix->d_bout_id, k, nb, ix->stream);
      if (rc) return rc;
    }
    MV_HIP(hipEventRecord(ix->ev[2], ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_s, ix->d_bout_s, (size_t)nb * k * 4, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_id, ix->d_bout_id, (size_t)nb * k * 8, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));
    for (int b = 0; b < nb; ++b) {
      const float* hs = ix->h_bout_s + (size_t)b * k;
      const int64_t* hi = ix->h_bout_id + (size_t)b * k;
      int32_t m = 0;
      while (m < k && hi[m] >= 0) ++m;
      memcpy(out_scores + (size_t)(b0 + b) * k, hs, (size_t)m * 4);
      memcpy(out_ids + (size_t)(b0 + b) * k, hi, (size_t)m * 8);
      out_n[b0 + b] = m;
    }
    if (stats) {
      float ms_scan = 0, ms_sel = 0;
      MV_HIP(hipEventElapsedTime(&ms_scan, ix->ev[0], ix->ev[1]));
      MV_HIP(hipEventElapsedTime(&ms_sel, ix->ev[1], ix->ev[2]));
      total.score_kernel_ms += ms_scan; total.topk_ms += ms_sel; total.total_device_ms += ms_scan + ms_sel;
      total.score_launches += 1; total.pages_scored += pages * nb; total.bytes_scanned += rows * (int64_t)kDim;
    }
  }
  if (stats) *stats = total;
  return MV_OK;
}

// mv_query_topk_batch(MV_MODE_FLOAT) on an index with a lo slab in cascade mode (MV_OPT_FLOAT_LO_SCAN 2): the single query's rule --
// hi-only scan of every page -> the best max(MV_OPT_RERANK_N, k) -> split-bf16 re-score of those (fp32-faithful scores) -> top-k --
// for a GROUP of requests per slab pass: the batched bf16 MFMA scan (queries' hi rows), one batched selection, the one-launch rerank
// of all lists on the hi + lo slabs (mv_internal_batch_rerank_lists), one batched final selection, one read-back.  (Round 6: such
// batches ran query by query -- B passes over the slab.)
static int float_cascade_batch_query(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k,
                                     const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores, int64_t* out_ids,
                                     int32_t* out_n, mv_query_stats* stats) {
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  for (int32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
  mv_query_stats total{};
  const int64_t n = ix->size.load(std::memory_order_acquire);
  if (n == 0) { if (stats) *stats = total; return MV_OK; }
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  const int group = std::min(512 / rpq, 32);
  int rc = mv_internal_ensure_batch_select_ws(ix);
  if (rc) return rc;
  const size_t lists = (size_t)kFdeBatchMaxQueries * kTopkMaxDeviceK;
  if (!ix->d_bscores) {
    hipError_t e = hipMalloc(&ix->d_bscores, (size_t)32 * ix->bscore_stride * 4);
    if (e != hipSuccess) { ix->d_bscores = nullptr; set_error("hipMalloc of the batched score vectors (%lld B) failed", (long long)32 * ix->bscore_stride * 4); return MV_ERR_NOMEM; }
  }
  if (!ix->d_bq) MV_HIP(hipMalloc(&ix->d_bq, (size_t)kBatchQRows * kRowBytes));
  if (!ix->d_bsel_s) MV_HIP(hipMalloc(&ix->d_bsel_s, lists * 4));
  if (!ix->d_bsel_id) MV_HIP(hipMalloc(&ix->d_bsel_id, lists * 8));
  if (!ix->d_bcand) MV_HIP(hipMalloc(&ix->d_bcand, lists * 4));
  if (!ix->d_bcand_pads) MV_HIP(hipMalloc(&ix->d_bcand_pads, lists * 4));
  if (!ix->d_bcand_scores) MV_HIP(hipMalloc(&ix->d_bcand_scores, lists * 4));
  const int64_t nc = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(std::max<int64_t>(ix->rerank_n, k), n), kTopkMaxDeviceK));
  const bool per_query = allow_bits && allow_per_query;
  const uint32_t* d_allow = nullptr;
  rc = upload_allow(ix, allow_bits, per_query ? n_allow_words * (int64_t)n_queries : n_allow_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  const bool ragged = ix->ragged.load();
  int64_t pages = 0;
  const int64_t rows = stats ? count_allowed_rows(ix, n, per_query ? nullptr : allow_bits, n_allow_words, &pages) : 0;
  for (int32_t b0 = 0; b0 < n_queries; b0 += group) {
    const int nb = std::min(group, n_queries - b0);
    rc = mv_internal_batch_upload_queries(ix, (const char*)q + (size_t)b0 * n_q_rows * kDim * esz, q_dtype, nb, n_q_rows, false, true, false);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
    BatchArgs a{};
    a.slab = ix->slab; a.n_rows = ragged ? ix->d_n_rows : nullptr; a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    a.allow = per_query ? d_allow + (size_t)b0 * n_allow_words : d_allow; a.n_allow_bits = n_allow_words * 32;
    a.allow_stride_bits = per_query ? n_allow_words * 32 : 0; a.q = ix->d_bq; a.scores = ix->d_bscores; a.n = n;
    a.score_stride = ix->bscore_stride; a.stride = ix->cfg.stride_rows; a.n_queries = nb; a.rows_per_query = rpq; a.row_off = ix->d_row_off;
    a.variant = (ix->batch_variant == 3 || ix->batch_variant == 4) ? ix->batch_variant : 0;
    rc = launch_maxsim_batch(a, ix->stream);
    if (rc) return rc;
    // every request's best nc pages of the hi scan (local page ids) -> rerank lists -> split-bf16 scores from the hi + lo slabs
    rc = launch_topk_batch(ix->d_bscores, ix->bscore_stride, n, (int32_t)nc, nullptr, 0, 0, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bsel_s,
                           ix->d_bsel_id, nc, nb, ix->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(cand_prepare_kernel, dim3((unsigned)((nc + kRerankBatch - 1) / kRerankBatch), (unsigned)nb), dim3(kRerankBatch), 0, ix->stream,
                       (const int64_t*)ix->d_bsel_id, (const int32_t*)nullptr, (int)nc, (const int32_t*)ix->d_n_rows, ix->cfg.stride_rows, /*pad_sem=*/0,
                       ix->d_bcand, ix->d_bcand_pads, nc);
    MV_HIP(hipGetLastError());
    int launches = 0;
    rc = mv_internal_batch_rerank_lists(ix, nb, n_q_rows, nc, &launches, kTierSlab, nullptr);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
    rc = launch_topk_batch(ix->d_bcand_scores, nc, nc, k, ix->d_bcand, nc, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s, ix->d_bout_id,
                           k, nb, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[2], ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_s, ix->d_bout_s, (size_t)nb * k * 4, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_id, ix->d_bout_id, (size_t)nb * k * 8, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));
    for (int b = 0; b < nb; ++b) {
      const float* hs = ix->h_bout_s + (size_t)b * k;
      const int64_t* hi = ix->h_bout_id + (size_t)b * k;
      int32_t m = 0;
      while (m < k && hi[m] >= 0) ++m;
      memcpy(out_scores + (size_t)(b0 + b) * k, hs, (size_t)m * 4);
      memcpy(out_ids + (size_t)(b0 + b) * k, hi, (size_t)m * 8);
      out_n[b0 + b] = m;
    }
    if (stats) {
      float ms_scan = 0, ms_sel = 0;
      MV_HIP(hipEventElapsedTime(&ms_scan, ix->ev[0], ix->ev[1]));
      MV_HIP(hipEventElapsedTime(&ms_sel, ix->ev[1], ix->ev[2]));
      total.score_kernel_ms += ms_scan; total.topk_ms += ms_sel; total.total_device_ms += ms_scan + ms_sel;
      total.score_launches += 1 + launches; total.pages_scored += pages * nb; total.bytes_scanned += rows * (int64_t)kRowBytes;
    }
  }
  if (stats) *stats = total;
  return MV_OK;
}

int mv_query_topk_batch(mv_index* ix, const void* q, int q_dtype, int32_t n_queries, int32_t n_q_rows, int32_t k, int mode,
                        const uint32_t* allow_bits, int64_t n_allow_words, int32_t allow_per_query, float* out_scores,
                        int64_t* out_ids, int32_t* out_n, mv_query_stats* stats) {
  if (!ix || !q || n_queries < 1 || n_q_rows < 1 || k < 0 || !out_n || (k > 0 && (!out_scores || !out_ids))) { set_error("mv_query_topk_batch: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_queries * n_q_rows * kDim, mode)) return frc;
  const size_t esz = q_dtype == MV_F32 ? 4 : 2;
  const int rpq = ((n_q_rows + 15) / 16) * 16;
  mv_query_stats total{};
  // the options that ROUTE a batch are read once, under the lock mv_index_set_option writes them under (a concurrent set_option takes
  // effect for the next batch; TSan run of round 6: the unlocked reads below raced with it)
  int opt_fde_batch_variant, opt_batch_variant, opt_float_lo_scan;
  {
    std::lock_guard<std::mutex> lk(ix->q_mu);
    opt_fde_batch_variant = ix->fde_batch_variant; opt_batch_variant = ix->batch_variant; opt_float_lo_scan = ix->float_lo_scan;
  }
  // FDE modes: the batched pipeline (rerank on the bf16 slab, or on the fp8 slab of an index without one: queries of <= 64 rows)
  if ((mode == MV_MODE_FDE_THEN_FLOAT || mode == MV_MODE_FDE_ONLY) && n_queries > 1 && k >= 1 && k <= kTopkMaxDeviceK && rpq <= 512 &&
      opt_fde_batch_variant != 1 && (ix->cfg.flags & MV_WITH_FDE) &&
      (mode == MV_MODE_FDE_ONLY || (ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT)) || ((ix->cfg.flags & MV_WITH_FP8) && rpq <= 64)) &&
      mv::fde_scan_batch_supported(ix->fde_t.out_dim) && ix->fde_t.cfg.projection_dimension <= 16)
    return fde_batch_query(ix, q, q_dtype, n_queries, n_q_rows, k, mode, allow_bits, n_allow_words, allow_per_query, out_scores, out_ids, out_n, stats);
  // e4m3 slab: the batched block-scaled MFMA scan (<= 512 query rows per slab pass); MV_OPT_BATCH_VARIANT 8 = query by query
  if ((mode == MV_MODE_FLOAT_FP8 || (mode == MV_MODE_FP8_THEN_FLOAT && rpq <= kMaxQRowsPerPass && (ix->cfg.flags & (MV_WITH_FLOAT | MV_WITH_HOST_EXACT)))) &&
      n_queries > 1 && rpq <= 512 && k >= 1 && k <= kTopkMaxDeviceK && (ix->cfg.flags & MV_WITH_FP8) && opt_batch_variant != 8)
    return fp8_batch_query(ix, q, q_dtype, n_queries, n_q_rows, k, mode == MV_MODE_FP8_THEN_FLOAT, allow_bits, n_allow_words, allow_per_query, out_scores,
                           out_ids, out_n, stats);
  // Split-bf16 operands (fp32 queries that are not bf16-representable; an index with a lo slab scanned with it): the batched MFMA
  // kernel multiplies ONE bf16 term per operand -- such batches are served query by query by the single-query kernels, which carry
  // the lo halves (the batch stays one call; bf16 queries on a plain index keep the one-pass form).
  // cascade mode of an index with a lo slab: the batched hi scan nominates, the split-bf16 rerank of every list scores (one slab pass per group)
  if (mode == MV_MODE_FLOAT && ix->slab_lo && opt_float_lo_scan == 2 && n_queries > 1 && k >= 1 && k <= kTopkMaxDeviceK && rpq <= 512 &&
      (ix->cfg.flags & MV_WITH_FLOAT) && opt_batch_variant != 8)
    return float_cascade_batch_query(ix, q, q_dtype, n_queries, n_q_rows, k, allow_bits, n_allow_words, allow_per_query, out_scores, out_ids, out_n, stats);
  bool batch_lo = mode == MV_MODE_FLOAT && ix->slab_lo && opt_float_lo_scan != 0;
  if (mode == MV_MODE_FLOAT && !batch_lo && q_dtype == MV_F32) {
    const float* qf = (const float*)q;
    const size_t ne = (size_t)n_queries * n_q_rows * kDim;
    for (size_t i = 0; i < ne && !batch_lo; ++i) batch_lo = host_bf16_to_f32(host_f32_to_bf16(qf[i])) != qf[i];
  }
  // anything but the exact float scan (and queries longer than one 512-row group) runs query by query
  if (mode != MV_MODE_FLOAT || rpq > 512 || k > kTopkMaxDeviceK || k == 0 || batch_lo) {
    for (int32_t b = 0; b < n_queries; ++b) {
      mv_query_stats st{};
      const uint32_t* ab = (allow_bits && allow_per_query) ? allow_bits + (size_t)b * n_allow_words : allow_bits;
      QueryFdeCursor cur(g_qfde.cur + b);
      int rc = mv_internal_query_common(ix, (const char*)q + (size_t)b * n_q_rows * kDim * esz, q_dtype, n_q_rows, k, mode, ab, n_allow_words,
                                        out_scores ? out_scores + (size_t)b * k : nullptr, out_ids ? out_ids + (size_t)b * k : nullptr, out_n + b, nullptr,
                                        nullptr, nullptr, stats ? &st : nullptr, 0);
      if (rc) return rc;
      total.score_kernel_ms += st.score_kernel_ms; total.topk_ms += st.topk_ms; total.total_device_ms += st.total_device_ms;
      total.score_launches += st.score_launches; total.pages_scored += st.pages_scored; total.bytes_scanned += st.bytes_scanned;
    }
    if (stats) *stats = total;
    return MV_OK;
  }
  if (!(ix->cfg.flags & MV_WITH_FLOAT)) { set_error("index has no float slab (MV_WITH_FLOAT)"); return MV_ERR_STATE; }
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  for (int32_t b = 0; b < n_queries; ++b) out_n[b] = 0;
  const int64_t n = ix->size.load(std::memory_order_acquire);  // snapshot of the published corpus
  if (n == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MV_OK; }
  const int group_rows = 512;
  if (rpq > group_rows) { set_error("query of %d rows exceeds the %d-row group of the batched scan", rpq, group_rows); return MV_ERR_INVALID; }
  const int group = std::min(group_rows / rpq, 32);
  if (!ix->d_bq) MV_HIP(hipMalloc(&ix->d_bq, (size_t)kBatchQRows * kRowBytes));
  if (!ix->d_bscores) {
    hipError_t e = hipMalloc(&ix->d_bscores, (size_t)32 * ix->bscore_stride * 4);
    if (e != hipSuccess) { ix->d_bscores = nullptr; set_error("hipMalloc of the batched score vectors (%lld B) failed", (long long)32 * ix->bscore_stride * 4); return MV_ERR_NOMEM; }
  }
  const bool per_query = allow_bits && allow_per_query;
  const uint32_t* d_allow = nullptr;
  // per-query bitmaps: all n_queries x n_allow_words words are uploaded once; a group reads its slice
  int rc = upload_allow(ix, allow_bits, per_query ? n_allow_words * (int64_t)n_queries : n_allow_words, &d_allow);
  if (rc) return rc;
  const bool need_meta = ix->tombstones.load() || d_allow != nullptr;
  const bool ragged = ix->ragged.load();
  int64_t pages = 0;
  // accounting: with per-query filters every live page is read (a page is skipped only when no query may see it)
  const int64_t rows = stats ? count_allowed_rows(ix, n, per_query ? nullptr : allow_bits, n_allow_words, &pages) : 0;
  std::vector<uint16_t> hq((size_t)512 * kDim);
  rc = mv_internal_ensure_batch_select_ws(ix);
  if (rc) return rc;
  for (int32_t b0 = 0; b0 < n_queries; b0 += group) {
    const int nb = std::min(group, n_queries - b0);
    std::fill(hq.begin(), hq.end(), (uint16_t)0);
    for (int b = 0; b < nb; ++b) {
      const char* src = (const char*)q + (size_t)(b0 + b) * n_q_rows * kDim * esz;
      uint16_t* dst = hq.data() + (size_t)b * rpq * kDim;
      if (q_dtype == MV_BF16) memcpy(dst, src, (size_t)n_q_rows * kRowBytes);
      else for (size_t i = 0; i < (size_t)n_q_rows * kDim; ++i) dst[i] = host_f32_to_bf16(((const float*)src)[i]);
    }
    MV_HIP(hipMemcpyAsync(ix->d_bq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice, ix->stream));
    MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
    BatchArgs a{};
    a.slab = ix->slab; a.n_rows = ragged ? ix->d_n_rows : nullptr; a.doc_ord = need_meta ? ix->d_doc_ord : nullptr;
    a.allow = per_query ? d_allow + (size_t)b0 * n_allow_words : d_allow; a.n_allow_bits = n_allow_words * 32;
    a.allow_stride_bits = per_query ? n_allow_words * 32 : 0; a.q = ix->d_bq; a.scores = ix->d_bscores; a.n = n;
    a.score_stride = ix->bscore_stride; a.stride = ix->cfg.stride_rows; a.n_queries = nb; a.rows_per_query = rpq; a.row_off = ix->d_row_off;
    a.variant = ix->batch_variant >= 0 ? ix->batch_variant : 0;  // auto: page-split form up to 128 rows, transposed row-split form above
    rc = launch_maxsim_batch(a, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
    // the group's selections in one chain of launches (grid.y = query), one read-back, one synchronisation
    rc = launch_topk_batch(ix->d_bscores, ix->bscore_stride, n, k, nullptr, 0, ix->cfg.id_base, ix->d_btopk_ws, ix->topk_ws_bytes, ix->d_bout_s,
                           ix->d_bout_id, k, nb, ix->stream);
    if (rc) return rc;
    MV_HIP(hipEventRecord(ix->ev[2], ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_s, ix->d_bout_s, (size_t)nb * k * 4, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipMemcpyAsync(ix->h_bout_id, ix->d_bout_id, (size_t)nb * k * 8, hipMemcpyDeviceToHost, ix->stream));
    MV_HIP(hipStreamSynchronize(ix->stream));
    for (int b = 0; b < nb; ++b) {
      const float* hs = ix->h_bout_s + (size_t)b * k;
      const int64_t* hi = ix->h_bout_id + (size_t)b * k;
      int32_t m = 0;
      while (m < k && hi[m] >= 0) ++m;
      memcpy(out_scores + (size_t)(b0 + b) * k, hs, (size_t)m * 4);
      memcpy(out_ids + (size_t)(b0 + b) * k, hi, (size_t)m * 8);
      out_n[b0 + b] = m;
    }
    if (stats) {
      MV_HIP(hipEventSynchronize(ix->ev[2]));
      float ms_scan = 0, ms_sel = 0;
      MV_HIP(hipEventElapsedTime(&ms_scan, ix->ev[0], ix->ev[1]));
      MV_HIP(hipEventElapsedTime(&ms_sel, ix->ev[1], ix->ev[2]));
      total.score_kernel_ms += ms_scan; total.topk_ms += ms_sel; total.total_device_ms += ms_scan + ms_sel;
      total.score_launches += 1; total.pages_scored += pages * nb; total.bytes_scanned += rows * (int64_t)kRowBytes;
    }
  }
  if (stats) *stats = total;
  return MV_OK;
}

int mv_score_all(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, int mode, const uint32_t* allow_bits,
                 int64_t n_allow_words, float* out_scores, int64_t out_cap, int64_t* out_n, mv_query_stats* stats) {
  if (!ix || !out_scores || out_cap < 0) { set_error("null argument"); return MV_ERR_INVALID; }
  if (mode == MV_MODE_FDE_THEN_FLOAT) mode = MV_MODE_FDE_ONLY;
  if (mode == MV_MODE_FP8_THEN_FLOAT) mode = MV_MODE_FLOAT_FP8;  // the first-stage scores
  std::lock_guard<std::mutex> lk(ix->q_mu);
  DeviceGuard g(ix->cfg.device);
  if (out_n) *out_n = 0;
  if (ix->size.load(std::memory_order_acquire) == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return MV_OK; }
  ScanResult r;
  int rc = run_scan(ix, q, q_dtype, n_q_rows, mode, allow_bits, n_allow_words, 0, &r, stats);
  if (rc) return rc;
  // the corpus may have grown since the caller sized its buffer: never write past out_cap
  const int64_t m = std::min<int64_t>(r.n, out_cap);
  MV_HIP(hipMemcpyAsync(out_scores, r.d_scores, (size_t)m * 4, hipMemcpyDeviceToHost, ix->stream));
  MV_HIP(hipStreamSynchronize(ix->stream));
  if (out_n) *out_n = m;
  return finish_stats(ix, stats, false);
}

// Shared body of mv_score_candidates / mv_score_candidates_pads.  pads: per-candidate pad_to (host) or null;
// pad_to >= 0: one pad length for the whole list, -1: the reference rule (longest page of each batch of 128).
static int score_candidates_common(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const int32_t* cand, int32_t n_cand,
                                   int32_t pad_to, const int32_t* pads, float* out_scores, mv_query_stats* stats) {
  if (!ix || !q || !cand || !out_scores || n_cand < 0 || n_cand > kMaxCand || n_q_rows < 1 || pad_to < -1) { set_error("score_candidates: bad argument"); return MV_ERR_INVALID; }
  if (q_dtype != MV_F32 && q_dtype != MV_BF16) { set_error("bad query dtype %d", q_dtype); return MV_ERR_INVALID; }
  if (int frc = check_query_finite(q, q_dtype, (size_t)n_q_rows * kDim, MV_MODE_FLOAT)) return frc;
  std::lock_guard<std::mutex> lk(ix->q_mu);
  // the named pages are scored on the exact tier (bf16 slab, else the pinned-host tier), else on the e4m3 slab
  const RerankPlan plan = rerank_plan(ix, MV_MODE_FLOAT, n_cand, 0, ((n_q_rows + 15) / 16) * 16, false);
  const bool use_fp8 = plan.final_fp8;
  if (use_fp8 && !(ix->cfg.flags & MV_WITH_FP8)) { set_error("index has neither a float slab, an exact host tier nor an fp8 slab"); return MV_ERR_STATE; }
  DeviceGuard g(ix->cfg.device);
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_cand == 0) return MV_OK;
  const int64_t size = ix->size.load(std::memory_order_acquire);
  int64_t rows = 0;
  for (int i = 0; i < n_cand; ++i) {
    if (cand[i] < 0 || cand[i] >= size) { set_error("candidate %d out of range", cand[i]); return MV_ERR_INVALID; }
    rows += ix->h_n_rows[cand[i]];
  }
  int rc = upload_query(ix, q, q_dtype, n_q_rows, !use_fp8, false, false, use_fp8);
  if (rc) return rc;
  // tombstones are NOT applied here: the caller named the pages explicitly
  MV_HIP(hipEventRecord(ix->ev[0], ix->stream));
  int launches = 0;
  if (pads) {
    MV_HIP(hipMemcpyAsync(ix->d_cand, cand, (size_t)n_cand * 4, hipMemcpyHostToDevice, ix->stream));
    MV_HIP(hipMemcpyAsync(ix->d_cand_pads, pads, (size_t)n_cand * 4, hipMemcpyHostToDevice, ix->stream));
  } else if (pad_to < 0) {
    // the ids are staged in d_cand_scores (same size; the scan overwrites it afterwards, in stream order);
    // cand_prepare copies them to d_cand and derives each batch-of-128's pad length on the device
    MV_HIP(hipMemcpyAsync(ix->d_cand_scores, cand, (size_t)n_cand * 4, hipMemcpyHostToDevice, ix->stream));
    rc = launch_cand_prepare(ix, nullptr, (const int32_t*)ix->d_cand_scores, n_cand, 1);
    if (rc) return rc;
    ++launches;
  } else {
    MV_HIP(hipMemcpyAsync(ix->d_cand, cand, (size_t)n_cand * 4, hipMemcpyHostToDevice, ix->stream));
  }
  const bool per_item = pads || pad_to < 0;
  rc = use_fp8 ? fp8_scan(ix, n_q_rows, nullptr, 0, ix->d_cand, n_cand, per_item ? 0 : pad_to, per_item ? ix->d_cand_pads : nullptr, ix->d_cand_scores, &launches, true)
               : exact_scan(ix, n_q_rows, plan.tier, ix->d_cand, n_cand, per_item ? 0 : pad_to, per_item ? ix->d_cand_pads : nullptr, ix->d_cand_scores, &launches);
  if (rc) return rc;
  MV_HIP(hipEventRecord(ix->ev[1], ix->stream));
  MV_HIP(hipMemcpyAsync(out_scores, ix->d_cand_scores, (size_t)n_cand * 4, hipMemcpyDeviceToHost, ix->stream));
  MV_HIP(hipStreamSynchronize(ix->stream));
  if (stats) {
    stats->score_launches = launches;
    stats->pages_scored = n_cand;
    stats->bytes_scanned = rows * (int64_t)(use_fp8 ? kDim : kRowBytes);
  }
  return finish_stats(ix, stats, false);
}

int mv_score_candidates(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const int32_t* cand, int32_t n_cand,
                        int32_t pad_to, float* out_scores, mv_query_stats* stats) {
  return score_candidates_common(ix, q, q_dtype, n_q_rows, cand, n_cand, pad_to, nullptr, out_scores, stats);
}

int mv_score_candidates_pads(mv_index* ix, const void* q, int q_dtype, int32_t n_q_rows, const int32_t* cand, int32_t n_cand,
                             const int32_t* pads, float* out_scores, mv_query_stats* stats) {
  if (!pads) { set_error("score_candidates_pads: null pads"); return MV_ERR_INVALID; }
  return score_candidates_common(ix, q, q_dtype, n_q_rows, cand, n_cand, 0, pads, out_scores, stats);
}

int mv_index_page_rows(mv_index* ix, const int32_t* pages, int64_t n_pages, int32_t* out_rows) {
  if (!ix || n_pages < 0 || (n_pages > 0 && (!pages || !out_rows))) { set_error("page_rows: bad argument"); return MV_ERR_INVALID; }
  const int64_t size = ix->size.load(std::memory_order_acquire);  // published pages are immutable: no lock needed
  for (int64_t i = 0; i < n_pages; ++i) {
    if (pages[i] < 0 || pages[i] >= size) { set_error("page %d out of range", pages[i]); return MV_ERR_INVALID; }
    out_rows[i] = ix->h_n_rows[pages[i]];
  }
  return MV_OK;
}

int mv_sign_pack(int device, const float* x, int64_t n_rows, int32_t d, uint8_t* out) {
  if (!x || !out || n_rows < 0 || d < 1) { set_error("mv_sign_pack: bad argument"); return MV_ERR_INVALID; }
  if (n_rows == 0) return MV_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available"); return MV_ERR_HIP; }
  DeviceGuard g(device);
  const size_t nb = (size_t)((d + 7) / 8);
  float* dx = nullptr;
  uint8_t* dout = nullptr;
  MV_HIP(hipMalloc(&dx, (size_t)n_rows * d * 4));
  hipError_t e = hipMalloc(&dout, (size_t)n_rows * nb);
  if (e != hipSuccess) { (void)hipFree(dx); return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
  int rc = MV_OK;
  e = hipMemcpy(dx, x, (size_t)n_rows * d * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) rc = hip_fail(e, "H2D", __FILE__, __LINE__);
  if (!rc) rc = launch_sign_pack_f32(dx, n_rows, d, dout, nullptr);
  if (!rc) {
    e = hipMemcpy(out, dout, (size_t)n_rows * nb, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = hip_fail(e, "D2H", __FILE__, __LINE__);
  }
  (void)hipFree(dx);
  (void)hipFree(dout);
  return rc;
}

int mv_hamming_batch(int device, const uint8_t* query, const uint8_t* cands, int64_t n_cands, int32_t n_bytes, int32_t* out) {
  if (!query || (!cands && n_cands) || !out || n_cands < 0 || n_bytes < 1) { set_error("mv_hamming_batch: bad argument"); return MV_ERR_INVALID; }
  if (n_cands == 0) return MV_OK;
  DeviceGuard g(device);
  uint8_t *dq = nullptr, *dc = nullptr;
  int32_t* dout = nullptr;
  MV_HIP(hipMalloc(&dq, (size_t)n_bytes));
  MV_HIP(hipMalloc(&dc, (size_t)n_cands * n_bytes));
  MV_HIP(hipMalloc(&dout, (size_t)n_cands * 4));
  int rc = MV_OK;
  if (hipMemcpy(dq, query, (size_t)n_bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dc, cands, (size_t)n_cands * n_bytes, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D failed"); rc = MV_ERR_HIP; }
  if (!rc) rc = launch_hamming_batch(dq, dc, n_cands, n_bytes, dout, nullptr);
  if (!rc && hipMemcpy(out, dout, (size_t)n_cands * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); rc = MV_ERR_HIP; }
  (void)hipFree(dq); (void)hipFree(dc); (void)hipFree(dout);
  return rc;
}

int64_t mv_fde_output_dim(const mv_fde_config* c) {
  if (!c) return 0;
  return (int64_t)c->num_repetitions * (1LL << c->num_simhash_projections) * c->projection_dimension;
}

int mv_fde_encode(int device, const mv_fde_config* cfg, const float* x, int32_t n_rows, int32_t is_query, float* out) {
  {
    const char* ev = getenv("MV_FDE_SCALAR");
    g_stateless_fde_variant = (ev && ev[0] == '1') ? 0 : 1;
  }
  if (!cfg || !x || !out || n_rows < 0) { set_error("mv_fde_encode: bad argument"); return MV_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device available"); return MV_ERR_HIP; }
  DeviceGuard g(device);
  FdeTables t;
  int rc = fde_tables_create(*cfg, &t);
  if (rc) return rc;
  float *dx = nullptr, *dout = nullptr;
  int64_t* doff = nullptr;
  const int64_t off[2] = {0, n_rows};
  if (hipMalloc(&dx, std::max<size_t>((size_t)n_rows * kDim * 4, 16)) != hipSuccess || hipMalloc(&dout, (size_t)t.out_dim * 4) != hipSuccess ||
      hipMalloc(&doff, 16) != hipSuccess) { set_error("out of device memory"); rc = MV_ERR_NOMEM; }
  if (!rc && (hipMemcpy(dx, x, (size_t)n_rows * kDim * 4, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(doff, off, 16, hipMemcpyHostToDevice) != hipSuccess)) { set_error("H2D failed"); rc = MV_ERR_HIP; }
  if (!rc) {
    FdeEncodeArgs e{};
    e.variant = g_stateless_fde_variant;
    e.x_f32 = dx; e.row_offsets = doff; e.n_pages = 1; e.is_query = is_query; e.out_f32 = dout;
    rc = launch_fde_encode(t, e, nullptr);
  }
  if (!rc && hipMemcpy(out, dout, (size_t)t.out_dim * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed"); rc = MV_ERR_HIP; }
  if (dx) (void)hipFree(dx);
  if (dout) (void)hipFree(dout);
  if (doff) (void)hipFree(doff);
  fde_tables_destroy(&t);
  return rc;
}

int mv_calibrate_read_bw(int device, int64_t bytes, int32_t iters, double* out_gbps) {
  if (!out_gbps || bytes < (1 << 20) || iters < 1) { set_error("calibrate: bad argument"); return MV_ERR_INVALID; }
  DeviceGuard g(device);
  void* buf = nullptr;
  float* sink = nullptr;
  MV_HIP(hipMalloc(&buf, (size_t)bytes));
  MV_HIP(hipMalloc(&sink, 4));
  (void)hipMemset(buf, 1, (size_t)bytes);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  int rc = launch_read_bw(buf, bytes, sink, nullptr);  // warm-up
  (void)hipEventRecord(a, nullptr);
  for (int i = 0; i < iters && !rc; ++i) rc = launch_read_bw(buf, bytes, sink, nullptr);
  (void)hipEventRecord(b, nullptr);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  *out_gbps = ms > 0 ? (double)bytes * iters / (ms * 1e-3) / 1e9 : 0.0;
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(buf);
  (void)hipFree(sink);
  return rc;
}

int mv_merge_topk(int device, const float* d_scores, const int64_t* d_ids, int32_t world, int32_t kk, int32_t k, float* d_out_scores,
                  int64_t* d_out_ids, void* stream) {
  if (!d_scores || !d_ids || !d_out_scores || !d_out_ids) { set_error("mv_merge_topk: null argument"); return MV_ERR_INVALID; }
  DeviceGuard g(device);
  return launch_merge_topk(d_scores, d_ids, world, kk, k, d_out_scores, d_out_ids, (hipStream_t)stream);
}

int64_t mv_topk_block_bytes(int32_t kk) { return kk < 1 ? 0 : (((int64_t)kk * 12 + 15) / 16) * 16; }

int mv_merge_topk_blocks(int device, const void* d_blocks, int32_t world, int32_t kk, int32_t k, float* d_out_scores, int64_t* d_out_ids, void* stream) {
  if (!d_blocks || !d_out_scores || !d_out_ids) { set_error("mv_merge_topk_blocks: null argument"); return MV_ERR_INVALID; }
  DeviceGuard g(device);
  const int64_t bb = mv_topk_block_bytes(kk);  // multiple of 16: both strides are whole elements
  const char* b = static_cast<const char*>(d_blocks);
  return launch_merge_topk(reinterpret_cast<const float*>(b + (size_t)kk * 8), reinterpret_cast<const int64_t*>(b), world, kk, k, d_out_scores, d_out_ids,
                           (hipStream_t)stream, bb / 4, bb / 8);
}

int mv_calibrate(int device, int what, int64_t bytes, int32_t iters, double* out) {
  if (!out || iters < 1) { set_error("calibrate: bad argument"); return MV_ERR_INVALID; }
  DeviceGuard g(device);
  float* sink = nullptr;
  MV_HIP(hipMalloc(&sink, 4));
  hipEvent_t a_ev, b_ev;
  (void)hipEventCreate(&a_ev);
  (void)hipEventCreate(&b_ev);
  int rc = MV_OK;
  float ms = 0;
  if (what == MV_CAL_READ_NT) {
    if (bytes < (1 << 24)) { set_error("calibrate: need >= 16 MiB"); rc = MV_ERR_INVALID; }
    void* buf = nullptr;
    if (!rc && hipMalloc(&buf, (size_t)bytes) != hipSuccess) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (!rc) {
      (void)hipMemset(buf, 1, (size_t)bytes);
      rc = launch_read_bw_nt(buf, bytes, sink, nullptr);
      (void)hipEventRecord(a_ev, nullptr);
      for (int i = 0; i < iters && !rc; ++i) rc = launch_read_bw_nt(buf, bytes, sink, nullptr);
      (void)hipEventRecord(b_ev, nullptr);
      (void)hipEventSynchronize(b_ev);
      (void)hipEventElapsedTime(&ms, a_ev, b_ev);
      *out = ms > 0 ? (double)(bytes / 16384 * 16384) * iters / (ms * 1e-3) / 1e9 : 0.0;  // GB/s
    }
    if (buf) (void)hipFree(buf);
  } else if (what == MV_CAL_READ_LDSDMA) {
    // the float scan's own transport (nt LDS-DMA ring, 4 waves per 256 KiB page) with the arithmetic removed
    const int64_t page_bytes = 1024 * kRowBytes;
    const int64_t n = bytes / page_bytes;
    if (n < 64) { set_error("calibrate: need >= 16 MiB"); rc = MV_ERR_INVALID; }
    void* buf = nullptr;
    float* sc = nullptr;
    if (!rc && (hipMalloc(&buf, (size_t)n * page_bytes + 32768) != hipSuccess || hipMalloc(&sc, (size_t)n * 4) != hipSuccess)) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (!rc) {
      (void)hipMemset(buf, 1, (size_t)n * page_bytes);
      MaxsimArgs a{};
      a.slab = (const uint16_t*)buf; a.q = (const uint16_t*)buf; a.scores = sc; a.n = n; a.stride = (int32_t)(page_bytes / kRowBytes); a.q_tiles = 2;
      rc = launch_maxsim_bf16(a, 13, nullptr);
      (void)hipEventRecord(a_ev, nullptr);
      for (int i = 0; i < iters && !rc; ++i) rc = launch_maxsim_bf16(a, 13, nullptr);
      (void)hipEventRecord(b_ev, nullptr);
      (void)hipEventSynchronize(b_ev);
      (void)hipEventElapsedTime(&ms, a_ev, b_ev);
      *out = ms > 0 ? (double)(n * page_bytes) * iters / (ms * 1e-3) / 1e9 : 0.0;  // GB/s
    }
    if (buf) (void)hipFree(buf);
    if (sc) (void)hipFree(sc);
  } else if ((what >= MV_CAL_READ_STRIDED_512 && what <= MV_CAL_READ_ROWS_20K) || (what >= MV_CAL_DMA_STRIDED_128 && what <= MV_CAL_DMA_STRIDED_2K)) {
    // the batched FDE pass's access pattern (tiles of rows, `piece` contiguous bytes per row and step) without LDS, barriers or arithmetic
    const int piece = what == MV_CAL_READ_STRIDED_512 ? 512 : what == MV_CAL_READ_STRIDED_1K ? 1024 : what == MV_CAL_READ_STRIDED_2K ? 2048 : what == MV_CAL_READ_STRIDED_4K ? 4096 :
                      what == MV_CAL_DMA_STRIDED_128 ? -128 : what == MV_CAL_DMA_STRIDED_512 ? -512 : what == MV_CAL_DMA_STRIDED_1K ? -1024 : what == MV_CAL_DMA_STRIDED_2K ? -2048 : 20480;
    const int64_t n_rows = bytes / 20480 / 64 * 64;
    if (n_rows < 64 * 256) { set_error("calibrate: need >= 336 MB"); rc = MV_ERR_INVALID; }
    void* buf = nullptr;
    if (!rc && hipMalloc(&buf, (size_t)n_rows * 20480) != hipSuccess) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (!rc) {
      (void)hipMemset(buf, 1, (size_t)n_rows * 20480);
      rc = launch_read_bw_strided(buf, n_rows, piece, sink, nullptr);
      (void)hipEventRecord(a_ev, nullptr);
      for (int i = 0; i < iters && !rc; ++i) rc = launch_read_bw_strided(buf, n_rows, piece, sink, nullptr);
      (void)hipEventRecord(b_ev, nullptr);
      (void)hipEventSynchronize(b_ev);
      (void)hipEventElapsedTime(&ms, a_ev, b_ev);
      *out = ms > 0 ? (double)n_rows * 20480 * (piece > 8192 ? 1.2 : 1.0) * iters / (ms * 1e-3) / 1e9 : 0.0;  // GB/s (whole rows: 3 x 8 KiB read per 20 KiB row)
    }
    if (buf) (void)hipFree(buf);
  } else if (what == MV_CAL_STREAM_PROBE) {
    // shape of the work from the environment (read per call: tools/stream_structure_probe.py sweeps it inside one process)
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const int ct = env_int("MV_PROBE_CT", 64), own = env_int("MV_PROBE_OWN", 0), sched = env_int("MV_PROBE_SCHED", 0), bpc = env_int("MV_PROBE_BPC", 2);
    const int qload = env_int("MV_PROBE_QLOAD", 0);
    void* buf = nullptr;
    uint32_t* work = nullptr;
    float *qv = nullptr, *uo = nullptr;
    const int64_t unit_bytes = (int64_t)ct * (own == 3 ? 20480 : 4096);
    if (qload && (hipMalloc((void**)&qv, 40960) != hipSuccess || hipMalloc((void**)&uo, (size_t)(bytes / unit_bytes + 1) * 4) != hipSuccess)) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (qv) (void)hipMemset(qv, 0, 40960);
    if (bytes < (1 << 24)) { set_error("calibrate: need >= 16 MiB"); rc = MV_ERR_INVALID; }
    if (!rc && (hipMalloc(&buf, (size_t)bytes) != hipSuccess || hipMalloc((void**)&work, 64) != hipSuccess)) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (!rc) {
      (void)hipMemset(buf, 1, (size_t)bytes);
      (void)hipMemset(work, 0, 64);
      rc = launch_stream_probe(buf, bytes, ct, own, sched, bpc, work, sink, nullptr, qv, uo);
      (void)hipEventRecord(a_ev, nullptr);
      for (int i = 0; i < iters && !rc; ++i) rc = launch_stream_probe(buf, bytes, ct, own, sched, bpc, work, sink, nullptr, qv, uo);
      (void)hipEventRecord(b_ev, nullptr);
      (void)hipEventSynchronize(b_ev);
      (void)hipEventElapsedTime(&ms, a_ev, b_ev);
      *out = ms > 0 ? (double)(bytes / unit_bytes * unit_bytes) * iters / (ms * 1e-3) / 1e9 : 0.0;  // GB/s
    }
    if (buf) (void)hipFree(buf);
    if (work) (void)hipFree(work);
    if (qv) (void)hipFree(qv);
    if (uo) (void)hipFree(uo);
  } else if (what == MV_CAL_FDE_SCAN_REGS || what == MV_CAL_FDE_SCAN_ROWQ) {
    // the single-query FDE coarse scan itself (10 240-d rows, cosine on, no filter), `iters` launches back to back: what the
    // kernel sustains without the host gaps between requests (variant 0: plain nt loads; variant 5: row quarters on the nt LDS-DMA ring)
    const int64_t od = 10240, n = bytes / (od * 2);
    if (n < 1024) { set_error("calibrate: need >= 21 MB"); rc = MV_ERR_INVALID; }
    void* buf = nullptr;
    float *sc = nullptr, *qv = nullptr, *inv = nullptr;
    if (!rc && (hipMalloc(&buf, (size_t)n * od * 2) != hipSuccess || hipMalloc(&sc, (size_t)n * 4) != hipSuccess || hipMalloc(&inv, (size_t)n * 4) != hipSuccess ||
                hipMalloc(&qv, (size_t)od * 4) != hipSuccess)) { set_error("calibrate: out of memory"); rc = MV_ERR_NOMEM; }
    if (!rc) {
      (void)hipMemset(buf, 1, (size_t)n * od * 2);
      (void)hipMemset(inv, 0, (size_t)n * 4);
      (void)hipMemset(qv, 0, (size_t)od * 4);
      FdeScanArgs a{};
      a.fde = (const uint16_t*)buf; a.inv_norm = inv; a.q = qv; a.scores = sc; a.n = n; a.out_dim = od;
      const int v = what == MV_CAL_FDE_SCAN_REGS ? 0 : 5;
      rc = launch_fde_scan(a, v, nullptr);
      (void)hipEventRecord(a_ev, nullptr);
      for (int i = 0; i < iters && !rc; ++i) rc = launch_fde_scan(a, v, nullptr);
      (void)hipEventRecord(b_ev, nullptr);
      (void)hipEventSynchronize(b_ev);
      (void)hipEventElapsedTime(&ms, a_ev, b_ev);
      *out = ms > 0 ? (double)(n * od * 2) * iters / (ms * 1e-3) / 1e9 : 0.0;  // GB/s
    }
    if (buf) (void)hipFree(buf);
    if (sc) (void)hipFree(sc);
    if (qv) (void)hipFree(qv);
    if (inv) (void)hipFree(inv);
  } else if (what == MV_CAL_MFMA_BF16 || what == MV_CAL_MFMA_BF16_32X32) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    const int shape = what == MV_CAL_MFMA_BF16_32X32 ? 1 : 0;
    const int blocks = ncu * 2, loop = 1 << 16;  // 8 waves / CU; ~4 ms per launch: the clock settles at its sustained value
    rc = launch_mfma_peak(blocks, loop, shape, sink, nullptr);  // warm-up of the same length
    (void)hipEventRecord(a_ev, nullptr);
    for (int i = 0; i < iters && !rc; ++i) rc = launch_mfma_peak(blocks, loop, shape, sink, nullptr);
    (void)hipEventRecord(b_ev, nullptr);
    (void)hipEventSynchronize(b_ev);
    (void)hipEventElapsedTime(&ms, a_ev, b_ev);
    // per wave and iteration: 8 x (16x16x32) or 4 x (32x32x16) MFMAs = 131 072 flop either way
    const double flops = (double)blocks * 4 * loop * (shape ? 4 * (2.0 * 32 * 32 * 16) : 8 * (2.0 * 16 * 16 * 32)) * iters;
    *out = ms > 0 ? flops / (ms * 1e-3) / 1e12 : 0.0;  // TFLOP/s
  } else {
    set_error("calibrate: unknown measurement %d", what);
    rc = MV_ERR_INVALID;
  }
  (void)hipEventDestroy(a_ev);
  (void)hipEventDestroy(b_ev);
  (void)hipFree(sink);
  return rc;
}

// ---------------------------------------------------------------------------------- persistence
struct SaveHeader {  // "MVIDX003"
  char magic[8];
  mv_config cfg;
  int64_t size;
  int64_t fde_out_dim;
};
struct SaveHeaderV2 {  // "MVIDX002": checkpoints written before mv_config grew capacity_rows (fixed layout only)
  char magic[8];
  int32_t dim, stride_rows;
  int64_t capacity_pages;
  int32_t device, flags;
  int64_t id_base;
  mv_fde_config fde;
  int64_t size;
  int64_t fde_out_dim;
};

int mv_index_save(mv_index* ix, const char* path) {
  if (!ix || !path) { set_error("save: null argument"); return MV_ERR_INVALID; }
  ExclusiveLock lk(ix);  // a consistent image: no ingest, no compaction while the slabs are dumped
  DeviceGuard g(ix->cfg.device);
  // crash safety: write <path>.tmp, fsync, then rename over the previous checkpoint -- a crash mid-save leaves the old
  // file intact
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { set_error("cannot open %s for writing", tmp.c_str()); return MV_ERR_IO; }
  const int64_t size = ix->size.load();
  SaveHeader h{};
  memcpy(h.magic, "MVIDX003", 8);
  h.cfg = ix->cfg;
  h.size = size;
  h.fde_out_dim = ix->fde_t.out_dim;
  int rc = MV_OK;
  auto wr = [&](const void* p, size_t n) { if (!rc && n && fwrite(p, 1, n, f) != n) { set_error("short write to %s", tmp.c_str()); rc = MV_ERR_IO; } };
  wr(&h, sizeof(h));
  wr(ix->h_n_rows.data(), (size_t)size * 4);
  wr(ix->h_doc_ord.data(), (size_t)size * 4);
  if (ix->packed) wr(ix->h_row_off.data(), (size_t)(size + 1) * 8);  // packed layout: the page -> first-row table
  std::vector<char> buf((size_t)64 << 20);
  auto dump = [&](const void* d, size_t bytes) {
    size_t off = 0;
    while (!rc && off < bytes) {
      const size_t n = std::min(buf.size(), bytes - off);
      if (hipMemcpy(buf.data(), (const char*)d + off, n, hipMemcpyDeviceToHost) != hipSuccess) { set_error("D2H failed during save"); rc = MV_ERR_HIP; break; }
      wr(buf.data(), n);
      off += n;
    }
  };
  const size_t rows = (size_t)rows_in_use(ix, size);
  if (ix->cfg.flags & MV_WITH_FLOAT) dump(ix->slab, rows * kRowBytes);
  if (ix->cfg.flags & MV_WITH_FLOAT_LO) dump(ix->slab_lo, rows * kRowBytes);
  if (ix->cfg.flags & MV_WITH_BINARY) dump(ix->bits, rows * kSignBytes);
  if (ix->cfg.flags & MV_WITH_FDE) {
    dump(ix->fde, (size_t)size * ix->fde_t.out_dim * 2);
    dump(ix->fde_inv_norm, (size_t)size * 4);
  }
  if (ix->cfg.flags & MV_WITH_FP8) {
    dump(ix->slab8, rows * kDim);
    dump(ix->inv_scale8, (size_t)size * 4);
  }
  if (ix->cfg.flags & MV_WITH_HOST_EXACT) {  // the exact host tier, last: ONE logical stream of pages whatever the split (a load may split elsewhere)
    const size_t pb = (size_t)ix->cfg.stride_rows * kRowBytes;
    const int64_t n_hbm = std::min<int64_t>(size, ix->x_split);
    if (!ix->x_loc.empty()) {  // a rebalanced split tier: page by page through the placement table (the file keeps page order)
      for (int64_t p2 = 0; p2 < size && !rc; ++p2) {
        const int64_t sl = ix->x_loc[(size_t)p2];
        if (sl < ix->x_split) dump(ix->slab_x + (size_t)sl * (pb / 2), pb);
        else wr(ix->h_exact + (size_t)(sl - ix->x_split) * (pb / 2), pb);
      }
    } else {
      if (n_hbm > 0) dump(ix->slab_x, (size_t)n_hbm * pb);
      if (size > n_hbm) wr(ix->h_exact, (size_t)(size - n_hbm) * pb);
    }
  }
  if (!rc && (fflush(f) != 0 || fsync(fileno(f)) != 0)) { set_error("flush failed for %s", tmp.c_str()); rc = MV_ERR_IO; }
  if (fclose(f) != 0 && !rc) { set_error("close failed for %s", tmp.c_str()); rc = MV_ERR_IO; }
  if (!rc && rename(tmp.c_str(), path) != 0) { set_error("cannot rename %s to %s", tmp.c_str(), path); rc = MV_ERR_IO; }
  if (rc) (void)remove(tmp.c_str());
  return rc;
}

int mv_index_load(const char* path, int32_t device, mv_index** out) {
  if (!path || !out) { set_error("load: null argument"); return MV_ERR_INVALID; }
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) { set_error("cannot open %s", path); return MV_ERR_IO; }
  SaveHeader h{};
  char magic[8] = {0};
  if (fread(magic, 1, 8, f) != 8) { fclose(f); set_error("%s is not an mv index file", path); return MV_ERR_IO; }
  if (memcmp(magic, "MVIDX003", 8) == 0) {
    memcpy(h.magic, magic, 8);
    if (fread((char*)&h + 8, 1, sizeof(h) - 8, f) != sizeof(h) - 8) { fclose(f); set_error("%s: truncated header", path); return MV_ERR_IO; }
  } else if (memcmp(magic, "MVIDX002", 8) == 0) {  // the previous header: the same fields without capacity_rows
    SaveHeaderV2 o{};
    if (fread((char*)&o + 8, 1, sizeof(o) - 8, f) != sizeof(o) - 8) { fclose(f); set_error("%s: truncated header", path); return MV_ERR_IO; }
    h.cfg.dim = o.dim; h.cfg.stride_rows = o.stride_rows; h.cfg.capacity_pages = o.capacity_pages; h.cfg.device = o.device; h.cfg.flags = o.flags & ~MV_LAYOUT_PACKED;
    h.cfg.id_base = o.id_base; h.cfg.fde = o.fde; h.cfg.capacity_rows = 0; h.size = o.size; h.fde_out_dim = o.fde_out_dim;
  } else { fclose(f); set_error("%s is not an mv index file", path); return MV_ERR_IO; }
  // the header is untrusted input: everything read below is sized by it
  if (h.size < 0 || h.size > h.cfg.capacity_pages) { fclose(f); set_error("%s: size %lld outside 0..capacity %lld", path, (long long)h.size, (long long)h.cfg.capacity_pages); return MV_ERR_IO; }
  if ((h.cfg.flags & MV_WITH_FDE) && h.fde_out_dim != mv_fde_output_dim(&h.cfg.fde)) { fclose(f); set_error("%s: FDE width %lld does not match its FDE config", path, (long long)h.fde_out_dim); return MV_ERR_IO; }
  h.cfg.device = device;
  mv_index* ix = nullptr;
  int rc = mv_index_create(&h.cfg, &ix);  // validates dim / stride / capacity / flags
  if (rc) { fclose(f); return rc; }
  DeviceGuard g(device);
  auto rd = [&](void* p, size_t n) { if (!rc && n && fread(p, 1, n, f) != n) { set_error("short read from %s", path); rc = MV_ERR_IO; } };
  rd(ix->h_n_rows.data(), (size_t)h.size * 4);
  rd(ix->h_doc_ord.data(), (size_t)h.size * 4);
  for (int64_t p = 0; p < h.size && !rc; ++p)
    if (ix->h_n_rows[p] < 0 || ix->h_n_rows[p] > h.cfg.stride_rows) { set_error("%s: page %lld has %d rows (stride %d)", path, (long long)p, ix->h_n_rows[p], h.cfg.stride_rows); rc = MV_ERR_IO; }
  if (ix->packed) {  // the row table is untrusted too: ascending whole tiles, every page inside its slot, all inside the slabs
    rd(ix->h_row_off.data(), (size_t)(h.size + 1) * 8);
    if (!rc && ix->h_row_off[0] != 0) { set_error("%s: row table does not start at 0", path); rc = MV_ERR_IO; }
    for (int64_t p = 0; p < h.size && !rc; ++p) {
      const int64_t slot = ix->h_row_off[(size_t)p + 1] - ix->h_row_off[(size_t)p];
      if (slot < 0 || slot % 16 || slot > h.cfg.stride_rows || ix->h_n_rows[p] > slot || ix->h_row_off[(size_t)p + 1] > ix->cap_rows) { set_error("%s: bad row table at page %lld", path, (long long)p); rc = MV_ERR_IO; }
    }
    for (int64_t p = h.size + 1; p <= h.cfg.capacity_pages && !rc; ++p) ix->h_row_off[(size_t)p] = ix->h_row_off[(size_t)h.size];
    if (!rc && hipMemcpy(ix->d_row_off, ix->h_row_off.data(), (size_t)(h.cfg.capacity_pages + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D of the row table failed"); rc = MV_ERR_HIP; }
  }
  std::vector<char> buf((size_t)64 << 20);
  auto fill = [&](void* d, size_t bytes) {
    size_t off = 0;
    while (!rc && off < bytes) {
      const size_t n = std::min(buf.size(), bytes - off);
      rd(buf.data(), n);
      if (!rc && hipMemcpy((char*)d + off, buf.data(), n, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D failed during load"); rc = MV_ERR_HIP; }
      off += n;
    }
  };
  const size_t rows = rc ? 0 : (size_t)rows_in_use(ix, h.size);
  if (h.cfg.flags & MV_WITH_FLOAT) fill(ix->slab, rows * kRowBytes);
  if (h.cfg.flags & MV_WITH_FLOAT_LO) fill(ix->slab_lo, rows * kRowBytes);
  if (h.cfg.flags & MV_WITH_BINARY) fill(ix->bits, rows * kSignBytes);
  if (h.cfg.flags & MV_WITH_FDE) {
    fill(ix->fde, (size_t)h.size * ix->fde_t.out_dim * 2);
    fill(ix->fde_inv_norm, (size_t)h.size * 4);
    if (!rc && (ix->fde8 || ix->fde4)) {  // the e4m3 / fp4 copy is not in the file: derived again (the quantisers are deterministic)
      rc = fde8_requantize(ix, 0, h.size, nullptr);
      if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("%s: quantising the FDE slab failed", path); rc = MV_ERR_HIP; }
    }
  }
  if (h.cfg.flags & MV_WITH_FP8) {
    fill(ix->slab8, rows * kDim);
    fill(ix->inv_scale8, (size_t)h.size * 4);
  }
  if (h.cfg.flags & MV_WITH_HOST_EXACT) {  // this process's split, not the saver's
    const size_t pb = (size_t)h.cfg.stride_rows * kRowBytes;
    const int64_t n_hbm = std::min<int64_t>(h.size, ix->x_split);
    if (n_hbm > 0) fill(ix->slab_x, (size_t)n_hbm * pb);
    if (h.size > n_hbm) rd(ix->h_exact, (size_t)(h.size - n_hbm) * pb);
  }
  fclose(f);
  if (!rc && h.size) {
    if (hipMemcpy(ix->d_n_rows, ix->h_n_rows.data(), (size_t)h.size * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_doc_ord, ix->h_doc_ord.data(), (size_t)h.size * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D of metadata failed"); rc = MV_ERR_HIP; }
    if (!rc) publish_pages(ix, 0, h.size);
  }
  if (rc) { std::string keep = g_err; mv_index_destroy(ix); g_err = keep; return rc; }
  *out = ix;
  return MV_OK;
}

}  // extern "C"
