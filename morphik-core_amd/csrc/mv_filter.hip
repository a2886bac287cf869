// mv_filter.hip -- ordered compaction of the pages a doc_ids filter lets through.
//
// Real requests always carry the authorised document set (core/services/document_service.py:336-341,414 passes
// doc_ids to query_similar; MultiVectorStore turns it into "WHERE document_id IN (...)", multi_vector_store.py:754-757).
// The scan kernels skip a masked page after one 4-byte metadata read, but a launch still starts one workgroup per
// page: a 0.1 % filter over 400 k pages costs 0.38 ms of empty workgroups against 0.015 ms of reading.  For selective
// filters the allowed (and live) pages are compacted IN PAGE ORDER into a candidate list first (three tiny kernels),
// and the scan is launched over the candidates only; ascending order keeps the (score desc, id asc) tie rule.
#include "mv_common.h"

namespace mv {
namespace {

constexpr int kFChunk = 2048;  // pages per block: 256 threads x 8 consecutive pages

__device__ __forceinline__ bool page_allowed(const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits, int64_t p) {
  const int32_t o = doc_ord[p];
  if (o < 0) return false;
  if (!allow) return true;
  if ((int64_t)o >= n_allow_bits) return false;
  return ((allow[o >> 5] >> (o & 31)) & 1u) != 0u;
}

__global__ __launch_bounds__(256) void filter_count_kernel(const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits, int64_t n,
                                                           int32_t* counts) {
  __shared__ int32_t red[4];
  const int64_t base = (int64_t)blockIdx.x * kFChunk + threadIdx.x * 8;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (base + i < n && page_allowed(doc_ord, allow, n_allow_bits, base + i)) ++c;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) c += __shfl_xor(c, s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of the per-block counts (in place) by one block; counts[nblocks] receives the total
__global__ __launch_bounds__(256) void filter_scan_kernel(int32_t* counts, int nblocks) {
  __shared__ int32_t part[256];
  const int per = (nblocks + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
    counts[nblocks] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = lo; i < hi; ++i) { const int v = counts[i]; counts[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void filter_write_kernel(const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits, int64_t n,
                                                           const int32_t* offsets, int32_t* cand) {
  __shared__ int32_t tcount[256];
  const int64_t base = (int64_t)blockIdx.x * kFChunk + threadIdx.x * 8;
  uint32_t mask = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (base + i < n && page_allowed(doc_ord, allow, n_allow_bits, base + i)) mask |= 1u << i;
  tcount[threadIdx.x] = __popc(mask);
  __syncthreads();
  if (threadIdx.x == 0) {  // 256-entry exclusive scan: tiny
    int run = offsets[blockIdx.x];
    for (int i = 0; i < 256; ++i) { const int v = tcount[i]; tcount[i] = run; run += v; }
  }
  __syncthreads();
  int pos = tcount[threadIdx.x];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (mask & (1u << i)) cand[pos++] = (int32_t)(base + i);
}

}  // namespace

size_t filter_ws_bytes(int64_t capacity) { return ((size_t)(capacity + kFChunk - 1) / kFChunk + 2) * sizeof(int32_t); }

// d_cand must hold n entries, d_counts filter_ws_bytes(n).  Returns the number of candidates in *out_n (host; syncs s).
int launch_filter_compact(const int32_t* d_doc_ord, const uint32_t* d_allow, int64_t n_allow_bits, int64_t n, int32_t* d_counts,
                          int32_t* d_cand, int64_t* out_n, hipStream_t s) {
  *out_n = 0;
  if (n <= 0) return MV_OK;
  const int nblocks = (int)((n + kFChunk - 1) / kFChunk);
  hipLaunchKernelGGL(filter_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, d_doc_ord, d_allow, n_allow_bits, n, d_counts);
  hipLaunchKernelGGL(filter_scan_kernel, dim3(1), dim3(256), 0, s, d_counts, nblocks);
  hipLaunchKernelGGL(filter_write_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, d_doc_ord, d_allow, n_allow_bits, n,
                     (const int32_t*)d_counts, d_cand);
  MV_HIP(hipGetLastError());
  int32_t total = 0;
  MV_HIP(hipMemcpyAsync(&total, d_counts + nblocks, 4, hipMemcpyDeviceToHost, s));
  MV_HIP(hipStreamSynchronize(s));
  *out_n = total;
  return MV_OK;
}

}  // namespace mv
