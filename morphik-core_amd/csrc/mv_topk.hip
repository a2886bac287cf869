// mv_topk.hip -- largest-k selection over the per-page score vector.
//
// Replaces torch.topk(scores, min(k, len)) (core/vector_store/fast_multivector_store.py:556) and
// "ORDER BY similarity DESC LIMIT k" (core/vector_store/multi_vector_store.py:759).
// Order: score descending, ties by ascending page index (upstream leaves ties unspecified).
//
// Each score becomes a 64-bit key  (ordered(score) << 32) | ~index  so one unsigned compare gives
// the total order and keys are unique.  A block bitonic-sorts 2048 keys in LDS and keeps its best
// kk; levels repeat on the survivors until one block remains (1M pages, k=10: 512 -> 3 -> 1 blocks).
// The score vector is 4 B/page against 262 144 B/page for the scan, so this stage is <0.1% of the
// HBM traffic; it is latency (3 launches), not bandwidth.
#include "mv_common.h"

namespace mv {
namespace {

constexpr int kChunk = 2048;
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t ordered_u32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered_f32(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// keys_in == nullptr: build keys from scores[base .. base+n).  Invalid (-inf / NaN / out of range) -> 0.
__global__ __launch_bounds__(kThreads) void topk_level_kernel(const float* scores, const uint64_t* keys_in, int64_t n,
                                                              int kk, uint64_t* keys_out) {
  __shared__ uint64_t sk[kChunk];
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  for (int i = threadIdx.x; i < kChunk; i += kThreads) {
    const int64_t gi = base + i;
    uint64_t key = 0;
    if (gi < n) {
      if (keys_in) {
        key = keys_in[gi];
      } else {
        const float s = scores[gi] + 0.0f;  // -0 -> +0 so equal scores tie on index
        if (s == s && s != -INFINITY) key = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)gi);
      }
    }
    sk[i] = key;
  }
  __syncthreads();
  // bitonic sort, descending
  for (int size = 2; size <= kChunk; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < kChunk / 2; t += kThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const uint64_t a = sk[lo], b = sk[hi];
        const bool swap = desc ? (a < b) : (a > b);
        if (swap) { sk[lo] = b; sk[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < kk; i += kThreads) keys_out[(int64_t)blockIdx.x * kk + i] = sk[i];
}

__global__ void topk_decode_kernel(const uint64_t* keys, int k, const int32_t* ids_map, int64_t id_base, float* out_s,
                                   int64_t* out_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const uint64_t key = keys[i];
  if (key == 0) {
    out_s[i] = -INFINITY;
    out_id[i] = -1;
    return;
  }
  const uint32_t idx = ~(uint32_t)(key & 0xffffffffu);
  out_s[i] = unordered_f32((uint32_t)(key >> 32));
  out_id[i] = id_base + (ids_map ? (int64_t)ids_map[idx] : (int64_t)idx);
}

inline int64_t nblocks(int64_t n) { return (n + kChunk - 1) / kChunk; }

}  // namespace

size_t topk_ws_bytes(int64_t n, int32_t k) {
  if (k < 1) k = 1;
  if (k > kTopkMaxDeviceK) k = kTopkMaxDeviceK;
  // two ping-pong key buffers sized for the first level's survivors (+ one chunk of slack)
  const int64_t l1 = nblocks(n > 0 ? n : 1) * (int64_t)k;
  return (size_t)(2 * (l1 + kChunk)) * sizeof(uint64_t);
}

int launch_topk(const float* d_scores, int64_t n, int32_t k, const int32_t* d_ids_map, int64_t id_base, void* ws,
                float* d_out_scores, int64_t* d_out_ids, hipStream_t s) {
  if (k < 1 || k > kTopkMaxDeviceK) {
    set_error("launch_topk: k=%d outside 1..%d", k, kTopkMaxDeviceK);
    return MV_ERR_INVALID;
  }
  if (n > 0xffffffffLL) {
    set_error("launch_topk: n too large");
    return MV_ERR_INVALID;
  }
  uint64_t* bufA = reinterpret_cast<uint64_t*>(ws);
  const int64_t l1 = nblocks(n > 0 ? n : 1) * (int64_t)k;
  uint64_t* bufB = bufA + (l1 + kChunk);
  int64_t cur_n = n > 0 ? n : 0;
  int64_t blocks = nblocks(cur_n > 0 ? cur_n : 1);
  hipLaunchKernelGGL(topk_level_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, d_scores, (const uint64_t*)nullptr,
                     cur_n, (int)k, bufA);
  cur_n = blocks * k;
  uint64_t* in = bufA;
  uint64_t* out = bufB;
  while (blocks > 1) {
    blocks = nblocks(cur_n);
    hipLaunchKernelGGL(topk_level_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, (const float*)nullptr,
                       (const uint64_t*)in, cur_n, (int)k, out);
    cur_n = blocks * k;
    uint64_t* t = in; in = out; out = t;
  }
  hipLaunchKernelGGL(topk_decode_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, (const uint64_t*)in, (int)k,
                     d_ids_map, id_base, d_out_scores, d_out_ids);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
