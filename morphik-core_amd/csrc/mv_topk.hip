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
constexpr size_t kRadixTailBytes = 4096 * 8 + 2048 * 4 + 64;  // [survivors][histogram][RadixState]  // survivors + histogram + state (radix path, k > 32)

__device__ __forceinline__ uint32_t ordered_u32(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered_f32(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

__device__ __forceinline__ void bitonic_sort_desc(uint64_t* sk) {
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
}

struct TopkOut {  // only read by the final (single-block) level: decode fused into it
  const int32_t* ids_map;
  int64_t id_base;
  float* out_s;
  int64_t* out_id;
  int32_t k;
};

// One level of the selection.  A block owns `sub` consecutive chunks of 2048 keys: each is bitonic-sorted in LDS and
// its best kk survive into a second LDS array, which is sorted once more when sub > 1 (sub * kk <= 2048).  Fewer,
// fatter blocks mean fewer LEVELS -- the selection is launch-latency bound (~20 us per dependent launch), not
// bandwidth bound -- and the last level writes (score, id) itself instead of a separate decode launch:
// 1 M pages, k = 10: 2 launches (was 4).
// keys_in == nullptr: build keys from scores[base .. base+n).  Invalid (-inf / NaN / out of range) -> 0.
__global__ __launch_bounds__(kThreads) void topk_level_kernel(const float* scores, const uint64_t* keys_in, int64_t n,
                                                              int kk, uint64_t* keys_out, int sub, int final, TopkOut o) {
  __shared__ uint64_t sk[kChunk];
  __shared__ uint64_t best[kChunk];
  int nbest = 0;
  for (int sc = 0; sc < sub; ++sc) {
    const int64_t base = ((int64_t)blockIdx.x * sub + sc) * kChunk;
    if (sc > 0 && base >= n) break;  // block-uniform
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
    bitonic_sort_desc(sk);
    if (sub > 1) {
      for (int i = threadIdx.x; i < kk; i += kThreads) best[nbest + i] = sk[i];
      nbest += kk;
      __syncthreads();
    }
  }
  if (sub > 1) {
    for (int i = threadIdx.x; i < kChunk; i += kThreads) sk[i] = i < nbest ? best[i] : 0;
    __syncthreads();
    bitonic_sort_desc(sk);
  }
  if (!final) {
    for (int i = threadIdx.x; i < kk; i += kThreads) keys_out[(int64_t)blockIdx.x * kk + i] = sk[i];
    return;
  }
  for (int i = threadIdx.x; i < o.k; i += kThreads) {
    const uint64_t key = sk[i];
    if (key == 0) {
      o.out_s[i] = -INFINITY;
      o.out_id[i] = -1;
    } else {
      const uint32_t idx = ~(uint32_t)(key & 0xffffffffu);
      o.out_s[i] = unordered_f32((uint32_t)(key >> 32));
      o.out_id[i] = o.id_base + (o.ids_map ? (int64_t)o.ids_map[idx] : (int64_t)idx);
    }
  }
}

// Small k (<= 32): no sort at all.  Every thread keeps its 8 * SUB keys in registers; each of the kk rounds finds the
// block maximum (thread-local max -> DPP row max -> 16 row maxima through LDS, read back by everyone) and the owner
// retires its key.  One barrier per round, ~0.3 us per round, instead of a 66-stage bitonic network (~20-40 us):
// 1 M pages, k = 10: 2 launches of a few microseconds each.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }

template <int SUB>
__global__ __launch_bounds__(kThreads) void topk_extract_kernel(const float* scores, const uint64_t* keys_in, int64_t n, int kk,
                                                                uint64_t* keys_out, int final, TopkOut o) {
  __shared__ uint64_t rowmax[2][16];
  __shared__ uint64_t winners[32];
  constexpr int NK = 8 * SUB;
  uint64_t key[NK];
  const int64_t base = (int64_t)blockIdx.x * SUB * kChunk;
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int64_t gi = base + threadIdx.x + (int64_t)j * kThreads;
    uint64_t k64 = 0;
    if (gi < n) {
      if (keys_in) {
        k64 = keys_in[gi];
      } else {
        const float s = scores[gi] + 0.0f;
        if (s == s && s != -INFINITY) k64 = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)gi);
      }
    }
    key[j] = k64;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < kk; ++r) {
    uint64_t m = key[0];
#pragma unroll
    for (int j = 1; j < NK; ++j) m = umax64(m, key[j]);
    m = umax64(m, dpp_u64<0x128>(m));  // row_ror 8/4/2/1: maximum of the 16-lane row in every lane
    m = umax64(m, dpp_u64<0x124>(m));
    m = umax64(m, dpp_u64<0x122>(m));
    m = umax64(m, dpp_u64<0x121>(m));
    if ((lane & 15) == 0) rowmax[r & 1][wave * 4 + (lane >> 4)] = m;
    __syncthreads();
    uint64_t w = rowmax[r & 1][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) w = umax64(w, rowmax[r & 1][i]);
    if (threadIdx.x == 0) winners[r] = w;
    if (w != 0) {  // keys are unique: exactly one thread retires it
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (key[j] == w) key[j] = 0;
    }
  }
  __syncthreads();
  if (!final) {
    for (int i = threadIdx.x; i < kk; i += kThreads) keys_out[(int64_t)blockIdx.x * kk + i] = winners[i];
    return;
  }
  for (int i = threadIdx.x; i < o.k; i += kThreads) {
    const uint64_t k64 = winners[i];
    if (k64 == 0) {
      o.out_s[i] = -INFINITY;
      o.out_id[i] = -1;
    } else {
      const uint32_t idx = ~(uint32_t)(k64 & 0xffffffffu);
      o.out_s[i] = unordered_f32((uint32_t)(k64 >> 32));
      o.out_id[i] = o.id_base + (o.ids_map ? (int64_t)o.ids_map[idx] : (int64_t)idx);
    }
  }
}

// Large k (> 32) over many scores: radix threshold instead of a 2048/k-per-level sort cascade (k = 1000 over 1.25 M FDE
// scores took ~10 dependent bitonic levels, ~0.5 ms).  Three histogram passes over the order-preserving 32-bit keys
// (11 + 11 + 10 bits, the bin choice stays on the device) give the exact k-th key T; one compaction pass collects
// every key >= T (all ties included) as 64-bit (key, ~index) words; a single block sorts those <= 4096 survivors, so
// order and the lowest-index tie rule are exactly those of the cascade.  More than 4096 survivors (masses of equal
// scores) fall back to the cascade.
struct RadixState {
  uint32_t prefix_val;
  uint32_t prefix_mask;
  uint32_t k_remaining;
  uint32_t count;  // compaction counter
  uint32_t done;   // 1 = the radix path produced the result; 0 = more than kRadixCap survivors -> serial fallback runs
};
constexpr int kRadixBins = 2048;
constexpr int kRadixCap = 4096;

__global__ __launch_bounds__(256) void radix_init_kernel(RadixState* st, uint32_t* hist, uint32_t k) {
  for (int i = threadIdx.x; i < kRadixBins; i += 256) hist[i] = 0;
  if (threadIdx.x == 0) *st = RadixState{0u, 0u, k, 0u, 0u};
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const float* scores, int64_t n, int shift, int nbits, const RadixState* st,
                                                         uint32_t* hist) {
  __shared__ uint32_t h[kRadixBins];
  for (int i = threadIdx.x; i < kRadixBins; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t pm = st->prefix_mask, pv = st->prefix_val, bm = (1u << nbits) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = scores[i] + 0.0f;
    if (s == s && s != -INFINITY) {
      const uint32_t key = ordered_u32(s);
      if ((key & pm) == pv) atomicAdd(&h[(key >> shift) & bm], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRadixBins; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// One block of 256 threads, 8 bins each: the bin b >= 1 (highest first) where the count of keys in higher bins is still
// below k_remaining but reaches it with bin b; none -> bin 0 (absorbs a k beyond the number of valid scores).  Fixes
// those bits of the threshold, then clears the histogram for the next pass.
__global__ __launch_bounds__(256) void radix_pick_kernel(uint32_t* hist, int nbits, int shift, RadixState* st) {
  __shared__ uint32_t part[256];
  __shared__ uint32_t found_bin, found_above;
  const int t = threadIdx.x;
  uint32_t loc[8], sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { loc[j] = hist[t * 8 + j]; sum += loc[j]; }  // bins beyond 1 << nbits are zero
  part[t] = sum;
  if (t == 0) { found_bin = 0u; found_above = 0xffffffffu; }
  __syncthreads();
  // inclusive suffix sum over threads (Hillis-Steele), then exclusive = inclusive - own
  uint32_t v = sum;
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t o = (t + d < 256) ? part[t + d] : 0u;
    __syncthreads();
    v += o;
    part[t] = v;
    __syncthreads();
  }
  const uint32_t kr = st->k_remaining;
  uint32_t above = v - sum;  // keys in bins owned by higher threads
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    const int b = t * 8 + j;
    if (b >= 1 && above < kr && above + loc[j] >= kr) { found_bin = (uint32_t)b; found_above = above; }
    above += loc[j];
  }
  __syncthreads();
  if (t == 0) {
    // not found: bin 0, with every key of bins >= 1 above it (part[0] = total, loc[0] = bin 0 of thread 0)
    const uint32_t cum = found_above != 0xffffffffu ? found_above : part[0] - loc[0];
    st->k_remaining = kr - cum;
    st->prefix_val |= found_bin << shift;
    st->prefix_mask |= ((1u << nbits) - 1u) << shift;
  }
  for (int i = t; i < kRadixBins; i += 256) hist[i] = 0;
}

__global__ __launch_bounds__(256) void radix_compact_kernel(const float* scores, int64_t n, RadixState* st, uint64_t* out) {
  const uint32_t T = st->prefix_val;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = scores[i] + 0.0f;
    if (s == s && s != -INFINITY) {
      const uint32_t key = ordered_u32(s);
      if (key >= T) {
        const uint32_t pos = atomicAdd(&st->count, 1u);
        if (pos < kRadixCap) out[pos] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)i);
      }
    }
  }
}

__device__ __forceinline__ void topk_write_out(const uint64_t* sk, const TopkOut& o) {
  for (int i = threadIdx.x; i < o.k; i += kThreads) {
    const uint64_t key = sk[i];
    if (key == 0) {
      o.out_s[i] = -INFINITY;
      o.out_id[i] = -1;
    } else {
      const uint32_t idx = ~(uint32_t)(key & 0xffffffffu);
      o.out_s[i] = unordered_f32((uint32_t)(key >> 32));
      o.out_id[i] = o.id_base + (o.ids_map ? (int64_t)o.ids_map[idx] : (int64_t)idx);
    }
  }
}

// Last step of the radix path: ONE block sorts the survivors (their number is read from the device state, so the host
// never waits for it) and writes the k results.  More than kRadixCap survivors: leaves done = 0 for the fallback.
__global__ __launch_bounds__(kThreads) void radix_final_kernel(const uint64_t* surv, RadixState* st, int kk, TopkOut o) {
  __shared__ uint64_t sk[kChunk];
  __shared__ uint64_t best[kChunk];
  const uint32_t cnt = st->count;
  if (cnt > (uint32_t)kRadixCap) return;  // done stays 0
  const int sub = cnt > (uint32_t)kChunk ? 2 : 1;
  for (int sc = 0; sc < sub; ++sc) {
    for (int i = threadIdx.x; i < kChunk; i += kThreads) {
      const uint32_t gi = (uint32_t)sc * kChunk + i;
      sk[i] = gi < cnt ? surv[gi] : 0;
    }
    __syncthreads();
    bitonic_sort_desc(sk);
    if (sub > 1) {
      for (int i = threadIdx.x; i < kk; i += kThreads) best[sc * kk + i] = sk[i];
      __syncthreads();
    }
  }
  if (sub > 1) {
    for (int i = threadIdx.x; i < kChunk; i += kThreads) sk[i] = i < 2 * kk ? best[i] : 0;
    __syncthreads();
    bitonic_sort_desc(sk);
  }
  topk_write_out(sk, o);
  if (threadIdx.x == 0) st->done = 1u;
}

// Fallback of the radix path (masses of equal scores at the threshold): one block streams the whole score vector,
// keeping the best kk (<= 1024) keys in the lower half of a 2048-key LDS array and sorting 1024 new keys against them
// per step.  Slow (one block), deterministic, and launched unconditionally BEHIND radix_final_kernel so that the
// selection never needs a host decision: it returns at once when done == 1.
__global__ __launch_bounds__(kThreads) void topk_serial_fallback_kernel(const float* scores, int64_t n, const RadixState* st, TopkOut o) {
  __shared__ uint64_t sk[kChunk];
  if (st->done) return;
  constexpr int kHalf = kChunk / 2;
  for (int i = threadIdx.x; i < kHalf; i += kThreads) sk[i] = 0;
  for (int64_t base = 0; base < n; base += kHalf) {
    for (int i = threadIdx.x; i < kHalf; i += kThreads) {
      const int64_t gi = base + i;
      uint64_t key = 0;
      if (gi < n) {
        const float s = scores[gi] + 0.0f;
        if (s == s && s != -INFINITY) key = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)gi);
      }
      sk[kHalf + i] = key;
    }
    __syncthreads();
    bitonic_sort_desc(sk);  // best 1024 of (previous best, new 1024) end up in the lower half
  }
  topk_write_out(sk, o);
}

inline int64_t nblocks(int64_t n) { return (n + kChunk - 1) / kChunk; }

}  // namespace

size_t topk_ws_bytes(int64_t n, int32_t k) {
  if (k < 1) k = 1;
  if (k > kTopkMaxDeviceK) k = kTopkMaxDeviceK;
  // two ping-pong key buffers sized for the first level's survivors (+ one chunk of slack)
  const int64_t l1 = nblocks(n > 0 ? n : 1) * (int64_t)k;
  return (size_t)(2 * (l1 + kChunk)) * sizeof(uint64_t) + kRadixTailBytes;
}

// Merge of per-shard top-k lists after the all-gather (row-sharded corpus, morphik_core_amd/sharded.py): scores/ids are
// [world][kk], each row sorted (score desc, id asc) and padded with (-inf, -1), rank r owning ids below rank r+1's.
// One block: 64-bit keys (ordered score, ~position) -> bitonic sort -> first k.  Equal scores keep (rank, position)
// order, which is ascending id order: the same tie rule as a single index.  One launch instead of ~8 framework ops.
__global__ __launch_bounds__(kThreads) void merge_topk_kernel(const float* scores, const int64_t* ids, int n, int k, float* out_s,
                                                              int64_t* out_id) {
  __shared__ uint64_t sk[kChunk];
  for (int i = threadIdx.x; i < kChunk; i += kThreads) {
    uint64_t key = 0;
    if (i < n && ids[i] >= 0) {
      const float s = scores[i] + 0.0f;
      if (s == s && s != -INFINITY) key = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)i);
    }
    sk[i] = key;
  }
  __syncthreads();
  bitonic_sort_desc(sk);
  for (int i = threadIdx.x; i < k; i += kThreads) {
    const uint64_t key = i < kChunk ? sk[i] : 0;
    if (key == 0) {
      out_s[i] = -INFINITY;
      out_id[i] = -1;
    } else {
      out_s[i] = unordered_f32((uint32_t)(key >> 32));
      out_id[i] = ids[~(uint32_t)(key & 0xffffffffu)];
    }
  }
}

int launch_merge_topk(const float* d_scores, const int64_t* d_ids, int32_t world, int32_t kk, int32_t k, float* d_out_scores,
                      int64_t* d_out_ids, hipStream_t s) {
  const int64_t n = (int64_t)world * kk;
  if (world < 1 || kk < 1 || k < 1 || n > kChunk || k > kChunk) {
    set_error("merge_topk: world*kk = %lld and k = %d must be within 1..%d", (long long)n, k, kChunk);
    return MV_ERR_INVALID;
  }
  hipLaunchKernelGGL(merge_topk_kernel, dim3(1), dim3(kThreads), 0, s, d_scores, d_ids, (int)n, (int)k, d_out_scores, d_out_ids);
  MV_HIP(hipGetLastError());
  return MV_OK;
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
  const TopkOut out{d_ids_map, id_base, d_out_scores, d_out_ids, k};
  // Intermediate levels run one 2048-key chunk per block (full parallelism); as soon as the survivors fit one block
  // looping over <= max_sub chunks (sub * k <= 2048 best-of-chunk keys), that block finishes the selection.
  const int max_sub = std::min<int>(k <= 32 ? 4 : 8, std::max(1, kChunk / k));  // extraction form: <= 32 keys per thread
  auto pick_sub = [&](int64_t cur) {
    if (cur <= (int64_t)kChunk * max_sub) return (int)std::max<int64_t>(1, (cur + kChunk - 1) / kChunk);
    return 1;
  };
  int64_t cur_n = n > 0 ? n : 0;
  const float* sc = d_scores;
  const uint64_t* in = nullptr;
  uint64_t* outk = bufA;
  if (k > 32 && cur_n > 2 * kChunk) {
    // radix threshold + compaction; workspace tail: [survivors 4096 x u64][hist 2048 x u32][state].  Entirely in stream
    // order: the survivor count never comes back to the host.
    char* tail = reinterpret_cast<char*>(ws) + topk_ws_bytes(n, k) - kRadixTailBytes;
    uint64_t* surv = reinterpret_cast<uint64_t*>(tail);
    uint32_t* hist = reinterpret_cast<uint32_t*>(tail + kRadixCap * 8);
    RadixState* st = reinterpret_cast<RadixState*>(tail + kRadixCap * 8 + kRadixBins * 4);
    hipLaunchKernelGGL(radix_init_kernel, dim3(1), dim3(256), 0, s, st, hist, (uint32_t)k);
    const int grid = (int)std::min<int64_t>((cur_n + 255) / 256, 256 * 8);
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
      hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)grid), dim3(256), 0, s, sc, cur_n, shifts[p], bits[p], (const RadixState*)st, hist);
      hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, s, hist, bits[p], shifts[p], st);
    }
    hipLaunchKernelGGL(radix_compact_kernel, dim3((unsigned)grid), dim3(256), 0, s, sc, cur_n, st, surv);
    hipLaunchKernelGGL(radix_final_kernel, dim3(1), dim3(kThreads), 0, s, (const uint64_t*)surv, st, (int)k, out);
    hipLaunchKernelGGL(topk_serial_fallback_kernel, dim3(1), dim3(kThreads), 0, s, sc, cur_n, (const RadixState*)st, out);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  while (true) {
    const int sub = pick_sub(cur_n);
    const int64_t per_block = (int64_t)kChunk * sub;
    const int64_t blocks = std::max<int64_t>(1, (cur_n + per_block - 1) / per_block);
    const int final = blocks == 1;
    if (k <= 32) {
      switch (sub) {
        case 1: hipLaunchKernelGGL((topk_extract_kernel<1>), dim3((unsigned)blocks), dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out); break;
        case 2: hipLaunchKernelGGL((topk_extract_kernel<2>), dim3((unsigned)blocks), dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out); break;
        case 3: case 4: hipLaunchKernelGGL((topk_extract_kernel<4>), dim3((unsigned)blocks), dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out); break;
        default: hipLaunchKernelGGL((topk_extract_kernel<8>), dim3((unsigned)blocks), dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out); break;
      }
    } else {
      hipLaunchKernelGGL(topk_level_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, sub, final, out);
    }
    if (final) break;
    cur_n = blocks * k;
    sc = nullptr;
    in = outk;
    outk = (outk == bufA) ? bufB : bufA;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
