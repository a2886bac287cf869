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

__device__ __forceinline__ uint32_t ordered_u32(float f) { return topk_ordered_u32(f); }
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

// Batched selection: blockIdx.y = query.  Every query has its own score vector, ids map, result row and workspace at a
// fixed stride from query 0's (all strides zero / unused for a single selection, grid.y = 1).
struct TopkBatch {
  int64_t score_stride;  // floats
  int64_t map_stride;    // int32 entries
  int64_t out_stride;    // result entries
  int64_t ws_stride;     // BYTES
};
template <typename T>
__device__ __forceinline__ T* tb_ws(T* p, const TopkBatch& tb) {
  return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)((int64_t)blockIdx.y * tb.ws_stride)) : p;
}
__device__ __forceinline__ void tb_out(TopkOut& o, const TopkBatch& tb) {
  if (o.ids_map) o.ids_map += (int64_t)blockIdx.y * tb.map_stride;
  o.out_s += (int64_t)blockIdx.y * tb.out_stride;
  o.out_id += (int64_t)blockIdx.y * tb.out_stride;
}

// One level of the selection.  A block owns `sub` consecutive chunks of 2048 keys: each is bitonic-sorted in LDS and
// its best kk survive into a second LDS array, which is sorted once more when sub > 1 (sub * kk <= 2048).  Fewer,
// fatter blocks mean fewer LEVELS -- the selection is launch-latency bound (~20 us per dependent launch), not
// bandwidth bound -- and the last level writes (score, id) itself instead of a separate decode launch:
// 1 M pages, k = 10: 2 launches (was 4).
// keys_in == nullptr: build keys from scores[base .. base+n).  Invalid (-inf / NaN / out of range) -> 0.
__global__ __launch_bounds__(kThreads) void topk_level_kernel(const float* scores, const uint64_t* keys_in, int64_t n,
                                                              int kk, uint64_t* keys_out, int sub, int final, TopkOut o, TopkBatch tb) {
  __shared__ uint64_t sk[kChunk];
  __shared__ uint64_t best[kChunk];
  if (scores) scores += (int64_t)blockIdx.y * tb.score_stride;
  keys_in = tb_ws(keys_in, tb);
  keys_out = tb_ws(keys_out, tb);
  tb_out(o, tb);
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
                                                                uint64_t* keys_out, int final, TopkOut o, TopkBatch tb) {
  __shared__ uint64_t rowmax[2][16];
  __shared__ uint64_t winners[32];
  if (scores) scores += (int64_t)blockIdx.y * tb.score_stride;
  keys_in = tb_ws(keys_in, tb);
  keys_out = tb_ws(keys_out, tb);
  tb_out(o, tb);
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
// scores took ~10 dependent bitonic levels, ~0.5 ms).  Two histogram passes over the order-preserving 32-bit keys
// (11 + 11 bits) fix the top 22 bits T of the k-th key; a compaction pass collects every key >= T (the top k plus the
// rest of the threshold bin, all ties included) as 64-bit (key, ~index) words; ONE block merge-sorts the survivors, so
// order and the lowest-index tie rule are exactly those of the cascade.  Nothing is decided on the host and no kernel
// takes a device-wide fence (on a multi-XCD part a fence per block costs more than a launch -- measured): the bin
// choice of a pass is recomputed by every block of the NEXT kernel from the finished histogram (2048 bins, ~2 us), so
// the chain is memset + 4 launches.  More than kRadixCap survivors (masses of equal scores): the final block streams
// the whole score vector instead (slow, deterministic, rare).
constexpr int kRadixBins = 2048;
constexpr int kRadixCap = 16384;
// workspace head (fixed place, zeroed once when the workspace is allocated):
//   [survivors kRadixCap x u64][histA 2048 x u32][histB 2048 x u32][RadixCtl]
// No memset in the chain: the histograms are cleared for the NEXT selection by the rank kernel of this one (they are
// dead by then), the survivor counter by the first histogram kernel of the next one (it is dead until the compaction).
struct RadixCtl {
  uint32_t count;
  uint32_t pad[15];
};
constexpr size_t kRadixTailBytes = (size_t)kRadixCap * 8 + 2 * kRadixBins * 4 + sizeof(RadixCtl);

// Every thread of a 256-thread block gets (bin, keys_above): the bin b >= 1 (highest first) where the number of keys in
// higher bins is still below kr but reaches it with bin b; none -> bin 0 with every key of bins >= 1 above it (absorbs a
// k beyond the number of valid scores).  Suffix sums: shuffles inside a wave, four wave totals through LDS.
__device__ __forceinline__ void radix_pick(const uint32_t* hist, uint32_t kr, uint32_t* out_bin, uint32_t* out_above) {
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t found_bin, found_above;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t loc[8], sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { loc[j] = hist[t * 8 + j]; sum += loc[j]; }
  uint32_t v = sum;  // inclusive suffix sum over the lanes of this wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_down(v, d);
    if (lane + d < 64) v += o;
  }
  __syncthreads();  // wtot / found_* may still be read from a previous call
  if (lane == 0) wtot[wave] = v;
  if (t == 0) { found_bin = 0u; found_above = 0xffffffffu; }
  __syncthreads();
  uint32_t above = v - sum;  // keys in bins owned by higher threads
  for (int w = wave + 1; w < 4; ++w) above += wtot[w];
#pragma unroll
  for (int j = 7; j >= 0; --j) {
    const int b = t * 8 + j;
    if (b >= 1 && above < kr && above + loc[j] >= kr) { found_bin = (uint32_t)b; found_above = above; }
    above += loc[j];
  }
  __syncthreads();
  *out_bin = found_bin;
  *out_above = found_above != 0xffffffffu ? found_above : (wtot[0] + wtot[1] + wtot[2] + wtot[3]) - hist[0];
}

constexpr int kShift0 = 21, kShift1 = 10;  // key bits [31:21] then [20:10]

__device__ __forceinline__ void hist_add_wave(uint32_t* h, uint32_t bin, bool valid) { topk_hist_add_wave(h, bin, valid); }

// Four consecutive scores of a selection pass: ONE 16-byte non-temporal load when the vector is 16-byte aligned and inside the
// vector, else guarded scalar loads (-inf past the end).  A 4-byte load per thread and trip keeps ~32 KiB per CU in flight; the
// 160 MB score matrix of a 32-request batch streamed at 3.4 TB/s that way (47 us per pass).
typedef float topk_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_scores4(float (&v)[4], const float* s, int64_t e0, int64_t n, bool vec) {
  if (vec && e0 + 3 < n) {
    const topk_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const topk_f32x4*>(s + e0));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = e0 + e < n ? __builtin_nontemporal_load(s + e0 + e) : -INFINITY;
  }
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const float* scores, int64_t n, int pass, uint32_t k, const uint32_t* hist_prev,
                                                         uint32_t* hist, RadixCtl* ctl, TopkBatch tb) {
  __shared__ uint32_t h[kRadixBins];
  scores += (int64_t)blockIdx.y * tb.score_stride;
  hist_prev = tb_ws(hist_prev, tb);
  hist = tb_ws(hist, tb);
  ctl = tb_ws(ctl, tb);
  if (blockIdx.x == 0 && threadIdx.x == 0) ctl->count = 0;  // dead until the compaction kernel (both passes: pass 0 may have run inside the scan)
  for (int i = threadIdx.x; i < kRadixBins; i += 256) h[i] = 0;
  uint32_t pm = 0, pv = 0;
  if (pass == 1) {
    uint32_t bin0, above0;
    radix_pick(hist_prev, k, &bin0, &above0);
    pm = 0x7ffu << kShift0;
    pv = bin0 << kShift0;
  }
  __syncthreads();
  const int shift = pass == 0 ? kShift0 : kShift1;
  const bool vec = (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
  const int64_t step = (int64_t)gridDim.x * 256;
  constexpr int U = 2;  // independent 16-byte loads in flight per thread
  const int64_t n4 = (n + 3) >> 2;
  const int64_t n_round = ((n4 + U * step - 1) / (U * step)) * (U * step);  // whole waves walk the loop together (the ballots need every lane)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += U * step) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) load_scores4(v[u], scores, (i + u * step) * 4, n, vec);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bool valid = false;
        uint32_t bin = 0;
        const float s = v[u][e] + 0.0f;
        if (s == s && s != -INFINITY) {
          const uint32_t key = ordered_u32(s);
          valid = (key & pm) == pv;
          bin = (key >> shift) & 0x7ffu;
        }
        if (pass == 0) hist_add_wave(h, bin, valid);
        else if (valid) atomicAdd(&h[bin], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRadixBins; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(256) void radix_compact_kernel(const float* scores, int64_t n, uint32_t k, const uint32_t* histA, const uint32_t* histB,
                                                            RadixCtl* ctl, uint64_t* out, TopkBatch tb) {
  scores += (int64_t)blockIdx.y * tb.score_stride;
  histA = tb_ws(histA, tb);
  histB = tb_ws(histB, tb);
  ctl = tb_ws(ctl, tb);
  out = tb_ws(out, tb);
  uint32_t* count = &ctl->count;
  uint32_t bin0, above0, bin1, above1;
  radix_pick(histA, k, &bin0, &above0);
  radix_pick(histB, k - above0, &bin1, &above1);
  const uint32_t T = (bin0 << kShift0) | (bin1 << kShift1);
  const bool vec = (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
  const int64_t step = (int64_t)gridDim.x * 256;
  constexpr int U = 2;  // independent 16-byte loads in flight per thread
  const int64_t n4 = (n + 3) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += U * step) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) load_scores4(v[u], scores, (i + u * step) * 4, n, vec);  // past the end: -inf
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s = v[u][e] + 0.0f;
        if (s == s && s != -INFINITY) {
          const uint32_t key = ordered_u32(s);
          if (key >= T) {
            const uint32_t pos = atomicAdd(count, 1u);
            if (pos < (uint32_t)kRadixCap) out[pos] = ((uint64_t)key << 32) | (uint32_t)(~(uint32_t)((i + u * step) * 4 + e));
          }
        }
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

// Last step: every survivor's RANK (number of larger keys; keys are unique) computed by brute force against the survivor
// list staged 2048 keys at a time in LDS -- ~1600 survivors x 1600 compares spread over several blocks is a few
// microseconds, against ~40 for a single-block bitonic merge -- and the owner of rank < k writes result slot `rank`.
// More survivors than the list holds (masses of equal scores): block 0 streams the whole score vector instead, keeping the
// best kk (<= 1024) keys in the lower half of a 2048-key LDS array and sorting 1024 new keys against them per step (slow,
// deterministic, rare); the other blocks leave.
__global__ __launch_bounds__(kThreads) void radix_rank_kernel(const float* scores, int64_t n, const uint64_t* surv, RadixCtl* ctl, uint32_t* hists,
                                                              TopkOut o, TopkBatch tb) {
  __shared__ __attribute__((aligned(16))) uint64_t sk[kChunk];
  scores += (int64_t)blockIdx.y * tb.score_stride;
  surv = tb_ws(surv, tb);
  ctl = tb_ws(ctl, tb);
  hists = tb_ws(hists, tb);
  tb_out(o, tb);
  const uint32_t cnt = ctl->count;
  if (blockIdx.x == gridDim.x - 1)  // housekeeping for the next selection (a block past the survivor list in practice)
    for (int i = threadIdx.x; i < 2 * kRadixBins; i += kThreads) hists[i] = 0;
  if (cnt <= (uint32_t)kRadixCap) {
    const uint32_t me = blockIdx.x * kThreads + threadIdx.x;
    const uint64_t mine = me < cnt ? surv[me] : 0;
    uint32_t rank = 0;
    if ((uint32_t)blockIdx.x * kThreads < cnt) {  // block-uniform: blocks past the list only pad
      for (uint32_t base = 0; base < cnt; base += kChunk) {
        __syncthreads();
        for (int i = threadIdx.x; i < kChunk; i += kThreads) sk[i] = base + i < cnt ? surv[base + i] : 0;
        __syncthreads();
        // broadcast reads, two keys per ds_read_b128 (the slots past the list hold 0, which is never larger than a key)
        const int m2 = ((int)min((uint32_t)kChunk, cnt - base) + 1) >> 1;
        const ulonglong2* sk2 = reinterpret_cast<const ulonglong2*>(sk);
#pragma unroll 8
        for (int i = 0; i < m2; ++i) {
          const ulonglong2 o2 = sk2[i];
          rank += (o2.x > mine ? 1u : 0u) + (o2.y > mine ? 1u : 0u);
        }
      }
      if (me < cnt && rank < (uint32_t)o.k) {
        const uint32_t idx = ~(uint32_t)(mine & 0xffffffffu);
        o.out_s[rank] = unordered_f32((uint32_t)(mine >> 32));
        o.out_id[rank] = o.id_base + (o.ids_map ? (int64_t)o.ids_map[idx] : (int64_t)idx);
      }
    }
    if (blockIdx.x == 0)  // fewer valid scores than k: pad the tail
      for (int i = (int)cnt + threadIdx.x; i < o.k; i += kThreads) { o.out_s[i] = -INFINITY; o.out_id[i] = -1; }
    return;
  }
  if (blockIdx.x != 0) return;
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

// The last step of every reranked request: k of a list of <= 256 scores (the reference's min(10 k, 75) candidates, torch.topk at
// fast_multivector_store.py:556).  ONE wave: four keys per lane in registers, k rounds of lane maximum -> xor butterfly -> the owner
// retires its key; no LDS, no barrier.  The block-wide extraction kernel spent ~10 us on a 75-entry list (rocprofv3 trace of one
// request, profiles/r5), this one is at the floor of a launch.  Same keys, same tie rule (score desc, list position asc).
__global__ __launch_bounds__(64) void topk_small_kernel(const float* scores, int n, int kk, TopkOut o, TopkBatch tb) {
  if (scores) scores += (int64_t)blockIdx.y * tb.score_stride;
  tb_out(o, tb);
  const int lane = threadIdx.x;
  uint64_t key[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gi = lane + 64 * j;
    uint64_t k64 = 0;
    if (gi < n) {
      const float s = scores[gi] + 0.0f;
      if (s == s && s != -INFINITY) k64 = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)gi);
    }
    key[j] = k64;
  }
  uint64_t mine = 0;  // lane r keeps the r-th winner
  for (int r = 0; r < kk; ++r) {
    uint64_t m = umax64(umax64(key[0], key[1]), umax64(key[2], key[3]));
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)m, sft);
      const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(m >> 32), sft);
      m = umax64(m, ((uint64_t)hi << 32) | lo);
    }
    if (lane == r) mine = m;
    if (m == 0) break;  // the list is exhausted (wave-uniform): the remaining lanes keep 0 = padding
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (key[j] == m) key[j] = 0;  // keys are unique (they carry their position)
  }
  if (lane < o.k) {
    if (mine == 0) {
      o.out_s[lane] = -INFINITY;
      o.out_id[lane] = -1;
    } else {
      const uint32_t idx = ~(uint32_t)(mine & 0xffffffffu);
      o.out_s[lane] = unordered_f32((uint32_t)(mine >> 32));
      o.out_id[lane] = o.id_base + (o.ids_map ? (int64_t)o.ids_map[idx] : (int64_t)idx);
    }
  }
}

// Merge of per-shard top-k lists after the all-gather (row-sharded corpus, morphik_core_amd/sharded.py): scores/ids are
// [world][kk], each row sorted (score desc, id asc) and padded with (-inf, -1), rank r owning ids below rank r+1's.
// One block: 64-bit keys (ordered score, ~position) -> bitonic sort -> first k.  Equal scores keep (rank, position)
// order, which is ascending id order: the same tie rule as a single index.  One launch instead of ~8 framework ops.
// Rank r's list: scores at scores + r * s_stride (floats), ids at ids + r * i_stride (int64s), kk entries each -- two separate
// [world][kk] arrays (s_stride = i_stride = kk) or the per-rank BLOCKS of one all-gather ({ids[kk], scores[kk]} per rank).
__global__ __launch_bounds__(kThreads) void merge_topk_kernel(const float* scores, const int64_t* ids, int n, int kk, int64_t s_stride, int64_t i_stride,
                                                              int k, float* out_s, int64_t* out_id) {
  __shared__ uint64_t sk[kChunk];
  for (int i = threadIdx.x; i < kChunk; i += kThreads) {
    uint64_t key = 0;
    if (i < n) {
      const int r = i / kk, j = i - r * kk;
      if (ids[r * i_stride + j] >= 0) {
        const float s = scores[r * s_stride + j] + 0.0f;
        if (s == s && s != -INFINITY) key = ((uint64_t)ordered_u32(s) << 32) | (uint32_t)(~(uint32_t)i);
      }
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
      const int pos = (int)(~(uint32_t)(key & 0xffffffffu));
      const int r = pos / kk, j = pos - r * kk;
      out_s[i] = unordered_f32((uint32_t)(key >> 32));
      out_id[i] = ids[r * i_stride + j];
    }
  }
}

int launch_merge_topk(const float* d_scores, const int64_t* d_ids, int32_t world, int32_t kk, int32_t k, float* d_out_scores,
                      int64_t* d_out_ids, hipStream_t s, int64_t s_stride, int64_t i_stride) {
  const int64_t n = (int64_t)world * kk;
  if (world < 1 || kk < 1 || k < 1 || n > kChunk || k > kChunk) {
    set_error("merge_topk: world*kk = %lld and k = %d must be within 1..%d", (long long)n, k, kChunk);
    return MV_ERR_INVALID;
  }
  hipLaunchKernelGGL(merge_topk_kernel, dim3(1), dim3(kThreads), 0, s, d_scores, d_ids, (int)n, (int)kk, s_stride > 0 ? s_stride : (int64_t)kk,
                     i_stride > 0 ? i_stride : (int64_t)kk, (int)k, d_out_scores, d_out_ids);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

bool topk_uses_radix(int64_t n, int32_t k) { return k > 32 && n > 2 * kChunk; }
uint32_t* topk_radix_hist0(void* ws) { return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ws) + (size_t)kRadixCap * 8); }

int launch_topk_batch(const float* d_scores, int64_t score_stride, int64_t n, int32_t k, const int32_t* d_ids_map, int64_t map_stride,
                      int64_t id_base, void* ws, size_t ws_stride, float* d_out_scores, int64_t* d_out_ids, int64_t out_stride, int nb,
                      hipStream_t s, bool hist0_done) {
  if (k < 1 || k > kTopkMaxDeviceK) {
    set_error("launch_topk: k=%d outside 1..%d", k, kTopkMaxDeviceK);
    return MV_ERR_INVALID;
  }
  if (n > 0xffffffffLL) {
    set_error("launch_topk: n too large");
    return MV_ERR_INVALID;
  }
  if (nb < 1 || nb > 65535) { set_error("launch_topk: batch of %d selections", nb); return MV_ERR_INVALID; }
  const TopkBatch tb{score_stride, map_stride, out_stride, (int64_t)ws_stride};
  const unsigned gy = (unsigned)nb;
  uint64_t* bufA = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(ws) + kRadixTailBytes);  // behind the radix head
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
  if (cur_n <= 256 && k <= 64 && d_scores) {  // a rerank list: one wave finishes it
    hipLaunchKernelGGL(topk_small_kernel, dim3(1, gy), dim3(64), 0, s, sc, (int)cur_n, (int)std::min<int64_t>(k, 64), out, tb);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if (topk_uses_radix(cur_n, k)) {
    char* head = reinterpret_cast<char*>(ws);
    uint64_t* surv = reinterpret_cast<uint64_t*>(head);
    uint32_t* histA = reinterpret_cast<uint32_t*>(head + (size_t)kRadixCap * 8);
    uint32_t* histB = histA + kRadixBins;
    RadixCtl* ctl = reinterpret_cast<RadixCtl*>(histB + kRadixBins);
    // fewer, fatter blocks for a batch: every block pays ~2-4 us of fixed work (LDS histogram clear / flush, the bin choice
    // recomputed from the previous pass) whatever it scans, and nb selections multiply the block count
    const int grid = (int)std::min<int64_t>(((cur_n + 3) / 4 + 255) / 256, nb > 8 ? 64 : (nb > 1 ? 256 : 256 * 8));  // a thread takes four scores per load
    if (!hist0_done)
      hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)grid, gy), dim3(256), 0, s, sc, cur_n, 0, (uint32_t)k, (const uint32_t*)nullptr, histA, ctl, tb);
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)grid, gy), dim3(256), 0, s, sc, cur_n, 1, (uint32_t)k, (const uint32_t*)histA, histB, ctl, tb);
    hipLaunchKernelGGL(radix_compact_kernel, dim3((unsigned)grid, gy), dim3(256), 0, s, sc, cur_n, (uint32_t)k, (const uint32_t*)histA,
                       (const uint32_t*)histB, ctl, surv, tb);
    hipLaunchKernelGGL(radix_rank_kernel, dim3((unsigned)(kRadixCap / kThreads), gy), dim3(kThreads), 0, s, sc, cur_n, (const uint64_t*)surv, ctl, histA, out, tb);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  while (true) {
    const int sub = pick_sub(cur_n);
    const int64_t per_block = (int64_t)kChunk * sub;
    const int64_t blocks = std::max<int64_t>(1, (cur_n + per_block - 1) / per_block);
    const int final = blocks == 1;
    const dim3 gr((unsigned)blocks, gy);
    if (k <= 32) {
      switch (sub) {
        case 1: hipLaunchKernelGGL((topk_extract_kernel<1>), gr, dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out, tb); break;
        case 2: hipLaunchKernelGGL((topk_extract_kernel<2>), gr, dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out, tb); break;
        case 3: case 4: hipLaunchKernelGGL((topk_extract_kernel<4>), gr, dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out, tb); break;
        default: hipLaunchKernelGGL((topk_extract_kernel<8>), gr, dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, final, out, tb); break;
      }
    } else {
      hipLaunchKernelGGL(topk_level_kernel, gr, dim3(kThreads), 0, s, sc, in, cur_n, (int)k, outk, sub, final, out, tb);
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

int launch_topk(const float* d_scores, int64_t n, int32_t k, const int32_t* d_ids_map, int64_t id_base, void* ws,
                float* d_out_scores, int64_t* d_out_ids, hipStream_t s, bool hist0_done) {
  return launch_topk_batch(d_scores, 0, n, k, d_ids_map, 0, id_base, ws, 0, d_out_scores, d_out_ids, 0, 1, s, hist0_done);
}

}  // namespace mv
