// mv_binary.hip -- the sign-bit path of MultiVectorStore (core/vector_store/multi_vector_store.py).
//
//  * sign pack      : _binary_quantize (:329-345) -> fast_ops.binary_quantize_packed
//                     (core/utils/fast_ops.py:191-227) -> morphik_rust binary_quantize_batch_packed
//                     (morphik_rust/src/binary_ops.rs:148-222): bit = v > 0.0 (0, -0, NaN -> 0), MSB first.
//  * binary MaxSim  : SQL max_sim(bit[],bit[]) (:285-313):
//                       SUM_q MAX_d ( 1 - bit_count(d # q) / bit_length(q) )   (COALESCE 0)
//                     == n_q - (SUM_q MIN_d popcount(d xor q)) / 128, an integer computation; the
//                     result is a multiple of 1/128 <= n_q and therefore exact in fp32 and fp64.
//  * hamming batch  : fast_ops.hamming_distance_batch (fast_ops.py:242-248, binary_ops.rs:267-292).
//
// The scan reads 16 B per patch row (16 KiB per 1024-patch page) and does 9 VALU ops per
// (query row, patch): it sits close to the crossover of the HBM and VALU rooflines.
#include <algorithm>

#include "mv_common.h"

namespace mv {
namespace {

// ------------------------------------------------------------------------------ sign pack
// generic fp32: one thread per output byte
__global__ void sign_pack_f32_kernel(const float* x, int64_t n_rows, int32_t d, uint8_t* out) {
  const int32_t nb = (d + 7) >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * nb) return;
  const int64_t row = i / nb;
  const int32_t b = (int32_t)(i - row * nb);
  const float* p = x + row * d + b * 8;
  const int32_t lim = d - b * 8 < 8 ? d - b * 8 : 8;
  uint32_t byte = 0;
  for (int32_t k = 0; k < lim; ++k)
    if (p[k] > 0.0f) byte |= 1u << (7 - k);
  out[i] = (uint8_t)byte;
}

// bf16 rows of width 128: one thread per 32 elements -> one little-endian dword of 4 packed bytes.
// bf16 v > 0  <=>  sign bit clear and magnitude in (0, inf]  <=>  (h - 1) < 0x7F80 as unsigned.
__device__ __forceinline__ uint32_t pos_bit(uint32_t h) { return ((h - 1u) & 0xffffu) < 0x7f80u ? 1u : 0u; }

__global__ __launch_bounds__(256) void sign_pack_bf16_kernel(const uint16_t* rows, int64_t n_words, uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  const uint4* p = reinterpret_cast<const uint4*>(rows + i * 32);
  uint32_t word = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const uint4 v = p[b];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t byte = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      byte |= pos_bit(w[k] & 0xffffu) << (7 - 2 * k);
      byte |= pos_bit(w[k] >> 16) << (6 - 2 * k);
    }
    word |= byte << (8 * b);
  }
  out[i] = word;
}

// ------------------------------------------------------------------------------ binary MaxSim
struct BArgs {
  const uint4* bits;
  const int32_t* n_rows;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const uint4* q;
  float* scores;
  int64_t n;
  int32_t stride;
  int32_t n_q;
};

__device__ __forceinline__ bool masked(const BArgs& a, int64_t page) {
  if (!a.doc_ord) return false;
  const int32_t o = a.doc_ord[page];
  if (o < 0) return true;
  if (!a.allow) return false;
  if ((int64_t)o >= a.n_allow_bits) return true;
  return ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u;
}

constexpr int kQChunk = 32;  // query rows whose running minima live in registers

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v = min(v, __shfl_xor(v, s));
  return v;
}

// One wave per page, 4 pages per block.  Lane l owns patches l, l+64, ...
__global__ __launch_bounds__(256) void maxsim_binary_kernel(BArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t page = (int64_t)blockIdx.x * 4 + wave;
  if (page >= a.n) return;
  if (masked(a, page)) {
    if (lane == 0) a.scores[page] = -INFINITY;
    return;
  }
  const int nr = a.n_rows ? a.n_rows[page] : a.stride;
  if (nr <= 0 || a.n_q <= 0) {
    if (lane == 0) a.scores[page] = 0.0f;  // COALESCE(SUM over nothing, 0.0)
    return;
  }
  const uint4* pg = a.bits + (size_t)page * (size_t)a.stride;
  int total = 0;
  for (int q0 = 0; q0 < a.n_q; q0 += kQChunk) {
    const int nq = min(kQChunk, a.n_q - q0);
    int mn[kQChunk];
#pragma unroll
    for (int i = 0; i < kQChunk; ++i) mn[i] = 1 << 20;
    for (int p = lane; p < nr; p += 64) {
      const uint4 d = pg[p];
#pragma unroll
      for (int i = 0; i < kQChunk; ++i) {
        if (i < nq) {  // wave-uniform
          const uint4 qv = a.q[q0 + i];  // scalar load (uniform address)
          const int hd = __popc(d.x ^ qv.x) + __popc(d.y ^ qv.y) + __popc(d.z ^ qv.z) + __popc(d.w ^ qv.w);
          mn[i] = min(mn[i], hd);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kQChunk; ++i)
      if (i < nq) total += wave_min(mn[i]);
  }
  if (lane == 0) a.scores[page] = (float)a.n_q - (float)total * (1.0f / 128.0f);
}

__global__ void hamming_batch_kernel(const uint8_t* q, const uint8_t* c, int64_t n, int32_t nb, int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = c + i * nb;
  int h = 0;
  for (int32_t k = 0; k < nb; ++k) h += __popc((uint32_t)(p[k] ^ q[k]));
  out[i] = h;
}

}  // namespace

int launch_sign_pack_f32(const float* d_x, int64_t n_rows, int32_t d, uint8_t* d_out, hipStream_t s) {
  const int64_t nb = (d + 7) / 8;
  if (n_rows * nb <= 0) return MV_OK;
  const int64_t rows_per = std::max<int64_t>(((int64_t)1 << 31) / nb, 1);  // < 2^32 work-items per launch
  for (int64_t r = 0; r < n_rows; r += rows_per) {
    const int64_t m = std::min(rows_per, n_rows - r);
    hipLaunchKernelGGL(sign_pack_f32_kernel, dim3((unsigned)((m * nb + 255) / 256)), dim3(256), 0, s, d_x + r * d, m, d,
                       d_out + r * nb);
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_sign_pack_bf16_rows(const uint16_t* d_rows, int64_t n_rows, uint8_t* d_out, hipStream_t s) {
  int64_t words = n_rows * (kDim / 32);
  int64_t done = 0;
  while (done < words) {  // keep each launch under 2^31 blocks
    int64_t w = words - done;
    const int64_t cap = (int64_t)1 << 31;  // work-items per launch must stay below 2^32
    if (w > cap) w = cap;
    hipLaunchKernelGGL(sign_pack_bf16_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, s, d_rows + done * 32, w,
                       reinterpret_cast<uint32_t*>(d_out) + done);
    done += w;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_maxsim_binary(const BinaryArgs& a, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  BArgs k{reinterpret_cast<const uint4*>(a.bits), a.n_rows, a.doc_ord, a.allow, a.n_allow_bits,
          reinterpret_cast<const uint4*>(a.qbits), a.scores, a.n, a.stride, a.n_q};
  if (a.n > ((int64_t)1 << 25)) { set_error("binary scan: more than 2^25 pages per launch is not supported"); return MV_ERR_INVALID; }
  hipLaunchKernelGGL(maxsim_binary_kernel, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, k);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_hamming_batch(const uint8_t* d_q, const uint8_t* d_c, int64_t n, int32_t n_bytes, int32_t* d_out,
                         hipStream_t s) {
  if (n <= 0) return MV_OK;
  hipLaunchKernelGGL(hamming_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_q, d_c, n, n_bytes, d_out);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
