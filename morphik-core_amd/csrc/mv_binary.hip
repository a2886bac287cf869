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
  const float* qpop;  // [n_q padded to 16] popc of each query row as float, -1 for padding rows (MFMA variant)
  const int32_t* cand;  // optional candidate list: item i scores page cand[i] into scores[i] (variants 0 and 2..4)
  const int64_t* row_off;  // packed layout: first slab row of every page; null: page * stride
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
template <bool PK>  // PK: packed layout (row-offset table) -- a template parameter so the fixed layout's code is untouched
__global__ __launch_bounds__(256) void maxsim_binary_kernel(BArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= a.n) return;
  const int64_t page = a.cand ? (int64_t)a.cand[item] : item;
  if (masked(a, page)) {
    if (lane == 0) a.scores[item] = -INFINITY;
    return;
  }
  const int nr = a.n_rows ? a.n_rows[page] : a.stride;
  if (nr <= 0 || a.n_q <= 0) {
    if (lane == 0) a.scores[item] = 0.0f;  // COALESCE(SUM over nothing, 0.0)
    return;
  }
  const uint4* pg = a.bits + (PK ? (size_t)a.row_off[page] : (size_t)page * (size_t)a.stride);
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
  if (lane == 0) a.scores[item] = (float)a.n_q - (float)total * (1.0f / 128.0f);
}

// ------------------------------------------------------------------------------ binary MaxSim on the matrix cores
// The same integer result as the popcount kernel above, computed by v_mfma_scale_f32_16x16x128_f8f6f4 with
// FP4 (e2m1) operands -- the whole K=128 bit row in ONE MFMA per 16 query rows x 16 patch rows:
//   A (query bits q)  : nibble = q ? +1.0 (0x2) : -1.0 (0xA)
//   B (page bits d)   : nibble = d ?  0.5 (0x1) :  0.0 (0x0)          block scales = 2^0
//   D[q][d] = sum_k A.B = popc(q & d) - popc(d) / 2        (multiples of 0.5, |D| <= 64: exact in fp32)
//   hamming(q, d) = popc(q) + popc(d) - 2 popc(q & d) = popc(q) - 2 D[q][d]
//   min_d hamming = popc(q) - 2 max_d D[q][d]
// so the row-max over patches is the same elementwise-max-then-butterfly as the float kernel.
// K-slot mapping (any bijection works as long as A and B agree): lane (r = l&15, g = l>>4) holds, for row r,
// nibble n of operand dword i = bit (4n + g) of the row's little-endian dword i  =>  B_i = (w_i >> g) & 0x11111111:
// 4 shifts + 4 ands expand a 16-byte row; the MFMA pipe (2 x ~22 cycles per tile at Q=32) and the VALU
// (~16 ops per tile) stay below the HBM time of a 256-byte tile.
// Data movement: one wave per page; 1 KiB (64 rows) per global_load_lds_dwordx4 into a wave-private ring,
// ds_read_b128 with 4-lane broadcast (rows r = 0..15 hit 64 distinct banks).  16 KiB per 1024-patch page.
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4b = __attribute__((ext_vector_type(4))) float;

constexpr int kBinSlotRows = 64;                       // rows per DMA instruction
constexpr int kBinSlotBytes = kBinSlotRows * kSignBytes;  // 1 KiB

template <int N>
__device__ __forceinline__ void bin_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// max over the 16 lanes of a DPP row (lanes with equal l>>4), result in every lane: rotate-by-8/4/2/1 within
// the row (v_max_f32_dpp row_ror) -- no LDS traffic, unlike __shfl_xor (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float bin_group16_max(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));  // row_ror:8
  v = fmaxf(v, dpp_mov<0x124>(v));  // row_ror:4
  v = fmaxf(v, dpp_mov<0x122>(v));  // row_ror:2
  v = fmaxf(v, dpp_mov<0x121>(v));  // row_ror:1
  return v;
}
__device__ __forceinline__ float bin_group16_sum(float v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x121>(v);  // row_ror:1
  return v;
}

struct BMArgs {
  BArgs b;
  int32_t accumulate;  // add to scores[] (second and later query passes)
};

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

// Variant 2: the same MFMA with 8 instead of 16 VALU ops per tile (4 operand masks + 4 v_max3).  Lane (r, g) takes dword g
// of patch row r (ds_read_b32) and feeds its 32 bits as the lane's 32 FP4 K-slots WITHOUT moving any of them:
//     P_0 = w & 0x11111111  -> nibble 0x1 = 0.5      query slot +-2.0   (product +-1)
//     P_1 = w & 0x22222222  -> nibble 0x2 = 1.0      query slot +-1.0
//     P_2 = w & 0x44444444  -> nibble 0x4 = 2.0      query slot +-0.5
//     P_3 = (w & 0x88888888) | 0x11111111  -> -+0.5 (bit 3 of a nibble is the FP4 SIGN)   query slot -+1.0, plus a
//           query-only constant folded into qpop[] (see expand() below)
// so D[d][q] = sum_k (2 q_k - 1) d_k (+ const_q) = 2 popc(q & d) - popc(d)  and  hamming = popc(q) - D.
// The patch rows are the MFMA's A operand: a lane's four results are four patches of ONE query column, so the running
// maximum is one register per query tile (two v_max3 per MFMA) and a page ends with two cross-group maxima.
// Measured (1 M pages, DMA-only run of the same kernel = 6.9 TB/s): 6.7 TB/s; 2 / 4 KiB slots, 2..8 consecutive pages
// per wave and block-interleaved slots were all tried and are no faster.
template <int MT, int D, bool PK = false>
__global__ __launch_bounds__(256) void maxsim_binary_mfma2_kernel(BMArgs args) {
  const BArgs& a = args.b;
  constexpr int SL = 1, SLB = kBinSlotBytes, SLR = kBinSlotRows;  // a slot = one DMA instruction (1 KiB); 2 / 4 KiB slots measured no better
  __shared__ __attribute__((aligned(16))) char lds[4 * D * SLB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= a.n) return;
  const int64_t page = a.cand ? (int64_t)a.cand[item] : item;
  if (masked(a, page)) {
    if (lane == 0) a.scores[item] = -INFINITY;
    return;
  }
  int64_t pk_row0 = 0;  // packed layout: loaded beside the row count (one s_waitcnt for both)
  if constexpr (PK) pk_row0 = a.row_off[page];
  const int nr = PK ? a.n_rows[page] : (a.n_rows ? a.n_rows[page] : a.stride);
  if (nr <= 0 || a.n_q <= 0) {
    if (lane == 0 && !args.accumulate) a.scores[item] = 0.0f;
    return;
  }
  const int ntiles = (nr + 15) >> 4;
  const int nslots = (nr + SLR - 1) / SLR;
  const char* pbase = reinterpret_cast<const char*>(a.bits) + (PK ? (size_t)pk_row0 : (size_t)page * (size_t)a.stride) * kSignBytes;
  char* ring = lds + wave * (D * SLB);
  const int src_off = lane * 16;
  const int rd_off = r * kSignBytes + g * 4;

  auto issue = [&](int it) {
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const char* tp = pbase + (size_t)it * SLB + j * kBinSlotBytes;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane(
          (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (it % D) * SLB + j * kBinSlotBytes));
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %3 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src_off), "s"(slot), "s"(tpu)
          : "memory");
    }
  };

#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (i < nslots) issue(i);

  // Operand roles: A = 16 patch rows (masked in place, see above), B = 16 query rows.  D[patch][query]: lane (c, g)
  // holds patches g*4 + i (i = 0..3) of query column c, so a lane's four results all fold into ONE running maximum
  // per query tile, and the page finish is two cross-group maxima instead of a 16-lane reduction per register.
  i32x8 qb[MT];
  float qpop[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const uint32_t w = reinterpret_cast<const uint32_t*>(a.q + m * 16 + r)[g];
    qb[m][0] = (int)(0xCCCCCCCCu - ((w & 0x11111111u) << 3));         // +-2.0
    qb[m][1] = (int)(0xAAAAAAAAu - (((w >> 1) & 0x11111111u) << 3));  // +-1.0
    qb[m][2] = (int)(0x99999999u - (((w >> 2) & 0x11111111u) << 3));  // +-0.5
    qb[m][3] = (int)(0x22222222u + (((w >> 3) & 0x11111111u) << 3));  // -+1.0 (the sign-slot class, see expand())
#pragma unroll
    for (int i = 4; i < 8; ++i) qb[m][i] = 0;
    qpop[m] = a.qpop[m * 16 + r];
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(qb[m][i]));
    asm volatile("" : "+v"(qpop[m]));
  }

  float mx[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mx[m] = -INFINITY;

  uint32_t ones = 0x11111111u;
  asm volatile("" : "+v"(ones));
  // Bits 0..2 of every nibble stay where they are (one AND each).  Bit 3 is the FP4 sign: instead of moving it, the
  // slot is given a constant magnitude -- (w & 0x8...) | 0x1... = -+0.5 -- and multiplied by -+1.0 on the query side:
  // (1 - 2q)(1 - 2d)/2 = (2q - 1) d + (1 - 2q)/2; the query-only term sums to 16 - popc(q & 0x88888888...) per row and is
  // folded into qpop[] by binary_qprep_kernel (signslot = 1).  One v_and_or_b32 instead of shift + and.
  auto expand = [&](uint32_t w) {
    i32x8 b;
    b[0] = (int)(w & 0x11111111u);
    b[1] = (int)(w & 0x22222222u);
    b[2] = (int)(w & 0x44444444u);
    // v_and_or_b32 is VOP3: no literals on gfx9 and one SGPR at most, so the second mask lives in a VGPR
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(b[3]) : "v"(w), "s"(0x88888888u), "v"(ones));
    b[4] = 0; b[5] = 0; b[6] = 0; b[7] = 0;
    return b;
  };
  auto mma = [&](const i32x8& pa, const i32x8& qbm) {
    f32x4b acc = {0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pa, qbm, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  };

  // Main loop: full slots only (no row masks, no divergent register assignment); a ragged tail slot is handled after
  // the loop, when every DMA has landed.
  const int nfull = nr / SLR;
  for (int it = 0; it < nfull; ++it) {
    if (it + D - 1 < nslots) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: last reads of the slot being refilled
      issue(it + D - 1);
      bin_wait_vmcnt<SL * (D - 1)>();
    } else {
      const int left = nslots - 1 - it;  // slots still allowed in flight (conservative ladder)
      if (left >= 4 && D > 4) bin_wait_vmcnt<SL * 4>();
      else if (left >= 2) bin_wait_vmcnt<SL * 2>();
      else if (left == 1) bin_wait_vmcnt<SL>();
      else bin_wait_vmcnt<0>();
    }
#pragma unroll
    for (int sub = 0; sub < SL; ++sub) {
      const char* slot = ring + (it % D) * SLB + sub * kBinSlotBytes + rd_off;
      uint32_t w[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) w[tt] = *reinterpret_cast<const uint32_t*>(slot + tt * 256);
#pragma unroll
      for (int tp = 0; tp < 4; tp += 2) {
        const i32x8 b0 = expand(w[tp]), b1 = expand(w[tp + 1]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4b c0 = mma(b0, qb[m]), c1 = mma(b1, qb[m]);
          const float t0 = fmaxf(fmaxf(c0[0], c0[1]), c0[2]);
          const float t1 = fmaxf(fmaxf(c0[3], c1[0]), c1[1]);
          const float t2 = fmaxf(fmaxf(c1[2], c1[3]), mx[m]);
          mx[m] = fmaxf(fmaxf(t0, t1), t2);
        }
      }
    }
  }
  if (nfull < nslots) {  // ragged tail: the tiles of the last slot, the final tile possibly partial
    bin_wait_vmcnt<0>();
    const char* slot = ring + (nfull % D) * SLB + rd_off;
#pragma unroll
    for (int tt = 0; tt < 4 * SL; ++tt) {
      const int t = nfull * 4 * SL + tt;
      if (t < ntiles) {  // wave-uniform
        const i32x8 b = expand(*reinterpret_cast<const uint32_t*>(slot + tt * 256));
        const int row0 = t * 16 + g * 4;  // this lane's four patch rows
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4b c = mma(b, qb[m]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (row0 + i < nr) mx[m] = fmaxf(mx[m], c[i]);
        }
      }
    }
  }

  float ham = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float v = mx[m];
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    if (qpop[m] >= 0.f) ham += qpop[m] - v;
  }
  ham = bin_group16_sum(ham);  // over the 16 query columns of the row (exact: small integers)
  if (lane == 0) {
    const float part = (float)a.n_q - ham * (1.0f / 128.0f);
    a.scores[item] = args.accumulate ? a.scores[item] + part : part;
  }
}

// Round-2 forms that lost by measurement and were removed in round 5 (records: profiles/r2, DESIGN.md "Sign-bit scan"):
//   variant 1  the first MFMA form (16 VALU ops per tile)                         345 M pages/s against 408 for variant 4
//   variant 5  persistent waves, one continuous DMA stream across pages           slower than fresh workgroups (as for the float scan)
//   variant 6  four-page burst per workgroup (64 KiB contiguous)                  6.29-6.31 TB/s against 6.56-6.61 for variant 4
//   variants 2 / 3  the ring of variant 4 with 8 / 16 slots                       390 / 295 M pages/s
// Round 5 built one more and removed it: the same arithmetic over PAGE QUARTERS (sixteen pages per fresh workgroup, wave w streaming the w-th
// 4 KiB of every page, one barrier per workgroup joining the quarters) -- the shape the ring transport streams 8 % faster than this kernel's
// (7.05 against 6.51 TB/s with no consumer, profiles/r5/stream_structure_probe_sweep3_*).  Bit-identical scores, 5.3-5.5 TB/s against 6.8
// (profiles/r5/sign_bit_page_quarter_form_experiment_r5m.jsonl): this scan is bound by issue slots (VALU 69 %, MFMA 41 % busy), and the ring
// and partial-sum LDS of the quarter form leave a CU 8 waves where this kernel has 32.

// popc(q row) as float for rows < n_q, -1 for the padding rows up to `padded`; also zero-fills the padding bit rows
__global__ void binary_qprep_kernel(uint4* qbits, int n_q, int padded, float* qpop, int signslot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  if (i < n_q) {
    const uint4 v = qbits[i];
    int pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    // variants 2..4: the sign-slot class contributes a query-only constant (see maxsim_binary_mfma2_kernel::expand)
    if (signslot) pc += 16 - (__popc(v.x & 0x88888888u) + __popc(v.y & 0x88888888u) + __popc(v.z & 0x88888888u) + __popc(v.w & 0x88888888u));
    qpop[i] = (float)pc;
  } else {
    qbits[i] = make_uint4(0u, 0u, 0u, 0u);
    qpop[i] = -1.0f;
  }
}

template <int MT>
static void launch_binary_mfma(const BArgs& k, int accumulate, hipStream_t s) {
  BMArgs m{k, accumulate};
  if (k.row_off) hipLaunchKernelGGL((maxsim_binary_mfma2_kernel<MT, 4, true>), dim3((unsigned)((k.n + 3) / 4)), dim3(256), 0, s, m);
  else hipLaunchKernelGGL((maxsim_binary_mfma2_kernel<MT, 4, false>), dim3((unsigned)((k.n + 3) / 4)), dim3(256), 0, s, m);
}

// variant: 0 = popcount on the VALU (the independent cross-check), 4 (default, -1) = FP4 MFMA, 4-slot ring -- 408 M pages/s at
// 1 M pages against 48 for the popcount form
int launch_maxsim_binary(const BinaryArgs& a, int variant, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  if (a.row_off && !a.n_rows) { set_error("sign-bit scan: a row-offset table needs the per-page row counts"); return MV_ERR_INVALID; }
  BArgs k{reinterpret_cast<const uint4*>(a.bits), a.n_rows, a.doc_ord, a.allow, a.n_allow_bits,
          reinterpret_cast<const uint4*>(a.qbits), a.scores, a.n, a.stride, a.n_q, a.qpop, a.cand, a.row_off};
  if (a.n > ((int64_t)1 << 25)) { set_error("binary scan: more than 2^25 pages per launch is not supported"); return MV_ERR_INVALID; }
  if (variant < 0) variant = 4;
  if (variant == 0 || a.n_q <= 0) {
    if (k.row_off) hipLaunchKernelGGL(maxsim_binary_kernel<true>, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, k);
    else hipLaunchKernelGGL(maxsim_binary_kernel<false>, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, k);
  } else if (variant == 4) {
    if (!a.qpop) { set_error("binary MFMA scan needs the qpop workspace"); return MV_ERR_INVALID; }
    const int padded = ((a.n_q + 15) / 16) * 16;
    hipLaunchKernelGGL(binary_qprep_kernel, dim3((unsigned)((padded + 63) / 64)), dim3(64), 0, s,
                       reinterpret_cast<uint4*>(const_cast<uint8_t*>(a.qbits)), a.n_q, padded, a.qpop_rw, 1);
    // query rows in passes of <= 64 (4 MFMA row tiles); later passes accumulate into scores[]
    for (int q0 = 0, pass = 0; q0 < a.n_q; q0 += 64, ++pass) {
      BArgs kp = k;
      kp.q = k.q + q0;
      kp.qpop = k.qpop + q0;
      kp.n_q = std::min(64, a.n_q - q0);
      const int mt = (kp.n_q + 15) / 16;
      switch (mt) {
        case 1: launch_binary_mfma<1>(kp, pass > 0, s); break;
        case 2: launch_binary_mfma<2>(kp, pass > 0, s); break;
        case 3: launch_binary_mfma<3>(kp, pass > 0, s); break;
        default: launch_binary_mfma<4>(kp, pass > 0, s); break;
      }
    }
  } else {
    set_error("unknown binary variant %d (0 = popcount, 4 = FP4 MFMA)", variant);
    return MV_ERR_INVALID;
  }
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
