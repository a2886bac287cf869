// mv_fde_batch.hip -- the BATCHED coarse pass over the FDE slab: up to 32 requests per read of the slab (bf16 MFMA).
//
// Split out of mv_fde.hip in round 5 (that file keeps the tables, the encoders and the single-query scan).  Replaces, for a serving
// process that batches concurrent requests, the per-request ANN query of fast_multivector_store.py:526-532.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "mv_common.h"
#include "mv_e4m3.h"

namespace mv {
namespace {

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// ------------------------------------------------------------------------------ batched coarse scan
// Up to 32 queries per pass over the FDE slab: the scan above reads 2 * out_dim bytes per page for ONE dot product, so
// under load the coarse stage is a GEMM  S[pages x 16] = F[pages x out_dim] . Qf^T  that is still HBM-bound (16 queries:
// ~16 flop/byte against a ~300 flop/byte ridge) -- sixteen searches for the slab traffic of one.
//
// Workgroup = 4 waves, page tile = 64 pages, ring slot = 64 pages x 256 dims (512 B per page, 32 KiB), 4 slots.
//   * transport: every wave DMAs a quarter of each slot (global_load_lds_dwordx4 nt, 2 pages x 512 B per instruction),
//     counted s_waitcnt, one barrier per slot -- three slots (96 KiB per CU) are in flight while one is consumed;
//   * arithmetic: v_mfma_f32_16x16x32_bf16, A = 16 pages x 32 dims from LDS (XOR-swizzled 16-byte chunks, conflict-free
//     ds_read_b128), B = 32 dims x 16 queries.  Wave w owns dims [64w, 64w+64) of every slot for all four page tiles
//     (K-split): its query fragments are 4 coalesced 1 KiB loads per slot from a fragment-ordered image in L2, issued
//     three slots ahead into a static 4-set register ring, and counted with the DMAs (all VMEM of the loop is inline
//     asm: the compiler's own vmcnt bookkeeping would drain the ring);
//   * the fp32 query FDE enters as bf16 hi + bf16 lo (two MFMAs per fragment): 16 mantissa bits, so the coarse scores
//     agree with the fp32-query scan above to ~1e-5 relative; the slab is bf16 either way;
//   * tile end: the four waves' partial sums meet in LDS and are added in a fixed order (deterministic), 64 x 16
//     scores leave as 256-byte rows.  Cosine rule / tombstones: in the tile epilogue of the FIN instantiations (metadata through
//     the DMA ring: a plain global load in this loop would make the compiler drain it), else by fde_batch_finish_kernel.
struct ScanBatchArgs {
  const char* fde;     // [n][out_dim] bf16
  const char* qfrag;   // fragment-ordered hi/lo image of the queries (fde_batch_qprep_kernel)
  float* scores;       // [n_queries][score_stride]
  int64_t score_stride;
  int64_t n;
  int32_t out_dim;
  int32_t n_queries;
  int32_t n_tiles;     // ceil(n / 64)
  const float* inv_norm;    // FIN kernels: the cosine rule and the tombstones are applied where the scores are written
  const int32_t* doc_ord;   // nullable (no tombstones)
  const float* qfac;        // E4 kernels: 32 per-query factors 2^-e (the scale the query's e4m3 terms were taken at); else unused
};

constexpr int kFbPages = 64;
constexpr int kFbSlotBytes = kFbPages * 512;
constexpr int kFbSlots = 4;
constexpr int kFbRedStride = 68;  // floats per (wave, query) row of the tile-end reduction: 64 pages + 4 (16-byte skew)

// [nb][out_dim] fp32 -> image[kc][wave][e][query tile][hi|lo][lane] of 16-byte B fragments: lane (query qt*16 + (l&15),
// group l>>4) holds dims kc*256 + (2*wave + e)*32 + 8*(l>>4) .. +8 of its query; queries >= nb are zero.
__global__ __launch_bounds__(256) void fde_batch_qprep_kernel(const float* q, int nb, int out_dim, int nqt, uint16_t* image) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (kc, wave, e, qt, lane)
  const int lane = t & 63;
  int r = t >> 6;
  const int qt = r % nqt; r /= nqt;
  const int e = r & 1, w = (r >> 1) & 3, kc = r >> 3;
  if (kc * 256 >= out_dim) return;
  const int ql = qt * 16 + (lane & 15), g = lane >> 4;
  const int d0 = kc * 256 + (2 * w + e) * 32 + g * 8;
  uint16_t hi[8], lo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = ql < nb ? q[(size_t)ql * out_dim + d0 + i] : 0.0f;
    hi[i] = f32_to_bf16_rne(x);
    lo[i] = f32_to_bf16_rne(x - bf16_to_f32(hi[i]));
  }
  uint16_t* dst = image + ((size_t)(((kc * 4 + w) * 2 + e) * nqt + qt) * 2) * 512 + lane * 8;  // 512 bf16 = 1 KiB per fragment
#pragma unroll
  for (int i = 0; i < 8; ++i) { dst[i] = hi[i]; dst[512 + i] = lo[i]; }
}

// ---- the e4m3 form of the pass (MV_WITH_FDE_E4M3: the slab's e4m3 copy, 512 codes per page and ring slot)
// Queries enter as TWO e4m3 terms under one power-of-two scale per query (amax -> 448, as the rows of the copy): hi = e4m3(x 2^e),
// lo = e4m3(x 2^e - hi) -- the rounding residual, at most 2^-4 of its value's binade, keeps three more mantissa bits wherever it stays
// above e4m3's subnormal floor 2^-9, i.e. for every |x 2^e| >= 2^-5: the entries that carry the dot product.  Both terms accumulate in
// the same fp32 register.  qfac[b] = 2^-e (1 for an all-zero query; 0 beyond nb).
__global__ __launch_bounds__(256) void fde_batch_qscale8_kernel(const float* q, int nb, int out_dim, float* qfac) {
  __shared__ uint32_t red[4];
  const int b = blockIdx.x;
  if (b >= nb) { if (threadIdx.x == 0) qfac[b] = 0.0f; return; }
  uint32_t amax = 0;
  for (int i = threadIdx.x; i < out_dim; i += 256) amax = max(amax, __float_as_uint(q[(size_t)b * out_dim + i]) & 0x7fffffffu);
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, sft));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = max(max(red[0], red[1]), max(red[2], red[3]));
    if (amax > 0x7f7fffffu) amax = 0x7f7fffffu;
    qfac[b] = pow2f(-pow2_scale_exp(amax));
  }
}
// image8[kc][wave][query tile][hi.0 | lo.0 | hi.1 | lo.1][lane] of 16-byte pieces (the order of the bf16 image: a one-term pass loads pieces 0 and 2): lane (query qt*16 + (l&15), group l>>4) holds the codes of
// dims kc*512 + wave*128 + 32*(l>>4) .. +32 of its query -- .0 the first sixteen, .1 the second (four 8-byte MFMA operands per term).
__global__ __launch_bounds__(256) void fde_batch_qprep8_kernel(const float* q, const float* qfac, int nb, int out_dim, int nqt, uint8_t* image) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (kc, wave, qt, lane)
  const int lane = t & 63;
  int r = t >> 6;
  const int qt = r % nqt; r /= nqt;
  const int w = r & 3, kc = r >> 2;
  if (kc * 512 >= out_dim) return;
  const int ql = qt * 16 + (lane & 15), g = lane >> 4;
  const int d0 = kc * 512 + w * 128 + g * 32;
  const float fac = ql < nb ? qfac[ql] : 0.0f;
  const float sc = fac > 0.0f ? 1.0f / fac : 0.0f;  // a power of two: exact
  uint32_t hi[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lo[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float x = ql < nb ? q[(size_t)ql * out_dim + d0 + i] * sc : 0.0f;
    const uint32_t ch = e4m3_encode(x);
    const uint32_t cl = e4m3_encode(x - e4m3_decode(ch));  // the residual in its own binade (no common 2^-4: the K = 32 MFMA has no scale operand)
    hi[i >> 2] |= ch << (8 * (i & 3));
    lo[i >> 2] |= cl << (8 * (i & 3));
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(image + ((size_t)((kc * 4 + w) * nqt + qt) * 4) * 1024 + lane * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) { dst[i] = hi[i]; dst[256 + i] = lo[i]; dst[512 + i] = hi[4 + i]; dst[768 + i] = lo[4 + i]; }
}

// ---- the FP4 form of the pass (MV_WITH_FDE_FP4: the slab's e2m1 copy, 1024 codes per page and ring slot).  BOTH MFMA operands
// are FP4 (v_mfma_scale_f32_16x16x128_f8f6f4, cbsz = blgp = 4: a lane's 16-byte ring register IS one A operand, a 16-byte fragment one B
// operand).  Queries enter as TWO e2m1 terms under one power-of-two scale per query (the smallest s with 6 s >= max|x|): hi = fp4(x / s),
// lo = fp4(4 (x / s - hi)) -- the second MFMA takes the factor 1/4 as its block scale (E8M0 125).  Priced before it was built
// (tools/fde_4bit_recall_probe.py: recall with two-term fp4 queries = recall with fp32 queries on the fp4 documents).  qfac[b] = s.
__global__ __launch_bounds__(256) void fde_batch_qscale4_kernel(const float* q, int nb, int out_dim, float* qfac) {
  __shared__ uint32_t red[4];
  const int b = blockIdx.x;
  if (b >= nb) { if (threadIdx.x == 0) qfac[b] = 0.0f; return; }
  uint32_t amax = 0;
  for (int i = threadIdx.x; i < out_dim; i += 256) amax = max(amax, __float_as_uint(q[(size_t)b * out_dim + i]) & 0x7fffffffu);
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, sft));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = max(max(red[0], red[1]), max(red[2], red[3]));
    int e = 0;
    if (amax >= 0x00800000u && amax < 0x7f800000u) {  // a normal fp32 value: 6 * 2^(e0-2) = 1.5 * 2^e0
      const int e0 = (int)(amax >> 23) - 127;
      e = ((amax & 0x7fffffu) <= 0x400000u) ? e0 - 2 : e0 - 1;
      e = min(max(e, -100), 100);
    }
    qfac[b] = pow2f(e);
  }
}
// image4[kc][wave][query tile][hi.0 | lo.0 | hi.1 | lo.1][lane] of 16-byte pieces: lane (query qt*16 + (l&15), group g = l>>4) holds, in piece e,
// the 32 codes of dims kc*1024 + 32 (wave*8 + 2 g + e) .. + 32 of its query, two per byte, the lower dim in the LOW nibble -- the order of the
// slab's bytes, so K slot j of the page operand meets K slot j of the query operand.  Dims >= out_dim (the chunks that pad KC to a multiple of
// four) are zero: whatever the ring reads there is multiplied by 0 (every FP4 code is finite).
__global__ __launch_bounds__(256) void fde_batch_qprep4_kernel(const float* q, const float* qfac, int nb, int out_dim, int kc_padded, int nqt, uint8_t* image) {
  const int t = blockIdx.x * 256 + threadIdx.x;  // (kc, wave, qt, lane)
  const int lane = t & 63;
  int r = t >> 6;
  const int qt = r % nqt; r /= nqt;
  const int w = r & 3, kc = r >> 2;
  if (kc >= kc_padded) return;
  const int ql = qt * 16 + (lane & 15), g = lane >> 4;
  const float fac = ql < nb ? qfac[ql] : 0.0f;
  const float sc = fac > 0.0f ? 1.0f / fac : 0.0f;  // a power of two: exact
  uint32_t* dst = reinterpret_cast<uint32_t*>(image + ((size_t)((kc * 4 + w) * nqt + qt) * 4) * 1024 + lane * 16);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int d0 = kc * 1024 + 32 * (w * 8 + 2 * g + e);
    uint32_t hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float x = (ql < nb && d0 + i < out_dim) ? q[(size_t)ql * out_dim + d0 + i] * sc : 0.0f;
      const uint32_t ch = fp4_encode(x);
      const uint32_t cl = fp4_encode((x - fp4_decode(ch)) * 4.0f);
      hi[i >> 3] |= ch << (4 * (i & 7));
      lo[i >> 3] |= cl << (4 * (i & 7));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { dst[(2 * e) * 256 + i] = hi[i]; dst[(2 * e + 1) * 256 + i] = lo[i]; }
  }
}

// NQT query tiles of 16 (16 or 32 queries per pass).
template <int NQT, bool LO = true>
__global__ __launch_bounds__(256) void fde_scan_batch_kernel(ScanBatchArgs a) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  constexpr int NF = 4 * NQT;       // query fragments per slot and wave: (e, qt, hi|lo)
  constexpr int OPS = 8 + (LO ? NF : NF / 2);  // VMEM operations per slot and wave
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[kFbSlots * kFbSlotBytes + 4 * 16 * kFbRedStride * 4];
  float* red = reinterpret_cast<float*>(lds + kFbSlots * kFbSlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int KC = a.out_dim >> 8;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles)
  const int total = n_my * KC;                   // ring slots of this workgroup; KC % 4 == 0
  const uint32_t row_bytes = (uint32_t)a.out_dim * 2u;

  // DMA source offsets: instruction i of this wave fills LDS bytes [(wave*8 + i) KiB, +1 KiB) of the slot = pages
  // pl, pl+1 (pl = wave*16 + 2i); lane -> page pl + (lane>>5), chunk position lane&31, which receives the page's
  // logical chunk (lane&31) ^ (page & 15).  The scalar base is 4 KiB below the tile so the offsets (which also carry
  // -1 KiB per instruction of a group of four, see issue()) stay positive.
  uint32_t src_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
    src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
  }
  const uint32_t q_off = (uint32_t)lane * 16u;

  bf16x8 qf[4][NF];  // [ring set][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

  int i_tile = b, i_kc = 0;  // issue-side position (advanced by issue_q: a slot's DMAs come first, its fragments second)
  auto issue_dma = [&](int slot_idx) {
    const int64_t page0 = (int64_t)i_tile * kFbPages;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * kFbSlotBytes + wave * 8192));
    uint32_t so[8];
    if (page0 + kFbPages > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
        so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %11 nt\n\t"
        "global_load_lds_dwordx4 %2, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %3, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %4, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %5, %11 nt\n\t"
        "global_load_lds_dwordx4 %6, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %7, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %8, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[4]), "v"(so[5]), "v"(so[6]), "v"(so[7]), "s"(slot),
          "s"(slot + 4096u), "s"(tpu)
        : "memory");
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // this wave's query fragments of the slot: NF KiB contiguous in the image, four per statement
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else  // bf16 query FDE (hi term only)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
    if (++i_kc == KC) { i_kc = 0; i_tile += G; }
  };

  // fragment read offsets inside a slot: page tile t, k-step kk = 2*wave + e -> page t*16 + p, logical chunk kk*4 + g
  uint32_t rd_off[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) rd_off[e] = (uint32_t)p * 512u + (((uint32_t)((2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);

  f32x4 acc[4][NQT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) acc[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Four slots in flight.  Per slot: counted wait + barrier (the slot has landed for all waves) -> every wave pulls
  // its fragments into registers -> barrier (the slot is drained) -> its refill is issued AT ONCE, before the MFMAs:
  // a ring position idles for one LDS read, not for a slot's arithmetic plus the wait for the next slot's data.
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u < total) {
      issue_dma(u);
      issue_q(qf[u]);
    }
  }

  int c_tile = b, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u;
      // OPS VMEM operations per slot and wave (8 DMAs + the fragment loads), completed in issue order
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * OPS) : "memory");
      else if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int j = 0; j < NF; ++j) asm volatile("" : "+v"(qf[u][j]));  // uses stay behind the wait
      const char* slot = lds + u * kFbSlotBytes;
      bf16x8 af[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (16 * 512) + rd_off[e]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + 4 < total) issue_dma(u);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[u][(e * NQT + qt) * 2 + 0], acc[t][qt], 0, 0, 0);
            if (LO) acc[t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[u][(e * NQT + qt) * 2 + 1], acc[t][qt], 0, 0, 0);
          }
      if (s + 4 < total) {
        // the register set is free after its last MFMA was issued; at most 63 VMEM operations may be outstanding
        if (4 * OPS > 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(63 - (OPS - 8)) : "memory");
        issue_q(qf[u]);
      }
      if (++c_kc == KC) {  // tile done: acc[t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p over this wave's dims
        const int pg = threadIdx.x & 63;
        const int64_t page = (int64_t)c_tile * kFbPages + pg;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * kFbRedStride + t * 16 + g * 4) = acc[t][qt];
            acc[t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ql = (threadIdx.x >> 6) + 4 * j;
              if (qt * 16 + ql < a.n_queries) {
                const float v = (red[(0 * 16 + ql) * kFbRedStride + pg] + red[(1 * 16 + ql) * kFbRedStride + pg]) +
                                (red[(2 * 16 + ql) * kFbRedStride + pg] + red[(3 * 16 + ql) * kFbRedStride + pg]);
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten KC slots (>= 4 barriers) later
        c_kc = 0;
        c_tile += G;
      }
    }
  }
}

// ---- the same scan with page tiles processed in PAIRS per query fragment (default form)
// A wave's query fragments cost as much L2 -> register traffic per slot as the slot's share of the pages costs HBM traffic
// once 32 queries ride a pass (measured: 16 KiB of fragments per 32 KiB slot: 6.5 TB/s, 32 KiB: 5.7).  Here a workgroup
// walks TWO of its page tiles together, slot order (kc, tile 0), (kc, tile 1), (kc+1, tile 0), ...: the fragments of a
// K chunk are loaded once and used for both tiles -- half the fragment traffic, twice the accumulators.  A workgroup's
// odd tile out runs through the single-tile phase afterwards (same code, T = 1), so the tile -> workgroup map and the
// per-page arithmetic (order of the K chunks, of the four waves' partial sums) are those of the single-tile kernel:
// identical scores.
template <typename F, int... I>
__device__ __forceinline__ void fb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void fb_static_for(F&& f) {
  fb_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// n_groups groups of T tiles: tiles i0 + grp*T + j of this workgroup's list b, b + G, b + 2G, ...
// FIN: the finish of the pass (scores *= 1/|d|, -inf for tombstoned pages) happens where a tile's scores are written, instead of in
// a second pass over the [queries][pages] matrix (fde_batch_finish_kernel: 54 us of a 4.0 ms pass at 1.25 M pages x 32 queries).
// A tile's 64 norms and 64 document ordinals travel like its pages: two global_load_lds_dword per wave into a 512-byte LDS
// record, issued with the tile's slots of every FOURTH K chunk (kc & 3 == 0 is a compile-time property of the unrolled slot, so the
// counted vmcnt waits stay constants; the re-loads bring the same 512 bytes, +0.4 % requests), landed -- in issue order -- before that
// slot's own wait returns, read with ds_read at the tile's end.  Every wave issues them (same data, same place: the per-wave
// counts stay uniform).  Records are double-buffered by group parity: the next group's first slots are issued before this group's
// epilogue runs.  The arithmetic is the finish kernel's (one fp32 multiply of the same sum): identical scores.
template <int NQT, bool LO, int T, bool FIN = false, bool E4 = false, bool F4 = false>
__device__ __forceinline__ void fb_phase(const ScanBatchArgs& a, char* lds, float* red, const int lane, const int wave, const int i0,
                                         const int n_groups, const uint32_t (&src_off)[8], const uint32_t (&rd_off)[2]) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  static_assert(!E4 || FIN, "the e4m3 form applies the page factor itself");
  static_assert(!F4 || E4, "the fp4 form is the e4m3 form with other codes: page factor, query factor and tombstones as there");
  constexpr int NF = 4 * NQT;              // query fragments per K chunk and wave: (e, qt, hi|lo); E4: (qt, half, hi|lo)
  constexpr int QOPS = LO ? NF : NF / 2;   // fragment loads per K chunk and wave
  constexpr int MOPS = FIN ? 2 : 0;        // metadata loads per slot of a K chunk with kc & 3 == 0
  char* meta = reinterpret_cast<char*>(red) + 4 * 16 * kFbRedStride * 4;  // FIN: [group parity][tile of the group][64 x 1/|d| | 64 x doc ordinal]
  if constexpr (FIN) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the previous phase's last epilogue has read its records
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  // E4: a slot holds 512 codes per page; F4: 1024, and the chunk count is padded to a multiple of four (the padding chunks meet zero query fragments)
  const int KC = F4 ? (((a.out_dim >> 10) + 3) & ~3) : (E4 ? a.out_dim >> 9 : a.out_dim >> 8);
  const int total = n_groups * KC * T;     // ring slots of this phase; a multiple of 4 * T
  if (total == 0) return;
  const uint32_t row_bytes = F4 ? (uint32_t)a.out_dim >> 1 : (E4 ? (uint32_t)a.out_dim : (uint32_t)a.out_dim * 2u);
  // E4: this thread's per-query factors for the tile ends (queries qt*16 + wave + 4x), loaded before the ring starts
  float qfac[E4 ? NQT * 4 : 1];
  if constexpr (E4) {
#pragma unroll
    for (int x = 0; x < NQT * 4; ++x) qfac[x] = a.qfac[(x >> 2) * 16 + (threadIdx.x >> 6) + 4 * (x & 3)];
#pragma unroll
    for (int x = 0; x < NQT * 4; ++x) asm volatile("" : "+v"(qfac[x]));
  }
  const uint32_t q_off = (uint32_t)lane * 16u;
  uint32_t f4_one = 0x7f7f7f7fu, f4_quarter = 0x7d7d7d7du;  // E8M0 block scales of the F4 MFMAs: 2^0, 2^-2
  asm volatile("" : "+v"(f4_one), "+v"(f4_quarter));

  bf16x8 qf[4][NF];  // [K chunk & 3][(e*NQT + qt)*2 + (0 = hi, 1 = lo)]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < NF; ++j) qf[u][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[T][4][NQT];
#pragma unroll
  for (int j = 0; j < T; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int i_grp = 0, i_kc = 0, i_j = 0;  // issue-side position: (group, K chunk, tile of the group)
  auto issue_dma = [&](int slot_idx) {
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * kFbPages;
    const char* tp = a.fde + (size_t)page0 * row_bytes + (size_t)i_kc * 512 - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * kFbSlotBytes + wave * 8192));
    uint32_t so[8];
    if (page0 + kFbPages > a.n) {  // last tile: rows past the corpus re-read its last page (their sums are never written)
      const uint32_t last = (uint32_t)(a.n - 1 - page0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
        so[i] = min(pl, last) * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) so[i] = src_off[i];
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %9\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %11 nt\n\t"
        "global_load_lds_dwordx4 %2, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %3, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %4, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %10\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %5, %11 nt\n\t"
        "global_load_lds_dwordx4 %6, %11 offset:1024 nt\n\t"
        "global_load_lds_dwordx4 %7, %11 offset:2048 nt\n\t"
        "global_load_lds_dwordx4 %8, %11 offset:3072 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[4]), "v"(so[5]), "v"(so[6]), "v"(so[7]), "s"(slot),
          "s"(slot + 4096u), "s"(tpu)
        : "memory");
  };
  auto issue_q = [&](bf16x8 (&qs)[NF]) {  // the fragments of K chunk i_kc
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
      const char* qp = a.qfrag + (size_t)(i_kc * 4 + wave) * (NF * 1024) + h * 4096;
      const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)qp);
      const uint32_t qhi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)qp >> 32));
      const uint64_t qpu = ((uint64_t)qhi << 32) | qlo;
      if (LO)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 1]), "+v"(qs[4 * h + 2]), "+v"(qs[4 * h + 3])
            : "v"(q_off), "s"(qpu)
            : "memory");
      else
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %2, %3\n\t"
            "global_load_dwordx4 %1, %2, %3 offset:2048"
            : "+v"(qs[4 * h + 0]), "+v"(qs[4 * h + 2])
            : "v"(q_off), "s"(qpu)
            : "memory");
    }
  };
  auto issue_meta = [&]() {  // the slot being issued belongs to tile (i_grp, i_j): its norms and ordinals -> meta[i_grp & 1][i_j]
    const int64_t tile = (int64_t)b + (int64_t)(i0 + i_grp * T + i_j) * G;
    const int64_t page0 = tile * kFbPages;
    const float* ip = a.inv_norm + page0;
    const int32_t* op = (a.doc_ord ? a.doc_ord : reinterpret_cast<const int32_t*>(a.inv_norm)) + page0;
    const uint32_t ilo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ip);
    const uint32_t ihi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ip >> 32));
    const uint32_t olo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)op);
    const uint32_t ohi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)op >> 32));
    const uint64_t ipu = ((uint64_t)ihi << 32) | ilo, opu = ((uint64_t)ohi << 32) | olo;
    const uint32_t rec = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(meta + ((i_grp & 1) * T + i_j) * 512));
    uint32_t vo = (uint32_t)lane * 4u;
    if (page0 + kFbPages > a.n) vo = min((uint32_t)lane, (uint32_t)(a.n - 1 - page0)) * 4u;  // last tile: lanes past the corpus re-read its last page
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %1, %4\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dword %1, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(vo), "s"(rec), "s"(rec + 256u), "s"(ipu), "s"(opu)
        : "memory");
  };
  auto advance = [&]() {
    if (++i_j == T) {
      i_j = 0;
      if (++i_kc == KC) { i_kc = 0; ++i_grp; }
    }
  };

  // prologue: slots 0..3 (slot x = K chunk x / T, tile x % T); a chunk's fragments ride with its first slot
  fb_static_for<4>([&](auto UC) {
    constexpr int u = decltype(UC)::value;
    issue_dma(u);
    if constexpr (FIN && ((u / T) & 3) == 0) issue_meta();
    if constexpr (u % T == 0) issue_q(qf[(u / T) & 3]);
    advance();
  });

  int c_grp = 0, c_kc = 0;  // consume-side position
  for (int s0 = 0; s0 < total; s0 += 4 * T) {
    fb_static_for<4 * T>([&](auto UC) {
      constexpr int u = decltype(UC)::value;
      constexpr int j = u % T;            // tile of the group
      constexpr int kcs = (u / T) & 3;    // K chunk & 3 -> fragment register set
      const int s = s0 + u;
      // VMEM operations of the slots x behind this one (8 DMAs + the fragment loads of a chunk's first slot + the metadata loads of
      // the slots of every fourth chunk), in issue order
      constexpr int o1 = 8 + (((u + 1) % T == 0) ? QOPS : 0) + (((((u + 1) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      constexpr int o2 = 8 + (((u + 2) % T == 0) ? QOPS : 0) + (((((u + 2) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      constexpr int o3 = 8 + (((u + 3) % T == 0) ? QOPS : 0) + (((((u + 3) % (4 * T)) / T) & 3) == 0 ? MOPS : 0);
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2 + o3) : "memory");
      else if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1 + o2) : "memory");
      else if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(o1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s landed for all waves
#pragma unroll
      for (int x = 0; x < NF; ++x) asm volatile("" : "+v"(qf[kcs][x]));  // uses stay behind the wait
      const char* slot = lds + (u & 3) * kFbSlotBytes;
      bf16x8 af[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[e][t] = *reinterpret_cast<const bf16x8*>(slot + t * (16 * 512) + rd_off[e]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slot s is in registers everywhere: refill it
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(af[e][t]));  // the fragment reads stay in front of the barrier
      if (s + 4 < total) {
        issue_dma(u & 3);
        if constexpr (FIN && ((((u + 4) % (4 * T)) / T) & 3) == 0) issue_meta();
      }
      if constexpr (F4) {
        // ONE K = 128 FP4 x FP4 MFMA per (piece, page tile, query tile) and term, written as inline assembly: operands pinned to the VGPRs the ring
        // and the fragment loads wrote (the compiler's own allocation of the builtin's operands moved the fragment sets through AGPRs, and an
        // asynchronous fragment load that lands in a register the allocator has reused corrupts an address); the lo term carries its 1/4 as the B
        // block scale (E8M0 125 = 2^-2).  Accumulators in VGPRs; consecutive MFMAs on one accumulator forward srcC in hardware.
        // (the hi terms of all 4 NQT accumulators first, then the lo terms: a dependent pair is 4 NQT instructions apart, not back to back)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
              asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                           : "+v"(acc[j][t][qt]) : "v"(af[e][t]), "v"(qf[kcs][qt * 4 + 2 * e]), "v"(f4_one));
          if (LO) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int qt = 0; qt < NQT; ++qt)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                             : "+v"(acc[j][t][qt]) : "v"(af[e][t]), "v"(qf[kcs][qt * 4 + 2 * e + 1]), "v"(f4_one), "v"(f4_quarter));
          }
        }
      } else if constexpr (E4) {
        // four K = 32 fp8 MFMAs per (page tile, query tile) and term: every 16-byte piece a lane read / loaded is two 8-byte operands, used
        // in place (the K = 128 block-scaled form wants 32 consecutive bytes per lane: building them from the ring's 16-byte registers cost
        // copies and 200 spilled registers).  Any K-slot order is valid as long as A and B agree: operand jj of piece e = codes
        // 32 g + 16 e + 8 jj .. + 8 of the wave's 128.
        using i64x2 = __attribute__((ext_vector_type(2))) long;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const long a8 = __builtin_bit_cast(i64x2, af[e][t])[jj];
#pragma unroll
              for (int qt = 0; qt < NQT; ++qt) {
                acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, __builtin_bit_cast(i64x2, qf[kcs][qt * 4 + 2 * e])[jj], acc[j][t][qt], 0, 0, 0);
                if (LO) acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, __builtin_bit_cast(i64x2, qf[kcs][qt * 4 + 2 * e + 1])[jj], acc[j][t][qt], 0, 0, 0);
              }
            }
      } else {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 0], acc[j][t][qt], 0, 0, 0);
            if (LO) acc[j][t][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[e][t], qf[kcs][(e * NQT + qt) * 2 + 1], acc[j][t][qt], 0, 0, 0);
          }
      }
      if (s + 4 < total) {
        if constexpr (j == 0) {  // slot s + 4 opens K chunk (u + 4) / T: its fragments go into that chunk's register set
          // (T = 1: the set the MFMAs above just read; at most 63 VMEM operations may be outstanding)
          constexpr int peak = 4 * 8 + ((T == 1) ? 4 : (T == 2 ? 2 : 1)) * QOPS + MOPS * T;
          if constexpr (peak > 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(63 - QOPS) : "memory");
          issue_q(qf[((u + 4) / T) & 3]);
        }
        advance();
      }
      // A tile ends with its last K chunk, and KC is a multiple of 4 (fde_scan_batch_supported: out_dim % 1024 == 0) while every pass of
      // this unrolled body starts at a chunk index that is one too: the last chunk always sits in fragment set 3.  Saying so at compile
      // time keeps the finish out of the other 3 T slots -- round 6: the body was 38 / 55 KB of code (16 / 32 requests) in front of a
      // 64 KB instruction cache shared by two CUs, and the pass ran in a fast or a slow mode depending on the process.
      if constexpr (kcs == 3)
      if (c_kc == KC - 1) {
        if constexpr (F4) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the inline-assembly MFMAs above: their results are read below (the compiler does not see the latency)  // tile j of the group is done: acc[j][t][qt][i] = partial dot of page t*16 + 4g + i with query qt*16 + p
        const int pg = threadIdx.x & 63;
        const int64_t tile = (int64_t)b + (int64_t)(i0 + c_grp * T + j) * G;
        const int64_t page = tile * kFbPages + pg;
        float inv = 1.0f;
        bool dead = false;
        if constexpr (FIN) {  // the record of this tile landed KC slots ago
          const char* rec = meta + ((c_grp & 1) * T + j) * 512;
          inv = reinterpret_cast<const float*>(rec)[pg];
          if (a.doc_ord) dead = reinterpret_cast<const int32_t*>(rec + 256)[pg] < 0;
        }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          if (qt > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone has read the previous query tile's sums
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<f32x4*>(red + (wave * 16 + p) * kFbRedStride + t * 16 + g * 4) = acc[j][t][qt];
            acc[j][t][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          if (page < a.n) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const int ql = (threadIdx.x >> 6) + 4 * x;
              if (qt * 16 + ql < a.n_queries) {
                float v = (red[(0 * 16 + ql) * kFbRedStride + pg] + red[(1 * 16 + ql) * kFbRedStride + pg]) +
                          (red[(2 * 16 + ql) * kFbRedStride + pg] + red[(3 * 16 + ql) * kFbRedStride + pg]);
                if constexpr (FIN) v = dead ? -INFINITY : v * inv;
                if constexpr (E4) v *= qfac[qt * 4 + x];
                a.scores[(size_t)(qt * 16 + ql) * a.score_stride + page] = v;
              }
            }
          }
        }
        // red[] is rewritten by the next tile end: at least one slot barrier later
      }
      if constexpr (j == T - 1) {
        if (++c_kc == KC) { c_kc = 0; ++c_grp; }
      }
    });
  }
}

template <int NQT, bool LO, bool FIN = false, bool E4 = false, bool F4 = false>
__global__ __launch_bounds__(256) void fde_scan_batch2_kernel(ScanBatchArgs a) {
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[kFbSlots * kFbSlotBytes + 4 * 16 * kFbRedStride * 4 + (FIN ? 2 * 2 * 512 : 0)];
  float* red = reinterpret_cast<float*>(lds + kFbSlots * kFbSlotBytes);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = lane & 15, g = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int n_my = (a.n_tiles - b + G - 1) / G;  // tiles b, b + G, ...  (grid <= n_tiles)
  const uint32_t row_bytes = F4 ? (uint32_t)a.out_dim >> 1 : (E4 ? (uint32_t)a.out_dim : (uint32_t)a.out_dim * 2u);
  uint32_t src_off[8];  // see fde_scan_batch_kernel
  uint32_t rd_off[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t pl = (uint32_t)(wave * 16 + 2 * i + (lane >> 5));
    src_off[i] = pl * row_bytes + ((((uint32_t)lane & 31u) ^ (pl & 15u)) << 4) + 4096u - (uint32_t)(i & 3) * 1024u;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e)  // bf16: the wave's dims [64 w, 64 w + 64) of the slot's 256 in two K = 32 operands; E4: its codes [128 w, 128 w + 128) of 512 in one
    rd_off[e] = (uint32_t)p * 512u + (((uint32_t)(E4 ? wave * 8 + g * 2 + e : (2 * wave + e) * 4 + g) ^ (uint32_t)p) << 4);
  fb_phase<NQT, LO, 2, FIN, E4, F4>(a, lds, red, lane, wave, 0, n_my / 2, src_off, rd_off);
  fb_phase<NQT, LO, 1, FIN, E4, F4>(a, lds, red, lane, wave, (n_my / 2) * 2, n_my & 1, src_off, rd_off);
}

// ------------------------------------------------------------------------------------------------------------
// Forms of this pass that were built, measured and removed (records: profiles/r3/fde_batch_scan_forms_r3.jsonl,
// profiles/r4/fde_batch_ring_experiments_r4.json): 32-page tiles with two workgroups per CU (variant 4), a private DMA
// ring per wave without slot barriers (6), 32-page tiles with a 9- / 4-slot ring at one workgroup per CU (7 / 8) -- all within
// 1.5 % of the paired 64-page form above or slower.

// Masks and the cosine rule of the batched scan, in place: scores[q][page] *= inv_norm[page]; -inf for tombstones and for
// pages outside query q's doc filter (allow_stride_bits = 0: one bitmap for all queries).
__global__ __launch_bounds__(256) void fde_batch_finish_kernel(float* scores, int64_t score_stride, int64_t n, int nq, const float* inv_norm,
                                                               const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits,
                                                               int64_t allow_stride_bits) {
  const int64_t page = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (page >= n) return;
  const float inv = inv_norm ? inv_norm[page] : 1.0f;
  const int32_t o = doc_ord ? doc_ord[page] : 0;
  for (int q = 0; q < nq; ++q) {
    bool m = false;
    if (doc_ord) {
      m = o < 0;
      if (!m && allow) {
        const uint32_t* ab = allow + (size_t)q * (size_t)(allow_stride_bits >> 5);
        m = (int64_t)o >= n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u;
      }
    }
    float* sp = scores + (size_t)q * score_stride + page;
    if (m) *sp = -INFINITY;
    else if (inv_norm) *sp = *sp * inv;
  }
}

// The same pass, also accumulating the FIRST histogram of every query's selection (order-preserving key bits [31:21], the bins
// radix_hist_kernel's pass 0 would count): the finish touches every score anyway, so the 32-request selection loses one of its
// three passes over the 4 * n * nq-byte score matrix.  Block = a page range x kFinQ queries (one 2048-bin LDS histogram each,
// wave-aggregated increments); page metadata is read once per page and block.
constexpr int kFinQ = 4;
__global__ __launch_bounds__(256) void fde_batch_finish_hist_kernel(float* scores, int64_t score_stride, int64_t n, int nq, const float* inv_norm,
                                                                    const int32_t* doc_ord, const uint32_t* allow, int64_t n_allow_bits,
                                                                    int64_t allow_stride_bits, uint32_t* hist0, int64_t hist0_stride_words) {
  __shared__ uint32_t h[kFinQ][2048];
  const int q0 = blockIdx.y * kFinQ;
  const int nql = min(kFinQ, nq - q0);
  for (int i = threadIdx.x; i < kFinQ * 2048; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const int64_t step = (int64_t)gridDim.x * 256;
  const int64_t n_round = ((n + step - 1) / step) * step;  // whole waves walk the loop together (the ballots need every lane)
  for (int64_t page = (int64_t)blockIdx.x * 256 + threadIdx.x; page < n_round; page += step) {
    const bool in = page < n;
    float inv = 1.0f;
    int32_t o = 0;
    if (in) {
      if (inv_norm) inv = inv_norm[page];
      if (doc_ord) o = doc_ord[page];
    }
    float v[kFinQ];
#pragma unroll
    for (int ql = 0; ql < kFinQ; ++ql)
      v[ql] = (in && ql < nql) ? __builtin_nontemporal_load(scores + (size_t)(q0 + ql) * score_stride + page) : -INFINITY;
#pragma unroll
    for (int ql = 0; ql < kFinQ; ++ql) {
      if (ql < nql) {  // block-uniform
        bool valid = false;
        uint32_t bin = 0;
        if (in) {
          bool m = false;
          if (doc_ord) {
            m = o < 0;
            if (!m && allow) {
              const uint32_t* ab = allow + (size_t)(q0 + ql) * (size_t)(allow_stride_bits >> 5);
              m = (int64_t)o >= n_allow_bits || ((ab[o >> 5] >> (o & 31)) & 1u) == 0u;
            }
          }
          float* sp = scores + (size_t)(q0 + ql) * score_stride + page;
          float s = v[ql];
          if (m) {
            s = -INFINITY;
            *sp = s;
          } else if (inv_norm) {
            s = s * inv;
            *sp = s;
          }
          const float s0 = s + 0.0f;
          if (s0 == s0 && s0 != -INFINITY) {
            valid = true;
            bin = topk_ordered_u32(s0) >> 21;
          }
        }
        topk_hist_add_wave(h[ql], bin, valid);
      }
    }
  }
  __syncthreads();
  for (int ql = 0; ql < nql; ++ql) {
    uint32_t* dst = hist0 + (size_t)(q0 + ql) * (size_t)hist0_stride_words;
    for (int i = threadIdx.x; i < 2048; i += 256)
      if (h[ql][i]) atomicAdd(&dst[i], h[ql][i]);
  }
}

}  // namespace

bool fde_scan_batch_supported(int64_t out_dim) { return out_dim >= 1024 && out_dim <= 65536 && out_dim % 1024 == 0; }
bool fde_scan_batch8_supported(int64_t out_dim) { return out_dim >= 2048 && out_dim <= 65536 && out_dim % 2048 == 0; }  // whole groups of four 512-code chunks
// FP4 copy: 1024-code chunks, their count padded to a multiple of four -- supported where the padding reads at most a quarter more (10 240: 12 for 10)
bool fde_scan_batch4_supported(int64_t out_dim) {
  if (out_dim < 4096 || out_dim > 65536 || out_dim % 1024) return false;
  const int64_t kc = out_dim / 1024, kcp = (kc + 3) & ~(int64_t)3;
  return kcp * 4 <= kc * 5;
}
size_t fde_scan_batch_image_bytes(int64_t out_dim) { return (size_t)(out_dim / 256) * 16384 * 2; }  // two query tiles

int launch_fde_scan_batch(const FdeScanBatchArgs& a, hipStream_t s) {
  if (a.n <= 0 || a.n_queries <= 0) return MV_OK;
  if (a.n_queries > kFdeBatchMaxQueries || !fde_scan_batch_supported(a.out_dim)) { set_error("batched FDE scan: %d queries / out_dim %lld not supported", a.n_queries, (long long)a.out_dim); return MV_ERR_INVALID; }
  if (a.n > ((int64_t)1 << 36)) { set_error("batched FDE scan: too many pages"); return MV_ERR_INVALID; }
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int KC = (int)(a.out_dim / 256);
  const int nqt = a.n_queries > 16 ? 2 : 1;
  if (fde_scan_batch_uses_e4m3(a) && a.copy_fp4) {
    // the FP4 copy of the slab: a quarter of the bytes per page (+ the padding chunks); factors and tombstones as in the e4m3 form
    const int kcp = (int)(((a.out_dim / 1024) + 3) & ~(int64_t)3);
    hipLaunchKernelGGL(fde_batch_qscale4_kernel, dim3(kFdeBatchMaxQueries), dim3(256), 0, s, a.q, a.n_queries, (int)a.out_dim, a.qfac);
    hipLaunchKernelGGL(fde_batch_qprep4_kernel, dim3((unsigned)(kcp * nqt)), dim3(256), 0, s, a.q, (const float*)a.qfac, a.n_queries, (int)a.out_dim, kcp, nqt,
                       reinterpret_cast<uint8_t*>(a.image));
    const int64_t n_tiles4 = (a.n + kFbPages - 1) / kFbPages;
    ScanBatchArgs k4{reinterpret_cast<const char*>(a.fde8), reinterpret_cast<const char*>(a.image), a.scores, a.score_stride, a.n,
                     (int32_t)a.out_dim, a.n_queries, (int32_t)n_tiles4, a.fde8_fac, a.doc_ord, a.qfac};
    const dim3 grid4((unsigned)std::min<int64_t>(n_tiles4, ncu));
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, true, true, true>), grid4, dim3(256), 0, s, k4);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, true, true, true>), grid4, dim3(256), 0, s, k4);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, true, true, true>), grid4, dim3(256), 0, s, k4);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, true, true, true>), grid4, dim3(256), 0, s, k4);
    if (a.allow && a.doc_ord)
      hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                         (const float*)nullptr, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  if (fde_scan_batch_uses_e4m3(a)) {
    // the e4m3 copy of the slab: half the bytes per page; page factor (scale, or scale / |d|) and tombstones applied where the scores are written
    hipLaunchKernelGGL(fde_batch_qscale8_kernel, dim3(kFdeBatchMaxQueries), dim3(256), 0, s, a.q, a.n_queries, (int)a.out_dim, a.qfac);
    hipLaunchKernelGGL(fde_batch_qprep8_kernel, dim3((unsigned)((a.out_dim / 512) * nqt)), dim3(256), 0, s, a.q, (const float*)a.qfac, a.n_queries, (int)a.out_dim, nqt,
                       reinterpret_cast<uint8_t*>(a.image));
    const int64_t n_tiles8 = (a.n + kFbPages - 1) / kFbPages;
    ScanBatchArgs k8{reinterpret_cast<const char*>(a.fde8), reinterpret_cast<const char*>(a.image), a.scores, a.score_stride, a.n,
                     (int32_t)a.out_dim, a.n_queries, (int32_t)n_tiles8, a.fde8_fac, a.doc_ord, a.qfac};
    const dim3 grid8((unsigned)std::min<int64_t>(n_tiles8, ncu));
    if (a.hi_only) {  // MV_OPT_FDE_BATCH_VARIANT 2: the queries' hi term only (half the matrix work; the query as coarse as the pages)
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, true, true>), grid8, dim3(256), 0, s, k8);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, true, true>), grid8, dim3(256), 0, s, k8);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, true, true>), grid8, dim3(256), 0, s, k8);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, true, true>), grid8, dim3(256), 0, s, k8);
    if (a.allow && a.doc_ord)  // per-request doc filters: masks only, a pass of their own (as behind the bf16 form)
      hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                         (const float*)nullptr, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
    MV_HIP(hipGetLastError());
    return MV_OK;
  }
  hipLaunchKernelGGL(fde_batch_qprep_kernel, dim3((unsigned)(KC * 2 * nqt)), dim3(256), 0, s, a.q, a.n_queries, (int)a.out_dim, nqt, a.image);
  const int64_t n_tiles = (a.n + kFbPages - 1) / kFbPages;
  ScanBatchArgs k{reinterpret_cast<const char*>(a.fde), reinterpret_cast<const char*>(a.image), a.scores, a.score_stride, a.n,
                  (int32_t)a.out_dim, a.n_queries, (int32_t)n_tiles, a.inv_norm, a.doc_ord, nullptr};
  const bool fin = fde_scan_batch_fuses_finish(a);  // the paired-tile kernel applies the cosine rule and the tombstones itself
  const dim3 grid((unsigned)std::min<int64_t>(n_tiles, ncu));
  if (a.single_tile) {  // one page tile per query fragment (MV_OPT_FDE_BATCH_VARIANT = 3: the cross-check of the paired form)
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch_kernel<1, false>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch_kernel<2, false>), grid, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch_kernel<1>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch_kernel<2>), grid, dim3(256), 0, s, k);
  } else if (fin) {
    if (a.hi_only) {
      if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false, true>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false, true>), grid, dim3(256), 0, s, k);
    } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true, true>), grid, dim3(256), 0, s, k);
    // per-request doc filters are a dependent lookup (ordinal -> bitmap word): they stay a pass of their own, masks only
    if (a.allow && a.doc_ord)
      hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                         (const float*)nullptr, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
    MV_HIP(hipGetLastError());
    return MV_OK;
  } else if (a.hi_only) {
    if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, false>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, false>), grid, dim3(256), 0, s, k);
  } else if (nqt == 1) hipLaunchKernelGGL((fde_scan_batch2_kernel<1, true>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((fde_scan_batch2_kernel<2, true>), grid, dim3(256), 0, s, k);
  if (fde_scan_batch_prebins(a)) {
    const int gx = (int)std::min<int64_t>((a.n + 255) / 256, 128);
    hipLaunchKernelGGL(fde_batch_finish_hist_kernel, dim3((unsigned)gx, (unsigned)((a.n_queries + kFinQ - 1) / kFinQ)), dim3(256), 0, s, a.scores,
                       a.score_stride, a.n, a.n_queries, a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits, a.hist0,
                       a.hist0_stride_bytes / 4);
  } else if (a.inv_norm || a.doc_ord)
    hipLaunchKernelGGL(fde_batch_finish_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a.scores, a.score_stride, a.n, a.n_queries,
                       a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.allow_stride_bits);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
