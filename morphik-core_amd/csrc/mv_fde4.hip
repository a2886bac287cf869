// mv_fde4.hip -- the FDE coarse stage of a SINGLE request on an FP4 (e2m1) COPY of the FDE slab (MV_WITH_FDE_FP4; round 6, closing session).
//
// The coarse stage of MV_MODE_FDE_THEN_FLOAT (reference: a TurboPuffer ANN query over the documents' FDE vectors,
// core/vector_store/fast_multivector_store.py:526-532 -- approximate by contract) is nine tenths of a request and reads
// out_dim x 2 bytes per page from the bf16 slab, out_dim from its e4m3 copy (mv_fde8.hip).  What the stage owes is a candidate list; priced through
// mv_index_import_fde before anything was built (tools/fde_4bit_recall_probe.py, profiles/r6/fde_4bit_copy_recall_priced_r6.json), a 4-bit copy
// under one power-of-two scale per row keeps the recall of the bf16 slab on the bench's hard negatives (0.9922 = the bf16 slab's at 75
// candidates with the saturating scale below, 1.0 at 1000).  So every FDE row is kept a third time as out_dim / 2 bytes:
//
//     value_i = decode(code_i) * scale[page]      code: bit 3 sign, bits 2..0 -> {0, 0.5, 1, 1.5, 2, 3, 4, 6}; element 2i in the LOW nibble
//     scale   = 2^e, the smallest power of two with 12 * 2^e >= max|x| over the row: half the covering scale -- elements beyond 6 * scale
//               saturate at the top code, the bulk gets a grid twice as fine (priced: no page lost on the hard negatives against 3 of 640
//               with the covering scale)                                                   (oracle: orc_quantize_fde_fp4)
//     score[page] = (sum_i q_i * decode(code[page][i])) * scale[page] (* 1 / |d| for the cosine rule)
//
// Scan kernel: fde_scan_rowq8_kernel's transport (mv_fde8.hip) on 5 KiB rows -- one fresh workgroup per 64 consecutive rows
// (320 KiB of contiguous slab), ONE wave per row of its 16-row group: the wave streams whole rows through a private ring of three row
// slots (nt LDS-DMA) and keeps the whole fp32 query FDE in registers (160 VGPRs at the reference's width).  A lane reads back the 16
// bytes it requested (32 codes), converts them two at a time with v_cvt_scalef32_pk_f32_fp4 and multiplies in fp32 with packed FMAs:
// 16 conversions + 16 v_pk_fma_f32 per 16 bytes.  The query stays fp32: only the documents are quantised.
// Batches of requests read the same codes through the F4 form of the batched pass (mv_fde_batch.hip: both MFMA operands FP4).
#include <algorithm>

#include "mv_common.h"
#include "mv_e4m3.h"

namespace mv {
namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// ------------------------------------------------------------------------------------------------ quantiser
// one workgroup per row: bf16 values [od] -> od / 2 code bytes + scale (+ scale / |d|)
__global__ __launch_bounds__(256) void fde4_quantize_kernel(const uint16_t* src, int64_t od, uint8_t* codes, float* scale, const float* inv_norm, float* cfac) {
  __shared__ uint32_t wmax[4];
  const int64_t row = blockIdx.x;
  const uint16_t* r = src + (size_t)row * od;
  uint32_t amax = 0;
  for (int64_t i = (int64_t)threadIdx.x * 8; i < od; i += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(r + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) amax = max(amax, max(w[k] & 0x7fffu, (w[k] >> 16) & 0x7fffu));
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, s));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
  int e = 0;
  if (amax >= 0x0080u && amax < 0x7f80u) {
    const int e0 = (int)(amax >> 7) - 127;
    e = ((amax & 0x7fu) <= 0x40u) ? e0 - 3 : e0 - 2;  // 12 * 2^e >= amax: half the covering scale (the largest elements saturate)
    e = min(max(e, -120), 120);
  }
  const float sc = __uint_as_float((uint32_t)(127 + e) << 23);
  const float inv = __uint_as_float((uint32_t)(127 - e) << 23);
  if (threadIdx.x == 0) {
    scale[row] = sc;
    cfac[row] = sc * inv_norm[row];
  }
  uint32_t* out = reinterpret_cast<uint32_t*>(codes + (size_t)row * (od / 2));
  for (int64_t i = (int64_t)threadIdx.x * 8; i < od; i += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(r + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float x0 = __uint_as_float(w[k] << 16), x1 = __uint_as_float(w[k] & 0xffff0000u);
      packed |= (fp4_encode(x0 * inv) | (fp4_encode(x1 * inv) << 4)) << (8 * k);
    }
    out[i >> 3] = packed;
  }
}

// ------------------------------------------------------------------------------------------------ scan
struct Scan4K {
  const uint8_t* fde4;
  const float* fac;  // scale, or scale / |d| under the cosine rule
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const float* q;
  float* scores;
  int64_t n;
  int32_t out_dim;
};

template <int N>
__device__ __forceinline__ void f4_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int CPW, int D>
__device__ __forceinline__ void f4_wait_left(int left) {  // all rows issued; `left` (< D - 1) rows are younger than the one needed
  if (D > 3 && left == 2) f4_wait_vmcnt<2 * CPW>();
  else if (D > 2 && left == 1) f4_wait_vmcnt<1 * CPW>();
  else f4_wait_vmcnt<0>();
}

template <int CTRL>
__device__ __forceinline__ float f4_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int CPW, int D>  // a row is CPW KiB of codes (out_dim = 2048 CPW); ring of D row slots per wave
__global__ __launch_bounds__(256) void fde_scan_row4_kernel(Scan4K a, int ru) {
  static_assert(CPW >= 1 && CPW <= 5 && D >= 2 && D <= 4 && CPW * (D - 1) <= 63, "row shape");
  constexpr int SLOT = CPW * 1024;
  __shared__ __attribute__((aligned(16))) char lds[4 * D * SLOT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = lds + wave * (D * SLOT);
  const int voff = lane * 16;

  // ---- group prologue: lane i <-> the group's i-th row
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (int64_t)ru;
  const int64_t myrow = base + lane;
  const bool valid = lane < ru && myrow < a.n;
  bool masked = false;
  float my_fac = 1.0f;
  if (valid) {
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[myrow];
      masked = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
    }
    if (!masked) my_fac = a.fac[myrow];
  }
  const uint64_t live = __ballot(valid && !masked);
  uint64_t iss = live, cons = live;
  int to_issue = __builtin_popcountll(live);
  int to_read = to_issue;
  int iss_slot = 0, cons_slot = 0;
  const char* qbase = reinterpret_cast<const char*>(a.fde4);
  const size_t row_bytes = (size_t)(a.out_dim >> 1);

  auto issue_next = [&]() {
    const int i = __builtin_ctzll(iss);
    iss &= iss - 1;
    const char* tp = qbase + (size_t)(base + i) * row_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t m0a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + iss_slot * SLOT));
    uint32_t keep;
    // the instruction offset (12 bits) walks BOTH addresses; the fifth chunk takes a second M0 and a +4 KiB lane offset
    if (CPW == 5) {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %4 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:3072 nt\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %5, %4 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff), "s"(m0a), "s"(m0a + 4096u), "s"(tpu), "v"(voff + 4096)
          : "memory");
    } else {
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %3 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff + c * 1024), "s"(m0a + (uint32_t)(c * 1024)), "s"(tpu)
            : "memory");
      }
    }
    iss_slot = (iss_slot + 1 == D) ? 0 : iss_slot + 1;
    --to_issue;
  };

#pragma unroll
  for (int k = 0; k < D - 1; ++k)
    if (to_issue > 0) issue_next();

  // the query FDE (32 floats per 16-byte chunk of codes), behind the first rows' DMAs; waited for HERE, once
  f32x2 q[CPW][16];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float* qp = a.q + (size_t)c * 2048 + lane * 32;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(qp + 4 * k);
      q[c][2 * k] = f32x2{v.x, v.y};
      q[c][2 * k + 1] = f32x2{v.z, v.w};
    }
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(q[c][k]));
  asm volatile("" : "+v"(my_fac));

  float my_score = 0.0f;
  while (cons) {
    const int i = __builtin_ctzll(cons);
    cons &= cons - 1;
    if (to_issue > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: the last reads of the slot being refilled
      issue_next();
      f4_wait_vmcnt<CPW * (D - 1)>();
    } else {
      f4_wait_left<CPW, D>(to_read - 1);
    }
    --to_read;
    const char* slot = ring + cons_slot * SLOT + voff;
    cons_slot = (cons_slot + 1 == D) ? 0 : cons_slot + 1;
    f32x2 acc2 = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(slot + c * 1024);
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // dword k: codes 8k .. 8k + 7 of the lane's 32 (byte b: codes 2b, 2b + 1; low nibble first)
        acc2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v[k], 1.0f, 0), q[c][4 * k + 0], acc2);
        acc2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v[k], 1.0f, 1), q[c][4 * k + 1], acc2);
        acc2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v[k], 1.0f, 2), q[c][4 * k + 2], acc2);
        acc2 = __builtin_elementwise_fma(__builtin_amdgcn_cvt_scalef32_pk_f32_fp4(v[k], 1.0f, 3), q[c][4 * k + 3], acc2);
      }
    }
    // wave sum without LDS traffic: four DPP rotations inside every 16-lane row, then the four row sums through v_readlane
    float acc = acc2[0] + acc2[1];
    acc += f4_dpp<0x128>(acc);  // row_ror:8
    acc += f4_dpp<0x124>(acc);  // row_ror:4
    acc += f4_dpp<0x122>(acc);  // row_ror:2
    acc += f4_dpp<0x121>(acc);  // row_ror:1
    const int ab = __float_as_int(acc);
    const float tot = (__int_as_float(__builtin_amdgcn_readlane(ab, 0)) + __int_as_float(__builtin_amdgcn_readlane(ab, 16))) +
                      (__int_as_float(__builtin_amdgcn_readlane(ab, 32)) + __int_as_float(__builtin_amdgcn_readlane(ab, 48)));
    if (lane == i) my_score = tot;
  }
  if (valid) a.scores[myrow] = masked ? -INFINITY : my_score * my_fac;
}

}  // namespace

// widths the kernel is instantiated for: 2048 * CPW codes per row (the reference's 10 240 = 5 KiB rows)
bool fde_scan4_supported(int64_t out_dim) { return out_dim == 10240 || out_dim == 4096 || out_dim == 2048; }

int launch_fde_quantize_fp4(const uint16_t* d_fde_rows, int64_t out_dim, int64_t n, uint8_t* d_codes, float* d_scale, const float* d_inv_norm, float* d_cfac,
                            hipStream_t s) {
  if (n <= 0) return MV_OK;
  if (out_dim % 2048) { set_error("fp4 FDE copy: width %lld is not a multiple of 2048", (long long)out_dim); return MV_ERR_INVALID; }
  hipLaunchKernelGGL(fde4_quantize_kernel, dim3((unsigned)n), dim3(256), 0, s, d_fde_rows, out_dim, d_codes, d_scale, d_inv_norm, d_cfac);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

// a.fde8 = the fp4 codes [pages][out_dim / 2]; a.scale = the per-page factor the scores take (scale, or scale / |d|): a.inv_norm is not read
int launch_fde_scan4(const FdeScan8Args& a, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  if (!fde_scan4_supported(a.out_dim)) { set_error("fp4 FDE scan: width %lld not supported", (long long)a.out_dim); return MV_ERR_INVALID; }
  Scan4K k{a.fde8, a.scale, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n, (int32_t)a.out_dim};
  const int ru = 16;  // rows per wave; four waves per workgroup: 64 rows = 320 KiB at 10 240 codes
  const int64_t units = (a.n + ru - 1) / ru;
  if (units > ((int64_t)1 << 26)) { set_error("fp4 FDE scan: more than 2^30 rows per launch is not supported"); return MV_ERR_INVALID; }
  const dim3 grid((unsigned)((units + 3) / 4));
  if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_row4_kernel<5, 3>), grid, dim3(256), 0, s, k, ru);
  else if (a.out_dim == 4096) hipLaunchKernelGGL((fde_scan_row4_kernel<2, 3>), grid, dim3(256), 0, s, k, ru);
  else hipLaunchKernelGGL((fde_scan_row4_kernel<1, 3>), grid, dim3(256), 0, s, k, ru);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
