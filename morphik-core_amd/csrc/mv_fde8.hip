// mv_fde8.hip -- the FDE coarse stage on an e4m3 COPY of the FDE slab (MV_WITH_FDE_E4M3; round 6).
//
// The coarse stage of MV_MODE_FDE_THEN_FLOAT (reference: a TurboPuffer ANN query over the documents' FDE vectors,
// core/vector_store/fast_multivector_store.py:527-533 -- approximate by contract) reads out_dim x 2 bytes per page from the bf16 slab and is
// nine tenths of every request.  Its job is a candidate list, so the slab it reads may be coarser than the one the library
// exports: every FDE row is kept a second time as out_dim e4m3 codes under ONE power-of-two scale (the page quantiser of mv_fp8.hip on the
// row viewed as out_dim / 128 rows of 128: oracle orc_quantize_page_fp8), and the scan below reads HALF the bytes:
//
//     score[page] = (sum_i q_i * decode(code[page][i])) * scale[page] (* 1 / |d| for the cosine rule)
//
// The kernel is fde_scan_rowq_kernel's shape (mv_fde.hip; DESIGN 3.15: what the nt LDS-DMA ring streams fastest) on 10 KiB rows: one fresh
// workgroup per 32 consecutive rows (320 KiB of contiguous slab), two waves per row -- wave w streams the (w & 1)-th half (5 KiB) of every
// row of ITS 16-row group through a private ring of three row slots and keeps its 80-float slice of the fp32 query FDE in registers; a lane
// reads back the 16 bytes it requested (16 codes), converts them with v_cvt_pk_f32_fp8 (two codes per instruction) and multiplies in fp32:
// 8 conversions + 16 FMAs per 16 bytes, against 8 + 8 per 16 bytes of the bf16 form -- 1.5 vector instructions per byte at twice the byte
// rate per page, far from the VALU's limit.  The query stays fp32: only the documents are quantised.
#include <algorithm>

#include "mv_common.h"

namespace mv {
namespace {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

struct Scan8K {
  const uint8_t* fde8;
  const float* scale;
  const float* inv_norm;
  const int32_t* doc_ord;
  const uint32_t* allow;
  int64_t n_allow_bits;
  const float* q;
  float* scores;
  int64_t n;
  int32_t out_dim;
};

template <int N>
__device__ __forceinline__ void f8_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int CPT, int D>
__device__ __forceinline__ void f8_wait_left(int left) {  // all rows issued; `left` (< D - 1) rows are younger than the one needed
  if (D > 3 && left == 2) f8_wait_vmcnt<2 * CPT>();
  else if (D > 2 && left == 1) f8_wait_vmcnt<1 * CPT>();
  else f8_wait_vmcnt<0>();
}

template <int CPW, int WPR, int D>  // out_dim = 1024 CPW WPR codes; CPW <= 5; WPR in {2, 4}; 4 / WPR row groups per workgroup
__global__ __launch_bounds__(256) void fde_scan_rowq8_kernel(Scan8K a, int ru) {
  static_assert(CPW >= 1 && CPW <= 5 && (WPR == 2 || WPR == 4) && D >= 2 && D <= 4 && CPW * (D - 1) <= 63, "row shape");
  constexpr int G = 4 / WPR;
  constexpr int SLOT = CPW * 1024;
  // one __shared__ object only (a second one makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) char lds[4 * D * SLOT + 4 * 64 * 4];
  float* part_sum = reinterpret_cast<float*>(lds + 4 * D * SLOT);  // [wave][row of the group]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int group = wave / WPR, part = wave % WPR;
  char* ring = lds + wave * (D * SLOT);
  const int voff = lane * 16;

  // ---- group prologue: lane i <-> the group's i-th row
  const int64_t base = ((int64_t)blockIdx.x * G + group) * (int64_t)ru;
  const int64_t myrow = base + lane;
  const bool valid = lane < ru && myrow < a.n;
  bool masked = false;
  float my_fac = 1.0f;
  if (valid) {
    if (a.doc_ord) {
      const int32_t o = a.doc_ord[myrow];
      masked = o < 0 || (a.allow && ((int64_t)o >= a.n_allow_bits || ((a.allow[o >> 5] >> (o & 31)) & 1u) == 0u));
    }
    if (!masked && part == 0) my_fac = a.inv_norm ? a.scale[myrow] * a.inv_norm[myrow] : a.scale[myrow];
  }
  const uint64_t live = __ballot(valid && !masked);
  uint64_t iss = live, cons = live;
  int to_issue = __builtin_popcountll(live);
  int to_read = to_issue;
  int iss_slot = 0, cons_slot = 0;
  const char* qbase = reinterpret_cast<const char*>(a.fde8) + (size_t)part * SLOT;
  const size_t row_bytes = (size_t)a.out_dim;

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

  // this wave's slice of the query FDE (16 floats per 16-byte chunk of codes), behind the first rows' DMAs; waited for HERE, once
  f32x2 q[CPW][8];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float* qp = a.q + (size_t)(part * CPW + c) * 1024 + lane * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(qp + 4 * k);
      q[c][2 * k] = f32x2{v.x, v.y};
      q[c][2 * k + 1] = f32x2{v.z, v.w};
    }
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(q[c][k]));
  asm volatile("" : "+v"(my_fac));

  float my_part = 0.0f;
  while (cons) {
    const int i = __builtin_ctzll(cons);
    cons &= cons - 1;
    if (to_issue > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // WAR: the last reads of the slot being refilled
      issue_next();
      f8_wait_vmcnt<CPW * (D - 1)>();
    } else {
      f8_wait_left<CPW, D>(to_read - 1);
    }
    --to_read;
    const char* slot = ring + cons_slot * SLOT + voff;
    cons_slot = (cons_slot + 1 == D) ? 0 : cons_slot + 1;
    f32x2 acc2 = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(slot + c * 1024);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x2 d0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[k], false);  // codes 0, 1 of the dword
        const f32x2 d1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[k], true);   // codes 2, 3
        acc2 = __builtin_elementwise_fma(d0, q[c][2 * k], acc2);
        acc2 = __builtin_elementwise_fma(d1, q[c][2 * k + 1], acc2);
      }
    }
    float acc = acc2[0] + acc2[1];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
    if (lane == i) my_part = acc;
  }

  // ---- group epilogue: join the parts, one store per row
  part_sum[wave * 64 + lane] = my_part;
  __syncthreads();
  if (part == 0 && valid) {
    const float* p = part_sum + wave * 64 + lane;  // waves [wave, wave + WPR) are this group's parts
    float t;
    if constexpr (WPR == 4) t = (p[0] + p[64]) + (p[128] + p[192]);
    else t = p[0] + p[64];
    a.scores[myrow] = masked ? -INFINITY : t * my_fac;
  }
}

}  // namespace

// widths the row-half kernel is instantiated for: 1024 * CPW * 2 codes per row (the reference's 10 240 = 5 * 2)
bool fde_scan8_supported(int64_t out_dim) { return out_dim == 10240 || out_dim == 4096 || out_dim == 2048; }

int launch_fde_scan8(const FdeScan8Args& a, hipStream_t s) {
  if (a.n <= 0) return MV_OK;
  if (!fde_scan8_supported(a.out_dim)) { set_error("e4m3 FDE scan: width %lld not supported", (long long)a.out_dim); return MV_ERR_INVALID; }
  Scan8K k{a.fde8, a.scale, a.inv_norm, a.doc_ord, a.allow, a.n_allow_bits, a.q, a.scores, a.n, (int32_t)a.out_dim};
  const int ru = 16;  // rows per group; two groups per workgroup: 32 rows = 320 KiB at 10 240 codes
  const int64_t units = (a.n + ru - 1) / ru;
  if (units > ((int64_t)1 << 24)) { set_error("e4m3 FDE scan: more than 2^28 rows per launch is not supported"); return MV_ERR_INVALID; }
  const dim3 grid((unsigned)((units + 1) / 2));
  if (a.out_dim == 10240) hipLaunchKernelGGL((fde_scan_rowq8_kernel<5, 2, 3>), grid, dim3(256), 0, s, k, ru);
  else if (a.out_dim == 4096) hipLaunchKernelGGL((fde_scan_rowq8_kernel<2, 2, 3>), grid, dim3(256), 0, s, k, ru);
  else hipLaunchKernelGGL((fde_scan_rowq8_kernel<1, 2, 3>), grid, dim3(256), 0, s, k, ru);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
