// mv_encops.hip -- the two elementwise chains of the ColPali / ColQwen encoder forward that PyTorch runs as strings of
// framework kernels (reference formulation: core/embedding/colpali_embedding_model.py:251-262 -> model(**processor(x)) under
// bf16 autocast; the chains are inside transformers' PaliGemma / Qwen2-VL blocks):
//
//   RMSNorm     x.float() -> pow(2) -> mean -> +eps -> rsqrt -> mul -> mul (1 + w) -> cast     7 kernels, ~6 passes over an fp32
//               copy of the activations, twice per decoder layer
//   gated MLP   act(gate_proj(x)) * up_proj(x)                                                  2 kernels, 5 passes over the
//               [tokens x intermediate] tensors (16 384 wide in Gemma-2B: 1 GB per tensor at 32 pages)
//
// rocprofv3 of the full-size ColPali-v1.2 forward (profiles/r1/rocprofv3_kernel_stats_embed_b32.csv): GEMM 55 %, attention 17 %,
// and these elementwise strings ~20 % of the device time.  Here each chain is ONE pass: read the bf16 operands once, do the
// arithmetic of the framework formulation in fp32 IN THE SAME ORDER (so the bf16 results match the module's, up to the summation
// order of the mean), write bf16 once.  HBM-bound by construction: 4 bytes per element for the norm, 6 for the gate.
// Device pointers in, device pointer out, on the caller's stream (torch's current stream): plumbing for the encoder adapters
// (morphik_core_amd/encoder_ops.py), not part of the retrieval path.
#include <algorithm>

#include "mv_common.h"
#include "mv_index_priv.h"

namespace mv {
namespace {

__device__ __forceinline__ float eo_bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t eo_f32_to_bf16(float f) {  // round to nearest even (the framework's cast)
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// One wave per row; dim = 512 * V (V 16-byte vectors per lane, V <= 8 -> dim <= 4096), row in registers between the two sweeps.
// style 0 (Gemma):  out = bf16( (x * r) * (offset + w) )            all in fp32, one rounding   (GemmaRMSNorm.forward)
// style 1 (Llama / Qwen2): out = bf16( w * bf16(x * r) )            the normalised row is rounded first (Qwen2RMSNorm.forward)
template <int V>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const uint16_t* x, const void* w, int w_is_f32, uint16_t* out, int64_t rows, float eps,
                                                           float offset, int style) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int dim = V * 512;
  const u32x4* src = reinterpret_cast<const u32x4*>(x + row * dim) + lane;
  u32x4 v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = __builtin_nontemporal_load(src + i * 64);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = __uint_as_float(v[i][k] << 16), b = __uint_as_float(v[i][k] & 0xffff0000u);
      ss = __builtin_fmaf(a, a, ss);
      ss = __builtin_fmaf(b, b, ss);
    }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) ss += __shfl_xor(ss, s);
  const float r = rsqrtf(ss / (float)dim + eps);
  u32x4* dst = reinterpret_cast<u32x4*>(out + row * dim) + lane;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c0 = (i * 64 + lane) * 8;
    float wf[8];
    if (w_is_f32) {
      const float4 w0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(w) + c0);
      const float4 w1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(w) + c0 + 4);
      wf[0] = w0.x; wf[1] = w0.y; wf[2] = w0.z; wf[3] = w0.w; wf[4] = w1.x; wf[5] = w1.y; wf[6] = w1.z; wf[7] = w1.w;
    } else {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(w) + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) { wf[2 * k] = __uint_as_float(wv[k] << 16); wf[2 * k + 1] = __uint_as_float(wv[k] & 0xffff0000u); }
    }
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = __uint_as_float(v[i][k] << 16), b = __uint_as_float(v[i][k] & 0xffff0000u);
      uint16_t ha, hb;
      if (style == 0) {
        ha = eo_f32_to_bf16((a * r) * (offset + wf[2 * k]));
        hb = eo_f32_to_bf16((b * r) * (offset + wf[2 * k + 1]));
      } else {
        ha = eo_f32_to_bf16(wf[2 * k] * eo_bf16_to_f32(eo_f32_to_bf16(a * r)));
        hb = eo_f32_to_bf16(wf[2 * k + 1] * eo_bf16_to_f32(eo_f32_to_bf16(b * r)));
      }
      o[k] = (uint32_t)ha | ((uint32_t)hb << 16);
    }
    dst[i * 64] = o;
  }
}

// Any dim that is a multiple of 8: one block per row, the row re-read for the second sweep (L2-resident).
__global__ __launch_bounds__(256) void rmsnorm_generic_kernel(const uint16_t* x, const void* w, int w_is_f32, uint16_t* out, int64_t rows, int dim,
                                                              float eps, float offset, int style) {
  __shared__ float part[4];
  const int64_t row = blockIdx.x;
  const uint16_t* xr = x + row * dim;
  float ss = 0.f;
  for (int c = threadIdx.x; c < dim; c += 256) {
    const float a = eo_bf16_to_f32(xr[c]);
    ss = __builtin_fmaf(a, a, ss);
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) ss += __shfl_xor(ss, s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float r = rsqrtf(((part[0] + part[1]) + (part[2] + part[3])) / (float)dim + eps);
  for (int c = threadIdx.x; c < dim; c += 256) {
    const float a = eo_bf16_to_f32(xr[c]);
    const float wf = w_is_f32 ? reinterpret_cast<const float*>(w)[c] : eo_bf16_to_f32(reinterpret_cast<const uint16_t*>(w)[c]);
    out[row * dim + c] = style == 0 ? eo_f32_to_bf16((a * r) * (offset + wf)) : eo_f32_to_bf16(wf * eo_bf16_to_f32(eo_f32_to_bf16(a * r)));
  }
}

// act: 0 = gelu, tanh form (gelu_pytorch_tanh: Gemma, SigLIP), 1 = silu (Qwen2), 2 = gelu, erf form.
// The framework rounds act(gate) to bf16 before the product; so does this.
template <int ACT>
__device__ __forceinline__ float eo_act(float x) {
  if (ACT == 0) {
    const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;  // sqrt(2/pi); the framework's expression, term by term
    const float x_cube = x * x * x;
    const float inner = kBeta * (x + kKappa * x_cube);
    return 0.5f * x * (1.0f + tanhf(inner));
  } else if (ACT == 1) {
    return x / (1.0f + expf(-x));
  } else {
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
  }
}

template <int ACT>
__global__ __launch_bounds__(256) void gated_act_kernel(const uint16_t* gate, const uint16_t* up, uint16_t* out, int64_t rows, int row_vecs,
                                                        int64_t gate_stride_vecs, int64_t up_stride_vecs) {
  // a row holds row_vecs 16-byte vectors; gate / up rows may sit inside a wider matrix (the two halves of one fused gate|up GEMM):
  // vector j of row r at r * stride + j.  One block per row (grid-stride over rows).
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const u32x4* gr = reinterpret_cast<const u32x4*>(gate) + row * gate_stride_vecs;
    const u32x4* ur = reinterpret_cast<const u32x4*>(up) + row * up_stride_vecs;
    u32x4* orow = reinterpret_cast<u32x4*>(out) + row * row_vecs;
    for (int j = threadIdx.x; j < row_vecs; j += 256) {
      const u32x4 g = __builtin_nontemporal_load(gr + j);
      const u32x4 u = __builtin_nontemporal_load(ur + j);
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ga = __uint_as_float(g[k] << 16), gb = __uint_as_float(g[k] & 0xffff0000u);
        const float ua = __uint_as_float(u[k] << 16), ub = __uint_as_float(u[k] & 0xffff0000u);
        const float aa = eo_bf16_to_f32(eo_f32_to_bf16(eo_act<ACT>(ga))), ab = eo_bf16_to_f32(eo_f32_to_bf16(eo_act<ACT>(gb)));
        o[k] = (uint32_t)eo_f32_to_bf16(aa * ua) | ((uint32_t)eo_f32_to_bf16(ab * ub) << 16);
      }
      __builtin_nontemporal_store(o, orow + j);
    }
  }
}

}  // namespace
}  // namespace mv

using namespace mv;

extern "C" {

int mv_enc_rmsnorm_bf16(int device, const void* d_x, const void* d_weight, int weight_dtype, void* d_out, int64_t rows, int32_t dim, float eps,
                        float weight_offset, int style, void* stream) {
  if (!d_x || !d_weight || !d_out || rows < 0 || dim < 8 || dim % 8 || (weight_dtype != MV_F32 && weight_dtype != MV_BF16) || (style != 0 && style != 1)) {
    set_error("mv_enc_rmsnorm_bf16: bad argument (dim must be a positive multiple of 8)");
    return MV_ERR_INVALID;
  }
  // the row kernels move 16-byte vectors (x, out: 8 bf16; weight: 4 fp32 / 8 bf16): a misaligned pointer is a memory fault on the device, not an error
  if ((reinterpret_cast<uintptr_t>(d_x) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 15) || (reinterpret_cast<uintptr_t>(d_weight) & 15)) {
    set_error("mv_enc_rmsnorm_bf16: x, weight and out must be 16-byte aligned");
    return MV_ERR_INVALID;
  }
  if (rows == 0) return MV_OK;
  DeviceGuard g(device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint16_t* x = reinterpret_cast<const uint16_t*>(d_x);
  uint16_t* out = reinterpret_cast<uint16_t*>(d_out);
  const int wf = weight_dtype == MV_F32;
  const dim3 grid((unsigned)((rows + 3) / 4));
  switch (dim % 512 == 0 && dim <= 4096 ? dim / 512 : 0) {
    case 1: hipLaunchKernelGGL((rmsnorm_rows_kernel<1>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 2: hipLaunchKernelGGL((rmsnorm_rows_kernel<2>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 3: hipLaunchKernelGGL((rmsnorm_rows_kernel<3>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 4: hipLaunchKernelGGL((rmsnorm_rows_kernel<4>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 5: hipLaunchKernelGGL((rmsnorm_rows_kernel<5>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 6: hipLaunchKernelGGL((rmsnorm_rows_kernel<6>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 7: hipLaunchKernelGGL((rmsnorm_rows_kernel<7>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    case 8: hipLaunchKernelGGL((rmsnorm_rows_kernel<8>), grid, dim3(256), 0, s, x, d_weight, wf, out, rows, eps, weight_offset, style); break;
    default:
      if (rows > 0x7fffffffLL) { set_error("mv_enc_rmsnorm_bf16: too many rows"); return MV_ERR_INVALID; }
      hipLaunchKernelGGL(rmsnorm_generic_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, d_weight, wf, out, rows, (int)dim, eps, weight_offset, style);
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int mv_enc_gated_act_bf16(int device, const void* d_gate, int64_t gate_row_stride, const void* d_up, int64_t up_row_stride, void* d_out, int64_t rows,
                          int64_t cols, int act, void* stream) {
  if (!d_gate || !d_up || !d_out || rows < 0 || cols < 8 || cols % 8 || gate_row_stride % 8 || up_row_stride % 8 || gate_row_stride < cols ||
      up_row_stride < cols || act < 0 || act > 2 || (reinterpret_cast<uintptr_t>(d_gate) & 15) || (reinterpret_cast<uintptr_t>(d_up) & 15) ||
      (reinterpret_cast<uintptr_t>(d_out) & 15)) {
    set_error("mv_enc_gated_act_bf16: bad argument (cols and row strides must be multiples of 8 elements, pointers 16-byte aligned)");
    return MV_ERR_INVALID;
  }
  if (rows == 0) return MV_OK;
  DeviceGuard g(device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (cols / 8 > 0x7fffffffLL) { set_error("mv_enc_gated_act_bf16: row too long"); return MV_ERR_INVALID; }
  const int row_vecs = (int)(cols / 8);
  const unsigned grid = (unsigned)std::min<int64_t>(rows, (int64_t)1 << 20);
  const uint16_t* gp = reinterpret_cast<const uint16_t*>(d_gate);
  const uint16_t* up = reinterpret_cast<const uint16_t*>(d_up);
  uint16_t* out = reinterpret_cast<uint16_t*>(d_out);
  switch (act) {
    case 0: hipLaunchKernelGGL((gated_act_kernel<0>), dim3(grid), dim3(256), 0, s, gp, up, out, rows, row_vecs, gate_row_stride / 8, up_row_stride / 8); break;
    case 1: hipLaunchKernelGGL((gated_act_kernel<1>), dim3(grid), dim3(256), 0, s, gp, up, out, rows, row_vecs, gate_row_stride / 8, up_row_stride / 8); break;
    default: hipLaunchKernelGGL((gated_act_kernel<2>), dim3(grid), dim3(256), 0, s, gp, up, out, rows, row_vecs, gate_row_stride / 8, up_row_stride / 8); break;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // extern "C"
