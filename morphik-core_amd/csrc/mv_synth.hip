// mv_synth.hip -- on-device synthetic corpus generator: any (unit,row) of the corpus
// is a pure function of (seed, unit, row), so a 262 GB slab is produced shard by shard on the GPU
// and the CPU oracle (oracle/mv_oracle.c:orc_synth_rows) regenerates any page it wants to check.
// Spec (must stay bit-identical to the oracle):
//   w = philox4x32-10(ctr=(unit_lo, unit_hi, row, chunk), key=(seed_lo, seed_hi)),  chunk = dim/4 index
//   x[4*chunk+j] = byte-sum(w[j]) - 510 ; ss = sum x^2 (exact integer)
//   y = bf16_rne( (float)( (double)x / sqrt((double)ss) ) )
// Also holds the small utility kernels (fp32->bf16, ragged scatter, read-bandwidth calibration).
#include <algorithm>

#include "mv_common.h"

namespace mv {
namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// Four dims (chunk) of row `row` of unit `unit`: 32 lanes of a half-wave make one row (the norm is a half-wave reduction).
__device__ __forceinline__ uint2 synth_row_chunk(uint64_t seed, uint64_t unit, int32_t row, int chunk);

// One wave = two rows (32 lanes x 4 dims each).  dim fixed at 128.
__global__ __launch_bounds__(256) void synth_rows_kernel(uint16_t* out, uint64_t seed, uint64_t first_unit,
                                                         int64_t n_units, int32_t n_rows, int32_t stride_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t total_rows = n_units * (int64_t)stride_rows;
  const int64_t grow = wave * 2 + (lane >> 5);  // global row index in the slab region
  if (grow >= total_rows) return;
  const int64_t ui = grow / stride_rows;
  const int32_t row = (int32_t)(grow - ui * stride_rows);
  const int chunk = lane & 31;
  uint2 packed = make_uint2(0u, 0u);
  if (row < n_rows) packed = synth_row_chunk(seed, first_unit + (uint64_t)ui, row, chunk);  // wave-uniform per half; rows >= n_rows are zero filled
  *reinterpret_cast<uint2*>(out + grow * kDim + chunk * 4) = packed;
}

// Ragged form: one block per unit; unit i gets its first n_rows[i] rows at base + row0(i) * 128 and zero rows up to the end of its
// allotment (row_off[i + 1], or the stride slot).
__global__ __launch_bounds__(256) void synth_rows_ragged_kernel(uint16_t* base, uint64_t seed, uint64_t first_unit, const int32_t* n_rows,
                                                                const int64_t* row_off, int32_t stride_rows) {
  const int64_t ui = blockIdx.x;
  const int64_t r0 = row_off ? row_off[ui] : ui * (int64_t)stride_rows;
  const int32_t slot = row_off ? (int32_t)(row_off[ui + 1] - r0) : stride_rows;
  const int32_t nr = n_rows[ui];
  const int half = threadIdx.x >> 5, chunk = threadIdx.x & 31;  // 8 rows per step
  for (int32_t row = half; row < slot; row += 8) {
    uint2 packed = make_uint2(0u, 0u);
    if (row < nr) packed = synth_row_chunk(seed, first_unit + (uint64_t)ui, row, chunk);
    *reinterpret_cast<uint2*>(base + (r0 + row) * kDim + chunk * 4) = packed;
  }
}

__device__ __forceinline__ uint2 synth_row_chunk(uint64_t seed, uint64_t unit, int32_t row, int chunk) {
  uint2 packed;
  {
    uint32_t w[4];
    philox4x32_10((uint32_t)unit, (uint32_t)(unit >> 32), (uint32_t)row, (uint32_t)chunk, (uint32_t)seed,
                  (uint32_t)(seed >> 32), w);
    int32_t x[4];
    int32_t ss = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = (int32_t)(w[j] & 0xff) + (int32_t)((w[j] >> 8) & 0xff) + (int32_t)((w[j] >> 16) & 0xff) + (int32_t)(w[j] >> 24) - 510;
      ss += x[j] * x[j];
    }
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) ss += __shfl_xor(ss, s);  // stays inside the 32-lane half
    if (ss == 0) {
      if (chunk == 0) x[0] = 1;
      ss = 1;
    }
    const double nrm = sqrt((double)ss);
    uint16_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = f32_to_bf16_rne((float)((double)x[j] / nrm));
    packed.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    packed.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
  }
  return packed;
}

__global__ void f32_to_bf16_kernel(const float* in, uint16_t* out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(in + i);
    uint2 p;
    p.x = (uint32_t)f32_to_bf16_rne(v.x) | ((uint32_t)f32_to_bf16_rne(v.y) << 16);
    p.y = (uint32_t)f32_to_bf16_rne(v.z) | ((uint32_t)f32_to_bf16_rne(v.w) << 16);
    *reinterpret_cast<uint2*>(out + i) = p;
  } else {
    for (int64_t j = i; j < n; ++j) out[j] = f32_to_bf16_rne(in[j]);
  }
}

// One block per page: copy/convert rows [off[p], off[p+1]) into slab page p, zero the tail.
// dst_row_off (packed layout): page p occupies rows [dst_row_off[p], dst_row_off[p + 1]) of the slab instead of the p-th stride slot.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const void* src, int dtype, const int64_t* off, int32_t stride,
                                                           uint16_t* slab, int32_t* nonfinite, uint16_t* slab_lo, const int64_t* dst_row_off) {
  const int64_t p = blockIdx.x;
  const int64_t r0 = off[p];
  const int32_t nr = (int32_t)(off[p + 1] - r0);
  const int64_t d0 = dst_row_off ? dst_row_off[p] : p * (int64_t)stride;
  const int32_t slot_rows = dst_row_off ? (int32_t)(dst_row_off[p + 1] - d0) : stride;
  uint16_t* dst = slab + d0 * kDim;
  uint16_t* dst_lo = slab_lo ? slab_lo + d0 * kDim : nullptr;
  const int total = slot_rows * (kDim / 4);  // 4 elements per thread-step
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int row = i / (kDim / 4), c4 = i % (kDim / 4);
    uint2 v = make_uint2(0u, 0u);
    uint2 l = make_uint2(0u, 0u);  // lo half of the split: bf16(x - bf16(x)); the subtraction is exact in fp32
    if (row < nr) {
      const int64_t e = (r0 + row) * kDim + c4 * 4;
      if (dtype == MV_BF16) {
        v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(src) + e);
      } else {
        const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + e);
        const uint32_t h0 = f32_to_bf16_rne(f.x), h1 = f32_to_bf16_rne(f.y), h2 = f32_to_bf16_rne(f.z), h3 = f32_to_bf16_rne(f.w);
        v.x = h0 | (h1 << 16);
        v.y = h2 | (h3 << 16);
        if (dst_lo) {
          const uint32_t l0 = f32_to_bf16_rne(f.x - __uint_as_float(h0 << 16)), l1 = f32_to_bf16_rne(f.y - __uint_as_float(h1 << 16));
          const uint32_t l2 = f32_to_bf16_rne(f.z - __uint_as_float(h2 << 16)), l3 = f32_to_bf16_rne(f.w - __uint_as_float(h3 << 16));
          l.x = l0 | (l1 << 16);
          l.y = l2 | (l3 << 16);
        }
      }
    }
    *reinterpret_cast<uint2*>(dst + (int64_t)row * kDim + c4 * 4) = v;
    // NaN / +-Inf (exponent all ones) in the bf16 image the slabs are derived from -- an fp32 value beyond the bf16 range counts:
    // it IS an Inf in the slab.  The ingest reports it (mv_index_add*: MV_ERR_INVALID, nothing published).
    bool bad = false;
    if (nonfinite) {
      const uint32_t a = v.x & 0x7f807f80u, b = v.y & 0x7f807f80u;
      bad = (a & 0xffffu) == 0x7f80u || (a >> 16) == 0x7f80u || (b & 0xffffu) == 0x7f80u || (b >> 16) == 0x7f80u;
      if (bad) atomicOr(nonfinite, 1);
    }
    if (dst_lo) *reinterpret_cast<uint2*>(dst_lo + (int64_t)row * kDim + c4 * 4) = bad ? make_uint2(0u, 0u) : l;  // (x - Inf would be a NaN)
  }
}

// Read-bandwidth calibration: grid-stride 16 B loads, xor-reduce into a sink so nothing is elided.
__global__ __launch_bounds__(256) void read_bw_kernel(const uint4* buf, int64_t n16, float* sink) {
  uint32_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = buf[i], b = buf[i + stride], c = buf[i + 2 * stride], d = buf[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) {
    const uint4 a = buf[i];
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x9e3779b9u) *sink = 1.0f;  // practically never; keeps the loads alive
}

// Streaming form of the same calibration: every wave reads whole contiguous 16 KiB pieces (16 non-temporal 1 KiB
// wave-loads in flight), consecutive waves take consecutive pieces -- the access pattern of the scan kernels without
// any arithmetic.  This is the "measured HBM peak" the rooflines are also quoted against.
__global__ __launch_bounds__(256) void read_bw_nt_kernel(const uint4* buf, int64_t n_pieces, float* sink) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  for (int64_t p = wave; p < n_pieces; p += nwaves) {
    const u32x4* src = reinterpret_cast<const u32x4*>(buf) + p * 1024 + lane;
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + i * 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
  }
  if (acc == 0x9e3779b9u) *sink = 1.0f;
}

// MFMA calibration: register-only MFMA chains, no memory traffic.  SHAPE 0 = v_mfma_f32_16x16x32_bf16 (8 independent
// accumulators per wave, what the round-1/2 scans issue), 1 = v_mfma_f32_32x32x16_bf16 (4 independent accumulators; the
// guide's micro-benchmarks put it ~15 % above the 16x16 shape).  The operands are pseudo-random bf16 values of embedding
// magnitude (|x| ~ 0.09, both signs), four distinct A and four distinct B registers per wave: the chip clocks to its
// power budget, and a loop over constant operands toggles far fewer datapath bits than the scans' real data do (the
// guide measures +19 % TFLOP/s on zero-filled inputs) -- that would overstate the ceiling.  A launch runs for
// milliseconds so the clock has settled.
__device__ __forceinline__ uint32_t cal_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ short cal_bf16(uint32_t h) {
  // sign = bit 31, exponent 0x3d (2^-4 .. 2^-3 scaled down by up to 4), 7 random mantissa bits
  const uint32_t sign = (h >> 16) & 0x8000u, expo = (0x7bu - ((h >> 8) & 3u)) << 7, man = h & 0x7fu;
  return (short)(sign | expo | man);
}
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, float* sink) {
  using bf16x8 = __attribute__((ext_vector_type(8))) short;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[r][i] = cal_bf16(cal_hash((blockIdx.x * 256u + threadIdx.x) * 64u + r * 8 + i));
      b[r][i] = cal_bf16(cal_hash((blockIdx.x * 256u + threadIdx.x) * 64u + 32 + r * 8 + i));
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(a[r]), "+v"(b[r]));
  float t = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[j & 3], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) t += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[j], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) t += acc[j][i];
  }
  if (t == 1.2345f) *sink = t;
}

}  // namespace

// The access pattern of the batched FDE coarse pass without LDS, barriers or arithmetic: a workgroup walks tiles of 32 KiB / PIECE rows
// of a [rows][20 480 B] matrix, K chunk by K chunk -- per step every row of the tile contributes PIECE contiguous bytes (512 in the
// scan), three steps (96 KiB a CU) in flight.  PIECE = 20 480: whole rows, the single-query scan's pattern.
template <int PIECE>
__global__ __launch_bounds__(256) void read_bw_strided_kernel(const char* buf, int64_t n_rows, float* sink) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int ROW = 20480;
  constexpr int P = PIECE > 8192 ? 8192 : PIECE;   // bytes a wave takes from a row per step (whole rows: 8 KiB of it, row after row)
  constexpr int TILE_ROWS = PIECE > 8192 ? 4 : 32768 / PIECE, WAVE_ROWS = TILE_ROWS / 4;
  constexpr int STEPS = PIECE > 8192 ? 3 : ROW / PIECE;  // whole rows: 2.5 steps of 8 KiB -> the last one re-reads (timing only)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_tiles = n_rows / TILE_ROWS;
  uint32_t acc = 0;
  u32x4 v[3][8];
  int64_t tile = blockIdx.x;
  int step = 0;
  auto load = [&](u32x4 (&d)[8]) {
    if (tile >= n_tiles) return;
    const char* base = buf + ((size_t)tile * TILE_ROWS + (size_t)wave * WAVE_ROWS) * ROW + (size_t)min(step * P, ROW - P);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int x = i * 1024 + lane * 16;
      d[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)(x / P) * ROW + (x % P)));
    }
    if (++step == STEPS) { step = 0; tile += gridDim.x; }
  };
  auto use = [&](const u32x4 (&d)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= d[i][0] ^ d[i][1] ^ d[i][2] ^ d[i][3];
  };
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const int64_t total = my_tiles * STEPS;
  load(v[0]); load(v[1]); load(v[2]);
  for (int64_t s = 0; s < total; s += 3) {
    use(v[0]); load(v[0]);
    use(v[1]); load(v[1]);
    use(v[2]); load(v[2]);
  }
  if (acc == 0x9e3779b9u) *sink = 1.0f;
}

// The same pattern through the scan's transport: non-temporal global_load_lds_dwordx4 into a 4-step LDS ring (32 KiB a step and
// workgroup, three steps in flight), nothing read back.  PIECE 128: the 8-rows-per-instruction form of the private rings.
template <int PIECE>
__global__ __launch_bounds__(256) void read_bw_dma_kernel(const char* buf, int64_t n_rows, float* sink) {
  constexpr int ROW = 20480;
  constexpr int TILE_ROWS = 32768 / (PIECE == 128 ? 512 : PIECE), WAVE_ROWS = PIECE == 128 ? TILE_ROWS : TILE_ROWS / 4;
  constexpr int STEPS = ROW / (PIECE == 128 ? 512 : PIECE);
  __shared__ __attribute__((aligned(16))) char lds[4 * 32768];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_tiles = n_rows / TILE_ROWS;
  uint32_t so[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int x = i * 1024 + lane * 16;
    uint32_t off;
    if (PIECE == 128) off = (uint32_t)(x / 128) * ROW + (uint32_t)wave * 128u + (uint32_t)(x % 128);  // 8 rows x this wave's 128 B of the 512
    else off = (uint32_t)(wave * WAVE_ROWS + x / PIECE) * ROW + (uint32_t)(x % PIECE);
    so[i] = off + 4096u - (uint32_t)(i & 3) * 1024u;
  }
  int64_t tile = blockIdx.x;
  int step = 0;
  const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const int64_t total = my_tiles * STEPS;
  auto issue = [&](int slot_idx) {
    const char* tp = buf + (size_t)tile * TILE_ROWS * ROW + (size_t)step * (PIECE == 128 ? 512 : PIECE) - 4096;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
    const uint64_t tpu = ((uint64_t)hi << 32) | lo;
    const uint32_t slot = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + slot_idx * 32768 + wave * 8192));
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
        : "v"(so[0]), "v"(so[1]), "v"(so[2]), "v"(so[3]), "v"(so[4]), "v"(so[5]), "v"(so[6]), "v"(so[7]), "s"(slot), "s"(slot + 4096u), "s"(tpu)
        : "memory");
    if (++step == STEPS) { step = 0; tile += gridDim.x; }
  };
  for (int u = 0; u < 4; ++u)
    if (u < total) issue(u);
  for (int64_t s = 0; s < total; ++s) {
    if (s + 3 < total) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // step s landed (three steps x 8 DMAs behind it)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (s + 4 < total) issue((int)(s & 3));
  }
  __syncthreads();
  if (reinterpret_cast<const uint32_t*>(lds)[threadIdx.x] == 0x9e3779b9u) *sink = 1.0f;
}

// Transport-STRUCTURE probe (round 5, MV_CAL_STREAM_PROBE): 4-wave workgroups stream work units of `ct` 4 KiB tiles through
// wave-private nt LDS-DMA rings (4 slots, nothing read back) -- the float scan's and the FDE scan's transport with the shape of the
// work as the only variable:
//   own   0: a unit belongs to a WORKGROUP, tiles interleaved over its waves (w, w+4, ...: 16 KiB bursts)   [float scan]
//         1: a unit belongs to a workgroup, every wave a contiguous quarter
//         2: a unit belongs to ONE WAVE (four independent streams per workgroup)                            [FDE scan, round 5]
//   sched 0: one fresh workgroup per unit (own 2: per four units), handed out by the dispatcher
//         1: persistent workgroups, static order      2: persistent, units claimed from a device counter
// The ring drains at the end of every unit (as the scans' do at a page / chunk end).
//   own   3: a unit = `ct` ROWS of 20 KiB; wave w streams the w-th 5 KiB quarter of every row (5 DMAs of 1 KiB per row and wave, ring of
//            3 rows): the workgroup reads whole rows                                                            [FDE scan, block per row]
//   qload 1: every workgroup first loads 40 KiB from `qbuf` into registers (10 KiB per wave), as a scan that keeps a slice of the
//            query FDE per wave would, and stores one float per unit
__global__ __launch_bounds__(256) void stream_probe_kernel(const char* base, int64_t n_units, int ct, int own, int sched, uint32_t* work, float* sink,
                                                           const float* qbuf, float* unit_out) {
  constexpr int D = 4;
  __shared__ __attribute__((aligned(16))) char lds[4 * D * 4096 + 64];
  uint32_t* bcast = reinterpret_cast<uint32_t*>(lds + 4 * D * 4096);
  float qsum = 0.f;
  if (qbuf) {
    const float4* qp = reinterpret_cast<const float4*>(qbuf) + (threadIdx.x >> 6) * 640 + (threadIdx.x & 63);
    float4 v[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) v[i] = qp[i * 64];
#pragma unroll
    for (int i = 0; i < 10; ++i) qsum += v[i].x + v[i].w;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = lds + wave * (D * 4096);
  const int voff = lane * 16;
  const int64_t stride_units = own == 2 ? (int64_t)gridDim.x * 4 : (int64_t)gridDim.x;
  int64_t u = own == 2 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  for (int round = 0;; ++round, u += stride_units) {
    if (sched == 2) {
      if (own == 2) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(work, 1u);
        u = (int64_t)__builtin_amdgcn_readfirstlane(t);
      } else {
        __syncthreads();  // the previous round's readers of bcast are done
        if (threadIdx.x == 0) *bcast = atomicAdd(work, 1u);
        __syncthreads();
        u = (int64_t)*bcast;
      }
    } else if (sched == 0 && round > 0) {
      break;
    }
    if (u >= n_units) break;
    if (own == 3) {  // rows of 20 KiB, quarters of 5 KiB: slot r % 3 of this wave's 15 KiB ring
      const char* rb = base + (size_t)u * (size_t)ct * 20480 + wave * 5120;
      auto issue_row = [&](int r) {
        const char* tp = rb + (size_t)r * 20480;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
        const uint64_t tpu = ((uint64_t)hi << 32) | lo;
        const uint32_t slot = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (r % 3) * 5120));
        uint32_t keep;
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
            : "v"(voff), "s"(slot), "s"(slot + 4096u), "s"(tpu), "v"(voff + 4096)
            : "memory");
      };
      for (int r = 0; r < 2; ++r)
        if (r < ct) issue_row(r);
      for (int r = 0; r < ct; ++r) {
        if (r + 2 < ct) {
          issue_row(r + 2);
          asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else if (r + 1 < ct) {
          asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      if (unit_out && threadIdx.x == 0) unit_out[u] = qsum;
      continue;
    }
    const char* ub = base + (size_t)u * (size_t)ct * 4096;
    int t0 = 0, tstep = 1, ntw = ct;
    if (own == 0) { t0 = wave; tstep = 4; ntw = (ct - wave + 3) / 4; }
    else if (own == 1) { const int tq = (ct + 3) / 4; t0 = wave * tq; ntw = max(0, min(tq, ct - t0)); }
    auto issue = [&](int it) {
      const char* tp = ub + (size_t)(t0 + it * tstep) * 4096;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tp);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)tp >> 32));
      const uint64_t tpu = ((uint64_t)hi << 32) | lo;
      const uint32_t slot = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(ring + (it % D) * 4096));
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %3 nt\n\t"
          "global_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
          "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\t"
          "global_load_lds_dwordx4 %1, %3 offset:3072 nt\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff), "s"(slot), "s"(tpu)
          : "memory");
    };
    for (int i = 0; i < D - 1; ++i)
      if (i < ntw) issue(i);
    for (int it = 0; it < ntw; ++it) {
      if (it + D - 1 < ntw) {
        issue(it + D - 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else {
        const int left = ntw - 1 - it;
        if (left >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (unit_out && lane == 0 && (own == 2 || wave == 0)) unit_out[u] = qsum;
  }
  if (sched == 2 && lane == 0 && (own == 2 || wave == 0)) {  // the last claimer re-arms the counters
    const uint32_t total = own == 2 ? gridDim.x * 4 : gridDim.x;
    const uint32_t done = atomicAdd(work + 1, 1u);
    if (done == total - 1u) {
      __threadfence();
      work[0] = 0u;
      work[1] = 0u;
    }
  }
  __syncthreads();
  if (reinterpret_cast<const uint32_t*>(lds)[threadIdx.x] == 0x9e3779b9u) *sink = 1.0f;
}

int launch_stream_probe(const void* d_buf, int64_t bytes, int ct, int own, int sched, int blocks_per_cu, uint32_t* d_work, float* d_sink, hipStream_t s,
                        const float* d_q, float* d_unit_out) {
  if (ct < 1 || own < 0 || own > 3 || sched < 0 || sched > 2) { set_error("stream probe: bad shape"); return MV_ERR_INVALID; }
  const int64_t n_units = bytes / ((int64_t)ct * (own == 3 ? 20480 : 4096));
  if (n_units < 1) { set_error("stream probe: buffer smaller than one unit"); return MV_ERR_INVALID; }
  const int64_t blocks_needed = own == 2 ? (n_units + 3) / 4 : n_units;
  const int64_t persistent = 256LL * (blocks_per_cu > 0 ? blocks_per_cu : 2);
  const int64_t grid = sched == 0 ? blocks_needed : std::min(blocks_needed, persistent);
  hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<const char*>(d_buf), n_units, ct, own, sched, d_work, d_sink, d_q, d_unit_out);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_read_bw_strided(const void* d_buf, int64_t n_rows, int piece, float* d_sink, hipStream_t s) {
  const char* b = reinterpret_cast<const char*>(d_buf);
  switch (piece) {
    case 512: hipLaunchKernelGGL(read_bw_strided_kernel<512>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case 1024: hipLaunchKernelGGL(read_bw_strided_kernel<1024>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case 2048: hipLaunchKernelGGL(read_bw_strided_kernel<2048>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case 4096: hipLaunchKernelGGL(read_bw_strided_kernel<4096>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case 20480: hipLaunchKernelGGL(read_bw_strided_kernel<20480>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case -128: hipLaunchKernelGGL(read_bw_dma_kernel<128>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;  // negative: through the LDS-DMA
    case -512: hipLaunchKernelGGL(read_bw_dma_kernel<512>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case -1024: hipLaunchKernelGGL(read_bw_dma_kernel<1024>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    case -2048: hipLaunchKernelGGL(read_bw_dma_kernel<2048>, dim3(256), dim3(256), 0, s, b, n_rows, d_sink); break;
    default: set_error("strided read calibration: piece %d", piece); return MV_ERR_INVALID;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_read_bw_nt(const void* d_buf, int64_t bytes, float* d_sink, hipStream_t s) {
  hipLaunchKernelGGL(read_bw_nt_kernel, dim3(256 * 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_buf), bytes / 16384, d_sink);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_mfma_peak(int blocks, int iters, int shape, float* d_sink, hipStream_t s) {
  if (shape == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, iters, d_sink);
  else hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, iters, d_sink);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_synth_rows(uint16_t* d_out, uint64_t seed, uint64_t first_unit, int64_t n_units, int32_t n_rows,
                      int32_t stride_rows, hipStream_t s) {
  if (n_units <= 0) return MV_OK;
  // A launch's total work-item count must stay below 2^32 (32-bit AQL grid): 32 threads per row,
  // so at most 2^26 rows per launch.
  const int64_t max_units = std::max<int64_t>(((int64_t)1 << 26) / stride_rows, 1);
  int64_t done = 0;
  while (done < n_units) {
    const int64_t units = std::min(n_units - done, max_units);
    const int64_t rows = units * (int64_t)stride_rows;
    const int64_t blocks = ((rows + 1) / 2 + 3) / 4;
    hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       d_out + done * (int64_t)stride_rows * kDim, seed, first_unit + (uint64_t)done, units, n_rows,
                       stride_rows);
    done += units;
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_synth_rows_ragged(uint16_t* d_base, uint64_t seed, uint64_t first_unit, int64_t n_units, const int32_t* d_n_rows,
                             const int64_t* d_row_off, int32_t stride_rows, hipStream_t s) {
  // one block per unit; grid.x chunks of 2^22 units (the table pointers move with the chunk; d_base does so only in the fixed layout)
  for (int64_t done = 0; done < n_units; done += (int64_t)1 << 22) {
    const int64_t c = std::min<int64_t>(n_units - done, (int64_t)1 << 22);
    hipLaunchKernelGGL(synth_rows_ragged_kernel, dim3((unsigned)c), dim3(256), 0, s, d_row_off ? d_base : d_base + done * (int64_t)stride_rows * kDim, seed,
                       first_unit + (uint64_t)done, d_n_rows + done, d_row_off ? d_row_off + done : nullptr, stride_rows);
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_f32_to_bf16(const float* d_in, uint16_t* d_out, int64_t n, hipStream_t s) {
  if (n <= 0) return MV_OK;
  const int64_t per = (int64_t)1 << 32;  // elements per launch (2^30 threads x 4)
  for (int64_t o = 0; o < n; o += per) {
    const int64_t m = std::min(per, n - o);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(((m + 3) / 4 + 255) / 256)), dim3(256), 0, s, d_in + o, d_out + o, m);
  }
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_scatter_rows(const void* d_src, int dtype, const int64_t* d_row_offsets, int64_t n_pages, int32_t stride,
                        uint16_t* d_slab_pages, hipStream_t s, int32_t* d_nonfinite, uint16_t* d_slab_lo_pages, const int64_t* d_dst_row_off) {
  if (n_pages <= 0) return MV_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)n_pages), dim3(256), 0, s, d_src, dtype, d_row_offsets, stride,
                     d_slab_pages, d_nonfinite, d_slab_lo_pages, d_dst_row_off);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

int launch_read_bw(const void* d_buf, int64_t bytes, float* d_sink, hipStream_t s) {
  hipLaunchKernelGGL(read_bw_kernel, dim3(256 * 8), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_buf), bytes / 16,
                     d_sink);
  MV_HIP(hipGetLastError());
  return MV_OK;
}

}  // namespace mv
