// Aligner attention map for gfx950: att[b][s][t] = -|| x[b][t][:] - mel[b][s][:] ||_2.
//
// Replaces reference models/tacotron.py:63-64, which materialises the [B,S,T,D] difference tensor
// (B=32, S=1250, T=190, D=32: 973 MB) before taking the norm; here the differences only ever exist in
// registers.  Bound: HBM writes of the [B,S,T] map (algorithmic bytes = 4*B*(S*T + (S+T)*D)).
#include "common.h"
#include "../../include/wrnn_mi355.h"

namespace {

using v4f = __attribute__((ext_vector_type(4))) float;
using v2f = __attribute__((ext_vector_type(2))) float;

constexpr int kTile = 64;        // frames x tokens per workgroup; thread = 4 frames x 4 tokens
constexpr int kDC = 32;          // feature dims staged per pass
constexpr int kLd = kTile + 4;   // LDS row stride (floats): keeps float4 reads aligned, spreads the transposed writes

__global__ __launch_bounds__(256) void aligner_distance_kernel(const float* __restrict__ mel, const float* __restrict__ x,
                                                               int S, int T, int D, int s_tiles, int t_tiles, int vec,
                                                               float* __restrict__ att) {
  __shared__ __attribute__((aligned(16))) float ms[kDC][kLd];  // [d][frame]
  __shared__ __attribute__((aligned(16))) float xs[kDC][kLd];  // [d][token]
  int bid = blockIdx.x;
  const int tt = bid % t_tiles;
  bid /= t_tiles;
  const int st = bid % s_tiles, b = bid / s_tiles;
  const int s0 = st * kTile, t0 = tt * kTile, tid = threadIdx.x;
  const int fs = (tid >> 4) * 4, ft = (tid & 15) * 4;  // the thread's 4 frames / 4 tokens inside the tile
  const float* mrow = mel + (size_t)b * S * D;
  const float* xrow = x + (size_t)b * T * D;
  // Accumulators as float2 pairs over tokens: sub and fma compile to the packed fp32 VALU forms (v_pk_add_f32,
  // v_pk_fma_f32), two lanes of arithmetic per instruction; the kernel is VALU-bound otherwise (3*B*S*T*D flops on
  // the vector pipe take longer than writing the map to HBM).
  v2f acc[4][2] = {};
  for (int d0 = 0; d0 < D; d0 += kDC) {
    const int dc = min(kDC, D - d0);
    if (d0) __syncthreads();
    // coalesced along the feature dim; rows past the end repeat the last row (their results are never stored)
    for (int i = tid; i < kTile * dc; i += 256) {
      const int r = dc == kDC ? i / kDC : i / dc, d = i - r * dc;
      ms[d][r] = mrow[(size_t)min(s0 + r, S - 1) * D + d0 + d];
      xs[d][r] = xrow[(size_t)min(t0 + r, T - 1) * D + d0 + d];
    }
    __syncthreads();
    auto step = [&](int d) {
      const v4f m4 = *reinterpret_cast<const v4f*>(&ms[d][fs]);
      const v4f x4 = *reinterpret_cast<const v4f*>(&xs[d][ft]);
      const v2f xa = {x4[0], x4[1]}, xb = {x4[2], x4[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v2f mi = {m4[i], m4[i]};
        const v2f da = xa - mi, db = xb - mi;
        acc[i][0] = __builtin_elementwise_fma(da, da, acc[i][0]);
        acc[i][1] = __builtin_elementwise_fma(db, db, acc[i][1]);
      }
    };
    if (dc == kDC) {
#pragma unroll
      for (int d = 0; d < kDC; ++d) step(d);
    } else {
      for (int d = 0; d < dc; ++d) step(d);
    }
  }
  float* orow = att + ((size_t)b * S + s0 + fs) * T + t0 + ft;
  // vec: T % 4 == 0 and a 16-byte aligned map, so every row start keeps the float4 aligned and ft + 3 < T iff ft < T
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (s0 + fs + i >= S) break;
    v4f v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = -sqrtf(acc[i][j >> 1][j & 1]);
    float* o = orow + (size_t)i * T;
    if (vec) {
      if (t0 + ft < T) *reinterpret_cast<v4f*>(o) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t0 + ft + j < T) o[j] = v[j];
    }
  }
}

}  // namespace

extern "C" int wrnn_aligner_distance(const float* mel_feats, const float* text_feats, int B, int S, int T, int D,
                                     float* att, void* stream) {
  WRNN_REQUIRE(B >= 0 && S >= 0 && T >= 0 && D >= 0, "wrnn_aligner_distance: negative size (B=%d, S=%d, T=%d, D=%d)", B, S,
               T, D);
  if (B == 0 || S == 0 || T == 0) return 0;
  WRNN_REQUIRE(att && (D == 0 || (mel_feats && text_feats)), "wrnn_aligner_distance: null pointer");
  if (D == 0) {  // the norm of an empty vector is 0
    WRNN_HIP(hipMemsetAsync(att, 0, sizeof(float) * (size_t)B * S * T, (hipStream_t)stream));
    return 0;
  }
  const long s_tiles = (S + kTile - 1) / kTile, t_tiles = (T + kTile - 1) / kTile;
  const long blocks = (long)B * s_tiles * t_tiles;
  WRNN_REQUIRE(blocks < (1L << 31), "wrnn_aligner_distance: B*S*T too large for one launch");
  const int vec = (T & 3) == 0 && (reinterpret_cast<uintptr_t>(att) & 15) == 0;
  hipLaunchKernelGGL(aligner_distance_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mel_feats,
                     text_feats, S, T, D, (int)s_tiles, (int)t_tiles, vec, att);
  WRNN_LAUNCH_CHECK();
  return 0;
}
