// Kernels shared by the bidirectional recurrences (bilstm.hip, bigru.hip): step-major packing of the
// input with the reverse direction aligned to each row's own last frame, padding of the output past a
// row's length, and the final-state re-layout.
#pragma once
#include "linjob.h"

namespace {

// xf[s][k4][b] = x[b][s][k4] ; xr[s][k4][b] = x[b][len_b-1-s][k4] (zero past the row's length).
// One thread per destination float4; consecutive threads walk b, so every store is coalesced and
// every load is one 16-B piece of a different row.
__global__ void birnn_pack_kernel(const f32x4* __restrict__ x, const int32_t* __restrict__ lens, int B, int M, int K4,
                                   f32x4* __restrict__ xf, f32x4* __restrict__ xr) {
  const size_t total = (size_t)M * K4 * B;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    const size_t r = i / B;
    const int k4 = (int)(r % K4), s = (int)(r / K4);
    const int len = lens ? min(max(lens[b], 0), M) : M;
    const f32x4* row = x + (size_t)b * M * K4 + k4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    xf[i] = s < len ? row[(size_t)s * K4] : z;
    xr[i] = s < len ? row[(size_t)(len - 1 - s) * K4] : z;
  }
}

// frames at or past a row's length hold the padding value (pad_packed_sequence, forward_tacotron.py:152)
__global__ void birnn_pad_kernel(const int32_t* __restrict__ lens, int B, int M, int H2q, float pad,
                                  f32x4* __restrict__ out) {
  const size_t total = (size_t)B * M * H2q;
  const f32x4 pv = {pad, pad, pad, pad};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bm = i / H2q;
    const int t = (int)(bm % M), b = (int)(bm / M);
    if (t >= min(max(lens[b], 0), M)) out[i] = pv;
  }
}

// final states, interleaved [H/4][B][4] -> row-major [B][H]
__global__ void birnn_state_kernel(const float* __restrict__ src, int B, int H, float* __restrict__ dst) {
  const int total = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / H, n = i - b * H;
    dst[i] = src[((size_t)(n >> 2) * B + b) * 4 + (n & 3)];
  }
}

}  // namespace
