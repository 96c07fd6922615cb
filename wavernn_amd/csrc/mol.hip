// Stand-alone mixture-of-logistics sampler for gfx950: replaces the reference's
// sample_from_discretized_mix_logistic (utils/distribution.py:87-123) -- the one piece of the WaveRNN path that
// survives in the mounted tree -- for a whole [B, 3*nr_mix, T] tensor in one HBM-streaming launch: a thread per
// (b, t), reads coalesced along t (the reference's own layout), algorithmic bytes 4 * B * T * (4 * nr_mix + 2).
#include "common.h"
#include "mol.h"
#include "../../include/wrnn_mi355.h"

namespace {

__global__ __launch_bounds__(256) void mol_sample_kernel(const float* __restrict__ y, const float* __restrict__ u_mix,
                                                         const float* __restrict__ u_log, int nr_mix, int T,
                                                         float log_scale_min, float* __restrict__ out,
                                                         int32_t* __restrict__ mix_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const float* yb = y + (size_t)b * 3 * nr_mix * T + t;
  const float* um = u_mix + ((size_t)b * T + t) * nr_mix;
  float best = -INFINITY;
  int k = 0;
  for (int i = 0; i < nr_mix; ++i) {  // first maximum wins, as torch.max(dim) on the CPU does
    const float s = wrnn_mol::gumbel_score(yb[(size_t)i * T], um[i]);
    if (s > best) best = s, k = i;
  }
  const float mean = yb[(size_t)(nr_mix + k) * T], ls = yb[(size_t)(2 * nr_mix + k) * T];
  out[(size_t)b * T + t] = wrnn_mol::logistic_sample(mean, ls, u_log[(size_t)b * T + t], log_scale_min);
  if (mix_out) mix_out[(size_t)b * T + t] = k;
}

}  // namespace

extern "C" int wrnn_mol_sample(const float* y, const float* u_mix, const float* u_logistic, int B, int nr_mix, int T,
                               float log_scale_min, float* out, int32_t* mix_out, void* stream) {
  WRNN_REQUIRE(B >= 0 && T >= 0 && nr_mix >= 1, "wrnn_mol_sample: bad size (B=%d, nr_mix=%d, T=%d)", B, nr_mix, T);
  WRNN_REQUIRE(B <= 65535, "wrnn_mol_sample: B=%d exceeds grid.y limit", B);
  if (B == 0 || T == 0) return 0;
  WRNN_REQUIRE(y && u_mix && u_logistic && out, "wrnn_mol_sample: null pointer");
  hipLaunchKernelGGL(mol_sample_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, y, u_mix, u_logistic,
                     nr_mix, T, log_scale_min, out, mix_out);
  WRNN_LAUNCH_CHECK();
  return 0;
}
