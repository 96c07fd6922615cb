// Mixture-of-logistics sampling for one (batch row, time step): the arithmetic of the reference's
// sample_from_discretized_mix_logistic (utils/distribution.py:87-123), with the function's two internal
// uniform_(1e-5, 1 - 1e-5) draws (:106, :118) supplied by the caller.  Shared by the stand-alone sampler (mol.hip) and
// by the WaveRNN generation loop's MOL mode (wavernn.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wrnn_mol {

constexpr float kDefaultLogScaleMin = -32.23619130191664f;  // float(np.log(1e-14)), distribution.py:96-97

// Gumbel-max score of one mixture component (:107): logit - log(-log(u)).  logf / expf are the accurate device-library
// versions (not the hardware approximations): the mixture pick is an argmax over these values and has to agree with the
// reference's fp32 evaluation wherever two components are not within an ulp or two of each other.
__device__ __forceinline__ float gumbel_score(float logit, float u) { return logit - logf(-logf(u)); }

// Logistic sample of the picked component (:113-121): clamp(mean + exp(max(log_scale, min)) * (log u - log(1 - u)), -1, 1)
__device__ __forceinline__ float logistic_sample(float mean, float log_scale, float u, float log_scale_min) {
  const float ls = fmaxf(log_scale, log_scale_min);
  const float x = mean + expf(ls) * (logf(u) - logf(1.0f - u));
  return fminf(fmaxf(x, -1.0f), 1.0f);
}

}  // namespace wrnn_mol
