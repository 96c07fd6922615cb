// Check of seam.h's wave_sum4 (four wave sums at once through v_permlane32_swap / v_permlane16_swap + DPP) against a plain
// shuffle reduction.  Build: hipcc --offload-arch=gfx950 -O3 -I../../wavernn_amd/csrc -o wave_sum4_check wave_sum4_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "seam.h"

__global__ void k(const float* in, float* out) {
  const int l = threadIdx.x;
  float a = in[l], b = in[64 + l], c = in[128 + l], d = in[192 + l];
  float ra = a, rb = b, rc = c, rd = d;
  for (int off = 32; off >= 1; off >>= 1)
    ra += __shfl_xor(ra, off, 64), rb += __shfl_xor(rb, off, 64), rc += __shfl_xor(rc, off, 64), rd += __shfl_xor(rd, off, 64);
  wave_sum4(a, b, c, d);
  out[l * 8 + 0] = a, out[l * 8 + 1] = b, out[l * 8 + 2] = c, out[l * 8 + 3] = d;
  out[l * 8 + 4] = ra, out[l * 8 + 5] = rb, out[l * 8 + 6] = rc, out[l * 8 + 7] = rd;
}

int main() {
  float h[256], *din, *dout, o[512];
  for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 101) + (i / 64) * 1000.0f;
  hipMalloc(&din, sizeof(h)), hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v)
      if (o[l * 8 + v] != o[l * 8 + 4 + v]) ++bad;
  printf("wave_sum4: %d mismatches (sums %.0f %.0f %.0f %.0f, expected %.0f %.0f %.0f %.0f)\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return bad != 0;
}
