// Tacotron decoder step / loops for gfx950 (MI355X), fp32 storage and fp32 accumulate.
//
// Replaces reference models/tacotron.py:86-100 (PreNet), :122-156 (LSA), :159-235
// (Decoder.forward) and the decoder loop of Tacotron.forward (:299-316,:334-348,:356).
//
// v1 structure ("stage kernels"): every dependency stage of a step is one launch.
//   linear16_kernel<NB,Epi>  small-batch GEMM on v_mfma_f32_16x16x4_f32 with the stage's
//                            pointwise work fused into the epilogue (bias/relu/dropout,
//                            GRU gates, LSTM gates + zoneout + residual, mel scatter)
//   lsa_energy_kernel        location conv + L projection + tanh + v-dot  (one (b, T-chunk) per WG)
//   lsa_softmax_ctx_kernel   softmax over T, LSA state update, context = s . enc_q
// Weights are re-laid-out once (wrnn_decoder_create) into MFMA fragment order, so that a
// wave-instruction reads 1 KiB contiguous, with gate rows interleaved per hidden unit so
// that the 4 accumulator registers of a lane are the 4 gates of one (unit, batch) pair.
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "../../include/wrnn_mi355.h"
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kFc1 = 256, kFc2 = 128;       // reference PreNet defaults (tacotron.py:87)
constexpr int kFilters = 32, kTaps = 31;    // reference LSA defaults (tacotron.py:123)
constexpr int kMaxR = 20;                   // Decoder.max_r (tacotron.py:162)
constexpr int kD = 256;                     // decoder_dims (forced by the reference, see header)
constexpr float kDropP = 0.5f, kZoneP = 0.1f;

// ----------------------------------------------------------------------------- math
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(x) = 1 - 2/(exp(2x)+1); absolute error ~1e-7, saturates correctly for |x| large
__device__ __forceinline__ float tanhf_(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// ----------------------------------------------------------------------------- linear16
struct Src {
  const float* p;  // row-major [rows][ld]
  int ld;
  int k;  // columns used (multiple of 4)
};

struct EpiLinear {
  const float* bias;  // [N] or null
  float* y;
  int ldy, N, relu;
  const float* drop_u;  // [rows][ldu] or null
  int ldu;
  __device__ __forceinline__ void operator()(int tile, int u, int row, f32x4 v) const {
    const int n = tile * 16 + u * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < N) {
        float val = v[e] + (bias ? bias[n + e] : 0.0f);
        if (relu) val = fmaxf(val, 0.0f);
        if (drop_u) val = (drop_u[(size_t)row * ldu + n + e] >= kDropP) ? val * (1.0f / (1.0f - kDropP)) : 0.0f;
        y[(size_t)row * ldy + n + e] = val;
      }
    }
  }
};

struct EpiGru {  // tile rows: 4 units x (r, z, gi_n, gh_n)
  const float* bias;  // tile order
  const float* h_prev;
  float* h_out;
  int d;
  __device__ __forceinline__ void operator()(int tile, int u, int row, f32x4 v) const {
    const float* bb = bias + tile * 16 + u * 4;
    const int U = tile * 4 + u;
    const float r = sigmoidf_(v[0] + bb[0]);
    const float z = sigmoidf_(v[1] + bb[1]);
    const float n = tanhf_(v[2] + bb[2] + r * (v[3] + bb[3]));
    const float h = h_prev[(size_t)row * d + U];
    h_out[(size_t)row * d + U] = (1.0f - z) * n + z * h;
  }
};

struct EpiLstm {  // tile rows: 4 units x (i, f, g, o)
  const float* bias;  // tile order, b_ih + b_hh
  const float* c_prev;
  float* c_out;
  const float* h_prev;
  float* h_out;
  const float* x_in;  // residual input
  float* x_out;       // x_in + h
  const float* zone_u;
  int l;
  __device__ __forceinline__ void operator()(int tile, int u, int row, f32x4 v) const {
    const float* bb = bias + tile * 16 + u * 4;
    const size_t o = (size_t)row * l + tile * 4 + u;
    const float ig = sigmoidf_(v[0] + bb[0]);
    const float fg = sigmoidf_(v[1] + bb[1]);
    const float gg = tanhf_(v[2] + bb[2]);
    const float og = sigmoidf_(v[3] + bb[3]);
    const float c = fg * c_prev[o] + ig * gg;
    float h = og * tanhf_(c);
    if (zone_u && zone_u[o] < kZoneP) h = h_prev[o];  // zoneout keeps the previous hidden value
    c_out[o] = c;
    h_out[o] = h;
    x_out[o] = x_in[o] + h;
  }
};

struct EpiMel {  // weight rows gathered as j = m*r + f  <->  mel_proj row m*20 + f
  float* out;
  long sb, sm;  // strides of the batch and mel dims of `out`
  int r, N, B;  // rows are (k, b) = (row / B, row % B); frame index = k*r + f
  __device__ __forceinline__ void operator()(int tile, int u, int row, f32x4 v) const {
    const int k = row / B, b = row - k * B;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = tile * 16 + u * 4 + e;
      if (j < N) {
        const int m = j / r, f = j - m * r;
        out[(size_t)b * sb + (size_t)m * sm + k * r + f] = v[e];
      }
    }
  }
};

// D[16 weight rows][16*NB batch rows] += W_tile . X^T, K split over the 4 waves of the WG.
// A operand = weights (lane: row i = lane&15, k-slot s = lane>>4), B operand = X^T
// (lane: batch j = lane&15, k-slot s).  Inside a 16-wide K chunk slot s owns
// k = 16*kc + 4*s + m for MFMA m = 0..3, so both operands are one float4 per lane per chunk.
template <int NB, class Epi>
__global__ __launch_bounds__(256) void linear16_kernel(const float4* __restrict__ wt, int KC, Src s0,
                                                        Src s1, Src s2, int rows, Epi epi) {
  __shared__ f32x4 red[4][NB][64];
  const int tile = blockIdx.x;
  const int row0 = blockIdx.y * (16 * NB);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, s = lane >> 4;
  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float4* wp = wt + (size_t)tile * KC * 64 + lane;
  for (int kc = wave; kc < KC; kc += 4) {
    const float4 w = wp[(size_t)kc * 64];
    int k = kc * 16 + s * 4;
    const float* src = s0.p;
    int ld = s0.ld;
    bool valid = true;
    if (k >= s0.k) {
      k -= s0.k;
      src = s1.p;
      ld = s1.ld;
      if (k >= s1.k) {
        k -= s1.k;
        src = s2.p;
        ld = s2.ld;
        valid = k < s2.k;
      }
    }
    float4 x[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int row = row0 + nb * 16 + j;
      x[nb] = (valid && row < rows) ? *reinterpret_cast<const float4*>(src + (size_t)row * ld + k)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x[nb].x, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x[nb].y, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x[nb].z, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x[nb].w, acc[nb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) red[wave][nb][lane] = acc[nb];
  __syncthreads();
  for (int idx = threadIdx.x; idx < NB * 64; idx += 256) {
    const int nb = idx >> 6, l = idx & 63;
    const f32x4 v = (red[0][nb][l] + red[1][nb][l]) + (red[2][nb][l] + red[3][nb][l]);
    const int row = row0 + nb * 16 + (l & 15);
    if (row < rows) epi(tile, l >> 4, row, v);  // lane holds tile rows 4u..4u+3, u = l>>4
  }
}

// ----------------------------------------------------------------------------- LSA
constexpr int kTC = 16;  // attention positions per workgroup

// grid (ceil(T/kTC), B), 256 threads (= decoder_dims).  u[b,t] = v . tanh(pq + enc_p + L conv(loc) + bL)
__global__ __launch_bounds__(256) void lsa_energy_kernel(
    const float* __restrict__ enc_p, const float* __restrict__ pq, const float* __restrict__ cum,
    const float* __restrict__ att, int first, const float* __restrict__ conv_t /*[2][31][32]*/,
    const float* __restrict__ L_w /*[256][32]*/, const float* __restrict__ L_b,
    const float* __restrict__ v_w, float* __restrict__ u_out, int T) {
  __shared__ float loc[2][kTC + kTaps - 1];
  __shared__ __attribute__((aligned(16))) float f[kTC][kFilters];
  __shared__ float red[kTC][kD + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * kTC, tid = threadIdx.x;
  constexpr int W = kTC + kTaps - 1;
  if (tid < 2 * W) {
    const int c = tid / W, p = tid - c * W;
    const int tp = t0 + p - (kTaps - 1) / 2;
    float val = 0.0f;
    if (!first && tp >= 0 && tp < T) val = (c == 0 ? cum : att)[(size_t)b * T + tp];
    loc[c][p] = val;
  }
  __syncthreads();
  // location conv (cross-correlation, no bias): kTC*32 outputs, 2 per thread
#pragma unroll
  for (int q = 0; q < (kTC * kFilters) / 256; ++q) {
    const int o = tid + q * 256;
    const int tt = o >> 5, fl = o & 31;
    float a = 0.0f;
    for (int c = 0; c < 2; ++c)
      for (int k = 0; k < kTaps; ++k) a = fmaf(conv_t[(c * kTaps + k) * kFilters + fl], loc[c][tt + k], a);
    f[tt][fl] = a;
  }
  __syncthreads();
  const int d = tid;
  float Lrow[kFilters];
#pragma unroll
  for (int q = 0; q < kFilters / 4; ++q) {
    const float4 t4 = reinterpret_cast<const float4*>(L_w + (size_t)d * kFilters)[q];
    Lrow[4 * q] = t4.x, Lrow[4 * q + 1] = t4.y, Lrow[4 * q + 2] = t4.z, Lrow[4 * q + 3] = t4.w;
  }
  const float base = pq[(size_t)b * kD + d] + L_b[d];
  const float vd = v_w[d];
#pragma unroll 4
  for (int tt = 0; tt < kTC; ++tt) {
    const int t = t0 + tt;
    float e = 0.0f;
    if (t < T) {
      float a = 0.0f;
#pragma unroll
      for (int q = 0; q < kFilters / 4; ++q) {
        const float4 f4 = reinterpret_cast<const float4*>(&f[tt][0])[q];
        a = fmaf(Lrow[4 * q], f4.x, a);
        a = fmaf(Lrow[4 * q + 1], f4.y, a);
        a = fmaf(Lrow[4 * q + 2], f4.z, a);
        a = fmaf(Lrow[4 * q + 3], f4.w, a);
      }
      a += base + enc_p[((size_t)b * T + t) * kD + d];
      e = vd * tanhf_(a);
    }
    red[tt][d] = e;
  }
  __syncthreads();
  // sum over d: 16 threads per position, then a 16-lane shuffle tree
  {
    const int tt = tid >> 4, part = tid & 15;
    float a = 0.0f;
#pragma unroll
    for (int q = 0; q < kD / 16; ++q) a += red[tt][part + 16 * q];
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) a += __shfl_xor(a, off, 16);
    if (part == 0 && t0 + tt < T) u_out[(size_t)b * T + t0 + tt] = a;
  }
}

__device__ __forceinline__ float block_reduce_256(float v, bool is_max, float* sh /*[4]*/) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();  // protect sh reuse
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// grid (kD/64, B), 256 threads.  softmax over ALL T (no mask, tacotron.py:147-152), LSA state
// update (:153-154), context quarter = sum_t s[t] * enc_q[b,t,dq*64 + lane].
__global__ __launch_bounds__(256) void lsa_softmax_ctx_kernel(
    const float* __restrict__ u, const float* __restrict__ enc_q, float* __restrict__ cum,
    float* __restrict__ att, int first, float* __restrict__ scores_out, long score_stride,
    float* __restrict__ ctx_out, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s = reinterpret_cast<float*>(smem);  // [T]
  float* sh = s + T;                          // [4]
  float* part = sh + 4;                       // [4][64]
  const int b = blockIdx.y, dq = blockIdx.x, tid = threadIdx.x;
  float mx = -INFINITY;
  for (int t = tid; t < T; t += 256) {
    const float val = u[(size_t)b * T + t];
    s[t] = val;
    mx = fmaxf(mx, val);
  }
  mx = block_reduce_256(mx, true, sh);
  float sum = 0.0f;
  for (int t = tid; t < T; t += 256) {
    const float e = __expf(s[t] - mx);
    s[t] = e;
    sum += e;
  }
  sum = block_reduce_256(sum, false, sh);
  for (int t = tid; t < T; t += 256) {
    const float p = s[t] / sum;
    s[t] = p;
    if (dq == 0) {
      const size_t o = (size_t)b * T + t;
      scores_out[(size_t)b * score_stride + t] = p;
      att[o] = p;
      cum[o] = first ? p : cum[o] + p;
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const float* eq = enc_q + (size_t)b * T * kD + dq * 64 + lane;
  float a = 0.0f;
  for (int t = wave; t < T; t += 4) a = fmaf(s[t], eq[(size_t)t * kD], a);
  part[wave * 64 + lane] = a;
  __syncthreads();
  if (wave == 0)
    ctx_out[(size_t)b * kD + dq * 64 + lane] =
        (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
}

// ----------------------------------------------------------------------------- loop helpers
// Xp[k][b][c] = mel_in[b][c][k*r - 1] (zeros for k == 0): the teacher-forcing inputs of every step
__global__ void gather_prenet_in_kernel(const float* __restrict__ mel_in, float* __restrict__ xp, int B,
                                        int P, int S, int r, int n) {
  const size_t total = (size_t)n * B * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % P;
    const size_t kb = i / P;
    const int b = kb % B, k = kb / B;
    xp[i] = k == 0 ? 0.0f : mel_in[((size_t)b * P + c) * S + (size_t)k * r - 1];
  }
}

// free-running feedback: next prenet input and the stop vote of group k
__global__ void feedback_kernel(const float* __restrict__ mel_out, long sb, long sm, int k, int r,
                                int n_mels, const float* __restrict__ ali, int A, int max_steps,
                                float* __restrict__ pin_next, float thr, int* __restrict__ alive) {
  const int b = blockIdx.x;
  const int P = n_mels + A;
  bool any_alive = false;
  for (int c = threadIdx.x; c < P; c += blockDim.x) {
    float val;
    if (c < n_mels) {
      const float* row = mel_out + (size_t)b * sb + (size_t)c * sm + (size_t)k * r;
      for (int f = 0; f < r; ++f) any_alive |= !(row[f] < thr);
      val = row[r - 1];
    } else {
      val = ali ? ali[((size_t)b * A + (c - n_mels)) * max_steps + k] : 0.0f;
    }
    pin_next[(size_t)b * P + c] = val;
  }
  if (any_alive) atomicOr(alive + k, 1);
}

}  // namespace

// ============================================================================ host side
namespace {

struct Layer {
  float4* wt = nullptr;   // [tiles][KC][64] float4
  float* bias = nullptr;  // tile order [tiles*16] (or [N] for plain linear) or null
  int tiles = 0, KC = 0, N = 0;
};

using RowFn = std::function<float(int tile, int i, int k)>;

int upload(const void* host, size_t bytes, void** dev) {
  WRNN_HIP(hipMalloc(dev, bytes));
  WRNN_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

int build_layer(Layer& L, int tiles, int K, int N, const RowFn& w, const std::function<float(int, int)>* bias) {
  L.tiles = tiles;
  L.KC = (K + 15) / 16;
  L.N = N;
  std::vector<float> h((size_t)tiles * L.KC * 64 * 4);
  size_t o = 0;
  for (int t = 0; t < tiles; ++t)
    for (int kc = 0; kc < L.KC; ++kc)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, s = lane >> 4;
        for (int m = 0; m < 4; ++m) {
          const int k = kc * 16 + s * 4 + m;
          h[o++] = k < K ? w(t, i, k) : 0.0f;
        }
      }
  if (upload(h.data(), h.size() * sizeof(float), (void**)&L.wt)) return 1;
  if (bias) {
    std::vector<float> hb((size_t)tiles * 16);
    for (int t = 0; t < tiles; ++t)
      for (int i = 0; i < 16; ++i) hb[t * 16 + i] = (*bias)(t, i);
    if (upload(hb.data(), hb.size() * sizeof(float), (void**)&L.bias)) return 1;
  }
  return 0;
}

void free_layer(Layer& L) {
  if (L.wt) (void)hipFree(L.wt);
  if (L.bias) (void)hipFree(L.bias);
  L = Layer();
}

template <class Epi>
int launch_linear(const Layer& L, Src s0, Src s1, Src s2, int rows, const Epi& epi, hipStream_t st) {
  if (rows <= 0) return 0;
  WRNN_REQUIRE(s0.k % 4 == 0 && s1.k % 4 == 0 && s2.k % 4 == 0, "linear16: source widths must be multiples of 4");
  WRNN_REQUIRE(s0.k + s1.k + s2.k <= L.KC * 16 && s0.k + s1.k + s2.k > (L.KC - 1) * 16,
               "linear16: source widths %d+%d+%d do not match the layer's K", s0.k, s1.k, s2.k);
  if (rows <= 16) {
    hipLaunchKernelGGL((linear16_kernel<1, Epi>), dim3(L.tiles, 1), dim3(256), 0, st, L.wt, L.KC, s0, s1, s2, rows, epi);
  } else if (rows < 1024) {
    hipLaunchKernelGGL((linear16_kernel<2, Epi>), dim3(L.tiles, (rows + 31) / 32), dim3(256), 0, st, L.wt, L.KC, s0, s1, s2, rows, epi);
  } else {
    WRNN_REQUIRE((rows + 63) / 64 <= 65535, "linear16: too many rows (%d)", rows);
    hipLaunchKernelGGL((linear16_kernel<4, Epi>), dim3(L.tiles, (rows + 63) / 64), dim3(256), 0, st, L.wt, L.KC, s0, s1, s2, rows, epi);
  }
  WRNN_LAUNCH_CHECK();
  return 0;
}

const Src kNoSrc{nullptr, 0, 0};

}  // namespace

struct wrnn_decoder {
  int device = 0;
  int n_mels = 0, d = 0, l = 0, A = 0, P = 0;
  Layer fc1, fc2, gru, pq, rin, l1, l2;
  Layer mel[kMaxR + 1];
  std::vector<float> mel_w_host;  // [n_mels*20][l] for the lazy per-r gather
  float *conv_t = nullptr, *L_w = nullptr, *L_b = nullptr, *v_w = nullptr;
  float* ws = nullptr;
  size_t ws_floats = 0;
  int* flags = nullptr;  // free-run stop votes
  int flags_n = 0;
};

namespace {

int ensure_ws(wrnn_decoder* dec, size_t floats) {
  if (floats <= dec->ws_floats) return 0;
  if (dec->ws) {
    WRNN_HIP(hipDeviceSynchronize());
    WRNN_HIP(hipFree(dec->ws));
    dec->ws = nullptr;
    dec->ws_floats = 0;
  }
  WRNN_HIP(hipMalloc((void**)&dec->ws, floats * sizeof(float)));
  dec->ws_floats = floats;
  return 0;
}

int ensure_mel_layer(wrnn_decoder* dec, int r) {
  Layer& L = dec->mel[r];
  if (L.wt) return 0;
  const int N = dec->n_mels * r, l = dec->l;
  const float* W = dec->mel_w_host.data();
  RowFn w = [&](int t, int i, int k) -> float {
    const int j = t * 16 + i;
    if (j >= N) return 0.0f;
    const int m = j / r, f = j % r;
    return W[(size_t)(m * kMaxR + f) * l + k];
  };
  return build_layer(L, (N + 15) / 16, l, N, w, nullptr);
}

struct Guard {  // sets the handle's device for the duration of a call
  int prev = -1;
  bool ok = true;
  explicit Guard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~Guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct StepPtrs {
  const float *enc_q, *enc_p;
  const float* p;  // prenet output [B,128]
  const float *attn_h_in, *h1_in, *h2_in, *c1_in, *c2_in, *ctx_in;
  float *attn_h_out, *h1_out, *h2_out, *c1_out, *c2_out, *ctx_out;
  float *cum, *att;
  const float *zone_u1, *zone_u2;
  float* scores_out;
  long score_stride;
  float *pq, *u, *y, *x1, *x2;  // scratch
};

// the attention-RNN -> LSA -> rnn_input -> LSTM x2 part of a step (everything but prenet and mel_proj)
int launch_core(wrnn_decoder* dec, const StepPtrs& q, int B, int T, int first, hipStream_t st) {
  const int d = dec->d, l = dec->l;
  EpiGru eg{dec->gru.bias, q.attn_h_in, q.attn_h_out, d};
  if (launch_linear(dec->gru, Src{q.ctx_in, d, d}, Src{q.p, kFc2, kFc2}, Src{q.attn_h_in, d, d}, B, eg, st)) return 1;
  EpiLinear epq{dec->pq.bias, q.pq, d, d, 0, nullptr, 0};
  if (launch_linear(dec->pq, Src{q.attn_h_out, d, d}, kNoSrc, kNoSrc, B, epq, st)) return 1;
  hipLaunchKernelGGL(lsa_energy_kernel, dim3((T + kTC - 1) / kTC, B), dim3(256), 0, st, q.enc_p, q.pq, q.cum, q.att,
                     first, dec->conv_t, dec->L_w, dec->L_b, dec->v_w, q.u, T);
  WRNN_LAUNCH_CHECK();
  const size_t lds = (size_t)(T + 4 + 256) * sizeof(float);
  hipLaunchKernelGGL(lsa_softmax_ctx_kernel, dim3(kD / 64, B), dim3(256), lds, st, q.u, q.enc_q, q.cum, q.att, first,
                     q.scores_out, q.score_stride, q.ctx_out, T);
  WRNN_LAUNCH_CHECK();
  EpiLinear ein{dec->rin.bias, q.y, l, l, 0, nullptr, 0};
  if (launch_linear(dec->rin, Src{q.ctx_out, d, d}, Src{q.attn_h_out, d, d}, kNoSrc, B, ein, st)) return 1;
  EpiLstm e1{dec->l1.bias, q.c1_in, q.c1_out, q.h1_in, q.h1_out, q.y, q.x1, q.zone_u1, l};
  if (launch_linear(dec->l1, Src{q.y, l, l}, Src{q.h1_in, l, l}, kNoSrc, B, e1, st)) return 1;
  EpiLstm e2{dec->l2.bias, q.c2_in, q.c2_out, q.h2_in, q.h2_out, q.x1, q.x2, q.zone_u2, l};
  if (launch_linear(dec->l2, Src{q.x1, l, l}, Src{q.h2_in, l, l}, kNoSrc, B, e2, st)) return 1;
  return 0;
}

int launch_prenet(wrnn_decoder* dec, const float* x, int rows, const float* u1, const float* u2, float* p1,
                  float* p, hipStream_t st) {
  EpiLinear e1{dec->fc1.bias, p1, kFc1, kFc1, 1, u1, kFc1};
  if (launch_linear(dec->fc1, Src{x, dec->P, dec->P}, kNoSrc, kNoSrc, rows, e1, st)) return 1;
  EpiLinear e2{dec->fc2.bias, p, kFc2, kFc2, 1, u2, kFc2};
  return launch_linear(dec->fc2, Src{p1, kFc1, kFc1}, kNoSrc, kNoSrc, rows, e2, st);
}

int check_common(const wrnn_decoder* dec, int B, int T, int r) {
  WRNN_REQUIRE(dec != nullptr, "null decoder handle");
  WRNN_REQUIRE(B >= 1 && B <= 65535, "batch size B=%d out of range [1,65535]", B);
  WRNN_REQUIRE(T >= 1 && T <= 8192, "encoder length T=%d out of range [1,8192]", T);
  WRNN_REQUIRE(r >= 1 && r <= kMaxR, "reduction factor r=%d out of range [1,%d]", r, kMaxR);
  return 0;
}

}  // namespace

// ============================================================================ C ABI
extern "C" const char* wrnn_last_error(void) { return wrnn::last_error().c_str(); }
extern "C" int wrnn_version(void) { return 100; }

extern "C" int wrnn_decoder_create(const wrnn_decoder_dims* dims, const wrnn_decoder_weights* hw, int device,
                                   wrnn_decoder** out) {
  WRNN_REQUIRE(dims && hw && out, "wrnn_decoder_create: null argument");
  *out = nullptr;
  const int d = dims->decoder_dims, l = dims->lstm_dims, nm = dims->n_mels, A = dims->aligner_out_dims;
  WRNN_REQUIRE(d == kD, "decoder_dims must be %d (reference GRUCell input = decoder_dims + decoder_dims/2 = "
               "context + 128 prenet outputs), got %d", kD, d);
  WRNN_REQUIRE(l >= 4 && l % 4 == 0, "lstm_dims=%d must be a positive multiple of 4", l);
  WRNN_REQUIRE(nm >= 1 && A >= 0 && (nm + A) % 4 == 0, "n_mels + aligner_out_dims = %d must be a multiple of 4", nm + A);
  const float* const* wp = reinterpret_cast<const float* const*>(hw);
  for (size_t i = 0; i < sizeof(*hw) / sizeof(float*); ++i)
    WRNN_REQUIRE(wp[i] != nullptr, "wrnn_decoder_create: weight pointer #%zu is null", i);
  int ndev = 0;
  WRNN_HIP(hipGetDeviceCount(&ndev));
  WRNN_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  WRNN_HIP(hipGetDeviceProperties(&prop, device));
  WRNN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libwrnn_mi355 is built for gfx950 only, device %d is %s",
               device, prop.gcnArchName);
  Guard g(device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", device);

  auto* dec = new wrnn_decoder();
  dec->device = device;
  dec->n_mels = nm, dec->d = d, dec->l = l, dec->A = A, dec->P = nm + A;
  const int P = dec->P;
  int rc = 0;
  auto lin_rows = [](const float* W, int K) {
    return RowFn([=](int t, int i, int k) { return W[(size_t)(t * 16 + i) * K + k]; });
  };
  auto lin_bias = [](const float* b) { return std::function<float(int, int)>([=](int t, int i) { return b[t * 16 + i]; }); };
  {  // prenet
    auto b1 = lin_bias(hw->prenet_fc1_b), b2 = lin_bias(hw->prenet_fc2_b);
    rc |= build_layer(dec->fc1, kFc1 / 16, P, kFc1, lin_rows(hw->prenet_fc1_w, P), &b1);
    rc |= build_layer(dec->fc2, kFc2 / 16, kFc1, kFc2, lin_rows(hw->prenet_fc2_w, kFc1), &b2);
  }
  {  // attention GRU: K = [context d | prenet 128 | h d], tile rows = 4 units x (r, z, gi_n, gh_n)
    const int Kx = d + kFc2, K = Kx + d;
    const float *Wih = hw->gru_w_ih, *Whh = hw->gru_w_hh, *bih = hw->gru_b_ih, *bhh = hw->gru_b_hh;
    RowFn w = [=](int t, int i, int k) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      const bool xpart = k < Kx;
      switch (gsel) {
        case 0: return xpart ? Wih[(size_t)U * Kx + k] : Whh[(size_t)U * d + (k - Kx)];
        case 1: return xpart ? Wih[(size_t)(d + U) * Kx + k] : Whh[(size_t)(d + U) * d + (k - Kx)];
        case 2: return xpart ? Wih[(size_t)(2 * d + U) * Kx + k] : 0.0f;
        default: return xpart ? 0.0f : Whh[(size_t)(2 * d + U) * d + (k - Kx)];
      }
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      switch (gsel) {
        case 0: return bih[U] + bhh[U];
        case 1: return bih[d + U] + bhh[d + U];
        case 2: return bih[2 * d + U];
        default: return bhh[2 * d + U];
      }
    };
    rc |= build_layer(dec->gru, d / 4, K, 4 * d, w, &b);
  }
  {  // LSA query projection and rnn_input
    auto bq = lin_bias(hw->lsa_W_b), bi = lin_bias(hw->rnn_input_b);
    rc |= build_layer(dec->pq, d / 16, d, d, lin_rows(hw->lsa_W_w, d), &bq);
    RowFn wi = [=](int t, int i, int k) -> float {
      const int n = t * 16 + i;
      return n < l ? hw->rnn_input_w[(size_t)n * 2 * d + k] : 0.0f;
    };
    std::function<float(int, int)> bi2 = [=](int t, int i) -> float { return t * 16 + i < l ? hw->rnn_input_b[t * 16 + i] : 0.0f; };
    (void)bi;
    rc |= build_layer(dec->rin, (l + 15) / 16, 2 * d, l, wi, &bi2);
  }
  auto build_lstm = [&](Layer& L, const float* Wih, const float* Whh, const float* bih, const float* bhh) {
    RowFn w = [=](int t, int i, int k) -> float {
      const int row = (i & 3) * l + t * 4 + (i >> 2);
      return k < l ? Wih[(size_t)row * l + k] : Whh[(size_t)row * l + (k - l)];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int row = (i & 3) * l + t * 4 + (i >> 2);
      return bih[row] + bhh[row];
    };
    return build_layer(L, l / 4, 2 * l, 4 * l, w, &b);
  };
  rc |= build_lstm(dec->l1, hw->lstm1_w_ih, hw->lstm1_w_hh, hw->lstm1_b_ih, hw->lstm1_b_hh);
  rc |= build_lstm(dec->l2, hw->lstm2_w_ih, hw->lstm2_w_hh, hw->lstm2_b_ih, hw->lstm2_b_hh);
  dec->mel_w_host.assign(hw->mel_proj_w, hw->mel_proj_w + (size_t)nm * kMaxR * l);
  {  // LSA small weights; conv transposed to [c][k][filter] so lanes read consecutive filters
    std::vector<float> ct((size_t)2 * kTaps * kFilters);
    for (int fl = 0; fl < kFilters; ++fl)
      for (int c = 0; c < 2; ++c)
        for (int k = 0; k < kTaps; ++k) ct[(c * kTaps + k) * kFilters + fl] = hw->lsa_conv_w[(fl * 2 + c) * kTaps + k];
    rc |= upload(ct.data(), ct.size() * sizeof(float), (void**)&dec->conv_t);
    rc |= upload(hw->lsa_L_w, (size_t)d * kFilters * sizeof(float), (void**)&dec->L_w);
    rc |= upload(hw->lsa_L_b, (size_t)d * sizeof(float), (void**)&dec->L_b);
    rc |= upload(hw->lsa_v_w, (size_t)d * sizeof(float), (void**)&dec->v_w);
  }
  if (rc) {
    std::string keep = wrnn::last_error();
    wrnn_decoder_destroy(dec);
    wrnn::last_error() = keep;
    return 1;
  }
  *out = dec;
  return 0;
}

extern "C" int wrnn_decoder_destroy(wrnn_decoder* dec) {
  if (!dec) return 0;
  Guard g(dec->device);
  (void)hipDeviceSynchronize();
  free_layer(dec->fc1), free_layer(dec->fc2), free_layer(dec->gru), free_layer(dec->pq);
  free_layer(dec->rin), free_layer(dec->l1), free_layer(dec->l2);
  for (auto& m : dec->mel) free_layer(m);
  for (float* p : {dec->conv_t, dec->L_w, dec->L_b, dec->v_w, dec->ws})
    if (p) (void)hipFree(p);
  if (dec->flags) (void)hipFree(dec->flags);
  delete dec;
  return 0;
}

extern "C" int wrnn_decoder_step(wrnn_decoder* dec, const float* enc_q, const float* enc_p, const float* prenet_in,
                                 const wrnn_decoder_state* in, const wrnn_decoder_state* out, int B, int T, int t,
                                 int r, const float* drop_u1, const float* drop_u2, const float* zone_u1,
                                 const float* zone_u2, float* mels_out, float* scores_out, void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(enc_q && enc_p && prenet_in && in && out && mels_out && scores_out, "wrnn_decoder_step: null pointer");
  WRNN_REQUIRE(t >= 0, "wrnn_decoder_step: t=%d must be >= 0", t);
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  WRNN_REQUIRE((zone_u1 == nullptr) == (zone_u2 == nullptr), "give both zoneout uniform tensors or neither");
  WRNN_REQUIRE(in->attn_h != out->attn_h && in->h1 != out->h1 && in->h2 != out->h2,
               "wrnn_decoder_step: hidden-state outputs must not alias the inputs");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l;
  const size_t need = (size_t)B * (kFc1 + kFc2 + d + T + 3 * l);
  if (ensure_ws(dec, need)) return 1;
  if (ensure_mel_layer(dec, r)) return 1;
  float* w = dec->ws;
  float* p1 = w; w += (size_t)B * kFc1;
  float* p = w; w += (size_t)B * kFc2;
  float* pq = w; w += (size_t)B * d;
  float* u = w; w += (size_t)B * T;
  float* y = w; w += (size_t)B * l;
  float* x1 = w; w += (size_t)B * l;
  float* x2 = w;
  if (launch_prenet(dec, prenet_in, B, drop_u1, drop_u2, p1, p, st)) return 1;
  StepPtrs q{enc_q, enc_p, p, in->attn_h, in->h1, in->h2, in->c1, in->c2, in->context,
             out->attn_h, out->h1, out->h2, out->c1, out->c2, out->context, out->lsa_cum, out->lsa_att,
             zone_u1, zone_u2, scores_out, (long)T, pq, u, y, x1, x2};
  if (t != 0 && (in->lsa_cum != out->lsa_cum || in->lsa_att != out->lsa_att)) {
    // state handed over in separate buffers: carry it into the output buffers first
    WRNN_HIP(hipMemcpyAsync(out->lsa_cum, in->lsa_cum, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st));
    WRNN_HIP(hipMemcpyAsync(out->lsa_att, in->lsa_att, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  if (launch_core(dec, q, B, T, t == 0, st)) return 1;
  EpiMel em{mels_out, (long)dec->n_mels * r, (long)r, r, dec->n_mels * r, B};
  return launch_linear(dec->mel[r], Src{x2, l, l}, kNoSrc, kNoSrc, B, em, st);
}

extern "C" int wrnn_decoder_run_teacher_forced(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                                               const float* mel_in, int B, int T, int S, int r,
                                               const float* drop_u1, const float* drop_u2, float* mel_out,
                                               float* att_out, void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(S >= 1, "wrnn_decoder_run_teacher_forced: S=%d must be >= 1", S);
  WRNN_REQUIRE(enc_q && enc_p && mel_in && mel_out && att_out, "wrnn_decoder_run_teacher_forced: null pointer");
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l, P = dec->P;
  const int n = (S + r - 1) / r;
  const size_t rows = (size_t)n * B;
  WRNN_REQUIRE(rows <= (size_t)1 << 30, "wrnn_decoder_run_teacher_forced: n*B too large");
  const size_t need = rows * (P + kFc1 + kFc2 + l) + (size_t)B * (2 * d + 2 * l * 2 + 2 * l + d + d + T * 3 + 2 * l);
  if (ensure_ws(dec, need)) return 1;
  if (ensure_mel_layer(dec, r)) return 1;
  float* w = dec->ws;
  auto take = [&](size_t nfl) { float* p = w; w += nfl; return p; };
  float* xp = take(rows * P);
  float* p1 = take(rows * kFc1);
  float* pp = take(rows * kFc2);
  float* x2 = take(rows * l);
  float* ah[2] = {take((size_t)B * d), take((size_t)B * d)};
  float* h1[2] = {take((size_t)B * l), take((size_t)B * l)};
  float* h2[2] = {take((size_t)B * l), take((size_t)B * l)};
  float* c1 = take((size_t)B * l);
  float* c2 = take((size_t)B * l);
  float* ctx = take((size_t)B * d);
  float* pq = take((size_t)B * d);
  float* cum = take((size_t)B * T);
  float* att = take((size_t)B * T);
  float* u = take((size_t)B * T);
  float* y = take((size_t)B * l);
  float* x1 = take((size_t)B * l);
  // zero initial state (tacotron.py:302-316): ah[0], h1[0], h2[0], c1, c2, ctx
  WRNN_HIP(hipMemsetAsync(ah[0], 0, (size_t)B * d * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(h1[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(h2[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(c1, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(c2, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(ctx, 0, (size_t)B * d * sizeof(float), st));
  // teacher forcing: every step's prenet input is known up front -> hoist the prenet (one batched GEMM pair)
  {
    const size_t total = rows * P;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_prenet_in_kernel, dim3(blocks), dim3(256), 0, st, mel_in, xp, B, P, S, r, n);
    WRNN_LAUNCH_CHECK();
  }
  if (launch_prenet(dec, xp, (int)rows, drop_u1, drop_u2, p1, pp, st)) return 1;
  for (int k = 0; k < n; ++k) {
    const int a = k & 1, b = a ^ 1;
    StepPtrs q{enc_q, enc_p, pp + (size_t)k * B * kFc2, ah[a], h1[a], h2[a], c1, c2, ctx,
               ah[b], h1[b], h2[b], c1, c2, ctx, cum, att, nullptr, nullptr,
               att_out + (size_t)k * T, (long)n * T, pq, u, y, x1, x2 + (size_t)k * B * l};
    if (launch_core(dec, q, B, T, k == 0, st)) return 1;
  }
  // mel projection of every step at once, scattered into mel_out[B, n_mels, n*r]
  EpiMel em{mel_out, (long)dec->n_mels * n * r, (long)n * r, r, dec->n_mels * r, B};
  return launch_linear(dec->mel[r], Src{x2, l, l}, kNoSrc, kNoSrc, (int)rows, em, st);
}

extern "C" int wrnn_decoder_run_free(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                                     const float* aligner_feats, int B, int T, int max_steps, int r,
                                     float stop_threshold, int use_stop, const float* drop_u1,
                                     const float* drop_u2, float* mel_out, float* att_out, int* n_steps_out,
                                     void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(max_steps >= 1, "wrnn_decoder_run_free: max_steps=%d must be >= 1", max_steps);
  WRNN_REQUIRE(enc_q && enc_p && mel_out && att_out && n_steps_out, "wrnn_decoder_run_free: null pointer");
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l, P = dec->P, n = max_steps;
  const size_t need = (size_t)B * (2 * P + kFc1 + kFc2 + 2 * d + 6 * l + 2 * l + d + d + 3 * T + 3 * l);
  if (ensure_ws(dec, need)) return 1;
  if (ensure_mel_layer(dec, r)) return 1;
  if (dec->flags_n < n) {
    if (dec->flags) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(dec->flags));
      dec->flags = nullptr;
    }
    WRNN_HIP(hipMalloc((void**)&dec->flags, (size_t)n * sizeof(int)));
    dec->flags_n = n;
  }
  float* w = dec->ws;
  auto take = [&](size_t nfl) { float* p = w; w += nfl; return p; };
  float* pin = take((size_t)B * P);
  float* p1 = take((size_t)B * kFc1);
  float* pp = take((size_t)B * kFc2);
  float* ah[2] = {take((size_t)B * d), take((size_t)B * d)};
  float* h1[2] = {take((size_t)B * l), take((size_t)B * l)};
  float* h2[2] = {take((size_t)B * l), take((size_t)B * l)};
  float* c1 = take((size_t)B * l);
  float* c2 = take((size_t)B * l);
  float* ctx = take((size_t)B * d);
  float* pq = take((size_t)B * d);
  float* cum = take((size_t)B * T);
  float* att = take((size_t)B * T);
  float* u = take((size_t)B * T);
  float* y = take((size_t)B * l);
  float* x1 = take((size_t)B * l);
  float* x2 = take((size_t)B * l);
  WRNN_HIP(hipMemsetAsync(pin, 0, (size_t)B * P * sizeof(float), st));  // go frame
  WRNN_HIP(hipMemsetAsync(ah[0], 0, (size_t)B * d * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(h1[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(h2[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(c1, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(c2, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(ctx, 0, (size_t)B * d * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(dec->flags, 0, (size_t)n * sizeof(int), st));
  const long sb = (long)dec->n_mels * n * r, sm = (long)n * r;
  constexpr int kPoll = 32;  // host looks at the stop votes every kPoll groups
  std::vector<int> votes(n, 1);
  int done = n;
  for (int k = 0; k < n; ++k) {
    const int a = k & 1, b = a ^ 1;
    if (launch_prenet(dec, pin, B, drop_u1 ? drop_u1 + (size_t)k * B * kFc1 : nullptr,
                      drop_u2 ? drop_u2 + (size_t)k * B * kFc2 : nullptr, p1, pp, st)) return 1;
    StepPtrs q{enc_q, enc_p, pp, ah[a], h1[a], h2[a], c1, c2, ctx, ah[b], h1[b], h2[b], c1, c2, ctx,
               cum, att, nullptr, nullptr, att_out + (size_t)k * T, (long)n * T, pq, u, y, x1, x2};
    if (launch_core(dec, q, B, T, k == 0, st)) return 1;
    // one group's frames straight into mel_out[:, :, k*r : (k+1)*r]
    EpiMel em{mel_out + (size_t)k * r, sb, sm, r, dec->n_mels * r, B};
    if (launch_linear(dec->mel[r], Src{x2, l, l}, kNoSrc, kNoSrc, B, em, st)) return 1;
    hipLaunchKernelGGL(feedback_kernel, dim3(B), dim3(128), 0, st, mel_out, sb, sm, k, r, dec->n_mels,
                       aligner_feats, dec->A, n, pin, stop_threshold, dec->flags);
    WRNN_LAUNCH_CHECK();
    if (use_stop && ((k + 1) % kPoll == 0 || k + 1 == n)) {
      const int lo = (k / kPoll) * kPoll;
      WRNN_HIP(hipMemcpyAsync(votes.data() + lo, dec->flags + lo, (size_t)(k + 1 - lo) * sizeof(int),
                              hipMemcpyDeviceToHost, st));
      WRNN_HIP(hipStreamSynchronize(st));
      bool stop = false;
      for (int kk = lo; kk <= k; ++kk)
        if (kk > 10 && votes[kk] == 0) {  // (mel < threshold).all() and k > 10
          done = kk + 1;
          stop = true;
          break;
        }
      if (stop) break;
    }
  }
  *n_steps_out = done;
  return 0;
}
