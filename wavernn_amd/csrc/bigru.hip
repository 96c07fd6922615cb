// Bidirectional single-layer GRU over M frames for gfx950 (MI355X), fp32 storage and accumulate.
//
// Replaces the `nn.GRU(channels, channels, batch_first=True, bidirectional=True)` call that ends the
// reference's CBHG block (models/common_layers.py:89,123): the Tacotron encoder (models/tacotron.py:76-83,
// T tokens) and postnet (models/tacotron.py:269,351, S mel frames), SURVEY.md section 8 row f2, and the
// forward models' prenet/postnet CBHGs.  On PyTorch-ROCm's own nn.GRU (MIOpen) the postnet recurrence alone
// takes 3x as long as the whole decoder loop it follows (tools/prof_tacotron.py).
//
// Structure: the same as bilstm.hip.  Both directions advance in lock step, one launch per frame carrying
// two LinJob<1,1,EpiBiGru> (forward and reverse cell) as block ranges of one tick_kernel; [x_t | h] is one
// K = in + H contraction per cell; the pre-pass of birnn.h aligns the reverse direction to each row's own
// last frame so that packed-sequence semantics need no gather inside the loop.
// torch GRU cell (gate order r, z, n):
//   r = s(W_ir x + b_ir + W_hr h + b_hr) ; z = s(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) ; h' = (1 - z) * n + z * h
#include "birnn.h"

namespace {

struct EpiBiGru {  // tile rows: 4 units x (r, z, gi_n, gh_n); the n rows hold zero in the other operand's columns
  const float* bias;  // tile order: b_ir + b_hr, b_iz + b_hz, b_in, b_hn
  Act h_prev, h_out;
  const int32_t* lens;  // [B] or null (= every row has M frames)
  float* out;           // [B, M, H2]
  int M, H2, col0, s, rev;
  struct Pre { f32x4 b; float hp; int len; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.hp = h_prev.rowbase(row)[h_prev.eoff(tile * 4 + u)];
    q.len = lens ? min(max(lens[row.b], 0), M) : M;
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const int U = tile * 4 + u;
    const float r = sigmoidf_(v[0] + q.b[0]);
    const float z = sigmoidf_(v[1] + q.b[1]);
    const float n = tanhf_(v[2] + q.b[2] + r * (v[3] + q.b[3]));
    float h = (1.0f - z) * n + z * q.hp;
    const bool active = s < q.len;  // past the row's last frame the state is frozen (packed semantics)
    if (!active) h = q.hp;
    h_out.rowbase(row)[h_out.eoff(U)] = h;
    if (active) {
      const int t = rev ? q.len - 1 - s : s;
      out[((size_t)row.b * M + t) * H2 + col0 + U] = h;
    }
  }
};

using BiGruJob = LinJob<1, 1, EpiBiGru>;
// two tiles per workgroup for wide cells: the two directions of a frame are two equally long jobs in one launch, and
// they do better with fewer workgroups sharing each compute unit (H = 256: 4.1 -> 3.6 ms per 64 x 768-frame batch)
using BiGruJobWide = LinJob<2, 1, EpiBiGru>;

// ---------------------------------------------------------------- weights-in-registers recurrence (H <= 128)
// For the CBHG sizes (H = 128: Tacotron encoder and postnet) a whole direction's W_hh (3H x H fp32 = 192 KB) fits the
// vector register file of ONE compute unit (512 KB).  Batch rows do not interact, so a workgroup per (row, direction)
// runs the row's whole recurrence by itself, the weights in registers for the entire sequence and the state in LDS:
// no launch, no grid-wide hand-over and no weight traffic per frame.  The input half W_ih x + b of every frame and
// both directions is one GEMM before the loop (gx).
// Thread mapping: a quad of lanes per hidden unit u; lane q of the quad holds columns [q H/4, (q+1) H/4) of the
// unit's three gate rows (r, z, n): 3H/4 registers.  A step is H/16 LDS reads of the lane's quarter of the state,
// 3H/8 packed fp32 FMAs, a quad all-reduce of the three partial sums on the DPP cross-lane path, the gate math
// (redundantly in the four lanes, so nothing has to be handed over), one LDS store of the new state into the other
// of two buffers and ONE LDS-only barrier.
using v2f = __attribute__((ext_vector_type(2))) float;
constexpr int kResChunk = 16;  // frames of gx in flight per thread

template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {  // quad_perm cross-lane move
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int H>
__global__ __launch_bounds__(4 * H) void bigru_resident_kernel(const float* __restrict__ gx /*[B*M][6H]*/,
                                                               const float* __restrict__ whh /*[2][3H][H]*/,
                                                               const float* __restrict__ bhn /*[2][H]*/,
                                                               const int32_t* __restrict__ lens, int B, int M,
                                                               float* __restrict__ out /*[B][M][2H]*/,
                                                               float* __restrict__ h_n /*[2][B][H] or null*/) {
  static_assert(H % 32 == 0 && H <= 128, "a quad of lanes per unit, 3H/4 weights in registers per lane");
  constexpr int Q = H / 4;       // columns per lane
  constexpr int QS = Q + 4;      // LDS stride of a quarter of the state: the four quarters start in different banks
  __shared__ __attribute__((aligned(16))) float hs[2][4 * QS];
  const int dir = blockIdx.x & 1, b = blockIdx.x >> 1;
  const int u = threadIdx.x >> 2, q = threadIdx.x & 3;
  const int len = lens ? min(max(lens[b], 0), M) : M;
  v2f w[3][Q / 2];
#pragma unroll
  for (int gi = 0; gi < 3; ++gi) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(whh + ((size_t)dir * 3 * H + gi * H + u) * H + q * Q);
#pragma unroll
    for (int k = 0; k < Q / 4; ++k) {
      const f32x4 v = wr[k];
      w[gi][2 * k] = v2f{v[0], v[1]}, w[gi][2 * k + 1] = v2f{v[2], v[3]};
    }
  }
  const float bn = bhn[dir * H + u];
  const int hslot = (u / Q) * QS + u % Q;  // where unit u lives in a state buffer
  if (q == 0) hs[0][hslot] = 0.0f;
  lds_barrier();
  // lane q fetches gate q's input half (lane 3 duplicates the n gate's); the quad shares the three by DPP
  const float* gxp = gx + (size_t)b * M * 6 * H + dir * 3 * H + min(q, 2) * H + u;
  // frame of step s: s (forward) or len-1-s (reverse: starts at the row's own last frame, packed semantics)
  auto frame = [&](int s) { return dir ? len - 1 - s : s; };
  auto fetch = [&](float (&v)[kResChunk], int s0) {  // unconditional, clamped: steps past the end re-read the last frame
#pragma unroll
    for (int i = 0; i < kResChunk; ++i) v[i] = gxp[(size_t)frame(min(s0 + i, len - 1)) * 6 * H];
  };
  float hreg = 0.0f;  // the unit's state (the same in the four lanes)
  auto step = [&](int s, float gmine, int cur) {
    const float* hq = &hs[cur][q * QS];
    v2f a[3][2];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) a[gi][0] = a[gi][1] = v2f{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; k += 4) {
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(hq + k);
      const v2f ha = {h4[0], h4[1]}, hb = {h4[2], h4[3]};
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        a[gi][0] = __builtin_elementwise_fma(w[gi][k / 2], ha, a[gi][0]);
        a[gi][1] = __builtin_elementwise_fma(w[gi][k / 2 + 1], hb, a[gi][1]);
      }
    }
    float p[3];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      const v2f t = a[gi][0] + a[gi][1];
      float v = t[0] + t[1];
      v += quad_dpp<0xB1>(v);  // lanes 1 0 3 2
      v += quad_dpp<0x4E>(v);  // lanes 2 3 0 1
      p[gi] = v;
    }
    // gx carries b_ih + b_hh for r and z, b_ih for n
    const float r = sigmoidf_(p[0] + quad_dpp<0x00>(gmine));
    const float z = sigmoidf_(p[1] + quad_dpp<0x55>(gmine));
    const float n = tanhf_(quad_dpp<0xAA>(gmine) + r * (p[2] + bn));
    hreg = (1.0f - z) * n + z * hreg;
    if (q == 0) {
      hs[cur ^ 1][hslot] = hreg;
      out[((size_t)b * M + frame(s)) * 2 * H + dir * H + u] = hreg;
    }
    lds_barrier();
  };
  if (len > 0) {
    float gc[kResChunk], gn[kResChunk];
    fetch(gc, 0);
    int s0 = 0;  // always a multiple of kResChunk (even): step s0 + i reads state buffer i & 1
    // whole chunks: straight-line code, the next chunk's gx requested before the chunk's steps and stores
    for (; s0 + kResChunk <= len; s0 += kResChunk) {
      fetch(gn, s0 + kResChunk);
#pragma unroll
      for (int i = 0; i < kResChunk; ++i) step(s0 + i, gc[i], i & 1);
#pragma unroll
      for (int i = 0; i < kResChunk; ++i) gc[i] = gn[i];
    }
#pragma unroll
    for (int i = 0; i < kResChunk; ++i)
      if (s0 + i < len) step(s0 + i, gc[i], i & 1);  // workgroup-uniform: every thread reaches the same barriers
  }
  if (h_n && q == 0) h_n[((size_t)dir * B + b) * H + u] = hreg;
}


// ---- 128 < H <= 256 (the forward models' 256-wide CBHGs): a PAIR of workgroups per (row, direction).  One direction's W_hh
// (3 x 256 x 256 fp32 = 786 KB) is more than one compute unit's registers and LDS, half of it (the gate rows of 128 units:
// 393 KB = 192 registers per lane) is not.  Each workgroup of a pair runs the kernel above for its half of the units and
// keeps the FULL state in LDS: after every frame it publishes its half (H/2 values as 16-byte {value, tag, value, tag}
// pairs, one whole 128-byte line per wave, write-through) and picks up the partner's (one wave polls, sc1 loads, bounded).
// The partners' block indices differ by 8, so that they sit on the same XCD (workgroups are dealt to the 8 XCDs in turn)
// and the hand-over is served by that XCD's L2; nothing depends on it but the time.  The grid needs all its workgroups resident
// at once (a pair spins on each other): at most 64 rows per launch on 256 compute units, on the device's stream for such grids.
struct BiGruPairParams {
  const float* gx;        // [B*M][6H]
  const float* whh;       // [2][3H][H]
  const float* bhn;       // [2][H]
  const int32_t* lens;
  float* out;             // [B][M][2H]
  float* h_n;             // [2][B][H] or null
  unsigned long long* xch;   // [2 dir][B][2 halves][2 parities][H/4 pairs][2]
  unsigned* err;
  int B, M, b0, nb;       // rows b0 .. b0 + nb - 1 in this launch
  int wt_only;            // diagnostic (WRNN_BIGRU_WRITE_THROUGH): write-through stores also between partners on one XCD
};

using pu32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <int H>
__global__ __launch_bounds__(2 * H) void bigru_pair_kernel(BiGruPairParams p) {
  static_assert(H % 64 == 0 && H > 128 && H <= 256, "half the units per workgroup, a quad of lanes per unit");
  constexpr int HH = H / 2;      // units per workgroup
  constexpr int Q = H / 4;       // columns per lane
  constexpr int QS = Q + 4;      // LDS stride of a quarter of the state
  constexpr int NP = HH / 2;     // pairs a workgroup publishes per frame (<= 64: one wave)
  constexpr unsigned kSpin = 1u << 21;
  constexpr int kChunk = H > 192 ? 2 : 8;   // frames of gx in flight per thread (192 registers of weights at H = 256: no room for more)
  __shared__ __attribute__((aligned(16))) float hs[2][4 * QS];
  // block -> (pair of workgroups, half); partners 8 blocks apart
  const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3, half = t & 1, pi = (t >> 1) * 8 + xcd;
  if (pi >= 2 * p.nb) return;   // (both partners leave)
  const int dir = pi & 1, b = p.b0 + (pi >> 1);
  const int M = p.M, B = p.B;
  const int ul = threadIdx.x >> 2, q = threadIdx.x & 3, u = half * HH + ul;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int len = p.lens ? min(max(p.lens[b], 0), M) : M;
  v2f w[3][Q / 2];
#pragma unroll
  for (int gi = 0; gi < 3; ++gi) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(p.whh + ((size_t)dir * 3 * H + gi * H + u) * H + q * Q);
#pragma unroll
    for (int k = 0; k < Q / 4; ++k) {
      const f32x4 v = wr[k];
      w[gi][2 * k] = v2f{v[0], v[1]}, w[gi][2 * k + 1] = v2f{v[2], v[3]};
    }
  }
  const float bn = p.bhn[dir * H + u];
  auto slot_of = [](int unit) { return (unit / Q) * QS + unit % Q; };
  const int hslot = slot_of(u);
  for (int i = threadIdx.x; i < 4 * QS; i += 2 * H) hs[0][i] = 0.0f;
  lds_barrier();
  const float* gxp = p.gx + (size_t)b * M * 6 * H + dir * 3 * H + min(q, 2) * H + u;
  auto frame = [&](int s) { return dir ? len - 1 - s : s; };
  auto fetch = [&](float (&v)[kChunk], int s0) {
#pragma unroll
    for (int i = 0; i < kChunk; ++i) v[i] = gxp[(size_t)frame(min(s0 + i, len - 1)) * 6 * H];
  };
  // hand-over slots: [dir][b][half][parity][NP pairs] x 16 bytes
  unsigned long long* const xmine = p.xch + ((((size_t)dir * B + b) * 2 + half) * 2) * NP * 2;
  const unsigned long long* const xother = p.xch + ((((size_t)dir * B + b) * 2 + (half ^ 1)) * 2) * NP * 2;
  bool gave_up = false;
  float hreg = 0.0f;
  // Which XCD this workgroup runs on (hardware register), told to the partner in the first frame's tag: partners on the same
  // XCD (the expected case, see above) publish with plain stores from the second frame on -- the value then lives in the L2
  // both read through and needs no acknowledgement from the memory side (a hand-over among 32 workgroups: 0.93 us against
  // 1.56 write-through, tools/microbench/gather_bench.hip x); partners on different XCDs keep the write-through stores.
  __shared__ unsigned sh_same_xcd;
  unsigned my_xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
  my_xcc = (my_xcc & 15u) + 1u;
  bool same_xcd = false;
  auto step = [&](int s, float gmine, int cur) {
    const float* hq = &hs[cur][q * QS];
    v2f a[3][2];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) a[gi][0] = a[gi][1] = v2f{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; k += 4) {
      const f32x4 h4 = *reinterpret_cast<const f32x4*>(hq + k);
      const v2f ha = {h4[0], h4[1]}, hb = {h4[2], h4[3]};
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        a[gi][0] = __builtin_elementwise_fma(w[gi][k / 2], ha, a[gi][0]);
        a[gi][1] = __builtin_elementwise_fma(w[gi][k / 2 + 1], hb, a[gi][1]);
      }
    }
    float pr[3];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      const v2f tt = a[gi][0] + a[gi][1];
      float v = tt[0] + tt[1];
      v += quad_dpp<0xB1>(v);
      v += quad_dpp<0x4E>(v);
      pr[gi] = v;
    }
    const float r = sigmoidf_(pr[0] + quad_dpp<0x00>(gmine));
    const float z = sigmoidf_(pr[1] + quad_dpp<0x55>(gmine));
    const float n = tanhf_(quad_dpp<0xAA>(gmine) + r * (pr[2] + bn));
    hreg = (1.0f - z) * n + z * hreg;
    const unsigned tag = (unsigned)s + 1u;   // (low 24 bits of the tag word; the first frame's carries the XCD above them)
    // publish: every wave its own 16 units as eight 16-byte pairs = one whole 128-byte line per store instruction, straight
    // from the registers (through LDS and one wave: a barrier and an LDS round trip more per frame, 0.2 us)
    const float hnext = __shfl_down(hreg, 4);   // the next unit's (same wave: 16 units per wave)
    if ((threadIdx.x & 7) == 0) {
      const unsigned tw = s == 0 ? tag | (my_xcc << 24) : tag;
      const pu32x4 qv = {__float_as_uint(hreg), tw, __float_as_uint(hnext), tw};
      unsigned long long* const dst = xmine + ((size_t)(s & 1) * NP + (ul >> 1)) * 2;
      if (same_xcd) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(dst), "v"(qv) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst), "v"(qv) : "memory");
    }
    if (q == 0) {
      hs[cur ^ 1][hslot] = hreg;
      p.out[((size_t)b * M + frame(s)) * 2 * H + dir * H + u] = hreg;
    }
    if (wave == 1) {                      // pick up the partner's half
      const unsigned long long* a0 = xother + ((size_t)(s & 1) * NP + min(lane, NP - 1)) * 2;
      pu32x4 qv;
      for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(qv) : "v"(a0) : "memory");
        if (__all((qv.y & 0xFFFFFFu) == tag && (qv.w & 0xFFFFFFu) == tag) || gave_up) break;
        if (!wrnn::keep_polling(spins, kSpin, p.err, gave_up)) break;
      }
      if (s == 0 && lane == 0) sh_same_xcd = !gave_up && (qv.y >> 24) == my_xcc;
      if (lane < NP) {
        hs[cur ^ 1][slot_of((half ^ 1) * HH + 2 * lane)] = __uint_as_float(qv.x);
        hs[cur ^ 1][slot_of((half ^ 1) * HH + 2 * lane + 1)] = __uint_as_float(qv.z);
      }
    }
    lds_barrier();   // the full new state is in LDS
    if (s == 0) same_xcd = __builtin_amdgcn_readfirstlane(sh_same_xcd) != 0u && !p.wt_only;
  };
  if (len > 0) {
    float gc[kChunk], gn[kChunk];
    fetch(gc, 0);
    int s0 = 0;
    for (; s0 + kChunk <= len; s0 += kChunk) {
      fetch(gn, s0 + kChunk);
#pragma unroll
      for (int i = 0; i < kChunk; ++i) step(s0 + i, gc[i], i & 1);
#pragma unroll
      for (int i = 0; i < kChunk; ++i) gc[i] = gn[i];
    }
#pragma unroll
    for (int i = 0; i < kChunk; ++i)
      if (s0 + i < len) step(s0 + i, gc[i], i & 1);
  }
  if (p.h_n && q == 0) p.h_n[((size_t)dir * B + b) * H + u] = hreg;
}

// a hand-over timed out: NaN in every output of the call, no silent garbage
__global__ void bigru_poison_kernel(const unsigned* err, float* out, size_t n, float* h_n, size_t ns) {
  if (*err == 0) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = __builtin_nanf("");
  if (h_n)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ns; i += (size_t)gridDim.x * blockDim.x) h_n[i] = __builtin_nanf("");
}

}  // namespace

struct wrnn_bigru {
  int device = 0, in = 0, H = 0;
  Layer fwd, rev;
  // weights-in-registers path (H a multiple of 32, <= 128)
  bool resident = false;
  bool pair = false;      // 128 < H <= 256, H % 64 == 0: a pair of workgroups per (row, direction) (bigru_pair_kernel)
  int cus = 0;
  unsigned long long* xch = nullptr;   // pair path: hand-over slots (grown with B)
  size_t xch_rows = 0;
  unsigned* err = nullptr;
  Layer xproj;            // [6H rows = dir x (r|z|n)] x in, bias b_ih (+ b_hh for r, z)
  float* whh = nullptr;   // [2][3H][H], torch layout
  float* bhn = nullptr;   // [2][H]
  float* ws = nullptr;
  size_t ws_floats = 0;
};

extern "C" int wrnn_bigru_destroy(wrnn_bigru* h) {
  if (!h) return 0;
  Guard g(h->device);
  (void)hipDeviceSynchronize();
  free_layer(h->fwd), free_layer(h->rev), free_layer(h->xproj);
  if (h->whh) (void)hipFree(h->whh);
  if (h->bhn) (void)hipFree(h->bhn);
  if (h->ws) (void)hipFree(h->ws);
  if (h->xch) (void)hipFree(h->xch);
  if (h->err) (void)hipFree(h->err);
  delete h;
  return 0;
}

extern "C" int wrnn_bigru_create(int in_dims, int rnn_dims, const wrnn_bigru_weights* hw, int device, wrnn_bigru** out) {
  WRNN_REQUIRE(hw && out, "wrnn_bigru_create: null argument");
  *out = nullptr;
  WRNN_REQUIRE(in_dims >= 4 && in_dims % 4 == 0, "wrnn_bigru_create: in_dims=%d must be a positive multiple of 4", in_dims);
  WRNN_REQUIRE(rnn_dims >= 4 && rnn_dims % 4 == 0, "wrnn_bigru_create: rnn_dims=%d must be a positive multiple of 4", rnn_dims);
  const float* const* wp = reinterpret_cast<const float* const*>(hw);
  for (size_t i = 0; i < sizeof(*hw) / sizeof(float*); ++i)
    WRNN_REQUIRE(wp[i] != nullptr, "wrnn_bigru_create: weight pointer #%zu is null", i);
  int ndev = 0;
  WRNN_HIP(hipGetDeviceCount(&ndev));
  WRNN_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  WRNN_HIP(hipGetDeviceProperties(&prop, device));
  WRNN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libwrnn_mi355 is built for gfx950 only, device %d is %s",
               device, prop.gcnArchName);
  Guard g(device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", device);
  auto* h = new wrnn_bigru();
  h->device = device, h->in = in_dims, h->H = rnn_dims;
  const int in = in_dims, H = rnn_dims;
  // K = [x (in) | h (H)]; torch rows are gate-major r | z | n
  auto build = [&](Layer& L, const float* Wih, const float* Whh, const float* bih, const float* bhh) {
    RowFn w = [=](int t, int i, int k) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      const bool xpart = k < in;
      switch (gsel) {
        case 0: return xpart ? Wih[(size_t)U * in + k] : Whh[(size_t)U * H + (k - in)];
        case 1: return xpart ? Wih[(size_t)(H + U) * in + k] : Whh[(size_t)(H + U) * H + (k - in)];
        case 2: return xpart ? Wih[(size_t)(2 * H + U) * in + k] : 0.0f;
        default: return xpart ? 0.0f : Whh[(size_t)(2 * H + U) * H + (k - in)];
      }
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      switch (gsel) {
        case 0: return bih[U] + bhh[U];
        case 1: return bih[H + U] + bhh[H + U];
        case 2: return bih[2 * H + U];
        default: return bhh[2 * H + U];
      }
    };
    return build_layer(L, H / 4, in + H, 4 * H, w, &b);
  };
  int rc = build(h->fwd, hw->w_ih, hw->w_hh, hw->b_ih, hw->b_hh);
  rc |= build(h->rev, hw->w_ih_reverse, hw->w_hh_reverse, hw->b_ih_reverse, hw->b_hh_reverse);
  h->resident = H % 32 == 0 && H <= 128;
  h->cus = prop.multiProcessorCount;
  h->pair = (H == 192 || H == 256) && h->cus >= 8;
  if ((h->resident || h->pair) && !rc) {
    const float *Wf = hw->w_ih, *Wr = hw->w_ih_reverse;
    const float *bif = hw->b_ih, *bir = hw->b_ih_reverse, *bhf = hw->b_hh, *bhr = hw->b_hh_reverse;
    RowFn w = [=](int t, int i, int k) -> float {
      const int n = t * 16 + i, d = n / (3 * H), gr = n - d * 3 * H;
      return (d ? Wr : Wf)[(size_t)gr * in + k];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int n = t * 16 + i, d = n / (3 * H), gr = n - d * 3 * H;
      return (d ? bir : bif)[gr] + (gr < 2 * H ? (d ? bhr : bhf)[gr] : 0.0f);
    };
    rc |= build_layer(h->xproj, 6 * H / 16, in, 6 * H, w, &b);
    std::vector<float> hh((size_t)2 * 3 * H * H), hb((size_t)2 * H);
    std::memcpy(hh.data(), hw->w_hh, sizeof(float) * 3 * H * H);
    std::memcpy(hh.data() + (size_t)3 * H * H, hw->w_hh_reverse, sizeof(float) * 3 * H * H);
    std::memcpy(hb.data(), bhf + 2 * H, sizeof(float) * H);
    std::memcpy(hb.data() + H, bhr + 2 * H, sizeof(float) * H);
    rc |= upload(hh.data(), hh.size() * sizeof(float), (void**)&h->whh);
    rc |= upload(hb.data(), hb.size() * sizeof(float), (void**)&h->bhn);
  }
  if (rc) {
    std::string keep = wrnn::last_error();
    wrnn_bigru_destroy(h);
    wrnn::last_error() = keep;
    return 1;
  }
  *out = h;
  return 0;
}

extern "C" int wrnn_bigru_forward(wrnn_bigru* h, const float* x, const int32_t* lens, int B, int M, float padding_value,
                                  float* out, float* h_n, void* stream) {
  WRNN_REQUIRE(h != nullptr, "null bigru handle");
  WRNN_REQUIRE(B >= 0 && M >= 0, "wrnn_bigru_forward: negative size (B=%d, M=%d)", B, M);
  WRNN_REQUIRE(B <= 65535, "wrnn_bigru_forward: B=%d out of range", B);
  if (B == 0) return 0;
  WRNN_REQUIRE((x && out) || M == 0, "wrnn_bigru_forward: null tensor");
  Guard g(h->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", h->device);
  hipStream_t st = (hipStream_t)stream;
  const int in = h->in, H = h->H;
  const size_t xs = (size_t)M * B * in, ss = (size_t)B * H;
  WRNN_REQUIRE((size_t)B * M < ((size_t)1 << 30), "wrnn_bigru_forward: B*M=%zu out of range", (size_t)B * M);
  const size_t need = (h->resident || h->pair) ? (size_t)B * M * 6 * H : 2 * xs + 4 * ss;
  if (need > h->ws_floats) {
    if (h->ws) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(h->ws));
      h->ws = nullptr, h->ws_floats = 0;
    }
    WRNN_HIP(hipMalloc((void**)&h->ws, need * sizeof(float)));
    h->ws_floats = need;
  }
  if (h->resident || h->pair) {
    if (M == 0) {
      if (h_n) WRNN_HIP(hipMemsetAsync(h_n, 0, 2 * ss * sizeof(float), st));
      return 0;
    }
    float* gx = h->ws;
    const int rows = B * M;
    LinJob<2, 4, EpiLinear, 2> jx;  // the input half of every frame, both directions: [B*M] x [6H]
    if (make_lin(jx, h->xproj, Src{act_rm(x, in), in}, kNoSrc, kNoSrc, rows, rows,
                 EpiLinear{h->xproj.bias, act_rm(gx, 6 * H), 6 * H, 0, nullptr}))
      return 1;
    if (launch_tick(st, jx)) return 1;
    if (lens) {
      const size_t tot = (size_t)B * M * (2 * H / 4);
      const unsigned pb = (unsigned)std::min<size_t>((tot + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(birnn_pad_kernel, dim3(pb), dim3(256), 0, st, lens, B, M, 2 * H / 4, padding_value,
                         reinterpret_cast<f32x4*>(out));
      WRNN_LAUNCH_CHECK();
    }
    if (h->pair) {
      if (!h->err) WRNN_HIP(hipMalloc((void**)&h->err, 64));
      if ((size_t)B > h->xch_rows) {
        if (h->xch) {
          WRNN_HIP(hipDeviceSynchronize());
          WRNN_HIP(hipFree(h->xch));
          h->xch = nullptr, h->xch_rows = 0;
        }
        WRNN_HIP(hipMalloc((void**)&h->xch, (size_t)2 * B * 2 * 2 * (H / 4) * 16));
        h->xch_rows = (size_t)B;
      }
      WRNN_HIP(hipMemsetAsync(h->err, 0, 64, st));
      WRNN_HIP(hipMemsetAsync(h->xch, 0, (size_t)2 * B * 2 * 2 * (H / 4) * 16, st));   // tag 0: nothing published
      BiGruPairParams q{};
      q.gx = gx, q.whh = h->whh, q.bhn = h->bhn, q.lens = lens, q.out = out, q.h_n = h_n, q.xch = h->xch, q.err = h->err;
      q.wt_only = getenv("WRNN_BIGRU_WRITE_THROUGH") != nullptr;
      q.B = B, q.M = M;
      // four workgroups per row (two directions x two halves), one per compute unit: all of a launch's workgroups resident at once
      const int rows_per_launch = std::max(1, h->cus / 4 / 2 * 2);
      for (int b0 = 0; b0 < B; b0 += rows_per_launch) {
        q.b0 = b0, q.nb = std::min(rows_per_launch, B - b0);
        const unsigned blocks = (unsigned)((2 * q.nb + 7) / 8) * 16;
        if (wrnn::on_persistent_stream(st, [&](hipStream_t ps) -> int {
              switch (H) {
                case 192: hipLaunchKernelGGL(bigru_pair_kernel<192>, dim3(blocks), dim3(2 * 192), 0, ps, q); break;
                default: hipLaunchKernelGGL(bigru_pair_kernel<256>, dim3(blocks), dim3(2 * 256), 0, ps, q); break;
              }
              WRNN_LAUNCH_CHECK();
              return 0;
            }))
          return 1;
      }
      hipLaunchKernelGGL(bigru_poison_kernel, dim3(256), dim3(256), 0, st, h->err, out, (size_t)B * M * 2 * H, h_n, (size_t)2 * B * H);
      WRNN_LAUNCH_CHECK();
      return 0;
    }
    const dim3 grid(2 * (unsigned)B), block(4 * (unsigned)H);
    switch (H) {
      case 32: hipLaunchKernelGGL(bigru_resident_kernel<32>, grid, block, 0, st, gx, h->whh, h->bhn, lens, B, M, out, h_n); break;
      case 64: hipLaunchKernelGGL(bigru_resident_kernel<64>, grid, block, 0, st, gx, h->whh, h->bhn, lens, B, M, out, h_n); break;
      case 96: hipLaunchKernelGGL(bigru_resident_kernel<96>, grid, block, 0, st, gx, h->whh, h->bhn, lens, B, M, out, h_n); break;
      default: hipLaunchKernelGGL(bigru_resident_kernel<128>, grid, block, 0, st, gx, h->whh, h->bhn, lens, B, M, out, h_n); break;
    }
    WRNN_LAUNCH_CHECK();
    return 0;
  }
  float* xf = h->ws;
  float* xr = xf + xs;
  float* state = xr + xs;  // [dir][ping-pong] x [H/4][B][4]
  auto hbuf = [&](int dir, int pp) { return state + ((size_t)dir * 2 + pp) * ss; };
  WRNN_HIP(hipMemsetAsync(state, 0, 4 * ss * sizeof(float), st));
  if (M > 0) {
    const size_t total = (size_t)M * (in / 4) * B;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(birnn_pack_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), lens, B, M,
                       in / 4, reinterpret_cast<f32x4*>(xf), reinterpret_cast<f32x4*>(xr));
    WRNN_LAUNCH_CHECK();
    if (lens) {
      const size_t tot = (size_t)B * M * (2 * H / 4);
      const unsigned pb = (unsigned)std::min<size_t>((tot + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(birnn_pad_kernel, dim3(pb), dim3(256), 0, st, lens, B, M, 2 * H / 4, padding_value,
                         reinterpret_cast<f32x4*>(out));
      WRNN_LAUNCH_CHECK();
    }
  }
  auto frames = [&](auto proto) -> int {
    using Job = decltype(proto);
    for (int s = 0; s < M; ++s) {
      const int pi = s & 1, po = pi ^ 1;
      Job jf, jr;
      for (int dir = 0; dir < 2; ++dir) {
        const Act xa = act_il((dir ? xr : xf) + (size_t)s * B * in, B, in);
        const Act hp = act_il(hbuf(dir, pi), B, H), ho = act_il(hbuf(dir, po), B, H);
        const Layer& L = dir ? h->rev : h->fwd;
        if (make_lin(dir ? jr : jf, L, Src{xa, in}, Src{hp, H}, kNoSrc, B, B,
                     EpiBiGru{L.bias, hp, ho, lens, out, M, 2 * H, dir * H, s, dir}))
          return 1;
      }
      if (launch_tick(st, jf, jr)) return 1;
    }
    return 0;
  };
  if (H >= 192 ? frames(BiGruJobWide{}) : frames(BiGruJob{})) return 1;
  if (h_n) {
    const int last = M & 1;  // buffer holding the state after M steps
    for (int dir = 0; dir < 2; ++dir) {
      hipLaunchKernelGGL(birnn_state_kernel, dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, st, hbuf(dir, last), B, H,
                         h_n + (size_t)dir * ss);
      WRNN_LAUNCH_CHECK();
    }
  }
  return 0;
}
