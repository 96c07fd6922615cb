// Shared device/host machinery of libwrnn_mi355.so (gfx950): activation views, small-batch GEMM jobs on
// v_mfma_f32_16x16x4_f32 with fused epilogues, the "tick" kernel that runs up to three independent jobs
// in one launch, and the host-side weight re-layout.  Included by taco_decoder.hip and bilstm.hip.
#pragma once

#include <cmath>
#include <algorithm>
#include <cstring>
#include <functional>
#include <type_traits>
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

// A product / sum that is rounded on its own: hipcc contracts a * b + c into an fma wherever it sees one (and HIP's
// __fmul_rn is a plain a * b), so code that must give the same bits in two places pins the intermediate in a register.
__device__ __forceinline__ float mul_rn(float a, float b) {
  float r = a * b;
  asm volatile("" : "+v"(r));
  return r;
}
__device__ __forceinline__ float add_rn(float a, float b) {
  float r = a + b;
  asm volatile("" : "+v"(r));
  return r;
}

// Workgroup barrier for LDS-only hand-offs.  __syncthreads() carries a workgroup-scope fence, which on
// gfx9 drains vmcnt as well: every global load still in flight (operands prefetched for a later phase)
// would have to land first.  Here only the LDS counter is drained.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic builds only (tools/microbench/stage_bench.hip): per-wave phase timestamps (100 MHz).
#ifdef WRNN_STAMP
__device__ unsigned long long* g_stamp = nullptr;
#define STAMP(bid, i)                                                                        \
  do {                                                                                       \
    if (g_stamp && (threadIdx.x & 63) == 0) {                                                                                        \
      g_stamp[((size_t)(bid) * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();     \
      g_stamp[((size_t)(bid) * 4 + (threadIdx.x >> 6)) * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime();     \
    }                                                                                        \
  } while (0)
#else
#define STAMP(bid, i)
#endif

// ----------------------------------------------------------------------------- activation views
// Activations between stages live in one of two layouts:
//   row-major           p[row*ld + n]                          (external tensors, step API)
//   step-major interleaved  rows are (step, b) = (row / B, row % B):
//                       p[step*ss + ((n>>2)*B + b)*4 + (n&3)]  (internal buffers of the loops)
// In the interleaved layout the MFMA B-operand load of a wave (16 batch rows x 4 k-quads) is four
// contiguous 256-B runs instead of sixteen 64-B pieces of sixteen rows: measured 7.3 -> 5.1 us on
// the K=1024 LSTM stage (tools/microbench/ablate_bench.hip).
struct RowId {
  int row, st, b;  // activation row = (step st, batch row b)
};
__device__ __forceinline__ RowId make_rowid(int row, int B, int multi) {
  if (!multi) return RowId{row, 0, row};  // per-step jobs: no division on the critical path
  const int st = row / B;
  return RowId{row, st, row - st * B};
}

struct Act {
  float* p;
  int il, ld, B;
  long ss;
  __device__ __forceinline__ float* rowbase(const RowId& r) const {
    return il ? p + (size_t)r.st * ss + r.b * 4 : p + (size_t)r.row * ld;
  }
  __device__ __forceinline__ int kmul() const { return il ? B : 1; }  // offset of feature k (k%4==0) = k*kmul
  __device__ __forceinline__ size_t eoff(int n) const { return il ? (size_t)(n >> 2) * 4 * B + (n & 3) : (size_t)n; }
};

struct Src {
  Act a;
  int k;  // features used (multiple of 4)
};

// Epilogues: prefetch() is called BEFORE the K loop (its loads overlap the operand round trip),
// apply() after the cross-wave reduction.  (tile, u, row): the thread holds accumulator rows
// 4u..4u+3 of weight tile `tile` for activation row `row`.
struct EpiLinear {
  const float* bias;  // [tiles*16] or null
  Act y;
  int N, relu;
  const float* drop_u;  // row-major [rows][N] or null
  Act add = Act{nullptr, 0, 0, 1, 0};  // optional per-row addend (same shape as y), e.g. a hoisted partial product
  // optional hand-over of the result to a job of the SAME launch (LinJob's HAND source): tagged granule pairs in EpiGru's
  // layout, {y(2p), tag, y(2p + 1), tag} at hand[((row / 16) * hand_pairs + p) * 16 + row % 16]; the data is the flag
  unsigned long long* hand = nullptr;
  unsigned hand_tag = 0;
  int hand_pairs = 0;  // N / 2
  struct Pre { f32x4 b, u; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    const int n = tile * 16 + u * 4;
    q.b = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (add.p) q.b += *reinterpret_cast<const f32x4*>(add.rowbase(row) + add.eoff(n));
    q.u = (drop_u && n + 3 < N) ? *reinterpret_cast<const f32x4*>(drop_u + (size_t)row.row * N + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const int n = tile * 16 + u * 4;
    if (n >= N) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float val = v[e] + q.b[e];
      if (relu) val = fmaxf(val, 0.0f);
      if (drop_u) val = q.u[e] >= kDropP ? val * (1.0f / (1.0f - kDropP)) : 0.0f;
      v[e] = val;
    }
    *reinterpret_cast<f32x4*>(y.rowbase(row) + y.eoff(n)) = v;  // N % 4 == 0 for every plain layer
    if (hand) {
      using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
      unsigned long long* dst = hand + (((size_t)(row.row >> 4) * hand_pairs + (n >> 1)) * 16 + (row.row & 15)) * 2;
      const u32x4_ g0 = {__float_as_uint(v[0]), hand_tag, __float_as_uint(v[1]), hand_tag};
      const u32x4_ g1 = {__float_as_uint(v[2]), hand_tag, __float_as_uint(v[3]), hand_tag};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst), "v"(g0) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst + 32), "v"(g1) : "memory");
    }
  }
};

struct EpiGru {  // tile rows: 4 units x (r, z, gi_n, gh_n)
  const float* bias;  // tile order
  Act h_prev, h_out;
  Act x_in = Act{nullptr, 0, 0, 1, 0}, x_out = Act{nullptr, 0, 0, 1, 0};  // optional residual x_out = x_in + h
  // optional hand-over of the new state to a job of the SAME launch (QueryJob, taco_decoder.hip): tagged granule pairs
  // {h(unit 2p), tag, h(unit 2p + 1), tag} at hand[((row / 16) * (units / 2) + p) * 16 + row % 16], written by one store
  // instruction per block (32 lanes x 16 bytes, write-through); the data is the flag
  unsigned long long* hand = nullptr;
  unsigned hand_tag = 0;
  int hand_pairs = 0;  // units / 2
  struct Pre { f32x4 b; float h, x; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.h = h_prev.rowbase(row)[h_prev.eoff(tile * 4 + u)];
    q.x = x_in.p ? x_in.rowbase(row)[x_in.eoff(tile * 4 + u)] : 0.0f;
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const float r = sigmoidf_(v[0] + q.b[0]);
    const float z = sigmoidf_(v[1] + q.b[1]);
    const float n = tanhf_(v[2] + q.b[2] + r * (v[3] + q.b[3]));
    const float h = (1.0f - z) * n + z * q.h;
    h_out.rowbase(row)[h_out.eoff(tile * 4 + u)] = h;
    if (x_out.p) x_out.rowbase(row)[x_out.eoff(tile * 4 + u)] = q.x + h;
    if (hand) {  // (the lane 16 up holds unit u + 1 of the same row; both are active or neither)
      const float h_next = __shfl_down(h, 16, 64);
      if ((u & 1) == 0) {
        using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
        const u32x4_ g = {__float_as_uint(h), hand_tag, __float_as_uint(h_next), hand_tag};
        unsigned long long* dst = hand + (((size_t)(row.row >> 4) * hand_pairs + tile * 2 + (u >> 1)) * 16 + (row.row & 15)) * 2;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst), "v"(g) : "memory");  // (wait state: store data hazard)
      }
    }
  }
};

struct EpiLstm {  // tile rows: 4 units x (i, f, g, o)
  const float* bias;  // tile order, b_ih + b_hh
  Act c_prev, c_out;               // cell state: only ever touched by this epilogue
  Act h_prev, h_out, x_in, x_out;  // x_out = x_in + h (residual, tacotron.py:219,227)
  const float* zone_u;             // row-major [rows][l] or null
  int l;
  struct Pre { f32x4 b; float c, hp, x, z; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    const int U = tile * 4 + u;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.c = c_prev.rowbase(row)[c_prev.eoff(U)];
    q.x = x_in.rowbase(row)[x_in.eoff(U)];
    q.z = zone_u ? zone_u[(size_t)row.row * l + U] : 1.0f;
    q.hp = zone_u ? h_prev.rowbase(row)[h_prev.eoff(U)] : 0.0f;
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const int U = tile * 4 + u;
    const float ig = sigmoidf_(v[0] + q.b[0]);
    const float fg = sigmoidf_(v[1] + q.b[1]);
    const float gg = tanhf_(v[2] + q.b[2]);
    const float og = sigmoidf_(v[3] + q.b[3]);
    const float c = fg * q.c + ig * gg;
    float h = og * tanhf_(c);
    if (q.z < kZoneP) h = q.hp;  // zoneout keeps the previous hidden value (cell state is not zoned out)
    c_out.rowbase(row)[c_out.eoff(U)] = c;
    h_out.rowbase(row)[h_out.eoff(U)] = h;
    x_out.rowbase(row)[x_out.eoff(U)] = q.x + h;
  }
};

struct EpiRaw {  // the accumulators as they are: partial sums for a later job's epilogue (gate-interleaved, [N/4][B][4])
  Act y;
  struct Pre {};
  __device__ __forceinline__ Pre prefetch(int, int, const RowId&) const { return Pre{}; }
  __device__ __forceinline__ void apply(const Pre&, int tile, int u, const RowId& row, f32x4 v) const {
    *reinterpret_cast<f32x4*>(y.rowbase(row) + y.eoff(tile * 16 + u * 4)) = v;
  }
};

struct EpiMel {  // weight rows gathered as j = m*r + f  <->  mel_proj row m*20 + f
  float* out;
  long sb, sm;  // strides of the batch and mel dims of `out`
  int r, N;  // rows are (step k, batch b); frame index = k*r + f
  int* alive = nullptr;  // free-running stop vote of this group: set to 1 unless every frame value < thr
  float thr = 0.0f;
  struct Pre {};
  __device__ __forceinline__ Pre prefetch(int, int, const RowId&) const { return Pre{}; }
  __device__ __forceinline__ void apply(const Pre&, int tile, int u, const RowId& row, f32x4 v) const {
    const int k = row.st, b = row.b;
    bool any_alive = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = tile * 16 + u * 4 + e;
      if (j < N) {
        const int m = j / r, f = j - m * r;
        out[(size_t)b * sb + (size_t)m * sm + k * r + f] = v[e];
        any_alive |= !(v[e] < thr);
      }
    }
    if (alive && any_alive) *alive = 1;  // every writer stores the same value: no atomic needed
  }
};

// ----------------------------------------------------------------------------- jobs
// A launch ("tick") runs up to three independent jobs side by side: block ranges are dealt to
// the jobs in order.  Every job type has nblocks(), lds_bytes() and run(block, smem).

// D[16 weight rows][16*NB activation rows] = W_tile . X^T, K split over the 4 waves of the WG.
// A operand = weights (lane: row i = lane&15, k-slot s = lane>>4), B operand = X^T (lane:
// activation row j = lane&15, k-slot s).  Inside a 16-wide K chunk slot s owns k = 16*kc + 4*s + m
// for MFMA m = 0..3, so both operands are one float4 per lane per chunk.
// A first source that does not exist as a vector yet: per activation row, `chunks` partial vectors with their softmax
// statistics (the attention's per-chunk records, taco_decoder.hip); the job forms x = sum_c scale_c * part_c itself, in
// registers, so that the producer needs no combine step (and no hand-over inside its own launch).
struct CombSrc {
  const float* part = nullptr;  // [rows][chunks][rec]: [0, s0.k) partial vector, [ml_off] chunk max, [ml_off + 1] chunk sum
  int chunks = 0, rec = 0, ml_off = 0;
};

// chunk scales exp(m_c - M) / sum_c l_c exp(m_c - M) from up to 8 chunk statistics; the sum is taken in the order of a
// 64-lane xor-shuffle tree over lanes 0..7 (what the in-launch combine of the attention job computes), so that both
// paths give the same bits
template <int MAXC>
__device__ __forceinline__ void chunk_scales(const float2 (&ml)[MAXC], int chunks, float (&scale)[MAXC]) {
  static_assert(MAXC == 4 || MAXC == 8, "tree order written out for 4 or 8 chunks");
  float M = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c < chunks) M = fmaxf(M, ml[c].x);
  float sc[8], t[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    sc[c] = (c < MAXC && c < chunks) ? __expf(ml[c < MAXC ? c : 0].x - M) : 0.0f;
    t[c] = (c < MAXC && c < chunks) ? mul_rn(ml[c < MAXC ? c : 0].y, sc[c]) : 0.0f;  // rounded: never fused into the adds
  }
  const float Ls = add_rn(add_rn(add_rn(t[0], t[4]), add_rn(t[2], t[6])), add_rn(add_rn(t[1], t[5]), add_rn(t[3], t[7])));
#pragma unroll
  for (int c = 0; c < MAXC; ++c) scale[c] = sc[c] / Ls;
}

// COMB = 0: plain sources.  COMB = 4 or 8: source 0 is a CombSrc of at most COMB chunks (its K must be one load batch
// per wave, 64 * KU features, and the rest of K at most two more batches: the attention GRU's shape).
// HAND = 16 or 8: behind the plain sources the layer has 256 / 128 more K columns (its last HAND chunks) whose operand does not
// exist when the launch starts: another job of the SAME launch publishes it as tagged granule pairs (EpiGru::hand /
// EpiLinear::hand / AttnJob::hand, 16 HAND features of the block's 16 rows).  The block does its plain sources first, then requests the weights of
// the hand chunks, sweeps the granules until every pair shows `htag` (bounded; QueryJob's sweep), stages them in LDS and
// finishes.  Its blocks must FOLLOW the producer's in the launch (every producer dispatched before the first consumer waits).
template <int TM, int NB, class Epi, int KU = 4, int COMB = 0, int HAND = 0>
struct LinJob {
  const float4* wt;
  int KC, tiles, tgroups, groups, rows, B, multi;  // rows = steps*B activation rows; multi = rows span several steps
  Src s0, s1, s2;
  Epi epi;
  CombSrc comb;
  const unsigned long long* hand = nullptr;   // HAND: [row groups][8 HAND pairs][16 rows] granule pairs
  unsigned* herr = nullptr;                   // HAND: error word (a sweep that never completed)
  unsigned htag = 0;
  int hstride = 0;                            // HAND: pairs per producing block (a row's pairs come from 8 HAND / hstride blocks); 0: no sentinel wait
  static constexpr int kHXS = 260;            // LDS row stride of the staged hand operand (floats)
  __host__ __device__ int nblocks() const { return tgroups * groups; }
  static constexpr size_t lds_bytes() { return sizeof(f32x4) * 4 * TM * NB * 64 + (HAND ? sizeof(float) * 16 * kHXS : 0); }
  static constexpr int kU = KU;                                // K chunks per load batch
  static constexpr int kEpi = (TM * NB * 64 + 255) / 256;      // epilogue items per thread

  struct Batch {
    float4 w[kU][TM], x[kU][NB];
  };

  // TM weight tiles (16 rows each) x NB activation tiles (16 rows each) per block.
  __device__ __forceinline__ void run(int bid, char* smem) const {
    f32x4(*red)[TM * NB][64] = reinterpret_cast<f32x4(*)[TM * NB][64]>(smem);
    const int tile0 = (bid % tgroups) * TM;
    const int row0 = (bid / tgroups) * (16 * NB);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: chunk counts are wave-uniform
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, s = lane >> 4;
    STAMP(bid, 0);
    // Every global load below is UNCONDITIONAL, from clamped (always valid) indices; what must not count
    // is kept out of the result instead (MFMAs under wave-uniform guards, epilogue items under pre_ok).
    // A load under a per-lane condition compiles to a divergent branch, and a load under any branch makes
    // the number of loads in flight unknown to the compiler, which then waits with vmcnt(0): every
    // batch would land before the first MFMA and the software pipeline below would be a serial loop.
    // epilogue operands first: item e = (tm, nb, l) = (e / (NB*64), (e / 64) % NB, e % 64)
    typename Epi::Pre pre[kEpi];
    RowId pre_row[kEpi];
    bool pre_ok[kEpi];
#pragma unroll
    for (int i = 0; i < kEpi; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int tm = idx / (NB * 64), nb = (idx >> 6) % NB;
      const int row = row0 + nb * 16 + (idx & 15);
      pre_ok[i] = idx < TM * NB * 64 && row < rows && tile0 + tm < tiles;
      pre_row[i] = make_rowid(row < rows ? row : 0, B, multi);
      pre[i] = epi.prefetch(min(tile0 + tm, tiles - 1), (idx & 63) >> 4, pre_row[i]);
    }
    f32x4 acc[TM][NB];
    const float* xb[3][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = row0 + nb * 16 + j;
      const RowId rr = make_rowid(row < rows ? row : 0, B, multi);  // rows past the end recompute row 0; never stored
      xb[0][nb] = s0.a.rowbase(rr);
      xb[1][nb] = s1.k ? s1.a.rowbase(rr) : xb[0][nb];
      xb[2][nb] = s2.k ? s2.a.rowbase(rr) : xb[0][nb];
    }
    const int km0 = s0.a.kmul(), km1 = s1.a.kmul(), km2 = s2.a.kmul();
    const int ktot = s0.k + s1.k + s2.k;
    const float4* wp[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) wp[tm] = wt + (size_t)min(tile0 + tm, tiles - 1) * KC * 64 + lane;
    // Each wave owns chunks kc = wave, wave+4, ...  Loads are issued a batch of kU chunks ahead of
    // the MFMAs that consume them (two register batches, software pipelined): the stage is bound by
    // operand latency, not by arithmetic.
    const int KCL = KC - HAND;   // chunks of the plain sources
    const int n_it = (KCL - wave + 3) >> 2;
    auto load = [&](Batch& t, int i0) {
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int kc = min(wave + 4 * (i0 + q), KCL - 1);  // chunks past the wave's share: loaded, not multiplied
        int k = min(kc * 16 + s * 4, ktot - 4);           // K padding of the last chunk: its weights are zero
        int which = 0, km = km0;
        if (k >= s0.k) {
          k -= s0.k, which = 1, km = km1;
          if (k >= s1.k) k -= s1.k, which = 2, km = km2;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) t.w[q][tm] = wp[tm][(size_t)kc * 64];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float* base = which == 0 ? xb[0][nb] : which == 1 ? xb[1][nb] : xb[2][nb];
          t.x[q][nb] = *reinterpret_cast<const float4*>(base + (size_t)k * km);
        }
      }
    };
    auto mma = [&](const Batch& t, int i0) {
#pragma unroll
      for (int q = 0; q < kU; ++q)
        if (i0 + q < n_it) {  // wave-uniform
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              acc[tm][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w[q][tm].x, t.x[q][nb].x, acc[tm][nb], 0, 0, 0);
              acc[tm][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w[q][tm].y, t.x[q][nb].y, acc[tm][nb], 0, 0, 0);
              acc[tm][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w[q][tm].z, t.x[q][nb].z, acc[tm][nb], 0, 0, 0);
              acc[tm][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w[q][tm].w, t.x[q][nb].w, acc[tm][nb], 0, 0, 0);
            }
        }
    };
    // Straight-line pipelines of 1..4 batches (two register batches, the refill of one issued under the
    // MFMAs of the other), selected by a wave-uniform switch: a rolled loop would carry the batches
    // across its back edge as copies, and the copies wait for the loads.  K > 64*kU chunks per wave
    // (K > 1024 at kU = 4) takes several trips, draining in between.
    auto pipe = [&](auto nbat, int base) {
      constexpr int N = decltype(nbat)::value;
      Batch ba, bb;
      load(ba, base);
      if constexpr (N >= 2) load(bb, base + kU);
      mma(ba, base);
      if constexpr (N >= 3) load(ba, base + 2 * kU);
      if constexpr (N >= 2) mma(bb, base + kU);
      if constexpr (N >= 4) load(bb, base + 3 * kU);
      if constexpr (N >= 3) mma(ba, base + 2 * kU);
      if constexpr (N >= 4) mma(bb, base + 3 * kU);
    };
    if constexpr (COMB > 0) {
      static_assert(NB == 1, "combined source: one activation tile per block");
      // the wave's first kU chunks are the combined source: request its partial vectors (all chunks), the chunk
      // statistics and the weights, then the two plain batches that follow, and only then start to compute
      const int row = row0 + j < rows ? row0 + j : 0;
      const float* rec = comb.part + (size_t)row * comb.chunks * comb.rec;
      float4 w0[kU][TM], xc[kU][COMB];
      float2 ml[COMB];
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int kc = wave + 4 * q;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) w0[q][tm] = wp[tm][(size_t)kc * 64];
#pragma unroll
        for (int c = 0; c < COMB; ++c)
          xc[q][c] = *reinterpret_cast<const float4*>(rec + (size_t)min(c, comb.chunks - 1) * comb.rec + kc * 16 + s * 4);
      }
#pragma unroll
      for (int c = 0; c < COMB; ++c)
        ml[c] = *reinterpret_cast<const float2*>(rec + (size_t)min(c, comb.chunks - 1) * comb.rec + comb.ml_off);
      Batch ba, bb;
      load(ba, kU);
      load(bb, 2 * kU);
      float scale[COMB];
      chunk_scales<COMB>(ml, comb.chunks, scale);
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < COMB; ++c)
          if (c < comb.chunks) {  // wave-uniform
            x.x = fmaf(xc[q][c].x, scale[c], x.x), x.y = fmaf(xc[q][c].y, scale[c], x.y);
            x.z = fmaf(xc[q][c].z, scale[c], x.z), x.w = fmaf(xc[q][c].w, scale[c], x.w);
          }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q][tm].x, x.x, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q][tm].y, x.y, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q][tm].z, x.z, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[q][tm].w, x.w, acc[tm][0], 0, 0, 0);
        }
      }
      mma(ba, kU);
      mma(bb, 2 * kU);
    } else {
    for (int base = 0; base < n_it; base += 4 * kU) {
      const int left = (n_it - base + kU - 1) / kU;
      if (left >= 4) pipe(std::integral_constant<int, 4>(), base);
      else if (left == 3) pipe(std::integral_constant<int, 3>(), base);
      else if (left == 2) pipe(std::integral_constant<int, 2>(), base);
      else pipe(std::integral_constant<int, 1>(), base);
    }
    }
    if constexpr (HAND) {
      static_assert(NB == 1 && COMB == 0 && (HAND == 16 || HAND == 8), "hand source: one activation tile per block, 256 or 128 features");
      constexpr int HP = HAND * 8, HQ = HAND / 4, HL = HAND / 2;   // pairs per row, chunks per wave, sweep loads per thread
      float* xs = reinterpret_cast<float*>(smem + sizeof(f32x4) * 4 * TM * NB * 64);   // [16 rows][kHXS]
      const int tid = threadIdx.x;
      float4 wh[HQ][TM];   // the wave's hand chunks KCL + wave + 4 q: requested before the wait
#pragma unroll
      for (int q = 0; q < HQ; ++q)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) wh[q][tm] = wp[tm][(size_t)(KCL + wave + 4 * q) * 64];
      using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
      const unsigned long long* g0 = hand + (size_t)(row0 >> 4) * HP * 16 * 2 + 2 * tid;
      const bool live = row0 + (tid & 15) < rows;   // rows past the batch are never published
      u32x4_ v[8] = {};
      bool failed = hand == nullptr;   // (a job without its array: poisoned output, never a wild load)
      // Wait on ONE pair per producing block first (wave 0, at most 1 KB per poll, asleep in between): hundreds of waiting
      // blocks sweeping 32 KB each take the memory path that the producer is bound by (measured: the attention launch
      // with rnn_input and LSTM1 waiting in it 21.5 us, against 9.2 + 7.9 as two launches).  The full sweep below then
      // checks every tag (the lines of one block's stores can land apart).
      if (hstride > 0 && wave == 0 && !failed) {
        const int nsent = HP / hstride;
        const unsigned long long* gs = hand + (size_t)(row0 >> 4) * HP * 16 * 2;
        for (unsigned spins = 0; spins < (1u << 16); ++spins) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int si = (lane >> 4) + 4 * i;
            if (4 * i < nsent) {   // wave-uniform
              u32x4_ sv;
              const unsigned long long* a = gs + ((size_t)min(si, nsent - 1) * hstride * 16 + (lane & 15)) * 2;
              asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(sv) : "v"(a) : "memory");
              ok &= row0 + (lane & 15) >= rows || (sv.y == htag && sv.w == htag);
            }
          }
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(32);
        }
      }
      __syncthreads();
      for (unsigned spins = 0; !failed; ++spins) {
        bool ok = true;
        // (the loads and their wait in ONE asm statement, see QueryJob)
        if constexpr (HL == 8) {
          asm volatile(
              "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
              "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
              "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
              "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
              : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
              : "v"(g0), "v"(g0 + 512), "v"(g0 + 1024), "v"(g0 + 1536), "v"(g0 + 2048), "v"(g0 + 2560), "v"(g0 + 3072), "v"(g0 + 3584)
              : "memory");
        } else {
          asm volatile(
              "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
              "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
              : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
              : "v"(g0), "v"(g0 + 512), "v"(g0 + 1024), "v"(g0 + 1536)
              : "memory");
        }
#pragma unroll
        for (int i = 0; i < HL; ++i) ok &= !live || (v[i].y == htag && v[i].w == htag);
        if (__all(ok)) break;
        if (spins > (1u << 18)) {
          if (lane == 0) atomicOr(herr, 1u);
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);   // (the producer is bound by the same memory path)
      }
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const int idx = tid + 256 * i, up = idx >> 4, row = idx & 15;
        float2 hv = make_float2(__uint_as_float(v[i].x), __uint_as_float(v[i].z));
        if (!live) hv = make_float2(0.f, 0.f);
        if (failed) hv = make_float2(NAN, NAN);
        *reinterpret_cast<float2*>(xs + row * kHXS + 2 * up) = hv;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(xs + j * kHXS + (wave + 4 * q) * 16 + s * 4);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[q][tm].x, x.x, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[q][tm].y, x.y, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[q][tm].z, x.z, acc[tm][0], 0, 0, 0);
          acc[tm][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[q][tm].w, x.w, acc[tm][0], 0, 0, 0);
        }
      }
    }
    STAMP(bid, 2);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) red[wave][tm * NB + nb][lane] = acc[tm][nb];
    lds_barrier();
    STAMP(bid, 3);
#pragma unroll
    for (int i = 0; i < kEpi; ++i) {
      if (pre_ok[i]) {
        const int idx = threadIdx.x + 256 * i;
        const int t = idx >> 6, l = idx & 63;
        const f32x4 v = (red[0][t][l] + red[1][t][l]) + (red[2][t][l] + red[3][t][l]);
        epi.apply(pre[i], tile0 + idx / (NB * 64), l >> 4, pre_row[i], v);  // lane holds tile rows 4u..4u+3, u = l>>4
      }
    }
    STAMP(bid, 4);
  }
};

struct NoJob {
  __host__ __device__ int nblocks() const { return 0; }
  static constexpr size_t lds_bytes() { return 0; }
  __device__ __forceinline__ void run(int, char*) const {}
};

template <class J0, class J1, class J2>
__global__ __launch_bounds__(256) void tick_kernel(J0 j0, J1 j1, J2 j2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  {
    // The job descriptors are kernel arguments: a few hundred bytes that the compiler loads field by field, lazily,
    // behind the branches below, each new 64-byte line a scalar-cache miss of its own on the way to the first
    // operand load.  Touch every line of the segment at once here: the lazy loads then hit the scalar cache.
    const __attribute__((address_space(4))) unsigned* ka =
        (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int kBytes = (int)(sizeof(J0) + sizeof(J1) + sizeof(J2));
    unsigned acc = 0;
#pragma unroll
    for (int o = 0; o < kBytes; o += 64) acc |= ka[o / 4];
    asm volatile("" ::"s"(acc));
  }
  int bid = blockIdx.x;
  const int n0 = j0.nblocks();
  if (bid < n0) {
    j0.run(bid, smem);
    return;
  }
  bid -= n0;
  const int n1 = j1.nblocks();
  if (bid < n1) {
    j1.run(bid, smem);
    return;
  }
  bid -= n1;
  if (bid < j2.nblocks()) j2.run(bid, smem);
}

// The same for any number of jobs (the free-running loop's first launch carries five): block ranges dealt in order.
template <class J, class... R>
__device__ __forceinline__ void run_jobs(int bid, char* smem, const J& j, const R&... r) {
  const int n = j.nblocks();
  if (bid < n) {
    j.run(bid, smem);
    return;
  }
  if constexpr (sizeof...(R) > 0) run_jobs(bid - n, smem, r...);
}
template <class... Js>
__global__ __launch_bounds__(256) void tick_kernel_n(Js... js) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  {  // (touch every line of the kernel-argument segment at once: see tick_kernel)
    const __attribute__((address_space(4))) unsigned* ka =
        (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int kBytes = (int)(sizeof(Js) + ...);
    unsigned acc = 0;
#pragma unroll
    for (int o = 0; o < kBytes; o += 64) acc |= ka[o / 4];
    asm volatile("" ::"s"(acc));
  }
  run_jobs(blockIdx.x, smem, js...);
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

inline Act act_rm(const float* p, int ld) { return Act{const_cast<float*>(p), 0, ld, 1, 0}; }
// interleaved view of a buffer holding `K` features for B rows per step (see Act)
inline Act act_il(const float* p, int B, int K) { return Act{const_cast<float*>(p), 1, 0, B, (long)K * B}; }
const Src kNoSrc{Act{nullptr, 0, 0, 1, 0}, 0};

template <int TM, int NB, class Epi, int KU, int COMB, int HAND>
int make_lin(LinJob<TM, NB, Epi, KU, COMB, HAND>& J, const Layer& L, Src s0, Src s1, Src s2, int rows, int B, const Epi& epi) {
  if (COMB > 0)
    WRNN_REQUIRE(s0.k == 64 * KU && s1.k + s2.k <= 2 * 64 * KU, "linear16: combined source needs K0 = %d and the rest <= %d", 64 * KU, 128 * KU);
  WRNN_REQUIRE(s0.k % 4 == 0 && s1.k % 4 == 0 && s2.k % 4 == 0, "linear16: source widths must be multiples of 4");
  if (HAND > 0) WRNN_REQUIRE((s0.k + s1.k + s2.k) % 16 == 0 && rows <= B, "linear16: a hand source follows whole K chunks, one step");
  WRNN_REQUIRE(s0.k + s1.k + s2.k + 16 * HAND <= L.KC * 16 && s0.k + s1.k + s2.k + 16 * HAND > (L.KC - 1) * 16,
               "linear16: source widths %d+%d+%d do not match the layer's K", s0.k, s1.k, s2.k);
  J.wt = L.wt, J.KC = L.KC, J.tiles = L.tiles, J.tgroups = (L.tiles + TM - 1) / TM;
  J.rows = rows, J.B = B, J.multi = rows > B;
  J.groups = rows > 0 ? (rows + 16 * NB - 1) / (16 * NB) : 0;
  J.s0 = s0, J.s1 = s1, J.s2 = s2, J.epi = epi;
  return 0;
}

template <class J0, class J1 = NoJob, class J2 = NoJob>
int launch_tick(hipStream_t st, const J0& j0, const J1& j1 = NoJob(), const J2& j2 = NoJob()) {
  const long blocks = (long)j0.nblocks() + j1.nblocks() + j2.nblocks();
  if (blocks == 0) return 0;
  WRNN_REQUIRE(blocks < (1L << 31), "tick: too many blocks");
  const size_t lds = std::max({(size_t)j0.lds_bytes(), (size_t)j1.lds_bytes(), (size_t)j2.lds_bytes()});
  hipLaunchKernelGGL((tick_kernel<J0, J1, J2>), dim3((unsigned)blocks), dim3(256), lds, st, j0, j1, j2);
  WRNN_LAUNCH_CHECK();
  return 0;
}

template <class... Js>
int launch_tick_n(hipStream_t st, const Js&... js) {
  const long blocks = ((long)js.nblocks() + ...);
  if (blocks == 0) return 0;
  WRNN_REQUIRE(blocks < (1L << 31), "tick: too many blocks");
  const size_t lds = std::max({(size_t)js.lds_bytes()...});
  hipLaunchKernelGGL((tick_kernel_n<Js...>), dim3((unsigned)blocks), dim3(256), lds, st, js...);
  WRNN_LAUNCH_CHECK();
  return 0;
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

}  // namespace
