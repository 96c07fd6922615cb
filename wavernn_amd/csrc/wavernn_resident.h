// Unbatched WaveRNN generation (one row: the non-fold, sample-by-sample path of BASELINE configs[4]) as ONE persistent,
// weights-resident launch per call.  Included by wavernn.hip; replaces its 5-launches-per-sample loop where a sample is
// pure latency.  Like the loop it replaces, the model is build-defined (the vocoder is absent from the mounted reference
// tree, see wavernn.hip) and parity is unpinned except for the MOL sampler (mol.h).
//
// 160 working workgroups x 512 threads (a grid of 256, launched cooperatively, all co-resident, at most one per compute unit), in
// three GROUPS; a workgroup keeps its rows in REGISTERS for the whole call:
//   C1  64 workgroups x 8 units   draw x(t-1) from the logits, first GRU cell (elementwise in x, see below)       XCDs 0, 1
//   C2  64 workgroups x 8 units   second GRU cell over [x1 ; a2]                                                   XCDs 2, 3
//   F   32 workgroups x 16 rows of fc1 (over [x2 ; a3], relu), of fc2 (over [y1 ; a4], relu) and of fc3             XCD 4
// (blocks are dealt to the eight XCDs in turn, so "XCD x" = the 32 blocks with blockIdx % 8 == x; the blocks of XCDs 5-7 leave
//  at once).  Round 3: the three fc layers were groups of their own before (224 workgroups, five hand-overs across XCDs, 6.6-6.9 us
//  per sample); as ONE group on one XCD their two hand-overs stay in that XCD's L2 -- 6.0 us with write-through stores, 5.4 with
//  plain stores, which the group uses only after it has SEEN that its 32 workgroups share an XCD (HW_REG_XCC_ID, published by
//  every workgroup at kernel entry); blocks in order (WRNN_RESIDENT_LINEAR_MAP): 6.6, same samples.
// A wave owns whole rows (one GRU unit = 3 + 3 rows, or two fc rows): read the gathered vector from LDS, 8-9 FMAs per
// lane and row, DPP wave sums, the pointwise rest in the same wave; the eight waves' results meet in LDS and leave as ONE
// store instruction (lanes 0-7, 16 bytes each = two {tag, value} granules: one 128-byte write-through write per workgroup
// and stage).  Eight separate 16-byte stores into the same line, one per wave, cost 0.8 us more PER HAND-OVER (10.2 vs
// 6.8 us per sample): partial write-through writes to one line serialise on the memory side.
//
// One sample = five hand-overs (a nonlinearity separates every pair of layers): logits -> C1 -> [x1, h1] -> C2 ->
// [x2, h2] -> fc1 -> [y1] -> fc2 -> [y2] -> fc3 -> logits.  A hand-over array has one 128-byte slot per producing workgroup
// (8 granule pairs, pair k written by wave k); only the NEXT stage's group sweeps it, with 16-byte sc1 loads, a pair per thread.
// Why groups: what a hand-over costs grows with the number of workgroups that take part in it -- all-to-all among 256
// workgroups 2.0 us, among 64 or fewer 1.5 us (tools/microbench/gather_bench.hip "g", profiles/r2_microbench_*) -- and a
// layer's arithmetic is as fast on 32-64 compute units as on 256 (8 waves x one row set each either way).  First version
// (every workgroup a slice of every layer, 8-byte granules, 256-way sweeps): 16.0 us per sample; 16-byte pair stores and
// loads: 11.2; groups: 10.2; one store instruction per workgroup and stage: 6.8.
// Off the chain, as in the launch loop: the input layer and the first cell's input half are hoisted GEMMs over all steps
// (cI, G), so the first cell is elementwise in the drawn sample; both cells' hidden halves W_hh . h only need the cell's
// own previous output: the cell's group sweeps its own array a second time, after publishing, and computes them in the
// shadow of the other groups' stages (these two arrays alternate between two buffers by step parity, so that a fast
// workgroup's next publication cannot overtake a slow one's second sweep).  Every C1 wave draws the sample itself from the
// gathered logits (no broadcast hand-over).  The draw follows the launch loop's definition (softmax in fp32, CDF in
// float64, MOL arithmetic of mol.h); in RAW mode only the order of the float64 additions differs.
#pragma once
#include "seam.h"

namespace {

constexpr int kWvR = 512;        // rnn_dims = fc_dims the kernel is built for
constexpr int kWvMaxA = 64;      // aux dims: one lane each
constexpr int kWvStampStep0 = 1000;

struct WvResParams {
  // weights, device, row-major fp32 as torch holds them
  const float *whh1 /*[3R][R]*/, *b1hn /*[R] b_hn of the first cell*/;
  const float *wih2 /*[3R][R + A]*/, *whh2 /*[3R][R]*/, *b2 /*[4][R]: b_ir + b_hr, b_iz + b_hz, b_in, b_hn*/;
  const float *fc1_w /*[R][R + A]*/, *fc1_b, *fc2_w /*[R][R + A]*/, *fc2_b, *fc3_w /*[C][R]*/, *fc3_b;
  const float *w_x /*[R]*/, *ug /*[3R] = W_ih1 . w_x*/;
  // per call (one row): hoisted input layer / first-cell input half, the four aux slices, noise
  const float *cI /*[seq][R]*/, *G /*[seq][3R]*/, *aux /*[seq][4A]*/, *u /*[seq] or [seq][nr_mix + 1]*/, *x_forced /*[seq] or null*/;
  float *out /*[seq]*/, *logits_out /*[seq][C] or null*/;
  u64 *g_lg, *g_s1 /*x2 buffers*/, *g_s2 /*x2 buffers*/, *g_y1, *g_y2;   // hand-over arrays of kWvArr granules (see the kernel)
  u32* err;
  u32* xcc;      // [256]: XCD number + 1 of every workgroup, written at the start of the launch
  int map;       // 1: groups by XCD (blocks are dealt to the XCDs in turn), 0: blocks in order
  int plain;     // plain stores inside the fc group where it is seen to share an XCD
  u64* stamps;   // diagnostic (WRNN_RESIDENT_STAMPS): [5 groups][8 steps][4 points] of each group's first workgroup, or null
  int seq, A, C;
};

// Sweep of a hand-over array: thread i < npairs waits for granule pair i (16 bytes at g + 2 i granules: both tags equal
// `epoch`) and stages its two values.  CELL: pair i = {x, h} of unit i -> dst[i], dst[512 + i]; otherwise rows 2 i, 2 i + 1
// -> dst[2 i], dst[2 i + 1].  Bounded like seam.h's gather; a wave that has given up once does not wait again.
template <bool CELL>
__device__ __forceinline__ void sweep_pairs(const u64* g, u32 epoch, int npairs, float* dst, u32* err, bool& gave_up) {
  using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
  const int i = threadIdx.x;
  const bool mine = i < npairs;
  if (gave_up || !__any(mine)) return;
  const u64* a = g + 2 * (mine ? i : 0);
  u32x4_ q;
  for (u32 spins = 0;; ++spins) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(a) : "memory");
    if (__all(!mine || (q.y == epoch && q.w == epoch))) break;
    if (!spin_again(spins, err, gave_up)) break;
  }
  if (mine) {
    if (CELL) dst[i] = __uint_as_float(q.x), dst[kWvR + i] = __uint_as_float(q.z);
    else dst[2 * i] = __uint_as_float(q.x), dst[2 * i + 1] = __uint_as_float(q.z);
  }
}

constexpr int kWvGrid = 256;   // C1: 64 workgroups (XCDs 0, 1), C2: 64 (XCDs 2, 3), fc1 + fc2 + fc3: 32 (XCD 4); the rest leave at once
constexpr size_t kWvArr = 64 * 16;   // granules per hand-over array (64 slots of 128 bytes; the fc arrays use 32 of them)

#define WSTAMP(slot_, i_)                                                                                          \
  do {                                                                                                             \
    if (p.stamps && stamp_wg && threadIdx.x == 0 && t >= kWvStampStep0 && t < kWvStampStep0 + 8)                   \
      p.stamps[((slot_) * 8 + (t - kWvStampStep0)) * 4 + (i_)] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)

// PER = C / 32 (RAW: the draw handles PER / 2 classes per lane), 0 = MOL
template <int PER>
__global__ __launch_bounds__(kRT) void wavernn_resident_kernel(const WvResParams p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kWvR];
  __shared__ float pub[16];   // the workgroup's 8 pairs of a stage, published by ONE store instruction (see below)
  // Block -> (group, workgroup of the group).  Blocks are dealt to the eight XCDs in turn: with map = 1 the fc group is the 32
  // blocks of one XCD, so fc1 -> fc2 -> fc3 hand over inside that XCD's L2 (round 3; before, each fc layer was a group of its own
  // and all five hand-overs crossed XCDs: 6.6-6.9 us per sample).  map = 0 (WRNN_RESIDENT_LINEAR_MAP) takes the blocks in order;
  // nothing but the time depends on the placement.
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 5;
  const int grp = p.map ? xcd : lin, gidx = p.map ? blockIdx.x >> 3 : blockIdx.x & 31;
  if (threadIdx.x == 0) {   // (every workgroup, also those that leave at once)
    unsigned my;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.xcc + blockIdx.x), "v"((my & 15u) + 1u) : "memory");
  }
  if (grp > 4) return;
  const int wg = grp < 4 ? (grp & 1) * 32 + gidx + (grp >= 2 ? 64 : 0) : 128 + gidx;   // C1 0..63, C2 64..127, fc 128..159
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int seq = p.seq, A = p.A, C = p.C, KA = kWvR + A;
  const int la = min(lane, A - 1);             // aux column of this lane (lanes >= A: weight 0)
  const float amask = lane < A ? 1.0f : 0.0f;
  bool gave_up = false;
  const auto row8 = [&](const float* r, float4& w0, float4& w1) __attribute__((always_inline)) {
    w0 = *reinterpret_cast<const float4*>(r + 4 * lane), w1 = *reinterpret_cast<const float4*>(r + 256 + 4 * lane);
  };
  const auto dot8 = [&](const float4& w0, const float4& w1, const float* v) __attribute__((always_inline)) {
    return dot4(w1, *reinterpret_cast<const float4*>(v + 256 + 4 * lane), dot4(w0, *reinterpret_cast<const float4*>(v + 4 * lane), 0.0f));
  };

  if (wg < 64) {
    // ================================================================ C1: draw + first GRU cell, unit 8 wg + wave
    const int unit = 8 * wg + wave;
    const bool stamp_wg = wg == 0;
    float* const v_lg = lds;        // [C] logits of the previous step, later reused: the group's own {x1 | h1}
    float4 wh[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) row8(p.whh1 + (size_t)(g * kWvR + unit) * kWvR, wh[g][0], wh[g][1]);
    const float b_hn = p.b1hn[unit], ug_r = p.ug[unit], ug_z = p.ug[kWvR + unit], ug_n = p.ug[2 * kWvR + unit], wxv = p.w_x[unit];
    const int nm = C / 3;   // MOL: mixtures
    float h_own = 0.0f, hs_r = 0.0f, hs_z = 0.0f, hs_n = b_hn;   // W_hh . 0 + [0, 0, b_hn]
    float g_r, g_z, g_n, ci_cur, u_cur, u_cur2 = 0.0f, xf_cur;
    // G / cI of step t_ and the noise / teacher-forced input of the sample drawn at the start of step t_ (the sample of
    // step t_ - 1; t_ = seq: the last sample, drawn after the loop); requested a step ahead, right after the publication
#define FETCH_C1(t_)                                                                      \
  do {                                                                                    \
    const int tc_ = min((t_), seq - 1), td_ = max(min((t_), seq) - 1, 0);                 \
    const float* Gt_ = p.G + (size_t)tc_ * 3 * kWvR + unit;                               \
    g_r = Gt_[0], g_z = Gt_[kWvR], g_n = Gt_[2 * kWvR];                                   \
    ci_cur = p.cI[(size_t)tc_ * kWvR + unit];                                             \
    if constexpr (PER == 0) {                                                             \
      u_cur = p.u[(size_t)td_ * (nm + 1) + min(lane & 31, nm - 1)];                       \
      u_cur2 = p.u[(size_t)td_ * (nm + 1) + nm];                                          \
    } else {                                                                              \
      u_cur = p.u[td_];                                                                   \
    }                                                                                     \
    xf_cur = p.x_forced ? p.x_forced[td_] : 0.0f;                                         \
  } while (0)
    FETCH_C1(0);
    // the sample of the logits in v_lg with the noise in u_cur / u_cur2 (every lane of the wave gets it)
    const auto draw = [&]() __attribute__((always_inline)) -> float {
      if constexpr (PER == 0) {
        const int ln = lane & 31, ci = min(ln, nm - 1);
        const float lp = v_lg[ci], mean = v_lg[nm + ci], ls = v_lg[2 * nm + ci];
        float sc = ln < nm ? wrnn_mol::gumbel_score(lp, u_cur) : -INFINITY;
        int k = ln;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {
          const float so = __shfl_xor(sc, off, 32);
          const int ko = __shfl_xor(k, off, 32);
          if (so > sc || (so == sc && ko < k)) sc = so, k = ko;
        }
        return wrnn_mol::logistic_sample(__shfl(mean, k, 32), __shfl(ls, k, 32), u_cur2, wrnn_mol::kDefaultLogScaleMin);
      } else {
        // RAW: class = number of inclusive-CDF values <= u * total, the CDF of p = softmax(logits) accumulated in float64
        // (as the sampler launch and the oracle do; only the summation order differs: lane by lane, then a DPP scan)
        constexpr int Q = PER / 2;   // consecutive classes per lane
        float l[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) l[i] = v_lg[lane * Q + i];
        float m = l[0];
#pragma unroll
        for (int i = 1; i < Q; ++i) m = fmaxf(m, l[i]);
        m = wave_max(m);
        double run = 0.0, cdf[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          run += (double)__expf(l[i] - m);
          cdf[i] = run;
        }
        const double incl = wave_scan_f64(run);
        const double base = incl - run, thr = (double)u_cur * readlane_f64(incl, 63);
        float cnt = 0.0f;
#pragma unroll
        for (int i = 0; i < Q; ++i) cnt += (base + cdf[i] <= thr) ? 1.0f : 0.0f;
        const int c = min((int)wave_sum(cnt), C - 1);   // small integers: exact in fp32
        return 2.0f * (float)c / (float)(C - 1) - 1.0f;
      }
    };
    for (int t = 0; t < seq; ++t) {
      WSTAMP(0, 0);
      float x = 0.0f;
      if (t > 0) {
        sweep_pairs<false>(p.g_lg, (u32)t, (C + 1) / 2, v_lg, p.err, gave_up);
        __syncthreads();
        WSTAMP(0, 1);
        const float sample = draw();
        x = p.x_forced ? xf_cur : sample;
        if (wg == 0 && tid == 0) p.out[t - 1] = sample;
        if (wg == 0 && p.logits_out)
          for (int c = tid; c < C; c += kRT) p.logits_out[(size_t)(t - 1) * C + c] = v_lg[c];
      }
      const float r = sigmoidf_(fmaf(ug_r, x, g_r) + hs_r);
      const float z = sigmoidf_(fmaf(ug_z, x, g_z) + hs_z);
      const float n = tanhf_(fmaf(ug_n, x, g_n) + r * hs_n);
      h_own = (1.0f - z) * n + z * h_own;
      u64* const gs = p.g_s1 + (t & 1) * kWvArr;
      if (lane == 0) pub[2 * wave] = fmaf(wxv, x, ci_cur) + h_own, pub[2 * wave + 1] = h_own;
      __syncthreads();
      if (tid < 8) st_gran2(gs + wg * 16 + 2 * tid, (u32)t + 1, pub[2 * tid], pub[2 * tid + 1]);
      WSTAMP(0, 2);
      FETCH_C1(t + 1);
      // hidden half of the next step: the whole new h1, i.e. the group's own array (off the chain)
      __syncthreads();   // every wave has drawn: v_lg may be overwritten
      sweep_pairs<true>(gs, (u32)t + 1, kWvR, lds, p.err, gave_up);
      __syncthreads();
      hs_r = dot8(wh[0][0], wh[0][1], lds + kWvR), hs_z = dot8(wh[1][0], wh[1][1], lds + kWvR), hs_n = dot8(wh[2][0], wh[2][1], lds + kWvR);
      {
        float unused = 0.0f;
        wave_sum4(hs_r, hs_z, hs_n, unused);
      }
      hs_n += b_hn;
      __syncthreads();   // the hidden half has been read: the next sweep may overwrite lds
    }
    if (wg == 0) {   // the last sample (no step follows to draw it)
      sweep_pairs<false>(p.g_lg, (u32)seq, (C + 1) / 2, v_lg, p.err, gave_up);
      __syncthreads();
      const float sample = draw();
      if (tid == 0) p.out[seq - 1] = sample;
      if (p.logits_out)
        for (int c = tid; c < C; c += kRT) p.logits_out[(size_t)(seq - 1) * C + c] = v_lg[c];
    }
#undef FETCH_C1
  } else if (wg < 128) {
    // ================================================================ C2: second GRU cell, unit 8 (wg - 64) + wave
    const int j = wg - 64, unit = 8 * j + wave;
    const bool stamp_wg = j == 0;
    float4 wi[3][2], wh[3][2];
    float wia[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float* ri = p.wih2 + (size_t)(g * kWvR + unit) * KA;
      row8(ri, wi[g][0], wi[g][1]);
      wia[g] = ri[kWvR + la] * amask;
      row8(p.whh2 + (size_t)(g * kWvR + unit) * kWvR, wh[g][0], wh[g][1]);
    }
    const float b_r = p.b2[unit], b_z = p.b2[kWvR + unit], b_in = p.b2[2 * kWvR + unit], b_hn = p.b2[3 * kWvR + unit];
    float h_own = 0.0f, hs_r = 0.0f, hs_z = 0.0f, hs_n = 0.0f;
    float a_cur = p.aux[A + la];
    for (int t = 0; t < seq; ++t) {
      const u32 ep = (u32)t + 1;
      WSTAMP(1, 0);
      sweep_pairs<true>(p.g_s1 + (t & 1) * kWvArr, ep, kWvR, lds, p.err, gave_up);
      __syncthreads();
      WSTAMP(1, 1);
      float s_r = fmaf(wia[0], a_cur, dot8(wi[0][0], wi[0][1], lds)), s_z = fmaf(wia[1], a_cur, dot8(wi[1][0], wi[1][1], lds));
      float s_n = fmaf(wia[2], a_cur, dot8(wi[2][0], wi[2][1], lds)), s_unused = 0.0f;
      wave_sum4(s_r, s_z, s_n, s_unused);   // three wave sums as one reduction (seam.h)
      const float r = sigmoidf_(s_r + hs_r + b_r);
      const float z = sigmoidf_(s_z + hs_z + b_z);
      const float n = tanhf_(s_n + b_in + r * (hs_n + b_hn));
      h_own = (1.0f - z) * n + z * h_own;
      u64* const gs = p.g_s2 + (t & 1) * kWvArr;
      if (lane == 0) pub[2 * wave] = lds[unit] + h_own, pub[2 * wave + 1] = h_own;
      __syncthreads();
      if (tid < 8) st_gran2(gs + j * 16 + 2 * tid, ep, pub[2 * tid], pub[2 * tid + 1]);
      WSTAMP(1, 2);
      a_cur = p.aux[(size_t)min(t + 1, seq - 1) * 4 * A + A + la];
      __syncthreads();   // x1 has been read
      sweep_pairs<true>(gs, ep, kWvR, lds, p.err, gave_up);
      __syncthreads();
      hs_r = dot8(wh[0][0], wh[0][1], lds + kWvR), hs_z = dot8(wh[1][0], wh[1][1], lds + kWvR), hs_n = dot8(wh[2][0], wh[2][1], lds + kWvR);
      {
        float unused = 0.0f;
        wave_sum4(hs_r, hs_z, hs_n, unused);
      }
      __syncthreads();
    }
  } else {
    // ================================================================ fc1, fc2, fc3 one after the other: rows 16 j + 2 wave, + 1
    const int j = wg - 128;
    const bool stamp_wg = j == 0;
    const int row0 = 16 * j + 2 * wave;
    float4 wa[3][2], wb[3][2];
    float waa[3], wba[3], ba[3], bb[3], a_cur[2];
    bool second[3];
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
      const int n_out = layer == 2 ? C : kWvR;
      const int ra = min(row0, n_out - 1), rb = min(row0 + 1, n_out - 1);
      const float* W = layer == 0 ? p.fc1_w : layer == 1 ? p.fc2_w : p.fc3_w;
      const float* bvec = layer == 0 ? p.fc1_b : layer == 1 ? p.fc2_b : p.fc3_b;
      const int ld = layer == 2 ? kWvR : KA;
      row8(W + (size_t)ra * ld, wa[layer][0], wa[layer][1]);
      row8(W + (size_t)rb * ld, wb[layer][0], wb[layer][1]);
      waa[layer] = layer == 2 ? 0.0f : W[(size_t)ra * ld + kWvR + la] * amask;
      wba[layer] = layer == 2 ? 0.0f : W[(size_t)rb * ld + kWvR + la] * amask;
      ba[layer] = bvec[ra], bb[layer] = bvec[rb];
      second[layer] = row0 + 1 < n_out;
    }
    a_cur[0] = p.aux[2 * A + la], a_cur[1] = p.aux[3 * A + la];   // aux slices a3 (fc1), a4 (fc2); fc3 has none
    // The two hand-overs that stay inside this group (fc1 -> fc2 -> fc3) use plain stores -- the value then lives in the L2 that
    // all of the group read through (sc1 loads) -- but only if the group's 32 workgroups are SEEN to share an XCD.
    __shared__ unsigned sh_same;
    bool plain = false;
    if (p.plain) {
      unsigned my;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
      my = (my & 15u) + 1u;
      if (wave == 0) {
        const unsigned* a = p.xcc + (p.map ? 4 + 8 * (lane & 31) : 128 + (lane & 31));
        unsigned v = 0;
        for (u32 spins = 0;; ++spins) {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
          if (__all(v != 0u) || gave_up) break;
          if (!spin_again(spins, p.err, gave_up)) break;
        }
        const bool same = __all(v == my) && !gave_up;
        if (lane == 0) sh_same = same ? 1u : 0u;
      }
      __syncthreads();
      plain = __builtin_amdgcn_readfirstlane(sh_same) != 0u;
    }
    for (int t = 0; t < seq; ++t) {
      const u32 ep = (u32)t + 1;
#pragma unroll
      for (int layer = 0; layer < 3; ++layer) {
        const u64* const gin = layer == 0 ? p.g_s2 + (t & 1) * kWvArr : layer == 1 ? p.g_y1 : p.g_y2;
        u64* const gout = (layer == 0 ? p.g_y1 : layer == 1 ? p.g_y2 : p.g_lg) + j * 16;
        WSTAMP(2 + layer, 0);
        if (layer == 0) sweep_pairs<true>(gin, ep, kWvR, lds, p.err, gave_up);
        else sweep_pairs<false>(gin, ep, kWvR / 2, lds, p.err, gave_up);
        __syncthreads();
        WSTAMP(2 + layer, 1);
        float va = dot8(wa[layer][0], wa[layer][1], lds), vb = dot8(wb[layer][0], wb[layer][1], lds);
        if (layer < 2) va = fmaf(waa[layer], a_cur[layer], va), vb = fmaf(wba[layer], a_cur[layer], vb);
        wave_sum2(va, vb);
        va += ba[layer], vb += bb[layer];
        if (layer < 2) va = fmaxf(va, 0.0f), vb = fmaxf(vb, 0.0f);
        if (!second[layer]) vb = 0.0f;
        if (lane == 0) pub[2 * wave] = va, pub[2 * wave + 1] = vb;
        __syncthreads();
        if (tid < 8 && 16 * j + 2 * tid < (layer == 2 ? C : kWvR)) {
          if (layer < 2 && plain) st_gran2_l2(gout + 2 * tid, ep, pub[2 * tid], pub[2 * tid + 1]);
          else st_gran2(gout + 2 * tid, ep, pub[2 * tid], pub[2 * tid + 1]);
        }
        WSTAMP(2 + layer, 2);
        __syncthreads();   // the vector and pub have been read: the next stage may overwrite them
      }
      a_cur[0] = p.aux[(size_t)min(t + 1, seq - 1) * 4 * A + 2 * A + la];
      a_cur[1] = p.aux[(size_t)min(t + 1, seq - 1) * 4 * A + 3 * A + la];
    }
  }
}
#undef WSTAMP

}  // namespace
