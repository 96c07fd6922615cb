// Fold-batched WaveRNN generation (2 .. 32 rows: the folds of an utterance, BASELINE configs[1]) as ONE persistent,
// weights-resident launch per call.  Included by wavernn.hip; replaces its five-launches-per-sample loop for these batch
// sizes.  Like the loop it replaces the model is build-defined (the vocoder is absent from the mounted reference tree, see
// wavernn.hip) and parity is unpinned except for the MOL sampler (mol.h).
//
// Rows are a real dense contraction here (20 folds x 512 hidden), so the matrix stages run on the matrix cores with the
// weights as register-resident A operands of v_mfma_f32_16x16x4_f32, taken straight from the launch loop's fragment-ordered
// layers (linjob.h Layer::wt: [tile][K chunk of 16][lane] float4).
//
// Decomposition.  The rows of a batch never meet, so each ROW TILE (16 rows) is a pipeline of its own: 128 workgroups x 512
// threads per row tile, a grid of 256 launched cooperatively, one per compute unit (the launch asks for more LDS than half a
// compute unit has).  A sample is a chain of five stages (C1 draw + first GRU cell -> C2 second cell -> F1 -> F2 -> F3 -> logits ->
// C1), active one after the other, so a workgroup serves several of them with different registers.  Blocks are dealt to the eight
// XCDs in turn, so the 32 blocks with the same blockIdx % 8 share an XCD and its L2: they form a GROUP (row tile = XCD / 4,
// group q = XCD % 4, workgroup j = blockIdx / 8 of the group), and the stages are placed so that hand-overs stay inside a group
// where they can:
//   every workgroup (q, j):  C2, tile 32 q + j = 4 units x (r, z, gi_n, gh_n) over [x1 ; a2] on the chain; the hidden half over h2
//                            after the publication (its partial sums stay in the accumulators until the next step)
//   q = 0   also F1 (fc1 over [x2 ; a3], relu) and F2 (fc2 over [y1 ; a4], relu), rows 16 j ..: y1 never leaves the group
//   q = 3   also F3 (fc3, tiles j < ceil(C / 16)) and C1, units 16 j ..: the logits never leave the group.  Workgroups 2 r and
//           2 r + 1 draw row r (one wave, the one-row kernel's draw: float64 CDF) and the group exchanges the 16 samples; the first
//           GRU cell is elementwise in x (its input half is hoisted, wavernn.hip)
//   q = 1, 2  also the first cell's hidden half W_hh1 . h1 (gate tiles r, z | n of units 16 j ..), handed back to q = 3 off the chain
// Across XCDs everything is published with write-through stores; inside a group with plain stores once the group has SEEN that
// it shares an XCD (HW_REG_XCC_ID of every workgroup, published at kernel entry).  A group that has just published does not poll
// the lines it wrote for the first 0.8 us (that alone delayed their arrival elsewhere from 0.65 to 1.9 us).
// Wave w of 8 contracts K chunks 4 w .. 4 w + 3 (and aux chunk w); the eight partial sums meet in LDS; wave 0 does the
// pointwise rest lane-locally (the four gates of a (unit, row) are the lane's four accumulator registers) and publishes.
// Measured at 20 rows, us per step: one stage per workgroup with both row tiles in each 15.9 (4.5 us of MFMA
// on the chain: a 16x16x4 fp32 MFMA holds its SIMD's matrix pipe for 32 cycles); row-tile pipelines, two stages per workgroup 11.5;
// groups by XCD 10.0; plain stores inside a group 9.9; the draw shared out over the C1 group 8.9 (launch loop: 21.0).
//
// Hand-over: activations live in "operand order" ([row tile][K chunk][lane = 16 kq + row % 16] float4 holding k = 16 chunk +
// 4 kq + 0..3: every load instruction of a wave is 1 KB contiguous and lands as the MFMA's B operand), two slots per array by
// step parity, and the data is the flag (bilstm_resident.h's tag form): every published word carries the tag (t / 2) & 1 in
// its mantissa's last bit, the arrays start as all-ones words, a consumer requests its block with sc1 loads and repeats the
// request until every word shows the step's tag.  No flag, no fence, no acknowledgement wait.  A slot is rewritten (step
// t + 2) only after every consumer is done with step t: the stages form one cycle, so a publication of step t + 1 depends on
// every workgroup of the row tile having consumed step t's inputs; the second reads of h1 and h2 (the hidden halves) precede
// their readers' own next publication.  The state a recurrence continues from IS the tagged value (at most one ulp from the
// cell's result); the logits carry the bit too.
// Polling: every wave sleeps before its first request at a wait, by an amount it adapts from the number of requests that came
// back stale there at the previous steps (aim: one or two), so that waiting workgroups do not load the memory side.
// Waits are bounded; a wave that gives up raises the error word, stops waiting (the launch drains) and the host reports it.
#pragma once

namespace {

constexpr int kWfT = 512;                                   // threads per workgroup
constexpr int kWfR = 512;                                   // rnn_dims = fc_dims the kernel is built for
constexpr int kWfKC = kWfR / 16;                            // K chunks of a 512-wide vector
constexpr int kWfPer = 128;                                 // workgroups per row tile
constexpr int kWfMaxRows = 32;
constexpr unsigned kWfSpin = 1u << 21;
constexpr int kWfStampStep0 = 1000;
constexpr size_t kWfLdsBytes = 96 * 1024;                   // dynamic LDS asked for (more than half a compute unit's)

constexpr size_t kWfArr512 = 2 * 2 * (size_t)kWfKC * 64 * 4;   // floats of a 512-wide hand-over array: 2 slots x 2 row tiles
constexpr size_t kWfTail = 512;   // words behind the arrays: the launch's error word at 0, the workgroups' XCD numbers at 64 ..
// the six 512-wide arrays, the hidden half's sums, the logits' array, the tail
inline size_t wf_array_floats(int C) { return 6 * kWfArr512 + 3 * kWfArr512 + 2 * 2 * (size_t)((C + 15) / 16) * 64 * 4 + kWfTail; }

struct WfParams {
  const float4 *w_g1h, *w_g2x, *w_g2h, *w_f1, *w_f2, *w_f3;   // Layer::wt
  const float *b_g1h /*[96][16]: b_hn on the n rows*/, *b_g2 /*[128][16]: b_r, b_z, b_in, b_hn per unit*/, *b_f1, *b_f2, *b_f3;
  const float *w_x /*[R]*/, *ug /*[3R]*/;
  // per call: hoisted input layer / first-cell input half / aux slices, step-major interleaved [seq][n4][B][4]
  const float *cI, *G, *aux;
  const float *u /*[B][seq] or [B][seq][nr_mix + 1]*/, *x_forced /*[B][seq] or null*/;
  float *out /*[B][seq]*/, *logits_out /*[B][seq][C] or null*/;
  f32x4 *X1, *H1, *X2, *H2, *Y1, *Y2, *LG;   // hand-over arrays [2 slots][MT][chunks][64]
  f32x4* HS;                                 // [2 slots][MT][3 gates x 32][64]: W_hh1 . h1 of every step (first cell's hidden half)
  unsigned* err;
  unsigned* xcc;   // [256]: XCD number + 1 of every workgroup, written at the start of the launch
  u64* xs;         // [2 slots][2 row tiles][16]: {drawn sample, step + 1} of every row (C1 group: a row is drawn by two workgroups)
  u64* stamps;   // diagnostics (WRNN_RESIDENT_STAMPS): [5 stages][8 steps][4 points] of each stage's first workgroup of row tile 0, or null
  int B, MT /*row tiles*/, seq, A, C, nf3 /*tiles of fc3 = chunks of LG*/, adapt /*sleep before polling*/, map /*1: groups by XCD*/, stamp_j /*which workgroup of a group leaves the stamps*/, cool /*publishers wait before they poll what they wrote*/, plain /*plain stores inside a group*/;
};

using wf_u32x4 = __attribute__((vector_size(16))) unsigned;

__device__ __forceinline__ f32x4 wf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {   // sc1: past this compute unit's L1
  const wf_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
  return f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
}
__device__ __forceinline__ void wf_store4_wt(f32x4* p, f32x4 v) {   // write-through 16-byte store
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wf_store4_l2(f32x4* p, f32x4 v) {   // plain 16-byte store: stays in this XCD's L2
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float wf_tag(float v, unsigned e) { return __uint_as_float((__float_as_uint(v) & ~1u) | e); }
__device__ __forceinline__ f32x4 wf_tag4(f32x4 v, unsigned e) { return f32x4{wf_tag(v[0], e), wf_tag(v[1], e), wf_tag(v[2], e), wf_tag(v[3], e)}; }
__device__ __forceinline__ float wf_pick(f32x4 v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; }

// The wave's part of an operand block: chunks c0 + g * cstep, g < nvalid (<= N), at byte offset `base` of the array (slot and
// row tile), requested until every word shows tag e; with EXTRA one more float4 at byte offset xoff of the block (the
// finishing lanes' residual input).  Bounded; adapts the wave's sleep before the first request at this wait (`nap`).
template <int N, bool EXTRA>
__device__ __forceinline__ void wf_get(__amdgpu_buffer_rsrc_t rs, unsigned base, int c0, int cstep, int nvalid, unsigned xoff, unsigned e,
                                       bool adapt, int& nap, unsigned* err, bool& gave_up, f32x4 (&v)[N], f32x4& xr) {
  const unsigned voff = (threadIdx.x & 63) * 16u;
  if (adapt && !gave_up)
    for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(16);
  int nfail = 0;
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int g = 0; g < N; ++g)
      if (g < nvalid) v[g] = wf_load(rs, voff, base + (unsigned)((c0 + g * cstep) << 10));
    if (EXTRA) xr = wf_load(rs, xoff, base);
    unsigned a = ~0u, o = 0u;
#pragma unroll
    for (int g = 0; g < N; ++g)
      if (g < nvalid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a &= __float_as_uint(v[g][i]), o |= __float_as_uint(v[g][i]);
      }
    if (EXTRA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a &= __float_as_uint(xr[i]), o |= __float_as_uint(xr[i]);
    }
    const bool ok = e ? (a & 1u) != 0u : (o & 1u) == 0u;
    if (__all(ok) || gave_up) break;
    ++nfail;
    if (!wrnn::keep_polling(spins, kWfSpin, err, gave_up)) break;
    __builtin_amdgcn_s_sleep(2);
  }
  if (nfail <= 1) nap = max(nap - 1, 0);
  else if (nfail >= 4) nap = min(nap + 1, 48);
}

// acc += W(wave's four chunks) . block
__device__ __forceinline__ void wf_contract(const float* wv, const f32x4 (&v)[4], f32x4& acc) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[4 * c + i], v[c][i], acc, 0, 0, 0);
}

#define WFSTAMP(slot_, i_)                                                                                         \
  do {                                                                                                             \
    if (p.stamps && m == 0 && j == p.stamp_j && threadIdx.x == 0 && t >= kWfStampStep0 && t < kWfStampStep0 + 8)           \
      p.stamps[((slot_) * 8 + (t - kWfStampStep0)) * 4 + (i_)] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)

// One workgroup's program.  PER = C / 32 classes per lane of the RAW draw, 0 = MOL; `role` = the workgroup's second stage (0 F1,
// 1 F2, 2 F3, 3 C1) as a template parameter: a stage's registers exist only in the workgroups that serve it (as one body with
// run-time roles the four stages' state added up to more than a lane's 256 registers).
template <int PER, int role>
__device__ __forceinline__ void wf_body(const WfParams& p, f32x4* wf_lds, float* xs, int m, int j, int gbase, int gstride) {
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int pp = role * 32 + j;               // the workgroup's C2 tile; j: its tile / unit group in the other stages
  const int B = p.B, seq = p.seq, A = p.A, C = p.C, MT = p.MT;
  const int NA = (A + 15) / 16;               // aux chunks (wave w < NA contracts aux chunk w)
  const int kq = l >> 4, r = l & 15;          // operand role: k quad, row of the tile; finishing role (wave 0): unit (quad), row
  const int bc = min(16 * m + r, B - 1);      // the lane's batch row (rows past B: row B - 1 again, finite and never reported)
  const bool adapt = p.adapt != 0;
  bool gave_up = false, plain = false;        // plain: see the XCD check in front of the step loop
  int nap_a = 0, nap_b = 0, nap_c = 0, nap_d = 0;   // sleeps before the C2 wait, the fc stages' waits, the logits' wait
  f32x4 unused_x = {0.f, 0.f, 0.f, 0.f};
  const unsigned blk = (unsigned)MT * kWfKC * 1024u;   // bytes of one slot of a 512-wide array
  const unsigned rt = (unsigned)m * kWfKC * 1024u;     // the row tile's block in a slot
  const size_t slot4 = (size_t)MT * kWfKC * 64, rt4 = (size_t)m * kWfKC * 64;   // the same in float4
  f32x4(*red)[64] = reinterpret_cast<f32x4(*)[64]>(wf_lds);   // [wave (x 3 gates)][lane]

  // aux operand of the wave's aux chunk for step t_, slice s_ (zero past A and for waves without an aux chunk)
  const auto aux_load = [&](int t_, int s_) __attribute__((always_inline)) -> f32x4 {
    const int k = 16 * w + 4 * kq;
    return w < NA && k < A ? *reinterpret_cast<const f32x4*>(p.aux + (((size_t)t_ * A + (size_t)s_ * (A / 4) + (k >> 2)) * B + bc) * 4)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  const auto load16 = [&](const float4* src, float* dst) __attribute__((always_inline)) {   // four consecutive chunks of a layer
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = src[(size_t)c * 64];
      dst[4 * c] = a.x, dst[4 * c + 1] = a.y, dst[4 * c + 2] = a.z, dst[4 * c + 3] = a.w;
    }
  };

  // ---------------------------------------------------------------- C2 (every workgroup): tile pp = units 4 pp .. 4 pp + 3
  const int KCX = kWfKC + NA;
  float wx[16], wxa[4], wh[16];
  load16(p.w_g2x + ((size_t)pp * KCX + 4 * w) * 64 + l, wx);
  load16(p.w_g2h + ((size_t)pp * kWfKC + 4 * w) * 64 + l, wh);
  {
    const float4 a = w < NA ? p.w_g2x[((size_t)pp * KCX + kWfKC + w) * 64 + l] : float4{0.f, 0.f, 0.f, 0.f};
    wxa[0] = a.x, wxa[1] = a.y, wxa[2] = a.z, wxa[3] = a.w;
  }
  const f32x4 bias2 = *reinterpret_cast<const f32x4*>(p.b_g2 + pp * 16 + kq * 4);
  // where the tile's four units of row r live in a row tile's block: chunk pp / 4, k quad pp % 4
  const unsigned own4 = (unsigned)((pp >> 2) * 64 + (pp & 3) * 16 + r);
  const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(p.X1, 0, 2u * blk, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h2 = __builtin_amdgcn_make_buffer_rsrc(p.H2, 0, 2u * blk, 0x00020000);
  float h2 = 0.f;
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};   // W_hh . 0
  f32x4 av2 = aux_load(0, 1);

  // ---------------------------------------------------------------- the second stage's registers
  const bool isF = role == 0 || (role == 3 && j < p.nf3);
  constexpr bool isC1 = role == 3;
  constexpr int NT = role == 1 ? 2 : role == 2 ? 1 : 0;   // helper groups: tiles of the first cell's hidden half (gates r, z | n)
  float wB[32], wBa[4] = {0.f, 0.f, 0.f, 0.f}, wC[16], wCa[4] = {0.f, 0.f, 0.f, 0.f};   // F1 / F3 / the helper's tiles; F2
  f32x4 biasB = {0.f, 0.f, 0.f, 0.f}, avB = biasB, biasC = biasB, avC = biasB;
  // C1: constants, state and the operands requested a step ahead
  f32x4 ugr = biasB, ugz = biasB, ugn = biasB, wxv = biasB, bhn = biasB, h4 = biasB, hs_r = biasB, hs_z = biasB, hs_n = biasB;
  f32x4 g_r = biasB, g_z = biasB, g_n = biasB, ci4 = biasB;
  float u_cur = 0.f, u_cur2 = 0.f, xf_cur = 0.f;
  const int n4 = 4 * j + kq;                 // C1, wave 0: the lane's unit quad
  const int jr = j >> 1;                     // C1 draw: workgroups 2 jr, 2 jr + 1 draw row 16 m + jr (wave 0), everybody reads all rows' x
  const bool drawer = 16 * m + jr < B;
  const int bd = min(16 * m + jr, B - 1), ln = l;
  const int nm = C / 3;                      // MOL: mixtures
#pragma unroll
  for (int i = 0; i < 32; ++i) wB[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) wC[i] = 0.f;
  if (isF) {
    const int na = role == 3 ? 0 : NA, kcx = kWfKC + na;
    const float4* W = role == 0 ? p.w_f1 : p.w_f3;
    load16(W + ((size_t)j * kcx + 4 * w) * 64 + l, wB);
    if (w < na) {
      const float4 a = W[((size_t)j * kcx + kWfKC + w) * 64 + l];
      wBa[0] = a.x, wBa[1] = a.y, wBa[2] = a.z, wBa[3] = a.w;
    }
    biasB = *reinterpret_cast<const f32x4*>((role == 0 ? p.b_f1 : p.b_f3) + j * 16 + kq * 4);
    if (role == 0) {
      avB = aux_load(0, 2);
      load16(p.w_f2 + ((size_t)j * kcx + 4 * w) * 64 + l, wC);
      if (w < na) {
        const float4 a = p.w_f2[((size_t)j * kcx + kWfKC + w) * 64 + l];
        wCa[0] = a.x, wCa[1] = a.y, wCa[2] = a.z, wCa[3] = a.w;
      }
      biasC = *reinterpret_cast<const f32x4*>(p.b_f2 + j * 16 + kq * 4);
      avC = aux_load(0, 3);
    }
  }
  if (NT > 0) {
#pragma unroll
    for (int g = 0; g < NT; ++g) load16(p.w_g1h + ((size_t)((role == 1 ? g : 2) * 32 + j) * kWfKC + 4 * w) * 64 + l, wB + 16 * g);
  }
  if (isC1) {
    ugr = *reinterpret_cast<const f32x4*>(p.ug + 4 * n4), ugz = *reinterpret_cast<const f32x4*>(p.ug + kWfR + 4 * n4);
    ugn = *reinterpret_cast<const f32x4*>(p.ug + 2 * kWfR + 4 * n4), wxv = *reinterpret_cast<const f32x4*>(p.w_x + 4 * n4);
    bhn = *reinterpret_cast<const f32x4*>(p.b_g1h + (64 + j) * 16 + 4 * kq);
    hs_n = bhn;   // W_hh . 0 + [0, 0, b_hn]
  }
  const auto fetch_g = [&](int t_) __attribute__((always_inline)) {   // hoisted operands of step t_ (wave 0's lanes)
    const int tc = min(t_, seq - 1);
    const float* Gt = p.G + (size_t)tc * 3 * kWfR * B;
    g_r = *reinterpret_cast<const f32x4*>(Gt + ((size_t)n4 * B + bc) * 4);
    g_z = *reinterpret_cast<const f32x4*>(Gt + ((size_t)(kWfR / 4 + n4) * B + bc) * 4);
    g_n = *reinterpret_cast<const f32x4*>(Gt + ((size_t)(2 * (kWfR / 4) + n4) * B + bc) * 4);
    ci4 = *reinterpret_cast<const f32x4*>(p.cI + (size_t)tc * kWfR * B + ((size_t)n4 * B + bc) * 4);
  };
  // the noise / teacher-forced value of sample s_ (drawn at the start of step s_ + 1), requested a step ahead
  const auto fetch_u = [&](int s_) __attribute__((always_inline)) {
    const int sc = min(max(s_, 0), seq - 1);
    if constexpr (PER == 0) {
      u_cur = p.u[((size_t)bd * seq + sc) * (nm + 1) + min(ln, nm - 1)];
      u_cur2 = p.u[((size_t)bd * seq + sc) * (nm + 1) + nm];
    } else {
      u_cur = p.u[(size_t)bd * seq + sc];
    }
    xf_cur = p.x_forced ? p.x_forced[(size_t)bd * seq + sc] : 0.f;
  };
  const unsigned lg_blk = (unsigned)MT * (unsigned)p.nf3 * 1024u, lg_rt = (unsigned)m * (unsigned)p.nf3 * 1024u;
  const __amdgpu_buffer_rsrc_t rs_lg = __builtin_amdgcn_make_buffer_rsrc(p.LG, 0, 2u * lg_blk, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h1 = __builtin_amdgcn_make_buffer_rsrc(p.H1, 0, 2u * blk, 0x00020000);
  const unsigned hs_blk = (unsigned)MT * 96u * 1024u, hs_rt = (unsigned)m * 96u * 1024u;
  const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc(p.HS, 0, 2u * hs_blk, 0x00020000);
  f32x4 hsq[3] = {biasB, biasB, biasB};      // C1, wave 0: the hidden half's sums of the previous step, requested ahead
  const auto hs_request = [&](int t_) __attribute__((always_inline)) {   // (not waited for here)
#pragma unroll
    for (int g = 0; g < 3; ++g) hsq[g] = wf_load(rs_hs, (unsigned)l * 16u, (unsigned)(t_ & 1) * hs_blk + hs_rt + (unsigned)((g * 32 + j) << 10));
  };
  // C1: sample s_ of the tile's rows from the logits of step s_.  The group's 32 workgroups share the work: workgroups 2 jr and
  // 2 jr + 1 draw row jr (wave 0, the one-row kernel's draw: the lane's consecutive classes straight from the row's piece of the
  // logits block into registers, float64 CDF scan over the wave) and publish {x, step} -- inside the group's XCD; then every
  // workgroup reads the 16 samples.  (Before: every workgroup staged the whole 32 KB block in LDS and drew all 16 rows, a
  // half-wave per row: 1.45 us; this: 0.7.)  `last`: the sample after the final step -- drawn and reported, not exchanged.
  constexpr int Q = PER == 0 ? 1 : PER / 2;           // RAW: consecutive classes per lane
  constexpr int NF = PER == 0 ? 3 : (Q + 3) / 4;      // float4 of the logits block per lane
  const auto draw_row = [&](int s_, bool last) __attribute__((always_inline)) {
    const u32 step_tag = (u32)s_ + 1u;
    if (w == 0) {
      int cls0[NF];
      f32x4 lv[NF];
      if (drawer) {
        // the float4 (chunk c, k quad kq, row jr) holds classes 16 c + 4 kq .. + 3 of the row
        if constexpr (PER == 0) cls0[0] = min(ln, nm - 1), cls0[1] = nm + min(ln, nm - 1), cls0[2] = 2 * nm + min(ln, nm - 1);
        else
#pragma unroll
          for (int i = 0; i < NF; ++i) cls0[i] = Q * ln + 4 * i;
        const unsigned e = (unsigned)(s_ >> 1) & 1u;
        if (adapt && !gave_up)
          for (int i = 0; i < nap_c; ++i) __builtin_amdgcn_s_sleep(16);
        int nfail = 0;
        for (unsigned spins = 0;; ++spins) {
          unsigned a = ~0u, o = 0u;
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            const int f = cls0[i] >> 2;   // float4 of the row: chunk f / 4, k quad f % 4
            lv[i] = wf_load(rs_lg, (unsigned)(((f >> 2) * 64 + (f & 3) * 16 + jr) * 16), (unsigned)(s_ & 1) * lg_blk + lg_rt);
          }
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) a &= __float_as_uint(lv[i][q]), o |= __float_as_uint(lv[i][q]);
          if (__all(e ? (a & 1u) != 0u : (o & 1u) == 0u) || gave_up) break;
          ++nfail;
          if (!wrnn::keep_polling(spins, kWfSpin, p.err, gave_up)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (nfail <= 1) nap_c = max(nap_c - 1, 0);
        else if (nfail >= 4) nap_c = min(nap_c + 1, 48);
        {
          const int t = s_ + 1;   // (stamp 3 of C1: the row's logits are there)
          WFSTAMP(0, 3);
        }
      }
      if (!last && s_ + 1 < seq) hs_request(s_);   // (every workgroup of the group) lands while the row is drawn
      if (drawer) {
        float sample;
        if constexpr (PER == 0) {
          const float lp = wf_pick(lv[0], cls0[0] & 3), mean = wf_pick(lv[1], cls0[1] & 3), ls = wf_pick(lv[2], cls0[2] & 3);
          float sc = ln < nm ? wrnn_mol::gumbel_score(lp, u_cur) : -INFINITY;
          int k = ln;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const float so = __shfl_xor(sc, off, 64);
            const int ko = __shfl_xor(k, off, 64);
            if (so > sc || (so == sc && ko < k)) sc = so, k = ko;
          }
          sample = wrnn_mol::logistic_sample(__shfl(mean, k, 64), __shfl(ls, k, 64), u_cur2, wrnn_mol::kDefaultLogScaleMin);
          if (p.logits_out && !(j & 1)) {
            const float* lgp = reinterpret_cast<const float*>(&lv[0]);
            if (ln < nm)
#pragma unroll
              for (int i = 0; i < 3; ++i) p.logits_out[((size_t)bd * seq + s_) * C + cls0[i]] = lgp[4 * i + (cls0[i] & 3)];
          }
        } else {
          // class = number of inclusive-CDF values <= u * total, the CDF of p = softmax(logits) accumulated in float64 (the
          // one-row kernel's arithmetic: lane by lane, then a DPP scan over the wave)
          float lg[Q];
          if constexpr (Q >= 4) {
#pragma unroll
            for (int i = 0; i < Q; ++i) lg[i] = lv[i >> 2][i & 3];
          } else {
#pragma unroll
            for (int i = 0; i < Q; ++i) lg[i] = wf_pick(lv[0], ((Q * ln) & 3) + i);
          }
          float mx = lg[0];
#pragma unroll
          for (int i = 1; i < Q; ++i) mx = fmaxf(mx, lg[i]);
          mx = wave_max(mx);
          double run = 0.0, cdf[Q];
#pragma unroll
          for (int i = 0; i < Q; ++i) {
            run += (double)__expf(lg[i] - mx);
            cdf[i] = run;
          }
          const double incl = wave_scan_f64(run);
          const double base = incl - run, thr = (double)u_cur * readlane_f64(incl, 63);
          float cnt = 0.0f;
#pragma unroll
          for (int i = 0; i < Q; ++i) cnt += (base + cdf[i] <= thr) ? 1.0f : 0.0f;
          const int c = min((int)wave_sum(cnt), C - 1);   // small integers: exact in fp32
          sample = 2.0f * (float)c / (float)(C - 1) - 1.0f;
          if (p.logits_out && !(j & 1)) {
#pragma unroll
            for (int i = 0; i < Q; ++i) p.logits_out[((size_t)bd * seq + s_) * C + Q * ln + i] = lg[i];
          }
        }
        if (ln == 0) {
          if (!(j & 1)) p.out[(size_t)bd * seq + s_] = sample;
          if (!last) {
            const float x = p.x_forced ? xf_cur : sample;
            u64* const dst = p.xs + ((size_t)(s_ & 1) * 2 + m) * 16 + jr;
            const unsigned lo = __float_as_uint(x);
            if (plain) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(dst), "v"(((u64)step_tag << 32) | lo) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst), "v"(((u64)step_tag << 32) | lo) : "memory");
          }
        }
      }
      if (!last) {   // every workgroup: the tile's samples
        const bool mine = l < 16 && 16 * m + l < B;
        const u64* a = p.xs + ((size_t)(s_ & 1) * 2 + m) * 16 + (l & 15);
        u64 v = 0;
        for (unsigned spins = 0;; ++spins) {
          asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
          if (__all(!mine || (u32)(v >> 32) == step_tag) || gave_up) break;
          if (!wrnn::keep_polling(spins, kWfSpin, p.err, gave_up)) break;
        }
        if (mine) xs[l] = __uint_as_float((u32)v);
      }
    }
    __syncthreads();
  };
  // an fc stage: tile j of layer `layer` (0 fc1, 1 fc2, 2 fc3) over the row tile's block of `in`, published into `outp` (kco
  // chunks per row tile)
  const auto fc_stage = [&](int t, int layer, const float* wv, const float* wva, const f32x4& bias, f32x4& av, int& nap, f32x4* in,
                            f32x4* outp, int kco) __attribute__((always_inline)) {
    const unsigned e = (unsigned)(t >> 1) & 1u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(in, 0, 2u * blk, 0x00020000);
    WFSTAMP(2 + layer, 0);
    f32x4 xv[4];
    wf_get<4, false>(rs_in, (unsigned)(t & 1) * blk + rt, 4 * w, 1, 4, 0u, e, adapt, nap, p.err, gave_up, xv, unused_x);
    WFSTAMP(2 + layer, 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    wf_contract(wv, xv, acc);
    if (layer < 2 && w < NA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wva[i], av[i], acc, 0, 0, 0);
    }
    red[w][l] = acc;
    __syncthreads();
    if (w == 0) {
      f32x4 g = red[0][l];
#pragma unroll
      for (int q = 1; q < 8; ++q) g += red[q][l];
      g += bias;
      if (layer < 2) g = f32x4{fmaxf(g[0], 0.f), fmaxf(g[1], 0.f), fmaxf(g[2], 0.f), fmaxf(g[3], 0.f)};
      f32x4* const dst = outp + (size_t)(t & 1) * MT * kco * 64 + ((size_t)m * kco + j) * 64 + l;
      if (layer != 1 && plain) wf_store4_l2(dst, wf_tag4(g, e));   // read by this group only (F1 -> F2, F3 -> C1)
      else wf_store4_wt(dst, wf_tag4(g, e));
    }
    WFSTAMP(2 + layer, 2);
    if (layer < 2) av = aux_load(min(t + 1, seq - 1), 2 + layer);
    __syncthreads();   // the sums have been read
  };

  if (tid < 16) xs[tid] = 0.f;
  if (isC1) fetch_g(0), fetch_u(0);
  // A hand-over that never leaves the group (F1 -> F2, F3 -> C1) may use plain stores -- the value then lives in the L2 that all
  // of the group read through (sc1 loads) -- but ONLY if the group's 32 workgroups really share an XCD: every workgroup has
  // published its XCD number (kernel entry), the group compares.  Otherwise, or with WRNN_FOLDS_WT, write-through stores.
  if ((role == 0 || role == 3) && p.plain) {
    unsigned my;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
    my = (my & 15u) + 1u;
    if (w == 0) {
      const unsigned* a = p.xcc + gbase + (l & 31) * gstride;
      unsigned v = 0;
      for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
        if (__all(v != 0u) || gave_up) break;
        if (!wrnn::keep_polling(spins, kWfSpin, p.err, gave_up)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      const bool same = __all(v == my) && !gave_up;
      if (l == 0) reinterpret_cast<unsigned*>(xs)[16] = same ? 1u : 0u;
    }
    __syncthreads();
    plain = __builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(xs)[16]) != 0u;
  }
  __syncthreads();
  for (int t = 0; t < seq; ++t) {
    const unsigned e = (unsigned)(t >> 1) & 1u;
    // ---- C1: draw x(t - 1), first cell, publish x1(t), h1(t)
    if (isC1) {
      WFSTAMP(0, 0);
      if (t > 0) draw_row(t - 1, false);
      WFSTAMP(0, 1);
      if (w == 0) {
        if (t > 0) {   // the hidden half W_hh . h1(t - 1) from the helper groups (requested after the previous step's fc3)
          const unsigned ep = (unsigned)((t - 1) >> 1) & 1u;
          for (unsigned spins = 0;; ++spins) {
            unsigned a = ~0u, o = 0u;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int i = 0; i < 4; ++i) a &= __float_as_uint(hsq[g][i]), o |= __float_as_uint(hsq[g][i]);
            const bool ok = ep ? (a & 1u) != 0u : (o & 1u) == 0u;
            if (__all(ok) || gave_up) break;
            if (!wrnn::keep_polling(spins, kWfSpin, p.err, gave_up)) break;
            __builtin_amdgcn_s_sleep(2);
            hs_request(t - 1);
          }
          hs_r = hsq[0], hs_z = hsq[1], hs_n = hsq[2] + bhn;
        }
        const float x = t > 0 ? xs[r] : 0.f;
        f32x4 x1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rr = sigmoidf_(fmaf(ugr[i], x, g_r[i]) + hs_r[i]);
          const float zz = sigmoidf_(fmaf(ugz[i], x, g_z[i]) + hs_z[i]);
          const float nn = tanhf_(fmaf(ugn[i], x, g_n[i]) + rr * hs_n[i]);
          h4[i] = wf_tag((1.0f - zz) * nn + zz * h4[i], e);
          x1[i] = wf_tag(fmaf(wxv[i], x, ci4[i]) + h4[i], e);
        }
        const size_t at = (size_t)(t & 1) * slot4 + rt4 + (size_t)j * 64 + l;
        wf_store4_wt(p.X1 + at, x1);
        wf_store4_wt(p.H1 + at, h4);
      }
      WFSTAMP(0, 2);
      fetch_g(t + 1);
      fetch_u(t);
    }
    // ---- C2: x half on top of the hidden half's sums, gates, publish x2(t), h2(t)
    f32x4 h1v[4];
    {
      if (role == 0) WFSTAMP(1, 0);
      f32x4 xv[4], xres;
      // The group that has just published x1 would poll the very lines it wrote, in its own XCD's L2, while they are on their
      // way through to the memory side: measured, that alone delays x1's arrival in the OTHER XCDs from 0.65 to 1.9 us (a
      // step: 11.0 -> 10.0 us).  It cannot arrive here in less than that time anyway.
      if (isC1 && p.cool) {
        __builtin_amdgcn_s_sleep(15);
        __builtin_amdgcn_s_sleep(15);
      }
      wf_get<4, true>(rs_x1, (unsigned)(t & 1) * blk + rt, 4 * w, 1, 4, own4 * 16u, e, adapt, nap_a, p.err, gave_up, xv, xres);
      if (role == 0) WFSTAMP(1, 1);
      if (NT > 0) {   // helper groups: h1(t) was published with x1(t); requested now, it lands behind the C2 work
#pragma unroll
        for (int c = 0; c < 4; ++c) h1v[c] = wf_load(rs_h1, (unsigned)l * 16u, (unsigned)(t & 1) * blk + rt + (unsigned)((4 * w + c) << 10));
      }
      wf_contract(wx, xv, acc2);
      if (w < NA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wxa[i], av2[i], acc2, 0, 0, 0);
      }
      red[w][l] = acc2;
      __syncthreads();
      if (w == 0) {
        f32x4 g = red[0][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) g += red[q][l];
        const float rr = sigmoidf_(g[0] + bias2[0]);
        const float zz = sigmoidf_(g[1] + bias2[1]);
        const float nn = tanhf_(g[2] + bias2[2] + rr * (g[3] + bias2[3]));
        h2 = wf_tag((1.0f - zz) * nn + zz * h2, e);
        const float xo = wf_tag(wf_pick(xres, kq) + h2, e);
        // the row's four units into one lane: 16-byte stores (a 4-byte write-through store is a fabric write of its own)
        const f32x4 h4v = {h2, __shfl(h2, r + 16), __shfl(h2, r + 32), __shfl(h2, r + 48)};
        const f32x4 x4v = {xo, __shfl(xo, r + 16), __shfl(xo, r + 32), __shfl(xo, r + 48)};
        if (kq == 0) {
          const size_t at = (size_t)(t & 1) * slot4 + rt4 + own4;
          wf_store4_wt(p.X2 + at, x4v);
          wf_store4_wt(p.H2 + at, h4v);
        }
      }
      if (role == 0) WFSTAMP(1, 2);
      av2 = aux_load(min(t + 1, seq - 1), 1);
      __syncthreads();   // the sums have been read
    }
    // ---- F1 right behind it (its input is what the C2 stage has just published), F2 behind F1: its input comes from this
    //      workgroup's own group, which shares an XCD (and its L2) when the blocks are dealt to the XCDs in turn
    if (role == 0) {
      fc_stage(t, 0, wB, wBa, biasB, avB, nap_b, p.X2, p.Y1, kWfKC);
      fc_stage(t, 1, wC, wCa, biasC, avC, nap_d, p.Y1, p.Y2, kWfKC);
    }
    // ---- off the chain: the hidden halves of the next step.  Helper groups first: their tiles of W_hh1 . h1(t) for the C1
    //      group, which needs them a good half step later
    if (NT > 0 && t + 1 < seq) {
      {   // (all of it there?  else ask again: rare)
        unsigned a = ~0u, o = 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) a &= __float_as_uint(h1v[c][i]), o |= __float_as_uint(h1v[c][i]);
        if (!__all(e ? (a & 1u) != 0u : (o & 1u) == 0u))
          wf_get<4, false>(rs_h1, (unsigned)(t & 1) * blk + rt, 4 * w, 1, 4, 0u, e, false, nap_a, p.err, gave_up, h1v, unused_x);
      }
#pragma unroll
      for (int g = 0; g < NT; ++g) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        wf_contract(wB + 16 * g, h1v, a);
        red[w * NT + g][l] = a;
      }
      __syncthreads();
      if (w < NT) {   // wave g finishes tile g
        f32x4 a = red[w][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) a += red[q * NT + w][l];
        const int tile = (role == 1 ? w : 2) * 32 + j;
        wf_store4_wt(p.HS + (size_t)(t & 1) * MT * 96 * 64 + ((size_t)m * 96 + tile) * 64 + l, wf_tag4(a, e));
      }
      __syncthreads();   // the sums have been read
    }
    if (t + 1 < seq) {
      f32x4 hv[4];
      if (p.cool > 1) {   // (the same for h2: published a moment ago, and not needed before the next step)
        __builtin_amdgcn_s_sleep(15);
        __builtin_amdgcn_s_sleep(15);
      }
      wf_get<4, false>(rs_h2, (unsigned)(t & 1) * blk + rt, 4 * w, 1, 4, 0u, e, false, nap_a, p.err, gave_up, hv, unused_x);
      acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
      wf_contract(wh, hv, acc2);
    }
    if (role == 0) WFSTAMP(1, 3);
    // ---- F3, in the C1 group: the logits' hand-over stays in the group's XCD
    if (isF && role == 3) fc_stage(t, 2, wB, wBa, biasB, avB, nap_b, p.Y2, p.LG, p.nf3);
  }
  if (isC1) draw_row(seq - 1, true);   // the last sample (no step follows to draw it)
}
#undef WFSTAMP

template <int PER>
__global__ __launch_bounds__(kWfT) void wavernn_folds_kernel(const WfParams p) {
  extern __shared__ f32x4 wf_lds[];          // the tile's logits (C1) / the waves' partial sums, one use after the other
  __shared__ float xs[17];                   // C1: the fed-back sample of the tile's rows; [16]: the group shares an XCD
  // Block -> (row tile m, group q of 32 workgroups, j).  Blocks are dealt to the eight XCDs in turn, so with map = 1 a group is
  // the 32 blocks of one XCD (q = XCD % 4, row tile = XCD / 4) and a hand-over inside a group stays in that XCD's L2; map = 0
  // (WRNN_FOLDS_LINEAR_MAP, for comparison) takes the blocks in order.  Nothing but the time depends on the placement.
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 5;
  const int m = p.map ? xcd >> 2 : lin >> 2, q = p.map ? xcd & 3 : lin & 3, j = p.map ? blockIdx.x >> 3 : blockIdx.x & 31;
  if (threadIdx.x == 0) {   // (every workgroup, also those that leave at once)
    unsigned my;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.xcc + blockIdx.x), "v"((my & 15u) + 1u) : "memory");
  }
  if (m >= p.MT) return;   // (the grid is always 256 workgroups)
  const int gbase = p.map ? m * 4 + q : (m * 4 + q) * 32, gstride = p.map ? 8 : 1;   // the group's blocks
  switch (q) {
    case 0: wf_body<PER, 0>(p, wf_lds, xs, m, j, gbase, gstride); break;   // C2 + F1 + F2
    case 1: wf_body<PER, 1>(p, wf_lds, xs, m, j, gbase, gstride); break;   // C2 + the first cell's hidden half, gates r and z
    case 2: wf_body<PER, 2>(p, wf_lds, xs, m, j, gbase, gstride); break;   // C2 + the first cell's hidden half, gate n
    default: wf_body<PER, 3>(p, wf_lds, xs, m, j, gbase, gstride); break;  // C2 + F3 + C1
  }
}

}  // namespace
