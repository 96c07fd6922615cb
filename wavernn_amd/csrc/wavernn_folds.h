// Fold-batched WaveRNN generation (2 .. 32 rows: the folds of an utterance, BASELINE configs[1]) as ONE persistent,
// weights-resident launch per call.  Included by wavernn.hip; replaces its five-launches-per-sample loop for these batch
// sizes.  Like the loop it replaces the model is build-defined (the vocoder is absent from the mounted reference tree, see
// wavernn.hip) and parity is unpinned except for the MOL sampler (mol.h).
//
// Rows are a real dense contraction here (20 folds x 512 hidden), so the matrix stages run on the matrix cores with the
// weights as register-resident A operands of v_mfma_f32_16x16x4_f32, taken straight from the launch loop's fragment-ordered
// layers (linjob.h Layer::wt: [tile][K chunk of 16][lane] float4).  Up to 192 workgroups x 512 threads, launched
// cooperatively, one per compute unit (the launch asks for more LDS than half a compute unit has), in five GROUPS, one per
// stage of the sample's dependent chain:
//   C1  32 workgroups x 16 units   draws x(t-1) of every row from the logits (a half-wave per row, the launch loop's
//                                  arithmetic), first GRU cell: elementwise in x (its input half is hoisted, wavernn.hip),
//                                  its hidden half W_hh . h1 (3 gate tiles x K = 512) computed AFTER the publication
//   C2  64 workgroups x 8 units    second GRU cell: two tiles of 4 units x (r, z, gi_n, gh_n) over [x1 ; a2] on the chain,
//                                  the hidden half over h2 after the publication (its sums stay in the accumulators)
//   F1 / F2  32 workgroups x 16 rows   fc1 over [x2 ; a3], fc2 over [y1 ; a4], relu
//   F3  ceil(C / 16) workgroups    fc3
// Wave w of 8 contracts K chunks 4 w .. 4 w + 3 (and aux chunk w) for all MT = ceil(B / 16) row tiles; the eight partial
// sums meet in LDS; a finishing wave per (tile, row tile) does the pointwise rest lane-locally (the four gates of a
// (unit, row) are the lane's four accumulator registers) and publishes.
//
// Hand-over: activations live in "operand order" ([row tile][K chunk][lane = 16 kq + row % 16] float4 holding k = 16 chunk +
// 4 kq + 0..3: every load instruction of a wave is 1 KB contiguous and lands as the MFMA's B operand), two slots per array by
// step parity, and the data is the flag (bilstm_resident.h's tag form): every published word carries the tag (t / 2) & 1 in
// its mantissa's last bit, the arrays start as all-ones words, a consumer requests its block with sc1 loads and repeats the
// request until every word shows the step's tag.  No flag, no fence, no acknowledgement wait.  A slot is rewritten (step
// t + 2) only after every consumer is done with step t: the groups form one cycle, so a publication of step t + 1 depends on
// every workgroup of every group having consumed step t's inputs; C1's and C2's own second reads (the hidden halves)
// precede their own next publication.  The state a recurrence continues from IS the tagged value (at most one ulp from the
// cell's result); the logits carry the bit too.
// Polling: a group has work for about a fifth of a step.  A wave that re-requests a 32-64 KB block all the time would load the
// memory side for nothing, so every wave sleeps before its first request of a step, by an amount it adapts from the number of
// requests that came back stale at the previous steps (aim: one or two).
// Waits are bounded; a wave that gives up raises the error word, stops waiting (the launch drains) and the host reports it.
#pragma once

namespace {

constexpr int kWfT = 512;                                   // threads per workgroup
constexpr int kWfR = 512;                                   // rnn_dims = fc_dims the kernel is built for
constexpr int kWfKC = kWfR / 16;                            // K chunks of a 512-wide vector
constexpr int kWfC1 = 32, kWfC2 = 64, kWfF = 32;            // workgroups per group (F3: tiles of fc3)
constexpr int kWfMaxRows = 32;
constexpr unsigned kWfSpin = 1u << 21;
constexpr int kWfStampStep0 = 1000;
constexpr size_t kWfLdsBytes = 96 * 1024;                   // dynamic LDS asked for (more than half a compute unit's)

constexpr size_t kWfArr512 = 2 * 2 * (size_t)kWfKC * 64 * 4;   // floats of a 512-wide hand-over array: 2 slots x 2 row tiles
// the six 512-wide arrays, the logits' array, 64 words for the launch's error word
inline size_t wf_array_floats(int C) { return 6 * kWfArr512 + 2 * 2 * (size_t)((C + 15) / 16) * 64 * 4 + 64; }

struct WfParams {
  const float4 *w_g1h, *w_g2x, *w_g2h, *w_f1, *w_f2, *w_f3;   // Layer::wt
  const float *b_g1h /*[96][16]: b_hn on the n rows*/, *b_g2 /*[128][16]: b_r, b_z, b_in, b_hn per unit*/, *b_f1, *b_f2, *b_f3;
  const float *w_x /*[R]*/, *ug /*[3R]*/;
  // per call: hoisted input layer / first-cell input half / aux slices, step-major interleaved [seq][n4][B][4]
  const float *cI, *G, *aux;
  const float *u /*[B][seq] or [B][seq][nr_mix + 1]*/, *x_forced /*[B][seq] or null*/;
  float *out /*[B][seq]*/, *logits_out /*[B][seq][C] or null*/;
  f32x4 *X1, *H1, *X2, *H2, *Y1, *Y2, *LG;   // hand-over arrays [2 slots][MT][chunks][64]
  unsigned* err;
  u64* stamps;   // diagnostics (WRNN_RESIDENT_STAMPS): [5 groups][8 steps][4 points] of each group's first workgroup, or null
  int B, seq, A, C, nf3 /*tiles of fc3 = chunks of LG*/, adapt /*sleep before polling*/;
};

using wf_u32x4 = __attribute__((vector_size(16))) unsigned;

__device__ __forceinline__ f32x4 wf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {   // sc1: past this compute unit's L1
  const wf_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16);
  return f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
}
__device__ __forceinline__ void wf_store4_wt(f32x4* p, f32x4 v) {   // write-through 16-byte store
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float wf_tag(float v, unsigned e) { return __uint_as_float((__float_as_uint(v) & ~1u) | e); }
__device__ __forceinline__ f32x4 wf_tag4(f32x4 v, unsigned e) { return f32x4{wf_tag(v[0], e), wf_tag(v[1], e), wf_tag(v[2], e), wf_tag(v[3], e)}; }
__device__ __forceinline__ float wf_pick(f32x4 v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; }

// The wave's part of an operand block of step parity `slot`: chunks c0 + g * cstep, g < nvalid (<= N), of each of the MT row
// tiles (KC chunks per row tile), requested until every word shows tag e; with EXTRA one more float4 at byte offset xoff of
// the slot (the finishing lanes' residual input).  Bounded; adapts the wave's sleep before the first request.
template <int MT, int N, bool EXTRA>
__device__ __forceinline__ void wf_get(__amdgpu_buffer_rsrc_t rs, unsigned slot_off, int KC, int c0, int cstep, int nvalid, unsigned xoff,
                                       unsigned e, bool adapt, int& nap, unsigned* err, bool& gave_up, f32x4 (&v)[N][MT], f32x4& xr) {
  const unsigned voff = (threadIdx.x & 63) * 16u;
  if (adapt && !gave_up)
    for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(32);
  int nfail = 0;
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int g = 0; g < N; ++g)
      if (g < nvalid) {
#pragma unroll
        for (int m = 0; m < MT; ++m) v[g][m] = wf_load(rs, voff, slot_off + (unsigned)((m * KC + c0 + g * cstep) << 10));
      }
    if (EXTRA) xr = wf_load(rs, xoff, slot_off);
    unsigned a = ~0u, o = 0u;
#pragma unroll
    for (int g = 0; g < N; ++g)
      if (g < nvalid) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i) a &= __float_as_uint(v[g][m][i]), o |= __float_as_uint(v[g][m][i]);
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
  else if (nfail >= 4) nap = min(nap + 1, 24);
}

#define WFSTAMP(slot_, i_)                                                                                         \
  do {                                                                                                             \
    if (p.stamps && stamp_wg && threadIdx.x == 0 && t >= kWfStampStep0 && t < kWfStampStep0 + 8)                   \
      p.stamps[((slot_) * 8 + (t - kWfStampStep0)) * 4 + (i_)] = __builtin_amdgcn_s_memrealtime();                 \
  } while (0)

// MT = ceil(B / 16) row tiles; PER = C / 32 classes per lane of the RAW draw, 0 = MOL
template <int MT, int PER>
__global__ __launch_bounds__(kWfT) void wavernn_folds_kernel(const WfParams p) {
  extern __shared__ f32x4 wf_lds[];          // C1: the rows' logits, later the waves' partial sums; others: partial sums
  __shared__ float xs[kWfMaxRows];           // C1: the fed-back sample of every row
  const int wg = blockIdx.x, tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int B = p.B, seq = p.seq, A = p.A, C = p.C;
  const int NA = (A + 15) / 16;               // aux chunks (wave w < NA contracts aux chunk w)
  const int kq = l >> 4, r = l & 15;          // operand role: k quad, row of the row tile; finishing role: unit quad / unit, row
  const bool adapt = p.adapt != 0;
  bool gave_up = false;
  int nap = 0;
  f32x4 unused_x = {0.f, 0.f, 0.f, 0.f};
  const unsigned blk = (unsigned)MT * kWfKC * 1024u;   // bytes of one slot of a 512-wide array
  f32x4(*red)[64] = reinterpret_cast<f32x4(*)[64]>(wf_lds);   // [wave * slots + slot][lane]

  // aux operand of the wave's aux chunk for step t_, slice s_ (zero past A and for waves without an aux chunk)
  const auto aux_load = [&](int t_, int s_, f32x4 (&av)[MT]) __attribute__((always_inline)) {
    const int k = 16 * w + 4 * kq;
    const bool have = w < NA && k < A;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int b = min(16 * m + r, B - 1);
      av[m] = have ? *reinterpret_cast<const f32x4*>(p.aux + (((size_t)t_ * A + (size_t)s_ * (A / 4) + (k >> 2)) * B + b) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  if (wg < kWfC1) {
    // ================================================================ C1: draw + first GRU cell, units 16 j .. 16 j + 15
    const int j = wg;
    const bool stamp_wg = j == 0;
    float wh[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 a = p.w_g1h[((size_t)(g * 32 + j) * kWfKC + 4 * w + c) * 64 + l];
        wh[g][4 * c] = a.x, wh[g][4 * c + 1] = a.y, wh[g][4 * c + 2] = a.z, wh[g][4 * c + 3] = a.w;
      }
    // finishing role: wave m < MT = row tile m, lane (u, r) = units 16 j + 4 u .. + 3 of row 16 m + r
    const bool fin = w < MT;
    const int row = 16 * w + r, bc = min(row, B - 1), n4 = 4 * j + kq;
    const f32x4 ugr = *reinterpret_cast<const f32x4*>(p.ug + 4 * n4), ugz = *reinterpret_cast<const f32x4*>(p.ug + kWfR + 4 * n4);
    const f32x4 ugn = *reinterpret_cast<const f32x4*>(p.ug + 2 * kWfR + 4 * n4), wxv = *reinterpret_cast<const f32x4*>(p.w_x + 4 * n4);
    const f32x4 bhn = *reinterpret_cast<const f32x4*>(p.b_g1h + (64 + j) * 16 + 4 * kq);
    f32x4 h4 = {0.f, 0.f, 0.f, 0.f}, hs_r = h4, hs_z = h4, hs_n = bhn;   // W_hh . 0 + [0, 0, b_hn]
    f32x4 g_r, g_z, g_n, ci4;
    const auto fetch_g = [&](int t_) __attribute__((always_inline)) {
      const int tc = min(t_, seq - 1);
      const float* Gt = p.G + (size_t)tc * 3 * kWfR * B;
      g_r = *reinterpret_cast<const f32x4*>(Gt + ((size_t)n4 * B + bc) * 4);
      g_z = *reinterpret_cast<const f32x4*>(Gt + ((size_t)(kWfR / 4 + n4) * B + bc) * 4);
      g_n = *reinterpret_cast<const f32x4*>(Gt + ((size_t)(2 * (kWfR / 4) + n4) * B + bc) * 4);
      ci4 = *reinterpret_cast<const f32x4*>(p.cI + (size_t)tc * kWfR * B + ((size_t)n4 * B + bc) * 4);
    };
    // draw role: half-wave hw = tid / 32 draws rows hw, hw + 16; lane ln
    const int hw = tid >> 5, ln = tid & 31;
    const int nm = C / 3;   // MOL: mixtures
    float u_cur[MT], u_cur2[MT], xf_cur[MT];
    // the noise / teacher-forced value of sample s_ (drawn at the start of step s_ + 1), requested a step ahead
    const auto fetch_u = [&](int s_) __attribute__((always_inline)) {
      const int sc = min(max(s_, 0), seq - 1);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int b = min(16 * m + hw, B - 1);
        if constexpr (PER == 0) {
          u_cur[m] = p.u[((size_t)b * seq + sc) * (nm + 1) + min(ln, nm - 1)];
          u_cur2[m] = p.u[((size_t)b * seq + sc) * (nm + 1) + nm];
        } else {
          u_cur[m] = p.u[(size_t)b * seq + sc];
          u_cur2[m] = 0.f;
        }
        xf_cur[m] = p.x_forced ? p.x_forced[(size_t)b * seq + sc] : 0.f;
      }
    };
    const int LS = p.nf3 * 4 + 1;             // float4 per staged row of logits (odd: rows land in different banks)
    constexpr int NLG = PER == 0 ? 1 : (2 * PER + 7) / 8;   // logit chunks per wave: w, w + 8, ..
    const int nlg = p.nf3 > w ? (p.nf3 - w + 7) / 8 : 0;
    const __amdgpu_buffer_rsrc_t rs_lg = __builtin_amdgcn_make_buffer_rsrc(p.LG, 0, 2u * MT * (unsigned)p.nf3 * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h1 = __builtin_amdgcn_make_buffer_rsrc(p.H1, 0, 2u * blk, 0x00020000);
    // sample s_ of every row from the logits of step s_: into xs (what is fed back) and, by workgroup 0, into out / logits_out
    const auto draw_all = [&](int s_) __attribute__((always_inline)) {
      f32x4 lv[NLG][MT];
      wf_get<MT, NLG, false>(rs_lg, (unsigned)(s_ & 1) * MT * (unsigned)p.nf3 * 1024u, p.nf3, w, 8, nlg, 0u, (unsigned)(s_ >> 1) & 1u, adapt,
                             nap, p.err, gave_up, lv, unused_x);
#pragma unroll
      for (int g = 0; g < NLG; ++g)
        if (g < nlg) {
#pragma unroll
          for (int m = 0; m < MT; ++m) wf_lds[(size_t)(16 * m + r) * LS + (w + 8 * g) * 4 + kq] = lv[g][m];
        }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        // (every half-wave draws, rows past B on row B - 1's logits: the DPP reductions see whole waves; nothing is stored for them)
        const int b = 16 * m + hw;
        const float* lrow = reinterpret_cast<const float*>(wf_lds + (size_t)min(b, B - 1) * LS);
        float sample;
        if constexpr (PER == 0) {
          const int ci = min(ln, nm - 1);
          const float lp = lrow[ci], mean = lrow[nm + ci], ls = lrow[2 * nm + ci];
          float sc = ln < nm ? wrnn_mol::gumbel_score(lp, u_cur[m]) : -INFINITY;
          int k = ln;
#pragma unroll
          for (int off = 16; off >= 1; off >>= 1) {
            const float so = __shfl_xor(sc, off, 32);
            const int ko = __shfl_xor(k, off, 32);
            if (so > sc || (so == sc && ko < k)) sc = so, k = ko;
          }
          sample = wrnn_mol::logistic_sample(__shfl(mean, k, 32), __shfl(ls, k, 32), u_cur2[m], wrnn_mol::kDefaultLogScaleMin);
        } else {
          float lg[PER];
          if constexpr (PER % 4 == 0) {
#pragma unroll
            for (int i = 0; i < PER; i += 4) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(lrow + ln * PER + i);
              lg[i] = q[0], lg[i + 1] = q[1], lg[i + 2] = q[2], lg[i + 3] = q[3];
            }
          } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) lg[i] = lrow[ln * PER + i];
          }
          const int c = raw_draw<PER>(lg, u_cur[m], ln, C);
          sample = 2.0f * (float)c / (float)(C - 1) - 1.0f;
        }
        if (ln == 0 && b < B) {
          xs[b] = p.x_forced ? xf_cur[m] : sample;
          if (wg == 0) p.out[(size_t)b * seq + s_] = sample;
        }
      }
      if (wg == 0 && p.logits_out)
        for (int i = tid; i < B * C; i += kWfT) {
          const int b = i / C, c = i - b * C;
          p.logits_out[((size_t)b * seq + s_) * C + c] = reinterpret_cast<const float*>(wf_lds + (size_t)b * LS)[c];
        }
      __syncthreads();
    };
    if (tid < kWfMaxRows) xs[tid] = 0.f;
    fetch_g(0);
    fetch_u(0);
    __syncthreads();
    for (int t = 0; t < seq; ++t) {
      WFSTAMP(0, 0);
      if (t > 0) draw_all(t - 1);
      WFSTAMP(0, 1);
      const unsigned e = (unsigned)(t >> 1) & 1u;
      if (fin) {
        const float x = t > 0 ? xs[row] : 0.f;
        f32x4 x1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rr = sigmoidf_(fmaf(ugr[i], x, g_r[i]) + hs_r[i]);
          const float zz = sigmoidf_(fmaf(ugz[i], x, g_z[i]) + hs_z[i]);
          const float nn = tanhf_(fmaf(ugn[i], x, g_n[i]) + rr * hs_n[i]);
          h4[i] = wf_tag((1.0f - zz) * nn + zz * h4[i], e);
          x1[i] = wf_tag(fmaf(wxv[i], x, ci4[i]) + h4[i], e);
        }
        const size_t at = (size_t)(t & 1) * MT * kWfKC * 64 + ((size_t)w * kWfKC + j) * 64 + l;
        wf_store4_wt(p.X1 + at, x1);
        wf_store4_wt(p.H1 + at, h4);
      }
      WFSTAMP(0, 2);
      fetch_g(t + 1);
      fetch_u(t);
      // hidden half of the next step: the whole new h1, i.e. the group's own array (off the chain)
      if (t + 1 < seq) {
        f32x4 hv[4][MT];
        wf_get<MT, 4, false>(rs_h1, (unsigned)(t & 1) * blk, kWfKC, 4 * w, 1, 4, 0u, e, false, nap, p.err, gave_up, hv, unused_x);
        f32x4 acc[3][MT];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
              for (int m = 0; m < MT; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[g][4 * c + i], hv[c][m][i], acc[g][m], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int m = 0; m < MT; ++m) red[w * (3 * MT) + g * MT + m][l] = acc[g][m];
        __syncthreads();
        if (fin) {
          hs_r = hs_z = f32x4{0.f, 0.f, 0.f, 0.f}, hs_n = bhn;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            hs_r += red[q * (3 * MT) + w][l], hs_z += red[q * (3 * MT) + MT + w][l], hs_n += red[q * (3 * MT) + 2 * MT + w][l];
          }
        }
        __syncthreads();   // the sums have been read: the next step's logits may overwrite them
      }
      WFSTAMP(0, 3);
    }
    if (wg == 0) draw_all(seq - 1);   // the last sample (no step follows to draw it)
  } else if (wg < kWfC1 + kWfC2) {
    // ================================================================ C2: second GRU cell, tiles 2 jj, 2 jj + 1 (4 units each)
    const int jj = wg - kWfC1;
    const bool stamp_wg = jj == 0;
    const int KCX = kWfKC + NA;
    float wx[2][16], wxa[2][4], wh[2][16];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int tile = 2 * jj + tn;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 a = p.w_g2x[((size_t)tile * KCX + 4 * w + c) * 64 + l], b = p.w_g2h[((size_t)tile * kWfKC + 4 * w + c) * 64 + l];
        wx[tn][4 * c] = a.x, wx[tn][4 * c + 1] = a.y, wx[tn][4 * c + 2] = a.z, wx[tn][4 * c + 3] = a.w;
        wh[tn][4 * c] = b.x, wh[tn][4 * c + 1] = b.y, wh[tn][4 * c + 2] = b.z, wh[tn][4 * c + 3] = b.w;
      }
      const float4 a = w < NA ? p.w_g2x[((size_t)tile * KCX + kWfKC + w) * 64 + l] : float4{0.f, 0.f, 0.f, 0.f};
      wxa[tn][0] = a.x, wxa[tn][1] = a.y, wxa[tn][2] = a.z, wxa[tn][3] = a.w;
    }
    // finishing role: wave w < 2 MT = (tile 2 jj + w / MT, row tile w % MT), lane (u, r) = unit u of the tile, row r
    const bool fin = w < 2 * MT;
    const int ft = fin ? w / MT : 0, fm = fin ? w % MT : 0, tile = 2 * jj + ft;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.b_g2 + tile * 16 + kq * 4);
    // where the tile's four units of row r live in a 512-wide block: chunk tile / 4, k quad tile % 4
    const unsigned own = (unsigned)(((fm * kWfKC + (tile >> 2)) * 64 + (tile & 3) * 16 + r) * 16);
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc(p.X1, 0, 2u * blk, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h2 = __builtin_amdgcn_make_buffer_rsrc(p.H2, 0, 2u * blk, 0x00020000);
    float h = 0.f;
    f32x4 acc[2][MT], av[MT];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[tn][m] = f32x4{0.f, 0.f, 0.f, 0.f};   // W_hh . 0
    aux_load(0, 1, av);
    for (int t = 0; t < seq; ++t) {
      const unsigned e = (unsigned)(t >> 1) & 1u;
      WFSTAMP(1, 0);
      f32x4 xv[4][MT], xres;
      wf_get<MT, 4, true>(rs_x1, (unsigned)(t & 1) * blk, kWfKC, 4 * w, 1, 4, own, e, adapt, nap, p.err, gave_up, xv, xres);
      WFSTAMP(1, 1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[tn][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[tn][4 * c + i], xv[c][m][i], acc[tn][m], 0, 0, 0);
      if (w < NA) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[tn][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxa[tn][i], av[m][i], acc[tn][m], 0, 0, 0);
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[w * (2 * MT) + tn * MT + m][l] = acc[tn][m];
      __syncthreads();
      if (fin) {
        f32x4 g = red[w][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) g += red[q * (2 * MT) + w][l];
        const float rr = sigmoidf_(g[0] + bias[0]);
        const float zz = sigmoidf_(g[1] + bias[1]);
        const float nn = tanhf_(g[2] + bias[2] + rr * (g[3] + bias[3]));
        h = wf_tag((1.0f - zz) * nn + zz * h, e);
        const float xo = wf_tag(wf_pick(xres, kq) + h, e);
        // the row's four units into one lane: 16-byte stores (a 4-byte write-through store is a fabric write of its own)
        const f32x4 h4 = {h, __shfl(h, r + 16), __shfl(h, r + 32), __shfl(h, r + 48)};
        const f32x4 x4 = {xo, __shfl(xo, r + 16), __shfl(xo, r + 32), __shfl(xo, r + 48)};
        if (kq == 0) {
          const size_t at = (size_t)(t & 1) * MT * kWfKC * 64 + (own >> 4);
          wf_store4_wt(p.X2 + at, x4);
          wf_store4_wt(p.H2 + at, h4);
        }
      }
      WFSTAMP(1, 2);
      aux_load(min(t + 1, seq - 1), 1, av);
      __syncthreads();   // the sums have been read
      // hidden half of the next step (off the chain): the group's own h2; its sums stay in the accumulators
      if (t + 1 < seq) {
        f32x4 hv[4][MT];
        wf_get<MT, 4, false>(rs_h2, (unsigned)(t & 1) * blk, kWfKC, 4 * w, 1, 4, 0u, e, false, nap, p.err, gave_up, hv, unused_x);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[tn][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
              for (int m = 0; m < MT; ++m) acc[tn][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wh[tn][4 * c + i], hv[c][m][i], acc[tn][m], 0, 0, 0);
      }
      WFSTAMP(1, 3);
    }
  } else {
    // ================================================================ F1 / F2 / F3: tile j = rows 16 j .. 16 j + 15
    const int o = wg - (kWfC1 + kWfC2);
    const int layer = o < kWfF ? 0 : o < 2 * kWfF ? 1 : 2, j = o - layer * kWfF;
    const bool stamp_wg = j == 0;
    const int na = layer == 2 ? 0 : NA, KCX = kWfKC + na;
    const float4* W = layer == 0 ? p.w_f1 : layer == 1 ? p.w_f2 : p.w_f3;
    const float* bvec = layer == 0 ? p.b_f1 : layer == 1 ? p.b_f2 : p.b_f3;
    float wv[16], wa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 a = W[((size_t)j * KCX + 4 * w + c) * 64 + l];
      wv[4 * c] = a.x, wv[4 * c + 1] = a.y, wv[4 * c + 2] = a.z, wv[4 * c + 3] = a.w;
    }
    {
      const float4 a = w < na ? W[((size_t)j * KCX + kWfKC + w) * 64 + l] : float4{0.f, 0.f, 0.f, 0.f};
      wa[0] = a.x, wa[1] = a.y, wa[2] = a.z, wa[3] = a.w;
    }
    // finishing role: wave m < MT = row tile m, lane (u, r) = rows 16 j + 4 u .. + 3 of the layer, row r of the batch tile
    const bool fin = w < MT;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bvec + j * 16 + kq * 4);
    const int KCO = layer == 2 ? p.nf3 : kWfKC;   // chunks per row tile of the output array
    f32x4* const outp = layer == 0 ? p.Y1 : layer == 1 ? p.Y2 : p.LG;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(layer == 0 ? p.X2 : layer == 1 ? p.Y1 : p.Y2, 0, 2u * blk, 0x00020000);
    f32x4 av[MT];
    aux_load(0, 2 + layer, av);
    for (int t = 0; t < seq; ++t) {
      const unsigned e = (unsigned)(t >> 1) & 1u;
      WFSTAMP(2 + layer, 0);
      f32x4 xv[4][MT];
      wf_get<MT, 4, false>(rs_in, (unsigned)(t & 1) * blk, kWfKC, 4 * w, 1, 4, 0u, e, adapt, nap, p.err, gave_up, xv, unused_x);
      WFSTAMP(2 + layer, 1);
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[4 * c + i], xv[c][m][i], acc[m], 0, 0, 0);
      if (w < na) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], av[m][i], acc[m], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) red[w * MT + m][l] = acc[m];
      __syncthreads();
      if (fin) {
        f32x4 g = red[w][l];
#pragma unroll
        for (int q = 1; q < 8; ++q) g += red[q * MT + w][l];
        g += bias;
        if (layer < 2) g = f32x4{fmaxf(g[0], 0.f), fmaxf(g[1], 0.f), fmaxf(g[2], 0.f), fmaxf(g[3], 0.f)};
        wf_store4_wt(outp + (size_t)(t & 1) * MT * KCO * 64 + ((size_t)w * KCO + j) * 64 + l, wf_tag4(g, e));
      }
      WFSTAMP(2 + layer, 2);
      if (layer < 2) aux_load(min(t + 1, seq - 1), 2 + layer, av);
      __syncthreads();   // the sums have been read
    }
  }
}
#undef WFSTAMP

}  // namespace
