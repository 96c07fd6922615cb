// Free-running decode for SMALL batches (B <= 4) as ONE persistent, weights-resident launch per call.
// Included by taco_decoder.hip (after struct wrnn_decoder); replaces the 7-launches-per-step loop of
// wrnn_decoder_run_free where a step is pure latency (reference loop being replaced: models/tacotron.py:185-235
// driven as :339-345 drives it, with the build-defined feedback rule of SURVEY.md section 8b).
//
// 224 working workgroups x 512 threads (a grid of 256, one per compute unit, launched cooperatively, all co-resident), one GROUP per
// stage of the step's dependent chain.  A workgroup keeps ITS rows of ITS layers in registers for the whole utterance
// (20.5 MB of fp32 weights over the chip); nothing is fetched from memory inside the step loop except the activations
// handed from group to group (and per-step noise / aligner terms):
//   P1  16 workgroups x 16 rows      first prenet layer, folded through the mel projection of the previous step's last frame
//   PQ  16 workgroups                second prenet layer (8 rows each); later in the step the query projection (16 rows each)
//   GR  32 workgroups x 8 units      attention GRU cell -- the SAME 32 workgroups as P1 + PQ (round 3): the 32 blocks of one XCD
//                                    (blocks are dealt to the eight XCDs in turn), so P1 -> PQ -> GR -> PQ hand over inside that
//                                    XCD's L2, p1 and p with plain stores once the group has seen that it shares an XCD
//                                    (9.6 -> 9.2 us per step at B = 1, 12.2 -> 11.6 at B = 2, 19.6 -> 18.6 at B = 4)
//   AT  64 workgroups x POS positions  attention energies (location conv + projections + tanh + v-dot)         XCDs 1, 2
//   L1  64 workgroups x 8 units      softmax + first LSTM cell (rnn_input folded in) + residual                XCDs 3, 4
//   L2  64 workgroups x 8 units      second LSTM cell; off the chain: the mel frames of the step and the stop vote   XCDs 5, 6
// A wave owns whole rows / units: read the gathered vector from LDS, FMAs, DPP wave sums, the pointwise rest in the same
// wave; the eight waves' results meet in LDS and leave as ONE store instruction per workgroup and stage.
//
// Hand-over: a produced vector is an array of 16-byte granule PAIRS {value, tag, value, tag} (tag = step + 1, arrays
// zeroed before the launch; flat order element-major, batch row minor), written with write-through (sc1) stores and swept
// by the consuming group with 16-byte sc1 loads until every tag matches -- the data is the flag, no counter, no fence
// (CDNA4 guide, Guideline 16 form R2).  Every array alternates between two buffers by step parity, so that a producer
// can never overwrite what a slow consumer of the step before last is still waiting for (the chain is a cycle through all
// groups: whatever is published at step k + 2 depends on every workgroup's step k + 1, hence on its having finished
// with step k).  Measured prices (tools/microbench/gather_bench.hip, profiles/r2_microbench_*): an all-to-all hand-over
// among 256 workgroups 2.0 us, among 64 or fewer 1.5 us, both including 0.5 us of the benchmark's own arithmetic; eight
// separate stores into one line instead of one store instruction: +0.8 us.  The first version of this kernel (every
// workgroup a slice of every layer, 8-byte granules, all 256 workgroups in every sweep) took 13.5 us per step at B = 1,
// 19.2 at B = 2 and 43 at B = 4 (register spills); this one 9.6 / 12.2 / 19.6 (launch loop: 36).  Starting the off-chain
// sweeps a microsecond late, so that they do not poll next to the on-chain consumer, changes nothing.
//
// The step's dependent chain is 7 hand-overs, cut to that by algebra:
//  * the context vector is never formed.  Every consumer of ctx = sum_j s_j enc_q_j is linear in it, so
//    W . ctx = E . s with E = W . enc_q^T computed once per utterance (per wave, for its own rows, at kernel start, kept in
//    LDS): the attention GRU, the rnn_input projection and LSTM1 contract over the T attention weights instead of the
//    256 context dims, and the attention stage ends at the energies (every consumer normalises them itself);
//  * rnn_input is folded into LSTM1 (as in the launch loop) and x1 = y + h1 is formed by the wave that owns the unit; fc1
//    is folded through the mel projection of the step's last frame, the mel frames themselves are projected off the chain;
//  * whatever depends on a group's OWN previous output only (hidden halves, location features, attention state) is
//    computed after the publication, in the shadow of the other groups' stages.
//  per step:  [x1, h2] P1 -> [p1] PQ -> [p] GR -> [ah] PQ -> [pq] AT -> [u] L1 -> [x1] L2 -> [h2] next step
// Stop rule: L2 publishes one vote per workgroup and step (any mel value of the step >= threshold); every workgroup reads the
// votes of step k - 2 after its own work of step k and leaves the loop when that step was silent -- all of them on the
// same data, hence at the same step; the steps decoded in excess are not reported.
#pragma once
#include "seam.h"

namespace {

constexpr int kRL = 512;         // lstm_dims the resident kernel is built for
constexpr int kRMaxB = 4;
constexpr int kRMaxBT = 1024;    // B * T (attention state lives in LDS; 64 AT workgroups x 16 positions)
constexpr int kCap = 2048;       // granules per hand-over buffer (the largest vector: 512 x 4 rows)
enum { kA_p1, kA_p, kA_ah, kA_pq, kA_u, kA_x1, kA_h1, kA_h2, kA_vote, kA_count };   // arrays: 4 buffers of kCap each (2 used, votes 4)
constexpr int kStampStep0 = 200;

struct ResidentParams {
  // weights, device, row-major fp32 as torch holds them (folded ones built on the host)
  const float *fc1m /*[256][512] fc1 o mel_proj(last frame)*/, *fc1_b /*[256]*/, *fc2_w /*[128][256]*/, *fc2_b;
  const float *gru_wih /*[768][384]*/, *gru_whh /*[768][256]*/, *gru_bih, *gru_bhh;
  const float *pq_w /*[256][256]*/, *pq_b;
  const float *conv_t /*[62][32]*/, *L_w /*[256][32]*/, *L_b, *v_w;
  const float *l1_wf /*[2048][512] W_ih1 . W_rin, columns ctx | attn_h*/, *l1_whh /*[2048][512]*/, *l1_b /*[2048] folded*/;
  const float *rin_w /*[512][512]*/, *rin_b;
  const float *l2_wih, *l2_whh /*[2048][512]*/, *l2_b /*[2048] b_ih + b_hh*/;
  const float* mel_w /*[nmr][512], row j = mel j / r, frame j % r*/;
  // inputs
  const float *enc_q, *enc_p /*[B][T][256]*/;
  const float* cadd /*[n][64][B][4] interleaved: fc1's aligner-feature part + bias of step k+1, or null*/;
  const float *drop_u1 /*[n][B][256]*/, *drop_u2 /*[n][B][128]*/;
  float *mel_out, *att_out;
  int* n_done;
  u32* err;
  u32* xcc;      // [256]: XCD number + 1 of every workgroup, written at the start of the launch
  int map;       // 1: groups by XCD (blocks are dealt to the XCDs in turn), 0: blocks in order
  int plain;     // plain stores inside the P1 / PQ / GR group where it is seen to share an XCD
  u64* arrays;   // kA_count arrays x 4 buffers x kCap granules
  int B, T, n, r, nmr;
  long mel_sb, mel_sm;
  float thr;
  int use_stop;
  // diagnostic (WRNN_RESIDENT_STAMPS=<file>): 100 MHz timestamps of each group's first workgroup at 3 points (step start,
  // on-chain input swept, output published) of steps kStampStep0 .. +7, [6][8][4] u64; null otherwise
  u64* stamps;
};

__device__ __forceinline__ u64* buf_of(const ResidentParams& p, int array, int which) { return p.arrays + ((size_t)array * 4 + which) * kCap; }

// Sweep of a hand-over buffer: waits until the `npairs` granule pairs at g carry `epoch`, then stages the values: flat value
// f = 2 i (+1) of pair i is element f / NBV of batch row f % NBV -> dst[(f % NBV) * N + f / NBV].  (NBV = 1: dst[f].)
// PER pairs per thread, all polled in one loop.  Bounded: a sweep that never completes raises the error word and falls
// through; a wave that has given up once does not wait again (the call has failed; the launch drains).
template <int NBV, int PER>
__device__ __forceinline__ void sweep(const u64* g, u32 epoch, int npairs, int N, float* __restrict__ dst, u32* err, bool& gave_up) {
  using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
  const int tid = threadIdx.x;
  bool mine[PER], any = false;
#pragma unroll
  for (int i = 0; i < PER; ++i) mine[i] = tid + kRT * i < npairs, any |= mine[i];
  if (gave_up || !__any(any)) return;
  u32x4_ q[PER];
  for (u32 spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (mine[i]) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[i]) : "v"(g + 2 * (tid + kRT * i)) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      asm volatile("" : "+v"(q[i]));   // the values are defined only after the wait above
      if (mine[i]) ok &= q[i].y == epoch && q[i].w == epoch;
    }
    if (__all(ok)) break;
    if (!spin_again(spins, err, gave_up)) break;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (mine[i]) {
      const int f = 2 * (tid + kRT * i);
      if (NBV == 1) {
        dst[f] = __uint_as_float(q[i].x), dst[f + 1] = __uint_as_float(q[i].z);
      } else {   // f even and NBV even: both values belong to the same element
        float* d = dst + (f % NBV) * N + f / NBV;
        d[0] = __uint_as_float(q[i].x), d[N] = __uint_as_float(q[i].z);
      }
    }
}

// Sweep of the energies array: 64 slots (one per AT workgroup, at least 64 bytes apart) of POS / 2 pairs; pair i of the
// flat position order sits at granule (i / (POS / 2)) * max(POS, 8) + 2 (i % (POS / 2)) and is staged at dst[2 i], dst[2 i + 1].
template <int POS, int PER>
__device__ __forceinline__ void sweep_u(const u64* g, u32 epoch, float* __restrict__ dst, u32* err, bool& gave_up) {
  using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
  constexpr int HP = POS / 2, SLOT = POS < 8 ? 8 : POS, npairs = 64 * HP;
  const int tid = threadIdx.x;
  bool mine[PER], any = false;
#pragma unroll
  for (int i = 0; i < PER; ++i) mine[i] = tid + kRT * i < npairs, any |= mine[i];
  if (gave_up || !__any(any)) return;
  u32x4_ q[PER];
  for (u32 spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int pi = tid + kRT * i;
      if (mine[i]) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[i]) : "v"(g + (pi / HP) * SLOT + 2 * (pi % HP)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      asm volatile("" : "+v"(q[i]));
      if (mine[i]) ok &= q[i].y == epoch && q[i].w == epoch;
    }
    if (__all(ok)) break;
    if (!spin_again(spins, err, gave_up)) break;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (mine[i]) dst[2 * (tid + kRT * i)] = __uint_as_float(q[i].x), dst[2 * (tid + kRT * i) + 1] = __uint_as_float(q[i].z);
}

// Publication of a workgroup's stage output: pub[0 .. 2 npairs) in flat order, written by the waves before the call; ONE
// store instruction (lanes 0 .. npairs-1 of wave 0, 16 bytes each) puts it at granule `slot` of the buffer.
// (plain: a plain store -- the value stays in this XCD's L2 -- for arrays whose readers are seen to share the writer's XCD)
__device__ __forceinline__ void publish(u64* slot, u32 epoch, const float* pub, int npairs, bool plain = false) {
  __syncthreads();
  if ((int)threadIdx.x < npairs) {
    if (plain) st_gran2_l2(slot + 2 * threadIdx.x, epoch, pub[2 * threadIdx.x], pub[2 * threadIdx.x + 1]);
    else st_gran2(slot + 2 * threadIdx.x, epoch, pub[2 * threadIdx.x], pub[2 * threadIdx.x + 1]);
  }
}

// dynamic LDS of the kernel in floats: the largest group's carve (L1: E rows of 8 units x (4 gates + y) over B x Tp positions)
inline size_t resident_lds_floats(int NB, int B, int T) {
  const size_t Tp = (size_t)((T + 63) & ~63);
  return 40 * (size_t)B * Tp + (size_t)NB * (256 + 512) + kRMaxBT + 64 + 2 * 8 * (size_t)NB + 128 + 2 * (size_t)NB * Tp;
}

// NB: batch rows the register arrays and the hand-over layout are sized for (B <= NB, a power of two);
// POS: attention positions per AT workgroup (even; 64 POS >= B T)
template <int NB, int POS>
__global__ __launch_bounds__(kRT) void resident_free_kernel(const ResidentParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // Block -> (group, workgroup of the group).  Blocks are dealt to the eight XCDs in turn, so with map = 1 "XCD x" = the 32
  // blocks with blockIdx % 8 == x: XCD 0 the P1 / PQ / GR group (three of the step's seven hand-overs stay in its L2), XCDs 1-2
  // AT, 3-4 L1, 5-6 L2, the blocks of XCD 7 leave at once.  map = 0 (WRNN_RESIDENT_LINEAR_MAP): the same groups, blocks in order.
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int grp8 = p.map ? (int)(blockIdx.x & 7) : (int)(blockIdx.x >> 5), gidx = p.map ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x & 31);
  if (tid == 0) {   // (every workgroup, also those that leave at once)
    unsigned my;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.xcc + blockIdx.x), "v"((my & 15u) + 1u) : "memory");
  }
  if (grp8 == 7) return;
  const int B = p.B, T = p.T, n = p.n;
  const int Tp = (T + 63) & ~63;   // padded attention length: E is zero beyond T
  bool gave_up = false;
  int done = n;
  constexpr int PER512 = (512 * NB / 2 + kRT - 1) / kRT, PER256 = (256 * NB / 2 + kRT - 1) / kRT, PER128 = (128 * NB / 2 + kRT - 1) / kRT;
  constexpr int PERU = (64 * POS / 2 + kRT - 1) / kRT;
  constexpr int USLOT = POS < 8 ? 8 : POS;   // granules between two AT workgroups' slots of the energies array: at least a
                                             // 64-byte line each (four writers sharing a line cost 0.8 us per hand-over)
  // stage of the other groups' workgroups: 3 AT, 4 L1, 5 L2 (0 .. 2: the P1 / PQ / GR group); their index in the group
  const int group = grp8 == 0 ? 0 : 3 + (grp8 - 1) / 2, gw = ((grp8 - 1) & 1) * 32 + gidx;
#define RSTAMPG(g_, ok_, i_)                                                                                       \
  do {                                                                                                             \
    if (p.stamps && (ok_) && tid == 0 && k >= kStampStep0 && k < kStampStep0 + 8)                                  \
      p.stamps[(((g_) * 8) + (k - kStampStep0)) * 4 + (i_)] = __builtin_amdgcn_s_memrealtime();                    \
  } while (0)
#define RSTAMP(i_) RSTAMPG(group, gw == 0, i_)
  // Stop rule, evaluated by every workgroup after its work of step k on the votes of step k - 2 (long published):
  // a silent step g > 10 ends the decode with g + 1 steps.  v_vote: 64 + 8 floats of LDS.
#define STOP_CHECK(v_vote_)                                                                                        \
  if (p.use_stop && k >= 2) {                                                                                      \
    __syncthreads();                                                                                               \
    sweep<1, 1>(buf_of(p, kA_vote, (k - 2) & 3), (u32)(k - 1), 32, 64, (v_vote_), p.err, gave_up);                 \
    __syncthreads();                                                                                               \
    if (tid < 64) {                                                                                                \
      const bool any_ = __any((v_vote_)[tid] != 0.0f);                                                             \
      if (tid == 0) (v_vote_)[64] = any_ ? 1.0f : 0.0f;                                                            \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    if (k - 2 > 10 && (v_vote_)[64] == 0.0f) {                                                                     \
      done = k - 1;                                                                                                \
      break;                                                                                                       \
    }                                                                                                              \
  }
  const auto row8 = [&](const float* r, float4& w0, float4& w1) __attribute__((always_inline)) {
    w0 = *reinterpret_cast<const float4*>(r + 4 * lane), w1 = *reinterpret_cast<const float4*>(r + 256 + 4 * lane);
  };
  const auto dot8 = [&](const float4& w0, const float4& w1, const float* v, float acc) __attribute__((always_inline)) {
    return dot4(w1, *reinterpret_cast<const float4*>(v + 256 + 4 * lane), dot4(w0, *reinterpret_cast<const float4*>(v + 4 * lane), acc));
  };
  // E rows of this wave: E[rr][b][j] = W_rr[0 .. 255] . enc_q[b][j], rr < NR, for all B T positions (lanes stride over them;
  // every lane of the wave reads the same weight address: one cache transaction)
  const auto build_E = [&](float* E /*[NR][B][Tp] of this wave*/, const float* const* rows, int NR) __attribute__((always_inline)) {
    for (int i = lane; i < NR * B * Tp; i += 64) E[i] = 0.0f;
    for (int q = lane; q < B * T; q += 64) {
      const int b = q / T, j = q - b * T;
      const float4* e4 = reinterpret_cast<const float4*>(p.enc_q + (size_t)q * 256);
      for (int rr = 0; rr < NR; ++rr) {
        const float4* w4 = reinterpret_cast<const float4*>(rows[rr]);
        float acc = 0.0f;
        for (int d4 = 0; d4 < 64; ++d4) acc = dot4(w4[d4], e4[d4], acc);
        E[(rr * B + b) * Tp + j] = acc;
      }
    }
  };
  // softmax of a row of the gathered energies: s = exp(u - m) * inv (this wave; every lane gets m and inv = 1 / sum)
  const auto row_stats = [&](const float* ub, float& m, float& inv) __attribute__((always_inline)) {
    m = -INFINITY;
    for (int j = lane; j < T; j += 64) m = fmaxf(m, ub[j]);
    m = wave_max(m);
    float l = 0.0f;
    for (int j = lane; j < T; j += 64) l += __expf(ub[j] - m);
    inv = 1.0f / wave_sum(l);
  };

  if (group == 0) {
    // ============================================================ P1 / PQ / GR on the 32 workgroups of one XCD (round 3; before:
    // groups of 16 / 16 / 32 workgroups of their own, every hand-over across XCDs: 9.6 us per step at B = 1).  Workgroup h:
    //   GR  attention GRU unit 8 h + wave (all 32)
    //   P1  (h < 16)   fc1 rows 16 h + 2 wave, + 1 over x2 = x1 + h2
    //   PQ  (h >= 16)  fc2 row 8 (h - 16) + wave; later in the step query rows 16 (h - 16) + 2 wave, + 1
    // per step:  [x1, h2] P1 -> [p1] PQ -> [p] GR -> [ah] PQ (query) -> [pq]: three hand-overs inside the group; p1 and p never
    // leave it -- plain stores where the group has SEEN that it shares an XCD, write-through otherwise (ah is read by L1 too).
    const int h = gidx, unit = 8 * h + wave;
    const bool isP1 = h < 16;
    const int g = isP1 ? h : h - 16;
    float* v_p = lds;                      // [NB][128]
    float* v_ah = v_p + NB * 128;          // [NB][256] attn_h (the group's own array)
    float* v_u = v_ah + NB * 256;          // [64 POS] energies of the previous step
    float* pub = v_u + kRMaxBT;            // [16 NB]
    float* v_vote = pub + 16 * NB;         // [72]
    float* v_E = v_vote + 72;              // [8 waves][3][B][Tp]
    float* myE = v_E + (size_t)wave * 3 * B * Tp;
    float* v_x1 = v_E + (size_t)8 * 3 * B * Tp;   // P1: [NB][512]; PQ: p1 [NB][256]
    float* v_h2 = v_x1 + NB * kRL;                // P1: [NB][512]
    float* v_in = v_x1;
    // ---- GR: rows r, z, n of torch's GRUCell: input part over [ctx (256, via E) | p (128)], hidden part over attn_h (256)
    float2 w_p[3];
    float4 w_a[3];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
      w_p[gi] = *reinterpret_cast<const float2*>(p.gru_wih + (size_t)(gi * 256 + unit) * 384 + 256 + 2 * lane);
      w_a[gi] = *reinterpret_cast<const float4*>(p.gru_whh + (size_t)(gi * 256 + unit) * 256 + 4 * lane);
    }
    const float b_r = p.gru_bih[unit] + p.gru_bhh[unit], b_z = p.gru_bih[256 + unit] + p.gru_bhh[256 + unit];
    const float b_in = p.gru_bih[512 + unit], b_hn = p.gru_bhh[512 + unit];
    {
      const float* rows[3] = {p.gru_wih + (size_t)unit * 384, p.gru_wih + (size_t)(256 + unit) * 384, p.gru_wih + (size_t)(512 + unit) * 384};
      build_E(myE, rows, 3);
    }
    float ah_own[NB], hh[3][NB], ce[3][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ah_own[b] = 0.0f;
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) hh[gi][b] = ce[gi][b] = 0.0f;   // attn_h(-1) = 0, no attention yet
    }
    // ---- P1
    const int r0 = 16 * g + 2 * wave;
    float4 wa[2], wb[2];
    row8(p.fc1m + (size_t)r0 * kRL, wa[0], wa[1]);
    row8(p.fc1m + (size_t)(r0 + 1) * kRL, wb[0], wb[1]);
    const int rs1 = (lane / NB) & 1, bs1 = lane % NB, myrow = r0 + rs1;   // epilogue lane < 2 NB: (row select, batch row)
    const bool epi = lane < 2 * NB && bs1 < B;
    const float b_fc1 = p.fc1_b[myrow];
    // ---- PQ
    const int rf = 8 * g + wave, rq = 16 * g + 2 * wave;
    const float4 w_fc2 = *reinterpret_cast<const float4*>(p.fc2_w + (size_t)rf * 256 + 4 * lane);
    const float4 w_q0 = *reinterpret_cast<const float4*>(p.pq_w + (size_t)rq * 256 + 4 * lane);
    const float4 w_q1 = *reinterpret_cast<const float4*>(p.pq_w + (size_t)(rq + 1) * 256 + 4 * lane);
    const float b_fc2 = p.fc2_b[rf], b_q0 = p.pq_b[rq], b_q1 = p.pq_b[rq + 1];
    // ---- does the group share an XCD?
    bool plain = false;
    if (p.plain) {
      unsigned my;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
      my = (my & 15u) + 1u;
      if (wave == 0) {
        const unsigned* a = p.xcc + (p.map ? 8 * (lane & 31) : (lane & 31));
        unsigned v = 0;
        for (u32 spins = 0;; ++spins) {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
          if (__all(v != 0u) || gave_up) break;
          if (!spin_again(spins, p.err, gave_up)) break;
        }
        const bool same = __all(v == my) && !gave_up;
        if (lane == 0) v_vote[65] = same ? 1.0f : 0.0f;
      }
      __syncthreads();
      plain = __builtin_amdgcn_readfirstlane(__float_as_uint(v_vote[65])) != 0u;
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const u32 ep = (u32)k + 1;
      if (isP1) {
        // ---------------- P1
        RSTAMPG(0, h == 0, 0);
        float pf_b = b_fc1, pf_u = 1.0f;
        if (epi) {
          if (k > 0 && p.cadd) pf_b = p.cadd[(size_t)(k - 1) * B * 256 + ((size_t)(myrow >> 2) * B + bs1) * 4 + (myrow & 3)];
          if (p.drop_u1) pf_u = p.drop_u1[((size_t)k * B + bs1) * 256 + myrow];
        }
        float pre0[NB], pre1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) pre0[b] = pre1[b] = 0.0f;
        if (k > 0) {   // go frame at k = 0: fc1 over zeros
          sweep<NB, PER512>(buf_of(p, kA_x1, (k - 1) & 1), (u32)k, 512 * NB / 2, kRL, v_x1, p.err, gave_up);
          sweep<NB, PER512>(buf_of(p, kA_h2, (k - 1) & 1), (u32)k, 512 * NB / 2, kRL, v_h2, p.err, gave_up);
          __syncthreads();
          RSTAMPG(0, h == 0, 1);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float4 xa = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), xb = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane);
            const float4 ha = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), hb = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane);
            const float4 sa = make_float4(xa.x + ha.x, xa.y + ha.y, xa.z + ha.z, xa.w + ha.w);   // x2 = x1 + h2
            const float4 sb = make_float4(xb.x + hb.x, xb.y + hb.y, xb.z + hb.z, xb.w + hb.w);
            pre0[b] = dot4(wa[1], sb, dot4(wa[0], sa, 0.0f)), pre1[b] = dot4(wb[1], sb, dot4(wb[0], sa, 0.0f));
            wave_sum2(pre0[b], pre1[b]);
          }
        }
        if (lane < 2 * NB) {
          float v = rs1 ? sel(pre1, bs1) : sel(pre0, bs1);
          v = fmaxf(v + pf_b, 0.0f);
          if (p.drop_u1) v = pf_u >= kDropP ? v * (1.0f / (1.0f - kDropP)) : 0.0f;
          pub[(2 * wave + rs1) * NB + bs1] = epi ? v : 0.0f;
        }
        publish(buf_of(p, kA_p1, k & 1) + 16 * g * NB, ep, pub, 8 * NB, plain);
        RSTAMPG(0, h == 0, 2);
      } else {
        // ---------------- PQ: second prenet layer
        RSTAMPG(1, h == 16, 0);
        float pf_u = 1.0f;
        if (lane < NB && lane < B && p.drop_u2) pf_u = p.drop_u2[((size_t)k * B + lane) * 128 + rf];
        sweep<NB, PER256>(buf_of(p, kA_p1, k & 1), ep, 256 * NB / 2, 256, v_in, p.err, gave_up);
        __syncthreads();
        RSTAMPG(1, h == 16, 1);
        {
          float pre[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) pre[b] = wave_sum(dot4(w_fc2, *reinterpret_cast<const float4*>(v_in + b * 256 + 4 * lane), 0.0f));
          if (lane < NB) {
            float v = fmaxf(sel(pre, lane) + b_fc2, 0.0f);
            if (p.drop_u2) v = pf_u >= kDropP ? v * (1.0f / (1.0f - kDropP)) : 0.0f;
            pub[wave * NB + lane] = lane < B ? v : 0.0f;
          }
        }
        publish(buf_of(p, kA_p, k & 1) + 8 * g * NB, ep, pub, 4 * NB, plain);
        RSTAMPG(1, h == 16, 2);
      }
      // ---------------- GR: the attention GRU cell
      RSTAMPG(2, h == 0, 0);
      sweep<NB, PER128>(buf_of(p, kA_p, k & 1), ep, 128 * NB / 2, 128, v_p, p.err, gave_up);
      __syncthreads();   // (also: pub has been stored)
      RSTAMPG(2, h == 0, 1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float2 pv = *reinterpret_cast<const float2*>(v_p + b * 128 + 2 * lane);
        float sr = fmaf(w_p[0].y, pv.y, w_p[0].x * pv.x), sz = fmaf(w_p[1].y, pv.y, w_p[1].x * pv.x);
        float sn = fmaf(w_p[2].y, pv.y, w_p[2].x * pv.x), s_unused = 0.0f;
        wave_sum4(sr, sz, sn, s_unused);
        const float r_ = sigmoidf_(sr + ce[0][b] + hh[0][b] + b_r);
        const float z_ = sigmoidf_(sz + ce[1][b] + hh[1][b] + b_z);
        const float n_ = tanhf_(sn + ce[2][b] + b_in + r_ * (hh[2][b] + b_hn));
        ah_own[b] = (1.0f - z_) * n_ + z_ * ah_own[b];
        if (lane == 0) pub[wave * NB + b] = b < B ? ah_own[b] : 0.0f;
      }
      publish(buf_of(p, kA_ah, k & 1) + 8 * h * NB, ep, pub, 4 * NB);   // (write-through: the L1 group reads attn_h too)
      RSTAMPG(2, h == 0, 2);
      // the whole new attn_h (the group's own array): the hidden part of the next step (off the chain) and, in the PQ
      // workgroups, the query projection (on it)
      __syncthreads();
      sweep<NB, PER256>(buf_of(p, kA_ah, k & 1), ep, 256 * NB / 2, 256, v_ah, p.err, gave_up);
      __syncthreads();
      if (!isP1) {
        float q0[NB], q1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 a = *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane);
          q0[b] = dot4(w_q0, a, 0.0f), q1[b] = dot4(w_q1, a, 0.0f);
          wave_sum2(q0[b], q1[b]);
        }
        if (lane < 2 * NB) {
          const int rs = lane / NB, bs = lane % NB;
          const float v = (rs ? sel(q1, bs) : sel(q0, bs)) + (rs ? b_q1 : b_q0);
          pub[(2 * wave + rs) * NB + bs] = bs < B ? v : 0.0f;
        }
        publish(buf_of(p, kA_pq, k & 1) + 16 * g * NB, ep, pub, 8 * NB);
        RSTAMPG(1, h == 16, 3);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) hh[gi][b] = wave_sum(dot4(w_a[gi], *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane), 0.0f));
      // ... and its context part E . s(k), once the attention group has published this step's energies
      sweep_u<POS, PERU>(buf_of(p, kA_u, k & 1), ep, v_u, p.err, gave_up);
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b < B) {
          float m, inv;
          row_stats(v_u + b * T, m, inv);
          float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
          for (int j = lane; j < T; j += 64) {
            const float s = __expf(v_u[b * T + j] - m) * inv;
            a0 = fmaf(myE[(0 * B + b) * Tp + j], s, a0), a1 = fmaf(myE[(1 * B + b) * Tp + j], s, a1), a2 = fmaf(myE[(2 * B + b) * Tp + j], s, a2);
          }
          ce[0][b] = wave_sum(a0), ce[1][b] = wave_sum(a1), ce[2][b] = wave_sum(a2);
        }
      }
      STOP_CHECK(v_vote)
      __syncthreads();
    }
  } else if (group == 3) {
    // ============================================================ AT: energies of positions q = g POS + i (q = b T + j)
    const int g = gw;
    float* v_pq = lds;                    // [NB][256]
    float* v_u = v_pq + NB * 256;         // [64 POS]
    float* v_s = v_u + kRMaxBT;           // [NB][Tp] attention weights of the last step
    float* v_cum = v_s + NB * Tp;         // [NB][Tp]
    float* red = v_cum + NB * Tp;         // [8]
    float* misc = red + 8;                // [64] conv features of 2 positions
    float* cpart = misc + 64;             // [2][8][32] conv partial sums
    float* pub = cpart + 2 * 8 * 32;      // [POS]
    float* v_vote = pub + POS;            // [72]
    for (int i = tid; i < 2 * NB * Tp; i += kRT) v_s[i] = 0.0f;
    // thread d = tid & 255 holds row d of L; thread (f = tid & 31, part = (tid >> 5) & 7) holds conv taps kk = 8 part ..
    // 8 part + 7 of filter f (kk = 31 ch + tap, 62 in all)
    const int ad = tid & 255, half = tid >> 8;
    float w_L[32], w_conv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 t4 = *reinterpret_cast<const float4*>(p.L_w + (size_t)ad * 32 + 4 * i);
      w_L[4 * i] = t4.x, w_L[4 * i + 1] = t4.y, w_L[4 * i + 2] = t4.z, w_L[4 * i + 3] = t4.w;
    }
    const int cf = tid & 31, cpart_i = (tid >> 5) & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = 8 * cpart_i + i;
      w_conv[i] = kk < 62 ? p.conv_t[kk * 32 + cf] : 0.0f;
    }
    const float c_Lb = p.L_b[ad], c_v = p.v_w[ad];
    // this thread's positions: ii = 2 t + half, t < POS / 2 (the two halves of the workgroup work on a pair of positions at
    // a time), attention dim ad: encoder projection; + L_b + L . conv features of the position (location part of the energy)
    constexpr int HP = POS / 2;
    float ep_own[HP], base_own[HP];
    int pos_b[HP], pos_j[HP];
#pragma unroll
    for (int t = 0; t < HP; ++t) {
      const int q = g * POS + 2 * t + half;
      const bool ok = q < B * T;
      pos_b[t] = ok ? q / T : -1;
      pos_j[t] = ok ? q - (q / T) * T : 0;
      ep_own[t] = ok ? p.enc_p[(size_t)q * 256 + ad] : 0.0f;
      base_own[t] = ep_own[t] + c_Lb;   // step 0: location features are zero
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const u32 ep = (u32)k + 1;
      RSTAMP(0);
      sweep<NB, PER256>(buf_of(p, kA_pq, k & 1), ep, 256 * NB / 2, 256, v_pq, p.err, gave_up);
      __syncthreads();
      RSTAMP(1);
#pragma unroll
      for (int t = 0; t < HP; ++t) {   // two positions at a time, one per half of the workgroup
        const bool have = pos_b[t] >= 0;
        const int b = have ? pos_b[t] : 0;
        float e = have ? c_v * tanhf_(v_pq[b * 256 + ad] + base_own[t]) : 0.0f;
        e = wave_sum(e);
        if (lane == 0) red[wave] = e;        // waves 0-3: position 2 t, waves 4-7: position 2 t + 1
        __syncthreads();
        if ((tid & 255) == 0) pub[2 * t + half] = (red[4 * half] + red[4 * half + 1]) + (red[4 * half + 2] + red[4 * half + 3]);
        if (t + 1 < HP) __syncthreads();
      }
      publish(buf_of(p, kA_u, k & 1) + g * USLOT, ep, pub, POS / 2);
      RSTAMP(2);
      // off the chain: this step's attention weights (every workgroup normalises the whole rows itself), the attention
      // state, and the location features of own positions for the NEXT step (conv over the new cum / att, L projection)
      __syncthreads();
      sweep_u<POS, PERU>(buf_of(p, kA_u, k & 1), ep, v_u, p.err, gave_up);
      __syncthreads();
      if (wave < NB && wave < B) {
        const int b = wave;
        float m, inv;
        row_stats(v_u + b * T, m, inv);
        float* arow = p.att_out + ((size_t)b * n + k) * T;
        for (int j = lane; j < T; j += 64) {
          const float s = __expf(v_u[b * T + j] - m) * inv;
          v_s[b * Tp + j] = s;
          v_cum[b * Tp + j] += s;
          if (g == b) arow[j] = s;
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < HP; ++t) {
        const bool have = pos_b[t] >= 0;
        const int b = have ? pos_b[t] : 0, j = pos_j[t];
        float a = 0.0f;
        if (have) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kk = 8 * cpart_i + q, ch = kk >= 31 ? 1 : 0, tap = kk - 31 * ch;
            const int tt = j + tap - 15;
            const float lv = (kk < 62 && tt >= 0 && tt < T) ? (ch ? v_s : v_cum)[b * Tp + tt] : 0.0f;
            a = fmaf(w_conv[q], lv, a);
          }
        }
        cpart[(half * 8 + cpart_i) * 32 + cf] = a;
        __syncthreads();
        if ((tid & 255) < 32) {
          float f = 0.0f;
#pragma unroll
          for (int q = 0; q < 8; ++q) f += cpart[(half * 8 + q) * 32 + (tid & 31)];
          misc[half * 32 + (tid & 31)] = f;
        }
        __syncthreads();
        float pl = c_Lb;
#pragma unroll
        for (int f = 0; f < 32; ++f) pl = fmaf(w_L[f], misc[half * 32 + f], pl);
        if (have) base_own[t] = ep_own[t] + pl;
        __syncthreads();
      }
      STOP_CHECK(v_vote)
      __syncthreads();
    }
  } else if (group == 4) {
    // ============================================================ L1: softmax + first LSTM cell, unit 8 g + wave
    const int g = gw, unit = 8 * g + wave;
    float* v_ah = lds;                     // [NB][256]
    float* v_h1 = v_ah + NB * 256;         // [NB][512] previous h1 (the group's own array)
    float* v_u = v_h1 + NB * kRL;          // [64 POS]
    float* pub = v_u + kRMaxBT;            // [2][8 NB]: x1 | h1
    float* v_vote = pub + 16 * NB;         // [72]
    float* v_E = v_vote + 72;              // [8 waves][5][B][Tp]: gates i, f, g, o and the rnn_input row of the unit
    float* myE = v_E + (size_t)wave * 5 * B * Tp;
    for (int i = tid; i < NB * kRL; i += kRT) v_h1[i] = 0.0f;   // h1(-1) = 0
    float4 w_a[4], w_h[4][2], w_y;
    float b_l1[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
      const int lrow = gt * kRL + unit;
      w_a[gt] = *reinterpret_cast<const float4*>(p.l1_wf + (size_t)lrow * 512 + 256 + 4 * lane);   // attn_h columns
      row8(p.l1_whh + (size_t)lrow * kRL, w_h[gt][0], w_h[gt][1]);
      b_l1[gt] = p.l1_b[lrow];
    }
    w_y = *reinterpret_cast<const float4*>(p.rin_w + (size_t)unit * 512 + 256 + 4 * lane);
    const float b_y = p.rin_b[unit];
    {
      const float* rows[5] = {p.l1_wf + (size_t)unit * 512, p.l1_wf + (size_t)(kRL + unit) * 512, p.l1_wf + (size_t)(2 * kRL + unit) * 512,
                              p.l1_wf + (size_t)(3 * kRL + unit) * 512, p.rin_w + (size_t)unit * 512};
      build_E(myE, rows, 5);
    }
    float c1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) c1[b] = 0.0f;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const u32 ep = (u32)k + 1;
      RSTAMP(0);
      // off the chain (attn_h arrives two hand-overs before the energies): the attn_h and h1 parts of the gates
      sweep<NB, PER256>(buf_of(p, kA_ah, k & 1), ep, 256 * NB / 2, 256, v_ah, p.err, gave_up);
      __syncthreads();
      float part[5][NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 a = *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane);
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) part[gt][b] = dot8(w_h[gt][0], w_h[gt][1], v_h1 + b * kRL, dot4(w_a[gt], a, 0.0f));
        part[4][b] = dot4(w_y, a, 0.0f);
      }
      sweep_u<POS, PERU>(buf_of(p, kA_u, k & 1), ep, v_u, p.err, gave_up);
      __syncthreads();
      RSTAMP(1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        // one pass over the row: unnormalised weights e_j = exp(u_j - max), their sum l and the five E contractions; the
        // normaliser is applied to the wave totals (eleven sums as three interleaved 4-way reductions)
        float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, l = 0.0f, unused = 0.0f;
        if (b < B) {
          float m = -INFINITY;
          for (int j = lane; j < T; j += 64) m = fmaxf(m, v_u[b * T + j]);
          m = wave_max(m);
          for (int j = lane; j < T; j += 64) {
            const float e = __expf(v_u[b * T + j] - m);
            l += e;
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) acc[rr] = fmaf(myE[(rr * B + b) * Tp + j], e, acc[rr]);
          }
        } else {
          l = 1.0f;
        }
        wave_sum4(part[0][b], part[1][b], part[2][b], part[3][b]);
        wave_sum4(acc[0], acc[1], acc[2], acc[3]);
        wave_sum4(part[4][b], acc[4], l, unused);
        const float inv = 1.0f / l;
        const float ig = sigmoidf_(fmaf(acc[0], inv, part[0][b]) + b_l1[0]);
        const float fg = sigmoidf_(fmaf(acc[1], inv, part[1][b]) + b_l1[1]);
        const float gg = tanhf_(fmaf(acc[2], inv, part[2][b]) + b_l1[2]);
        const float og = sigmoidf_(fmaf(acc[3], inv, part[3][b]) + b_l1[3]);
        const float y = fmaf(acc[4], inv, part[4][b]) + b_y;
        c1[b] = fg * c1[b] + ig * gg;
        const float h1 = og * tanhf_(c1[b]);
        if (lane == 0) pub[wave * NB + b] = b < B ? y + h1 : 0.0f, pub[8 * NB + wave * NB + b] = b < B ? h1 : 0.0f;
      }
      publish(buf_of(p, kA_x1, k & 1) + 8 * g * NB, ep, pub, 4 * NB);
      if (tid < 4 * NB) st_gran2(buf_of(p, kA_h1, k & 1) + 8 * g * NB + 2 * tid, ep, pub[8 * NB + 2 * tid], pub[8 * NB + 2 * tid + 1]);
      RSTAMP(2);
      // off the chain: the whole new h1 (the group's own array) for the next step
      __syncthreads();
      sweep<NB, PER512>(buf_of(p, kA_h1, k & 1), ep, 512 * NB / 2, kRL, v_h1, p.err, gave_up);
      STOP_CHECK(v_vote)
      __syncthreads();
    }
  } else {
    // ============================================================ L2: second LSTM cell, unit 8 g + wave; mel frames; votes
    const int g = gw, unit = 8 * g + wave;
    float* v_x1 = lds;                     // [NB][512]
    float* v_h2 = v_x1 + NB * kRL;         // [NB][512] previous h2 (the group's own array)
    float* pub = v_h2 + NB * kRL;          // [8 NB]
    float* alive_w = pub + 8 * NB;         // [8]
    float* v_vote = alive_w + 8;           // [72]
    for (int i = tid; i < NB * kRL; i += kRT) v_h2[i] = 0.0f;
    float4 w_x[4][2], w_h[4][2];
    float b_l2[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
      const int lrow = gt * kRL + unit;
      row8(p.l2_wih + (size_t)lrow * kRL, w_x[gt][0], w_x[gt][1]);
      row8(p.l2_whh + (size_t)lrow * kRL, w_h[gt][0], w_h[gt][1]);
      b_l2[gt] = p.l2_b[lrow];
    }
    float c2[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) c2[b] = 0.0f;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
      const u32 ep = (u32)k + 1;
      RSTAMP(0);
      float part[4][NB];   // the h2 part of the gates: needs the previous step only
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) part[gt][b] = dot8(w_h[gt][0], w_h[gt][1], v_h2 + b * kRL, 0.0f);
      sweep<NB, PER512>(buf_of(p, kA_x1, k & 1), ep, 512 * NB / 2, kRL, v_x1, p.err, gave_up);
      __syncthreads();
      RSTAMP(1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float s0 = dot8(w_x[0][0], w_x[0][1], v_x1 + b * kRL, part[0][b]), s1 = dot8(w_x[1][0], w_x[1][1], v_x1 + b * kRL, part[1][b]);
        float s2 = dot8(w_x[2][0], w_x[2][1], v_x1 + b * kRL, part[2][b]), s3 = dot8(w_x[3][0], w_x[3][1], v_x1 + b * kRL, part[3][b]);
        wave_sum4(s0, s1, s2, s3);
        const float ig = sigmoidf_(s0 + b_l2[0]), fg = sigmoidf_(s1 + b_l2[1]), gg = tanhf_(s2 + b_l2[2]), og = sigmoidf_(s3 + b_l2[3]);
        c2[b] = fg * c2[b] + ig * gg;
        const float h2 = og * tanhf_(c2[b]);
        if (lane == 0) pub[wave * NB + b] = b < B ? h2 : 0.0f;
      }
      publish(buf_of(p, kA_h2, k & 1) + 8 * g * NB, ep, pub, 4 * NB);
      RSTAMP(2);
      // off the chain: the whole new h2 (the group's own array); the mel frames of this step from x2 = x1 + h2 (rows
      // j = 8 g + wave + 512 i of the r n_mels projected rows) and the step's stop vote (any value >= threshold)
      __syncthreads();
      sweep<NB, PER512>(buf_of(p, kA_h2, k & 1), ep, 512 * NB / 2, kRL, v_h2, p.err, gave_up);
      __syncthreads();
      bool alive = false;
      for (int j = unit; j < p.nmr; j += 512) {
        float4 wa, wb;
        row8(p.mel_w + (size_t)j * kRL, wa, wb);
        const int m = j / p.r, f = j - m * p.r;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 xa = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), xb = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane);
          const float4 ha = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), hb = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane);
          const float4 sa = make_float4(xa.x + ha.x, xa.y + ha.y, xa.z + ha.z, xa.w + ha.w);
          const float4 sb = make_float4(xb.x + hb.x, xb.y + hb.y, xb.z + hb.z, xb.w + hb.w);
          const float v = wave_sum(dot4(wb, sb, dot4(wa, sa, 0.0f)));
          if (b < B) {
            if (lane == 0) p.mel_out[(size_t)b * p.mel_sb + (size_t)m * p.mel_sm + (size_t)k * p.r + f] = v;
            alive |= !(v < p.thr);
          }
        }
      }
      if (p.use_stop) {
        if (lane == 0) alive_w[wave] = alive ? 1.0f : 0.0f;
        __syncthreads();
        if (tid == 0) {
          float a = 0.0f;
#pragma unroll
          for (int w = 0; w < 8; ++w) a += alive_w[w];
          st_gran(buf_of(p, kA_vote, k & 3) + g, ep, a > 0.0f ? 1.0f : 0.0f);
        }
      }
      STOP_CHECK(v_vote)
      __syncthreads();
    }
  }
#undef RSTAMP
#undef RSTAMPG
#undef STOP_CHECK
  if (group == 0 && gidx == 0 && tid == 0) *p.n_done = done;
}

}  // namespace
