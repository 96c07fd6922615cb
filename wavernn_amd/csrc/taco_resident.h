// Free-running decode for SMALL batches (B <= 8) as ONE persistent, weights-resident launch per call.
// Included by taco_decoder.hip (after struct wrnn_decoder); replaces the 7-launches-per-step loop of
// wrnn_decoder_run_free where a step is pure latency (reference loop being replaced: models/tacotron.py:185-235
// driven as :339-345 drives it, with the build-defined feedback rule of SURVEY.md section 8b).
//
// 256 workgroups x 512 threads, one per compute unit, launched cooperatively (all co-resident).  Every workgroup owns
// a fixed slice of every layer -- one fc1 row, one fc2 row (half of them), one attention-GRU unit, one query row, two
// units of each LSTM cell, the attention positions p = wg, wg + 256, ... -- and keeps that slice's weights in REGISTERS
// for the whole utterance (20.5 MB of fp32 weights over 131,072 threads: ~100 registers each).  Nothing is fetched from
// memory inside the step loop except the activations handed from stage to stage.
//
// Stage hand-over ("seam"): every value a stage produces is published as ONE 8-byte {tag, value} granule with a
// write-through (sc1) store; a consumer sweeps the stage's granule array with sc1 loads until every tag equals the
// epoch it waits for (tag = consuming step + 1, never 0; the arrays are zeroed before the launch), then stages the
// values in LDS.  The data is the flag: no counter, no fence (CDNA4 guide, Guideline 16 form R2).  Measured price of
// such an all-to-all seam: 2.2-2.9 us for 128..4096 granules (tools/microbench/gather_bench.hip), against 1.7 us of
// launch boundary PLUS ~3 us of cold weight fetch per stage in the launch-per-stage loop.
//
// The step's dependent chain is cut to 7 seams by algebra:
//  * the context vector is never formed.  Every consumer of ctx = sum_j s_j enc_q_j is linear in it, so
//    W . ctx = sum_j s_j (W . enc_q_j) = E . s with E = W . enc_q^T computed once per utterance (per workgroup, for its
//    own rows, at kernel start, kept in LDS): the attention GRU, the rnn_input projection and LSTM1 contract over the
//    T attention weights instead of the 256 context dims, and the attention stage ends at the energies;
//  * rnn_input is folded into LSTM1 (as in the launch loop) and x1 = y + h1 is formed by the workgroup that owns the
//    unit; fc1 is folded through the mel projection of the group's last frame, the mel frames themselves are
//    projected off the chain;
//  * the location features (conv + L projection) of a workgroup's positions depend on the previous step's attention
//    only and are computed in the shadow of the following seams.
//  per step:  [h2'] fc1 row -> [p1] fc2 row -> [p] GRU unit -> [ah] query row -> [pq] energies of own positions
//             -> [u] softmax (every workgroup, redundantly) + LSTM1 units -> [x1] LSTM2 units -> next step
#pragma once
#include "seam.h"

namespace {

constexpr int kRL = 512;         // lstm_dims the resident kernel is built for
constexpr int kRMaxB = 4;
constexpr int kRMaxBT = 1024;    // B * T (E matrices and attention state live in LDS)

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
  u64 *g_h2, *g_p1, *g_h1, *g_vote, *g_p, *g_ah, *g_pq, *g_u, *g_x1;   // granule arrays: 256 slots of kSlot granules each
  int B, T, n, r, nmr;
  long mel_sb, mel_sm;
  float thr;
  int use_stop;
  // diagnostic (WRNN_RESIDENT_STAMPS=<file>): 100 MHz timestamps of workgroups 0 and 131 at 16 points of steps
  // kStampStep0 .. +7, [2][8][16] u64; null otherwise
  u64* stamps;
};
constexpr int kStampStep0 = 200;

// NB: batch rows the register arrays are sized for (B <= NB); POS: attention positions per workgroup (B T <= 256 POS);
// FC2L: the whole second prenet layer (128 x 256, 128 KiB) lives in every workgroup's LDS and is evaluated redundantly by
// all of them, which removes the [p] seam (6 seams per step instead of 7); needs the LDS room, i.e. B = 1 and short T.
template <int NB, int POS, bool FC2L>
__global__ __launch_bounds__(kRT) void resident_free_kernel(const ResidentParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int B = p.B, T = p.T, n = p.n;
  const int Tp = (T + 63) & ~63;  // padded attention length: E and s are zero beyond T
  // ---- LDS carve (floats); vectors hold B rows back to back (granule index = LDS index)
  float* v_x1 = lds;                       // [NB][512]
  float* v_h2 = v_x1 + NB * kRL;           // [NB][512]
  float* v_s2 = v_h2 + NB * kRL;           // p1 [B][256] | h1 [B][512] | votes [256]   (runtime offsets)
  float* v_p = v_s2 + NB * 768 + 256;      // [NB][128]
  float* v_ah = v_p + NB * 128;            // [NB][256]
  float* v_pq = v_ah + NB * 256;           // [NB][256]
  float* v_s = v_pq + NB * 256;            // [NB][Tp] attention weights of the last step
  float* v_cum = v_s + NB * Tp;            // [NB][Tp]
  float* v_E = v_cum + NB * Tp;            // [13][NB][Tp]: rows 0-7 LSTM1 gates, 8-9 rnn_input, 10-12 GRU r,z,n
  float* v_u = v_E + 13 * NB * Tp;         // [B * T <= 1024] gathered energies
  float* red = v_u + kRMaxBT;              // [16][NB] row sums
  float* misc = red + 16 * NB;             // [0..63] conv features of 2 positions, [64..71] alive flags, [80..83] vote ballots
  float* cpart = misc + 128;               // [2][8][32] conv partial sums
  float* v_fc2 = cpart + 2 * 8 * 32;       // FC2L: [16][128][4] float4: element (i, row, q) = fc2_w[row][4 (q + 4 i) .. + 3]
  float* const v_p1 = v_s2;
  float* const v_h1 = v_s2 + 256 * B;
  float* const v_vote = v_s2 + 768 * B;

  // ---- weights into registers -------------------------------------------------------------------------------------
  const int unit = 2 * wg + (wave >> 2), gate = wave & 3;  // LSTM row of this wave (torch row gate * 512 + unit)
  const int lrow = gate * kRL + unit;
  float4 w_l1[3], w_l2[4], w_y, w_pq, w_fc1[2], w_fc2, w_gah;
  float2 w_gp;
  {
    const float* a = p.l1_wf + (size_t)lrow * 512 + 256;   // attn_h columns
    const float* h = p.l1_whh + (size_t)lrow * kRL;
    w_l1[0] = *reinterpret_cast<const float4*>(a + 4 * lane);
    w_l1[1] = *reinterpret_cast<const float4*>(h + 4 * lane);
    w_l1[2] = *reinterpret_cast<const float4*>(h + 256 + 4 * lane);
    const float* x = p.l2_wih + (size_t)lrow * kRL;
    const float* hh = p.l2_whh + (size_t)lrow * kRL;
    w_l2[0] = *reinterpret_cast<const float4*>(x + 4 * lane);
    w_l2[1] = *reinterpret_cast<const float4*>(x + 256 + 4 * lane);
    w_l2[2] = *reinterpret_cast<const float4*>(hh + 4 * lane);
    w_l2[3] = *reinterpret_cast<const float4*>(hh + 256 + 4 * lane);
    // rnn_input rows of the two units: waves 0, 1
    w_y = *reinterpret_cast<const float4*>(p.rin_w + (size_t)(2 * wg + (wave & 1)) * 512 + 256 + 4 * lane);
    w_pq = *reinterpret_cast<const float4*>(p.pq_w + (size_t)wg * 256 + 4 * lane);
    w_fc1[0] = *reinterpret_cast<const float4*>(p.fc1m + (size_t)wg * kRL + 4 * lane);
    w_fc1[1] = *reinterpret_cast<const float4*>(p.fc1m + (size_t)wg * kRL + 256 + 4 * lane);
    w_fc2 = *reinterpret_cast<const float4*>(p.fc2_w + (size_t)(wg & 127) * 256 + 4 * lane);
    // attention GRU unit wg, waves 0..3 = rows r, z, gi_n, gh_n over [p (128: two per lane) | attn_h (256: four per lane)]
    const int g3 = gate == 3 ? 2 : gate;
    const float2 gp = *reinterpret_cast<const float2*>(p.gru_wih + (size_t)(g3 * 256 + wg) * 384 + 256 + 2 * lane);
    const float4 ga = *reinterpret_cast<const float4*>(p.gru_whh + (size_t)(g3 * 256 + wg) * 256 + 4 * lane);
    w_gp = gate == 3 ? make_float2(0.f, 0.f) : gp;
    w_gah = gate == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : ga;
  }
  // location-sensitive part: thread d = tid & 255 holds row d of L; thread (f = tid & 31, part = (tid >> 5) & 7) holds
  // conv taps kk = 8 part .. 8 part + 7 of filter f (kk = 31 ch + tap, 62 in all)
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
  // biases, read once: a load from memory in a stage's epilogue would sit on the step's dependent chain
  const float b_fc1 = p.fc1_b[wg], b_fc2 = p.fc2_b[wg & 127], b_pq = p.pq_b[wg];
  const float b_gr = p.gru_bih[wg] + p.gru_bhh[wg], b_gz = p.gru_bih[256 + wg] + p.gru_bhh[256 + wg];
  const float b_gin = p.gru_bih[512 + wg], b_ghn = p.gru_bhh[512 + wg];
  const int eu = tid < 2 * NB ? tid / NB : 0, eb = tid % NB, eU = 2 * wg + eu;   // LSTM epilogue thread: (unit, batch row)
  float b_l1[4], b_l2[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) b_l1[g] = p.l1_b[g * kRL + eU], b_l2[g] = p.l2_b[g * kRL + eU];
  const float b_y = p.rin_b[eU];
  const int frow = tid >> 2, fq = tid & 3;                                           // FC2L: (row, quarter of K)
  const float b_fc2r = p.fc2_b[frow];
  // ---- own attention positions: q = wg + 256 i < B T; encoder projection of the position in registers
  float ep_own[POS];
  int pos_b[POS], pos_j[POS];
#pragma unroll
  for (int i = 0; i < POS; ++i) {
    const int q = wg + kRW * i;
    const bool ok = q < B * T;
    pos_b[i] = ok ? q / T : -1;
    pos_j[i] = ok ? q - (q / T) * T : 0;
    ep_own[i] = ok ? p.enc_p[(size_t)q * 256 + ad] : 0.0f;
  }

  // ---- zero the LDS state, then E = W . enc_q^T for the 13 context-consuming rows of this workgroup -----------------
  {
    const int total = (int)(v_fc2 - lds);
    for (int i = tid; i < total; i += kRT) lds[i] = 0.0f;
    if (FC2L)
      for (int e = tid; e < 16 * 128 * 4; e += kRT) {  // e = (i * 128 + row) * 4 + q
        const int q = e & 3, row = (e >> 2) & 127, i = e >> 9;
        reinterpret_cast<float4*>(v_fc2)[e] = *reinterpret_cast<const float4*>(p.fc2_w + (size_t)row * 256 + 4 * (q + 4 * i));
      }
  }
  __syncthreads();
  {
    // the 13 weight rows (context columns 0..255) are read straight from memory: every lane reads the same address
    // (one cache transaction per wave), once per utterance
    const float4* wsrc[13];
#pragma unroll
    for (int rr = 0; rr < 13; ++rr) {
      const float* src;
      if (rr < 8) src = p.l1_wf + (size_t)((rr & 3) * kRL + 2 * wg + (rr >> 2)) * 512;
      else if (rr < 10) src = p.rin_w + (size_t)(2 * wg + (rr - 8)) * 512;
      else src = p.gru_wih + (size_t)((rr - 10) * 256 + wg) * 384;
      wsrc[rr] = reinterpret_cast<const float4*>(src);
    }
    for (int q = tid; q < B * T; q += kRT) {
      const int b = q / T, j = q - b * T;
      float acc[13];
#pragma unroll
      for (int rr = 0; rr < 13; ++rr) acc[rr] = 0.0f;
      const float4* e4 = reinterpret_cast<const float4*>(p.enc_q + (size_t)q * 256);
      for (int d4 = 0; d4 < 64; ++d4) {
        const float4 e = e4[d4];
#pragma unroll
        for (int rr = 0; rr < 13; ++rr) acc[rr] = dot4(wsrc[rr][d4], e, acc[rr]);
      }
#pragma unroll
      for (int rr = 0; rr < 13; ++rr) v_E[(rr * NB + b) * Tp + j] = acc[rr];
    }
  }
  // recurrent state of the units this workgroup owns, in the registers of its epilogue threads
  // (thread t < 2 NB: LSTM unit t / NB, batch row t % NB; thread t < NB: GRU unit, batch row t)
  float c1 = 0.f, c2 = 0.f, h1 = 0.f, h2 = 0.f, ahu = 0.f;
  // initial publications: h1(-1) = 0 and "alive" votes for the first stage-2 sweep (epoch 1)
  {
    if (tid < 2 * NB && eb < B) st_gran(p.g_h1 + wg * kSlot + 2 * eb + eu, 1u, 0.0f);
    if (tid == 0) st_gran(p.g_vote + wg * kSlot, 1u, 1.0f);
  }
  float base_own[POS];  // ep + L_b + L . conv features of own position i, attention dim ad (location part of the energy)
#pragma unroll
  for (int i = 0; i < POS; ++i) base_own[i] = ep_own[i] + c_Lb;   // step 0: location features are zero
  __syncthreads();

  // the hand-over segments (what every sweep reads and where it stages it)
  const int BT = B * T;
  const SlotSeg sg_h2[1] = {{p.g_h2, 8, 2 * B, 2, kRL, (int)(v_h2 - lds), 1 << 30}};
  const SlotSeg sg_x1[1] = {{p.g_x1, 8, 2 * B, 2, kRL, (int)(v_x1 - lds), 1 << 30}};
  const SlotSeg sg_s2[3] = {{p.g_p1, 8, B, 1, 256, (int)(v_p1 - lds), 1 << 30},
                            {p.g_h1, 8, 2 * B, 2, kRL, (int)(v_h1 - lds), 1 << 30},
                            {p.g_vote, 8, p.use_stop ? 1 : 0, 1, 256, (int)(v_vote - lds), 1 << 30}};
  const SlotSeg sg_p[1] = {{p.g_p, 7, B, 1, 128, (int)(v_p - lds), 1 << 30}};
  const SlotSeg sg_ah[1] = {{p.g_ah, 8, B, 1, 256, (int)(v_ah - lds), 1 << 30}};
  const SlotSeg sg_pq[1] = {{p.g_pq, 8, B, 1, 256, (int)(v_pq - lds), 1 << 30}};
  const SlotSeg sg_u[1] = {{p.g_u, 8, POS, 1, 256, (int)(v_u - lds), BT}};   // position q = w + 256 e
  int done = n;
  bool gave_up = false;   // this wave has timed out on a sweep: the call has failed, do not wait again
  const int swg = wg == 0 ? 0 : wg == 131 ? 1 : -1;
#define RSTAMP(i)                                                                                                  \
  do {                                                                                                             \
    if (p.stamps && swg >= 0 && tid == 0 && k >= kStampStep0 && k < kStampStep0 + 8)                               \
      p.stamps[(swg * 8 + (k - kStampStep0)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                       \
  } while (0)
  for (int k = 0; k < n; ++k) {
    const u32 ep = (u32)k + 1;
    RSTAMP(0);
    // per-step operands from memory, requested before the first seam of the step and used after it
    float pf_b1 = b_fc1, pf_u1 = 1.0f, pf_u2[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pf_u2[b] = 1.0f;
    if (wave == 0 && lane < B) {
      if (k > 0 && p.cadd) pf_b1 = p.cadd[(size_t)(k - 1) * B * 256 + ((size_t)(wg >> 2) * B + lane) * 4 + (wg & 3)];
      if (p.drop_u1) pf_u1 = p.drop_u1[((size_t)k * B + lane) * 256 + wg];
      if (!FC2L && p.drop_u2 && wg < 128) pf_u2[0] = p.drop_u2[((size_t)k * B + lane) * 128 + wg];
    }
    if (FC2L && p.drop_u2 && fq == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) pf_u2[b] = p.drop_u2[((size_t)k * B + min(b, B - 1)) * 128 + frow];
    }
    // ================= stage 1: fc1 row wg of step k; mel frames (and stop votes) of group k-1 =====================
    if (k > 0) {
      gather<NB, 1>(sg_h2, ep, lds, p.err, gave_up);
      __syncthreads();
    }
    RSTAMP(1);
    if (wave == 0) {
      float pre[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 xa = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), xb = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane);
        const float4 ha = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), hb = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane);
        const float4 sa = make_float4(xa.x + ha.x, xa.y + ha.y, xa.z + ha.z, xa.w + ha.w);   // x2 = x1 + h2
        const float4 sb = make_float4(xb.x + hb.x, xb.y + hb.y, xb.z + hb.z, xb.w + hb.w);
        pre[b] = wave_sum(dot4(w_fc1[1], sb, dot4(w_fc1[0], sa, 0.0f)));
      }
      if (lane < NB && lane < B) {
        float v = k == 0 ? 0.0f : sel(pre, lane);   // go frame: fc1 over zeros
        v = fmaxf(v + pf_b1, 0.0f);
        if (p.drop_u1) v = pf_u1 >= kDropP ? v * (1.0f / (1.0f - kDropP)) : 0.0f;
        st_gran(p.g_p1 + wg * kSlot + lane, ep, v);
      }
    } else if (k > 0) {
      // mel rows j = wg, wg + 256, ... of group k-1, dealt to waves 1..7; any value >= thr keeps the decode alive
      bool alive = false;
      for (int i = 0, j = wg; j < p.nmr; j += kRW, ++i) {
        if (1 + (i % 7) != wave) continue;
        const float4 wa = *reinterpret_cast<const float4*>(p.mel_w + (size_t)j * kRL + 4 * lane);
        const float4 wb = *reinterpret_cast<const float4*>(p.mel_w + (size_t)j * kRL + 256 + 4 * lane);
        const int m = j / p.r, f = j - m * p.r;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 xa = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), xb = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane);
          const float4 ha = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), hb = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane);
          const float4 sa = make_float4(xa.x + ha.x, xa.y + ha.y, xa.z + ha.z, xa.w + ha.w);
          const float4 sb = make_float4(xb.x + hb.x, xb.y + hb.y, xb.z + hb.z, xb.w + hb.w);
          const float v = wave_sum(dot4(wb, sb, dot4(wa, sa, 0.0f)));
          if (b < B) {
            if (lane == 0) p.mel_out[(size_t)b * p.mel_sb + (size_t)m * p.mel_sm + (size_t)(k - 1) * p.r + f] = v;
            alive |= !(v < p.thr);
          }
        }
      }
      if (lane == 0 && alive) misc[64 + wave] = 1.0f;
    }
    if (p.use_stop && k > 0) {  // one vote granule per workgroup, every step (the sweep waits for all 256)
      __syncthreads();
      if (tid == 0) {
        float a = 0.0f;
#pragma unroll
        for (int w = 1; w < 8; ++w) a += misc[64 + w], misc[64 + w] = 0.0f;
        st_gran(p.g_vote + wg * kSlot, ep, a > 0.0f ? 1.0f : 0.0f);
      }
    }
    // ================= stage 2: second prenet layer; h1(k-1) and the votes arrive in the same sweep ===================
    {
      RSTAMP(2);
      gather<(NB * 768 + 256 + kRT - 1) / kRT, 3>(sg_s2, ep, lds, p.err, gave_up);
      __syncthreads();
      RSTAMP(3);
    }
    if (p.use_stop && k - 1 > 10) {  // group k-1 all below the threshold: the decode ended with it
      if (tid < kRW) {
        const bool any = __any(v_vote[tid] != 0.0f);
        if (lane == 0) misc[80 + wave] = any ? 1.0f : 0.0f;
      }
      __syncthreads();
      const bool alive = (misc[80] + misc[81]) + (misc[82] + misc[83]) > 0.0f;   // the same in every thread
      if (!alive) {
        done = k;
        break;
      }
    }
    // the part of the GRU rows that does not need p: attn_h(k-1) and the previous attention weights
    // (rows r, z, gi_n, gh_n on waves 0..3; context part = Eg . s(k-1) for r, z, gi_n)
    float gsum[NB];
    if (wave < 4) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float a = dot4(w_gah, *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane), 0.0f);
        if (wave < 3)
          for (int j = lane; j < Tp; j += 64) a = fmaf(v_E[((10 + wave) * NB + b) * Tp + j], v_s[b * Tp + j], a);
        gsum[b] = a;
      }
    }
    if (FC2L) {
      // every workgroup evaluates all 128 rows: thread (row, q) contracts a quarter of K, the quad adds up
      float acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 w = reinterpret_cast<const float4*>(v_fc2)[(i * 128 + frow) * 4 + fq];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(v_p1 + b * 256 + 4 * (fq + 4 * i)), acc[b]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float v = dpp_add<0x4E>(dpp_add<0xB1>(acc[b]));
        v = fmaxf(v + b_fc2r, 0.0f);
        if (p.drop_u2) v = pf_u2[b] >= kDropP ? v * (1.0f / (1.0f - kDropP)) : 0.0f;
        if (fq == 0) v_p[b * 128 + frow] = v;
      }
    } else {
      if (wave == 0 && wg < 128) {
        float pre[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
          pre[b] = wave_sum(dot4(w_fc2, *reinterpret_cast<const float4*>(v_p1 + b * 256 + 4 * lane), 0.0f));
        if (lane < NB && lane < B) {
          float v = fmaxf(sel(pre, lane) + b_fc2, 0.0f);
          if (p.drop_u2) v = pf_u2[0] >= kDropP ? v * (1.0f / (1.0f - kDropP)) : 0.0f;
          st_gran(p.g_p + wg * kSlot + lane, ep, v);
        }
      }
      // ================= stage 3: attention GRU unit wg ================================================================
      gather<(NB * 128 + kRT - 1) / kRT, 1>(sg_p, ep, lds, p.err, gave_up);
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float2 pv = *reinterpret_cast<const float2*>(v_p + b * 128 + 2 * lane);
        const float a = wave_sum(fmaf(w_gp.y, pv.y, fmaf(w_gp.x, pv.x, gsum[b])));
        if (lane == 0) red[wave * NB + b] = a;
      }
    }
    __syncthreads();
    if (tid < NB && tid < B) {
      const int b = tid;
      const float r_ = sigmoidf_(red[0 * NB + b] + b_gr);
      const float z_ = sigmoidf_(red[1 * NB + b] + b_gz);
      const float n_ = tanhf_(red[2 * NB + b] + b_gin + r_ * (red[3 * NB + b] + b_ghn));
      ahu = (1.0f - z_) * n_ + z_ * ahu;
      st_gran(p.g_ah + wg * kSlot + b, ep, ahu);
    }
    RSTAMP(4);
    // ================= stage 4: query projection row wg ================================================================
    gather<(NB * 256 + kRT - 1) / kRT, 1>(sg_ah, ep, lds, p.err, gave_up);
    __syncthreads();
    RSTAMP(5);
    if (wave == 0) {
      float pre[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) pre[b] = wave_sum(dot4(w_pq, *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane), 0.0f));
      if (lane < NB && lane < B) st_gran(p.g_pq + wg * kSlot + lane, ep, sel(pre, lane) + b_pq);
    }
    // LSTM1's attn_h and h1 parts do not need the attention: computed here, in the shadow of the next two seams
    float l1sum[NB], ysum[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float a = dot4(w_l1[0], *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane), 0.0f);
      a = dot4(w_l1[1], *reinterpret_cast<const float4*>(v_h1 + b * kRL + 4 * lane), a);
      l1sum[b] = dot4(w_l1[2], *reinterpret_cast<const float4*>(v_h1 + b * kRL + 256 + 4 * lane), a);
      ysum[b] = wave < 2 ? dot4(w_y, *reinterpret_cast<const float4*>(v_ah + b * 256 + 4 * lane), 0.0f) : 0.0f;
    }
    // ================= stage 5: energies of own positions ==============================================================
    RSTAMP(6);
    gather<(NB * 256 + kRT - 1) / kRT, 1>(sg_pq, ep, lds, p.err, gave_up);
    __syncthreads();
    RSTAMP(7);
#pragma unroll
    for (int i = 0; i < POS; i += 2) {   // two positions at a time, one per half of the workgroup
      if (pos_b[i] < 0) break;           // uniform: position i is the lower of the pair
      const int ii = min(i + half, POS - 1);
      const bool have = i + half < POS && sel(pos_b, ii) >= 0;
      const int b = have ? sel(pos_b, ii) : 0, j = sel(pos_j, ii);
      float e = have ? c_v * tanhf_(v_pq[b * 256 + ad] + sel(base_own, ii)) : 0.0f;
      e = wave_sum(e);
      if (lane == 0) red[wave] = e;        // waves 0-3: position i, waves 4-7: position i + 1
      __syncthreads();
      if ((tid & 255) == 0 && have) st_gran(p.g_u + wg * kSlot + ii, ep, (red[4 * half] + red[4 * half + 1]) + (red[4 * half + 2] + red[4 * half + 3]));
      __syncthreads();
    }
    // LSTM2's h2 part, in the shadow of the next seam
    float l2sum[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float a = dot4(w_l2[2], *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), 0.0f);
      l2sum[b] = dot4(w_l2[3], *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane), a);
    }
    // ================= stage 6: softmax (redundantly in every wave), LSTM1 units, x1 = y + h1 ==========================
    RSTAMP(8);
    gather<(POS + 1) / 2, 1>(sg_u, ep, lds, p.err, gave_up);
    __syncthreads();
    RSTAMP(9);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= B) break;
      // every wave normalises the row itself, in registers, and contracts its gate row's E with the weights right away;
      // wave b also keeps the row for the steps to come (attention state, att_out)
      const float* ub = v_u + b * T;
      float m = -INFINITY;
      for (int j = lane; j < T; j += 64) m = fmaxf(m, ub[j]);
      m = wave_max(m);
      float l = 0.0f;
      for (int j = lane; j < T; j += 64) l += __expf(ub[j] - m);
      const float inv = 1.0f / wave_sum(l);
      float a = l1sum[b], y = ysum[b];
      float* arow = p.att_out + ((size_t)b * n + k) * T;
      for (int j = lane; j < T; j += 64) {
        const float s = __expf(ub[j] - m) * inv;
        a = fmaf(v_E[(wave * NB + b) * Tp + j], s, a);
        if (wave < 2) y = fmaf(v_E[((8 + wave) * NB + b) * Tp + j], s, y);
        if (wave == b) {
          v_s[b * Tp + j] = s;
          v_cum[b * Tp + j] += s;
          if (wg == b) arow[j] = s;
        }
      }
      a = wave_sum(a);
      if (wave < 2) y = wave_sum(y);
      if (lane == 0) {
        red[wave * NB + b] = a;
        if (wave < 2) red[(8 + wave) * NB + b] = y;
      }
    }
    __syncthreads();
    if (tid < 2 * NB && eb < B) {
      const float ig = sigmoidf_(red[(4 * eu + 0) * NB + eb] + b_l1[0]);
      const float fg = sigmoidf_(red[(4 * eu + 1) * NB + eb] + b_l1[1]);
      const float gg = tanhf_(red[(4 * eu + 2) * NB + eb] + b_l1[2]);
      const float og = sigmoidf_(red[(4 * eu + 3) * NB + eb] + b_l1[3]);
      c1 = fg * c1 + ig * gg;
      h1 = og * tanhf_(c1);
      const float y = red[(8 + eu) * NB + eb] + b_y;
      st_gran(p.g_x1 + wg * kSlot + 2 * eb + eu, ep, y + h1);
      st_gran(p.g_h1 + wg * kSlot + 2 * eb + eu, ep + 1, h1);   // consumed by the next step's stage-2 sweep
    }
    RSTAMP(10);
    // location features of own positions for the NEXT step (conv over the new cum / att, L projection): off the chain
#pragma unroll
    for (int i = 0; i < POS; i += 2) {
      if (pos_b[i] < 0) break;
      const int ii = min(i + half, POS - 1);
      const bool have = i + half < POS && sel(pos_b, ii) >= 0;
      const int b = have ? sel(pos_b, ii) : 0, j = sel(pos_j, ii);
      float a = 0.0f;
      if (have) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kk = 8 * cpart_i + q, ch = kk >= 31 ? 1 : 0, tap = kk - 31 * ch;
          const int t = j + tap - 15;
          const float lv = (kk < 62 && t >= 0 && t < T) ? (ch ? v_s : v_cum)[b * Tp + t] : 0.0f;
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
      const float nb = sel(ep_own, ii) + pl;
#pragma unroll
      for (int q = 0; q < POS; ++q)
        if (have && ii == q) base_own[q] = nb;
      __syncthreads();
    }
    // ================= stage 7: LSTM2 units ============================================================================
    RSTAMP(11);
    gather<NB, 1>(sg_x1, ep, lds, p.err, gave_up);
    __syncthreads();
    RSTAMP(12);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float a = dot4(w_l2[0], *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), l2sum[b]);
      a = wave_sum(dot4(w_l2[1], *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane), a));
      if (lane == 0) red[wave * NB + b] = a;
    }
    __syncthreads();
    if (tid < 2 * NB && eb < B) {
      const float ig = sigmoidf_(red[(4 * eu + 0) * NB + eb] + b_l2[0]);
      const float fg = sigmoidf_(red[(4 * eu + 1) * NB + eb] + b_l2[1]);
      const float gg = tanhf_(red[(4 * eu + 2) * NB + eb] + b_l2[2]);
      const float og = sigmoidf_(red[(4 * eu + 3) * NB + eb] + b_l2[3]);
      c2 = fg * c2 + ig * gg;
      h2 = og * tanhf_(c2);
      st_gran(p.g_h2 + wg * kSlot + 2 * eb + eu, ep + 1, h2);   // consumed by the next step's stage 1
    }
    RSTAMP(13);
    __syncthreads();   // red and the vectors are reused by the next step
  }
  // ---- the last group's mel frames (no step follows to carry them) ---------------------------------------------------
  if (done == n) {
    gather<NB, 1>(sg_h2, (u32)n + 1, lds, p.err, gave_up);
    __syncthreads();
    for (int i = 0, j = wg; j < p.nmr; j += kRW, ++i) {
      if ((i & 7) != wave) continue;
      const float4 wa = *reinterpret_cast<const float4*>(p.mel_w + (size_t)j * kRL + 4 * lane);
      const float4 wb = *reinterpret_cast<const float4*>(p.mel_w + (size_t)j * kRL + 256 + 4 * lane);
      const int m = j / p.r, f = j - m * p.r;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 xa = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 4 * lane), xb = *reinterpret_cast<const float4*>(v_x1 + b * kRL + 256 + 4 * lane);
        const float4 ha = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 4 * lane), hb = *reinterpret_cast<const float4*>(v_h2 + b * kRL + 256 + 4 * lane);
        const float4 sa = make_float4(xa.x + ha.x, xa.y + ha.y, xa.z + ha.z, xa.w + ha.w);
        const float4 sb = make_float4(xb.x + hb.x, xb.y + hb.y, xb.z + hb.z, xb.w + hb.w);
        const float v = wave_sum(dot4(wb, sb, dot4(wa, sa, 0.0f)));
        if (b < B && lane == 0) p.mel_out[(size_t)b * p.mel_sb + (size_t)m * p.mel_sm + (size_t)(n - 1) * p.r + f] = v;
      }
    }
  }
  if (wg == 0 && tid == 0) *p.n_done = done;
}

// dynamic LDS of the kernel in floats (the carve at its top)
inline size_t resident_lds_floats(int NB, int T, bool fc2l) {
  const size_t Tp = (size_t)((T + 63) & ~63);
  return (size_t)NB * kRL * 2 + ((size_t)NB * 768 + 256) + (size_t)NB * 128 + (size_t)NB * 256 * 2 + (size_t)NB * Tp * 2 +
         13 * (size_t)NB * Tp + kRMaxBT + 16 * (size_t)NB + 128 + 2 * 8 * 32 + (fc2l ? 16 * 128 * 16 : 0);
}

}  // namespace
