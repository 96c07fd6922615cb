// The attention recurrence of the TEACHER-FORCED decoder loop as persistent launches of 16 .. 64 steps (B <= 64, T <= 128).
//
// With the prenet inputs known up front, the only loop-carried chain of Decoder.forward (reference tacotron.py:185-207) is
//     attention GRU(k) -> query projection(k) -> location-sensitive attention(k) -> context(k) -> GRU(k+1);
// rnn_input, the two LSTM cells and the mel projection only consume it.  The launch loop pays two kernel boundaries per
// step for that chain and fetches each row's encoder memories (2 x 128 KB at T = 128) from a cold L2 at every step.  Here
// every workgroup keeps its share of the step's operands for the WHOLE utterance and the stages hand over through tagged
// granule pairs (seam.h; tag = step + 1, arrays zeroed before the launch, every wait bounded):
//
//   workgroup i < 4 B      attention role (row i / 4, attention dims 64 (i % 4) .. + 63, ALL positions of the row): its slice of
//                          the encoder projection and of the encoder outputs, the L fragments, L bias, v and the conv taps
//                          in registers (160 per thread), a 64-row slice of the query weights in LDS (64 KB)
//   workgroup i < 64 RG    GRU role (16-row group i / 64, tile i % 64 = 4 units): the tile's weights in registers (40 per
//                          thread), arithmetic = LinJob<1,1,EpiGru> chunk for chunk
//
//   step k:  AT: location conv of step k (needs only the attention state of step k-1; runs while the contexts are on their way)
//            G:  sweep h(k-1) [H] and context(k-1) [C] of the 16 rows in one round trip -> MFMA -> gates -> publish h(k) [H
//                and, row-major, HT], store ah_all[k]
//            AT: projection of the location features (while the state is on its way) -> wait h(k) of the row [HT] -> query
//                slice (its own 64 dims: no exchange) -> tanh, v-dot: the slice's share of all 128 energies -> publish [U] ->
//                sweep the other three slices' shares -> energies -> chunk softmax statistics, scales, the row's new
//                attention / cumulative attention (kept in LDS by all four), scores of chunk i % 4 -> context slice (complete:
//                no exchange) -> publish [C], store ctx_all[k]
//
// Three hand-overs per step; no encoder memory traffic, no weight traffic, no launch on the chain.  ah_all / ctx_all
// (write-through stores) feed the LSTM stack, which runs as launches of its own beside the chain (taco_decoder.hip): what a
// chain workgroup fetches and polls per step goes through the same compute unit's memory path as the LSTM workgroups' fills.
// The shares are summed in the order of AttnJob's four waves, statistics / scales / context follow AttnJob's chunks of 32
// positions: the attention arithmetic is AttnJob's, bit for bit.
// Bit-compatibility: GRU, attention and combine repeat the launch path's arithmetic exactly; the query projection is summed
// in a different order (vector FMAs over four K quarters instead of the MFMA chunks), so states agree to rounding.
#pragma once

namespace {

constexpr int kCT = 256;                 // threads per workgroup
constexpr int kCMaxB = 64, kCMaxT = 128; // 4 B and 64 ceil(B / 16) workgroups <= 256; four chunks of 32 positions
constexpr int kCXS = 644;                // LDS row stride of the GRU's staged input [ctx 256 | prenet 128 | h 256] (floats)
constexpr u32 kCSpin = 1u << 21;   // polls (about 1 us each) before a wave gives up: seconds, not milliseconds -- the workgroups of a
                                    // launch become resident one by one while other streams' kernels still hold compute units

struct ChainParams {
  const float *enc_p, *enc_q;   // [B][T][256]: encoder projection (energies), encoder outputs (context)
  const float* pp;              // prenet outputs of all steps, interleaved [n][128/4][B][4]
  const float4* gru_wt;         // [64 tiles][40][64] fragments (Layer)
  const float* gru_bias;        // tile order
  const float *pq_w, *pq_b;     // [256][256] row-major, [256]
  const float *conv_t, *L_w, *L_b, *v_w;
  float* scores;                // scores[b * score_stride + k * T + t]
  long score_stride;
  float *ctx_all, *ah_all;      // [n][256/4][B][4]
  u64 *H, *U, *C;               // pairs: H [2][RG][128][16], U [B][4][64] (energy shares), C [B][4][32]
  u64* HT;                      // pairs: [2][B][128], the new state once more, row-major: what an attention workgroup polls (2 KB
                                // contiguous instead of 128 pieces 256 bytes apart)
  u32* err;
  float* state;                 // [workgroups][384]: the row's attention / cumulative attention between the launches of a call
  int B, T, n;
  int k0, k1;                   // steps of this launch
  u64* stamps;                  // diagnostics (WRNN_CHAIN_STAMPS): [4 steps][16] 100 MHz timestamps of workgroup 0, or null
};

constexpr size_t chain_lds_floats() {
  return 256 * 64 + 4 * 32 * kFS + 16 * 64 + 128 + 4 * 128 + 128 + 16 + 256 + 64 + 256 + 192 + 192 + 16 * kCXS + 4 * 64 * 4;
}

using u32x4c = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void st_pair_c(u64* p, u32 tag, float a, float b) {
  const u32x4c q = {__float_as_uint(a), tag, __float_as_uint(b), tag};
  // (s_nop: a store of more than 8 bytes needs one wait state before its data registers may be overwritten; the
  //  compiler's hazard pass does not look inside the asm statement)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void st_f4_wt(float* p, float a, float b, float c, float d) {   // write-through 16-byte store
  const u32x4c q = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(q) : "memory");
}

// Poll one pair per thread until both of its tags equal `tag`.  Bounded; a wave that has given up once does not wait again
// (the launch then drains in about one spin limit) and the error word makes the host poison the call's output.
template <int N, int SLEEP = 8>
__device__ __forceinline__ void poll_pairs(const u64* const (&a)[N], const bool (&live)[N], u32 tag, u32x4c (&v)[N], u32* err, bool& gave_up) {
  static_assert(N == 1, "one pair per thread (the 2 x 8-pair sweep of the GRU role is poll_hc)");
  for (u32 spins = 0;; ++spins) {
    // the load and its wait in ONE asm statement: between separate statements the compiler may copy the destination
    // register (this kernel spills to AGPRs) before the data has arrived.  Lanes that wait for nothing issue no load: the
    // spinning polls of 1024 waves are what the LSTM launches beside this kernel pay for (every 16-byte piece is a fabric
    // request of its own).
    if (live[0]) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(a[0]) : "memory");
    const bool ok = !live[0] || (v[0].y == tag && v[0].w == tag);
    if (__all(ok) || gave_up) break;
    if ((spins & 63) == 63 && *(volatile u32*)err) gave_up = true;   // somebody timed out: drain
    if (spins > kCSpin) {
      if ((threadIdx.x & 63) == 0) atomicOr(err, 1u);
      gave_up = true;
      break;
    }
    // back off: the compute unit's memory path is shared with the LSTM launches that run beside this kernel
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

// The GRU role's two sweeps in ONE round trip: 8 pairs of H and 8 pairs of C per thread, each group addressed from one base
// (the thread's pairs are 1 KB apart: immediate offsets -4096 .. 3072 around the fifth), all sixteen loads and their wait in
// one asm statement.  H (the previous state) has been complete for a whole step; C (the previous context) usually is by the
// time this workgroup gets here, otherwise the wave polls again.
__device__ __forceinline__ void poll_hc(const u64* hmid, const u64* cmid, bool hlive, int c_nlive, u32 tag, u32x4c (&hv)[8], u32x4c (&cv)[8],
                                        u32* err, bool& gave_up) {
  for (u32 spins = 0;; ++spins) {
    asm volatile(
        "global_load_dwordx4 %0, %16, off offset:-4096 sc1\n\tglobal_load_dwordx4 %1, %16, off offset:-3072 sc1\n\t"
        "global_load_dwordx4 %2, %16, off offset:-2048 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:-1024 sc1\n\t"
        "global_load_dwordx4 %4, %16, off sc1\n\tglobal_load_dwordx4 %5, %16, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %16, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %16, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %8, %17, off offset:-4096 sc1\n\tglobal_load_dwordx4 %9, %17, off offset:-3072 sc1\n\t"
        "global_load_dwordx4 %10, %17, off offset:-2048 sc1\n\tglobal_load_dwordx4 %11, %17, off offset:-1024 sc1\n\t"
        "global_load_dwordx4 %12, %17, off sc1\n\tglobal_load_dwordx4 %13, %17, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %14, %17, off offset:2048 sc1\n\tglobal_load_dwordx4 %15, %17, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(hv[0]), "=&v"(hv[1]), "=&v"(hv[2]), "=&v"(hv[3]), "=&v"(hv[4]), "=&v"(hv[5]), "=&v"(hv[6]), "=&v"(hv[7]),
          "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3]), "=&v"(cv[4]), "=&v"(cv[5]), "=&v"(cv[6]), "=&v"(cv[7])
        : "v"(hmid), "v"(cmid)
        : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ok &= !hlive || (hv[i].y == tag && hv[i].w == tag);
      ok &= i >= c_nlive || (cv[i].y == tag && cv[i].w == tag);
    }
    if (__all(ok) || gave_up) break;
    if ((spins & 63) == 63 && *(volatile u32*)err) gave_up = true;
    if (spins > kCSpin) {
      if ((threadIdx.x & 63) == 0) atomicOr(err, 1u);
      gave_up = true;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

#define CSTAMP(i)                                                                                       \
  do {                                                                                                  \
    if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0 && k >= 100 && k < 104)                         \
      p.stamps[(k - 100) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                                \
  } while (0)

__global__ __launch_bounds__(kCT) void chain_kernel(ChainParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wq = reinterpret_cast<float*>(smem);   // [256 k][64]: query weights of dims 64c .. 64c + 63, k-major
  float* f = wq + 256 * 64;                     // [4 waves][32][kFS] conv features of the wave's 32 positions
  float* red = f + 4 * 32 * kFS;                // [16 position groups][64] context partials of the slice
  float* uo = red + 16 * 64;                    // [128] this slice's share of the energies
  float* uall = uo + 128;                       // [4 slices][128]
  float* wsh = uall + 4 * 128;                  // [128] unnormalised weights, per chunk of 32 positions
  float* stats = wsh + 128;                     // [4 chunks][2]: chunk max, chunk sum
  float* hrow = stats + 16;                     // [256] the row's new attention state
  float* qs = hrow + 256;                       // [64] the query's slice
  float* qpart = qs + 64;                       // [4][64]
  float* st_att = qpart + 256;                  // [192]: position t at [16 + t]; zero outside [0, T)
  float* st_cum = st_att + 192;                 // [192]
  float* xs = st_cum + 192;                     // [16][kCXS] GRU input of the row group
  f32x4(*gred)[64] = reinterpret_cast<f32x4(*)[64]>(xs + 16 * kCXS);   // [4][64]

  const int blk = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, s = lane >> 4;
  const int B = p.B, T = p.T;
  const int RG = (B + 15) >> 4;
  const bool at_role = blk < 4 * B, g_role = blk < 64 * RG;
  // attention role
  const int b = min(blk >> 2, B - 1), c = blk & 3;   // row, 64-dim slice
  // GRU role
  const int rg = min(blk >> 6, RG - 1), tile = blk & 63;
  bool gave_up = false;

  // ---- set-up: operands that stay for the whole utterance
  // attention role: wave w owns positions 32 w .. 32 w + 31 of the row for the energies (dims 64c + 16n + 4s ..), thread
  // (tid / 16, tid % 16) owns positions 8 (tid / 16) .. + 7, dims 64c + 4 (tid % 16) .. + 3 for the context
  float4 ep[kMT][4], eq[kEP];
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) {
    const float* row = p.enc_p + ((size_t)b * T + min(32 * wave + 16 * mt + j, T - 1)) * kD + 64 * c + 4 * s;
#pragma unroll
    for (int n = 0; n < 4; ++n) ep[mt][n] = *reinterpret_cast<const float4*>(row + 16 * n);
  }
#pragma unroll
  for (int i = 0; i < kEP; ++i)
    eq[i] = *reinterpret_cast<const float4*>(p.enc_q + ((size_t)b * T + min(kEP * (tid >> 4) + i, T - 1)) * kD + 64 * c + 4 * (tid & 15));
  for (int i = tid; i < 256 * 64; i += kCT) {   // wq[k][dim] = W[64c + dim][k]: read along k (coalesced), written transposed
    const int dim = i >> 8, k = i & 255;
    wq[k * 64 + dim] = p.pq_w[(size_t)(64 * c + dim) * 256 + k];
  }
  // conv taps in the MFMA A layout, in registers (32): filter 16 h + j, tap index 4 q + s (62 taps, two zero rows)
  float av[2][16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int h = 0; h < 2; ++h) av[h][q] = 4 * q + s < 2 * kTaps ? p.conv_t[(4 * q + s) * kFilters + 16 * h + j] : 0.0f;
  for (int i = tid; i < 384; i += kCT) st_att[i] = p.k0 > 0 ? p.state[(size_t)blk * 384 + i] : 0.0f;   // st_att and st_cum
  // the GRU tile's weights (the wave's ten K chunks, 40 registers) and the small attention operands (L fragments, L bias, v:
  // 64 registers) stay in registers for the whole launch: every byte this workgroup fetches per step goes through the
  // compute unit's fill path, which is what bounds the LSTM launches that run beside it
  float4 w[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) w[q] = p.gru_wt[((size_t)tile * 40 + wave + 4 * q) * 64 + lane];
  float4 lf[4][2], vv[4], lb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float* lrow = p.L_w + (size_t)(64 * c + 16 * n + j) * kFilters + 8 * s;
    lf[n][0] = reinterpret_cast<const float4*>(lrow)[0];
    lf[n][1] = reinterpret_cast<const float4*>(lrow)[1];
    const int d0 = 64 * c + 16 * n + 4 * s;
    lb[n] = *reinterpret_cast<const float4*>(p.L_b + d0);
    vv[n] = *reinterpret_cast<const float4*>(p.v_w + d0);
  }
  const float qbias = p.pq_b[64 * c + (tid & 63)];
  const f32x4 gbias = *reinterpret_cast<const f32x4*>(p.gru_bias + tile * 16 + s * 4);
  __syncthreads();

  const int tid_setup = tid;
  for (int k = p.k0; k < p.k1; ++k) {
    const u32 tag = (u32)k + 1;
    // The thread index is laundered once per step: everything derived from it (dozens of addresses) would otherwise be
    // hoisted out of the loop and kept in registers for the whole launch (about 100 AGPRs of loop invariants); recomputing
    // them is a few hundred ALU operations per step, and the registers are what lets the LSTM launches' workgroups run
    // beside this one (512 per SIMD lane: this kernel + an LSTM wave of 144).
    int tid_l = tid_setup;
    asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63, j = lane & 15, s = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ================================================================== GRU role
    CSTAMP(0);
    // ---- attention role, location conv of the wave's 32 positions on the matrix cores (AttnJob phase 3; 2 x 2 tiles of 16).
    //      It depends only on the attention state of step k-1 (LDS: window position p + k of channel ch is
    //      st[ch][16 + t + k - 15]), so it runs HERE, while the contexts the GRU role is about to wait for are still on
    //      their way; the projection of the features follows the GRU role, while the new state is on its way.
    if (at_role) {
      float* fw = f + wave * (32 * kFS);
      float bw[2][16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int kk = 4 * q + s;
        const int ch = kk >= kTaps ? 1 : 0, kt = kk - kTaps * ch;
#pragma unroll
        for (int h = 0; h < 2; ++h) bw[h][q] = (ch ? st_att : st_cum)[32 * wave + 16 * h + j + kt + 1];
      }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ft][q], bw[pt][q], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ft][q + 1], bw[pt][q + 1], c1, 0, 0, 0);
          }
          *reinterpret_cast<f32x4*>(fw + (16 * pt + j) * kFS + 16 * ft + 4 * s) = c0 + c1;
        }
    }
    if (g_role) {
      // prenet outputs of the step: 16 rows x 32 quads
      const float* ppk = p.pp + (size_t)k * B * kFc2 + (size_t)min(16 * rg + (tid & 15), B - 1) * 4;
      const float4 pv0 = *reinterpret_cast<const float4*>(ppk + (size_t)(tid >> 4) * B * 4);
      const float4 pv1 = *reinterpret_cast<const float4*>(ppk + (size_t)((tid >> 4) + 16) * B * 4);
      if (k > 0) {
        // h(k-1) and context(k-1) of the 16 rows in one round trip (poll_hc).  Thread's pairs: idx = 512 wave + 64 i + lane.
        //   H (of the row group): pair idx = (unit pair idx >> 4, row idx & 15): row lane & 15, unit pair 32 wave + 4 i + lane / 16
        //   C (of the row group): pair idx = (row idx >> 7, pair idx & 127): row 4 wave + i / 2, pair 64 (i & 1) + lane
        // Rows past the batch are never published: H is masked per lane, C per pair (rows that exist come first).
        {
          const u64* hb = p.H + ((size_t)((k - 1) & 1) * RG + rg) * 128 * 16 * 2;
          const u64* cb = p.C + (size_t)rg * 16 * 128 * 2;
          const bool hlive = 16 * rg + (lane & 15) < B;
          const int c_nlive = max(0, min(8, 2 * (B - 16 * rg - 4 * wave)));
          u32x4c hv[8], cv[8];
          poll_hc(hb + 2 * (512 * wave + 256 + lane), cb + 2 * (512 * wave + 256 + lane), hlive, c_nlive, tag - 1, hv, cv, p.err, gave_up);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float2 hh = make_float2(__uint_as_float(hv[i].x), __uint_as_float(hv[i].z));
            float2 cc2 = make_float2(__uint_as_float(cv[i].x), __uint_as_float(cv[i].z));
            if (!hlive) hh = make_float2(0.f, 0.f);
            if (i >= c_nlive) cc2 = make_float2(0.f, 0.f);
            *reinterpret_cast<float2*>(xs + (lane & 15) * kCXS + 384 + 2 * (32 * wave + 4 * i + (lane >> 4))) = hh;
            *reinterpret_cast<float2*>(xs + (4 * wave + (i >> 1)) * kCXS + 2 * (64 * (i & 1) + lane)) = cc2;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int idx = tid + kCT * i, r = idx & 15, up = idx >> 4;
          *reinterpret_cast<float2*>(xs + r * kCXS + 384 + 2 * up) = make_float2(0.f, 0.f);
          *reinterpret_cast<float2*>(xs + r * kCXS + 2 * up) = make_float2(0.f, 0.f);
        }
      }
      CSTAMP(1);
      *reinterpret_cast<float4*>(xs + (tid & 15) * kCXS + 256 + 4 * (tid >> 4)) = pv0;
      *reinterpret_cast<float4*>(xs + (tid & 15) * kCXS + 256 + 4 * ((tid >> 4) + 16)) = pv1;
      __syncthreads();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 10; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(xs + j * kCXS + (wave + 4 * q) * 16 + s * 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].x, x.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].y, x.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].z, x.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].w, x.w, acc, 0, 0, 0);
      }
      gred[wave][lane] = acc;
      lds_barrier();
      if (wave == 0) {   // lane: row j, unit 4 tile + s; gates as EpiGru::apply
        const f32x4 v = (gred[0][lane] + gred[1][lane]) + (gred[2][lane] + gred[3][lane]);
        const float hp = xs[j * kCXS + 384 + tile * 4 + s];
        const float r = sigmoidf_(v[0] + gbias[0]);
        const float z = sigmoidf_(v[1] + gbias[1]);
        const float nn = tanhf_(v[2] + gbias[2] + r * (v[3] + gbias[3]));
        const float h = (1.0f - z) * nn + z * hp;
        const float h1 = __shfl_down(h, 16, 64), h2 = __shfl_down(h, 32, 64), h3 = __shfl_down(h, 48, 64);
        const bool rlive = 16 * rg + j < B;
        if (rlive && (s & 1) == 0) {
          st_pair_c(p.H + (((size_t)(k & 1) * RG + rg) * 128 * 16 + (size_t)(tile * 2 + (s >> 1)) * 16 + j) * 2, tag, h, h1);
          st_pair_c(p.HT + (((size_t)(k & 1) * B + 16 * rg + j) * 128 + tile * 2 + (s >> 1)) * 2, tag, h, h1);
        }
        if (rlive && s == 0)
          st_f4_wt(p.ah_all + (size_t)k * B * kD + ((size_t)tile * B + 16 * rg + j) * 4, h, h1, h2, h3);
      }
      __syncthreads();   // xs / gred are free again
      CSTAMP(2);
    }
    // ================================================================== attention role
    if (at_role) {
      // Workgroup (row b, slice c) computes, for ALL positions of the row, the energies' share of its 64 attention dims and
      // the context's 64 dims: the query slice it needs is its own (no exchange), the context slice it produces is complete
      // (no exchange); what the row's four workgroups exchange is their shares of the 128 energies.  Shares are summed in
      // the order of AttnJob's four waves, chunk statistics / scales / context follow AttnJob's chunks of 32 positions, so
      // the arithmetic is AttnJob's.
      //
      float* fw = f + wave * (32 * kFS);
      // ---- L projection (AttnJob phase 5): wave-local, the wave reads only its own features
      f32x4 acc[kMT][4];
      {
        float bv[kMT][8];
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
          const float4 b0 = *reinterpret_cast<const float4*>(fw + (16 * mt + j) * kFS + 8 * s);
          const float4 b1 = *reinterpret_cast<const float4*>(fw + (16 * mt + j) * kFS + 8 * s + 4);
          bv[mt][0] = b0.x, bv[mt][1] = b0.y, bv[mt][2] = b0.z, bv[mt][3] = b0.w;
          bv[mt][4] = b1.x, bv[mt][5] = b1.y, bv[mt][6] = b1.z, bv[mt][7] = b1.w;
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[mt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const float avv = m < 4 ? (m == 0 ? lf[n][0].x : m == 1 ? lf[n][0].y : m == 2 ? lf[n][0].z : lf[n][0].w)
                                    : (m == 4 ? lf[n][1].x : m == 5 ? lf[n][1].y : m == 6 ? lf[n][1].z : lf[n][1].w);
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
              acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(avv, bv[mt][m], acc[mt][n], 0, 0, 0);
          }
      }
      CSTAMP(3);
      // ---- the row's new state: unit pair tid of the row-major copy
      {
        const u64* a[1] = {p.HT + (((size_t)(k & 1) * B + b) * 128 + (tid & 127)) * 2};
        u32x4c v[1];
        const bool lv[1] = {tid < 128};
        poll_pairs<1>(a, lv, tag, v, p.err, gave_up);
        if (tid < 128) *reinterpret_cast<float2*>(hrow + 2 * tid) = make_float2(__uint_as_float(v[0].x), __uint_as_float(v[0].z));
      }
      __syncthreads();
      CSTAMP(4);
      // ---- query slice: thread = (dim tid & 63, K quarter tid >> 6)
      {
        const int dim = tid & 63, kq = tid >> 6;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 4) {
          const float4 h4 = *reinterpret_cast<const float4*>(hrow + 64 * kq + kk);
          const float* wr = wq + (size_t)(64 * kq + kk) * 64 + dim;
          a0 = fmaf(wr[0], h4.x, a0), a1 = fmaf(wr[64], h4.y, a1), a0 = fmaf(wr[128], h4.z, a0), a1 = fmaf(wr[192], h4.w, a1);
        }
        qpart[kq * 64 + dim] = a0 + a1;
      }
      lds_barrier();
      if (tid < 64) qs[tid] = qbias + ((qpart[tid] + qpart[64 + tid]) + (qpart[128 + tid] + qpart[192 + tid]));
      lds_barrier();
      CSTAMP(5);
      // ---- tanh, v-dot: this slice's share of the energy of positions 32 wave + 16 mt + j
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        float a = 0.0f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float4 q4 = *reinterpret_cast<const float4*>(qs + 16 * n + 4 * s);
          const float bx = q4.x + lb[n].x, by = q4.y + lb[n].y, bz = q4.z + lb[n].z, bw2 = q4.w + lb[n].w;
          a = fmaf(vv[n].x, tanhf_(acc[mt][n][0] + bx + ep[mt][n].x), a);
          a = fmaf(vv[n].y, tanhf_(acc[mt][n][1] + by + ep[mt][n].y), a);
          a = fmaf(vv[n].z, tanhf_(acc[mt][n][2] + bz + ep[mt][n].z), a);
          a = fmaf(vv[n].w, tanhf_(acc[mt][n][3] + bw2 + ep[mt][n].w), a);
        }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (s == 0) uo[32 * wave + 16 * mt + j] = a, uall[c * 128 + 32 * wave + 16 * mt + j] = a;
      }
      lds_barrier();
      CSTAMP(6);
      // ---- exchange the shares: publish 64 pairs (one store instruction), sweep the other three slices' 3 x 64
      if (tid < 64) st_pair_c(p.U + ((size_t)(b * 4 + c) * 64 + tid) * 2, tag, uo[2 * tid], uo[2 * tid + 1]);
      {
        const int which = min(tid >> 6, 2), c2 = (c + 1 + which) & 3, pi = tid & 63;
        const u64* a[1] = {p.U + ((size_t)(b * 4 + c2) * 64 + pi) * 2};
        u32x4c v[1];
        const bool lv[1] = {tid < 192};
        poll_pairs<1>(a, lv, tag, v, p.err, gave_up);
        if (tid < 192) *reinterpret_cast<float2*>(uall + c2 * 128 + 2 * pi) = make_float2(__uint_as_float(v[0].x), __uint_as_float(v[0].z));
      }
      __syncthreads();
      CSTAMP(7);
      // ---- chunk softmax statistics (AttnJob phase 6): wave w does chunk w
      {
        const int tt = lane & (kAT - 1), t = 32 * wave + tt;
        const bool ok = t < T;
        const float u = (uall[t] + uall[128 + t]) + (uall[256 + t] + uall[384 + t]);
        float m = ok ? u : -INFINITY;
#pragma unroll
        for (int off = kAT / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kAT));
        const float w = ok ? __expf(u - m) : 0.0f;
        float l = w;
#pragma unroll
        for (int off = kAT / 2; off >= 1; off >>= 1) l += __shfl_xor(l, off, kAT);
        if (lane < kAT) wsh[t] = w;
        if (lane == 0) stats[2 * wave] = m, stats[2 * wave + 1] = l;
      }
      lds_barrier();
      // ---- context partials of the slice (AttnJob phase 7): thread = (8 positions, 4 dims)
      {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kEP; ++i) {
          const float w = wsh[kEP * (tid >> 4) + i];
          a.x = fmaf(w, eq[i].x, a.x), a.y = fmaf(w, eq[i].y, a.y), a.z = fmaf(w, eq[i].z, a.z), a.w = fmaf(w, eq[i].w, a.w);
        }
        *reinterpret_cast<float4*>(red + (tid >> 4) * 64 + 4 * (tid & 15)) = a;
      }
      lds_barrier();
      CSTAMP(8);
      // ---- combine (AttnJob phase 9)
      {
        float2 ml[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ml[cc] = make_float2(stats[2 * cc], stats[2 * cc + 1]);
        float scale[4];
        chunk_scales<4>(ml, 4, scale);
        if (tid < 128) {   // the row's new attention and cumulative attention (kept by all four workgroups of the row)
          const int cc = tid >> 5;
          float s0 = scale[0], s1 = scale[1], s2 = scale[2], s3 = scale[3];   // (opaque values: a dynamically indexed
          asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));          //  array would go to scratch memory)
          const float pr = mul_rn(wsh[tid], (cc & 2) ? ((cc & 1) ? s3 : s2) : ((cc & 1) ? s1 : s0));
          if (tid < T) {
            st_att[16 + tid] = pr;
            st_cum[16 + tid] += pr;
            if (cc == c) p.scores[(size_t)b * p.score_stride + (size_t)k * T + tid] = pr;
          }
        } else if (tid < 160) {   // context dims 64c + 2i, + 1: per chunk (r0 + r1) + (r2 + r3) over its four position groups,
                                  // then the fma chain over the chunks; lane pairs make the 16-byte store of ctx_all
          const int i = tid - 128;
          float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const float* r4 = red + (4 * cc) * 64 + 2 * i;
            const float v0 = (r4[0] + r4[64]) + (r4[128] + r4[192]);
            const float v1 = (r4[1] + r4[65]) + (r4[129] + r4[193]);
            a0 = fmaf(v0, scale[cc], a0), a1 = fmaf(v1, scale[cc], a1);
          }
          st_pair_c(p.C + ((size_t)(b * 4 + c) * 32 + i) * 2, tag, a0, a1);
          const float b0 = __shfl_down(a0, 1, 64), b1 = __shfl_down(a1, 1, 64);
          if ((i & 1) == 0) st_f4_wt(p.ctx_all + (size_t)k * B * kD + ((size_t)(16 * c + (i >> 1)) * B + b) * 4, a0, a1, b0, b1);
        }
      }
      __syncthreads();
      CSTAMP(9);
    }
  }
  for (int i = tid; i < 384; i += kCT) p.state[(size_t)blk * 384 + i] = st_att[i];
}

}  // namespace
