// The attention recurrence of the TEACHER-FORCED decoder loop as persistent launches of 16 .. 64 steps (B <= 64 at T <= 128, B <= 32 at T <= 256).
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
//
// The description above is chain_kernel<4> (NS = 4 slices of 64 dims per row, T <= 128).  chain_kernel<8> (128 < T <= 256,
// B <= 32) cuts a row into NS = 8 slices of 32 dims over 256 positions: the same registers per thread, a 32 KB query slice,
// eight shares per energy (summed slice 0 .. 7: pairs of AttnJob's wave order, so energies agree to rounding there too).
#pragma once

namespace {

constexpr int kCT = 256;                 // threads per workgroup
constexpr int kCMaxB = 64, kCMaxT = 256; // NS B and 64 ceil(B / 16) workgroups <= 256, NS = 4 slices (T <= 128) or 8 (T <= 256)
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
  u64 *H, *U, *C;               // pairs: H [2][RG][128][16], U [B][NS][16 NS] (energy shares), C [B][128] (by context dim)
  u64* HT;                      // pairs: [2][B][128], the new state once more, row-major: what an attention workgroup polls (2 KB
                                // contiguous instead of 128 pieces 256 bytes apart)
  u32* err;
  float* state;                 // [workgroups][kCState]: the row's attention / cumulative attention between the launches of a call
  int B, T, n;
  int k0, k1;                   // steps of this launch
  int stamp_k;                  // first of the four stamped steps
  int u_plain;                  // plain stores for the energy shares where a row's workgroups are seen to share an XCD
  u32* xcc;                     // [256]: (first step of the launch + 1) << 8 | XCD number + 1 of every workgroup
  u64* stamps;                  // diagnostics (WRNN_CHAIN_STAMPS): [4 steps][16] 100 MHz timestamps of workgroup 0, or null
};

constexpr int kCState = 2 * (16 + kCMaxT + 48);   // floats per workgroup in ChainParams::state
// NS slices per row: DS = 256 / NS attention dims per workgroup, TM = 32 NS positions at most
constexpr size_t chain_lds_floats(int NS) {
  return 256 * (256 / NS) + 4 * (8 * NS) * kFS + 1024 + 32 * NS + NS * 32 * NS + 32 * NS + 16 + 256 + 256 / NS + 256 + 2 * (16 + 32 * NS + 48) +
         16 * kCXS + 4 * 64 * 4;
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
    if (!wrnn::keep_polling(spins, kCSpin, err, gave_up)) break;   // somebody timed out (drain), or this wave does now
    // back off: the compute unit's memory path is shared with the LSTM launches that run beside this kernel
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

// Four pairs per thread in one round trip (the share exchange with eight slices per row).
__device__ __forceinline__ void poll_pairs4(const u64* const (&a)[4], const bool (&live)[4], u32 tag, u32x4c (&v)[4], u32* err, bool& gave_up) {
  for (u32 spins = 0;; ++spins) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
        : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) ok &= !live[i] || (v[i].y == tag && v[i].w == tag);
    if (__all(ok) || gave_up) break;
    if (!wrnn::keep_polling(spins, kCSpin, err, gave_up)) break;
    __builtin_amdgcn_s_sleep(8);
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
    if (!wrnn::keep_polling(spins, kCSpin, err, gave_up)) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

#define CSTAMP(i)                                                                                       \
  do {                                                                                                  \
    if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0 && k >= p.stamp_k && k < p.stamp_k + 4)         \
      p.stamps[(k - p.stamp_k) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                          \
  } while (0)

template <int NS>
__global__ __launch_bounds__(kCT) void chain_kernel(ChainParams p) {
  static_assert(NS == 4 || NS == 8, "four 64-dim slices (T <= 128) or eight 32-dim slices (T <= 256) per row");
  constexpr int DS = 256 / NS;      // attention dims per workgroup
  constexpr int NT = DS / 16;       // 16-dim tiles of the slice
  constexpr int TM = 32 * NS;       // positions covered (NS chunks of 32)
  constexpr int PW = TM / 4;        // positions per wave
  constexpr int MT = PW / 16;       // 16-position tiles per wave
  constexpr int DQ = DS / 4;        // dim quads of the slice: context thread = (position group tid / DQ, quad tid % DQ)
  constexpr int KQ = kCT / DS;      // K parts of the query projection: thread = (dim tid % DS, part tid / DS)
  constexpr int SA = 16 + TM + 48;  // floats per state array
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* wq = reinterpret_cast<float*>(smem);   // [256 k][DS]: query weights of the slice, k-major
  float* f = wq + 256 * DS;                     // [4 waves][PW][kFS] conv features of the wave's positions
  float* red = f + 4 * PW * kFS;                // [TM / 8 position groups][DS] context partials of the slice (1024 floats)
  float* uo = red + 1024;                       // [TM] this slice's share of the energies
  float* uall = uo + TM;                        // [NS slices][TM]
  float* wsh = uall + NS * TM;                  // [TM] unnormalised weights, per chunk of 32 positions
  float* stats = wsh + TM;                      // [NS chunks][2]: chunk max, chunk sum
  float* hrow = stats + 16;                     // [256] the row's new attention state
  float* qs = hrow + 256;                       // [DS] the query's slice
  float* qpart = qs + DS;                       // [KQ][DS]
  float* st_att = qpart + 256;                  // [SA]: position t at [16 + t]; zero outside [0, T)
  float* st_cum = st_att + SA;                  // [SA]
  float* xs = st_cum + SA;                      // [16][kCXS] GRU input of the row group
  f32x4(*gred)[64] = reinterpret_cast<f32x4(*)[64]>(xs + 16 * kCXS);   // [4][64]

  const int blk = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, s = lane >> 4;
  const int B = p.B, T = p.T;
  const int RG = (B + 15) >> 4;
  // attention role: row b = (blk % 8) + 8 (blk / (8 NS)), slice c = (blk / 8) % NS -- the NS workgroups of a row are 8 block
  // indices apart, i.e. on the same XCD (workgroups are dealt to the eight XCDs in turn): the exchange of their energy shares
  // is then served by that XCD's L2.  Nothing depends on the placement but the time.
  const int b_raw = (blk & 7) + 8 * (blk / (8 * NS));
  const bool at_role = b_raw < B, g_role = blk < 64 * RG;
  const int b = min(b_raw, B - 1), c = (blk >> 3) % NS;   // row, slice of DS attention dims
  // GRU role
  const int rg = min(blk >> 6, RG - 1), tile = blk & 63;
  bool gave_up = false;

  // ---- The energy shares of a row travel between its NS workgroups only.  By the block -> row mapping above these share an XCD
  // and the exchange is served by its L2; with PLAIN stores the pairs never leave that L2 (13.81 -> 13.62 us per decoder step
  // at B = 64) -- allowed only once the row's workgroups have SEEN that they do share an XCD: every workgroup publishes its XCD
  // number under the launch's first step, a row's workgroups compare.  Otherwise write-through stores, as for everything else.
  bool u_plain = false;
  if (p.u_plain) {
    unsigned my;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
    const unsigned word = ((unsigned)(p.k0 + 1) << 8) | ((my & 15u) + 1u);
    if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.xcc + blk), "v"(word) : "memory");
    if (wave == 0) {
      const unsigned* a = p.xcc + (b & 7) + 8 * ((lane % NS) + NS * (b >> 3));   // slice lane % NS of row b
      unsigned v = 0;
      for (u32 spins = 0;; ++spins) {
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
        if (__all((v >> 8) == (unsigned)(p.k0 + 1)) || gave_up || !at_role) break;
        if (!wrnn::keep_polling(spins, kCSpin, p.err, gave_up)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      const bool same = __all(v == word) && !gave_up && at_role;
      if (lane == 0) qs[0] = same ? 1.0f : 0.0f;
    }
    __syncthreads();
    u_plain = __builtin_amdgcn_readfirstlane(__float_as_uint(qs[0])) != 0u;
    __syncthreads();
  }

  // ---- set-up: operands that stay for the whole utterance
  // attention role: wave w owns positions PW w .. PW w + PW - 1 of the row for the energies (dims DS c + 16 n + 4 s ..), thread
  // (tid / DQ, tid % DQ) owns positions 8 (tid / DQ) .. + 7, dims DS c + 4 (tid % DQ) .. + 3 for the context
  float4 ep[MT][NT], eq[kEP];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float* row = p.enc_p + ((size_t)b * T + min(PW * wave + 16 * mt + j, T - 1)) * kD + DS * c + 4 * s;
#pragma unroll
    for (int n = 0; n < NT; ++n) ep[mt][n] = *reinterpret_cast<const float4*>(row + 16 * n);
  }
#pragma unroll
  for (int i = 0; i < kEP; ++i)
    eq[i] = *reinterpret_cast<const float4*>(p.enc_q + ((size_t)b * T + min(kEP * (tid / DQ) + i, T - 1)) * kD + DS * c + 4 * (tid % DQ));
  for (int i = tid; i < 256 * DS; i += kCT) {   // wq[k][dim] = W[DS c + dim][k]: read along k (coalesced), written transposed
    const int dim = i >> 8, k = i & 255;
    wq[k * DS + dim] = p.pq_w[(size_t)(DS * c + dim) * 256 + k];
  }
  // conv taps in the MFMA A layout, in registers (32): filter 16 h + j, tap index 4 q + s (62 taps, two zero rows)
  float av[2][16];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int h = 0; h < 2; ++h) av[h][q] = 4 * q + s < 2 * kTaps ? p.conv_t[(4 * q + s) * kFilters + 16 * h + j] : 0.0f;
  for (int i = tid; i < 2 * SA; i += kCT) st_att[i] = p.k0 > 0 ? p.state[(size_t)blk * kCState + i] : 0.0f;   // st_att and st_cum
  // the GRU tile's weights (the wave's ten K chunks, 40 registers) and the small attention operands (L fragments, L bias, v:
  // 64 registers) stay in registers for the whole launch: every byte this workgroup fetches per step goes through the
  // compute unit's fill path, which is what bounds the LSTM launches that run beside it
  float4 w[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) w[q] = p.gru_wt[((size_t)tile * 40 + wave + 4 * q) * 64 + lane];
  float4 lf[NT][2], vv[NT], lb[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const float* lrow = p.L_w + (size_t)(DS * c + 16 * n + j) * kFilters + 8 * s;
    lf[n][0] = reinterpret_cast<const float4*>(lrow)[0];
    lf[n][1] = reinterpret_cast<const float4*>(lrow)[1];
    const int d0 = DS * c + 16 * n + 4 * s;
    lb[n] = *reinterpret_cast<const float4*>(p.L_b + d0);
    vv[n] = *reinterpret_cast<const float4*>(p.v_w + d0);
  }
  const float qbias = p.pq_b[DS * c + tid % DS];
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
      float* fw = f + wave * (PW * kFS);
      float bw[MT][16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int kk = 4 * q + s;
        const int ch = kk >= kTaps ? 1 : 0, kt = kk - kTaps * ch;
#pragma unroll
        for (int h = 0; h < MT; ++h) bw[h][q] = (ch ? st_att : st_cum)[PW * wave + 16 * h + j + kt + 1];
      }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int pt = 0; pt < MT; ++pt) {
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
      float* fw = f + wave * (PW * kFS);
      // ---- L projection (AttnJob phase 5): wave-local, the wave reads only its own features
      f32x4 acc[MT][NT];
      {
        float bv[MT][8];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4 b0 = *reinterpret_cast<const float4*>(fw + (16 * mt + j) * kFS + 8 * s);
          const float4 b1 = *reinterpret_cast<const float4*>(fw + (16 * mt + j) * kFS + 8 * s + 4);
          bv[mt][0] = b0.x, bv[mt][1] = b0.y, bv[mt][2] = b0.z, bv[mt][3] = b0.w;
          bv[mt][4] = b1.x, bv[mt][5] = b1.y, bv[mt][6] = b1.z, bv[mt][7] = b1.w;
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[mt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float avv = m < 4 ? (m == 0 ? lf[n][0].x : m == 1 ? lf[n][0].y : m == 2 ? lf[n][0].z : lf[n][0].w)
                                    : (m == 4 ? lf[n][1].x : m == 5 ? lf[n][1].y : m == 6 ? lf[n][1].z : lf[n][1].w);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
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
      // ---- query slice: thread = (dim tid % DS, K part tid / DS of 256 / KQ)
      {
        const int dim = tid % DS, kq = tid / DS;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 256 / KQ; kk += 4) {
          const float4 h4 = *reinterpret_cast<const float4*>(hrow + (256 / KQ) * kq + kk);
          const float* wr = wq + (size_t)((256 / KQ) * kq + kk) * DS + dim;
          a0 = fmaf(wr[0], h4.x, a0), a1 = fmaf(wr[DS], h4.y, a1), a0 = fmaf(wr[2 * DS], h4.z, a0), a1 = fmaf(wr[3 * DS], h4.w, a1);
        }
        qpart[kq * DS + dim] = a0 + a1;
      }
      lds_barrier();
      if (tid < DS) {
        float q = (qpart[tid] + qpart[DS + tid]) + (qpart[2 * DS + tid] + qpart[3 * DS + tid]);
        if constexpr (KQ == 8) q += (qpart[4 * DS + tid] + qpart[5 * DS + tid]) + (qpart[6 * DS + tid] + qpart[7 * DS + tid]);
        qs[tid] = qbias + q;
      }
      lds_barrier();
      CSTAMP(5);
      // ---- tanh, v-dot: this slice's share of the energy of positions PW wave + 16 mt + j
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float a = 0.0f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 q4 = *reinterpret_cast<const float4*>(qs + 16 * n + 4 * s);
          const float bx = q4.x + lb[n].x, by = q4.y + lb[n].y, bz = q4.z + lb[n].z, bw2 = q4.w + lb[n].w;
          a = fmaf(vv[n].x, tanhf_(acc[mt][n][0] + bx + ep[mt][n].x), a);
          a = fmaf(vv[n].y, tanhf_(acc[mt][n][1] + by + ep[mt][n].y), a);
          a = fmaf(vv[n].z, tanhf_(acc[mt][n][2] + bz + ep[mt][n].z), a);
          a = fmaf(vv[n].w, tanhf_(acc[mt][n][3] + bw2 + ep[mt][n].w), a);
        }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (s == 0) uo[PW * wave + 16 * mt + j] = a, uall[c * TM + PW * wave + 16 * mt + j] = a;
      }
      lds_barrier();
      CSTAMP(6);
      // ---- exchange the shares: publish TM / 2 pairs (one store instruction per wave), sweep the other slices' (NS - 1) TM / 2
      if (tid < TM / 2) {
        u64* const dst = p.U + ((size_t)(b * NS + c) * (TM / 2) + tid) * 2;
        if (u_plain) {   // the row's slices share an XCD (checked at the start of the launch): the pair stays in its L2
          const u32x4c q = {__float_as_uint(uo[2 * tid]), tag, __float_as_uint(uo[2 * tid + 1]), tag};
          asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(dst), "v"(q) : "memory");
        } else {
          st_pair_c(dst, tag, uo[2 * tid], uo[2 * tid + 1]);
        }
      }
      if constexpr (NS == 4) {
        const int which = min(tid >> 6, 2), c2 = (c + 1 + which) & 3, pi = tid & 63;
        const u64* a[1] = {p.U + ((size_t)(b * 4 + c2) * 64 + pi) * 2};
        u32x4c v[1];
        const bool lv[1] = {tid < 192};
        poll_pairs<1>(a, lv, tag, v, p.err, gave_up);
        if (tid < 192) *reinterpret_cast<float2*>(uall + c2 * 128 + 2 * pi) = make_float2(__uint_as_float(v[0].x), __uint_as_float(v[0].z));
      } else {
        // 7 slices x 128 pairs: thread (half tid >> 7, pair tid & 127) sweeps slices c + 1 + 2 i + half, i = 0 .. 3 (the
        // eighth, slice c itself, is masked): four pairs per thread in one round trip
        const int half = tid >> 7, pi = tid & 127;
        const u64* a4[4];
        bool l4[4];
        int c4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c4[i] = (c + 1 + 2 * i + half) & 7;
          l4[i] = c4[i] != c;
          a4[i] = p.U + ((size_t)(b * 8 + c4[i]) * 128 + pi) * 2;
        }
        u32x4c v4[4];
        poll_pairs4(a4, l4, tag, v4, p.err, gave_up);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (l4[i]) *reinterpret_cast<float2*>(uall + c4[i] * TM + 2 * pi) = make_float2(__uint_as_float(v4[i].x), __uint_as_float(v4[i].z));
      }
      __syncthreads();
      CSTAMP(7);
      // ---- chunk softmax statistics (AttnJob phase 6): wave w does the chunks of its PW positions (one per 32 lanes)
      {
        const int t = PW * wave + (lane & (PW - 1));
        const bool ok = t < T;
        float u = (uall[t] + uall[TM + t]) + (uall[2 * TM + t] + uall[3 * TM + t]);
        if constexpr (NS == 8) u += (uall[4 * TM + t] + uall[5 * TM + t]) + (uall[6 * TM + t] + uall[7 * TM + t]);
        float m = ok ? u : -INFINITY;
#pragma unroll
        for (int off = kAT / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kAT));
        const float w = ok ? __expf(u - m) : 0.0f;
        float l = w;
#pragma unroll
        for (int off = kAT / 2; off >= 1; off >>= 1) l += __shfl_xor(l, off, kAT);
        if (lane < PW) wsh[t] = w;
        if (lane < PW && (lane & 31) == 0) stats[2 * (t >> 5)] = m, stats[2 * (t >> 5) + 1] = l;
      }
      lds_barrier();
      // ---- context partials of the slice (AttnJob phase 7): thread = (8 positions, 4 dims)
      {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < kEP; ++i) {
          const float w = wsh[kEP * (tid / DQ) + i];
          a.x = fmaf(w, eq[i].x, a.x), a.y = fmaf(w, eq[i].y, a.y), a.z = fmaf(w, eq[i].z, a.z), a.w = fmaf(w, eq[i].w, a.w);
        }
        *reinterpret_cast<float4*>(red + (tid / DQ) * DS + 4 * (tid % DQ)) = a;
      }
      lds_barrier();
      CSTAMP(8);
      // ---- combine (AttnJob phase 9)
      {
        float2 ml[NS];
#pragma unroll
        for (int cc = 0; cc < NS; ++cc) ml[cc] = make_float2(stats[2 * cc], stats[2 * cc + 1]);
        float scale[NS];
        chunk_scales<NS>(ml, NS, scale);
        constexpr int CT0 = NS == 4 ? 128 : 0;   // first thread of the context part (NS = 4: beside the attention part)
        if (tid < TM) {   // the row's new attention and cumulative attention (kept by all workgroups of the row)
          const int cc = tid >> 5;
          float sc = scale[0];   // (selects on opaque values: a dynamically indexed array would go to scratch memory)
#pragma unroll
          for (int q = 1; q < NS; ++q) {
            float sq = scale[q];
            asm volatile("" : "+v"(sq));
            sc = cc == q ? sq : sc;
          }
          const float pr = mul_rn(wsh[tid], sc);
          if (tid < T) {
            st_att[16 + tid] = pr;
            st_cum[16 + tid] += pr;
            if (cc == c) p.scores[(size_t)b * p.score_stride + (size_t)k * T + tid] = pr;
          }
        }
        if (tid >= CT0 && tid < CT0 + DS / 2) {   // context dims DS c + 2i, + 1: per chunk (r0 + r1) + (r2 + r3) over its four position
                                                  // groups, then the fma chain over the chunks; lane pairs make the 16-byte store of ctx_all
          const int i = tid - CT0;
          float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
          for (int cc = 0; cc < NS; ++cc) {
            const float* r4 = red + (4 * cc) * DS + 2 * i;
            const float v0 = (r4[0] + r4[DS]) + (r4[2 * DS] + r4[3 * DS]);
            const float v1 = (r4[1] + r4[DS + 1]) + (r4[2 * DS + 1] + r4[3 * DS + 1]);
            a0 = fmaf(v0, scale[cc], a0), a1 = fmaf(v1, scale[cc], a1);
          }
          st_pair_c(p.C + ((size_t)b * 128 + (DS / 2) * c + i) * 2, tag, a0, a1);
          const float b0 = __shfl_down(a0, 1, 64), b1 = __shfl_down(a1, 1, 64);
          if ((i & 1) == 0) st_f4_wt(p.ctx_all + (size_t)k * B * kD + ((size_t)(DQ * c + (i >> 1)) * B + b) * 4, a0, a1, b0, b1);
        }
      }
      __syncthreads();
      CSTAMP(9);
    }
  }
  for (int i = tid; i < 2 * SA; i += kCT) p.state[(size_t)blk * kCState + i] = st_att[i];
}

}  // namespace
