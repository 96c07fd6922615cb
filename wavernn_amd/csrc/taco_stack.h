// The decoder's LSTM stack (reference tacotron.py:213-227: res_rnn1, res_rnn2, both residual adds, zoneout) of the
// TEACHER-FORCED loop as persistent, weights-resident launches beside the chain launches (taco_chain.h), lstm_dims = 512, B <= 64.
//
// The launch path runs LSTM1(k) || LSTM2(k-1) as one launch per step and re-fetches both cells' 16.8 MB of weights from cold
// L2s 641 times per pass (14.3 of the pass's 19.4 GB of memory-side traffic, profiles/r2_v9_pmc_traffic.json).  Here workgroup
// i of 256 owns, for a whole chunk of steps, cell i / 128 and gate-row tile i % 128 = 4 units x (i, f, g, o) = 16 gate rows x
// K = [x (512) | h (512)]: 64 KB = 64 registers per lane, the A operand of v_mfma_f32_16x16x4_f32, taken straight from the
// launch path's fragment-ordered Layer (wave w keeps K chunks 8 w .. 8 w + 7 of each half).  Wave w contracts its K slice
// for all batch rows (MT = ceil(B / 16) B-operand tiles), the four waves' partial sums meet in LDS and wave m finishes row
// tile m lane-locally: the four gates of a (unit, row) are the lane's four accumulator registers; c and h stay in registers.
//
// It has to live BESIDE a chain workgroup on every compute unit (the chain needs all 256): chain_kernel holds 336 of a SIMD
// lane's 512 registers and 137 of the 160 KB of LDS, so this kernel is built for at most 168 registers (three waves per SIMD
// as far as the compiler is concerned) and 16 KB of LDS: the B operands of a K half (32 float4 per lane) pass through a
// small ring of registers (NG groups of MT float4 in flight) instead of all being requested at once as in bilstm_resident.h.
// Measured (B = 64, T = 128; profiles/r3_stack_*): the stack alone 6.0 us per step with four operand groups in flight and 8.5-9.8
// with one, 7-9 beside the chain; the chain's steps take
// 22 us while a stack launch runs beside them and 9.4 alone, and the pass 14.2-14.3 us per decoder step, 14.0 with the end-of-pass
// scheduling in taco_decoder.hip (launch path of the
// stack: 14.5; B = 1 .. 32: 10.9-12.7 against 13.2-14.4).  What the chain pays for is the stack's MFMAs on the SIMD's shared
// matrix pipe, not its bytes: with the MFMAs removed (timing only) the pass takes 12.4 us per step, with the operand loads
// removed and the MFMAs kept 14.6, with neither 12.7; a "wide" variant (two tiles x 32 rows per workgroup, 128 registers of
// weights, half the bytes) measured 14.3 like this one, a K-split variant (32 gate rows x one K half, the x halves publishing
// partial gate sums) 19.7 (all its MFMAs on the recurrence's critical path).  A THIN ring costs least: one group in flight
// 14.2, two / three / four 14.4 / 14.7 / 15.0.  Spacing the stack's MFMAs with s_nop so that the chain's find the pipe free more
// often: 18-38 us per step (an s_nop 15 between MFMAs costs the issuing wave about 75 cycles, not 16).
#pragma once

namespace {

constexpr int kSL = 512;             // lstm_dims this kernel is built for
constexpr int kSRing = 64;           // steps per launch at most (slots of the x1 ring)
constexpr unsigned kSSpin = 1u << 21;
constexpr int kSStateFloats = kSL * 64;   // floats of one [128][B][4] block at B = 64

struct StackParams {
  const float4* wt[2];        // Layer::wt of LSTM1 / LSTM2: [128 tiles][64 chunks][64 lanes]
  const float* bias[2];       // tile order (b_ih + b_hh)
  const float* y;             // [steps][128][B][4]: rnn_input of the launch's steps (step k0 first)
  float* x1r;                 // [kSRing][128][B][4]
  float* hr[2];               // [2 slots][128][B][4] per cell
  float* cst[2];              // [128][B][4] per cell: cell state between launches
  float* x2;                  // [steps][128][B][4] (step k0 first)
  const float* zone_u[2];     // row-major [steps][B][512] (step k0 first) or null
  unsigned* flags;            // [2][128] steps done, error word at [256]
  int B, k0, k1;
  unsigned long long* stamps; // diagnostics (WRNN_STACK_STAMPS): [2 cells][4 steps][8] 100 MHz timestamps of tiles 0, or null
};

using st_u32x4 = __attribute__((vector_size(16))) unsigned;

template <bool SC1>
__device__ __forceinline__ f32x4 st_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  const st_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, SC1 ? 16 : 0);
  return f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
}
__device__ __forceinline__ void st_store4_wt(float* p, f32x4 v) {   // write-through 16-byte store
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

// all 128 flags of a cell >= want (two per lane); bounded (wrnn::keep_polling: wave-uniform control flow)
__device__ __forceinline__ void st_wait_flags(const unsigned* f, unsigned want, unsigned* err, bool& gave_up) {
  using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
  if (gave_up) return;
  const u32x2* a = reinterpret_cast<const u32x2*>(f) + (threadIdx.x & 63);
  for (unsigned spins = 0;; ++spins) {
    u32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    if (__all(v.x >= want && v.y >= want)) break;
    if (!wrnn::keep_polling(spins, kSSpin, err, gave_up)) break;
    __builtin_amdgcn_s_sleep(2);
  }
}

// One K half (the wave's eight 16-wide chunks) of one step: acc[m] += W . X^T for the MT row tiles.  The B operands pass
// through a ring of NG groups (one chunk x MT row tiles each): NG groups are requested up front, every group is refilled
// right after the MFMAs that consumed it.  Fully unrolled: the ring is plain registers.
template <int MT, int NG, bool SC1>
__device__ __forceinline__ void st_half(__amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[MT], unsigned sbase, unsigned sstep,
                                        const float (&wv)[32], f32x4 (&acc)[MT]) {
  f32x4 win[NG][MT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int m = 0; m < MT; ++m) win[g][m] = st_load<SC1>(rs, voff[m], sbase + g * sstep);
  // (scheduling barriers: left alone, the compiler regroups the loads into "two groups, wait, 32 MFMAs, two groups, ..."
  //  whatever NG says, which leaves one round trip exposed per pair of groups)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[4 * g + e], win[g % NG][m][e], acc[m], 0, 0, 0);
    if (g + NG < 8) {
#pragma unroll
      for (int m = 0; m < MT; ++m) win[g % NG][m] = st_load<SC1>(rs, voff[m], sbase + (g + NG) * sstep);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#define SSTAMP(i)                                                                               \
  do {                                                                                          \
    if (p.stamps && j == 0 && threadIdx.x == 0 && k >= 40 && k < 44)                            \
      p.stamps[(cell * 4 + (k - 40)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)

template <int MT, int NG>   // 16-row tiles of the batch, B-operand groups in flight
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void stack_kernel(StackParams p) {
  __shared__ f32x4 red[4][MT][64];   // [wave][row tile][lane] partial gate sums
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
  const int cell = blockIdx.x >> 7, j = blockIdx.x & 127;
  const int B = p.B;
  // resident weights: chunk 8 w + g of the x half and of the h half, lane (row l % 16 of the tile, k slot l / 16)
  float wx[32], wh[32];
  {
    const float4* wp = p.wt[cell] + ((size_t)j * 64 + 8 * w) * 64 + l;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 a = wp[(size_t)g * 64], b = wp[(size_t)(32 + g) * 64];
      wx[4 * g] = a.x, wx[4 * g + 1] = a.y, wx[4 * g + 2] = a.z, wx[4 * g + 3] = a.w;
      wh[4 * g] = b.x, wh[4 * g + 1] = b.y, wh[4 * g + 2] = b.z, wh[4 * g + 3] = b.w;
    }
  }
  // operand role: lane (k slot kq = l / 16, row r = l % 16 of each row tile); finishing role: wave w = row tile w, lane = (unit l / 16, row l % 16)
  const int kq = l >> 4, r = l & 15;
  unsigned voff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) voff[m] = (unsigned)(((32 * w + kq) * B + min(16 * m + r, B - 1)) * 16);
  const unsigned sstep = (unsigned)(64 * B);        // bytes between consecutive chunks of a wave: 4 k-quads x B rows x 16
  const unsigned blk = (unsigned)(kSL * B * 4);     // bytes of one [128][B][4] block
  const int u = kq, b = 16 * w + r;
  const bool fin = w < MT, row_ok = fin && b < B;
  const int bc = min(b, B - 1);
  const size_t own = ((size_t)j * B + bc) * 4;      // floats: the lane's row of the workgroup's four units in a block
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias[cell] + j * 16 + u * 4);
  float c = 0.f, h = 0.f;
  if (p.k0 > 0) c = p.cst[cell][own + u], h = p.hr[cell][(size_t)((p.k0 - 1) & 1) * kSL * B + own + u];
  bool gave_up = false;
  const unsigned steps = (unsigned)(p.k1 - p.k0);
  const __amdgpu_buffer_rsrc_t rs_x = cell == 0
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.y), 0, steps * blk, 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(p.x1r, 0, (unsigned)kSRing * blk, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.hr[cell], 0, 2 * blk, 0x00020000);
  const unsigned* flags_own = p.flags + cell * 128;
  unsigned* err = p.flags + 256;
  const float* zu = p.zone_u[cell];

  f32x4 acc[MT];
  for (int k = p.k0; k < p.k1; ++k) {
    const int ks = k - p.k0;   // step of the launch; x1 slot
    SSTAMP(0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- x half: cell 0 needs nobody, cell 1 needs cell 0's step k (which runs ahead)
    if (cell == 0) {
      st_half<MT, NG, false>(rs_x, voff, (unsigned)ks * blk, sstep, wx, acc);
    } else {
      st_wait_flags(p.flags, (unsigned)k + 1, err, gave_up);
      st_half<MT, NG, true>(rs_x, voff, (unsigned)ks * blk, sstep, wx, acc);
    }
    SSTAMP(1);
    // the finishing lanes' own operands: residual input, zoneout uniform
    float xin = 0.f, zz = 1.f;
    if (fin) {
      if (cell == 0) xin = p.y[(size_t)ks * kSL * B + own + u];
      else xin = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, (unsigned)((own + u) * 4), (unsigned)ks * blk, 16));
      if (zu) zz = zu[((size_t)ks * B + bc) * kSL + 4 * j + u];
    }
    // ---- h half: the own cell's step k - 1
    if (k > 0) {
      st_wait_flags(flags_own, (unsigned)k, err, gave_up);
      SSTAMP(2);
      st_half<MT, NG, true>(rs_h, voff, (unsigned)((k - 1) & 1) * blk, sstep, wh, acc);
    }
    SSTAMP(3);
#pragma unroll
    for (int m = 0; m < MT; ++m) red[w][m][l] = acc[m];
    __syncthreads();
    if (fin) {
      const f32x4 g = (red[0][w][l] + red[1][w][l]) + (red[2][w][l] + red[3][w][l]);
      const float ig = sigmoidf_(g[0] + bias[0]);
      const float fg = sigmoidf_(g[1] + bias[1]);
      const float gg = tanhf_(g[2] + bias[2]);
      const float og = sigmoidf_(g[3] + bias[3]);
      c = fg * c + ig * gg;
      const float hn = og * tanhf_(c);
      h = zz < kZoneP ? h : hn;   // zoneout keeps the previous hidden value (the cell state is not zoned out)
      const float xo = xin + h;
      // the row's four units into one lane: 16-byte stores (a 4-byte write-through store is a fabric write of its own)
      const f32x4 h4 = {h, __shfl(h, r + 16), __shfl(h, r + 32), __shfl(h, r + 48)};
      const f32x4 x4 = {xo, __shfl(xo, r + 16), __shfl(xo, r + 32), __shfl(xo, r + 48)};
      if (u == 0 && row_ok) {
        st_store4_wt(p.hr[cell] + (size_t)(k & 1) * kSL * B + own, h4);
        if (cell == 0) st_store4_wt(p.x1r + (size_t)ks * kSL * B + own, x4);
        else *reinterpret_cast<f32x4*>(p.x2 + (size_t)ks * kSL * B + own) = x4;
      }
    }
    SSTAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier in front of the flag
    __syncthreads();
    if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.flags + cell * 128 + j), "v"((unsigned)(k + 1)) : "memory");
    SSTAMP(5);
  }
  if (row_ok) p.cst[cell][own + u] = c;
}

__global__ void stack_poison_kernel(const unsigned* err, float* a, size_t na, float* b, size_t nb) {   // a hand-over timed out: no silent garbage
  if (*err == 0) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x)
    (i < na ? a[i] : b[i - na]) = NAN;
}

}  // namespace
