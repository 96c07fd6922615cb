// The bidirectional LSTM (in = H = 512, B <= 64) as ONE persistent launch with the weights in registers.
// (bilstm_resident_kernel below: B <= 16; B > 16 runs bilstm_resident_wide_kernel at the end of the file, same scheme with two
//  tiles and one of two row groups per workgroup.)
//
// The launch loop (bilstm.hip) fetches both cells' 16.8 MB of weights from a cold L2 at every frame: 9 us per frame for
// 537 MFLOP.  Here workgroup i of 256 owns, for the whole sequence, direction i / 128 and tile i % 128 = 4 units x
// (i, f, g, o) = 16 gate rows x K = [x (512) | h (512)]: 64 KB = 64 registers per lane as the A operand of
// v_mfma_f32_16x16x4_f32.  Wave w contracts K slice [128 w, 128 w + 128) of the x half and of the h half for all 64 batch rows
// (four 16-row B operands), the four waves' partial sums meet in LDS, and wave w finishes batch rows 16 w .. 16 w + 15: the
// four gates of a (unit, row) are the lane's four accumulator registers, so the cell update is lane-local and c, h stay in
// registers.
//
// Activations are laid out as the matrix cores consume them ("operand order"): per (direction, frame) a block of
// [m = row / 16][wave][s4 0..7][lane = 16 kq + row % 16] float4, where the float4 holds k = 128 wave + 32 kq + 4 s4 + (0..3):
// every load instruction of a wave is 1 KB contiguous.  x is put into that order once per call (bilstm_operand_pack_kernel,
// the reverse direction aligned to each row's own last frame as birnn_pack_kernel does); h(s) is written in it by its
// producers: a workgroup's 4 units x 16 rows are 256 contiguous bytes per m (whole cache lines, one writer each).
//
// Hand-over per frame and direction: 16-byte write-through (sc1) stores of h(s) -> every storing wave waits for their
// acknowledgement -> workgroup barrier -> one flag per producer workgroup (value = frames done, sc1 store); consumers poll the
// direction's 128 flags (512 bytes per wave, sc1 loads), then fetch h(s) with sc1 loads (past the compute unit's L1; every
// frame has its own block of h, written once per launch).  The x half of frame s + 1 is contracted between the stores and
// their acknowledgement and while the flags are on their way: it needs nothing from the other workgroups.
// Measured at B = 64 (us per frame): poll 0.25 (the flags are up by then), fetch of h + its contraction 4.0-4.5 (2.3 of it the
// fetch's latency), sums through LDS + gates + stores 0.65, x contraction 1.9 (0.16 left of the acknowledgement): 7.0, against
// 9.3 for the launch loop and 3.4 for the matrix cores alone.  The chain flag -> visible (1.8) -> fetch (2.3) -> contraction
// -> gates is what a frame waits for; running the batch rows as two halves half a frame apart (each half's chain behind the
// other half's contraction; built, same results) gave 7.1: two barriers' worth of finishing per frame instead of one and the
// same chain per half.
//
// Waits are bounded; a wave that gives up stops waiting (the launch drains) and raises the error word, and the host turns
// the output into NaN.  Summation order differs from the launch loop's (K slices per wave, then waves 0..3), so the two
// paths agree to fp32 rounding, not bit for bit.
#pragma once

namespace {

constexpr int kBRH = 512;            // in = H
constexpr int kBRBlockF4 = 4 * 8 * 64;   // float4 per (m-tile) of one K half: [wave][s4][lane]
constexpr unsigned kBRSpin = 1u << 21;
constexpr int kBRChunk = 224;        // frames per launch: 2 x 224 x 128 KB = 56 MB of operand-order x at B = 64

struct BiResParams {
  const float* wres;        // [2][128][4 waves][64 slots][64 lanes]: slots 0..31 x half, 32..63 h half
  const float* bias[2];     // tile order [128][16] (b_ih + b_hh)
  const f32x4* xop;         // [2][s1 - s0][MT][kBRBlockF4]: the launch's frames of x in operand order
  f32x4* hop;               // [2][2 slots][MT][kBRBlockF4]: h(s) at slot s % 2 (whoever writes h(s) has seen every flag >= s:
                            //  everybody has consumed h(s-2)), so the workspace does not grow with M
  unsigned* flags;          // [2][128] frames done (counting on over the launches of a call), then the error word at [256]
  const int32_t* lens;      // [B] or null
  float* out;               // [B][M][1024]
  float *h_n, *c_n;         // [2][B][512] or null
  float *st_h, *st_c;       // [2][B][512]: state between the launches of a call
  int B, M, MT;
  int s0, s1;               // frames of this launch (a call runs chunks of at most kBRChunk frames: x in operand order for
                            //  one chunk at a time)
  unsigned long long* stamps;   // diagnostics (WRNN_BILSTM_STAMPS): [4 frames][8] 100 MHz timestamps of workgroup 0, or null
};

// x [B][M][512] -> operand order, both directions (zero past a row's length and for rows >= B)
__global__ void bilstm_operand_pack_kernel(const f32x4* __restrict__ x, const int32_t* __restrict__ lens, int B, int M, int MT,
                                            int s0, int S, f32x4* __restrict__ xop) {   // frames s0 .. s0 + S - 1
  const size_t per = (size_t)MT * kBRBlockF4, total = 2 * (size_t)S * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), s4 = (int)((i >> 6) & 7), w = (int)((i >> 9) & 3);
    const size_t r = i >> 11;
    const int m = (int)(r % MT);
    const size_t ds = r / MT;
    const int s = s0 + (int)(ds % S), dir = (int)(ds / S);
    const int b = 16 * m + (lane & 15), k4 = 32 * w + 8 * (lane >> 4) + s4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b < B) {
      const int len = lens ? min(max(lens[b], 0), M) : M;
      if (s < len) v = x[((size_t)b * M + (dir ? len - 1 - s : s)) * (kBRH / 4) + k4];
    }
    xop[i] = v;
  }
}

// a hand-over timed out: NaN in EVERY output of the call (out, h_n, c_n), no silent garbage
__global__ void bilstm_poison_kernel(const unsigned* err, float* out, size_t n, float* h_n, float* c_n, size_t ns) {
  if (*err == 0) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = __builtin_nanf("");
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ns; i += (size_t)gridDim.x * blockDim.x) {
    if (h_n) h_n[i] = __builtin_nanf("");
    if (c_n) c_n[i] = __builtin_nanf("");
  }
}

__device__ __forceinline__ void br_store_wt(float* p, float v) {   // write-through 4-byte store
  asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

// all 128 flags of the direction >= want (two per lane); bounded
__device__ __forceinline__ void br_wait_flags(const unsigned* f, unsigned want, unsigned* err, bool& gave_up) {
  using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
  const u32x2* a = reinterpret_cast<const u32x2*>(f) + (threadIdx.x & 63);
  for (unsigned spins = 0;; ++spins) {
    u32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    const bool ok = v.x >= want && v.y >= want;
    if (__all(ok) || gave_up) break;
    if (!wrnn::keep_polling(spins, kBRSpin, err, gave_up)) break;
    __builtin_amdgcn_s_sleep(2);
  }
}

// one K half of one frame: the wave's 32 KB of the operand block into registers (all loads in flight at once) ...
template <int MT>
__device__ __forceinline__ void br_fetch(const f32x4* __restrict__ op, f32x4 (&v)[8][4]) {
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < MT) v[s4][m] = op[(size_t)m * kBRBlockF4 + s4 * 64];
}
// the same for a block other workgroups wrote in this launch: every load sc1 (aux 16: past this compute unit's L1)
using br_u32x4 = __attribute__((vector_size(16))) unsigned;
template <int MT>
__device__ __forceinline__ void br_fetch_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, f32x4 (&v)[8][4]) {
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < MT) {
        const br_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)(m * kBRBlockF4 + s4 * 64) * 16u, soff, 16);
        v[s4][m] = f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
      }
}
// ... and acc[m] += W(slots) . block, K pieces [S0, S1) (consecutive MFMAs are independent: four row tiles)
template <int MT, int S0 = 0, int S1 = 8>
__device__ __forceinline__ void br_contract(const float (&wv)[32], const f32x4 (&v)[8][4], f32x4 (&acc)[4]) {
#pragma unroll
  for (int s4 = S0; s4 < S1; ++s4)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m < MT) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[4 * s4 + e], v[s4][m][e], acc[m], 0, 0, 0);
}
__device__ __forceinline__ void br_store4_wt(f32x4* p, f32x4 v) {   // write-through 16-byte store
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}

#define BRSTAMP(i)                                                                          \
  do {                                                                                      \
    if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0 && s >= 100 && s < 104)             \
      p.stamps[(s - 100) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                     \
  } while (0)

template <int MT>   // 16-row tiles of the batch: ceil(B / 16)
__global__ __launch_bounds__(256) void bilstm_resident_kernel(BiResParams p) {
  __shared__ f32x4 red[4][4][64];   // [wave][m][lane] partial gate sums
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int dir = blockIdx.x >> 7, j = blockIdx.x & 127;
  const int M = p.M;
  float wx[32], wh[32];
  {
    const float* wp = p.wres + ((((size_t)dir * 128 + j) * 4 + w) * 64) * 64 + l;
#pragma unroll
    for (int i = 0; i < 32; ++i) wx[i] = wp[i * 64], wh[i] = wp[(32 + i) * 64];
  }
  // finishing role: wave w = batch rows 16 w .. 16 w + 15, lane = (unit l / 16, row l % 16)
  const int u = l >> 4, n = l & 15, b = 16 * w + n, U = 4 * j + u;
  const bool row_ok = b < p.B;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias[dir] + j * 16 + u * 4);
  const int len = row_ok ? (p.lens ? min(max(p.lens[b], 0), M) : M) : 0;
  float c = 0.f, h = 0.f;
  const int s0 = p.s0, s1 = p.s1, S = s1 - s0;
  if (s0 > 0 && row_ok) c = p.st_c[((size_t)dir * p.B + b) * kBRH + U], h = p.st_h[((size_t)dir * p.B + b) * kBRH + U];
  bool gave_up = false;
  const size_t per = (size_t)MT * kBRBlockF4;            // float4 per (direction, frame)
  const f32x4* xop = p.xop + (size_t)dir * S * per + (size_t)w * 512 + l;
  // where the row's four units go (lanes 0..15 store for their row): k = 4 j .. 4 j + 3 -> wave j / 32, kq = (j % 32) / 8, s4 = j % 8; block m = w
  f32x4* hst = p.hop + (size_t)dir * 2 * per + (size_t)w * kBRBlockF4 + (size_t)(j >> 5) * 512 + (j & 7) * 64 + ((j & 31) >> 3) * 16 + n;
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(p.hop + (size_t)dir * 2 * per), 0,
                                                                       (unsigned)(2 * per * 16), 0x00020000);
  const unsigned hvoff = (unsigned)(w * 512 + l) * 16u;
  const unsigned* flags = p.flags + dir * 128;
  unsigned* err = p.flags + 256;

  f32x4 acc[4], xv[8][4], hv[8][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (S > 0) {
    br_fetch<MT>(xop, xv);
    br_contract<MT>(wx, xv, acc);
  }
  for (int s = s0; s < s1; ++s) {
    BRSTAMP(0);
    if (s > 0) {
      br_wait_flags(flags, (unsigned)s, err, gave_up);
      BRSTAMP(1);
      br_fetch_sc1<MT>(hrs, hvoff, (unsigned)((size_t)((s - 1) & 1) * per * 16), hv);
    }
    // behind the h block the next frame's x block: it lands while h is contracted
    if (s + 1 < s1) br_fetch<MT>(xop + (size_t)(s + 1 - s0) * per, xv);
    if (s > 0) br_contract<MT>(wh, hv, acc);
    BRSTAMP(2);
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m < MT) red[w][m][l] = acc[m];
    __syncthreads();
    BRSTAMP(3);
    if (w < MT) {
      const f32x4 g = red[0][w][l] + red[1][w][l] + red[2][w][l] + red[3][w][l];
      const float ig = sigmoidf_(g[0] + bias[0]);
      const float fg = sigmoidf_(g[1] + bias[1]);
      const float gg = tanhf_(g[2] + bias[2]);
      const float og = sigmoidf_(g[3] + bias[3]);
      const float cn = fg * c + ig * gg;
      const float hn = og * tanhf_(cn);
      const bool active = s < len;   // past the row's last frame the state is frozen (packed semantics)
      if (active) c = cn, h = hn;
      // the row's four units into one lane: 16-byte stores (a 4-byte write-through store is a fabric write of its own)
      const f32x4 h4 = {h, __shfl(h, n + 16), __shfl(h, n + 32), __shfl(h, n + 48)};
      if (u == 0) {
        br_store4_wt(hst + (size_t)(s & 1) * per, h4);
        if (active) *reinterpret_cast<f32x4*>(p.out + ((size_t)b * M + (dir ? len - 1 - s : s)) * (2 * kBRH) + dir * kBRH + 4 * j) = h4;
      }
    }
    BRSTAMP(4);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s + 1 < s1) br_contract<MT, 0, 4>(wx, xv, acc);   // needs nobody: covers the stores' acknowledgement ...
    BRSTAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave, before the barrier in front of the flag
    BRSTAMP(6);
    __syncthreads();
    if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(p.flags + dir * 128 + j), "v"((unsigned)(s + 1)) : "memory");
    BRSTAMP(7);
    if (s + 1 < s1) br_contract<MT, 4, 8>(wx, xv, acc);   // ... and the flags' way to the other workgroups
  }
  if (row_ok) {
    p.st_h[((size_t)dir * p.B + b) * kBRH + U] = h, p.st_c[((size_t)dir * p.B + b) * kBRH + U] = c;
    if (s1 == M && p.h_n) p.h_n[((size_t)dir * p.B + b) * kBRH + U] = h;
    if (s1 == M && p.c_n) p.c_n[((size_t)dir * p.B + b) * kBRH + U] = c;
  }
}

// (Round 3, measured and not kept: ONE tile per workgroup and 512 workgroups, two per compute unit, so that the hardware
//  interleaves two independent sequences -- a direction's contraction beside the other direction's hand-over -- took 11.7-12.0 us
//  per frame at B = 64 against 5.7-5.9 for the decomposition below, with either assignment of directions to the two workgroups
//  of a compute unit: each of them fetches its own 64 KB of h and of x per frame, twice the bytes per compute unit, and their
//  MFMAs share one matrix pipe per SIMD.  The same holds for one workgroup that owns a tile of both directions.)
// ---- B > 16: the "wide" decomposition (7.07 -> 5.64 us per frame at B = 64, 4.23 -> 3.66 at B = 32).  With all 64 rows per workgroup every compute unit fetches the direction's whole h
// (128 KB per frame; 32 compute units per XCD ask their L2 for the same 4 MB at the same moment: the fetch takes 2.3 us
// against 0.7 at 16 rows).  Here a workgroup owns TWO gate-row tiles (128 registers of weights per lane) for ONE of two groups
// of batch rows (one or two 16-row tiles each): the same MFMA count per workgroup, half the bytes to fetch, and a group's frame
// waits for the 64 workgroups of its own group only.  Wave w finishes (tile w / 2, row tile w % 2).
template <int NM>   // row tiles in this workgroup's group: 2, or 1 where the batch ends in the group's first tile
__device__ __forceinline__ void br_wide_body(const BiResParams& p, int MT, int dir, int g, int m0, int jj, f32x4 (*red)[4][64]) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int M = p.M;
  float wx[2][32], wh[2][32];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float* wp = p.wres + ((((size_t)dir * 128 + 2 * jj + t) * 4 + w) * 64) * 64 + l;
#pragma unroll
    for (int i = 0; i < 32; ++i) wx[t][i] = wp[i * 64], wh[t][i] = wp[(32 + i) * 64];
  }
  // finishing role: wave w = (tile 2 jj + w / 2, row tile m0 + w % 2), lane = (unit l / 16, row l % 16)
  const int ft = w >> 1, fm = w & 1, j = 2 * jj + ft, m = m0 + fm;
  const bool mine = fm < NM;
  const int u = l >> 4, n = l & 15, b = 16 * m + n;
  const bool row_ok = mine && b < p.B;
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias[dir] + j * 16 + u * 4);
  const int len = row_ok ? (p.lens ? min(max(p.lens[b], 0), M) : M) : 0;
  float c = 0.f, h = 0.f;
  const int s0 = p.s0, s1 = p.s1, S = s1 - s0;
  if (s0 > 0 && row_ok) c = p.st_c[((size_t)dir * p.B + b) * kBRH + 4 * j + u], h = p.st_h[((size_t)dir * p.B + b) * kBRH + 4 * j + u];
  bool gave_up = false;
  const size_t per = (size_t)MT * kBRBlockF4;            // float4 per (direction, frame)
  const size_t goff = (size_t)m0 * kBRBlockF4;           // the group's first row tile
  const f32x4* xop = p.xop + (size_t)dir * S * per + goff + (size_t)w * 512 + l;
  f32x4* hst = p.hop + (size_t)dir * 2 * per + (size_t)m * kBRBlockF4 + (size_t)(j >> 5) * 512 + (j & 7) * 64 + ((j & 31) >> 3) * 16 + n;
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(p.hop + (size_t)dir * 2 * per), 0,
                                                                       (unsigned)(2 * per * 16), 0x00020000);
  const unsigned hvoff = (unsigned)(goff + (size_t)w * 512 + l) * 16u;
  unsigned* const flags = p.flags + (dir * 2 + g) * 64;   // one per workgroup of the (direction, group)
  unsigned* err = p.flags + 256;

  f32x4 acc[2][2], xv[8][2], hv[8][2];
  auto zero = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) acc[t][mm] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto fetch_x = [&](int s) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) xv[s4][mm] = xop[(size_t)(s - s0) * per + (size_t)mm * kBRBlockF4 + s4 * 64];
  };
  auto fetch_h = [&](int s) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) {
        const br_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(hrs, hvoff + (unsigned)(mm * kBRBlockF4 + s4 * 64) * 16u,
                                                                 (unsigned)((size_t)(s & 1) * per * 16), 16);
        hv[s4][mm] = f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
      }
  };
  auto contract = [&](const float (&wv)[2][32], const f32x4 (&v)[8][2], int s0, int s1) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
      if (s4 >= s0 && s4 < s1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)
              acc[t][mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][4 * s4 + e], v[s4][mm][e], acc[t][mm], 0, 0, 0);
      }
  };
  // all 64 flags of the (direction, group) >= want (one per lane); bounded
  auto wait_flags = [&](unsigned want) {
    const unsigned* a = flags + l;
    for (unsigned spins = 0;; ++spins) {
      unsigned v;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
      if (__all(v >= want) || gave_up) break;
      if (!wrnn::keep_polling(spins, kBRSpin, err, gave_up)) break;
      __builtin_amdgcn_s_sleep(2);
    }
  };

  zero();
  if (S > 0) {
    fetch_x(s0);
    contract(wx, xv, 0, 8);
  }
  for (int s = s0; s < s1; ++s) {
    if (s > 0) {
      wait_flags((unsigned)s);
      fetch_h(s - 1);
    }
    if (s + 1 < s1) fetch_x(s + 1);
    if (s > 0) contract(wh, hv, 0, 8);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) red[w][t * 2 + mm][l] = acc[t][mm];
    __syncthreads();
    if (mine) {
      const int slot = ft * 2 + fm;
      const f32x4 gsum = red[0][slot][l] + red[1][slot][l] + red[2][slot][l] + red[3][slot][l];
      const float ig = sigmoidf_(gsum[0] + bias[0]);
      const float fg = sigmoidf_(gsum[1] + bias[1]);
      const float gg = tanhf_(gsum[2] + bias[2]);
      const float og = sigmoidf_(gsum[3] + bias[3]);
      const float cn = fg * c + ig * gg;
      const float hn = og * tanhf_(cn);
      const bool active = s < len;
      if (active) c = cn, h = hn;
      const f32x4 h4 = {h, __shfl(h, n + 16), __shfl(h, n + 32), __shfl(h, n + 48)};
      if (u == 0) {
        br_store4_wt(hst + (size_t)(s & 1) * per, h4);
        if (active) *reinterpret_cast<f32x4*>(p.out + ((size_t)b * M + (dir ? len - 1 - s : s)) * (2 * kBRH) + dir * kBRH + 4 * j) = h4;
      }
    }
    zero();
    if (s + 1 < s1) contract(wx, xv, 0, 4);                  // covers the stores' acknowledgement ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave, before the barrier in front of the flag
    __syncthreads();
    if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 0" ::"v"(flags + jj), "v"((unsigned)(s + 1)) : "memory");
    if (s + 1 < s1) contract(wx, xv, 4, 8);                  // ... and the flags' way to the other workgroups
  }
  if (row_ok) {
    const int U = 4 * j + u;
    p.st_h[((size_t)dir * p.B + b) * kBRH + U] = h, p.st_c[((size_t)dir * p.B + b) * kBRH + U] = c;
    if (s1 == M && p.h_n) p.h_n[((size_t)dir * p.B + b) * kBRH + U] = h;
    if (s1 == M && p.c_n) p.c_n[((size_t)dir * p.B + b) * kBRH + U] = c;
  }
}

template <int MT>   // 2 .. 4 row tiles: group 0 = the first ceil(MT / 2) tiles, group 1 = the rest
__global__ __launch_bounds__(256) void bilstm_resident_wide_kernel(BiResParams p) {
  static_assert(MT >= 2 && MT <= 4, "the wide decomposition is for 17 .. 64 batch rows");
  constexpr int GS = (MT + 1) / 2;
  __shared__ f32x4 red[4][4][64];   // [wave][tile * 2 + row tile][lane] partial gate sums
  const int dir = blockIdx.x >> 7, g = (blockIdx.x >> 6) & 1, jj = blockIdx.x & 63;
  if (g == 0) br_wide_body<GS>(p, MT, dir, 0, 0, jj, red);
  else br_wide_body<MT - GS>(p, MT, dir, 1, GS, jj, red);
}

}  // namespace
