// The bidirectional LSTM (in = H = 512, B <= 64) as ONE persistent launch with the weights in registers.
// (bilstm_resident_kernel below: all rows per workgroup, flags; bilstm_resident_wide_kernel: two tiles and one of two row groups
//  per workgroup, flags; bilstm_resident_tag_kernel at the end of the file: the wide decomposition with the data as the flag,
//  which is what runs -- 5.1 us per frame at B = 64 against 5.65 with flags, 2.7 against 3.6 at B = 32.  WRNN_BILSTM_FLAGS=1
//  selects the flag kernels for comparison.)
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
constexpr int kBRErr = 512;          // index of the error word in the flag block
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
  unsigned* err = p.flags + kBRErr;

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
  unsigned* err = p.flags + kBRErr;

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

// ---- B > 16, the wide decomposition with the data as the flag ("tag").  What bounds a frame of the flag forms above is not the
// matrix cores (3.4 us + 0.65 of finishing) but the recurrence's own latency: stores -> acknowledgement (1.1) -> barrier -> flag ->
// visible (1.2) -> sweep (0.3) -> fetch of h (0.9) -> h half (1.7) -> gates (0.65) = 5.85 us.  Here every published h carries ONE
// tag bit, its mantissa's last bit = (frame / 2) & 1; slot frame % 2 of the ring held frame - 2 before (the other tag; the ring
// starts as all-ones words, frames 0 and 1 carry tag 0).  A consumer requests its operand block in the middle of the next
// frame's x half and, when that is done, looks at the last bit of every word: all equal to the frame's tag and it has the
// frame's values, else it requests the block again.  No acknowledgement wait, no flag, no sweep.  The state the recurrence
// continues from IS the tagged value (out, h_n and every consumer see the same bits), at most one ulp from the cell's result.
// One bit is enough because a workgroup publishes h(s) only after it has consumed ALL of h(s - 1): everybody is done with
// h(s - 2), and nothing older than s - 2 can show in the slot because each wave's stores of one frame are acknowledged before it
// issues the next frame's.
__device__ __forceinline__ float br_tag(float v, unsigned e) { return __uint_as_float((__float_as_uint(v) & ~1u) | e); }

template <int NM>   // row tiles in this workgroup's group: 2, or 1 where the batch ends in the group's first tile
__device__ __forceinline__ void br_wide_tag_body(const BiResParams& p, int MT, int dir, int g, int m0, int jj, f32x4 (*red)[4][64]) {
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
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
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(p.xop + (size_t)dir * S * per), 0,
                                                                       (unsigned)((size_t)S * per * 16), 0x00020000);
  f32x4* hst = p.hop + (size_t)dir * 2 * per + (size_t)m * kBRBlockF4 + (size_t)(j >> 5) * 512 + (j & 7) * 64 + ((j & 31) >> 3) * 16 + n;
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(p.hop + (size_t)dir * 2 * per), 0,
                                                                       (unsigned)(2 * per * 16), 0x00020000);
  const unsigned hvoff = (unsigned)(goff + (size_t)w * 512 + l) * 16u;
  unsigned* err = p.flags + kBRErr;
  (void)g;

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
      for (int mm = 0; mm < NM; ++mm) {
        const br_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xrs, hvoff + (unsigned)(mm * kBRBlockF4 + s4 * 64) * 16u,
                                                                 (unsigned)((size_t)(s - s0) * per * 16), 0);
        xv[s4][mm] = f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
      }
  };
  auto fetch_x_group = [&](int s, int s4) {   // one K piece of the x block (NM requests)
#pragma unroll
    for (int mm = 0; mm < NM; ++mm) {
      const br_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xrs, hvoff + (unsigned)(mm * kBRBlockF4 + s4 * 64) * 16u,
                                                               (unsigned)((size_t)(s - s0) * per * 16), 0);
      xv[s4][mm] = f32x4{__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
    }
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
  auto contract = [&](const float (&wv)[2][32], const f32x4 (&v)[8][2], int a0, int a1) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
      if (s4 >= a0 && s4 < a1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)
              acc[t][mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][4 * s4 + e], v[s4][mm][e], acc[t][mm], 0, 0, 0);
      }
  };
  auto fresh = [&](unsigned e) -> bool {   // every word of hv shows tag e
    unsigned a = ~0u, o = 0u;
    if (e) {
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
        for (int mm = 0; mm < NM; ++mm)
#pragma unroll
          for (int i = 0; i < 4; ++i) a &= __float_as_uint(hv[s4][mm][i]);
      return (a & 1u) != 0u;
    }
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4)
#pragma unroll
      for (int mm = 0; mm < NM; ++mm)
#pragma unroll
        for (int i = 0; i < 4; ++i) o |= __float_as_uint(hv[s4][mm][i]);
    return (o & 1u) == 0u;
  };

  // prologue: the x half of the first frame; then the two requests every frame leaves behind, in the order it leaves them
  // (x block of the next frame, h block of the previous one), so that the wait counts the compiler derives at the loop's head
  // are the same on the way in as on the way round
  zero();
  fetch_x(s0);
  contract(wx, xv, 0, 8);
  __builtin_amdgcn_sched_barrier(0);
  fetch_h(max(s0 - 1, 0));
  for (int s = s0; s < s1; ++s) {
    BRSTAMP(0);
    __builtin_amdgcn_sched_barrier(0);   // (nothing of this frame moves up into the previous frame's MFMAs)
    // ---- h half: hv = h(s - 1), requested in the middle of the previous frame's x half
    if (s > 0) {
      const unsigned e = (unsigned)((s - 1) >> 1) & 1u;
      if (!(__all(fresh(e)) || gave_up)) {   // rare: requested too early
        for (unsigned spins = 0;; ++spins) {
          if (!wrnn::keep_polling(spins, kBRSpin, err, gave_up)) break;
          __builtin_amdgcn_s_sleep(1);
          fetch_h(s - 1);
          if (__all(fresh(e))) break;
        }
      }
      BRSTAMP(1);
    }
    // the h half's MFMAs carry the requests for the next frame's x block, four per K piece of the first half (a request blocks
    // the wave for as long as the compute unit's address path is busy with the other waves' -- 16 requests in a row cost
    // 0.3 us; spread over 64 MFMAs they cost nothing, and the block still has a microsecond and a half to arrive)
    {
      const int sx = min(s + 1, s1 - 1);
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        if (s > 0) contract(wh, hv, s4, s4 + 1);
        if (s4 < 4) fetch_x_group(sx, 2 * s4), fetch_x_group(sx, 2 * s4 + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    BRSTAMP(2);
    // Every request so far has landed (the next frame's x block among them, requested a frame ago), and the compiler is to
    // know it: as a builtin, not as asm text, this wait enters its bookkeeping, so the x half below does not wait again --
    // for "everything", which would then include the acknowledgement of the stores in between.  It also makes a fact of what
    // the tag protocol relies on: this wave's stores of the previous frame are acknowledged before the next ones go out.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();   // the finishing waves are done with the previous frame's sums
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) red[w][t * 2 + mm][l] = acc[t][mm];
    __syncthreads();
    BRSTAMP(3);
    if (mine) {
      const int slot = ft * 2 + fm;
      const f32x4 gsum = red[0][slot][l] + red[1][slot][l] + red[2][slot][l] + red[3][slot][l];
      const float ig = sigmoidf_(gsum[0] + bias[0]);
      const float fg = sigmoidf_(gsum[1] + bias[1]);
      const float gg = tanhf_(gsum[2] + bias[2]);
      const float og = sigmoidf_(gsum[3] + bias[3]);
      const float cn = fg * c + ig * gg;
      const unsigned e = (unsigned)(s >> 1) & 1u;
      const float hn = br_tag(og * tanhf_(cn), e);
      const bool active = s < len;
      if (active) c = cn, h = hn;
      const float hp = br_tag(h, e);   // (a frozen row republishes its last state under the frame's tag)
      const f32x4 h4 = {hp, __shfl(hp, n + 16), __shfl(hp, n + 32), __shfl(hp, n + 48)};
      if (u == 0) {
        br_store4_wt(hst + (size_t)(s & 1) * per, h4);
        if (active) *reinterpret_cast<f32x4*>(p.out + ((size_t)b * M + (dir ? len - 1 - s : s)) * (2 * kBRH) + dir * kBRH + 4 * j) = h4;
      }
    }
    BRSTAMP(4);
    zero();
    // ---- x half of the next frame; in its middle the request for h(s), which the other workgroups are publishing now
    if (s + 1 < s1) contract(wx, xv, 0, 4);
    __builtin_amdgcn_sched_barrier(0);
    BRSTAMP(5);
    // (the request goes out on every path, the last frame included: a path without it would make the wait counts at the loop's
    //  head conservative.  The four waves take turns, half a K piece apart: 16 requests take the address path 0.1 us when it
    //  is free and 0.35 when all four waves ask at once)
#pragma unroll
    for (int hp = 0; hp < 8; ++hp) {
      if (hp < 4 && hp == w) fetch_h(s);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < s1) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mm = 0; mm < NM; ++mm)
              acc[t][mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[t][4 * (4 + hp / 2) + 2 * (hp & 1) + e], xv[4 + hp / 2][mm][2 * (hp & 1) + e], acc[t][mm], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    BRSTAMP(6);
  }
  if (row_ok) {
    const int U = 4 * j + u;
    p.st_h[((size_t)dir * p.B + b) * kBRH + U] = h, p.st_c[((size_t)dir * p.B + b) * kBRH + U] = c;
    if (s1 == M && p.h_n) p.h_n[((size_t)dir * p.B + b) * kBRH + U] = h;
    if (s1 == M && p.c_n) p.c_n[((size_t)dir * p.B + b) * kBRH + U] = c;
  }
}

template <int MT>   // 2 .. 4 row tiles: group 0 = the first ceil(MT / 2) tiles, group 1 = the rest
__global__ __launch_bounds__(256) void bilstm_resident_tag_kernel(BiResParams p) {
  static_assert(MT >= 2 && MT <= 4, "17 .. 64 batch rows");
  constexpr int GS = (MT + 1) / 2;
  __shared__ f32x4 red[4][4][64];   // [wave][tile * 2 + row tile][lane] partial gate sums
  const int dir = blockIdx.x >> 7, g = (blockIdx.x >> 6) & 1, jj = blockIdx.x & 63;
  if (g == 0) br_wide_tag_body<GS>(p, MT, dir, 0, 0, jj, red);
  else br_wide_tag_body<MT - GS>(p, MT, dir, 1, GS, jj, red);
}

// (Measured and not kept: both directions in ONE workgroup -- tile jt of both, the same 128 registers of weights -- so that a
//  workgroup works on one direction while the other's new state travels.  With flags 6.9 us per frame at B = 64: the cycle
//  above does not get shorter by running something beside it.  With the tag bit 6.0-6.5 at B = 64, 3.3 at B = 32: every
//  workgroup then fetches both directions' operand blocks, 256 KB per frame through a compute unit's address path instead of
//  128, and with one wave per SIMD a request that waits for that path is time the wave's MFMAs do not issue.)

}  // namespace
