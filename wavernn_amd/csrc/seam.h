// Hand-over ("seam") primitives of the persistent, weights-resident kernels (taco_resident.h: free-running Tacotron decode;
// wavernn_resident.h: unbatched WaveRNN generation): 256 workgroups x 512 threads, one per compute unit, all co-resident,
// every value a stage produces published as ONE 8-byte {tag, value} granule with a write-through (sc1) store and swept by
// its consumers with sc1 loads until the tag equals the epoch they wait for.  The data is the flag: no counter, no fence
// (CDNA4 guide, Guideline 16 form R2).  Measured: tools/microbench/gather_bench.hip.
#pragma once

namespace {

using u64 = unsigned long long;
using u32 = unsigned;

constexpr int kRW = 256;         // workgroups = compute units
constexpr int kRT = 512;         // threads per workgroup
constexpr u32 kRSpinLimit = 1u << 21;   // sweeps (about 1 us each) before a wave gives up: seconds, not milliseconds -- the workgroups
                                        // of a launch become resident one by one while other streams' kernels still hold compute units
constexpr int kSlot = 8;         // granules (64 bytes) per workgroup slot of a granule array


__device__ __forceinline__ u64 ld_gran(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(u64* p, u32 epoch, float v) {
  __hip_atomic_store(p, ((u64)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two granules that sit side by side (16-byte aligned) in ONE write-through store: a single fabric write instead of two
// (an 8-byte sc1 store costs 2.7x a 16-byte one per byte, MI355X_MICROARCH.md).  Each granule keeps its own tag; an
// aligned 8-byte half of an aligned 16-byte store is never torn.
__device__ __forceinline__ void st_gran2(u64* p, u32 epoch, float v0, float v1) {
  using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
  const u32x4_ q = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
  // (s_nop: a store of more than 8 bytes needs one wait state before its data registers may be overwritten, and the
  //  compiler does not look inside the asm statement)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(q) : "memory");
}

// (the same as a plain store: the pair stays in this XCD's L2; for hand-overs whose readers are seen to share the writer's XCD)
__device__ __forceinline__ void st_gran2_l2(u64* p, u32 epoch, float v0, float v1) {
  using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
  const u32x4_ q = {__float_as_uint(v0), epoch, __float_as_uint(v1), epoch};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(q) : "memory");
}

// sum over the 64 lanes, result in every lane: four DPP steps inside each row of 16, then the four row sums
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}
// Four wave sums at once, each result in every lane: the halves of the wave are folded with v_permlane32_swap (a, b and
// c, d pairwise: one swap + one add leaves a's half sums in lanes 0-31 and b's in lanes 32-63), the rows of 16 with
// v_permlane16_swap (rows then hold a, c, b, d), the 16 lanes of a row with four DPP steps; four v_readlane broadcast the
// results.  17 instructions instead of 4 x 11, and one dependent chain instead of four.
__device__ __forceinline__ float swap_add32(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  float v = swap_add16(swap_add32(a, b), swap_add32(c, d));   // rows of 16 lanes: a | c | b | d
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  const int iv = __float_as_int(v);
  a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), c = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  b = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), d = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  float v = swap_add32(a, b);   // lanes 0-31: a's half sums, lanes 32-63: b's
  v = dpp_add<0xB1>(v), v = dpp_add<0x4E>(v), v = dpp_add<0x141>(v), v = dpp_add<0x140>(v);
  const int iv = __float_as_int(v);
  a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  b = __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max<0xB1>(v), v = dpp_max<0x4E>(v), v = dpp_max<0x141>(v), v = dpp_max<0x140>(v);
  const int iv = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

// Inclusive prefix sum of a double over the 64 lanes (lane i gets v_0 + .. + v_i) on the DPP path: Hillis-Steele inside
// each row of 16 (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then the last lane of rows 0 and 2
// added to rows 1 and 3 (row_bcast15), then lane 31 to rows 2 and 3 (row_bcast31).  About 20 DPP moves and 6 adds; the
// same scan through ds_bpermute (__shfl_up) costs a LDS-crossbar round trip per step and dword.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_f64(double v) {
  v += dpp_shifted<0x111, 0xf>(v);   // row_shr:1
  v += dpp_shifted<0x112, 0xf>(v);   // row_shr:2
  v += dpp_shifted<0x114, 0xf>(v);   // row_shr:4
  v += dpp_shifted<0x118, 0xf>(v);   // row_shr:8
  v += dpp_shifted<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_shifted<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc), acc = fmaf(a.y, b.y, acc), acc = fmaf(a.z, b.z, acc), acc = fmaf(a.w, b.w, acc);
  return acc;
}

// After a sweep that did not find all its tags: keep waiting?  (wrnn::keep_polling, common.h: wave-uniform, one scalar compare
// per sweep for the first 4096 sweeps -- a hand-over takes one or two, a group that is off the step's critical path 30-70 --
// then the error word and a back-off between sweeps.)
__device__ __forceinline__ bool spin_again(u32 spins, u32* err, bool& gave_up) {
  if (!wrnn::keep_polling(spins, kRSpinLimit, err, gave_up)) return false;
  if (spins >= 4096u) __builtin_amdgcn_s_sleep(8);
  return true;
}

// Granule arrays.  Every workgroup publishes into its OWN 64-byte slot of an array (slot w = granules w*kSlot ..): with the
// granules of different workgroups packed side by side, sixteen writers on eight XCDs share one 128-byte line and a
// 256-granule hand-over takes 3.2 us; one writer per 64 bytes: 1.9 us (tools/microbench/gather_bench.hip, "placement").
// Element e of a slot: batch row e / G, value e % G of the G (1 or 2) values a workgroup contributes per row.
// A sweep stages element (w, e) at lds[dst + (e / G) * size + G * w + e % G], i.e. as row-major [B][size] vectors.
struct SlotSeg {
  const u64* g;
  int lg_slots;  // log2 of the number of publishing workgroups (256 or 128)
  int E, G;      // elements per slot (E = G * B, or the positions per workgroup), values per row
  int size;      // row length of the staged vector
  int dst;       // LDS offset (floats) of the staged vector
  int limit;     // staged indices >= limit do not exist (ragged last position row)
};

// Sweep the segments until every tag equals `epoch`, then stage the values in LDS.  All segments of a stage are polled in
// ONE loop (their round trips overlap).  Only lanes that own a granule poll.  A wave leaves as soon as ITS granules are
// there; the workgroup barrier that follows joins the waves.  Bounded: a sweep that never completes raises the error word
// and falls through (the host reports it); a wave that has given up once (`gave_up`, a flag the caller keeps per wave) does
// not wait again, so a launch that cannot make progress drains after about one spin limit instead of one per seam.
template <int PER, int NSEG>
__device__ __forceinline__ void gather(const SlotSeg (&seg)[NSEG], u32 epoch, float* __restrict__ lds, u32* err, bool& gave_up) {
  const int tid = threadIdx.x;
  if (gave_up) return;
  const u64* addr[PER];
  int dsti[PER];
  bool mine = false;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = tid + kRT * i;
    addr[i] = nullptr, dsti[i] = -1;
    int off = 0;
#pragma unroll
    for (int q = 0; q < NSEG; ++q) {
      const int cnt = seg[q].E << seg[q].lg_slots, local = idx - off;
      if (local >= 0 && local < cnt) {
        const int w = local & ((1 << seg[q].lg_slots) - 1), e = local >> seg[q].lg_slots;
        const int b = seg[q].G == 2 ? e >> 1 : e, u = seg[q].G == 2 ? e & 1 : 0;
        const int d = b * seg[q].size + seg[q].G * w + u;
        if (d < seg[q].limit) addr[i] = seg[q].g + w * kSlot + e, dsti[i] = seg[q].dst + d;
      }
      off += cnt;
    }
    mine |= dsti[i] >= 0;
  }
  if (!__any(mine)) return;  // the whole wave has nothing to fetch
  u64 x[PER];
  for (u32 spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (dsti[i] >= 0) {
        x[i] = ld_gran(addr[i]);
        ok &= (u32)(x[i] >> 32) == epoch;
      }
    if (__all(ok)) break;
    if (!spin_again(spins, err, gave_up)) break;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (dsti[i] >= 0) lds[dsti[i]] = __uint_as_float((u32)x[i]);
}

template <int N>
__device__ __forceinline__ float sel(const float (&a)[N], int i) {
  float v = a[0];
#pragma unroll
  for (int q = 1; q < N; ++q) v = i == q ? a[q] : v;
  return v;
}
template <int N>
__device__ __forceinline__ int sel(const int (&a)[N], int i) {
  int v = a[0];
#pragma unroll
  for (int q = 1; q < N; ++q) v = i == q ? a[q] : v;
  return v;
}


}  // namespace
