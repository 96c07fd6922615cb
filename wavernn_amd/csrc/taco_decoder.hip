// Tacotron decoder step / loops for gfx950 (MI355X), fp32 storage and fp32 accumulate.
//
// Replaces reference models/tacotron.py:86-100 (PreNet), :122-156 (LSA), :159-235
// (Decoder.forward) and the decoder loop of Tacotron.forward (:299-316,:334-348,:356).
//
// Structure: every dependency stage of a step is a job (linjob.h: small-batch GEMMs on
// v_mfma_f32_16x16x4_f32 with the stage's pointwise work fused as epilogue; AttnJob below: the whole
// location-sensitive attention incl. split softmax and context); a launch ("tick") carries up to
// three independent jobs.  The step API and free-running decode run one job per launch; the
// teacher-forced loop software-pipelines three ticks per step with the prenet and the mel projection
// hoisted out of the loop (see wrnn_decoder_run_teacher_forced and DESIGN.md section 4).
#include <functional>
#include <mutex>

#include "linjob.h"

namespace {

// ----------------------------------------------------------------------------- LSA
// Whole attention stage in ONE launch: energies (location conv and L projection on
// v_mfma_f32_16x16x4_f32, tanh, v-dot), then a flash-style split softmax: every (batch row,
// 32-position chunk) block publishes its local max / sum / unnormalised weights / partial
// context with write-through (sc1) stores, takes a ticket on the row's counter, and the block
// that draws the last ticket combines the chunks into scores, LSA state and context.
// Replaces tacotron.py:142-156 (LSA.forward) and :206-207 (context) for one step.
constexpr int kAT = 32;       // positions per block (16 or 32; 16 measured slower: the per-block
                              // operand fetch (L fragments, conv taps) is paid twice as often)
constexpr int kMT = kAT / 16; // MFMA position tiles per block
constexpr int kEP = kAT / 4;  // context positions per wave
constexpr int kMaxChunks = 128; // T <= 4096
static_assert(kAT == 32 && kFilters == 32, "the conv-on-MFMA phase maps its 4 waves onto 2 x 2 tiles of 16");
constexpr int kFS = 36;       // LDS row stride of the conv features: conflict-free ds_read_b128 A fragments
// What a (batch row, chunk) block publishes for the combine: one record of its own, a multiple of 128 bytes, written
// with 16-byte write-through stores (a 4-byte sc1 store is a fabric write of its own: 6x the time per byte):
//   [0, 256) partial context | [256, 288) unnormalised weights | [288] chunk max, [289] chunk sum | padding
constexpr int kRec = 320;
constexpr int kCtrStride = 32;  // one ticket counter per 128-byte line: neighbouring rows' atomics would serialise on a shared line
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword sc1
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dword sc1
}

struct AttnJob {
  const float *enc_p, *enc_q, *conv_t /*[2][31][32]*/, *L_w /*[256][32]*/, *L_b, *v_w;
  Act pq;
  float *cum, *att, *scores_out;
  long score_stride;
  Act ctx_out;
  float* part /*[B][chunks][kRec] published chunk records*/;
  unsigned* counter /*[B][kCtrStride], zero between launches*/;
  int first, T, B, chunks;
  int defer = 0;  // 1: publish the chunk's record and return; the NEXT launch combines (CombineJob, the GRU's combined source)
  // optional hand-over of the context to jobs of the SAME launch (LinJob's HAND source): tagged granule pairs in EpiGru's layout
  unsigned long long* hand = nullptr;
  unsigned hand_tag = 0;
  __host__ __device__ int nblocks() const { return chunks * B; }
  static constexpr size_t lds_bytes() {
    return sizeof(float) * (kAT * kFS + 64 * kFilters + 4 * kD + 4 * 64 + 4 * kAT + kAT + 4 + kMaxChunks);
  }

  __device__ __forceinline__ void run(int bid, char* smem) const {
    float* f = reinterpret_cast<float*>(smem);   // [kAT][kFS] conv features
    float* cwl = f + kAT * kFS;                   // [64][32] conv taps, rows 2*kTaps.. zero
    float* red = cwl + 64 * kFilters;             // [4][256] cross-wave context partials
    float* loc = red + 4 * kD;                    // [2][64]: window t0-15 .. t0+46 (+2 pad); [2][64] more, unused,
                                                  // so that every thread stores (keeps the loads unconditional)
    float* upart = loc + 4 * 64;                  // [4][kAT]
    float* wsh = upart + 4 * kAT;                 // [kAT] unnormalised weights of the chunk
    float* misc = wsh + kAT;                      // [0],[1]=chunk max / sum, [2]=is_last, [4+c]=scale of chunk c
    const int b = bid / chunks, c = bid - b * chunks, t0 = c * kAT, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, j = lane & 15, s = lane >> 4;
    constexpr int kHalf = (kTaps - 1) / 2;
    const RowId rb{b, 0, b};
    STAMP(bid, 0);

    // ---- 1. location window and conv taps: the first loads in program order, so that the in-order
    //         vmcnt wait in front of their LDS writes does not cover the big register operands below
    float locv = 0.0f, loc_mask = 0.0f;
    float4 tap0 = make_float4(0.f, 0.f, 0.f, 0.f), tap1 = tap0;
    constexpr int n4 = 2 * kTaps * kFilters / 4;
    if (!first) {  // unconditional loads from clamped indices (see linjob.h), selected afterwards
      const int ch = (tid >> 6) & 1, p = tid & 63;
      const int tp = t0 + p - kHalf;
      locv = (ch == 0 ? cum : att)[(size_t)b * T + min(max(tp, 0), T - 1)];
      // applied by multiplication at the LDS write (the state is finite): a select would let the compiler
      // sink the load under the condition again
      loc_mask = (p < kAT + kTaps - 1 && tp >= 0 && tp < T) ? 1.0f : 0.0f;
      tap0 = reinterpret_cast<const float4*>(conv_t)[tid];
      tap1 = reinterpret_cast<const float4*>(conv_t)[min(tid + 256, n4 - 1)];
    }
    // ---- 2. shared operands of the projection D[d][position] = L . f^T in MFMA layouts: accumulator element
    //         (mt, n, reg) of this lane is attention dim 64*wave + 16*n + 4*s + reg, position t0 + 16*mt + j
    float4 lf[4][2], base[4], vv[4];  // A operand: L[d = 64w + 16n + j][fl = 8*s + m], m = 0..7
    const float* pqrow = pq.rowbase(rb);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const float* lrow = L_w + (size_t)(64 * wave + 16 * n + j) * kFilters + 8 * s;
      lf[n][0] = reinterpret_cast<const float4*>(lrow)[0];
      lf[n][1] = reinterpret_cast<const float4*>(lrow)[1];
      const int d0 = 64 * wave + 16 * n + 4 * s;
      const float4 q4 = *reinterpret_cast<const float4*>(pqrow + pq.eoff(d0));
      const float4 b4 = *reinterpret_cast<const float4*>(L_b + d0);
      base[n] = make_float4(q4.x + b4.x, q4.y + b4.y, q4.z + b4.z, q4.w + b4.w);
      vv[n] = *reinterpret_cast<const float4*>(v_w + d0);
    }
    // ---- 3. location conv (cross-correlation, no bias) as one 32 x 32 x 64 product on the matrix cores:
    //         f[p][fl] = sum_kk cw[kk][fl] * loc[ch(kk)][p + k(kk)], kk = 31*ch + k padded to 64 with zero taps.
    //         Wave w owns filter tile w&1 and position tile w>>1.
    if (!first) {  // at t == 0 the location features are all zero (tacotron.py:135-136)
      loc[tid] = locv * loc_mask;
      reinterpret_cast<float4*>(cwl)[tid] = tap0;
      if (tid + 256 < n4) reinterpret_cast<float4*>(cwl)[tid + 256] = tap1;
      if (tid < 2 * kFilters) cwl[2 * kTaps * kFilters + tid] = 0.0f;  // taps kk = 62, 63
    }
    STAMP(bid, 1);
    lds_barrier();
    // the per-position memory operands (enc_p now, enc_q after the conv) are issued behind the small
    // shared ones: they fly during the conv and the L projection and are first needed by the tanh
    float4 ep[kMT][4];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      const int t = t0 + 16 * mt + j;
      // Unconditional loads from a clamped row (positions >= T are masked out of the softmax below, so
      // their operand only has to be finite): a load under `t < T ? .. : ..`, or one whose value is
      // only selected under it, compiles to a divergent branch with a full vmcnt(0) wait inside, which
      // serialises every load of the phase.
      const float* row = enc_p + ((size_t)b * T + min(t, T - 1)) * kD + 64 * wave + 4 * s;
#pragma unroll
      for (int n = 0; n < 4; ++n) ep[mt][n] = *reinterpret_cast<const float4*>(row + 16 * n);
    }
    {
      const int ft = wave & 1, pt = wave >> 1;
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (!first) {
        float av[16], bw[16];  // all LDS operands first, then the MFMA chain (no LDS round trip per step)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int kk = 4 * q + s;
          const int ch = kk >= kTaps ? 1 : 0, k = kk - kTaps * ch;
          av[q] = cwl[kk * kFilters + 16 * ft + j];   // A[filter 16ft + j][kk]
          bw[q] = loc[ch * 64 + 16 * pt + j + k];      // B[kk][position 16pt + j]
        }
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bw[q], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q + 1], bw[q + 1], c1, 0, 0, 0);
        }
      }
      // D lane: position 16pt + j, filters 16ft + 4s .. +3
      *reinterpret_cast<f32x4*>(f + (16 * pt + j) * kFS + 16 * ft + 4 * s) = c0 + c1;
    }
    lds_barrier();
    STAMP(bid, 2);
    // ---- 4. context operand: thread = (4 context dims 4*(tid&63).., 8 positions of wave tid>>6);
    //         lands during the MFMA / tanh phase
    float4 eq[kEP];
#pragma unroll
    for (int i = 0; i < kEP; ++i) {
      const int t = t0 + kEP * wave + i;
      // clamped row, weight 0 for t >= T (see the note on enc_p above)
      eq[i] = *reinterpret_cast<const float4*>(enc_q + ((size_t)b * T + min(t, T - 1)) * kD + 4 * lane);
    }
    // ---- 5. L projection on the matrix cores (8 independent accumulator chains), tanh, v-dot
    f32x4 acc[kMT][4];
    {
      float bv[kMT][8];
#pragma unroll
      for (int mt = 0; mt < kMT; ++mt) {
        const float4 b0 = *reinterpret_cast<const float4*>(f + (16 * mt + j) * kFS + 8 * s);  // B: f[pos][8s+m]
        const float4 b1 = *reinterpret_cast<const float4*>(f + (16 * mt + j) * kFS + 8 * s + 4);
        bv[mt][0] = b0.x, bv[mt][1] = b0.y, bv[mt][2] = b0.z, bv[mt][3] = b0.w;
        bv[mt][4] = b1.x, bv[mt][5] = b1.y, bv[mt][6] = b1.z, bv[mt][7] = b1.w;
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[mt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float av = m < 4 ? (m == 0 ? lf[n][0].x : m == 1 ? lf[n][0].y : m == 2 ? lf[n][0].z : lf[n][0].w)
                                 : (m == 4 ? lf[n][1].x : m == 5 ? lf[n][1].y : m == 6 ? lf[n][1].z : lf[n][1].w);
#pragma unroll
          for (int mt = 0; mt < kMT; ++mt)
            acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[mt][m], acc[mt][n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt) {
      float a = 0.0f;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        a = fmaf(vv[n].x, tanhf_(acc[mt][n][0] + base[n].x + ep[mt][n].x), a);
        a = fmaf(vv[n].y, tanhf_(acc[mt][n][1] + base[n].y + ep[mt][n].y), a);
        a = fmaf(vv[n].z, tanhf_(acc[mt][n][2] + base[n].z + ep[mt][n].z), a);
        a = fmaf(vv[n].w, tanhf_(acc[mt][n][3] + base[n].w + ep[mt][n].w), a);
      }
      a += __shfl_xor(a, 16, 64);  // sum over the 4 k-slot groups: same position, other dims
      a += __shfl_xor(a, 32, 64);
      if (s == 0) upart[wave * kAT + 16 * mt + j] = a;
    }
    lds_barrier();
    STAMP(bid, 3);
    // ---- 6. chunk-local softmax statistics (wave 0), unnormalised weights w = exp(u - m_c)
    if (wave == 0) {
      const int tt = lane & (kAT - 1);
      const bool ok = t0 + tt < T;
      const float u = (upart[tt] + upart[kAT + tt]) + (upart[2 * kAT + tt] + upart[3 * kAT + tt]);
      float m = ok ? u : -INFINITY;
#pragma unroll
      for (int off = kAT / 2; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kAT));
      const float w = ok ? __expf(u - m) : 0.0f;
      float l = w;
#pragma unroll
      for (int off = kAT / 2; off >= 1; off >>= 1) l += __shfl_xor(l, off, kAT);
      // LDS only here: the chunk's statistics are published together with its partial context below.
      // (A store issued now would be younger than the enc_q loads still in flight; the compiler cannot
      // count stores under a branch and would drain them, write-through round trip included, before
      // the context phase may touch its operands.)
      if (lane < kAT) wsh[tt] = w;
      if (lane == 0) misc[0] = m, misc[1] = l;
    }
    lds_barrier();
    STAMP(bid, 4);
    // ---- 7. partial context of the chunk: per wave over its 8 positions, then across waves
    {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < kEP; ++i) {
        const float w = wsh[kEP * wave + i];
        a.x = fmaf(w, eq[i].x, a.x), a.y = fmaf(w, eq[i].y, a.y), a.z = fmaf(w, eq[i].z, a.z), a.w = fmaf(w, eq[i].w, a.w);
      }
      *reinterpret_cast<float4*>(red + wave * kD + 4 * lane) = a;
    }
    lds_barrier();
    {
      float* rec = part + ((size_t)b * chunks + c) * kRec;
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, kRec * 4, 0x00020000);
      f32x4 v;
      bool store = false;
      if (tid < 64) {
        const float* r4 = red + 4 * tid;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (r4[e] + r4[kD + e]) + (r4[2 * kD + e] + r4[3 * kD + e]);
        store = true;
      } else if (tid < 64 + kAT / 4) {
        v = *reinterpret_cast<const f32x4*>(wsh + 4 * (tid - 64));
        store = true;
      } else if (tid == 64 + kAT / 4) {
        v = f32x4{misc[0], misc[1], 0.f, 0.f};
        store = true;
      }
      if (defer) {  // consumed across a kernel boundary: plain stores, no ticket, nothing to wait for
        if (store) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, tid * 16, 0, 0);
        return;
      }
      if (store) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, tid * 16, 0, 16 /*sc1: write-through*/);
    }
    // ---- 8. ticket: every storing wave drains its write-through stores, then ONE lane adds
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    STAMP(bid, 5);
    if (tid == 0) {
      unsigned* ctr = counter + (size_t)b * kCtrStride;
      const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)chunks - 1;
      if (last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      misc[2] = last ? 1.0f : 0.0f;
    }
    __syncthreads();
    STAMP(bid, 6);
    if (misc[2] == 0.0f) return;
    // ---- 9. last block of the row: combine the chunks (every load of published data is sc1).  The payload of the
    //         first four chunks does not depend on the chunk scales: it is requested first, from clamped (always
    //         valid) chunk indices, so that it flies together with the statistics instead of one round trip later.
    const float* rec0 = part + (size_t)b * chunks * kRec;
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rec0), 0, chunks * kRec * 4, 0x00020000);
    const int t_first = min(tid, T - 1);
    const float pw0 = ld_sc1(rec0 + (size_t)(t_first / kAT) * kRec + kD + (t_first % kAT));
    const float cum0 = cum[(size_t)b * T + t_first];  // the thread's first position of the state update below
    f32x4 pv[4] = {};  // wave 0: context dims 4 tid .. 4 tid + 3 of chunks 0..3
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (min(i, chunks - 1) * kRec + 4 * tid) * 4, 0, 16));
    }
    if (tid < 64) {  // up to kMaxChunks = 128 chunks: two per lane
      float m[2], l[2];
      bool ok[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int cc = tid + 64 * h;
        ok[h] = cc < chunks;
        const float* ml = rec0 + (size_t)min(cc, chunks - 1) * kRec + kD + kAT;
        m[h] = ok[h] ? ld_sc1(ml) : -INFINITY;
        l[h] = ok[h] ? ld_sc1(ml + 1) : 0.0f;
      }
      float M = fmaxf(m[0], m[1]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
      const float sc0 = ok[0] ? __expf(m[0] - M) : 0.0f, sc1v = ok[1] ? __expf(m[1] - M) : 0.0f;
      float Ls = add_rn(mul_rn(l[0], sc0), mul_rn(l[1], sc1v));
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) Ls += __shfl_xor(Ls, off, 64);
      if (ok[0]) misc[4 + tid] = sc0 / Ls;  // scale of the chunk, normaliser folded in
      if (ok[1]) misc[4 + 64 + tid] = sc1v / Ls;
    }
    lds_barrier();
    if (tid < 64) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      const auto acc4 = [&](const f32x4 v, float sc) {  // explicit fma chain: the GRU's combined source repeats it bit for bit
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fmaf(v[e], sc, a[e]);
      };
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < chunks) acc4(pv[i], misc[4 + i]);
      for (int c0 = 4; c0 < chunks; c0 += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, (min(c0 + i, chunks - 1) * kRec + 4 * tid) * 4, 0, 16));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (c0 + i < chunks) acc4(pv[i], misc[4 + c0 + i]);
      }
      *reinterpret_cast<f32x4*>(ctx_out.rowbase(rb) + ctx_out.eoff(4 * tid)) = a;
      if (hand) {
        using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
        unsigned long long* dst = hand + (((size_t)(b >> 4) * 128 + 2 * tid) * 16 + (b & 15)) * 2;
        const u32x4_ g0 = {__float_as_uint(a[0]), hand_tag, __float_as_uint(a[1]), hand_tag};
        const u32x4_ g1 = {__float_as_uint(a[2]), hand_tag, __float_as_uint(a[3]), hand_tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst), "v"(g0) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(dst + 32), "v"(g1) : "memory");
      }
    }
    for (int t = tid; t < T; t += 256) {
      // (mul_rn: the product is rounded before it is added to cum; left to the compiler, one of the two combine paths
      //  may fuse it into the add and the other not, a last-bit difference between them)
      const float p = mul_rn(t == tid ? pw0 : ld_sc1(rec0 + (size_t)(t / kAT) * kRec + kD + (t % kAT)), misc[4 + t / kAT]);
      const size_t o = (size_t)b * T + t;
      scores_out[(size_t)b * score_stride + t] = p;
      att[o] = p;
      cum[o] = first ? p : (t == tid ? cum0 : cum[o]) + p;
    }
    STAMP(bid, 7);
  }
};

// Combine of the chunk records of the PREVIOUS launch's attention (defer = 1), one block per batch row, up to 8 chunks:
// scores, LSA state and context of that step.  Rides as the first job of the next step's GRU launch: the kernel boundary
// is the hand-over, so the attention launch carries no ticket seam (2.9 us: drain of the write-through stores, atomic,
// combine loads) and everything here is plain loads.  The GRU of that launch does not wait for this job: it forms the
// context itself from the same records (LinJob's combined source).
struct CombineJob {
  const float* part;
  float *cum, *att, *scores_out;
  long score_stride;
  Act ctx_out;
  int first, T, B, chunks;
  __host__ __device__ int nblocks() const { return B; }
  static constexpr size_t lds_bytes() { return sizeof(float) * 8; }
  __device__ __forceinline__ void run(int bid, char* smem) const {
    float* sc = reinterpret_cast<float*>(smem);
    const int b = bid, tid = threadIdx.x;
    const float* rec0 = part + (size_t)b * chunks * kRec;
    const RowId rb{b, 0, b};
    f32x4 pv[8];
    if (tid < 64) {
#pragma unroll
      for (int c = 0; c < 8; ++c) pv[c] = *reinterpret_cast<const f32x4*>(rec0 + (size_t)min(c, chunks - 1) * kRec + 4 * tid);
    }
    float2 ml[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) ml[c] = *reinterpret_cast<const float2*>(rec0 + (size_t)min(c, chunks - 1) * kRec + kD + kAT);
    float scale[8];
    chunk_scales<8>(ml, chunks, scale);
    if (tid < 8) sc[tid] = scale[tid];
    if (tid < 64) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < chunks) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = fmaf(pv[c][e], scale[c], a[e]);
        }
      *reinterpret_cast<f32x4*>(ctx_out.rowbase(rb) + ctx_out.eoff(4 * tid)) = a;
    }
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
      const float p = mul_rn(rec0[(size_t)(t / kAT) * kRec + kD + (t % kAT)], sc[t / kAT]);
      const size_t o = (size_t)b * T + t;
      scores_out[(size_t)b * score_stride + t] = p;
      att[o] = p;
      cum[o] = first ? p : cum[o] + p;
    }
  }
};

// Query projection INSIDE the attention GRU's launch: block (row group rg, tile) waits for the new attention state of
// its 16 batch rows -- the granule pairs that the GRU job's epilogue publishes (EpiGru::hand: the data is the flag, tag =
// step + 1, array zeroed when a loop starts) -- stages it in LDS and projects it.  Its blocks follow the GRU's in the
// launch, so every producer has been dispatched before the first consumer starts to wait; the wait is bounded (a sweep
// that never completes raises the error word and poisons the block's output with NaN).  The weights are requested
// before the wait and the arithmetic repeats LinJob<1,1,EpiLinear> chunk for chunk (wave w: K chunks w, w + 4, ..;
// one accumulator chain; (r0 + r1) + (r2 + r3)), so the result has the bits of the stand-alone query launch of the
// step API.  What it buys: the 3.2 us query launch of the teacher-forced loop disappears behind the riding LSTM cell.
struct QueryJob {
  const float4* wt;   // [tiles][KC = 16][64] float4, fragment order (linjob.h)
  const float* bias;  // tile order
  const unsigned long long* hand;
  unsigned* err;
  Act y;
  unsigned tag;
  int rows, tiles, groups;
  int sentinel;  // 1: wait on one pair per producer first (see run); 0: sweep everything at once (nothing rides beside the job)
  static constexpr int kXS = 260;  // LDS row stride of the staged state (floats): conflict-free float4 B fragments
  __host__ __device__ int nblocks() const { return tiles * groups; }
  static constexpr size_t lds_bytes() { return sizeof(float) * 16 * kXS + sizeof(f32x4) * 4 * 64; }
  __device__ __forceinline__ void run(int bid, char* smem) const {
    float* xs = reinterpret_cast<float*>(smem);                                          // [16 rows][kXS]
    f32x4(*red)[64] = reinterpret_cast<f32x4(*)[64]>(smem + sizeof(float) * 16 * kXS);   // [4 waves][64]
    const int tile = bid % tiles, rg = bid / tiles, row0 = rg * 16, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, s = lane >> 4;
    // weights and bias first: they fly while the block waits
    float4 w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = wt[((size_t)tile * 16 + wave + 4 * q) * 64 + lane];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + tile * 16 + s * 4);
    // sweep: pair i = (unit pair i / 16, row i % 16) of the row group, 8 pairs per thread
    using u32x4_ = __attribute__((ext_vector_type(4))) unsigned;
    const unsigned long long* g = hand + (size_t)rg * 128 * 16 * 2;
    const bool live = row0 + (tid & 15) < rows;   // rows past the batch are never published
    u32x4_ v[8];
    bool failed = false;
    // Wait on ONE pair per producing block first (wave 0: the first pair of each of the 64 GRU blocks of the row group,
    // 1 KB per sweep): polling the whole 32 KB from all 64 blocks of this job takes the fill bandwidth that the GRU and
    // the riding LSTM cell are bound by (22.1 us per step instead of 21.4; four tiles per block, so that only 16 blocks
    // sweep, is slower still, 22.6: a wave's 64 fp32 MFMAs take 0.85 us).  The full sweep below then checks every tag
    // (the lines of one block's store can land apart).
    if (sentinel && wave == 0) {
      const unsigned long long* sg = g + (size_t)lane * 2 * 16 * 2;
      u32x4_ sv;
      for (unsigned spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(sv) : "v"(sg) : "memory");
        if (__all(sv.y == tag && sv.w == tag) || spins > (1u << 16)) break;
      }
    }
    __syncthreads();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
      // the eight loads and their wait in ONE asm statement (early-clobber outputs): between separate statements the compiler
      // may copy or spill a destination register before the data has arrived
      const unsigned long long* g0 = g + 2 * tid;
      asm volatile(
          "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
          "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
          "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
          "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(g0), "v"(g0 + 512), "v"(g0 + 1024), "v"(g0 + 1536), "v"(g0 + 2048), "v"(g0 + 2560), "v"(g0 + 3072), "v"(g0 + 3584)
          : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) ok &= !live || (v[i].y == tag && v[i].w == tag);
      if (__all(ok)) break;
      if (spins > (1u << 16)) {
        if (lane == 0) atomicOr(err, 1u);
        failed = true;
        break;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, up = idx >> 4, row = idx & 15;
      float2 hv = make_float2(__uint_as_float(v[i].x), __uint_as_float(v[i].z));
      if (!live) hv = make_float2(0.f, 0.f);
      if (failed) hv = make_float2(NAN, NAN);
      *reinterpret_cast<float2*>(xs + row * kXS + 2 * up) = hv;
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 x = *reinterpret_cast<const float4*>(xs + j * kXS + (wave + 4 * q) * 16 + s * 4);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].x, x.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].y, x.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].z, x.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[q].w, x.w, acc, 0, 0, 0);
    }
    red[wave][lane] = acc;
    lds_barrier();
    if (wave == 0 && row0 + j < rows) {
      f32x4 r = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = r[e] + bv[e];
      const RowId rr{row0 + j, 0, row0 + j};
      *reinterpret_cast<f32x4*>(y.rowbase(rr) + y.eoff(tile * 16 + s * 4)) = r;
    }
  }
};

// ----------------------------------------------------------------------------- loop helpers
// xp (step-major interleaved [n][P/4][B][4]): xp(k, b, c) = mel_in[b][c][k*r - 1], zeros for k == 0:
// the teacher-forcing inputs of every step (tacotron.py:340)
__global__ void gather_prenet_in_kernel(const float* __restrict__ mel_in, float* __restrict__ xp, int B,
                                        int P, int S, int r, int n) {
  const size_t total = (size_t)n * B * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i enumerates the destination: (k, c4, b, e)
    const int e = i & 3;
    const size_t i4 = i >> 2;
    const int b = i4 % B;
    const size_t r4 = i4 / B;
    const int c = (int)(r4 % (P / 4)) * 4 + e, k = (int)(r4 / (P / 4));
    xp[i] = k == 0 ? 0.0f : mel_in[((size_t)b * P + c) * S + (size_t)k * r - 1];
  }
}

// aligner features of every step, step-major interleaved and zero-padded to A4 columns:
// out(k, c, b) = ali[b][c][k]  (ali is [B, A, n])
__global__ void gather_aligner_kernel(const float* __restrict__ ali, float* __restrict__ out, int B, int A, int A4, int n) {
  const size_t total = (size_t)n * B * A4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 3;
    const size_t i4 = i >> 2;
    const int b = (int)(i4 % B);
    const size_t r4 = i4 / B;
    const int c = (int)(r4 % (A4 / 4)) * 4 + e, k = (int)(r4 / (A4 / 4));
    out[i] = c < A ? ali[((size_t)b * A + c) * n + k] : 0.0f;
  }
}

}  // namespace


struct wrnn_decoder {
  int device = 0;
  int n_mels = 0, d = 0, l = 0, A = 0, P = 0;
  Layer fc1, fc2, gru, pq, rin, l1, l2;
  Layer mel[kMaxR + 1];
  std::vector<float> mel_w_host;  // [n_mels*20][l] for the lazy per-r gather
  // free-running decode: fc1 folded through the mel projection of a group's last frame ([256 x l], per r)
  // and fc1's aligner-feature columns ([256 x A] + b1), see wrnn_decoder_run_free
  std::vector<float> fc1_w_host, fc1_b_host;
  Layer fc1m[kMaxR + 1];
  Layer fc1a;
  // free-running decode (ensure_free_layers): LSTM1 folded through rnn_input, LSTM2 over [y | h1 | h2]
  std::vector<float> rin_w, rin_b, l1_wih, l1_whh, l1_bih, l1_bhh, l2_wih, l2_whh, l2_bih, l2_bhh;
  Layer fr_l1f, fr_l2x;
  Layer gru_h;             // the attention GRU with K = [ctx | attn_h | prenet output] (its bias is gru.bias)
  Layer fr_l1h, fr_rinh;   // the same LSTM1 / rnn_input with the context LAST in K: [attn_h | h1 | ctx], [attn_h | ctx] (hand source)
  bool fr_ready = false;
  std::vector<float> l1_wf_host, l1_bf_host;  // LSTM1 folded through rnn_input: [4l][2d] and [4l] (ensure_l1_fold)
  std::vector<float> fc1m_host[kMaxR + 1];    // fc1 o mel_proj(last frame of a group of r): [256][l]
  // small-batch free-running decode as one persistent launch (taco_resident.h): raw row-major device copies
  std::vector<float> gru_wih_h, gru_whh_h, gru_bih_h, gru_bhh_h, pq_w_h, pq_b_h, fc2_w_h, fc2_b_h;
  struct Resident {
    float *fc1_b = nullptr, *fc2_w = nullptr, *fc2_b = nullptr, *gru_wih = nullptr, *gru_whh = nullptr, *gru_bih = nullptr,
          *gru_bhh = nullptr, *pq_w = nullptr, *pq_b = nullptr, *l1_wf = nullptr, *l1_whh = nullptr, *l1_b = nullptr,
          *rin_w = nullptr, *rin_b = nullptr, *l2_wih = nullptr, *l2_whh = nullptr, *l2_b = nullptr;
    float* fc1m[kMaxR + 1] = {};
    float* mel_w[kMaxR + 1] = {};
    unsigned long long* gran = nullptr;  // granule arrays + status words, zeroed on the stream before every launch
    bool ready = false;
  } res;
  int free_path = WRNN_FREE_PATH_AUTO;
  int tf_path = WRNN_TF_PATH_AUTO;
  int n_cu = 0;
  int last_stack_launches = 0;   // launches of the LSTM stack in the last chain-path call (profiling)
  float *conv_t = nullptr, *L_w = nullptr, *L_b = nullptr, *v_w = nullptr;
  float* ws = nullptr;
  size_t ws_floats = 0;
  int* flags = nullptr;  // free-run stop votes
  int flags_n = 0;
  unsigned* row_counter = nullptr;  // [B] attention tickets, zero between launches
  unsigned long long* qhand = nullptr;  // GRU -> query hand-over of the teacher-forced loop (QueryJob), + an error word
  int qhand_groups = 0;
  int row_counter_n = 0;
  // teacher-forced loop with the attention recurrence as one persistent launch (taco_chain.h)
  float *chain_pq_w = nullptr, *chain_pq_b = nullptr;
  unsigned long long* chain_gran = nullptr;  // H, QS, R, C pair arrays + error word, zeroed on the stream before every launch
  hipEvent_t chain_ev[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> chain_chunk_ev;    // one per chain launch of a call
  hipEvent_t chain_prof[2] = {nullptr, nullptr};   // timing events around the chain launch (profiling)
  // ... with the LSTM stack as persistent weights-resident launches beside it (taco_stack.h)
  float* stack_buf = nullptr;        // x1 ring, both cells' h rings and cell states (sized for B = 64)
  unsigned* stack_flags = nullptr;   // [2][128] steps done + error word
  hipEvent_t stack_ev[3] = {nullptr, nullptr};
  // optional per-launch timing of the teacher-forced loop (wrnn_decoder_set_profiling)
  bool profiling = false;
  std::vector<hipEvent_t> prof_events;
  std::vector<int> prof_kind;  // kind of the launch between events i and i+1
  float prof_ms[WRNN_PROFILE_KINDS] = {};
  int prof_count[WRNN_PROFILE_KINDS] = {};
};

#include "taco_resident.h"
#include "taco_chain.h"
#include "taco_stack.h"

namespace {

int ensure_ws(wrnn_decoder* dec, size_t floats) {
  if (floats <= dec->ws_floats) return 0;
  if (dec->ws) {
    WRNN_HIP(hipDeviceSynchronize());
    WRNN_HIP(hipFree(dec->ws));
    dec->ws = nullptr;
    dec->ws_floats = 0;
  }
  WRNN_HIP(hipMalloc((void**)&dec->ws, floats * sizeof(float)));
  dec->ws_floats = floats;
  return 0;
}

// The attention tickets must read zero when a call's first attention launch runs.  They are zeroed on the CALL'S
// stream at every entry (not with a blocking null-stream memset at allocation: the caller's stream may be a
// non-blocking one, which is not ordered behind null-stream work), which also re-arms them after a launch that
// aborted half way.  Allocated in multiples of 64 words so that the memset is a whole number of 16-byte pieces.
int ensure_counters(wrnn_decoder* dec, int B, hipStream_t st, bool rearm = true) {
  B *= kCtrStride;  // one counter per 128-byte line
  if (B > dec->row_counter_n) {
    rearm = true;
    if (dec->row_counter) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(dec->row_counter));
      dec->row_counter = nullptr;
      dec->row_counter_n = 0;
    }
    const int n = (B + 63) / 64 * 64;
    WRNN_HIP(hipMalloc((void**)&dec->row_counter, (size_t)n * sizeof(unsigned)));
    dec->row_counter_n = n;
  }
  if (rearm) WRNN_HIP(hipMemsetAsync(dec->row_counter, 0, (size_t)dec->row_counter_n * sizeof(unsigned), st));
  return 0;
}

// In-launch hand-over arrays: 128 unit pairs x 16 rows of granule pairs per row group, three regions (GRU -> query of both
// loops; fc1 -> fc2, attention -> rnn_input / LSTM1 and fc2 -> GRU of the free-running loop), then one error word; zeroed (tags and
// error) on the call's stream
constexpr size_t kQHandGroup = (size_t)128 * 16 * 2;   // granules per row group
constexpr int kQHandRegions = 4;
inline unsigned long long* qhand_region(wrnn_decoder* dec, int r) { return dec->qhand + (size_t)r * dec->qhand_groups * kQHandGroup; }
inline unsigned* qhand_err(wrnn_decoder* dec) { return reinterpret_cast<unsigned*>(qhand_region(dec, kQHandRegions)); }
int ensure_qhand(wrnn_decoder* dec, int B, hipStream_t st) {
  const int groups = (B + 15) / 16;
  if (groups > dec->qhand_groups) {
    if (dec->qhand) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(dec->qhand));
      dec->qhand = nullptr, dec->qhand_groups = 0;
    }
    WRNN_HIP(hipMalloc((void**)&dec->qhand, (kQHandRegions * groups * kQHandGroup + 1) * sizeof(unsigned long long)));
    dec->qhand_groups = groups;
  }
  WRNN_HIP(hipMemsetAsync(dec->qhand, 0, (kQHandRegions * dec->qhand_groups * kQHandGroup + 1) * sizeof(unsigned long long), st));
  return 0;
}

int ensure_mel_layer(wrnn_decoder* dec, int r) {
  Layer& L = dec->mel[r];
  if (L.wt) return 0;
  const int N = dec->n_mels * r, l = dec->l;
  const float* W = dec->mel_w_host.data();
  RowFn w = [&](int t, int i, int k) -> float {
    const int j = t * 16 + i;
    if (j >= N) return 0.0f;
    const int m = j / r, f = j % r;
    return W[(size_t)(m * kMaxR + f) * l + k];
  };
  return build_layer(L, (N + 15) / 16, l, N, w, nullptr);
}

// fc1 o mel_proj for the last frame of a group: W'[j][k] = sum_m W1[j][m] * Wmel[m*20 + r-1][k]
void ensure_fc1_mel_host(wrnn_decoder* dec, int r) {
  std::vector<float>& Wp = dec->fc1m_host[r];
  if (!Wp.empty()) return;
  const int nm = dec->n_mels, l = dec->l, P = dec->P;
  Wp.resize((size_t)kFc1 * l);
  for (int j = 0; j < kFc1; ++j)
    for (int k = 0; k < l; ++k) {
      double acc = 0.0;
      for (int m = 0; m < nm; ++m)
        acc += (double)dec->fc1_w_host[(size_t)j * P + m] * (double)dec->mel_w_host[(size_t)(m * kMaxR + r - 1) * l + k];
      Wp[(size_t)j * l + k] = (float)acc;
    }
}

int ensure_fc1_mel_layer(wrnn_decoder* dec, int r) {
  Layer& L = dec->fc1m[r];
  if (L.wt) return 0;
  const int l = dec->l;
  ensure_fc1_mel_host(dec, r);
  const std::vector<float>& Wp = dec->fc1m_host[r];
  RowFn w = [&](int t, int i, int k) -> float { return Wp[(size_t)(t * 16 + i) * l + k]; };
  return build_layer(L, kFc1 / 16, l, kFc1, w, nullptr);
}

// ---- free-running decode: LSTM cell with up to two residual inputs (x1 = y + h1 is never formed), eval mode
struct EpiLstm2 {  // tile rows: 4 units x (i, f, g, o)
  const float* bias;
  Act c_prev, c_out, h_out;
  Act x_in, x_in2, x_out;  // optional residual: x_out = x_in + x_in2 + h (tacotron.py:219,227)
  struct Pre { f32x4 b; float c, x; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    const int U = tile * 4 + u;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.c = c_prev.rowbase(row)[c_prev.eoff(U)];
    q.x = x_out.p ? x_in.rowbase(row)[x_in.eoff(U)] + x_in2.rowbase(row)[x_in2.eoff(U)] : 0.0f;
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const int U = tile * 4 + u;
    const float ig = sigmoidf_(v[0] + q.b[0]);
    const float fg = sigmoidf_(v[1] + q.b[1]);
    const float gg = tanhf_(v[2] + q.b[2]);
    const float og = sigmoidf_(v[3] + q.b[3]);
    const float c = fg * q.c + ig * gg;
    const float h = og * tanhf_(c);
    c_out.rowbase(row)[c_out.eoff(U)] = c;
    h_out.rowbase(row)[h_out.eoff(U)] = h;
    if (x_out.p) x_out.rowbase(row)[x_out.eoff(U)] = q.x + h;
  }
};

// LSTM1 folded through rnn_input on the host (0.5 G multiply-adds, once per handle):
// W' = W_ih1 . W_rin [4l x 2d], b' = W_ih1 . b_rin + b_ih1 + b_hh1
void ensure_l1_fold(wrnn_decoder* dec) {
  if (!dec->l1_wf_host.empty()) return;
  const int d = dec->d, l = dec->l, K2 = 2 * d;
  std::vector<float>& Wf = dec->l1_wf_host;
  std::vector<float>& bf = dec->l1_bf_host;
  Wf.resize((size_t)4 * l * K2), bf.resize((size_t)4 * l);
  std::vector<double> acc(K2);
  const float *W1 = dec->l1_wih.data(), *Wr = dec->rin_w.data(), *br = dec->rin_b.data();
  for (int n = 0; n < 4 * l; ++n) {
    std::fill(acc.begin(), acc.end(), 0.0);
    double ab = 0.0;
    for (int j = 0; j < l; ++j) {
      const double wv = W1[(size_t)n * l + j];
      const float* row = Wr + (size_t)j * K2;
      for (int k = 0; k < K2; ++k) acc[k] += wv * (double)row[k];
      ab += wv * (double)br[j];
    }
    for (int k = 0; k < K2; ++k) Wf[(size_t)n * K2 + k] = (float)acc[k];
    bf[n] = (float)(ab + (double)dec->l1_bih[n] + (double)dec->l1_bhh[n]);
  }
}

// Layers of the free-running loop, built on first use.
int ensure_free_layers(wrnn_decoder* dec) {
  if (dec->fr_ready) return 0;
  const int d = dec->d, l = dec->l;
  int rc = 0;
  auto lrow = [l](int t, int i) { return (i & 3) * l + t * 4 + (i >> 2); };  // tile row -> torch gate-major row
  {  // LSTM1 folded through rnn_input: K = [ctx | attn_h | h1], W' = W_ih1 . W_rin [4l x 2d], b' = W_ih1 . b_rin + b_ih1 + b_hh1
    const int K2 = 2 * d;
    ensure_l1_fold(dec);
    const float* Wh = dec->l1_whh.data();
    const float *Wfp = dec->l1_wf_host.data(), *bfp = dec->l1_bf_host.data();
    RowFn w = [=](int t, int i, int k) -> float {
      return k < K2 ? Wfp[(size_t)lrow(t, i) * K2 + k] : Wh[(size_t)lrow(t, i) * l + (k - K2)];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float { return bfp[lrow(t, i)]; };
    rc |= build_layer(dec->fr_l1f, l / 4, K2 + l, 4 * l, w, &b);
    RowFn wh = [=](int t, int i, int k) -> float {   // [attn_h | h1 | ctx]
      const size_t row = (size_t)lrow(t, i);
      return k < d ? Wfp[row * K2 + d + k] : k < d + l ? Wh[row * l + (k - d)] : Wfp[row * K2 + (k - d - l)];
    };
    rc |= build_layer(dec->fr_l1h, l / 4, K2 + l, 4 * l, wh, &b);
    const float *Wr = dec->rin_w.data(), *br = dec->rin_b.data();
    RowFn wr = [=](int t, int i, int k) -> float {   // rnn_input over [attn_h | ctx]
      const int n = t * 16 + i;
      return n < l ? (k < d ? Wr[(size_t)n * K2 + d + k] : Wr[(size_t)n * K2 + (k - d)]) : 0.0f;
    };
    std::function<float(int, int)> bri = [=](int t, int i) -> float { return t * 16 + i < l ? br[t * 16 + i] : 0.0f; };
    rc |= build_layer(dec->fr_rinh, (l + 15) / 16, K2, l, wr, &bri);
  }
  {  // LSTM2 over [y | h1 | h2]: W_ih2 multiplies y and h1 (their sum is the reference's x1), W_hh2 multiplies h2
    const float *W2 = dec->l2_wih.data(), *Wh = dec->l2_whh.data(), *b1 = dec->l2_bih.data(), *b2 = dec->l2_bhh.data();
    RowFn w = [=](int t, int i, int k) -> float {
      const size_t row = (size_t)lrow(t, i) * l;
      return k < l ? W2[row + k] : k < 2 * l ? W2[row + k - l] : Wh[row + k - 2 * l];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float { return b1[lrow(t, i)] + b2[lrow(t, i)]; };
    rc |= build_layer(dec->fr_l2x, l / 4, 3 * l, 4 * l, w, &b);
  }
  if (rc) return 1;
  dec->fr_ready = true;
  return 0;
}

// Buffers of one decoder step, every activation as a view (row-major for the step API,
// interleaved for the loops).
struct StepBufs {
  const float *enc_q, *enc_p;
  Act p;                                 // prenet output [B,128]
  Act attn_h_in, attn_h_out, ctx, ctx_out;  // ctx: previous step's context (read by the GRU); ctx_out: written by the
                                            // attention, read by rnn_input (may be the same buffer as ctx)
  Act h1_in, h1_out, h2_in, h2_out;
  Act c1_in, c2_in, c1_out, c2_out;
  float *cum, *att;
  const float *zone_u1, *zone_u2;
  float* scores_out;
  long score_stride;
  Act pq, y, x1, x2;
  float* attn_scratch;  // attn_scratch_floats(B, T)
};

using GruJob = LinJob<1, 1, EpiGru>;     // 16 x 16 output tile per block: many blocks, few bytes each
using GruCombJob4 = LinJob<1, 1, EpiGru, 4, 4>;  // the same with the context formed from <= 4 / <= 8 chunk records
using GruCombJob8 = LinJob<1, 1, EpiGru, 4, 8>;
using LinearJob = LinJob<1, 1, EpiLinear>;
using LstmJob = LinJob<2, 1, EpiLstm>;   // two weight tiles share one activation tile: the activations
                                         // (read by every block) are the expensive operand

struct StepJobs {
  GruJob gru;
  LinearJob pq, rin;
  AttnJob attn;
  LstmJob l1, l2;
};

inline int attn_chunks(int T) { return (T + kAT - 1) / kAT; }
inline size_t attn_scratch_floats(int B, int T) { return (size_t)B * attn_chunks(T) * kRec; }

int make_step_jobs(wrnn_decoder* dec, const StepBufs& q, int B, int T, int first, StepJobs& J) {
  const int d = dec->d, l = dec->l;
  if (make_lin(J.gru, dec->gru, Src{q.ctx, d}, Src{q.p, kFc2}, Src{q.attn_h_in, d}, B, B,
               EpiGru{dec->gru.bias, q.attn_h_in, q.attn_h_out})) return 1;
  if (make_lin(J.pq, dec->pq, Src{q.attn_h_out, d}, kNoSrc, kNoSrc, B, B,
               EpiLinear{dec->pq.bias, q.pq, d, 0, nullptr})) return 1;
  const int chunks = attn_chunks(T);
  J.attn = AttnJob{q.enc_p, q.enc_q, dec->conv_t, dec->L_w, dec->L_b, dec->v_w, q.pq, q.cum, q.att,
                   q.scores_out, q.score_stride, q.ctx_out, q.attn_scratch, dec->row_counter, first, T, B, chunks};
  if (make_lin(J.rin, dec->rin, Src{q.ctx_out, d}, Src{q.attn_h_out, d}, kNoSrc, B, B,
               EpiLinear{dec->rin.bias, q.y, l, 0, nullptr})) return 1;
  if (make_lin(J.l1, dec->l1, Src{q.y, l}, Src{q.h1_in, l}, kNoSrc, B, B,
               EpiLstm{dec->l1.bias, q.c1_in, q.c1_out, q.h1_in, q.h1_out, q.y, q.x1, q.zone_u1, l})) return 1;
  if (make_lin(J.l2, dec->l2, Src{q.x1, l}, Src{q.h2_in, l}, kNoSrc, B, B,
               EpiLstm{dec->l2.bias, q.c2_in, q.c2_out, q.h2_in, q.h2_out, q.x1, q.x2, q.zone_u2, l})) return 1;
  return 0;
}

// one step as six dependent launches (step API, free-running decode)
int launch_step_sequential(const StepJobs& J, hipStream_t st) {
  if (launch_tick(st, J.gru)) return 1;
  if (launch_tick(st, J.pq)) return 1;
  if (launch_tick(st, J.attn)) return 1;
  if (launch_tick(st, J.rin)) return 1;
  if (launch_tick(st, J.l1)) return 1;
  return launch_tick(st, J.l2);
}

// prenet over `rows` activation rows; the hoisted (all steps at once) case uses 32 x 64 output tiles: measured about
// twice as fast as 16 x 64 on such many-row GEMMs (same summation order, so the results are bit-identical)
template <int TM, int NB, int KU>
int launch_prenet(wrnn_decoder* dec, Act x, int rows, int B, const float* u1, const float* u2, Act p1, Act p,
                  hipStream_t st) {
  LinJob<TM, NB, EpiLinear, KU> a, b;
  if (make_lin(a, dec->fc1, Src{x, dec->P}, kNoSrc, kNoSrc, rows, B, EpiLinear{dec->fc1.bias, p1, kFc1, 1, u1})) return 1;
  if (make_lin(b, dec->fc2, Src{p1, kFc1}, kNoSrc, kNoSrc, rows, B, EpiLinear{dec->fc2.bias, p, kFc2, 1, u2})) return 1;
  if (launch_tick(st, a)) return 1;
  return launch_tick(st, b);
}

int check_common(const wrnn_decoder* dec, int B, int T, int r) {
  WRNN_REQUIRE(dec != nullptr, "null decoder handle");
  WRNN_REQUIRE(B >= 1 && B <= 65535, "batch size B=%d out of range [1,65535]", B);
  WRNN_REQUIRE(T >= 1 && T <= kAT * kMaxChunks, "encoder length T=%d out of range [1,%d]", T, kAT * kMaxChunks);
  WRNN_REQUIRE(r >= 1 && r <= kMaxR, "reduction factor r=%d out of range [1,%d]", r, kMaxR);
  return 0;
}

}  // namespace

// ============================================================================ C ABI
extern "C" const char* wrnn_last_error(void) { return wrnn::last_error().c_str(); }
extern "C" int wrnn_version(void) { return 200; }

extern "C" int wrnn_decoder_create(const wrnn_decoder_dims* dims, const wrnn_decoder_weights* hw, int device,
                                   wrnn_decoder** out) {
  WRNN_REQUIRE(dims && hw && out, "wrnn_decoder_create: null argument");
  *out = nullptr;
  const int d = dims->decoder_dims, l = dims->lstm_dims, nm = dims->n_mels, A = dims->aligner_out_dims;
  WRNN_REQUIRE(d == kD, "decoder_dims must be %d (reference GRUCell input = decoder_dims + decoder_dims/2 = "
               "context + 128 prenet outputs), got %d", kD, d);
  WRNN_REQUIRE(l >= 4 && l % 4 == 0, "lstm_dims=%d must be a positive multiple of 4", l);
  WRNN_REQUIRE(nm >= 1 && A >= 0 && (nm + A) % 4 == 0, "n_mels + aligner_out_dims = %d must be a multiple of 4", nm + A);
  const float* const* wp = reinterpret_cast<const float* const*>(hw);
  for (size_t i = 0; i < sizeof(*hw) / sizeof(float*); ++i)
    WRNN_REQUIRE(wp[i] != nullptr, "wrnn_decoder_create: weight pointer #%zu is null", i);
  int ndev = 0;
  WRNN_HIP(hipGetDeviceCount(&ndev));
  WRNN_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  WRNN_HIP(hipGetDeviceProperties(&prop, device));
  WRNN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libwrnn_mi355 is built for gfx950 only, device %d is %s",
               device, prop.gcnArchName);
  Guard g(device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", device);

  auto* dec = new wrnn_decoder();
  dec->device = device;
  dec->n_mels = nm, dec->d = d, dec->l = l, dec->A = A, dec->P = nm + A;
  const int P = dec->P;
  int rc = 0;
  auto lin_rows = [](const float* W, int K) {
    return RowFn([=](int t, int i, int k) { return W[(size_t)(t * 16 + i) * K + k]; });
  };
  auto lin_bias = [](const float* b) { return std::function<float(int, int)>([=](int t, int i) { return b[t * 16 + i]; }); };
  {  // prenet
    auto b1 = lin_bias(hw->prenet_fc1_b), b2 = lin_bias(hw->prenet_fc2_b);
    rc |= build_layer(dec->fc1, kFc1 / 16, P, kFc1, lin_rows(hw->prenet_fc1_w, P), &b1);
    rc |= build_layer(dec->fc2, kFc2 / 16, kFc1, kFc2, lin_rows(hw->prenet_fc2_w, kFc1), &b2);
  }
  {  // attention GRU: K = [context d | prenet 128 | h d], tile rows = 4 units x (r, z, gi_n, gh_n)
    const int Kx = d + kFc2, K = Kx + d;
    const float *Wih = hw->gru_w_ih, *Whh = hw->gru_w_hh, *bih = hw->gru_b_ih, *bhh = hw->gru_b_hh;
    RowFn w = [=](int t, int i, int k) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      const bool xpart = k < Kx;
      switch (gsel) {
        case 0: return xpart ? Wih[(size_t)U * Kx + k] : Whh[(size_t)U * d + (k - Kx)];
        case 1: return xpart ? Wih[(size_t)(d + U) * Kx + k] : Whh[(size_t)(d + U) * d + (k - Kx)];
        case 2: return xpart ? Wih[(size_t)(2 * d + U) * Kx + k] : 0.0f;
        default: return xpart ? 0.0f : Whh[(size_t)(2 * d + U) * d + (k - Kx)];
      }
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      switch (gsel) {
        case 0: return bih[U] + bhh[U];
        case 1: return bih[d + U] + bhh[d + U];
        case 2: return bih[2 * d + U];
        default: return bhh[2 * d + U];
      }
    };
    rc |= build_layer(dec->gru, d / 4, K, 4 * d, w, &b);
    // the same cell with the prenet output LAST in K, [ctx | attn_h | prenet]: the free-running loop's GRU takes it as a hand source
    RowFn wh = [=](int t, int i, int k) -> float { return w(t, i, k < d ? k : k < 2 * d ? Kx + (k - d) : d + (k - 2 * d)); };
    rc |= build_layer(dec->gru_h, d / 4, K, 4 * d, wh, nullptr);
  }
  {  // LSA query projection and rnn_input
    auto bq = lin_bias(hw->lsa_W_b);
    rc |= build_layer(dec->pq, d / 16, d, d, lin_rows(hw->lsa_W_w, d), &bq);
    RowFn wi = [=](int t, int i, int k) -> float {
      const int n = t * 16 + i;
      return n < l ? hw->rnn_input_w[(size_t)n * 2 * d + k] : 0.0f;
    };
    std::function<float(int, int)> bi2 = [=](int t, int i) -> float { return t * 16 + i < l ? hw->rnn_input_b[t * 16 + i] : 0.0f; };
    rc |= build_layer(dec->rin, (l + 15) / 16, 2 * d, l, wi, &bi2);
  }
  auto build_lstm = [&](Layer& L, const float* Wih, const float* Whh, const float* bih, const float* bhh) {
    RowFn w = [=](int t, int i, int k) -> float {
      const int row = (i & 3) * l + t * 4 + (i >> 2);
      return k < l ? Wih[(size_t)row * l + k] : Whh[(size_t)row * l + (k - l)];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int row = (i & 3) * l + t * 4 + (i >> 2);
      return bih[row] + bhh[row];
    };
    return build_layer(L, l / 4, 2 * l, 4 * l, w, &b);
  };
  rc |= build_lstm(dec->l1, hw->lstm1_w_ih, hw->lstm1_w_hh, hw->lstm1_b_ih, hw->lstm1_b_hh);
  rc |= build_lstm(dec->l2, hw->lstm2_w_ih, hw->lstm2_w_hh, hw->lstm2_b_ih, hw->lstm2_b_hh);
  dec->mel_w_host.assign(hw->mel_proj_w, hw->mel_proj_w + (size_t)nm * kMaxR * l);
  {
    dec->rin_w.assign(hw->rnn_input_w, hw->rnn_input_w + (size_t)l * 2 * d);
    dec->rin_b.assign(hw->rnn_input_b, hw->rnn_input_b + l);
    dec->l1_wih.assign(hw->lstm1_w_ih, hw->lstm1_w_ih + (size_t)4 * l * l);
    dec->l1_whh.assign(hw->lstm1_w_hh, hw->lstm1_w_hh + (size_t)4 * l * l);
    dec->l1_bih.assign(hw->lstm1_b_ih, hw->lstm1_b_ih + 4 * l);
    dec->l1_bhh.assign(hw->lstm1_b_hh, hw->lstm1_b_hh + 4 * l);
    dec->l2_wih.assign(hw->lstm2_w_ih, hw->lstm2_w_ih + (size_t)4 * l * l);
    dec->l2_whh.assign(hw->lstm2_w_hh, hw->lstm2_w_hh + (size_t)4 * l * l);
    dec->l2_bih.assign(hw->lstm2_b_ih, hw->lstm2_b_ih + 4 * l);
    dec->l2_bhh.assign(hw->lstm2_b_hh, hw->lstm2_b_hh + 4 * l);
  }
  dec->gru_wih_h.assign(hw->gru_w_ih, hw->gru_w_ih + (size_t)3 * d * (d + kFc2));
  dec->gru_whh_h.assign(hw->gru_w_hh, hw->gru_w_hh + (size_t)3 * d * d);
  dec->gru_bih_h.assign(hw->gru_b_ih, hw->gru_b_ih + 3 * d);
  dec->gru_bhh_h.assign(hw->gru_b_hh, hw->gru_b_hh + 3 * d);
  dec->pq_w_h.assign(hw->lsa_W_w, hw->lsa_W_w + (size_t)d * d);
  dec->pq_b_h.assign(hw->lsa_W_b, hw->lsa_W_b + d);
  dec->fc2_w_h.assign(hw->prenet_fc2_w, hw->prenet_fc2_w + (size_t)kFc2 * kFc1);
  dec->fc2_b_h.assign(hw->prenet_fc2_b, hw->prenet_fc2_b + kFc2);
  dec->n_cu = prop.multiProcessorCount;
  dec->fc1_w_host.assign(hw->prenet_fc1_w, hw->prenet_fc1_w + (size_t)kFc1 * P);
  dec->fc1_b_host.assign(hw->prenet_fc1_b, hw->prenet_fc1_b + kFc1);
  if (A > 0) {  // fc1's aligner-feature columns, K padded to a multiple of 4
    const float* W1 = hw->prenet_fc1_w;
    RowFn wa = [=](int t, int i, int k) -> float { return k < A ? W1[(size_t)(t * 16 + i) * P + nm + k] : 0.0f; };
    auto b1 = lin_bias(hw->prenet_fc1_b);
    rc |= build_layer(dec->fc1a, kFc1 / 16, (A + 3) / 4 * 4, kFc1, wa, &b1);
  }
  {  // LSA small weights; conv transposed to [c][k][filter] so lanes read consecutive filters
    std::vector<float> ct((size_t)2 * kTaps * kFilters);
    for (int fl = 0; fl < kFilters; ++fl)
      for (int c = 0; c < 2; ++c)
        for (int k = 0; k < kTaps; ++k) ct[(c * kTaps + k) * kFilters + fl] = hw->lsa_conv_w[(fl * 2 + c) * kTaps + k];
    rc |= upload(ct.data(), ct.size() * sizeof(float), (void**)&dec->conv_t);
    rc |= upload(hw->lsa_L_w, (size_t)d * kFilters * sizeof(float), (void**)&dec->L_w);
    rc |= upload(hw->lsa_L_b, (size_t)d * sizeof(float), (void**)&dec->L_b);
    rc |= upload(hw->lsa_v_w, (size_t)d * sizeof(float), (void**)&dec->v_w);
  }
  if (rc) {
    std::string keep = wrnn::last_error();
    wrnn_decoder_destroy(dec);
    wrnn::last_error() = keep;
    return 1;
  }
  *out = dec;
  return 0;
}

extern "C" int wrnn_decoder_destroy(wrnn_decoder* dec) {
  if (!dec) return 0;
  Guard g(dec->device);
  (void)hipDeviceSynchronize();
  free_layer(dec->fc1), free_layer(dec->fc2), free_layer(dec->gru), free_layer(dec->pq);
  free_layer(dec->rin), free_layer(dec->l1), free_layer(dec->l2);
  for (auto& m : dec->mel) free_layer(m);
  for (auto& m : dec->fc1m) free_layer(m);
  free_layer(dec->fc1a);
  free_layer(dec->fr_l1f), free_layer(dec->fr_l2x), free_layer(dec->fr_l1h), free_layer(dec->fr_rinh), free_layer(dec->gru_h);
  for (float* p : {dec->conv_t, dec->L_w, dec->L_b, dec->v_w, dec->ws})
    if (p) (void)hipFree(p);
  if (dec->flags) (void)hipFree(dec->flags);
  if (dec->row_counter) (void)hipFree(dec->row_counter);
  if (dec->qhand) (void)hipFree(dec->qhand);
  for (float* p : {dec->chain_pq_w, dec->chain_pq_b})
    if (p) (void)hipFree(p);
  if (dec->chain_gran) (void)hipFree(dec->chain_gran);
  for (hipEvent_t e : dec->chain_ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : dec->chain_prof)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : dec->chain_chunk_ev) (void)hipEventDestroy(e);
  if (dec->stack_buf) (void)hipFree(dec->stack_buf);
  if (dec->stack_flags) (void)hipFree(dec->stack_flags);
  for (hipEvent_t e : dec->stack_ev)
    if (e) (void)hipEventDestroy(e);
  {
    auto& R = dec->res;
    for (float* p : {R.fc1_b, R.fc2_w, R.fc2_b, R.gru_wih, R.gru_whh, R.gru_bih, R.gru_bhh, R.pq_w, R.pq_b, R.l1_wf, R.l1_whh, R.l1_b,
                     R.rin_w, R.rin_b, R.l2_wih, R.l2_whh, R.l2_b})
      if (p) (void)hipFree(p);
    for (float* p : R.fc1m) if (p) (void)hipFree(p);
    for (float* p : R.mel_w) if (p) (void)hipFree(p);
    if (R.gran) (void)hipFree(R.gran);
  }
  for (hipEvent_t e : dec->prof_events) (void)hipEventDestroy(e);
  delete dec;
  return 0;
}

extern "C" int wrnn_decoder_step(wrnn_decoder* dec, const float* enc_q, const float* enc_p, const float* prenet_in,
                                 const wrnn_decoder_state* in, const wrnn_decoder_state* out, int B, int T, int t,
                                 int r, const float* drop_u1, const float* drop_u2, const float* zone_u1,
                                 const float* zone_u2, float* mels_out, float* scores_out, void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(enc_q && enc_p && prenet_in && in && out && mels_out && scores_out, "wrnn_decoder_step: null pointer");
  WRNN_REQUIRE(t >= 0, "wrnn_decoder_step: t=%d must be >= 0", t);
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  WRNN_REQUIRE((zone_u1 == nullptr) == (zone_u2 == nullptr), "give both zoneout uniform tensors or neither");
  WRNN_REQUIRE(in->attn_h != out->attn_h && in->h1 != out->h1 && in->h2 != out->h2 && in->context != out->context,
               "wrnn_decoder_step: hidden-state and context outputs must not alias the inputs");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l;
  const size_t need = (size_t)B * (kFc1 + kFc2 + d + 3 * l) + attn_scratch_floats(B, T) + 32;
  if (ensure_ws(dec, need)) return 1;
  if (ensure_mel_layer(dec, r)) return 1;
  if (ensure_counters(dec, B, st, t == 0)) return 1;  // the step API re-arms at the start of a sequence
  float* w = dec->ws;
  float* p1 = w; w += (size_t)B * kFc1;
  float* p = w; w += (size_t)B * kFc2;
  float* pq = w; w += (size_t)B * d;
  float* y = w; w += (size_t)B * l;
  float* x1 = w; w += (size_t)B * l;
  float* x2 = w; w += (size_t)B * l;
  w = dec->ws + (((size_t)(w - dec->ws) + 31) & ~(size_t)31);  // chunk records start on a 128-byte line
  float* ascr = w;
  if (launch_prenet<1, 1, 4>(dec, act_rm(prenet_in, dec->P), B, B, drop_u1, drop_u2, act_rm(p1, kFc1), act_rm(p, kFc2), st)) return 1;
  // the context is both an input of the GRU and an output of the attention: carry it over first
  if (t != 0 && (in->lsa_cum != out->lsa_cum || in->lsa_att != out->lsa_att)) {
    // LSA state handed over in separate buffers: carry it into the output buffers first
    WRNN_HIP(hipMemcpyAsync(out->lsa_cum, in->lsa_cum, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st));
    WRNN_HIP(hipMemcpyAsync(out->lsa_att, in->lsa_att, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  StepBufs q{enc_q, enc_p, act_rm(p, kFc2), act_rm(in->attn_h, d), act_rm(out->attn_h, d), act_rm(in->context, d), act_rm(out->context, d),
             act_rm(in->h1, l), act_rm(out->h1, l), act_rm(in->h2, l), act_rm(out->h2, l),
             act_rm(in->c1, l), act_rm(in->c2, l), act_rm(out->c1, l), act_rm(out->c2, l), out->lsa_cum, out->lsa_att, zone_u1, zone_u2,
             scores_out, (long)T, act_rm(pq, d), act_rm(y, l), act_rm(x1, l), act_rm(x2, l), ascr};
  StepJobs J;
  if (make_step_jobs(dec, q, B, T, t == 0, J)) return 1;
  if (launch_step_sequential(J, st)) return 1;
  LinJob<1, 1, EpiMel> jm;
  if (make_lin(jm, dec->mel[r], Src{act_rm(x2, l), l}, kNoSrc, kNoSrc, B, B,
               EpiMel{mels_out, (long)dec->n_mels * r, (long)r, r, dec->n_mels * r})) return 1;
  return launch_tick(st, jm);
}

namespace {

// workspace carving for the loops
struct Carver {
  float* w;
  float* take(size_t n) {
    float* p = w;
    w += (n + 3) & ~(size_t)3;  // keep every buffer 16-B aligned
    return p;
  }
  float* take_lines(size_t n) {  // starts on a 128-byte line (the workspace base is 256-byte aligned)
    w = base + (((size_t)(w - base) + 31) & ~(size_t)31);
    return take(n);
  }
  float* base = w;
};

struct LoopBufs {
  float *ah[2], *h1[2], *h2[2], *c1, *c2, *ctx, *ctx2[2], *pq, *cum, *att, *ascr, *y, *x1;
};

size_t loop_floats(int B, int T, int d, int l) {
  return (size_t)B * (2 * d + 4 * l + 2 * l + 2 * d + d + 2 * T + 2 * l) + attn_scratch_floats(B, T) + 64 + 32;
}

void carve_loop(Carver& c, LoopBufs& b, int B, int T, int d, int l) {
  for (int i = 0; i < 2; ++i) b.ah[i] = c.take((size_t)B * d);
  for (int i = 0; i < 2; ++i) b.h1[i] = c.take((size_t)B * l);
  for (int i = 0; i < 2; ++i) b.h2[i] = c.take((size_t)B * l);
  b.c1 = c.take((size_t)B * l), b.c2 = c.take((size_t)B * l);
  b.ctx2[0] = c.take((size_t)B * d), b.ctx2[1] = c.take((size_t)B * d), b.pq = c.take((size_t)B * d);
  b.ctx = b.ctx2[0];  // loops with one context buffer (free-running)
  b.cum = c.take((size_t)B * T), b.att = c.take((size_t)B * T), b.ascr = c.take_lines(attn_scratch_floats(B, T));
  b.y = c.take((size_t)B * l), b.x1 = c.take((size_t)B * l);
}

int zero_loop_state(const LoopBufs& b, int B, int d, int l, hipStream_t st) {
  // zero initial state (tacotron.py:302-316); zeros are layout independent
  WRNN_HIP(hipMemsetAsync(b.ah[0], 0, (size_t)B * d * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.h1[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.h2[0], 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.c1, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.c2, 0, (size_t)B * l * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.ctx2[0], 0, (size_t)B * d * sizeof(float), st));
  WRNN_HIP(hipMemsetAsync(b.ctx2[1], 0, (size_t)B * d * sizeof(float), st));
  return 0;
}

// jobs of step k of a loop; the step's LSTM output x2 goes to `x2` (interleaved, one step)
int loop_step_jobs(wrnn_decoder* dec, const LoopBufs& b, const float* enc_q, const float* enc_p, const float* pp,
                   float* x2, float* scores, long score_stride, int k, int B, int T, const float* zone_u1,
                   const float* zone_u2, StepJobs& J) {
  const int d = dec->d, l = dec->l;
  const int a = k & 1, o = a ^ 1;
  StepBufs q{enc_q, enc_p, act_il(pp, B, kFc2), act_il(b.ah[a], B, d), act_il(b.ah[o], B, d), act_il(b.ctx2[o], B, d), act_il(b.ctx2[a], B, d),
             act_il(b.h1[a], B, l), act_il(b.h1[o], B, l), act_il(b.h2[a], B, l), act_il(b.h2[o], B, l),
             act_il(b.c1, B, l), act_il(b.c2, B, l), act_il(b.c1, B, l), act_il(b.c2, B, l), b.cum, b.att, zone_u1, zone_u2,
             scores, score_stride,
             act_il(b.pq, B, d), act_il(b.y, B, l), act_il(b.x1, B, l), act_il(x2, B, l), b.ascr};
  return make_step_jobs(dec, q, B, T, k == 0, J);
}

}  // namespace

namespace {
// records an event in front of the next launch of the given kind (no-op unless profiling)
int prof_mark(wrnn_decoder* dec, int kind, hipStream_t st) {
  if (!dec->profiling) return 0;
  const size_t i = dec->prof_kind.size();
  if (i >= dec->prof_events.size()) {
    hipEvent_t e;
    WRNN_HIP(hipEventCreate(&e));
    dec->prof_events.push_back(e);
  }
  WRNN_HIP(hipEventRecord(dec->prof_events[i], st));
  dec->prof_kind.push_back(kind);
  return 0;
}
int prof_finish(wrnn_decoder* dec, hipStream_t st) {
  if (!dec->profiling) return 0;
  if (prof_mark(dec, -1, st)) return 1;
  WRNN_HIP(hipStreamSynchronize(st));
  for (int k = 0; k < WRNN_PROFILE_KINDS; ++k) dec->prof_ms[k] = 0.f, dec->prof_count[k] = 0;
  for (size_t i = 0; i + 1 < dec->prof_kind.size(); ++i) {
    float ms = 0.f;
    WRNN_HIP(hipEventElapsedTime(&ms, dec->prof_events[i], dec->prof_events[i + 1]));
    dec->prof_ms[dec->prof_kind[i]] += ms;
    dec->prof_count[dec->prof_kind[i]] += 1;
  }
  dec->prof_kind.clear();
  return 0;
}
}  // namespace

extern "C" int wrnn_decoder_set_profiling(wrnn_decoder* dec, int enable) {
  WRNN_REQUIRE(dec != nullptr, "null decoder handle");
  dec->profiling = enable != 0;
  dec->prof_kind.clear();
  return 0;
}

extern "C" int wrnn_decoder_get_profile(wrnn_decoder* dec, float* ms_out, int* count_out) {
  WRNN_REQUIRE(dec && ms_out && count_out, "wrnn_decoder_get_profile: null argument");
  for (int k = 0; k < WRNN_PROFILE_KINDS; ++k) ms_out[k] = dec->prof_ms[k], count_out[k] = dec->prof_count[k];
  return 0;
}

namespace {

// ---- teacher-forced loop, attention recurrence as one persistent launch (kernel: taco_chain.h) -------------------------
// pair arrays of one call, in u64 words from the start of chain_gran (all zeroed before the launch), then the error word
struct ChainLayout {
  static constexpr size_t H = 0, nH = (size_t)2 * 4 * 128 * 16 * 2;
  static constexpr size_t U = H + nH, nU = (size_t)32 * 8 * 128 * 2;   // [B][NS][16 NS] pairs: 64 x 4 x 64 or 32 x 8 x 128
  static constexpr size_t C = U + nU, nC = (size_t)kCMaxB * 4 * 32 * 2;
  static constexpr size_t HT = C + nC, nHT = (size_t)2 * kCMaxB * 128 * 2;
  static constexpr size_t err = HT + nHT, xcc = err + 2, total = xcc + 128;   // (xcc: 256 x u32, the workgroups' XCD numbers)
  static constexpr size_t state_floats = (size_t)256 * kCState;   // attention state of every workgroup between launches
};

// slices per row: four 64-dim slices cover 128 encoder positions, eight 32-dim slices 256; a row's slices are one
// workgroup each and all workgroups of a launch must be resident, one per compute unit
inline int chain_slices(int T) { return T <= 128 ? 4 : 8; }
bool chain_supported(const wrnn_decoder* dec, int B, int T) {
  return dec->d == kD && B <= kCMaxB && T <= kCMaxT && B * chain_slices(T) <= 256 && dec->n_cu >= 256;
}

__global__ void chain_poison_kernel(const unsigned* err, float* out, size_t n) {   // a hand-over timed out: no silent garbage
  if (*err == 0) return;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = NAN;
}

int ensure_chain(wrnn_decoder* dec) {
  if (!dec->chain_pq_w && upload(dec->pq_w_h.data(), dec->pq_w_h.size() * sizeof(float), (void**)&dec->chain_pq_w)) return 1;
  if (!dec->chain_pq_b && upload(dec->pq_b_h.data(), dec->pq_b_h.size() * sizeof(float), (void**)&dec->chain_pq_b)) return 1;
  if (!dec->chain_gran) WRNN_HIP(hipMalloc((void**)&dec->chain_gran, ChainLayout::total * sizeof(u64) + ChainLayout::state_floats * sizeof(float)));
  if (!dec->chain_ev[0]) {
    for (auto& e : dec->chain_ev) WRNN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : dec->chain_prof) WRNN_HIP(hipEventCreate(&e));
  }
  return 0;
}

// ---- the LSTM stack of the chain path as persistent weights-resident launches (kernel: taco_stack.h) ---------------------
constexpr size_t kStackBlocks = kSRing + 2 * 2 + 2;   // x1 ring, two h rings of two slots, two cell states: [128][B][4] blocks
bool stack_supported(const wrnn_decoder* dec, int B) { return dec->l == kSL && B <= 64 && dec->n_cu >= 256; }

int ensure_stack(wrnn_decoder* dec) {
  if (!dec->stack_buf) WRNN_HIP(hipMalloc((void**)&dec->stack_buf, kStackBlocks * kSStateFloats * sizeof(float)));
  if (!dec->stack_flags) WRNN_HIP(hipMalloc((void**)&dec->stack_flags, 260 * sizeof(unsigned)));
  for (auto& e : dec->stack_ev)
    if (!e) WRNN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// steps [k0, k1) (at most kSRing) of both cells on stream ss; y_all / x2 / zone_u are the whole call's arrays
int launch_stack(wrnn_decoder* dec, const float* y_all, float* x2, const float* zone_u1, const float* zone_u2, int B, int k0, int k1,
                 unsigned long long* stamps, hipStream_t ss) {
  const size_t blk = (size_t)kSL * B;
  StackParams q{};
  q.wt[0] = dec->l1.wt, q.wt[1] = dec->l2.wt, q.bias[0] = dec->l1.bias, q.bias[1] = dec->l2.bias;
  q.y = y_all + (size_t)k0 * blk, q.x2 = x2 + (size_t)k0 * blk;
  float* sb = dec->stack_buf;
  q.x1r = sb, sb += (size_t)kSRing * kSStateFloats;
  for (int c = 0; c < 2; ++c) q.hr[c] = sb, sb += (size_t)2 * kSStateFloats;
  for (int c = 0; c < 2; ++c) q.cst[c] = sb, sb += (size_t)kSStateFloats;
  q.zone_u[0] = zone_u1 ? zone_u1 + (size_t)k0 * blk : nullptr, q.zone_u[1] = zone_u2 ? zone_u2 + (size_t)k0 * blk : nullptr;
  q.flags = dec->stack_flags, q.B = B, q.k0 = k0, q.k1 = k1, q.stamps = stamps;
  static const int ng_env = getenv("WRNN_STACK_NG") ? atoi(getenv("WRNN_STACK_NG")) : 0;   // experiments: groups in flight at B > 48
  const int MT = (B + 15) / 16;
  // groups of B operands in flight per wave: thin is what costs the chain launches beside this kernel least (B = 64: 14.2 us
  // per decoder step with one group, 14.4 / 14.7 / 15.0 with two / three / four); WRNN_STACK_NG overrides (experiments)
  const int ng = ng_env > 0 ? ng_env : 1;
  auto* kfn = MT == 1 ? (ng == 1 ? stack_kernel<1, 1> : ng == 2 ? stack_kernel<1, 2> : stack_kernel<1, 8>)
            : MT == 2 ? (ng == 1 ? stack_kernel<2, 1> : ng == 2 ? stack_kernel<2, 2> : stack_kernel<2, 8>)
            : MT == 3 ? (ng == 1 ? stack_kernel<3, 1> : ng == 2 ? stack_kernel<3, 2> : stack_kernel<3, 5>)
                      : (ng == 1 ? stack_kernel<4, 1> : ng == 2 ? stack_kernel<4, 2> : ng == 3 ? stack_kernel<4, 3> : stack_kernel<4, 4>);
  hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 0, ss, q);
  WRNN_LAUNCH_CHECK();
  return 0;
}

// Everything of the loop after the hoisted prenet: the chain launch (GRU -> query -> attention for all n steps), rnn_input
// for all steps as one GEMM, then one launch per step with LSTM1(k) || LSTM2(k-1).  x2 receives the LSTM stack's output of
// every step (interleaved), scores go to att_out.
int run_tf_chain(wrnn_decoder* dec, const float* enc_q, const float* enc_p, const float* pp, float* x2, float* ctx_all,
                 float* ah_all, float* y_all, const LoopBufs& b, int B, int T, int n, const float* zone_u1,
                 const float* zone_u2, float* att_out, float* mel_out, size_t mel_floats, hipStream_t st,
                 const std::function<int(int, int)>& mel_steps, bool* mel_done) {
  const int d = dec->d, l = dec->l;
  *mel_done = false;
  if (ensure_chain(dec)) return 1;
  u64* g = dec->chain_gran;
  WRNN_HIP(hipMemsetAsync(g, 0, ChainLayout::total * sizeof(u64), st));
  ChainParams q{};
  q.enc_p = enc_p, q.enc_q = enc_q, q.pp = pp;
  q.gru_wt = dec->gru.wt, q.gru_bias = dec->gru.bias, q.pq_w = dec->chain_pq_w, q.pq_b = dec->chain_pq_b;
  q.conv_t = dec->conv_t, q.L_w = dec->L_w, q.L_b = dec->L_b, q.v_w = dec->v_w;
  q.scores = att_out, q.score_stride = (long)n * T;
  q.ctx_all = ctx_all, q.ah_all = ah_all;
  q.H = g + ChainLayout::H, q.U = g + ChainLayout::U, q.C = g + ChainLayout::C, q.HT = g + ChainLayout::HT;
  q.err = reinterpret_cast<u32*>(g + ChainLayout::err);
  q.state = reinterpret_cast<float*>(g + ChainLayout::total);
  q.B = B, q.T = T, q.n = n;
  q.stamp_k = getenv("WRNN_CHAIN_STAMP_STEP") ? atoi(getenv("WRNN_CHAIN_STAMP_STEP")) : 100;
  q.u_plain = getenv("WRNN_CHAIN_U_WT") ? 0 : 1, q.xcc = reinterpret_cast<u32*>(g + ChainLayout::xcc);
  u64* stamps = nullptr;
  const char* stamp_file = getenv("WRNN_CHAIN_STAMPS");   // diagnostics: phase timestamps of workgroup 0 at steps 100..103
  if (stamp_file && n >= q.stamp_k + 4) {
    WRNN_HIP(hipMalloc((void**)&stamps, 64 * sizeof(u64)));
    WRNN_HIP(hipMemsetAsync(stamps, 0, 64 * sizeof(u64), st));
  }
  q.stamps = stamps;
  const int NS = chain_slices(T);
  const size_t lds = chain_lds_floats(NS) * sizeof(float);
  static_assert(chain_lds_floats(4) * sizeof(float) <= ((size_t)144 << 10) && chain_lds_floats(4) * sizeof(float) > ((size_t)80 << 10) &&
                chain_lds_floats(8) * sizeof(float) <= ((size_t)144 << 10) && chain_lds_floats(8) * sizeof(float) > ((size_t)80 << 10),
                "one chain workgroup per compute unit, 16 KB left for an LSTM workgroup beside it");
  auto* kfn = NS == 4 ? chain_kernel<4> : chain_kernel<8>;
  WRNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = std::max(NS * 8 * ((B + 7) / 8), 64 * ((B + 15) / 16));   // (attention role: rows in groups of eight, taco_chain.h)
  // The chain runs on a stream of its own, as launches of 16 .. 64 steps (the attention state of a row crosses the
  // boundary in q.state, the tags count on); each launch records an event, and the caller's stream follows chunk by chunk:
  // wait for the chunk's event, rnn_input of the chunk as one GEMM, then one launch per step with LSTM1(k) || LSTM2(k-1).
  // (Events, not an in-kernel wait on the chain's progress: a kernel that spins on the caller's stream can share a hardware
  //  queue with the chain's stream and then waits for a launch queued behind itself.)
  // WRNN_TF_CHAIN_SERIAL=1 (diagnostic): everything on the caller's stream, the stack after the whole chain.
  const bool serial = getenv("WRNN_TF_CHAIN_SERIAL") != nullptr;
  // ONE chain stream per device, shared by every handle: chain launches cannot overlap anyway (each needs all its
  // workgroups resident, one per compute unit: two half-resident grids would wait for each other until their spin
  // limits), so stream order is exactly the order they need, and a process with many handles does not oversubscribe the
  // hardware queues (with a prioritised stream per handle, a test process holding a dozen handles stalled for seconds).
  // (wrnn::persistent_stream: the other persistent grids of the library queue there too)
  std::lock_guard<std::mutex> chain_lock(wrnn::persistent_mutex());
  hipStream_t shared = nullptr;
  if (wrnn::persistent_stream(dec->device, &shared)) return 1;
  hipStream_t cs = serial ? st : shared;
  if (!serial) {
    WRNN_HIP(hipEventRecord(dec->chain_ev[0], st));
    WRNN_HIP(hipStreamWaitEvent(cs, dec->chain_ev[0], 0));
  }
  // chunk sizes: 16, 32, then 64s, and 32, 16, 8 at the end -- the stack can only start when the first chain launch has
  // ended and still has the last chunk to do when the last one ends, so the ends are short; the middle is long (a chain
  // launch costs a set-up of its own).  WRNN_TF_CHAIN_CHUNK=<c>: constant chunks of c steps (experiments).
  const char* ch_env = getenv("WRNN_TF_CHAIN_CHUNK");
  static const bool tail_env = getenv("WRNN_TF_CHAIN_OLD_TAIL") != nullptr;   // experiments: round 2's chunk schedule
  std::vector<int> bounds{0};   // chunk boundaries
  {
    std::vector<int> sizes;
    if (serial) sizes = {n};
    else if (ch_env) sizes.assign((n + std::max(1, atoi(ch_env)) - 1) / std::max(1, atoi(ch_env)), std::max(1, atoi(ch_env)));
    else if (n <= 104) sizes.assign((n + 15) / 16, 16);
    else if (n <= 176 || tail_env) {
      sizes = {16, 32};
      for (int left = n - 104; left > 0; left -= 64) sizes.push_back(std::min(64, left));
      sizes.insert(sizes.end(), {32, 16, 8});
    } else {
      // the end shrinks gently (each chunk >= 2/3 of the one before): a stack launch of c steps (9 us per step) has to be done
      // when the chain's next launch ends (13 us per step), or the stack falls behind and the pass ends that much later
      const int body = n - 176, rem = body % 64;
      sizes = {16, 32};
      if (rem >= 16) sizes.push_back(rem);
      else sizes.back() += rem;
      sizes.insert(sizes.end(), body / 64, 64);
      sizes.insert(sizes.end(), {48, 32, 24, 16, 8});
    }
    for (int sz : sizes) bounds.push_back(std::min(n, bounds.back() + sz));
  }
  const int nchunks = (int)bounds.size() - 1;
  // The LSTM stack: persistent weights-resident launches on the device's second stream for such grids (one launch per chunk
  // of at most kSRing steps, queued behind the chunk's rnn_input GEMM), or, for other lstm_dims and under
  // WRNN_TF_PATH_CHAIN_LAUNCH_STACK, one launch per step on the caller's stream (round 2's schedule).
  const bool stack_res = dec->tf_path != WRNN_TF_PATH_CHAIN_LAUNCH_STACK && stack_supported(dec, B);
  hipStream_t ss = st;
  if (stack_res) {
    if (ensure_stack(dec)) return 1;
    WRNN_HIP(hipMemsetAsync(dec->stack_flags, 0, 260 * sizeof(unsigned), st));
    if (!serial && wrnn::persistent_stream(dec->device, &ss, 1)) return 1;
  }
  u64* sstamps = nullptr;
  const char* sstamp_file = getenv("WRNN_STACK_STAMPS");   // diagnostics: phase timestamps of tile 0 of both cells at steps 40..43
  if (stack_res && sstamp_file && n >= 48) {
    WRNN_HIP(hipMalloc((void**)&sstamps, 64 * sizeof(u64)));
    WRNN_HIP(hipMemsetAsync(sstamps, 0, 64 * sizeof(u64), st));
  }
  while ((int)dec->chain_chunk_ev.size() < nchunks) {
    hipEvent_t e;
    WRNN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    dec->chain_chunk_ev.push_back(e);
  }
  if (dec->profiling) WRNN_HIP(hipEventRecord(dec->chain_prof[0], cs));
  for (int c = 0; c < nchunks; ++c) {
    q.k0 = bounds[c], q.k1 = bounds[c + 1];
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(kCT), lds, cs, q);
    WRNN_LAUNCH_CHECK();
    if (!serial) WRNN_HIP(hipEventRecord(dec->chain_chunk_ev[c], cs));
  }
  if (dec->profiling) WRNN_HIP(hipEventRecord(dec->chain_prof[1], cs));
  if (getenv("WRNN_CHAIN_DEBUG")) {
    WRNN_HIP(hipDeviceSynchronize());
    unsigned e = 0;
    WRNN_HIP(hipMemcpy(&e, q.err, 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "chain: blocks %d nchunks %d n %d err %u serial %d\n", blocks, nchunks, n, e, (int)serial);
    std::vector<u64> hC((size_t)4 * B * 32 * 2);
    WRNN_HIP(hipMemcpy(hC.data(), q.C, hC.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4 * B && i < 8; ++i) fprintf(stderr, "  C slot %d tag %u\n", i, (unsigned)(hC[(size_t)i * 64] >> 32));
  }
  float* x1[2] = {b.y, b.x1};
  int stack_launches = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int k0 = bounds[c], k1 = bounds[c + 1];
    // (profiling: the caller's stream's wait for the chunk's chain launch is booked under CHAIN, which the chain launches' own
    //  duration replaces afterwards; the other kinds then hold their launches only)
    if (prof_mark(dec, WRNN_PROFILE_CHAIN, st)) return 1;
    if (!serial) WRNN_HIP(hipStreamWaitEvent(st, dec->chain_chunk_ev[c], 0));
    if (prof_mark(dec, WRNN_PROFILE_RIN_CHUNKS, st)) return 1;
    {  // rnn_input over the chunk's steps
      // (16-row activation tiles: the 32 x 64 tile job of the hoisted prenet needs 200 registers and would not fit beside a
      //  chain workgroup, 349: it would wait for the chain launch to end)
      LinJob<2, 1, EpiLinear> jr;
      if (make_lin(jr, dec->rin, Src{act_il(ctx_all + (size_t)k0 * B * d, B, d), d}, Src{act_il(ah_all + (size_t)k0 * B * d, B, d), d}, kNoSrc,
                   (k1 - k0) * B, B, EpiLinear{dec->rin.bias, act_il(y_all + (size_t)k0 * B * l, B, l), l, 0, nullptr})) return 1;
      if (launch_tick(st, jr)) return 1;
    }
    if (prof_mark(dec, WRNN_PROFILE_LSTM_STACK, st)) return 1;
    if (stack_res) {
      for (int s0 = k0; s0 < k1; s0 += kSRing) {
        const int s1 = std::min(k1, s0 + kSRing);
        if (ss != st) {
          WRNN_HIP(hipEventRecord(dec->stack_ev[0], st));
          WRNN_HIP(hipStreamWaitEvent(ss, dec->stack_ev[0], 0));
        }
        if (launch_stack(dec, y_all, x2, zone_u1, zone_u2, B, s0, s1, sstamps, ss)) return 1;
        ++stack_launches;
        // The caller's stream does not wait for the stack here: the next chunk's rnn_input GEMM needs the chain only, and with
        // the short chunks at the end of a pass a stack launch that is still running would hold it back (the pass ended 450 us
        // after the chain, now 250).  It waits once, after the last chunk -- except under profiling, where the stack's launches
        // are bracketed on the caller's stream.
        const bool last = c + 1 == nchunks && s1 == k1;
        if (ss != st && last && nchunks >= 2) {
          // the mel projection of every step but the last chunk's goes out now, beside the last stack launch (the chain has
          // ended: there is room for its 200-register workgroups); only the last chunk's frames are left for afterwards
          WRNN_HIP(hipStreamWaitEvent(st, dec->stack_ev[2], 0));
          if (prof_mark(dec, WRNN_PROFILE_MEL, st)) return 1;
          if (mel_steps(0, k0)) return 1;
          if (prof_mark(dec, WRNN_PROFILE_LSTM_STACK, st)) return 1;
        }
        if (ss != st && (dec->profiling || last)) {
          WRNN_HIP(hipEventRecord(dec->stack_ev[1], ss));
          WRNN_HIP(hipStreamWaitEvent(st, dec->stack_ev[1], 0));
        }
        if (ss != st && last && nchunks >= 2) {
          if (prof_mark(dec, WRNN_PROFILE_MEL, st)) return 1;
          if (mel_steps(k0, k1)) return 1;
          *mel_done = true;
        }
        if (ss != st && c + 2 == nchunks && s1 == k1) WRNN_HIP(hipEventRecord(dec->stack_ev[2], ss));   // "all but the last chunk done"
      }
      continue;
    }
    // the LSTM stack, LSTM2 one step behind LSTM1 in the same launch
    for (int k = k0; k < k1 + (k1 == n ? 1 : 0); ++k) {
      LinJob<4, 1, EpiLstm, 2> j1{}, j2{};   // four weight tiles per workgroup: one workgroup per compute unit and cell
      if (k < n) {
        const int a = k & 1, o = a ^ 1;
        const Act yk = act_il(y_all + (size_t)k * B * l, B, l);
        if (make_lin(j1, dec->l1, Src{yk, l}, Src{act_il(b.h1[a], B, l), l}, kNoSrc, B, B,
                     EpiLstm{dec->l1.bias, act_il(b.c1, B, l), act_il(b.c1, B, l), act_il(b.h1[a], B, l), act_il(b.h1[o], B, l), yk,
                             act_il(x1[a], B, l), zone_u1 ? zone_u1 + (size_t)k * B * l : nullptr, l})) return 1;
      }
      if (k >= 1) {
        const int kk = k - 1, a = kk & 1, o = a ^ 1;
        if (make_lin(j2, dec->l2, Src{act_il(x1[a], B, l), l}, Src{act_il(b.h2[a], B, l), l}, kNoSrc, B, B,
                     EpiLstm{dec->l2.bias, act_il(b.c2, B, l), act_il(b.c2, B, l), act_il(b.h2[a], B, l), act_il(b.h2[o], B, l),
                             act_il(x1[a], B, l), act_il(x2 + (size_t)kk * B * l, B, l), zone_u2 ? zone_u2 + (size_t)kk * B * l : nullptr, l})) return 1;
      }
      if (launch_tick(st, j1, j2)) return 1;
    }
  }
  (void)mel_out, (void)mel_floats;
  dec->last_stack_launches = stack_res ? stack_launches : n + 1;
  if (sstamps) {
    WRNN_HIP(hipDeviceSynchronize());
    u64 h[64];
    WRNN_HIP(hipMemcpy(h, sstamps, sizeof(h), hipMemcpyDeviceToHost));
    (void)hipFree(sstamps);
    if (FILE* f = fopen(sstamp_file, "w")) {
      // us from the step's start: x half done, own flags seen, h half done, finished + stores issued, flag stored
      for (int cell = 0; cell < 2; ++cell)
        for (int r = 0; r < 4; ++r) {
          const u64* q = h + (cell * 4 + r) * 8;
          fprintf(f, "cell%d step%d:", cell, 40 + r);
          for (int i = 1; i < 6; ++i) fprintf(f, " %6.2f", q[i] ? (double)(long long)(q[i] - q[0]) * 0.01 : 0.0);
          if (r < 3) fprintf(f, "   next step +%.2f", (double)(long long)(q[8] - q[0]) * 0.01);
          fprintf(f, "\n");
        }
      fclose(f);
    }
  }
  if (stamps) {
    WRNN_HIP(hipDeviceSynchronize());
    u64 h[64];
    WRNN_HIP(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    (void)hipFree(stamps);
    if (FILE* f = fopen(stamp_file, "w")) {
      // us from the step's start: GRU inputs swept, state published, location features done, state arrived, query slice,
      // energies' share, shares exchanged, context partials, context published
      for (int r = 0; r < 4; ++r) {
        fprintf(f, "step%d:", q.stamp_k + r);
        for (int i = 1; i < 10; ++i) fprintf(f, " %6.2f", (double)(long long)(h[r * 16 + i] - h[r * 16]) * 0.01);
        if (r < 3) fprintf(f, "   next step +%.2f", (double)(long long)(h[(r + 1) * 16] - h[r * 16]) * 0.01);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  return 0;
}

}  // namespace

extern "C" int wrnn_decoder_run_teacher_forced(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                                               const float* mel_in, int B, int T, int S, int r,
                                               const float* drop_u1, const float* drop_u2, const float* zone_u1,
                                               const float* zone_u2, float* mel_out, float* att_out, void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(S >= 1, "wrnn_decoder_run_teacher_forced: S=%d must be >= 1", S);
  WRNN_REQUIRE(enc_q && enc_p && mel_in && mel_out && att_out, "wrnn_decoder_run_teacher_forced: null pointer");
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  WRNN_REQUIRE((zone_u1 == nullptr) == (zone_u2 == nullptr), "give both zoneout uniform tensors or neither");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l, P = dec->P;
  const int n = (S + r - 1) / r;
  const size_t rows = (size_t)n * B;
  WRNN_REQUIRE(rows <= (size_t)1 << 30, "wrnn_decoder_run_teacher_forced: n*B too large");
  // B <= 64 at T <= 128, B <= 32 at T <= 256: the attention recurrence as one persistent launch (taco_chain.h) with the LSTM stack's launches
  // beside it (16.8 us per step at B=64, T=128 against 21.4 for the two-launch schedule below); other shapes: launches
  const bool use_chain = dec->tf_path != WRNN_TF_PATH_LAUNCHES && chain_supported(dec, B, T);
  WRNN_REQUIRE(use_chain || (dec->tf_path != WRNN_TF_PATH_CHAIN && dec->tf_path != WRNN_TF_PATH_CHAIN_LAUNCH_STACK),
               "wrnn_decoder_run_teacher_forced: the chain path needs B <= %d at T <= 128, B <= 32 at T <= %d, and a 256-CU device", kCMaxB, kCMaxT);
  const size_t need = rows * (P + kFc1 + kFc2 + l) + loop_floats(B, T, d, l) + (use_chain ? rows * (2 * d + l) + 16 : 0);
  if (ensure_ws(dec, need)) return 1;
  if (ensure_mel_layer(dec, r)) return 1;
  if (ensure_counters(dec, B, st)) return 1;
  Carver cv{dec->ws};
  float* xp = cv.take(rows * P);      // all interleaved, step-major
  float* p1 = cv.take(rows * kFc1);
  float* pp = cv.take(rows * kFc2);
  float* x2 = cv.take(rows * l);
  float *ctx_all = nullptr, *ah_all = nullptr, *y_all = nullptr;
  if (use_chain) ctx_all = cv.take(rows * d), ah_all = cv.take(rows * d), y_all = cv.take(rows * l);
  LoopBufs b;
  carve_loop(cv, b, B, T, d, l);
  if (zero_loop_state(b, B, d, l, st)) return 1;
  if (prof_mark(dec, WRNN_PROFILE_PRENET, st)) return 1;
  // teacher forcing: every step's prenet input is known up front -> the prenet is hoisted out of the
  // loop as one batched GEMM pair over all n*B rows
  {
    const size_t total = rows * P;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_prenet_in_kernel, dim3(blocks), dim3(256), 0, st, mel_in, xp, B, P, S, r, n);
    WRNN_LAUNCH_CHECK();
  }
  if (launch_prenet<2, 4, 2>(dec, act_il(xp, B, P), (int)rows, B, drop_u1, drop_u2, act_il(p1, B, kFc1), act_il(pp, B, kFc2), st))
    return 1;
  // Software pipeline over steps.  The attention recurrence (GRU -> query projection -> attention)
  // is the dependent chain: one launch per stage.  The LSTM stack only consumes that chain's
  // outputs, so it rides along as extra blocks of those launches, one and two steps behind:
  //   tick 1: GRU(k)        || LSTM1(k-2)
  //   tick 2: query-proj(k)
  //   tick 3: attention(k)  || rnn_input(k-1) || LSTM2(k-2)
  // (an LSTM cell costs its host about 3.3 us wherever it rides, and both cells in the attention launch outrun the
  // fill path; rnn_input is free between the attention's and the cell's block ranges -- after the cell it costs
  // 0.5 us -- and reads the context of step k-1 while the attention writes that of step k: two context buffers.
  // See DESIGN.md for the measurements.)
  if (use_chain) {
    // mel projection of steps [k0, k1)
    auto mel_steps = [&](int k0, int k1) -> int {
      LinJob<2, 4, EpiMel, 2> jm;
      if (make_lin(jm, dec->mel[r], Src{act_il(x2 + (size_t)k0 * B * l, B, l), l}, kNoSrc, kNoSrc, (k1 - k0) * B, B,
                   EpiMel{mel_out + (size_t)k0 * r, (long)dec->n_mels * n * r, (long)n * r, r, dec->n_mels * r})) return 1;
      return launch_tick(st, jm);
    };
    bool mel_done = false;
    if (run_tf_chain(dec, enc_q, enc_p, pp, x2, ctx_all, ah_all, y_all, b, B, T, n, zone_u1, zone_u2, att_out, mel_out,
                     (size_t)B * dec->n_mels * n * r, st, mel_steps, &mel_done)) return 1;
    if (!mel_done) {   // (per-step LSTM launches, one chunk, or the serial diagnostic: one GEMM over all steps)
      if (prof_mark(dec, WRNN_PROFILE_MEL, st)) return 1;
      if (mel_steps(0, n)) return 1;
    }
    // a hand-over that timed out (chain or stack) leaves NaN in BOTH outputs of the call
    hipLaunchKernelGGL(stack_poison_kernel, dim3(64), dim3(256), 0, st, reinterpret_cast<const unsigned*>(dec->chain_gran + ChainLayout::err),
                       mel_out, (size_t)B * dec->n_mels * n * r, att_out, (size_t)B * n * T);
    WRNN_LAUNCH_CHECK();
    if (dec->stack_flags && dec->tf_path != WRNN_TF_PATH_CHAIN_LAUNCH_STACK && stack_supported(dec, B)) {
      hipLaunchKernelGGL(stack_poison_kernel, dim3(64), dim3(256), 0, st, dec->stack_flags + 256, mel_out, (size_t)B * dec->n_mels * n * r,
                         att_out, (size_t)B * n * T);
      WRNN_LAUNCH_CHECK();
    }
    if (prof_finish(dec, st)) return 1;
    if (dec->profiling) {  // the chain launch ran on its own stream: its duration from the events around it; the stack's
                           // interval (progress waits, rnn_input GEMMs, n + 1 LSTM launches) counted per LSTM launch
      float ms = 0.f;
      WRNN_HIP(hipEventSynchronize(dec->chain_prof[1]));
      WRNN_HIP(hipEventElapsedTime(&ms, dec->chain_prof[0], dec->chain_prof[1]));
      dec->prof_ms[WRNN_PROFILE_CHAIN] = ms, dec->prof_count[WRNN_PROFILE_CHAIN] = 1;
      dec->prof_count[WRNN_PROFILE_LSTM_STACK] = dec->last_stack_launches;
    }
    return 0;
  }
  StepJobs ring[3] = {};
  auto off = [](auto job) { job.groups = 0; return job; };
  auto off_attn = [](AttnJob job) { job.B = 0; return job; };
  // Deferred combine (WRNN_DEFERRED_COMBINE=1, T <= 256; OFF by default because it measured slower): the attention launch
  // only PUBLISHES its chunk records (defer = 1) and the combine moves across the kernel boundary into the next step's GRU
  // launch, as a job of its own (scores, LSA state, context for rnn_input) and inside the GRU job itself (LinJob's combined
  // source):   tick 1: combine(k-1) || GRU(k) || LSTM1(k-2).  Bit-identical to the ticket combine.  Measured at B=64, T=128:
  // the attention launch loses only 0.45 us (with its seam gone the riding LSTM2 bounds it), the GRU launch gains 3 us (every
  // GRU workgroup fetches 4 chunk records per row instead of one context): 25.6 us per step against 23.1.
  const int chunks = attn_chunks(T);
  const char* defer_env = getenv("WRNN_DEFERRED_COMBINE");
  const bool defer = chunks <= 8 && defer_env && defer_env[0] == '1';
  // The query projection rides in the GRU launch behind an in-launch hand-over (QueryJob): 2 launches per step.
  // WRNN_SEPARATE_QUERY_LAUNCH=1 keeps it a launch of its own (3 per step; same bits).
  const bool merge_query = getenv("WRNN_SEPARATE_QUERY_LAUNCH") == nullptr;
  if (merge_query && ensure_qhand(dec, B, st)) return 1;
  for (int k = 0; k <= n + 1; ++k) {
    StepJobs& cur = ring[k % 3];
    const StepJobs& p1j = ring[(k + 2) % 3];  // step k-1
    const StepJobs& p2j = ring[(k + 1) % 3];  // step k-2
    const bool have_cur = k < n, have_p1 = k >= 1 && k <= n, have_p2 = k >= 2;
    if (have_cur &&
        loop_step_jobs(dec, b, enc_q, enc_p, pp + (size_t)k * B * kFc2, x2 + (size_t)k * B * l,
                       att_out + (size_t)k * T, (long)n * T, k, B, T, zone_u1 ? zone_u1 + (size_t)k * B * l : nullptr,
                       zone_u2 ? zone_u2 + (size_t)k * B * l : nullptr, cur)) return 1;
    if (have_cur) cur.attn.defer = defer;
    // (a job of an absent step is launched with zero blocks; its stale descriptor is never read)
    if (prof_mark(dec, WRNN_PROFILE_TICK_GRU, st)) return 1;
    if (!defer && merge_query) {
      // tick 1: GRU(k) || query(k) || LSTM1(k-2): the query blocks wait INSIDE the launch for the GRU's tagged output
      GruJob jg = have_cur ? cur.gru : off(cur.gru);
      jg.epi.hand = dec->qhand, jg.epi.hand_tag = (unsigned)k + 1, jg.epi.hand_pairs = d / 2;
      const QueryJob jq{cur.pq.wt, cur.pq.epi.bias, dec->qhand, qhand_err(dec),
                        cur.pq.epi.y, (unsigned)k + 1, B, cur.pq.tiles, have_cur ? (B + 15) / 16 : 0, 1};
      if (launch_tick(st, jg, jq, have_p2 ? p2j.l1 : off(p2j.l1))) return 1;
    } else if (!defer) {
      if (launch_tick(st, have_cur ? cur.gru : off(cur.gru), have_p2 ? p2j.l1 : off(p2j.l1))) return 1;
    } else {
      CombineJob jc{b.ascr, b.cum, b.att, att_out + (size_t)(k - 1) * T, (long)n * T, act_il(b.ctx2[(k - 1) & 1], B, d), k == 1, T,
                    have_p1 ? B : 0, chunks};
      const auto l1j = have_p2 ? p2j.l1 : off(p2j.l1);
      if (k == 0) {  // the first step's context is zero: plain source
        if (launch_tick(st, jc, cur.gru, l1j)) return 1;
      } else {
        const CombSrc cs{b.ascr, chunks, kRec, kD + kAT};
        const auto launch_comb = [&](auto jg) -> int {
          jg.wt = cur.gru.wt, jg.KC = cur.gru.KC, jg.tiles = cur.gru.tiles, jg.tgroups = cur.gru.tgroups;
          jg.groups = have_cur ? cur.gru.groups : 0, jg.rows = cur.gru.rows, jg.B = cur.gru.B, jg.multi = cur.gru.multi;
          jg.s0 = cur.gru.s0, jg.s1 = cur.gru.s1, jg.s2 = cur.gru.s2, jg.epi = cur.gru.epi, jg.comb = cs;
          return launch_tick(st, jc, jg, l1j);
        };
        if (chunks <= 4 ? launch_comb(GruCombJob4()) : launch_comb(GruCombJob8())) return 1;
      }
    }
    if (defer || !merge_query) {
      if (prof_mark(dec, WRNN_PROFILE_TICK_QUERY, st)) return 1;
      if (have_cur && launch_tick(st, cur.pq)) return 1;
    }
    if (prof_mark(dec, WRNN_PROFILE_TICK_ATTENTION, st)) return 1;
    if (launch_tick(st, have_cur ? cur.attn : off_attn(cur.attn), have_p1 ? p1j.rin : off(p1j.rin), have_p2 ? p2j.l2 : off(p2j.l2))) return 1;
  }
  // mel projection of every step at once, scattered into mel_out[B, n_mels, n*r]
  LinJob<2, 4, EpiMel, 2> jm;
  if (make_lin(jm, dec->mel[r], Src{act_il(x2, B, l), l}, kNoSrc, kNoSrc, (int)rows, B,
               EpiMel{mel_out, (long)dec->n_mels * n * r, (long)n * r, r, dec->n_mels * r})) return 1;
  if (prof_mark(dec, WRNN_PROFILE_MEL, st)) return 1;
  if (launch_tick(st, jm)) return 1;
  return prof_finish(dec, st);
}

namespace {

// ---- small-batch free-running decode as one persistent launch (kernel: taco_resident.h) -------------------------------
inline int resident_nb(int B) { return B <= 1 ? 1 : B <= 2 ? 2 : 4; }
constexpr size_t kLdsLimit = (size_t)160 << 10;

bool resident_supported(const wrnn_decoder* dec, int B, int T) {
  return dec->d == kD && dec->l == kRL && B <= kRMaxB && (long)B * T <= kRMaxBT && dec->n_cu >= kRW && dec->A % 4 == 0 &&
         resident_lds_floats(resident_nb(B), B, T) * sizeof(float) <= kLdsLimit;
}

int upload_vec(const std::vector<float>& h, float** dev) {
  if (*dev) return 0;
  return upload(h.data(), h.size() * sizeof(float), (void**)dev);
}

// hand-over arrays of one call, in u64 words from the start of res.gran (all zeroed before the launch): kA_count arrays of
// 4 buffers of kCap granules each (taco_resident.h), then the status words
struct GranLayout {
  static constexpr size_t status = (size_t)kA_count * 4 * kCap, total = status + 2 + 128;  // status: [0] error word (u32), [1] n_done (int), then the workgroups' XCD numbers (256 x u32)
};

int ensure_resident(wrnn_decoder* dec, int r) {
  auto& R = dec->res;
  const int l = dec->l;
  if (!R.ready) {
    ensure_l1_fold(dec);
    std::vector<float> l2b((size_t)4 * l);
    for (int i = 0; i < 4 * l; ++i) l2b[i] = dec->l2_bih[i] + dec->l2_bhh[i];
    if (upload_vec(dec->fc1_b_host, &R.fc1_b) || upload_vec(dec->fc2_w_h, &R.fc2_w) || upload_vec(dec->fc2_b_h, &R.fc2_b) ||
        upload_vec(dec->gru_wih_h, &R.gru_wih) || upload_vec(dec->gru_whh_h, &R.gru_whh) || upload_vec(dec->gru_bih_h, &R.gru_bih) ||
        upload_vec(dec->gru_bhh_h, &R.gru_bhh) || upload_vec(dec->pq_w_h, &R.pq_w) || upload_vec(dec->pq_b_h, &R.pq_b) ||
        upload_vec(dec->l1_wf_host, &R.l1_wf) || upload_vec(dec->l1_whh, &R.l1_whh) || upload_vec(dec->l1_bf_host, &R.l1_b) ||
        upload_vec(dec->rin_w, &R.rin_w) || upload_vec(dec->rin_b, &R.rin_b) || upload_vec(dec->l2_wih, &R.l2_wih) ||
        upload_vec(dec->l2_whh, &R.l2_whh) || upload_vec(l2b, &R.l2_b))
      return 1;
    WRNN_HIP(hipMalloc((void**)&R.gran, GranLayout::total * sizeof(u64)));
    R.ready = true;
  }
  if (!R.fc1m[r]) {
    ensure_fc1_mel_host(dec, r);
    if (upload_vec(dec->fc1m_host[r], &R.fc1m[r])) return 1;
    const int nmr = dec->n_mels * r;
    std::vector<float> mw((size_t)nmr * l);
    for (int j = 0; j < nmr; ++j) {
      const int m = j / r, f = j % r;
      std::copy_n(dec->mel_w_host.data() + (size_t)(m * kMaxR + f) * l, l, mw.data() + (size_t)j * l);
    }
    if (upload_vec(mw, &R.mel_w[r])) return 1;
  }
  return 0;
}

constexpr int kResNoLaunch = 2;   // the cooperative launch was refused (nothing enqueued)

template <int NB, int POS>
int launch_resident(const ResidentParams& prm, size_t lds_bytes, hipStream_t st) {
  auto* fn = resident_free_kernel<NB, POS>;
  WRNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // The 256 workgroups must all become resident (one per compute unit: each asks for more than half a CU's LDS and the
  // device has at least 256 CUs, checked by resident_supported; occupancy checked here).  A cooperative launch makes the
  // runtime check the same thing at launch time; it is used unless WRNN_RESIDENT_PLAIN_LAUNCH is set (rocprofv3's kernel
  // trace crashes on cooperative launches on this stack; a plain launch of the same grid has the same residency).
  int per_cu = 0;
  WRNN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), kRT, lds_bytes));
  WRNN_REQUIRE(per_cu >= 1, "resident decode: the kernel does not fit a compute unit (%zu bytes of LDS)", lds_bytes);
  // queued on the device's stream for persistent grids (common.h): it must not share the device with another such grid
  return wrnn::on_persistent_stream(st, [&](hipStream_t ps) -> int {
    if (getenv("WRNN_RESIDENT_PLAIN_LAUNCH")) {
      hipLaunchKernelGGL(fn, dim3(kRW), dim3(kRT), lds_bytes, ps, prm);
      WRNN_LAUNCH_CHECK();
      return 0;
    }
    void* args[] = {const_cast<ResidentParams*>(&prm)};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(kRW), dim3(kRT), args, (unsigned)lds_bytes, ps);
    if (e != hipSuccess) {   // e.g. another process holds compute units: nothing has run, the caller may take the launch loop
      (void)hipGetLastError();
      wrnn::fail("hipLaunchCooperativeKernel(resident_free_kernel) failed: %s", hipGetErrorString(e));
      return kResNoLaunch;
    }
    return 0;
  });
}

int run_free_resident(wrnn_decoder* dec, const float* enc_q, const float* enc_p, const float* cadd, int B, int T, int n, int r,
                      float thr, int use_stop, const float* drop_u1, const float* drop_u2, float* mel_out, float* att_out,
                      int* n_steps_out, hipStream_t st) {
  if (ensure_resident(dec, r)) return 1;
  auto& R = dec->res;
  WRNN_HIP(hipMemsetAsync(R.gran, 0, GranLayout::total * sizeof(u64), st));
  ResidentParams q{};
  q.fc1m = R.fc1m[r], q.fc1_b = R.fc1_b, q.fc2_w = R.fc2_w, q.fc2_b = R.fc2_b;
  q.gru_wih = R.gru_wih, q.gru_whh = R.gru_whh, q.gru_bih = R.gru_bih, q.gru_bhh = R.gru_bhh;
  q.pq_w = R.pq_w, q.pq_b = R.pq_b;
  q.conv_t = dec->conv_t, q.L_w = dec->L_w, q.L_b = dec->L_b, q.v_w = dec->v_w;
  q.l1_wf = R.l1_wf, q.l1_whh = R.l1_whh, q.l1_b = R.l1_b, q.rin_w = R.rin_w, q.rin_b = R.rin_b;
  q.l2_wih = R.l2_wih, q.l2_whh = R.l2_whh, q.l2_b = R.l2_b, q.mel_w = R.mel_w[r];
  q.enc_q = enc_q, q.enc_p = enc_p, q.cadd = cadd, q.drop_u1 = drop_u1, q.drop_u2 = drop_u2;
  q.mel_out = mel_out, q.att_out = att_out;
  q.err = reinterpret_cast<u32*>(R.gran + GranLayout::status);
  q.n_done = reinterpret_cast<int*>(R.gran + GranLayout::status + 1);
  q.xcc = reinterpret_cast<u32*>(R.gran + GranLayout::status + 2);
  q.map = getenv("WRNN_RESIDENT_LINEAR_MAP") ? 0 : 1, q.plain = getenv("WRNN_RESIDENT_WT") ? 0 : 1;
  q.arrays = R.gran;
  q.B = B, q.T = T, q.n = n, q.r = r, q.nmr = dec->n_mels * r;
  q.mel_sb = (long)dec->n_mels * n * r, q.mel_sm = (long)n * r;
  q.thr = thr, q.use_stop = use_stop;
  const char* stamp_file = getenv("WRNN_RESIDENT_STAMPS");
  u64* stamps = nullptr;
  if (stamp_file && n >= kStampStep0 + 8) {
    WRNN_HIP(hipMalloc((void**)&stamps, 6 * 8 * 4 * sizeof(u64)));
    WRNN_HIP(hipMemsetAsync(stamps, 0, 6 * 8 * 4 * sizeof(u64), st));
  }
  q.stamps = stamps;
  const int NB = resident_nb(B);
  const int bt = B * T;   // attention positions per AT workgroup: even, 64 of them cover B T
  const int pos = bt <= 128 ? 2 : bt <= 256 ? 4 : bt <= 512 ? 8 : 16;
  // at least 84 KiB so that no two workgroups ever share a compute unit (one per CU is what the hand-overs are priced for)
  const size_t lds = std::max(resident_lds_floats(NB, B, T) * sizeof(float), (size_t)84 << 10);
  WRNN_REQUIRE(lds <= kLdsLimit, "resident decode: %zu bytes of LDS needed", lds);
  int rc;
#define WRNN_POS_SWITCH(NB_) (pos == 2 ? launch_resident<NB_, 2>(q, lds, st) : pos == 4 ? launch_resident<NB_, 4>(q, lds, st) : \
                              pos == 8 ? launch_resident<NB_, 8>(q, lds, st) : launch_resident<NB_, 16>(q, lds, st))
  if (NB == 1) rc = WRNN_POS_SWITCH(1);
  else if (NB == 2) rc = WRNN_POS_SWITCH(2);
  else rc = WRNN_POS_SWITCH(4);
#undef WRNN_POS_SWITCH
  if (rc) {
    if (stamps) (void)hipFree(stamps);
    return rc;
  }
  // one launch per call: its status (bounded spins, steps decoded) is read back when it has finished
  u64 status[2] = {0, 0};
  WRNN_HIP(hipMemcpyAsync(status, R.gran + GranLayout::status, sizeof(status), hipMemcpyDeviceToHost, st));
  WRNN_HIP(hipStreamSynchronize(st));
  if (stamps) {  // diagnostic dump: per step and group (its first workgroup): step start, input swept, output published (PQ: p
                 // and pq), in us from P1's step start
    u64 h[6 * 8 * 4];
    WRNN_HIP(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    (void)hipFree(stamps);
    if (FILE* f = fopen(stamp_file, "w")) {
      static const char* names[6] = {"P1", "PQ", "GR", "AT", "L1", "L2"};
      for (int rw = 0; rw < 8; ++rw) {
        fprintf(f, "step%d:", kStampStep0 + rw);
        for (int gi = 0; gi < 6; ++gi) {
          fprintf(f, "  %s", names[gi]);
          for (int i = 0; i < 4; ++i) {
            const u64 v = h[(gi * 8 + rw) * 4 + i];
            if (v) fprintf(f, " %6.2f", (double)(long long)(v - h[rw * 4]) * 0.01);
          }
        }
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
  WRNN_REQUIRE((u32)status[0] == 0, "resident decode: a stage hand-over timed out (the 256 workgroups were not co-resident?)");
  *n_steps_out = (int)(u32)status[1];
  return 0;
}

}  // namespace

extern "C" int wrnn_decoder_set_tf_path(wrnn_decoder* dec, int path) {
  WRNN_REQUIRE(dec != nullptr, "null decoder handle");
  WRNN_REQUIRE(path == WRNN_TF_PATH_AUTO || path == WRNN_TF_PATH_LAUNCHES || path == WRNN_TF_PATH_CHAIN || path == WRNN_TF_PATH_CHAIN_LAUNCH_STACK,
               "wrnn_decoder_set_tf_path: unknown path %d", path);
  dec->tf_path = path;
  return 0;
}

extern "C" int wrnn_decoder_set_free_path(wrnn_decoder* dec, int path) {
  WRNN_REQUIRE(dec != nullptr, "null decoder handle");
  WRNN_REQUIRE(path == WRNN_FREE_PATH_AUTO || path == WRNN_FREE_PATH_LAUNCHES || path == WRNN_FREE_PATH_RESIDENT,
               "wrnn_decoder_set_free_path: unknown path %d", path);
  dec->free_path = path;
  return 0;
}

extern "C" int wrnn_decoder_run_free(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                                     const float* aligner_feats, int B, int T, int max_steps, int r,
                                     float stop_threshold, int use_stop, const float* drop_u1,
                                     const float* drop_u2, float* mel_out, float* att_out, int* n_steps_out,
                                     void* stream) {
  if (check_common(dec, B, T, r)) return 1;
  WRNN_REQUIRE(max_steps >= 1, "wrnn_decoder_run_free: max_steps=%d must be >= 1", max_steps);
  WRNN_REQUIRE(enc_q && enc_p && mel_out && att_out && n_steps_out, "wrnn_decoder_run_free: null pointer");
  WRNN_REQUIRE((drop_u1 == nullptr) == (drop_u2 == nullptr), "give both dropout uniform tensors or neither");
  Guard g(dec->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", dec->device);
  hipStream_t st = (hipStream_t)stream;
  const int d = dec->d, l = dec->l, P = dec->P, A = dec->A, n = max_steps;
  const int A4 = (A + 3) / 4 * 4;
  const bool with_ali = aligner_feats != nullptr && A > 0;
  const size_t ali_floats = with_ali ? (size_t)n * B * (A4 + kFc1) : 0;
  const size_t need = (size_t)B * (P + kFc1 + kFc2 + l) + ali_floats + loop_floats(B, T, d, l) + 64;
  if (ensure_ws(dec, need)) return 1;
  // B <= 4: one persistent, weights-resident launch for the whole call (taco_resident.h: 9.6 / 12.2 / 19.6 us per step at
  // B = 1 / 2 / 4 against 36 for the launch loop at any small B); larger batches, other dims, long rows: launches
  bool resident = dec->free_path != WRNN_FREE_PATH_LAUNCHES && resident_supported(dec, B, T);
  WRNN_REQUIRE(resident || dec->free_path != WRNN_FREE_PATH_RESIDENT,
               "wrnn_decoder_run_free: the resident path needs B <= %d, B*T <= %d with room in LDS (about 40 * B * T floats), lstm_dims = %d and %d compute units",
               kRMaxB, kRMaxBT, kRL, kRW);
  if (resident) {
    float* cadd_r = nullptr;
    if (with_ali) {
      Carver cr{dec->ws};
      float* alig_r = cr.take((size_t)n * B * A4);
      cadd_r = cr.take((size_t)n * B * kFc1);
      const size_t total = (size_t)n * B * A4;
      hipLaunchKernelGGL(gather_aligner_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                         st, aligner_feats, alig_r, B, A, A4, n);
      WRNN_LAUNCH_CHECK();
      LinJob<2, 4, EpiLinear, 2> ja;
      if (make_lin(ja, dec->fc1a, Src{act_il(alig_r, B, A4), A4}, kNoSrc, kNoSrc, n * B, B,
                   EpiLinear{dec->fc1a.bias, act_il(cadd_r, B, kFc1), kFc1, 0, nullptr})) return 1;
      if (launch_tick(st, ja)) return 1;
    }
    const int rc = run_free_resident(dec, enc_q, enc_p, cadd_r, B, T, n, r, stop_threshold, use_stop, drop_u1, drop_u2, mel_out, att_out,
                                     n_steps_out, st);
    // AUTO: a refused cooperative launch (the device is shared with another process) is not an error, the launch loop runs
    if (!(rc == kResNoLaunch && dec->free_path == WRNN_FREE_PATH_AUTO)) return rc ? 1 : 0;
    wrnn::last_error().clear();
  }
  if (ensure_mel_layer(dec, r)) return 1;
  if (ensure_fc1_mel_layer(dec, r)) return 1;
  if (ensure_free_layers(dec)) return 1;
  if (ensure_counters(dec, B, st)) return 1;
  if (dec->flags_n < n) {
    if (dec->flags) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(dec->flags));
      dec->flags = nullptr;
    }
    WRNN_HIP(hipMalloc((void**)&dec->flags, (size_t)n * sizeof(int)));
    dec->flags_n = n;
  }
  Carver cv{dec->ws};
  float* pin = cv.take((size_t)B * P);
  float* p1 = cv.take((size_t)B * kFc1);
  float* pp = cv.take((size_t)B * kFc2);
  float* x2 = cv.take((size_t)B * l);
  float* alig = with_ali ? cv.take((size_t)n * B * A4) : nullptr;    // [n][A4/4][B][4]
  float* cadd = with_ali ? cv.take((size_t)n * B * kFc1) : nullptr;  // [n][256/4][B][4]
  LoopBufs b;
  carve_loop(cv, b, B, T, d, l);
  if (zero_loop_state(b, B, d, l, st)) return 1;
  WRNN_HIP(hipMemsetAsync(pin, 0, (size_t)B * P * sizeof(float), st));  // go frame
  WRNN_HIP(hipMemsetAsync(dec->flags, 0, (size_t)n * sizeof(int), st));
  // The feedback  prenet_in(k) = [last frame of group k-1 ; aligner_feats[:, :, k-1]]  never materialises:
  // the last frame is linear in LSTM2's output x2, so fc1's pre-activation is
  //   (W1[:, :n_mels] . Wmel_last) . x2(k-1)  +  (W1[:, n_mels:] . aligner_feats[:, :, k-1] + b1)
  // with the first product folded into one weight matrix (ensure_fc1_mel_layer) and the second hoisted
  // out of the loop as one GEMM over all steps.  fc1(k) then rides in the launch that projects the mel
  // frames of group k-1, which drops the mel projection, the feedback gather and fc1 from the step's
  // dependent chain (10 -> 8 launches per step).
  if (with_ali) {
    const size_t total = (size_t)n * B * A4;
    hipLaunchKernelGGL(gather_aligner_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                       st, aligner_feats, alig, B, A, A4, n);
    WRNN_LAUNCH_CHECK();
    LinJob<2, 4, EpiLinear, 2> ja;
    if (make_lin(ja, dec->fc1a, Src{act_il(alig, B, A4), A4}, kNoSrc, kNoSrc, n * B, B,
                 EpiLinear{dec->fc1a.bias, act_il(cadd, B, kFc1), kFc1, 0, nullptr})) return 1;
    if (launch_tick(st, ja)) return 1;
  }
  const long sb = (long)dec->n_mels * n * r, sm = (long)n * r;
  constexpr int kPoll = 32;  // host looks at the stop votes every kPoll groups
  std::vector<int> votes(n, 1);
  int done = n;
  // votes of groups [lo, hi) are complete on the device once the launch carrying mel(hi-1) is enqueued
  auto poll = [&](int lo, int hi, bool& stop) -> int {
    stop = false;
    if (hi <= lo) return 0;
    WRNN_HIP(hipMemcpyAsync(votes.data() + lo, dec->flags + lo, (size_t)(hi - lo) * sizeof(int), hipMemcpyDeviceToHost, st));
    WRNN_HIP(hipStreamSynchronize(st));
    for (int kk = lo; kk < hi; ++kk)
      if (kk > 10 && votes[kk] == 0) {  // (mel < threshold).all() and k > 10
        done = kk + 1;
        stop = true;
        break;
      }
    return 0;
  };
  auto mel_job = [&](int k, LinJob<1, 1, EpiMel>& jm) {  // group k's frames straight into mel_out[:, :, k*r : (k+1)*r]
    return make_lin(jm, dec->mel[r], Src{act_il(x2, B, l), l}, kNoSrc, kNoSrc, B, B,
                    EpiMel{mel_out + (size_t)k * r, sb, sm, r, dec->n_mels * r, dec->flags + k, stop_threshold});
  };
  // Nothing rides along in a free-running step: every stage is on the dependent chain, so the chain is shortened
  // (7 launches per step; 6 with the query projection riding in the GRU launch, 5 with fc2 riding in fc1's, 4 with the GRU
  //  and the query projection there too: the default):
  //   A  fc1 folded (x2(k-1))  || mel frames of group k-1
  //   B  fc2        C  attention GRU        D  query projection        E  attention
  //   F  LSTM1 folded through rnn_input: gates = (W_ih1 . W_rin) . [ctx | attn_h] + W_hh1 . h1 + b'
  //      || y = rnn_input([ctx | attn_h])   (only the residual path still needs it)
  //   G  LSTM2 over [y | h1 | h2] (x1 = y + h1 is never formed), x2 = y + h1 + h2
  int polled = 0;
  bool stopped = false;
  const bool merge_query = getenv("WRNN_SEPARATE_QUERY_LAUNCH") == nullptr;
  // One more in-launch hand-over of the same kind (LinJob's HAND source; WRNN_FREE_SIX_LAUNCHES=1: without it): fc2 waits in
  // fc1's launch for the first prenet layer -- 5 launches per step, A + B 7.4 -> 6.3 us.  WRNN_FREE_MERGE_ATTENTION=1
  // (measured, not the default): rnn_input and LSTM1 wait in the attention's launch for the context, their attention-state
  // and h1 halves done while the attention still runs -- 4 launches per step, but that launch takes 17.2 us, what the two
  // took (9.2 + 7.9): the weights' and memories' way into the compute units is what both are bound by, and it is shared.
  const bool merge_fc2 = merge_query && getenv("WRNN_FREE_SIX_LAUNCHES") == nullptr;
  const bool merge_attn = merge_fc2 && getenv("WRNN_FREE_MERGE_ATTENTION") != nullptr;
  // The attention GRU and the query projection in that first launch as well, the cell waiting for the prenet's output
  // (K = [ctx | attn_h | prenet]: the same sums in another order, fp32 rounding apart; WRNN_FREE_NO_GRU_MERGE=1: a launch
  // of their own as before): fc1 || mel || fc2 || GRU || query, attention, LSTM1 || rnn_input, LSTM2 -- 4 launches per step.
  const bool merge_gru = merge_fc2 && getenv("WRNN_FREE_NO_GRU_MERGE") == nullptr;
  if (merge_query && ensure_qhand(dec, B, st)) return 1;
  unsigned long long* const phand = merge_fc2 ? qhand_region(dec, 1) : nullptr;
  unsigned long long* const chand = merge_attn ? qhand_region(dec, 2) : nullptr;
  unsigned long long* const ghand = merge_gru ? qhand_region(dec, 3) : nullptr;   // (regions: only after ensure_qhand)
  WRNN_REQUIRE(!merge_query || dec->qhand != nullptr, "wrnn_decoder_run_free: the hand-over arrays are missing");
  const Act ctx = act_il(b.ctx, B, d), yv = act_il(b.y, B, l), x2v = act_il(x2, B, l), none = kNoSrc.a;
  const Act c1 = act_il(b.c1, B, l), c2 = act_il(b.c2, B, l);
  const int chunks = attn_chunks(T);
  for (int k = 0; k < n && !stopped; ++k) {
    const float* u1 = drop_u1 ? drop_u1 + (size_t)k * B * kFc1 : nullptr;
    const float* u2 = drop_u2 ? drop_u2 + (size_t)k * B * kFc2 : nullptr;
    const int a = k & 1, o = a ^ 1;
    const Act ah_in = act_il(b.ah[a], B, d), ah_out = act_il(b.ah[o], B, d);
    const Act h1_in = act_il(b.h1[a], B, l), h1_out = act_il(b.h1[o], B, l);
    const Act h2_in = act_il(b.h2[a], B, l), h2_out = act_il(b.h2[o], B, l);
    LinearJob f1, f2, jq, jr;
    LinJob<1, 1, EpiLinear, 4, 0, 16> f2h, jrh;
    LinJob<2, 1, EpiLstm2, 4, 0, 16> j1h;
    LinJob<1, 1, EpiGru, 4, 0, 8> jgh;
    LinJob<1, 1, EpiMel> jm;
    GruJob jg;
    LinJob<2, 1, EpiLstm2> j1, j2;
    const unsigned tag = (unsigned)k + 1;
    // ---- the jobs of the step
    EpiLinear e1{k == 0 || !with_ali ? dec->fc1.bias : nullptr, act_il(p1, B, kFc1), kFc1, 1, u1};
    if (k > 0 && with_ali) e1.add = act_il(cadd + (size_t)(k - 1) * B * kFc1, B, kFc1);
    if (merge_fc2) e1.hand = phand, e1.hand_tag = tag, e1.hand_pairs = kFc1 / 2;
    if (k == 0) {  // go frame: fc1 over zeros
      if (make_lin(f1, dec->fc1, Src{act_il(pin, B, P), P}, kNoSrc, kNoSrc, B, B, e1)) return 1;
    } else {
      if (mel_job(k - 1, jm)) return 1;
      if (make_lin(f1, dec->fc1m[r], Src{x2v, l}, kNoSrc, kNoSrc, B, B, e1)) return 1;
    }
    EpiLinear e2{dec->fc2.bias, act_il(pp, B, kFc2), kFc2, 1, u2};
    if (merge_gru) e2.hand = ghand, e2.hand_tag = tag, e2.hand_pairs = kFc2 / 2;
    if (merge_fc2) {
      if (make_lin(f2h, dec->fc2, kNoSrc, kNoSrc, kNoSrc, B, B, e2)) return 1;
      f2h.hand = phand, f2h.herr = qhand_err(dec), f2h.htag = tag;   // (no sentinel: the wait is short, and nothing beside it is bound by the memory path)
    } else {
      if (make_lin(f2, dec->fc2, Src{act_il(p1, B, kFc1), kFc1}, kNoSrc, kNoSrc, B, B, e2)) return 1;
    }
    EpiGru eg{dec->gru.bias, ah_in, ah_out};
    if (merge_query) eg.hand = dec->qhand, eg.hand_tag = tag, eg.hand_pairs = d / 2;
    if (merge_gru) {
      if (make_lin(jgh, dec->gru_h, Src{ctx, d}, Src{ah_in, d}, kNoSrc, B, B, eg)) return 1;
      jgh.hand = ghand, jgh.herr = qhand_err(dec), jgh.htag = tag;
    } else {
      if (make_lin(jg, dec->gru, Src{ctx, d}, Src{act_il(pp, B, kFc2), kFc2}, Src{ah_in, d}, B, B, eg)) return 1;
    }
    if (make_lin(jq, dec->pq, Src{ah_out, d}, kNoSrc, kNoSrc, B, B, EpiLinear{dec->pq.bias, act_il(b.pq, B, d), d, 0, nullptr})) return 1;
    // (merge_query: the query projection waits inside the GRU's launch for the cell's tagged output)
    const QueryJob jqm{jq.wt, jq.epi.bias, dec->qhand, merge_query ? qhand_err(dec) : nullptr,
                       jq.epi.y, tag, B, jq.tiles, (B + 15) / 16, 0};
    AttnJob ja{enc_p, enc_q, dec->conv_t, dec->L_w, dec->L_b, dec->v_w, act_il(b.pq, B, d), b.cum, b.att,
               att_out + (size_t)k * T, (long)n * T, ctx, b.ascr, dec->row_counter, k == 0, T, B, chunks};
    const EpiLstm2 el1{dec->fr_l1f.bias, c1, c1, h1_out, none, none, none};
    const EpiLinear eri{dec->rin.bias, yv, l, 0, nullptr};
    if (merge_attn) {
      ja.hand = chand, ja.hand_tag = tag;
      if (make_lin(j1h, dec->fr_l1h, Src{ah_out, d}, Src{h1_in, l}, kNoSrc, B, B, el1)) return 1;
      if (make_lin(jrh, dec->fr_rinh, Src{ah_out, d}, kNoSrc, kNoSrc, B, B, eri)) return 1;
      j1h.hand = jrh.hand = chand, j1h.herr = jrh.herr = qhand_err(dec), j1h.htag = jrh.htag = tag;
      j1h.hstride = jrh.hstride = 128;   // (a row's context comes from one block)
    } else {
      if (make_lin(j1, dec->fr_l1f, Src{ctx, d}, Src{ah_out, d}, Src{h1_in, l}, B, B, el1)) return 1;
      if (make_lin(jr, dec->rin, Src{ctx, d}, Src{ah_out, d}, kNoSrc, B, B, eri)) return 1;
    }
    if (make_lin(j2, dec->fr_l2x, Src{yv, l}, Src{h1_out, l}, Src{h2_in, l}, B, B,
                 EpiLstm2{dec->fr_l2x.bias, c2, c2, h2_out, yv, h1_out, x2v})) return 1;
    // ---- the launches
    if (merge_gru) {   // fc1 || mel || fc2 || GRU || query: one launch
      if (k == 0 ? launch_tick_n(st, f1, f2h, jgh, jqm) : launch_tick_n(st, f1, jm, f2h, jgh, jqm)) return 1;
    } else {
      if (merge_fc2) {
        if (k == 0 ? launch_tick(st, f1, f2h) : launch_tick(st, f1, jm, f2h)) return 1;
      } else {
        if ((k == 0 ? launch_tick(st, f1) : launch_tick(st, f1, jm)) || launch_tick(st, f2)) return 1;
      }
      if (merge_query) {
        if (launch_tick(st, jg, jqm)) return 1;
      } else {
        if (launch_tick(st, jg) || launch_tick(st, jq)) return 1;
      }
    }
    if (merge_attn) {
      if (launch_tick(st, ja, jrh, j1h) || launch_tick(st, j2)) return 1;
    } else {
      if (launch_tick(st, ja) || launch_tick(st, jr, j1) || launch_tick(st, j2)) return 1;
    }
    if (use_stop && k - polled >= kPoll) {  // groups polled .. k-1 have been projected
      if (poll(polled, k, stopped)) return 1;
      polled = k;
    }
  }
  if (!stopped) {  // the last group's frames, and the votes not looked at yet
    LinJob<1, 1, EpiMel> jm;
    if (mel_job(n - 1, jm)) return 1;
    if (launch_tick(st, jm)) return 1;
    if (use_stop && poll(polled, n, stopped)) return 1;
  }
  if (use_stop && merge_query) {
    // the call has synchronised with the stream anyway: look at the hand-overs' error word (a wait inside a launch that gave
    // up has poisoned its block's output with NaN; say so instead of returning it)
    unsigned herr = 0;
    WRNN_HIP(hipMemcpyAsync(&herr, qhand_err(dec), sizeof(herr), hipMemcpyDeviceToHost, st));
    WRNN_HIP(hipStreamSynchronize(st));
    WRNN_REQUIRE(herr == 0, "wrnn_decoder_run_free: an in-launch hand-over timed out (the outputs hold NaN)");
  }
  *n_steps_out = done;
  return 0;
}
