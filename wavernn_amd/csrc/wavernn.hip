// WaveRNN sample-by-sample generation loop for gfx950 (MI355X), fp32 storage and accumulate.
//
// SURVEY.md section 8 rows a9 / f4.  The WaveRNN vocoder is ABSENT from the mounted reference tree (no
// model file, no `wavernn` subcommand; only hyper-parameters in tests/resources/test_config.yaml:47-75
// and the mixture-of-logistics sampler in utils/distribution.py survive), so this loop is BUILD-DEFINED
// after the fatchord WaveRNN the reference forked from: parity is unpinned and is checked against the
// build's own CPU restatement only (oracle/wavernn_oracle.py).
//
// One step, per fold (batch row) b, with m_t / a1..a4_t the upsampled mel and auxiliary features:
//   x  = I([x_prev ; m_t ; a1_t])                                   Linear(1 + feat + aux -> rnn)
//   h1 = GRUCell(x, h1) ; x = x + h1
//   h2 = GRUCell([x ; a2_t], h2) ; x = x + h2
//   x  = relu(fc1([x ; a3_t])) ; x = relu(fc2([x ; a4_t])) ; logits = fc3(x)
//   RAW mode: class c ~ softmax(logits) by inverse CDF with the caller's uniform u_t ; x_prev = 2c/(C-1) - 1
// Structure of the loop (5 launches per step):
//  * every matrix stage is one launch of a linjob.h GEMM job with its pointwise work fused as epilogue; the
//    per-step feature slices are extra K sources of the jobs (no concatenation is materialised);
//  * the input layer I is hoisted: its mel / aux columns are one GEMM over all steps before the loop (cI), its
//    sample column w_x is applied by the sampler;
//  * the first GRU cell needs no product in the loop at all (Gru1Row below): its input half is hoisted too, its
//    hidden half is computed ahead, the elementwise rest runs in the sampler right after the draw;
//  * the hidden-state half of each cell's contraction (W_hh . h, known since the previous step) is computed by jobs
//    riding in the fc1 (cell 1) and sampler (cell 2) launches, so the second cell's own launch contracts over its
//    input columns only;
//  * no launch parameter depends on the step or the call (the step index lives in per-row device counters, the
//    call's pointers in a device-side call block, the step's aux slices in a fixed buffer the sampler refills), so
//    64 steps are captured once as a HIP graph and replayed.
// The folds of one utterance (or of several, WaveRNN.generate_batch) are the batch rows.
#include "linjob.h"
#include "mol.h"
#include "seam.h"

namespace {

// step-major interleaved copies of the per-step features: dst[t][K4][B][4] = src[b][t][col0 + 4*k4 ..]
__global__ void wavernn_pack_kernel(const float* __restrict__ src, int ld, int col0, int K, float* __restrict__ dst,
                                    int B, int seq) {
  const int K4 = K / 4;
  const size_t total = (size_t)seq * K4 * B;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % B);
    const size_t r = i / B;
    const int k4 = (int)(r % K4), t = (int)(r / K4);
    reinterpret_cast<f32x4*>(dst)[i] =
        *reinterpret_cast<const f32x4*>(src + ((size_t)b * seq + t) * ld + col0 + 4 * k4);
  }
}

// What one wrnn_wavernn_generate call hands the per-step kernels through device memory, so that the
// kernels' launch parameters are the same on every step and every call (the step is a captured graph).
struct SampleCall {
  const float* u;         // [B][seq] uniforms (RAW) or [B][seq][nr_mix + 1] (MOL)
  const float* x_forced;  // [B][seq] or null
  float* out;             // [B][seq]
  float* logits_out;      // [B][seq][C] or null
  const float* cI;        // [seq][R/4][B][4] hoisted input layer (without its sample column)
  const float* G;         // [seq][3R/4][B][4] hoisted input half of the first GRU cell (without the sample term)
  const float* aux;       // [seq][4A/4][B][4] the four aux slices
  int seq;
};

// The whole call block in one round trip.  Read field by field the compiler sinks every scalar load to its first
// use, each behind the branches in between with a wait of its own: eight dependent ~0.3 us round trips at the head of
// the sampler.  Four 16-byte loads issued together and pinned here cost one.
__device__ __forceinline__ SampleCall load_call(const SampleCall* callp) {
  static_assert(sizeof(SampleCall) == 64, "SampleCall is read as four 16-byte pieces");
  int zero;  // opaque to the compiler: keeps the loads vector loads (a uniform address would be scalarised and sunk again)
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  const uint4* cp = reinterpret_cast<const uint4*>(callp) + zero;
  const uint4 q[4] = {cp[0], cp[1], cp[2], cp[3]};
  unsigned w[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // readfirstlane is convergent: it stays here, and with it the loads and their one wait
    w[4 * i + 0] = __builtin_amdgcn_readfirstlane(q[i].x), w[4 * i + 1] = __builtin_amdgcn_readfirstlane(q[i].y);
    w[4 * i + 2] = __builtin_amdgcn_readfirstlane(q[i].z), w[4 * i + 3] = __builtin_amdgcn_readfirstlane(q[i].w);
  }
  SampleCall c;
  __builtin_memcpy(&c, w, sizeof(c));
  return c;
}

// The first GRU cell needs no matrix product inside the loop.  Its input is the input layer's output
//   xI(t) = w_x * x(t-1) + cI(t),            cI = W_I[:, 1:] . [m ; a1] + b_I      (hoisted, one GEMM over all steps)
// so its input half is  W_ih . xI(t) = (W_ih . w_x) * x(t-1) + G(t),  G = (W_ih . W_I[:, 1:]) . [m ; a1] + W_ih . b_I + b
// (hoisted as well, the two matrices folded at create), and its hidden half  hp = W_hh . h1(t-1) + [0, 0, b_hn]  only
// needs the previous state and is computed ahead by a job riding in another launch.  What is left per step is
// elementwise in the sample x: it runs in the sampler, right after the sample is drawn.
//
// Thread mapping: every operand is step-major interleaved [n4][B][4], i.e. for one group of 4 units the rows are
// adjacent 16-B pieces.  A workgroup therefore owns 8 ROWS x 32 unit groups (thread = (row, unit group), row
// fastest), so that 8 threads read one full 128-B line; one workgroup per row would touch a line per thread and use
// an eighth of it (measured: 4 us per step at 20 rows).
struct Gru1Row {
  const float *w_x /*[R]*/, *ug /*[3R] = W_ih . w_x*/, *hp /*[3R/4][B][4]*/;
  float *h1 /*[R/4][B][4], updated in place*/, *x1 /*[R/4][B][4] = xI + h1'*/;
  int R, B;
  static constexpr int kRows = 8, kGroups = 32, kPer = kRows * kGroups / 256;  // unit groups per thread: 1
  __host__ __device__ int slices() const { return (R / 4 + kGroups - 1) / kGroups; }
  struct Pre { f32x4 g[3], hq[3], hv, cv, wv, uv[3]; };
  __device__ __forceinline__ Pre load(const float* G_t, const float* cI_t, int b, int n4) const {
    Pre q;
    const int R4 = R / 4;
#pragma unroll
    for (int gsel = 0; gsel < 3; ++gsel) {
      q.g[gsel] = *reinterpret_cast<const f32x4*>(G_t + ((size_t)(gsel * R4 + n4) * B + b) * 4);
      q.hq[gsel] = *reinterpret_cast<const f32x4*>(hp + ((size_t)(gsel * R4 + n4) * B + b) * 4);
      q.uv[gsel] = *reinterpret_cast<const f32x4*>(ug + gsel * R + 4 * n4);
    }
    q.hv = *reinterpret_cast<const f32x4*>(h1 + ((size_t)n4 * B + b) * 4);
    q.cv = *reinterpret_cast<const f32x4*>(cI_t + ((size_t)n4 * B + b) * 4);
    q.wv = *reinterpret_cast<const f32x4*>(w_x + 4 * n4);
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, float x, int b, int n4) const {
    f32x4 hn, xo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float r = sigmoidf_(fmaf(q.uv[0][i], x, q.g[0][i]) + q.hq[0][i]);
      const float z = sigmoidf_(fmaf(q.uv[1][i], x, q.g[1][i]) + q.hq[1][i]);
      const float n = tanhf_(fmaf(q.uv[2][i], x, q.g[2][i]) + r * q.hq[2][i]);
      hn[i] = (1.0f - z) * n + z * q.hv[i];
      xo[i] = fmaf(q.wv[i], x, q.cv[i]) + hn[i];  // residual: xI + h1'
    }
    *reinterpret_cast<f32x4*>(h1 + ((size_t)n4 * B + b) * 4) = hn;
    *reinterpret_cast<f32x4*>(x1 + ((size_t)n4 * B + b) * 4) = xo;
  }
  // the thread's unit group j of slice sl, clamped for the loads
  __device__ __forceinline__ int group(int sl, int j) const { return sl * kGroups + (threadIdx.x >> 3) + 32 * j; }
};

// Start of a call: publish the call block, zero the per-row step counters, stage step 0's aux slices and run
// the first GRU cell of step 0 (previous sample zero, zero state; hp already holds W_hh . 0 + bias).
// Grid: (rows / 8) x slices, thread mapping as in Gru1Row.
__global__ __launch_bounds__(256) void wavernn_begin_kernel(SampleCall call, SampleCall* __restrict__ dst,
                                                            int* __restrict__ tcount, int B, int A4 /*4*aux*/,
                                                            float* __restrict__ a_cur, Gru1Row g1) {
  const int slices = g1.slices();
  const int oct = blockIdx.x / slices, sl = blockIdx.x - oct * slices, tid = threadIdx.x;
  const int b = oct * Gru1Row::kRows + (tid & 7);
  if (blockIdx.x == 0 && tid == 0) *dst = call;
  if (b >= B) return;
  if (sl == 0) {
    if (tid < Gru1Row::kRows) tcount[b] = 0;
    for (int n4 = tid >> 3; n4 < A4 / 4; n4 += 32)
      *reinterpret_cast<f32x4*>(a_cur + ((size_t)n4 * B + b) * 4) = *reinterpret_cast<const f32x4*>(call.aux + ((size_t)n4 * B + b) * 4);
  }
#pragma unroll
  for (int j = 0; j < Gru1Row::kPer; ++j) {
    const int n4 = g1.group(sl, j);
    if (n4 < g1.R / 4) g1.apply(g1.load(call.G, call.cI, b, n4), 0.0f, b, n4);
  }
}

// RAW-mode draw by a half wave: lane ln (0..31) holds the logits of classes ln * PER .. ln * PER + PER - 1.  p = softmax in
// fp32; the class is the number of inclusive-CDF values (accumulated in float64, lane by lane, then a scan over the
// lanes) that are <= u * total, clamped to C - 1.
template <int PER>
__device__ __forceinline__ int raw_draw(const float (&l)[PER], float uu, int ln, int C) {
  // reductions over the 32 lanes of a half wave: four DPP steps inside each row of 16, then ONE cross-row exchange
  // (ds_bpermute); the same through five shuffles each costs five LDS-crossbar round trips per reduction
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) m = fmaxf(m, l[i]);
  m = dpp_max<0xB1>(m), m = dpp_max<0x4E>(m), m = dpp_max<0x141>(m), m = dpp_max<0x140>(m);
  m = fmaxf(m, __shfl_xor(m, 16, 32));
  double run = 0.0, cdf[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    run += (double)__expf(l[i] - m);
    cdf[i] = run;
  }
  // inclusive prefix of the lane totals within the half wave: Hillis-Steele inside the rows of 16 (lanes shifted in from
  // outside a row read 0), then the first row's total (its lane 15) added to the second row (row_bcast15 into rows 1, 3)
  double incl = run;
  incl += dpp_shifted<0x111, 0xf>(incl);
  incl += dpp_shifted<0x112, 0xf>(incl);
  incl += dpp_shifted<0x114, 0xf>(incl);
  incl += dpp_shifted<0x118, 0xf>(incl);
  incl += dpp_shifted<0x142, 0xa>(incl);
  const double total = __shfl(incl, 31, 32);
  const double base = incl - run;
  const double thr = (double)uu * total;
  float cnt = 0.0f;   // small integers: exact in fp32
#pragma unroll
  for (int i = 0; i < PER; ++i) cnt += (base + cdf[i] <= thr) ? 1.0f : 0.0f;
  cnt = dpp_add<0xB1>(cnt), cnt = dpp_add<0x4E>(cnt), cnt = dpp_add<0x141>(cnt), cnt = dpp_add<0x140>(cnt);
  cnt += __shfl_xor(cnt, 16, 32);
  return min((int)cnt, C - 1);
}

// RAW-mode sampler + first GRU cell of the next step.  A workgroup owns 8 rows and one slice of 32 unit groups
// (Gru1Row); each of its 8 half-waves draws one row's sample (every slice's workgroup redundantly: no exchange across
// workgroups), the samples go through LDS to the (row, unit group) threads.  p = softmax(logits) in fp32; the class
// is the number of inclusive-CDF values (accumulated in float64) that are <= u * total, clamped to C-1.
// A row's step t comes from its counter: read from tc_in, t+1 written to tc_out (the two alternate from step to
// step, so that a workgroup of the same launch that starts late still reads the step it belongs to).  Slice 0
// writes the sample into out[b][t] and stages the next step's aux slices in a_cur.  With x_forced the fed-back x is
// the caller's (teacher forcing).
// PER == 0 selects the MOL mode instead: the C = 3 * nr_mix outputs of fc3 are the parameters of a mixture of logistics
// (logit_probs | means | log_scales), sampled as the reference's sample_from_discretized_mix_logistic does
// (utils/distribution.py:87-123; arithmetic in mol.h) from nr_mix + 1 caller uniforms per step; lane i < nr_mix of the
// row's half-wave scores component i, a shuffle argmax (first maximum wins) picks, the sample is the step's output.
template <int PER /* classes per lane = C / 32 */>
struct SampleJob {  // a tick job (linjob.h)
  const float* logits;  // [Cp/4][B][4], Cp = C rounded up to a multiple of 4
  int C, B, A4;
  const SampleCall* callp;
  const int* tc_in;
  int* tc_out;
  float* a_cur;  // [4A/4][B][4]
  Gru1Row g1;
  __host__ __device__ int nblocks() const { return (B + Gru1Row::kRows - 1) / Gru1Row::kRows * g1.slices(); }
  static constexpr size_t lds_bytes() { return 64; }
  __device__ __forceinline__ void run(int bid, char* smem) const {
    float* xs = reinterpret_cast<float*>(smem);  // [8] fed-back sample of the workgroup's rows
    const int slices = g1.slices(), R = g1.R;
    const int oct = bid / slices, sl = bid - oct * slices, tid = threadIdx.x;
    // sampling: half-wave rs, lane ln;  first cell: row rg, unit groups group(sl, j)
    const int rs = tid >> 5, ln = tid & 31, rg = tid & 7;
    const int row0 = oct * Gru1Row::kRows;
    const bool live_s = row0 + rs < B, live_g = row0 + rg < B;
    const int bs = min(row0 + rs, B - 1), bg = min(row0 + rg, B - 1);  // rows past B: computed on row B-1, nothing stored
    // two dependent memory round trips only: {call block, step counter} -> {every operand of the step}
    const SampleCall call = load_call(callp);
    const int t = tc_in[bs], seq = call.seq;  // every row of a call is at the same step
    if (t >= seq) return;  // a replayed graph can hold more steps than the call has left: no-op
    const int tn = min(t + 1, seq - 1);  // the staging loads are unconditional (clamped), the stores are not
    const float* cI_next = call.cI + (size_t)tn * B * R;
    const float* G_next = call.G + (size_t)tn * B * 3 * R;
    const float* a_next = call.aux + (size_t)tn * B * A4;
    Gru1Row::Pre gp[Gru1Row::kPer];
#pragma unroll
    for (int j = 0; j < Gru1Row::kPer; ++j) gp[j] = g1.load(G_next, cI_next, bg, min(g1.group(sl, j), R / 4 - 1));
    constexpr int kMaxA = 4;  // 4*aux <= 512 floats per row: at most 4 float4 per (row, unit-group) thread
    f32x4 av[kMaxA];
#pragma unroll
    for (int j = 0; j < kMaxA; ++j)
      av[j] = *reinterpret_cast<const f32x4*>(a_next + ((size_t)min((tid >> 3) + 32 * j, A4 / 4 - 1) * B + bg) * 4);
    const float xf = call.x_forced ? call.x_forced[(size_t)bs * seq + t] : 0.0f;
    float sample;
    if constexpr (PER == 0) {
      const int nm = C / 3, ci = min(ln, nm - 1);
      const auto lg = [&](int c) { return logits[((size_t)(c >> 2) * B + bs) * 4 + (c & 3)]; };
      const float* up = call.u + ((size_t)bs * seq + t) * (nm + 1);
      const float lp = lg(ci), mean = lg(nm + ci), ls = lg(2 * nm + ci), um = up[ci], ul = up[nm];
      float sc = ln < nm ? wrnn_mol::gumbel_score(lp, um) : -INFINITY;
      int k = ln;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) {
        const float so = __shfl_xor(sc, off, 32);
        const int ko = __shfl_xor(k, off, 32);
        if (so > sc || (so == sc && ko < k)) sc = so, k = ko;
      }
      sample = wrnn_mol::logistic_sample(__shfl(mean, k, 32), __shfl(ls, k, 32), ul, wrnn_mol::kDefaultLogScaleMin);
      if (call.logits_out && sl == 0 && live_s)
        for (int c = ln; c < C; c += 32) call.logits_out[((size_t)bs * seq + t) * C + c] = lg(c);
    } else {
    const float uu = call.u[(size_t)bs * seq + t];
    float l[PER];  // the lane's classes are consecutive: ln * PER + i
    if constexpr (PER % 4 == 0) {
#pragma unroll
      for (int i = 0; i < PER; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(logits + ((size_t)((ln * PER + i) >> 2) * B + bs) * 4);
        l[i] = v[0], l[i + 1] = v[1], l[i + 2] = v[2], l[i + 3] = v[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int c = ln * PER + i;
        l[i] = logits[((size_t)(c >> 2) * B + bs) * 4 + (c & 3)];
      }
    }
    const int c = raw_draw<PER>(l, uu, ln, C);
    sample = 2.0f * (float)c / (float)(C - 1) - 1.0f;
    if (call.logits_out && sl == 0 && live_s) {
#pragma unroll
      for (int i = 0; i < PER; ++i) call.logits_out[((size_t)bs * seq + t) * C + ln * PER + i] = l[i];
    }
    }
    if (ln == 0) {
      xs[rs] = call.x_forced ? xf : sample;
      if (sl == 0 && live_s) {
        call.out[(size_t)bs * seq + t] = sample;
        tc_out[bs] = t + 1;
      }
    }
    __syncthreads();
    if (live_g && t + 1 < seq) {
      const float x = xs[rg];
#pragma unroll
      for (int j = 0; j < Gru1Row::kPer; ++j) {
        const int n4 = g1.group(sl, j);
        if (n4 < R / 4) g1.apply(gp[j], x, bg, n4);
      }
      if (sl == 0) {
#pragma unroll
        for (int j = 0; j < kMaxA; ++j) {
          const int n4 = (tid >> 3) + 32 * j;
          if (n4 < A4 / 4) *reinterpret_cast<f32x4*>(a_cur + ((size_t)n4 * B + bg) * 4) = av[j];
        }
      }
    }
  }
};

// Second GRU cell: its hidden-state half of the contraction was computed ahead (EpiRaw job below, riding in
// the previous step's sampler launch): tile rows 4 units x (r, z, gi_n, gh_n), `add` holds the four partial sums.
struct EpiGruAdd {
  const float* bias;  // tile order: b_ir + b_hr, b_iz + b_hz, b_in, b_hn
  Act h_prev, h_out, x_in, x_out, add;
  struct Pre { f32x4 b, a; float h, x; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.a = *reinterpret_cast<const f32x4*>(add.rowbase(row) + add.eoff(tile * 16 + u * 4));
    q.h = h_prev.rowbase(row)[h_prev.eoff(tile * 4 + u)];
    q.x = x_in.rowbase(row)[x_in.eoff(tile * 4 + u)];
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const float r = sigmoidf_(v[0] + q.a[0] + q.b[0]);
    const float z = sigmoidf_(v[1] + q.a[1] + q.b[1]);
    const float n = tanhf_(v[2] + q.a[2] + q.b[2] + r * (v[3] + q.a[3] + q.b[3]));
    const float h = (1.0f - z) * n + z * q.h;
    h_out.rowbase(row)[h_out.eoff(tile * 4 + u)] = h;
    x_out.rowbase(row)[x_out.eoff(tile * 4 + u)] = q.x + h;   // residual
  }
};

}  // namespace

constexpr int kGraphSteps = 64;  // steps per captured graph (even: the GRU states ping-pong)

struct wrnn_wavernn {
  int device = 0;
  int R = 0, F = 0, feat = 0, aux = 0, C = 0, Cp = 0, mode = 0;  // Cp: C rounded up to a multiple of 4 (logit buffer)
  // I: the mel / aux columns of the input layer (hoisted), its sample column is w_x.  First GRU cell: g1c = its input
  // half folded through I (hoisted), ug = W_ih . w_x, g1h = its hidden half.  Second cell: g2x / g2h.
  Layer I, g1c, g1h, g2x, g2h, f1, f2, f3;
  float *w_x = nullptr, *ug = nullptr;
  float* ws = nullptr;
  size_t ws_floats = 0;
  // kGraphSteps steps captured once per (workspace, B): every launch parameter of a step is call- and
  // step-independent (see SampleCall), so long utterances replay the graph instead of enqueueing
  // 5 launches per step from the host.
  hipStream_t cap = nullptr;
  hipGraphExec_t gexec = nullptr;
  const float* g_ws = nullptr;
  int g_B = 0;
  // unbatched generation as one persistent launch (wavernn_resident.h): raw row-major device copies of the loop's weights
  struct Resident {
    bool ok = false;   // the model has the kernel's shape and the device 256 compute units
    float *whh1 = nullptr, *b1hn = nullptr, *wih2 = nullptr, *whh2 = nullptr, *b2 = nullptr;
    float *fc1_w = nullptr, *fc1_b = nullptr, *fc2_w = nullptr, *fc2_b = nullptr, *fc3_w = nullptr, *fc3_b = nullptr;
    unsigned long long* gran = nullptr;
  } res;
  // fold-batched generation (2 .. 32 rows) as one persistent launch (wavernn_folds.h): the loop's own fragment-ordered layers,
  // plus the hand-over arrays
  struct Folds {
    bool ok = false;
    float* arr = nullptr;      // seven hand-over arrays (two slots x two row tiles each), then the error word
  } fold;
  int path = WRNN_WAVERNN_PATH_AUTO;
};

#include "wavernn_resident.h"
#include "wavernn_folds.h"

namespace {

struct StepBufs {  // the fixed (step-independent) part of the workspace
  SampleCall* call;
  int* tc[2];  // per-row step counters, read / written alternately
  float *h1, *h2[2], *x1, *x2, *y1, *y2, *lg, *a_cur;
  float* hp1;  // [3R/4][B][4]: W_hh . h1 + [0, 0, b_hn] of the first cell for the coming step
  float* hp2;  // [4R/4][B][4]: W_hh . h2 of the second cell (tile rows r, z, -, gh_n)
};

void drop_graph(wrnn_wavernn* h) {
  if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
  h->gexec = nullptr, h->g_ws = nullptr, h->g_B = 0;
}

Gru1Row gru1_row(const wrnn_wavernn* h, const StepBufs& f, int B) {
  return Gru1Row{h->w_x, h->ug, f.hp1, f.h1, f.x1, h->R, B};
}

// enqueue `n` consecutive steps starting at an even step.  On entry x1 holds the first cell's output of the
// first of them (written by the begin kernel or by the previous step's sampler).
int enqueue_steps(const wrnn_wavernn* h, hipStream_t st, int B, int n, const StepBufs& f) {
  const int R = h->R, F = h->F, A = h->aux, C = h->C, Cp = h->Cp;
  Act a_t[4];
  for (int i = 0; i < 4; ++i) a_t[i] = act_il(f.a_cur + (size_t)i * A * B, B, A);
  for (int t = 0; t < n; ++t) {
    const int a = t & 1, o = a ^ 1;
    LinJob<1, 1, EpiLinear> j1, j2, j3, jh1;
    LinJob<1, 1, EpiGruAdd> jg2;
    LinJob<1, 1, EpiRaw> jh2;
    const Act h2p = act_il(f.h2[a], B, R), h2n = act_il(f.h2[o], B, R);
    if (make_lin(jg2, h->g2x, Src{act_il(f.x1, B, R), R}, Src{a_t[1], A}, kNoSrc, B, B,
                 EpiGruAdd{h->g2x.bias, h2p, h2n, act_il(f.x1, B, R), act_il(f.x2, B, R), act_il(f.hp2, B, 4 * R)})) return 1;
    if (make_lin(j1, h->f1, Src{act_il(f.x2, B, R), R}, Src{a_t[2], A}, kNoSrc, B, B,
                 EpiLinear{h->f1.bias, act_il(f.y1, B, F), F, 1, nullptr})) return 1;
    if (make_lin(j2, h->f2, Src{act_il(f.y1, B, F), F}, Src{a_t[3], A}, kNoSrc, B, B,
                 EpiLinear{h->f2.bias, act_il(f.y2, B, F), F, 1, nullptr})) return 1;
    if (make_lin(j3, h->f3, Src{act_il(f.y2, B, F), F}, kNoSrc, kNoSrc, B, B,
                 EpiLinear{h->f3.bias, act_il(f.lg, B, Cp), Cp, 0, nullptr})) return 1;
    // the hidden-state halves of the NEXT step's cells ride in this step's fc1 (cell 1: h1 is this step's already)
    // and sampler (cell 2) launches; one riding job per launch, two would put two workgroups on every CU
    if (make_lin(jh1, h->g1h, Src{act_il(f.h1, B, R), R}, kNoSrc, kNoSrc, B, B,
                 EpiLinear{h->g1h.bias, act_il(f.hp1, B, 3 * R), 3 * R, 0, nullptr})) return 1;
    if (make_lin(jh2, h->g2h, Src{h2n, R}, kNoSrc, kNoSrc, B, B, EpiRaw{act_il(f.hp2, B, 4 * R)})) return 1;
    if (launch_tick(st, jg2) || launch_tick(st, j1, jh1) || launch_tick(st, j2) || launch_tick(st, j3)) return 1;
    const auto sample = [&](auto per) -> int {
      const SampleJob<decltype(per)::value> js{f.lg, C, B, 4 * A, f.call, f.tc[a], f.tc[o], f.a_cur, gru1_row(h, f, B)};
      return launch_tick(st, js, jh2);
    };
    int rc;
    if (h->mode == WRNN_WAVERNN_MOL) rc = sample(std::integral_constant<int, 0>());
    else switch (C / 32) {
      case 2: rc = sample(std::integral_constant<int, 2>()); break;
      case 4: rc = sample(std::integral_constant<int, 4>()); break;
      case 8: rc = sample(std::integral_constant<int, 8>()); break;
      case 16: rc = sample(std::integral_constant<int, 16>()); break;
      default: rc = sample(std::integral_constant<int, 32>()); break;
    }
    if (rc) return 1;
  }
  return 0;
}

}  // namespace

extern "C" int wrnn_wavernn_destroy(wrnn_wavernn* h) {
  if (!h) return 0;
  Guard g(h->device);
  (void)hipDeviceSynchronize();
  drop_graph(h);
  if (h->cap) (void)hipStreamDestroy(h->cap);
  for (Layer* L : {&h->I, &h->g1c, &h->g1h, &h->g2x, &h->g2h, &h->f1, &h->f2, &h->f3}) free_layer(*L);
  if (h->ws) (void)hipFree(h->ws);
  if (h->w_x) (void)hipFree(h->w_x);
  if (h->ug) (void)hipFree(h->ug);
  for (void* q : {(void*)h->res.whh1, (void*)h->res.b1hn, (void*)h->res.wih2, (void*)h->res.whh2, (void*)h->res.b2, (void*)h->res.fc1_w,
                  (void*)h->res.fc1_b, (void*)h->res.fc2_w, (void*)h->res.fc2_b, (void*)h->res.fc3_w, (void*)h->res.fc3_b, (void*)h->res.gran,
                  (void*)h->fold.arr})
    if (q) (void)hipFree(q);
  delete h;
  return 0;
}

extern "C" int wrnn_wavernn_create(const wrnn_wavernn_dims* dims, const wrnn_wavernn_weights* hw, int device,
                                   wrnn_wavernn** out) {
  WRNN_REQUIRE(dims && hw && out, "wrnn_wavernn_create: null argument");
  *out = nullptr;
  const int R = dims->rnn_dims, F = dims->fc_dims, feat = dims->feat_dims, aux = dims->aux_dims, C = dims->n_classes;
  WRNN_REQUIRE(R >= 4 && R % 4 == 0 && F >= 4 && F % 4 == 0, "rnn_dims=%d and fc_dims=%d must be positive multiples of 4", R, F);
  WRNN_REQUIRE(feat >= 4 && feat % 4 == 0 && aux >= 4 && aux % 4 == 0, "feat_dims=%d and aux_dims=%d must be positive multiples of 4", feat, aux);
  const int mode = dims->mode;
  WRNN_REQUIRE(mode == WRNN_WAVERNN_RAW || mode == WRNN_WAVERNN_MOL, "mode=%d must be WRNN_WAVERNN_RAW or WRNN_WAVERNN_MOL", mode);
  if (mode == WRNN_WAVERNN_RAW)
    WRNN_REQUIRE(C >= 64 && C <= 1024 && (C & (C - 1)) == 0, "RAW mode: n_classes=%d must be a power of two in [64, 1024]", C);
  else
    WRNN_REQUIRE(C >= 3 && C <= 96 && C % 3 == 0, "MOL mode: n_classes=%d must be 3 * nr_mix with nr_mix in [1, 32]", C);
  WRNN_REQUIRE(R <= 1024 && aux <= 128, "rnn_dims=%d must be <= 1024 and aux_dims=%d <= 128", R, aux);
  const float* const* wp = reinterpret_cast<const float* const*>(hw);
  for (size_t i = 0; i < sizeof(*hw) / sizeof(float*); ++i)
    WRNN_REQUIRE(wp[i] != nullptr, "wrnn_wavernn_create: weight pointer #%zu is null", i);
  int ndev = 0;
  WRNN_HIP(hipGetDeviceCount(&ndev));
  WRNN_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  WRNN_HIP(hipGetDeviceProperties(&prop, device));
  WRNN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libwrnn_mi355 is built for gfx950 only, device %d is %s",
               device, prop.gcnArchName);
  Guard g(device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", device);
  auto* h = new wrnn_wavernn();
  h->device = device, h->R = R, h->F = F, h->feat = feat, h->aux = aux, h->C = C, h->Cp = (C + 3) / 4 * 4, h->mode = mode;
  int rc = 0;
  auto bias_of = [](const float* b, int N) {
    return std::function<float(int, int)>([=](int t, int i) { return t * 16 + i < N ? b[t * 16 + i] : 0.0f; });
  };
  {  // I: columns [x | mel (feat) | a1 (aux)]: the mel / aux part as a layer, the x column as a vector
    const int Kin = 1 + feat + aux;
    const float* W = hw->I_w;
    RowFn w = [=](int t, int i, int k) -> float {
      const int n = t * 16 + i;
      return n < R ? W[(size_t)n * Kin + 1 + k] : 0.0f;
    };
    auto b = bias_of(hw->I_b, R);
    rc |= build_layer(h->I, (R + 15) / 16, feat + aux, R, w, &b);
    std::vector<float> wx(R);
    for (int n = 0; n < R; ++n) wx[n] = W[(size_t)n * Kin];
    rc |= upload(wx.data(), wx.size() * sizeof(float), (void**)&h->w_x);
  }
  // GRU cells, tile rows = 4 units x (r, z, gi_n, gh_n); torch rows r|z|n.  Two layers per cell: the input
  // columns (K = Kx, bias here) and the hidden-state columns (K = R, no bias), whose product only needs the
  // previous step's state and is computed ahead of the step.
  auto build_gru = [&](Layer& Lx, Layer& Lh, int Kx, const float* Wih, const float* Whh, const float* bih, const float* bhh) {
    RowFn wx = [=](int t, int i, int k) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      return gsel == 3 ? 0.0f : Wih[(size_t)(gsel * R + U) * Kx + k];
    };
    RowFn wh = [=](int t, int i, int k) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      return gsel == 2 ? 0.0f : Whh[(size_t)((gsel == 3 ? 2 : gsel) * R + U) * R + k];
    };
    std::function<float(int, int)> bb = [=](int t, int i) -> float {
      const int U = t * 4 + (i >> 2), gsel = i & 3;
      switch (gsel) {
        case 0: return bih[U] + bhh[U];
        case 1: return bih[R + U] + bhh[R + U];
        case 2: return bih[2 * R + U];
        default: return bhh[2 * R + U];
      }
    };
    return build_layer(Lx, R / 4, Kx, 4 * R, wx, &bb) | build_layer(Lh, R / 4, R, 4 * R, wh, nullptr);
  };
  {  // first cell (see Gru1Row): torch row order r | z | n, plain linear layouts
    const int Kin = 1 + feat + aux, Km = feat + aux;
    const float *Wih = hw->rnn1_w_ih, *Whh = hw->rnn1_w_hh, *bih = hw->rnn1_b_ih, *bhh = hw->rnn1_b_hh;
    const float *WI = hw->I_w, *bI = hw->I_b;
    std::vector<float> Wg((size_t)3 * R * Km), bg((size_t)3 * R), ugv((size_t)3 * R);
    std::vector<double> acc(Km);
    for (int n = 0; n < 3 * R; ++n) {
      std::fill(acc.begin(), acc.end(), 0.0);
      double ab = 0.0, au = 0.0;
      for (int j = 0; j < R; ++j) {
        const double wv = Wih[(size_t)n * R + j];
        const float* row = WI + (size_t)j * Kin;
        for (int k = 0; k < Km; ++k) acc[k] += wv * (double)row[1 + k];
        ab += wv * (double)bI[j];
        au += wv * (double)row[0];
      }
      for (int k = 0; k < Km; ++k) Wg[(size_t)n * Km + k] = (float)acc[k];
      bg[n] = (float)(ab + (double)bih[n] + (n < 2 * R ? (double)bhh[n] : 0.0));
      ugv[n] = (float)au;
    }
    const float* Wgp = Wg.data();
    RowFn wg = [=](int t, int i, int k) -> float { return t * 16 + i < 3 * R ? Wgp[(size_t)(t * 16 + i) * Km + k] : 0.0f; };
    auto bgf = bias_of(bg.data(), 3 * R);
    rc |= build_layer(h->g1c, (3 * R + 15) / 16, Km, 3 * R, wg, &bgf);
    RowFn wh = [=](int t, int i, int k) -> float { return t * 16 + i < 3 * R ? Whh[(size_t)(t * 16 + i) * R + k] : 0.0f; };
    std::function<float(int, int)> bh = [=](int t, int i) -> float {
      const int n = t * 16 + i;
      return n >= 2 * R && n < 3 * R ? bhh[n] : 0.0f;
    };
    rc |= build_layer(h->g1h, (3 * R + 15) / 16, R, 3 * R, wh, &bh);
    rc |= upload(ugv.data(), ugv.size() * sizeof(float), (void**)&h->ug);
  }
  rc |= build_gru(h->g2x, h->g2h, R + aux, hw->rnn2_w_ih, hw->rnn2_w_hh, hw->rnn2_b_ih, hw->rnn2_b_hh);
  auto build_lin = [&](Layer& L, const float* W, const float* bvec, int K, int N) {
    RowFn w = [=](int t, int i, int k) -> float { return t * 16 + i < N ? W[(size_t)(t * 16 + i) * K + k] : 0.0f; };
    auto b = bias_of(bvec, N);
    return build_layer(L, (N + 15) / 16, K, N, w, &b);
  };
  rc |= build_lin(h->f1, hw->fc1_w, hw->fc1_b, R + aux, F);
  rc |= build_lin(h->f2, hw->fc2_w, hw->fc2_b, F + aux, F);
  rc |= build_lin(h->f3, hw->fc3_w, hw->fc3_b, F, C);
  if (!rc && hipStreamCreateWithFlags(&h->cap, hipStreamNonBlocking) != hipSuccess) {
    rc = wrnn::fail("wrnn_wavernn_create: cannot create the capture stream");
  }
  // persistent kernel for unbatched generation: built for rnn_dims = fc_dims = 512 (the reference configuration,
  // tests/resources/test_config.yaml:47-75), one compute unit per workgroup
  const bool raw_ok = mode == WRNN_WAVERNN_RAW && C <= kWvR && (C == 64 || C == 128 || C == 256 || C == 512);
  if (!rc && R == kWvR && F == kWvR && aux <= kWvMaxA && (raw_ok || mode == WRNN_WAVERNN_MOL) && prop.multiProcessorCount >= kRW) {
    auto& Q = h->res;
    std::vector<float> b2v((size_t)4 * R);
    for (int U = 0; U < R; ++U) {
      b2v[U] = hw->rnn2_b_ih[U] + hw->rnn2_b_hh[U];
      b2v[R + U] = hw->rnn2_b_ih[R + U] + hw->rnn2_b_hh[R + U];
      b2v[2 * R + U] = hw->rnn2_b_ih[2 * R + U];
      b2v[3 * R + U] = hw->rnn2_b_hh[2 * R + U];
    }
    const size_t f = sizeof(float);
    rc |= upload(hw->rnn1_w_hh, (size_t)3 * R * R * f, (void**)&Q.whh1) | upload(hw->rnn1_b_hh + 2 * R, (size_t)R * f, (void**)&Q.b1hn) |
          upload(hw->rnn2_w_ih, (size_t)3 * R * (R + aux) * f, (void**)&Q.wih2) | upload(hw->rnn2_w_hh, (size_t)3 * R * R * f, (void**)&Q.whh2) |
          upload(b2v.data(), b2v.size() * f, (void**)&Q.b2) | upload(hw->fc1_w, (size_t)F * (R + aux) * f, (void**)&Q.fc1_w) |
          upload(hw->fc1_b, (size_t)F * f, (void**)&Q.fc1_b) | upload(hw->fc2_w, (size_t)F * (F + aux) * f, (void**)&Q.fc2_w) |
          upload(hw->fc2_b, (size_t)F * f, (void**)&Q.fc2_b) | upload(hw->fc3_w, (size_t)C * F * f, (void**)&Q.fc3_w) |
          upload(hw->fc3_b, (size_t)C * f, (void**)&Q.fc3_b);
    if (!rc && hipMalloc((void**)&Q.gran, (7 * kWvArr + 2 + 128) * sizeof(u64)) != hipSuccess)   // arrays, status, XCD numbers
      rc = wrnn::fail("wrnn_wavernn_create: cannot allocate the hand-over arrays");
    Q.ok = !rc;
  }
  // persistent kernel for 2 .. 32 rows (wavernn_folds.h): the same shape limits, 128 compute units per 16 rows
  if (!rc && R == kWfR && F == kWfR && aux <= 64 && (raw_ok || mode == WRNN_WAVERNN_MOL) && prop.multiProcessorCount >= 2 * kWfPer) {
    if (hipMalloc((void**)&h->fold.arr, wf_array_floats(C) * sizeof(float)) != hipSuccess)
      rc = wrnn::fail("wrnn_wavernn_create: cannot allocate the hand-over arrays");
    h->fold.ok = !rc;
  }
  if (rc) {
    std::string keep = wrnn::last_error();
    wrnn_wavernn_destroy(h);
    wrnn::last_error() = keep;
    return 1;
  }
  *out = h;
  return 0;
}

namespace {

constexpr int kWvNoLaunch = 2;   // the cooperative launch was refused (nothing enqueued)

template <int PER>
int launch_wv_resident(const WvResParams& prm, hipStream_t st) {
  auto* fn = wavernn_resident_kernel<PER>;
  // all 224 workgroups must be resident together (they wait for each other): checked by the cooperative launch; a plain
  // launch of the same grid (WRNN_RESIDENT_PLAIN_LAUNCH, for rocprofv3's kernel trace) has the same residency
  int per_cu = 0;
  WRNN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), kRT, 0));
  WRNN_REQUIRE(per_cu >= 1, "wavernn persistent kernel does not fit a compute unit");
  // queued on the device's stream for persistent grids (common.h): it must not share the device with another such grid
  return wrnn::on_persistent_stream(st, [&](hipStream_t ps) -> int {
    if (getenv("WRNN_RESIDENT_PLAIN_LAUNCH")) {
      hipLaunchKernelGGL(fn, dim3(kWvGrid), dim3(kRT), 0, ps, prm);
      WRNN_LAUNCH_CHECK();
      return 0;
    }
    void* args[] = {const_cast<WvResParams*>(&prm)};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(kWvGrid), dim3(kRT), args, 0, ps);
    if (e != hipSuccess) {   // e.g. another process holds compute units: nothing has run, the caller may take the launch loop
      (void)hipGetLastError();
      wrnn::fail("hipLaunchCooperativeKernel(wavernn_resident_kernel) failed: %s", hipGetErrorString(e));
      return kWvNoLaunch;
    }
    return 0;
  });
}

// one row, the whole call in one launch; returns when the samples are there (the launch's status word is read back)
int run_resident(wrnn_wavernn* h, const float* cI, const float* G, const float* aux, const float* uniforms, const float* x_forced,
                 float* out, float* logits_out, int seq, hipStream_t st) {
  auto& Q = h->res;
  constexpr size_t arr = kWvArr;
  WRNN_HIP(hipMemsetAsync(Q.gran, 0, (7 * arr + 2 + 128) * sizeof(u64), st));
  WvResParams q{};
  q.whh1 = Q.whh1, q.b1hn = Q.b1hn, q.wih2 = Q.wih2, q.whh2 = Q.whh2, q.b2 = Q.b2;
  q.fc1_w = Q.fc1_w, q.fc1_b = Q.fc1_b, q.fc2_w = Q.fc2_w, q.fc2_b = Q.fc2_b, q.fc3_w = Q.fc3_w, q.fc3_b = Q.fc3_b;
  q.w_x = h->w_x, q.ug = h->ug;
  q.cI = cI, q.G = G, q.aux = aux, q.u = uniforms, q.x_forced = x_forced, q.out = out, q.logits_out = logits_out;
  q.g_lg = Q.gran, q.g_s1 = Q.gran + arr, q.g_s2 = Q.gran + 3 * arr, q.g_y1 = Q.gran + 5 * arr, q.g_y2 = Q.gran + 6 * arr;
  q.err = reinterpret_cast<u32*>(Q.gran + 7 * arr), q.xcc = reinterpret_cast<u32*>(Q.gran + 7 * arr + 2);
  q.map = getenv("WRNN_RESIDENT_LINEAR_MAP") ? 0 : 1, q.plain = getenv("WRNN_RESIDENT_WT") ? 0 : 1;
  q.seq = seq, q.A = h->aux, q.C = h->C;
  const char* stamp_file = getenv("WRNN_RESIDENT_STAMPS");
  u64* stamps = nullptr;
  if (stamp_file && seq >= kWvStampStep0 + 8) {
    WRNN_HIP(hipMalloc((void**)&stamps, 5 * 8 * 4 * sizeof(u64)));
    WRNN_HIP(hipMemsetAsync(stamps, 0, 5 * 8 * 4 * sizeof(u64), st));
  }
  q.stamps = stamps;
  int rc;
  if (h->mode == WRNN_WAVERNN_MOL) rc = launch_wv_resident<0>(q, st);
  else switch (h->C / 32) {
    case 2: rc = launch_wv_resident<2>(q, st); break;
    case 4: rc = launch_wv_resident<4>(q, st); break;
    case 8: rc = launch_wv_resident<8>(q, st); break;
    default: rc = launch_wv_resident<16>(q, st); break;
  }
  if (rc) {
    if (stamps) (void)hipFree(stamps);
    return rc;
  }
  u64 status = 0;
  WRNN_HIP(hipMemcpyAsync(&status, Q.gran + 7 * arr, sizeof(status), hipMemcpyDeviceToHost, st));
  WRNN_HIP(hipStreamSynchronize(st));
  if (stamps) {  // diagnostic dump: per group (first workgroup) and step: sweep start, swept, published, in us from C1's step start
    u64 hs[5 * 8 * 4];
    WRNN_HIP(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    (void)hipFree(stamps);
    if (FILE* fo = fopen(stamp_file, "w")) {
      static const char* names[5] = {"C1", "C2", "F1", "F2", "F3"};
      for (int rw = 0; rw < 8; ++rw) {
        fprintf(fo, "step%d:", kWvStampStep0 + rw);
        for (int gi = 0; gi < 5; ++gi) {
          fprintf(fo, "  %s", names[gi]);
          for (int i = 0; i < 3; ++i) {
            const u64 v = hs[(gi * 8 + rw) * 4 + i];
            fprintf(fo, " %6.2f", v ? (double)(long long)(v - hs[rw * 4]) * 0.01 : -1.0);
          }
        }
        fprintf(fo, "\n");
      }
      fclose(fo);
    }
  }
  WRNN_REQUIRE((u32)status == 0, "wavernn persistent kernel: a stage hand-over timed out (the workgroups were not co-resident?)");
  return 0;
}

// ---- 2 .. 32 rows, the whole call in one launch (wavernn_folds.h)

constexpr int kWfNoLaunch = 2;   // the cooperative launch was refused (nothing enqueued)

template <int PER>
int launch_wf(const WfParams& prm, int grid, hipStream_t st) {
  auto* fn = wavernn_folds_kernel<PER>;
  WRNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWfLdsBytes));
  int per_cu = 0;
  WRNN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), kWfT, kWfLdsBytes));
  WRNN_REQUIRE(per_cu >= 1, "wavernn fold kernel does not fit a compute unit");
  return wrnn::on_persistent_stream(st, [&](hipStream_t ps) -> int {
    if (getenv("WRNN_RESIDENT_PLAIN_LAUNCH")) {   // for rocprofv3's kernel trace (it crashes on cooperative launches)
      hipLaunchKernelGGL(fn, dim3(grid), dim3(kWfT), kWfLdsBytes, ps, prm);
      WRNN_LAUNCH_CHECK();
      return 0;
    }
    void* args[] = {const_cast<WfParams*>(&prm)};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3(grid), dim3(kWfT), args, (unsigned)kWfLdsBytes, ps);
    if (e != hipSuccess) {   // e.g. another process holds compute units: nothing has run, the caller may take the launch loop
      (void)hipGetLastError();
      wrnn::fail("hipLaunchCooperativeKernel(wavernn_folds_kernel) failed: %s", hipGetErrorString(e));
      return kWfNoLaunch;
    }
    return 0;
  });
}

int launch_wf_mode(const wrnn_wavernn* h, const WfParams& q, int grid, hipStream_t st) {
  if (h->mode == WRNN_WAVERNN_MOL) return launch_wf<0>(q, grid, st);
  switch (h->C / 32) {
    case 2: return launch_wf<2>(q, grid, st);
    case 4: return launch_wf<4>(q, grid, st);
    case 8: return launch_wf<8>(q, grid, st);
    default: return launch_wf<16>(q, grid, st);
  }
}

// returns when the samples are there (the launch's status word is read back)
int run_folds(wrnn_wavernn* h, const float* cI, const float* G, const float* aux, const float* uniforms, const float* x_forced,
              float* out, float* logits_out, int B, int seq, hipStream_t st) {
  const int C = h->C, nf3 = (C + 15) / 16, MT = (B + 15) / 16;
  const size_t total = wf_array_floats(C);
  WRNN_HIP(hipMemsetAsync(h->fold.arr, 0xFF, (total - kWfTail) * sizeof(float), st));   // every word: tag 1, steps 0 and 1 carry tag 0
  WRNN_HIP(hipMemsetAsync(h->fold.arr + total - kWfTail, 0, kWfTail * sizeof(float), st));
  WfParams q{};
  q.w_g1h = h->g1h.wt, q.w_g2x = h->g2x.wt, q.w_g2h = h->g2h.wt, q.w_f1 = h->f1.wt, q.w_f2 = h->f2.wt, q.w_f3 = h->f3.wt;
  q.b_g1h = h->g1h.bias, q.b_g2 = h->g2x.bias, q.b_f1 = h->f1.bias, q.b_f2 = h->f2.bias, q.b_f3 = h->f3.bias;
  q.w_x = h->w_x, q.ug = h->ug;
  q.cI = cI, q.G = G, q.aux = aux, q.u = uniforms, q.x_forced = x_forced, q.out = out, q.logits_out = logits_out;
  f32x4* a = reinterpret_cast<f32x4*>(h->fold.arr);
  const size_t s512 = kWfArr512 / 4;
  q.X1 = a, q.H1 = a + s512, q.X2 = a + 2 * s512, q.H2 = a + 3 * s512, q.Y1 = a + 4 * s512, q.Y2 = a + 5 * s512, q.HS = a + 6 * s512, q.LG = a + 9 * s512;
  q.err = reinterpret_cast<unsigned*>(h->fold.arr + total - kWfTail), q.xcc = q.err + 64;
  q.xs = reinterpret_cast<u64*>(q.err + 320);   // 64 pairs behind the XCD numbers (the tail is zeroed above)
  q.B = B, q.MT = MT, q.seq = seq, q.A = h->aux, q.C = C, q.nf3 = nf3, q.adapt = getenv("WRNN_FOLDS_NO_ADAPT") ? 0 : 1, q.map = getenv("WRNN_FOLDS_LINEAR_MAP") ? 0 : 1;
  q.plain = getenv("WRNN_FOLDS_WT") ? 0 : 1;   // (plain stores inside a group only where the group is seen to share an XCD)
  q.cool = getenv("WRNN_FOLDS_COOL") ? atoi(getenv("WRNN_FOLDS_COOL")) : 2;
  const char* stamp_file = getenv("WRNN_RESIDENT_STAMPS");
  u64* stamps = nullptr;
  if (stamp_file && seq >= kWfStampStep0 + 8) {
    WRNN_HIP(hipMalloc((void**)&stamps, 5 * 8 * 4 * sizeof(u64)));
    WRNN_HIP(hipMemsetAsync(stamps, 0, 5 * 8 * 4 * sizeof(u64), st));
  }
  q.stamps = stamps, q.stamp_j = getenv("WRNN_FOLDS_STAMP_J") ? atoi(getenv("WRNN_FOLDS_STAMP_J")) & 31 : 0;
  if (const int rc = launch_wf_mode(h, q, 2 * kWfPer, st)) {
    if (stamps) (void)hipFree(stamps);
    return rc;
  }
  unsigned status = 0;
  WRNN_HIP(hipMemcpyAsync(&status, q.err, sizeof(status), hipMemcpyDeviceToHost, st));
  WRNN_HIP(hipStreamSynchronize(st));
  if (stamps) {  // diagnostic dump: per group (first workgroup) and step, four points in us from C1's step start
    u64 hs[5 * 8 * 4];
    WRNN_HIP(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
    (void)hipFree(stamps);
    if (FILE* fo = fopen(stamp_file, "w")) {
      static const char* names[5] = {"C1", "C2", "F1", "F2", "F3"};
      for (int rw = 0; rw < 8; ++rw) {
        fprintf(fo, "step%d:", kWfStampStep0 + rw);
        for (int gi = 0; gi < 5; ++gi) {
          fprintf(fo, "  %s", names[gi]);
          for (int i = 0; i < 4; ++i) {
            const u64 v = hs[(gi * 8 + rw) * 4 + i];
            fprintf(fo, " %6.2f", v ? (double)(long long)(v - hs[rw * 4]) * 0.01 : -1.0);
          }
        }
        fprintf(fo, "\n");
      }
      fclose(fo);
    }
  }
  WRNN_REQUIRE(status == 0, "wavernn fold kernel: a stage hand-over timed out (the workgroups were not co-resident?)");
  return 0;
}

}  // namespace

extern "C" int wrnn_wavernn_set_path(wrnn_wavernn* h, int path) {
  WRNN_REQUIRE(h != nullptr, "null wavernn handle");
  WRNN_REQUIRE(path == WRNN_WAVERNN_PATH_AUTO || path == WRNN_WAVERNN_PATH_LAUNCHES || path == WRNN_WAVERNN_PATH_RESIDENT,
               "wrnn_wavernn_set_path: unknown path %d", path);
  h->path = path;
  return 0;
}

extern "C" int wrnn_wavernn_generate(wrnn_wavernn* h, const float* mels, const float* aux, const float* uniforms, int B,
                                     int seq, const float* x_forced, float* out, float* logits_out, void* stream) {
  WRNN_REQUIRE(h != nullptr, "null wavernn handle");
  WRNN_REQUIRE(B >= 0 && seq >= 0, "wrnn_wavernn_generate: negative size (B=%d, seq=%d)", B, seq);
  WRNN_REQUIRE(B <= 65535, "wrnn_wavernn_generate: B=%d out of range", B);
  if (B == 0 || seq == 0) return 0;
  WRNN_REQUIRE(mels && aux && uniforms && out, "wrnn_wavernn_generate: null tensor");
  Guard g(h->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", h->device);
  // AUTO, 33 .. 64 rows: two calls of the persistent kernel one after the other (measured at 64 rows: 2 x 10.1 us per step
  // against 25.1 for the launch loop; from about 96 rows on the launch loop's larger batch wins: 34 us at 128 rows)
  if (h->path == WRNN_WAVERNN_PATH_AUTO && h->fold.ok && B > kWfMaxRows && B <= 2 * kWfMaxRows) {
    const int b0 = (B + 1) / 2;
    const size_t us = h->mode == WRNN_WAVERNN_MOL ? (size_t)(h->C / 3 + 1) : 1;
    for (int part = 0; part < 2; ++part) {
      const size_t o = part ? (size_t)b0 * seq : 0;
      if (wrnn_wavernn_generate(h, mels + o * h->feat, aux + o * 4 * h->aux, uniforms + o * us, part ? B - b0 : b0, seq,
                                x_forced ? x_forced + o : nullptr, out + o, logits_out ? logits_out + o * h->C : nullptr, stream))
        return 1;
    }
    return 0;
  }
  hipStream_t st = (hipStream_t)stream;
  const int R = h->R, F = h->F, feat = h->feat, A = h->aux, C = h->Cp;
  const size_t rows = (size_t)seq * B;
  WRNN_REQUIRE(rows <= (size_t)1 << 30, "wrnn_wavernn_generate: seq*B = %zu too large", rows);
  const size_t fixed = 64 + 2 * (size_t)((B + 3) / 4 * 4) + (size_t)B * (12 * R + 2 * F + C + 4 * A);
  const size_t need = fixed + rows * (feat + 4 * A + 4 * (size_t)R);
  if (need > h->ws_floats) {
    if (h->ws) {
      WRNN_HIP(hipDeviceSynchronize());
      drop_graph(h);
      WRNN_HIP(hipFree(h->ws));
      h->ws = nullptr, h->ws_floats = 0;
    }
    WRNN_HIP(hipMalloc((void**)&h->ws, need * sizeof(float)));
    h->ws_floats = need;
  }
  float* w = h->ws;
  auto take = [&](size_t n) { float* p = w; w += n; return p; };
  StepBufs f;
  static_assert(sizeof(SampleCall) <= 64 * sizeof(float), "call block");
  f.call = reinterpret_cast<SampleCall*>(take(64));
  for (int i = 0; i < 2; ++i) f.tc[i] = reinterpret_cast<int*>(take((size_t)((B + 3) / 4 * 4)));
  f.h1 = take((size_t)B * R), f.h2[0] = take((size_t)B * R), f.h2[1] = take((size_t)B * R);
  f.x1 = take((size_t)B * R), f.x2 = take((size_t)B * R);
  f.y1 = take((size_t)B * F), f.y2 = take((size_t)B * F), f.lg = take((size_t)B * C);
  f.a_cur = take((size_t)B * 4 * A);
  f.hp1 = take((size_t)B * 3 * R), f.hp2 = take((size_t)B * 4 * R);
  float* mp = take(rows * feat);     // [seq][feat/4][B][4]
  float* ap = take(rows * 4 * A);    // [seq][4A/4][B][4]: slice i of step t at + t*B*4A + i*A*B
  float* cI = take(rows * R);        // [seq][R/4][B][4]: the input layer without its sample column, every step
  float* G = take(rows * 3 * R);     // [seq][3R/4][B][4]: the first cell's input half without its sample term
  auto pack = [&](const float* src, int ld, int K, float* dst) -> int {
    const size_t total = rows * (K / 4);
    hipLaunchKernelGGL(wavernn_pack_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, st,
                       src, ld, 0, K, dst, B, seq);
    WRNN_LAUNCH_CHECK();
    return 0;
  };
  if (pack(mels, feat, feat, mp) || pack(aux, 4 * A, 4 * A, ap)) return 1;
  {  // hoisted: the input layer and the first cell's input half, one GEMM each over all seq*B rows
    LinJob<2, 4, EpiLinear, 2> jc, jg;
    Act a1 = act_il(ap, B, A);
    a1.ss = 4L * A * B;
    if (make_lin(jc, h->I, Src{act_il(mp, B, feat), feat}, Src{a1, A}, kNoSrc, (int)rows, B,
                 EpiLinear{h->I.bias, act_il(cI, B, R), R, 0, nullptr})) return 1;
    if (make_lin(jg, h->g1c, Src{act_il(mp, B, feat), feat}, Src{a1, A}, kNoSrc, (int)rows, B,
                 EpiLinear{h->g1c.bias, act_il(G, B, 3 * R), 3 * R, 0, nullptr})) return 1;
    if (launch_tick(st, jc) || launch_tick(st, jg)) return 1;
  }
  const bool auto_res = h->path == WRNN_WAVERNN_PATH_AUTO && ((B == 1 && h->res.ok) || (B >= 2 && B <= kWfMaxRows && h->fold.ok));
  if (h->path == WRNN_WAVERNN_PATH_RESIDENT || auto_res) {
    WRNN_REQUIRE(h->res.ok && h->fold.ok, "wrnn_wavernn_generate: the persistent kernels need rnn_dims = fc_dims = 512, aux_dims <= 64, at most 512 "
                                          "classes and 256 compute units");
    WRNN_REQUIRE(B <= kWfMaxRows, "wrnn_wavernn_generate: the persistent kernels generate at most %d rows per call (B=%d)", kWfMaxRows, B);
    const int rc = B == 1 ? run_resident(h, cI, G, ap, uniforms, x_forced, out, logits_out, seq, st) : run_folds(h, cI, G, ap, uniforms, x_forced, out, logits_out, B, seq, st);
    // AUTO: a refused cooperative launch (the device is shared with another process) is not an error, the launch loop runs
    static_assert(kWfNoLaunch == kWvNoLaunch, "one code for a refused launch");
    if (!(rc == kWfNoLaunch && auto_res && h->path == WRNN_WAVERNN_PATH_AUTO)) return rc ? 1 : 0;
    wrnn::last_error().clear();
  }
  WRNN_HIP(hipMemsetAsync(f.h1, 0, (size_t)B * 3 * R * sizeof(float), st));   // h1, h2[0], h2[1]: zero state
  WRNN_HIP(hipMemsetAsync(f.hp2, 0, (size_t)B * 4 * R * sizeof(float), st));  // W_hh . 0
  {  // hp1 = W_hh . 0 + [0, 0, b_hn], then step 0's first cell in the begin kernel
    LinJob<1, 1, EpiLinear> jh1;
    if (make_lin(jh1, h->g1h, Src{act_il(f.h1, B, R), R}, kNoSrc, kNoSrc, B, B,
                 EpiLinear{h->g1h.bias, act_il(f.hp1, B, 3 * R), 3 * R, 0, nullptr})) return 1;
    if (launch_tick(st, jh1)) return 1;
  }
  const SampleCall call{uniforms, x_forced, out, logits_out, cI, G, ap, seq};
  {
    const Gru1Row g1 = gru1_row(h, f, B);
    hipLaunchKernelGGL(wavernn_begin_kernel, dim3((B + Gru1Row::kRows - 1) / Gru1Row::kRows * g1.slices()), dim3(256), 0, st,
                       call, f.call, f.tc[0], B, 4 * A, f.a_cur, g1);
  }
  WRNN_LAUNCH_CHECK();
  int done = 0;
  if (seq >= 4 * kGraphSteps) {
    if (!h->gexec || h->g_ws != h->ws || h->g_B != B) {
      drop_graph(h);
      WRNN_HIP(hipStreamBeginCapture(h->cap, hipStreamCaptureModeRelaxed));
      const int rc = enqueue_steps(h, h->cap, B, kGraphSteps, f);
      hipGraph_t graph = nullptr;
      const hipError_t e = hipStreamEndCapture(h->cap, &graph);
      if (rc || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc ? 1 : wrnn::fail("wrnn_wavernn_generate: graph capture failed: %s", hipGetErrorString(e));
      }
      const hipError_t ei = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) {
        h->gexec = nullptr;
        return wrnn::fail("wrnn_wavernn_generate: graph instantiate failed: %s", hipGetErrorString(ei));
      }
      h->g_ws = h->ws, h->g_B = B;
    }
    for (; done + kGraphSteps <= seq; done += kGraphSteps) WRNN_HIP(hipGraphLaunch(h->gexec, st));
  }
  return enqueue_steps(h, st, B, seq - done, f);
}
