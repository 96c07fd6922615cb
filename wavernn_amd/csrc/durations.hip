// Attention -> durations for gfx950: the monotonic shortest path of
// duration_extraction/duration_extractor.py:23-84 as a dynamic programme, one workgroup per utterance.
//
// The reference builds an S*T-node graph (edges right / down / down-right, weight = 1 - attention at
// the target, :94-130) in Python and runs scipy's Dijkstra from the top-left node.  On that DAG the
// distances are the fixed point of  D[i][j] = w[i][j] + min(D[i][j-1], D[i-1][j], D[i-1][j-1]),  which
// is evaluated here along anti-diagonals in float64 (scipy's dtype), so the path cost is bit-identical.
// Where several predecessors tie, scipy's choice depends on its heap's scan order; this kernel prefers
// down-right, then down, then right (the order that reproduces the reference's own known answers,
// tests/test_duration_extractor.py).  Integer / byte work, L2-resident: no MFMA, no HBM streaming.
#include "common.h"
#include "../../include/wrnn_mi355.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWinBytes = 16 * 1024;  // back-pointer window staged in LDS for the backtrack (reused for the final reduction)
constexpr int kMaxS = 8192, kMaxT = 1024;

struct DurArgs {
  const float* att;   // [B][att_rows][att_ld]
  const float* mel;   // [B][n_mels][mel_ld]
  const int32_t* tok; // [B][tok_ld]
  const int32_t *mel_len, *x_len;
  const int32_t* sil_idx;  // silent token ids
  int n_sil;
  int att_rows, att_ld, n_mels, mel_ld, tok_ld, T_out, S_max, T_max;
  float thr, shift;
  unsigned char* bp;  // [B][att_rows][att_ld] back pointers
  float* dur;         // [B][T_out]
  double *score, *cost;
};

__device__ __forceinline__ float shifted_att(float a, bool sil_row, bool sil_tok, float shift) {
  if (sil_row) a = a + (sil_tok ? shift : -shift);  // attention[i,:] + (isin*shift*2 - shift), fp32 (:49-50)
  return fminf(fmaxf(a, 0.0f), 1.0f);               // torch.clamp(attention, 0, 1) (:52)
}

__global__ __launch_bounds__(kThreads) void durations_kernel(DurArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  // lengths are device data: clamp them to what the launch sized its LDS and the tensors for
  const int S = min(max(A.mel_len[b], 1), A.S_max), T = min(max(A.x_len[b], 1), A.T_max);
  // LDS carve-up
  double* dbuf = reinterpret_cast<double*>(smem);                   // 3 x [T]
  int* durs = reinterpret_cast<int*>(dbuf + 3 * (size_t)T);          // [T]
  unsigned char* sil = reinterpret_cast<unsigned char*>(durs + T);   // [S]
  unsigned char* stok = sil + S;                                     // [T]
  unsigned short* jlo = reinterpret_cast<unsigned short*>(stok + T + ((S + T) & 1));  // [S] first / last column
  unsigned short* jhi = jlo + S;                                     // [S] of the path in every row
  unsigned char* win = reinterpret_cast<unsigned char*>(jhi + S);    // [kWinBytes]
  win += (8 - (reinterpret_cast<size_t>(win) & 7)) & 7;
  __shared__ int sh_state[4];  // i, j, done, silent-frame count
  const float* att = A.att + (size_t)b * A.att_rows * A.att_ld;
  const float* mel = A.mel + (size_t)b * A.n_mels * A.mel_ld;
  unsigned char* bp = A.bp + (size_t)b * A.att_rows * A.att_ld;

  if (tid == 0) sh_state[3] = 0;
  __syncthreads();
  // silent frames: mel.mean(dim=0) < threshold (:42); silent tokens: isin(x, silent ids) (:48)
  int my_sil = 0;
  for (int i = tid; i < S; i += kThreads) {
    double acc = 0.0;
    for (int m = 0; m < A.n_mels; ++m) acc += (double)mel[(size_t)m * A.mel_ld + i];
    const bool s = (float)(acc / A.n_mels) < A.thr;
    sil[i] = s;
    my_sil += s;
    jlo[i] = jhi[i] = 0;  // every row is visited by a valid walk; defined even if the guard cuts one short
  }
  if (my_sil) atomicAdd(&sh_state[3], my_sil);
  for (int j = tid; j < T; j += kThreads) {
    const int t = A.tok[(size_t)b * A.tok_ld + j];
    bool s = false;
    for (int q = 0; q < A.n_sil; ++q) s |= (t == A.sil_idx[q]);
    stok[j] = s;
    durs[j] = 0;
  }
  __syncthreads();
  // `.nonzero().squeeze()` of exactly one silent frame is 0-dim and the reference then applies no
  // shift at all (:43-44); the silence mask still gates the score (:76)
  const bool do_shift = sh_state[3] >= 2;

  // ---- forward DP over anti-diagonals d = i + j
  double *cur = dbuf, *p1 = dbuf + T, *p2 = dbuf + 2 * (size_t)T;
  const double INF = __longlong_as_double(0x7ff0000000000000LL);
  // the attention value of this thread's first column is fetched one diagonal ahead (clamped index,
  // unconditional load): the L2 round trip overlaps the previous diagonal instead of heading each one
  auto att_at = [&](int d, int j) {
    const int i = min(max(d - j, 0), S - 1);
    return att[(size_t)i * A.att_ld + min(j, T - 1)];
  };
  float a_next = att_at(0, tid);
  for (int d = 0; d < S + T - 1; ++d) {
    const float a_cur = a_next;
    a_next = att_at(d + 1, tid);
    for (int j = tid; j < T; j += kThreads) {
      const int i = d - j;
      if (i < 0 || i >= S) continue;
      const float a = shifted_att(j == tid ? a_cur : att[(size_t)i * A.att_ld + j], do_shift && sil[i], stok[j], A.shift);
      const double w = (double)(1.0f - a);  // path_probs = 1. - attention in fp32 (:53), float64 inside scipy
      double best;
      unsigned char code;
      if (d == 0) {
        best = 0.0, code = 3;  // source node: distance 0, its own weight is never paid
        cur[j] = best;
        bp[0] = code;
        continue;
      }
      best = (i > 0 && j > 0) ? p2[j - 1] : INF, code = 2;          // down-right from (i-1, j-1)
      const double up = i > 0 ? p1[j] : INF;                         // down from (i-1, j)
      if (up < best) best = up, code = 1;
      const double left = j > 0 ? p1[j - 1] : INF;                   // right from (i, j-1)
      if (left < best) best = left, code = 0;
      cur[j] = best + w;
      bp[(size_t)i * A.att_ld + j] = code;
    }
    // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetch and the
    // back-pointer stores of every diagonal (the stores are fenced once, after the loop)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    double* t = p2;
    p2 = p1, p1 = cur, cur = t;
  }
  // after the last rotation p1 holds the last diagonal: cell (S-1, T-1) at index T-1
  if (tid == 0) {
    if (A.cost) A.cost[b] = p1[T - 1];
    sh_state[0] = S - 1, sh_state[1] = T - 1, sh_state[2] = 0;
  }
  __threadfence_block();
  __syncthreads();

  // ---- backtrack: windows of R rows of back pointers staged in LDS, walked by one lane that touches
  //      LDS only: it records the column range [jlo, jhi] of the path in every row
  const int R = max(kWinBytes / T, 1);
  int last_row = -1;
  while (true) {
    const int i_hi = sh_state[0];
    if (sh_state[2]) break;
    const int lo = max(i_hi - R + 1, 0);
    const int n = (i_hi - lo + 1) * T;
    for (int e = tid; e < n; e += kThreads) {
      const int r = e / T, c = e - r * T;
      win[e] = bp[(size_t)(lo + r) * A.att_ld + c];
    }
    __syncthreads();
    if (tid == 0) {
      int i = i_hi, j = sh_state[1];
      bool done = false;
      int guard = S + T;  // a valid walk takes fewer steps; never spin on corrupted back pointers
      while (i >= lo && j >= 0 && guard-- > 0) {
        if (i != last_row) {  // first visit in reverse = last node of the row in path order (:73-80)
          jhi[i] = (unsigned short)j;
          last_row = i;
        }
        jlo[i] = (unsigned short)j;
        const unsigned char code = win[(i - lo) * T + j];
        if (code == 3) {
          done = true;
          break;
        }
        if (code == 2) --i, --j;
        else if (code == 1) --i;
        else --j;
      }
      sh_state[0] = i, sh_state[1] = j, sh_state[2] = done || j < 0 || i < 0 || guard <= 0;
    }
    __syncthreads();
  }
  // ---- durations: one frame for the last column of every row (:73-80); score: mean clamped attention over
  //      every path node of the voiced rows (:76-82), rows in parallel, fixed-order tree reduction in float64
  double sum = 0.0;
  int cnt = 0;
  for (int i = tid; i < S; i += kThreads) {
    atomicAdd(&durs[jhi[i]], 1);
    if (!sil[i])
      for (int j = jlo[i]; j <= jhi[i]; ++j) {
        sum += (double)shifted_att(att[(size_t)i * A.att_ld + j], false, stok[j], A.shift);
        ++cnt;
      }
  }
  double* rs = reinterpret_cast<double*>(win);            // [kThreads]
  int* rc = reinterpret_cast<int*>(rs + kThreads);        // [kThreads]
  rs[tid] = sum, rc[tid] = cnt;
  __syncthreads();
  for (int off = kThreads / 2; off >= 1; off >>= 1) {
    if (tid < off) rs[tid] += rs[tid + off], rc[tid] += rc[tid + off];
    __syncthreads();
  }
  for (int j = tid; j < A.T_out; j += kThreads) A.dur[(size_t)b * A.T_out + j] = j < T ? (float)durs[j] : 0.0f;
  if (tid == 0) A.score[b] = rc[0] ? rs[0] / rc[0] : __longlong_as_double(0x7ff8000000000000LL);  // NaN: the reference divides by zero
}

size_t dur_lds_bytes(int S, int T) {
  return 3 * (size_t)T * sizeof(double) + (size_t)T * sizeof(int) + 5 * (size_t)S + (size_t)T + kWinBytes + 32;
}

}  // namespace

extern "C" size_t wrnn_durations_workspace_bytes(int B, int att_rows, int att_ld) {
  if (B <= 0 || att_rows <= 0 || att_ld <= 0) return 0;
  return (size_t)B * att_rows * att_ld;
}

extern "C" int wrnn_durations_extract(const float* att, int B, int att_rows, int att_ld, const float* mel, int n_mels,
                                      int mel_ld, const int32_t* tokens, int tok_ld, const int32_t* mel_len,
                                      const int32_t* x_len, int max_mel_len, int max_x_len,
                                      const int32_t* silent_ids, int n_silent, float silence_threshold,
                                      float silence_prob_shift, void* workspace, float* durations, int dur_ld,
                                      double* att_score, double* path_cost, void* stream) {
  WRNN_REQUIRE(B >= 0, "wrnn_durations_extract: negative batch size");
  if (B == 0) return 0;
  WRNN_REQUIRE(att && mel && tokens && mel_len && x_len && workspace && durations && att_score,
               "wrnn_durations_extract: null pointer");
  WRNN_REQUIRE(n_silent == 0 || silent_ids, "wrnn_durations_extract: null silent_ids");
  WRNN_REQUIRE(max_mel_len >= 1 && max_mel_len <= att_rows && max_mel_len <= mel_ld && max_mel_len <= kMaxS,
               "wrnn_durations_extract: max_mel_len=%d must be in [1, min(att_rows=%d, mel_ld=%d, %d)]", max_mel_len,
               att_rows, mel_ld, kMaxS);
  WRNN_REQUIRE(max_x_len >= 1 && max_x_len <= att_ld && max_x_len <= tok_ld && max_x_len <= dur_ld && max_x_len <= kMaxT,
               "wrnn_durations_extract: max_x_len=%d must be in [1, min(att_ld=%d, tok_ld=%d, dur_ld=%d, %d)]", max_x_len,
               att_ld, tok_ld, dur_ld, kMaxT);
  WRNN_REQUIRE(n_mels >= 1 && n_silent >= 0, "wrnn_durations_extract: bad n_mels / n_silent");
  const size_t lds = dur_lds_bytes(max_mel_len, max_x_len);
  WRNN_REQUIRE(lds <= 64 * 1024, "wrnn_durations_extract: %zu bytes of LDS needed, limit 65536", lds);
  DurArgs A{att, mel, tokens, mel_len, x_len, silent_ids, n_silent, att_rows, att_ld, n_mels, mel_ld, tok_ld, dur_ld, max_mel_len, max_x_len,
            silence_threshold, silence_prob_shift, static_cast<unsigned char*>(workspace), durations, att_score, path_cost};
  hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(kThreads), lds, (hipStream_t)stream, A);
  WRNN_LAUNCH_CHECK();
  return 0;
}
