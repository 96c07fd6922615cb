// Bidirectional single-layer LSTM over M frames for gfx950 (MI355X), fp32 storage and accumulate.
//
// Replaces the `nn.LSTM(2*prenet_dims, rnn_dims, batch_first=True, bidirectional=True)` call that
// follows the length regulator in the reference's forward models (models/forward_tacotron.py:96-99,
// :147-152 packed training path, :224 generate path; models/multi_forward_tacotron.py:164-168,294).
//
// Structure: both directions advance in lock step, one launch per time step carrying two
// LinJob<2,1,EpiBiLstm> (forward and reverse cell) as block ranges of one tick_kernel.  The input
// projection is not hoisted: [x_t | h] is one K = in + H contraction per cell, the same shape the
// decoder's LSTM stage runs.  A pre-pass re-lays x out step-major interleaved (see linjob.h) once per
// direction; for the reverse direction it also aligns every row to its own last frame
// (x[b, len_b-1-s]) so that packed-sequence semantics need no per-row gather inside the loop.
#include "birnn.h"
#include "bilstm_resident.h"

namespace {

struct EpiBiLstm {  // tile rows: 4 units x (i, f, g, o)
  const float* bias;  // tile order, b_ih + b_hh
  Act c_prev, c_out, h_prev, h_out;
  const int32_t* lens;  // [B] or null (= every row has M frames)
  float* out;           // [B, M, H2]
  int M, H2, col0, s, rev;
  struct Pre { f32x4 b; float c, hp; int len; };
  __device__ __forceinline__ Pre prefetch(int tile, int u, const RowId& row) const {
    Pre q;
    const int U = tile * 4 + u;
    q.b = *reinterpret_cast<const f32x4*>(bias + tile * 16 + u * 4);
    q.c = c_prev.rowbase(row)[c_prev.eoff(U)];
    q.hp = h_prev.rowbase(row)[h_prev.eoff(U)];
    q.len = lens ? min(max(lens[row.b], 0), M) : M;
    return q;
  }
  __device__ __forceinline__ void apply(const Pre& q, int tile, int u, const RowId& row, f32x4 v) const {
    const int U = tile * 4 + u;
    const float ig = sigmoidf_(v[0] + q.b[0]);
    const float fg = sigmoidf_(v[1] + q.b[1]);
    const float gg = tanhf_(v[2] + q.b[2]);
    const float og = sigmoidf_(v[3] + q.b[3]);
    float c = fg * q.c + ig * gg;
    float h = og * tanhf_(c);
    const bool active = s < q.len;  // past the row's last frame the state is frozen (packed semantics)
    if (!active) c = q.c, h = q.hp;
    c_out.rowbase(row)[c_out.eoff(U)] = c;
    h_out.rowbase(row)[h_out.eoff(U)] = h;
    if (active) {
      const int t = rev ? q.len - 1 - s : s;
      out[((size_t)row.b * M + t) * H2 + col0 + U] = h;
    }
  }
};

using BiLstmJob = LinJob<4, 1, EpiBiLstm, 2>;  // 128 workgroups per direction: one per compute unit for both (measured: 10.5 -> 7.8 ms per 867-frame batch vs <2,1,.,4>)

}  // namespace

struct wrnn_bilstm {
  int device = 0, in = 0, H = 0;
  Layer fwd, rev;
  float* ws = nullptr;
  size_t ws_floats = 0;
  float* wres = nullptr;   // resident path: both directions' weights in register order (bilstm_resident.h), in = H = 512 only
  unsigned* flags = nullptr;
  int path = WRNN_BILSTM_PATH_AUTO;
  int cus = 0;
};

extern "C" int wrnn_bilstm_destroy(wrnn_bilstm* h) {
  if (!h) return 0;
  Guard g(h->device);
  (void)hipDeviceSynchronize();
  free_layer(h->fwd), free_layer(h->rev);
  if (h->ws) (void)hipFree(h->ws);
  if (h->wres) (void)hipFree(h->wres);
  if (h->flags) (void)hipFree(h->flags);
  delete h;
  return 0;
}

extern "C" int wrnn_bilstm_create(int in_dims, int rnn_dims, const wrnn_bilstm_weights* hw, int device,
                                  wrnn_bilstm** out) {
  WRNN_REQUIRE(hw && out, "wrnn_bilstm_create: null argument");
  *out = nullptr;
  WRNN_REQUIRE(in_dims >= 4 && in_dims % 4 == 0, "wrnn_bilstm_create: in_dims=%d must be a positive multiple of 4", in_dims);
  WRNN_REQUIRE(rnn_dims >= 4 && rnn_dims % 4 == 0, "wrnn_bilstm_create: rnn_dims=%d must be a positive multiple of 4", rnn_dims);
  const float* const* wp = reinterpret_cast<const float* const*>(hw);
  for (size_t i = 0; i < sizeof(*hw) / sizeof(float*); ++i)
    WRNN_REQUIRE(wp[i] != nullptr, "wrnn_bilstm_create: weight pointer #%zu is null", i);
  int ndev = 0;
  WRNN_HIP(hipGetDeviceCount(&ndev));
  WRNN_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d visible)", device, ndev);
  hipDeviceProp_t prop;
  WRNN_HIP(hipGetDeviceProperties(&prop, device));
  WRNN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libwrnn_mi355 is built for gfx950 only, device %d is %s",
               device, prop.gcnArchName);
  Guard g(device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", device);
  auto* h = new wrnn_bilstm();
  h->device = device, h->in = in_dims, h->H = rnn_dims, h->cus = prop.multiProcessorCount;
  const int in = in_dims, H = rnn_dims;
  // K = [x (in) | h (H)], tile rows = 4 units x (i, f, g, o); torch row order is gate-major
  auto build = [&](Layer& L, const float* Wih, const float* Whh, const float* bih, const float* bhh) {
    RowFn w = [=](int t, int i, int k) -> float {
      const int row = (i & 3) * H + t * 4 + (i >> 2);
      return k < in ? Wih[(size_t)row * in + k] : Whh[(size_t)row * H + (k - in)];
    };
    std::function<float(int, int)> b = [=](int t, int i) -> float {
      const int row = (i & 3) * H + t * 4 + (i >> 2);
      return bih[row] + bhh[row];
    };
    return build_layer(L, H / 4, in + H, 4 * H, w, &b);
  };
  int rc = build(h->fwd, hw->w_ih, hw->w_hh, hw->b_ih, hw->b_hh);
  rc |= build(h->rev, hw->w_ih_reverse, hw->w_hh_reverse, hw->b_ih_reverse, hw->b_hh_reverse);
  if (!rc && in == kBRH && H == kBRH) {
    // resident path: workgroup (dir, tile j), wave w, slot (0..31 x half, 32..63 h half), lane (kq = lane / 16, tile row lane % 16)
    std::vector<float> wr((size_t)2 * 128 * 4 * 64 * 64);
    for (int dir = 0; dir < 2; ++dir) {
      const float* Wih = dir ? hw->w_ih_reverse : hw->w_ih;
      const float* Whh = dir ? hw->w_hh_reverse : hw->w_hh;
      for (int j = 0; j < 128; ++j)
        for (int w = 0; w < 4; ++w)
          for (int slot = 0; slot < 64; ++slot)
            for (int lane = 0; lane < 64; ++lane) {
              const int i = lane & 15, kq = lane >> 4;
              const int row = (i & 3) * H + j * 4 + (i >> 2);
              const int k = 128 * w + 32 * kq + (slot & 31);
              wr[((((size_t)dir * 128 + j) * 4 + w) * 64 + slot) * 64 + lane] = (slot < 32 ? Wih : Whh)[(size_t)row * kBRH + k];
            }
    }
    rc |= upload(wr.data(), wr.size() * sizeof(float), (void**)&h->wres);
    if (!rc && hipMalloc((void**)&h->flags, (kBRErr + 4) * sizeof(unsigned)) != hipSuccess)
      rc = wrnn::fail("wrnn_bilstm_create: hipMalloc of the flags failed");
  }
  if (rc) {
    std::string keep = wrnn::last_error();
    wrnn_bilstm_destroy(h);
    wrnn::last_error() = keep;
    return 1;
  }
  *out = h;
  return 0;
}

extern "C" int wrnn_bilstm_set_path(wrnn_bilstm* h, int path) {
  WRNN_REQUIRE(h != nullptr, "null bilstm handle");
  WRNN_REQUIRE(path == WRNN_BILSTM_PATH_AUTO || path == WRNN_BILSTM_PATH_LAUNCHES || path == WRNN_BILSTM_PATH_RESIDENT,
               "wrnn_bilstm_set_path: unknown path %d", path);
  h->path = path;
  return 0;
}

extern "C" int wrnn_bilstm_forward(wrnn_bilstm* h, const float* x, const int32_t* lens, int B, int M,
                                   float padding_value, float* out, float* h_n, float* c_n, void* stream) {
  WRNN_REQUIRE(h != nullptr, "null bilstm handle");
  WRNN_REQUIRE(B >= 0 && M >= 0, "wrnn_bilstm_forward: negative size (B=%d, M=%d)", B, M);
  WRNN_REQUIRE(B <= 65535, "wrnn_bilstm_forward: B=%d out of range", B);
  if (B == 0) return 0;
  WRNN_REQUIRE((x && out) || M == 0, "wrnn_bilstm_forward: null tensor");
  Guard g(h->device);
  WRNN_REQUIRE(g.ok, "cannot select device %d", h->device);
  hipStream_t st = (hipStream_t)stream;
  const int in = h->in, H = h->H;
  const size_t xs = (size_t)M * B * in, ss = (size_t)B * H;
  const bool can_resident = h->wres && B <= 64 && M >= 1 && h->cus >= 256;
  WRNN_REQUIRE(h->path != WRNN_BILSTM_PATH_RESIDENT || can_resident,
               "wrnn_bilstm_forward: the resident path needs in_dims = rnn_dims = 512, B <= 64, M >= 1 and a 256-CU device");
  const bool resident = can_resident && h->path != WRNN_BILSTM_PATH_LAUNCHES;
  static const bool flag_kernels = getenv("WRNN_BILSTM_FLAGS") != nullptr;
  const int MT = std::max((B + 15) / 16, resident && !flag_kernels ? 2 : 1);   // (the tag kernel's two row groups need two row tiles)
  // resident path workspace: x in operand order for ONE chunk of frames (both directions), the two-slot h ring, the state
  // between the chunks' launches: independent of M beyond kBRChunk frames (56 MB + 1 MB at B = 64)
  const size_t per_f = (size_t)MT * kBRBlockF4 * 4;                       // floats per (direction, frame)
  const size_t xopf = (size_t)2 * std::min(M, kBRChunk) * per_f, hopf = (size_t)2 * 2 * per_f, stf = (size_t)2 * 2 * ss;
  const size_t need = resident ? xopf + hopf + stf : 2 * xs + 8 * ss;
  if (need > h->ws_floats) {
    if (h->ws) {
      WRNN_HIP(hipDeviceSynchronize());
      WRNN_HIP(hipFree(h->ws));
      h->ws = nullptr, h->ws_floats = 0;
    }
    WRNN_HIP(hipMalloc((void**)&h->ws, need * sizeof(float)));
    h->ws_floats = need;
  }
  if (resident) {
    float* xop = h->ws;
    float* hop = xop + xopf;
    float* st_h = hop + hopf;
    float* st_c = st_h + 2 * ss;
    WRNN_HIP(hipMemsetAsync(h->flags, 0, (kBRErr + 4) * sizeof(unsigned), st));
    WRNN_HIP(hipMemsetAsync(hop, 0xFF, hopf * sizeof(float), st));   // the h ring starts as all-ones words: stale under either tag's first frames (bilstm_resident.h)
    if (lens) {
      const size_t tot = (size_t)B * M * (2 * H / 4);
      const unsigned pb = (unsigned)std::min<size_t>((tot + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(birnn_pad_kernel, dim3(pb), dim3(256), 0, st, lens, B, M, 2 * H / 4, padding_value,
                         reinterpret_cast<f32x4*>(out));
      WRNN_LAUNCH_CHECK();
    }
    BiResParams q{};
    q.wres = h->wres, q.bias[0] = h->fwd.bias, q.bias[1] = h->rev.bias;
    q.xop = reinterpret_cast<const f32x4*>(xop), q.hop = reinterpret_cast<f32x4*>(hop);
    q.flags = h->flags, q.lens = lens, q.out = out, q.h_n = h_n, q.c_n = c_n, q.B = B, q.M = M, q.MT = MT;
    q.st_h = st_h, q.st_c = st_c;
    static const bool want_stamps = getenv("WRNN_BILSTM_STAMPS") != nullptr;
    unsigned long long* stamps = nullptr;
    if (want_stamps && M > 104) {
      WRNN_HIP(hipMalloc((void**)&stamps, 32 * sizeof(unsigned long long)));
      WRNN_HIP(hipMemsetAsync(stamps, 0, 32 * sizeof(unsigned long long), st));
      q.stamps = stamps;
    }
    // default: the wide decomposition with the data as the flag (bilstm_resident_tag_kernel; B <= 16 runs it with a second, empty
    // row tile: 2.7 us per frame against 3.1 for the all-rows flag kernel).  WRNN_BILSTM_FLAGS=1: the flag kernels (comparison)
    static const bool flagk = getenv("WRNN_BILSTM_FLAGS") != nullptr;
    static const bool narrow = getenv("WRNN_BILSTM_NARROW") != nullptr;   // with WRNN_BILSTM_FLAGS: all rows per workgroup at every B
    auto kfn = !flagk ? (MT <= 2 ? bilstm_resident_tag_kernel<2> : MT == 3 ? bilstm_resident_tag_kernel<3> : bilstm_resident_tag_kernel<4>)
             : MT == 1 ? bilstm_resident_kernel<1>
             : MT == 2 ? (narrow ? bilstm_resident_kernel<2> : bilstm_resident_wide_kernel<2>)
             : MT == 3 ? (narrow ? bilstm_resident_kernel<3> : bilstm_resident_wide_kernel<3>)
                       : (narrow ? bilstm_resident_kernel<4> : bilstm_resident_wide_kernel<4>);
    // one launch per chunk of at most kBRChunk frames (x is put into operand order for one chunk at a time; c and h cross
    // the boundary in the state buffers, h(s0 - 1) in the ring, the flags count on).  The grid needs all its 256 workgroups
    // resident at once: it is queued on the device's stream for such grids (common.h), between two events that tie it into
    // the caller's stream
    for (int s0 = 0; s0 < M; s0 += kBRChunk) {
      const int s1 = std::min(M, s0 + kBRChunk);
      const size_t total = (size_t)2 * (s1 - s0) * per_f / 4;
      const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(bilstm_operand_pack_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), lens, B, M, MT,
                         s0, s1 - s0, reinterpret_cast<f32x4*>(xop));
      WRNN_LAUNCH_CHECK();
      q.s0 = s0, q.s1 = s1;
      if (wrnn::on_persistent_stream(st, [&](hipStream_t ps) -> int {
            hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 0, ps, q);
            WRNN_LAUNCH_CHECK();
            return 0;
          }))
        return 1;
    }
    if (stamps) {
      unsigned long long hs[32];
      WRNN_HIP(hipStreamSynchronize(st));
      WRNN_HIP(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
      (void)hipFree(stamps);
      for (int f = 0; f < 4; ++f) {
        fprintf(stderr, "bilstm frame %d:", 100 + f);
        for (int i = 1; i < 8; ++i) fprintf(stderr, " %d:%+.2f", i, (double)(long long)(hs[f * 8 + i] - hs[f * 8 + i - 1]) * 0.01);
        if (f < 3) fprintf(stderr, "  next:%+.2f", (double)(long long)(hs[(f + 1) * 8] - hs[f * 8 + 7]) * 0.01);
        fprintf(stderr, " us\n");
      }
    }
    hipLaunchKernelGGL(bilstm_poison_kernel, dim3(256), dim3(256), 0, st, h->flags + kBRErr, out, (size_t)B * M * 2 * H, h_n, c_n, (size_t)2 * B * H);
    WRNN_LAUNCH_CHECK();
    return 0;
  }
  float* xf = h->ws;
  float* xr = xf + xs;
  float* state = xr + xs;  // [dir][h|c][ping-pong] x [H/4][B][4]
  auto hbuf = [&](int dir, int pp) { return state + ((size_t)dir * 4 + pp) * ss; };
  auto cbuf = [&](int dir, int pp) { return state + ((size_t)dir * 4 + 2 + pp) * ss; };
  WRNN_HIP(hipMemsetAsync(state, 0, 8 * ss * sizeof(float), st));
  if (M > 0) {
    const size_t total = (size_t)M * (in / 4) * B;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(birnn_pack_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4*>(x), lens, B,
                       M, in / 4, reinterpret_cast<f32x4*>(xf), reinterpret_cast<f32x4*>(xr));
    WRNN_LAUNCH_CHECK();
    if (lens) {
      const size_t tot = (size_t)B * M * (2 * H / 4);
      const unsigned pb = (unsigned)std::min<size_t>((tot + 255) / 256, 256 * 64);
      hipLaunchKernelGGL(birnn_pad_kernel, dim3(pb), dim3(256), 0, st, lens, B, M, 2 * H / 4, padding_value,
                         reinterpret_cast<f32x4*>(out));
      WRNN_LAUNCH_CHECK();
    }
  }
  for (int s = 0; s < M; ++s) {
    const int pi = s & 1, po = pi ^ 1;
    BiLstmJob jf, jr;
    for (int dir = 0; dir < 2; ++dir) {
      const Act xa = act_il((dir ? xr : xf) + (size_t)s * B * in, B, in);
      const Act hp = act_il(hbuf(dir, pi), B, H), ho = act_il(hbuf(dir, po), B, H);
      const Act cp = act_il(cbuf(dir, pi), B, H), co = act_il(cbuf(dir, po), B, H);
      const Layer& L = dir ? h->rev : h->fwd;
      if (make_lin(dir ? jr : jf, L, Src{xa, in}, Src{hp, H}, kNoSrc, B, B,
                   EpiBiLstm{L.bias, cp, co, hp, ho, lens, out, M, 2 * H, dir * H, s, dir}))
        return 1;
    }
    if (launch_tick(st, jf, jr)) return 1;
  }
  const int last = M & 1;  // buffer holding the state after M steps
  for (int dir = 0; dir < 2; ++dir) {
    const unsigned blocks = (unsigned)((B * H + 255) / 256);
    if (h_n) {
      hipLaunchKernelGGL(birnn_state_kernel, dim3(blocks), dim3(256), 0, st, hbuf(dir, last), B, H, h_n + (size_t)dir * ss);
      WRNN_LAUNCH_CHECK();
    }
    if (c_n) {
      hipLaunchKernelGGL(birnn_state_kernel, dim3(blocks), dim3(256), 0, st, cbuf(dir, last), B, H, c_n + (size_t)dir * ss);
      WRNN_LAUNCH_CHECK();
    }
  }
  return 0;
}
