// Diagnostic: times each stage job in isolation (1000 launches in a hipGraph) and prints in-kernel
// phase stamps.  Includes the library source directly (WRNN_STAMP build) to reach its internals.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stage_bench stage_bench.hip
#define WRNN_STAMP 1
#include "../../wavernn_amd/csrc/taco_decoder.hip"
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float* dev_rand(size_t n, float scale = 0.1f) {
  std::vector<float> h(n); std::mt19937 g(n); std::uniform_real_distribution<float> u(-scale, scale);
  for (auto& v : h) v = u(g);
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
template <class F> static void time_graph(const char* name, F&& launch, int iters = 1000) {
  hipStream_t st; CK(hipStreamCreate(&st)); hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; ++i) launch(st);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  printf("%-34s %7.2f us per launch (graph)\n", name, best * 1e3 / iters);
}
static unsigned long long* st_d;
static void report(const char* name, int nblocks, int nst) {
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)nblocks * 4 * 16);
  CK(hipMemcpy(h.data(), st_d, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; for (int i = 0; i < nblocks * 4; ++i) if (h[i * 16]) t0 = std::min(t0, h[i * 16]);
  printf("%-10s stamps us (min..max over waves):", name);
  int last = 0;
  for (int k = 0; k < nst; ++k) { double mx = 0, mn = 1e30; bool any = false; for (int i = 0; i < nblocks * 4; ++i) if (h[i * 16 + k]) { any = true; double v = (h[i * 16 + k] - t0) * 0.01; mx = std::max(mx, v); mn = std::min(mn, v); } if (any) { printf(" [%d] %.2f..%.2f", k, mn, mx); last = k; } }
  // shader clock seen by wave 0 of block 0 between its first and last stamp
  int lk = 0; for (int k = 0; k < nst; ++k) if (h[k]) lk = k;
  if (lk > 0) printf("  | clock %.2f GHz", (double)(h[8 + lk] - h[8]) / ((h[lk] - h[0]) * 10.0));
  printf("\n");
}

int main() {
  const int B = 64, T = 128, d = 256, l = 512, nm = 80, A = 32, P = nm + A;
  wrnn_decoder_dims dims{nm, d, l, A};
  std::vector<std::vector<float>> store;
  auto hr = [&](size_t n) { store.emplace_back(n); std::mt19937 g(n); std::uniform_real_distribution<float> u(-0.05f, 0.05f); for (auto& v : store.back()) v = u(g); return store.back().data(); };
  wrnn_decoder_weights w{hr(256 * P), hr(256), hr(128 * 256), hr(128), hr(32 * 2 * 31), hr(d * 32), hr(d), hr(d * d), hr(d), hr(d),
                         hr(3 * d * (d + 128)), hr(3 * d * d), hr(3 * d), hr(3 * d), hr(l * 2 * d), hr(l),
                         hr(4 * l * l), hr(4 * l * l), hr(4 * l), hr(4 * l), hr(4 * l * l), hr(4 * l * l), hr(4 * l), hr(4 * l), hr(nm * 20 * l)};
  wrnn_decoder* dec;
  if (wrnn_decoder_create(&dims, &w, 0, &dec)) { printf("create failed: %s\n", wrnn_last_error()); return 1; }
  if (ensure_counters(dec, B)) return 1;
  float *enc_q = dev_rand((size_t)B * T * d, 1.f), *enc_p = dev_rand((size_t)B * T * d, 1.f);
  float *pp = dev_rand(B * 128), *ah0 = dev_rand(B * d), *ah1 = dev_rand(B * d), *ctx = dev_rand(B * d), *pq = dev_rand(B * d);
  float *h1a = dev_rand(B * l), *h1b = dev_rand(B * l), *h2a = dev_rand(B * l), *h2b = dev_rand(B * l), *c1 = dev_rand(B * l), *c2 = dev_rand(B * l);
  float *y = dev_rand(B * l), *x1 = dev_rand(B * l), *x2 = dev_rand(B * l);
  float *cum = dev_rand(B * T), *att = dev_rand(B * T), *sc = dev_rand(B * T), *ascr = dev_rand(attn_scratch_floats(B, T));
  StepBufs q{enc_q, enc_p, act_il(pp, B, kFc2), act_il(ah0, B, d), act_il(ah1, B, d), act_il(ctx, B, d),
             act_il(h1a, B, l), act_il(h1b, B, l), act_il(h2a, B, l), act_il(h2b, B, l), act_il(c1, B, l), act_il(c2, B, l), act_il(c1, B, l), act_il(c2, B, l), cum, att, nullptr, nullptr,
             sc, (long)T, act_il(pq, B, d), act_il(y, B, l), act_il(x1, B, l), act_il(x2, B, l), ascr};
  StepJobs J;
  if (make_step_jobs(dec, q, B, T, 0, J)) { printf("jobs failed: %s\n", wrnn_last_error()); return 1; }
  time_graph("gru (K=640,N=1024)", [&](hipStream_t st) { launch_tick(st, J.gru); });
  time_graph("pq (K=256,N=256)", [&](hipStream_t st) { launch_tick(st, J.pq); });
  time_graph("attn", [&](hipStream_t st) { launch_tick(st, J.attn); });
  time_graph("rnn_input (K=512,N=512)", [&](hipStream_t st) { launch_tick(st, J.rin); });
  time_graph("lstm (K=1024,N=2048)", [&](hipStream_t st) { launch_tick(st, J.l1); });
  time_graph("tick gru+rin", [&](hipStream_t st) { launch_tick(st, J.gru, J.rin); });
  time_graph("tick pq+lstm", [&](hipStream_t st) { launch_tick(st, J.pq, J.l1); });
  time_graph("tick attn+lstm", [&](hipStream_t st) { launch_tick(st, J.attn, J.l2); });
  time_graph("3 ticks of a step", [&](hipStream_t st) { launch_tick(st, J.gru, J.rin); launch_tick(st, J.pq, J.l1); launch_tick(st, J.attn, J.l2); }, 500);
  {  // LSTM tile-shape variants (weight tiles x activation tiles per block)
    auto variant = [&](auto job, const char* name) {
      if (make_lin(job, dec->l1, Src{q.y, l}, Src{q.h1_in, l}, kNoSrc, B, B, J.l1.epi)) { printf("%s\n", wrnn_last_error()); exit(1); }
      char buf[64]; snprintf(buf, sizeof buf, "lstm %s (%d blocks)", name, job.nblocks());
      time_graph(buf, [&](hipStream_t st) { launch_tick(st, job); });
    };
    variant(LinJob<1, 1, EpiLstm>(), "TM1 NB1");
    variant(LinJob<2, 1, EpiLstm>(), "TM2 NB1");
    variant(LinJob<4, 1, EpiLstm>(), "TM4 NB1");
    variant(LinJob<1, 2, EpiLstm>(), "TM1 NB2");
    variant(LinJob<2, 2, EpiLstm>(), "TM2 NB2");
    variant(LinJob<1, 4, EpiLstm>(), "TM1 NB4");
    variant(LinJob<2, 1, EpiLstm, 8>(), "TM2 NB1 kU8");
    variant(LinJob<2, 1, EpiLstm, 2>(), "TM2 NB1 kU2");
    variant(LinJob<2, 1, EpiLstm, 6>(), "TM2 NB1 kU6");
    auto gvariant = [&](auto job, const char* name) {
      if (make_lin(job, dec->gru, Src{q.ctx, d}, Src{q.p, kFc2}, Src{q.attn_h_in, d}, B, B, J.gru.epi)) { printf("%s\n", wrnn_last_error()); exit(1); }
      char buf[64]; snprintf(buf, sizeof buf, "gru %s (%d blocks)", name, job.nblocks());
      time_graph(buf, [&](hipStream_t st) { launch_tick(st, job); });
    };
    gvariant(LinJob<1, 1, EpiGru>(), "TM1 NB1");
    gvariant(LinJob<2, 1, EpiGru>(), "TM2 NB1");
    gvariant(LinJob<1, 2, EpiGru>(), "TM1 NB2");
    gvariant(LinJob<2, 2, EpiGru>(), "TM2 NB2");
    gvariant(LinJob<1, 1, EpiGru, 8>(), "TM1 NB1 kU8");
    gvariant(LinJob<1, 1, EpiGru, 2>(), "TM1 NB1 kU2");
    gvariant(LinJob<1, 1, EpiGru, 6>(), "TM1 NB1 kU6");
  }
  CK(hipMalloc(&st_d, 4096 * 4 * 16 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &st_d, sizeof(st_d)));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(st_d, 0, 4096 * 4 * 16 * 8)); launch_tick(0, J.attn); report("attn", J.attn.nblocks(), 8);
    CK(hipMemset(st_d, 0, 4096 * 4 * 16 * 8)); launch_tick(0, J.l1); report("lstm", J.l1.nblocks(), 5);
    CK(hipMemset(st_d, 0, 4096 * 4 * 16 * 8)); launch_tick(0, J.gru); report("gru", J.gru.nblocks(), 5);
  }
  wrnn_decoder_destroy(dec);
  return 0;
}
