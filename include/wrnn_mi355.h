/* wrnn_mi355.h -- C ABI of libwrnn_mi355.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the reference's autoregressive hot path.  The reference
 * (as-ideas/WaveRNN @ ForwardTacotron commit) has no FFI of its own: the
 * interface it exposes for this path is the Python class interface of
 * models/tacotron.py and models/common_layers.py.  Each entry point below names
 * the reference code it replaces; wavernn_amd/_lib.py binds them with ctypes and
 * wavernn_amd/{tacotron,length_regulator}.py put the reference's class
 * signatures back on top.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; the message is
 *    available from wrnn_last_error() (thread-local).
 *  - all tensor arguments are DEVICE pointers to contiguous fp32 (row-major, the
 *    layout the PyTorch reference uses) unless the comment says "host".
 *    The caller owns them; the library borrows them for the duration of the
 *    call and enqueues its work on `stream` (a hipStream_t, NULL = default).
 *  - a decoder handle owns device copies of the weights (re-laid-out for MFMA)
 *    and a workspace; it is bound to one device and is NOT re-entrant (the
 *    reference's LSA keeps per-module state the same way, tacotron.py:129-136).
 *    Different handles may be used from different host threads.
 *  - noise is always caller-supplied uniforms in [0,1): prenet dropout keeps an
 *    element iff u >= 0.5 and scales by 2; zoneout keeps the previous hidden
 *    value iff u < 0.1.  NULL disables the respective noise (eval mode).
 */
#ifndef WRNN_MI355_H
#define WRNN_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* wrnn_last_error(void);
/* library / ABI version (major*100 + minor) */
int wrnn_version(void);

/* ---------------------------------------------------------------- decoder */

/* Shapes of reference Decoder(n_mels, decoder_dims, lstm_dims, aligner_out_dims)
 * (models/tacotron.py:164-178).  prenet dims 256/128, LSA 32 filters x 31 taps
 * and max_r = 20 are the reference's fixed defaults (tacotron.py:87,123,162);
 * decoder_dims must be 2*prenet_fc2 (= 256), as the reference's GRUCell input
 * size decoder_dims + decoder_dims/2 implies (tacotron.py:174,199). */
typedef struct wrnn_decoder_dims {
  int32_t n_mels;
  int32_t decoder_dims;
  int32_t lstm_dims;
  int32_t aligner_out_dims;
} wrnn_decoder_dims;

/* HOST pointers to fp32 parameters in the reference state_dict layout
 * (SURVEY.md section 8b key list; shapes in comments, d = decoder_dims,
 * l = lstm_dims, p = n_mels + aligner_out_dims). */
typedef struct wrnn_decoder_weights {
  const float* prenet_fc1_w; /* [256,p]   decoder.prenet.fc1.weight */
  const float* prenet_fc1_b; /* [256] */
  const float* prenet_fc2_w; /* [128,256] */
  const float* prenet_fc2_b; /* [128] */
  const float* lsa_conv_w;   /* [32,2,31] decoder.attn_net.conv.weight */
  const float* lsa_L_w;      /* [d,32] */
  const float* lsa_L_b;      /* [d] */
  const float* lsa_W_w;      /* [d,d] */
  const float* lsa_W_b;      /* [d] */
  const float* lsa_v_w;      /* [1,d] */
  const float* gru_w_ih;     /* [3d, d+d/2]  decoder.attn_rnn.weight_ih (r|z|n) */
  const float* gru_w_hh;     /* [3d, d] */
  const float* gru_b_ih;     /* [3d] */
  const float* gru_b_hh;     /* [3d] */
  const float* rnn_input_w;  /* [l, 2d] */
  const float* rnn_input_b;  /* [l] */
  const float* lstm1_w_ih;   /* [4l, l]  decoder.res_rnn1 (i|f|g|o) */
  const float* lstm1_w_hh;   /* [4l, l] */
  const float* lstm1_b_ih;   /* [4l] */
  const float* lstm1_b_hh;   /* [4l] */
  const float* lstm2_w_ih;
  const float* lstm2_w_hh;
  const float* lstm2_b_ih;
  const float* lstm2_b_hh;
  const float* mel_proj_w;   /* [n_mels*20, l]  row j <-> (mel j/20, frame j%20) */
} wrnn_decoder_weights;

typedef struct wrnn_decoder wrnn_decoder;

/* Replaces Decoder.__init__ + load_state_dict (tacotron.py:164-178): uploads
 * and re-lays-out the weights on `device`. */
int wrnn_decoder_create(const wrnn_decoder_dims* dims, const wrnn_decoder_weights* host_weights,
                        int device, wrnn_decoder** out);
int wrnn_decoder_destroy(wrnn_decoder* dec);

/* Recurrent state of one decoder step, all [B, .] fp32 device buffers.
 * lsa_cum / lsa_att are the LSA module state (tacotron.py:129-136), [B,T]. */
typedef struct wrnn_decoder_state {
  float* attn_h;  /* [B,d] */
  float* h1;      /* [B,l] */
  float* h2;      /* [B,l] */
  float* c1;      /* [B,l] */
  float* c2;      /* [B,l] */
  float* context; /* [B,d] */
  float* lsa_cum; /* [B,T] */
  float* lsa_att; /* [B,T] */
} wrnn_decoder_state;

/* Replaces Decoder.forward (tacotron.py:185-235) for one call.
 *   enc_q  [B,T,d]  the tensor the reference passes as `encoder_seq`
 *                   (= encoder_seq_proj_query, tacotron.py:342)
 *   enc_p  [B,T,d]  `encoder_seq_proj`
 *   prenet_in [B,p]
 *   in / out: state before / after the step; `out` buffers must not alias `in`
 *             except lsa_cum / lsa_att, which may (and normally do) alias.
 *   t == 0 starts a sequence: the LSA state is taken as zero and not read
 *   (tacotron.py:140), r in [1,20] frames are produced (tacotron.py:231).
 *   drop_u1 [B,256], drop_u2 [B,128]: prenet dropout uniforms or NULL
 *   zone_u1, zone_u2 [B,l]: zoneout uniforms or NULL (training mode, :215-226)
 *   mels_out [B,n_mels,r], scores_out [B,T] (the reference returns it as [B,1,T]). */
int wrnn_decoder_step(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                      const float* prenet_in, const wrnn_decoder_state* in,
                      const wrnn_decoder_state* out, int B, int T, int t, int r,
                      const float* drop_u1, const float* drop_u2,
                      const float* zone_u1, const float* zone_u2,
                      float* mels_out, float* scores_out, void* stream);

/* Replaces the decoder loop of Tacotron.forward (tacotron.py:299-316,334-348,356):
 * zero initial state, go-frame at t = 0, teacher forcing with frame t-1 of
 *   mel_in [B,p,S]  (= cat[mel, aligner mel], tacotron.py:336)
 * for t in range(0, S, r); n = ceil(S / r) steps.
 *   drop_u1 [n,B,256], drop_u2 [n,B,128] or NULL
 *   zone_u1, zone_u2 [n,B,l] or NULL: zoneout uniforms of every step (training mode, :215-226)
 *   mel_out [B,n_mels,n*r], att_out [B,n,T]. 
 * Inside the loop the query projection shares the attention GRU's launch and waits there, bounded, for the cell's output
 * (tagged hand-over); a wait that timed out -- never observed: the producers are dispatched first -- leaves
 * NaN in the outputs instead of stale values. */
int wrnn_decoder_run_teacher_forced(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                                    const float* mel_in, int B, int T, int S, int r,
                                    const float* drop_u1, const float* drop_u2,
                                    const float* zone_u1, const float* zone_u2,
                                    float* mel_out, float* att_out, void* stream);

/* How wrnn_decoder_run_teacher_forced runs its loop (no counterpart in the reference; default AUTO):
 *   LAUNCHES  two launches per decoder step (any shape): GRU || query || LSTM1, attention || rnn_input || LSTM2;
 *   CHAIN     the loop-carried chain GRU -> query projection -> attention runs as persistent launches (16 .. 64 steps
 *             each) on a stream of the library's own, one per device: each workgroup keeps its slice of the encoder
 *             memories and its weights in registers / LDS for the whole launch, three in-launch hand-overs per step through
 *             tagged pairs, every wait bounded; rnn_input and the two LSTM cells follow on `stream` chunk by chunk behind
 *             events, beside the chain launches.  Needs B <= 64 at T <= 128 or B <= 32 at T <= 256, a 256-CU device.
 *             Chain launches of all handles of a device run one after the other (each needs every compute unit).
 *             With lstm_dims = 512 the two LSTM cells are persistent launches too (one per chunk of at most 64 steps, on a
 *             second stream of the library's own): a workgroup keeps 16 gate rows of one cell in registers for the whole
 *             chunk, one hand-over per cell and step; other lstm_dims run the cells as one launch per step.
 *             A hand-over that timed out (never observed) leaves NaN in mel_out and att_out.
 *   CHAIN_LAUNCH_STACK   CHAIN with the two LSTM cells as one launch per step at any lstm_dims (round 2's schedule; kept for
 *             comparison).
 *   AUTO      CHAIN where it applies, LAUNCHES otherwise.
 * Same arithmetic except the summation order of the query projection (and, in the persistent LSTM launches, of the gate
 * sums): results agree to fp32 rounding. */
#define WRNN_TF_PATH_AUTO 0
#define WRNN_TF_PATH_LAUNCHES 1
#define WRNN_TF_PATH_CHAIN 2
#define WRNN_TF_PATH_CHAIN_LAUNCH_STACK 3
int wrnn_decoder_set_tf_path(wrnn_decoder* dec, int path);

/* Free-running decode.  NOT in the reference (Tacotron.generate does not exist at
 * the mounted commit); defined by the build as the same
 * step fed with its own output: prenet_in = cat[last frame of the previous
 * group, aligner_feats[:, :, k] or zeros], go-frame zeros at k = 0.
 *   aligner_feats [B,aligner_out_dims,max_steps] or NULL
 *   use_stop != 0: stop after the first group k > 10 whose frames are all
 *   < stop_threshold (the registered buffer, tacotron.py:275).
 *   mel_out [B,n_mels,max_steps*r], att_out [B,max_steps,T]: only the first
 *   *n_steps_out (host int) groups are defined on return.
 * Synchronises `stream` before returning when use_stop != 0 (and then also reports a wait inside a launch that gave up --
 * never observed -- as an error; without use_stop such a wait leaves NaN in the outputs). */
int wrnn_decoder_run_free(wrnn_decoder* dec, const float* enc_q, const float* enc_p,
                          const float* aligner_feats, int B, int T, int max_steps, int r,
                          float stop_threshold, int use_stop,
                          const float* drop_u1, const float* drop_u2,
                          float* mel_out, float* att_out, int* n_steps_out, void* stream);

/* How wrnn_decoder_run_free runs (no counterpart in the reference; default AUTO):
 *   LAUNCHES  four launches per decoder step (any shape): fc1 || mel frames || fc2 || attention GRU || query projection as one
 *             launch with three in-launch hand-overs (tagged granules), attention, rnn_input || LSTM1, LSTM2;
 *   RESIDENT  ONE persistent cooperative launch for the whole call: 256 workgroups, one per compute unit, each holding
 *             its slice of every layer's weights in registers for the whole utterance, stages handed over in-launch;
 *             needs B <= 4, B * T <= 1024 with 160 * B * T' bytes of LDS free (T' = T rounded up to 64: B * T' <= 896 or so),
 *             lstm_dims = 512, a 256-CU device; the call synchronises `stream`;
 *   AUTO      RESIDENT where it applies (the small-batch latency path), LAUNCHES otherwise.
 * Both give the same results to fp32 summation order. */
#define WRNN_FREE_PATH_AUTO 0
#define WRNN_FREE_PATH_LAUNCHES 1
#define WRNN_FREE_PATH_RESIDENT 2
int wrnn_decoder_set_free_path(wrnn_decoder* dec, int path);

/* Measurement aid, no counterpart in the reference: while enabled, run_teacher_forced brackets every
 * launch with HIP events on `stream`, synchronises at the end of the call and keeps, per launch kind,
 * the summed duration (ms) and the launch count of the LAST call.  Adds event overhead; never enable
 * it in a timed region. */
#define WRNN_PROFILE_PRENET 0         /* gather + hoisted prenet GEMM pair (one interval) */
#define WRNN_PROFILE_TICK_GRU 1       /* GRU(k) || query projection(k), behind an in-launch hand-over || LSTM1(k-2) */
#define WRNN_PROFILE_TICK_QUERY 2     /* query projection(k) as a launch of its own: only under WRNN_SEPARATE_QUERY_LAUNCH=1 */
#define WRNN_PROFILE_TICK_ATTENTION 3 /* attention(k) || rnn_input(k-1) || LSTM2(k-2) */
#define WRNN_PROFILE_MEL 4            /* hoisted mel projection */
#define WRNN_PROFILE_CHAIN 5          /* chain path: the persistent GRU -> query -> attention launches of the call taken together (their
                                         own stream: first launch's start to last launch's end); count = 1 */
#define WRNN_PROFILE_LSTM_STACK 6     /* chain path: the LSTM stack's launches (persistent: one per chunk of at most 64 steps, bracketed
                                         around each launch; CHAIN_LAUNCH_STACK: LSTM1(k) || LSTM2(k-1) per step, bracketed chunk by
                                         chunk); the waits for the chain launches are not in it; count = launches */
#define WRNN_PROFILE_RIN_CHUNKS 7     /* chain path: rnn_input as one GEMM per chunk of steps; count = GEMM launches */
#define WRNN_PROFILE_KINDS 8
int wrnn_decoder_set_profiling(wrnn_decoder* dec, int enable);
int wrnn_decoder_get_profile(wrnn_decoder* dec, float* ms_out /*[WRNN_PROFILE_KINDS]*/,
                             int* count_out /*[WRNN_PROFILE_KINDS]*/);

/* ------------------------------------------------------- length regulator */

/* Replaces LengthRegulator.forward (models/common_layers.py:17-24) in two calls,
 * because the output length is data dependent:
 *  1) prepare: clamps negative durations of `dur` [B,T] IN PLACE (:18), computes
 *     inclusive prefix sums of (dur + 0.5) truncated to integer (:21) into
 *     cumsum [B,T] int32 (device) and the per-item totals into lens [B] int32
 *     (device).  The caller reads `lens` back, M = max(lens) (pad_sequence, :23).
 *  2) expand: out[b, m, :] = x[b, j, :] for cumsum[b,j-1] <= m < cumsum[b,j],
 *     zero for m >= lens[b].  x [B,T,C], out [B,M,C].  C must be a multiple of 4. */
int wrnn_lr_prepare(float* dur, int B, int T, int32_t* cumsum, int32_t* lens, void* stream);
int wrnn_lr_expand(const float* x, const int32_t* cumsum, int B, int T, int C,
                   float* out, int M, void* stream);
/* `M` may be any caller-chosen cap: rows m >= the item's length are zero-filled, rows past M are dropped (the
 * item's true length is in `lens` either way), so a caller that knows an upper bound of the output length can
 * enqueue prepare + expand without reading `lens` back in between and narrow afterwards. */

/* Allocation-free variant of the pair above, in ONE call.  It has NO counterpart in the reference; it exists for
 * callers that cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same results as wrnn_lr_prepare(dur, B, T, cumsum, lens) followed by
 * wrnn_lr_expand(x, cumsum, B, T, C, out, M_cap): `dur` clamped in place, `lens` [B] int32 (device) = each item's
 * true length, out [B, M_cap, C] with rows >= lens[b] zero and rows >= M_cap dropped.  The prefix sums live in a
 * file-scope device array of the library (1 MiB per GPU, held until the process ends) instead of a caller buffer:
 *   limit  B * T <= 262144 tokens, larger calls are rejected with an error (use the pair above);
 *   the call makes no hipMalloc / hipFree and calls no allocating routine; its first call on a device resolves the
 *   array's device address (hipGetSymbolAddress), so make that first call outside a graph capture;
 *   every call on one GPU shares that GPU's array: calls that overlap in time on one GPU (different streams or
 *   host threads) race on it and must be serialised by the caller. */
int wrnn_lr_regulate_static(const float* x, float* dur, int B, int T, int C, float* out, int M_cap,
                            int32_t* lens, void* stream);

/* ------------------------------------------------ bidirectional LSTM */

/* HOST pointers to the fp32 parameters of torch.nn.LSTM(in_dims, rnn_dims, bidirectional=True), one
 * layer, in state_dict order (`lstm.weight_ih_l0` ... `lstm.bias_hh_l0_reverse`); gate rows i|f|g|o. */
typedef struct wrnn_bilstm_weights {
  const float* w_ih;         /* [4H, in] */
  const float* w_hh;         /* [4H, H] */
  const float* b_ih;         /* [4H] */
  const float* b_hh;         /* [4H] */
  const float* w_ih_reverse;
  const float* w_hh_reverse;
  const float* b_ih_reverse;
  const float* b_hh_reverse;
} wrnn_bilstm_weights;

typedef struct wrnn_bilstm wrnn_bilstm;

/* Replaces the construction + load_state_dict of `self.lstm` (models/forward_tacotron.py:96-99,
 * models/multi_forward_tacotron.py:164-167). */
int wrnn_bilstm_create(int in_dims, int rnn_dims, const wrnn_bilstm_weights* host_weights, int device,
                       wrnn_bilstm** out);
int wrnn_bilstm_destroy(wrnn_bilstm* h);

/* Replaces `x, _ = self.lstm(x)` on the length-regulated sequence:
 *  - lens == NULL: the generate path (forward_tacotron.py:224), every row runs all M frames;
 *  - lens != NULL (device int32 [B]): the packed training path, pack_padded_sequence -> lstm ->
 *    pad_packed_sequence(padding_value) (forward_tacotron.py:147-152): the reverse direction of row b
 *    starts at frame lens[b]-1, frames t >= lens[b] of `out` hold padding_value, and the final states
 *    are those after each row's own last frame.  lens are clamped to [0, M].
 *  x [B,M,in_dims], out [B,M,2*rnn_dims] (forward | reverse), h_n / c_n [2,B,rnn_dims] or NULL.
 * Zero initial state, as in the reference. */
int wrnn_bilstm_forward(wrnn_bilstm* h, const float* x, const int32_t* lens, int B, int M,
                        float padding_value, float* out, float* h_n, float* c_n, void* stream);

/* How wrnn_bilstm_forward runs (no counterpart in the reference, where it is cuDNN's / MIOpen's choice):
 *   LAUNCHES  one launch per frame carrying the forward and the reverse cell;
 *   RESIDENT  one persistent launch per call, both cells' weights in registers for the whole sequence (256 workgroups,
 *             two 16-gate-row tiles for one of two groups of batch rows each; a frame's hidden state hands over as
 *             tagged data: the last mantissa bit of every published h is the frame's tag (frame / 2 & 1), consumers
 *             re-request their operand block until every word shows it -- so the state the recurrence continues from,
 *             out, h_n, is at most one ulp (6e-8 relative) from the cell's fp32 result).  Needs in_dims = rnn_dims = 512
 *             (the reference's ForwardTacotron sizes), B <= 64, M >= 1, a 256-CU device; asked for explicitly on
 *             any other shape, forward fails.  The launch is queued on the library's per-device stream for persistent grids
 *             (as the decoder's chain launches and the one-launch free-running / WaveRNN kernels are) and tied into the
 *             caller's stream by events: calls from several
 *             handles, streams and host threads may be in flight, the persistent grids run one after the other;
 *             a call runs as launches of at most 224 frames (the state crosses the boundary in a small buffer);
 *             workspace (grown on demand, kept by the handle), independent of M beyond 224 frames: x of one chunk of frames
 *             in the matrix cores' operand order (2 directions x min(M, 224) x max(2, ceil(B / 16)) x 128 KiB) plus a two-slot
 *             ring of h and the state: 57 MB at B = 64;
 *   AUTO      (default) RESIDENT where it applies, else LAUNCHES.
 * The two agree to fp32 summation order and the tag bit. */
#define WRNN_BILSTM_PATH_AUTO 0
#define WRNN_BILSTM_PATH_LAUNCHES 1
#define WRNN_BILSTM_PATH_RESIDENT 2
int wrnn_bilstm_set_path(wrnn_bilstm* h, int path);

/* ------------------------------------- bidirectional GRU of the CBHG block */

/* HOST pointers to the fp32 parameters of torch.nn.GRU(in_dims, rnn_dims, bidirectional=True), one
 * layer, in state_dict order (`rnn.weight_ih_l0` ... `rnn.bias_hh_l0_reverse`); gate rows r|z|n. */
typedef struct wrnn_bigru_weights {
  const float* w_ih;         /* [3H, in] */
  const float* w_hh;         /* [3H, H] */
  const float* b_ih;         /* [3H] */
  const float* b_hh;         /* [3H] */
  const float* w_ih_reverse;
  const float* w_hh_reverse;
  const float* b_ih_reverse;
  const float* b_hh_reverse;
} wrnn_bigru_weights;

typedef struct wrnn_bigru wrnn_bigru;

/* Replaces the construction + load_state_dict of CBHG's `self.rnn` (models/common_layers.py:89). */
int wrnn_bigru_create(int in_dims, int rnn_dims, const wrnn_bigru_weights* host_weights, int device,
                      wrnn_bigru** out);
int wrnn_bigru_destroy(wrnn_bigru* h);

/* Replaces `x, _ = self.rnn(x)` (models/common_layers.py:123; reached from the Tacotron encoder,
 * models/tacotron.py:82, its postnet, :351, and the forward models' CBHGs).  Same conventions as
 * wrnn_bilstm_forward: x [B,M,in_dims], out [B,M,2*rnn_dims] (forward | reverse), h_n [2,B,rnn_dims] or
 * NULL; lens == NULL runs every row for all M frames (what CBHG does), lens != NULL gives
 * pack_padded_sequence / pad_packed_sequence(padding_value) semantics.  Zero initial state.
 * rnn_dims a multiple of 32 up to 128 runs as one kernel per call with W_hh held in registers (a workgroup per row
 * and direction); rnn_dims 192 and 256 (the forward models' CBHGs) likewise with a PAIR of workgroups per row and
 * direction, half of W_hh each, which exchange their halves of the state after every frame (launches of at most
 * compute_units / 4 rows on the library's per-device stream for grids that must be resident at once, tied into the
 * caller's stream by events; a timed-out exchange leaves NaN in out and h_n); other sizes launch once per frame.
 * Same results either way, to fp32 summation order. */
int wrnn_bigru_forward(wrnn_bigru* h, const float* x, const int32_t* lens, int B, int M,
                       float padding_value, float* out, float* h_n, void* stream);

/* ------------------------------------- Aligner attention map */

/* Replaces the pairwise distance of Aligner.forward (models/tacotron.py:63-64):
 *   att[b, s, t] = -|| text_feats[b, t, :] - mel_feats[b, s, :] ||_2
 * mel_feats [B,S,D] (`mel` after mel_lin), text_feats [B,T,D] (`x` after text_lin), att [B,S,T].  The reference
 * materialises the [B,S,T,D] difference tensor first; this entry point never does. */
int wrnn_aligner_distance(const float* mel_feats, const float* text_feats, int B, int S, int T, int D,
                          float* att, void* stream);

/* ------------------------------------- attention -> durations */

/* Replaces DurationExtractor.__call__ (duration_extraction/duration_extractor.py:23-84, graph
 * construction :94-130) for a padded batch, one utterance per workgroup, consuming the decoder loop's
 * attention output where it lies in HBM (the reference goes through att_pred/<id>.npy files,
 * duration_extraction_pipe.py:120-125).
 *   att [B, att_rows, att_ld]      attention, row i = mel frame, column j = token (e.g. att_out of
 *                                  wrnn_decoder_run_teacher_forced with r = 1); only [:mel_len[b], :x_len[b]] is read
 *   mel [B, n_mels, mel_ld]        frame i is silent iff mean over n_mels < silence_threshold (:42)
 *   tokens [B, tok_ld] int32, silent_ids [n_silent] int32 (utils/text/symbols.py silent_phonemes_indices)
 *   mel_len, x_len [B] int32 (device), each clamped to [1, max_mel_len] / [1, max_x_len]
 *   workspace: wrnn_durations_workspace_bytes(B, att_rows, att_ld) bytes of device scratch
 *   durations [B, dur_ld] fp32: frames per token, zero for j >= x_len[b] (:68,79-80)
 *   att_score [B] fp64: mean clamped attention along the path over non-silent frames (:70-82); NaN where
 *                       the reference would divide by zero (every frame silent)
 *   path_cost [B] fp64 or NULL: distance of the bottom-right node, bit-identical to scipy's Dijkstra.
 * Ties between predecessors: down-right, then down, then right (scipy's own choice depends on its
 * heap order and is not reproducible in a data-parallel form).  The reference also shifts the caller's
 * attention rows in place (:50); this entry point leaves `att` untouched. */
size_t wrnn_durations_workspace_bytes(int B, int att_rows, int att_ld);
int wrnn_durations_extract(const float* att, int B, int att_rows, int att_ld, const float* mel, int n_mels,
                           int mel_ld, const int32_t* tokens, int tok_ld, const int32_t* mel_len,
                           const int32_t* x_len, int max_mel_len, int max_x_len,
                           const int32_t* silent_ids, int n_silent, float silence_threshold,
                           float silence_prob_shift, void* workspace, float* durations, int dur_ld,
                           double* att_score, double* path_cost, void* stream);

/* --------------------------------- WaveRNN generation loop */

/* NO counterpart at the mounted reference commit: the WaveRNN vocoder is absent from the tree (SURVEY.md
 * section 0.1; only its hyper-parameters, tests/resources/test_config.yaml:47-75, survive).  Build-defined
 * after the fatchord WaveRNN the reference forked from (README.md:4); parity is UNPINNED and checked against
 * the build's own CPU restatement only. */
typedef struct wrnn_wavernn_dims {
  int32_t rnn_dims;   /* 512 */
  int32_t fc_dims;    /* 512 */
  int32_t feat_dims;  /* mel channels, 80 */
  int32_t aux_dims;   /* res_out_dims / 4, 32 */
  int32_t n_classes;  /* outputs of fc3: 2 ** bits (RAW mode, 512) or 3 * nr_mix (MOL mode, 30) */
  int32_t mode;       /* WRNN_WAVERNN_RAW or WRNN_WAVERNN_MOL */
} wrnn_wavernn_dims;
#define WRNN_WAVERNN_RAW 0 /* softmax over n_classes, inverse-CDF sampling */
#define WRNN_WAVERNN_MOL 1 /* mixture of logistics: the sampler of utils/distribution.py:87-123 (see wrnn_mol_sample) */

/* HOST pointers to fp32 parameters, torch layouts (R = rnn_dims, F = fc_dims, f = feat_dims, a = aux_dims). */
typedef struct wrnn_wavernn_weights {
  const float* I_w;        /* [R, 1 + f + a]  columns: previous sample | mel | aux slice 1 */
  const float* I_b;        /* [R] */
  const float* rnn1_w_ih;  /* [3R, R]  GRU rows r|z|n */
  const float* rnn1_w_hh;  /* [3R, R] */
  const float* rnn1_b_ih;  /* [3R] */
  const float* rnn1_b_hh;  /* [3R] */
  const float* rnn2_w_ih;  /* [3R, R + a] */
  const float* rnn2_w_hh;  /* [3R, R] */
  const float* rnn2_b_ih;
  const float* rnn2_b_hh;
  const float* fc1_w;      /* [F, R + a] */
  const float* fc1_b;
  const float* fc2_w;      /* [F, F + a] */
  const float* fc2_b;
  const float* fc3_w;      /* [n_classes, F] */
  const float* fc3_b;
} wrnn_wavernn_weights;

typedef struct wrnn_wavernn wrnn_wavernn;

int wrnn_wavernn_create(const wrnn_wavernn_dims* dims, const wrnn_wavernn_weights* host_weights, int device,
                        wrnn_wavernn** out);
int wrnn_wavernn_destroy(wrnn_wavernn* h);

/* The sample-by-sample loop of WaveRNN.generate in RAW mode over `seq` steps for B independent rows
 * (the folds of an utterance, or whole utterances), zero initial state:
 *   mels [B, seq, feat_dims], aux [B, seq, 4 * aux_dims]: upsampled features (already folded)
 *   uniforms RAW mode: [B, seq] in [0, 1): the class of step t is the number of inclusive softmax-CDF values
 *                     <= u * sum (inverse-CDF sampling), so results are reproducible
 *            MOL mode: [B, seq, nr_mix + 1] in (1e-5, 1 - 1e-5): nr_mix Gumbel-max uniforms, then the logistic
 *                     one (the two draws of utils/distribution.py:106,118; log_scale_min = ln 1e-14)
 *   x_forced [B, seq] or NULL: when given, the value fed back as the next step's input is
 *                     x_forced[b, t] instead of the drawn sample (teacher forcing, for parity checks)
 *   out [B, seq]: samples 2c / (n_classes - 1) - 1 (RAW) or the logistic sample clamped to [-1, 1] (MOL)
 *   logits_out [B, seq, n_classes] or NULL: the pre-softmax outputs of every step. */
int wrnn_wavernn_generate(wrnn_wavernn* h, const float* mels, const float* aux, const float* uniforms, int B,
                          int seq, const float* x_forced, float* out, float* logits_out, void* stream);

/* How wrnn_wavernn_generate runs its sample loop (default AUTO).  Build-defined, no counterpart in the reference:
 *   LAUNCHES  five launches per sample, replayed as a HIP graph (any B);
 *   RESIDENT  ONE persistent cooperative launch for the whole call, weights in registers, in-launch hand-overs:
 *             B = 1 (the unbatched, sample-by-sample path): 160 workgroups, each holding a few units of one GRU cell or
 *             rows of fc1..fc3, five hand-overs per sample (wavernn_resident.h);
 *             B = 2 .. 32 (the folds of an utterance): every 16 rows a pipeline of 128 workgroups, the matrix stages on the
 *             matrix cores with register-resident weights, five tagged hand-overs per sample (wavernn_folds.h);
 *             needs rnn_dims = fc_dims = 512, aux_dims <= 64, n_classes <= 512, B <= 32 and a 256-CU device, otherwise the
 *             call fails; the call synchronises `stream`.  In the B >= 2 kernel every value handed from stage to stage
 *             (the GRU states among them) carries the step's parity in its last mantissa bit;
 *   AUTO      RESIDENT for B <= 32 where it applies, 33 .. 64 rows as two RESIDENT calls one after the other, LAUNCHES otherwise.
 * All paths draw with the same arithmetic from their logits; the logits agree to fp32 summation order. */
#define WRNN_WAVERNN_PATH_AUTO 0
#define WRNN_WAVERNN_PATH_LAUNCHES 1
#define WRNN_WAVERNN_PATH_RESIDENT 2
int wrnn_wavernn_set_path(wrnn_wavernn* h, int path);

/* Replaces sample_from_discretized_mix_logistic (utils/distribution.py:87-123), the mixture-of-logistics sampler of
 * the WaveRNN MOL mode and the one piece of the vocoder that the mounted tree still holds (pinned by a fixture that
 * tools/make_golden.py produced from the reference function itself).
 *   y [B, 3*nr_mix, T]: logit_probs | means | log_scales along dim 1, the layout the reference takes (:91)
 *   u_mix [B, T, nr_mix], u_logistic [B, T]: uniforms in (1e-5, 1 - 1e-5), in place of the reference's two internal
 *       uniform_(1e-5, 1 - 1e-5) draws (:106 Gumbel-max mixture pick, :118 logistic sample)
 *   log_scale_min: ln(1e-14) = -32.2361913 is the reference's default (:96-97)
 *   out [B, T] in [-1, 1]; mix_out [B, T] int32 (the picked component) or NULL. */
int wrnn_mol_sample(const float* y, const float* u_mix, const float* u_logistic, int B, int nr_mix, int T,
                    float log_scale_min, float* out, int32_t* mix_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WRNN_MI355_H */
