/*
 * eeg_dcrnn.h — C ABI of libeeg_dcrnn_hip.so: MI355X (gfx950) kernels for the DCRNN hot path of
 * tsy935/eeg-gnn-ssl (model/cell.py DiffusionGraphConv + DCGRUCell, and the encoder sequence
 * loop of model/model.py).  Plain pointers and sizes only; every pointer is a DEVICE pointer to
 * contiguous fp32 (int64 for lengths) unless stated; `stream` is a hipStream_t passed as void*.
 * All functions enqueue work on `stream` and return without synchronising; they never allocate.
 * Return value: 0 on success, non-zero on error (see eeg_dcrnn_last_error()).
 * Empty operands (zero clips, zero steps, zero widths) are an error of every compute entry, raised before anything is launched --
 * the reference raises on them too (model.py:253-255: its reshape(..., -1) of a tensor without elements); the host-only size /
 * capability queries (size_t eeg_dcrnn_*_floats, eeg_dcrnn_supported, *_ok, *_is_persistent) never fail and return 0 for such dims.
 *
 * The reference has no FFI of its own (pure Python); each entry point below states
 * the reference code it replaces.  The ctypes binding that a maintainer of the reference would
 * add is shown in INTEGRATION.md and lives in eeg_gnn_ssl_amd/_lib.py.
 *
 * Layouts.  N = nodes (<= 32), H = rnn_units (16 | 32 | 64), F/Fin = per-node input features of a
 * layer (multiple of 4), M = number of hop matrices incl. identity = n_supports*K + 1 (<= 8),
 * S = T*B "samples" in time-major order (s = t*B + b).
 *   supports : n_supports pointers, each (G, N, N) with G = B (per-clip graphs) or 1 (shared)
 *   P        : (G, M-1, N, N) hop-polynomial matrices P_1..P_{M-1} (P_0 = I is implicit)
 *   X        : (T, B, N, Fin)            planes : (M-1, S, N, Fin)  = P_m X_s, m = 1..M-1
 *   Hext     : (T+1, B, N, H)  slot 0 = initial state h0, slot t+1 = h_t  (so the layer output
 *              sequence is Hext + B*N*H and the "previous state" sequence is Hext itself)
 *   Rs,Us,Cs,RHs : (T, B, N, H) saved reset gate, update gate, candidate, r*h_prev
 *   Wg (((Fin+H)*M), 2H), bg (2H), Wc (((Fin+H)*M), H), bc (H): reference parameter layout,
 *              row = f*M + m  (cell.py:40-46, 98-116)
 */
#ifndef EEG_DCRNN_H
#define EEG_DCRNN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eeg_layer_dims {
    int32_t T, B, N, H, Fin, M;
    int32_t act;        /* 0 = tanh, 1 = relu  (cell.py:146 `nonlinearity`) */
    int32_t p_batched;  /* 1: P holds one graph per clip (B graphs); 0: one shared graph */
    int32_t x_planes_ready;  /* 1: `planes` already holds P_m X (the previous layer's Hplanes, slots 1..T) */
    int32_t x_batch_major;   /* 1: X is the BATCH-major (B,T,N,Fin) model input and is consumed as it is; `planes` then
                                holds P_m X in the same (b,t) order and the hoisted GEMMs address both through a row
                                map: no time-major copy is written (only where eeg_dcrnn_batch_major_ok() returns 2;
                                pass the same X and planes to layer_bwd) */
    int64_t x_plane_stride;  /* floats between two hop planes of `planes`; 0 = T*B*N*Fin (contiguous) */
    const uint16_t* pack3;   /* OPT-IN (NULL = off, the default and the contract's arithmetic): the cell's three-term bf16 weight packs
                                of eeg_dcrnn_pack_cell_bf16x3.  When set, the two hoisted NN GEMMs of the layer -- the x-part
                                pre-activations (layer_fwd) and the input gradient (layer_bwd) -- run as a three-term bf16 split
                                (6 of the 9 partial products) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: fp32 operands and
                                results, error against an fp64 sum at the level of the fp32 matrix pipe's own (~2e-6 on values
                                of a few units), 1.3-1.4 x its speed.  Everything else of the layer is unchanged. */
    const float* spectral;   /* NULL = off.  Else: the eeg_dcrnn_spectral_basis block of the ONE symmetric support all clips share
                                (p_batched = 0; the scaled Laplacian of the distance graph, filter_type "laplacian") and `spack` the
                                cell's per-frequency weight packs (eeg_dcrnn_pack_cell_spectral).  The hoisted x-part of the layer
                                then runs in the eigenbasis of the support: all hop matrices are Chebyshev polynomials of one
                                symmetric S = U diag(lam) U^T (cell.py:83-93), so sum_m P_m X W_m = U [ (U^T X)_i Wt_i ]_i with
                                Wt_i = sum_m T_m(lam_i) W_m -- the GEMMs contract over Fin instead of M*Fin; the node mixes with U / U^T
                                run inside the recurrent kernels and the input-gradient GEMM (one pass is left: U^T of the layer-0
                                input); backward likewise.  Exact up to re-association (~1e-6); the recurrence is unchanged.
                                Only where eeg_dcrnn_spectral_ok() says so.  `planes` of layer_fwd / layer_bwd is then the
                                node-major transformed input Xh (N, eeg_dcrnn_spectral_rows(T*B), Fin) (written by layer_fwd, read
                                by layer_bwd) and x_planes_ready must be 0. */
    const float* spack;
} eeg_layer_dims;

typedef struct eeg_decoder_dims {
    int32_t T, B, N, H;  /* T = output horizon (decoder steps) */
    int32_t Dout;        /* output_dim = input_dim of the first decoding cell (model.py:131-143) */
    int32_t M, L;        /* hop matrices; num_rnn_layers (layers >= 1 share ONE cell, model.py:126-143) */
    int32_t act, p_batched;
    float dropout_p;     /* nn.Dropout in front of the projection (model.py:191), p when the module is training, else 0 */
    int32_t teacher_on_device;  /* 1: `teacher` of decoder_fwd / _bwd is a DEVICE int32[T] array (e.g. written by
                                   eeg_dcrnn_teacher_flags earlier on the stream) that the persistent kernels read when they
                                   start -- a captured HIP graph then replays with fresh curriculum-learning flags; 0: HOST array */
} eeg_decoder_dims;

/* Human-readable text of the last error raised on the calling thread. */
const char* eeg_dcrnn_last_error(void);
/* ABI version (bumped on any signature change). */
int eeg_dcrnn_abi_version(void);
/* 1 if the library was built for the GPU (always, for the product build), 0 for the test emulator. */
int eeg_dcrnn_is_device_build(void);
/* 1 if kernels are instantiated for this (N, H, Fin, M); else 0 and last_error says why. */
int eeg_dcrnn_supported(int N, int H, int Fin, int M);

/* Clears `bytes` bytes at p on the stream (one small kernel node in a captured graph): `optimizer.zero_grad()` on the flat gradient
 * bucket (train.py:262) and the never-read slot 0 of an upper layer's input gradient, without a framework fill kernel. */
int eeg_dcrnn_zero(void* p, size_t bytes, void* stream);

/* Hop polynomials from the supports (replaces the per-step `torch.matmul(support, x)` chain of
 * cell.py:83-93 incl. the carried-x0 quirk): P_out (G, n_supports*K, N, N). */
int eeg_dcrnn_hop_polys(const float* const* supports, int n_supports, int n_graphs, int N, int K,
                        float* P_out, void* stream);

/* Input featurisation (replaces the DataLoader-side CPU code: data_utils.py:13-35 `computeFFT` applied to
 * every 1-second step as dataloader_detection.py:57-71 does; :233-256 augmentation; utils.py:393-428
 * StandardScaler.transform).  raw (B,N,T*W) resampled signals; W = samples per step (200).
 *   feat_raw (B,T,N,W/2), nullable : log|FFT| of the W/2 positive frequencies (amp == 0 -> 1e-8), the
 *                                    un-augmented clip the correlation graph is built from;
 *   feat_std (B,T,N,W/2), nullable : ((log|FFT| of channel perm[b][n]) + log_scale[b] - mean) / std,
 *                                    the model input.  perm (B,N) int32 / log_scale (B) may be NULL
 *                                    (no reflection / no amplitude jitter).  The transform runs in fp64. */
int eeg_dcrnn_fft_features(const float* raw, int B, int N, int T, int W, const int32_t* perm,
                           const float* log_scale, float mean, float std_, float* feat_raw,
                           float* feat_std, void* stream);

/* Per-clip correlation graph and its dual random-walk supports, from the clips themselves
 * (replaces the DataLoader-side CPU code: dataloader_detection.py:258-307 `_get_indiv_graphs`
 * = |normalised lag-0 cross-correlation| of every electrode pair of the (N, T*D) clip, diag 1;
 * data_utils.py:174-200 keep_topk(top_k, directed=True); dataloader_detection.py:346-349 +
 * utils.py:220-230: S1 = (D^-1 A)^T, S2 = (D_in^-1 A^T)^T).  X (B,T,N,D) batch-major clips;
 * adj (B,N,N) may be NULL; S1, S2 (B,N,N).  HBM-bound: algorithmic bytes 4*B*T*N*D. */
size_t eeg_dcrnn_corr_graph_ws_floats(int B, int T);
int eeg_dcrnn_corr_graph(const float* X, int B, int T, int N, int D, int top_k, float* adj, float* S1,
                         float* S2, float* ws, void* stream);

/* Number of floats of the packed weight block of one DCGRU cell. */
size_t eeg_dcrnn_pack_floats(int Fin, int H, int M);
/* Reference-layout parameters of one cell (cell.py:40-46,160-175) -> MFMA-fragment-ordered block. */
int eeg_dcrnn_pack_cell(const float* Wg, const float* bg, const float* Wc, const float* bc,
                        int Fin, int H, int M, float* pack, void* stream);

/* Three-term bf16 packs (hi / mid / lo, round to nearest even) of the x-part weights of one cell for eeg_layer_dims.pack3:
 * eeg_dcrnn_pack3_halves() 16-bit elements; available for rnn_units = 64 (else 0 / an error). */
size_t eeg_dcrnn_pack3_halves(int Fin, int H, int M);
int eeg_dcrnn_pack_cell_bf16x3(const float* Wg, const float* Wc, int Fin, int H, int M, uint16_t* pack3, void* stream);

/* Spectral form of the hoisted x-part for a shared symmetric support (eeg_layer_dims.spectral).
 *   eeg_dcrnn_spectral_basis: support (N,N) -> basis block of eeg_dcrnn_spectral_basis_floats(N) floats:
 *       U (N*N; column i = eigenvector i) | T_m(lam_i) as [8][32] | info[32]: info[0] = max |S - U diag(lam) U^T| (it contains the
 *       asymmetry of S: a caller reads it ONCE per support and keeps the general path when it is not at rounding level),
 *       info[1] = largest off-diagonal left by the Jacobi sweeps, info[2] = max |S|.  One workgroup, fp64, ~50 us: once per
 *       support, not per step.
 *   eeg_dcrnn_pack_cell_spectral: the cell's x-part weights (reference layout) -> per-frequency packs Wt_i, Wt_i^T.
 *   eeg_dcrnn_spectral_ok: 1 if layer_fwd / layer_bwd (with or without an input gradient) take d->spectral for this shape
 *       (64 units, Fin % 4 == 0; need_dx: Fin == 64), else 0.  eeg_dcrnn_spectral_rows(S) = rows per frequency (S rounded up to 16). */
size_t eeg_dcrnn_spectral_basis_floats(int N);
int eeg_dcrnn_spectral_basis(const float* support, int N, float* basis, void* stream);
size_t eeg_dcrnn_spectral_pack_floats(int Fin, int H, int M, int N);
int eeg_dcrnn_pack_cell_spectral(const float* Wg, const float* Wc, const float* basis, int Fin, int H, int M, int N,
                                 float* spack, void* stream);
/* The packs of ALL cells of an encoder in ONE launch (their weights change with every optimisation step, cell.py:40-46):
 * cell c: eeg_dcrnn_pack_cell(Wg[c], bg[c], Wc[c], bc[c], Fin[c], H, M, packs[c]) and, when basis / spacks are given,
 * eeg_dcrnn_pack_cell_spectral(Wg[c], Wc[c], basis, Fin[c], H, M, N, spacks[c]).  The pointer TABLES are host arrays of
 * device pointers (n_cells <= 4); basis and spacks are NULL together. */
int eeg_dcrnn_pack_cells(int n_cells, const float* const* Wg, const float* const* bg, const float* const* Wc,
                         const float* const* bc, const int32_t* Fin, int H, int M, float* const* packs, const float* basis,
                         int N, float* const* spacks, void* stream);
int eeg_dcrnn_spectral_ok(const eeg_layer_dims* d, int need_dx);
size_t eeg_dcrnn_spectral_rows(size_t S);

/* The HBM-bound diffusion step over all samples: planes[m-1][s] = P_m X_s  (cell.py:83-93 applied
 * to the input features of every time step at once).  Algorithmic bytes: 4*S*N*F*M. */
int eeg_dcrnn_diffuse_fwd(const float* X, const float* P, int p_batched, int S, int B, int N, int F,
                          int M, float* planes, void* stream);
/* Adjoint: dX[s] = Z_0[s] + sum_{m>=1} P_m^T Z_m[s] for Z (S, N, M*F). */
int eeg_dcrnn_diffuse_adj(const float* Z, const float* P, int p_batched, int S, int B, int N, int F,
                          int M, float* dX, void* stream);

/* DiffusionGraphConv.forward (cell.py:66-118) on its own: X (B,N,F) = [inputs | state] per node,
 * W ((F*M), O) and bias (O) in reference layout, out (B,N,O). */
size_t eeg_dcrnn_dconv_fwd_ws_floats(int B, int N, int F, int M, int O);
int eeg_dcrnn_dconv_fwd(const float* X, const float* P, int p_batched, int B, int N, int F, int M,
                        const float* W, const float* bias, int O, float* out, float* ws, void* stream);

/* Backward of the same convolution (autograd's replay of cell.py:66-118): dOut (B,N,O) ->
 * dX (B,N,F), dW ((F*M), O) and dbias (O) in reference layout (each output may be NULL).  O <= 192. */
size_t eeg_dcrnn_dconv_bwd_ws_floats(int B, int N, int F, int M, int O);
int eeg_dcrnn_dconv_bwd(const float* X, const float* P, int p_batched, int B, int N, int F, int M,
                        const float* W, int O, const float* dOut, float* dX, float* dW, float* dbias,
                        float* ws, void* stream);

/* One DCGRU layer over a whole sequence = the `for t` loop of model.py:93-96 around
 * DCGRUCell.forward (cell.py:182-210).  h0 may be NULL (zeros).  Rs/Us/Cs/RHs may all be NULL
 * (inference: nothing saved).  Hplanes / RHplanes (each (M-1, T+1, B, N, H); both NULL or both set):
 * the hop rows P_m h_{t-1} (slot t; slot T = P_m h_{T-1}) and P_m (r*h_{t-1}) (slots 0..T-1) that the
 * recurrent kernel forms in LDS anyway, kept as a by-product: the backward need not re-diffuse h and
 * r*h for its weight-gradient GEMMs, and Hplanes + B*N*H (slots 1..T) ARE the input hop planes of the
 * next layer (pass them as its `planes` with d->x_planes_ready = 1, d->x_plane_stride = (T+1)*B*N*H).  ws: eeg_dcrnn_layer_fwd_ws_floats() floats of scratch.
 * Xtm (nullable): when given, X is BATCH-major (B,T,N,Fin) -- the trainer's `input_seq` before
 * model.py:253's transpose -- and Xtm (T,B,N,Fin) receives its time-major copy as a by-product of
 * the diffusion kernel (pass Xtm as X to eeg_dcrnn_layer_bwd); only where
 * eeg_dcrnn_batch_major_ok() returns >= 1.  Where it returns 2, d->x_batch_major = 1 (with Xtm = NULL) consumes
 * the batch-major input without any copy (the diffusion step then moves exactly 4*S*N*F*M bytes). */
size_t eeg_dcrnn_layer_fwd_ws_floats(const eeg_layer_dims* d);
int eeg_dcrnn_batch_major_ok(const eeg_layer_dims* d);
int eeg_dcrnn_layer_fwd(const eeg_layer_dims* d, const float* X, float* Xtm, const float* h0, const float* P,
                        const float* pack, float* planes, float* Hext, float* Rs, float* Us,
                        float* Cs, float* RHs, float* Hplanes, float* RHplanes, float* ws, void* stream);

/* Backward of the same layer (replaces autograd's replay of model.py:93-96 / cell.py).
 * Incoming gradients (each may be NULL): dHseq (T,B,N,H) w.r.t. every h_t; d_at_end (B,N,H) w.r.t.
 * h_{T-1} (the encoder's per-layer final state, model.py:97); d_at_len (B,N,H) w.r.t.
 * h_{lengths[b]-1} (utils.last_relevant_pytorch; lengths int64 (B), NULL -> T).
 * Hplanes / RHplanes: as written by eeg_dcrnn_layer_fwd, or NULL (recomputed here).
 * Outputs: dX (T,B,N,Fin) or NULL; dh0 (B,N,H) or NULL; dWg/dbg/dWc/dbc in reference layout
 * (overwritten).  Reductions are fixed-order: results are run-to-run deterministic. */
size_t eeg_dcrnn_layer_bwd_ws_floats(const eeg_layer_dims* d, int need_dx);
int eeg_dcrnn_layer_bwd(const eeg_layer_dims* d, const float* X, const float* P, const float* pack,
                        const float* planes, const float* Hext, const float* Rs, const float* Us,
                        const float* Cs, const float* RHs, const float* Hplanes, const float* RHplanes,
                        const float* dHseq, const float* d_at_end,
                        const float* d_at_len, const int64_t* lengths, float* dX, float* dh0,
                        float* dWg, float* dbg, float* dWc, float* dbc, float* ws, void* stream);

/* DCGRUDecoder.forward (model.py:160-204): T autoregressive steps through L cells + the projection
 * Linear(H -> Dout), GO symbol = zeros, next input = the projection of step t or, where the HOST
 * array teacher[t] != 0 (curriculum learning, model.py:194-200; NULL = never), targets[t].
 *   targets (T,B,N,Dout) (only read under teacher forcing, may be NULL when teacher is NULL);
 *   h0 (L,B,N,H) encoder finals; packs: HOST array of L device pointers to cell packs
 *   (packs[1..L-1] = the shared cell); Wp (Dout,H), bp (Dout): nn.Linear layout; out (T,B,N,Dout).
 *   saved: eeg_dcrnn_decoder_saved_floats() floats kept for the backward; ws: scratch.
 * The recurrence cannot be hoisted (the input of step t+1 is the output of step t), so the forward
 * launches per step; the backward runs BPTT per step and ALL parameter gradients as GEMMs hoisted
 * over the T steps.  dWg/dbg/dWc/dbc: HOST arrays of L device pointers (entries 1..L-1 equal:
 * gradients of the shared cell are summed); dh0 (L,B,N,H); dWp (Dout,H); dbp (Dout).
 * d->teacher_on_device = 1: `teacher` is a device array (targets must then be given); available where the persistent decoder
 * kernels run (64 units, <= 20 nodes, <= 4 layers, T <= 64, Dout <= 128 with Dout/4 divisible by 4 or 5), refused elsewhere:
 * the per-step launch sequence is selected by the flags and cannot depend on device memory. */
size_t eeg_dcrnn_decoder_saved_floats(const eeg_decoder_dims* d);
size_t eeg_dcrnn_decoder_fwd_ws_floats(const eeg_decoder_dims* d);
size_t eeg_dcrnn_decoder_bwd_ws_floats(const eeg_decoder_dims* d);
/* d->dropout_p > 0 (model.py:191: `self.projection_layer(self.dropout(output))`, a fresh mask every step): the projection of step t
 * reads mask_t * h_top_t, the recurrence keeps h_top_t.  rng_used = the {seed, offset} pair eeg_dcrnn_rng_take (below) handed out
 * for T*B*N*H/4 counters: element e = ((t*B + b)*N + n)*H + h of the top-layer outputs takes word e%4 of counter offset + e/4.
 * The masks are fused into the persistent kernels (nothing is stored but the dropped rows that dW_p needs) and recomputed in the
 * backward from the same pair.  dropout_p == 0: rng_used may be NULL. */
/* 1 if the persistent decoder kernels cover this shape (then d->teacher_on_device = 1 is available), else 0. */
int eeg_dcrnn_decoder_is_persistent(const eeg_decoder_dims* d);
int eeg_dcrnn_decoder_fwd(const eeg_decoder_dims* d, const float* targets, const int32_t* teacher,
                          const float* h0, const float* P, const float* const* packs, const float* Wp,
                          const float* bp, const uint64_t* rng_used, float* out, float* saved,
                          float* ws, void* stream);
int eeg_dcrnn_decoder_bwd(const eeg_decoder_dims* d, const int32_t* teacher, const float* P,
                          const float* const* packs, const float* Wp, const float* saved, const float* dOut,
                          const uint64_t* rng_used, float* dh0, float* const* dWg, float* const* dbg,
                          float* const* dWc, float* const* dbc, float* dWp, float* dbp, float* ws, void* stream);

/* Scheduled sampling drawn on the device (model.py:194-200: one `random.random() < teacher_forcing_ratio` per decoder step;
 * utils.py:385-390: ratio = k / (k + exp(batches_seen / k)), k = cl_decay_steps): flags[t] (DEVICE int32[T]) = 1 iff
 * u_t < ratio, u_t = word t%4 of Philox counter offset + t/4 of rng_state (the dropout generator's {seed, offset} pair) / 2^32,
 * evaluated in fp64.  ON THE STREAM: rng_state's offset advances by ceil(T/4) and samples_seen[0] (DEVICE int64, the
 * reference's `step`, train_ssl.py:163,178) by `increment` (the global batch), so every replay of a captured step draws fresh
 * flags against the decayed threshold. */
int eeg_dcrnn_teacher_flags(uint64_t* rng_state, int64_t* samples_seen, int64_t increment, double cl_decay_steps, int T,
                            int32_t* flags, void* stream);

/* Data augmentation drawn on the device (data/dataloader_detection.py:233-256 `_random_reflect`, `_random_scale`, applied per
 * sample at :384-389 in the DataLoader workers).  rng_used = the {seed, offset} pair eeg_dcrnn_rng_take handed out for B
 * counters; clip b uses counter offset + b: word 0's top bit = the reflection coin, word 1 / 2^32 = u, scale = 0.8 + 0.4 u.
 * swap_perm: DEVICE int32[N], the source channel of every node of a reflected clip (data_utils.py:37-62 `get_swap_pairs`).
 * Outputs (DEVICE): flags int32[B]; perm int32[B][N] (swap_perm where flags[b], else the identity) and log_scale float[B]
 * (= log(scale), float) -- the operands of eeg_dcrnn_fft_features; and, when S_out != NULL, the per-clip supports of the
 * distance graph (`_get_combined_graph(swap_nodes)`, :309-333,405-409): S_out[s][b] = flags[b] ? S_reflected[s] : S_plain[s]
 * (S_plain / S_reflected: n_supports x N x N, S_out: n_supports x B x N x N). */
int eeg_dcrnn_augment_draw(const uint64_t* rng_used, int B, int N, const int32_t* swap_perm, int32_t* flags, int32_t* perm,
                           float* log_scale, const float* S_plain, const float* S_reflected, int n_supports, float* S_out,
                           void* stream);

/* utils.last_relevant_pytorch (utils.py:346-357): last[b] = Htop[lengths[b]-1, b]. Htop (T,B,NH). */
int eeg_dcrnn_gather_last(const float* Htop, const int64_t* lengths, int T, int B, int NH,
                          float* last, void* stream);
/* The generator behind the fused dropout masks (nn.Dropout of model.py:191,267): Philox4x32-10, state = device uint64[2]
 * {seed, offset}.  rng_take copies the pair to rng_used (device uint64[2]) and advances the state's offset by `groups` counters --
 * ON THE STREAM, so a replayed HIP graph draws fresh masks every time.  A forward entry point that drops is handed rng_used and
 * keeps element e of its dropped tensor iff word e%4 of counter offset + e/4 under key seed is >= p * 2^32, scaled by 1/(1-p);
 * nothing is stored, the backward entry point recomputes the mask from the same pair. */
int eeg_dcrnn_rng_take(uint64_t* rng_state, uint64_t groups, uint64_t* rng_used, void* stream);
/* model.py:267-270: logits[b][c] = max_n fc(relu(dropout(z[b][n]))); arg[b][c] = maximising node.
 * dropout_p = nn.Dropout's p while the module is training (README.md:83 trains the 4-class model with --dropout 0.5), else 0;
 * dropout_p > 0: rng_used = the pair rng_take handed out for B*N*H/4 counters (NULL allowed otherwise). */
int eeg_dcrnn_cls_head_fwd(const float* z, const float* W, const float* bias, int B, int N, int H,
                           int C, float dropout_p, const uint64_t* rng_used,
                           float* logits, int32_t* arg, void* stream);
int eeg_dcrnn_cls_head_bwd(const float* z, const float* W, const float* dlogits, const int32_t* arg,
                           int B, int N, int H, int C, float dropout_p, const uint64_t* rng_used,
                           float* dz, float* dW, float* dbias, void* stream);

/* Head + criterion + their backward for the optimisation step (model.py:260-270 forward, train.py:203-206,266-272 loss and the
 * head's share of `loss.backward()`), two launches: logits (B,C), arg (B,C) as eeg_dcrnn_cls_head_fwd; loss[0] = mean criterion,
 * kind 0 = nn.BCEWithLogitsLoss (C == 1, targets float[B]), kind 1 = nn.CrossEntropyLoss (targets int64[B]; a label outside
 * 0..C-1 makes the loss NaN); dlogits (B,C) = d loss / d logits; dz (B,N,H) = d loss / d z (the seed of the encoder's backward);
 * dW (C,H), dbias (C) = the gradients of the fc layer (overwritten).  Sums over the batch run in a fixed order.
 * ws: eeg_dcrnn_cls_head_loss_ws_floats(B, H, C) floats. */
size_t eeg_dcrnn_cls_head_loss_ws_floats(int B, int H, int C);
int eeg_dcrnn_cls_head_loss(const float* z, const float* W, const float* bias, const void* targets, int kind, int B, int N,
                            int H, int C, float dropout_p, const uint64_t* rng_used, float* logits, int32_t* arg,
                            float* dlogits, float* dz, float* dW, float* dbias, float* loss, float* ws, void* stream);
/* mask[e] = keep(e) / (1 - p) for e < n: the factors the fused kernels apply for the {seed, offset} pair in rng_used (a forward
 * call's output), materialised -- the parity tests hand them to the oracle. */
int eeg_dcrnn_dropout_mask(const uint64_t* rng_used, size_t n, float dropout_p, float* mask, void* stream);

/* Losses that seed backward (train.py:203-206,266-268), value + gradient in one launch:
 * nn.BCEWithLogitsLoss() on logits (B,) / nn.CrossEntropyLoss() on logits (B,C); loss[0] = mean. */
int eeg_dcrnn_bce_logits(const float* logits, const float* y, int B, float* loss, float* dlogits, void* stream);
int eeg_dcrnn_ce_logits(const float* logits, const int64_t* y, int B, int C, float* loss, float* dlogits,
                        void* stream);
/* utils.compute_regression_loss (utils.py:431-495; train_ssl.py:165-170) on n elements: optional
 * scalar StandardScaler inverse transform v*std + mean of both tensors, mask = (y_true != mask_val),
 * kind 0: masked MAE (loss_fn == 'mae'); kind 1: `masked_mse_loss`, which returns the masked RMSE
 * (what train_ssl.py's "MAE" string actually selects).  loss[0] = value; dpred (nullable) = gradient
 * w.r.t. pred.  ws: eeg_dcrnn_masked_loss_ws_floats() floats. */
size_t eeg_dcrnn_masked_loss_ws_floats(void);
int eeg_dcrnn_masked_loss(const float* pred, const float* y, size_t n, int use_scaler, float mean, float std_,
                          float mask_val, int kind, float* loss, float* dpred, float* ws, void* stream);
/* clip_grad_norm_(max_norm) + torch.optim.Adam(lr, betas, eps, weight_decay = coupled L2) step
 * `step` (1-based) over flat fp32 buffers of n elements (train.py:222-223,273-275).  grads are
 * first multiplied by grad_scale (1/world_size after a summed all-reduce).  ws: 64 floats scratch;
 * norm_out (nullable) receives the pre-clip gradient norm.  Deterministic (fixed-order sums). */
size_t eeg_dcrnn_clip_adam_ws_floats(void);
int eeg_dcrnn_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                        float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, float* ws, float* norm_out, void* stream);

/* The same update with the optimiser's step count and learning rate in DEVICE memory: step_dev[0] (int32, the number of updates
 * applied so far) is incremented on the stream and the bias corrections 1 - beta^step are formed in the kernel; lr_dev[0] is the
 * current learning rate (the host rewrites it between epochs, train.py:224,329 cosine schedule).  With these two the whole
 * optimisation step -- zero_grad ... backward, [all-reduce], clip, Adam -- is capturable as ONE HIP graph. */
int eeg_dcrnn_clip_adam_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                            float max_norm, const float* lr_dev, float beta1, float beta2, float eps,
                            float weight_decay, int32_t* step_dev, float grad_scale, float* ws, float* norm_out,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEG_DCRNN_H */
