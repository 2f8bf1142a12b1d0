/* mrmt3_hip.h — C ABI of libmrmt3_hip.so: the MI355X (gfx950) kernels of the MR-MT3 hot path.
 *
 * The reference (gudgud96/MR-MT3) is pure Python: this path has NO FFI in the reference.  The
 * boundary it replaces is the chain of stock PyTorch / HuggingFace / torchaudio ops called from
 *   contrib/spectrograms.py:105-145          compute_spectrogram            -> mrmt3_logmel_fwd
 *   models/t5.py:99-180, 478-702             T5Stack wiring over HF T5Block  -> gemm / norm / attn /
 *                                                                              geglu / embed entry points
 *   tasks/mt3_net.py:32-35                   CrossEntropyLoss(ignore=-100)   -> mrmt3_ce_fwd_bwd
 *   tasks/mt3_net.py:54-68                   AdamW + LambdaLR                -> mrmt3_adamw_step
 *   models/t5.py:251-302, t5_segmem_v2_with_prev.py:226-296  greedy generate -> mrmt3_decoder_*
 * Each entry point cites the reference lines it stands for.  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - Plain C: pointers are raw DEVICE pointers into caller-owned allocations (torch tensors); sizes
 *    and strides are explicit, in ELEMENTS unless a name ends in _bytes.  No torch types.
 *  - dtype codes: MRMT3_F32 (float) or MRMT3_BF16 (raw bfloat16 bits, uint16_t).
 *  - Every call is asynchronous on the hipStream_t passed as `void* stream`, re-entrant across
 *    streams, never synchronises the device and never allocates (graph-capture safe).  The only
 *    library-owned objects are the opaque mrmt3_decoder handles.
 *  - Dropout: every entry point that draws a mask takes (p_drop, seed, step_dev, stream id): the mask is a pure function
 *    of (seed, stream id, element index) salted, when step_dev is not NULL, by the int32 it points to in DEVICE memory —
 *    read by the kernel at run time, so a captured hipGraph draws new masks every optimizer step (mrmt3_adamw_step
 *    increments the counter).  Forward and backward of a site must pass the same four values.
 *  - Return value: 0 = MRMT3_OK, otherwise an error code; mrmt3_last_error() returns a
 *    thread-local message.  Nothing aborts or throws across this ABI.
 */
#ifndef MRMT3_HIP_H
#define MRMT3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRMT3_OK 0
#define MRMT3_ERR_INVALID_ARG 1
#define MRMT3_ERR_HIP 2
#define MRMT3_ERR_UNSUPPORTED 3
#define MRMT3_ERR_COMM 4          /* RCCL missing or a collective call failed */

#define MRMT3_F32 0
#define MRMT3_BF16 1

/* ABI version (major*10000 + minor*100 + patch) and last error text of the calling thread. */
int mrmt3_version(void);
const char* mrmt3_last_error(void);
/* page-locked host memory for tables that reach the device through an async copy (mrmt3_tn_group_run);
 * NULL on failure.  Allocate outside stream capture. */
void* mrmt3_host_alloc(size_t bytes);
void mrmt3_host_free(void* p);

/* Diagnostics: launches per kernel family since the process started (or since the last call with reset != 0).
 * Writes min(n, MRMT3_CNT_N) counters to out and returns MRMT3_CNT_N.  Tests use it to assert that a shape really
 * dispatched to the kernel they mean to check (e.g. the tall-shape ping-pong GEMM needs >= 4096 rows). */
#define MRMT3_CNT_GEMM_NT_TILE 0      /* round-1 256x256 / 256x128 tile kernels (bf16 or exact f32) */
#define MRMT3_CNT_GEMM_NT8 1          /* ping-pong NT kernel */
#define MRMT3_CNT_GEMM_NT_GEGLU 2     /* fused wi + gated-GELU launches (not the two-kernel fallback) */
#define MRMT3_CNT_TN_GROUP 3          /* grouped weight-gradient launches */
#define MRMT3_CNT_TN8 4               /* one-gradient ping-pong TN launches */
#define MRMT3_CNT_TN_TILE 5           /* round-1 128x128 TN kernel */
#define MRMT3_CNT_ATTN_FWD 6          /* bf16 flash forward */
#define MRMT3_CNT_ATTN_BWD 7          /* bf16 flash backward (two-pass: dQ, then dK/dV) */
#define MRMT3_CNT_ATTN_BWD_ONEPASS 8  /* bf16 one-pass backward (keys of a (batch, head) owned by one workgroup) */
#define MRMT3_CNT_ATTN_F32 9          /* exact-f32 attention kernels */
#define MRMT3_CNT_TN_F32 10           /* exact-f32 weight-gradient kernel */
#define MRMT3_CNT_GEMM_NT_SPLITK 11   /* ping-pong NT kernel split over K + reduce (mrmt3_gemm_nt_ws, short inputs) */
#define MRMT3_CNT_GEMM_NT_ADDNORM 12  /* projection + residual add + RMS norm in one launch (mrmt3_gemm_nt_addnorm) */
#define MRMT3_CNT_GEMM_NT_NORMBWD 13  /* data gradient + norm backward in one launch (mrmt3_gemm_nt_normbwd) */
#define MRMT3_CNT_GEMM_NT_GEGLUBWD 14 /* wo data gradient + gated-GELU backward in one launch (mrmt3_gemm_nt_geglubwd) */
#define MRMT3_CNT_N 15
int mrmt3_dispatch_counts(unsigned long long* out, int n, int reset);

/* Dispatch / tuning switches ("knobs").  Which kernel or tile shape a call takes is a function of its arguments; a few
 * MRMT3_* environment variables override that choice for A/B measurements and for the parity tests that hold two
 * implementations against each other (MRMT3_ROWS_BM, MRMT3_GEMM8, MRMT3_TN8, MRMT3_LOGMEL, MRMT3_ATTN_ONEPASS, ...; the
 * sources name each one where it is read).  The library reads a knob from the environment ONCE per process, at the first
 * launch that asks for it; later changes of the environment are not seen and no launch path calls getenv again.
 * mrmt3_set_knob(name, value) overrides a knob in-process from the next launch on (name = the variable's name, value = the
 * integer the variable would hold); mrmt3_reset_knobs() drops every override and re-reads the environment at the next use.
 * Process-wide, for tests and tuning: not to be changed while another thread launches. */
int mrmt3_set_knob(const char* name, int value);
int mrmt3_reset_knobs(void);

/* ---- K1: log-mel frontend ---------------------------------------------------------------------
 * contrib/spectrograms.py:92-103,128-145 (pad_end, MelSpectrogram(n_fft 2048, hop, power 1,
 * center False), safe_log) + dataset/dataset_2_random.py:288-289 (clip/scale when normalize!=0)
 * + inference.py:125-126 (frames >= valid_frames[b] are zeroed; valid_frames may be NULL).
 * audio [batch][n_samples] f32 -> out [batch][ceil(n_samples/hop)][n_mels] f32 (or bf16).
 * window [2048] f32; twiddle [1024][2] f32 = exp(-2*pi*i*k/2048); the filterbank is passed in
 * compressed rows: filter m = sum_{q<fb_cnt[m]} fb_w[m*max_taps+q] * |X[fb_start[m]+q]|, fb_w [n_mels][max_taps];
 * what fb_w holds at taps q >= fb_cnt[m] is never used (no zero padding required), 0 <= fb_start[m] <= 1024.
 * Two kernels compute this: the wave-per-frame kernel (n_mels = 512, max_taps <= 12, even hop, fb_start 16-byte and
 * window 8-byte aligned, out 16-byte aligned — the model's configuration) and the general one for everything else;
 * the choice is by these arguments alone and both honour the contract above. */
int mrmt3_logmel_fwd(const float* audio, int batch, int n_samples, int hop, const float* window,
                     const float* twiddle, const int* fb_start, const int* fb_cnt, const float* fb_w,
                     int n_mels, int max_taps, const int* valid_frames, int normalize, int out_bf16,
                     void* out, void* stream);

/* Same frontend for crops gathered out of ONE long recording (the producer side of
 * dataset/dataset_2_random.py:329-344,281-306: _random_chunk picks mel_length frames of the song,
 * _compute_spectrogram runs on those frames alone, _pad_length zero-pads short rows).  Crop b covers
 * samples [seg_start[b], seg_start[b]+n_samples) of audio[total_samples]; it sees zeros past the end
 * of the recording and past valid_frames[b]*hop (its own frames only, never its neighbour's), and
 * output frames >= valid_frames[b] are zero.  seg_start is int64 on the device. */
int mrmt3_logmel_crops_fwd(const float* audio, long long total_samples, const long long* seg_start,
                           int batch, int n_samples, int hop, const float* window,
                           const float* twiddle, const int* fb_start, const int* fb_cnt,
                           const float* fb_w, int n_mels, int max_taps, const int* valid_frames,
                           int normalize, int out_bf16, void* out, void* stream);

/* ---- K2/K4/K6/K7/K9: bias-free Linear layers (nn.Linear(bias=False) inside HF T5Block,
 * models/t5.py:51,72,487-490) ----------------------------------------------------------------------
 * NT:  C[M,N] (+)= A[M,K] . B[N,K]^T      forward y = x W^T, and dgrad with a pre-transposed W.
 * in_dtype applies to A and B; out_dtype to C; accumulate!=0 adds to the existing C (f32 only).
 * Requirements: K*sizeof(in) % 128 == 0; lda/ldb/ldc are row strides in elements. */
int mrmt3_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N,
                  int K, int in_dtype, int out_dtype, int accumulate, void* stream);
/* The same product with a caller-owned scratch buffer.  Short inputs with a long K (the encoder's 3072 rows at 12
 * segments per GPU: 24-48 tiles for 256 CUs) are cut into 2-4 K ranges that run as tiles of ONE launch — f32 partial sums
 * in the workspace — and are summed in split order by a second kernel (fixed order, no atomics; C bf16 or f32, += for the
 * accumulate form).  mrmt3_gemm_nt_workspace_bytes returns what that takes for a shape, 0 when the shape does not split:
 * then, and with workspace == NULL or too small, the call IS mrmt3_gemm_nt. */
size_t mrmt3_gemm_nt_workspace_bytes(int M, int N, int K, int in_dtype);
int mrmt3_gemm_nt_ws(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     int in_dtype, int out_dtype, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* TN (weight gradient):  C[N1,N2] (+)= A[M,N1]^T . B[M,N2], bf16 inputs, f32 output, reduction over
 * the M rows split across workgroups; `workspace` must hold mrmt3_gemm_tn_workspace_bytes(...). */
size_t mrmt3_gemm_tn_workspace_bytes(int M, int N1, int N2);
int mrmt3_gemm_tn(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1,
                  int N2, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The same product in exact f32 (plain FMAs in row order, one writer per element, no workspace): the weight gradients
 * of the fp32 training / parity path — the reference trains at `precision: 32`
 * (config/config_slakh_segmem.yaml:47); torch autograd's `grad_output.t().mm(input)` of nn.Linear. */
int mrmt3_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2,
                      int accumulate, void* stream);
/* Deferred form for the 45-50 weight gradients of a training step: mrmt3_gemm_tn_partial only leaves the
 * mrmt3_gemm_tn_splits(M,N1,N2) f32 slabs [split][N1][N2] in `slabs` (>= mrmt3_gemm_tn_workspace_bytes, one buffer per
 * site, kept until reduced); mrmt3_tn_reduce_sites then sums the slabs of n_sites sites into their C in ONE launch
 * (same per-element order as mrmt3_gemm_tn: bit-identical).  `sites_dev` is a DEVICE array of mrmt3_tn_site with
 * block0 = running sum of ceil(N1*N2/1024) over the preceding sites; total_blocks = that sum over all sites. */
typedef struct {
  uint64_t slabs;      /* device address of the site's slabs */
  uint64_t C;          /* device address of C [N1][ldc] f32 */
  int32_t N1, N2, ldc, splits, accumulate, block0, pad0, pad1;
} mrmt3_tn_site;
int mrmt3_gemm_tn_splits(int M, int N1, int N2);
int mrmt3_gemm_tn_partial(const void* A, int lda, const void* B, int ldb, int M, int N1, int N2, void* slabs,
                          size_t slab_bytes, void* stream);
int mrmt3_tn_reduce_sites(const void* sites_dev, int n_sites, int total_blocks, void* stream);

/* Grouped form: the weight gradients of several linear layers (one gradient bucket of the backward, or all of it)
 * in ONE MFMA launch + ONE reduce.  Launched one by one a gradient is only 2-16 tiles of 256 x 256, so it has to
 * cut its token rows into 16-32 ranges to occupy the chip and every range leaves a 256-KiB f32 partial tile; together,
 * the items (gradient, token range, tile) are planned on the host so that they fill whole rounds over the CUs with
 * 2-8 ranges per gradient.  The caller keeps A (dY), B (X) alive until the run.  mrmt3_tn_group_plan with
 * table_host == NULL only sizes (info->table_bytes of host+device table, info->slab_bytes of device scratch); with the
 * buffers it writes the table (pointers baked in: re-plan when an address changes), which the caller copies to the
 * device; mrmt3_tn_group_run first copies table_host (page-locked, nullable: table_dev already holds it) to table_dev
 * on `stream` — a memcpy node when the stream is capturing: the replays re-send the same bytes, so table_host must
 * outlive the graph — and launches from the device copy.  Per-element summation order is fixed by the plan
 * (bitwise reproducible for the same plan).  Shapes: mrmt3_tn_group_ok (M >= 1024, N1 % 128 == 0, N2 % 64 == 0,
 * both >= 256, 16-byte aligned rows); everything else goes through mrmt3_gemm_tn. */
typedef struct {
  const void* A;       /* dY [M][lda] bf16 */
  const void* B;       /* X  [M][ldb] bf16 */
  float* C;            /* dW [N1][ldc] f32 */
  int32_t lda, ldb, ldc, M, N1, N2, accumulate, pad;
} mrmt3_tn_gsite;
typedef struct {
  int32_t n_ctas, n_items, n_rtiles, rounds;
  uint64_t rtile_offset, list_offset, sync_offset, table_bytes, slab_bytes;
} mrmt3_tn_group_info;
int mrmt3_tn_group_ok(int M, int N1, int N2, int lda, int ldb, int ldc);
int mrmt3_tn_group_plan(const mrmt3_tn_gsite* sites, int n_sites, void* slab_dev, void* table_host, size_t table_cap,
                        mrmt3_tn_group_info* info);
int mrmt3_tn_group_run(void* table_dev, const void* table_host, const mrmt3_tn_group_info* info, void* stream);

/* ---- K3: T5LayerNorm (RMS norm) fused with the residual add and dropout that precede it --------
 * HF T5LayerNorm + `hidden + dropout(sublayer_out)` (T5LayerSelfAttention/CrossAttention/FF), and
 * models/t5.py:598-601,676-678.
 *   x1 = x0 + dropmask(y)            (y may be NULL: x1 = x0;  x1 may alias x0)
 *   xn = x1 * rsqrt(mean(x1^2)+eps) * w      -> xn (act_dtype), rstd[rows]
 *   out_drop!=0 additionally applies dropout to xn (final_layer_norm + dropout, t5.py:676-678).
 * Dropout masks are regenerated from (seed, stream_id, element index); p==0 disables them. */
int mrmt3_add_rmsnorm_fwd(const float* x0, const void* y, int y_dtype, const float* w, float eps,
                          float* x1, void* xn, int xn_dtype, float* rstd, int rows, int cols,
                          float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out,
                          int out_drop, void* stream);
/* backward of the above:
 *   g    = dxn (f32 or bf16, dxn_dtype) [* out-dropout mask]
 *   dx1  = dres (nullable) + rmsnorm_bwd(g; x1, rstd, w)          -> dx1 (may alias dres when the dtypes agree)
 *          dres / dx1 are f32 or — the residual-gradient stream of the bf16 engine, cols == 512 — bf16
 *   dy   = dropmask_y(dx1) as bf16 (nullable)                      -> dy
 *   dw  += sum_rows g * x1 * rstd      (per-workgroup partials in `workspace`, then a 16-way reduction;
 *                                        workspace >= mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, cols))
 *   dw == NULL with a workspace: only the partial rows are written; mrmt3_norm_dw_reduce sums the partial rows of
 *   several such workspaces (one per norm site, all with the same `cols`) into their dw vectors in ONE launch.
 *   `workspaces` / `dws` are DEVICE arrays of n_sites 64-bit addresses, `partial_rows` a device array of
 *   mrmt3_add_rmsnorm_bwd_partial_rows(rows of that site).  Same fixed summation order as the immediate form. */
size_t mrmt3_add_rmsnorm_bwd_workspace_bytes(int rows, int cols);
int mrmt3_add_rmsnorm_bwd_partial_rows(int rows);
int mrmt3_norm_dw_reduce(const void* workspaces, const void* dws, const int* partial_rows, int n_sites, int cols,
                         void* stream);
int mrmt3_add_rmsnorm_bwd(const void* dxn, int dxn_dtype, const void* dres, int dres_dtype, const float* x1,
                          const float* rstd, const float* w, void* dx1, int dx1_dtype, void* dy_bf16,
                          float* dw, int rows, int cols,
                          float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out,
                          int out_drop, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K5: attention core (HF T5Attention without relative bias: softmax(q k^T [+causal]) v, scale
 * 1.0, fp32 softmax; models/t5.py:487-490,636-648) -------------------------------------------------
 * q [B][Lq][*] , k/v [B][Lk][*] with head h at columns [h*64, h*64+64) of the row (row strides
 * ldq/ldk/ldv/ldo in elements; batch strides = L*ld).  head_dim is fixed at 64 (d_kv).
 * o [B][Lq][H*64] (ldo), lse [B][H][Lq] f32.  o_lo (nullable, bf16 kernel only, same layout and stride as o) receives
 * bf16(O - bf16(O)), the low half of the f32 output: hand it to mrmt3_attn_bwd and delta = rowsum(dO*O) is formed from
 * O to ~16 bits, which keeps dS = P*(dP - delta) accurate when the value rows share a large common component.
 * causal!=0 masks key > query.  Attention-probability
 * dropout (p_drop) uses a counter hash of (seed, b, h, q, k >> 2), a byte per key.  dtype = MRMT3_BF16 (MFMA flash
 * kernel) or MRMT3_F32 (exact-f32 kernel: fp32 training, parity, decoding prefill; same masks). */
int mrmt3_attn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o,
                   int ldo, void* o_lo, float* lse, int B, int H, int Lq, int Lk, int causal, int dtype,
                   float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);
/* backward of the bf16 kernel: delta [B][H][Lq] f32 scratch is written by the call.
 * dq/dk/dv share the layout (and strides) of q/k/v. */
int mrmt3_attn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                   const void* o, int ldo, const void* o_lo, const void* d_o, int lddo, const float* lse, float* delta,
                   void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int H, int Lq,
                   int Lk, int causal, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);
/* backward of the exact-f32 kernel (autograd of HF T5Attention at `precision: 32`): everything f32, the same masks as
 * the f32 forward; one workgroup per query row (delta, dQ) and per key row (dK, dV), fixed summation order. */
int mrmt3_attn_bwd_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                       int ldo, const float* d_o, int lddo, const float* lse, float* delta, float* dq, int lddq,
                       float* dk, int lddk, float* dv, int lddv, int B, int H, int Lq, int Lk, int causal,
                       float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);

/* The same attention with HF T5Attention's ADDITIVE BIAS (`scores += position_bias`, models/t5.py:636-648: the
 * relative-position bias of stock T5 — SURVEY §8b `attn(q, k, v, bias_or_null, ...)`; MR-MT3's own position_bias is all
 * zeros, models/t5.py:487-490, so the training step never calls these two).  bias: f32 [H][Lq][Lk] shared by the batch
 * (bias_batch_stride = 0) or [B][H][Lq][Lk] (bias_batch_stride = H*Lq*Lk, e.g. a padding mask of -inf / 0 per
 * sequence); NULL = no bias.  dtype = MRMT3_F32 or MRMT3_BF16 operands (q, k, v, o, d_o, dq, dk, dv), f32 arithmetic:
 * this is the general (parity) kernel, one workgroup per query / key row — correct for any bias, not the MFMA fast path.
 * dbias (nullable, layout of bias) receives dS = P (dP - delta); for a shared bias summed over the batch in batch
 * order.  Dropout as above. */
int mrmt3_attn_fwd_bias(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* bias,
                        long long bias_batch_stride, void* o, int ldo, float* lse, int B, int H, int Lq, int Lk,
                        int causal, int dtype, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id,
                        void* stream);
int mrmt3_attn_bwd_bias(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* o, int ldo,
                        const void* d_o, int lddo, const float* lse, float* delta, const float* bias,
                        long long bias_batch_stride, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                        float* dbias, int B, int H, int Lq, int Lk, int causal, int dtype, float p_drop, uint64_t seed,
                        const int32_t* step_dev, uint32_t stream_id, void* stream);

/* ---- K7: gated-GELU (HF T5DenseGatedGeluDense: gelu_new(h0) * h1, then dropout) ----------------
 * h [rows][2*dff] = [wi_0 x | wi_1 x] -> g [rows][dff] */
int mrmt3_geglu_fwd(const void* h, void* g, int rows, int dff, int dtype, float p_drop,
                    uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);
/* h, dg, dh all of `dtype` (bf16 or f32) */
int mrmt3_geglu_bwd(const void* h, const void* dg, void* dh, int rows, int dff, int dtype, float p_drop,
                    uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);

/* K2 + K7 in one launch (the feed-forward input projection, models/t5.py T5DenseGatedGeluDense.forward):
 * h [rows][ldh >= 2 dff] = x [rows][K] . wi [2 dff][K]^T (bf16, kept for the backward) and
 * g [rows][ldg >= dff] = dropout(gelu_new(h[:, :dff]) * h[:, dff:]).  Bit-identical to mrmt3_gemm_nt followed by
 * mrmt3_geglu_fwd (which is what runs when the shape is outside the fused kernel's: rows < 4096, dff or K not a
 * multiple of 128).  bf16 only. */
int mrmt3_gemm_nt_geglu(const void* x, int ldx, const void* wi, int ldw, void* h, int ldh, void* g, int ldg,
                        int rows, int dff, int K, float p_drop, uint64_t seed, const int32_t* step_dev,
                        uint32_t stream_id, void* stream);

/* ---- A projection and the row kernel behind it in ONE launch (csrc/gemm_rows.hip) ----------------------------------
 * The product's 64-row x 512-column tiles stay on chip (LDS) and the workgroup that computed them runs the row
 * kernel's arithmetic on them: the product's output never makes its HBM round trip.  bf16 operands, model width 512.
 * mrmt3_gemm_rows_ok(M, N, K, lda, ldw): 1 when the fused kernels take the shape (N = 512, or 1024 for the GEGLU
 * backward; K a multiple of 128; row strides multiples of 8 elements; buffers below 2 GiB) — otherwise call the two
 * kernels.
 *
 * mrmt3_gemm_nt_addnorm = mrmt3_gemm_nt (y = A . W^T, [rows][512] bf16) + mrmt3_add_rmsnorm_fwd(x0, y, ...):
 *   x1 = x0 + dropout(y) (f32, may be x0 itself or NULL), xn = w_norm * x1 * rsqrt(mean(x1^2) + eps) (bf16, dropped by
 *   stream_out when out_drop != 0), rstd [rows] (nullable).  HF T5LayerSelfAttention / T5LayerCrossAttention / T5LayerFF:
 *   `hidden + dropout(sublayer(...))` followed by the next sublayer's T5LayerNorm (models/t5.py:636-648).  Same bits as
 *   the two kernels.
 * mrmt3_gemm_nt_normbwd = mrmt3_gemm_nt (dxn = A . WT^T, [rows][512] bf16) + mrmt3_add_rmsnorm_bwd(dxn, dres, ...) with
 *   out_drop = 0: dx1 (f32 or bf16, may be dres itself), dy_bf16 (masked by stream_y, nullable) and — when `workspace`
 *   is given (mrmt3_add_rmsnorm_bwd_workspace_bytes(rows, 512) bytes suffice) — the norm-weight gradient's partial rows,
 *   mrmt3_gemm_nt_normbwd_partial_rows(rows) of them (one per tile of rows), left for mrmt3_norm_dw_reduce.
 * Tile height: 64 rows (two workgroups per CU) until ceil(rows / 128) reaches the CU count, 128 rows (one workgroup per
 * CU, half the weight bytes staged per row) from there on — a function of `rows` alone, so the partial-row count is too.
 * The knob MRMT3_ROWS_BM = 64 / 128 forces one height (parity tests run every case under both).
 * mrmt3_gemm_nt_geglubwd = mrmt3_gemm_nt (dg = dy . WT^T, [rows][dff] bf16, WT = wo^T [dff][K]) + mrmt3_geglu_bwd(h, dg):
 *   dh [rows][2 dff] bf16.  Same bits as the two kernels. */
int mrmt3_gemm_rows_ok(int M, int N, int K, int lda, int ldw);
int mrmt3_gemm_nt_addnorm(const void* A, int lda, const void* W, int ldw, int rows, int K, const float* x0,
                          const float* w_norm, float eps, float* x1, void* xn_bf16, float* rstd, float p_drop,
                          uint64_t seed, const int32_t* step_dev, uint32_t stream_y, uint32_t stream_out, int out_drop,
                          void* stream);
int mrmt3_gemm_nt_normbwd_partial_rows(int rows);
int mrmt3_gemm_nt_normbwd(const void* A, int lda, const void* WT, int ldw, int rows, int K, const void* dres,
                          int dres_dtype, const float* x1, const float* rstd, const float* w_norm, void* dx1,
                          int dx1_dtype, void* dy_bf16, float p_drop, uint64_t seed, const int32_t* step_dev,
                          uint32_t stream_y, void* workspace, size_t workspace_bytes, void* stream);
int mrmt3_gemm_nt_geglubwd(const void* dy, int ldy, const void* WT, int ldw, const void* h, void* dh, int rows, int dff,
                           int K, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);

/* ---- K8: embedding gather + sinusoid add (+dropout) and its scatter-add backward ---------------
 * models/t5.py:539-540,596-601 and `_shift_right` (t5.py:148-150).
 * ids [rows] int64.  shift!=0 applies HF _shift_right inside the gather: position t of a
 * sequence of length seq_len uses ids[t-1] (start_id at t==0) and maps -100 -> pad_id.
 * x [rows][d] f32 = table[id] + pos[(row % seq_len) + pos_offset], then dropout.
 * pos may be NULL (plain gather, used for the segment-memory ids). */
int mrmt3_embed_fwd(const int64_t* ids, const float* table, const float* pos, float* x, int rows,
                    int seq_len, int d, int vocab, int shift, int start_id, int pad_id,
                    int pos_offset, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id, void* stream);
/* dtable[id] += dropmask(dx[row]).  No atomics: rows are ranked by id (counting sort), summed in sorted order and
 * every table row has one writer, so the result is bitwise reproducible and insensitive to how skewed the ids are.
 * workspace >= mrmt3_embed_bwd_workspace_bytes(rows, vocab, d); vocab <= 16384. */
size_t mrmt3_embed_bwd_workspace_bytes(int rows, int vocab, int d);
int mrmt3_embed_bwd(const int64_t* ids, const float* dx, float* dtable, int rows, int seq_len, int d,
                    int vocab, int shift, int start_id, int pad_id, float p_drop, uint64_t seed, const int32_t* step_dev,
                    uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream);
/* x[rows][d] f32 = src[rows][d] (src_dtype) + pos[(row % seq_len)+pos_offset], then dropout
 * (the encoder side of models/t5.py:596-601, input = proj(mel)); backward is dropmask only. */
int mrmt3_addpos_fwd(const void* src, int src_dtype, const float* pos, float* x, int rows, int seq_len,
                     int d, int pos_offset, float p_drop, uint64_t seed, const int32_t* step_dev, uint32_t stream_id,
                     void* stream);
/* out (bf16 or f32) = dropmask(dx)  (gradient w.r.t. the GEMM output feeding addpos / any dropout site) */
int mrmt3_dropmask_cast(const float* dx, void* out, int out_dtype, size_t n, float p_drop, uint64_t seed,
                        const int32_t* step_dev, uint32_t stream_id, void* stream);

/* ---- K9: cross-entropy over lm_head logits (tasks/mt3_net.py:32-35; weighted variant :96-108) ---
 * logits [rows][V] f32, targets [rows] int64 (ignore_index -100).  Per row (w_i, n_i) = (1,1) for a
 * normal target, (3,2) for an instrument token in [inst_lo,inst_hi] when weighted!=0 (the
 * reference's `sum_nonpad + 2*sum_inst` over `n_inst + n_nonpad`), (0,0) for ignored rows.
 * mrmt3_ce_count:   denom_dev[0] += sum_i n_i                      (zero it first)
 * mrmt3_ce_fwd_bwd: loss_dev[0]  += sum_i w_i * nll_i / denom      (zero it first; a DOUBLE: one atomic add per
 *                   workgroup in arrival order perturbs it at 1e-16, so the logged float is the same run after run)
 *                   dlogits (nullable, dl_dtype) = grad_scale * w_i * (softmax_i - onehot_i) / denom
 * Both scalars stay on the device: no host synchronisation. */
int mrmt3_ce_count(const int64_t* targets, int rows, int weighted, int inst_lo, int inst_hi,
                   float* denom_dev, void* stream);
int mrmt3_ce_fwd_bwd(const float* logits, const int64_t* targets, const float* denom_dev,
                     double* loss_dev, void* dlogits, int dl_dtype, int rows, int V, int weighted,
                     int inst_lo, int inst_hi, float grad_scale, void* stream);

/* lm_head + cross-entropy without the [rows][V] f32 logits in memory (models/t5.py:72,176-180 followed by
 * tasks/mt3_net.py:32-35).  dec [rows][ld_dec] bf16 = the final-normed decoder states, W [V][ldw] bf16 = lm_head.weight.
 * Rows are processed in chunks of chunk_rows: logits of a chunk = dec_chunk . W^T go (f32) into `workspace`
 * (>= min(rows, chunk_rows)*V*4 bytes, reused by every chunk), then exactly mrmt3_ce_fwd_bwd on that chunk: loss_dev[0]
 * accumulates, dlogits [rows][V] (dl_dtype; nullable: loss only) receives the gradient.  denom_dev as for
 * mrmt3_ce_fwd_bwd (run mrmt3_ce_count over ALL targets first; zero loss_dev first). */
int mrmt3_lmhead_ce_fwd_bwd(const void* dec, int ld_dec, const void* W, int ldw, const int64_t* targets,
                            const float* denom_dev, double* loss_dev, void* dlogits, int dl_dtype, int rows, int V,
                            int d, int weighted, int inst_lo, int inst_hi, float grad_scale, void* workspace,
                            size_t workspace_bytes, int chunk_rows, void* stream);

/* ---- K11: AdamW over the flat parameter buffer (torch.optim.AdamW defaults; tasks/mt3_net.py:55)
 * p,g,m,v: flat f32 [n].  lr is read from lr_dev[0] (device), step count from step_dev[0] (int32,
 * incremented by the call).  grad_scale multiplies g (e.g. 1/world_size).  shadow_bf16 (nullable)
 * receives the updated parameters in bf16. */
int mrmt3_adamw_step(float* p, const float* g, float* m, float* v, size_t n, const float* lr_dev,
                     int32_t* step_dev, float beta1, float beta2, float eps, float weight_decay,
                     float grad_scale, void* shadow_bf16, void* stream);
/* out[c][r] = in[r][c] (2-D transpose, with optional f32->bf16 cast) for the pre-transposed dgrad
 * weights. */
int mrmt3_transpose(const void* in, int in_dtype, void* out, int out_dtype, int rows, int cols,
                    void* stream);
int mrmt3_cast(const void* in, int in_dtype, void* out, int out_dtype, size_t n, void* stream);
/* One launch transposes n_mats bf16 matrices living in two flat buffers.  desc_table: device array of
 * {int64 src_off, int64 dst_off, int32 rows, int32 cols} (element offsets; dst is [cols][rows]);
 * tile_start: device int32 prefix sums of ceil(rows/64)*ceil(cols/64), length n_mats. */
int mrmt3_transpose_batched(const void* src_bf16, void* dst_bf16, const void* desc_table,
                            const int* tile_start, int n_mats, int total_tiles, void* stream);

/* ---- K12: greedy decode with a KV cache, one hipGraph replay per token -------------------------
 * models/t5.py:251-302 (batched, MT3Net) and models/t5_segmem_v2_with_prev.py:273-291 (B=1 per
 * segment).  The decoder handle owns: the captured graph, the self-attention KV cache
 * [layers][2][B][max_len][H*64], scratch activations and the device-side step/finished state.
 * Weights are referenced, not copied: `weights` is an array of mrmt3_decoder_weights (device
 * pointers into the caller's flat parameter buffer, dtype = act dtype of the handle).
 * max_batch <= 256 sequences per handle (d_model 512, heads*64 <= 512, d_ff <= 1024). */
typedef struct mrmt3_decoder mrmt3_decoder;
typedef struct {
  const void* embed;        /* [vocab][d] f32 decoder_embed_tokens */
  const float* pos;         /* [>=max_len][d] f32 sinusoid table */
  const void* lm_head;      /* [vocab][d] */
  const float* final_ln;    /* [d] */
  /* per layer l (arrays of n_layers pointers on the HOST): */
  const float* const* ln_self;   /* [d] */
  const void* const* w_qkv;      /* [3*inner][d]  (q|k|v rows) */
  const void* const* w_o_self;   /* [d][inner] */
  const float* const* ln_cross;  /* [d] */
  const void* const* w_q_cross;  /* [inner][d] */
  const void* const* w_o_cross;  /* [d][inner] */
  const float* const* ln_ff;     /* [d] */
  const void* const* w_wi;       /* [2*dff][d]   (wi_0|wi_1 rows) */
  const void* const* w_wo;       /* [d][dff] */
} mrmt3_decoder_weights;

int mrmt3_decoder_create(mrmt3_decoder** out, int n_layers, int d_model, int n_heads, int d_ff,
                         int vocab, int max_batch, int max_len, int max_enc_len, int w_dtype,
                         float eps);
void mrmt3_decoder_destroy(mrmt3_decoder* dec);
/* Start a batch: cross_kv [layers][B*enc_len][2*inner] (k | v columns, dtype = w_dtype) is
 * caller-owned (computed with mrmt3_gemm_nt from the encoder output [+ segment memory]); resets the
 * step counter and finished flags and writes start_id as token 0.  tokens_out [B][max_len+1] int64 is
 * caller-owned.  A row that has emitted eos_id keeps emitting pad_id (models/t5.py:288). */
int mrmt3_decoder_begin(mrmt3_decoder* dec, const mrmt3_decoder_weights* w, const void* cross_kv,
                        int batch, int enc_len, int64_t* tokens_out, int start_id, int eos_id,
                        int pad_id, void* stream);
/* Optional, after mrmt3_decoder_begin: feed n_prefix caller-owned f32 rows per batch row
 * (prefix [batch][n_prefix][d_model], WITHOUT the positional term) as decoder positions
 * 0..n_prefix-1 before the start token, which moves to position n_prefix.  This is
 * T5SegMem.generate_2's memory-prefixed decoder input (models/t5_segmem.py:198-213): the first
 * n_prefix steps of mrmt3_decoder_run only fill the self-attention cache, token steps follow.
 * Needs n_prefix + token steps <= max_len.  mrmt3_decoder_begin clears the prefix. */
int mrmt3_decoder_set_prefix(mrmt3_decoder* dec, const float* prefix, int n_prefix, void* stream);
/* Run n_steps decode steps (graph replays; captured on first use).  No host synchronisation. */
int mrmt3_decoder_run(mrmt3_decoder* dec, int n_steps, void* stream);
/* 1 if the current configuration is being replayed from a captured hipGraph (0 = plain launches). */
int mrmt3_decoder_graph_captured(const mrmt3_decoder* dec);
/* state_out[0] = steps taken so far, [1] = 1 if every row has emitted EOS, [2] = step index at
 * which the last row finished (or -1); [0] counts prefix positions too, [2] counts token steps only.  Copies 3 int32 asynchronously to caller-owned PINNED host
 * memory; the caller synchronises the stream before reading. */
int mrmt3_decoder_poll(mrmt3_decoder* dec, int32_t* state_out_pinned, void* stream);

/* ---- gradient exchange (data parallel, one process per GPU): RCCL communicators as opaque handles ----------------
 * Replaces what the reference gets from Lightning's `ddp_find_unused_parameters_false` strategy (config/config.yaml:45,
 * train.sh:6): torch DDP's bucketed NCCL all-reduce of the gradients.  The flat f32 gradient buffer is exchanged as a few
 * contiguous buckets (mrmt3/ddp.py), each with one in-place all-reduce on a stream of the caller's choosing.
 * RCCL is resolved at first use (dlopen: $MRMT3_RCCL_LIB, an already mapped librccl, the loader's path, /opt/rocm/lib);
 * the library itself does not link against it.
 *   mrmt3_comm_unique_id : rank 0 fills id_out[MRMT3_COMM_ID_BYTES] and hands the bytes to every other rank (any channel).
 *   mrmt3_comm_create    : every rank, same id; blocks until all `world` ranks have called it.  The GPU the communicator
 *                          is bound to is the calling thread's current HIP device.
 *   mrmt3_allreduce      : buf[0..count) (dtype MRMT3_F32 or MRMT3_BF16, device memory) = sum over ranks, or the mean when
 *                          average != 0; in place, asynchronous on `stream`; every rank must call it with the same count,
 *                          dtype and order of calls.
 *   mrmt3_comm_destroy   : releases the handle (NULL is accepted). */
#define MRMT3_COMM_ID_BYTES 128
int mrmt3_comm_unique_id(void* id_out);
int mrmt3_comm_create(const void* id, int rank, int world, void** comm_out);
int mrmt3_comm_destroy(void* comm);
int mrmt3_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream);

/* Counting hand-offs between two streams whose work is replayed as two separate hipGraphs (the data-parallel step with its
 * all-reduces captured: one graph of compute, one of collectives, mrmt3/trainer.py).  flag / seen / err: int32 in device
 * memory, zero before first use.
 *   mrmt3_flag_signal : *flag += 1 once everything enqueued before it on `stream` has completed (release, device scope).
 *   mrmt3_flag_wait   : `stream` proceeds once *flag >= *seen + 1, then *seen += 1 (only waits on this stream touch seen).
 *                       After timeout_ms without the signal it sets *err = 1 and lets the stream go on: a step whose other
 *                       graph never ran ends in an error word for the host to read, not in a GPU that spins for ever.
 * Both capture into a graph as ordinary kernel nodes; unlike an event-wait node, a wait cannot be satisfied by the previous
 * replay's signal. */
int mrmt3_flag_signal(int32_t* flag, void* stream);
int mrmt3_flag_wait(const int32_t* flag, int32_t* seen, int32_t* err, int timeout_ms, void* stream);

/* ---- capture hygiene (host code that records the training step into hipGraphs: mrmt3/trainer.py) -----------------------
 * The reference's training loop (train.py:99-103, Lightning) has no graph capture; these exist because this path replays the
 * step, and a capture that fails must leave the process able to go on with plain launches (the library never aborts).
 *   mrmt3_stream_capture_status  : 0 = `stream` is not capturing, 1 = capturing, 2 = its capture was invalidated; < 0 = error.
 *   mrmt3_stream_abandon_capture : if `stream` is (still) in capture mode, ends that capture and destroys whatever graph came
 *                                  out of it; clears the calling thread's HIP error slot.  Returns the status BEFORE the call.
 *   mrmt3_runtime_error_pop      : returns (and clears) the calling thread's pending HIP runtime error code — 0 = none — and
 *                                  writes its name to text[0..n).  A launch wrapper of this library reports whatever is in
 *                                  that slot as ITS launch failure, so a host that has just survived a failed HIP call of
 *                                  its own (a refused capture, an RCCL error) empties the slot before launching again.
 *   mrmt3_stream_create / _destroy : a non-blocking stream of the caller's own (priority as hipStreamCreateWithPriority),
 *                                  outside every framework's stream pool: torch's `Stream()` comes round robin out of a pool of 32
 *                                  per priority, and ROCm 7.2 never takes an INVALIDATED stream out of capture mode, so a stream
 *                                  poisoned by one failed capture would come back to a later one.  Destroy only when nothing
 *                                  enqueued on it is pending and no graph is being captured on it.
 *   mrmt3_abort_trace_install    : opt-in diagnostics — on SIGABRT / SIGSEGV write the native frames of the faulting thread to
 *                                  `path` (NULL or "": stderr), then hand the signal to the previous handler. */
int mrmt3_stream_capture_status(void* stream);
int mrmt3_stream_abandon_capture(void* stream);
int mrmt3_runtime_error_pop(char* text, int n);
int mrmt3_stream_create(void** stream_out, int priority);
int mrmt3_stream_destroy(void* stream);
int mrmt3_abort_trace_install(const char* path);

#ifdef __cplusplus
}
#endif
#endif /* MRMT3_HIP_H */
