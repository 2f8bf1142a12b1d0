// K9 — lm_head projection + cross-entropy without the [rows][V] f32 logits in memory.
//
// Reference: `lm_logits = self.lm_head(sequence_output)` (models/t5.py:72,176-180) followed by
// `CrossEntropyLoss(ignore_index=-100)(lm_logits.view(-1, V), targets.view(-1))` (tasks/mt3_net.py:32-35): at the benchmark
// batch that is a [65536][1536] f32 tensor — 403 MB written by the GEMM, read back by the loss, and kept alive until the
// backward.  Here the rows are processed in chunks: the GEMM of a chunk writes its logits into a workspace that is reused
// (and, at 50-100 MB, stays in the 256 MB Infinity Cache), the loss kernel reads it there, adds the chunk's share of the
// loss and leaves only the bf16 gradient w.r.t. the logits, which the two backward GEMMs of lm_head consume.
// Arithmetic per row is exactly mrmt3_gemm_nt + mrmt3_ce_fwd_bwd (same kernels), so loss and gradients equal the unfused
// path bit for bit; the loss scalar is accumulated in double (atomic adds in arrival order: 1e-16, invisible in the logged float).
#include "common.h"

extern "C" int mrmt3_lmhead_ce_fwd_bwd(const void* dec, int ld_dec, const void* W, int ldw, const int64_t* targets,
                                       const float* denom_dev, double* loss_dev, void* dlogits, int dl_dtype, int rows,
                                       int V, int d, int weighted, int inst_lo, int inst_hi, float grad_scale,
                                       void* workspace, size_t workspace_bytes, int chunk_rows, void* stream) {
  MR_CHECK_ARG(dec && W && targets && denom_dev && loss_dev && workspace, "lmhead_ce: null pointer");
  MR_CHECK_ARG(rows > 0 && V > 0 && d > 0 && chunk_rows > 0, "lmhead_ce: bad sizes");
  MR_CHECK_ARG(workspace_bytes >= (size_t)(chunk_rows < rows ? chunk_rows : rows) * V * sizeof(float),
               "lmhead_ce: workspace smaller than one chunk of logits");
  const size_t dl_elt = dl_dtype == MRMT3_BF16 ? 2 : 4;
  for (int r0 = 0; r0 < rows; r0 += chunk_rows) {
    const int n = rows - r0 < chunk_rows ? rows - r0 : chunk_rows;
    int rc = mrmt3_gemm_nt((const char*)dec + (size_t)r0 * ld_dec * 2, ld_dec, W, ldw, workspace, V, n, V, d, MRMT3_BF16,
                           MRMT3_F32, 0, stream);
    if (rc != MRMT3_OK) return rc;
    rc = mrmt3_ce_fwd_bwd((const float*)workspace, targets + r0, denom_dev, loss_dev,
                          dlogits ? (char*)dlogits + (size_t)r0 * V * dl_elt : nullptr, dl_dtype, n, V, weighted, inst_lo,
                          inst_hi, grad_scale, stream);
    if (rc != MRMT3_OK) return rc;
  }
  return MRMT3_OK;
}
