/*
 * ggpm_hip.h -- C ABI of the MI355X (gfx950) hierarchical message-passing library.
 *
 * The reference (quocdat32461997/ggpm) is pure Python/PyTorch and has NO native
 * interface for this path; the seam is Python class substitution.
 * Each entry point below therefore cites the reference *Python* lines whose
 * arithmetic it replaces.  All paths are relative to the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (the host side uses the
 *     PyTorch caching allocator); the library allocates nothing and keeps no state;
 *   - fp32 row-major matrices with an explicit leading dimension (in floats);
 *     index arrays are int32 (CSR) or the int64 padded tensors MolGraph.tensorize
 *     delivers (ggpm/mol_graph.py:199-281, ggpm/nnutils.py:210-214);
 *   - "E1"/"N1" row counts INCLUDE the pad row 0 of the reference layout;
 *   - feature matrices use a padded row stride Hp = ggpm_padded_hidden(H)
 *     (multiple of 16); pad columns are zero on input and kept zero on output;
 *   - work is enqueued on `stream` (a hipStream_t) and never synchronised;
 *   - return value 0 = ok, otherwise a GGPM_ERR_* code (ggpm_error_string()).
 *     Nothing aborts; the Python wrappers raise RuntimeError.
 *   - re-entrant: no globals, safe from autograd's worker threads.
 */
#ifndef GGPM_HIP_H
#define GGPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ggpm_stream_t; /* hipStream_t */

enum {
    GGPM_OK = 0,
    GGPM_ERR_ARG = 1,         /* bad size / null pointer / misaligned leading dimension */
    GGPM_ERR_LAUNCH = 2,      /* hipGetLastError() != hipSuccess after a launch */
    GGPM_ERR_UNSUPPORTED = 3, /* shape outside what the kernels were built for */
    GGPM_ERR_WORKSPACE = 4    /* workspace too small */
};

enum { GGPM_ACT_NONE = 0, GGPM_ACT_RELU = 1, GGPM_ACT_TANH = 2, GGPM_ACT_SIGMOID = 3 };

int ggpm_version(void);
const char* ggpm_error_string(int code);
/* Padded feature stride used by the message kernels: H rounded up to a multiple of 16. */
int ggpm_padded_hidden(int H);

/* ------------------------------------------------------------------ graph layout
 * A0 (ggpm/mol_graph.py:238-281, create_pad_tensor ggpm/nnutils.py:105-110): agraph/bgraph/cgraph
 * arrive as zero-padded int64 [rows, width]; entry 0 means "no neighbour" (row 0 is the pad row, so
 * index 0 always gathers zeros in the reference).  These build the equivalent CSR over the real
 * entries, which is what makes the padded gather of index_select_ND (ggpm/nnutils.py:65-70) unnecessary.
 */
/* rowptr[rows+1], col[>= rows*width]; entries keep their in-row order. */
int ggpm_padded_to_csr(const int64_t* padded, int rows, int width, int32_t* rowptr, int32_t* col,
                       ggpm_stream_t stream);
/* Transpose of a CSR whose column ids lie in [0, ncols): rowptrT[ncols+1], colT[>= nnz capacity]
 * hold, for every column id, the ascending list of rows that reference it (deterministic order);
 * `cursor` is int32 scratch of ncols entries. Used for the atomics-free backward gathers. */
int ggpm_csr_transpose(const int32_t* rowptr, const int32_t* col, int rows, int ncols,
                       int32_t* rowptrT, int32_t* colT, int32_t* cursor, ggpm_stream_t stream);
/* out[r] = (int32) mat[r*width + column]   (e.g. fmess[:,0], fnode[:,1]) */
int ggpm_extract_column(const int64_t* mat, int rows, int width, int column, int32_t* out,
                        ggpm_stream_t stream);

/* ------------------------------------------------------------------ dense algebra (fp32 MFMA)
 * C[m,n] = act( sum_k A'(m,k) B'(k,n) + bias[n] + (accumulate ? C[m,n] : 0) ),
 *   A'(m,k) = trans_a ? A[k*lda+m] : A[m*lda+k],   B'(k,n) = trans_b ? B[n*ldb+k] : B[k*ldb+n].
 * Columns N..n_pad-1 of C are zero-filled (n_pad <= ldc; pass n_pad = N for none); zero_row0 forces
 * row 0 of C to zero (the reference's "first node/message is padding" masks, ggpm/encoder.py:36-38).
 * Replaces nn.Linear on this path (ggpm/rnn.py:13-16,69-72, ggpm/encoder.py:15-19,62-82) and autograd's
 * mm/addmm for its gradients.  `splitk_ws` (may be NULL) enables a deterministic split-K reduction for
 * tall contractions (weight gradients over depth*E rows); it needs ggpm_gemm_workspace_bytes() bytes.
 */
size_t ggpm_gemm_workspace_bytes(int M, int N, int K);
int ggpm_gemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B,
              int ldb, float* C, int ldc, int n_pad, const float* bias, int accumulate, int act,
              int zero_row0, float* splitk_ws, size_t splitk_ws_bytes, ggpm_stream_t stream);
/* Several products in ONE launch (the motif / attachment levels have a few hundred rows, so one product fills a
 * fraction of the chip and is bound by launch latency):
 *   ggpm_gemm_grouped   -- `count` (<= 4) independent products of the same (M, N, K, transposes), e.g. the gate input
 *                          projections x W_z^T, x W_r^T, x W_h^T of GRU.forward (ggpm/rnn.py:25-39; LSTM: rnn.py:85-94);
 *   ggpm_gemm_ksegments -- C = act(sum_s A_s B_s' + bias (+ C)) over `nseg` (<= 4) K segments in one accumulator chain:
 *                          A_s is [M x K_s] row-major, B_s is [K_s x N] (trans_b = 0) or [N x K_s] (trans_b = 1), each
 *                          with its own leading dimension -- nn.Linear over a torch.cat of inputs without the cat
 *                          (ggpm/encoder.py:31-35,62-82), and its input gradient summed over the gate slabs.
 * Both fall back to a sequence of ggpm_gemm calls when an operand does not allow 16-byte loads.
 */
typedef struct ggpm_gemm_problem {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc; int n_pad;
    const float* bias;
    int accumulate, act, zero_row0;
} ggpm_gemm_problem;
int ggpm_gemm_grouped(int trans_a, int trans_b, int M, int N, int K, int count, const ggpm_gemm_problem* problems,
                      ggpm_stream_t stream);
/* ggpm_gemm_grouped with the K range cut into chunks (deterministic: partial tiles to `ws`, summed in fixed order by a
 * second launch that applies bias / accumulate / activation): for groups with few output tiles and a long K -- the input
 * halves dW[:, :I] = dX^T x of a level's gate weights (autograd's mm backward of the nn.Linear layers of ggpm/rnn.py:13-16,
 * 69-72 over all E messages), whose 60 tiles would otherwise walk 2 848 rows each.  `ws` needs
 * ggpm_gemm_grouped_splitk_workspace_bytes() bytes; when that is 0 (nothing to split) or `ws` is NULL / too small the call
 * is ggpm_gemm_grouped. */
size_t ggpm_gemm_grouped_splitk_workspace_bytes(int M, int N, int K, int count);
int ggpm_gemm_grouped_splitk(int trans_a, int trans_b, int M, int N, int K, int count, const ggpm_gemm_problem* problems,
                             float* ws, size_t ws_bytes, ggpm_stream_t stream);
int ggpm_gemm_ksegments(int trans_b, int M, int N, int nseg, const float* const* A, const int* lda,
                        const float* const* B, const int* ldb, const int* K, float* C, int ldc, int n_pad,
                        const float* bias, int accumulate, int act, int zero_row0, ggpm_stream_t stream);
/* C[M x N] = rne_bf16(A)^T rne_bf16(B), fp32 accumulate (v_mfma_f32_16x16x32_bf16), A [K x lda] and B [K x ldb] row-major
 * fp32: the tall weight-gradient contraction dW = dY^T X over K = depth*E stash rows -- autograd's mm backward of the
 * hidden x hidden gate products of GRU.GRU / LSTM.LSTM (ggpm/rnn.py:27-36, 88-91) -- with bf16 operands, as the level
 * calls use it under gate_dtype = bf16 (BASELINE configs[4]).  `ws`: ggpm_gemm_workspace_bytes(M, N, K) bytes (split-K
 * slabs, summed in fixed order).  Shapes the tall kernel does not take (K < 6144, a 160 x 160 tiling that would be
 * mostly padding, operands not 16-byte aligned) are computed on fp32 operands by ggpm_gemm instead. */
int ggpm_gemm_tn_bf16(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* ws,
                      size_t ws_bytes, ggpm_stream_t stream);
/* 1 when a contraction of this shape runs on bf16 operands (host-only query; 16-byte aligned operands assumed). */
int ggpm_gemm_tn_bf16_applies(int M, int N, int K);
/* A batch of Linear weight gradients in one call: item i forms dW_i[N x K] = dpre_i[M x N]^T x_i[M x K] (ggpm_gemm(1, 0, N, K, M,
 * ...), split-K over the M rows through `ws` where that pays) and, when `db` is not NULL, db_i[N] = column sums of dpre_i -- autograd's
 * mm / sum backward of a Linear (the score heads of ggpm/decoder.py:35-58, the read-outs of ggpm/encoder.py:15-19,62-72), which the
 * deferred-gradient queue of ggpm_amd/functional.py forms once per parameter and pass.  The items run one after the other on
 * `stream` and share `ws` (>= the largest ggpm_gemm_workspace_bytes(N, K, M); NULL / 0: no split-K) and `csws` (256 * max N
 * floats).  Same launches and results as the separate calls. */
typedef struct ggpm_wgrad_item {
    const float* dpre; int ld_dpre;
    const float* x; int ld_x;
    float* dW; int ld_dw;
    float* db;
    int M, N, K;
} ggpm_wgrad_item;
int ggpm_linear_wgrads_batch(int count, const ggpm_wgrad_item* items, float* ws, size_t ws_bytes, float* csws,
                             ggpm_stream_t stream);
/* out[n] = sum_m A[m*lda+n] (bias gradients), deterministic two-stage; ws >= 256*N floats. */
int ggpm_colsum(const float* A, int lda, int M, int N, float* out, float* ws, ggpm_stream_t stream);
/* dpre = dy * act'(y) given the activation OUTPUT y; optional row-0 zeroing. In-place allowed. */
int ggpm_act_backward(const float* dy, const float* y, int rows, int cols, int ld, int act,
                      int zero_row0, float* dpre, ggpm_stream_t stream);

/* ------------------------------------------------------------------ gathers / segmented sums
 * out[r, 0:width] = sum_{j in rowptr[r]..rowptr[r+1]} src[col[j], 0:width]
 * = index_select_ND(h, 0, agraph).sum(1) of ggpm/encoder.py:31-32,99,135-136 (and, through the
 * transposed CSR, every scatter-add autograd would run for their backward).  accumulate!=0 adds to out.
 * zero_to (all three below): columns [end of the written block, zero_to) of every row are set to 0 as well (the pad
 * columns of the Hp-strided matrices), 0 = leave them alone. */
int ggpm_segment_sum(const float* src, int ld_src, const int32_t* rowptr, const int32_t* col, int rows,
                     int width, float* out, int ld_out, int accumulate, int zero_to, ggpm_stream_t stream);
/* out[r, col_off : col_off+width] = table[idx[r], 0:width]; rows with idx<0 give zeros.
 * = nn.Embedding / index_select of ggpm/encoder.py:98,103,111,114 */
int ggpm_gather_rows(const float* table, int ld_table, const int32_t* idx, int rows, int width,
                     float* out, int ld_out, int col_off, int zero_to, ggpm_stream_t stream);
/* dst[idx[r], 0:width] = src[r, 0:width] (accumulate != 0: +=); rows with idx < 0 are skipped.  The indices must be
 * UNIQUE (plain stores).  The write side of index_select for the compact row sets of the teacher-forced atom-level
 * decode (ggpm/encoder.py:165-179 run on the rows a step touches; ggpm_amd/atom_decode.py). */
int ggpm_scatter_rows(const float* src, int ld_src, const int32_t* idx, int rows, int width, float* dst, int ld_dst,
                      int accumulate, ggpm_stream_t stream);
/* out[r, col_off + idx[r]] = 1, the other `classes` columns of that block 0: the one-hot tables
 * E_a/E_b/E_apos/E_pos of ggpm/encoder.py:74-77,121-125,104-105. */
/* nn.Dropout in training mode (ggpm/encoder.py:15-19, 52-72), in place on x[rows, cols] (leading dimension ld):
 * x[r][c] = keep ? x[r][c] / (1 - p) : 0 with keep = (mix(seed, site, r * cols + c) >> 8) >= p * 2^24, mix = two rounds of
 * the murmur3 32-bit finaliser (see gather.hip; restated in numpy by tests/golden_utils.dropout_keep).  Stateless: the
 * backward applies the same call to the incoming gradient.  Cannot be bit-matched to torch's generator; parity tests
 * inject the same mask into the oracle. */
int ggpm_dropout(float* x, int rows, int cols, int ld, float p, unsigned int seed_lo, unsigned int seed_hi, int site,
                 ggpm_stream_t stream);
/* One Adam step (torch.optim.Adam's arithmetic; ggpm/../vae_train.py:60,83 `optimizer.step()`) over ONE flat fp32 buffer that
 * all parameters are views of; p, g, m, v 16-byte aligned, n elements; step counts from 1.
 *   m += (1-b1)(g' - m); v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  g' = g + wd p */
int ggpm_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, ggpm_stream_t stream);
int ggpm_onehot(const int32_t* idx, int rows, int classes, float* out, int ld_out, int col_off, int zero_to,
                ggpm_stream_t stream);
/* embed_graph (ggpm/encoder.py:119-126) in one launch: hnode[N1, ld_n] = onehot(fnode), hmess[E1, ld_m] =
 * [onehot(fnode[src]) | onehot(bond) | onehot(pos)] from the int64 A0 tensors. */
int ggpm_embed_graph(const int64_t* fnode, int N1, const int64_t* fmess, int E1, int atom_size,
                     int bond_types, int max_pos, float* hnode, int ld_n, float* hmess, int ld_m,
                     ggpm_stream_t stream);

/* One-shot hint for the NEXT dense ggpm_gru_backward / ggpm_lstm_backward of the calling thread: do not accumulate dXz / dXh per depth (their
 * outputs are then undefined).  The per-depth gate gradients are stashed anyway (for the weight-gradient contractions), so
 * a caller that does not need the sums on the critical path -- the atom level has no input gradient: its inputs are one-hot
 * constants, ggpm/encoder.py:119-126 -- forms dXz = sum_t DZP_t, dXh = sum_t DMP_t afterwards with ggpm_sum_slots over
 * the slots ggpm_gru_backward_stashes names, e.g. on a second stream: 13 MB less HBM traffic and four memory operations
 * fewer per element in every depth launch.  ggpm_sum_slots: out[i] = sum_t src[t * slot_floats + i], fixed order. */
void ggpm_backward_skip_x_sums(int yes);
int ggpm_gru_backward_stashes(float* work, int E1, int H, int depth, float** DMP, float** DZP);
/* ... and for ggpm_lstm_backward: dXi, dXo, dXu = sums of the DI / DO / DU stash slots (dXf still accumulates per depth) */
int ggpm_lstm_backward_stashes(float* work, int E1, int H, int depth, float** DI, float** DO, float** DU);
int ggpm_sum_slots(const float* src, int slots, size_t slot_floats, float* out, ggpm_stream_t stream);
/* Gate-product dtype of the level calls (ggpm_gru_/ggpm_lstm_ forward, backward, weight_grads) issued by the CALLING
 * THREAD from now on: 0 = fp32 operands (default, the 1e-4 parity mode: on dense levels large enough for one column group
 * the H x H products run at fp32 accuracy on the bf16 matrix pipe, every operand split exactly into three bf16 values
 * and six of the nine partial products kept; on v_mfma_f32_16x16x4_f32 otherwise),
 * 1 = bf16 operands with fp32 accumulate for the hidden x hidden products of the depth loops and the tall
 * weight-gradient contractions (BASELINE configs[4]; the reference's cells are ggpm/rnn.py:27-36, 88-91), 2 = fp32 on
 * v_mfma_f32_16x16x4_f32 only, 3 = fp32 on split operands wherever their images fit the LDS (0 chooses between 2 and 3
 * per level: split operands for dense levels of one column group).  Returns the previous value; any other argument only
 * queries.  The
 * whole-encoder drivers set it from ggpm_enc_dims.gate_dtype for the duration of their call. */
int ggpm_level_gate_dtype(int dtype);
/* 1 when a DENSE TRAINING level (GRU or LSTM) of E1 message rows (pad row included) and hidden size H keeps the arrays of
 * its depth loop in bf16 under gate dtype 1 (BASELINE configs[4]: "bf16 storage"): the state h and the per-message product
 * q, the stashes S / G / Z / M (LSTM: S / I / O / U), the backward's dS / dG and its DQ / DZP / DMP (DQ / DI / DO / DU)
 * stashes -- rounded (RNE) where they are written, in
 * the first half of the fp32 buffers the caller passes (no size or layout changes at this boundary; the level's result, slot
 * `depth` of Hs, stays fp32 at its usual place).  The rule: E1 >= 6144 and the H x H x E1 contraction qualifies for the
 * bf16 tall kernel, which then reads the stashes as they are.  Kept fp32: gate inputs X and their gradients, R / F, the
 * LSTM cell state c and dFC, every weight and weight gradient.  oracle/ref_encoder.py restates the roundings as gate_dtype "bf16s" (ggpm/rnn.py:25-50). */
int ggpm_level_bf16_storage(int E1, int H);
/* ------------------------------------------------------------------ GRU message function
 * GRU.forward (ggpm/rnn.py:41-50) with GRU.GRU (ggpm/rnn.py:25-39) restated over CSR predecessors with
 * the depth-invariant input halves hoisted:  Xz = x W_z[:, :I]^T + b_z, Xr = x W_r^T, Xh = x W_h[:, :I]^T + b_h
 * are computed once by ggpm_gemm; per depth
 *     s_e = sum_p h_p,  g_e = sum_p sigmoid(Xr_e + q_p) * h_p,   q_p = U_r h_p + b_u,
 *     z = sigmoid(Xz + Wz_h s), m = tanh(Xh + Wh_h g), h' = (1-z) s + z m, row 0 := 0.
 * Two launches per depth over a (row tiles) x (column groups) grid: A = CSR gather -> LDS tiles -> MFMA gate
 * GEMMs -> gate math -> h'; B = MFMA q' GEMM from the h' rows.
 * Stash layout ([t] = depth slot): Hs[depth+1][E1][Hp] (Hs[0]=0, Hs[depth] = result), Qs/Ss/Gs/Zs/Ms/Rs
 * [depth][E1][Hp] (Rs = sum_p h_p r(1-r), which lets the backward form dXr without a gather).  With
 * save_for_backward = 0 only Hs[2][E1][Hp] and Qs[2][E1][Hp] are needed and the result is Hs[depth & 1].
 * wpack: ggpm_gru_pack_floats(H) floats of scratch.
 */
size_t ggpm_gru_pack_floats(int H);
int ggpm_gru_forward(int E1, int H, int depth, const float* Xz, const float* Xr, const float* Xh,
                     const float* Wz_h, int ld_wz, const float* Ur, int ld_ur, const float* bu,
                     const float* Wh_h, int ld_wh, const int32_t* pred_rowptr, const int32_t* pred_col,
                     float* Hs, float* Qs, float* Ss, float* Gs, float* Zs, float* Ms, float* Rs,
                     float* wpack, int save_for_backward, ggpm_stream_t stream);
/* Backward of the above (replaces autograd's replay of ggpm/rnn.py:41-50): given dHD = dL/dh_D it
 * overwrites dXz/dXr/dXh [E1][Hp] and the weight gradients (written as [H,H] blocks with the given
 * leading dimension so they can land inside the full W_z/W_h gradient tensors), dbu[H].
 * succ_* is the CSR transpose of pred_* (ggpm_csr_transpose).  work: ggpm_gru_backward_workspace_bytes(). */
size_t ggpm_gru_backward_workspace_bytes(int E1, int H, int depth);
int ggpm_gru_backward(int E1, int H, int depth, const float* Xr, const float* Wz_h, int ld_wz,
                      const float* Ur, int ld_ur, const float* Wh_h, int ld_wh,
                      const int32_t* pred_rowptr, const int32_t* pred_col, const int32_t* succ_rowptr,
                      const int32_t* succ_col, const float* Hs, const float* Qs, const float* Ss,
                      const float* Gs, const float* Zs, const float* Ms, const float* Rs, const float* dHD, float* dXz,
                      float* dXr, float* dXh, float* dWz_h, int ld_dwz, float* dUr, int ld_dur,
                      float* dbu, float* dWh_h, int ld_dwh, float* work, size_t work_bytes,
                      int weight_grads, ggpm_stream_t stream);
/* The weight-gradient tail of ggpm_gru_backward (called with weight_grads = 0) as its own entry point, so the
 * host may enqueue it on a second stream beside the next level's depth loop. Same `work` buffer. */
int ggpm_gru_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, const float* Gs, float* work,
                          size_t work_bytes, float* dWz_h, int ld_dwz, float* dUr, int ld_dur, float* dbu,
                          float* dWh_h, int ld_dwh, ggpm_stream_t stream);

/* GRU.sparse_forward (ggpm/rnn.py:52-59) -- the incremental form the decoder uses (IncMPNEncoder, ggpm/encoder.py:160-179):
 * rows with frozen[row] != 0 keep their state for the whole loop, the other rows (the `submess` subset) start from 0 and
 * are recomputed `depth` times from the current states of their predecessors.  Same kernels as the dense level:
 * frozen rows have empty predecessor lists, copy their state in the epilogue and pass their gradient through a carry
 * buffer; Hs[0] = masked h_in, Qs[0] = U_r Hs[0] + b_u.  The backward also returns dHin [E1][Hp] (zero on recomputed rows). */
int ggpm_gru_sparse_forward(int E1, int H, int depth, const float* h_in, const unsigned char* frozen, const float* Xz,
                            const float* Xr, const float* Xh, const float* Wz_h, int ld_wz, const float* Ur, int ld_ur,
                            const float* bu, const float* Wh_h, int ld_wh, const int32_t* pred_rowptr,
                            const int32_t* pred_col, float* Hs, float* Qs, float* Ss, float* Gs, float* Zs, float* Ms,
                            float* Rs, float* wpack, int save_for_backward, ggpm_stream_t stream);
int ggpm_gru_sparse_backward(int E1, int H, int depth, const unsigned char* frozen, const float* Xr, const float* Wz_h,
                             int ld_wz, const float* Ur, int ld_ur, const float* Wh_h, int ld_wh,
                             const int32_t* pred_rowptr, const int32_t* pred_col, const int32_t* succ_rowptr,
                             const int32_t* succ_col, const float* Hs, const float* Qs, const float* Ss, const float* Gs,
                             const float* Zs, const float* Ms, const float* Rs, const float* dHD, float* dHin,
                             float* dXz, float* dXr, float* dXh, float* dWz_h, int ld_dwz, float* dUr, int ld_dur,
                             float* dbu, float* dWh_h, int ld_dwh, float* work, size_t work_bytes, ggpm_stream_t stream);

/* ------------------------------------------------------------------ LSTM message function
 * LSTM.forward (ggpm/rnn.py:96-108) with LSTM.LSTM (ggpm/rnn.py:85-94), same restatement:
 *     Xi/Xo/Xu/Xf = x W_*[:, :I]^T + b_*  (hoisted),   qf_p = Wf_h h_p,
 *     s = sum_p h_p,  fc = sum_p sigmoid(Xf_e + qf_p) * c_p,
 *     i = sigmoid(Xi + Wi_h s), o = sigmoid(Xo + Wo_h s), u = tanh(Xu + Wu_h s),
 *     c' = i u + fc,  h' = o tanh(c'),  rows 0 := 0.
 * Stash: Hs, Cs [depth+1][E1][Hp]; Qs (qf), Ss, Is, Os, Us, Fs [depth][E1][Hp] (Fs = sum_p c_p f(1-f)).
 */
size_t ggpm_lstm_pack_floats(int H);
int ggpm_lstm_forward(int E1, int H, int depth, const float* Xi, const float* Xo, const float* Xu,
                      const float* Xf, const float* Wi_h, int ld_wi, const float* Wo_h, int ld_wo,
                      const float* Wu_h, int ld_wu, const float* Wf_h, int ld_wf,
                      const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs, float* Cs,
                      float* Qs, float* Ss, float* Is, float* Os, float* Us, float* Fs, float* wpack,
                      int save_for_backward, ggpm_stream_t stream);
size_t ggpm_lstm_backward_workspace_bytes(int E1, int H, int depth);
int ggpm_lstm_backward(int E1, int H, int depth, const float* Xf, const float* Wi_h, int ld_wi,
                       const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                       int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col,
                       const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs,
                       const float* Cs, const float* Qs, const float* Ss, const float* Is,
                       const float* Os, const float* Us, const float* Fs, const float* dHD, float* dXi,
                       float* dXo, float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                       float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                       size_t work_bytes, int weight_grads, ggpm_stream_t stream);
int ggpm_lstm_weight_grads(int E1, int H, int depth, const float* Hs, const float* Ss, float* work, size_t work_bytes,
                           float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo, float* dWu_h, int ld_dwu,
                           float* dWf_h, int ld_dwf, ggpm_stream_t stream);

/* LSTM.sparse_forward (ggpm/rnn.py:110-121): as ggpm_gru_sparse_forward, with the cell state carried too.  The backward
 * takes dL/dh_D and dL/dc_D (the decoder keeps both) and returns dHin / dCin. */
int ggpm_lstm_sparse_forward(int E1, int H, int depth, const float* h_in, const float* c_in, const unsigned char* frozen,
                             const float* Xi, const float* Xo, const float* Xu, const float* Xf, const float* Wi_h,
                             int ld_wi, const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                             int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col, float* Hs, float* Cs,
                             float* Qs, float* Ss, float* Is, float* Os, float* Us, float* Fs, float* wpack,
                             int save_for_backward, ggpm_stream_t stream);
int ggpm_lstm_sparse_backward(int E1, int H, int depth, const unsigned char* frozen, const float* Xf, const float* Wi_h,
                              int ld_wi, const float* Wo_h, int ld_wo, const float* Wu_h, int ld_wu, const float* Wf_h,
                              int ld_wf, const int32_t* pred_rowptr, const int32_t* pred_col,
                              const int32_t* succ_rowptr, const int32_t* succ_col, const float* Hs, const float* Cs,
                              const float* Qs, const float* Ss, const float* Is, const float* Os, const float* Us,
                              const float* Fs, const float* dHD, const float* dCD, float* dHin, float* dCin, float* dXi,
                              float* dXo, float* dXu, float* dXf, float* dWi_h, int ld_dwi, float* dWo_h, int ld_dwo,
                              float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work, size_t work_bytes,
                              ggpm_stream_t stream);

/* Deferred hidden-half weight gradients for a SEQUENCE of sparse backward calls (the decode steps of one batch,
 * ggpm/decoder.py:201-222 -> ggpm/encoder.py:165-179 once per step; ggpm_amd/atom_decode.py): every step contracts a few
 * hundred rows against the same four H x H matrices, so instead of three or four small contractions per step the
 * gate-gradient stashes of all steps are kept, stacked row-wise, and contracted ONCE.
 *   ggpm_backward_defer_stash(s0, s1, s2, s3): the NEXT ggpm_gru_sparse_backward / ggpm_lstm_sparse_backward call of this
 *   thread writes its stashes to caller memory and leaves dW*_h / dUr / dbu untouched.  GRU: s0 = DMP, s1 = DZP
 *   ([depth][E1][Hp] each, pairing with the forward's Gs / Ss), s2 = DQ ([depth][E1][Hp], slot t pairs with Hs slot t; give
 *   the block depth+1 slots, the last one zero, so that it lines up with Hs), s3 unused.  LSTM: s0..s2 = DI, DO, DU (pair with
 *   Ss), s3 = DQ (pairs with Hs, as above).  One call consumes the setting.
 *   ggpm_*_weight_grads_stacked: rows = total stash rows (sum over the calls of depth*E1), rows_q = total rows of the
 *   DQ / Hs stacks (sum of (depth+1)*E1); every buffer holds the calls' blocks in the same order.  Outputs are
 *   overwritten.  work: ggpm_weight_grads_stacked_workspace_bytes(H, max(rows, rows_q)).
 *   ggpm_weights_packed(1): the NEXT forward / backward / sparse call of this thread skips packing its gate weights: `wpack`
 *   (forward) resp. the head of `work` (backward) still holds the fragments an earlier call with the SAME weights and
 *   the same buffer left there -- the decode steps share one set of weights.  One call consumes the setting. */
void ggpm_weights_packed(int yes);
void ggpm_backward_defer_stash(float* s0, float* s1, float* s2, float* s3);
/* Start state / incoming-state gradient of a sparse call through an index (the decode steps keep every step's rows in
 * stacked blocks; a step's frozen rows are the final states of rows of earlier blocks, ggpm/encoder.py:165-179 reads them
 * out of the level-wide `hmess` instead):
 *   ggpm_forward_gather_state(src_h, src_c, idx): the NEXT ggpm_gru_sparse_forward / ggpm_lstm_sparse_forward of this thread
 *   takes the start state of row r from src_h[idx[r]] (src_c likewise, LSTM; zero where idx[r] < 0) and writes it to slot 0
 *   of Hs (Cs) itself, inside its first launch; `h_in` / `c_in` are not read.
 *   ggpm_backward_scatter_state(dst_h, dst_c, idx): the NEXT ggpm_*_sparse_backward of this thread ADDS the gradient of the
 *   incoming state of row r to dst_h[idx[r]] (dst_c likewise; idx unique, rows with idx[r] < 0 dropped) inside its last
 *   launch; dHin / dCin are not written.  Rows are [Hp] floats.  One call consumes the setting. */
/* ggpm_level_prefer_narrow(1): until switched off again, the dense fp32 level calls of this thread (GRU and LSTM) use two row tiles per
 * workgroup whatever the level's size, i.e. half as many workgroups (the same products in the same order per row; results
 * agree with the default form to rounding, 5e-6 norm-wise after 20 depth steps -- the two instantiations contract the gate
 * expressions differently).  For a
 * level that runs BESIDE a latency-bound chain on another stream -- the encoder next to the decoder's atom level in the
 * full VAE step (ggpm/property_vae.py:47-58 runs them one after the other) -- this leaves the chain's launches free CUs. */
void ggpm_level_prefer_narrow(int yes);
void ggpm_forward_gather_state(const float* src_h, const float* src_c, const int32_t* idx);
void ggpm_backward_scatter_state(float* dst_h, float* dst_c, const int32_t* idx);
size_t ggpm_weight_grads_stacked_workspace_bytes(int H, int rows);
int ggpm_gru_weight_grads_stacked(int rows, int rows_q, int H, const float* DMP, const float* Gs, const float* DZP,
                                  const float* Ss, const float* DQ, const float* Hs, float* dWz_h, int ld_dwz, float* dUr,
                                  int ld_dur, float* dbu, float* dWh_h, int ld_dwh, float* work, size_t work_bytes,
                                  ggpm_stream_t stream);
int ggpm_lstm_weight_grads_stacked(int rows, int rows_q, int H, const float* DI, const float* DO, const float* DU,
                                   const float* Ss, const float* DQ, const float* Hs, float* dWi_h, int ld_dwi,
                                   float* dWo_h, int ld_dwo, float* dWu_h, int ld_dwu, float* dWf_h, int ld_dwf, float* work,
                                   size_t work_bytes, ggpm_stream_t stream);

/* The atom level's decode loop as ONE call per direction (ggpm/decoder.py:201-222 -> ggpm/encoder.py:235-239,165-179 once per
 * step, on the compact row sets of ggpm_amd/atom_decode.py): per step `gather the frozen rows from the stacked state buffer ->
 * ggpm_*_sparse_forward`, backwards `ggpm_*_sparse_backward (stashes deferred, see above) -> scatter-add of dHin to the rows'
 * producers`.  All arrays of the descriptor are HOST arrays of length T (T + 1 for the offsets); the pointers in them are
 * device pointers.  foff: first F id (one per (step, local row)) of a step; roff / qoff: first row of the step's [depth][n]
 * resp. [depth + 1][n] block in the stacked stash / state buffers.  X_all: the steps' gate inputs [G][n][Hp] per step at row
 * G * foff[t]; St_all: five stash kinds, `st_stride` floats apart; DG_all likewise (two kinds GRU, three LSTM).
 * W / ldw: hidden halves, GRU {Wz_h, U_r, Wh_h} (+ bu), LSTM {Wi_h, Wo_h, Wu_h, Wf_h}.  tmp: 2 * max(n) * Hp floats.
 * dW_unused: four H x H (ld H) scratch outputs the deferred backward leaves untouched (GRU: [3] = H floats). */
typedef struct ggpm_decode_steps {
    int T, H, depth, lstm;
    const int32_t* n;
    const int64_t *foff, *roff, *qoff;
    const int32_t* const* srcH;          /* rows of Hs_all / Cs_all a step's frozen rows are read from (-1: zero) */
    const int32_t* const* srcF;          /* the same as F ids (where the backward adds dHin / dCin) */
    const unsigned char* const* frozen;
    const int32_t* const* pred_rowptr;
    const int32_t* const* pred_col;
    const int32_t* const* succ_rowptr;
    const int32_t* const* succ_col;
} ggpm_decode_steps;
int ggpm_decode_steps_forward(const ggpm_decode_steps* steps, const float* const* W, const int* ldw, const float* bu,
                              const float* X_all, float* Hs_all, float* Cs_all, float* Qs_all, float* St_all, size_t st_stride,
                              float* wpack, float* tmp, ggpm_stream_t stream);
int ggpm_decode_steps_backward(const ggpm_decode_steps* steps, const float* const* W, const int* ldw, const float* X_all,
                               const float* Hs_all, const float* Cs_all, const float* Qs_all, const float* St_all,
                               size_t st_stride, float* dF, float* dCF, float* dX_all, float* DG_all, size_t dg_stride,
                               float* DQ_all, float* const* dW_unused, float* work, size_t work_bytes, float* tmp,
                               ggpm_stream_t stream);
/* The same two loops ISSUED by a worker thread of the library: the call returns at once, ggpm_decode_join() waits until
 * the worker has issued everything posted so far (not for the GPU) and returns its first error.  Until then the caller
 * keeps the descriptor, the arrays it points to and every buffer alive, and enqueues what follows the loop on `stream`
 * only after the join.  Lets the ~1.5 ms of host time a loop takes run beside the calling thread's other launches (the
 * encoder's backward, which autograd reaches at the same moment). */
int ggpm_decode_steps_forward_async(const ggpm_decode_steps* steps, const float* const* W, const int* ldw, const float* bu,
                                    const float* X_all, float* Hs_all, float* Cs_all, float* Qs_all, float* St_all,
                                    size_t st_stride, float* wpack, float* tmp, ggpm_stream_t stream);
int ggpm_decode_steps_backward_async(const ggpm_decode_steps* steps, const float* const* W, const int* ldw, const float* X_all,
                                     const float* Hs_all, const float* Cs_all, const float* Qs_all, const float* St_all,
                                     size_t st_stride, float* dF, float* dCF, float* dX_all, float* DG_all, size_t dg_stride,
                                     float* DQ_all, float* const* dW_unused, float* work, size_t work_bytes, float* tmp,
                                     ggpm_stream_t stream);
int ggpm_decode_join(void);

/* ------------------------------------------------------------------ one tree-side level of the teacher-forced decoder
 * IncHierMPNEncoder.embed_sub_tree + IncMPNEncoder.forward (ggpm/encoder.py:208-245, 165-179), which
 * HierMPNDecoder.forward calls once per decode step (ggpm/decoder.py:201-222), for ALL steps at once over the decode-time
 * DAG of the level's messages (ggpm_amd/decoder.py: DecodeSchedule._level_plan), as one call per direction:
 *   finput = E[ids]; hnode = relu([finput | lower] W^T + b); hmess = [hnode[mess_inst] | onehot(mess_pos)];
 *   h = sparse_forward(h0, hmess, all real messages, DAG, `depth` = longest chain); node = relu([hnode | sum_in h] W_o^T + b_o)
 * E1 message rows (row 0 = pad) + n_extra frozen rows that carry `extra` (the motif level's pseudo-messages with the root
 * vectors); n_inst visits.  Index tables int32 on the device: ids[n_inst], mess_inst / mess_pos[E1-1], frozen[E1+n_extra],
 * pred_* / succ_* the DAG's CSR and its transpose over E1+n_extra rows, in_* the incoming-message CSR of every visit
 * (n_inst rows) and inT_* its transpose (E1+n_extra rows), srcT_* the transpose of the message -> visit index (n_inst rows).
 * gate_w / gate_b: GRU {W_z, W_r, W_h} ([H, H+20+H]; W_r [H, H+20], gate_b[1] = null) + Ur, bu; LSTM {W_i, W_o, W, W_f}.
 * Dropout must be inactive (the caller falls back to the per-op entry points otherwise).
 * forward: `saved` = ggpm_tree_level_saved_floats floats; on return `views` names the intermediates inside it (node
 * [n_inst, Hp] and slot `depth` of Hs [E1+n_extra, Hp] are the level's two results).  backward: d_node / d_hid nullable;
 * `grads` names caller-owned outputs -- full-shape gate weight gradients (both halves written), gate bias gradients
 * (null where the gate has none), dUr / dbu (GRU), dpre_w / dpre_o [n_inst, Hp] and d_finput [n_inst, pad16(He)] for the
 * caller's contractions of W, W_o and the embedding table, d_lower (nullable) and dHin [E1+n_extra, Hp] whose rows E1.. are
 * d(extra); work: ggpm_tree_level_work_bytes. */
typedef struct ggpm_tree_level {
    int lstm, H, He, E1, n_extra, depth, n_inst;
    const int32_t *ids, *mess_inst, *mess_pos;
    const unsigned char* frozen;
    const int32_t *pred_rowptr, *pred_col, *succ_rowptr, *succ_col;
    const int32_t *in_rowptr, *in_col, *inT_rowptr, *inT_col;
    const int32_t *srcT_rowptr, *srcT_col;
    const float* emb; int ld_emb;
    const float *W, *b; int ld_w;
    const float *Wo, *bo; int ld_wo;
    const float* gate_w[4]; int ld_gate[4]; const float* gate_b[4];
    const float *Ur, *bu; int ld_ur;
    const float* lower; int ld_lower;
    const float* extra; int ld_extra;
} ggpm_tree_level;
typedef struct ggpm_tree_level_views {
    float *finput, *hnode, *hmess, *X, *hp, *cp, *Hs, *Cs, *Qs, *St, *wpack, *nei, *node;
} ggpm_tree_level_views;
typedef struct ggpm_tree_level_grads {
    float* dgate_w[4]; int ld_dgate[4]; float* dgate_b[4];
    float *dUr, *dbu;
    float *dpre_w, *dpre_o, *d_finput;
    float* d_lower; int ld_dlower, n_pad_dlower;
    float* dHin;
} ggpm_tree_level_grads;
size_t ggpm_tree_level_saved_floats(const ggpm_tree_level* level);
size_t ggpm_tree_level_work_bytes(const ggpm_tree_level* level);
int ggpm_tree_level_forward(const ggpm_tree_level* level, float* saved, size_t saved_floats, ggpm_tree_level_views* views,
                            ggpm_stream_t stream);
/* side_stream (nullable): where the level's PARAMETER gradients (grads->dgate_w / dgate_b / dUr / dbu) are formed, behind an
 * event the main stream records after the depth loop -- the gradients that flow on (d_lower, dHin, dpre_*, d_finput) are
 * then not queued behind ~145 us of contractions.  Same launches, same results.  The caller joins that stream before it
 * reads those four outputs and keeps `work` and the forward's `saved` alive until then. */
int ggpm_tree_level_backward(const ggpm_tree_level* level, const ggpm_tree_level_views* views, const float* d_node,
                             const float* d_hid, const ggpm_tree_level_grads* grads, float* work, size_t work_bytes,
                             ggpm_stream_t stream, ggpm_stream_t side_stream);

/* ------------------------------------------------------------------ decoder score-head losses
 * Softmax cross entropy with reduction = sum and the additive vocabulary mask of ggpm/vocab.py:34-41,56-58 fused in
 * (ggpm/decoder.py:66-69,143-157,268-271): z[m,:] = logits[m,:] + mask[mask_row[m],:] (mask / mask_row both null for
 * no mask), loss[0] = sum_m logsumexp(z[m,:]) - z[m,label[m]]; dlogits (nullable) = softmax(z) - onehot(label);
 * argmax (nullable) = first maximal class per row (get_accuracy, ggpm/nnutils.py:84-87).  work: M floats.
 * ggpm_bce_logits: nn.BCEWithLogitsLoss(size_average=False) of the topology head (decoder.py:66,262-264).
 * ggpm_scale_rows: d[m,:] *= scale[0] (upstream gradient held on the device).  All sums are order-fixed. */
int ggpm_softmax_ce(const float* logits, int ld, int M, int N, const float* mask, int ld_mask, const int32_t* mask_row,
                    const int32_t* label, float* loss, float* dlogits, int ld_d, int32_t* argmax, float* work,
                    ggpm_stream_t stream);
int ggpm_bce_logits(const float* x, const float* y, int M, float* loss, float* dx, float* work, ggpm_stream_t stream);
int ggpm_scale_rows(float* d, int ld, int M, int N, const float* scale, ggpm_stream_t stream);
/* The four accuracies of the teacher-forced decoder pass (ggpm/decoder.py:262-283 with get_accuracy / get_accuracy_bin /
 * get_accuracy_sym, ggpm/nnutils.py:84-97) in one launch: out4 = {motif class, attachment class, topology, attachment}.
 * cls_pred / icls_pred: the arg-max ggpm_softmax_ce returned; topo: the topology scores (element i at topo[i * ld_topo]);
 * assm: the [P x C] attachment scores (row stride ld_assm; P = 0: out4[3] = 1 as the reference's `assm_acc = 1`); labels
 * int64 (labels_int64 = 1) or int32. */
int ggpm_head_accuracies(const int32_t* cls_pred, const void* cls_lab, const int32_t* icls_pred, const void* icls_lab,
                         int n_cls, const float* topo, int ld_topo, const void* topo_lab, int n_topo, const float* assm,
                         int ld_assm, int P, int C, int labels_int64, float* out4, ggpm_stream_t stream);

/* KL head, elementwise part of HierPropertyVAE.rsample (ggpm/property_vae.py:26-33) after the two [B,H]x[H,L] products:
 * lv = -|pv|; kl[0] = -0.5 * sum(1 + lv - mean^2 - exp(lv)) / B; z = mean + exp(lv/2) * eps (eps null: z = mean).
 * mean, pv, eps, z, dz, dmean, dpv: contiguous [B, L]; dz / dkl may be null (no gradient from that output). */
int ggpm_rsample_forward(const float* mean, const float* pv, const float* eps, int B, int L, float* z, float* kl,
                         ggpm_stream_t stream);
int ggpm_rsample_backward(const float* mean, const float* pv, const float* eps, const float* dz, const float* dkl,
                          int B, int L, float* dmean, float* dpv, ggpm_stream_t stream);

/* ------------------------------------------------------------------ whole-encoder drivers
 * HierMPNEncoder.forward (ggpm/encoder.py:140-157, with embed_graph/inter/tree/root :96-138) and its backward as ONE
 * call each: the same kernels the op-by-op host path issues, sequenced from C++ (GRU or LSTM message function).
 * Inputs are the A0 tensors of MolGraph.tensorize() after make_cuda (int64, row-major) plus the molecules' root node
 * ids; outputs are [rows, Hp] with zero pad columns (Hp = ggpm_padded_hidden(H)).
 * params / grads: 35 (GRU) or 38 (LSTM) device pointers, contiguous fp32, in this order (shapes as in the reference
 * state_dict): E_c.0.weight, E_i.0.weight, W_c.0.weight, W_c.0.bias, W_i.0.weight, W_i.0.bias, W_root.0.weight,
 *   W_root.0.bias, then for tree_encoder, inter_encoder, graph_encoder: W_o.0.weight, W_o.0.bias and the cell's
 *   GRU:  rnn.W_z.weight, rnn.W_z.bias, rnn.W_r.weight, rnn.U_r.weight, rnn.U_r.bias, rnn.W_h.weight, rnn.W_h.bias
 *   LSTM: rnn.W_i.0.weight, rnn.W_i.0.bias, rnn.W_o.0.weight, rnn.W_o.0.bias, rnn.W.0.weight, rnn.W.0.bias,
 *         rnn.W_f.0.weight, rnn.W_f.0.bias.
 * saved: ggpm_encoder_saved_bytes() bytes written by the forward and read by the backward; work: backward scratch of
 * ggpm_encoder_work_bytes().  side_stream (may be 0): transposed CSRs and all weight-gradient contractions run there,
 * event-ordered against `stream`; on return from the backward `stream` is ordered behind it.  d_* may be null.
 * phase: 0 = whole backward; 1 = all but the atom level, 2 = atom level + final join (same arguments, same work
 * arena): between the two a data-parallel caller starts all-reducing the gradients of every slot before graph_encoder.
 * Dropout (ggpm/encoder.py:15-19, 52-72: after E_c, E_i, W_c, W_i and every level's W_o): `dropout` > 0 applies the
 * counter-based masks of ggpm_dropout() with `seed_lo/seed_hi` at sites 0 E_i rows, 1 E_c rows, 2 atom-level W_o,
 * 3 W_i, 4 attachment-level W_o, 5 W_c, 6 motif-level W_o; the backward regenerates the same masks from the seed. */
typedef struct ggpm_enc_dims {
    int H, He, depthT, depthG, atom_size, n_motif, n_attach;
    int N1g, E1g, Kg_a, Kg_b;            /* atom graph: nodes+1, messages+1, agraph / bgraph widths */
    int N1t, E1t, Kt_a, Kt_b, Kt_c;      /* motif tree: nodes+1, messages+1, agraph / bgraph / cgraph widths */
    int B;                               /* molecules */
    int rnn_type;                        /* 0 GRU, 1 LSTM */
    int tree_chain;                      /* longest dependency chain among the motif-tree messages of this batch (message
                                            u->v depends on the messages w->u, w != v), or 0 if unknown.  The tree is
                                            acyclic, so after that many steps every message has reached its fixed point:
                                            the two tree-side levels then run tree_chain + 1 of their depthT steps and
                                            replicate the last stash slot -- bit-identical to running all of them
                                            (ggpm/rnn.py:41-50 iterates a fixed depth regardless). */
    float dropout;                       /* drop probability of the training forward (0: none / eval) */
    unsigned int seed_lo, seed_hi;       /* mask stream of this forward/backward pair */
    int gate_dtype;                      /* as ggpm_level_gate_dtype.  0: fp32 gate products (the 1e-4 parity configurations);
                                            2: fp32 on v_mfma_f32_16x16x4_f32 only;
                                            1: bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16) for the H x H gate
                                            products of the depth loops -- BASELINE configs[4].  State, stashes, gate math,
                                            input projections and weight-gradient contractions stay fp32. */
} ggpm_enc_dims;
size_t ggpm_encoder_saved_bytes(const ggpm_enc_dims* dims);
size_t ggpm_encoder_work_bytes(const ggpm_enc_dims* dims);
int ggpm_encoder_forward(const ggpm_enc_dims* dims, float* const* params, const int64_t* tfnode, const int64_t* tfmess,
                         const int64_t* tagraph, const int64_t* tbgraph, const int64_t* tcgraph, const int64_t* gfnode,
                         const int64_t* gfmess, const int64_t* gagraph, const int64_t* gbgraph, const int32_t* roots,
                         void* saved, size_t saved_bytes, float* hroot, float* hnode, float* hinter, float* hatom,
                         ggpm_stream_t stream, ggpm_stream_t side_stream);
int ggpm_encoder_backward(const ggpm_enc_dims* dims, float* const* params, float* const* grads, const int32_t* roots,
                          void* saved, size_t saved_bytes, const float* hroot, const float* hnode, const float* hinter,
                          const float* hatom, const float* d_hroot, const float* d_hnode, const float* d_hinter,
                          const float* d_hatom, void* work, size_t work_bytes, int phase, ggpm_stream_t stream,
                          ggpm_stream_t side_stream);

/* ------------------------------------------------------------------ instrumentation
 * When a timing sink is installed, every depth-step kernel launch is bracketed by HIP events on its own
 * stream; ggpm_timing_collect() synchronises those events and returns launches / total milliseconds /
 * algorithmic flops for kernel class `which` (0 gru_fwd_a, 1 gru_bwd_a, 2 lstm_fwd_a, 3 lstm_bwd_a,
 * 4 gru_fwd_b, 5 gru_bwd_b, 6 lstm_fwd_b, 7 lstm_bwd_b) + 8 * level tag (0 launches issued through the per-op
 * entry points, 1 atom level, 2 attachment level, 3 motif level of ggpm_encoder_forward / _backward), so that the
 * atom-level launch (one workgroup per CU, MFMA bound) is reported apart from the small latency-bound levels.
 * Used by bench.py for roofline.achieved; off by default (no events, no overhead). */
int ggpm_timing_enable(int on);
int ggpm_timing_collect(int which, int* launches, double* total_ms, double* flops);

/* ------------------------------------------------------------------ decode schedule (host only, no GPU work)
 * The integer bookkeeping of the teacher-forced decoder for one tensorized batch, which the reference interleaves
 * with device work on every step of HierMPNDecoder.forward (ggpm/decoder.py:186-259: the per-step subtree / subgraph
 * lists, update_graph_mask :85-100, init_decoder_state :102-122, apply_tree_mask / apply_graph_mask :72-83, the
 * prediction tuples) and IncHierMPNEncoder.get_sub_tensor (ggpm/encoder.py:195-206), derived once per batch:
 * the tables of ggpm_amd.decoder.DecodeSchedule (from_tensors, _level_plan) and ggpm_amd.atom_decode.AtomPlan
 * (compact row sets, compact_tables) -- the numpy forms there are the checker (tests/test_schedule_native.py).
 * Inputs are host arrays, int64, row-major, in the MolGraph.tensorize layout (ggpm/mol_graph.py:199-281):
 *   tree  fnode [Nt1 x 2], fmess [Et1 x 4], agraph [Nt1 x At], bgraph [Et1 x Kt], cgraph [Nt1 x C], scope [B x 2];
 *   graph fmess [Eg1 x 4], agraph [Ng1 x Ag], bgraph [Eg1 x Kg];
 *   orders [n x 3] = (x, y or -1 for None, label) of all molecules back to back, order_off [B + 1];
 *   per tree node v: the attachment ids of its inter_label, icls[icls_off[v] .. icls_off[v+1]) (k_v = their number), and
 *   its assm_cands as cand_off[v+1] - cand_off[v] candidates of k_v atoms each, flat from cands[cand_atom_off[v]].
 * depth / gates: also build the tables that depend on the decoder's diterG and cell (3 GRU / 4 LSTM); 0 = leave out.
 * The int32 pack also carries the index structures the device would otherwise derive per batch from those tables
 * (ggpm_padded_to_csr / ggpm_csr_transpose semantics: in-row order kept, transposed rows ascending): for the two
 * tree-side levels L in {inter, tree} pred_L / succ_L / in_L / inT_L (_rp, _col) and frozen_L (bytes), the message ->
 * visit transpose srcT, the transposes topoT / clsT / assmT of the heads' molecule indices, and iota (the rowptr of any
 * one-entry-per-row CSR); every _col table ends with one pad entry so that none is empty.
 * Returns an opaque handle (NULL: malformed input); tables are read with ggpm_schedule_get by name (pack 0 = host
 * only; 1 / 2 = inside the int64 / int32 device pack; ggpm_schedule_pack returns a pack's base, `offset` is in bytes from it).  The
 * call keeps no global state and may run on any thread. */
typedef struct ggpm_sched_in {
    int B, Nt1, Et1, At, Kt, C;
    int Ng1, Eg1, Ag, Kg;
    int depth, gates;
    const int64_t *tfnode, *tfmess, *tagraph, *tbgraph, *cgraph, *tree_scope;
    const int64_t *gfmess, *gagraph, *gbgraph;
    const int64_t *orders, *order_off;
    const int64_t *icls_off, *icls;
    const int64_t *cand_off, *cand_atom_off, *cands;
} ggpm_sched_in;
void* ggpm_schedule_build(const ggpm_sched_in* in);
int ggpm_schedule_get(void* handle, const char* name, const void** data, int64_t* count, int* elem_bytes, int* pack,
                      int64_t* offset);
int ggpm_schedule_pack(void* handle, int pack, const void** data, int64_t* bytes);
int ggpm_schedule_names(void* handle, char* out, int64_t capacity);      /* newline-separated table names */
/* (pack, byte offset, count, element bytes) per table in the order of ggpm_schedule_names.  Returns the number of
 * tables, or -GGPM_ERR_* on error.  `capacity` in int64 elements (4 per table). */
int ggpm_schedule_directory(void* handle, int64_t* out, int64_t capacity);
void ggpm_schedule_free(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* GGPM_HIP_H */
