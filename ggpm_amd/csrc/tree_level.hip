// One tree-side level of the teacher-forced decoder as ONE C call per direction.
//
// Reference: IncHierMPNEncoder.embed_sub_tree + IncMPNEncoder.forward (ggpm/encoder.py:208-245, 165-179), called once per
// decode step from HierMPNDecoder.forward (ggpm/decoder.py:201-222); de-sequentialised over the decode-time DAG of the
// level's messages (ggpm_amd/decoder.py: DecodeSchedule._level_plan), so that the attachment level and the motif level are
// each
//     finput = E[ids]                               embedding rows of every visit
//     hnode  = relu([finput | lower] W^T + b)       W_i / W_c
//     hmess  = [hnode[visit of the message] | onehot(position)]
//     h      = sparse_forward(h0, hmess, all real messages, DAG, chain)      GRU / LSTM message function
//     node   = relu([hnode | sum of the incoming messages revealed] W_o^T + b_o)
// ggpm_amd/tree_decode.py issued these ~12 (forward) / ~18 (backward) launches from one autograd node through ctypes;
// the two drivers below issue exactly the same launches in the same order from C++ (results bit-identical), carving every
// intermediate out of one caller-provided arena per direction.  The Python node keeps the tape and the deferred
// parameter-gradient queue.
#include "common.h"

namespace {

constexpr int MAX_POS = 20;

struct Dims {
    int H, He, Hp, Hep, I, ldm, E1, Etot, ms, n_inst, depth, G;
    bool lstm;
    size_t slot;       // Etot * Hp
};

inline bool dims_of(const ggpm_tree_level* L, Dims& d) {
    if (!L || L->H <= 0 || L->He <= 0 || L->E1 < 2 || L->n_extra < 0 || L->depth <= 0 || L->n_inst <= 0) return false;
    d.H = L->H; d.He = L->He; d.Hp = ggpm_padded_hidden(L->H); d.Hep = ggpm_padded_hidden(L->He);
    d.I = L->H + MAX_POS; d.ldm = (d.I + 3) / 4 * 4;
    d.E1 = L->E1; d.Etot = L->E1 + L->n_extra; d.ms = L->E1 - 1; d.n_inst = L->n_inst; d.depth = L->depth;
    d.lstm = L->lstm != 0; d.G = d.lstm ? 4 : 3;
    d.slot = (size_t)d.Etot * d.Hp;
    return true;
}

inline size_t r64(size_t floats) { return (floats + 63) & ~(size_t)63; }      // 256-byte granules

struct Carve {
    float* p;
    size_t used, cap;
    float* take(size_t floats) {
        float* q = p ? p + used : nullptr;
        used += r64(floats);
        return q;
    }
};

void views_of(const Dims& d, float* saved, ggpm_tree_level_views& v, size_t& total) {
    Carve c = {saved, 0, 0};
    v.finput = c.take((size_t)d.n_inst * d.Hep);
    v.hnode = c.take((size_t)d.n_inst * d.Hp);
    v.hmess = c.take((size_t)d.ms * d.ldm);
    v.X = c.take((size_t)d.G * d.slot);
    v.hp = c.take(d.slot);
    v.cp = d.lstm ? c.take(d.slot) : nullptr;
    v.Hs = c.take((size_t)(d.depth + 1) * d.slot);
    v.Cs = d.lstm ? c.take((size_t)(d.depth + 1) * d.slot) : nullptr;
    v.Qs = c.take((size_t)d.depth * d.slot);
    v.St = c.take((size_t)5 * d.depth * d.slot);
    v.wpack = c.take(d.lstm ? ggpm_lstm_pack_floats(d.H) : ggpm_gru_pack_floats(d.H));
    v.nei = c.take((size_t)d.n_inst * d.Hp);
    v.node = c.take((size_t)d.n_inst * d.Hp);
    total = c.used;
}

#define CK(x) do { const int rc_ = (x); if (rc_ != GGPM_OK) return rc_; } while (0)

}  // namespace

extern "C" size_t ggpm_tree_level_saved_floats(const ggpm_tree_level* L) {
    Dims d;
    if (!dims_of(L, d)) return 0;
    ggpm_tree_level_views v;
    size_t total = 0;
    views_of(d, nullptr, v, total);
    return total;
}

extern "C" size_t ggpm_tree_level_work_bytes(const ggpm_tree_level* L) {
    Dims d;
    if (!dims_of(L, d)) return 0;
    size_t f = 0;
    f += 3 * r64((size_t)d.n_inst * d.Hp);                 // d_node (when absent), d_hnode, d_nei
    f += r64(d.slot);                                      // dHD
    f += r64((size_t)d.G * d.slot);                        // dX
    f += d.lstm ? 2 * r64(d.slot) : 0;                     // dCD, dCin
    f += r64((size_t)d.ms * d.ldm);                        // dhmess
    f += r64((size_t)256 * d.H);                           // column-sum scratch
    size_t bytes = f * sizeof(float);
    bytes += (d.lstm ? ggpm_lstm_backward_workspace_bytes(d.Etot, d.H, d.depth)
                     : ggpm_gru_backward_workspace_bytes(d.Etot, d.H, d.depth)) + 256;
    bytes += ggpm_gemm_workspace_bytes(d.H, d.I, d.ms) + 256;      // split-K slabs of the input-half weight gradients
    return bytes + 1024;
}

extern "C" int ggpm_tree_level_forward(const ggpm_tree_level* L, float* saved, size_t saved_floats,
                                       ggpm_tree_level_views* out, ggpm_stream_t stream) {
    GGPM_CLEAR_STALE_ERROR();
    Dims d;
    if (!dims_of(L, d) || !saved || !out || !L->ids || !L->mess_inst || !L->mess_pos || !L->frozen || !L->pred_rowptr ||
        !L->pred_col || !L->in_rowptr || !L->in_col || !L->emb || !L->W || !L->b || !L->Wo || !L->bo || !L->lower)
        return GGPM_ERR_ARG;
    for (int k = 0; k < d.G; ++k)
        if (!L->gate_w[k]) return GGPM_ERR_ARG;
    if (!d.lstm && (!L->Ur || !L->bu)) return GGPM_ERR_ARG;
    ggpm_tree_level_views v;
    size_t total = 0;
    views_of(d, saved, v, total);
    if (saved_floats < total) return GGPM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int H = d.H, Hp = d.Hp, I = d.I;
    // 1-2. visit vectors: relu([E[ids] | lower] W^T + b)
    CK(ggpm_gather_rows(L->emb, L->ld_emb, L->ids, d.n_inst, d.He, v.finput, d.Hep, 0, d.Hep, stream));
    {
        const float* A[2] = {v.finput, L->lower};
        const int lda[2] = {d.Hep, L->ld_lower};
        const float* B[2] = {L->W, L->W + d.He};
        const int ldb[2] = {L->ld_w, L->ld_w}, K[2] = {d.He, H};
        CK(ggpm_gemm_ksegments(1, d.n_inst, H, 2, A, lda, B, ldb, K, v.hnode, Hp, Hp, L->b, 0, GGPM_ACT_RELU, 0, stream));
    }
    // 3. message inputs: [hnode[visit of the message] | onehot(position)]
    CK(ggpm_gather_rows(v.hnode, Hp, L->mess_inst, d.ms, H, v.hmess, d.ldm, 0, 0, stream));
    CK(ggpm_onehot(L->mess_pos, d.ms, MAX_POS, v.hmess, d.ldm, H, d.ldm, stream));
    // 4. hoisted gate inputs, straight into the rows 1 .. E1-1 they belong to (row 0 / the extra rows: zero)
    if (hipMemsetAsync(v.X, 0, (size_t)d.G * d.slot * sizeof(float), s) != hipSuccess) return GGPM_ERR_LAUNCH;
    {       // (one grouped launch, as the encoder's levels have it: three launches of 17-18 us -> one of ~12)
        ggpm_gemm_problem gp[4];
        for (int k = 0; k < d.G; ++k)
            gp[k] = {v.hmess, d.ldm, L->gate_w[k], L->ld_gate[k], v.X + (size_t)k * d.slot + Hp, Hp, Hp, L->gate_b[k], 0,
                     GGPM_ACT_NONE, 0};
        CK(ggpm_gemm_grouped(0, 1, d.ms, H, I, d.G, gp, stream));
    }
    // 5. start state: zero, the extra (frozen) rows carry `extra`
    if (hipMemsetAsync(v.hp, 0, d.slot * sizeof(float), s) != hipSuccess) return GGPM_ERR_LAUNCH;
    if (L->extra && L->n_extra > 0 &&
        hipMemcpy2DAsync(v.hp + (size_t)d.E1 * Hp, (size_t)Hp * sizeof(float), L->extra, (size_t)L->ld_extra * sizeof(float),
                         (size_t)H * sizeof(float), (size_t)L->n_extra, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return GGPM_ERR_LAUNCH;
    const size_t ds = (size_t)d.depth * d.slot;
    if (d.lstm) {
        if (hipMemsetAsync(v.cp, 0, d.slot * sizeof(float), s) != hipSuccess) return GGPM_ERR_LAUNCH;
        CK(ggpm_lstm_sparse_forward(d.Etot, H, d.depth, v.hp, v.cp, L->frozen, v.X, v.X + d.slot, v.X + 2 * d.slot,
                                    v.X + 3 * d.slot, L->gate_w[0] + I, L->ld_gate[0], L->gate_w[1] + I, L->ld_gate[1],
                                    L->gate_w[2] + I, L->ld_gate[2], L->gate_w[3] + I, L->ld_gate[3], L->pred_rowptr,
                                    L->pred_col, v.Hs, v.Cs, v.Qs, v.St, v.St + ds, v.St + 2 * ds, v.St + 3 * ds,
                                    v.St + 4 * ds, v.wpack, 1, stream));
    } else {
        CK(ggpm_gru_sparse_forward(d.Etot, H, d.depth, v.hp, L->frozen, v.X, v.X + d.slot, v.X + 2 * d.slot,
                                   L->gate_w[0] + I, L->ld_gate[0], L->Ur, L->ld_ur, L->bu, L->gate_w[2] + I, L->ld_gate[2],
                                   L->pred_rowptr, L->pred_col, v.Hs, v.Qs, v.St, v.St + ds, v.St + 2 * ds, v.St + 3 * ds,
                                   v.St + 4 * ds, v.wpack, 1, stream));
    }
    // 6. read-out of every visit
    const float* hid = v.Hs + (size_t)d.depth * d.slot;
    CK(ggpm_segment_sum(hid, Hp, L->in_rowptr, L->in_col, d.n_inst, H, v.nei, Hp, 0, Hp, stream));
    {
        const float* A[2] = {v.hnode, v.nei};
        const int lda[2] = {Hp, Hp};
        const float* B[2] = {L->Wo, L->Wo + H};
        const int ldb[2] = {L->ld_wo, L->ld_wo}, K[2] = {H, H};
        CK(ggpm_gemm_ksegments(1, d.n_inst, H, 2, A, lda, B, ldb, K, v.node, Hp, Hp, L->bo, 0, GGPM_ACT_RELU, 0, stream));
    }
    *out = v;
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}

// `side_stream` (nullable): the level's PARAMETER gradients -- the hidden-half contractions over the stashes, the input
// halves of the gate weights, the gate biases: ~145 us of a ~470 us level at configs[1] -- are issued there, behind an event
// that the main stream records after the depth loop, so that the gradient the caller waits for (d_lower: what flows on to the
// level below and, on the attachment level, to the atom level's 2.8 ms backward chain) is not queued behind them.  The same
// launches in the same order either way: results do not depend on it.  The caller joins the second stream before it reads
// those gradients and keeps `work` and the forward's `saved` alive until then.
extern "C" int ggpm_tree_level_backward(const ggpm_tree_level* L, const ggpm_tree_level_views* vin, const float* d_node,
                                        const float* d_hid, const ggpm_tree_level_grads* g, float* work, size_t work_bytes,
                                        ggpm_stream_t stream, ggpm_stream_t side_stream) {
    GGPM_CLEAR_STALE_ERROR();
    Dims d;
    if (!dims_of(L, d) || !vin || !g || !work || !L->succ_rowptr || !L->succ_col || !L->inT_rowptr || !L->inT_col ||
        !L->srcT_rowptr || !L->srcT_col || !g->dpre_w || !g->dpre_o || !g->d_finput || !g->dHin)
        return GGPM_ERR_ARG;
    for (int k = 0; k < d.G; ++k)
        if (!g->dgate_w[k]) return GGPM_ERR_ARG;
    if (!d.lstm && (!g->dUr || !g->dbu)) return GGPM_ERR_ARG;
    if (work_bytes < ggpm_tree_level_work_bytes(L)) return GGPM_ERR_WORKSPACE;
    const ggpm_tree_level_views& v = *vin;
    hipStream_t s = (hipStream_t)stream;
    const int H = d.H, Hp = d.Hp, I = d.I, G = d.G;
    Carve c = {work, 0, 0};
    float* zero_node = c.take((size_t)d.n_inst * Hp);
    float* d_hnode = c.take((size_t)d.n_inst * Hp);
    float* d_nei = c.take((size_t)d.n_inst * Hp);
    float* dHD = c.take(d.slot);
    float* dX = c.take((size_t)G * d.slot);
    float* dCD = d.lstm ? c.take(d.slot) : nullptr;
    float* dCin = d.lstm ? c.take(d.slot) : nullptr;
    float* dhmess = c.take((size_t)d.ms * d.ldm);
    float* csws = c.take((size_t)256 * H);
    const size_t lwb = d.lstm ? ggpm_lstm_backward_workspace_bytes(d.Etot, H, d.depth)
                              : ggpm_gru_backward_workspace_bytes(d.Etot, H, d.depth);
    float* lwork = c.take((lwb + 3) / 4);
    const size_t skb = ggpm_gemm_workspace_bytes(H, I, d.ms);
    float* skws = skb ? c.take((skb + 3) / 4) : nullptr;
    // ---- read-out: dpre_o -> d(hnode), d(nei) -> d(final state)
    if (!d_node) {
        if (hipMemsetAsync(zero_node, 0, (size_t)d.n_inst * Hp * sizeof(float), s) != hipSuccess) return GGPM_ERR_LAUNCH;
        d_node = zero_node;
    }
    CK(ggpm_act_backward(d_node, v.node, d.n_inst, H, Hp, GGPM_ACT_RELU, 0, g->dpre_o, stream));
    {
        const ggpm_gemm_problem gp[2] = {{g->dpre_o, Hp, L->Wo, L->ld_wo, d_hnode, Hp, Hp, nullptr, 0, GGPM_ACT_NONE, 0},
                                         {g->dpre_o, Hp, L->Wo + H, L->ld_wo, d_nei, Hp, Hp, nullptr, 0, GGPM_ACT_NONE, 0}};
        CK(ggpm_gemm_grouped(0, 0, d.n_inst, H, H, 2, gp, stream));
    }
    int acc = 0;
    if (d_hid) {
        if (hipMemcpyAsync(dHD, d_hid, d.slot * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return GGPM_ERR_LAUNCH;
        acc = 1;
    }
    CK(ggpm_segment_sum(d_nei, Hp, L->inT_rowptr, L->inT_col, d.Etot, H, dHD, Hp, acc, acc ? 0 : Hp, stream));
    // ---- the level
    const size_t ds = (size_t)d.depth * d.slot;
    hipStream_t ws = side_stream ? (hipStream_t)side_stream : s;       // where the parameter gradients are formed
    if (side_stream) ggpm_sparse_backward_skip_wgrads(1);
    if (d.lstm) {
        if (hipMemsetAsync(dCD, 0, d.slot * sizeof(float), s) != hipSuccess) return GGPM_ERR_LAUNCH;
        CK(ggpm_lstm_sparse_backward(d.Etot, H, d.depth, L->frozen, v.X + 3 * d.slot, L->gate_w[0] + I, L->ld_gate[0],
                                     L->gate_w[1] + I, L->ld_gate[1], L->gate_w[2] + I, L->ld_gate[2], L->gate_w[3] + I,
                                     L->ld_gate[3], L->pred_rowptr, L->pred_col, L->succ_rowptr, L->succ_col, v.Hs, v.Cs, v.Qs,
                                     v.St, v.St + ds, v.St + 2 * ds, v.St + 3 * ds, v.St + 4 * ds, dHD, dCD, g->dHin, dCin, dX,
                                     dX + d.slot, dX + 2 * d.slot, dX + 3 * d.slot, g->dgate_w[0] + I, g->ld_dgate[0],
                                     g->dgate_w[1] + I, g->ld_dgate[1], g->dgate_w[2] + I, g->ld_dgate[2], g->dgate_w[3] + I,
                                     g->ld_dgate[3], lwork, lwb, stream));
    } else {
        CK(ggpm_gru_sparse_backward(d.Etot, H, d.depth, L->frozen, v.X + d.slot, L->gate_w[0] + I, L->ld_gate[0], L->Ur,
                                    L->ld_ur, L->gate_w[2] + I, L->ld_gate[2], L->pred_rowptr, L->pred_col, L->succ_rowptr,
                                    L->succ_col, v.Hs, v.Qs, v.St, v.St + ds, v.St + 2 * ds, v.St + 3 * ds, v.St + 4 * ds, dHD,
                                    g->dHin, dX, dX + d.slot, dX + 2 * d.slot, g->dgate_w[0] + I, g->ld_dgate[0], g->dUr, H,
                                    g->dbu, g->dgate_w[2] + I, g->ld_dgate[2], lwork, lwb, stream));
    }
    if (side_stream) {      // the stashes and dX are complete: hidden halves on the second stream, in the order the level call had them
        hipEvent_t ev = ggpm_wgrad_event(41);
        if (!ev || hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(ws, ev, 0) != hipSuccess) return GGPM_ERR_LAUNCH;
        if (d.lstm)
            CK(ggpm_lstm_sparse_weight_grads(d.Etot, H, d.depth, v.Hs, v.St, lwork, lwb, g->dgate_w[0] + I, g->ld_dgate[0],
                                             g->dgate_w[1] + I, g->ld_dgate[1], g->dgate_w[2] + I, g->ld_dgate[2],
                                             g->dgate_w[3] + I, g->ld_dgate[3], (ggpm_stream_t)ws));
        else
            CK(ggpm_gru_sparse_weight_grads(d.Etot, H, d.depth, v.Hs, v.St, v.St + ds, lwork, lwb, g->dgate_w[0] + I,
                                            g->ld_dgate[0], g->dUr, H, g->dbu, g->dgate_w[2] + I, g->ld_dgate[2],
                                            (ggpm_stream_t)ws));
    }
    // input halves of the gate weights, gate biases (the rows of the real messages, 1 .. E1-1, are contiguous)
    const float* dXs[4];
    for (int k = 0; k < G; ++k) dXs[k] = dX + (size_t)k * d.slot + Hp;
    for (int k = 0; k < G; ++k) {
        CK(ggpm_gemm(1, 0, H, I, d.ms, dXs[k], Hp, v.hmess, d.ldm, g->dgate_w[k], g->ld_dgate[k], I, nullptr, 0, GGPM_ACT_NONE, 0,
                     skws, skb, (ggpm_stream_t)ws));
        if (g->dgate_b[k]) CK(ggpm_colsum(dXs[k], Hp, d.ms, H, g->dgate_b[k], csws, (ggpm_stream_t)ws));
    }
    // ---- message inputs -> visit vectors
    {
        const float* B[4];
        int lda[4], ldb[4], K[4];
        for (int k = 0; k < G; ++k) { B[k] = L->gate_w[k]; lda[k] = Hp; ldb[k] = L->ld_gate[k]; K[k] = H; }
        CK(ggpm_gemm_ksegments(0, d.ms, I, G, dXs, lda, B, ldb, K, dhmess, d.ldm, d.ldm, nullptr, 0, GGPM_ACT_NONE, 0, stream));
    }
    CK(ggpm_segment_sum(dhmess, d.ldm, L->srcT_rowptr, L->srcT_col, d.n_inst, H, d_hnode, Hp, 1, 0, stream));
    CK(ggpm_act_backward(d_hnode, v.hnode, d.n_inst, H, Hp, GGPM_ACT_RELU, 0, g->dpre_w, stream));
    CK(ggpm_gemm(0, 0, d.n_inst, d.He, H, g->dpre_w, Hp, L->W, L->ld_w, g->d_finput, d.Hep, d.Hep, nullptr, 0, GGPM_ACT_NONE, 0,
                 nullptr, 0, stream));
    if (g->d_lower)
        CK(ggpm_gemm(0, 0, d.n_inst, H, H, g->dpre_w, Hp, L->W + d.He, L->ld_w, g->d_lower, g->ld_dlower, g->n_pad_dlower, nullptr,
                     0, GGPM_ACT_NONE, 0, nullptr, 0, stream));
    GGPM_CHECK_LAUNCH();
    return GGPM_OK;
}
