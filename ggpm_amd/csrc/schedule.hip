// Host-side integer bookkeeping of the teacher-forced decoder, built once per batch -- no GPU work in this file.
//
// Reference: HierMPNDecoder.forward interleaves this bookkeeping with device work on every one of its `maxt` steps
// (ggpm/decoder.py:186-259: networkx look-ups, Python lists of prediction tuples, update_graph_mask :85-100,
// init_decoder_state :102-122, apply_tree_mask / apply_graph_mask :72-83, get_sub_tensor ggpm/encoder.py:195-206).
// ggpm_amd/decoder.py (DecodeSchedule.from_tensors, _level_plan) and ggpm_amd/atom_decode.py (AtomPlan, compact_tables)
// separate it out as integer tables; those numpy builders stay as the readable statement and the checker
// (tests/test_schedule_native.py compares every table), this file is the same construction in C++: ~25 ms of numpy per
// batch of 32 become a fraction of a millisecond, which is what lets vae_train.py:78's `model(*batch, beta=beta)` derive
// the schedule inside the step.  The call holds no global state and does not touch the Python runtime, so a loader
// thread can run it beside the training thread.
//
// Tables destined for the device are packed into ONE int64 and ONE int32 buffer (two uploads per batch); the directory
// (name -> pack, offset, count) is read back through ggpm_schedule_get.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../include/ggpm_hip.h"
#ifdef GGPM_DEV_SWITCHES      // (host-only file: no common.h)
static inline const char* ggpm_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* ggpm_dev_env(const char*) { return nullptr; }
#endif

namespace {

typedef std::vector<int64_t> V;

struct Entry { int pack; int64_t off, count; int elem; };      // pack: 0 = host int64, 1 = device int64, 2 = device int32 (+ bytes)

struct Sched {
    std::map<std::string, V> a;             // every table while it is built (int64)
    std::map<std::string, int> kind;        // 0 host, 1 dev64, 2 dev32, 3 dev bytes (tail of the int32 pack)
    std::vector<std::string> order;         // insertion order (pack layout)
    std::vector<int64_t> host, dev64;
    std::vector<int32_t> dev32;
    std::map<std::string, Entry> dir;
    V& put(const std::string& name, int k) {
        if (!a.count(name)) { order.push_back(name); kind[name] = k; }
        return a[name];
    }
    void finalize() {
        size_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        for (const std::string& n : order) {
            const size_t sz = a[n].size();
            const int k = kind[n];
            if (k == 0) n0 += sz; else if (k == 1) n1 += sz; else if (k == 2) n2 += sz; else n3 += (sz + 3) / 4;
        }
        host.resize(n0); dev64.resize(n1); dev32.assign(n2 + n3, 0);
        size_t o0 = 0, o1 = 0, o2 = 0;
        for (const std::string& n : order) {
            const V& v = a[n];
            const int k = kind[n];
            if (k == 0) { dir[n] = {0, (int64_t)o0, (int64_t)v.size(), 8}; if (!v.empty()) std::memcpy(&host[o0], v.data(), v.size() * 8); o0 += v.size(); }
            else if (k == 1) { dir[n] = {1, (int64_t)o1, (int64_t)v.size(), 8}; if (!v.empty()) std::memcpy(&dev64[o1], v.data(), v.size() * 8); o1 += v.size(); }
            else if (k == 2) {
                dir[n] = {2, (int64_t)o2, (int64_t)v.size(), 4};
                int32_t* d = dev32.data() + o2;
                for (size_t i = 0; i < v.size(); ++i) d[i] = (int32_t)v[i];
                o2 += v.size();
            }
        }
        for (const std::string& n : order) {      // byte tables behind all int32 ones (4-byte aligned starts)
            if (kind[n] != 3) continue;
            const V& v = a[n];
            uint8_t* p = reinterpret_cast<uint8_t*>(dev32.data() + o2);
            for (size_t i = 0; i < v.size(); ++i) p[i] = (uint8_t)v[i];
            dir[n] = {2, (int64_t)o2 * 4, (int64_t)v.size(), 1};
            o2 += (v.size() + 3) / 4;
        }
        a.clear();
    }
};

// CSR of the transpose of an entry list (row r_i, column c_i): rows of the result = columns, entries ascending by
// (column, row) -- numpy's lexsort((rows, cols)) of atom_decode._transpose
void transpose(const V& rows, const V& cols, int64_t ncols, V& rp, V& col) {
    rp.assign(ncols + 1, 0);
    for (int64_t c : cols) rp[c + 1]++;
    for (int64_t i = 0; i < ncols; ++i) rp[i + 1] += rp[i];
    // counting sort by column (stable), then each column's few rows ascending
    V cursor(rp.begin(), rp.end() - 1);
    col.resize(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) col[(size_t)cursor[(size_t)cols[i]]++] = rows[i];
    for (int64_t c = 0; c < ncols; ++c)
        if (rp[c + 1] - rp[c] > 1 && !std::is_sorted(col.begin() + rp[c], col.begin() + rp[c + 1]))
            std::sort(col.begin() + rp[c], col.begin() + rp[c + 1]);
}

inline void append(V& dst, const V& src) { dst.insert(dst.end(), src.begin(), src.end()); }

}  // namespace

extern "C" void* ggpm_schedule_build(const ggpm_sched_in* in) {
    if (!in || in->B <= 0 || !in->tfnode || !in->tfmess || !in->tagraph || !in->tbgraph || !in->cgraph || !in->tree_scope ||
        !in->gfmess || !in->gagraph || !in->gbgraph || !in->orders || !in->order_off || !in->icls_off || !in->cand_off)
        return nullptr;
    const int64_t B = in->B, Nt1 = in->Nt1, E1 = in->Et1, At = in->At, Kt = in->Kt, C = in->C;
    const int64_t Ng1 = in->Ng1, Eg1 = in->Eg1, Ag = in->Ag, Kg = in->Kg;
    const int64_t *tfnode = in->tfnode, *tfmess = in->tfmess, *tagraph = in->tagraph, *tbgraph = in->tbgraph;
    const int64_t *cgraph = in->cgraph, *gfmess = in->gfmess, *gagraph = in->gagraph, *gbgraph = in->gbgraph;
    {       // the tables index each other: refuse anything that would read outside them
        auto in_range = [](const int64_t* a, int64_t n, int64_t lo, int64_t hi) {
            for (int64_t i = 0; i < n; ++i) if (a[i] < lo || a[i] >= hi) return false;
            return true;
        };
        if (Nt1 < 1 || E1 < 1 || Ng1 < 1 || Eg1 < 1 || At < 1 || Kt < 1 || C < 1 || Ag < 1 || Kg < 1) return nullptr;
        bool ok = in_range(cgraph, Nt1 * C, 0, Ng1) && in_range(gagraph, Ng1 * Ag, 0, Eg1) && in_range(gbgraph, Eg1 * Kg, 0, Eg1) &&
                  in_range(tagraph, Nt1 * At, 0, E1) && in_range(tbgraph, E1 * Kt, 0, E1);
        for (int64_t e = 1; e < E1 && ok; ++e) ok = tfmess[e * 4] >= 0 && tfmess[e * 4] < Nt1 && tfmess[e * 4 + 1] >= 0 && tfmess[e * 4 + 1] < Nt1;
        for (int64_t e = 1; e < Eg1 && ok; ++e) ok = gfmess[e * 4] >= 0 && gfmess[e * 4] < Ng1 && gfmess[e * 4 + 1] >= 0 && gfmess[e * 4 + 1] < Ng1;
        for (int64_t i = 0; i < B && ok; ++i) ok = in->tree_scope[2 * i] >= 0 && in->tree_scope[2 * i] < Nt1 && in->order_off[i + 1] >= in->order_off[i];
        ok = ok && in->order_off[0] == 0;
        for (int64_t q = 0; ok && q < in->order_off[B]; ++q)
            ok = in->orders[3 * q] >= 0 && in->orders[3 * q] < Nt1 && in->orders[3 * q + 1] >= -1 && in->orders[3 * q + 1] < Nt1;
        for (int64_t v = 0; ok && v < Nt1; ++v)
            ok = in->icls_off[v + 1] >= in->icls_off[v] && in->cand_off[v + 1] >= in->cand_off[v] && in->cand_atom_off &&
                 in->cand_atom_off[v + 1] - in->cand_atom_off[v] == (in->cand_off[v + 1] - in->cand_off[v]) * (in->icls_off[v + 1] - in->icls_off[v]);
        if (ok && in->cand_atom_off[Nt1] > 0) ok = in->cands && in_range(in->cands, in->cand_atom_off[Nt1], 0, Ng1);
        if (ok && in->icls_off[Nt1] > 0) ok = in->icls != nullptr;
        if (!ok) return nullptr;
    }
    Sched* S = new Sched();
    static const bool dbg = ggpm_dev_env("GGPM_SCHED_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_begin = now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        auto t = now();
        fprintf(stderr, "[sched] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_begin).count());
        t_begin = t;
    };

    // ---------------------------------------------------------------- DecodeSchedule.from_tensors
    std::unordered_map<int64_t, int64_t> tmess;          // (u, v) -> tree message id
    tmess.reserve((size_t)E1 * 2);
    for (int64_t e = 1; e < E1; ++e) tmess[tfmess[e * 4] * Nt1 + tfmess[e * 4 + 1]] = e;
    auto mess = [&](int64_t u, int64_t v) -> int64_t {
        auto it = tmess.find(u * Nt1 + v);
        return it == tmess.end() ? -1 : it->second;
    };
    // src atom -> (dst atom, bond) lists, bonds ascending, as one CSR
    V gadj_rp((size_t)Ng1 + 1, 0), gadj_dst((size_t)Eg1), gadj_e((size_t)Eg1);
    for (int64_t e = 1; e < Eg1; ++e) { const int64_t u = gfmess[e * 4]; if (u >= 0 && u < Ng1) gadj_rp[(size_t)u + 1]++; }
    for (int64_t i = 0; i < Ng1; ++i) gadj_rp[(size_t)i + 1] += gadj_rp[(size_t)i];
    {
        V cur(gadj_rp.begin(), gadj_rp.end() - 1);
        for (int64_t e = 1; e < Eg1; ++e) {
            const int64_t u = gfmess[e * 4];
            if (u >= 0 && u < Ng1) { const int64_t q = cur[(size_t)u]++; gadj_dst[(size_t)q] = gfmess[e * 4 + 1]; gadj_e[(size_t)q] = e; }
        }
    }
    auto cluster_size = [&](int64_t v) { int64_t n = 0; for (int64_t j = 0; j < C; ++j) n += cgraph[v * C + j] > 0; return n; };
    auto add_cluster = [&](int64_t v, V& out) { for (int64_t j = 0; j < C; ++j) if (cgraph[v * C + j] > 0) out.push_back(cgraph[v * C + j]); };
    std::vector<char> in_new((size_t)Ng1, 0), revealed((size_t)Ng1, 0), bond_live((size_t)Eg1, 0);
    auto reveal = [&](const V& atoms, V& bonds) {        // update_graph_mask: the bonds induced by the new atoms
        bonds.clear();
        for (int64_t z : atoms) in_new[(size_t)z] = 1;
        for (int64_t z : atoms)
            for (int64_t q = gadj_rp[(size_t)z]; q < gadj_rp[(size_t)z + 1]; ++q)
                if (in_new[(size_t)gadj_dst[(size_t)q]]) bonds.push_back(gadj_e[(size_t)q]);
        for (int64_t z : atoms) in_new[(size_t)z] = 0;
    };

    V &root_clab = S->put("root_clab", 0), &root_ilab = S->put("root_ilab", 0);
    V cur_atoms, cur_bonds, new_atoms;
    for (int64_t i = 0; i < B; ++i) {
        const int64_t root = in->tree_scope[2 * i];
        root_clab.push_back(tfnode[root * 2]);
        root_ilab.push_back(tfnode[root * 2 + 1]);
        add_cluster(root, cur_atoms);
    }
    reveal(cur_atoms, cur_bonds);
    const int64_t max_cls = 2 * C;
    int64_t maxt = 0;
    for (int64_t i = 0; i < B; ++i) maxt = std::max(maxt, in->order_off[i + 1] - in->order_off[i]);
    const int64_t T = maxt;

    V mess_time((size_t)E1, -1), mess_inst((size_t)E1, 0);
    V &inst_node = S->put("inst_node", 0), inst_step;
    V &pool = S->put("pool", 0);
    V &atoms_all = S->put("atoms_all", 1), &bonds_all = S->put("bonds_all", 0);
    V &g_agraph = S->put("g_agraph", 0), &g_bgraph = S->put("g_bgraph", 0);
    V &atom_off = S->put("atom_off", 0), &bond_off = S->put("bond_off", 0), &inst_off = S->put("inst_off", 0);
    V &submess_all = S->put("submess_all", 0), &submess_off = S->put("submess_off", 0);
    V &topo_batch = S->put("topo_batch", 0), &topo_label = S->put("topo_label", 1);
    V &cls_mess = S->put("cls_mess", 1), &cls_off = S->put("cls_off", 0);
    V cls_batch, cls_clab, cls_ilab;          // (the B roots come first in the packed forms)
    V &assm_step = S->put("assm_step", 0), &assm_yid = S->put("assm_yid", 0), &assm_nth = S->put("assm_nth", 0);
    V &assm_bidx = S->put("assm_bidx", 0);
    atom_off.push_back(0); bond_off.push_back(0); inst_off.push_back(0); submess_off.push_back(0); cls_off.push_back(0);
    bool ok_atoms = true, bad = false;
    for (int64_t t = 0; t < T; ++t) {
        append(atoms_all, cur_atoms);
        append(bonds_all, cur_bonds);
        atom_off.push_back((int64_t)atoms_all.size());
        bond_off.push_back((int64_t)bonds_all.size());
        if (cur_atoms.empty()) ok_atoms = false;
        for (int64_t z : cur_atoms) revealed[(size_t)z] = 1;
        // apply_graph_mask + get_sub_tensor of this step: the rows of the step's atoms / bonds with the entries revealed by now
        for (int64_t e : cur_bonds) bond_live[(size_t)e] = 1;
        for (int64_t z : cur_atoms)
            for (int64_t j = 0; j < Ag; ++j) { const int64_t x = gagraph[z * Ag + j]; g_agraph.push_back(bond_live[(size_t)x] ? x : 0); }
        for (int64_t e : cur_bonds)
            for (int64_t j = 0; j < Kg; ++j) { const int64_t x = gbgraph[e * Kg + j]; g_bgraph.push_back(bond_live[(size_t)x] ? x : 0); }
        for (int64_t i = 0; i < B; ++i) {
            if (t >= in->order_off[i + 1] - in->order_off[i]) continue;
            const int64_t* o = in->orders + 3 * (in->order_off[i] + t);
            const int64_t xid = o[0], yid = o[1];
            for (int64_t j = 0; j < C; ++j) { const int64_t av = cgraph[xid * C + j]; pool.push_back(av > 0 && revealed[(size_t)av] ? av : 0); }
            inst_node.push_back(xid);
            inst_step.push_back(t);
            if (yid >= 0) {
                const int64_t m = mess(xid, yid);
                if (m < 0) { bad = true; continue; }
                submess_all.push_back(m);
                mess_time[(size_t)m] = t;
                mess_inst[(size_t)m] = (int64_t)inst_node.size() - 1;
            }
        }
        new_atoms.clear();
        for (int64_t i = 0; i < B; ++i) {
            if (t >= in->order_off[i + 1] - in->order_off[i]) continue;
            const int64_t* o = in->orders + 3 * (in->order_off[i] + t);
            const int64_t xid = o[0], yid = o[1], tlab = o[2];
            topo_batch.push_back(i);
            topo_label.push_back(tlab);
            if (yid >= 0) add_cluster(yid, new_atoms);           // "regardless of tlab", ggpm/decoder.py:230
            if (tlab == 0) continue;
            const int64_t m = yid >= 0 ? mess(xid, yid) : -1;
            if (m < 0) { bad = true; continue; }
            cls_mess.push_back(m);
            cls_batch.push_back(i);
            cls_clab.push_back(tfnode[yid * 2]);
            cls_ilab.push_back(tfnode[yid * 2 + 1]);
            if (cluster_size(xid) > 2) {                          // attachment is ambiguous only inside a ring
                const int64_t back = mess(yid, xid);
                if (back < 0) { bad = true; continue; }
                const int64_t k = in->icls_off[yid + 1] - in->icls_off[yid], n = in->cand_off[yid + 1] - in->cand_off[yid];
                if (k <= 0 || n <= 0 || n > max_cls) { bad = true; continue; }
                assm_step.push_back(t); assm_yid.push_back(yid); assm_nth.push_back(tfmess[back * 4 + 2]); assm_bidx.push_back(i);
            }
        }
        inst_off.push_back((int64_t)inst_node.size());
        submess_off.push_back((int64_t)submess_all.size());
        cls_off.push_back((int64_t)cls_mess.size());
        cur_atoms = new_atoms;
        reveal(cur_atoms, cur_bonds);
    }
    lap("steps");
    if (bad) { delete S; return nullptr; }
    const int64_t n_inst = (int64_t)inst_node.size();
    {       // packed prediction lists (DecodeSchedule.cls / assm_batch)
        V &cb = S->put("cls_batch", 0), &cc = S->put("cls_clab", 1), &ci = S->put("cls_ilab", 1);
        for (int64_t i = 0; i < B; ++i) cb.push_back(i);
        append(cb, cls_batch);
        append(cc, root_clab); append(cc, cls_clab);
        append(ci, root_ilab); append(ci, cls_ilab);
        V& ab = S->put("assm_batch", 0);
        for (int64_t b : assm_bidx) for (int64_t j = 0; j < max_cls; ++j) ab.push_back(b);
        // the molecule index of every prediction as int32 (row ids of the context gather, ggpm_gather_rows)
        S->put("topo_batch32", 2) = topo_batch;
        S->put("cls_batch32", 2) = cb;
        S->put("assm_batch32", 2) = ab;
    }

    // ---------------------------------------------------------------- DecodeSchedule._level_plan
    const int64_t BIG = INT64_MAX;
    V mt((size_t)E1);
    bool all_live = true;
    for (int64_t e = 0; e < E1; ++e) {
        mt[(size_t)e] = mess_time[(size_t)e] >= 0 ? mess_time[(size_t)e] : BIG;
        if (e > 0 && mess_time[(size_t)e] < 0) all_live = false;
    }
    V dec_ag(tagraph, tagraph + Nt1 * At), dec_bg(tbgraph, tbgraph + E1 * Kt);
    for (int64_t i = 0; i < B; ++i) {                    // init_decoder_state, ggpm/decoder.py:108-115
        const int64_t root = in->tree_scope[2 * i];
        dec_ag[(size_t)(root * At + At - 1)] = E1 + i;
        for (int64_t e = 1; e < E1; ++e)
            if (tfmess[e * 4] == root) dec_bg[(size_t)(e * Kt + Kt - 1)] = E1 + i;
    }
    auto dag = [&](const int64_t* table, bool pseudo, V& out) {
        out.assign((size_t)((E1 - 1) * Kt), 0);
        for (int64_t e = 1; e < E1; ++e)
            for (int64_t j = 0; j < Kt; ++j) {
                const int64_t p = table[e * Kt + j];
                bool keep = p > 0 && p < E1 && mt[(size_t)p] < mt[(size_t)e];
                if (pseudo && p >= E1) keep = true;
                if (keep) out[(size_t)((e - 1) * Kt + j)] = p;
            }
    };
    auto incoming = [&](const int64_t* table, bool pseudo, V& out) {
        out.assign((size_t)(n_inst * At), 0);
        for (int64_t r = 0; r < n_inst; ++r)
            for (int64_t j = 0; j < At; ++j) {
                const int64_t p = table[inst_node[(size_t)r] * At + j];
                bool keep = p > 0 && p < E1 && mt[(size_t)p] <= inst_step[(size_t)r];
                if (pseudo && p >= E1) keep = true;
                if (keep) out[(size_t)(r * At + j)] = p;
            }
    };
    V &dag_inter = S->put("dag_inter", 1), &dag_tree = S->put("dag_tree", 1);
    V &in_inter = S->put("in_inter", 1), &in_tree = S->put("in_tree", 1);
    dag(dec_bg.data(), true, dag_tree);
    dag(tbgraph, false, dag_inter);
    incoming(dec_ag.data(), true, in_tree);
    incoming(tagraph, false, in_inter);
    int64_t chain_max = 0;
    {
        std::vector<int64_t> by_time;
        for (int64_t e = 1; e < E1; ++e) if (mess_time[(size_t)e] >= 0) by_time.push_back(e);
        std::stable_sort(by_time.begin(), by_time.end(), [&](int64_t x, int64_t y) { return mess_time[(size_t)x] < mess_time[(size_t)y]; });
        V chain((size_t)E1, 0);
        for (int64_t m : by_time) {
            int64_t best = 0;
            for (int64_t j = 0; j < Kt; ++j) {
                const int64_t p = dag_inter[(size_t)((m - 1) * Kt + j)];
                if (p > 0) best = std::max(best, chain[(size_t)p]);
            }
            chain[(size_t)m] = 1 + best;
            chain_max = std::max(chain_max, chain[(size_t)m]);
        }
        if (E1 <= 1) chain_max = 0;
    }
    {
        V &im = S->put("inst_motif", 2), &ia = S->put("inst_attach", 2), &mi = S->put("mess_inst", 2), &mp = S->put("mess_pos", 2);
        for (int64_t r = 0; r < n_inst; ++r) { im.push_back(tfnode[inst_node[(size_t)r] * 2]); ia.push_back(tfnode[inst_node[(size_t)r] * 2 + 1]); }
        for (int64_t e = 1; e < E1; ++e) { mi.push_back(mess_inst[(size_t)e]); mp.push_back(tfmess[e * 4 + 2]); }
    }

    // ---------------------------------------------------------------- index structures of the batched tree-side levels and of
    // the heads, exactly as the device kernels derive them from the tables above (ggpm_padded_to_csr keeps the in-row order of
    // the non-zero entries, ggpm_csr_transpose lists the rows of every column ascending): a batch that arrives with its schedule
    // brings them in the same upload instead of ~30 single-workgroup launches in front of the levels and the backward
    {
        auto csr_pair = [&](const V& table, int64_t nrows_tab, int64_t width, int64_t first_row, int64_t total_rows, int64_t ncols,
                            const std::string& name, const std::string& nameT) {
            V &rp = S->put(name + "_rp", 2), &col = S->put(name + "_col", 2);
            V er, ec;
            rp.assign((size_t)total_rows + 1, 0);
            for (int64_t r = 0; r < nrows_tab; ++r) {
                int64_t cnt = 0;
                for (int64_t j = 0; j < width; ++j) {
                    const int64_t p = table[(size_t)(r * width + j)];
                    if (p != 0) { col.push_back(p); er.push_back(first_row + r); ec.push_back(p); ++cnt; }
                }
                rp[(size_t)(first_row + r) + 1] = cnt;
            }
            for (int64_t i = 0; i < total_rows; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
            col.push_back(0);                              // (never an empty table: its address is handed to kernels)
            V &rpT = S->put(nameT + "_rp", 2), &colT = S->put(nameT + "_col", 2);
            transpose(er, ec, ncols, rpT, colT);
            colT.push_back(0);
        };
        auto index_T = [&](const V& idx, int64_t ncols, const std::string& nameT) {
            V rows(idx.size());
            for (size_t i = 0; i < rows.size(); ++i) rows[i] = (int64_t)i;
            V &rpT = S->put(nameT + "_rp", 2), &colT = S->put(nameT + "_col", 2);
            transpose(rows, idx, ncols, rpT, colT);
            colT.push_back(0);
        };
        const int64_t n_extra[2] = {0, B};
        const V* dags[2] = {&dag_inter, &dag_tree};
        const V* ins[2] = {&in_inter, &in_tree};
        const char* tag[2] = {"inter", "tree"};
        for (int l = 0; l < 2; ++l) {
            const int64_t Etot = E1 + n_extra[l];
            const std::string t = tag[l];
            csr_pair(*dags[l], E1 - 1, Kt, 1, Etot, Etot, "pred_" + t, "succ_" + t);
            csr_pair(*ins[l], n_inst, At, 0, n_inst, Etot, "in_" + t, "inT_" + t);
            V& fz = S->put("frozen_" + t, 3);              // rows the level never recomputes: the pad row and the pseudo rows
            fz.assign((size_t)Etot, 0);
            fz[0] = 1;
            for (int64_t e = E1; e < Etot; ++e) fz[(size_t)e] = 1;
        }
        V mi(mess_inst.begin() + 1, mess_inst.end());
        index_T(mi, std::max<int64_t>(n_inst, 1), "srcT");
        index_T(topo_batch, B, "topoT");
        index_T(S->a["cls_batch"], B, "clsT");
        index_T(S->a["assm_batch"], B, "assmT");
        V& iota = S->put("iota", 2);                        // rowptr of every one-entry-per-row CSR: a prefix of 0, 1, 2, ...
        const size_t longest = std::max({mi.size(), topo_batch.size(), S->a["cls_batch"].size(), S->a["assm_batch"].size()});
        iota.resize(longest + 1);
        for (size_t i = 0; i < iota.size(); ++i) iota[i] = (int64_t)i;
    }
    lap("level plan");
    // ---------------------------------------------------------------- AtomPlan (compact row sets)
    V &nloc = S->put("nloc", 0), &floc_off = S->put("floc_off", 0), &frozen_loc = S->put("frozen_loc", 3);
    V &lpred_rp = S->put("lpred_rp", 2), &lpred_col = S->put("lpred_col", 2), &lsucc_rp = S->put("lsucc_rp", 2);
    V &lsucc_col = S->put("lsucc_col", 2), &rows_all = S->put("rows", 2), &loc_all = S->put("loc", 0);
    V &lp_rp_off = S->put("lpred_rp_off", 0), &lp_col_off = S->put("lpred_col_off", 0), &rows_off = S->put("rows_off", 0);
    lp_rp_off.push_back(0); lp_col_off.push_back(0); rows_off.push_back(0); floc_off.push_back(0);
    std::map<int64_t, V> cand_pos, meta_icls, meta_nth, meta_dest;     // by atoms-per-candidate k
    struct Here { int64_t k, start, n; };
    std::vector<std::vector<Here>> per_step_cands((size_t)T);
    V lpos((size_t)Eg1, -1), pos((size_t)Ng1, -1);
    size_t assm_i = 0;
    int64_t pred_i = 0;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t b0 = bond_off[(size_t)t], b1 = bond_off[(size_t)t + 1], a0 = atom_off[(size_t)t], a1 = atom_off[(size_t)t + 1];
        const int64_t nb = b1 - b0;
        std::vector<int64_t> ord((size_t)nb);
        for (int64_t i = 0; i < nb; ++i) ord[(size_t)i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return bonds_all[(size_t)(b0 + x)] < bonds_all[(size_t)(b0 + y)]; });
        // compact row set: the step's bonds + the frozen rows they read + the null row, ascending
        V rows;
        rows.push_back(0);
        for (int64_t i = 0; i < nb; ++i) {
            rows.push_back(bonds_all[(size_t)(b0 + i)]);
            for (int64_t j = 0; j < Kg; ++j) { const int64_t p = g_bgraph[(size_t)((b0 + i) * Kg + j)]; if (p > 0) rows.push_back(p); }
        }
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        const int64_t n = (int64_t)rows.size();
        for (int64_t i = 0; i < n; ++i) lpos[(size_t)rows[(size_t)i]] = i;
        V lcounts((size_t)n, 0), ecol, erow;
        for (int64_t q = 0; q < nb; ++q) {
            const int64_t i = ord[(size_t)q], lr = lpos[(size_t)bonds_all[(size_t)(b0 + i)]];
            int64_t cnt = 0;
            for (int64_t j = 0; j < Kg; ++j) {
                const int64_t p = g_bgraph[(size_t)((b0 + i) * Kg + j)];
                if (p > 0) { ecol.push_back(lpos[(size_t)p]); erow.push_back(lr); ++cnt; }
            }
            lcounts[(size_t)lr] = cnt;          // (ascending global id = ascending local id; a repeated bond: the last wins)
        }
        int64_t run = 0;
        lpred_rp.push_back(0);
        for (int64_t i = 0; i < n; ++i) { run += lcounts[(size_t)i]; lpred_rp.push_back(run); }
        append(lpred_col, ecol);
        V rpT, colT;
        transpose(erow, ecol, n, rpT, colT);
        append(lsucc_rp, rpT);
        append(lsucc_col, colT);
        V fl((size_t)n, 1);
        for (int64_t i = 0; i < nb; ++i) fl[(size_t)lpos[(size_t)bonds_all[(size_t)(b0 + i)]]] = 0;
        nloc.push_back(n);
        const int64_t f0 = floc_off.back();
        floc_off.push_back(f0 + (n + 15) / 16 * 16);
        frozen_loc.resize((size_t)floc_off.back(), 1);
        for (int64_t i = 0; i < n; ++i) frozen_loc[(size_t)(f0 + i)] = fl[(size_t)i];
        append(rows_all, rows);
        lp_rp_off.push_back((int64_t)lpred_rp.size()); lp_col_off.push_back((int64_t)lpred_col.size());
        rows_off.push_back((int64_t)rows_all.size());
        for (int64_t r : rows) lpos[(size_t)r] = -1;
        // pooled cluster vectors / attachment candidates read the step's atom vectors (position inside the step)
        for (int64_t i = a0; i < a1; ++i) pos[(size_t)atoms_all[(size_t)i]] = i - a0;
        for (int64_t r = inst_off[(size_t)t]; r < inst_off[(size_t)t + 1]; ++r)
            for (int64_t j = 0; j < C; ++j) { const int64_t av = pool[(size_t)(r * C + j)]; loc_all.push_back(av > 0 ? pos[(size_t)av] : -1); }
        while (assm_i < assm_step.size() && assm_step[assm_i] == t) {
            const int64_t yid = assm_yid[assm_i];
            const int64_t k = in->icls_off[yid + 1] - in->icls_off[yid], n_c = in->cand_off[yid + 1] - in->cand_off[yid];
            // the node's candidates: n_c tuples of k atoms, flat; their offset = k * (candidates before this node) is
            // carried by the caller as cand_atom_off
            const int64_t* ca = in->cands + in->cand_atom_off[yid];
            V &cp = cand_pos[k], &mi = meta_icls[k], &mn = meta_nth[k], &md = meta_dest[k];
            const int64_t start = (int64_t)cp.size();
            for (int64_t q = 0; q < n_c * k; ++q) cp.push_back(pos[(size_t)ca[q]]);
            for (int64_t q = 0; q < n_c; ++q) for (int64_t j = 0; j < k; ++j) mi.push_back(in->icls[in->icls_off[yid] + j]);
            for (int64_t q = 0; q < n_c * k; ++q) mn.push_back(assm_nth[assm_i]);
            for (int64_t q = 0; q < n_c; ++q) md.push_back(pred_i * max_cls + q);
            per_step_cands[(size_t)t].push_back({k, start, n_c * k});
            ++pred_i;
            ++assm_i;
        }
        for (int64_t i = a0; i < a1; ++i) pos[(size_t)atoms_all[(size_t)i]] = -1;
    }
    V &cand_blocks = S->put("cand_blocks", 0);           // (k, base, count) per group, k ascending
    std::map<int64_t, int64_t> kbase;
    int64_t n_cand = 0;
    for (auto& kv : cand_pos) {
        cand_blocks.push_back(kv.first); cand_blocks.push_back(n_cand); cand_blocks.push_back((int64_t)kv.second.size());
        kbase[kv.first] = n_cand;
        n_cand += (int64_t)kv.second.size();
    }
    for (auto& kv : cand_pos) {
        const std::string sfx = "/" + std::to_string(kv.first);
        S->put("meta_icls" + sfx, 2) = meta_icls[kv.first];
        S->put("meta_nth" + sfx, 1) = meta_nth[kv.first];
        S->put("meta_dest" + sfx, 1) = meta_dest[kv.first];
        S->put("cand_pos" + sfx, 0) = kv.second;
    }
    {
        V& psc = S->put("step_cands", 0);            // (step, k, start, count) per prediction
        for (int64_t t = 0; t < T; ++t) for (const Here& h : per_step_cands[(size_t)t]) { psc.push_back(t); psc.push_back(h.k); psc.push_back(h.start); psc.push_back(h.n); }
    }

    lap("atom plan");
    // ---------------------------------------------------------------- AtomPlan.compact_tables(depth, gates)
    int64_t Ftot = 0;
    V& foff = S->put("foff", 0);
    foff.push_back(0);
    for (int64_t t = 0; t < T; ++t) foff.push_back(foff.back() + nloc[(size_t)t]);
    Ftot = foff.back();
    if (in->depth > 0 && in->gates > 0) {
        const int64_t depth = in->depth, gates = in->gates;
        V Hid((size_t)Ftot);
        for (int64_t t = 0; t < T; ++t)
            for (int64_t i = 0; i < nloc[(size_t)t]; ++i)
                Hid[(size_t)(foff[(size_t)t] + i)] = (depth + 1) * foff[(size_t)t] + depth * nloc[(size_t)t] + i;
        V last((size_t)Eg1, -1);
        V &srcF = S->put("srcF", 2), &srcH = S->put("srcH", 2);
        V agr_r, agr_c, pool_r, pool_c;
        for (int64_t t = 0; t < T; ++t) {
            const int64_t r0 = rows_off[(size_t)t], n = nloc[(size_t)t], f0 = floc_off[(size_t)t];
            for (int64_t i = 0; i < n; ++i) {
                const int64_t sf = frozen_loc[(size_t)(f0 + i)] == 1 ? last[(size_t)rows_all[(size_t)(r0 + i)]] : -1;
                srcF.push_back(sf);
                srcH.push_back(sf >= 0 ? Hid[(size_t)sf] : -1);
            }
            for (int64_t i = 0; i < n; ++i)
                if (frozen_loc[(size_t)(f0 + i)] == 0) last[(size_t)rows_all[(size_t)(r0 + i)]] = foff[(size_t)t] + i;
            const int64_t a0 = atom_off[(size_t)t], a1 = atom_off[(size_t)t + 1];
            for (int64_t r = 0; r < a1 - a0; ++r)
                for (int64_t j = 0; j < Ag; ++j) {
                    const int64_t e = g_agraph[(size_t)((a0 + r) * Ag + j)];
                    if (e > 0 && last[(size_t)e] >= 0) { agr_r.push_back(a0 + r); agr_c.push_back(last[(size_t)e]); }
                }
            for (int64_t r = inst_off[(size_t)t]; r < inst_off[(size_t)t + 1]; ++r)
                for (int64_t j = 0; j < C; ++j) {
                    const int64_t l = loc_all[(size_t)(r * C + j)];
                    if (l >= 0) { pool_r.push_back(r); pool_c.push_back(a0 + l); }
                }
        }
        const int64_t ns_tot = atom_off.back();
        auto rowptr = [&](const V& r, int64_t nrows, V& out) {
            out.assign((size_t)nrows + 1, 0);
            for (int64_t x : r) out[(size_t)x + 1]++;
            for (int64_t i = 0; i < nrows; ++i) out[(size_t)i + 1] += out[(size_t)i];
        };
        rowptr(agr_r, ns_tot, S->put("agr_rp", 2));
        { V& c = S->put("agr_col", 2); for (int64_t f : agr_c) c.push_back(Hid[(size_t)f]); }
        transpose(agr_r, agr_c, Ftot, S->put("agrT_rp", 2), S->put("agrT_col", 2));
        rowptr(pool_r, n_inst, S->put("pool_rp", 2));
        S->put("pool_col", 2) = pool_c;
        transpose(pool_r, pool_c, ns_tot, S->put("poolT_rp", 2), S->put("poolT_col", 2));
        V cidx((size_t)std::max<int64_t>(n_cand, 1), -1);
        for (int64_t t = 0; t < T; ++t)
            for (const Here& h : per_step_cands[(size_t)t])
                for (int64_t q = 0; q < h.n; ++q) {
                    const int64_t p = cand_pos[h.k][(size_t)(h.start + q)];
                    cidx[(size_t)(kbase[h.k] + h.start + q)] = p >= 0 ? atom_off[(size_t)t] + p : -1;
                }
        V okr, okc;
        for (size_t i = 0; i < cidx.size(); ++i) if (cidx[i] >= 0) { okr.push_back((int64_t)i); okc.push_back(cidx[i]); }
        S->put("cand_idx", 2) = cidx;
        transpose(okr, okc, ns_tot, S->put("candT_rp", 2), S->put("candT_col", 2));
        V& xrows = S->put("xrows", 2);
        for (int64_t t = 0; t < T; ++t)
            for (int64_t k = 0; k < gates; ++k)
                for (int64_t i = rows_off[(size_t)t]; i < rows_off[(size_t)t + 1]; ++i) xrows.push_back(k * Eg1 + rows_all[(size_t)i]);
        V iota(xrows.size());
        for (size_t i = 0; i < iota.size(); ++i) iota[i] = (int64_t)i;
        transpose(iota, xrows, gates * Eg1, S->put("xT_rp", 2), S->put("xT_col", 2));
    }

    lap("compact tables");
    V& sc = S->put("scalars", 0);
    sc = {T, n_inst, E1, chain_max, all_live ? 1 : 0, max_cls, (int64_t)assm_step.size(), n_cand, Ftot, ok_atoms ? 1 : 0, B,
          (int64_t)in->depth, (int64_t)in->gates, Ng1, Eg1};
    S->finalize();
    lap("finalize");
    return S;
}

extern "C" int ggpm_schedule_get(void* h, const char* name, const void** data, int64_t* count, int* elem_bytes, int* pack,
                                 int64_t* offset) {
    if (!h || !name) return GGPM_ERR_ARG;
    Sched* S = static_cast<Sched*>(h);
    auto it = S->dir.find(name);
    if (it == S->dir.end()) return GGPM_ERR_ARG;
    const Entry& e = it->second;
    const uint8_t* base = e.pack == 0 ? reinterpret_cast<const uint8_t*>(S->host.data())
                          : e.pack == 1 ? reinterpret_cast<const uint8_t*>(S->dev64.data())
                                        : reinterpret_cast<const uint8_t*>(S->dev32.data());
    const int64_t byte_off = e.elem == 1 ? e.off : e.off * e.elem;
    if (data) *data = base + byte_off;
    if (count) *count = e.count;
    if (elem_bytes) *elem_bytes = e.elem;
    if (pack) *pack = e.pack;
    if (offset) *offset = byte_off;
    return GGPM_OK;
}

extern "C" int ggpm_schedule_pack(void* h, int pack, const void** data, int64_t* bytes) {
    if (!h || pack < 0 || pack > 2 || !data || !bytes) return GGPM_ERR_ARG;
    Sched* S = static_cast<Sched*>(h);
    if (pack == 0) { *data = S->host.data(); *bytes = (int64_t)S->host.size() * 8; }
    else if (pack == 1) { *data = S->dev64.data(); *bytes = (int64_t)S->dev64.size() * 8; }
    else { *data = S->dev32.data(); *bytes = (int64_t)S->dev32.size() * 4; }
    return GGPM_OK;
}

// (pack, byte offset, count, element bytes) of every table, in the order of ggpm_schedule_names
extern "C" int ggpm_schedule_directory(void* h, int64_t* out, int64_t capacity) {
    if (!h || !out) return -GGPM_ERR_ARG;
    Sched* S = static_cast<Sched*>(h);
    if ((int64_t)S->order.size() * 4 > capacity) return -GGPM_ERR_WORKSPACE;
    int64_t* o = out;
    for (const std::string& n : S->order) {
        const Entry& e = S->dir[n];
        *o++ = e.pack; *o++ = e.elem == 1 ? e.off : e.off * e.elem; *o++ = e.count; *o++ = e.elem;
    }
    return (int)S->order.size();
}

extern "C" int ggpm_schedule_names(void* h, char* out, int64_t cap) {
    if (!h || !out || cap <= 0) return GGPM_ERR_ARG;
    Sched* S = static_cast<Sched*>(h);
    std::string all;
    for (const std::string& n : S->order) { all += n; all += '\n'; }
    if ((int64_t)all.size() + 1 > cap) return GGPM_ERR_WORKSPACE;
    std::memcpy(out, all.c_str(), all.size() + 1);
    return GGPM_OK;
}

extern "C" void ggpm_schedule_free(void* h) { delete static_cast<Sched*>(h); }
