// graph.cpp — the traversal-facing slice of graph.rs plus the operators that sit on it:
// CondTraverseOp::expand_batch / expand_row, ExpandIntoOp, algo.BFS.
#include <algorithm>

#include "host.hpp"

namespace falkor {

// ---- Graph ------------------------------------------------------------------------------------------
Graph::Graph(Context& ctx, u64 node_cap, u64 label_cap)
    : ctx_(&ctx), n_(node_cap), adj_(ctx, node_cap, node_cap), labels_(ctx, node_cap, label_cap) {}

LabelId Graph::add_label(const std::string& name) {
    auto it = label_ids_.find(name);
    if (it != label_ids_.end()) return it->second;
    LabelId id = label_ids_.size();
    if (id >= labels_.ncols()) labels_.resize(labels_.nrows(), labels_.ncols() * 2);
    label_ids_[name] = id;
    return id;
}

u64 Graph::add_type(const std::string& name) {
    auto it = type_ids_.find(name);
    if (it != type_ids_.end()) return it->second;
    u64 id = tensors_.size();
    tensors_.emplace_back(*ctx_, n_, n_);
    type_ids_[name] = id;
    return id;
}

std::optional<LabelId> Graph::label_id(const std::string& name) const {
    auto it = label_ids_.find(name);
    if (it == label_ids_.end()) return std::nullopt;
    return it->second;
}
std::optional<u64> Graph::type_id(const std::string& name) const {
    auto it = type_ids_.find(name);
    if (it == type_ids_.end()) return std::nullopt;
    return it->second;
}

void Graph::create_edge(u64 type, u64 src, u64 dst, u64 edge_id) {
    tensors_.at(type).set_all_from_slices({src}, {dst}, {edge_id});
    adj_.set_all({{src, dst}}, false);   // two edges may share one pair: NEW = false
}

void Graph::delete_edge(u64 type, u64 src, u64 dst, u64 edge_id) {
    auto emptied = tensors_.at(type).remove_all({{edge_id, src, dst}});
    if (emptied.empty()) return;
    for (auto& t : tensors_)
        if (t.eff_get(src, dst)) return;   // another type still connects the pair
    adj_.remove(src, dst);
}

void Graph::create_edges(u64 type, const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                         const std::vector<u64>& ids) {
    tensors_.at(type).set_all_from_slices(srcs, dsts, ids);
    std::vector<std::pair<u64, u64>> pairs(srcs.size());
    for (size_t k = 0; k < srcs.size(); ++k) pairs[k] = {srcs[k], dsts[k]};
    adj_.set_all(pairs, false);
}

void Graph::new_version() {
    adj_ = adj_.dup();
    labels_ = labels_.dup();
    for (auto& t : tensors_) t = t.dup();
}

void Graph::fold_oversized_deltas() {
    adj_.fold_oversized();
    labels_.fold_oversized();
    for (auto& t : tensors_) t.fold_oversized();
}

bool Graph::node_has_label_id(u64 node, LabelId l) const { return labels_.get(node, l).has_value(); }

std::optional<std::vector<LabelId>> Graph::resolve_label_ids(const std::vector<std::string>& labels) const {
    std::vector<LabelId> out;
    for (auto& l : labels) {
        auto id = label_id(l);
        if (!id) return std::nullopt;
        out.push_back(*id);
    }
    return out;
}

std::vector<u64> Graph::label_bitmap(const std::vector<LabelId>& ids) const {
    const u64 words = (n_ + 63) / 64;
    std::vector<u64> bits(words, ~0ull);
    if (n_ % 64) bits[words - 1] = (1ull << (n_ % 64)) - 1;
    if (ids.empty()) return bits;
    // the label matrix is node x label: its transpose has one row per label = the node set of that label
    VersionedMatrix lt = labels_.transpose();
    for (LabelId l : ids) {
        std::vector<u64> has(words, 0);
        for (auto& e : lt.iter(l, l)) has[e.col >> 6] |= 1ull << (e.col & 63);
        for (u64 w = 0; w < words; ++w) bits[w] &= has[w];
    }
    return bits;
}

Matrix Graph::build_relationship_matrix_unrestricted(const std::vector<u64>& type_ids) const {
    // graph.rs:2520-2549: the first type's extract, then for every further type
    //   m<!dm_t> U= pattern(m_t) ; m U= pattern(dp_t)
    if (type_ids.empty()) return adj_.extract();
    Matrix m = tensors_.at(type_ids[0]).extract();
    for (size_t k = 1; k < type_ids.size(); ++k) {
        const Tensor& t = tensors_.at(type_ids[k]);
        t.wait_fwd();
        m.set_pattern(&t.fwd_dm(), t.fwd_m(), Descriptor::C);
        m.set_pattern(nullptr, t.fwd_dp(), Descriptor::None);
    }
    return m;
}

Matrix Graph::build_adjacency_matrix(const std::vector<std::string>& types) const {
    if (types.empty()) return adj_.extract();                       // graph.rs:3874-3876
    if (types.size() == 1) {
        auto id = type_id(types[0]);
        return id ? tensors_[*id].extract() : Matrix(*ctx_, Type::Bool, n_, n_);
    }
    Matrix result(*ctx_, Type::Bool, n_, n_);
    for (auto& t : types)
        if (auto id = type_id(t)) {
            Matrix e = tensors_[*id].extract();
            result.element_wise_add(nullptr, nullptr, &e, Descriptor::None);
        }
    return result;
}

Matrix Graph::build_symmetric_adjacency_matrix(const std::vector<std::string>& types) const {
    // graph.rs:3898-3907: result = A (+) A'  (the undirected view WCC / CDLP / MSF run on)
    Matrix a = build_adjacency_matrix(types);
    Matrix at = a.transpose();
    Matrix result(*ctx_, Type::Bool, n_, n_);
    result.element_wise_add(nullptr, &a, &at, Descriptor::None);
    return result;
}

std::vector<u64> Graph::get_src_dest_relationships(u64 src, u64 dst, const std::vector<u64>& type_ids) const {
    std::vector<u64> out;
    for (u64 t : type_ids) {
        auto ids = tensors_.at(t).get(src, dst);
        out.insert(out.end(), ids.begin(), ids.end());
    }
    return out;
}

// ---- CondTraverse -------------------------------------------------------------------------------------
bool CondTraverseOp::batched_eligible() const {
    // cond_traverse.rs:308-316
    return !emit_relationship && !bidirectional && !has_sibling_edges && (hops.size() > 1 || !has_inline_attrs);
}

namespace {
struct HopLayers {
    // keeps temporaries (materialized unions) alive for the duration of the call
    std::vector<Matrix> owned;
    std::vector<const fgpu_mat*> m, dp, dm;
};

// Type names of one hop -> tensor ids, the way the reference resolves them: a single unknown type is no_match
// (cond_traverse.rs:481-490), an alternation drops its unknown names (`filter_map` in
// build_relationship_matrix_unrestricted, graph.rs:2524-2527, and in edge_type_indices, cond_traverse.rs:418-426)
// and is no_match only when none is known.  Returns false for no_match.
bool resolve_hop_types(const Graph& g, const std::vector<std::string>& types, std::vector<u64>& ids) {
    ids.clear();
    for (auto& t : types)
        if (auto id = g.type_id(t)) ids.push_back(*id);
    return types.empty() || !ids.empty();
}

// Matrix choice per hop (cond_traverse.rs:478-505): no type -> adjacency; one type -> that tensor's forward
// layers; an alternation -> materialized union of its known types with clean deltas (also when only one of them
// is known: the reference branches on the number of NAMES).  Returns false for no_match.
bool hop_layers(const Graph& g, const std::vector<Hop>& hops, HopLayers& hl, std::vector<std::vector<u64>>& type_ids,
                bool transposed = false) {
    hl.owned.reserve(3 * hops.size());
    for (auto& h : hops) {
        std::vector<u64> ids;
        if (!resolve_hop_types(g, h.types, ids)) return false;
        type_ids.push_back(ids);
        if (transposed) {
            // the structures build_transposed_iter walks (cond_traverse.rs:221-235), as device layers: the tensor's own `mt`
            // (kept by every insert / delete, Tensor::matrix_t), the transposes of the adjacency layers (cached on their
            // snapshots: paid once per matrix version), the transpose of a materialized alternation
            auto keep = [&](Matrix m) { hl.owned.push_back(std::move(m)); return hl.owned.back().snapshot(); };
            if (h.types.size() == 1) {
                const VersionedMatrix& mt = g.relationship_tensors()[ids[0]].matrix_t();
                mt.wait();
                hl.m.push_back(mt.m().snapshot());
                hl.dp.push_back(mt.dp().nvals() ? mt.dp().snapshot() : nullptr);
                hl.dm.push_back(mt.dm().nvals() ? mt.dm().snapshot() : nullptr);
            } else if (h.types.empty()) {
                const VersionedMatrix& a = g.adjacency_matrix();
                a.wait();
                hl.m.push_back(keep(a.m().transpose()));
                hl.dp.push_back(a.dp().nvals() ? keep(a.dp().transpose()) : nullptr);
                hl.dm.push_back(a.dm().nvals() ? keep(a.dm().transpose()) : nullptr);
            } else {
                hl.m.push_back(keep(g.build_relationship_matrix_unrestricted(ids).transpose()));
                hl.dp.push_back(nullptr);
                hl.dm.push_back(nullptr);
            }
            continue;
        }
        if (h.types.empty()) {
            const VersionedMatrix& a = g.adjacency_matrix();
            a.wait();
            hl.m.push_back(a.m().snapshot());
            hl.dp.push_back(a.dp().nvals() ? a.dp().snapshot() : nullptr);
            hl.dm.push_back(a.dm().nvals() ? a.dm().snapshot() : nullptr);
        } else if (h.types.size() == 1) {
            const Tensor& t = g.relationship_tensors()[ids[0]];
            t.wait_fwd();
            hl.m.push_back(t.fwd_m().snapshot());
            hl.dp.push_back(t.fwd_dp().nvals() ? t.fwd_dp().snapshot() : nullptr);
            hl.dm.push_back(t.fwd_dm().nvals() ? t.fwd_dm().snapshot() : nullptr);
        } else {
            hl.owned.push_back(g.build_relationship_matrix_unrestricted(ids));
            hl.m.push_back(hl.owned.back().snapshot());
            hl.dp.push_back(nullptr);
            hl.dm.push_back(nullptr);
        }
    }
    return true;
}
}  // namespace

bool CondTraverseOp::expand_batch(const Graph& g, const std::vector<Value>& src, const std::vector<Value>* to_bound,
                                  ExpandedRows& rows, std::vector<u64>& null_rows, u64* flops) const {
    rows.clear();
    null_rows.clear();
    if (flops) *flops = 0;
    const u64 k = src.size();
    auto no_match = [&]() {
        if (optional)
            for (u64 i = 0; i < k; ++i) null_rows.push_back(i);
        return true;
    };
    // FH_TIMING=1: phase times of every call on stderr (tools/bench_paths.py host)
    static const bool timing = getenv("FH_TIMING") != nullptr;
    auto tp = timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();   // no clock reads on the hot path
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[expand_batch] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    };
    HopLayers hl;
    std::vector<std::vector<u64>> type_ids;
    // A `transposed` operator has the matrix DESTINATION bound (select_scan_node swaps from / to): the reference declines
    // the batch there (:556-568) and walks row d of the transposed structure per row (:221-235, 840-870).  The device holds
    // those structures as matrices, so the same batch machinery runs over them — F[i, bound_i] = 1, one product with the
    // transposed layers, `src_labels` on the bound node, the hop's labels on the node it reaches, the representative edge
    // looked up as (reached, bound) — and gives what the per-row path gives row by row.  One hop only (a transposed operator
    // is never fused, fuse_anonymous_traverse.rs:118-122).
    if (transposed && hops.size() != 1) return false;
    if (!hop_layers(g, hops, hl, type_ids, transposed)) return no_match();          // unknown type (:485-490)
    lap("hop_layers");
    auto last_dst = g.resolve_label_ids(hops.back().dst_labels);
    auto src_lids = g.resolve_label_ids(src_labels);
    if (!last_dst || !src_lids) return no_match();                      // unknown label

    // F[i, src_i] = 1 for bound Node sources passing ALL source labels (:556-601)
    std::vector<u64> src_ids(k, ~0ull);
    std::vector<u64> cand_rows, cand_nodes;
    for (u64 i = 0; i < k; ++i) {
        if (src[i].kind != Value::Node) {
            if (optional) continue;
            return false;                                               // per-row fallback (:566-575)
        }
        cand_rows.push_back(i);
        cand_nodes.push_back(src[i].id);
    }
    std::vector<uint8_t> ok(cand_rows.size(), 1);
    if (!src_lids->empty() && !cand_rows.empty()) {
        // node_has_label_id for every (row, label) in one batch of probes per layer of the label matrix
        const VersionedMatrix& lab = g.node_labels_matrix();
        lab.wait();
        for (LabelId l : *src_lids) {
            std::vector<u64> cols(cand_nodes.size(), l);
            std::vector<uint8_t> in_m, in_dm, in_dp;
            lab.m().probe(cand_nodes, cols, in_m, nullptr);
            lab.dm().probe(cand_nodes, cols, in_dm, nullptr);
            lab.dp().probe(cand_nodes, cols, in_dp, nullptr);
            for (size_t c = 0; c < cand_nodes.size(); ++c)
                if (!((in_m[c] && !in_dm[c]) || (!in_m[c] && in_dp[c]))) ok[c] = 0;
        }
    }
    bool any = false;
    for (size_t c = 0; c < cand_rows.size(); ++c)
        if (ok[c]) { src_ids[cand_rows[c]] = cand_nodes[c]; any = true; }
    if (!any) return no_match();

    std::vector<u64> bitmap;
    if (!last_dst->empty()) bitmap = g.label_bitmap(*last_dst);         // dst label filter of the LAST hop (:647-651)
    lap("source labels, bitmap");

    // The chain runs once on the device and BOTH result columns are built there (fgpu_expand_pairs): the source-row index of
    // every pair expanded from the row pointers, a row with a pre-bound `to` cut down to that one destination (:657-661) —
    // they arrive by DMA in pinned blocks the operator hands on as they are.  (Round 4 streamed the destinations and built the
    // row column on the host: 10.5 ms against 4.1 ms for the bare device call on a 16.6 M-pair batch.)
    fgpu_ctx* ctx = g.ctx().raw();
    u64 fl = 0, np = 0;
    std::vector<u64> pinned_to;
    if (to_bound) {
        bool any_pinned = false;
        pinned_to.assign(k, ~0ull);
        for (u64 i = 0; i < k; ++i)
            if ((*to_bound)[i].kind == Value::Node) { pinned_to[i] = (*to_bound)[i].id; any_pinned = true; }
        if (!any_pinned) pinned_to.clear();
    }
    const bool want_edge = bind_relationship && hops.size() == 1;
    bool all_pinned = !pinned_to.empty();
    for (u64 i = 0; i < k && all_pinned; ++i)
        if (src_ids[i] != ~0ull && pinned_to[i] == ~0ull) all_pinned = false;
    if (all_pinned) {
        // every row of the batch has its destination bound (the multi-hop ExpandInto shape, test_expand_into.py:63-95): the
        // last hop is one probe per row of the chain's state (fgpu_expand_probe) — nothing is expanded, emitted or copied
        std::vector<uint8_t> present(k, 0);
        check(fgpu_expand_probe(ctx, src_ids.data(), pinned_to.data(), k, hl.m.data(), hl.dp.data(), hl.dm.data(), (int)hops.size(),
                                bitmap.empty() ? nullptr : bitmap.data(), present.data(), &fl),
              "CondTraverse::expand_batch (pinned)");
        for (u64 i = 0; i < k; ++i)
            if (present[i] && src_ids[i] != ~0ull) { rows.active_row.push_back(i); rows.dest.push_back(pinned_to[i]); }
        if (flops) *flops = fl;
        lap("fgpu_expand_probe");
    } else {
    void* prow = nullptr;
    uint32_t* pdest = nullptr;
    const int row_bits = k > 65536 ? 32 : 16;                // (more than 64 child batches coalesced into one call)
    check(fgpu_expand_pairs32(ctx, src_ids.data(), k, hl.m.data(), hl.dp.data(), hl.dm.data(), (int)hops.size(),
                              bitmap.empty() ? nullptr : bitmap.data(), pinned_to.empty() ? nullptr : pinned_to.data(), row_bits, &prow,
                              &pdest, &np, &fl),
          "CondTraverse::expand_batch");
    rows.pin_ctx = ctx;
    rows.row_pin = row_bits == 16 ? (const uint16_t*)prow : nullptr;
    rows.row_pin32 = row_bits == 32 ? (const uint32_t*)prow : nullptr;
    rows.dest_pin = pdest;
    rows.n_pin = np;
    if (flops) *flops = fl;
    lap("fgpu_expand_pairs");
    }
    std::vector<uint8_t> matched(k, 0);
    if (want_edge) {
        // representative edge: first id found scanning the types in order (:663-695), batched per type
        std::vector<u64> tids = type_ids[0];
        if (tids.empty())
            for (u64 t = 0; t < g.relationship_tensors().size(); ++t) tids.push_back(t);
        rows.materialize();                                          // (this path drops and annotates pairs in place)
        const size_t n = rows.dest.size();
        std::vector<u64> es(n);
        for (size_t r = 0; r < n; ++r) es[r] = src_ids[rows.active_row[r]];
        std::vector<std::optional<u64>> rep(n);
        for (u64 t : tids) {
            std::vector<std::vector<u64>> ids;
            if (transposed) g.relationship_tensors()[t].get_batch(rows.dest, es, ids);   // the stored pair is (reached, bound)
            else g.relationship_tensors()[t].get_batch(es, rows.dest, ids);
            for (size_t r = 0; r < n; ++r)
                if (!rep[r] && !ids[r].empty()) rep[r] = ids[r][0];
        }
        size_t o = 0;
        rows.edge.resize(n);
        for (size_t r = 0; r < n; ++r) {
            if (!rep[r]) continue;                                                   // no edge: the pair is dropped
            rows.active_row[o] = rows.active_row[r];
            rows.dest[o] = rows.dest[r];
            rows.edge[o] = *rep[r];
            ++o;
        }
        rows.active_row.resize(o);
        rows.dest.resize(o);
        rows.edge.resize(o);
    }
    if (optional) {
        for (size_t r = 0; r < rows.size(); ++r) matched[rows.row_at(r)] = 1;
        for (u64 i = 0; i < k; ++i)
            if (!matched[i]) null_rows.push_back(i);
    }
    return true;
}

void ExpandedRows::release_pinned() {
    if (pin_ctx) {
        if (row_pin) fgpu_free(pin_ctx, (void*)row_pin);
        if (row_pin32) fgpu_free(pin_ctx, (void*)row_pin32);
        if (dest_pin) fgpu_free(pin_ctx, (void*)dest_pin);
    }
    pin_ctx = nullptr; row_pin = nullptr; row_pin32 = nullptr; dest_pin = nullptr; n_pin = 0;
}

void ExpandedRows::materialize() {
    if (!pinned()) return;
    active_row.resize(n_pin);
    dest.resize(n_pin);
    for (size_t i = 0; i < n_pin; ++i) { active_row[i] = row_pin ? (u64)row_pin[i] : (u64)row_pin32[i]; dest[i] = dest_pin[i]; }
    release_pinned();
}

void CondTraverseOp::expand_row(const Graph& g, std::optional<u64> from_id, std::optional<u64> to_id, bool transposed,
                                const std::vector<u64>& used_edges, std::vector<std::array<u64, 3>>& out,
                                BidirDedup* dedup, std::optional<u64> dedup_src) const {
    const Hop& h = hops.at(0);
    // build_state (cond_traverse.rs:362-440): labels of the matrix source / destination for the forward pass and,
    // for a bidirectional pattern, for the reverse pass; any unknown label or an unresolvable type is no_match
    std::vector<u64> tids;
    if (!resolve_hop_types(g, h.types, tids)) return;                    // build_unrestricted_iter -> None
    auto from_l = g.resolve_label_ids(src_labels);
    auto to_l = g.resolve_label_ids(h.dst_labels);
    if (!from_l || !to_l) return;
    const std::vector<LabelId>& fwd_src_l = transposed ? *to_l : *from_l;
    const std::vector<LabelId>& fwd_dst_l = transposed ? *from_l : *to_l;
    const std::vector<LabelId>& rev_src_l = transposed ? *from_l : *to_l;       // :376-389
    const std::vector<LabelId>& rev_dst_l = transposed ? *to_l : *from_l;
    std::vector<u64> scan = tids;                                        // edge_type_indices (:418-426)
    if (h.types.empty())
        for (u64 t = 0; t < g.relationship_tensors().size(); ++t) scan.push_back(t);

    // rows [lo, hi] of the unrestricted pair matrix in ascending (row, col) order (build_unrestricted_iter :196-209)
    auto fwd_rows = [&](u64 lo, u64 hi) {
        if (h.types.empty()) return g.adjacency_matrix().iter(lo, hi);
        if (h.types.size() == 1) return g.relationship_tensors()[tids[0]].structural_iter(lo, hi);
        return g.build_relationship_matrix_unrestricted(tids).iter(lo, hi);
    };
    // row d of the TRANSPOSED pair matrix: (dest, src) ascending (build_transposed_iter :221-235)
    auto bwd_row = [&](u64 d) {
        if (h.types.size() == 1) return g.relationship_tensors()[tids[0]].matrix_t().iter(d, d);
        Matrix a = h.types.empty() ? g.adjacency_matrix().extract() : g.build_relationship_matrix_unrestricted(tids);
        return a.transpose().iter(d, d);
    };
    // (src, dst) matrix coordinates of one pass (:852-874 / :897-921)
    auto pairs_of = [&](std::optional<u64> msrc, std::optional<u64> mdst, bool drop_loops) {
        std::vector<std::pair<u64, u64>> pairs;
        if (!msrc && mdst) {
            for (auto& e : bwd_row(*mdst)) pairs.push_back({e.col, e.row});
        } else {
            for (auto& e : fwd_rows(msrc ? *msrc : 0, msrc ? *msrc : ~0ull))
                if (!mdst || *mdst == e.col) pairs.push_back({e.row, e.col});
        }
        if (drop_loops)
            pairs.erase(std::remove_if(pairs.begin(), pairs.end(), [](auto& p) { return p.first == p.second; }), pairs.end());
        return pairs;
    };
    // process_pairs (:978-1117), without the attribute filters (the attribute store is out of scope)
    auto process = [&](const std::vector<std::pair<u64, u64>>& pairs, bool is_reverse, const std::vector<LabelId>& sl,
                       const std::vector<LabelId>& dl) {
        for (auto& pr : pairs) {
            const u64 s = pr.first, d = pr.second;
            bool okl = true;
            for (LabelId l : sl) okl = okl && g.node_has_label_id(s, l);
            for (LabelId l : dl) okl = okl && g.node_has_label_id(d, l);
            if (!okl) continue;
            const u64 from_node = is_reverse ? d : s, to_node = is_reverse ? s : d;
            if (from_id && *from_id != from_node) continue;
            if (to_id && *to_id != to_node) continue;
            const bool first_only = !emit_relationship;                  // one representative edge per pair (:1063-1081)
            bool done = false;
            for (u64 t : scan) {
                for (u64 id : g.relationship_tensors()[t].get(s, d)) {
                    if (std::find(used_edges.begin(), used_edges.end(), id) != used_edges.end()) continue;
                    out.push_back({from_node, to_node, id});
                    if (first_only) { done = true; break; }
                }
                if (done) break;
            }
        }
    };
    const size_t start = out.size();
    const std::optional<u64> fwd_src = transposed ? to_id : from_id, fwd_dst = transposed ? from_id : to_id;
    process(pairs_of(fwd_src, fwd_dst, false), transposed, fwd_src_l, fwd_dst_l);
    if (bidirectional) {                                                 // the reverse relationships (:894-945)
        const std::optional<u64> rev_src = transposed ? from_id : to_id, rev_dst = transposed ? to_id : from_id;
        process(pairs_of(rev_src, rev_dst, true), !transposed, rev_src_l, rev_dst_l);
    }
    // anonymous bidirectional CT over an anonymous bidirectional child: one row per (scan source, final dest) across the
    // expand_row calls of a batch — swap_remove, as the reference does, so the surviving order is the reference's (:948-970)
    if (dedup && dedup_src) {
        size_t i = start;
        while (i < out.size()) {
            if (!dedup->seen.insert({*dedup_src, out[i][1]}).second) {
                out[i] = out.back();
                out.pop_back();
                continue;
            }
            ++i;
        }
    }
}

// ---- CondVarLenTraverse ---------------------------------------------------------------------------------
namespace {
struct VlEdge { u64 src, dst, id; };

// Graph::get_node_relationships_by_type (graph.rs:1797-1835): per tensor (all of them, or the known ones among `types`,
// in order) the outgoing half — Tensor::iter(id, id, false): (dst, edge id) ascending — then the incoming half —
// Tensor::iter(id, id, true) over `mt`: (src, edge id) ascending — without the self-loops the outgoing half already gave
std::vector<VlEdge> node_relationships(const Graph& g, u64 id, const std::vector<u64>& tids, bool outgoing, bool incoming) {
    std::vector<VlEdge> out;
    for (u64 t : tids) {
        const Tensor& T = g.relationship_tensors()[t];
        if (outgoing)
            for (auto& e : T.structural_iter(id, id))
                for (u64 eid : T.get(e.row, e.col)) out.push_back({e.row, e.col, eid});
        if (incoming)
            for (auto& e : T.matrix_t().iter(id, id)) {          // row = dst (= id), col = src
                if (outgoing && e.col == id) continue;
                for (u64 eid : T.get(e.col, id)) out.push_back({e.col, id, eid});
            }
    }
    return out;
}
}  // namespace

void CondVarLenTraverseOp::expand_row(const Graph& g, u64 start, std::optional<u64> dest, std::vector<VarLenRow>& out,
                                      VarLenStats* stats) const {
    VarLenStats local;
    VarLenStats& st = stats ? *stats : local;
    // types: none named = every tensor; named = the known ones (filter_map, graph.rs:1803-1810)
    std::vector<u64> tids;
    if (types.empty())
        for (u64 t = 0; t < g.relationship_tensors().size(); ++t) tids.push_back(t);
    else
        for (auto& t : types)
            if (auto id = g.type_id(t)) tids.push_back(*id);
    // destination labels resolved once per row; an unknown label suppresses every emission (:100-107)
    std::vector<LabelId> dl;
    bool label_missing = false;
    for (auto& l : dst_labels) {
        if (auto id = g.label_id(l)) dl.push_back(*id);
        else label_missing = true;
    }
    auto labels_ok = [&](u64 v) {
        for (LabelId l : dl)
            if (!g.node_has_label_id(v, l)) return false;
        return true;
    };
    const bool with_out = bidirectional || !reversed, with_in = bidirectional || reversed;
    auto emit = [&](u64 other, const std::vector<u64>& walk) {
        VarLenRow r;
        r.from = reversed ? other : start;
        r.to = reversed ? start : other;
        if (emit_path) {
            r.path = walk;
            if (reversed) std::reverse(r.path.begin(), r.path.end());   // path_value (:134-142)
        }
        out.push_back(std::move(r));
    };
    // 0-hop emission (:153-171)
    if (min_hops == 0 && (!dest || *dest == start) && !label_missing && labels_ok(start)) emit(start, {start});

    // reach[r] (r >= 1): nodes from which `dest` is reachable by a walk of 1..r steps in the direction the DFS moves.
    // Walking back from dest: the DFS follows A (outgoing), A' (reversed) or A + A' (bidirectional), so the sets grow by
    // products with A', A or A + A'.  Only worth it for a bound destination and a finite budget.
    std::vector<std::vector<uint64_t>> reach;
    const u64 n = g.node_cap();
    if (prune && dest && max_hops != UINT32_MAX && max_hops >= 2 && max_hops <= 64 && !tids.empty()) {
        // (build_adjacency_matrix takes the names: none = the adjacency of every type, unknown names are skipped)
        Matrix back = bidirectional ? g.build_symmetric_adjacency_matrix(types)
                                    : (reversed ? g.build_adjacency_matrix(types) : g.build_adjacency_matrix(types).transpose());
        Matrix f(g.ctx(), Type::Bool, 1, n);
        f.build({0}, {*dest});
        std::vector<uint64_t> acc((n + 63) / 64, 0);
        reach.resize(max_hops);                       // reach[r] for r in [1, max_hops - 1]
        for (uint32_t r = 1; r + 1 <= max_hops; ++r) {
            f.lmxm(back);
            ++st.reach_products;
            for (auto& e : f.iter(0, 0)) acc[e.col >> 6] |= 1ull << (e.col & 63);
            reach[r] = acc;
            if (f.nvals() == 0) {                     // the frontier died out: larger budgets add nothing
                for (uint32_t q = r + 1; q + 1 <= max_hops; ++q) reach[q] = acc;
                break;
            }
        }
    }
    auto can_reach = [&](u64 v, uint32_t budget) {
        if (reach.empty() || budget == 0) return true;
        if (budget >= reach.size()) budget = (uint32_t)reach.size() - 1;
        return ((reach[budget][v >> 6] >> (v & 63)) & 1ull) != 0;
    };

    struct Frame { u64 node; std::vector<u64> walk; std::vector<u64> used; uint32_t depth; };
    std::vector<Frame> stack;
    std::unordered_map<u64, std::vector<VlEdge>> adj_cache;          // per row, like the reference's (:115-116)
    stack.push_back({start, emit_path ? std::vector<u64>{start} : std::vector<u64>{}, {}, 0});
    std::vector<std::pair<u64, u64>> scratch;                        // (edge id, neighbour)
    while (!stack.empty()) {
        Frame fr = std::move(stack.back());
        stack.pop_back();
        const uint32_t hop = fr.depth + 1;
        if (hop > max_hops) continue;
        ++st.frames;
        auto it = adj_cache.find(fr.node);
        if (it == adj_cache.end()) it = adj_cache.emplace(fr.node, node_relationships(g, fr.node, tids, with_out, with_in)).first;
        scratch.clear();
        for (const VlEdge& e : it->second) {
            if (std::find(fr.used.begin(), fr.used.end(), e.id) != fr.used.end()) continue;   // relationship uniqueness
            if (reversed) { if (e.dst == fr.node) scratch.push_back({e.id, e.src}); }
            else if (e.src == fr.node) scratch.push_back({e.id, e.dst});
            else if (bidirectional && e.dst == fr.node) scratch.push_back({e.id, e.src});
        }
        for (auto& [eid, nb] : scratch) {
            const bool will_emit = hop >= min_hops && (!dest || *dest == nb) && !label_missing && labels_ok(nb);
            bool will_continue = hop < max_hops;
            if (will_continue && !can_reach(nb, max_hops - hop)) { will_continue = false; ++st.pruned; }
            if (!will_emit && !will_continue) continue;
            std::vector<u64> walk = fr.walk;
            if (emit_path) { walk.push_back(eid); walk.push_back(nb); }
            if (will_emit) emit(nb, walk);
            if (will_continue) {
                std::vector<u64> used = fr.used;
                used.push_back(eid);
                stack.push_back({nb, std::move(walk), std::move(used), hop});
            }
        }
    }
}

// ---- ExpandInto ---------------------------------------------------------------------------------------
static std::vector<u64> resolve_types(const Graph& g, const std::vector<std::string>& types) {
    std::vector<u64> tids;
    if (types.empty()) {
        for (u64 t = 0; t < g.relationship_tensors().size(); ++t) tids.push_back(t);
        return tids;
    }
    for (auto& t : types)
        if (auto id = g.type_id(t)) tids.push_back(*id);
    return tids;
}

std::vector<std::array<u64, 3>> ExpandIntoOp::expand_row(const Graph& g, u64 src, u64 dst,
                                                         const std::vector<u64>& used_edges) const {
    std::vector<std::array<u64, 3>> out;
    std::vector<std::pair<u64, u64>> pairs{{src, dst}};
    if (bidirectional && src != dst) pairs.push_back({dst, src});
    auto tids = resolve_types(g, types);
    for (auto& pr : pairs) {
        size_t before = out.size();
        for (u64 t : tids)
            for (u64 e : g.relationship_tensors()[t].get(pr.first, pr.second)) {
                if (std::find(used_edges.begin(), used_edges.end(), e) != used_edges.end()) continue;
                out.push_back({pr.first, pr.second, e});
            }
        if (!emit_relationship && out.size() > before + 1) out.resize(before + 1);   // one representative per pair
    }
    return out;
}

void ExpandIntoOp::expand_batch(const Graph& g, const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                                std::vector<std::vector<std::array<u64, 3>>>& out) const {
    const size_t k = srcs.size();
    out.assign(k, {});
    auto tids = resolve_types(g, types);
    // probe list: (src, dst) of every row, then (dst, src) of the rows that also look backwards
    std::vector<u64> ps(srcs), pd(dsts);
    std::vector<size_t> back_row;
    if (bidirectional)
        for (size_t i = 0; i < k; ++i)
            if (srcs[i] != dsts[i]) { ps.push_back(dsts[i]); pd.push_back(srcs[i]); back_row.push_back(i); }
    std::vector<std::vector<std::vector<u64>>> per_type(tids.size());
    for (size_t t = 0; t < tids.size(); ++t) g.relationship_tensors()[tids[t]].get_batch(ps, pd, per_type[t]);
    auto emit = [&](size_t row, size_t probe) {
        size_t before = out[row].size();
        for (size_t t = 0; t < tids.size(); ++t)
            for (u64 e : per_type[t][probe]) out[row].push_back({ps[probe], pd[probe], e});
        if (!emit_relationship && out[row].size() > before + 1) out[row].resize(before + 1);
    };
    for (size_t i = 0; i < k; ++i) emit(i, i);
    for (size_t b = 0; b < back_row.size(); ++b) emit(back_row[b], k + b);
}

// ---- algo.BFS --------------------------------------------------------------------------------------------
// The partitioned form of the search below: the adjacency is cut into nnz-balanced column slabs, one
// per context of `gang` (one context per GPU; several contexts on one device work too and are how this is tested),
// and ONE call — fgpu_bfs_dist_run — drives every rank's level kernels and the per-level frontier exchange inside
// libfgpu.so (RCCL when the contexts were joined by fgpu_comm_init_all, event-ordered peer copies otherwise).
// Everything goes through include/fgpu.h.  Slabs that live on another context travel through the host once per
// call (no cross-device snapshot copy in the ABI yet; a cache keyed on the adjacency snapshot is the obvious next step).
static void bfs_partitioned(const Graph& g, const std::vector<Context*>& gang, const Matrix& adj, const std::string& key,
                            u64 source, int64_t max_depth, bool want_edges, std::vector<int32_t>& level,
                            std::vector<int64_t>& parent) {
    const int nr = (int)gang.size();
    fgpu_ctx* c0 = g.ctx().raw();
    std::vector<fgpu_ctx*> raw(nr);
    for (int r = 0; r < nr; ++r) raw[r] = gang[r]->raw();
    std::shared_ptr<Graph::BfsGangCache> gc = g.bfs_gang_cache_;
    if (!gc || gc->key != key || gc->gang != raw || gc->plans.empty() || gc->adj.snapshot() != adj.snapshot()) {
        // a new adjacency, filter or gang: cut the slabs again (the previous set is released with its last user)
        gc = std::make_shared<Graph::BfsGangCache>(adj);
        gc->key = key;
        gc->gang = raw;
        gc->splits.assign((size_t)nr + 1, 0);
        gc->slabs.assign(nr, nullptr);
        gc->slabs_t.assign(nr, nullptr);
        gc->plans.assign(nr, nullptr);
        check(fgpu_mat_balanced_splits(c0, adj.snapshot(), nr, gc->splits.data()), "fgpu_mat_balanced_splits");
        const u64 n = g.node_cap();
        for (int r = 0; r < nr; ++r) {
            fgpu_ctx* cr = raw[r];
            fgpu_mat* local = nullptr;
            check(fgpu_mat_col_slab(c0, &local, adj.snapshot(), gc->splits[r], gc->splits[r + 1] < n ? gc->splits[r + 1] : n),
                  "fgpu_mat_col_slab");
            if (cr == c0) {
                gc->slabs[r] = local;
            } else {
                u64 *rp = nullptr, *ci = nullptr, nnz = 0;
                fgpu_info i = fgpu_mat_export_csr(c0, local, &rp, &ci, nullptr, &nnz);
                fgpu_mat_free(local);
                check(i, "GxB_unload_Matrix_into_Container");
                i = fgpu_mat_from_csr(cr, &gc->slabs[r], n, n, nnz, rp, 64, ci, 64, nullptr, nullptr, 0);
                fgpu_free(c0, rp);
                fgpu_free(c0, ci);
                check(i, "GxB_load_Matrix_from_Container");
            }
            check(fgpu_mat_transpose(cr, &gc->slabs_t[r], gc->slabs[r]), "GrB_transpose");
            check(fgpu_bfs_plan_create_slab(cr, &gc->plans[r], gc->slabs[r], gc->slabs_t[r], r, nr, gc->splits.data()),
                  "fgpu_bfs_plan_create_slab");
        }
        g.bfs_gang_cache_ = gc;            // (an exception above leaves the old cache in place; gc frees what it built)
        Graph::register_gang_cache(gc);
    }
    check(fgpu_bfs_dist_run(gc->plans.data(), nr, source, max_depth < 0 ? -1 : max_depth, want_edges ? 1 : 0),
          "LAGr_BreadthFirstSearch (partitioned)");
    for (int r = 0; r < nr; ++r)   // every rank fills its own range [splits[r], splits[r+1])
        check(fgpu_bfs_fetch(gc->plans[r], level.data(), want_edges ? parent.data() : nullptr), "fgpu_bfs_fetch");
}

BfsResult algo_bfs(const Graph& g, std::optional<u64> source, int64_t max_depth,
                   const std::optional<std::string>& rel_type, bool want_edges, const std::vector<Context*>* gang) {
    BfsResult res;
    const u64 n = g.node_cap();
    if (!source) return res;                                             // NULL source: no row (:1029)
    // node_count() == 0 is tested BEFORE the deleted-source check (:1043-1048): with every node deleted the reference
    // returns the empty batch, not "Source node not found"
    const u64 live = n > g.deleted_nodes_count() ? n - g.deleted_nodes_count() : 0;
    if (live == 0) return res;
    if (g.is_node_deleted(*source)) throw GrbError(FGPU_INVALID, "Source node not found in graph");
    std::vector<std::string> types;
    if (rel_type) types.push_back(*rel_type);
    Matrix adj = g.build_adjacency_matrix(types);                        // graph.rs:3870-3894
    const std::string key = rel_type ? *rel_type : std::string();
    std::vector<int32_t> level_v;
    std::vector<int64_t> parent_v;
    const int32_t* level = nullptr;                    // what the result loop reads: the vectors, or the plan cache's pinned blocks
    const int64_t* parent = nullptr;
    const bool partitioned = gang && gang->size() > 1 && adj.nvals() > 0;
    if (partitioned) {
        level_v.resize(n);
        parent_v.resize(want_edges ? n : 0);
        bfs_partitioned(g, *gang, adj, key, *source, max_depth, want_edges, level_v, parent_v);
        level = level_v.data(); parent = parent_v.data();
    }
    std::shared_ptr<Graph::BfsPlanCache> pc;
    std::unique_lock<std::mutex> run_lock;                 // held across run, fetch AND the result loop (it reads the entry's pinned blocks)
    if (!partitioned) {
        std::lock_guard<std::mutex> cache_guard(g.bfs_cache_mu_);
        pc = g.bfs_cache_;
        if (!pc || pc->key != key || pc->adj.snapshot() != adj.snapshot()) {
            // a new adjacency (the layers changed, or another type): new plan; a clean committed graph keeps handing
            // out the same snapshot (VersionedMatrix::extract shares the base) and its cached transpose
            pc = std::make_shared<Graph::BfsPlanCache>(adj, adj.transpose());
            pc->key = key;
            // plans index dense row pointers: a hypersparse adjacency (an unknown type's empty matrix, a tiny delta-only
            // graph) goes through the one-shot entry point below, which densifies it
            if (fgpu_bfs_plan_create(g.ctx().raw(), &pc->plan, pc->adj.snapshot(), pc->adj_t.snapshot(), 0, 1) != FGPU_OK) {
                pc->plan = nullptr;
                pc.reset();
            }
            g.bfs_cache_ = pc;
        }
        if (pc) {
            run_lock = std::unique_lock<std::mutex>(pc->mu, std::try_to_lock);
            if (!run_lock.owns_lock()) pc.reset();         // another thread is searching on it: the one-shot path below
        }
    }
    if (partitioned) {
        // levels / parents were assembled from the ranks above
    } else if (pc) {
        fgpu_ctx* raw = g.ctx().raw();
        if (pc->pin_rows < n) {                                          // (a new cache entry, or the node capacity grew)
            if (pc->level_pin) { fgpu_free(pc->raw, pc->level_pin); pc->level_pin = nullptr; }
            if (pc->parent_pin) { fgpu_free(pc->raw, pc->parent_pin); pc->parent_pin = nullptr; }
            pc->raw = raw;
            pc->pin_rows = 0;
            check(fgpu_host_alloc(raw, n * sizeof(int32_t), (void**)&pc->level_pin), "fgpu_host_alloc");
            pc->pin_rows = n;
        }
        if (want_edges && !pc->parent_pin) check(fgpu_host_alloc(raw, pc->pin_rows * sizeof(int64_t), (void**)&pc->parent_pin), "fgpu_host_alloc");
        check(fgpu_bfs_run(pc->plan, *source, max_depth < 0 ? -1 : max_depth, want_edges ? 1 : 0), "LAGr_BreadthFirstSearch");
        check(fgpu_bfs_fetch(pc->plan, pc->level_pin, want_edges ? pc->parent_pin : nullptr), "LAGr_BreadthFirstSearch");
        level = pc->level_pin; parent = pc->parent_pin;
    } else {
        level_v.resize(n);
        parent_v.resize(want_edges ? n : 0);
        Matrix adj_t = adj.transpose();
        check(fgpu_bfs(g.ctx().raw(), adj.snapshot(), adj_t.snapshot(), *source, max_depth < 0 ? -1 : max_depth,
                       level_v.data(), want_edges ? parent_v.data() : nullptr, nullptr),
              "LAGr_BreadthFirstSearch");
        level = level_v.data(); parent = parent_v.data();
    }
    std::vector<u64> ps, pd;
    for (u64 v = 0; v < n; ++v) {
        if (level[v] < 0 || v == *source || g.is_node_deleted(v)) continue;
        if (want_edges) {
            u64 p = (u64)parent[v];
            if (g.is_node_deleted(p)) continue;
            ps.push_back(p);
            pd.push_back(v);
        }
        res.nodes.push_back(v);
    }
    if (want_edges && !res.nodes.empty()) {
        // edges[k] = first id of get_src_dest_relationships(parent, child, types), batched per type
        std::vector<u64> tids;
        if (rel_type) { if (auto id = g.type_id(*rel_type)) tids.push_back(*id); }
        else for (u64 t = 0; t < g.relationship_tensors().size(); ++t) tids.push_back(t);
        std::vector<std::optional<u64>> rep(ps.size());
        for (u64 t : tids) {
            std::vector<std::vector<u64>> ids;
            g.relationship_tensors()[t].get_batch(ps, pd, ids);
            for (size_t r = 0; r < ps.size(); ++r)
                if (!rep[r] && !ids[r].empty()) rep[r] = ids[r][0];
        }
        for (auto& e : rep)
            if (e) res.edges.push_back(*e);
    }
    res.has_row = !res.nodes.empty();
    return res;
}

// ---- algo.pageRank -------------------------------------------------------------------------------------
PageRankResult algo_pagerank(const Graph& g, const std::optional<std::string>& label,
                             const std::optional<std::string>& rel_type) {
    PageRankResult res;
    const u64 n = g.node_cap();
    u64 live = 0;
    for (u64 v = 0; v < n; ++v) live += g.is_node_deleted(v) ? 0 : 1;
    if (live == 0) return res;                                           // node_count() == 0 (:699-701)
    std::vector<std::string> types;
    if (rel_type) types.push_back(*rel_type);
    // a label that covers every live node is the unfiltered run (:711-713); otherwise the labelled nodes form a
    // compact graph of their own (:725-733) — here the induced subgraph selected by a bitmap
    std::vector<u64> active;
    bool filtered = false;
    if (label) {
        auto lid = g.label_id(*label);
        if (!lid) return res;                                            // no node carries an unknown label
        active = g.label_bitmap({*lid});
        u64 cnt = 0;
        for (u64 v = 0; v < n; ++v) {
            if (g.is_node_deleted(v)) active[v >> 6] &= ~(1ull << (v & 63));
            cnt += (active[v >> 6] >> (v & 63)) & 1ull;
        }
        filtered = cnt != live;
        if (cnt == 0) return res;
    }
    Matrix adj = g.build_adjacency_matrix(types);                        // graph.rs:3870-3894
    std::vector<float> score(n);
    int32_t iters = 0;
    // deleted ids stay in the unfiltered matrix as isolated vertices (n = node_count + deleted_nodes_count, :718-720)
    Matrix adj_t = adj.transpose();                                      // LAGraph_Cached_AT (:736-737); cached per snapshot
    check(fgpu_pagerank(g.ctx().raw(), adj.snapshot(), adj_t.snapshot(), filtered ? active.data() : nullptr, 0.85f, 1e-4f, 100,
                        score.data(), &iters),
          "LAGr_PageRank");
    for (u64 v = 0; v < n; ++v) {
        if (g.is_node_deleted(v)) continue;                              // :768-770
        if (filtered && !((active[v >> 6] >> (v & 63)) & 1ull)) continue;
        res.nodes.push_back(v);
        res.scores.push_back((double)score[v]);
    }
    return res;
}

}  // namespace falkor
