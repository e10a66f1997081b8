// capi.cpp — C surface of the host layer (include/falkor_host.h): handles, error codes, array hand-off.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <sstream>

#include "../../include/falkor_host.h"
#include "host.hpp"

using namespace falkor;

struct fh_ctx { Context c; explicit fh_ctx(int d) : c(d) {} };
struct fh_mat { Matrix m; };
struct fh_vm { VersionedMatrix v; };
struct fh_graph { Graph g; fh_graph(Context& c, u64 n) : g(c, n) {} };
struct fh_tn { Tensor t; };

static thread_local std::string g_err;

template <class F>
static int guard(F f) {
    try {
        return f();
    } catch (const GrbError& e) {
        g_err = e.what();
        return e.info ? (int)e.info : -1;
    } catch (const std::exception& e) {
        g_err = e.what();
        return FGPU_INVALID;
    }
}

static uint64_t* hand(const std::vector<u64>& v) {
    uint64_t* p = (uint64_t*)malloc((v.size() ? v.size() : 1) * sizeof(uint64_t));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(uint64_t));
    return p;
}

static std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    std::istringstream in(s);
    while (std::getline(in, cur, sep))
        if (!cur.empty()) out.push_back(cur);
    return out;
}

static thread_local uint64_t g_last_op_ns = 0;   // duration of the last timed operator inside this thread (fh_last_op_ns)

static CondTraverseOp parse_spec(const char* spec) {
    CondTraverseOp op;
    for (auto& kv : split(spec ? spec : "", ';')) {
        auto eq = kv.find('=');
        if (eq == std::string::npos) continue;
        std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        if (k == "src") op.src_labels = split(v, ',');
        else if (k == "hop") {
            Hop h;
            auto bar = v.find('|');
            h.types = split(v.substr(0, bar), ',');
            if (bar != std::string::npos) h.dst_labels = split(v.substr(bar + 1), ',');
            op.hops.push_back(h);
        } else if (k == "optional") op.optional = v == "1";
        else if (k == "bind") op.bind_relationship = v == "1";
        else if (k == "emit") op.emit_relationship = v == "1";
        else if (k == "bidir") op.bidirectional = v == "1";
        else if (k == "siblings") op.has_sibling_edges = v == "1";
        else if (k == "attrs") op.has_inline_attrs = v == "1";
        else if (k == "transposed") op.transposed = v == "1";
    }
    if (op.hops.empty()) op.hops.push_back(Hop{});
    return op;
}

extern "C" {

const char* fh_last_error(void) { return g_err.c_str(); }
void fh_free(void* p) { free(p); }

int fh_init(fh_ctx** ctx, int device) {
    return guard([&] { *ctx = new fh_ctx(device); return 0; });
}
void fh_finalize(fh_ctx* ctx) { delete ctx; }

int fh_should_fold(uint64_t d, uint64_t tx, uint64_t base) { return should_fold(d, tx, base) ? 1 : 0; }
int fh_should_fold_read(uint64_t d, uint64_t tx, uint64_t base) { return should_fold_read(d, tx, base) ? 1 : 0; }
int fh_delta_dominates_base(uint64_t d, uint64_t base) { return delta_dominates_base(d, base) ? 1 : 0; }
int fh_compound_key(uint64_t src, uint64_t dst, uint64_t* key) {
    return guard([&] { *key = compound_key(src, dst); return 0; });
}

// ---- Matrix ----------------------------------------------------------------------------------------
int fh_mat_new(fh_ctx* ctx, fh_mat** out, int type, uint64_t nrows, uint64_t ncols) {
    return guard([&] { *out = new fh_mat{Matrix(ctx->c, type ? Type::UInt64 : Type::Bool, nrows, ncols)}; return 0; });
}
void fh_mat_free(fh_mat* m) { delete m; }
int fh_mat_build(fh_mat* m, const uint64_t* rows, const uint64_t* cols, const uint64_t* vals, uint64_t n) {
    return guard([&] {
        std::vector<u64> r(rows, rows + n), c(cols, cols + n), v;
        if (vals) v.assign(vals, vals + n);
        m->m.build(r, c, vals ? &v : nullptr);
        return 0;
    });
}
int fh_mat_set(fh_mat* m, uint64_t i, uint64_t j, uint64_t v) { return guard([&] { m->m.set_element(i, j, v); return 0; }); }
int fh_mat_remove(fh_mat* m, uint64_t i, uint64_t j) { return guard([&] { m->m.remove_element(i, j); return 0; }); }
int fh_mat_get(fh_mat* m, uint64_t i, uint64_t j, uint64_t* v) {
    return guard([&] {
        auto x = m->m.get(i, j);
        if (!x) return 1;
        if (v) *v = *x;
        return 0;
    });
}
int fh_mat_nvals(fh_mat* m, uint64_t* out) { return guard([&] { *out = m->m.nvals(); return 0; }); }
int fh_mat_dims(fh_mat* m, uint64_t* nrows, uint64_t* ncols) {
    *nrows = m->m.nrows();
    *ncols = m->m.ncols();
    return 0;
}
int fh_mat_pending(fh_mat* m, int* out) { *out = m->m.pending() ? 1 : 0; return 0; }
int fh_mat_wait(fh_mat* m) { return guard([&] { m->m.wait(); return 0; }); }
int fh_mat_iter(fh_mat* m, uint64_t min_row, uint64_t max_row, uint64_t** rows, uint64_t** cols, uint64_t** vals,
                uint64_t* n) {
    return guard([&] {
        auto es = m->m.iter(min_row, max_row);
        std::vector<u64> r, c, v;
        for (auto& e : es) { r.push_back(e.row); c.push_back(e.col); v.push_back(e.val); }
        *rows = hand(r);
        *cols = hand(c);
        if (vals) *vals = hand(v);
        *n = es.size();
        return 0;
    });
}
// streaming cursor (matrix::Iter): new / seek / next-batch / free
struct fh_mat_cursor { MatrixIter it; };
int fh_mat_cursor_new(fh_mat* m, uint64_t min_row, uint64_t max_row, fh_mat_cursor** out) {
    return guard([&] { *out = new fh_mat_cursor{MatrixIter(m->m, min_row, max_row)}; return 0; });
}
int fh_mat_cursor_seek(fh_mat_cursor* c, uint64_t min_row, uint64_t max_row) {
    return guard([&] { c->it.seek(min_row, max_row); return 0; });
}
// up to `cap` entries into caller arrays; *n < cap means the cursor is exhausted
int fh_mat_cursor_next(fh_mat_cursor* c, uint64_t cap, uint64_t* rows, uint64_t* cols, uint64_t* vals, uint64_t* n) {
    return guard([&] {
        uint64_t k = 0;
        for (; k < cap; ++k) {
            auto e = c->it.next();
            if (!e) break;
            rows[k] = e->row;
            cols[k] = e->col;
            if (vals) vals[k] = e->val;
        }
        *n = k;
        return 0;
    });
}
void fh_mat_cursor_free(fh_mat_cursor* c) { delete c; }

int fh_mat_dup(fh_mat* m, fh_mat** out) { return guard([&] { *out = new fh_mat{m->m.dup()}; return 0; }); }
int fh_mat_transpose(fh_mat* m, fh_mat** out) { return guard([&] { *out = new fh_mat{m->m.transpose()}; return 0; }); }
int fh_mat_grown(fh_mat* m, uint64_t nrows, uint64_t ncols, fh_mat** out) {
    return guard([&] { *out = new fh_mat{m->m.grown(nrows, ncols)}; return 0; });
}
int fh_mat_resize(fh_mat* m, uint64_t nrows, uint64_t ncols) { return guard([&] { m->m.resize(nrows, ncols); return 0; }); }
int fh_mat_lmxm(fh_mat* self, fh_mat* b) { return guard([&] { self->m.lmxm(b->m); return 0; }); }
int fh_mat_rmxm(fh_mat* self, fh_mat* b) { return guard([&] { self->m.rmxm(b->m); return 0; }); }
int fh_mat_delta_lmxm(fh_mat* self, fh_mat* m, fh_mat* dp, fh_mat* dm) {
    return guard([&] { self->m.delta_lmxm(m->m, dp->m, dm->m); return 0; });
}
int fh_mat_intersection_nvals(fh_mat* a, fh_mat* b, uint64_t* out) {
    return guard([&] { *out = a->m.intersection_nvals(b->m); return 0; });
}

// ---- VersionedMatrix ------------------------------------------------------------------------------------
int fh_vm_new(fh_ctx* ctx, fh_vm** out, uint64_t nrows, uint64_t ncols) {
    return guard([&] { *out = new fh_vm{VersionedMatrix(ctx->c, nrows, ncols)}; return 0; });
}
int fh_vm_from_coo(fh_ctx* ctx, fh_vm** out, uint64_t nrows, uint64_t ncols, const uint64_t* rows,
                   const uint64_t* cols, uint64_t n) {
    return guard([&] {
        Matrix m(ctx->c, Type::Bool, nrows, ncols);
        m.build(std::vector<u64>(rows, rows + n), std::vector<u64>(cols, cols + n));
        *out = new fh_vm{VersionedMatrix::from_matrix(m)};
        return 0;
    });
}
void fh_vm_free(fh_vm* v) { delete v; }
int fh_vm_set(fh_vm* v, uint64_t i, uint64_t j) { return guard([&] { v->v.set(i, j, true); return 0; }); }
int fh_vm_remove(fh_vm* v, uint64_t i, uint64_t j) { return guard([&] { v->v.remove(i, j); return 0; }); }
int fh_vm_get(fh_vm* v, uint64_t i, uint64_t j) { return guard([&] { return v->v.get(i, j) ? 0 : 1; }); }
int fh_vm_nvals(fh_vm* v, uint64_t* out) { return guard([&] { *out = v->v.nvals(); return 0; }); }
int fh_vm_iter(fh_vm* v, uint64_t min_row, uint64_t max_row, uint64_t** rows, uint64_t** cols, uint64_t* n) {
    return guard([&] {
        auto es = v->v.iter(min_row, max_row);
        std::vector<u64> r, c;
        for (auto& e : es) { r.push_back(e.row); c.push_back(e.col); }
        *rows = hand(r);
        *cols = hand(c);
        *n = es.size();
        return 0;
    });
}
int fh_vm_set_all(fh_vm* v, const uint64_t* rows, const uint64_t* cols, uint64_t n, int is_new) {
    return guard([&] {
        std::vector<std::pair<u64, u64>> e(n);
        for (u64 k = 0; k < n; ++k) e[k] = {rows[k], cols[k]};
        v->v.set_all(e, is_new != 0);
        return 0;
    });
}
int fh_vm_remove_mask(fh_vm* v, const uint64_t* rows, const uint64_t* cols, uint64_t n) {
    return guard([&] {
        Matrix mask(v->v.m().ctx(), Type::Bool, v->v.nrows(), v->v.ncols());
        mask.build(std::vector<u64>(rows, rows + n), std::vector<u64>(cols, cols + n));
        v->v.remove_mask(mask);
        return 0;
    });
}
int fh_vm_dup(fh_vm* v, fh_vm** out) { return guard([&] { *out = new fh_vm{v->v.dup()}; return 0; }); }
int fh_vm_wait(fh_vm* v) { return guard([&] { v->v.wait(); return 0; }); }
int fh_vm_flush(fh_vm* v) { return guard([&] { v->v.flush(); return 0; }); }
int fh_vm_fold_oversized(fh_vm* v) { return guard([&] { v->v.fold_oversized(); return 0; }); }
int fh_vm_extract(fh_vm* v, fh_mat** out) { return guard([&] { *out = new fh_mat{v->v.extract()}; return 0; }); }
int fh_vm_transpose(fh_vm* v, fh_vm** out) { return guard([&] { *out = new fh_vm{v->v.transpose()}; return 0; }); }
int fh_vm_state(fh_vm* v, uint64_t out[4]) {
    return guard([&] {
        out[0] = v->v.m().nvals();
        out[1] = v->v.dp().nvals();
        out[2] = v->v.dm().nvals();
        out[3] = v->v.needs_flush() ? 1 : 0;
        return 0;
    });
}

// ---- Graph ------------------------------------------------------------------------------------------------
int fh_graph_new(fh_ctx* ctx, fh_graph** out, uint64_t node_cap) {
    return guard([&] { *out = new fh_graph(ctx->c, node_cap); return 0; });
}
void fh_graph_free(fh_graph* g) { delete g; }
int fh_graph_add_label(fh_graph* g, const char* name, uint64_t* id) { return guard([&] { *id = g->g.add_label(name); return 0; }); }
int fh_graph_add_type(fh_graph* g, const char* name, uint64_t* id) { return guard([&] { *id = g->g.add_type(name); return 0; }); }
int fh_graph_label_node(fh_graph* g, uint64_t node, uint64_t label_id) { return guard([&] { g->g.label_node(node, label_id); return 0; }); }
int fh_graph_delete_node(fh_graph* g, uint64_t node) { return guard([&] { g->g.delete_node(node); return 0; }); }
int fh_graph_create_edge(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t edge_id) {
    return guard([&] { g->g.create_edge(type_id, src, dst, edge_id); return 0; });
}
int fh_graph_create_edges(fh_graph* g, uint64_t type_id, const uint64_t* srcs, const uint64_t* dsts,
                          const uint64_t* ids, uint64_t n) {
    return guard([&] {
        g->g.create_edges(type_id, std::vector<u64>(srcs, srcs + n), std::vector<u64>(dsts, dsts + n),
                          std::vector<u64>(ids, ids + n));
        return 0;
    });
}
int fh_graph_delete_edge(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t edge_id) {
    return guard([&] { g->g.delete_edge(type_id, src, dst, edge_id); return 0; });
}
int fh_graph_commit(fh_graph* g) {
    return guard([&] {
        g->g.fold_oversized_deltas();
        g->g.new_version();
        return 0;
    });
}
int fh_graph_node_has_label(fh_graph* g, uint64_t node, uint64_t label_id) {
    return guard([&] { return g->g.node_has_label_id(node, label_id) ? 0 : 1; });
}
int fh_tensor_get(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t** ids, uint64_t* n) {
    return guard([&] {
        auto v = g->g.relationship_tensors().at(type_id).get(src, dst);
        *ids = hand(v);
        *n = v.size();
        return 0;
    });
}
int fh_tensor_edge_count(fh_graph* g, uint64_t type_id, uint64_t* out) {
    return guard([&] { *out = g->g.relationship_tensors().at(type_id).edge_count(); return 0; });
}
int fh_tensor_iter_edges(fh_graph* g, uint64_t type_id, uint64_t** srcs, uint64_t** dsts, uint64_t** ids, uint64_t* n) {
    return guard([&] {
        auto es = g->g.relationship_tensors().at(type_id).iter_edges();
        std::vector<u64> s, d, i;
        for (auto& e : es) { s.push_back(e.row); d.push_back(e.col); i.push_back(e.val); }
        *srcs = hand(s);
        *dsts = hand(d);
        *ids = hand(i);
        *n = es.size();
        return 0;
    });
}
int fh_tensor_state(fh_graph* g, uint64_t type_id, uint64_t out[5]) {
    return guard([&] {
        const Tensor& t = g->g.relationship_tensors().at(type_id);
        t.wait_fwd();
        out[0] = t.fwd_m().nvals();
        out[1] = t.fwd_dp().nvals();
        out[2] = t.fwd_dm().nvals();
        out[3] = t.multi_pairs();
        out[4] = t.matrix_t().nvals();
        return 0;
    });
}

// ---- a Tensor on its own (the unit tests of tensor.rs:1340-1669 drive one directly) ----------------------------
int fh_tn_new(fh_ctx* ctx, fh_tn** out, uint64_t nrows, uint64_t ncols) {
    return guard([&] { *out = new fh_tn{Tensor(ctx->c, nrows, ncols)}; return 0; });
}
void fh_tn_free(fh_tn* t) { delete t; }
int fh_tn_dup(fh_tn* t, fh_tn** out) {
    return guard([&] { *out = new fh_tn{t->t.dup()}; return 0; });
}
int fh_tn_set_all(fh_tn* t, const uint64_t* srcs, const uint64_t* dsts, const uint64_t* ids, uint64_t n) {
    return guard([&] {
        t->t.set_all_from_slices(std::vector<u64>(srcs, srcs + n), std::vector<u64>(dsts, dsts + n),
                                 std::vector<u64>(ids, ids + n));
        return 0;
    });
}
int fh_tn_remove_all(fh_tn* t, const uint64_t* rels, uint64_t n, uint64_t** emptied_src, uint64_t** emptied_dst,
                     uint64_t* n_emptied) {
    return guard([&] {
        std::vector<std::array<u64, 3>> r(n);
        for (u64 i = 0; i < n; ++i) r[i] = {rels[3 * i], rels[3 * i + 1], rels[3 * i + 2]};   // (edge id, src, dst)
        auto e = t->t.remove_all(r);
        std::vector<u64> es, ed;
        for (auto& p : e) { es.push_back(p.first); ed.push_back(p.second); }
        *emptied_src = hand(es);
        *emptied_dst = hand(ed);
        *n_emptied = e.size();
        return 0;
    });
}
int fh_tn_op(fh_tn* t, int op, uint64_t a, uint64_t b) {
    return guard([&] {
        switch (op) {
            case 0: t->t.flush(); break;
            case 1: t->t.fold_oversized(); break;
            case 2: t->t.wait(); break;
            case 3: t->t.wait_fwd(); break;
            case 4: t->t.resize(a, b); break;
            default: throw GrbError(FGPU_INVALID, "fh_tn_op: unknown op");
        }
        return 0;
    });
}
int fh_tn_get(fh_tn* t, uint64_t src, uint64_t dst, uint64_t** ids, uint64_t* n) {
    return guard([&] {
        auto v = t->t.get(src, dst);
        *ids = hand(v);
        *n = v.size();
        return 0;
    });
}
int fh_tn_probe(fh_tn* t, int which, uint64_t src, uint64_t dst, uint64_t* val) {
    return guard([&] {
        std::optional<u64> v;
        if (which == 0) v = t->t.eff_get(src, dst);                     // eff_get_for_test
        else if (which == 1) { t->t.fwd_m().wait(); v = t->t.fwd_m().get(src, dst); }
        else if (which == 2) { Matrix e = t->t.extract(); e.wait(); v = e.get(src, dst); }
        else throw GrbError(FGPU_INVALID, "fh_tn_probe: which must be 0 (effective), 1 (m) or 2 (extract)");
        if (!v) return (int)FGPU_NO_VALUE;
        *val = *v;
        return 0;
    });
}
int fh_tn_state(fh_tn* t, uint64_t out[8]) {
    return guard([&] {
        out[7] = t->t.fwd_m().pending() ? 1 : 0;     // read BEFORE anything waits (resize_leaves_base_materialized)
        t->t.wait_fwd();
        out[0] = t->t.fwd_m().nvals();
        out[1] = t->t.fwd_dp().nvals();
        out[2] = t->t.fwd_dm().nvals();
        out[3] = t->t.multi_pairs();
        out[4] = t->t.matrix_t().extract().nvals();
        out[5] = t->t.me_nvals();
        out[6] = t->t.edge_count();
        return 0;
    });
}

int fh_tn_encode(fh_tn* t, uint8_t** bytes, uint64_t* len) {
    return guard([&] {
        ByteWriter w;
        t->t.encode(w);
        uint8_t* out = (uint8_t*)malloc(w.buf.size() ? w.buf.size() : 1);
        if (!out) throw GrbError(FGPU_OOM, "out of host memory");
        memcpy(out, w.buf.data(), w.buf.size());
        *bytes = out;
        *len = w.buf.size();
        return 0;
    });
}
int fh_tn_decode(fh_ctx* ctx, const uint8_t* bytes, uint64_t len, fh_tn** out, uint64_t* consumed) {
    return guard([&] {
        ByteReader r(bytes, len);
        Tensor t = Tensor::decode(ctx->c, r);
        t.rebuild_backward();                              // what the reference's caller does after decode (:1193-1195)
        *out = new fh_tn{std::move(t)};
        *consumed = r.pos;
        return 0;
    });
}

int fh_graph_layer_iter(fh_graph* g, int64_t type_id, int which, uint64_t** rows, uint64_t** cols, uint64_t** vals,
                        uint64_t* n) {
    return guard([&] {
        const Matrix* m;
        if (type_id < 0) {
            const VersionedMatrix& a = g->g.adjacency_matrix();
            a.wait();
            m = which == 0 ? &a.m() : which == 1 ? &a.dp() : &a.dm();
        } else {
            const Tensor& t = g->g.relationship_tensors().at((size_t)type_id);
            t.wait_fwd();
            m = which == 0 ? &t.fwd_m() : which == 1 ? &t.fwd_dp() : &t.fwd_dm();
        }
        auto es = m->iter(0, ~0ull);
        std::vector<u64> r, c, v;
        for (auto& e : es) { r.push_back(e.row); c.push_back(e.col); v.push_back(e.val); }
        *rows = hand(r);
        *cols = hand(c);
        *vals = hand(v);
        *n = es.size();
        return 0;
    });
}

// ---- operators -----------------------------------------------------------------------------------------------
static Value to_value(int64_t x) {
    if (x >= 0) return Value::node((u64)x);
    if (x == -2) return Value::null();
    return Value{};
}

uint64_t fh_last_op_ns(void) { return g_last_op_ns; }

int fh_cond_traverse_eligible(const char* spec) { return parse_spec(spec).batched_eligible() ? 1 : 0; }

int fh_cond_traverse_batch(fh_graph* g, const char* spec, const int64_t* src, const int64_t* to_bound, uint64_t k,
                           int* batched, uint64_t** out_row, uint64_t** out_dest, int64_t** out_edge, uint64_t* n,
                           uint64_t** null_rows, uint64_t* n_null, uint64_t* flops) {
    return guard([&] {
        CondTraverseOp op = parse_spec(spec);
        std::vector<Value> s(k), tb(k);
        for (u64 i = 0; i < k; ++i) {
            s[i] = to_value(src[i]);
            if (to_bound) tb[i] = to_value(to_bound[i]);
        }
        // result columns keep their capacity between calls, as an operator's own batch buffers would: a fresh
        // 40 MB vector per call is mostly page faults (29 ms against 8 for a 4.9 M-row batch)
        static thread_local ExpandedRows rows;
        static thread_local std::vector<u64> nulls;
        u64 fl = 0;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok = op.expand_batch(g->g, s, to_bound ? &tb : nullptr, rows, nulls, &fl);
        g_last_op_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        *batched = ok ? 1 : 0;
        const size_t n_rows = rows.size();
        int64_t* e = (int64_t*)malloc((rows.size() ? rows.size() : 1) * sizeof(int64_t));
        if (rows.edge.empty()) memset(e, 0xFF, (rows.size() ? rows.size() : 1) * sizeof(int64_t));   // -1: none
        else memcpy(e, rows.edge.data(), rows.size() * sizeof(int64_t));
        if (rows.pinned()) {                                 // the harness wants malloc'ed u64 arrays: copy out of the pinned columns
            const size_t nn = rows.size();
            u64* r_ = (u64*)malloc((nn ? nn : 1) * sizeof(u64));
            u64* d_ = (u64*)malloc((nn ? nn : 1) * sizeof(u64));
            for (size_t q = 0; q < nn; ++q) { r_[q] = rows.row_at(q); d_[q] = rows.dest_at(q); }
            *out_row = r_;
            *out_dest = d_;
            rows.release_pinned();                           // (not kept across calls: the context may be gone before this thread is)
        } else {
            *out_row = hand(rows.active_row);
            *out_dest = hand(rows.dest);
        }
        *out_edge = e;
        *n = n_rows;
        *null_rows = hand(nulls);
        *n_null = nulls.size();
        if (flops) *flops = fl;
        return 0;
    });
}

int fh_cond_traverse_rows(fh_graph* g, const char* spec, const int64_t* from_ids, const int64_t* to_ids,
                          const int64_t* dedup_src, uint64_t k, int transposed, const uint64_t* used_edges,
                          uint64_t n_used, uint64_t** out_row, uint64_t** out_from, uint64_t** out_to,
                          uint64_t** out_edge, uint64_t* n) {
    return guard([&] {
        CondTraverseOp op = parse_spec(spec);
        CondTraverseOp::BidirDedup dd;                       // one input batch: fresh dedup state (:1268-1270)
        std::vector<u64> used(used_edges, used_edges + (used_edges ? n_used : 0));
        std::vector<u64> r, f, t, e;
        for (u64 i = 0; i < k; ++i) {
            if (from_ids[i] == -2 || to_ids[i] == -2) continue;   // bound to a non-node: no rows (:792-803)
            std::vector<std::array<u64, 3>> out;
            const bool dedup = dedup_src != nullptr;
            op.expand_row(g->g, from_ids[i] >= 0 ? std::optional<u64>((u64)from_ids[i]) : std::nullopt,
                          to_ids[i] >= 0 ? std::optional<u64>((u64)to_ids[i]) : std::nullopt, transposed != 0, used, out,
                          dedup ? &dd : nullptr,
                          dedup && dedup_src[i] >= 0 ? std::optional<u64>((u64)dedup_src[i]) : std::nullopt);
            for (auto& x : out) { r.push_back(i); f.push_back(x[0]); t.push_back(x[1]); e.push_back(x[2]); }
        }
        *out_row = hand(r);
        *out_from = hand(f);
        *out_to = hand(t);
        *out_edge = hand(e);
        *n = r.size();
        return 0;
    });
}

int fh_cond_traverse_row(fh_graph* g, const char* spec, int64_t from_id, int64_t to_id, int transposed,
                         uint64_t** out_from, uint64_t** out_to, uint64_t** out_edge, uint64_t* n) {
    uint64_t* rows = nullptr;
    int rc = fh_cond_traverse_rows(g, spec, &from_id, &to_id, nullptr, 1, transposed, nullptr, 0, &rows, out_from, out_to,
                                   out_edge, n);
    if (rc == 0) fh_free(rows);
    return rc;
}

int fh_expand_into(fh_graph* g, const char* types, int bidirectional, int emit_relationship, int batched,
                   const uint64_t* srcs, const uint64_t* dsts, uint64_t k, uint64_t** out_row, uint64_t** out_src,
                   uint64_t** out_dst, uint64_t** out_edge, uint64_t* n) {
    return guard([&] {
        ExpandIntoOp op;
        op.types = split(types ? types : "", ',');
        op.bidirectional = bidirectional != 0;
        op.emit_relationship = emit_relationship != 0;
        std::vector<std::vector<std::array<u64, 3>>> per_row(k);
        if (batched) {
            op.expand_batch(g->g, std::vector<u64>(srcs, srcs + k), std::vector<u64>(dsts, dsts + k), per_row);
        } else {
            for (u64 i = 0; i < k; ++i) per_row[i] = op.expand_row(g->g, srcs[i], dsts[i]);
        }
        std::vector<u64> r, s, d, e;
        for (u64 i = 0; i < k; ++i)
            for (auto& x : per_row[i]) { r.push_back(i); s.push_back(x[0]); d.push_back(x[1]); e.push_back(x[2]); }
        *out_row = hand(r);
        *out_src = hand(s);
        *out_dst = hand(d);
        *out_edge = hand(e);
        *n = r.size();
        return 0;
    });
}

int fh_var_len_traverse(fh_graph* g, const char* types, const char* dst_labels, int reversed, int bidirectional,
                        uint32_t min_hops, uint32_t max_hops, uint64_t start, int64_t dest, int emit_path, int prune,
                        uint64_t** out_from, uint64_t** out_to, uint64_t** path, uint64_t** path_off, uint64_t* n,
                        uint64_t stats[3]) {
    return guard([&] {
        CondVarLenTraverseOp op;
        op.types = split(types ? types : "", ',');
        op.dst_labels = split(dst_labels ? dst_labels : "", ',');
        op.reversed = reversed != 0;
        op.bidirectional = bidirectional != 0;
        op.min_hops = min_hops;
        op.max_hops = max_hops;
        op.emit_path = emit_path != 0;
        op.prune = prune != 0;
        std::vector<VarLenRow> rows;
        VarLenStats st;
        op.expand_row(g->g, start, dest >= 0 ? std::optional<u64>((u64)dest) : std::nullopt, rows, &st);
        std::vector<u64> f, t, p, off{0};
        for (auto& r : rows) {
            f.push_back(r.from);
            t.push_back(r.to);
            p.insert(p.end(), r.path.begin(), r.path.end());
            off.push_back(p.size());
        }
        *out_from = hand(f);
        *out_to = hand(t);
        *path = hand(p);
        *path_off = hand(off);
        *n = rows.size();
        if (stats) { stats[0] = st.frames; stats[1] = st.pruned; stats[2] = st.reach_products; }
        return 0;
    });
}

int fh_algo_bfs(fh_graph* g, int64_t source, int64_t max_depth, const char* rel_type, int want_edges, int* has_row,
                uint64_t** nodes, uint64_t* n_nodes, uint64_t** edges, uint64_t* n_edges) {
    return guard([&] {
        const auto t0 = std::chrono::steady_clock::now();
        BfsResult r = algo_bfs(g->g, source >= 0 ? std::optional<u64>((u64)source) : std::nullopt, max_depth,
                               rel_type ? std::optional<std::string>(rel_type) : std::nullopt, want_edges != 0);
        g_last_op_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        *has_row = r.has_row ? 1 : 0;
        *nodes = hand(r.nodes);
        *n_nodes = r.nodes.size();
        *edges = hand(r.edges);
        *n_edges = r.edges.size();
        return 0;
    });
}

int fh_algo_bfs_multi(fh_graph* g, fh_ctx* const* gang, int n_gang, int64_t source, int64_t max_depth,
                      const char* rel_type, int want_edges, int* has_row, uint64_t** nodes, uint64_t* n_nodes,
                      uint64_t** edges, uint64_t* n_edges) {
    return guard([&] {
        std::vector<Context*> cs;
        for (int i = 0; i < n_gang; ++i) cs.push_back(&gang[i]->c);
        BfsResult r = algo_bfs(g->g, source >= 0 ? std::optional<u64>((u64)source) : std::nullopt, max_depth,
                               rel_type ? std::optional<std::string>(rel_type) : std::nullopt, want_edges != 0, &cs);
        *has_row = r.has_row ? 1 : 0;
        *nodes = hand(r.nodes);
        *n_nodes = r.nodes.size();
        *edges = hand(r.edges);
        *n_edges = r.edges.size();
        return 0;
    });
}

// ---- v19 matrix payload (serialize.cpp) -------------------------------------------------------------------
// CPU-only parse of a container payload: dims, flags and the index / value arrays it carries
int fh_container_parse(const uint8_t* bytes, uint64_t len, uint64_t* dims /* nrows, ncols, nvals, hyper, valued, consumed */,
                       uint64_t** p, uint64_t* np, uint64_t** h, uint64_t* nh, uint64_t** i, uint64_t** x) {
    return guard([&] {
        ByteReader r(bytes, len);
        ContainerData c = parse_container(r);
        dims[0] = c.nrows; dims[1] = c.ncols; dims[2] = c.nvals; dims[3] = c.hyper ? 1 : 0; dims[4] = c.valued ? 1 : 0;
        dims[5] = r.pos;
        *p = hand(c.p); *np = c.p.size();
        *h = hand(c.h); *nh = c.h.size();
        *i = hand(c.i);
        *x = hand(c.x);
        return 0;
    });
}
int fh_mat_decode(fh_ctx* ctx, const uint8_t* bytes, uint64_t len, fh_mat** out, uint64_t* consumed) {
    return guard([&] {
        ByteReader r(bytes, len);
        *out = new fh_mat{Matrix::decode(ctx->c, r)};
        if (consumed) *consumed = r.pos;
        return 0;
    });
}
int fh_mat_encode(fh_mat* m, uint8_t** bytes, uint64_t* len) {
    return guard([&] {
        ByteWriter w;
        m->m.encode(w);
        uint8_t* out = (uint8_t*)malloc(w.buf.size() ? w.buf.size() : 1);
        memcpy(out, w.buf.data(), w.buf.size());
        *bytes = out;
        *len = w.buf.size();
        return 0;
    });
}

// build_adjacency_matrix / build_symmetric_adjacency_matrix (graph.rs:3870-3907); types = comma list, "" = all
int fh_graph_build_adjacency(fh_graph* g, const char* types, int symmetric, fh_mat** out) {
    return guard([&] {
        std::vector<std::string> ts;
        for (auto& t : split(types ? types : "", ','))
            if (!t.empty()) ts.push_back(t);
        *out = new fh_mat{symmetric ? g->g.build_symmetric_adjacency_matrix(ts) : g->g.build_adjacency_matrix(ts)};
        return 0;
    });
}

// plan text in, plan text out after fuse_anonymous_traverse; *spec (nullable) receives the runtime spec string
// (the fh_cond_traverse_batch format) of the CondTraverse node `lower_id` of the RESULT, or "" if lower_id < 0
int fh_plan_fuse(const char* plan_text, int lower_id, char** out_text, char** spec) {
    return guard([&] {
        Plan plan = parse_plan(plan_text ? plan_text : "");
        fuse_anonymous_traverse(plan);
        const std::string txt = print_plan(plan);
        *out_text = strdup(txt.c_str());
        if (spec) {
            std::string sp;
            if (lower_id >= 0 && (size_t)lower_id < plan.ops.size() && plan.ops[lower_id].kind == PlanOp::CondTraverse) {
                CondTraverseOp op = lower_cond_traverse(plan.ops[lower_id]);
                sp = "src=";
                for (size_t i = 0; i < op.src_labels.size(); ++i) sp += (i ? "," : "") + op.src_labels[i];
                for (auto& h : op.hops) {
                    sp += ";hop=";
                    for (size_t i = 0; i < h.types.size(); ++i) sp += (i ? "," : "") + h.types[i];
                    sp += "|";
                    for (size_t i = 0; i < h.dst_labels.size(); ++i) sp += (i ? "," : "") + h.dst_labels[i];
                }
                sp += std::string(";optional=") + (op.optional ? "1" : "0") + ";bind=" + (op.bind_relationship ? "1" : "0") +
                      ";emit=" + (op.emit_relationship ? "1" : "0") + ";bidir=" + (op.bidirectional ? "1" : "0") +
                      ";siblings=" + (op.has_sibling_edges ? "1" : "0") + ";attrs=" + (op.has_inline_attrs ? "1" : "0");
            }
            *spec = strdup(sp.c_str());
        }
        return 0;
    });
}

int fh_algo_pagerank(fh_graph* g, const char* label, const char* rel_type, uint64_t** nodes, double** scores,
                     uint64_t* n) {
    return guard([&] {
        const auto t0 = std::chrono::steady_clock::now();
        PageRankResult r = algo_pagerank(g->g, label ? std::optional<std::string>(label) : std::nullopt,
                                         rel_type ? std::optional<std::string>(rel_type) : std::nullopt);
        g_last_op_ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        *nodes = hand(r.nodes);
        double* sc = (double*)malloc((r.scores.size() ? r.scores.size() : 1) * sizeof(double));
        if (sc && !r.scores.empty()) memcpy(sc, r.scores.data(), r.scores.size() * sizeof(double));
        *scores = sc;
        *n = r.nodes.size();
        return 0;
    });
}

}  // extern "C"
