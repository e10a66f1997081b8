// matrix.cpp — Matrix<T> over libfgpu (mirrors graph/src/graph/graphblas/matrix.rs).
#include <string.h>

#include <atomic>

#include <unordered_map>

#include "host.hpp"

namespace falkor {

void check(fgpu_info i, const char* where) {
    if (i == FGPU_OK) return;
    const char* msg = fgpu_last_error();
    throw GrbError(i, std::string(where) + ": " + (msg ? msg : "error") + " (info " + std::to_string(i) + ")");
}

// ---- Context -------------------------------------------------------------------------------
Context::Context(int device) {
    // matrix::init propagates failure as Err(String) so the module refuses to load (matrix.rs:109-134)
    check(fgpu_init(&ctx_, device, nullptr, nullptr), "matrix::init");
}
// partitioned-BFS caches (Graph::BfsGangCache) hold slabs and plans in SEVERAL contexts: a context that goes away first
// releases every cache it takes part in
namespace {
std::mutex g_gang_mu;
std::vector<std::weak_ptr<Graph::BfsGangCache>> g_gang_caches;
}  // namespace
void Graph::register_gang_cache(const std::shared_ptr<BfsGangCache>& c) {
    std::lock_guard<std::mutex> g(g_gang_mu);
    size_t k = 0;
    for (auto& w : g_gang_caches)
        if (!w.expired()) g_gang_caches[k++] = w;
    g_gang_caches.resize(k);
    g_gang_caches.push_back(c);
}
Context::~Context() {
    if (!ctx_) return;
    std::vector<std::shared_ptr<Graph::BfsGangCache>> mine;
    {
        std::lock_guard<std::mutex> g(g_gang_mu);
        for (auto& w : g_gang_caches)
            if (auto c = w.lock()) {
                std::lock_guard<std::mutex> cg(c->mu);
                for (fgpu_ctx* x : c->gang)
                    if (x == ctx_) { mine.push_back(c); break; }
            }
    }
    for (auto& c : mine) c->release();
    fgpu_finalize(ctx_);
}

// ---- Matrix state ----------------------------------------------------------------------------
namespace {
struct Snap {  // one immutable device snapshot; shared between a matrix and its dup()s until either mutates
    fgpu_mat* h = nullptr;
    // the transpose of an immutable snapshot is itself immutable: computed once, shared by every Matrix::transpose()
    // of every handle on this snapshot (the reference keeps `mt` per tensor for the same reason, tensor.rs:814-816;
    // algo.BFS / algo.pageRank on a clean committed graph would otherwise pay a 13 ms transpose per call at RMAT-22)
    std::shared_ptr<Snap> transposed;
    std::mutex tmu;
    explicit Snap(fgpu_mat* m) : h(m) {}
    ~Snap() { if (h) fgpu_mat_free(h); }
    Snap(const Snap&) = delete;
    Snap& operator=(const Snap&) = delete;
};
}  // namespace

struct Matrix::State {
    Context* ctx;
    Type type;
    u64 nrows, ncols;
    std::shared_ptr<Snap> snap;
    // pending tuples / zombies: coordinate -> value to store, or nullopt to delete (last write wins).  Hashed, not ordered:
    // wait() hands the tuples to fgpu_mat_from_coo, which sorts on the device, and a bulk load queues one entry per edge in
    // three of these logs (67 M edges through ordered maps were most of a three-minute load)
    struct CoordHash {
        size_t operator()(const std::pair<u64, u64>& p) const noexcept {
            u64 x = (p.first + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull ^ (p.second * 0x94D049BB133111EBull);
            x ^= x >> 31;
            return (size_t)(x * 0xD6E8FEB86659FD93ull);
        }
    };
    std::unordered_map<std::pair<u64, u64>, std::optional<u64>, CoordHash> pend;
    // Matrix::wait's protocol (matrix.rs:781-796): `has_pending` is what concurrent READERS look at (acquire); only
    // the one that wins `lock` touches `pend` / `snap`, and it publishes the new snapshot with a release store of
    // the flag.  Writers (set / remove / build ...) hold the matrix exclusively, as `&mut self` makes them in Rust.
    std::atomic<bool> has_pending{false};
    std::mutex lock;
};

static fgpu_mat* new_empty(Context& ctx, u64 nrows, u64 ncols) {
    fgpu_mat* h = nullptr;
    check(fgpu_mat_new(ctx.raw(), &h, nrows, ncols), "GrB_Matrix_new");
    return h;
}

Matrix::Matrix(Context& ctx, Type t, u64 nrows, u64 ncols) : s_(std::make_shared<State>()) {
    s_->ctx = &ctx;
    s_->type = t;
    s_->nrows = nrows;
    s_->ncols = ncols;
    s_->snap = std::make_shared<Snap>(new_empty(ctx, nrows, ncols));
}

Matrix Matrix::adopt(Context& ctx, Type t, fgpu_mat* snapshot) {
    auto s = std::make_shared<State>();
    s->ctx = &ctx;
    s->type = t;
    check(fgpu_mat_nrows(snapshot, &s->nrows), "GrB_Matrix_nrows");
    check(fgpu_mat_ncols(snapshot, &s->ncols), "GrB_Matrix_ncols");
    s->snap = std::make_shared<Snap>(snapshot);
    return Matrix(std::move(s));
}

Type Matrix::type() const { return s_->type; }
Context& Matrix::ctx() const { return *s_->ctx; }
u64 Matrix::nrows() const { return s_->nrows; }
u64 Matrix::ncols() const { return s_->ncols; }
bool Matrix::pending() const { return s_->has_pending.load(std::memory_order_acquire); }
bool Matrix::is_synced() const { return !s_->has_pending.load(std::memory_order_acquire); }

void Matrix::replace(fgpu_mat* fresh) const {
    s_->snap = std::make_shared<Snap>(fresh);
    s_->pend.clear();
    s_->has_pending.store(false, std::memory_order_release);
}

void Matrix::wait() const {
    if (!s_->has_pending.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> g(s_->lock);
    if (!s_->has_pending.load(std::memory_order_relaxed)) return;
    std::vector<u64> ar, ac, av, dr, dc;
    for (auto& kv : s_->pend) {
        if (kv.second) {
            ar.push_back(kv.first.first);
            ac.push_back(kv.first.second);
            av.push_back(*kv.second);
        } else {
            dr.push_back(kv.first.first);
            dc.push_back(kv.first.second);
        }
    }
    fgpu_ctx* c = s_->ctx->raw();
    fgpu_mat *adds = nullptr, *dels = nullptr, *out = nullptr;
    fgpu_info i = FGPU_OK;
    if (!ar.empty())
        i = fgpu_mat_from_coo(c, &adds, s_->nrows, s_->ncols, ar.data(), ac.data(),
                              s_->type == Type::UInt64 ? av.data() : nullptr, ar.size());
    if (i == FGPU_OK && !dr.empty())
        i = fgpu_mat_from_coo(c, &dels, s_->nrows, s_->ncols, dr.data(), dc.data(), nullptr, dr.size());
    // GrB_Matrix_wait(MATERIALIZE): zombies leave, pending tuples land (a stored value is overwritten)
    if (i == FGPU_OK) i = fgpu_mat_merge(c, &out, s_->snap->h, adds, dels, 0);
    if (adds) fgpu_mat_free(adds);
    if (dels) fgpu_mat_free(dels);
    check(i, "GrB_Matrix_wait");
    s_->snap = std::make_shared<Snap>(out);
    s_->pend.clear();
    s_->has_pending.store(false, std::memory_order_release);
}

const fgpu_mat* Matrix::snapshot() const {
    wait();
    return s_->snap->h;
}

u64 Matrix::nvals() const {
    u64 n = 0;
    check(fgpu_mat_nvals(snapshot(), &n), "GrB_Matrix_nvals");
    return n;
}

void Matrix::build(const std::vector<u64>& rows, const std::vector<u64>& cols, const std::vector<u64>* vals) {
    if (rows.size() != cols.size() || (vals && vals->size() != rows.size()))
        throw GrbError(FGPU_INVALID, "Matrix::build: slices differ in length");
    if (nvals() != 0) throw GrbError(FGPU_INVALID, "Matrix::build: output already has entries (GrB_OUTPUT_NOT_EMPTY)");
    fgpu_mat* h = nullptr;
    const u64* v = (s_->type == Type::UInt64 && vals) ? vals->data() : nullptr;
    std::vector<u64> ones;
    if (s_->type == Type::UInt64 && !vals) {
        ones.assign(rows.size(), 1);
        v = ones.data();
    }
    check(fgpu_mat_from_coo(s_->ctx->raw(), &h, s_->nrows, s_->ncols, rows.data(), cols.data(), v, rows.size()),
          "GrB_Matrix_build");
    replace(h);
}

void Matrix::set_element(u64 i, u64 j, u64 v) {
    if (i >= s_->nrows || j >= s_->ncols)
        throw GrbError(FGPU_OUT_OF_BOUNDS, "GrB_Matrix_setElement: index out of bounds");
    s_->pend[{i, j}] = s_->type == Type::Bool ? 1 : v;
    s_->has_pending.store(true, std::memory_order_release);
}

void Matrix::remove_element(u64 i, u64 j) {
    if (i >= s_->nrows || j >= s_->ncols)
        throw GrbError(FGPU_OUT_OF_BOUNDS, "GrB_Matrix_removeElement: index out of bounds");
    s_->pend[{i, j}] = std::nullopt;
    s_->has_pending.store(true, std::memory_order_release);
}

void Matrix::probe(const std::vector<u64>& rows, const std::vector<u64>& cols, std::vector<uint8_t>& present,
                   std::vector<u64>* vals) const {
    present.assign(rows.size(), 0);
    if (vals) vals->assign(rows.size(), 0);
    if (rows.empty()) return;
    check(fgpu_mat_probe(s_->ctx->raw(), snapshot(), rows.data(), cols.data(), rows.size(), present.data(),
                         vals ? vals->data() : nullptr),
          "GrB_Matrix_extractElement");
    if (vals && s_->type == Type::Bool)
        for (size_t k = 0; k < rows.size(); ++k) (*vals)[k] = present[k] ? 1 : 0;
}

std::optional<u64> Matrix::get(u64 i, u64 j) const {
    uint8_t p = 0;
    u64 v = 0;
    check(fgpu_mat_probe(s_->ctx->raw(), snapshot(), &i, &j, 1, &p, &v), "GrB_Matrix_extractElement");
    if (!p) return std::nullopt;  // GrB_NO_VALUE
    return s_->type == Type::Bool ? 1 : v;
}

bool Matrix::contains(u64 i, u64 j) const {
    uint8_t p = 0;
    check(fgpu_mat_probe(s_->ctx->raw(), snapshot(), &i, &j, 1, &p, nullptr), "GxB_Matrix_isStoredElement");
    return p != 0;
}

std::vector<Entry> Matrix::iter(u64 min_row, u64 max_row) const {
    u64 *r = nullptr, *c = nullptr, *v = nullptr, n = 0;
    fgpu_ctx* ctx = s_->ctx->raw();
    check(fgpu_mat_extract(ctx, snapshot(), min_row, max_row, &r, &c, s_->type == Type::UInt64 ? &v : nullptr, &n),
          "GxB_rowIterator");
    std::vector<Entry> out(n);
    for (u64 k = 0; k < n; ++k) out[k] = Entry{r[k], c[k], v ? v[k] : 1};
    fgpu_free(ctx, r);
    fgpu_free(ctx, c);
    fgpu_free(ctx, v);
    return out;
}

// ---- MatrixIter ---------------------------------------------------------------------------------------
MatrixIter::MatrixIter(const Matrix& m, u64 min_row, u64 max_row) : m_(m) { seek(min_row, max_row); }

void MatrixIter::seek(u64 min_row, u64 max_row) {
    m_.wait();
    const u64 last = m_.nrows() ? m_.nrows() - 1 : 0;
    cur_ = min_row;
    max_ = max_row < last ? max_row : last;
    depleted_ = m_.nrows() == 0 || min_row > max_;
    window_ = 1024;
    buf_.clear();
    pos_ = 0;
}

void MatrixIter::refill() {
    buf_.clear();
    pos_ = 0;
    while (!depleted_ && buf_.empty()) {
        const u64 hi = (max_ - cur_ < window_) ? max_ : cur_ + window_ - 1;
        buf_ = m_.iter(cur_, hi);
        if (hi >= max_) depleted_ = true;
        else cur_ = hi + 1;
        // empty windows grow, dense ones shrink back: a sparse range is crossed in O(log) extracts, a dense one
        // never pulls more than a few hundred thousand entries at a time
        if (buf_.empty()) window_ = window_ < (1ull << 40) ? window_ * 8 : window_;
        else if (buf_.size() > (1u << 18) && window_ > 64) window_ /= 8;
    }
}

std::optional<Entry> MatrixIter::next() {
    if (pos_ >= buf_.size()) {
        if (depleted_) return std::nullopt;
        refill();
        if (buf_.empty()) return std::nullopt;
    }
    return buf_[pos_++];
}

Matrix Matrix::dup() const {
    auto s = std::make_shared<State>();
    s->ctx = s_->ctx;
    s->type = s_->type;
    s->nrows = s_->nrows;
    s->ncols = s_->ncols;
    s->snap = s_->snap;   // immutable on the device: copy-on-write for free
    s->pend = s_->pend;
    s->has_pending.store(!s->pend.empty(), std::memory_order_release);
    return Matrix(std::move(s));
}

Matrix Matrix::transpose() const {
    snapshot();   // wait()
    std::shared_ptr<Snap> base = s_->snap;
    std::shared_ptr<Snap> tr;
    {
        std::lock_guard<std::mutex> g(base->tmu);
        if (!base->transposed) {
            fgpu_mat* t = nullptr;
            check(fgpu_mat_transpose(s_->ctx->raw(), &t, base->h), "GrB_transpose");
            base->transposed = std::make_shared<Snap>(t);
        }
        tr = base->transposed;
    }
    auto st = std::make_shared<State>();
    st->ctx = s_->ctx;
    st->type = s_->type;
    st->nrows = s_->ncols;
    st->ncols = s_->nrows;
    st->snap = tr;        // copy-on-write like dup(): a mutation of the result installs its own snapshot
    return Matrix(std::move(st));
}

Matrix Matrix::grown(u64 nrows, u64 ncols) const {
    fgpu_mat* g = nullptr;
    check(fgpu_mat_resize(s_->ctx->raw(), &g, snapshot(), nrows, ncols), "GrB_Matrix_resize");
    return adopt(*s_->ctx, s_->type, g);
}

void Matrix::resize(u64 nrows, u64 ncols) {
    fgpu_mat* g = nullptr;
    check(fgpu_mat_resize(s_->ctx->raw(), &g, snapshot(), nrows, ncols), "GrB_Matrix_resize");
    replace(g);
    s_->nrows = nrows;
    s_->ncols = ncols;
}

void Matrix::clear() { replace(new_empty(*s_->ctx, s_->nrows, s_->ncols)); }

void Matrix::lmxm(const Matrix& b) {
    fgpu_mat* c = nullptr;
    check(fgpu_mxm(s_->ctx->raw(), &c, snapshot(), b.snapshot()), "GrB_mxm");
    replace(c);
    s_->ncols = b.ncols();
}

void Matrix::rmxm(const Matrix& b) {
    fgpu_mat* c = nullptr;
    check(fgpu_mxm(s_->ctx->raw(), &c, b.snapshot(), snapshot()), "GrB_mxm");
    replace(c);
    s_->nrows = b.nrows();
}

void Matrix::delta_lmxm(const Matrix& m, const Matrix& dp, const Matrix& dm) {
    fgpu_mat* c = nullptr;
    check(fgpu_delta_lmxm(s_->ctx->raw(), &c, snapshot(), m.snapshot(), dp.snapshot(), dm.snapshot()),
          "GrB_mxm (delta_lmxm)");
    replace(c);
    s_->ncols = m.ncols();
}

static fgpu_mat* merged(Context& ctx, const fgpu_mat* m, const fgpu_mat* dp, const fgpu_mat* dm, bool dm_masks_dp,
                        bool pattern_only, const char* where) {
    fgpu_mat* o = nullptr;
    if (pattern_only)
        check(fgpu_mat_merge_pattern(ctx.raw(), &o, m, dp, dm, dm_masks_dp ? 1 : 0), where);
    else
        check(fgpu_mat_merge(ctx.raw(), &o, m, dp, dm, dm_masks_dp ? 1 : 0), where);
    return o;
}

void Matrix::remove_all(const Matrix& b) {
    replace(merged(*s_->ctx, snapshot(), nullptr, b.snapshot(), false, false, "GrB_transpose<!mask> (remove_all)"));
}

void Matrix::select(const Matrix& mask, const Matrix& a) {
    replace(merged(*s_->ctx, a.snapshot(), nullptr, mask.snapshot(), false, s_->type == Type::Bool,
                   "GrB_transpose<!mask> (select)"));
}

void Matrix::element_wise_add(const Matrix* mask, const Matrix* a, const Matrix* b, Descriptor d) {
    const Matrix& A = a ? *a : *this;
    const Matrix& B = b ? *b : *this;
    if (mask && d != Descriptor::RC)
        throw GrbError(FGPU_INVALID, "element_wise_add: only the (mask, RC) form is used by the Delta layer");
    // bool: ANY (pattern union); u64: SECOND (b's value on a shared coordinate)  matrix.rs:300-311
    replace(merged(*s_->ctx, A.snapshot(), B.snapshot(), mask ? mask->snapshot() : nullptr, mask != nullptr,
                   s_->type == Type::Bool, "GrB_Matrix_eWiseAdd"));
}

void Matrix::element_wise_multiply(const Matrix* a, const Matrix* b) {
    const Matrix& A = a ? *a : *this;
    const Matrix& B = b ? *b : *this;
    fgpu_mat* o = nullptr;
    if (s_->type == Type::Bool) {
        // ANY_PAIR result is iso true: keep the structure only (entries of B found in A carry no values
        // when the second operand is the BOOL one)
        const Matrix& valued = A.type() == Type::UInt64 ? A : B;
        const Matrix& other = &valued == &A ? B : A;
        check(fgpu_mat_intersect(s_->ctx->raw(), &o, valued.snapshot(), other.snapshot()), "GrB_eWiseMult");
        if (other.type() == Type::UInt64) {  // both valued: strip the values
            fgpu_mat* p = merged(*s_->ctx, o, nullptr, nullptr, false, true, "GrB_eWiseMult");
            fgpu_mat_free(o);
            o = p;
        }
    } else {
        check(fgpu_mat_intersect(s_->ctx->raw(), &o, A.snapshot(), B.snapshot()), "GrB_eWiseMult");
    }
    replace(o);
}

void Matrix::set_pattern(const Matrix* mask, const Matrix& a, Descriptor d) {
    if (s_->type != Type::Bool) throw GrbError(FGPU_INVALID, "set_pattern: output must be BOOL");
    if (mask && d != Descriptor::C)
        throw GrbError(FGPU_INVALID, "set_pattern: only the (mask, C) form is used (graph.rs:2520-2549)");
    // self<!mask> U= pattern(a): the addition is masked, what self already holds stays
    fgpu_mat* add = merged(*s_->ctx, a.snapshot(), nullptr, mask ? mask->snapshot() : nullptr, false, true,
                           "GrB_Matrix_apply (set_pattern)");
    fgpu_mat* o = nullptr;
    fgpu_info i = fgpu_mat_merge_pattern(s_->ctx->raw(), &o, snapshot(), add, nullptr, 0);
    fgpu_mat_free(add);
    check(i, "GrB_Matrix_apply (set_pattern)");
    replace(o);
}

u64 Matrix::intersection_nvals(const Matrix& b) const {
    u64 n = 0;
    check(fgpu_mat_intersect_nvals(s_->ctx->raw(), snapshot(), b.snapshot(), &n), "GrB_eWiseMult (nvals)");
    return n;
}

}  // namespace falkor
