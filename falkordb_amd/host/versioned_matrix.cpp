// versioned_matrix.cpp — Delta<T> and VersionedMatrix<bool> (mirrors
// graph/src/graph/graphblas/versioned_matrix.rs; line numbers in host.hpp).
#include <algorithm>

#include "host.hpp"

namespace falkor {

// ---- fold policy (versioned_matrix.rs:140-200) ------------------------------------------------
static u64 sat_mul(u64 a, u64 b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    return p > (unsigned __int128)~0ull ? ~0ull : (u64)p;
}
bool fold_balance(u64 delta_nvals, u64 tx_added, u64 base_nvals, u64 k) {
    return tx_added > 0 && delta_nvals >= MIN_FOLD_DELTA &&
           (sat_mul(delta_nvals, 2) >= base_nvals || sat_mul(delta_nvals, delta_nvals) >= sat_mul(k, tx_added));
}
bool should_fold(u64 d, u64 tx, u64 base) { return fold_balance(d, tx, base, WRITE_FOLD_K); }
bool should_fold_read(u64 d, u64 tx, u64 base) { return fold_balance(d, tx, base, READ_FOLD_K); }
bool delta_dominates_base(u64 d, u64 base) { return d >= MIN_FOLD_DELTA && sat_mul(d, 2) >= base; }

// ---- Delta ---------------------------------------------------------------------------------------
Delta::Delta(Context& ctx, Type t, u64 nrows, u64 ncols) : m_(ctx, t, nrows, ncols) {}
Delta::Delta(Matrix m) : m_(std::move(m)) { count_ = m_.nvals(); }

Delta Delta::new_version(bool fold) const {
    Delta d(m_.dup());   // COW share of the layer; the counter carries over without materializing
    d.count_ = count_;
    d.tx_nvals_ = count_;
    d.fold_ = fold;
    return d;
}

Delta Delta::transposed() const {
    Delta d(m_.transpose());
    d.count_ = count_;
    d.tx_nvals_ = tx_nvals_;
    d.fold_ = fold_;
    return d;
}

void Delta::resync() {
    m_.wait();
    count_ = m_.nvals();
}

bool Delta::fold_decision(bool (*policy)(u64, u64, u64), u64 base) const {
    return fold_ || policy(count_, count_ > tx_nvals_ ? count_ - tx_nvals_ : 0, base);
}

bool Delta::take_fold() {
    bool f = fold_;
    fold_ = false;
    return f && m_.nvals() > 0;
}

void Delta::clear(u64 nrows, u64 ncols) {
    m_ = Matrix(m_.ctx(), m_.type(), nrows, ncols);
    count_ = 0;
    tx_nvals_ = 0;
    fold_ = false;
}

void Delta::insert(u64 i, u64 j, u64 v) {
    m_.set_element(i, j, v);
    count_ += 1;   // moves whether or not the key existed: the counter is approximate by design
}

void Delta::erase(u64 i, u64 j) {
    m_.remove_element(i, j);
    count_ = count_ ? count_ - 1 : 0;
}

void Delta::tombstone_masked(const Matrix& mask, const Matrix& base) {
    // self<mask> = mask & base without REPLACE: what the layer held outside the mask stays
    Matrix hit(m_.ctx(), Type::Bool, m_.nrows(), m_.ncols());
    hit.element_wise_multiply(&mask, &base);
    m_.element_wise_add(nullptr, nullptr, &hit, Descriptor::None);
    resync();
}

void Delta::remove_all(const Matrix& mask) {
    m_.remove_all(mask);
    resync();
}

// ---- VersionedMatrix<bool> ---------------------------------------------------------------------------
VersionedMatrix::VersionedMatrix(Context& ctx, u64 nrows, u64 ncols)
    : m_(ctx, Type::Bool, nrows, ncols), dp_(ctx, Type::Bool, nrows, ncols), dm_(ctx, Type::Bool, nrows, ncols) {}

VersionedMatrix VersionedMatrix::from_matrix(Matrix m) {
    Context& c = m.ctx();
    u64 nr = m.nrows(), nc = m.ncols();
    return VersionedMatrix(std::move(m), Delta(c, Type::Bool, nr, nc), Delta(c, Type::Bool, nr, nc));
}

void VersionedMatrix::wait() const {
    if (dp_.is_synced() && dm_.is_synced()) return;
    dp_.resync();
    dm_.resync();
    u64 base = m_.nvals();
    dp_.latch(dp_.fold_decision(should_fold_read, base));
    dm_.latch(dm_.fold_decision(should_fold_read, base));
    // needs_flush is NOT set: the latched decision is executed by the next version (dup -> flush)
}

void VersionedMatrix::wait_all() const {
    m_.wait();
    dp_.wait();
    dm_.wait();
}

u64 VersionedMatrix::nvals() const {
    wait();
    return m_.nvals() + dp_.nvals() - dm_.nvals();
}

Matrix VersionedMatrix::extract() const {
    wait();
    // clean layers: the effective state IS the base; share its immutable snapshot (copy-on-write) instead of a device
    // copy — the result is a fresh handle exactly as versioned_matrix.rs:609-620 returns, and a caller that mutates
    // it (set_pattern in build_relationship_matrix_unrestricted) gets its own snapshot on the first write
    if (m_.type() == Type::Bool && dp_.layer().nvals() == 0 && dm_.layer().nvals() == 0) return m_.dup();
    fgpu_mat* o = nullptr;
    check(fgpu_mat_merge_pattern(m_.ctx().raw(), &o, m_.snapshot(), dp_.layer().snapshot(), dm_.layer().snapshot(), 0),
          "VersionedMatrix::extract");
    return Matrix::adopt(m_.ctx(), Type::Bool, o);
}

std::optional<bool> VersionedMatrix::get(u64 i, u64 j) const {
    wait();
    if (m_.contains(i, j)) {
        if (dm_.contains(i, j)) return std::nullopt;
        return true;
    }
    if (dp_.contains(i, j)) return true;
    return std::nullopt;
}

// Three sorted row iterators merged on the host, exactly the reference's Iter (versioned_matrix.rs:1116-1253):
// an m entry equal to the current dm key is dropped, dp interleaves, dp wins a tie.
std::vector<Entry> merge_layers(const std::vector<Entry>& m, const std::vector<Entry>& dp, const std::vector<Entry>& dm) {
    auto key_lt = [](const Entry& a, const Entry& b) { return a.row != b.row ? a.row < b.row : a.col < b.col; };
    auto key_eq = [](const Entry& a, const Entry& b) { return a.row == b.row && a.col == b.col; };
    std::vector<Entry> out;
    out.reserve(m.size() + dp.size());
    size_t im = 0, ip = 0, id = 0;
    while (im < m.size() || ip < dp.size()) {
        if (im < m.size()) {
            while (id < dm.size() && key_lt(dm[id], m[im])) ++id;
            if (id < dm.size() && key_eq(dm[id], m[im])) {  // tombstoned
                ++im;
                continue;
            }
        }
        if (im >= m.size()) {
            out.push_back(dp[ip++]);
        } else if (ip >= dp.size()) {
            out.push_back(m[im++]);
        } else if (key_lt(dp[ip], m[im])) {
            out.push_back(dp[ip++]);
        } else if (key_eq(dp[ip], m[im])) {  // shadowed pair: dp's value is the live one
            out.push_back(dp[ip++]);
            ++im;
        } else {
            out.push_back(m[im++]);
        }
    }
    return out;
}

std::vector<Entry> VersionedMatrix::iter(u64 min_row, u64 max_row) const {
    wait();
    return merge_layers(m_.iter(min_row, max_row), dp_.layer().iter(min_row, max_row),
                        dm_.layer().iter(min_row, max_row));
}

void VersionedMatrix::flush() {
    if (!needs_flush_) return;
    wait_all();
    bool fold_dp = dp_.take_fold();
    bool fold_dm = dm_.take_fold();
    if (fold_dp || fold_dm) {
        u64 nr = m_.nrows(), nc = m_.ncols();
        Matrix new_m(m_.ctx(), Type::Bool, nr, nc);
        if (fold_dp && fold_dm)
            new_m.element_wise_add(&dm_.layer(), &m_, &dp_.layer(), Descriptor::RC);  // new_m<!dm,replace> = m + dp
        else if (fold_dp)
            new_m.element_wise_add(nullptr, &m_, &dp_.layer(), Descriptor::None);
        else
            new_m.select(dm_.layer(), m_);                                            // new_m<!dm,replace> = m
        new_m.wait();
        m_ = new_m;
        if (fold_dp) dp_.clear(nr, nc);
        if (fold_dm) dm_.clear(nr, nc);
    }
    needs_flush_ = false;
}

void VersionedMatrix::set(u64 i, u64 j, bool) {
    flush();
    if (m_.contains(i, j))
        dm_.erase(i, j);     // un-delete a committed entry
    else
        dp_.insert(i, j);
}

void VersionedMatrix::remove(u64 i, u64 j) {
    flush();
    if (m_.contains(i, j))
        dm_.insert(i, j);
    else
        dp_.erase(i, j);
}

void VersionedMatrix::remove_mask(const Matrix& mask) {
    flush();
    m_.wait();
    dm_.tombstone_masked(mask, m_);   // dm U= mask & m
    dp_.remove_all(mask);             // dp \= mask
}

void VersionedMatrix::set_all(const std::vector<std::pair<u64, u64>>& entries, bool is_new) {
    flush();
    dm_.wait();
    // one device probe of the committed base for the whole batch instead of a get per entry: `m` is never
    // pending and no entry of this call changes it, so the answers are those the per-entry calls would see
    const bool dm_empty = dm_.nvals() == 0;
    std::vector<uint8_t> in_m;
    if (!(dm_empty && is_new)) {
        std::vector<u64> r(entries.size()), c(entries.size());
        for (size_t k = 0; k < entries.size(); ++k) { r[k] = entries[k].first; c[k] = entries[k].second; }
        m_.probe(r, c, in_m, nullptr);
    }
    if (dm_empty) {
        for (size_t k = 0; k < entries.size(); ++k) {
            if (!is_new && in_m[k]) continue;   // keeps dp & m = {}
            dp_.insert(entries[k].first, entries[k].second);
        }
    } else {
        for (size_t k = 0; k < entries.size(); ++k) {   // the body of set(), probe hoisted
            if (in_m[k]) dm_.erase(entries[k].first, entries[k].second);
            else dp_.insert(entries[k].first, entries[k].second);
        }
    }
}

VersionedMatrix VersionedMatrix::dup() const {
    u64 base = m_.nvals();
    bool fold_dp = dp_.fold_decision(should_fold, base);
    bool fold_dm = dm_.fold_decision(should_fold, base);
    VersionedMatrix v(m_.dup(), dp_.new_version(fold_dp), dm_.new_version(fold_dm));
    v.needs_flush_ = fold_dp || fold_dm;
    return v;
}

void VersionedMatrix::fold_oversized() {
    u64 base = m_.nvals();
    bool odp = delta_dominates_base(dp_.count(), base);
    bool odm = delta_dominates_base(dm_.count(), base);
    if (odp || odm) {
        dp_.latch(odp);
        dm_.latch(odm);
        needs_flush_ = true;
        flush();
    }
}

void VersionedMatrix::fold_latched() {
    wait();
    if (dp_.folding() || dm_.folding()) {
        needs_flush_ = true;
        flush();
    }
}

void VersionedMatrix::resize(u64 nrows, u64 ncols) {
    wait_all();
    m_.resize(nrows, ncols);
    dp_.layer().resize(nrows, ncols);
    dm_.layer().resize(nrows, ncols);
}

VersionedMatrix VersionedMatrix::transpose() const {
    wait_all();
    VersionedMatrix v(m_.transpose(), dp_.transposed(), dm_.transposed());
    v.needs_flush_ = needs_flush_;
    return v;
}

}  // namespace falkor
