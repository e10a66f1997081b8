// tensor.cpp — Tensor: per-relationship-type edge storage with inline edge ids
// (mirrors graph/src/graph/graphblas/tensor.rs; per-pair state diagram at tensor.rs:72-108).
#include <algorithm>

#include <unordered_map>

#include "host.hpp"

namespace falkor {

u64 compound_key(u64 src, u64 dst) {
    if ((src >> 32) || (dst >> 32))
        throw GrbError(FGPU_INVALID, "Tensor compound key overflow: src=" + std::to_string(src) +
                                         ", dst=" + std::to_string(dst) + " (each must fit in u32)");
    return (src << 32) | dst;
}

Tensor::Tensor(Context& ctx, u64 nrows, u64 ncols)
    : m_(ctx, Type::UInt64, nrows, ncols),
      dp_(ctx, Type::UInt64, nrows, ncols),
      dm_(ctx, Type::Bool, nrows, ncols),
      mt_(ctx, ncols, nrows) {}

void Tensor::wait_fwd() const {
    if (dp_.is_synced() && dm_.is_synced()) return;
    dp_.resync();
    dm_.resync();
    u64 base = m_.nvals();
    dp_.latch(dp_.fold_decision(should_fold_read, base));
    dm_.latch(dm_.fold_decision(should_fold_read, base));
}

std::optional<u64> Tensor::eff_get(u64 src, u64 dst) const {
    wait_fwd();
    if (auto v = dp_.get(src, dst)) return v;
    if (dm_.nvals() != 0 && dm_.contains(src, dst)) return std::nullopt;
    return m_.get(src, dst);
}

std::vector<u64> Tensor::get(u64 src, u64 dst) const {
    auto v = eff_get(src, dst);
    if (!v) return {};
    if (*v == MULTI_EDGE) {
        auto it = me_.find(compound_key(src, dst));
        return it == me_.end() ? std::vector<u64>{} : it->second;   // ascending
    }
    return {*v};
}

// effective inline value of every pair: dp wins, then m unless dm-masked (three probes for the batch)
static void eff_get_batch(const Matrix& m, const Matrix& dp, const Matrix& dm, const std::vector<u64>& srcs,
                          const std::vector<u64>& dsts, std::vector<uint8_t>& present, std::vector<u64>& vals,
                          std::vector<uint8_t>* masked_out = nullptr, std::vector<uint8_t>* in_dp_out = nullptr,
                          std::vector<uint8_t>* in_m_out = nullptr, std::vector<u64>* m_vals_out = nullptr) {
    std::vector<uint8_t> pd, pm, pk;
    std::vector<u64> vd, vm;
    dp.probe(srcs, dsts, pd, &vd);
    m.probe(srcs, dsts, pm, &vm);
    if (dm.nvals() != 0)
        dm.probe(srcs, dsts, pk, nullptr);
    else
        pk.assign(srcs.size(), 0);
    present.assign(srcs.size(), 0);
    vals.assign(srcs.size(), 0);
    for (size_t k = 0; k < srcs.size(); ++k) {
        if (pd[k]) { present[k] = 1; vals[k] = vd[k]; }
        else if (pk[k]) { present[k] = 0; }
        else if (pm[k]) { present[k] = 1; vals[k] = vm[k]; }
    }
    if (masked_out) *masked_out = pk;
    if (in_dp_out) *in_dp_out = pd;
    if (in_m_out) *in_m_out = pm;
    if (m_vals_out) *m_vals_out = vm;
}

void Tensor::get_batch(const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                       std::vector<std::vector<u64>>& ids) const {
    wait_fwd();
    std::vector<uint8_t> present;
    std::vector<u64> vals;
    eff_get_batch(m_, dp_.layer(), dm_.layer(), srcs, dsts, present, vals);
    ids.assign(srcs.size(), {});
    for (size_t k = 0; k < srcs.size(); ++k) {
        if (!present[k]) continue;
        if (vals[k] == MULTI_EDGE) {
            auto it = me_.find(compound_key(srcs[k], dsts[k]));
            if (it != me_.end()) ids[k] = it->second;
        } else {
            ids[k] = {vals[k]};
        }
    }
}

static void me_insert(std::map<u64, std::vector<u64>>& me, u64 key, u64 id) {
    auto& v = me[key];
    auto it = std::lower_bound(v.begin(), v.end(), id);
    if (it == v.end() || *it != id) v.insert(it, id);
}

void Tensor::set_all_from_slices(const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                                 const std::vector<u64>& ids) {
    if (srcs.size() != dsts.size() || srcs.size() != ids.size())
        throw GrbError(FGPU_INVALID, "set_all_from_slices: slices differ in length");
    if (srcs.empty()) return;
    flush();
    dp_.wait();
    dm_.wait();
    // Read phase (tensor.rs:353-421): each edge's placement is decided against the state the layers hold
    // BEFORE this batch; the three probes below read that state for every pair at once.
    std::vector<uint8_t> present, masked, in_dp, in_m;
    std::vector<u64> cur, m_vals;
    eff_get_batch(m_, dp_.layer(), dm_.layer(), srcs, dsts, present, cur, &masked, &in_dp, &in_m, &m_vals);
    constexpr size_t PROMOTED = ~(size_t)0;
    std::unordered_map<u64, size_t> batch;         // pair (compound key) -> index of its pending inline slot, or PROMOTED
    batch.reserve(srcs.size());
    std::vector<u64> w_src, w_dst, w_id;
    std::vector<std::optional<u64>> w_masked;      // committed m value of pairs needing delta reconciliation
    for (size_t k = 0; k < srcs.size(); ++k) {
        const u64 s = srcs[k], d = dsts[k], id = ids[k];
        const u64 key = compound_key(s, d);
        auto it = batch.find(key);
        if (it != batch.end()) {
            if (it->second != PROMOTED) {          // second edge of a pair new in this batch: promote in place
                me_insert(me_, key, w_id[it->second]);
                w_id[it->second] = MULTI_EDGE;
                it->second = PROMOTED;
            }
            me_insert(me_, key, id);
            continue;
        }
        auto committed = [&]() -> std::optional<u64> { return in_m[k] ? std::optional<u64>(m_vals[k]) : std::nullopt; };
        if (present[k] && cur[k] == MULTI_EDGE) {  // already multi-edge: just add the id
            me_insert(me_, key, id);
            batch[key] = PROMOTED;
        } else if (present[k]) {                   // present single edge: promote
            me_insert(me_, key, cur[k]);
            me_insert(me_, key, id);
            batch[key] = PROMOTED;
            w_src.push_back(s); w_dst.push_back(d); w_id.push_back(MULTI_EDGE);
            w_masked.push_back(in_dp[k] ? committed() : std::nullopt);
        } else {                                   // first edge of the pair: inline
            batch[key] = w_id.size();
            w_src.push_back(s); w_dst.push_back(d); w_id.push_back(id);
            w_masked.push_back(masked[k] ? committed() : std::nullopt);
        }
    }
    // Write phase (tensor.rs:429-454).  mt.set(d, s) of every written pair goes out as one set_all: the same
    // per-entry body with its probe of mt's committed base hoisted into one batch.
    {
        std::vector<std::pair<u64, u64>> back(w_src.size());
        for (size_t i = 0; i < w_src.size(); ++i) back[i] = {w_dst[i], w_src[i]};
        mt_.set_all(back, false);
    }
    for (size_t i = 0; i < w_src.size(); ++i) {
        const u64 s = w_src[i], d = w_dst[i], id = w_id[i];
        if (w_masked[i]) {
            dm_.erase(s, d);
            if (*w_masked[i] == id) {              // cancel to clean: committed value restored
                dp_.erase(s, d);
                continue;
            }
        }
        dp_.insert(s, d, id);
    }
}

std::vector<std::pair<u64, u64>> Tensor::remove_all(const std::vector<std::array<u64, 3>>& rels) {
    std::vector<std::pair<u64, u64>> emptied;
    if (rels.empty()) return emptied;
    flush();
    Context& ctx = m_.ctx();
    if (!has_multi_edge()) {
        // Fast path (tensor.rs:473-505): every edge is the inline value of its pair
        wait_fwd();
        std::vector<u64> mr, mc;
        for (auto& r : rels) { mr.push_back(r[1]); mc.push_back(r[2]); }
        Matrix m_mask(ctx, Type::Bool, m_.nrows(), m_.ncols());
        m_mask.build(mr, mc);
        Matrix mt_mask(ctx, Type::Bool, m_.ncols(), m_.nrows());
        mt_mask.build(mc, mr);
        dm_.tombstone_masked(m_mask, m_);
        dp_.remove_all(m_mask);
        mt_.remove_mask(mt_mask);
        for (auto& r : rels) emptied.push_back({r[1], r[2]});
        return emptied;
    }
    // Slow path (tensor.rs:507-655): read phase replays each touched pair's transitions, write phase applies
    wait_fwd();
    struct Plan {
        enum Kind { Multi, Single, Emptied, Absent } kind = Absent;
        std::vector<u64> ids;   // Multi: ids still in the me row
        u64 id = 0;             // Single
        bool demoted = false;
    };
    std::vector<std::pair<u64, u64>> order;
    std::map<std::pair<u64, u64>, Plan> plans;
    {   // effective state of every distinct pair, one batch of probes
        std::vector<u64> ps, pd;
        for (auto& r : rels)
            if (plans.emplace(std::make_pair(r[1], r[2]), Plan{}).second) { ps.push_back(r[1]); pd.push_back(r[2]); }
        std::vector<uint8_t> present;
        std::vector<u64> vals;
        eff_get_batch(m_, dp_.layer(), dm_.layer(), ps, pd, present, vals);
        for (size_t k = 0; k < ps.size(); ++k) {
            Plan& p = plans[{ps[k], pd[k]}];
            if (!present[k]) p.kind = Plan::Absent;
            else if (vals[k] == MULTI_EDGE) {
                p.kind = Plan::Multi;
                auto it = me_.find(compound_key(ps[k], pd[k]));
                if (it != me_.end()) p.ids = it->second;
            } else { p.kind = Plan::Single; p.id = vals[k]; }
            order.push_back({ps[k], pd[k]});
        }
    }
    std::vector<std::pair<u64, u64>> me_del;   // (key, id)
    for (auto& r : rels) {
        const u64 id = r[0], src = r[1], dst = r[2];
        const u64 key = compound_key(src, dst);
        Plan& p = plans[{src, dst}];
        if (p.kind == Plan::Multi) {
            auto it = std::lower_bound(p.ids.begin(), p.ids.end(), id);
            if (it == p.ids.end() || *it != id) continue;   // unknown id
            p.ids.erase(it);
            me_del.push_back({key, id});
            if (p.ids.size() == 1) {                        // demote: survivor returns inline
                u64 last = p.ids[0];
                me_del.push_back({key, last});
                p.kind = Plan::Single;
                p.id = last;
                p.demoted = true;
                p.ids.clear();
            }
        } else if (p.kind == Plan::Single && p.id == id) {
            p.kind = Plan::Emptied;
            emptied.push_back({src, dst});
        }
    }
    // Write phase: me first, then the forward / backward layers
    for (auto& kd : me_del) {
        auto it = me_.find(kd.first);
        if (it == me_.end()) continue;
        auto pos = std::lower_bound(it->second.begin(), it->second.end(), kd.second);
        if (pos != it->second.end() && *pos == kd.second) it->second.erase(pos);
        if (it->second.empty()) me_.erase(it);
    }
    std::vector<std::array<u64, 3>> dp_set;
    {   // committed values of the touched pairs (m is never pending): one probe
        std::vector<u64> ps, pd;
        for (auto& pr : order) { ps.push_back(pr.first); pd.push_back(pr.second); }
        std::vector<uint8_t> in_m;
        std::vector<u64> mv;
        m_.probe(ps, pd, in_m, &mv);
        for (size_t k = 0; k < order.size(); ++k) {
            const u64 src = order[k].first, dst = order[k].second;
            const Plan& p = plans[order[k]];
            if (p.kind == Plan::Emptied) {
                dp_.erase(src, dst);
                if (in_m[k]) dm_.insert(src, dst);
                mt_.remove(dst, src);
            } else if (p.kind == Plan::Single && p.demoted) {
                if (in_m[k] && mv[k] == p.id) dp_.erase(src, dst);   // cancel to clean
                else dp_set.push_back({src, dst, p.id});
            }
        }
    }
    for (auto& t : dp_set) dp_.insert(t[0], t[1], t[2]);
    return emptied;
}

void Tensor::resize(u64 nrows, u64 ncols) {
    if (nrows < m_.nrows() || ncols < m_.ncols()) flush();
    m_.wait();
    dp_.wait();
    dm_.wait();
    m_ = m_.grown(nrows, ncols);   // also the shrink path: entries past the new dims are dropped
    dp_.replace(dp_.nvals() > 0 ? dp_.layer().grown(nrows, ncols) : Matrix(m_.ctx(), Type::UInt64, nrows, ncols));
    dm_.replace(dm_.nvals() > 0 ? dm_.layer().grown(nrows, ncols) : Matrix(m_.ctx(), Type::Bool, nrows, ncols));
    mt_.resize(ncols, nrows);
}

void Tensor::flush() {
    if (needs_flush_) {
        m_.wait();
        dp_.wait();
        dm_.wait();
        bool fold_dp = dp_.take_fold();
        bool fold_dm = dm_.take_fold();
        if (fold_dp || fold_dm) {
            u64 nr = m_.nrows(), nc = m_.ncols();
            Matrix new_m(m_.ctx(), Type::UInt64, nr, nc);
            if (fold_dp && fold_dm)
                new_m.element_wise_add(&dm_.layer(), &m_, &dp_.layer(), Descriptor::RC);  // dp wins on shadowed pairs
            else if (fold_dp)
                new_m.element_wise_add(nullptr, &m_, &dp_.layer(), Descriptor::None);
            else
                new_m.select(dm_.layer(), m_);
            new_m.wait();
            m_ = new_m;
            if (fold_dp) dp_.clear(nr, nc);
            if (fold_dm) dm_.clear(nr, nc);
        }
        needs_flush_ = false;
    }
    mt_.flush();
}

void Tensor::fold_oversized() {
    u64 base = m_.nvals();
    bool odp = delta_dominates_base(dp_.count(), base);
    bool odm = delta_dominates_base(dm_.count(), base);
    if (odp || odm) {
        dp_.latch(odp);
        dm_.latch(odm);
        needs_flush_ = true;
        flush();
    }
    mt_.fold_oversized();
}

Matrix Tensor::extract() const {
    wait_fwd();
    fgpu_mat* o = nullptr;
    check(fgpu_mat_merge_pattern(m_.ctx().raw(), &o, m_.snapshot(), dp_.layer().snapshot(), dm_.layer().snapshot(), 0),
          "Tensor::extract");
    return Matrix::adopt(m_.ctx(), Type::Bool, o);
}

Tensor Tensor::dup() const {
    u64 base = m_.nvals();
    bool fold_dp = dp_.fold_decision(should_fold, base);
    bool fold_dm = dm_.fold_decision(should_fold, base);
    Tensor t(*this);
    t.m_ = m_.dup();
    t.dp_ = dp_.new_version(fold_dp);
    t.dm_ = dm_.new_version(fold_dm);
    t.mt_ = mt_.dup();
    t.needs_flush_ = fold_dp || fold_dm;
    return t;
}

std::vector<Entry> Tensor::structural_iter(u64 min_row, u64 max_row) const {
    wait_fwd();
    auto out = merge_layers(m_.iter(min_row, max_row), dp_.layer().iter(min_row, max_row),
                            dm_.layer().iter(min_row, max_row));
    for (auto& e : out) e.val = 1;
    return out;
}

std::vector<Entry> Tensor::iter_edges() const {
    wait_fwd();
    auto fwd = merge_layers(m_.iter(0, ~0ull), dp_.layer().iter(0, ~0ull), dm_.layer().iter(0, ~0ull));
    std::vector<Entry> out;
    for (auto& e : fwd)
        if (e.val != MULTI_EDGE) out.push_back(e);
    for (auto& kv : me_)
        for (u64 id : kv.second) out.push_back(Entry{kv.first >> 32, kv.first & 0xFFFFFFFFull, id});
    return out;
}

u64 Tensor::edge_count() const {
    wait_fwd();
    u64 shadow = dp_.nvals() == 0 ? 0 : dp_.layer().intersection_nvals(m_);
    u64 me_nvals = 0;
    for (auto& kv : me_) me_nvals += kv.second.size();
    return m_.nvals() + dp_.nvals() - dm_.nvals() - shadow - multi_pairs() + me_nvals;
}

}  // namespace falkor
