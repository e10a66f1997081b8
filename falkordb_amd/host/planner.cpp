// planner.cpp — the one planner rule that decides which queries reach the fused k-hop matrix chain:
// fuse_anonymous_traverse (graph/src/planner/optimizer/fuse_anonymous_traverse.rs:83-284), on a small plan-tree
// model (only what the rule inspects), plus the lowering of a (fused) CondTraverse node to the runtime
// CondTraverseOp this library executes.  No device work here.
#include <algorithm>
#include <deque>
#include <sstream>

#include "host.hpp"

namespace falkor {

static bool is_anon(const std::string& alias) { return alias.rfind("_anon", 0) == 0; }   // :38-40

// ir_references_variable (reduce_expand_into.rs:22-75): Project / Filter / Sort / Aggregate / ... reference the
// aliases their expressions name; a CondTraverse ancestor never counts (`_ => false`)
static bool references(const PlanOp& op, const std::string& alias) {
    if (op.kind != PlanOp::Other) return false;
    return std::find(op.references.begin(), op.references.end(), alias) != op.references.end();
}

// intermediate_unreferenced (:66-79): no ancestor of the outer CondTraverse references the intermediate
static bool intermediate_unreferenced(const Plan& plan, int idx, const std::string& alias) {
    for (int cur = plan.ops[idx].parent; cur >= 0; cur = plan.ops[cur].parent)
        if (references(plan.ops[cur], alias)) return false;
    return true;
}

// can_fuse (:83-188): `parent` is the outer hop (b)-->(c), `child` its only child, the hop (a)-->(b)
bool can_fuse(const Plan& plan, int parent_idx, int child_idx) {
    const PlanOp& p = plan.ops[parent_idx];
    const PlanOp& c = plan.ops[child_idx];
    if (p.kind != PlanOp::CondTraverse || c.kind != PlanOp::CondTraverse) return false;
    if (!p.bind_relationship || !c.bind_relationship) return false;      // the pattern matches bind_relationship: true
    if (p.optional || c.optional) return false;                          // :111-115
    if (p.transposed || c.transposed) return false;                      // :118-122
    if (!is_anon(p.rel.alias) || !is_anon(c.rel.alias)) return false;    // :125-127
    if (p.emit_relationship || c.emit_relationship) return false;        // :131-133
    if (!p.sibling_edges.empty() || !c.sibling_edges.empty()) return false;   // :135-137
    if (p.rel.bidirectional || c.rel.bidirectional) return false;        // :139-141
    if (p.rel.var_len || c.rel.var_len) return false;                    // :142-144
    if (!p.rel.attrs_empty || !c.rel.attrs_empty) return false;          // :146-148
    if (p.rel.from.alias != c.rel.to.alias) return false;                // :150-154 the shared intermediate
    const PlanNodeRef& mid = p.rel.from;
    if (!is_anon(mid.alias)) return false;                               // :157-159
    if (!mid.labels.empty()) return false;                               // :160-162
    if (!mid.attrs_empty) return false;                                  // :163-165
    return intermediate_unreferenced(plan, parent_idx, mid.alias);       // :180-186
}

// fuse_anonymous_traverse (:190-284): repeat { first fusable (parent, only-child) pair in BFS order -> merge }
void fuse_anonymous_traverse(Plan& plan) {
    for (;;) {
        int target = -1;
        std::deque<int> q;
        if (plan.root >= 0) q.push_back(plan.root);
        while (!q.empty() && target < 0) {
            const int idx = q.front();
            q.pop_front();
            const PlanOp& op = plan.ops[idx];
            for (int ch : op.children) q.push_back(ch);
            if (op.kind != PlanOp::CondTraverse || op.children.size() != 1) continue;
            if (can_fuse(plan, idx, op.children[0])) target = idx;
        }
        if (target < 0) return;
        PlanOp& parent = plan.ops[target];
        const int child_idx = parent.children[0];
        PlanOp child = plan.ops[child_idx];
        // chain = child's chain (entry-side hops), the parent's relationship, the parent's chain (:236-241)
        std::vector<PlanRel> merged = child.chain;
        merged.push_back(parent.rel);
        merged.insert(merged.end(), parent.chain.begin(), parent.chain.end());
        parent.rel = child.rel;            // the entry hop; emit_relationship / sibling_edges stay the parent's (:255-263)
        parent.transposed = false;
        parent.optional = false;
        parent.bind_relationship = true;
        parent.chain = merged;
        parent.children = child.children;  // the grandchildren move up, the child is pruned (:266-271)
        for (int g : parent.children) plan.ops[g].parent = target;
        plan.ops[child_idx].kind = PlanOp::Pruned;
        plan.ops[child_idx].children.clear();
        plan.ops[child_idx].parent = -1;
    }
}

// The runtime operator of a CondTraverse plan node: source labels from the entry hop's `from`, one Hop per
// relationship (types, destination labels) — mid-chain nodes carry no labels by construction (:160-162)
CondTraverseOp lower_cond_traverse(const PlanOp& op) {
    CondTraverseOp rt;
    rt.src_labels = op.rel.from.labels;
    auto hop_of = [](const PlanRel& r) { Hop h; h.types = r.types; h.dst_labels = r.to.labels; return h; };
    rt.hops.push_back(hop_of(op.rel));
    for (auto& r : op.chain) rt.hops.push_back(hop_of(r));
    rt.optional = op.optional;
    rt.emit_relationship = op.emit_relationship;
    rt.bind_relationship = op.bind_relationship && op.chain.empty();   // representative edge: single hop only (cond_traverse.rs:663)
    rt.bidirectional = op.rel.bidirectional;
    rt.has_sibling_edges = !op.sibling_edges.empty();
    rt.has_inline_attrs = !op.rel.attrs_empty;
    for (auto& r : op.chain) rt.bidirectional = rt.bidirectional || r.bidirectional;
    return rt;
}

// ---- text form (C surface / tests) ---------------------------------------------------------------------
//   X  <id> <parent> <name> refs=a,b
//   CT <id> <parent> rel=<alias>|<from>|<to>|<types>|<flags> flags=<letters> sib=a,b chain=<rel>;<rel>
//   node = alias[:label:label][*]   (* = inline attributes present)
//   rel flags: b bidirectional, v variable length, a inline attributes;  op flags: e emit_relationship,
//   t transposed, o optional, n bind_relationship = false
static std::vector<std::string> split_s(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    std::istringstream in(s);
    while (std::getline(in, cur, sep)) out.push_back(cur);
    return out;
}

static PlanNodeRef parse_node(std::string s) {
    PlanNodeRef n;
    if (!s.empty() && s.back() == '*') { n.attrs_empty = false; s.pop_back(); }
    auto parts = split_s(s, ':');
    if (!parts.empty()) n.alias = parts[0];
    for (size_t i = 1; i < parts.size(); ++i)
        if (!parts[i].empty()) n.labels.push_back(parts[i]);
    return n;
}
static std::string print_node(const PlanNodeRef& n) {
    std::string s = n.alias;
    for (auto& l : n.labels) s += ":" + l;
    if (!n.attrs_empty) s += "*";
    return s;
}
static PlanRel parse_rel(const std::string& s) {
    auto f = split_s(s, '|');
    f.resize(5);
    PlanRel r;
    r.alias = f[0];
    r.from = parse_node(f[1]);
    r.to = parse_node(f[2]);
    for (auto& t : split_s(f[3], ','))
        if (!t.empty()) r.types.push_back(t);
    r.bidirectional = f[4].find('b') != std::string::npos;
    r.var_len = f[4].find('v') != std::string::npos;
    r.attrs_empty = f[4].find('a') == std::string::npos;
    return r;
}
static std::string print_rel(const PlanRel& r) {
    std::string s = r.alias + "|" + print_node(r.from) + "|" + print_node(r.to) + "|";
    for (size_t i = 0; i < r.types.size(); ++i) s += (i ? "," : "") + r.types[i];
    s += "|";
    if (r.bidirectional) s += "b";
    if (r.var_len) s += "v";
    if (!r.attrs_empty) s += "a";
    return s;
}

Plan parse_plan(const std::string& text) {
    Plan plan;
    std::vector<PlanOp> tmp;
    std::vector<int> ids;
    for (auto& line : split_s(text, '\n')) {
        std::istringstream in(line);
        std::string kind;
        if (!(in >> kind)) continue;
        PlanOp op;
        int id = 0;
        in >> id >> op.parent;
        if (kind == "X") {
            op.kind = PlanOp::Other;
            in >> op.name;
        } else if (kind == "CT") {
            op.kind = PlanOp::CondTraverse;
        } else {
            throw GrbError(FGPU_INVALID, "plan: unknown node kind " + kind);
        }
        std::string tok;
        while (in >> tok) {
            auto eq = tok.find('=');
            if (eq == std::string::npos) continue;
            const std::string k = tok.substr(0, eq), v = tok.substr(eq + 1);
            if (k == "refs") { for (auto& a : split_s(v, ',')) if (!a.empty()) op.references.push_back(a); }
            else if (k == "rel") op.rel = parse_rel(v);
            else if (k == "sib") { for (auto& a : split_s(v, ',')) if (!a.empty()) op.sibling_edges.push_back(a); }
            else if (k == "chain") { for (auto& r : split_s(v, ';')) if (!r.empty()) op.chain.push_back(parse_rel(r)); }
            else if (k == "flags") {
                op.emit_relationship = v.find('e') != std::string::npos;
                op.transposed = v.find('t') != std::string::npos;
                op.optional = v.find('o') != std::string::npos;
                op.bind_relationship = v.find('n') == std::string::npos;
            }
        }
        if ((size_t)id >= tmp.size()) tmp.resize(id + 1);
        tmp[id] = op;
        ids.push_back(id);
    }
    // ids need not be dense (a plan printed after a fusion has lost the pruned children): the gaps are not nodes
    {
        std::vector<bool> listed(tmp.size(), false);
        for (int id : ids) listed[id] = true;
        for (size_t k = 0; k < tmp.size(); ++k)
            if (!listed[k]) tmp[k].kind = PlanOp::Pruned;
    }
    plan.ops = tmp;
    plan.root = -1;
    for (int id : ids) {
        const int par = plan.ops[id].parent;
        if (par < 0) plan.root = id;
        else plan.ops[par].children.push_back(id);    // children in the order the lines list them
    }
    return plan;
}

std::string print_plan(const Plan& plan) {
    std::ostringstream out;
    for (size_t id = 0; id < plan.ops.size(); ++id) {
        const PlanOp& op = plan.ops[id];
        if (op.kind == PlanOp::Pruned) continue;
        if (op.kind == PlanOp::Other) {
            out << "X " << id << " " << op.parent << " " << op.name << " refs=";
            for (size_t i = 0; i < op.references.size(); ++i) out << (i ? "," : "") << op.references[i];
        } else {
            out << "CT " << id << " " << op.parent << " rel=" << print_rel(op.rel) << " flags=";
            if (op.emit_relationship) out << "e";
            if (op.transposed) out << "t";
            if (op.optional) out << "o";
            if (!op.bind_relationship) out << "n";
            out << " sib=";
            for (size_t i = 0; i < op.sibling_edges.size(); ++i) out << (i ? "," : "") << op.sibling_edges[i];
            out << " chain=";
            for (size_t i = 0; i < op.chain.size(); ++i) out << (i ? ";" : "") << print_rel(op.chain[i]);
        }
        out << "\n";
    }
    return out.str();
}

}  // namespace falkor
