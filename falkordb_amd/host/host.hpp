// host.hpp — C++ host layer above the C ABI (include/fgpu.h): the part of FalkorDB that sits between
// the execution-plan operators and the GraphBLAS boundary, restated over libfgpu.
//
// The reference writes this layer in Rust (graph/src/graph/graphblas/{matrix,versioned_matrix,tensor}.rs,
// graph/src/graph/graph.rs, graph/src/runtime/ops/{cond_traverse,expand_into}.rs,
// graph/src/runtime/functions/algo_procedures.rs).  rustc is not part of the build image, so the
// same types are provided in C++17 with the reference's names, argument meaning and error behaviour;
// every method cites the lines it mirrors (file:line relative to /root/reference/graph/src).
//
// Nothing here computes on the CPU what the reference computes in GraphBLAS: products, merges,
// transposes, probes, builds and BFS all go through fgpu_* calls.  What stays on the host is what the
// reference keeps on the host too: handle ownership, pending-tuple logs, fold bookkeeping, the Tensor
// per-pair state machine, operator control flow.  There is no CPU fallback: constructing a Context
// without a HIP device throws.
#pragma once
#include <stdint.h>

#include <array>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/fgpu.h"

namespace falkor {

using u64 = uint64_t;

// GrB_Info other than GrB_SUCCESS / GrB_NO_VALUE.  The Rust wrappers debug_assert on these
// (matrix.rs passim); here they surface as exceptions which the C surface (capi.cpp) turns back into codes.
struct GrbError : std::runtime_error {
    fgpu_info info;
    GrbError(fgpu_info i, const std::string& what) : std::runtime_error(what), info(i) {}
};
void check(fgpu_info i, const char* where);

// matrix::init / matrix::shutdown (graphblas/matrix.rs:116-221): one engine context per process+device.
class Context {
   public:
    explicit Context(int device);
    ~Context();
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    fgpu_ctx* raw() const { return ctx_; }

   private:
    fgpu_ctx* ctx_ = nullptr;
};

enum class Type { Bool, UInt64 };

// GrB_DESC_* subset the traversal path uses (matrix.rs:313-351)
enum class Descriptor { None, C, RC, RSC };

struct Entry {
    u64 row, col, val;  // val == 1 for BOOL
    bool operator==(const Entry& o) const { return row == o.row && col == o.col && val == o.val; }
};

// Byte stream primitives of the v19 encoders (the reference's Reader / Writer traits; serialize.cpp)
struct ByteWriter {
    std::vector<uint8_t> buf;
    void write_unsigned(u64 v);
    void write_signed(int64_t v);
    void write_buffer(const void* p, size_t n);
};
struct ByteReader {
    const uint8_t* p = nullptr;
    size_t n = 0, pos = 0;
    ByteReader(const uint8_t* data, size_t len) : p(data), n(len) {}
    u64 read_unsigned();
    int64_t read_signed();
    std::vector<uint8_t> read_buffer();
};
// The content of a GxB_Container as the graph stores matrices: sparse or hypersparse, row-major, iso BOOL or UINT64
struct ContainerData {
    u64 nrows = 0, ncols = 0, nvals = 0;
    int format = 2, orientation = 0;
    bool iso = false, jumbled = false, hyper = false, valued = false;
    std::vector<u64> p, h, i, x;   // row pointers (nvec + 1), stored rows (hypersparse), column ids, values
};
ContainerData parse_container(ByteReader& r);                 // Decode<19> for Matrix<T>, CPU part (matrix.rs:428-504)
void write_container(ByteWriter& w, const ContainerData& c);  // Encode<19> for Matrix<T> (matrix.rs:506-546)

// Matrix<T> (graphblas/matrix.rs:360-368): an Arc-shared handle; copies share the underlying matrix,
// `dup()` deep-copies.  The device holds the materialized state as an immutable fgpu_mat snapshot; writes
// queue in a host pending log (GraphBLAS pending tuples / zombies) until wait() folds them in with one
// device merge.
class Matrix {
   public:
    Matrix(Context& ctx, Type t, u64 nrows, u64 ncols);   // Matrix::new   matrix.rs:1151-1156, 1214-1235
    static Matrix adopt(Context& ctx, Type t, fgpu_mat* snapshot);  // wrap an engine-made snapshot

    Type type() const;
    Context& ctx() const;
    u64 nrows() const;                                   // matrix.rs:706-712
    u64 ncols() const;                                   // matrix.rs:714-720
    u64 nvals() const;                                   // matrix.rs:722-729 (waits: GrB_Matrix_nvals does)
    bool pending() const;                                // matrix.rs:764-779
    bool is_synced() const;                              // matrix.rs:802-804
    void wait() const;                                   // matrix.rs:781-796

    void build(const std::vector<u64>& rows, const std::vector<u64>& cols,
               const std::vector<u64>* vals = nullptr);  // matrix.rs:1281-1303 (bool), 1186-1210 (u64)
    void set_element(u64 i, u64 j, u64 v = 1);           // matrix.rs:1174-1184, 1264-1279
    void remove_element(u64 i, u64 j);                   // matrix.rs:664-676
    std::optional<u64> get(u64 i, u64 j) const;          // matrix.rs:1158-1172, 1248-1262 (NO_VALUE -> nullopt)
    bool contains(u64 i, u64 j) const;                   // matrix.rs:731-737
    // batched form of get/contains: one device probe for the whole list (ExpandInto, label checks)
    void probe(const std::vector<u64>& rows, const std::vector<u64>& cols, std::vector<uint8_t>& present,
               std::vector<u64>* vals) const;
    std::vector<Entry> iter(u64 min_row, u64 max_row) const;  // matrix::Iter  matrix.rs:1471-1605

    Matrix dup() const;                                  // Dup  matrix.rs:370-385
    Matrix transpose() const;                            // matrix.rs:633-662
    void resize(u64 nrows, u64 ncols);                   // matrix.rs:576-598
    Matrix grown(u64 nrows, u64 ncols) const;            // matrix.rs:664-704
    void clear();                                        // matrix.rs:815-822
    void lmxm(const Matrix& b);                          // self = self * b     matrix.rs:930-947
    void rmxm(const Matrix& b);                          // self = b * self     matrix.rs:951-968
    void delta_lmxm(const Matrix& m, const Matrix& dp, const Matrix& dm);  // matrix.rs:1317-1402
    void remove_all(const Matrix& b);                    // self = self \ pattern(b)   matrix.rs:824-833
    void select(const Matrix& mask, const Matrix& a);    // self = a \ pattern(mask)   matrix.rs:835-845
    // self<mask> = a (+) b; a / b default to self.  Supported forms are the ones the Delta layer issues:
    // no mask, or (mask, RC).                              matrix.rs:852-874
    void element_wise_add(const Matrix* mask, const Matrix* a, const Matrix* b, Descriptor d);
    // self = pattern(a) & pattern(b), values of b           matrix.rs:876-896
    void element_wise_multiply(const Matrix* a, const Matrix* b);
    // self<mask> U= pattern(a): no mask, or (mask, C)        matrix.rs:906-924
    void set_pattern(const Matrix* mask, const Matrix& a, Descriptor d);
    u64 intersection_nvals(const Matrix& b) const;       // matrix.rs:743-761

    const fgpu_mat* snapshot() const;  // wait()ed device state, valid until the next mutation
    static Matrix decode(Context& ctx, ByteReader& r);   // Decode<19>  matrix.rs:428-504
    void encode(ByteWriter& w) const;                    // Encode<19>  matrix.rs:506-546

   private:
    struct State;
    std::shared_ptr<State> s_;
    explicit Matrix(std::shared_ptr<State> s) : s_(std::move(s)) {}
    void replace(fgpu_mat* fresh) const;
};

// matrix::Iter<E> (matrix.rs:1471-1605): a reusable streaming cursor over the entries of rows [min, max] in
// ascending (row, col) order.  `seek` re-aims the same cursor (what CondTraverseOp::expand_row does once per input
// row, cond_traverse.rs:758-974) and keeps the matrix alive like the reference's Arc clone (:1472, 1485-1497).
// Entries come from the device in windows of rows (fgpu_mat_extract), not one FFI call per entry.
class MatrixIter {
   public:
    MatrixIter(const Matrix& m, u64 min_row, u64 max_row);   // Iter::new   matrix.rs:1500-1540
    void seek(u64 min_row, u64 max_row);                      // Iter::seek  matrix.rs:1542-1570
    std::optional<Entry> next();                              // Iterator::next  matrix.rs:1572-1605
   private:
    Matrix m_;
    u64 cur_ = 0, max_ = 0, window_ = 1024;
    bool depleted_ = true;
    std::vector<Entry> buf_;
    size_t pos_ = 0;
    void refill();
};

// fold policy (versioned_matrix.rs:140-200) — pure integer arithmetic, pinned by versioned_matrix.rs:1278-1330
constexpr u64 WRITE_FOLD_K = 20500000;
constexpr u64 READ_FOLD_K = 82000;
constexpr u64 MIN_FOLD_DELTA = 256;
bool fold_balance(u64 delta_nvals, u64 tx_added, u64 base_nvals, u64 k);   // :175-188
bool should_fold(u64 delta_nvals, u64 tx_added, u64 base_nvals);           // :152-158
bool should_fold_read(u64 delta_nvals, u64 tx_added, u64 base_nvals);      // :164-170
bool delta_dominates_base(u64 delta_nvals, u64 base_nvals);                // :195-200

// Delta<T> (versioned_matrix.rs:214-478): one delta layer plus its fold bookkeeping.
class Delta {
   public:
    Delta(Context& ctx, Type t, u64 nrows, u64 ncols);
    explicit Delta(Matrix m);
    Matrix& layer() { return m_; }
    const Matrix& layer() const { return m_; }
    Delta new_version(bool fold) const;                  // :337-349
    bool is_synced() const { return m_.is_synced(); }
    void wait() const { m_.wait(); }
    void resync();                                       // :355-358
    void latch(bool decision) { if (decision) fold_ = true; }             // :360-367
    bool fold_decision(bool (*policy)(u64, u64, u64), u64 base) const;     // :369-377
    bool folding() const { return fold_; }
    bool take_fold();                                    // :383-385
    void clear(u64 nrows, u64 ncols);                    // :387-398
    u64 count() const { return count_; }
    u64 nvals() const { return m_.nvals(); }
    void insert(u64 i, u64 j, u64 v = 1);                // :429-437
    void erase(u64 i, u64 j);                            // :414-422
    std::optional<u64> get(u64 i, u64 j) const { return m_.get(i, j); }
    bool contains(u64 i, u64 j) const { return m_.contains(i, j); }
    void tombstone_masked(const Matrix& mask, const Matrix& base);         // :439-447
    void remove_all(const Matrix& mask);                 // :451-458
    void replace(Matrix m) { m_ = std::move(m); }
    Delta transposed() const;                            // :320-335 (bookkeeping carried verbatim)

   private:
    Matrix m_;
    u64 count_ = 0, tx_nvals_ = 0;
    bool fold_ = false;
};

// the 3-way sorted merge of versioned_matrix::Iter (:1116-1253) over already-extracted layer rows
std::vector<Entry> merge_layers(const std::vector<Entry>& m, const std::vector<Entry>& dp,
                                const std::vector<Entry>& dm);

// VersionedMatrix<bool> — the Delta_Matrix (versioned_matrix.rs:480-1079): committed base m, pending
// additions dp, tombstones dm.  Invariants: dp & m = {}, dm subset of m, dp & dm = {}.
class VersionedMatrix {
   public:
    VersionedMatrix(Context& ctx, u64 nrows, u64 ncols);                   // :494-511
    static VersionedMatrix from_matrix(Matrix m);                          // :877-890
    const Matrix& m() const { return m_; }                                 // :514-528
    const Matrix& dp() const { return dp_.layer(); }
    const Matrix& dm() const { return dm_.layer(); }
    u64 nrows() const { return m_.nrows(); }
    u64 ncols() const { return m_.ncols(); }
    void wait() const;                                                     // :545-556
    void wait_all() const;                                                 // :562-566
    u64 nvals() const;                                                     // :629-632
    Matrix extract() const;                                                // :609-620
    std::optional<bool> get(u64 i, u64 j) const;                           // :819-835
    std::vector<Entry> iter(u64 min_row, u64 max_row) const;               // :647-654 + Iter :1116-1253
    void flush();                                                          // :892-938
    void set(u64 i, u64 j, bool v = true);                                 // :844-857
    void remove(u64 i, u64 j);                                             // :780-791
    void remove_mask(const Matrix& mask);                                  // :799-816
    void set_all(const std::vector<std::pair<u64, u64>>& entries, bool is_new);   // :1006-1035
    VersionedMatrix dup() const;                                           // :1038-1051
    void fold_oversized();                                                 // :953-965
    void fold_latched();                                                   // :940-951
    void resize(u64 nrows, u64 ncols);                                     // :967-1004
    VersionedMatrix transpose() const;                                     // :1070-1079
    bool needs_flush() const { return needs_flush_; }
    const Delta& dp_delta() const { return dp_; }
    const Delta& dm_delta() const { return dm_; }

   private:
    VersionedMatrix(Matrix m, Delta dp, Delta dm) : m_(std::move(m)), dp_(std::move(dp)), dm_(std::move(dm)) {}
    Matrix m_;
    mutable Delta dp_, dm_;
    bool needs_flush_ = false;
};

constexpr u64 MULTI_EDGE = ~0ull;                        // tensor.rs:207
constexpr u64 GrB_INDEX_MAX = (1ull << 60) - 1;          // tensor.rs:136
u64 compound_key(u64 src, u64 dst);                      // tensor.rs:154-163 (throws when an id needs > 32 bits)

// Tensor (tensor.rs:184-205): per-relationship-type edge storage with inline edge ids.  Forward layers
// (m, dp UINT64; dm BOOL) and the BOOL transpose `mt` live on the device.  `me` — all ids of the pairs
// that hold more than one edge, keyed by compound_key(src, dst) — is a GrB_INDEX_MAX-square hypersparse
// matrix in the reference; its keys do not fit the 32-bit device id space and it is only ever read one
// row at a time, so it is kept as an ordered host map with the same observable behaviour (ascending ids).
class Tensor {
   public:
    Tensor(Context& ctx, u64 nrows, u64 ncols);                            // tensor.rs:241-251
    u64 nrows() const { return m_.nrows(); }
    u64 ncols() const { return m_.ncols(); }
    void wait_fwd() const;                                                 // :265-282
    std::optional<u64> eff_get(u64 src, u64 dst) const;                    // :286-299
    std::vector<u64> get(u64 src, u64 dst) const;                          // :307-319 (ascending edge ids)
    // batched eff_get / get: three device probes for the whole list of pairs
    void get_batch(const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                   std::vector<std::vector<u64>>& ids) const;
    void set_all_from_slices(const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                             const std::vector<u64>& ids);                 // :333-455
    std::vector<std::pair<u64, u64>> remove_all(const std::vector<std::array<u64, 3>>& rels);  // :461-657
    bool has_multi_edge() const { return !me_.empty(); }
    void resize(u64 nrows, u64 ncols);                                     // :659-726
    void flush();                                                          // :741-797
    void fold_oversized();                                                 // :815-833
    Matrix extract() const;                                                // :838-850
    Tensor dup() const;                                                    // :871-889
    const Matrix& fwd_m() const { return m_; }                             // :891-907
    const Matrix& fwd_dp() const { return dp_.layer(); }
    const Matrix& fwd_dm() const { return dm_.layer(); }
    const VersionedMatrix& matrix_t() const { return mt_; }                // :936-943
    std::vector<Entry> structural_iter(u64 min_row, u64 max_row) const;    // :909-921
    std::vector<Entry> iter_edges() const;                                 // :973-989 (src, dst, edge id)
    u64 edge_count() const;                                                // :955-967
    // Encode<19> / Decode<19> for Tensor (tensor.rs:1053-1209): three container payloads (effective forward matrix —
    // single-edge pairs store the id, multi-edge pairs `count | 1 << 63` — then empty dp and dm), the edge count, and
    // the tensor section: two groups (base, delta-plus) of (src, dst, id-list blob).  The blob is what C FalkorDB
    // writes with GxB_Vector_serialize — GraphBLAS' own serialisation of a BOOL vector whose INDICES are the edge ids;
    // it can only be produced / parsed by GraphBLAS itself, so it goes through `BlobCodec` (see serialize.cpp).  The
    // backward matrix is rebuilt after decode (rebuild_backward, as the reference's caller does).
    struct BlobCodec {
        std::function<std::vector<uint8_t>(const std::vector<u64>& ids)> encode;
        std::function<std::vector<u64>(const std::vector<uint8_t>& blob)> decode;
    };
    static const BlobCodec& plain_blob_codec();
    void encode(ByteWriter& w, const BlobCodec& codec = plain_blob_codec()) const;
    static Tensor decode(Context& ctx, ByteReader& r, const BlobCodec& codec = plain_blob_codec());
    void rebuild_backward();
    u64 multi_pairs() const { return me_.size(); }
    u64 me_nvals() const { u64 n = 0; for (auto& kv : me_) n += kv.second.size(); return n; }   // `me.nvals()` of the tests
    void wait() const { wait_fwd(); mt_.wait(); }                           // Tensor::wait: every layer materialised

   private:
    Matrix m_;
    mutable Delta dp_, dm_;
    VersionedMatrix mt_;
    std::map<u64, std::vector<u64>> me_;   // compound key -> ascending edge ids (pairs with >= 2 edges)
    bool needs_flush_ = false;
};

using LabelId = u64;

// The traversal-facing slice of graph.rs.
class Graph {
   public:
    Graph(Context& ctx, u64 node_cap, u64 label_cap = 64);
    Context& ctx() const { return *ctx_; }
    u64 node_cap() const { return n_; }
    VersionedMatrix& adjacency_matrix() { return adj_; }                   // graph.rs:2251
    const VersionedMatrix& adjacency_matrix() const { return adj_; }
    VersionedMatrix& node_labels_matrix() { return labels_; }              // graph.rs:1057-1066
    const VersionedMatrix& node_labels_matrix() const { return labels_; }
    std::vector<Tensor>& relationship_tensors() { return tensors_; }       // graph.rs:2256
    const std::vector<Tensor>& relationship_tensors() const { return tensors_; }
    LabelId add_label(const std::string& name);
    u64 add_type(const std::string& name);
    std::optional<LabelId> label_id(const std::string& name) const;
    std::optional<u64> type_id(const std::string& name) const;
    void label_node(u64 node, LabelId l) { labels_.set(node, l, true); }
    void delete_node(u64 node) { deleted_[node] = true; }
    bool is_node_deleted(u64 node) const { return !deleted_.empty() && deleted_.count(node) != 0; }
    u64 deleted_nodes_count() const { return deleted_.size(); }   // graph.rs deleted_nodes_count
    // create one edge of `type` (adjacency + tensor), the write-side minimum the tests need
    void create_edge(u64 type, u64 src, u64 dst, u64 edge_id);             // graph.rs:1493-1560 (effect only)
    void delete_edge(u64 type, u64 src, u64 dst, u64 edge_id);             // graph.rs:1623-1700 (effect only)
    void create_edges(u64 type, const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                      const std::vector<u64>& ids);                        // bulk form of create_edge
    void new_version();              // graph.rs:851-890: every matrix dup()s (fold decisions latch here)
    void fold_oversized_deltas();    // graph.rs:2155-2172, called at MVCC commit (mvcc_graph.rs:161-180)

    bool node_has_label_id(u64 node, LabelId l) const;                     // graph.rs:1057-1066
    // all labels known? (unknown label => the pattern can match nothing)  graph.rs:2554-2559
    std::optional<std::vector<LabelId>> resolve_label_ids(const std::vector<std::string>& labels) const;
    // N-bit bitmap of the nodes carrying every label of `ids` (all ones when ids is empty)
    std::vector<u64> label_bitmap(const std::vector<LabelId>& ids) const;
    Matrix build_relationship_matrix_unrestricted(const std::vector<u64>& type_ids) const;  // graph.rs:2520-2549
    Matrix build_adjacency_matrix(const std::vector<std::string>& types) const;             // graph.rs:3870-3894
    Matrix build_symmetric_adjacency_matrix(const std::vector<std::string>& types) const;   // graph.rs:3898-3907 (A + A')
    std::vector<u64> get_src_dest_relationships(u64 src, u64 dst, const std::vector<u64>& type_ids) const;  // :1797-1837

   private:
    Context* ctx_;
    u64 n_;
    VersionedMatrix adj_, labels_;
    std::vector<Tensor> tensors_;
    std::unordered_map<std::string, LabelId> label_ids_;
    std::unordered_map<std::string, u64> type_ids_;
    std::unordered_map<u64, bool> deleted_;

   public:
    // algo.BFS keeps its device plan (workspace, pinned control block, acceleration indexes) between calls for as
    // long as the adjacency it was built on is the same immutable snapshot — creating one costs more than the search
    struct BfsPlanCache {
        Matrix adj, adj_t;                 // keep the snapshots (and their cached indexes) alive
        fgpu_bfs_plan* plan = nullptr;
        std::string key;                   // relationship types the adjacency was built for
        // level[] / parent[] in pinned blocks of the context's pool (fgpu_host_alloc): fgpu_bfs_fetch DMAs into them — the
        // procedure reads the vectors element by element afterwards (algo_procedures.rs:1096-1160), it never needs a copy
        fgpu_ctx* raw = nullptr;
        int32_t* level_pin = nullptr;
        int64_t* parent_pin = nullptr;
        u64 pin_rows = 0;
        // one search at a time runs on the cached plan AND reads its pinned result blocks: the reference's worker pool calls
        // algo.BFS on a shared const Graph from several threads (threadpool.rs:89-128); a second caller that finds the plan
        // busy takes the one-shot path with vectors of its own instead of waiting
        std::mutex mu;
        BfsPlanCache(Matrix a, Matrix at) : adj(std::move(a)), adj_t(std::move(at)) {}
        ~BfsPlanCache() {
            if (plan) fgpu_bfs_plan_free(plan);
            if (level_pin) fgpu_free(raw, level_pin);
            if (parent_pin) fgpu_free(raw, parent_pin);
        }
    };
    mutable std::shared_ptr<BfsPlanCache> bfs_cache_;
    mutable std::mutex bfs_cache_mu_;      // orders readers and replacers of bfs_cache_ (the entry itself has its own lock)
    // the same for a partitioned search over a gang of contexts: balanced splits, one column slab (+ transpose) and one
    // slab plan per device — building them costs far more than a search, so they live as long as the
    // adjacency snapshot, the relationship filter and the gang stay the same
    struct BfsGangCache {
        Matrix adj;                         // keeps the snapshot the slabs were cut from alive (and comparable)
        std::string key;
        std::vector<fgpu_ctx*> gang;
        std::vector<u64> splits;
        std::vector<fgpu_mat*> slabs, slabs_t;
        std::vector<fgpu_bfs_plan*> plans;
        std::mutex mu;
        explicit BfsGangCache(Matrix a) : adj(std::move(a)) {}
        // frees the device objects (idempotent).  Called by the destructor and by Context::~Context of ANY context of the
        // gang, BEFORE that context is finalised: the slabs and plans of a gang live in several contexts, and none of them
        // may be touched once one of those is gone
        void release() {
            std::lock_guard<std::mutex> g(mu);
            for (auto* p : plans) if (p) fgpu_bfs_plan_free(p);
            for (auto* m : slabs_t) if (m) fgpu_mat_free(m);
            for (auto* m : slabs) if (m) fgpu_mat_free(m);
            plans.clear(); slabs_t.clear(); slabs.clear(); gang.clear();
        }
        ~BfsGangCache() { release(); }
    };
    static void register_gang_cache(const std::shared_ptr<BfsGangCache>& c);   // (matrix.cpp, next to Context::~Context)
    mutable std::shared_ptr<BfsGangCache> bfs_gang_cache_;
};

// A bound value of one batch row, reduced to what the traversal operators inspect.
struct Value {
    enum Kind { Unbound, Null, Node, Other } kind = Unbound;
    u64 id = 0;
    static Value node(u64 id) { return Value{Node, id}; }
    static Value null() { return Value{Null, 0}; }
};

struct Hop {                               // one (types, dst labels) step; hop 0 is the operator's own pattern
    std::vector<std::string> types;
    std::vector<std::string> dst_labels;
};

// Result columns of expand_batch, in emission order: the emitter gathers the parent columns by `active_row`
// and appends `dest` as a NodeIds column (+ `edge` as RelIds when a representative edge was bound),
// cond_traverse.rs:700-735.
struct ExpandedRows {
    std::vector<u64> active_row;   // index into the input batch
    std::vector<u64> dest;
    std::vector<u64> edge;         // empty unless bind_relationship
    // The batched path hands over the two columns as the DEVICE built them (fgpu_expand_pairs32): pinned blocks of the context's
    // pool filled by DMA — 16-bit row indices (a child batch holds at most 1024 rows, batch.rs:81; 32-bit ones when a host layer
    // coalesces more than 65536 rows into one call) and the destinations as the 32-bit node ids the device works in (every node
    // id of a traversed graph fits 32 bits: tensor.rs:154-163 demands it of the multi-edge keys); row_at / dest_at widen on
    // access, which is where the reference's NodeId(u64) is needed.  They are owned until clear(); materialize() copies them into
    // the vectors above for the rare consumers that edit the columns in place.
    fgpu_ctx* pin_ctx = nullptr;
    const uint16_t* row_pin = nullptr;
    const uint32_t* row_pin32 = nullptr;
    const uint32_t* dest_pin = nullptr;
    size_t n_pin = 0;
    ExpandedRows() = default;
    ExpandedRows(const ExpandedRows&) = delete;
    ExpandedRows& operator=(const ExpandedRows&) = delete;
    ~ExpandedRows() { release_pinned(); }
    bool pinned() const { return dest_pin != nullptr; }
    size_t size() const { return pinned() ? n_pin : dest.size(); }
    u64 row_at(size_t i) const { return pinned() ? (row_pin ? (u64)row_pin[i] : (u64)row_pin32[i]) : active_row[i]; }
    u64 dest_at(size_t i) const { return pinned() ? (u64)dest_pin[i] : dest[i]; }
    void release_pinned();
    void materialize();
    void clear() { active_row.clear(); dest.clear(); edge.clear(); release_pinned(); }
};

// CondTraverseOp (runtime/ops/cond_traverse.rs).  Only the matrix path and its eligibility rule are
// mirrored; the attribute filters of the per-row path need the attribute store (out of scope).
struct CondTraverseOp {
    std::vector<std::string> src_labels;
    std::vector<Hop> hops;               // hops[0] + chain (fuse_anonymous_traverse.rs:83-188)
    bool optional = false;
    bool bind_relationship = false;      // representative edge wanted (cond_traverse.rs:663-695)
    bool emit_relationship = false, bidirectional = false, has_sibling_edges = false, has_inline_attrs = false;
    // the pattern runs against the storage direction: the bound alias is the matrix DESTINATION (cond_traverse.rs:221-235).
    // The reference serves these per row only; expand_batch takes them over the transposed layers (an extension, row-equal)
    bool transposed = false;

    // cond_traverse.rs:308-316: the batched matrix path may run at all
    bool batched_eligible() const;
    // expand_batch (cond_traverse.rs:452-751).  Returns false when the batch must take the per-row path
    // (a non-node source on a non-optional traverse, :566-575); otherwise fills `rows` in emission order
    // and `null_rows` with the active rows an OPTIONAL traverse null-pads (:737-747).
    bool expand_batch(const Graph& g, const std::vector<Value>& src, const std::vector<Value>* to_bound,
                      ExpandedRows& rows, std::vector<u64>& null_rows, u64* flops = nullptr) const;
    // (scan source, final dest) pairs already emitted by an anonymous bidirectional CT whose child is one too
    // (cond_traverse.rs:176-186, 262-299); cleared for every new input batch (:1268-1270)
    struct BidirDedup { std::set<std::pair<u64, u64>> seen; };
    // expand_row + process_pairs (cond_traverse.rs:758-974, 978-1117) without attribute filters: forward pass (over
    // the transposed structure when only the matrix destination is bound), the reverse pass of a bidirectional
    // pattern (self-loops dropped there), label / endpoint checks, per-pair edge lookup (one representative edge
    // unless emit_relationship), cross-row (src, dst) dedup.  An endpoint bound to a non-node gives no rows: the
    // caller does not call.  `dedup_src` = the value of the dedup source alias on this row.
    void expand_row(const Graph& g, std::optional<u64> from_id, std::optional<u64> to_id, bool transposed,
                    const std::vector<u64>& used_edges, std::vector<std::array<u64, 3>>& out,
                    BidirDedup* dedup = nullptr, std::optional<u64> dedup_src = std::nullopt) const;
};

// ExpandIntoOp (runtime/ops/expand_into.rs:121-258)
struct ExpandIntoOp {
    std::vector<std::string> types;
    bool bidirectional = false;
    bool emit_relationship = true;
    // one row: (src, dst, edge id) triples, types in order, ids ascending per type
    std::vector<std::array<u64, 3>> expand_row(const Graph& g, u64 src, u64 dst,
                                               const std::vector<u64>& used_edges = {}) const;
    // the whole batch with three device probes per type instead of per-pair calls
    void expand_batch(const Graph& g, const std::vector<u64>& srcs, const std::vector<u64>& dsts,
                      std::vector<std::vector<std::array<u64, 3>>>& out) const;
};

// CondVarLenTraverse (runtime/ops/cond_var_len_traverse.rs:81-387): the per-row DFS of `(a)-[:T*min..max]->(b)` with
// Cypher trail semantics (a relationship is used at most once per path, nodes may repeat), in the reference's emission
// order: frames leave a LIFO stack, a frame's emissions follow its adjacency order (types in order; per type the
// outgoing edges by (dst, id) ascending, then the incoming ones by (src, id) ascending, a self-loop only once).
// The device is the DFS's pruning oracle when the destination is bound: the nodes that can still reach it within r more
// hops (r = 1 .. max_hops - 1; walks, a superset of trails) come from r boolean products on the device, and a branch that
// is not in the set for its remaining budget is never expanded — the emitted sequence is unchanged, only frames that
// could not emit disappear.  Edge-attribute / WHERE filters need the attribute store (out of scope).
struct VarLenRow {
    u64 from, to;
    std::vector<u64> path;   // emit_path: node, edge, node, ... in pattern order (from -> to); else empty
};
struct VarLenStats {
    u64 frames = 0;          // DFS frames expanded (adjacency lists walked)
    u64 pruned = 0;          // continuations the reach sets cut
    u64 reach_products = 0;  // device products spent on the reach sets
};
struct CondVarLenTraverseOp {
    std::vector<std::string> types;
    std::vector<std::string> dst_labels;   // of the pattern's far endpoint (the one the DFS walks towards)
    bool reversed = false;                 // the bound endpoint is the pattern's `to`: walk incoming edges
    bool bidirectional = false;
    uint32_t min_hops = 1, max_hops = 1;   // UINT32_MAX = unbounded
    bool emit_path = false;
    bool prune = true;                     // use the device reach sets when dest is bound
    // one input row: DFS from `start`; `dest` = the other endpoint when it is bound too
    void expand_row(const Graph& g, u64 start, std::optional<u64> dest, std::vector<VarLenRow>& out,
                    VarLenStats* stats = nullptr) const;
};

struct BfsResult {
    bool has_row = false;
    std::vector<u64> nodes, edges;
};
// algo.BFS (runtime/functions/algo_procedures.rs:1021-1160)
// `gang` (nullable): contexts of the GPUs to partition the search over (one nnz-balanced column slab each; gang[0] is
// normally the graph's own context); the level loop and the frontier exchange then run inside libfgpu.so
// (fgpu_bfs_dist_run).  Same results as the single-device call.
BfsResult algo_bfs(const Graph& g, std::optional<u64> source, int64_t max_depth,
                   const std::optional<std::string>& rel_type, bool want_edges,
                   const std::vector<Context*>* gang = nullptr);

// ---- planner slice: the rule that decides which queries reach the fused chain ---------------------------
// (planner/optimizer/fuse_anonymous_traverse.rs).  The model keeps only what the rule inspects.
struct PlanNodeRef {
    std::string alias;                 // "_anon*" = anonymous (fuse_anonymous_traverse.rs:38-40)
    std::vector<std::string> labels;
    bool attrs_empty = true;           // inline attribute map is `{}` (:55-63)
};
struct PlanRel {                       // QueryRelationship
    std::string alias;
    PlanNodeRef from, to;
    std::vector<std::string> types;
    bool bidirectional = false;
    bool var_len = false;              // min_hops.is_some()
    bool attrs_empty = true;           // :43-53
};
struct PlanOp {
    enum Kind { CondTraverse, Other, Pruned } kind = Other;
    int parent = -1;
    std::vector<int> children;
    // IR::CondTraverse (planner/mod.rs:183-213)
    PlanRel rel;
    bool emit_relationship = false, transposed = false, optional = false, bind_relationship = true;
    std::vector<std::string> sibling_edges;
    std::vector<PlanRel> chain;
    // any other IR node: its name and the aliases its expressions reference (reduce_expand_into.rs:22-75)
    std::string name;
    std::vector<std::string> references;
};
struct Plan {
    std::vector<PlanOp> ops;           // index = node id
    int root = -1;
};
bool can_fuse(const Plan& plan, int parent_idx, int child_idx);            // fuse_anonymous_traverse.rs:83-188
void fuse_anonymous_traverse(Plan& plan);                                   // :190-284
CondTraverseOp lower_cond_traverse(const PlanOp& op);                       // plan node -> runtime operator
Plan parse_plan(const std::string& text);                                   // text form: see planner.cpp
std::string print_plan(const Plan& plan);

struct PageRankResult {
    std::vector<u64> nodes;
    std::vector<double> scores;   // Column::Floats: the FP32 centrality widened (extract_vector_f64)
};
// algo.pageRank (runtime/functions/algo_procedures.rs:687-783): label / rel_type NULL = all
PageRankResult algo_pagerank(const Graph& g, const std::optional<std::string>& label,
                             const std::optional<std::string>& rel_type);

}  // namespace falkor
