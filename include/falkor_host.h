/*
 * falkor_host.h — C surface of the C++ host layer (falkordb_amd/host/, libfalkor_host.so).
 *
 * The host layer is the C++17 stand-in for the Rust code that sits between FalkorDB's execution-plan
 * operators and the GraphBLAS boundary (Matrix<T>, VersionedMatrix, Tensor, the traversal slice of Graph,
 * CondTraverseOp::expand_batch, ExpandIntoOp, algo.BFS).  It calls the device engine only through
 * include/fgpu.h.  This header flattens it to plain C so that tests (ctypes) and embedders can drive it;
 * every function names the reference method it forwards to (file:line relative to
 * /root/reference/graph/src).  Return value: 0 ok, 1 = GrB_NO_VALUE where noted, negative = fgpu_info;
 * fh_last_error() holds the message.  Out arrays are malloc'ed and released with fh_free().
 */
#ifndef FALKOR_HOST_H
#define FALKOR_HOST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fh_ctx fh_ctx;     /* matrix::init context                         graphblas/matrix.rs:116-221 */
typedef struct fh_mat fh_mat;     /* Matrix<bool> / Matrix<u64>                   graphblas/matrix.rs */
typedef struct fh_vm fh_vm;       /* VersionedMatrix<bool> (Delta_Matrix)         graphblas/versioned_matrix.rs */
typedef struct fh_graph fh_graph; /* traversal-facing slice of Graph              graph/graph.rs */

int fh_init(fh_ctx** ctx, int device);
void fh_finalize(fh_ctx* ctx);
const char* fh_last_error(void);
void fh_free(void* p);

/* fold policy — versioned_matrix.rs:152-200 (pure integer arithmetic; needs no device) */
int fh_should_fold(uint64_t delta_nvals, uint64_t tx_added, uint64_t base_nvals);
int fh_should_fold_read(uint64_t delta_nvals, uint64_t tx_added, uint64_t base_nvals);
int fh_delta_dominates_base(uint64_t delta_nvals, uint64_t base_nvals);
/* tensor.rs:154-163; returns -3 when an id needs more than 32 bits */
int fh_compound_key(uint64_t src, uint64_t dst, uint64_t* key);

/* ---- Matrix<T>: type 0 = bool, 1 = u64 ------------------------------------------------------------ */
int fh_mat_new(fh_ctx* ctx, fh_mat** out, int type, uint64_t nrows, uint64_t ncols);   /* matrix.rs:1151, 1214 */
void fh_mat_free(fh_mat* m);
int fh_mat_build(fh_mat* m, const uint64_t* rows, const uint64_t* cols, const uint64_t* vals, uint64_t n); /* :1186, :1281 */
int fh_mat_set(fh_mat* m, uint64_t i, uint64_t j, uint64_t v);                         /* :1174, :1264 */
int fh_mat_remove(fh_mat* m, uint64_t i, uint64_t j);                                  /* :664 */
int fh_mat_get(fh_mat* m, uint64_t i, uint64_t j, uint64_t* v);                        /* :1158, :1248; 1 = NO_VALUE */
int fh_mat_nvals(fh_mat* m, uint64_t* out);                                            /* :722 */
int fh_mat_dims(fh_mat* m, uint64_t* nrows, uint64_t* ncols);
int fh_mat_pending(fh_mat* m, int* out);                                               /* :764 */
int fh_mat_wait(fh_mat* m);                                                            /* :781 */
/* matrix::Iter as a reusable streaming cursor (matrix.rs:1471-1605): new / seek (re-aim, :1542-1570) / next.
 * fh_mat_cursor_next fills up to `cap` entries of the caller's arrays (vals nullable); *n < cap = exhausted. */
typedef struct fh_mat_cursor fh_mat_cursor;
int fh_mat_cursor_new(fh_mat* m, uint64_t min_row, uint64_t max_row, fh_mat_cursor** out);
int fh_mat_cursor_seek(fh_mat_cursor* c, uint64_t min_row, uint64_t max_row);
int fh_mat_cursor_next(fh_mat_cursor* c, uint64_t cap, uint64_t* rows, uint64_t* cols, uint64_t* vals, uint64_t* n);
void fh_mat_cursor_free(fh_mat_cursor* c);
int fh_mat_iter(fh_mat* m, uint64_t min_row, uint64_t max_row, uint64_t** rows, uint64_t** cols,
                uint64_t** vals, uint64_t* n);                                         /* Iter :1471-1605 */
int fh_mat_dup(fh_mat* m, fh_mat** out);                                               /* :370 */
int fh_mat_transpose(fh_mat* m, fh_mat** out);                                         /* :633 */
int fh_mat_grown(fh_mat* m, uint64_t nrows, uint64_t ncols, fh_mat** out);             /* :664-704 */
int fh_mat_resize(fh_mat* m, uint64_t nrows, uint64_t ncols);                          /* :576 */
int fh_mat_lmxm(fh_mat* self, fh_mat* b);                                              /* :930 */
int fh_mat_rmxm(fh_mat* self, fh_mat* b);                                              /* :951 */
int fh_mat_delta_lmxm(fh_mat* self, fh_mat* m, fh_mat* dp, fh_mat* dm);                /* :1317 */
int fh_mat_intersection_nvals(fh_mat* a, fh_mat* b, uint64_t* out);                    /* :743 */

/* ---- VersionedMatrix<bool> ---------------------------------------------------------------------- */
int fh_vm_new(fh_ctx* ctx, fh_vm** out, uint64_t nrows, uint64_t ncols);               /* versioned_matrix.rs:494 */
int fh_vm_from_coo(fh_ctx* ctx, fh_vm** out, uint64_t nrows, uint64_t ncols, const uint64_t* rows,
                   const uint64_t* cols, uint64_t n);                                  /* from_matrix :877 */
void fh_vm_free(fh_vm* v);
int fh_vm_set(fh_vm* v, uint64_t i, uint64_t j);                                       /* :844 */
int fh_vm_remove(fh_vm* v, uint64_t i, uint64_t j);                                    /* :780 */
int fh_vm_get(fh_vm* v, uint64_t i, uint64_t j);                                       /* :819; 0 present, 1 NO_VALUE */
int fh_vm_nvals(fh_vm* v, uint64_t* out);                                              /* :629 */
int fh_vm_iter(fh_vm* v, uint64_t min_row, uint64_t max_row, uint64_t** rows, uint64_t** cols, uint64_t* n); /* :647 */
int fh_vm_set_all(fh_vm* v, const uint64_t* rows, const uint64_t* cols, uint64_t n, int is_new);   /* :1006 */
int fh_vm_remove_mask(fh_vm* v, const uint64_t* rows, const uint64_t* cols, uint64_t n);           /* :799 */
int fh_vm_dup(fh_vm* v, fh_vm** out);                                                  /* :1038 */
int fh_vm_wait(fh_vm* v);                                                              /* :545 */
int fh_vm_flush(fh_vm* v);                                                             /* :892 */
int fh_vm_fold_oversized(fh_vm* v);                                                    /* :953 */
int fh_vm_extract(fh_vm* v, fh_mat** out);                                             /* :609 */
int fh_vm_transpose(fh_vm* v, fh_vm** out);                                            /* :1070 */
/* out[0..2] = nvals of m, dp, dm; out[3] = needs_flush */
int fh_vm_state(fh_vm* v, uint64_t out[4]);

/* ---- Graph slice ------------------------------------------------------------------------------------ */
int fh_graph_new(fh_ctx* ctx, fh_graph** out, uint64_t node_cap);
void fh_graph_free(fh_graph* g);
int fh_graph_add_label(fh_graph* g, const char* name, uint64_t* id);
int fh_graph_add_type(fh_graph* g, const char* name, uint64_t* id);
int fh_graph_label_node(fh_graph* g, uint64_t node, uint64_t label_id);
int fh_graph_delete_node(fh_graph* g, uint64_t node);
int fh_graph_create_edge(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t edge_id);
int fh_graph_create_edges(fh_graph* g, uint64_t type_id, const uint64_t* srcs, const uint64_t* dsts,
                          const uint64_t* ids, uint64_t n);   /* Tensor::set_all_from_slices + adjacency set_all */
int fh_graph_delete_edge(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t edge_id);
/* MVCC commit of every matrix: dup() (fold decision) then fold_oversized, as mvcc_graph.rs:161-180 does */
int fh_graph_commit(fh_graph* g);
int fh_graph_node_has_label(fh_graph* g, uint64_t node, uint64_t label_id);            /* graph.rs:1057; 0 yes, 1 no */
int fh_tensor_get(fh_graph* g, uint64_t type_id, uint64_t src, uint64_t dst, uint64_t** ids, uint64_t* n); /* tensor.rs:307 */
int fh_tensor_edge_count(fh_graph* g, uint64_t type_id, uint64_t* out);                /* tensor.rs:955 */
int fh_tensor_iter_edges(fh_graph* g, uint64_t type_id, uint64_t** srcs, uint64_t** dsts, uint64_t** ids,
                         uint64_t* n);                                                  /* tensor.rs:973 */
/* out[0..2] = nvals of fwd m, dp, dm; out[3] = multi pairs; out[4] = nvals of mt (effective) */
int fh_tensor_state(fh_graph* g, uint64_t type_id, uint64_t out[5]);

/* A Tensor on its own (tensor.rs:184-989) — what the unit tests of tensor.rs:1340-1669 drive.
 * rels = n triples (edge id, src, dst) as Tensor::remove_all takes them; emptied = the pairs left without an edge.
 * fh_tn_op: 0 flush, 1 fold_oversized, 2 wait (every layer), 3 wait_fwd, 4 resize(a, b).
 * fh_tn_probe: which 0 = effective forward value (eff_get; the MULTI_EDGE sentinel is UINT64_MAX), 1 = committed base
 * m, 2 = extract() pattern; returns 1 (GrB_NO_VALUE) when absent.
 * fh_tn_state: nvals of m, dp, dm; multi pairs; nvals of mt.extract(); ids held in me; edge_count; m.pending(). */
typedef struct fh_tn fh_tn;
int fh_tn_new(fh_ctx* ctx, fh_tn** out, uint64_t nrows, uint64_t ncols);
void fh_tn_free(fh_tn* t);
int fh_tn_dup(fh_tn* t, fh_tn** out);
int fh_tn_set_all(fh_tn* t, const uint64_t* srcs, const uint64_t* dsts, const uint64_t* ids, uint64_t n);
int fh_tn_remove_all(fh_tn* t, const uint64_t* rels, uint64_t n, uint64_t** emptied_src, uint64_t** emptied_dst,
                     uint64_t* n_emptied);
int fh_tn_op(fh_tn* t, int op, uint64_t a, uint64_t b);
int fh_tn_get(fh_tn* t, uint64_t src, uint64_t dst, uint64_t** ids, uint64_t* n);
int fh_tn_probe(fh_tn* t, int which, uint64_t src, uint64_t dst, uint64_t* val);
int fh_tn_state(fh_tn* t, uint64_t out[8]);
/* Encode<19> / Decode<19> for Tensor (tensor.rs:1053-1209): forward matrix (multi-edge pairs as count | 1 << 63), empty
 * dp / dm, edge count, tensor section (base group, delta-plus group) of (src, dst, id-list blob).  The blob of the
 * default codec is this library's plain list; a GxB_Vector_serialize blob needs GraphBLAS (see serialize.cpp). */
int fh_tn_encode(fh_tn* t, uint8_t** bytes, uint64_t* len);
int fh_tn_decode(fh_ctx* ctx, const uint8_t* bytes, uint64_t len, fh_tn** out, uint64_t* consumed);

/* Raw layer dump for tests: type_id < 0 = the adjacency matrix; which 0 = m, 1 = dp, 2 = dm. */
int fh_graph_layer_iter(fh_graph* g, int64_t type_id, int which, uint64_t** rows, uint64_t** cols,
                        uint64_t** vals, uint64_t* n);

/* ---- operators ---------------------------------------------------------------------------------------
 * `spec` is "key=value;..." with keys: src=<labels,>  hop=<types,>|<dst labels,> (repeatable: hop 0 then the
 * fused chain)  optional= bind= emit= bidir= siblings= attrs= transposed= (0/1).
 * src[i] / to_bound[i]: node id, -1 = unbound, -2 = bound to NULL / a non-node.
 * transposed=1: src[i] is the matrix DESTINATION (cond_traverse.rs:221-235); the batch runs over the transposed layers and
 * out_dest holds the matrix sources reached — row-equal to fh_cond_traverse_rows(..., transposed = 1) (an extension: the
 * reference takes these per row). */
int fh_cond_traverse_batch(fh_graph* g, const char* spec, const int64_t* src, const int64_t* to_bound,
                           uint64_t k, int* batched, uint64_t** out_row, uint64_t** out_dest,
                           int64_t** out_edge, uint64_t* n, uint64_t** null_rows, uint64_t* n_null,
                           uint64_t* flops);                                            /* cond_traverse.rs:452-751 */
/* duration (ns) of the C++ operator inside this thread's last fh_cond_traverse_batch / fh_algo_bfs / fh_algo_pagerank call:
 * what the operator costs without the ctypes harness' result copies (tools/bench_paths.py host) */
uint64_t fh_last_op_ns(void);
int fh_cond_traverse_eligible(const char* spec);                                       /* cond_traverse.rs:308-316 */
int fh_cond_traverse_row(fh_graph* g, const char* spec, int64_t from_id, int64_t to_id, int transposed,
                         uint64_t** out_from, uint64_t** out_to, uint64_t** out_edge, uint64_t* n); /* :758-1117 */
/* The per-row fallback over one input batch of k rows (from / to: node id, -1 unbound, -2 bound to a non-node), with
 * the sibling-edge uniqueness list and, when dedup_src != NULL, the cross-row (scan source, final dest) dedup of an
 * anonymous bidirectional CT over an anonymous bidirectional child (cond_traverse.rs:262-299, 948-970). */
int fh_cond_traverse_rows(fh_graph* g, const char* spec, const int64_t* from_ids, const int64_t* to_ids,
                          const int64_t* dedup_src, uint64_t k, int transposed, const uint64_t* used_edges,
                          uint64_t n_used, uint64_t** out_row, uint64_t** out_from, uint64_t** out_to,
                          uint64_t** out_edge, uint64_t* n);
/* types: comma list ("" = all).  batched != 0 runs the whole input through one set of device probes. */
int fh_expand_into(fh_graph* g, const char* types, int bidirectional, int emit_relationship, int batched,
                   const uint64_t* srcs, const uint64_t* dsts, uint64_t k, uint64_t** out_row,
                   uint64_t** out_src, uint64_t** out_dst, uint64_t** out_edge, uint64_t* n);  /* expand_into.rs:121-258 */
/* CondVarLenTraverse, one input row (cond_var_len_traverse.rs:81-387): the trail DFS of (start)-[:types*min..max]-(far end)
 * in the reference's emission order.  types / dst_labels: comma lists ("" = all / none).  reversed: the bound endpoint is the
 * pattern's `to` (walk incoming edges); max_hops = UINT32_MAX: unbounded; dest < 0: the far end is not bound.  emit_path:
 * `path` holds node, edge, node, ... of every row back to back in pattern order, `path_off` (n + 1 offsets) cuts it.
 * prune != 0: with a bound far end the device computes which nodes can still reach it within the remaining budget and
 * branches outside that set are not expanded (same rows, same order).  stats (nullable): frames expanded, continuations
 * pruned, device products spent on the reach sets. */
int fh_var_len_traverse(fh_graph* g, const char* types, const char* dst_labels, int reversed, int bidirectional,
                        uint32_t min_hops, uint32_t max_hops, uint64_t start, int64_t dest, int emit_path, int prune,
                        uint64_t** out_from, uint64_t** out_to, uint64_t** path, uint64_t** path_off, uint64_t* n,
                        uint64_t stats[3]);
/* source < 0 = NULL; rel_type NULL = all types.  `edges` holds one id per (parent, child) pair that HAS a
 * representative edge among the types, in node order; a pair without one is skipped in `edges` but its child stays in
 * `nodes` — exactly what the reference yields (algo_procedures.rs:1121-1150) — so the two lists are parallel only
 * while n_edges == n_nodes, which holds whenever the adjacency was built from the same types. */
int fh_algo_bfs(fh_graph* g, int64_t source, int64_t max_depth, const char* rel_type, int want_edges,
                int* has_row, uint64_t** nodes, uint64_t* n_nodes, uint64_t** edges, uint64_t* n_edges); /* algo_procedures.rs:1021-1160 */

/* The same procedure partitioned over the contexts of `gang` (one per GPU; gang[0] = the graph's context): nnz-balanced
 * column slabs, level loop + frontier exchange inside libfgpu.so (fgpu_bfs_dist_run).  Same output. */
int fh_algo_bfs_multi(fh_graph* g, fh_ctx* const* gang, int n_gang, int64_t source, int64_t max_depth,
                      const char* rel_type, int want_edges, int* has_row, uint64_t** nodes, uint64_t* n_nodes,
                      uint64_t** edges, uint64_t* n_edges);

/* The v19 on-disk form of a Matrix<T> (Encode<19> / Decode<19>, matrix.rs:428-546): the 608 bytes of
 * GxB_Container_struct + its vectors x, h, p, i, b in Vector<bool>'s unload-to-array form (vector.rs:241-309).
 * Stream framing: unsigned / signed = 8 bytes little-endian, buffer = unsigned length + bytes.
 * fh_container_parse is CPU-only: dims = {nrows, ncols, nvals, hypersparse, valued, bytes consumed}; p (np entries),
 * h (nh), i (nvals), x (nvals when valued) are returned as u64 arrays (fh_free). */
int fh_container_parse(const uint8_t* bytes, uint64_t len, uint64_t* dims, uint64_t** p, uint64_t* np, uint64_t** h,
                       uint64_t* nh, uint64_t** i, uint64_t** x);
int fh_mat_decode(fh_ctx* ctx, const uint8_t* bytes, uint64_t len, fh_mat** out, uint64_t* consumed);
int fh_mat_encode(fh_mat* m, uint8_t** bytes, uint64_t* len);   /* the wait()ed state; free with fh_free */

/* build_adjacency_matrix (graph.rs:3870-3894) or, symmetric != 0, build_symmetric_adjacency_matrix (:3898-3907):
 * types = comma-separated relationship types, "" / NULL = the adjacency of all types */
int fh_graph_build_adjacency(fh_graph* g, const char* types, int symmetric, fh_mat** out);

/* fuse_anonymous_traverse (planner/optimizer/fuse_anonymous_traverse.rs:83-284) on a plan in the text form
 * documented in falkordb_amd/host/planner.cpp; *out_text = the plan after the pass, *spec (nullable) = the runtime
 * spec (fh_cond_traverse_batch format) of CondTraverse node `lower_id` of the result.  Free both with fh_free. */
int fh_plan_fuse(const char* plan_text, int lower_id, char** out_text, char** spec);

/* label / rel_type NULL = all; nodes ascending, scores[k] = centrality of nodes[k] (free both with fh_free) */
int fh_algo_pagerank(fh_graph* g, const char* label, const char* rel_type, uint64_t** nodes, double** scores,
                     uint64_t* n);                                                          /* algo_procedures.rs:687-783 */

#ifdef __cplusplus
}
#endif
#endif /* FALKOR_HOST_H */
