/*
 * fgpu.h — C ABI of the MI355X-native traversal engine (device tier).
 *
 * This is the drop-in boundary for FalkorDB's sparse-linear-algebra hot path.
 * In the reference the boundary is the `extern "C"` GraphBLAS/LAGraph symbol set
 * bound by bindgen (graph/src/graph/graphblas/mod.rs:42-79) and consumed only
 * through `Matrix<T>` (graph/src/graph/graphblas/matrix.rs).  Every entry point
 * below names the reference call it replaces (file:line relative to
 * /root/reference).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add beside graphblas/mod.rs.
 *
 * Conventions (mirroring SURVEY.md §8b):
 *  - plain pointers and sizes only; no C++ / torch types cross this boundary;
 *  - every call returns an fgpu_info (values 0/1/-2/-3/-102/-105 mirror
 *    GrB_Info, graphblas/mod.rs:274-296); never aborts, never throws;
 *  - fgpu_last_error() returns a thread-local message for the last failure;
 *  - an fgpu_mat is an IMMUTABLE device-resident snapshot (the committed,
 *    `wait()`ed state of a reference Matrix: CSR, sorted unique columns per
 *    row, pattern-only for BOOL, +u64 values for UINT64 tensors);
 *  - node ids must fit in 32 bits (the reference's Tensor::compound_key makes
 *    the same demand, tensor.rs:154-163) and nnz per matrix must be < 2^32;
 *  - outputs returned through `T**` are host buffers owned by the caller until
 *    fgpu_free().
 *  - threading: ONE fgpu_ctx per process and device, shared by all host threads — the reference calls GraphBLAS
 *    from a worker pool on shared materialised handles (threadpool.rs:89-128, matrix.rs:781-796).  Every calling
 *    thread is bound, on its first call, to a LANE of the context: its own HIP stream, pinned staging block and
 *    free-list of device blocks, so concurrent calls never share a stream-ordered resource.  Any number of threads
 *    may concurrently call read-side entry points (fgpu_expand*, fgpu_mxm, fgpu_delta_lmxm, fgpu_mat_probe /
 *    extract / export, fgpu_vxm, fgpu_bfs, fgpu_pagerank, merges, transposes ...) on the SAME snapshots: a snapshot's
 *    stored content never changes, its acceleration indexes are built once under a per-snapshot mutex and
 *    published only when complete, and a call that returns a new fgpu_mat* synchronises its lane first whenever
 *    more than one lane exists, so a handle can be handed to another thread as soon as the call returns.
 *    What the caller keeps exclusive — exactly what the reference keeps exclusive (the caller's own F, a matrix
 *    under its `wait` mutex): an fgpu_bfs_plan is used by one thread at a time (run_async / wait on the same
 *    thread); fgpu_mat_free may be called from any thread once no call that takes the snapshot is executing
 *    (work other threads queued asynchronously is fenced inside the library); fgpu_set_option, fgpu_mat_build_tiles
 *    with explicit parameters and fgpu_prof_* are configuration / measurement calls made while no other call runs.
 *  - There is NO CPU fallback: without a HIP device fgpu_init fails with
 *    FGPU_DEVICE and nothing else can be called.
 */
#ifndef FGPU_H
#define FGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fgpu_ctx fgpu_ctx; /* one per process+device, shared by all threads (matrix::init, matrix.rs:116-185) */
typedef struct fgpu_mat fgpu_mat; /* immutable device CSR snapshot of one matrix layer        */
typedef struct fgpu_bfs_plan fgpu_bfs_plan; /* per-(A,At) BFS workspace + partition state       */

typedef int32_t fgpu_info;
#define FGPU_OK 0                  /* GrB_SUCCESS                */
#define FGPU_NO_VALUE 1            /* GrB_NO_VALUE               */
#define FGPU_NULL_POINTER (-2)     /* GrB_NULL_POINTER           */
#define FGPU_INVALID (-3)          /* GrB_INVALID_VALUE          */
#define FGPU_DIM_MISMATCH (-6)     /* GrB_DIMENSION_MISMATCH     */
#define FGPU_OOM (-102)            /* GrB_OUT_OF_MEMORY          */
#define FGPU_OUT_OF_BOUNDS (-105)  /* GrB_INDEX_OUT_OF_BOUNDS    */
#define FGPU_DEVICE (-7002)        /* GxB_GPU_ERROR              */

/* ---- lifecycle ---------------------------------------------------------- */

/* Replaces matrix::init -> GxB_init(GrB_NONBLOCKING, allocators) (matrix.rs:116-185).
 * `mal`/`fre` (nullable) are the host allocator hooks used for every buffer
 * handed back to the caller (the reference passes Redis' allocator). */
fgpu_info fgpu_init(fgpu_ctx** ctx, int device, void* (*mal)(size_t), void (*fre)(void*));
/* Replaces matrix::shutdown -> GrB_finalize (matrix.rs:214-221). */
fgpu_info fgpu_finalize(fgpu_ctx* ctx);
const char* fgpu_last_error(void);
void fgpu_free(fgpu_ctx* ctx, void* p);
/* Result arrays (`T**` outputs) of 256 KiB and more are PINNED host blocks from a pool the context keeps (SURVEY.md
 * §8b: "outputs are pinned-host buffers owned by the caller until fgpu_free"): the device fills them with one DMA, ids
 * widened to 64 bits by a kernel, and fgpu_free returns them to the pool (the first result of a size pays the pinning,
 * ~0.1 ms per MiB; later ones do not).  Smaller outputs come from `mal`.  fgpu_host_alloc hands the caller a block of
 * the same pool for the arrays the CALLER provides (level[] / parent[] of fgpu_bfs and fgpu_bfs_fetch, centrality[] of
 * fgpu_pagerank): entry points that write into caller memory check whether it is pinned (a pool block, or memory the
 * caller registered with HIP) and then DMA straight into it; pageable memory goes through a ring of four 2 MiB pinned
 * chunks and a host copy.  Release with fgpu_free.  (The reference's GrB_Vector results live in GraphBLAS-owned memory
 * and are read element by element, algo_procedures.rs:1096-1160; a maintainer's binding would allocate its level
 * vector here.) */
fgpu_info fgpu_host_alloc(fgpu_ctx* ctx, uint64_t bytes, void** out);
/* Run all subsequent work of the CALLING THREAD's lane on an externally owned hipStream_t
 * (plumbing for torch.distributed: pass torch.cuda.current_stream().cuda_stream).
 * NULL restores the lane's own stream. */
fgpu_info fgpu_set_stream(fgpu_ctx* ctx, void* hip_stream);
/* Wait for the work the calling thread has queued on its lane. */
fgpu_info fgpu_sync(fgpu_ctx* ctx);
/* Engine tunables (the analogue of GrB_Global_set_INT32, matrix.rs:151-159): "tiled_u" (items in
 * flight per wavefront of the LDS-tiled vxm: 1/2/4/8), "tiled_threads" (256/512/1024),
 * "tiled_wgs" (grid of that kernel, 0 = fill the CUs), "tiled_nt" (nontemporal entry loads),
 * "transpose_mode" (pattern transpose: 0 = counting transpose, 1 = COO rebuild through the sorter), "transpose_wb"
 * (low-digit bits of the counting transpose, 0 = pick; a process-wide experiment knob),
 * "expand_mode" (fgpu_expand: 0 = pick per hop, 1 = sorted-CSR products only, 2 = bit-parallel from the
 * first hop), "expand_fuse_count" (fgpu_expand_count: 1 = the last bit-parallel hop counts its rows where it produces
 * them, 0 = it writes them and a separate pass counts), "expand_row_groups" (sparse mid-chain pull of the bit-parallel
 * form: 1 = a wavefront per 32-row group, 0 = a wavefront per row item), "bfs_wgs_per_cu" (grid of the fused BFS level kernel), "merge_mode" (fgpu_mat_merge:
 * 0 = entry-parallel with the base layer's keep bits cleared from the delta side, 1 = one wavefront per row, pattern
 * layers only, 2 = entry-parallel with every base entry of a touched row searching the deltas), "bfs_tiny" (consecutive tiny BFS levels in one single-workgroup launch: 0 off, 1 on,
 * 2 = when the plan's previous search took more than 12 levels), "dist_collective" (frontier exchange of
 * fgpu_bfs_dist_run: 0 = grouped ncclSend / ncclRecv, 1 = one ncclBroadcast per rank), "dist_timing" (1 = fgpu_bfs_dist_run
 * records HIP events around every level kernel and exchange for fgpu_bfs_dist_times; off by default, the events cost
 * ~20 us of stream idle time per level), "bfs_prof_split" (1 = a profiled plan launches
 * the push / pull twins of the level kernel so rocprofv3 can tell them apart by name), "bfs_hub_first" (1 = BFS plans
 * read the pull direction from a copy of At whose rows are reordered by descending out-degree class), "bfs_pb" (heavy push
 * levels of a BFS by propagation blocking — the frontier's edges binned by destination window, a workgroup per window marking
 * its discoveries in LDS: 0 off, 1 = plans of at least 2^24 vertices, 2 = every single-rank plan) with "bfs_pb_min_edges" (a
 * push level with at least this many edges goes that way; default 2 Mi), "bfs_alive_rule" (1 = the push <-> pull rule takes
 * the unvisited share over the vertices that have an in-edge; 0 = over all vertices), "expand_emit_sort" (bit state -> rows of
 * fgpu_expand*: 2 = (row, vertex) pairs + a stable sort by row, 0 = ballot transpose, 1 = pairs + sort unless the result
 * holds more than 8 entries per vertex; the default), "pinned_results" / "pinned_pool_mb" (result arrays from 256 KiB up to
 * the pool's size come from the context's pinned pool and are filled by DMA; blocks kept for reuse up to that many MiB). */
fgpu_info fgpu_set_option(fgpu_ctx* ctx, const char* name, int64_t value);
/* Read-back of measurement / test counters kept by the context (a subset of the option names plus counters that
 * have no setter): "dist_force_self" (test-only, set through fgpu_set_option: a communicator of ONE rank still issues the
 * grouped self ncclSend / ncclRecv, ncclBroadcast and ncclAllReduce of a multi-rank exchange — tests/test_gpu_dist.py),
 * "dist_self_calls" (how many such calls ran), "expand_kernel_launches" (kernels launched by fgpu_expand* on this context so
 * far: the launch count of a batch is a difference of two reads), "bfs_pb_last_levels" (levels the search fgpu_bfs_stats last
 * read ran by propagation blocking), "bfs_cp_last_mask" (bit k: fused launch k of that search ran behind the list kernel — a
 * sparse frontier listed into the queue, or a pull of listed candidates), "expand_scan_last_live" / "expand_scan_last_passes" (live source rows and passes of the
 * last whole-frontier fgpu_expand_count).  Unknown names return FGPU_INVALID. */
fgpu_info fgpu_get_option(fgpu_ctx* ctx, const char* name, int64_t* value);
/* name[256]; returns CU count, wave size, LDS bytes per block, total HBM bytes. */
fgpu_info fgpu_device_info(fgpu_ctx* ctx, char* name, int32_t* cus, int32_t* wave,
                           int64_t* lds_bytes, int64_t* hbm_bytes);
/* Bytes currently held by the ctx's device pool (GRAPH.MEMORY analogue, graph.rs:3935). */
fgpu_info fgpu_device_bytes(fgpu_ctx* ctx, uint64_t* in_use, uint64_t* pooled);

/* ---- matrices ----------------------------------------------------------- */

/* Empty nrows x ncols matrix: Matrix::<bool>::new (matrix.rs:1214-1235). */
fgpu_info fgpu_mat_new(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols);

/* COO -> CSR with duplicate coordinates collapsed.
 * vals == NULL : Matrix::<bool>::build -> GxB_Matrix_build_Scalar (matrix.rs:1281-1303;
 *                dup-collapse pinned by matrix.rs:1686-1695).
 * vals != NULL : Matrix::<u64>::build -> GrB_Matrix_build_UINT64(..., GxB_ANY_UINT64)
 *                (matrix.rs:1186-1210); the LAST duplicate wins (a legal ANY). */
fgpu_info fgpu_mat_from_coo(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols,
                            const uint64_t* rows, const uint64_t* cols, const uint64_t* vals,
                            uint64_t n);

/* Host CSR (what GxB_unload_Matrix_into_Container yields: p, i, x [, h];
 * matrix.rs:508-546) -> device snapshot.  rowptr has nvec+1 entries when
 * `hyper_rows` (sorted non-empty row ids, nvec of them) is given, else
 * nrows+1.  rowptr_bits / colidx_bits are 32 or 64.  Rows must already be
 * sorted and unique (the state after Matrix::wait). */
fgpu_info fgpu_mat_from_csr(fgpu_ctx* ctx, fgpu_mat** out, uint64_t nrows, uint64_t ncols,
                            uint64_t nnz, const void* rowptr, int rowptr_bits,
                            const void* colidx, int colidx_bits, const uint64_t* vals,
                            const uint64_t* hyper_rows, uint64_t nvec);

/* Synthetic Graph500 R-MAT adjacency generated ON DEVICE (bench/tests data):
 * 2^scale vertices, edge_factor*2^scale raw edges, (a,b,c) quadrant probabilities
 * as 16.16 fixed point (0 -> Graph500 defaults .57/.19/.19), counter-based
 * splitmix64 keyed by `seed`, vertex ids scrambled by a fixed bijection, directed,
 * self-loops dropped, duplicates collapsed.  Not a reference API. */
fgpu_info fgpu_mat_rmat(fgpu_ctx* ctx, fgpu_mat** out, int scale, int edge_factor,
                        uint64_t seed, uint32_t a16, uint32_t b16, uint32_t c16);

fgpu_info fgpu_mat_free(fgpu_mat* m);                       /* Drop, matrix.rs:387-399 */
fgpu_info fgpu_mat_nrows(const fgpu_mat* m, uint64_t* out);  /* GrB_Matrix_nrows */
fgpu_info fgpu_mat_ncols(const fgpu_mat* m, uint64_t* out);  /* GrB_Matrix_ncols */
fgpu_info fgpu_mat_nvals(const fgpu_mat* m, uint64_t* out);  /* GrB_Matrix_nvals, matrix.rs:722-729 */
fgpu_info fgpu_mat_has_values(const fgpu_mat* m, int32_t* out);

/* D2H export of the whole snapshot in CSR form (GxB_Container p/i/x):
 * rowptr[nrows+1] (always full, hyper rows expanded), colidx[nnz], vals[nnz] or NULL. */
fgpu_info fgpu_mat_export_csr(fgpu_ctx* ctx, const fgpu_mat* m, uint64_t** rowptr,
                              uint64_t** colidx, uint64_t** vals, uint64_t* nnz);
/* Entries of rows [min_row, max_row] in ascending (row, col) order:
 * matrix::Iter::new/next (matrix.rs:1471-1605). */
fgpu_info fgpu_mat_extract(fgpu_ctx* ctx, const fgpu_mat* m, uint64_t min_row, uint64_t max_row,
                           uint64_t** rows, uint64_t** cols, uint64_t** vals, uint64_t* n);

/* GrB_transpose full: Matrix::transpose (matrix.rs:633-662).  Pattern only when
 * `a` is BOOL; values carried when UINT64. */
fgpu_info fgpu_mat_transpose(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a);

/* Batched point probes (K11): GrB_Matrix_extractElement_* / GxB_Matrix_isStoredElement
 * (matrix.rs:731-737, 1158-1172, 1248-1262).  present[i] = 1 iff (rows[i], cols[i])
 * is stored; vals (nullable) receives the u64 value (0 for BOOL). */
fgpu_info fgpu_mat_probe(fgpu_ctx* ctx, const fgpu_mat* m, const uint64_t* rows,
                         const uint64_t* cols, uint64_t n, uint8_t* present, uint64_t* vals);

/* Delta merge (K3/K6): out = (m \ dm) U dp, dp's value wins on a shared coordinate
 * (GrB_SECOND_UINT64) — VersionedMatrix::flush / extract (versioned_matrix.rs:609-620,
 * 892-938), Tensor::flush (tensor.rs:702-751).  dp/dm may be NULL.
 * `dm_masks_dp`: 0 => dp entries survive dm (extract / Tensor structure semantics,
 * where the invariants make it moot), 1 => eWiseAdd<!dm>(m, dp) exactly as flush's
 * (true,true) arm (mask applies to the whole union). */
fgpu_info fgpu_mat_merge(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                         const fgpu_mat* dm, int dm_masks_dp);
/* The same merge with the values dropped: out = pattern((m \ dm) U dp), a BOOL snapshot in one pass —
 * Tensor::extract (tensor.rs:838-850), Matrix::set_pattern = GrB_Matrix_apply(GxB_ONE_BOOL)
 * (matrix.rs:906-924), and with dp = dm = NULL the plain structure copy of a UINT64 layer that
 * build_relationship_matrix_unrestricted / build_adjacency_matrix start from (graph.rs:2520-2549, 3870-3894). */
fgpu_info fgpu_mat_merge_pattern(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* m, const fgpu_mat* dp,
                                 const fgpu_mat* dm, int dm_masks_dp);
/* GrB_Matrix_resize (Matrix::resize, matrix.rs:576-598; the `grown` re-emit of tensor.rs:613-667):
 * a new snapshot of `a` at nrows x ncols; every entry keeps its coordinate and value, entries at or
 * past the new dims are dropped (grown fixture: matrix.rs:1617-1672). */
fgpu_info fgpu_mat_resize(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t nrows,
                          uint64_t ncols);
/* Pattern intersection (K7): eWiseMult ANY_PAIR (matrix.rs:876-896); values from `b`. */
fgpu_info fgpu_mat_intersect(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, const fgpu_mat* b);
/* intersection_nvals (matrix.rs:743-761). */
fgpu_info fgpu_mat_intersect_nvals(fgpu_ctx* ctx, const fgpu_mat* a, const fgpu_mat* b,
                                   uint64_t* out);

/* Acceleration index for the dense-frontier vxm (tiled.hip): regroup the entries of `m` by column
 * tile of 2^tile_bits ids so the kernel can stage the frontier tile in LDS and stream packed 4-byte
 * entries from HBM.  tile_bits / vec / k == 0 pick defaults (2^20-id tiles = 128 KiB of LDS).
 * Logically const: the matrix content is unchanged; the index is owned and freed by the matrix.
 * No reference counterpart (GraphBLAS keeps its own internal formats, matrix.rs:405-426). */
fgpu_info fgpu_mat_build_tiles(fgpu_ctx* ctx, fgpu_mat* m, int tile_bits, int vec, int k);
/* info[0]=tile_bits [1]=tiles [2]=64-row groups [3]=items [4]=padded entries [5]=vec [6]=k
 * [7]=device bytes of the index.  FGPU_NO_VALUE when the matrix has no tiles. */
fgpu_info fgpu_mat_tiles_info(const fgpu_mat* m, uint64_t info[8]);

/* ---- products (ANY_PAIR structural semiring) ------------------------------ */

/* C = F x B, no mask: Matrix::lmxm -> GrB_mxm(GxB_ANY_PAIR_BOOL) (matrix.rs:930-947).
 * The result is a new snapshot (the reference overwrites F in place; the caller
 * frees the old F). */
fgpu_info fgpu_mxm(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* b);

/* C = (F x (m U dp)) with every (i,j) of (F x dm) removed from the F x m part —
 * the exact algebra of Matrix::delta_lmxm (matrix.rs:1317-1402): dp results are
 * NOT masked; the mask is row-level (any source of row i tombstoning j kills
 * (i,j)).  dp / dm may be NULL or empty => plain lmxm. */
fgpu_info fgpu_delta_lmxm(fgpu_ctx* ctx, fgpu_mat** c, const fgpu_mat* f, const fgpu_mat* m,
                          const fgpu_mat* dp, const fgpu_mat* dm);

/* The device core of CondTraverseOp::expand_batch (cond_traverse.rs:452-751):
 *   F[i, src_ids[i]] = 1 for i in [0,nsrc)  (src_ids[i] == UINT64_MAX => row i left empty:
 *   a source that failed its label pre-filter, cond_traverse.rs:561-589);
 *   F <- delta_lmxm(F; m[h], dp[h], dm[h]) for h in [0,nhops);
 *   entries whose dest lacks a bit in `dst_label_bitmap` (nullable, ncols bits,
 *   LSB-first in 64-bit words) are dropped (cond_traverse.rs:644-651);
 *   output CSR over the nsrc rows, dest ascending & unique per row
 *   (= the ascending (row_i, dest) stream of F.iter, cond_traverse.rs:644).
 * dp[h] / dm[h] may be NULL.  flops (nullable) receives sum over hops of
 * sum_{(i,s) in F_h} deg_{m U dp}(s) — the traversed-edge count used for TEPS. */
fgpu_info fgpu_expand(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                      const fgpu_mat* const* m, const fgpu_mat* const* dp,
                      const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                      uint64_t** out_rowptr, uint64_t** out_dest, uint64_t* out_nnz,
                      uint64_t* flops);

/* fgpu_expand with the result in the device's own 32-bit form: out_rowptr[nsrc + 1] and out_dest[nnz] are uint32_t (the
 * engine's row pointers and node ids are 32-bit throughout; a result of 2^32 entries or more is refused by every emitting
 * entry) — two DMAs of the arrays as they lie, nothing widened on the device or the host, half the PCIe bytes of fgpu_expand.
 * Release both with fgpu_free. */
fgpu_info fgpu_expand32(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                      const fgpu_mat* const* m, const fgpu_mat* const* dp,
                      const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                      uint32_t** out_rowptr, uint32_t** out_dest, uint64_t* out_nnz,
                      uint64_t* flops);

/* Same chain, the result F left on the device as a matrix handle — what the reference holds between
 * `F = delta_lmxm(...)` and `F.iter(...)` (cond_traverse.rs:602-608: F IS a Matrix<bool>; its rows are then walked
 * with the row iterator, :644).  *out is a BOOL snapshot with nsrc rows (row i = source i), dest ascending and unique
 * per row, already filtered by `dst_label_bitmap`; read it with fgpu_mat_extract / fgpu_mat_export_csr, release it with
 * fgpu_mat_free.  fgpu_expand = this + one export of the whole result into host arrays. */
fgpu_info fgpu_expand_mat(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                          const fgpu_mat* const* m, const fgpu_mat* const* dp,
                          const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                          fgpu_mat** out, uint64_t* flops);

/* The same chain with the result as the two COLUMNS CondTraverseOp::expand_batch hands on (cond_traverse.rs:644-751: walk
 * (row_i, dest) ascending; drop a destination that differs from a pre-bound `to`, :657-661; emit `gather(out_indices)` +
 * NodeIds): out_row[q] = index of the source row of pair q (uint16_t or uint32_t per `row_bits`; a child batch holds at most
 * 1024 rows, batch.rs:81), out_dest[q] = its destination, pairs ascending by (row, dest).  `pinned_dest` (nullable, nsrc
 * entries): ~0 = the row keeps every destination, anything else = the row keeps only that destination if it is reached.  Both
 * columns are built on the device (row indices expanded from the row pointers, the pinned rows cut down by a binary search)
 * and arrive by DMA in pinned blocks of the context's pool — the host never walks the result.  Release both with fgpu_free;
 * *out_n = 0 leaves both NULL. */
fgpu_info fgpu_expand_pairs(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                            const fgpu_mat* const* m, const fgpu_mat* const* dp,
                            const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                            const uint64_t* pinned_dest, int row_bits, void** out_row, uint64_t** out_dest,
                            uint64_t* out_n, uint64_t* flops);

/* The same two columns with the destinations as the 32-bit node ids the device holds (node ids of a traversed graph fit 32
 * bits: Tensor's multi-edge keys demand it, tensor.rs:154-163): half the bytes over PCIe for the column that IS the result —
 * without pinned rows it is the chain's own column-id array, copied out as it lies — and what CondTraverseOp's C++ twin takes
 * (falkordb_amd/host/graph.cpp: ExpandedRows widens on access, where the reference's NodeId(u64) is needed).  Otherwise
 * identical to fgpu_expand_pairs. */
fgpu_info fgpu_expand_pairs32(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                            const fgpu_mat* const* m, const fgpu_mat* const* dp,
                            const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                            const uint64_t* pinned_dest, int row_bits, void** out_row, uint32_t** out_dest,
                            uint64_t* out_n, uint64_t* flops);

/* The same chain when EVERY row has a pre-bound destination — CondTraverse with `to` bound on the whole batch, the multi-hop
 * ExpandInto shape `MATCH (a)-[*]->(b) ... MATCH (a)-->()-->(b)` (tests/flow/test_expand_into.py:63-95; the reference runs
 * the whole chain and drops every destination but the bound one, cond_traverse.rs:657-661): present[i] = 1 iff dst_ids[i]
 * is reached from src_ids[i] by the chain (and passes the destination label).  The hops before the last run as in
 * fgpu_expand; the last one is ONE entry of (F·m)<not (F·dm)> U (F·dp) per row — in bit form one bit of one row of the
 * state, found by walking the in-neighbours of dst_ids[i]; in sorted-CSR form a binary search per frontier entry — so no
 * result is materialised, emitted or copied back.  present[nsrc] is a HOST array; *flops (nullable) counts the traversed
 * edges of the hops that ran (the last hop is not expanded). */
fgpu_info fgpu_expand_probe(fgpu_ctx* ctx, const uint64_t* src_ids, const uint64_t* dst_ids, uint64_t nsrc,
                            const fgpu_mat* const* m, const fgpu_mat* const* dp, const fgpu_mat* const* dm, int nhops,
                            const uint64_t* dst_label_bitmap, uint8_t* present, uint64_t* flops);

/* The same chain with the result STREAMED to the host in chunks of whole source rows — the shape in which
 * CondTraverseOp::expand_batch consumes it (cond_traverse.rs:644-751: walk (row_i, dest) ascending, emit an output batch
 * every 1024 pairs): the chain runs once, F stays on the device, and chunks of at most `chunk_rows` consecutive source
 * rows are copied into a ring of four pinned buffers, three copies on the link while the caller walks the fourth.
 *   open : runs the chain; *nnz / *flops (nullable) as fgpu_expand.  dest_bits = 64 (GrB_Index ids, widened on the device)
 *          or 32 (half the bytes on the link; node ids fit 32 bits, tensor.rs:154-163).
 *   next : blocks until the next chunk has landed; *first_row / *nrows name its source rows, rowptr[0 .. nrows] are entry
 *          offsets relative to the chunk (rowptr[0] = 0), dest[0 .. rowptr[nrows]) the destinations (uint64_t or
 *          uint32_t per dest_bits), ascending and unique per row.  The arrays stay valid until the next call on the
 *          stream.  Returns FGPU_NO_VALUE (and *nrows = 0) after the last chunk.
 *   close: releases the device result and the ring (valid at any point).
 * A stream belongs to the thread that opened it. */
typedef struct fgpu_expand_stream fgpu_expand_stream;
fgpu_info fgpu_expand_stream_open(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                                  const fgpu_mat* const* m, const fgpu_mat* const* dp,
                                  const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                                  uint64_t chunk_rows, int dest_bits, fgpu_expand_stream** out,
                                  uint64_t* nnz, uint64_t* flops);
fgpu_info fgpu_expand_stream_next(fgpu_expand_stream* s, uint64_t* first_row, uint64_t* nrows,
                                  const uint64_t** rowptr, const void** dest);
fgpu_info fgpu_expand_stream_close(fgpu_expand_stream* s);

/* Same as fgpu_expand but the result stays on device and only its size and an
 * order-independent checksum come back (full-size configs whose output would not
 * fit a host buffer): checksum = sum over the (row, dest) entries of
 * mix64(row) * (mix64(dest ^ 0x9e3779b97f4a7c15) | 1) mod 2^64, mix64 = the splitmix64 finaliser (a product of a row
 * hash and a destination hash: the bit-parallel state sums it per vertex through look-up tables instead of per entry). */
fgpu_info fgpu_expand_count(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                            const fgpu_mat* const* m, const fgpu_mat* const* dp,
                            const fgpu_mat* const* dm, int nhops,
                            const uint64_t* dst_label_bitmap, uint64_t* out_nnz,
                            uint64_t* checksum, uint64_t* flops);

/* Variable-length reachability core (BASELINE config 5).  The reference evaluates
 * `[*1..k]` with a per-row DFS (CondVarLenTraverseOp, cond_var_len_traverse.rs:81-128); its DISTINCT
 * end-point set and the per-hop frontiers are set algebra over the same layers: ONE chain of `nhops` hops
 * (each hop a delta_lmxm, quirk included) evaluated once in bit form (one bit per source row),
 *   hop_nnz[h] / hop_checksum[h] : size / checksum of F x A_1 .. A_{h+1}  (walks of exactly h+1 hops),
 *   union_nnz / union_checksum   : the DISTINCT (row, dest) pairs reached by 1..nhops hops
 *                                  (= MATCH (a)-[*1..k]->(b) RETURN DISTINCT a, b; also the pruning set for the DFS).
 * hop_checksum / union_* / flops are nullable; checksum as in fgpu_expand_count; the destination-label
 * bitmap (nullable) filters every reported set.  Base matrices must be non-hypersparse with nnz < 2^31. */
fgpu_info fgpu_expand_levels(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                             const fgpu_mat* const* m, const fgpu_mat* const* dp,
                             const fgpu_mat* const* dm, int nhops, const uint64_t* dst_label_bitmap,
                             uint64_t* hop_nnz, uint64_t* hop_checksum, uint64_t* union_nnz,
                             uint64_t* union_checksum, uint64_t* flops);

/* Exact trail counts for variable-length patterns of at most two hops.  The reference's `[*1..k]`
 * DFS emits one row per TRAIL (edge-unique path, cond_var_len_traverse.rs:196-387); for exactly `nhops` (1 or 2) hops
 * the number of trails from source i to a destination is a counting product over the effective layers (m \ dm) U dp:
 * PLUS_PAIR on pattern layers, PLUS_TIMES on UINT64 layers holding per-pair edge multiplicities (`weighted`), less the
 * length-2 walks that cross one self-loop twice.  Output CSR over the nsrc rows: dest ascending, count > 0.  Longer
 * trails have no product form (a walk may revisit an edge): use fgpu_expand_levels' reachable sets to prune the DFS. */
fgpu_info fgpu_expand_trail_counts(fgpu_ctx* ctx, const uint64_t* src_ids, uint64_t nsrc,
                                   const fgpu_mat* const* m, const fgpu_mat* const* dp,
                                   const fgpu_mat* const* dm, int nhops, int weighted,
                                   uint64_t** out_rowptr, uint64_t** out_dest, uint64_t** out_count,
                                   uint64_t* out_nnz);

/* ---- boolean vxm / BFS (K9) ---------------------------------------------- */

/* w<!mask, replace> = f x A over the boolean (ANY_PAIR) semiring, vectors as
 * ncols-/nrows-bit bitmaps (LSB-first 64-bit words, HOST pointers; copied in/out).
 * This is GrB_vxm (graphblas/mod.rs:11173) in the form LAGraph's BFS issues it.
 * mask may be NULL.  `At` (nullable) enables the pull direction; `direction`:
 * 0 = auto (push over A while fewer than 1 / 32 of the vertices are in the frontier, else — given At — the LDS-tiled
 * pull), 1 = push over A, 2 = pull over At's CSR (full pass, no early exit),
 * 3 = pull over At's LDS-tile layout (built on first use, fgpu_mat_build_tiles). */
fgpu_info fgpu_vxm(fgpu_ctx* ctx, uint64_t* w, const uint64_t* f, const uint64_t* mask,
                   const fgpu_mat* A, const fgpu_mat* At, int direction);

/* PageRank: replaces LAGr_PageRank(&centrality, &iters, G, damping, tol, itermax, msg) as called by
 * algo.pageRank (algo_procedures.rs:741-752 with 0.85 / 1e-4 / 100; binding lagraph_bindings.rs:549-558) on the
 * adjacency matrix A (At nullable: transposed internally).  FP32 like the reference's GrB_FP32 vectors; sinks
 * (vertices without out-edges) spread their score evenly.  `active_bitmap` (nullable, nrows bits) restricts the run
 * to the induced subgraph of the flagged vertices — the label-filtered form (algo_procedures.rs:725-733) — and
 * leaves 0 in the other slots.  centrality[nrows] is a HOST array; *iters (nullable) the iterations taken. */
fgpu_info fgpu_pagerank(fgpu_ctx* ctx, const fgpu_mat* A, const fgpu_mat* At, const uint64_t* active_bitmap,
                        float damping, float tol, int32_t itermax, float* centrality, int32_t* iters);
/* The same run; *converged (nullable) = 1 when the last iteration changed the scores by no more than tol, 0 when itermax
 * ended it first — what LAGr_PageRank turns into LAGRAPH_CONVERGENCE_FAILURE (lagraph_bindings.rs:549-558). */
fgpu_info fgpu_pagerank_status(fgpu_ctx* ctx, const fgpu_mat* A, const fgpu_mat* At, const uint64_t* active_bitmap,
                               float damping, float tol, int32_t itermax, float* centrality, int32_t* iters,
                               int32_t* converged);

/* Level-synchronous BFS: replaces LAGr_BreadthFirstSearch_Extended(level, parent, G,
 * src, max_level, -1, false) as called by algo.BFS (algo_procedures.rs:1079-1088;
 * binding lagraphx_bindings.rs:585-594).  level[n] (int32, -1 = unreached, source = 0),
 * parent[n] (nullable, int64, -1 = none, parent[src] = src) are HOST arrays (pinned ones — fgpu_host_alloc — are
 * filled by DMA: 0.3 ms for the 16 MiB level[] of RMAT-22 instead of 0.9 ms through staging into pageable memory).
 * max_level < 0 => unlimited.  At may be NULL (push only).  edges_traversed
 * (nullable) = sum of out-degrees of reached vertices (TEPS numerator).
 * The search plan of the last call over the same (A, At) pair stays attached to A (released with either matrix, rebuilt
 * after fgpu_set_option), so repeated calls pay the search and the 4 N-byte copy-out, not plan creation. */
fgpu_info fgpu_bfs(fgpu_ctx* ctx, const fgpu_mat* A, const fgpu_mat* At, uint64_t src,
                   int64_t max_level, int32_t* level, int64_t* parent,
                   uint64_t* edges_traversed);

/* Plan API: keeps the BFS workspace resident so repeated searches (bench steps,
 * many roots) pay no allocation, and results can stay on device.
 * Column-slab partition for multi-GPU: rank `rank` of `nranks`
 * owns destination vertices [rank*slab, (rank+1)*slab), slab = ceil(n/nranks)
 * rounded up to 4096; A / At passed here are the rank's slab (A restricted to the
 * owned columns, At restricted to the owned rows, both with GLOBAL ids).
 * nranks == 1 is the plain single-GPU case. */
fgpu_info fgpu_bfs_plan_create(fgpu_ctx* ctx, fgpu_bfs_plan** plan, const fgpu_mat* A,
                               const fgpu_mat* At, int rank, int nranks);
fgpu_info fgpu_bfs_plan_free(fgpu_bfs_plan* plan);
/* Tunables: alpha/beta of the push<->pull switch (Beamer), 0 keeps defaults;
 * force_direction 0 auto / 1 push only / 2 pull only. */
fgpu_info fgpu_bfs_plan_tune(fgpu_bfs_plan* plan, double alpha, double beta, int force_direction);
/* Run one whole BFS on a single-rank plan; results stay on device. */
fgpu_info fgpu_bfs_run(fgpu_bfs_plan* plan, uint64_t src, int64_t max_level, int want_parent);
/* The same search without the final host wait: clears the workspace, seeds `src` and enqueues
 * `levels` (<= 0: 10) level kernels on the ctx stream; kernels past the last level are no-ops.
 * fgpu_bfs_wait() returns once the search has ended (it enqueues more levels if `levels` was too
 * few).  Two plans over the same matrices let the host enqueue search i+1 while search i runs. */
fgpu_info fgpu_bfs_run_async(fgpu_bfs_plan* plan, uint64_t src, int64_t max_level, int want_parent,
                             int levels);
fgpu_info fgpu_bfs_wait(fgpu_bfs_plan* plan);
/* Copy results of the last run to host (either pointer may be NULL). */
fgpu_info fgpu_bfs_fetch(fgpu_bfs_plan* plan, int32_t* level, int64_t* parent);
/* Stats of the last run: stats[0]=levels, [1]=reached vertices (incl. source),
 * [2]=edges traversed (sum outdeg of reached), [3]=push levels, [4]=pull levels,
 * [5]=edges scanned by push kernels, [6]=edges scanned by pull kernels. */
fgpu_info fgpu_bfs_stats(fgpu_bfs_plan* plan, uint64_t stats[8]);

/* Multi-rank stepping (driven by the host loop that owns the collective):
 *   begin(src) ; repeat { step() ; <allgather slab words> ; commit() } until done.
 * local_words / global_words are DEVICE pointers owned by the plan:
 *   local  = this rank's new-frontier slab: slab/64 bitmap words + 2 stat words
 *            (count, sum of out-degrees) => `words_per_rank` uint64 each;
 *   global = nranks * words_per_rank, the allgather destination. */
fgpu_info fgpu_bfs_part_buffers(fgpu_bfs_plan* plan, void** local_words, void** global_words,
                                uint64_t* words_per_rank);
/* Let the caller own the exchange buffers instead (e.g. torch tensors handed to
 * torch.distributed.all_gather_into_tensor): local = words_per_rank uint64, global =
 * nranks * words_per_rank uint64, both DEVICE memory that outlives the plan's use.  Both are
 * zeroed by the call. */
fgpu_info fgpu_bfs_part_set_buffers(fgpu_bfs_plan* plan, void* local_words, void* global_words);
fgpu_info fgpu_bfs_part_begin(fgpu_bfs_plan* plan, uint64_t src, int64_t max_level);
fgpu_info fgpu_bfs_part_step(fgpu_bfs_plan* plan);
fgpu_info fgpu_bfs_part_commit(fgpu_bfs_plan* plan);
/* Non-blocking-ish poll of the device control block (one small D2H): done != 0 when
 * the last committed frontier was empty or max_level was reached. */
fgpu_info fgpu_bfs_part_done(fgpu_bfs_plan* plan, int32_t* done, int32_t* level);

/* Fused slab path (multi-rank v2): ONE level kernel and ONE frontier all-gather per level — no commit pass,
 * no control kernel.  The rank owns destinations [lo, hi) exactly as above; because a column slab only ever
 * discovers vertices it owns, the level kernel updates visited / level / parent at discovery and writes its
 * owned words of the next frontier into a send buffer; the host all-gathers that buffer into `global_words`
 * (nranks * words_per_rank uint64 = the global frontier bitmap) and launches the next level.  Everything a
 * level needs from other ranks is that bitmap: termination comes from its population (identical on all
 * ranks), the push / pull choice is each rank's own.
 *   set_buffers : send0 / send1 (words_per_rank uint64 each, double-buffered) and global_words, caller-owned
 *                 DEVICE memory (torch tensors handed to all_gather_into_tensor);
 *   set_degrees : optional global out-degree of every vertex (uint32[n], device) so that the owner accounts a
 *                 vertex's whole out-degree in edges_traversed (a column slab holds only its share);
 *   begin       : clears the workspace and seeds `src` on every rank (no collective for level 0);
 *   level       : enqueues one level; *send_index (0/1) = the send buffer to all-gather next;
 *   fgpu_bfs_part_done / _stats / _fetch work as for the stepped path (reached / edges_traversed are the
 *   rank's own share: sum them over ranks). */
fgpu_info fgpu_bfs_slab_set_buffers(fgpu_bfs_plan* plan, void* send0, void* send1, void* global_words);
fgpu_info fgpu_bfs_slab_set_degrees(fgpu_bfs_plan* plan, const uint32_t* global_out_degrees);
fgpu_info fgpu_bfs_slab_begin(fgpu_bfs_plan* plan, uint64_t src, int64_t max_level, int want_parent);
fgpu_info fgpu_bfs_slab_level(fgpu_bfs_plan* plan, int* send_index);
/* deg[r] = stored entries of row r (DEVICE uint32[nrows]); summed over the ranks' column slabs it is the global
 * out-degree vector fgpu_bfs_slab_set_degrees takes. */
fgpu_info fgpu_mat_row_degrees(fgpu_ctx* ctx, const fgpu_mat* a, uint32_t* out_dev);

/* Build this rank's column slab of a full matrix: out = A[:, lo:hi) (global ids
 * kept), and its transpose restricted to rows [lo,hi).  Used to shard a replicated
 * or host-loaded adjacency. */
fgpu_info fgpu_mat_col_slab(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo,
                            uint64_t hi);
fgpu_info fgpu_mat_row_slab(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t lo,
                            uint64_t hi);

/* ---- multi-GPU: RCCL communicator + in-library partitioned BFS -------------------------
 * The reference reaches its BFS through one call (LAGr_BreadthFirstSearch_Extended, algo_procedures.rs:1079-1088);
 * a process that links this library reaches the multi-GPU form the same way: the communicator, the level loop and
 * the frontier exchange (an all-gather-v: grouped ncclSend / ncclRecv to every peer, one xGMI link per piece) are
 * inside libfgpu.so.
 *   one process per GPU : rank 0 calls fgpu_comm_unique_id, the launcher's own channel carries the 128 bytes,
 *                         every rank calls fgpu_comm_init_rank on its context;
 *   one process, N GPUs : one context per device, fgpu_comm_init_all (ncclCommInitAll) — the Redis-module form. */
#define FGPU_COMM_ID_BYTES 128
fgpu_info fgpu_comm_unique_id(uint8_t* id /* FGPU_COMM_ID_BYTES */);
fgpu_info fgpu_comm_init_rank(fgpu_ctx* ctx, int nranks, int rank, const uint8_t* id);
fgpu_info fgpu_comm_init_all(fgpu_ctx* const* ctxs, int n);
fgpu_info fgpu_comm_finalize(fgpu_ctx* ctx);
fgpu_info fgpu_comm_info(fgpu_ctx* ctx, int32_t* rank, int32_t* nranks);
/* nnz-balanced slab boundaries of the destination vertices (= columns of A): splits[0] = 0, splits[nparts] = the
 * vertex count rounded up to 4096, every boundary a multiple of 4096, part k holding ~ nnz / nparts of A's entries
 * (prefix sum of in-degrees; R-MAT skew otherwise overloads the parts that own the hubs). */
fgpu_info fgpu_mat_balanced_splits(fgpu_ctx* ctx, const fgpu_mat* a, int nparts, uint64_t* splits /* nparts + 1 */);
/* The partition arithmetic on its own — pure host functions (no device, no context), so that a launcher, the host layer and
 * the CPU tests all use the ONE definition the level loop uses:
 *   fgpu_splits_shift             log2 of the column-block width fgpu_mat_balanced_splits histograms with (>= 12);
 *   fgpu_balanced_splits_from_hist the boundary choice from per-block entry counts (the host half of the call above);
 *   fgpu_slab_layout              rank r's vertex range [lo, hi) and its words (offset, count) in the global frontier
 *                                 bitmap = what the per-level all-gather-v moves; splits == NULL: the equal slabs of
 *                                 fgpu_bfs_plan_create (ceil(n / nranks) rounded up to 4096).  Outputs are nullable. */
uint32_t fgpu_splits_shift(uint64_t ncols);
fgpu_info fgpu_balanced_splits_from_hist(const uint64_t* block_counts, uint64_t nblocks, uint32_t shift, uint64_t ncols,
                                         int nparts, uint64_t* splits /* nparts + 1 */);
fgpu_info fgpu_slab_layout(const uint64_t* splits /* nullable: nranks + 1 */, uint64_t n, int nranks, uint64_t* lo,
                           uint64_t* hi, uint64_t* word_off, uint64_t* word_cnt);
/* fgpu_bfs_plan_create with caller-chosen slab boundaries: rank owns destinations [splits[rank], splits[rank+1]);
 * A_slab = A[:, slab], At_slab = A'[slab, :] (fgpu_mat_col_slab + fgpu_mat_transpose), global ids. */
fgpu_info fgpu_bfs_plan_create_slab(fgpu_ctx* ctx, fgpu_bfs_plan** plan, const fgpu_mat* A_slab,
                                    const fgpu_mat* At_slab, int rank, int nranks, const uint64_t* splits);
/* One whole search over the partition: per level one kernel per rank and one frontier exchange.  `plans` = this
 * process' ranks: ONE plan when every GPU has its own process (exchange over the context's communicator), ALL plans
 * in rank order when one process drives the node (RCCL if the contexts were joined by fgpu_comm_init_all, event-ordered
 * peer copies otherwise).  Results per rank through fgpu_bfs_fetch (the owned range) / fgpu_bfs_stats (the rank's
 * share of reached / edges_traversed: sum over ranks). */
fgpu_info fgpu_bfs_dist_run(fgpu_bfs_plan* const* plans, int nplans, uint64_t src, int64_t max_level,
                            int want_parent);
/* Time split of the plan's last fgpu_bfs_dist_run: HIP-event sums over its level kernels and over its exchanges
 * (an exchange includes the wait for the slowest rank) — zero unless the "dist_timing" option was on during the run —
 * and the number of level launches. */
fgpu_info fgpu_bfs_dist_times(fgpu_bfs_plan* plan, double* level_ms, double* collective_ms, uint64_t* launches);

/* ---- measurement hooks (bench.py; not reference APIs) --------------------- */

/* Time `iters` launches of one named kernel with HIP events on the ctx stream.
 * which: 0 = full-pass boolean pull SpMV over the CSR of the matrix passed (dense frontier, no
 * mask, no early exit), 1 = push over A with a dense frontier, 2 = the same full pass as 0 over
 * the LDS-tile layout (the "RMAT-22 boolean SpMV" roofline case), 3 = as 2 with the caches flushed before every timed
 * launch (a 512 MiB scratch buffer is read between launches: the 256 MiB Infinity Cache holds none of the
 * layout when the pass starts, and no write-back is pending).  Returns avg ms per launch and algorithmic bytes per launch
 * (SURVEY.md §8d formulas). */
fgpu_info fgpu_bench_spmv(fgpu_ctx* ctx, const fgpu_mat* A, int which, int iters,
                          double* avg_ms, uint64_t* alg_bytes);
/* Context-wide kernel profiler for the non-BFS paths (k-hop products, merges, transposes): while enabled, every
 * launch of a modelled kernel is bracketed by HIP events on its lane's stream.  fgpu_prof_read synchronises,
 * folds the records by kernel name (static strings), returns total ms / launch count / algorithmic bytes
 * (SURVEY.md §8d accounting; 0 = not modelled) per name and clears them.  Enabling clears earlier records. */
fgpu_info fgpu_prof_enable(fgpu_ctx* ctx, int enable);
fgpu_info fgpu_prof_read(fgpu_ctx* ctx, const char** names, double* ms, uint64_t* launches,
                         uint64_t* alg_bytes, int cap, int* n);
/* Uniform sample of the stored entries of `a` (bench / test data for "0.1 % random tombstones"):
 * (r, c) is kept iff mix64(seed ^ mix64(r << 32 | c)) % denom == 0, mix64 = the splitmix64 finaliser — a function
 * of the coordinate, so a CPU model draws the same sample.  Pattern only.  Not a reference API. */
fgpu_info fgpu_mat_sample(fgpu_ctx* ctx, fgpu_mat** out, const fgpu_mat* a, uint64_t seed, uint32_t denom);
/* Per-kernel accumulated HIP-event timings of a plan (enabled by
 * fgpu_bfs_plan_profile(plan,1)): names[i] (static strings), ms[i], launches[i],
 * alg_bytes[i]; returns count in *n (<= cap). */
fgpu_info fgpu_bfs_plan_profile(fgpu_bfs_plan* plan, int enable);
fgpu_info fgpu_bfs_plan_profile_read(fgpu_bfs_plan* plan, const char** names, double* ms,
                                     uint64_t* launches, uint64_t* alg_bytes, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* FGPU_H */
