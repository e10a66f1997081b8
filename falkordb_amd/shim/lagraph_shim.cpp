// lagraph_shim.cpp — the LAGraph-named part of the tier-2 boundary: `liblagraph.so` / `liblagraphx.so`,
// exporting the LAGraph entry points the reference's BFS / PageRank procedures bind, on the MI355X engine.  With them next
// to libgraphblas.so (graphblas_shim.cpp) the reference's UNMODIFIED call sequences run on the GPU:
//   algo.BFS       algo_procedures.rs:1060-1165  LAGraph_New (borrowed adjacency, :389-405) -> LAGr_BreadthFirstSearch_Extended
//                  (lagraphx_bindings.rs:585-594; level, parent|NULL, src, max_level, -1, false) -> GrB_Vector_nvals +
//                  GrB_Vector_extractTuples_INT64 on level / parent (:431-447) -> GrB_Vector_free -> G->A = NULL; LAGraph_Delete
//   algo.pageRank  algo_procedures.rs:734-760    LAGraph_New -> LAGraph_Cached_AT + LAGraph_Cached_OutDegree ->
//                  LAGr_PageRank(0.85, 1e-4, 100) (lagraph_bindings.rs:549-558) -> GrB_Vector_extractTuples_FP64 (:415-429)
//   matrix::init / shutdown  matrix.rs:174-183, 215-221  LAGraph_Init after GxB_init, LAGraph_Finalize
// LAGraph itself is an un-vendored dependency (build.rs:50-52 links prebuilt static archives); what is restated here is its
// published contract as the bindings' own doc comments state it (argument meaning, cached-property rules, return codes:
// lagraph_bindings.rs:23-31) — the algorithms are the engine's fgpu_bfs / fgpu_pagerank, pinned against the oracle.
// The seven other LAGraph algorithms algo_procedures.rs calls (WCC, betweenness, harmonic centrality, max-flow, CDLP, MSF
// and the EMin property) are outside this engine's path (out of scope): they are exported so the file links,
// and return GrB_NOT_IMPLEMENTED with a message instead of computing anything.
//
// One source, two libraries: -DFG_LAGRAPHX builds the LAGraphX (experimental) symbols, without it the LAGraph core ones.
#include "shim_internal.hpp"

extern "C" {
GrB_Info GrB_init(int mode);
GrB_Info GrB_finalize();
GrB_Info GrB_Matrix_free(GrB_Matrix* A);
GrB_Info GrB_Vector_free(GrB_Vector* v);
GrB_Info GrB_Scalar_free(GrB_Scalar* s);
}

// LAGraph_Graph_struct, field for field as bindgen lays it out (lagraph_bindings.rs:108-129; 88 bytes — the caller writes
// G->A itself, algo_procedures.rs:409-413)
struct LAGraph_Graph_struct {
    GrB_Matrix A;
    int32_t kind;                     // LAGraph_Kind: 0 undirected, 1 directed, -1 unknown (lagraph_bindings.rs:77-84)
    GrB_Matrix AT;
    GrB_Vector out_degree, in_degree;
    int32_t is_symmetric_structure;   // LAGraph_Boolean: 0 / 1 / -1 unknown
    int64_t nself_edges;
    GrB_Scalar emin;
    int32_t emin_state;
    GrB_Scalar emax;
    int32_t emax_state;
};
static_assert(sizeof(LAGraph_Graph_struct) == 88, "LAGraph_Graph_struct must match the bindgen layout (lagraph_bindings.rs:132)");
typedef LAGraph_Graph_struct* LAGraph_Graph;

enum { LAGRAPH_INVALID_GRAPH = -1000, LAGRAPH_NOT_CACHED = -1003, LAGRAPH_CONVERGENCE_FAILURE = -1005, LAGRAPH_CACHE_NOT_NEEDED = 1000,
       LAGRAPH_UNKNOWN = -1, LAGRAPH_MSG_LEN = 256 };

namespace {
void clear_msg(char* msg) { if (msg) msg[0] = 0; }
int fail(char* msg, int code, const char* what) {
    if (msg) snprintf(msg, LAGRAPH_MSG_LEN, "%s", what);
    return code;
}
template <typename F>
int guarded(char* msg, F&& f) {
    try {
        return f();
    } catch (const falkor::GrbError& e) {
        if (msg) snprintf(msg, LAGRAPH_MSG_LEN, "%s", e.what());
        switch (e.info) {
            case FGPU_OOM: return GrB_OUT_OF_MEMORY;
            case FGPU_OUT_OF_BOUNDS: return GrB_INDEX_OUT_OF_BOUNDS;
            case FGPU_DIM_MISMATCH: return GrB_DIMENSION_MISMATCH;
            case FGPU_NULL_POINTER: return GrB_NULL_POINTER;
            case FGPU_INVALID: return GrB_INVALID_VALUE;
            default: return GrB_PANIC;
        }
    } catch (const std::bad_alloc&) {
        return fail(msg, GrB_OUT_OF_MEMORY, "out of memory");
    } catch (...) {
        return fail(msg, GrB_PANIC, "unexpected exception");
    }
}
// LAGraph_CheckGraph's O(1) rules (lagraph_bindings.rs:254): A present and square, a recognised kind, cached AT of the
// transposed shape, degree vectors of the matching length
int check_graph(LAGraph_Graph G, char* msg) {
    if (!G) return fail(msg, GrB_NULL_POINTER, "graph is NULL");
    if (!G->A) return fail(msg, LAGRAPH_INVALID_GRAPH, "graph adjacency matrix is NULL");
    if (G->kind != 0 && G->kind != 1) return fail(msg, LAGRAPH_INVALID_GRAPH, "graph kind invalid");
    if (G->A->m.nrows() != G->A->m.ncols()) return fail(msg, LAGRAPH_INVALID_GRAPH, "adjacency matrix must be square");
    if (G->AT && (G->AT->m.nrows() != G->A->m.ncols() || G->AT->m.ncols() != G->A->m.nrows()))
        return fail(msg, LAGRAPH_INVALID_GRAPH, "G->AT has the wrong dimensions");
    if (G->out_degree && G->out_degree->n != G->A->m.nrows()) return fail(msg, LAGRAPH_INVALID_GRAPH, "out_degree has the wrong size");
    if (G->in_degree && G->in_degree->n != G->A->m.ncols()) return fail(msg, LAGRAPH_INVALID_GRAPH, "in_degree has the wrong size");
    return GrB_SUCCESS;
}
void check(fgpu_info i, const char* where) { falkor::check(i, where); }

#ifndef FG_LAGRAPHX
// degree(i) = entries of A(i,:) as a GrB_INT64 vector that stores only the non-zero degrees (lagraph_bindings.rs:115-116)
GrB_Vector degrees_of(const Matrix& m) {
    falkor::Context* c = fgshim::context();
    const uint64_t n = m.nrows();
    int64_t* out = nullptr;
    check(fgpu_host_alloc(c->raw(), (n ? n : 1) * sizeof(int64_t), (void**)&out), "LAGraph_Cached_OutDegree");
    if (n) {
        // the engine writes 32-bit degrees; a pinned block is device-visible, so the kernel fills it directly
        uint32_t* d32 = nullptr;
        fgpu_info r = fgpu_host_alloc(c->raw(), n * sizeof(uint32_t), (void**)&d32);
        if (r == FGPU_OK) r = fgpu_mat_row_degrees(c->raw(), m.snapshot(), d32);
        if (r == FGPU_OK)
            for (uint64_t i = 0; i < n; ++i) out[i] = d32[i];
        if (d32) (void)fgpu_free(c->raw(), d32);
        if (r != FGPU_OK) { (void)fgpu_free(c->raw(), out); check(r, "LAGraph_Cached_OutDegree"); }
    }
    return fgshim::vector_over_pinned(fgshim::type_int64(), n, out, 2);
}
#endif
}  // namespace

extern "C" {

#ifndef FG_LAGRAPHX
// ---- LAGraph core -------------------------------------------------------------------------------------------------------
// matrix.rs:174-183: called after GxB_init; the reference's LAGraph accepts an initialised GraphBLAS, and brings it up
// itself when it is not (LAGraph's own programs call only LAGraph_Init)
int LAGraph_Init(char* msg) {
    clear_msg(msg);
    if (fgshim::context()) return GrB_SUCCESS;
    const GrB_Info r = GrB_init(0 /* GrB_NONBLOCKING */);
    return r == GrB_SUCCESS ? r : fail(msg, r, "GrB_init failed: no HIP device (this library has no CPU path)");
}
int LAGraph_Finalize(char* msg) {      // matrix.rs:215-221: the only shutdown call — GraphBLAS goes down with it
    clear_msg(msg);
    return GrB_finalize();
}
int LAGraph_Version(int* version_number, char* version_date, char* msg) {
    clear_msg(msg);
    if (!version_number || !version_date) return GrB_NULL_POINTER;
    version_number[0] = 1; version_number[1] = 2; version_number[2] = 1;    // lagraph_bindings.rs:16-19
    strcpy(version_date, "Sept 8, 2025");
    return GrB_SUCCESS;
}
// { G->A = *A; *A = NULL; } — cached properties NULL / unknown (lagraph_bindings.rs:175-181)
int LAGraph_New(LAGraph_Graph* G, GrB_Matrix* A, int kind, char* msg) {
    clear_msg(msg);
    if (!G) return fail(msg, GrB_NULL_POINTER, "G is NULL");
    LAGraph_Graph g = new (std::nothrow) LAGraph_Graph_struct();
    if (!g) return fail(msg, GrB_OUT_OF_MEMORY, "out of memory");
    memset(g, 0, sizeof(*g));
    g->kind = kind;
    g->is_symmetric_structure = kind == 0 ? 1 : LAGRAPH_UNKNOWN;
    g->nself_edges = LAGRAPH_UNKNOWN;
    g->emin_state = g->emax_state = LAGRAPH_UNKNOWN;
    if (A) { g->A = *A; *A = nullptr; }
    *G = g;
    return GrB_SUCCESS;
}
int LAGraph_DeleteCached(LAGraph_Graph G, char* msg) {
    clear_msg(msg);
    if (!G) return GrB_SUCCESS;
    GrB_Matrix_free(&G->AT);
    GrB_Vector_free(&G->out_degree);
    GrB_Vector_free(&G->in_degree);
    GrB_Scalar_free(&G->emin);
    GrB_Scalar_free(&G->emax);
    G->is_symmetric_structure = G->kind == 0 ? 1 : LAGRAPH_UNKNOWN;
    G->nself_edges = LAGRAPH_UNKNOWN;
    G->emin_state = G->emax_state = LAGRAPH_UNKNOWN;
    return GrB_SUCCESS;
}
// frees G->A too: a caller that keeps the matrix sets G->A = NULL first (algo_procedures.rs:409-413)
int LAGraph_Delete(LAGraph_Graph* G, char* msg) {
    clear_msg(msg);
    if (!G || !*G) return GrB_SUCCESS;
    LAGraph_DeleteCached(*G, msg);
    GrB_Matrix_free(&(*G)->A);
    delete *G;
    *G = nullptr;
    return GrB_SUCCESS;
}
int LAGraph_CheckGraph(LAGraph_Graph G, char* msg) {
    clear_msg(msg);
    return check_graph(G, msg);
}
// G->AT = A' unless it exists already (left unchanged then, lagraph_bindings.rs:198); the engine keeps one transpose per
// snapshot, so a second graph over the same adjacency reuses it
int LAGraph_Cached_AT(LAGraph_Graph G, char* msg) {
    clear_msg(msg);
    if (const int r = check_graph(G, msg)) return r;
    if (G->AT) return GrB_SUCCESS;
    if (G->kind == 0) return LAGRAPH_CACHE_NOT_NEEDED;
    return guarded(msg, [&]() -> int {
        G->AT = new GB_Matrix_opaque(G->A->m.transpose());
        return GrB_SUCCESS;
    });
}
int LAGraph_Cached_OutDegree(LAGraph_Graph G, char* msg) {
    clear_msg(msg);
    if (const int r = check_graph(G, msg)) return r;
    if (G->out_degree) return GrB_SUCCESS;
    return guarded(msg, [&]() -> int {
        G->out_degree = degrees_of(G->A->m);
        return GrB_SUCCESS;
    });
}
int LAGraph_Cached_InDegree(LAGraph_Graph G, char* msg) {
    clear_msg(msg);
    if (const int r = check_graph(G, msg)) return r;
    if (G->in_degree) return GrB_SUCCESS;
    if (G->kind == 0) return LAGRAPH_CACHE_NOT_NEEDED;
    return guarded(msg, [&]() -> int {
        G->in_degree = degrees_of(G->AT ? G->AT->m : G->A->m.transpose());
        return GrB_SUCCESS;
    });
}
// LAGr_PageRank (lagraph_bindings.rs:549-558): an Advanced method — G->AT and G->out_degree must be cached
// (LAGRAPH_NOT_CACHED otherwise); centrality is a full GrB_FP32 vector; LAGRAPH_CONVERGENCE_FAILURE when itermax
// iterations did not reach tol
int LAGr_PageRank(GrB_Vector* centrality, int* iters, LAGraph_Graph G, float damping, float tol, int itermax, char* msg) {
    clear_msg(msg);
    if (!centrality || !iters) return fail(msg, GrB_NULL_POINTER, "centrality / iters is NULL");
    *centrality = nullptr;
    if (const int r = check_graph(G, msg)) return r;
    const bool symmetric = G->kind == 0 || (G->kind == 1 && G->is_symmetric_structure == 1);
    GrB_Matrix AT = symmetric ? G->A : G->AT;
    if (!AT) return fail(msg, LAGRAPH_NOT_CACHED, "G->AT is required");
    if (!G->out_degree) return fail(msg, LAGRAPH_NOT_CACHED, "G->out_degree is required");
    return guarded(msg, [&]() -> int {
        falkor::Context* c = fgshim::context();
        const uint64_t n = G->A->m.nrows();
        float* score = nullptr;
        check(fgpu_host_alloc(c->raw(), (n ? n : 1) * sizeof(float), (void**)&score), "LAGr_PageRank");
        GrB_Vector out = fgshim::vector_over_pinned(fgshim::type_fp32(), n, score, 0);
        int32_t it = 0, converged = 1;
        fgpu_info r = fgpu_pagerank_status(c->raw(), G->A->m.snapshot(), AT->m.snapshot(), nullptr, damping, tol, itermax, score, &it,
                                           &converged);
        if (r == FGPU_OK && itermax > 0 && !converged) {   // itermax iterations did not reach tol (one run: the engine reports it)
            *iters = it;
            GrB_Vector_free(&out);
            return fail(msg, LAGRAPH_CONVERGENCE_FAILURE, "pagerank failed to converge");
        }
        if (r != FGPU_OK) { GrB_Vector_free(&out); check(r, "LAGr_PageRank"); }
        *iters = it;
        *centrality = out;
        return GrB_SUCCESS;
    });
}
// ---- outside the engine's path: exported so algo_procedures.rs links, loud when called --------------------------------------
#define FG_NOT_ON_PATH(NAME) return fail(msg, GrB_NOT_IMPLEMENTED, #NAME ": not provided by the MI355X engine (traversal / BFS / PageRank only)")
int LAGr_ConnectedComponents(GrB_Vector* component, LAGraph_Graph, char* msg) { if (component) *component = nullptr; FG_NOT_ON_PATH(LAGr_ConnectedComponents); }
int LAGr_Betweenness(GrB_Vector* centrality, LAGraph_Graph, const GrB_Index*, int32_t, char* msg) { if (centrality) *centrality = nullptr; FG_NOT_ON_PATH(LAGr_Betweenness); }
int LAGraph_Cached_EMin(LAGraph_Graph, char* msg) { FG_NOT_ON_PATH(LAGraph_Cached_EMin); }
#else
// ---- LAGraphX -----------------------------------------------------------------------------------------------------------
// LAGr_BreadthFirstSearch_Extended (lagraphx_bindings.rs:585-594) as algo.BFS calls it (algo_procedures.rs:1079-1088):
// level(i) = hops from src for every vertex reached within max_level (max_level < 0: no limit), parent(i) = the vertex i was
// discovered from, parent(src) = src; unreached vertices hold no entry.  `dest` >= 0 (stop once a destination is reached) is
// a form the reference never issues: refused rather than guessed.
int LAGr_BreadthFirstSearch_Extended(GrB_Vector* level, GrB_Vector* parent, LAGraph_Graph G, GrB_Index src, int64_t max_level,
                                     int64_t dest, bool many_expected, char* msg) {
    clear_msg(msg);
    (void)many_expected;                                  // a hint about the expected frontier size: the engine's push / pull rule decides
    if (level) *level = nullptr;
    if (parent) *parent = nullptr;
    if (const int r = check_graph(G, msg)) return r;
    if (!level && !parent) return GrB_SUCCESS;            // nothing to compute
    const uint64_t n = G->A->m.nrows();
    if (src >= n) return fail(msg, GrB_INVALID_INDEX, "invalid source node");
    if (dest >= 0) return fail(msg, GrB_NOT_IMPLEMENTED, "LAGr_BreadthFirstSearch_Extended: dest >= 0 is not provided");
    return guarded(msg, [&]() -> int {
        falkor::Context* c = fgshim::context();
        int32_t* lv = nullptr;
        int64_t* pa = nullptr;
        check(fgpu_host_alloc(c->raw(), n * sizeof(int32_t), (void**)&lv), "LAGr_BreadthFirstSearch");
        GrB_Vector lvec = fgshim::vector_over_pinned(fgshim::type_int32(), n, lv, 1), pvec = nullptr;
        if (parent) {
            const fgpu_info r = fgpu_host_alloc(c->raw(), n * sizeof(int64_t), (void**)&pa);
            if (r != FGPU_OK) { GrB_Vector_free(&lvec); check(r, "LAGr_BreadthFirstSearch"); }
            pvec = fgshim::vector_over_pinned(fgshim::type_int64(), n, pa, 1);
        }
        // pull needs A'; a directed graph without a cached AT gets the engine's per-snapshot transpose (built once)
        const bool symmetric = G->kind == 0 || (G->kind == 1 && G->is_symmetric_structure == 1);
        Matrix at = symmetric ? G->A->m : (G->AT ? G->AT->m : G->A->m.transpose());
        const fgpu_info r = fgpu_bfs(c->raw(), G->A->m.snapshot(), at.snapshot(), src, max_level < 0 ? -1 : max_level, lv, pa, nullptr);
        if (r != FGPU_OK) { GrB_Vector_free(&lvec); GrB_Vector_free(&pvec); check(r, "LAGr_BreadthFirstSearch"); }
        if (level) *level = lvec; else GrB_Vector_free(&lvec);
        if (parent) *parent = pvec;
        return GrB_SUCCESS;
    });
}
// ---- outside the engine's path: exported so algo_procedures.rs links, loud when called --------------------------------------
#define FG_NOT_ON_PATH(NAME) return fail(msg, GrB_NOT_IMPLEMENTED, #NAME ": not provided by the MI355X engine (traversal / BFS / PageRank only)")
int LAGr_HarmonicCentrality(GrB_Vector* scores, GrB_Vector* reachable, LAGraph_Graph, GrB_Vector, char* msg) {
    if (scores) *scores = nullptr;
    if (reachable) *reachable = nullptr;
    FG_NOT_ON_PATH(LAGr_HarmonicCentrality);
}
int LAGr_MaxFlow(double* f, GrB_Matrix* flow, GrB_Matrix* res, LAGraph_Graph, GrB_Index, GrB_Index, char* msg) {
    if (f) *f = 0;
    if (flow) *flow = nullptr;
    if (res) *res = nullptr;
    FG_NOT_ON_PATH(LAGr_MaxFlow);
}
int LAGraph_cdlp(GrB_Vector* out, LAGraph_Graph, int, char* msg) { if (out) *out = nullptr; FG_NOT_ON_PATH(LAGraph_cdlp); }
int LAGraph_msf(GrB_Matrix* forest, GrB_Vector* comp, GrB_Matrix, bool, char* msg) {
    if (forest) *forest = nullptr;
    if (comp) *comp = nullptr;
    FG_NOT_ON_PATH(LAGraph_msf);
}
#endif

}  // extern "C"
