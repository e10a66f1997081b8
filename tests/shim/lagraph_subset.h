/* lagraph_subset.h — C declarations TRANSCRIBED from the reference's bindgen output for LAGraph
 * (/root/reference/graph/src/graph/graphblas/lagraph_bindings.rs and lagraphx_bindings.rs; the line of each `pub fn` /
 * `pub struct` is cited) plus the three GraphBLAS vector calls algo_procedures.rs reads results with (mod.rs).
 * tests/shim/replay_algo_rs.c is written against THIS file and graphblas_subset.h only, and is linked with
 * falkordb_amd/lib/liblagraphx.so, liblagraph.so and libgraphblas.so — the three names build.rs:50-52 links. */
#ifndef LAGRAPH_SUBSET_H
#define LAGRAPH_SUBSET_H
#include "graphblas_subset.h"

#define LAGRAPH_MSG_LEN 256                                   /* lagraph_bindings.rs:30 */
enum { LAGRAPH_INVALID_GRAPH = -1000, LAGRAPH_NOT_CACHED = -1003, LAGRAPH_CONVERGENCE_FAILURE = -1005,
       LAGRAPH_CACHE_NOT_NEEDED = 1000, LAGRAPH_UNKNOWN = -1 };   /* lagraph_bindings.rs:23-31 */
typedef enum { LAGraph_ADJACENCY_UNDIRECTED = 0, LAGraph_ADJACENCY_DIRECTED = 1, LAGraph_KIND_UNKNOWN = -1 } LAGraph_Kind;   /* :77-84 */
typedef enum { LAGraph_FALSE = 0, LAGraph_TRUE = 1, LAGraph_BOOLEAN_UNKNOWN = -1 } LAGraph_Boolean;   /* :88-95 */
typedef int32_t LAGraph_State;
typedef struct {                                              /* LAGraph_Graph_struct, lagraph_bindings.rs:108-129 (88 bytes) */
    GrB_Matrix A;
    LAGraph_Kind kind;
    GrB_Matrix AT;
    GrB_Vector out_degree;
    GrB_Vector in_degree;
    LAGraph_Boolean is_symmetric_structure;
    int64_t nself_edges;
    GrB_Scalar emin;
    LAGraph_State emin_state;
    GrB_Scalar emax;
    LAGraph_State emax_state;
} LAGraph_Graph_struct;
typedef LAGraph_Graph_struct* LAGraph_Graph;                  /* lagraph_bindings.rs:157 */

int LAGraph_Init(char* msg);                                  /* lagraph_bindings.rs:160 */
int LAGraph_Finalize(char* msg);                              /* lagraph_bindings.rs:172 */
int LAGraph_New(LAGraph_Graph* G, GrB_Matrix* A, LAGraph_Kind kind, char* msg);   /* lagraph_bindings.rs:176-181 */
int LAGraph_Delete(LAGraph_Graph* G, char* msg);              /* lagraph_bindings.rs:185-188 */
int LAGraph_Cached_AT(LAGraph_Graph G, char* msg);            /* lagraph_bindings.rs:199-202 */
int LAGraph_Cached_OutDegree(LAGraph_Graph G, char* msg);     /* lagraph_bindings.rs:213-216 */
int LAGr_PageRank(GrB_Vector* centrality, int* iters, LAGraph_Graph G, float damping, float tol, int itermax, char* msg);   /* :550-558 */
int LAGr_ConnectedComponents(GrB_Vector* component, LAGraph_Graph G, char* msg);   /* lagraph_bindings.rs:522-526 */
int LAGr_BreadthFirstSearch_Extended(GrB_Vector* level, GrB_Vector* parent, LAGraph_Graph G, GrB_Index src, int64_t max_level,
                                     int64_t dest, bool many_expected, char* msg);   /* lagraphx_bindings.rs:585-594 */

GrB_Info GxB_init(int mode, void* (*user_malloc)(size_t), void* (*user_calloc)(size_t, size_t),
                  void* (*user_realloc)(void*, size_t), void (*user_free)(void*));            /* mod.rs:7970 */
GrB_Info GrB_Vector_nvals(GrB_Index* nvals, GrB_Vector v);                                    /* mod.rs:8933 */
GrB_Info GrB_Vector_extractTuples_INT64(GrB_Index* I, int64_t* X, GrB_Index* nvals, GrB_Vector v);   /* mod.rs:9356 */
GrB_Info GrB_Vector_extractTuples_FP64(GrB_Index* I, double* X, GrB_Index* nvals, GrB_Vector v);     /* mod.rs:9404 */
#endif
