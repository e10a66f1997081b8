/* replay_matrix_rs.c — the call sequences of the reference's matrix.rs, issued call for call through the GraphBLAS C ABI
 * (declarations: graphblas_subset.h, transcribed from the bindgen output) against falkordb_amd/lib/libgraphblas.so:
 *   - Matrix::new + pin_sparse (matrix.rs:405-426, 1214-1235), Matrix::<bool>::build via GxB_Matrix_build_Scalar
 *     (:1281-1303), Matrix::<u64>::build via GrB_Matrix_build_UINT64 (:1186-1210);
 *   - Matrix::delta_lmxm (:1317-1402) — dp.wait / dm.wait, the clean fast path, mk = F*dm, ac = F*dp, the masked product
 *     under GrB_DESC_RSC with C aliasing A, the closing eWiseAdd — once per hop, as expand_batch chains them
 *     (cond_traverse.rs:600-605);
 *   - matrix::Iter::new / next (:1500-1605) to read F back through the row iterator protocol.
 * Input (text, argv[1]): n nsrc nhops valued; then nnz_m + pairs (+ value if valued); nnz_dp + pairs (+ value); nnz_dm + pairs;
 * then nsrc source ids.  Output (stdout): "nvals <k>" then one "row col" line per entry in iteration order, then
 * "probe <info> <value>" lines for a few extractElement / isStoredElement calls.  tests/test_gpu_shim.py compares it with
 * the oracle. */
#include <stdio.h>
#include <stdlib.h>

#include "graphblas_subset.h"

#define OK(call)                                                                       \
    do {                                                                               \
        GrB_Info info_ = (call);                                                       \
        if (info_ != GrB_SUCCESS) { fprintf(stderr, "%s -> %d (line %d)\n", #call, (int)info_, __LINE__); exit(2); } \
    } while (0)

static void pin_sparse(GrB_Matrix m) {                        /* matrix.rs:405-426 */
    OK(GrB_Matrix_set_INT32(m, GxB_SPARSE | GxB_HYPERSPARSE, GxB_SPARSITY_CONTROL));
    OK(GrB_Matrix_set_INT32(m, GrB_ROWMAJOR, GrB_STORAGE_ORIENTATION_HINT));
}

static GrB_Matrix new_matrix(GrB_Type t, GrB_Index nrows, GrB_Index ncols) {
    GrB_Matrix m = NULL;
    OK(GrB_Matrix_new(&m, t, nrows, ncols));
    pin_sparse(m);
    return m;
}

static GrB_Matrix read_matrix(FILE* f, GrB_Index n, int valued) {
    unsigned long long nnz = 0;
    if (fscanf(f, "%llu", &nnz) != 1) exit(3);
    GrB_Index* I = malloc((nnz + 1) * sizeof(GrB_Index));
    GrB_Index* J = malloc((nnz + 1) * sizeof(GrB_Index));
    uint64_t* X = malloc((nnz + 1) * sizeof(uint64_t));
    for (unsigned long long k = 0; k < nnz; ++k) {
        unsigned long long i, j, x = 1;
        if (fscanf(f, "%llu %llu", &i, &j) != 2) exit(3);
        if (valued && fscanf(f, "%llu", &x) != 1) exit(3);
        I[k] = i; J[k] = j; X[k] = x;
    }
    GrB_Matrix m = new_matrix(valued ? GrB_UINT64 : GrB_BOOL, n, n);
    if (valued) {
        OK(GrB_Matrix_build_UINT64(m, I, J, X, nnz, GrB_SECOND_UINT64));           /* matrix.rs:1186-1210 */
    } else {
        GrB_Scalar s = NULL;                                                         /* matrix.rs:1281-1303 */
        OK(GrB_Scalar_new(&s, GrB_BOOL));
        OK(GrB_Scalar_setElement_BOOL(s, true));
        OK(GxB_Matrix_build_Scalar(m, I, J, s, nnz));
        OK(GrB_Scalar_free(&s));
    }
    free(I); free(J); free(X);
    return m;
}

static GrB_Index nvals_of(GrB_Matrix m) {
    GrB_Index k = 0;
    OK(GrB_Matrix_nvals(&k, m));
    return k;
}

/* Matrix::delta_lmxm (matrix.rs:1317-1402), `self` = F */
static void delta_lmxm(GrB_Matrix F, GrB_Matrix m, GrB_Matrix dp, GrB_Matrix dm) {
    OK(GrB_Matrix_wait(dp, GrB_MATERIALIZE));                                        /* :1328-1329 */
    OK(GrB_Matrix_wait(dm, GrB_MATERIALIZE));
    const GrB_Index dp_nvals = nvals_of(dp), dm_nvals = nvals_of(dm);
    if (dp_nvals == 0 && dm_nvals == 0) {                                            /* :1333-1337 -> lmxm :930-947 */
        OK(GrB_mxm(F, NULL, NULL, GxB_ANY_PAIR_BOOL, F, m, NULL));
        return;
    }
    GrB_Index nrows = 0, ncols = 0;
    OK(GrB_Matrix_nrows(&nrows, F));
    OK(GrB_Matrix_ncols(&ncols, m));
    GrB_Matrix mask = NULL, accum = NULL;
    if (dm_nvals > 0) {                                                              /* :1343-1361 */
        GrB_Matrix mk = new_matrix(GrB_BOOL, nrows, ncols);
        OK(GrB_mxm(mk, NULL, NULL, GxB_ANY_PAIR_BOOL, F, dm, NULL));
        if (nvals_of(mk) > 0) mask = mk; else OK(GrB_Matrix_free(&mk));
    }
    if (dp_nvals > 0) {                                                              /* :1363-1380 */
        GrB_Matrix ac = new_matrix(GrB_BOOL, nrows, ncols);
        OK(GrB_mxm(ac, NULL, NULL, GxB_ANY_PAIR_BOOL, F, dp, NULL));
        if (nvals_of(ac) > 0) accum = ac; else OK(GrB_Matrix_free(&ac));
    }
    OK(GrB_mxm(F, mask, NULL, GxB_ANY_PAIR_BOOL, F, m, mask ? GrB_DESC_RSC : NULL)); /* :1382-1396, C aliases A */
    if (accum) OK(GrB_Matrix_eWiseAdd_BinaryOp(F, NULL, NULL, GxB_ANY_BOOL, F, accum, NULL));   /* :1398-1400 via :852-874 */
    if (mask) OK(GrB_Matrix_free(&mask));
    if (accum) OK(GrB_Matrix_free(&accum));
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "r");
    if (!f) return 1;
    unsigned long long n, nsrc, nhops;
    int valued;
    if (fscanf(f, "%llu %llu %llu %d", &n, &nsrc, &nhops, &valued) != 4) return 3;
    OK(GrB_init(GrB_NONBLOCKING));
    GrB_Matrix m = read_matrix(f, n, valued), dp = read_matrix(f, n, valued), dm = read_matrix(f, n, 0);
    GrB_Matrix F = new_matrix(GrB_BOOL, nsrc, n);
    for (unsigned long long i = 0; i < nsrc; ++i) {                                   /* cond_traverse.rs:600-601 */
        unsigned long long s;
        if (fscanf(f, "%llu", &s) != 1) return 3;
        OK(GrB_Matrix_setElement_BOOL(F, true, i, s));                                /* pending tuples: non-blocking mode */
    }
    fclose(f);
    for (unsigned long long h = 0; h < nhops; ++h) delta_lmxm(F, m, dp, dm);          /* cond_traverse.rs:602-605 */
    OK(GrB_Matrix_wait(F, GrB_MATERIALIZE));
    printf("nvals %llu\n", (unsigned long long)nvals_of(F));
    /* matrix::Iter::new(F, 0, nsrc - 1) + next() until depleted (matrix.rs:1500-1605) */
    GxB_Iterator it = NULL;
    OK(GxB_Iterator_new(&it));
    OK(GxB_rowIterator_attach(it, F, NULL));
    const GrB_Index max_row = nsrc - 1;
    GrB_Info info = GxB_rowIterator_seekRow(it, 0);
    while (info == GrB_NO_VALUE && GxB_rowIterator_getRowIndex(it) < max_row) info = GxB_rowIterator_nextRow(it);
    int depleted = info != GrB_SUCCESS || GxB_rowIterator_getRowIndex(it) > max_row;
    while (!depleted) {
        printf("%llu %llu\n", (unsigned long long)GxB_rowIterator_getRowIndex(it), (unsigned long long)GxB_rowIterator_getColIndex(it));
        if (GxB_rowIterator_nextCol(it) != GrB_SUCCESS) {
            info = GxB_rowIterator_nextRow(it);
            while (info == GrB_NO_VALUE && GxB_rowIterator_getRowIndex(it) < max_row) info = GxB_rowIterator_nextRow(it);
            depleted = info != GrB_SUCCESS || GxB_rowIterator_getRowIndex(it) > max_row;
        }
    }
    OK(GxB_Iterator_free(&it));
    /* point probes the way ExpandInto issues them (matrix.rs:1158-1172, 731-737): first / last stored entry of m, a hole */
    {
        bool b = false;
        uint64_t x = 0;
        GrB_Info i1 = valued ? GrB_Matrix_extractElement_UINT64(&x, m, 0, 0) : GrB_Matrix_extractElement_BOOL(&b, m, 0, 0);
        printf("probe %d %llu\n", (int)i1, valued ? (unsigned long long)x : (unsigned long long)b);
        printf("probe %d 0\n", (int)GxB_Matrix_isStoredElement(m, n - 1, n - 1));
        printf("probe %d 0\n", (int)GxB_Matrix_isStoredElement(m, n, 0));            /* out of range: GrB_INVALID_INDEX */
    }
    OK(GrB_Matrix_free(&F)); OK(GrB_Matrix_free(&m)); OK(GrB_Matrix_free(&dp)); OK(GrB_Matrix_free(&dm));
    OK(GrB_finalize());
    return 0;
}
