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
 * "probe <info> <value>" lines for a few extractElement / isStoredElement calls, then
 *   - "container ..." : Encode<19> for Matrix (matrix.rs:506-546: unload into a container, every vector unloaded to its
 *     array and loaded back, the matrix reloaded) followed by Decode<19> (:428-504) of the captured bytes into a NEW
 *     matrix, compared entry for entry with the original;
 *   - "ids ..." / "u64vec ..." : the Vector<bool> id list of a multi-edge pair (tensor.rs:1111-1120) through
 *     GxB_Vector_serialize / _deserialize and the vector iterator (vector.rs:150-239, 525-606), and a Vector<u64>;
 *   - "mem <bytes> iso <0|1>" : GxB_Matrix_memoryUsage / GxB_Matrix_iso.
 * tests/test_gpu_shim.py compares it with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

/* ---- Encode<19> / Decode<19> for Matrix<T> (matrix.rs:428-546) and Vector<bool> (vector.rs:241-420) ------------------------ */
typedef struct { void* bytes; uint64_t n_entries, n_bytes; int handling; char type_name[GxB_MAX_NAME_LEN]; } WireVector;

static void encode_vector(GrB_Vector v, WireVector* out) {      /* vector.rs:241-309 */
    void* arr = NULL; GrB_Type type = NULL; uint64_t n = 0, nb = 0; int handling = 0;
    OK(GxB_Vector_unload(v, &arr, &type, &n, &nb, &handling, NULL));
    memset(out->type_name, 0, sizeof(out->type_name));
    OK(GrB_Type_get_String(type, out->type_name, GrB_NAME));
    out->bytes = malloc(nb ? nb : 1);
    if (nb) memcpy(out->bytes, arr, nb);
    out->n_entries = n; out->n_bytes = nb; out->handling = handling;
    OK(GxB_Vector_load(v, &arr, type, n, nb, handling, NULL));   /* "Reload the vector so it remains usable" */
}
static GrB_Vector decode_vector(const WireVector* w) {           /* vector.rs:311-420 */
    GrB_Type type = NULL;
    OK(GxB_Type_from_name(&type, w->type_name));
    GrB_Vector v = NULL;
    OK(GrB_Vector_new(&v, type, 0));
    void* arr = NULL;
    if (w->n_bytes) { arr = malloc(w->n_bytes); memcpy(arr, w->bytes, w->n_bytes); }
    OK(GxB_Vector_load(v, &arr, type, w->n_entries, w->n_bytes, w->handling, NULL));
    return v;
}
/* returns 1 when decode(encode(m)) holds exactly m's entries (and values) and m itself survived its own encode */
static int container_round_trip(GrB_Matrix m, int valued, unsigned long long* format, int* iso) {
    const GrB_Index before = nvals_of(m);
    GxB_Container c = NULL;
    OK(GxB_Container_new(&c));
    OK(GxB_unload_Matrix_into_Container(m, c, NULL));
    unsigned char raw[sizeof(GxB_Container_struct)];
    memcpy(raw, c, sizeof(raw));
    *format = (unsigned long long)c->format; *iso = c->iso;
    WireVector wx, wh, wp, wi, wb;
    encode_vector(c->x, &wx); encode_vector(c->h, &wh); encode_vector(c->p, &wp); encode_vector(c->i, &wi); encode_vector(c->b, &wb);
    OK(GxB_load_Matrix_from_Container(m, c, NULL));
    OK(GxB_Container_free(&c));
    if (nvals_of(m) != before) return 0;
    /* decode: a fresh container, the struct bytes copied over it, the pointers nullified and refilled (matrix.rs:441-468) */
    OK(GxB_Container_new(&c));
    GrB_Vector fresh[5] = {c->x, c->h, c->p, c->i, c->b};
    for (int k = 0; k < 5; ++k) OK(GrB_Vector_free(&fresh[k]));      /* (the Rust code leaks these into the overwritten struct) */
    memcpy(c, raw, sizeof(raw));
    c->x = c->h = c->b = c->i = c->p = NULL; c->Y = NULL;
    c->x = decode_vector(&wx); c->h = decode_vector(&wh); c->p = decode_vector(&wp); c->i = decode_vector(&wi); c->b = decode_vector(&wb);
    GrB_Matrix m2 = NULL;
    OK(GrB_Matrix_new(&m2, GrB_BOOL, 0, 0));
    pin_sparse(m2);
    OK(GxB_load_Matrix_from_Container(m2, c, NULL));
    OK(GrB_Matrix_wait(m2, GrB_MATERIALIZE));
    OK(GxB_Container_free(&c));
    free(wx.bytes); free(wh.bytes); free(wp.bytes); free(wi.bytes); free(wb.bytes);
    int same = nvals_of(m2) == before;
    GrB_Index nr = 0, nr2 = 0;
    OK(GrB_Matrix_nrows(&nr, m)); OK(GrB_Matrix_nrows(&nr2, m2));
    same = same && nr == nr2;
    GxB_Iterator a = NULL, b = NULL;
    OK(GxB_Iterator_new(&a)); OK(GxB_Iterator_new(&b));
    OK(GxB_rowIterator_attach(a, m, NULL)); OK(GxB_rowIterator_attach(b, m2, NULL));
    for (GrB_Index r = 0; r < nr && same; ++r) {
        GrB_Info ia = GxB_rowIterator_seekRow(a, r), ib = GxB_rowIterator_seekRow(b, r);
        if (ia != ib) { same = 0; break; }
        while (ia == GrB_SUCCESS) {
            if (GxB_rowIterator_getColIndex(a) != GxB_rowIterator_getColIndex(b)) { same = 0; break; }
            if (valued && GxB_Iterator_get_UINT64(a) != GxB_Iterator_get_UINT64(b)) { same = 0; break; }
            ia = GxB_rowIterator_nextCol(a); ib = GxB_rowIterator_nextCol(b);
            if (ia != ib) { same = 0; break; }
        }
    }
    OK(GxB_Iterator_free(&a)); OK(GxB_Iterator_free(&b));
    OK(GrB_Matrix_free(&m2));
    return same;
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
    {   /* Encode<19> + Decode<19> of the base matrix and of the (small, hypersparse-leaning) delta-minus layer */
        unsigned long long fmt = 0; int iso = 0;
        const int ok_m = container_round_trip(m, valued, &fmt, &iso);
        printf("container m %llu %llu %d %d\n", (unsigned long long)nvals_of(m), fmt, iso, ok_m);
        const int ok_dm = container_round_trip(dm, 0, &fmt, &iso);
        printf("container dm %llu %llu %d %d\n", (unsigned long long)nvals_of(dm), fmt, iso, ok_dm);
    }
    {   /* the id list of a multi-edge pair: Vector<bool> of length GrB_INDEX_MAX, true at every edge id (tensor.rs:1111-1120) */
        GrB_Vector v = NULL;
        OK(GrB_Vector_new(&v, GrB_BOOL, GrB_INDEX_MAX));
        const GrB_Index ids[] = {500, 5, 501, 7, 1ull << 40};
        for (int k = 0; k < 5; ++k) OK(GrB_Vector_setElement_BOOL(v, true, ids[k]));
        OK(GrB_Vector_removeElement(v, 7));
        OK(GrB_Vector_wait(v, GrB_MATERIALIZE));
        void* blob = NULL; GrB_Index blob_size = 0;
        OK(GxB_Vector_serialize(&blob, &blob_size, v, NULL));                 /* encode_blob, vector.rs:150-174 */
        GrB_Vector back = NULL;
        OK(GxB_Vector_deserialize(&back, NULL, blob, blob_size, NULL));      /* decode_blob, vector.rs:176-196 */
        free(blob);
        GrB_Index sz = 0;
        OK(GrB_Vector_size(&sz, back));
        printf("ids %d", sz == GrB_INDEX_MAX);
        GxB_Iterator vi = NULL;
        OK(GxB_Iterator_new(&vi));
        OK(GxB_Vector_Iterator_attach(vi, back, NULL));                       /* Iter::new, vector.rs:540-565 */
        GrB_Info vinfo = GxB_Vector_Iterator_seek(vi, 0);
        while (vinfo != GxB_EXHAUSTED) {
            printf(" %llu", (unsigned long long)GxB_Vector_Iterator_getIndex(vi));
            vinfo = GxB_Vector_Iterator_next(vi);
        }
        printf("\n");
        OK(GxB_Iterator_free(&vi));
        OK(GrB_Vector_free(&back)); OK(GrB_Vector_free(&v));
        GrB_Vector u = NULL;                                                  /* Vector<u64> (vector.rs:421-450, 591-606) */
        OK(GrB_Vector_new(&u, GrB_UINT64, 100));
        OK(GrB_Vector_setElement_UINT64(u, 77, 9)); OK(GrB_Vector_setElement_UINT64(u, 1ull << 63, 3));
        OK(GxB_Vector_serialize(&blob, &blob_size, u, NULL));
        OK(GxB_Vector_deserialize(&back, NULL, blob, blob_size, NULL));
        free(blob);
        OK(GxB_Iterator_new(&vi));
        OK(GxB_Vector_Iterator_attach(vi, back, NULL));
        printf("u64vec");
        vinfo = GxB_Vector_Iterator_seek(vi, 0);
        while (vinfo != GxB_EXHAUSTED) {
            printf(" %llu:%llu", (unsigned long long)GxB_Vector_Iterator_getIndex(vi), (unsigned long long)GxB_Iterator_get_UINT64(vi));
            vinfo = GxB_Vector_Iterator_next(vi);
        }
        printf("\n");
        OK(GxB_Iterator_free(&vi));
        OK(GrB_Vector_clear(u)); OK(GrB_Vector_resize(u, 10));
        OK(GrB_Vector_free(&back)); OK(GrB_Vector_free(&u));
    }
    {
        size_t mem = 0; bool iso = false;
        OK(GxB_Matrix_memoryUsage(&mem, m));
        OK(GxB_Matrix_iso(&iso, m));
        printf("mem %llu iso %d\n", (unsigned long long)mem, (int)iso);
        OK(GxB_Matrix_fprint(m, "m", 0, NULL));                              /* GxB_SILENT */
    }
    OK(GrB_Matrix_free(&F)); OK(GrB_Matrix_free(&m)); OK(GrB_Matrix_free(&dp)); OK(GrB_Matrix_free(&dm));
    OK(GrB_finalize());
    return 0;
}
