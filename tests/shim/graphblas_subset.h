/* graphblas_subset.h — C declarations TRANSCRIBED from the reference's bindgen output
 * (/root/reference/graph/src/graph/graphblas/mod.rs; the line of each `pub fn` / `pub static` is cited), i.e. exactly what
 * the unmodified Rust wrapper (matrix.rs:79-102) links against.  tests/shim/replay_matrix_rs.c is written against THIS
 * file only — it never sees falkordb_amd headers — and is linked with falkordb_amd/lib/libgraphblas.so. */
#ifndef GRAPHBLAS_SUBSET_H
#define GRAPHBLAS_SUBSET_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

typedef uint64_t GrB_Index;                                   /* mod.rs:271 */
typedef enum {                                                /* mod.rs:274-296 */
    GrB_SUCCESS = 0, GrB_NO_VALUE = 1, GxB_EXHAUSTED = 7089, GrB_UNINITIALIZED_OBJECT = -1, GrB_NULL_POINTER = -2,
    GrB_INVALID_VALUE = -3, GrB_INVALID_INDEX = -4, GrB_DOMAIN_MISMATCH = -5, GrB_DIMENSION_MISMATCH = -6,
    GrB_OUTPUT_NOT_EMPTY = -7, GrB_NOT_IMPLEMENTED = -8, GrB_ALREADY_SET = -9, GrB_PANIC = -101, GrB_OUT_OF_MEMORY = -102,
    GrB_INSUFFICIENT_SPACE = -103, GrB_INVALID_OBJECT = -104, GrB_INDEX_OUT_OF_BOUNDS = -105, GrB_EMPTY_OBJECT = -106
} GrB_Info;
enum { GrB_NONBLOCKING = 0, GrB_BLOCKING = 1 };               /* mod.rs:299-304 */
enum { GrB_COMPLETE = 0, GrB_MATERIALIZE = 1 };               /* mod.rs:3031-3034 */
enum { GrB_STORAGE_ORIENTATION_HINT = 100, GxB_SPARSITY_CONTROL = 7036, GxB_SPARSITY_STATUS = 7034 };   /* mod.rs:2887-2919 */
enum { GxB_HYPERSPARSE = 1, GxB_SPARSE = 2, GrB_ROWMAJOR = 0 };   /* mod.rs:159-160, 3006 */

typedef struct GB_Type_opaque* GrB_Type;
typedef struct GB_BinaryOp_opaque* GrB_BinaryOp;
typedef struct GB_UnaryOp_opaque* GrB_UnaryOp;
typedef struct GB_Semiring_opaque* GrB_Semiring;
typedef struct GB_Descriptor_opaque* GrB_Descriptor;         /* mod.rs:310 */
typedef struct GB_Scalar_opaque* GrB_Scalar;                 /* mod.rs:352 */
typedef struct GB_Matrix_opaque* GrB_Matrix;                 /* mod.rs:364 */
typedef struct GB_Iterator_opaque* GxB_Iterator;             /* mod.rs:383 */
typedef struct GB_Vector_opaque* GrB_Vector;                 /* mod.rs:358 */
enum { GrB_NAME = 10 };                                       /* mod.rs:2879 */
enum { GxB_MAX_NAME_LEN = 128 };                              /* mod.rs:158 */
#define GrB_INDEX_MAX ((1ull << 60) - 1)                      /* tensor.rs:143 */
typedef struct {                                              /* GxB_Container_struct, mod.rs:14165-14188 (608 bytes) */
    uint64_t nrows, ncols;
    int64_t nrows_nonempty, ncols_nonempty;
    uint64_t nvals;
    uint64_t u64_future[11];
    int32_t format, orientation, header_arena;
    uint32_t u32_future[13];
    GrB_Vector p, h, b, i, x;
    GrB_Vector vector_future[11];
    GrB_Matrix Y;
    GrB_Matrix matrix_future[15];
    bool iso, jumbled;
    bool bool_future[30];
    void* void_future[16];
} GxB_Container_struct;
typedef GxB_Container_struct* GxB_Container;

extern GrB_Type GrB_BOOL, GrB_UINT64;                         /* mod.rs:520, 544 */
extern GrB_UnaryOp GxB_ONE_BOOL;                              /* mod.rs:721 */
extern GrB_BinaryOp GrB_SECOND_UINT64, GxB_ANY_BOOL;          /* mod.rs:1301, 1547 */
extern GrB_Semiring GxB_ANY_PAIR_BOOL;                        /* mod.rs:6852 */
extern GrB_Descriptor GrB_DESC_C, GrB_DESC_RC, GrB_DESC_RCT0, GrB_DESC_RSC;   /* mod.rs:436, 484, 490, 508 */

GrB_Info GrB_init(int mode);                                  /* mod.rs:7964 */
GrB_Info GrB_finalize(void);                                  /* mod.rs:7967 */
GrB_Info GrB_Scalar_new(GrB_Scalar* s, GrB_Type type);        /* mod.rs:8677 */
GrB_Info GrB_Scalar_setElement_BOOL(GrB_Scalar s, bool x);    /* mod.rs:8726 */
GrB_Info GrB_Scalar_free(GrB_Scalar* s);
GrB_Info GrB_Matrix_new(GrB_Matrix* A, GrB_Type type, GrB_Index nrows, GrB_Index ncols);   /* mod.rs:9444 */
GrB_Info GrB_Matrix_free(GrB_Matrix* A);
GrB_Info GrB_Matrix_dup(GrB_Matrix* C, GrB_Matrix A);         /* mod.rs:9452 */
GrB_Info GrB_Matrix_clear(GrB_Matrix A);                      /* mod.rs:9476 */
GrB_Info GrB_Matrix_nrows(GrB_Index* nrows, GrB_Matrix A);    /* mod.rs:9479 */
GrB_Info GrB_Matrix_ncols(GrB_Index* ncols, GrB_Matrix A);    /* mod.rs:9485 */
GrB_Info GrB_Matrix_nvals(GrB_Index* nvals, GrB_Matrix A);    /* mod.rs:9491 */
GrB_Info GrB_Matrix_build_UINT64(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const uint64_t* X, GrB_Index nvals,
                                 GrB_BinaryOp dup);           /* mod.rs:9589 */
GrB_Info GxB_Matrix_build_Scalar(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, GrB_Scalar scalar, GrB_Index nvals);   /* mod.rs:9659 */
GrB_Info GrB_Matrix_setElement_BOOL(GrB_Matrix C, bool x, GrB_Index i, GrB_Index j);         /* mod.rs:9685 */
GrB_Info GrB_Matrix_setElement_UINT64(GrB_Matrix C, uint64_t x, GrB_Index i, GrB_Index j);   /* mod.rs:9749 */
GrB_Info GrB_Matrix_extractElement_BOOL(bool* x, GrB_Matrix A, GrB_Index i, GrB_Index j);    /* mod.rs:9797 */
GrB_Info GrB_Matrix_extractElement_UINT64(uint64_t* x, GrB_Matrix A, GrB_Index i, GrB_Index j);   /* mod.rs:9861 */
GrB_Info GxB_Matrix_isStoredElement(GrB_Matrix A, GrB_Index i, GrB_Index j);                 /* mod.rs:9917 */
GrB_Info GrB_Matrix_removeElement(GrB_Matrix C, GrB_Index i, GrB_Index j);                   /* mod.rs:9924 */
GrB_Info GrB_Matrix_get_INT32(GrB_Matrix object, int32_t* value, int field);                 /* mod.rs:10230 */
GrB_Info GrB_Matrix_set_INT32(GrB_Matrix object, int32_t value, int field);                  /* mod.rs:10713 */
GrB_Info GrB_Matrix_wait(GrB_Matrix object, int waitmode);                                    /* mod.rs:11078 */
GrB_Info GrB_mxm(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Matrix B,
                 GrB_Descriptor desc);                                                        /* mod.rs:11162-11171 */
GrB_Info GrB_Matrix_eWiseMult_Semiring(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A,
                                       GrB_Matrix B, GrB_Descriptor desc);                    /* mod.rs:11228 */
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_BinaryOp add, GrB_Matrix A,
                                      GrB_Matrix B, GrB_Descriptor desc);                     /* mod.rs:11316 */
GrB_Info GrB_Matrix_apply(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_UnaryOp op, GrB_Matrix A,
                          GrB_Descriptor desc);                                               /* mod.rs:12375 */
GrB_Info GrB_transpose(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Matrix A, GrB_Descriptor desc);   /* mod.rs:14013 */
GrB_Info GrB_Matrix_resize(GrB_Matrix C, GrB_Index nrows_new, GrB_Index ncols_new);          /* mod.rs:14055 */
GrB_Info GxB_Iterator_new(GxB_Iterator* iterator);                                            /* mod.rs:14848 */
GrB_Info GxB_Iterator_free(GxB_Iterator* iterator);
GrB_Info GxB_rowIterator_attach(GxB_Iterator iterator, GrB_Matrix A, GrB_Descriptor desc);   /* mod.rs:14875 */
GrB_Index GxB_rowIterator_kount(GxB_Iterator iterator);                                       /* mod.rs:14882 */
GrB_Info GxB_rowIterator_seekRow(GxB_Iterator iterator, GrB_Index row);                       /* mod.rs:14885 */
GrB_Info GxB_rowIterator_nextRow(GxB_Iterator iterator);                                      /* mod.rs:14897 */
GrB_Info GxB_rowIterator_nextCol(GxB_Iterator iterator);                                      /* mod.rs:14900 */
GrB_Index GxB_rowIterator_getRowIndex(GxB_Iterator iterator);                                 /* mod.rs:14903 */
GrB_Index GxB_rowIterator_getColIndex(GxB_Iterator iterator);                                 /* mod.rs:14906 */
uint64_t GxB_Iterator_get_UINT64(GxB_Iterator iterator);                                      /* mod.rs:15024 */
/* the rest of matrix.rs:79-102's import list, and vector.rs:43-60's */
GrB_Info GrB_Matrix_build_BOOL(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const bool* X, GrB_Index nvals,
                               GrB_BinaryOp dup);                                             /* mod.rs:9509 */
GrB_Info GxB_Matrix_memoryUsage(size_t* size, GrB_Matrix A);                                  /* mod.rs:9497 */
GrB_Info GxB_Matrix_iso(bool* iso, GrB_Matrix A);
GrB_Info GxB_Matrix_fprint(GrB_Matrix A, const char* name, int pr, void* f);                  /* mod.rs:14132 */
GrB_Info GxB_Container_new(GxB_Container* Container);                                         /* mod.rs:14240 */
GrB_Info GxB_Container_free(GxB_Container* object);                                           /* mod.rs:15081 */
GrB_Info GxB_load_Matrix_from_Container(GrB_Matrix A, GxB_Container Container, GrB_Descriptor desc);     /* mod.rs:14250 */
GrB_Info GxB_unload_Matrix_into_Container(GrB_Matrix A, GxB_Container Container, GrB_Descriptor desc);   /* mod.rs:14264 */
GrB_Info GrB_Vector_new(GrB_Vector* v, GrB_Type type, GrB_Index n);                           /* mod.rs:8894 */
GrB_Info GrB_Vector_free(GrB_Vector* object);                                                 /* mod.rs:15069 */
GrB_Info GrB_Vector_clear(GrB_Vector v);                                                      /* mod.rs:8924 */
GrB_Info GrB_Vector_size(GrB_Index* n, GrB_Vector v);                                         /* mod.rs:8927 */
GrB_Info GrB_Vector_resize(GrB_Vector w, GrB_Index nrows_new);                                /* mod.rs:14062 */
GrB_Info GrB_Vector_wait(GrB_Vector object, int waitmode);                                    /* mod.rs:11072 */
GrB_Info GrB_Vector_setElement_BOOL(GrB_Vector w, bool x, GrB_Index i);                       /* mod.rs:9102 */
GrB_Info GrB_Vector_setElement_UINT64(GrB_Vector w, uint64_t x, GrB_Index i);                 /* mod.rs:9158 */
GrB_Info GrB_Vector_removeElement(GrB_Vector v, GrB_Index i);                                 /* mod.rs:9318 */
GrB_Info GrB_Type_get_String(GrB_Type object, char* value, int field);                        /* mod.rs:10503 */
GrB_Info GxB_Type_from_name(GrB_Type* type, const char* type_name);                           /* mod.rs:8075 */
GrB_Info GxB_Vector_Iterator_attach(GxB_Iterator iterator, GrB_Vector v, GrB_Descriptor desc);   /* mod.rs:14972 */
GrB_Info GxB_Vector_Iterator_seek(GxB_Iterator iterator, GrB_Index p);                        /* mod.rs:14985 */
GrB_Info GxB_Vector_Iterator_next(GxB_Iterator iterator);                                     /* mod.rs:14991 */
GrB_Index GxB_Vector_Iterator_getIndex(GxB_Iterator iterator);                                /* mod.rs:14997 */
GrB_Info GxB_Vector_load(GrB_Vector V, void** X, GrB_Type type, uint64_t n, uint64_t X_memsize, int handling,
                         GrB_Descriptor desc);                                                /* mod.rs:14278 */
GrB_Info GxB_Vector_unload(GrB_Vector V, void** X, GrB_Type* type, uint64_t* n, uint64_t* X_memsize, int* handling,
                           GrB_Descriptor desc);                                              /* mod.rs:14289 */
GrB_Info GxB_Vector_serialize(void** blob_handle, GrB_Index* blob_size, GrB_Vector u, GrB_Descriptor desc);   /* mod.rs:14717 */
GrB_Info GxB_Vector_deserialize(GrB_Vector* w, GrB_Type type, const void* blob, GrB_Index blob_size,
                                GrB_Descriptor desc);                                         /* mod.rs:14768 */
#endif
