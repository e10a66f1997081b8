// shim_internal.hpp — the objects behind the opaque GraphBLAS handles of the tier-2 shim, shared by libgraphblas.so
// (graphblas_shim.cpp) and the LAGraph-named libraries on top of it (lagraph_shim.cpp).  Not part of any ABI: the reference
// only ever holds these through pointers (mod.rs declares every GrB_* handle as *mut of an opaque struct).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../host/host.hpp"

using falkor::Matrix;
using falkor::Type;
typedef uint64_t GrB_Index;

// GrB_Info (mod.rs:274-296)
enum {
    GrB_SUCCESS = 0, GrB_NO_VALUE = 1, GxB_EXHAUSTED = 7089, GrB_UNINITIALIZED_OBJECT = -1, GrB_NULL_POINTER = -2,
    GrB_INVALID_VALUE = -3, GrB_INVALID_INDEX = -4, GrB_DOMAIN_MISMATCH = -5, GrB_DIMENSION_MISMATCH = -6,
    GrB_OUTPUT_NOT_EMPTY = -7, GrB_NOT_IMPLEMENTED = -8, GrB_PANIC = -101, GrB_OUT_OF_MEMORY = -102, GrB_INSUFFICIENT_SPACE = -103,
    GrB_INVALID_OBJECT = -104, GrB_INDEX_OUT_OF_BOUNDS = -105,
};
typedef int GrB_Info;

struct GB_Type_opaque { int code; const char* name; size_t size; };   // 0 = BOOL, 1 = UINT64; 2.. = the integer types a
                                                                       // GxB_Container's p / h / i / b vectors come in
struct GB_BinaryOp_opaque { int code; };        // 0 = ANY_BOOL, 1 = SECOND_UINT64, 2 = ANY_UINT64
struct GB_UnaryOp_opaque { int code; };         // 0 = ONE_BOOL
struct GB_Semiring_opaque { int code; };        // 0 = ANY_PAIR_BOOL
struct GB_Descriptor_opaque { bool replace, structural, complement, t0, t1; };
struct GB_Global_opaque { int dummy; };
struct GB_Scalar_opaque { bool has; bool value; };
struct GB_Matrix_opaque {
    Matrix m;
    int32_t sparsity_control = 3;   // GxB_HYPERSPARSE | GxB_SPARSE
    int32_t orientation = 0;        // GrB_ROWMAJOR
    explicit GB_Matrix_opaque(Matrix mm) : m(std::move(mm)) {}
};
// GrB_Vector as the wrapper uses it (vector.rs): (a) a sparse BOOL / UINT64 vector filled by setElement and walked by the
// vector iterator (the id list of a multi-edge pair, tensor.rs:1111-1120: indices = edge ids), serialised as a blob;
// (b) the dense array a GxB_Container field holds, moved in and out with GxB_Vector_load / _unload (vector.rs:241-420).
struct GB_Vector_opaque {
    GB_Type_opaque* type = nullptr;
    GrB_Index n = 0;                         // length
    std::map<GrB_Index, uint64_t> s;         // (a) stored entries
    void* data = nullptr;                    // (b) adopted array: n entries of type->size bytes (iso vectors: 1 entry)
    uint64_t nbytes = 0;
    uint64_t nstored = 0;                    // entries `data` holds (n, or 1 for an iso array)
    int handling = 0;
    // (c) a result of the engine (LAGr_BreadthFirstSearch_Extended / LAGr_PageRank, lagraph_shim.cpp): `data` is a pinned
    // block of the engine's result pool holding n entries; `absent` says which of them count as not stored
    fgpu_ctx* pinned_owner = nullptr;        // non-null: release `data` with fgpu_free, not the GxB_init allocator
    int absent = 0;                          // 0 all stored, 1 negative = absent, 2 zero = absent
    int64_t stored_count = -1;               // entries present under `absent` (counted once, on first use)
};
struct GB_Iterator_opaque {
    GB_Vector_opaque* vec = nullptr;                                  // vector mode (GxB_Vector_Iterator_*)
    std::map<GrB_Index, uint64_t>::const_iterator vit;
    std::unique_ptr<Matrix> m;       // keeps the handle's state alive (the wrapper holds an Arc as well, matrix.rs:1472)
    GrB_Index nrows = 0, row = 0;    // current row; == nrows when exhausted
    GrB_Index w_lo = 1, w_hi = 0;    // rows covered by `buf` (empty window when w_lo > w_hi)
    std::vector<falkor::Entry> buf;  // entries of rows [w_lo, w_hi], ascending (row, col)
    size_t pos = 0, row_end = 0;     // current entry, end of the current row's run in buf
};

typedef GB_Type_opaque* GrB_Type;
typedef GB_BinaryOp_opaque* GrB_BinaryOp;
typedef GB_UnaryOp_opaque* GrB_UnaryOp;
typedef GB_Semiring_opaque* GrB_Semiring;
typedef GB_Descriptor_opaque* GrB_Descriptor;
typedef GB_Global_opaque* GrB_Global;
typedef GB_Scalar_opaque* GrB_Scalar;
typedef GB_Matrix_opaque* GrB_Matrix;
typedef GB_Iterator_opaque* GxB_Iterator;
typedef GB_Vector_opaque* GrB_Vector;

// GxB_Container_struct, field for field as bindgen lays it out (mod.rs:14165-14188; 608 bytes, the wrapper copies it raw,
// matrix.rs:451-456, 517-520)
struct GxB_Container_struct {
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
};
static_assert(sizeof(GxB_Container_struct) == 608, "GxB_Container_struct must match the bindgen layout (mod.rs:14190)");
typedef GxB_Container_struct* GxB_Container;


// helpers libgraphblas.so exports for the LAGraph-named libraries (C++ linkage: no GrB_* / LAGraph_* name is taken)
namespace fgshim {
falkor::Context* context();                      // the context GxB_init made; nullptr before it
GB_Type_opaque* type_int32();
GB_Type_opaque* type_int64();
GB_Type_opaque* type_fp32();
// a vector of n entries over a pinned result block of the engine (fgpu_host_alloc; released with fgpu_free): absent == 0
// every entry is stored, 1: negative entries are absent (BFS level / parent), 2: zero entries are absent (degrees)
GB_Vector_opaque* vector_over_pinned(GB_Type_opaque* type, GrB_Index n, void* pinned, int absent);
}  // namespace fgshim
