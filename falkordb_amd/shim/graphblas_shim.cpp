// graphblas_shim.cpp — tier 2 of the drop-in boundary: a `libgraphblas`-named library exporting the GrB_* /
// GxB_* symbols the reference's hot path binds (graph/src/graph/graphblas/mod.rs: GrB_mxm :11162-11171, GrB_Matrix_* :9444-
// 9935, GrB_Matrix_eWiseAdd_BinaryOp :11316, GrB_Matrix_eWiseMult_Semiring :11228, GrB_Matrix_apply :12375, GrB_transpose
// :14013, GxB_rowIterator_* :14875-14906; the list matrix.rs:79-102 imports) for GrB_BOOL / GrB_UINT64 matrices and the
// GxB_ANY_PAIR_BOOL / GxB_ANY_BOOL / GrB_SECOND_UINT64 / GxB_ONE_BOOL operators — implemented on the MI355X engine through
// the host layer's Matrix (host.hpp), i.e. on fgpu_* calls.  With it the UNMODIFIED Rust wrapper (matrix.rs) can link
// against this engine: every call form matrix.rs issues on the traversal path is accepted; any other form returns
// GrB_NOT_IMPLEMENTED instead of computing something else.  GrB_Info codes are the reference's (mod.rs:274-296).
//
// Semantics kept from GraphBLAS where the wrapper depends on them: non-blocking mode (writes queue as pending tuples until
// GrB_Matrix_wait / a reading call), in-place output with C aliasing an input (matrix.rs:935-943), GrB_NO_VALUE from
// extractElement / isStoredElement for an absent entry, the row iterator's SUCCESS / NO_VALUE / EXHAUSTED protocol
// (matrix.rs:1500-1605 drives it row by row), duplicate collapse in build (SECOND for UINT64).
#include <stdint.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../host/host.hpp"

using falkor::Matrix;
using falkor::Type;
typedef uint64_t GrB_Index;

// GrB_Info (mod.rs:274-296)
enum {
    GrB_SUCCESS = 0, GrB_NO_VALUE = 1, GxB_EXHAUSTED = 7089, GrB_UNINITIALIZED_OBJECT = -1, GrB_NULL_POINTER = -2,
    GrB_INVALID_VALUE = -3, GrB_INVALID_INDEX = -4, GrB_DOMAIN_MISMATCH = -5, GrB_DIMENSION_MISMATCH = -6,
    GrB_OUTPUT_NOT_EMPTY = -7, GrB_NOT_IMPLEMENTED = -8, GrB_PANIC = -101, GrB_OUT_OF_MEMORY = -102,
    GrB_INVALID_OBJECT = -104, GrB_INDEX_OUT_OF_BOUNDS = -105,
};
typedef int GrB_Info;

struct GB_Type_opaque { int code; };            // 0 = BOOL, 1 = UINT64
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
struct GB_Iterator_opaque {
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

namespace {
GB_Type_opaque t_bool{0}, t_u64{1};
GB_BinaryOp_opaque op_any_bool{0}, op_second_u64{1}, op_any_u64{2};
GB_UnaryOp_opaque op_one_bool{0};
GB_Semiring_opaque sr_any_pair_bool{0};
GB_Global_opaque global_obj{0};

std::mutex g_mu;
std::unique_ptr<falkor::Context> g_ctx;

falkor::Context* ctx() {
    std::lock_guard<std::mutex> g(g_mu);
    return g_ctx.get();
}

GrB_Info map_error(const falkor::GrbError& e) {
    switch (e.info) {
        case FGPU_NO_VALUE: return GrB_NO_VALUE;
        case FGPU_OOM: return GrB_OUT_OF_MEMORY;
        case FGPU_OUT_OF_BOUNDS: return GrB_INDEX_OUT_OF_BOUNDS;
        case FGPU_DIM_MISMATCH: return GrB_DIMENSION_MISMATCH;
        case FGPU_NULL_POINTER: return GrB_NULL_POINTER;
        case FGPU_INVALID: return GrB_INVALID_VALUE;
        default: return GrB_PANIC;
    }
}

// every entry point: C++ exceptions never cross the C ABI
template <typename F>
GrB_Info guarded(F&& f) {
    try {
        return f();
    } catch (const falkor::GrbError& e) {
        return map_error(e);
    } catch (const std::bad_alloc&) {
        return GrB_OUT_OF_MEMORY;
    } catch (...) {
        return GrB_PANIC;
    }
}

bool desc_is(GrB_Descriptor d, bool r, bool s, bool c, bool t0, bool t1) {
    if (!d) return !r && !s && !c && !t0 && !t1;
    return d->replace == r && d->structural == s && d->complement == c && d->t0 == t0 && d->t1 == t1;
}
bool same(GrB_Matrix a, GrB_Matrix b) { return a == b; }

// ---- row iterator over windows of rows ------------------------------------------------------------------------
void it_load(GB_Iterator_opaque* it, GrB_Index row) {
    // window of rows starting at `row`: small first (point lookups of expand_row), the MatrixIter growth rule is not
    // needed here because an empty window is crossed row by row by the caller anyway (matrix.rs:1529-1531)
    const GrB_Index hi = (it->nrows - 1 - row < 4095) ? it->nrows - 1 : row + 4095;
    it->buf = it->m->iter(row, hi);
    it->w_lo = row;
    it->w_hi = hi;
    it->pos = 0;
}
// position on `row` (< nrows): SUCCESS when it holds entries, NO_VALUE when it is empty
GrB_Info it_at(GB_Iterator_opaque* it, GrB_Index row) {
    it->row = row;
    if (row < it->w_lo || row > it->w_hi) it_load(it, row);
    // first entry with .row >= row (rows only move forward inside a window in the wrapper's loops: linear from pos when
    // possible, binary search otherwise)
    size_t lo = 0, hi = it->buf.size();
    if (it->pos < it->buf.size() && it->buf[it->pos].row <= row) lo = it->pos;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (it->buf[mid].row < row) lo = mid + 1; else hi = mid;
    }
    it->pos = lo;
    size_t e = lo;
    while (e < it->buf.size() && it->buf[e].row == row) ++e;
    it->row_end = e;
    return e > lo ? GrB_SUCCESS : GrB_NO_VALUE;
}
}  // namespace

#define SHIM_REQUIRE_INIT() do { if (!ctx()) return GrB_PANIC; } while (0)

extern "C" {

// ---- globals the wrapper imports (matrix.rs:79-102) ------------------------------------------------------------
GrB_Type GrB_BOOL = &t_bool;
GrB_Type GrB_UINT64 = &t_u64;
GrB_BinaryOp GxB_ANY_BOOL = &op_any_bool;
GrB_BinaryOp GrB_SECOND_UINT64 = &op_second_u64;
GrB_BinaryOp GxB_ANY_UINT64 = &op_any_u64;
GrB_UnaryOp GxB_ONE_BOOL = &op_one_bool;
GrB_Semiring GxB_ANY_PAIR_BOOL = &sr_any_pair_bool;
GrB_Global GrB_GLOBAL = &global_obj;

// the 31 predefined descriptors (mod.rs:424-612; matrix.rs:313-351 maps all of them): R = replace, S = structural mask,
// C = complemented mask, T0 / T1 = transpose the first / second input
#define SHIM_DESC(NAME, R, S, C, T0, T1)                \
    static GB_Descriptor_opaque d_##NAME{R, S, C, T0, T1}; \
    GrB_Descriptor GrB_DESC_##NAME = &d_##NAME;
SHIM_DESC(T1, 0, 0, 0, 0, 1) SHIM_DESC(T0, 0, 0, 0, 1, 0) SHIM_DESC(T0T1, 0, 0, 0, 1, 1)
SHIM_DESC(C, 0, 0, 1, 0, 0) SHIM_DESC(CT1, 0, 0, 1, 0, 1) SHIM_DESC(CT0, 0, 0, 1, 1, 0) SHIM_DESC(CT0T1, 0, 0, 1, 1, 1)
SHIM_DESC(S, 0, 1, 0, 0, 0) SHIM_DESC(ST1, 0, 1, 0, 0, 1) SHIM_DESC(ST0, 0, 1, 0, 1, 0) SHIM_DESC(ST0T1, 0, 1, 0, 1, 1)
SHIM_DESC(SC, 0, 1, 1, 0, 0) SHIM_DESC(SCT1, 0, 1, 1, 0, 1) SHIM_DESC(SCT0, 0, 1, 1, 1, 0) SHIM_DESC(SCT0T1, 0, 1, 1, 1, 1)
SHIM_DESC(R, 1, 0, 0, 0, 0) SHIM_DESC(RT1, 1, 0, 0, 0, 1) SHIM_DESC(RT0, 1, 0, 0, 1, 0) SHIM_DESC(RT0T1, 1, 0, 0, 1, 1)
SHIM_DESC(RC, 1, 0, 1, 0, 0) SHIM_DESC(RCT1, 1, 0, 1, 0, 1) SHIM_DESC(RCT0, 1, 0, 1, 1, 0) SHIM_DESC(RCT0T1, 1, 0, 1, 1, 1)
SHIM_DESC(RS, 1, 1, 0, 0, 0) SHIM_DESC(RST1, 1, 1, 0, 0, 1) SHIM_DESC(RST0, 1, 1, 0, 1, 0) SHIM_DESC(RST0T1, 1, 1, 0, 1, 1)
SHIM_DESC(RSC, 1, 1, 1, 0, 0) SHIM_DESC(RSCT1, 1, 1, 1, 0, 1) SHIM_DESC(RSCT0, 1, 1, 1, 1, 0) SHIM_DESC(RSCT0T1, 1, 1, 1, 1, 1)
#undef SHIM_DESC

// ---- init / options (matrix.rs:116-221) ---------------------------------------------------------------------------
GrB_Info GxB_init(int mode, void* (*mal)(size_t), void* (*cal)(size_t, size_t), void* (*rea)(void*, size_t), void (*fre)(void*)) {
    (void)cal; (void)rea; (void)mal; (void)fre;   // results are handed out through the GrB calls' own out-parameters
    if (mode != 0 && mode != 1) return GrB_INVALID_VALUE;
    std::lock_guard<std::mutex> g(g_mu);
    if (g_ctx) return GrB_INVALID_VALUE;          // initialised twice
    try {
        int dev = 0;
        if (const char* e = getenv("FGPU_DEVICE")) dev = atoi(e);
        g_ctx.reset(new falkor::Context(dev));   // throws without a HIP device: no CPU fallback behind this ABI either
    } catch (...) {
        return GrB_PANIC;                         // the wrapper turns this into Err(String): Redis refuses the module
    }
    return GrB_SUCCESS;
}
GrB_Info GrB_init(int mode) { return GxB_init(mode, nullptr, nullptr, nullptr, nullptr); }
GrB_Info GrB_finalize() {
    std::lock_guard<std::mutex> g(g_mu);
    g_ctx.reset();
    return GrB_SUCCESS;
}
GrB_Info GrB_Global_set_INT32(GrB_Global, int32_t, int) { return ctx() ? GrB_SUCCESS : GrB_PANIC; }   // JIT / thread knobs: nothing to set
GrB_Info GxB_Global_Option_set_INT32(int, int32_t) { return ctx() ? GrB_SUCCESS : GrB_PANIC; }

// ---- scalars (only what GxB_Matrix_build_Scalar needs, matrix.rs:1281-1303) ---------------------------------------
GrB_Info GrB_Scalar_new(GrB_Scalar* s, GrB_Type type) {
    if (!s || !type) return GrB_NULL_POINTER;
    if (type != GrB_BOOL) return GrB_NOT_IMPLEMENTED;
    *s = new (std::nothrow) GB_Scalar_opaque{false, false};
    return *s ? GrB_SUCCESS : GrB_OUT_OF_MEMORY;
}
GrB_Info GrB_Scalar_setElement_BOOL(GrB_Scalar s, bool x) {
    if (!s) return GrB_NULL_POINTER;
    s->has = true; s->value = x;
    return GrB_SUCCESS;
}
GrB_Info GrB_Scalar_free(GrB_Scalar* s) {
    if (s) { delete *s; *s = nullptr; }
    return GrB_SUCCESS;
}

// ---- matrix life cycle -----------------------------------------------------------------------------------------
GrB_Info GrB_Matrix_new(GrB_Matrix* A, GrB_Type type, GrB_Index nrows, GrB_Index ncols) {
    if (!A || !type) return GrB_NULL_POINTER;
    SHIM_REQUIRE_INIT();
    return guarded([&]() -> GrB_Info {
        *A = new GB_Matrix_opaque(Matrix(*ctx(), type == GrB_UINT64 ? Type::UInt64 : Type::Bool, nrows, ncols));
        return GrB_SUCCESS;
    });
}
GrB_Info GrB_Matrix_free(GrB_Matrix* A) {
    if (A) { delete *A; *A = nullptr; }
    return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_dup(GrB_Matrix* C, GrB_Matrix A) {
    if (!C || !A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info {
        GB_Matrix_opaque* c = new GB_Matrix_opaque(A->m.dup());
        c->sparsity_control = A->sparsity_control;
        c->orientation = A->orientation;
        *C = c;
        return GrB_SUCCESS;
    });
}
GrB_Info GrB_Matrix_nrows(GrB_Index* n, GrB_Matrix A) { if (!n || !A) return GrB_NULL_POINTER; *n = A->m.nrows(); return GrB_SUCCESS; }
GrB_Info GrB_Matrix_ncols(GrB_Index* n, GrB_Matrix A) { if (!n || !A) return GrB_NULL_POINTER; *n = A->m.ncols(); return GrB_SUCCESS; }
GrB_Info GrB_Matrix_nvals(GrB_Index* n, GrB_Matrix A) {
    if (!n || !A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info { *n = A->m.nvals(); return GrB_SUCCESS; });   // finishes pending work, as GraphBLAS does
}
GrB_Info GxB_Matrix_type(GrB_Type* type, GrB_Matrix A) {
    if (!type || !A) return GrB_NULL_POINTER;
    *type = A->m.type() == Type::UInt64 ? GrB_UINT64 : GrB_BOOL;
    return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_wait(GrB_Matrix A, int /* GrB_COMPLETE | GrB_MATERIALIZE */) {
    if (!A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info { A->m.wait(); return GrB_SUCCESS; });
}
GrB_Info GrB_Matrix_clear(GrB_Matrix A) {
    if (!A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info { A->m.clear(); return GrB_SUCCESS; });
}
GrB_Info GrB_Matrix_resize(GrB_Matrix C, GrB_Index nrows, GrB_Index ncols) {
    if (!C) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info { C->m.resize(nrows, ncols); return GrB_SUCCESS; });
}
// options the wrapper sets on every new matrix (pin_sparse, matrix.rs:405-426) and reads back in its tests
GrB_Info GrB_Matrix_set_INT32(GrB_Matrix A, int32_t value, int field) {
    if (!A) return GrB_NULL_POINTER;
    if (field == 7036 /* GxB_SPARSITY_CONTROL (mod.rs:2915) */) A->sparsity_control = value;
    else if (field == 100 /* GrB_STORAGE_ORIENTATION_HINT (mod.rs:2887) */) { if (value != 0 /* GrB_ROWMAJOR */) return GrB_NOT_IMPLEMENTED; A->orientation = value; }
    return GrB_SUCCESS;    // hyper-hash / will-wait hints: accepted, nothing to do
}
GrB_Info GrB_Matrix_get_INT32(GrB_Matrix A, int32_t* value, int field) {
    if (!A || !value) return GrB_NULL_POINTER;
    if (field == 7036) *value = A->sparsity_control;
    else if (field == 7034 /* GxB_SPARSITY_STATUS (mod.rs:2919) */) *value = 2 /* GxB_SPARSE */;
    else if (field == 100) *value = A->orientation;
    else *value = 0;
    return GrB_SUCCESS;
}

// ---- build / element access --------------------------------------------------------------------------------------
GrB_Info GxB_Matrix_build_Scalar(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, GrB_Scalar scalar, GrB_Index nvals) {
    if (!C || !scalar || (nvals && (!I || !J))) return GrB_NULL_POINTER;
    if (C->m.type() != Type::Bool || !scalar->has || !scalar->value) return GrB_NOT_IMPLEMENTED;   // iso TRUE build only
    return guarded([&]() -> GrB_Info {
        if (C->m.nvals()) return GrB_OUTPUT_NOT_EMPTY;
        C->m.build(std::vector<uint64_t>(I, I + nvals), std::vector<uint64_t>(J, J + nvals));   // duplicates collapse (:1686-1695)
        return GrB_SUCCESS;
    });
}
GrB_Info GrB_Matrix_build_UINT64(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const uint64_t* X, GrB_Index nvals,
                                 GrB_BinaryOp dup) {
    if (!C || (nvals && (!I || !J || !X))) return GrB_NULL_POINTER;
    if (C->m.type() != Type::UInt64) return GrB_DOMAIN_MISMATCH;
    if (dup && dup != GrB_SECOND_UINT64) return GrB_NOT_IMPLEMENTED;      // the wrapper passes SECOND (matrix.rs:1186-1210)
    return guarded([&]() -> GrB_Info {
        if (C->m.nvals()) return GrB_OUTPUT_NOT_EMPTY;
        std::vector<uint64_t> v(X, X + nvals);
        C->m.build(std::vector<uint64_t>(I, I + nvals), std::vector<uint64_t>(J, J + nvals), &v);
        return GrB_SUCCESS;
    });
}
GrB_Info GrB_Matrix_setElement_BOOL(GrB_Matrix C, bool x, GrB_Index i, GrB_Index j) {
    if (!C) return GrB_NULL_POINTER;
    if (i >= C->m.nrows() || j >= C->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info { C->m.set_element(i, j, x ? 1 : 0); return GrB_SUCCESS; });
}
GrB_Info GrB_Matrix_setElement_UINT64(GrB_Matrix C, uint64_t x, GrB_Index i, GrB_Index j) {
    if (!C) return GrB_NULL_POINTER;
    if (i >= C->m.nrows() || j >= C->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info { C->m.set_element(i, j, x); return GrB_SUCCESS; });
}
GrB_Info GrB_Matrix_removeElement(GrB_Matrix C, GrB_Index i, GrB_Index j) {
    if (!C) return GrB_NULL_POINTER;
    if (i >= C->m.nrows() || j >= C->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info { C->m.remove_element(i, j); return GrB_SUCCESS; });
}
GrB_Info GrB_Matrix_extractElement_BOOL(bool* x, GrB_Matrix A, GrB_Index i, GrB_Index j) {
    if (!x || !A) return GrB_NULL_POINTER;
    if (i >= A->m.nrows() || j >= A->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info {
        auto v = A->m.get(i, j);
        if (!v) return GrB_NO_VALUE;
        *x = *v != 0;
        return GrB_SUCCESS;
    });
}
GrB_Info GrB_Matrix_extractElement_UINT64(uint64_t* x, GrB_Matrix A, GrB_Index i, GrB_Index j) {
    if (!x || !A) return GrB_NULL_POINTER;
    if (i >= A->m.nrows() || j >= A->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info {
        auto v = A->m.get(i, j);
        if (!v) return GrB_NO_VALUE;
        *x = *v;
        return GrB_SUCCESS;
    });
}
GrB_Info GxB_Matrix_isStoredElement(GrB_Matrix A, GrB_Index i, GrB_Index j) {
    if (!A) return GrB_NULL_POINTER;
    if (i >= A->m.nrows() || j >= A->m.ncols()) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info { return A->m.contains(i, j) ? GrB_SUCCESS : GrB_NO_VALUE; });
}

// ---- products and set algebra: the call forms matrix.rs issues --------------------------------------------------
// GrB_mxm over ANY_PAIR_BOOL (mod.rs:11162-11171): lmxm / rmxm in place (matrix.rs:930-968), the fresh-output products of
// delta_lmxm (:1343-1380) and its masked product C<!M, replace, structural> = A * B (:1382-1396, GrB_DESC_RSC)
GrB_Info GrB_mxm(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Matrix B,
                 GrB_Descriptor desc) {
    if (!C || !A || !B || !semiring) return GrB_NULL_POINTER;
    if (accum || semiring != GxB_ANY_PAIR_BOOL || C->m.type() != Type::Bool) return GrB_NOT_IMPLEMENTED;
    if (desc && (desc->t0 || desc->t1)) return GrB_NOT_IMPLEMENTED;
    if (!Mask && desc && (desc->complement || desc->structural)) Mask = nullptr;
    if (Mask && !(desc && desc->complement && desc->replace)) return GrB_NOT_IMPLEMENTED;   // only C<!M, replace> = A * B
    if (A->m.ncols() != B->m.nrows() || C->m.nrows() != A->m.nrows() || C->m.ncols() != B->m.ncols()) return GrB_DIMENSION_MISMATCH;
    return guarded([&]() -> GrB_Info {
        if (!Mask && same(C, A)) { C->m.lmxm(B->m); return GrB_SUCCESS; }
        if (!Mask && same(C, B)) { C->m.rmxm(A->m); return GrB_SUCCESS; }
        if (A->m.type() != Type::Bool) return GrB_NOT_IMPLEMENTED;      // the left operand is the frontier F (BOOL)
        Matrix P = A->m.dup();                                           // shares A's device snapshot
        P.lmxm(B->m);
        if (Mask) {
            Matrix out(*ctx(), Type::Bool, P.nrows(), P.ncols());
            out.select(Mask->m, P);                                      // P minus pattern(Mask)
            C->m = out;
        } else {
            C->m = P;
        }
        return GrB_SUCCESS;
    });
}

// C<Mask> = A (+) B: pattern union for BOOL (GxB_ANY_BOOL), B's value on a shared pair for UINT64 (GrB_SECOND_UINT64)
// (matrix.rs:852-874: no mask, or (mask, GrB_DESC_RC) — the fold of versioned_matrix.rs:892-938)
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_BinaryOp add, GrB_Matrix A,
                                      GrB_Matrix B, GrB_Descriptor desc) {
    if (!C || !A || !B || !add) return GrB_NULL_POINTER;
    if (accum) return GrB_NOT_IMPLEMENTED;
    if ((C->m.type() == Type::Bool && add != GxB_ANY_BOOL) || (C->m.type() == Type::UInt64 && add != GrB_SECOND_UINT64))
        return GrB_NOT_IMPLEMENTED;
    falkor::Descriptor d = falkor::Descriptor::None;
    if (Mask) {
        if (!desc_is(desc, true, false, true, false, false) && !desc_is(desc, true, true, true, false, false)) return GrB_NOT_IMPLEMENTED;
        d = falkor::Descriptor::RC;
    } else if (desc && (desc->t0 || desc->t1)) {
        return GrB_NOT_IMPLEMENTED;
    }
    return guarded([&]() -> GrB_Info {
        C->m.element_wise_add(Mask ? &Mask->m : nullptr, same(A, C) ? nullptr : &A->m, same(B, C) ? nullptr : &B->m, d);
        return GrB_SUCCESS;
    });
}

// C = pattern(A) & pattern(B) (matrix.rs:876-896; the semiring is ANY_PAIR: an iso-true / B-valued intersection)
GrB_Info GrB_Matrix_eWiseMult_Semiring(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A,
                                       GrB_Matrix B, GrB_Descriptor desc) {
    if (!C || !A || !B || !semiring) return GrB_NULL_POINTER;
    if (Mask || accum || semiring != GxB_ANY_PAIR_BOOL || (desc && (desc->t0 || desc->t1))) return GrB_NOT_IMPLEMENTED;
    return guarded([&]() -> GrB_Info {
        C->m.element_wise_multiply(same(A, C) ? nullptr : &A->m, same(B, C) ? nullptr : &B->m);
        return GrB_SUCCESS;
    });
}

// C<Mask> U= pattern(A) as all-true entries (GrB_Matrix_apply with accum GxB_ANY_BOOL and GxB_ONE_BOOL, matrix.rs:906-924)
GrB_Info GrB_Matrix_apply(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_UnaryOp op, GrB_Matrix A, GrB_Descriptor desc) {
    if (!C || !A || !op) return GrB_NULL_POINTER;
    if (accum != GxB_ANY_BOOL || op != GxB_ONE_BOOL) return GrB_NOT_IMPLEMENTED;
    falkor::Descriptor d = falkor::Descriptor::None;
    if (Mask) {
        if (!desc_is(desc, false, false, true, false, false) && !desc_is(desc, false, true, true, false, false)) return GrB_NOT_IMPLEMENTED;
        d = falkor::Descriptor::C;
    }
    return guarded([&]() -> GrB_Info { C->m.set_pattern(Mask ? &Mask->m : nullptr, A->m, d); return GrB_SUCCESS; });
}

// GrB_transpose (mod.rs:14013): the plain transpose into a fresh matrix (matrix.rs:633-662), and the wrapper's masked
// assignment C<!Mask, replace> = A written as a transpose of the transposed input (GrB_DESC_RCT0: remove_all / select,
// matrix.rs:824-845)
GrB_Info GrB_transpose(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, GrB_Matrix A, GrB_Descriptor desc) {
    if (!C || !A) return GrB_NULL_POINTER;
    if (accum) return GrB_NOT_IMPLEMENTED;
    if (!Mask && desc_is(desc, false, false, false, false, false)) {
        if (C->m.nrows() != A->m.ncols() || C->m.ncols() != A->m.nrows()) return GrB_DIMENSION_MISMATCH;
        if (C->m.type() != A->m.type()) return GrB_DOMAIN_MISMATCH;
        return guarded([&]() -> GrB_Info { C->m = A->m.transpose(); return GrB_SUCCESS; });
    }
    if (Mask && (desc_is(desc, true, false, true, true, false) || desc_is(desc, true, true, true, true, false))) {
        return guarded([&]() -> GrB_Info {
            if (same(A, C)) C->m.remove_all(Mask->m);
            else C->m.select(Mask->m, A->m);
            return GrB_SUCCESS;
        });
    }
    return GrB_NOT_IMPLEMENTED;
}

// ---- row iterator (mod.rs:14848-14906, driven by matrix.rs:1500-1605) -------------------------------------------
GrB_Info GxB_Iterator_new(GxB_Iterator* it) {
    if (!it) return GrB_NULL_POINTER;
    *it = new (std::nothrow) GB_Iterator_opaque();
    return *it ? GrB_SUCCESS : GrB_OUT_OF_MEMORY;
}
GrB_Info GxB_Iterator_free(GxB_Iterator* it) {
    if (it) { delete *it; *it = nullptr; }
    return GrB_SUCCESS;
}
GrB_Info GxB_rowIterator_attach(GxB_Iterator it, GrB_Matrix A, GrB_Descriptor) {
    if (!it || !A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info {
        A->m.wait();                                  // an iterator reads the materialised state
        it->m.reset(new Matrix(A->m));                // shares the handle
        it->nrows = A->m.nrows();
        it->row = it->nrows;
        it->w_lo = 1; it->w_hi = 0;
        it->buf.clear();
        it->pos = it->row_end = 0;
        return GrB_SUCCESS;
    });
}
GrB_Index GxB_rowIterator_kount(GxB_Iterator it) { return it ? it->nrows : 0; }
GrB_Info GxB_rowIterator_seekRow(GxB_Iterator it, GrB_Index row) {
    if (!it || !it->m) return GrB_NULL_POINTER;
    if (row >= it->nrows) { it->row = it->nrows; it->pos = it->row_end = 0; return GxB_EXHAUSTED; }
    return guarded([&]() -> GrB_Info { return it_at(it, row); });
}
GrB_Info GxB_rowIterator_nextRow(GxB_Iterator it) {
    if (!it || !it->m) return GrB_NULL_POINTER;
    if (it->row + 1 >= it->nrows) { it->row = it->nrows; it->pos = it->row_end = 0; return GxB_EXHAUSTED; }
    return guarded([&]() -> GrB_Info { return it_at(it, it->row + 1); });
}
GrB_Info GxB_rowIterator_nextCol(GxB_Iterator it) {
    if (!it) return GrB_NULL_POINTER;
    if (it->pos + 1 < it->row_end) { ++it->pos; return GrB_SUCCESS; }
    it->pos = it->row_end;
    return GrB_NO_VALUE;                              // end of the row
}
GrB_Index GxB_rowIterator_getRowIndex(GxB_Iterator it) { return it ? it->row : 0; }
GrB_Index GxB_rowIterator_getColIndex(GxB_Iterator it) {
    return (it && it->pos < it->row_end) ? it->buf[it->pos].col : 0;
}
uint64_t GxB_Iterator_get_UINT64(GxB_Iterator it) { return (it && it->pos < it->row_end) ? it->buf[it->pos].val : 0; }
bool GxB_Iterator_get_BOOL(GxB_Iterator it) { return it && it->pos < it->row_end && it->buf[it->pos].val != 0; }

}  // extern "C"
