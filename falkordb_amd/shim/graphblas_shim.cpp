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
#include "shim_internal.hpp"

namespace {
GB_Type_opaque t_bool{0, "GrB_BOOL", 1}, t_u64{1, "GrB_UINT64", 8}, t_u32{2, "GrB_UINT32", 4}, t_i32{3, "GrB_INT32", 4},
    t_i64{4, "GrB_INT64", 8}, t_i8{5, "GrB_INT8", 1}, t_u8{6, "GrB_UINT8", 1}, t_u16{7, "GrB_UINT16", 2}, t_i16{8, "GrB_INT16", 2},
    t_f32{9, "GrB_FP32", 4}, t_f64{10, "GrB_FP64", 8};     // (the types LAGraph's results come in: level / parent / centrality)
GB_Type_opaque* const all_types[] = {&t_bool, &t_u64, &t_u32, &t_i32, &t_i64, &t_i8, &t_u8, &t_u16, &t_i16, &t_f32, &t_f64};

// the allocator GxB_init was handed (matrix.rs:116-185 passes Redis'): arrays that change owner across the ABI — what
// GxB_Vector_unload / GxB_Vector_serialize give out, what GxB_Vector_load adopts — are allocated and released with it
void* (*g_malloc)(size_t) = nullptr;
void (*g_free)(void*) = nullptr;
void* shim_malloc(size_t n) { return g_malloc ? g_malloc(n ? n : 1) : malloc(n ? n : 1); }
void shim_free(void* p) { if (!p) return; if (g_free) g_free(p); else free(p); }

void vec_drop_array(GB_Vector_opaque* v) {
    if (v->pinned_owner) (void)fgpu_free(v->pinned_owner, v->data); else shim_free(v->data);
    v->data = nullptr; v->nbytes = 0; v->nstored = 0; v->pinned_owner = nullptr; v->absent = 0; v->stored_count = -1;
}
// entry i of a dense array as a signed / floating value (results of the engine are INT32 / INT64 / FP32; container
// arrays are the unsigned index types)
inline int64_t dense_i64(const GB_Vector_opaque* v, GrB_Index i) {
    const char* p = (const char*)v->data + (v->nstored == 1 ? 0 : i * v->type->size);
    switch (v->type->code) {
        case 3: { int32_t x; memcpy(&x, p, 4); return x; }
        case 4: { int64_t x; memcpy(&x, p, 8); return x; }
        case 2: { uint32_t x; memcpy(&x, p, 4); return x; }
        case 1: { uint64_t x; memcpy(&x, p, 8); return (int64_t)x; }
        case 5: return *(const int8_t*)p;
        case 6: case 0: return *(const uint8_t*)p;
        case 7: { uint16_t x; memcpy(&x, p, 2); return x; }
        case 8: { int16_t x; memcpy(&x, p, 2); return x; }
        case 9: { float x; memcpy(&x, p, 4); return (int64_t)x; }
        case 10: { double x; memcpy(&x, p, 8); return (int64_t)x; }
    }
    return 0;
}
inline double dense_f64(const GB_Vector_opaque* v, GrB_Index i) {
    const char* p = (const char*)v->data + (v->nstored == 1 ? 0 : i * v->type->size);
    if (v->type->code == 9) { float x; memcpy(&x, p, 4); return x; }
    if (v->type->code == 10) { double x; memcpy(&x, p, 8); return x; }
    if (v->type->code == 1) { uint64_t x; memcpy(&x, p, 8); return (double)x; }
    return (double)dense_i64(v, i);
}
inline bool dense_present(const GB_Vector_opaque* v, GrB_Index i) {
    if (!v->absent) return true;
    if (v->type->code == 9 || v->type->code == 10) return v->absent == 1 ? dense_f64(v, i) >= 0 : dense_f64(v, i) != 0;
    const int64_t x = dense_i64(v, i);
    return v->absent == 1 ? x >= 0 : x != 0;
}
uint64_t dense_count(GB_Vector_opaque* v) {
    if (!v->absent) return v->n;
    if (v->stored_count < 0) {
        int64_t c = 0;
        for (GrB_Index i = 0; i < v->n; ++i) c += dense_present(v, i) ? 1 : 0;
        v->stored_count = c;
    }
    return (uint64_t)v->stored_count;
}
GB_Vector_opaque* vec_new(GB_Type_opaque* t, GrB_Index n) {
    GB_Vector_opaque* v = new GB_Vector_opaque();
    v->type = t; v->n = n;
    return v;
}
void vec_set_array(GB_Vector_opaque* v, GB_Type_opaque* t, const void* bytes, uint64_t nbytes, uint64_t n_entries) {
    vec_drop_array(v);
    v->s.clear();
    v->type = t; v->n = n_entries; v->nstored = t->size ? nbytes / t->size : 0; v->nbytes = nbytes;
    if (nbytes) { v->data = shim_malloc(nbytes); memcpy(v->data, bytes, nbytes); }
}
GB_Type_opaque* type_by_name(const char* name) {
    for (GB_Type_opaque* t : all_types)
        if (!strcmp(t->name, name)) return t;
    // the C type names GraphBLAS also accepts (GxB_Type_from_name)
    static const struct { const char* c; GB_Type_opaque* t; } alias[] = {{"bool", &t_bool}, {"uint64_t", &t_u64}, {"uint32_t", &t_u32},
        {"int32_t", &t_i32}, {"int64_t", &t_i64}, {"int8_t", &t_i8}, {"uint8_t", &t_u8}, {"uint16_t", &t_u16}, {"int16_t", &t_i16},
        {"float", &t_f32}, {"double", &t_f64}};
    for (auto& a : alias)
        if (!strcmp(a.c, name)) return a.t;
    return nullptr;
}
// the five vectors of a container in the Vector<bool> wire form (vector.rs:241-309) — the form serialize.cpp parses
void put_vector(falkor::ByteWriter& w, const GB_Vector_opaque* v) {
    static const char none[] = "GrB_INT8";
    const char* tn = (v && v->type) ? v->type->name : none;
    w.write_buffer(v ? v->data : nullptr, v ? v->nbytes : 0);
    w.write_buffer(tn, strlen(tn) + 1);
    w.write_unsigned(v ? v->n : 0);
    w.write_unsigned(v ? v->nbytes : 0);
    w.write_signed(0);
}
void take_vector(falkor::ByteReader& r, GB_Vector_opaque* v) {
    std::vector<uint8_t> bytes = r.read_buffer(), name = r.read_buffer();
    const uint64_t n = r.read_unsigned();
    (void)r.read_unsigned();
    (void)r.read_signed();
    GB_Type_opaque* t = name.empty() ? nullptr : type_by_name((const char*)name.data());
    if (!t) throw falkor::GrbError(FGPU_INVALID, "container vector of an unknown type");
    vec_set_array(v, t, bytes.data(), bytes.size(), n);
}
constexpr char BLOB_IDS[8] = {'F', 'G', 'I', 'D', 'L', 'S', 'T', '1'};   // = serialize.cpp PLAIN_MAGIC: BOOL vector, the set indices
constexpr char BLOB_U64[8] = {'F', 'G', 'V', 'E', 'C', 'U', '6', '4'};   // UINT64 vector: length, count, (index, value) pairs
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

namespace fgshim {
falkor::Context* context() { return ctx(); }
GB_Type_opaque* type_int32() { return &t_i32; }
GB_Type_opaque* type_int64() { return &t_i64; }
GB_Type_opaque* type_fp32() { return &t_f32; }
GB_Vector_opaque* vector_over_pinned(GB_Type_opaque* type, GrB_Index n, void* pinned, int absent) {
    GB_Vector_opaque* v = vec_new(type, n);
    v->data = pinned; v->nbytes = n * type->size; v->nstored = n; v->absent = absent;
    v->pinned_owner = ctx()->raw();
    return v;
}
}  // namespace fgshim

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
GrB_Type GrB_INT32 = &t_i32;       // types of the vectors LAGraph hands back (lagraph_shim.cpp)
GrB_Type GrB_INT64 = &t_i64;
GrB_Type GrB_FP32 = &t_f32;
GrB_Type GrB_FP64 = &t_f64;

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
    (void)cal; (void)rea;
    if (mode != 0 && mode != 1) return GrB_INVALID_VALUE;
    std::lock_guard<std::mutex> g(g_mu);
    if (g_ctx) return GrB_INVALID_VALUE;          // initialised twice
    try {
        int dev = 0;
        if (const char* e = getenv("FGPU_DEVICE")) dev = atoi(e);
        g_ctx.reset(new falkor::Context(dev));   // throws without a HIP device: no CPU fallback behind this ABI either
        g_malloc = mal; g_free = fre;            // ownership-changing arrays (GxB_Vector_load / _unload / _serialize) use these
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
uint64_t GxB_Iterator_get_UINT64(GxB_Iterator it) {
    if (it && it->vec) return it->vit != it->vec->s.end() ? it->vit->second : 0;
    return (it && it->pos < it->row_end) ? it->buf[it->pos].val : 0;
}
bool GxB_Iterator_get_BOOL(GxB_Iterator it) { return it && it->pos < it->row_end && it->buf[it->pos].val != 0; }


// ---- what matrix.rs imports beside the traversal path (matrix.rs:79-102) ------------------------------------------------
GrB_Info GrB_Matrix_build_BOOL(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const bool* X, GrB_Index nvals, GrB_BinaryOp dup) {
    if (!C || (nvals && (!I || !J || !X))) return GrB_NULL_POINTER;
    if (C->m.type() != Type::Bool) return GrB_DOMAIN_MISMATCH;
    if (dup && dup != GxB_ANY_BOOL) return GrB_NOT_IMPLEMENTED;
    for (GrB_Index k = 0; k < nvals; ++k)
        if (!X[k]) return GrB_NOT_IMPLEMENTED;     // a stored FALSE: the graph's boolean matrices are patterns (matrix.rs:1709-1775)
    return guarded([&]() -> GrB_Info {
        if (C->m.nvals()) return GrB_OUTPUT_NOT_EMPTY;
        C->m.build(std::vector<uint64_t>(I, I + nvals), std::vector<uint64_t>(J, J + nvals));
        return GrB_SUCCESS;
    });
}
// every BOOL matrix of this engine is a pattern (iso TRUE); UINT64 matrices carry a value per entry
GrB_Info GxB_Matrix_iso(bool* iso, GrB_Matrix A) {
    if (!iso || !A) return GrB_NULL_POINTER;
    *iso = A->m.type() == Type::Bool;
    return GrB_SUCCESS;
}
// the device bytes of the wait()ed snapshot: row pointers + column ids (+ values), 32-bit ids (fgpu.h)
GrB_Info GxB_Matrix_memoryUsage(size_t* size, GrB_Matrix A) {
    if (!size || !A) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info {
        const uint64_t nv = A->m.nvals();
        *size = sizeof(GB_Matrix_opaque) + (size_t)(A->m.nrows() + 1) * 4 + (size_t)nv * (A->m.type() == Type::UInt64 ? 12 : 4);
        return GrB_SUCCESS;
    });
}
GrB_Info GxB_Matrix_fprint(GrB_Matrix A, const char* name, int pr, FILE* f) {
    if (!A) return GrB_NULL_POINTER;
    if (pr <= 0) return GrB_SUCCESS;               // GxB_SILENT
    return guarded([&]() -> GrB_Info {
        FILE* o = f ? f : stdout;
        const uint64_t nv = A->m.nvals();
        fprintf(o, "\n  %llu x %llu GraphBLAS %s matrix, sparse by row\n  %s, %llu entr%s\n", (unsigned long long)A->m.nrows(),
                (unsigned long long)A->m.ncols(), A->m.type() == Type::UInt64 ? "uint64_t" : "bool", name ? name : "", (unsigned long long)nv,
                nv == 1 ? "y" : "ies");
        if (pr >= 2 && nv) {                         // GxB_SHORT: the first 30 entries; GxB_COMPLETE: all of them
            uint64_t shown = 0;
            const uint64_t cap = pr >= 3 ? nv : 30, nr = A->m.nrows();
            for (uint64_t lo = 0; lo < nr && shown < cap; lo += 4096) {
                for (auto& e : A->m.iter(lo, lo + 4095 < nr ? lo + 4095 : nr - 1)) {
                    if (shown++ >= cap) break;
                    fprintf(o, "    (%llu,%llu)   %llu\n", (unsigned long long)e.row, (unsigned long long)e.col, (unsigned long long)e.val);
                }
            }
            if (shown < nv) fprintf(o, "    ...\n");
        }
        return GrB_SUCCESS;
    });
}

// ---- GrB_Vector (vector.rs:43-60) -------------------------------------------------------------------------------------
GrB_Info GrB_Vector_new(GrB_Vector* v, GrB_Type type, GrB_Index n) {
    if (!v || !type) return GrB_NULL_POINTER;
    *v = vec_new(type, n);
    return GrB_SUCCESS;
}
GrB_Info GrB_Vector_free(GrB_Vector* v) {
    if (v && *v) { vec_drop_array(*v); delete *v; *v = nullptr; }
    return GrB_SUCCESS;
}
GrB_Info GrB_Vector_clear(GrB_Vector v) { if (!v) return GrB_NULL_POINTER; v->s.clear(); vec_drop_array(v); return GrB_SUCCESS; }
GrB_Info GrB_Vector_size(GrB_Index* n, GrB_Vector v) { if (!n || !v) return GrB_NULL_POINTER; *n = v->n; return GrB_SUCCESS; }
GrB_Info GrB_Vector_nvals(GrB_Index* n, GrB_Vector v) {
    if (!n || !v) return GrB_NULL_POINTER;
    *n = v->data ? dense_count(v) : v->s.size();
    return GrB_SUCCESS;
}
GrB_Info GrB_Vector_resize(GrB_Vector v, GrB_Index n) {
    if (!v) return GrB_NULL_POINTER;
    if (v->data) return GrB_NOT_IMPLEMENTED;       // a loaded array is moved, not resized (the wrapper never does)
    v->s.erase(v->s.lower_bound(n), v->s.end());
    v->n = n;
    return GrB_SUCCESS;
}
GrB_Info GrB_Vector_wait(GrB_Vector v, int) { return v ? GrB_SUCCESS : GrB_NULL_POINTER; }
static GrB_Info vec_set(GrB_Vector v, uint64_t x, GrB_Index i) {
    if (!v) return GrB_NULL_POINTER;
    if (v->data) return GrB_NOT_IMPLEMENTED;
    if (i >= v->n) return GrB_INVALID_INDEX;
    return guarded([&]() -> GrB_Info { v->s[i] = x; return GrB_SUCCESS; });
}
GrB_Info GrB_Vector_setElement_BOOL(GrB_Vector v, bool x, GrB_Index i) { return vec_set(v, x ? 1 : 0, i); }
GrB_Info GrB_Vector_setElement_UINT64(GrB_Vector v, uint64_t x, GrB_Index i) { return vec_set(v, x, i); }
GrB_Info GrB_Vector_removeElement(GrB_Vector v, GrB_Index i) {
    if (!v) return GrB_NULL_POINTER;
    if (v->data) return GrB_NOT_IMPLEMENTED;
    if (i >= v->n) return GrB_INVALID_INDEX;
    v->s.erase(i);
    return GrB_SUCCESS;
}
// GrB_Vector_extractTuples_INT64 / _FP64 (mod.rs; algo_procedures.rs:415-447 reads level / parent / centrality with them):
// entries in ascending index order, values typecast to the requested type, *nvals in = room, out = entries written
}  // extern "C"
template <typename T, typename Get>
static GrB_Info vec_extract(GrB_Index* I, T* X, GrB_Index* nvals, GrB_Vector v, Get get_dense) {
    if (!nvals || !v) return GrB_NULL_POINTER;
    return guarded([&]() -> GrB_Info {
        const uint64_t have = v->data ? dense_count(v) : v->s.size();
        if (*nvals < have) return GrB_INSUFFICIENT_SPACE;
        uint64_t k = 0;
        if (v->data) {
            for (GrB_Index i = 0; i < v->n; ++i) {
                if (!dense_present(v, i)) continue;
                if (I) I[k] = i;
                if (X) X[k] = get_dense(v, i);
                ++k;
            }
        } else {
            for (auto& kv : v->s) {
                if (I) I[k] = kv.first;
                if (X) X[k] = (T)kv.second;
                ++k;
            }
        }
        *nvals = k;
        return GrB_SUCCESS;
    });
}
extern "C" {
GrB_Info GrB_Vector_extractTuples_INT64(GrB_Index* I, int64_t* X, GrB_Index* nvals, GrB_Vector v) {
    return vec_extract<int64_t>(I, X, nvals, v, dense_i64);
}
GrB_Info GrB_Vector_extractTuples_FP64(GrB_Index* I, double* X, GrB_Index* nvals, GrB_Vector v) {
    return vec_extract<double>(I, X, nvals, v, dense_f64);
}
GrB_Info GrB_Type_get_String(GrB_Type type, char* value, int field) {
    if (!type || !value) return GrB_NULL_POINTER;
    if (field != 10 /* GrB_NAME (mod.rs GxB_Option_Field) */) return GrB_INVALID_VALUE;
    strcpy(value, type->name);                     // (the caller's buffer holds GxB_MAX_NAME_LEN = 128 bytes, vector.rs:267)
    return GrB_SUCCESS;
}
GrB_Info GxB_Type_from_name(GrB_Type* type, const char* name) {
    if (!type || !name) return GrB_NULL_POINTER;
    *type = type_by_name(name);
    return *type ? GrB_SUCCESS : GrB_INVALID_VALUE;
}
// vector iterator (vector.rs:525-606): attach, seek(0), then getIndex / get_UINT64 / next until GxB_EXHAUSTED
GrB_Info GxB_Vector_Iterator_attach(GxB_Iterator it, GrB_Vector v, GrB_Descriptor) {
    if (!it || !v) return GrB_NULL_POINTER;
    if (v->data) return GrB_NOT_IMPLEMENTED;
    it->vec = v;
    it->vit = v->s.begin();
    return GrB_SUCCESS;
}
GrB_Info GxB_Vector_Iterator_seek(GxB_Iterator it, GrB_Index p) {
    if (!it || !it->vec) return GrB_NULL_POINTER;
    it->vit = it->vec->s.begin();
    for (GrB_Index k = 0; k < p && it->vit != it->vec->s.end(); ++k) ++it->vit;   // p-th stored entry
    return it->vit == it->vec->s.end() ? GxB_EXHAUSTED : GrB_SUCCESS;
}
GrB_Info GxB_Vector_Iterator_next(GxB_Iterator it) {
    if (!it || !it->vec) return GrB_NULL_POINTER;
    if (it->vit != it->vec->s.end()) ++it->vit;
    return it->vit == it->vec->s.end() ? GxB_EXHAUSTED : GrB_SUCCESS;
}
GrB_Index GxB_Vector_Iterator_getIndex(GxB_Iterator it) { return (it && it->vec && it->vit != it->vec->s.end()) ? it->vit->first : 0; }
// GxB_Vector_load / _unload (vector.rs:241-420): the array changes owner; *X is NULL afterwards on load, the vector is
// empty (length 0) afterwards on unload
GrB_Info GxB_Vector_load(GrB_Vector v, void** X, GrB_Type type, uint64_t n, uint64_t X_memsize, int handling, GrB_Descriptor) {
    if (!v || !X || !type) return GrB_NULL_POINTER;
    if (n && X_memsize / type->size < 1) return GrB_INVALID_VALUE;      // (an iso array holds one entry for n of them)
    vec_drop_array(v);
    v->s.clear();
    v->type = type; v->n = n; v->data = *X; v->nbytes = X_memsize; v->nstored = X_memsize / type->size; v->handling = handling;
    *X = nullptr;
    return GrB_SUCCESS;
}
GrB_Info GxB_Vector_unload(GrB_Vector v, void** X, GrB_Type* type, uint64_t* n, uint64_t* X_memsize, int* handling, GrB_Descriptor) {
    if (!v || !X || !type || !n || !X_memsize || !handling) return GrB_NULL_POINTER;
    if (!v->data && !v->s.empty()) {
        // a sparse vector can only be unloaded when it is full (GraphBLAS' rule): materialise it then
        if (v->s.size() != v->n) return GrB_INVALID_VALUE;
        const size_t sz = v->type->size;
        char* a = (char*)shim_malloc(v->n * sz);
        GrB_Index k = 0;
        for (auto& kv : v->s) { memcpy(a + (k++) * sz, &kv.second, sz); }     // little-endian: the low bytes are the value
        v->s.clear();
        v->data = a; v->nbytes = v->n * sz; v->nstored = v->n;
    }
    if (v->pinned_owner) {                       // an engine result: the caller gets an array of ITS allocator, the pinned block goes back
        if (v->absent) return GrB_INVALID_VALUE;  // (not full: same rule as above)
        void* a = shim_malloc(v->nbytes);
        if (!a) return GrB_OUT_OF_MEMORY;
        memcpy(a, v->data, v->nbytes);
        const uint64_t nb = v->nbytes, ns = v->nstored;
        vec_drop_array(v);
        v->data = a; v->nbytes = nb; v->nstored = ns;
    }
    *X = v->data; *type = v->type; *n = v->n; *X_memsize = v->nbytes; *handling = v->handling;
    v->data = nullptr; v->nbytes = 0; v->nstored = 0; v->n = 0;
    return GrB_SUCCESS;
}
// GxB_Vector_serialize / _deserialize (vector.rs:150-239): SuiteSparse's own blob format is not restated (it belongs to
// the un-vendored library, and no fixture of it exists in the reference tree — DESIGN.md §9); the blob written here is the
// plain little-endian list serialize.cpp's Tensor codec writes, so payloads of the two tiers read each other.  The blob is
// allocated with the GxB_init allocator: the wrapper releases it itself (vector.rs:166-167).
GrB_Info GxB_Vector_serialize(void** blob, GrB_Index* blob_size, GrB_Vector u, GrB_Descriptor) {
    if (!blob || !blob_size || !u) return GrB_NULL_POINTER;
    if (u->data) return GrB_NOT_IMPLEMENTED;
    const bool ids = u->type == GrB_BOOL;
    const uint64_t cnt = u->s.size();
    const size_t sz = ids ? 16 + 8 * cnt : 24 + 16 * cnt;
    uint8_t* b = (uint8_t*)shim_malloc(sz);
    if (!b) return GrB_OUT_OF_MEMORY;
    memcpy(b, ids ? BLOB_IDS : BLOB_U64, 8);
    uint8_t* q = b + 8;
    if (!ids) { memcpy(q, &u->n, 8); q += 8; }
    memcpy(q, &cnt, 8); q += 8;
    for (auto& kv : u->s) {
        if (ids && !kv.second) { shim_free(b); return GrB_NOT_IMPLEMENTED; }   // a stored FALSE has no place in an id list
        memcpy(q, &kv.first, 8); q += 8;
        if (!ids) { memcpy(q, &kv.second, 8); q += 8; }
    }
    *blob = b; *blob_size = sz;
    return GrB_SUCCESS;
}
GrB_Info GxB_Vector_deserialize(GrB_Vector* w, GrB_Type type, const void* blob, GrB_Index blob_size, GrB_Descriptor) {
    if (!w || !blob) return GrB_NULL_POINTER;
    const uint8_t* b = (const uint8_t*)blob;
    if (blob_size < 16) return GrB_INVALID_OBJECT;
    const bool ids = !memcmp(b, BLOB_IDS, 8);
    if (!ids && memcmp(b, BLOB_U64, 8)) return GrB_INVALID_OBJECT;             // e.g. a real GxB_Vector_serialize blob
    if (type && type != (ids ? GrB_BOOL : GrB_UINT64)) return GrB_DOMAIN_MISMATCH;
    return guarded([&]() -> GrB_Info {
        uint64_t n = (1ull << 60) - 1, cnt = 0;                                 // GrB_INDEX_MAX: the length of an id-list vector (tensor.rs:1115)
        const uint8_t* q = b + 8;
        if (!ids) { if (blob_size < 24) return GrB_INVALID_OBJECT; memcpy(&n, q, 8); q += 8; }
        memcpy(&cnt, q, 8); q += 8;
        if ((blob_size - (size_t)(q - b)) / (ids ? 8 : 16) < cnt) return GrB_INVALID_OBJECT;
        std::unique_ptr<GB_Vector_opaque> v(vec_new(ids ? GrB_BOOL : GrB_UINT64, n));
        for (uint64_t k = 0; k < cnt; ++k) {
            uint64_t i = 0, x = 1;
            memcpy(&i, q, 8); q += 8;
            if (!ids) { memcpy(&x, q, 8); q += 8; }
            if (i >= n) return GrB_INVALID_OBJECT;
            v->s[i] = x;
        }
        *w = v.release();
        return GrB_SUCCESS;
    });
}

// ---- GxB_Container (matrix.rs:428-546: Decode<19> / Encode<19> of a Matrix) ------------------------------------------------
GrB_Info GxB_Container_new(GxB_Container* c) {
    if (!c) return GrB_NULL_POINTER;
    GxB_Container_struct* k = (GxB_Container_struct*)calloc(1, sizeof(GxB_Container_struct));
    if (!k) return GrB_OUT_OF_MEMORY;
    k->p = vec_new(&t_u32, 0); k->h = vec_new(&t_u32, 0); k->b = vec_new(&t_i8, 0); k->i = vec_new(&t_u32, 0); k->x = vec_new(&t_bool, 0);
    k->format = 2 /* GxB_SPARSE */; k->orientation = 0 /* GrB_ROWMAJOR */; k->iso = false; k->jumbled = false;
    k->nrows_nonempty = k->ncols_nonempty = -1;
    *c = k;
    return GrB_SUCCESS;
}
GrB_Info GxB_Container_free(GxB_Container* c) {
    if (!c || !*c) return GrB_SUCCESS;
    GxB_Container_struct* k = *c;
    for (GrB_Vector* v : {&k->p, &k->h, &k->b, &k->i, &k->x}) GrB_Vector_free(v);   // (NULL fields — the decoder nullifies and refills them — are skipped)
    if (k->Y) GrB_Matrix_free(&k->Y);
    free(k);
    *c = nullptr;
    return GrB_SUCCESS;
}
// A -> container: the wait()ed CSR moves into the container's vectors (x iso BOOL / UINT64, h, p, i, b) and A is left
// without entries until a load puts content back (the encoder reloads the same container, matrix.rs:527-528)
GrB_Info GxB_unload_Matrix_into_Container(GrB_Matrix A, GxB_Container C, GrB_Descriptor) {
    if (!A || !C) return GrB_NULL_POINTER;
    SHIM_REQUIRE_INIT();
    return guarded([&]() -> GrB_Info {
        falkor::ByteWriter w;
        A->m.encode(w);                                               // serialize.cpp: struct bytes + x, h, p, i, b
        falkor::ByteReader r(w.buf.data(), w.buf.size());
        std::vector<uint8_t> st = r.read_buffer();
        GrB_Vector keep[5] = {C->p, C->h, C->b, C->i, C->x};
        GrB_Matrix y = C->Y;
        memcpy(C, st.data(), sizeof(GxB_Container_struct));
        C->p = keep[0]; C->h = keep[1]; C->b = keep[2]; C->i = keep[3]; C->x = keep[4]; C->Y = y;
        for (GrB_Vector* v : {&C->x, &C->h, &C->p, &C->i, &C->b}) {
            if (!*v) *v = vec_new(&t_i8, 0);
            take_vector(r, *v);
        }
        A->m = Matrix(*ctx(), A->m.type(), A->m.nrows(), A->m.ncols());
        return GrB_SUCCESS;
    });
}
// container -> A: the arrays go to the device as they are (fgpu_mat_from_csr behind Matrix::decode); A takes the type of the
// value vector (the decoder creates A as a 0 x 0 BOOL matrix first, matrix.rs:470-478); the container's vectors are left empty
GrB_Info GxB_load_Matrix_from_Container(GrB_Matrix A, GxB_Container C, GrB_Descriptor) {
    if (!A || !C) return GrB_NULL_POINTER;
    SHIM_REQUIRE_INIT();
    return guarded([&]() -> GrB_Info {
        falkor::ByteWriter w;
        GxB_Container_struct raw = *C;
        raw.p = raw.h = raw.b = raw.i = raw.x = nullptr; raw.Y = nullptr;          // (pointers are not content)
        w.write_buffer(&raw, sizeof(raw));
        for (GrB_Vector v : {C->x, C->h, C->p, C->i, C->b}) put_vector(w, v);
        falkor::ByteReader r(w.buf.data(), w.buf.size());
        A->m = Matrix::decode(*ctx(), r);
        for (GrB_Vector v : {C->x, C->h, C->p, C->i, C->b})
            if (v) { vec_drop_array(v); v->s.clear(); v->n = 0; }
        return GrB_SUCCESS;
    });
}

}  // extern "C"
