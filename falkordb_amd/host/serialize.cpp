// serialize.cpp — the v19 on-disk form of a Matrix<T> (graphblas/matrix.rs:428-546: `Encode<19>` / `Decode<19>`):
// the 608 raw bytes of GxB_Container_struct (bindgen layout, graphblas/mod.rs:14165-14188) followed by its five
// vectors x, h, p, i, b, each in Vector<bool>'s unload-to-array form (graphblas/vector.rs:241-309): buffer(array
// bytes), buffer(type name + NUL), unsigned(n_entries), unsigned(n_bytes), signed(handling).  This is what a live
// FalkorDB (or its RDB file) hands over for a committed base — arrays that go to the device as they are
// (fgpu_mat_from_csr), no tuple extraction.
//
// The byte stream itself (how unsigned / signed / buffer are framed) belongs to the Redis module API in the reference
// (Reader / Writer traits); here: unsigned and signed = 8 bytes little-endian, buffer = unsigned length + bytes.
// Tensor's multi-edge section (tensor.rs:1100-1128) uses GxB_Vector_serialize blobs — GraphBLAS' own compressed
// format — and is not decoded here.
#include <algorithm>
#include <cstring>
#include <map>

#include "host.hpp"

namespace falkor {

// GxB_Container_struct offsets (graphblas/mod.rs:14165-14188; size 608, align 8)
constexpr size_t C_NROWS = 0, C_NCOLS = 8, C_NROWS_NONEMPTY = 16, C_NCOLS_NONEMPTY = 24, C_NVALS = 32, C_FORMAT = 128,
                 C_ORIENTATION = 132, C_ISO = 448, C_JUMBLED = 449, C_SIZE = 608;
constexpr int GXB_HYPERSPARSE = 1, GXB_SPARSE = 2, GXB_BITMAP = 4, GXB_FULL = 8;   // graphblas/mod.rs:159-162
constexpr int GRB_ROWMAJOR = 0;                                                      // graphblas/mod.rs:3006-3007

void ByteWriter::write_unsigned(u64 v) { for (int k = 0; k < 8; ++k) buf.push_back((uint8_t)(v >> (8 * k))); }
void ByteWriter::write_signed(int64_t v) { write_unsigned((u64)v); }
void ByteWriter::write_buffer(const void* p, size_t n) {
    write_unsigned(n);
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
}
u64 ByteReader::read_unsigned() {
    if (pos + 8 > n) throw GrbError(FGPU_INVALID, "decode: truncated stream");
    u64 v = 0;
    for (int k = 0; k < 8; ++k) v |= (u64)p[pos + k] << (8 * k);
    pos += 8;
    return v;
}
int64_t ByteReader::read_signed() { return (int64_t)read_unsigned(); }
std::vector<uint8_t> ByteReader::read_buffer() {
    const u64 len = read_unsigned();
    if (len > n - pos) throw GrbError(FGPU_INVALID, "decode: buffer length past the end of the stream");
    std::vector<uint8_t> out(p + pos, p + pos + len);
    pos += len;
    return out;
}

namespace {

struct RawVector {            // Vector<bool>::decode (vector.rs:311-420)
    std::vector<uint8_t> bytes;
    std::string type;
    u64 n_entries = 0;
};

RawVector read_vector(ByteReader& r) {
    RawVector v;
    v.bytes = r.read_buffer();
    std::vector<uint8_t> name = r.read_buffer();
    v.n_entries = r.read_unsigned();
    const u64 n_bytes = r.read_unsigned();
    (void)r.read_signed();   // handling: ownership hint of GxB_Vector_load, meaningless off-process
    // the same validation the reference applies to GRAPH.RESTORE payloads (vector.rs:318-345)
    if (n_bytes != v.bytes.size()) throw GrbError(FGPU_INVALID, "Vector decode: declared byte length does not match buffer length");
    if (name.empty() || name.back() != 0) throw GrbError(FGPU_INVALID, "Vector decode: type name is not NUL-terminated");
    for (size_t k = 0; k + 1 < name.size(); ++k)
        if (name[k] == 0) throw GrbError(FGPU_INVALID, "Vector decode: type name is not NUL-terminated");
    v.type.assign((const char*)name.data(), name.size() - 1);
    return v;
}

void write_vector(ByteWriter& w, const void* data, size_t bytes, const char* type, u64 n_entries) {
    w.write_buffer(data, bytes);
    w.write_buffer(type, strlen(type) + 1);
    w.write_unsigned(n_entries);
    w.write_unsigned(bytes);
    w.write_signed(0);
}

int index_bits(const RawVector& v, const char* what) {
    if (v.n_entries == 0) return 64;
    if (v.type == "GrB_UINT32" || v.type == "GrB_INT32") {
        if (v.bytes.size() / 4 < v.n_entries) throw GrbError(FGPU_INVALID, std::string("container: short ") + what);
        return 32;
    }
    if (v.type == "GrB_UINT64" || v.type == "GrB_INT64") {
        if (v.bytes.size() / 8 < v.n_entries) throw GrbError(FGPU_INVALID, std::string("container: short ") + what);
        return 64;
    }
    throw GrbError(FGPU_INVALID, std::string("container: ") + what + " has type " + v.type);
}

u64 index_at(const RawVector& v, int bits, u64 k) {
    if (bits == 32) { uint32_t x; memcpy(&x, v.bytes.data() + 4 * k, 4); return x; }
    u64 x; memcpy(&x, v.bytes.data() + 8 * k, 8); return x;
}

template <typename T> T field(const std::vector<uint8_t>& s, size_t off) { T x; memcpy(&x, s.data() + off, sizeof(T)); return x; }

}  // namespace

ContainerData parse_container(ByteReader& r) {
    ContainerData c;
    std::vector<uint8_t> st = r.read_buffer();
    if (st.size() < C_SIZE) throw GrbError(FGPU_INVALID, "container buffer too small");        // matrix.rs:433-439
    c.nrows = field<u64>(st, C_NROWS);
    c.ncols = field<u64>(st, C_NCOLS);
    c.nvals = field<u64>(st, C_NVALS);
    c.format = field<int32_t>(st, C_FORMAT);
    c.orientation = field<int32_t>(st, C_ORIENTATION);
    c.iso = st[C_ISO] != 0;
    c.jumbled = st[C_JUMBLED] != 0;
    RawVector x = read_vector(r), h = read_vector(r), p = read_vector(r), i = read_vector(r), b = read_vector(r);
    (void)b;
    if (c.format != GXB_SPARSE && c.format != GXB_HYPERSPARSE)
        throw GrbError(FGPU_INVALID, "container: only sparse / hypersparse matrices are stored by the graph "
                                     "(pin_sparse, matrix.rs:405-426); got format " + std::to_string(c.format));
    if (c.orientation != GRB_ROWMAJOR) throw GrbError(FGPU_INVALID, "container: column-major matrix");
    if (c.jumbled) throw GrbError(FGPU_INVALID, "container: jumbled rows (the encoder writes wait()ed matrices)");
    // sizes come from an untrusted GRAPH.RESTORE-style payload: bound them by the device format (32-bit ids / row
    // pointers, fgpu.h) BEFORE anything is allocated from them, and compare lengths by division, never by a product
    if (c.nrows >= 0xFFFFFFFFull || c.ncols >= 0xFFFFFFFFull || c.nvals >= 0xFFFFFFFFull)
        throw GrbError(FGPU_INVALID, "container: dims / nvals exceed the 32-bit device format");
    const int pb = index_bits(p, "p"), ib = index_bits(i, "i"), hb = index_bits(h, "h");
    if (p.n_entries == 0) {                      // an empty matrix may come with an empty pointer vector
        if (c.nvals) throw GrbError(FGPU_INVALID, "container: nvals without row pointers");
        c.valued = x.type == "GrB_UINT64";       // an empty Matrix<u64> stays a Matrix<u64> (Tensor::decode checks the type)
        c.hyper = c.format == GXB_HYPERSPARSE;
        if (!c.hyper) c.p.assign(c.nrows + 1, 0);
        else c.p.assign(1, 0);
        return c;
    }
    const u64 nvec = p.n_entries - 1;
    if (nvec > c.nrows) throw GrbError(FGPU_INVALID, "container: more stored vectors than rows");
    c.hyper = c.format == GXB_HYPERSPARSE;
    if (c.hyper && h.n_entries < nvec) throw GrbError(FGPU_INVALID, "container: hyper list shorter than the pointer vector");
    if (!c.hyper && nvec != c.nrows) throw GrbError(FGPU_INVALID, "container: sparse matrix with nvec != nrows");
    c.p.resize(nvec + 1);
    for (u64 k = 0; k <= nvec; ++k) c.p[k] = index_at(p, pb, k);
    if (c.p[0] != 0 || c.p[nvec] != c.nvals) throw GrbError(FGPU_INVALID, "container: row pointers do not span nvals");
    for (u64 k = 0; k < nvec; ++k)
        if (c.p[k] > c.p[k + 1]) throw GrbError(FGPU_INVALID, "container: row pointers decrease");
    if (i.n_entries < c.nvals) throw GrbError(FGPU_INVALID, "container: index vector shorter than nvals");
    c.i.resize(c.nvals);
    for (u64 k = 0; k < c.nvals; ++k) {
        c.i[k] = index_at(i, ib, k);
        if (c.i[k] >= c.ncols) throw GrbError(FGPU_OUT_OF_BOUNDS, "container: column index out of range");
    }
    if (c.hyper) {
        c.h.resize(nvec);
        for (u64 k = 0; k < nvec; ++k) {
            c.h[k] = index_at(h, hb, k);
            if (c.h[k] >= c.nrows || (k && c.h[k] <= c.h[k - 1])) throw GrbError(FGPU_INVALID, "container: hyper list not ascending / in range");
        }
    }
    // values: GrB_BOOL (iso true for the graph's boolean matrices, matrix.rs:1709-1775) or GrB_UINT64
    if (x.type == "GrB_UINT64") {
        c.valued = true;
        const u64 need = c.iso ? (c.nvals ? 1 : 0) : c.nvals;
        if (x.bytes.size() / 8 < need) throw GrbError(FGPU_INVALID, "container: value vector too short");
        c.x.resize(c.nvals);
        for (u64 k = 0; k < c.nvals; ++k) memcpy(&c.x[k], x.bytes.data() + 8 * (c.iso ? 0 : k), 8);
    } else if (x.type == "GrB_BOOL" || x.n_entries == 0) {
        c.valued = false;
    } else {
        throw GrbError(FGPU_INVALID, "container: value type " + x.type + " (the graph stores GrB_BOOL and GrB_UINT64)");
    }
    return c;
}

void write_container(ByteWriter& w, const ContainerData& c) {
    std::vector<uint8_t> st(C_SIZE, 0);
    auto put = [&](size_t off, const void* v, size_t n) { memcpy(st.data() + off, v, n); };
    const u64 nvec = c.p.size() - 1;
    int64_t nonempty = 0;
    for (u64 k = 0; k < nvec; ++k) nonempty += c.p[k + 1] > c.p[k] ? 1 : 0;
    const int64_t unknown = -1;
    const int32_t format = c.hyper ? GXB_HYPERSPARSE : GXB_SPARSE, orient = GRB_ROWMAJOR;
    put(C_NROWS, &c.nrows, 8); put(C_NCOLS, &c.ncols, 8); put(C_NROWS_NONEMPTY, &nonempty, 8);
    put(C_NCOLS_NONEMPTY, &unknown, 8); put(C_NVALS, &c.nvals, 8); put(C_FORMAT, &format, 4); put(C_ORIENTATION, &orient, 4);
    st[C_ISO] = c.valued ? 0 : 1;
    st[C_JUMBLED] = 0;
    w.write_buffer(st.data(), st.size());
    // 32-bit integers whenever they fit, as GraphBLAS v10 prefers (the decoder accepts either width)
    const bool p32 = c.nvals < (1ull << 31), i32 = c.ncols <= (1ull << 31) && c.nrows <= (1ull << 31);
    auto pack = [](const std::vector<u64>& v, bool narrow) {
        std::vector<uint8_t> out(v.size() * (narrow ? 4 : 8));
        for (size_t k = 0; k < v.size(); ++k) {
            if (narrow) { uint32_t x = (uint32_t)v[k]; memcpy(out.data() + 4 * k, &x, 4); }
            else memcpy(out.data() + 8 * k, &v[k], 8);
        }
        return out;
    };
    // x, h, p, i, b — the order matrix.rs:520-525 writes them in
    if (c.valued) write_vector(w, c.x.data(), c.x.size() * 8, "GrB_UINT64", c.x.size());
    else { const uint8_t one = 1; write_vector(w, &one, c.nvals ? 1 : 0, "GrB_BOOL", c.nvals ? 1 : 0); }
    { auto hb = pack(c.h, i32); write_vector(w, hb.data(), hb.size(), i32 ? "GrB_UINT32" : "GrB_UINT64", c.h.size()); }
    { auto pb = pack(c.p, p32); write_vector(w, pb.data(), pb.size(), p32 ? "GrB_UINT32" : "GrB_UINT64", c.p.size()); }
    { auto ib = pack(c.i, i32); write_vector(w, ib.data(), ib.size(), i32 ? "GrB_UINT32" : "GrB_UINT64", c.i.size()); }
    write_vector(w, nullptr, 0, "GrB_INT8", 0);
}

// Decode<19> for Matrix<T> (matrix.rs:428-504): the arrays go to the device as they are
Matrix Matrix::decode(Context& ctx, ByteReader& r) {
    ContainerData c = parse_container(r);
    fgpu_mat* snap = nullptr;
    const u64 nvec = c.p.size() - 1;
    // fgpu_mat_from_csr reads "hypersparse" off a non-NULL hyper list: a hypersparse container without stored rows
    // (p = [0], h = [] — every clean dp / dm layer of a real graph) has an empty vector whose data() is NULL
    static const u64 no_rows = 0;
    const u64* hlist = c.hyper ? (c.h.empty() ? &no_rows : c.h.data()) : nullptr;
    check(fgpu_mat_from_csr(ctx.raw(), &snap, c.nrows, c.ncols, c.nvals, c.p.data(), 64, c.i.data(), 64,
                            c.valued ? c.x.data() : nullptr, hlist, c.hyper ? nvec : 0),
          "GxB_load_Matrix_from_Container");
    return Matrix::adopt(ctx, c.valued ? Type::UInt64 : Type::Bool, snap);
}

// Encode<19> for Matrix<T> (matrix.rs:506-546): the wait()ed state, sparse row-major (hypersparse when most rows are empty)
void Matrix::encode(ByteWriter& w) const {
    ContainerData c;
    c.nrows = nrows();
    c.ncols = ncols();
    c.valued = type() == Type::UInt64;
    u64 *rp = nullptr, *ci = nullptr, *vals = nullptr, nnz = 0;
    fgpu_ctx* raw = ctx().raw();
    check(fgpu_mat_export_csr(raw, snapshot(), &rp, &ci, c.valued ? &vals : nullptr, &nnz), "GxB_unload_Matrix_into_Container");
    c.nvals = nnz;
    c.i.assign(ci, ci + nnz);
    if (c.valued) c.x.assign(vals, vals + nnz);
    u64 stored = 0;
    for (u64 r = 0; r < c.nrows; ++r) stored += rp[r + 1] > rp[r] ? 1 : 0;
    c.hyper = c.nrows > 1024 && stored * 16 < c.nrows;     // what the deltas are pinned to (into_hyper, matrix.rs:558-575)
    if (c.hyper) {
        c.p.push_back(0);
        for (u64 r = 0; r < c.nrows; ++r)
            if (rp[r + 1] > rp[r]) { c.h.push_back(r); c.p.push_back(rp[r + 1]); }
    } else {
        c.p.assign(rp, rp + c.nrows + 1);
    }
    fgpu_free(raw, rp);
    fgpu_free(raw, ci);
    if (vals) fgpu_free(raw, vals);
    write_container(w, c);
}

// ---- Tensor (tensor.rs:1053-1209) ------------------------------------------------------------------------------
namespace {
constexpr u64 MSB_MASK = 1ull << 63;            // tensor.rs:1049-1051: multi-edge marker of the on-disk forward matrix
constexpr char PLAIN_MAGIC[8] = {'F', 'G', 'I', 'D', 'L', 'S', 'T', '1'};
}  // namespace

// The id list of a multi-edge pair.  C FalkorDB and the reference write it with GxB_Vector_serialize
// (vector.rs:150-174): GraphBLAS' internal, optionally ZSTD/LZ4-compressed blob of a GrB_BOOL vector of length
// GrB_INDEX_MAX with one `true` per edge id.  That format belongs to the un-vendored SuiteSparse:GraphBLAS v10.5.0 and
// no fixture of it exists in the reference tree, so it is NOT restated here: a process that embeds this layer next to
// GraphBLAS (the reference does) passes a codec built on GxB_Vector_serialize / _deserialize; the default codec
// writes a plain little-endian list (magic, count, ids) for this repository's own round trips and refuses anything
// else with an explicit error instead of guessing.  PARITY UNPINNED for the blob, pinned for the framing around it.
const Tensor::BlobCodec& Tensor::plain_blob_codec() {
    static const BlobCodec codec{
        [](const std::vector<u64>& ids) {
            std::vector<uint8_t> b(sizeof(PLAIN_MAGIC) + 8 + ids.size() * 8);
            memcpy(b.data(), PLAIN_MAGIC, sizeof(PLAIN_MAGIC));
            const u64 n = ids.size();
            memcpy(b.data() + 8, &n, 8);
            if (n) memcpy(b.data() + 16, ids.data(), n * 8);
            return b;
        },
        [](const std::vector<uint8_t>& b) {
            if (b.size() < 16 || memcmp(b.data(), PLAIN_MAGIC, sizeof(PLAIN_MAGIC)) != 0)
                throw GrbError(FGPU_INVALID,
                               "Tensor decode: the id-list blob is not in this library's plain form — a GxB_Vector_serialize blob "
                               "must be decoded by GraphBLAS (pass a BlobCodec built on GxB_Vector_deserialize)");
            u64 n = 0;
            memcpy(&n, b.data() + 8, 8);
            if ((b.size() - 16) / 8 < n) throw GrbError(FGPU_INVALID, "Tensor decode: truncated id list");
            std::vector<u64> ids(n);
            if (n) memcpy(ids.data(), b.data() + 16, n * 8);
            for (u64 k = 1; k < n; ++k)
                if (ids[k] <= ids[k - 1]) throw GrbError(FGPU_INVALID, "Tensor decode: id list not ascending");
            return ids;
        }};
    return codec;
}

void Tensor::encode(ByteWriter& w, const BlobCodec& codec) const {
    // the effective inline state, C-compatible (:1063-1081)
    std::vector<u64> rows, cols, vals;
    std::vector<std::pair<std::pair<u64, u64>, const std::vector<u64>*>> multi;
    wait_fwd();
    for (auto& e : merge_layers(m_.iter(0, ~0ull), dp_.layer().iter(0, ~0ull), dm_.layer().iter(0, ~0ull))) {
        rows.push_back(e.row);
        cols.push_back(e.col);
        if (e.val == MULTI_EDGE) {
            auto it = me_.find(compound_key(e.row, e.col));
            static const std::vector<u64> none;
            const std::vector<u64>& ids = it == me_.end() ? none : it->second;
            vals.push_back((u64)ids.size() | MSB_MASK);
            multi.push_back({{e.row, e.col}, &ids});
        } else {
            vals.push_back(e.val);
        }
    }
    Context& ctx = m_.ctx();
    Matrix empty(ctx, Type::UInt64, nrows(), ncols());
    if (rows.empty()) {
        empty.encode(w);
    } else {
        Matrix fm(ctx, Type::UInt64, nrows(), ncols());
        fm.build(rows, cols, &vals);
        fm.encode(w);
    }
    empty.encode(w);                                      // delta-plus (:1093)
    empty.encode(w);                                      // delta-minus: the same empty Matrix<u64> the reference writes (:1094)
    const u64 total = edge_count();
    w.write_unsigned(total);
    if (total == 0) return;
    w.write_unsigned(multi.size());                       // base group (:1106-1124)
    for (auto& mp : multi) {
        w.write_unsigned(mp.first.first);
        w.write_unsigned(mp.first.second);
        std::vector<uint8_t> blob = codec.encode(*mp.second);
        w.write_buffer(blob.data(), blob.size());
    }
    w.write_unsigned(0);                                  // empty delta-plus group (:1125)
}

Tensor Tensor::decode(Context& ctx, ByteReader& r, const BlobCodec& codec) {
    Matrix fwd_m = Matrix::decode(ctx, r);
    Matrix fwd_dp = Matrix::decode(ctx, r);
    Matrix fwd_dm = Matrix::decode(ctx, r);
    // an empty layer carries no values, so its element type is not recoverable from every encoder's payload: only a
    // layer that HAS entries must be UINT64 (base and delta-plus hold edge ids; delta-minus is read for its pattern only)
    if (fwd_m.nvals() && fwd_m.type() != Type::UInt64) throw GrbError(FGPU_INVALID, "Tensor decode: the forward matrix is not UINT64");
    if (fwd_dp.nvals() && fwd_dp.type() != Type::UInt64) throw GrbError(FGPU_INVALID, "Tensor decode: the forward delta-plus is not UINT64");
    if (fwd_dp.nrows() != fwd_m.nrows() || fwd_dp.ncols() != fwd_m.ncols() || fwd_dm.nrows() != fwd_m.nrows() ||
        fwd_dm.ncols() != fwd_m.ncols())
        throw GrbError(FGPU_DIM_MISMATCH, "Tensor decode: the forward layers differ in shape");
    const u64 nr = fwd_m.nrows(), nc = fwd_m.ncols();
    // (fwd_m \ fwd_dm) U fwd_dp, MSB-flagged values -> the MULTI_EDGE sentinel (:1141-1166)
    std::vector<u64> rows, cols, vals;
    const bool dm_empty = fwd_dm.nvals() == 0;
    std::map<std::pair<u64, u64>, u64> inl;
    auto take = [&](const Entry& e) {
        inl[{e.row, e.col}] = (e.val & MSB_MASK) ? MULTI_EDGE : e.val;
    };
    if (fwd_m.nvals())
        for (auto& e : fwd_m.iter(0, ~0ull)) {
            if (!dm_empty && fwd_dm.contains(e.row, e.col)) continue;
            take(e);
        }
    if (fwd_dp.nvals())
        for (auto& e : fwd_dp.iter(0, ~0ull)) take(e);
    for (auto& kv : inl) { rows.push_back(kv.first.first); cols.push_back(kv.first.second); vals.push_back(kv.second); }
    Tensor t(ctx, nr, nc);
    if (!rows.empty()) t.m_.build(rows, cols, &vals);
    const u64 total = r.read_unsigned();
    if (total > 0) {
        for (int group = 0; group < 2; ++group) {           // base, then delta-plus (:1170-1187)
            const u64 count = r.read_unsigned();
            for (u64 k = 0; k < count; ++k) {
                const u64 src = r.read_unsigned(), dst = r.read_unsigned();
                std::vector<uint8_t> blob = r.read_buffer();
                std::vector<u64> ids = codec.decode(blob);
                // the reference stores every id list it is handed under the pair's compound key, whatever the pair's inline
                // value says (tensor.rs:1175-1186): readers go through the inline value first, so a list for a single-edge
                // pair is simply never consulted
                auto& row = t.me_[compound_key(src, dst)];
                row.insert(row.end(), ids.begin(), ids.end());
                std::sort(row.begin(), row.end());
                row.erase(std::unique(row.begin(), row.end()), row.end());
            }
        }
    }
    t.m_.wait();                                             // the committed base is never pending (:1190-1192)
    // The u64 before the groups is `total_tensor_count` in the reference and only its being > 0 is tested
    // (tensor.rs:1169-1170): writers differ in what they count there (edges, tensors), so nothing is derived from its
    // value, nor from the count the MSB-flagged inline value carries.  A pair flagged multi-edge with NO id list anywhere is
    // accepted as the reference accepts it (tensor.rs:1169-1186 never cross-checks the two sections): Tensor::get finds no
    // row under its compound key and returns no ids, exactly what the reference's `me` row iterator yields.
    return t;
}

void Tensor::rebuild_backward() {
    std::vector<u64> r, c;
    for (auto& e : structural_iter(0, ~0ull)) { r.push_back(e.col); c.push_back(e.row); }
    mt_ = VersionedMatrix(m_.ctx(), ncols(), nrows());
    if (!r.empty()) {
        Matrix b(m_.ctx(), Type::Bool, ncols(), nrows());
        b.build(r, c);
        mt_ = VersionedMatrix::from_matrix(b);
    }
}

}  // namespace falkor
