/*
 * oracle.c — CPU restatement of the reference's traversal algebra.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (falkordb_amd/, libfgpu.so) never links or calls it.
 *
 * The reference (FalkorDB Rust engine, /root/reference) delegates the arithmetic of this path
 * to SuiteSparse:GraphBLAS v10.5.0 and LAGraph v1.3.x (graphblas.sh:71-72), neither of which is
 * vendored, and cannot be built here (no rustc/cargo, no libgraphblas).  This file therefore
 * restates the *observable semantics* the reference relies on, function by function, citing
 * the reference call site each one follows.  PARITY PINNING: the restatement is checked
 * against every golden vector the reference's own tests hold for this path (tests/golden/,
 * tests/test_oracle_golden.py: matrix.rs:1686-1695 dup-collapse, versioned_matrix.rs:1278-1330
 * fold thresholds, versioned_matrix.rs:1399-1523 delta model tests, tests/flow/test_bfs.py,
 * tests/flow/social/ fixtures, tests/flow/test_expand_into.py) and cross-checked against scipy.sparse.
 * For large-graph results of GrB_mxm with complemented masks and LAGraph's parent choice the
 * reference holds no fixtures: there parity is "unpinned" and defined structurally.
 *
 * All products are over the ANY_PAIR boolean semiring: structure only, values never read.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef int32_t i32;

static int cmp_u64(const void* a, const void* b) {
    u64 x = *(const u64*)a, y = *(const u64*)b;
    return (x > y) - (x < y);
}

/* Matrix::<bool>::build -> GxB_Matrix_build_Scalar (matrix.rs:1281-1303): COO -> CSR, duplicate
 * coordinates collapse to one entry (pinned by matrix.rs:1686-1695), rows sorted ascending
 * (the state Matrix::wait + row iterator expose, matrix.rs:781-796, 1471-1605).
 * rowptr[nrows+1], colidx[>= n].  Returns nnz. */
u64 orc_build_csr(u64 nrows, const u64* rows, const u64* cols, u64 n, u64* rowptr, u64* colidx) {
    u64* key = (u64*)malloc((n ? n : 1) * sizeof(u64));
    for (u64 i = 0; i < n; ++i) key[i] = (rows[i] << 32) | cols[i];
    qsort(key, n, sizeof(u64), cmp_u64);
    memset(rowptr, 0, (nrows + 1) * sizeof(u64));
    u64 nnz = 0;
    for (u64 i = 0; i < n; ++i) {
        if (i && key[i] == key[i - 1]) continue;
        colidx[nnz++] = key[i] & 0xFFFFFFFFull;
        rowptr[(key[i] >> 32) + 1]++;
    }
    for (u64 r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
    free(key);
    return nnz;
}

/* GrB_transpose (Matrix::transpose, matrix.rs:633-662).  trp[ncols+1], tci[nnz]. */
void orc_transpose(u64 nrows, u64 ncols, const u64* rp, const u64* ci, u64* trp, u64* tci) {
    memset(trp, 0, (ncols + 1) * sizeof(u64));
    for (u64 k = 0; k < rp[nrows]; ++k) trp[ci[k] + 1]++;
    for (u64 c = 0; c < ncols; ++c) trp[c + 1] += trp[c];
    u64* cur = (u64*)malloc((ncols + 1) * sizeof(u64));
    memcpy(cur, trp, (ncols + 1) * sizeof(u64));
    for (u64 r = 0; r < nrows; ++r)
        for (u64 k = rp[r]; k < rp[r + 1]; ++k) tci[cur[ci[k]]++] = r; /* ascending r per column */
    free(cur);
}

/* C = F x B over ANY_PAIR, no mask: Matrix::lmxm -> GrB_mxm(GxB_ANY_PAIR_BOOL)
 * (matrix.rs:930-947).  Gustavson row gather with a marker array, then per-row sort — the same
 * algorithm family as GraphBLAS saxpy3.  Pass crp[k+1]; cci may be NULL for a counting pass.
 * Returns nnz(C).  *flops (nullable) receives sum_{(i,s) in F} deg_B(s). */
u64 orc_mxm(u64 k, const u64* frp, const u64* fci, const u64* brp, const u64* bci, u64 ncols_b, u64* crp,
            u64* cci, u64* flops) {
    u64* mark = (u64*)malloc((ncols_b ? ncols_b : 1) * sizeof(u64));
    for (u64 j = 0; j < ncols_b; ++j) mark[j] = ~0ull;
    u64 nnz = 0, fl = 0;
    crp[0] = 0;
    for (u64 i = 0; i < k; ++i) {
        u64 row_start = nnz;
        for (u64 p = frp[i]; p < frp[i + 1]; ++p) {
            u64 s = fci[p];
            fl += brp[s + 1] - brp[s];
            for (u64 q = brp[s]; q < brp[s + 1]; ++q) {
                u64 j = bci[q];
                if (mark[j] != i) {
                    mark[j] = i;
                    if (cci) cci[nnz] = j;
                    nnz++;
                }
            }
        }
        if (cci && nnz - row_start > 1) qsort(cci + row_start, nnz - row_start, sizeof(u64), cmp_u64);
        crp[i + 1] = nnz;
    }
    free(mark);
    if (flops) *flops = fl;
    return nnz;
}

/* Row-wise (a \ mask) U add on sorted-unique CSR rows: the GrB_DESC_RSC masked assignment +
 * eWiseAdd(ANY) that close Matrix::delta_lmxm (matrix.rs:1382-1400), and the pattern algebra
 * of VersionedMatrix::extract (versioned_matrix.rs:609-620).  mask / add may be NULL.
 * mask_covers_add != 0 reproduces flush's eWiseAdd<!dm>(m, dp) arm (versioned_matrix.rs:905-911). */
u64 orc_merge(u64 nrows, const u64* arp, const u64* aci, const u64* prp, const u64* pci, const u64* mrp,
              const u64* mci, int mask_covers_add, u64* orp, u64* oci) {
    u64 nnz = 0;
    orp[0] = 0;
    for (u64 r = 0; r < nrows; ++r) {
        u64 a = arp[r], ae = arp[r + 1];
        u64 p = prp ? prp[r] : 0, pe = prp ? prp[r + 1] : 0;
        u64 m = mrp ? mrp[r] : 0, me = mrp ? mrp[r + 1] : 0;
        while (a < ae || p < pe) {
            u64 x;
            int from_a = 0, from_p = 0;
            if (a < ae && (p >= pe || aci[a] <= pci[p])) { x = aci[a]; from_a = 1; }
            else { x = pci[p]; }
            if (p < pe && pci[p] == x) from_p = 1;
            if (from_a) a++;
            if (from_p) p++;
            while (m < me && mci[m] < x) m++;
            int masked = (m < me && mci[m] == x);
            int keep = from_p ? (mask_covers_add ? !masked : 1) : !masked;
            if (keep) { if (oci) oci[nnz] = x; nnz++; }
        }
        orp[r + 1] = nnz;
    }
    return nnz;
}

/* Level-synchronous BFS: the observable contract of LAGr_BreadthFirstSearch_Extended(level,
 * parent, G, src, max_level, -1, false) as algo.BFS consumes it (algo_procedures.rs:1079-1160):
 * level[src] = 0, level[v] = hop distance, unreached = -1; max_level < 0 unlimited, otherwise
 * vertices deeper than max_level stay unreached (test_bfs.py: depth 1 from a => {b}).
 * parent[v] = the lowest-id vertex of the previous level with an edge to v (LAGraph's ANY
 * monoid permits any valid parent; tests compare parents by validity).
 * Returns the number of out-edges of reached vertices (TEPS numerator). */
u64 orc_bfs(u64 n, const u64* rp, const u64* ci, u64 src, i64 max_level, i32* level, i64* parent) {
    for (u64 v = 0; v < n; ++v) { level[v] = -1; if (parent) parent[v] = -1; }
    u64* q = (u64*)malloc((n ? n : 1) * sizeof(u64));
    u64 head = 0, tail = 0, edges = 0;
    level[src] = 0;
    if (parent) parent[src] = (i64)src;
    q[tail++] = src;
    while (head < tail) {
        u64 v = q[head++];
        edges += rp[v + 1] - rp[v];
        if (max_level >= 0 && level[v] >= max_level) continue;
        for (u64 k = rp[v]; k < rp[v + 1]; ++k) {
            u64 u = ci[k];
            if (level[u] < 0) {
                level[u] = level[v] + 1;
                if (parent) parent[u] = (i64)v;
                q[tail++] = u;
            }
        }
    }
    free(q);
    return edges;
}

/* w<!mask, replace> = f x A over the boolean semiring, bitmap vectors (GrB_vxm as LAGraph's BFS
 * issues it; graphblas/mod.rs:11173).  mask nullable. */
void orc_vxm(u64 n, const u64* rp, const u64* ci, const u64* f, const u64* mask, u64* w) {
    u64 nw = (n + 63) / 64;
    memset(w, 0, nw * sizeof(u64));
    for (u64 v = 0; v < n; ++v) {
        if (!((f[v >> 6] >> (v & 63)) & 1ull)) continue;
        for (u64 k = rp[v]; k < rp[v + 1]; ++k) {
            u64 u = ci[k];
            if (mask && ((mask[u >> 6] >> (u & 63)) & 1ull)) continue;
            w[u >> 6] |= 1ull << (u & 63);
        }
    }
}

/* sum over entries of mix64((row << 32) | col): the order-independent checksum fgpu_expand_count
 * reports (full-size parity property). */
static u64 mix64(u64 z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
/* order-independent checksum of a (row, dest) result: sum of row_hash(row) * dest_hash(dest) mod 2^64 with
 * row_hash = mix64, dest_hash = mix64(dest ^ golden) | 1 — the definition include/fgpu.h gives for fgpu_expand_count */
u64 orc_checksum(u64 nrows, const u64* rp, const u64* ci) {
    u64 s = 0;
    for (u64 r = 0; r < nrows; ++r) {
        const u64 hr = mix64(r);
        for (u64 k = rp[r]; k < rp[r + 1]; ++k) s += hr * (mix64(ci[k] ^ 0x9e3779b97f4a7c15ull) | 1ull);
    }
    return s;
}
