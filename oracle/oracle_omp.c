/* oracle_omp.c — multi-threaded CPU baseline for the BFS bench (TEST / BENCH INFRASTRUCTURE ONLY).
 *
 * The reference runs algo.BFS through LAGr_BreadthFirstSearch (LAGraph v1.3.x over SuiteSparse:GraphBLAS
 * v10.5.0, OpenMP inside every GrB_vxm / GrB_mxv; algo_procedures.rs:1079-1088).  Neither library is
 * vendored or installable here, so bench.py's `cpu_baseline` times this stand-in on the GPU box's host
 * cores instead: the same algorithm family LAGraph's BFS uses — level-synchronous, push (vxm over the
 * frontier) or pull (mxv over the unvisited rows of A') chosen per level — written directly with OpenMP.
 * It is checked against the serial oracle (oracle.c orc_bfs) by tests/test_oracle_golden.py; it is never
 * linked into the product.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef int64_t i64;
typedef int32_t i32;

int orc_omp_threads(void) { return omp_get_max_threads(); }

/* rp/ci: CSR of A (out-edges); trp/tci: CSR of A' (in-edges), nullable => push only.
 * Returns the traversed-edge count (sum of out-degrees of reached vertices), fills level[n] (-1 unreached). */
u64 orc_bfs_omp(u64 n, const u64* rp, const u64* ci, const u64* trp, const u64* tci, u64 src, i64 max_level,
                i32* level, int threads, double alpha) {
    if (threads > 0) omp_set_num_threads(threads);
    const int T = omp_get_max_threads();
#pragma omp parallel for schedule(static)
    for (u64 v = 0; v < n; ++v) level[v] = -1;
    u64* cur = (u64*)malloc((n ? n : 1) * sizeof(u64));
    u64* nxt = (u64*)malloc((n ? n : 1) * sizeof(u64));
    u64** loc = (u64**)malloc((size_t)T * sizeof(u64*));
    u64* loc_n = (u64*)calloc((size_t)T, sizeof(u64));
    u64* loc_cap = (u64*)calloc((size_t)T, sizeof(u64));
    for (int t = 0; t < T; ++t) { loc_cap[t] = 1024; loc[t] = (u64*)malloc(1024 * sizeof(u64)); }
    u64 ncur = 1, edges = 0, reached = 1;
    const u64 nnz_t = trp ? trp[n] : 0;
    level[src] = 0;
    cur[0] = src;
    i32 L = 0;
    while (ncur) {
        u64 mf = 0;
#pragma omp parallel for reduction(+ : mf) schedule(static)
        for (u64 i = 0; i < ncur; ++i) mf += rp[cur[i] + 1] - rp[cur[i]];
        edges += mf;
        if (max_level >= 0 && L >= max_level) break;
        const double unvisited_edges = trp ? (double)nnz_t * (double)(n - reached) / (double)(n ? n : 1) : 0.0;
        const int pull = trp && (double)mf * alpha > unvisited_edges;
        for (int t = 0; t < T; ++t) loc_n[t] = 0;
        if (!pull) {
#pragma omp parallel
            {
                const int t = omp_get_thread_num();
#pragma omp for schedule(dynamic, 64)
                for (u64 i = 0; i < ncur; ++i) {
                    const u64 v = cur[i];
                    for (u64 k = rp[v]; k < rp[v + 1]; ++k) {
                        const u64 u = ci[k];
                        if (level[u] < 0 && __sync_bool_compare_and_swap(&level[u], -1, L + 1)) {
                            if (loc_n[t] == loc_cap[t]) {
                                loc_cap[t] *= 2;
                                loc[t] = (u64*)realloc(loc[t], loc_cap[t] * sizeof(u64));
                            }
                            loc[t][loc_n[t]++] = u;
                        }
                    }
                }
            }
        } else {
#pragma omp parallel
            {
                const int t = omp_get_thread_num();
#pragma omp for schedule(dynamic, 1024)
                for (u64 u = 0; u < n; ++u) {
                    if (level[u] >= 0) continue;
                    for (u64 k = trp[u]; k < trp[u + 1]; ++k) {
                        if (level[tci[k]] == L) {   /* first in-neighbour in the frontier: early exit */
                            level[u] = L + 1;       /* only this thread writes level[u] in a pull level */
                            if (loc_n[t] == loc_cap[t]) {
                                loc_cap[t] *= 2;
                                loc[t] = (u64*)realloc(loc[t], loc_cap[t] * sizeof(u64));
                            }
                            loc[t][loc_n[t]++] = u;
                            break;
                        }
                    }
                }
            }
        }
        u64 nn = 0;
        for (int t = 0; t < T; ++t) {
            memcpy(nxt + nn, loc[t], loc_n[t] * sizeof(u64));
            nn += loc_n[t];
        }
        u64* tmp = cur; cur = nxt; nxt = tmp;
        ncur = nn;
        reached += nn;
        ++L;
    }
    for (int t = 0; t < T; ++t) free(loc[t]);
    free(loc); free(loc_n); free(loc_cap); free(cur); free(nxt);
    return edges;
}

/* ------------------------------------------------------------------------------------------------------
 * C = F x B over ANY_PAIR, no mask — Matrix::lmxm -> GrB_mxm(GxB_ANY_PAIR_BOOL) (matrix.rs:930-947) — row-parallel.
 * The same Gustavson gather as oracle.c's orc_mxm (which it is checked against, tests/test_oracle_golden.py), one
 * row of F per task, a per-thread sparse accumulator: a column bitmap when the row gathers more than ncols/16
 * entries (emission = an ascending scan of the bitmap, no sort), a marker bitmap + sorted list otherwise.
 * Used (a) by the -m gpu parity tests at BASELINE sizes, where the serial oracle would take minutes, and (b) as
 * bench.py's k-hop `cpu_baseline` — the CPU stand-in for SuiteSparse's saxpy3 (same algorithm family).
 * crp[k+1] is filled; *cci_out is malloc'ed (release with orc_free).  Returns nnz(C).
 * ------------------------------------------------------------------------------------------------------ */
static int cmp_u64_omp(const void* a, const void* b) {
    u64 x = *(const u64*)a, y = *(const u64*)b;
    return (x > y) - (x < y);
}

void orc_free(void* p) { free(p); }

u64 orc_mxm_omp(u64 k, const u64* frp, const u64* fci, const u64* brp, const u64* bci, u64 ncols_b, u64* crp,
                u64** cci_out, u64* flops, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    const u64 nw = (ncols_b + 63) / 64;
    u64** rows = (u64**)calloc(k ? k : 1, sizeof(u64*));
    u64* cnt = (u64*)calloc(k + 1, sizeof(u64));
    u64 fl_total = 0;
#pragma omp parallel reduction(+ : fl_total)
    {
        u64* bm = (u64*)calloc(nw ? nw : 1, sizeof(u64));
        u64 cap = 1024, *list = (u64*)malloc(cap * sizeof(u64));
#pragma omp for schedule(dynamic, 1)
        for (u64 i = 0; i < k; ++i) {
            u64 fl = 0;
            for (u64 p = frp[i]; p < frp[i + 1]; ++p) fl += brp[fci[p] + 1] - brp[fci[p]];
            fl_total += fl;
            if (fl == 0) continue;
            if (fl * 16 >= ncols_b) { /* dense row: bitmap accumulator, ascending scan */
                for (u64 p = frp[i]; p < frp[i + 1]; ++p) {
                    const u64 s = fci[p];
                    for (u64 q = brp[s]; q < brp[s + 1]; ++q) bm[bci[q] >> 6] |= 1ull << (bci[q] & 63);
                }
                u64 c = 0;
                for (u64 w = 0; w < nw; ++w) c += (u64)__builtin_popcountll(bm[w]);
                u64* out = (u64*)malloc(c * sizeof(u64));
                u64 o = 0;
                for (u64 w = 0; w < nw; ++w) {
                    u64 bits = bm[w];
                    bm[w] = 0;
                    while (bits) {
                        out[o++] = (w << 6) | (u64)__builtin_ctzll(bits);
                        bits &= bits - 1;
                    }
                }
                rows[i] = out;
                cnt[i] = c;
            } else { /* sparse row: marker bits + list, sorted afterwards */
                u64 c = 0;
                if (fl > cap) { cap = fl; list = (u64*)realloc(list, cap * sizeof(u64)); }
                for (u64 p = frp[i]; p < frp[i + 1]; ++p) {
                    const u64 s = fci[p];
                    for (u64 q = brp[s]; q < brp[s + 1]; ++q) {
                        const u64 j = bci[q];
                        if (!((bm[j >> 6] >> (j & 63)) & 1ull)) {
                            bm[j >> 6] |= 1ull << (j & 63);
                            list[c++] = j;
                        }
                    }
                }
                for (u64 t = 0; t < c; ++t) bm[list[t] >> 6] = 0;
                if (c > 1) qsort(list, c, sizeof(u64), cmp_u64_omp);
                u64* out = (u64*)malloc((c ? c : 1) * sizeof(u64));
                memcpy(out, list, c * sizeof(u64));
                rows[i] = out;
                cnt[i] = c;
            }
        }
        free(bm);
        free(list);
    }
    crp[0] = 0;
    for (u64 i = 0; i < k; ++i) crp[i + 1] = crp[i] + cnt[i];
    const u64 nnz = crp[k];
    u64* cci = (u64*)malloc((nnz ? nnz : 1) * sizeof(u64));
#pragma omp parallel for schedule(dynamic, 1)
    for (u64 i = 0; i < k; ++i) {
        if (rows[i]) {
            memcpy(cci + crp[i], rows[i], cnt[i] * sizeof(u64));
            free(rows[i]);
        }
    }
    free(rows);
    free(cnt);
    *cci_out = cci;
    if (flops) *flops = fl_total;
    return nnz;
}

/* Row-parallel (a \ mask) U add on sorted-unique rows — the closing step of Matrix::delta_lmxm (matrix.rs:1382-1400);
 * the same semantics as oracle.c's orc_merge with mask_covers_add = 0 (the accumulated dp product is not masked).
 * orp[nrows+1]; *oci_out malloc'ed.  Returns nnz. */
static u64 merge_row(const u64* a, u64 na, const u64* p, u64 np_, const u64* m, u64 nm, u64* out) {
    u64 ia = 0, ip = 0, im = 0, o = 0;
    while (ia < na || ip < np_) {
        u64 x;
        int from_a = 0, from_p = 0;
        if (ia < na && (ip >= np_ || a[ia] <= p[ip])) { x = a[ia]; from_a = 1; }
        else x = p[ip];
        if (ip < np_ && p[ip] == x) from_p = 1;
        if (from_a) ia++;
        if (from_p) ip++;
        while (im < nm && m[im] < x) im++;
        const int masked = im < nm && m[im] == x;
        if (from_p || !masked) { if (out) out[o] = x; o++; }   /* add entries survive the mask */
    }
    return o;
}

u64 orc_merge_omp(u64 nrows, const u64* arp, const u64* aci, const u64* prp, const u64* pci, const u64* mrp,
                  const u64* mci, u64* orp, u64** oci_out, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    u64* cnt = (u64*)calloc(nrows + 1, sizeof(u64));
#pragma omp parallel for schedule(dynamic, 1)
    for (u64 r = 0; r < nrows; ++r)
        cnt[r] = merge_row(aci + arp[r], arp[r + 1] - arp[r], prp ? pci + prp[r] : 0, prp ? prp[r + 1] - prp[r] : 0,
                           mrp ? mci + mrp[r] : 0, mrp ? mrp[r + 1] - mrp[r] : 0, 0);
    orp[0] = 0;
    for (u64 r = 0; r < nrows; ++r) orp[r + 1] = orp[r] + cnt[r];
    u64* oci = (u64*)malloc((orp[nrows] ? orp[nrows] : 1) * sizeof(u64));
#pragma omp parallel for schedule(dynamic, 1)
    for (u64 r = 0; r < nrows; ++r)
        merge_row(aci + arp[r], arp[r + 1] - arp[r], prp ? pci + prp[r] : 0, prp ? prp[r + 1] - prp[r] : 0,
                  mrp ? mci + mrp[r] : 0, mrp ? mrp[r + 1] - mrp[r] : 0, oci + orp[r]);
    free(cnt);
    *oci_out = oci;
    return orp[nrows];
}

/* order-independent checksum of a CSR result: sum of mix64(row) * (mix64(col ^ golden) | 1) mod 2^64 (fgpu_expand_count's) */
static inline u64 mix64_omp(u64 z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
u64 orc_checksum_omp(u64 nrows, const u64* rp, const u64* ci, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    u64 sum = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : sum)
    for (u64 r = 0; r < nrows; ++r) {
        const u64 hr = mix64_omp(r);
        for (u64 k = rp[r]; k < rp[r + 1]; ++k) sum += hr * (mix64_omp(ci[k] ^ 0x9e3779b97f4a7c15ull) | 1ull);
    }
    return sum;
}
