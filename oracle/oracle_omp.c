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
