/* CPU count behind DESIGN.md §4.3 (round 4, VERDICT r03 #4a): how much of hop 3's gather work a "saturation exit" could
 * skip.  Input (binary dumps written by khop_saturation.py): A' as CSR (rowptr u64[n+1], colidx u64[nnz]) and the bit state X
 * after hop 2 (n rows x W u64 words, bit s of row u = source s reaches u in two hops).  For every row v of A' the entries
 * are walked in storage order OR-ing X[u]; reported: rows / entries whose final Y[v] is all-ones over the batch's `bits`
 * source bits, and how many of their gathers come AFTER the accumulator became full (what an exit test after every gather,
 * after every 64 gathers — one round of the pull's wavefront — and per 256-entry item could skip).
 * build: gcc -O3 -fopenmp -o /tmp/w/khop_saturation tools/experiments/khop_saturation.c */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
static void* slurp(const char* p, size_t* n) {
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(1); }
    fseek(f, 0, SEEK_END); *n = ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*n); if (fread(b, 1, *n, f) != *n) { perror("read"); exit(1); } fclose(f); return b;
}
int main(int argc, char** argv) {
    size_t nb;
    u64* rp = slurp(argv[1], &nb); const u64 n = nb / 8 - 1;
    u64* ci = slurp(argv[2], &nb); const u64 nnz = nb / 8;
    u64* x = slurp(argv[3], &nb); const int W = (int)(nb / 8 / n);
    int bits = atoi(argv[4]);
    if (bits <= 0) {   /* saturation against what CAN be reached: U = OR of every row of X (sources with an empty 2-hop set never set a bit) */
        u64 U[64]; memset(U, 0, sizeof(U));
        for (u64 v = 0; v < n; ++v) for (int w = 0; w < W; ++w) U[w] |= x[v * W + w];
        bits = 0;
        for (int w = 0; w < W; ++w) bits += __builtin_popcountll(U[w]);
        printf("U = OR of all rows of X: %d of %d source bits\n", bits, W * 64);
    }
    u64 full_rows = 0, full_entries = 0, skip1 = 0, skip64 = 0, skip_item = 0, nz_rows = 0, popc = 0;
    u64 hist[6] = {0};   /* entries by final popcount class: 0, <64, <256, <512, <bits, bits */
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : full_rows, full_entries, skip1, skip64, skip_item, nz_rows, popc, hist[:6])
    for (u64 v = 0; v < n; ++v) {
        u64 acc[64]; memset(acc, 0, sizeof(acc));
        const u64 b = rp[v], e = rp[v + 1];
        u64 first_full = e;
        for (u64 k = b; k < e; ++k) {
            const u64* r = x + ci[k] * W;
            int pc = 0;
            for (int w = 0; w < W; ++w) { acc[w] |= r[w]; pc += __builtin_popcountll(acc[w]); }
            if (pc == bits && first_full == e) first_full = k;   /* full after gather k (inclusive) */
        }
        int pc = 0;
        for (int w = 0; w < W; ++w) pc += __builtin_popcountll(acc[w]);
        popc += pc; nz_rows += pc != 0;
        const u64 d = e - b;
        hist[pc == 0 ? 0 : pc < 64 ? 1 : pc < 256 ? 2 : pc < 512 ? 3 : pc < bits ? 4 : 5] += d;
        if (pc == bits) {
            ++full_rows; full_entries += d;
            const u64 done = first_full - b + 1;                 /* gathers needed */
            skip1 += d - done;
            const u64 r64 = (done + 63) / 64 * 64;  skip64 += d > r64 ? d - r64 : 0;
            /* per 256-entry item, items of a row processed independently: only the gathers of THIS item's later rounds */
            for (u64 i0 = 0; i0 < d; i0 += 256) {
                u64 a2[64]; memset(a2, 0, sizeof(a2));
                const u64 i1 = i0 + 256 < d ? i0 + 256 : d;
                u64 ff = i1;
                for (u64 k = i0; k < i1; ++k) {
                    const u64* r = x + ci[b + k] * W; int p2 = 0;
                    for (int w = 0; w < W; ++w) { a2[w] |= r[w]; p2 += __builtin_popcountll(a2[w]); }
                    if (p2 == bits) { ff = k; break; }
                }
                if (ff < i1) { const u64 dn = (ff - i0 + 1 + 63) / 64 * 64; skip_item += (i1 - i0) > dn ? (i1 - i0) - dn : 0; }
            }
        }
    }
    printf("n %llu nnz %llu W %d bits %d: result nnz %llu (density %.4f), non-zero rows %llu\n", (unsigned long long)n,
           (unsigned long long)nnz, W, bits, (unsigned long long)popc, (double)popc / ((double)n * bits), (unsigned long long)nz_rows);
    const char* cls[6] = {"0", "1-63", "64-255", "256-511", "512-(bits-1)", "all bits"};
    for (int i = 0; i < 6; ++i) printf("  entries in rows with final popcount %-13s %12llu  %.3f\n", cls[i], (unsigned long long)hist[i], (double)hist[i] / nnz);
    printf("rows with Y[v] all-ones: %llu, their entries %llu (%.3f of nnz)\n", (unsigned long long)full_rows, (unsigned long long)full_entries, (double)full_entries / nnz);
    printf("gathers after saturation, exit test after every gather (whole row, storage order): %llu (%.3f of nnz)\n", (unsigned long long)skip1, (double)skip1 / nnz);
    printf("   ... test after every 64 gathers (whole row as one sequence):                    %llu (%.3f of nnz)\n", (unsigned long long)skip64, (double)skip64 / nnz);
    printf("   ... per 256-entry item on its own, test after every 64 gathers (the kernel's shape): %llu (%.3f of nnz)\n", (unsigned long long)skip_item, (double)skip_item / nnz);
    return 0;
}
