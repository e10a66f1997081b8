// bitexpand.hpp — what the two files of the bit-parallel k-hop chain share (bitexpand.hip: states, hops, emission;
// bitpart.hip: the XCD-partitioned form of the dense counting hop).
#pragma once
#include "common.hpp"

namespace fgpu {

// Two 64-bit sums of a kernel (nnz + checksum, flops + rows) without same-address atomics: every wavefront adding into ONE
// pair of words costs ~5.6 ns per atomic at the memory side — 65 K wavefronts ending together made a ~100 us
// tail of the counting kernels.  A workgroup reduces its wavefronts through LDS and adds into one of BP_ACC_SLOTS pairs, a
// 128-byte line apart (atomics to different words of one line serialise too); the host sums the slots after ONE copy.
constexpr u32 BP_ACC_SLOTS = 256, BP_ACC_STRIDE = 16;   // stride in 64-bit words
constexpr size_t BP_ACC_WORDS = (size_t)BP_ACC_SLOTS * BP_ACC_STRIDE;
__device__ __forceinline__ void bp_block_add2(u64 a, u64 b, unsigned long long* __restrict__ acc) {
    __shared__ u64 s_acc_red[32];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
    }
    const u32 wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane_id() == 0) { s_acc_red[wv] = a; s_acc_red[16 + wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 sa = 0, sb = 0;
        for (u32 i = 0; i < nw; ++i) { sa += s_acc_red[i]; sb += s_acc_red[16 + i]; }
        unsigned long long* slot = acc + (size_t)(blockIdx.x % BP_ACC_SLOTS) * BP_ACC_STRIDE;
        if (sa) atomicAdd(slot, (unsigned long long)sa);
        if (sb) atomicAdd(slot + 1, (unsigned long long)sb);
    }
}

// What happens to a finished row of a COUNTING hop (the last hop of a count-only chain): it is counted where it is
// produced (MODE 2: and its checksum terms summed through the LDS nibble tables) and never written, except for "touched"
// rows — rows named by a delta layer (plain pull: also rows cut into several items) — which go to their slot of the side
// buffer (slot = rank of v in the touched bitmap).
struct BpFinal {
    const u64* tbits;        // touched bitmap (n bits)
    const u32* tpref;        // exclusive prefix of its word popcounts
    const u64* label;        // destination-label bitmap (nullable)
    const u64* tab;          // checksum tables, w x 256 (MODE 2)
    unsigned long long* acc; // BP_ACC_WORDS slots: [0] nnz, [1] checksum
    u32 w;
};

// ---- XCD-partitioned dense counting hop (bitpart.hip) ---------------------------------------------------------------------
struct BpXPlan;
// the plan of `t` (= the cached transpose of m), built under m's index mutex on the first dense counting hop over it;
// *out = nullptr when the partitioned form does not apply (option off, matrix too small / too wide, ids beyond 2^26)
fgpu_info bp_xplan(fgpu_ctx* ctx, const fgpu_mat* m, const fgpu_mat* t, const BpXPlan** out);
void bp_xplan_release(fgpu_ctx* ctx, BpXPlan* p);
const u32* bp_xplan_perm(const BpXPlan* p);   // slot of row u of X in the state the plan gathers from (nullptr: slot = u)
// rows of Y = OR of the gathered rows of X, per vertex, counted / check-summed (mode 1 / 2) or — touched rows — stored into
// their side-buffer slot; `side` is zeroed by the caller, the delta fix-ups and the side-row count follow in bp_hop_impl
fgpu_info bp_xpull_count(fgpu_ctx* ctx, const BpXPlan* xp, const fgpu_mat* t, const u64* x, u32 ws, int mode, const BpFinal& fin,
                         u64* side, size_t lds_tables, u64 xrows);

}  // namespace fgpu
