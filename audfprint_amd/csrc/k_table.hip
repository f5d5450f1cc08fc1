// k_table.hip -- SURVEY.md §8(f) row f1: batch build of the reference hash table on the GPU.
//
// Replaces the per-hash Python loop of HashTable.store (hash_table.py:91-138) for a whole batch of
// clips: entry = ((id+1) << maxtimebits) + (time & timemask) appended to bucket (hash & hashmask)
// at slot counts[bucket] while the bucket has room (count < depth).  Slot order inside a bucket
// is the reference's insertion order (clip order, then row order), which is the order of the
// global row index -- so the result is bit-identical to calling store() clip by clip.  Insertions
// into a FULL bucket draw from Python's `random` in the reference (:128-132); those are not
// performed here but reported as ordered events (row, bucket, value, count) that the host replays
// with the very same RNG calls (audfprint_amd/table.py).
//
// Integer scatter work, HBM/atomic bound: count per bucket (atomics, order-free) -> exclusive scan
// -> scatter (row, value) pairs into per-bucket segments -> sort each small segment by row index
// (restores insertion order deterministically) -> write table slots / overflow events.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "afp_common.h"

#define TB_SMALL 32

__global__ __launch_bounds__(256)
void k_tb_count(TableArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.nrows) return;
    const int k = A.rows[2 * i + 1] & ((1 << A.hashbits) - 1);
    atomicAdd(reinterpret_cast<unsigned long long*>(A.newcnt + k), 1ull);
}

__global__ __launch_bounds__(256)
void k_tb_scatter(TableArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.nrows) return;
    int lo = 0, hi = A.nclips;                       // clip c with clip_off[c] <= i < clip_off[c+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.clip_off[mid] <= i) lo = mid; else hi = mid; }
    const uint32_t idval = (uint32_t)(A.clip_ids[lo] + 1) << A.maxtimebits;                // :110
    const uint32_t val = idval + ((uint32_t)A.rows[2 * i] & ((1u << A.maxtimebits) - 1u));  // :118-120
    const int k = A.rows[2 * i + 1] & ((1 << A.hashbits) - 1);                             // :113
    const int64_t pos = A.first[k] + atomicAdd(A.fill + k, 1);
    A.seg[pos] = ((unsigned long long)(uint32_t)i << 32) | val;
}

__device__ __forceinline__ void tb_place(const TableArgs& A, int k, int r, unsigned long long e, int base)
{
    const int c = base + r;                                   // count at the time of this insertion (:115)
    const uint32_t val = (uint32_t)(e & 0xffffffffull);
    if (c < A.depth) {
        A.table[(int64_t)k * A.depth + c] = val;              // :121-124
    } else {
        const int o = atomicAdd(A.ovcnt, 1);                  // full bucket: the host replays random.randint (:125-131)
        A.overflow[4 * (int64_t)o] = (int32_t)(e >> 32);
        A.overflow[4 * (int64_t)o + 1] = k;
        A.overflow[4 * (int64_t)o + 2] = (int32_t)val;
        A.overflow[4 * (int64_t)o + 3] = c;
    }
}

// one thread per bucket: sort the (short) segment by row index, then place
__global__ __launch_bounds__(256)
void k_tb_fill(TableArgs A)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= (1 << A.hashbits)) return;
    const int64_t f = A.first[k];
    const int n = (int)(A.first[k + 1] - f);
    if (n == 0) return;
    if (n > TB_SMALL) { A.biglist[atomicAdd(A.bigcnt, 1)] = k; return; }
    unsigned long long e[TB_SMALL];
    for (int j = 0; j < n; j++) {                              // insertion sort (row index = high half)
        unsigned long long v = A.seg[f + j];
        int m = j;
        while (m > 0 && e[m - 1] > v) { e[m] = e[m - 1]; m--; }
        e[m] = v;
    }
    const int base = A.counts[k];
    for (int j = 0; j < n; j++) tb_place(A, k, j, e[j], base);
    A.counts[k] = base + n;                                    // :134
}

// long segments (a very popular hash): one workgroup per bucket, rank by counting
__global__ __launch_bounds__(256)
void k_tb_fill_big(TableArgs A)
{
    const int nbig = *A.bigcnt;
    for (int b = blockIdx.x; b < nbig; b += gridDim.x) {
        const int k = A.biglist[b];
        const int64_t f = A.first[k];
        const int n = (int)(A.first[k + 1] - f);
        const int base = A.counts[k];
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            const unsigned long long v = A.seg[f + j];
            int r = 0;
            for (int m = 0; m < n; m++) r += (A.seg[f + m] < v) ? 1 : 0;  // rows are distinct: a strict rank
            tb_place(A, k, r, v, base);
        }
        __syncthreads();
        if (threadIdx.x == 0) A.counts[k] = base + n;
    }
}

// ---- row f4 (first half): HashTable.get_hits (hash_table.py:150-176) -----------------------------
// rows [N][2] (time, hash) -> for every row the first min(depth, counts) entries of its bucket, as
// [id, stored_time - time, hash, time] int32 rows in the reference's order (row order, slot order).
__global__ __launch_bounds__(256)
void k_gh_count(const int32_t* __restrict__ rows, int64_t n, int hashbits, int depth, const int32_t* __restrict__ counts,
                int64_t* __restrict__ nids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = rows[2 * i + 1] & ((1 << hashbits) - 1);
    const int c = counts[k];
    nids[i] = c < depth ? c : depth;                                  // :164
}
__global__ __launch_bounds__(256)
void k_gh_fill(const int32_t* __restrict__ rows, int64_t n, int hashbits, int depth, int maxtimebits,
               const uint32_t* __restrict__ table, const int32_t* __restrict__ counts,
               const int64_t* __restrict__ off, int32_t* __restrict__ hits)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int time_ = rows[2 * i];
    const int k = rows[2 * i + 1] & ((1 << hashbits) - 1);            // :163
    const int c = counts[k];
    const int nid = c < depth ? c : depth;
    const uint32_t tmask = (1u << maxtimebits) - 1u;
    int4* o = reinterpret_cast<int4*>(hits) + off[i];
    for (int j = 0; j < nid; j++) {
        const uint32_t v = table[(int64_t)k * depth + j];
        o[j] = make_int4((int)(v >> maxtimebits) - 1, (int)(v & tmask) - time_, k, time_);   // :168-171
    }
}
extern "C" void afp_launch_gh_count(const int32_t* rows, int64_t n, int hashbits, int depth, const int32_t* counts,
                                    int64_t* nids, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_gh_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows, n, hashbits, depth, counts, nids);
}
extern "C" void afp_launch_gh_fill(const int32_t* rows, int64_t n, int hashbits, int depth, int maxtimebits,
                                   const uint32_t* table, const int32_t* counts, const int64_t* off, int32_t* hits,
                                   hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_gh_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows, n, hashbits, depth, maxtimebits, table, counts, off, hits);
}

extern "C" void afp_launch_tb_count(const TableArgs* a, hipStream_t st)
{
    if (a->nrows > 0) hipLaunchKernelGGL(k_tb_count, dim3((unsigned)((a->nrows + 255) / 256)), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_tb_scatter(const TableArgs* a, hipStream_t st)
{
    if (a->nrows > 0) hipLaunchKernelGGL(k_tb_scatter, dim3((unsigned)((a->nrows + 255) / 256)), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_tb_fill(const TableArgs* a, hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_fill, dim3((unsigned)(((1u << a->hashbits) + 255) / 256)), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_tb_fill_big(const TableArgs* a, hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_fill_big, dim3(1024), dim3(256), 0, st, *a);
}
