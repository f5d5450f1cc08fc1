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
// ---- row f4, remaining modes: the hit rows Matcher._exact_match_counts / _unique_match_hashes / _calculate_time_ranges
// (audfprint_match.py:149-239) select -- `allids == id` and `abs(alltimes - mode) <= window` -- for MANY (id, mode) queries
// in one pass over the hits resident in HBM.  rank[id] = position of the id among the wanted ids (-1: not wanted); the
// queries of wanted id r are qstart[r] .. qstart[r + 1]; query q keeps hits with qlo[q] <= skew <= qhi[q].  Pass 0 counts
// per query, pass 1 (cursor zeroed, off = exclusive scan of the counts) writes (orig_time, hash) rows -- in no particular order:
// the consumers take np.unique / np.sort of them.
__global__ __launch_bounds__(256)
void k_vote_select(const int4* __restrict__ hits, int64_t n, int nid, const int32_t* __restrict__ rank, const int32_t* __restrict__ qstart,
                   const int32_t* __restrict__ qlo, const int32_t* __restrict__ qhi, int32_t* __restrict__ cursor,
                   const int64_t* __restrict__ off, int2* __restrict__ out, int fill)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 h = hits[i];
    if (h.x < 0 || h.x >= nid) return;
    const int r = rank[h.x];
    if (r < 0) return;
    for (int q = qstart[r]; q < qstart[r + 1]; q++) {
        if (h.y >= qlo[q] && h.y <= qhi[q]) {
            const int k = atomicAdd(&cursor[q], 1);
            if (fill) out[off[q] + k] = make_int2(h.w, h.z);         // (orig_time, hash)
        }
    }
}
extern "C" void afp_launch_vote_select(const int32_t* hits, int64_t n, int nid, const int32_t* rank, const int32_t* qstart, const int32_t* qlo,
                                       const int32_t* qhi, int32_t* cursor, const int64_t* off, int32_t* out, int fill, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_vote_select, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const int4*>(hits), n, nid,
                                  rank, qstart, qlo, qhi, cursor, off, reinterpret_cast<int2*>(out), fill);
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

// ---- row f4 (second half): the counting steps of Matcher._best_count_ids / _approx_match_counts ----
// (audfprint_match.py:124-147, 241-312) over the hit rows [id, skew, hash, time] still resident in HBM.
// np.unique + np.bincount of the ids become one atomic histogram over the dense id range plus an ordered
// compaction; the per-id np.bincount(alltimes[allids == id]) loop becomes one pass that bins every hit of a
// wanted id into that id's row of a [nids][width] histogram.
__global__ __launch_bounds__(256)
void k_vote_count(const int4* __restrict__ hits, int64_t n, int nid, int32_t* __restrict__ idcount, int32_t* __restrict__ misc)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int mn = 0x7fffffff, mx = -0x7fffffff - 1, bad = 0, mo = -0x7fffffff - 1;
    if (i < n) {
        const int4 h = hits[i];
        if (h.x >= 0 && h.x < nid) atomicAdd(&idcount[h.x], 1); else bad = 1;
        mn = h.y; mx = h.y; mo = h.w;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        mn = min(mn, __shfl_xor(mn, s));
        mx = max(mx, __shfl_xor(mx, s));
        mo = max(mo, __shfl_xor(mo, s));
        bad |= __shfl_xor(bad, s);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&misc[0], mn);
        atomicMax(&misc[1], mx);
        if (bad) atomicOr(&misc[2], 1);
        atomicMax(&misc[4], mo);                                      // np.amax(allotimes), audfprint_match.py:157
    }
}
// ids with a non-zero count, ascending (= np.unique(allids)) and their counts (= np.bincount(allids)[ids]); one workgroup
__global__ __launch_bounds__(1024)
void k_vote_compact(const int32_t* __restrict__ idcount, int nid, int32_t* __restrict__ ids, int32_t* __restrict__ cnts,
                    int32_t* __restrict__ misc)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nid; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int c = i < nid ? idcount[i] : 0;
        const unsigned long long m = __ballot(c != 0);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        if (c != 0) { ids[carry + woff + before] = i; cnts[carry + woff + before] = c; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + __popcll(m);
        __syncthreads();
    }
    if (threadIdx.x == 0) misc[3] = carry_s;
}
__global__ __launch_bounds__(256)
void k_vote_setrank(const int32_t* __restrict__ ids, int nids, int nid, int32_t* __restrict__ rank)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nids && ids[i] >= 0 && ids[i] < nid) rank[ids[i]] = i;
}
__global__ __launch_bounds__(256)
void k_vote_hist(const int4* __restrict__ hits, int64_t n, int nid, const int32_t* __restrict__ rank, int mintime, int width,
                 int32_t* __restrict__ hist)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 h = hits[i];
    if (h.x < 0 || h.x >= nid) return;
    const int r = rank[h.x];
    if (r >= 0) atomicAdd(&hist[(int64_t)r * width + (h.y - mintime)], 1);
}
extern "C" void afp_launch_vote_count(const int32_t* hits, int64_t n, int nid, int32_t* idcount, int32_t* misc, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_vote_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int4*)hits, n, nid, idcount, misc);
}
extern "C" void afp_launch_vote_compact(const int32_t* idcount, int nid, int32_t* ids, int32_t* cnts, int32_t* misc, hipStream_t st)
{
    hipLaunchKernelGGL(k_vote_compact, dim3(1), dim3(1024), 0, st, idcount, nid, ids, cnts, misc);
}
extern "C" void afp_launch_vote_setrank(const int32_t* ids, int nids, int nid, int32_t* rank, hipStream_t st)
{
    if (nids > 0) hipLaunchKernelGGL(k_vote_setrank, dim3((unsigned)((nids + 255) / 256)), dim3(256), 0, st, ids, nids, nid, rank);
}
extern "C" void afp_launch_vote_hist(const int32_t* hits, int64_t n, int nid, const int32_t* rank, int mintime, int width,
                                     int32_t* hist, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_vote_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const int4*)hits, n, nid, rank, mintime, width, hist);
}

// ---- exclusive scan of the per-bucket row counts (2^hashbits int64 values) over many workgroups: block sums, a
// one-workgroup scan of those, then every block scans its own slice -- three launches of a few microseconds where the
// single-workgroup k_excl_scan64 walks a million entries in a thousand barrier-separated steps (~1 ms per store)
#define SCAN_BLK 2048                                  // values per workgroup: 256 threads x 8
__device__ __forceinline__ long long wave_incl_scan_ll(long long x, int lane)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int lo = __shfl_up((int)(x & 0xFFFFFFFFll), s);
        const int hi = __shfl_up((int)(x >> 32), s);
        const long long y = ((long long)hi << 32) | (unsigned int)lo;
        if (lane >= s) x += y;
    }
    return x;
}
__global__ __launch_bounds__(256)
void k_scan_blocksum(const int64_t* __restrict__ in, int64_t* __restrict__ bsum, int n)
{
    __shared__ long long ws[4];
    const int base = blockIdx.x * SCAN_BLK + threadIdx.x * 8;
    long long s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += (base + j < n) ? in[base + j] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int lo = __shfl_xor((int)(s & 0xFFFFFFFFll), d);
        const int hi = __shfl_xor((int)(s >> 32), d);
        s += ((long long)hi << 32) | (unsigned int)lo;
    }
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// one workgroup: exclusive scan of the block sums in place (nb <= 1024 * 16); bsum[nb] <- grand total
__global__ __launch_bounds__(1024)
void k_scan_bsums(int64_t* __restrict__ bsum, int nb)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nb; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const long long v = (i < nb) ? bsum[i] : 0;
        const long long x = wave_incl_scan_ll(v, lane);
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const long long carry = carry_s;
        if (i < nb) bsum[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry_s;
}
__global__ __launch_bounds__(256)
void k_scan_apply(const int64_t* __restrict__ in, const int64_t* __restrict__ bsum, int64_t* __restrict__ out, int n, int nb)
{
    __shared__ long long ws[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * SCAN_BLK + threadIdx.x * 8;
    long long v[8], t = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = (base + j < n) ? in[base + j] : 0; t += v[j]; }
    const long long incl = wave_incl_scan_ll(t, lane);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    long long off = bsum[blockIdx.x] + incl - t;
    for (int w = 0; w < wave; w++) off += ws[w];
#pragma unroll
    for (int j = 0; j < 8; j++) { if (base + j < n) out[base + j] = off; off += v[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}
// in[n] -> out[n + 1] (exclusive offsets, out[n] = total); scratch: int64[(n + SCAN_BLK - 1) / SCAN_BLK + 1]
extern "C" void afp_launch_excl_scan64_wide(const int64_t* in, int64_t* out, int n, int64_t* scratch, hipStream_t st)
{
    const int nb = (n + SCAN_BLK - 1) / SCAN_BLK;
    hipLaunchKernelGGL(k_scan_blocksum, dim3(nb), dim3(256), 0, st, in, scratch, n);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, st, scratch, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, in, scratch, out, n, nb);
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

// ---- row f1, second half: HashTable.merge (hash_table.py:291-323) over device-resident tables ----------
// For every bucket the other table uses: allvals = r_[table[k, :counts[k]], other[k, :ocounts[k]] + idoffset]
// (:304-305; a slice past the row length stops at the row length, so the two parts hold min(count, depth)
// entries).  If it fits (:315-321) it is stored and counts[k] = len(allvals); if not (:306-314) the reference
// draws np.random.permutation(allvals)[:depth] -- an RNG call the host has to make, in ascending bucket order --
// so the bucket is only listed here (counts[k] += ocounts[k] is done, the row is patched later by
// k_tb_patch) and k_tb_merge_gather hands the host its allvals.
__global__ __launch_bounds__(256)
void k_tb_merge(uint32_t* __restrict__ table, int32_t* __restrict__ counts, const uint32_t* __restrict__ otable,
                const int32_t* __restrict__ ocounts, const int64_t* __restrict__ ooff, int hashbits, int depth, int odepth,
                uint32_t idoffset, int32_t* __restrict__ ovlist, int32_t* __restrict__ ovcnt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= (1 << hashbits)) return;
    const int oc = ocounts[k];
    if (oc == 0) return;                                           // np.nonzero(ht.counts), :302
    const int c = counts[k];
    const int n1 = c < depth ? c : depth;
    const int n2 = oc < odepth ? oc : odepth;
    if (n1 + n2 > depth) {
        ovlist[atomicAdd(ovcnt, 1)] = k;
        counts[k] = c + oc;                                        // :314
        return;
    }
    uint32_t* row = table + (int64_t)k * depth + n1;
    // the other table's row: dense [k][odepth], or PACKED (ooff: only the filled prefixes, bucket after bucket)
    const uint32_t* orow = ooff ? otable + ooff[k] : otable + (int64_t)k * odepth;
    for (int j = 0; j < n2; j++) row[j] = orow[j] + idoffset;      // :320
    counts[k] = n1 + n2;                                           // :321
}

// allvals of the listed (over-full) buckets, one wavefront per bucket: out[i][0 .. nvals[i])
__global__ __launch_bounds__(64)
void k_tb_merge_gather(const uint32_t* __restrict__ table, const int32_t* __restrict__ counts_before_n1,
                       const uint32_t* __restrict__ otable, const int32_t* __restrict__ ocounts, const int64_t* __restrict__ ooff,
                       int depth, int odepth, uint32_t idoffset, const int32_t* __restrict__ buckets, int nb,
                       uint32_t* __restrict__ out, int32_t* __restrict__ nvals)
{
    const int i = blockIdx.x;
    if (i >= nb) return;
    const int k = buckets[i];
    const int oc = ocounts[k];
    // counts[k] was already advanced by oc (k_tb_merge): the count before the merge is counts[k] - oc
    const int c = counts_before_n1[k] - oc;
    const int n1 = c < depth ? c : depth;
    const int n2 = oc < odepth ? oc : odepth;
    uint32_t* o = out + (int64_t)i * (depth + odepth);
    const uint32_t* orow = ooff ? otable + ooff[k] : otable + (int64_t)k * odepth;
    for (int j = threadIdx.x; j < n1; j += 64) o[j] = table[(int64_t)k * depth + j];
    for (int j = threadIdx.x; j < n2; j += 64) o[n1 + j] = orow[j] + idoffset;
    if (threadIdx.x == 0) nvals[i] = n1 + n2;
}

// ---- the PACKED form of a table: only the slots store / merge can have written ---------------------------------------
// Neither HashTable.store (hash_table.py:115-131) nor merge (:304-321) ever writes table[k, j] for j >= min(counts[k], depth),
// and a reader (get_hits :164, merge :304-305) never looks there.  A 12 500-clip table is 7.7 % full: the filled prefixes,
// bucket after bucket, are 32 MB where the rows are 420 -- this is what leaves the device (afp_table_download_filled) and what
// crosses xGMI to the merging rank (afp_table_merge_packed*).  len[k] = min(counts[k], depth) -> exclusive scan -> gather.
__global__ __launch_bounds__(256)
void k_tb_pack_len(const int32_t* __restrict__ counts, int hashbits, int depth, int64_t* __restrict__ len)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= (1 << hashbits)) return;
    const int c = counts[k];
    len[k] = c < 0 ? 0 : c < depth ? c : depth;
}
// 16 lanes per bucket: 64 contiguous bytes of the row per step, contiguous writes
__global__ __launch_bounds__(256)
void k_tb_pack_gather(const uint32_t* __restrict__ table, const int64_t* __restrict__ off, int hashbits, int depth,
                      uint32_t* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(t >> 4), l = (int)(t & 15);
    if (k >= (1 << hashbits)) return;
    const int64_t o = off[k];
    const int n = (int)(off[k + 1] - o);
    const uint32_t* row = table + (int64_t)k * depth;
    for (int j = l; j < n; j += 16) out[o + j] = row[j];
}
extern "C" void afp_launch_tb_pack_len(const int32_t* counts, int hashbits, int depth, int64_t* len, hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_pack_len, dim3((unsigned)(((1u << hashbits) + 255) / 256)), dim3(256), 0, st, counts, hashbits, depth, len);
}
extern "C" void afp_launch_tb_pack_gather(const uint32_t* table, const int64_t* off, int hashbits, int depth, uint32_t* out, hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_pack_gather, dim3((unsigned)((((int64_t)16 << hashbits) + 255) / 256)), dim3(256), 0, st, table, off, hashbits,
                       depth, out);
}

// table[bucket, slot] = value for host-decided writes (replayed random replacements of store(), permuted rows of
// merge()); the host passes every (bucket, slot) at most once
__global__ __launch_bounds__(256)
void k_tb_patch(uint32_t* __restrict__ table, int depth, const int32_t* __restrict__ patches, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    table[(int64_t)patches[3 * i] * depth + patches[3 * i + 1]] = (uint32_t)patches[3 * i + 2];
}

// What HashTable.merge into an EMPTY table leaves of a bucket's count: allvals = ht.table[k, :ht.counts[k]] holds
// min(ht.counts[k], ht.depth) entries, it fits, and counts[k] = len(allvals) (hash_table.py:304-305, 315-321).  The parent
// of `new --ncores N` receives every worker's table that way -- core 0's included (audfprint.py:226-235); a rank that
// uses its OWN table as the merge base clips its counts with this first and is then exactly that parent.
__global__ __launch_bounds__(256)
void k_tb_clip_counts(int32_t* __restrict__ counts, int hashbits, int depth)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= (1 << hashbits)) return;
    const int c = counts[k];
    if (c > depth) counts[k] = depth;
}

extern "C" void afp_launch_tb_clip_counts(int32_t* counts, int hashbits, int depth, hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_clip_counts, dim3((unsigned)(((1u << hashbits) + 255) / 256)), dim3(256), 0, st, counts, hashbits, depth);
}
extern "C" void afp_launch_tb_merge(uint32_t* table, int32_t* counts, const uint32_t* otable, const int32_t* ocounts, const int64_t* ooff,
                                    int hashbits, int depth, int odepth, uint32_t idoffset, int32_t* ovlist, int32_t* ovcnt,
                                    hipStream_t st)
{
    hipLaunchKernelGGL(k_tb_merge, dim3((unsigned)(((1u << hashbits) + 255) / 256)), dim3(256), 0, st, table, counts, otable,
                       ocounts, ooff, hashbits, depth, odepth, idoffset, ovlist, ovcnt);
}
extern "C" void afp_launch_tb_merge_gather(const uint32_t* table, const int32_t* counts, const uint32_t* otable,
                                           const int32_t* ocounts, const int64_t* ooff, int depth, int odepth, uint32_t idoffset,
                                           const int32_t* buckets, int nb, uint32_t* out, int32_t* nvals, hipStream_t st)
{
    if (nb > 0) hipLaunchKernelGGL(k_tb_merge_gather, dim3((unsigned)nb), dim3(64), 0, st, table, counts, otable, ocounts, ooff, depth,
                                   odepth, idoffset, buckets, nb, out, nvals);
}
extern "C" void afp_launch_tb_patch(uint32_t* table, int depth, const int32_t* patches, int64_t n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_tb_patch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, depth, patches, n);
}
