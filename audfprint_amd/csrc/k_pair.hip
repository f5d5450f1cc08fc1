// k_pair.hip -- K4..K8: peak pairing, hash packing, per-frame sort / cross-shift merge +
// de-dup, CSR compaction (gfx950).  All integer / bit work, HBM- and latency-bound; the
// per-frame 256-bit peak masks written by k_scan are the only input.
//
// Replaces Analyzer.peaks2landmarks (audfprint_analyze.py:310-343), landmarks2hashes
// (:81-96) and the uint64 unique/sort of wavfile2hashes (:414-422); the peak list of
// find_peaks (:303-308) is produced from the same masks.
//
// Ordering argument: within one unit the hashes of frame `col` sorted by
// (f1 asc, then hash asc inside one source peak) are globally sorted, because f1 is the top 8
// hash bits and a frame holds each bin at most once; distinct landmarks of one unit pack to
// distinct hashes.  So: sort inside each source peak (<= fanout items), and for shifts > 1 do
// an S-way merge with de-dup of the per-shift lists of the same (clip, col).
#include <hip/hip_runtime.h>
#include "afp_common.h"

__device__ __forceinline__ unsigned long long window_word(int q, int lo, int hi)
{
    int l = lo - 64 * q, h = hi - 64 * q;
    if (l < 0) l = 0;
    if (h > 63) h = 63;
    if (l > h) return 0ull;
    return (~0ull << l) & (~0ull >> (63 - h));
}

// K4: one thread per (unit, col).  The workgroup stages the 256-bit masks of its COL_CHUNK frames
// plus the targetdt look-ahead halo in LDS (word-major, so a wavefront's reads are conflict-free).
__global__ __launch_bounds__(COL_CHUNK)
void k_pair(PairArgs A)
{
    extern __shared__ uint64_t sm[];               // [4][NF] mask words, then [NF] "any" flags (as uint32)
    const int u = A.cblk_unit[blockIdx.x];
    const int t0 = A.cblk_t0[blockIdx.x];
    const int T = A.unit_T[u];
    const int64_t fb = A.unit_fbase[u];
    const int NF = COL_CHUNK + A.targetdt;
    uint32_t* any = reinterpret_cast<uint32_t*>(sm + 4 * NF);
    const int avail = min(NF, T - t0);
    for (int f = threadIdx.x; f < NF; f += COL_CHUNK) {
        uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        if (f < avail) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(A.masks + (fb + t0 + f) * 4);
            ulonglong2 a = p[0], b = p[1];
            w0 = a.x; w1 = a.y; w2 = b.x; w3 = b.y;
        }
        sm[f] = w0; sm[NF + f] = w1; sm[2 * NF + f] = w2; sm[3 * NF + f] = w3;
        any[f] = (w0 | w1 | w2 | w3) != 0ull;
    }
    __syncthreads();
    const int lc = threadIdx.x;
    const int col = t0 + lc;
    if (col >= T) return;
    const int64_t g = fb + col;
    int n_out = 0;
    uint32_t* gout = A.hslots + g * (int64_t)A.slot;
    // working list of this thread: in LDS when it is small (odd stride: conflict-free across lanes,
    // and the insertion sort never touches HBM), else directly in the output slot
    uint32_t* lists = any + NF;
    uint32_t* out = A.lds_lists ? lists + (size_t)lc * (A.slot | 1) : gout;
    if (any[lc]) {
        const int dmax = min(T - col, A.targetdt);                        // :331-332 (scols <= T; frames past the last peak are empty)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t w = sm[q * NF + lc];
            while (w) {
                const int f1 = 64 * q + __ffsll((long long)w) - 1;
                w &= w - 1;
                const int lo = f1 - A.targetdf + 1, hi = f1 + A.targetdf - 1;   // abs(f2 - f1) < targetdf, :335
                const uint64_t win0 = window_word(0, lo, hi), win1 = window_word(1, lo, hi);
                const uint64_t win2 = window_word(2, lo, hi), win3 = window_word(3, lo, hi);
                const int seg0 = n_out;
                int np = 0;
                for (int dt = A.mindt; dt < dmax && np < A.fanout; dt++) {
                    if (!any[lc + dt]) continue;
                    uint64_t hit[4];
                    hit[0] = sm[lc + dt] & win0;
                    hit[1] = sm[NF + lc + dt] & win1;
                    hit[2] = sm[2 * NF + lc + dt] & win2;
                    hit[3] = sm[3 * NF + lc + dt] & win3;
                    if ((hit[0] | hit[1] | hit[2] | hit[3]) == 0ull) continue;
#pragma unroll
                    for (int q2 = 0; q2 < 4; q2++) {
                        uint64_t w2 = hit[q2];
                        while (w2 && np < A.fanout) {
                            const int f2 = 64 * q2 + __ffsll((long long)w2) - 1;
                            w2 &= w2 - 1;
                            if (A.lm_mode) {                      // raw landmark, emission order (:339-340)
                                out[n_out] = (uint32_t)f1 | ((uint32_t)f2 << 8) | ((uint32_t)dt << 16);
                            } else {
                                const uint32_t h = ((uint32_t)(f1 & 0xFF) << 12)          // :92-95
                                                 | ((uint32_t)((f2 - f1) & 0x3F) << 6)
                                                 | (uint32_t)(dt & 0x3F);
                                // insertion into the sorted run of this source peak
                                int k = n_out;
                                while (k > seg0 && out[k - 1] > h) { out[k] = out[k - 1]; k--; }
                                out[k] = h;
                            }
                            n_out++;
                            np++;
                        }
                    }
                }
            }
        }
    }
    if (A.lds_lists) for (int k = 0; k < n_out; k++) gout[k] = out[k];
    A.hcnt[g] = n_out;
}

// ------------------------------------------------------------------------------------------
// K4': fused pairing + hash packing + cross-shift merge/de-dup + sort, wavefront-cooperative.
// Workgroup = 4 wavefronts on `ch` consecutive columns of ONE clip (all S shifts); the masks of
// those columns plus the look-ahead halo are staged word-major in LDS.  A wavefront takes
// ch/4 columns; for every source peak its 64 lanes test 64 target frames at once (window
// popcount -> DPP prefix sum -> the first `fanout` in (frame, bin) order), appending packed
// hashes to a per-wavefront LDS list; the list of one (clip, col) is then de-duplicated and
// ranked (counting sort by broadcast compares) straight into the output slot.
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ int dpp_add(int v)
{
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v = dpp_add<0x111>(v);          // row_shr:1
    v = dpp_add<0x112>(v);          // row_shr:2
    v = dpp_add<0x114>(v);          // row_shr:4
    v = dpp_add<0x118>(v);          // row_shr:8
    v = dpp_add<0x142, 0xA>(v);     // row_bcast:15
    v = dpp_add<0x143, 0xC>(v);     // row_bcast:31
    return v;
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffull));
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256)
void k_pairmerge(PairMergeArgs A)
{
    extern __shared__ uint64_t sm[];               // [S][4][NF] mask words | [S][NF] any flags (u32) | [4][oslot] lists (u32)
    const int clip = A.pblk_clip[blockIdx.x];
    const int t0 = A.pblk_t0[blockIdx.x];
    const int S = A.S;
    const int NF = A.ch + A.targetdt;
    uint32_t* any = reinterpret_cast<uint32_t*>(sm + (size_t)S * 4 * NF);
    uint32_t* lists = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(any + (size_t)S * NF) + 15) & ~(uintptr_t)15);
    const int lstride = ((A.oslot + 3) & ~3) + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* list = lists + (size_t)wave * lstride;
    const int Tm = A.clip_T0[clip];
    const int64_t mfb = A.clip_mfbase[clip];
    __shared__ int Ts[16];                         // frames of each shift's unit (loop-invariant: keep out of the column loop)
    if (threadIdx.x < 16) Ts[threadIdx.x] = threadIdx.x < S ? A.unit_T[clip * S + threadIdx.x] : 0;
    // ---- stage masks
    for (int s = 0; s < S; s++) {
        const int u = clip * S + s;
        const int T = A.unit_T[u];
        const int64_t fb = A.unit_fbase[u];
        const int avail = min(NF, T - t0);
        uint64_t* ms = sm + (size_t)s * 4 * NF;
        for (int f = threadIdx.x; f < NF; f += 256) {
            uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (f < avail) {
                const ulonglong2* p = reinterpret_cast<const ulonglong2*>(A.masks + (fb + t0 + f) * 4);
                ulonglong2 a = p[0], b = p[1];
                w0 = a.x; w1 = a.y; w2 = b.x; w3 = b.y;
            }
            ms[f] = w0; ms[NF + f] = w1; ms[2 * NF + f] = w2; ms[3 * NF + f] = w3;
            any[s * NF + f] = (w0 | w1 | w2 | w3) != 0ull;
        }
    }
    __syncthreads();
    const int wc = A.ch >> 2;                       // columns per wavefront
    const int cbase = wave * wc;
    const int F = A.fanout;
    for (int c0 = 0; c0 < wc; c0 += 64) {
        // which of my (up to 64) columns have a source peak in any shift?
        const int lc_l = cbase + c0 + lane;
        bool nz = false;
        if (c0 + lane < wc && t0 + lc_l < Tm)
            for (int s = 0; s < S; s++) nz = nz || (any[s * NF + lc_l] != 0);
        unsigned long long colmask = __ballot(nz);
        while (colmask) {
            const int ci = __ffsll((long long)colmask) - 1;
            colmask &= colmask - 1;
            const int lc = cbase + c0 + ci;          // column inside the staged window (uniform)
            const int col = t0 + lc;
            int M = 0;
            for (int s = 0; s < S; s++) {
                const int T = Ts[s];
                if (col >= T || !any[s * NF + lc]) continue;
                const uint64_t* ms = sm + (size_t)s * 4 * NF;
                const uint32_t* an = any + s * NF;
                const int dmax = min(T - col, A.targetdt);               // :331-332
                unsigned long long wsrc[4];
#pragma unroll
                for (int q = 0; q < 4; q++) wsrc[q] = ms[q * NF + lc];     // four reads in flight, then made uniform
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned long long w = uniform64(wsrc[q]);
                    while (w) {
                        const int f1 = 64 * q + __ffsll((long long)w) - 1;
                        w &= w - 1;
                        const int lo = f1 - A.targetdf + 1, hi = f1 + A.targetdf - 1;   // :335
                        int np = 0;
                        if (A.targetdf <= 32) {
                            // the +-(targetdf-1) window spans at most 63 bins: pull it out of the 256-bit mask
                            // as ONE 64-bit word (two adjacent words, funnel-shifted) -- bit i = bin lo_c + i
                            const int lo_c = lo < 0 ? 0 : lo, hi_c = hi > 255 ? 255 : hi;
                            const int q0 = lo_c >> 6, sh = lo_c & 63;
                            const unsigned long long wmask = ~0ull >> (63 - (hi_c - lo_c));
                            const uint64_t* m0 = ms + q0 * NF + lc;
                            const uint64_t* m1 = ms + (q0 < 3 ? q0 + 1 : q0) * NF + lc;
                            for (int d0 = A.mindt; d0 < dmax && np < F; d0 += 64) {
                                const int dt = d0 + lane;
                                unsigned long long wv = 0;
                                if (dt < dmax && an[lc + dt]) {
                                    const unsigned long long a = m0[dt];
                                    const unsigned long long b = (sh != 0 && q0 < 3) ? m1[dt] : 0ull;
                                    wv = ((a >> sh) | (sh ? (b << (64 - sh)) : 0ull)) & wmask;
                                }
                                const int c = __popcll(wv);
                                const int incl = wave_incl_scan(c);
                                const int total = __builtin_amdgcn_readlane(incl, 63);
                                if (total == 0) continue;
                                int r = np + incl - c;                   // rank of my first hit in (frame, bin) order
                                const uint32_t hbase = ((uint32_t)(f1 & 0xFF) << 12) | (uint32_t)(dt & 0x3F);
                                for (unsigned long long bb = wv; bb != 0ull && r < F; bb &= bb - 1) {
                                    const int f2 = lo_c + __ffsll((long long)bb) - 1;
                                    list[M + r] = hbase | ((uint32_t)((f2 - f1) & 0x3F) << 6);    // :92-95
                                    r++;
                                }
                                np = min(F, np + total);
                            }
                        } else
                        for (int d0 = A.mindt; d0 < dmax && np < F; d0 += 64) {
                            const int dt = d0 + lane;
                            unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
                            if (dt < dmax && an[lc + dt]) {
                                b0 = ms[lc + dt] & window_word(0, lo, hi);
                                b1 = ms[NF + lc + dt] & window_word(1, lo, hi);
                                b2 = ms[2 * NF + lc + dt] & window_word(2, lo, hi);
                                b3 = ms[3 * NF + lc + dt] & window_word(3, lo, hi);
                            }
                            const int c = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
                            const int incl = wave_incl_scan(c);
                            const int total = __builtin_amdgcn_readlane(incl, 63);
                            if (total == 0) continue;
                            int r = np + incl - c;                       // rank of my first hit in (frame, bin) order
                            const uint32_t hbase = ((uint32_t)(f1 & 0xFF) << 12) | (uint32_t)(dt & 0x3F);
#define AFP_EMIT(BQ, QQ)                                                                           \
                            for (unsigned long long bb = (BQ); bb != 0ull && r < F; bb &= bb - 1) {                \
                                const int f2 = 64 * (QQ) + __ffsll((long long)bb) - 1;                             \
                                list[M + r] = hbase | ((uint32_t)((f2 - f1) & 0x3F) << 6);    /* :92-95 */         \
                                r++;                                                                               \
                            }
                            AFP_EMIT(b0, 0)
                            AFP_EMIT(b1, 1)
                            AFP_EMIT(b2, 2)
                            AFP_EMIT(b3, 3)
#undef AFP_EMIT
                            np = min(F, np + total);
                        }
                        M += np;
                    }
                }
            }
            // ---- de-dup + rank the M hashes of (clip, col) into the output slot
            // (the list is padded to a multiple of 4 with 0xFFFFFFFF so it can be swept 16 bytes at a time)
            if (lane < 4 && (M & 3) != 0 && lane >= (M & 3)) list[(M & ~3) + lane] = 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t mg = mfb + col;
            uint32_t* out = A.oslots + mg * (int64_t)A.oslot;
            const uint4* list4 = reinterpret_cast<const uint4*>(list);
            const int M4 = (M + 3) >> 2;
            int nuniq = M;
            if (A.dedupe && M > 1) {
                // mark later duplicates (bit 31); hashes use 20 bits
                int ndup = 0;
                for (int i0 = 0; i0 < M; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t v = i < M ? list[i] : 0xFFFFFFFEu;
                    bool dup = false;
                    const int jend = min(M4, (i0 + 64) >> 2);
                    for (int j4 = 0; j4 < jend; j4++) {
                        const uint4 w = list4[j4];
                        const int j = 4 * j4;
                        dup = dup || (j < i && (w.x & 0x7FFFFFFFu) == v) || (j + 1 < i && (w.y & 0x7FFFFFFFu) == v)
                                  || (j + 2 < i && (w.z & 0x7FFFFFFFu) == v) || (j + 3 < i && (w.w & 0x7FFFFFFFu) == v);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i < M && dup) list[i] = v | 0x80000000u;
                    ndup += __popcll(__ballot(i < M && dup));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                nuniq = M - ndup;
            }
            for (int i0 = 0; i0 < M; i0 += 64) {
                const int i = i0 + lane;
                const uint32_t v = i < M ? list[i] : 0xFFFFFFFFu;
                int rank = 0;
                for (int j4 = 0; j4 < M4; j4++) {                        // flagged duplicates / padding are > every hash
                    const uint4 w = list4[j4];
                    rank += (w.x < v ? 1 : 0) + (w.y < v ? 1 : 0) + (w.z < v ? 1 : 0) + (w.w < v ? 1 : 0);
                }
                if (i < M && !(v & 0x80000000u)) out[rank] = v;
            }
            if (lane == 0) A.ocnt[mg] = nuniq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// K4'' (one shift, |df| window <= 63 bins, no field wrap, K <= 8, fanout <= 8 -- the default parameters):
// one LANE per source peak.  A wavefront owns ch/4 consecutive columns: it scatters the (column, bin) of every
// source peak into an LDS list, then 64 peaks at a time walk the target frames dt = mindt.. together -- each lane
// pulls its own +-(targetdf-1) window out of the staged 256-bit masks as one funnel-shifted 64-bit word and takes
// set bits in (frame, bin) order until it has `fanout` pairs (:331-341).  The <= fanout hashes of a peak are
// ranked by counting, peaks of one column are consecutive lanes in ascending bin = ascending top hash bits, so a
// segmented prefix sum gives every hash its final position in the column's sorted slot.  ~4x fewer vector
// instructions than the wavefront-per-peak kernel above, which remains for every other parameter set.
#define PL_KMAX 8
__global__ __launch_bounds__(256)
void k_pairlane(PairMergeArgs A)
{
    extern __shared__ uint64_t sm[];               // [4][NF] mask words | per wavefront: plist | hlist | ccount | segb
    const int clip = A.pblk_clip[blockIdx.x];
    const int t0 = A.pblk_t0[blockIdx.x];
    const int NF = A.ch + A.targetdt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = A.unit_T[clip];                  // one shift: unit == clip
    const int64_t fb = A.unit_fbase[clip];
    const int64_t mfb = A.clip_mfbase[clip];
    const int F = A.fanout, Fs = A.fanout | 1;     // odd list stride: conflict-free across lanes
    const int wc = A.ch >> 2;                      // columns per wavefront (multiple of 64)
    {
        const int avail = min(NF, T - t0);
        for (int f = threadIdx.x; f < NF; f += 256) {
            uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (f < avail) {
                const ulonglong2* p = reinterpret_cast<const ulonglong2*>(A.masks + (fb + t0 + f) * 4);
                ulonglong2 a = p[0], b = p[1];
                w0 = a.x; w1 = a.y; w2 = b.x; w3 = b.y;
            }
            sm[f] = w0; sm[NF + f] = w1; sm[2 * NF + f] = w2; sm[3 * NF + f] = w3;
        }
    }
    __syncthreads();
    uint32_t* plist = reinterpret_cast<uint32_t*>(sm + (size_t)4 * NF) + (size_t)wave * (wc * PL_KMAX + 64 * Fs + 2 * wc);
    uint32_t* hlist = plist + wc * PL_KMAX;
    uint32_t* ccount = hlist + 64 * Fs;
    uint32_t* segb = ccount + wc;
    const int cbase = wave * wc;
    // ---- 1. the source peaks of my columns, in (column, bin) order
    int P = 0;
    for (int c0 = 0; c0 < wc; c0 += 64) {
        const int ci = c0 + lane;
        const int lc = cbase + ci;
        uint64_t w[4] = {0ull, 0ull, 0ull, 0ull};
        if (t0 + lc < T) {
#pragma unroll
            for (int q = 0; q < 4; q++) w[q] = sm[q * NF + lc];
        }
        const int cnt = __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
        const int incl = wave_incl_scan(cnt);
        int k = P + incl - cnt;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t x = w[q];
            while (x) {
                plist[k++] = ((uint32_t)ci << 8) | (uint32_t)(64 * q + __ffsll((long long)x) - 1);
                x &= x - 1;
            }
        }
        P += __builtin_amdgcn_readlane(incl, 63);
        ccount[ci] = 0;
        // every column of the clip belongs to exactly one wavefront of one workgroup: its count starts at zero here (the
        // non-empty ones are overwritten further down by this same wavefront) -- no memset in front of the kernel
        if (t0 + lc < T) A.ocnt[mfb + t0 + lc] = 0;
    }
    if (P == 0) return;                            // (no workgroup barrier below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 2. rounds of up to 64 peaks, evenly filled
    const int nr = (P + 63) >> 6;
    const int per = (P + nr - 1) / nr;
    for (int r = 0; r < nr; r++) {
        const int pi = r * per + lane;
        const bool active = lane < per && pi < P;
        const uint32_t e = active ? plist[pi] : 0u;
        const int ci = (int)(e >> 8), f1 = (int)(e & 255u);
        const int lc = cbase + ci, col = t0 + lc;
        const int dmax = active ? min(T - col, A.targetdt) : 0;              // :331-332
        const int lo = f1 - A.targetdf + 1, hi = f1 + A.targetdf - 1;       // abs(f2 - f1) < targetdf, :335
        const int lo_c = lo < 0 ? 0 : lo, hi_c = hi > 255 ? 255 : hi;
        const int q0 = lo_c >> 6, sh = lo_c & 63;
        const unsigned long long wmask = ~0ull >> (63 - (hi_c - lo_c));
        const bool useb = sh != 0 && q0 < 3;
        const uint64_t* m0 = sm + q0 * NF + lc;
        const uint64_t* m1 = m0 + (useb ? NF : 0);
        const int ish = useb ? 64 - sh : 0;
        uint32_t* hl = hlist + lane * Fs;
        const uint32_t hb = (uint32_t)(f1 & 0xFF) << 12;
        int np = 0;
        for (int dt = A.mindt; dt < A.targetdt; dt++) {
            const bool need = np < F && dt < dmax;
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
            if (need) {
                const unsigned long long a = m0[dt];
                const unsigned long long b = m1[dt];
                unsigned long long wv = ((a >> sh) | (useb ? (b << ish) : 0ull)) & wmask;   // bit i = bin lo_c + i
                while (wv != 0ull && np < F) {
                    const int f2 = lo_c + __ffsll((long long)wv) - 1;
                    wv &= wv - 1;
                    hl[np++] = hb | ((uint32_t)((f2 - f1) & 0x3F) << 6) | (uint32_t)(dt & 0x3F);   // :92-95
                }
            }
        }
        // ---- 3. position of my hashes inside the column's sorted slot
        const int incl = wave_incl_scan(np);
        const int X = incl - np;
        const int ci_prev = __shfl_up(ci, 1), ci_next = __shfl_down(ci, 1);
        const bool head = active && (lane == 0 || ci != ci_prev);
        const bool tail = active && (lane == per - 1 || pi == P - 1 || ci != ci_next);
        if (head) segb[ci] = (uint32_t)X;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int off = active ? (int)ccount[ci] + X - (int)segb[ci] : 0;
        uint32_t* out = A.oslots + (mfb + col) * (int64_t)A.oslot + off;
        for (int i = 0; i < np; i++) {
            const uint32_t v = hl[i];
            int rank = 0;
            for (int j = 0; j < np; j++) rank += hl[j] < v ? 1 : 0;
            out[rank] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tail) ccount[ci] = (uint32_t)(off + np);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- 4. hashes per column (columns without pairs keep the 0 of the memset)
    for (int c0 = 0; c0 < wc; c0 += 64) {
        const int ci = c0 + lane;
        const int col = t0 + cbase + ci;
        if (col < T) {
            const uint32_t n = ccount[ci];
            if (n) A.ocnt[mfb + col] = (int32_t)n;
        }
    }
}

// Ascending bitonic sort of one 32-bit key per lane across the wavefront (21 compare-exchange steps): the partner
// lane ^ j comes through DPP (j = 1, 2, 8), ds_swizzle (j = 4, 16) or a permute (j = 32) -- no LDS array traffic.
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v)
{
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                     // bit mode: xor 4
    else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (J == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                    // bit mode: xor 16
    else return (uint32_t)__shfl_xor((int)v, 32);
}
template <int K2, int J>
__device__ __forceinline__ uint32_t bitonic_step(uint32_t v, int lane)
{
    const uint32_t pv = lane_xor_u32<J>(v);
    const bool up = K2 >= 64 || (lane & K2) == 0, lower = (lane & J) == 0;
    return (lower == up) ? min(v, pv) : max(v, pv);
}
// the same network over 128 keys, two per lane (element index = lane + 64 r for register r): the j = 64 step is a
// compare inside the lane; in the k = 64 merges register 1 sorts descending (its element index has bit 6 set)
template <int K2, int J>
__device__ __forceinline__ void bitonic_step2(uint32_t& v0, uint32_t& v1, int lane)
{
    const uint32_t p0 = lane_xor_u32<J>(v0), p1 = lane_xor_u32<J>(v1);
    const bool lower = (lane & J) == 0;
    const bool up0 = K2 >= 64 || (lane & K2) == 0;
    const bool up1 = K2 == 64 ? false : up0;
    v0 = (lower == up0) ? min(v0, p0) : max(v0, p0);
    v1 = (lower == up1) ? min(v1, p1) : max(v1, p1);
}
__device__ __forceinline__ void wave_sort128_u32(uint32_t& v0, uint32_t& v1, int lane)
{
    bitonic_step2<2, 1>(v0, v1, lane);
    bitonic_step2<4, 2>(v0, v1, lane); bitonic_step2<4, 1>(v0, v1, lane);
    bitonic_step2<8, 4>(v0, v1, lane); bitonic_step2<8, 2>(v0, v1, lane); bitonic_step2<8, 1>(v0, v1, lane);
    bitonic_step2<16, 8>(v0, v1, lane); bitonic_step2<16, 4>(v0, v1, lane); bitonic_step2<16, 2>(v0, v1, lane); bitonic_step2<16, 1>(v0, v1, lane);
    bitonic_step2<32, 16>(v0, v1, lane); bitonic_step2<32, 8>(v0, v1, lane); bitonic_step2<32, 4>(v0, v1, lane); bitonic_step2<32, 2>(v0, v1, lane);
    bitonic_step2<32, 1>(v0, v1, lane);
    bitonic_step2<64, 32>(v0, v1, lane); bitonic_step2<64, 16>(v0, v1, lane); bitonic_step2<64, 8>(v0, v1, lane); bitonic_step2<64, 4>(v0, v1, lane);
    bitonic_step2<64, 2>(v0, v1, lane); bitonic_step2<64, 1>(v0, v1, lane);
    { const uint32_t lo = min(v0, v1), hi = max(v0, v1); v0 = lo; v1 = hi; }                      // k = 128, j = 64
    bitonic_step2<128, 32>(v0, v1, lane); bitonic_step2<128, 16>(v0, v1, lane); bitonic_step2<128, 8>(v0, v1, lane);
    bitonic_step2<128, 4>(v0, v1, lane); bitonic_step2<128, 2>(v0, v1, lane); bitonic_step2<128, 1>(v0, v1, lane);
}
__device__ __forceinline__ uint32_t wave_sort64_u32(uint32_t v, int lane)
{
    v = bitonic_step<2, 1>(v, lane);
    v = bitonic_step<4, 2>(v, lane); v = bitonic_step<4, 1>(v, lane);
    v = bitonic_step<8, 4>(v, lane); v = bitonic_step<8, 2>(v, lane); v = bitonic_step<8, 1>(v, lane);
    v = bitonic_step<16, 8>(v, lane); v = bitonic_step<16, 4>(v, lane); v = bitonic_step<16, 2>(v, lane); v = bitonic_step<16, 1>(v, lane);
    v = bitonic_step<32, 16>(v, lane); v = bitonic_step<32, 8>(v, lane); v = bitonic_step<32, 4>(v, lane); v = bitonic_step<32, 2>(v, lane);
    v = bitonic_step<32, 1>(v, lane);
    v = bitonic_step<64, 32>(v, lane); v = bitonic_step<64, 16>(v, lane); v = bitonic_step<64, 8>(v, lane); v = bitonic_step<64, 4>(v, lane);
    v = bitonic_step<64, 2>(v, lane); v = bitonic_step<64, 1>(v, lane);
    return v;
}


// K4-ms (SEVERAL shifts, |df| window <= 63 bins, <= 8 peaks per (shift, column), shifts x 8 <= 64): one LANE per source
// peak, like k_pairlane, for the multi-shift query / high-recall configuration (Analyzer.shifts = 4 at match time,
// audfprint.py:295-297; BASELINE configs[4]).  A wavefront owns ch/4 consecutive columns of ONE clip with the masks of all
// S shifts staged in LDS.  It lists the source peaks of its columns in (column, shift, bin) order, then works through
// them in column-aligned rounds of at most 64 peaks: every lane walks the target frames of ITS peak in ITS shift's
// masks (:331-341) and collects <= fanout packed hashes (:92-95); the hashes of a round are compacted into one LDS
// list in which every column owns a contiguous segment, and each segment -- the union over the shifts, which the
// reference concatenates and passes through np.unique / np.sort (:404-422) -- is de-duplicated and rank-sorted by
// broadcast compares straight into the column's output slot.
#define PLM_KMAX 8
__global__ __launch_bounds__(256)
void k_pairlane_ms(PairMergeArgs A)
{
    extern __shared__ uint64_t sm[];               // [S][4][NF] mask words | per wavefront: plist | hlist | list | colstart | xs
    const int clip = A.pblk_clip[blockIdx.x];
    const int t0 = A.pblk_t0[blockIdx.x];
    const int S = A.S;
    const int NF = A.ch + A.targetdt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Tm = A.clip_T0[clip];
    const int64_t mfb = A.clip_mfbase[clip];
    const int F = A.fanout, Fs = A.fanout | 1;     // odd list stride: conflict-free across lanes
    const int wc = A.ch >> 2;                      // columns per wavefront (<= 64)
    __shared__ int Ts[16];
    if (threadIdx.x < 16) Ts[threadIdx.x] = threadIdx.x < S ? A.unit_T[clip * S + threadIdx.x] : 0;
    for (int s = 0; s < S; s++) {
        const int u = clip * S + s;
        const int T = A.unit_T[u];
        const int64_t fb = A.unit_fbase[u];
        const int avail = min(NF, T - t0);
        uint64_t* ms = sm + (size_t)s * 4 * NF;
        for (int f = threadIdx.x; f < NF; f += 256) {
            uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
            if (f < avail) {
                const ulonglong2* p = reinterpret_cast<const ulonglong2*>(A.masks + (fb + t0 + f) * 4);
                ulonglong2 a = p[0], b = p[1];
                w0 = a.x; w1 = a.y; w2 = b.x; w3 = b.y;
            }
            ms[f] = w0; ms[NF + f] = w1; ms[2 * NF + f] = w2; ms[3 * NF + f] = w3;
        }
    }
    __syncthreads();
    const int Kc = A.oslot / (S * F);              // peaks a (shift, column) may hold (the scan's maxpksperframe)
    const int pcap = (wc * S * Kc + 3) & ~3;       // source peaks a wavefront may list
    const int scap = (S * Kc * F + 4 + 3) & ~3;    // one column's hashes (all shifts) + padding
    const int hcap = (64 * Fs + 3) & ~3, ccap = (wc + 1 + 65 + 3) & ~3;   // every sub-array starts 16-byte aligned
    uint32_t* plist = reinterpret_cast<uint32_t*>(sm + (size_t)S * 4 * NF) + (size_t)wave * (pcap + hcap + scap + ccap);
    uint32_t* hlist = plist + pcap;
    uint32_t* sl = hlist + hcap;
    uint32_t* colstart = sl + scap;
    uint32_t* xs = colstart + (wc + 1);
    const int cbase = wave * wc;
    // ---- 1. the source peaks of my columns in (column, shift, bin) order
    int P;
    {
        const int ci = lane;
        const int lc = cbase + ci;
        int tot = 0;
        if (ci < wc) {
            for (int s = 0; s < S; s++) {
                if (t0 + lc < Ts[s]) {
                    const uint64_t* ms = sm + (size_t)s * 4 * NF;
                    tot += __popcll(ms[lc]) + __popcll(ms[NF + lc]) + __popcll(ms[2 * NF + lc]) + __popcll(ms[3 * NF + lc]);
                }
            }
        }
        const int incl = wave_incl_scan(tot);
        int k = incl - tot;
        if (ci < wc) {
            colstart[ci] = (uint32_t)k;
            for (int s = 0; s < S; s++) {
                if (t0 + lc < Ts[s]) {
                    const uint64_t* ms = sm + (size_t)s * 4 * NF;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        uint64_t x = ms[q * NF + lc];
                        while (x) {
                            plist[k++] = ((uint32_t)ci << 12) | ((uint32_t)s << 8) | (uint32_t)(64 * q + __ffsll((long long)x) - 1);
                            x &= x - 1;
                        }
                    }
                }
            }
        }
        P = __builtin_amdgcn_readlane(incl, 63);
        if (lane == 0) colstart[wc] = (uint32_t)P;
    }
    if (P == 0) return;                            // (no workgroup barrier below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 2. column-aligned rounds of at most 64 peaks
    int c0 = 0;
    while (c0 < wc) {
        const int pbase = (int)colstart[c0];
        // the columns c0 .. c1-1 of this round: as many as keep the round within 64 peaks (a column holds <= S * 8 <= 64)
        const bool fits = lane >= c0 && lane < wc && (int)colstart[lane + 1] - pbase <= 64;
        int c1 = c0 + __popcll(__ballot(fits));
        if (c1 == c0) c1 = c0 + 1;                 // (cannot happen: guarded on the host)
        const int npk = min(64, (int)colstart[c1] - pbase);
        if (npk == 0) { c0 = c1; continue; }
        const bool active = lane < npk;
        const uint32_t e = active ? plist[pbase + lane] : 0u;
        const int ci = (int)(e >> 12), s = (int)((e >> 8) & 15u), f1 = (int)(e & 255u);
        const int lc = cbase + ci, col = t0 + lc;
        const int T = Ts[s];
        const int dmax = active ? min(T - col, A.targetdt) : 0;              // :331-332
        const int lo = f1 - A.targetdf + 1, hi = f1 + A.targetdf - 1;       // abs(f2 - f1) < targetdf, :335
        const int lo_c = lo < 0 ? 0 : lo, hi_c = hi > 255 ? 255 : hi;
        const int q0 = lo_c >> 6, sh = lo_c & 63;
        const unsigned long long wmask = ~0ull >> (63 - (hi_c - lo_c));
        const bool useb = sh != 0 && q0 < 3;
        const uint64_t* m0 = sm + (size_t)s * 4 * NF + q0 * NF + lc;
        const uint64_t* m1 = m0 + (useb ? NF : 0);
        const int ish = useb ? 64 - sh : 0;
        const uint32_t hb = (uint32_t)(f1 & 0xFF) << 12;
        // ---- 2a. which target frames hold pairs for my peak, and how many: one pass over dt in lockstep, no per-hit work.
        //      The frames that contribute are remembered as 6-bit entries of a register pair (at most F <= 16 of them: the pass
        //      stops counting a lane once it has F hits).  The per-hit loop this replaces ran, in every dt step, as long as the
        //      lane with the MOST hits in that frame needed -- with ~40 peaks in flight that was nearly always two rounds of
        //      twenty instructions for one or two lanes' benefit (r04: k_pairlane_ms 686 M VALU per C5 launch, 93 % VALU-bound).
        unsigned long long dl0 = 0ull, dl1 = 0ull;
        int ne6 = 0, hits = 0;
        for (int dt = A.mindt; dt < A.targetdt; dt++) {
            const bool need = hits < F && dt < dmax;
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
            const unsigned long long a = m0[dt];
            const unsigned long long b = m1[dt];
            const unsigned long long wv = need ? (((a >> sh) | (useb ? (b << ish) : 0ull)) & wmask) : 0ull;   // bit i = bin lo_c + i
            const int c = __popcll(wv);
            if (c) {
                const bool lowhalf = ne6 < 60;
                dl0 |= lowhalf ? ((unsigned long long)dt << (lowhalf ? ne6 : 0)) : 0ull;
                dl1 |= lowhalf ? 0ull : ((unsigned long long)dt << (lowhalf ? 0 : ne6 - 60));
                ne6 += 6;
                hits += c;
            }
        }
        const int np = hits < F ? hits : F;
        // ---- 3. where the hashes of every lane fall in (column, shift, bin) order: a column = a run of lanes
        const int incl = wave_incl_scan(np);
        const int X = incl - np;
        xs[lane] = (uint32_t)X;
        if (lane == 63) xs[64] = (uint32_t)incl;
        // ---- 2b. the hashes themselves, hit number h of every lane in the same step, straight into the round's list: the
        //      lane walks its remembered frames (dt ascending) and takes the set bits of each window in ascending bin order
        //      -- (dt, bin) order, :331-341 -- re-reading a frame's mask words only when it moves on to the next entry
        {
            unsigned long long wv = 0ull;
            int k6 = 0, dtc = 0;
            for (int h = 0;; h++) {
                const bool act = h < np;
                if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
                if (act && wv == 0ull) {
                    const bool lowhalf = k6 < 60;
                    dtc = (int)((lowhalf ? (dl0 >> (lowhalf ? k6 : 0)) : (dl1 >> (lowhalf ? 0 : k6 - 60))) & 63ull);
                    k6 += 6;
                    const unsigned long long a = m0[dtc];
                    const unsigned long long b = m1[dtc];
                    wv = ((a >> sh) | (useb ? (b << ish) : 0ull)) & wmask;
                }
                if (act) {
                    const int f2 = lo_c + __ffsll((long long)wv) - 1;
                    wv &= wv - 1;
                    hlist[X + h] = hb | ((uint32_t)((f2 - f1) & 0x3F) << 6) | (uint32_t)(dtc & 0x3F);   // :92-95
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- 4. per column: de-duplicate + rank into the output slot (hashes use 20 bits; bit 31 marks a later duplicate).
        //      The round's list `hlist` is read-only from here to the end of the round: a column's M hashes are the contiguous
        //      run hlist[a .. b) -- no per-column copy, no fences between columns (only the rare M > 128 column copies its run
        //      into the 16-byte aligned scratch list it flags duplicates in)
        const uint4* sl4 = reinterpret_cast<const uint4*>(sl);
        for (int c = c0; c < c1; c++) {
            const int la = (int)colstart[c] - pbase, lb = (int)colstart[c + 1] - pbase;     // lanes of this column
            const int a = (int)xs[la < 64 ? la : 64], b = (int)xs[lb < 64 ? lb : 64];
            const int M = b - a;
            if (M <= 0) continue;
            const int ccol = t0 + cbase + c;
            uint32_t* out = A.oslots + (mfb + ccol) * (int64_t)A.oslot;
            const int M4 = (M + 3) >> 2;
            if (M <= 64) {
                // the common case: one hash per lane, sorted in registers; duplicates are neighbours afterwards
                uint32_t v = lane < M ? hlist[a + lane] : 0xFFFFFFFFu;
                v = wave_sort64_u32(v, lane);
                const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false);     // wave_shr:1
                const bool keep = lane < M && (lane == 0 || v != pv);
                const unsigned long long km = __ballot(keep);
                if (keep) out[__popcll(km & ((1ull << lane) - 1ull))] = v;
                if (lane == 0 && ccol < Tm) A.ocnt[mfb + ccol] = __popcll(km);
                continue;
            }
            if (M <= 128) {
                // two hashes per lane: element lane + 64 r in register r
                uint32_t v0 = hlist[a + lane], v1 = lane + 64 < M ? hlist[a + lane + 64] : 0xFFFFFFFFu;
                wave_sort128_u32(v0, v1, lane);
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_update_dpp((int)v0, (int)v0, 0x138, 0xF, 0xF, false);    // wave_shr:1
                uint32_t q1 = (uint32_t)__builtin_amdgcn_update_dpp((int)v1, (int)v1, 0x138, 0xF, 0xF, false);
                const uint32_t last0 = (uint32_t)__builtin_amdgcn_readlane((int)v0, 63);
                if (lane == 0) q1 = last0;
                const bool keep0 = lane == 0 || v0 != q0;                     // (M > 64: register 0 is full)
                const bool keep1 = lane + 64 < M && v1 != q1;
                const unsigned long long km0 = __ballot(keep0), km1 = __ballot(keep1);
                const unsigned long long lt = (1ull << lane) - 1ull;
                const int n0 = __popcll(km0);
                if (keep0) out[__popcll(km0 & lt)] = v0;
                if (keep1) out[n0 + __popcll(km1 & lt)] = v1;
                if (lane == 0 && ccol < Tm) A.ocnt[mfb + ccol] = n0 + __popcll(km1);
                continue;
            }
            // (rare: more than 128 hashes in one column)  copy the run into the scratch list, padded to a multiple of four
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < 4 * M4; i += 64) sl[i] = i < M ? hlist[a + i] : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int ndup = 0;
            if (M > 1) {
                for (int i0 = 0; i0 < M; i0 += 64) {
                    const int i = i0 + lane;
                    const uint32_t v = i < M ? sl[i] : 0xFFFFFFFEu;
                    bool dup = false;
                    const int jend = min(M4, (i0 + 64) >> 2);
                    for (int j4 = 0; j4 < jend; j4++) {
                        const uint4 w = sl4[j4];
                        const int j = 4 * j4;
                        dup = dup || (j < i && (w.x & 0x7FFFFFFFu) == v) || (j + 1 < i && (w.y & 0x7FFFFFFFu) == v)
                                  || (j + 2 < i && (w.z & 0x7FFFFFFFu) == v) || (j + 3 < i && (w.w & 0x7FFFFFFFu) == v);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i < M && dup) sl[i] = v | 0x80000000u;
                    ndup += __popcll(__ballot(i < M && dup));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            for (int i0 = 0; i0 < M; i0 += 64) {
                const int i = i0 + lane;
                const uint32_t v = i < M ? sl[i] : 0xFFFFFFFFu;
                int rank = 0;
                for (int j4 = 0; j4 < M4; j4++) {                        // flagged duplicates / padding are > every hash
                    const uint4 w = sl4[j4];
                    rank += (w.x < v ? 1 : 0) + (w.y < v ? 1 : 0) + (w.z < v ? 1 : 0) + (w.w < v ? 1 : 0);
                }
                if (i < M && !(v & 0x80000000u)) out[rank] = v;
            }
            if (lane == 0 && ccol < Tm) A.ocnt[mfb + ccol] = M - ndup;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        c0 = c1;
    }
}

// K5 (shifts > 1): one thread per (clip, col): S-way merge of sorted per-shift lists, dropping duplicates.
__global__ __launch_bounds__(COL_CHUNK)
void k_merge(MergeArgs A)
{
    const int clip = A.mblk_clip[blockIdx.x];
    const int col = A.mblk_t0[blockIdx.x] + threadIdx.x;
    const int S = A.S;
    const int u0 = clip * S;
    if (col >= A.clip_T0[clip]) return;
    const int64_t mg = A.clip_mfbase[clip] + col;
    uint32_t* out = A.mslots + mg * (int64_t)A.mslot;
    int idx[16];
    int cnt[16];
    const uint32_t* lst[16];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        idx[s] = 0; cnt[s] = 0; lst[s] = nullptr;
        if (s < S && col < A.unit_T[u0 + s]) {
            const int64_t g = A.unit_fbase[u0 + s] + col;
            cnt[s] = A.hcnt[g];
            lst[s] = A.hslots + g * (int64_t)A.slot;
        }
    }
    int n = 0;
    uint32_t last = 0xFFFFFFFFu;
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;
        int bs = -1;
#pragma unroll
        for (int s = 0; s < 16; s++) {
            if (idx[s] < cnt[s]) {
                uint32_t v = lst[s][idx[s]];
                if (bs < 0 || v < best) { best = v; bs = s; }
            }
        }
        if (bs < 0) break;
#pragma unroll
        for (int s = 0; s < 16; s++) if (s == bs) idx[s]++;
        if (n == 0 || best != last) { out[n++] = best; last = best; }
    }
    A.mcnt[mg] = n;
}

// K6: exclusive scan of per-frame counts inside each segment (clip or unit); one workgroup per segment.
__global__ __launch_bounds__(256)
void k_seg_scan(SegScanArgs A)
{
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int seg = blockIdx.x;
    const int64_t base = A.seg_base[seg];
    const int len = A.seg_len[seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    long long total = 0;
    // EPT consecutive counts per thread and pass (a single long file has 12 920 frames in one segment: 7 passes, not 51)
    constexpr int EPT = 8;
    for (int t0 = 0; t0 < len; t0 += 256 * EPT) {
        const int tb = t0 + threadIdx.x * EPT;
        int v[EPT];
        int x = 0;
#pragma unroll
        for (int i = 0; i < EPT; i++) { v[i] = (tb + i < len) ? A.counts[base + tb + i] : 0; x += v[i]; }
        const int mine = x;                          // inclusive scan of the thread sums in the wavefront
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { int y = __shfl_up(x, s); if (lane >= s) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        int run = carry + woff + x - mine;
#pragma unroll
        for (int i = 0; i < EPT; i++) { if (tb + i < len) A.offs[base + tb + i] = run; run += v[i]; }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) A.seg_total[seg] = (len > 0) ? (int64_t)carry_s : 0;
    (void)total;
}

// K7: exclusive scan of int64 segment totals -> CSR offsets [n+1]; single workgroup.
__global__ __launch_bounds__(1024)
void k_excl_scan64(const int64_t* __restrict__ in, int64_t* __restrict__ out, int n)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const long long v = (i < n) ? in[i] : 0;
        long long x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            int lo = __shfl_up((int)(x & 0xFFFFFFFFll), s);
            int hi = __shfl_up((int)(x >> 32), s);
            long long y = ((long long)hi << 32) | (unsigned int)lo;
            if (lane >= s) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const long long carry = carry_s;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

// K8a: scatter (time, hash) rows into the CSR output.  One wavefront takes 64 consecutive columns of
// a clip: their rows are contiguous in the output, so the lanes write consecutive rows (coalesced
// 512-B stores) and find "their" column by a binary search over the 64 exclusive offsets in LDS.
__global__ __launch_bounds__(COL_CHUNK)
void k_scatter_hashes(ScatterHashArgs A)
{
    __shared__ int ex_s[COL_CHUNK / 64][64];
    const int seg = A.blk_seg[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = A.blk_t0[blockIdx.x] + wave * 64;
    const int len = A.seg_len[seg];
    if (col0 >= len) return;                                   // wave-uniform
    const int64_t g0 = A.seg_base[seg] + col0;
    const bool valid = col0 + lane < len;
    const int n = valid ? A.cnt[g0 + lane] : 0;
    int incl = n;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { int y = __shfl_up(incl, s); if (lane >= s) incl += y; }
    const int excl = incl - n;
    const int total = __shfl(incl, 63);
    if (total == 0) return;
    const int64_t base = A.seg_off[seg] + A.offs[g0];
    if (base + total > A.cap) return;                          // output buffer too small: host re-runs the scatter
    int* ex = ex_s[wave];
    ex[lane] = excl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int2* out = reinterpret_cast<int2*>(A.out) + base;
    for (int r0 = 0; r0 < total; r0 += 64) {
        const int r = r0 + lane;
        if (r < total) {
            int j = 0;                                         // largest j with ex[j] <= r
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) if (ex[j + step] <= r) j += step;
            const uint32_t v = A.slots[(g0 + j) * (int64_t)A.slot + (r - ex[j])];
            out[r] = make_int2(col0 + j, (int)v);
        }
    }
}

// K8b: scatter (col, bin) rows of the final peak masks; one thread per (unit, col).
__global__ __launch_bounds__(COL_CHUNK)
void k_scatter_peaks(ScatterPeakArgs A)
{
    const int seg = A.blk_seg[blockIdx.x];
    const int col = A.blk_t0[blockIdx.x] + threadIdx.x;
    if (col >= A.seg_len[seg]) return;
    const int64_t g = A.seg_base[seg] + col;
    const int64_t row = A.seg_off[seg] + A.offs[g];
    if (row + 256 > A.cap) {                         // conservative bound first, exact bound second
        int n = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) n += __popcll(A.masks[g * 4 + q]);
        if (row + n > A.cap) return;
    }
    int2* out = reinterpret_cast<int2*>(A.out) + row;
    int k = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint64_t w = A.masks[g * 4 + q];
        while (w) {
            const int b = 64 * q + __ffsll((long long)w) - 1;
            w &= w - 1;
            out[k++] = make_int2(col, b);
        }
    }
}

// K9: results of a small batch -> pinned host memory (ExportArgs, afp_common.h).  Every thread derives the layout from the
// CSR totals; the rows leave as 8-byte stores over the grid.  Plain stores to fine-grained host memory: they are visible
// to the host once the kernel has completed (the host waits for the stream / event before it reads).
// the four status words of the segment-parallel scan -> the pinned totals; then the block is cleared for the NEXT batch
// (the host skips its memset when the previous batch ended this way: one stream operation less per file)
__device__ __forceinline__ void export_seg_status(const ExportArgs& A)
{
    if (A.seg_status) {
        A.totals[4] = ((int64_t)(uint32_t)A.seg_status[1] << 32) | (uint32_t)A.seg_status[0];
        A.totals[5] = ((int64_t)(uint32_t)A.seg_status[3] << 32) | (uint32_t)A.seg_status[2];
    }
    if (A.seg_zero) for (int i = 0; i < 64 && i < A.zero_words; i++) A.seg_zero[i] = 0;
    if (A.nt_count) A.totals[6] = (int64_t)A.nt_count[0];
}
__global__ __launch_bounds__(256)
void k_export(ExportArgs A)
{
    const int64_t th = A.hashes ? A.clip_hoff[A.nclips] : 0;
    const int64_t tp = A.peaks ? A.unit_poff[A.nunits] : 0;
    int64_t o = AFP_EXPORT_HDR_BYTES;
    const int64_t o_hoff = o;  if (A.hashes) o += 8 * ((int64_t)A.nclips + 1);
    const int64_t o_poff = o;  if (A.peaks) o += 8 * ((int64_t)A.nunits + 1);
    const int64_t o_flags = o; o += 4 * (int64_t)A.nunits;
    o = (o + 15) & ~(int64_t)15;
    const int64_t o_h = o;     o += 8 * th;
    const int64_t o_p = o;     o += 8 * tp;
    const bool ok = th <= A.cap_h && tp <= A.cap_p && o <= A.host_cap;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if (tid == 0) {
        A.totals[0] = th; A.totals[1] = tp;
        export_seg_status(A);
        int64_t* hdr = reinterpret_cast<int64_t*>(A.host);
        hdr[0] = ok ? 1 : 0; hdr[1] = th; hdr[2] = tp; hdr[3] = o;
    }
    // (words 0..63 are the status block thread 0 has just read and cleared; the marks behind it are nobody's input any more)
    for (int64_t i = 64 + tid; i < A.zero_words; i += nth) A.seg_zero[i] = 0;
    if (!ok) return;
    if (A.hashes) {
        int64_t* d = reinterpret_cast<int64_t*>(A.host + o_hoff);
        for (int64_t i = tid; i <= A.nclips; i += nth) d[i] = A.clip_hoff[i];
        const int2* src = reinterpret_cast<const int2*>(A.hashes);
        int2* dst = reinterpret_cast<int2*>(A.host + o_h);
        for (int64_t i = tid; i < th; i += nth) dst[i] = src[i];
    }
    if (A.peaks) {
        int64_t* d = reinterpret_cast<int64_t*>(A.host + o_poff);
        for (int64_t i = tid; i <= A.nunits; i += nth) d[i] = A.unit_poff[i];
        const int2* src = reinterpret_cast<const int2*>(A.peaks);
        int2* dst = reinterpret_cast<int2*>(A.host + o_p);
        for (int64_t i = tid; i < tp; i += nth) dst[i] = src[i];
    }
    int32_t* fl = reinterpret_cast<int32_t*>(A.host + o_flags);
    for (int64_t i = tid; i < A.nunits; i += nth) fl[i] = A.stats[i].flags;
}

// ONE clip, hashes only (Analyzer.wavfile2hashes: audfprint.py:164-165 feeds one file per call): k_seg_scan + k_excl_scan64 +
// k_scatter_hashes + k_export as one launch of one workgroup -- per-column offsets, the CSR pair (0, total), the rows into the
// device buffer (afp_table_store / afp_result_device_ptrs read them there) AND into the pinned host image, header, unit flags,
// segment status.  Three dispatches less at the end of a chain whose kernels run for a few microseconds each.
#define FIN_MAXBLK 4096                           // 64-column blocks the workgroup keeps offsets for: clips of up to 262 144 frames
__global__ __launch_bounds__(1024)
void k_finish_one(ScatterHashArgs A, int32_t* __restrict__ offs_out, int64_t* __restrict__ clip_tot, int64_t* __restrict__ clip_hoff, ExportArgs E)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    __shared__ int blkoff[FIN_MAXBLK];
    __shared__ int ex_s[16][64];
    const int len = A.seg_len[0];
    const int64_t base = A.seg_base[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    constexpr int EPT = 8;
    for (int t0 = 0; t0 < len; t0 += 1024 * EPT) {
        const int tb = t0 + threadIdx.x * EPT;
        int v[EPT];
        int x = 0;
#pragma unroll
        for (int i = 0; i < EPT; i++) { v[i] = (tb + i < len) ? A.cnt[base + tb + i] : 0; x += v[i]; }
        const int mine = x;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { int y = __shfl_up(x, s); if (lane >= s) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        int run = carry + woff + x - mine;
        if ((tb & 63) == 0 && tb < len) blkoff[tb >> 6] = run;
#pragma unroll
        for (int i = 0; i < EPT; i++) { if (tb + i < len) offs_out[base + tb + i] = run; run += v[i]; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    const int64_t th = len > 0 ? (int64_t)carry_s : 0;
    // image layout of k_export for (hashes, no peaks, one clip, nunits units)
    int64_t o = AFP_EXPORT_HDR_BYTES;
    const int64_t o_hoff = o;  o += 8 * 2;
    const int64_t o_flags = o; o += 4 * (int64_t)E.nunits;
    o = (o + 15) & ~(int64_t)15;
    const int64_t o_h = o;     o += 8 * th;
    const bool ok = th <= A.cap && o <= E.host_cap;
    if (threadIdx.x == 0) {
        clip_tot[0] = th; clip_hoff[0] = 0; clip_hoff[1] = th;
        E.totals[0] = th; E.totals[1] = 0;
        export_seg_status(E);
        int64_t* hdr = reinterpret_cast<int64_t*>(E.host);
        hdr[0] = ok ? 1 : 0; hdr[1] = th; hdr[2] = 0; hdr[3] = o;
        if (ok) { int64_t* d = reinterpret_cast<int64_t*>(E.host + o_hoff); d[0] = 0; d[1] = th; }
    }
    for (int i = 64 + (int)threadIdx.x; i < E.zero_words; i += 1024) E.seg_zero[i] = 0;
    if (ok) {
        int32_t* fl = reinterpret_cast<int32_t*>(E.host + o_flags);
        for (int i = threadIdx.x; i < E.nunits; i += 1024) fl[i] = E.stats[i].flags;
    }
    // rows: one wavefront per 64 columns, as k_scatter_hashes (rows of a block are contiguous: coalesced stores)
    int2* outd = reinterpret_cast<int2*>(A.out);
    int2* outh = reinterpret_cast<int2*>(E.host + o_h);
    int* ex = ex_s[wave];
    for (int col0 = wave * 64; col0 < len; col0 += 16 * 64) {
        const int64_t g0 = base + col0;
        const bool valid = col0 + lane < len;
        const int n = valid ? A.cnt[g0 + lane] : 0;
        int incl = n;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { int y = __shfl_up(incl, s); if (lane >= s) incl += y; }
        const int total = __shfl(incl, 63);
        if (total == 0) continue;                                  // wave-uniform
        const int64_t rb = blkoff[col0 >> 6];
        if (rb + total > A.cap) continue;                          // device buffer too small: finalize() re-runs k_scatter_hashes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ex[lane] = incl - n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int r0 = 0; r0 < total; r0 += 64) {
            const int r = r0 + lane;
            if (r < total) {
                int j = 0;                                         // largest j with ex[j] <= r
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) if (ex[j + step] <= r) j += step;
                const uint32_t v = A.slots[(g0 + j) * (int64_t)A.slot + (r - ex[j])];
                const int2 row = make_int2(col0 + j, (int)v);
                outd[rb + r] = row;
                if (ok) outh[rb + r] = row;
            }
        }
    }
}

// raw landmarks -> (col, f1, f2, dt) int32 rows
__global__ __launch_bounds__(COL_CHUNK)
void k_scatter_landmarks(ScatterLmArgs A)
{
    const int seg = A.blk_seg[blockIdx.x];
    const int col = A.blk_t0[blockIdx.x] + threadIdx.x;
    if (col >= A.seg_len[seg]) return;
    const int64_t g = A.seg_base[seg] + col;
    const int n = A.cnt[g];
    if (n == 0) return;
    const uint32_t* in = A.slots + g * (int64_t)A.slot;
    const int64_t row = A.seg_off[seg] + A.offs[g];
    if (row + n > A.cap) return;
    int4* out = reinterpret_cast<int4*>(A.out) + row;
    for (int i = 0; i < n; i++) {
        const uint32_t v = in[i];
        out[i] = make_int4(col, (int)(v & 0xFF), (int)((v >> 8) & 0xFF), (int)(v >> 16));
    }
}

// peak rows (col, bin) of all units -> 256-bit per-frame masks (masks pre-zeroed)
__global__ __launch_bounds__(256)
void k_masks_from_peaks(const int32_t* __restrict__ peaks, const int64_t* __restrict__ upo, int nunits, int64_t np,
                        const int64_t* __restrict__ unit_fbase, uint64_t* __restrict__ masks)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    int lo = 0, hi = nunits;                          // unit u with upo[u] <= i < upo[u+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (upo[mid] <= i) lo = mid; else hi = mid; }
    const int col = peaks[2 * i], bin = peaks[2 * i + 1];
    atomicOr(reinterpret_cast<unsigned long long*>(masks + (unit_fbase[lo] + col) * 4 + (bin >> 6)), 1ull << (bin & 63));
}

// rows per column of given peak lists (rcnt pre-zeroed): the column index of peaks_at[col] (audfprint_analyze.py:323-326)
__global__ __launch_bounds__(256)
void k_rows_count(const int32_t* __restrict__ peaks, const int64_t* __restrict__ upo, int nunits, int64_t np,
                  const int64_t* __restrict__ unit_fbase, int32_t* __restrict__ rcnt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    int lo = 0, hi = nunits;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (upo[mid] <= i) lo = mid; else hi = mid; }
    atomicAdd(&rcnt[unit_fbase[lo] + peaks[2 * i]], 1);
}

// Pairing of peak lists in LIST order (audfprint_analyze.py:328-341 as written: `for peak in peaks_at[col]`, `for peak2 in
// peaks_at[col2]` follow the order the rows were appended in, and a bin may be listed twice).  The 256-bit masks of k_pair
// cannot say that, so this kernel walks the rows: one thread per (unit, col), sources and targets in row order.  Not a hot
// path -- find_peaks and peaks_load emit ascending unique bins and take the mask kernels; this is the rest of the input
// domain of Analyzer.peaks2landmarks.  Hash mode sorts the WHOLE column (sources are not in f1 order here) and leaves
// duplicates to k_merge.
__global__ __launch_bounds__(COL_CHUNK)
void k_pair_rows(PairArgs A, PairRowsArgs R)
{
    const int u = A.cblk_unit[blockIdx.x];
    const int col = A.cblk_t0[blockIdx.x] + threadIdx.x;
    const int T = A.unit_T[u];
    if (col >= T) return;
    const int64_t g = A.unit_fbase[u] + col;
    const int64_t r0 = R.upo[u];
    uint32_t* out = A.hslots + g * (int64_t)A.slot;
    int n_out = 0;
    const int ns = R.rcnt[g];
    const int64_t sa = r0 + R.roffs[g];
    const int dmax = min(T - col, A.targetdt);                                   // :331-332
    for (int i = 0; i < ns; i++) {
        const int f1 = R.rows[2 * (sa + i) + 1];
        int np = 0;
        for (int dt = A.mindt; dt < dmax && np < A.fanout; dt++) {
            const int nt = R.rcnt[g + dt];
            const int64_t ta = r0 + R.roffs[g + dt];
            for (int j = 0; j < nt && np < A.fanout; j++) {
                const int f2 = R.rows[2 * (ta + j) + 1];
                const int d = f2 - f1;
                if ((d < 0 ? -d : d) >= A.targetdf) continue;                    // :335
                if (A.lm_mode) {
                    out[n_out] = (uint32_t)f1 | ((uint32_t)f2 << 8) | ((uint32_t)dt << 16);
                } else {
                    const uint32_t h = ((uint32_t)(f1 & 0xFF) << 12) | ((uint32_t)(d & 0x3F) << 6) | (uint32_t)(dt & 0x3F);   // :92-95
                    int k = n_out;
                    while (k > 0 && out[k - 1] > h) { out[k] = out[k - 1]; k--; }
                    out[k] = h;
                }
                n_out++;
                np++;
            }
        }
    }
    A.hcnt[g] = n_out;
}

// landmarks2hashes (audfprint_analyze.py:92-95)
__global__ __launch_bounds__(256)
void k_lm2hash(const int32_t* __restrict__ lm, int32_t* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 r = reinterpret_cast<const int4*>(lm)[i];
    const int h = ((r.y & 0xFF) << 12) | (((r.z - r.y) & 0x3F) << 6) | (r.w & 0x3F);
    reinterpret_cast<int2*>(out)[i] = make_int2(r.x, h);
}

extern "C" void afp_launch_scatter_landmarks(const ScatterLmArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_scatter_landmarks, dim3(nblk), dim3(COL_CHUNK), 0, st, *a);
}
extern "C" void afp_launch_masks_from_peaks(const int32_t* peaks, const int64_t* upo, int nunits, int64_t np,
                                            const int64_t* unit_fbase, uint64_t* masks, hipStream_t st)
{
    if (np > 0) hipLaunchKernelGGL(k_masks_from_peaks, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, peaks, upo, nunits, np, unit_fbase, masks);
}
extern "C" void afp_launch_lm2hash(const int32_t* lm, int32_t* out, int64_t n, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_lm2hash, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lm, out, n);
}
extern "C" void afp_launch_rows_count(const int32_t* peaks, const int64_t* upo, int nunits, int64_t np,
                                      const int64_t* unit_fbase, int32_t* rcnt, hipStream_t st)
{
    if (np > 0) hipLaunchKernelGGL(k_rows_count, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, peaks, upo, nunits, np, unit_fbase, rcnt);
}
extern "C" void afp_launch_pair_rows(const PairArgs* a, const PairRowsArgs* r, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_pair_rows, dim3(nblk), dim3(COL_CHUNK), 0, st, *a, *r);
}
extern "C" void afp_launch_pair(const PairArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) {
        const size_t nf = COL_CHUNK + a->targetdt;
        const size_t lds = nf * 36 + (a->lds_lists ? (size_t)COL_CHUNK * (a->slot | 1) * 4 : 0);
        hipLaunchKernelGGL(k_pair, dim3(nblk), dim3(COL_CHUNK), lds, st, *a);
    }
}
extern "C" void afp_launch_pairmerge(const PairMergeArgs* a, int nblk, hipStream_t st)
{
    if (nblk <= 0) return;
    const size_t nf = (size_t)a->ch + a->targetdt;
    const size_t lds = (size_t)a->S * nf * 36 + 16 + (size_t)4 * ((((size_t)a->oslot + 3) & ~(size_t)3) + 4) * 4;
    hipLaunchKernelGGL(k_pairmerge, dim3(nblk), dim3(256), lds, st, *a);
}
extern "C" size_t afp_pairlane_lds(int ch, int targetdt, int fanout)
{
    const size_t nf = (size_t)ch + targetdt, wc = (size_t)ch / 4;
    return nf * 32 + 4 * (wc * PL_KMAX + 64 * (size_t)(fanout | 1) + 2 * wc) * 4;
}
extern "C" void afp_launch_pairlane(const PairMergeArgs* a, int nblk, hipStream_t st)
{
    if (nblk <= 0) return;
    hipLaunchKernelGGL(k_pairlane, dim3(nblk), dim3(256), afp_pairlane_lds(a->ch, a->targetdt, a->fanout), st, *a);
}
extern "C" size_t afp_pairlane_ms_lds(int ch, int targetdt, int fanout, int S, int K)
{
    const size_t nf = (size_t)ch + targetdt, wc = (size_t)ch / 4;
    const size_t hcap = (64 * (size_t)(fanout | 1) + 3) & ~(size_t)3, ccap = (wc + 1 + 65 + 3) & ~(size_t)3;
    const size_t pcap = (wc * S * K + 3) & ~(size_t)3, scap = ((size_t)S * K * fanout + 4 + 3) & ~(size_t)3;
    return (size_t)S * nf * 32 + 64 + 4 * (pcap + hcap + scap + ccap) * 4;
}
extern "C" void afp_launch_pairlane_ms(const PairMergeArgs* a, int nblk, hipStream_t st)
{
    if (nblk <= 0) return;
    hipLaunchKernelGGL(k_pairlane_ms, dim3(nblk), dim3(256), afp_pairlane_ms_lds(a->ch, a->targetdt, a->fanout, a->S, a->oslot / (a->S * a->fanout)), st, *a);
}
extern "C" void afp_launch_merge(const MergeArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_merge, dim3(nblk), dim3(COL_CHUNK), 0, st, *a);
}
extern "C" void afp_launch_seg_scan(const SegScanArgs* a, int nseg, hipStream_t st)
{
    if (nseg > 0) hipLaunchKernelGGL(k_seg_scan, dim3(nseg), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_excl_scan64(const int64_t* in, int64_t* out, int n, hipStream_t st)
{
    hipLaunchKernelGGL(k_excl_scan64, dim3(1), dim3(1024), 0, st, in, out, n);
}
extern "C" void afp_launch_scatter_hashes(const ScatterHashArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_scatter_hashes, dim3(nblk), dim3(COL_CHUNK), 0, st, *a);
}
extern "C" void afp_launch_export(const ExportArgs* a, int nblk, hipStream_t st)
{
    hipLaunchKernelGGL(k_export, dim3(nblk), dim3(256), 0, st, *a);
}
extern "C" int afp_finish_one_max_frames(void) { return FIN_MAXBLK * 64; }
extern "C" void afp_launch_finish_one(const ScatterHashArgs* a, int32_t* offs, int64_t* clip_tot, int64_t* clip_hoff, const ExportArgs* e, hipStream_t st)
{
    hipLaunchKernelGGL(k_finish_one, dim3(1), dim3(1024), 0, st, *a, offs, clip_tot, clip_hoff, *e);
}
extern "C" void afp_launch_scatter_peaks(const ScatterPeakArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_scatter_peaks, dim3(nblk), dim3(COL_CHUNK), 0, st, *a);
}
