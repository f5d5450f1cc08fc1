// k_scan.hip -- K2 (per-unit statistics, floor correction) and K3 (the sequential
// decaying-threshold peak picker) for gfx950.  COMPILED WITH -ffp-contract=off: the HPF and
// threshold recurrences must round exactly like the reference's separate numpy operations.
//
// K3 replaces, per unit: the floor/mean of find_peaks (audfprint_analyze.py:285-286), the
// lfilter HPF (:293-295), _decaying_threshold_fwd_prune (:199-231) and
// _decaying_threshold_bwd_prune_peaks (:233-253).  Frame t depends on frame t-1 (the
// threshold vector), so time is sequential; parallelism is across units, with the 256 bins of
// a unit spread 4-per-lane over one wavefront: lane L owns bins 4L..4L+3, threshold and HPF state
// live in VGPRs for the whole clip, local maxima need one DPP neighbour move per side, candidates
// are found by ballots and the Gaussian bumps come from an LDS copy of the host-computed table
// (bits equal to the reference's __sp_vals, :191-192).  See the K3 block comment below for the
// producer / scanner wavefront pair.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "afp_common.h"

// SCAN_SMALL_LDS=1 (second compilation of this file, build.py): k_scan with 8 KB of LDS per workgroup
// instead of 15.5 KB -- see the note at CF below.
#ifndef SCAN_SMALL_LDS
#define SCAN_SMALL_LDS 0
#endif


__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
    int lo = __shfl_xor(__double2loint(v), mask);
    int hi = __shfl_xor(__double2hiint(v), mask);
    return __hiloint2double(hi, lo);
}
// wavefront-uniform read of lane `src` (src must be uniform)
__device__ __forceinline__ double readlane_d(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// DPP cross-lane moves (VALU latency, no LDS round trip).  Lanes without a source keep `old`.
template <int CTRL, int ROWMASK = 0xF, int BANKMASK = 0xF>
__device__ __forceinline__ double dpp_d(double old, double v)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, BANKMASK, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, BANKMASK, false);
    return __hiloint2double(hi, lo);
}
// the same with zeros in the lanes that have no source (bound_ctrl): the destination needs no initialisation, which saves
// one v_mov per dword -- for callers that mask those lanes anyway
template <int CTRL>
__device__ __forceinline__ double dpp_d_zero(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_WAVE_SHR1 0x138
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// wavefront max of a double; the result is returned wave-uniform
__device__ __forceinline__ double wave_max_uniform(double v)
{
    v = fmax(v, dpp_d<DPP_ROW_SHR(1)>(v, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR(2)>(v, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR(4)>(v, v));
    v = fmax(v, dpp_d<DPP_ROW_SHR(8)>(v, v));
    v = fmax(v, dpp_d<DPP_ROW_BCAST15, 0xA>(v, v));
    v = fmax(v, dpp_d<DPP_ROW_BCAST31, 0xC>(v, v));
    return readlane_d(v, 63);
}

// ------------------------------------------------------------------------------------------
// K2a: one wavefront per unit reduces the STFT partials in a fixed order.
#if !SCAN_SMALL_LDS          // (this file is compiled twice, see build.py; the second object carries only the scan)
// (one wavefront; `publish`: this wavefront writes the unit's record and lists its chunks -- k_stats_corr runs the reduction
//  in every workgroup of a unit, identically, and lets the first one publish)
__device__ __forceinline__ UnitStats unit_stats_wave(const StatsArgs& A, int u, int lane, bool publish)
{
    const int T = A.unit_T[u];
    UnitStats st;
    st.logfloor = 0.0; st.lsum = 0.0; st.pmax = 0.0; st.flags = 0; st.pad = 0; st.tie_first = 0; st.tie_last = -1;
    if (T <= 0) {
        st.flags = UNIT_EMPTY;
        if (lane == 0 && publish) A.stats[u] = st;
        return st;
    }
    const int64_t b0 = A.unit_bbase[u], b1 = A.unit_bbase[u + 1];
    double pmax = 0.0, lmin = INFINITY, lsum = 0.0;
    double flat = 0.0;
    for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) {
        pmax = fmax(pmax, A.blk_pmax[b]);
        lmin = fmin(lmin, A.blk_lmin[b]);
        lsum += A.blk_lsum[b];
        flat = fmax(flat, A.blk_flat[b]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        pmax = fmax(pmax, shfl_xor_d(pmax, s));
        lmin = fmin(lmin, shfl_xor_d(lmin, s));
        lsum += shfl_xor_d(lsum, s);
        flat = fmax(flat, shfl_xor_d(flat, s));
    }
    st.pmax = pmax;
    st.lsum = lsum;
    // a NaN / Inf sample makes every bin of its frames NaN: the sum of logs is not finite (this file is compiled without
    // NaN semantics, so the test looks at the exponent bits); an overflow of |S|^2 shows in pmax
    const bool nonfinite = ((__double2hiint(lsum) >> 20) & 0x7ff) == 0x7ff || ((__double2hiint(pmax) >> 20) & 0x7ff) == 0x7ff;
    if (nonfinite) {
        // the reference: smax = np.max(S) is NaN, `smax > 0` is false -> "identically zero" warning, no peaks (:283-290)
        st.flags = UNIT_ZERO | UNIT_NONFINITE;
        st.pmax = 0.0; st.lsum = 0.0;
    } else if (!(pmax > 0.0)) {
        st.flags = UNIT_ZERO;                         // identically-zero input (audfprint_analyze.py:287-290)
    } else {
        st.logfloor = log(sqrt(pmax) / 1e6);          // log(max|S| / 1e6), :285
        if (lmin < st.logfloor) st.flags |= UNIT_CORR;
        if (flat * flat * 1e12 > pmax) {              // a single-parity frame whose level bound lies ABOVE the floor max|S| / 1e6 (:285)
            st.flags |= UNIT_TIE;
            // which frames: the chunks whose flat level passes the same test (their first / last single-parity frame)
            int f0 = 0x7fffffff, f1 = -1;
            for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) {
                const double fl = A.blk_flat[b];
                if (fl * fl * 1e12 > pmax) { f0 = min(f0, (int)A.blk_flat[b + A.part_stride]); f1 = max(f1, (int)A.blk_flat[b + 2 * A.part_stride]); }
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) { f0 = min(f0, __shfl_xor(f0, s)); f1 = max(f1, __shfl_xor(f1, s)); }
            st.tie_first = f0; st.tie_last = f1;
        }
    }
    if (lane == 0 && publish) A.stats[u] = st;
    if (publish && (st.flags & UNIT_CORR) && A.corr_cnt) {
        // compact pipeline: this unit goes through the dense kernels -- list its STFT chunks (order is irrelevant)
        const int nch = (T + STFT_FPB - 1) / STFT_FPB;
        int base = 0;
        if (lane == 0) base = atomicAdd(A.corr_cnt, nch);
        base = __shfl(base, 0);
        for (int i = lane; i < nch; i += AFP_WAVE) { ChunkDesc c; c.unit = u; c.t0 = i * STFT_FPB; A.corr_list[base + i] = c; }
    }
    return st;
}
__global__ __launch_bounds__(AFP_WAVE)
void k_unit_stats(StatsArgs A)
{
    (void)unit_stats_wave(A, blockIdx.x, threadIdx.x, true);
}

// ------------------------------------------------------------------------------------------
// K2b: where some log|S| fell under the floor, sum (floor - value) so that
//      mean(max(log|S|, floor)) = (lsum + corr) / (257 T).   One workgroup per STFT chunk;
//      chunks that never went under the floor leave at once.
__device__ __forceinline__ void floor_corr_unit(const CorrArgs& A, int u, const UnitStats& st, double (&red)[4])
{
    if (!(st.flags & UNIT_CORR)) return;           // (blk_corr is only read for flagged units)
    const int T = A.unit_T[u];
    const double lf = st.logfloor;
    // (the chunks of a unit are spread over gridDim.y workgroups: a single long file would otherwise walk them one by one)
    for (int64_t blk = A.unit_bbase[u] + blockIdx.y; blk < A.unit_bbase[u + 1]; blk += gridDim.y) {
        if (!(A.blk_lmin[blk] < lf)) {
            if (threadIdx.x == 0) A.blk_corr[blk] = 0.0;
            continue;
        }
        const int t0 = A.blk_t0[blk];
        const int nt = min(STFT_FPB, T - t0);
        const int64_t fb = A.unit_fbase[u] + t0;
        double acc = 0.0;
        // thread = bin: the rows of the chunk are independent loads (8 in flight)
#pragma unroll 8
        for (int t = 0; t < nt; t++) {
            const double v = A.logS[(fb + t) * AFP_NBINS + threadIdx.x];      // finite: k_stft's log of 0 is -354.9
            acc += (v < lf) ? (lf - v) : 0.0;
        }
        if ((int)threadIdx.x < nt) {
            const double v = A.nyq[fb + threadIdx.x];
            acc += (v < lf) ? (lf - v) : 0.0;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc += shfl_xor_d(acc, s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) A.blk_corr[blk] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}
__global__ __launch_bounds__(256)
void k_floor_corr(CorrArgs A)
{
    __shared__ double red[4];
    const int u = blockIdx.x;                      // one workgroup per unit: almost always nothing to do
    const UnitStats st = A.stats[u];
    floor_corr_unit(A, u, st, red);
}
// K2a + K2b in one launch (the dense and the segment path; the compact path needs the statistics BEFORE its dense
// re-transform and keeps the two kernels): every workgroup of a unit reduces the unit's partials itself -- a few hundred
// values, the same instructions in the same order, so all of them hold the same record -- and goes on to its share of the
// floor correction; workgroup (u, 0) publishes the record.  One dispatch less in the one-file chain.
__global__ __launch_bounds__(256)
void k_stats_corr(StatsArgs S, CorrArgs A)
{
    __shared__ double red[4];
    __shared__ UnitStats st_s;
    const int u = blockIdx.x;
    if (threadIdx.x < AFP_WAVE) {
        const UnitStats st = unit_stats_wave(S, u, threadIdx.x, blockIdx.y == 0);
        if (threadIdx.x == 0) st_s = st;
    }
    __syncthreads();
    const UnitStats st = st_s;
    floor_corr_unit(A, u, st, red);
}

// ------------------------------------------------------------------------------------------
#endif  // !SCAN_SMALL_LDS

// K3 helpers.  thr[j] / y[j] belong to bin 4*lane + j.

// sthresh = max(sthresh, val * G[. - bin])  (audfprint_analyze.py:194-196, 226-228).
// The Gaussian lives in LDS DE-INTERLEAVED so that every read is a wavefront of consecutive doubles (no bank
// conflicts; the natural layout G[255 + 4 lane + j - bin] strides the lanes by 32 bytes: 2- to 4-way conflicts on
// every bump):  Gd[e][64 + q] = G[|4 q + e|],  e = 0..3, q = -63..63.   With bin = 4 B + S, lane L needs for its
// bin 4 L + j the distance 4 (L - B) + (j - S): table |j - S| at index 64 + (L - B) when j >= S, and -- by symmetry
// |4 q + e| = |4 (-q) + (-e)| -- at index 64 - (L - B) when j < S.  S is a compile-time constant wherever the
// candidate came out of the ballot of register S, so the four reads have immediate offsets.
#define GD_ROW 128
template <int S>
__device__ __forceinline__ void bump_s(double (&thr)[4], double val, int B, int lane, const double* Gd)
{
    const double* up = Gd + (64 + lane - B);
    const double* dn = Gd + (64 - lane + B);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double g = (j >= S) ? up[(j - S) * GD_ROW] : dn[(S - j) * GD_ROW];
        thr[j] = fmax(thr[j], val * g);
    }
}
// the same for a register index known only at run time (wave-uniform): the branches select nothing but the four LDS
// reads -- the threshold update itself is common code, so no threshold register is copied around the join
__device__ __forceinline__ void bump(double (&thr)[4], double val, int bin, int lane, const double* Gd)
{
    const int B = bin >> 2, S = bin & 3;
    const double* up = Gd + (64 + lane - B);
    const double* dn = Gd + (64 - lane + B);
    double g0, g1, g2, g3;
    if (S < 2) {
        if (S == 0) { g0 = up[0]; g1 = up[GD_ROW]; g2 = up[2 * GD_ROW]; g3 = up[3 * GD_ROW]; }
        else { g0 = dn[GD_ROW]; g1 = up[0]; g2 = up[GD_ROW]; g3 = up[2 * GD_ROW]; }
    } else {
        if (S == 2) { g0 = dn[2 * GD_ROW]; g1 = dn[GD_ROW]; g2 = up[0]; g3 = up[GD_ROW]; }
        else { g0 = dn[3 * GD_ROW]; g1 = dn[2 * GD_ROW]; g2 = dn[GD_ROW]; g3 = up[0]; }
    }
    thr[0] = fmax(thr[0], val * g0);
    thr[1] = fmax(thr[1], val * g1);
    thr[2] = fmax(thr[2], val * g2);
    thr[3] = fmax(thr[3], val * g3);
}
// The BACKWARD pass meets its peaks with a register index known only at run time, where the de-interleaved table costs a
// branch tree or a dozen address instructions per bump; it uses the plain LINEAR layout Gl[255 + d] = G[|d|] instead
// (one address, four consecutive doubles; 2- to 4-way bank conflicts, which cost less than the address arithmetic).
// The same 4 KB of LDS hold first the de-interleaved, then -- re-filled between the passes -- the linear table.
__device__ __forceinline__ void bump_lin(double (&thr)[4], double val, int bin, int lane, const double* Gl)
{
    const double* g = Gl + (255 + 4 * lane - bin);
#pragma unroll
    for (int j = 0; j < 4; j++) thr[j] = fmax(thr[j], val * g[j]);
}
__device__ __forceinline__ void fill_gauss_linear(double* Gl, const double* __restrict__ gauss, int tid, int nthreads)
{
    for (int i = tid; i < 512; i += nthreads) { const int dd = i - 255; Gl[i] = (i < 511) ? gauss[dd < 0 ? -dd : dd] : 0.0; }
}
__device__ __forceinline__ void fill_gauss(double* Gd, const double* __restrict__ gauss, int tid, int nthreads)
{
    for (int i = tid; i < 4 * GD_ROW; i += nthreads) {
        const int e = i / GD_ROW, q = (i % GD_ROW) - 64;
        int d = 4 * q + e;
        d = d < 0 ? -d : d;
        Gd[i] = d < AFP_NBINS ? gauss[d] : 0.0;
    }
}

// locmax (audfprint_analyze.py:36-52): >= on the left, strict on the right, ends allowed.
__device__ __forceinline__ void locmax4(const double (&y)[4], int lane, bool (&lm)[4])
{
    double left = dpp_d_zero<DPP_WAVE_SHR1>(y[3]);      // bin 4L-1 (lane 0 gets 0; masked below)
    double right = dpp_d_zero<DPP_WAVE_SHL1>(y[0]);     // bin 4L+4 (lane 63 likewise)
    // five compares, not eight: "y[j+1] >= y[j]" is the complement of "y[j+1] < y[j]" (the values are finite: floored logs
    // through a stable filter; NaN input is garbage in the reference too)
    const bool d01 = y[1] < y[0], d12 = y[2] < y[1], d23 = y[3] < y[2];
    lm[0] = (lane == 0 || y[0] >= left) && d01;
    lm[1] = !d01 && d12;
    lm[2] = !d12 && d23;
    lm[3] = !d23 && (lane == 63 || right < y[3]);
}

// spreadpeaksinvector (audfprint_analyze.py:153-160): start from zeros, spread every local max.
__device__ __forceinline__ void spread_all(double (&thr)[4], const double (&v)[4], int lane, const double* Gs)
{
    bool lm[4];
    locmax4(v, lane, lm);
#pragma unroll
    for (int j = 0; j < 4; j++) thr[j] = 0.0;
#define AFP_SPREAD(J)                                                              \
    for (unsigned long long m = __ballot(lm[J]); m != 0ull; m &= m - 1) {         \
        const int wl = __ffsll((long long)m) - 1;                                 \
        bump_s<J>(thr, readlane_d(v[J], wl), wl, lane, Gs);                       \
    }
    AFP_SPREAD(0)
    AFP_SPREAD(1)
    AFP_SPREAD(2)
    AFP_SPREAD(3)
#undef AFP_SPREAD
}

struct __attribute__((aligned(16))) dpair { double a, b; };

// What the producer hands the scanner for a bin that is NOT a local maximum: anything that compares below every
// threshold (thresholds are >= 0).  Only the HIGH word is replaced, by 0xBF800000 -- as the top half of a double that is
// a small negative number (-2^-7 .. -2^-6), and as an operand it is the hardware's inline constant -1.0f, so the select
// needs neither a second instruction for the low word nor a register for the constant.
__device__ __forceinline__ double keep_if_max(double y, bool lm)
{
    return __hiloint2double(lm ? __double2hiint(y) : (int)0xBF800000, __double2loint(y));
}

// floor + mean (audfprint_analyze.py:285-286) then one step of lfilter([1,-1],[1,-pole]) in
// direct form II transposed (:293-294):  y = x + z ;  z = -x + pole*y
// RAW: the rows ARE the onset-filtered spectrogram already (afp_prune_spectrogram: the caller's sgram goes straight
// into _decaying_threshold_fwd_prune / _bwd_prune_peaks, audfprint_analyze.py:199-253)
template <bool RAW = false>
__device__ __forceinline__ void hpf_step(const double (&raw)[4], double lf, double mean, double pole,
                                         double (&z)[4], double (&y)[4])
{
    if (RAW) {
#pragma unroll
        for (int j = 0; j < 4; j++) y[j] = raw[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double x = fmax(raw[j], lf) - mean;
        double yy = x + z[j];
        z[j] = (-x) + pole * yy;
        y[j] = yy;
    }
}

// ---- K3 ------------------------------------------------------------------------------------
// Workgroup = 2 wavefronts per unit, a software pipeline across two SIMDs:
//   wave 1, the PRODUCER: streams log|S| from HBM (PFC chunks of CF frames in flight in VGPRs,
//     all loads unconditional so vmcnt is counted exactly), applies floor/mean + the HPF
//     recurrence, finds the local maxima (neighbours via DPP) and writes to an LDS ring the
//     column with every NON-local-max bin replaced by -1 (thresholds are >= 0, so "candidate"
//     becomes a single compare y > thr).  In the backward pass it streams the forward survivors of
//     a chunk of frames, ranks the records of each frame by (value, bin) descending -- the order the
//     backward pass must see (:241) -- with wavefront permutes, and hands them over through LDS.
//   wave 0, the SCANNER: runs only the sequential threshold recurrence out of LDS; per frame
//     without candidates that is 4 compares + 4 ballots + the decay multiply.  Survivors are
//     collected lane by lane with v_writelane and stored in ballot order (no ranking on the critical
//     chain).  It issues only global STORES, so it never waits on vmcnt.
// One s_barrier per chunk joins the two.
// SCAN_SMALL_LDS: 8 KB of LDS per workgroup instead of 12 (1-frame ring slots; the backward record ring lives in
// ring space that is idle by then), and the kernel stays within 64 VGPRs, so that four scan workgroups leave a CU
// room for THREE k_stft workgroups (3 x 128 + 2 x 64 registers per SIMD lane, 3 x 31 KB + 4 x 8 KB of LDS).
#if SCAN_SMALL_LDS
#define k_scan k_scan_small                    // second compilation of this file: distinct kernel symbols
#define CF 1                                   // frames per forward chunk
#else
#define CF 2                                   // frames per forward chunk
#endif
#define PFB 4                                  // backward record chunks in flight
#define FROW 256                               // doubles per frame row in the ring
#define SORT_IN_BWD_MAXK 16                    // up to this many peaks per frame the backward producer ranks the records

__device__ __forceinline__ double readfirstlane_d(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// record (val, bin) -> lane `idx` of (ev, eb); val / bin / idx are wave-uniform (scalar registers).  v_writelane_b32 may read
// only one scalar register besides M0 (constant-bus rule), so the lane select travels in M0 (no builtin in this compiler).
// (the value is kept as two separate dwords all the way to the store: a 64-bit register pair would be shuffled around the asm)
__device__ __forceinline__ void put_record(int& ev_lo, int& ev_hi, int& eb, double val, int bin, int idx)
{
    asm("s_mov_b32 m0, %5\n\ts_nop 0\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %6, m0"
        : "+v"(ev_lo), "+v"(ev_hi), "+v"(eb)
        : "s"(__double2loint(val)), "s"(__double2hiint(val)), "s"(idx), "s"(bin)
        : "m0");
}
// clear bit `b` (wave-uniform) of a wave-uniform 64-bit mask: one scalar instruction instead of the add / addc / and of m &= m - 1
__device__ __forceinline__ unsigned long long clear_bit(unsigned long long m, int b)
{
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
}
__device__ __forceinline__ double bpermute_d(int src_lane, double v)
{
    int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// per-unit mean of the floored log-spectrogram (audfprint_analyze.py:286); both waves compute it
__device__ __forceinline__ double unit_mean(const ScanArgs& A, const UnitStats& st, int u, int T, int lane)
{
    double corr = 0.0;
    if (st.flags & UNIT_CORR) {
        const int64_t b0 = A.unit_bbase[u], b1 = A.unit_bbase[u + 1];
        for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) corr += A.blk_corr[b];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) corr += shfl_xor_d(corr, s);
    }
    return (st.lsum + corr) / (257.0 * (double)T);
}

__device__ __forceinline__ void prod_load_chunk(const double* __restrict__ L, int64_t fb, int T, int chunk, int lane,
                                                dpair (&q)[CF][2])
{
#pragma unroll
    for (int i = 0; i < CF; i++) {
        int t = chunk * CF + i;
        if (t > T - 1) t = T - 1;                                  // clamped: always issued
#if defined(SCAN_NT) && SCAN_NT
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v* p = reinterpret_cast<const d2v*>(L + (fb + t) * AFP_NBINS + 4 * lane);
        const d2v v0 = __builtin_nontemporal_load(p), v1 = __builtin_nontemporal_load(p + 1);     // read once: streaming
        q[i][0].a = v0.x; q[i][0].b = v0.y; q[i][1].a = v1.x; q[i][1].b = v1.y;
#else
        const dpair* p = reinterpret_cast<const dpair*>(L + (fb + t) * AFP_NBINS + 4 * lane);
        q[i][0] = p[0]; q[i][1] = p[1];
#endif
    }
}

// HPF + local-max masking of one chunk, written to ring slot `dst` (lf / mean / pole are wave-uniform: scalar registers)
template <bool RAW>
__device__ __forceinline__ void prod_proc_chunk(const dpair (&q)[CF][2], int chunk, int T, int lane, double lf, double mean,
                                                double pole, double (&z)[4], double* dst, double* ylast_g, double* sgram_dbg,
                                                int64_t fb)
{
#pragma unroll
    for (int i = 0; i < CF; i++) {
        const int t = chunk * CF + i;
        const double raw[4] = {q[i][0].a, q[i][0].b, q[i][1].a, q[i][1].b};
        double y[4];
        hpf_step<RAW>(raw, lf, mean, pole, z, y);
        bool lm[4];
        locmax4(y, lane, lm);
        dpair o0, o1;
        o0.a = keep_if_max(y[0], lm[0]); o0.b = keep_if_max(y[1], lm[1]);
        o1.a = keep_if_max(y[2], lm[2]); o1.b = keep_if_max(y[3], lm[3]);
        // LDS row layout: lane L's bins (4L, 4L+1) at doubles [2L, 2L+1], bins (4L+2, 4L+3) at [128+2L, ...]:
        // both halves are lane-contiguous 16-byte accesses (conflict-free ds_*_b128)
        dpair* o = reinterpret_cast<dpair*>(dst + i * FROW + 2 * lane);
        if (t < T) { o[0] = o0; o[64] = o1; }                      // (padded chunks write nothing)
        if (t == T - 1) {
            // the raw last column seeds the backward pass (:237): parked in HBM once per unit (a register copy of it
            // would cost eight VGPRs for the whole pass)
            asm volatile("" ::: "memory");                         // (keeps this a branch, not eight selects per frame)
            dpair a, b;
            a.a = y[0]; a.b = y[1]; b.a = y[2]; b.b = y[3];
            dpair* yl = reinterpret_cast<dpair*>(ylast_g + 4 * lane);
            yl[0] = a; yl[1] = b;
        }
        if (sgram_dbg && t < T) {
            double* g = sgram_dbg + (fb + t) * AFP_NBINS + 4 * lane;
#pragma unroll
            for (int j = 0; j < 4; j++) g[j] = y[j];
        }
    }
}

// The same for a chunk that lies wholly inside the clip and does not hold its last frame: no bounds checks, no debug tap
// (the forward loops run a guard-free MAIN part and a guarded TAIL of at most a few chunks)
template <bool RAW>
__device__ __forceinline__ void prod_proc_chunk_main(const dpair (&q)[CF][2], int lane, double lf, double mean, double pole,
                                                     double (&z)[4], double* dst)
{
#pragma unroll
    for (int i = 0; i < CF; i++) {
        const double raw[4] = {q[i][0].a, q[i][0].b, q[i][1].a, q[i][1].b};
        double y[4];
        hpf_step<RAW>(raw, lf, mean, pole, z, y);
        bool lm[4];
        locmax4(y, lane, lm);
        dpair o0, o1;
        o0.a = keep_if_max(y[0], lm[0]); o0.b = keep_if_max(y[1], lm[1]);
        o1.a = keep_if_max(y[2], lm[2]); o1.b = keep_if_max(y[3], lm[3]);
        dpair* o = reinterpret_cast<dpair*>(dst + i * FROW + 2 * lane);
        o[0] = o0; o[64] = o1;
    }
}
// rows of a chunk through a RUNNING 32-bit per-lane byte offset from the unit's (scalar) base address: one vector add
// per chunk instead of a 64-bit scalar address computation per row; the caller guarantees the rows exist
__device__ __forceinline__ void prod_load_chunk_ptr(const char* ubase, unsigned& voff, dpair (&q)[CF][2])
{
#pragma unroll
    for (int i = 0; i < CF; i++) {
        const dpair* rp = reinterpret_cast<const dpair*>(ubase + voff + (unsigned)(i * AFP_NBINS * 8));
        q[i][0] = rp[0];
        q[i][1] = rp[1];
    }
    voff += (unsigned)(CF * AFP_NBINS * 8);
}

__device__ __forceinline__ void read_frame(const double* src, int lane, double (&x)[4])
{
    const dpair* p = reinterpret_cast<const dpair*>(src + 2 * lane);      // row layout: see prod_proc_chunk
    dpair q0 = p[0], q1 = p[64];
    x[0] = q0.a; x[1] = q0.b; x[2] = q1.a; x[3] = q1.b;
}

// PFC = forward chunks the producer keeps in flight in VGPRs
#if SCAN_SMALL_LDS
#define SCAN_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))        // <= 64 VGPRs: two of these + three k_stft wavefronts per SIMD
#else
#define SCAN_OCC
#endif
// CMP (k_scan_small only, CF == 1): the COMPACT spectral stage feeds this kernel (k_stft<ST, true>, k_stft.hip): per frame the
// 256-bit local-maximum mask and the onset-filtered values of the maxima, filtered WITHOUT the per-unit mean (:286).  The
// filter is linear, so the true value is  v - mean * pole^t ; the producer subtracts that term (c_t, formed by the same
// rounded multiplications wherever it is needed: the scanner's initial threshold and the seed of the backward pass must
// meet the forward candidates with bit-identical values, see :217 / :242) and rebuilds the dense column the scanner reads,
// non-maxima marked as in the dense path.  Units that needed the floor (UNIT_CORR) are left to the dense kernels.
// SEG (dense rows that are already onset-filtered, RAW): the workgroup scans one SEGMENT of a unit -- see SegDesc in
// afp_common.h -- as a "virtual unit" [tb, te) of the unit's rows: forward phase [s - W, e) recording from s, backward phase
// [s, e + 1 + W) downwards recording the masks of frames s + 1 .. e; threshold vectors at the segment boundaries are left in
// seg_state; the repair launch re-runs the segments whose entry state is not the bit pattern the neighbour ended with.
__device__ __forceinline__ void seg_store_state(double* dst, int lane, const double (&thr)[4])
{
    dpair a, b;
    a.a = thr[0]; a.b = thr[1]; b.a = thr[2]; b.b = thr[3];
    dpair* o = reinterpret_cast<dpair*>(dst + 4 * lane);
    o[0] = a; o[1] = b;
}
__device__ __forceinline__ void seg_load_state(const double* src, int lane, double (&thr)[4])
{
    const dpair* o = reinterpret_cast<const dpair*>(src + 4 * lane);
    const dpair a = o[0], b = o[1];
    thr[0] = a.a; thr[1] = a.b; thr[2] = b.a; thr[3] = b.b;
}
// SEG: what one call scans -- the first launch of a phase: segment `seg` from its warm-up (init_state null), leaving its
// entry / exit states in entry_out / exit_out; the chain launch: segment `seg` from init_state (the neighbour's final end
// state, copied to entry_out), no warm-up, every record / mask row of its frames rewritten.
struct SegRun {
    int seg;
    const double* init_state;
    double* entry_out;
    double* exit_out;
};
template <bool PROF, int PFC, bool RAW, bool CMP, bool SEG = false, bool GUARD = false>
__device__ __forceinline__ void scan_unit(const ScanArgs& A, double* Gs, double (*ring)[CF * FROW], double& cshare,
                                          double (*cvring_s)[AFP_WAVE], int (*cbring_s)[AFP_WAVE], const SegRun& sr = SegRun())
{
    static_assert(!CMP || (CF == 1 && PFC == 4 && !RAW), "compact rows: one frame per chunk, four frames in flight");
    static_assert(!SEG || (!RAW && !CMP && !PROF), "segments filter their own rows from the state k_hpf left");
    const int lane = threadIdx.x & 63;
    const bool scanner = threadIdx.x < AFP_WAVE;
    int u_ = blockIdx.x, T_ = 0, tb_ = 0;
    // segment mode: which passes this launch runs, the first recorded frame of the forward pass (relative to tb), the
    // relative index of frame e in the backward pass, start from a given state / write empty records too (repair)
    bool run_fwd = true, run_bwd = true, from_state = false, clear = false, seg_bottom = true;
    int rec0 = 0, rtop = 0x7fffffff;
    const double* init_state = nullptr;
    double *dump_entry = nullptr, *dump_exit = nullptr;
    const double *seg_z0 = nullptr, *seg_yl = nullptr;      // k_hpf records: filter state at entry of tb, filtered last column
    if constexpr (SEG) {
        const SegDesc sd = A.segs[sr.seg];
        u_ = sd.unit;
        const int Tu = A.unit_T[u_];
        const bool fwdp = A.seg_phase == SEG_FWD;
        run_fwd = fwdp; run_bwd = !fwdp;
        const int nb = fwdp ? sd.prev : sd.next;                        // the neighbour whose end state this segment continues
        int tb = sd.s, te = sd.e;
        seg_bottom = sd.prev < 0;
        dump_exit = sr.exit_out;
        if (!sr.init_state) {
            if (fwdp) { if (nb >= 0) { tb = sd.s - A.seg_W; if (tb < 0) tb = 0; } }
            else if (nb >= 0) { te = sd.e + 1 + A.seg_W; if (te > Tu) te = Tu; }
            if (nb >= 0) dump_entry = sr.entry_out;
            if (fwdp) { if (sd.dz_fwd >= 0) seg_z0 = A.hpf_dump + (int64_t)sd.dz_fwd * 2 * AFP_NBINS; }
            else seg_yl = A.hpf_dump + ((int64_t)sd.dy_bwd * 2 + 1) * AFP_NBINS;
        } else {
            init_state = sr.init_state;                                 // the neighbour's end state = the true state here
            from_state = true; clear = true;
            if (!fwdp) te = sd.e + 1;                                   // frame e is scanned again, from the state at its entry
            if (fwdp) seg_z0 = A.hpf_dump + (int64_t)sd.dz_rep * 2 * AFP_NBINS;
            else seg_yl = A.hpf_dump + ((int64_t)sd.dy_rep * 2 + 1) * AFP_NBINS;
            if (scanner) { double v[4]; seg_load_state(init_state, lane, v); seg_store_state(sr.entry_out, lane, v); }
        }
        rec0 = sd.s - tb;
        rtop = sd.e - tb;
        tb_ = tb; T_ = te - tb;
    } else {
        T_ = A.unit_T[u_];
        if (!CMP && A.only_if && A.only_if[3] == 0 && A.only_if_unit[u_] == 0) return;      // dense fallback behind the segment kernels: not needed
        clear = !CMP && A.clear_all != 0;
    }
    const int u = u_;
    const int T = T_;
    if (T <= 0) return;
#ifdef SCAN_PRIO
    __builtin_amdgcn_s_setprio(SCAN_PRIO);
#endif
#if SCAN_SMALL_LDS
    // the forward ring is idle once the forward pass is over: it carries the backward record ring
    double (*cvring)[AFP_WAVE] = reinterpret_cast<double (*)[AFP_WAVE]>(ring[0]);
    int (*cbring)[AFP_WAVE] = reinterpret_cast<int (*)[AFP_WAVE]>(ring[0] + 2 * AFP_WAVE);
#else
    double (*cvring)[AFP_WAVE] = cvring_s;
    int (*cbring)[AFP_WAVE] = cbring_s;
#endif
    const int64_t fb = A.unit_fbase[u] + tb_;
    const int K = A.K;
    const UnitStats st = A.stats[u];
    // Near-tie guard (ScanArgs::nt_eps > 0): the values compared below carry absolute errors of ~1e-13 against the reference's
    // (log of a 512-point FFT: DESIGN.md), so a comparison decided by less than nt_eps is one the reference's own arithmetic
    // could decide the other way.  The scanner ORs the ballots of |a - b| <= nt_eps into a scalar and marks the unit at the
    // end; the comparisons themselves are untouched.  (The ORDER of near-equal records inside one frame needs no guard: a
    // kept record raises the threshold at another bin by val * G(d) < val, which cannot reject a record of nearly the same
    // value -- only the old threshold can, and that comparison is guarded.)
    // GUARD is a template parameter: the kernels a handle runs by default (guard off) carry none of this -- same registers,
    // same instructions as before the guard existed.
    const double nte = GUARD ? A.nt_eps : 0.0;
    const bool guard = GUARD && nte > 0.0;
    unsigned long long nt = 0ull;
    auto flush_nt = [&]() {
        if (guard && nt != 0ull && lane == 0) {
            const int old = atomicOr(&A.stats_rw[u].flags, UNIT_NEARTIE);
            if (!(old & UNIT_NEARTIE) && A.nt_count) atomicAdd(A.nt_count, 1);
        }
    };

    if (st.flags & UNIT_ZERO) {
        // all-zero spectrogram: HPF of zeros is zero, nothing exceeds the (zero) threshold
        // (masks are pre-zeroed by k_stft before this launch)
        if (threadIdx.x == 0) A.unit_mean[u] = 0.0;
        return;
    }

    fill_gauss(Gs, A.gauss, threadIdx.x, 2 * AFP_WAVE);

    const double* __restrict__ L = A.logS;
    // wave-uniform constants live in scalar registers
    const double mean = readfirstlane_d(unit_mean(A, st, u, SEG ? A.unit_T[u] : T, lane));      // (SEG: T is the segment's span)
    const double lf = readfirstlane_d(st.logfloor);
    const double pole = A.pole;
    const double a_dec = A.a_dec;
    double* ylast_g = A.ylast + (int64_t)(SEG ? (int)blockIdx.x : u) * AFP_NBINS;      // (SEG: a private slot; the backward phase reads the rows)

    const int nch = (T + CF - 1) / CF;
    const int nch4 = (nch + 3) & ~3;                       // both waves run the same padded trip count
    // guard-free main part of the forward loops: chunks [0, nmain), nmain a multiple of 4 such that every frame either
    // wavefront touches there -- up to chunk nmain + PFC of the producer's prefetch -- lies before the last frame
    int nmain = ((T - 1) / CF - 1 - PFC) & ~3;
    if (nmain < 0 || A.sgram_dbg != nullptr || PROF || T >= (1 << 21)) nmain = 0;
    // backward chunks: as many whole frames as fit 64 record lanes
    const int CFB = K <= AFP_WAVE ? (AFP_WAVE / K > 0 ? AFP_WAVE / K : 1) : 1;
    const int CKB = CFB * K;                               // <= 64 records per backward chunk
    const int nchb = (T + CFB - 1) / CFB;
    const int nchb4 = (nchb + 3) & ~3;
    const bool sort_in_bwd = K <= SORT_IN_BWD_MAXK;        // else the forward pass stores its records ranked

    if (!scanner) {
        // =========================== PRODUCER wavefront ===========================
        if constexpr (CMP) {
            // lane L rebuilds bins 4L..4L+3: its nibble sits in word L >> 4 of the frame's mask, at bit 4 (L & 15); the
            // frame's values are stored in ascending bin order, so the lane's first value has rank
            //   popcount(words before) + popcount(word & bits below the nibble)        (at most two maxima among 4 bins)
            const int sh = 4 * (lane & 15);
            const unsigned mlo = sh < 32 ? ((1u << sh) - 1u) : 0xffffffffu;
            const unsigned mhi = sh < 32 ? 0u : ((1u << (sh - 32)) - 1u);
            const uint64_t* lmb = A.lmask + fb * 4 + (lane >> 4);
            const char* cvb = reinterpret_cast<const char*>(A.cvals + fb * CV_ROW);      // wave-uniform
            double cn = mean;                                           // c_t = mean * pole^t of the frame processed next
            unsigned wlo[4], whi[4], nb[4];
            double v0[4], v1[4];
            auto load_mask = [&](int ms, int t) {
                if (t > T - 1) t = T - 1;                               // clamped: always issued
                const uint64_t w = lmb[(int64_t)t * 4];
                wlo[ms] = (unsigned)w; whi[ms] = (unsigned)(w >> 32);
            };
            auto load_vals = [&](int vs, int t) {                       // (the mask of frame t sits in slot vs as well)
                if (t > T - 1) t = T - 1;
                const unsigned lo = wlo[vs], hi = whi[vs];
                const int own = __popc(lo) + __popc(hi);
                int incl = own;                                         // the 16 lanes of a DPP row share a word
                incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_BCAST15, 0xA, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_BCAST31, 0xC, 0xF, false);
                const int r0 = incl - own + __popc(lo & mlo) + __popc(hi & mhi);
                nb[vs] = ((sh < 32 ? lo : hi) >> (sh & 31)) & 15u;
                const double* row = reinterpret_cast<const double*>(cvb + (unsigned)t * (unsigned)(CV_ROW * 8)) + r0;
                v0[vs] = row[0]; v1[vs] = row[1];                       // (unconditional: vmcnt stays exact; row[1] may be the next frame's)
            };
            auto process = [&](int vs, int t, double* dst) {
                const unsigned n = nb[vs];
                const double s0 = v0[vs] - cn, s1 = v1[vs] - cn;
                // two maxima are never adjacent: bins 0 and 1 can only hold the lane's first value
                const double y2 = (n & 1u) ? s1 : s0, y3 = (n & 3u) ? s1 : s0;
                dpair o0, o1;
                o0.a = keep_if_max(s0, (n & 1u) != 0u); o0.b = keep_if_max(s0, (n & 2u) != 0u);
                o1.a = keep_if_max(y2, (n & 4u) != 0u); o1.b = keep_if_max(y3, (n & 8u) != 0u);
                dpair* o = reinterpret_cast<dpair*>(dst + 2 * lane);
                if (t < T) { o[0] = o0; o[64] = o1; }
                if (t == T - 1 && lane == 0) cshare = cn;
                cn = cn * pole;
            };
#pragma unroll
            for (int p = 0; p < 4; p++) load_mask(p, p);
#pragma unroll
            for (int p = 0; p < 4; p++) { load_vals(p, p); load_mask(p, p + 4); }
            process(0, 0, ring[0]);
            load_vals(0, 4); load_mask(0, 8);
            __syncthreads();                                            // (B0) frame 0 + Gs ready
            for (int cb = 0; cb < nch4; cb += 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int c = cb + k;                               // the scanner is on frame c: prepare c + 1
                    process((k + 1) & 3, c + 1, ring[(k + 1) & 1]);
                    load_vals((k + 1) & 3, c + 5); load_mask((k + 1) & 3, c + 9);
                    __syncthreads();                                    // (Bf) end of forward frame c
                }
            }
        } else if (!SEG || run_fwd) {
        double z[4] = {0.0, 0.0, 0.0, 0.0};
        if (SEG && seg_z0) seg_load_state(seg_z0, lane, z);          // the filter state k_hpf carried to this frame
        dpair raw[PFC][CF][2];
#pragma unroll
        for (int p = 0; p < PFC; p++) prod_load_chunk(L, fb, T, p, lane, raw[p]);
        prod_proc_chunk<RAW>(raw[0], 0, T, lane, lf, mean, pole, z, ring[0], ylast_g, A.sgram_dbg, fb);
        prod_load_chunk(L, fb, T, PFC, lane, raw[0]);
        __syncthreads();                                            // (B0) chunk 0 + Gs ready
        // MAIN part: chunk groups whose prepared chunks (c + 1) and prefetched chunks (c + 1 + PFC) lie wholly inside the
        // clip and before its last frame -- no bounds checks, loads through a running pointer
        int cb = 0;
        {
            const char* ubase = reinterpret_cast<const char*>(L + fb * AFP_NBINS);      // wave-uniform
            unsigned voff = (unsigned)((1 + PFC) * CF * AFP_NBINS * 8 + 32 * lane);       // (a unit's rows span < 4 GB: T < 2^21)
            for (; cb < nmain; cb += 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    prod_proc_chunk_main<RAW>(raw[(k + 1) & (PFC - 1)], lane, lf, mean, pole, z, ring[(k + 1) & 1]);
                    prod_load_chunk_ptr(ubase, voff, raw[(k + 1) & (PFC - 1)]);
                    __syncthreads();                                // (Bf) end of forward chunk cb + k
                }
            }
        }
        for (; cb < nch4; cb += 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = cb + k;                               // the scanner is on chunk c: prepare c+1
                prod_proc_chunk<RAW>(raw[(k + 1) & (PFC - 1)], c + 1, T, lane, lf, mean, pole, z, ring[(k + 1) & 1], ylast_g,
                                     A.sgram_dbg, fb);
                prod_load_chunk(L, fb, T, c + 1 + PFC, lane, raw[(k + 1) & (PFC - 1)]);
                // (the parked last column reaches the scanner through the barrier: __syncthreads orders global memory at
                //  workgroup scope, which is all two wavefronts of one workgroup need -- an agent-scope __threadfence here
                //  costs an L2 write-back + L1 invalidate per unit, microseconds that add up for short clips)
                __syncthreads();                                    // (Bf) end of forward chunk c
            }
        }
        }   // !CMP
        if (SEG && !run_bwd) return;
        if (SEG && !run_fwd) __syncthreads();                       // (B0') Gs ready (the forward prologue's barrier is skipped)
        // ---- backward: stream the forward survivors; chunk index jb counts from the END of the clip
        double rv[PFB];
        int rb[PFB];
        const int kl = lane < CKB ? lane : CKB - 1;
        const int gbase = (kl / K) * K;                             // first record lane of this lane's frame
        const int64_t emax = (fb + T) * (int64_t)K - 1;
        auto load_rec = [&](int p, int cc) {
            if (cc < 0) cc = 0;
            int64_t e = (fb + (int64_t)cc * CFB) * K + kl;
            if (e > emax) e = emax;
            rv[p] = A.cand_val[e]; rb[p] = A.cand_bin[e];
        };
        // records of one frame -> descending (value, bin), the order sorted(zip(vals, bins), reverse=True) gives (:241);
        // empty slots (bin < 0) go last.  Every lane learns its rank by looking at the K lanes of its frame.
        auto put_sorted = [&](int slot, double v, int b) {
            int dst = kl;
            if (sort_in_bwd) {
                const double mv = b >= 0 ? v : -INFINITY;
                const int mb = b >= 0 ? b : -1 - (kl - gbase);
                int rank = 0;
                for (int s = 0; s < K; s++) {
                    const double pv = bpermute_d(gbase + s, mv);
                    const int pb = __builtin_amdgcn_ds_bpermute((gbase + s) << 2, mb);
                    rank += (pv > mv || (pv == mv && pb > mb)) ? 1 : 0;
                }
                dst = gbase + rank;
            }
            cvring[slot][dst] = v; cbring[slot][dst] = b;
        };
#pragma unroll
        for (int p = 0; p < PFB; p++) load_rec(p, nchb - 1 - p);
        put_sorted(0, rv[0], rb[0]);
        load_rec(0, nchb - 1 - PFB);
        __syncthreads();                                            // (B1) first backward chunk ready
        for (int jb = 0; jb < nchb4; jb += 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = jb + k;                               // the scanner is on backward chunk j: prepare j+1
                put_sorted((k + 1) & 1, rv[(k + 1) & 3], rb[(k + 1) & 3]);
                load_rec((k + 1) & 3, nchb - 1 - (j + 1 + PFB));
                __syncthreads();                                    // (Bb) end of backward chunk j
            }
        }
        return;
    }

    // =========================== SCANNER wavefront ===========================
    if (!SEG && lane == 0) A.unit_mean[u] = mean;
    double thr[4];
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, fwd_wait = 0, bwd_wait = 0;
    unsigned long long pc_read = 0, pc_zero = 0, pc_fast = 0, pc_slow = 0, n_zero = 0, n_fast = 0, n_slow = 0;
    unsigned long long pb_empty = 0, pb_rec = 0, nb_empty = 0, nb_rec = 0, nb_records = 0, nb_kept = 0;
    if (PROF) tk0 = __builtin_readcyclecounter();

    // ---- initial forward threshold: spread the per-bin max over the first min(10,T) HPF'd columns
    //      (:204-206); those columns come straight from HBM, once per unit
    if (!SEG || run_fwd) {
        double vmax[4], y[4], z[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { z[j] = 0.0; vmax[j] = -INFINITY; y[j] = 0.0; }
        if (SEG && seg_z0) seg_load_state(seg_z0, lane, z);
        const int n0 = T < 10 ? T : 10;
        // five batches of 2 columns: the whole kernel has to stay within 64 VGPRs (see the note at CF above), and this
        // once-per-unit prologue must not be what sets the register count
#pragma unroll 1
        for (int h5 = 0; h5 < 5; h5++) {
            dpair pre[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const int t = 2 * h5 + tt;
                // CMP: the dense onset-filtered rows k_stft kept for exactly these columns (mean not yet subtracted)
                const dpair* p = CMP ? reinterpret_cast<const dpair*>(A.head + ((int64_t)u * CV_HEAD + (t < n0 ? t : n0 - 1)) * AFP_NBINS + 4 * lane)
                                     : reinterpret_cast<const dpair*>(L + (fb + (t < T ? t : T - 1)) * AFP_NBINS + 4 * lane);
                pre[tt][0] = p[0]; pre[tt][1] = p[1];
            }
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const int t = 2 * h5 + tt;
                if (t < n0) {
                    double raw[4] = {pre[tt][0].a, pre[tt][0].b, pre[tt][1].a, pre[tt][1].b};
                    if (CMP) {
                        // z[0] carries c_t = mean * pole^t, formed exactly like the producer's
                        if (t == 0) z[0] = mean;
#pragma unroll
                        for (int j = 0; j < 4; j++) y[j] = raw[j] - z[0];
                        z[0] = z[0] * pole;
                    } else {
                        hpf_step<RAW>(raw, lf, mean, pole, z, y);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) vmax[j] = fmax(vmax[j], y[j]);
                }
            }
        }
        __syncthreads();                                            // (B0) (Gs is needed by spread_all)
        if (PROF) tk1 = __builtin_readcyclecounter();
        spread_all(thr, vmax, lane, Gs);
        if (RAW && A.fwd_off) {                                     // backward prune of GIVEN peaks: the forward pass finds nothing
#pragma unroll
            for (int j = 0; j < 4; j++) thr[j] = INFINITY;
        }
        if (SEG && from_state) seg_load_state(init_state, lane, thr);         // repair: continue from the neighbour's state
    }

    // ---- forward pass (:214-230)
    if (PROF) tk2 = __builtin_readcyclecounter();
    int ev_lo = 0, ev_hi = 0;                                       // survivor records of the current frame, one per lane
    int eb = -1;
    if (!SEG || run_fwd) {
    for (int cb = 0; cb < nch4; cb += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = cb + k;
            const double* buf = ring[k & 1];
            double ych[CF][4];
            unsigned long long ta = 0, tb = 0;
            if (PROF) ta = __builtin_readcyclecounter();
#pragma unroll
            for (int i = 0; i < CF; i++) read_frame(buf + i * FROW, lane, ych[i]);      // whole chunk in flight at once
            if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tb = __builtin_readcyclecounter(); pc_read += tb - ta; }
#pragma unroll
            for (int i = 0; i < CF; i++) {
                const int t = c * CF + i;
                if (t < T) {
                    double (&y)[4] = ych[i];
                    if (PROF) tb = __builtin_readcyclecounter();
                    if (SEG && dump_entry && t == rec0) seg_store_state(dump_entry, lane, thr);   // the warm-up ends here
                    const unsigned long long m0 = __ballot(y[0] > thr[0]);     // strict >, :217 (non-maxima are -1)
                    const unsigned long long m1 = __ballot(y[1] > thr[1]);
                    const unsigned long long m2 = __ballot(y[2] > thr[2]);
                    const unsigned long long m3 = __ballot(y[3] > thr[3]);
                    const unsigned long long many = m0 | m1 | m2 | m3;
                    if (guard && t > 0) {
                        // (frame 0 is left out: the initial threshold is the spread maximum of the first columns (:204-206), so
                        //  `y == sthresh` holds there EXACTLY wherever a bin's own value dominates -- the same double on both sides,
                        //  through G[0] = 1 -- in the reference as here; from frame 1 on the threshold carries a factor a_dec)
                        const double d01 = fmin(fabs(y[0] - thr[0]), fabs(y[1] - thr[1]));
                        const double d23 = fmin(fabs(y[2] - thr[2]), fabs(y[3] - thr[3]));
                        nt |= __ballot(fmin(d01, d23) <= nte);
                    }
                    if (many != 0ull) {
                        unsigned long long c0 = m0, c1 = m1, c2 = m2, c3 = m3;
                        int n = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
                        if (n > K) {
                            // More candidates than maxpksperframe (:221 keeps the first K of the list sorted by (value, bin)
                            // descending): K rounds of wavefront arg-max pick the survivors.  This rare path only REDUCES the
                            // four candidate masks -- the threshold and the records are updated by the common code below.
                            unsigned cm = (y[0] > thr[0] ? 1u : 0u) | (y[1] > thr[1] ? 2u : 0u)
                                        | (y[2] > thr[2] ? 4u : 0u) | (y[3] > thr[3] ? 8u : 0u);
                            unsigned long long anym = many;
                            c0 = c1 = c2 = c3 = 0ull;
                            int cnt = 0;
                            double lastv = 0.0;                                // value of the candidate selected last
                            while (anym != 0ull && cnt < K) {
                                // lane-local best of the remaining candidates (ties -> larger bin)
                                double bv = -1.0;
                                int bs = -1;
#pragma unroll
                                for (int j = 0; j < 4; j++) if ((cm >> j) & 1u) { if (y[j] >= bv) { bv = y[j]; bs = j; } }
                                int wl;
                                if ((anym & (anym - 1)) == 0ull) {
                                    wl = __ffsll((long long)anym) - 1;         // a single candidate lane left
                                } else {
                                    const double wv = wave_max_uniform(bv);
                                    const unsigned long long wm = __ballot(bs >= 0 && bv == wv);
                                    if (wm == 0ull) break;                     // only reachable with NaN input
                                    wl = 63 - __clzll((long long)wm);          // highest lane = larger bin (:220)
                                }
                                const int ws = __builtin_amdgcn_readlane(bs, wl);
                                if (guard) lastv = readlane_d(bv, wl);
                                const unsigned long long bit = 1ull << wl;
                                if (ws == 0) c0 |= bit; else if (ws == 1) c1 |= bit; else if (ws == 2) c2 |= bit; else c3 |= bit;
                                if (lane == wl) cm &= ~(1u << ws);
                                cnt++;
                                anym = __ballot(cm != 0);
                            }
                            if (guard && anym != 0ull && cnt > 0) {
                                // the cut behind the K largest (:221): the best candidate left out against the last one taken
                                double bv = -INFINITY;
#pragma unroll
                                for (int j = 0; j < 4; j++) if ((cm >> j) & 1u) bv = fmax(bv, y[j]);
                                if (lastv - wave_max_uniform(bv) <= nte) nt |= 1ull;
                            }
                            n = cnt;
                        }
                        // Every remaining candidate is kept and the threshold updates commute (max), so no ordering is
                        // needed here: bump in ballot order and drop the record of the i-th candidate into lane i
                        // (v_writelane: no vector compare/select).
                        int idx = 0;
#define AFP_TAKE(J, MJ)                                                                 \
                        for (unsigned long long mm = (MJ); mm != 0ull;) {                       \
                            const int wl = __ffsll((long long)mm) - 1;                          \
                            mm = clear_bit(mm, wl);                                             \
                            const double val = readlane_d(y[J], wl);                            \
                            const int bin = 4 * wl + (J);                                       \
                            bump_s<J>(thr, val, wl, lane, Gs);             /* :226-228 */       \
                            put_record(ev_lo, ev_hi, eb, val, bin, idx);                        \
                            idx++;                                                              \
                        }
                        AFP_TAKE(0, c0)
                        AFP_TAKE(1, c1)
                        AFP_TAKE(2, c2)
                        AFP_TAKE(3, c3)
#undef AFP_TAKE
                        int rank = lane;
                        if (!sort_in_bwd && n > 1) {
                            // many peaks per frame allowed: store ranked by (val, bin) descending (:241), the backward
                            // producer does not sort
                            rank = 0;
                            const double evd = __hiloint2double(ev_hi, ev_lo);
                            for (int q = 0; q < n; q++) {
                                const double vq = readlane_d(evd, q);
                                const int bq = __builtin_amdgcn_readlane(eb, q);
                                rank += (vq > evd || (vq == evd && bq > eb)) ? 1 : 0;
                            }
                        }
                        // cand_bin was pre-filled with -1: only survivors are written
                        if (lane < n && (!SEG || t >= rec0)) {
                            reinterpret_cast<int2*>(A.cand_val)[(fb + t) * K + rank] = make_int2(ev_lo, ev_hi);
                            A.cand_bin[(fb + t) * K + rank] = eb;
                        }
                        if (clear && lane >= n && lane < K) A.cand_bin[(fb + t) * K + lane] = -1;      // (an earlier attempt's records)
                    } else if (clear && (!SEG || t >= rec0)) {
                        if (lane < K) A.cand_bin[(fb + t) * K + lane] = -1;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) thr[j] = thr[j] * a_dec;      // :230
                    if (PROF) {
                        const unsigned long long tc = __builtin_readcyclecounter();
                        const int n = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
                        if (n == 0) { pc_zero += tc - tb; n_zero++; } else if (n == 1) { pc_fast += tc - tb; n_fast++; } else { pc_slow += tc - tb; n_slow++; }
                    }
                }
            }
            // (the records reach the producer wave through the barrier below: workgroup-scope ordering, see the producer)
            unsigned long long tw0 = 0;
            if (PROF) tw0 = __builtin_readcyclecounter();
            __syncthreads();                                        // (Bf)
            if (PROF) fwd_wait += __builtin_readcyclecounter() - tw0;
        }
    }
    if (SEG && dump_exit) seg_store_state(dump_exit, lane, thr);     // state at entry of frame e
    }   // run_fwd
    if (SEG && !run_bwd) { flush_nt(); return; }
    if (SEG && !run_fwd) __syncthreads();                           // (B0') Gs ready

    // ---- backward pass (:233-253)
    if (PROF) tk3 = __builtin_readcyclecounter();
    {
        double ylast[4];
        // parked by the producer (fenced before (Bf)); SEG: the filtered last column of the virtual unit, from k_hpf
        const dpair* yl = SEG ? reinterpret_cast<const dpair*>(seg_yl + 4 * lane)
                              : reinterpret_cast<const dpair*>(ylast_g + 4 * lane);
        const dpair q0 = yl[0], q1 = yl[1];
        ylast[0] = q0.a; ylast[1] = q0.b; ylast[2] = q1.a; ylast[3] = q1.b;
        if (CMP) {                                                            // k_stft parked the row without the mean term
            const double cl = cshare;
#pragma unroll
            for (int j = 0; j < 4; j++) ylast[j] = ylast[j] - cl;
        }
        spread_all(thr, ylast, lane, Gs);                                     // :237
        if (SEG && from_state) seg_load_state(init_state, lane, thr);         // repair: the neighbour's state at entry of frame e
        // from here on the table is used in its linear layout (see bump_lin); only this wavefront touches it, and LDS
        // operations of one wavefront stay in program order
#if !SCAN_BWD_DEINT
        fill_gauss_linear(Gs, A.gauss, lane, AFP_WAVE);
#endif
    }
    // peak masks live LANE-DISTRIBUTED: lane q (0..3) holds the 64-bit word q of a 256-bit mask (other lanes stay 0), so
    // keeping / clearing a bin is a handful of straight-line vector instructions instead of a 4-way scalar branch tree.
    // c = peaks kept in the frame being scanned, p = the pending mask of frame t+1 (which :247-248 may still clear)
    int p_lo = 0, p_hi = 0;
    __syncthreads();                                                // (B1)
    if (PROF) tk4 = __builtin_readcyclecounter();
    const unsigned long long kmask = K >= 64 ? ~0ull : ((1ull << K) - 1ull);
    for (int jb = 0; jb < nchb4; jb += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = jb + k;
            const int c = nchb - 1 - j;                             // backward chunk j = frames [c*CFB, ...)
            if (c >= 0) {
                const double evc = cvring[k & 1][lane];
                const int ebc = lane < CKB ? cbring[k & 1][lane] : -1;
                const unsigned long long mvalid = __ballot(ebc >= 0);
                for (int i = CFB - 1; i >= 0; i--) {
                    const int t = c * CFB + i;
                    if (t < T) {
                        unsigned long long tq = 0;
                        if (PROF) tq = __builtin_readcyclecounter();
                        if (SEG) {
                            if (dump_entry && t == rtop) seg_store_state(dump_entry, lane, thr);      // the warm-up ends here
                            if (t == 0 && dump_exit) seg_store_state(dump_exit, lane, thr);          // state at entry of frame s
                        }
                        const int base = i * K;
                        const int cnt = __popcll((mvalid >> base) & kmask);
                        int c_lo = 0, c_hi = 0;
                        for (int r = 0; r < cnt; r++) {
                            const double val = readlane_d(evc, base + r);
                            const int bin = __builtin_amdgcn_readlane(ebc, base + r);
                            const int sub = bin & 3, owner = bin >> 2;        // both wave-uniform
                            // val >= sthresh[bin] (:242, >=): compare against all four threshold registers at once and
                            // pick lane `owner` of register `sub` on the scalar unit (no branch tree, no cross-lane read)
                            const unsigned long long g0 = __ballot(val >= thr[0]);
                            const unsigned long long g1 = __ballot(val >= thr[1]);
                            const unsigned long long g2 = __ballot(val >= thr[2]);
                            const unsigned long long g3 = __ballot(val >= thr[3]);
                            const unsigned long long g01 = (sub & 1) ? g1 : g0, g23 = (sub & 1) ? g3 : g2;
                            const unsigned long long gs = (sub & 2) ? g23 : g01;
                            if (guard && t != T - 1) {
                                // the same selection for |val - sthresh[bin]| <= eps; the lanes of the other three registers and of
                                // the other bins do not count.  (The first frame of the pass is left out: its threshold is the spread
                                // LAST column (:237), which a peak of that column equals exactly -- see the forward pass.)
                                // (`sub` is wave-uniform: a scalar branch picks the one register that holds the bin)
                                unsigned long long nb_;
                                if (sub & 2) nb_ = (sub & 1) ? __ballot(fabs(val - thr[3]) <= nte) : __ballot(fabs(val - thr[2]) <= nte);
                                else nb_ = (sub & 1) ? __ballot(fabs(val - thr[1]) <= nte) : __ballot(fabs(val - thr[0]) <= nte);
                                nt |= (nb_ >> owner) & 1ull;
                            }
                            if ((gs >> owner) & 1ull) {
                                if (PROF) nb_kept++;
#if SCAN_BWD_DEINT      // (r05 experiment, DESIGN.md: the conflict-free de-interleaved table behind a scalar branch tree; off)
                                bump(thr, val, bin, lane, Gs);
#else
                                bump_lin(thr, val, bin, lane, Gs);             // :244
#endif
                                const unsigned long long bit = 1ull << (bin & 63);
                                const bool me = lane == (bin >> 6);            // the lane that holds this word
                                const int blo = me ? (int)(unsigned)bit : 0, bhi = me ? (int)(unsigned)(bit >> 32) : 0;
                                c_lo |= blo; c_hi |= bhi;                      // keep (bin, t)
                                p_lo &= ~blo; p_hi &= ~bhi;                    // :247-248 clears (bin, t+1)
                            }                                                  // else :251 drops (bin, t)
                        }
                        // masks were pre-zeroed: only non-empty frames are written (SEG: the masks of frames s + 1 .. e are this
                        // segment's; `clear`: an earlier attempt may have left bits behind)
                        if ((!SEG || t < rtop) && ((clear && t + 1 < (SEG ? 0x7fffffff : T)) || __ballot((p_lo | p_hi) != 0) != 0ull)) {
                            if (lane < 4) A.masks[(fb + t + 1) * 4 + lane] = ((unsigned long long)(unsigned)p_hi << 32) | (unsigned)p_lo;
                        }
                        p_lo = c_lo; p_hi = c_hi;
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) thr[jj] = a_dec * thr[jj];  // :252
                        if (PROF) {
                            const unsigned long long dq = __builtin_readcyclecounter() - tq;
                            if (cnt == 0) { pb_empty += dq; nb_empty++; } else { pb_rec += dq; nb_rec++; nb_records += cnt; }
                        }
                    }
                }
            }
            unsigned long long tw0 = 0;
            if (PROF) tw0 = __builtin_readcyclecounter();
            __syncthreads();                                        // (Bb)
            if (PROF) bwd_wait += __builtin_readcyclecounter() - tw0;
        }
    }
    if ((!SEG || seg_bottom) && ((clear) || __ballot((p_lo | p_hi) != 0) != 0ull)) {
        if (lane < 4) A.masks[fb * 4 + lane] = ((unsigned long long)(unsigned)p_hi << 32) | (unsigned)p_lo;
    }
    flush_nt();
    if (PROF && lane == 0) {
        unsigned long long* o = A.prof + (size_t)u * 32;
        o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = tk3; o[4] = tk4; o[5] = __builtin_readcyclecounter(); o[6] = (unsigned long long)T;
        o[7] = (fwd_wait << 32) | (bwd_wait & 0xffffffffull);
        o[8] = pc_read; o[9] = pc_zero; o[10] = pc_fast; o[11] = pc_slow; o[12] = n_zero; o[13] = n_fast; o[14] = n_slow; o[15] = 0;
        o[16] = pb_empty; o[17] = pb_rec; o[18] = nb_empty; o[19] = nb_rec; o[20] = nb_records; o[21] = nb_kept;
    }
}

// The kernel: one workgroup per unit.  CMP: the compact rows of k_stft<ST, true>; a unit that needed the floor (UNIT_CORR:
// its chunks were transformed again by the dense k_stft) takes the dense path IN THE SAME LAUNCH -- a unit's scan lasts as
// long whether 1 or 1024 of them run (a sequential chain per unit), so a second launch for a handful of units would cost
// a whole extra scan time.
template <bool PROF, int PFC, bool RAW = false, bool CMP = false, bool GUARD = false>
__global__ __launch_bounds__(2 * AFP_WAVE) SCAN_OCC
void k_scan(ScanArgs A)
{
    __shared__ double Gs[512];
    __shared__ __attribute__((aligned(16))) double ring[2][CF * FROW];       // forward ring (2 slots of CF frames)
    __shared__ double cshare;                                                // CMP: c_(T-1), handed from the producer to the scanner
#if !SCAN_SMALL_LDS
    __shared__ double cvring_s[2][AFP_WAVE];                                // backward record ring
    __shared__ int cbring_s[2][AFP_WAVE];
#else
    double (*cvring_s)[AFP_WAVE] = nullptr;
    int (*cbring_s)[AFP_WAVE] = nullptr;
#endif
    if constexpr (CMP) {
        if (A.stats[blockIdx.x].flags & UNIT_CORR) scan_unit<PROF, PFC, RAW, false, false, GUARD>(A, Gs, ring, cshare, cvring_s, cbring_s);
        else scan_unit<PROF, PFC, RAW, true, false, GUARD>(A, Gs, ring, cshare, cvring_s, cbring_s);
    } else {
        scan_unit<PROF, PFC, RAW, false, false, GUARD>(A, Gs, ring, cshare, cvring_s, cbring_s);
    }
}

#if !SCAN_SMALL_LDS
// Segment kernels (few long units: a single file).  The phase (forward / backward pass) is a launch argument; per phase:
//   k_scan_seg, seg_repair 0 (one workgroup per segment): every segment scans its warm-up and its own frames, leaving the
//     state it had at entry of its first own frame (ENTRY) and at its end (EXIT0);
//   k_seg_flags (one wavefront per segment): segment i is FLAGGED when ENTRY[i] is not the bit pattern EXIT0[neighbour] -- its
//     warm-up did not reach the true state (typically a quiet stretch after a loud one: the thresholds of :226-230 remember
//     the loud part for hundreds of frames);
//   k_scan_seg, seg_repair 1, the CHAIN launch (one workgroup per UNIT): flagged segments come in runs, and a run is
//     inherently sequential -- the workgroup walks its unit's segments in pass order, skipping from flagged segment to
//     flagged segment, and re-runs each from the final end state of the segment before it (no warm-up); after a re-run the
//     NEXT segment is checked against the new end state (not the precomputed flag) and re-run as well unless that state IS
//     its ENTRY, and so on until a first-launch result stands again.  The first launch's states stay untouched; a re-run
//     leaves its states in the *1 planes and marks seg_rerun.  Cost: nothing but two tiny launches when every warm-up
//     converged; the sequential scan of just the unconverged stretches otherwise.
__device__ __forceinline__ bool seg_state_differs(const double* a, const double* b, int lane)
{
    const unsigned long long* pa = reinterpret_cast<const unsigned long long*>(a + 4 * lane);
    const unsigned long long* pb = reinterpret_cast<const unsigned long long*>(b + 4 * lane);
    return __ballot(pa[0] != pb[0] || pa[1] != pb[1] || pa[2] != pb[2] || pa[3] != pb[3]) != 0ull;
}
__global__ __launch_bounds__(AFP_WAVE)
void k_seg_flags(ScanArgs A)
{
    const int seg = blockIdx.x;
    const SegDesc sd = A.segs[seg];
    const bool fwdp = A.seg_phase == SEG_FWD;
    const int nb = fwdp ? sd.prev : sd.next;
    const int64_t NS = (int64_t)A.nseg * AFP_NBINS;
    bool f = false;
    if (nb >= 0 && !(A.stats[sd.unit].flags & (UNIT_ZERO | UNIT_EMPTY)))
        f = seg_state_differs(A.seg_state + (fwdp ? ST_FENTRY : ST_BENTRY) * NS + (int64_t)seg * AFP_NBINS,
                              A.seg_state + (fwdp ? ST_FEXIT0 : ST_BEXIT0) * NS + (int64_t)nb * AFP_NBINS, threadIdx.x);
    if (threadIdx.x == 0) A.seg_flag[seg] = f ? 1 : 0;
}
#ifndef SEG_PFC
#define SEG_PFC 2                              // forward chunks (of CF frames) the segment scan's producer keeps in flight
#endif
template <bool GUARD>
__global__ __launch_bounds__(2 * AFP_WAVE)
void k_scan_seg(ScanArgs A)
{
    __shared__ double Gs[512];
    __shared__ __attribute__((aligned(16))) double ring[2][CF * FROW];
    __shared__ double cshare;
    __shared__ double cvring_s[2][AFP_WAVE];
    __shared__ int cbring_s[2][AFP_WAVE];
    const int lane = threadIdx.x & 63;
    const bool fwdp = A.seg_phase == SEG_FWD;
    const int64_t NS = (int64_t)A.nseg * AFP_NBINS;
    double* entry0 = A.seg_state + (fwdp ? ST_FENTRY : ST_BENTRY) * NS;
    double* exit0 = A.seg_state + (fwdp ? ST_FEXIT0 : ST_BEXIT0) * NS;
    double* entry1 = A.seg_state + (fwdp ? ST_FENTRY1 : ST_BENTRY1) * NS;
    double* exit1 = A.seg_state + (fwdp ? ST_FEXIT1 : ST_BEXIT1) * NS;
    SegRun sr;
    if (!A.seg_repair) {
        const int cur = blockIdx.x;
        if (A.stats[A.segs[cur].unit].flags & (UNIT_ZERO | UNIT_EMPTY)) return;      // nothing to scan (the final check skips these units too)
        sr.seg = cur; sr.init_state = nullptr;
        sr.entry_out = entry0 + (int64_t)cur * AFP_NBINS; sr.exit_out = exit0 + (int64_t)cur * AFP_NBINS;
        scan_unit<false, SEG_PFC, false, false, true, GUARD>(A, Gs, ring, cshare, cvring_s, cbring_s, sr);
        return;
    }
    // ---- chain launch: this workgroup owns unit blockIdx.x, segments [s0, s1) in ascending frame order
    const int u = blockIdx.x;
    if (A.stats[u].flags & (UNIT_ZERO | UNIT_EMPTY)) return;
    const int s0 = A.seg_ufirst[u], s1 = A.seg_ufirst[u + 1];
    const int n = s1 - s0;
    int32_t* rerun = A.seg_rerun + (fwdp ? 0 : A.nseg);
    // k-th segment in pass order (k = 0 starts from the true state and is never flagged)
    auto seg_at = [&](int k) { return fwdp ? s0 + k : s1 - 1 - k; };
    int k = 1;
    while (k < n) {
        // next flagged segment at or after position k: 64 flags per look
        {
            const int kk = k + lane;
            const bool f = kk < n && A.seg_flag[seg_at(kk)] != 0;
            const unsigned long long m = __ballot(f);
            if (m == 0ull) { k += 64; continue; }
            k += __ffsll((long long)m) - 1;
        }
        // a run starts at position k: its predecessor's first-launch end state is the true state
        const double* state = exit0 + (int64_t)seg_at(k - 1) * AFP_NBINS;
        for (; k < n; k++) {
            const int cur = seg_at(k);
            if (!seg_state_differs(entry0 + (int64_t)cur * AFP_NBINS, state, lane)) break;      // this first-launch result stands: the run is over
            sr.seg = cur; sr.init_state = state;
            sr.entry_out = entry1 + (int64_t)cur * AFP_NBINS; sr.exit_out = exit1 + (int64_t)cur * AFP_NBINS;
            scan_unit<false, SEG_PFC, false, false, true, GUARD>(A, Gs, ring, cshare, cvring_s, cbring_s, sr);
            __syncthreads();                                            // both wavefronts are through; the new states are visible to both
            if (threadIdx.x == 0) { rerun[cur] = 1; atomicAdd(&A.seg_status[fwdp ? 1 : 2], 1); }
            state = sr.exit_out;
        }
        // (position k, if any, was checked against the state that reaches it and stands; flagged segments behind it start new runs)
        k++;
    }
}

// Final check of every segment boundary (one wavefront per segment): the state a segment started its own frames from must
// be the bit pattern its neighbour ended with, in both passes -- then, by induction from the unit's first (last) segment,
// every segment scanned its frames from the true state.  A mismatch marks the UNIT (seg_ufail) and counts it in
// seg_status[0]; the dense sequential kernel launched next re-does the marked units.
__global__ __launch_bounds__(AFP_WAVE)
void k_seg_verify(ScanArgs A)
{
    const int seg = blockIdx.x;
    const int lane = threadIdx.x;
    const SegDesc sd = A.segs[seg];
    if (A.stats[sd.unit].flags & (UNIT_ZERO | UNIT_EMPTY)) return;
    const int64_t NS = (int64_t)A.nseg * AFP_NBINS;
    const int32_t* rr = A.seg_rerun;
    bool bad = A.seg_force_fail != 0;
    if (sd.prev >= 0) {
        const double* en = A.seg_state + (rr[seg] ? ST_FENTRY1 : ST_FENTRY) * NS + (int64_t)seg * AFP_NBINS;
        const double* ex = A.seg_state + (rr[sd.prev] ? ST_FEXIT1 : ST_FEXIT0) * NS + (int64_t)sd.prev * AFP_NBINS;
        bad = bad || seg_state_differs(ex, en, lane);
    }
    if (sd.next >= 0) {
        const double* en = A.seg_state + (rr[A.nseg + seg] ? ST_BENTRY1 : ST_BENTRY) * NS + (int64_t)seg * AFP_NBINS;
        const double* ex = A.seg_state + (rr[A.nseg + sd.next] ? ST_BEXIT1 : ST_BEXIT0) * NS + (int64_t)sd.next * AFP_NBINS;
        bad = bad || seg_state_differs(ex, en, lane);
    }
    if (bad && lane == 0 && atomicExch(&A.seg_ufail[sd.unit], 1) == 0) atomicAdd(&A.seg_status[0], 1);
}

// k_hpf: floor + mean (audfprint_analyze.py:285-286) and the onset filter lfilter([1,-1],[1,-pole]) (:293-295) carried through
// a whole unit, so that the scan can be cut into segments: the filter state does not converge bit-exactly, it has to be
// carried through the unit once.  Per frame and bin the chain is add, mul, add (the same separately rounded operations as
// hpf_step); nothing is written but the state at the frames the segments start from.
//
// What bounds it is how many instructions ONE wavefront has to get through per frame: a lone wavefront pays issue time for
// every instruction, scalar ones and waits included (tools/chain_latency.hip at 2.4 GHz: the three dependent FP64
// operations 21 cycles; the same with a taken branch 47; with an LDS read and floor + mean 40).  Not bytes, not the number of
// loads in flight, not the CU count, not the clock -- all measured on a 300 s clip = 12 920 frames (rocprofv3):
//   446 us  4 wavefronts x 64 bins in one workgroup, a listed-frame test (taken branch) in every step, loop-head wait vmcnt(3)
//   389 us  one wavefront of 16 / 32 / 64 bins per workgroup on 16 / 8 / 4 CUs, one test per 8 frames, 56 rows in flight
//   410 us  the same with four frames per load instruction (lanes = frame x bin, transposed through LDS)
//   302 us  floor + mean moved to a second wavefront (below), records written by the filter wavefront
//   274 us  one test per PHASE (32 frames) on a mask the loader prepares
//   218 us  warm-up = a multiple of the segment length (afp_abi.hip): one listed frame per segment instead of three
//   169 us  that one frame, always the first of its phase, recorded without leaving the straight-line code
// So the work of a frame is split over two wavefronts:
//   * the LOADER fetches the rows -- one load instruction brings HPF_FR = 4 FRAMES of the workgroup's HPF_BINS = 16 bins
//     (lane = frame-in-group x bin), 16 instructions in flight -- applies floor and mean (the two operations that do not
//     depend on the filter state), leaves  xx = max(raw, floor) - mean  in an LDS ring, and marks the phase's listed frames
//     in a mask;
//   * the FILTER wavefront reads the four frames of ITS bin back (broadcast reads: every lane gets its bin's frames; lanes
//     16..63 carry copies of lanes 0..15) and runs only  y = xx + z ; z = (-xx) + pole y  -- 96 dependent operations per
//     phase in straight-line code (848 cycles per phase measured, 27 per frame).
// They work in PHASES of HPF_PG groups (32 frames): the loader fills one half of the ring while the filter consumes the
// other, one s_barrier per phase.  The filter wavefront -- which loads nothing from memory -- writes the records of listed
// frames itself: stores issued by the LOADER would share its vmcnt with the row loads, and the compiler then drains every
// load in flight in front of them.  A unit's 256 bins are split over 16 workgroups (the bins are independent).
#ifndef HPF_BINS
#define HPF_BINS 16
#endif
#define HPF_FR (AFP_WAVE / HPF_BINS)
#ifndef HPF_PG
#define HPF_PG 8
#endif
__global__ __launch_bounds__(2 * AFP_WAVE)
void k_hpf(HpfArgs A)
{
    constexpr int FR = HPF_FR, PG = HPF_PG, PFR = PG * FR, G = 2 * PG;      // G load instructions (two phases) in flight
    static_assert(PFR <= 32, "one 32-bit mask of listed frames per phase");
    __shared__ double ring[2][PG][AFP_WAVE];
    __shared__ unsigned pmask[2];                                   // per ring half: which frames of the phase are listed, and the
    __shared__ int pfirst[2];                                       //   index of the first of their records (written by the loader)
    __shared__ int dfr_s[HPF_MAX_DUMPS + 1];                        // the unit's listed frames (read back with LDS loads: no vmcnt)
    // chunk mode: this workgroup filters frames [t_begin, t_end) of its unit (afp_common.h, HpfChunk); frames are numbered
    // from t_begin below
    const bool CH = A.chunks != nullptr;
    HpfChunk ch;
    if (CH) ch = A.chunks[blockIdx.x];
    const int u = CH ? ch.unit : (int)blockIdx.x;
    const int tb0 = CH ? ch.t_begin : 0;
    const int T = A.unit_T[u];
    const UnitStats st = A.stats[u];
    if (T <= 0 || (st.flags & UNIT_ZERO)) {
        // nothing to filter (the scan skips such units too) -- but a chunk's boundary slots must not keep an earlier batch's
        // bits: k_hpf_verify compares them, and a stale pair would send the whole batch to the sequential kernel
        if (CH && threadIdx.x < HPF_BINS) {
            const int b = blockIdx.y * HPF_BINS + threadIdx.x;
            if (ch.zmid >= 0) A.zbnd[(int64_t)ch.zmid * AFP_NBINS + b] = 0.0;
            if (ch.zend >= 0) A.zbnd[(int64_t)ch.zend * AFP_NBINS + b] = 0.0;
        }
        return;
    }
    const int lane = threadIdx.x & (AFP_WAVE - 1);
    const int tid = lane & (HPF_BINS - 1);                          // slot of the lane's bin inside the workgroup's slice
    const int fr = lane / HPF_BINS;                                 // frame of a load group this lane fetches
    const int bin = blockIdx.y * HPF_BINS + tid;
    const bool owner = lane < HPF_BINS;
    const bool loader = threadIdx.x >= AFP_WAVE;
    const int d0 = CH ? ch.d0 : A.dump_off[u], dend = CH ? ch.d1 : A.dump_off[u + 1];
    const int nd = dend - d0;
    if (nd <= 0 && !CH) return;
    if (nd > HPF_MAX_DUMPS) {                          // (never: the host sizes the segments so that the list fits)
        if (threadIdx.x == 0) atomicOr(A.fail, 1);     // the sequential kernel takes over
        return;
    }
    auto bar = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    for (int i = threadIdx.x; i <= nd && i <= HPF_MAX_DUMPS; i += 2 * AFP_WAVE) dfr_s[i] = i < nd ? A.dump_frame[d0 + i] - tb0 : 0x7fffffff;
    __syncthreads();
    const int Tl = CH ? ch.t_end - tb0 : dfr_s[nd - 1] + 1;         // nothing is recorded after the last listed frame
    const int ng = (Tl + FR - 1) / FR;                              // load groups
    const int nph = (ng + PG - 1) / PG;                             // phases (both wavefronts pass 1 + nph barriers)
    if (loader) {
        double corr = 0.0;
        if (st.flags & UNIT_CORR) {                                // the same ordered sum as unit_mean()
            const int64_t b0 = A.unit_bbase[u], b1 = A.unit_bbase[u + 1];
            for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) corr += A.blk_corr[b];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) corr += shfl_xor_d(corr, s);
        }
        const double mean = (st.lsum + corr) / (257.0 * (double)T);
        const double lf = st.logfloor;
        const double* base = A.logS + (A.unit_fbase[u] + tb0) * AFP_NBINS + bin;
        // group g = frames FR g .. FR g + FR - 1: this lane's element (clamped at the end: always a valid address; the
        // filter never uses a frame >= Tl)
        auto gload = [&](int g) -> double {
            int t = g * FR + fr;
            t = t < Tl ? t : Tl - 1;
            return base[(int64_t)t * AFP_NBINS];
        };
        double x[G];
#pragma unroll
        for (int k = 0; k < G; k++) { x[k] = gload(k); asm volatile("" ::: "memory"); }      // (issue order = use order)
        // phase q -> ring[q & 1]; its groups sit in x[(q & 1) PG + i]
        int ld = 0, lnext = dfr_s[0];                               // the loader's own cursor through the listed frames
        auto fill = [&](int q, auto HALF) {
            constexpr int H = decltype(HALF)::value;
            {   // the listed frames of phase q as a mask (wave-uniform): the filter then spends no instruction looking for them
                const int tb = q * PFR;
                unsigned m = 0;
                const int first = ld;
                while (lnext < tb + PFR) { m |= 1u << (lnext - tb); ld++; lnext = dfr_s[ld]; }      // (dfr_s[nd] is a sentinel)
                if (lane == 0) { pmask[H] = m; pfirst[H] = first; }
            }
#pragma unroll
            for (int i = 0; i < PG; i++) {
                double raw = x[H * PG + i];
                asm volatile("" : "+v"(raw) :: "memory");          // (the use stays HERE: hoisted above younger loads it becomes a vmcnt(0))
                const double xx = fmax(raw, lf) - mean;
                ring[H][i][lane] = xx;
                x[H * PG + i] = gload(q * PG + i + G);
                asm volatile("" ::: "memory");
            }
        };
        fill(0, std::integral_constant<int, 0>{});
        bar();                                                      // phase 0 is in the ring
        // two phases per trip, so that each half's rows keep their registers (with the half chosen at run time the compiler
        // rotates the sixteen row registers by copies -- behind a vmcnt(0): every load in flight drained once per phase)
        int p = 0;
        for (; p + 2 < nph; p += 2) {                               // (both fills unconditional: the load counts stay exact)
            fill(p + 1, std::integral_constant<int, 1>{});
            bar();                                                  // the filter is done with phase p; phase p + 1 is in the ring
            fill(p + 2, std::integral_constant<int, 0>{});
            bar();
        }
        if (p + 1 < nph) { fill(p + 1, std::integral_constant<int, 1>{}); bar(); bar(); }
        else if (p < nph) bar();
        return;
    }
    // ---- FILTER wavefront.  A lone wavefront pays issue time for EVERY instruction, scalar ones included (the three
    // dependent operations of a frame take 21 cycles; with some three more instructions per frame around them -- a test for a
    // listed frame per group, waits, LDS reads -- the same loop took 56): the phase is straight-line code but for one test of
    // the loader's mask per phase, and one per group only in phases that hold a listed frame.
    const double pole = A.pole;
    double z = 0.0;
    int pmid = -1;                                                  // chunk mode: the phase at whose entry the own range starts
    if (CH) {
        // entry state: the granules before t_begin, folded (z~ = L_j + pole^GRAN z~: a few ulps off the sequential state,
        // which the HPF_WARM frames ahead of the own range absorb -- k_hpf_verify checks that they did)
        // (at most HPF_FOLD granules: what lies further back weighs pole^2048 = 1e-18 and the loads are issued together)
        const double* L = A.gran + (int64_t)ch.zin_first * AFP_NBINS + bin;
        double lv[HPF_FOLD];
#pragma unroll
        for (int j = 0; j < HPF_FOLD; j++) lv[j] = j < ch.zin_n ? L[(int64_t)j * AFP_NBINS] : 0.0;
#pragma unroll
        for (int j = 0; j < HPF_FOLD; j++) z = j < ch.zin_n ? fma(A.polepow, z, lv[j]) : z;
        if (ch.zmid >= 0) pmid = (ch.own - tb0) / PFR;
    }
    unsigned long long* prof = (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) ? A.prof : nullptr;
    bar();
    for (int p = 0; p < nph; p++) {
        const int half = p & 1;
        if (prof && p < 2048) prof[4 * p] = __builtin_readcyclecounter();
        if (p == pmid && owner) A.zbnd[(int64_t)ch.zmid * AFP_NBINS + bin] = z;      // (wave-uniform test, once per phase)
        // the whole phase is read up front (32 values per lane, returned in order): the LDS latency is paid once per phase,
        // not once per group in front of the chain
        int mv = (int)pmask[half], fv = pfirst[half];               // (first in the LDS queue: the chain starts behind the first row read)
        asm volatile("" : "+v"(mv), "+v"(fv) :: "memory");
        double xa[PG][FR];
#pragma unroll
        for (int i = 0; i < PG; i++)
#pragma unroll
            for (int j = 0; j < FR; j++) xa[i][j] = ring[half][i][j * HPF_BINS + tid];
        asm volatile("" ::: "memory");
        const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane(mv);
        const int first = __builtin_amdgcn_readfirstlane(fv);
        if (__builtin_expect(m <= 1u, 1)) {
            // no listed frame, or only the first frame of the phase (with the default cut every listed frame is a segment start,
            // segments are two phases long: every other phase): its record is the state as it stands and the first y
            if (m != 0u && owner) {
                double* o = A.dump_state + (int64_t)(d0 + first) * 2 * AFP_NBINS + bin;
                o[0] = z;
                o[AFP_NBINS] = xa[0][0] + z;
            }
#pragma unroll
            for (int i = 0; i < PG; i++)
#pragma unroll
                for (int j = 0; j < FR; j++) {
                    const double yy = xa[i][j] + z;
                    z = (-xa[i][j]) + pole * yy;
                }
        } else {
#pragma unroll
            for (int i = 0; i < PG; i++) {
                const unsigned gm = (m >> (FR * i)) & ((1u << FR) - 1u);
                if (gm == 0u) {
#pragma unroll
                    for (int j = 0; j < FR; j++) {
                        const double yy = xa[i][j] + z;
                        z = (-xa[i][j]) + pole * yy;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FR; j++) {
                        const double yy = xa[i][j] + z;
                        if ((gm >> j) & 1u) {                       // a listed frame leaves the state at its entry and its filtered value
                            const int rec = first + __builtin_popcount(m & ((1u << (FR * i + j)) - 1u));
                            if (owner) {
                                double* o = A.dump_state + (int64_t)(d0 + rec) * 2 * AFP_NBINS + bin;
                                o[0] = z;
                                o[AFP_NBINS] = yy;
                            }
                        }
                        z = (-xa[i][j]) + pole * yy;
                    }
                }
            }
        }
        if (prof && p < 2048) prof[4 * p + 1] = __builtin_readcyclecounter();
        bar();
    }
    // chunk mode: the state this chunk ends with (t_end - t_begin is a whole number of phases wherever zend is asked for)
    if (CH && ch.zend >= 0 && owner) A.zbnd[(int64_t)ch.zend * AFP_NBINS + bin] = z;
}
// chunk mode, the check: boundary i holds in slots 2 i (the earlier chunk's end state) and 2 i + 1 (the later chunk's state at
// the same frame, reached through its warm-up): every bin must carry the SAME bit pattern.  One mismatch and the batch's units
// are re-done by the sequential kernel (fail = ScanArgs::only_if[3]).
__global__ __launch_bounds__(256)
void k_hpf_verify(const double* zbnd, int nbnd, int32_t* fail, int force)
{
    bool bad = force != 0;
    for (int i = blockIdx.x; i < nbnd; i += gridDim.x) {
        const unsigned long long a = ((const unsigned long long*)zbnd)[(int64_t)(2 * i) * AFP_NBINS + threadIdx.x];
        const unsigned long long b = ((const unsigned long long*)zbnd)[(int64_t)(2 * i + 1) * AFP_NBINS + threadIdx.x];
        bad = bad || a != b;
    }
    if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) atomicOr(fail, 1);
}
extern "C" void afp_launch_hpf(const HpfArgs* a, int nunits, hipStream_t st)
{
    if (nunits > 0) hipLaunchKernelGGL(k_hpf, dim3(nunits, AFP_NBINS / HPF_BINS), dim3(2 * AFP_WAVE), 0, st, *a);
}
extern "C" void afp_launch_hpf_verify(const double* zbnd, int nbnd, int32_t* fail, int force, hipStream_t st)
{
    if (nbnd > 0 || force) hipLaunchKernelGGL(k_hpf_verify, dim3(nbnd > 64 ? 64 : (nbnd > 0 ? nbnd : 1)), dim3(256), 0, st, zbnd, nbnd, fail, force);
}
extern "C" void afp_launch_scan_seg(const ScanArgs* a, int nunits, hipStream_t st)
{
    if (a->nseg <= 0) return;
    const bool guard = a->nt_eps > 0.0;            // near-tie guard on: the guarded instantiation
    const int grid = a->seg_repair ? nunits : a->nseg;
    if (a->seg_repair) hipLaunchKernelGGL(k_seg_flags, dim3(a->nseg), dim3(AFP_WAVE), 0, st, *a);
    if (guard) hipLaunchKernelGGL(k_scan_seg<true>, dim3(grid), dim3(2 * AFP_WAVE), 0, st, *a);
    else hipLaunchKernelGGL(k_scan_seg<false>, dim3(grid), dim3(2 * AFP_WAVE), 0, st, *a);
}
extern "C" void afp_launch_seg_verify(const ScanArgs* a, hipStream_t st)
{
    if (a->nseg > 0) hipLaunchKernelGGL(k_seg_verify, dim3(a->nseg), dim3(AFP_WAVE), 0, st, *a);
}
#endif

#if !SCAN_SMALL_LDS
// popcount of the final masks: the per-frame peak counts the (col, bin) list output is compacted with (only when
// peak lists are wanted -- the hash path never needs them)
__global__ __launch_bounds__(256)
void k_mask_popc(const uint64_t* __restrict__ masks, int32_t* __restrict__ pcnt, int64_t nframes)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= nframes) return;
    const uint64_t* m = masks + f * 4;
    pcnt[f] = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
}
extern "C" void afp_launch_mask_popc(const uint64_t* masks, int32_t* pcnt, int64_t nframes, hipStream_t st)
{
    if (nframes > 0) hipLaunchKernelGGL(k_mask_popc, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, st, masks, pcnt, nframes);
}
extern "C" void afp_launch_unit_stats(const StatsArgs* a, hipStream_t st)
{
    if (a->nunits > 0) hipLaunchKernelGGL(k_unit_stats, dim3(a->nunits), dim3(AFP_WAVE), 0, st, *a);
}
static int floor_corr_per(const CorrArgs* a, int nblk)
{
    int per = (int)(((int64_t)nblk / a->nunits + 3) / 4);            // about four chunks per workgroup
    if (per < 1) per = 1;
    if (per > 256) per = 256;
    return per;
}
extern "C" void afp_launch_floor_corr(const CorrArgs* a, int nblk, hipStream_t st)
{
    if (a->nunits <= 0) return;
    hipLaunchKernelGGL(k_floor_corr, dim3(a->nunits, floor_corr_per(a, nblk)), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_stats_corr(const StatsArgs* s, const CorrArgs* a, int nblk, hipStream_t st)
{
    if (a->nunits <= 0) return;
    hipLaunchKernelGGL(k_stats_corr, dim3(a->nunits, floor_corr_per(a, nblk)), dim3(256), 0, st, *s, *a);
}
#endif
#if SCAN_SMALL_LDS
// Measurement aid (AFP_SCAN_DUMMY=<microseconds>, tools/: what does the scan cost the STFT beside it?): a kernel with the
// scan's footprint -- 2 wavefronts per workgroup, 64 VGPRs, 8 KB of LDS, one workgroup per unit -- that only sleeps.
__global__ __launch_bounds__(2 * AFP_WAVE) SCAN_OCC
void k_scan_dummy(int usec, double* sink)
{
    __shared__ double pad[1024];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (threadIdx.x == 0) pad[blockIdx.x & 1023] = 1.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)usec * 100ull) __builtin_amdgcn_s_sleep(32);
    if (pad[threadIdx.x & 1023] == 12345.0) sink[0] = 1.0;
}
extern "C" void afp_launch_scan_dummy(int nunits, int usec, double* sink, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_dummy, dim3(nunits), dim3(2 * AFP_WAVE), 0, st, usec, sink);
}
// compact rows in (k_stft<ST, true>); the units that needed the floor are skipped (afp_launch_scan_small with only_corr follows)
extern "C" void afp_launch_scan_compact(const ScanArgs* a, int nunits, hipStream_t st)
{
    if (nunits <= 0) return;
    if (a->prof) hipLaunchKernelGGL((k_scan<true, 4, false, true>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
    else if (a->nt_eps > 0.0) hipLaunchKernelGGL((k_scan<false, 4, false, true, true>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
    else hipLaunchKernelGGL((k_scan<false, 4, false, true>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
}
extern "C" void afp_launch_scan_small(const ScanArgs* a, int nunits, hipStream_t st)
#else
extern "C" void afp_launch_scan(const ScanArgs* a, int nunits, hipStream_t st)
#endif
{
    if (nunits <= 0) return;
    static int force_pfc = -1;
    if (force_pfc < 0) { const char* e = getenv("AFP_SCAN_PFC"); force_pfc = e ? atoi(e) : 0; }
    // frames the producer keeps in flight: 4 (62 VGPRs: the most that fits the 64-register budget, see CF) for the
    // batch variant -- its forward pass is HBM-latency bound, time per frame ~ latency / frames in flight
    const int pfc = force_pfc ? force_pfc : (SCAN_SMALL_LDS ? 4 : 2);
#if SCAN_SMALL_LDS
    if (a->prof) hipLaunchKernelGGL((k_scan<true, 4>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);   // same depth as production
#else
    if (a->raw_rows) { hipLaunchKernelGGL((k_scan<false, 2, true>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a); return; }
    if (a->prof) hipLaunchKernelGGL((k_scan<true, 4>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
#endif
    // near-tie guard on (afp_set_neartie_eps): the guarded instantiation of the default depth
    else if (a->nt_eps > 0.0) hipLaunchKernelGGL((k_scan<false, SCAN_SMALL_LDS ? 4 : 2, false, false, true>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
    else if (pfc >= 4) hipLaunchKernelGGL((k_scan<false, 4>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
    else if (pfc >= 2) hipLaunchKernelGGL((k_scan<false, 2>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
    else hipLaunchKernelGGL((k_scan<false, 1>), dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
}
