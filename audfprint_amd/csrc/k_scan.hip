// k_scan.hip -- K2 (per-unit statistics, floor correction) and K3 (the sequential
// decaying-threshold peak picker) for gfx950.  COMPILED WITH -ffp-contract=off: the HPF and
// threshold recurrences must round exactly like the reference's separate numpy operations.
//
// K3 replaces, per unit: the floor/mean of find_peaks (audfprint_analyze.py:285-286), the
// lfilter HPF (:293-295), _decaying_threshold_fwd_prune (:199-231) and
// _decaying_threshold_bwd_prune_peaks (:233-253).  Frame t depends on frame t-1 (the
// threshold vector), so time is sequential; parallelism is one WAVEFRONT PER UNIT with the
// 256 bins spread 4-per-lane: lane L owns bins 4L..4L+3, the threshold and HPF state live in
// VGPRs for the whole clip, local maxima need one neighbour shuffle per side, the per-frame
// top-K is K rounds of wavefront arg-max (ballot picks ties towards the larger bin exactly
// like sorted(zip(val, bin), reverse=True), :220), and the Gaussian bumps come from an LDS
// copy of the host-computed table (bits equal to the reference's __sp_vals, :191-192).
#include <hip/hip_runtime.h>
#include <math.h>
#include "afp_common.h"


__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
    int lo = __shfl_xor(__double2loint(v), mask);
    int hi = __shfl_xor(__double2hiint(v), mask);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up_d(double v, int k)
{
    int lo = __shfl_up(__double2loint(v), k);
    int hi = __shfl_up(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_down_d(double v, int k)
{
    int lo = __shfl_down(__double2loint(v), k);
    int hi = __shfl_down(__double2hiint(v), k);
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
__global__ __launch_bounds__(AFP_WAVE)
void k_unit_stats(StatsArgs A)
{
    const int u = blockIdx.x;
    const int lane = threadIdx.x;
    const int T = A.unit_T[u];
    UnitStats st;
    st.logfloor = 0.0; st.lsum = 0.0; st.pmax = 0.0; st.flags = 0; st.pad = 0;
    if (T <= 0) {
        st.flags = UNIT_EMPTY;
        if (lane == 0) A.stats[u] = st;
        return;
    }
    const int64_t b0 = A.unit_bbase[u], b1 = A.unit_bbase[u + 1];
    double pmax = 0.0, lmin = INFINITY, lsum = 0.0;
    for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) {
        pmax = fmax(pmax, A.blk_pmax[b]);
        lmin = fmin(lmin, A.blk_lmin[b]);
        lsum += A.blk_lsum[b];
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        pmax = fmax(pmax, shfl_xor_d(pmax, s));
        lmin = fmin(lmin, shfl_xor_d(lmin, s));
        lsum += shfl_xor_d(lsum, s);
    }
    st.pmax = pmax;
    st.lsum = lsum;
    if (!(pmax > 0.0)) {
        st.flags = UNIT_ZERO;                         // identically-zero input (audfprint_analyze.py:287-290)
    } else {
        st.logfloor = log(sqrt(pmax) / 1e6);          // log(max|S| / 1e6), :285
        if (lmin < st.logfloor) st.flags |= UNIT_CORR;
    }
    if (lane == 0) A.stats[u] = st;
}

// ------------------------------------------------------------------------------------------
// K2b: where some log|S| fell under the floor, sum (floor - value) so that
//      mean(max(log|S|, floor)) = (lsum + corr) / (257 T).   One workgroup per STFT chunk;
//      chunks that never went under the floor leave at once.
__global__ __launch_bounds__(256)
void k_floor_corr(CorrArgs A)
{
    __shared__ double red[4];
    const int blk = blockIdx.x;
    const int u = A.blk_unit[blk];
    const UnitStats st = A.stats[u];
    if (!(st.flags & UNIT_CORR) || !(A.blk_lmin[blk] < st.logfloor)) {
        if (threadIdx.x == 0) A.blk_corr[blk] = 0.0;
        return;
    }
    const int T = A.unit_T[u];
    const int t0 = A.blk_t0[blk];
    const int nt = min(STFT_FPB, T - t0);
    const int64_t fb = A.unit_fbase[u] + t0;
    double acc = 0.0;
    const double lf = st.logfloor;
    for (int i = threadIdx.x; i < nt * 257; i += 256) {
        int t = i / 257, b = i - t * 257;
        double v = (b < 256) ? A.logS[(fb + t) * AFP_NBINS + b] : A.nyq[fb + t];
        if (v < lf) acc += (v > -INFINITY) ? (lf - v) : lf;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += shfl_xor_d(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) A.blk_corr[blk] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ------------------------------------------------------------------------------------------
// K3 helpers.  thr[j] / y[j] belong to bin 4*lane + j.

// sthresh = max(sthresh, val * G[. - bin])  (audfprint_analyze.py:194-196, 226-228)
__device__ __forceinline__ void bump(double (&thr)[4], double val, int bin, int lane, const double* Gs)
{
    const double* g = Gs + (255 + 4 * lane - bin);
#pragma unroll
    for (int j = 0; j < 4; j++) thr[j] = fmax(thr[j], val * g[j]);
}

// locmax (audfprint_analyze.py:36-52): >= on the left, strict on the right, ends allowed.
__device__ __forceinline__ void locmax4(const double (&y)[4], int lane, bool (&lm)[4])
{
    double left = dpp_d<DPP_WAVE_SHR1>(y[3], y[3]);     // bin 4L-1 (lane 0 keeps its own; masked below)
    double right = dpp_d<DPP_WAVE_SHL1>(y[0], y[0]);    // bin 4L+4 (lane 63 likewise)
    lm[0] = (lane == 0 || y[0] >= left) && (y[1] < y[0]);
    lm[1] = (y[1] >= y[0]) && (y[2] < y[1]);
    lm[2] = (y[2] >= y[1]) && (y[3] < y[2]);
    lm[3] = (y[3] >= y[2]) && (lane == 63 || right < y[3]);
}

// spreadpeaksinvector (audfprint_analyze.py:153-160): start from zeros, spread every local max.
__device__ __forceinline__ void spread_all(double (&thr)[4], const double (&v)[4], int lane, const double* Gs)
{
    bool lm[4];
    locmax4(v, lane, lm);
#pragma unroll
    for (int j = 0; j < 4; j++) thr[j] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        unsigned long long m = __ballot(lm[j]);
        while (m) {
            int wl = __ffsll((long long)m) - 1;
            m &= m - 1;
            double val = readlane_d(v[j], wl);
            bump(thr, val, 4 * wl + j, lane, Gs);
        }
    }
}

struct __attribute__((aligned(16))) dpair { double a, b; };

// floor + mean (audfprint_analyze.py:285-286) then one step of lfilter([1,-1],[1,-pole]) in
// direct form II transposed (:293-294):  y = x + z ;  z = -x + pole*y
__device__ __forceinline__ void hpf_step(const double (&raw)[4], double lf, double mean, double pole,
                                         double (&z)[4], double (&y)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        double x = fmax(raw[j], lf) - mean;
        double yy = x + z[j];
        z[j] = (-x) + pole * yy;
        y[j] = yy;
    }
}

// ---- K3 ------------------------------------------------------------------------------------
// Workgroup = 2 wavefronts per unit.  Wave 1 (the LOADER) streams the unit's spectrogram
// (forward pass) or candidate records (backward pass) from HBM into a double-buffered LDS ring,
// CF frames at a time; wave 0 (the SCANNER) runs the sequential recurrence out of LDS and only
// ever issues global STORES, so it never waits on vmcnt: HBM latency is fully hidden behind a
// CF-frame chunk of scanning, and the two roles meet at one s_barrier per chunk.
#define CF 4                                   // frames per chunk (small ring: leaves LDS for a co-resident k_stft)
#define FROW 256                               // doubles per frame row in the ring

__device__ __forceinline__ void loader_fill_frames(const double* __restrict__ L, int64_t fb, int T, int chunk,
                                                   double* dst, int lane)
{
    dpair q[CF][2];
#pragma unroll
    for (int i = 0; i < CF; i++) {
        int t = chunk * CF + i;
        if (t > T - 1) t = T - 1;                                  // clamped: always issued
        const dpair* p = reinterpret_cast<const dpair*>(L + (fb + t) * AFP_NBINS + 4 * lane);
        q[i][0] = p[0]; q[i][1] = p[1];
    }
#pragma unroll
    for (int i = 0; i < CF; i++) {
        dpair* o = reinterpret_cast<dpair*>(dst + i * FROW + 4 * lane);
        o[0] = q[i][0]; o[1] = q[i][1];
    }
}

__device__ __forceinline__ void read_frame(const double* src, int lane, double (&x)[4])
{
    const dpair* p = reinterpret_cast<const dpair*>(src + 4 * lane);
    dpair q0 = p[0], q1 = p[1];
    x[0] = q0.a; x[1] = q0.b; x[2] = q1.a; x[3] = q1.b;
}

__global__ __launch_bounds__(2 * AFP_WAVE)
void k_scan(ScanArgs A)
{
    __shared__ double Gs[512];
    __shared__ __attribute__((aligned(16))) double fbuf[2][CF * FROW];          // 32 KiB ring
    const int u = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const bool scanner = threadIdx.x < AFP_WAVE;
    const int T = A.unit_T[u];
    if (T <= 0) return;
    const int64_t fb = A.unit_fbase[u];
    const int K = A.K;
    const UnitStats st = A.stats[u];

    if (st.flags & UNIT_ZERO) {
        // all-zero spectrogram: HPF of zeros is zero, nothing exceeds the (zero) threshold
        // (masks / pcnt are pre-zeroed by the host before this launch)
        if (threadIdx.x == 0) A.unit_mean[u] = 0.0;
        return;
    }

    for (int i = threadIdx.x; i < 512; i += 2 * AFP_WAVE) { int dd = i - 255; Gs[i] = (i < 511) ? A.gauss[dd < 0 ? -dd : dd] : 0.0; }

    const int nch = (T + CF - 1) / CF;
    const double* __restrict__ L = A.logS;
    // candidate ring for the backward pass re-uses the frame ring's LDS
    double* cvring = &fbuf[0][0];                                   // [2][CF*K] doubles
    int* cbring = reinterpret_cast<int*>(&fbuf[1][0]);              // [2][CF*K] ints
    const int CK = CF * K;

    if (!scanner) {
        // =========================== LOADER wavefront ===========================
        loader_fill_frames(L, fb, T, 0, fbuf[0], lane);
        loader_fill_frames(L, fb, T, 1, fbuf[1], lane);
        __syncthreads();                                            // (B0) chunks 0,1 + Gs ready
        for (int c = 0; c < nch; c++) {
            if (c >= 1 && c + 1 < nch) loader_fill_frames(L, fb, T, c + 1, fbuf[(c + 1) & 1], lane);
            __syncthreads();                                        // (Bf) end of forward chunk c
        }
        // backward: records of chunk c -> ring[c & 1]
        {
            const int c = nch - 1;
            const int n = (min(T, (c + 1) * CF) - c * CF) * K;
            for (int i = lane; i < n; i += AFP_WAVE) {
                cvring[(c & 1) * CK + i] = A.cand_val[(fb + (int64_t)c * CF) * K + i];
                cbring[(c & 1) * CK + i] = A.cand_bin[(fb + (int64_t)c * CF) * K + i];
            }
        }
        __syncthreads();                                            // (B1) first backward chunk ready
        for (int c = nch - 1; c >= 0; c--) {
            if (c >= 1) {
                const int cn = c - 1;
                for (int i = lane; i < CK; i += AFP_WAVE) {
                    cvring[(cn & 1) * CK + i] = A.cand_val[(fb + (int64_t)cn * CF) * K + i];
                    cbring[(cn & 1) * CK + i] = A.cand_bin[(fb + (int64_t)cn * CF) * K + i];
                }
            }
            __syncthreads();                                        // (Bb) end of backward chunk c
        }
        return;
    }

    // =========================== SCANNER wavefront ===========================
    // mean of the floored log-spectrogram over all 257 x T entries
    double corr = 0.0;
    if (st.flags & UNIT_CORR) {
        const int64_t b0 = A.unit_bbase[u], b1 = A.unit_bbase[u + 1];
        for (int64_t b = b0 + lane; b < b1; b += AFP_WAVE) corr += A.blk_corr[b];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) corr += shfl_xor_d(corr, s);
    }
    const double mean = (st.lsum + corr) / (257.0 * (double)T);
    const double lf = st.logfloor;
    const double pole = A.pole;
    const double a_dec = A.a_dec;
    if (lane == 0) A.unit_mean[u] = mean;

    double thr[4], z[4], ylast[4];
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, fwd_wait = 0, bwd_wait = 0;
    if (A.prof) tk0 = __builtin_readcyclecounter();
    __syncthreads();                                                // (B0)
    if (A.prof) tk1 = __builtin_readcyclecounter();

    // ---- initial forward threshold: spread the per-bin max over the first min(10,T) HPF'd columns (:204-206)
    {
        double vmax[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { z[j] = 0.0; vmax[j] = -INFINITY; y[j] = 0.0; ylast[j] = 0.0; }
        const int n0 = T < 10 ? T : 10;
        // the first 10 columns come straight from HBM (once per unit; the ring holds only 2*CF frames)
        dpair pre[10][2];
#pragma unroll
        for (int t = 0; t < 10; t++) {
            const dpair* p = reinterpret_cast<const dpair*>(L + (fb + (t < T ? t : T - 1)) * AFP_NBINS + 4 * lane);
            pre[t][0] = p[0]; pre[t][1] = p[1];
        }
#pragma unroll
        for (int t = 0; t < 10; t++) {
            if (t < n0) {
                double raw[4] = {pre[t][0].a, pre[t][0].b, pre[t][1].a, pre[t][1].b};
                hpf_step(raw, lf, mean, pole, z, y);
#pragma unroll
                for (int j = 0; j < 4; j++) vmax[j] = fmax(vmax[j], y[j]);
            }
        }
        spread_all(thr, vmax, lane, Gs);
    }

    // ---- forward pass (:214-230).  Per chunk: phase A computes the HPF'd columns and the
    //      local-max flags of all CF frames (independent of the threshold: plenty of ILP),
    //      phase B runs the threshold recurrence, which for a frame without candidates is
    //      just 4 compares + a ballot + the decay multiply.
    if (A.prof) tk2 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < 4; j++) z[j] = 0.0;
    for (int c = 0; c < nch; c++) {
        const double* buf = fbuf[c & 1];
        double yy[CF][4];
        unsigned lmb[CF];
#pragma unroll
        for (int i = 0; i < CF; i++) {
            const int t = c * CF + i;
            double raw[4];
            read_frame(buf + i * FROW, lane, raw);
            hpf_step(raw, lf, mean, pole, z, yy[i]);
            bool lm[4];
            locmax4(yy[i], lane, lm);
            lmb[i] = (t < T) ? ((lm[0] ? 1u : 0u) | (lm[1] ? 2u : 0u) | (lm[2] ? 4u : 0u) | (lm[3] ? 8u : 0u)) : 0u;
            if (t == T - 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) ylast[j] = yy[i][j];
            }
            if (A.sgram_dbg && t < T) {
                double* o = A.sgram_dbg + (fb + t) * AFP_NBINS + 4 * lane;
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = yy[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < CF; i++) {
            unsigned cm = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (((lmb[i] >> j) & 1u) && (yy[i][j] > thr[j])) cm |= 1u << j;   // strict >, :217
            unsigned long long anym = __ballot(cm != 0);
            if (anym != 0ull) {
                const int t = c * CF + i;
                int cnt = 0;
                double ev = 0.0;
                int eb = -1;
                while (anym != 0ull && cnt < K) {
                    // lane-local best of the remaining candidates (ties -> larger bin)
                    double bv = -1.0;
                    int bs = -1;
#pragma unroll
                    for (int j = 0; j < 4; j++) if ((cm >> j) & 1u) { if (yy[i][j] >= bv) { bv = yy[i][j]; bs = j; } }
                    int wl;
                    if ((anym & (anym - 1)) == 0ull) {
                        wl = __ffsll((long long)anym) - 1;                 // a single candidate lane: no reduction
                    } else {
                        const double wv = wave_max_uniform(bv);
                        const unsigned long long wm = __ballot(bs >= 0 && bv == wv);
                        if (wm == 0ull) break;                             // only reachable with NaN input
                        wl = 63 - __clzll((long long)wm);                  // highest lane = larger bin
                    }
                    const int ws = __builtin_amdgcn_readlane(bs, wl);
                    const double val = readlane_d(bv, wl);
                    const int bin = 4 * wl + ws;
                    if (lane == wl) cm &= ~(1u << ws);
                    bump(thr, val, bin, lane, Gs);                         // :226-228
                    if (lane == cnt) { ev = val; eb = bin; }
                    cnt++;
                    anym = __ballot(cm != 0);
                }
                // candidate records: cand_bin was pre-filled with -1, only survivors are written
                if (lane < cnt) {
                    A.cand_val[(fb + t) * K + lane] = ev;
                    A.cand_bin[(fb + t) * K + lane] = eb;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) thr[j] = thr[j] * a_dec;          // :230
        }
        if (c == nch - 1) __threadfence();      // candidate records must be visible to the loader wave
        unsigned long long tw0 = 0;
        if (A.prof) tw0 = __builtin_readcyclecounter();
        __syncthreads();                                            // (Bf)
        if (A.prof) fwd_wait += __builtin_readcyclecounter() - tw0;
    }

    // ---- backward pass (:233-253)
    if (A.prof) tk3 = __builtin_readcyclecounter();
    spread_all(thr, ylast, lane, Gs);                                     // :237
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;                    // pending mask of frame t+1
    __syncthreads();                                                // (B1)
    if (A.prof) tk4 = __builtin_readcyclecounter();
    const bool packed = CK <= AFP_WAVE;           // the whole chunk's records fit one register per lane
    for (int c = nch - 1; c >= 0; c--) {
        const double* cv = cvring + (c & 1) * CK;
        const int* cb = cbring + (c & 1) * CK;
        double evc = 0.0;
        int ebc = -1;
        unsigned long long mvalid = 0ull;
        if (packed) {
            if (lane < CK) { evc = cv[lane]; ebc = cb[lane]; }
            mvalid = __ballot(ebc >= 0);
        }
#pragma unroll
        for (int i = CF - 1; i >= 0; i--) {
            const int t = c * CF + i;
            if (t < T) {
                double ev = evc;
                int eb = ebc;
                int cnt, base;
                if (packed) {
                    base = i * K;
                    cnt = __popcll((mvalid >> base) & ((1ull << K) - 1ull));
                } else {
                    ev = 0.0; eb = -1; base = 0;
                    if (lane < K) { ev = cv[i * K + lane]; eb = cb[i * K + lane]; }
                    cnt = __popcll(__ballot(eb >= 0));
                }
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                for (int r = 0; r < cnt; r++) {
                    const double val = readlane_d(ev, base + r);
                    const int bin = __builtin_amdgcn_readlane(eb, base + r);
                    const int sub = bin & 3;
                    const double tsel = sub == 0 ? thr[0] : sub == 1 ? thr[1] : sub == 2 ? thr[2] : thr[3];
                    const double tb_ = readlane_d(tsel, bin >> 2);
                    if (val >= tb_) {                                      // :242  (>=)
                        bump(thr, val, bin, lane, Gs);                     // :244
                        const unsigned long long bit = 1ull << (bin & 63);
                        const int q = bin >> 6;
                        if (q == 0) { c0 |= bit; p0 &= ~bit; }             // keep; :247-248 clears (bin, t+1)
                        else if (q == 1) { c1 |= bit; p1 &= ~bit; }
                        else if (q == 2) { c2 |= bit; p2 &= ~bit; }
                        else { c3 |= bit; p3 &= ~bit; }
                    }                                                      // else :251 drops (bin, t)
                }
                // masks / pcnt were pre-zeroed: only non-empty frames are written
                if ((p0 | p1 | p2 | p3) != 0ull) {
                    const unsigned long long w = lane == 0 ? p0 : lane == 1 ? p1 : lane == 2 ? p2 : p3;
                    if (lane < 4) A.masks[(fb + t + 1) * 4 + lane] = w;
                    if (lane == 4) A.pcnt[fb + t + 1] = __popcll(p0) + __popcll(p1) + __popcll(p2) + __popcll(p3);
                }
                p0 = c0; p1 = c1; p2 = c2; p3 = c3;
#pragma unroll
                for (int j = 0; j < 4; j++) thr[j] = a_dec * thr[j];      // :252
            }
        }
        unsigned long long tw0 = 0;
        if (A.prof) tw0 = __builtin_readcyclecounter();
        __syncthreads();                                            // (Bb)
        if (A.prof) bwd_wait += __builtin_readcyclecounter() - tw0;
    }
    if ((p0 | p1 | p2 | p3) != 0ull) {
        const unsigned long long w = lane == 0 ? p0 : lane == 1 ? p1 : lane == 2 ? p2 : p3;
        if (lane < 4) A.masks[fb * 4 + lane] = w;
        if (lane == 4) A.pcnt[fb] = __popcll(p0) + __popcll(p1) + __popcll(p2) + __popcll(p3);
    }
    if (A.prof && lane == 0) {
        unsigned long long* o = A.prof + (size_t)u * 8;
        o[0] = tk0; o[1] = tk1; o[2] = tk2; o[3] = tk3; o[4] = tk4; o[5] = __builtin_readcyclecounter(); o[6] = (unsigned long long)T; o[7] = (fwd_wait << 32) | (bwd_wait & 0xffffffffull);
    }
}

extern "C" void afp_launch_unit_stats(const StatsArgs* a, hipStream_t st)
{
    if (a->nunits > 0) hipLaunchKernelGGL(k_unit_stats, dim3(a->nunits), dim3(AFP_WAVE), 0, st, *a);
}
extern "C" void afp_launch_floor_corr(const CorrArgs* a, int nblk, hipStream_t st)
{
    if (nblk > 0) hipLaunchKernelGGL(k_floor_corr, dim3(nblk), dim3(256), 0, st, *a);
}
extern "C" void afp_launch_scan(const ScanArgs* a, int nunits, hipStream_t st)
{
    if (nunits > 0) hipLaunchKernelGGL(k_scan, dim3(nunits), dim3(2 * AFP_WAVE), 0, st, *a);
}
