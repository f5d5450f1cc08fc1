// k_stft.hip -- K1: framed PCM -> log|S| (float64) for gfx950.
//
// Replaces stft.stft (stft.py:62-94: reflect pad, 512/256 framing, window multiply, rfft) and
// the abs / log of Analyzer.find_peaks (audfprint_analyze.py:280,285).  The floor max/1e6 and
// the mean (:285-286) need per-unit reductions, so this kernel stores the un-floored log|S|
// and per-workgroup partials {max |S|^2, min log|S|, sum of finite log|S|}; k_unit_stats /
// k_floor_corr / k_scan finish the job.
//
// Mapping: one wavefront transforms two consecutive frames as one complex 512-point FFT
// (fft512_core.h).  Lane L loads samples L + 64 j (coalesced 256-B rows of float32); the
// output lane m owns bins m + 64 c and writes four coalesced 512-B rows of float64 per frame.
// float64 throughout: the reference promotes to float64 at the window multiply (stft.py:93).
//
// The two exchanges between the radix-8 passes go through LDS in two rounds -- the eight real parts, then the eight
// imaginary parts, through the same 4.6 KB per wavefront (a wavefront's DS instructions execute in order, so the
// second round can overwrite the buffer without a wait) -- which keeps the workgroup at 31 KB of LDS (8 KB of them the log table): four workgroups
// per CU by registers (<= 128 VGPRs) when the kernel runs alone, three beside the scan kernel's wavefronts
// (round 2: (re, im) pairs in one round needed 43 KB and capped a CU at three workgroups; k_stft 1.05 -> 0.96 ms on C3).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>
#include "afp_common.h"
#include "fft512_core.h"

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Source sample for padded position: generalised numpy 'reflect' (stft.py:87-88).
template <typename ST>
__device__ __forceinline__ ST fetch_sample(const ST* __restrict__ d, int64_t n, int64_t i)
{
    if (i < 0 || i >= n) {
        if (n == 1) {
            i = 0;
        } else {
            int64_t period = 2 * (n - 1);
            int64_t m = i % period;
            if (m < 0) m += period;
            i = (m >= n) ? period - m : m;
        }
    }
    return d[i];
}

__device__ __forceinline__ double shfl_d(double v, int src)
{
    int lo = __shfl(__double2loint(v), src);
    int hi = __shfl(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double shfl_xor_d(double v, int mask)
{
    int lo = __shfl_xor(__double2loint(v), mask);
    int hi = __shfl_xor(__double2hiint(v), mask);
    return __hiloint2double(hi, lo);
}

// 0.5 * ln(p) for p >= 0, table driven: p = 2^k m with m in [0.5, 1) (v_frexp_mant), AFP_LOGTAB_N = 512 intervals of
// width 2^-10 with centres c_i, h = (m / c_i - 1) / 2, |h| < 2^-11, degree-4 log1p: 7 double ops, 5 integer ops and one
// 16-byte LDS read instead of the ~70-instruction library log (with 128 intervals the polynomial needs degree 6: two
// more FMAs per value, k_stft 0.93 instead of 0.89 ms; the table costs 8 KB of LDS instead of 2).
// Absolute error of a few 1e-16 over the magnitudes that occur; the reference's own log differs from
// ours by the same order, far inside the 1e-4 float tolerance, and ties (equal inputs) stay ties.
// p == 0 gives a finite -354.9 (k = -1022, m = 0) rather than -inf: every consumer floors at
// log(max|S|/1e6) first.
struct __attribute__((aligned(16))) d2 { double x, y; };

// same as split_power (fft512_core.h) without the 1/4: the window was pre-scaled by 1/2
__device__ __forceinline__ void split_power_unscaled(double zr, double zi, double pr, double pi, double& pa, double& pb)
{
    const double ar = zr + pr, ai = zi - pi;
    const double br = zr - pr, bi = zi + pi;
    pa = ar * ar + ai * ai;
    pb = br * br + bi * bi;
}
__device__ __forceinline__ double half_log(double p, const d2* __restrict__ tab)
{
    const int hw = __double2hiint(p);
    const double m = __builtin_amdgcn_frexp_mant(p);
    const int k = (hw - (1022 << 20)) >> 20;              // exponent of m's scaling (sign bit is clear: p >= 0)
    const d2 e = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(tab) +
                                              ((hw >> (16 - AFP_LOGTAB_BITS)) & ((AFP_LOGTAB_N - 1) << 4)));
    const double h = fma(m, e.x, -0.5);                   // e = (0.5 / c_i, 0.5 ln c_i)
#if AFP_LOGTAB_BITS >= 9
    // |h| < 2^-11:  log1p(2h) / 2 = h + h^2 (-1 + h (4/3 - 2 h)) - (16/5) h^5 ..., the dropped term below 9e-17
    const double lp = fma(h * h, fma(h, fma(h, -2.0, 4.0 / 3.0), -1.0), h);
#else
    // |h| < 2^-9:   log1p(2h) / 2 = h + h^2 (-1 + 4/3 h + h^2 (-2 + 16/5 h - 16/3 h^2))
    const double h2 = h * h;
    const double qa = fma(h, 4.0 / 3.0, -1.0);
    const double qb = fma(h, 16.0 / 5.0, -2.0);
    const double q = fma(h2, fma(h2, -16.0 / 3.0, qb), qa);
    const double lp = fma(h2, q, h);
#endif
    return fma((double)k, 0.34657359027997264, e.y) + lp; // k ln2 / 2 + ln(c_i) / 2 + log1p(r) / 2
}

// the log-spectrogram is written once and read once by k_scan, far more than any cache holds: STFT_NT=1 marks the
// stores non-temporal (streaming)
#if defined(STFT_NT) && STFT_NT
#define STFT_STORE(P, V) __builtin_nontemporal_store((V), (P))
#else
#define STFT_STORE(P, V) (*(P) = (V))
#endif
// eight ds_read_b64 at base + j * STRIDE_BYTES, written out because the compiler pairs plain loads into ds_read2_b64,
// which the LDS serves at half the rate of ds_read_b64 (MI355X_MICROARCH.md, LDS table).  The compiler does not count
// these loads in lgkmcnt: the caller must pass the values through lds_wait8 before using them.
template <int STRIDE_BYTES>
__device__ __forceinline__ void lds_read8_b64(double (&x)[8], const double* base)
{
    const uint32_t a = (uint32_t)(uintptr_t)base;          // LDS byte address (low 32 bits of the generic-to-local pointer)
    asm volatile("ds_read_b64 %0, %8 offset:%9\n\tds_read_b64 %1, %8 offset:%10\n\tds_read_b64 %2, %8 offset:%11\n\tds_read_b64 %3, %8 offset:%12\n\t"
                 "ds_read_b64 %4, %8 offset:%13\n\tds_read_b64 %5, %8 offset:%14\n\tds_read_b64 %6, %8 offset:%15\n\tds_read_b64 %7, %8 offset:%16"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                 : "v"(a), "n"(0 * STRIDE_BYTES), "n"(1 * STRIDE_BYTES), "n"(2 * STRIDE_BYTES), "n"(3 * STRIDE_BYTES),
                   "n"(4 * STRIDE_BYTES), "n"(5 * STRIDE_BYTES), "n"(6 * STRIDE_BYTES), "n"(7 * STRIDE_BYTES)
                 : "memory");
}
__device__ __forceinline__ void lds_wait8(double (&x)[8], double (&y)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
                 :: "memory");
}
#ifndef STFT_VGPR_CAP
#define STFT_VGPR_CAP 128              // 3 x 128 + 2 x 64 (k_scan_small) = 512 registers per SIMD lane
#endif
#ifndef STFT_MINW
#define STFT_MINW 4                    // waves per SIMD the register allocation targets (4 x 128 VGPRs alone; 3 beside k_scan)
#endif
// ST = double (a float64 waveform handed to Analyzer.find_peaks: the reference keeps it in float64, stft.py:87-93),
// float (audio_read.buf_to_float output, audio_read.py:121-145) or int16_t (the raw s16le
// samples ffmpeg pipes, audio_read.py:196-203): x/32768 is exact in float32, so converting the
// integer straight to double and folding 2^-15 into the window scale gives bit-identical products.
//
// CMP = true, the COMPACT spectral stage (round 3): the float64 log-spectrogram never goes to HBM.  The workgroup also runs the
// onset filter lfilter([1,-1],[1,-pole]) (audfprint_analyze.py:293-295) over its frames and finds the local maxima along
// frequency (:36-52); a frame leaves as its 256-bit local-maximum mask + the filtered values of the maxima only.  Two
// things make that possible before the per-unit mean (:286) is known:
//   * the filter is linear, so HPF(L - mean)[n] = HPF(L)[n] - mean * pole^n: the mean only shifts every bin of a frame
//     by the same amount and cannot change which bins are local maxima; k_scan_c subtracts the term from the values;
//   * a unit none of whose values falls under the floor max|S|/1e6 (:285) needs no flooring; units that do (UNIT_CORR,
//     known only after the whole unit) are re-done by the dense kernels (k_stft<ST,false> + k_scan), see afp_abi.hip.
// The filter state crosses frames, so chunks are listed TIME-MAJOR and chunk k + 1 of a unit picks the state chunk k left
// in HBM (write-through stores, flag; the predecessor was dispatched a whole residency earlier, so the wait is nearly
// always over before it starts).  Inside the chunk the four wavefronts keep their frame pairs (wave w: frames
// t0 + 8 i + 2 w, +1 in iteration i) and exchange the filter state through LDS once per iteration: every wave filters
// its two frames from a zero state (exact: the state enters linearly), publishes the state it ends with, and folds the
// states of the waves before it:  z_in(w+1) = z_loc(w) + pole^2 z_in(w).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v)
{
    // (bound_ctrl: every lane of a rotation has a source, and the destination then needs no initialisation)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#define DPP_WAVE_ROL1 0x134
// LDS writes of this wave done, then the workgroup barrier -- without the vmcnt(0) a __syncthreads() carries (the PCM rows of
// the next pair and the compact stores of the last one stay in flight)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LIST (dense mode behind the compact stage): the workgroups stride over the chunks k_unit_stats listed
template <typename ST, bool CMP, bool LIST = false>
__global__ __launch_bounds__(STFT_WAVES * AFP_WAVE, STFT_MINW) __attribute__((amdgpu_num_vgpr(STFT_VGPR_CAP)))
void k_stft(StftArgs A)
{
    __shared__ d2 ltab[AFP_LOGTAB_N];
    // compact mode: filter states, [0] = the state before the iteration's first frame, [1 + w] = wave w's local end state
    // (w = 0..2); element [lane][c] belongs to bin lane + 64 c
    __shared__ d2 zx[CMP ? 4 : 1][2][AFP_WAVE];      // [..][h][lane] = bins lane + 64 (2h), lane + 64 (2h + 1): 16-byte lane stride
    __shared__ double wlds[AFP_NFFT];
    __shared__ double lds_c[STFT_WAVES][FFT_LDS_DOUBLES];      // real parts, then imaginary parts, through the same 4.6 KB
    __shared__ double red[3][STFT_WAVES];
    __shared__ double flat_s[STFT_WAVES];
    __shared__ int flat_f[STFT_WAVES][2];         // first / last frame whose non-zero samples share one parity (per wavefront)
#ifdef STFT_PAD_LDS
    __shared__ char pad_s[STFT_PAD_LDS];          // occupancy experiment: fewer workgroups per CU
    if (threadIdx.x == 0) pad_s[A.K & 15] = 1;
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: frame indices / LDS bases stay scalar
    // The arguments used once per chunk (hand-off, dense head / last rows, epilogue) are loaded from the kernel-argument
    // segment where they are needed -- through a pointer laundered at the point of use, so the loads cannot be hoisted --
    // instead of sitting in scalar registers through the whole pair loop (the kernel runs at the SGPR limit: every pointer
    // held there is a spill elsewhere).  StftArgs is the only kernel parameter: it starts at offset 0 of the segment.
    typedef const __attribute__((address_space(4))) char* kptr_t;
    const kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
#define KARG(T, field) ([&]() -> T { kptr_t kr_ = ka; asm volatile("" : "+s"(kr_)); \
                                     typedef T karg_t_; return *(karg_t_ const __attribute__((address_space(4)))*)(kr_ + offsetof(StftArgs, field)); }())
    if (CMP && KARG(int32_t*, list_zero) && blockIdx.x == 0 && threadIdx.x == 0) *KARG(int32_t*, list_zero) = 0;
    // dense mode behind the compact stage: the workgroups stride over the listed chunks (usually a handful, often none)
    static_assert(!(CMP && LIST), "the chunk list belongs to the dense mode");
    const int list_n = LIST ? *A.list_cnt : -1;
    for (int blk = blockIdx.x;; blk += gridDim.x) {
    if (LIST && blk >= list_n) break;
    const ChunkDesc ch = LIST ? A.list[blk] : A.blk[blk];
    const int u = ch.unit;
    const int t0 = ch.t0;
    const UnitDesc ud = A.units[u];
    const int T = ud.T;
    const int64_t n = ud.n;
    const ST* __restrict__ d = reinterpret_cast<const ST*>(A.pcm) + ud.pcm_off;
    const double wscale = sizeof(ST) == 2 ? 0.5 / 32768.0 : 0.5;
    const int64_t fb = ud.fbase;
    double* lc = lds_c[wave];
    unsigned long long flag_seen = 0, flag_want = 0;
    if (CMP && t0 > 0 && threadIdx.x == 0) {
        flag_want = (KARG(unsigned long long, epoch) << 32) | (unsigned long long)(unsigned)(t0 / STFT_FPB);
        flag_seen = __hip_atomic_load(&KARG(unsigned long long*, zflag)[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = threadIdx.x; i < AFP_LOGTAB_N; i += STFT_WAVES * AFP_WAVE) { ltab[i].x = A.tables[TAB_LOGTAB + 2 * i]; ltab[i].y = A.tables[TAB_LOGTAB + 2 * i + 1]; }
    if (threadIdx.x < STFT_WAVES) { flat_s[threadIdx.x] = 0.0; flat_f[threadIdx.x][0] = 0x7fffffff; flat_f[threadIdx.x][1] = -1; }
    // window taps, pre-scaled: x0.5 here == x0.25 on |.|^2 (exact power-of-two scaling)
    for (int i = threadIdx.x; i < AFP_NFFT; i += STFT_WAVES * AFP_WAVE) wlds[i] = wscale * A.tables[TAB_WINDOW + i];
    if (CMP && t0 > 0 && threadIdx.x == 0) {
        // The chunk before this one publishes the filter state it ends with; the first look at its flag was issued before the
        // table fills above.
        // FORWARD PROGRESS.  The chunk list is time-major, so a chunk's predecessor (same unit, the 64 frames before) has a
        // SMALLER blockIdx.  Assumption: every XCD dispatches ITS share of a 1-D grid (workgroup i belongs to XCD i mod 8) in
        // ascending blockIdx order -- the XCDs may drift apart.  Let m be the smallest unfinished blockIdx.  On m's XCD every
        // smaller workgroup has finished, so m was dispatched before any other unfinished workgroup of that XCD: either m is
        // resident, or nothing unfinished of that XCD is and m is the next to get one of its free slots.  m waits only for a
        // smaller blockIdx -- all finished -- so it runs to its end and publishes; by induction every chunk does, whatever
        // the occupancy and however far the XCDs drift.
        // HIP does not PROMISE that order.  If it ever failed -- a waiter resident, its predecessor not
        // dispatchable because waiters hold every slot -- the bound below ends the wait after ~0.3 s, `err` is raised, every
        // workgroup still runs to its end (nothing can hang), and finalize() re-runs the batch on the dense path, which has
        // no cross-workgroup dependency (afp_abi.hip; exercised by afp_set_compact_force_timeout).  A ticket drawn from an
        // atomic counter at workgroup start would need no assumption at all; measured r04: k_stft 1.00-1.04 -> 1.09 ms, C3
        // step 1.494 -> 1.531 ms (the atomic's round trip and the chunk descriptor behind it sit in front of the first PCM
        // load of every chunk) -- not adopted.
        const int lim = KARG(int32_t, spin_limit);
        int spins = 0;
        while (flag_seen != flag_want) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > lim) { *KARG(int32_t*, err) = 1; break; }
            flag_seen = __hip_atomic_load(&KARG(unsigned long long*, zflag)[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const double pole = A.pole, pole2 = A.pole * A.pole;

    // loop-invariant per-lane constants: only the twiddle GENERATORS stay resident -- W_512^L for pass 1 and
    // W_64^n0 for pass 2 -- and the other twiddles are formed by repeated complex multiplication in
    // each pair (<= 128 VGPRs: three of these wavefronts fit on a SIMD beside two k_scan wavefronts).  Reading the
    // twiddles from an LDS table instead (52 fewer FP64 instructions per pair, 15 more ds_read_b128) measured slower.
    double u1r, u1i, t2sr, t2si;
    {
        int e = fft_tw1_exp(lane, 1); u1r = A.tables[TAB_TWIDDLE + 2 * e]; u1i = A.tables[TAB_TWIDDLE + 2 * e + 1];
        e = fft_tw2_exp(lane, 1); t2sr = A.tables[TAB_TWIDDLE + 2 * e]; t2si = A.tables[TAB_TWIDDLE + 2 * e + 1];
    }
    double pmax = 0.0;
    double lmin = INFINITY;
    double lsum = 0.0;

    // Raw samples of one frame pair: the two frames overlap by half, so 12 rows of 64 samples
    // (row m = samples baseA + 64 m + lane) cover both.  The rows of pair p+1 are requested while
    // pair p is being transformed (software prefetch; the window taps come from LDS instead of
    // 16 resident VGPRs).
    // (raw s16 samples are held as the sign-extended 32-bit integers the load instruction delivers -- global_load_sshort --
    //  so the compiler converts each register ONCE, v_cvt_f64_i32, instead of re-extending a 16-bit value at every use)
    using HT = std::conditional_t<sizeof(ST) == 2, int32_t, ST>;
    HT f[12];
    auto load_pair = [&](int p) {
        const int tA = t0 + 2 * (wave + STFT_WAVES * p);
        if (p >= STFT_PAIRS_PER_WAVE || tA >= T) return;
        const int64_t baseA = (int64_t)256 * tA - 256;      // padded index of frame t, tap q is 256 t + q; source = that - 256
        if ((baseA >= 0) && (baseA + 768 <= n)) {
            if constexpr (sizeof(ST) == 2 || LIST) {
                // s16 (and the list variants): row pointer wave-uniform, lane offset a laundered non-negative int -- the loads take the scalar-base +
                // 32-bit-offset form and no loop-invariant 64-bit per-lane pointer (d + lane) is kept alive across the
                // transform: that pair of registers was this variant's spill.  (The float32 kernel has the room and is 3 %
                // faster with the hoisted pointer: A/B r06, k_stft 1.03 vs 1.06 ms.)
                const ST* __restrict__ rowp = d + baseA;
                unsigned lo = (unsigned)lane;
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int m = 0; m < 12; m++) f[m] = (HT)rowp[lo + 64u * m];
            } else {
#pragma unroll
                for (int m = 0; m < 12; m++) f[m] = (HT)d[baseA + lane + 64 * m];
            }
        } else {
            const bool haveB = tA + 1 < T;
#pragma unroll
            for (int m = 0; m < 12; m++) f[m] = (m < 8 || haveB) ? (HT)fetch_sample(d, n, baseA + lane + 64 * m) : (HT)0;
        }
    };
    // Degenerate-frame detector (AFP_UNIT_TIE): a frame ALL of whose non-zero samples sit at offsets of one parity (all even
    // or all odd -- a lone click is the smallest member of the class).  With x[q] != 0 only for q = r (mod 2) the transform
    // obeys S(k + 256) = (-1)^r S(k), and x being real gives |S(256 - k)| = |S(k)|: bins k and 256 - k are EQUAL in exact
    // arithmetic (for one sample, or samples 256 apart, whole runs of bins are).  Which of two equal bins wins a place among the
    // maxpksperframe largest (audfprint_analyze.py:217-229: sorted by value), and for a flat spectrum which bins are local
    // maxima at all (:36-52), is then decided by the FFT's rounding noise -- only numpy's own pocketfft reproduces the reference
    // there (tools/sparse_frame_jitter.py: the live reference changes 2-24 of ~30 peaks under a 1e-15 relative jitter of its
    // own rfft output for every such frame class, and none for any frame holding both parities).  The hop is even, so the
    // parity of a sample's offset in its frame is the parity of its index in the clip, reflect padding included.
    // Frame A = rows 0..7 of the pair's 12 sample rows, frame B = rows 4..11; lane L holds offsets L + 64 m: the lane's parity is
    // the sample's.  It looks at the rows `f` holds, i.e. it runs for pair p once load_pair(p) has been issued and pair p - 1 is done.
    auto check_pair = [&](int p) {
        const int tA = t0 + 2 * (wave + STFT_WAVES * p);
        if (p >= STFT_PAIRS_PER_WAVE || tA >= T) return;
        const bool haveB = tA + 1 < T;
        auto nzbits = [](HT v) -> uint32_t {
            if constexpr (sizeof(ST) == 2) return (uint32_t)v;
            else if constexpr (sizeof(ST) == 4) return __float_as_uint((float)v) << 1;          // (-0.0 is zero)
            else { const double dv = (double)v; return ((uint32_t)__double2hiint(dv) << 1) | (uint32_t)__double2loint(dv); }
        };
        const unsigned long long EV = 0x5555555555555555ull;
        // quick reject: row 4 (offsets 256..319 of frame A = 0..63 of frame B) belongs to BOTH frames; non-zero samples of
        // both parities there rule out both (true for anything but near-silence): one compare + three scalar operations per pair
        {
            const unsigned long long b4 = __ballot(nzbits(f[4]) != 0u);
            if ((b4 & EV) != 0ull && (b4 & ~EV) != 0ull) return;
        }
        // (a value the compiler cannot see through, born BEHIND the branch: without it the twelve compares below are
        //  speculated above the quick reject and run for every pair -- 11 vector and 24 scalar instructions of 613 / pair)
        uint32_t behind = 0u;
        asm volatile("" : "+v"(behind));
        // occupied offsets per frame as 64-bit lane masks on the SCALAR unit (one ballot per row of 64 samples): no vector
        // register is spent on it, and scalar instructions issue beside the other wavefronts' FP64 work
        unsigned long long occA = 0ull, occB = 0ull;
#pragma unroll
        for (int m = 0; m < 12; m++) {
            const unsigned long long b = __ballot((nzbits(f[m]) | behind) != 0u);
            if (m < 8) occA |= b;
            if (m >= 4) occB |= b;
        }
        const bool degA = occA != 0ull && ((occA & EV) == 0ull || (occA & ~EV) == 0ull);
        const bool degB = haveB && occB != 0ull && ((occB & EV) == 0ull || (occB & ~EV) == 0ull);
        if (degA || degB) {
            // rare.  Level of the frame: sum |x[q]| w[q] >= |S(k)| for every k (for a lone click it IS |S(k)|, in every bin);
            // the largest such level of the chunk goes to k_unit_stats, which compares it with the unit's floor max|S| / 1e6
            // (a frame wholly under the floor is floored to a plateau of exactly equal values, reproduced bit for bit)
            double vA = 0.0, vB = 0.0;
#pragma unroll
            for (int m = 0; m < 12; m++) {
                const double a = fabs((double)f[m]);
                if (m < 8 && degA) vA += a * wlds[lane + 64 * m];
                if (m >= 4 && degB) vB += a * wlds[lane + 64 * (m - 4)];
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) { vA += shfl_xor_d(vA, sft); vB += shfl_xor_d(vB, sft); }
            if (lane == 0) {
                flat_s[wave] = fmax(flat_s[wave], 2.0 * fmax(vA, vB));          // (the window taps carry a factor 1/2)
                flat_f[wave][0] = min(flat_f[wave][0], degA ? tA : tA + 1);
                flat_f[wave][1] = max(flat_f[wave][1], degB ? tA + 1 : tA);
            }
        }
    };
    // compact mode: one onset-filtered frame (y[c] = bin lane + 64 c) -> local-maximum mask + the values of the maxima
    auto emit_frame = [&](int t, const double (&y)[4], int ln) {
        // locmax (audfprint_analyze.py:36-52): bin i is a maximum iff (i == 0 or y[i] >= y[i-1]) and (i == 255 or y[i+1] < y[i]).
        // R = "right neighbour is strictly smaller", as a 256-bit scalar mask; the left test is its complement one bin up
        // (values are finite), so only the right neighbours travel: one wave rotate per register, and bin 64 c + 63 takes
        // its neighbour from lane 0 of register c + 1 (= lane 63 of that register's rotation).
        double r[4];
#pragma unroll
        for (int c = 0; c < 4; c++) r[c] = dpp_mov_d<DPP_WAVE_ROL1>(y[c]);
        const unsigned long long B63 = 1ull << 63;
        unsigned long long R[4];
#pragma unroll
        for (int c = 0; c < 4; c++) R[c] = __ballot(r[c] < y[c]);
#pragma unroll
        for (int c = 0; c < 3; c++) R[c] = (R[c] & ~B63) | (__ballot(r[c + 1] < y[c]) & B63);
        R[3] |= B63;                                            // bin 255 has no right neighbour
        unsigned long long M[4];
        M[0] = (~(R[0] << 1) | 1ull) & R[0];                    // bin 0 has no left neighbour
#pragma unroll
        for (int c = 1; c < 4; c++) M[c] = ~((R[c] << 1) | (R[c - 1] >> 63)) & R[c];
        double* cv = A.cvals + (fb + t) * CV_ROW;
        int base = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(M[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)M[c], 0u));
            if (__builtin_amdgcn_inverse_ballot_w64(M[c])) cv[idx] = y[c];
            base += __popcll(M[c]);
        }
        // the mask lives in scalar registers: two scalar stores (gfx9 SMEM writes; the scalar cache is written back once,
        // at the end of the kernel) instead of eight lane moves and a vector store
        {
            // (the 128-bit store operands are staged in fixed registers: under this kernel's scalar-register pressure the
            //  compiler would otherwise hand the asm a vector register for an "s" operand)
            const uint64_t* mp = A.lmask + (fb + t) * 4;
            asm volatile("s_mov_b64 s[96:97], %0\n\ts_mov_b64 s[98:99], %1\n\ts_store_dwordx4 s[96:99], %4, 0x0\n\t"
                         "s_mov_b64 s[96:97], %2\n\ts_mov_b64 s[98:99], %3\n\ts_store_dwordx4 s[96:99], %4, 0x10"
                         :: "s"(M[0]), "s"(M[1]), "s"(M[2]), "s"(M[3]), "s"(mp) : "memory", "s96", "s97", "s98", "s99");
        }
        (void)ln;
        if (t < CV_HEAD) {                                      // dense rows the initial threshold is built from (:204-206)
            asm volatile("" ::: "memory");
            double* hd = KARG(double*, head) + ((int64_t)u * CV_HEAD + t) * AFP_NBINS;
#pragma unroll
            for (int c = 0; c < 4; c++) hd[ln + 64 * c] = y[c];
        }
        if (t == T - 1) {                                       // dense last row: seeds the backward pass (:237)
            asm volatile("" ::: "memory");
            double* yl = KARG(double*, ylast) + (int64_t)u * AFP_NBINS;
#pragma unroll
            for (int c = 0; c < 4; c++) yl[ln + 64 * c] = y[c];
        }
    };
    // A float64 waveform's twelve rows are 24 registers: prefetching them across the transform does not fit the 128-register
    // budget of four waves per SIMD (it spilled 40-56 bytes per lane), so that variant -- a float64 array handed to
    // Analyzer.find_peaks, never a bulk ingest -- loads each pair where it is used and lets the other waves cover the latency.
    // The LIST variant (the handful of chunks a compact batch's floored units are re-done from) carries the chunk loop's state
    // on top and spilled 28-40 bytes with the prefetch: it does without as well.
    constexpr bool PREFETCH = sizeof(ST) <= 4 && !LIST;
    if (PREFETCH) load_pair(0);
    if constexpr (CMP) {
        // state before the chunk's first frame: zero at the start of the unit (lfilter's zero initial state, :293); loaded
        // behind the first pair's sample rows, so the two latencies overlap
        double z0 = 0.0;
        if (t0 > 0) z0 = __hip_atomic_load(&KARG(double*, zcarry)[(int64_t)u * AFP_NBINS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<double*>(&zx[0][threadIdx.x >> 7][threadIdx.x & 63])[(threadIdx.x >> 6) & 1] = z0;      // bin = lane + 64 c (visible after the first lds_barrier)
    }
    if (PREFETCH) check_pair(0);

    for (int p = 0; p < STFT_PAIRS_PER_WAVE; p++) {
        const int tA = t0 + 2 * (wave + STFT_WAVES * p);
        const int tB = tA + 1;
        if (CMP) { if (t0 + 2 * STFT_WAVES * p >= T) break; }      // workgroup-uniform: every wave takes part in the state exchange
        else if (tA >= T) break;                 // wave-uniform
        const bool valid = !CMP || tA < T;       // (CMP, last chunk: this wave may have no frame left in the iteration)
        const bool haveB = tB < T;
        double LA[4], LB[4];                     // CMP: log|S| of the wave's two frames, bins lane + 64 c
        if (CMP && (!valid || !haveB)) {         // (rare: the unit's last pair)
#pragma unroll
            for (int c = 0; c < 4; c++) { LA[c] = 0.0; LB[c] = 0.0; }
        }
        if (valid) {
        if (!PREFETCH) { load_pair(p); check_pair(p); }
        double xr[8], xi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double w = wlds[lane + 64 * j];
            xr[j] = (double)f[j] * w;
            xi[j] = (double)f[j + 4] * w;
        }
        if (!haveB) {                            // wave-uniform, last frame of an odd-length unit only
            asm volatile("" ::: "memory");       // (keeps this a branch instead of 16 selects per pair)
#pragma unroll
            for (int j = 0; j < 8; j++) xi[j] = 0.0;
        }
        if (PREFETCH) load_pair(p + 1);
        // pass 1 + twiddle W_512^(L a)  (fft512_core.h: pass 2's b-independent factor is applied here)
        dft8(xr, xi);
        asm volatile("" : "+v"(u1r), "+v"(u1i), "+v"(t2sr), "+v"(t2si));   // (no hoisting of the powers)
        {
            double wr = u1r, wi = u1i;
#pragma unroll
            for (int a = 1; a < 8; a++) {
                cmul(xr[a], xi[a], wr, wi);
                if (a < 7) cmul(wr, wi, u1r, u1i);
            }
        }
#pragma unroll
        for (int a = 0; a < 8; a++) lc[fft_x1_waddr(lane, a)] = xr[a];
        wave_lds_fence();
        lds_read8_b64<64>(xr, &lc[fft_x1_raddr(lane, 0)]);
        wave_lds_fence();
#pragma unroll
        for (int a = 0; a < 8; a++) lc[fft_x1_waddr(lane, a)] = xi[a];
        wave_lds_fence();
        lds_read8_b64<64>(xi, &lc[fft_x1_raddr(lane, 0)]);
        lds_wait8(xr, xi);
        wave_lds_fence();
        // pass 2 + twiddle W_64^(n0 b), b = 1..7
        dft8(xr, xi);
        {
            double wr = t2sr, wi = t2si;
#pragma unroll
            for (int b = 1; b < 8; b++) {
                cmul(xr[b], xi[b], wr, wi);
                if (b < 7) cmul(wr, wi, t2sr, t2si);
            }
        }
#pragma unroll
        for (int b = 0; b < 8; b++) lc[fft_x2_waddr(lane, b)] = xr[b];
        wave_lds_fence();
        lds_read8_b64<8 * FFT_X2_STRIDE>(xr, &lc[fft_x2_raddr(lane, 0)]);
        wave_lds_fence();
#pragma unroll
        for (int b = 0; b < 8; b++) lc[fft_x2_waddr(lane, b)] = xi[b];
        wave_lds_fence();
        lds_read8_b64<8 * FFT_X2_STRIDE>(xi, &lc[fft_x2_raddr(lane, 0)]);
        lds_wait8(xr, xi);
        wave_lds_fence();
        // pass 3: lane m now holds Z[m + 64 c] in register c
        dft8(xr, xi);
        // Nyquist bin 256 = Z[256] (lane 0, register 4, self-paired): parked in the 8 exchange-buffer elements
        // no FFT pass touches (568..575), one per pair, and finished in one vector pass after the loop
        if (lane == 0) { lc[FFT_LDS_DOUBLES - 2 * STFT_PAIRS_PER_WAVE + 2 * p] = xr[4]; lc[FFT_LDS_DOUBLES - 2 * STFT_PAIRS_PER_WAVE + 2 * p + 1] = xi[4]; }
        // partner lane (64 - m) & 63 holds Z[512 - (m + 64 c)] in register 7 - c -- except lane 0, which pairs with ITSELF:
        // Z[64 c] with Z[64 (8 - c)], register 8 - c (and Z[0] with Z[0]).  Its registers 4..7 are read by no other lane, so
        // lane 0 alone shifts them down by one and puts Z[0] in register 7 (eight v_mov_b64 under exec = 1), and every lane
        // then takes register 7 - c of its partner: no per-lane select on the sixteen shuffled words.
        {
            unsigned long long ex_;
            asm volatile("s_mov_b64 %8, exec\n\ts_mov_b64 exec, 1\n\t"
                         "v_mov_b64 %0, %1\n\tv_mov_b64 %1, %2\n\tv_mov_b64 %2, %3\n\tv_mov_b64 %3, %9\n\t"
                         "v_mov_b64 %4, %5\n\tv_mov_b64 %5, %6\n\tv_mov_b64 %6, %7\n\tv_mov_b64 %7, %10\n\t"
                         "s_mov_b64 exec, %8"
                         : "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]), "+v"(xi[4]), "+v"(xi[5]), "+v"(xi[6]), "+v"(xi[7]),
                           "=&s"(ex_)
                         : "v"(xr[0]), "v"(xi[0]));
        }
        const int pl = (64 - lane) & 63;
        double Pr[4], Pi[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { Pr[c] = shfl_d(xr[7 - c], pl); Pi[c] = shfl_d(xi[7 - c], pl); }
        double* outA = A.logS + (fb + tA) * AFP_NBINS;
        double* outB = A.logS + (fb + tB) * AFP_NBINS;
        // Straight-line per variant (the pair has a second frame or not, wave-uniform): the eight logs of a
        // lane are independent, so their table reads and polynomial chains interleave.
        auto out_stage = [&](auto HB) {
            constexpr bool withB = decltype(HB)::value;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double pa, pb;
                split_power_unscaled(xr[c], xi[c], Pr[c], Pi[c], pa, pb);
                const double la = half_log(pa, ltab);
                if (CMP) LA[c] = la; else STFT_STORE(&outA[lane + 64 * c], la);
                pmax = fmax(pmax, pa);
                lmin = fmin(lmin, la);
                lsum += la;
                if (withB) {
                    const double lb = half_log(pb, ltab);
                    if (CMP) LB[c] = lb; else STFT_STORE(&outB[lane + 64 * c], lb);
                    pmax = fmax(pmax, pb);
                    lmin = fmin(lmin, lb);
                    lsum += lb;
                }
            }
        };
        if (haveB) out_stage(std::true_type{}); else out_stage(std::false_type{});
        if (PREFETCH) check_pair(p + 1);
        }   // valid
        if constexpr (CMP) {
            // onset filter  y = x + z ; z = -x + pole y  (:293-295) over the wave's two frames from a ZERO state:
            //   yA' = LA            zA' = -LA + pole LA
            //   yB' = LB + zA'      zB' = -LB + pole yB'
            // with the true state z before frame A:  yA = yA' + z,  yB = yB' + pole z,  state after B = zB' + pole^2 z
            int ln = lane;
            asm volatile("" : "+v"(ln));         // (per-lane addresses of this block are formed here, not carried through the FFT)
            double zl[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double za = fma(pole, LA[c], -LA[c]);
                const double yb = LB[c] + za;
                zl[c] = fma(pole, yb, -LB[c]);
                LB[c] = yb;
            }
            if (wave < STFT_WAVES - 1) {
                d2 w0, w1;
                w0.x = zl[0]; w0.y = zl[1]; w1.x = zl[2]; w1.y = zl[3];
                zx[1 + wave][0][ln] = w0; zx[1 + wave][1][ln] = w1;
            }
            lds_barrier();                       // (B1) the local end states of this iteration are in LDS
            // state before this wave's first frame: the iteration's entry state folded with the end states of the waves before
            // it, in time order.  One straight-line variant per wave (wave-uniform branch) whose LAST operation writes zin:
            // chained `if (wave > K) zin = fma(..)` steps made the compiler accumulate into the loaded registers and copy
            // them back at every join (four v_mov_b64 per step).
            double zin[4];
            {
                const d2 q0 = zx[0][0][ln], q1 = zx[0][1][ln];
                double a[4] = {q0.x, q0.y, q1.x, q1.y};
                auto fold = [&](int K) {
                    const d2 e0 = zx[1 + K][0][ln], e1 = zx[1 + K][1][ln];
                    a[0] = fma(pole2, a[0], e0.x); a[1] = fma(pole2, a[1], e0.y);
                    a[2] = fma(pole2, a[2], e1.x); a[3] = fma(pole2, a[3], e1.y);
                };
                if (wave == 1) { fold(0); }
                else if (wave == 2) { fold(0); fold(1); }
                else if (wave == 3) { fold(0); fold(1); fold(2); }
#pragma unroll
                for (int c = 0; c < 4; c++) zin[c] = a[c];
            }
            static_assert(STFT_WAVES == 4, "the fold above is written out for four wavefronts");
            lds_barrier();                       // (B2) everyone has read the states of this iteration
            if (wave == STFT_WAVES - 1) {        // state before the next iteration's first frame
                double znext[4];
#pragma unroll
                for (int c = 0; c < 4; c++) znext[c] = fma(pole2, zin[c], zl[c]);
                if (p + 1 < STFT_PAIRS_PER_WAVE) {
                    d2 w0, w1;
                    w0.x = znext[0]; w0.y = znext[1]; w1.x = znext[2]; w1.y = znext[3];
                    zx[0][0][ln] = w0; zx[0][1][ln] = w1;
                } else if (t0 + STFT_FPB < T) {
                    // end of the chunk: hand the filter state to the unit's next chunk -- write-through stores, drained, then
                    // the flag (MI355X_MICROARCH.md, "Valid forms": sc1 payload -> vmcnt(0) -> sc1 flag; the reader uses sc1 loads)
                    double* zc = KARG(double*, zcarry) + (int64_t)u * AFP_NBINS;
#pragma unroll
                    for (int c = 0; c < 4; c++) __hip_atomic_store(&zc[ln + 64 * c], znext[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (ln == 0 && !(u == KARG(int32_t, skip_unit) && t0 / STFT_FPB == KARG(int32_t, skip_chunk)))      // (test hook: -1 = none)
                        __hip_atomic_store(&KARG(unsigned long long*, zflag)[u], (KARG(unsigned long long, epoch) << 32) | (unsigned long long)(unsigned)(t0 / STFT_FPB + 1),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < 4; c++) { LA[c] = LA[c] + zin[c]; LB[c] = fma(pole, zin[c], LB[c]); }
                emit_frame(tA, LA, ln);
                if (haveB) emit_frame(tB, LB, ln);
            }
        }
    }
    if (CMP) asm volatile("s_dcache_wb" ::: "memory");          // the local-maximum masks went out through the scalar cache
    static_assert(FFT_LDS_DOUBLES - 2 * STFT_PAIRS_PER_WAVE >= 7 * FFT_X1_STRIDE + 64 && 7 * FFT_X1_STRIDE >= 7 * FFT_X2_STRIDE,
                  "the last 16 exchange-buffer elements, which no FFT pass touches, hold the Nyquist bins");
    wave_lds_fence();
    double nqr = 0.0, nqi = 0.0;
    if (lane < STFT_PAIRS_PER_WAVE) { nqr = lc[FFT_LDS_DOUBLES - 2 * STFT_PAIRS_PER_WAVE + 2 * lane]; nqi = lc[FFT_LDS_DOUBLES - 2 * STFT_PAIRS_PER_WAVE + 2 * lane + 1]; }
    if (lane < STFT_PAIRS_PER_WAVE) {
        const int tA = t0 + 2 * (wave + STFT_WAVES * lane);
        if (tA < T) {
            const int tB = tA + 1;
            double pa, pb;
            split_power_unscaled(nqr, nqi, nqr, nqi, pa, pb);
            const double la = half_log(pa, ltab);
            if (!CMP) A.nyq[fb + tA] = la;
            pmax = fmax(pmax, pa);
            lmin = fmin(lmin, la);
            lsum += la;
            if (tB < T) {
                const double lb = half_log(pb, ltab);
                if (!CMP) A.nyq[fb + tB] = lb;
                pmax = fmax(pmax, pb);
                lmin = fmin(lmin, lb);
                lsum += lb;
            }
        }
    }

    // pre-fill this chunk's slice of the buffers k_scan only writes sparsely
    {
        const int nt = min(STFT_FPB, T - t0);
        const int KK = KARG(int32_t, K);
        int tid = threadIdx.x;
        if constexpr (sizeof(ST) != 4 || LIST) asm volatile("" : "+v"(tid));     // (all but the float32 headline kernels: the per-lane offsets of this block are formed HERE, not before the pair loop and spilled across it)
        uint64_t* mk = KARG(uint64_t*, masks) + (fb + t0) * 4;
        for (int i = tid; i < nt * 4; i += STFT_WAVES * AFP_WAVE) mk[i] = 0ull;
        int32_t* cb = KARG(int32_t*, cand_bin) + (fb + t0) * (int64_t)KK;
        for (int i = tid; i < nt * KK; i += STFT_WAVES * AFP_WAVE) cb[i] = -1;
    }

    // deterministic reduction: xor-butterfly inside the wavefront, then waves in order
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        pmax = fmax(pmax, shfl_xor_d(pmax, s));
        lmin = fmin(lmin, shfl_xor_d(lmin, s));
        lsum += shfl_xor_d(lsum, s);
    }
    if (lane == 0) { red[0][wave] = pmax; red[1][wave] = lmin; red[2][wave] = lsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = red[0][0], mn = red[1][0], s = red[2][0];
        for (int w = 1; w < STFT_WAVES; w++) { m = fmax(m, red[0][w]); mn = fmin(mn, red[1][w]); s += red[2][w]; }
        // (compact mode lists its chunks time-major; the partials keep the unit-major order k_unit_stats reduces in)
        const int64_t pb = (CMP || LIST) ? ud.bbase + t0 / STFT_FPB : (int64_t)blk;
        double fv = 0.0;
        int f0 = 0x7fffffff, f1 = -1;
        for (int w = 0; w < STFT_WAVES; w++) { fv = fmax(fv, flat_s[w]); f0 = min(f0, flat_f[w][0]); f1 = max(f1, flat_f[w][1]); }
        double* pp = KARG(double*, blk_part) + pb; const int64_t ps = KARG(int64_t, part_stride);
        pp[0] = m; pp[ps] = mn; pp[2 * ps] = s; pp[3 * ps] = fv;
        if (fv > 0.0) { pp[4 * ps] = (double)f0; pp[5 * ps] = (double)f1; }
    }
    if (!LIST) break;
    __syncthreads();                                  // the tables and the reduction scratch are re-used by the next chunk
    }
}

extern "C" void afp_launch_stft(const StftArgs* a, int nblk, hipStream_t st)
{
    if (a->pcm_is_s16 == 1) hipLaunchKernelGGL((k_stft<int16_t, false>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
    else if (a->pcm_is_s16 == 2) hipLaunchKernelGGL((k_stft<double, false>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
    else hipLaunchKernelGGL((k_stft<float, false>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
}
// dense transform of the chunks listed by k_unit_stats (a->list_*): a fixed grid strides over the list
extern "C" void afp_launch_stft_list(const StftArgs* a, int grid, hipStream_t st)
{
    if (a->pcm_is_s16 == 1) hipLaunchKernelGGL((k_stft<int16_t, false, true>), dim3(grid), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
    else if (a->pcm_is_s16 == 2) hipLaunchKernelGGL((k_stft<double, false, true>), dim3(grid), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
    else hipLaunchKernelGGL((k_stft<float, false, true>), dim3(grid), dim3(STFT_WAVES * AFP_WAVE), 0, st, *a);
}
// compact spectral stage: a->blk_unit / blk_t0 must be the TIME-MAJOR chunk list
// Compact mode needs 39.6 KB of LDS: four such workgroups take 158 of a CU's 160 KB, so the scan workgroups of the previous
// batch (8 KB each) share a CU with three of them at most.  AFP_STFT_PAD_LDS=<bytes> of unused DYNAMIC LDS (the register
// allocation still targets four wavefronts per SIMD) caps a CU at three STFT workgroups from the start; measured on C3:
// 1.507 ms per step with 2 KB against 1.516 without, and 1.11 against 1.00 ms for the kernel alone -- off by default.
static size_t compact_pad_lds()
{
    static long pad = -1;
    if (pad < 0) { const char* e = getenv("AFP_STFT_PAD_LDS"); pad = e ? atol(e) : 0; }
    return (size_t)pad;
}
extern "C" void afp_launch_stft_compact(const StftArgs* a, int nblk, hipStream_t st)
{
    const size_t dyn = compact_pad_lds();
    if (a->pcm_is_s16 == 1) hipLaunchKernelGGL((k_stft<int16_t, true>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), dyn, st, *a);
    else if (a->pcm_is_s16 == 2) hipLaunchKernelGGL((k_stft<double, true>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), dyn, st, *a);
    else hipLaunchKernelGGL((k_stft<float, true>), dim3(nblk), dim3(STFT_WAVES * AFP_WAVE), dyn, st, *a);
}
