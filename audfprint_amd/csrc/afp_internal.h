// afp_internal.h -- what the host-side translation units of libafp_hip.so share: the handle, the launchers of the kernel
// files, the workspace / event helpers (afp_host.hip), the batch finalizer (afp_abi.hip).  Nothing here is exported: the
// library is built with -fvisibility=hidden and only the AFP_API declarations of include/afp.h are visible.
#pragma once
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <math.h>
#include <signal.h>
#include <unistd.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/afp.h"
#include "afp_common.h"

// launchers defined in k_stft.hip / k_scan.hip / k_pair.hip / k_table.hip (C linkage, hidden visibility)
extern "C" {
void afp_launch_stft(const StftArgs*, int, hipStream_t);
void afp_launch_stft_compact(const StftArgs*, int, hipStream_t);
void afp_launch_stft_list(const StftArgs*, int, hipStream_t);
void afp_launch_scan_compact(const ScanArgs*, int, hipStream_t);
void afp_launch_scan_dummy(int, int, double*, hipStream_t);
void afp_launch_hpf(const HpfArgs*, int, hipStream_t);
void afp_launch_hpf_verify(const double*, int, int32_t*, int, hipStream_t);
void afp_launch_scan_seg(const ScanArgs*, int, hipStream_t);
void afp_launch_seg_verify(const ScanArgs*, hipStream_t);
void afp_launch_unit_stats(const StatsArgs*, hipStream_t);
void afp_launch_floor_corr(const CorrArgs*, int, hipStream_t);
void afp_launch_stats_corr(const StatsArgs*, const CorrArgs*, int, hipStream_t);
void afp_launch_scan(const ScanArgs*, int, hipStream_t);
void afp_launch_scan_small(const ScanArgs*, int, hipStream_t);
void afp_launch_mask_popc(const uint64_t*, int32_t*, int64_t, hipStream_t);
void afp_launch_pair(const PairArgs*, int, hipStream_t);
void afp_launch_pair_rows(const PairArgs*, const PairRowsArgs*, int, hipStream_t);
void afp_launch_rows_count(const int32_t*, const int64_t*, int, int64_t, const int64_t*, int32_t*, hipStream_t);
void afp_launch_merge(const MergeArgs*, int, hipStream_t);
void afp_launch_pairmerge(const PairMergeArgs*, int, hipStream_t);
void afp_launch_pairlane(const PairMergeArgs*, int, hipStream_t);
void afp_launch_vote_count(const int32_t*, int64_t, int, int32_t*, int32_t*, hipStream_t);
void afp_launch_vote_compact(const int32_t*, int, int32_t*, int32_t*, int32_t*, hipStream_t);
void afp_launch_vote_setrank(const int32_t*, int, int, int32_t*, hipStream_t);
void afp_launch_vote_hist(const int32_t*, int64_t, int, const int32_t*, int, int, int32_t*, hipStream_t);
void afp_launch_vote_select(const int32_t*, int64_t, int, const int32_t*, const int32_t*, const int32_t*, const int32_t*, int32_t*, const int64_t*, int32_t*, int, hipStream_t);
size_t afp_pairlane_lds(int, int, int);
size_t afp_pairlane_ms_lds(int, int, int, int, int);
void afp_launch_pairlane_ms(const PairMergeArgs*, int, hipStream_t);
void afp_launch_seg_scan(const SegScanArgs*, int, hipStream_t);
void afp_launch_excl_scan64(const int64_t*, int64_t*, int, hipStream_t);
void afp_launch_excl_scan64_wide(const int64_t*, int64_t*, int, int64_t*, hipStream_t);
void afp_launch_scatter_hashes(const ScatterHashArgs*, int, hipStream_t);
void afp_launch_scatter_peaks(const ScatterPeakArgs*, int, hipStream_t);
void afp_launch_export(const ExportArgs*, int, hipStream_t);
int afp_finish_one_max_frames(void);
void afp_launch_finish_one(const ScatterHashArgs*, int32_t*, int64_t*, int64_t*, const ExportArgs*, hipStream_t);
void afp_launch_scatter_landmarks(const ScatterLmArgs*, int, hipStream_t);
void afp_launch_masks_from_peaks(const int32_t*, const int64_t*, int, int64_t, const int64_t*, uint64_t*, hipStream_t);
void afp_launch_lm2hash(const int32_t*, int32_t*, int64_t, hipStream_t);
void afp_launch_tb_count(const TableArgs*, hipStream_t);
void afp_launch_tb_scatter(const TableArgs*, hipStream_t);
void afp_launch_tb_fill(const TableArgs*, hipStream_t);
void afp_launch_tb_fill_big(const TableArgs*, hipStream_t);
void afp_launch_tb_merge(uint32_t*, int32_t*, const uint32_t*, const int32_t*, const int64_t*, int, int, int, uint32_t, int32_t*, int32_t*, hipStream_t);
void afp_launch_tb_merge_gather(const uint32_t*, const int32_t*, const uint32_t*, const int32_t*, const int64_t*, int, int, uint32_t, const int32_t*, int,
                                uint32_t*, int32_t*, hipStream_t);
void afp_launch_tb_pack_len(const int32_t*, int, int, int64_t*, hipStream_t);
void afp_launch_tb_pack_gather(const uint32_t*, const int64_t*, int, int, uint32_t*, hipStream_t);
void afp_launch_tb_patch(uint32_t*, int, const int32_t*, int64_t, hipStream_t);
void afp_launch_tb_clip_counts(int32_t*, int, int, hipStream_t);
void afp_launch_gh_count(const int32_t*, int64_t, int, int, const int32_t*, int64_t*, hipStream_t);
void afp_launch_gh_fill(const int32_t*, int64_t, int, int, int, const uint32_t*, const int32_t*, const int64_t*, int32_t*, hipStream_t);
}

extern thread_local std::string g_hip_err;      // text of the last failed runtime call of this thread (afp_last_hip_error)

#define HIPCHK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            g_hip_err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return AFP_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

enum { KS_STFT = 0, KS_STATS, KS_CORR, KS_SCAN, KS_PAIR, KS_MERGE, KS_SEGSCAN_H, KS_EXCL, KS_SCAT_H,
       KS_SEGSCAN_P, KS_SCAT_P, KS_PIPELINE };

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct EvPair {
    int slot;
    hipEvent_t a, b;
};

struct Geometry {
    int32_t nclips, nunits, S;
    int64_t total_frames, total_mframes, nblk, ncblk, nmblk, npblk;
    int32_t pch;                 // columns per k_pairmerge workgroup
};

struct afp_handle {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // staged mode (afp_set_stage_streams): the spectral stage and the scan/pair stage of one batch go to
    // two caller-owned streams shared between handles, so that consecutive batches pipeline stage against stage
    hipStream_t stage_a = nullptr, stage_b = nullptr, stage_c = nullptr;
    hipEvent_t ev_in = nullptr, ev_a = nullptr, ev_s = nullptr, ev_b = nullptr;
    hipEvent_t ev_up_done = nullptr;       // this handle's upload on the device's upload stream has landed (extract_host_any)
    hipStream_t tstream = nullptr;       // stream the per-kernel timing events of the current stage go to
    bool join_pending = false;           // a staged batch is in flight; ev_b marks its end
    bool have_params = false;
    afp_params prm;
    int64_t ws_limit = (int64_t)200 << 30;
    // constant tables
    DevBuf d_tables, d_gauss;              // d_tables: window | twiddles | half-log table (k_stft reads them through one pointer)
    // descriptors: host staging (pinned) + device image
    void* h_stage = nullptr;
    size_t h_stage_cap = 0;

    DevBuf d_desc;
    std::vector<int32_t> unit_T_host;      // frames per unit of the current descriptors
    std::vector<int64_t> last_offsets;
    int last_S = -1;
    std::vector<int32_t> last_shift_offsets;
    bool desc_valid = false;
    // geometry of the current batch
    int32_t nclips = 0, nunits = 0, S = 1;
    int64_t total_frames = 0, total_mframes = 0;
    int64_t nblk = 0, ncblk = 0, nmblk = 0;
    // device descriptor pointers (into d_desc)
    int64_t *unit_pcm_off = nullptr, *unit_n = nullptr, *unit_fbase = nullptr, *unit_bbase = nullptr;
    int32_t *unit_T = nullptr, *blk_unit = nullptr, *blk_t0 = nullptr, *cblk_unit = nullptr, *cblk_t0 = nullptr;
    UnitDesc* udesc = nullptr;                              // the unit_* arrays again, one record per unit (k_stft)
    ChunkDesc *blk2 = nullptr, *tblk2 = nullptr;            // the STFT chunks as records: unit-major, and TIME-MAJOR (compact spectral stage)
    int64_t* clip_mfbase = nullptr;
    int32_t *clip_T0 = nullptr, *mblk_clip = nullptr, *mblk_t0 = nullptr, *pblk_clip = nullptr, *pblk_t0 = nullptr;
    // workspace
    DevBuf pcm_stage, logS, nyq, blk_part, blk_corr, stats, cand_val, cand_bin, masks,
        pcnt, ylast, unit_mean, sgram_dbg, cvals, lmask, head, zcarry, zflag, cerr, corr_list, seg_desc, seg_state, seg_status, seg_flag, hpf_dump, hslots, hcnt, mslots, mcnt, hoffs, poffs, clip_tot, unit_tot, clip_hoff,
        unit_poff, out_hashes, out_peaks, scan_prof, lslots, lcnt, loffs, unit_ltot, unit_loff, out_landmarks,
        in_peaks, in_upo, lm_in, lm_out, tb_table, tb_counts, tb_newcnt, tb_first, tb_fill, tb_seg, tb_overflow, tb_misc,
        tb_biglist, tb_scan, tb_pklen, tb_pkoff, tb_packed, tb_olen, tb_ooff, tb_rows, tb_off, tb_ids, tb_otable, tb_ocounts, tb_mlist, tb_mvals, tb_mnv, tb_patch, gh_rows, gh_nids, gh_off, gh_hits, vt_idcount, vt_misc, vt_ids, vt_cnt,
        vt_rank, vt_hist, vt_want, vs_q, vs_cursor, vs_off, vs_out;
    std::vector<int64_t> vs_offsets;         // afp_table_select_hits: row offsets per query, in the caller's query order
    std::vector<int32_t> vs_perm;            // caller's query -> position in the id-sorted list the kernel walked
    std::vector<int32_t> vs_cnt;             // rows per query, caller's order
    int64_t vs_total = -1;
    int64_t gh_total = 0;
    // vote counting over the hits of the last afp_table_get_hits
    bool vt_counted = false;
    int32_t vt_nids = 0, vt_mintime = 0, vt_width = 0, vt_hist_rows = 0, vt_maxotime = 0;
    int32_t tb_hashbits = 0, tb_depth = 0, tb_maxtimebits = 0;
    int64_t tb_novf = 0;
    void* h_dl = nullptr;                   // pinned ring the table download is staged through (afp_table_download)
    void* h_dlc = nullptr;                  // pinned: the counts on their way out (afp_table_download_filled)
    size_t h_dlc_cap = 0;
    hipEvent_t dlc_ev = nullptr;
    std::vector<int64_t> pk_hoff;           // host: exclusive offsets of min(counts, depth) (afp_table_download_filled)
    int64_t pk_total = -1;                  // entries of the last afp_table_pack (-1: none / the table has changed since)
    hipEvent_t dl_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* h_ovf = nullptr;                  // pinned: overflow events of the last store (afp_table_replay_overflow)
    size_t h_ovf_cap = 0;
    std::vector<int32_t> ovf_slot, ovf_patch;
    std::vector<uint32_t> ovf_ord, ovf_tmp;
    std::vector<uint64_t> ovf_seen;
    hipStream_t tb_stream = nullptr;        // table / vote kernels and copies (highest priority; see tbs())
    hipStream_t probe_stream = nullptr;     // afp_clock_probe_*
    DevBuf probe_buf;
    int probe_khz = 100000;
    // HashTable.merge in flight: the other table (device), its depth / id offset, the over-full buckets
    const uint32_t* mg_otable = nullptr;
    const int32_t* mg_ocounts = nullptr;
    const int64_t* mg_ooff = nullptr;       // the other table came PACKED: its row offsets (tb_ooff)
    int32_t mg_odepth = 0, mg_nov = 0;
    uint32_t mg_idoffset = 0;
    // results
    int64_t* h_totals = nullptr;          // pinned: [0] hashes, [1] peaks of the batch in flight
    // small batches (one file per call): k_export leaves the results in this pinned image at the end of the chain
    char* h_export = nullptr;
    int64_t h_export_cap = 0;
    bool export_mode = false;             // the batch in flight ends with k_export (which also delivers the totals)
    bool export_redo = false;             // finalize() had to re-run a scatter: the image is void
    bool fuse_finish = false;             // one clip, hashes only: offsets + scatter + export are ONE launch (k_finish_one)
    void* seg_clean_ptr = nullptr;        // seg_status block known to be all zero (k_export / k_finish_one of the previous batch cleared it)
    size_t seg_clean_bytes = 0;
    size_t seg_zero_bytes = 0;            // ... bytes of it the batch in flight uses
    size_t seg_clean_keep = 0;
    int export_max_units = 64;            // AFP_EXPORT_MAX_UNITS (0: never)
    bool finalized = true;
    ScatterHashArgs sh; int sh_nblk = 0; bool have_sh = false;
    ScatterPeakArgs sp; int sp_nblk = 0; bool have_sp = false;
    ScatterLmArgs sl; int sl_nblk = 0; bool have_sl = false;
    int64_t last_th = 0, last_tp = 0, last_tl = 0;
    int64_t total_landmarks = 0;
    Geometry geom;
    bool extracted = false;
    uint32_t flags = 0;
    int64_t total_hashes = 0, total_peaks = 0;
    int32_t K = 0;
    // compact spectral stage (k_stft<ST, true> -> k_scan_c): see run_spectral
    int compact_mode = -1;                 // AFP_COMPACT=0|1 forces the dense / compact pipeline (default: by batch size)
    int compact_min_units = 768;           // AFP_COMPACT_MIN_UNITS: fewer units than about one residency of chunks would serialise on the state hand-off
    bool batch_compact = false;            // the batch in flight went through the compact stage
    unsigned long long epoch = 0;          // launches of the compact STFT on this handle (tags the hand-off flags)
    double nt_eps = 0.0;                   // near-tie guard of the scan (afp_set_neartie_eps / AFP_NEARTIE_EPS; 0: off, the default)
    int32_t nt_units_last = 0;             // units the guard marked in the batch last finalized
    int32_t nt_redone_total = 0;           // compact batches re-run densely because the guard fired
    bool batch_nt_redone = false;
    int compact_force_timeout = 0;         // test hook (afp_set_compact_force_timeout): one chunk withholds its state, the wait bound is short
    int32_t compact_redone_total = 0;      // batches whose compact stage reported a hand-off fault and were re-run on the dense path
    bool batch_redone = false;             // ... the batch last finalized was one of them
    // what finalize() needs to re-run the batch in flight: the caller's PCM (device pointer as given; it must stay valid until
    // the results have been fetched -- afp.h), its sample type and the extract flags; the offsets are last_offsets
    const void* cur_pcm = nullptr;
    int cur_kind = 0;
    uint32_t cur_flags = 0;
    // pipeline selection as it stood after afp_create (defaults + AFP_COMPACT / AFP_SEG* of the environment): what
    // afp_set_pipeline's "creation-time value" arguments restore
    int init_compact_mode = -1, init_compact_min_units = 768, init_seg_mode = -1, init_seg_max_units = 128, init_seg_len = 0, init_seg_warm = 0;
    // segment-parallel scan of few long units (k_scan_seg): see run_scan
    int seg_mode = -1;                     // AFP_SEG=0|1 forces it off / on (default: few units)
    int seg_max_units = 128;               // AFP_SEG_MAX_UNITS
    int seg_len = 0;                       // AFP_SEG_LEN: own frames per segment (0: from the warm-up length)
    int seg_warm = 0;                      // AFP_SEG_WARM: warm-up frames (0: 1 / (1 - a_dec), clamped)
    int seg_force_fail = 0;                // test hook (afp_set_seg_force_fail): the final check marks every unit
    // short files (r05): a cut of (32, 96) instead of (64, 128) while it converges -- see run_scan
    bool seg_adapt = true;                 // AFP_SEG_ADAPT=0: always the standard cut
    int seg_short_penalty = 0;             // batches that still take the standard cut after a short cut re-ran too many segments
    bool batch_short_cut = false;
    int32_t seg_short_total = 0, seg_short_backoffs = 0;
    std::vector<SegDesc> seg_host;         // host images of the last cut (copied into the pinned h_seg_stage for the upload; kept so
    std::vector<int32_t> seg_doff, seg_dfr; // that a repeated batch shape re-uses the device image: seg_cache_ok)
    std::vector<int32_t> seg_ufirst_host;
    // One upload, one memset per segmented batch: seg_desc holds [SegDesc x nseg | dump offsets, dump frames | first segment
    // per unit] (staged in pinned memory), seg_status holds [status (256 B) | per-unit fail flags | per-segment re-run marks]
    char* h_seg_stage = nullptr;
    size_t h_seg_stage_cap = 0;
    int32_t *seg_ufail_p = nullptr, *seg_rerun_p = nullptr, *seg_ufirst_p = nullptr, *hpf_idx_p = nullptr;
    // chunk mode of k_hpf (long units): host images of the two chunk lists (cached with the segment cut), their device copies
    std::vector<HpfChunk> hpf_c1, hpf_c2;
    const HpfChunk *hpf_c1_p = nullptr, *hpf_c2_p = nullptr;
    int hpf_nbnd = 0, hpf_ngran = 0;
    int32_t hpf_par_total = 0;             // batches whose onset filter ran chunked
    DevBuf hpf_gran, hpf_bnd;
    bool desc_cached = false;              // this batch re-used the descriptors of the previous one
    bool seg_cache_ok = false;             // ... and seg_desc still holds the segments cut for them with (seg_cache_W, seg_cache_S)
    int seg_cache_W = 0, seg_cache_S = 0, seg_cache_longest = 0;
    bool batch_seg = false;
    int seg_ndoff = 0;
    int32_t batch_nseg = 0;
    // timing
    bool timing = false;
    bool force_generic_pair = false;       // AFP_GENERIC_PAIR=1: use k_pair + k_merge instead of k_pairmerge
    int scan_lds_mode = 0;                 // AFP_SCAN_LDS=small|big forces a k_scan variant (default: by batch size)
    int pair_K = 0;                        // peaks per column the pairing stage must allow for (0: maxpksperframe)
    bool pair_rows = false;            // afp_pairs_from_peaks on list-order lists: k_pair_rows instead of the mask kernels
    bool no_pairlane = false;              // AFP_NO_PAIRLANE=1: keep k_pairmerge where k_pairlane would apply
    int pairlane_ms_pch = 32;              // AFP_PAIRLANE_MS_PCH: columns per k_pairlane_ms workgroup (measured best on C5: 32)
    bool pairlane_ms = true;               // AFP_PAIRLANE_MS=0: k_pairmerge instead of the lane-per-peak kernel for several shifts
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> ev_pool;
    double t_ms[AFP_NKERNELS] = {0};
    int64_t t_n[AFP_NKERNELS] = {0};
};

// ---- afp_host.hip ----------------------------------------------------------------------------------------------------------
void drain_retired(bool force);
int ensure(DevBuf& b, size_t bytes, bool rows = false);
#define ENSURE(buf, bytes)                        \
    do {                                          \
        int r_ = ensure(buf, (size_t)(bytes));    \
        if (r_ != AFP_OK) return r_;              \
    } while (0)
hipError_t sync_handle(afp_handle* h);
hipEvent_t get_event(afp_handle* h);
void resolve_timings(afp_handle* h);
struct Timed {
    afp_handle* h;
    EvPair ep;
    bool on;
    Timed(afp_handle* h_, int slot) : h(h_), on(h_->timing)
    {
        if (on) {
            ep.slot = slot; ep.a = get_event(h); ep.b = get_event(h);
            if (!ep.a || !ep.b) { on = false; return; }
            (void)hipEventRecord(ep.a, h->tstream ? h->tstream : h->stream);
        }
    }
    ~Timed()
    {
        if (on) { (void)hipEventRecord(ep.b, h->tstream ? h->tstream : h->stream); h->pending.push_back(ep); }
    }
};

// ---- host helpers of the big device -> host copies --------------------------------------------------------------------
// A small PERSISTENT pool (r04 created and joined seven threads per download, and they spun for the whole copy -- ADVICE r4):
// the workers sleep on a condition variable between jobs and spin only inside one (a table download: a few milliseconds).
// Size: AFP_DL_THREADS, else min(8, CPUs this process may run on -- a NUMA-bound rank counts its own node's cores).  Thread
// creation that fails (std::system_error must not cross the C ABI) just leaves a smaller pool; one thread = the caller alone.
// The workers make NO runtime calls.  A forked child starts with a fresh pool (threads do not survive fork).
struct HostPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> th;
    std::function<void(int)> fn;
    uint64_t job = 0;
    std::atomic<int> left{0};
    pid_t pid = 0;
    int W = 1;
    void worker(int w)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return job != seen; });
                seen = job;
                f = fn;
            }
            f(w);
            left.fetch_sub(1, std::memory_order_release);
        }
    }
    // fn(w) on every thread of the pool, w = 0 (the caller) .. W - 1; returns when all are done
    std::mutex run_mu;                      // one job at a time (two handles may download from two host threads)
    void run(const std::function<void(int)>& f)
    {
        std::lock_guard<std::mutex> only(run_mu);
        if (W > 1) {
            left.store(W - 1, std::memory_order_relaxed);
            { std::lock_guard<std::mutex> lk(mu); fn = f; job++; }
            cv.notify_all();
        }
        f(0);
        while (left.load(std::memory_order_acquire) > 0) { __builtin_ia32_pause(); }
    }
};
HostPool* host_pool();

// The ring both downloads stage through: R pinned chunks of CH bytes, an event per slot
static constexpr int DL_R = 4;
static constexpr int64_t DL_CH = (int64_t)8 << 20;
int dl_ring(afp_handle* h);
// `bytes` of device memory through the ring; `consume(k, n, ring_chunk, w, W)` runs on every pool thread for chunk k (n bytes)
// once it has landed.  Only the calling thread talks to the runtime.
template <class F>
static inline int ring_download(afp_handle* h, const char* src, int64_t bytes, hipStream_t st, F consume)
{
    { const int r = dl_ring(h); if (r != AFP_OK) return r; }
    HostPool* P = host_pool();
    const int W = P->W;
    const int64_t nch = (bytes + DL_CH - 1) / DL_CH;
    char* ring = (char*)h->h_dl;
    auto len_of = [&](int64_t k) { return std::min<int64_t>(DL_CH, bytes - k * DL_CH); };
    hipError_t herr = hipSuccess;
    auto issue = [&](int64_t k) {
        hipError_t e = hipMemcpyAsync(ring + (k % DL_R) * DL_CH, src + k * DL_CH, (size_t)len_of(k), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(h->dl_ev[k % DL_R], st);
        if (e != hipSuccess && herr == hipSuccess) herr = e;
    };
    for (int64_t k = 0; k < std::min<int64_t>(DL_R, nch); k++) issue(k);
    // workers: their share of chunk `gen - 1` once `gen` says it has landed
    std::atomic<int64_t> gen{0}, done{0};
    P->run([&](int w) {
        if (w != 0) {
            for (int64_t k = 0; k < nch; k++) {
                while (gen.load(std::memory_order_acquire) <= k) { __builtin_ia32_pause(); }
                if (gen.load(std::memory_order_acquire) > nch) return;          // (error: released without data)
                consume(k, len_of(k), ring + (k % DL_R) * DL_CH, w, W);
                done.fetch_add(1, std::memory_order_release);
            }
            return;
        }
        for (int64_t k = 0; k < nch && herr == hipSuccess; k++) {
            hipError_t e = hipEventSynchronize(h->dl_ev[k % DL_R]);
            if (e != hipSuccess) { herr = e; break; }
            gen.store(k + 1, std::memory_order_release);
            consume(k, len_of(k), ring + (k % DL_R) * DL_CH, 0, W);
            while (done.load(std::memory_order_acquire) < (k + 1) * (int64_t)(W - 1)) { __builtin_ia32_pause(); }
            if (k + DL_R < nch) issue(k + DL_R);
        }
        if (herr != hipSuccess) gen.store(nch + 1, std::memory_order_release);
    });
    if (herr != hipSuccess) { (void)hipStreamSynchronize(st); HIPCHK(herr); }
    return AFP_OK;
}

int download_pageable(afp_handle* h, char* dst, const char* src, int64_t bytes, hipStream_t st);

// ---- afp_abi.hip: wait for the batch in flight and settle its results (every afp_result_* / afp_fetch_* / afp_table_* entry) ----
int finalize(afp_handle* h);
#define FINALIZE(h)                      \
    do {                                 \
        int r_ = finalize(h);            \
        if (r_ != AFP_OK) return r_;     \
    } while (0)

// ---- afp_table.hip: the stream the table / vote kernels and copies run on ---------------------------------------------------
static inline hipStream_t tbs(afp_handle* h) { return h->tb_stream ? h->tb_stream : h->stream; }
hipError_t tb_sync(afp_handle* h);

