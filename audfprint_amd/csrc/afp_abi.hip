// afp_abi.hip -- host side of libafp_hip.so: the C ABI declared in include/afp.h.
// Builds the per-batch descriptors (units, frame chunks), owns the grow-only HBM workspace,
// enqueues the kernels of k_stft.hip / k_scan.hip / k_pair.hip on one HIP stream and copies
// results out.  No torch types, no exceptions across the boundary.
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

extern "C" {
void afp_launch_stft(const StftArgs*, int, hipStream_t);
void afp_launch_stft_compact(const StftArgs*, int, hipStream_t);
void afp_launch_stft_list(const StftArgs*, int, hipStream_t);
void afp_launch_scan_compact(const ScanArgs*, int, hipStream_t);
void afp_launch_scan_dummy(int, int, double*, hipStream_t);
void afp_launch_hpf(const HpfArgs*, int, hipStream_t);
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

static thread_local std::string g_hip_err;

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
static const char* k_names[AFP_NKERNELS] = {"k_stft", "k_unit_stats", "k_floor_corr", "k_scan", "k_pair",
                                            "k_merge", "k_seg_scan(hashes)", "k_excl_scan64",
                                            "k_scatter_hashes", "k_seg_scan(peaks)", "k_scatter_peaks",
                                            "pipeline(first launch..last launch)"};

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

// Buffers that had to grow leave their old allocation HERE instead of calling hipFree on the spot: hipFree waits for every
// stream of the device (r04: 8 ms in the middle of the pipelined c4 job, behind two queued uploads), hipMalloc does not.
// The retired allocations are released in one go at a moment that is idle anyway -- the end of a batch whose results are
// being fetched (finalize), the end of a table download, afp_destroy -- once they add up to AFP_RETIRE_MAX_MB (default
// 1024), or at once if an allocation fails.  Releasing them is safe at any time (hipFree's own wait makes it so); the list
// only decides WHEN the wait is paid.  Process-wide, per device.
struct Retired { int device; void* p; size_t bytes; };
static std::mutex g_retire_mu;
static std::vector<Retired> g_retired;
static size_t g_retired_bytes = 0;
static size_t retire_limit()
{
    static size_t lim = 0;
    if (!lim) { const char* e = getenv("AFP_RETIRE_MAX_MB"); lim = ((size_t)(e && atol(e) >= 0 ? atol(e) : 1024) << 20) + 1; }
    return lim;
}
static void drain_retired(bool force)
{
    std::vector<Retired> take;
    {
        std::lock_guard<std::mutex> g(g_retire_mu);
        if (g_retired.empty() || (!force && g_retired_bytes < retire_limit())) return;
        take.swap(g_retired);
        g_retired_bytes = 0;
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (const Retired& r : take) { if (hipSetDevice(r.device) == hipSuccess) (void)hipFree(r.p); }
    if (have_cur) (void)hipSetDevice(cur);
}
extern "C" int64_t afp_retired_bytes(void) { std::lock_guard<std::mutex> g(g_retire_mu); return (int64_t)g_retired_bytes; }

static int ensure(DevBuf& b, size_t bytes, bool rows = false)
{
    if (bytes <= b.cap && b.p) return AFP_OK;
    if (bytes == 0) bytes = 256;
    const bool regrow = b.p != nullptr;
    if (b.p) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(g_retire_mu);
        g_retired.push_back(Retired{dev, b.p, b.cap});
        g_retired_bytes += b.cap;
        b.p = nullptr; b.cap = 0;
    }
    // a buffer that has to GROW gets headroom: batches of a real ingest differ by a few rows.  First allocations are exact,
    // except buffers sized by a batch's ROW count (`rows`: the next batch of the same shape has a few rows more or less):
    // those start with an eighth to spare.
    size_t want = bytes;
    if (regrow) want += bytes >= ((size_t)1 << 30) ? bytes / 8 : bytes / 4;
    else if (rows) want += bytes / 8;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        // out of memory with allocations parked on the retire list: release them (this is the wait the list postpones), then
        // once more, exact size last
        (void)hipGetLastError();
        drain_retired(true);
        e = hipMalloc(&b.p, want);
        if (e != hipSuccess && want != bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&b.p, want); }
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return AFP_ERR_NOMEM;
    }
    b.cap = want;
    return AFP_OK;
}
#define ENSURE(buf, bytes)                        \
    do {                                          \
        int r_ = ensure(buf, (size_t)(bytes));    \
        if (r_ != AFP_OK) return r_;              \
    } while (0)

// Wait (on the host) for everything this handle has queued: a staged batch is joined through its completion
// event -- NOT by making the handle's stream wait for it: HIP multiplexes streams onto a few hardware queues,
// and a queue barrier parked on a stream that shares its queue with a stage stream would stall the stages
// of the other handles behind it.
static hipError_t sync_handle(afp_handle* h)
{
    if (h->join_pending) {
        hipError_t e = hipEventSynchronize(h->ev_b);
        if (e != hipSuccess) return e;
        h->join_pending = false;
    }
    return hipStreamSynchronize(h->stream);
}

static hipEvent_t get_event(afp_handle* h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
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
static void resolve_timings(afp_handle* h)
{
    for (auto& ep : h->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            h->t_ms[ep.slot] += ms;
            h->t_n[ep.slot] += 1;
        }
        h->ev_pool.push_back(ep.a);
        h->ev_pool.push_back(ep.b);
    }
    h->pending.clear();
}

extern "C" int afp_abi_version(void) { return AFP_ABI_VERSION; }
#ifndef AFP_BUILD_ID
#define AFP_BUILD_ID "unknown"
#endif
extern "C" const char* afp_build_id(void) { return AFP_BUILD_ID; }

extern "C" const char* afp_strerror(int s)
{
    switch (s) {
        case AFP_OK: return "ok";
        case AFP_ERR_ARG: return "bad argument";
        case AFP_ERR_PARAM: return "parameter outside the supported range";
        case AFP_ERR_HIP: return "HIP runtime error (see afp_last_hip_error)";
        case AFP_ERR_NOMEM: return "device workspace limit exceeded or allocation failed";
        case AFP_ERR_STATE: return "call order violated";
        case AFP_ERR_NODEVICE: return "no usable gfx950 device";
        default: return "unknown afp status";
    }
}
extern "C" const char* afp_last_hip_error(void) { return g_hip_err.c_str(); }
extern "C" const char* afp_kernel_name(int slot) { return (slot >= 0 && slot < AFP_NKERNELS) ? k_names[slot] : ""; }

// out[0] = HIP_VERSION the library was COMPILED against (hipcc of the build), out[1] = hipRuntimeGetVersion() of the runtime the
// process actually bound (PyTorch wheels bundle their own libamdhip64 under the same SONAME: audfprint_amd/_lib.py),
// out[2] = hipDriverGetVersion(), out[3] = devices visible.  Makes a HIP call: the runtime is initialised afterwards.
extern "C" int afp_runtime_info(int32_t* out)
{
    if (!out) return AFP_ERR_ARG;
    int rt = 0, drv = 0, n = 0;
    out[0] = (int32_t)HIP_VERSION;
    HIPCHK(hipRuntimeGetVersion(&rt));
    if (hipDriverGetVersion(&drv) != hipSuccess) drv = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    out[1] = rt; out[2] = drv; out[3] = n;
    return AFP_OK;
}

extern "C" int afp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// AFP_BACKTRACE=1 (debugging aid): a SIGSEGV inside the process prints the native stack (addresses resolve with addr2line
// against this library) before the default action takes over
static void afp_segv_handler(int sig)
{
    void* fr[64];
    const int n = backtrace(fr, 64);
    const char msg[] = "libafp_hip: SIGSEGV, native stack:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(fr, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
extern "C" int afp_create(int device, afp_handle** out)
{
    { static bool once = false; if (!once && getenv("AFP_BACKTRACE")) { once = true; signal(SIGSEGV, afp_segv_handler); } }
    if (!out) return AFP_ERR_ARG;
    *out = nullptr;
    int n = afp_device_count();
    if (n <= 0 || device < 0 || device >= n) return AFP_ERR_NODEVICE;
    HIPCHK(hipSetDevice(device));
    afp_handle* h = new afp_handle();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) { delete h; return AFP_ERR_HIP; }
    h->stream = h->own_stream;
    { const char* e = getenv("AFP_GENERIC_PAIR"); h->force_generic_pair = e && e[0] == '1'; }
    { const char* e = getenv("AFP_NO_PAIRLANE"); h->no_pairlane = e && e[0] == '1'; }
    { const char* e = getenv("AFP_PAIRLANE_MS"); if (e) h->pairlane_ms = e[0] == '1'; }
    { const char* e = getenv("AFP_PAIRLANE_MS_PCH"); if (e && atoi(e) >= 16 && atoi(e) % 4 == 0) h->pairlane_ms_pch = atoi(e); }
    { const char* e = getenv("AFP_SCAN_LDS"); h->scan_lds_mode = !e ? 0 : e[0] == 's' ? 1 : e[0] == 'b' ? 2 : 0; }
    { const char* e = getenv("AFP_COMPACT"); if (e && (e[0] == '0' || e[0] == '1')) h->compact_mode = e[0] - '0'; }
    { const char* e = getenv("AFP_COMPACT_MIN_UNITS"); if (e && atoi(e) >= 1) h->compact_min_units = atoi(e); }
    { const char* e = getenv("AFP_SEG"); if (e && (e[0] == '0' || e[0] == '1')) h->seg_mode = e[0] - '0'; }
    { const char* e = getenv("AFP_SEG_MAX_UNITS"); if (e && atoi(e) >= 1) h->seg_max_units = atoi(e); }
    { const char* e = getenv("AFP_SEG_LEN"); if (e && atoi(e) >= 8) h->seg_len = atoi(e); }
    { const char* e = getenv("AFP_SEG_WARM"); if (e && atoi(e) >= 1) h->seg_warm = atoi(e); }
    { const char* e = getenv("AFP_SEG_ADAPT"); if (e && e[0] == '0') h->seg_adapt = false; }
    { const char* e = getenv("AFP_EXPORT_MAX_UNITS"); if (e && atoi(e) >= 0) h->export_max_units = atoi(e); }
    { const char* e = getenv("AFP_NEARTIE_EPS"); if (e && atof(e) >= 0.0) h->nt_eps = atof(e); }
    h->init_compact_mode = h->compact_mode; h->init_compact_min_units = h->compact_min_units; h->init_seg_mode = h->seg_mode;
    h->init_seg_max_units = h->seg_max_units; h->init_seg_len = h->seg_len; h->init_seg_warm = h->seg_warm;
    // twiddles W_512^m = (cos, -sin)(2 pi m / 512), rounded from long double
    std::vector<double> tw(1024);
    for (int m = 0; m < 512; m++) {
        long double ang = -2.0L * 3.14159265358979323846264338327950288L * m / 512.0L;
        tw[2 * m] = (double)cosl(ang);
        tw[2 * m + 1] = (double)sinl(ang);
    }
    // half-log table: interval i of the frexp mantissa m in [0.5, 1) (AFP_LOGTAB_N intervals), centre c_i:
    // (0.5/c_i with 1/c_i rounded to double, -log(that double)/2 from long double)
    std::vector<double> lt(2 * AFP_LOGTAB_N);
    for (int i = 0; i < AFP_LOGTAB_N; i++) {
        const long double c = 0.5L + ((long double)i + 0.5L) / (2.0L * AFP_LOGTAB_N);
        const double invc = (double)(1.0L / c);
        lt[2 * i] = 0.5 * invc;
        lt[2 * i + 1] = (double)(-logl((long double)invc) / 2.0L);
    }
    if (ensure(h->d_tables, TAB_DOUBLES * sizeof(double)) != AFP_OK ||
        hipMemcpy((double*)h->d_tables.p + TAB_TWIDDLE, tw.data(), 1024 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy((double*)h->d_tables.p + TAB_LOGTAB, lt.data(), 2 * AFP_LOGTAB_N * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return AFP_ERR_HIP;
    }
    *out = h;
    return AFP_OK;
}

extern "C" void afp_destroy(afp_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)sync_handle(h);
    resolve_timings(h);
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : {h->ev_in, h->ev_a, h->ev_s, h->ev_b, h->ev_up_done}) if (e) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&h->d_tables, &h->d_gauss, &h->d_desc, &h->pcm_stage, &h->logS, &h->nyq,
                      &h->blk_part, &h->blk_corr, &h->stats, &h->cand_val,
                      &h->cand_bin, &h->masks, &h->pcnt, &h->ylast, &h->unit_mean, &h->sgram_dbg, &h->cvals, &h->lmask, &h->head,
                      &h->zcarry, &h->zflag, &h->cerr, &h->corr_list, &h->seg_desc, &h->seg_state, &h->seg_status, &h->seg_flag, &h->hpf_dump, &h->hslots, &h->hcnt,
                      &h->mslots, &h->mcnt, &h->hoffs, &h->poffs, &h->clip_tot, &h->unit_tot, &h->clip_hoff,
                      &h->unit_poff, &h->out_hashes, &h->out_peaks, &h->scan_prof, &h->lslots, &h->lcnt, &h->loffs,
                      &h->unit_ltot, &h->unit_loff, &h->out_landmarks, &h->in_peaks, &h->in_upo, &h->lm_in, &h->lm_out,
                      &h->tb_table, &h->tb_counts, &h->tb_newcnt, &h->tb_first, &h->tb_fill, &h->tb_seg, &h->tb_overflow,
                      &h->tb_misc, &h->tb_biglist, &h->tb_scan, &h->tb_pklen, &h->tb_pkoff, &h->tb_packed, &h->tb_olen, &h->tb_ooff, &h->tb_rows, &h->tb_off, &h->tb_ids, &h->tb_otable, &h->tb_ocounts, &h->tb_mlist,
                      &h->tb_mvals, &h->tb_mnv, &h->tb_patch, &h->gh_rows, &h->gh_nids, &h->gh_off,
                      &h->gh_hits, &h->vt_idcount, &h->vt_misc, &h->vt_ids, &h->vt_cnt, &h->vt_rank, &h->vt_hist, &h->vt_want, &h->vs_q, &h->vs_cursor, &h->vs_off, &h->vs_out};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    drain_retired(true);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_totals) (void)hipHostFree(h->h_totals);
    if (h->h_export) (void)hipHostFree(h->h_export);
    if (h->h_ovf) (void)hipHostFree(h->h_ovf);
    if (h->h_dl) (void)hipHostFree(h->h_dl);
    if (h->h_dlc) (void)hipHostFree(h->h_dlc);
    if (h->dlc_ev) (void)hipEventDestroy(h->dlc_ev);
    for (int i = 0; i < 4; i++) if (h->dl_ev[i]) (void)hipEventDestroy(h->dl_ev[i]);
    if (h->h_seg_stage) (void)hipHostFree(h->h_seg_stage);
    if (h->probe_stream) { (void)hipStreamSynchronize(h->probe_stream); (void)hipStreamDestroy(h->probe_stream); }
    if (h->probe_buf.p) (void)hipFree(h->probe_buf.p);
    if (h->tb_stream) { (void)hipStreamSynchronize(h->tb_stream); (void)hipStreamDestroy(h->tb_stream); }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

extern "C" int afp_set_stream(afp_handle* h, void* s)
{
    if (!h) return AFP_ERR_ARG;
    HIPCHK(sync_handle(h));
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return AFP_OK;
}

extern "C" int afp_set_stage_streams(afp_handle* h, void* spectral, void* scan, void* pair)
{
    if (!h || ((spectral == nullptr) != (scan == nullptr)) || (spectral && spectral == scan)) return AFP_ERR_ARG;
    if (pair && (!spectral || pair == spectral)) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(sync_handle(h));
    if (spectral && !h->ev_in) {
        HIPCHK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_s, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming));
    }
    h->stage_a = (hipStream_t)spectral;
    h->stage_b = (hipStream_t)scan;
    h->stage_c = pair ? (hipStream_t)pair : (hipStream_t)scan;
    return AFP_OK;
}

// A stream whose kernels run only on the compute units [first_cu, first_cu + n_cus) (hipExtStreamCreateWithCUMask; the
// mask bits are spread round-robin over the 8 XCDs, so a contiguous bit range is an even slice of every XCD).
extern "C" int afp_stream_create_cu_range(int device, int first_cu, int n_cus, void** stream)
{
    if (!stream || first_cu < 0 || n_cus < 1) return AFP_ERR_ARG;
    *stream = nullptr;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int ncu = prop.multiProcessorCount;
    if (first_cu + n_cus > ncu) return AFP_ERR_ARG;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = first_cu; i < first_cu + n_cus; i++) mask[(size_t)i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    *stream = (void*)s;
    return AFP_OK;
}
extern "C" int afp_stream_destroy(void* stream)
{
    if (!stream) return AFP_OK;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return AFP_OK;
}

// Page-locked host memory for callers that have no allocator of their own for it (a host without torch): PCM handed to
// afp_extract_host* from such a buffer is uploaded asynchronously by the copy engine (Extractor.submit), pageable memory
// goes through the runtime's staging copies.
extern "C" int afp_pinned_alloc(int device, int64_t bytes, void** out)
{
    if (!out || bytes <= 0) return AFP_ERR_ARG;
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return AFP_OK;
}
extern "C" int afp_pinned_free(void* p)
{
    if (p) HIPCHK(hipHostFree(p));
    return AFP_OK;
}

extern "C" int afp_set_workspace_limit(afp_handle* h, int64_t bytes)
{
    if (!h || bytes <= 0) return AFP_ERR_ARG;
    h->ws_limit = bytes;
    return AFP_OK;
}

extern "C" int afp_set_params(afp_handle* h, const afp_params* p)
{
    if (!h || !p || !p->window || !p->gauss) return AFP_ERR_ARG;
    if (p->maxpksperframe < 1 || p->maxpksperframe > AFP_MAX_PKS) return AFP_ERR_PARAM;
    if (p->maxpairsperpeak < 1 || p->maxpairsperpeak > 4096) return AFP_ERR_PARAM;
    if (p->nshifts < 1 || p->nshifts > AFP_MAX_SHIFTS) return AFP_ERR_PARAM;
    if (p->mindt < 0 || p->targetdt < 0 || p->targetdt > 1024 || p->targetdf < 0) return AFP_ERR_PARAM;
    if (!(p->a_dec > 0.0) || !(p->a_dec <= 1.0)) return AFP_ERR_PARAM;
    for (int s = 0; s < p->nshifts; s++)
        if (p->shift_offsets[s] < 0) return AFP_ERR_PARAM;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(sync_handle(h));
    ENSURE(h->d_gauss, AFP_NBINS * sizeof(double));
    HIPCHK(hipMemcpy((double*)h->d_tables.p + TAB_WINDOW, p->window, AFP_NFFT * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_gauss.p, p->gauss, AFP_NBINS * sizeof(double), hipMemcpyHostToDevice));
    h->prm = *p;
    h->prm.window = nullptr;
    h->prm.gauss = nullptr;
    h->have_params = true;
    return AFP_OK;
}

// ---- descriptor construction ----------------------------------------------------------------
static int frames_of(int64_t n) { return n > 0 ? (int)(1 + n / AFP_NHOP) : 0; }   // stft.py:33 after the 2x256 pad

struct UnitIn { int64_t pcm_off, n; int32_t T; };

// units of a PCM batch: unit = clip * S + shift (audfprint_analyze.py:369-377)
static int units_from_offsets(const afp_handle* h, const int64_t* off, int32_t nclips, std::vector<UnitIn>& units)
{
    if (nclips < 0 || (nclips > 0 && !off)) return AFP_ERR_ARG;
    const int S = h->prm.nshifts;
    if ((int64_t)nclips * S > 0x7fffffffLL) return AFP_ERR_ARG;
    units.resize((size_t)nclips * S);
    for (int c = 0; c < nclips; c++) {
        const int64_t n = off[c + 1] - off[c];
        // one unit's log-spectrogram rows are addressed with a 32-bit byte offset in k_scan (2 KB per frame): a clip is
        // limited to 2^21 - 64 frames (13.5 hours at 11025 Hz)
        if (n < 0 || n / AFP_NHOP >= (1 << 21) - 64) return AFP_ERR_ARG;
        for (int s = 0; s < S; s++) {
            const int64_t so = h->prm.shift_offsets[s];
            const int64_t nu = n - so > 0 ? n - so : 0;
            UnitIn& u = units[(size_t)c * S + s];
            u.n = nu;
            u.T = frames_of(nu);
            u.pcm_off = off[c] + (nu > 0 ? so : 0);
        }
    }
    return AFP_OK;
}

static void compute_geometry(const afp_handle* h, int32_t nclips, const std::vector<UnitIn>& units, Geometry& g)
{
    const int S = h->prm.nshifts;
    g.nclips = nclips; g.S = S; g.nunits = nclips * S;
    g.total_frames = g.total_mframes = g.nblk = g.ncblk = g.nmblk = g.npblk = 0;
    g.pch = S <= 2 ? 256 : S <= 4 ? 128 : S <= 8 ? 64 : 32;
    if (h->pairlane_ms && S > 1 && g.pch > h->pairlane_ms_pch) g.pch = h->pairlane_ms_pch;   // small column blocks: less LDS, more wavefronts per CU
    for (int c = 0; c < nclips; c++) {
        int Tmax = 0;
        for (int s = 0; s < S; s++) {
            const int T = units[(size_t)c * S + s].T;
            g.total_frames += T;
            g.nblk += (T + STFT_FPB - 1) / STFT_FPB;
            g.ncblk += (T + COL_CHUNK - 1) / COL_CHUNK;
            if (T > Tmax) Tmax = T;
        }
        g.total_mframes += Tmax;
        g.nmblk += (Tmax + COL_CHUNK - 1) / COL_CHUNK;
        g.npblk += (Tmax + g.pch - 1) / g.pch;
    }
}

// device bytes the batch would allocate ON THE PATH IT WOULD TAKE (the same rules as run_spectral / run_scan; an upper bound
// for the segment-parallel scan, whose cut is made later)
static int64_t workspace_bytes(const afp_handle* h, const Geometry& g, uint32_t flags)
{
    const int64_t K = h->prm.maxpksperframe, F = h->prm.maxpairsperpeak, S = g.S;
    const bool compact = (h->compact_mode == 1 || (h->compact_mode < 0 && g.nunits >= h->compact_min_units)) &&
                         !(flags & AFP_KEEP_DEBUG) && g.total_frames > 0;
    const bool seg = !compact && !(flags & AFP_KEEP_DEBUG) && (h->seg_mode == 1 || (h->seg_mode < 0 && g.nunits <= h->seg_max_units));
    int64_t b = 0;
    b += g.total_frames * (AFP_NBINS + 1) * 8;                 // logS + nyq (the compact path keeps them for the units that need the floor)
    if (compact)                                               // compact rows, masks, head rows, filter states
        b += g.total_frames * (CV_ROW * 8 + 32) + (int64_t)g.nunits * (CV_HEAD + 1) * AFP_NBINS * 8 + (2 * g.nblk + 64) * 4;
    if (seg) {
        // segments of at least 64 frames: threshold planes, k_hpf records (<= 4 per segment), descriptors, flags
        const int64_t nseg = g.total_frames / 64 + g.nunits;
        b += nseg * ((int64_t)SEG_NSTATE * AFP_NBINS * 8 + 4 * 2 * AFP_NBINS * 8 + (int64_t)sizeof(SegDesc) + 32 + AFP_NBINS * 8);
    }
    b += g.nblk * 7 * 8;                                       // partials, floor corrections
    b += g.total_frames * K * 12;                              // candidates
    b += g.total_frames * (32 + 4 + 4);                        // masks, pcnt, poffs
    if (flags & AFP_KEEP_DEBUG) b += g.total_frames * AFP_NBINS * 8;
    if (flags & AFP_WANT_HASHES) {
        b += g.total_frames * (K * F * 4 + 4);
        if (S > 1) b += g.total_mframes * (S * K * F * 4 + 4);
        b += g.total_mframes * 4;
    }
    b += (int64_t)g.nunits * (128 + AFP_NBINS * 8) + (int64_t)g.nblk * 8 + (int64_t)(g.ncblk + g.nmblk) * 8;
    return b;
}

extern "C" int64_t afp_workspace_bytes(afp_handle* h, const int64_t* off, int32_t nclips, uint32_t flags)
{
    if (!h || !h->have_params) return AFP_ERR_STATE;
    Geometry g;
    std::vector<UnitIn> units;
    int r = units_from_offsets(h, off, nclips, units);
    if (r != AFP_OK) return r;
    compute_geometry(h, nclips, units, g);
    return workspace_bytes(h, g, flags);
}

template <typename T>
static T* carve(char*& cur, size_t count)
{
    T* p = reinterpret_cast<T*>(cur);
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    cur += bytes;
    return p;
}

static int build_descriptors(afp_handle* h, const std::vector<UnitIn>& units, const Geometry& g)
{
    h->desc_valid = false;
    h->seg_cache_ok = false;
    const size_t nu = g.nunits, nc = g.nclips;
    size_t total = 0;
    auto add = [&](size_t count, size_t sz) { total += (count * sz + 255) & ~(size_t)255; };
    add(nu, 8); add(nu, 8); add(nu, 8); add(nu + 1, 8); add(nu, 4);
    add(g.nblk, 4); add(g.nblk, 4); add(g.nblk, 8); add(g.nblk, 8); add(nu, sizeof(UnitDesc)); add(g.ncblk, 4); add(g.ncblk, 4);
    add(nc, 8); add(nc, 4); add(g.nmblk, 4); add(g.nmblk, 4); add(g.npblk, 4); add(g.npblk, 4);
    total += 256;
    HIPCHK(sync_handle(h));      // the staging buffer may still feed a copy in flight
    if (total > h->h_stage_cap) {
        if (h->h_stage) (void)hipHostFree(h->h_stage);
        h->h_stage = nullptr; h->h_stage_cap = 0;
        HIPCHK(hipHostMalloc(&h->h_stage, total, hipHostMallocDefault));
        h->h_stage_cap = total;
    }
    ENSURE(h->d_desc, total);
    char* hc = (char*)h->h_stage;
    char* dc = (char*)h->d_desc.p;
#define CARVE(T, name, count)                \
    T* hp_##name = carve<T>(hc, count);      \
    h->name = carve<T>(dc, count);
    CARVE(int64_t, unit_pcm_off, nu)
    CARVE(int64_t, unit_n, nu)
    CARVE(int64_t, unit_fbase, nu)
    CARVE(int64_t, unit_bbase, nu + 1)
    CARVE(int32_t, unit_T, nu)
    CARVE(int32_t, blk_unit, g.nblk)
    CARVE(int32_t, blk_t0, g.nblk)
    CARVE(ChunkDesc, blk2, g.nblk)
    CARVE(ChunkDesc, tblk2, g.nblk)
    CARVE(UnitDesc, udesc, nu)
    CARVE(int32_t, cblk_unit, g.ncblk)
    CARVE(int32_t, cblk_t0, g.ncblk)
    CARVE(int64_t, clip_mfbase, nc)
    CARVE(int32_t, clip_T0, nc)
    CARVE(int32_t, mblk_clip, g.nmblk)
    CARVE(int32_t, mblk_t0, g.nmblk)
    CARVE(int32_t, pblk_clip, g.npblk)
    CARVE(int32_t, pblk_t0, g.npblk)
#undef CARVE
    int64_t fb = 0, bb = 0, cb = 0, mfb = 0, mb = 0, pb = 0;
    for (int c = 0; c < g.nclips; c++) {
        int Tmax = 0;
        for (int s = 0; s < g.S; s++) {
            const int u = c * g.S + s;
            const UnitIn& ui = units[u];
            const int T = ui.T;
            hp_unit_pcm_off[u] = ui.pcm_off;
            hp_unit_n[u] = ui.n;
            hp_unit_T[u] = T;
            hp_unit_fbase[u] = fb;
            hp_unit_bbase[u] = bb;
            { UnitDesc& d = hp_udesc[u]; d.pcm_off = ui.pcm_off; d.n = ui.n; d.fbase = fb; d.bbase = bb; d.T = T; d.pad = 0; }
            for (int t0 = 0; t0 < T; t0 += STFT_FPB) { hp_blk_unit[bb] = u; hp_blk_t0[bb] = t0; hp_blk2[bb].unit = u; hp_blk2[bb].t0 = t0; bb++; }
            for (int t0 = 0; t0 < T; t0 += COL_CHUNK) { hp_cblk_unit[cb] = u; hp_cblk_t0[cb] = t0; cb++; }
            fb += T;
            if (T > Tmax) Tmax = T;
        }
        hp_clip_mfbase[c] = mfb;
        hp_clip_T0[c] = Tmax;                      // merged frames of the clip = longest shift
        for (int t0 = 0; t0 < Tmax; t0 += COL_CHUNK) { hp_mblk_clip[mb] = c; hp_mblk_t0[mb] = t0; mb++; }
        for (int t0 = 0; t0 < Tmax; t0 += g.pch) { hp_pblk_clip[pb] = c; hp_pblk_t0[pb] = t0; pb++; }
        mfb += Tmax;
    }
    hp_unit_bbase[nu] = bb;
    h->unit_T_host.resize(nu);
    for (size_t u = 0; u < nu; u++) h->unit_T_host[u] = units[u].T;
    {
        // the same chunks time-major: chunk k of every unit that has one, then chunk k + 1 (a counting sort by k: units
        // stay in ascending order inside a time step)
        int Tmax_all = 0;
        for (size_t u = 0; u < nu; u++) if (units[u].T > Tmax_all) Tmax_all = units[u].T;
        const int nk = (Tmax_all + STFT_FPB - 1) / STFT_FPB;
        std::vector<int64_t> start((size_t)nk + 1, 0);
        for (size_t u = 0; u < nu; u++) { const int ck = (units[u].T + STFT_FPB - 1) / STFT_FPB; if (ck > 0) start[(size_t)ck]++; }
        // start[k] = units with exactly k chunks -> units alive at step k = sum over ck > k
        std::vector<int64_t> alive((size_t)nk + 1, 0);
        { int64_t acc = 0; for (int k = nk; k >= 1; k--) { acc += start[(size_t)k]; alive[(size_t)k - 1] = acc; } }
        std::vector<int64_t> pos((size_t)nk + 1, 0);
        for (int k = 0; k < nk; k++) pos[(size_t)k + 1] = pos[(size_t)k] + alive[(size_t)k];
        std::vector<int64_t> seg0(pos.begin(), pos.end());       // first entry of every time step
        for (size_t u = 0; u < nu; u++) {
            const int ck = (units[u].T + STFT_FPB - 1) / STFT_FPB;
            for (int k = 0; k < ck; k++) { const int64_t i = pos[(size_t)k]++; hp_tblk2[i].unit = (int32_t)u; hp_tblk2[i].t0 = k * STFT_FPB; }
        }
        // Several shifts (audfprint_analyze.py:369-377): the S units of a clip read the SAME samples, 64 s' apart.  Workgroup i
        // of a 1-D grid runs on XCD i mod 8, each XCD has its own L2 -- in unit order the S shifts of a clip's chunk land on S
        // different XCDs and the PCM is fetched from HBM once per shift grid (C5, r03 counters: k_stft FETCH 5.58 GB for
        // 1.355 GB of samples).  Inside a time step (any order inside a step keeps the hand-off's "predecessor has a smaller
        // index") the entries are therefore re-ordered in blocks of 8 clips: [shift][clip] -- the shifts of one clip sit 8
        // entries apart, on the same XCD, within 8 S consecutive dispatches, and find their rows in that XCD's L2.
        static const bool xcd_order = !(getenv("AFP_XCD_ORDER") && getenv("AFP_XCD_ORDER")[0] == '0');      // (A/B switch)
        if (g.S > 1 && xcd_order) {
            const int S = g.S, NX = 8;
            std::vector<ChunkDesc> tmp;
            for (int k = 0; k < nk; k++) {
                const int64_t a = seg0[(size_t)k], b = pos[(size_t)k];           // [a, b): the step's entries, units ascending
                if (b - a < 2) continue;
                tmp.assign(hp_tblk2 + a, hp_tblk2 + b);
                int64_t w = a;
                size_t i = 0;
                while (i < tmp.size()) {
                    const int c0 = tmp[i].unit / S;
                    size_t j = i;                                                // the entries of clips [c0, c0 + NX)
                    while (j < tmp.size() && tmp[j].unit / S < c0 + NX) j++;
                    for (int sft = 0; sft < S; sft++)
                        for (size_t e = i; e < j; e++)
                            if (tmp[e].unit % S == sft) hp_tblk2[w++] = tmp[e];
                    i = j;
                }
            }
        }
    }
    HIPCHK(hipMemcpyAsync(h->d_desc.p, h->h_stage, total, hipMemcpyHostToDevice, h->stream));
    return AFP_OK;
}

static void adopt_geometry(afp_handle* h, const Geometry& g, uint32_t flags)
{
    h->nclips = g.nclips; h->nunits = g.nunits; h->S = g.S;
    h->total_frames = g.total_frames; h->total_mframes = g.total_mframes;
    h->nblk = g.nblk; h->ncblk = g.ncblk; h->nmblk = g.nmblk;
    h->flags = flags;
    h->total_hashes = h->total_peaks = h->total_landmarks = 0;
    h->K = h->prm.maxpksperframe;
}

// ---- stage runners ----------------------------------------------------------------------------
// spectral stage: PCM -> log|S| -> per-unit stats -> floor correction
// `st`: the spectral-stage stream (the STFT itself); `st2`: where the short kernels behind it go (per-unit statistics, the dense
// re-transform of the units that need the floor, the floor correction).  In staged mode that is the SCAN-stage stream --
// they only have to precede this batch's scan, and on the spectral stream they would sit between two batches' STFTs
// (0.07 ms of a 1.5 ms step with nothing else to run beside the previous scan); `ev_mid` orders them behind the STFT.
static int run_spectral(afp_handle* h, const void* d_pcm, int s16, const Geometry& g, uint32_t flags, hipStream_t st,
                        hipStream_t st2, hipEvent_t ev_mid)
{
    const int64_t TF = g.total_frames;
    const int K = h->prm.maxpksperframe;
    h->tstream = st;
    // COMPACT pipeline (k_stft<ST, true> -> k_scan_c): the log-spectrogram stays on chip; needs enough units that a unit's
    // next chunk is dispatched about one residency after the previous one (chunks are listed time-major, k_stft.hip)
    const bool compact = h->compact_mode == 1 || (h->compact_mode < 0 && g.nunits >= h->compact_min_units);
    h->batch_compact = compact && !(flags & AFP_KEEP_DEBUG) && TF > 0;
    ENSURE(h->logS, TF * AFP_NBINS * 8);
    ENSURE(h->nyq, TF * 8);
    ENSURE(h->blk_part, 6 * g.nblk * 8);         // the six partial arrays, back to back
    ENSURE(h->blk_corr, g.nblk * 8);
    ENSURE(h->stats, (int64_t)g.nunits * sizeof(UnitStats));
    ENSURE(h->cand_val, TF * K * 8);
    ENSURE(h->cand_bin, TF * K * 4);
    ENSURE(h->masks, TF * 32);
    ENSURE(h->pcnt, TF * 4);
    ENSURE(h->unit_mean, (int64_t)g.nunits * 8);
    ENSURE(h->ylast, (int64_t)g.nunits * AFP_NBINS * 8);
    if (flags & AFP_KEEP_DEBUG) ENSURE(h->sgram_dbg, TF * AFP_NBINS * 8);
    if (h->batch_compact) {
        ENSURE(h->cvals, (TF * CV_ROW + 64) * 8);           // (+ slack: the reader's second load may run one value past a full row)
        ENSURE(h->lmask, TF * 32);
        ENSURE(h->head, (int64_t)g.nunits * CV_HEAD * AFP_NBINS * 8);
        ENSURE(h->zcarry, (int64_t)g.nunits * AFP_NBINS * 8);
        if ((size_t)g.nunits * 8 > h->zflag.cap || !h->zflag.p) {
            ENSURE(h->zflag, (int64_t)g.nunits * 8);
            HIPCHK(hipMemsetAsync(h->zflag.p, 0, h->zflag.cap, st));      // flags carry the launch epoch: cleared once
        }
        if (!h->cerr.p) { ENSURE(h->cerr, 256); HIPCHK(hipMemsetAsync(h->cerr.p, 0, 256, st)); }
        ENSURE(h->corr_list, (2 * g.nblk + 64) * 4);        // [0] counter, [64..] units, then first frames
    }
    if (TF > 0) {
        StftArgs a;
        memset(&a, 0, sizeof(a));
        a.pcm = d_pcm;                       // clip offsets are absolute sample indices into d_pcm
        a.pcm_is_s16 = s16;                 // 0 float32, 1 int16, 2 float64
        a.units = h->udesc; a.blk = h->blk2;
        a.tables = (const double*)h->d_tables.p;
        a.logS = (double*)h->logS.p; a.nyq = (double*)h->nyq.p;
        a.blk_part = (double*)h->blk_part.p; a.part_stride = g.nblk;
        a.masks = (uint64_t*)h->masks.p; a.cand_bin = (int32_t*)h->cand_bin.p; a.K = K;
        a.pole = h->prm.hpf_pole;
        if (h->batch_compact) {
            StftArgs c = a;
            c.blk = h->tblk2;
            c.cvals = (double*)h->cvals.p; c.lmask = (uint64_t*)h->lmask.p; c.head = (double*)h->head.p;
            c.ylast = (double*)h->ylast.p; c.zcarry = (double*)h->zcarry.p; c.zflag = (unsigned long long*)h->zflag.p;
            c.epoch = ++h->epoch; c.err = (int32_t*)h->cerr.p; c.list_zero = (int32_t*)h->corr_list.p;
            c.spin_limit = h->compact_force_timeout ? (1 << 10) : (1 << 18);
            c.skip_unit = h->compact_force_timeout ? 0 : -1; c.skip_chunk = 0;
            { Timed t(h, KS_STFT); afp_launch_stft_compact(&c, (int)g.nblk, st); }
        } else {
            Timed t(h, KS_STFT);
            afp_launch_stft(&a, (int)g.nblk, st);
        }
        if (st2 != st) { HIPCHK(hipEventRecord(ev_mid, st)); HIPCHK(hipStreamWaitEvent(st2, ev_mid, 0)); h->tstream = st2; st = st2; }
        StatsArgs sa;
        sa.unit_T = h->unit_T; sa.unit_bbase = h->unit_bbase;
        sa.blk_pmax = (const double*)h->blk_part.p; sa.blk_lmin = sa.blk_pmax + g.nblk;
        sa.blk_lsum = sa.blk_pmax + 2 * g.nblk; sa.blk_flat = sa.blk_pmax + 3 * g.nblk; sa.part_stride = g.nblk;
        sa.stats = (UnitStats*)h->stats.p; sa.nunits = g.nunits;
        sa.corr_cnt = nullptr; sa.corr_list = nullptr;
        CorrArgs ca;
        ca.unit_T = h->unit_T; ca.unit_fbase = h->unit_fbase; ca.blk_unit = h->blk_unit; ca.blk_t0 = h->blk_t0;
        ca.unit_bbase = h->unit_bbase; ca.nunits = g.nunits;
        ca.blk_lmin = (const double*)h->blk_part.p + g.nblk; ca.stats = (const UnitStats*)h->stats.p;
        ca.logS = (const double*)h->logS.p; ca.nyq = (const double*)h->nyq.p; ca.blk_corr = (double*)h->blk_corr.p;
        if (h->batch_compact) {
            sa.corr_cnt = (int32_t*)h->corr_list.p; sa.corr_list = (ChunkDesc*)(sa.corr_cnt + 64);
            { Timed t(h, KS_STATS); afp_launch_unit_stats(&sa, st); }
            // units with values under the floor max|S|/1e6 (UNIT_CORR, known now) go through the dense kernels: the dense
            // STFT again for their chunks only (k_unit_stats listed them; on noise about 1 % of the units -- a DC or
            // Nyquist bin close to zero)
            a.list_cnt = sa.corr_cnt; a.list = sa.corr_list;
            { Timed t(h, KS_CORR); afp_launch_stft_list(&a, (int)std::min<int64_t>(g.nblk, 2048), st); }
            { Timed t(h, KS_CORR); afp_launch_floor_corr(&ca, (int)g.nblk, st); }
        } else {
            // per-unit statistics and the floor correction in ONE launch (k_stats_corr)
            Timed t(h, KS_STATS);
            afp_launch_stats_corr(&sa, &ca, (int)g.nblk, st);
        }
    } else {
        if (st2 != st) { HIPCHK(hipEventRecord(ev_mid, st)); HIPCHK(hipStreamWaitEvent(st2, ev_mid, 0)); h->tstream = st2; st = st2; }
        StatsArgs sa;
        sa.unit_T = h->unit_T; sa.unit_bbase = h->unit_bbase;
        sa.blk_pmax = (const double*)h->blk_part.p; sa.blk_lmin = sa.blk_pmax + g.nblk;
        sa.blk_lsum = sa.blk_pmax + 2 * g.nblk; sa.blk_flat = sa.blk_pmax + 3 * g.nblk; sa.part_stride = g.nblk;
        sa.stats = (UnitStats*)h->stats.p; sa.nunits = g.nunits;
        sa.corr_cnt = nullptr; sa.corr_list = nullptr;
        Timed t(h, KS_STATS);
        afp_launch_unit_stats(&sa, st);
    }
    return AFP_OK;
}

// scan stage, first half: log|S| -> onset filter -> decaying-threshold peak pick (masks, pcnt)
static int run_scan(afp_handle* h, const Geometry& g, uint32_t flags, hipStream_t st)
{
    const int64_t TF = g.total_frames;
    const int K = h->prm.maxpksperframe;
    h->tstream = st;
    if (TF > 0) {
        ScanArgs s;
        memset(&s, 0, sizeof(s));
        s.unit_T = h->unit_T; s.unit_fbase = h->unit_fbase; s.unit_bbase = h->unit_bbase;
        s.stats = (const UnitStats*)h->stats.p; s.blk_corr = (const double*)h->blk_corr.p;
        s.logS = (const double*)h->logS.p; s.gauss = (const double*)h->d_gauss.p;
        s.a_dec = h->prm.a_dec; s.pole = h->prm.hpf_pole; s.K = K;
        s.cand_val = (double*)h->cand_val.p; s.cand_bin = (int32_t*)h->cand_bin.p;
        s.masks = (uint64_t*)h->masks.p; s.ylast = (double*)h->ylast.p; s.unit_mean = (double*)h->unit_mean.p;
        s.sgram_dbg = (flags & AFP_KEEP_DEBUG) ? (double*)h->sgram_dbg.p : nullptr;
        s.prof = nullptr; s.raw_rows = 0; s.fwd_off = 0;
        s.cvals = (const double*)h->cvals.p; s.lmask = (const uint64_t*)h->lmask.p; s.head = (const double*)h->head.p;
        // AFP_SCAN_PROF=1: cycle stamps of the scanner wave (tap 5) on the production configuration (no debug spectrogram)
        static const bool prof_env = getenv("AFP_SCAN_PROF") != nullptr;
        if ((flags & AFP_KEEP_DEBUG) || prof_env) { ENSURE(h->scan_prof, (int64_t)g.nunits * 256); s.prof = (unsigned long long*)h->scan_prof.p; }
        s.segs = nullptr; s.seg_state = nullptr; s.seg_status = nullptr; s.nseg = 0; s.seg_W = 0; s.seg_phase = 0; s.seg_repair = 0;
        s.only_if = nullptr; s.only_if_unit = nullptr; s.clear_all = 0; s.seg_ufail = nullptr; s.seg_rerun = nullptr; s.seg_force_fail = 0; s.seg_flag = nullptr; s.seg_ufirst = nullptr;
        // near-tie guard: the scanners mark units whose decisive comparisons were closer than nt_eps and count them in cerr[2]
        s.nt_eps = h->nt_eps; s.stats_rw = (UnitStats*)h->stats.p; s.nt_count = nullptr;
        if (h->nt_eps > 0.0) {
            if (!h->cerr.p) { ENSURE(h->cerr, 256); HIPCHK(hipMemsetAsync(h->cerr.p, 0, 256, st)); }
            HIPCHK(hipMemsetAsync((int32_t*)h->cerr.p + 2, 0, 4, st));
            s.nt_count = (int32_t*)h->cerr.p + 2;
        }
        // Few long units (a single file): cut the scan into segments with a warm-up (SegDesc, afp_common.h).  The threshold
        // decays by a_dec per frame; the warm-up is a few decay lengths.
        h->batch_seg = false; h->batch_nseg = 0;
        const double decay = 1.0 - h->prm.a_dec;
        const bool seg_want = h->seg_mode == 1 || (h->seg_mode < 0 && g.nunits <= h->seg_max_units);
        if (seg_want && !h->batch_compact && !(flags & AFP_KEEP_DEBUG) && !s.prof && decay > 1e-4 && h->unit_T_host.size() == (size_t)g.nunits) {
            // Warm-up: 0.625 decay lengths (129 frames at density 20).  tools/seg_convergence.py (numpy oracle): a pass started
            // in mid-clip holds the sequential pass's state bit for bit after a median of 15-57 frames and at most 0.57
            // decay lengths (117 frames at density 20) over noise / tonal x density 20 / 70; a boundary that has NOT
            // converged only costs a re-run of its segment by the chain launch.  tools/analyzer_breakdown.py, one noise clip
            // per call, (segment, warm-up) = (104, 205) -> (64, 128): 10 s 0.400 -> 0.339 ms (a 431-frame file now has room
            // for segments), 30 s 0.479 -> 0.420, 60 s 0.594 -> 0.546, 300 s 1.37 -> 1.29, no re-runs; (48, 96) and below
            // start to re-run segments on the longer clips and lose what they gain.
            int W = h->seg_warm > 0 ? h->seg_warm : (int)std::min(4096.0, std::max(64.0, ceil(0.625 / decay)));
            int S = h->seg_len > 0 ? h->seg_len : std::max(64, (W / 2 + 7) & ~7);
            if (TF / S > 8192) S = (int)((TF + 8191) / 8192);
            h->batch_short_cut = false;
            {   // k_hpf keeps a unit's listed frames (four per segment) in LDS
                int longest_T = 0;
                for (int u = 0; u < g.nunits; u++) longest_T = std::max(longest_T, h->unit_T_host[(size_t)u]);
                // SHORT FILES (r05, tools/seg_cut_sweep.py -> profiles/r05_seg_cut_sweep.jsonl): a file of up to ~23 s scans
                // three times its frames with the standard cut (64 own + 128 warm-up frames per segment, four launches as long
                // as the longest segment).  (32, 96) converges just as well on noise (0 re-runs over 10 / 20 s x 3 seeds) and
                // costs 15 % less per call (10 s: 0.228 -> 0.195 ms); on a gated tonal clip -- thresholds that remember a loud
                // passage for hundreds of frames -- it re-runs 8 of 14 segments and costs 15 % MORE (0.290 -> 0.334).  So the
                // short cut is tried, and a batch that re-ran more than 5 % of its segments sends the next 32 batches of this
                // handle back to the standard cut (finalize()).  Either cut is bit-exact: the boundary check + chain repair
                // see to that; the choice is only about time.  Only for the standard cut of the default density.
                if (h->seg_adapt && h->seg_len <= 0 && h->seg_warm <= 0 && (W / S) * S == 128 && S == 64 && longest_T <= 1000) {
                    if (h->seg_short_penalty > 0) h->seg_short_penalty--;
                    else { S = 32; W = 96; h->batch_short_cut = true; h->seg_short_total++; }
                }
                const int smin = (int)(((int64_t)longest_T * 4 + HPF_MAX_DUMPS - 9) / (HPF_MAX_DUMPS - 8));
                if (S < smin) S = smin;
            }
            // a warm-up that is a multiple of the segment length puts every frame k_hpf has to record (segment starts, starts
            // - W, ends + W) ON a segment start: one listed frame per segment instead of three, and k_hpf's filter wavefront
            // runs every phase as straight-line code (129 -> 128 at density 20: k_hpf 274 -> 218 -> 169 us on a 300 s clip)
            if (h->seg_warm <= 0 && W > S) W = (W / S) * S;
            std::vector<SegDesc>& sv = h->seg_host;
            // frames at which k_hpf leaves the filter state, per unit (ascending, unique): dz_* / dy_* index them
            std::vector<int32_t>& doff = h->seg_doff;
            std::vector<int32_t>& dfr = h->seg_dfr;
            std::vector<int32_t>& ufirst = h->seg_ufirst_host;
            // the same batch shape and the same cut as last time (steady-state ingest): the device image is still valid
            const bool reuse = h->seg_cache_ok && h->desc_cached && h->seg_cache_W == W && h->seg_cache_S == S;
            int longest = h->seg_cache_longest;
            if (!reuse) {
            h->seg_cache_ok = false;
            sv.clear();
            doff.assign((size_t)g.nunits + 1, 0);
            dfr.clear();
            ufirst.assign((size_t)g.nunits + 1, 0);
            longest = 0;
            for (int u = 0; u < g.nunits; u++) {
                const int T = h->unit_T_host[(size_t)u];
                if (T > longest) longest = T;
                const int n = (T + S - 1) / S;
                const size_t first = sv.size();
                std::vector<int32_t> fr;
                for (int k = 0; k < n; k++) {
                    SegDesc d;
                    d.unit = u; d.s = k * S; d.e = std::min(T, (k + 1) * S);
                    d.prev = k > 0 ? (int)sv.size() - 1 : -1;
                    d.next = k + 1 < n ? (int)sv.size() + 1 : -1;
                    d.dz_fwd = d.dz_rep = d.dy_bwd = d.dy_rep = -1; d.pad = 0;
                    if (d.prev >= 0) { fr.push_back(std::max(0, d.s - W)); fr.push_back(d.s); }
                    fr.push_back(d.next >= 0 ? std::min(T, d.e + 1 + W) - 1 : d.e - 1);
                    if (d.next >= 0) fr.push_back(d.e);
                    sv.push_back(d);
                }
                std::sort(fr.begin(), fr.end());
                fr.erase(std::unique(fr.begin(), fr.end()), fr.end());
                const int base = (int)dfr.size();
                auto at = [&](int f) { return base + (int)(std::lower_bound(fr.begin(), fr.end(), f) - fr.begin()); };
                for (size_t i = first; i < sv.size(); i++) {
                    SegDesc& d = sv[i];
                    if (d.prev >= 0) { const int tb = std::max(0, d.s - W); d.dz_fwd = tb > 0 ? at(tb) : -1; d.dz_rep = at(d.s); }
                    d.dy_bwd = at(d.next >= 0 ? std::min(T, d.e + 1 + W) - 1 : d.e - 1);
                    if (d.next >= 0) d.dy_rep = at(d.e);
                }
                dfr.insert(dfr.end(), fr.begin(), fr.end());
                doff[(size_t)u + 1] = (int32_t)dfr.size();
                ufirst[(size_t)u + 1] = (int32_t)sv.size();
            }
            }
            if (longest > 2 * (S + W) && !sv.empty()) {        // (a short unit gains nothing: the segments cost launches and warm-up)
                const int nseg = (int)sv.size();
                const size_t ndump = dfr.size();
                auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
                const size_t o_idx = al((size_t)nseg * sizeof(SegDesc));
                const size_t o_uf = o_idx + al((doff.size() + ndump) * 4);
                const size_t pack_bytes = o_uf + al(((size_t)g.nunits + 1) * 4);
                const size_t z_uf = 256, z_rr = z_uf + al((size_t)g.nunits * 4), zero_bytes = z_rr + al((size_t)nseg * 8);
                ENSURE(h->seg_desc, (int64_t)pack_bytes);
                ENSURE(h->seg_state, (int64_t)SEG_NSTATE * nseg * AFP_NBINS * 8);
                ENSURE(h->seg_status, (int64_t)zero_bytes);
                ENSURE(h->seg_flag, (int64_t)nseg * 4);
                ENSURE(h->hpf_dump, (int64_t)(ndump + 1) * 2 * AFP_NBINS * 8);
                ENSURE(h->ylast, (int64_t)std::max(nseg, g.nunits) * AFP_NBINS * 8);
                s.ylast = (double*)h->ylast.p;
                h->hpf_idx_p = (int32_t*)((char*)h->seg_desc.p + o_idx);
                h->seg_ufirst_p = (int32_t*)((char*)h->seg_desc.p + o_uf);
                h->seg_ufail_p = (int32_t*)((char*)h->seg_status.p + z_uf);
                h->seg_rerun_p = (int32_t*)((char*)h->seg_status.p + z_rr);
                if (!reuse) {
                    // (descriptors built anew this call: build_descriptors has waited for everything that could still read
                    //  the staging image; a changed cut over cached descriptors has not)
                    if (h->desc_cached) HIPCHK(sync_handle(h));
                    if (pack_bytes > h->h_seg_stage_cap) {
                        if (h->h_seg_stage) (void)hipHostFree(h->h_seg_stage);
                        h->h_seg_stage = nullptr; h->h_seg_stage_cap = 0;
                        HIPCHK(hipHostMalloc((void**)&h->h_seg_stage, pack_bytes + pack_bytes / 2, hipHostMallocDefault));
                        h->h_seg_stage_cap = pack_bytes + pack_bytes / 2;
                    }
                    memcpy(h->h_seg_stage, sv.data(), (size_t)nseg * sizeof(SegDesc));
                    memcpy(h->h_seg_stage + o_idx, doff.data(), doff.size() * 4);
                    if (ndump) memcpy(h->h_seg_stage + o_idx + doff.size() * 4, dfr.data(), ndump * 4);
                    memcpy(h->h_seg_stage + o_uf, ufirst.data(), ((size_t)g.nunits + 1) * 4);
                    HIPCHK(hipMemcpyAsync(h->seg_desc.p, h->h_seg_stage, pack_bytes, hipMemcpyHostToDevice, st));
                    h->seg_cache_ok = true; h->seg_cache_W = W; h->seg_cache_S = S; h->seg_cache_longest = longest;
                }
                // (all zero already when the previous batch of this handle ended in k_export / k_finish_one, which clear the block
                //  behind themselves)
                const bool clean = h->seg_clean_ptr == h->seg_status.p && h->seg_clean_bytes >= zero_bytes;
                if (!clean) HIPCHK(hipMemsetAsync(h->seg_status.p, 0, zero_bytes, st));
                h->seg_clean_keep = clean ? h->seg_clean_bytes : zero_bytes;      // what is zero again once [0, zero_bytes) has been cleared
                h->seg_zero_bytes = zero_bytes;
                h->seg_clean_ptr = nullptr; h->seg_clean_bytes = 0;              // dirty from here on (until an export clears it)
                s.segs = (const SegDesc*)h->seg_desc.p; s.seg_state = (double*)h->seg_state.p;
                s.seg_status = (int32_t*)h->seg_status.p; s.seg_ufail = h->seg_ufail_p; s.nseg = nseg; s.seg_W = W;
                s.seg_rerun = h->seg_rerun_p; s.seg_force_fail = h->seg_force_fail;
                s.seg_flag = (int32_t*)h->seg_flag.p; s.seg_ufirst = h->seg_ufirst_p;
                s.hpf_dump = (const double*)h->hpf_dump.p;
                h->batch_seg = true; h->batch_nseg = nseg;
                h->seg_ndoff = (int)doff.size();
            }
        }
        if (h->batch_seg) {
            Timed t(h, KS_SCAN);
            HpfArgs ha;
            ha.unit_T = h->unit_T; ha.unit_fbase = h->unit_fbase; ha.unit_bbase = h->unit_bbase;
            ha.stats = (const UnitStats*)h->stats.p; ha.blk_corr = (const double*)h->blk_corr.p;
            ha.logS = (const double*)h->logS.p; ha.pole = h->prm.hpf_pole;
            ha.dump_off = h->hpf_idx_p; ha.dump_frame = ha.dump_off + h->seg_ndoff;
            ha.dump_state = (double*)h->hpf_dump.p; ha.fail = (int32_t*)h->seg_status.p + 3;
            ha.prof = nullptr;
            static const bool hpf_prof = getenv("AFP_HPF_PROF") != nullptr;      // (measurement aid: debug tap 6)
            if (hpf_prof) { ENSURE(h->scan_prof, 2048 * 4 * 8); HIPCHK(hipMemsetAsync(h->scan_prof.p, 0, 2048 * 4 * 8, st)); ha.prof = (unsigned long long*)h->scan_prof.p; }
            afp_launch_hpf(&ha, g.nunits, st);                     // the onset-filter state at the frames the segments start from
            for (int phase = SEG_FWD; phase <= SEG_BWD; phase++) {
                s.seg_phase = phase;
                s.seg_repair = 0; afp_launch_scan_seg(&s, g.nunits, st);
                s.seg_repair = 1; afp_launch_scan_seg(&s, g.nunits, st);     // runs of segments whose warm-up did not reach the true state
            }
            afp_launch_seg_verify(&s, st);
            // units with a boundary that still does not meet: the sequential kernel over their rows, overwriting everything
            ScanArgs f = s;
            f.segs = nullptr; f.nseg = 0; f.only_if = s.seg_status; f.only_if_unit = s.seg_ufail; f.clear_all = 1; f.hpf_dump = nullptr;
            afp_launch_scan(&f, g.nunits, st);
        } else {
            Timed t(h, KS_SCAN);
            // k_scan writes only non-empty records; k_stft pre-filled "no candidate" / "no peak"
            // enough units to share CUs with the next batch's k_stft: the 8 KB-of-LDS variant (four scan workgroups
            // then leave room for three STFT workgroups per CU); few units (a single file): the 2-frame ring,
            // which is ~9 % faster on its own
            const bool small = h->scan_lds_mode == 1 || (h->scan_lds_mode == 0 && g.nunits >= 256 && !(flags & AFP_KEEP_DEBUG));
            static const int dummy_us = getenv("AFP_SCAN_DUMMY") ? atoi(getenv("AFP_SCAN_DUMMY")) : 0;      // (measurement aid, results void)
            if (h->batch_compact && dummy_us > 0) afp_launch_scan_dummy(g.nunits, dummy_us, (double*)h->unit_mean.p, st);
            else if (h->batch_compact) afp_launch_scan_compact(&s, g.nunits, st);      // (units that needed the floor take the dense path inside)
            else if (small) afp_launch_scan_small(&s, g.nunits, st);
            else afp_launch_scan(&s, g.nunits, st);
        }
    }
    return AFP_OK;
}

// back: masks -> pairs -> hashes (sorted unique per clip, CSR) / landmarks (per unit, CSR) / peak lists
static int run_back(afp_handle* h, const Geometry& g, uint32_t flags, hipStream_t st)
{
    const int64_t TF = g.total_frames;
    // peaks a column can hold: the scan's K, or more when the peak lists came from the caller
    const int K = h->pair_K > 0 ? h->pair_K : h->prm.maxpksperframe;
    const int F = h->prm.maxpairsperpeak, S = g.S;
    h->tstream = st;
    if (!h->h_totals) HIPCHK(hipHostMalloc((void**)&h->h_totals, 8 * sizeof(int64_t), hipHostMallocDefault));
    h->h_totals[0] = h->h_totals[1] = h->h_totals[2] = 0;
    h->have_sh = h->have_sp = h->have_sl = false;
    if (TF <= 0) return AFP_OK;
    const int slot = K * F;
    PairArgs pa;
    pa.unit_T = h->unit_T; pa.unit_fbase = h->unit_fbase; pa.cblk_unit = h->cblk_unit; pa.cblk_t0 = h->cblk_t0;
    pa.masks = (const uint64_t*)h->masks.p;
    pa.lds_lists = slot <= 48 ? 1 : 0;
    pa.slot = slot; pa.fanout = F; pa.targetdf = h->prm.targetdf; pa.mindt = h->prm.mindt; pa.targetdt = h->prm.targetdt;
    PairRowsArgs pr;                                  // list-order peak lists (afp_pairs_from_peaks): k_pair_rows reads the rows
    pr.rows = (const int32_t*)h->in_peaks.p; pr.upo = (const int64_t*)h->in_upo.p;
    pr.rcnt = (const int32_t*)h->pcnt.p; pr.roffs = (const int32_t*)h->poffs.p;

    if (flags & AFP_WANT_LANDMARKS) {
        // raw landmarks in the reference's nested emission order (audfprint_analyze.py:328-341), per unit
        ENSURE(h->lslots, TF * (int64_t)slot * 4);
        ENSURE(h->lcnt, TF * 4);
        pa.hslots = (uint32_t*)h->lslots.p; pa.hcnt = (int32_t*)h->lcnt.p; pa.lm_mode = 1;
        { Timed t(h, KS_PAIR); if (h->pair_rows) afp_launch_pair_rows(&pa, &pr, (int)g.ncblk, st); else afp_launch_pair(&pa, (int)g.ncblk, st); }
        ENSURE(h->loffs, TF * 4);
        ENSURE(h->unit_ltot, (int64_t)g.nunits * 8);
        ENSURE(h->unit_loff, (int64_t)(g.nunits + 1) * 8);
        SegScanArgs sa;
        sa.counts = (const int32_t*)h->lcnt.p; sa.seg_base = h->unit_fbase; sa.seg_len = h->unit_T;
        sa.offs = (int32_t*)h->loffs.p; sa.seg_total = (int64_t*)h->unit_ltot.p;
        { Timed t(h, KS_SEGSCAN_P); afp_launch_seg_scan(&sa, g.nunits, st); }
        { Timed t(h, KS_EXCL); afp_launch_excl_scan64((const int64_t*)h->unit_ltot.p, (int64_t*)h->unit_loff.p, g.nunits, st); }
        int64_t est = h->last_tl > 0 ? h->last_tl + h->last_tl / 4 + 4096 : TF * 4 + 4096;
        const int64_t ub = TF * (int64_t)slot;
        if (est > ub) est = ub;
        if (est < 1) est = 1;
        ENSURE(h->out_landmarks, est * 16);
        ScatterLmArgs& a = h->sl;
        a.seg_len = h->unit_T; a.seg_base = h->unit_fbase; a.blk_seg = h->cblk_unit; a.blk_t0 = h->cblk_t0;
        a.slots = (const uint32_t*)h->lslots.p; a.cnt = (const int32_t*)h->lcnt.p; a.offs = (const int32_t*)h->loffs.p;
        a.seg_off = (const int64_t*)h->unit_loff.p; a.out = (int32_t*)h->out_landmarks.p; a.slot = slot;
        a.cap = (int64_t)(h->out_landmarks.cap / 16);
        h->sl_nblk = (int)g.ncblk; h->have_sl = true;
        { Timed t(h, KS_SCAT_P); afp_launch_scatter_landmarks(&a, h->sl_nblk, st); }
        HIPCHK(hipMemcpyAsync(&h->h_totals[2], (int64_t*)h->unit_loff.p + g.nunits, 8, hipMemcpyDeviceToHost, st));
    }

    if (flags & AFP_WANT_HASHES) {
        const uint32_t* fin_slots;
        const int32_t* fin_cnt;
        int fin_slot;
        // the 6-bit dt / df fields wrap for targetdt > 64 or targetdf > 32, so one unit can emit equal hashes
        // (a list-order peak list may name a bin twice: equal hashes again)
        const bool wrap_dups = h->prm.targetdt > 64 || h->prm.targetdf > 32 || h->pair_rows;
        const int64_t oslot = (int64_t)S * slot;
        const size_t fused_lds = (size_t)S * (g.pch + h->prm.targetdt) * 36 + (size_t)16 * (oslot + 4) + 64;
        const bool thread_path = h->force_generic_pair || h->pair_rows;     // measured: the fused kernel wins even for one shift (c3 0.30 vs 0.39 ms)
        if (!thread_path && oslot <= 2048 && fused_lds <= 64 * 1024) {
            // fused wavefront-cooperative pairing + merge + sort (k_pairmerge)
            DevBuf& sl = S > 1 ? h->mslots : h->hslots;
            DevBuf& ct = S > 1 ? h->mcnt : h->hcnt;
            ENSURE(sl, g.total_mframes * oslot * 4);
            ENSURE(ct, g.total_mframes * 4);
            PairMergeArgs pm;
            pm.unit_T = h->unit_T; pm.unit_fbase = h->unit_fbase; pm.clip_mfbase = h->clip_mfbase; pm.clip_T0 = h->clip_T0;
            pm.pblk_clip = h->pblk_clip; pm.pblk_t0 = h->pblk_t0; pm.masks = (const uint64_t*)h->masks.p;
            pm.oslots = (uint32_t*)sl.p; pm.ocnt = (int32_t*)ct.p; pm.oslot = (int32_t)oslot;
            pm.dedupe = (S > 1 || wrap_dups) ? 1 : 0;
            pm.S = S; pm.ch = g.pch; pm.fanout = F; pm.targetdf = h->prm.targetdf; pm.mindt = h->prm.mindt; pm.targetdt = h->prm.targetdt;
            {
                Timed t(h, KS_PAIR);
                // one shift, narrow window, no wrapping fields, short lists: lane-per-peak kernel
                const bool lane_path = !h->no_pairlane && S == 1 && !wrap_dups && h->prm.targetdf <= 32 && K <= 8 && F <= 8 &&
                                       (g.pch % 256) == 0 && afp_pairlane_lds(g.pch, h->prm.targetdt, F) <= 64 * 1024;
                // several shifts with the same restrictions (and shifts x 8 peaks <= one wavefront): lane-per-peak too
                const bool lane_path_ms = h->pairlane_ms && !h->no_pairlane && S > 1 && S * 8 <= 64 && !wrap_dups && h->prm.targetdf <= 32 && K <= 8 &&
                                          F <= 16 && g.pch / 4 <= 64 && afp_pairlane_ms_lds(g.pch, h->prm.targetdt, F, S, K) <= 64 * 1024;
                // (k_pairlane zeroes the counts of its own columns; the other two skip empty columns)
                if (!lane_path) HIPCHK(hipMemsetAsync(ct.p, 0, g.total_mframes * 4, st));
                if (lane_path) afp_launch_pairlane(&pm, (int)g.npblk, st);
                else if (lane_path_ms) afp_launch_pairlane_ms(&pm, (int)g.npblk, st);
                else afp_launch_pairmerge(&pm, (int)g.npblk, st);
            }
            fin_slots = (const uint32_t*)sl.p; fin_cnt = (const int32_t*)ct.p; fin_slot = (int)oslot;
        } else {
        ENSURE(h->hslots, TF * (int64_t)slot * 4);
        ENSURE(h->hcnt, TF * 4);
        pa.hslots = (uint32_t*)h->hslots.p; pa.hcnt = (int32_t*)h->hcnt.p; pa.lm_mode = 0;
        { Timed t(h, KS_PAIR); if (h->pair_rows) afp_launch_pair_rows(&pa, &pr, (int)g.ncblk, st); else afp_launch_pair(&pa, (int)g.ncblk, st); }
        fin_slots = (const uint32_t*)h->hslots.p;
        fin_cnt = (const int32_t*)h->hcnt.p;
        fin_slot = slot;
        if (S > 1 || wrap_dups) {
            const int mslot = S * slot;
            ENSURE(h->mslots, g.total_mframes * (int64_t)mslot * 4);
            ENSURE(h->mcnt, g.total_mframes * 4);
            MergeArgs m;
            m.unit_T = h->unit_T; m.unit_fbase = h->unit_fbase; m.clip_mfbase = h->clip_mfbase;
            m.clip_T0 = h->clip_T0; m.mblk_clip = h->mblk_clip; m.mblk_t0 = h->mblk_t0;
            m.hslots = (const uint32_t*)h->hslots.p; m.hcnt = (const int32_t*)h->hcnt.p;
            m.mslots = (uint32_t*)h->mslots.p; m.mcnt = (int32_t*)h->mcnt.p;
            m.slot = slot; m.mslot = mslot; m.S = S;
            { Timed t(h, KS_MERGE); afp_launch_merge(&m, (int)g.nmblk, st); }
            fin_slots = (const uint32_t*)h->mslots.p; fin_cnt = (const int32_t*)h->mcnt.p; fin_slot = mslot;
        }
        }
        ENSURE(h->hoffs, g.total_mframes * 4);
        ENSURE(h->clip_tot, (int64_t)g.nclips * 8);
        ENSURE(h->clip_hoff, (int64_t)(g.nclips + 1) * 8);
        // one clip, hashes only, ending in the pinned image (Analyzer.wavfile2hashes): offsets, scatter and export are one launch
        // of one workgroup, issued by the caller once the export arguments are known (k_finish_one)
        h->fuse_finish = h->export_mode && g.nclips == 1 && !(flags & AFP_WANT_PEAKS) && g.total_mframes <= afp_finish_one_max_frames() &&
                         !h->timing && !getenv("AFP_NO_FUSED_FINISH");
        SegScanArgs sa;
        sa.counts = fin_cnt; sa.seg_base = h->clip_mfbase; sa.seg_len = h->clip_T0;
        sa.offs = (int32_t*)h->hoffs.p; sa.seg_total = (int64_t*)h->clip_tot.p;
        if (!h->fuse_finish) {
            { Timed t(h, KS_SEGSCAN_H); afp_launch_seg_scan(&sa, g.nclips, st); }
            { Timed t(h, KS_EXCL); afp_launch_excl_scan64((const int64_t*)h->clip_tot.p, (int64_t*)h->clip_hoff.p, g.nclips, st); }
        }
        // Outputs are sized from the previous batch (x1.25) or a first-call estimate; the scatter drops
        // rows that do not fit and finalize() re-runs it after growing the buffer -- so the call never
        // blocks on the GPU and batches on different handles/streams overlap.
        int64_t est = h->last_th > 0 ? h->last_th + h->last_th / 4 + 4096 : TF * 4 + 4096;
        const int64_t ub = g.total_mframes * (int64_t)fin_slot;
        if (est > ub) est = ub;
        if (est < 1) est = 1;
        ENSURE(h->out_hashes, est * 8);
        ScatterHashArgs& a = h->sh;
        a.seg_len = h->clip_T0; a.seg_base = h->clip_mfbase; a.blk_seg = h->mblk_clip; a.blk_t0 = h->mblk_t0;
        a.slots = fin_slots; a.cnt = fin_cnt; a.offs = (const int32_t*)h->hoffs.p;
        a.seg_off = (const int64_t*)h->clip_hoff.p; a.out = (int32_t*)h->out_hashes.p; a.slot = fin_slot;
        a.cap = (int64_t)(h->out_hashes.cap / 8);
        h->sh_nblk = (int)g.nmblk; h->have_sh = true;
        if (!h->fuse_finish) { Timed t(h, KS_SCAT_H); afp_launch_scatter_hashes(&a, h->sh_nblk, st); }
        if (!h->export_mode) HIPCHK(hipMemcpyAsync(&h->h_totals[0], (int64_t*)h->clip_hoff.p + g.nclips, 8, hipMemcpyDeviceToHost, st));
    }
    if (flags & AFP_WANT_PEAKS) {
        ENSURE(h->poffs, TF * 4);
        ENSURE(h->unit_tot, (int64_t)g.nunits * 8);
        ENSURE(h->unit_poff, (int64_t)(g.nunits + 1) * 8);
        afp_launch_mask_popc((const uint64_t*)h->masks.p, (int32_t*)h->pcnt.p, TF, st);
        SegScanArgs sa;
        sa.counts = (const int32_t*)h->pcnt.p; sa.seg_base = h->unit_fbase; sa.seg_len = h->unit_T;
        sa.offs = (int32_t*)h->poffs.p; sa.seg_total = (int64_t*)h->unit_tot.p;
        { Timed t(h, KS_SEGSCAN_P); afp_launch_seg_scan(&sa, g.nunits, st); }
        { Timed t(h, KS_EXCL); afp_launch_excl_scan64((const int64_t*)h->unit_tot.p, (int64_t*)h->unit_poff.p, g.nunits, st); }
        int64_t est = h->last_tp > 0 ? h->last_tp + h->last_tp / 4 + 4096 : TF * 2 + 4096;
        const int64_t ub = TF * (int64_t)K;
        if (est > ub) est = ub;
        if (est < 1) est = 1;
        ENSURE(h->out_peaks, est * 8);
        ScatterPeakArgs& a = h->sp;
        a.seg_len = h->unit_T; a.seg_base = h->unit_fbase; a.blk_seg = h->cblk_unit; a.blk_t0 = h->cblk_t0;
        a.masks = (const uint64_t*)h->masks.p; a.offs = (const int32_t*)h->poffs.p;
        a.seg_off = (const int64_t*)h->unit_poff.p; a.out = (int32_t*)h->out_peaks.p;
        a.cap = (int64_t)(h->out_peaks.cap / 8);
        h->sp_nblk = (int)g.ncblk; h->have_sp = true;
        { Timed t(h, KS_SCAT_P); afp_launch_scatter_peaks(&a, h->sp_nblk, st); }
        if (!h->export_mode) HIPCHK(hipMemcpyAsync(&h->h_totals[1], (int64_t*)h->unit_poff.p + g.nunits, 8, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipGetLastError());
    return AFP_OK;
}

static int extract_device_any(afp_handle* h, const void* d_pcm, int s16, const int64_t* off, int32_t nclips,
                              uint32_t flags)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->have_params) return AFP_ERR_STATE;
    if (nclips < 0 || (nclips > 0 && !off)) return AFP_ERR_ARG;
    if (nclips > 0 && (!d_pcm && off[nclips] > off[0])) return AFP_ERR_ARG;
    if (flags & AFP_WANT_LANDMARKS) return AFP_ERR_ARG;      // landmarks come from afp_pairs_from_peaks
    HIPCHK(hipSetDevice(h->device));
    h->extracted = false;
    h->export_mode = false;
    h->cur_pcm = d_pcm; h->cur_kind = s16; h->cur_flags = flags;
    const int S = h->prm.nshifts;
    // reuse the previous descriptor upload when the batch shape is unchanged (steady-state ingest)
    const bool cached = h->desc_valid && h->last_S == S && (int32_t)h->last_offsets.size() == nclips + 1 && nclips > 0 &&
        memcmp(h->last_offsets.data(), off, sizeof(int64_t) * (nclips + 1)) == 0 &&
        memcmp(h->last_shift_offsets.data(), h->prm.shift_offsets, sizeof(int32_t) * S) == 0;
    Geometry g;
    h->desc_cached = cached;
    if (cached) {
        g = h->geom;
    } else {
        std::vector<UnitIn> units;
        int r = units_from_offsets(h, off, nclips, units);
        if (r != AFP_OK) return r;
        compute_geometry(h, nclips, units, g);
        if (g.nblk > 0x7fffffffLL || g.total_frames > ((int64_t)1 << 40)) return AFP_ERR_ARG;
        if (workspace_bytes(h, g, flags) > h->ws_limit) return AFP_ERR_NOMEM;
        if (g.nunits > 0) {
            r = build_descriptors(h, units, g);
            if (r != AFP_OK) return r;
            h->last_offsets.assign(off, off + nclips + 1);
            h->last_S = S;
            h->last_shift_offsets.assign(h->prm.shift_offsets, h->prm.shift_offsets + S);
            h->geom = g;
            h->desc_valid = true;
        }
    }
    if (workspace_bytes(h, g, flags) > h->ws_limit) return AFP_ERR_NOMEM;
    adopt_geometry(h, g, flags);
    if (g.nunits == 0) { h->extracted = true; h->finalized = true; h->have_sh = h->have_sp = h->have_sl = false; return AFP_OK; }

    hipStream_t st = h->stream;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (h->timing) { pe0 = get_event(h); pe1 = get_event(h); }
    // staged mode: spectral stage on stage_a, scan + pairing on stage_b, both ordered after what is already
    // queued on the handle's stream (PCM / descriptor uploads) and after this handle's previous batch (its
    // workspace is reused).  The end of the batch is ev_b; the result accessors wait for it on the host.
    const bool staged = h->stage_a != nullptr;
    hipStream_t sa = staged ? h->stage_a : st, sb = staged ? h->stage_b : st, sc = staged ? h->stage_c : st;
    if (staged) {
        HIPCHK(hipEventRecord(h->ev_in, st));
        HIPCHK(hipStreamWaitEvent(sa, h->ev_in, 0));
        if (h->join_pending) HIPCHK(hipStreamWaitEvent(sa, h->ev_b, 0));
    }
    if (pe0) (void)hipEventRecord(pe0, sa);
    int r = run_spectral(h, d_pcm, s16, g, flags, sa, sb, h->ev_a);      // (staged: orders sb behind the STFT through ev_a)
    if (r == AFP_OK) r = run_scan(h, g, flags, sb);
    if (staged && sc != sb) { HIPCHK(hipEventRecord(h->ev_s, sb)); HIPCHK(hipStreamWaitEvent(sc, h->ev_s, 0)); }
    h->pair_K = 0;
    // One file per call (the Analyzer class): the chain ends with k_export, which writes rows, offsets, unit flags and the
    // totals into pinned host memory -- afp_fetch_all is then one wait and a host memcpy (measured on a 10 s file: 80 us of
    // pageable device-to-host copies -> a few us).  Larger batches keep the copies: their rows do not fit the image.
    h->export_mode = h->export_max_units > 0 && g.nunits <= h->export_max_units && g.total_frames > 0 &&
                     (flags & (AFP_WANT_HASHES | AFP_WANT_PEAKS)) != 0;
    h->export_redo = false;
    h->fuse_finish = false;
    if (h->export_mode && !h->h_export) {
        h->h_export_cap = (int64_t)4 << 20;
        HIPCHK(hipHostMalloc((void**)&h->h_export, (size_t)h->h_export_cap, hipHostMallocDefault));
    }
    if (r == AFP_OK) r = run_back(h, g, flags, sc);
    if (r == AFP_OK && h->h_totals) {
        h->h_totals[4] = h->h_totals[5] = h->h_totals[6] = 0;
        const bool nt_on = h->nt_eps > 0.0 && g.total_frames > 0 && h->cerr.p;
        if (nt_on && !h->export_mode) HIPCHK(hipMemcpyAsync(&h->h_totals[6], (int32_t*)h->cerr.p + 2, 4, hipMemcpyDeviceToHost, sc));
        if (h->export_mode) {
            ExportArgs ea;
            memset(&ea, 0, sizeof(ea));
            if (nt_on) ea.nt_count = (const int32_t*)h->cerr.p + 2;
            if (h->have_sh) { ea.hashes = (const int32_t*)h->out_hashes.p; ea.clip_hoff = (const int64_t*)h->clip_hoff.p; ea.cap_h = h->sh.cap; }
            if (h->have_sp) { ea.peaks = (const int32_t*)h->out_peaks.p; ea.unit_poff = (const int64_t*)h->unit_poff.p; ea.cap_p = h->sp.cap; }
            ea.stats = (const UnitStats*)h->stats.p;
            ea.seg_status = h->batch_seg ? (const int32_t*)h->seg_status.p : nullptr;
            ea.totals = h->h_totals; ea.host = h->h_export; ea.host_cap = h->h_export_cap;
            ea.nclips = g.nclips; ea.nunits = g.nunits;
            if (h->batch_seg) {
                // the kernel clears the segment scan's status block once it has copied the status out: the next segmented
                // batch on this handle needs no memset (run_scan)
                ea.seg_zero = (int32_t*)h->seg_status.p; ea.zero_words = (int32_t)(h->seg_zero_bytes / 4);
                h->seg_clean_ptr = h->seg_status.p;
                h->seg_clean_bytes = h->seg_clean_keep;
            }
            if (h->fuse_finish && h->have_sh) afp_launch_finish_one(&h->sh, (int32_t*)h->hoffs.p, (int64_t*)h->clip_tot.p, (int64_t*)h->clip_hoff.p, &ea, sc);
            else afp_launch_export(&ea, 32, sc);
            HIPCHK(hipGetLastError());
        } else if (h->batch_seg) HIPCHK(hipMemcpyAsync(&h->h_totals[4], h->seg_status.p, 16, hipMemcpyDeviceToHost, sc));
        h->h_totals[3] = 0;
        if (h->batch_compact) HIPCHK(hipMemcpyAsync(&h->h_totals[3], h->cerr.p, 4, hipMemcpyDeviceToHost, sc));
    }
    if (staged) { HIPCHK(hipEventRecord(h->ev_b, sc)); h->join_pending = true; }
    h->tstream = nullptr;
    if (r != AFP_OK) return r;
    h->finalized = false;
    if (h->timing && pe0 && pe1) {
        (void)hipEventRecord(pe1, sc);
        EvPair ep; ep.slot = KS_PIPELINE; ep.a = pe0; ep.b = pe1;
        h->pending.push_back(ep);
    }
    h->extracted = true;
    return AFP_OK;
}

extern "C" int afp_extract_device(afp_handle* h, const float* d_pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_device_any(h, d_pcm, 0, off, nclips, flags);
}
extern "C" int afp_extract_device_f64(afp_handle* h, const double* d_pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_device_any(h, d_pcm, 2, off, nclips, flags);
}
extern "C" int afp_extract_device_s16(afp_handle* h, const int16_t* d_pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_device_any(h, d_pcm, 1, off, nclips, flags);
}

// Pairing / hashing from given peak lists: replaces Analyzer.peaks2landmarks (audfprint_analyze.py:310-343)
// + landmarks2hashes (:81-96) + unique/sort (:414-422) for peaks that did not come from this
// handle's own scan (e.g. a .afpk file, wavfile2peaks :351-354).
extern "C" int afp_pairs_from_peaks(afp_handle* h, const int32_t* peaks, const int64_t* upo, int32_t nclips,
                                    uint32_t flags)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->have_params) return AFP_ERR_STATE;
    if (nclips < 0 || (nclips > 0 && !upo)) return AFP_ERR_ARG;
    if (flags & (AFP_WANT_PEAKS | AFP_KEEP_DEBUG)) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->join_pending) HIPCHK(sync_handle(h));
    h->extracted = false;
    h->desc_valid = false;                     // descriptors below do not describe a PCM batch
    const int S = h->prm.nshifts;
    if ((int64_t)nclips * S > 0x7fffffffLL) return AFP_ERR_ARG;
    const int nunits = nclips * S;
    std::vector<UnitIn> units((size_t)nunits);
    const int64_t np = nunits > 0 ? upo[nunits] - upo[0] : 0;
    if (np < 0 || (np > 0 && !peaks)) return AFP_ERR_ARG;
    int maxrun = 0;                            // most peaks any one column holds (a .afpk may exceed maxpksperframe)
    bool list_order = false;                   // some column lists its bins out of ascending order, or one twice
    for (int u = 0; u < nunits; u++) {
        if (upo[u + 1] < upo[u]) return AFP_ERR_ARG;
        int32_t last = -1, lastbin = -1;
        int run = 0;
        for (int64_t i = upo[u]; i < upo[u + 1]; i++) {
            const int32_t col = peaks[2 * i], bin = peaks[2 * i + 1];
            if (col < 0 || col >= (1 << 24) || bin < 0 || bin >= AFP_NBINS || col < last) return AFP_ERR_ARG;   // (2^24 frames = 108 h: bounds the mask workspace)
            if (col == last && bin <= lastbin) list_order = true;
            run = col == last ? run + 1 : 1;
            if (run > maxrun) maxrun = run;
            last = col; lastbin = bin;
        }
        units[u].pcm_off = 0; units[u].n = 0;
        units[u].T = last + 1;                  // scols = column of the final peak + 1 (:321)
    }
    if (list_order && maxrun > 256) return AFP_ERR_ARG;      // (a slot of k_pair_rows holds the pairs of 256 rows of one column)
    Geometry g;
    compute_geometry(h, nclips, units, g);
    if (g.total_frames > ((int64_t)1 << 40)) return AFP_ERR_ARG;
    adopt_geometry(h, g, flags);
    h->export_mode = false;                    // (results of this entry point are fetched by afp_fetch_hashes / afp_fetch_landmarks)
    if (nunits == 0 || g.total_frames == 0) {
        h->extracted = true; h->finalized = true; h->have_sh = h->have_sp = h->have_sl = false;
        if (h->h_totals) h->h_totals[0] = h->h_totals[1] = h->h_totals[2] = 0;
        h->total_frames = 0;
        return AFP_OK;
    }
    int r = build_descriptors(h, units, g);
    if (r != AFP_OK) return r;
    hipStream_t st = h->stream;
    const int64_t TF = g.total_frames;
    ENSURE(h->masks, TF * 32);
    ENSURE(h->in_peaks, np * 8);
    ENSURE(h->in_upo, (int64_t)(nunits + 1) * 8);
    HIPCHK(hipMemsetAsync(h->masks.p, 0, TF * 32, st));
    HIPCHK(hipMemcpyAsync(h->in_peaks.p, peaks + 2 * upo[0], np * 8, hipMemcpyHostToDevice, st));
    {   // offsets relative to the first row copied
        std::vector<int64_t> rel((size_t)nunits + 1);
        for (int u = 0; u <= nunits; u++) rel[u] = upo[u] - upo[0];
        HIPCHK(hipMemcpyAsync(h->in_upo.p, rel.data(), (size_t)(nunits + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));       // rel is a stack-lifetime buffer
    }
    afp_launch_masks_from_peaks((const int32_t*)h->in_peaks.p, (const int64_t*)h->in_upo.p, nunits, np,
                                h->unit_fbase, (uint64_t*)h->masks.p, st);
    h->pair_K = std::min(256, std::max(h->prm.maxpksperframe, maxrun));
    h->pair_rows = list_order;
    if (list_order) {
        // peaks_at[col] in list order (audfprint_analyze.py:323-326): rows per column, then the first row of every column
        ENSURE(h->pcnt, TF * 4);
        ENSURE(h->poffs, TF * 4);
        ENSURE(h->unit_tot, (int64_t)nunits * 8);
        HIPCHK(hipMemsetAsync(h->pcnt.p, 0, TF * 4, st));
        afp_launch_rows_count((const int32_t*)h->in_peaks.p, (const int64_t*)h->in_upo.p, nunits, np, h->unit_fbase, (int32_t*)h->pcnt.p, st);
        SegScanArgs sa;
        sa.counts = (const int32_t*)h->pcnt.p; sa.seg_base = h->unit_fbase; sa.seg_len = h->unit_T;
        sa.offs = (int32_t*)h->poffs.p; sa.seg_total = (int64_t*)h->unit_tot.p;
        afp_launch_seg_scan(&sa, nunits, st);
    }
    r = run_back(h, g, flags, st);
    h->pair_K = 0;
    h->pair_rows = false;
    h->tstream = nullptr;
    if (r != AFP_OK) return r;
    h->finalized = false;
    h->extracted = true;
    return AFP_OK;
}

// Analyzer._decaying_threshold_fwd_prune (audfprint_analyze.py:199-231) and _decaying_threshold_bwd_prune_peaks
// (:233-253) over a spectrogram the CALLER supplies (the two semi-private methods take `sgram` as an argument).
extern "C" int afp_prune_spectrogram(afp_handle* h, const double* sgram, int32_t T, double a_dec, const uint8_t* peaks_in,
                                     uint8_t* fwd_out, uint8_t* bwd_out)
{
    if (!h || T < 0 || (T > 0 && !sgram)) return AFP_ERR_ARG;
    if (!h->have_params) return AFP_ERR_STATE;
    if (!(a_dec > 0.0) || T > 0x3fffffff) return AFP_ERR_PARAM;
    if (T == 0) return AFP_OK;
    HIPCHK(hipSetDevice(h->device));
    if (h->join_pending) HIPCHK(sync_handle(h));
    h->extracted = false;
    h->desc_valid = false;
    const int S_saved = h->prm.nshifts;
    h->prm.nshifts = 1;                                    // one unit, whatever the extraction parameters say
    std::vector<UnitIn> units(1);
    units[0].pcm_off = 0; units[0].n = 0; units[0].T = T;
    Geometry g;
    compute_geometry(h, 1, units, g);
    h->prm.nshifts = S_saved;
    adopt_geometry(h, g, 0);
    h->export_mode = false;
    int r = build_descriptors(h, units, g);
    if (r != AFP_OK) return r;
    hipStream_t st = h->stream;
    const int64_t TF = T;
    // forward candidates per frame: maxpksperframe, or as many as the densest column of the given mask holds
    int K = h->prm.maxpksperframe;
    std::vector<double> cv;
    std::vector<int32_t> cb;
    if (peaks_in) {
        int most = 1;
        for (int t = 0; t < T; t++) {
            int c = 0;
            for (int b = 0; b < AFP_NBINS; b++) c += peaks_in[(size_t)t * AFP_NBINS + b] ? 1 : 0;
            if (c > most) most = c;
        }
        if (most > AFP_MAX_PKS) return AFP_ERR_PARAM;      // one wavefront lane per peak of a column
        K = most;
        cv.assign((size_t)T * K, 0.0);
        cb.assign((size_t)T * K, -1);
        std::vector<std::pair<double, int>> col;
        for (int t = 0; t < T; t++) {
            col.clear();
            for (int b = 0; b < AFP_NBINS; b++)
                if (peaks_in[(size_t)t * AFP_NBINS + b]) col.push_back({sgram[(size_t)t * AFP_NBINS + b], b});
            std::sort(col.begin(), col.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b2) {
                return a.first > b2.first || (a.first == b2.first && a.second > b2.second);       // sorted(..., reverse=True), :241
            });
            for (size_t i = 0; i < col.size(); i++) { cv[(size_t)t * K + i] = col[i].first; cb[(size_t)t * K + i] = col[i].second; }
        }
    }
    h->K = K;
    ENSURE(h->logS, TF * AFP_NBINS * 8);
    ENSURE(h->stats, sizeof(UnitStats));
    ENSURE(h->blk_corr, g.nblk * 8);
    ENSURE(h->cand_val, TF * K * 8);
    ENSURE(h->cand_bin, TF * K * 4);
    ENSURE(h->masks, TF * 32);
    ENSURE(h->pcnt, TF * 4);
    ENSURE(h->unit_mean, 8);
    ENSURE(h->ylast, AFP_NBINS * 8);
    UnitStats us;
    us.logfloor = 0.0; us.lsum = 0.0; us.pmax = 1.0; us.flags = 0; us.pad = 0; us.tie_first = 0; us.tie_last = -1;
    HIPCHK(hipMemcpyAsync(h->logS.p, sgram, TF * AFP_NBINS * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h->stats.p, &us, sizeof(us), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(h->masks.p, 0, TF * 32, st));
    if (peaks_in) {
        HIPCHK(hipMemcpyAsync(h->cand_val.p, cv.data(), TF * K * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(h->cand_bin.p, cb.data(), TF * K * 4, hipMemcpyHostToDevice, st));
    } else {
        HIPCHK(hipMemsetAsync(h->cand_bin.p, 0xFF, TF * K * 4, st));
    }
    ScanArgs s;
    memset(&s, 0, sizeof(s));
    s.unit_T = h->unit_T; s.unit_fbase = h->unit_fbase; s.unit_bbase = h->unit_bbase;
    s.stats = (const UnitStats*)h->stats.p; s.blk_corr = (const double*)h->blk_corr.p;
    s.logS = (const double*)h->logS.p; s.gauss = (const double*)h->d_gauss.p;
    s.a_dec = a_dec; s.pole = h->prm.hpf_pole; s.K = K;
    s.cand_val = (double*)h->cand_val.p; s.cand_bin = (int32_t*)h->cand_bin.p;
    s.masks = (uint64_t*)h->masks.p; s.ylast = (double*)h->ylast.p; s.unit_mean = (double*)h->unit_mean.p;
    s.sgram_dbg = nullptr; s.prof = nullptr; s.raw_rows = 1; s.fwd_off = peaks_in ? 1 : 0;
    s.cvals = nullptr; s.lmask = nullptr; s.head = nullptr;
    afp_launch_scan(&s, 1, st);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> fb;
    std::vector<uint64_t> mk;
    if (fwd_out && !peaks_in) { fb.resize((size_t)T * K); HIPCHK(hipMemcpyAsync(fb.data(), h->cand_bin.p, TF * K * 4, hipMemcpyDeviceToHost, st)); }
    if (bwd_out) { mk.resize((size_t)T * 4); HIPCHK(hipMemcpyAsync(mk.data(), h->masks.p, TF * 32, hipMemcpyDeviceToHost, st)); }
    HIPCHK(hipStreamSynchronize(st));
    if (fwd_out) {
        if (peaks_in) memcpy(fwd_out, peaks_in, (size_t)T * AFP_NBINS);
        else {
            memset(fwd_out, 0, (size_t)T * AFP_NBINS);
            for (int t = 0; t < T; t++)
                for (int k = 0; k < K; k++) { const int b = fb[(size_t)t * K + k]; if (b >= 0) fwd_out[(size_t)t * AFP_NBINS + b] = 1; }
        }
    }
    if (bwd_out)
        for (int t = 0; t < T; t++)
            for (int b = 0; b < AFP_NBINS; b++) bwd_out[(size_t)t * AFP_NBINS + b] = (uint8_t)((mk[(size_t)t * 4 + (b >> 6)] >> (b & 63)) & 1ull);
    return AFP_OK;
}

// landmarks2hashes (audfprint_analyze.py:81-96) over an arbitrary (L,4) int32 array of
// (time, bin1, bin2, dtime) rows -> (L,2) int32 rows (time, hash).  Host buffers in and out.
extern "C" int afp_hashes_from_landmarks(afp_handle* h, const int32_t* lm, int64_t nrows, int32_t* out)
{
    if (!h || nrows < 0 || (nrows > 0 && (!lm || !out))) return AFP_ERR_ARG;
    if (nrows == 0) return AFP_OK;
    HIPCHK(hipSetDevice(h->device));
    ENSURE(h->lm_in, nrows * 16);
    ENSURE(h->lm_out, nrows * 8);
    HIPCHK(hipMemcpyAsync(h->lm_in.p, lm, nrows * 16, hipMemcpyHostToDevice, h->stream));
    afp_launch_lm2hash((const int32_t*)h->lm_in.p, (int32_t*)h->lm_out.p, nrows, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, h->lm_out.p, nrows * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    return AFP_OK;
}

// Wait for the batch in flight; if an output buffer was too small, grow it and re-run the scatter.
static int finalize(afp_handle* h)
{
    if (h->finalized) return AFP_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(sync_handle(h));
    h->batch_redone = false;
    if (h->h_totals && (int32_t)h->h_totals[3] != 0) {
        // A chunk of the compact spectral stage gave up waiting for its predecessor's filter state (k_stft.hip: the protocol
        // itself cannot reach that bound -- a fault, or the test hook): nothing of the batch can be trusted.  Re-run it on the
        // DENSE path, which has no cross-workgroup dependency, from the same PCM and offsets; the caller sees the results a
        // little later and afp_get_path_stats counts the event.
        h->h_totals[3] = 0;
        HIPCHK(hipMemsetAsync(h->cerr.p, 0, 4, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        const int saved_mode = h->compact_mode;
        const std::vector<int64_t> off = h->last_offsets;            // (extract_device_any re-assigns last_offsets)
        h->compact_mode = 0;
        h->finalized = true;
        const int r = extract_device_any(h, h->cur_pcm, h->cur_kind, off.data(), h->nclips, h->cur_flags);
        h->compact_mode = saved_mode;
        if (r != AFP_OK) { h->extracted = false; return r; }
        HIPCHK(sync_handle(h));
        h->compact_redone_total++;
        h->batch_redone = true;
    }
    if (h->batch_seg && h->batch_short_cut && h->h_totals) {
        const int32_t* sg = reinterpret_cast<const int32_t*>(&h->h_totals[4]);       // [0] failed units, [1] / [2] segments re-run
        if ((int64_t)(sg[1] + sg[2]) * 20 > (int64_t)h->batch_nseg) { h->seg_short_penalty = 32; h->seg_short_backoffs++; }
        h->batch_short_cut = false;            // (judged once)
    }
    h->batch_nt_redone = false;
    h->nt_units_last = h->h_totals ? (int32_t)h->h_totals[6] : 0;
    if (h->nt_units_last > 0 && h->batch_compact && !h->batch_redone) {
        // The guard fired on the COMPACT path, whose filtered values differ from the dense path's by a few ulps (the mean is
        // subtracted after the onset filter): the dense path -- the reference's operation order -- decides.  Re-run the batch
        // there; the units keep their UNIT_NEARTIE mark if the dense comparison is that close too.
        const int saved_mode = h->compact_mode;
        const std::vector<int64_t> off = h->last_offsets;
        h->compact_mode = 0;
        h->finalized = true;
        const int r = extract_device_any(h, h->cur_pcm, h->cur_kind, off.data(), h->nclips, h->cur_flags);
        h->compact_mode = saved_mode;
        if (r != AFP_OK) { h->extracted = false; return r; }
        HIPCHK(sync_handle(h));
        h->nt_redone_total++;
        h->batch_nt_redone = true;
        h->nt_units_last = (int32_t)h->h_totals[6];
    }
    const int64_t th = h->h_totals ? h->h_totals[0] : 0, tp = h->h_totals ? h->h_totals[1] : 0;
    const int64_t tl = h->h_totals ? h->h_totals[2] : 0;
    bool redo = false;
    if (h->have_sh && th > h->sh.cap) {
        ENSURE(h->out_hashes, th * 8);
        h->sh.out = (int32_t*)h->out_hashes.p; h->sh.cap = (int64_t)(h->out_hashes.cap / 8);
        afp_launch_scatter_hashes(&h->sh, h->sh_nblk, h->stream);
        redo = true;
    }
    if (h->have_sp && tp > h->sp.cap) {
        ENSURE(h->out_peaks, tp * 8);
        h->sp.out = (int32_t*)h->out_peaks.p; h->sp.cap = (int64_t)(h->out_peaks.cap / 8);
        afp_launch_scatter_peaks(&h->sp, h->sp_nblk, h->stream);
        redo = true;
    }
    if (h->have_sl && tl > h->sl.cap) {
        ENSURE(h->out_landmarks, tl * 16);
        h->sl.out = (int32_t*)h->out_landmarks.p; h->sl.cap = (int64_t)(h->out_landmarks.cap / 16);
        afp_launch_scatter_landmarks(&h->sl, h->sl_nblk, h->stream);
        redo = true;
    }
    if (redo) { HIPCHK(hipGetLastError()); HIPCHK(sync_handle(h)); h->export_redo = true; }
    drain_retired(false);                 // (the batch is complete and its results are about to be read: as idle as this handle gets)
    h->total_hashes = th; h->total_peaks = tp; h->total_landmarks = tl;
    if (h->have_sh) h->last_th = th;
    if (h->have_sp) h->last_tp = tp;
    if (h->have_sl) h->last_tl = tl;
    h->finalized = true;
    return AFP_OK;
}
#define FINALIZE(h)                      \
    do {                                 \
        int r_ = finalize(h);            \
        if (r_ != AFP_OK) return r_;     \
    } while (0)

// ONE upload stream per device, shared by every handle of the process.  Batches submitted through several contexts used to
// copy on their own streams; the copies overlapped, and of two host-to-device copies in flight the runtime runs one on the
// DMA engine and the other as a shader copy (`__amd_rocclr_copyBuffer`) that fills the compute units with wavefronts waiting
// on PCIe: in the r04 trace of the 12 500-clip job four of ten uploads went that way, k_stft beside them took 2.8-5.3 ms
// instead of 0.5, and the link idled 6 of 56 ms.  The link is one resource: uploads queue on one stream, each handle orders
// its own work against it with one event.  (AFP_UPLOAD_STREAM=0: copies on the handle's stream, as before.)
static hipStream_t upload_stream(int device)
{
    static std::mutex mu;
    static hipStream_t up[64];
    static bool off = false, init = false;
    std::lock_guard<std::mutex> g(mu);
    if (!init) { const char* e = getenv("AFP_UPLOAD_STREAM"); off = e && atoi(e) == 0; init = true; }
    if (off || device < 0 || device >= 64) return nullptr;
    if (!up[device]) {
        // LOWEST priority: priorities have their own hardware queues, and nothing else here uses this one -- on a queue shared
        // with a kernel stream the stream's event packets wait behind that stream's kernels (measured: every upload then
        // started only when the batch before it had finished)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        if (hipStreamCreateWithPriority(&up[device], hipStreamNonBlocking, least) != hipSuccess) up[device] = nullptr;
    }
    return up[device];
}

static int extract_host_any(afp_handle* h, const void* pcm, size_t ssz, const int64_t* off, int32_t nclips, uint32_t flags)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->have_params) return AFP_ERR_STATE;
    if (nclips < 0 || (nclips > 0 && !off)) return AFP_ERR_ARG;
    const int kind = ssz == 2 ? 1 : ssz == 8 ? 2 : 0;
    if (nclips == 0) return extract_device_any(h, nullptr, kind, off, 0, flags);
    const int64_t lo = off[0], hi = off[nclips];
    if (hi < lo || (hi > lo && !pcm)) return AFP_ERR_ARG;
    {   // refuse a malformed batch BEFORE the caller's buffer is read
        std::vector<UnitIn> units;
        const int r = units_from_offsets(h, off, nclips, units);
        if (r != AFP_OK) return r;
    }
    HIPCHK(hipSetDevice(h->device));
    if (h->join_pending) HIPCHK(sync_handle(h));      // the staged batch in flight may still read pcm_stage
    ENSURE(h->pcm_stage, (hi - lo) * (int64_t)ssz + 256);
    const int64_t bytes = (hi - lo) * (int64_t)ssz;
    // (r04: bouncing small uploads through a pinned buffer of the handle's own -- memcpy, then an asynchronous copy -- was tried
    //  to take the blocking pageable copy out of the one-file path; the SECOND memcpy into that buffer faulted under the HIP
    //  runtime PyTorch bundles, so the runtime's own pageable path stays)
    hipStream_t up = bytes >= ((int64_t)4 << 20) ? upload_stream(h->device) : nullptr;      // (a small upload is not worth two events)
    if (up) {
        if (!h->ev_up_done) HIPCHK(hipEventCreateWithFlags(&h->ev_up_done, hipEventDisableTiming));
        // Whatever this handle still has queued may read pcm_stage.  Resolved on the HOST: in a pipeline the handle's last
        // batch has been fetched and its stream is idle (one query); an event recorded on h->stream for the upload stream to
        // wait on sat behind other streams' packets in a shared hardware queue and held uploads back by 2-4 ms (r04 trace).
        if (hipStreamQuery(h->stream) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipStreamSynchronize(h->stream)); }
        HIPCHK(hipMemcpyAsync(h->pcm_stage.p, (const char*)pcm + lo * (int64_t)ssz, (size_t)bytes, hipMemcpyHostToDevice, up));
        HIPCHK(hipEventRecord(h->ev_up_done, up));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_up_done, 0));     // (the stage streams are ordered behind h->stream by ev_in)
    } else if (bytes > 0)
        HIPCHK(hipMemcpyAsync(h->pcm_stage.p, (const char*)pcm + lo * (int64_t)ssz, (size_t)bytes,
                              hipMemcpyHostToDevice, h->stream));
    // kernels index pcm with absolute offsets: rebase the device pointer
    const char* dbase = (const char*)h->pcm_stage.p - lo * (int64_t)ssz;
    return extract_device_any(h, dbase, kind, off, nclips, flags);
}
extern "C" int afp_extract_host(afp_handle* h, const float* pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_host_any(h, pcm, sizeof(float), off, nclips, flags);
}
extern "C" int afp_extract_host_s16(afp_handle* h, const int16_t* pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_host_any(h, pcm, sizeof(int16_t), off, nclips, flags);
}
extern "C" int afp_extract_host_f64(afp_handle* h, const double* pcm, const int64_t* off, int32_t nclips, uint32_t flags)
{
    return extract_host_any(h, pcm, sizeof(double), off, nclips, flags);
}

extern "C" int afp_result_counts(afp_handle* h, int64_t* th, int64_t* tp, int64_t* nunits)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    FINALIZE(h);
    if (th) *th = h->total_hashes;
    if (tp) *tp = h->total_peaks;
    if (nunits) *nunits = h->nunits;
    return AFP_OK;
}

extern "C" int afp_fetch_hashes(afp_handle* h, int32_t* hashes, int64_t* clip_off)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted || !(h->flags & AFP_WANT_HASHES)) return AFP_ERR_STATE;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    if (h->total_frames == 0) {
        if (clip_off) for (int i = 0; i <= h->nclips; i++) clip_off[i] = 0;
        return AFP_OK;
    }
    if (hashes && h->total_hashes > 0)
        HIPCHK(hipMemcpyAsync(hashes, h->out_hashes.p, h->total_hashes * 8, hipMemcpyDeviceToHost, h->stream));
    if (clip_off)
        HIPCHK(hipMemcpyAsync(clip_off, h->clip_hoff.p, (int64_t)(h->nclips + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    return AFP_OK;
}

extern "C" int afp_fetch_peaks(afp_handle* h, int32_t* peaks, int64_t* unit_off)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted || !(h->flags & AFP_WANT_PEAKS)) return AFP_ERR_STATE;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    if (h->total_frames == 0) {
        if (unit_off) for (int i = 0; i <= h->nunits; i++) unit_off[i] = 0;
        return AFP_OK;
    }
    if (peaks && h->total_peaks > 0)
        HIPCHK(hipMemcpyAsync(peaks, h->out_peaks.p, h->total_peaks * 8, hipMemcpyDeviceToHost, h->stream));
    if (unit_off)
        HIPCHK(hipMemcpyAsync(unit_off, h->unit_poff.p, (int64_t)(h->nunits + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    return AFP_OK;
}

extern "C" int afp_fetch_landmarks(afp_handle* h, int32_t* lm, int64_t* unit_off, int64_t* total)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted || !(h->flags & AFP_WANT_LANDMARKS)) return AFP_ERR_STATE;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    if (total) *total = h->total_landmarks;
    if (h->total_frames == 0) {
        if (unit_off) for (int i = 0; i <= h->nunits; i++) unit_off[i] = 0;
        return AFP_OK;
    }
    if (lm && h->total_landmarks > 0)
        HIPCHK(hipMemcpyAsync(lm, h->out_landmarks.p, h->total_landmarks * 16, hipMemcpyDeviceToHost, h->stream));
    if (unit_off)
        HIPCHK(hipMemcpyAsync(unit_off, h->unit_loff.p, (int64_t)(h->nunits + 1) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    return AFP_OK;
}

// Pipeline selection.  compact / seg: -1 the library's default (by batch size), 0 never, 1 always, -2 the value the handle
// was created with (the default, or what AFP_COMPACT / AFP_SEG of the environment chose); the other arguments: a positive
// value, or <= 0 for the creation-time value (the defaults, or AFP_COMPACT_MIN_UNITS / AFP_SEG_MAX_UNITS / AFP_SEG_LEN /
// AFP_SEG_WARM).  So afp_set_pipeline(h, -2, 0, -2, 0, 0, 0) undoes every earlier call.
extern "C" int afp_set_pipeline(afp_handle* h, int32_t compact, int32_t compact_min_units, int32_t seg, int32_t seg_max_units,
                                int32_t seg_len, int32_t seg_warm)
{
    if (!h || compact < -2 || compact > 1 || seg < -2 || seg > 1) return AFP_ERR_ARG;
    h->compact_mode = compact == -2 ? h->init_compact_mode : compact;
    h->seg_mode = seg == -2 ? h->init_seg_mode : seg;
    h->compact_min_units = compact_min_units > 0 ? compact_min_units : h->init_compact_min_units;
    h->seg_max_units = seg_max_units > 0 ? seg_max_units : h->init_seg_max_units;
    h->seg_len = seg_len >= 8 ? seg_len : h->init_seg_len;
    h->seg_warm = seg_warm >= 1 ? seg_warm : h->init_seg_warm;
    return AFP_OK;
}

// Test hook: the next compact launches let chunk 0 of unit 0 withhold the filter state it should hand on and bound the wait
// of its successor to a millisecond; the successor reports the fault and finalize() re-runs the batch on the dense path.
extern "C" int afp_set_compact_force_timeout(afp_handle* h, int32_t on)
{
    if (!h) return AFP_ERR_ARG;
    h->compact_force_timeout = on ? 1 : 0;
    return AFP_OK;
}

// Which path the batch last finalized took: out[0] 1 if its spectral stage was the compact one, [1] 1 if its scan was the
// segment-parallel one, [2] 1 if the compact stage reported a hand-off fault and the batch was re-run on the dense path
// (then [0] is 0: the results come from the dense kernels), [3] such re-runs since afp_create.
extern "C" int afp_get_path_stats(afp_handle* h, int32_t* out)
{
    if (!h || !out) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    FINALIZE(h);
    out[0] = h->batch_compact ? 1 : 0; out[1] = h->batch_seg ? 1 : 0; out[2] = h->batch_redone ? 1 : 0; out[3] = h->compact_redone_total;
    out[4] = h->nt_units_last; out[5] = h->batch_nt_redone ? 1 : 0; out[6] = h->nt_redone_total; out[7] = 0;
    return AFP_OK;
}

// Near-tie guard of the scan: a unit in which a decisive comparison (forward `val > sthresh`, audfprint_analyze.py:217;
// backward `val >= sthresh`, :242; the cut behind the maxpksperframe largest, :221) was decided by |a - b| <= eps carries
// AFP_UNIT_NEARTIE; a compact-path batch in which that happened is re-run on the dense path.  eps = 0 switches the guard off.
extern "C" int afp_set_neartie_eps(afp_handle* h, double eps)
{
    if (!h || !(eps >= 0.0) || eps > 1.0) return AFP_ERR_ARG;
    h->nt_eps = eps;
    return AFP_OK;
}

extern "C" int afp_set_seg_force_fail(afp_handle* h, int32_t on)
{
    if (!h) return AFP_ERR_ARG;
    h->seg_force_fail = on ? 1 : 0;
    return AFP_OK;
}

// Segment-parallel scan of the last batch: out[0] 1 if it was used, [1] segments, [2] forward / [3] backward segments
// re-run by the chain launches, [4] units whose final boundary check failed (the sequential kernel then produced their result;
// every unit if k_hpf gave up).
extern "C" int afp_get_seg_stats(afp_handle* h, int32_t* out)
{
    if (!h || !out) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    FINALIZE(h);
    out[0] = h->batch_seg ? 1 : 0; out[1] = h->batch_nseg; out[2] = out[3] = out[4] = 0;
    out[5] = h->batch_seg ? h->seg_cache_S : 0; out[6] = h->batch_seg ? h->seg_cache_W : 0; out[7] = h->seg_short_backoffs;
    if (h->batch_seg && h->h_totals) {
        const int32_t* st = reinterpret_cast<const int32_t*>(&h->h_totals[4]);
        out[4] = st[3] ? h->nunits : st[0]; out[2] = st[1]; out[3] = st[2];
    }
    return AFP_OK;
}

extern "C" int afp_fetch_unit_flags(afp_handle* h, int32_t* unit_flags)
{
    if (!h || !unit_flags) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    if (h->nunits == 0) return AFP_OK;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    if (!h->desc_valid) { for (int i = 0; i < h->nunits; i++) unit_flags[i] = 0; return AFP_OK; }   // peaks-only batch
    std::vector<UnitStats> st(h->nunits);
    HIPCHK(hipMemcpyAsync(st.data(), h->stats.p, (size_t)h->nunits * sizeof(UnitStats), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    for (int i = 0; i < h->nunits; i++) unit_flags[i] = st[i].flags;
    return AFP_OK;
}

// Per unit, the first / last frame whose non-zero samples share one parity and rise above the floor (units flagged AFP_UNIT_TIE; 0 / -1
// otherwise): outside [first, last] the spectrogram is the reference's to the usual accuracy.
extern "C" int afp_fetch_unit_tie_frames(afp_handle* h, int32_t* first, int32_t* last)
{
    if (!h || !first || !last) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    if (h->nunits == 0) return AFP_OK;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    if (!h->desc_valid) { for (int i = 0; i < h->nunits; i++) { first[i] = 0; last[i] = -1; } return AFP_OK; }
    std::vector<UnitStats> st(h->nunits);
    HIPCHK(hipMemcpyAsync(st.data(), h->stats.p, (size_t)h->nunits * sizeof(UnitStats), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_handle(h));
    for (int i = 0; i < h->nunits; i++) { first[i] = st[i].tie_first; last[i] = st[i].tie_last; }
    return AFP_OK;
}

// Everything a caller usually takes from a batch, with ONE wait: hash rows + per-clip offsets, peak rows + per-unit offsets
// (each pair only if requested at extract time and non-null here) and the per-unit flags (may be null).
extern "C" int afp_fetch_all(afp_handle* h, int32_t* hashes, int64_t* clip_off, int32_t* peaks, int64_t* unit_off, int32_t* unit_flags)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    FINALIZE(h);
    HIPCHK(hipSetDevice(h->device));
    const bool wh = (h->flags & AFP_WANT_HASHES) != 0, wp = (h->flags & AFP_WANT_PEAKS) != 0;
    if (h->total_frames == 0) {
        if (clip_off && wh) for (int i = 0; i <= h->nclips; i++) clip_off[i] = 0;
        if (unit_off && wp) for (int i = 0; i <= h->nunits; i++) unit_off[i] = 0;
        if (unit_flags) return afp_fetch_unit_flags(h, unit_flags);
        return AFP_OK;
    }
    if (h->export_mode && !h->export_redo && h->h_export) {
        // small batch: k_export left everything in the pinned image at the end of the chain (FINALIZE has waited for it)
        const int64_t* hdr = reinterpret_cast<const int64_t*>(h->h_export);
        if (hdr[0] == 1 && hdr[1] == (wh ? h->total_hashes : 0) && hdr[2] == (wp ? h->total_peaks : 0)) {
            const char* im = h->h_export;
            int64_t o = AFP_EXPORT_HDR_BYTES;
            if (wh) { if (clip_off) memcpy(clip_off, im + o, 8 * ((size_t)h->nclips + 1)); o += 8 * ((int64_t)h->nclips + 1); }
            if (wp) { if (unit_off) memcpy(unit_off, im + o, 8 * ((size_t)h->nunits + 1)); o += 8 * ((int64_t)h->nunits + 1); }
            if (unit_flags) memcpy(unit_flags, im + o, 4 * (size_t)h->nunits);
            o += 4 * (int64_t)h->nunits;
            o = (o + 15) & ~(int64_t)15;
            if (wh) { if (hashes && hdr[1] > 0) memcpy(hashes, im + o, (size_t)hdr[1] * 8); o += 8 * hdr[1]; }
            if (wp) { if (peaks && hdr[2] > 0) memcpy(peaks, im + o, (size_t)hdr[2] * 8); }
            return AFP_OK;
        }
    }
    hipStream_t st = h->stream;
    if (wh && hashes && h->total_hashes > 0) HIPCHK(hipMemcpyAsync(hashes, h->out_hashes.p, h->total_hashes * 8, hipMemcpyDeviceToHost, st));
    if (wh && clip_off) HIPCHK(hipMemcpyAsync(clip_off, h->clip_hoff.p, (int64_t)(h->nclips + 1) * 8, hipMemcpyDeviceToHost, st));
    if (wp && peaks && h->total_peaks > 0) HIPCHK(hipMemcpyAsync(peaks, h->out_peaks.p, h->total_peaks * 8, hipMemcpyDeviceToHost, st));
    if (wp && unit_off) HIPCHK(hipMemcpyAsync(unit_off, h->unit_poff.p, (int64_t)(h->nunits + 1) * 8, hipMemcpyDeviceToHost, st));
    std::vector<UnitStats> us;
    const bool wf = unit_flags && h->nunits > 0 && h->desc_valid;
    if (wf) { us.resize((size_t)h->nunits); HIPCHK(hipMemcpyAsync(us.data(), h->stats.p, (size_t)h->nunits * sizeof(UnitStats), hipMemcpyDeviceToHost, st)); }
    HIPCHK(sync_handle(h));
    if (unit_flags) for (int i = 0; i < h->nunits; i++) unit_flags[i] = wf ? us[(size_t)i].flags : 0;
    return AFP_OK;
}


extern "C" int afp_result_device_ptrs(afp_handle* h, const int32_t** dh, const int64_t** dho, const int32_t** dp,
                                      const int64_t** dpo)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    FINALIZE(h);
    if (dh) *dh = (h->flags & AFP_WANT_HASHES) ? (const int32_t*)h->out_hashes.p : nullptr;
    if (dho) *dho = (h->flags & AFP_WANT_HASHES) ? (const int64_t*)h->clip_hoff.p : nullptr;
    if (dp) *dp = (h->flags & AFP_WANT_PEAKS) ? (const int32_t*)h->out_peaks.p : nullptr;
    if (dpo) *dpo = (h->flags & AFP_WANT_PEAKS) ? (const int64_t*)h->unit_poff.p : nullptr;
    return AFP_OK;
}

// ---- hash-table build: HashTable.store for a whole batch, hash_table.py:91-138 ----
// The table lives on a stream of its own, at the highest priority the device offers: its kernels are tiny (a few microseconds
// each) and the host waits for several of them per batch, while the extraction contexts that feed the table keep every CU busy
// with kernels a thousand times longer -- on an ordinary stream each of those waits sat behind whatever was queued (r04, c4 job:
// "store" 6 ms + "replay" 15 ms of host time that was mostly waiting for a slot).
static hipStream_t tbs(afp_handle* h) { return h->tb_stream ? h->tb_stream : h->stream; }
static hipError_t tb_sync(afp_handle* h)
{
    hipError_t e = sync_handle(h);
    if (e != hipSuccess) return e;
    return h->tb_stream ? hipStreamSynchronize(h->tb_stream) : hipSuccess;
}
extern "C" int afp_table_create(afp_handle* h, int32_t hashbits, int32_t depth, int32_t maxtimebits)
{
    if (!h || hashbits < 1 || hashbits > 24 || depth < 1 || depth > 4096 || maxtimebits < 1 || maxtimebits > 24) return AFP_ERR_PARAM;
    HIPCHK(hipSetDevice(h->device));
    if (!h->tb_stream && !getenv("AFP_TABLE_PLAIN_STREAM")) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&h->tb_stream, hipStreamNonBlocking, greatest) != hipSuccess) h->tb_stream = nullptr;
    }
    HIPCHK(tb_sync(h));
    const int64_t nb = (int64_t)1 << hashbits;
    ENSURE(h->tb_table, nb * depth * 4);
    ENSURE(h->tb_counts, nb * 4);
    HIPCHK(hipMemsetAsync(h->tb_table.p, 0, nb * depth * 4, tbs(h)));
    HIPCHK(hipMemsetAsync(h->tb_counts.p, 0, nb * 4, tbs(h)));
    h->tb_hashbits = hashbits; h->tb_depth = depth; h->tb_maxtimebits = maxtimebits;
    h->tb_novf = 0;
    h->pk_total = -1;
    return AFP_OK;
}
extern "C" int afp_table_upload(afp_handle* h, const uint32_t* table, const int32_t* counts)
{
    if (!h || !table || !counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    h->pk_total = -1;
    HIPCHK(hipMemcpyAsync(h->tb_table.p, table, nb * h->tb_depth * 4, hipMemcpyHostToDevice, tbs(h)));
    HIPCHK(hipMemcpyAsync(h->tb_counts.p, counts, nb * 4, hipMemcpyHostToDevice, tbs(h)));
    HIPCHK(tb_sync(h));
    return AFP_OK;
}
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
static HostPool* host_pool()
{
    static std::mutex mu;
    static HostPool* pool = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (pool && pool->pid == getpid()) return pool;
    HostPool* np = new HostPool();          // (a pool inherited through fork is abandoned, not destroyed: its threads are gone)
    np->pid = getpid();
    int want;
    const char* e = getenv("AFP_DL_THREADS");
    if (e) want = atoi(e);
    else {
        cpu_set_t set;
        CPU_ZERO(&set);
        const int nc = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        want = std::min(8, std::max(1, nc));
    }
    want = std::max(1, std::min(want, 64));
    for (int w = 1; w < want; w++) {
        try { np->th.emplace_back([np, w]() { np->worker(w); }); }
        catch (...) { break; }
    }
    for (auto& t : np->th) t.detach();      // they sleep on the condition variable until the process ends
    np->W = 1 + (int)np->th.size();
    pool = np;
    return pool;
}
extern "C" int afp_host_threads(void) { return host_pool()->W; }

// Populate the pages of a (large, freshly allocated) host array in the BACKGROUND: a HashTable's table is 420 MB of
// np.zeros -- untouched zero pages -- and the first write to each page costs a fault plus the kernel's zero fill; left to the
// table download at the end of a job that is 4-5 ms of its 6 (the scatter touches every page).  MADV_POPULATE_WRITE (Linux
// 5.14) faults the range in without changing its contents; a few detached threads do it while the device works on the
// job's first batches.  Best effort: an older kernel (EINVAL), a range that goes away meanwhile (ENOMEM) or a failed thread
// start just leave the pages to be faulted by their first real write, as before.  Returns the threads started.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
extern "C" int afp_host_prefault(void* p, int64_t bytes)
{
    if (!p || bytes <= 0) return 0;
    static const bool off = getenv("AFP_NO_PREFAULT") != nullptr;
    if (off) return 0;
    const uintptr_t a0 = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)p + (uintptr_t)bytes) & ~(uintptr_t)4095;
    if (a1 <= a0) return 0;
    // The caller asks for this when it knows the host will sit idle meanwhile (a pipelined job waiting for its first
    // batches): while the threads populate, OTHER runtime calls of the process crawl -- a table store issued right behind the
    // TableBuilder's creation took 3.7 ms instead of 1.3 with the pool's eight threads (5 ms of populating) and 17 ms with two
    // threads (20 ms of it): bench.py table_build, r05.  So: as many threads as the pool has (AFP_PREFAULT_THREADS), a short
    // window, and TableBuilder does not start it unless told to (prefault=True).
    static const int want_th = []() { const char* e = getenv("AFP_PREFAULT_THREADS"); const int v = e ? atoi(e) : host_pool()->W; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(want_th, (int64_t)(a1 - a0) >> 25));
    const uintptr_t per = (((a1 - a0) / nth) + 4095) & ~(uintptr_t)4095;
    int started = 0;
    for (int t = 0; t < nth; t++) {
        const uintptr_t lo = a0 + per * t, hi = std::min<uintptr_t>(a1, lo + per);
        if (hi <= lo) break;
        try {
            std::thread([lo, hi]() {
                for (uintptr_t q = lo; q < hi; q += (uintptr_t)1 << 20)          // in 1 MB steps: a vanished range stops the loop early
                    if (madvise((void*)q, (size_t)std::min<uintptr_t>((uintptr_t)1 << 20, hi - q), MADV_POPULATE_WRITE) != 0) break;
            }).detach();
            started++;
        } catch (...) { break; }
    }
    return started;
}

// The ring both downloads stage through: R pinned chunks of CH bytes, an event per slot
static constexpr int DL_R = 4;
static constexpr int64_t DL_CH = (int64_t)8 << 20;
static int dl_ring(afp_handle* h)
{
    if (!h->h_dl) HIPCHK(hipHostMalloc(&h->h_dl, (size_t)(DL_R * DL_CH), hipHostMallocDefault));
    for (int i = 0; i < DL_R; i++) if (!h->dl_ev[i]) HIPCHK(hipEventCreateWithFlags(&h->dl_ev[i], hipEventDisableTiming));
    return AFP_OK;
}
// `bytes` of device memory through the ring; `consume(k, n, ring_chunk, w, W)` runs on every pool thread for chunk k (n bytes)
// once it has landed.  Only the calling thread talks to the runtime.
template <class F>
static int ring_download(afp_handle* h, const char* src, int64_t bytes, hipStream_t st, F consume)
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

// Device -> PAGEABLE host memory, large: the copy engine fills the ring and the pool's threads move each chunk on into the
// destination -- plain memcpy, whose page faults on a freshly allocated numpy array then run in parallel too.  The runtime's
// own pageable path does the same with one thread: ~17 GB/s.
static int download_pageable(afp_handle* h, char* dst, const char* src, int64_t bytes, hipStream_t st)
{
    if (host_pool()->W <= 1 || bytes < 4 * DL_CH) {
        HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, st));
        return AFP_OK;
    }
    return ring_download(h, src, bytes, st, [&](int64_t k, int64_t n, const char* chunk, int w, int W) {
        const int64_t per = ((n + W - 1) / W + 4095) & ~(int64_t)4095;
        const int64_t a = std::min<int64_t>(n, w * per), b = std::min<int64_t>(n, a + per);
        if (b > a) memcpy(dst + k * DL_CH + a, chunk + a, (size_t)(b - a));
    });
}

extern "C" int afp_table_download(afp_handle* h, uint32_t* table, int32_t* counts)
{
    if (!h || !table || !counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    const int64_t bytes = nb * h->tb_depth * 4;
    HIPCHK(hipStreamSynchronize(tbs(h)));                       // stores / patches / merges queued on the handle's stream
    // The destination is the HashTable's own numpy array: pageable memory, which the runtime fills through its bounce
    // buffers at about 17 GB/s (25 ms for the default 420 MB table).  r04: slices copied by four host threads that each
    // called hipMemcpyAsync on a stream of their own faulted inside the runtime (every thread) -- so the runtime is driven
    // from this thread only and the helpers just memcpy (download_pageable).
    HIPCHK(hipMemcpyAsync(counts, h->tb_counts.p, nb * 4, hipMemcpyDeviceToHost, tbs(h)));
    { const int r = download_pageable(h, (char*)table, (const char*)h->tb_table.p, bytes, tbs(h)); if (r != AFP_OK) return r; }
    HIPCHK(tb_sync(h));
    drain_retired(false);
    return AFP_OK;
}

// ---- the PACKED table (k_table.hip: k_tb_pack_*): filled prefixes only --------------------------------------------------
// len[k] = min(counts[k], depth) -> exclusive scan (tb_pkoff, nb + 1 entries) -> gather into tb_packed.  Queued on the
// table's stream; `total_hint` (entries, when the caller already knows them: the host has the counts) sizes the buffer
// without a round trip, otherwise the total is read back.
static int table_pack(afp_handle* h, int64_t total_hint, int64_t* total)
{
    hipStream_t st = tbs(h);
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    ENSURE(h->tb_pklen, nb * 8);
    ENSURE(h->tb_pkoff, (nb + 1) * 8);
    ENSURE(h->tb_scan, (nb / 2048 + 2) * 8);
    afp_launch_tb_pack_len((const int32_t*)h->tb_counts.p, h->tb_hashbits, h->tb_depth, (int64_t*)h->tb_pklen.p, st);
    afp_launch_excl_scan64_wide((const int64_t*)h->tb_pklen.p, (int64_t*)h->tb_pkoff.p, (int)nb, (int64_t*)h->tb_scan.p, st);
    HIPCHK(hipGetLastError());
    int64_t tot = total_hint;
    if (tot < 0) {
        HIPCHK(hipMemcpyAsync(&tot, (int64_t*)h->tb_pkoff.p + nb, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    { int r_ = ensure(h->tb_packed, (size_t)std::max<int64_t>(tot, 1) * 4, true); if (r_ != AFP_OK) return r_; }
    afp_launch_tb_pack_gather((const uint32_t*)h->tb_table.p, (const int64_t*)h->tb_pkoff.p, h->tb_hashbits, h->tb_depth,
                              (uint32_t*)h->tb_packed.p, st);
    HIPCHK(hipGetLastError());
    if (total) *total = tot;
    return AFP_OK;
}
extern "C" int afp_table_pack(afp_handle* h, int64_t* total)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    int64_t tot = 0;
    { const int r = table_pack(h, -1, &tot); if (r != AFP_OK) return r; }
    HIPCHK(tb_sync(h));
    h->pk_total = tot;
    if (total) *total = tot;
    return AFP_OK;
}
extern "C" int afp_table_packed_device_ptrs(afp_handle* h, uint32_t** d_values, int32_t** d_counts, int64_t* total)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits || h->pk_total < 0) return AFP_ERR_STATE;
    if (d_values) *d_values = (uint32_t*)h->tb_packed.p;
    if (d_counts) *d_counts = (int32_t*)h->tb_counts.p;
    if (total) *total = h->pk_total;
    return AFP_OK;
}
extern "C" int afp_table_fetch_packed(afp_handle* h, uint32_t* values, int32_t* counts)
{
    if (!h || !counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits || h->pk_total < 0) return AFP_ERR_STATE;
    if (h->pk_total > 0 && !values) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    HIPCHK(hipMemcpyAsync(counts, h->tb_counts.p, nb * 4, hipMemcpyDeviceToHost, tbs(h)));
    if (h->pk_total > 0) {
        const int r = download_pageable(h, (char*)values, (const char*)h->tb_packed.p, h->pk_total * 4, tbs(h));
        if (r != AFP_OK) return r;
    }
    HIPCHK(tb_sync(h));
    return AFP_OK;
}

// afp_table_download for a host array that was IN STEP with the device table when the table was created or uploaded: only
// counts[] and table[k][0 .. min(counts[k], depth)) are written -- every other slot holds on the device what it held then
// (store / merge / patch never write it), i.e. what the host array still holds.  Counts leave through a pinned buffer, the
// pool's threads copy them out and build the offsets (two passes: per-thread sums, then the prefix), the packed values follow
// through the ring and each thread scatters its share of every chunk into the rows.  The c4 job's table: 4 + 32 MB over the
// link instead of 424.
extern "C" int afp_table_download_filled(afp_handle* h, uint32_t* table, int32_t* counts, int64_t* n_entries)
{
    if (!h || !table || !counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    const int depth = h->tb_depth;
    if ((size_t)nb * 4 > h->h_dlc_cap) {
        if (h->h_dlc) { HIPCHK(hipStreamSynchronize(st)); (void)hipHostFree(h->h_dlc); h->h_dlc = nullptr; h->h_dlc_cap = 0; }
        HIPCHK(hipHostMalloc(&h->h_dlc, (size_t)nb * 4, hipHostMallocDefault));
        h->h_dlc_cap = (size_t)nb * 4;
    }
    if (!h->dlc_ev) HIPCHK(hipEventCreateWithFlags(&h->dlc_ev, hipEventDisableTiming));
    { const int r = dl_ring(h); if (r != AFP_OK) return r; }
    // counts first (they size everything), the length / offset kernels behind them on the same stream
    HIPCHK(hipMemcpyAsync(h->h_dlc, h->tb_counts.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(h->dlc_ev, st));
    ENSURE(h->tb_pklen, nb * 8);
    ENSURE(h->tb_pkoff, (nb + 1) * 8);
    ENSURE(h->tb_scan, (nb / 2048 + 2) * 8);
    afp_launch_tb_pack_len((const int32_t*)h->tb_counts.p, h->tb_hashbits, depth, (int64_t*)h->tb_pklen.p, st);
    afp_launch_excl_scan64_wide((const int64_t*)h->tb_pklen.p, (int64_t*)h->tb_pkoff.p, (int)nb, (int64_t*)h->tb_scan.p, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(h->dlc_ev));
    HostPool* P = host_pool();
    const int W = P->W;
    const int32_t* hc = (const int32_t*)h->h_dlc;
    std::vector<int64_t>& off = h->pk_hoff;
    off.resize((size_t)nb + 1);
    std::vector<int64_t> part((size_t)W + 1, 0);
    auto range = [&](int w, int64_t& a, int64_t& b) { a = nb * w / W; b = nb * (w + 1) / W; };
    P->run([&](int w) {
        int64_t a, b; range(w, a, b);
        memcpy(counts + a, hc + a, (size_t)(b - a) * 4);
        int64_t s = 0;
        for (int64_t k = a; k < b; k++) { const int32_t c = hc[k]; s += c < 0 ? 0 : c < depth ? c : depth; }
        part[(size_t)w + 1] = s;
    });
    for (int w = 0; w < W; w++) part[(size_t)w + 1] += part[(size_t)w];
    const int64_t total = part[(size_t)W];
    P->run([&](int w) {
        int64_t a, b; range(w, a, b);
        int64_t s = part[(size_t)w];
        for (int64_t k = a; k < b; k++) { off[(size_t)k] = s; const int32_t c = hc[k]; s += c < 0 ? 0 : c < depth ? c : depth; }
    });
    off[(size_t)nb] = total;
    if (n_entries) *n_entries = total;
    if (total > 0) {
        { int r_ = ensure(h->tb_packed, (size_t)total * 4, true); if (r_ != AFP_OK) return r_; }
        afp_launch_tb_pack_gather((const uint32_t*)h->tb_table.p, (const int64_t*)h->tb_pkoff.p, h->tb_hashbits, depth,
                                  (uint32_t*)h->tb_packed.p, st);
        HIPCHK(hipGetLastError());
        const int64_t E = DL_CH / 4;                                  // entries per chunk
        const int r = ring_download(h, (const char*)h->tb_packed.p, total * 4, st,
            [&](int64_t k, int64_t n, const char* chunk, int w, int Wn) {
                const int64_t ne = n / 4, e0 = k * E;
                int64_t a = e0 + ne * w / Wn, b = e0 + ne * (w + 1) / Wn;      // this thread's entries [a, b) of the packed stream
                if (b <= a) return;
                // bucket holding entry a: the last i with off[i] <= a
                int64_t lo = 0, hi = nb;
                while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (off[(size_t)mid] <= a) lo = mid; else hi = mid; }
                const uint32_t* v = (const uint32_t*)chunk;                  // entry e of the stream sits at v[e - e0]
                for (int64_t i = lo; a < b; i++) {
                    const int64_t end = std::min<int64_t>(off[(size_t)i + 1], b);
                    if (end > a) {
                        memcpy(table + i * depth + (a - off[(size_t)i]), v + (a - e0), (size_t)(end - a) * 4);
                        a = end;
                    }
                }
            });
        if (r != AFP_OK) return r;
    }
    HIPCHK(tb_sync(h));
    h->pk_total = total;
    drain_retired(false);
    return AFP_OK;
}
// rows / clip offsets already in HBM -> table; N rows
static int table_store_rows(afp_handle* h, const int32_t* d_rows, const int64_t* d_clip_off, int64_t N, const int32_t* clip_ids,
                            int32_t nclips, int64_t* n_overflow)
{
    hipStream_t st = tbs(h);
    if (n_overflow) *n_overflow = 0;
    h->tb_novf = 0;
    h->pk_total = -1;
    if (N == 0 || nclips == 0) return AFP_OK;
    TableArgs a;
    a.rows = d_rows; a.clip_off = d_clip_off;
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    ENSURE(h->tb_ids, (int64_t)nclips * 4);
    ENSURE(h->tb_newcnt, (nb + 1) * 8);
    ENSURE(h->tb_first, (nb + 1) * 8);
    ENSURE(h->tb_fill, nb * 4);
    { int r_ = ensure(h->tb_seg, (size_t)N * 8, true); if (r_ != AFP_OK) return r_; }
    { int r_ = ensure(h->tb_overflow, (size_t)N * 16, true); if (r_ != AFP_OK) return r_; }
    ENSURE(h->tb_biglist, nb * 4);
    ENSURE(h->tb_misc, 256);
    HIPCHK(hipMemcpyAsync(h->tb_ids.p, clip_ids, (size_t)nclips * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(h->tb_newcnt.p, 0, (nb + 1) * 8, st));
    HIPCHK(hipMemsetAsync(h->tb_fill.p, 0, nb * 4, st));
    HIPCHK(hipMemsetAsync(h->tb_misc.p, 0, 256, st));
    a.clip_ids = (const int32_t*)h->tb_ids.p; a.nrows = N; a.nclips = nclips;
    a.hashbits = h->tb_hashbits; a.depth = h->tb_depth; a.maxtimebits = h->tb_maxtimebits;
    a.table = (uint32_t*)h->tb_table.p; a.counts = (int32_t*)h->tb_counts.p;
    a.newcnt = (int64_t*)h->tb_newcnt.p; a.first = (int64_t*)h->tb_first.p; a.fill = (int32_t*)h->tb_fill.p;
    a.seg = (unsigned long long*)h->tb_seg.p; a.overflow = (int32_t*)h->tb_overflow.p;
    a.ovcnt = (int32_t*)h->tb_misc.p; a.bigcnt = (int32_t*)h->tb_misc.p + 16; a.biglist = (int32_t*)h->tb_biglist.p;
    afp_launch_tb_count(&a, st);
    if (nb >= 65536) {
        ENSURE(h->tb_scan, (nb / 2048 + 2) * 8);
        afp_launch_excl_scan64_wide((const int64_t*)h->tb_newcnt.p, (int64_t*)h->tb_first.p, (int)nb, (int64_t*)h->tb_scan.p, st);
    } else afp_launch_excl_scan64((const int64_t*)h->tb_newcnt.p, (int64_t*)h->tb_first.p, (int)nb, st);
    afp_launch_tb_scatter(&a, st);
    afp_launch_tb_fill(&a, st);
    afp_launch_tb_fill_big(&a, st);
    HIPCHK(hipGetLastError());
    int32_t novf = 0;
    HIPCHK(hipMemcpyAsync(&novf, h->tb_misc.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->tb_novf = novf;
    if (n_overflow) *n_overflow = novf;
    return AFP_OK;
}
extern "C" int afp_table_store(afp_handle* h, const int32_t* rows, const int64_t* clip_off, const int32_t* clip_ids,
                               int32_t nclips, int64_t* n_overflow)
{
    if (!h || nclips < 0 || (nclips > 0 && !clip_ids)) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    if (rows) {                                   // host rows (e.g. loaded from .afpt files)
        if (!clip_off) return AFP_ERR_ARG;
        const int64_t N = clip_off[nclips] - clip_off[0];
        if (N < 0 || N > 0x7fffffffLL) return AFP_ERR_ARG;
        ENSURE(h->tb_rows, (N > 0 ? N : 1) * 8);
        ENSURE(h->tb_off, (int64_t)(nclips + 1) * 8);
        std::vector<int64_t> rel((size_t)nclips + 1);
        for (int c = 0; c <= nclips; c++) rel[c] = clip_off[c] - clip_off[0];
        if (N > 0) HIPCHK(hipMemcpyAsync(h->tb_rows.p, rows + 2 * clip_off[0], N * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(h->tb_off.p, rel.data(), (size_t)(nclips + 1) * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return table_store_rows(h, (const int32_t*)h->tb_rows.p, (const int64_t*)h->tb_off.p, N, clip_ids, nclips, n_overflow);
    }
    // the (time, hash) rows of the last extract, still in HBM
    if (!h->extracted || !(h->flags & AFP_WANT_HASHES) || nclips != h->nclips) return AFP_ERR_STATE;
    FINALIZE(h);
    if (h->total_hashes > 0x7fffffffLL) return AFP_ERR_ARG;
    return table_store_rows(h, (const int32_t*)h->out_hashes.p, (const int64_t*)h->clip_hoff.p, h->total_hashes, clip_ids, nclips, n_overflow);
}
// the same from rows that already sit in HBM and belong to somebody else -- typically ANOTHER handle's results
// (afp_result_device_ptrs after afp_result_counts, which has waited for them): several extraction contexts feed one table
extern "C" int afp_table_store_device(afp_handle* h, const int32_t* d_rows, const int64_t* d_clip_off, int64_t nrows,
                                      const int32_t* clip_ids, int32_t nclips, int64_t* n_overflow)
{
    if (!h || nclips < 0 || nrows < 0 || nrows > 0x7fffffffLL || (nclips > 0 && (!clip_ids || !d_clip_off)) || (nrows > 0 && !d_rows)) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    return table_store_rows(h, d_rows, d_clip_off, nrows, clip_ids, nclips, n_overflow);
}

// ---- the random replacements of HashTable.store (hash_table.py:125-131), replayed on the host ---------------------
// The reference draws `random.randint(0, count)` from Python's GLOBAL Mersenne Twister for every insertion into a full
// bucket, in insertion order.  CPython: randint(a, b) -> randrange(a, b + 1) -> _randbelow_with_getrandbits(n = b + 1 - a):
// k = n.bit_length(); r = getrandbits(k) until r < n; getrandbits(k <= 32) = genrand_uint32() >> (32 - k)
// (Lib/random.py, Modules/_randommodule.c).  The same stream is produced here from the 624 state words + position that
// random.getstate() hands out; the caller puts the advanced state back with random.setstate(), so every later draw of the
// process continues as if Python had made these calls itself (audfprint_amd/table.py checks the equivalence once per process
// against Python's own generator and falls back to the Python loop if it ever differs).
static inline uint32_t mt_next(uint32_t* mt, int32_t& pos)
{
    if (pos >= 624) {
        static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u]; }
        for (; kk < 623; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u]; }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static inline int32_t mt_randint0(uint32_t* mt, int32_t& pos, int32_t count)      // random.randint(0, count), count >= 0
{
    const uint32_t n = (uint32_t)count + 1u;
    const int k = 32 - __builtin_clz(n);             // n.bit_length(), n >= 1
    uint32_t r;
    do { r = mt_next(mt, pos) >> (32 - k); } while (r >= n);
    return (int32_t)r;
}
extern "C" int afp_mt_randint_replay(uint32_t* mt_state, int32_t* mt_pos, const int32_t* counts, int64_t n, int32_t* out)
{
    if (!mt_state || !mt_pos || n < 0 || (n > 0 && (!counts || !out)) || *mt_pos < 0 || *mt_pos > 624) return AFP_ERR_ARG;
    int32_t pos = *mt_pos;
    for (int64_t i = 0; i < n; i++) {
        if (counts[i] < 0) return AFP_ERR_ARG;
        out[i] = mt_randint0(mt_state, pos, counts[i]);
    }
    *mt_pos = pos;
    return AFP_OK;
}
// Everything HashTable.store does with the overflow events of the last afp_table_store*: fetch them, put them in insertion
// order (row order), draw slot = random.randint(0, count) for each from the given Mersenne-Twister state (:128), keep the draws
// with slot < depth (:130-131; of several writes to one (bucket, slot) the LAST wins, as in the loop) and patch them into the
// device table.  mt_state / mt_pos are advanced exactly as Python's generator would be.  n_written: slots patched.
extern "C" int afp_table_replay_overflow(afp_handle* h, uint32_t* mt_state, int32_t* mt_pos, int64_t* n_written)
{
    if (!h || !mt_state || !mt_pos || *mt_pos < 0 || *mt_pos > 624) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (n_written) *n_written = 0;
    const int64_t n = h->tb_novf;
    if (n == 0) return AFP_OK;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    struct Ev { uint32_t row; int32_t bucket; int32_t val; int32_t count; };
    static_assert(sizeof(Ev) == 16, "event layout of k_tb_fill");
    if ((size_t)n * 28 > h->h_ovf_cap) {                 // 16 n bytes of events + up to 12 n of patches
        HIPCHK(hipStreamSynchronize(st));                 // (the previous replay's patch upload reads the old buffer)
        if (h->h_ovf) (void)hipHostFree(h->h_ovf);
        h->h_ovf = nullptr; h->h_ovf_cap = 0;
        // (grown geometrically: a long ingest meets more full buckets batch after batch, and every re-allocation of pinned
        //  memory costs more than the draws of a batch)
        const size_t want = std::max<size_t>((size_t)n * 56, (size_t)4 << 20);
        HIPCHK(hipHostMalloc(&h->h_ovf, want, hipHostMallocDefault));
        h->h_ovf_cap = want;
    }
    static const bool prof = getenv("AFP_REPLAY_PROF") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = prof ? now() : 0.0;
    Ev* ev = (Ev*)h->h_ovf;
    HIPCHK(hipMemcpyAsync(ev, h->tb_overflow.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const double tp1 = prof ? now() : 0.0;
    // insertion order = row order (rows are distinct).  LSD radix sort of the event indices by row, 11 bits a pass (a comparison
    // sort of the 16-byte records took 55 ns per event -- five times the draws)
    std::vector<uint32_t>& ord = h->ovf_ord;
    std::vector<uint32_t>& tmp = h->ovf_tmp;
    ord.resize((size_t)n); tmp.resize((size_t)n);
    uint32_t maxrow = 0;
    const int64_t nbk = (int64_t)1 << h->tb_hashbits;
    for (int64_t i = 0; i < n; i++) {
        ord[(size_t)i] = (uint32_t)i;
        if (ev[i].row > maxrow) maxrow = ev[i].row;
        // a malformed event is refused HERE, before a single draw: the generator state and the table are untouched
        if (ev[i].count < 0 || ev[i].bucket < 0 || ev[i].bucket >= nbk) {
            g_hip_err = "afp_table_replay_overflow: malformed overflow event; nothing was drawn, nothing was patched";
            return AFP_ERR_STATE;
        }
    }
    for (int shift = 0; shift < 32 && (maxrow >> shift) != 0; shift += 11) {
        uint32_t cnt[2049];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < n; i++) cnt[((ev[ord[(size_t)i]].row >> shift) & 2047u) + 1]++;
        for (int d = 0; d < 2048; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; i++) { const uint32_t e = ord[(size_t)i]; tmp[cnt[(ev[e].row >> shift) & 2047u]++] = e; }
        ord.swap(tmp);
    }
    const double tp2 = prof ? now() : 0.0;
    const int depth = h->tb_depth;
    // the draws advance a COPY of the generator; the caller's state is replaced only once the patches are queued
    uint32_t mt[624];
    memcpy(mt, mt_state, sizeof(mt));
    int32_t pos = *mt_pos;
    // slot per event (indexed like ev), drawn in insertion order; -1 = not kept
    std::vector<int32_t>& slot = h->ovf_slot;
    slot.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const uint32_t e = ord[(size_t)i];
        const int32_t sl = mt_randint0(mt, pos, ev[e].count);
        slot[(size_t)e] = sl < depth ? sl : -1;
    }
    const double tp3 = prof ? now() : 0.0;
    // last write per (bucket, slot) wins: walk backwards, remember the cells already taken -- in a small open-addressing set
    // sized for THIS batch's kept draws (r04: a bit per table cell, 13 MB, cost a DRAM miss per kept draw: 18 of the c4 job's
    // 93 ms)
    int64_t nkept = 0;
    for (int64_t i = 0; i < n; i++) nkept += slot[(size_t)i] >= 0 ? 1 : 0;
    size_t tsz = 1024;
    while (tsz < (size_t)nkept * 4) tsz <<= 1;
    std::vector<uint64_t>& seen = h->ovf_seen;
    seen.assign(tsz, 0ull);                                          // key = cell + 1
    std::vector<int32_t>& patch = h->ovf_patch;
    patch.clear();
    for (int64_t k = n - 1; k >= 0; k--) {
        const uint32_t i = ord[(size_t)k];
        if (slot[(size_t)i] < 0) continue;
        const uint64_t key = (uint64_t)((int64_t)ev[i].bucket * depth + slot[(size_t)i]) + 1ull;
        size_t p = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (tsz - 1);
        bool dup = false;
        while (seen[p] != 0ull) { if (seen[p] == key) { dup = true; break; } p = (p + 1) & (tsz - 1); }
        if (dup) continue;
        seen[p] = key;
        patch.push_back(ev[i].bucket); patch.push_back(slot[(size_t)i]); patch.push_back(ev[i].val);
    }
    const double tp4 = prof ? now() : 0.0;
    const int64_t np = (int64_t)patch.size() / 3;
    if (np > 0) {
        // (never a small allocation: growing a device buffer means hipFree, which waits for EVERY stream of the device -- measured
        //  3.4 ms in the middle of the pipelined c4 job, seven times the replay itself)
        ENSURE(h->tb_patch, std::max<int64_t>(np * 12 * 2, (int64_t)4 << 20));
        // the patches leave through the tail of the pinned event buffer (np <= n: 12 n bytes behind the 16 n of the events), so
        // nothing has to be waited for here: the copy and the kernel are ordered on the table's stream in front of whatever
        // touches the table next, and the next replay writes the buffer only after its own events have arrived behind them
        int32_t* pp = reinterpret_cast<int32_t*>((char*)h->h_ovf + (size_t)n * 16);
        memcpy(pp, patch.data(), (size_t)np * 12);
        HIPCHK(hipMemcpyAsync(h->tb_patch.p, pp, (size_t)np * 12, hipMemcpyHostToDevice, st));
        afp_launch_tb_patch((uint32_t*)h->tb_table.p, depth, (const int32_t*)h->tb_patch.p, np, st);
        HIPCHK(hipGetLastError());
    }
    if (prof) fprintf(stderr, "replay n=%lld kept=%lld: fetch %.0f us, order %.0f, draws %.0f, dedupe %.0f, patch %.0f\n", (long long)n, (long long)np, tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3, now() - tp4);
    if (n_written) *n_written = np;
    memcpy(mt_state, mt, sizeof(mt));                     // commit: table and generator advance together
    *mt_pos = pos;
    h->pk_total = -1;
    h->tb_novf = 0;                                       // the events are consumed: a second replay must not draw again
    return AFP_OK;
}
extern "C" int afp_table_fetch_overflow(afp_handle* h, int32_t* events)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (h->tb_novf == 0) return AFP_OK;
    if (!events) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(events, h->tb_overflow.p, h->tb_novf * 16, hipMemcpyDeviceToHost, tbs(h)));
    HIPCHK(tb_sync(h));
    return AFP_OK;
}

// ---- HashTable.merge (hash_table.py:291-323) into the device table ------------------------------------
static int table_merge_device(afp_handle* h, const uint32_t* d_ot, const int32_t* d_oc, const int64_t* d_ooff, int32_t odepth,
                              int32_t ncurrent, int64_t* n_overflow)
{
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (odepth < 1 || odepth > 4096 || ncurrent < 0) return AFP_ERR_PARAM;
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    hipStream_t st = tbs(h);
    ENSURE(h->tb_mlist, nb * 4);
    ENSURE(h->tb_misc, 256);
    HIPCHK(hipMemsetAsync(h->tb_misc.p, 0, 256, st));
    h->pk_total = -1;
    h->mg_otable = d_ot; h->mg_ocounts = d_oc; h->mg_ooff = d_ooff; h->mg_odepth = odepth;
    h->mg_idoffset = (uint32_t)ncurrent << h->tb_maxtimebits;            // :300  idoffset = (1 << maxtimebits) * ncurrent
    afp_launch_tb_merge((uint32_t*)h->tb_table.p, (int32_t*)h->tb_counts.p, d_ot, d_oc, d_ooff, h->tb_hashbits, h->tb_depth, odepth,
                        h->mg_idoffset, (int32_t*)h->tb_mlist.p, (int32_t*)h->tb_misc.p + 32, st);
    HIPCHK(hipGetLastError());
    int32_t nov = 0;
    HIPCHK(hipMemcpyAsync(&nov, (int32_t*)h->tb_misc.p + 32, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->mg_nov = nov;
    if (n_overflow) *n_overflow = nov;
    return AFP_OK;
}
extern "C" int afp_table_merge_device(afp_handle* h, const uint32_t* d_other_table, const int32_t* d_other_counts,
                                      int32_t other_depth, int32_t ncurrent, int64_t* n_overflow)
{
    if (!h || !d_other_table || !d_other_counts) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(tb_sync(h));
    return table_merge_device(h, d_other_table, d_other_counts, nullptr, other_depth, ncurrent, n_overflow);
}
extern "C" int afp_table_merge(afp_handle* h, const uint32_t* other_table, const int32_t* other_counts, int32_t other_depth,
                               int32_t ncurrent, int64_t* n_overflow)
{
    if (!h || !other_table || !other_counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (other_depth < 1 || other_depth > 4096) return AFP_ERR_PARAM;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(tb_sync(h));
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    ENSURE(h->tb_otable, nb * other_depth * 4);
    ENSURE(h->tb_ocounts, nb * 4);
    HIPCHK(hipMemcpyAsync(h->tb_otable.p, other_table, nb * other_depth * 4, hipMemcpyHostToDevice, tbs(h)));
    HIPCHK(hipMemcpyAsync(h->tb_ocounts.p, other_counts, nb * 4, hipMemcpyHostToDevice, tbs(h)));
    return table_merge_device(h, (const uint32_t*)h->tb_otable.p, (const int32_t*)h->tb_ocounts.p, nullptr, other_depth, ncurrent, n_overflow);
}
// The same from the other table's PACKED form (afp_table_pack on the sending side): its counts and the filled prefixes of
// its rows, bucket after bucket -- min(counts[k], other_depth) entries each.  The row offsets are rebuilt here (one length
// kernel + the scan).  Device pointers (a table that came over xGMI) must stay valid until afp_table_fetch_merge_overflow.
static int merge_packed_device(afp_handle* h, const uint32_t* d_vals, const int32_t* d_oc, int32_t odepth, int32_t ncurrent, int64_t* n_overflow)
{
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    hipStream_t st = tbs(h);
    ENSURE(h->tb_olen, nb * 8);
    ENSURE(h->tb_ooff, (nb + 1) * 8);
    ENSURE(h->tb_scan, (nb / 2048 + 2) * 8);
    afp_launch_tb_pack_len(d_oc, h->tb_hashbits, odepth, (int64_t*)h->tb_olen.p, st);
    afp_launch_excl_scan64_wide((const int64_t*)h->tb_olen.p, (int64_t*)h->tb_ooff.p, (int)nb, (int64_t*)h->tb_scan.p, st);
    HIPCHK(hipGetLastError());
    return table_merge_device(h, d_vals, d_oc, (const int64_t*)h->tb_ooff.p, odepth, ncurrent, n_overflow);
}
extern "C" int afp_table_merge_packed_device(afp_handle* h, const uint32_t* d_other_values, const int32_t* d_other_counts,
                                             int32_t other_depth, int32_t ncurrent, int64_t* n_overflow)
{
    if (!h || !d_other_counts) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (other_depth < 1 || other_depth > 4096) return AFP_ERR_PARAM;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(tb_sync(h));
    return merge_packed_device(h, d_other_values, d_other_counts, other_depth, ncurrent, n_overflow);
}
extern "C" int afp_table_merge_packed(afp_handle* h, const uint32_t* other_values, int64_t n_values, const int32_t* other_counts,
                                      int32_t other_depth, int32_t ncurrent, int64_t* n_overflow)
{
    if (!h || !other_counts || n_values < 0 || (n_values > 0 && !other_values)) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (other_depth < 1 || other_depth > 4096) return AFP_ERR_PARAM;
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    {   // the packed stream must hold exactly what the counts announce (checked BEFORE anything is uploaded or merged)
        int64_t want = 0;
        for (int64_t k = 0; k < nb; k++) { const int32_t c = other_counts[k]; if (c < 0) return AFP_ERR_ARG; want += c < other_depth ? c : other_depth; }
        if (want != n_values) return AFP_ERR_ARG;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(tb_sync(h));
    ENSURE(h->tb_otable, std::max<int64_t>(n_values, 1) * 4);
    ENSURE(h->tb_ocounts, nb * 4);
    if (n_values > 0) HIPCHK(hipMemcpyAsync(h->tb_otable.p, other_values, n_values * 4, hipMemcpyHostToDevice, tbs(h)));
    HIPCHK(hipMemcpyAsync(h->tb_ocounts.p, other_counts, nb * 4, hipMemcpyHostToDevice, tbs(h)));
    return merge_packed_device(h, (const uint32_t*)h->tb_otable.p, (const int32_t*)h->tb_ocounts.p, other_depth, ncurrent, n_overflow);
}
extern "C" int afp_table_fetch_merge_overflow(afp_handle* h, int32_t* buckets, int32_t* nvals, uint32_t* allvals)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    const int n = h->mg_nov;
    if (n == 0) return AFP_OK;
    if (!buckets || !nvals || !allvals || !h->mg_otable) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    // ascending bucket order = the order of the reference's loop over np.nonzero(ht.counts) (:302)
    std::vector<int32_t> list((size_t)n);
    HIPCHK(hipMemcpyAsync(list.data(), h->tb_mlist.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::sort(list.begin(), list.end());
    HIPCHK(hipMemcpyAsync(h->tb_mlist.p, list.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    const int64_t w = (int64_t)h->tb_depth + h->mg_odepth;
    ENSURE(h->tb_mvals, (int64_t)n * w * 4);
    ENSURE(h->tb_mnv, (int64_t)n * 4);
    afp_launch_tb_merge_gather((const uint32_t*)h->tb_table.p, (const int32_t*)h->tb_counts.p, h->mg_otable, h->mg_ocounts, h->mg_ooff,
                               h->tb_depth, h->mg_odepth, h->mg_idoffset, (const int32_t*)h->tb_mlist.p, n,
                               (uint32_t*)h->tb_mvals.p, (int32_t*)h->tb_mnv.p, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(allvals, h->tb_mvals.p, (int64_t)n * w * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(nvals, h->tb_mnv.p, (int64_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(buckets, list.data(), (size_t)n * 4);
    h->mg_otable = nullptr; h->mg_ocounts = nullptr; h->mg_ooff = nullptr; h->mg_nov = 0;      // the caller may free the other table now: a second fetch finds nothing
    return AFP_OK;
}
extern "C" int afp_table_patch(afp_handle* h, const int32_t* patches, int64_t n)
{
    if (!h || n < 0 || (n > 0 && !patches)) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (n == 0) return AFP_OK;
    const int64_t nb = (int64_t)1 << h->tb_hashbits;
    for (int64_t i = 0; i < n; i++)
        if (patches[3 * i] < 0 || patches[3 * i] >= nb || patches[3 * i + 1] < 0 || patches[3 * i + 1] >= h->tb_depth) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    h->pk_total = -1;
    ENSURE(h->tb_patch, n * 12);
    HIPCHK(hipMemcpyAsync(h->tb_patch.p, patches, n * 12, hipMemcpyHostToDevice, tbs(h)));
    afp_launch_tb_patch((uint32_t*)h->tb_table.p, h->tb_depth, (const int32_t*)h->tb_patch.p, n, tbs(h));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(tbs(h)));                 // `patches` is the caller's buffer
    return AFP_OK;
}
extern "C" int afp_table_clip_counts(afp_handle* h)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    h->pk_total = -1;
    afp_launch_tb_clip_counts((int32_t*)h->tb_counts.p, h->tb_hashbits, h->tb_depth, tbs(h));
    HIPCHK(hipGetLastError());
    return AFP_OK;
}
extern "C" int afp_table_device_ptrs(afp_handle* h, uint32_t** d_table, int32_t** d_counts)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(tb_sync(h));
    if (d_table) *d_table = (uint32_t*)h->tb_table.p;
    if (d_counts) *d_counts = (int32_t*)h->tb_counts.p;
    return AFP_OK;
}

// HashTable.get_hits (hash_table.py:150-176) over the device-resident table
extern "C" int afp_table_get_hits(afp_handle* h, const int32_t* rows, int64_t nrows, int64_t* nhits)
{
    if (!h || nrows < 0 || (nrows > 0 && !rows) || !nhits) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (nrows > 0x7fffffffLL) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    *nhits = 0; h->gh_total = 0;
    h->vt_counted = false; h->vt_hist_rows = 0; h->vs_total = -1;
    if (nrows == 0) return AFP_OK;
    hipStream_t st = tbs(h);
    ENSURE(h->gh_rows, nrows * 8);
    ENSURE(h->gh_nids, nrows * 8);
    ENSURE(h->gh_off, (nrows + 1) * 8);
    HIPCHK(hipMemcpyAsync(h->gh_rows.p, rows, nrows * 8, hipMemcpyHostToDevice, st));
    afp_launch_gh_count((const int32_t*)h->gh_rows.p, nrows, h->tb_hashbits, h->tb_depth, (const int32_t*)h->tb_counts.p,
                        (int64_t*)h->gh_nids.p, st);
    afp_launch_excl_scan64((const int64_t*)h->gh_nids.p, (int64_t*)h->gh_off.p, (int)nrows, st);
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, (int64_t*)h->gh_off.p + nrows, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    ENSURE(h->gh_hits, (total > 0 ? total : 1) * 16);
    afp_launch_gh_fill((const int32_t*)h->gh_rows.p, nrows, h->tb_hashbits, h->tb_depth, h->tb_maxtimebits,
                       (const uint32_t*)h->tb_table.p, (const int32_t*)h->tb_counts.p, (const int64_t*)h->gh_off.p,
                       (int32_t*)h->gh_hits.p, st);
    HIPCHK(hipGetLastError());
    h->gh_total = total;
    *nhits = total;
    return AFP_OK;
}
extern "C" int afp_table_fetch_hits(afp_handle* h, int32_t* hits)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    if (h->gh_total > 0) {
        if (!hits) return AFP_ERR_ARG;
        HIPCHK(hipMemcpyAsync(hits, h->gh_hits.p, h->gh_total * 16, hipMemcpyDeviceToHost, tbs(h)));
    }
    HIPCHK(tb_sync(h));
    return AFP_OK;
}

// ---- row f4, second half: vote counting over the resident hits -------------------------------------
static int vote_id_range(const afp_handle* h) { return 1 << (32 - h->tb_maxtimebits); }   // ids are (value >> maxtimebits) - 1

extern "C" int afp_table_count_ids(afp_handle* h, int64_t* n_ids)
{
    if (!h || !n_ids) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (h->tb_maxtimebits < 8) return AFP_ERR_PARAM;              // dense id histogram of at most 2^24 entries
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    const int nid = vote_id_range(h);
    const int64_t n = h->gh_total;
    *n_ids = 0;
    h->vt_nids = 0; h->vt_mintime = 0; h->vt_width = 0; h->vt_hist_rows = 0;
    h->vt_counted = true;
    if (n == 0) return AFP_OK;
    ENSURE(h->vt_idcount, (int64_t)nid * 4);
    ENSURE(h->vt_misc, 32);
    const int64_t cap = n < nid ? n : nid;
    ENSURE(h->vt_ids, cap * 4);
    ENSURE(h->vt_cnt, cap * 4);
    const int32_t init[8] = {0x7fffffff, -0x7fffffff - 1, 0, 0, -0x7fffffff - 1, 0, 0, 0};
    HIPCHK(hipMemsetAsync(h->vt_idcount.p, 0, (int64_t)nid * 4, st));
    HIPCHK(hipMemcpyAsync(h->vt_misc.p, init, 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));                           // `init` is a stack buffer
    afp_launch_vote_count((const int32_t*)h->gh_hits.p, n, nid, (int32_t*)h->vt_idcount.p, (int32_t*)h->vt_misc.p, st);
    afp_launch_vote_compact((const int32_t*)h->vt_idcount.p, nid, (int32_t*)h->vt_ids.p, (int32_t*)h->vt_cnt.p,
                            (int32_t*)h->vt_misc.p, st);
    int32_t misc[8];
    HIPCHK(hipMemcpyAsync(misc, h->vt_misc.p, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (misc[2]) return AFP_ERR_STATE;                           // an id outside the table's id range: not hits of this table
    h->vt_mintime = misc[0];
    h->vt_width = misc[1] - misc[0] + 1;
    h->vt_nids = misc[3];
    h->vt_maxotime = misc[4];
    *n_ids = misc[3];
    return AFP_OK;
}
// np.amax(hits[:, 3]) over the hits of the last afp_table_get_hits (after afp_table_count_ids): Matcher._unique_match_hashes packs
// time + (hash << timebits) with timebits = max(1, encpowerof2(that maximum)) (audfprint_match.py:157, 166-167)
extern "C" int afp_table_hits_max_time(afp_handle* h, int32_t* max_time)
{
    if (!h || !max_time) return AFP_ERR_ARG;
    if (!h->tb_hashbits || !h->vt_counted) return AFP_ERR_STATE;
    *max_time = h->gh_total > 0 ? h->vt_maxotime : 0;
    return AFP_OK;
}
extern "C" int afp_table_fetch_id_counts(afp_handle* h, int32_t* ids, int32_t* counts)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits || !h->vt_counted) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    if (h->vt_nids > 0) {
        if (!ids || !counts) return AFP_ERR_ARG;
        HIPCHK(hipMemcpyAsync(ids, h->vt_ids.p, (int64_t)h->vt_nids * 4, hipMemcpyDeviceToHost, tbs(h)));
        HIPCHK(hipMemcpyAsync(counts, h->vt_cnt.p, (int64_t)h->vt_nids * 4, hipMemcpyDeviceToHost, tbs(h)));
    }
    HIPCHK(tb_sync(h));
    return AFP_OK;
}
extern "C" int afp_table_skew_hist(afp_handle* h, const int32_t* ids, int32_t nids, int32_t* mintime, int32_t* width)
{
    if (!h || nids < 0 || (nids > 0 && !ids) || !mintime || !width) return AFP_ERR_ARG;
    if (!h->tb_hashbits || !h->vt_counted) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    const int nid = vote_id_range(h);
    *mintime = h->vt_mintime; *width = h->vt_width;
    h->vt_hist_rows = 0;
    if (nids == 0 || h->gh_total == 0) return AFP_OK;
    for (int i = 0; i < nids; i++) if (ids[i] < 0 || ids[i] >= nid) return AFP_ERR_ARG;
    const int64_t cells = (int64_t)nids * h->vt_width;
    if (cells > ((int64_t)1 << 28)) return AFP_ERR_NOMEM;
    ENSURE(h->vt_rank, (int64_t)nid * 4);
    ENSURE(h->vt_want, (int64_t)nids * 4);
    ENSURE(h->vt_hist, cells * 4);
    HIPCHK(hipMemsetAsync(h->vt_rank.p, 0xFF, (int64_t)nid * 4, st));
    HIPCHK(hipMemsetAsync(h->vt_hist.p, 0, cells * 4, st));
    HIPCHK(hipMemcpyAsync(h->vt_want.p, ids, (int64_t)nids * 4, hipMemcpyHostToDevice, st));
    afp_launch_vote_setrank((const int32_t*)h->vt_want.p, nids, nid, (int32_t*)h->vt_rank.p, st);
    afp_launch_vote_hist((const int32_t*)h->gh_hits.p, h->gh_total, nid, (const int32_t*)h->vt_rank.p, h->vt_mintime,
                         h->vt_width, (int32_t*)h->vt_hist.p, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));                           // `ids` is the caller's buffer
    h->vt_hist_rows = nids;
    return AFP_OK;
}
extern "C" int afp_table_fetch_skew_hist(afp_handle* h, int32_t* hist)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->tb_hashbits || !h->vt_counted) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    const int64_t cells = (int64_t)h->vt_hist_rows * h->vt_width;
    if (cells > 0) {
        if (!hist) return AFP_ERR_ARG;
        HIPCHK(hipMemcpyAsync(hist, h->vt_hist.p, cells * 4, hipMemcpyDeviceToHost, tbs(h)));
    }
    HIPCHK(tb_sync(h));
    return AFP_OK;
}

// ---- row f4, remaining modes (audfprint_match.py:149-239): the hits of (id, skew range) queries, for exact counts / time ranges
extern "C" int afp_table_select_hits(afp_handle* h, const int32_t* ids, const int32_t* lo, const int32_t* hi, int32_t nq, int64_t* total)
{
    if (!h || nq < 0 || (nq > 0 && (!ids || !lo || !hi)) || !total) return AFP_ERR_ARG;
    if (!h->tb_hashbits) return AFP_ERR_STATE;
    if (h->tb_maxtimebits < 8) return AFP_ERR_PARAM;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = tbs(h);
    const int nid = vote_id_range(h);
    *total = 0;
    h->vs_total = 0;
    h->vs_offsets.assign((size_t)nq + 1, 0);
    h->vs_perm.assign((size_t)nq, 0);
    if (nq == 0) return AFP_OK;
    for (int q = 0; q < nq; q++) if (ids[q] < 0 || ids[q] >= nid) return AFP_ERR_ARG;
    // queries grouped by id (stable): the kernel finds the queries of a hit's id through rank[id] -> qstart
    std::vector<int32_t> ord((size_t)nq);
    for (int q = 0; q < nq; q++) ord[(size_t)q] = q;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return ids[a] < ids[b]; });
    std::vector<int32_t> want, pack;                       // pack: [qstart (nwant + 1) | qlo (nq) | qhi (nq)]
    std::vector<int32_t> qstart;
    for (int k = 0; k < nq; k++) {
        const int q = ord[(size_t)k];
        if (want.empty() || want.back() != ids[q]) { want.push_back(ids[q]); qstart.push_back(k); }
        h->vs_perm[(size_t)q] = k;
    }
    qstart.push_back(nq);
    const int nwant = (int)want.size();
    pack = qstart;
    for (int k = 0; k < nq; k++) pack.push_back(lo[ord[(size_t)k]]);
    for (int k = 0; k < nq; k++) pack.push_back(hi[ord[(size_t)k]]);
    if (h->gh_total == 0) return AFP_OK;
    ENSURE(h->vt_rank, (int64_t)nid * 4);
    ENSURE(h->vt_want, (int64_t)nwant * 4);
    ENSURE(h->vs_q, (int64_t)pack.size() * 4);
    ENSURE(h->vs_cursor, (int64_t)nq * 4);
    ENSURE(h->vs_off, (int64_t)(nq + 1) * 8);
    HIPCHK(hipMemsetAsync(h->vt_rank.p, 0xFF, (int64_t)nid * 4, st));
    HIPCHK(hipMemsetAsync(h->vs_cursor.p, 0, (int64_t)nq * 4, st));
    HIPCHK(hipMemcpyAsync(h->vt_want.p, want.data(), (size_t)nwant * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(h->vs_q.p, pack.data(), pack.size() * 4, hipMemcpyHostToDevice, st));
    afp_launch_vote_setrank((const int32_t*)h->vt_want.p, nwant, nid, (int32_t*)h->vt_rank.p, st);
    const int32_t* d_qstart = (const int32_t*)h->vs_q.p;
    const int32_t* d_lo = d_qstart + (nwant + 1);
    const int32_t* d_hi = d_lo + nq;
    afp_launch_vote_select((const int32_t*)h->gh_hits.p, h->gh_total, nid, (const int32_t*)h->vt_rank.p, d_qstart, d_lo, d_hi,
                           (int32_t*)h->vs_cursor.p, nullptr, nullptr, 0, st);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> cnt((size_t)nq);
    HIPCHK(hipMemcpyAsync(cnt.data(), h->vs_cursor.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));                       // (also: want / pack are stack-lifetime buffers)
    std::vector<int64_t> off((size_t)nq + 1, 0);
    for (int k = 0; k < nq; k++) off[(size_t)k + 1] = off[(size_t)k] + cnt[(size_t)k];
    const int64_t tot = off[(size_t)nq];
    for (int q = 0; q < nq; q++) { const int k = h->vs_perm[(size_t)q]; h->vs_offsets[(size_t)q] = off[(size_t)k]; }
    // (vs_offsets[q] = start of query q's rows in the id-sorted buffer; the fetch re-packs in the caller's order)
    h->vs_offsets[(size_t)nq] = tot;
    h->vs_total = tot;
    *total = tot;
    if (tot == 0) return AFP_OK;
    ENSURE(h->vs_out, tot * 8);
    HIPCHK(hipMemsetAsync(h->vs_cursor.p, 0, (int64_t)nq * 4, st));
    HIPCHK(hipMemcpyAsync(h->vs_off.p, off.data(), (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, st));
    afp_launch_vote_select((const int32_t*)h->gh_hits.p, h->gh_total, nid, (const int32_t*)h->vt_rank.p, d_qstart, d_lo, d_hi,
                           (int32_t*)h->vs_cursor.p, (const int64_t*)h->vs_off.p, (int32_t*)h->vs_out.p, 1, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));                       // (`off` is a stack-lifetime buffer)
    // keep the per-query counts for the fetch
    h->vs_cnt.assign((size_t)nq, 0);
    for (int q = 0; q < nq; q++) h->vs_cnt[(size_t)q] = cnt[(size_t)h->vs_perm[(size_t)q]];
    return AFP_OK;
}
extern "C" int afp_table_fetch_selected(afp_handle* h, int32_t* rows, int64_t* offsets)
{
    if (!h || !offsets) return AFP_ERR_ARG;
    if (!h->tb_hashbits || h->vs_total < 0) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    const int nq = (int)h->vs_perm.size();
    offsets[0] = 0;
    for (int q = 0; q < nq; q++) offsets[q + 1] = offsets[q] + (h->vs_total > 0 ? h->vs_cnt[(size_t)q] : 0);
    if (h->vs_total == 0) return AFP_OK;
    if (!rows) return AFP_ERR_ARG;
    // one copy per query, into the caller's order (queries are few: the candidates of one match_hashes call)
    for (int q = 0; q < nq; q++) {
        const int64_t n = h->vs_cnt[(size_t)q];
        if (n > 0) HIPCHK(hipMemcpyAsync(rows + 2 * offsets[q], (const int32_t*)h->vs_out.p + 2 * h->vs_offsets[(size_t)q], (size_t)n * 8,
                                         hipMemcpyDeviceToHost, tbs(h)));
    }
    HIPCHK(tb_sync(h));
    return AFP_OK;
}

// ---- shader clock under load: one wavefront spins for `ms` of the constant-rate counter (s_memrealtime) and
// reports how many shader cycles (s_memtime) went by -- run it on its own stream beside the pipeline.
__global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
extern "C" int afp_clock_probe_start(afp_handle* h, int ms)
{
    if (!h || ms < 1 || ms > 2000) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (!h->probe_stream) HIPCHK(hipStreamCreateWithFlags(&h->probe_stream, hipStreamNonBlocking));
    ENSURE(h->probe_buf, 64);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
    h->probe_khz = khz;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, h->probe_stream, (unsigned long long)ms * (unsigned long long)khz,
                       (unsigned long long*)h->probe_buf.p);
    HIPCHK(hipGetLastError());
    return AFP_OK;
}
extern "C" int afp_clock_probe_stop(afp_handle* h, double* shader_mhz)
{
    if (!h || !shader_mhz) return AFP_ERR_ARG;
    if (!h->probe_stream) return AFP_ERR_STATE;
    HIPCHK(hipSetDevice(h->device));
    unsigned long long v[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(v, h->probe_buf.p, 16, hipMemcpyDeviceToHost, h->probe_stream));
    HIPCHK(hipStreamSynchronize(h->probe_stream));
    *shader_mhz = v[1] ? (double)v[0] / (double)v[1] * (double)h->probe_khz / 1000.0 : 0.0;
    return AFP_OK;
}

extern "C" int afp_set_timing(afp_handle* h, int enable)
{
    if (!h) return AFP_ERR_ARG;
    h->timing = enable != 0;
    return AFP_OK;
}
extern "C" int afp_reset_timings(afp_handle* h)
{
    if (!h) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(sync_handle(h));
    resolve_timings(h);
    for (int i = 0; i < AFP_NKERNELS; i++) { h->t_ms[i] = 0; h->t_n[i] = 0; }
    return AFP_OK;
}
extern "C" int afp_get_timings(afp_handle* h, double* ms, int64_t* launches)
{
    if (!h) return AFP_ERR_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(sync_handle(h));
    resolve_timings(h);
    for (int i = 0; i < AFP_NKERNELS; i++) { if (ms) ms[i] = h->t_ms[i]; if (launches) launches[i] = h->t_n[i]; }
    return AFP_OK;
}

extern "C" int64_t afp_debug_fetch(afp_handle* h, int what, void* out, int64_t nbytes)
{
    if (!h) return AFP_ERR_ARG;
    if (!h->extracted) return AFP_ERR_STATE;
    { int r_ = finalize(h); if (r_ != AFP_OK) return r_; }
    const int64_t TF = h->total_frames;
    const void* src = nullptr;
    int64_t have = 0;
    std::vector<double> tmp;
    switch (what) {
        case 0: src = h->logS.p; have = TF * AFP_NBINS * 8; break;
        case 1: src = h->nyq.p; have = TF * 8; break;
        case 2:
            if (!(h->flags & AFP_KEEP_DEBUG)) return AFP_ERR_STATE;
            src = h->sgram_dbg.p; have = TF * AFP_NBINS * 8; break;
        case 3: src = h->cand_bin.p; have = TF * h->K * 4; break;
        case 5:
            if (!(h->flags & AFP_KEEP_DEBUG) && !getenv("AFP_SCAN_PROF")) return AFP_ERR_STATE;
            src = h->scan_prof.p; have = (int64_t)h->nunits * 256; break;
        case 6:
            if (!getenv("AFP_HPF_PROF") || !h->scan_prof.p) return AFP_ERR_STATE;
            src = h->scan_prof.p; have = 2048 * 4 * 8; break;
        case 4: {
            std::vector<UnitStats> st(h->nunits);
            std::vector<double> mean(h->nunits);
            std::vector<int32_t> T(h->nunits);
            if (h->nunits) {
                if (hipMemcpy(st.data(), h->stats.p, (size_t)h->nunits * sizeof(UnitStats), hipMemcpyDeviceToHost) != hipSuccess) return AFP_ERR_HIP;
                if (hipMemcpy(mean.data(), h->unit_mean.p, (size_t)h->nunits * 8, hipMemcpyDeviceToHost) != hipSuccess) return AFP_ERR_HIP;
                if (hipMemcpy(T.data(), h->unit_T, (size_t)h->nunits * 4, hipMemcpyDeviceToHost) != hipSuccess) return AFP_ERR_HIP;
            }
            tmp.resize((size_t)h->nunits * 4);
            for (int i = 0; i < h->nunits; i++) {
                tmp[4 * i] = st[i].logfloor; tmp[4 * i + 1] = mean[i]; tmp[4 * i + 2] = st[i].pmax; tmp[4 * i + 3] = T[i];
            }
            have = (int64_t)tmp.size() * 8;
            if (out && nbytes > 0) memcpy(out, tmp.data(), (size_t)(nbytes < have ? nbytes : have));
            return have;
        }
        default: return AFP_ERR_ARG;
    }
    if (out && nbytes > 0 && have > 0) {
        if (hipMemcpy(out, src, (size_t)(nbytes < have ? nbytes : have), hipMemcpyDeviceToHost) != hipSuccess) return AFP_ERR_HIP;
    }
    return have;
}
