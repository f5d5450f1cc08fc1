// afp_common.h -- structures shared by the gfx950 kernels and the host-side C ABI.
// Domain vocabulary: clip = one audio excerpt; unit = (clip, part-frame shift) -- one
// independent run of find_peaks (audfprint_analyze.py:369-377); frame = one STFT column.
#pragma once
#include <stdint.h>

#define AFP_NFFT 512
#define AFP_NHOP 256
#define AFP_NBINS 256
#define AFP_WAVE 64

// K1 geometry: one workgroup = 4 wavefronts, each transforming PAIRS of frames
// (two real frames packed into one complex 512-point FFT).
// half-log table of k_stft: 2^BITS intervals of the frexp mantissa (width 2^-(BITS+1)): 9 -> |h| < 2^-11, degree-4 log1p
#ifndef AFP_LOGTAB_BITS
#define AFP_LOGTAB_BITS 9
#endif
#define AFP_LOGTAB_N (1 << AFP_LOGTAB_BITS)
#define STFT_WAVES 4
#define STFT_FPB 64                      // frames per workgroup = 4 wavefronts x 8 pairs x 2 frames
#define STFT_PAIRS_PER_WAVE (STFT_FPB / 2 / STFT_WAVES)
#define COL_CHUNK 256                    // frames per workgroup in the per-(unit,col) kernels
// compact spectral stage (k_stft_c -> k_scan_c): a frame leaves the spectral stage as its 256-bit local-maximum mask plus
// the onset-filtered values of the maxima only (<= 128 of 256 bins; ~70 on noise) instead of 256 float64 values
#define CV_ROW 128                       // doubles per frame row of the compact value buffer (only the first popcount(mask) are touched)
#define CV_HEAD 10                       // dense rows kept per unit: the columns the initial threshold is built from (audfprint_analyze.py:204)

#define UNIT_EMPTY 1
#define UNIT_ZERO 2
#define UNIT_CORR 4
#define UNIT_TIE 8                       // a frame whose non-zero samples share one parity (lone click, even-spaced clicks): peaks decided by FFT rounding noise
#define UNIT_NEARTIE 32                  // a decisive comparison of the scan (forward `val > sthresh`, backward `val >= sthresh`, the cut behind the
                                         // maxpksperframe largest) was decided by less than ScanArgs::nt_eps: set by the scanner wavefront (guard on)
#define UNIT_NONFINITE 16                // a NaN / Inf sample: the reference's max() is NaN, it takes its "identically zero" branch
                                         // (audfprint_analyze.py:283-290: warning, no peaks); set together with UNIT_ZERO

struct UnitStats {        // per unit, written by k_unit_stats
    double logfloor;      // log(max|S| / 1e6)              audfprint_analyze.py:285
    double lsum;          // ordered sum of finite log|S| over 257 x T entries (before flooring)
    double pmax;          // max |S|^2
    int32_t flags;        // UNIT_*
    int32_t pad;
    int32_t tie_first;    // UNIT_TIE: first / last single-parity frame above the floor (else 0 / -1)
    int32_t tie_last;
};

// one unit as k_stft sees it (the other kernels read the separate unit_* arrays)
struct UnitDesc {
    int64_t pcm_off;              // first sample of the unit (clip offset + shift offset)
    int64_t n;                    // samples in the unit
    int64_t fbase;                // first global frame index
    int64_t bbase;                // first unit-major STFT chunk index (the partials keep that order)
    int32_t T;                    // frames = 1 + n/256 (stft.py:33 after the 2x256 pad), 0 if n == 0
    int32_t pad;
};
struct ChunkDesc { int32_t unit, t0; };        // one STFT_FPB-frame chunk

// few pointers on purpose: k_stft runs at the SGPR limit, and every pointer argument is a live scalar register pair
struct StftArgs {
    const void* pcm;              // all clips back to back: float32, int16 or float64 samples
    int32_t pcm_is_s16;           // sample type: 0 float32, 1 int16, 2 float64
    int32_t K;
    const UnitDesc* units;        // [nunits]
    const ChunkDesc* blk;         // [nblk] chunk descriptors: unit-major (dense mode) or TIME-MAJOR (compact mode: chunk k of
                                  //        every unit before chunk k + 1 of any)
    const double* tables;         // [512] host-computed np.hanning(514)[1:-1] | [512][2] cos, -sin of 2*pi*m/512 | [AFP_LOGTAB_N][2] (0.5/c_i, log(c_i)/2)
    double* logS;                 // [total_frames][256]  log|S| (not floored, not mean-subtracted)      (dense mode)
    double* nyq;                  // [total_frames]       log|S| of bin 256                                  (dense mode)
    double* blk_part;             // [6][part_stride] per-chunk partials: max |S|^2, min log|S|, sum log|S|, flat-frame level (0: none),
                                  //                   first / last frame whose non-zero samples share one parity
    int64_t part_stride;
    // pre-fill for k_scan, which writes only non-empty records (saves three memset launches)
    uint64_t* masks;              // [total_frames][4] <- 0
    int32_t* cand_bin;            // [total_frames][K] <- -1
    // ---- compact mode (k_stft<ST, true>): chunk k + 1 of a unit takes the onset-filter state from chunk k (zcarry / zflag)
    double* cvals;                // [total_frames][CV_ROW] onset-filtered values (mean not yet subtracted) of the local maxima, ascending bin
    uint64_t* lmask;              // [total_frames][4] 256-bit local-maximum mask (audfprint_analyze.py:36-52)
    double* head;                 // [nunits][CV_HEAD][256] dense onset-filtered rows of the first CV_HEAD frames
    double* ylast;                // [nunits][256] dense onset-filtered last row (seeds the backward pass, :237)
    double* zcarry;               // [nunits][256] filter state after the unit's latest finished chunk
    unsigned long long* zflag;    // [nunits] (epoch << 32) | chunks finished
    unsigned long long epoch;     // launch counter of the handle (flags of earlier launches never match)
    double pole;
    int32_t* err;                 // [1] set when a chunk gave up waiting for its predecessor
    int32_t* list_zero;           // compact launch: reset the chunk-list counter k_unit_stats appends to (runs before it in stream order)
    // dense mode behind the compact stage (k_stft<ST, false, true>): transform only the listed chunks (those of the units
    // flagged UNIT_CORR; written by k_unit_stats), a fixed grid striding over the list
    const int32_t* list_cnt;
    const ChunkDesc* list;
    int32_t spin_limit;           // bound of the hand-off wait in sleeps of 64 x 16 cycles + a flag load (2^18: ~0.3 s; the test hook sets 2^10)
    int32_t skip_unit, skip_chunk; // test hook (afp_set_compact_force_timeout): this chunk does not publish its state (-1: none)
};
#define TAB_WINDOW 0
#define TAB_TWIDDLE AFP_NFFT
#define TAB_LOGTAB (AFP_NFFT + 2 * AFP_NFFT)
#define TAB_DOUBLES (AFP_NFFT + 2 * AFP_NFFT + 2 * AFP_LOGTAB_N)

struct StatsArgs {
    const int32_t* unit_T;
    const int64_t* unit_bbase;    // [nunits+1] first STFT chunk of the unit
    const double* blk_pmax;
    const double* blk_lmin;
    const double* blk_lsum;
    const double* blk_flat;
    int64_t part_stride;          // blk_flat + part_stride / + 2 part_stride: first / last single-parity frame of the chunk
    UnitStats* stats;
    int32_t nunits;
    // compact pipeline: the STFT chunks of every unit that needs the floor (UNIT_CORR) are appended here (null: no list)
    int32_t* corr_cnt;
    ChunkDesc* corr_list;
};

struct CorrArgs {
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int64_t* unit_bbase;    // [nunits+1]
    int32_t nunits;
    const int32_t* blk_unit;
    const int32_t* blk_t0;
    const double* blk_lmin;
    const UnitStats* stats;
    const double* logS;
    const double* nyq;
    double* blk_corr;             // [nblk]
};

struct ScanArgs {
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int64_t* unit_bbase;
    const UnitStats* stats;
    const double* blk_corr;
    const double* logS;
    const double* gauss;          // [256] host-computed exp(-0.5*(k/f_sd)^2)
    double a_dec;
    double pole;
    int32_t K;                    // maxpksperframe
    double* cand_val;             // [total_frames][K] forward-pass survivors, descending (val, bin)
    int32_t* cand_bin;            // [total_frames][K] (-1 = none)
    uint64_t* masks;              // [total_frames][4] final 256-bit peak mask per frame
    double* ylast;                // [nunits][256] scratch: the onset-filtered LAST column of each unit (seeds the backward pass)
    double* unit_mean;            // [nunits] debug/report: the mean that was subtracted
    double* sgram_dbg;            // optional [total_frames][256] HPF'd spectrogram (debug) or null
    unsigned long long* prof;     // optional [nunits][32] shader-clock stamps / per-class cycle sums of the scanner wave (debug) or null
    int32_t raw_rows;             // logS rows are the onset-filtered spectrogram itself (afp_prune_spectrogram)
    int32_t fwd_off;              // raw_rows only: skip the forward selection (cand_* hold the caller's peaks)
    // compact mode (k_scan_c): rows come from k_stft<ST, true>
    const double* cvals;          // [total_frames][CV_ROW]
    const uint64_t* lmask;        // [total_frames][4]
    const double* head;           // [nunits][CV_HEAD][256]
    // segment mode (k_scan_seg, few long units): see SegDesc
    const struct SegDesc* segs;   // [nseg]
    double* seg_state;            // [SEG_NSTATE][nseg][256] threshold vectors at the segment boundaries
    int32_t* seg_status;          // [0] units that failed the final check, [1] forward / [2] backward segments re-run, [3] k_hpf gave up: every unit falls back
    int32_t* seg_ufail;           // [nunits] a boundary of this unit failed the final check: the sequential kernel re-does the unit
    int32_t nseg;
    int32_t seg_W;                // warm-up frames
    int32_t seg_phase;            // SEG_FWD / SEG_BWD
    int32_t seg_repair;           // 0: first launch of a phase (every segment from its warm-up); 1: the CHAIN launch -- see k_scan_seg
    int32_t seg_force_fail;       // test hook: k_seg_verify marks every unit (the sequential kernel then re-does them)
    int32_t* seg_rerun;           // [2][nseg] (forward, backward) 1: the chain launch re-ran the segment, its state is in the *1 planes
    int32_t* seg_flag;            // [nseg] k_seg_flags, current phase: the segment's warm-up did not reach its neighbour's end state
    const int32_t* seg_ufirst;    // [nunits + 1] first segment of each unit (a unit's segments are consecutive, ascending frames)
    const double* hpf_dump;       // [ndump][2][256] records of k_hpf (SegDesc::dz_* / dy_* index them)
    // dense fallback behind the segment kernels: unit u runs only if only_if[3] != 0 (k_hpf gave up) or only_if_unit[u] != 0,
    // and writes EVERY record / mask row (the segmented attempt left its own behind)
    const int32_t* only_if;
    const int32_t* only_if_unit;
    int32_t clear_all;
    // near-tie guard (0: off): |a - b| <= nt_eps in a decisive comparison marks the unit (UNIT_NEARTIE) and counts it once
    double nt_eps;
    UnitStats* stats_rw;          // = stats
    int32_t* nt_count;            // units marked by this batch (may be null)
};

// Segment-parallel scan of a long unit (a single file: 12 920 sequential frames, twice, on one workgroup otherwise).
// The threshold of both passes is an element-wise max of decayed bumps (audfprint_analyze.py:226-230, 244-252), so a scan
// that starts W frames early from the standard initialisation on its own first column(s) reaches, within a few dozen
// frames, a state BIT-identical to the sequential scan's -- and stays identical (tools/seg_convergence.py: median 15-57
// frames, maximum 117 over density 20 / 70, noise / tonal).  Every segment therefore scans [s - W, e) (forward) or
// [s, e + 1 + W) downwards (backward), records only its own frames, and leaves the threshold vectors it had at its
// boundaries; where a segment's entry state is not the bit pattern its neighbour ended with (a quiet stretch after a loud
// one remembers the loud part for hundreds of frames), a CHAIN launch -- one workgroup per unit -- re-runs such runs of
// segments sequentially from the last true state (k_scan.hip, k_scan_seg); a final check of every boundary guards the
// result (failure: the dense sequential kernel re-does the unit).  The onset filter does not forget its state bit-exactly, so k_hpf carries it through the whole unit
// first (a 3-operation chain per frame instead of the scan's several hundred cycles) and leaves the state at the frames the
// segments start from; the segments then filter their own rows exactly like the sequential kernel.
struct SegDesc {
    int32_t unit;
    int32_t s, e;                 // own frames [s, e)
    int32_t prev, next;           // neighbouring segments of the same unit (-1: none)
    // records of k_hpf (HpfArgs::dump_state) this segment starts from: the onset-filter state at entry of the first frame
    // of its forward scan (first launch: frame s - W; repair: frame s; -1: the unit's first frame, zero state) and the
    // onset-filtered column its backward scan is seeded with (first launch: the last warm-up frame; repair: frame e)
    int32_t dz_fwd, dz_rep, dy_bwd, dy_rep;
    int32_t pad;
};
#define SEG_FWD 1
#define SEG_BWD 2
#define SEG_NSTATE 8
#define ST_FENTRY 0               // forward: state at entry of frame s (after the warm-up), first launch
#define ST_FEXIT0 1               //          state at entry of frame e, first launch
#define ST_FEXIT1 2               //          the same where the chain launch re-ran the segment
#define ST_BENTRY 3               // backward: state at entry of frame e (after the warm-up; e belongs to the next segment too)
#define ST_BEXIT0 4               //           state at entry of frame s (the segment's last frame), first launch
#define ST_BEXIT1 5               //           the same where the chain launch re-ran the segment
#define ST_FENTRY1 6              // the state a re-run started from (= the neighbour's final end state)
#define ST_BENTRY1 7

#define HPF_MAX_DUMPS 4096         // listed frames per unit (the host sizes the segments accordingly)
// CHUNK MODE of k_hpf (round 6): a long unit's onset filter in parallel, exact by verification.  The recurrence
// y = x + z ; z = -x + pole y  is a contraction (pole = 0.98): two runs over the same frames that start from states a few
// ulps apart meet the SAME bit pattern after a few hundred frames (tools/hpf_merge.py: 0.46 % of bins still apart after 256
// frames, 7.6e-5 after 512, none of 75 000 after 640 / 768; the tail falls ~6x per 128 frames) and stay together from there.
//   pass 1  every granule of HPF_GRAN frames from a ZERO state -> its local end state L_j (the state enters linearly)
//   pass 2  chunk k = frames [k C, k C + HPF_WARM + C), C = HPF_OWN: entry state folded from the granules before it
//           (z~ = L_j + pole^GRAN z~, a few ulps off the true state), then the EXACT recurrence; it records the listed frames
//           of its last C frames only, the state at entry of that own range (zmid) and the state it ends with (zend).  The
//           first chunk starts from the true zero state and owns everything it filters.
//   verify  zend of chunk k == zmid of chunk k + 1, bit for bit, for every bin: by induction from the first chunk every
//           recorded state is then the sequential filter's.  Any mismatch raises HpfArgs::fail and the sequential scan
//           kernel re-does the unit (ScanArgs::only_if) -- expected once in a few thousand long files.
#define HPF_GRAN 256
#define HPF_WARM 1024
#define HPF_OWN 512
#define HPF_FOLD 8                 // granules folded into an entry state: pole^(8 x 256) = 1e-18, nothing further back can be seen
struct HpfChunk {
    int32_t unit;
    int32_t t_begin, t_end;       // frames filtered: [t_begin, t_end); multiples of 32 but for a unit's last frame
    int32_t own;                  // first frame whose records this chunk writes (t_begin for a unit's first chunk)
    int32_t d0, d1;               // those records: dump_frame[d0 .. d1)
    int32_t zin_first, zin_n;     // granule end states folded into the entry state (zin_n == 0: the zero state, t_begin == 0)
    int32_t zmid, zend;           // slots in HpfArgs::zbnd for the state at entry of frame `own` / of frame t_end (-1: none)
    int32_t pad[2];
};
struct HpfArgs {                  // k_hpf: floor + mean + onset filter through the whole unit; leaves the state at the listed frames
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int64_t* unit_bbase;
    const UnitStats* stats;
    const double* blk_corr;
    const double* logS;
    const int32_t* dump_off;      // [nunits+1] range of the unit's records in dump_frame (frames ascending)
    const int32_t* dump_frame;    // [ndump]
    double* dump_state;           // [ndump][2][256]: filter state at ENTRY of the frame, onset-filtered column of the frame
    int32_t* fail;                // seg_status[3]: set when a unit's list does not fit (the sequential kernel then re-does EVERY unit: ScanArgs::only_if)
    double pole;
    unsigned long long* prof;     // AFP_HPF_PROF=1 (measurement aid): cycle stamps of the filter wavefront of workgroup (0, 0): [phase][4] = start, end, -, -
    // chunk mode (chunks != nullptr: blockIdx.x indexes chunks instead of units)
    const HpfChunk* chunks;
    const double* gran;           // [granules][256] local end states of pass 1 (read by pass 2)
    double* zbnd;                 // [slots][256] pass 1: the granule end states; pass 2: zend / zmid pairs for k_hpf_verify
    double polepow;               // pole^HPF_GRAN
};


struct PairArgs {
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int32_t* cblk_unit;     // COL_CHUNK-frame chunk descriptors over units
    const int32_t* cblk_t0;
    const uint64_t* masks;
    uint32_t* hslots;             // [total_frames][slot] hashes of the frame's source peaks
    int32_t* hcnt;                // [total_frames]
    int32_t slot;                 // K * fanout
    int32_t fanout, targetdf, mindt, targetdt;
    int32_t lds_lists;            // per-thread working list lives in LDS (slot <= 48 words)
    int32_t lm_mode;              // 1: emit raw landmarks f1 | f2<<8 | dt<<16 in the reference's nested order
};

// peak lists whose columns hold bins in LIST order (not ascending / not unique): k_pair_rows pairs straight from the rows
struct PairRowsArgs {
    const int32_t* rows;          // (col, bin) rows of all units, columns non-decreasing inside a unit
    const int64_t* upo;           // [nunits + 1] row offsets
    const int32_t* rcnt;          // [total_frames] rows per column
    const int32_t* roffs;         // [total_frames] first row of the column, relative to the unit's first row
};

struct MergeArgs {                // shifts > 1: S-way merge + de-dup of the per-shift lists of one (clip, col)
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int64_t* clip_mfbase;   // [nclips] first merged-frame index
    const int32_t* clip_T0;       // [nclips] merged frames of the clip = max T over its shifts
    const int32_t* mblk_clip;     // COL_CHUNK chunk descriptors over clips
    const int32_t* mblk_t0;
    const uint32_t* hslots;
    const int32_t* hcnt;
    uint32_t* mslots;             // [total_mframes][mslot]
    int32_t* mcnt;
    int32_t slot, mslot, S;
};

struct SegScanArgs {              // exclusive scan of counts inside each segment
    const int32_t* counts;
    const int64_t* seg_base;      // [nseg]
    const int32_t* seg_len;       // [nseg]
    int32_t* offs;                // [total] exclusive offset inside the segment
    int64_t* seg_total;           // [nseg]
};

struct ScatterHashArgs {
    const int32_t* seg_len;       // [nclips] merged frames per clip
    const int64_t* seg_base;      // [nclips]
    const int32_t* blk_seg;       // COL_CHUNK chunk descriptors over clips
    const int32_t* blk_t0;
    const uint32_t* slots;
    const int32_t* cnt;
    const int32_t* offs;
    const int64_t* seg_off;       // [nclips+1] CSR offsets
    int32_t* out;                 // [total][2]
    int64_t cap;                  // rows the output buffer holds (writes beyond are dropped; host re-runs)
    int32_t slot;
};

struct ScatterPeakArgs {
    const int32_t* seg_len;       // [nunits]
    const int64_t* seg_base;
    const int32_t* blk_seg;
    const int32_t* blk_t0;
    const uint64_t* masks;
    const int32_t* offs;
    const int64_t* seg_off;       // [nunits+1]
    int32_t* out;                 // [total][2]
    int64_t cap;
};

struct ScatterLmArgs {            // raw landmarks -> (col, f1, f2, dt) rows, CSR over units
    const int32_t* seg_len;
    const int64_t* seg_base;
    const int32_t* blk_seg;
    const int32_t* blk_t0;
    const uint32_t* slots;
    const int32_t* cnt;
    const int32_t* offs;
    const int64_t* seg_off;       // [nunits+1]
    int32_t* out;                 // [total][4]
    int64_t cap;
    int32_t slot;
};

// k_export: the results of a SMALL batch (one file through the Analyzer class, audfprint.py:164-165) written straight into
// pinned host memory by ONE launch at the end of the chain, so that fetching them costs one wait and a host memcpy instead
// of five pageable device-to-host copies.  Host image: int64 hdr[8] = {ok, hashes, peaks, ...}; from byte 64: clip_hoff
// [nclips + 1] (hashes wanted), unit_poff [nunits + 1] (peaks wanted), unit flags int32 [nunits], padding to 16 bytes,
// hash rows, peak rows.  ok = 0 (nothing but the totals is written) when a scatter dropped rows (finalize() re-runs it)
// or the image does not fit.
#define AFP_EXPORT_HDR_BYTES 64
struct ExportArgs {
    const int32_t* hashes;        // [th][2]   (null: hashes not wanted)
    const int64_t* clip_hoff;     // [nclips + 1]
    int64_t cap_h;                // rows the scatter could write
    const int32_t* peaks;         // [tp][2]   (null: peaks not wanted)
    const int64_t* unit_poff;     // [nunits + 1]
    int64_t cap_p;
    const UnitStats* stats;       // [nunits]
    const int32_t* seg_status;    // 4 x int32 of the segment-parallel scan, or null
    int64_t* totals;              // pinned: [0] hashes, [1] peaks, [4..5] <- seg_status
    char* host;                   // pinned image
    int64_t host_cap;
    int32_t nclips, nunits;
    int32_t* seg_zero;            // the segment scan's [status | per-unit fail flags | re-run marks | ...] block, cleared (zero_words
    int32_t zero_words;           // int32 words) for the NEXT batch once the status has been copied out; null: leave it
    const int32_t* nt_count;      // units the near-tie guard marked (ScanArgs::nt_count) -> totals[6]; null: guard off
};

// Fused pairing + cross-shift merge (k_pairmerge): one wavefront works through the columns of
// one clip; per source peak the 64 lanes examine 64 target frames at once.
struct PairMergeArgs {
    const int32_t* unit_T;
    const int64_t* unit_fbase;
    const int64_t* clip_mfbase;   // [nclips]
    const int32_t* clip_T0;       // [nclips] merged frames = max T over the clip's shifts
    const int32_t* pblk_clip;     // chunk descriptors over clips, `ch` columns each
    const int32_t* pblk_t0;
    const uint64_t* masks;
    uint32_t* oslots;             // [total_mframes][oslot] sorted unique hashes of (clip, col)
    int32_t* ocnt;                // [total_mframes]
    int32_t oslot;                // S * K * fanout
    int32_t S, ch;                // shifts, columns per workgroup (multiple of 4)
    int32_t dedupe;               // hashes of one (clip, col) may repeat: several shifts, or dt / df wrapping mod 64
    int32_t fanout, targetdf, mindt, targetdt;
};

// hash-table build (k_table.hip)
struct TableArgs {
    const int32_t* rows;          // [N][2] (time, hash)
    const int64_t* clip_off;      // [nclips+1] row offsets
    const int32_t* clip_ids;      // [nclips] table id of each clip
    int64_t nrows;
    int32_t nclips;
    int32_t hashbits, depth, maxtimebits;
    uint32_t* table;              // [2^hashbits][depth]
    int32_t* counts;              // [2^hashbits]
    int64_t* newcnt;              // [2^hashbits + 1] rows of this batch per bucket, then (scanned) first position
    int64_t* first;               // [2^hashbits + 1]
    int32_t* fill;                // [2^hashbits]
    unsigned long long* seg;      // [N] (row << 32) | value
    int32_t* overflow;            // [cap][4] row, bucket, value, count-at-insertion
    int32_t* ovcnt;               // [1]
    int32_t* biglist;             // [2^hashbits] buckets with long segments
    int32_t* bigcnt;              // [1]
};

