/*
 * afp.h -- C ABI of libafp_hip.so: MI355X (gfx950) landmark-fingerprint extraction.
 *
 * This is the drop-in boundary for ONE hot path of dpwe/audfprint:
 *
 *     PCM -> 512-pt STFT -> log|S| -> HPF -> decaying-threshold peak pick
 *         -> peak pairs -> 20-bit hashes -> sorted unique (time, hash)
 *
 * The reference has no FFI of its own (it is pure Python); the seam it offers is the
 * class audfprint_analyze.Analyzer.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference repository root).  The
 * Python host code in audfprint_amd/ binds these with ctypes and re-creates the
 * Analyzer class surface on top (see INTEGRATION.md for the binding a maintainer adds).
 *
 * Conventions: plain pointers and sizes, no exceptions across the ABI, every function
 * returns 0 on success or a negative afp_status; afp_strerror() describes it.
 * The caller owns every buffer it passes in; the library never frees caller memory and
 * never hands out pointers the caller must free (device result pointers stay owned by
 * the handle and are valid until the next afp_extract_* call on that handle).
 */
#ifndef AFP_H
#define AFP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): afp_get_path_stats and afp_get_seg_stats write EIGHT int32 (version 1 wrote 4 and 5): a caller built
 * against the version-1 header must not run against this library -- afp_abi_version() tells it, the Python binding checks. */
#define AFP_ABI_VERSION 2

/* The library is built with -fvisibility=hidden: these declarations ARE its dynamic symbol table. */
#if defined(__GNUC__) || defined(__clang__)
#define AFP_API __attribute__((visibility("default")))
#else
#define AFP_API
#endif
#define AFP_MAX_SHIFTS 16   /* Analyzer.shifts (audfprint_analyze.py:130, audfprint.py:295-297) */
#define AFP_MAX_PKS 64      /* Analyzer.maxpksperframe upper bound: one wavefront lane per kept peak */
#define AFP_NFFT 512        /* audfprint_analyze.py:64 N_FFT  (audfprint.py:292 hard-wires it) */
#define AFP_NHOP 256        /* audfprint_analyze.py:65 N_HOP  (audfprint.py:293) */
#define AFP_NBINS 256       /* bins kept after the Nyquist row is dropped (audfprint_analyze.py:295) */

typedef enum afp_status {
    AFP_OK = 0,
    AFP_ERR_ARG = -1,        /* bad argument (null pointer, negative size, unsorted offsets) */
    AFP_ERR_PARAM = -2,      /* parameter outside the supported range (see afp_params) */
    AFP_ERR_HIP = -3,        /* a HIP runtime call failed; afp_last_hip_error() has the text */
    AFP_ERR_NOMEM = -4,      /* workspace would exceed the configured limit / allocation failed */
    AFP_ERR_STATE = -5,      /* call order violated (e.g. fetch before extract) */
    AFP_ERR_NODEVICE = -6    /* no usable gfx950 device */
} afp_status;

/*
 * Parameters = the Analyzer attributes that steer the path, read at call time
 * (audfprint_analyze.py:125-151; set from the CLI at audfprint.py:285-298).
 * Float constants that must equal the reference's numpy values bit-for-bit are computed
 * by the HOST (numpy) and passed in -- the library never recomputes them:
 *   a_dec   = (1 - 0.01*(density*sqrt(n_hop/352.8)/35))**(1/OVERSAMP)  audfprint_analyze.py:277
 *   window  = np.hanning(n_fft+2)[1:-1]                                audfprint_analyze.py:279
 *   gauss   = exp(-0.5*(k/f_sd)^2), k = 0..255 (the symmetric half of __sp_vals, :191-192)
 *   shift_offsets[s] = int(s/shifts*n_hop)                             audfprint_analyze.py:375
 */
typedef struct afp_params {
    double a_dec;
    double hpf_pole;                 /* HPF_POLE**(1/OVERSAMP) = 0.98, audfprint_analyze.py:67,294 */
    int32_t maxpksperframe;          /* 1..AFP_MAX_PKS            audfprint_analyze.py:134 */
    int32_t maxpairsperpeak;         /* >= 1  (fanout)            audfprint_analyze.py:136 */
    int32_t targetdf;                /* default 31                audfprint_analyze.py:139 */
    int32_t mindt;                   /* default 2                 audfprint_analyze.py:141 */
    int32_t targetdt;                /* default 63                audfprint_analyze.py:143 */
    int32_t nshifts;                 /* 1..AFP_MAX_SHIFTS         audfprint_analyze.py:369-377 */
    int32_t shift_offsets[AFP_MAX_SHIFTS];
    const double* window;            /* host pointer, AFP_NFFT doubles  */
    const double* gauss;             /* host pointer, AFP_NBINS doubles */
} afp_params;

typedef struct afp_handle afp_handle;

/* flags for afp_extract_* */
#define AFP_WANT_HASHES 1u   /* run pairing/hash/sort/unique: Analyzer.wavfile2hashes, :385-426 */
#define AFP_WANT_PEAKS  2u   /* emit (col, bin) lists: Analyzer.find_peaks, :255-308 */
#define AFP_KEEP_DEBUG  4u   /* keep intermediates for afp_debug_fetch */
#define AFP_WANT_LANDMARKS 8u /* afp_pairs_from_peaks only: raw (col, f1, f2, dt) rows, Analyzer.peaks2landmarks :310-343 */

/* per-unit (clip x shift) flags reported by afp_fetch_unit_flags */
#define AFP_UNIT_EMPTY 1   /* zero samples: find_peaks returns [] (audfprint_analyze.py:273-274) */
#define AFP_UNIT_ZERO  2   /* identically-zero signal: the reference prints a warning and finds no peaks (:287-290) */
#define AFP_UNIT_CORR  4   /* some |S| fell under max/1e6 and was floored (:285) -- informational */
#define AFP_UNIT_TIE   8   /* SPARSE FRAME: a 512-sample frame all of whose non-zero samples sit at offsets of ONE parity
                            * (all even or all odd) and whose level lies above the unit's floor max|S|/1e6 -- a lone click in
                            * digital silence, two clicks an even distance apart, a click train of even spacing, the +-1 LSB
                            * tail of an undithered fade-out.  For such a frame |S(k)| == |S(256 - k)| in exact arithmetic
                            * (for one sample: all bins are equal), so which of the equal bins are local maxima (:36-52) and
                            * which win a place among the maxpksperframe largest (:217-229) is decided by the rounding noise
                            * of numpy's own FFT IN THE REFERENCE ITSELF (tools/sparse_frame_jitter.py: 2-24 of ~30 peaks
                            * move under a 1e-15 relative perturbation of its rfft output; none for a frame holding both
                            * parities).  The integer output of such a unit may differ from the reference's by the bins
                            * picked in those frames and, through the decaying thresholds they raise (:226-230, :241-251),
                            * in frames within about two decay lengths of them.  afp_fetch_unit_tie_frames tells which
                            * frames.  Unflagged units are bit-exact (constructed on the dense / segment paths, a tested
                            * property on the compact path -- see afp_set_pipeline). */
#define AFP_UNIT_NEARTIE 32 /* a decisive comparison of the threshold passes was closer than the near-tie epsilon
                            * (afp_set_neartie_eps): the integers are this library's dense-path decision, the reference's
                            * own rounding could have decided otherwise */
#define AFP_UNIT_NONFINITE 16 /* a NaN or Inf sample: the reference's max() is NaN and `smax > 0` false, so it prints the
                            * "identically zero" warning and finds no peaks (:283-290); set together with AFP_UNIT_ZERO */

AFP_API int afp_abi_version(void);
/* sha256 (first 16 hex digits) of the kernel / ABI sources this binary was compiled from, embedded by
 * audfprint_amd/build.py; the Python binding compares it with the sources in the tree and refuses a stale
 * library (the .so files are git-ignored build products). */
AFP_API const char* afp_build_id(void);
AFP_API const char* afp_strerror(int status);
AFP_API const char* afp_last_hip_error(void);
AFP_API int afp_device_count(void);

/* Create / destroy a context bound to one GPU (one process per GPU; the context is not
 * thread-safe, like the reference Analyzer -- SURVEY.md §8b "Threading / process model"). */
AFP_API int afp_create(int device, afp_handle** out);
AFP_API void afp_destroy(afp_handle* h);

/* Use an externally-owned hipStream_t (e.g. torch's current stream) instead of the
 * handle's own stream.  Pass NULL to go back to the internal stream. */
AFP_API int afp_set_stream(afp_handle* h, void* hip_stream);

/* Staged mode for bulk ingest (no reference counterpart: the reference handles one file at a time).
 * With two (or three) caller-owned hipStream_t given, every later afp_extract_* enqueues its spectral stage
 * (stft.py:62-94 + the log/mean part of find_peaks, audfprint_analyze.py:281-301) on `spectral_stream`
 * and the scan + pairing stage (find_peaks :263-308 via _decaying_threshold_*, peaks2landmarks :310-343,
 * landmarks2hashes :81-96) on `scan_stream`; both are ordered after the work already queued on the
 * handle's stream and after the handle's previous batch.  The batch is joined on the HOST by the next
 * afp_result_* / afp_fetch_* / afp_table_* call on the handle (the handle's stream itself does not wait:
 * HIP shares hardware queues between streams and a parked queue barrier would stall other handles' stages).
 * Handles that share the SAME two streams pipeline: the FP64-issue-bound spectral stage of batch i+1 runs
 * beside the latency-bound scan stage of batch i instead of two spectral stages contending.
 * `pair_stream` (may be NULL = use scan_stream) optionally splits the pairing / hashing / scatter kernels
 * off the scan stage as a third stage.  Pass (NULL, NULL, NULL) to go back to single-stream operation. */
AFP_API int afp_set_stage_streams(afp_handle* h, void* spectral_stream, void* scan_stream, void* pair_stream);

/* Streams confined to a slice of the chip, for afp_set_stage_streams: the spectral stage and the scan / pairing stages of
 * consecutive batches then run on DISJOINT compute units instead of time-sharing all of them (each stage keeps the
 * occupancy it is tuned for and neither lengthens the other's dependent chains).  CUs [first_cu,
 * first_cu + n_cus); bit ranges are spread evenly over the XCDs by the runtime. */
AFP_API int afp_stream_create_cu_range(int device, int first_cu, int n_cus, void** hip_stream);
AFP_API int afp_stream_destroy(void* hip_stream);

/* Upload parameters + host-computed tables.  Replaces the attribute reads scattered
 * through Analyzer.find_peaks / peaks2landmarks (audfprint_analyze.py:277-279,221,331-337). */
AFP_API int afp_set_params(afp_handle* h, const afp_params* p);

/* Page-locked host memory (hipHostMalloc) for hosts without an allocator of their own for it: PCM that
 * afp_extract_host* reads from such a buffer is uploaded by the copy engine without a staging copy, asynchronously
 * (several handles then overlap uploads with kernels: audfprint.py:173-186 as a pipelined job).  audfprint_amd.batch
 * wraps it as pinned_empty(); torch.Tensor.pin_memory() gives the same kind of memory. */
/* out[4]: HIP_VERSION of the build, hipRuntimeGetVersion() of the runtime the process bound, hipDriverGetVersion(),
 * visible devices.  (A PyTorch-ROCm wheel brings its own libamdhip64 under the system library's SONAME; whichever is
 * mapped first serves the process -- audfprint_amd.runtime_info() reports the pair.) */
AFP_API int afp_runtime_info(int32_t* out);
AFP_API int afp_pinned_alloc(int device, int64_t bytes, void** out);
AFP_API int afp_pinned_free(void* p);

/* Upper bound on device workspace bytes the next extract may allocate (default 200 GiB). */
AFP_API int afp_set_workspace_limit(afp_handle* h, int64_t bytes);
/* HBM parked on the retire list: a workspace buffer that has to grow leaves its old allocation there instead of calling
 * hipFree on the spot (hipFree waits for every stream of the device -- milliseconds in the middle of a pipelined ingest);
 * the list is released at the end of a batch / of a table download once it exceeds AFP_RETIRE_MAX_MB (default 1024), when
 * an allocation fails, and by afp_destroy. */
AFP_API int64_t afp_retired_bytes(void);
/* Workspace bytes a batch of these clip lengths would need (host-only computation). */
AFP_API int64_t afp_workspace_bytes(afp_handle* h, const int64_t* clip_offsets, int32_t nclips, uint32_t flags);

/*
 * The hot path over a batch of clips.  Replaces, per clip, the body of
 * Analyzer.wavfile2peaks' shifts loop + wavfile2hashes (audfprint_analyze.py:369-377,
 * 400-422), i.e. find_peaks (:255-308), peaks2landmarks (:310-343), landmarks2hashes
 * (:81-96) and the uint64 unique/sort (:414-422).
 *
 *   pcm           float32 mono samples of all clips back to back (values exactly as
 *                 audio_read.buf_to_float produces them, audio_read.py:121-145)
 *   clip_offsets  HOST array, nclips+1 non-decreasing sample offsets into pcm
 *                 (a clip may hold at most 2^21 - 64 frames = 13.5 hours at 11025 Hz: AFP_ERR_ARG beyond)
 *
 * afp_extract_device: pcm is a DEVICE pointer (already resident in HBM); the whole pipeline is
 *   queued on the handle's stream and the call returns WITHOUT waiting for the GPU (batches on
 *   different handles overlap).  Any afp_result_* / afp_fetch_* call waits for completion; d_pcm must stay
 *   valid and unchanged until then (the kernels read it asynchronously, and a batch whose compact stage reported
 *   a fault is re-run from it: afp_get_path_stats).
 * afp_extract_host:   pcm is a HOST pointer; copied H2D first (into memory the handle owns).
 */
AFP_API int afp_extract_device(afp_handle* h, const float* d_pcm, const int64_t* clip_offsets,
                       int32_t nclips, uint32_t flags);
AFP_API int afp_extract_host(afp_handle* h, const float* pcm, const int64_t* clip_offsets,
                     int32_t nclips, uint32_t flags);
/* Same, for raw signed 16-bit samples (what ffmpeg pipes: '-f s16le', audio_read.py:196-203).  The
 * integer is converted on the GPU exactly as audio_read.buf_to_float does on the host
 * (x / 32768 in float32, audio_read.py:121-145), so results are identical while the bytes that
 * cross PCIe / are read from HBM halve. */
AFP_API int afp_extract_device_s16(afp_handle* h, const int16_t* d_pcm, const int64_t* clip_offsets,
                           int32_t nclips, uint32_t flags);
AFP_API int afp_extract_host_s16(afp_handle* h, const int16_t* pcm, const int64_t* clip_offsets,
                         int32_t nclips, uint32_t flags);
/* Same, for float64 samples: Analyzer.find_peaks(d, sr) works in the dtype of `d` (np.pad and the float64
 * window multiply of stft.py:87-93 keep a float64 waveform in float64), so API callers who hold float64 audio
 * get the reference's result only if it is not rounded to float32 on the way in.  Range: the kernels form |S|^2 before
 * the log (the reference's np.abs is a hypot), so |x| must stay within about 1e-150 .. 1e+150; the Python Analyzer applies
 * an exact power-of-two gain to waveforms outside 2^-300 .. 2^300 (the path is invariant to it). */
AFP_API int afp_extract_device_f64(afp_handle* h, const double* d_pcm, const int64_t* clip_offsets,
                           int32_t nclips, uint32_t flags);
AFP_API int afp_extract_host_f64(afp_handle* h, const double* pcm, const int64_t* clip_offsets,
                         int32_t nclips, uint32_t flags);

/*
 * Pairing / hashing of GIVEN peak lists (peaks that did not come from this handle's scan, e.g. a
 * .afpk file -- wavfile2peaks' short-circuit, audfprint_analyze.py:351-354).  Replaces
 * Analyzer.peaks2landmarks (:310-343), landmarks2hashes (:81-96) and the unique/sort (:414-422).
 *   peaks             HOST int32 rows (col, bin): 0 <= col < 2^24 non-decreasing inside a unit (the Python binding
 *                     stable-sorts by column first; a decreasing column is AFP_ERR_ARG here), 0 <= bin < 256.  Bins
 *                     ascending and unique inside every column -- what find_peaks / peaks_load produce -- take the mask
 *                     kernels.  Any other order inside a column (descending bins, a bin listed twice) is paired in LIST
 *                     order like the reference's nested loops over peaks_at[col] (:321-341), by a row-walking kernel
 *                     (k_pair_rows, one thread per column; up to 2^16 rows per column, else AFP_ERR_ARG)
 *   unit_peak_offsets HOST int64[nclips*nshifts + 1] row offsets; unit = clip*nshifts + shift
 *   flags             AFP_WANT_HASHES (merged sorted-unique per clip -> afp_fetch_hashes) and/or
 *                     AFP_WANT_LANDMARKS (per unit, reference order -> afp_fetch_landmarks)
 */
AFP_API int afp_pairs_from_peaks(afp_handle* h, const int32_t* peaks, const int64_t* unit_peak_offsets,
                         int32_t nclips, uint32_t flags);
/*   landmarks  int32[4*total] rows (col, f1, f2, dt); unit_offsets int64[nunits+1]; total may be NULL.
 *   Call once with landmarks == NULL to learn *total, then again with a buffer. */
AFP_API int afp_fetch_landmarks(afp_handle* h, int32_t* landmarks, int64_t* unit_offsets, int64_t* total);

/* The two passes of the peak picker over a spectrogram the CALLER supplies: Analyzer._decaying_threshold_fwd_prune
 * (audfprint_analyze.py:199-231) and Analyzer._decaying_threshold_bwd_prune_peaks (:233-253) -- semi-private, but
 * public-named methods that take `sgram` as an argument.  Uses maxpksperframe and the Gaussian table of the last
 * afp_set_params (the f_sd the reference reads from self.f_sd).
 *   sgram     HOST float64 [T][256], frame-major (the transpose of the reference's (256, T) array)
 *   peaks_in  NULL: run the forward pass (and, if bwd_out, the backward pass on its result);
 *             else HOST uint8 [T][256] mask of peaks to backward-prune (at most AFP_MAX_PKS per frame)
 *   fwd_out   uint8 [T][256] forward-pass mask, or NULL;   bwd_out  uint8 [T][256] after the backward pass, or NULL */
AFP_API int afp_prune_spectrogram(afp_handle* h, const double* sgram, int32_t T, double a_dec, const uint8_t* peaks_in,
                          uint8_t* fwd_out, uint8_t* bwd_out);

/* landmarks2hashes (audfprint_analyze.py:81-96) over arbitrary (L,4) int32 rows
 * (time, bin1, bin2, dtime) -> (L,2) int32 rows (time, hash); host buffers in and out. */
AFP_API int afp_hashes_from_landmarks(afp_handle* h, const int32_t* landmarks, int64_t nrows, int32_t* out);

/* Result sizes of the last extract (synchronises the stream). */
AFP_API int afp_result_counts(afp_handle* h, int64_t* total_hashes, int64_t* total_peaks, int64_t* nunits);

/* Copy results to caller-owned host buffers.
 *   hashes            int32[2*total_hashes], rows (time, hash) sorted unique per clip -- the
 *                     (N,2) int32 array wavfile2hashes returns (audfprint_analyze.py:418-422)
 *   clip_hash_offsets int64[nclips+1] CSR offsets (rows) into hashes
 *   peaks             int32[2*total_peaks], rows (col, bin) in find_peaks' order (:303-308)
 *   unit_peak_offsets int64[nunits+1], unit = clip*nshifts + shift
 *   unit_flags        int32[nunits]  AFP_UNIT_* bits                                   */
AFP_API int afp_fetch_hashes(afp_handle* h, int32_t* hashes, int64_t* clip_hash_offsets);
AFP_API int afp_fetch_peaks(afp_handle* h, int32_t* peaks, int64_t* unit_peak_offsets);
AFP_API int afp_fetch_unit_flags(afp_handle* h, int32_t* unit_flags);

/* Device-resident results for GPU consumers (valid until the next extract on h). */
AFP_API int afp_result_device_ptrs(afp_handle* h, const int32_t** d_hashes, const int64_t** d_clip_hash_offsets,
                           const int32_t** d_peaks, const int64_t** d_unit_peak_offsets);

/*
 * ---- "next" row f1: batch build of the reference hash table -------------------
 * Replaces the per-hash Python loop of HashTable.store (hash_table.py:91-138) for all clips of a
 * batch: table[hash & mask, counts++] = ((id+1) << maxtimebits) + (time & timemask) while the
 * bucket has room; slot order = the reference's insertion order (clip order, then row order), so
 * the device table equals the one store() builds clip by clip.  Insertions into FULL buckets use
 * Python's `random` in the reference (:125-131); they are returned as events for the host to
 * replay with the same RNG calls (audfprint_amd/table.py).
 *   afp_table_create   device table uint32[2^hashbits][depth] + counts int32[2^hashbits], zeroed
 *                      (HashTable.__init__, hash_table.py:61-83)
 *   afp_table_upload / afp_table_download   whole-table copies (host arrays as in HashTable)
 *   afp_table_store    rows == NULL: insert the (time, hash) rows of the LAST afp_extract_* on this
 *                      handle (still resident in HBM); else host rows int32[N][2] + clip_offsets
 *                      int64[nclips+1].  clip_ids int32[nclips] = HashTable.name_to_id of each clip.
 *   afp_table_fetch_overflow   int32[n_overflow][4] = (row index, bucket, value, count at insertion)
 */
AFP_API int afp_table_create(afp_handle* h, int32_t hashbits, int32_t depth, int32_t maxtimebits);
AFP_API int afp_table_upload(afp_handle* h, const uint32_t* table, const int32_t* counts);
AFP_API int afp_table_download(afp_handle* h, uint32_t* table, int32_t* counts);
AFP_API int afp_table_store(afp_handle* h, const int32_t* rows, const int64_t* clip_offsets, const int32_t* clip_ids,
                    int32_t nclips, int64_t* n_overflow);
AFP_API int afp_table_fetch_overflow(afp_handle* h, int32_t* events);
/* afp_table_store from rows that already sit in HBM and belong to somebody else -- typically ANOTHER handle's results
 * (afp_result_device_ptrs, after afp_result_counts has waited for them): several extraction contexts, whose uploads and
 * kernels overlap, feed ONE table in the caller's clip order.  d_rows int32[nrows][2], d_clip_off int64[nclips + 1]
 * (device), clip_ids int32[nclips] (host). */
AFP_API int afp_table_store_device(afp_handle* h, const int32_t* d_rows, const int64_t* d_clip_off, int64_t nrows,
                           const int32_t* clip_ids, int32_t nclips, int64_t* n_overflow);
/* The replacement draws of HashTable.store (hash_table.py:125-131) for the overflow events of the last afp_table_store*,
 * done in one call: events fetched and put in insertion order, slot = random.randint(0, count) drawn for each from the
 * Mersenne-Twister state handed in (CPython's own algorithm: Lib/random.py _randbelow_with_getrandbits over
 * Modules/_randommodule.c genrand_uint32), draws with slot < depth written to the device table, the last write to a cell
 * winning as in the reference's loop.  mt_state: the 624 words of random.getstate()[1], mt_pos: its 625th entry; both are
 * advanced exactly as Python's generator would be -- the caller hands them back with random.setstate().  n_written: cells
 * patched.  afp_mt_randint_replay is the bare generator (host only): out[i] = random.randint(0, counts[i]). */
AFP_API int afp_table_replay_overflow(afp_handle* h, uint32_t* mt_state, int32_t* mt_pos, int64_t* n_written);
AFP_API int afp_mt_randint_replay(uint32_t* mt_state, int32_t* mt_pos, const int32_t* counts, int64_t n, int32_t* out);
/* table[bucket][slot] = value for host-decided writes: patches int32[n][3] rows (bucket, slot, value bits), each
 * (bucket, slot) at most once.  Used for the replayed random replacements of HashTable.store
 * (hash_table.py:125-131) and the permuted rows of HashTable.merge (:312-313), so that the DEVICE table
 * stays the authoritative copy. */
AFP_API int afp_table_patch(afp_handle* h, const int32_t* patches, int64_t n);
/* HashTable.merge (hash_table.py:291-323): merge another table with the same hashbits / maxtimebits into the
 * device table.  Per bucket the other table uses: allvals = r_[ours[:count], other[:ocount] + idoffset] with
 * idoffset = ncurrent << maxtimebits (:300; ncurrent = len(self.names) before the merge); if it fits it is
 * stored and count = len(allvals) (:315-321), else count += ocount (:314) and the bucket is reported:
 * the reference draws np.random.permutation(allvals)[:depth] there (:312), an RNG call that stays with the
 * host.  afp_table_fetch_merge_overflow returns those buckets in ascending order (the reference's loop order)
 * with their allvals (row stride depth + other_depth, nvals[i] valid entries); the host permutes and writes
 * the rows back with afp_table_patch.  names / hashesperid bookkeeping (:297-298) is the caller's.
 *   afp_table_merge         other_table uint32[2^hashbits][other_depth], other_counts int32[2^hashbits]: HOST arrays
 *   afp_table_merge_device  the same as DEVICE pointers (e.g. a table received from another GPU over xGMI);
 *                           they must stay valid until afp_table_fetch_merge_overflow has been called        */
AFP_API int afp_table_merge(afp_handle* h, const uint32_t* other_table, const int32_t* other_counts, int32_t other_depth,
                    int32_t ncurrent, int64_t* n_overflow);
AFP_API int afp_table_merge_device(afp_handle* h, const uint32_t* d_other_table, const int32_t* d_other_counts,
                           int32_t other_depth, int32_t ncurrent, int64_t* n_overflow);
AFP_API int afp_table_fetch_merge_overflow(afp_handle* h, int32_t* buckets /* [n] */, int32_t* nvals /* [n] */,
                                   uint32_t* allvals /* [n][depth + other_depth] */);
/* counts[k] = min(counts[k], depth) over the device table: what HashTable.merge into an EMPTY table leaves of every
 * bucket's count (hash_table.py:304-305, 315-321: len(allvals) <= depth).  The parent of `new --ncores N` takes every
 * worker's table that way, core 0's included (audfprint.py:226-235); a rank that merges the others into its OWN table
 * calls this first and is then exactly that parent. */
AFP_API int afp_table_clip_counts(afp_handle* h);
/* Device addresses of the table / counts arrays (valid until afp_table_create / afp_destroy): lets a caller
 * ship a per-GPU table to the merging rank without a host round trip. */
AFP_API int afp_table_device_ptrs(afp_handle* h, uint32_t** d_table, int32_t** d_counts);
/*
 * ---- the PACKED form of the table: only what store / merge can have written ---------------------
 * Neither HashTable.store (hash_table.py:115-131) nor HashTable.merge (:304-321) writes table[k][j] for
 * j >= min(counts[k], depth), and no reader looks there (get_hits :164, merge :304-305).  The packed form is
 * counts int32[2^hashbits] + values uint32[sum_k min(counts[k], depth)]: the filled prefix of every row, bucket
 * after bucket.  A 12 500-clip table (BASELINE configs[3], one GPU's slice) is 7.7 % full: 4 + 32 MB instead of
 * 424 -- this is what leaves the device and what crosses xGMI to the merging rank (audfprint.py:226-235).
 *   afp_table_download_filled   afp_table_download for a host array that was IN STEP with the device table when
 *                               it was created (all zero, HashTable.__init__ hash_table.py:61-83) or uploaded:
 *                               writes counts[] and table[k][0 .. min(counts[k], depth)) only; every other slot
 *                               of the host array keeps what it held, which is what the device holds there.
 *                               n_entries (may be NULL): values moved.
 *   afp_table_pack              build the packed values in HBM; total = number of values.  Valid until the table
 *                               changes (store / merge / patch / upload / create) or is packed again.
 *   afp_table_packed_device_ptrs  device addresses of the packed values and of the counts (for a GPU-to-GPU send)
 *   afp_table_fetch_packed      host copies: values uint32[total], counts int32[2^hashbits]
 *   afp_table_merge_packed      afp_table_merge from the other table's packed form (HOST arrays; n_values must
 *                               equal sum_k min(other_counts[k], other_depth), else AFP_ERR_ARG and nothing is merged)
 *   afp_table_merge_packed_device  the same from DEVICE pointers; they must stay valid until
 *                               afp_table_fetch_merge_overflow has been called                                */
AFP_API int afp_table_download_filled(afp_handle* h, uint32_t* table, int32_t* counts, int64_t* n_entries);
AFP_API int afp_table_pack(afp_handle* h, int64_t* total);
AFP_API int afp_table_packed_device_ptrs(afp_handle* h, uint32_t** d_values, int32_t** d_counts, int64_t* total);
AFP_API int afp_table_fetch_packed(afp_handle* h, uint32_t* values, int32_t* counts);
AFP_API int afp_table_merge_packed(afp_handle* h, const uint32_t* other_values, int64_t n_values, const int32_t* other_counts,
                           int32_t other_depth, int32_t ncurrent, int64_t* n_overflow);
AFP_API int afp_table_merge_packed_device(afp_handle* h, const uint32_t* d_other_values, const int32_t* d_other_counts,
                                  int32_t other_depth, int32_t ncurrent, int64_t* n_overflow);
/* Host threads the large device -> host copies use (a persistent pool: AFP_DL_THREADS, else min(8, CPUs the process may
 * run on); they sleep between copies and make no runtime calls). */
AFP_API int afp_host_threads(void);
/* Best-effort background population (MADV_POPULATE_WRITE, contents unchanged) of a large host array that a later
 * afp_table_download* will write -- a fresh HashTable's 420 MB of untouched zero pages cost more to fault in than the
 * packed table costs to move.  Returns at once; the number of helper threads started (0: not supported / switched off by
 * AFP_NO_PREFAULT).  The caller may free the array at any time. */
AFP_API int afp_host_prefault(void* p, int64_t bytes);
/* HashTable.get_hits (hash_table.py:150-176) over the device-resident table: for every query row
 * (time, hash) the first min(depth, counts) entries of its bucket as int32 rows
 * [id, stored_time - time, hash & mask, time], in the reference's order (row order, then slot order). */
AFP_API int afp_table_get_hits(afp_handle* h, const int32_t* rows, int64_t nrows, int64_t* nhits);
AFP_API int afp_table_fetch_hits(afp_handle* h, int32_t* hits /* [nhits][4] */);
/* The counting half of Matcher._best_count_ids (audfprint_match.py:124-147) over the hit rows of the last
 * afp_table_get_hits, still resident in HBM: ids = np.unique(hits[:,0]) (ascending) and
 * counts = np.bincount(hits[:,0])[ids].  The weighting / argsort / depth cut of :133-147 stay with the caller
 * (numpy's own argsort decides ties). */
AFP_API int afp_table_count_ids(afp_handle* h, int64_t* n_ids);
AFP_API int afp_table_fetch_id_counts(afp_handle* h, int32_t* ids /* [n_ids] */, int32_t* counts /* [n_ids] */);
/* The per-id time-skew histograms of Matcher._approx_match_counts (:279-289), after afp_table_count_ids:
 * mintime = np.amin(hits[:,1]) over ALL hits (:281), width = max skew - mintime + 1, and
 * hist[i][d] = #{hits with id == ids[i] and skew - mintime == d}.  np.bincount(alltimes[allids == ids[i]])
 * is row i cut after its last non-zero entry.  Mode picking (:291-311) stays with the caller. */
AFP_API int afp_table_skew_hist(afp_handle* h, const int32_t* ids, int32_t nids, int32_t* mintime, int32_t* width);
AFP_API int afp_table_fetch_skew_hist(afp_handle* h, int32_t* hist /* [nids][width] */);
/* The row selection of Matcher._exact_match_counts / _unique_match_hashes / _calculate_time_ranges (audfprint_match.py:149-239)
 * for MANY candidate alignments at once, over the hits of the last afp_table_get_hits (still in HBM): query q keeps the hits with
 * id == ids[q] and lo[q] <= skew <= hi[q] (the reference's `allids == id` and `abs(alltimes - mode) <= window`, i.e.
 * lo = mode - window, hi = mode + window).  afp_table_fetch_selected returns their (orig_time, hash) rows, query after query in
 * the caller's order (offsets[nq + 1]); inside a query the rows come in no particular order -- the reference takes np.unique of
 * time + (hash << timebits) (:166-167, the exact count) and the quantiles of the SORTED times (:186-187) of exactly these rows. */
/* np.amax(hits[:, 3]) over the hits (after afp_table_count_ids): the reference sizes its packed keys with it (:157). */
AFP_API int afp_table_hits_max_time(afp_handle* h, int32_t* max_time);
AFP_API int afp_table_select_hits(afp_handle* h, const int32_t* ids, const int32_t* lo, const int32_t* hi, int32_t nq, int64_t* total);
AFP_API int afp_table_fetch_selected(afp_handle* h, int32_t* rows /* [total][2] */, int64_t* offsets /* [nq + 1] */);

/* Per-kernel timing with HIP events on the launch stream (off by default; when on, every
 * kernel launch is bracketed by an event pair).  afp_get_timings sums elapsed ms and launch
 * counts per kernel slot since the last afp_reset_timings; names via afp_kernel_name. */
#define AFP_NKERNELS 12
AFP_API int afp_set_timing(afp_handle* h, int enable);
AFP_API int afp_reset_timings(afp_handle* h);
AFP_API int afp_get_timings(afp_handle* h, double* ms /*[AFP_NKERNELS]*/, int64_t* launches /*[AFP_NKERNELS]*/);
AFP_API const char* afp_kernel_name(int slot);

/* AFP_UNIT_TIE units: the first / last frame holding exactly one non-zero sample above the floor (0 / -1 for the other
 * units).  Which of such a frame's equal bins count as local maxima is decided by the FFT's rounding noise in the reference
 * (audfprint_analyze.py:36-52 over np.fft.rfft); peaks can differ in these frames and, through the decaying threshold,
 * in the frames after them. */
AFP_API int afp_fetch_unit_tie_frames(afp_handle* h, int32_t* first, int32_t* last);

/* afp_fetch_hashes + afp_fetch_peaks + afp_fetch_unit_flags with one wait instead of three (the per-file calls of the
 * Analyzer class are dominated by such round trips).  Pointers may be null; rows that were not requested at extract time
 * are left alone. */
AFP_API int afp_fetch_all(afp_handle* h, int32_t* hashes, int64_t* clip_off, int32_t* peaks, int64_t* unit_off, int32_t* unit_flags);

/* Which kernels a batch goes through.  Defaults: the COMPACT spectral stage (the float64 log-spectrogram never
 * reaches HBM; k_stft.hip) for batches of at least compact_min_units units, the SEGMENT-parallel scan for batches of at
 * most seg_max_units units, the dense kernels otherwise.
 *   compact, seg        -1 the library's rule (by batch size), 0 never, 1 always, -2 the value the handle was created with
 *                       (the rule, or what AFP_COMPACT / AFP_SEG in the environment chose)
 *   the other arguments a positive value, or <= 0: the creation-time value (defaults 768 / 128 / derived from a_dec, or
 *                       AFP_COMPACT_MIN_UNITS / AFP_SEG_MAX_UNITS / AFP_SEG_LEN / AFP_SEG_WARM of the environment)
 * (-2, 0, -2, 0, 0, 0) undoes every earlier call.
 * Exactness per path.  The dense and segment paths round every operation of the onset filter and the threshold
 * recurrences like the reference's separate numpy / scipy calls (-ffp-contract=off) and the segment path is checked bit
 * for bit at every boundary: same floats in, same integers out.  The COMPACT path subtracts the per-unit mean AFTER the
 * onset filter, y = HPF(L)[n] - mean * pole^n, which equals the reference's HPF(L - mean)[n] only in exact arithmetic: the
 * filtered values differ from the dense path's by a few ulps.  Structural ties are preserved by construction (local
 * maxima are decided before the subtraction, which shifts a whole frame; the `c_t` term is formed identically wherever two
 * values that must tie are compared), but a comparison between two INDEPENDENT values closer than those ulps could come
 * out differently.  None has on anything run so far (every golden fixture, 24 random parameter sets, a 2048-clip near-tie
 * sweep, ragged 1100-clip batches compact against dense row for row, every clip of every bench batch): identical integer
 * output is a tested property of the compact path, not a proven one. */
AFP_API int afp_set_pipeline(afp_handle* h, int32_t compact, int32_t compact_min_units, int32_t seg, int32_t seg_max_units,
                     int32_t seg_len, int32_t seg_warm);

/* Test hook for the compact path's recovery: while on, chunk 0 of unit 0 of every compact launch withholds the filter
 * state its successor waits for and the wait is bounded to about a millisecond; the successor reports a hand-off fault and
 * the next afp_result_* / afp_fetch_* call re-runs the whole batch on the dense path (same PCM, same offsets) instead of
 * failing.  The PCM handed to afp_extract_device* must therefore stay valid until the results have been fetched. */
AFP_API int afp_set_compact_force_timeout(afp_handle* h, int32_t on);

/* Path taken by the batch last finalized: out[0] 1 = compact spectral stage, [1] 1 = segment-parallel scan, [2] 1 = the
 * compact stage reported a hand-off fault and the batch was re-run on the dense path (with in-order workgroup dispatch the
 * protocol cannot time out -- the forward-progress argument is in k_stft.hip -- so this counts a violated assumption, a fault
 * or the test hook; a wait is bounded to ~0.3 s and nothing can hang), [3] such re-runs since afp_create. */
AFP_API int afp_get_path_stats(afp_handle* h, int32_t out[8]);
/* ... [4] units the near-tie guard marked in that batch (AFP_UNIT_NEARTIE), [5] 1 = it fired on the compact path and the
 * batch was re-run on the dense path, [6] such re-runs since afp_create, [7] batches (since afp_create) whose onset filter
 * ran CHUNKED: units of 4096 frames (95 s) and more on the segment path filter their chunks in parallel from approximate
 * entry states behind a 1024-frame warm-up, and the chunk boundaries are compared bit for bit -- a mismatch sends the unit to
 * the sequential kernel (k_hpf chunk mode, audfprint_amd/csrc/afp_common.h; AFP_HPF_PAR_MIN=<frames>, 0 = never).
 * afp_set_seg_force_fail(h, 2) makes that comparison fail (test hook). */

/* Near-tie guard of the two threshold passes.  The library's log-spectrogram differs from numpy's by ~1e-13 (absolute: the
 * values are logarithms) and the compact path shifts the filtered values by a few ulps more, so a comparison the reference
 * decides by less than that is one this library may decide the other way.  With eps > 0 the scanner marks every unit in
 * which a DECISIVE comparison -- forward `s_col > sthresh` (audfprint_analyze.py:217), backward `val >= sthresh[bin]`
 * (:242), the cut behind the maxpksperframe largest candidates (:220-221) -- came out with |a - b| <= eps (the first frame
 * of each pass excepted: there a value meets a threshold built from that very value, :204-206 / :237, an exact tie the
 * reference takes the same way): the unit carries
 * AFP_UNIT_NEARTIE, and a batch of the compact path in which that happened is re-run on the dense path (the reference's
 * operation order) before its results are handed out.  An UNMARKED, unflagged unit is then exact by a margin, not by
 * statistics: every decision it took stands under any perturbation of the compared values below eps.  The guard adds
 * comparisons and changes no decision, so a guarded pass that marks nothing also vouches for an unguarded pass of the same
 * batch on the same kernel path.  OFF by default (eps = 0): measured on C3, A/B on one box, it costs 3 % of a step (8 FP64
 * instructions per frame on the scanner wavefront) and 10 % of a one-file call; bench.py runs its parity passes with
 * eps = 1e-11 (a hundred times the log difference, far below anything audio decides) and reports `near_tie_units`.
 * AFP_NEARTIE_EPS in the environment sets the default of new handles. */
AFP_API int afp_set_neartie_eps(afp_handle* h, double eps);

/* Test hook: the final boundary check of the segment-parallel scan marks every unit, so that the sequential kernel
 * re-does them all (exercises the fallback, which real input is not known to reach). */
AFP_API int afp_set_seg_force_fail(afp_handle* h, int32_t on);

/* Segment-parallel scan of the last batch (few long units, e.g. one file through the Analyzer class: the two sequential
 * threshold passes of audfprint_analyze.py:199-253 are cut into segments that warm up on the frames before them, checked
 * bit for bit at every boundary).  out[0] 1 if used, [1] segments, [2] forward / [3] backward segments re-run from their
 * neighbour's end state (runs of segments whose warm-up did not converge: a quiet stretch after a loud one),
 * [4] units whose final check failed: the sequential kernel produced their result, [5] own frames per segment and
 * [6] warm-up frames of the cut, [7] how often (since afp_create) a SHORT cut -- files of up to 1000 frames take segments of
 * 32 + 96 frames instead of 64 + 128: 15 % less per call where it converges -- re-ran more than 5 % of its segments and
 * sent the handle's next 32 batches back to the standard cut (AFP_SEG_ADAPT=0: always the standard cut). */
AFP_API int afp_get_seg_stats(afp_handle* h, int32_t out[8]);

/* Shader clock actually held while other work runs: afp_clock_probe_start queues a one-wavefront kernel on a
 * private stream that spins for `ms` milliseconds of the constant-rate counter; afp_clock_probe_stop waits for it
 * and returns shader cycles / elapsed time in MHz.  (Measurement aid for the roofline figures; DVFS makes the
 * nominal 2400 MHz an upper bound only.) */
AFP_API int afp_clock_probe_start(afp_handle* h, int ms);
AFP_API int afp_clock_probe_stop(afp_handle* h, double* shader_mhz);

/* Debug taps (need AFP_KEEP_DEBUG on the extract).  what:
 *   0 = log|S| before floor/mean, float64 [total_frames][256]   (abs+log, :280,285)
 *   1 = Nyquist-bin log|S|,        float64 [total_frames]
 *   2 = HPF'd spectrogram,         float64 [total_frames][256]  (frame-major; :293-295)
 *   3 = forward-pass candidates:   int32   [total_frames][maxpksperframe] bins (-1 = none)
 *   4 = per-unit stats:            float64 [nunits][4] = logfloor, mean, max|S|^2, nframes
 *   5 = k_scan phase stamps:       uint64  [nunits][32] shader-clock (start, after first barrier,
 *       forward start, backward init start, backward loop start, end, nframes, 0)
 * Returns the number of BYTES the tap holds; copies min(that, nbytes) into out. */
AFP_API int64_t afp_debug_fetch(afp_handle* h, int what, void* out, int64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* AFP_H */
