// afp_abi.hip -- host side of libafp_hip.so: the C ABI declared in include/afp.h.
// Builds the per-batch descriptors (units, frame chunks), owns the grow-only HBM workspace,
// enqueues the kernels of k_stft.hip / k_scan.hip / k_pair.hip on one HIP stream and copies
// results out.  No torch types, no exceptions across the boundary.
#include "afp_internal.h"

static const char* k_names[AFP_NKERNELS] = {"k_stft", "k_unit_stats", "k_floor_corr", "k_scan", "k_pair",
                                            "k_merge", "k_seg_scan(hashes)", "k_excl_scan64",
                                            "k_scatter_hashes", "k_seg_scan(peaks)", "k_scatter_peaks",
                                            "pipeline(first launch..last launch)"};

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
    DevBuf* bufs[] = {&h->hpf_gran, &h->hpf_bnd, &h->d_tables, &h->d_gauss, &h->d_desc, &h->pcm_stage, &h->logS, &h->nyq,
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
            // chunk mode of k_hpf (afp_common.h, HpfChunk): worth its two extra launches from ~95 s of audio on (the sequential
            // filter costs 13 ns per frame, the three launches of the chunked one ~30 us whatever the length)
            std::vector<HpfChunk>& c1 = h->hpf_c1;
            std::vector<HpfChunk>& c2 = h->hpf_c2;
            c1.clear(); c2.clear(); h->hpf_nbnd = 0; h->hpf_ngran = 0;
            static const int hpf_par_min = []() { const char* e = getenv("AFP_HPF_PAR_MIN"); return e ? atoi(e) : 4096; }();
            if (hpf_par_min > 0 && longest >= std::max(hpf_par_min, HPF_WARM + 2 * HPF_OWN)) {
                for (int u = 0; u < g.nunits; u++) {
                    const int T = h->unit_T_host[(size_t)u];
                    if (T <= 0) continue;
                    const int32_t* fr = dfr.data() + doff[(size_t)u];
                    const int nfr = doff[(size_t)u + 1] - doff[(size_t)u];
                    auto rec_at = [&](int f) { return doff[(size_t)u] + (int)(std::lower_bound(fr, fr + nfr, f) - fr); };
                    // chunks k = 0 .. K - 1 filter [k C, k C + WARM + C) and own its last C frames (the first: all of it, the
                    // last: up to T).  A unit shorter than WARM + 2 C is one chunk from the zero state = the sequential filter.
                    const int K = T < HPF_WARM + 2 * HPF_OWN ? 1 : (T - HPF_WARM + HPF_OWN - 1) / HPF_OWN;
                    const int gran0 = h->hpf_ngran;
                    if (K > 1) {
                        // granules [j G, (j + 1) G) up to the last chunk's first frame
                        const int ng = (K - 1) * HPF_OWN / HPF_GRAN;
                        for (int j = 0; j < ng; j++) {
                            HpfChunk c = {};
                            c.unit = u; c.t_begin = j * HPF_GRAN; c.t_end = c.t_begin + HPF_GRAN; c.own = c.t_begin;
                            c.d0 = c.d1 = 0; c.zin_first = 0; c.zin_n = 0; c.zmid = -1; c.zend = gran0 + j;
                            c1.push_back(c);
                        }
                        h->hpf_ngran += ng;
                    }
                    for (int k = 0; k < K; k++) {
                        HpfChunk c = {};
                        c.unit = u; c.t_begin = k * HPF_OWN;
                        c.t_end = k + 1 < K ? c.t_begin + HPF_WARM + HPF_OWN : T;
                        c.own = k == 0 ? 0 : c.t_begin + HPF_WARM;
                        c.d0 = rec_at(c.own); c.d1 = k + 1 < K ? rec_at(c.t_end) : doff[(size_t)u + 1];
                        c.zin_n = std::min(HPF_FOLD, c.t_begin / HPF_GRAN);
                        c.zin_first = gran0 + c.t_begin / HPF_GRAN - c.zin_n;
                        // boundary b between chunk k and k + 1 (frame t_end of k = own of k + 1): slots 2 b and 2 b + 1
                        c.zmid = k > 0 ? 2 * (h->hpf_nbnd - 1) + 1 : -1;
                        c.zend = k + 1 < K ? 2 * h->hpf_nbnd : -1;
                        if (k + 1 < K) h->hpf_nbnd++;
                        c2.push_back(c);
                    }
                }
            }
            }
            if (longest > 2 * (S + W) && !sv.empty()) {        // (a short unit gains nothing: the segments cost launches and warm-up)
                const int nseg = (int)sv.size();
                const size_t ndump = dfr.size();
                auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
                const size_t o_idx = al((size_t)nseg * sizeof(SegDesc));
                const size_t o_uf = o_idx + al((doff.size() + ndump) * 4);
                const size_t o_c1 = o_uf + al(((size_t)g.nunits + 1) * 4);
                const size_t o_c2 = o_c1 + al(h->hpf_c1.size() * sizeof(HpfChunk));
                const size_t pack_bytes = o_c2 + al(h->hpf_c2.size() * sizeof(HpfChunk));
                const size_t z_uf = 256, z_rr = z_uf + al((size_t)g.nunits * 4), zero_bytes = z_rr + al((size_t)nseg * 8);
                ENSURE(h->seg_desc, (int64_t)pack_bytes);
                ENSURE(h->seg_state, (int64_t)SEG_NSTATE * nseg * AFP_NBINS * 8);
                ENSURE(h->seg_status, (int64_t)zero_bytes);
                ENSURE(h->seg_flag, (int64_t)nseg * 4);
                ENSURE(h->hpf_dump, (int64_t)(ndump + 1) * 2 * AFP_NBINS * 8);
                ENSURE(h->ylast, (int64_t)std::max(nseg, g.nunits) * AFP_NBINS * 8);
                if (!h->hpf_c2.empty()) {
                    ENSURE(h->hpf_gran, (int64_t)std::max(1, h->hpf_ngran) * AFP_NBINS * 8);
                    ENSURE(h->hpf_bnd, (int64_t)std::max(1, 2 * h->hpf_nbnd) * AFP_NBINS * 8);
                }
                h->hpf_c1_p = (const HpfChunk*)((char*)h->seg_desc.p + o_c1);
                h->hpf_c2_p = (const HpfChunk*)((char*)h->seg_desc.p + o_c2);
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
                    if (!h->hpf_c1.empty()) memcpy(h->h_seg_stage + o_c1, h->hpf_c1.data(), h->hpf_c1.size() * sizeof(HpfChunk));
                    if (!h->hpf_c2.empty()) memcpy(h->h_seg_stage + o_c2, h->hpf_c2.data(), h->hpf_c2.size() * sizeof(HpfChunk));
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
                s.seg_rerun = h->seg_rerun_p; s.seg_force_fail = h->seg_force_fail == 1 ? 1 : 0;
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
            ha.chunks = nullptr; ha.gran = nullptr; ha.zbnd = nullptr; ha.polepow = 0.0;
            if (!h->hpf_c2.empty()) {
                // long units: granule end states -> chunks with a warm-up -> bit-compare of the chunk boundaries (afp_common.h)
                HpfArgs p1 = ha;
                p1.chunks = h->hpf_c1_p; p1.zbnd = (double*)h->hpf_gran.p; p1.prof = nullptr;
                if (!h->hpf_c1.empty()) afp_launch_hpf(&p1, (int)h->hpf_c1.size(), st);
                ha.chunks = h->hpf_c2_p; ha.gran = (const double*)h->hpf_gran.p; ha.zbnd = (double*)h->hpf_bnd.p;
                ha.polepow = pow(h->prm.hpf_pole, (double)HPF_GRAN);
                afp_launch_hpf(&ha, (int)h->hpf_c2.size(), st);
                afp_launch_hpf_verify((const double*)h->hpf_bnd.p, h->hpf_nbnd, ha.fail, h->seg_force_fail == 2 ? 1 : 0, st);
                h->hpf_par_total++;
            } else
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
    // (round 6: a list-order column may hold any number of rows -- peaks_at[col] of the reference has no limit,
    //  audfprint_analyze.py:321-341 -- up to 2^16: a slot of k_pair_rows then holds maxrun x fanout pairs per column)
    if (list_order && maxrun > (1 << 16)) return AFP_ERR_ARG;
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
    h->pair_K = list_order ? std::max(h->prm.maxpksperframe, maxrun) : std::min(256, std::max(h->prm.maxpksperframe, maxrun));
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
int finalize(afp_handle* h)
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
    out[4] = h->nt_units_last; out[5] = h->batch_nt_redone ? 1 : 0; out[6] = h->nt_redone_total; out[7] = h->hpf_par_total;
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
    h->seg_force_fail = on == 2 ? 2 : on ? 1 : 0;          // (2: the boundary check of the chunked onset filter fails instead)
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
