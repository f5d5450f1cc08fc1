// afp_table.hip -- the "next" rows of SURVEY.md §8f on the device-resident table: HashTable.store / merge (hash_table.py:91-138,
// 291-323) for whole batches, the packed hand-off, HashTable.get_hits (:150-176) and the matcher's vote counting
// (audfprint_match.py:124-312).  Kernels: k_table.hip.
#include "afp_internal.h"

// ---- hash-table build: HashTable.store for a whole batch, hash_table.py:91-138 ----
// The table lives on a stream of its own, at the highest priority the device offers: its kernels are tiny (a few microseconds
// each) and the host waits for several of them per batch, while the extraction contexts that feed the table keep every CU busy
// with kernels a thousand times longer -- on an ordinary stream each of those waits sat behind whatever was queued (r04, c4 job:
// "store" 6 ms + "replay" 15 ms of host time that was mostly waiting for a slot).
hipError_t tb_sync(afp_handle* h)
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
        if (!clip_off || clip_off[0] < 0) return AFP_ERR_ARG;          // (rows + 2 * clip_off[0] is read below)
        for (int c = 0; c < nclips; c++) if (clip_off[c + 1] < clip_off[c]) return AFP_ERR_ARG;
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

