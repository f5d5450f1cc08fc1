// tools/hbm_mix.hip -- what HBM rate does k_stft's access MIX reach on this chip?  (NOT part of the product library.)
//
// k_stft moves, per 64-frame chunk, 64 KB of float32 PCM in (coalesced 256-B rows) and 128.5 KB of float64 log|S| out
// (coalesced 512-B rows): a 1:2 read:write stream.  The guide's "achievable 6.3 TB/s" is a read figure; this probe
// times pure reads, pure writes and the 1:2 mix with the same row shapes, grid and workgroup size, no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_mix tools/hbm_mix.hip && /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one workgroup (256 threads) per chunk: reads RD_ROWS rows of 64 floats per wave (256 B per row), writes WR_ROWS rows
// of 64 doubles per wave (512 B per row)
template <int RD_ROWS, int WR_ROWS, bool NT>
__global__ __launch_bounds__(256) void k_mix(const float* __restrict__ src, double* __restrict__ dst, double* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t blk = blockIdx.x;
    float acc = 0.f;
    if (RD_ROWS > 0) {
        const float* s = src + (blk * 4 + wave) * (size_t)(RD_ROWS * 64);
#pragma unroll 8
        for (int r = 0; r < RD_ROWS; r++) acc += s[r * 64 + lane];
    }
    if (WR_ROWS > 0) {
        double* d = dst + (blk * 4 + wave) * (size_t)(WR_ROWS * 64);
        // values with random mantissa bits (all-zero buffers could flatter the memory system)
        unsigned long long h = (blk * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + (unsigned long long)__float_as_uint(acc);
#pragma unroll 8
        for (int r = 0; r < WR_ROWS; r++) {
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            const double v = __longlong_as_double((long long)((h >> 12) | 0x3FF0000000000000ull));
            if (NT) __builtin_nontemporal_store(v, &d[r * 64 + lane]);
            else d[r * 64 + lane] = v;
        }
    } else if (acc == 12345.678f) sink[0] = acc;
}

// k_scan's read pattern: one wavefront per unit walks T frame rows of 2 KB (two 16-byte loads per lane and frame),
// DEPTH frames in flight, 1024 units = 1024 sequential streams
template <int DEPTH>
__global__ __launch_bounds__(128) void k_streams(const double* __restrict__ src, int T, double* sink)
{
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const double2* s = reinterpret_cast<const double2*>(src + (size_t)blockIdx.x * T * 256);
    double acc = 0.0;
    for (int t = 0; t + DEPTH <= T; t += DEPTH) {
        double2 v[DEPTH][2];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) { v[k][0] = s[(size_t)(t + k) * 128 + lane]; v[k][1] = s[(size_t)(t + k) * 128 + 64 + lane]; }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) acc += v[k][0].x + v[k][0].y + v[k][1].x + v[k][1].y;
    }
    if (acc == 12345.678) sink[0] = acc;
}

__global__ void k_fill_random(unsigned long long* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        p[i] = (h & 0x3FEFFFFF3F7FFFFFull);          // finite doubles / finite floats with random mantissas
    }
}

template <typename F> static float time_ms(F launch, int reps)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const int nblk = 20672;                  // C3: 1024 units x 1292 frames / 64 frames per chunk
    const size_t rd_bytes = (size_t)nblk * 4 * 64 * 64 * 4;        // 64 rows x 256 B per wave
    const size_t wr_bytes = (size_t)nblk * 4 * 64 * 64 * 8;        // 64 rows x 512 B per wave
    float* src; double* dst; double* sink;
    CHECK(hipMalloc(&src, rd_bytes * 2)); CHECK(hipMalloc(&dst, wr_bytes)); CHECK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, (unsigned long long*)src, rd_bytes * 2 / 8);
    hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, (unsigned long long*)dst, wr_bytes / 8);
    CHECK(hipDeviceSynchronize());
    struct R { const char* name; float ms; double bytes; };
    std::vector<R> out;
    out.push_back({"read  64 rows f32 (1.35 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 0, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes});
    out.push_back({"read 128 rows f32 (2.7 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<128, 0, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes * 2});
    out.push_back({"write 64 rows f64 (2.7 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 64, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)wr_bytes});
    out.push_back({"write 64 rows f64, nontemporal", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 64, true>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)wr_bytes});
    out.push_back({"mix read 64 f32 + write 64 f64 (k_stft's 1:2)", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 64, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes + wr_bytes});
    out.push_back({"mix 1:2, nontemporal stores", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 64, true>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes + wr_bytes});
    out.push_back({"mix read 128 f32 + write 32 f64 (2:1 read:write)", time_ms([&] { hipLaunchKernelGGL((k_mix<128, 32, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes * 2 + wr_bytes / 2});
    // the co-running pair of the pipeline: k_stft's mix on one stream, k_scan's 1024 read streams on another
    {
        const int T = 1292, units = 1024;
        double* spec; CHECK(hipMalloc(&spec, (size_t)units * T * 256 * 8)); hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, (unsigned long long*)spec, (size_t)units * T * 256); CHECK(hipDeviceSynchronize());
        const double sbytes = (double)units * T * 256 * 8;
        out.push_back({"1024 read streams of 2 KB rows, 4 in flight (k_scan)", time_ms([&] { hipLaunchKernelGGL((k_streams<4>), dim3(units), dim3(128), 0, 0, spec, T, sink); }, 20), sbytes});
        out.push_back({"4096 read streams, 4 in flight", time_ms([&] { hipLaunchKernelGGL((k_streams<4>), dim3(units * 4), dim3(128), 0, 0, spec, T / 4, sink); }, 20), sbytes});
        out.push_back({"1024 read streams, 16 in flight", time_ms([&] { hipLaunchKernelGGL((k_streams<16>), dim3(units), dim3(128), 0, 0, spec, T, sink); }, 20), sbytes});
        hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
        hipEvent_t e1, e2; CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
        auto both = [&](auto scan_launch) {
            // steady state of the staged pipeline: both streams busy back to back; time = total / iterations
            const int reps = 20;
            CHECK(hipDeviceSynchronize());
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a, 0));
            CHECK(hipStreamWaitEvent(s1, a, 0)); CHECK(hipStreamWaitEvent(s2, a, 0));
            for (int i = 0; i < reps; i++) {
                hipLaunchKernelGGL((k_mix<64, 64, false>), dim3(nblk), dim3(256), 0, s1, src, dst, sink);
                scan_launch(s2);
            }
            CHECK(hipEventRecord(e1, s1)); CHECK(hipEventRecord(e2, s2));
            CHECK(hipStreamWaitEvent(0, e1, 0)); CHECK(hipStreamWaitEvent(0, e2, 0));
            CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            return ms / reps;
        };
        out.push_back({"PAIR: 1:2 mix || 1024 read streams (4 in flight)", both([&](hipStream_t st) { hipLaunchKernelGGL((k_streams<4>), dim3(units), dim3(128), 0, st, spec, T, sink); }), (double)rd_bytes + wr_bytes + sbytes});
        out.push_back({"PAIR: 1:2 mix || 1024 read streams (16 in flight)", both([&](hipStream_t st) { hipLaunchKernelGGL((k_streams<16>), dim3(units), dim3(128), 0, st, spec, T, sink); }), (double)rd_bytes + wr_bytes + sbytes});
        out.push_back({"PAIR: 1:2 mix || 4096 read streams (4 in flight)", both([&](hipStream_t st) { hipLaunchKernelGGL((k_streams<4>), dim3(units * 4), dim3(128), 0, st, spec, T / 4, sink); }), (double)rd_bytes + wr_bytes + sbytes});
    }
    for (auto& r : out) printf("%-56s %7.3f ms  %7.2f TB/s\n", r.name, r.ms, r.bytes / r.ms * 1e-9);
    return 0;
}
