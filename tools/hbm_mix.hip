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
        const double v = (double)acc + lane;
#pragma unroll 8
        for (int r = 0; r < WR_ROWS; r++) {
            if (NT) __builtin_nontemporal_store(v + r, &d[r * 64 + lane]);
            else d[r * 64 + lane] = v + r;
        }
    } else if (acc == 12345.678f) sink[0] = acc;
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
    CHECK(hipMemset(src, 0, rd_bytes * 2)); CHECK(hipMemset(dst, 0, wr_bytes));
    struct R { const char* name; float ms; double bytes; };
    std::vector<R> out;
    out.push_back({"read  64 rows f32 (1.35 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 0, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes});
    out.push_back({"read 128 rows f32 (2.7 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<128, 0, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes * 2});
    out.push_back({"write 64 rows f64 (2.7 GB)", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 64, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)wr_bytes});
    out.push_back({"write 64 rows f64, nontemporal", time_ms([&] { hipLaunchKernelGGL((k_mix<0, 64, true>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)wr_bytes});
    out.push_back({"mix read 64 f32 + write 64 f64 (k_stft's 1:2)", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 64, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes + wr_bytes});
    out.push_back({"mix 1:2, nontemporal stores", time_ms([&] { hipLaunchKernelGGL((k_mix<64, 64, true>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes + wr_bytes});
    out.push_back({"mix read 128 f32 + write 32 f64 (2:1 read:write)", time_ms([&] { hipLaunchKernelGGL((k_mix<128, 32, false>), dim3(nblk), dim3(256), 0, 0, src, dst, sink); }, 20), (double)rd_bytes * 2 + wr_bytes / 2});
    for (auto& r : out) printf("%-52s %7.3f ms  %7.2f TB/s\n", r.name, r.ms, r.bytes / r.ms * 1e-9);
    return 0;
}
