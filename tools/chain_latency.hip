// tools/chain_latency.hip -- what does ONE step of a dependent FP64 recurrence cost a lone wavefront?  (measurement aid, not part
// of the product library; build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_latency.hip -o tools/_bin/chain_latency)
// The onset filter carried through a unit by k_hpf is   y = x + z ;  z = (-x) + pole * y   per frame: add -> mul -> add, each
// rounded separately.  This runs that chain without any memory traffic and reports shader cycles and nanoseconds per step and
// the shader clock the kernel saw -- for one workgroup on an otherwise idle chip (what a single file through the Analyzer gets)
// and for a grid that fills the chip -- with and without a taken branch per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool BRANCH>
__global__ __launch_bounds__(256) void k_chain(const double* __restrict__ xin, double* out, unsigned long long* stamps, int steps, int never)
{
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = xin[(threadIdx.x + 37 * i) & 255];
    double z = xin[threadIdx.x & 255] * 0.5;
    const double pole = 0.98;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s += 8) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const double y = x[i] + z;
            if (BRANCH) {
                // a wave-uniform test that is never true, compiled as a branch around a side effect (the listed-frame test of
                // the old k_hpf step)
                if (__builtin_amdgcn_readfirstlane(s + i) == never) { out[threadIdx.x] = y; asm volatile("s_nop 0" ::: "memory"); }
            }
            z = (-x[i]) + pole * y;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (z == 12345.678) out[0] = z;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

// the VALU mix of one k_hpf frame: floor + mean (two independent operations) + the three dependent ones; MODE 1: the two
// independent operations only on every other step (what a helper wavefront would leave of them: none = MODE 2)
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(const double* __restrict__ xin, double* out, unsigned long long* stamps, int steps, double lf, double mean,
                                             double pole)
{
    __shared__ double xs[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) xs[i] = xin[i & 255];
    __syncthreads();
    double z = xin[threadIdx.x & 255] * 0.5;
    const int lane16 = threadIdx.x & 15;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; s += 8) {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = xs[((s + i) * 16 + lane16) & 511];        // (one LDS read per step, as after the transposition)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double xx;
            if (MODE == 0 || (MODE == 1 && (i & 1))) xx = fmax(x[i], lf) - mean;
            else xx = x[i];
            const double y = xx + z;
            z = (-xx) + pole * y;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (z == 12345.678) out[0] = z;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = r1 - r0; }
}

// Do two wavefronts of one workgroup get in each other's way?  Wavefront 0 runs the dependent chain; wavefront `busy` (1..3,
// 0 = none) runs independent 64-bit integer / FP64 VALU work for about as long (what k_hpf's loader does beside its filter).
__global__ __launch_bounds__(256) void k_two(const double* __restrict__ xin, double* out, unsigned long long* stamps, int steps, int busy)
{
    const int wave = threadIdx.x >> 6;
    double z = xin[threadIdx.x & 255] * 0.5, x0 = xin[(threadIdx.x + 5) & 255];
    if (wave == 0) {
        const unsigned long long c0 = __builtin_readcyclecounter();
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) { const double y = x0 + z; z = (-x0) + 0.98 * y; }
        }
        const unsigned long long c1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) stamps[0] = c1 - c0;
        if (z == 12345.678) out[0] = z;
    } else if (wave == busy) {
        long long a = threadIdx.x, b = 3;
        double q = x0, r = z;
        __shared__ double lds_sink[64];
        for (int s = 0; s < 2 * steps; s++) {
            a = (a << 11) + b; b = b + (a >> 31);                  // 64-bit shift / add (address arithmetic)
            q = fmax(q, r) - 0.125; r = r + 1e-9;                  // independent FP64
            lds_sink[threadIdx.x & 63] = q;                        // an LDS write per turn, like the loader's
        }
        if (a == 12345 && q == 3.0) out[1] = r;
    }
}

// How fast does ONE wavefront get rows out of memory?  16 loads in flight, one dwordx2 per lane and load, nloads loads.
//   MODE 0: 512 contiguous bytes per load, consecutive loads consecutive (a packed stream)
//   MODE 1: lane = (frame-in-group, bin): four pieces of 128 B, 2 KB apart, per load (k_hpf's rows: [frame][256 bins])
//   MODE 2: 512 contiguous bytes per load, consecutive loads 2 KB apart (one whole quarter-row per frame)
template <int MODE>
__global__ __launch_bounds__(64) void k_stream(const double* __restrict__ src, double* out, unsigned long long* stamps, int nloads)
{
    const int lane = threadIdx.x;
    const double* p = src + (size_t)blockIdx.x * 16 + (MODE == 1 ? (size_t)(lane >> 4) * 256 + (lane & 15) : (size_t)lane);
    const size_t step = MODE == 0 ? 64 : MODE == 1 ? 4 * 256 : 256;
    double x[16];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < 16; k++) { x[k] = p[(size_t)k * step]; asm volatile("" ::: "memory"); }
    double acc = 0.0;
    for (int g = 0; g < nloads; g += 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            acc += x[k];
            x[k] = p[(size_t)(g + 16 + k) * step];
            asm volatile("" ::: "memory");
        }
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 12345.678) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = r1 - r0;
}

int main()
{
    double hx[256];
    for (int i = 0; i < 256; i++) hx[i] = -3.0 + 0.01 * i;
    double *dx, *dout;
    unsigned long long* dst;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, 4096 * 8); hipMalloc(&dst, 64);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    int khz = 100000;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const int steps = 200000;
    for (int busy = 0; busy < 4; busy++) {
        unsigned long long st[2] = {0, 0};
        hipLaunchKernelGGL(k_two, dim3(1), dim3(256), 0, 0, dx, dout, dst, steps, busy);
        hipDeviceSynchronize();
        hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
        printf("chain on wavefront 0, busy VALU loop on wavefront %d of the same workgroup (0 = none): %.1f cycles/step\n", busy, (double)st[0] / steps);
    }
    {
        double* big;
        const size_t nbig = (size_t)64 << 20;                      // 512 MB of doubles
        hipMalloc(&big, nbig * 8);
        hipMemset(big, 0, nbig * 8);
        for (int rep = 0; rep < 2; rep++)
            for (int mode = 0; mode < 3; mode++)
                for (int blocks : {1, 16}) {
                    const int nloads = mode == 0 ? 52000 : mode == 1 ? 3200 * 4 : 12800;      // all within 64 M doubles incl. the 16 ahead
                    unsigned long long st[2] = {0, 0};
                    hipDeviceSynchronize();
                    if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(blocks), dim3(64), 0, 0, big + (size_t)rep * 1024, dout, dst, nloads);
                    else if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(64), 0, 0, big + (size_t)rep * 1024, dout, dst, nloads);
                    else hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(64), 0, 0, big + (size_t)rep * 1024, dout, dst, nloads);
                    hipDeviceSynchronize();
                    hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
                    printf("stream mode %d (0 packed, 1 four 128-B pieces 2 KB apart per load, 2 512 B per load 2 KB apart), %2d wavefronts on %2d CUs, rep %d: %.1f ns per load\n",
                           mode, blocks, blocks, rep, (double)st[1] / (double)khz * 1e6 / nloads);
                }
        hipFree(big);
    }
    for (int mode = 0; mode < 3; mode++) {
        unsigned long long st[2] = {0, 0};
        if (mode == 0) hipLaunchKernelGGL(k_mix<0>, dim3(1), dim3(64), 0, 0, dx, dout, dst, steps, -2.5, 0.125, 0.98);
        else if (mode == 1) hipLaunchKernelGGL(k_mix<1>, dim3(1), dim3(64), 0, 0, dx, dout, dst, steps, -2.5, 0.125, 0.98);
        else hipLaunchKernelGGL(k_mix<2>, dim3(1), dim3(64), 0, 0, dx, dout, dst, steps, -2.5, 0.125, 0.98);
        hipDeviceSynchronize();
        hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
        printf("k_hpf mix, mode %d (0: floor + mean every step, 1: every other step, 2: never): %.1f cycles/step  %.2f ns/step\n", mode,
               (double)st[0] / steps, (double)st[1] / (double)khz * 1e6 / steps);
    }
    for (int rep = 0; rep < 1; rep++) {
        for (int variant = 0; variant < 2; variant++) {
            for (int blocks : {1, 4, 256, 2048}) {
                for (int threads : {64, 256}) {
                    unsigned long long st[2] = {0, 0};
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0, 0);
                    if (variant == 0) hipLaunchKernelGGL(k_chain<false>, dim3(blocks), dim3(threads), 0, 0, dx, dout, dst, steps, -1);
                    else hipLaunchKernelGGL(k_chain<true>, dim3(blocks), dim3(threads), 0, 0, dx, dout, dst, steps, -1);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    hipMemcpy(st, dst, 16, hipMemcpyDeviceToHost);
                    const double ns = (double)st[1] / (double)khz * 1e6 / steps;
                    printf("rep %d  %s  blocks %4d x %3d threads: %.1f cycles/step  %.2f ns/step  shader clock %.0f MHz  (kernel %.3f ms)\n", rep,
                           variant ? "branch per step" : "straight line  ", blocks, threads, (double)st[0] / steps, ns,
                           (double)st[0] * (double)khz / (double)st[1] / 1e3, ms);
                    hipEventDestroy(e0); hipEventDestroy(e1);
                }
            }
        }
    }
    return 0;
}
