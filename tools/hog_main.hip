// tools/hog_main.hip -- keep part of the GPU busy for a few seconds (measurement aid: does a sparse single-file pipeline run
// at a lower shader clock than a busy chip?).  usage: hog_main <seconds> <blocks>      (256 threads per block, FP64 FMA chains)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_hog(double* out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1.0;
    for (int i = 0; i < iters; i++) { a = fma(a, b, c); d = fma(d, b, c); }
    if (a + d == 12345.678) out[0] = a;
}
int main(int argc, char** argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 5.0;
    const int blocks = argc > 2 ? atoi(argv[2]) : 64;
    double* out;
    hipMalloc(&out, 64);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        hipLaunchKernelGGL(k_hog, dim3(blocks), dim3(256), 0, 0, out, 400000);      // ~ a few ms each
        hipDeviceSynchronize();
        n++;
    }
    printf("hog: %ld launches of %d blocks in %.1f s\n", n, blocks, secs);
    return 0;
}
