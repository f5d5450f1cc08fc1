// tools/hog.hip -- contention probes for tools/contention.py (NOT part of the product library).
//   hog_valu: FP64 FMA chains, no memory traffic, `waves` wavefronts per SIMD worth of blocks
//   hog_hbm : streaming read+write of a big buffer (float4), no arithmetic
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void k_hog_valu(double* out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1.0, e = a + 2.0, f = a + 3.0;
    for (int i = 0; i < iters; i++) {
        a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c);
    }
    if (a + d + e + f == 12345.678) out[0] = a;
}
__global__ __launch_bounds__(256) void k_hog_hbm(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" void hog_valu(void* out, int blocks, int iters, void* stream)
{
    hipLaunchKernelGGL(k_hog_valu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)out, iters);
}
extern "C" void hog_hbm(const void* src, void* dst, size_t nbytes, int blocks, void* stream)
{
    hipLaunchKernelGGL(k_hog_hbm, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, nbytes / 16);
}
