// tools/cumask_probe.hip -- does a CU-masked stream confine a kernel, and how do mask bits map to XCDs / CUs?
// (measurement aid for the CU-partitioned pipeline of DESIGN.md; NOT part of the product library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ void k_where(unsigned* out)
{
    // HW_REG_HW_ID (id 4): cu_id bits, se_id ...; HW_REG_XCC_ID (id 20)
    unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    // stay a while so blocks spread over everything that is allowed
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000ull) { __builtin_amdgcn_s_sleep(8); }
}
__global__ __launch_bounds__(256) void k_spin(double* out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1.0;
    for (int i = 0; i < iters; i++) { a = fma(a, b, c); d = fma(d, b, c); }
    if (a + d == 12345.678) out[0] = a;
}
int main(int argc, char** argv)
{
    int nbits = argc > 1 ? atoi(argv[1]) : 64;          // how many mask bits to set
    int stride = argc > 2 ? atoi(argv[2]) : 1;          // set every `stride`-th bit
    int first = argc > 3 ? atoi(argv[3]) : 0;           // first bit
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    unsigned mask[8] = {0};
    int set = 0;
    for (int i = first; i < 256 && set < nbits; i += stride) { mask[i >> 5] |= 1u << (i & 31); set++; }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    printf("create: %s  (bits set %d, stride %d, first %d)\n", hipGetErrorString(e), set, stride, first);
    // two masked streams side by side: do complementary masks run concurrently without slowing each other?
    if (argc > 4) {
        unsigned m2[8] = {0};
        int first2 = atoi(argv[4]), n2 = argc > 5 ? atoi(argv[5]) : 64, set2 = 0;
        for (int i = first2; i < 256 && set2 < n2; i++) { m2[i >> 5] |= 1u << (i & 31); set2++; }
        hipStream_t s2; hipExtStreamCreateWithCUMask(&s2, 8, m2);
        double* o2; hipMalloc(&o2, 64);
        hipEvent_t a1, b1, a2, b2; hipEventCreate(&a1); hipEventCreate(&b1); hipEventCreate(&a2); hipEventCreate(&b2);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a1, s); hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s, o2, 20000); hipEventRecord(b1, s);
            hipEventRecord(a2, s2); hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s2, o2, 20000); hipEventRecord(b2, s2);
            hipDeviceSynchronize();
        }
        float m1 = 0, m2s = 0; hipEventElapsedTime(&m1, a1, b1); hipEventElapsedTime(&m2s, a2, b2);
        printf("side by side: stream A (%d bits from %d) %.3f ms, stream B (%d bits from %d) %.3f ms\n", set, first, m1, set2, first2, m2s);
    }
    unsigned* d; hipMalloc(&d, 2 * 4096 * 4);
    hipMemset(d, 0xff, 2 * 4096 * 4);
    hipLaunchKernelGGL(k_where, dim3(2048), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    unsigned h[2 * 2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int seen[8][64]; memset(seen, 0, sizeof(seen));
    int perx[8] = {0};
    for (int b = 0; b < 2048; b++) {
        unsigned hw = h[2 * b], x = h[2 * b + 1] & 7;
        unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;     // gfx9 HW_ID layout: cu_id[11:8], sh_id[12], se_id[15:13]
        unsigned id = (se << 5) | (sh << 4) | cu;
        if (!seen[x][id & 63]) { seen[x][id & 63] = 1; perx[x]++; }
    }
    int tot = 0;
    for (int x = 0; x < 8; x++) { printf("xcc %d: %d distinct CUs used\n", x, perx[x]); tot += perx[x]; }
    printf("total distinct CUs used: %d\n", tot);
    // timing: FP64 spin on masked vs unmasked stream
    double* o; hipMalloc(&o, 64);
    hipStream_t s0; hipStreamCreate(&s0);
    for (int pass = 0; pass < 2; pass++) {
        hipStream_t st = pass ? s : s0;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, st, o, 20000);
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, st, o, 20000);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s stream: spin kernel %.3f ms\n", pass ? "masked" : "full", ms);
    }
    return 0;
}
