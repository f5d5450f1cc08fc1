// tools/graph_latency.hip -- what does a chain of N tiny dependent kernels cost in a stream, and as a hipGraph?  (measurement aid for
// the one-file path: its chain is 19-21 dispatches, thirteen of which run ~5 us each with nothing to do)
// build: hipcc --offload-arch=gfx950 -O3 tools/graph_latency.hip -o tools/_bin/graph_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(int* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i & 63] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int* d;
    hipMalloc(&d, 256);
    hipMemset(d, 0, 256);
    hipStream_t st;
    hipStreamCreate(&st);
    const int reps = 200;
    for (int N : {1, 5, 20}) {
        for (int blocks : {1, 256}) {
            // plain stream launches
            for (int w = 0; w < 3; w++) { for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(64), 0, st, d, i); hipStreamSynchronize(st); }
            double t0 = now();
            for (int r = 0; r < reps; r++) { for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(64), 0, st, d, i); hipStreamSynchronize(st); }
            const double t_stream = (now() - t0) / reps * 1e6;
            // the same chain as a graph
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(64), 0, st, d, i);
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int w = 0; w < 3; w++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
            t0 = now();
            for (int r = 0; r < reps; r++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
            const double t_graph = (now() - t0) / reps * 1e6;
            printf("chain of %2d tiny kernels (%3d blocks): stream %.1f us per chain (%.2f per kernel), graph %.1f us per chain (%.2f per kernel)\n", N, blocks,
                   t_stream, t_stream / N, t_graph, t_graph / N);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
