// Back-to-back kernel launch cost on one stream, plain launches vs a replayed hipGraph.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void touch(float *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int N = 2000;
    float *buf;
    CK(hipMalloc(&buf, 64 << 20));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int blocks : {1, 1024, 16384}) {
        const int n = blocks * 256;
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float plain = ms * 1e3f / N;
        // the same chain as a graph of 200 nodes, replayed 10 times
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(touch, dim3(blocks), dim3(256), 0, s, buf, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%6d blocks x 256: plain %.2f us/launch, graph %.2f us/node\n", blocks, plain, ms * 1e3f / 2000);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
