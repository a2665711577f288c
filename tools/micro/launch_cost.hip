// Experiment helper (not part of the product): what does a dependent kernel launch cost as a function of its geometry?
// Empty kernels of different grid / block sizes replayed back to back from one hipGraph (each launch depends on the one
// before, like the kernels of a decode step).   hipcc -O2 --offload-arch=gfx950 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void lds_kernel(int* p) { __shared__ int s[2048]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (p && s[(threadIdx.x + 1) % blockDim.x] == -1) *p = 1; }
__global__ void store_kernel(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }

static double run(void (*launch)(hipStream_t, int, int), int grid, int block, int n) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(st, grid, block);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    return 1e3 * ms / (5.0 * n);
}

static float* g_buf;
static void l_empty(hipStream_t st, int g, int b) { hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(b), 0, st, (int*)nullptr); }
static void l_lds(hipStream_t st, int g, int b) { hipLaunchKernelGGL(lds_kernel, dim3(g), dim3(b), 0, st, (int*)nullptr); }
static void l_store(hipStream_t st, int g, int b) { hipLaunchKernelGGL(store_kernel, dim3(g), dim3(b), 0, st, g_buf); }

int main() {
    hipMalloc(&g_buf, 4096 * 1024 * sizeof(float));
    const int geo[][2] = {{1, 64}, {64, 64}, {256, 64}, {1024, 64}, {64, 256}, {256, 256}, {64, 512}, {128, 512}, {256, 512}, {512, 512}, {256, 1024}};
    for (auto& ge : geo) {
        printf("grid %4d x %4d threads (%5d waves): empty %.2f us, LDS+barrier %.2f us, one store per thread %.2f us per dependent launch\n", ge[0], ge[1],
               ge[0] * ge[1] / 64, run(l_empty, ge[0], ge[1], 200), run(l_lds, ge[0], ge[1], 200), run(l_store, ge[0], ge[1], 200));
    }
    return 0;
}
