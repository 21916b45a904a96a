// Does capturing a kernel that needs scratch (private segment) into a HIP graph work when it is the kernel's first use?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); printf("%-60s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
__global__ void k_scratch(float* out, int n, int idx) {
    float a[64];
    for (int i = 0; i < 64; ++i) a[i] = out[(threadIdx.x + i) % n];
    out[threadIdx.x] = a[idx & 63] + a[(idx * 7) & 63];       // dynamic index -> the array lives in scratch
}
int main() {
    float* d; CK(hipMalloc(&d, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k_scratch, dim3(4), dim3(256), 0, st, d, 1024, 5);
    CK(hipGetLastError());
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    return 0;
}
