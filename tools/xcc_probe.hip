#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
}
int main() {
    int* d; hipMalloc(&d, 64 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d%c", h[i], (i % 16 == 15) ? '\n' : ' ');
    return 0;
}
