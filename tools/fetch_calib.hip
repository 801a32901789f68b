// Calibrates rocprofv3 FETCH_SIZE for coalesced 4-byte-per-lane loads (the staging access shape).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void read_dword(const float* __restrict__ in, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) acc += in[i];
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    size_t n = (size_t)1 << 30;  // 4 GiB of floats... 1Gi elements = 4 GiB (beyond the 256 MiB Infinity Cache)
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, 4);
    hipMemset(in, 0, n * 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, in, out, n);
    hipDeviceSynchronize();
    printf("read %zu bytes per launch\n", n * 4);
    return 0;
}
