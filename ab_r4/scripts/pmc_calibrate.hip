// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths our kernels use
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O3 scripts/pmc_calibrate.hip -o /tmp/pmc_calibrate
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o c -- /tmp/pmc_calibrate
// Each kernel streams a 256 MiB buffer (larger than L2; cold for every launch) once.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T>
__global__ void read_k(const T *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        acc += *reinterpret_cast<float *>(&v);
    }
    if (acc == 12345.678f) *out = acc;
}
template <typename T>
__global__ void write_k(T *__restrict__ p, size_t n) {
    T v;
    float *f = reinterpret_cast<float *>(&v);
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) f[k] = 1.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const size_t bytes = 256ull << 20;
    void *buf; float *out;
    hipMalloc(&buf, bytes); hipMalloc((void **)&out, 4);
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_k<float>, dim3(4096), dim3(256), 0, 0, (const float *)buf, bytes / 4, out);
        hipLaunchKernelGGL(read_k<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, out);
        hipLaunchKernelGGL(read_k<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(write_k<float>, dim3(4096), dim3(256), 0, 0, (float *)buf, bytes / 4);
        hipLaunchKernelGGL(write_k<float2>, dim3(4096), dim3(256), 0, 0, (float2 *)buf, bytes / 8);
        hipLaunchKernelGGL(write_k<float4>, dim3(4096), dim3(256), 0, 0, (float4 *)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel\n", bytes);
    return 0;
}
