// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access width: each kernel streams
// the same 512 MiB buffer once (coalesced, one element per lane per instruction) and sums it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_probe tools/probes/fetch_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/probes/fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void read_kernel(const T* __restrict__ p, long n, float* __restrict__ out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const T v = p[i];
    const float* f = reinterpret_cast<const float*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) s += f[k];
  }
  if (s == 123.456f) out[0] = s;
}
template <typename T>
__global__ void write_kernel(T* __restrict__ p, long n) {
  T v;
  float* f = reinterpret_cast<float*>(&v);
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) f[k] = 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
// rows of 64 floats read as 16-byte lanes, 4 lanes per row (a tile-staging pattern)
__global__ void read_rows64_kernel(const float4* __restrict__ p, long rows, long ld4, float* __restrict__ out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * 4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = p[(i >> 2) * ld4 + (i & 3)];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  const long bytes = 512L << 20;
  void* buf; float* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_kernel<float>, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
    hipLaunchKernelGGL(read_kernel<float2>, dim3(4096), dim3(256), 0, 0, (const float2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read_kernel<float4>, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(read_rows64_kernel, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 1024, 64L, out);
    hipLaunchKernelGGL(write_kernel<float>, dim3(4096), dim3(256), 0, 0, (float*)buf, bytes / 4);
    hipLaunchKernelGGL(write_kernel<float4>, dim3(4096), dim3(256), 0, 0, (float4*)buf, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("streamed %ld bytes per kernel (read_rows64: %ld)\n", bytes, bytes / 4);
  return 0;
}
