// Probe: issue rate of fp32 FMAs on gfx950 -- scalar v_fma_f32 (VGPR and SGPR multiplicand) and
// packed v_pk_fma_f32 -- at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 512, ACC = 16;

__global__ __launch_bounds__(256) void fma_vgpr(float* out, float a, float b) {
  float acc[ACC];
  float x = threadIdx.x * 1e-3f + a;
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = fmaf(acc[i], x, b + i);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_sgpr(float* out, const float* __restrict__ w) {
  float acc[ACC];
  const float x = threadIdx.x * 1e-3f;
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = fmaf(w[(it & 7) * ACC + i], x, acc[i]);   // uniform -> SGPR
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_packed(float* out, float a, float b) {
  f2 acc[ACC / 2];
  const f2 x = {threadIdx.x * 1e-3f + a, threadIdx.x * 2e-3f + a};
#pragma unroll
  for (int i = 0; i < ACC / 2; ++i) acc[i] = f2{(float)i, (float)i + 1};
  const f2 c = {b, b + 1.f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < ACC / 2; ++i) acc[i] = __builtin_elementwise_fma(acc[i], x, c);
  }
  f2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ACC / 2; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <typename F>
float time_it(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *out, *w;
  hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
  hipMalloc(&w, 8 * ACC * sizeof(float));
  hipMemset(w, 0, 8 * ACC * sizeof(float));
  for (int wps : {1, 2, 4, 8}) {                     // waves per SIMD = blocks per CU (4 waves each)
    const int blocks = 256 * wps;
    const double fmas = (double)blocks * 256 * ITERS * ACC;
    float t1 = time_it([&] { hipLaunchKernelGGL(fma_vgpr, dim3(blocks), dim3(256), 0, 0, out, 1.f, 2.f); });
    float t2 = time_it([&] { hipLaunchKernelGGL(fma_sgpr, dim3(blocks), dim3(256), 0, 0, out, w); });
    float t3 = time_it([&] { hipLaunchKernelGGL(fma_packed, dim3(blocks), dim3(256), 0, 0, out, 1.f, 2.f); });
    printf("waves/SIMD %d: v_fma(vgpr) %.1f TFLOP/s  v_fma(sgpr) %.1f  v_pk_fma %.1f\n", wps,
           2 * fmas / t1 / 1e9, 2 * fmas / t2 / 1e9, 2 * fmas / t3 / 1e9);
  }
  return 0;
}
