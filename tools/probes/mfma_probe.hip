// Probe (scratch): verifies the v_mfma_f32_32x32x2_f32 operand / accumulator lane maps and the
// "accumulator as the next MFMA's A^T operand" identity used by the attention backward.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A, const float* B, const float* Q, float* D, float* Z) {
  // D(32x32) = A(32x32) * B(32x32);  Z(32x32) = D^T * Q  (Q 32x32)
  const int l = threadIdx.x, lo = l & 31, hi = l >> 5;
  f32x16 acc = {0};
  for (int s = 0; s < 16; ++s) {
    const int k = hi + 2 * s;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[lo * 32 + k], B[k * 32 + lo], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    D[row * 32 + lo] = acc[r];
  }
  // Z = D^T Q: A-operand of k-step s is accumulator register s (row(s,hi) of D, col lo of D)
  f32x16 z = {0};
  for (int s = 0; s < 16; ++s) {
    const int row = (s & 3) + 8 * (s >> 2) + 4 * hi;   // the k index this lane supplies
    z = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[s], Q[row * 32 + lo], z, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    Z[row * 32 + lo] = z[r];
  }
}

int main() {
  std::vector<float> A(1024), B(1024), Q(1024), D(1024), Z(1024);
  for (int i = 0; i < 1024; ++i) { A[i] = sinf(i * 0.37f); B[i] = cosf(i * 0.11f + 1); Q[i] = sinf(i * 0.05f + 2); }
  float *dA, *dB, *dQ, *dD, *dZ;
  hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dQ, 4096); hipMalloc(&dD, 4096); hipMalloc(&dZ, 4096);
  hipMemcpy(dA, A.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 4096, hipMemcpyHostToDevice);
  hipMemcpy(dQ, Q.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dQ, dD, dZ);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(Z.data(), dZ, 4096, hipMemcpyDeviceToHost);
  double eD = 0, eZ = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double d = 0; for (int k = 0; k < 32; ++k) d += (double)A[i * 32 + k] * B[k * 32 + j];
    eD = fmax(eD, fabs(d - D[i * 32 + j]));
  }
  for (int j = 0; j < 32; ++j) for (int n = 0; n < 32; ++n) {
    double z = 0; for (int i = 0; i < 32; ++i) z += (double)D[i * 32 + j] * Q[i * 32 + n];
    eZ = fmax(eZ, fabs(z - Z[j * 32 + n]));
  }
  printf("max err D=A*B: %g   Z=D^T*Q: %g\n", eD, eZ);
  return 0;
}
