// BEST-RQ label kernel for gfx950: frame stacking (two unfold(1,3,2) = 9 taps over 7 frames)
// -> random projection (720 -> D) -> nearest code by cosine similarity (== nearest by
// euclidean distance between the L2-normalised vectors) -> label = index + 1.
// Reference: model/ssl/best_rq.py:168-217 (_get_subsampling_arrangment), :259-294
// (_make_label).  Integer output: to make the argmax independent of reduction order the
// whole computation runs in fp64 on the fp32 inputs (products of two fp32 are exact in fp64),
// ties break towards the lower index like torch.argmax/argmin.
//
// One wave per label.  stacked[b,t2, d*9 + k1*3 + k2] = feats[b, 4*t2 + 2*k2 + k1, d].
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bestrq_labels_kernel(
    const float* __restrict__ feats, int B, int T, int F, const float* __restrict__ proj, int D,
    const float* __restrict__ codebooks, int ncb, int K, int T2, long* __restrict__ labels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* s_t = reinterpret_cast<double*>(smem_raw) + (threadIdx.x >> 6) * D;  // per-wave target
  const long lab = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (lab >= (long)B * T2) return;
  const int b = (int)(lab / T2), t2 = (int)(lab % T2);
  const float* fb = feats + ((long)b * T + 4 * t2) * F;
  // projection: lane handles output o = lane % D for input slice (lane / D), then reduce
  const int o = lane % D, part = lane / D, nparts = 64 / D;
  double acc = 0.0;
  if (part < nparts) {
    const int nin = F * 9;
    for (int i = part; i < nin; i += nparts) {
      const int d = i / 9, k1 = (i % 9) / 3, k2 = i % 3;
      acc += (double)fb[(long)(2 * k2 + k1) * F + d] * (double)proj[(long)i * D + o];
    }
  }
  // reduce the parts (lanes o, o+D, o+2D, ...)
  for (int off = 32; off >= D; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane < D) s_t[lane] = acc;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double nrm = 0.0;
  for (int d = 0; d < D; ++d) nrm += s_t[d] * s_t[d];
  nrm = sqrt(nrm);
  const double inv_t = 1.0 / fmax(nrm, 1e-12);
  for (int cb = 0; cb < ncb; ++cb) {
    const float* C = codebooks + (long)cb * K * D;
    double best = -1e300;
    int besti = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
      const float* c = C + (long)k * D;
      double dot = 0.0, cn = 0.0;
      for (int d = 0; d < D; ++d) {
        const double cv = (double)c[d];
        dot += s_t[d] * cv;
        cn += cv * cv;
      }
      const double sim = dot * inv_t / fmax(sqrt(cn), 1e-12);
      if (sim > best) {
        best = sim;
        besti = k;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(besti, off, 64);
      if (ob > best || (ob == best && oi < besti)) {
        best = ob;
        besti = oi;
      }
    }
    if (lane == 0) labels[((long)cb * B + b) * T2 + t2] = (long)besti + 1;
  }
}

}  // namespace

// feats (B,T,F) f32; proj (F*9, D); codebooks (ncb, K, D); labels (ncb, B, T2) int64,
// T2 = ((T-3)/2+1 - 3)/2 + 1.  D must be a power of two <= 64.
extern "C" int s2t_bestrq_labels(const float* feats, int B, int T, int F, const float* proj,
                                 int D, const float* codebooks, int ncb, int K, int T2,
                                 long* labels, void* stream) {
  if (B <= 0 || T2 <= 0) return 0;
  if (D <= 0 || D > 64 || (D & (D - 1)) || F <= 0 || K <= 0 || ncb <= 0) return -1;
  if (4 * (T2 - 1) + 6 >= T) return -1;
  const long n = (long)B * T2;
  hipLaunchKernelGGL(bestrq_labels_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256),
                     sizeof(double) * 4 * D, (hipStream_t)stream, feats, B, T, F, proj, D,
                     codebooks, ncb, K, T2, labels);
  S2T_CHECK_LAUNCH();
  return 0;
}
