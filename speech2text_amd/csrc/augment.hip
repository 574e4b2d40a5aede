// Batched on-device data augmentation next to the fbank kernel:
// the reference runs these per utterance on CPU DataLoader workers
// (dataset/frontend/data_augmentation.py:13-56 AddNoise, :59-118 MixFeats, :150-196 SpecAugment;
// dataset/utils.py:182-202 batch() padding).  The random decisions stay on the host (same
// random.* calls per utterance as the reference); each op is ONE launch over the padded batch.
// All of them stream the batch once: HBM-bound, algorithmic bytes = read + write of the batch.
#include "common.h"

namespace {

// sum over an utterance's valid elements of f(x): mode 0 exp(x) (MixFeats.compute_energy),
// mode 1 x^2 (AddNoise.rms_db numerator).  x [B][stride] rows of len[b]*D valid elements.
__global__ __launch_bounds__(256) void row_energy_kernel(const float* __restrict__ x, long stride,
                                                         const long* __restrict__ len, int D,
                                                         int mode, float* __restrict__ out) {
  __shared__ float scratch[8];
  const int b = blockIdx.x;
  const long n = len[b] * D;
  const float* p = x + (long)b * stride;
  float acc = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float v = p[i];
    acc += mode == 0 ? expf(v) : v * v;
  }
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) out[b] = acc;
}

// out[b][t][d] = mix(src[b][t][d], noise[b][(start[b] + t) % nlen[b]][d]) for t < slen[b];
// frames past slen[b] are copied through.  mode 0 (MixFeats, log-mel):
//   gain = src_e > 0 && noise_e > 0 ? src_e 10^(-snr/10) / noise_e : 1;  log(max(e^a + gain e^b, 1e-10))
// mode 1 (AddNoise, PCM, D = 1): gain_db = min(rms_db(src) - rms_db(noise) - snr, max_gain);
//   clip(a + b 10^(gain_db/20), -1, 1)
__global__ __launch_bounds__(256) void mix_kernel(const float* __restrict__ src, long sstride,
                                                  const long* __restrict__ slen,
                                                  const float* __restrict__ noise, long nstride,
                                                  const long* __restrict__ nlen,
                                                  const long* __restrict__ start,
                                                  const float* __restrict__ snr,
                                                  const float* __restrict__ src_e,
                                                  const float* __restrict__ noise_e, int D,
                                                  int mode, float max_gain_db, long rows_max,
                                                  float* __restrict__ out) {
  const int b = blockIdx.y;
  const long L = slen[b], NL = nlen[b], st = start[b];
  float gain;
  if (mode == 0) {
    gain = 1.f;
    if (src_e[b] > 0.f && noise_e[b] > 0.f)
      gain = src_e[b] * powf(10.f, -snr[b] / 10.f) / noise_e[b];
  } else {
    const float drms = 10.f * log10f(src_e[b] / (float)(L * D));
    const float nrms = 10.f * log10f(noise_e[b] / (float)(NL * D));
    gain = powf(10.f, fminf(drms - nrms - snr[b], max_gain_db) / 20.f);
  }
  const float* s = src + (long)b * sstride;
  const float* nz = noise + (long)b * nstride;
  float* o = out + (long)b * sstride;
  const long total = rows_max * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long t = i / D;
    const int d = (int)(i - t * D);
    float a = s[i];
    if (t < L && NL > 0) {
      const float bv = nz[((st + t) % NL) * D + d];
      if (mode == 0) a = logf(fmaxf(expf(a) + gain * expf(bv), 1.0e-10f));
      else a = fminf(fmaxf(a + bv * gain, -1.f), 1.f);
    }
    o[i] = a;
  }
}

// SpecAugment in place: zero feats[b][t][f] when t in any of the nt time spans or f in any of the
// nf frequency spans of utterance b.  spans [B][n][2] = (start, end) int32, end exclusive.
__global__ __launch_bounds__(256) void specaug_kernel(float* __restrict__ feats, int T, int F,
                                                      const int* __restrict__ tspan, int nt,
                                                      const int* __restrict__ fspan, int nf) {
  const int b = blockIdx.y;
  float* x = feats + (long)b * T * F;
  const long total = (long)T * F;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i / F), f = (int)(i - (long)t * F);
    bool hit = false;
    for (int k = 0; k < nt; ++k) hit |= t >= tspan[(b * nt + k) * 2] && t < tspan[(b * nt + k) * 2 + 1];
    for (int k = 0; k < nf; ++k) hit |= f >= fspan[(b * nf + k) * 2] && f < fspan[(b * nf + k) * 2 + 1];
    if (hit) x[i] = 0.f;
  }
}

// dataset/utils.py batch(): rows of a packed buffer -> zero-padded (B, Lmax, D)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ packed,
                                                       const long* __restrict__ offsets, int D,
                                                       long Lmax, float* __restrict__ out) {
  const int b = blockIdx.y;
  const long beg = offsets[b], n = (offsets[b + 1] - beg);   // elements (rows * D)
  const long total = Lmax * D;
  float* o = out + (long)b * total;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    o[i] = i < n ? packed[beg + i] : 0.f;
}

inline unsigned grid_x(long total) {
  long g = (total + 255) / 256;
  return (unsigned)(g > 1024 ? 1024 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int s2t_row_energy(const float* x, long stride, const long* len, int B, int D, int mode,
                              float* out, void* stream) {
  if (B <= 0) return 0;
  if (D <= 0 || mode < 0 || mode > 1) return -1;
  hipLaunchKernelGGL(row_energy_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, stride, len,
                     D, mode, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_mix(const float* src, long sstride, const long* slen, const float* noise,
                       long nstride, const long* nlen, const long* start, const float* snr,
                       const float* src_e, const float* noise_e, int B, long rows_max, int D,
                       int mode, float max_gain_db, float* out, void* stream) {
  if (B <= 0 || rows_max <= 0) return 0;
  if (D <= 0 || mode < 0 || mode > 1) return -1;
  hipLaunchKernelGGL(mix_kernel, dim3(grid_x(rows_max * D), B), dim3(256), 0, (hipStream_t)stream,
                     src, sstride, slen, noise, nstride, nlen, start, snr, src_e, noise_e, D, mode,
                     max_gain_db, rows_max, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_specaug(float* feats, int B, int T, int F, const int* tspan, int nt,
                           const int* fspan, int nf, void* stream) {
  if (B <= 0 || T <= 0 || F <= 0) return 0;
  if (nt < 0 || nf < 0) return -1;
  hipLaunchKernelGGL(specaug_kernel, dim3(grid_x((long)T * F), B), dim3(256), 0,
                     (hipStream_t)stream, feats, T, F, tspan, nt, fspan, nf);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_pad_rows(const float* packed, const long* offsets, int B, long Lmax, int D,
                            float* out, void* stream) {
  if (B <= 0 || Lmax <= 0) return 0;
  if (D <= 0) return -1;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_x(Lmax * D), B), dim3(256), 0, (hipStream_t)stream,
                     packed, offsets, D, Lmax, out);
  S2T_CHECK_LAUNCH();
  return 0;
}
