// Batched kaldi-style log-mel filterbank ("fbank") for gfx950.
//
// Replaces, for a whole padded batch of utterances in one launch, the
// per-utterance CPU call  dataset/frontend/frontend.py:85-94
// (torchaudio.compliance.kaldi.fbank, whose op graph is stated in
// sample_data/model/frontend.script) and optionally the following
// GlobalCmvnLayer (model/layer/global_cmvn.py:30-38).
//
// Algorithm per frame (snip_edges): 400 samples at hop 160 -> subtract frame
// mean -> pre-emphasis 0.97 (replicate first sample) -> Povey window ->
// zero-pad to 512 -> |rFFT|^2 -> triangular mel filterbank -> log(max(.,eps)).
//
// Mapping to CDNA4: one 256-thread workgroup stages the samples of 32
// consecutive frames of one utterance in LDS with coalesced 4-byte loads
// (HBM is read once: 160 new samples per frame).  Each 64-lane wave packs TWO
// real frames into one complex 512-point FFT (re = frame A, im = frame B);
// the FFT is 8x8x8: three register-resident radix-8 butterflies with two LDS
// transposes, twiddles from a host-computed (double precision) table held in
// LDS.  The mel filterbank is stored compactly (each FFT bin touches <= 2
// filters) and applied from LDS; the log/CMVN epilogue is fused and frames
// are written as contiguous rows.
#include "common.h"

namespace {

constexpr int kFrameLen = 400;
constexpr int kHop = 160;
constexpr int kNfft = 512;
constexpr int kBins = 257;
constexpr int kFramesPerBlock = 32;
constexpr int kWaves = 4;
constexpr int kStage = (kFramesPerBlock - 1) * kHop + kFrameLen;  // 5360 samples

struct c32 {
  float x, y;
};
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ c32 mul_negi(c32 a) { return {a.y, -a.x}; }  // a * (-i)

// In-register forward 8-point DFT (decimation in frequency), natural order out.
__device__ __forceinline__ void dft8(c32 v[8]) {
  const float c = 0.70710678118654752440f;
  c32 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  c32 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  c32 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  c32 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  a5 = {c * (a5.x + a5.y), c * (a5.y - a5.x)};   // * w8^1
  a6 = mul_negi(a6);                             // * w8^2
  a7 = {c * (a7.y - a7.x), -c * (a7.x + a7.y)};  // * w8^3
  // even outputs: DFT4(a0..a3)
  c32 c0 = cadd(a0, a2), c2 = csub(a0, a2), c1 = cadd(a1, a3), c3 = mul_negi(csub(a1, a3));
  v[0] = cadd(c0, c1);
  v[4] = csub(c0, c1);
  v[2] = cadd(c2, c3);
  v[6] = csub(c2, c3);
  // odd outputs: DFT4(a4..a7)
  c32 d0 = cadd(a4, a6), d2 = csub(a4, a6), d1 = cadd(a5, a7), d3 = mul_negi(csub(a5, a7));
  v[1] = cadd(d0, d1);
  v[5] = csub(d0, d1);
  v[3] = cadd(d2, d3);
  v[7] = csub(d2, d3);
}

__global__ __launch_bounds__(256) void fbank_kernel(
    const float* __restrict__ pcm, long pcm_stride, const long* __restrict__ num_samples,
    const float* __restrict__ window,   // [400]
    const float* __restrict__ twiddle,  // [512][2]  exp(-2 pi i m / 512)
    const int* __restrict__ mel_off,    // [M+1] offsets into mel_w
    const int* __restrict__ mel_k0,     // [M] first FFT bin of each filter
    const float* __restrict__ mel_w,    // [nnz]
    int nnz, int num_mel, float eps, float scale_in, const float* __restrict__ cmvn_mean,
    const float* __restrict__ cmvn_istd, float* __restrict__ out, int max_frames,
    long* __restrict__ out_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_pcm = reinterpret_cast<float*>(smem_raw);      // kStage
  c32* s_tw = reinterpret_cast<c32*>(s_pcm + kStage);     // 512
  float* s_win = reinterpret_cast<float*>(s_tw + kNfft);  // 400
  float* s_melw = s_win + kFrameLen;                      // nnz (<= 1024)
  c32* s_fft = reinterpret_cast<c32*>(s_melw + 1024);     // kWaves * 512
  float* s_pow = reinterpret_cast<float*>(s_fft + kWaves * kNfft);  // kWaves * 2 * 260

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * kFramesPerBlock;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long ns = num_samples[b];
  const int nframes = ns >= kFrameLen ? (int)(1 + (ns - kFrameLen) / kHop) : 0;
  if (blockIdx.x == 0 && tid == 0 && out_frames) out_frames[b] = nframes;

  // ---- stage samples + constant tables (coalesced) ----
  const float* src = pcm + (long)b * pcm_stride + (long)f0 * kHop;
  const long remain = ns - (long)f0 * kHop;
  for (int i = tid; i < kStage; i += 256) s_pcm[i] = (i < remain) ? src[i] * scale_in : 0.f;
  for (int i = tid; i < kNfft; i += 256) s_tw[i] = {twiddle[2 * i], twiddle[2 * i + 1]};
  for (int i = tid; i < kFrameLen; i += 256) s_win[i] = window[i];
  for (int i = tid; i < nnz; i += 256) s_melw[i] = mel_w[i];
  __syncthreads();

  c32* fbuf = s_fft + wave * kNfft;
  float* pA = s_pow + wave * 2 * 260;
  float* pB = pA + 260;

  for (int it = 0; it < kFramesPerBlock / (2 * kWaves); ++it) {
    const int fa = it * 2 * kWaves + wave * 2;  // local frame index of frame A
    const float* xa = s_pcm + fa * kHop;
    const float* xb = xa + kHop;
    // frame means (reference: torch.mean over the 400 raw samples)
    float sa = 0.f, sb = 0.f;
    for (int i = lane; i < kFrameLen; i += 64) {
      sa += xa[i];
      sb += xb[i];
    }
    const float ma = wave_sum(sa) * (1.0f / kFrameLen);
    const float mb = wave_sum(sb) * (1.0f / kFrameLen);

    // ---- pass A: lane = n2, element j = n1, sample index n = 64*j + lane ----
    c32 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = 64 * j + lane;
      float ra = 0.f, rb = 0.f;
      if (n < kFrameLen) {
        const int np = n > 0 ? n - 1 : 0;
        const float w = s_win[n];
        const float a0 = __fsub_rn(xa[n], ma), a1 = __fsub_rn(xa[np], ma);
        const float b0 = __fsub_rn(xb[n], mb), b1 = __fsub_rn(xb[np], mb);
        ra = __fmul_rn(__fsub_rn(a0, __fmul_rn(a1, 0.97f)), w);
        rb = __fmul_rn(__fsub_rn(b0, __fmul_rn(b1, 0.97f)), w);
      }
      v[j] = {ra, rb};
    }
    dft8(v);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      c32 t = cmul(v[k1], s_tw[(lane * k1) & 511]);
      fbuf[k1 * 64 + lane] = t;
    }
    __syncthreads();
    // ---- pass B: lane = (k1, m2), element m1 ----
    {
      const int k1 = lane >> 3, m2 = lane & 7;
#pragma unroll
      for (int m1 = 0; m1 < 8; ++m1) v[m1] = fbuf[k1 * 64 + 8 * m1 + m2];
      dft8(v);
      __syncthreads();
#pragma unroll
      for (int q1 = 0; q1 < 8; ++q1) {
        c32 t = cmul(v[q1], s_tw[(8 * m2 * q1) & 511]);
        fbuf[k1 * 64 + q1 * 8 + m2] = t;
      }
    }
    __syncthreads();
    // ---- pass C: lane = (k1, q1), element m2 -> X[k1 + 8 q1 + 64 q2] ----
    {
      const int k1 = lane >> 3, q1 = lane & 7;
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) v[m2] = fbuf[k1 * 64 + q1 * 8 + m2];
      dft8(v);
      __syncthreads();
#pragma unroll
      for (int q2 = 0; q2 < 8; ++q2) fbuf[k1 + 8 * q1 + 64 * q2] = v[q2];
    }
    __syncthreads();
    // ---- un-pack the two real spectra, power ----
    for (int k = lane; k < kBins; k += 64) {
      const c32 z = fbuf[k], zc = fbuf[(kNfft - k) & 511];
      const float ar = z.x + zc.x, ai = z.y - zc.y;
      const float br = z.x - zc.x, bi = z.y + zc.y;
      pA[k] = 0.25f * (ar * ar + ai * ai);
      pB[k] = 0.25f * (br * br + bi * bi);
    }
    __syncthreads();
    // ---- mel filterbank + log + optional CMVN, coalesced row stores ----
    const int ga = f0 + fa, gb = ga + 1;
    for (int m = lane; m < num_mel; m += 64) {
      const int o0 = mel_off[m], o1 = mel_off[m + 1], k0 = mel_k0[m];
      float ea = 0.f, eb = 0.f;
      for (int i = o0; i < o1; ++i) {
        const float w = s_melw[i];
        ea = fmaf(pA[k0 + i - o0], w, ea);
        eb = fmaf(pB[k0 + i - o0], w, eb);
      }
      float la = logf(fmaxf(ea, eps)), lb = logf(fmaxf(eb, eps));
      float mean = 0.f, istd = 1.f;
      if (cmvn_mean) {
        mean = cmvn_mean[m];
        istd = cmvn_istd[m];
      }
      if (ga < max_frames) {
        const float val = ga < nframes ? la : 0.f;
        out[((long)b * max_frames + ga) * num_mel + m] = cmvn_mean ? (val - mean) * istd : val;
      }
      if (gb < max_frames) {
        const float val = gb < nframes ? lb : 0.f;
        out[((long)b * max_frames + gb) * num_mel + m] = cmvn_mean ? (val - mean) * istd : val;
      }
    }
    __syncthreads();
  }
}

}  // namespace

// C ABI -- see include/s2t_mi355.h
extern "C" int s2t_fbank_f32(const float* pcm, long pcm_stride, const long* num_samples,
                             int batch, const float* window400, const float* twiddle512,
                             const int* mel_off, const int* mel_k0, const float* mel_w, int nnz,
                             int num_mel, float eps, float scale_in, const float* cmvn_mean,
                             const float* cmvn_istd, float* out, int max_frames,
                             long* out_frames, void* stream) {
  if (batch <= 0 || max_frames <= 0) return 0;
  if (nnz > 1024 || num_mel > 128 || num_mel <= 0) return -1;
  const size_t smem = sizeof(float) * (kStage + 2 * kNfft + kFrameLen + 1024 +
                                       2 * kWaves * kNfft + kWaves * 2 * 260);
  dim3 grid((max_frames + kFramesPerBlock - 1) / kFramesPerBlock, batch);
  hipLaunchKernelGGL(fbank_kernel, grid, dim3(256), smem, (hipStream_t)stream, pcm, pcm_stride,
                     num_samples, window400, twiddle512, mel_off, mel_k0, mel_w, nnz, num_mel,
                     eps, scale_in, cmvn_mean, cmvn_istd, out, max_frames, out_frames);
  S2T_CHECK_LAUNCH();
  return 0;
}
