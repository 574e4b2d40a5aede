/* ORACLE (test infrastructure, never on the product path).
 *
 * Plain-C restatement of k2 v1.24.3 mutual_information_recursion (k2/python/k2/mutual_information.py
 * -> k2/csrc/mutual_information_cpu.cu), the lattice recursion behind
 * k2.rnnt_loss_smoothed / k2.rnnt_loss_pruned as called from the reference at
 * model/joiner/joiner.py:100-110 and model/loss/pruned_rnnt_loss.py:39-48.
 * PARITY UNPINNED (k2 is absent from the reference tree and this image); it is checked against
 * oracle/k2_rnnt.py::mutual_information_np, which is anchored by brute-force path enumeration.
 *
 *   p[s,t] = logaddexp(p[s-1,t] + px[s-1,t], p[s,t-1] + py[s,t-1]),  p[0,0] = 0
 *   ans[b] = p[S_b, T_b];  px_grad / py_grad = arc occupation probabilities.
 *
 * Layouts: px (B,S,T+1), py (B,S+1,T), p (B,S+1,T+1), boundary (B,4) int64 = (0,0,S_b,T_b).
 * Built by oracle/build.py:  gcc -O2 -fopenmp -shared -fPIC.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline float logaddexpf_(float x, float y) {
    if (x == y) return x + 0.693147180559945309417232121458176568f;
    float d = x - y;
    if (d > 0) return x + log1pf(expf(-d));
    if (d <= 0) return y + log1pf(expf(d));
    return d; /* nan */
}

int oracle_mutual_information(const float* px, const float* py, const int64_t* boundary, int B,
                              int S, int T, float* p, float* ans, float* gx, float* gy) {
    const int T1 = T + 1;
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const int Sb = (int)boundary[4 * b + 2], Tb = (int)boundary[4 * b + 3];
        const float* pxb = px + (size_t)b * S * T1;
        const float* pyb = py + (size_t)b * (S + 1) * T;
        float* pb = p + (size_t)b * (S + 1) * T1;
        for (int i = 0; i < (S + 1) * T1; ++i) pb[i] = -INFINITY;
        pb[0] = 0.0f;
        for (int t = 1; t <= Tb; ++t) pb[t] = pb[t - 1] + pyb[t - 1];
        for (int s = 1; s <= Sb; ++s) {
            float* row = pb + (size_t)s * T1;
            const float* up = pb + (size_t)(s - 1) * T1;
            const float* pxr = pxb + (size_t)(s - 1) * T1;
            const float* pyr = pyb + (size_t)s * T;
            float cur = up[0] + pxr[0];
            row[0] = cur;
            for (int t = 1; t <= Tb; ++t) {
                cur = logaddexpf_(up[t] + pxr[t], cur + pyr[t - 1]);
                row[t] = cur;
            }
        }
        ans[b] = pb[(size_t)Sb * T1 + Tb];
        if (!gx || !gy) continue;
        float* gxb = gx + (size_t)b * S * T1;
        float* gyb = gy + (size_t)b * (S + 1) * T;
        for (int i = 0; i < S * T1; ++i) gxb[i] = 0.0f;
        for (int i = 0; i < (S + 1) * T; ++i) gyb[i] = 0.0f;
        double* pg = (double*)calloc((size_t)(Sb + 2) * (Tb + 2), sizeof(double));
        if (!pg) {
            rc = 1;
            continue;
        }
        const int W = Tb + 2;
        pg[(size_t)Sb * W + Tb] = 1.0;
        for (int s = Sb; s >= 0; --s)
            for (int t = Tb; t >= 0; --t) {
                if (s == Sb && t == Tb) continue;
                double xg = 0.0, yg = 0.0;
                const float pst = pb[(size_t)s * T1 + t];
                if (s < Sb) {
                    const double nxt = pg[(size_t)(s + 1) * W + t];
                    const float pn = pb[(size_t)(s + 1) * T1 + t];
                    if (nxt != 0.0 && isfinite(pn)) {
                        const double e = exp((double)pst + (double)pxb[(size_t)s * T1 + t] - (double)pn);
                        xg = isfinite(e) ? nxt * e : 0.0;
                    }
                    gxb[(size_t)s * T1 + t] = (float)xg;
                }
                if (t < Tb) {
                    const double nxt = pg[(size_t)s * W + t + 1];
                    const float pn = pb[(size_t)s * T1 + t + 1];
                    if (nxt != 0.0 && isfinite(pn)) {
                        const double e = exp((double)pst + (double)pyb[(size_t)s * T + t] - (double)pn);
                        yg = isfinite(e) ? nxt * e : 0.0;
                    }
                    gyb[(size_t)s * T + t] = (float)yg;
                }
                pg[(size_t)s * W + t] = xg + yg;
            }
        free(pg);
    }
    return rc;
}
