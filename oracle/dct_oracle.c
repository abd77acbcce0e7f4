/*
 * dct_oracle.c -- plain-C CPU restatement of the DCT-fingerprint path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file.
 *
 * Follows mgtools/DCTdomain src/fingerprint.py (paths relative to the reference):
 *   idct_quant  :126-142   dct(type=2, norm='ortho') -> keep num -> idct(type=2, norm='ortho')
 *   scale       :110-123   (v - min) / (max - min), no epsilon
 *   quantize    :192-195   stage A over rows (n), stage B over channels (m), (x*127).astype(int8)
 * scipy.fft's orthonormal DCT-II (pocketfft; third-party, not in the reference tree) is
 * restated from its published definition
 *   f_k = s_k sum_t x_t cos(pi k (2t+1) / (2N)),  s_0 = sqrt(1/N), s_k = sqrt(2/N)
 * and its inverse (DCT-III)  x_t = sum_k s_k f_k cos(pi k (2t+1) / (2N)).
 *
 * Written as two explicit steps per axis (forward coefficients, then the short inverse)
 * with the true normalisation and the k = 0 term -- deliberately NOT the fused, DC-free
 * form the HIP kernels use, so that it checks that simplification independently.
 * One liberty, shared with oracle/dct_oracle.py's matrix form: the first row is
 * subtracted before the k >= 1 sums (exact for the min-max scaled result, and it makes a
 * constant channel give exactly 0/0 = NaN like the reference's FFT does).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double PI = 3.14159265358979323846264338327950288;

/* cos(pi p / q) with exact integer argument reduction */
static double cospi_ratio(long long p, long long q) {
    p %= 2 * q;
    if (p > q) p = 2 * q - p;
    double sign = 1.0;
    if (2 * p > q) { p = q - p; sign = -1.0; }
    double r = (4 * p > q) ? sin(PI * (double)(q - 2 * p) / (double)(2 * q)) : cos(PI * (double)p / (double)q);
    return sign * r;
}

/* one axis: in (len x cnt, element (t, c) at in[t*ld + c]) -> out (num x cnt) scaled per column.
 * coef (cnt x num) receives the true ortho coefficients when not NULL. */
static int resample_scale(const double* in, long len, long cnt, long ld, int num, double* out, double* coef) {
    if (num > len) return -2;
    double* ctab = (double*)malloc(sizeof(double) * (size_t)len * (size_t)num);
    double* f = (double*)malloc(sizeof(double) * (size_t)num * (size_t)cnt);
    double* itab = (double*)malloc(sizeof(double) * (size_t)num * (size_t)num);
    if (!ctab || !f || !itab) { free(ctab); free(f); free(itab); return -4; }
    for (long t = 0; t < len; ++t)
        for (int k = 0; k < num; ++k) ctab[t * num + k] = cospi_ratio((long long)k * (2 * t + 1), 2 * len);
    for (int j = 0; j < num; ++j)
        for (int k = 0; k < num; ++k) itab[j * num + k] = cospi_ratio((long long)k * (2 * j + 1), 2 * (long long)num);
    const double s0 = sqrt(1.0 / (double)len), sk = sqrt(2.0 / (double)len);
    const double i0 = sqrt(1.0 / (double)num), ik = sqrt(2.0 / (double)num);
    memset(f, 0, sizeof(double) * (size_t)num * (size_t)cnt);
    for (long t = 0; t < len; ++t) {
        const double* row = in + t * ld;
        for (long c = 0; c < cnt; ++c) {
            const double x = row[c];
            const double d = x - in[c];
            f[c] += x;                                   /* k = 0: plain sum */
            for (int k = 1; k < num; ++k) f[(long)k * cnt + c] += ctab[t * num + k] * d;
        }
    }
    for (long c = 0; c < cnt; ++c) {
        f[c] *= s0;
        for (int k = 1; k < num; ++k) f[(long)k * cnt + c] *= sk;
        if (coef) for (int k = 0; k < num; ++k) coef[c * num + k] = f[(long)k * cnt + c];
    }
    for (long c = 0; c < cnt; ++c) {
        double mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        for (int j = 0; j < num; ++j) {
            double y = i0 * f[c];
            for (int k = 1; k < num; ++k) y += ik * itab[j * num + k] * f[(long)k * cnt + c];
            out[(long)j * cnt + c] = y;
            if (y != y) bad = 1;
            if (y < mn) mn = y;
            if (y > mx) mx = y;
        }
        for (int j = 0; j < num; ++j)
            out[(long)j * cnt + c] = bad ? NAN : (out[(long)j * cnt + c] - mn) / (mx - mn);
    }
    free(ctab); free(f); free(itab);
    return 0;
}

/* x: (n_rows x n_cols) float32 rows of ONE domain (already gathered), leading dim ld.
 * out: n*m values 0..127.  Optional float64 intermediates (NULL to skip):
 *   coef (n_cols x n)  f[:, :n] of src/fingerprint.py:137
 *   yprime (n x n_cols) stage-A result, z (n x m) stage-B result before the *127.
 * Returns 0, -2 (reference's reshape ValueError) or -4 (out of memory). */
int oracle_quantize_layer(const float* x, long n_rows, long n_cols, long ld, int n, int m, signed char* out,
                          double* coef, double* yprime, double* z) {
    if (n_rows < n || n_cols < m) return -2;
    double* xd = (double*)malloc(sizeof(double) * (size_t)n_rows * (size_t)n_cols);
    double* a = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n_cols);
    double* at = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n_cols);
    double* b = (double*)malloc(sizeof(double) * (size_t)m * (size_t)n);
    if (!xd || !a || !at || !b) { free(xd); free(a); free(at); free(b); return -4; }
    for (long t = 0; t < n_rows; ++t)
        for (long c = 0; c < n_cols; ++c) xd[t * n_cols + c] = (double)x[t * ld + c];   /* get_doms: float64 */
    int rc = resample_scale(xd, n_rows, n_cols, n_cols, n, a, coef);                    /* (n x D) */
    if (rc == 0) {
        if (yprime) memcpy(yprime, a, sizeof(double) * (size_t)n * (size_t)n_cols);
        for (int j = 0; j < n; ++j)
            for (long c = 0; c < n_cols; ++c) at[c * n + j] = a[(long)j * n_cols + c];   /* dct.T: (D x n) */
        rc = resample_scale(at, n_cols, n, n, m, b, NULL);                               /* (m x n) */
    }
    if (rc == 0) {
        for (int j = 0; j < n; ++j)
            for (int c = 0; c < m; ++c) {
                const double v = b[(long)c * n + j];                                     /* .T -> (n x m) */
                if (z) z[j * m + c] = v;
                const double q = v * 127.0;
                out[j * m + c] = (q >= 0.0 && q <= 127.0) ? (signed char)(int)q : 0;     /* NaN -> 0 */
            }
    }
    free(xd); free(a); free(at); free(b);
    return rc;
}
