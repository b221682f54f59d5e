/* Plain-C restatement of the integer/index arithmetic of the hot path plus naive fp32 convolutions.
 *
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Built by oracle/Makefile into oracle/_build/libvsseg_oracle_c.so and
 * loaded only by tests/, smoke() and bench.py's cpu_baseline leg.
 *
 *  - vsseg_c_swi_starts / vsseg_c_gaussian_1d: MONAI 0.4.0 sliding-window geometry (call site
 *    ref:params/VSparams.py:568-574; algorithm un-vendored, SURVEY.md App. B — parity unpinned, pinned by
 *    known-answer tables in tests/test_swi_geometry.py).
 *  - vsseg_c_conv3d / vsseg_c_conv_transpose3d: textbook NCDHW loops equal to what
 *    ref:params/networks/blocks/convolutions.py:114-146 asks torch for; an implementation-independent cross-check of
 *    the torch-functional oracle at small sizes.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* Window starts along one dimension. Returns the count; writes starts[]. interval follows
 * scan_interval = roi if roi==size else max((int)(roi*(1-overlap)),1) computed in double like Python. */
int vsseg_c_swi_starts(int size_in, int roi, double overlap, int* padded_out, int* pad_before_out, int* interval_out, int* starts, int max_starts) {
  int size = size_in > roi ? size_in : roi;
  int diff = roi - size_in > 0 ? roi - size_in : 0;
  int interval = (roi == size) ? roi : (int)((double)roi * (1.0 - overlap));
  if (interval < 1) interval = 1;
  int num = (int)ceil((double)size / (double)interval);
  int scan = -1;
  for (int d = 0; d < num; ++d)
    if (d * interval + roi >= size) { scan = d + 1; break; }
  if (scan < 0 || scan > max_starts) return -1;
  for (int d = 0; d < scan; ++d) {
    int s = d * interval;
    int over = s + roi - size;
    starts[d] = s - (over > 0 ? over : 0);
  }
  *padded_out = size;
  *pad_before_out = diff / 2;
  *interval_out = interval;
  return scan;
}

/* erf-approximated 1-D Gaussian taps (MONAI gaussian_1d, approx="erf", truncated=4). Returns number of taps (2*tail+1). */
int vsseg_c_gaussian_1d(double sigma, float* taps, int max_taps) {
  double t4 = sigma * 4.0;
  int tail = (int)((t4 > 0.5 ? t4 : 0.5) + 0.5);
  int n = 2 * tail + 1;
  if (n > max_taps) return -1;
  float t = (float)(0.70710678 / fabs(sigma));
  for (int i = 0; i < n; ++i) {
    float x = (float)(i - tail);
    float v = 0.5f * (erff(t * (x + 0.5f)) - erff(t * (x - 0.5f)));
    taps[i] = v < 0.f ? 0.f : v;
  }
  return n;
}

/* y[N,Co,Xo,Yo,Zo] = conv3d(x[N,Ci,X,Y,Z], w[Co,Ci,kx,ky,kz]) + b, stride s, zero padding p. */
void vsseg_c_conv3d(const float* x, const float* w, const float* b, float* y, int N, int Ci, int X, int Y, int Z, int Co, const int* k, const int* s, const int* p) {
  int Xo = (X + 2 * p[0] - k[0]) / s[0] + 1, Yo = (Y + 2 * p[1] - k[1]) / s[1] + 1, Zo = (Z + 2 * p[2] - k[2]) / s[2] + 1;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int xo = 0; xo < Xo; ++xo)
        for (int yo = 0; yo < Yo; ++yo)
          for (int zo = 0; zo < Zo; ++zo) {
            double acc = b ? b[co] : 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int a = 0; a < k[0]; ++a) {
                int xi = xo * s[0] - p[0] + a;
                if (xi < 0 || xi >= X) continue;
                for (int c = 0; c < k[1]; ++c) {
                  int yi = yo * s[1] - p[1] + c;
                  if (yi < 0 || yi >= Y) continue;
                  for (int d = 0; d < k[2]; ++d) {
                    int zi = zo * s[2] - p[2] + d;
                    if (zi < 0 || zi >= Z) continue;
                    acc += (double)x[(((size_t)(n * Ci + ci) * X + xi) * Y + yi) * Z + zi] * w[((((size_t)co * Ci + ci) * k[0] + a) * k[1] + c) * k[2] + d];
                  }
                }
              }
            y[((((size_t)n * Co + co) * Xo + xo) * Yo + yo) * Zo + zo] = (float)acc;
          }
}

/* y = conv_transpose3d(x[N,Ci,X,Y,Z], w[Ci,Co,kx,ky,kz]) + b with stride s, padding p, output_padding op. */
void vsseg_c_conv_transpose3d(const float* x, const float* w, const float* b, float* y, int N, int Ci, int X, int Y, int Z, int Co, const int* k, const int* s, const int* p, const int* op) {
  int Xo = (X - 1) * s[0] - 2 * p[0] + k[0] + op[0], Yo = (Y - 1) * s[1] - 2 * p[1] + k[1] + op[1], Zo = (Z - 1) * s[2] - 2 * p[2] + k[2] + op[2];
  size_t osz = (size_t)N * Co * Xo * Yo * Zo;
  double* acc = (double*)__builtin_malloc(osz * sizeof(double));
  for (size_t i = 0; i < osz; ++i) acc[i] = 0.0;
  for (int n = 0; n < N; ++n)
    for (int ci = 0; ci < Ci; ++ci)
      for (int xi = 0; xi < X; ++xi)
        for (int yi = 0; yi < Y; ++yi)
          for (int zi = 0; zi < Z; ++zi) {
            double v = x[(((size_t)(n * Ci + ci) * X + xi) * Y + yi) * Z + zi];
            for (int co = 0; co < Co; ++co)
              for (int a = 0; a < k[0]; ++a) {
                int xo = xi * s[0] - p[0] + a;
                if (xo < 0 || xo >= Xo) continue;
                for (int c = 0; c < k[1]; ++c) {
                  int yo = yi * s[1] - p[1] + c;
                  if (yo < 0 || yo >= Yo) continue;
                  for (int d = 0; d < k[2]; ++d) {
                    int zo = zi * s[2] - p[2] + d;
                    if (zo < 0 || zo >= Zo) continue;
                    acc[((((size_t)n * Co + co) * Xo + xo) * Yo + yo) * Zo + zo] += v * w[((((size_t)ci * Co + co) * k[0] + a) * k[1] + c) * k[2] + d];
                  }
                }
              }
          }
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (size_t i = 0; i < (size_t)Xo * Yo * Zo; ++i) {
        size_t o = ((size_t)n * Co + co) * Xo * Yo * Zo + i;
        y[o] = (float)(acc[o] + (b ? b[co] : 0.0));
      }
  __builtin_free(acc);
}
