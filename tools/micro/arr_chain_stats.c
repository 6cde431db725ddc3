// arr_chain_stats.c -- how often does an augmenting-row-reduction iteration (lapjv.cpp:76-149) continue with the
// row the previous one displaced?  (uniform n=2048: 99 %; the iterations are one dependent chain.)
//   gcc -O2 -o arr_chain_stats arr_chain_stats.c && ./arr_chain_stats 2048 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define LARGE 1000000.0
int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 2048; unsigned seed = argc > 2 ? atoi(argv[2]) : 1;
    double *C = malloc(sizeof(double) * n * n); srand(seed);
    for (long k = 0; k < (long)n * n; ++k) C[k] = rand() / (double)RAND_MAX;
    int *x = malloc(4 * n), *y = malloc(4 * n), *fr = malloc(4 * n); double *v = malloc(8 * n);
    for (int i = 0; i < n; ++i) x[i] = -1;
    for (int j = 0; j < n; ++j) { double m = LARGE; int a = 0; for (int i = 0; i < n; ++i) if (C[(long)i*n+j] < m) { m = C[(long)i*n+j]; a = i; } v[j] = m; y[j] = a; }
    char *uq = malloc(n); memset(uq, 1, n);
    for (int j = n - 1; j >= 0; --j) { int i = y[j]; if (x[i] < 0) x[i] = j; else { uq[i] = 0; y[j] = -1; } }
    int nf = 0;
    for (int i = 0; i < n; ++i) { if (x[i] < 0) fr[nf++] = i; else if (uq[i]) { int j = x[i]; double m = LARGE; for (int j2 = 0; j2 < n; ++j2) { if (j2 == j) continue; double c = C[(long)i*n+j2] - v[j2]; if (c < m) m = c; } v[j] -= m; } }
    printf("n=%d free after colred %d\n", n, nf);
    for (int sweep = 0; sweep < 2 && nf > 0; ++sweep) {
        unsigned cur = 0, rr = 0; int newf = 0; long iters = 0, fwd = 0, chainmax = 0, chain = 0;
        while (cur < (unsigned)nf) {
            rr++; int fi = fr[cur++]; const double *row = C + (long)fi * n; int j1 = 0, j2 = -1; double v1 = row[0] - v[0], v2 = LARGE;
            for (int j = 1; j < n; ++j) { double c = row[j] - v[j]; if (c < v2) { if (c >= v1) { v2 = c; j2 = j; } else { v2 = v1; v1 = c; j2 = j1; j1 = j; } } }
            iters++; int i0 = y[j1]; double vn = v[j1] - (v2 - v1); int low = vn < v[j1];
            int isfwd = 0;
            if (rr < cur * (unsigned)n) { if (low) v[j1] = vn; else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
                if (i0 >= 0) { if (low) { fr[--cur] = i0; isfwd = 1; } else fr[newf++] = i0; } }
            else if (i0 >= 0) fr[newf++] = i0;
            x[fi] = j1; y[j1] = fi;
            if (isfwd) { fwd++; chain++; if (chain > chainmax) chainmax = chain; } else chain = 0;
        }
        printf(" sweep %d: iters %ld, next row is the displaced one in %ld (%.1f%%), longest chain %ld, free after %d\n", sweep, iters, fwd, 100.0*fwd/iters, chainmax, newf);
        nf = newf;
    }
    return 0;
}
