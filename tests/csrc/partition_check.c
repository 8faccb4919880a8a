/* CPU check of the rank formulation used by wave_intro_select (csrc/engine.hip): the reference's partition loop
 * (jly_sorting.hpp:282-303, restated here as seq()) against "pair the i-th element above the pivot from the left with the
 * i-th element below it from the right while they have not crossed" (par()), on random rows with many duplicates.
 * Test infrastructure only. Prints "bad=<n>"; exit status 0 iff n == 0. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static int seq(float* d, int s, int e) {
    float piv = d[s], t; int lo = s + 1, hi = e;
    for (;;) { while (lo < hi && piv >= d[lo]) ++lo; while (lo < hi && piv <= d[hi]) --hi; if (lo >= hi) break; t = d[lo]; d[lo] = d[hi]; d[hi] = t; }
    int sp = d[lo] < piv ? lo : lo - 1; d[s] = d[sp]; d[sp] = piv; return sp;
}
static int par(float* d, int s, int e, int* idx) {
    float piv = d[s]; int L = e - s, nG = 0, nLt = 0;
    for (int p = s + 1; p <= e; p++) if (d[p] > piv) idx[nG++] = p;                 /* ranks from the left  */
    for (int p = e; p > s; p--) if (d[p] < piv) idx[L - 1 - (nLt++)] = p;           /* ranks from the right */
    int t = 0, m = nG < nLt ? nG : nLt;
    for (int i = 0; i < m; i++) { int a = idx[i], b = idx[L - 1 - i]; if (a < b) { float v = d[a]; d[a] = d[b]; d[b] = v; t++; } }
    int hi0 = t > 0 ? idx[L - 1 - (t - 1)] : e, lo1 = hi0, big;
    if (t < nG && idx[t] < hi0) { lo1 = idx[t]; big = 1; } else big = t > 0 ? 1 : !(d[e] < piv);
    int sp = big ? lo1 - 1 : lo1; d[s] = d[sp]; d[sp] = piv; return sp;
}
int main(int argc, char** argv) {
    long iters = argc > 1 ? atol(argv[1]) : 300000, bad = 0;
    static int idx[4096]; static float a[4096], b[4096];
    srand(1);
    for (long it = 0; it < iters; it++) {
        int n = 2 + rand() % (it % 7 == 0 ? 1500 : 12), vals = 1 + rand() % (it % 3 == 0 ? 3 : 1000);
        for (int i = 0; i < n; i++) a[i] = b[i] = (float)(rand() % vals);
        int s1 = seq(a, 0, n - 1), s2 = par(b, 0, n - 1, idx);
        if (s1 != s2 || memcmp(a, b, n * sizeof(float))) bad++;
    }
    printf("bad=%ld\n", bad);
    return bad != 0;
}
