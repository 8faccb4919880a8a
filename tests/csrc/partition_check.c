/* CPU check of the rank formulation used by wave_intro_select (csrc/engine.hip): the reference's partition loop
 * (jly_sorting.hpp:282-303, restated here as seq()) against "pair the i-th element above the pivot from the left with the
 * i-th element below it from the right while they have not crossed" (par()), on random rows with many duplicates.
 * reg(): the same arrangement for ranges of <= 64 elements besides the pivot with ONE ELEMENT PER LANE and no index lists in memory (the
 * register-resident form sized in DESIGN.md for the next round): a lane decides on its own whether its element is swapped ('>' of left-rank g
 * is swapped iff more than g '<' lie to its right, and symmetrically), the two positions of pair i meet on lane i through a scatter of lane
 * ids by rank (ds_permute on the device), and the values move with one gather and one scatter. Lanes are emulated by loops.
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
static int popc(unsigned long long x) { return __builtin_popcountll(x); }
static unsigned long long below(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
static unsigned long long above(int lane) { return lane == 63 ? 0ull : (~0ull << (lane + 1)); }
static int reg(float* d, int s, int e) {           /* e - s <= 64 */
    const float piv = d[s]; const int L = e - s;
    float v[64]; int A[64], B[64]; unsigned long long g = 0, l = 0, sw = 0;
    for (int lane = 0; lane < L; lane++) { v[lane] = d[s + 1 + lane]; if (v[lane] > piv) g |= 1ull << lane; if (v[lane] < piv) l |= 1ull << lane; }
    const int nG = popc(g);
    for (int lane = 0; lane < L; lane++) {          /* every lane on its own: rank, swap decision, scatter of its id to lane <rank> */
        if ((g >> lane) & 1) { const int r = popc(g & below(lane)); A[r] = lane; if (popc(l & above(lane)) > r) sw |= 1ull << lane; }
        if ((l >> lane) & 1) { const int r = popc(l & above(lane)); B[r] = lane; if (popc(g & below(lane)) > r) sw |= 1ull << lane; }
    }
    const int t = popc(sw & g);
    if (t != popc(sw & l)) return -1;
    float va[64], vb[64];
    for (int i = 0; i < t; i++) { if (!(A[i] < B[i])) return -1; va[i] = v[A[i]]; vb[i] = v[B[i]]; }     /* gather (ds_bpermute) */
    for (int i = 0; i < t; i++) { v[B[i]] = va[i]; v[A[i]] = vb[i]; }                                     /* scatter (ds_permute)  */
    for (int i = t; i < 64; i++) if (i < nG && i < popc(l) && A[i] < B[i]) return -1;                     /* pairs beyond t have crossed */
    const int hi0 = t > 0 ? s + 1 + B[t - 1] : e; int lo1 = hi0, big;
    if (t < nG && s + 1 + A[t] < hi0) { lo1 = s + 1 + A[t]; big = 1; } else big = t > 0 ? 1 : !(v[L - 1] < piv);
    const int sp = big ? lo1 - 1 : lo1;
    for (int lane = 0; lane < L; lane++) d[s + 1 + lane] = v[lane];
    d[s] = d[sp]; d[sp] = piv; return sp;
}
int main(int argc, char** argv) {
    long iters = argc > 1 ? atol(argv[1]) : 300000, bad = 0;
    static int idx[4096]; static float a[4096], b[4096], c[4096];
    srand(1);
    for (long it = 0; it < iters; it++) {
        int n = 2 + rand() % (it % 7 == 0 ? 1500 : (it % 7 == 1 ? 64 : 12)), vals = 1 + rand() % (it % 3 == 0 ? 3 : 1000);
        for (int i = 0; i < n; i++) a[i] = b[i] = c[i] = (float)(rand() % vals);
        int s1 = seq(a, 0, n - 1), s2 = par(b, 0, n - 1, idx);
        if (s1 != s2 || memcmp(a, b, n * sizeof(float))) bad++;
        if (n <= 65) { const int s3 = reg(c, 0, n - 1); if (s3 != s1 || memcmp(a, c, n * sizeof(float))) bad++; }
    }
    printf("bad=%ld\n", bad);
    return bad != 0;
}
