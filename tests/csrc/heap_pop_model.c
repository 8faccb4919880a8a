/* Lane-level model of heap_pop_wave (csrc/engine_dev.hpp; test infrastructure, CPU only): the path of the hole comes from sibling
 * comparisons alone (one ballot resolves six levels), the displaced last element's landing slot from one more ballot, and the stores
 * go out together. The queue array is compared with what the sequential routine leaves (libstdc++'s pop_heap order, as heap_pop
 * restates it; reference jly_goicp.h:75-87 through std::priority_queue) on random queues with many equal keys, each pop followed by
 * pushes of the popped node's children as the search does.   usage: heap_pop_model [trials]; prints bad=N */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float lb, w; int tag; } Node;
static int key_lower(float plb, float pw, float qlb, float qw) { return plb != qlb ? plb > qlb : pw < qw; }
static void up_seq(Node* h, int hole, Node v) {
    while (hole > 0) { const int par = (hole - 1) / 2; if (!key_lower(h[par].lb, h[par].w, v.lb, v.w)) break; h[hole] = h[par]; hole = par; }
    h[hole] = v;
}
static void pop_seq(Node* h, int* n) {
    if (*n > 1) {
        const int len = *n - 1; const Node v = h[len]; h[len] = h[0];
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) { child = 2 * (child + 1); if (key_lower(h[child].lb, h[child].w, h[child - 1].lb, h[child - 1].w)) child--; h[hole] = h[child]; hole = child; }
        if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); h[hole] = h[child - 1]; hole = child - 1; }
        up_seq(h, hole, v);
    }
    (*n)--;
}
static void pop_wave(Node* h, int* n) {
    if (*n > 1) {
        const int len = *n - 1; const Node v = h[len], top = h[0];
        const int half = (len - 1) / 2;
        int hole = 0, lvl = 0, from[64], to[64];
        while (hole < half) {
            unsigned long long choice = 0;
            for (int lane = 0; lane < 63; lane++) {
                const int sd = 31 - __builtin_clz(lane + 1), sj = lane + 1 - (1 << sd);
                const int nidx = ((hole + 1) << sd) - 1 + sj;
                if (nidx < half && !key_lower(h[2 * nidx + 2].lb, h[2 * nidx + 2].w, h[2 * nidx + 1].lb, h[2 * nidx + 1].w)) choice |= 1ull << lane;
            }
            int rel = 0;
            for (int s = 0; s < 6; s++) {
                if (hole >= half) break;
                const int bit = (int)((choice >> rel) & 1ull), child = 2 * hole + 1 + bit;
                from[lvl] = child; to[lvl] = hole; hole = child; rel = 2 * rel + 1 + bit; lvl++;
            }
        }
        if ((len & 1) == 0 && hole == (len - 2) / 2) { const int child = 2 * hole + 1; from[lvl] = child; to[lvl] = hole; hole = child; lvl++; }
        Node mine[64]; unsigned long long rises = 0;
        for (int lane = 0; lane < lvl; lane++) { mine[lane] = h[from[lane]]; if (key_lower(mine[lane].lb, mine[lane].w, v.lb, v.w)) rises |= 1ull << lane; }
        const unsigned long long stays = ~rises & ((1ull << lvl) - 1ull);
        const int m = stays ? 64 - __builtin_clzll(stays) : 0;
        const int land = m == lvl ? hole : to[m];
        for (int lane = 0; lane < m; lane++) h[to[lane]] = mine[lane];
        h[land] = v; h[len] = top;
    }
    (*n)--;
}
int main(int argc, char** argv) {
    const long trials = argc > 1 ? atol(argv[1]) : 200000;
    srand(11); long bad = 0;
    static Node A[1 << 14], B[1 << 14];
    for (long t = 0; t < trials; t++) {
        const int n0 = 1 + rand() % (t % 3 == 0 ? 40 : 3000); int na = 0;
        const int levels = 2 + rand() % 6;                          /* few distinct keys: many ties */
        for (int i = 0; i < n0; i++) { Node v; v.lb = (float)(rand() % (levels * 5)); v.w = (float)(1 << (rand() % 3)); v.tag = i; up_seq(A, na++, v); }
        memcpy(B, A, sizeof(Node) * na); int nb = na;
        /* a pop, then the children of the popped node (as the search does), twice */
        for (int rep = 0; rep < 2 && na > 0; rep++) {
            const float base = A[0].lb;
            pop_seq(A, &na); pop_wave(B, &nb);
            Node v[8]; const unsigned mask = rand() & 0xff;
            for (int j = 0; j < 8; j++) { v[j].lb = base + (float)(rand() % (levels * 3)) - (rand() % 4 == 0 ? 3.f : 0.f); v[j].w = (float)(1 << (rand() % 3)); v[j].tag = 100000 + 8 * rep + j; }
            for (int j = 0; j < 8; j++) if (mask >> j & 1u) up_seq(A, na++, v[j]);
            for (int j = 0; j < 8; j++) if (mask >> j & 1u) up_seq(B, nb++, v[j]);
            if (na != nb || memcmp(A, B, sizeof(Node) * na)) { bad++; break; }
        }
    }
    printf("bad=%ld\n", bad);
    return bad != 0;
}
