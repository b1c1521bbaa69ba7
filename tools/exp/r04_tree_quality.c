/* r04_tree_quality.c -- EXPERIMENT (round 4, not shipped): how many node + triangle visits do the shadow rays that graze the stress scene's sphere make
 * under different BVHs over the same triangles?  Links the oracle's builder and uses its traversal rules (R1-R5, oracle/oracle_trace.c) on a single BLAS.
 *
 *   python: dump positions of make_sample_scene(subdiv=S) sphere to /tmp/tq/sphereS.f32 ([n][9] float32)
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma -Ioracle tools/exp/r04_tree_quality.c oracle/oracle_bvh.c oracle/oracle_util.c -lm -o /tmp/tq/tq
 *   /tmp/tq/tq /tmp/tq/sphere7.f32
 *
 * Trees: (a) the shipped LBVH (30-bit Morton keys), (b) 63-bit keys (21 bits per axis), (c) a binned-SAH top-down tree (quality reference, not a candidate
 * for the GPU), (d) LBVH + bottom-up treelet re-linking candidates as they get written.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

typedef struct { float o[3], d[3], inv[3], oi[3]; } RaySpace;
static void ray_space(const float o[3], const float d[3], RaySpace *r) {
    for (int k = 0; k < 3; k++) {
        r->o[k] = o[k]; r->d[k] = d[k];
        float ds = fabsf(d[k]) < 1e-20f ? copysignf(1e-20f, d[k]) : d[k];
        r->inv[k] = 1.0f / ds; r->oi[k] = -(o[k] * r->inv[k]);
    }
}
static inline int box_hit(const RaySpace *r, const float lo[3], const float hi[3], float tmin, float tmax, float *tnear) {
    float ax = fmaf(lo[0], r->inv[0], r->oi[0]), bx = fmaf(hi[0], r->inv[0], r->oi[0]);
    float ay = fmaf(lo[1], r->inv[1], r->oi[1]), by = fmaf(hi[1], r->inv[1], r->oi[1]);
    float az = fmaf(lo[2], r->inv[2], r->oi[2]), bz = fmaf(hi[2], r->inv[2], r->oi[2]);
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax)) * 1.0000004f;
    *tnear = tn; return tn <= tf;
}
static inline int tri_hit(const RaySpace *r, const float *tri, float tmin, float tmax, float *t) {
    float e1[3], e2[3], p[3], q[3], tv[3];
    for (int k = 0; k < 3; k++) { e1[k] = tri[3 + k] - tri[k]; e2[k] = tri[6 + k] - tri[k]; }
    g_cross3(r->d, e2, p);
    float det = g_dot3(e1, p);
    if (det == 0.0f || det != det) return 0;
    float inv = 1.0f / det;
    for (int k = 0; k < 3; k++) tv[k] = r->o[k] - tri[k];
    float uu = g_dot3(tv, p) * inv; g_cross3(tv, e1, q);
    float vv = g_dot3(r->d, q) * inv, tt = g_dot3(e2, q) * inv;
    if (!(uu >= 0.0f) || !(vv >= 0.0f) || !(uu + vv <= 1.0f) || !(tt > tmin) || !(tt < tmax)) return 0;
    *t = tt; return 1;
}

typedef struct { ONode *nodes; uint32_t *leafTri; uint32_t n; } Tree;     /* leafTri[slot] = triangle number */

/* first-hit (shadow) or closest-hit walk; returns visits (nodes + tris) */
static uint32_t walk(const Tree *T, const float *tris, const float o[3], const float d[3], float tmin, float tmax, int anyHit, int *hit, uint32_t *nodesOut) {
    RaySpace r; ray_space(o, d, &r);
    uint32_t stack[256]; int sp = 0; uint32_t cur = 0, nodes = 0, nt = 0; *hit = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            if (cur != 0xFFFFFFFFu) {
                float t; nt++;
                if (tri_hit(&r, tris + 9 * (size_t)T->leafTri[cur & 0x7FFFFFFFu], tmin, tmax, &t)) { *hit = 1; if (anyHit) break; tmax = t; }
            }
            if (sp == 0) break;
            cur = stack[--sp]; continue;
        }
        const ONode *nd = &T->nodes[cur]; nodes++;
        float tl, tr;
        int hl = box_hit(&r, nd->lmin, nd->lmax, tmin, tmax, &tl), hr = box_hit(&r, nd->rmin, nd->rmax, tmin, tmax, &tr);
        if (hl && hr) { if (tr < tl) { stack[sp++] = nd->left; cur = nd->right; } else { stack[sp++] = nd->right; cur = nd->left; } }
        else if (hl) cur = nd->left;
        else if (hr) cur = nd->right;
        else { if (sp == 0) break; cur = stack[--sp]; }
    }
    if (nodesOut) *nodesOut = nodes;
    return nodes + nt;
}

/* ---- (c) binned SAH, top-down ---- */
typedef struct { float mn[3], mx[3]; } Box;
static void box_init(Box *b) { for (int k = 0; k < 3; k++) { b->mn[k] = INFINITY; b->mx[k] = -INFINITY; } }
static void box_add(Box *b, const float *mn, const float *mx) { for (int k = 0; k < 3; k++) { b->mn[k] = fminf(b->mn[k], mn[k]); b->mx[k] = fmaxf(b->mx[k], mx[k]); } }
static float box_area(const Box *b) { float x = b->mx[0] - b->mn[0], y = b->mx[1] - b->mn[1], z = b->mx[2] - b->mn[2]; return x * y + y * z + z * x; }
static const float *gMin, *gMax; static uint32_t *gOrder; static ONode *gNodes; static uint32_t gNext;
static uint32_t sah_build(uint32_t lo, uint32_t hi, Box *out) {          /* returns child reference; fills *out with the subtree box */
    Box b; box_init(&b);
    for (uint32_t i = lo; i < hi; i++) box_add(&b, gMin + 3 * gOrder[i], gMax + 3 * gOrder[i]);
    *out = b;
    if (hi - lo == 1) return 0x80000000u | lo;
    Box cb; box_init(&cb);
    for (uint32_t i = lo; i < hi; i++) { float c[3]; for (int k = 0; k < 3; k++) c[k] = 0.5f * (gMin[3 * gOrder[i] + k] + gMax[3 * gOrder[i] + k]); box_add(&cb, c, c); }
    enum { BINS = 16 };
    float best = INFINITY; int bestAxis = -1, bestBin = 0;
    for (int ax = 0; ax < 3; ax++) {
        float ext = cb.mx[ax] - cb.mn[ax]; if (!(ext > 0.0f)) continue;
        Box bb[BINS]; uint32_t cnt[BINS] = { 0 }; for (int k = 0; k < BINS; k++) box_init(&bb[k]);
        for (uint32_t i = lo; i < hi; i++) {
            float c = 0.5f * (gMin[3 * gOrder[i] + ax] + gMax[3 * gOrder[i] + ax]); int k = (int)((c - cb.mn[ax]) / ext * BINS); if (k >= BINS) k = BINS - 1;
            cnt[k]++; box_add(&bb[k], gMin + 3 * gOrder[i], gMax + 3 * gOrder[i]);
        }
        Box L[BINS], R[BINS]; uint32_t cl[BINS], cr[BINS]; Box acc; box_init(&acc); uint32_t c = 0;
        for (int k = 0; k < BINS; k++) { if (cnt[k]) box_add(&acc, bb[k].mn, bb[k].mx); c += cnt[k]; L[k] = acc; cl[k] = c; }
        box_init(&acc); c = 0;
        for (int k = BINS - 1; k >= 0; k--) { if (cnt[k]) box_add(&acc, bb[k].mn, bb[k].mx); c += cnt[k]; R[k] = acc; cr[k] = c; }
        for (int k = 0; k + 1 < BINS; k++) { if (!cl[k] || !cr[k + 1]) continue; float cost = box_area(&L[k]) * cl[k] + box_area(&R[k + 1]) * cr[k + 1]; if (cost < best) { best = cost; bestAxis = ax; bestBin = k; } }
    }
    uint32_t mid;
    if (bestAxis < 0) mid = (lo + hi) / 2;
    else {
        float ext = cb.mx[bestAxis] - cb.mn[bestAxis]; uint32_t i = lo, j = hi;
        while (i < j) {
            float c = 0.5f * (gMin[3 * gOrder[i] + bestAxis] + gMax[3 * gOrder[i] + bestAxis]); int k = (int)((c - cb.mn[bestAxis]) / ext * BINS); if (k >= BINS) k = BINS - 1;
            if (k <= bestBin) i++; else { j--; uint32_t t = gOrder[i]; gOrder[i] = gOrder[j]; gOrder[j] = t; }
        }
        mid = i; if (mid == lo || mid == hi) mid = (lo + hi) / 2;
    }
    uint32_t me = gNext++; Box bl, br;
    uint32_t l = sah_build(lo, mid, &bl), r = sah_build(mid, hi, &br);
    ONode *nd = &gNodes[me]; nd->left = l; nd->right = r;
    memcpy(nd->lmin, bl.mn, 12); memcpy(nd->lmax, bl.mx, 12); memcpy(nd->rmin, br.mn, 12); memcpy(nd->rmax, br.mx, 12);
    return me;
}

/* ---- (b) LBVH with wider keys: K bits per axis (K <= 21), key = morton(3K) : index ---- */
typedef struct { uint64_t hi, lo; } Key128;
static int cmp128(const void *a, const void *b) { const Key128 *x = a, *y = b; if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1; return (x->lo > y->lo) - (x->lo < y->lo); }
static Key128 *gKeys; static int gN;
static inline int delta128(int i, int j) {
    if (j < 0 || j >= gN) return -1;
    uint64_t h = gKeys[i].hi ^ gKeys[j].hi; if (h) return __builtin_clzll(h);
    return 64 + __builtin_clzll(gKeys[i].lo ^ gKeys[j].lo);
}
static uint64_t morton3(uint32_t x, uint32_t y, uint32_t z, int bits) { uint64_t c = 0; for (int b = 0; b < bits; b++) c |= ((uint64_t)((x >> b) & 1) << (3 * b)) | ((uint64_t)((y >> b) & 1) << (3 * b + 1)) | ((uint64_t)((z >> b) & 1) << (3 * b + 2)); return c; }
static void fit_tree(Tree *T, const float *bmin, const float *bmax);
static void lbvh_bits(Tree *T, uint32_t n, const float *bmin, const float *bmax, int bits) {
    float smn[3] = { INFINITY, INFINITY, INFINITY }, smx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { smn[k] = fminf(smn[k], bmin[3 * i + k]); smx[k] = fmaxf(smx[k], bmax[3 * i + k]); }
    gKeys = malloc(sizeof(Key128) * n); gN = (int)n;
    const double cells = (double)(1u << bits);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) { double ext = (double)smx[k] - smn[k]; double f = ext > 0 ? ((0.5 * ((double)bmin[3 * i + k] + bmax[3 * i + k])) - smn[k]) / ext * cells : 0.0; long v = (long)f; q[k] = (uint32_t)(v < 0 ? 0 : (v > (long)cells - 1 ? (long)cells - 1 : v)); }
        gKeys[i].hi = morton3(q[0], q[1], q[2], bits); gKeys[i].lo = i;
    }
    qsort(gKeys, n, sizeof(Key128), cmp128);
    T->n = n; T->nodes = calloc(n - 1, sizeof(ONode)); T->leafTri = malloc(4 * n);
    for (uint32_t s = 0; s < n; s++) T->leafTri[s] = (uint32_t)gKeys[s].lo;
    for (int i = 0; i < (int)n - 1; i++) {
        int d = (delta128(i, i + 1) - delta128(i, i - 1)) >= 0 ? 1 : -1, dmin = delta128(i, i - d), lmax = 2;
        while (delta128(i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0; for (int t = lmax / 2; t >= 1; t /= 2) if (delta128(i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d, dnode = delta128(i, j), s = 0;
        for (int div = 2, t = (l + div - 1) / div;; div *= 2, t = (l + div - 1) / div) { if (delta128(i, i + (s + t) * d) > dnode) s += t; if (t <= 1) break; }
        int g = i + s * d + (d < 0 ? -1 : 0), lo = i < j ? i : j, hi = i < j ? j : i;
        ONode *nd = &T->nodes[i];
        nd->left = lo == g ? 0x80000000u | (uint32_t)g : (uint32_t)g; nd->right = hi == g + 1 ? 0x80000000u | (uint32_t)(g + 1) : (uint32_t)(g + 1);
    }
    free(gKeys);
    fit_tree(T, bmin, bmax);
}
/* boxes bottom-up by recursion from the root (children indices are arbitrary, so recurse) */
static void fit_rec(Tree *T, uint32_t node, const float *bmin, const float *bmax, Box *out) {
    ONode *nd = &T->nodes[node]; Box bl, br;
    if (nd->left & 0x80000000u) { uint32_t t = T->leafTri[nd->left & 0x7FFFFFFFu]; memcpy(bl.mn, bmin + 3 * t, 12); memcpy(bl.mx, bmax + 3 * t, 12); } else fit_rec(T, nd->left, bmin, bmax, &bl);
    if (nd->right & 0x80000000u) { uint32_t t = T->leafTri[nd->right & 0x7FFFFFFFu]; memcpy(br.mn, bmin + 3 * t, 12); memcpy(br.mx, bmax + 3 * t, 12); } else fit_rec(T, nd->right, bmin, bmax, &br);
    memcpy(nd->lmin, bl.mn, 12); memcpy(nd->lmax, bl.mx, 12); memcpy(nd->rmin, br.mn, 12); memcpy(nd->rmax, br.mx, 12);
    box_init(out); box_add(out, bl.mn, bl.mx); box_add(out, br.mn, br.mx);
}
static void fit_tree(Tree *T, const float *bmin, const float *bmax) { Box b; fit_rec(T, 0, bmin, bmax, &b); }

static double sah_cost(const Tree *T, uint32_t node, const Box *self, double rootArea) {          /* sum over inner nodes of area / root area (traversal term) */
    const ONode *nd = &T->nodes[node]; double c = box_area(self) / rootArea;
    Box bl, br; memcpy(bl.mn, nd->lmin, 12); memcpy(bl.mx, nd->lmax, 12); memcpy(br.mn, nd->rmin, 12); memcpy(br.mx, nd->rmax, 12);
    if (!(nd->left & 0x80000000u)) c += sah_cost(T, nd->left, &bl, rootArea); else c += box_area(&bl) / rootArea;
    if (!(nd->right & 0x80000000u)) c += sah_cost(T, nd->right, &br, rootArea); else c += box_area(&br) / rootArea;
    return c;
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return (x > y) - (x < y); }
static void evaluate(const char *name, const Tree *T, const float *tris) {
    /* shadow rays of floor points under the sample's light (15000, 30000, 15000), grid over the floor [-15, 10] x [-10, 10] at y = 0; first hit ends the walk */
    const int GX = 1500, GZ = 1200; const float L[3] = { 15000.0f, 30000.0f, 15000.0f };
    uint32_t *v = malloc(sizeof(uint32_t) * GX * GZ); double sum = 0; uint32_t mx = 0; double nodeSum = 0;
#pragma omp parallel for reduction(+ : sum, nodeSum) reduction(max : mx) schedule(dynamic, 8)
    for (int iz = 0; iz < GZ; iz++) for (int ix = 0; ix < GX; ix++) {
        float o[3] = { -15.0f + 25.0f * (ix + 0.5f) / GX, 0.0f, -10.0f + 20.0f * (iz + 0.5f) / GZ }, d[3] = { L[0] - o[0], L[1] - o[1], L[2] - o[2] };
        float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); for (int k = 0; k < 3; k++) d[k] /= len;
        int hit; uint32_t nn; uint32_t c = walk(T, tris, o, d, 0.1f, len, 1, &hit, &nn);
        v[iz * GX + ix] = c; sum += c; nodeSum += nn; if (c > mx) mx = c;
    }
    qsort(v, (size_t)GX * GZ, 4, cmp_u32);
    size_t N = (size_t)GX * GZ;
    /* primary rays of the sample camera (0, 2, 10) looking -z, fov 45 deg, 1920 x 1080: closest hit */
    double psum = 0; uint32_t pmx = 0; const int W = 960, H = 540; const float th = tanf(0.5f * 0.785398163f), asp = 16.0f / 9.0f;
#pragma omp parallel for reduction(+ : psum) reduction(max : pmx) schedule(dynamic, 8)
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        float ndx = ((x + 0.5f) / W) * 2.0f - 1.0f, ndy = ((y + 0.5f) / H) * 2.0f - 1.0f;
        float o[3] = { 0.0f, 2.0f, 10.0f }, d[3] = { ndx * asp * th, -ndy * th, -1.0f };
        int hit; uint32_t c = walk(T, tris, o, d, 0.1f, 100000.0f, 0, &hit, NULL); psum += c; if (c > pmx) pmx = c;
    }
    Box root; box_init(&root); box_add(&root, T->nodes[0].lmin, T->nodes[0].lmax); box_add(&root, T->nodes[0].rmin, T->nodes[0].rmax);
    printf("%-28s shadow: mean %.2f (nodes %.2f) p99 %u p99.9 %u p99.99 %u max %u | primary: mean %.2f max %u | SAH %.1f\n", name, sum / N, nodeSum / N, v[(size_t)(N * 0.99)], v[(size_t)(N * 0.999)],
           v[(size_t)(N * 0.9999)], mx, psum / (W * H), pmx, sah_cost(T, 0, &root, box_area(&root)));
    free(v);
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 3;
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    uint32_t n = (uint32_t)(bytes / 36); float *tris = malloc(bytes); if (fread(tris, 1, bytes, f) != (size_t)bytes) return 4; fclose(f);
    float *bmin = malloc(12 * (size_t)n), *bmax = malloc(12 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
        bmin[3 * i + k] = fminf(fminf(tris[9 * (size_t)i + k], tris[9 * (size_t)i + 3 + k]), tris[9 * (size_t)i + 6 + k]);
        bmax[3 * i + k] = fmaxf(fmaxf(tris[9 * (size_t)i + k], tris[9 * (size_t)i + 3 + k]), tris[9 * (size_t)i + 6 + k]);
    }
    printf("%u triangles\n", n);
    {   /* (a) the shipped builder */
        OBvh b; memset(&b, 0, sizeof(b)); obvh_build(&b, n, bmin, bmax, NULL, NULL);
        Tree T = { b.nodes, b.sortedIndex, n }; evaluate("LBVH 30-bit (shipped)", &T, tris);
    }
    for (int bits = 10; bits <= 21; bits += (bits == 10 ? 4 : 7)) { Tree T; lbvh_bits(&T, n, bmin, bmax, bits); char nm[64]; snprintf(nm, 64, "LBVH %d bits/axis", bits); evaluate(nm, &T, tris); free(T.nodes); free(T.leafTri); }
    {   /* (c) binned SAH */
        gMin = bmin; gMax = bmax; gOrder = malloc(4 * (size_t)n); for (uint32_t i = 0; i < n; i++) gOrder[i] = i;
        gNodes = calloc(n - 1, sizeof(ONode)); gNext = 0; Box rb; sah_build(0, n, &rb);
        Tree T = { gNodes, gOrder, n }; evaluate("binned SAH top-down", &T, tris);
    }
    return 0;
}
