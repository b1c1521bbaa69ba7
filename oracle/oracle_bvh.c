/* oracle_bvh.c -- CPU LBVH builder of the oracle (TEST INFRASTRUCTURE, see rt64_oracle.h).
 *
 * Replaces the DXR driver's BuildRaytracingAccelerationStructure, which the reference calls at
 *   ref:contrib/nv_helpers_dx12/BottomLevelASGenerator.cpp:245  (per RT64_SetMesh, ref:private/rt64_mesh.cpp:114-158)
 *   ref:contrib/nv_helpers_dx12/TopLevelASGenerator.cpp:244     (every frame, ref:private/rt64_view.cpp:412-452)
 * The driver's algorithm is not part of the reference; the published algorithm restated here is
 *   Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (LBVH).
 *
 * ---------------------------------------------------------------------------------------------------
 * Geometry spec, part 1 (shared contract with the HIP builder; results must match bit for bit)
 *   G1  leaf box      = min/max of the triangle's three positions (BLAS) or of the 8 transformed corners
 *                       of the mesh box (TLAS, g_xform_point chain).
 *   G2  scene box     = union of leaf boxes.  centre c = (bmin + bmax) * 0.5f.
 *       scale_k       = ext_k > 0 ? 1024.0f / ext_k : 0.0f         (ext = sceneMax - sceneMin)
 *       q_k           = (int)((c_k - sceneMin_k) * scale_k) clamped to [0, 1023]
 *       code          = 30-bit Morton interleave, x in bit 0, y in bit 1, z in bit 2.
 *       key           = code << 32 | leafIndex   (64 bit, unique)
 *   G3  leaves sorted ascending by key.
 *   G4  inner node i in [0, n-2] by Karras' range/split search with delta(i,j) = clz64(key_i ^ key_j),
 *       delta = -1 outside [0, n-1].  Children: split g -> left = (min(range)==g ? leaf g : inner g),
 *       right = (max(range)==g+1 ? leaf g+1 : inner g+1).  Root = inner 0.  The node's `pad` word keeps j, the other end of
 *       its leaf range [min(i, j), max(i, j)] (0 in a single-leaf tree).
 *   G5  node i stores the boxes of both children (ONode).  n == 1: node 0 = {left = leaf 0, right = none with an
 *       empty box (min=+inf, max=-inf)}.
 *   G6  refit keeps sort order and topology and recomputes boxes only.
 * ---------------------------------------------------------------------------------------------------
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "oracle_internal.h"

uint32_t oracle_morton30(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t v[3] = { x & 1023u, y & 1023u, z & 1023u };
    uint32_t code = 0;
    for (int k = 0; k < 3; k++) {
        uint32_t t = v[k];
        t = (t | (t << 16)) & 0x030000FFu;
        t = (t | (t << 8)) & 0x0300F00Fu;
        t = (t | (t << 4)) & 0x030C30C3u;
        t = (t | (t << 2)) & 0x09249249u;
        code |= t << k;
    }
    return code;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

static inline int delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(keys[i] ^ keys[j]);   /* keys are unique, so the argument is never 0 */
}

void obvh_free(OBvh *b) {
    free(b->nodes); free(b->sortedIndex); free(b->morton);
    memset(b, 0, sizeof(*b));
}

static void box_empty(float mn[3], float mx[3]) {
    for (int k = 0; k < 3; k++) { mn[k] = INFINITY; mx[k] = -INFINITY; }
}

/* Boxes of all inner nodes from the leaf boxes (in sorted leaf order). */
void obvh_fit(OBvh *b, const float *leafMin, const float *leafMax) {
    uint32_t n = b->count;
    if (n == 1) {
        ONode *nd = &b->nodes[0];
        memcpy(nd->lmin, leafMin, 12); memcpy(nd->lmax, leafMax, 12);
        box_empty(nd->rmin, nd->rmax);
        return;
    }
    /* Karras' inner nodes are not topologically ordered by index; resolve boxes by repeated passes from the
     * leaves upward using the parent links (at most depth passes, trivially correct, order independent). */
    uint32_t inner = n - 1;
    uint8_t *done = (uint8_t *)calloc(inner, 1);
    float *nmin = (float *)malloc(sizeof(float) * 3 * inner), *nmax = (float *)malloc(sizeof(float) * 3 * inner);
    uint32_t remaining = inner;
    while (remaining) {
        for (uint32_t i = 0; i < inner; i++) {
            if (done[i]) continue;
            ONode *nd = &b->nodes[i];
            int okL = (nd->left & 0x80000000u) || done[nd->left];
            int okR = (nd->right & 0x80000000u) || done[nd->right];
            if (!okL || !okR) continue;
            const float *lmn, *lmx, *rmn, *rmx;
            if (nd->left & 0x80000000u) { uint32_t s = nd->left & 0x7FFFFFFFu; lmn = leafMin + 3 * s; lmx = leafMax + 3 * s; }
            else { lmn = nmin + 3 * nd->left; lmx = nmax + 3 * nd->left; }
            if (nd->right & 0x80000000u) { uint32_t s = nd->right & 0x7FFFFFFFu; rmn = leafMin + 3 * s; rmx = leafMax + 3 * s; }
            else { rmn = nmin + 3 * nd->right; rmx = nmax + 3 * nd->right; }
            for (int k = 0; k < 3; k++) {
                nd->lmin[k] = lmn[k]; nd->lmax[k] = lmx[k]; nd->rmin[k] = rmn[k]; nd->rmax[k] = rmx[k];
                nmin[3 * i + k] = fminf(lmn[k], rmn[k]);
                nmax[3 * i + k] = fmaxf(lmx[k], rmx[k]);
            }
            done[i] = 1; remaining--;
        }
    }
    free(done); free(nmin); free(nmax);
}

/* Build over n leaf boxes given in leaf-index order.  Returns sorted boxes through outMin/outMax (malloc'd, 3n floats). */
void obvh_build(OBvh *b, uint32_t n, const float *boxMin, const float *boxMax, float **outMin, float **outMax) {
    obvh_free(b);
    b->count = n;
    box_empty(b->bmin, b->bmax);
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            b->bmin[k] = fminf(b->bmin[k], boxMin[3 * i + k]);
            b->bmax[k] = fmaxf(b->bmax[k], boxMax[3 * i + k]);
        }
    float scale[3];
    for (int k = 0; k < 3; k++) {
        float ext = b->bmax[k] - b->bmin[k];
        scale[k] = ext > 0.0f ? 1024.0f / ext : 0.0f;
    }
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            float c = (boxMin[3 * i + k] + boxMax[3 * i + k]) * 0.5f;
            float f = (c - b->bmin[k]) * scale[k];
            int qi = (int)f;
            q[k] = (uint32_t)(qi < 0 ? 0 : (qi > 1023 ? 1023 : qi));
        }
        keys[i] = ((uint64_t)oracle_morton30(q[0], q[1], q[2]) << 32) | i;
    }
    qsort(keys, n, sizeof(uint64_t), cmp_u64);

    b->sortedIndex = (uint32_t *)malloc(sizeof(uint32_t) * n);
    b->morton = (uint32_t *)malloc(sizeof(uint32_t) * n);
    float *smin = (float *)malloc(sizeof(float) * 3 * n), *smax = (float *)malloc(sizeof(float) * 3 * n);
    for (uint32_t s = 0; s < n; s++) {
        uint32_t i = (uint32_t)(keys[s] & 0xFFFFFFFFu);
        b->sortedIndex[s] = i;
        b->morton[s] = (uint32_t)(keys[s] >> 32);
        memcpy(smin + 3 * s, boxMin + 3 * i, 12);
        memcpy(smax + 3 * s, boxMax + 3 * i, 12);
    }
    uint32_t inner = n > 1 ? n - 1 : 1;
    b->nodes = (ONode *)calloc(inner, sizeof(ONode));
    if (n == 1) {
        b->nodes[0].left = 0x80000000u; b->nodes[0].right = 0xFFFFFFFFu; b->nodes[0].parent = 0xFFFFFFFFu;
    }
    else {
        b->nodes[0].parent = 0xFFFFFFFFu;
        for (int i = 0; i < (int)n - 1; i++) {
            int d = (delta(keys, (int)n, i, i + 1) - delta(keys, (int)n, i, i - 1)) >= 0 ? 1 : -1;
            int dmin = delta(keys, (int)n, i, i - d);
            int lmax = 2;
            while (delta(keys, (int)n, i, i + lmax * d) > dmin) lmax *= 2;
            int l = 0;
            for (int t = lmax / 2; t >= 1; t /= 2)
                if (delta(keys, (int)n, i, i + (l + t) * d) > dmin) l += t;
            int j = i + l * d;
            int dnode = delta(keys, (int)n, i, j);
            int s = 0;
            for (int div = 2, t = (l + div - 1) / div; ; div *= 2, t = (l + div - 1) / div) {
                if (delta(keys, (int)n, i, i + (s + t) * d) > dnode) s += t;
                if (t <= 1) break;
            }
            int g = i + s * d + (d < 0 ? -1 : 0);
            int lo = i < j ? i : j, hi = i < j ? j : i;
            ONode *nd = &b->nodes[i];
            if (lo == g) nd->left = 0x80000000u | (uint32_t)g; else { nd->left = (uint32_t)g; b->nodes[g].parent = (uint32_t)i; }
            if (hi == g + 1) nd->right = 0x80000000u | (uint32_t)(g + 1); else { nd->right = (uint32_t)(g + 1); b->nodes[g + 1].parent = (uint32_t)i; }
            nd->pad = (uint32_t)j;          /* G4: the node keeps the other end of its leaf range [min(i, j), max(i, j)] (the GPU's chunked box fit asks for it) */
        }
    }
    obvh_fit(b, smin, smax);
    free(keys);
    if (outMin) *outMin = smin; else free(smin);
    if (outMax) *outMax = smax; else free(smax);
}

/* ---- meshes (BLAS) ------------------------------------------------------------------------------------ */

OMesh *oracle_mesh_create(int flags) {
    OMesh *m = (OMesh *)calloc(1, sizeof(OMesh));
    m->flags = flags;
    return m;
}

void oracle_mesh_destroy(OMesh *m) {
    if (!m) return;
    free(m->vertices); free(m->indices); free(m->tris);
    obvh_free(&m->bvh);
    free(m);
}

const OBvh *oracle_mesh_bvh(const OMesh *m) { return &m->bvh; }
const OTri *oracle_mesh_tris(const OMesh *m) { return m->tris; }

static void tri_positions(const OMesh *m, uint32_t prim, float v[3][3]) {
    for (int k = 0; k < 3; k++) {
        uint32_t idx = m->indices[3 * prim + k];
        memcpy(v[k], m->vertices + (size_t)idx * m->vertexStride, 12);   /* position is always the first 12 bytes, ref:rt64_shader.cpp:88 */
    }
}

void oracle_mesh_set(OMesh *m, const void *vertices, int vertexCount, int vertexStride, const uint32_t *indices, int indexCount) {
    /* ref:private/rt64_mesh.cpp:30-39,76-82: a change of counts/stride discards the BLAS even if updatable. */
    int sameShape = m->vertices && m->vertexCount == vertexCount && m->vertexStride == vertexStride && m->indexCount == indexCount;
    free(m->vertices); free(m->indices);
    m->vertices = (uint8_t *)malloc((size_t)vertexCount * vertexStride);
    memcpy(m->vertices, vertices, (size_t)vertexCount * vertexStride);
    m->indices = (uint32_t *)malloc(sizeof(uint32_t) * indexCount);
    memcpy(m->indices, indices, sizeof(uint32_t) * indexCount);
    m->vertexCount = vertexCount; m->vertexStride = vertexStride; m->indexCount = indexCount;
    m->version++;
    if (!(m->flags & 0x1)) return;                                     /* RT64_MESH_RAYTRACE_ENABLED, ref:rt64_mesh.cpp:115 */

    uint32_t n = (uint32_t)indexCount / 3;
    int refit = (m->flags & 0x2) && sameShape && m->bvh.count == n;    /* RT64_MESH_RAYTRACE_UPDATABLE, ref:rt64_mesh.cpp:129,149-157 */
    float *bmin = (float *)malloc(sizeof(float) * 3 * n), *bmax = (float *)malloc(sizeof(float) * 3 * n);
    if (!refit) {
        for (uint32_t p = 0; p < n; p++) {
            float v[3][3]; tri_positions(m, p, v);
            for (int k = 0; k < 3; k++) {
                bmin[3 * p + k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
                bmax[3 * p + k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
            }
        }
        obvh_build(&m->bvh, n, bmin, bmax, NULL, NULL);
        free(m->tris);
        m->tris = (OTri *)calloc(n, sizeof(OTri));
    }
    for (uint32_t s = 0; s < n; s++) {
        uint32_t p = m->bvh.sortedIndex[s];
        float v[3][3]; tri_positions(m, p, v);
        OTri *t = &m->tris[s];
        memcpy(t->v0, v[0], 12); memcpy(t->v1, v[1], 12); memcpy(t->v2, v[2], 12);
        t->prim = p;
        if (refit)
            for (int k = 0; k < 3; k++) {
                bmin[3 * s + k] = fminf(fminf(v[0][k], v[1][k]), v[2][k]);
                bmax[3 * s + k] = fmaxf(fmaxf(v[0][k], v[1][k]), v[2][k]);
            }
    }
    if (refit) {
        obvh_fit(&m->bvh, bmin, bmax);
        for (int k = 0; k < 3; k++) { m->bvh.bmin[k] = INFINITY; m->bvh.bmax[k] = -INFINITY; }
        for (uint32_t s = 0; s < n; s++)
            for (int k = 0; k < 3; k++) {
                m->bvh.bmin[k] = fminf(m->bvh.bmin[k], bmin[3 * s + k]);
                m->bvh.bmax[k] = fmaxf(m->bvh.bmax[k], bmax[3 * s + k]);
            }
    }
    free(bmin); free(bmax);
}
