/* ORACLE -- test infrastructure only.  Never linked into or called by the product.
 *
 * Plain-C, single-threaded statement of the one-pass integer spec in oracle/onepass.py
 * (per-label count / bounding box / raw first and second coordinate moments, and
 * per-axis shared-face counts per unordered label pair), for volumes too large for the
 * numpy version to finish in seconds.  Semantics follow the reference's
 * spatial_image_analysis.py: nd.find_objects (SIA:517), nd.sum (SIA:1231),
 * nd.center_of_mass (SIA:466), cov = P.P^T/N (SIA:137-150), 6-connected contact via
 * binary_dilation (SIA:45-52) and directional face counts (SIA:947-956).
 * Pinning status: see oracle/sia_oracle.py ("parity unpinned" beyond the docstring
 * examples); tests check this file against onepass.py and sia_oracle.py.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libonepass_oracle.so oracle/onepass_c.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t key; uint64_t f[3]; } pair_slot;

static pair_slot *g_tab = NULL;
static uint64_t g_cap = 0, g_used = 0;
static pair_slot *g_sorted = NULL;
static int64_t g_nsorted = 0;

static uint64_t mix(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

static int tab_reserve(uint64_t cap) {
    pair_slot *old = g_tab; uint64_t oldcap = g_cap;
    g_tab = (pair_slot *)malloc(cap * sizeof(pair_slot));
    if (!g_tab) { g_tab = old; return -1; }
    for (uint64_t i = 0; i < cap; ++i) { g_tab[i].key = ~0ULL; g_tab[i].f[0] = g_tab[i].f[1] = g_tab[i].f[2] = 0; }
    g_cap = cap; g_used = 0;
    for (uint64_t i = 0; i < oldcap; ++i) {
        if (old[i].key == ~0ULL) continue;
        uint64_t h = mix(old[i].key) & (cap - 1);
        while (g_tab[h].key != ~0ULL) h = (h + 1) & (cap - 1);
        g_tab[h] = old[i]; ++g_used;
    }
    free(old);
    return 0;
}

static int add_face(uint32_t a, uint32_t b, int axis) {
    uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    uint64_t key = ((uint64_t)lo << 32) | hi;
    if (2 * (g_used + 1) > g_cap && tab_reserve(g_cap ? 2 * g_cap : (1u << 16))) return -1;
    uint64_t h = mix(key) & (g_cap - 1);
    while (g_tab[h].key != ~0ULL && g_tab[h].key != key) h = (h + 1) & (g_cap - 1);
    if (g_tab[h].key == ~0ULL) { g_tab[h].key = key; ++g_used; }
    g_tab[h].f[axis] += 1;
    return 0;
}

static int cmp_slot(const void *x, const void *y) {
    uint64_t a = ((const pair_slot *)x)->key, b = ((const pair_slot *)y)->key;
    return a < b ? -1 : (a > b ? 1 : 0);
}

static inline uint32_t vox(const void *vol, int itemsize, int64_t idx) {
    return itemsize == 2 ? (uint32_t)((const uint16_t *)vol)[idx] : ((const uint32_t *)vol)[idx];
}

/* vol: C-ordered [n0][n1][n2]; if !own_first_plane, plane 0 is a halo (faces along axis 0
 * with plane 1 only).  origin = global coordinate of vol[0][0][0].  bbox rows are
 * min0,min1,min2,max0+1,max1+1,max2+1 or -1 when the label is absent.
 * Returns 0, -1 (bad args / out of memory) or -2 (label above max_label). */
int oracle_extract(const void *vol, int itemsize, const int64_t *dims, const int64_t *origin,
                   int own_first_plane, uint32_t max_label,
                   uint64_t *count, int32_t *bbox, uint64_t *sum1, uint64_t *sum2,
                   int64_t *npairs) {
    if (!vol || (itemsize != 2 && itemsize != 4)) return -1;
    int64_t n0 = dims[0], n1 = dims[1], n2 = dims[2];
    uint64_t L1 = (uint64_t)max_label + 1;
    memset(count, 0, L1 * sizeof(uint64_t));
    memset(sum1, 0, L1 * 3 * sizeof(uint64_t));
    memset(sum2, 0, L1 * 6 * sizeof(uint64_t));
    for (uint64_t i = 0; i < L1 * 6; ++i) bbox[i] = -1;
    free(g_tab); g_tab = NULL; g_cap = g_used = 0;
    free(g_sorted); g_sorted = NULL; g_nsorted = 0;
    if (tab_reserve(1u << 16)) return -1;
    for (int64_t a = 0; a < n0; ++a) {
        int owned = own_first_plane || a > 0;
        for (int64_t b = 0; b < n1; ++b) {
            for (int64_t c = 0; c < n2; ++c) {
                int64_t idx = (a * n1 + b) * n2 + c;
                uint32_t v = vox(vol, itemsize, idx);
                if (v > max_label) return -2;
                if (!owned) continue;
                uint64_t p[3] = { (uint64_t)(a + origin[0]), (uint64_t)(b + origin[1]), (uint64_t)(c + origin[2]) };
                count[v] += 1;
                int32_t *bb = bbox + (uint64_t)v * 6;
                for (int d = 0; d < 3; ++d) {
                    sum1[(uint64_t)v * 3 + d] += p[d];
                    int32_t q = (int32_t)p[d];
                    if (bb[d] < 0 || q < bb[d]) bb[d] = q;
                    if (q + 1 > bb[3 + d]) bb[3 + d] = q + 1;
                }
                uint64_t *s2 = sum2 + (uint64_t)v * 6;
                s2[0] += p[0] * p[0]; s2[1] += p[0] * p[1]; s2[2] += p[0] * p[2];
                s2[3] += p[1] * p[1]; s2[4] += p[1] * p[2]; s2[5] += p[2] * p[2];
                if (a > 0) { uint32_t u = vox(vol, itemsize, idx - n1 * n2); if (u > max_label) return -2; if (u != v && add_face(u, v, 0)) return -1; }
                if (b > 0) { uint32_t u = vox(vol, itemsize, idx - n2); if (u != v && add_face(u, v, 1)) return -1; }
                if (c > 0) { uint32_t u = vox(vol, itemsize, idx - 1); if (u != v && add_face(u, v, 2)) return -1; }
            }
        }
    }
    g_sorted = (pair_slot *)malloc((g_used ? g_used : 1) * sizeof(pair_slot));
    if (!g_sorted) return -1;
    for (uint64_t i = 0; i < g_cap; ++i) if (g_tab[i].key != ~0ULL) g_sorted[g_nsorted++] = g_tab[i];
    qsort(g_sorted, (size_t)g_nsorted, sizeof(pair_slot), cmp_slot);
    *npairs = g_nsorted;
    return 0;
}

/* Pairs of the last oracle_extract call, sorted by (lo, hi); faces is [npairs][3]. */
int oracle_pairs_get(uint32_t *lo, uint32_t *hi, uint64_t *faces) {
    for (int64_t i = 0; i < g_nsorted; ++i) {
        lo[i] = (uint32_t)(g_sorted[i].key >> 32);
        hi[i] = (uint32_t)(g_sorted[i].key & 0xffffffffu);
        faces[3 * i + 0] = g_sorted[i].f[0]; faces[3 * i + 1] = g_sorted[i].f[1]; faces[3 * i + 2] = g_sorted[i].f[2];
    }
    return 0;
}
