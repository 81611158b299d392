/* ORACLE (test infrastructure, not product code): plain-C restatement of the reference's integer graph transforms,
 * for bit-exact checks at sizes where the pure-Python restatement (oracle/transforms.py) is too slow.
 * Pinned: tests/test_oracle_golden.py::test_c_oracle_* checks it against the golden vectors captured from the
 * reference and against oracle/transforms.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 *   dno_dummy_augment_gc   tu_data_processing.py:186-200,213-214  (interleaved (n,v),(v,n); labels 0)
 *   dno_dummy_augment_si   subgraph_isomorphism/train.py:404-474   (blocked u->d then d->u; vocabulary ids/labels)
 *   dno_conjugate          tu_data_processing.py:223-338 (mode 0 GC, 2 LINE) / utils/graph.py:177-267 (mode 1 SI)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

void dno_dummy_augment_gc(i64 G, const i64* node_ptr, const i64* edge_ptr, const i64* src, const i64* dst,
                          const i64* node_label, const i64* edge_label, i64* o_node_ptr, i64* o_edge_ptr, i64* o_src,
                          i64* o_dst, i64* o_node_label, i64* o_edge_label, i64* o_dn, i64* o_de, i64* o_nid, i64* o_eid) {
    i64 nb = 0, eb = 0;
    o_node_ptr[0] = 0; o_edge_ptr[0] = 0;
    for (i64 g = 0; g < G; ++g) {
        const i64 n0 = node_ptr[g], n = node_ptr[g + 1] - n0, e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
        for (i64 v = 0; v < n; ++v) { o_node_label[nb + v] = node_label[n0 + v]; o_dn[nb + v] = 0; o_nid[nb + v] = v; }
        o_node_label[nb + n] = 0; o_dn[nb + n] = 1; o_nid[nb + n] = n;                     /* (:188-189) */
        for (i64 e = 0; e < m; ++e) {                                                       /* (:192) */
            o_src[eb + e] = src[e0 + e] - n0 + nb; o_dst[eb + e] = dst[e0 + e] - n0 + nb;
            o_edge_label[eb + e] = edge_label[e0 + e]; o_de[eb + e] = 0; o_eid[eb + e] = e;
        }
        for (i64 v = 0; v < n; ++v) {                                                       /* (:193) (n,v),(v,n) */
            const i64 a = eb + m + 2 * v;
            o_src[a] = nb + n; o_dst[a] = nb + v; o_src[a + 1] = nb + v; o_dst[a + 1] = nb + n;
            o_edge_label[a] = o_edge_label[a + 1] = 0; o_de[a] = o_de[a + 1] = 1;
            o_eid[a] = m + 2 * v; o_eid[a + 1] = m + 2 * v + 1;
        }
        nb += n + 1; eb += m + 2 * n;
        o_node_ptr[g + 1] = nb; o_edge_ptr[g + 1] = eb;
    }
}

void dno_dummy_augment_si(i64 G, const i64* node_ptr, const i64* edge_ptr, const i64* src, const i64* dst,
                          const i64* node_id, const i64* node_label, const i64* edge_id, const i64* edge_label,
                          const i64* in_rev, i64 max_nv, i64 max_nvl, i64 max_ne, i64 max_nel, i64* o_node_ptr,
                          i64* o_edge_ptr, i64* o_src, i64* o_dst, i64* o_nid, i64* o_nl, i64* o_eid, i64* o_el, i64* o_dn,
                          i64* o_de, i64* o_rev) {
    i64 nb = 0, eb = 0;
    o_node_ptr[0] = 0; o_edge_ptr[0] = 0;
    for (i64 g = 0; g < G; ++g) {
        const i64 n0 = node_ptr[g], n = node_ptr[g + 1] - n0, e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
        for (i64 v = 0; v < n; ++v) { o_nid[nb + v] = node_id[n0 + v]; o_nl[nb + v] = node_label[n0 + v]; o_dn[nb + v] = 0; }
        o_nid[nb + n] = max_nv; o_nl[nb + n] = max_nvl; o_dn[nb + n] = 1;                   /* (:416-423) */
        for (i64 e = 0; e < m; ++e) {
            o_src[eb + e] = src[e0 + e] - n0 + nb; o_dst[eb + e] = dst[e0 + e] - n0 + nb;
            o_eid[eb + e] = edge_id[e0 + e]; o_el[eb + e] = edge_label[e0 + e]; o_de[eb + e] = 0;
            o_rev[eb + e] = in_rev ? in_rev[e0 + e] : 0;
        }
        for (i64 u = 0; u < n; ++u) {                                                       /* (:424-431) blocked */
            const i64 a = eb + m + u, b = eb + m + n + u;
            o_src[a] = nb + u; o_dst[a] = nb + n; o_eid[a] = max_ne; o_el[a] = max_nel; o_de[a] = 1; o_rev[a] = 0;
            o_src[b] = nb + n; o_dst[b] = nb + u; o_eid[b] = max_ne + 1; o_el[b] = max_nel + 1; o_de[b] = 1; o_rev[b] = 1;
        }
        nb += n + 1; eb += m + 2 * n;
        o_node_ptr[g + 1] = nb; o_edge_ptr[g + 1] = eb;
    }
}

/* open-addressing set of (a, b, c) triples */
typedef struct { i64 a, b, c; } key3;
static int set_insert(key3* tab, unsigned char* used, i64 cap, i64 a, i64 b, i64 c) {
    uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull ^ ((uint64_t)b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full ^
                 (uint64_t)c * 0x165667B19E3779F9ull;
    i64 i = (i64)(h % (uint64_t)cap);
    while (used[i]) {
        if (tab[i].a == a && tab[i].b == b && tab[i].c == c) return 0;
        if (++i == cap) i = 0;
    }
    used[i] = 1; tab[i].a = a; tab[i].b = b; tab[i].c = c;
    return 1;
}

/* Counts pass: returns total raw 2-path count (upper bound of conj edges) */
i64 dno_conjugate_raw_count(i64 N, i64 E, const i64* src, const i64* dst) {
    i64* indeg = (i64*)calloc((size_t)(N > 0 ? N : 1), sizeof(i64));
    for (i64 e = 0; e < E; ++e) indeg[dst[e]]++;
    i64 t = 0;
    for (i64 e = 0; e < E; ++e) t += indeg[src[e]];
    free(indeg);
    return t;
}

/* mode 0 = GC (merge IS_DUMMY edges, drop (Phi,Phi), dedupe (uid,vid)); 1 = SI (vertices keyed by edge_id, dedupe
 * (uid, label(shared), vid)); 2 = LINE (no merge, no dedupe).  Outputs sized by E (rep) / raw count (edges).
 * Returns via counts[0] = #conj vertices, counts[1] = #conj edges. */
void dno_conjugate(int mode, i64 G, const i64* node_ptr, const i64* edge_ptr, const i64* src, const i64* dst,
                   const i64* node_label, const i64* edge_id, const i64* is_dummy, i64* cnode_ptr, i64* cedge_ptr,
                   i64* csrc, i64* cdst, i64* rep_edge, i64* shared_node, i64* counts) {
    i64 vb = 0, eb = 0;
    cnode_ptr[0] = 0; cedge_ptr[0] = 0;
    for (i64 g = 0; g < G; ++g) {
        const i64 n0 = node_ptr[g], n = node_ptr[g + 1] - n0, e0 = edge_ptr[g], m = edge_ptr[g + 1] - e0;
        if (m == 0) { cnode_ptr[g + 1] = vb; cedge_ptr[g + 1] = eb; continue; }
        /* conj ids + representative (first edge with that id) (:228-251 / :183-195) */
        i64* eid = (i64*)malloc(sizeof(i64) * (size_t)m);
        i64 slots = 0;
        for (i64 e = 0; e < m; ++e) { eid[e] = edge_id ? edge_id[e0 + e] : e; if (eid[e] + 1 > slots) slots = eid[e] + 1; }
        i64* id2v = (i64*)malloc(sizeof(i64) * (size_t)slots);
        for (i64 s = 0; s < slots; ++s) id2v[s] = -1;
        for (i64 e = 0; e < m; ++e) if (id2v[eid[e]] < 0) id2v[eid[e]] = e;
        /* sorted in-incidence lists */
        i64* inptr = (i64*)calloc((size_t)n + 1, sizeof(i64));
        for (i64 e = 0; e < m; ++e) inptr[dst[e0 + e] - n0 + 1]++;
        for (i64 v = 0; v < n; ++v) inptr[v + 1] += inptr[v];
        i64* inlist = (i64*)malloc(sizeof(i64) * (size_t)m);
        i64* fill = (i64*)malloc(sizeof(i64) * (size_t)(n > 0 ? n : 1));
        memcpy(fill, inptr, sizeof(i64) * (size_t)n);
        for (i64 e = 0; e < m; ++e) inlist[fill[dst[e0 + e] - n0]++] = e;
        i64 raw = 0;
        for (i64 e = 0; e < m; ++e) { const i64 s = src[e0 + e] - n0; raw += inptr[s + 1] - inptr[s]; }
        const i64 cap = 2 * raw + 16;
        key3* tab = (key3*)malloc(sizeof(key3) * (size_t)cap);
        unsigned char* used = (unsigned char*)calloc((size_t)cap, 1);
        i64* ru = (i64*)malloc(sizeof(i64) * (size_t)(raw > 0 ? raw : 1));
        i64* rv = (i64*)malloc(sizeof(i64) * (size_t)(raw > 0 ? raw : 1));
        i64* rs = (i64*)malloc(sizeof(i64) * (size_t)(raw > 0 ? raw : 1));
        i64 cnt = 0;
        for (i64 e = 0; e < m; ++e) {                                                       /* (:261-274 / :214-227) */
            const i64 s = src[e0 + e] - n0, vid = eid[e], lab = node_label[n0 + s];
            for (i64 k = inptr[s]; k < inptr[s + 1]; ++k) {
                const i64 uid = eid[inlist[k]];
                if (mode == 2 || set_insert(tab, used, cap, uid, lab, vid)) { ru[cnt] = uid; rv[cnt] = vid; rs[cnt] = s; ++cnt; }
            }
        }
        if (mode == 0) {                                                                     /* (:289-318) */
            i64 phi = -1;
            for (i64 e = 0; e < m; ++e) if (is_dummy[e0 + e]) { if (phi < 0) phi = eid[e]; else id2v[eid[e]] = -1; }
            if (phi >= 0) {
                memset(used, 0, (size_t)cap);
                set_insert(tab, used, cap, phi, 0, phi);
                i64 k2 = 0;
                for (i64 t = 0; t < cnt; ++t) {
                    i64 u = ru[t], v = rv[t];
                    if (is_dummy[e0 + u]) u = phi;      /* ids are local edge indices in GC mode */
                    if (is_dummy[e0 + v]) v = phi;
                    if (set_insert(tab, used, cap, u, 0, v)) { ru[k2] = u; rv[k2] = v; rs[k2] = rs[t]; ++k2; }
                }
                cnt = k2;
            }
        }
        /* compact vertex numbering (:333-336 / :264-267) */
        i64* remap = (i64*)malloc(sizeof(i64) * (size_t)slots);
        i64 nv = 0;
        for (i64 s = 0; s < slots; ++s) if (id2v[s] >= 0) { remap[s] = nv; rep_edge[vb + nv] = e0 + id2v[s]; ++nv; } else remap[s] = -1;
        for (i64 t = 0; t < cnt; ++t) { csrc[eb + t] = vb + remap[ru[t]]; cdst[eb + t] = vb + remap[rv[t]]; shared_node[eb + t] = n0 + rs[t]; }
        vb += nv; eb += cnt;
        cnode_ptr[g + 1] = vb; cedge_ptr[g + 1] = eb;
        free(eid); free(id2v); free(inptr); free(inlist); free(fill); free(tab); free(used); free(ru); free(rv); free(rs); free(remap);
    }
    counts[0] = vb; counts[1] = eb;
}
