"""Synthetic TU-shaped / SI-shaped batches of SURVEY.md 8(d) (numpy.random.default_rng(seed); no datasets on the
GPU box).  Returned as plain numpy arrays in the batched layout (node_ptr, edge_ptr, src, dst, labels)."""
import numpy as np


def si_uniform_batch(seed, graphs, nodes_per_graph, edges_per_graph, num_real_types):
    """configs 3 and 5: every graph has `nodes_per_graph` real nodes and `edges_per_graph` real directed edges drawn
    uniformly (multi-edges allowed), real edge type ~ U{0..num_real_types-1}."""
    rng = np.random.default_rng(seed)
    G, n, m = int(graphs), int(nodes_per_graph), int(edges_per_graph)
    base = np.repeat(np.arange(G, dtype=np.int64) * n, m)
    src = rng.integers(0, n, size=G * m, dtype=np.int64) + base
    dst = rng.integers(0, n, size=G * m, dtype=np.int64) + base
    etype = rng.integers(0, num_real_types, size=G * m, dtype=np.int64)
    return dict(node_ptr=np.arange(G + 1, dtype=np.int64) * n, edge_ptr=np.arange(G + 1, dtype=np.int64) * m,
                src=src, dst=dst, edge_label=etype,
                node_id=np.tile(np.arange(n, dtype=np.int64), G), node_label=rng.integers(0, 4, size=G * n, dtype=np.int64),
                edge_id=np.tile(np.arange(m, dtype=np.int64), G))


def config5(seed=5, graphs=32768):
    """32 768 graphs x (30 real + 1 dummy) nodes, 62 real + 60 dummy edges, 14 real + 2 dummy relation types
    => N = 1 015 808, E = 3 997 696, R = 16 after the SI dummy augmentation."""
    b = si_uniform_batch(seed, graphs, 30, 62, 14)
    b.update(max_nv=30, max_nvl=4, max_ne=62, max_nel=14, num_rels=16)
    return b


def config3(seed=3, graphs=512):
    """512 graphs x (49 + 1) nodes, 102 real + 98 dummy edges, 6 + 2 types => N = 25 600, E = 102 400, R = 8."""
    b = si_uniform_batch(seed, graphs, 49, 102, 6)
    b.update(max_nv=49, max_nvl=4, max_ne=102, max_nel=6, num_rels=8)
    return b


def tu_shaped_batch(seed, graphs, n_sampler, und_edges_per_node, num_node_labels, num_edge_labels=0):
    """configs 1, 2, 4: TU-shaped graphs; each undirected edge stored as two directed edges, no self loops."""
    rng = np.random.default_rng(seed)
    node_ptr, edge_ptr, src, dst, el = [0], [0], [], [], []
    for _ in range(graphs):
        n = int(n_sampler(rng))
        k = max(int(round(und_edges_per_node * n)), 1) if n > 1 else 0
        u = rng.integers(0, max(n, 1), size=k)
        v = (u + 1 + rng.integers(0, max(n - 1, 1), size=k)) % max(n, 1)
        base = node_ptr[-1]
        src.extend((base + np.concatenate([u, v])).tolist())
        dst.extend((base + np.concatenate([v, u])).tolist())
        if num_edge_labels:
            lab = rng.integers(1, num_edge_labels + 1, size=k)
            el.extend(np.concatenate([lab, lab]).tolist())
        else:
            el.extend([1] * (2 * k))
        node_ptr.append(base + n)
        edge_ptr.append(len(src))
    N = node_ptr[-1]
    return dict(node_ptr=np.array(node_ptr, dtype=np.int64), edge_ptr=np.array(edge_ptr, dtype=np.int64),
                src=np.array(src, dtype=np.int64), dst=np.array(dst, dtype=np.int64),
                node_label=rng.integers(1, num_node_labels + 1, size=N).astype(np.int64),
                edge_label=np.array(el, dtype=np.int64))


def config1(seed=1):
    """32 MUTAG-shaped graphs."""
    return tu_shaped_batch(seed, 32, lambda r: np.clip(round(r.normal(17.9, 4.6)), 10, 28), 1.1, 7, 4)


def config2(seed=2, graphs=512):
    """512 PROTEINS-shaped graphs (mean ~39 nodes, ~1.86 undirected edges per node)."""
    def n(r):
        return int(np.clip(r.lognormal(3.4, 0.7) * 39.0 / 38.3, 4, 620))
    return tu_shaped_batch(seed, graphs, n, 1.86, 3, 0)


def config4(seed=4, graphs=512):
    """512 NCI1-shaped graphs per GPU."""
    return tu_shaped_batch(seed, graphs, lambda r: np.clip(round(r.normal(29.9, 13.6)), 3, 111), 1.08, 37, 0)


def proteins_si(seed=2, graphs=16384, num_real_types=14):
    """PROTEINS-shaped graphs (config 2's sizes: mean ~39 nodes, up to 620; ~1.86 undirected edges per node) as an SI-style batch
    for the RGIN layer: real edge type ~ U{0..num_real_types-1}, +2 dummy relations after the SI dummy augmentation (R = 16).
    What bench.py --workload proteins times: graphs over 32 nodes, where the closing launch cannot absorb the fold."""
    b = config2(seed, graphs)
    rng = np.random.default_rng(seed + 1000)
    E, N = int(b["src"].shape[0]), int(b["node_ptr"][-1])
    sizes, esizes = np.diff(b["node_ptr"]), np.diff(b["edge_ptr"])
    b.update(edge_label=rng.integers(0, num_real_types, size=E, dtype=np.int64), node_label=rng.integers(0, 4, size=N, dtype=np.int64),
             node_id=np.concatenate([np.arange(n, dtype=np.int64) for n in sizes]) if len(sizes) else np.zeros(0, np.int64),
             edge_id=np.concatenate([np.arange(m, dtype=np.int64) for m in esizes]) if len(esizes) else np.zeros(0, np.int64),
             max_nv=int(sizes.max()) if len(sizes) else 0, max_nvl=4, max_ne=int(esizes.max()) if len(esizes) else 0,
             max_nel=num_real_types, num_rels=num_real_types + 2)
    return b

