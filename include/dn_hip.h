/* dn_hip.h -- C ABI of libdn_hip.so: the MI355X (gfx950) hot path of
 * HKUST-KnowComp/DummyNode4GraphLearning (dummy-node / edge-to-vertex index builds and the
 * gather -> segment-reduce message passing of the GIN / RGCN / RGIN layers).
 *
 * The reference has NO C/FFI interface for this path: the arithmetic runs inside third-party
 * engines (DGL update_all / torch-scatter) behind Python nn.Modules.  The entry
 * points below are therefore exactly what a ctypes binding on the reference side would bind in
 * place of those engine calls; each one cites the reference call site it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*.
 *   - the caller (PyTorch) owns every buffer: inputs, outputs and workspace; the library never
 *     allocates, frees or retains a pointer.  Workspace sizes come from *_workspace_bytes().
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing else orders it.
 *   - return 0 on success, <0 on error (DN_ERR_*); dn_last_error() gives the thread-local message.
 *   - re-entrant, no global mutable state, no environment access of its own (the experiment knobs of the kernels are
 *     compile-time constants; only a -DDN_TUNING_ENV build reads DN_* variables.  rocPRIM, which the index builds call,
 *     consults the environment for its own target selection), deterministic: no floating-point atomics anywhere,
 *     every sum has a fixed order (bitwise reproducible run to run).
 *   - node / edge ids are int32 on the device (N, E < 2^31); the Python boundary converts the
 *     reference's int64 ids.
 *   - feature matrices are row-major [rows, H]; `_f32` = float, `_bf16` = bfloat16 storage with
 *     fp32 accumulation.
 */
#ifndef DN_HIP_H
#define DN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_OK 0
#define DN_ERR_ARG (-1)
#define DN_ERR_HIP (-2)
#define DN_ERR_WORKSPACE (-3)
#define DN_ERR_UNSUPPORTED (-4)

typedef void* dn_stream_t; /* hipStream_t */

int dn_version(void);
const char* dn_last_error(void);
/* Returns 0 iff `device_ptr` is device memory known to the HIP runtime this library resolved to
 * (guards against a second libamdhip64 being loaded next to PyTorch's). */
int dn_runtime_probe(const void* device_ptr);

/* ------------------------------------------------------------------------------------------
 * Gather + segment-sum (the message-passing core).
 *   out[s,:] = self_coef * self_in[s,:] + sum_{i in [ptr[s], ptr[s+1])} scale[i] * in[idx[i],:]
 * idx == NULL -> idx[i] = i;  ptr == NULL -> segment s = {s} (pure row gather, S == M);
 * scale == NULL -> 1;  self_in == NULL -> no self term.  M = number of gathered elements.
 * mean != 0 divides the gathered sum (not the self term) by the segment length (0 for empty segments).
 * Replaces: PyG GINConv.propagate / RGCNConv.propagate = x.index_select(0, src) + torch_scatter
 *   (graph_classification/graph_neural_networks/models/gconv.py:212, rgconv.py:40-41,121) and
 *   DGL update_all(msg, fn.sum) (subgraph_isomorphism/models/rgin.py:159, rgcn.py:196).
 * The backward of this operator w.r.t. `in` is the same call on the transposed index.
 * ------------------------------------------------------------------------------------------ */
int dn_gather_segsum_f32(const float* in, int64_t in_rows, int32_t H, const int32_t* idx,
                         const float* scale, const int32_t* ptr, int64_t S, int64_t M, float* out,
                         const float* self_in, float self_coef, int32_t mean, dn_stream_t stream);
int dn_gather_segsum_bf16(const void* in, int64_t in_rows, int32_t H, const int32_t* idx,
                          const float* scale, const int32_t* ptr, int64_t S, int64_t M, void* out,
                          const void* self_in, float self_coef, int32_t mean, dn_stream_t stream);

/* Per-graph readouts over CONTIGUOUS rows: out[g,:] = reduce_{v in [ptr[g], ptr[g+1])} in[v,:].
 * Replaces: torch_geometric global_add_pool / global_mean_pool / global_max_pool
 *   (gconv.py:53,95,148,210,213; rgconv.py:42,119,124) and SI SumPredictNet's sum over the padded
 *   node dimension (subgraph_isomorphism/models/pred.py:215-216).
 * mean: empty segments give 0.  max: empty segments give 0 and argmax -1; ties keep the lowest row
 *   (argmax [S,H] int32 is the row index, used by dn_segment_max_bwd_*). */
int dn_segment_sum_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, dn_stream_t stream);
int dn_segment_sum_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, dn_stream_t stream);
int dn_segment_mean_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, dn_stream_t stream);
int dn_segment_mean_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, dn_stream_t stream);
int dn_segment_max_f32(const float* in, int32_t H, const int32_t* ptr, int64_t S, float* out, int32_t* argmax,
                       dn_stream_t stream);
int dn_segment_max_bf16(const void* in, int32_t H, const int32_t* ptr, int64_t S, void* out, int32_t* argmax,
                        dn_stream_t stream);
/* grad_in[v,h] = (argmax[g(v),h] == v) ? grad_out[g(v),h] : 0, rows of segment g contiguous. */
int dn_segment_max_bwd_f32(const float* grad_out, const int32_t* argmax, int32_t H, const int32_t* ptr, int64_t S,
                           float* grad_in, dn_stream_t stream);
int dn_segment_max_bwd_bf16(const void* grad_out, const int32_t* argmax, int32_t H, const int32_t* ptr, int64_t S,
                            void* grad_in, dn_stream_t stream);

/* Per-edge dot product (SDDMM): out[e] = < a[ia[e], :], b[ib[e], :] >, fp32 out; ia / ib may be NULL (row e).
 * The gradient of a per-edge scalar weight w_e in  out[dst] += w_e * x[src]  is < x[src_e], grad_out[dst_e] >: this
 * is what lets the reference's trainable dummy-edge weight (gconv.py:29-34,46-49: edge_attr[is_dummy_edge] =
 * dummy_weight, passed to GCNConv) receive its gradient without PyG's autograd through index_select / scatter. */
int dn_edge_dot_f32(const float* a, const int32_t* ia, const float* b, const int32_t* ib, int32_t H, int64_t E,
                    float* out, dn_stream_t stream);
int dn_edge_dot_bf16(const void* a, const int32_t* ia, const void* b, const int32_t* ib, int32_t H, int64_t E,
                     float* out, dn_stream_t stream);

/* Gather + segment MAX: out[s, h] = max_{i in [ptr[s], ptr[s+1])} in[idx[i], h]; argmax[s, h] = that slot i (-1 and
 * out 0 for an empty segment; ties keep the lowest slot).  Replaces PyG SAGEConv's 'max' neighbour aggregation
 * (gconv.py:130-132 `conv.aggr = self.aggregation`).  Backward: grad_in[u, h] = sum over the slots i that gathered
 * row u (tptr [rows+1], tslot: slots grouped by gathered row; seg_of_slot [M]) of
 * (argmax[seg(i), h] == i ? grad_out[seg(i), h] : 0) -- a gather over the transposed index, no atomics. */
int dn_gather_segmax_f32(const float* in, const int32_t* idx, const int32_t* ptr, int64_t S, int32_t H, float* out,
                         int32_t* argmax, dn_stream_t stream);
int dn_gather_segmax_bf16(const void* in, const int32_t* idx, const int32_t* ptr, int64_t S, int32_t H, void* out,
                          int32_t* argmax, dn_stream_t stream);
int dn_gather_segmax_bwd_f32(const float* grad_out, const int32_t* argmax, const int32_t* tptr, const int32_t* tslot,
                             const int32_t* seg_of_slot, int64_t rows, int32_t H, float* grad_in, dn_stream_t stream);
int dn_gather_segmax_bwd_bf16(const void* grad_out, const int32_t* argmax, const int32_t* tptr, const int32_t* tslot,
                              const int32_t* seg_of_slot, int64_t rows, int32_t H, void* grad_in, dn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One-shot device-side index builds.
 * ------------------------------------------------------------------------------------------ */

/* Stable grouping of element indices by integer key in [0, num_keys):
 *   ptr[k] .. ptr[k+1] = positions in `perm` of the elements with key k, ascending element index.
 * This is the CSR/CSC build DGL / torch-scatter perform internally for the call sites above. */
size_t dn_csr_build_workspace_bytes(int64_t M, int64_t num_keys);
int dn_csr_build_i32(const int32_t* key, int64_t M, int64_t num_keys, int32_t* ptr /*[num_keys+1]*/,
                     int32_t* perm /*[M]*/, void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* Dummy-node augmentation of a batched graph (disjoint union; node_ptr/edge_ptr [G+1]; src/dst global
 * ids; edges of a graph contiguous in eid order).  Outputs: N' = N + G nodes, E' = E + 2N edges.
 * GC layout -- replaces load_graph_data_from_TUDatadir(with_dummy=True)'s per-graph igraph build
 *   (graph_classification/data_processing/tu_data_processing.py:186-200,213-214): dummy vertex n with
 *   label 0; edges = m originals then INTERLEAVED (n,v),(v,n), label 0, IS_DUMMY 1; ids = local index. */
int dn_dummy_augment_gc_i32(int64_t G, int64_t N, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                            const int32_t* src, const int32_t* dst, const int32_t* node_label,
                            const int32_t* edge_label, int32_t* out_node_ptr, int32_t* out_edge_ptr,
                            int32_t* out_src, int32_t* out_dst, int32_t* out_node_label, int32_t* out_edge_label,
                            uint8_t* out_is_dummy_node, uint8_t* out_is_dummy_edge, int32_t* out_node_id,
                            int32_t* out_edge_id, dn_stream_t stream);
/* SI layout -- replaces add_dummy_nodes_edges, GraphAdj branch (subgraph_isomorphism/train.py:404-474):
 *   dummy node (id = max_nv, label = max_nvl); 2n edges BLOCKED: all (u -> dummy) then all (dummy -> u);
 *   edge id max_ne / max_ne+1; relation max_nel / max_nel+1; is_dummy 1; is_reversed 0 / 1.
 *   in_is_reversed may be NULL (treated as 0). */
int dn_dummy_augment_si_i32(int64_t G, int64_t N, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                            const int32_t* src, const int32_t* dst, const int32_t* node_id,
                            const int32_t* node_label, const int32_t* edge_id, const int32_t* edge_label,
                            const uint8_t* in_is_reversed, int32_t max_nv, int32_t max_nvl, int32_t max_ne,
                            int32_t max_nel, int32_t* out_node_ptr, int32_t* out_edge_ptr, int32_t* out_src,
                            int32_t* out_dst, int32_t* out_node_id, int32_t* out_node_label,
                            int32_t* out_edge_id, int32_t* out_edge_label, uint8_t* out_is_dummy_node,
                            uint8_t* out_is_dummy_edge, uint8_t* out_is_reversed, dn_stream_t stream);

/* Edge-to-vertex ("conjugate") transform L_Phi of a batched graph.
 * Replaces convert_conjugate_graph_forward (tu_data_processing.py:223-338; mode DN_CONJ_GC, and
 * DN_CONJ_LINE for graphs without the IS_DUMMY attribute) and convert_conjugate_graph, igraph branch
 * (subgraph_isomorphism/utils/graph.py:177-267; mode DN_CONJ_SI).  Bit-exact incl. output order.
 *   step 1  dn_conjugate_count_i32: *host_num_raw = sum_e in_deg(src(e)) (synchronises the stream).
 *   step 2  dn_conjugate_build_i32: outputs sized by the upper bounds N'<=E, E'<=num_raw;
 *           host_counts[0] = N' (conj vertices), host_counts[1] = E' (conj edges) (synchronises).
 *   rep_edge[k]    input edge whose attributes conj-vertex k copies (first edge carrying that id)
 *   shared_node[t] input vertex whose attributes conj-edge t copies (the vertex the 2-path goes through)
 *   edge_id: per-graph edge ids (SI: edata["id"], dummies share ids); NULL = local edge index (GC). */
#define DN_CONJ_GC 0
#define DN_CONJ_SI 1
#define DN_CONJ_LINE 2
/* num_raw = -1 sizes the workspace for the count step only. */
size_t dn_conjugate_workspace_bytes(int64_t G, int64_t N, int64_t E, int64_t num_raw);
int dn_conjugate_count_i32(int64_t N, int64_t E, const int32_t* src, const int32_t* dst, int64_t* host_num_raw,
                           void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_conjugate_build_i32(int32_t mode, int64_t G, int64_t N, int64_t E, int64_t num_raw,
                           const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* src,
                           const int32_t* dst, const int32_t* node_label, const int32_t* edge_id,
                           const uint8_t* is_dummy_edge, int32_t* out_cnode_ptr /*[G+1]*/,
                           int32_t* out_cedge_ptr /*[G+1]*/, int32_t* out_csrc /*[num_raw]*/,
                           int32_t* out_cdst /*[num_raw]*/, int32_t* out_rep_edge /*[E]*/,
                           int32_t* out_shared_node /*[num_raw]*/, int64_t* host_counts /*[2]*/,
                           void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* Graph-local neighbour sum on the matrix cores, for the rows of the listed tiles only:
 *   out[v, :] = self_coef * x[v, :] + sum_{i in [ptr[v], ptr[v+1])} x[idx[i], :]
 * -- GINConv's aggregation (graph_classification/graph_neural_networks/models/gconv.py:197: PyG propagate(aggr="add") plus
 * (1 + eps) x_i) and, on the transposed index, its input gradient.  tiles [num_tiles][4] = {first row, end row, ptr[first row],
 * ptr[end row]} of runs of WHOLE graphs with at most 64 rows each; seg (may be NULL) = the row every index entry belongs to
 * (saves a search in the tile's bounds): every idx of a tile's rows must lie inside the tile (else *bad |= 1 and that entry is
 * skipped; tiles that leave [0, num_rows) or hold more than 64 rows are flagged the same way and not touched; *bad must be
 * zeroed by the caller).  x / out have num_rows rows, idx num_entries entries.  A tile's rows are read once, split into three bf16 planes (hi + mid + lo = the
 * fp32 value to rounding), the tile's adjacency counts form a bf16 matrix in LDS and the sum is a dense product with fp32
 * accumulation: same value as dn_gather_segsum_f32 up to fp32 summation order for FINITE inputs; a non-finite element turns its
 * column of the whole tile into NaN (0 x Inf in the dense product, Inf - Inf in the split), where the gather would pass it to its
 * neighbours only.  fp32 rows, H in {64, 128, 256}. */
int dn_graph_tile_sum_f32(const float* x, int64_t num_rows, int32_t H, const int32_t* ptr, const int32_t* idx,
                          const int32_t* seg, int64_t num_entries, const int32_t* tiles, int64_t num_tiles,
                          float self_coef, float* out, int32_t* bad, dn_stream_t stream);

/* The same sum for a LIST of rows, straight from the full CSR (no compacted copy of lists or rows): the companion of
 * dn_graph_tile_sum_f32 for the rows of graphs too large for a tile.  rows [num_listed]: row ids (rows_are_records = 0) or
 * 16-byte records {row, ptr[row], ptr[row + 1], 0} (rows_are_records = 1: saves the ptr round trip), each row listed once; any
 * list length (an empty list gives self_coef * x[s]).  workgroup_per_row = 0: one lane group per
 * listed row; 1: one workgroup per listed row, its lane groups sharing the list (hubs: hundreds of entries), partial rows added
 * in fixed order.  Rows not listed are left untouched.  Replaces the same PyG aggregation (gconv.py:197). */
int dn_gather_rows_sum_f32(const float* x, int32_t H, const int32_t* ptr, const int32_t* idx, const int32_t* rows,
                           int32_t rows_are_records, int64_t num_listed, int32_t workgroup_per_row, float self_coef,
                           float* out, dn_stream_t stream);

/* Host-side tile packing for dn_graph_tile_sum_f32 (no GPU): greedy runs of whole graphs with at most max_rows rows each
 * (node_ptr [G+1] on the HOST); graphs with more rows are left out (their rows go to dn_gather_rows_sum_f32).  tiles receives
 * {first row, end row} pairs, *num_tiles their number; DN_ERR_WORKSPACE when cap pairs do not suffice (G always do). */
int dn_graph_tiles_host(const int32_t* node_ptr, int64_t G, int32_t max_rows, int32_t* tiles, int64_t cap, int64_t* num_tiles);

/* Relation-aware segment index for the aggregate-then-transform form of the RGCN/RGIN message pass
 *   sum_e x[src_e] W[etype_e]  ==  sum_r ( sum_{e in r, dst=v} x[src_e] ) W_r      
 * which replaces the reference's per-edge weight gather + bmm (subgraph_isomorphism/models/rgin.py:102-120,
 * rgcn.py:100-122) and PyG RGCNConv's per-relation mask/propagate loop (rgconv.py:17-18,96).
 * Segments p = distinct (rel, dst) pairs, ordered by (rel, dst); P = number of segments (<= E).
 *   perm1 [E]      edge ids sorted by (rel, dst), stable;  src1 [E] = src[perm1]
 *   seg_ptr [P+1]  (allocate E+1) edge range of each segment in perm1 order;  seg_dst [P] (allocate E)
 *   rel_ptr [R+1]  segment range of each relation (also copied to host_rel_ptr)
 *   dptr [N+1], sperm [P] (allocate E)   segments grouped by destination, ascending relation
 *   optr [N+1], operm [E]                edges grouped by source, ascending edge id
 *   seg_by_src [E]                       segment of edge operm[i]
 * Synchronises the stream (P and rel_ptr are returned to the host). */
size_t dn_rel_index_workspace_bytes(int64_t N, int64_t R, int64_t E);
int dn_rel_index_build_i32(int64_t N, int64_t R, int64_t E, const int32_t* src, const int32_t* dst,
                           const int32_t* etype, int32_t* perm1, int32_t* src1, int32_t* seg_ptr,
                           int32_t* seg_dst, int32_t* rel_ptr, int32_t* dptr, int32_t* sperm, int32_t* optr,
                           int32_t* operm, int32_t* seg_by_src, int64_t* host_P, int32_t* host_rel_ptr,
                           void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* Row factorisation of the relation-wise message pass for the bf16 matrix-core path:
 *   out[v] = sum_{e: dst(e)=v} x[src(e)] W[etype(e)]   ==   sum over rows p with v in out(p) of  in_row(p) @ W[rel(p)]
 * Every row has ONE input row.  Per relation r (E_r edges, D_r distinct destinations, S_r distinct sources):
 *   EDGE (min(D_r,S_r) > edge_frac*E_r): one row per edge, in = x[src], out -> dst
 *   AGG  (D_r <= S_r): one row per distinct dst, in = N + a (a-th pre-aggregated row: sum of x[src] over the segment), out -> dst
 *   TF   (else):       one row per distinct src, in = x[src], out = N + b (the b-th backward pre-aggregated row) and the
 *                       product is added to every dst of that (rel, src)
 * Edge rows are relation-major (host_rel_ptr [R+1]); with self_loop != 0, N more rows (relation R, in = out = v) follow.
 * Outputs (device, caller-allocated upper bounds): row_in / row_out [E+N]; aux_f_ptr [E+1], aux_f_idx [E] (lists of x
 * rows per AGG row); aux_b_ptr [E+1], aux_b_idx [E] (lists of destination rows per TF row); dst_ptr [N+2], dst_rows [2E+N]
 * (rows contributing to each node, forward); src_ptr [N+2], src_rows [2E+N] (rows contributing to each node's input
 * gradient); only the first N+1 ptr entries are meaningful.  host_counts[5] = {#edge rows P, #AGG rows, #TF rows,
 * #AGG edges, #TF edges}; host_modes [R] (0 EDGE, 1 AGG, 2 TF).  Synchronises the stream.
 * This is the index the reference's DGL path never needs because it materialises a per-edge [E,H,H] weight tensor instead
 * (subgraph_isomorphism/models/rgin.py:109-110). */
size_t dn_row_index_workspace_bytes(int64_t N, int64_t R, int64_t E);
int dn_row_index_build_i32(int64_t N, int64_t R, int64_t E, const int32_t* src, const int32_t* dst,
                           const int32_t* etype, int32_t self_loop, float edge_frac, int32_t* row_in,
                           int32_t* row_out, int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr,
                           int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows, int32_t* src_ptr,
                           int32_t* src_rows, int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes,
                           void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* The same tables as dn_row_index_build_i32, bit for bit, for a batch that is a disjoint union of G graphs: node_ptr [G+1] /
 * edge_ptr [G+1] (device) give every graph a contiguous node range and a contiguous edge range (node_ptr[G] == N,
 * edge_ptr[G] == E) -- the layout dgl.batch / the PyG collate produce (subgraph_isomorphism/dataset.py:1605-1611,
 * graph_classification main.py:245-247).  Every ordering the general builder establishes with device-wide radix sorts is local
 * to a graph in such a batch, so one wavefront rank-sorts one graph in LDS and the only batch-wide step is one exclusive scan
 * of the packed counts: 3 kernels + 1 scan instead of ~60 launches.  Graphs of more than 1024 edges (up to 8191 edges and 8191
 * nodes; round 6) are taken by a second launch of the statistics and fill passes, one 1024-thread workgroup per graph -- a
 * PER-GRAPH fallback: the rest of the batch stays on the one-wavefront kernels.  R <= 64.  *host_status = 1 (outputs undefined) when the
 * batch does not qualify -- a graph of 8192 edges or more (or over 16384 nodes), an endpoint outside its graph's node range, a relation id
 * outside [0, R), ranges that do not tile [0, N) / [0, E): the caller then runs dn_row_index_build_i32.  Other arguments and
 * outputs as dn_row_index_build_i32.  rel_ptr_dev (device, may be NULL; round 5): [R + 2] = the relation offsets of host_rel_ptr
 * followed by the end of the self-loop rows (P + N), for the device-side table builders -- no upload of what the device already
 * has.  host_absorb (host int32 [2], may be NULL; round 5) with tile_ptr_f / _b (device [G + 1]) and fold_info_f / _b (device
 * [G][12], 16-byte aligned): the builder also answers, per direction, whether dn_rows_close_bf16 can ABSORB the fold of this batch
 * -- exactly one collapsed relation in the direction (mode AGG forward / TF backward) owning all the direction's aux lists, with
 * a self loop, and its segments passing dn_fold_graph_tiles_build_i32's test with the relation's own rows as targets -- and
 * leaves that call's tile_ptr / fold_info behind (host_absorb[d] bit 0); bit 1 (round 6): the segments pass the test WITHOUT the
 * 32-node limit (dn_fold_graph_tiles_multi_build_i32 would say yes); host_absorb[2] / [3] = the nodes / edges of the batch's
 * largest graph (what dn_conv_graphs_bf16 asks before it takes a batch); host_absorb: int32 [4]; all ride in the builder's ONE read-back.
 * Synchronises the stream (one read-back). */
size_t dn_row_index_local_workspace_bytes(int64_t G, int64_t N, int64_t R, int64_t E);
int dn_row_index_build_local_i32(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr,
                                 const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype,
                                 int32_t self_loop, float edge_frac, int32_t* row_in, int32_t* row_out,
                                 int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr, int32_t* aux_b_idx,
                                 int32_t* dst_ptr, int32_t* dst_rows, int32_t* src_ptr, int32_t* src_rows,
                                 int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes, int32_t* host_status, int32_t* rel_ptr_dev,
                                 int32_t* tile_ptr_f, int32_t* fold_info_f, int32_t* tile_ptr_b, int32_t* fold_info_b,
                                 int32_t* host_absorb,
                                 void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* One launch per direction of a relational conv on a batch of SMALL graphs at the reference's default width (round 6; H = 64
 * bf16, the CLI default `--hid_dim` of subgraph_isomorphism/config.py:456-461, BASELINE config 3):
 *   out[v, :] = sum_{e: key_out[e] = v} X[key_in[e], :] @ W[etype[e]] + X[v, :] @ W_loop (+ bias)
 * = message UDF + fn.sum + self loop of rgin.py:102-160 / rgcn.py:160-196 with (key_in, key_out) = (src, dst); the input gradient
 * with (dst, src) and w_kn = 0.  W [num_rels][H][H] and W_loop [H][H]: w_kn = 1: stored [k][n] (the parameters `weight` /
 * `loop_weight` as they are: out = x @ W); w_kn = 0: the same memory read as [n][k] (out = x @ W^T: the input-gradient pass on the
 * untransposed parameters).  One workgroup takes one graph straight from the batch's raw arrays (node_ptr / edge_ptr [num_graphs + 1],
 * key_in / key_out / etype [E], global node ids, every edge inside its graph's node range): no row index, no intermediate rows in
 * HBM; every relation is taken edge by edge.  Limits: every graph within dn_conv_graphs_max_nodes() = 64 nodes and
 * dn_conv_graphs_max_edges() = 1024 edges (ask dn_row_index_build_local_i32's host_absorb[2..3]), num_rels <= 16; a graph outside
 * them, or an edge outside its graph / relation range, sets *dev_err (device int32, OR-ed; its rows are left unwritten).
 * aux (may be NULL): [num_graphs][H], aux[j] = bf16 column sum of the INPUT rows of segment j = nodes seg_nodes[seg_ptr[j] ..
 * seg_ptr[j+1]) (a contiguous run inside graph j: the pre-aggregated operand dn_rows_wgrad_bf16 takes for a collapsed relation).
 * fp32 accumulation in a fixed order (bitwise repeatable), every per-edge product rounded to bf16 once, the output row once. */
/* dn_layer_graphs_fwd_bf16 / _bwd_bf16: a whole RGIN layer (rgin.py:102-160 + the MLP of rgin.py:50-57,147-151: Linear - act - Linear
 * - act with act = ReLU (act_slope 0) or leaky ReLU) around the same per-graph conv, ONE launch each way:
 *   fwd: conv_out = conv(X) (+ bias), mid = act(conv_out @ W1^T + b1), out = act(mid @ W2^T + b2); bits1 / bits2 [N][H / 8] = the
 *        sign bits (> 0) of mid / out; aux as in dn_conv_graphs_bf16 (column sums of X).  W / W_loop stored [k][n] (the parameters),
 *        W1 / W2 [out][in] (nn.Linear.weight).
 *   bwd: G = the gradient of out: g_mid = mask1(mask2(G) @ W2), g_conv = g_mid @ W1, g_in = the conv's input gradient of g_conv
 *        (edges dst -> src, W read transposed); aux = column sums of g_conv over the segments (the weight gradient's operand).
 * Same limits, error flag and numerics as dn_conv_graphs_bf16; every row tensor [N][H] bf16. */
int dn_layer_graphs_fwd_bf16(const void* X, int32_t H, const void* W, const void* W_loop, const void* bias, int32_t num_rels,
                             const void* W1, const void* b1, const void* W2, const void* b2, float act_slope, const int32_t* node_ptr,
                             const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t num_graphs,
                             int64_t N, void* conv_out, void* mid, void* out, void* bits1, void* bits2, const int32_t* seg_ptr,
                             const int32_t* seg_nodes, void* aux, int32_t* dev_err, dn_stream_t stream);
int dn_layer_graphs_bwd_bf16(const void* G, int32_t H, const void* W, const void* W_loop, int32_t num_rels, const void* W1,
                             const void* W2, float act_slope, const void* bits1, const void* bits2, const int32_t* node_ptr,
                             const int32_t* edge_ptr, const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t num_graphs,
                             int64_t N, void* g_mid, void* g_conv, void* g_in, const int32_t* seg_ptr, const int32_t* seg_nodes,
                             void* aux, int32_t* dev_err, dn_stream_t stream);
int32_t dn_conv_graphs_max_nodes(void);
int32_t dn_conv_graphs_max_edges(void);
int dn_conv_graphs_bf16(const void* X, int32_t H, const void* W, int32_t w_kn, const void* W_loop, const void* bias, int32_t num_rels,
                        const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* key_in, const int32_t* key_out,
                        const int32_t* etype, int64_t num_graphs, int64_t N, void* out, const int32_t* seg_ptr, const int32_t* seg_nodes,
                        void* aux, int32_t* dev_err, dn_stream_t stream);

/* The per-batch index of the H = 256 bf16 conv path as ONE call on one arena (round 5) -- what dgl.batch + update_all pay per step
 * in the reference (subgraph_isomorphism/dataset.py:1605-1611; models/rgin.py:156-160): dn_row_index_build_local_i32 (same
 * arguments and outputs, rel_ptr_dev and the absorbed-fold buffers required) and, queued behind it BEFORE the one read-back,
 *   - the unit streams of both closing launches with the graphs as tiles and the AGG units appended (dn_close_units_build_i32 with
 *     tile_ptr_f / _b, agg_units = 1, xcd_order = close_xcd_order, the folded relation's rows dropped): unit_ptr_* [num_wg + 1], units_* [unit_capacity][4]
 *     (16-byte aligned; unit_capacity >= dn_close_units_capacity(G, E + N, num_wg)), ent_row_* / ent_mask_* [E + N];
 *   - the sweep orders of both transform launches with the folded relation skipped (dn_sweep_tables_build_i32):
 *     sweep_f / sweep_b [8 * sweep_wg_per_group * sweep_tiles_per_wg][4] as CAPACITY (sized by E, a bound of the rows): the
 *     builder lays each table out with the slots its fullest group needs, host_plan[4] / host_plan[5] (forward / backward) slots per
 *     workgroup -- the table is [8 * sweep_wg_per_group * host_plan[4 + d]][4] -- or -1 when not even the plain order fits;
 *     sweep_tiles_per_wg = 0: none;
 * the counts these builders need (edge rows, the folded relation and its row range) are read from device memory.  They serve the
 * batches whose fold can be absorbed -- host_plan[0] (forward) / host_plan[1] (backward) != 0: the build is valid, the direction
 * has its one collapsed relation (host_absorb != 0: bit 0 = every block within 32 nodes, bit 1 = the segments pass the test without
 * that limit) and its segments are the batch's G graphs.  host_plan[d] = 1: the tables above, the graphs as tiles.  host_plan[d] = 2
 * (round 6; a graph over 32 nodes, chunks_per_wg > 0): the SAME launches have built the chunked form instead -- chunk_tile_* /
 * chunk_graph_* [chunks_per_wg * num_wg + 1], tile_ptr_m* [tile_capacity + 1], fold_info_m* [tile_capacity][12] as
 * dn_fold_graph_tiles_multi_build_i32 leaves them (tile_capacity >= dn_fold_graph_tiles_multi_capacity(N, chunks_per_wg * num_wg)),
 * unit streams in order 2 + close_xcd_order (unit_capacity and the workspace then cover max(G, tile_capacity) tiles); which form was
 * needed is decided on the device.  For a direction with 0 its tables are left untouched and the caller builds them with the
 * separate entry points.  chunks_per_wg = 0: no chunked form (the eight pointers may be NULL).  After the read-back the split-K chunk table of the weight gradient over all rows
 * (dn_row_tables_build_i32 with piece_ptr) is queued: host_plan[2] = rows per chunk -- the smallest multiple of 64 (>= 256, <=
 * wgrad_max_chunk_rows) for which the chunks of all relations fit one round of wgrad_workgroups -- host_plan[3] = its entries
 * (rows / chunk + relations + 1 <= chunk_capacity); chunk_table [chunk_capacity][4], chunk_ptr [R + 2]; host_plan: int32 [6].
 * *host_status != 0 as for
 * dn_row_index_build_local_i32 (nothing else is valid).  workspace: 256-byte aligned.  Synchronises the stream once. */
size_t dn_conv_index_workspace_bytes(int64_t G, int64_t N, int64_t R, int64_t E, int32_t num_wg, int64_t tile_capacity);
int dn_conv_index_build_i32(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr,
                            const int32_t* src, const int32_t* dst, const int32_t* etype, int32_t self_loop, float edge_frac,
                            int32_t* row_in, int32_t* row_out, int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr,
                            int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows, int32_t* src_ptr, int32_t* src_rows,
                            int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes, int32_t* host_status,
                            int32_t* rel_ptr_dev, int32_t* tile_ptr_f, int32_t* fold_info_f, int32_t* tile_ptr_b,
                            int32_t* fold_info_b, int32_t* host_absorb, int32_t num_wg, int32_t close_xcd_order, int64_t unit_capacity, int32_t* unit_ptr_f,
                            int32_t* units_f, int32_t* ent_row_f, uint32_t* ent_mask_f, int32_t* unit_ptr_b, int32_t* units_b,
                            int32_t* ent_row_b, uint32_t* ent_mask_b, int32_t chunks_per_wg, int64_t tile_capacity, int32_t* chunk_tile_f,
                            int32_t* chunk_graph_f, int32_t* tile_ptr_mf, int32_t* fold_info_mf, int32_t* chunk_tile_b,
                            int32_t* chunk_graph_b, int32_t* tile_ptr_mb, int32_t* fold_info_mb, int32_t sweep_wg_per_group,
                            int32_t sweep_tiles_per_wg,
                            int32_t* sweep_f, int32_t* sweep_b, int32_t wgrad_workgroups, int32_t wgrad_max_chunk_rows,
                            int64_t chunk_capacity, int32_t* chunk_table, int32_t* chunk_ptr, int32_t* host_plan, void* workspace,
                            size_t workspace_bytes, dn_stream_t stream);

/* Tile / chunk tables of relation-major rows for dn_rows_transform_* (step = 32 rows) and dn_rows_wgrad_* (step = the
 * split-K chunk size), built on the device from rel_ptr [num_rels + 1] (device): entry i = {rel, beg, end, 0}.  The caller
 * sizes `table` by the upper bound max_entries >= rows / step + num_rels; unused entries become empty pieces (beg == end) of
 * the last relation.  piece_ptr (may be NULL) receives the [num_rels + 1] piece ranges per relation (dn_rows_wgrad_*'s
 * chunk_ptr).  skip_mask: bit r set (r < 64) leaves relation r out of the table (no pieces) -- the relation a caller handles
 * in a launch of its own.  Replaces the per-batch bookkeeping DGL does inside dgl.batch / update_all
 * (subgraph_isomorphism/dataset.py:1605-1611, models/rgin.py:156-160); no device -> host synchronisation. */
int dn_row_tables_build_i32(int32_t num_rels, const int32_t* rel_ptr, int32_t step, int64_t max_entries, int32_t* table,
                            int32_t* piece_ptr, uint64_t skip_mask, dn_stream_t stream);

/* L2-blocked ("sweep") tile order of relation-major rows for the persistent launch of dn_rows_transform_bf16 (H = 256: one
 * workgroup per CU, 256 in all).  The batch is cut into 8 contiguous key ranges (key of row p = the node it belongs to:
 * row_out[p] when < num_nodes, else row_in[p]; non-decreasing inside a relation for the tables of dn_row_index_build_*); group x
 * is walked by the workgroups b = j * 8 + x (blocks b, b + 8, ... share an XCD and its L2).  Inside a group most workgroups serve
 * ONE relation for the whole launch (weights stay in registers) and take every k-th of its tiles, so all of them move through
 * the group's graphs at the same pace and a source row fetched for one relation is still in that XCD's L2 when the other
 * relations ask for it; a few helper workgroups take the left-overs of several relations one after the other (their tiles cost
 * more -- rows out of phase with the group's sweep, a weight reload per relation -- so the quota of the others grows by a few tiles
 * until both kinds finish together; the transform launch does not walk a workgroup's trailing empty slots).  Groups with fewer
 * than 64 tiles per workgroup (an eighth of BASELINE config 5: 32) have no such workgroups: the group's tile line, relation-major,
 * is cut into equal segments (at most two relations per workgroup when a relation has more tiles than a segment) -- a helper's
 * weight reloads, one per relation of the group, were a third of such a launch.
 * table: [8 * workgroups_per_group][tiles_per_workgroup][4] int32 {rel, beg, end, 0}, unused slots empty -- pass it to
 * dn_rows_transform_bf16 with num_tiles = 8 * workgroups_per_group * tiles_per_workgroup and workgroups_per_group = 32 (the
 * launch then gives workgroup b the entries [b * tiles_per_workgroup, (b + 1) * tiles_per_workgroup)).  Every row of every
 * relation not in skip_mask is covered exactly once whatever the keys are (placement is a speed matter only).  When a group
 * needs more than tiles_per_workgroup slots the table holds the plain relation-major order instead (always valid provided
 * 8 * workgroups_per_group * tiles_per_workgroup >= rows / 32 + num_rels); info (may be NULL) receives {0: sweep order, 1: plain order taken, 2: plain
 * order taken AND the table cannot hold all its tiles (8 * workgroups_per_group * tiles_per_workgroup < sum over relations of
 * ceil(rows / 32)): rows would go untransformed, size the table as stated; slots the largest group needs}.  One launch, no host synchronisation.  Replaces nothing in the reference (DGL's update_all has no
 * notion of a traversal order, models/rgin.py:156-160): it is the order in which the replacement reads x. */
int dn_sweep_tables_build_i32(int32_t num_rels, const int32_t* rel_ptr, const int32_t* row_in, const int32_t* row_out,
                              int64_t num_nodes, int32_t workgroups_per_group, int32_t tiles_per_workgroup, uint64_t skip_mask,
                              int32_t* table, int32_t* info, dn_stream_t stream);

/* Fixed-width slot table of per-node row lists for dn_rows_selfsum_bf16 (one-shot index build, ONE launch, no workspace, no
 * host synchronisation; replaces the reference's per-node reduce bookkeeping inside `g.update_all(..., fn.sum(...))`,
 * subgraph_isomorphism/models/rgin.py:137).
 * list_ptr [N+1] / list_rows: CSR of row ids per node (dn_row_index_build_i32's dst_ptr/dst_rows or src_ptr/src_rows).
 * Rows >= num_edge_rows (the self-loop rows) and rows in [drop_beg, drop_end) (a relation the caller adds in a launch of its
 * own; drop_beg == drop_end for none) are dropped; drop_enable (device, may be NULL) switches the drop range off when
 * *drop_enable == 0 -- the verdict dn_fold_tables_build_async_i32 leaves on the device, so the tables of both directions of a
 * batch are queued back to back (the per-batch bookkeeping of subgraph_isomorphism/dataset.py:1605-1611).
 * slots [N, K]: the kept rows in list order, -1 padded; a node with MORE than K kept rows keeps its first K-1 and gets -2 in
 * slot K-1 and 1 in overflow[v] (uint8 [N], may be NULL; 0 for every other node): dn_overflow_rows_add_bf16 adds the rest
 * straight from the node's list after the closing launch (pass it `overflow`, the same list and the drop range). */
int dn_slot_table_build_i32(int64_t N, int32_t num_edge_rows, int32_t K, const int32_t* list_ptr, const int32_t* list_rows,
                            int32_t drop_beg, int32_t drop_end, const int32_t* drop_enable, int32_t* slots, uint8_t* overflow,
                            dn_stream_t stream);

/* Weight gradient of the relation-wise transform Y[p] = A[p] W[rel(p)] on the matrix cores (bf16 in, fp32 acc):
 *   out[r] = sum_{p in relation r} A[idx_a[p], :]^T G[idx_g[p], :]            ([Hi x Ho] per relation)
 * Replaces autograd's backward of the reference's per-edge `th.bmm(x[src], W[etype])`
 * (subgraph_isomorphism/models/rgin.py:109-110,117-118) / PyG RGCNConv's `h @ weight[i]` (rgconv.py:96).
 * Rows are relation-major; the caller splits them into row chunks {rel, beg, end, 0} (int32 x4 each, any chunk
 * inside one relation, chunks of a relation contiguous: chunk_ptr [R+1]).  One workgroup per chunk accumulates the
 * whole Hi x Ho tile; partials (workspace, fp32 [num_chunks, Hi, Ho]) are then added in chunk order: deterministic.
 * idx_a / idx_g may be NULL (row p itself); A2/na1 and G2/ng1 give each operand a second row source exactly as
 * in dn_rows_transform_bf16.  Supported: Hi == Ho in {64, 128, 256}.  out is fp32 or bf16.
 * colsum_of = 1 (A) or 2 (G) also returns out_colsum[r, :] = sum over relation r's rows of that operand (fp32 [R, H]):
 * the bias gradient, taken from the rows while they are staged (0 = off, out_colsum may be NULL).  dn_rows_wgrad_bf16 takes
 * (r + 1) << 8 on top: the column sums of relation r's rows ONLY (the other rows of out_colsum are returned as zeros) -- the
 * reference's bias sees one relation's rows (the self loop's, rgin.py:156-160), and summing the rest costs a sixth of the launch.
 * mask_a (may be NULL; needs A2 == NULL): A rows are first zeroed where mask_a[row, k] <= 0 (ReLU backward folded into the
 * staging); a_out (may be NULL; needs idx_a == NULL) receives those masked rows, so the elementwise pass disappears.
 * mask_a_bits (may be NULL; excludes mask_a, A2 and idx_a: the bit-masked operand is read in row order): the same mask as
 * a bit tensor, uint8 [rows of A, Hi/8], bit i of byte (row, c) = keep element (row, 8c + i) -- the bits1 / bits2 outputs of
 * dn_rows_chain2_bf16.
 * out_colsum_lp (may be NULL; needs colsum_of): a second copy of the column sums in out's element type ([R, H]): the bias
 * gradient in the parameter's dtype without a cast launch. */
size_t dn_rows_wgrad_workspace_bytes(int64_t num_chunks, int32_t Hi, int32_t Ho);
int dn_rows_wgrad_bf16(const void* A, const void* A2, int32_t na1, const int32_t* idx_a, const void* G,
                       const void* G2, int32_t ng1, const int32_t* idx_g, int32_t Hi, int32_t Ho, int64_t R,
                       const int32_t* chunks, int64_t num_chunks, const int32_t* chunk_ptr, void* out,
                       int32_t out_is_f32, int32_t colsum_of, float* out_colsum, const void* mask_a, void* a_out,
                       const void* mask_a_bits, void* out_colsum_lp, float act_slope, void* workspace, size_t workspace_bytes,
                       dn_stream_t stream);

/* Several weight gradients in ONE launch + ONE reduce (round 6): the jobs' relations are numbered through -- job k owns relations
 * [first_rel_k, first_rel_{k+1}) of R -- and their rows lie end to end in one virtual row space (job k's row p is virtual row row0_k
 * + p), so ONE chunk table (dn_row_tables_build_i32 over the virtual relation offsets) and one chunk_ptr cover them; operands,
 * indices, colsum_of, mask_a_bits and act_slope per job as in dn_rows_wgrad_bf16 (no mask_a / a_out here).  out [R][H][H],
 * out_colsum [R][H] fp32 (zeros for a job with colsum_of = 0), out_colsum_lp [R][H] in out's type or NULL.  H = 64 / 128: the
 * widths at which a layer's three weight-gradient launches and their reduces are launch latency (BASELINE config 3: the whole
 * backward of an RGIN layer then needs one weight-gradient launch, subgraph_isomorphism/models/rgin.py:50-67,102-160).
 * H = 256 (rows_wgrad_ls_multi_kernel): a job is gathered on both sides (idx_a and idx_g; second sources allowed), or in row order
 * on both, with or without mask_a_bits; colsum_of may carry (relation + 1) << 8 -- a relation of THAT job, counted from its
 * first_rel -- as in dn_rows_wgrad_bf16; any chunk table over the virtual rows serves (the host side gives the row-order jobs
 * chunks of half the rows: ops.wide_layer_chunks).  One launch for a layer's three gradients pays on small batches only -- an
 * eighth of BASELINE config 5, one rank's share at 8 GPUs: 175 -> 150 us -- where the fixed costs of a launch count; every one of
 * them is HBM-bound. */
typedef struct dn_wgrad_job {
    const void *A, *A2;
    const int32_t* idx_a;
    const void *G, *G2;
    const int32_t* idx_g;
    const void* mask_a_bits;
    int32_t na1, ng1, colsum_of, first_rel, row0;
    float act_slope;
} dn_wgrad_job;
int dn_rows_wgrad_multi_bf16(const dn_wgrad_job* jobs, int32_t num_jobs, int32_t H, int64_t R, const int32_t* chunks, int64_t num_chunks,
                             const int32_t* chunk_ptr, void* out, int32_t out_is_f32, float* out_colsum, void* out_colsum_lp,
                             void* workspace, size_t workspace_bytes, dn_stream_t stream);
/* The same for fp32 rows on the 3-term bf16 split (dn_rows_wgrad_f32 with precision 0): A / A2 / G / G2 of a job are float tensors;
 * a job's mask_a_bits, when not NULL, points to a FLOAT tensor [rows of A, H] -- the saved activation, dn_rows_wgrad_f32's mask_a:
 * A rows are kept where it is > 0 and multiplied by act_slope elsewhere (needs A2 == NULL and idx_a == NULL).  out [R][H][H] and
 * out_colsum [R][H] fp32.  H = 64 / 128.  With this the
 * backward of an fp32 RGIN layer at the reference's default width (config.py:456-461) takes one weight-gradient launch and one
 * reduce instead of three of each (rgin.py:102-160's three parameter groups: weight / loop_weight + bias, mlp[0], mlp[2]). */
int dn_rows_wgrad_multi_f32(const dn_wgrad_job* jobs, int32_t num_jobs, int32_t H, int64_t R, const int32_t* chunks, int64_t num_chunks,
                            const int32_t* chunk_ptr, float* out, float* out_colsum, void* workspace, size_t workspace_bytes,
                            dn_stream_t stream);

/* Two dense layers in ONE pass over fp32 rows (3-term bf16 split; H = 64 / 128):
 *   X0 = mask0 ? keep_or_scale(X, mask0) : X;   Y1 = keep_or_scale?(epi1(X0 @ W1n^T), mask1);   Y2 = epi2(Y1 @ W2n^T)
 * epi = (+ bias) then the optional activation (relu flag, act_slope as in dn_rows_transform_f32); mask0 / mask1 (may be NULL) are
 * float tensors [N, H]: elements are kept where the mask is > 0 and multiplied by act_slope elsewhere.  w_kn bit i: weight i is given
 * [in][out] (the parameter of the OTHER direction as stored) instead of [out][in].  Forward of the reference MLP
 * (subgraph_isomorphism/models/rgin.py:50-57: Linear, act, Linear + the layer's activation :147-151) and -- with mask0 = the saved
 * output, W1n = Linear 2's weight, mask1 = the saved hidden rows, W2n = Linear 1's weight, w_kn = 3 -- autograd's input-gradient
 * chain through both, each as one launch (the fp32 twin of dn_rows_chain2_bf16; its masks are the saved activations, not bits).
 * residual / Y2_plus (both or neither; float [N, H]): Y2_plus = Y2 + residual is written next to Y2 -- the representation net's
 * residual connection (rgin.py:243-245: `outputs[-1] + o`) without a launch of its own; Y2 stays the activation the backward masks by. */
int dn_rows_chain2_f32(const float* X, int32_t H, const float* W1n, const float* b1, int32_t relu1, const float* mask0,
                       const float* mask1, const float* W2n, const float* b2, int32_t relu2, int64_t N, float* Y1, float* Y2,
                       int32_t w_kn, float act_slope, const float* residual, float* Y2_plus, dn_stream_t stream);

/* Relation-wise transform of gathered rows on the matrix cores (bf16 in, fp32 acc, bf16 out):
 *   Y[p, n] = epi( sum_k Xcat[idx[p], k] * Wn[rel(p)][n][k] ),  epi = (+ bias[rel(p)][n]) then optional ReLU
 * Xcat is the virtual concatenation of X (rows [0, n1)) and X2 (rows n1, n1+1, ...): an index i >= n1 reads
 * X2[i - n1]; pass X2 = NULL with n1 = INT32_MAX for a single source.  idx == NULL means row p itself.
 * Wn[r] is [Ho][Hi] with k contiguous (i.e. W_r transposed for Y = X W_r).  Rows are relation-major and the caller
 * provides the tile table {rel, beg, end, 0} (int32 x4; a tile has at most 32 rows and lies inside one relation).
 * Replaces the reference's per-edge `weight.index_select(0, etype)` + `th.bmm` message function
 * (subgraph_isomorphism/models/rgin.py:102-120, rgcn.py:100-122), its self-loop matmul (rgin.py:141) and, in the
 * backward direction (X = grad rows, Wn = W_r), autograd's transposed product.  Ho == Hi in {64, 128, 256}.
 * mask_pos (may be NULL): after epi, elements where mask_pos[p, n] <= 0 are zeroed.
 * act_slope: the activation of `relu` / `mask_pos` generalised to leaky ReLU -- relu: max(v, 0) + act_slope * min(v, 0); mask_pos:
 * elements whose saved activation is <= 0 are multiplied by act_slope instead of zeroed.  0 = ReLU; 1 / 5.5 = the reference's
 * default `leaky_relu` (subgraph_isomorphism/utils/act.py:466, constants.py:10).  The same argument, with the same meaning for
 * their ReLU flags / masks / mask bits, is taken by dn_rows_wgrad_*, dn_rows_chain2_bf16 and dn_relu_bwd_*.
 * w_kn != 0 (Hi == 256: with idx != NULL or X2 == NULL only, DN_ERR_UNSUPPORTED otherwise; Hi == 64 / 128: round 5, transposed through
 * LDS at every change of relation): Wn[r] is stored [k][n] -- the layout
 * of the reference's `weight` parameter (rgin.py:61-67), so the forward pass needs no transposed copy of the weights. */
int dn_rows_transform_bf16(const void* X, const void* X2, int32_t n1, const int32_t* idx, int32_t Hi, int32_t Ho,
                           const void* Wn, const void* bias, int32_t relu, const void* mask_pos,
                           const int32_t* tiles, int64_t num_tiles, void* Y, int32_t w_kn, float act_slope, dn_stream_t stream);

/* Closing launch of the row-factorised message pass (bf16 in, fp32 acc, bf16 out):
 *   out[v, :] = X[v, :] @ Wn^T (+ bias)  +  sum_{k < num_slots} Scat[slots[v * num_slots + k], :]
 * i.e. the self-loop transform `th.matmul(node_feat, self.loop_weight)` + bias (subgraph_isomorphism/models/rgin.py:
 * 140-145, rgcn.py:168-182) fused with the per-node sum of the transformed rows (the reference's `fn.sum(msg, out)`
 * reduce, rgin.py:137 / rgcn.py:166) -- and, in the backward direction, the same for the input gradient.
 * Wn is [H][H] with k contiguous (w_kn = 0) or [k][n] as the parameter `loop_weight` stores it (w_kn = 1, round 5: no transposed
 * copy in front of the launch).  slots is an [N, num_slots] int32 table of row ids into Scat = S (rows [0, n1))
 * followed by S2 (row n1, n1+1, ...; S2 = NULL with n1 = INT32_MAX for none); negative ids are empty slots.  A node with
 * more than num_slots rows carries -2 in its last slot (dn_slot_table_build_i32) and is finished by dn_overflow_rows_add_bf16
 * right after this launch -- or, with list_ptr / list_rows (the lists the slot table was built from, may be NULL) and the builder's
 * filter (num_edge_rows, drop_beg, drop_end), INSIDE this launch (round 5): the node's thread walks its list and adds the rows behind
 * the first num_slots - 1 kept ones in fp32 before the one rounding.  For small batches, where a second launch costs more than the
 * walk; large ones keep dn_overflow_rows_add_bf16 (the walk stalled every other tile: + 130 us at config 5).  num_slots must be 6.
 * H in {64, 128, 256}.
 * Folded pre-aggregation (fold_info != NULL): the launch also sums the X rows it reads per SEGMENT -- the input row of a
 * collapsed relation (all nodes of a graph -> its dummy node: one row per graph whose input is the sum of the graph's rows,
 * the reference's per-edge messages of the dummy edge type, rgin.py:102-120 on dataset.py:1563-1603's dummy edges) -- so the
 * separate pass over X that dn_gather_segsum_bf16 would make disappears.  Segments must be CONTIGUOUS ascending row ranges;
 * the caller numbers the (segment, 32-row tile) pairs that share a row ("partial rows"), segment-major, so that a tile's pairs
 * are consecutive.  fold_info (dn_fold_tables_build_i32) holds one 12-word record per tile: 32 bytes = each row's partial row
 * minus the tile's first one (255: the row belongs to no segment), then {first partial row, number of partial rows, 0, 0}.
 * The launch writes the fp32 column sums of every pair to seg_part[pair * H ...] (one extra MFMA per wave with a 0/1 indicator
 * operand); dn_fold_tail_bf16 adds a segment's partial rows in tile order: deterministic. */
int dn_rows_selfsum_bf16(const void* X, int32_t H, const void* Wn, const void* bias, const void* S, const void* S2,
                         int32_t n1, const int32_t* slots, int32_t num_slots, int64_t N, void* out,
                         const int32_t* fold_info, float* seg_part, int32_t w_kn, const int32_t* list_ptr,
                         const int32_t* list_rows, int32_t num_edge_rows, int32_t drop_beg, int32_t drop_end, dn_stream_t stream);

/* Nodes with more rows than slots: out[v, :] += sum of the rows of v's list beyond the first num_slots - 1 kept ones, for every
 * node with overflow[v] != 0 (dn_slot_table_build_i32's byte per node).  The walk applies the table builder's filter (rows >= num_edge_rows
 * and rows in [drop_beg, drop_end) are not edge rows of this launch) in list order; fp32 sum added to the bf16 row, one rounding.
 * Runs after dn_rows_selfsum_bf16 on the same `out` (the remainder of the reference's fn.sum reduce for high in-degree nodes,
 * subgraph_isomorphism/models/rgin.py:137).  In the closing launch itself the walk -- dependent loads of a few nodes -- stalled
 * every other tile (measured: + 130 us per config-5 launch); as a launch of its own it is a screen of N ints and a few thousand
 * short lists.  H in {64, 128, 256}. */
int dn_overflow_rows_add_bf16(const void* S, int32_t H, const uint8_t* overflow, int32_t num_slots, int64_t N, const int32_t* list_ptr,
                              const int32_t* list_rows, int32_t num_edge_rows, int32_t drop_beg, int32_t drop_end, void* out,
                              dn_stream_t stream);

/* The closing launch at H = 256 as a stream of 32-row units (csrc/dn_close.hip; same result as dn_rows_selfsum_bf16 +
 * dn_overflow_rows_add_bf16, with no slot limit and no second launch):
 *   out[v, :] = X[v, :] @ W_loop (+ bias)  +  sum over the kept rows p of v's list of S[p, :]
 * (the reference's `fn.sum(msg, out)` reduce + `th.matmul(node_feat, self.loop_weight)` + bias,
 * subgraph_isomorphism/models/rgin.py:137-146 / rgcn.py:166-182; backward direction: the same for the input gradient).
 *
 * dn_close_units_build_i32 (one-shot index build per batch and direction, three launches, no host synchronisation; the
 * bookkeeping dgl.batch + update_all do per step, subgraph_isomorphism/dataset.py:1605-1611): nodes are cut into num_tiles
 * tiles of at most 32 -- tile t = nodes [tile_ptr[t], tile_ptr[t+1]) (device, [num_tiles + 1]), or the 32-node windows when
 * tile_ptr == NULL (num_tiles = ceil(N / 32)).  For every tile the DISTINCT kept rows of its nodes' lists (list_ptr [N+1] /
 * list_rows as in dn_slot_table_build_i32, same num_edge_rows / drop range / drop_enable filter), sorted by row, are written to
 * ent_row with a 32-bit membership mask in ent_mask (bit i: node tile_ptr[t] + i adds the row; a row that ONE node lists twice
 * stays a second entry), from the offset list_ptr[tile_ptr[t]] onwards -- so ent_row / ent_mask need num_list_entries elements.
 * units [unit_capacity][4] int32 records {flags, beg, end, aux} in WORKGROUP-MAJOR order (its records are
 * units[unit_ptr[w] .. unit_ptr[w+1])).  xcd_order = 0: workgroup w of num_wg takes tiles w, w + num_wg, ... (one front walking the
 * batch upwards); xcd_order = 1 (round 5; num_wg a multiple of 8): workgroup w = 8 j + x -- dispatched to XCD x -- takes the tiles
 * of the x-th EIGHTH of the batch, [floor(x T / 8), floor((x + 1) T / 8)), DOWNWARDS from its end: end - 1 - (j + (num_wg / 8) n).
 * The transform launch in front of the closing launch walks the eighth of XCD x upwards (dn_sweep_tables_build_i32), so the closing
 * launch begins on the rows the transform wrote and gathered last (that XCD's L2, the Infinity Cache): -2 % on the pair at config 5.
 * Per tile one X record {rows << 8 | 2 if no entries,
 * first node, end node, tile} followed by one record per 32 entries {rows << 8 | 1 | 2 on the last, first entry, end entry, first
 * node}; with agg_units != 0 a workgroup's tiles are followed by 8 records {8, 0, 1, 0} (a gap) and one record {4 | 2, n, n', 0}
 * ({4 | 2 | 16, n, n', num_tiles} with xcd_order = 1) per 32 of its tiles (the n-th .. n'-th of them): the AGG units of the absorbed
 * fold below.
 * unit_capacity >= dn_close_units_capacity(num_tiles, num_list_entries, num_wg) = 3 num_tiles + num_list_entries / 32 + 1 + 9 num_wg.
 *
 * dn_fold_graph_tiles_build_i32 (one launch): the tiles of a batch of GRAPHS for the absorbed fold.  Segment j = the nodes
 * seg_nodes[seg_ptr[j] .. seg_ptr[j+1]) as in dn_fold_tables_build_i32; block j = [first node of segment j (0 for j = 0), first
 * node of segment j + 1 (N for the last)).  *dev_ok (device) stays non-zero when every segment is a non-empty contiguous
 * ascending run, the segments ascend, every block has at most 32 nodes and (add_idx != NULL, device [num_segments]: the output
 * row the segment's product is added to, i.e. the agg_idx of dn_rows_close_bf16) add_idx[j] lies inside block j -- the AGG unit
 * of tile j is a read-modify-write of out[add_idx[j]] by the workgroup that stored tile j, so a target in another block (the
 * dummy node in front of its graph, all dummy nodes at the end of the batch) keeps the partial rows + dn_fold_tail_bf16.  Then
 * tile_ptr [num_segments + 1] = the block starts (tile j = block j) and fold_info [num_segments][12] = per tile {32 bytes: 0 for a node of the segment, 255 otherwise; j; 1; 2; 0}.
 *
 * dn_fold_graph_tiles_multi_build_i32 (round 6; three launches, no read-back): the same for graphs of ANY size -- TU graphs are
 * not all within 32 nodes (graph_classification/data_processing/tu_data_processing.py:179-218 keeps whatever sizes the dataset
 * has; PROTEINS reaches 620).  The batch is cut into num_chunks CHUNKS at graph boundaries -- graph j (block start b0_j) belongs to
 * chunk floor(b0_j num_chunks / N) -- and every chunk into consecutive 32-node tiles that run ACROSS the graphs inside it (only a
 * chunk's last tile is partial: T <= N / 32 + num_chunks).  chunk_tile [num_chunks + 1] = a chunk's first tile (chunk_tile[num_chunks]
 * = T, known on the device only), chunk_graph [num_chunks + 1] = its first graph, tile_ptr [T + 1], fold_info [T][12] = per tile
 * {32 bytes: the local number of the node's segment in order of appearance, 255 outside every segment; the first such segment (=
 * graph = aux row); how many; bit 0 = the first one continues the previous tile's column sum | bit 1 = the last one ends in this
 * tile; 0}.  The tables must hold tile_capacity >= dn_fold_graph_tiles_multi_capacity(N, num_chunks) tiles; nothing is written for
 * an invalid batch (*dev_ok = 0).  Same validity test as above without the 32-node limit.  num_chunks < 16384.
 * dn_close_units_build_i32 with xcd_order = 2 / 3 takes these tables (tile_ptr, chunk_tile, chunk_graph, chunks_per_wg = K with
 * num_chunks = K num_wg; num_tiles = the capacity the tables were sized by) and deals the CHUNKS to the workgroups as xcd_order 0 / 1
 * deal tiles (K each), so that a graph's tiles stay in ONE stream: a segment's column sum continues from tile to tile in the
 * workgroup (fp32, one rounding) and the AGG unit's read-modify-write of out[add_idx[j]] stays inside the workgroup that stored
 * that row -- while the launch still sweeps the batch as a front of short runs.  A workgroup's stream = the tiles of its chunks,
 * the gap, and per chunk one AGG record {4 | 2 | 32, first graph, end graph, 0} per 32 graphs (the aux rows themselves).
 *
 * dn_rows_close_bf16: one persistent workgroup per entry of unit_ptr (launch num_wg = the builder's).  W: the self-loop weight,
 * w_kn = 0: [H][H] with k contiguous (W_loop transposed, as dn_rows_selfsum_bf16 takes it), w_kn = 1: [k][n] as the
 * parameter stores it (`loop_weight`, rgin.py:61) -- no transposed copy needed.  bias may be NULL.  fold_info + seg_part: the
 * folded pre-aggregation exactly as in dn_rows_selfsum_bf16 (fp32 partial rows, finished by dn_fold_tail_bf16).
 * fold_info + W_agg + aux + agg_idx (seg_part NULL; tables built with tile_ptr / agg_units from dn_fold_graph_tiles_build_i32):
 * the ABSORBED fold -- every segment lies inside one tile, so its column sum leaves as the bf16 row aux[segment] (kept by the
 * caller: the collapsed relation's operand of dn_rows_wgrad_bf16), and the workgroup's AGG units multiply its segments' aux rows
 * by W_agg (same layout flag as W) and add each product to out[agg_idx[segment]] -- what dn_fold_tail_bf16 does, inside this
 * launch.  A non-finite element of S turns its column of the whole tile into NaN (0 x Inf inside the selection product).
 * H must be 256. */
int64_t dn_close_units_capacity(int64_t num_tiles, int64_t num_list_entries, int32_t num_wg);
size_t dn_close_units_workspace_bytes(int64_t num_tiles, int32_t num_wg);
int dn_close_units_build_i32(int64_t N, int32_t num_edge_rows, int32_t num_wg, const int32_t* tile_ptr, int64_t num_tiles,
                             int32_t agg_units, int32_t xcd_order, const int32_t* list_ptr, const int32_t* list_rows, int64_t num_list_entries,
                             int32_t drop_beg, int32_t drop_end, const int32_t* drop_enable, int32_t* unit_ptr, int32_t* units,
                             int64_t unit_capacity, int32_t* ent_row, uint32_t* ent_mask, const int32_t* chunk_tile,
                             const int32_t* chunk_graph, int32_t chunks_per_wg, void* workspace, size_t workspace_bytes,
                             dn_stream_t stream);
int64_t dn_fold_graph_tiles_multi_capacity(int64_t N, int32_t num_chunks);
int dn_fold_graph_tiles_multi_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                        const int32_t* add_idx, int32_t num_chunks, int32_t* chunk_tile, int32_t* chunk_graph,
                                        int32_t* tile_ptr, int32_t* fold_info, int64_t tile_capacity, int32_t* dev_ok,
                                        dn_stream_t stream);
int dn_fold_graph_tiles_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                  const int32_t* add_idx, int32_t* tile_ptr, int32_t* fold_info, int32_t* dev_ok,
                                  dn_stream_t stream);
int dn_rows_close_bf16(const void* X, int32_t H, const void* W, int32_t w_kn, const void* bias, const void* S,
                       const int32_t* unit_ptr, const int32_t* units, int32_t num_wg, const int32_t* ent_row,
                       const uint32_t* ent_mask, int64_t N, void* out, const int32_t* fold_info, float* seg_part,
                       const void* W_agg, void* aux, const int32_t* agg_idx, dn_stream_t stream);

/* Tables of a folded pre-aggregation (one-shot index build, like dn_slot_table_build_i32): segment j = the rows
 * seg_nodes[seg_ptr[j] .. seg_ptr[j+1]) (dn_row_index_build_i32's aux_f_ptr/aux_f_idx or aux_b_ptr/aux_b_idx: the nodes of a graph
 * that feed / are fed by its dummy node, subgraph_isomorphism/dataset.py:1563-1603).  host_ok = 1 when every segment is a
 * non-empty contiguous ascending run of rows and the segments ascend -- otherwise the outputs are undefined and the caller keeps
 * the separate dn_gather_segsum pass.  Outputs: fold_info [ceil(N/32)][12] int32 as dn_rows_selfsum_bf16 reads it, part_ptr
 * [num_segments + 1] = the partial-row range of each segment (dn_fold_tail_bf16).  The number of partial rows is at most
 * 2 * num_segments + N / 32 + 1.  The call synchronises the stream. */
size_t dn_fold_tables_workspace_bytes(int64_t num_segments);
int dn_fold_tables_build_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                             int32_t* fold_info, int32_t* part_ptr, int32_t* host_ok, void* workspace, size_t workspace_bytes,
                             dn_stream_t stream);
/* The same build without the read-back: *dev_ok (device) ends up non-zero when the tables are valid, 0 otherwise. */
int dn_fold_tables_build_async_i32(int64_t N, int64_t num_segments, const int32_t* seg_ptr, const int32_t* seg_nodes,
                                   int32_t* fold_info, int32_t* part_ptr, int32_t* dev_ok, void* workspace,
                                   size_t workspace_bytes, dn_stream_t stream);

/* Tail of a folded pre-aggregation, one launch:  aux[j, :] = bf16( sum_{k in [part_ptr[j], part_ptr[j+1])} part[k, :] ) in k
 * order (kept by the caller: the collapsed relation's operand of dn_rows_wgrad_bf16), then the relation's transform of those
 * rows added to their one output row each:  out[idx[j], :] += aux[j, :] @ Wn^T   (Wn [H][H], k contiguous; idx distinct; the
 * fp32 product is added to the bf16 row and rounded once).  H in {64, 128, 256}.  w_kn != 0: Wn is stored [k][n]. */
int dn_fold_tail_bf16(const float* part, const int32_t* part_ptr, int64_t num_segments, int32_t H, const void* Wn,
                      const int32_t* idx, void* aux, void* out, int32_t w_kn, dn_stream_t stream);

/* Block-diagonal relation weights (regularizer "bdd", the reference CLI's default: subgraph_isomorphism/config.py:145-158):
 * blocks [R, B, si, so] (the layer's `weight` parameter viewed per relation, rgin.py:68-78) -> dense [R, B*si, B*so] with the B
 * blocks of a relation on its diagonal and zeros elsewhere, so that `bdd` runs on the same relation-transform kernels as `basis`
 * (the reference multiplies per block, rgin.py:114-120; the zeros add nothing).  dn_bdd_extract is the gradient: the diagonal
 * blocks of a dense [R, B*si, B*so] gradient.  elem_bytes = 2 (bf16) or 4 (f32).  One launch each. */
int dn_bdd_compose(const void* blocks, int64_t R, int32_t B, int32_t si, int32_t so, int32_t elem_bytes, void* dense,
                   dn_stream_t stream);
int dn_bdd_extract(const void* dense, int64_t R, int32_t B, int32_t si, int32_t so, int32_t elem_bytes, void* blocks,
                   dn_stream_t stream);

/* Two dense layers in one pass over the rows (bf16 in, fp32 acc, bf16 out):
 *   Y1 = epi1(m0(X) @ W1n^T),  Y2 = epi2(Y1 @ W2n^T)
 *   m0   = zero the elements of X whose bit in mask0_bits is clear (mask0_bits may be NULL)
 *   epi1 = (+ b1), ReLU if relu1, then zero where the bit in mask1_bits is clear (mask1_bits may be NULL)
 *   epi2 = (+ b2), ReLU if relu2
 *   bits1 / bits2 (may be NULL): bit tensors of Y1 / Y2, bit set where the element is > 0
 * Bit tensors are uint8 [N, H/8]: bit i of byte (p, c) belongs to element (p, 8c + i).
 * Forward of the reference's two-layer MLP after the aggregate (subgraph_isomorphism/models/rgin.py:50-57 followed by the
 * layer activation, :147-151: Linear-ReLU-Linear-ReLU), emitting the ReLU masks as bits, and -- with mask0 = bits of the
 * output activation, mask1 = bits of the hidden one -- the input-gradient chain of its backward.  W1n / W2n are [H][H]
 * with k contiguous (nn.Linear.weight for the forward); b1 / b2 may be NULL.  w_kn: bit 0 / bit 1 = W1n / W2n is stored
 * [k][n] instead -- the backward chain takes the Linear weights as they are (its first product is g @ W2, k = W2's rows), so no
 * transposed copies are made.  Y1 is written but never re-read.  H in {64, 128, 256}. */
int dn_rows_chain2_bf16(const void* X, int32_t H, const void* W1n, const void* b1, int32_t relu1, const void* mask0_bits,
                        const void* mask1_bits, const void* W2n, const void* b2, int32_t relu2, int64_t N, void* Y1, void* Y2,
                        void* bits1, void* bits2, int32_t w_kn, float act_slope, dn_stream_t stream);

/* ReLU backward of the post-aggregate MLP (act_func "relu": utils/act.py:463; applied at rgin.py:56,147-151):
 * out = (y > 0) ? g : 0 on bf16 tensors of `numel` elements (multiple of 8).  The same mask is available as the
 * `mask_pos` epilogue of dn_rows_transform_bf16 ([rows, Ho] saved activations), so a Linear's input gradient comes
 * out already masked for the ReLU in front of it. */
int dn_relu_bwd_bf16(const void* g, const void* y, void* out, int64_t numel, float act_slope, dn_stream_t stream);

/* fp32 twins of the three entry points above: same semantics and argument meaning, float tensors, float partials /
 * outputs -- the path that keeps the reference's fp32 numerics (outputs within 1e-4) while still avoiding its [E,H,H]
 * per-edge weight gather.  `precision` selects the arithmetic:
 *   0  3-term bf16 split on the fast matrix path: operands cut into hi = bf16(x), lo = bf16(x - hi) as they are staged,
 *      products evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with f32 accumulation (O(2^-16) relative
 *      per product, 1e-5-level agreement with the reference; 3/16 of the exact path's matrix time);
 *   1  exact f32 (v_mfma_f32_16x16x4_f32, bit for bit an fmaf chain): the checker.
 * dn_rows_transform_f32 reads the layer's parameters where they lie (rgin.py:61-67: weight [R, in, out], loop_weight
 * [in, out], h_bias [out]), so a step needs no concatenated / transposed weight copy and no padded bias matrix:
 *   W_loop / loop_rel  tiles of relation `loop_rel` use the matrix W_loop instead of Wn[loop_rel] (loop_rel < 0 or W_loop
 *                      NULL: every relation indexes Wn);
 *   bias_rel           >= 0: `bias` is ONE row [Ho] that only tiles of that relation add (the self-loop rows, rgin.py:146);
 *                      < 0: `bias` is [R, Ho], one row per relation (as dn_rows_transform_bf16);
 *   w_kn               1: every matrix is [k = in][n = out] (parameter layout), 0: [n][k]. */
int dn_rows_transform_f32(const float* X, const float* X2, int32_t n1, const int32_t* idx, int32_t Hi, int32_t Ho,
                          const float* Wn, const float* bias, int32_t relu, const float* mask_pos,
                          const int32_t* tiles, int64_t num_tiles, float* Y, int32_t precision, float act_slope,
                          const float* W_loop, int32_t loop_rel, int32_t bias_rel, int32_t w_kn, dn_stream_t stream);
int dn_rows_wgrad_f32(const float* A, const float* A2, int32_t na1, const int32_t* idx_a, const float* G,
                      const float* G2, int32_t ng1, const int32_t* idx_g, int32_t Hi, int32_t Ho, int64_t R,
                      const int32_t* chunks, int64_t num_chunks, const int32_t* chunk_ptr, float* out,
                      int32_t colsum_of, float* out_colsum, const float* mask_a, float* a_out, int32_t precision,
                      float act_slope, void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_relu_bwd_f32(const float* g, const float* y, float* out, int64_t numel, float act_slope, dn_stream_t stream);

/* Relation-grouped dense products for ANY widths (what PyG's RGCNConv computes with a Python loop over relations,
 * `for i in range(num_relations): out += h @ weight[i]`; call sites graph_classification/graph_neural_networks/models/
 * rgconv.py:17-18,96): rows p relation-major, pieces from dn_row_tables_build_i32 (step 64 for the tiles).
 *   dn_rows_gemm_*       Y[p,:] = A[p,:] @ Wr (+ bias[rel]),  Wr[k][n] = transposed ? W[rel][n][k] : W[rel][k][n]   (A [P,K], Y [P,N];
 *                        bias [R,N] in the storage type, may be NULL)
 *   dn_rows_wgrad_any_*  out[r] = sum_{p in relation r} A[p,:]^T G[p,:]   ([K,N] per relation; split-K chunks, partials added
 *                        in chunk order; workspace from dn_rows_wgrad_any_workspace_bytes); colsum_out (fp32 [R,K], may be NULL)
 *                        = sum_{p in relation r} A[p,:], the bias gradient of a Linear layer whose output gradient is A
 * fp32 accumulation, plain FMA tiles (the widths here are the ones the matrix-core kernels do not cover). */
int dn_rows_gemm_f32(const float* A, const float* W, const float* bias, int32_t K, int32_t N, int32_t transposed, const int32_t* tiles,
                     int64_t num_tiles, float* Y, dn_stream_t stream);
int dn_rows_gemm_bf16(const void* A, const void* W, const void* bias, int32_t K, int32_t N, int32_t transposed, const int32_t* tiles,
                      int64_t num_tiles, void* Y, dn_stream_t stream);
size_t dn_rows_wgrad_any_workspace_bytes(int64_t num_chunks, int32_t K, int32_t N);
int dn_rows_wgrad_any_f32(const float* A, const float* G, int32_t K, int32_t N, int64_t R, const int32_t* chunks, int64_t num_chunks,
                          const int32_t* chunk_ptr, float* out, float* colsum_out, void* workspace, size_t workspace_bytes,
                          dn_stream_t stream);
int dn_rows_wgrad_any_bf16(const void* A, const void* G, int32_t K, int32_t N, int64_t R, const int32_t* chunks, int64_t num_chunks,
                           const int32_t* chunk_ptr, void* out, float* colsum_out, void* workspace, size_t workspace_bytes,
                           dn_stream_t stream);

/* BatchNorm over the rows of [N, C] node features, training mode (batch statistics), optionally with the ReLU that follows it
 * -- the `BatchNorm1d, ReLU` pairs inside the MLPs of the GC models (graph_classification/graph_neural_networks/models/
 * gconv.py:187-194, rgconv.py:85-93).
 *   forward:   mean / var (biased) / rstd [C] over the rows;  Y = act((X - mean) * rstd * weight + bias), act = ReLU when relu != 0
 *              (weight / bias may be NULL)
 *   backward:  DY' = DY where Y > 0 (relu; Y is recomputed from X), else DY;  sum_dy [C] (= dbias), sum_dy_xhat [C] (= dweight)
 *              of DY';  DX = weight * rstd * (DY' - sum_dy / N - xhat * sum_dy_xhat / N)
 * fp32 statistics whatever the storage type; C a multiple of 4, <= 1024; deterministic.  running_mean / running_var (fp32 [C],
 * both or neither; NULL: skipped) receive torch's update r = (1 - momentum) r + momentum * new with the unbiased variance;
 * num_batches_tracked (int64 [1] on the device, may be NULL) is incremented by the same launch (BatchNorm1d's buffer of that name). */
size_t dn_batchnorm_rows_workspace_bytes(int64_t N, int32_t C);
int dn_batchnorm_rows_f32(const float* X, int64_t N, int32_t C, const float* weight, const float* bias, float eps, float* Y, float* mean,
                          float* var, float* rstd, float* running_mean, float* running_var, float momentum, int32_t relu,
                          int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_batchnorm_rows_bf16(const void* X, int64_t N, int32_t C, const float* weight, const float* bias, float eps, void* Y, float* mean,
                           float* var, float* rstd, float* running_mean, float* running_var, float momentum, int32_t relu,
                           int64_t* num_batches_tracked, void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_batchnorm_rows_bwd_f32(const float* DY, const float* X, int64_t N, int32_t C, const float* mean, const float* rstd,
                              const float* weight, const float* bias, int32_t relu, float* DX, float* sum_dy, float* sum_dy_xhat,
                              void* workspace, size_t workspace_bytes, dn_stream_t stream);
int dn_batchnorm_rows_bwd_bf16(const void* DY, const void* X, int64_t N, int32_t C, const float* mean, const float* rstd,
                               const float* weight, const float* bias, int32_t relu, void* DX, float* sum_dy, float* sum_dy_xhat,
                               void* workspace, size_t workspace_bytes, dn_stream_t stream);

/* RGCN degree normalisation.  Replaces RGCNLayer._node_init_func/_edge_init_func
 * (subgraph_isomorphism/models/rgcn.py:132-165): in_norm = 1/(in_deg+1) with self-loop else 1/in_deg
 * (0 for isolated), same for out; edge norm = in_norm[dst] ("in", mode 1) or
 * sqrt(out_norm[src]*in_norm[dst]) ("both", mode 2).  Any output may be NULL. */
int dn_edge_norm_f32(int32_t mode, int32_t self_loop, int64_t N, int64_t E, const int32_t* src,
                     const int32_t* dst, const int32_t* in_deg, const int32_t* out_deg, float* in_norm,
                     float* out_norm, float* edge_norm, dn_stream_t stream);
/* in_deg / out_deg by counting (graph.in_degrees() / out_degrees(), rgcn.py:134,144). */
int dn_degrees_i32(int64_t N, int64_t E, const int32_t* src, const int32_t* dst, int32_t* in_deg,
                   int32_t* out_deg, dn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DN_HIP_H */
