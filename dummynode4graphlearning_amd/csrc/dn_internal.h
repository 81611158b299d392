// Launchers shared between translation units of libdn_hip.so (C++ linkage, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dn_internal {

// dn_rel_ring.hip: dn_rows_transform_bf16 at Hi == Ho == 256 (persistent workgroups, LDS-DMA ring, wave-specialised).
// tiles_per_wg <= 0: contiguous tile ranges, one workgroup per CU; > 0: the table is laid out [workgroup][tiles_per_wg].
// w_kn != 0: Wn[r] is stored [k][n] (the parameter's own layout) instead of [n][k].  slope: leaky-ReLU slope of the epilogue / mask (0: ReLU).
int launch_transform_ring256(const void* X, const void* X2, int32_t n1, const int32_t* idx, const void* Wn, const void* bias,
                             int32_t relu, int32_t nt_store, const void* mask_pos, const int32_t* tiles, int64_t num_tiles,
                             int64_t tiles_per_wg, void* Y, int32_t w_kn, float slope, hipStream_t st);

// dn_chain2.hip: dn_rows_chain2_bf16 at H == 256 (LDS-DMA ring, 8 waves, results straight from the accumulators).  Serves the
// forward (no masks; both sign-bit outputs or neither) and the backward (both masks, no sign-bit outputs) forms.
bool chain2_ring_supported(bool has_mask0, bool has_mask1, bool has_bits1, bool has_bits2);
int launch_chain2_ring256(const void* X, const void* W1n, const void* b1, const void* W2n, const void* b2, int32_t flags,
                          const void* mask0, const void* mask1, int64_t N, void* Y1, void* Y2, void* bits1, void* bits2,
                          float slope, hipStream_t st);

// ---- the per-batch index as ONE call (dn_conv_index.hip: dn_conv_index_build_i32) -- the builders' launches without their host sides
constexpr int kRilPlanWords = 14;      // ril_plan's 12 words behind the 5 + 2 R + 4 meta words of the graph-local row index, then the
                                       // largest graph's nodes and edges (ril_fill_kernel)

// dn_index_local.hip
int ril_queue(int64_t G, int64_t N, int64_t R, int64_t E, const int32_t* node_ptr, const int32_t* edge_ptr, const int32_t* src,
              const int32_t* dst, const int32_t* etype, int32_t self_loop, float edge_frac, int32_t* row_in, int32_t* row_out,
              int32_t* aux_f_ptr, int32_t* aux_f_idx, int32_t* aux_b_ptr, int32_t* aux_b_idx, int32_t* dst_ptr, int32_t* dst_rows,
              int32_t* src_ptr, int32_t* src_rows, int32_t* rel_ptr_dev, int32_t* tile_ptr_f, int32_t* fold_info_f,
              int32_t* tile_ptr_b, int32_t* fold_info_b, bool verdicts, bool plan, void* workspace, size_t workspace_bytes,
              int32_t** meta_dev, hipStream_t st);
void ril_unpack(const int32_t* h_meta, int64_t R, int64_t* host_counts, int32_t* host_rel_ptr, int32_t* host_modes,
                int32_t* host_status, int32_t* host_absorb);

// dn_close.hip
struct CloseUnitsDir {                 // one direction's arguments of dn_close_units_build_i32 (dyn: see close_units_queue)
    const int32_t *tile_ptr, *list_ptr, *list_rows;
    int32_t num_edge_rows, drop_beg, drop_end;
    const int32_t *drop_enable, *dyn;
    int32_t *unit_ptr, *units, *ent_row;
    uint32_t* ent_mask;
    const int32_t *chunk_tile = nullptr, *chunk_graph = nullptr;   // orders 2 / 3 (graphs of any size): first tile / first graph of every
    int32_t chunks_per_wg = 0;                                     //   chunk [K num_wg + 1]; num_tiles is then a BOUND (chunk_tile[K num_wg] exist)
    const int32_t* tile_ptr_alt = nullptr;                         // close_units_queue's alternative form: the chunked tables' tile_ptr
};
// (dn_fold_graph_tiles_multi_build_i32 without its argument checks, for one or both directions of a batch in ONE set of launches;
//  gate != NULL: a device word that must be 2 -- ril_plan's "chunked tiles" -- for the launches to do anything:
//  dn_conv_index_build_i32 queues them before it knows whether the batch wants them; with_valid = false: the validity launch is left
//  out, dev_ok already holds the verdict)
struct FoldMultiDir {
    const int32_t *seg_ptr, *seg_nodes, *add_idx;
    int32_t *chunk_tile, *chunk_graph, *tile_ptr, *fold_info, *dev_ok;
    const int32_t* gate;
};
int fold_multi_queue(int64_t N, int64_t num_segments, int nd, const FoldMultiDir* dirs, int32_t num_chunks, int64_t tile_capacity,
                     bool with_valid, hipStream_t st);
int close_units_queue(int64_t N, int32_t num_wg, int64_t num_tiles, int32_t agg_units, int32_t xcd_order, int64_t num_list_entries,
                      int64_t unit_capacity, int nd, const CloseUnitsDir* dirs, void* workspace, size_t workspace_bytes, hipStream_t st,
                      int32_t alt_order, int64_t alt_num_tiles);

// dn_index.hip
int sweep_tables_queue(int32_t num_rels, const int32_t* rel_ptr, const int32_t* row_in, const int32_t* row_out, int64_t num_nodes,
                       int32_t workgroups_per_group, int32_t tiles_per_workgroup, int nd, const uint64_t* skip_mask,
                       const int32_t* const* dyn, int32_t* const* table, int32_t* const* info, hipStream_t st);

}  // namespace dn_internal
