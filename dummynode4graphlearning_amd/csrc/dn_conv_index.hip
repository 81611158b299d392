// The per-batch index of the H = 256 bf16 conv path as ONE call (dn_conv_index_build_i32): what ops.RowIndex + ops.prepare_closing
// used to assemble from six library calls with the host in between -- the graph-local row index (dn_row_index_build_local_i32),
// the unit streams of both closing launches with the fold absorbed (dn_close_units_build_i32 over the graphs as tiles), the
// L2-blocked sweep orders of both transform launches (dn_sweep_tables_build_i32) and the split-K chunk table of the weight
// gradient (dn_row_tables_build_i32) -- is what dgl.batch + update_all pay per step in the reference
// (subgraph_isomorphism/dataset.py:1605-1611).  Everything is queued back to back on one arena BEFORE the one read-back: the
// table builders take the counts the host does not know yet (edge rows, the folded relation's row range and id) from the words
// the row index's last launch leaves on the device (ril_plan), and are sized for the one case they serve -- every graph (with its dummy node) inside one
// 32-node tile, so the tiles are the batch's G graphs.  Where that does not hold (go = 0 for a direction: a graph over 32 nodes, no
// dummy relation, a batch the local builder rejects) the queued builders do nothing and the caller builds the tables of that batch
// the old way from the row index this call still returns.  Only the chunk table waits for the read-back: its length (the split-K
// partials are sized by it) follows from the relation sizes.
#include "dn_common.h"
#include "dn_internal.h"
#include "../../include/dn_hip.h"

namespace {

struct Arena {
    size_t ril, close;
};

// (the closing tables' tiles: the batch's G graphs, or -- a graph over 32 nodes -- the chunked tiles, at most tile_capacity)
bool arena_sizes(int64_t G, int64_t N, int64_t R, int64_t E, int32_t num_wg, int64_t tile_capacity, Arena& a) {
    a.ril = dn_row_index_local_workspace_bytes(G, N, R, E);
    a.close = dn_close_units_workspace_bytes(G > tile_capacity ? G : tile_capacity, num_wg);
    return a.ril != 0 && a.close != 0;
}

}  // namespace

extern "C" {

size_t dn_conv_index_workspace_bytes(int64_t G, int64_t N, int64_t R, int64_t E, int32_t num_wg, int64_t tile_capacity) {
    Arena a;
    if (!arena_sizes(G, N, R, E, num_wg, tile_capacity, a)) return 0;
    return dn_align_up(a.ril, 256) + 2 * dn_align_up(a.close, 256) + 256;
}

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
                            size_t workspace_bytes, dn_stream_t stream) {
    DN_REQUIRE(G >= 1 && N >= 1 && R >= 1 && E >= 0 && num_wg >= 1, "dn_conv_index_build: bad sizes");
    DN_REQUIRE(host_counts && host_rel_ptr && host_modes && host_status && host_absorb && host_plan && rel_ptr_dev && workspace,
               "dn_conv_index_build: NULL pointer");
    DN_REQUIRE(unit_ptr_f && units_f && ent_row_f && ent_mask_f && unit_ptr_b && units_b && ent_row_b && ent_mask_b,
               "dn_conv_index_build: NULL pointer");
    DN_REQUIRE(sweep_tiles_per_wg == 0 || (sweep_f && sweep_b), "dn_conv_index_build: NULL pointer");
    DN_REQUIRE(wgrad_workgroups >= 1 && wgrad_max_chunk_rows >= 256 && wgrad_max_chunk_rows % 64 == 0 && chunk_table && chunk_ptr,
               "dn_conv_index_build: bad chunk-table arguments");
    // chunks_per_wg = 0: no chunked form (batches with a graph over 32 nodes are then left to the caller, as in round 5)
    const bool chunked = chunks_per_wg > 0;
    DN_REQUIRE(!chunked || (chunk_tile_f && chunk_graph_f && tile_ptr_mf && fold_info_mf && chunk_tile_b && chunk_graph_b && tile_ptr_mb &&
                            fold_info_mb && tile_capacity >= dn_fold_graph_tiles_multi_capacity(N, chunks_per_wg * num_wg)),
               "dn_conv_index_build: the chunked form needs its eight tables, sized by dn_fold_graph_tiles_multi_capacity");
    if (!chunked) tile_capacity = 0;
    DN_REQUIRE(unit_capacity >= dn_close_units_capacity(G > tile_capacity ? G : tile_capacity, E + N, num_wg), "dn_conv_index_build: unit table too small");
    Arena a;
    DN_REQUIRE(arena_sizes(G, N, R, E, num_wg, tile_capacity, a), "dn_conv_index_build: bad sizes");
    DN_REQUIRE(workspace_bytes >= dn_conv_index_workspace_bytes(G, N, R, E, num_wg, tile_capacity), "dn_conv_index_build: workspace too small");
    DN_REQUIRE(reinterpret_cast<uintptr_t>(workspace) % 256 == 0, "dn_conv_index_build: unaligned workspace");
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    char* ws_close = ws + dn_align_up(a.ril, 256);
    int32_t* meta = nullptr;
    int rc = dn_internal::ril_queue(G, N, R, E, node_ptr, edge_ptr, src, dst, etype, self_loop, edge_frac, row_in, row_out, aux_f_ptr,
                                    aux_f_idx, aux_b_ptr, aux_b_idx, dst_ptr, dst_rows, src_ptr, src_rows, rel_ptr_dev, tile_ptr_f,
                                    fold_info_f, tile_ptr_b, fold_info_b, true, true, ws, a.ril, &meta, st);
    if (rc != DN_OK) return rc;
    const int32_t* plan = meta + 5 + 2 * R + 4;
    // a graph over 32 nodes (go = 2 on the device): the chunked tiles of both directions -- the verdict launch has done the validity
    // test, the aux lists are the segments, the verdict words double as the builders' "valid" words
    if (chunked) {
        int32_t* verdict = meta + 5 + 2 * R + 2;
        const dn_internal::FoldMultiDir fm[2] = {
            {aux_f_ptr, aux_f_idx, nullptr, chunk_tile_f, chunk_graph_f, tile_ptr_mf, fold_info_mf, verdict, plan + 3},
            {aux_b_ptr, aux_b_idx, nullptr, chunk_tile_b, chunk_graph_b, tile_ptr_mb, fold_info_mb, verdict + 1, plan + 7}};
        rc = dn_internal::fold_multi_queue(N, G, 2, fm, chunks_per_wg * num_wg, tile_capacity, false, st);
        if (rc != DN_OK) return rc;
    }
    // the unit streams of both directions in one set of launches, AGG units appended; edge rows / dropped range / go from the device:
    // go = 1: tiles = the graphs, go = 2: the chunked tiles
    const dn_internal::CloseUnitsDir dirs[2] = {
        {tile_ptr_f, dst_ptr, dst_rows, 0, 0, 0, nullptr, plan, unit_ptr_f, units_f, ent_row_f, ent_mask_f, chunk_tile_f, chunk_graph_f,
         chunks_per_wg, tile_ptr_mf},
        {tile_ptr_b, src_ptr, src_rows, 0, 0, 0, nullptr, plan + 4, unit_ptr_b, units_b, ent_row_b, ent_mask_b, chunk_tile_b, chunk_graph_b,
         chunks_per_wg, tile_ptr_mb}};
    rc = dn_internal::close_units_queue(N, num_wg, G, 1, close_xcd_order, E + N, unit_capacity, 2, dirs, ws_close, 2 * dn_align_up(a.close, 256), st,
                                        chunked ? 2 + close_xcd_order : 0, tile_capacity);
    if (rc != DN_OK) return rc;
    if (sweep_tiles_per_wg > 0) {                                             // and both sweep orders in one launch
        const uint64_t no_mask[2] = {0, 0};
        const int32_t* dyn[2] = {plan + 8, plan + 10};
        int32_t* tables[2] = {sweep_f, sweep_b};
        int32_t* info[2] = {meta + 5 + 2 * R + 4 + dn_internal::kRilPlanWords + 1, meta + 5 + 2 * R + 4 + dn_internal::kRilPlanWords + 3};
        rc = dn_internal::sweep_tables_queue((int32_t)R, rel_ptr_dev, row_in, row_out, N, sweep_wg_per_group, sweep_tiles_per_wg, 2,
                                             no_mask, dyn, tables, info, st);
        if (rc != DN_OK) return rc;
    }
    int32_t h_meta[5 + 2 * 64 + 4 + dn_internal::kRilPlanWords + 5];
    DN_REQUIRE(R <= 64, "dn_conv_index_build: more than 64 relations");
    const size_t words = (size_t)(5 + 2 * R + 4 + dn_internal::kRilPlanWords + 5);   // (+ the ticket and the sweep builder's 2 x 2 words)
    DN_CHECK_HIP(hipMemcpyAsync(h_meta, meta, sizeof(int32_t) * words, hipMemcpyDeviceToHost, st));
    DN_CHECK_HIP(hipStreamSynchronize(st));                                   // the one read-back
    dn_internal::ril_unpack(h_meta, R, host_counts, host_rel_ptr, host_modes, host_status, host_absorb);
    const int32_t* hp = h_meta + 5 + 2 * R + 4;
    host_plan[0] = hp[3]; host_plan[1] = hp[7]; host_plan[2] = 0; host_plan[3] = 0; host_plan[4] = 0; host_plan[5] = 0;
    if (!chunked)                                                             // (go = 2 without the chunked tables: nothing was built)
        for (int d = 0; d < 2; ++d)
            if (host_plan[d] == 2) host_plan[d] = 0;
    if (sweep_tiles_per_wg > 0) {                                             // slots per workgroup of each order as the builder laid it out
        const int32_t* si = hp + dn_internal::kRilPlanWords + 1;             // {0 sweep | 1 plain | 2 does not fit, slots of the fullest group} x 2
        for (int d = 0; d < 2; ++d)
            if (host_plan[d]) host_plan[4 + d] = si[2 * d] == 0 ? (si[2 * d + 1] > 0 ? si[2 * d + 1] : 1) : (si[2 * d] == 1 ? sweep_tiles_per_wg : -1);
    }
    if (*host_status != 0) return DN_OK;                                      // (the caller runs the general builder)
    // split-K chunk table of the weight gradient over ALL rows (the self loop as relation R): the smallest multiple of 64 rows
    // (>= 256) for which every relation's chunks -- each ends in a partial one -- fit one round of the workgroups
    const int64_t P = host_counts[0], P_all = P + (self_loop ? N : 0);
    const int Rall = (int)R + (self_loop ? 1 : 0);
    int64_t sizes[65];
    int64_t total = 0;
    for (int r = 0; r < (int)R; ++r) { sizes[r] = (int64_t)host_rel_ptr[r + 1] - host_rel_ptr[r]; total += sizes[r]; }
    if (self_loop) { sizes[R] = N; total += N; }
    int64_t c = 256;
    if (total > 0) {
        c = dn_cdiv(dn_cdiv(total, (int64_t)wgrad_workgroups), (int64_t)64) * 64;
        if (c < 256) c = 256;
        for (;;) {
            if (c >= wgrad_max_chunk_rows) break;
            int64_t n = 0;
            for (int r = 0; r < Rall; ++r) n += sizes[r] > 0 ? dn_cdiv(sizes[r], c) : 0;
            if (n <= wgrad_workgroups) break;
            c += 64;
        }
        if (c > wgrad_max_chunk_rows) c = wgrad_max_chunk_rows;
    }
    const int64_t M = P_all / c + Rall + 1;
    DN_REQUIRE(M <= chunk_capacity, "dn_conv_index_build: chunk table too small (%lld < %lld)", (long long)chunk_capacity, (long long)M);
    rc = dn_row_tables_build_i32(Rall, rel_ptr_dev, (int32_t)c, M, chunk_table, chunk_ptr, 0, stream);
    if (rc != DN_OK) return rc;
    host_plan[2] = (int32_t)c; host_plan[3] = (int32_t)M;
    return DN_OK;
}

}  // extern "C"
