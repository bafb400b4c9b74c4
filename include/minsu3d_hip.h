/*
 * minsu3d_hip.h -- C ABI of libminsu3d_hip.so (hand-written gfx950 HIP kernels).
 *
 * Drop-in boundary for the reference's native layer: every entry point below replaces one
 * function a `COMMON_OPS` / MinkowskiEngine binding calls on the hot path.  Plain pointers and
 * sizes only (no torch types).  All pointers are DEVICE pointers unless a parameter is marked
 * [host].  `stream` is a hipStream_t (pass NULL for the default stream).  Every function
 * returns 0 on success or a non-zero hipError_t / MS3D_E_* code; nothing prints or exits
 * (the reference's launchers fprintf+exit(-1): bfs_cluster.cu:82-86).
 *
 * Citations are file:line into /root/reference/minsu3d/common_ops/src unless noted.
 */
#ifndef MINSU3D_HIP_H
#define MINSU3D_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS3D_E_WORKSPACE 10001 /* workspace too small */
#define MS3D_E_UNSUPPORTED 10002 /* shape outside what the kernels support */
#define MS3D_E_INTERNAL 10003 /* an invariant of the algorithm does not hold (a bug): results are not to be used */

typedef void *ms3d_stream_t; /* hipStream_t */

const char *ms3d_version(void);

/* ---- ball query: replaces ballquery_batch_p_cuda, bfs_cluster/bfs_cluster.h:16, kernel
 * bfs_cluster.cu:15-60.  Same arguments plus the scene count, a workspace and the stream.
 * Canonical output: start_len[i] = (exclusive prefix sum of len, len); lists in
 * ascending index, self included, len = min(hits, 1000).  Entries at positions >= n*meanActive
 * are not written (bfs_cluster.cu:51-58) and the total is returned through *n_active [host]
 * so the wrapper's retry loop (functions/common_ops.py:31-38) behaves identically.
 * Uniform hash grid (cell = 1.01*radius) + count / scan / fill; no per-thread 1000-int stack. */
size_t ms3d_ballquery_workspace_bytes(int n);
int ms3d_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz, const uint8_t *batch_idxs,
                           const int *batch_offsets, int n_scenes, int max_scene_points, int *idx,
                           int *start_len, int *n_active /*[host]*/, int *capped /*[host] or NULL: 1 if any list hit 1000*/,
                           void *workspace, size_t workspace_bytes, ms3d_stream_t stream);

/* ---- BFS clustering: replaces pg_bfs_cluster / sg_bfs_cluster, bfs_cluster/bfs_cluster.h:18-19
 * (host C++ bfs_cluster.cpp:28-187).  Runs on the DEVICE (the reference copies the ball-query
 * result to the host and runs a serial FIFO BFS); output order is identical to the serial BFS:
 * clusters by ascending seed, members in FIFO visit order, including the directed case when
 * the 1000-neighbour cap bites.  cluster_idxs has capacity [N,2], cluster_offsets [N+1];
 * counts[0] = nCluster, counts[1] = sumNPoint are returned to the [host]. */
size_t ms3d_bfs_workspace_bytes(int N);
/* capped_hint: what ms3d_ballquery_batch_p reported for this graph: 0 = no list reached 1000 -> symmetric graph with
 * ascending lists that contain the point itself; 1 = some did (directed: a capped list holds the point's lowest-index
 * neighbours).  -1 = NOBODY VOUCHES for the graph -- any adjacency lists, as the reference's host BFS accepts them
 * (bfs_cluster.cpp:28-54): nothing is assumed (no symmetry, no order inside a list, lists laid out in any order, any
 * length), at the price of a validation pass and two host syncs.  Then MS3D_E_UNSUPPORTED is returned -- instead of a
 * wrong answer or an out-of-bounds read -- for a list header outside [0, n_edges], a target outside [0, N), or a list that
 * names one neighbour twice (the serial loop skips the second mention; the parallel claims cannot). */
int ms3d_pg_bfs_cluster(const int16_t *semantic_label, const int *ball_query_idxs, long n_edges /* = nActive */,
                        const int *start_len, int N, int threshold, int capped_hint, int *cluster_idxs, int *cluster_offsets, int *counts /*[host,2]*/,
                        void *workspace, size_t workspace_bytes, ms3d_stream_t stream);
int ms3d_sg_bfs_cluster(const float *class_numpoint_mean /*[host]*/, const int *ball_query_idxs, long n_edges,
                        const int *start_len, int N, float threshold, int capped_hint, int class_id, int *cluster_idxs,
                        int *cluster_offsets, int *counts /*[host,2]*/, void *workspace, size_t workspace_bytes,
                        ms3d_stream_t stream);

/* all SoftGroup classes in one call: group_of_point u8[N] (class*B + scene, non-decreasing), thr_per_group f32[G]
 * (device); output clusters are class-major = the reference's per-class concatenation (model/softgroup.py:43-83) */
int ms3d_sg_bfs_cluster_batched(const uint8_t *group_of_point, const float *thr_per_group, const int *ball_query_idxs,
                                long n_edges, const int *start_len, int N, int capped_hint, int *cluster_idxs,
                                int *cluster_offsets,
                                int *counts /*[host,2]*/, void *workspace, size_t workspace_bytes, ms3d_stream_t stream);

/* ---- HAIS: replaces hierarchical_aggregation, hierarchical_aggregation/hierarchical_aggregation.h:14-28
 * (host .cpp:8-184 + .cu:20-204) AND the kept/primary merge of functions/hais_ops.py:55-73: the output is the final
 * (cluster_idxs, cluster_offsets) pair -- kept fragments first, then primaries with their absorbed fragments
 * (ascending fragment index).  cluster_idxs capacity [2N,2], cluster_offsets [N+1]; counts -> (nCluster, rows). */
size_t ms3d_hais_workspace_bytes(int N, int nclass);
int ms3d_hierarchical_aggregation(const int16_t *semantic_label, const float *coord_shift, const uint8_t *batch_idxs,
                                  const int *ball_query_idxs, long n_edges, const int *start_len, int N,
                                  int capped_hint /* as for ms3d_pg_bfs_cluster */,
                                  int using_set_aggr, const float *point_num_avg /*[host]*/,
                                  const float *radius_avg /*[host]*/, int nclass, int *cluster_idxs,
                                  int *cluster_offsets, int *counts /*[host,2]*/, void *workspace,
                                  size_t workspace_bytes, ms3d_stream_t stream);

/* The same operator with the reference's OWN output contract (hierarchical_aggregation/hierarchical_aggregation.h:14-28,
 * .cpp:105-184): kept fragments, primaries and -- with set aggregation -- all fragments and the primaries with their
 * absorbed fragments, each as idxs [rows,2], offsets [n+1], centers [n,5] (x, y, z, class, scene).  Used by the
 * `COMMON_OPS.hierarchical_aggregation` shim (minsu3d_amd/dropin/COMMON_OPS.py) so that the reference's wrapper
 * (functions/hais_ops.py:6-79) runs unchanged.  Capacities: N rows / N+1 offsets / 5N floats per list; post_idxs rows
 * beyond post_offsets[n_primary] are zero.  counts [host,8] = n_kept, kept rows, n_primary, post rows, n_fragment,
 * fragment rows, primary rows, 0. */
int ms3d_hierarchical_aggregation_parts(const int16_t *semantic_label, const float *coord_shift, const uint8_t *batch_idxs,
                                        const int *ball_query_idxs, long n_edges, const int *start_len, int N,
                                        int capped_hint, int using_set_aggr, const float *point_num_avg /*[host]*/,
                                        const float *radius_avg /*[host]*/, int nclass, int *kept_idxs, int *kept_offsets,
                                        float *kept_centers, int *prim_idxs, int *prim_offsets, float *prim_centers,
                                        int *frag_idxs, int *frag_offsets, float *frag_centers, int *post_idxs,
                                        int *post_offsets, int *counts /*[host,8]*/, void *workspace,
                                        size_t workspace_bytes, ms3d_stream_t stream);

/* ---- segment ops: replace sec_mean_cuda / sec_min_cuda / sec_max_cuda, sec_mean/sec_mean.h:15-21
 * (kernels sec_mean.cu:12-79).  sec_mean keeps the reference's sequential divide-then-add order
 * per (proposal, channel), so results are bit-identical. */
int ms3d_sec_mean(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);
int ms3d_sec_min(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);
int ms3d_sec_max(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);

/* ---- proposal voxelisation: the arithmetic of clusters_voxelization (minsu3d/model/general_model.py:152-193) between
 * the proposal lists and sparse_quantize -- gather the member coordinates, centre them on the proposal mean
 * (sec_mean's serial order), per-proposal scale = clamp(1 / max_c((hi - lo) / spatial_shape) - 0.01, max = scale),
 * random placement inside the cube with the two U(0,1)^3 draws rand6 = (u1, u2) (device memory), truncate.
 * clusters_idx [S,2] int64 (proposal, point) grouped by proposal; offsets [P+1]; out [S,4] int32 (proposal, x, y, z).
 * Every float operation is the correctly rounded f32 operation of the reference's torch expression, in its order.
 * workspaces: xyz_ws 3*S, mean_ws 3*P, param_ws 4*P floats. */
int ms3d_proposal_voxel_coords(const long long *clusters_idx, int S, const int *offsets, int P, const float *coords,
                               float scale, int spatial_shape, const float *rand6, float *xyz_ws, float *mean_ws,
                               float *param_ws, int *out, ms3d_stream_t stream);

/* ---- pools: replace roipool_fp_cuda / roipool_bp_cuda / global_avg_pool_fp_cuda / _bp_cuda,
 * roipool/roipool.h:17-37 (kernels roipool.cu:12-108).  argmax = first maximum (strict >). */
int ms3d_roipool_fp(int nProposal, int C, const float *feats, const int *proposals_offset, float *output_feats,
                    int *output_maxidx, ms3d_stream_t stream);
int ms3d_roipool_bp(int nProposal, int C, float *d_feats, const int *proposals_offset, const int *output_maxidx,
                    const float *d_output_feats, ms3d_stream_t stream);
int ms3d_global_avg_pool_fp(int nProposal, int C, const float *feats, const int *proposals_offset,
                            float *output_feats, ms3d_stream_t stream);
int ms3d_global_avg_pool_bp(int nProposal, int C, float *d_feats, const int *proposals_offset,
                            const float *d_output_feats, ms3d_stream_t stream);
/* the same with the number of rows the proposals cover (offsets[nProposal] - offsets[0], which the Python wrapper
 * knows as sum_npoint, common_ops.py:160): element-parallel launch instead of one wave per proposal */
int ms3d_global_avg_pool_bp_rows(int nProposal, int C, long n_rows, float *d_feats, const int *proposals_offset,
                                 const float *d_output_feats, ms3d_stream_t stream);

/* out[i, :] = x[idx[i], :] (f32 rows, int64 index): the forward of the row gathers whose backward is
 * ms3d_scatter_add_rows (`features[v2p_map]`, backbone.py:40; `feats[p2v]`, pointgroup.py:89) */
int ms3d_gather_rows(const float *x, const long long *idx /* int64 */, long n, int C, float *out, ms3d_stream_t stream);

/* dst[idx[i], :] += src[i, :] (dst pre-zeroed by the caller): backward of the row gathers features[v2p_map],
 * feats[c_idxs], features[p2v_map] (reference backbone.py:40, general_model.py:156, pointgroup.py:88) */
int ms3d_scatter_add_rows(const float *src, const long long *idx /* int64 */, long n, int C, float *dst,
                          ms3d_stream_t stream);
/* The same sum in a FIXED order (bit-reproducible; the float atomics of ms3d_scatter_add_rows add in arrival order, which
 * is only harmless while no destination row has more than two sources): keys_sorted = idx sorted ascending by a STABLE
 * sort, order[p] = the source row at sorted position p.  dst pre-zeroed by the caller; one writer per destination row. */
int ms3d_scatter_add_rows_sorted(const float *src, const long long *keys_sorted, const long long *order, long n, int C,
                                 float *dst, ms3d_stream_t stream);

/* ---- IoU family: replace get_iou_cuda (get_iou/get_iou.h:16, get_iou.cu:12-38) and
 * get_mask_iou_on_cluster_cuda / get_mask_iou_on_pred_cuda / get_mask_label_cuda
 * (cal_iou_and_masklabel/cal_iou_and_masklabel.h:28-47, .cu:14-140).  One LDS histogram per
 * proposal instead of the reference's O(P*I*np) rescans; same integer counts, same
 * double-precision quotient rounded to f32. */
int ms3d_get_iou(int nInstance, int nProposal, const int *proposals_idx, const int *proposals_offset,
                 const int16_t *instance_labels, const int *instance_pointnum, float *proposals_iou,
                 ms3d_stream_t stream);
int ms3d_get_mask_iou_on_cluster(int nInstance, int nProposal, const int *proposals_idx,
                                 const int *proposals_offset, const int16_t *instance_labels,
                                 const int *instance_pointnum, float *proposals_iou, ms3d_stream_t stream);
int ms3d_get_mask_iou_on_pred(int nInstance, int nProposal, const int *proposals_idx, const int *proposals_offset,
                              const int16_t *instance_labels, const int *instance_pointnum, float *proposals_iou,
                              const float *mask_scores_sigmoid, ms3d_stream_t stream);
int ms3d_get_mask_label(int nInstance, int nProposal, int ignored_label, float iou_thr, const int *proposals_idx,
                        const int *proposals_offset, const int16_t *instance_labels, const int16_t *instance_cls,
                        const float *proposals_iou, uint8_t *mask_label /*bool*/, uint8_t *mask_label_mask /*bool*/,
                        ms3d_stream_t stream);

/* ======================================================================================
 * Sparse-voxel engine: the MinkowskiEngine subset the reference backbone calls.  MinkowskiEngine is a
 * third-party dependency of the reference (un-pinned, README.md:45,73) and is NOT under /root/reference;
 * the entry points below are what a binding for the reference's call sites would need:
 *   ME.utils.sparse_quantize      data/dataset/general_dataset.py:159-163, model/general_model.py:187-189
 *   ME.SparseTensor / coordinate manager + kernel maps   model/module/backbone.py:38, common.py:69,77
 *   ME.MinkowskiConvolution / ConvolutionTranspose fwd+bwd  model/module/common.py:31,37,40,69,77
 *   ME.MinkowskiBatchNorm / MinkowskiReLU (fused into the conv gather)  common.py:35-39,67-68,75-76
 * Kernel maps are output-stationary neighbour tables stored offset-major: nbr[k * V_out + i] = input row
 * feeding output row i through kernel offset k, or -1.
 * ====================================================================================== */
size_t ms3d_coord_workspace_bytes(int n);
/* first-occurrence unique of int32 rows [n,4] (b,x,y,z): unique_idx[u] ascending, inverse[i] = u;
 * *n_unique -> [host].  unique_idx may be NULL. */
int ms3d_sparse_quantize(const int *coords, int n, int *unique_idx, int *inverse, int *n_unique /*[host]*/,
                         void *workspace, size_t workspace_bytes, ms3d_stream_t stream);
/* submanifold 3x3x3 table nbr[27][V]; offset k = ix + 3*iy + 9*iz <-> (ix-1, iy-1, iz-1)*tensor_stride */
int ms3d_kmap_k3(const int *coords, int V, int tensor_stride, int *nbr, void *workspace, size_t workspace_bytes,
                 ms3d_stream_t stream);
/* stride-2 coarse coordinate set (first-occurrence order), parent row and in-cell offset of every fine row */
int ms3d_downsample(const int *coords, int V, int tensor_stride, int *out_coords, int *parent, int *koff,
                    int *n_coarse /*[host]*/, void *workspace, size_t workspace_bytes, ms3d_stream_t stream);
/* k2 s2 tables: nbr_down[8][Vc] (conv) and nbr_up[8][Vf] (transposed conv) */
int ms3d_kmap_k2(const int *parent, const int *koff, int Vf, int Vc, int *nbr_down, int *nbr_up,
                 ms3d_stream_t stream);

/* Pair list = tile-compacted form of an offset-major table, built once per table and shared by every convolution of
 * the level (forward, backward-data, backward-weight).  Output rows are cut into tiles of 64; per tile and offset the
 * valid (input row, output row) pairs are stored contiguously, padded to a multiple of 16 ("batch" = one MFMA group).
 *   tile_start[header_ints] tiles + 1 batch offsets (exclusive scan; last = number of batches), then the schedule of
 *                           the kernels that walk the list: part_start[257] cuts the tiles into 256 parts of near-equal
 *                           batch count (tile t is in part floor(256 * tile_start[t] / batches)); then, 16-byte aligned,
 *                           the pick list int4[tiles] = (tile, first batch, end batch, 0): the tiles of each part by
 *                           descending batch count (stable, per run of 64 tiles)
 *   entries[2 * 16 * batches]  int2 per pair: (input row, (k << 8) | output row inside the tile); pad = (0, k<<8 | 64)
 * capacity() is the worst case in entries (allocate 8 bytes each); only the used prefix is ever touched. */
int ms3d_kmap_pairlist_tiles(int Vout);
int ms3d_kmap_pairlist_header_ints(int Vout);
size_t ms3d_kmap_pairlist_capacity(int K, int Vout);
int ms3d_kmap_pairlist_build(const int *nbr, int K, int Vout, int *tile_start, int *entries, void *workspace,
                             size_t workspace_bytes /* >= ms3d_coord_workspace_bytes(1) */, ms3d_stream_t stream);
/* The same list with tiles of rows_per_tile = 64 (above) or 128 output rows: the convolution kernel for layers with more
 * than 32 channels on a side reads its weights from L2 once per run of batches with the same offset, and a 128-row tile
 * has ~3 batches per offset where a 64-row tile has ~1.5 (and pads 2 % of its slots instead of 25 %). */
int ms3d_kmap_pairlist_header_ints_rows(int Vout, int rows_per_tile);
size_t ms3d_kmap_pairlist_capacity_rows(int K, int Vout, int rows_per_tile);
int ms3d_kmap_pairlist_build_rows(const int *nbr, int K, int Vout, int rows_per_tile, int *tile_start, int *entries,
                                  void *workspace, size_t workspace_bytes, ms3d_stream_t stream);
/* rows_per_tile may also be 32 (round 6): the list a 32 -> 32 layer walks with BOTH column blocks in one wave when the table
 * is dense enough (ms3d_spconv_pairlist_rows_dense).  The library remembers on the host which tile size every list it built
 * has (keyed by the tile_start address; the newest build at an address wins), so that the convolution entry points -- whose
 * signatures carry the two list pointers only -- launch the kernel variant the list was built for: */
int ms3d_kmap_pairlist_rows_of(const int *tile_start);   /* 32 / 64 / 128; 64 for a list this library did not build */

/* Offset-major pair list of a table (the classic per-offset in/out index pairs) for the backward-weight kernel:
 *   kt_start[header_ints]    K * tiles + 1 pair offsets: first pair of (offset k, 64-row tile t) at [k * tiles + t],
 *                            last = number of pairs; then part_start[257] (tile ranges of near-equal pair count, the
 *                            workgroups of the backward-weight kernel) and the per-tile pair prefix [tiles + 1] it is cut from
 *   entries[2 * pairs]       int2 per pair: (input row, output row), ascending output row inside an offset */
size_t ms3d_kmap_offsetlist_header_ints(int K, int Vout);
size_t ms3d_kmap_offsetlist_capacity(int K, int Vout);
int ms3d_kmap_offsetlist_build(const int *nbr, int K, int Vout, int *kt_start, int *entries, void *workspace,
                               size_t workspace_bytes, ms3d_stream_t stream);

/* spatial sort keys (batch | 45-bit Morton code): rows sorted by this key keep a voxel's 26 neighbours close in
 * memory, so the conv gathers of one XCD stay inside its own L2 slice */
int ms3d_morton_keys(const int *coords, int V, long long *keys, ms3d_stream_t stream);

/* weights W[K][Cin][Cout] -> MFMA-fragment order.  transpose=1 (+mirror=1 for k3) gives the backward-data
 * operator: Weff[k] = W[mirror ? K-1-k : k]^T with Cin_eff = Cout, Cout_eff = Cin. */
size_t ms3d_spconv_wf_floats(int K, int Cin_eff, int Cout_eff);
/* wf_stream (same size as wf, or NULL): the same weights in "streamed" order, read 16 bytes per lane straight from L2 by
 * the kernels for layers whose weights do not fit LDS (more than 32 channels on a side, pair-listed tables) */
int ms3d_spconv_prep_weights(const float *W, int K, int Cin_eff, int Cout_eff, int transpose, int mirror, float *wf,
                             float *wf_stream, ms3d_stream_t stream);
/* out[i,:] = sum_k act(in[nbr[k][i],:]) @ Weff[k] (+ residual); act = optional x*pre_scale+pre_shift (+ReLU).
 * With bn_x != NULL the epilogue is the backward of a fused BN+ReLU: out = dz = acc * [bn_x*bn_scale+bn_shift > 0]
 * and bn_partial [ms3d_spconv_partial_blocks()][2][Cout] receives per-block sums of dz and dz*xhat. */
int ms3d_spconv_partial_blocks(int Vout, int K, int Cin, int Cout, int with_pairlist /* 0 = no list; 1 = a 64-row list;
                               otherwise the list's rows per tile (ms3d_kmap_pairlist_rows_of) */);
/* rows per tile of the pair list a forward / backward-data convolution of this shape wants in pl_tile_start / pl_entries:
 * 0 = none, 64 = ms3d_kmap_pairlist_build, 128 = ms3d_kmap_pairlist_build_rows(.., 128, ..) */
int ms3d_spconv_pairlist_rows(int Vout, int K, int Cin, int Cout);
/* the same for a DENSE table (about 8+ of 27 neighbours per row: every level but the full-resolution one): 32 for the
 * 32 -> 32 layers (both column blocks per wave on 32-row tiles: every row gathered once instead of once per 16-column
 * slice; at 5.5 neighbours per row a 32-row tile pads 6.5 pairs per offset to 16 and loses), otherwise
 * ms3d_spconv_pairlist_rows.  The caller decides from the table's pair count which of the two lists to build. */
int ms3d_spconv_pairlist_rows_dense(int Vout, int K, int Cin, int Cout);
/* 1 when the convolution kernels have a pair-list variant worth building the list for (full-resolution levels) */
int ms3d_kmap_pairlist_wanted(int K, int Vout);
int ms3d_spconv_forward(const float *in, const float *wf, const int *nbr, int Vout, int K, int Cin, int Cout,
                        float *out, const float *pre_scale, const float *pre_shift, int pre_relu,
                        const float *residual, const float *bn_x, const float *bn_scale, const float *bn_shift,
                        const float *bn_mean, const float *bn_invstd, float *bn_partial, int out_stats,
                        const float *bias /* [Cout] or NULL */,
                        const int *pl_tile_start /* pair list of `nbr` (ms3d_kmap_pairlist_build) or NULL */,
                        const int *pl_entries,
                        const float *wf_stream /* streamed image of the same weights; required when a pair list is given
                                                  and a side has more than 32 channels, NULL otherwise */,
                        ms3d_stream_t stream);
/* out_stats != 0 (forward only): bn_partial [ms3d_spconv_partial_blocks()][2][Cout] receives per-block
 * (sum, sum of squares) of the OUTPUT rows (after the residual add) -> feed ms3d_bn_finalize, no extra pass. */
int ms3d_spconv_prep_weights_pair(const float *W, int K, int Cin, int Cout, int mirror_bwd, float *wf, float *wft,
                                  float *wf_stream /* or NULL */, float *wft_stream /* or NULL */, ms3d_stream_t stream);
/* 1 if a layer of this shape can be served by the weight-streaming kernel (then its descriptor in
 * ms3d_spconv_prep_weights_multi must ask for the streamed images: field `stream`) */
int ms3d_spconv_wants_stream_image(int K, int Cin, int Cout);
/* what the aux slot (2n floats) behind each weight image of a layer buffer holds: 0 nothing, 1 the streamed f32 image,
 * 2 the three-piece bf16 image (wide square layers).  Layer buffer = [image n | aux 2n | transposed image n | aux 2n],
 * n = ms3d_spconv_wf_floats(K, Cin, Cout). */
int ms3d_spconv_aux_kind(int K, int Cin, int Cout);
/* Both images of n layers in ONE launch (a U-Net re-lays ~90 weight tensors per step, ~5 us of dispatch each).
 * descs: device array of n 48-byte records {const float *W; float *wf; float *wft; int K, Cin, Cout, mirror_bwd,
 * block_begin, stream}, stream = ms3d_spconv_aux_kind(K, Cin, Cout), block_begin = running sum of ms3d_spconv_prep_blocks(K, Cin, Cout); total_blocks = the full sum.
 * wf and wft each have room for 3 * ms3d_spconv_wf_floats() floats: the image, then its aux image (slot of 2n). */
int ms3d_spconv_prep_blocks(int K, int Cin, int Cout);
int ms3d_spconv_prep_weights_multi(const void *descs, int n, int total_blocks, ms3d_stream_t stream);
int ms3d_bn_finalize(const float *partial, int nparts, long V, int C, float eps, float momentum, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, float *mean, float *invstd,
                     float *scale, float *shift, ms3d_stream_t stream);
/* dW[k] = sum_i act(in[nbr[k][i],:])^T dout[i,:].  Deterministic: per-row-chunk partial slabs reduced in a fixed
 * order.  partial_ws MUST hold ms3d_spconv_wgrad_ws_floats(Vout, K, Cin, Cout) floats: the slab count depends on the
 * kernel that serves the shape (K = 1 heads: up to 1024 slabs; bf16x3 layers: the operand images behind the slabs).
 * ms3d_spconv_wgrad_row_chunks(Vout) * K*Cin*Cout is only a LOWER bound (the f32 table walk's slab count). */
int ms3d_spconv_wgrad_row_chunks(int Vout);
/* floats of partial_ws a backward-weight call may use (slabs; wide K = 27 layers add the three-piece bf16 images of both
 * operands).  K = 27 is the SUBMANIFOLD case: `in` and `dout` have the same Vout rows -- a 27-offset table whose input row
 * set differs from its output row set is outside this entry point's contract (the bf16x3 path splits Vout rows of `in`). */
size_t ms3d_spconv_wgrad_ws_floats(int Vout, int K, int Cin, int Cout);
/* 1 when a backward-weight call of this shape runs on three-piece bf16 operands (offset_list: an offset list is passed) */
int ms3d_spconv_wgrad_is_bf16x3(int Vout, int K, int Cin, int Cout, int offset_list);
int ms3d_spconv_backward_weight(const float *in, const float *dout, const int *nbr, int Vout, int K, int Cin,
                                int Cout, float *dW, const float *pre_scale, const float *pre_shift, int pre_relu,
                                float *partial_ws,
                                const int *ol_kt_start /* offset list of `nbr` (ms3d_kmap_offsetlist_build) or NULL */,
                                const int *ol_entries, ms3d_stream_t stream);
/* One-call layer entry points (forward / backward of a fused [BN -> ReLU ->] conv): same kernels as above, enqueued
 * from native code.  wf_buf holds both weight images, each followed by its streamed form
 * (4 * ms3d_spconv_wf_floats(K,Cin,Cout) floats: [wf | wf streamed | wft | wft streamed]) and is
 * kept by the caller between forward and backward; ws: ms3d_spconv_layer_ws_floats() floats of scratch.
 * layer_forward with W == NULL skips the re-lay: wf_buf already holds the current images (prep_weights_multi). */
size_t ms3d_spconv_layer_ws_floats(int Vin, int Vout, int K, int Cin, int Cout);
int ms3d_spconv_layer_forward(const float *x, const float *W, const int *nbr_fwd, int Vout, int K, int Cin, int Cout,
                              int mirror_bwd, const float *pre_scale, const float *pre_shift, int pre_relu,
                              const float *residual, const float *bias, float *wf_buf, float *y, float *stat_partial,
                              const int *pl_tile_start /* pair list of nbr_fwd or NULL */, const int *pl_entries,
                              void *ev_start /* hipEvent_t or NULL */, void *ev_stop, ms3d_stream_t stream);
/* HIP events for timing a launch on the stream it is issued on (recorded inside ms3d_spconv_layer_forward around
 * the convolution kernel only) */
void *ms3d_event_create(void);
void ms3d_event_destroy(void *event);
int ms3d_event_record(void *event, ms3d_stream_t stream);
float ms3d_event_elapsed_ms(void *start, void *stop);
int ms3d_spconv_layer_backward(const float *x, const float *dy, const float *wf_buf, const int *nbr_fwd,
                               const int *nbr_bwd, int Vin, int Vout, int K, int Cin, int Cout, const float *scale,
                               const float *shift, const float *mean, const float *invstd, int pre_relu, int training,
                               int need_dx, float *dx,
                               const float *dx_add /* or NULL: [Vin, Cin] added to dx -- the gradient that reaches x over a skip
                                                      connection (fused into the BatchNorm-backward pass / the residual epilogue) */,
                               float *dgb, float *dW, float *ws,
                               const int *ol_fwd_kt_start /* offset list of nbr_fwd or NULL */, const int *ol_fwd_entries,
                               const int *pl_bwd_tile_start /* pair list of nbr_bwd or NULL */, const int *pl_bwd_entries,
                               void *ev_start /* hipEvent_t or NULL: around the backward-data kernel */, void *ev_stop,
                               void *ev_wg_start /* hipEvent_t or NULL: around the backward-weight kernels */, void *ev_wg_stop,
                               float *ws_wgrad /* slab workspace of the second stream (ms3d_spconv_layer_ws_floats) or NULL */,
                               ms3d_stream_t wgrad_stream /* NULL: backward-weight on `stream`; else it runs on this
                                                             stream beside the backward-data chain */,
                               int join /* 1: `stream` waits for the backward-weight before the call returns control
                                           of it; 0: the caller joins the streams before dW is read */,
                               float *wgrad_slabs /* NULL, or ms3d_spconv_wgrad_ws_floats() floats of the caller's that outlive
                                                     the call: the backward-weight slabs go there instead of into ws */,
                               int *wgrad_deferred_nblk /* [host] NULL, or (with wgrad_slabs): the slab reduction is NOT
                                                           launched; receives the number of slabs to reduce later with
                                                           ms3d_wgrad_reduce_multi (0: dW is final) */,
                               void *wgrad_deferred_launch /* [host] NULL, or (with the two above, no timing events) a buffer of
                                                              ms3d_spconv_wgrad_launch_bytes(): when the layer's backward-weight
                                                              takes the f32 table walk (the small levels) the KERNEL is not
                                                              launched either but described here; int[4] of the buffer =
                                                              variant (0: it was launched as usual), int[1..3] = its grid */,
                               ms3d_stream_t stream);
/* Deferred backward-weight launches of MANY layers of one variant as one launch: descs = DEVICE array of the 128-byte
 * descriptions, each with its first int set to the layer's first block (blocks are numbered layer after layer, a layer
 * has int[1] * int[2] * int[3] of them), total_blocks = their sum.  x, dy, the tables and the slab areas the descriptions
 * point to must still be alive.  Follow with ms3d_wgrad_reduce_multi over the same layers. */
size_t ms3d_spconv_wgrad_launch_bytes(void);
int ms3d_spconv_wgrad_is_table_walk(int Vout, int K, int Cin, int Cout, int offset_list);
int ms3d_spconv_wgrad_multi(const void *descs, int n_desc, int total_blocks, int variant, ms3d_stream_t stream);
/* The slab reductions dW = sum of slabs of MANY layers in one launch, bit-identical to the per-layer reduction.
 * descs: DEVICE array of n_desc records {const float *slabs; float *dW; int64_t n (floats per slab); int32_t nblk (slabs);
 * int32_t block_begin} (32 bytes each) in ascending block order; a layer takes ms3d_wgrad_reduce_blocks() blocks and sets
 * bit 31 of block_begin when that call reports the 16-byte geometry (*wide = 1); total_blocks = their sum. */
int ms3d_wgrad_reduce_blocks(long n, const float *slabs, const float *dW, int *wide /*[host]*/);
int ms3d_wgrad_reduce_multi(const void *descs, int n_desc, int total_blocks, ms3d_stream_t stream);

/* BatchNorm1d over rows, training mode (biased var for normalisation, unbiased into running_var) */
int ms3d_bn_stats(const float *x, long V, int C, float eps, float momentum, const float *gamma, const float *beta,
                  float *running_mean, float *running_var, float *mean, float *invstd, float *scale, float *shift,
                  float *partial_ws, int partial_rows, ms3d_stream_t stream);
int ms3d_bn_apply(const float *x, long V, int C, const float *scale, const float *shift, int relu, float *y,
                  ms3d_stream_t stream);
int ms3d_reduce_partials(const float *partial, int nparts, int n, float *out, ms3d_stream_t stream);
/* column sums of x [V, C] -> out2c[0..C) (out2c[C..2C) = column sums of squares); partial_ws: partial_rows*2*C floats */
int ms3d_column_sum(const float *x, long V, int C, float *partial_ws, int partial_rows, float *out2c, ms3d_stream_t stream);
int ms3d_bn_bwd_apply(const float *dz, const float *x, long V, int C, const float *scale, const float *mean,
                      const float *invstd, const float *s1s2, float *dx, ms3d_stream_t stream);
/* ms3d_reduce_partials + ms3d_bn_bwd_apply_add in ONE launch, bit-identical: s1s2 [2][C] = column sums of partial
 * [nparts][2][C]; dx (or NULL: sums only) = scale * (dz - s1/V - xhat * s2/V) [+ add]; dz == dx allowed.  2C <= 1024. */
int ms3d_bn_bwd_reduce_apply(const float *partial, int nparts, const float *dz, const float *x, long V, int C,
                             const float *scale, const float *mean, const float *invstd, const float *add, float *dx,
                             float *s1s2, ms3d_stream_t stream);
/* the same with `add` [V, C] (or NULL) added to the result */
int ms3d_bn_bwd_apply_add(const float *dz, const float *x, long V, int C, const float *scale, const float *mean,
                          const float *invstd, const float *s1s2, const float *add, float *dx, ms3d_stream_t stream);
int ms3d_bn_bwd_partial(const float *dy, const float *x, long V, int C, const float *scale, const float *shift,
                        const float *mean, const float *invstd, int relu, float *dz, float *partial_ws,
                        int partial_rows, int *nparts_out /*[host]*/, ms3d_stream_t stream);

/* ======================================================================================
 * Instance post-processing (validation / test time): replaces the dense [P, N] mask algebra of
 * model/pointgroup.py:197-265 (cross IoU by mask matrix product on the host + numpy greedy NMS).
 * ====================================================================================== */
/* inter[a][b] = number of points shared by proposals a and b (diagonal = proposal sizes).  The (cluster, point) pairs
 * must be unique and sorted by point: pair_point[S] ascending, pair_cluster[S] the proposal of each pair. */
int ms3d_proposal_cross_intersection(const int *pair_point, const int *pair_cluster, int S, int P, int *inter /*[P,P]*/,
                                     ms3d_stream_t stream);
/* greedy non-maximum suppression in the given order (descending score): a proposal is picked unless an earlier pick
 * has IoU = inter/(n_a+n_b-inter) > threshold with it (float32, as the reference).  pick[P], *n_pick on the device;
 * suppressed_ws: P bytes of scratch. */
int ms3d_nms_greedy(const int *inter, const int *order, int P, float threshold, unsigned char *suppressed_ws, int *pick,
                    int *n_pick, ms3d_stream_t stream);

/* ======================================================================================
 * Augmentation: elastic distortion (util/transform.py:65-84, called twice per training scene from
 * data/dataset/general_dataset.py:118-120).  noise: three float32 grids [3][bx][by][bz] drawn by the caller
 * (host RNG, same shapes/order as the reference), blurred in place (noise_tmp: same size scratch);
 * out = xyz + mag * trilinear(noise)(xyz), float64, grid nodes at linspace(-(b-1)*gran, (b-1)*gran, b).
 * ====================================================================================== */
int ms3d_elastic_distort(const double *xyz /*[N,3]*/, int N, float *noise, float *noise_tmp, int bx, int by, int bz,
                         double gran, double mag, double *out /*[N,3]*/, ms3d_stream_t stream);

/* ---- optimizer: one Adam step over all parameter tensors of a model in one launch (the reference's optimizer is
 * torch.optim.Adam through Hydra, config/model/base.yaml:23-28; arithmetic as torch's fused Adam, f32).
 * chunks: int2 (tensor, chunk index) per workgroup, ms3d_adam_chunk_elems() elements per chunk; p / g / m / v: device
 * arrays of device pointers (parameter, gradient, exp_avg, exp_avg_sq), sizes: elements per tensor (device, int64).
 * bias_correction_i = 1 - beta_i^step. */
int ms3d_adam_chunk_elems(void);
int ms3d_adam_step(const int *chunks, int n_chunks, void *const *p_ptrs, const void *const *g_ptrs, void *const *m_ptrs,
                   void *const *v_ptrs, const long *sizes, float lr, float beta1, float beta2, float eps,
                   float weight_decay, double bias_correction1, double bias_correction2, ms3d_stream_t stream);
/* the same with a step counter PER TENSOR (torch.optim.Adam starts a parameter's counter with its first gradient):
 * coef = device float2 per tensor (lr / bias_correction1, sqrt(bias_correction2)) of that tensor's own step */
int ms3d_adam_step_multi(const int *chunks, int n_chunks, void *const *p_ptrs, const void *const *g_ptrs,
                         void *const *m_ptrs, void *const *v_ptrs, const long *sizes, const float *coef, float beta1,
                         float beta2, float eps, float weight_decay, ms3d_stream_t stream);

/* ---- device-wide exclusive prefix sum of int32 (the utility behind the ball query's cell / list starts, the clustering
 * output assembly and the coordinate engine): out[i] = in[0] + .. + in[i-1], in == out allowed, *total_out_dev (device,
 * optional) = the sum of all.  One launch (decoupled look-back between workgroups).  workspace: ms3d_scan_i32_workspace_bytes(). */
size_t ms3d_scan_i32_workspace_bytes(void);
int ms3d_scan_i32(const int *in, int *out, int n, int *total_out_dev, void *workspace, ms3d_stream_t stream);

/* ---- per-point losses of the backbone heads, forward and gradients (the reference: GeneralModel._loss,
 * model/general_model.py:36-50 -- cross_entropy(ignore_index=-1) -- and PTOffsetLoss, loss/pt_offset_loss.py:11-38 -- mean
 * L1 norm of the offset error and mean negative cosine over the points with instance_ids != -1; with no valid point a
 * loss is 0).  forward: out5 = (semantic loss, offset norm loss, offset direction loss, 1 / #labelled, 1 / #instance
 * points); d_scores [N, C], d_norm / d_dir [N, 3] receive the UNNORMALISED gradients; partial_ws: 5 doubles per block of
 * ms3d_point_losses_blocks(N).  scale_grads (backward): d_scores *= *g_sem * out5[3] in place and
 * d_norm = (*g_norm * d_norm + *g_dir * d_dir) * out5[4]; g_* are DEVICE scalars (the upstream gradients) or NULL (= 0). */
int ms3d_point_losses_blocks(long N);
int ms3d_point_losses_forward(const float *scores /*[N,C]*/, const short *labels /*[N]*/, const float *pred_offsets /*[N,3]*/,
                              const float *centre /*[N,3]*/, const float *xyz /*[N,3]*/, const short *instance_ids /*[N]*/,
                              long N, int C, float *d_scores, float *d_norm, float *d_dir, double *partial_ws, float *out5,
                              ms3d_stream_t stream);
int ms3d_point_losses_scale_grads(float *d_scores, long n_scores, float *d_norm, const float *d_dir, long n_off,
                                  const float *out5, const float *g_sem, const float *g_norm, const float *g_dir,
                                  ms3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MINSU3D_HIP_H */
