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
                           int *start_len, int *n_active /*[host]*/, void *workspace, size_t workspace_bytes,
                           ms3d_stream_t stream);

/* ---- BFS clustering: replaces pg_bfs_cluster / sg_bfs_cluster, bfs_cluster/bfs_cluster.h:18-19
 * (host C++ bfs_cluster.cpp:28-187).  Runs on the DEVICE (the reference copies the ball-query
 * result to the host and runs a serial FIFO BFS); output order is identical to the serial BFS:
 * clusters by ascending seed, members in FIFO visit order, including the directed case when
 * the 1000-neighbour cap bites.  cluster_idxs has capacity [N,2], cluster_offsets [N+1];
 * counts[0] = nCluster, counts[1] = sumNPoint are returned to the [host]. */
size_t ms3d_bfs_workspace_bytes(int N);
int ms3d_pg_bfs_cluster(const int16_t *semantic_label, const int *ball_query_idxs, const int *start_len, int N,
                        int threshold, int *cluster_idxs, int *cluster_offsets, int *counts /*[host,2]*/,
                        void *workspace, size_t workspace_bytes, ms3d_stream_t stream);
int ms3d_sg_bfs_cluster(const float *class_numpoint_mean /*[host]*/, const int *ball_query_idxs,
                        const int *start_len, int N, float threshold, int class_id, int *cluster_idxs,
                        int *cluster_offsets, int *counts /*[host,2]*/, void *workspace, size_t workspace_bytes,
                        ms3d_stream_t stream);

/* ---- segment ops: replace sec_mean_cuda / sec_min_cuda / sec_max_cuda, sec_mean/sec_mean.h:15-21
 * (kernels sec_mean.cu:12-79).  sec_mean keeps the reference's sequential divide-then-add order
 * per (proposal, channel), so results are bit-identical. */
int ms3d_sec_mean(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);
int ms3d_sec_min(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);
int ms3d_sec_max(int nProposal, int C, const float *inp, const int *offsets, float *out, ms3d_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* MINSU3D_HIP_H */
