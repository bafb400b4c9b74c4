// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin extern "C" doorway (our own code) onto the REFERENCE's own common_ops functions,
// compiled by oracle/build_ref.py straight from /root/reference/minsu3d/common_ops/src
// into oracle/_ref/libminsu3d_ref.so.  No reference source lives in this repo: the
// prototypes below only name the reference entry points
//   bfs_cluster/bfs_cluster.h:15-19, sec_mean/sec_mean.h:14-21, roipool/roipool.h:15-37,
//   get_iou/get_iou.h:15-16, cal_iou_and_masklabel/cal_iou_and_masklabel.h:12-47,
//   hierarchical_aggregation/hierarchical_aggregation.h:14-28.
// CPU entry points (BFS, HAIS split) run anywhere; GPU entry points launch the reference's
// own kernels and therefore need a device (the GPU box).
#include <ATen/ATen.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

// ---- reference symbols (defined in the reference translation units) ----
void pg_bfs_cluster(at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, const int N, int threshold);
void sg_bfs_cluster(at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, const int N, float threshold,
                    const int class_id);
void hierarchical_aggregation(at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor,
                              at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor,
                              at::Tensor, at::Tensor, at::Tensor, at::Tensor, const int N, const int using_set_aggr_,
                              const int ignored_label);
int ballquery_batch_p_cuda(int n, int meanActive, float radius, const float *xyz, const uint8_t *batch_idxs,
                           const int *batch_offsets, int *idx, int *start_len, hipStream_t stream);
void sec_mean_cuda(int nProposal, int C, float *inp, int *offsets, float *out);
void sec_min_cuda(int nProposal, int C, float *inp, int *offsets, float *out);
void sec_max_cuda(int nProposal, int C, float *inp, int *offsets, float *out);
void roipool_fp_cuda(int nProposal, int C, float *feats, int *proposals_offset, float *output_feats,
                     int *output_maxidx);
void roipool_bp_cuda(int nProposal, int C, float *d_feats, int *proposals_offset, int *output_maxidx,
                     float *d_output_feats);
void global_avg_pool_fp_cuda(int nProposal, int C, float *feats, int *proposals_offset, float *output_feats);
void global_avg_pool_bp_cuda(int nProposal, int C, float *d_feats, int *proposals_offset, float *d_output_feats);
void get_iou_cuda(int nInstance, int nProposal, int *proposals_idx, int *proposals_offset,
                  int16_t *instance_labels, int *instance_pointnum, float *proposals_iou);
void get_mask_iou_on_cluster_cuda(int nInstance, int nProposal, int *proposals_idx, int *proposals_offset,
                                  int16_t *instance_labels, int *instance_pointnum, float *proposals_iou);
void get_mask_iou_on_pred_cuda(int nInstance, int nProposal, int *proposals_idx, int *proposals_offset,
                               int16_t *instance_labels, int *instance_pointnum, float *proposals_iou,
                               float *mask_scores_sigmoid);
void get_mask_label_cuda(int nInstance, int nProposal, int ignored_label, float iou_thr, int *proposals_idx,
                         int *proposals_offset, int16_t *instance_labels, int16_t *instance_cls,
                         float *proposals_iou, bool *mask_label, bool *mask_label_mask);

namespace {
template <typename T>
at::Tensor wrap(const T *p, std::initializer_list<int64_t> shape, at::ScalarType st)
{
    return at::from_blob(const_cast<T *>(p), shape, at::TensorOptions().dtype(st).device(at::kCPU));
}
int copy_out(const at::Tensor &idxs, const at::Tensor &offs, int *out_idxs, int *out_offsets, int *sum_out)
{
    const int64_t rows = idxs.numel() / 2, no = offs.numel();
    if (rows) std::memcpy(out_idxs, idxs.contiguous().data_ptr<int>(), sizeof(int) * 2 * rows);
    if (no) std::memcpy(out_offsets, offs.contiguous().data_ptr<int>(), sizeof(int) * no);
    *sum_out = (int)rows;
    return (int)no - 1;
}
}  // namespace

extern "C" {

int ref_pg_bfs_cluster(const int16_t *sem, const int *ball_idx, long nActive, const int *start_len, int N,
                       int threshold, int *out_idxs, int *out_offsets, int *sum_out)
{
    auto t_sem = wrap(sem, {N}, at::kShort);
    auto t_idx = wrap(ball_idx, {nActive}, at::kInt);
    auto t_sl = wrap(start_len, {N, 2}, at::kInt);
    auto o_idx = at::empty({0}, at::kInt), o_off = at::empty({0}, at::kInt);
    pg_bfs_cluster(t_sem, t_idx, t_sl, o_idx, o_off, N, threshold);
    return copy_out(o_idx, o_off, out_idxs, out_offsets, sum_out);
}

int ref_sg_bfs_cluster(const float *class_numpoint_mean, int nclass, const int *ball_idx, long nActive,
                       const int *start_len, int N, float threshold, int class_id, int *out_idxs,
                       int *out_offsets, int *sum_out)
{
    auto t_mean = wrap(class_numpoint_mean, {nclass}, at::kFloat);
    auto t_idx = wrap(ball_idx, {nActive}, at::kInt);
    auto t_sl = wrap(start_len, {N, 2}, at::kInt);
    auto o_idx = at::empty({0}, at::kInt), o_off = at::empty({0}, at::kInt);
    sg_bfs_cluster(t_mean, t_idx, t_sl, o_idx, o_off, N, threshold, class_id);
    return copy_out(o_idx, o_off, out_idxs, out_offsets, sum_out);
}

// Returns (kept, primary[/post]) exactly as the reference leaves them; the Python-side
// merge of functions/hais_ops.py:55-73 is restated by the caller (oracle/oracle.py).
// out_* buffers: kept_idxs [N,2], kept_offs [N+1], prim_idxs [2N,2], prim_offs [N+1].
int ref_hierarchical_aggregation(const int16_t *sem, const float *coord_shift, const uint8_t *batch_idxs,
                                 const int *ball_idx, long nActive, const int *start_len, int N,
                                 int using_set_aggr, const float *point_num_avg, const float *radius_avg,
                                 int nclass, int ignored_label, int *kept_idxs, int *kept_offs, int *kept_sum,
                                 int *prim_idxs, int *prim_offs, int *prim_sum, int *n_prim)
{
    auto t_sem = wrap(sem, {N}, at::kShort);
    auto t_cs = wrap(coord_shift, {N, 3}, at::kFloat);
    auto t_b = wrap(batch_idxs, {N}, at::kByte);
    auto t_idx = wrap(ball_idx, {nActive}, at::kInt);
    auto t_sl = wrap(start_len, {N, 2}, at::kInt);
    auto t_pna = wrap(point_num_avg, {nclass}, at::kFloat);
    auto t_ra = wrap(radius_avg, {nclass}, at::kFloat);
    auto E = [] { return at::empty({0}, at::kInt); };
    auto F = [] { return at::empty({0}, at::kFloat); };
    auto f_i = E(), f_o = E(), f_c = F(), k_i = E(), k_o = E(), k_c = F(), p_i = E(), p_o = E(), p_c = F(),
         pp_i = E(), pp_o = E();
    hierarchical_aggregation(t_sem, t_cs, t_b, t_idx, t_sl, f_i, f_o, f_c, k_i, k_o, k_c, p_i, p_o, p_c, pp_i,
                             pp_o, t_pna, t_ra, N, using_set_aggr, ignored_label);
    int nk = copy_out(k_i, k_o, kept_idxs, kept_offs, kept_sum);
    if (using_set_aggr && pp_o.numel() > 0) {
        // hais_ops.py:58-61: cut the tail at primary_offsets_post[-1]
        const int total = pp_o.data_ptr<int>()[pp_o.numel() - 1];
        auto cut = pp_i.narrow(0, 0, total);
        *n_prim = copy_out(cut, pp_o, prim_idxs, prim_offs, prim_sum);
    } else {
        *n_prim = copy_out(p_i, p_o, prim_idxs, prim_offs, prim_sum);
    }
    return nk;
}

// ---- GPU: the reference's own kernels on device pointers (default stream, then sync) ----
int ref_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz, const uint8_t *batch_idxs,
                          const int *batch_offsets, int *idx, int *start_len)
{
    int r = ballquery_batch_p_cuda(n, meanActive, radius, xyz, batch_idxs, batch_offsets, idx, start_len, nullptr);
    (void)hipDeviceSynchronize();
    return r;
}
void ref_sec_mean(int P, int C, float *inp, int *off, float *out) { sec_mean_cuda(P, C, inp, off, out); (void)hipDeviceSynchronize(); }
void ref_sec_min(int P, int C, float *inp, int *off, float *out) { sec_min_cuda(P, C, inp, off, out); (void)hipDeviceSynchronize(); }
void ref_sec_max(int P, int C, float *inp, int *off, float *out) { sec_max_cuda(P, C, inp, off, out); (void)hipDeviceSynchronize(); }
void ref_roipool_fp(int P, int C, float *feats, int *off, float *out, int *maxidx) { roipool_fp_cuda(P, C, feats, off, out, maxidx); (void)hipDeviceSynchronize(); }
void ref_roipool_bp(int P, int C, float *d_feats, int *off, int *maxidx, float *d_out) { roipool_bp_cuda(P, C, d_feats, off, maxidx, d_out); (void)hipDeviceSynchronize(); }
void ref_global_avg_pool_fp(int P, int C, float *feats, int *off, float *out) { global_avg_pool_fp_cuda(P, C, feats, off, out); (void)hipDeviceSynchronize(); }
void ref_global_avg_pool_bp(int P, int C, float *d_feats, int *off, float *d_out) { global_avg_pool_bp_cuda(P, C, d_feats, off, d_out); (void)hipDeviceSynchronize(); }
void ref_get_iou(int I, int P, int *pi, int *po, int16_t *lab, int *pn, float *iou) { get_iou_cuda(I, P, pi, po, lab, pn, iou); (void)hipDeviceSynchronize(); }
void ref_get_mask_iou_on_cluster(int I, int P, int *pi, int *po, int16_t *lab, int *pn, float *iou) { get_mask_iou_on_cluster_cuda(I, P, pi, po, lab, pn, iou); }
void ref_get_mask_iou_on_pred(int I, int P, int *pi, int *po, int16_t *lab, int *pn, float *iou, float *sig) { get_mask_iou_on_pred_cuda(I, P, pi, po, lab, pn, iou, sig); }
void ref_get_mask_label(int I, int P, int ignored, float thr, int *pi, int *po, int16_t *lab, int16_t *cls, float *iou, bool *ml, bool *mlm) { get_mask_label_cuda(I, P, ignored, thr, pi, po, lab, cls, iou, ml, mlm); }

}  // extern "C"
