// `import COMMON_OPS` as a PyTorch-ROCm C++ extension: the 15 functions of the reference's pybind module
// (minsu3d/common_ops/src/common_ops_api.cpp:6-30) with the reference's own signatures (at::Tensor by value + C ints /
// floats; headers bfs_cluster/bfs_cluster.h:15-19, sec_mean/sec_mean.h:14-21, roipool/roipool.h:15-37,
// get_iou/get_iou.h:15-16, cal_iou_and_masklabel/cal_iou_and_masklabel.h:12-47,
// hierarchical_aggregation/hierarchical_aggregation.h:14-28), served by libminsu3d_hip.so through the C ABI of
// include/minsu3d_hip.h on torch's CURRENT HIP stream (c10::hip::getCurrentHIPStream).  Host compiler only: there is no
// device code here, the kernels live in the library.
//
// Ownership rules are the reference's: the caller allocates every output and passes it in; the clustering functions
// resize_() the caller's (possibly empty) outputs to the data-dependent sizes (bfs_cluster.cpp:157-160,
// hierarchical_aggregation.cpp:133-175) and accept HOST tensors, which is what model/pointgroup.py:49-52, hais.py:52-56
// and softgroup.py:60-63 pass: inputs are copied to the GPU, the clustering runs there (the product has no CPU
// implementation), results are copied into the caller's tensors.  ballquery_batch_p returns the total hit count and
// leaves positions >= n*meanActive unwritten, so the wrapper's retry loop (functions/common_ops.py:31-38) behaves as
// with the reference.  Errors of the library raise (the reference prints and exit(-1)s, bfs_cluster.cu:82-86).
//
// minsu3d_amd/dropin/COMMON_OPS.py is the same module over ctypes (the fallback when this extension is not built).
#include <torch/extension.h>
#include <mutex>
#include <c10/hip/HIPStream.h>

#include <vector>

#include "../../include/minsu3d_hip.h"

namespace {

inline ms3d_stream_t cur() { return (ms3d_stream_t)c10::hip::getCurrentHIPStream().stream(); }

inline void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed with code ", rc, " (libminsu3d_hip.so; there is no CPU fallback)");
}

// contiguous device tensor (host tensors are uploaded: the computation always runs in the HIP library)
inline at::Tensor dev(const at::Tensor &t)
{
    return (t.is_cuda() ? t : t.cuda()).contiguous();
}

inline void assign(at::Tensor &dst, const at::Tensor &src)   // the callee-side resize_ + fill (bfs_cluster.cpp:157-160)
{
    dst.resize_(src.sizes());
    dst.copy_(src);
}

inline at::Tensor scratch(size_t bytes, const at::Tensor &like)
{
    return at::empty({(int64_t)bytes + 256}, like.options().dtype(at::kByte));
}

// the last ball query's "no list reached the 1000 cap" flag, keyed by the start_len buffer it wrote: lets the BFS that
// consumes that very graph skip a device->host check (-1 = unknown: decided on the device)
struct LastGraph { const void *start_len = nullptr; int n = 0; int capped = -1; } g_last;
std::mutex g_last_lock;   // ball queries and their consumers may run on several host threads

// ONE use per ball query: the key is an address and a length, and the caching allocator may hand the same
// address to a later start_len of equal length (or the caller may edit the tensor in place); the grouping that follows a
// query consumes the hint, anything later decides on the device
inline int capped_hint(const at::Tensor &start_len_dev)
{
    // consumed by the call that MATCHES it only: an unrelated clustering call in between (another thread's graph, a host
    // graph) takes the validated general route for itself and leaves the hint for the grouping it belongs to
    std::lock_guard<std::mutex> guard(g_last_lock);
    if (g_last.start_len != start_len_dev.data_ptr() || g_last.n != start_len_dev.size(0)) return -1;
    const int hint = g_last.capped;
    g_last = LastGraph();
    return hint;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- common
int ballquery_batch_p(at::Tensor xyz, at::Tensor batch_idxs, at::Tensor batch_offsets, at::Tensor idx, at::Tensor start_len,
                      int n, int meanActive, float radius)
{
    TORCH_CHECK(idx.is_cuda() && start_len.is_cuda() && idx.is_contiguous() && start_len.is_contiguous(),
                "ballquery_batch_p: idx / start_len are written in place and must be contiguous device tensors");
    if (n == 0) return 0;
    at::Tensor x = dev(xyz), b = dev(batch_idxs), o = dev(batch_offsets);
    const size_t wsb = ms3d_ballquery_workspace_bytes(n);
    at::Tensor ws = scratch(wsb, x);
    int n_active = 0, capped = 0;
    check(ms3d_ballquery_batch_p(n, meanActive, radius, x.data_ptr<float>(), b.data_ptr<uint8_t>(), o.data_ptr<int>(),
                                 (int)o.numel() - 1, 0, idx.data_ptr<int>(), start_len.data_ptr<int>(), &n_active, &capped,
                                 ws.data_ptr(), (size_t)ws.numel(), cur()),
          "ms3d_ballquery_batch_p");
    {
        std::lock_guard<std::mutex> guard(g_last_lock);
        g_last.start_len = start_len.data_ptr();
        g_last.n = (int)start_len.size(0);
        g_last.capped = capped;
    }
    return n_active;
}

#define MS3D_SEG(NAME)                                                                                                   \
    void NAME(at::Tensor inp, at::Tensor offsets, at::Tensor out, int nProposal, int C)                                 \
    {                                                                                                                    \
        TORCH_CHECK(out.is_cuda() && out.is_contiguous(), #NAME ": out must be a contiguous device tensor");           \
        at::Tensor i = dev(inp), o = dev(offsets);                                                                      \
        check(ms3d_##NAME(nProposal, C, i.data_ptr<float>(), o.data_ptr<int>(), out.data_ptr<float>(), cur()), "ms3d_" #NAME); \
    }
MS3D_SEG(sec_mean)
MS3D_SEG(sec_min)
MS3D_SEG(sec_max)
#undef MS3D_SEG

void roipool_fp(at::Tensor feats, at::Tensor proposals_offset, at::Tensor output_feats, at::Tensor output_maxidx,
                int nProposal, int C)
{
    at::Tensor f = dev(feats), o = dev(proposals_offset);
    check(ms3d_roipool_fp(nProposal, C, f.data_ptr<float>(), o.data_ptr<int>(), output_feats.data_ptr<float>(),
                          output_maxidx.data_ptr<int>(), cur()), "ms3d_roipool_fp");
}

void roipool_bp(at::Tensor d_feats, at::Tensor proposals_offset, at::Tensor output_maxidx, at::Tensor d_output_feats,
                int nProposal, int C)
{
    at::Tensor o = dev(proposals_offset), m = dev(output_maxidx), d = dev(d_output_feats);
    check(ms3d_roipool_bp(nProposal, C, d_feats.data_ptr<float>(), o.data_ptr<int>(), m.data_ptr<int>(),
                          d.data_ptr<float>(), cur()), "ms3d_roipool_bp");
}

void global_avg_pool_fp(at::Tensor feats, at::Tensor proposals_offset, at::Tensor output_feats, int nProposal, int C)
{
    at::Tensor f = dev(feats), o = dev(proposals_offset);
    check(ms3d_global_avg_pool_fp(nProposal, C, f.data_ptr<float>(), o.data_ptr<int>(), output_feats.data_ptr<float>(),
                                  cur()), "ms3d_global_avg_pool_fp");
}

void global_avg_pool_bp(at::Tensor d_feats, at::Tensor proposals_offset, at::Tensor d_output_feats, int nProposal, int C)
{
    at::Tensor o = dev(proposals_offset), d = dev(d_output_feats);
    check(ms3d_global_avg_pool_bp(nProposal, C, d_feats.data_ptr<float>(), o.data_ptr<int>(), d.data_ptr<float>(), cur()),
          "ms3d_global_avg_pool_bp");
}

void get_iou(at::Tensor proposals_idx, at::Tensor proposals_offset, at::Tensor instance_labels,
             at::Tensor instance_pointnum, at::Tensor proposals_iou, int nInstance, int nProposal)
{
    at::Tensor pi = dev(proposals_idx), po = dev(proposals_offset), il = dev(instance_labels), pn = dev(instance_pointnum);
    check(ms3d_get_iou(nInstance, nProposal, pi.data_ptr<int>(), po.data_ptr<int>(), il.data_ptr<int16_t>(),
                       pn.data_ptr<int>(), proposals_iou.data_ptr<float>(), cur()), "ms3d_get_iou");
}

void get_mask_iou_on_cluster(at::Tensor proposals_idx, at::Tensor proposals_offset, at::Tensor instance_labels,
                             at::Tensor instance_pointnum, at::Tensor proposals_iou, int nInstance, int nProposal)
{
    at::Tensor pi = dev(proposals_idx), po = dev(proposals_offset), il = dev(instance_labels), pn = dev(instance_pointnum);
    check(ms3d_get_mask_iou_on_cluster(nInstance, nProposal, pi.data_ptr<int>(), po.data_ptr<int>(),
                                       il.data_ptr<int16_t>(), pn.data_ptr<int>(), proposals_iou.data_ptr<float>(), cur()),
          "ms3d_get_mask_iou_on_cluster");
}

void get_mask_iou_on_pred(at::Tensor proposals_idx, at::Tensor proposals_offset, at::Tensor instance_labels,
                          at::Tensor instance_pointnum, at::Tensor proposals_iou, int nInstance, int nProposal,
                          at::Tensor mask_scores_sigmoid)
{
    at::Tensor pi = dev(proposals_idx), po = dev(proposals_offset), il = dev(instance_labels), pn = dev(instance_pointnum),
               ms = dev(mask_scores_sigmoid);
    check(ms3d_get_mask_iou_on_pred(nInstance, nProposal, pi.data_ptr<int>(), po.data_ptr<int>(), il.data_ptr<int16_t>(),
                                    pn.data_ptr<int>(), proposals_iou.data_ptr<float>(), ms.data_ptr<float>(), cur()),
          "ms3d_get_mask_iou_on_pred");
}

void get_mask_label(at::Tensor proposals_idx, at::Tensor proposals_offset, at::Tensor instance_labels,
                    at::Tensor instance_cls, at::Tensor proposals_iou, int nInstance, int nProposal, int ignored_label,
                    float iou_thr, at::Tensor mask_labels, at::Tensor mask_labels_mask)
{
    at::Tensor pi = dev(proposals_idx), po = dev(proposals_offset), il = dev(instance_labels), ic = dev(instance_cls),
               iou = dev(proposals_iou);
    check(ms3d_get_mask_label(nInstance, nProposal, ignored_label, iou_thr, pi.data_ptr<int>(), po.data_ptr<int>(),
                              il.data_ptr<int16_t>(), ic.data_ptr<int16_t>(), iou.data_ptr<float>(),
                              (uint8_t *)mask_labels.data_ptr<bool>(), (uint8_t *)mask_labels_mask.data_ptr<bool>(), cur()),
          "ms3d_get_mask_label");
}

// ---------------------------------------------------------------------------------------------- clustering
namespace {
// shared by pg / sg: run one BFS entry point over a graph that may live on the host (uploaded) or on the device
template <class Launch>
void bfs_common(at::Tensor &ball_query_idxs, at::Tensor &start_len, at::Tensor &cluster_idxs, at::Tensor &cluster_offsets,
                int N, Launch launch)
{
    TORCH_CHECK(N == start_len.size(0), "N must be the number of rows of start_len");
    at::Tensor sl = dev(start_len);
    const int hint = start_len.is_cuda() ? capped_hint(sl) : -1;
    at::Tensor bq = dev(ball_query_idxs);
    auto iopt = sl.options().dtype(at::kInt);
    at::Tensor idxs = at::empty({std::max(N, 1), 2}, iopt), offs = at::empty({N + 1}, iopt);
    at::Tensor ws = scratch(ms3d_bfs_workspace_bytes(N), sl);
    int counts[2] = {0, 0};
    check(launch(bq.data_ptr<int>(), (long)bq.numel(), sl.data_ptr<int>(), hint, idxs.data_ptr<int>(), offs.data_ptr<int>(),
                 counts, ws.data_ptr(), (size_t)ws.numel()), "bfs_cluster");
    assign(cluster_idxs, idxs.narrow(0, 0, counts[1]));
    assign(cluster_offsets, offs.narrow(0, 0, counts[0] + 1));
}
}  // namespace

void pg_bfs_cluster(at::Tensor semantic_label, at::Tensor ball_query_idxs, at::Tensor start_len, at::Tensor cluster_idxs,
                    at::Tensor cluster_offsets, const int N, int threshold)
{
    at::Tensor sem = dev(semantic_label);
    TORCH_CHECK(sem.scalar_type() == at::kShort, "semantic_label must be int16");
    bfs_common(ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N,
               [&](const int *bq, long ne, const int *sl, int hint, int *ci, int *co, int *counts, void *ws, size_t wsb) {
                   return ms3d_pg_bfs_cluster(sem.data_ptr<int16_t>(), bq, ne, sl, N, threshold, hint, ci, co, counts, ws, wsb,
                                              cur());
               });
}

void sg_bfs_cluster(at::Tensor class_numpoint_mean, at::Tensor ball_query_idxs, at::Tensor start_len,
                    at::Tensor cluster_idxs, at::Tensor cluster_offsets, const int N, float threshold, const int class_id)
{
    at::Tensor mean = class_numpoint_mean.cpu().to(at::kFloat).contiguous();    // a float32 CPU tensor (softgroup_ops.py:23)
    bfs_common(ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N,
               [&](const int *bq, long ne, const int *sl, int hint, int *ci, int *co, int *counts, void *ws, size_t wsb) {
                   return ms3d_sg_bfs_cluster(mean.data_ptr<float>(), bq, ne, sl, N, threshold, hint, class_id, ci, co, counts,
                                              ws, wsb, cur());
               });
}

void hierarchical_aggregation(at::Tensor semantic_label, at::Tensor coord_shift, at::Tensor batch_idxs,
                              at::Tensor ball_query_idxs, at::Tensor start_len, at::Tensor fragment_idxs,
                              at::Tensor fragment_offsets, at::Tensor fragment_centers, at::Tensor cluster_idxs_kept,
                              at::Tensor cluster_offsets_kept, at::Tensor cluster_centers_kept, at::Tensor primary_idxs,
                              at::Tensor primary_offsets, at::Tensor primary_centers, at::Tensor primary_idxs_post,
                              at::Tensor primary_offsets_post, at::Tensor point_num_avg, at::Tensor radius_avg, const int N,
                              const int using_set_aggr_, const int ignored_label)
{
    (void)ignored_label;   // only the initial value of cc.cls_label, overwritten by the seed's label (.cpp:12-18)
    TORCH_CHECK(N == start_len.size(0), "N must be the number of rows of start_len");
    at::Tensor sem = dev(semantic_label), cs = dev(coord_shift), bi = dev(batch_idxs), sl = dev(start_len);
    const int hint = start_len.is_cuda() ? capped_hint(sl) : -1;
    at::Tensor bq = dev(ball_query_idxs);
    at::Tensor pna = point_num_avg.cpu().to(at::kFloat).contiguous(), ra = radius_avg.cpu().to(at::kFloat).contiguous();
    const int ncls = (int)pna.numel();
    const int M = std::max(N, 1);
    auto iopt = sl.options().dtype(at::kInt);
    auto fopt = sl.options().dtype(at::kFloat);
    at::Tensor ib = at::empty({4, M, 2}, iopt), ob = at::empty({4, M + 1}, iopt), cb = at::empty({3, M, 5}, fopt);
    at::Tensor ws = scratch(ms3d_hais_workspace_bytes(N, ncls), sl);
    int counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int *ip = ib.data_ptr<int>(), *op = ob.data_ptr<int>();
    float *cp = cb.data_ptr<float>();
    const size_t is = (size_t)M * 2, os = (size_t)M + 1, csz = (size_t)M * 5;
    check(ms3d_hierarchical_aggregation_parts(
              sem.data_ptr<int16_t>(), cs.data_ptr<float>(), bi.data_ptr<uint8_t>(), bq.data_ptr<int>(), (long)bq.numel(),
              sl.data_ptr<int>(), N, hint, using_set_aggr_ ? 1 : 0, pna.data_ptr<float>(), ra.data_ptr<float>(), ncls, ip, op, cp,
              ip + is, op + os, cp + csz, ip + 2 * is, op + 2 * os, cp + 2 * csz, ip + 3 * is, op + 3 * os, counts,
              ws.data_ptr(), (size_t)ws.numel(), cur()),
          "ms3d_hierarchical_aggregation_parts");
    const int nk = counts[0], rk = counts[1], npr = counts[2], nf = counts[4], rf = counts[5], rp = counts[6];
    assign(cluster_idxs_kept, ib[0].narrow(0, 0, rk));
    assign(cluster_offsets_kept, ob[0].narrow(0, 0, nk + 1));
    assign(cluster_centers_kept, cb[0].narrow(0, 0, nk));
    assign(primary_idxs, ib[1].narrow(0, 0, rp));
    assign(primary_offsets, ob[1].narrow(0, 0, npr + 1));
    assign(primary_centers, cb[1].narrow(0, 0, npr));
    if (!using_set_aggr_) return;           // the early return at hierarchical_aggregation.cpp:146-148
    assign(fragment_idxs, ib[2].narrow(0, 0, rf));
    assign(fragment_offsets, ob[2].narrow(0, 0, nf + 1));
    assign(fragment_centers, cb[2].narrow(0, 0, nf));
    assign(primary_idxs_post, ib[3].narrow(0, 0, rf + rp));
    assign(primary_offsets_post, ob[3].narrow(0, 0, npr + 1));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "minsu3d COMMON_OPS on MI355X: libminsu3d_hip.so behind the reference's pybind signatures";
    // SoftGroup
    m.def("sg_bfs_cluster", &sg_bfs_cluster, "sg_bfs_cluster");
    m.def("global_avg_pool_fp", &global_avg_pool_fp, "global_avg_pool_fp");
    m.def("global_avg_pool_bp", &global_avg_pool_bp, "global_avg_pool_bp");
    // Common
    m.def("ballquery_batch_p", &ballquery_batch_p, "ballquery_batch_p");
    m.def("sec_mean", &sec_mean, "sec_mean");
    m.def("sec_min", &sec_min, "sec_min");
    m.def("sec_max", &sec_max, "sec_max");
    m.def("roipool_fp", &roipool_fp, "roipool_fp");
    m.def("roipool_bp", &roipool_bp, "roipool_bp");
    m.def("get_iou", &get_iou, "get_iou");
    m.def("get_mask_iou_on_cluster", &get_mask_iou_on_cluster, "get_mask_iou_on_cluster");
    m.def("get_mask_iou_on_pred", &get_mask_iou_on_pred, "get_mask_iou_on_pred");
    m.def("get_mask_label", &get_mask_label, "get_mask_label");
    // PointGroup
    m.def("pg_bfs_cluster", &pg_bfs_cluster, "pg_bfs_cluster");
    // HAIS
    m.def("hierarchical_aggregation", &hierarchical_aggregation, "hierarchical_aggregation");
}
