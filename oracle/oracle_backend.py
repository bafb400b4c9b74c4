"""CPU test double for minsu3d_amd.backend.HipBackend -- TEST INFRASTRUCTURE ONLY.

Same method surface, CPU torch tensors in and out, every operator answered by the oracle
(oracle/liboracle.so + a few lines of torch for the BatchNorm algebra).  Installed with
`minsu3d_amd.backend.set_backend(OracleBackend())` by tests/, by bench.py's cpu_baseline leg and by the
config-1 CPU plumbing run; the product package never imports it.
"""
import numpy as np
import torch

from . import oracle as O


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


class OracleBackend:
    name = "oracle-cpu"

    # ---------------------------------------------------------------- grouping
    def ballquery_batch_p(self, coords, batch_idxs, batch_offsets, radius, meanActive, max_scene_points=0):
        idx, sl = O.ballquery_batch_p(_np(coords), _np(batch_idxs), _np(batch_offsets), radius)
        return _t(idx), _t(sl)

    def pg_bfs_cluster(self, sem, idx, start_len, threshold):
        a, b = O.pg_bfs_cluster(_np(sem), _np(idx), _np(start_len), threshold)
        return _t(a.reshape(-1, 2)), _t(b)

    def sg_bfs_cluster(self, mean, idx, start_len, threshold, class_id):
        a, b = O.sg_bfs_cluster(np.asarray(mean, np.float32), _np(idx), _np(start_len), threshold, class_id)
        return _t(a.reshape(-1, 2)), _t(b)

    def sg_bfs_cluster_batched(self, group_of_point, thr_per_group, idx, start_len):
        """checker for the batched SoftGroup grouping: BFS without threshold, then the per-group float test"""
        a, b = O.sg_bfs_cluster(np.array([-1.0], np.float32), _np(idx), _np(start_len), 0.0, 0)
        g, t = _np(group_of_point), _np(thr_per_group)
        sizes = np.diff(b)
        seeds = a.reshape(-1, 2)[b[:-1], 1] if sizes.size else np.zeros(0, np.int64)
        keep = sizes.astype(np.float32) >= t[g[seeds]] if sizes.size else np.zeros(0, bool)
        rows = np.repeat(keep, sizes)
        out = a.reshape(-1, 2)[rows].copy()
        new_id = np.cumsum(keep) - 1
        out[:, 0] = np.repeat(new_id[keep], sizes[keep])
        off = np.concatenate([[0], np.cumsum(sizes[keep])]).astype(np.int32)
        return _t(out.astype(np.int32)), _t(off)

    def hierarchical_aggregation(self, sem, coord_shift, idx, start_len, batch_idxs, using_set_aggr, pna, ra,
                                 ignored_label):
        a, b = O.hierarchical_aggregation(_np(sem), _np(coord_shift), _np(idx), _np(start_len), _np(batch_idxs),
                                          using_set_aggr, np.asarray(pna, np.float32), np.asarray(ra, np.float32),
                                          ignored_label)
        return _t(a.reshape(-1, 2)), _t(b)

    def sec_mean(self, x, off): return _t(O.sec_mean(_np(x), _np(off)))
    def sec_min(self, x, off): return _t(O.sec_min(_np(x), _np(off)))
    def sec_max(self, x, off): return _t(O.sec_max(_np(x), _np(off)))
    def global_avg_pool_fp(self, x, off): return _t(O.global_avg_pool_fp(_np(x), _np(off)))

    def roipool_fp(self, x, off):
        a, b = O.roipool_fp(_np(x), _np(off))
        return _t(a), _t(b)

    def roipool_bp(self, d_out, off, maxidx, n): return _t(O.roipool_bp(_np(d_out), _np(off), _np(maxidx), n))
    def global_avg_pool_bp(self, d_out, off, n): return _t(O.global_avg_pool_bp(_np(d_out), _np(off), n))
    def scatter_add_rows(self, src, idx, n_rows, max_dup=None):   # max_dup: a scheduling hint of the device backend
        dst = torch.zeros((n_rows, src.size(1)), dtype=torch.float32)
        return dst.index_add_(0, idx, src)

    def get_iou(self, pi, po, il, pn): return _t(O.get_iou(_np(pi), _np(po), _np(il), _np(pn)))
    def get_mask_iou_on_cluster(self, pi, po, il, pn): return _t(O.get_mask_iou_on_cluster(_np(pi), _np(po), _np(il), _np(pn)))
    def get_mask_iou_on_pred(self, pi, po, il, pn, sg): return _t(O.get_mask_iou_on_pred(_np(pi), _np(po), _np(il), _np(pn), _np(sg)))

    def get_mask_label(self, pi, po, il, ic, iou, ignored_label, iou_thr):
        a, b = O.get_mask_label(_np(pi), _np(po), _np(il), _np(ic), _np(iou), ignored_label, iou_thr)
        return _t(a), _t(b)

    # ---------------------------------------------------------------- augmentation
    def elastic(self, xyz, noise, gran, mag):
        from minsu3d_amd.util import transform as T    # the host restatement pinned against the reference's elastic()
        x = _np(xyz).astype(np.float64)
        grids = [T.blur_noise(g) for g in _np(noise).astype(np.float32)]
        return _t(x + np.hstack([T.trilinear(g, gran, x)[:, None] for g in grids]) * mag)

    # ---------------------------------------------------------------- instance post-processing
    def proposal_cross_intersection(self, pair_point, pair_cluster, P):
        from . import postprocess_oracle as PO
        return _t(PO.cross_intersection(_np(pair_point), _np(pair_cluster), P))

    def nms_greedy(self, inter, order, threshold):
        from . import postprocess_oracle as PO
        return _t(PO.nms_from_counts(_np(inter), _np(order), threshold))

    # ---------------------------------------------------------------- coordinates (tables offset-major [K, V])
    def sparse_quantize(self, coords):
        u, inv = O.sparse_quantize(_np(coords))
        return _t(u), _t(inv)

    def spatial_order(self, coords):
        return None   # the CPU checker keeps the caller's row order

    def kmap_k3(self, coords, ts):
        return _t(O.kmap_k3(_np(coords), ts).T)

    def downsample(self, coords, ts):
        oc, par, ko = O.downsample(_np(coords), ts)
        return _t(oc), _t(par), _t(ko)

    def kmap_k2(self, parent, koff, vc):
        d, u = O.kmap_k2(_np(parent), _np(koff), int(vc))
        return _t(d.T), _t(u.T)

    # ---------------------------------------------------------------- convolution
    def prep_weights(self, W, K, cin_e, cout_e, transpose=False, mirror=False):
        W = _np(W).reshape(K, -1, cout_e if not transpose else cin_e)
        if transpose:
            W = np.ascontiguousarray(W.transpose(0, 2, 1))   # [K, cin_e(=cout_o), cout_e(=cin_o)]
        if mirror:
            W = np.ascontiguousarray(W[::-1])
        return np.ascontiguousarray(W, np.float32)

    @staticmethod
    def _act(x, pre, pre_relu):
        if pre is None:
            return x
        a = x * pre[0] + pre[1]
        return torch.relu(a) if pre_relu else a

    def prep_weights_pair(self, W, K, cin, cout, mirror_bwd=False):
        return (self.prep_weights(W, K, cin, cout), self.prep_weights(W, K, cout, cin, transpose=True, mirror=mirror_bwd))

    def identity_table(self, n, device):
        return torch.arange(n, dtype=torch.int32).view(1, n)

    def conv_forward(self, x, wf, nbr, vout, K, cin, cout, pre=None, pre_relu=False, residual=None, bn_bwd=None,
                     out_stats=False, bias=None):
        a = self._act(x.detach(), pre, pre_relu)
        out = _t(O.conv_fwd(_np(a), wf, _np(nbr).T))
        if residual is not None:
            out = out + residual.detach()
        if bias is not None:
            out = out + bias.detach()
        if bn_bwd is None:
            if out_stats:
                return out, torch.stack([out.sum(0), (out * out).sum(0)])[None]
            return out
        bx, scale, shift, mean, invstd = bn_bwd
        dz = out * ((bx * scale + shift) > 0)
        xh = (bx - mean) * invstd
        return dz, torch.stack([dz.sum(0), (dz * xh).sum(0)])

    def conv_layer_forward(self, x, W3, nbr_fwd, vout, K, cin, cout, mirror_bwd, pre, pre_relu, residual, bias,
                           want_stats):
        wf, wft = self.prep_weights_pair(W3, K, cin, cout, mirror_bwd)
        res = self.conv_forward(x, wf, nbr_fwd, vout, K, cin, cout, pre=pre, pre_relu=pre_relu, residual=residual,
                                out_stats=want_stats, bias=bias)
        y, stats = res if want_stats else (res, None)
        return y, stats, (wf, wft)

    @staticmethod
    def fuses_dx_add(bn):
        # as HipBackend: frozen (eval-mode) statistics leave the skip gradient to the caller's fallback
        return bn is None or (bool(bn["relu"]) and bool(bn["training"]))

    def conv_layer_backward(self, x, dy, wf_buf, nbr_fwd, nbr_bwd, vin, vout, K, cin, cout, bn, need_dx, dx_add=None):
        wft = wf_buf[1]
        dx = dgb = None
        if bn is None:
            if need_dx:
                dx = self.conv_forward(dy, wft, nbr_bwd, vin, K, cout, cin)
        else:
            assert bn["relu"]
            dz, dgb = self.conv_forward(dy, wft, nbr_bwd, vin, K, cout, cin,
                                        bn_bwd=(x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"]))
            if need_dx:
                dx = (self.bn_bwd_apply(dz, x, bn["scale"], bn["mean"], bn["invstd"], dgb) if bn["training"]
                      else dz * bn["scale"])
        if dx is not None and dx_add is not None:
            dx = dx + dx_add        # the gradient that reaches x over a skip connection (MinkowskiEngine/functional.py)
        pre = (bn["scale"], bn["shift"]) if bn is not None else None
        dW = self.conv_backward_weight(x, dy, nbr_fwd, vout, K, cin, cout, pre=pre, pre_relu=bool(bn and bn["relu"]))
        return dx, dgb, dW

    def conv_backward_weight(self, x, dout, nbr, vout, K, cin, cout, pre=None, pre_relu=False):
        a = self._act(x.detach(), pre, pre_relu)
        return _t(O.conv_bwd_weight(_np(a), _np(dout), _np(nbr).T, K))

    # ---------------------------------------------------------------- batch norm algebra
    def bn_stats(self, x, eps, momentum, gamma, beta, running_mean, running_var):
        V = x.size(0)
        xd = x.double()
        mean = xd.mean(0)
        var = (xd * xd).mean(0) - mean * mean
        var = var.clamp_min(0)
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        g = gamma if gamma is not None else torch.ones_like(invstd)
        b = beta if beta is not None else torch.zeros_like(invstd)
        scale = g * invstd
        shift = b - mean.float() * scale
        if running_mean is not None:
            unbiased = var * V / max(V - 1, 1)
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * unbiased.float())
        return mean.float(), invstd, scale, shift

    def bn_finalize(self, partial, V, eps, momentum, gamma, beta, running_mean, running_var):
        s = partial.double().sum(0)
        mean = s[0] / V
        var = (s[1] / V - mean * mean).clamp_min(0)
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        g = gamma if gamma is not None else torch.ones_like(invstd)
        b = beta if beta is not None else torch.zeros_like(invstd)
        scale = g * invstd
        shift = b - mean.float() * scale
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
            running_var.mul_(1 - momentum).add_(momentum * (var * V / max(V - 1, 1)).float())
        return mean.float(), invstd, scale, shift

    def bn_apply(self, x, scale, shift, relu):
        y = x * scale + shift
        return torch.relu(y) if relu else y

    def bn_bwd_reduce(self, dy, x, scale, shift, mean, invstd, relu):
        dz = dy * ((x * scale + shift) > 0) if relu else dy.clone()
        xh = (x - mean) * invstd
        return dz, torch.stack([dz.sum(0), (dz * xh).sum(0)])

    def bn_bwd_apply(self, dz, x, scale, mean, invstd, s1s2):
        V = x.size(0)
        xh = (x - mean) * invstd
        return scale * (dz - s1s2[0] / V - xh * s1s2[1] / V)
