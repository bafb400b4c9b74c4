"""numpy/ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (minsu3d_amd) never does and fails loudly without its HIP library.

Two libraries sit behind it:
  * liboracle.so            our plain-C restatement (grouping_oracle.c, sparse_oracle.c)
  * _ref/libminsu3d_ref.so  the REFERENCE's own common_ops code (build_ref.py), when present
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

i32p = C.POINTER(C.c_int)
f32p = C.POINTER(C.c_float)
i16p = C.POINTER(C.c_int16)
u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_ballquery_batch_p.restype = C.c_long
    return _LIB


def ref(required=False):
    """The reference's own code (oracle/_ref).  None when it has not been built."""
    global _REF
    if _REF is None:
        path = os.path.join(HERE, "_ref", "libminsu3d_ref.so")
        if os.path.exists(path):
            import torch  # noqa: F401  (resolves libtorch / libc10 for the shim)
            _REF = C.CDLL(path)
        elif required:
            raise FileNotFoundError(path)
    return _REF


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------ grouping ops
def ballquery_batch_p(xyz, batch_idxs, batch_offsets, radius):
    """canonical form: returns (idx [nActive] i32, start_len [n,2] i32)"""
    xyz = _c(xyz, np.float32); bi = _c(batch_idxs, np.uint8); bo = _c(batch_offsets, np.int32)
    n = xyz.shape[0]
    sl = np.zeros((n, 2), np.int32)
    L = lib()
    total = L.orc_ballquery_batch_p(n, C.c_float(radius), _p(xyz, f32p), _p(bi, u8p), _p(bo, i32p), None,
                                    C.c_long(0), _p(sl, i32p))
    idx = np.zeros(max(total, 1), np.int32)
    L.orc_ballquery_batch_p(n, C.c_float(radius), _p(xyz, f32p), _p(bi, u8p), _p(bo, i32p), _p(idx, i32p),
                            C.c_long(total), _p(sl, i32p))
    return idx[:total], sl


def _bfs_out(n, cap_rows=None):
    cap_rows = n if cap_rows is None else cap_rows
    return np.zeros((max(cap_rows, 1), 2), np.int32), np.zeros(n + 1, np.int32), C.c_int(0)


def pg_bfs_cluster(sem, ball_idx, start_len, threshold, use_ref=False):
    sem = _c(sem, np.int16); bi = _c(ball_idx, np.int32); sl = _c(start_len, np.int32)
    n = sl.shape[0]
    oi, oo, s = _bfs_out(n)
    if use_ref:
        nc = ref(True).ref_pg_bfs_cluster(_p(sem, i16p), _p(bi, i32p), C.c_long(bi.size), _p(sl, i32p), n,
                                          int(threshold), _p(oi, i32p), _p(oo, i32p), C.byref(s))
    else:
        nc = lib().orc_pg_bfs_cluster(_p(sem, i16p), _p(bi, i32p), _p(sl, i32p), n, int(threshold),
                                      _p(oi, i32p), _p(oo, i32p), C.byref(s))
    return oi[:s.value].copy(), oo[:nc + 1].copy()


def sg_bfs_cluster(class_numpoint_mean, ball_idx, start_len, threshold, class_id, use_ref=False):
    m = _c(class_numpoint_mean, np.float32); bi = _c(ball_idx, np.int32); sl = _c(start_len, np.int32)
    n = sl.shape[0]
    oi, oo, s = _bfs_out(n)
    if use_ref:
        nc = ref(True).ref_sg_bfs_cluster(_p(m, f32p), m.size, _p(bi, i32p), C.c_long(bi.size), _p(sl, i32p), n,
                                          C.c_float(threshold), int(class_id), _p(oi, i32p), _p(oo, i32p),
                                          C.byref(s))
    else:
        nc = lib().orc_sg_bfs_cluster(_p(m, f32p), _p(bi, i32p), _p(sl, i32p), n, C.c_float(threshold),
                                      int(class_id), _p(oi, i32p), _p(oo, i32p), C.byref(s))
    return oi[:s.value].copy(), oo[:nc + 1].copy()


def hierarchical_aggregation(sem, coord_shift, ball_idx, start_len, batch_idxs, using_set_aggr, point_num_avg,
                             radius_avg, ignored_label=-1, use_ref=False, parts=False):
    """-> (cluster_idxs [S,2], cluster_offsets) after the wrapper's merge (hais_ops.py:55-73); with parts=True the raw
    lists hierarchical_aggregation.cpp:133-175 leaves in the caller's tensors instead: a dict of
    'fragment' / 'kept' / 'primary' -> (idxs [S,2], offsets, centres [P,5]) and 'post' -> (idxs, offsets)"""
    sem = _c(sem, np.int16); cs = _c(coord_shift, np.float32); bi = _c(ball_idx, np.int32)
    sl = _c(start_len, np.int32); b = _c(batch_idxs, np.uint8)
    pna = _c(point_num_avg, np.float32); ra = _c(radius_avg, np.float32)
    n = sl.shape[0]
    if parts:
        return _ha_parts(sem, cs, b, bi, sl, n, using_set_aggr, pna, ra)
    if not use_ref:
        oi, oo, s = _bfs_out(n, 2 * n)
        nc = lib().orc_hierarchical_aggregation(_p(sem, i16p), _p(cs, f32p), _p(b, u8p), _p(bi, i32p),
                                                _p(sl, i32p), n, int(bool(using_set_aggr)), _p(pna, f32p),
                                                _p(ra, f32p), _p(oi, i32p), _p(oo, i32p), C.byref(s))
        return oi[:s.value].copy(), oo[:nc + 1].copy()
    ki, ko, ks = _bfs_out(n)
    pi, po, ps = _bfs_out(n, 2 * n)
    npr = C.c_int(0)
    nk = ref(True).ref_hierarchical_aggregation(
        _p(sem, i16p), _p(cs, f32p), _p(b, u8p), _p(bi, i32p), C.c_long(bi.size), _p(sl, i32p), n,
        int(bool(using_set_aggr)), _p(pna, f32p), _p(ra, f32p), pna.size, int(ignored_label), _p(ki, i32p),
        _p(ko, i32p), C.byref(ks), _p(pi, i32p), _p(po, i32p), C.byref(ps), C.byref(npr))
    ki, ko = ki[:ks.value].copy(), ko[:nk + 1].copy()
    pi, po = pi[:ps.value].copy(), po[:npr.value + 1].copy()
    # merge exactly like minsu3d/common_ops/functions/hais_ops.py:63-73
    if pi.shape[0] != 0:
        pi[:, 0] += ko.size - 1
        po = po + ko[-1]
        ki = np.concatenate([ki, pi], 0)
        ko = np.concatenate([ko, po[1:]])
    return ki, ko


class _HaParts(C.Structure):
    _fields_ = [(k, t) for grp in ("frag", "kept", "prim") for k, t in
                ((grp + "_idx", i32p), (grp + "_off", i32p), (grp + "_ctr", f32p), ("n_" + grp, C.c_int),
                 (grp + "_sum", C.c_int))] + \
               [("post_idx", i32p), ("post_off", i32p), ("n_post", C.c_int), ("post_sum", C.c_int)]


def _ha_parts(sem, cs, b, bi, sl, n, using_set_aggr, pna, ra):
    oi, oo, s = _bfs_out(n, 2 * n)
    bufs, P = {}, _HaParts()
    for grp in ("frag", "kept", "prim", "post"):
        bufs[grp] = (np.zeros((2 * max(n, 1), 2), np.int32), np.zeros(n + 1, np.int32), np.zeros((max(n, 1), 5), np.float32))
        setattr(P, grp + "_idx", _p(bufs[grp][0], i32p)); setattr(P, grp + "_off", _p(bufs[grp][1], i32p))
        if grp != "post":
            setattr(P, grp + "_ctr", _p(bufs[grp][2], f32p))
    lib().orc_hierarchical_aggregation_parts(_p(sem, i16p), _p(cs, f32p), _p(b, u8p), _p(bi, i32p), _p(sl, i32p), n,
                                             int(bool(using_set_aggr)), _p(pna, f32p), _p(ra, f32p), _p(oi, i32p),
                                             _p(oo, i32p), C.byref(s), C.byref(P))
    out = {}
    for grp, name in (("frag", "fragment"), ("kept", "kept"), ("prim", "primary"), ("post", "post")):
        k, m = getattr(P, "n_" + grp), getattr(P, grp + "_sum")
        out[name] = (bufs[grp][0][:m].copy(), bufs[grp][1][:k + 1].copy()) + \
            ((bufs[grp][2][:k].copy(),) if grp != "post" else ())
    return out


def _seg(name, inp, offsets):
    inp = _c(inp, np.float32); off = _c(offsets, np.int32)
    P, Cc = off.size - 1, inp.shape[1]
    out = np.zeros((P, Cc), np.float32)
    getattr(lib(), name)(P, Cc, _p(inp, f32p), _p(off, i32p), _p(out, f32p))
    return out


def sec_mean(inp, offsets): return _seg("orc_sec_mean", inp, offsets)
def sec_min(inp, offsets): return _seg("orc_sec_min", inp, offsets)
def sec_max(inp, offsets): return _seg("orc_sec_max", inp, offsets)
def global_avg_pool_fp(inp, offsets): return _seg("orc_global_avg_pool_fp", inp, offsets)


def roipool_fp(feats, offsets):
    feats = _c(feats, np.float32); off = _c(offsets, np.int32)
    P, Cc = off.size - 1, feats.shape[1]
    out = np.zeros((P, Cc), np.float32); mi = np.zeros((P, Cc), np.int32)
    lib().orc_roipool_fp(P, Cc, _p(feats, f32p), _p(off, i32p), _p(out, f32p), _p(mi, i32p))
    return out, mi


def roipool_bp(d_out, offsets, maxidx, sum_npoint):
    d_out = _c(d_out, np.float32); off = _c(offsets, np.int32); mi = _c(maxidx, np.int32)
    P, Cc = d_out.shape
    d_feats = np.zeros((sum_npoint, Cc), np.float32)
    lib().orc_roipool_bp(P, Cc, _p(d_feats, f32p), _p(off, i32p), _p(mi, i32p), _p(d_out, f32p))
    return d_feats


def global_avg_pool_bp(d_out, offsets, sum_npoint):
    d_out = _c(d_out, np.float32); off = _c(offsets, np.int32)
    P, Cc = d_out.shape
    d_feats = np.zeros((sum_npoint, Cc), np.float32)
    lib().orc_global_avg_pool_bp(P, Cc, _p(d_feats, f32p), _p(off, i32p), _p(d_out, f32p))
    return d_feats


def _iou(name, prop_idx, prop_off, inst_labels, inst_pointnum, sigmoid=None):
    pi = _c(prop_idx, np.int32); po = _c(prop_off, np.int32); il = _c(inst_labels, np.int16)
    pn = _c(inst_pointnum, np.int32)
    P, I = po.size - 1, pn.size
    iou = np.zeros((P, I), np.float32)
    args = [I, P, _p(pi, i32p), _p(po, i32p), _p(il, i16p), _p(pn, i32p), _p(iou, f32p)]
    if sigmoid is not None:
        sg = _c(sigmoid, np.float32)
        args.append(_p(sg, f32p))
    getattr(lib(), name)(*args)
    return iou


def get_iou(pi, po, il, pn): return _iou("orc_get_iou", pi, po, il, pn)
def get_mask_iou_on_cluster(pi, po, il, pn): return _iou("orc_get_mask_iou_on_cluster", pi, po, il, pn)
def get_mask_iou_on_pred(pi, po, il, pn, sg): return _iou("orc_get_mask_iou_on_pred", pi, po, il, pn, sg)


def get_mask_label(prop_idx, prop_off, inst_labels, inst_cls, iou, ignored_label, iou_thr):
    pi = _c(prop_idx, np.int32); po = _c(prop_off, np.int32); il = _c(inst_labels, np.int16)
    ic = _c(inst_cls, np.int16); iou = _c(iou, np.float32)
    P, I = iou.shape
    ml = np.zeros(pi.size, np.uint8); mlm = np.zeros(pi.size, np.uint8)
    lib().orc_get_mask_label(I, P, int(ignored_label), C.c_float(iou_thr), _p(pi, i32p), _p(po, i32p),
                             _p(il, i16p), _p(ic, i16p), _p(iou, f32p), _p(ml, u8p), _p(mlm, u8p))
    return ml.astype(bool), mlm.astype(bool)


# ------------------------------------------------------------------ sparse engine (ME subset)
def sparse_quantize(coords):
    """coords int32 [N,4] -> (unique_idx [U], inverse [N]); first occurrence wins."""
    c = _c(coords, np.int32)
    n = c.shape[0]
    ui = np.zeros(max(n, 1), np.int32); inv = np.zeros(max(n, 1), np.int32)
    nu = lib().orc_sparse_quantize(_p(c, i32p), n, _p(ui, i32p), _p(inv, i32p))
    return ui[:nu].copy(), inv[:n].copy()


def kmap_k3(coords, ts):
    c = _c(coords, np.int32)
    V = c.shape[0]
    nbr = np.zeros((max(V, 1), 27), np.int32)
    lib().orc_kmap_k3(_p(c, i32p), V, int(ts), _p(nbr, i32p))
    return nbr[:V]


def downsample(coords, ts):
    c = _c(coords, np.int32)
    V = c.shape[0]
    oc = np.zeros((max(V, 1), 4), np.int32); par = np.zeros(max(V, 1), np.int32); ko = np.zeros(max(V, 1), np.int32)
    vc = lib().orc_downsample(_p(c, i32p), V, int(ts), _p(oc, i32p), _p(par, i32p), _p(ko, i32p))
    return oc[:vc].copy(), par[:V].copy(), ko[:V].copy()


def kmap_k2(parent, koff, vc):
    par = _c(parent, np.int32); ko = _c(koff, np.int32)
    vf = par.size
    down = np.zeros((max(vc, 1), 8), np.int32); up = np.zeros((max(vf, 1), 8), np.int32)
    lib().orc_kmap_k2(_p(par, i32p), _p(ko, i32p), vf, vc, _p(down, i32p), _p(up, i32p))
    return down[:vc], up[:vf]


def conv_fwd(x, W, nbr):
    x = _c(x, np.float32); W = _c(W, np.float32); nbr = _c(nbr, np.int32)
    K, Cin, Cout = W.shape
    Vout = nbr.shape[0]
    out = np.zeros((Vout, Cout), np.float32)
    lib().orc_conv_fwd(_p(x, f32p), _p(W, f32p), _p(nbr, i32p), Vout, K, Cin, Cout, _p(out, f32p))
    return out


def conv_bwd_data(dout, W, nbr, vin):
    dout = _c(dout, np.float32); W = _c(W, np.float32); nbr = _c(nbr, np.int32)
    K, Cin, Cout = W.shape
    din = np.zeros((vin, Cin), np.float32)
    lib().orc_conv_bwd_data(_p(dout, f32p), _p(W, f32p), _p(nbr, i32p), nbr.shape[0], K, Cin, Cout, _p(din, f32p))
    return din


def conv_bwd_weight(x, dout, nbr, K):
    x = _c(x, np.float32); dout = _c(dout, np.float32); nbr = _c(nbr, np.int32)
    Cin, Cout = x.shape[1], dout.shape[1]
    dW = np.zeros((K, Cin, Cout), np.float32)
    lib().orc_conv_bwd_weight(_p(x, f32p), _p(dout, f32p), _p(nbr, i32p), nbr.shape[0], K, Cin, Cout, _p(dW, f32p))
    return dW
