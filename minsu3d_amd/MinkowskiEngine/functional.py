"""autograd glue around the engine kernels.  Two functions carry the whole backbone:

  SparseConvFn : y = conv(act(x)),  act = the (lazy) BatchNorm[+ReLU] in front of the convolution, applied in the
                 gather.  Backward = one backward-data launch whose epilogue applies the ReLU mask and collects the
                 BN-backward channel sums, one elementwise BN-backward pass, one backward-weight launch.
  BNActFn      : stand-alone BN(+ReLU) for tensors whose consumer is not a convolution.
"""
import torch

from ..backend import get_backend


class ConvSpec:
    """non-tensor launch description shared by forward and backward"""
    __slots__ = ("nbr_fwd", "nbr_bwd", "vin", "vout", "K", "cin", "cout", "mirror")

    def __init__(self, nbr_fwd, nbr_bwd, vin, vout, K, cin, cout, mirror):
        self.nbr_fwd, self.nbr_bwd, self.vin, self.vout = nbr_fwd, nbr_bwd, vin, vout
        self.K, self.cin, self.cout, self.mirror = K, cin, cout, mirror


class SkipLink:
    """The gradient of a residual connection, handed from the LAST convolution of the block (whose epilogue added the
    skip) to the FIRST one (whose input is the skipped tensor): `out = x + branch(x)` sends dy to x twice -- through the
    branch and directly -- and autograd would add the two with an elementwise kernel per block and step.  The tail keeps
    its dy here instead of returning it for `residual`; the head's backward-data pass adds it to dx inside the
    BatchNorm-backward kernel it runs anyway (same two operands, same sum)."""
    __slots__ = ("grad", "armed")

    def __init__(self):
        self.grad = None
        self.armed = False     # set by the head's forward: only then may the tail keep its dy back


class GroupFlushFn(torch.autograd.Function):
    """Identity on a GROUP of convolution kernels whose backward runs once, after the last layer of the group has produced
    its weight gradient: there the group's queued backward-weight slab reductions are launched as ONE kernel
    (backend.WgradQueue) and the -- now complete -- gradients go on to the parameters.  Everything downstream of a
    parameter gradient (accumulation into an existing .grad, DistributedDataParallel's bucket hooks, the optimizer) sits
    behind this node, so it never sees an unreduced dW.  Built per forward by modules.prepare_conv_weights."""

    @staticmethod
    def forward(ctx, queue, *kernels):
        ctx.queue = queue
        ctx.set_materialize_grads(False)
        return tuple(k.view_as(k) for k in kernels)

    @staticmethod
    def backward(ctx, *grads):
        ctx.queue.flush()
        return (None,) + grads


def _defer_of(W, tok):
    """The deferred-reduction stamp of a kernel alias for this forward (prepare_conv_weights) -> (queue, use counter) or
    None, counting this use.  The deferral hands autograd a dW that is only filled when the group's GroupFlushFn node
    runs, which is sound while nothing computes on dW in between: a SECOND convolution on the same alias in one forward
    (weight sharing, a module called in a loop) makes autograd's input buffer add the two still-unwritten tensors before
    the flush.  The decision is therefore taken at BACKWARD time, when the count is final: `_queue`."""
    d = getattr(W, "_ms3d_defer", None)
    if d is None or d[1] != tok:
        return None
    d[2][0] += 1
    return (d[0], d[2])


def _queue(defer):
    """the queue a backward may defer to: only when its kernel alias was used exactly once in the forward"""
    return defer[0] if (defer is not None and defer[1][0] == 1) else None


class SparseConvFn(torch.autograd.Function):
    """(y, stats) = conv(act(x)) [+ residual];  stats = per-block (sum, sum of squares) of y from the kernel epilogue
    (non-differentiable side output that lets the next BatchNorm skip its statistics pass)"""

    @staticmethod
    def forward(ctx, x, W, gamma, beta, residual, spec, bn, want_stats, skip=None):
        be = get_backend()
        pre = (bn["scale"], bn["shift"]) if bn is not None else None
        # weight images laid out for the whole model in one launch at the start of its forward (modules.prepare_conv_weights):
        # the stamp on the parameter says whether that buffer still holds the current weights
        ready = getattr(W, "_ms3d_wf", None)
        ready = ready[0] if (ready is not None and ready[1] == getattr(be, "weight_token", None)) else None
        y, stats, wf_buf = be.conv_layer_forward(x, W.view(spec.K, spec.cin, spec.cout), spec.nbr_fwd, spec.vout, spec.K,
                                                 spec.cin, spec.cout, spec.mirror, pre, bool(bn and bn["relu"]), residual,
                                                 None, want_stats, **({"wf_ready": ready} if ready is not None else {}))
        if stats is None:
            stats = x.new_zeros(0)
        ctx.spec, ctx.bn, ctx.wf_buf, ctx.has_res = spec, bn, wf_buf, residual is not None
        # (deferred slab reduction: the queue of this kernel's group, stamped by prepare_conv_weights for this forward)
        ctx.defer = _defer_of(W, getattr(be, "weight_token", None))
        ctx.w_direct = W.is_leaf or ctx.defer is not None    # nobody computes on dW before the parameter / the flush node
        if skip is not None:     # None | ("head", SkipLink) | ("tail", SkipLink)
            if skip[0] == "head":
                skip[1].armed = True
            elif not (skip[1].armed and residual is not None):
                skip = None      # no head took the link: the skip gradient goes the ordinary way
        ctx.skip = skip
        ctx.save_for_backward(x, W)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        be = get_backend()
        spec, bn = ctx.spec, ctx.bn
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        role, link = ctx.skip if ctx.skip is not None else (None, None)
        add = None
        if role == "head":
            add, link.grad = link.grad, None
        if bn is not None and not bn["relu"]:
            # BatchNorm without ReLU in front of a convolution (not used by the reference): unfused path
            wft = be.prep_weights(W.view(spec.K, spec.cin, spec.cout), spec.K, spec.cout, spec.cin, transpose=True,
                                  mirror=spec.mirror)
            da = be.conv_forward(dy, wft, spec.nbr_bwd, spec.vin, spec.K, spec.cout, spec.cin)
            dz, dgb = be.bn_bwd_reduce(da, x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"], False)
            dx = (be.bn_bwd_apply(dz, x, bn["scale"], bn["mean"], bn["invstd"], dgb) if bn["training"] else dz * bn["scale"])
            dW = be.conv_backward_weight(x, dy, spec.nbr_fwd, spec.vout, spec.K, spec.cin, spec.cout,
                                         pre=(bn["scale"], bn["shift"]), pre_relu=False)
        else:
            fused = add is not None and ctx.needs_input_grad[0] and be.fuses_dx_add(bn)
            extra = {"dx_add": add} if fused else {}
            if _queue(ctx.defer) is not None:
                extra["defer"] = _queue(ctx.defer)
            elif (not ctx.w_direct or (W.is_leaf and W.grad is not None)) and hasattr(be, "wgrad_stream_mode") \
                    and be.wgrad_stream_mode() == 2:
                # MS3D_WGRAD_STREAM=2 only: dW is consumed on this stream right away (a slice of a padded kernel's
                # gradient, the accumulation into an existing .grad): this layer's backward-weight is joined now
                extra["join_now"] = True
            dx, dgb, dW = be.conv_layer_backward(x, dy, ctx.wf_buf, spec.nbr_fwd, spec.nbr_bwd, spec.vin, spec.vout,
                                                 spec.K, spec.cin, spec.cout, bn, ctx.needs_input_grad[0], **extra)
            if fused:
                add = None
        if add is not None and dx is not None:
            dx = dx + add
        dgamma = dgb[1] if dgb is not None else None
        dbeta = dgb[0] if dgb is not None else None
        d_res = dy if ctx.has_res else None
        if role == "tail" and ctx.has_res:
            link.grad, d_res = dy, None      # the head adds it to its dx
        return dx, dW.view_as(W), dgamma, dbeta, d_res, None, None, None, None


class ResBlockFn(torch.autograd.Function):
    """A whole identity-skip residual block in training mode -- out = x + conv2(relu(bn1(conv1(relu(bn0(x)))))) -- as ONE
    autograd node over the same four library calls the module chain makes (bn_finalize, conv, bn_finalize, conv with the
    residual in its epilogue; backward: the two fused backward calls, the skip gradient added inside the first one's
    BatchNorm-backward pass, exactly what SparseConvFn + SkipLink do).  Nothing is computed differently: what goes away is
    interpreter work -- five nn.Module calls, four SparseTensor wrappers, one autograd node and the BatchNorm modules'
    bookkeeping per block (~150 -> ~55 us of host time; 21 of the 28 blocks of the backbone and 4 of the 6 of a proposal
    network qualify), which is what the lock-step phases of a step and a slow host wait for."""

    @staticmethod
    def forward(ctx, x, W1, g0, b0, W2, g1, b1, spec, stats_in, bn0, bn1, want_stats):
        be = get_backend()
        V = spec.vout
        tok = getattr(be, "weight_token", None)
        r1 = getattr(W1, "_ms3d_wf", None)
        r1 = r1[0] if (r1 is not None and r1[1] == tok) else None
        r2 = getattr(W2, "_ms3d_wf", None)
        r2 = r2[0] if (r2 is not None and r2[1] == tok) else None
        fast = None
        if (r1 is not None and r2 is not None and spec.cin == spec.cout and torch.is_tensor(stats_in) and x.is_contiguous()
                and hasattr(be, "res_block_forward")):
            # round 6: the four calls below from ONE host call (same library calls, same order, same stream)
            fast = be.res_block_forward(x, stats_in, r1, r2, spec.nbr_fwd, V, spec.cout, bn0, g0.detach(), b0.detach(), bn1,
                                        g1.detach(), b1.detach(), want_stats)
        if fast is not None:
            y1, y2, st2, (m0, i0, s0, h0), (m1, i1, s1, h1) = fast
            wf1, wf2 = r1, r2
        else:
            m0, i0, s0, h0 = be.bn_finalize(stats_in, V, bn0.eps, bn0.momentum, g0.detach(), b0.detach(), bn0.running_mean,
                                            bn0.running_var)
            y1, st1, wf1 = be.conv_layer_forward(x, W1, spec.nbr_fwd, V, 27, spec.cin, spec.cout, True, (s0, h0), True, None,
                                                 None, True, **({"wf_ready": r1} if r1 is not None else {}))
            m1, i1, s1, h1 = be.bn_finalize(st1, V, bn1.eps, bn1.momentum, g1.detach(), b1.detach(), bn1.running_mean,
                                            bn1.running_var)
            y2, st2, wf2 = be.conv_layer_forward(y1, W2, spec.nbr_fwd, V, 27, spec.cout, spec.cout, True, (s1, h1), True, x,
                                                 None, want_stats, **({"wf_ready": r2} if r2 is not None else {}))
        ctx.spec, ctx.wf = spec, (wf1, wf2)
        ctx.bn = (dict(scale=s0, shift=h0, mean=m0, invstd=i0, relu=True, training=True),
                  dict(scale=s1, shift=h1, mean=m1, invstd=i1, relu=True, training=True))
        ctx.defer = (_defer_of(W1, tok), _defer_of(W2, tok))
        ctx.save_for_backward(x, y1, W1, W2)
        if st2 is None:
            st2 = x.new_zeros(0)
        ctx.mark_non_differentiable(st2)
        ctx.set_materialize_grads(False)
        return y2, st2

    @staticmethod
    def backward(ctx, dy, _dstats):
        be = get_backend()
        spec = ctx.spec
        x, y1, W1, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        V, c0, c1 = spec.vout, spec.cin, spec.cout
        e2 = {"defer": _queue(ctx.defer[1])} if _queue(ctx.defer[1]) is not None else {}
        dx1, dgb1, dW2 = be.conv_layer_backward(y1, dy, ctx.wf[1], spec.nbr_fwd, spec.nbr_bwd, V, V, 27, c1, c1, ctx.bn[1], True, **e2)
        e1 = {"defer": _queue(ctx.defer[0])} if _queue(ctx.defer[0]) is not None else {}
        dx, dgb0, dW1 = be.conv_layer_backward(x, dx1, ctx.wf[0], spec.nbr_fwd, spec.nbr_bwd, V, V, 27, c0, c1, ctx.bn[0],
                                               ctx.needs_input_grad[0], dx_add=dy, **e1)
        return dx, dW1.view_as(W1), dgb0[1], dgb0[0], dW2.view_as(W2), dgb1[1], dgb1[0], None, None, None, None, None


class ResBlockDownFn(torch.autograd.Function):
    """The residual block with a 1x1 projection on its skip path (the first block behind a skip concatenation):
    out = x W_d + conv2(relu(bn1(conv1(relu(bn0(x)))))) as one autograd node -- the projection (K = 1 convolution), the
    two bn_finalize (bn0 over the concatenation's parts) and the two fused convolutions forward; backward: the two fused
    backward calls of the branch, then the projection's, whose residual epilogue adds the branch's dx (the module chain
    leaves that sum to an elementwise autograd kernel)."""

    @staticmethod
    def forward(ctx, x, Wd, W1, g0, b0, W2, g1, b1, spec, ident, stats_in, bn0, bn1):
        be = get_backend()
        V, cin, cout = spec.vout, spec.cin, spec.cout
        tok = getattr(be, "weight_token", None)

        def ready(W):
            r = getattr(W, "_ms3d_wf", None)
            return {"wf_ready": r[0]} if (r is not None and r[1] == tok) else {}

        skip, _, wfd = be.conv_layer_forward(x, Wd.view(1, cin, cout), ident, V, 1, cin, cout, False, None, False, None, None,
                                             False, **ready(Wd))
        if isinstance(stats_in, tuple):
            m0, i0, s0, h0 = be.bn_finalize_parts(stats_in, V, bn0.eps, bn0.momentum, g0.detach(), b0.detach(),
                                                  bn0.running_mean, bn0.running_var)
        else:
            m0, i0, s0, h0 = be.bn_finalize(stats_in, V, bn0.eps, bn0.momentum, g0.detach(), b0.detach(), bn0.running_mean,
                                            bn0.running_var)
        y1, st1, wf1 = be.conv_layer_forward(x, W1, spec.nbr_fwd, V, 27, cin, cout, True, (s0, h0), True, None, None, True,
                                             **ready(W1))
        m1, i1, s1, h1 = be.bn_finalize(st1, V, bn1.eps, bn1.momentum, g1.detach(), b1.detach(), bn1.running_mean,
                                        bn1.running_var)
        y2, st2, wf2 = be.conv_layer_forward(y1, W2, spec.nbr_fwd, V, 27, cout, cout, True, (s1, h1), True, skip, None, True,
                                             **ready(W2))
        ctx.spec, ctx.wf, ctx.ident = spec, (wfd, wf1, wf2), ident
        ctx.bn = (dict(scale=s0.contiguous(), shift=h0.contiguous(), mean=m0.contiguous(), invstd=i0.contiguous(), relu=True,
                       training=True),
                  dict(scale=s1, shift=h1, mean=m1, invstd=i1, relu=True, training=True))

        ctx.defer = (_defer_of(Wd, tok), _defer_of(W1, tok), _defer_of(W2, tok))
        ctx.save_for_backward(x, y1, Wd, W1, W2)
        ctx.mark_non_differentiable(st2)
        ctx.set_materialize_grads(False)
        return y2, st2

    @staticmethod
    def backward(ctx, dy, _dstats):
        be = get_backend()
        spec = ctx.spec
        x, y1, Wd, W1, W2 = ctx.saved_tensors
        dy = dy.contiguous()
        V, cin, cout = spec.vout, spec.cin, spec.cout
        ex = lambda d: ({"defer": _queue(d)} if _queue(d) is not None else {})
        dx1, dgb1, dW2 = be.conv_layer_backward(y1, dy, ctx.wf[2], spec.nbr_fwd, spec.nbr_bwd, V, V, 27, cout, cout, ctx.bn[1],
                                                True, **ex(ctx.defer[2]))
        need = ctx.needs_input_grad[0]
        dxb, dgb0, dW1 = be.conv_layer_backward(x, dx1, ctx.wf[1], spec.nbr_fwd, spec.nbr_bwd, V, V, 27, cin, cout, ctx.bn[0],
                                                need, **ex(ctx.defer[1]))
        dx, _, dWd = be.conv_layer_backward(x, dy, ctx.wf[0], ctx.ident, ctx.ident, V, V, 1, cin, cout, None, need,
                                            **({"dx_add": dxb} if need else {}), **ex(ctx.defer[0]))
        return dx, dWd.view_as(Wd), dW1.view_as(W1), dgb0[1], dgb0[0], dW2.view_as(W2), dgb1[1], dgb1[0], None, None, None, None, None


class BnReluConvFn(torch.autograd.Function):
    """[BatchNorm, ReLU, convolution] in training mode -- the strided / transposed convolutions between the U-Net levels --
    as one autograd node: bn_finalize + the fused convolution forward, the fused backward call (see ResBlockFn: the same
    library calls as the module chain, without its interpreter work)."""

    @staticmethod
    def forward(ctx, x, W, g, b, spec, stats_in, bn, want_stats):
        be = get_backend()
        tok = getattr(be, "weight_token", None)
        if isinstance(stats_in, tuple):
            m0, i0, s0, h0 = be.bn_finalize_parts(stats_in, spec.vin, bn.eps, bn.momentum, g.detach(), b.detach(),
                                                  bn.running_mean, bn.running_var)
        else:
            m0, i0, s0, h0 = be.bn_finalize(stats_in, spec.vin, bn.eps, bn.momentum, g.detach(), b.detach(), bn.running_mean,
                                            bn.running_var)
        r = getattr(W, "_ms3d_wf", None)
        r = r[0] if (r is not None and r[1] == tok) else None
        y, st, wf = be.conv_layer_forward(x, W.view(spec.K, spec.cin, spec.cout), spec.nbr_fwd, spec.vout, spec.K, spec.cin,
                                          spec.cout, spec.mirror, (s0, h0), True, None, None, want_stats,
                                          **({"wf_ready": r} if r is not None else {}))
        ctx.spec, ctx.wf = spec, wf
        ctx.bn = dict(scale=s0.contiguous(), shift=h0.contiguous(), mean=m0.contiguous(), invstd=i0.contiguous(), relu=True,
                      training=True)
        ctx.defer = _defer_of(W, tok)
        ctx.save_for_backward(x, W)
        if st is None:
            st = x.new_zeros(0)
        ctx.mark_non_differentiable(st)
        ctx.set_materialize_grads(False)
        return y, st

    @staticmethod
    def backward(ctx, dy, _dstats):
        be = get_backend()
        spec = ctx.spec
        x, W = ctx.saved_tensors
        e = {"defer": _queue(ctx.defer)} if _queue(ctx.defer) is not None else {}
        dx, dgb, dW = be.conv_layer_backward(x, dy.contiguous(), ctx.wf, spec.nbr_fwd, spec.nbr_bwd, spec.vin, spec.vout, spec.K,
                                             spec.cin, spec.cout, ctx.bn, ctx.needs_input_grad[0], **e)
        return dx, dW.view_as(W), dgb[1], dgb[0], None, None, None, None


class BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, bn):
        ctx.bn = bn
        ctx.save_for_backward(x)
        return get_backend().bn_apply(x, bn["scale"], bn["shift"], bn["relu"])

    @staticmethod
    def backward(ctx, dy):
        be = get_backend()
        bn = ctx.bn
        (x,) = ctx.saved_tensors
        dz, s1s2 = be.bn_bwd_reduce(dy.contiguous(), x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"], bn["relu"])
        dx = be.bn_bwd_apply(dz, x, bn["scale"], bn["mean"], bn["invstd"], s1s2) if bn["training"] else dz * bn["scale"]
        return dx, s1s2[1], s1s2[0], None


def bn_act(x, pending):
    if pending.get("gamma") is None:   # plain ReLU without a BatchNorm in front
        return torch.relu(x)
    return BNActFn.apply(x, pending["gamma"], pending["beta"], pending)


def conv(x, W, spec, pending, residual=None, want_stats=False, skip=None):
    """-> (features, stats or None).  skip: ("head" | "tail", SkipLink) of a residual block, see SkipLink"""
    if pending is not None and pending.get("gamma") is None:
        x, pending = torch.relu(x), None
        if skip is not None and skip[0] == "head":
            skip = None       # x is no longer the skipped tensor itself: leave that block to autograd
    if pending is None:
        y, st = SparseConvFn.apply(x, W, None, None, residual, spec, None, want_stats, skip)
    else:
        y, st = SparseConvFn.apply(x, W, pending["gamma"], pending["beta"], residual, spec, pending, want_stats, skip)
    return y, (st if want_stats else None)


class DenseLinearFn(torch.autograd.Function):
    """y = x @ weight.T + bias for tall-skinny per-point matrices (N ~ 10^5..10^6, 16..32 channels) through the
    K = 1 path of the conv kernels: one coalesced pass over x instead of a library GEMM tuned for square shapes."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        be = get_backend()
        n, cin, cout = x.size(0), weight.size(1), weight.size(0)
        W3 = weight.t().contiguous().view(1, cin, cout)
        wf, wft = be.prep_weights_pair(W3, 1, cin, cout, mirror_bwd=False)
        ident = be.identity_table(n, x.device)
        y = be.conv_forward(x.contiguous(), wf, ident, n, 1, cin, cout, bias=bias)
        ctx.wft, ctx.dims, ctx.has_bias = wft, (n, cin, cout), bias is not None
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = get_backend()
        (x,) = ctx.saved_tensors
        n, cin, cout = ctx.dims
        dy = dy.contiguous()
        ident = be.identity_table(n, x.device)
        dx = be.conv_forward(dy, ctx.wft, ident, n, 1, cout, cin) if ctx.needs_input_grad[0] else None
        dW = be.conv_backward_weight(x, dy, ident, n, 1, cin, cout).view(cin, cout).t()
        db = None
        if ctx.has_bias:
            db = be.column_sum(dy) if (hasattr(be, "column_sum") and dy.is_cuda and dy.dtype == torch.float32) else dy.sum(0)
        return dx, dW, db


def dense_linear(x, weight, bias):
    return DenseLinearFn.apply(x, weight, bias)


def _rows(x, idx):
    """x[idx]: the library's row gather for device f32 rows (copy speed), torch indexing otherwise"""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and idx.dim() == 1 and idx.dtype == torch.int64:
        return get_backend().gather_rows(x, idx)
    return x[idx]


class GatherRowsFn(torch.autograd.Function):
    """y = x[idx] with a many-to-one index (voxel -> points).  torch's backward (index_put_ with accumulate) sorts the
    indices on every call; here the backward is one scatter-add kernel."""

    PRESORT_MIN_ROWS = 1 << 16

    @staticmethod
    def forward(ctx, x, idx, max_dup=None):
        ctx.save_for_backward(idx)
        ctx.n_rows, ctx.max_dup, ctx.sorted = x.size(0), max_dup, None
        be = get_backend()
        if idx.is_cuda and (max_dup is None or max_dup > 2) and hasattr(be, "presort_rows") and be.deterministic():
            # the backward sums in a fixed order over a stable sort of idx: take the one the loader's prefetch left on the
            # tensor, or queue it now on the helper thread / side stream
            cached = getattr(idx, "_ms3d_sorted", None)
            queued = getattr(idx, "_ms3d_presort", None)
            if cached is not None and cached[2] == idx._version:
                ctx.sorted = cached               # the whole entry: it carries the event of the stream that sorted
            elif queued is not None and queued[1] == idx._version:
                ctx.sorted = queued[0]            # a second gather over the same index (HAIS: features and mask scores)
            elif idx.numel() >= GatherRowsFn.PRESORT_MIN_ROWS:
                ctx.sorted = be.presort_rows(idx)
                idx._ms3d_presort = (ctx.sorted, idx._version)
        return _rows(x, idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        extra = {}
        if ctx.max_dup is not None:
            extra["max_dup"] = ctx.max_dup
        if ctx.sorted is not None:
            extra["sorted_"] = ctx.sorted
        return get_backend().scatter_add_rows(dy.contiguous(), idx, ctx.n_rows, **extra), None, None


class PermuteRowsFn(torch.autograd.Function):
    """y = x[inv] for a PERMUTATION inv (perm = its inverse): the backward is the gather dy[perm], no accumulation"""

    @staticmethod
    def forward(ctx, x, inv, perm):
        ctx.save_for_backward(perm)
        return _rows(x, inv)

    @staticmethod
    def backward(ctx, dy):
        (perm,) = ctx.saved_tensors
        return _rows(dy, perm), None, None


def gather_rows(x, idx, max_dup=None):
    """differentiable x[idx] for 2-D float features and an int64 row index.  max_dup: the caller's bound on how many
    NON-ZERO addends one element of the gradient can collect -- how many entries of idx name the same row, or fewer when
    the caller knows the incoming gradient is sparse (<= 2: the backward's float atomics are order-independent -- a + b =
    b + a and adding zeros is exact -- no sort needed)"""
    if x.dim() == 2 and idx.dim() == 1 and idx.dtype == torch.int64 and x.requires_grad:
        return GatherRowsFn.apply(x, idx, max_dup)
    return _rows(x, idx) if x.dim() == 2 and idx.dim() == 1 else x[idx]
