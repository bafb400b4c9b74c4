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


class SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, gamma, beta, spec, bn):
        be = get_backend()
        W3 = W.view(spec.K, spec.cin, spec.cout)
        wf = be.prep_weights(W3, spec.K, spec.cin, spec.cout)
        pre = (bn["scale"], bn["shift"]) if bn is not None else None
        y = be.conv_forward(x, wf, spec.nbr_fwd, spec.vout, spec.K, spec.cin, spec.cout, pre=pre,
                            pre_relu=bool(bn and bn["relu"]))
        ctx.spec, ctx.bn = spec, bn
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        be = get_backend()
        spec, bn = ctx.spec, ctx.bn
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        W3 = W.view(spec.K, spec.cin, spec.cout)
        dx = dgamma = dbeta = None
        if ctx.needs_input_grad[0]:
            wft = be.prep_weights(W3, spec.K, spec.cout, spec.cin, transpose=True, mirror=spec.mirror)
            if bn is None:
                dx = be.conv_forward(dy, wft, spec.nbr_bwd, spec.vin, spec.K, spec.cout, spec.cin)
            elif bn["relu"]:
                dz, s1s2 = be.conv_forward(dy, wft, spec.nbr_bwd, spec.vin, spec.K, spec.cout, spec.cin,
                                           bn_bwd=(x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"]))
                dbeta, dgamma = s1s2[0], s1s2[1]
                dx = (be.bn_bwd_apply(dz, x, bn["scale"], bn["mean"], bn["invstd"], s1s2) if bn["training"]
                      else dz * bn["scale"])
            else:
                da = be.conv_forward(dy, wft, spec.nbr_bwd, spec.vin, spec.K, spec.cout, spec.cin)
                dz, s1s2 = be.bn_bwd_reduce(da, x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"], False)
                dbeta, dgamma = s1s2[0], s1s2[1]
                dx = (be.bn_bwd_apply(dz, x, bn["scale"], bn["mean"], bn["invstd"], s1s2) if bn["training"]
                      else dz * bn["scale"])
        pre = (bn["scale"], bn["shift"]) if bn is not None else None
        dW = be.conv_backward_weight(x, dy, spec.nbr_fwd, spec.vout, spec.K, spec.cin, spec.cout, pre=pre,
                                     pre_relu=bool(bn and bn["relu"])).view_as(W)
        if bn is not None and dgamma is None:
            # x itself needs no gradient but gamma / beta still do
            wft = be.prep_weights(W3, spec.K, spec.cout, spec.cin, transpose=True, mirror=spec.mirror)
            da = be.conv_forward(dy, wft, spec.nbr_bwd, spec.vin, spec.K, spec.cout, spec.cin)
            _, s1s2 = be.bn_bwd_reduce(da, x, bn["scale"], bn["shift"], bn["mean"], bn["invstd"], bn["relu"])
            dbeta, dgamma = s1s2[0], s1s2[1]
        return dx, dW, dgamma, dbeta, None, None


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


def conv(x, W, spec, pending):
    if pending is not None and pending.get("gamma") is None:
        x, pending = torch.relu(x), None
    if pending is None:
        return SparseConvFn.apply(x, W, None, None, spec, None)
    return SparseConvFn.apply(x, W, pending["gamma"], pending["beta"], spec, pending)
