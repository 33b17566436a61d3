"""Storage-rounding emulation on top of the fp32 oracle (test helper).

The HIP path stores weights and activations in bf16/f16 (fp32 accumulate).  This subclass rounds at the same
points (packed weights, raw conv output, BN+ReLU output, residual output) with a straight-through estimator,
so a comparison against it isolates LOGIC errors from low-precision noise."""
import torch
import torch.nn.functional as F

from oracle import facenet_oracle as fo


def _q(x, dt):
    return x + (x.to(dt).to(torch.float32) - x).detach()


class _ReluWithMask(torch.autograd.Function):
    """ReLU whose active set is GIVEN (the device path's): forward x*mask, backward grad*mask.  Two implementations of the
    same forward differ by rounding noise eps, so a fraction ~eps of the pre-activations lands on the other side of zero;
    every such element changes the gradient by its full magnitude and the relative gradient distance is ~sqrt(eps) -- a
    property of ReLU, not an error of either side.  Sharing the masks removes it and leaves the backward ARITHMETIC."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class QuantOracle(fo.Oracle):
    def __init__(self, params, dt, masks=None, **kw):
        super().__init__(params, **kw)
        self.dt = dt
        self.masks = masks or {}          # layer / block prefix -> float {0,1} tensor [N,C,H,W]: ReLU active sets to use

    def _relu(self, x, prefix):
        m = self.masks.get(prefix)
        return F.relu(x) if m is None else _ReluWithMask.apply(x, m)

    def _conv(self, x, prefix, spec, bias=False):
        w = _q(self.p[prefix + "/kernel"], self.dt).permute(3, 2, 0, 1)
        kh, kw = spec["k"]
        pad = (kh // 2, kw // 2) if spec["padding"] == "same" else (0, 0)
        b = self.p[prefix + "/bias"] if bias else None
        return F.conv2d(x, w, b, stride=spec["stride"], padding=pad)

    def _cbr(self, x, prefix, spec, training):
        y = self._conv(x, prefix, spec)
        # batch statistics come from the un-rounded accumulators, normalisation reads the rounded tensor
        if training:
            dims = (0, 2, 3)
            mean, var = y.mean(dim=dims), y.var(dim=dims, unbiased=False)
            mm, mv = self.p[prefix + "/bn/moving_mean"], self.p[prefix + "/bn/moving_variance"]
            self.new_stats[prefix + "/bn/moving_mean"] = mm * fo.BN_MOMENTUM + mean.detach() * (1 - fo.BN_MOMENTUM)
            self.new_stats[prefix + "/bn/moving_variance"] = mv * fo.BN_MOMENTUM + var.detach() * (1 - fo.BN_MOMENTUM)
        else:
            mean, var = self.p[prefix + "/bn/moving_mean"], self.p[prefix + "/bn/moving_variance"]
        yq = _q(y, self.dt)
        z = (yq - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + fo.BN_EPS) + self.p[prefix + "/bn/beta"].view(1, -1, 1, 1)
        return _q(self._relu(z, prefix), self.dt)

    def _block(self, net, prefix, blk, scale, activation, training):
        mixed = torch.cat([self._tower(net, f"{prefix}/tower_conv{i}", t, training) for i, t in enumerate(blk["towers"])], dim=1)
        up = self._conv(mixed, prefix + "/up", fo._cbr("Conv2d_1x1", blk["up"], 1), bias=True)
        net = net + scale * up
        if activation:
            net = self._relu(net, prefix)
        return _q(net, self.dt)

    def forward(self, images, training=False, preprocessed=False):
        x = fo.image_processing(images, self.normalization, self.image_size)
        return super().forward(_q(x, self.dt), training=training, preprocessed=True)


def quant_train_step_grads(params, trainable, images, loss_kind, dt, labels=None, alpha=0.2):
    for k in trainable:
        params[k].requires_grad_(True)
        params[k].grad = None
    o = QuantOracle(params, dt)
    # avgpool output is stored in low precision too
    emb = o.forward(images, training=True)
    if loss_kind == "softmax":
        w = _q(params["classifier/logits/kernel"], dt)
        data = fo.softmax_cross_entropy(_q(emb, dt) @ w + params["classifier/logits/bias"], torch.as_tensor(labels))
    else:
        data = fo.triplet_loss(fo.l2_normalize(emb), alpha)
    data.backward()
    grads = {k: params[k].grad.detach().clone() for k in trainable}
    for k in trainable:
        params[k].requires_grad_(False)
        params[k].grad = None
    return float(data.detach()), grads, emb.detach()
