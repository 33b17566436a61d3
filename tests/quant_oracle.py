"""Storage-rounding emulation on top of the fp32 oracle (test helper).

The HIP path stores weights and activations in bf16/f16 (fp32 accumulate).  This subclass rounds at the same
points (packed weights, raw conv output, BN+ReLU output, residual output) with a straight-through estimator,
so a comparison against it isolates LOGIC errors from low-precision noise."""
import torch
import torch.nn.functional as F

from oracle import facenet_oracle as fo


def _q(x, dt):
    return x + (x.to(dt).to(torch.float32) - x).detach()


class QuantOracle(fo.Oracle):
    def __init__(self, params, dt, **kw):
        super().__init__(params, **kw)
        self.dt = dt

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
        return _q(F.relu(z), self.dt)

    def _block(self, net, prefix, blk, scale, activation, training):
        return _q(super()._block(net, prefix, blk, scale, activation, training), self.dt)

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
