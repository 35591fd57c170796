"""GAN losses of the path (host-side glue over D outputs): src/loss/gan.py:5-22, 39-49 and
src/loss/position.py:4-18.  The R1 term differentiates through oi_amd's discriminator, whose
autograd Functions supply the double-backward from HIP kernels."""
import torch.nn.functional as F
from torch import autograd


def compute_grad2(d_out, x_in):
    batch_size = x_in.size(0)
    (grad_dout,) = autograd.grad(outputs=d_out.sum(), inputs=x_in, create_graph=True, retain_graph=True,
                                 only_inputs=True)
    return grad_dout.pow(2).reshape(batch_size, -1).sum(1).mean()


class GANLoss:
    def __init__(self, gan_str):
        if gan_str != "bce":
            raise NotImplementedError(gan_str)

    def __call__(self, d_out, target):
        assert d_out.dim() == 2 and d_out.shape[1] == 1, d_out.shape
        return F.binary_cross_entropy_with_logits(d_out, d_out.new_full(d_out.size(), float(target)))


class PositionLoss:
    def __init__(self, loss_str):
        self.loss = {"mse": F.mse_loss, "smooth_l1": F.smooth_l1_loss}[loss_str]

    def __call__(self, pred, target, reduction="mean"):
        return self.loss(pred, target, reduction=reduction)


def linear_increase(max_it, max_weight):
    return lambda it: min(it / max_it, 1) * max_weight
