"""One GAN training iteration over the oi_amd modules, with the call pattern, loss composition and
weights of the reference trainer (src/trainers/gan_pose_trainer.py:77-202; configs/train.yaml:120-147):

    G step:   render (grad) -> D(image)[:, :1], maskD(mask) -> BCE(.,1) + 0.1*BCE(.,1) + 10*eikonal -> backward, Adam
    D step:   render (no grad) -> BCE(D(real),1) + BCE(D(fake)[:, :1],0) + 10*R1(real) + w(it)*MSE(D(fake)[:,1:7], pose)
    maskD:    render (no grad) -> same without the pose term
i.e. 3 renders, 3+3 discriminator forwards, 3 backward passes (each followed by the flat-gradient
all-reduce when wrapped in oi_amd.ddp.FlatGradDDP) per iteration.  Pinned by F13 (tests/test_gpu_trainer_f13.py): two
iterations of the reference's OWN `Trainer.train_step` (imported in the build container with torchvision / imageio
stubbed, oracle/gen_golden_r3.py; ADA on, pinned percentile) -- every render, every returned loss, the generator's
gradient norms and the three networks' weights after each iteration -- eager and with captured discriminator steps;
F9 pins the per-parameter gradients of one iteration assembled from the reference's pieces.  Running the reference's
class itself on top of the oi_amd modules stays untested (its module imports tu.loggers / visualisation helpers)."""
import torch

from .losses import GANLoss, PositionLoss, gan_losses, grad_wrt_input, linear_increase, ones_scalar, weighted_sum

import os

MODULE_KEYS = ("generator", "discriminator", "mask_discriminator")
# Where the two graphed discriminator steps run (each a replayed chain of ~75 launches of a few microseconds that wait for each
# other: 0.42 ms during which the chip is almost idle):
#   OI_TRAIN_D_STEPS=concurrent (default): both no-grad renders first (they read generator state only), then the discriminator's
#     step on a second stream NEXT TO the mask discriminator's on the main one -- two latency-bound chains side by side.  Same-box
#     A/B at C2 (tools/dbg/run_overlap_ab.sh): 9.20 -> 9.00 ms per iteration.
#   OI_TRAIN_D_STEPS=overlap: the discriminator's step on a second stream under the SECOND RENDER.  Built and measured in round 6
#     (tools/dbg/run_overlap_ab.sh, profiles/r6_overlap_timeline.txt): no gain, 9.19-9.27 ms per iteration either way -- the
#     render's MLP launches own every compute unit (one workgroup per CU: the whole register file, 150 KB of LDS), a freed CU goes
#     to their next workgroup first, and each of the step's dependent launches then takes 25-90 us instead of 5-10: the step
#     stretches from 0.42 to 1.5 ms, exactly the render it hides behind (a high-priority stream: the same).
#   OI_TRAIN_D_STEPS=serial: one stream, the reference's order (render, step, render, step).
# Same arithmetic and the same order of host-side draws in all three (the steps draw nothing from torch's generator).
D_STEPS_MODE = os.environ.get("OI_TRAIN_D_STEPS", "concurrent")
CONCURRENT_MAX_PIXELS = int(os.environ.get("OI_TRAIN_CONCURRENT_MAX_PIXELS", 2 * 128 * 128))
assert D_STEPS_MODE in ("concurrent", "overlap", "serial"), D_STEPS_MODE
DATA_KEYS = {"generator": ["image"], "discriminator": ["image"], "mask_discriminator": ["mask"]}


def toggle_grad(model, requires_grad):
    for p in model.parameters():
        p.requires_grad_(requires_grad)


class _GradToggle:
    """`toggle_grad` over the three networks, three times per iteration (gan_pose_trainer.py:103-106, 148-151), walks
    ~100 parameters through nn.Module.parameters() each time: remember each network's state and parameter list and
    touch a network only when its state changes."""

    def __init__(self):
        self.state, self.params = {}, {}

    def __call__(self, key, model, requires_grad):
        cached = self.params.get(key)
        if cached is None or cached[0] is not model:
            cached = self.params[key] = (model, list(model.parameters()))
            self.state.pop(key, None)
        ps = cached[1]
        # (first / last parameter probed: somebody else may have toggled the network since)
        if self.state.get(key) is requires_grad and (not ps or (ps[0].requires_grad is requires_grad
                                                               and ps[-1].requires_grad is requires_grad)):
            return
        for p in cached[1]:
            p.requires_grad_(requires_grad)
        self.state[key] = requires_grad


def _cat(tensors):
    """torch.cat along the channel axis; a single tensor is returned as is (both discriminators take ONE map:
    MODULE_KEYS_TO_DATA_KEYS of gan_pose_trainer.py:27-31 -- the reference's cat of one tensor is a copy launch)."""
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim=-3)


def _backward_to(loss, net):
    """loss.backward() restricted to the network's parameters: the reference marks the discriminator INPUTS as requiring
    grad (gan_pose_trainer.py:160, 168) and so also back-propagates both losses through the augmentation into the images,
    a gradient nothing reads; the R1 term's own second-order path is unaffected."""
    torch.autograd.backward(loss, inputs=[p for p in net.parameters() if p.requires_grad])


def _unwrap(m):
    return m.module if hasattr(m, "module") and isinstance(m.module, torch.nn.Module) else m


def _sync(net):
    """Gradient exchange of a FlatGradDDP-wrapped network (issued by its end-of-backward callback, or here if this rank's
    backward produced no gradient for it) must have completed before the optimiser reads the gradients."""
    if hasattr(net, "sync"):
        net.sync()


def _zero_grad(net, opt):
    """FlatGradDDP keeps every gradient as a view of one flat buffer: one fill.  Plain modules drop their gradients
    (`set_to_none=True`, the reference's default `opt.zero_grad()`): no fill per parameter now, no `+=` per parameter in
    the next backward (~250 launches per iteration for the three networks)."""
    if hasattr(net, "flat_grad"):
        net.zero_grad()
    else:
        opt.zero_grad(set_to_none=True)


class Trainer:
    def __init__(self, modules, loss_weight=None, it=-1, graph_d_steps=False):
        """graph_d_steps: replay the two discriminator steps from captured hipGraphs (oi_amd.graphed.GraphedDStep); the
        ADA augmentation then pads with its largest margins (same pixels, shape-static)."""
        self.modules = modules
        for k in MODULE_KEYS:
            setattr(self, k, modules[k])
            setattr(self, f"opt_{k}", modules[f"opt_{k}"])
        lw = dict(disc_in_gen=1.0, mask_disc_in_gen=0.1, eikonal=10.0, reg=10.0, aux_pose=linear_increase(1000, 1))
        lw.update(loss_weight or {})
        self.loss_weight = lw
        self.gan, self.aux_pose = GANLoss("bce"), PositionLoss("mse")
        self.it = it
        self._toggle = _GradToggle()
        self._graphed = {} if graph_d_steps else None

    def train_step(self, data):
        self.it += 1
        for k in MODULE_KEYS:
            self.modules[k].train()
        bs = data["image"].shape[0]
        out = {}
        out.update(self.train_step_generator(bs))
        with torch.no_grad():
            blob = self.generator(bs=bs, it=self.it, data={}, return_raw=False)["box"]
        # The second no-grad render does not depend on the discriminator's update: it is ENQUEUED between the
        # discriminator's backward and its `sync(); opt.step()` so that, under FlatGradDDP, the 11 MB gradient exchange on
        # the communication stream overlaps it.  Same arithmetic and same RNG draw order as the reference's sequence
        # (gan_pose_trainer.py:84-90): the render only reads generator state.
        # (side by side while the steps are chains of launches too small to fill the chip -- up to 128 x 128 x 2 pixels per batch,
        #  OI_TRAIN_CONCURRENT_MAX_PIXELS.  At the shipped 128 x 128 crop the same A/B first measured a LOSS, 7.32 -> 7.51 ms per
        #  iteration, with the two-launch augmentation and its six-launch adjoint in every pass; with the one-launch forms it is a
        #  gain: 6.55-6.75 -> 6.38-6.49 ms, three alternating pairs on one box)
        small = data["image"].shape[0] * data["image"].shape[-2] * data["image"].shape[-1] <= CONCURRENT_MAX_PIXELS
        if self._graphed is not None and D_STEPS_MODE == "concurrent" and small and data["image"].is_cuda:
            # inputs of the discriminator's step staged (and its augmentation drawn: the host's draw order stays render, step,
            # render, step) on the second stream; the second render on the main one; then the two replays side by side
            start_d = self.train_step_discriminator("discriminator", data, {**blob["render_out"], "c2b": blob["prior_info"]["c2b"]},
                                                    defer_step=True, side="staged")
            with torch.no_grad():
                blob2 = self.generator(bs=bs, it=self.it, data={}, return_raw=False)["box"]
            ret_d, join = start_d()
            ret_m = self.train_step_discriminator("mask_discriminator", data, blob2["render_out"])
            join()
            out.update(ret_d)
            out.update(ret_m)
            return out
        ret_d, finish_d = self.train_step_discriminator("discriminator", data, {**blob["render_out"], "c2b": blob["prior_info"]["c2b"]},
                                                        defer_step=True, side=D_STEPS_MODE == "overlap")
        with torch.no_grad():
            blob = self.generator(bs=bs, it=self.it, data={}, return_raw=False)["box"]
        finish_d()
        out.update(ret_d)
        out.update(self.train_step_discriminator("mask_discriminator", data, blob["render_out"]))
        return out

    def train_step_generator(self, bs):
        for k in MODULE_KEYS:
            self._toggle(k, self.modules[k], k == "generator")
        _zero_grad(self.generator, self.opt_generator)
        with self._zero_pool():
            ret = self._generator_loss_backward(bs)
        _sync(self.generator)
        self.opt_generator.step()
        return ret

    def _zero_pool(self):
        """ONE fill for the split-K / scatter outputs of the step's discriminator passes (ops.ZeroPool: ~20 fill launches of
        the generator step otherwise; sized by the first step, valid until the next one begins)."""
        from . import ops
        params = list(_unwrap(self.generator).parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("oi_amd.trainer.Trainer: the modules must be on the GPU (there is no host-tensor path)")
        if getattr(self, "_gpool", None) is None:
            self._gpool = ops.ZeroPool()
        # Gradients taken from pool slices become `.grad` of leaf parameters (AccumulateGrad steals them): begin() zeroes that
        # memory again, which is only right because _zero_grad(set_to_none=True) ran first -- checked, not assumed.
        self._gpool.begin(dev, must_not_alias=params)
        return self._gpool

    def _generator_loss_backward(self, bs):
        blob = self.generator(bs=bs, it=self.it, data={}, return_raw=False)["box"]
        # BCE(D(fake)[:, :1], 1) per discriminator: one launch each way (losses.gan_losses = GANLoss("bce") fused)
        x_fake = _cat([blob["render_out"][k] for k in DATA_KEYS["discriminator"]])
        loss_disc, _ = gan_losses(d_real=self.discriminator(x_fake, it=self.it))
        m_fake = _cat([blob["render_out"][k] for k in DATA_KEYS["mask_discriminator"]])
        loss_mask, _ = gan_losses(d_real=self.mask_discriminator(m_fake, it=self.it))
        # the weighted sum of the terms (gan_pose_trainer.py:122-137) in one launch each way (losses.weighted_sum)
        terms, weights = [loss_disc, loss_mask], [self.loss_weight["disc_in_gen"], self.loss_weight["mask_disc_in_gen"]]
        ret = {"generator/loss": loss_disc, "generator/loss_mask": loss_mask}
        for k, v in blob["loss"].items():
            terms.append(v)
            weights.append(self.loss_weight[k])
            ret[f"generator/{k}"] = v
        loss = weighted_sum(terms, weights)
        loss.backward(ones_scalar(loss))  # (a cached 1.0: autograd's default is a fill launch per call)
        return ret

    def train_step_discriminator(self, key, real, fake, defer_step=False, side=False):
        """side (graphed steps only): enqueue the step, its gradient exchange and its optimiser step on the trainer's second
        stream; the returned callable joins the main stream to it."""
        for k in MODULE_KEYS:
            self._toggle(k, self.modules[k], k == key)
        disc, opt = self.modules[key], self.modules[f"opt_{key}"]
        if self._graphed is not None:
            return self._graphed_d_step(key, disc, opt, real, fake, defer_step, side)
        _zero_grad(disc, opt)
        x_real = _cat([real[k] for k in DATA_KEYS[key]]).detach().clone().requires_grad_()
        d_real = disc(x_real, it=self.it)
        gx = grad_wrt_input(d_real[:, :1], x_real)                  # the R1 penalty's inner gradient (compute_grad2)
        x_fake = _cat([fake[k] for k in DATA_KEYS[key]]).detach()   # (the reference also marks it requires_grad: unused)
        d_fake = disc(x_fake, it=self.it)
        pose, aux_w = None, None
        if d_fake.size(1) > 1:
            pose = _unwrap(self.generator).pose_prior.pose_to_vec_repr(fake["c2b"])
            aux_w = torch.full((), float(self.loss_weight["aux_pose"](self.it)), device=d_fake.device)
        # BCE(real, 1) + BCE(fake, 0) + reg R1 + aux_w MSE(pose) in one launch each way (losses.gan_losses)
        loss, parts = gan_losses(d_real, d_fake, pose, gx, aux_w, self.loss_weight["reg"])
        _backward_to(loss, disc)
        ret = {f"{key}/loss": parts[0], f"{key}/reg": parts[1], f"{key}/fake": parts[2], f"{key}/real": parts[3],
               f"{key}/aux_pose": parts[4] if pose is not None else 0}

        def finish():
            _sync(disc)
            opt.step()

        if defer_step:
            return ret, finish
        finish()
        return ret

    def _graphed_d_step(self, key, disc, opt, real, fake, defer_step, side=False):
        from .graphed import GraphedDStep
        gd = self._graphed.get(key)
        if gd is None:
            gd = self._graphed[key] = GraphedDStep(disc, self.gan, self.aux_pose, self.loss_weight["reg"],
                                                   _unwrap(self.generator).pose_prior)
        x_real = _cat([real[k] for k in DATA_KEYS[key]]).detach()
        x_fake = _cat([fake[k] for k in DATA_KEYS[key]]).detach()
        has_aux = _unwrap(disc).out_dim > 1
        c2b = fake["c2b"].detach() if has_aux else None
        aux_w = self.loss_weight["aux_pose"](self.it) if has_aux else 0.0
        named = lambda out: {f"{key}/loss": out["loss"], f"{key}/reg": out["reg"], f"{key}/fake": out["fake"],
                             f"{key}/real": out["real"], f"{key}/aux_pose": out["aux_pose"] if has_aux else 0}
        if defer_step and side and x_real.is_cuda:
            # the step, its gradient exchange and its optimiser step on the SECOND stream; the images it reads are kept alive for
            # that stream; the caller joins through the returned callable
            main = torch.cuda.current_stream()
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=x_real.device)
            side_mode, side = side, self._side_stream
            side.wait_stream(main)
            for t in (x_real, x_fake, c2b):
                if t is not None:
                    t.record_stream(side)

            def replay():
                with torch.cuda.stream(side):
                    out = gd.run()
                    _sync(disc)
                    opt.step()
                out["loss"].record_stream(main)   # (the five scalars are views of one clone made on the side stream)
                return named(out), lambda: main.wait_stream(side)

            with torch.cuda.stream(side):
                gd.prepare(x_real, x_fake, c2b, aux_w)
            if side_mode == "staged":   # the caller enqueues more on the main stream first; the replay waits for that too

                def start():
                    side.wait_stream(main)
                    return replay()

                return start
            return replay()
        out = gd(x_real, x_fake, c2b, aux_w)
        ret = named(out)

        def finish():
            _sync(disc)
            opt.step()

        if defer_step:
            return ret, finish
        finish()
        return ret
