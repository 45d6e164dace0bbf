"""Fused Adam over the engine's flat parameter buffer (replaces torch.optim.Adam, model_manager.py:27).

One kernel launch updates all 196 live tensors (31 012 944 elements): the parameters, gradients and both
moments are contiguous fp32 buffers in the same order.  `state_dict()` / `load_state_dict()` speak
torch.optim.Adam's format (per-parameter 'step', 'exp_avg', 'exp_avg_sq'; the 72 dead decoder BN parameters
have no state, exactly like torch skips params whose grad is None), so `optimiser.pth` stays interchangeable
with the reference's checkpoints.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        params = list(model.parameters())          # same param list / order the reference hands to Adam
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._step = 0
        self._m = self._v = None
        self.grad_scale = 1.0                        # data-parallel: 1/world_size folded into the update

    def _buffers(self, eng):
        if self._m is None or self._m.numel() != eng.flat_param.numel() or self._m.device != eng.flat_param.device:
            self._m = torch.zeros_like(eng.flat_param)
            self._v = torch.zeros_like(eng.flat_param)
        return self._m, self._v

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model.engine()
        if not eng.params_alias_flat():
            raise RuntimeError("FusedAdam: parameters were re-allocated; run a forward pass first")
        for p, v in zip(eng.live_params, eng.grad_views):
            if p.grad is None:
                raise RuntimeError("FusedAdam.step: live parameter without gradient (call backward first)")
            if p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)                      # foreign gradient tensor: bring it into the flat buffer
        m, v = self._buffers(eng)
        g = self.param_groups[0]
        self._step += 1
        ops.adam_step(eng.flat_param, eng.flat_grad, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._step,
                      self.grad_scale)
        eng.weights_dirty = True
        return None

    def fused_step(self, eng):
        """Fast path used by TrainStep (no per-parameter checks)."""
        m, v = self._buffers(eng)
        g = self.param_groups[0]
        self._step += 1
        ops.adam_step(eng.flat_param, eng.flat_grad, m, v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._step,
                      self.grad_scale)
        eng.weights_dirty = True
        self._opt_called = True                      # what lr_scheduler's wrapper of step() records (no "scheduler before optimizer" warning)

    # ---- the same update in pieces: a range of the flat buffer as soon as its gradients are complete (TrainStep, FP_ADAM_STAGED) ----
    def begin_fused_step(self, eng):
        """what fused_step does before its launch: moments allocated, step counter advanced once for all ranges of this step"""
        self._buffers(eng)
        self._step += 1
        self._opt_called = True

    def fused_range(self, eng, lo, hi, hyper_dev=None):
        """Adam on elements [lo, hi) of the flat buffers (multiples of 4: parameter slots are 16-byte aligned) on the current launch
        stream.  Element-wise, so any partition of [0, total) gives the bits of the one-launch form.  hyper_dev: the step's scalars in
        device memory (recorded plans / graph replay); None: host scalars of the step begin_fused_step opened."""
        m, v = self._buffers(eng)
        if hyper_dev is not None:
            ops.adam_step_dev(eng.flat_param[lo:hi], eng.flat_grad[lo:hi], m[lo:hi], v[lo:hi], hyper_dev)
        else:
            g = self.param_groups[0]
            ops.adam_step(eng.flat_param[lo:hi], eng.flat_grad[lo:hi], m[lo:hi], v[lo:hi], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                          self._step, self.grad_scale)
        eng.weights_dirty = True

    # ---- graph replay: launch with device-resident scalars, refreshed by the host before each replay ----------
    def graph_step(self, eng, hyper_dev):
        """captured once by TrainStep: the update with its scalars read from `hyper_dev`"""
        m, v = self._buffers(eng)
        ops.adam_step_dev(eng.flat_param, eng.flat_grad, m, v, hyper_dev)
        eng.weights_dirty = True

    def next_hyper(self):
        """advance the step counter and return the host scalars of that step (same values fused_step would use)"""
        g = self.param_groups[0]
        self._step += 1
        return ops.adam_hyper(g["lr"], g["betas"][0], g["betas"][1], g["eps"], self._step, self.grad_scale)

    # ---- torch.optim.Adam-compatible checkpoint format ---------------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        if self._m is not None and self._step > 0:
            eng = self.model.engine()
            index = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
            for p, o in zip(eng.live_params, eng.offsets):
                n = p.numel()
                state[index[id(p)]] = {"step": torch.tensor(float(self._step)),
                                       "exp_avg": self._m[o:o + n].view(p.shape).clone(),
                                       "exp_avg_sq": self._v[o:o + n].view(p.shape).clone()}
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k] if k != "betas" else tuple(groups[0][k])
        state = sd.get("state", {})
        if not state:
            return
        eng = self.model.engine()
        m, v = self._buffers(eng)
        index = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
        steps = set()
        for p, o in zip(eng.live_params, eng.offsets):
            st = state.get(index[id(p)])
            if st is None:
                continue
            n = p.numel()
            m[o:o + n].view(p.shape).copy_(st["exp_avg"])
            v[o:o + n].view(p.shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError("FusedAdam: per-parameter step counts differ (%s)" % sorted(steps))
        self._step = steps.pop() if steps else 0
