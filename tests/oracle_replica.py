"""A CPU replica with AdeNetModel's call surface as far as ``ip_avsr_amd.parallel.DataParallel`` and ``bench.py`` use it,
computed by the oracle in float64 (the real replica needs an MI355X).  Test infrastructure: lets the N > 1 code paths run
under ``gloo`` on CPU."""
import numpy as np
import torch


class OracleReplica(object):
    """Same call surface as AdeNetModel for what DataParallel uses, computed by the oracle in float64;
    the flat gradient 'buffer' is a CPU tensor with the same 8-float tail (cost share in tail[0])."""

    def __init__(self, spec, params):
        from oracle import adenet_oracle as O
        self.O, self.spec = O, spec
        self.p = {k: v.copy() for k, v in params.items()}
        self.names = O.param_names(spec)
        self.sizes = [self.p[n].size for n in self.names]
        self.grad = torch.zeros(sum(self.sizes) + 8, dtype=torch.float64)
        self.state = O.adam_init(self.p)

    def compute_grads(self, inputs, targets, mask, window, total_frames=0.0, want_loss=True):
        loss, g, _ = self.O.loss_and_grads(self.spec, self.p, inputs, targets, mask, window,
                                           total_frames=total_frames if total_frames > 0 else None)
        flat = np.concatenate([np.asarray(g[n], np.float64).reshape(-1) for n in self.names] + [np.zeros(8)])
        flat[-8] = loss
        self.grad.copy_(torch.from_numpy(flat))
        return loss if want_loss else None

    def apply_adam(self, lr):
        flat = self.grad.numpy()
        g, off = {}, 0
        for n, sz in zip(self.names, self.sizes):
            g[n] = flat[off:off + sz].reshape(self.p[n].shape)
            off += sz
        self.O.adam_step(self.p, g, self.state, lr)

    # --- the rest of the surface DataParallel / bench.py touch
    def zero_grads(self):
        self.grad.zero_()

    def adam_step_count(self):
        return int(self.state["t"])

    def set_adam_step_count(self, t):
        self.state["t"] = int(t)

    def predict(self, inputs, mask, window):
        return self.O.forward(self.spec, self.p, [np.asarray(x, np.float64) for x in inputs], np.asarray(mask), window)

    def loss(self, inputs, targets, mask, window, deterministic=True):
        l, _, _ = self.O.loss_and_grads(self.spec, self.p, [np.asarray(x, np.float64) for x in inputs], np.asarray(targets),
                                        np.asarray(mask), window)
        return l

    def train_step(self, inputs, targets, mask, window, lr, want_loss=True):
        l = self.compute_grads(inputs, targets, mask, window)
        self.apply_adam(lr)
        return l if want_loss else None

    def count_params(self):
        return int(sum(self.sizes))

    def set_precision(self, precision):
        pass

    def synchronize(self):
        pass
