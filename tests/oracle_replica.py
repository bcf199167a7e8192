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

    # --- the bucket-by-bucket surface (DataParallel(overlap=True) on a host tensor): two parameter tensors per bucket,
    # listed back to front like the library's completion order, the 8-float tail in the first bucket
    def grad_buckets(self):
        edges = np.concatenate([[0], np.cumsum(self.sizes)]).astype(int)
        cuts = list(edges[::2])
        if cuts[-1] != edges[-1]:
            cuts.append(int(edges[-1]))
        ranges = [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]
        ranges[-1] = (ranges[-1][0], ranges[-1][1] + 8)
        return ranges[::-1]

    def grad_bucket_groups(self):
        n = len(self.grad_buckets())
        return [0] + [1 + (k - 1) // 2 for k in range(1, n)]          # pairs of buckets share a release point

    def adam_begin(self, lr):
        self._ranged = dict(lr=lr, t=self.state["t"] + 1, done=[])

    def adam_range(self, begin, end):
        """Adam on the parameter tensors inside [begin, end) -- the oracle's own adam_step on that subset, with the
        step count held at the value adam_begin fixed."""
        flat = self.grad.numpy()
        off, sub_p, sub_g, sub_state = 0, {}, {}, dict(t=self._ranged["t"] - 1, m={}, v={})
        for n, sz in zip(self.names, self.sizes):
            if begin <= off and off + sz <= end:
                sub_p[n] = self.p[n]; sub_g[n] = flat[off:off + sz].reshape(self.p[n].shape)
                sub_state["m"][n] = self.state["m"][n]; sub_state["v"][n] = self.state["v"][n]
            else:
                assert off + sz <= begin or off >= min(end, sum(self.sizes)), "a bucket must not cut a tensor"
            off += sz
        self.O.adam_step(sub_p, sub_g, sub_state, self._ranged["lr"])
        for n in sub_p:
            self.p[n] = sub_p[n]; self.state["m"][n] = sub_state["m"][n]; self.state["v"][n] = sub_state["v"][n]
        self._ranged["done"] += list(sub_p)

    def adam_end(self):
        assert sorted(self._ranged["done"]) == sorted(self.names), "every parameter exactly once"
        self.state["t"] = self._ranged["t"]
        del self._ranged

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

    def snapshot_state(self):
        return dict(p={k: v.copy() for k, v in self.p.items()}, m={k: v.copy() for k, v in self.state["m"].items()},
                    v={k: v.copy() for k, v in self.state["v"].items()}, t=int(self.state["t"]))

    def restore_state(self, snap):
        self.p = {k: v.copy() for k, v in snap["p"].items()}
        self.state["m"] = {k: v.copy() for k, v in snap["m"].items()}
        self.state["v"] = {k: v.copy() for k, v in snap["v"].items()}
        self.state["t"] = snap["t"]

    def count_params(self):
        return int(sum(self.sizes))

    def set_precision(self, precision):
        pass

    def synchronize(self):
        pass
