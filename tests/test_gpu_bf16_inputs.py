"""ADN_FLAG_BF16_INPUTS (include/adenet.h): stream inputs handed over as bfloat16 arrays.  In bf16 mode a device array of an
encoder stream is the first GEMM's operand as it is (no staging copy, no conversion pass); everywhere else the array is
widened to float32, which is exact.  Either way the results must equal, BIT FOR BIT, those of float32 inputs that hold
the same (bf16-representable) values."""
import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu


def _case(dims, has_encoder, rng, B=9, T=7):
    spec = O.spec_nstream(dims, enc_shapes=(32, 16, 8), enc_acts=("rectify", "rectify", "linear"), lstm_size=12, classes=5,
                          fusion="concat", has_encoder=has_encoder)
    p = O.init_params(spec, rng, np.float32, enc_std=0.3, perturb=0.1)
    lens = rng.integers(2, T + 1, size=B); lens[0] = T
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.uint8)
    y = np.repeat(rng.integers(0, 5, size=(B, 1)), T, axis=1).astype(np.int32)
    return spec, p, mask, y


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "f32"])
@pytest.mark.parametrize("dims,has_encoder", [([48, 40], [True, True]), ([48, 30], [True, False]), ([44, 40], [True, True])])
def test_bf16_inputs_equal_float32_inputs_of_the_same_values(precision, dims, has_encoder):
    """(48 / 40: multiples of 8 -> read in place in bf16 mode; 44: not -> widened; a stream without an encoder feeds the
    delta layer, which reads float32 -> widened.)  Device arrays, then host arrays (torch CPU bfloat16 tensors)."""
    import torch
    from ip_avsr_amd.model import AdeNetModel
    torch.cuda.set_device(0)
    rng = np.random.default_rng(3)
    spec, p, mask, y = _case(dims, has_encoder, rng)
    B, T = mask.shape
    x16 = [torch.tensor((rng.normal(size=(B, T, d)) * mask[..., None]).astype(np.float32)).to(torch.bfloat16) for d in dims]
    x32 = [x.to(torch.float32) for x in x16]                      # the same values as float32
    m = AdeNetModel(dict(spec, precision=precision))
    m.set_params_dict(p)
    ref_probs = m.predict([x.cuda() for x in x32], mask, 2)
    ref_loss = m.compute_grads([x.cuda() for x in x32], y, mask, 2)
    ref_g = m.get_grads_dict()
    for where in ("device", "host"):
        xs = [x.cuda() for x in x16] if where == "device" else x16
        np.testing.assert_array_equal(m.predict(xs, mask, 2), ref_probs)
        loss = m.compute_grads(xs, y, mask, 2)
        assert loss == ref_loss, (where, loss, ref_loss)
        g = m.get_grads_dict()
        for k in ref_g:
            if precision == "f32" or not k.endswith(".W"):
                # (weight gradients of the bf16 modes are split-K sums: arrival-order noise between two runs of ANY input form)
                tol = 0 if precision == "f32" else 1e-6 * max(1e-30, np.abs(ref_g[k]).max())
                assert np.abs(g[k] - ref_g[k]).max() <= tol, (where, k)
            else:
                assert np.abs(g[k] - ref_g[k]).max() <= 1e-5 * max(1e-30, np.abs(ref_g[k]).max()), (where, k)
    m.close()


def test_bf16_inputs_through_train_steps_at_real_widths():
    """The bench's configuration: three 1200-pixel encoder streams, bf16 mode, bfloat16 device batch -- four Adam steps give
    the costs of the float32-input run (same values) to the last bit of the forward pass."""
    import torch
    import bench
    from ip_avsr_amd.model import AdeNetModel
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    xs, y, m_d, _ = bench.synthetic_batch(torch, 0, 52, dev)
    x16 = [x.to(torch.bfloat16) for x in xs]
    x32 = [x.to(torch.float32) for x in x16]
    costs = {}
    for name, batch in (("bf16", x16), ("f32", x32)):
        m = AdeNetModel(bench.build_spec())
        m.set_precision("bf16")
        bench.synthetic_params(m)
        costs[name] = [float(m.train_step(batch, y, m_d, bench.THETA, 1e-3)) for _ in range(4)]
        m.close()
    assert costs["bf16"][0] == costs["f32"][0]                     # same operands, same products, same order
    np.testing.assert_allclose(costs["bf16"], costs["f32"], rtol=2e-4)      # (later steps: split-K arrival order in the gradients)
