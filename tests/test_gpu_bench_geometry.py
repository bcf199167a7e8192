"""bf16 mode AT THE BENCH GEOMETRY (BASELINE configs[1]: B = 520 utterances x T = 40 frames, three 1200-2000-1000-500-50
encoder streams, H = 250, concat, summed BLSTM): the weight-stationary LSTM launches (17 groups x 4 workgroups x 3 LSTMs =
204 resident workgroups), the ping-pong GEMM kernel with its grouped and split-K forms and the concat fusion all run here
and nowhere else at this size -- bench.py itself only asserts a finite loss.

  (a) weight-stationary kernels against the one-workgroup-per-slice kernels (ADN_LSTM_NO_CLUSTER): same arithmetic
      (bf16 operands, fp32 accumulate) in a different summation order -> probabilities equal to 1e-3; gradients compared
      as whole tensors (relative L2 distance and cosine): every layer below an LSTM re-rounds its gradient to bf16, so a
      last-bit difference upstream flips roundings element-wise downstream (measured and printed by the test);
  (b) ping-pong GEMMs against the register-staged ones (ADN_GEMM_PP=0): the same;
  (c) against the fp64 ORACLE on a 26-utterance slice (the graph is independent per utterance, so rows 0..25 of the
      520-utterance result must be what the oracle computes for those 26 alone): probabilities and the measured
      majority-vote agreement, which is reported and asserted;
  (d) round 6 -- WHAT bench.py TIMES: the frame-compacted step (lengths announced, the library's own 8192-row threshold, the
      selection rules as they fall at 13 730 rows) in bf16, bf16x3 and mixed against the padded run of the same parameters,
      against the 26-utterance fp64-oracle slice, and on bench.py's own zero-bias parameters, where every padding row sits
      exactly on the rectifier kink."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import adenet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUN = r'''
import time as _t; _t0 = _t.time()
def _tick(w):
    if os.environ.get("GEOM_TICKS"): print("[tick] %%6.2f s %%s" %% (_t.time() - _t0, w), file=sys.stderr, flush=True)
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import bench
from ip_avsr_amd.model import AdeNetModel
torch.cuda.set_device(0)
_tick("imports")
m = AdeNetModel(bench.build_spec())
prec = os.environ.get("GEOM_PRECISION", "bf16")
m.set_precision(prec)
bench.synthetic_params(m)
xs, y, m_d, mask = bench.synthetic_batch(torch, 0, int(os.environ.get("GEOM_BATCH", bench.B_PER_GPU)), torch.device("cuda", 0))
x26 = [x[:26].cpu().numpy() for x in xs]
lens = mask.sum(axis=1).astype(np.int32)
compact = bool(os.environ.get("GEOM_COMPACT"))       # what bench.py does: the lengths announced ahead of every call
announce = (lambda: m.set_batch_lengths(lens)) if compact else (lambda: None)
if os.environ.get("GEOM_INPUTS") == "bench":         # ... and the resident form bench.py hands over in this arithmetic
    if prec == "bf16":
        xs = [x.to(torch.bfloat16) for x in xs]
    elif prec in ("bf16x3", "mixed"):
        from ip_avsr_amd.model import PlaneInput
        xs = [PlaneInput.split(x) for x in xs]
if len(sys.argv) > 2:                                # every variant evaluates the SAME parameters
    saved = np.load(sys.argv[2])
    for p in m.params:
        p.set_value(saved["p_" + p.name])
elif not os.environ.get("GEOM_FRESH"):              # (GEOM_FRESH: bench.py's own start -- zero biases)
    for _ in range(3):                               # a few steps so that the outputs are not flat
        announce()
        m.train_step(xs, y, m_d, bench.THETA, 2e-3, want_loss=False)
_tick("params set / warm steps done")
params = {"p_" + p.name: p.get_value() for p in m.params}
announce()
probs = m.predict(xs, m_d, bench.THETA)
_tick("predict")
rows = m.compact_rows()
announce()
loss = m.compute_grads(xs, y, m_d, bench.THETA)
rows = min(rows, m.compact_rows())
brows = m.bucket_rows()                              # (the train step's time-major rows when it ran over length buckets, else 0)
_tick("compute_grads")
g = m.get_grads_dict()
keep = ["fc1_s1.W", "fc2_s2.W", "bottleneck_s3.W", "lstm_s1.W_hid_to_cell", "lstm_s3.W_in_to_ingate", "lstm_s2.b_outgate",
        "f_lstm_agg.W_in_to_forgetgate", "b_lstm_agg.W_hid_to_outgate", "f_lstm_agg.hid_init", "softmax.W", "softmax.b"]
keep += [n for n in g if n.endswith(".b") and n.split("_")[0] in ("fc1", "fc2", "fc3", "bottleneck")]     # where the padding rows' sum lands
np.savez(sys.argv[1], probs=probs, loss=loss, mask=mask, rows=rows, brows=brows, **{"g_" + k: g[k] for k in keep}, **params,
         **{"x%%d" %% k: x26[k] for k in range(3)})
''' % ROOT


def _run(tmp_path, tag, params_from=None, **env):
    out = os.path.join(str(tmp_path), tag + ".npz")
    e = dict(os.environ, **env)
    subprocess.run([sys.executable, "-c", RUN, out] + ([params_from] if params_from else []), check=True, env=e, cwd=ROOT,
                   timeout=600)
    return dict(np.load(out))


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    d = tmp_path_factory.mktemp("geom")
    default = _run(d, "default")
    ref = os.path.join(str(d), "default.npz")
    return dict(default=default, nocluster=_run(d, "nocluster", ref, ADN_LSTM_NO_CLUSTER="1"),
                nopp=_run(d, "nopp", ref, ADN_GEMM_PP="0"), streams=_run(d, "streams", ref, ADN_STREAMS="1"),
                streammajor=_run(d, "streammajor", ref, ADN_NO_GROUPED_BACKWARD="1"),
                norsgroups=_run(d, "norsgroups", ref, ADN_GEMM_NO_RS_GROUPS="1"),
                b26=_run(d, "b26", ref, GEOM_BATCH="26"),
                b26_norsgroups=_run(d, "b26_norsgroups", ref, GEOM_BATCH="26", ADN_GEMM_NO_RS_GROUPS="1"),
                ref_path=ref, dir=d)


@pytest.fixture(scope="module")
def timed_path_runs(runs):
    """The step bench.py times, per arithmetic: lengths announced (-> frame compaction at the library's own threshold), inputs in the
    resident form bench.py hands over (bfloat16 / hi-lo planes), ADN_CHECK_PADDING=1 (mask and padding frames verified on the device);
    each beside the padded run of the same arithmetic and the same parameters (those of runs['default'])."""
    d, ref = runs["dir"], runs["ref_path"]
    out = {}
    for prec in ("bf16", "bf16x3", "mixed"):
        out[prec + "_compact"] = _run(d, prec + "_compact", ref, GEOM_PRECISION=prec, GEOM_COMPACT="1", GEOM_INPUTS="bench", ADN_CHECK_PADDING="1")
        out[prec + "_padded"] = runs["default"] if prec == "bf16" else _run(d, prec + "_padded", ref, GEOM_PRECISION=prec)
        # ... and beside the same compacted step with its recurrent side over B x T rows (no length buckets)
        out[prec + "_compact_bxt"] = _run(d, prec + "_compact_bxt", ref, GEOM_PRECISION=prec, GEOM_COMPACT="1", GEOM_INPUTS="bench",
                                          ADN_NO_LENGTH_BUCKETS="1")
    # the bf16 step with the rectifier's mask read from the bf16 activations instead of the forward epilogue's bit images
    out["bf16_compact_nobits"] = _run(d, "bf16_compact_nobits", ref, GEOM_PRECISION="bf16", GEOM_COMPACT="1", GEOM_INPUTS="bench", ADN_NO_RELU_BITS="1")
    # bench.py's own start (no training step first): zero biases, every padding row exactly on the rectifier kink
    out["fresh_compact"] = _run(d, "fresh_compact", None, GEOM_FRESH="1", GEOM_COMPACT="1", GEOM_INPUTS="bench")
    out["fresh_padded"] = _run(d, "fresh_padded", None, GEOM_FRESH="1")
    return out


def _close(a, b, what, p_tol=1e-3, g_tol=3e-2, cos_tol=0.9995, bit_equal_forward=False):
    valid = a["mask"][..., None].astype(bool)
    if bit_equal_forward:                            # the same products in the same order
        np.testing.assert_array_equal(a["probs"] * valid, b["probs"] * valid)
    assert np.abs(a["probs"] - b["probs"]).max() <= p_tol, what
    assert abs(a["loss"] - b["loss"]) <= 1e-3 * abs(b["loss"]), what
    worst = {}
    for k in a:
        if k.startswith("g_"):
            x, y = a[k].astype(np.float64).ravel(), b[k].astype(np.float64).ravel()
            rel = np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-30)
            cos = x @ y / max(np.linalg.norm(x) * np.linalg.norm(y), 1e-30)
            worst[k[2:]] = (round(rel, 5), round(cos, 6))
    print(what, "| max |dp| %.2e | (relative L2, cosine) per gradient tensor:" % np.abs(a["probs"] - b["probs"]).max(), worst)
    for k, (rel, cos) in worst.items():
        assert rel <= g_tol and cos >= cos_tol, (what, k, rel, cos)


def test_weight_stationary_lstm_equals_slice_kernels_at_b520(runs):
    """Forward: bit-identical on the valid frames (the same bf16 products in the same order).  Backward: the exchange carries
    the partial dh sums with 19 significant bits and every encoder layer re-rounds its gradient to bf16; measured on an
    MI355X at this geometry: relative L2 distance 0.24 % at fc1 (the deepest tensor), 0.03 % at the LSTM inputs, cosine
    >= 0.999997 everywhere (printed by the test)."""
    _close(runs["default"], runs["nocluster"], "cluster vs one-workgroup LSTM kernels", g_tol=1e-2, cos_tol=0.9999,
           bit_equal_forward=True)


def test_pingpong_gemm_equals_register_staged_at_b520(runs):
    """Forward GEMMs: the same k order, fp32 accumulate -> bit-identical.  Weight gradients: ordered partial slabs instead of
    float atomics in arrival order -> fp32 summation noise only (measured: < 5e-6 relative L2)."""
    _close(runs["default"], runs["nopp"], "ping-pong vs register-staged GEMM kernels", g_tol=1e-4, cos_tol=0.999999,
           bit_equal_forward=True)


def test_schedules_of_the_backward_pass_agree_at_b520(runs):
    """Layer-major back-propagation with grouped launches (the single-GPU default) against the stream-major order (what data
    parallel runs use) and against the streams on forked HIP streams (ADN_STREAMS=1: no shared split-K workspace there -- a
    shared one was a race this test caught).  The same products; the weight gradients are summed in a different order."""
    _close(runs["default"], runs["streammajor"], "layer-major vs stream-major backward", g_tol=1e-4, cos_tol=0.999999,
           bit_equal_forward=True)
    _close(runs["default"], runs["streams"], "default vs forked HIP streams", g_tol=1e-4, cos_tol=0.999999, bit_equal_forward=True)


def test_grouped_launches_of_the_register_staged_kernels_change_nothing(runs):
    """Same-shape GEMMs the ping-pong kernel declines go out as one launch of the register-staged kernels (blockIdx.z = problem;
    csrc/gemm_f32.hip::gemm_rs): a few at B = 520, nearly every GEMM of the step at the reference's minibatch B = 26.  The
    same products in the same k order; a group may pick another tile shape or split than one problem alone."""
    _close(runs["default"], runs["norsgroups"], "grouped vs single register-staged launches, B = 520", g_tol=1e-4, cos_tol=0.999999,
           bit_equal_forward=True)
    # (B = 26: forward bit-identical.  The gradients carry the run-to-run noise of this batch size -- float atomics in arrival
    #  order (split-K partial sums, the LSTM kernels' bias / initial-state sums), re-rounded to bf16 on the way down: two runs
    #  of the SAME configuration differ by 2e-4 ... 1.7e-3 relative L2, grouped against single launches by the same amounts
    #  (profiles/scripts/b26_noise.py))
    _close(runs["b26"], runs["b26_norsgroups"], "grouped vs single register-staged launches, B = 26", g_tol=1e-2, cos_tol=0.9999,
           bit_equal_forward=True)


def test_bf16_against_the_fp64_oracle_on_a_26_utterance_slice(runs):
    r = runs["default"]
    spec = O.spec_nstream([1200, 1200, 1200])
    p64 = {k[2:].replace("_s1", "_s1"): v.astype(np.float64) for k, v in r.items() if k.startswith("p_")}
    # bench.build_spec names its layers fc1_s1 ... like spec_nstream does
    assert set(p64) == set(O.param_names(spec))
    mask = r["mask"][:26]
    xs = [r["x%d" % k].astype(np.float64) for k in range(3)]
    ref = O.forward(spec, p64, xs, mask, 9)
    got = r["probs"][:26]
    err = np.abs(got - ref).max()
    votes_ref, votes = O.majority_vote(ref, mask), O.majority_vote(got, mask)
    agree = float((votes_ref == votes).mean())
    frame_agree = float(((got.argmax(-1) == ref.argmax(-1)) | (mask == 0)).mean())
    print("bf16 vs fp64 oracle at the bench geometry: max |dp| = %.2e, majority-vote agreement %.3f, per-frame top-1 %.4f"
          % (err, agree, frame_agree))
    assert err <= 1e-3                                   # measured 3.2e-4
    assert agree == 1.0                                  # measured: all 26 majority votes identical
    assert frame_agree >= 0.995                          # measured: every valid frame's top-1 identical


def test_bf16_gradients_against_the_fp64_oracle_at_the_reference_minibatch(runs):
    """The b26 run (bench model, the reference's 26 utterances x 40 frames, bf16 mode: grouped register-staged GEMMs,
    weight-stationary LSTM launches) against the ORACLE's gradients for the same parameters and batch -- not against another
    HIP path: every kept gradient tensor by cosine and relative L2 distance.  Measured: relative L2 0.4 % at the recurrent
    and classifier tensors, 0.9 % at the bottleneck, 1.9 % at fc2, 4.1 % at fc1 (every encoder layer re-rounds the gradient it
    passes down to bf16; fc1 sits below four of them and the delta layer), cosine 0.99916 (fc1) ... 0.99999.  The bounds:
    5 % relative L2 everywhere, i.e. cosine >= 1 - 0.05^2 / 2."""
    r = runs["b26"]
    spec = O.spec_nstream([1200, 1200, 1200])
    p64 = {k[2:]: v.astype(np.float64) for k, v in r.items() if k.startswith("p_")}
    mask = r["mask"]
    assert mask.shape == (26, 40)
    xs = [r["x%d" % k].astype(np.float64) for k in range(3)]
    y = np.repeat((np.arange(26) % 26)[:, None], 40, axis=1).astype(np.int32)      # bench.synthetic_batch: labels i mod 26
    loss, g_ref, _ = O.loss_and_grads(spec, p64, xs, y, mask, 9)
    assert abs(float(r["loss"]) - loss) <= 1e-3 * abs(loss)
    worst = {}
    for k in r:
        # (the encoder bias gradients are kept for the compacted-vs-padded tests -- sums over 1 040 rows of bf16-rounded terms: 6 % at
        #  this batch size, outside this test's 5 % band for weight tensors)
        if k.startswith("g_") and not (k.endswith(".b") and k[2:].split("_")[0] in ("fc1", "fc2", "fc3", "bottleneck")):
            a, b = r[k].astype(np.float64).ravel(), g_ref[k[2:]].ravel()
            worst[k[2:]] = (round(float(np.linalg.norm(a - b) / np.linalg.norm(b)), 5), round(float(a @ b / np.sqrt((a @ a) * (b @ b))), 6))
    print("bf16 vs fp64 oracle gradients at B = 26 (relative L2, cosine):", worst)
    for k, (rel, cos) in worst.items():
        assert rel <= 0.05 and cos >= 0.9987, (k, rel, cos)       # (cosine >= 1 - 0.05^2 / 2; fc1 and the learnt initial states sit
                                                                  #  lowest: 0.99916 / 0.99949, every weight matrix above fc1 >= 0.9998)


def _valid_rows(r):
    return int(r["mask"].sum()) + 1


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "mixed"])
def test_the_compacted_step_bench_times_equals_the_padded_one(timed_path_runs, prec):
    """VERDICT r5 next #1a: B = 520 x T = 40, real widths, default ADN_COMPACT_MIN_ROWS, lengths announced: the encoders ran over
    sum(len) + 1 rows (asserted), and probabilities, loss and the kept gradient tensors (+ every encoder bias gradient: where
    the padding rows' summed gradient lands) equal the padded run's at the arithmetic's grade."""
    c, p = timed_path_runs[prec + "_compact"], timed_path_runs[prec + "_padded"]
    declined = bool(os.environ.get("ADN_NO_COMPACT") or os.environ.get("ADN_STREAMS") or os.environ.get("ADN_BF16_NO_SHADOW") or
                    (prec != "bf16" and os.environ.get("ADN_X3_NO_PLANES")))
    assert int(c["rows"]) == (0 if declined else _valid_rows(c)) and int(p["rows"]) == 0
    if prec == "bf16":       # another row count = other tiles / summation orders, and bf16-resident inputs: one bf16 rounding step
        _close(c, p, "bf16: compacted (lengths announced, bfloat16-resident) vs padded", p_tol=1e-3, g_tol=2e-2, cos_tol=0.9998)
    elif prec == "bf16x3":
        # (measured 0 .. 2e-5; 2.6e-4 at fc1 under ADN_GEMM_PP=0, where the two row counts take different split-image routes)
        # (... where the weight gradients are also summed by float atomics in arrival order: one run in two envmatrix passes read 5e-4+)
        off_pp = os.environ.get("ADN_GEMM_PP") in ("0", "7")          # (the products over planes leave the persistent kernels: split images + atomics)
        _close(c, p, "bf16x3: compacted (planes resident) vs padded", p_tol=1e-5, g_tol=1.5e-3 if off_pp else 5e-4, cos_tol=0.99999 if off_pp else 0.999999)
    else:                    # forward fp32-grade, back-propagation one bf16 product per GEMM
        _close(c, p, "mixed: compacted (planes resident) vs padded", p_tol=1e-5, g_tol=2e-2, cos_tol=0.9998)


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "mixed"])
def test_the_bucketed_step_bench_times_equals_the_step_over_b_x_t_rows(timed_path_runs, prec):
    """Round 6, length buckets (include/adenet.h adn_set_length_buckets) at B = 520 x T = 40: the timed train step keeps its
    time-major tensors in 4 buckets of 130 utterances (asserted: fewer than 0.9 B T rows), and its loss and gradients are those of
    the same compacted step over B x T rows up to the order of the sums -- every product is the same product in either layout."""
    c, u = timed_path_runs[prec + "_compact"], timed_path_runs[prec + "_compact_bxt"]
    declined = bool(os.environ.get("ADN_NO_COMPACT") or os.environ.get("ADN_STREAMS") or os.environ.get("ADN_BF16_NO_SHADOW") or
                    os.environ.get("ADN_NO_LENGTH_BUCKETS") or os.environ.get("ADN_DETERMINISTIC") or os.environ.get("ADN_LSTM_NO_CLUSTER") or
                    os.environ.get("ADN_LSTM_CUS") or (prec == "bf16" and os.environ.get("ADN_LSTM_NO_CLUSTER_BWD")) or
                    (prec != "bf16" and (os.environ.get("ADN_X3_NO_PLANES") or os.environ.get("ADN_LSTM_NO_X3_CLUSTER") or
                                         os.environ.get("ADN_LSTM_NO_X3_CLUSTER_BWD"))))
    assert int(u["brows"]) == 0
    if declined:
        assert int(c["brows"]) == 0
        return
    n = int(c["mask"].size)
    assert 0 < int(c["brows"]) <= 0.9 * n, int(c["brows"])
    print("%s: %d time-major rows instead of %d" % (prec, int(c["brows"]), n))
    np.testing.assert_array_equal(c["probs"], u["probs"])                 # (the forward-only pass never buckets: the same launches)
    # bf16 / mixed re-round every gradient below an LSTM to bf16: a last-bit difference of a sum upstream flips roundings downstream
    g_tol = {"bf16": 5e-3, "bf16x3": 1e-5, "mixed": 5e-3}[prec]
    _close(c, u, "%s: length buckets vs B x T rows" % prec, p_tol=0.0, g_tol=g_tol, cos_tol=0.99999)
    assert abs(float(c["loss"]) - float(u["loss"])) <= 2e-6 * abs(float(u["loss"]))


def test_rectifier_bit_images_give_the_masks_of_the_bf16_activations(timed_path_runs):
    """Round 6: in the bf16 arithmetic the input-gradient epilogues of the two wide encoder layers read the rectifier's mask as the
    bit image the forward epilogue of the same tile grid left (one 16-byte load per thread and tile instead of 32 eight-byte
    loads of the bf16 activation; gemm_common.h GemmGroup::Cbits).  `y > 0` of the fp32 value and of its bf16 rounding are the
    same bit (short of fp32 denormals), every product and every sum is the same: the step must equal the ADN_NO_RELU_BITS=1
    step bit for bit in its forward pass and in every kept gradient tensor."""
    a, b = timed_path_runs["bf16_compact"], timed_path_runs["bf16_compact_nobits"]
    np.testing.assert_array_equal(a["probs"], b["probs"])
    assert float(a["loss"]) == float(b["loss"])
    if os.environ.get("ADN_GEMM_PP") == "0":
        return                       # (no ping-pong launches, no bit images; and float atomics in arrival order: equal to rounding only)
    worst = 0.0
    for k in a:
        if k.startswith("g_"):
            x, y = a[k].astype(np.float64), b[k].astype(np.float64)
            worst = max(worst, float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30)))
    print("bit images vs bf16 masks: worst kept-gradient difference %.2e of its scale" % worst)
    assert worst <= 1e-5, worst       # (the split-K weight gradients of the LSTMs are summed by atomics: last bits may differ)


def test_the_mixed_forward_pass_is_the_bf16x3_one_on_the_compacted_path(timed_path_runs):
    a, b = timed_path_runs["mixed_compact"], timed_path_runs["bf16x3_compact"]
    np.testing.assert_array_equal(a["probs"], b["probs"])
    assert float(a["loss"]) == float(b["loss"])


@pytest.mark.parametrize("prec,p_tol", [("bf16", 1e-3), ("bf16x3", 1e-4), ("mixed", 1e-4)])
def test_the_compacted_step_against_the_fp64_oracle_slice(timed_path_runs, prec, p_tol):
    """rows 0..25 of the compacted 520-utterance result against the oracle's forward pass of those 26 utterances alone: the parity
    gate of north_star (1e-4 on probabilities, identical votes) for bf16x3 / mixed, bf16's own 1e-3 -- ON THE PATH THE BENCH TIMES."""
    r = timed_path_runs[prec + "_compact"]
    spec = O.spec_nstream([1200, 1200, 1200])
    p64 = {k[2:]: v.astype(np.float64) for k, v in r.items() if k.startswith("p_")}
    mask = r["mask"][:26]
    xs = [r["x%d" % k].astype(np.float64) for k in range(3)]
    ref = O.forward(spec, p64, xs, mask, 9)
    got = r["probs"][:26]
    err = np.abs(got - ref).max()
    agree = float((O.majority_vote(ref, mask) == O.majority_vote(got, mask)).mean())
    print("%s compacted vs fp64 oracle at the bench geometry: max |dp| = %.2e, majority-vote agreement %.3f" % (prec, err, agree))
    assert err <= p_tol and agree == 1.0


def test_zero_bias_start_compacted_equals_padded_at_the_bench_geometry(timed_path_runs):
    """bench.py's own parameters (zero encoder biases: every padding row at pre-activation exactly 0 in all three rectifier layers):
    probabilities / loss / gradients of the compacted and the padded computation of the SAME fresh parameters."""
    c, p = timed_path_runs["fresh_compact"], timed_path_runs["fresh_padded"]
    assert all(np.all(v == 0) for k, v in p.items() if k.startswith("p_") and k.endswith(".b") and not k.startswith("p_softmax"))
    assert int(c["rows"]) in (0, _valid_rows(c)) and int(p["rows"]) == 0
    # (the learnt initial states' gradients are sums of float atomics in arrival order over bf16-rounded terms: 2 % run to run)
    _close(c, p, "zero-bias start: compacted vs padded", p_tol=1e-3, g_tol=5e-2, cos_tol=0.998)
