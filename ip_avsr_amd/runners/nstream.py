"""Dataset-agnostic N-stream trainer: the epoch driver of reference runners/{1,2,3,4}stream.py, reading the
same ``.ini`` files (SURVEY.md App. B schema 1) and ``.mat`` inputs (App. C), driving the MI355X model.

    python ip_avsr_amd/runners/3stream.py --config config/trimodal.ini [--write_results F] [--learning_rate f]
                                          [--save_best F] [--save_plot PREFIX]

Data parallel (new; the reference is single-device): launch with
``python -m torch.distributed.run --nproc-per-node N ip_avsr_amd/runners/3stream.py --config ...``; every rank
draws the same minibatch order (``--seed``), trains on ``batch_idxs[rank::N]`` and all-reduces gradients
(ip_avsr_amd/parallel.py).  Rank 0 evaluates and reports.

Behaviour kept from the reference: per-epoch train cost is the cost of the LAST minibatch re-evaluated after
its update (App. E-7); GL / Pk / PQ statistics; best-parameter snapshot on validation-cost improvement;
``early_stop2``; results line ``test_cr,best_cr,best_val``; pickled parameter list for ``--save_best``.
"""
from __future__ import print_function

import argparse
import configparser
import os
import sys
import time

import numpy as np
import scipy.io as sio

from .. import init as las_init
from ..custom.nonlinearities import select_nonlinearity
from ..modelzoo import (adenet_2stream, adenet_3stream, adenet_3stream_dropout, adenet_4stream, adenet_v2_2,
                        adenet_v2, adenet_v2_nodelta, deltanet_majority_vote, deltanet_v1,
                        lstm_classifier_majority_vote)
from ..utils.data_structures import circular_list
from ..utils.datagen import compute_integral_len, gen_lstm_batch_random, gen_seq_batch_from_idx
from ..utils.io import load_mat_file, read_data_split_file, save_model_params
from ..utils.plotting_utils import plot_confusion_matrix, plot_validation_cost, print_network
from ..utils.preprocessing import (compute_diff_images, concat_first_second_deltas, featurewise_normalize_sequence,
                                   multistream_force_align,
                                   normalize_input, reorder_data, sequencewise_mean_image_subtraction, split_seq_data)
from ..utils.regularization import early_stop2


def load_decoder(path, shapes, nonlinearities):
    """``.mat`` with w1..wN / b1..bN -> (weights, biases, shapes, nonlinearities): the ``ae`` argument of every
    factory (reference runners/3stream.py:31-40)."""
    nn = sio.loadmat(path)
    shapes = [int(s) for s in shapes.split(',')]
    nonlins = [select_nonlinearity(n.strip()) for n in nonlinearities.split(',')]
    weights = [nn['w{}'.format(i + 1)].astype('float32') for i in range(len(shapes))]
    biases = [nn['b{}'.format(i + 1)][0].astype('float32') for i in range(len(shapes))]
    return weights, biases, shapes, nonlins


def evaluate_model2(X_vals, y_val, mask_val, window_size, eval_fn):
    """Majority vote over each utterance's valid frames, ties to the lowest class id; returns
    (classification rate, confusion matrix [target, prediction]) (reference runners/3stream.py:48-82).
    ``X_vals`` is the list of stream inputs."""
    output = eval_fn(*(list(X_vals) + [mask_val, window_size]))
    num_classes = output.shape[-1]
    seq_lens = np.sum(mask_val, axis=-1).astype(int)
    frame_pred = np.argmax(output, axis=-1)                               # (N, T)
    valid = np.arange(output.shape[1])[None, :] < seq_lens[:, None]
    votes = np.zeros((len(output), num_classes), dtype='int')
    for cls in range(num_classes):
        votes[:, cls] = np.sum((frame_pred == cls) & valid, axis=1)
    ix = np.argmax(votes, axis=1)
    classification_rate = np.sum(ix == np.asarray(y_val)) / float(len(ix))
    confusion_matrix = np.zeros((num_classes, num_classes), dtype='int')
    np.add.at(confusion_matrix, (np.asarray(y_val).astype(int), ix), 1)
    return classification_rate, confusion_matrix


def presplit_dataprocessing(data_matrix, vidlens, config, stream_name, **kwargs):
    """reorder -> mean removal -> diff images -> per-frame z-norm (reference runners/3stream.py:85-99)."""
    if config.getboolean(stream_name, 'reorderdata'):
        data_matrix = reorder_data(data_matrix, kwargs['imagesize'])
    if config.getboolean(stream_name, 'meanremove'):
        data_matrix = sequencewise_mean_image_subtraction(data_matrix, vidlens)
    if config.getboolean(stream_name, 'diffimage'):
        data_matrix = compute_diff_images(data_matrix, vidlens)
    if config.getboolean(stream_name, 'samplewisenormalize'):
        data_matrix = normalize_input(data_matrix)
    return data_matrix


def postsplit_datapreprocessing(train_X, val_X, test_X, config, stream_name):
    """feature-wise z-norm with the TRAIN split's statistics (reference runners/3stream.py:102-108)."""
    if config.getboolean(stream_name, 'featurewisenormalize'):
        train_X, mean, std = featurewise_normalize_sequence(train_X)
        val_X = (val_X - mean) / std
        test_X = (test_X - mean) / std
    return train_X, val_X, test_X


def parse_options(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', help='[CONFIG_FILE] config file to use')
    parser.add_argument('--write_results', help='[FILE] write results to file')
    parser.add_argument('--learning_rate', help='[LEARNING_RATE] learning rate')
    parser.add_argument('--save_best', help='[FILE] save the best model')
    parser.add_argument('--save_plot', help='[FILE_PREFIX] plot the train/validation loss curve')
    parser.add_argument('--layer_lr', default=None,
                        help='per-layer Adam learning rates "layer=rate,..." (adam_vlr, reference '
                             'runners/1stream_variable_lr.py:235-243); other layers use --learning_rate')
    parser.add_argument('--explode_layer_lr', default=None,
                        help='"EPOCH:RATE": after that epoch set every --layer_lr rate to RATE (the reference script '
                             'sets fc1..fc3 to 100.0 after epoch 4 to SHOW that the per-layer rates act: the loss '
                             'must diverge, runners/1stream_variable_lr.py:327-333)')
    parser.add_argument('--seed', type=int, default=None, help='seed for initialisers and minibatch order '
                                                               '(the reference never seeds; required >1 GPU)')
    parser.add_argument('--precision', default=None, choices=['f32', 'bf16x3', 'mixed', 'bf16'],
                        help='arithmetic of the model (default bf16x3 = fp32-grade products on the bf16 matrix pipe, meets the '
                             '1e-4 parity gate against the fp32 reference; f32 = exact fp32 MFMA products, a diagnostic mode ~5x slower; '
                             'mixed = bf16x3 forward pass, bf16 products in back-propagation; bf16 = fastest); also ADN_PRECISION')
    args = parser.parse_args(argv)
    options = {'config': args.config or 'config/bimodal_meanrm_raw_diff.ini', 'precision': args.precision}
    for key in ('write_results', 'save_best', 'save_plot'):
        if getattr(args, key):
            options[key] = getattr(args, key)
    if args.learning_rate:
        options['learning_rate'] = float(args.learning_rate)
    options['seed'] = args.seed
    if args.layer_lr:
        options['layer_lr'] = {kv.split('=')[0].strip(): float(kv.split('=')[1]) for kv in args.layer_lr.split(',')}
    if args.explode_layer_lr:
        e, r = args.explode_layer_lr.split(':')
        options['explode_layer_lr'] = (int(e), float(r))
    return options


def _dist_context():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    return dist, dist.get_rank(), world


def build_network(n_streams, aes, dims, lstm_weights, cfg):
    """Picks the same factory the reference runner of that stream count picks."""
    shapes = [(None, None, d) for d in dims]
    ms = (None, None)
    kw = dict(w_init_fn=cfg['weight_init_fn'], use_peepholes=cfg['use_peepholes'])
    H, C, fuse = cfg['lstm_size'], cfg['output_classes'], cfg['fusiontype']
    if n_streams == 1:           # runners/1stream.py:224-229
        net = deltanet_majority_vote.create_model(aes[0], shapes[0], None, ms, None, H, None, C,
                                                  cfg['weight_init_fn'], cfg['use_peepholes'], cfg['use_blstm'])
        return net, None
    if n_streams == 2:           # runners/2stream.py:256-272
        if lstm_weights[0] is not None and lstm_weights[1] is not None:
            return adenet_2stream.create_pretrained_model(aes[0], lstm_weights[0], aes[1], lstm_weights[1], shapes[0],
                                                          None, shapes[1], None, ms, None, H, None, C, fuse,
                                                          use_blstm_substream=cfg['use_blstm_substream'], **kw)
        return adenet_v2_2.create_model(aes[0], aes[1], shapes[0], None, ms, None, shapes[1], None, H, None, C, fuse,
                                        **kw)
    if n_streams == 3 and cfg.get('use_dropout'):      # runners/3stream.py:284-291
        return adenet_3stream_dropout.create_model(aes[0], aes[1], aes[2], shapes[0], None, shapes[1], None, shapes[2], None,
                                                   ms, None, H, None, C, fuse, **kw)
    if n_streams == 3:           # runners/3stream.py:293-299
        return adenet_3stream.create_model(aes[0], aes[1], aes[2], shapes[0], None, shapes[1], None, shapes[2], None,
                                           ms, None, H, None, C, fuse, **kw)
    if n_streams == 4:           # runners/4stream.py:321-328
        return adenet_4stream.create_model(aes[0], aes[1], aes[2], aes[3], shapes[0], None, shapes[1], None, shapes[2],
                                           None, shapes[3], None, ms, None, H, None, C, fuse, **kw)
    raise ValueError('1 to 4 streams are supported')


def build_network_avletters(n_streams, has_encoder, load_ae, dims, lstm_weights, cfg):
    """The factory reference avletters/{1,2,3}stream.py picks from the streams' ``has_encoder`` options (schema 1 of SURVEY
    App. B as those three scripts read it).  ``load_ae(k)`` loads stream k's encoder (only called where the script does):

      1 stream   has_encoder -> deltanet_majority_vote, else deltanet_v1 on the raw features      (avletters/1stream.py:228-238)
      2 streams  stream 2 has an encoder -> adenet_2stream.create_pretrained_model (both ``lstm_model`` given) / adenet_v2_2,
                 else adenet_v2: encoder stream + encoder-less second stream                       (avletters/2stream.py:265-286)
      3 streams  stream 2 has an encoder -> adenet_3stream (pretrained form when all three ``lstm_model`` are given);
                 the script has NO branch for an encoder-less stream 2 (``network`` is never bound: NameError at
                 avletters/3stream.py:298) and needs ae1 / ae3 in every branch                     (avletters/3stream.py:278-296)

    Where the reference script would stop with a NameError this raises a ValueError that says which option did it."""
    shapes = [(None, None, d) for d in dims]
    ms = (None, None)
    kw = dict(w_init_fn=cfg['weight_init_fn'], use_peepholes=cfg['use_peepholes'])
    H, C, fuse = cfg['lstm_size'], cfg['output_classes'], cfg['fusiontype']

    def need(k):
        if not has_encoder[k]:
            raise ValueError('stream{}: has_encoder = false, but the {}-stream script uses its encoder in every branch '
                             '(the reference stops with a NameError on ae{})'.format(k + 1, n_streams, k + 1))
        return load_ae(k)
    if n_streams == 1:
        if has_encoder[0]:
            return deltanet_majority_vote.create_model(load_ae(0), shapes[0], None, ms, None, H, None, C, cfg['weight_init_fn'],
                                                       cfg['use_peepholes'], cfg['use_blstm']), None
        return deltanet_v1.create_model(shapes[0], None, ms, None, None, H, C, cfg['weight_init_fn'], cfg['use_peepholes'],
                                        cfg['use_blstm']), None
    if n_streams == 2:
        ae1 = need(0)
        if not has_encoder[1]:
            return adenet_v2.create_model(ae1, shapes[0], None, ms, None, shapes[1], None, H, None, C, fuse,
                                          cfg['weight_init_fn'], cfg['use_peepholes'])
        ae2 = load_ae(1)
        if lstm_weights[0] is not None and lstm_weights[1] is not None:
            return adenet_2stream.create_pretrained_model(ae1, lstm_weights[0], ae2, lstm_weights[1], shapes[0], None, shapes[1],
                                                          None, ms, None, H, None, C, fuse,
                                                          use_blstm_substream=cfg['use_blstm_substream'], **kw)
        return adenet_v2_2.create_model(ae1, ae2, shapes[0], None, ms, None, shapes[1], None, H, None, C, fuse, **kw)
    if n_streams == 3:
        if not has_encoder[1]:
            raise ValueError('stream2: has_encoder = false -- avletters/3stream.py builds a network only when stream 2 has an '
                             'encoder (:278); use the 2-stream script for an encoder-less second stream')
        ae1, ae2, ae3 = need(0), load_ae(1), need(2)
        if all(w is not None for w in lstm_weights[:3]):
            return adenet_3stream.create_pretrained_model(ae1, lstm_weights[0], ae2, lstm_weights[1], ae3, lstm_weights[2],
                                                          shapes[0], None, shapes[1], None, shapes[2], None, ms, None, H, None, C,
                                                          fuse, cfg['weight_init_fn'], cfg['use_peepholes'],
                                                          cfg['use_blstm_substream'])
        return adenet_3stream.create_model(ae1, ae2, ae3, shapes[0], None, shapes[1], None, shapes[2], None, ms, None, H, None, C,
                                           fuse, **kw)
    raise ValueError('the avletters N-stream scripts exist for 1, 2 and 3 streams')


def resident_dtype(network):
    """Element type of the HBM-resident splits: bfloat16 when the model computes in bf16 and every stream enters through an
    encoder GEMM (which rounds its input to bfloat16 anyway: identical results, half the bytes, ADN_FLAG_BF16_INPUTS);
    'planes' (hi / lo bfloat16 pairs, ADN_FLAG_PLANE_INPUTS) in the bf16x3 / mixed arithmetic under the same condition;
    float32 otherwise (an encoder-less stream feeds the delta layer / LSTM projection in fp32)."""
    spec = network.spec
    if spec.get('precision') == 'bf16' and all(s.get('enc_shapes') and not s.get('aux_dim') for s in spec['streams']) \
            and not getattr(network, '_front', None) and not os.environ.get('ADN_FP32_RESIDENT'):
        return 'bfloat16'
    # bf16x3 / mixed: the two bfloat16 planes of every frame (the bytes of float32; the model's split pass over each batch goes)
    if spec.get('precision') in ('bf16x3', 'mixed') and all(s.get('enc_shapes') and not s.get('aux_dim') and s['input_dim'] % 8 == 0
                                                            for s in spec['streams']) \
            and not getattr(network, '_front', None) and not os.environ.get('ADN_FP32_RESIDENT'):
        return 'planes'
    return 'float32'


def fit(network, split, ys, lens, n_streams, windowsize, num_epoch, epochsize, batchsize, validation_window, learning_rate,
        options=None, say=print, rank=0, world=1, dp=None, lr_map=None, host_batches=None, prefetch=None, progress=True,
        two_way=False):
    """The epoch loop of reference runners/3stream.py:322-427 (identical in 1/2/4stream.py) on splits that stay resident in
    HBM: ``split[k]`` = list of (sum of lengths, D_s) frame matrices for k in train / val / test, ``ys[k]`` per-frame labels,
    ``lens[k]`` utterance lengths.  Minibatches are assembled on the GPU by index (utils/datagen_gpu.py; same utterance
    order, padding and label conventions as gen_lstm_batch_random / gen_seq_batch_from_idx), the held-out batches are built
    once and stay resident, and ``train`` does not wait for its cost (the reference discards it, runners/3stream.py:370), so
    the host runs ahead of the device inside an epoch.  ``host_batches`` (or ADN_HOST_BATCHES=1): the reference's host-side
    assembly instead, every batch uploaded -- the slow path, kept for A/B runs and for the bit-equality test between the two.
    ``two_way``: the loop of reference avletters/{1,2,3}stream.py (:322-404) -- train / val only: no test split, the confusion
    matrix kept is the validation one of the best epoch, no 'Test CR' in the epoch line.
    Returns the statistics dict of the run (incl. ``epoch_seconds``)."""
    options = options or {}
    if host_batches is None:
        host_batches = bool(os.environ.get('ADN_HOST_BATCHES'))
    if prefetch is None:
        prefetch = bool(os.environ.get('ADN_PREFETCH'))
    order = 'in1,targets,mask,in2,window' if n_streams == 2 else 'inputs,targets,mask,window'
    train, compute_train_cost, compute_test_cost, val_fn = network.compile(learning_rate, order)

    def call(fn, Xs, *rest):                        # the 2-stream runner interleaves its arguments
        if n_streams == 2:
            if len(rest) == 3:
                return fn(Xs[0], rest[0], rest[1], Xs[1], rest[2])
            return fn(Xs[0], rest[0], Xs[1], rest[1])
        return fn(*(list(Xs) + list(rest)))

    def eval_fn(*args):                             # evaluate_model2 passes inputs..., mask, window
        if dp is not None:                          # held-out utterances split over the ranks, predictions gathered
            return dp.predict_sharded(network.predict, list(args[:n_streams]), args[n_streams], args[n_streams + 1])
        return call(val_fn, list(args[:n_streams]), *args[n_streams:])

    def heldout_cost(Xs_, y_, m_):                  # compute_test_cost of a whole split
        if dp is not None:
            return dp.loss_sharded(network.loss, Xs_, y_, m_, windowsize,
                                   weights=None if network.head == "frames" else np.ones(len(m_)))
        return float(call(compute_test_cost, Xs_, y_, m_, windowsize))

    say('begin training...')
    cost_train, cost_val, class_rate, epoch_seconds, train_seconds = [], [], [], [], []
    STRIP_SIZE = 3
    val_window = circular_list(validation_window)
    train_strip = np.zeros((STRIP_SIZE,))
    best_val, best_cr, test_cr, test_conf, best_params = float('inf'), 0.0, 0.0, None, None

    tr_lens = lens['train']
    if host_batches:
        tmax_train = int(np.max(tr_lens))
        datagen = gen_lstm_batch_random(split['train'][0], ys['train'], tr_lens, batchsize=batchsize)
        integral_lens = compute_integral_len(tr_lens)

        def whole_split(k):
            gen = gen_lstm_batch_random(split[k][0], ys[k], lens[k], batchsize=len(lens[k]))
            X1, y, m, idxs = next(gen)
            il = compute_integral_len(lens[k])
            Xs = [X1] + [gen_seq_batch_from_idx(split[k][s], idxs, lens[k], il, np.max(lens[k]))
                         for s in range(1, n_streams)]
            return Xs, y, m, y.reshape((-1, 1)).repeat(m.shape[-1], axis=-1)
    else:
        from ..utils.datagen_gpu import DeviceSplit
        dtype = resident_dtype(network)
        resident = {k: DeviceSplit(split[k], ys[k], lens[k], dtype=dtype) for k in (('train', 'val') if two_way else ('train', 'val', 'test'))}
        # (one stream of batches per rank: a rank gathers only its own rows of every global minibatch)
        datagen = resident['train'].batches(batchsize, rank=rank if dp is not None else 0, world=world if dp is not None else 1,
                                            prefetch=prefetch)

        def whole_split(k):
            b = resident[k].whole()
            return b.Xs, b.y, b.mask, b.targets

    X_val, y_val_evaluate, mask_val, y_val = whole_split('val')
    X_test, y_test, mask_test, _ = (None, None, None, None) if two_way else whole_split('test')

    for epoch in range(num_epoch):
        time_start = time.time()
        for i in range(epochsize):
            if host_batches:
                X1, y, m, batch_idxs = next(datagen)
                y = y.reshape((-1, 1)).repeat(m.shape[-1], axis=-1)
                Xs = [X1] + [gen_seq_batch_from_idx(split['train'][s], batch_idxs, tr_lens, integral_lens, tmax_train)
                             for s in range(1, n_streams)]
                n_examples, total_frames = len(X1), float(m.sum())
                if dp is not None:
                    mine = list(range(len(X1)))[rank::world]
                    Xs_r, y_r, m_r = [x[mine] for x in Xs], y[mine], m[mine]
                else:
                    Xs_r, y_r, m_r = Xs, y, m
            else:
                batch = next(datagen)
                Xs_r, y_r, m_r = batch.Xs, batch.targets, batch.mask
                n_examples, total_frames = len(batch.global_idxs), batch.total_frames
            # frame compaction (include/adenet.h adn_set_batch_lengths): the loop knows the minibatch's lengths and both assemblies pad
            # with zero frames -- announced for the training call right below (ADN_RUNNER_PADDED=1: not announced)
            if hasattr(network, "set_batch_lengths") and not os.environ.get("ADN_RUNNER_PADDED"):
                rows = batch.idxs if not host_batches else (np.asarray(batch_idxs)[mine] if dp is not None else batch_idxs)
                if len(rows):
                    network.set_batch_lengths(np.asarray(tr_lens)[np.asarray(rows, dtype=np.int64)])
            if rank == 0 and progress:
                print('Epoch {} batch {}/{}: {} examples using adam with learning rate = {}'.format(
                    epoch + 1, i + 1, epochsize, n_examples, learning_rate), end='')
                sys.stdout.flush()
            if lr_map is not None and dp is None:
                network.compute_grads(Xs_r, y_r, m_r, windowsize, want_loss=False)
                network.apply_adam_vlr(lr_map)
            elif dp is None:
                # train(...): the cost it returns is discarded by the reference loop (runners/3stream.py:370) -- not waited for
                network.train_step(Xs_r, y_r, m_r, windowsize, learning_rate, want_loss=False)
            else:
                # (a short last minibatch can leave high ranks without an utterance: they contribute zero gradients)
                upd = (lambda mdl: mdl.apply_adam_vlr(lr_map)) if lr_map is not None else None
                dp.train_step(Xs_r, y_r, m_r, windowsize, learning_rate, total_frames, update=upd)
            if rank == 0 and progress:
                print('\r', end='')
        network.synchronize()                        # (the cost call below waits for the steps anyway)
        train_seconds.append(time.time() - time_start)
        # the train cost of an epoch is the cost of its LAST minibatch, re-evaluated after the update (App. E-7), on the WHOLE
        # minibatch: a data-parallel rank that holds only its rows gathers the others for this one call
        if not host_batches and dp is not None:
            full = resident['train'].gather(batch.global_idxs, slot='epoch_cost')
            Xs, y, m = full.Xs, full.targets, full.mask
        elif not host_batches:
            Xs, y, m = batch.Xs, batch.targets, batch.mask
        cost = float(call(compute_train_cost, Xs, y, m, windowsize))
        # validation cost and the predictions evaluate_model2 votes on: one forward pass over the split instead of the
        # reference's two (compute_test_cost, then val_fn: the same deterministic graph on the same inputs)
        val_probs = None
        if dp is None and hasattr(network, 'loss_and_probs') and not os.environ.get('ADN_TWO_PASS_EVAL'):
            val_cost, val_probs = network.loss_and_probs(X_val, y_val, mask_val, windowsize)
            val_cost = float(val_cost)
        else:
            val_cost = heldout_cost(X_val, y_val, mask_val)
        cost_train.append(cost)
        cost_val.append(val_cost)
        train_strip[epoch % STRIP_SIZE] = cost
        val_window.push(val_cost)
        gl = 100 * (cost_val[-1] / np.min(cost_val) - 1)
        with np.errstate(divide='ignore', invalid='ignore'):    # the strip holds zeros until STRIP_SIZE epochs ran
            pk = 1000 * (np.sum(train_strip) / (STRIP_SIZE * np.min(train_strip)) - 1)
            pq = gl / pk
        cr, val_conf = evaluate_model2(X_val, y_val_evaluate, mask_val, windowsize,
                                       eval_fn if val_probs is None else (lambda *a: val_probs))
        class_rate.append(cr)
        if val_cost < best_val and two_way:          # avletters/3stream.py:382-392: the best epoch's VALIDATION confusion matrix
            best_val, best_cr, test_conf = val_cost, cr, val_conf
            epoch_seconds.append(time.time() - time_start)
            say("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f} ({:.1f}sec)"
                .format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, epoch_seconds[-1]))
            best_params = network.snapshot_params() if hasattr(network, 'snapshot_params') else network.get_all_param_values()
        elif val_cost < best_val:
            best_val, best_cr = val_cost, cr
            test_cr, test_conf = evaluate_model2(X_test, y_test, mask_test, windowsize, eval_fn)
            epoch_seconds.append(time.time() - time_start)
            say("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f}, "
                "Test CR= {:.3f} ({:.1f}sec)".format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, test_cr,
                                                     epoch_seconds[-1]))
            # (kept in HBM: one device copy of the flat buffer instead of a download per tensor; --save_best fetches it at the end)
            best_params = network.snapshot_params() if hasattr(network, 'snapshot_params') else network.get_all_param_values()
        else:
            epoch_seconds.append(time.time() - time_start)
            say("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f} ({:.1f}sec)"
                .format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, epoch_seconds[-1]))
        if epoch >= validation_window and early_stop2(val_window, best_val, validation_window):
            break
        if lr_map is not None and 'explode_layer_lr' in options and epoch + 1 == options['explode_layer_lr'][0]:
            rate = options['explode_layer_lr'][1]
            say('explode {} learning rates to {}'.format(','.join(sorted(options['layer_lr'])), rate))
            from ..custom.updates import generate_lr_map
            lr_map = generate_lr_map(network.get_all_params(trainable=True), {k: rate for k in options['layer_lr']},
                                     learning_rate)
    return dict(best_cr=best_cr, best_val=best_val, test_cr=test_cr, test_conf=test_conf, best_params=best_params,
                cost_train=cost_train, cost_val=cost_val, class_rate=class_rate, epoch_seconds=epoch_seconds, train_seconds=train_seconds,
                heldout=dict(X_val=X_val, y_val=y_val_evaluate, mask_val=mask_val, X_test=X_test, y_test=y_test, mask_test=mask_test))


def _main(n_streams, argv=None, variant=None):
    """variant: None = runners/{1,2,3,4}stream.py; 1 stream: 'noencoder' = runners/1stream_noencoder.py (deltanet_v1 on
    the raw features), 'dct' = runners/1stream_dct.py (host deltas of the DCT features, lstm_classifier_majority_vote);
    2 streams: 'dct' = runners/2stream_dct.py (adenet_v2: encoder stream + encoder-less DCT stream), 'nodelta' =
    runners/2stream_nodelta.py (adenet_v2_nodelta); 'avletters' = avletters/{1,2,3}stream.py: the same INI sections with a
    ``has_encoder`` option per stream, a train / val split by the ``iterVec`` of the ``.mat`` file instead of subject files,
    no test split (results line ``best_cr,best_val``)."""
    if (variant, n_streams) not in ((None, n_streams), ('noencoder', 1), ('dct', 1), ('dct', 2), ('nodelta', 2),
                                    ('avletters', 1), ('avletters', 2), ('avletters', 3)):
        raise ValueError('unknown runner variant %r for %d stream(s)' % (variant, n_streams))
    options = parse_options(argv)
    dist, rank, world = _dist_context()
    if world > 1 and options['seed'] is None:
        options['seed'] = 1234                      # ranks must agree on initial weights and minibatch order
    if options['seed'] is not None:
        np.random.seed(options['seed'])
        las_init.set_rng(np.random.RandomState(options['seed']))
    say = print if rank == 0 else (lambda *a, **k: None)

    config = configparser.ConfigParser()
    config.read(options['config'])
    names = ['stream{}'.format(k + 1) for k in range(n_streams)]
    say('CLI options: {}'.format(list(options.items())))
    say('Reading Config File: {}...'.format(options['config']))
    for sec in names + ['lstm_classifier', 'training']:
        say(config.items(sec))

    say('preprocessing dataset...')
    data = [load_mat_file(config.get(n, 'data')) for n in names]
    imagesizes = [tuple(int(d) for d in config.get(n, 'imagesize').split(',')) if config.has_option(n, 'imagesize') else None
                  for n in names]
    dims = [config.getint(n, 'input_dimensions') for n in names]
    lstm_weights = [sio.loadmat(config.get(n, 'lstm_model')) if config.has_option(n, 'lstm_model') else None
                    for n in names]

    lc = 'lstm_classifier'
    cfg = dict(
        fusiontype=config.get(lc, 'fusiontype') if config.has_option(lc, 'fusiontype') else 'none',
        use_peepholes=config.getboolean(lc, 'use_peepholes'),
        lstm_size=config.getint(lc, 'lstm_size'),
        output_classes=config.getint(lc, 'output_classes'),
        use_blstm=config.getboolean(lc, 'use_blstm') if config.has_option(lc, 'use_blstm') else True,
        use_blstm_substream=(config.getboolean(lc, 'use_blstm_substream')
                             if config.has_option(lc, 'use_blstm_substream') else False))
    if variant == 'avletters':                      # avletters/1stream.py:161: the option's PRESENCE selects the BLSTM
        cfg['use_blstm'] = config.has_option(lc, 'use_blstm')
    cfg['weight_init_fn'] = las_init.select(config.get(lc, 'weight_init'))
    windowsize = config.getint(lc, 'windowsize')
    output_classnames = config.get(lc, 'output_classnames').split(',')
    matlab_target_offset = config.getboolean(lc, 'matlab_target_offset')
    cfg['use_dropout'] = config.has_option(lc, 'use_dropout') and config.getboolean(lc, 'use_dropout')
    if cfg['use_dropout'] and n_streams != 3:
        raise ValueError('use_dropout selects adenet_3stream_dropout: only the 3-stream runner has it (runners/3stream.py:284-291)')

    validation_window = config.getint('training', 'validation_window')
    num_epoch = config.getint('training', 'num_epoch')
    learning_rate = options.get('learning_rate', config.getfloat('training', 'learning_rate'))
    epochsize = config.getint('training', 'epochsize')
    batchsize = config.getint('training', 'batchsize')
    avl = variant == 'avletters'
    if not avl or config.has_option('training', 'train_subjects_file'):
        # (avletters/2stream.py:199-201 reads the three files and never uses them: read, so that a missing file fails alike)
        train_ids = read_data_split_file(config.get('training', 'train_subjects_file'))
        val_ids = read_data_split_file(config.get('training', 'val_subjects_file'))
        test_ids = read_data_split_file(config.get('training', 'test_subjects_file'))

    mats = [d['dataMatrix'].astype('float32') for d in data]
    targets_vec = data[0]['targetsVec'].reshape((-1,)).astype('int64')
    subjects_vec = data[0]['subjectsVec'].reshape((-1,))
    vidlen_vec = data[0]['videoLengthVec'].reshape((-1,))
    if matlab_target_offset:
        targets_vec = targets_vec - 1

    if avl:
        # avletters/3stream.py:217-250 (2stream.py / 1stream.py alike): optional force_align of streams 1 and 2, the
        # pre-split chain per stream, the iterVec split (iterations 1, 2 train; the rest validation), train-split statistics
        from ..utils.preprocessing import create_split_index, force_align, split_videolen
        iter_vec = data[0]['iterVec'].reshape((-1,))
        if n_streams >= 2 and config.getboolean('stream1', 'force_align_data'):
            s1_new, s2_new = force_align((mats[0], targets_vec, vidlen_vec),
                                         (mats[1], data[1]['targetsVec'].reshape((-1,)), data[1]['videoLengthVec'].reshape((-1,))))
            mats[0], targets_vec, vidlen_vec = s1_new
            mats[1] = s2_new[0]
        mats = [presplit_dataprocessing(mats[k], vidlen_vec, config, names[k], imagesize=imagesizes[k]) for k in range(n_streams)]
        indexes = create_split_index(len(mats[0]), vidlen_vec, iter_vec)
        tr_lens, va_lens = split_videolen(vidlen_vec, iter_vec)
        split = dict(train=[], val=[])
        for k in range(n_streams):
            tr, va = mats[k][indexes], mats[k][~indexes]
            if config.getboolean(names[k], 'featurewisenormalize'):       # avletters/3stream.py:102-107
                tr, mean, std = featurewise_normalize_sequence(tr)
                va = (va - mean) / std
            split['train'].append(tr); split['val'].append(va)
        ys = dict(train=targets_vec[indexes].reshape((-1,)), val=targets_vec[~indexes].reshape((-1,)))
        lens = dict(train=np.asarray(tr_lens), val=np.asarray(va_lens))
    elif variant == 'dct' and n_streams == 1:
        # runners/1stream_dct.py:185-206: normalise, mean-remove, host deltas (x3 features) BEFORE the split, then the
        # train-split featurewise normalisation
        X = mats[0]
        if config.getboolean('stream1', 'samplewisenormalize'):
            X = normalize_input(X)
        if config.getboolean('stream1', 'meanremove'):
            X = sequencewise_mean_image_subtraction(X, vidlen_vec)
        X = concat_first_second_deltas(X, vidlen_vec, windowsize)
        parts = split_seq_data(X, targets_vec, subjects_vec, vidlen_vec, train_ids, val_ids, test_ids)
        ys = dict(train=parts[1], val=parts[5], test=parts[9])
        lens = dict(train=parts[2], val=parts[6], test=parts[10])
        tr, va, te = postsplit_datapreprocessing(parts[0], parts[4], parts[8], config, 'stream1')
        split = dict(train=[tr], val=[va], test=[te])
        dims = [3 * dims[0]]
    elif n_streams == 1:
        # runners/1stream.py:175-207: reorder, split, THEN the per-split preprocessing
        if config.getboolean('stream1', 'reorderdata'):
            mats[0] = reorder_data(mats[0], imagesizes[0])
        parts = split_seq_data(mats[0], targets_vec, subjects_vec, vidlen_vec, train_ids, val_ids, test_ids)
        split = dict(zip(('train', 'val', 'test'), ([parts[0]], [parts[4]], [parts[8]])))
        ys = dict(train=parts[1], val=parts[5], test=parts[9])
        lens = dict(train=parts[2], val=parts[6], test=parts[10])
        for k in split:
            X = split[k][0]
            if config.getboolean('stream1', 'meanremove'):
                X = sequencewise_mean_image_subtraction(X, lens[k])
            if config.getboolean('stream1', 'diffimage'):
                X = compute_diff_images(X, lens[k])
            if config.getboolean('stream1', 'samplewisenormalize'):
                X = normalize_input(X)
            split[k][0] = X
        tr, va, te = postsplit_datapreprocessing(split['train'][0], split['val'][0], split['test'][0], config, 'stream1')
        split = dict(train=[tr], val=[va], test=[te])
    else:
        mats = [presplit_dataprocessing(mats[k], vidlen_vec, config, names[k], imagesize=imagesizes[k])
                for k in range(n_streams)]
        if config.has_option('stream1', 'force_align_data') and config.getboolean('stream1', 'force_align_data'):
            orig = [(mats[0], targets_vec, vidlen_vec)]
            for k in range(1, n_streams):
                orig.append((mats[k], data[k]['targetsVec'].reshape((-1,)), data[k]['videoLengthVec'].reshape((-1,))))
            new = multistream_force_align(orig)
            mats[0], targets_vec, vidlen_vec = new[0]
            for k in range(1, n_streams):
                mats[k] = new[k][0]
        split = dict(train=[], val=[], test=[])
        for k in range(n_streams):
            parts = split_seq_data(mats[k], targets_vec, subjects_vec, vidlen_vec, train_ids, val_ids, test_ids)
            tr, va, te = postsplit_datapreprocessing(parts[0], parts[4], parts[8], config, names[k])
            split['train'].append(tr); split['val'].append(va); split['test'].append(te)
            if k == 0:
                ys = dict(train=parts[1], val=parts[5], test=parts[9])
                lens = dict(train=parts[2], val=parts[6], test=parts[10])

    say('constructing end to end model...')
    from ..modelzoo import _factory
    _factory.set_default_precision(options.get('precision') or os.environ.get('ADN_PRECISION', _factory.PRODUCT_DEFAULT_PRECISION))
    if avl:
        has_encoder = [config.getboolean(n, 'has_encoder') for n in names]
        network, l_fuse = build_network_avletters(
            n_streams, has_encoder,
            lambda k: load_decoder(config.get(names[k], 'model'), config.get(names[k], 'shape'), config.get(names[k], 'nonlinearities')),
            dims, lstm_weights, cfg)
    elif variant == 'noencoder':           # runners/1stream_noencoder.py:233-236
        network, l_fuse = deltanet_v1.create_model((None, None, dims[0]), None, (None, None), None, None, cfg['lstm_size'],
                                                   cfg['output_classes'], cfg['weight_init_fn'], cfg['use_peepholes']), None
    elif variant == 'dct' and n_streams == 2:           # runners/2stream_dct.py: stream 2 has no encoder
        ae1 = load_decoder(config.get('stream1', 'model'), config.get('stream1', 'shape'), config.get('stream1', 'nonlinearities'))
        network, l_fuse = adenet_v2.create_model(ae1, (None, None, dims[0]), None, (None, None), None, (None, None, dims[1]),
                                                 None, cfg['lstm_size'], None, cfg['output_classes'], cfg['fusiontype'],
                                                 w_init_fn=cfg['weight_init_fn'], use_peepholes=cfg['use_peepholes'])
    elif variant == 'nodelta':                          # runners/2stream_nodelta.py
        aes = [load_decoder(config.get(n, 'model'), config.get(n, 'shape'), config.get(n, 'nonlinearities')) for n in names]
        network, l_fuse = adenet_v2_nodelta.create_model(aes[0], aes[1], (None, None, dims[0]), None, (None, None), None,
                                                         (None, None, dims[1]), None, cfg['lstm_size'], cfg['output_classes'],
                                                         cfg['fusiontype'], w_init_fn=cfg['weight_init_fn'],
                                                         use_peepholes=cfg['use_peepholes'])
    elif variant == 'dct':               # runners/1stream_dct.py:220-223 (no delta layer: the window argument is unused)
        network, l_fuse = lstm_classifier_majority_vote.create_model((None, None, dims[0]), None, (None, None), None,
                                                                     cfg['lstm_size'], cfg['output_classes'],
                                                                     cfg['weight_init_fn'], cfg['use_peepholes']), None
    else:
        aes = [load_decoder(config.get(n, 'model'), config.get(n, 'shape'), config.get(n, 'nonlinearities'))
               for n in names]
        network, l_fuse = build_network(n_streams, aes, dims, lstm_weights, cfg)
    if rank == 0:
        print_network(network)
    say('compiling model...')
    lr_map = None
    if 'layer_lr' in options:
        from ..custom.updates import generate_lr_map
        lr_map = generate_lr_map(network.get_all_params(trainable=True), options['layer_lr'], learning_rate)
    dp = None
    if world > 1:
        from ..parallel import DataParallel, shard_indices
        dp = DataParallel(network)
        dp.broadcast_parameters(0)

    st = fit(network, split, ys, lens, n_streams, windowsize=windowsize, num_epoch=num_epoch, epochsize=epochsize,
             batchsize=batchsize, validation_window=validation_window, learning_rate=learning_rate, options=options, say=say,
             rank=rank, world=world, dp=dp, lr_map=lr_map, two_way=avl)
    best_cr, best_val, test_cr, test_conf, best_params = st['best_cr'], st['best_val'], st['test_cr'], st['test_conf'], st['best_params']
    cost_train, cost_val, class_rate = st['cost_train'], st['cost_val'], st['class_rate']

    say('Final Model')
    if avl:
        say('CR: {}, val loss: {}'.format(best_cr, best_val))              # avletters/3stream.py:397
    else:
        say('CR: {}, val loss: {}, Test CR: {}'.format(best_cr, best_val, test_cr))
    if rank == 0:
        table_str = plot_confusion_matrix(test_conf, output_classnames, fmt='pipe')
        print('confusion matrix: ')
        print(table_str)
        if l_fuse is not None and cfg['fusiontype'] == 'adasum':
            print('adascale coefficients: {}'.format(l_fuse.get_all_param_values(scaling_param=True)))
        if 'save_plot' in options:
            prefix = options['save_plot']
            plot_validation_cost(cost_train, cost_val, savefilename='{}.validloss.png'.format(prefix))
            with open('{}.confmat.txt'.format(prefix), mode='a') as f:
                f.write(table_str + '\n\n')
        if 'write_results' in options:
            print('writing results to {}'.format(options['write_results']))
            with open(options['write_results'], mode='a') as f:
                f.write('{},{}\n'.format(best_cr, best_val) if avl else '{},{},{}\n'.format(test_cr, best_cr, best_val))
        if 'save_best' in options:
            print('saving best model...')
            if hasattr(network, 'restore_params') and not isinstance(best_params, list):
                network.restore_params(best_params)
            else:
                network.set_all_param_values(best_params)
            save_model_params(network, options['save_best'])
            print('best model saved to {}'.format(options['save_best']))
    if dist is not None:
        dist.destroy_process_group()
    st.update(network=network, windowsize=windowsize)
    return st


def main(n_streams, argv=None, variant=None):
    """(the model-zoo factories take their arithmetic from a module-level default that ``--precision`` sets: restored on the
    way out, so that a process which calls several drivers -- the tests do -- builds every model in the mode it asked for)"""
    from ..modelzoo import _factory
    saved = _factory.DEFAULT_PRECISION
    try:
        return _main(n_streams, argv, variant)
    finally:
        _factory.DEFAULT_PRECISION = saved
