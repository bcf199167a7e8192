"""Drivers for the reference's second family of training scripts -- ``{avletters,cuave,oulu}/{bi,tri}modal*.py`` --
which read ``[data] / [models] / [training]`` ``.ini`` files (SURVEY.md App. B schema 2) instead of the ``[streamK]``
files of ``runners/*.py``:

    script                          reference                                model / objective / update rule
    cuave/bimodal_with_val.py       cuave/bimodal_with_val.py:162-379       adenet_v2 (encoder stream + DCT stream), per-frame
                                                                            softmax + temporal_softmax_loss, adam, majority vote;
                                                                            pre-split ``trData/valData/testData`` .mat (App. C)
    oulu/trimodal_with_val.py       oulu/trimodal_with_val.py:259-520       adenet_v3 (last-timestep head), categorical
                                                                            cross-entropy, adadelta with lr decay, subject split
    avletters/trimodal.py           avletters/trimodal.py:183-450           adenet_v3, adadelta with lr decay, iterVec split,
                                                                            the test split doubles as the validation split
    avletters/bimodal.py            avletters/bimodal.py:296-609            adenet_v2 / adenet_v2_3, temporal loss,
                                                                            ``update_rule`` adadelta | sgdm | sgdnm | adam with
                                                                            ``decay_rate/decay_start``, ``t1`` and
                                                                            ``momentum_schedule``

Same keys, same preprocessing order, same epoch statistics (GL / Pk / PQ), same printed lines and results files.  What
differs, on purpose:
  * the encoders: the reference un-pickles *nolearn* networks (``finetuned``, ``finetuned_diff``); nolearn does not
    exist here, so those paths may point to a ``.mat`` with ``w1..w4 / b1..b4`` (what ``dbn/extractNN.m`` writes), to a
    pickle of ``(weights, biases)`` lists, or to a pickle of any object with ``get_all_layers()`` (a nolearn network
    un-pickles to that where nolearn is installed).  ``do_finetune`` / ``save_finetune`` (nolearn's ``fit``) are refused.
  * constants the reference hard-codes (epochs, epoch size, batch size, validation window, 1144 / 1200 input pixels,
    10 / 26 classes, 250 LSTM units, the ``data/{train,val,test}.txt`` split files) are the DEFAULTS here and can be
    overridden by optional keys of the same section (``num_epoch``, ``epochsize``, ``batchsize``, ``validation_window``,
    ``input_dimension``, ``no_coeff``, ``output_classes``, ``lstm_size``, ``{train,val,test}_subjects_file``), so that
    the same driver runs on small synthetic files in the tests.
"""
from __future__ import print_function

import argparse
import configparser
import pickle
import sys
import time

import numpy as np
import scipy.io as sio

from .. import init as las_init
from ..custom.nonlinearities import select_nonlinearity
from ..modelzoo import adenet_v2, adenet_v2_3, adenet_v3
from ..utils.data_structures import circular_list
from ..utils.datagen import compute_integral_len, gen_lstm_batch_random, gen_seq_batch_from_idx
from ..utils.io import load_mat_file, read_data_split_file
from ..utils.plotting_utils import plot_confusion_matrix, plot_validation_cost, print_network
from ..utils.preprocessing import (compute_diff_images, create_split_index, featurewise_normalize_sequence, normalize_input,
                                   reorder_data, sequencewise_mean_image_subtraction, split_seq_data, split_videolen)
from ..utils.regularization import early_stop, early_stop2
from .nstream import evaluate_model2

SCRIPTS = {('cuave', 'bimodal_with_val'), ('oulu', 'trimodal_with_val'), ('avletters', 'trimodal'), ('avletters', 'bimodal')}


# --------------------------------------------------------------------------------------------------------- encoders
def load_dbn(path, shapes=(2000, 1000, 500, 50), nonlinearities=('rectify', 'rectify', 'rectify', 'linear')):
    """``.mat`` with w1..w4 / b1..b4 -> (weights, biases, shapes, nonlinearities) (cuave/bimodal_with_val.py:31-51)."""
    nn = sio.loadmat(path)
    n = len(shapes)
    weights = [nn['w{}'.format(i + 1)].astype('float32') for i in range(n)]
    biases = [nn['b{}'.format(i + 1)][0].astype('float32') for i in range(n)]
    return weights, biases, [int(w.shape[1]) for w in weights], [select_nonlinearity(a) for a in nonlinearities]


def load_ae(path):
    """What the tri-modal scripts un-pickle as ``ae`` / ``diff_ae`` (avletters/trimodal.py:268-276): returns something
    ``modelzoo._factory.nolearn_weights`` accepts."""
    if path.endswith('.mat'):
        weights, biases, _, _ = load_dbn(path)
        return weights, biases
    with open(path, 'rb') as f:
        obj = pickle.load(f)
    if hasattr(obj, 'get_all_layers') or (isinstance(obj, (tuple, list)) and len(obj) in (2, 4)):
        return obj
    raise ValueError('%s: expected a .mat with w1..w4/b1..b4, a pickled (weights, biases) pair or a network object' % path)


# --------------------------------------------------------------------------------------------------------- options
def parse_options(argv, default_config):
    """Union of the scripts' command lines (avletters/bimodal.py:242-289 is the richest); every value overrides the
    key of the same name in the .ini."""
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', help='config file to use, default=' + default_config)
    parser.add_argument('--write_results', help='write results to file')
    parser.add_argument('--update_rule', help='adadelta, sgdm, sgdnm, adam')
    parser.add_argument('--learning_rate', help='learning rate')
    parser.add_argument('--decay_rate', help='learning rate decay')
    parser.add_argument('--momentum', help='momentum')
    parser.add_argument('--momentum_schedule', help='eg: 0.9,0.9,0.95,0.99')
    parser.add_argument('--validation_window', help='validation window length, eg: 6')
    parser.add_argument('--t1', help='epoch to start learning rate decay, eg: 10')
    parser.add_argument('--weight_init', help='norm,glorot,ortho,uniform')
    parser.add_argument('--num_epoch', help='number of epochs to run')
    parser.add_argument('--use_peepholes', action='store_true', help='use peephole connections in LSTM')
    parser.add_argument('--no_plot', dest='no_plot', action='store_true', help='disable plots')
    parser.add_argument('--seed', type=int, default=None, help='seed for initialisers, dropout and minibatch order '
                                                               '(the reference never seeds)')
    args = parser.parse_args(argv)
    options = {'config': args.config or default_config, 'no_plot': bool(args.no_plot), 'seed': args.seed}
    for key in ('write_results', 'update_rule', 'learning_rate', 'decay_rate', 'momentum', 'momentum_schedule',
                'validation_window', 't1', 'weight_init', 'num_epoch'):
        if getattr(args, key):
            options[key] = getattr(args, key)
    if args.use_peepholes:
        options['use_peepholes'] = True
    return options


class _Cfg(object):
    """option-then-.ini lookup with the reference's precedence (CLI wins) and optional defaults for the constants the
    reference hard-codes."""

    def __init__(self, config, options):
        self.config, self.options = config, options

    def get(self, section, key, conv=str, default=None):
        if key in self.options and section == 'training':
            return conv(self.options[key])
        if self.config.has_option(section, key):
            raw = self.config.get(section, key)
            if conv is bool:
                return self.config.getboolean(section, key)
            return conv(raw)
        if default is None:
            raise configparser.NoOptionError(key, section)
        return default


# --------------------------------------------------------------------------------------------------------- updates
class Updater(object):
    """The ``updates`` dictionary of the reference scripts as a callable: one training step with the CURRENT values of
    the shared variables ``lr`` / ``mm`` (avletters/bimodal.py:446-455; oulu/trimodal_with_val.py:388)."""

    def __init__(self, network, rule, learning_rate, momentum=0.9):
        if rule not in ('adadelta', 'sgdm', 'sgdnm', 'adam'):
            raise ValueError('update_rule must be one of adadelta, sgdm, sgdnm, adam (got %r)' % (rule,))
        self.network, self.rule = network, rule
        self.lr, self.mm = float(learning_rate), float(momentum)

    def __call__(self, inputs, targets, mask, window):
        net = self.network
        if self.rule == 'adam':                 # las.updates.adam(cost, all_params): DEFAULT parameters, lr is not passed
            return net.train_step(inputs, targets, mask, window, 1e-3)
        cost = net.compute_grads(inputs, targets, mask, window)
        if self.rule == 'adadelta':
            net.apply_adadelta(self.lr)
        else:                                   # sgd + apply_momentum / apply_nesterov_momentum
            net.apply_sgd(self.lr, self.mm, nesterov=(self.rule == 'sgdnm'))
        return cost


def evaluate_model(X_vals, y_val, mask_val, window_size, eval_fn):
    """Last-timestep models: arg-max of the (B, C) output (oulu/trimodal_with_val.py:215-239)."""
    output = eval_fn(*(list(X_vals) + [mask_val, window_size]))
    no_gps = output.shape[1]
    ix = np.argmax(output, axis=1)
    y_val = np.asarray(y_val).astype(int)
    classification_rate = np.sum(ix == y_val) / float(len(ix))
    confusion_matrix = np.zeros((no_gps, no_gps), dtype='int')
    np.add.at(confusion_matrix, (y_val, ix), 1)
    return classification_rate, confusion_matrix


# --------------------------------------------------------------------------------------------------------- data
def _vec(d, key, dtype='int'):
    return d[key].astype(dtype).reshape((-1,))


def _load_cuave(cfg):
    """Pre-split CUAVE .mat (cuave/bimodal_with_val.py:211-249): reorder to C order, per-sequence mean removal,
    per-frame z-normalisation; DCT features z-normalised feature-wise with the TRAIN statistics."""
    data = load_mat_file(cfg.get('data', 'images'))
    dct_data = load_mat_file(cfg.get('data', 'dct'))
    imagesize = tuple(int(v) for v in cfg.get('data', 'imagesize', str, '30,50').split(','))
    split, lens, ys = {}, {}, {}
    for k, pre in (('train', 'tr'), ('val', 'val'), ('test', 'test')):
        lens[k] = _vec(data, pre + 'VideoLengthVec')
        X = data[pre + 'Data'].astype('float32')
        ys[k] = _vec(data, pre + 'TargetsVec') + 1          # +1 to handle the -1 introduced in lstm_gendata (:223)
        X = reorder_data(X, imagesize)
        X = sequencewise_mean_image_subtraction(X, lens[k])
        X = normalize_input(X)
        split[k] = [X, dct_data[pre + 'DctFeatures'].astype('float32')]
    tr, mean, std = featurewise_normalize_sequence(split['train'][1])
    split['train'][1] = tr
    for k in ('val', 'test'):
        split[k][1] = (split[k][1] - mean) / std
    return split, ys, lens


def _load_subject_split(cfg):
    """OuluVS tri-modal (oulu/trimodal_with_val.py:274-334): diff images of the RAW frames, per-sequence mean removal of
    the DCT features, subject split, per-frame z-normalisation of the raw split (NOT of the diff images), train-split
    feature-wise normalisation of the DCT features.  Stream order of adenet_v3: raw, dct, diff."""
    data = load_mat_file(cfg.get('data', 'images'))
    dct_data = load_mat_file(cfg.get('data', 'dct'))
    X = data['dataMatrix'].astype('float32')
    y = _vec(data, 'targetsVec', 'int32')
    dct_feats = dct_data['dctFeatures'].astype('float32')
    subjects = _vec(data, 'subjectsVec')
    video_lens = _vec(data, 'videoLengthVec')
    X_diff = compute_diff_images(X, video_lens)
    dct_feats = sequencewise_mean_image_subtraction(dct_feats, video_lens)
    ids = [read_data_split_file(cfg.get('training', k + '_subjects_file', str, 'data/%s.txt' % k)) for k in ('train', 'val', 'test')]
    split = dict(train=[], val=[], test=[])
    for mat in (X, dct_feats, X_diff):
        parts = split_seq_data(mat, y, subjects, video_lens, ids[0], ids[1], ids[2])
        split['train'].append(parts[0]); split['val'].append(parts[4]); split['test'].append(parts[8])
        ys = dict(train=parts[1], val=parts[5], test=parts[9])
        lens = dict(train=parts[2], val=parts[6], test=parts[10])
    for k in split:
        split[k][0] = normalize_input(split[k][0], centralize=True)
    tr, mean, std = featurewise_normalize_sequence(split['train'][1])
    split['train'][1] = tr
    for k in ('val', 'test'):
        split[k][1] = (split[k][1] - mean) / std
    return split, ys, lens


def _load_avletters(cfg, with_diff, normalise_images, target_offset):
    """AVLetters (avletters/trimodal.py:198-251, bimodal.py:343-375): repetitions 1-2 train, 3 test (iterVec); the test
    split is also the validation split; DCT features feature-wise normalised with the train statistics."""
    data = load_mat_file(cfg.get('data', 'images'))
    dct_data = load_mat_file(cfg.get('data', 'dct'))
    data_matrix = data['dataMatrix'].astype('float32')
    targets_vec = _vec(data, 'targetsVec')
    vid_len_vec = _vec(data, 'videoLengthVec')
    iter_vec = _vec(data, 'iterVec')
    dct_feats = dct_data['dctFeatures'].astype('float32')
    if target_offset:
        targets_vec = targets_vec - 1                       # bimodal.py:351
    if normalise_images:
        data_matrix = normalize_input(data_matrix, True)    # bimodal.py:354
    mats = [data_matrix, dct_feats]
    if with_diff:
        mats.append(load_mat_file(cfg.get('data', 'diff'))['dataMatrix'].astype('float32'))
    indexes = create_split_index(len(data_matrix), vid_len_vec, iter_vec)
    train_lens, test_lens = split_videolen(vid_len_vec, iter_vec)
    assert np.sum(vid_len_vec) == len(data_matrix)
    split = dict(train=[m[indexes] for m in mats], test=[m[~indexes] for m in mats])
    tr, mean, std = featurewise_normalize_sequence(split['train'][1].astype(np.float32))
    split['train'][1] = tr
    split['test'][1] = (split['test'][1].astype(np.float32) - mean) / std
    split['val'] = split['test']
    ys = dict(train=targets_vec[indexes], test=targets_vec[~indexes])
    ys['val'] = ys['test']
    lens = dict(train=np.asarray(train_lens, int), test=np.asarray(test_lens, int))
    lens['val'] = lens['test']
    return split, ys, lens


# --------------------------------------------------------------------------------------------------------- driver
def main(dataset, script, argv=None):
    if (dataset, script) not in SCRIPTS:
        raise ValueError('no driver for %s/%s.py (have: %s)' % (dataset, script, sorted(SCRIPTS)))
    default_cfg = {'bimodal_with_val': 'config/bimodal_meanrm_raw_dct.ini', 'trimodal_with_val': 'config/trimodal.ini',
                   'trimodal': 'config/trimodal.ini', 'bimodal': 'config/bimodal.ini'}[script]
    options = parse_options(argv, default_cfg)
    if options['seed'] is not None:
        np.random.seed(options['seed'])
        las_init.set_rng(np.random.RandomState(options['seed']))
    config = configparser.ConfigParser()
    if not config.read(options['config']):
        raise IOError('cannot read config file %s' % options['config'])
    cfg = _Cfg(config, options)
    print('CLI options: {}'.format(list(options.items())))
    print('Reading Config File: {}...'.format(options['config']))
    for sec in ('data', 'models', 'training'):
        print(config.items(sec))
    print('preprocessing dataset...')

    trimodal = script in ('trimodal', 'trimodal_with_val')
    frames_head = not trimodal                       # per-frame softmax + temporal loss vs. last timestep + cross-entropy
    fusiontype = cfg.get('models', 'fusiontype')
    if trimodal:
        for key in ('do_finetune', 'save_finetune'):
            if cfg.get('training', key, bool, False):
                raise NotImplementedError('%s: nolearn auto-encoder fine-tuning is outside this package (SURVEY 8: out of scope); '
                                          'fine-tune offline and point `finetuned` to the weights' % key)
    # ---- data
    if dataset == 'cuave':
        split, ys, lens = _load_cuave(cfg)
    elif dataset == 'oulu':
        split, ys, lens = _load_subject_split(cfg)
    elif script == 'trimodal':
        split, ys, lens = _load_avletters(cfg, with_diff=True, normalise_images=False, target_offset=False)
    else:
        split, ys, lens = _load_avletters(cfg, with_diff=False, normalise_images=True, target_offset=True)
    n_streams = len(split['train'])

    # ---- hyper-parameters (reference constants as defaults)
    consts = {('cuave', 'bimodal_with_val'): dict(num_epoch=None, epochsize=None, batchsize=None, validation_window=None, dim=1500,
                                                  classes=10, names='0,1,2,3,4,5,6,7,8,9'),
              ('oulu', 'trimodal_with_val'): dict(num_epoch=12, epochsize=120, batchsize=10, validation_window=4, dim=1144, classes=10,
                                                  names='p1,p2,p3,p4,p5,p6,p7,p8,p9,p10'),
              ('avletters', 'trimodal'): dict(num_epoch=25, epochsize=20, batchsize=26, validation_window=4, dim=1200, classes=26,
                                              names=','.join('abcdefghijklmnopqrstuvwxyz')),
              ('avletters', 'bimodal'): dict(num_epoch=None, epochsize=20, batchsize=26, validation_window=None, dim=1200, classes=26,
                                             names=','.join('abcdefghijklmnopqrstuvwxyz'))}[(dataset, script)]
    num_epoch = cfg.get('training', 'num_epoch', int, consts['num_epoch'])
    epochsize = cfg.get('training', 'epochsize', int, consts['epochsize'])
    batchsize = cfg.get('training', 'batchsize', int, consts['batchsize'])
    validation_window = cfg.get('training', 'validation_window', int, consts['validation_window'])
    learning_rate = cfg.get('training', 'learning_rate', float)
    input_dimension = cfg.get('models', 'input_dimension', int, consts['dim'])
    output_classes = cfg.get('models', 'output_classes', int, consts['classes'])
    lstm_size = cfg.get('models', 'lstm_size', int, 250)
    no_coeff = cfg.get('models', 'no_coeff', int, 30)
    classnames = cfg.get('models', 'output_classnames', str, consts['names']).split(',')
    WINDOW_SIZE = cfg.get('models', 'delta_window', int, 9)
    STRIP_SIZE = 3

    # ---- model
    print('constructing end to end model...')
    ms = (None, None)
    if trimodal:
        decay_rate = cfg.get('training', 'decay_rate', float)
        decay_start = cfg.get('training', 'decay_start', int)
        if not cfg.get('training', 'load_finetune', bool, True) or not cfg.get('training', 'load_finetune_diff', bool, True):
            raise ValueError('load_finetune / load_finetune_diff must be true: the scripts define `ae` / `diff_ae` nowhere else')
        ae = load_ae(cfg.get('models', 'finetuned'))
        ae_diff = load_ae(cfg.get('models', 'finetuned_diff'))
        network, l_fuse = adenet_v3.create_model(ae, ae_diff, (None, None, input_dimension), None, ms, None,
                                                 (None, None, split['train'][1].shape[1]), None,
                                                 (None, None, split['train'][2].shape[1]), None, lstm_size, None, output_classes,
                                                 fusiontype)
        update = Updater(network, 'adadelta', learning_rate)
        rule = 'adadelta'
    elif dataset == 'cuave':
        weight_init = cfg.get('training', 'weight_init')
        use_peepholes = cfg.get('training', 'use_peepholes', bool)
        nonlinearity = select_nonlinearity(cfg.get('models', 'nonlinearity'))
        cfg.get('training', 'use_blstm', bool); cfg.get('training', 'use_finetuning', bool)      # read (and required) like the script
        ae = load_dbn(cfg.get('models', 'pretrained'))
        network, l_fuse = adenet_v2.create_model(ae, (None, None, input_dimension), None, ms, None, (None, None, no_coeff * 3), None,
                                                 lstm_size, None, output_classes, fusiontype, las_init.select(weight_init),
                                                 use_peepholes, nonlinearity)
        update = Updater(network, 'adam', learning_rate)       # (only carries lr for the progress line; see `train` below)
        rule = 'adam'
    else:                                            # avletters/bimodal.py
        rule = cfg.get('training', 'update_rule')
        decay_rate = cfg.get('training', 'decay_rate', float)
        decay_start = cfg.get('training', 'decay_start', int)
        t1 = cfg.get('training', 't1', int)
        weight_init = cfg.get('training', 'weight_init')
        use_peepholes = cfg.get('training', 'use_peepholes', bool)
        use_blstm = cfg.get('training', 'use_blstm', bool)
        use_finetuning = cfg.get('training', 'use_finetuning', bool)
        momentum, mm_schedule = 0.9, []
        if rule in ('sgdm', 'sgdnm'):
            momentum = cfg.get('training', 'momentum', float)
            mm_schedule = [float(m) for m in cfg.get('training', 'momentum_schedule').split(',')]
        dbn = load_ae(cfg.get('models', 'pretrained'))
        if not isinstance(dbn, (tuple, list)) or len(dbn) == 2:
            from ..modelzoo._factory import nolearn_weights
            dbn = nolearn_weights(dbn, nonlinearities=('rectify', 'rectify', 'rectify', 'linear'))
        factory = adenet_v2 if use_blstm else adenet_v2_3
        network, l_fuse = factory.create_model(dbn, (None, None, input_dimension), None, ms, None, (None, None, no_coeff * 3), None,
                                               lstm_size, None, output_classes, fusiontype,
                                               w_init_fn=las_init.select(weight_init), use_peepholes=use_peepholes)
        update = Updater(network, rule, learning_rate, momentum)
    print_network(network)
    print('compiling model...')
    if rule == 'adam' and dataset == 'cuave':        # adam(cost, all_params, learning_rate=learning_rate) (:286)
        def train(ins, y, m, w):
            return network.train_step(ins, y, m, w, learning_rate)
    else:
        train = update
    compute_train_cost = lambda ins, y, m, w: network.loss(ins, y, m, w, deterministic=False)
    compute_test_cost = lambda ins, y, m, w: network.loss(ins, y, m, w)
    eval_fn = lambda *a: network.predict(list(a[:n_streams]), a[n_streams], a[n_streams + 1])
    evaluate = evaluate_model2 if frames_head else evaluate_model

    # ---- loop
    print('begin training...')
    cost_train, cost_val, class_rate = [], [], []
    val_window = circular_list(validation_window)
    train_strip = np.zeros((STRIP_SIZE,))
    best_val, best_tr, best_cr, best_conf, test_cr, test_conf, adascale_param = float('inf'), float('inf'), 0.0, None, None, None, None
    tr_lens = np.asarray(lens['train'], int)
    tmax_train = int(np.max(tr_lens))
    datagen = gen_lstm_batch_random(split['train'][0], ys['train'], tr_lens, batchsize=batchsize)
    integral_lens = compute_integral_len(tr_lens)

    def whole_split(k):
        ln = np.asarray(lens[k], int)
        X1, y, m, idxs = next(gen_lstm_batch_random(split[k][0], ys[k], ln, batchsize=len(ln)))
        il = compute_integral_len(ln)
        return [X1] + [gen_seq_batch_from_idx(split[k][s], idxs, ln, il, np.max(ln)) for s in range(1, n_streams)], y, m

    def targets_of(y, m):                            # per-frame targets for the temporal loss; the last-step head takes (B,)
        y = np.asarray(y).reshape((-1, 1)).repeat(m.shape[-1], axis=-1)
        return y

    X_val, y_val_evaluate, mask_val = whole_split('val')
    y_val = targets_of(y_val_evaluate, mask_val)
    has_test = dataset in ('cuave', 'oulu')          # the AVLetters scripts evaluate on their validation (= test) split only
    if has_test:
        X_test, y_test, mask_test = whole_split('test')
    stop = (lambda w, best: early_stop(w)) if trimodal else (lambda w, best: early_stop2(w, best, validation_window))

    for epoch in range(num_epoch):
        time_start = time.time()
        for i in range(epochsize):
            X1, y, m, batch_idxs = next(datagen)
            yy = targets_of(y, m)
            Xs = [X1] + [gen_seq_batch_from_idx(split['train'][s], batch_idxs, tr_lens, integral_lens, tmax_train)
                         for s in range(1, n_streams)]
            if rule == 'adam' and dataset != 'cuave':
                msg = 'Epoch {} batch {}/{}: {} examples with {} using default params'.format(epoch + 1, i + 1, epochsize, len(X1), rule)
            elif rule in ('sgdm', 'sgdnm'):
                msg = 'Epoch {} batch {}/{}: {} examples at learning rate = {:.4f}, momentum = {:.4f} with {}'.format(
                    epoch + 1, i + 1, epochsize, len(X1), update.lr, update.mm, rule)
            else:
                msg = 'Epoch {} batch {}/{}: {} examples at learning rate = {:.4f}'.format(epoch + 1, i + 1, epochsize, len(X1),
                                                                                          update.lr)
            print(msg, end='')
            sys.stdout.flush()
            train(Xs, yy, m, WINDOW_SIZE)
            print('\r', end='')
        cost = float(compute_train_cost(Xs, yy, m, WINDOW_SIZE))
        val_cost = float(compute_test_cost(X_val, y_val, mask_val, WINDOW_SIZE))
        cost_train.append(cost)
        cost_val.append(val_cost)
        train_strip[epoch % STRIP_SIZE] = cost
        val_window.push(val_cost)
        gl = 100 * (cost_val[-1] / np.min(cost_val) - 1)
        with np.errstate(divide='ignore', invalid='ignore'):
            pk = 1000 * (np.sum(train_strip) / (STRIP_SIZE * np.min(train_strip)) - 1)
            pq = gl / pk
        cr, val_conf = evaluate(X_val, y_val_evaluate, mask_val, WINDOW_SIZE, eval_fn)
        class_rate.append(cr)
        improved = val_cost < best_val
        if has_test:
            if improved:
                best_val, best_tr, best_conf, best_cr = val_cost, cost, val_conf, cr
                if fusiontype == 'adasum':
                    adascale_param = l_fuse.get_all_param_values(scaling_param=True)
                test_cr, test_conf = evaluate(X_test, y_test, mask_test, WINDOW_SIZE, eval_fn)
                print("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f}, Test CR= {:.3f} "
                      "({:.1f}sec)".format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, test_cr, time.time() - time_start))
            else:
                print("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f} ({:.1f}sec)"
                      .format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, time.time() - time_start))
        else:
            print("Epoch {} train cost = {}, validation cost = {}, generalization loss = {:.3f}, GQ = {:.3f}, "
                  "classification rate = {:.3f} ({:.1f}sec)".format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr,
                                                                    time.time() - time_start))
            if improved:
                best_val, best_tr, best_conf, best_cr = val_cost, cost, val_conf, cr
                if fusiontype == 'adasum':
                    adascale_param = l_fuse.get_all_param_values(scaling_param=True)
            elif script == 'bimodal' and epoch >= t1 and rule in ('sgdm', 'sgdnm'):     # avletters/bimodal.py:541-545
                update.lr = max(update.lr * decay_rate, 0.001)
                if mm_schedule:
                    update.mm = mm_schedule.pop(0)
        if epoch >= validation_window and stop(val_window, best_val):
            break
        if dataset != 'cuave' and epoch + 1 >= decay_start:      # learning rate decay (oulu/trimodal_with_val.py:508-510)
            update.lr = float(np.float32(update.lr) * np.float32(decay_rate))

    # ---- report
    if has_test:
        print('Final Model')
        print('CR: {}, val loss: {}, Test CR: {}'.format(best_cr, best_val, test_cr))
    else:
        print('Best Model')
        print('classification rate: {}, validation loss: {}'.format(best_cr, best_val))
    if fusiontype == 'adasum':
        if script == 'bimodal':
            adascale_param = l_fuse.get_all_param_values(scaling_param=True)
        print("final scaling params: {}".format(adascale_param))
    print('confusion matrix: ')
    conf = test_conf if has_test else best_conf
    if not options['no_plot'] and conf is not None:
        print(plot_confusion_matrix(conf, classnames[:conf.shape[0]], fmt='latex' if script != 'bimodal' else 'pipe'))
        try:
            plot_validation_cost(cost_train, cost_val, savefilename='valid_cost' if has_test else 'e2e_valid_cost')
        except Exception as e:                        # matplotlib is optional here
            print('(no plot: %s)' % e)
    if options.get('write_results'):
        with open(options['write_results'], mode='a') as f:
            if dataset == 'cuave':                    # cuave/bimodal_with_val.py:376-381
                f.write('{},{},{},{},{},{},{},{},{},{},{},{}\n'.format(
                    cfg.get('training', 'use_finetuning', bool), 'yes', use_peepholes, 'adam', weight_init, 'RELU',
                    cfg.get('training', 'use_blstm', bool), learning_rate, best_tr, best_val, best_cr * 100, test_cr * 100))
                for series in (cost_train, cost_val, class_rate):
                    f.write('{}\n'.format(','.join(str(v) for v in series)))
            elif script == 'bimodal':                 # avletters/bimodal.py:592-606
                f.write('{},{},{},{},{}\n'.format(validation_window, weight_init, use_peepholes, use_blstm, use_finetuning))
                for series in (cost_train, cost_val, class_rate):
                    f.write('{}\n'.format(','.join(str(v) for v in series)))
                f.write('{},{},{}\n'.format(fusiontype, best_cr, best_val))
            else:
                f.write('{},{},{}\n'.format(fusiontype, best_cr, best_val))
    return dict(best_cr=best_cr, best_val=best_val, test_cr=test_cr, cost_train=cost_train, cost_val=cost_val,
                class_rate=class_rate, network=network, learning_rate=update.lr, momentum=update.mm)
