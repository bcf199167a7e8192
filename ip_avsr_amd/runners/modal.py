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

    cuave/unimodal_with_val.py      cuave/unimodal_with_val.py:150-372      deltanet_majority_vote (pretrained encoder), subject
                                                                            split of the whole-set .mat, adam(lr)
    cuave/unimodal_dct_with_val.py  cuave/unimodal_dct_with_val.py:130-329  lstm_classifier_majority_vote on the DCT features
                                                                            (no delta layer), adam with DEFAULT parameters
    cuave/trimodal_with_val.py      cuave/trimodal_with_val.py:163-377      adenet_v3 on the pre-split file (raw as stored, DCT,
                                                                            diff of raw), adadelta with lr decay
    cuave/audio_visual_runner.py    cuave/audio_visual_runner.py:242-472    avnet: visual + audio encoder sub-streams
                                                                            (create_pretrained_substream x 2 -> create_model)
    oulu/unimodal_with_val.py       oulu/unimodal_with_val.py:219-403       deltanet_majority_vote, subject split, adam(lr)
    oulu/bimodal_with_val.py        oulu/bimodal_with_val.py:217-451        adenet_v2 (encoder + DCT stream), adam(lr); the
                                                                            script overwrites the .ini's epochsize / batchsize
                                                                            with 120 / 10 (:366-367)
    avletters/bimodal_diff_image.py avletters/bimodal_diff_image.py:215-498 adenet_v2_1 (raw + diff encoders, last-timestep
                                                                            head), the four update rules, BOTH decay rules
    avletters/unimodal.py           avletters/unimodal.py:121-304           schema-1 ``[stream1]`` .ini, iterVec split,
                                                                            deltanet_majority_vote / deltanet_v1, adam(lr)

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
import os
import configparser
import pickle
import sys
import time

import numpy as np
import scipy.io as sio

from .. import init as las_init
from ..custom.nonlinearities import select_nonlinearity
from ..modelzoo import (adenet_v2, adenet_v2_1, adenet_v2_3, adenet_v3, avnet, deltanet_majority_vote, deltanet_v1,
                        lstm_classifier_majority_vote)
from ..utils.data_structures import circular_list
from ..utils.datagen import compute_integral_len, gen_lstm_batch_random, gen_seq_batch_from_idx
from ..utils.io import load_mat_file, read_data_split_file
from ..utils.plotting_utils import plot_confusion_matrix, plot_validation_cost, print_network
from ..utils.preprocessing import (compute_diff_images, create_split_index, featurewise_normalize_sequence, normalize_input,
                                   reorder_data, sequencewise_mean_image_subtraction, split_seq_data, split_videolen)
from ..utils.regularization import early_stop, early_stop2
from .nstream import evaluate_model2

# (dataset, script) -> default --config of the reference script
SCRIPTS = {('cuave', 'bimodal_with_val'): 'config/bimodal_meanrm_raw_dct.ini',
           ('cuave', 'unimodal_with_val'): 'config/unimodal_meanrmraw.ini',
           ('cuave', 'unimodal_dct_with_val'): 'config/unimodal_dct.ini',
           ('cuave', 'trimodal_with_val'): 'config/trimodal.ini',
           ('cuave', 'audio_visual_runner'): 'config/avnet.ini',
           ('oulu', 'trimodal_with_val'): 'config/trimodal.ini',
           ('oulu', 'unimodal_with_val'): 'config/unimodal.ini',
           ('oulu', 'bimodal_with_val'): 'config/bimodal.ini',
           ('avletters', 'trimodal'): 'config/trimodal.ini',
           ('avletters', 'bimodal'): 'config/bimodal.ini',
           ('avletters', 'bimodal_diff_image'): 'config/bimodal_diff_image.ini',
           ('avletters', 'unimodal'): 'config/normal.ini'}


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
    parser.add_argument('--no_epochs', help='Max epochs to run (cuave/unimodal*_with_val.py)')
    parser.add_argument('--epochsize', help='Number of mini batches to run for each epoch')
    parser.add_argument('--batchsize', help='Mini batch size')
    parser.add_argument('--save_best', help='save best model (cuave/unimodal_with_val.py)')
    parser.add_argument('--use_peepholes', action='store_true', help='use peephole connections in LSTM')
    parser.add_argument('--no_plot', dest='no_plot', action='store_true', help='disable plots')
    parser.add_argument('--seed', type=int, default=None, help='seed for initialisers, dropout and minibatch order '
                                                               '(the reference never seeds)')
    parser.add_argument('--precision', default=None, choices=['f32', 'bf16x3', 'mixed', 'bf16'],
                        help='arithmetic of the model (default bf16x3: fp32-grade products on the bf16 matrix pipe, the 1e-4 parity '
                             'gate against the fp32 reference; f32 = exact fp32 MFMA products, diagnostic, ~5x slower); also ADN_PRECISION')
    args = parser.parse_args(argv)
    options = {'config': args.config or default_config, 'no_plot': bool(args.no_plot), 'seed': args.seed,
               'precision': args.precision}
    for key in ('write_results', 'update_rule', 'learning_rate', 'decay_rate', 'momentum', 'momentum_schedule',
                'validation_window', 't1', 'weight_init', 'num_epoch', 'no_epochs', 'epochsize', 'batchsize', 'save_best'):
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
        """``key`` may be a tuple of spellings (the scripts disagree: num_epoch / no_epochs, lstm_size / lstm_units,
        no_coeff / no_coeffs, input_dimension / input_dimensions); the first one present wins."""
        keys = key if isinstance(key, tuple) else (key,)
        for k in keys:
            if k in self.options and section == 'training':
                return conv(self.options[k])
        for k in keys:
            if self.config.has_option(section, k):
                if conv is bool:
                    return self.config.getboolean(section, k)
                return conv(self.config.get(section, k))
        if default is None:
            raise configparser.NoOptionError(keys[0], section)
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

    def __call__(self, inputs, targets, mask, window, want_loss=False):
        """(the loops discard the cost ``train`` returns -- avletters/bimodal.py:515 -- so it is not waited for by default)"""
        net = self.network
        if self.rule == 'adam':                 # las.updates.adam(cost, all_params): DEFAULT parameters, lr is not passed
            return net.train_step(inputs, targets, mask, window, 1e-3, want_loss=want_loss)
        cost = net.compute_grads(inputs, targets, mask, window, want_loss=want_loss)
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




def _presplit_cuave(cfg, key='images'):
    """The pre-split CUAVE file (App. C): per split its frames, lengths and targets (+1: the -1 introduced in lstm_gendata)."""
    data = load_mat_file(cfg.get('data', key))
    X, lens, ys = {}, {}, {}
    for k, pre in (('train', 'tr'), ('val', 'val'), ('test', 'test')):
        lens[k] = _vec(data, pre + 'VideoLengthVec')
        X[k] = data[pre + 'Data'].astype('float32')
        if pre + 'TargetsVec' in data:
            ys[k] = _vec(data, pre + 'TargetsVec') + 1
    return X, lens, ys


def _cuave_dct(cfg, with_train_stats=True):
    dct_data = load_mat_file(cfg.get('data', 'dct'))
    d = {k: dct_data[pre + 'DctFeatures'].astype('float32') for k, pre in (('train', 'tr'), ('val', 'val'), ('test', 'test'))}
    tr, mean, std = featurewise_normalize_sequence(d['train'])
    return dict(train=tr, val=(d['val'] - mean) / std, test=(d['test'] - mean) / std)


def _load_cuave_dct(cfg):
    """cuave/unimodal_dct_with_val.py:176-194: the DCT features alone (lengths / targets from the images file)."""
    _, lens, ys = _presplit_cuave(cfg)
    d = _cuave_dct(cfg)
    return {k: [d[k]] for k in d}, ys, lens


def _load_cuave_trimodal(cfg):
    """cuave/trimodal_with_val.py:189-208: raw frames AS STORED (no reordering, mean removal or normalisation), the
    train-normalised DCT features, diff images of the raw frames.  Stream order of adenet_v3: raw, dct, diff."""
    X, lens, ys = _presplit_cuave(cfg)
    d = _cuave_dct(cfg)
    return {k: [X[k], d[k], compute_diff_images(X[k], lens[k])] for k in X}, ys, lens


def _load_cuave_av(cfg):
    """cuave/audio_visual_runner.py:295-312: the visual frames re-ordered to C order (nothing else), the audio features of
    the second pre-split file as stored."""
    X, lens, ys = _presplit_cuave(cfg)
    A, _, _ = _presplit_cuave(cfg, 'audio')
    imagesize = tuple(int(v) for v in cfg.get('data', 'imagesize', str, '30,50').split(','))
    return {k: [reorder_data(X[k], imagesize), A[k]] for k in X}, ys, lens


def _subject_ids(cfg):
    return [read_data_split_file(cfg.get('training', k + '_subjects_file', str, 'data/%s.txt' % k)) for k in ('train', 'val', 'test')]


def _load_cuave_subject_unimodal(cfg):
    """cuave/unimodal_with_val.py:201-244: per-sequence mean removal of the WHOLE set, subject split, targets + 1,
    re-ordering to C order, a second per-sequence mean removal per split, per-frame z-normalisation."""
    data = load_mat_file(cfg.get('data', 'images'))
    X = data['dataMatrix'].astype('float32')
    y, subjects, lens_all = _vec(data, 'targetsVec'), _vec(data, 'subjectsVec'), _vec(data, 'videoLengthVec')
    X = sequencewise_mean_image_subtraction(X, lens_all)
    ids = _subject_ids(cfg)
    parts = split_seq_data(X, y, subjects, lens_all, ids[0], ids[1], ids[2])
    imagesize = tuple(int(v) for v in cfg.get('data', 'imagesize', str, '30,50').split(','))
    split, ys, lens = {}, {}, {}
    for k, o in (('train', 0), ('val', 4), ('test', 8)):
        lens[k] = np.asarray(parts[o + 2], int)
        ys[k] = np.asarray(parts[o + 1]) + 1
        Xk = reorder_data(parts[o], imagesize)
        Xk = sequencewise_mean_image_subtraction(Xk, lens[k])
        split[k] = [normalize_input(Xk, centralize=True)]
    return split, ys, lens


def _load_oulu(cfg, with_dct):
    """oulu/unimodal_with_val.py:263-296 / oulu/bimodal_with_val.py:270-316: subject split, per-frame z-normalisation of
    the frames; with_dct: the DCT features ride along, feature-wise normalised with the TRAIN statistics."""
    data = load_mat_file(cfg.get('data', 'images'))
    X = data['dataMatrix'].astype('float32')
    y, subjects, lens_all = _vec(data, 'targetsVec', 'int32'), _vec(data, 'subjectsVec'), _vec(data, 'videoLengthVec')
    ids = _subject_ids(cfg)
    mats = [X] + ([load_mat_file(cfg.get('data', 'dct'))['dctFeatures'].astype('float32')] if with_dct else [])
    split = dict(train=[], val=[], test=[])
    for mat in mats:
        parts = split_seq_data(mat, y, subjects, lens_all, ids[0], ids[1], ids[2])
        split['train'].append(parts[0]); split['val'].append(parts[4]); split['test'].append(parts[8])
        ys = dict(train=parts[1], val=parts[5], test=parts[9])
        lens = dict(train=parts[2], val=parts[6], test=parts[10])
    for k in split:
        split[k][0] = normalize_input(split[k][0], centralize=True)
    if with_dct:
        tr, mean, std = featurewise_normalize_sequence(split['train'][1])
        split['train'][1] = tr
        for k in ('val', 'test'):
            split[k][1] = (split[k][1] - mean) / std
    return split, ys, lens


def _load_avletters_diff(cfg):
    """avletters/bimodal_diff_image.py:269-304: raw and diff-image matrices as stored (every normalisation is commented
    out in the script), targets as stored, iterVec split; the test split doubles as the validation split."""
    data = load_mat_file(cfg.get('data', 'images'))
    mats = [data['dataMatrix'].astype('float32'), load_mat_file(cfg.get('data', 'diff'))['dataMatrix'].astype('float32')]
    targets_vec, vid_len_vec, iter_vec = _vec(data, 'targetsVec'), _vec(data, 'videoLengthVec'), _vec(data, 'iterVec')
    indexes = create_split_index(len(mats[0]), vid_len_vec, iter_vec)
    train_lens, test_lens = split_videolen(vid_len_vec, iter_vec)
    assert np.sum(vid_len_vec) == len(mats[0])
    split = dict(train=[m[indexes] for m in mats], test=[m[~indexes] for m in mats])
    split['val'] = split['test']
    ys = dict(train=targets_vec[indexes], test=targets_vec[~indexes])
    ys['val'] = ys['test']
    lens = dict(train=np.asarray(train_lens, int), test=np.asarray(test_lens, int))
    lens['val'] = lens['test']
    return split, ys, lens


def _load_avletters_stream1(config):
    """avletters/unimodal.py:178-204: the schema-1 ``[stream1]`` section's switches (reorderdata / meanremove / diffimage /
    samplewisenormalize ahead of the split, featurewisenormalize behind it), iterVec split, matlab_target_offset."""
    from .nstream import presplit_dataprocessing
    data = load_mat_file(config.get('stream1', 'data'))
    imagesize = tuple(int(d) for d in config.get('stream1', 'imagesize').split(','))
    X = data['dataMatrix'].astype('float32')
    targets_vec, vid_len_vec, iter_vec = _vec(data, 'targetsVec'), _vec(data, 'videoLengthVec'), _vec(data, 'iterVec')
    X = presplit_dataprocessing(X, vid_len_vec, config, 'stream1', imagesize=imagesize)
    indexes = create_split_index(len(X), vid_len_vec, iter_vec)
    train_lens, test_lens = split_videolen(vid_len_vec, iter_vec)
    if config.getboolean('lstm_classifier', 'matlab_target_offset'):
        targets_vec = targets_vec - 1
    tr, te = X[indexes], X[~indexes]
    if config.getboolean('stream1', 'featurewisenormalize'):
        tr, mean, std = featurewise_normalize_sequence(tr)
        te = (te - mean) / std
    split = dict(train=[tr], test=[te])
    split['val'] = split['test']
    ys = dict(train=targets_vec[indexes], test=targets_vec[~indexes])
    ys['val'] = ys['test']
    lens = dict(train=np.asarray(train_lens, int), test=np.asarray(test_lens, int))
    lens['val'] = lens['test']
    return split, ys, lens


# --------------------------------------------------------------------------------------------------------- per-script plans
_DIGITS = '0,1,2,3,4,5,6,7,8,9'
_PHRASES = 'p1,p2,p3,p4,p5,p6,p7,p8,p9,p10'
_LETTERS = ','.join('abcdefghijklmnopqrstuvwxyz')
_MS = (None, None)


class Plan(object):
    """Everything one reference script decides ahead of its epoch loop.  Defaults = the most common choice."""
    head = 'frames'            # per-frame softmax + temporal loss + majority vote | 'last': SliceLayer(-1) + cross-entropy
    has_test = True            # a held-out test split besides the validation split (the AVLetters scripts have none)
    stop = 'early_stop2'       # utils/regularization.py:14-23 | 'early_stop' (strictly increasing window, :1-11)
    decay = False              # lr *= decay_rate after every epoch from decay_start on
    t1_rule = False            # avletters/bimodal*.py: no improvement and epoch >= t1 -> lr = max(lr * decay, 0.001), next momentum
    rule = 'adam'
    l_fuse = None
    fusiontype = None
    window = 9                 # WINDOW_SIZE of the older scripts
    conf_fmt = 'latex'
    plot_name = 'valid_cost'
    final = 'CR: {best_cr}, val loss: {best_val}, Test CR: {test_cr}'
    progress = 'Epoch {e} batch {i}/{n}: {b} examples using adam at learning rate = {lr:.4f}'
    results = None             # callable(f, st) writing the --write_results lines

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _hyper(cfg, plan, num_epoch=None, epochsize=None, batchsize=None, validation_window=None):
    """Loop sizes: the .ini / CLI value, else the constant the reference script hard-codes."""
    plan.num_epoch = cfg.get('training', ('num_epoch', 'no_epochs'), int, num_epoch)
    plan.epochsize = cfg.get('training', 'epochsize', int, epochsize)
    plan.batchsize = cfg.get('training', 'batchsize', int, batchsize)
    plan.validation_window = cfg.get('training', 'validation_window', int, validation_window)


def _series(f, st):
    for series in (st['cost_train'], st['cost_val'], st['class_rate']):
        f.write('{}\n'.format(','.join(str(v) for v in series)))


def _adam_plan(cfg, network, learning_rate, **kw):
    plan = Plan(network=network, rule='adam', **kw)
    plan.update = Updater(network, 'adam', learning_rate)     # (carries lr for the progress line)
    plan.train = lambda ins, y, m, w: network.train_step(ins, y, m, w, learning_rate, want_loss=False)   # adam(cost, params, learning_rate)
    return plan


def _plan_cuave_bimodal(cfg, config, options):
    split, ys, lens = _load_cuave(cfg)
    weight_init = cfg.get('training', 'weight_init')
    use_peepholes = cfg.get('training', 'use_peepholes', bool)
    nonlinearity = select_nonlinearity(cfg.get('models', 'nonlinearity'))
    use_blstm, use_finetuning = cfg.get('training', 'use_blstm', bool), cfg.get('training', 'use_finetuning', bool)
    learning_rate = cfg.get('training', 'learning_rate', float)
    fusiontype = cfg.get('models', 'fusiontype')
    ae = load_dbn(cfg.get('models', 'pretrained'))
    network, l_fuse = adenet_v2.create_model(ae, (None, None, cfg.get('models', 'input_dimension', int, 1500)), None, _MS, None,
                                             (None, None, cfg.get('models', 'no_coeff', int, 30) * 3), None,
                                             cfg.get('models', 'lstm_size', int, 250), None,
                                             cfg.get('models', 'output_classes', int, 10), fusiontype,
                                             las_init.select(weight_init), use_peepholes, nonlinearity)
    plan = _adam_plan(cfg, network, learning_rate, split=split, ys=ys, lens=lens, l_fuse=l_fuse, fusiontype=fusiontype,
                      names=_DIGITS, progress='Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f}')
    _hyper(cfg, plan)

    def results(f, st):                              # cuave/bimodal_with_val.py:376-381
        f.write('{},{},{},{},{},{},{},{},{},{},{},{}\n'.format(use_finetuning, 'yes', use_peepholes, 'adam', weight_init, 'RELU',
                                                               use_blstm, learning_rate, st['best_tr'], st['best_val'],
                                                               st['best_cr'] * 100, st['test_cr'] * 100))
        _series(f, st)
    plan.results = results
    return plan


def _encoder_of(cfg, nonlinearity):
    weights, biases, _, _ = load_dbn(cfg.get('models', 'pretrained'))
    return weights, biases


def _plan_unimodal_encoder(cfg, config, options, dataset):
    """cuave/unimodal_with_val.py, oulu/unimodal_with_val.py: ONE encoder stream, summed BLSTM, per-frame softmax.  Both
    scripts call ``deltanet_majority_vote.create_model_using_pretrained_encoder`` -- a function the reference module does
    not define (only modelzoo/deltanet.py:12 has one of that name, with the last-timestep head, while the scripts train
    with temporal_softmax_loss and vote per frame): the per-frame model with that argument list is what they mean."""
    cuave = dataset == 'cuave'
    split, ys, lens = _load_cuave_subject_unimodal(cfg) if cuave else _load_oulu(cfg, with_dct=False)
    nonlinearity = select_nonlinearity(cfg.get('models', 'nonlinearity'))
    weight_init = cfg.get('training' if cuave else 'models', 'weight_init')
    use_peepholes = cfg.get('training', 'use_peepholes', bool)
    learning_rate = cfg.get('training', 'learning_rate', float)
    weights, biases = _encoder_of(cfg, nonlinearity)
    network = deltanet_majority_vote.create_model_using_pretrained_encoder(
        weights, biases, (None, None, cfg.get('models', 'input_dimension', int, 1500 if cuave else 1144)), None, _MS, None,
        cfg.get('models', ('lstm_size', 'lstm_units'), int), None, cfg.get('models', 'output_classes', int),
        las_init.select(weight_init), use_peepholes, nonlinearity)
    plan = _adam_plan(cfg, network, learning_rate, split=split, ys=ys, lens=lens, names=_DIGITS if cuave else _PHRASES,
                      conf_fmt='grid', plot_name=None,
                      final='test CR: {test_cr}, val CR: {best_cr}, val loss: {best_val}' if cuave
                      else 'classification rate: {test_cr}, validation loss: {best_val}')
    plan.window = 9 if cuave else cfg.get('models', 'delta_window', int)
    _hyper(cfg, plan)
    if cuave:
        plan.results = lambda f, st: f.write('{},{},{}\n'.format(st['test_cr'], st['best_cr'], st['best_val']))
        plan.save_best = options.get('save_best')
    return plan


def _plan_cuave_unimodal_dct(cfg, config, options):
    split, ys, lens = _load_cuave_dct(cfg)
    weight_init = cfg.get('training', 'weight_init')
    use_peepholes = cfg.get('training', 'use_peepholes', bool)
    use_blstm, use_finetuning = cfg.get('training', 'use_blstm', bool), cfg.get('training', 'use_finetuning', bool)
    learning_rate = cfg.get('training', 'learning_rate', float)       # read and reported; adam runs with its defaults (:217)
    network = lstm_classifier_majority_vote.create_model((None, None, cfg.get('models', 'no_coeff', int) * 3), None, _MS, None,
                                                         cfg.get('models', 'lstm_size', int), cfg.get('models', 'output_classes', int),
                                                         w_init=las_init.select(weight_init))
    plan = Plan(network=network, rule='adam', split=split, ys=ys, lens=lens, names=_DIGITS,
                progress='Epoch {e} batch {i}/{n}: {b} examples using adam')
    plan.update = Updater(network, 'adam', 1e-3)
    plan.train = plan.update
    _hyper(cfg, plan)

    def results(f, st):                              # cuave/unimodal_dct_with_val.py:313-325
        f.write('{},{},{},{},{},{},{},{},{},{},{},{}\n'.format(use_finetuning, 'yes', use_peepholes, 'adam', weight_init, 'N/A',
                                                               use_blstm, learning_rate, st['best_tr'], st['best_val'],
                                                               st['best_cr'] * 100, st['test_cr'] * 100))
        _series(f, st)
    plan.results = results
    return plan


def _plan_trimodal(cfg, config, options, dataset, script):
    """adenet_v3 scripts: oulu/trimodal_with_val.py, avletters/trimodal.py, cuave/trimodal_with_val.py."""
    for key in ('do_finetune', 'save_finetune'):
        if cfg.get('training', key, bool, False):
            raise NotImplementedError('%s: nolearn auto-encoder fine-tuning is outside this package (SURVEY 8: out of scope); '
                                      'fine-tune offline and point `finetuned` to the weights' % key)
    if dataset == 'oulu':
        split, ys, lens = _load_subject_split(cfg)
        consts = dict(num_epoch=12, epochsize=120, batchsize=10, validation_window=4, dim=1144, classes=10, names=_PHRASES)
    elif dataset == 'cuave':
        split, ys, lens = _load_cuave_trimodal(cfg)
        consts = dict(num_epoch=30, epochsize=45, batchsize=20, validation_window=4, dim=1500, classes=10, names=_DIGITS)
    else:
        split, ys, lens = _load_avletters(cfg, with_diff=True, normalise_images=False, target_offset=False)
        consts = dict(num_epoch=25, epochsize=20, batchsize=26, validation_window=4, dim=1200, classes=26, names=_LETTERS)
    fusiontype = cfg.get('models', 'fusiontype')
    learning_rate = cfg.get('training', 'learning_rate', float)
    if not cfg.get('training', 'load_finetune', bool, True) or not cfg.get('training', 'load_finetune_diff', bool, True):
        raise ValueError('load_finetune / load_finetune_diff must be true: the scripts define `ae` / `diff_ae` nowhere else')
    ae = load_ae(cfg.get('models', 'finetuned'))
    ae_diff = load_ae(cfg.get('models', 'finetuned_diff'))
    network, l_fuse = adenet_v3.create_model(ae, ae_diff, (None, None, cfg.get('models', 'input_dimension', int, consts['dim'])), None,
                                             _MS, None, (None, None, split['train'][1].shape[1]), None,
                                             (None, None, split['train'][2].shape[1]), None,
                                             cfg.get('models', 'lstm_size', int, 250), None,
                                             cfg.get('models', 'output_classes', int, consts['classes']), fusiontype)
    plan = Plan(network=network, l_fuse=l_fuse, fusiontype=fusiontype, rule='adadelta', head='last', stop='early_stop', decay=True,
                split=split, ys=ys, lens=lens, names=consts['names'], has_test=dataset != 'avletters',
                progress='Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f}')
    plan.update = Updater(network, 'adadelta', learning_rate)
    plan.train = plan.update
    plan.decay_rate, plan.decay_start = cfg.get('training', 'decay_rate', float), cfg.get('training', 'decay_start', int)
    _hyper(cfg, plan, consts['num_epoch'], consts['epochsize'], consts['batchsize'], consts['validation_window'])
    if dataset == 'avletters':
        plan.final, plan.plot_name = 'classification rate: {best_cr}, validation loss: {best_val}', 'e2e_valid_cost'
        plan.results = lambda f, st: f.write('{},{},{}\n'.format(fusiontype, st['best_cr'], st['best_val']))
    elif dataset == 'cuave':                         # cuave/trimodal_with_val.py:374-377
        plan.results = lambda f, st: f.write('{},{},{}\n'.format(fusiontype, st['test_cr'], st['best_val']))
    else:
        plan.results = lambda f, st: f.write('{},{},{}\n'.format(fusiontype, st['best_cr'], st['best_val']))
    return plan


def _plan_oulu_bimodal(cfg, config, options):
    split, ys, lens = _load_oulu(cfg, with_dct=True)
    fusiontype = cfg.get('models', 'fusiontype')
    learning_rate = cfg.get('training', 'learning_rate', float)
    dbn = load_dbn(cfg.get('models', 'pretrained'))
    network, l_fuse = adenet_v2.create_model(dbn, (None, None, cfg.get('models', ('input_dimensions', 'input_dimension'), int)), None,
                                             _MS, None, (None, None, cfg.get('models', ('no_coeffs', 'no_coeff'), int)), None,
                                             cfg.get('models', 'lstm_size', int), None, cfg.get('models', 'output_classes', int),
                                             fusiontype, w_init_fn=las_init.select(cfg.get('training', 'weight_init')),
                                             use_peepholes=cfg.get('models', 'use_peepholes', bool))
    plan = _adam_plan(cfg, network, learning_rate, split=split, ys=ys, lens=lens, l_fuse=l_fuse, fusiontype=fusiontype,
                      names=_PHRASES)
    plan.window = cfg.get('models', 'delta_window', int)
    _hyper(cfg, plan)
    if not cfg.get('training', 'honour_ini_sizes', bool, False):
        plan.epochsize, plan.batchsize = 120, 10     # oulu/bimodal_with_val.py:366-367 overwrites what it read from the .ini
    plan.results = lambda f, st: f.write('{},{},{}\n'.format(fusiontype, st['test_cr'], st['best_val']))
    return plan


def _rule_plan(cfg, plan, momentum_default=0.9):
    """``update_rule`` switch + both decay rules of avletters/bimodal.py:446-455,541-555 / bimodal_diff_image.py."""
    rule = cfg.get('training', 'update_rule')
    plan.rule = rule
    plan.decay_rate, plan.decay_start = cfg.get('training', 'decay_rate', float), cfg.get('training', 'decay_start', int)
    plan.t1 = cfg.get('training', 't1', int)
    plan.momentum, plan.mm_schedule = momentum_default, []
    if rule in ('sgdm', 'sgdnm'):
        plan.momentum = cfg.get('training', 'momentum', float)
        plan.mm_schedule = [float(m) for m in cfg.get('training', 'momentum_schedule').split(',')]
    plan.update = Updater(plan.network, rule, cfg.get('training', 'learning_rate', float), plan.momentum)
    plan.train = plan.update
    plan.decay, plan.t1_rule = True, True
    if rule == 'adam':
        plan.progress = 'Epoch {e} batch {i}/{n}: {b} examples with {rule} using default params'
    elif rule in ('sgdm', 'sgdnm'):
        plan.progress = 'Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f}, momentum = {mm:.4f} with {rule}'
    else:
        plan.progress = plan.adadelta_progress


def _plan_avletters_bimodal(cfg, config, options):
    split, ys, lens = _load_avletters(cfg, with_diff=False, normalise_images=True, target_offset=True)
    fusiontype = cfg.get('models', 'fusiontype')
    weight_init = cfg.get('training', 'weight_init')
    use_peepholes = cfg.get('training', 'use_peepholes', bool)
    use_blstm, use_finetuning = cfg.get('training', 'use_blstm', bool), cfg.get('training', 'use_finetuning', bool)
    dbn = load_ae(cfg.get('models', 'pretrained'))
    if not isinstance(dbn, (tuple, list)) or len(dbn) == 2:
        from ..modelzoo._factory import nolearn_weights
        dbn = nolearn_weights(dbn, nonlinearities=('rectify', 'rectify', 'rectify', 'linear'))
    factory = adenet_v2 if use_blstm else adenet_v2_3
    network, l_fuse = factory.create_model(dbn, (None, None, cfg.get('models', 'input_dimension', int, 1200)), None, _MS, None,
                                           (None, None, cfg.get('models', 'no_coeff', int, 30) * 3), None,
                                           cfg.get('models', 'lstm_size', int, 250), None,
                                           cfg.get('models', 'output_classes', int, 26), fusiontype,
                                           w_init_fn=las_init.select(weight_init), use_peepholes=use_peepholes)
    plan = Plan(network=network, l_fuse=l_fuse, fusiontype=fusiontype, split=split, ys=ys, lens=lens, names=_LETTERS,
                has_test=False, conf_fmt='pipe', plot_name='e2e_valid_cost', adasum_at_end=True,
                final='classification rate: {best_cr}, validation loss: {best_val}',
                adadelta_progress='Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f}')
    _rule_plan(cfg, plan)                            # t1 rule AND the per-epoch decay (avletters/bimodal.py:541-555)
    _hyper(cfg, plan, None, 20, 26, None)

    def results(f, st):                              # avletters/bimodal.py:592-606
        f.write('{},{},{},{},{}\n'.format(plan.validation_window, weight_init, use_peepholes, use_blstm, use_finetuning))
        _series(f, st)
        f.write('{},{},{}\n'.format(fusiontype, st['best_cr'], st['best_val']))
    plan.results = results
    return plan


def _plan_avletters_bimodal_diff(cfg, config, options):
    for key in ('do_finetune', 'save_finetune'):
        if cfg.get('training', key, bool, False):
            raise NotImplementedError('%s: nolearn auto-encoder fine-tuning is outside this package (SURVEY 8: out of scope)' % key)
    split, ys, lens = _load_avletters_diff(cfg)
    fusiontype = cfg.get('models', 'fusiontype')
    model = cfg.get('models', 'model')
    if model != 'adenet_v2_1':
        raise ValueError("avletters/bimodal_diff_image.py:344 builds a network only for model = adenet_v2_1 (got %r)" % model)
    weight_init = cfg.get('training', 'weight_init')
    # parse_options() of the script sets use_peepholes=True as the parser default and copies it into `options` whenever it is
    # truthy (:176,207-208): the .ini's value never wins
    use_peepholes = True
    if not cfg.get('training', 'load_finetune', bool, True) or not cfg.get('training', 'load_finetune_diff', bool, True):
        raise ValueError('load_finetune / load_finetune_diff must be true: the script defines `ae` / `diff_ae` nowhere else')
    ae, diff_ae = load_ae(cfg.get('models', 'finetuned')), load_ae(cfg.get('models', 'finetuned_diff'))
    network, l_fuse = adenet_v2_1.create_model(ae, diff_ae, (None, None, cfg.get('models', 'input_dimension', int, 1200)), None, _MS,
                                               None, (None, None, split['train'][1].shape[1]), None,
                                               cfg.get('models', 'lstm_size', int, 250), None,
                                               cfg.get('models', 'output_classes', int, 26), fusiontype,
                                               w_init_fn=las_init.select(weight_init), use_peepholes=use_peepholes)
    plan = Plan(network=network, l_fuse=l_fuse, fusiontype=fusiontype, head='last', split=split, ys=ys, lens=lens, names=_LETTERS,
                has_test=False, plot_name='e2e_valid_cost', adasum_at_end=True,
                final='classification rate: {best_cr}, validation loss: {best_val}',
                adadelta_progress='Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f} with {rule}')
    _rule_plan(cfg, plan)
    _hyper(cfg, plan, None, 20, 26, None)
    learning_rate = cfg.get('training', 'learning_rate', float)

    def results(f, st):                              # avletters/bimodal_diff_image.py:481-497
        f.write('{},{},{},{},{},{},{},{},{}\n'.format(plan.rule, learning_rate, plan.decay_rate, plan.momentum, plan.decay_start,
                                                      plan.t1, plan.validation_window, weight_init, use_peepholes))
        _series(f, st)
        f.write('{},{},{}\n'.format(fusiontype, st['best_cr'], st['best_val']))
    plan.results = results
    return plan


def _plan_avletters_unimodal(cfg, config, options):
    """avletters/unimodal.py: the one schema-1 file in this family ([stream1] / [lstm_classifier] / [training])."""
    from .nstream import load_decoder
    split, ys, lens = _load_avletters_stream1(config)
    lc = 'lstm_classifier'
    has_encoder = config.getboolean('stream1', 'has_encoder')
    dim = config.getint('stream1', 'input_dimensions')
    w_init = las_init.select(options.get('weight_init', config.get(lc, 'weight_init')))
    use_peepholes = options.get('use_peepholes', config.getboolean(lc, 'use_peepholes'))
    use_blstm = config.has_option(lc, 'use_blstm')              # presence test, avletters/unimodal.py:155
    H, C = config.getint(lc, 'lstm_size'), config.getint(lc, 'output_classes')
    learning_rate = float(options.get('learning_rate', config.getfloat('training', 'learning_rate')))
    if has_encoder:
        ae1 = load_decoder(config.get('stream1', 'model'), config.get('stream1', 'shape'), config.get('stream1', 'nonlinearities'))
        network = deltanet_majority_vote.create_model(ae1, (None, None, dim), None, _MS, None, H, None, C, w_init, use_peepholes)
    else:
        network = deltanet_v1.create_model((None, None, dim), None, _MS, None, None, H, C, w_init, use_peepholes, use_blstm)
    plan = _adam_plan(cfg, network, learning_rate, split=split, ys=ys, lens=lens, names=config.get(lc, 'output_classnames'),
                      has_test=False, plot_name=None, final='classification rate: {best_cr}, validation loss: {best_val}',
                      progress='Epoch {e} batch {i}/{n}: {b} examples at learning rate = {lr:.4f}')
    plan.window = config.getint(lc, 'windowsize')
    plan.num_epoch = int(options.get('num_epoch', config.getint('training', 'num_epoch')))
    plan.validation_window = int(options.get('validation_window', config.getint('training', 'validation_window')))
    plan.epochsize, plan.batchsize = config.getint('training', 'epochsize'), config.getint('training', 'batchsize')
    return plan


def _plan_cuave_avnet(cfg, config, options):
    """cuave/audio_visual_runner.py:242-472."""
    split, ys, lens = _load_cuave_av(cfg)
    fusiontype = cfg.get('models', 'fusiontype')
    lstm_size, output_classes = cfg.get('models', 'lstm_size', int), cfg.get('models', 'output_classes', int)
    nonlinearity = select_nonlinearity(cfg.get('models', 'nonlinearity'))
    weight_init = cfg.get('training', 'weight_init')
    w_init = las_init.select(weight_init)
    use_peepholes = cfg.get('training', 'use_peepholes', bool)
    use_blstm, use_finetuning = cfg.get('training', 'use_blstm', bool), cfg.get('training', 'use_finetuning', bool)
    learning_rate = cfg.get('training', 'learning_rate', float)
    vw, vb, _, _ = load_dbn(cfg.get('models', 'pretrained'))
    aw, ab, _, _ = load_dbn(cfg.get('models', 'pretrained_diff'))
    visual_net = avnet.create_pretrained_substream(vw, vb, (None, None, cfg.get('models', 'input_dimension', int)), None, _MS, None,
                                                   'visual', lstm_size, None, nonlinearity, w_init, use_peepholes)
    audio_net = avnet.create_pretrained_substream(aw, ab, (None, None, cfg.get('models', 'input_dimension2', int)), None, _MS, None,
                                                  'audio', lstm_size, None, nonlinearity, w_init, use_peepholes)
    network, l_fuse = avnet.create_model([visual_net, audio_net], _MS, None, lstm_size, output_classes, fusiontype, w_init,
                                         use_peepholes)
    plan = _adam_plan(cfg, network, learning_rate, split=split, ys=ys, lens=lens, l_fuse=l_fuse, fusiontype=fusiontype,
                      names=_DIGITS, progress='Epoch {e} batch {i}/{n}: {b} examples using adam with learning rate {lr:.4f}')
    _hyper(cfg, plan, None, 90, 10, None)

    def results(f, st):                              # cuave/audio_visual_runner.py:456-472
        f.write('{},{},{},{},{},{},{},{},{},{},{},{}\n'.format(use_finetuning, 'yes', use_peepholes, 'adam', weight_init, 'RELU',
                                                               use_blstm, learning_rate, st['best_tr'], st['best_val'],
                                                               st['best_cr'] * 100, st['test_cr'] * 100))
        _series(f, st)
    plan.results = results
    return plan


_PLANS = {('cuave', 'bimodal_with_val'): _plan_cuave_bimodal,
          ('cuave', 'unimodal_with_val'): lambda c, k, o: _plan_unimodal_encoder(c, k, o, 'cuave'),
          ('cuave', 'unimodal_dct_with_val'): _plan_cuave_unimodal_dct,
          ('cuave', 'trimodal_with_val'): lambda c, k, o: _plan_trimodal(c, k, o, 'cuave', 'trimodal_with_val'),
          ('cuave', 'audio_visual_runner'): _plan_cuave_avnet,
          ('oulu', 'trimodal_with_val'): lambda c, k, o: _plan_trimodal(c, k, o, 'oulu', 'trimodal_with_val'),
          ('oulu', 'unimodal_with_val'): lambda c, k, o: _plan_unimodal_encoder(c, k, o, 'oulu'),
          ('oulu', 'bimodal_with_val'): _plan_oulu_bimodal,
          ('avletters', 'trimodal'): lambda c, k, o: _plan_trimodal(c, k, o, 'avletters', 'trimodal'),
          ('avletters', 'bimodal'): _plan_avletters_bimodal,
          ('avletters', 'bimodal_diff_image'): _plan_avletters_bimodal_diff,
          ('avletters', 'unimodal'): _plan_avletters_unimodal}


# --------------------------------------------------------------------------------------------------------- driver
def _main(dataset, script, argv=None):
    if (dataset, script) not in SCRIPTS:
        raise ValueError('no driver for %s/%s.py (have: %s)' % (dataset, script, sorted(SCRIPTS)))
    options = parse_options(argv, SCRIPTS[(dataset, script)])
    if options['seed'] is not None:
        np.random.seed(options['seed'])
        las_init.set_rng(np.random.RandomState(options['seed']))
    config = configparser.ConfigParser()
    if not config.read(options['config']):
        raise IOError('cannot read config file %s' % options['config'])
    cfg = _Cfg(config, options)
    print('CLI options: {}'.format(list(options.items())))
    print('Reading Config File: {}...'.format(options['config']))
    for sec in (('stream1', 'lstm_classifier', 'training') if config.has_section('stream1') else ('data', 'models', 'training')):
        print(config.items(sec))
    print('preprocessing dataset...')
    from ..modelzoo import _factory
    _factory.set_default_precision(options.get('precision') or os.environ.get('ADN_PRECISION', _factory.PRODUCT_DEFAULT_PRECISION))
    plan = _PLANS[(dataset, script)](cfg, config, options)
    network, update, train = plan.network, plan.update, plan.train
    split, ys, lens = plan.split, plan.ys, plan.lens
    n_streams = len(split['train'])
    frames_head, has_test, rule, fusiontype = plan.head == 'frames', plan.has_test, plan.rule, plan.fusiontype
    num_epoch, epochsize, batchsize, validation_window = plan.num_epoch, plan.epochsize, plan.batchsize, plan.validation_window
    WINDOW_SIZE, STRIP_SIZE = plan.window, 3
    classnames = cfg.get('models', 'output_classnames', str, plan.names).split(',') if config.has_section('models') \
        else plan.names.split(',')
    print('constructing end to end model...')
    print_network(network)
    print('compiling model...')
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
    best_params = None
    tr_lens = np.asarray(lens['train'], int)
    # Minibatches are assembled on the GPU from splits resident in HBM (utils/datagen_gpu.py: same utterance order, padding and
    # label conventions as gen_lstm_batch_random / gen_seq_batch_from_idx); ADN_HOST_BATCHES=1 selects the reference's host-side
    # assembly with an upload per batch (A/B runs, the equality test between the two)
    host_batches = bool(os.environ.get('ADN_HOST_BATCHES'))

    def targets_of(y, m):                            # per-frame targets for the temporal loss; the last-step head takes (B,)
        y = np.asarray(y).reshape((-1, 1)).repeat(m.shape[-1], axis=-1)
        return y

    if host_batches:
        tmax_train = int(np.max(tr_lens))
        datagen = gen_lstm_batch_random(split['train'][0], ys['train'], tr_lens, batchsize=batchsize)
        integral_lens = compute_integral_len(tr_lens)

        def whole_split(k):
            ln = np.asarray(lens[k], int)
            X1, y, m, idxs = next(gen_lstm_batch_random(split[k][0], ys[k], ln, batchsize=len(ln)))
            il = compute_integral_len(ln)
            return ([X1] + [gen_seq_batch_from_idx(split[k][s], idxs, ln, il, np.max(ln)) for s in range(1, n_streams)], y, m,
                    targets_of(y, m))
    else:
        from ..utils.datagen_gpu import DeviceSplit
        from .nstream import resident_dtype
        dtype = resident_dtype(network)
        resident = {}
        for k in ('train', 'val') + (('test',) if has_test else ()):
            same = [j for j in resident if split[j] is split[k] and lens[j] is lens[k]]     # (avletters: 'val' IS the test split)
            resident[k] = resident[same[0]] if same else DeviceSplit(split[k], ys[k], np.asarray(lens[k], int), dtype=dtype)
        datagen = resident['train'].batches(batchsize, prefetch=bool(os.environ.get('ADN_PREFETCH')))

        def whole_split(k):
            b = resident[k].whole()
            return b.Xs, b.y, b.mask, b.targets

    X_val, y_val_evaluate, mask_val, y_val = whole_split('val')
    if has_test:
        X_test, y_test, mask_test, _ = whole_split('test')
    stop = (lambda w, best: early_stop(w)) if plan.stop == 'early_stop' else (lambda w, best: early_stop2(w, best, validation_window))

    for epoch in range(num_epoch):
        time_start = time.time()
        for i in range(epochsize):
            if host_batches:
                X1, y, m, batch_idxs = next(datagen)
                yy = targets_of(y, m)
                Xs = [X1] + [gen_seq_batch_from_idx(split['train'][s], batch_idxs, tr_lens, integral_lens, tmax_train)
                             for s in range(1, n_streams)]
            else:
                batch = next(datagen)
                Xs, yy, m = batch.Xs, batch.targets, batch.mask
            print(plan.progress.format(e=epoch + 1, i=i + 1, n=epochsize, b=len(m), lr=update.lr, mm=update.mm, rule=rule), end='')
            sys.stdout.flush()
            train(Xs, yy, m, WINDOW_SIZE)
            print('\r', end='')
        cost = float(compute_train_cost(Xs, yy, m, WINDOW_SIZE))
        # (validation cost and the predictions the vote / arg-max is taken on: one forward pass instead of the scripts' two --
        #  compute_test_cost, then val_fn, the same deterministic graph on the same inputs)
        val_probs = None
        if not os.environ.get('ADN_TWO_PASS_EVAL'):
            val_cost, val_probs = network.loss_and_probs(X_val, y_val, mask_val, WINDOW_SIZE)
            val_cost = float(val_cost)
        else:
            val_cost = float(compute_test_cost(X_val, y_val, mask_val, WINDOW_SIZE))
        cost_train.append(cost)
        cost_val.append(val_cost)
        train_strip[epoch % STRIP_SIZE] = cost
        val_window.push(val_cost)
        gl = 100 * (cost_val[-1] / np.min(cost_val) - 1)
        with np.errstate(divide='ignore', invalid='ignore'):
            pk = 1000 * (np.sum(train_strip) / (STRIP_SIZE * np.min(train_strip)) - 1)
            pq = gl / pk
        cr, val_conf = evaluate(X_val, y_val_evaluate, mask_val, WINDOW_SIZE, eval_fn if val_probs is None else (lambda *a: val_probs))
        class_rate.append(cr)
        improved = val_cost < best_val
        if has_test:
            if improved:
                best_val, best_tr, best_conf, best_cr = val_cost, cost, val_conf, cr
                if fusiontype == 'adasum':
                    adascale_param = plan.l_fuse.get_all_param_values(scaling_param=True)
                test_cr, test_conf = evaluate(X_test, y_test, mask_test, WINDOW_SIZE, eval_fn)
                print("Epoch {} train cost = {}, val cost = {}, GL loss = {:.3f}, GQ = {:.3f}, CR = {:.3f}, Test CR= {:.3f} "
                      "({:.1f}sec)".format(epoch + 1, cost_train[-1], cost_val[-1], gl, pq, cr, test_cr, time.time() - time_start))
                if getattr(plan, 'save_best', None):
                    best_params = network.snapshot_params()      # (a device copy; fetched only when the model is saved)
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
                    adascale_param = plan.l_fuse.get_all_param_values(scaling_param=True)
            elif plan.t1_rule and epoch >= plan.t1 and rule in ('sgdm', 'sgdnm'):     # avletters/bimodal.py:541-545
                update.lr = max(update.lr * plan.decay_rate, 0.001)
                if plan.mm_schedule:
                    update.mm = plan.mm_schedule.pop(0)
        if epoch >= validation_window and stop(val_window, best_val):
            break
        if plan.decay and epoch + 1 >= plan.decay_start:         # learning rate decay (oulu/trimodal_with_val.py:508-510)
            update.lr = float(np.float32(update.lr) * np.float32(plan.decay_rate))

    # ---- report
    print('Final Model' if has_test else 'Best Model')
    print(plan.final.format(best_cr=best_cr, best_val=best_val, test_cr=test_cr))
    if fusiontype == 'adasum':
        if getattr(plan, 'adasum_at_end', False):
            adascale_param = plan.l_fuse.get_all_param_values(scaling_param=True)
        print("final scaling params: {}".format(adascale_param))
    print('confusion matrix: ')
    conf = test_conf if has_test else best_conf
    if not options['no_plot'] and conf is not None:
        print(plot_confusion_matrix(conf, classnames[:conf.shape[0]], fmt=plan.conf_fmt))
        try:
            plot_validation_cost(cost_train, cost_val, class_rate if plan.plot_name is None else None, savefilename=plan.plot_name)
        except Exception as e:                        # matplotlib is optional here
            print('(no plot: %s)' % e)
    st = dict(best_cr=best_cr, best_val=best_val, best_tr=best_tr, test_cr=test_cr, cost_train=cost_train, cost_val=cost_val,
              class_rate=class_rate)
    if options.get('write_results') and plan.results is not None:
        with open(options['write_results'], mode='a') as f:
            plan.results(f, st)
    if getattr(plan, 'save_best', None) and best_params is not None:      # cuave/unimodal_with_val.py:364-368
        from ..utils.io import save_model_params
        print('Saving the best model so far...')
        network.restore_params(best_params)
        save_model_params(network, plan.save_best)
        print('Model Saved!')
    st.update(network=network, learning_rate=update.lr, momentum=update.mm)
    return st


def main(dataset, script, argv=None):
    """(the model-zoo factories take their arithmetic from a module-level default that ``--precision`` sets: restored on the
    way out, so that a process which calls several drivers -- the tests do -- builds every model in the mode it asked for)"""
    from ..modelzoo import _factory
    saved = _factory.DEFAULT_PRECISION
    try:
        return _main(dataset, script, argv)
    finally:
        _factory.DEFAULT_PRECISION = saved
