"""Trainer of the convolutional auto-encoder: reference avletters/avletters_convae.py:118-329 on the MI355X model
(``ip_avsr_amd.convae.ConvAE`` behind ``modelzoo.avletters_convae*.create_model``).

    python -m ip_avsr_amd.avletters.avletters_convae [--epochs N] [--bottleneck 50] [--dense 500] [--model normal]
                                                     [--data X.mat]

Kept from the reference: the AVLetters split by ``iterVec`` (repetitions 1-2 train, 3 validation), resize 60x80 -> 30x40
and per-frame z-normalisation (:118-157); ``batch_iterator`` minibatches of 128 (with its zero-padded remainder batch and
its cursor arithmetic); adadelta with learning rate 0.8, multiplied by 0.9 after every epoch past the 11th (:241-242,
:312-313); epoch costs as the mean over ``NO_STRIDES`` equal slices -- the TRAIN cost with the stochastic layers active
(``train_cost_fn`` is built on ``deterministic=False``), the validation cost deterministic (:159-165, :257-268); SIGINT
stops after the current batch (:204-210); the encoder and the whole network are saved at the end (:325-326; here as
parameter lists: utils/io.save_model, the Lasagne layer objects do not exist).
New: ``--data`` (the reference hard-codes data/allData_mouthROIs.mat), ``--epoch_size``, ``--save_prefix``, ``--seed``,
``--precision``; plots are skipped when matplotlib is missing.
"""
from __future__ import print_function

import argparse
import signal
import sys
import time

import numpy as np

from ..modelzoo import avletters_convae, avletters_convae_bn, avletters_convae_bndrop, avletters_convae_drop
from ..utils.datagen import batch_iterator
from ..utils.io import load_mat_file, save_model
from ..utils.plotting_utils import plot_validation_cost
from ..utils.preprocessing import create_split_index, normalize_input, resize_images, split_videolen

terminate = False


def generate_data(path='data/allData_mouthROIs.mat', orig_dim=(60, 80), dim=(30, 40)):
    """(train, validation) image stacks (n, 30, 40) float32 (reference :118-157)."""
    print('preprocessing dataset...')
    data = load_mat_file(path)
    data_matrix = data['dataMatrix']
    vid_len_vec = data['videoLengthVec'].reshape((-1,)).astype(int)
    iter_vec = data['iterVec'].reshape((-1,))
    indexes = create_split_index(data_matrix.shape[0], vid_len_vec, iter_vec)
    train_lens, test_lens = split_videolen(vid_len_vec, iter_vec)
    assert np.sum(vid_len_vec) == data_matrix.shape[0]
    out = []
    for part in (data_matrix[indexes], data_matrix[~indexes]):
        r = resize_images(part, orig_dim, dim).astype(np.float32)
        r = normalize_input(r, centralize=True)
        out.append(np.reshape(r, (-1, dim[0], dim[1])).astype(np.float32))
    return out[0], out[1]


def batch_compute_cost(X, y, no_strides, cost_fn):
    """Mean of ``cost_fn`` over ``no_strides`` equal consecutive slices (reference :159-165; integer stride, the
    remainder rows are not visited)."""
    cost = 0.0
    stride_size = len(X) // no_strides
    for j in range(no_strides):
        j *= stride_size
        cost += float(cost_fn(X[j:j + stride_size], y[j:j + stride_size]))
    return cost / float(no_strides)


def parse_options(argv=None):
    options = dict(NUM_EPOCHS=20, EPOCH_SIZE=96, NO_STRIDES=3, VAL_NO_STRIDES=3, DENSE=500, BOTTLENECK=50, MODEL='normal')
    parser = argparse.ArgumentParser()
    parser.add_argument('--epochs', help='number of epochs to run')
    parser.add_argument('--bottleneck', help='bottleneck size')
    parser.add_argument('--dense', help='dense layer size')
    parser.add_argument('--model', help='model to run: normal | batchnorm | dropout | bn+dropout')
    parser.add_argument('--data', default='data/allData_mouthROIs.mat', help='.mat with dataMatrix / videoLengthVec / iterVec')
    parser.add_argument('--epoch_size', help='minibatches per epoch (reference: 96)')
    parser.add_argument('--save_prefix', default='models/conv', help="writes <prefix>_encoder.dat and <prefix>_ae.dat")
    parser.add_argument('--image', default='60,80,30,40', help='original and resized image size "H0,W0,H,W"')
    parser.add_argument('--precision', default='f32', choices=['f32', 'bf16'])
    parser.add_argument('--seed', type=int, default=None)
    args = parser.parse_args(argv)
    if args.epochs:
        options['NUM_EPOCHS'] = int(args.epochs)
    if args.bottleneck:
        options['BOTTLENECK'] = int(args.bottleneck)
    if args.dense:
        options['DENSE'] = int(args.dense)
    if args.model:
        options['MODEL'] = args.model
    if args.epoch_size:
        options['EPOCH_SIZE'] = int(args.epoch_size)
    options.update(DATA=args.data, SAVE_PREFIX=args.save_prefix, PRECISION=args.precision, SEED=args.seed,
                   IMAGE=tuple(int(v) for v in args.image.split(',')))
    return options


# --model values of avletters/avletters_convae.py:245-252
FACTORIES = {'normal': avletters_convae, 'batchnorm': avletters_convae_bn, 'dropout': avletters_convae_drop,
             'bn+dropout': avletters_convae_bndrop}


def main(argv=None, data=None):
    """``data``: optional (X, X_val) image stacks instead of ``--data`` (tests)."""
    global terminate
    terminate = False

    def signal_handler(sig, frame):
        global terminate
        terminate = True
        print('terminating...')

    try:
        signal.signal(signal.SIGINT, signal_handler)
    except ValueError:                                # not the main thread
        pass
    options = parse_options(argv)
    if options['SEED'] is not None:
        np.random.seed(options['SEED'])
    h0, w0, h, w = options['IMAGE']
    X, X_val = data if data is not None else generate_data(options['DATA'], (h0, w0), (h, w))
    print('X type and shape:', X.dtype, X.shape)
    print('X.min():', X.min())
    print('X.max():', X.max())
    print('X_val type and shape:', X_val.dtype, X_val.shape)
    X_out = X.reshape((X.shape[0], -1))
    X_val_out = X_val.reshape((X_val.shape[0], -1))

    print('constructing and compiling model...')
    if options['MODEL'] not in FACTORIES:
        raise ValueError('--model must be one of %s' % sorted(FACTORIES))
    network, encoder = FACTORIES[options['MODEL']].create_model((None, 1, h, w), options)
    print('AE Network architecture: {}'.format(options['MODEL']))
    if options.get('SEED') is not None:
        network.set_dropout_state(options['SEED'])
    lr, lr_decay = np.float32(0.8), np.float32(0.9)

    def train(bx, by):
        return network.train(bx.reshape((len(bx), -1)), by, learning_rate=float(lr), want_loss=False)

    # get_output(network, deterministic=False) for the training cost, deterministic=True for evaluation and reconstruction
    # (avletters/avletters_convae.py:254-268)
    train_cost_fn = lambda bx, by: network.cost(bx.reshape((len(bx), -1)), by, deterministic=False)
    eval_cost_fn = lambda bx, by: network.cost(bx.reshape((len(bx), -1)), by)
    recon_fn = lambda bx: network.recon_fn(bx.reshape((len(bx), -1)))

    NUM_EPOCHS, EPOCH_SIZE = options['NUM_EPOCHS'], options['EPOCH_SIZE']
    print('begin training for {} epochs...'.format(NUM_EPOCHS))
    datagen = batch_iterator(X, X_out, 128)
    costs, val_costs = [], []
    for epoch in range(NUM_EPOCHS):
        time_start = time.time()
        for i in range(EPOCH_SIZE):
            batch_X, batch_y = next(datagen)
            print('Epoch {} batch {}/{}: {} examples at learning rate = {:.4f}'.format(epoch + 1, i + 1, EPOCH_SIZE, len(batch_X),
                                                                                      float(lr)), end='')
            sys.stdout.flush()
            train(batch_X, batch_y)
            print('\r', end='')
            if terminate:
                break
        if terminate:
            break
        cost = batch_compute_cost(X, X_out, options['NO_STRIDES'], train_cost_fn)
        val_cost = batch_compute_cost(X_val, X_val_out, options['VAL_NO_STRIDES'], eval_cost_fn)
        costs.append(cost)
        val_costs.append(val_cost)
        print("Epoch {} train cost = {}, validation cost = {} ({:.1f}sec) ".format(epoch + 1, cost, val_cost, time.time() - time_start))
        if epoch > 10:
            lr = np.float32(lr * lr_decay)
    X_val_recon = recon_fn(X_val[:512])
    try:
        plot_validation_cost(costs, val_costs, None, savefilename='valid_cost')
    except Exception as e:
        print('(no plot: %s)' % e)
    print('saving encoder...')
    save_model(encoder.get_all_param_values(), options['SAVE_PREFIX'] + '_encoder.dat')
    save_model(network.get_all_param_values(), options['SAVE_PREFIX'] + '_ae.dat')
    return dict(costs=costs, val_costs=val_costs, network=network, encoder=encoder, learning_rate=float(lr), recon=X_val_recon)


if __name__ == '__main__':
    main()
