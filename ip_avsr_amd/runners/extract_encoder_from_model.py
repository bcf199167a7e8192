#!/usr/bin/env python3
"""Pull the fine-tuned encoder out of a saved 1-stream model (.pkl parameter list) and write it as the w1..w4 / b1..b4
``.mat`` file the stream loaders read (reference runners/extract_encoder_from_model.py:12-77; same options)."""
from __future__ import print_function

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ip_avsr_amd.modelzoo import deltanet_majority_vote  # noqa: E402
from ip_avsr_amd.utils.io import save_mat  # noqa: E402

ENC_LAYERS = ['fc1', 'fc2', 'fc3', 'bottleneck']
ENC_KEYS = [('w1', 'b1'), ('w2', 'b2'), ('w3', 'b3'), ('w4', 'b4')]


def parse_options(argv=None, with_layer_names=False):
    p = argparse.ArgumentParser()
    p.add_argument('--shape', default='2000,1000,500,50', help='shape of encoder. Default: 2000,1000,500,50')
    p.add_argument('--input_dim', type=int, default=1200, help='input dimension. Default: 1200')
    p.add_argument('--nonlinearities', default='rectify,rectify,rectify,linear',
                   help='nonlinearities used by the encoder. Default: rectify,rectify,rectify,linear')
    p.add_argument('--output', help='output file to write results')
    p.add_argument('--lstm_size', type=int, default=250, help='lstm layer size. Default: 250')
    p.add_argument('--output_classes', type=int, default=26, help='number of output classes')
    if with_layer_names:
        p.add_argument('--layer_names', default='f_blstm1,b_blstm1', help='names of lstm layers to extract')
    p.add_argument('--use_blstm', action='store_true', help='use blstm')
    p.add_argument('input', help='input model.pkl file')
    return vars(p.parse_args(argv))


def load(options):
    shape = [int(i) for i in options['shape'].split(',')]
    nonlinearities = options['nonlinearities'].split(',')
    return deltanet_majority_vote.load_saved_model(options['input'], (shape, nonlinearities),
                                                   (None, None, options['input_dim']), None, (None, None), None,
                                                   options['lstm_size'], None, options['output_classes'],
                                                   use_blstm=options['use_blstm'])


def main(argv=None):
    options = parse_options(argv)
    print(options)
    network = load(options)
    n = len(options['shape'].split(','))
    d = deltanet_majority_vote.extract_encoder_weights(network, ENC_LAYERS[:n], ENC_KEYS[:n])
    for k, v in d.items():
        assert isinstance(v, np.ndarray), k
    if options.get('output'):
        print('save extracted weights to {}'.format(options['output']))
        save_mat(d, options['output'])
    network.close()
    return d


if __name__ == '__main__':
    main()
