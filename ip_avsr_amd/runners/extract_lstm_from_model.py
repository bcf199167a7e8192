#!/usr/bin/env python3
"""Pull the LSTM layers out of a saved 1-stream model and write them as the ``{prefix}_w_in_to_*`` / ``_w_hid_to_*`` /
``_b_*`` ``.mat`` file that ``create_pretrained_model`` reads (reference runners/extract_lstm_from_model.py:12-87)."""
from __future__ import print_function

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ip_avsr_amd.modelzoo import deltanet_majority_vote  # noqa: E402
from ip_avsr_amd.runners.extract_encoder_from_model import load, parse_options  # noqa: E402
from ip_avsr_amd.utils.io import save_mat  # noqa: E402


def main(argv=None):
    options = parse_options(argv, with_layer_names=True)
    print(options)
    network = load(options)
    layer_names = options['layer_names'].split(',')
    d = deltanet_majority_vote.extract_lstm_weights(network, layer_names, ['f_lstm', 'b_lstm'][:len(layer_names)])
    for k, v in d.items():
        assert k.startswith(('f_lstm_', 'b_lstm_')) and isinstance(v, np.ndarray), k
    if options.get('output'):
        print('save extracted weights to {}'.format(options['output']))
        save_mat(d, options['output'])
    network.close()
    return d


if __name__ == '__main__':
    main()
