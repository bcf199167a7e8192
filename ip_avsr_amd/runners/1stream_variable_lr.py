#!/usr/bin/env python3
"""1-stream trainer with per-layer Adam learning rates (reference runners/1stream_variable_lr.py): fc1, fc2, fc3 at 0.001,
every other layer at the configured rate, and -- as in the reference -- the three rates set to 100.0 after epoch 4 to
show that they act (the training loss must blow up).  Pass your own ``--layer_lr`` / ``--explode_layer_lr`` to change it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ip_avsr_amd.runners.nstream import main  # noqa: E402


def run(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not any(a.startswith('--layer_lr') for a in argv):
        argv += ['--layer_lr', 'fc1=0.001,fc2=0.001,fc3=0.001']
    if not any(a.startswith('--explode_layer_lr') for a in argv):
        argv += ['--explode_layer_lr', '4:100.0']
    return main(1, argv)


if __name__ == '__main__':
    run()
