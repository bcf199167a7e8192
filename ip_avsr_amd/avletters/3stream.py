#!/usr/bin/env python3
"""AVLetters 3-stream trainer (reference avletters/3stream.py: ``has_encoder`` per stream, iterVec train / val split);
see ip_avsr_amd/runners/nstream.py, variant 'avletters'."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ip_avsr_amd.runners.nstream import main  # noqa: E402

if __name__ == '__main__':
    main(3, variant='avletters')
