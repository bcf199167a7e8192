#!/usr/bin/env python3
"""2-stream trainer, 'dct' variant (reference runners/2stream_dct.py); see ip_avsr_amd/runners/nstream.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ip_avsr_amd.runners.nstream import main  # noqa: E402

if __name__ == '__main__':
    main(2, variant='dct')
