#!/usr/bin/env python3
"""Entry point with the reference script's name: ``python3 n_best_asr_bert.py <flags>`` runs the HIP fine-tuning
path (flag surface documented in nbest_amd/cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nbest_amd  # noqa: E402,F401
from nbest_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
