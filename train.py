#!/usr/bin/env python3
"""CLI shim: `python train.py --cfg_file ... --workdir ... --logdir ... [--opts K V ...]` like CARL_MVF/train.py."""
from video_rep_learning_amd.train import main

if __name__ == '__main__':
    main()
