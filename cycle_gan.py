#!/usr/bin/env python3
"""Drop-in entry point with the reference's command line (cycle_gan.py:379-502):
python3 cycle_gan.py --train|--predict --input-images <dir> [--target-images <dir>] --output <dir> [...]"""
from gan_amd.cycle_gan import main, parse_opt

if __name__ == '__main__':
    main(parse_opt())
