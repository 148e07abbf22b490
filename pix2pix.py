#!/usr/bin/env python3
"""Drop-in entry point with the reference's command line (pix2pix.py:341-461):
python3 pix2pix.py --train|--predict --data <dir> --output <dir> [...]"""
from gan_amd.pix2pix import main, parse_opt

if __name__ == '__main__':
    main(parse_opt())
