#!/usr/bin/env python3
"""helloworld under torch.autocast (reference: tutel/examples/helloworld_amp.py): fp32 master weights, tokens and expert
GEMMs in the autocast dtype."""
from .helloworld import main

if __name__ == "__main__":
    main(amp=True)
