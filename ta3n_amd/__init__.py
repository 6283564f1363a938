"""ta3n_amd: MI355X-native (gfx950) implementation of TA3N's temporal-adversarial
train step behind the reference's VideoModel.forward()/opts.py surface.

Importing the package never loads the HIP library; the first compute call does,
and raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
