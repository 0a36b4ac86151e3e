"""The synthetic BASELINE inputs live in the package (genome-downsampler_amd/synthetic.py); the tests and lab
scripts keep importing them from here."""
import importlib

_syn = importlib.import_module("genome-downsampler_amd.synthetic")
GRCH38_MB = _syn.GRCH38_MB
amplicon_panel = _syn.amplicon_panel
amplicon_reads = _syn.amplicon_reads
wgs_contigs = _syn.wgs_contigs
wgs_shape = _syn.wgs_shape
cfg5_heaviest_share = _syn.cfg5_heaviest_share
clipped_mix = _syn.clipped_mix
lengthened_mix = _syn.lengthened_mix
