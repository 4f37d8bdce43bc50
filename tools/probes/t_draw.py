"""Probe: in-kernel clock and cycles per 624-word block of the MT19937 generator (TFR_RNG_DEBUG=1)."""
import os, sys, time
os.environ["TFR_RNG_DEBUG"] = "1"
sys.path.insert(0, ".")
import tfrecomm_amd as T
m = T.SvdModel(50, 40, 8, device=0)
m.rng_seed(1)
for high, count in ((900188, 10000), (900188, 10000), (900188, 1000000), (1 << 20, 1000000), (90_000_000, 262144), (90_000_000, 2000000)):
    m.draw_ids(high, count)
