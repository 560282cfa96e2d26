"""Regenerates the oracle's golden frames (tests/golden/oracle_*.npy).

These are outputs of oracle/mcpt_oracle.c itself (the reference cannot be built in this image: no Eigen3),
kept to detect unintended changes of the oracle and to give the GPU tests a fixed target that does not
depend on rebuilding the oracle.  reference_cornellbox_demo.png is the reference repository's own
cornellbox_demo.png, copied as data.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mcpt_loader  # noqa: E402
from oracle import oracle  # noqa: E402

pkg = mcpt_loader.load()
here = os.path.dirname(os.path.abspath(__file__))
fb, _ = oracle.OracleScene(pkg.scenes.cornell_demo(48, 48, 4)).render(spp=4, seed=1)
np.save(os.path.join(here, "oracle_cornell_demo_48x48_spp4.npy"), fb)
fb, _ = oracle.OracleScene(pkg.scenes.chess_scene(width=96, height=54, spp=2)).render(spp=2, seed=1)
np.save(os.path.join(here, "oracle_chess_96x54_spp2.npy"), fb)
print("golden frames written")
