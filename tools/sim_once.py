import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MSA_DIAGNOSTICS", "1")  # (the library reads its MSA_* diagnostic switches only under this one)
import numpy as np, torch
from pytrimal_amd import _lib
from pytrimal_amd.matrix import SimilarityMatrix
from pytrimal_amd.synth import synth_msa
m, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
vhash, dist = SimilarityMatrix.aa()._device_arrays()
a = synth_msa(m, n, seed)
ctx = _lib.Context(0)
for _ in range(2):
    ctx.upload(a, ord("X")); ctx.similarity(vhash, dist)
