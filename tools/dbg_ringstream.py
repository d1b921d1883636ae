"""one-pass form against the oracle, per operator and row, on the ragged test column; run-to-run reproducibility"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from geopolars_amd.geoseries import GeoSeries
from geopolars_amd import synth
import test_gpu_ops as T
from oracle import pyoracle as oracle
oracle.build()
oracle.lib()
for name in ("ragged", "multipoly", "holes"):
    a = T._arrays()[name]
    s = GeoSeries(a)
    for op, got, exp in (("area", s.area(), oracle.area(a)), ("signed", s.signed_area(), oracle.area(a, signed=True)), ("length", s.euclidean_length(), oracle.euclidean_length(a))):
        rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-300)
        bad = np.nonzero(~(rel <= 1e-9) & ~(np.isnan(got) & np.isnan(exp)))[0]
        print(name, op, "bad rows", bad[:10], [(got[i], exp[i]) for i in bad[:5]])
    b = s.bounds(); eb = oracle.bounds(a)
    print(name, "bounds equal", np.array_equal(b, eb, equal_nan=True))
    r1 = s.area(); r2 = s.area()
    print(name, "area reproducible", np.array_equal(r1, r2, equal_nan=True))
