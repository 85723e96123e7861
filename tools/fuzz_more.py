"""one-off: the randomized parity test of tests/test_gpu_parity.py over many more seeds (developer tool; prints the seeds that fail)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pyoracle as O
import test_gpu_parity as T
O.lib()
bad = []
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    try:
        T.test_random_materials_textures_and_lights_match_checker.__wrapped__(O, seed) if hasattr(T.test_random_materials_textures_and_lights_match_checker, "__wrapped__") else T.test_random_materials_textures_and_lights_match_checker(O, seed)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED", str(e)[:200], flush=True)
print("seeds", lo, "..", hi, "failed:", bad)
