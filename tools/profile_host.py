"""tools/profile_host.py — where the HOST time of prove() goes (cProfile over a few proofs on the HIP backend).
usage: python tools/profile_host.py [log2 steps] > profiles/xxx.txt"""
import cProfile
import gc
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
import genstark_amd as ga
from genstark_amd._abi import Backend

log_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = Backend(device=0)
stark = bench.make_stark(ga, be, 1 << log_steps, 16, 64)
a = bench.assertions_for(stark, 1 << log_steps, 3)
for _ in range(2):
    stark.prove(a, [], [3])
gc.collect()
gc.freeze()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    stark.prove(a, [], [3])
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumulative').print_stats(45)
