import sys, time, gc
sys.path.insert(0, '.')
import genstark_amd as ga, bench
from genstark_amd._abi import Backend
from genstark_amd.native import NativeProver
be = Backend(device=0)
for logt, fri in ((13, 24), (17, 24), (20, 64)):
    steps = 1 << logt
    stark = bench.make_stark(ga, be, steps, 16, fri)
    a = bench.assertions_for(stark, steps, 3)
    nat = NativeProver(stark)
    for _ in range(3):
        d = nat.prove_bytes(a, [], [3]); p = stark.prove(a, [], [3])
    gc.collect(); gc.freeze()
    tn, tp = [], []
    for _ in range(8):
        t0 = time.perf_counter(); d = nat.prove_bytes(a, [], [3]); tn.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); p = stark.prove(a, [], [3]); be.sync(); tp.append((time.perf_counter() - t0) * 1e3)
    assert d == stark.serialize(p)
    print(f'2^{logt}: native best {min(tn):.2f} mean {sum(tn)/len(tn):.2f} ms | python mirror best {min(tp):.2f} mean {sum(tp)/len(tp):.2f} ms')
