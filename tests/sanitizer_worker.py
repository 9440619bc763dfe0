"""tests/sanitizer_worker.py <mode> [count] — run by tests/test_sanitizers.py under LD_PRELOAD=libasan.so with GSTARK_PROVER_LIB_DIR pointing
at the ASAN + UBSAN build of the native driver (tools/build_sanitized.sh).  Any sanitizer report aborts the process (exit code != 0).

  verify  <count>: per (AIR, hash algorithm) `count` corrupted / truncated / extended serialized proofs through gs_prover_verify_on —
                   MiMC, Poseidon (program AIR), the ledger module (input registers: the proof carries shapes) and the 32-bit flavour;
                   every one must come back as True or as an error value, never as a sanitizer report;
  prove          : gs_prover_prove_on on the error cases of tests/test_native_prover.py::test_native_prover_errors and on jobs whose
                   input shapes do not fit (the job struct is caller-controlled too).
"""
import os
import random
import sys
import zlib

SEED = int(os.environ.get('GSTARK_FUZZ_SEED', '0'))        # another sweep: GSTARK_FUZZ_SEED=n

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from genstark_amd._abi import MODULUS_32, Backend, GstarkError      # noqa: E402
from genstark_amd.errors import StarkError                           # noqa: E402
from genstark_amd.field import PrimeField                            # noqa: E402
from genstark_amd.prover import Prover                               # noqa: E402


def oracle(modulus=None):
    name = 'liboracle.so' if modulus is None else 'liboracle_q32.so'
    return Backend(lib_path=os.path.join(ROOT, 'oracle', name), allow_test_double=True, **({} if modulus is None else {'modulus': modulus}))


def statements():
    """(name, Prover, assertions, inputs, seed, public inputs)"""
    import genstark_amd as ga
    from genstark_amd import airassembly
    from genstark_amd.air_generic import GenericAir
    from genstark_amd.poseidon import poseidon6x128_air
    be = oracle()
    f = PrimeField(backend=be)
    out = []
    for alg in ('sha256', 'blake2s256'):
        o = {'hashAlgorithm': alg, 'extensionFactor': 16, 'exeQueryCount': 24, 'friQueryCount': 12}
        st = ga.instantiateMimc(256, o, backend=be)
        ctl = ga.runMimc(f, 256, st.air.roundConstants, 3)
        out.append((f'mimc-{alg}', Prover(st.air, o), [{'step': 0, 'register': 0, 'value': 3}, {'step': 255, 'register': 0, 'value': ctl[-1]}], [], [3], None))
        air = poseidon6x128_air(128, 16, f)
        tr = air.hostTrace([1, 2, 3, 4])
        out.append((f'poseidon-{alg}', Prover(air, o), [{'step': 63, 'register': 0, 'value': tr[63][0]}, {'step': 127, 'register': 5, 'value': tr[127][5]}], [], [1, 2, 3, 4], None))
        from test_airassembly import AA, ledger_model
        runs = 4
        balances, factors = [100 + 7 * i for i in range(runs)], [3 + i for i in range(runs)]
        deposits = [[5 + i + 2 * j for j in range(4)] for i in range(runs)]
        model = ledger_model(f.modulus, balances, factors, deposits)
        lo = {'hashAlgorithm': alg, 'exeQueryCount': 24, 'friQueryCount': 12}
        lair = airassembly.AssemblyAir(open(os.path.join(AA, 'ledger.aa')).read(), 'default', None, f)
        out.append((f'ledger-{alg}', Prover(lair, lo), [{'step': 0, 'register': 0, 'value': balances[0]}, {'step': 31, 'register': 2, 'value': model[31][2]}],
                    [balances, factors, deposits], None, [deposits]))
    f32 = PrimeField(backend=oracle(MODULUS_32))
    foo = GenericAir(64, 1, [1], [], lambda r, k: [r[0] + 2], lambda r, n, k: [n[0] - (r[0] + 2)], lambda seed: [seed[0]], None, f32)
    out.append(('foo-q32-sha256', Prover(foo, {'hashAlgorithm': 'sha256', 'extensionFactor': foo.extensionFactor}),
                [{'step': 0, 'register': 0, 'value': 1}, {'step': 63, 'register': 0, 'value': 127}], [], [1], None))
    return out


def mutate(rng, data):
    b = bytearray(data)
    kind = rng.randrange(10)
    if kind < 3:                                          # a few single-bit flips anywhere
        for _ in range(rng.randrange(1, 4)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    elif kind < 5:                                        # random bytes in the first 400 bytes (roots, counts, column heads, depths)
        for _ in range(rng.randrange(1, 6)):
            b[rng.randrange(min(400, len(b)))] = rng.randrange(256)
    elif kind == 5:                                       # a count / length byte forced to an extreme
        b[rng.randrange(len(b))] = rng.choice((0, 1, 255, 254, 128))
    elif kind == 6:                                       # truncated
        del b[rng.randrange(len(b) + 1):]
    elif kind == 7:                                       # extended
        b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 80)))
    elif kind == 8:                                       # the tail (remainder, input shapes) rewritten
        k = rng.randrange(1, min(40, len(b)))
        b[-k:] = bytes(rng.randrange(256) for _ in range(k))
    else:                                                 # a slice moved
        i, j, n = rng.randrange(len(b)), rng.randrange(len(b)), rng.randrange(1, 64)
        b[i:i + n] = b[j:j + n]
    return bytes(b)


def verify_mode(count):
    total = accepted = same = 0
    for name, p, a, inputs, seed, public in statements():
        data = p.prove_bytes(a, inputs, seed)
        assert p.verify_native(a, data, public) is True, name
        parsed = p.parse(data)
        rng = random.Random(zlib.crc32(name.encode()) + SEED)
        # the full count for MiMC and the program AIR under both hash algorithms; 2/5 of it for the ledger (shapes) and the 32-bit flavour
        for _ in range(count if name.startswith(('mimc', 'poseidon')) else 2 * count // 5):
            bad = mutate(rng, data)
            try:
                ok = p.verify_native(a, bad, public)
                # accepted: then it must BE the same proof in another encoding — trailing bytes (ignored, as in the reference's parser) or
                # the leaf flag of an empty authentication column (lib/utils/serialization.ts:93-101 never reads it): same parsed object
                if ok and p.parse(bad) != parsed:
                    accepted += 1
                    diff = [i for i in range(min(len(bad), len(data))) if bad[i] != data[i]]
                    print(f'ACCEPTED {name}: lengths {len(data)} -> {len(bad)}, {len(diff)} bytes differ at {diff[:12]}: {[(data[i], bad[i]) for i in diff[:12]]}', flush=True)
                same += bool(ok)
            except (StarkError, GstarkError):
                pass
            total += 1
        # the statement's side is caller-controlled too: public inputs of the wrong size, assertions out of range
        for wrong in ([], [[]], [[[1] * 3] * 4], [[list(range(70000))]]):
            try:
                p.verify_native(a, data, wrong)
            except (StarkError, GstarkError):
                pass
        for aa in ([dict(a[0], step=1 << 40)], [dict(a[0], register=77)]):
            try:
                p.verify_native(aa, data, public)
            except (StarkError, GstarkError):
                pass
    assert accepted == 0, f'{accepted} corrupted proofs were accepted'
    loaded_instrumented_driver()
    print(f'sanitized verify: {total} corrupted proofs, 0 accepted ({same} were other encodings of the same proof), no report')


def prove_mode():
    n = 0
    for name, p, a, inputs, seed, public in statements():
        for bad in ([dict(a[0], value=a[0]['value'] + 1)], [dict(a[0], step=1 << 30)], [dict(a[0], register=99)]):
            try:
                p.prove_bytes(bad, inputs, seed)
                raise AssertionError(f'{name}: a false statement was proved')
            except StarkError:
                n += 1
        if public is not None:               # shapes that lay out another trace than the inputs' columns / no trace at all
            nat = p._native
            inner, packed, firsts, shapes = p.air.plan(inputs, seed)
            import ctypes as C
            from genstark_amd.native import NativeProver, _Shim
            for wrong in ([[8], [4], [4, 4]], [[4], [4], [4, 3]], [[0], [0], [0, 0]], [[4], [4]], [[4, 4], [4], [4, 4]], [[1 << 31], [1 << 31], [1 << 31, 1 << 31]]):
                flat = [w for sh in wrong for w in [len(sh)] + list(sh)]
                ip = NativeProver(_Shim(inner, nat._exe, nat._fri, nat._alg))
                ip._shape_args = (nat._input_decl, nat._ninputs if len(wrong) == 3 else len(wrong), (C.c_uint32 * len(flat))(*flat))
                try:
                    ip.prove_bytes(a, packed, firsts)
                    raise AssertionError(f'{name}: shapes {wrong} were accepted')
                except StarkError:
                    n += 1
    loaded_instrumented_driver()
    print(f'sanitized prove: {n} refused jobs, no report')


def loaded_instrumented_driver():
    """the driver image this process mapped is the one of GSTARK_PROVER_LIB_DIR, and it carries the sanitizer runtime's symbols"""
    want = os.path.realpath(os.environ['GSTARK_PROVER_LIB_DIR'])
    maps = open('/proc/self/maps').read()
    assert any(os.path.realpath(l.split()[-1]).startswith(want) for l in maps.splitlines() if 'libgstark_prover' in l), 'the instrumented driver is not the one loaded'
    assert 'libasan' in maps


if __name__ == '__main__':
    if sys.argv[1] == 'verify':
        verify_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
    else:
        prove_mode()
