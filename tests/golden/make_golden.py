"""tests/golden/make_golden.py — regenerates the committed fixtures.  Run in the BUILD container:

    python tests/golden/make_golden.py

1. oracle_proofs.json   — proofs, roots and positions produced by THIS repo's oracle (oracle/pyref.py) for
                          small MiMC instances: the byte-exact targets of the GPU path.
2. reference_driver_proofs.json — proofs produced by the reference's OWN compiled prover (bin/lib/Stark.js +
                          bin/lib/components/*.js, unmodified, run under node from /root/reference) orchestrating THIS
                          repository's drop-in modules for galois / merkle / air-assembly (js/shims via NODE_PATH) on the
                          CPU oracle backend: every line of prove()/serialize()/verify() executed is the reference's.
3. reference_vectors.json — outputs of the reference's OWN compiled modules that can run here
                          (QueryIndexGenerator, Serializer, sizeOf, powLog2, read/writeBigInt), produced by
                          gen_reference_vectors.js under node with /root/reference/bin/lib.  The reference's
                          arithmetic packages are absent, so nothing else of the reference can be executed.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

CASES = [
    dict(name='mimc_T64_E16_blake2s', steps=64, extension_factor=16, exe_query_count=48, fri_query_count=24, hash_algorithm='blake2s256'),
    dict(name='mimc_T64_E8_sha256', steps=64, extension_factor=8, exe_query_count=80, fri_query_count=40, hash_algorithm='sha256'),
    dict(name='mimc_T256_E16_blake2s', steps=256, extension_factor=16, exe_query_count=48, fri_query_count=24, hash_algorithm='blake2s256'),
    dict(name='mimc_T1024_E8_blake2s', steps=1024, extension_factor=8, exe_query_count=48, fri_query_count=24, hash_algorithm='blake2s256'),
    dict(name='mimc_T1024_E16_sha256', steps=1024, extension_factor=16, exe_query_count=48, fri_query_count=64, hash_algorithm='sha256'),
]


def hexify_merkle(p):
    return {'values': [v.hex() for v in p['values']], 'nodes': [[x.hex() for x in c] for c in p['nodes']], 'depth': p['depth']}


def hexify(proof):
    ld = proof['ldProof']
    return {'evRoot': proof['evRoot'].hex(), 'evProof': hexify_merkle(proof['evProof']),
            'ldProof': {'lcRoot': ld['lcRoot'].hex(), 'lcProof': hexify_merkle(ld['lcProof']),
                        'components': [{'columnRoot': c['columnRoot'].hex(), 'columnProof': hexify_merkle(c['columnProof']),
                                        'polyProof': hexify_merkle(c['polyProof'])} for c in ld['components']],
                        'remainder': [str(v) for v in ld['remainder']]},
            'iShapes': proof['iShapes']}


def main():
    oracle_out, for_node = [], []
    for case in CASES:
        kw = {k: v for k, v in case.items() if k != 'name'}
        cfg = pyref.MimcConfig(**kw)
        assertions = pyref.mimc_assertions(cfg)
        proof, info = pyref.prove(cfg, assertions)
        buf = pyref.serialize(cfg, proof)
        assert len(buf) == pyref.size_of(proof)
        assert pyref.verify(cfg, assertions, pyref.parse(cfg, buf))
        rec = dict(case)
        rec.update(seed=cfg.seed, root_of_unity=str(cfg.root), round_constants_sha256=hashlib.sha256(
            b''.join(pyref.to_bytes(k) for k in cfg.rc)).hexdigest(),
            assertions=[{'step': a['step'], 'register': a['register'], 'value': str(a['value'])} for a in assertions],
            evRoot=info['evRoot'].hex(), lcRoot=info['lcRoot'].hex(), columnRoots=[r.hex() for r in info['columnRoots']],
            exePositions=info['exePositions'], remainderLength=len(proof['ldProof']['remainder']),
            friLayers=len(proof['ldProof']['components']), proofSize=len(buf),
            proofSha256=hashlib.sha256(buf).hexdigest(), proofHex=buf.hex())
        oracle_out.append(rec)
        for_node.append({'name': case['name'], 'elementSize': 16, 'digestSize': 32, 'traceRegisterCount': 1, 'secretInputCount': 0,
                         'proof': hexify(proof)})
        print(case['name'], 'size', len(buf), 'layers', rec['friLayers'])
    # a structurally richer synthetic proof for the wire-format test: 2 trace + 1 secret register, input shapes
    syn = json.loads(json.dumps(for_node[0]))
    syn['name'] = 'synthetic_shapes'
    syn['traceRegisterCount'], syn['secretInputCount'] = 2, 1
    syn['proof']['iShapes'] = [[1], [2, 4], []]
    syn['proof']['evProof']['values'] = [(v * 3) for v in syn['proof']['evProof']['values']]  # 48-byte leaves
    for_node.append(syn)
    with open(os.path.join(HERE, 'oracle_proofs.json'), 'w') as f:
        json.dump(oracle_out, f)
    tmp = os.path.join(HERE, '_proofs_for_node.json')
    with open(tmp, 'w') as f:
        json.dump(for_node, f)
    ref = os.environ.get('GENSTARK_REFERENCE', '/root/reference')
    subprocess.check_call(['node', os.path.join(HERE, 'gen_reference_vectors.js'), os.path.join(ref, 'bin', 'lib'), tmp,
                           os.path.join(HERE, 'reference_vectors.json')])
    os.remove(tmp)
    run_reference_driver(oracle_out, ref)


def run_reference_driver(oracle_cases, ref):
    """The reference's compiled Stark.js drives our drop-in JS modules (N-API -> C ABI -> oracle backend on CPU)."""
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    cases = [{k: c[k] for k in ('name', 'steps', 'extension_factor', 'exe_query_count', 'fri_query_count', 'hash_algorithm', 'seed',
                                'assertions')} for c in oracle_cases]
    tmp_in, tmp_out = os.path.join(HERE, '_driver_cases.json'), os.path.join(HERE, '_driver_out.json')
    with open(tmp_in, 'w') as f:
        json.dump(cases, f)
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, 'js', 'shims'), GSTARK_LIB=os.path.join(ROOT, 'oracle', 'liboracle.so'),
               GSTARK_ALLOW_TEST_DOUBLE='1')
    subprocess.check_call(['node', os.path.join(HERE, 'run_reference_stark.js'), os.path.join(ref, 'bin', 'lib'), tmp_in, tmp_out], env=env)
    out = json.load(open(tmp_out))
    for rec in out:   # keep the fixture small: the proof itself is identified by its hash and length
        data = bytes.fromhex(rec.pop('proofHex'))
        rec['proofSize'], rec['proofSha256'] = len(data), hashlib.sha256(data).hexdigest()
    with open(os.path.join(HERE, 'reference_driver_proofs.json'), 'w') as f:
        json.dump({'generator': 'genSTARK bin/lib/Stark.js (unmodified) under node, over js/shims + oracle backend', 'cases': cases,
                   'results': out}, f)
    os.remove(tmp_in)
    os.remove(tmp_out)


def run_reference_driver_generic(ref, names=None):
    """reference_driver_generic.json: the reference's compiled Stark.js proving Rescue 4x128 / Poseidon 6x128 (tests/generic_cases.py,
    up to BASELINE.json's C3 / C4 shapes) over the drop-in modules — `instantiate({generic: descriptor})` of
    js/shims/@guildofweavers/air-assembly, js/air_generic.js, N-API, C ABI — on the CPU oracle backend.  Kept per proof: hash,
    size, roots (the descriptors are re-derived from genstark_amd/rescue.py / poseidon.py by whoever checks)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import generic_cases
    from genstark_amd._abi import Backend
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    lib = os.path.join(ROOT, 'oracle', 'liboracle.so')
    backend = Backend(lib_path=lib, allow_test_double=True)
    names = names or list(generic_cases.GENERIC_CASES)
    tmp_in, tmp_out = os.path.join(HERE, '_generic_cases.json'), os.path.join(HERE, '_generic_out.json')
    with open(tmp_in, 'w') as f:
        json.dump([generic_cases.node_case(n, backend) for n in names], f)
    env = dict(os.environ, NODE_PATH=os.path.join(ROOT, 'js', 'shims'), GSTARK_LIB=lib, GSTARK_ALLOW_TEST_DOUBLE='1')
    subprocess.check_call(['node', os.path.join(HERE, 'run_reference_stark.js'), os.path.join(ref, 'bin', 'lib'), tmp_in, tmp_out], env=env)
    out = json.load(open(tmp_out))
    for rec in out:
        data = bytes.fromhex(rec.pop('proofHex'))
        rec.pop('nativeDriverEqualsReference', None)
        rec['proofSize'], rec['proofSha256'] = len(data), hashlib.sha256(data).hexdigest()
    with open(os.path.join(HERE, 'reference_driver_generic.json'), 'w') as f:
        json.dump({'generator': 'genSTARK bin/lib/Stark.js (unmodified) under node, over js/shims ({generic: descriptor}) + oracle backend; '
                                'cases: tests/generic_cases.py', 'results': out}, f, indent=0)
    os.remove(tmp_in)
    os.remove(tmp_out)


if __name__ == '__main__':
    import sys
    if sys.argv[1:2] == ['generic']:
        run_reference_driver_generic(os.environ.get('GENSTARK_REFERENCE', '/root/reference'), sys.argv[2:] or None)
    else:
        main()
        run_reference_driver_generic(os.environ.get('GENSTARK_REFERENCE', '/root/reference'))
