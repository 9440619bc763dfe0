"""The node-side boundary: napi/gstark_napi.cc (N-API shim over include/gstark.h) + js/galois.js, js/merkle.js
(the FiniteField / Hash / MerkleTree objects lib/Stark.ts consumes).  On CPU the whole JS -> N-API -> C-ABI plumbing
is exercised against the oracle's implementation of the ABI (test double, explicitly allowed by an env var that
only tests set); on the GPU box the same script runs against libgstark_hip.so."""
import os
import shutil
import subprocess

import pytest

from conftest import ORACLE_LIB, ROOT

NODE = shutil.which('node')
HAVE_HEADERS = os.path.exists('/usr/include/node/node_api.h')
pytestmark = pytest.mark.skipif(not (NODE and HAVE_HEADERS), reason='node or its headers are not in this image')


@pytest.fixture(scope='module')
def addon():
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    return os.path.join(ROOT, 'napi', 'gstark_napi.node')


def run_node(env_extra, script='js/smoke.js', args=()):
    env = dict(os.environ, **env_extra)
    return subprocess.run([NODE, os.path.join(ROOT, script), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_js_facade_plumbing_on_oracle_double(addon, oracle_backend):
    r = run_node({'GSTARK_LIB': ORACLE_LIB, 'GSTARK_ALLOW_TEST_DOUBLE': '1'})
    assert r.returncode == 0, r.stderr
    assert 'js smoke OK' in r.stdout


def test_js_facade_refuses_non_hip_backend(addon, oracle_backend):
    r = run_node({'GSTARK_LIB': ORACLE_LIB})
    assert r.returncode != 0 and 'refusing backend' in r.stderr


def test_js_facade_fails_loudly_without_library(addon):
    r = run_node({'GSTARK_LIB': '/nonexistent/libgstark_hip.so'})
    assert r.returncode != 0 and 'no CPU fallback' in r.stderr


@pytest.mark.gpu
def test_js_facade_on_hip(addon):
    r = run_node({})
    assert r.returncode == 0, r.stderr
    assert 'hip-gfx950' in r.stdout


def test_js_create_prime_field_for_any_modulus_on_oracle_double(addon, oracle_backend):
    """createPrimeField(modulus) for a modulus without a build of its own: the runtime-modulus library (here its checker, liboracle_rt.so)"""
    r = run_node({'GSTARK_LIB': os.path.join(ROOT, 'oracle', 'liboracle_rt.so'), 'GSTARK_ALLOW_TEST_DOUBLE': '1', 'GSTARK_SET_MODULUS': '1'},
                 script='tests/js_runtime_modulus.js', args=['2042167297'])
    assert r.returncode == 0 and 'js runtime modulus OK' in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_js_create_prime_field_for_any_modulus_on_hip(addon):
    r = run_node({}, script='tests/js_runtime_modulus.js', args=['1945555039024054273'])      # 27 * 2^56 + 1, a 61-bit prime
    assert r.returncode == 0 and 'js runtime modulus OK' in r.stdout, r.stderr[-2000:]


# ---- AIRs as descriptors (js/air_generic.js): the members lib/Stark.ts calls on the AIR, node vs the Python host, same library ---
def _generic_members(backend, lib_env, names, tmp_path):
    import hashlib
    import json
    import generic_cases
    cin, cout = tmp_path / 'cases.json', tmp_path / 'out.json'
    cin.write_text(json.dumps([generic_cases.node_case(n, backend) for n in names]))
    env = dict(os.environ, **lib_env)
    r = subprocess.run([NODE, os.path.join(ROOT, 'js', 'smoke_generic.js'), str(cin), str(cout)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    sha = lambda b: hashlib.sha256(b).hexdigest()
    with open(os.path.join(ROOT, 'tests', 'golden', 'reference_driver_generic.json')) as fh:
        fixture = {r['name']: r for r in json.load(fh)['results']}
    for rec in json.loads(cout.read_text()):
        # the native driver called from node: the bytes the reference's own Stark.js produced for this case (committed fixture)
        assert (rec['proofSize'], rec['proofSha256']) == (fixture[rec['name']]['proofSize'], fixture[rec['name']]['proofSha256']), rec['name']
        air, stark, seed, _ = generic_cases.build(rec['name'], backend)
        f = air.field
        ctx = air.initProvingContext([], seed)
        trace = ctx.generateExecutionTrace()
        polys = f.interpolateRoots(ctx.executionDomain, trace)
        q = ctx.evaluateTransitionConstraints(polys)
        assert rec['rows'] == [trace.rowCount, q.rowCount, ctx.generateStaticTrace().rowCount] and rec['cols'] == [trace.colCount, q.colCount]
        assert rec['trace'] == sha(trace.toBuffer()), rec['name']
        assert rec['constraints'] == sha(q.toBuffer()), rec['name']
        assert rec['statics'] == sha(ctx.generateStaticTrace().toBuffer()), rec['name']
        p_ev = f.evalPolysAtRoots(polys, ctx.evaluationDomain)
        pos, n, ef = 8, ctx.evaluationDomain.length, ctx.extensionFactor
        r_vals = [p_ev.getValue(i, pos) for i in range(air.traceRegisterCount)]
        n_vals = [p_ev.getValue(i, (pos + ef) % n) for i in range(air.traceRegisterCount)]
        at = air.initVerificationContext([], []).evaluateConstraintsAt(ctx.evaluationDomain.getValue(pos), r_vals, n_vals, [])
        assert rec['constraintsAt'] == [str(v) for v in at], rec['name']
        # ... and the interpreter agrees with the device's evaluation at the same point of the composition domain
        cf = ctx.compositionFactor
        if pos % (ef // cf) == 0:
            assert at == [q.getValue(k, pos // (ef // cf)) for k in range(q.rowCount)]


def test_js_generic_air_members_on_oracle_double(addon, oracle_backend, tmp_path):
    import generic_cases
    _generic_members(oracle_backend, {'GSTARK_LIB': ORACLE_LIB, 'GSTARK_ALLOW_TEST_DOUBLE': '1'}, generic_cases.SMALL, tmp_path)


@pytest.mark.gpu
def test_js_generic_air_members_on_hip(addon, hip_backend, tmp_path):
    """node -> N-API -> libgstark_hip.so, Rescue / Poseidon up to the C3 / C4 shapes: the same bytes as the Python host over the same kernels."""
    import generic_cases
    _generic_members(hip_backend, {}, list(generic_cases.GENERIC_CASES), tmp_path)


def test_js_generic_air_rejects_bad_descriptors(addon, oracle_backend, tmp_path):
    import json
    import generic_cases
    case = generic_cases.node_case('poseidon_128', oracle_backend)
    script = tmp_path / 'bad.js'
    script.write_text("""
const path = require('path'), assert = require('assert');
const { instantiate } = require(path.join(process.argv[2], 'js', 'shims', '@guildofweavers', 'air-assembly'));
const c = JSON.parse(require('fs').readFileSync(process.argv[3], 'utf8'));
assert.throws(() => instantiate({}, 'default', {}), /expected an AirSchema .* or an AIR descriptor/);
assert.throws(() => instantiate({ generic: Object.assign({}, c.generic, { steps: 96 }) }, 'default', {}), /power of 2/);
assert.throws(() => instantiate({ generic: c.generic }, 'default', { extensionFactor: 8 }), /Extension factor/);
assert.throws(() => instantiate({ generic: Object.assign({}, c.generic, { modulus: '97' }) }, 'default', {}), /field of/);
const bad = JSON.parse(JSON.stringify(c.generic)); bad.transition.code[0] = 77;
assert.throws(() => instantiate({ generic: bad }, 'default', {}), /unknown opcode/);
const air = instantiate({ generic: c.generic }, 'default', {});
assert.throws(() => air.initProvingContext([], [1n, 2n]), /seed values/);
assert.throws(() => air.initProvingContext([[1n]], [1n, 2n, 3n, 4n]), /has 0 secret input registers/);
console.log('descriptor checks OK');
""")
    cj = tmp_path / 'case.json'
    cj.write_text(json.dumps(case))
    env = dict(os.environ, GSTARK_LIB=ORACLE_LIB, GSTARK_ALLOW_TEST_DOUBLE='1')
    r = subprocess.run([NODE, str(script), ROOT, str(cj)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'descriptor checks OK' in r.stdout, r.stderr[-2000:]
