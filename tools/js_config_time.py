"""tools/js_config_time.py — the BASELINE statements proved FROM NODE through the one-call native entry points (js/prover.js ->
napi/gstark_napi.node -> csrc/prover.cc): C2 (MiMC 2^13, E = 16), C3 (Rescue 2^16), C4 (Poseidon 2^16) — wall-clock per proof from
JavaScript beside the Python host's (tools/config_runs.py), and the proofs' sha256 (the same bytes: tests/golden/config_digests.json).
The AIRs of C3 / C4 travel as descriptors (GenericAir.descriptor(): what js/air_generic.js takes).
    python tools/js_config_time.py            (on the GPU box)"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genstark_amd.hostfield import HostField
from genstark_amd.poseidon import poseidon6x128_air
from genstark_amd.rescue import rescue4x128_air

f = HostField()
cases = []
t = 1 << 16
air = rescue4x128_air(t, 16, f, segmented=True)
seeds = [[42 + s, 43 + 2 * s] for s in range(t // 32)]
tr = air.hostTrace(seeds, steps=32)
cases.append({'name': 'C3', 'generic': air.descriptor(), 'extension_factor': 16, 'options': {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 68, 'friQueryCount': 24},
              'seed': [[str(v) for v in s] for s in seeds], 'assertions': [{'step': 31, 'register': 0, 'value': str(tr[31][0])}, {'step': t - 1, 'register': 1, 'value': str(rescue4x128_air(32, 16, f, segmented=True).hostTrace([seeds[-1]])[31][1])}]})   # (a chain's values do not depend on how many chains there are)
air = poseidon6x128_air(t, 16, f, segmented=True)
seeds = [[1 + s, 2, 3 + s, 4] for s in range(t // 64)]
cases.append({'name': 'C4', 'generic': air.descriptor(), 'extension_factor': 16, 'options': {'hashAlgorithm': 'blake2s256', 'extensionFactor': 16, 'exeQueryCount': 48, 'friQueryCount': 24},
              'seed': [[str(v) for v in s] for s in seeds], 'assertions': [{'step': 0, 'register': 0, 'value': '1'}, {'step': t - 64, 'register': 2, 'value': str(3 + t // 64 - 1)}]})
js = r"""
const path = require('path'), fs = require('fs'), crypto = require('crypto');
const ROOT = process.argv[2];
const { instantiate } = require(path.join(ROOT, 'js', 'shims', '@guildofweavers', 'air-assembly'));
const { proveGenericSerialized, proveMimcSerialized, packSeed } = require(path.join(ROOT, 'js', 'prover.js'));
const big = v => Array.isArray(v) ? v.map(big) : BigInt(v);
const ms = fn => { const t = process.hrtime.bigint(); fn(); return Number(process.hrtime.bigint() - t) / 1e6; };
const med = a => a.sort((x, y) => x - y)[a.length >> 1];
const out = [];
{   // C2: MiMC 2^13, E = 16, exe 48, fri 24 (examples/mimc/mimc128.ts)
    const o = { hashAlgorithm: 'blake2s256', extensionFactor: 16, exeQueryCount: 48, friQueryCount: 24 };
    const air = instantiate({ mimc: { steps: 1 << 13 } }, 'default', o);
    const tr = air.initProvingContext([], [3n]).generateExecutionTrace();
    const a = [{ step: 0, register: 0, value: tr.getValue(0, 0) }, { step: (1 << 13) - 1, register: 0, value: tr.getValue(0, (1 << 13) - 1) }];
    let p; for (let k = 0; k < 5; k++) p = proveMimcSerialized(air, o, a, 3n);
    const t = []; for (let k = 0; k < 30; k++) t.push(ms(() => { p = proveMimcSerialized(air, o, a, 3n); }));
    out.push({ name: 'C2_E16', prove_ms_from_node: med(t), proof_bytes: p.length, proof_sha256: crypto.createHash('sha256').update(p).digest('hex') });
}
for (const c of JSON.parse(fs.readFileSync(process.argv[3], 'utf8'))) {
    const air = instantiate({ generic: c.generic }, 'default', c.options);
    require(path.join(ROOT, 'js', 'galois.js')).native().call('gs_air_jit', air.field.ctx, 1);      // compiled AIR programs, as tools/config_runs.py runs them (Backend.jit())
    const a = c.assertions.map(x => ({ step: x.step, register: x.register, value: BigInt(x.value) })), seed = big(c.seed);
    let p; for (let k = 0; k < 5; k++) p = proveGenericSerialized(air, c.options, a, seed);
    const t = []; for (let k = 0; k < 20; k++) t.push(ms(() => { p = proveGenericSerialized(air, c.options, a, seed); }));
    const packed = packSeed(air, seed), tp = [];           // the first rows packed once (what tools/config_runs.py does through Prover.pack_seed)
    let q; for (let k = 0; k < 20; k++) tp.push(ms(() => { q = proveGenericSerialized(air, c.options, a, packed); }));
    if (Buffer.compare(p, q)) throw new Error('a packed seed gave other bytes');
    out.push({ name: c.name, prove_ms_from_node: med(t), prove_ms_from_node_packed_seed: med(tp), proof_bytes: p.length, proof_sha256: crypto.createHash('sha256').update(p).digest('hex') });
}
console.log(JSON.stringify(out));
"""
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, 'cases.json'), 'w').write(json.dumps(cases))
    open(os.path.join(d, 'run.js'), 'w').write(js)
    env = dict(os.environ)
    r = subprocess.run(['node', os.path.join(d, 'run.js'), ROOT, os.path.join(d, 'cases.json')], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout.strip() or r.stderr[-2000:])
    want = {x['name']: x for x in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'config_digests.json')))} if os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'config_digests.json')) else {}
    try:
        for rec in json.loads(r.stdout.strip().splitlines()[-1]):
            w = want.get(rec['name'])
            print(rec['name'], rec['prove_ms_from_node'], 'ms from node' + (f" ({rec['prove_ms_from_node_packed_seed']} with a packed seed)" if 'prove_ms_from_node_packed_seed' in rec else '') + ';', 'digest', 'as committed' if w and (w.get('proof_sha256') or w.get('sha256')) == rec['proof_sha256'] else ('not among the committed digests' if w else ''))
    except Exception as e:   # noqa: BLE001
        print('could not read the node side\'s answer:', e)
