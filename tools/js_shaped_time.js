'use strict';
// tools/js_shaped_time.js [runs=4096] — the ledger statement of tools/config_runs.py: X_shaped (tests/golden/aa/ledger.aa: 2 secret + 1 public
// input register, 4 096 runs = 2^15 steps) proved and verified FROM NODE through the one-call native entry points (js/prover.js):
// wall-clock per proof and per verification, and the proof's sha256 — the same bytes as the Python host's (tests/golden/config_digests.json).
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const ROOT = path.resolve(__dirname, '..');
const { compile, AssemblyAir } = require(path.join(ROOT, 'js', 'air_assembly.js'));
const { proveAssemblySerialized, verifyAssemblySerialized } = require(path.join(ROOT, 'js', 'prover.js'));
const runs = parseInt(process.argv[2] || '4096', 10);
const options = { hashAlgorithm: 'sha256', exeQueryCount: 24, friQueryCount: 12 };
const air = new AssemblyAir(compile(fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'aa', 'ledger.aa'), 'utf8')), 'default', options);
const balances = [], factors = [], deposits = [];
for (let i = 0; i < runs; i++) { balances.push(BigInt(100 + 7 * i)); factors.push(BigInt(3 + i)); deposits.push([0, 1, 2, 3].map(j => BigInt(5 + i + 2 * j))); }
const inputs = [balances, factors, deposits];
// the asserted cells: step 0 of register 0 (the first balance), and the last cell of register 2 from a trace of this statement
const ctx = air.initProvingContext(inputs, undefined);
const trace = ctx.generateExecutionTrace();
const last = 8 * runs - 1;
const assertions = [{ step: 0, register: 0, value: balances[0] }, { step: last, register: 2, value: trace.getValue(2, last) }];
let proof;
for (let k = 0; k < 3; k++) proof = proveAssemblySerialized(air, options, assertions, inputs, undefined);
const ms = fn => { const t = process.hrtime.bigint(); fn(); return Number(process.hrtime.bigint() - t) / 1e6; };
const prove = [], verify = [];
for (let k = 0; k < 10; k++) prove.push(ms(() => { proof = proveAssemblySerialized(air, options, assertions, inputs, undefined); }));
for (let k = 0; k < 5; k++) verify.push(ms(() => { if (verifyAssemblySerialized(air, options, assertions, proof, [deposits]) !== true) throw new Error('refused'); }));
prove.sort((a, b) => a - b); verify.sort((a, b) => a - b);
console.log(JSON.stringify({ runs, steps: 8 * runs, prove_ms: prove[5], prove_ms_min: prove[0], verify_ms: verify[2], proof_bytes: proof.length,
                             proof_sha256: crypto.createHash('sha256').update(proof).digest('hex') }));
