'use strict';
// tests/js_assembly_onecall.js <cases.json> <out.json> — air-assembly components through the ONE-CALL native entry points from node
// (js/prover.js: proveAssemblySerialized / verifyAssemblySerialized -> napi -> gs_prover_prove_on / gs_prover_verify_on): statements
// with secret and public input registers — the proof carries the inputs' shapes, the native verifier sizes the trace from them.
// Per case: proof bytes (hex), the native verifier's verdict, and that a wrong public input, a flipped byte and a changed shape are
// refused.  tests/test_airassembly.py compares the bytes with the Python host's.
const fs = require('fs');
const path = require('path');
const assert = require('assert');
const ROOT = path.resolve(__dirname, '..');
const { compile, AssemblyAir } = require(path.join(ROOT, 'js', 'air_assembly.js'));
const { proveAssemblySerialized, verifyAssemblySerialized } = require(path.join(ROOT, 'js', 'prover.js'));
const big = x => Array.isArray(x) ? x.map(big) : BigInt(x);

const cases = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = [];
for (const c of cases) {
    const air = new AssemblyAir(compile(c.source), c.component, c.options);
    const options = Object.assign({ exeQueryCount: 80, friQueryCount: 40, hashAlgorithm: 'sha256' }, c.options);
    const assertions = c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) }));
    const inputs = big(c.inputs || []), seed = c.seed ? big(c.seed) : undefined, pub = big(c.publicInputs || []);
    const proof = proveAssemblySerialized(air, options, assertions, inputs, seed);
    assert.strictEqual(verifyAssemblySerialized(air, options, assertions, proof, pub), true);
    const rec = { name: c.name, proofHex: proof.toString('hex'), verified: true };
    const bad = Buffer.from(proof); bad[bad.length >> 1] ^= 0x10;
    assert.throws(() => verifyAssemblySerialized(air, options, assertions, bad, pub), /./);
    if (air.info.inputRegisters) {
        const shaped = Buffer.from(proof); shaped[shaped.length - 1] ^= 1;            // the last dimension of the last shape
        assert.throws(() => verifyAssemblySerialized(air, options, assertions, shaped, pub), /./);
        if (pub.length) {
            const wrong = JSON.parse(JSON.stringify(c.publicInputs)); let w = wrong; while (Array.isArray(w[0])) w = w[0]; w[0] = String(BigInt(w[0]) + 1n);
            assert.throws(() => verifyAssemblySerialized(air, options, assertions, proof, big(wrong)), /linear combination/);
            assert.throws(() => verifyAssemblySerialized(air, options, assertions, proof, []), /public input registers are needed/);
        }
    }
    rec.tamperRejected = true;
    out.push(rec);
}
fs.writeFileSync(process.argv[3], JSON.stringify(out));
console.log(`js one-call assembly OK: ${out.length} statements`);
