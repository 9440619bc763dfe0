'use strict';
// js/smoke_generic.js <cases.json> <out.json> — the AIR-side members lib/Stark.ts calls (generateExecutionTrace :97,
// evaluateTransitionConstraints via CompositionPolynomial.ts:76, evaluateConstraintsAt :153) for AIRs given as descriptors
// (js/air_generic.js), run from node through the N-API shim.  Writes digests of what each member returned; the caller
// (tests/test_napi_addon.py) compares them with the same members of the Python host on the same library.
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { instantiate } = require(path.join(__dirname, 'shims', '@guildofweavers', 'air-assembly'));
const sha = b => crypto.createHash('sha256').update(b).digest('hex');
const big = v => Array.isArray(v) ? v.map(big) : BigInt(v);
const out = [];
for (const c of JSON.parse(fs.readFileSync(process.argv[2], 'utf8'))) {
    const air = instantiate({ generic: c.generic }, 'default', { extensionFactor: c.extension_factor });
    const f = air.field, ctx = air.initProvingContext([], big(c.seed));
    const trace = ctx.generateExecutionTrace();
    const pPolys = f.interpolateRoots(ctx.executionDomain, trace);
    const q = ctx.evaluateTransitionConstraints(pPolys);
    const statics = ctx.generateStaticTrace();
    // the verifier's side: constraints at one out-of-trace point of the evaluation domain, from P's values there
    const pEv = f.evalPolysAtRoots(pPolys, ctx.evaluationDomain);
    const pos = 8, n = ctx.evaluationDomain.length, ef = ctx.extensionFactor;
    const x = ctx.evaluationDomain.getValue(pos);
    const r = [], nx = [];
    for (let i = 0; i < air.traceRegisterCount; i++) { r.push(pEv.getValue(i, pos)); nx.push(pEv.getValue(i, (pos + ef) % n)); }
    const at = air.initVerificationContext([], []).evaluateConstraintsAt(x, r, nx, []);
    // ONE call of the native driver (js/prover.js -> N-API -> csrc/prover.cc) for the whole proof
    const { proveGenericSerialized, packSeed } = require(path.join(__dirname, 'prover.js'));
    const options = { hashAlgorithm: c.hash_algorithm, extensionFactor: c.extension_factor, exeQueryCount: c.exe_query_count, friQueryCount: c.fri_query_count };
    const proof = proveGenericSerialized(air, options, c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) })), big(c.seed));
    // the same statement from a seed packed once (packSeed: the first rows in the driver's wire form), and once more through the AIR's cached
    // job context: the same bytes
    const again = proveGenericSerialized(air, options, c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) })), packSeed(air, big(c.seed)));
    if (Buffer.compare(proof, again)) throw new Error(`${c.name}: a packed seed gave other proof bytes`);
    out.push({ name: c.name, proofSize: proof.length, proofSha256: sha(proof), trace: sha(trace.toBuffer()), constraints: sha(q.toBuffer()), statics: sha(statics.toBuffer()),
               constraintsAt: at.map(String), rows: [trace.rowCount, q.rowCount, statics.rowCount], cols: [trace.colCount, q.colCount] });
}
fs.writeFileSync(process.argv[3], JSON.stringify(out));
console.log(`generic AIR members via N-API: ${out.length} cases`);
