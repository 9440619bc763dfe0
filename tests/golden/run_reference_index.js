// tests/golden/run_reference_index.js — the REFERENCE's own entry point (bin/index.js: instantiate(source, component, options), unmodified,
// loaded from the genSTARK checkout) handed AirAssembly SOURCE, on top of this repository's drop-in modules for
// @guildofweavers/{galois,merkle,air-assembly,air-script} (js/shims, found through NODE_PATH).  index.ts:29 calls the package's
// compile(); the shim's compile() asks this repository's loader (genstark_amd/airassembly.py) for the AIR's programs.
//   usage: NODE_PATH=<repo>/js/shims node run_reference_index.js /root/reference/bin <cases.json> <out.json>
// cases: [{name, source, component, options, seed: [decimal strings], inputs, assertions: [{step, register, value}]}]
const path = require('path');
const fs = require('fs');
const { instantiate } = require(path.join(process.argv[2], 'index.js'));
const cases = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
const big = v => Array.isArray(v) ? v.map(big) : BigInt(v);
const out = [];
for (const c of cases) {
    const stark = instantiate(Buffer.from(c.source), c.component, c.options, null);
    const assertions = c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) }));
    const proof = stark.prove(assertions, big(c.inputs || []), c.seed === undefined ? undefined : big(c.seed));
    const bytes = stark.serialize(proof);
    if (bytes.byteLength !== stark.sizeOf(proof)) throw new Error('size mismatch');
    const pub = c.publicInputs === undefined ? undefined : big(c.publicInputs);
    const ok = stark.verify(assertions, stark.parse(bytes), pub);
    let tamperRejected = false;
    try { const bad = Buffer.from(bytes); bad[40] ^= 1; stark.verify(assertions, stark.parse(bad), pub); } catch (e) { tamperRejected = true; }
    out.push({ name: c.name, proofHex: bytes.toString('hex'), verified: ok === true, tamperRejected, securityLevel: stark.securityLevel,
               friLayers: proof.ldProof.components.length, iShapes: proof.iShapes });
    console.log(c.name, bytes.byteLength, 'verified', ok, 'security', stark.securityLevel);
}
fs.writeFileSync(process.argv[4], JSON.stringify(out));
