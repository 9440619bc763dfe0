// tests/golden/run_reference_stark.js — drives the REFERENCE's own compiled prover/verifier (bin/lib/Stark.js and
// bin/lib/components/*.js, unmodified, loaded from the genSTARK checkout) on top of this repository's drop-in modules
// for @guildofweavers/{galois,merkle,air-assembly} (js/shims, found through NODE_PATH).
//   usage: NODE_PATH=<repo>/js/shims node run_reference_stark.js /root/reference/bin/lib <cases.json> <out.json>
// Every line of prove()/verify()/serialize() orchestration executed here is the reference's; only the arithmetic
// objects it is handed are ours.
const path = require('path');
const fs = require('fs');
const libDir = process.argv[2];
const { Stark } = require(path.join(libDir, 'Stark.js'));
const cases = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
// the native driver (C++ above the C ABI) behind the same addon: its bytes must equal what the reference's Stark.js produces
const repoJs = path.join(__dirname, '..', '..', 'js');
const { MimcAir } = require(path.join(repoJs, 'air_mimc.js'));
const { proveMimcSerialized, proveGenericSerialized, verifyMimcSerialized, verifyGenericSerialized } = require(path.join(repoJs, 'prover.js'));
const out = [];
const noopLogger = { start() { return () => {}; }, sub() { return () => {}; }, done() {} };
for (const c of cases) {
    const options = { hashAlgorithm: c.hash_algorithm, extensionFactor: c.extension_factor, exeQueryCount: c.exe_query_count, friQueryCount: c.fri_query_count };
    // c.modulus (optional, decimal string): a field other than the 128-bit one — one field per process (js/galois.js)
    // c.generic (optional): an AIR given as register-machine programs (GenericAir.descriptor() of genstark_amd/air_generic.py —
    // the reference's Rescue / Poseidon examples); c.seed is then the (possibly nested) list of seed values as decimal strings
    const wide = c.modulus !== undefined, generic = c.generic !== undefined;
    const schema = generic ? { generic: c.generic } : { mimc: wide ? { steps: c.steps, modulus: BigInt(c.modulus) } : { steps: c.steps } };
    const stark = new Stark(schema, generic ? 'default' : 'mimc', options, noopLogger);
    const assertions = c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) }));
    const big = v => Array.isArray(v) ? v.map(big) : BigInt(v);
    const proof = stark.prove(assertions, [], generic ? big(c.seed) : [BigInt(c.seed)]);
    const bytes = stark.serialize(proof);
    if (bytes.byteLength !== stark.sizeOf(proof)) throw new Error('size mismatch');
    const ok = stark.verify(assertions, stark.parse(bytes));
    let tamperRejected = false;
    try { const bad = Buffer.from(bytes); bad[40] ^= 1; stark.verify(assertions, stark.parse(bad)); } catch (e) { tamperRejected = true; }
    // (init programs extend the transition program's pool in place: pools of a descriptor with an init program cannot be concatenated blindly,
    //  proveGenericSerialized checks)
    // ONE call of the native driver through the same addon: the build of libgstark_prover*.so for the loaded library's field (js/prover.js)
    const nativeBytes = generic ? proveGenericSerialized(stark.air, options, assertions, big(c.seed)) : proveMimcSerialized(wide ? stark.air : new MimcAir(c.steps, c.extension_factor), options, assertions, BigInt(c.seed));
    // ... and the native verifier through the same addon: accepts the reference's bytes, rejects the tampered ones with a message
    const vAir = generic ? stark.air : (wide ? stark.air : new MimcAir(c.steps, c.extension_factor));
    const nativeVerify = generic ? verifyGenericSerialized : verifyMimcSerialized;
    const nativeVerified = nativeVerify(vAir, options, assertions, bytes) === true;
    let nativeTamperRejected = false;
    try { const bad = Buffer.from(bytes); bad[40] ^= 1; nativeVerify(vAir, options, assertions, bad); } catch (e) { nativeTamperRejected = true; }
    out.push({ name: c.name, nativeVerified, nativeTamperRejected, nativeDriverEqualsReference: Buffer.from(bytes).equals(nativeBytes), proofHex: bytes.toString('hex'), evRoot: proof.evRoot.toString('hex'), lcRoot: proof.ldProof.lcRoot.toString('hex'),
               friLayers: proof.ldProof.components.length, remainderLength: proof.ldProof.remainder.length, verified: ok === true,
               tamperRejected, securityLevel: stark.securityLevel });
    console.log(c.name, bytes.byteLength, 'verified', ok, 'security', stark.securityLevel);
}
fs.writeFileSync(process.argv[4], JSON.stringify(out));
