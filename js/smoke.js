'use strict';
// js/smoke.js — exercises the galois / merkle drop-in objects from node through the N-API shim on the GPU:
// NTT round trip, LDE vs direct evaluation, hashing vs node crypto, Merkle batch proof, FRI fold consistency.
const assert = require('assert');
const crypto = require('crypto');
const { createPrimeField, MODULUS } = require('./galois');
const { createHash, MerkleTree } = require('./merkle');

const field = createPrimeField(MODULUS);
const P = MODULUS;
const n = 1 << 14, T = n / 16;
const w = field.getRootOfUnity(n);
const domain = field.getPowerSeries(w, n);
const coeffs = field.getPowerSeries(123456789123456789n, T);
const ev = field.evalPolyAtRoots(coeffs, domain);
// spot values by direct evaluation
const cv = coeffs.toValues();
for (const q of [0, 1, 77, n - 1]) {
    const x = field.exp(w, BigInt(q));
    let s = 0n;
    for (let i = cv.length - 1; i >= 0; i--) s = (s * x + cv[i]) % P;
    assert.strictEqual(ev.getValue(q), s);
}
const back = field.interpolateRoots(domain, ev).toValues();
assert.deepStrictEqual(back.slice(0, T), cv);
assert(back.slice(T).every(v => v === 0n));
// pointwise + batch inverse
const inv = field.invVectorElements(ev), one = field.mulVectorElements(ev, inv).toValues();
assert(one.every((v, i) => v === 1n || ev.getValue(i) === 0n));
// hashing vs node crypto
for (const alg of ['sha256', 'blake2s256']) {
    const h = createHash(alg, field);
    const leaves = h.mergeVectorRows([ev]);
    const raw = ev.toBuffer(), dg = leaves.toBuffer();
    for (const i of [0, 5, n - 1]) assert(dg.slice(32 * i, 32 * i + 32).equals(crypto.createHash(alg).update(raw.slice(16 * i, 16 * i + 16)).digest()));
    const tree = MerkleTree.create(leaves, h);
    const idx = [3, 4, 900, n - 1, 17];
    const proof = tree.proveBatch(idx);
    assert(MerkleTree.verifyBatch(tree.root, idx, proof, h));
    proof.values[0] = Buffer.alloc(32);
    assert(!MerkleTree.verifyBatch(tree.root, idx, proof, h));
}
// FRI row polynomials: domain fast path == generic path
const ys = field.transposeVector(ev, 4), xs = field.transposeVector(domain, 4);
const fast = field.interpolateQuarticBatch(xs, ys).toBuffer();
const generic = field.interpolateQuarticBatch(field.newMatrixFrom(xs.toValues()), ys).toBuffer();
assert(fast.equals(generic));
// polynomial members: NTT-based product vs schoolbook, zero-extending sum/difference
{
    const a = field.getPowerSeries(987654321987654321n, 150).toValues(), b = field.getPowerSeries(31337n, 90).toValues();
    const want = new Array(a.length + b.length - 1).fill(0n);
    for (let i = 0; i < a.length; i++) for (let j = 0; j < b.length; j++) want[i + j] = (want[i + j] + a[i] * b[j]) % P;
    const got = field.mulPolys(field.newVectorFrom(a), field.newVectorFrom(b));
    assert.strictEqual(got.length, want.length);
    assert.deepStrictEqual(got.toValues(), want);
    const pad = b.concat(new Array(a.length - b.length).fill(0n));
    assert.deepStrictEqual(field.addPolys(field.newVectorFrom(a), field.newVectorFrom(b)).toValues(), a.map((x, i) => (x + pad[i]) % P));
    assert.deepStrictEqual(field.subPolys(field.newVectorFrom(b), field.newVectorFrom(a)).toValues(), a.map((x, i) => (pad[i] - x + P) % P));
    assert.deepStrictEqual(field.mulPolyByConstant(field.newVectorFrom(b), 5n).toValues(), b.map(x => x * 5n % P));
    const mat = [[1n, 2n, 3n], [P - 1n, 5n, 7n]], vec = [11n, P - 2n, 13n];
    assert.deepStrictEqual(field.mulMatrixByVector(field.newMatrixFrom(mat), field.newVectorFrom(vec)).toValues(), mat.map(row => row.reduce((s, x, i) => (s + x * vec[i]) % P, 0n)));
}
// one native call for a whole proof (js/prover.js -> N-API -> csrc/prover.cc): the golden MiMC proofs, byte for byte
{
    const fs = require('fs'), path = require('path');
    const { MimcAir } = require('./air_mimc');
    const { proveMimcSerialized } = require('./prover');
    const golden = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'oracle_proofs.json'), 'utf8'));
    for (const c of golden.slice(0, 3)) {
        const options = { hashAlgorithm: c.hash_algorithm, extensionFactor: c.extension_factor, exeQueryCount: c.exe_query_count, friQueryCount: c.fri_query_count };
        const assertions = c.assertions.map(a => ({ step: a.step, register: a.register, value: BigInt(a.value) }));
        const bytes = proveMimcSerialized(new MimcAir(c.steps, c.extension_factor, field), options, assertions, BigInt(c.seed));
        assert.strictEqual(bytes.length, c.proofSize);
        assert.strictEqual(crypto.createHash('sha256').update(bytes).digest('hex'), c.proofSha256);
    }
    assert.throws(() => proveMimcSerialized(new MimcAir(64, 16, field), { hashAlgorithm: 'blake2s256', extensionFactor: 16, exeQueryCount: 48, friQueryCount: 24 },
                                          [{ step: 0, register: 0, value: 4n }], 3n), /conflicts with execution trace/);
}
console.log(`js smoke OK: galois/merkle drop-in objects via N-API on backend ${process.env.GSTARK_ALLOW_TEST_DOUBLE === '1' ? '(test double allowed)' : 'hip-gfx950'}`);
