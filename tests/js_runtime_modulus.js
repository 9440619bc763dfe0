'use strict';
// tests/js_runtime_modulus.js <modulus> — createPrimeField(modulus) (index.ts:14) for a modulus no fixed build knows: js/galois.js loads the
// runtime-modulus library and hands it the modulus (gs_set_modulus); vector product and NTT against BigInt arithmetic.
const path = require('path');
const assert = require('assert');
const { createPrimeField } = require(path.join(__dirname, '..', 'js', 'galois.js'));
const q = BigInt(process.argv[2]);
const f = createPrimeField(q);
assert(f.modulus === q && f.elementSize === 32);
const a = f.newVectorFrom([1n, 2n, q - 1n, 12345n]), b = f.newVectorFrom([5n, q - 2n, q - 1n, 99999n]);
const m = f.mulVectorElements(a, b).toValues();
assert.strictEqual(m.map(String).join(','), [5n, (2n * (q - 2n)) % q, 1n, (12345n * 99999n) % q].map(String).join(','));
const w = f.getRootOfUnity(8);
const ev = f.evalPolyAtRoots(f.newVectorFrom([1n, 2n, 3n]), f.getPowerSeries(w, 8)).toValues();
let ok = true;
for (let i = 0; i < 8; i++) { const x = f.exp(w, BigInt(i)); ok = ok && ev[i] === (1n + 2n * x + 3n * x * x) % q; }
assert(ok);
assert.throws(() => createPrimeField(q + 2n), /./);          // one field per process
console.log(`js runtime modulus OK: ${q}`);
