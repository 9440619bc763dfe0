const fs = require('fs'), path = require('path');
const ROOT = require('path').resolve(__dirname, '..');
const A = require(path.join(ROOT, 'js', 'air_assembly.js'));
const loader = require(path.join(ROOT, 'js', 'aa_loader.js'));
const { GenericAir } = require(path.join(ROOT, 'js', 'air_generic.js'));
const runs = 4096;
const options = { hashAlgorithm: 'sha256', exeQueryCount: 24, friQueryCount: 12 };
const src = fs.readFileSync(path.join(ROOT, 'tests', 'golden', 'aa', 'ledger.aa'), 'utf8');
const air = new A.AssemblyAir(A.compile(src), 'default', options);
const balances = [], factors = [], deposits = [];
for (let i = 0; i < runs; i++) { balances.push(BigInt(100 + 7 * i)); factors.push(BigInt(3 + i)); deposits.push([0, 1, 2, 3].map(j => BigInt(5 + i + 2 * j))); }
const inputs = [balances, factors, deposits];
const T = {}; const tm = (k, fn) => { const t = process.hrtime.bigint(); const r = fn(); T[k] = (T[k] || 0) + Number(process.hrtime.bigint() - t) / 1e6; return r; };
for (let rep = 0; rep < 4; rep++) {
  for (const k in T) delete T[k];
  const plan = tm('plan', () => loader.handle(air._req('plan', { inputs, seed: null }), true));
  const g = tm('GenericAir', () => new GenericAir(plan.descriptor, air.extensionFactor, air.field));
  const ctx = tm('ProvingContext', () => g.initProvingContext([], undefined));
  tm('staticValuesPacked', () => ctx.staticValuesPacked());
  console.log(JSON.stringify(T));
}
// (scratch profiler of tools/js_shaped_time.js's host side: plan / GenericAir / ProvingContext / packing, ms per proof)
