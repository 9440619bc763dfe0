'use strict';
// js/air_assembly.js — compile(source) / AirSchema / instantiate(schema, component, options) of `@guildofweavers/air-assembly` as
// index.ts:4,18-33 and lib/Stark.ts:40 use them, for AirAssembly SOURCE (the inline module of examples/mimc/mimc128Assembly.ts:28-51,
// assembly/*.aa).  The loader is js/aa_loader.js (JavaScript, in this process: no Python interpreter on the node side); it answers a
// request with a descriptor that this file hands to the register-machine AIR of js/air_generic.js, whose programs the device runs.
// Input registers (secret and public, nested shapes) are laid out by the loader when the inputs arrive: the plan it returns carries the
// public registers as static registers and the secret ones as this proof's columns.  GSTARK_AA_LOADER=python asks the Python loader
// (genstark_amd/airassembly.py through `python3 -m genstark_amd.aa_json`, a child process) instead: the two give the same answers
// object for object (tests/test_airassembly.py), the switch exists for that comparison.
const { spawnSync } = require('child_process');
const path = require('path');
const loader = require('./aa_loader');
const { GenericAir } = require('./air_generic');
const { defaultField } = require('./context');

const REPO = path.resolve(__dirname, '..');

// The loader's answers are pure functions of the request (source, component, extension factor, and for plan / verify the inputs): an
// AssemblyAir asks `info` and `describe` at construction and the same `verify` descriptor for every proof of one shape, so answers are
// kept (64 most recent, requests under 256 KB) instead of compiling the component's programs again per call.
const answers = new Map();
function ask(request) {
    if (process.env.GSTARK_AA_LOADER !== 'python' && (request.op === 'plan' || request.op === 'verify')) {
        // this proof's inputs: nothing to keep, nothing to turn into text — the loader takes the BigInts and answers in BigInts
        try { return loader.handle(request, true); } catch (e) { throw new Error(`AirAssembly: ${e.message}`); }
    }
    const key = JSON.stringify(request);
    if (key.length < (1 << 18) && answers.has(key)) {
        const hit = answers.get(key);
        answers.delete(key); answers.set(key, hit);           // most recently used last
        return hit;
    }
    let out;
    if (process.env.GSTARK_AA_LOADER === 'python') out = askLoader(key);
    else {
        try { out = loader.handle(request); } catch (e) { throw new Error(`AirAssembly: ${e.message}`); }
    }
    if (key.length < (1 << 18)) {
        answers.set(key, out);
        if (answers.size > 64) answers.delete(answers.keys().next().value);
    }
    return out;
}
function askLoader(requestJson) {
    const r = spawnSync(process.env.GSTARK_PYTHON || 'python3', ['-m', 'genstark_amd.aa_json'], { cwd: REPO, input: requestJson, encoding: 'utf8', maxBuffer: 1 << 28 });
    if (r.error) throw new Error(`AirAssembly loader (python3 -m genstark_amd.aa_json) could not be started: ${r.error.message}`);
    let out;
    try { out = JSON.parse(r.stdout.trim().split('\n').pop()); } catch (e) { throw new Error(`AirAssembly loader: ${r.stderr.slice(-400) || 'no answer'}`); }
    if (out.error) throw new Error(`AirAssembly: ${out.error}`);
    return out;
}
const toStrings = x => Array.isArray(x) ? x.map(toStrings) : String(x);

class AirSchema {
    /** source: AirAssembly text; parsed (and rejected, if malformed) by the loader at construction */
    constructor(source) {
        this.source = Buffer.isBuffer(source) ? source.toString('utf8') : String(source);
        const info = ask({ op: 'check', source: this.source });
        this.modulus = BigInt(info.modulus);
        this.exports = info.exports;
    }
}

function compile(source) {          // index.ts:29 — compileAirAssembly(source: Buffer | string): AirSchema
    if (typeof source === 'string' && !source.includes('(')) source = require('fs').readFileSync(source, 'utf8');      // a path to an .aa file
    return new AirSchema(source);
}

// The AirModule of a component.  Without input registers the trace shape is fixed and the inner AIR is built once; with (public) input
// registers it is sized when the inputs arrive (initProvingContext) or from the proof's shapes (initVerificationContext).
class AssemblyAir {
    constructor(schema, component, options) {
        const ex = schema.exports[component];
        if (!ex) throw new Error(`component ${component} is not exported (exports: ${Object.keys(schema.exports).sort().join(', ')})`);
        this.schema = schema; this.component = component;
        this.field = defaultField(schema.modulus);
        this._ef = options && options.extensionFactor;
        // counts, degrees and the extension factor do not depend on the inputs' shape (lib/Stark.ts:40-75 reads them at construction)
        const info = ask(this._req('info'));
        this.traceRegisterCount = info.traceRegisterCount; this.secretInputCount = info.secretInputCount;
        this.constraintDegrees = info.constraintDegrees; this.maxConstraintDegree = info.maxConstraintDegree; this.extensionFactor = info.extensionFactor;
        this.constraints = info.constraintDegrees.map(degree => ({ degree }));
        this._inner = info.inputRegisters ? null : this._build(ask(this._req('describe')).descriptor);
        this.info = info;          // (with input registers: declarations, static sources, cycles, the shape-independent evaluator — js/prover.js)
    }
    /** the register-machine AIR of a component without input registers (what js/prover.js: proveGenericSerialized takes) */
    get generic() { if (!this._inner) throw new Error('the component has input registers: its AIR is built when the inputs arrive'); return this._inner; }
    _req(op, more) { return Object.assign({ op, source: this.schema.source, component: this.component, extensionFactor: this._ef || null }, more || {}); }
    _build(desc) { return new GenericAir(desc, this.extensionFactor, this.field); }
    initProvingContext(inputs, seed) {
        if (this._inner) return this._inner.initProvingContext(inputs, seed);
        const text = process.env.GSTARK_AA_LOADER === 'python' ? toStrings : (x => x);          // (JSON to the child process; the loader in this process takes BigInt)
        const plan = ask(this._req('plan', { inputs: text(inputs || []), seed: seed === undefined || seed === null ? null : text(seed) }));
        let ctx;
        if (plan.innerAir) {
            // the loader in this process: the AIR of these public columns is built once (programs, static registers, their tables on the
            // device) and found again by the loader's own object for it; the proof's part is the first rows and the secret columns
            if (!this._generic) this._generic = new WeakMap();
            let g = this._generic.get(plan.innerAir);
            if (!g) {
                const d = Object.assign({}, plan.descriptor);
                delete d.firstRows; delete d.secretRegisters;
                d.seedWidth = d.registers;
                g = this._build(d);
                this._generic.set(plan.innerAir, g);
            }
            const p = this.field.modulus, red = v => (v >= 0n && v < p ? v : ((v % p) + p) % p);
            ctx = g.contextFor(plan.firstRows.map(row => row.map(red)), plan.secretColumns);
        } else ctx = this._build(plan.descriptor).initProvingContext([], undefined);
        ctx.inputShapes = plan.inputShapes;
        return ctx;
    }
    initVerificationContext(inputShapes, publicInputs) {
        if (this._inner) return this._inner.initVerificationContext(inputShapes, publicInputs);
        const text = process.env.GSTARK_AA_LOADER === 'python' ? toStrings : (x => x);
        const d = ask(this._req('verify', { inputShapes: inputShapes || [], publicInputs: text(publicInputs || []) })).descriptor;
        const ctx = this._build(d).initVerificationContext(inputShapes, publicInputs);
        ctx.inputShapes = inputShapes || [];
        return ctx;
    }
}

module.exports = { AirSchema, compile, AssemblyAir };
