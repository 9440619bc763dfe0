'use strict';
// js/air_mimc.js — the AirModule / ProvingContext / VerificationContext surface of @guildofweavers/air-assembly that
// lib/Stark.ts and lib/components/*.ts consume, instantiated for the MiMC AIR of examples/mimc/mimc128Assembly.ts:28-51
// (JS twin of genstark_amd/air.py; the AirAssembly compiler itself is out of scope).
const crypto = require('crypto');
const { createPrimeField, MODULUS, Matrix, native, le } = require('./galois');

function sha256Prng(seed, count, field) {  // UNVERIFIED restatement of air-assembly prng.sha256 (see genstark_amd/air.py)
    const out = [];
    for (let i = 0; i < count; i++) {
        const head = Buffer.alloc(2); head.writeUInt16BE(i + 1, 0);
        out.push(field.mod(BigInt('0x' + crypto.createHash('sha256').update(Buffer.concat([head, seed])).digest().toString('hex'))));
    }
    return out;
}

class Context {
    constructor(air) {
        const f = air.field;
        this.air = air; this.field = f;
        this.traceLength = air.steps; this.extensionFactor = air.extensionFactor;
        this.constraints = [{ degree: 3 }];
        this.inputShapes = [];
        const n = this.traceLength * this.extensionFactor;
        if (air._root === undefined) air._root = f.getRootOfUnity(n);
        this.rootOfUnity = air._root;
        this.compositionFactor = 4;
        const nk = air.roundConstants.length;
        this.cycleCount = this.traceLength / nk;
        if (air._kPoly === undefined) {
            const g = f.exp(this.rootOfUnity, BigInt(this.extensionFactor * this.cycleCount));
            const ginv = f.inv(g), ninv = f.inv(BigInt(nk));
            const pw = [1n]; for (let i = 1; i < nk; i++) pw.push(f.mul(pw[i - 1], ginv));
            air._kPoly = [];
            for (let j = 0; j < nk; j++) {
                let s = 0n;
                for (let i = 0; i < nk; i++) s += air.roundConstants[i] * pw[(i * j) % nk];
                air._kPoly.push(f.mul(f.mod(s), ninv));
            }
        }
        this.kPoly = air._kPoly;
    }
}

class ProvingContext extends Context {
    constructor(air, seed) {
        super(air);
        const f = this.field, n = this.traceLength * this.extensionFactor, nc = this.traceLength * this.compositionFactor;
        this.seed = f.mod(seed);
        this.evaluationDomain = f.getPowerSeries(this.rootOfUnity, n);
        this.compositionDomain = f.getPowerSeries(f.exp(this.rootOfUnity, BigInt(n / nc)), nc);
        this.executionDomain = f.getPowerSeries(f.exp(this.rootOfUnity, BigInt(this.extensionFactor)), this.traceLength);
        this.secretRegisterTraces = [];
        this.kTable = air.kTableOnDevice(this);
    }
    generateExecutionTrace() {
        const f = this.field, m = new Matrix(f, 1, this.traceLength), rc = this.air.roundConstants;
        native().call('gs_mimc_trace', f.ctx, le(this.seed), Buffer.concat(rc.map(le)), rc.length, this.traceLength, m.ptr);
        return m;
    }
    generateStaticTrace() {
        const rc = this.air.roundConstants, row = [];
        for (let i = 0; i < this.traceLength; i++) row.push(rc[i % rc.length]);
        return this.field.newMatrixFrom([row]);
    }
    evaluateTransitionConstraints(pPolys) {
        const f = this.field, nc = this.compositionDomain.length;
        const pComp = f.evalPolysAtRoots(pPolys, this.compositionDomain);
        const q = new Matrix(f, 1, nc);
        native().call('gs_mimc_constraints', f.ctx, pComp.ptr, nc, nc / this.traceLength, this.kTable.ptr, this.kTable.length, q.ptr);
        return q;
    }
}

class VerificationContext extends Context {
    evaluateConstraintsAt(x, rValues, nValues, hValues) {
        const f = this.field, xc = f.exp(x, BigInt(this.cycleCount));
        let k = 0n;
        for (let i = this.kPoly.length - 1; i >= 0; i--) k = f.mod(k * xc + this.kPoly[i]);
        return [f.sub(nValues[0], f.add(f.exp(rValues[0], 3n), k))];
    }
}

class MimcAir {
    constructor(steps, extensionFactor, field) {
        this.field = field || createPrimeField(MODULUS);
        this.steps = steps; this.extensionFactor = extensionFactor || 8;
        this.maxConstraintDegree = 3; this.traceRegisterCount = 1; this.secretInputCount = 0;
        this.roundConstants = sha256Prng(Buffer.from('4d694d43', 'hex'), 64, this.field);
    }
    /** the cyclic register over the composition domain: a constant of the AIR, kept on the device with it */
    kTableOnDevice(context) {
        if (!this._kTable) {
            const f = this.field, nc = context.traceLength * context.compositionFactor, n = context.traceLength * context.extensionFactor;
            const klen = this.roundConstants.length * context.compositionFactor;
            const wk = f.exp(f.exp(context.rootOfUnity, BigInt(n / nc)), BigInt(context.cycleCount));
            this._kTable = f.evalPolyAtRoots(f.newVectorFrom(context.kPoly), f.getPowerSeries(wk, klen));
        }
        return this._kTable;
    }
    /** what the ONE-CALL driver needs of a proving context (js/prover.js: proveMimcSerialized): the root of unity and the table — no
     *  power series of the domains */
    jobContext() { const c = new Context(this); c.kTable = this.kTableOnDevice(c); return c; }
    initProvingContext(inputs, seed) { return new ProvingContext(this, seed[0]); }
    initVerificationContext(inputShapes, publicInputs) { return new VerificationContext(this); }
}

module.exports = { MimcAir, sha256Prng };
