'use strict';
// js/air_generic.js — the AirModule / ProvingContext / VerificationContext surface of @guildofweavers/air-assembly that
// lib/Stark.ts and lib/components/*.ts consume, for AIRs given as register-machine programs (JS twin of
// genstark_amd/air_generic.py: GenericAir / GenericProvingContext / GenericVerificationContext).  The AIR arrives as a plain
// JSON descriptor — what `GenericAir.descriptor()` exports for the reference's Rescue 4x128 and Poseidon 6x128 examples
// (examples/rescue/hash4x128.ts:60-108, examples/poseidon/hash6x128.ts:62-87): field, trace shape, cyclic static registers and
// the transition / constraint / init programs in the {op, dst, a, b} encoding of include/gstark.h.  The device interprets the
// programs (gs_air_trace, gs_air_trace_segments, gs_air_constraints through the N-API shim); the verifier side interprets the
// constraint program on BigInt.
const { Matrix, Vector, native, le, packLe } = require('./galois');

const OP = { LOADC: 0, LOADR: 1, LOADN: 2, LOADS: 3, ADD: 4, SUB: 5, MUL: 6, POW: 7, POWC: 8, OUT: 9 };

class Program {
    constructor(desc) {
        if (!desc || !Array.isArray(desc.code) || desc.code.length % 4) throw new TypeError('program: code must hold 4 words per instruction');
        this.code = desc.code.map(Number);
        this.ninstr = this.code.length / 4;
        this.consts = (desc.consts || []).map(BigInt);
        this.nregs = desc.nregs; this.nout = desc.nout;
        for (let i = 0; i < this.ninstr; i++) {
            const op = this.code[4 * i];
            if (!(op >= 0 && op <= OP.OUT)) throw new TypeError(`program: unknown opcode ${op}`);
        }
    }
    constsBuffer() { return this.consts.length ? packLe(this.consts) : le(0n); }
    run(field, cur, next, statics) {   // host interpreter, as Program.run of genstark_amd/air_generic.py
        const vm = new Array(this.nregs).fill(0n), out = new Array(this.nout).fill(0n), c = this.code;
        for (let i = 0; i < c.length; i += 4) {
            const d = c[i + 1], a = c[i + 2], b = c[i + 3];
            switch (c[i]) {
                case OP.LOADC: vm[d] = this.consts[a]; break;
                case OP.LOADR: vm[d] = cur[a]; break;
                case OP.LOADN: vm[d] = next[a]; break;
                case OP.LOADS: vm[d] = statics[a]; break;
                case OP.ADD: vm[d] = field.add(vm[a], vm[b]); break;
                case OP.SUB: vm[d] = field.sub(vm[a], vm[b]); break;
                case OP.MUL: vm[d] = field.mul(vm[a], vm[b]); break;
                case OP.POW: vm[d] = field.exp(vm[a], BigInt(b)); break;
                case OP.POWC: vm[d] = field.exp(vm[a], this.consts[b]); break;
                default: out[d] = vm[a];
            }
        }
        return out;
    }
}

class Context {
    constructor(air) {
        const f = air.field;
        this.air = air; this.field = f;
        this.traceLength = air.steps; this.extensionFactor = air.extensionFactor;
        this.constraints = air.constraintDegrees.map(degree => ({ degree }));
        this.inputShapes = [];
        this.rootOfUnity = air.rootOfUnity;
        this.compositionFactor = air.compositionFactor;
    }
    staticPolys() {   // host coefficients of each cyclic register's polynomial K_s (degree < period): its value at x is K_s(x^(T/period))
        const air = this.air, f = this.field;
        if (!air._staticPolys) {
            air._staticPolys = air.staticRegisters.map(values => {
                const m = values.length;
                const g = f.exp(this.rootOfUnity, BigInt(this.extensionFactor * (this.traceLength / m)));
                return f.interpolateRoots(f.getPowerSeries(g, m), f.newVectorFrom(values)).toValues();
            });
        }
        return air._staticPolys;
    }
}

class ProvingContext extends Context {
    /** secretColumns: the SECRET registers' values for this proof (one BigInt array per register, length a power of two dividing the
     *  trace length: the column repeats), as the loader laid them out — they join the static registers of the trace and constraint
     *  programs after the public ones, their low-degree extensions are committed beside P(x) (lib/Stark.ts:113-114) */
    constructor(air, firstRows, secretColumns) {
        super(air);
        this.secretColumns = secretColumns || [];
        if (this.secretColumns.length !== air.secretInputCount) throw new Error(`the AIR has ${air.secretInputCount} secret input registers`);
        const f = this.field, n = this.traceLength * this.extensionFactor, nc = this.traceLength * this.compositionFactor;
        this.firstRows = firstRows;
        this.evaluationDomain = f.getPowerSeries(this.rootOfUnity, n);
        this.compositionDomain = f.getPowerSeries(f.exp(this.rootOfUnity, BigInt(n / nc)), nc);
        this.executionDomain = f.getPowerSeries(f.exp(this.rootOfUnity, BigInt(this.extensionFactor)), this.traceLength);
        // secret registers: the polynomial K_s through one period of the column (degree < period m, in the variable x^(T/m)); its values
        // over the evaluation domain are what is committed — they repeat with period m * E there
        // (every column is packed ONCE: the bytes go to the device for the interpolation and to the trace generator's static values)
        const columnPoly = (col, packed) => {           // the device vector of K's coefficients for one period `col` of a register
            const m = col.length;
            const g = f.exp(this.rootOfUnity, BigInt(this.extensionFactor * (this.traceLength / m)));
            const values = new Vector(f, m);
            native().call('gs_upload', f.ctx, values.ptr, packed, packed.length);
            return f.interpolateRoots(f.getPowerSeries(g, m), values);
        };
        this.packedSecret = this.secretColumns.map(col => packLe(col));
        const secretPolys = this.secretColumns.map((col, s) => {
            const m = col.length;
            if (!isPow2(m) || this.traceLength % m) throw new Error('a secret register column must be a power of 2 long and divide the trace length');
            return columnPoly(col, this.packedSecret[s]);
        });
        this.secretRegisterTraces = secretPolys.map((poly, s) => {
            const m = this.secretColumns[s].length, period = m * this.extensionFactor;
            const onePeriod = f.evalPolyAtRoots(poly, f.getPowerSeries(f.exp(this.rootOfUnity, BigInt(this.traceLength / m)), period));
            return period === n ? onePeriod : f.pluckVector(onePeriod, 1, n);            // v[i mod period]
        });
        // static registers over the composition domain: K_s at the (period * compositionFactor)-th roots of unity, back to back; public, then secret
        // The PUBLIC registers' tables are constants of the AIR: computed once and kept on the device with it (their coefficients never
        // come back to the host: that copy, staticPolys(), is the verifier's); the secret registers' are this proof's.  An AIR without
        // secret registers shares the finished table block between its proofs.
        const table = (m, poly) => {
            const ln = m * this.compositionFactor;
            const wk = f.exp(this.compositionDomain.seriesBase, BigInt(this.traceLength / m));
            return { ln, tab: f.evalPolyAtRoots(poly, f.getPowerSeries(wk, ln)) };
        };
        if (!air._packedStatic) air._packedStatic = air.staticRegisters.map(v => packLe(v));
        if (!air._publicTables) air._publicTables = air.staticRegisters.map((values, s) => table(values.length, columnPoly(values, air._packedStatic[s])));
        const all = air._publicTables.concat(secretPolys.map((poly, s) => table(this.secretColumns[s].length, poly)));
        this.staticLens = all.map(e => e.ln);
        if (!secretPolys.length && air._tableBlock) { this.staticTables = air._tableBlock; return; }
        const total = this.staticLens.reduce((a, b) => a + b, 0);
        this.staticTables = new Vector(f, Math.max(total, 1));
        let off = 0;
        all.forEach(e => {
            native().call('gs_copy', f.ctx, this.staticTables.ptr + BigInt(off * f.elementSize), e.tab.ptr, e.ln * f.elementSize);
            off += e.ln;
        });
        if (!secretPolys.length) { air._tableBlock = this.staticTables; air._tableLens = this.staticLens; }
    }
    allStaticColumns() { return this.air.staticRegisters.concat(this.secretColumns); }
    staticValuesPacked() {
        const parts = (this.air._packedStatic || this.air.staticRegisters.map(v => packLe(v))).concat(this.packedSecret);
        return parts.length ? Buffer.concat(parts) : le(0n);
    }
    generateExecutionTrace() {   // lib/Stark.ts:97
        const air = this.air, f = this.field, t = air.transitionProgram;
        const m = new Matrix(f, air.traceRegisterCount, this.traceLength);
        const periods = this.allStaticColumns().map(v => v.length);
        const first = this.firstRows.packedFirstRows || Buffer.concat(this.firstRows.map(row => packLe(row)));
        const nrows = this.firstRows.packedFirstRows ? this.firstRows.rows : this.firstRows.length;
        if (air.segmentLength === null) {
            native().call('gs_air_trace', f.ctx, t.code, t.ninstr, t.constsBuffer(), t.consts.length, t.nregs, air.traceRegisterCount,
                this.staticValuesPacked(), periods, periods.length, first, this.traceLength, m.ptr);
        } else {
            const init = air.initProgram;
            native().call('gs_air_trace_segments', f.ctx, t.code, t.ninstr, init ? init.code : [], init ? init.ninstr : 0, t.constsBuffer(), t.consts.length,
                t.nregs, air.traceRegisterCount, this.staticValuesPacked(), periods, periods.length, first, nrows, air.segmentLength, m.ptr);
        }
        return m;
    }
    generateStaticTrace() {
        const T = this.traceLength;
        return this.field.newMatrixFrom(this.allStaticColumns().map(v => { const row = new Array(T); for (let i = 0; i < T; i++) row[i] = v[i % v.length]; return row; }));
    }
    evaluateTransitionConstraints(pPolys) {   // CompositionPolynomial.ts:76
        const air = this.air, f = this.field, e = air.evaluationProgram, nc = this.compositionDomain.length;
        const pComp = f.evalPolysAtRoots(pPolys, this.compositionDomain);
        const q = new Matrix(f, air.constraintDegrees.length, nc);
        native().call('gs_air_constraints', f.ctx, e.code, e.ninstr, e.constsBuffer(), e.consts.length, e.nregs, air.traceRegisterCount,
            air.constraintDegrees.length, pComp.ptr, nc, nc / this.traceLength, this.staticTables.ptr, this.staticLens, this.staticLens.length, q.ptr);
        return q;
    }
}

class VerificationContext extends Context {
    evaluateConstraintsAt(x, rValues, nValues, hValues) {   // CompositionPolynomial.ts:153
        const f = this.field, polys = this.staticPolys();
        if (hValues.length !== this.air.secretInputCount) throw new Error('wrong number of secret register values');
        const statics = this.air.staticRegisters.map((values, s) => {
            const xc = f.exp(x, BigInt(this.traceLength / values.length)), poly = polys[s];
            let k = 0n;
            for (let i = poly.length - 1; i >= 0; i--) k = f.mod(k * xc + poly[i]);
            return k;
        });
        return this.air.evaluationProgram.run(f, rValues, nValues, statics.concat(hValues.map(v => f.mod(BigInt(v)))));      // public registers, then the secret ones (from the proof's leaves)
    }
}

const isPow2 = v => Number.isInteger(v) && v > 0 && (v & (v - 1)) === 0;

class GenericAir {
    /** desc: the object GenericAir.descriptor() of genstark_amd/air_generic.py exports (numbers beyond 2^53 as decimal strings). */
    constructor(desc, extensionFactor, field) {
        this.field = field;
        if (BigInt(desc.modulus) !== field.modulus) throw new TypeError(`the AIR is defined over the field of ${desc.modulus} elements`);
        this.steps = desc.steps; this.traceRegisterCount = desc.registers; this.secretInputCount = desc.secretInputCount || 0;
        // the secret registers' columns for ONE proof, when the descriptor was made for it (js/air_assembly.js: the loader's plan)
        this.secretColumns = desc.secretRegisters ? desc.secretRegisters.map(col => col.map(v => field.mod(BigInt(v)))) : null;
        if (!isPow2(this.steps) || this.steps < 2) throw new Error('steps must be a power of 2');
        this.constraintDegrees = desc.constraintDegrees.slice();
        this.maxConstraintDegree = Math.max(...this.constraintDegrees);
        this.compositionFactor = 1; while (this.compositionFactor < this.maxConstraintDegree) this.compositionFactor *= 2;
        this.extensionFactor = extensionFactor || desc.extensionFactor;
        const ef = this.extensionFactor;
        if (!isPow2(ef) || ef < 2 * this.compositionFactor || ef > 32) throw new Error('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32');
        this.staticRegisters = desc.staticRegisters.map(values => values.map(v => field.mod(BigInt(v))));
        for (const values of this.staticRegisters) if (!isPow2(values.length) || this.steps % values.length) throw new Error('static register cycles must be powers of 2 dividing the trace length');
        this.segmentLength = desc.segmentLength === undefined ? null : desc.segmentLength;
        if (this.segmentLength !== null && (!isPow2(this.segmentLength) || this.segmentLength < 2 || this.steps % this.segmentLength)) throw new Error('segment length must be a power of 2 dividing the trace length');
        this.transitionProgram = new Program(desc.transition);
        this.evaluationProgram = new Program(desc.evaluation);
        this.initProgram = desc.init ? new Program(desc.init) : null;
        if (this.transitionProgram.nout !== this.traceRegisterCount || this.evaluationProgram.nout !== this.constraintDegrees.length) throw new Error('transition must yield one value per register, evaluation one per constraint');
        this.seedWidth = desc.seedWidth;              // first row = the seed's values, zero-padded to the register count ...
        this.fixedFirstRows = desc.firstRows ? desc.firstRows.map(r => r.map(v => field.mod(BigInt(v)))) : null;   // ... unless the descriptor pins it
        this.rootOfUnity = field.getRootOfUnity(this.steps * ef);
    }
    /** the statement's first rows already in the driver's wire form, for many proofs of one seed (js/prover.js: packSeed) */
    packSeed(seed) { const rows = this.firstRows(seed); const flat = []; for (const row of rows) for (const v of row) flat.push(v); return { packedFirstRows: packLe(flat), rows: rows.length }; }
    firstRows(seed) {
        if (this.fixedFirstRows) return this.fixedFirstRows;
        if (seed && seed.packedFirstRows) return seed;            // a packed seed: nothing to lay out
        const pad = s => {
            if (!Array.isArray(s) || (this.seedWidth !== undefined && s.length !== this.seedWidth) || s.length > this.traceRegisterCount) throw new Error(`the AIR's first row takes ${this.seedWidth} seed values`);
            const row = s.map(v => this.field.mod(BigInt(v)));
            while (row.length < this.traceRegisterCount) row.push(0n);
            return row;
        };
        if (this.segmentLength === null) return [pad(seed || [])];
        const segments = this.steps / this.segmentLength;
        if (!Array.isArray(seed) || seed.length !== segments) throw new Error(`a segmented AIR needs one seed per segment (${segments})`);
        return seed.map(pad);
    }
    /** what the ONE-CALL driver needs of a proving context (js/prover.js: genericJob) — first rows, the static registers' packed values
     *  and their table block — without the domains' power series a member-by-member caller asks for: for an AIR without secret registers
     *  all of it but the first rows is a constant of the AIR, built by the first full context */
    jobContext(seed) {
        if (this.secretInputCount || !this._tableBlock) return this.initProvingContext([], seed);
        const air = this;
        return { air, firstRows: this.firstRows(seed), staticTables: this._tableBlock, staticLens: this._tableLens, secretRegisterTraces: [],
                 allStaticColumns() { return air.staticRegisters; },
                 staticValuesPacked() { return air._packedStatic.length ? Buffer.concat(air._packedStatic) : le(0n); } };
    }
    /** a proving context from first rows and secret columns that are already reduced field elements (the loader's plan: js/air_assembly.js) */
    contextFor(firstRows, secretColumns) { return new ProvingContext(this, firstRows, secretColumns || []); }
    initProvingContext(inputs, seed) {
        // inputs: one column per SECRET register (BigInt arrays), unless the descriptor already carries this proof's columns
        let cols = this.secretColumns;
        if (inputs && inputs.length) {
            if (inputs.length !== this.secretInputCount) throw new Error(`the AIR has ${this.secretInputCount} secret input registers`);
            cols = inputs.map(col => col.map(v => this.field.mod(BigInt(v))));
        }
        return new ProvingContext(this, this.firstRows(seed), cols || []);
    }
    initVerificationContext(inputShapes, publicInputs) { return new VerificationContext(this); }
}

module.exports = { GenericAir, Program };
