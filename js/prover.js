'use strict';
// js/prover.js — ONE native call for a whole proof: Stark.prove() + Serializer.serializeProof() through the
// native driver (include/gstark_prover.h, genstark_amd/csrc/prover.cc), reached through the same N-API addon that carries the
// member-by-member galois / merkle surface.  The returned Buffer is what lib/Serializer.ts:83-144 (`stark.parse`) reads.
const path = require('path');
const { native, le } = require('./galois');

const HASH_ALG = { sha256: 0, blake2s256: 1 };

function proveMimcSerialized(air, options, assertions, seed) {
    if (!(options.hashAlgorithm in HASH_ALG)) throw new TypeError(`Hash algorithm ${options.hashAlgorithm} is not supported`);
    const context = air.initProvingContext([], [seed]);      // the cyclic register's table over the composition domain, root of unity
    const f = air.field;
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(context.rootOfUnity), seed: le(f.mod(seed)),
        roundConstants: Buffer.concat(air.roundConstants.map(le)), kTable: context.kTable.ptr, kLen: context.kTable.length,
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
    };
    const lib = process.env.GSTARK_PROVER_LIB || path.join(__dirname, '..', 'genstark_amd', 'csrc', 'libgstark_prover.so');
    return native().proveMimcSerialized(f.ctx, lib, job);
}

// ... and for an AIR given as register-machine programs (js/air_generic.js: the reference's Rescue / Poseidon examples): kind 1 of
// gs_prover_air.  The two programs keep separate constant pools; the driver takes one, so the evaluator's constant indexes are rebased.
function proveGenericSerialized(air, options, assertions, seed) {
    if (!(options.hashAlgorithm in HASH_ALG)) throw new TypeError(`Hash algorithm ${options.hashAlgorithm} is not supported`);
    const f = air.field, context = air.initProvingContext([], seed);
    const t = air.transitionProgram, e = air.evaluationProgram, init = air.initProgram;
    if (init && init.consts.length > t.consts.length) throw new Error('the init program extends the transition program\'s constant pool');
    const base = t.consts.length, eCode = e.code.slice();
    for (let i = 0; i < eCode.length; i += 4) {
        if (eCode[i] === 0) eCode[i + 2] += base;            // LOADC: constant index
        else if (eCode[i] === 8) eCode[i + 3] += base;       // POWC: exponent index
    }
    const pool = t.consts.concat(e.consts);
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(air.rootOfUnity),
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
        registers: air.traceRegisterCount, degrees: air.constraintDegrees, tCode: t.code, iCode: init ? init.code : [], eCode,
        consts: pool.length ? Buffer.concat(pool.map(le)) : Buffer.alloc(0), vmRegs: Math.max(t.nregs, e.nregs, init ? init.nregs : 0),
        staticValues: context.staticValuesPacked(), staticPeriods: air.staticRegisters.map(v => v.length), staticTables: context.staticTables.ptr,
        staticLens: context.staticLens, firstRows: Buffer.concat(context.firstRows.map(row => Buffer.concat(row.map(le)))),
        segments: air.segmentLength === null ? 0 : context.firstRows.length, segmentLen: air.segmentLength === null ? 0 : air.segmentLength,
    };
    const lib = process.env.GSTARK_PROVER_LIB || path.join(__dirname, '..', 'genstark_amd', 'csrc', 'libgstark_prover.so');
    return native().proveGenericSerialized(f.ctx, lib, job);
}

module.exports = { proveMimcSerialized, proveGenericSerialized };
