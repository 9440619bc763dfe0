'use strict';
// js/prover.js — ONE native call for a whole proof: Stark.prove() + Serializer.serializeProof() through the
// native driver (include/gstark_prover.h, genstark_amd/csrc/prover.cc), reached through the same N-API addon that carries the
// member-by-member galois / merkle surface.  The returned Buffer is what lib/Serializer.ts:83-144 (`stark.parse`) reads.
const path = require('path');
const { native, le } = require('./galois');

const HASH_ALG = { sha256: 0, blake2s256: 1 };
// one build of the driver per field flavour, like the ABI library (genstark_amd/csrc/build.sh)
const DRIVERS = new Map([
    [2n ** 128n - 9n * 2n ** 32n + 1n, 'libgstark_prover.so'], [2n ** 64n - 21n * 2n ** 30n + 1n, 'libgstark_prover_q64.so'],
    [2n ** 32n - 3n * 2n ** 25n + 1n, 'libgstark_prover_q32.so'], [96769n, 'libgstark_prover_q17.so'],
    [2n ** 256n - 351n * 2n ** 32n + 1n, 'libgstark_prover_p256.so'], [2n ** 224n - 2n ** 96n + 1n, 'libgstark_prover_p224.so'],
]);
function driverPath(field) {
    if (process.env.GSTARK_PROVER_LIB) return process.env.GSTARK_PROVER_LIB;
    // (a modulus without a build of its own: the runtime-modulus driver, which adopts the modulus of the library it is bound to)
    return path.join(__dirname, '..', 'genstark_amd', 'csrc', DRIVERS.has(field.modulus) ? DRIVERS.get(field.modulus) : 'libgstark_prover_rt.so');
}

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
    return native().proveMimcSerialized(f.ctx, driverPath(f), job);
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
    return native().proveGenericSerialized(f.ctx, driverPath(f), job);
}

// Stark.verify() of serialized proof bytes by the NATIVE verifier (genstark_amd/csrc/verifier.h; CPU only, no device work): true, or throws
// the reference's message.  The job carries what a verifier needs of the statement: no trace tables, no first rows.
function verifyMimcSerialized(air, options, assertions, proof) {
    const f = air.field, root = air._root !== undefined ? air._root : f.getRootOfUnity(air.steps * air.extensionFactor);
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(root), seed: le(0n),
        roundConstants: Buffer.concat(air.roundConstants.map(le)), kTable: 0n, kLen: 0,
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
    };
    return native().proveMimcSerialized(f.ctx, driverPath(f), job, Buffer.from(proof));
}
function verifyGenericSerialized(air, options, assertions, proof) {
    const f = air.field, e = air.evaluationProgram;
    const statics = [].concat(...air.staticRegisters);
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(air.rootOfUnity),
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
        registers: air.traceRegisterCount, degrees: air.constraintDegrees, tCode: [], iCode: [], eCode: e.code,
        consts: e.consts.length ? Buffer.concat(e.consts.map(le)) : Buffer.alloc(0), vmRegs: e.nregs,
        staticValues: statics.length ? Buffer.concat(statics.map(v => le(f.mod(v)))) : le(0n), staticPeriods: air.staticRegisters.map(v => v.length),
        staticTables: 0n, staticLens: air.staticRegisters.map(() => 0), firstRows: Buffer.alloc(air.traceRegisterCount * f.elementSize),
        segments: 0, segmentLen: 0,
    };
    return native().proveGenericSerialized(f.ctx, driverPath(f), job, Buffer.from(proof));
}

module.exports = { proveMimcSerialized, proveGenericSerialized, verifyMimcSerialized, verifyGenericSerialized };
