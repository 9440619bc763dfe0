'use strict';
// js/prover.js — ONE native call for a whole proof: Stark.prove() + Serializer.serializeProof() through the
// native driver (include/gstark_prover.h, genstark_amd/csrc/prover.cc), reached through the same N-API addon that carries the
// member-by-member galois / merkle surface.  The returned Buffer is what lib/Serializer.ts:83-144 (`stark.parse`) reads.
const path = require('path');
const { native, le, packLe } = require('./galois');

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
    const context = air.jobContext ? air.jobContext() : air.initProvingContext([], [seed]);      // the cyclic register's table over the composition domain, root of unity
    const f = air.field;
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(context.rootOfUnity), seed: le(f.mod(seed)),
        roundConstants: air._rcPacked || (air._rcPacked = packLe(air.roundConstants)), kTable: context.kTable.ptr, kLen: context.kTable.length,
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
    };
    return native().proveMimcSerialized(f.ctx, driverPath(f), job);
}

// ... and for an AIR given as register-machine programs (js/air_generic.js: the reference's Rescue / Poseidon examples): kind 1 of
// gs_prover_air.  The two programs keep separate constant pools; the driver takes one, so the evaluator's constant indexes are rebased.
const flatRows = rows => { const out = []; for (const row of rows) for (const v of row) out.push(v); return out; };
function genericJob(air, context, options, assertions) {
    if (!(options.hashAlgorithm in HASH_ALG)) throw new TypeError(`Hash algorithm ${options.hashAlgorithm} is not supported`);
    const f = air.field;
    const t = air.transitionProgram, e = air.evaluationProgram, init = air.initProgram;
    if (init && init.consts.length > t.consts.length) throw new Error('the init program extends the transition program\'s constant pool');
    // (the programs' marshalled form is a constant of the AIR: kept with it)
    if (!air._jobPrograms) {
        const base = t.consts.length, code = e.code.slice();
        for (let i = 0; i < code.length; i += 4) {
            if (code[i] === 0) code[i + 2] += base;            // LOADC: constant index
            else if (code[i] === 8) code[i + 3] += base;       // POWC: exponent index
        }
        const all = t.consts.concat(e.consts);
        air._jobPrograms = { eCode: code, consts: all.length ? packLe(all) : Buffer.alloc(0) };
    }
    const eCode = air._jobPrograms.eCode;
    const job = {
        steps: air.steps, extensionFactor: air.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(air.rootOfUnity),
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
        registers: air.traceRegisterCount, degrees: air.constraintDegrees, tCode: t.code, iCode: init ? init.code : [], eCode,
        consts: air._jobPrograms.consts, vmRegs: Math.max(t.nregs, e.nregs, init ? init.nregs : 0),
        // static registers: the public ones, then this proof's secret columns (struct gs_prover_air: static_values / static_tables hold both)
        staticValues: context.staticValuesPacked(), staticPeriods: context.allStaticColumns().map(v => v.length), staticTables: context.staticTables.ptr,
        staticLens: context.staticLens, firstRows: context.firstRows.packedFirstRows || packLe(flatRows(context.firstRows)),
        segments: air.segmentLength === null ? 0 : (context.firstRows.packedFirstRows ? context.firstRows.rows : context.firstRows.length),
        segmentLen: air.segmentLength === null ? 0 : air.segmentLength,
    };
    if (context.secretRegisterTraces.length) job.secretTraces = context.secretRegisterTraces.map(v => v.ptr);
    return job;
}
function proveGenericSerialized(air, options, assertions, seed) {
    const context = air.jobContext ? air.jobContext(seed) : air.initProvingContext([], seed);
    return native().proveGenericSerialized(air.field.ctx, driverPath(air.field), genericJob(air, context, options, assertions));
}

// ... and for an air-assembly component WITH input registers (js/air_assembly.js: AssemblyAir): the loader lays the inputs out (the inner
// AIR of this shape, the secret columns, the first rows), the native driver proves it and writes the inputs' shapes into the proof
// (iShapes, lib/Stark.ts:161); the native verifier sizes the trace from the shapes the proof carries (lib/Stark.ts:176).
const declWords = d => [d.parent === null || d.parent === undefined ? 0 : d.parent + 1, d.peer === null || d.peer === undefined ? 0 : d.peer + 1, d.steps || 0, d.shift >>> 0, d.secret ? 1 : 0];
function proveAssemblySerialized(assemblyAir, options, assertions, inputs, seed) {
    if (!assemblyAir.info.inputRegisters) return proveGenericSerialized(assemblyAir.generic, options, assertions, seed);
    const context = assemblyAir.initProvingContext(inputs, seed);
    const job = genericJob(context.air, context, options, assertions);
    job.inputRegisters = [].concat(...assemblyAir.info.inputDeclarations.map(declWords));
    job.inputShapes = [].concat(...context.inputShapes.map(sh => [sh.length].concat(sh)));
    return native().proveGenericSerialized(context.air.field.ctx, driverPath(context.air.field), job);
}
function verifyAssemblySerialized(assemblyAir, options, assertions, proof, publicInputs) {
    if (!(options.hashAlgorithm in HASH_ALG)) throw new TypeError(`Hash algorithm ${options.hashAlgorithm} is not supported`);
    if (!assemblyAir.info.inputRegisters) return verifyGenericSerialized(assemblyAir.generic, options, assertions, proof);
    const f = assemblyAir.field, info = assemblyAir.info, e = info.evaluation;
    // the field's root of unity of the largest power-of-two order (at most 2^32): the driver squares it down to the evaluation domain's
    let adicity = 0;
    for (let x = f.modulus - 1n; x % 2n === 0n; x /= 2n) adicity++;
    const log2 = Math.min(adicity, 32);
    const flatInto = (x, out) => { if (Array.isArray(x)) for (const v of x) flatInto(v, out); else out.push(x); return out; };   // (no spread: 10^5 lists would not fit an argument list)
    const lists = (publicInputs || []).map(x => flatInto(x, []));
    const cycles = info.cycles.map(c => c.map(BigInt));
    const job = {
        steps: 0, extensionFactor: assemblyAir.extensionFactor, exeQueryCount: options.exeQueryCount, friQueryCount: options.friQueryCount,
        hashAlg: HASH_ALG[options.hashAlgorithm], rootOfUnity: le(f.getRootOfUnity(2 ** log2)), rootOfUnityLog2: log2,
        assertions: assertions.map(a => ({ step: a.step, register: a.register, value: le(f.mod(a.value)) })),
        registers: assemblyAir.traceRegisterCount, degrees: assemblyAir.constraintDegrees, tCode: [], iCode: [], eCode: e.code,
        consts: e.consts.length ? Buffer.concat(e.consts.map(v => le(BigInt(v)))) : Buffer.alloc(0), vmRegs: e.nregs,
        staticValues: cycles.length ? Buffer.concat([].concat(...cycles).map(v => le(f.mod(v)))) : le(0n), staticPeriods: cycles.map(c => c.length),
        staticTables: 0n, staticLens: cycles.map(() => 0), firstRows: Buffer.alloc(assemblyAir.traceRegisterCount * f.elementSize), segments: 0, segmentLen: 0,
        nsecret: assemblyAir.secretInputCount, inputRegisters: [].concat(...info.inputDeclarations.map(declWords)),
        staticSources: [].concat(...info.staticSources), publicInputs: lists.length ? Buffer.concat(lists.map(l => packLe(l, v => f.mod(v)))) : Buffer.alloc(0),
        publicInputCounts: lists.map(l => l.length),
    };
    return native().proveGenericSerialized(f.ctx, driverPath(f), job, Buffer.from(proof));
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
        staticValues: statics.length ? packLe(statics, v => f.mod(v)) : le(0n), staticPeriods: air.staticRegisters.map(v => v.length),
        staticTables: 0n, staticLens: air.staticRegisters.map(() => 0), firstRows: Buffer.alloc(air.traceRegisterCount * f.elementSize),
        segments: 0, segmentLen: 0,
    };
    return native().proveGenericSerialized(f.ctx, driverPath(f), job, Buffer.from(proof));
}

/** seed (as proveGenericSerialized takes it) -> the first rows packed once, for many proofs: what a caller whose inputs already are bytes
 *  never pays per proof (the Python host's Prover.pack_seed) */
function packSeed(air, seed) { return air.packSeed(seed); }

module.exports = { packSeed, proveMimcSerialized, proveGenericSerialized, proveAssemblySerialized, verifyMimcSerialized, verifyGenericSerialized, verifyAssemblySerialized };
