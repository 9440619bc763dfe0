'use strict';
// js/prover.js — ONE native call for a whole proof: Stark.prove() + Serializer.serializeProof() of the MiMC AIR through the
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

module.exports = { proveMimcSerialized };
