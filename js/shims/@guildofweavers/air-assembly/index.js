'use strict';
// replacement for the part of `@guildofweavers/air-assembly` that index.ts:4,18-33, lib/Stark.ts:40 and the examples use:
//   compile(source) -> AirSchema            AirAssembly text (Buffer | string | path to an .aa file): js/air_assembly.js asks the loader
//                                           of this repository (js/aa_loader.js, in process) for the AIR's programs
//   instantiate(schema, component, options) -> AirModule.  `schema` is an AirSchema, or a descriptor:
//       { mimc: { steps[, modulus] } }      the MiMC AirAssembly module of examples/mimc/mimc128Assembly.ts with its dedicated kernels
//       { generic: { ... } }                any AIR as register-machine programs — what GenericAir.descriptor() exports
//                                           (genstark_amd/air_generic.py), e.g. the reference's Rescue 4x128 / Poseidon 6x128 examples
//   prng.sha256(seed, count, field)         the round-constant generator the examples call (examples/mimc/mimc128Assembly.ts:55)
const crypto = require('crypto');
const { MimcAir } = require('../../../air_mimc');
const { GenericAir } = require('../../../air_generic');
const { AirSchema, compile, AssemblyAir } = require('../../../air_assembly');
const { defaultField } = require('../../../context');
module.exports = {
    AirSchema,
    compile,
    prng: {
        // UNVERIFIED restatement (the package is absent: SURVEY appendix A.2); the same function as genstark_amd/air.py: sha256_prng and
        // the `(prng sha256 seed n)` static registers of the loader, so a module's constants and an example's control values agree
        sha256(seed, count, field) {
            const out = [];
            for (let i = 0; i < count; i++) {
                const head = Buffer.alloc(2); head.writeUInt16BE(i + 1, 0);
                const v = BigInt('0x' + crypto.createHash('sha256').update(Buffer.concat([head, seed])).digest().toString('hex'));
                out.push(field ? v % field.modulus : v);
            }
            return out;
        },
    },
    instantiate(schema, component, options) {
        const ef = options && options.extensionFactor;
        if (schema instanceof AirSchema) return new AssemblyAir(schema, component, options);
        if (schema && schema.mimc) return new MimcAir(schema.mimc.steps, ef, defaultField(schema.mimc.modulus));
        if (schema && schema.generic) return new GenericAir(schema.generic, ef, defaultField(BigInt(schema.generic.modulus)));
        throw new Error('expected an AirSchema (compile(source)) or an AIR descriptor: { mimc: {...} } or { generic: {...} }');
    },
};
