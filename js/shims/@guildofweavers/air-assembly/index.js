'use strict';
// replacement for the part of `@guildofweavers/air-assembly` lib/Stark.ts:40 uses: instantiate(schema, component, options).
// `schema` is a descriptor instead of a compiled AirSchema (the AirAssembly compiler is out of scope on the JS side — the Python
// host has one, genstark_amd/airassembly.py):
//   { mimc: { steps[, modulus] } }      the MiMC AirAssembly module of examples/mimc/mimc128Assembly.ts
//   { generic: { ... } }                any AIR as register-machine programs — what GenericAir.descriptor() exports
//                                       (genstark_amd/air_generic.py), e.g. the reference's Rescue 4x128 / Poseidon 6x128 examples
const { MimcAir } = require('../../../air_mimc');
const { GenericAir } = require('../../../air_generic');
const { defaultField } = require('../../../context');
module.exports = {
    instantiate(schema, component, options) {
        const ef = options && options.extensionFactor;
        if (schema && schema.mimc) return new MimcAir(schema.mimc.steps, ef, defaultField(schema.mimc.modulus));
        if (schema && schema.generic) return new GenericAir(schema.generic, ef, defaultField(BigInt(schema.generic.modulus)));
        throw new Error('expected an AIR descriptor: { mimc: {...} } or { generic: {...} }');
    },
};
