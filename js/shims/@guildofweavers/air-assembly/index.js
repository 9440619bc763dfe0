'use strict';
// replacement for the part of `@guildofweavers/air-assembly` lib/Stark.ts:40 uses: instantiate(schema, component, options).
// Only the MiMC AirAssembly module of examples/mimc/mimc128Assembly.ts is provided; `schema` is the descriptor
// { mimc: { steps[, modulus] } } instead of a compiled AirSchema (the AirAssembly compiler is out of scope).
const { MimcAir } = require('../../../air_mimc');
const { defaultField } = require('../../../context');
module.exports = {
    instantiate(schema, component, options) {
        if (!schema || !schema.mimc) throw new Error('only the MiMC AIR is available in this build');
        return new MimcAir(schema.mimc.steps, options && options.extensionFactor, defaultField(schema.mimc.modulus));
    },
};
