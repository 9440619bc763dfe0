'use strict';
// one device context per process, shared by the galois / merkle / air-assembly replacements
const { createPrimeField, MODULUS } = require('./galois');
let field;
module.exports = { defaultField(modulus) { return field || (field = createPrimeField(modulus === undefined ? MODULUS : modulus)); } };
