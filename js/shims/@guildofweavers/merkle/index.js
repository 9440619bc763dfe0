'use strict';
// replacement for `@guildofweavers/merkle` (INTEGRATION.md section 2)
const m = require('../../../merkle');
const { defaultField } = require('../../../context');
module.exports = {
    MerkleTree: m.MerkleTree,
    // upstream: createHash(algorithm, useWasm) — lib/Stark.ts:50; the flag selected the wasm build, here the process-wide device context
    createHash: (algorithm, _useWasm) => m.createHash(algorithm, defaultField()),
};
