'use strict';
// `@guildofweavers/air-script` is required by the reference's index.js at load time (index.ts:5) and used by instantiateScript only.
// The AirScript compiler is out of scope (SURVEY section 7): the module loads, the call says so.
module.exports = {
    compile() { throw new Error('AirScript sources are not supported here: compile the script to AirAssembly upstream and pass the assembly (instantiate)'); },
};
