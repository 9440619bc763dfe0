'use strict';
// replacement for `@guildofweavers/galois` (INTEGRATION.md section 2): same exports, MI355X-backed
module.exports = require('../../../galois');
