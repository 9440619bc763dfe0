'use strict';
// tests/addon_validation.js — the N-API addon's argument validation (napi/gstark_napi.cc): every call below hands the addon something a
// JavaScript caller can get wrong — a missing library, wrong types, wrong arity, Buffers of the wrong length, job objects with fields
// missing or inconsistent, a garbage proof — and must come back as a thrown Error, never as a crash.  tests/test_sanitizers.py runs this
// under the ASAN + UBSAN build of the addon (GSTARK_ADDON) and of the driver (GSTARK_PROVER_LIB), against the oracle's library.
const assert = require('assert');
const a = require(process.env.GSTARK_ADDON);
const LIB = process.env.GSTARK_LIB, DRIVER = process.env.GSTARK_PROVER_LIB;
let n = 0;
function refuses(fn, pattern) { assert.throws(fn, pattern || /./); n++; }

// before load()
refuses(() => a.ctxCreate(0), /load\(path\) first/);
refuses(() => a.alloc({}, 16), /load\(path\) first/);
refuses(() => a.ctxDestroy({}), /load\(path\) first/);
refuses(() => a.call('gs_sync', {}), /load\(path\) first/);
refuses(() => a.merkleProveBatch({}, 0n, 0n, 4, [0]), /load\(path\) first/);
refuses(() => a.fieldInfo(), /no library loaded/);
refuses(() => a.proveMimcSerialized({}, DRIVER, {}), /load\(path\) first/);
refuses(() => a.load(), /load\(path\)/);
refuses(() => a.load(123), /load\(path\)/);
refuses(() => a.load('/nonexistent/libgstark_hip.so'), /no CPU fallback/);
refuses(() => a.load(DRIVER), /not a gstark library/);

assert.strictEqual(typeof a.load(LIB), 'string');
const es = a.fieldInfo().elementSize;
const ctx = a.ctxCreate(0);
const ptr = a.alloc(ctx, 4096);

// the generic forwarder
refuses(() => a.call(), /call\(name/);
refuses(() => a.call(7), /string/);
refuses(() => a.call('gs_nope', ctx), /unknown gstark function/);
refuses(() => a.call('gs_sync'), /wrong number of arguments/);
refuses(() => a.call('gs_sync', ctx, 1), /wrong number of arguments/);
refuses(() => a.call('gs_sync', 5), /expected a context/);
refuses(() => a.call('gs_sync', {}), /expected a context/);
refuses(() => a.call('gs_upload', ctx, 'x', Buffer.alloc(16), 16), /number\/BigInt/);
refuses(() => a.call('gs_upload', ctx, ptr, 'not a buffer', 16), /expected a Buffer/);
refuses(() => a.call('gs_upload', ctx, ptr, Buffer.alloc(16), -1), /number\/BigInt/);
refuses(() => a.call('gs_combine_many', ctx, [ptr, 'x'], Buffer.alloc(2 * es), 2, 4, ptr), /bad array element/);
refuses(() => a.call('gs_combine_many', ctx, 'notarray', Buffer.alloc(2 * es), 2, 4, ptr));
refuses(() => a.call('gs_air_trace', ctx, [0, 'x'], 1, Buffer.alloc(es), 1, 4, 1, Buffer.alloc(es), [1], 1, Buffer.alloc(es), 8, ptr), /bad array element/);
refuses(() => a.alloc(ctx, -5), /alloc\(ctx, bytes\)/);
refuses(() => a.alloc(ctx, 'x'), /alloc\(ctx, bytes\)/);
refuses(() => a.alloc(5, 16), /alloc\(ctx, bytes\)/);
refuses(() => a.ctxDestroy(), /ctxDestroy\(ctx\)/);
refuses(() => a.ctxDestroy(17), /ctxDestroy\(ctx\)/);
refuses(() => a.merkleProveBatch(ctx, 'a', ptr, 4, [0]), /bad arguments/);
refuses(() => a.merkleProveBatch(ctx, ptr, ptr, 3, [0]), /bad arguments/);
refuses(() => a.merkleProveBatch(ctx, ptr, ptr, 0, [0]), /bad arguments/);
refuses(() => a.merkleProveBatch(ctx, ptr, ptr, 4, 'x'), /bad index list/);
refuses(() => a.merkleProveBatch(ctx, ptr, ptr, 4, [0, 'x']), /bad index/);
refuses(() => a.merkleProveBatch(ctx, ptr, ptr, 4, [9]), /bad index/);

// one-call prove / verify: the job object is caller-controlled
const elt = v => { const b = Buffer.alloc(es); b.writeUInt32LE(v, 0); return b; };
const mimc = { steps: 64, extensionFactor: 16, exeQueryCount: 8, friQueryCount: 8, hashAlg: 0, rootOfUnity: elt(3), seed: elt(3), roundConstants: Buffer.alloc(es * 4),
               kTable: ptr, kLen: 4, assertions: [{ step: 0, register: 0, value: elt(3) }] };
refuses(() => a.proveMimcSerialized(ctx, DRIVER), /job/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, 5), /job/);
refuses(() => a.proveMimcSerialized(5, DRIVER, mimc), /job/);
refuses(() => a.proveMimcSerialized(ctx, 7, mimc), /driver library path/);
refuses(() => a.proveMimcSerialized(ctx, '/nonexistent/driver.so', mimc), /cannot load/);
refuses(() => a.proveMimcSerialized(ctx, LIB, mimc), /gs_prover_open failed/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, {}), /malformed job/);
for (const [k, v] of [['steps', 'x'], ['rootOfUnity', Buffer.alloc(es - 1)], ['seed', Buffer.alloc(es + 1)], ['roundConstants', Buffer.alloc(es + 3)], ['kTable', {}], ['hashAlg', -1]])
    refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { [k]: v })), /malformed job/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { assertions: 'x' })));
refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { assertions: [{ step: 0, register: 0, value: Buffer.alloc(3) }] })), /malformed assertion/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { assertions: [{ step: 'x', register: 0, value: elt(1) }] })), /malformed assertion/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, mimc, 'not a buffer'), /proof must be a Buffer/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, mimc, Buffer.alloc(0)), /malformed proof|truncated/);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, mimc, Buffer.alloc(5000, 0xff)), /./);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { steps: 48 }), Buffer.alloc(5000, 1)), /./);
refuses(() => a.proveMimcSerialized(ctx, DRIVER, Object.assign({}, mimc, { assertions: [] }), Buffer.alloc(5000, 1)), /At least one assertion/);

const generic = { steps: 64, extensionFactor: 16, exeQueryCount: 8, friQueryCount: 8, hashAlg: 1, rootOfUnity: elt(3), assertions: [{ step: 0, register: 0, value: elt(1) }],
                  registers: 1, degrees: [1], tCode: [1, 0, 0, 0, 9, 0, 0, 0], iCode: [], eCode: [1, 0, 0, 0, 9, 0, 0, 0], consts: Buffer.alloc(es), vmRegs: 2,
                  staticValues: Buffer.alloc(es * 2), staticPeriods: [2], staticTables: ptr, staticLens: [8], firstRows: elt(1), segments: 0, segmentLen: 0 };
refuses(() => a.proveGenericSerialized(ctx, DRIVER, {}), /malformed job/);
for (const [k, v] of [['tCode', [1, 0, 0]], ['eCode', [1]], ['iCode', [0, 0]], ['staticLens', []], ['staticValues', Buffer.alloc(es)], ['firstRows', Buffer.alloc(es * 2)],
                      ['consts', Buffer.alloc(es + 1)], ['degrees', 'x'], ['staticPeriods', [2, 'x']], ['registers', -1], ['rootOfUnity', 'x']])
    refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, generic, { [k]: v })), /malformed job/);
// ... and the optional secret / input register fields (round 6)
for (const [k, v] of [['inputRegisters', [0, 0, 8]], ['inputRegisters', [2, 0, 8, 0, 0]], ['secretTraces', [0n]], ['secretTraces', 'x'], ['nsecret', 999], ['rootOfUnityLog2', 99]])
    refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, generic, { [k]: v })), /secret \/ input registers/);
const shaped = Object.assign({}, generic, { inputRegisters: [0, 0, 8, 0, 1] });
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 8, 1] })), /secret \/ input registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [3, 8] })), /secret \/ input registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 8], staticSources: [0] })), /secret \/ input registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 8], staticSources: [1, 0] })), /cyclic static registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 8], publicInputs: Buffer.alloc(es), publicInputCounts: [2] })), /secret \/ input registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 8], publicInputs: 'x', publicInputCounts: [] })), /secret \/ input registers/);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 3] })), /./);          // shapes that lay out no trace: the driver's message
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, shaped, { inputShapes: [1, 4], staticSources: [0, 0], steps: 0, rootOfUnityLog2: 20 }), Buffer.alloc(4000, 3)), /./);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, generic, Buffer.alloc(3000, 7)), /./);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, generic, { eCode: [77, 0, 0, 0] }), Buffer.alloc(3000, 0)), /./);
refuses(() => a.proveGenericSerialized(ctx, DRIVER, Object.assign({}, generic, { eCode: [0, 0, 99, 0] }), Buffer.alloc(3000, 0)), /./);

a.ctxDestroy(ctx);
// element packing (packElements / unpackElements: BigInt words copied natively)
refuses(() => a.packElements(), /packElements\(values/);
refuses(() => a.packElements('x', 16), /packElements\(values/);
refuses(() => a.packElements([1n], 8), /packElements\(values/);
refuses(() => a.packElements([1n, 2], 16), /must be a BigInt/);
refuses(() => a.packElements([1n, 'x'], 16), /must be a BigInt/);
refuses(() => a.packElements([-1n], 16), /negative or wider/);
refuses(() => a.packElements([1n << 128n], 16), /negative or wider/);
refuses(() => a.packElements([1n << 256n], 32), /negative or wider/);
refuses(() => a.unpackElements(), /unpackElements\(buffer/);
refuses(() => a.unpackElements([1, 2], 16), /unpackElements\(buffer/);
refuses(() => a.unpackElements(Buffer.alloc(16), 7), /unpackElements\(buffer/);
refuses(() => a.unpackElements(Buffer.alloc(17), 16), /whole number of elements/);
{
    const values = [0n, 1n, (1n << 64n) - 1n, 1n << 64n, (1n << 128n) - 1n, 0x0123456789abcdef0fedcba987654321n];
    const packed = a.packElements(values, 16);
    assert.strictEqual(packed.length, 16 * values.length);
    assert.strictEqual(packed.slice(80, 96).toString('hex'), '21436587a9cbed0fefcdab8967452301');       // little-endian bytes
    assert.deepStrictEqual(a.unpackElements(packed, 16), values);
    const wide = [(1n << 256n) - 1n, 1n << 200n, 5n];
    assert.deepStrictEqual(a.unpackElements(a.packElements(wide, 32), 32), wide);
    assert.strictEqual(a.packElements([], 16).length, 0);
    assert.deepStrictEqual(a.unpackElements(Buffer.alloc(0), 32), []);
}

console.log(`addon validation OK: ${n} malformed calls refused`);
