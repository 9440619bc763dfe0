'use strict';
// js/galois.js — drop-in for the `@guildofweavers/galois` surface genSTARK uses (SURVEY.md section 8b):
// createPrimeField(modulus) -> FiniteField whose Vector / Matrix objects live in MI355X HBM and whose vector,
// matrix and polynomial members each forward to one entry point of include/gstark.h through the N-API shim
// (napi/gstark_napi.node -> libgstark_hip.so).  Scalar bigint members stay JS BigInt arithmetic, as upstream.
const crypto = require('crypto');
const path = require('path');

const MODULUS = 2n ** 128n - 9n * 2n ** 32n + 1n;
// one build flavour of the library per field (genstark_amd/csrc/build.sh); a process works in ONE field, like the reference's
// example scripts: the first createPrimeField(modulus) picks the library, ELEMENT_SIZE / LOADED_MODULUS follow from it
const LIBRARIES = new Map([
    [MODULUS, 'libgstark_hip.so'],
    [2n ** 64n - 21n * 2n ** 30n + 1n, 'libgstark_hip_q64.so'], [2n ** 32n - 3n * 2n ** 25n + 1n, 'libgstark_hip_q32.so'], [96769n, 'libgstark_hip_q17.so'],
    [2n ** 256n - 351n * 2n ** 32n + 1n, 'libgstark_hip_p256.so'], [2n ** 224n - 2n ** 96n + 1n, 'libgstark_hip_p224.so'],
]);
let ELEMENT_SIZE = 16;
let LOADED_MODULUS = null;

let addon = null;
function native(modulus) {
    if (!addon) {
        const wanted = modulus === undefined ? MODULUS : BigInt(modulus);
        // a modulus none of the fixed builds knows: the runtime-modulus build (gs_set_modulus: any odd modulus below 2^256, once per process)
        const runtime = !process.env.GSTARK_LIB && !LIBRARIES.has(wanted);
        if (runtime && (wanted < 3n || wanted % 2n === 0n || wanted >> 256n)) throw new TypeError(`no build of the library for the field of ${wanted} elements`);
        const lib = process.env.GSTARK_LIB || path.join(__dirname, '..', 'genstark_amd', 'csrc', runtime ? 'libgstark_hip_rt.so' : LIBRARIES.get(wanted));
        const a = require(process.env.GSTARK_ADDON || path.join(__dirname, '..', 'napi', 'gstark_napi.node'));      // GSTARK_ADDON: an instrumented build (tools/build_sanitized.sh)
        const name = a.load(lib);
        if (name !== 'hip-gfx950' && process.env.GSTARK_ALLOW_TEST_DOUBLE !== '1') {
            throw new Error(`refusing backend ${name}: the product path runs on hip-gfx950 only (no CPU fallback)`);
        }
        if (runtime || process.env.GSTARK_SET_MODULUS === '1') {      // (GSTARK_SET_MODULUS=1: GSTARK_LIB names a runtime-modulus library — the tests' double)
            const bytes = Buffer.alloc(32);
            let x = wanted;
            for (let i = 0; i < 32; i++) { bytes[i] = Number(x & 0xFFn); x >>= 8n; }
            a.call('gs_set_modulus', bytes, 32);
        }
        const info = a.fieldInfo();
        ELEMENT_SIZE = info.elementSize;
        LOADED_MODULUS = fromLe(info.modulus, 0, info.elementSize);
        addon = a;
    }
    return addon;
}

// bigint <-> little-endian bytes (lib/utils/serialization.ts:140-146 layout), a 64-bit word at a time: a column of an input register is
// 10^4..10^5 of these per proof, and a byte at a time they cost more than the proof (element sizes are 16 or 32 bytes)
const M64 = (1n << 64n) - 1n;
function putLe(b, off, v) {
    let x = BigInt(v);
    for (let i = 0; i < ELEMENT_SIZE; i += 8) { b.writeBigUInt64LE(x & M64, off + i); x >>= 64n; }
}
function le(v) {  // bigint -> elementSize-byte little-endian Buffer
    const b = Buffer.allocUnsafe(ELEMENT_SIZE);
    putLe(b, 0, v);
    return b;
}
/** values (BigInt, already reduced; or anything BigInt() takes when `mod` is given) -> ONE Buffer of their little-endian elements */
function packLe(values, mod) {
    if (mod) { const reduced = new Array(values.length); for (let i = 0; i < values.length; i++) reduced[i] = mod(BigInt(values[i])); values = reduced; }
    // the addon copies the BigInts' words (napi_get_value_bigint_words: ~30 ns per element); without it — or for the handful of values
    // most calls carry — the loop below
    if (values.length >= 64 && addon && addon.packElements) return addon.packElements(values, ELEMENT_SIZE);
    const b = Buffer.allocUnsafe(values.length * ELEMENT_SIZE);
    for (let i = 0; i < values.length; i++) putLe(b, i * ELEMENT_SIZE, values[i]);
    return b;
}
/** a Buffer of little-endian elements -> BigInt[] */
function unpackLe(raw, size) {
    if (size === undefined) size = ELEMENT_SIZE;
    const n = raw.length / size;
    if (n >= 64 && (size === 16 || size === 32) && addon && addon.unpackElements) return addon.unpackElements(raw, size);
    const out = new Array(n);
    for (let i = 0; i < n; i++) out[i] = fromLe(raw, i * size, size);
    return out;
}
function fromLe(buf, off = 0, size) {
    if (size === undefined) size = ELEMENT_SIZE;
    let v = 0n;
    if (size % 8 === 0) { for (let i = size - 8; i >= 0; i -= 8) v = (v << 64n) | buf.readBigUInt64LE(off + i); return v; }
    for (let i = size - 1; i >= 0; i--) v = (v << 8n) | BigInt(buf[off + i]);
    return v;
}
function sha256(value) {  // same helper as lib/components/QueryIndexGenerator.ts:61-67
    const buffer = (typeof value === 'bigint') ? Buffer.from(value.toString(16), 'hex') : value;
    return BigInt('0x' + crypto.createHash('sha256').update(buffer).digest().toString('hex'));
}

const registry = (typeof FinalizationRegistry !== 'undefined')
    ? new FinalizationRegistry(({ ctx, ptr }) => { try { native().call('gs_free', ctx, ptr); } catch (e) { /* context gone */ } })
    : null;

class DeviceBuffer {
    constructor(field, bytes) {
        this.field = field;
        this.ptr = native().alloc(field.ctx, bytes > 16 ? bytes : 16);
        if (registry) registry.register(this, { ctx: field.ctx, ptr: this.ptr });
    }
}

class Vector {
    constructor(field, length, owner, offset = 0n, elementSize) {
        if (elementSize === undefined) elementSize = ELEMENT_SIZE;
        this.field = field; this.length = length; this.elementSize = elementSize;
        this.owner = owner || new DeviceBuffer(field, length * elementSize);
        this.offset = offset;
        this.seriesBase = undefined;
    }
    get ptr() { return this.owner.ptr + this.offset; }
    get byteLength() { return this.length * this.elementSize; }
    toBuffer(start = 0, count) {
        count = (count === undefined) ? this.length - start : count;
        const out = Buffer.alloc(count * this.elementSize);
        if (count) native().call('gs_download', this.field.ctx, out, this.ptr + BigInt(start * this.elementSize), out.length);
        return out;
    }
    getValue(index) { return fromLe(this.toBuffer(index, 1), 0, this.elementSize); }
    toValues() { return unpackLe(this.toBuffer(), this.elementSize); }
    copyValue(index, destination, offset) {  // lib/Stark.ts:290
        this.toBuffer(index, 1).copy(destination, offset);
        return this.elementSize;
    }
    valuesAt(indexes) {
        const out = Buffer.alloc(indexes.length * this.elementSize);
        if (indexes.length) native().call('gs_gather', this.field.ctx, this.ptr, this.elementSize, indexes, indexes.length, out);
        return indexes.map((_, i) => out.slice(i * this.elementSize, (i + 1) * this.elementSize));
    }
}

class Matrix {
    constructor(field, rowCount, colCount, owner, offset = 0n) {
        this.field = field; this.rowCount = rowCount; this.colCount = colCount; this.elementSize = ELEMENT_SIZE;
        this.owner = owner || new DeviceBuffer(field, rowCount * colCount * ELEMENT_SIZE);
        this.offset = offset;
        this.quarticDomain = undefined;
    }
    get ptr() { return this.owner.ptr + this.offset; }
    toBuffer() {
        const out = Buffer.alloc(this.rowCount * this.colCount * ELEMENT_SIZE);
        if (out.length) native().call('gs_download', this.field.ctx, out, this.ptr, out.length);
        return out;
    }
    getValue(row, col) {
        const out = Buffer.alloc(ELEMENT_SIZE);
        native().call('gs_download', this.field.ctx, out, this.ptr + BigInt((row * this.colCount + col) * ELEMENT_SIZE), ELEMENT_SIZE);
        return fromLe(out);
    }
    toValues() {
        const flat = unpackLe(this.toBuffer(), ELEMENT_SIZE), out = [];
        for (let r = 0; r < this.rowCount; r++) out.push(flat.slice(r * this.colCount, (r + 1) * this.colCount));
        return out;
    }
    rowsToBuffers(indexes) {  // lib/components/LowDegreeProver.ts:53,214,217
        const rec = this.colCount * ELEMENT_SIZE;
        const out = Buffer.alloc(indexes.length * rec);
        if (indexes.length) native().call('gs_gather', this.field.ctx, this.ptr, rec, indexes, indexes.length, out);
        return indexes.map((_, i) => out.slice(i * rec, (i + 1) * rec));
    }
    row(r) { return new Vector(this.field, this.colCount, this.owner, this.offset + BigInt(r * this.colCount * ELEMENT_SIZE)); }
}

class PrimeField {
    constructor(modulus, options) {
        native(modulus);
        if (BigInt(modulus) !== LOADED_MODULUS) throw new TypeError(`the loaded library computes in the field of ${LOADED_MODULUS} elements, not ${modulus} (one field per process)`);
        this.modulus = LOADED_MODULUS; this.elementSize = ELEMENT_SIZE; this.isOptimized = true;
        this.zero = 0n; this.one = 1n;
        this.ctx = (options && options.ctx) || native().ctxCreate((options && options.device) || 0);
    }
    // ---- scalars
    mod(v) { return v >= 0n ? v % this.modulus : ((v % this.modulus) + this.modulus) % this.modulus; }
    add(a, b) { return this.mod(a + b); }
    sub(a, b) { return this.mod(a - b); }
    mul(a, b) { return this.mod(a * b); }
    neg(a) { return this.mod(-a); }
    exp(b, e) {
        if (e < 0n) { b = this.inv(b); e = -e; }
        let r = 1n; b = this.mod(b);
        while (e > 0n) { if (e & 1n) r = (r * b) % this.modulus; b = (b * b) % this.modulus; e >>= 1n; }
        return r;
    }
    inv(a) { return this.mod(a) === 0n ? 0n : this.exp(a, this.modulus - 2n); }
    div(a, b) { return this.mul(a, this.inv(b)); }
    prng(seed, length) {  // UNVERIFIED restatement (SURVEY appendix A.1)
        if (length === undefined) return this.mod(sha256(seed));
        const out = new Array(length); let state = sha256(seed);
        for (let i = 0; i < length; i++) { out[i] = this.mod(state); state = sha256(state); }
        return this.newVectorFrom(out);
    }
    getRootOfUnity(order) {  // UNVERIFIED restatement (SURVEY appendix A.3)
        const o = BigInt(order);
        for (let i = 2n; i < 65536n; i++) {
            const g = this.exp(i, (this.modulus - 1n) / o);
            if (this.exp(g, o) === 1n && (o === 1n || this.exp(g, o / 2n) !== 1n)) return g;
        }
        throw new Error(`Root of unity for order ${order} was not found`);
    }
    // ---- construction
    newVector(length) { return new Vector(this, length); }
    newVectorFrom(values) {
        const v = new Vector(this, values.length);
        if (values.length) native().call('gs_upload', this.ctx, v.ptr, packLe(values, x => this.mod(x)), values.length * ELEMENT_SIZE);
        return v;
    }
    newMatrix(rows, cols) { return new Matrix(this, rows, cols); }
    newMatrixFrom(values) {
        const rows = values.length, cols = rows ? values[0].length : 0;
        const m = new Matrix(this, rows, cols);
        if (rows * cols) {
            const flat = new Array(rows * cols);
            for (let r = 0; r < rows; r++) for (let c = 0; c < cols; c++) flat[r * cols + c] = values[r][c];
            native().call('gs_upload', this.ctx, m.ptr, packLe(flat, x => this.mod(x)), rows * cols * ELEMENT_SIZE);
        }
        return m;
    }
    newMatrixFromVectors(vectors) {
        // shorter rows are zero-extended (polynomials of different degrees: BoundaryConstraints.ts:84-85)
        const cols = Math.max(...vectors.map(v => v.length)); const m = new Matrix(this, vectors.length, cols);
        vectors.forEach((v, r) => {
            native().call('gs_copy', this.ctx, m.ptr + BigInt(r * cols * ELEMENT_SIZE), v.ptr, v.length * ELEMENT_SIZE);
            if (v.length < cols) native().call('gs_upload', this.ctx, m.ptr + BigInt((r * cols + v.length) * ELEMENT_SIZE), Buffer.alloc((cols - v.length) * ELEMENT_SIZE), (cols - v.length) * ELEMENT_SIZE);
        });
        return m;
    }
    matrixRowsToVectors(m) { const out = []; for (let r = 0; r < m.rowCount; r++) out.push(m.row(r)); return out; }
    // ---- vector ops
    _binary(fnVec, fnScalar, a, b) {
        const out = new Vector(this, a.length);
        if (typeof b === 'bigint') native().call(fnScalar, this.ctx, a.ptr, le(this.mod(b)), a.length, out.ptr);
        else {
            if (a.length !== b.length) throw new Error('Cannot combine vector elements: vectors have different lengths');
            native().call(fnVec, this.ctx, a.ptr, b.ptr, a.length, out.ptr);
        }
        return out;
    }
    addVectorElements(a, b) { return this._binary('gs_vec_add', 'gs_vec_add_scalar', a, b); }
    subVectorElements(a, b) { return this._binary('gs_vec_sub', 'gs_vec_sub_scalar', a, b); }
    mulVectorElements(a, b) { return this._binary('gs_vec_mul', 'gs_vec_mul_scalar', a, b); }
    divVectorElements(a, b) {
        if (typeof b === 'bigint') return this.mulVectorElements(a, this.inv(b));
        const out = new Vector(this, a.length);
        native().call('gs_vec_div', this.ctx, a.ptr, b.ptr, a.length, out.ptr);
        return out;
    }
    invVectorElements(a) { const out = new Vector(this, a.length); native().call('gs_vec_inv', this.ctx, a.ptr, a.length, out.ptr); return out; }
    expVectorElements(a, e) {
        if (e < 0n) { a = this.invVectorElements(a); e = -e; }
        const out = new Vector(this, a.length); native().call('gs_vec_exp', this.ctx, a.ptr, le(e), a.length, out.ptr); return out;
    }
    combineVectors(a, b) { const out = Buffer.alloc(ELEMENT_SIZE); native().call('gs_combine', this.ctx, a.ptr, b.ptr, a.length, out); return fromLe(out); }
    mulMatrixByVector(m, v) {   // examples/poseidon/utils.ts:45
        const out = [];
        for (let r = 0; r < m.rowCount; r++) out.push(this.combineVectors(new Vector(this, m.colCount, m.owner, m.offset + BigInt(r * m.colCount * ELEMENT_SIZE)), v));
        return this.newVectorFrom(out);
    }
    combineManyVectors(vectors, coefficients) {
        const ks = Array.isArray(coefficients) ? coefficients : coefficients.toValues();
        const out = new Vector(this, vectors[0].length);
        native().call('gs_combine_many', this.ctx, vectors.map(v => v.ptr), packLe(ks), vectors.length, vectors[0].length, out.ptr);
        return out;
    }
    getPowerSeries(base, length) {
        const out = new Vector(this, length);
        native().call('gs_power_series', this.ctx, le(this.mod(base)), length, out.ptr);
        out.seriesBase = this.mod(base);
        return out;
    }
    pluckVector(v, skip, times) { const out = new Vector(this, times); native().call('gs_pluck', this.ctx, v.ptr, v.length, skip, times, out.ptr); return out; }
    transposeVector(v, columns, step = 1) {
        const rows = v.length / (columns * step);
        const m = new Matrix(this, rows, columns);
        native().call('gs_transpose_vector', this.ctx, v.ptr, v.length, columns, step, m.ptr);
        if (columns === 4 && v.seriesBase !== undefined) m.quarticDomain = { omega: v.seriesBase, n: v.length, step };
        return m;
    }
    // ---- matrix ops
    transposeMatrix(m) { const out = new Matrix(this, m.colCount, m.rowCount); native().call('gs_transpose_matrix', this.ctx, m.ptr, m.rowCount, m.colCount, out.ptr); return out; }
    joinMatrixRows(m) { return new Vector(this, m.rowCount * m.colCount, m.owner, m.offset); }
    subMatrixElementsFromVectors(vectors, m) {
        const out = new Matrix(this, m.rowCount, m.colCount);
        native().call('gs_sub_matrix_from_vectors', this.ctx, vectors.map(v => v.ptr), m.ptr, m.rowCount, m.colCount, out.ptr);
        return out;
    }
    divMatrixElements(a, b) { const out = new Matrix(this, a.rowCount, a.colCount); native().call('gs_vec_div', this.ctx, a.ptr, b.ptr, a.rowCount * a.colCount, out.ptr); return out; }
    // ---- polynomials
    _omegaOf(roots) { return roots.seriesBase !== undefined ? roots.seriesBase : (roots.length > 1 ? roots.getValue(1) : 1n); }
    evalPolyAtRoots(poly, roots) {
        const out = new Vector(this, roots.length);
        native().call('gs_eval_polys_at_roots', this.ctx, poly.ptr, 1, poly.length, le(this._omegaOf(roots)), roots.length, out.ptr);
        return out;
    }
    evalPolysAtRoots(polys, roots) {
        const out = new Matrix(this, polys.rowCount, roots.length);
        native().call('gs_eval_polys_at_roots', this.ctx, polys.ptr, polys.rowCount, polys.colCount, le(this._omegaOf(roots)), roots.length, out.ptr);
        return out;
    }
    interpolateRoots(roots, ys) {
        const n = roots.length, isM = ys instanceof Matrix;
        const out = isM ? new Matrix(this, ys.rowCount, n) : new Vector(this, n);
        native().call('gs_interpolate_roots', this.ctx, ys.ptr, isM ? ys.rowCount : 1, le(this._omegaOf(roots)), n, out.ptr);
        return out;
    }
    evalPolyAt(poly, x) { const out = Buffer.alloc(ELEMENT_SIZE); native().call('gs_eval_poly_at', this.ctx, poly.ptr, poly.length, le(this.mod(x)), out); return fromLe(out); }
    mulPolys(a, b) {
        // tiny operands (BoundaryConstraints.ts:30) on the host; larger ones through the device NTT
        const la = a.length, lb = b.length;
        if (la * lb <= 4096) {
            const av = a.toValues(), bv = b.toValues(), out = new Array(la + lb - 1).fill(0n);
            for (let i = 0; i < la; i++) for (let j = 0; j < lb; j++) out[i + j] = this.mod(out[i + j] + av[i] * bv[j]);
            return this.newVectorFrom(out);
        }
        let n = 1; while (n < la + lb - 1) n <<= 1;
        const roots = this.getPowerSeries(this.getRootOfUnity(n), n);
        const full = this.interpolateRoots(roots, this.mulVectorElements(this.evalPolyAtRoots(a, roots), this.evalPolyAtRoots(b, roots)));
        return new Vector(this, la + lb - 1, full.owner, full.offset);
    }
    padPoly(v, length) {
        if (v.length === length) return v;
        const out = new Vector(this, length);
        native().call('gs_copy', this.ctx, out.ptr, v.ptr, v.length * ELEMENT_SIZE);
        const zeros = Buffer.alloc((length - v.length) * ELEMENT_SIZE);
        native().call('gs_upload', this.ctx, out.ptr + BigInt(v.length * ELEMENT_SIZE), zeros, zeros.length);
        return out;
    }
    addPolys(a, b) { const n = Math.max(a.length, b.length); return this.addVectorElements(this.padPoly(a, n), this.padPoly(b, n)); }
    subPolys(a, b) { const n = Math.max(a.length, b.length); return this.subVectorElements(this.padPoly(a, n), this.padPoly(b, n)); }
    mulPolyByConstant(a, c) { return this.mulVectorElements(a, this.mod(c)); }
    interpolate(xs, ys) {
        const n = xs.length, out = Buffer.alloc(ELEMENT_SIZE * n);
        native().call('gs_small_interpolate', xs.toBuffer(), ys.toBuffer(), n, out);
        const v = new Vector(this, n); native().call('gs_upload', this.ctx, v.ptr, out, out.length); return v;
    }
    interpolateQuarticBatch(xs, ys) {
        const out = new Matrix(this, ys.rowCount, 4);
        if (xs.quarticDomain) native().call('gs_interpolate_quartic_domain', this.ctx, le(xs.quarticDomain.omega), xs.quarticDomain.n, xs.quarticDomain.step, ys.ptr, ys.rowCount, out.ptr);
        else native().call('gs_interpolate_quartic_batch', this.ctx, xs.ptr, ys.ptr, ys.rowCount, out.ptr);
        return out;
    }
    evalQuarticBatch(polys, x) { const out = new Vector(this, polys.rowCount); native().call('gs_eval_quartic_batch', this.ctx, polys.ptr, polys.rowCount, le(this.mod(x)), out.ptr); return out; }
}

function createPrimeField(modulus, options) { return new PrimeField(modulus, options); }

module.exports = { createPrimeField, PrimeField, Vector, Matrix, MODULUS, LIBRARIES, native, le, packLe, unpackLe, fromLe, sha256 };
