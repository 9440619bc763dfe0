'use strict';
// js/merkle.js — drop-in for the `@guildofweavers/merkle` surface genSTARK uses: createHash(algorithm, useWasm) ->
// Hash {digest, merge, mergeVectorRows, digestValues, digestSize, isOptimized}; MerkleTree {create, root, proveBatch,
// verifyBatch}.  Call sites: lib/Stark.ts:50,115,118,150,206; lib/components/LowDegreeProver.ts:45-53,163-164,201-217.
const { native, Vector } = require('./galois');

const ALGS = { sha256: 0, blake2s256: 1 };
const DIGEST = 32;

class Hash {
    constructor(algorithm, field) {
        if (!(algorithm in ALGS)) throw new TypeError(`Hash algorithm ${algorithm} is not supported`);
        this.algorithm = algorithm; this.alg = ALGS[algorithm]; this.field = field;
        this.digestSize = DIGEST; this.isOptimized = true;
    }
    digest(value) { const out = Buffer.alloc(DIGEST); native().call('gs_hash_digest', this.field.ctx, this.alg, value, value.length, out); return out; }
    merge(a, b) { return this.digest(Buffer.concat([a, b])); }
    mergeVectorRows(vectors) {
        const n = vectors[0].length, out = new Vector(this.field, n, undefined, 0n, DIGEST);
        native().call('gs_hash_merge_rows', this.field.ctx, this.alg, vectors.map(v => v.ptr), vectors.length, n, out.ptr);
        return out;
    }
    digestValues(values, valueSize) {
        let src = values, bytes;
        if (Buffer.isBuffer(values)) {
            src = new Vector(this.field, values.length, undefined, 0n, 1);
            native().call('gs_upload', this.field.ctx, src.ptr, values, values.length);
            bytes = values.length;
        } else bytes = values.rowCount !== undefined ? values.rowCount * values.colCount * values.elementSize : values.byteLength;
        const count = bytes / valueSize, out = new Vector(this.field, count, undefined, 0n, DIGEST);
        native().call('gs_hash_digest_values', this.field.ctx, this.alg, src.ptr, valueSize, count, out.ptr);
        return out;
    }
}
// upstream signature is createHash(algorithm, useWasm); the field object carries the device context here
function createHash(algorithm, field) { return new Hash(algorithm, field); }

function normalize(indexes) {
    const out = new Set();
    for (const ix of indexes.slice().sort((a, b) => a - b)) out.add(ix - (ix & 1));
    return Array.from(out);
}

class MerkleTree {
    constructor(values, nodes, hash) { this.values = values; this.nodes = nodes; this.hash = hash; this.depth = Math.log2(values.length); }
    static create(leaves, hash) {
        const nodes = new Vector(hash.field, leaves.length, undefined, 0n, DIGEST);
        native().call('gs_merkle_build', hash.field.ctx, hash.alg, leaves.ptr, leaves.length, nodes.ptr);
        return new MerkleTree(leaves, nodes, hash);
    }
    get root() { if (!this._root) this._root = this.nodes.toBuffer(1, 1); return this._root; }
    proveBatch(indexes) {
        const r = native().merkleProveBatch(this.hash.field.ctx, this.values.ptr, this.nodes.ptr, this.values.length, indexes);
        const values = indexes.map((_, i) => r.values.slice(i * DIGEST, (i + 1) * DIGEST));
        const nodes = []; let o = 0;
        for (const k of r.colLens) { const col = []; for (let t = 0; t < k; t++, o++) col.push(r.nodes.slice(o * DIGEST, (o + 1) * DIGEST)); nodes.push(col); }
        return { values, nodes, depth: this.depth };
    }
    static verifyBatch(root, indexes, proof, hash) {
        const offset = 2 ** proof.depth, indexMap = new Map();
        indexes.forEach((ix, i) => indexMap.set(ix, i));
        if (indexMap.size !== indexes.length) return false;
        const norm = normalize(indexes);
        if (norm.length !== proof.nodes.length) return false;
        const v = new Map(), ptr = new Array(norm.length).fill(0);
        let next = [];
        for (let i = 0; i < norm.length; i++) {
            const ix = norm[i], i1 = indexMap.get(ix), i2 = indexMap.get(ix + 1);
            let v1, v2;
            if (i1 !== undefined && i2 !== undefined) { v1 = proof.values[i1]; v2 = proof.values[i2]; }
            else if (i1 !== undefined) { v1 = proof.values[i1]; v2 = proof.nodes[i][0]; ptr[i] = 1; }
            else { v1 = proof.nodes[i][0]; v2 = proof.values[i2]; ptr[i] = 1; }
            if (!v1 || !v2) return false;
            const parent = (offset + ix) >> 1;
            v.set(parent, hash.merge(v1, v2)); next.push(parent);
        }
        for (let d = proof.depth - 1; d > 0; d--) {
            const cur = next; next = [];
            for (let i = 0; i < cur.length; i++) {
                const node = cur[i], sib = node ^ 1;
                let s;
                if (i + 1 < cur.length && cur[i + 1] === sib) { s = v.get(sib); i++; }
                else { s = proof.nodes[i][ptr[i]]; ptr[i]++; }
                const me = v.get(node);
                if (!me || !s) return false;
                v.set(node >> 1, (node & 1) ? hash.merge(s, me) : hash.merge(me, s));
                next.push(node >> 1);
            }
        }
        return Buffer.compare(v.get(1), root) === 0;
    }
}

module.exports = { createHash, Hash, MerkleTree };
