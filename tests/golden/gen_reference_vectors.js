// tests/golden/gen_reference_vectors.js — runs the REFERENCE's own compiled modules (those that have no
// absent npm dependency) under node in the build container and records input/output vectors.
//   usage: node gen_reference_vectors.js /root/reference/bin/lib <proofs.json> <out.json>
// Modules exercised (all require only node built-ins or sibling files):
//   lib/components/QueryIndexGenerator.js   (Fiat-Shamir query positions)
//   lib/Serializer.js + lib/utils/{serialization,sizeof,index}.js   (proof wire format, sizeOf, powLog2)
// The output file is data (inputs + expected outputs); it is what travels to the GPU box.
const path = require('path');
const fs = require('fs');
const libDir = process.argv[2], proofsPath = process.argv[3], outPath = process.argv[4];
const { QueryIndexGenerator } = require(path.join(libDir, 'components', 'QueryIndexGenerator.js'));
const { Serializer } = require(path.join(libDir, 'Serializer.js'));
const utils = require(path.join(libDir, 'utils'));
const crypto = require('crypto');

const out = { generator: 'genSTARK bin/lib run under node ' + process.version, queryIndexes: [], serializer: [], powLog2: [], bigint: [] };

// ---- QueryIndexGenerator
const seeds = [];
for (let i = 0; i < 12; i++) seeds.push(crypto.createHash('sha256').update('seed' + i).digest());
seeds.push(Buffer.alloc(32, 0)); seeds.push(Buffer.alloc(32, 0xff)); seeds.push(Buffer.from('00'.repeat(31) + '01', 'hex'));
const shapes = [
    { ef: 16, exe: 48, fri: 24, domain: 1024 }, { ef: 16, exe: 48, fri: 24, domain: 1 << 17 }, { ef: 8, exe: 80, fri: 40, domain: 1 << 16 },
    { ef: 16, exe: 48, fri: 64, domain: 1 << 24 }, { ef: 4, exe: 80, fri: 40, domain: 256 }, { ef: 32, exe: 128, fri: 64, domain: 1 << 20 },
    { ef: 16, exe: 48, fri: 24, domain: 256 }, { ef: 2, exe: 128, fri: 1, domain: 128 },
];
for (const s of shapes) {
    const g = new QueryIndexGenerator({ extensionFactor: s.ef, exeQueryCount: s.exe, friQueryCount: s.fri, hashAlgorithm: 'sha256' });
    for (const seed of seeds) {
        const rec = { seed: seed.toString('hex'), ef: s.ef, exeQueryCount: s.exe, friQueryCount: s.fri, domain: s.domain };
        rec.exe = g.getExeIndexes(seed, s.domain);
        rec.fri = [];
        for (let len = s.domain / 4; len >= 64 && len >= s.ef * 2; len = len / 4) {
            try { rec.fri.push({ columnLength: len, indexes: g.getFriIndexes(seed, len) }); }
            catch (e) { rec.fri.push({ columnLength: len, error: String(e.message) }); }
        }
        out.queryIndexes.push(rec);
    }
}

// ---- Serializer / sizeOf on proofs produced by this repo's oracle (structure only matters here)
const hexToBuf = (h) => Buffer.from(h, 'hex');
function reviveMerkle(p) { return { values: p.values.map(hexToBuf), nodes: p.nodes.map(c => c.map(hexToBuf)), depth: p.depth }; }
function revive(p) {
    return {
        evRoot: hexToBuf(p.evRoot), evProof: reviveMerkle(p.evProof),
        ldProof: {
            lcRoot: hexToBuf(p.ldProof.lcRoot), lcProof: reviveMerkle(p.ldProof.lcProof),
            components: p.ldProof.components.map(c => ({ columnRoot: hexToBuf(c.columnRoot), columnProof: reviveMerkle(c.columnProof), polyProof: reviveMerkle(c.polyProof) })),
            remainder: p.ldProof.remainder.map(BigInt)
        },
        iShapes: p.iShapes
    };
}
function dumpMerkle(p) { return { values: p.values.map(b => b.toString('hex')), nodes: p.nodes.map(c => c.map(b => b.toString('hex'))), depth: p.depth }; }
const proofs = JSON.parse(fs.readFileSync(proofsPath, 'utf8'));
for (const item of proofs) {
    const ser = new Serializer({ field: { elementSize: item.elementSize }, traceRegisterCount: item.traceRegisterCount, secretInputCount: item.secretInputCount }, item.digestSize);
    const proof = revive(item.proof);
    const bytes = ser.serializeProof(proof);
    const size = utils.sizeOf(proof, item.elementSize, item.digestSize);
    const parsed = ser.parseProof(bytes);
    out.serializer.push({
        name: item.name, elementSize: item.elementSize, digestSize: item.digestSize, traceRegisterCount: item.traceRegisterCount,
        secretInputCount: item.secretInputCount, proof: item.proof,
        serialized: bytes.toString('hex'), sizeOfTotal: size.total,
        parsed: {
            evRoot: parsed.evRoot.toString('hex'), evProof: dumpMerkle(parsed.evProof),
            lcRoot: parsed.ldProof.lcRoot.toString('hex'), lcProof: dumpMerkle(parsed.ldProof.lcProof),
            components: parsed.ldProof.components.map(c => ({ columnRoot: c.columnRoot.toString('hex'), columnProof: dumpMerkle(c.columnProof), polyProof: dumpMerkle(c.polyProof) })),
            remainder: parsed.ldProof.remainder.map(String), iShapes: parsed.iShapes
        }
    });
}

// ---- small helpers
for (const [b, e] of [[16 / 3, 48], [8 / 3, 80], [4, 24], [2, 128], [32 / 8, 68], [16 / 5, 48]]) out.powLog2.push({ base: b, exponent: e, value: utils.powLog2(b, e) });
for (const v of [0n, 1n, 2n ** 32n, 2n ** 127n + 12345n, 2n ** 128n - 9n * 2n ** 32n]) {
    const buf = Buffer.alloc(16);
    utils.writeBigInt(v, buf, 0, 16);
    out.bigint.push({ value: String(v), bytes: buf.toString('hex'), back: String(utils.readBigInt(buf, 0, 16)) });
}
fs.writeFileSync(outPath, JSON.stringify(out));
console.log('wrote', outPath, 'queryIndexes', out.queryIndexes.length, 'serializer', out.serializer.length);
