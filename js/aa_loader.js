'use strict';
// js/aa_loader.js — the AirAssembly loader in JavaScript: source text -> the JSON descriptors js/air_assembly.js hands to the
// register-machine AIR of js/air_generic.js (compile / instantiate of `@guildofweavers/air-assembly` as index.ts:4,18-33 and
// lib/Stark.ts:40 use them).  Same requests, same answers as `python3 -m genstark_amd.aa_json` (genstark_amd/airassembly.py +
// air_generic.py: Program), so that the node side needs no Python interpreter: tests/test_airassembly.py compares the two loaders'
// answers object for object on every module of the test tree (and the reference's assembly/*.aa where the checkout is present).
// The language subset, the meaning of the static-register declarations and the degree rules are those airassembly.py documents
// (the compiler itself lives in the absent package: SURVEY 8c).  Integers are BigInt inside and decimal strings in the answers.
//
//     handle({op: 'check' | 'info' | 'describe' | 'plan' | 'verify', source, component, extensionFactor, ...}[, raw]) -> answer object
//     (raw: field elements of the answer stay BigInt — for a caller in this process; inputs may be BigInt, numbers or decimal strings)
//     errors: Error('GstarkError: ...') with the Python loader's wording
const crypto = require('crypto');

const OP = { LOADC: 0, LOADR: 1, LOADN: 2, LOADS: 3, ADD: 4, SUB: 5, MUL: 6, POW: 7, POWC: 8, OUT: 9 };
const MAX_VM_REGS = 64;

function fail(msg) { throw new Error(`GstarkError: ${msg}`); }
const isList = Array.isArray;
const mod = (v, p) => { const r = v % p; return r < 0n ? r + p : r; };
function modpow(b, e, p) {
    let r = 1n; b = mod(b, p);
    while (e > 0n) { if (e & 1n) r = r * b % p; b = b * b % p; e >>= 1n; }
    return r;
}
const bitLength = n => { let k = 0; while (n > 0) { n = Math.floor(n / 2); k++; } return k; };
const isPow2 = n => n >= 1 && Number.isInteger(n) && (n & (n - 1)) === 0;

// ---- reading -----------------------------------------------------------------------------------------------------------------------
function parse(text) {          // S-expressions -> nested arrays of atoms (strings); `#` starts a comment
    const tokens = text.replace(/#[^\n]*/g, '').match(/[()]|[^\s()]+/g) || [];
    let pos = 0;
    function node() {
        if (pos >= tokens.length) fail('AirAssembly source: unexpected end of input');
        if (tokens[pos] !== '(') return tokens[pos++];
        pos++;
        const out = [];
        while (tokens[pos] !== ')') {
            if (pos >= tokens.length) fail('AirAssembly source: unexpected end of input');
            out.push(node());
        }
        pos++;
        return out;
    }
    const out = [];
    while (pos < tokens.length) out.push(node());
    return out;
}
const big = tok => BigInt(tok);                                  // decimal or 0x...
const num = tok => { const n = Number(tok); if (!Number.isInteger(n)) fail(`integer expected, got ${tok}`); return n; };

class Fn {
    constructor(tree) {
        this.name = tree[1]; this.params = []; this.locals = []; this.body = [];
        for (const item of tree.slice(2)) {
            if (isList(item) && item[0] === 'result') continue;
            if (isList(item) && item[0] === 'param') this.params.push(item[1].startsWith('$') ? item[1] : null);
            else if (isList(item) && item[0] === 'local') this.locals.push(item[1].startsWith('$') ? item[1] : null);
            else this.body.push(item);
        }
    }
}

class Export {
    constructor(tree) {
        this.name = tree[1];
        this.registers = this.constraints = this.steps = null;
        this.statics = []; this.init = null; this.transition = null; this.evaluation = null;
        for (const item of tree.slice(2)) {
            const kind = item[0];
            if (kind === 'registers') this.registers = num(item[1]);
            else if (kind === 'constraints') this.constraints = num(item[1]);
            else if (kind === 'steps') this.steps = num(item[1]);
            else if (kind === 'static') this.statics = item.slice(1).map(Export.staticOf);
            else if (kind === 'init') {
                const body = item.slice(1).filter(x => !(isList(x) && x[0] === 'param'));
                const params = item.slice(1).filter(x => isList(x) && x[0] === 'param');
                this.init = { param: params.length ? params[0][1] : null, body };
            } else if (kind === 'transition' || kind === 'evaluation') this[kind] = item.slice(1).filter(x => !(isList(x) && x[0] === 'local'));
            else fail(`export ${this.name}: unknown section ${kind}`);
        }
    }
    static staticOf(s) {
        if (s[0] === 'input') {
            const d = { kind: 'input', secret: s[1] === 'secret', parent: null, peer: null, steps: null, shift: 0 };
            if (s[1] !== 'secret' && s[1] !== 'public') fail('input register: secret or public expected');
            for (const opt of s.slice(2)) {
                if (opt[0] === 'childof') d.parent = num(opt[1]);
                else if (opt[0] === 'peerof') d.peer = num(opt[1]);
                else if (opt[0] === 'steps') d.steps = num(opt[1]);
                else if (opt[0] === 'shift') d.shift = num(opt[1]);
                else fail(`input register: unknown option ${opt[0]}`);
            }
            return d;
        }
        if (s[0] === 'mask') {
            if (!(isList(s[1]) && s[1][0] === 'input')) fail('mask register: (mask (input i)) expected');
            return { kind: 'mask', input: num(s[1][1]) };
        }
        if (s[0] === 'cycle') return { kind: 'cycle', values: s.slice(1) };
        return fail(`unknown static register kind ${s[0]}`);
    }
}

class Module {
    constructor(text) {
        const tree = parse(Buffer.isBuffer(text) ? text.toString('utf8') : String(text));
        if (tree.length !== 1 || tree[0][0] !== 'module') fail('AirAssembly source: one (module ...) expected');
        this.modulus = null; this.constNames = []; this.consts = []; this.functions = new Map(); this.exports = new Map();
        for (const item of tree[0].slice(1)) {
            const kind = item[0];
            if (kind === 'field') {
                if (item[1] !== 'prime') fail('only prime fields are supported');
                this.modulus = big(item[2]);
            } else if (kind === 'const') {
                const named = item[1].startsWith('$');
                this.constNames.push(named ? item[1] : null);
                this.consts.push(item.slice(named ? 2 : 1));
            } else if (kind === 'function') { const fn = new Fn(item); this.functions.set(fn.name, fn); }
            else if (kind === 'export') { const ex = new Export(item); this.exports.set(ex.name, ex); }
            else fail(`module: unknown section ${kind}`);
        }
        if (this.modulus === null) fail('module: no field');
    }
    constValues(algebra) {
        return this.consts.map(spec => {
            if (spec[0] === 'scalar') return algebra.const(big(spec[1]));
            if (spec[0] === 'vector') return spec.slice(1).map(v => algebra.const(big(v)));
            if (spec[0] === 'matrix') return spec.slice(1).map(row => row.map(v => algebra.const(big(v))));
            return fail(`constant of unknown type ${spec[0]}`);
        });
    }
}

// ---- expression DAG nodes and the register-machine program (genstark_amd/air_generic.py: Expr, Program) ------------------------------
class Expr {
    constructor(kind, a, b) { this.kind = kind; this.a = a; this.b = b; }      // const: a = value; reg / next / static: a = index; pow: a = base, b = exponent
}
const wrap = v => (v instanceof Expr ? v : new Expr('const', BigInt(v)));
const exprArgs = e => (e.kind === 'add' || e.kind === 'sub' || e.kind === 'mul') ? [e.a, e.b] : (e.kind === 'pow' ? [e.a] : []);

class Program {
    /** straight-line code {op, dst, a, b} + constant pool, scratch registers reused after their last use; the schedule holds
     *  exponentiations back so that independent ones stand next to each other, grouped by exponent (air_generic.py: Program) */
    constructor(outputs, modulus) {
        this.modulus = modulus;
        let order = [];
        const index = new Map();
        const visit = e => {
            if (index.has(e)) return index.get(e);
            const ins = exprArgs(e).map(visit);
            index.set(e, order.length);
            order.push([e, ins]);
            return index.get(e);
        };
        let outs = outputs.map(o => visit(wrap(o)));
        const level = new Array(order.length).fill(0);
        order.forEach(([e, ins], n) => { level[n] = ins.reduce((m, i) => Math.max(m, level[i]), 0) + (e.kind === 'pow' ? 1 : 0); });
        const groups = new Map();
        const gkey = n => `${level[n]}:${order[n][0].b}`;
        order.forEach(([e], n) => { if (e.kind === 'pow') { if (!groups.has(gkey(n))) groups.set(gkey(n), []); groups.get(gkey(n)).push(n); } });
        const schedule = [], done = new Set();
        const needs = n => Array.from(new Set(order[n][1])).sort((x, y) => x - y);
        const emit = n => {
            if (done.has(n)) return;
            for (const i of needs(n)) emit(i);
            done.add(n);
            schedule.push(n);
        };
        order.forEach(([e], n) => {
            if (done.has(n)) return;
            if (e.kind !== 'pow') { emit(n); return; }
            const members = groups.get(gkey(n)).filter(m => !done.has(m)).slice(0, 8);
            for (const m of members) for (const i of needs(m)) emit(i);
            for (const m of members) { done.add(m); schedule.push(m); }
        });
        if (schedule.length !== order.length) fail('program: scheduling lost a node');
        const position = new Map(schedule.map((old, now) => [old, now]));
        order = schedule.map(old => [order[old][0], order[old][1].map(i => position.get(i))]);
        outs = outs.map(o => position.get(o));
        const lastUse = new Map();
        order.forEach(([, ins], n) => { for (const i of ins) lastUse.set(i, n); });
        for (const o of outs) lastUse.set(o, order.length);
        this.consts = [];
        const constIx = new Map();
        const cidx = (v, raw) => {
            const key = raw ? `raw:${v}` : `${v = mod(v, modulus)}`;
            if (!constIx.has(key)) { constIx.set(key, this.consts.length); this.consts.push(v); }
            return constIx.get(key);
        };
        const free = [], where = new Map(), code = [];
        let nregs = 0;
        order.forEach(([e, ins], n) => {
            const srcs = ins.map(i => where.get(i));
            for (const i of ins) if (lastUse.get(i) === n && !free.includes(where.get(i))) free.push(where.get(i));
            const dst = free.length ? free.pop() : nregs++;
            where.set(n, dst);
            switch (e.kind) {
                case 'const': code.push(OP.LOADC, dst, cidx(e.a, false), 0); break;
                case 'reg': code.push(OP.LOADR, dst, e.a, 0); break;
                case 'next': code.push(OP.LOADN, dst, e.a, 0); break;
                case 'static': code.push(OP.LOADS, dst, e.a, 0); break;
                case 'add': code.push(OP.ADD, dst, srcs[0], srcs[1]); break;
                case 'sub': code.push(OP.SUB, dst, srcs[0], srcs[1]); break;
                case 'mul': code.push(OP.MUL, dst, srcs[0], srcs[1]); break;
                case 'pow':
                    if (e.b < (1n << 32n)) code.push(OP.POW, dst, srcs[0], Number(e.b));
                    else code.push(OP.POWC, dst, srcs[0], cidx(e.b, true));
                    break;
                default: fail(`unknown expression kind ${e.kind}`);
            }
        });
        outs.forEach((o, k) => code.push(OP.OUT, k, where.get(o), 0));
        if (nregs > MAX_VM_REGS) fail(`program needs ${nregs} scratch registers (max ${MAX_VM_REGS})`);
        this.code = code; this.nregs = Math.max(nregs, 1); this.nout = outs.length;
    }
    run(cur, statics) {          // host interpreter on BigInt (transition programs: no next row)
        const p = this.modulus, vm = new Array(this.nregs).fill(0n), out = new Array(this.nout).fill(0n), c = this.code;
        for (let i = 0; i < c.length; i += 4) {
            const d = c[i + 1], a = c[i + 2], b = c[i + 3];
            switch (c[i]) {
                case OP.LOADC: vm[d] = this.consts[a]; break;
                case OP.LOADR: vm[d] = cur[a]; break;
                case OP.LOADS: vm[d] = statics[a]; break;
                case OP.ADD: vm[d] = mod(vm[a] + vm[b], p); break;
                case OP.SUB: vm[d] = mod(vm[a] - vm[b], p); break;
                case OP.MUL: vm[d] = vm[a] * vm[b] % p; break;
                case OP.POW: vm[d] = modpow(vm[a], BigInt(b), p); break;
                case OP.POWC: vm[d] = modpow(vm[a], this.consts[b], p); break;
                case OP.OUT: out[d] = vm[a]; break;
                default: fail('program: a transition program does not read the next row');
            }
        }
        return out;
    }
    toJSON(raw) { return { code: this.code, consts: raw ? this.consts.slice() : this.consts.map(String), nregs: this.nregs, nout: this.nout }; }
}

// ---- evaluation over an algebra ------------------------------------------------------------------------------------------------------
class Ints {            // field elements as BigInt
    constructor(p) { this.p = p; }
    const(v) { return mod(v, this.p); }
    add(a, b) { return mod(a + b, this.p); }
    sub(a, b) { return mod(a - b, this.p); }
    mul(a, b) { return a * b % this.p; }
    pow(a, e) { return modpow(a, e, this.p); }
    inv(a) { return modpow(a, this.p - 2n, this.p); }
}
class Exprs extends Ints {      // expression DAG nodes; constants stay BigInt and fold
    add(a, b) { return (a instanceof Expr || b instanceof Expr) ? new Expr('add', wrap(a), wrap(b)) : super.add(a, b); }
    sub(a, b) { return (a instanceof Expr || b instanceof Expr) ? new Expr('sub', wrap(a), wrap(b)) : super.sub(a, b); }
    mul(a, b) { return (a instanceof Expr || b instanceof Expr) ? new Expr('mul', wrap(a), wrap(b)) : super.mul(a, b); }
    pow(a, e) {
        if (!(a instanceof Expr)) return super.pow(a, e);
        if (e < 0n) fail('negative exponents must be rewritten as e mod (p - 1)');
        return new Expr('pow', a, e);
    }
    inv(a) { return a instanceof Expr ? this.pow(a, this.p - 2n) : super.inv(a); }
}
class Degrees {         // degree of an expression in units of the trace length: registers 1, constants 0
    const() { return 0; }
    add(a, b) { return Math.max(a, b); }
    sub(a, b) { return Math.max(a, b); }
    mul(a, b) { return a + b; }
    pow(a, e) { return a * Number(e); }
    inv(a) {
        if (a) fail('division by a register inside a constraint has no degree: check the slope by cross-multiplication');
        return 0;
    }
}
class Lane { constructor(v) { this.v = v; } }          // one scalar per run of the computation
class Lanes extends Ints {
    lift(fn, a, b) {
        const la = a instanceof Lane, lb = b instanceof Lane;
        if (la && lb) return new Lane(a.v.map((x, i) => fn(x, b.v[i])));
        if (la) return new Lane(a.v.map(x => fn(x, b)));
        if (lb) return new Lane(b.v.map(y => fn(a, y)));
        return fn(a, b);
    }
    add(a, b) { return this.lift((x, y) => mod(x + y, this.p), a, b); }
    sub(a, b) { return this.lift((x, y) => mod(x - y, this.p), a, b); }
    mul(a, b) { return this.lift((x, y) => x * y % this.p, a, b); }
    pow(a, e) { return a instanceof Lane ? new Lane(a.v.map(x => modpow(x, e, this.p))) : modpow(a, e, this.p); }
    inv(a) { return this.pow(a, this.p - 2n); }
}

class Evaluator {
    constructor(module, algebra) { this.m = module; this.a = algebra; this.ints = new Ints(module.modulus); }
    zip(fn, x, y) {
        const vx = isList(x), vy = isList(y);
        if (vx && vy) {
            if (x.length !== y.length) fail(`vector lengths differ: ${x.length} and ${y.length}`);
            return x.map((a, i) => this.zip(fn, a, y[i]));
        }
        if (vx) return x.map(a => this.zip(fn, a, y));
        if (vy) return y.map(b => this.zip(fn, x, b));
        return fn(x, y);
    }
    constant(node) {            // a compile-time integer (exponents)
        let v = new Evaluator(this.m, this.ints).eval(node, { params: {}, locals: new Map() });
        if (isList(v)) {
            if (v.length !== 1) fail('a scalar constant was expected');
            v = v[0];
        }
        return v;
    }
    eval(node, env) {
        const a = this.a;
        if (!isList(node)) fail(`unexpected atom ${node}`);
        const op = node[0];
        switch (op) {
            case 'scalar': return a.const(big(node[1]));
            case 'vector': {
                const out = [];
                for (const item of node.slice(1)) { const v = this.eval(item, env); if (isList(v)) out.push(...v); else out.push(v); }
                return out;
            }
            case 'get': return this.eval(node[1], env)[num(node[2])];
            case 'slice': return this.eval(node[1], env).slice(num(node[2]), num(node[3]) + 1);
            case 'add': case 'sub': case 'mul': { const x = this.eval(node[1], env), y = this.eval(node[2], env); return this.zip((u, v) => a[op](u, v), x, y); }
            case 'div': { const x = this.eval(node[1], env), y = this.eval(node[2], env); return this.zip((u, v) => a.mul(u, a.inv(v)), x, y); }
            case 'exp': {
                const e = this.constantIn(node[2], env), base = this.eval(node[1], env);
                return isList(base) ? base.map(b => a.pow(b, e)) : a.pow(base, e);
            }
            case 'prod': {
                const m = this.eval(node[1], env), v = this.eval(node[2], env);
                if (!(isList(m) && m.length && isList(m[0])) || !isList(v)) fail('prod: matrix x vector expected');
                return m.map(row => {
                    let acc = null;
                    const n = Math.min(row.length, v.length);
                    for (let i = 0; i < n; i++) { const t = a.mul(row[i], v[i]); acc = acc === null ? t : a.add(acc, t); }
                    return acc;
                });
            }
            case 'load.const': return Evaluator.named(this.m.constValues(a), this.m.constNames, node[1], 'constant');
            case 'load.param': return Evaluator.named(env.params.values, env.params.names, node[1], 'parameter');
            case 'load.local':
                if (!env.locals.has(node[1])) fail(`local ${node[1]} read before it is stored`);
                return env.locals.get(node[1]);
            case 'load.trace': return env.trace[num(node[1])];
            case 'load.static': return env.static;
            case 'call': {
                const fn = this.m.functions.get(node[1]);
                if (!fn) fail(`unknown function ${node[1]}`);
                const args = node.slice(2).map(x => this.eval(x, env));
                if (args.length !== fn.params.length) fail(`${fn.name}: ${fn.params.length} arguments expected`);
                return this.run(fn.body, Object.assign({}, env, { params: { values: args, names: fn.params }, locals: new Map() }));
            }
            default: return fail(`unknown operation ${op}`);
        }
    }
    constantIn(node, env) {     // an exponent: a literal, a constant, or a parameter bound to one
        if (node[0] === 'load.param' && env.params.values) {
            const v = Evaluator.named(env.params.values, env.params.names, node[1], 'parameter');
            if (typeof v === 'bigint' && !(this.a instanceof Degrees)) return v;
        }
        return this.constant(node);
    }
    static named(values, names, key, what) {
        if (key.startsWith('$')) {
            if (!names || !names.includes(key)) fail(`unknown ${what} ${key}`);
            return values[names.indexOf(key)];
        }
        return values[num(key)];
    }
    run(body, env) {            // stmt* expr
        for (const stmt of body.slice(0, -1)) {
            if (stmt[0] !== 'store.local') fail(`statement expected, got ${stmt[0]}`);
            env.locals.set(stmt[1], this.eval(stmt[2], env));
        }
        return this.eval(body[body.length - 1], env);
    }
}

// ---- input registers -> columns ------------------------------------------------------------------------------------------------------
function shapeOf(value) {
    const shape = [];
    while (isList(value)) {
        shape.push(value.length);
        if (!value.length) break;
        const nested = isList(value[0]);
        if (value.some(v => isList(v) !== nested || (isList(v) && v.length !== value[0].length))) fail('input register: ragged values');
        value = value[0];
    }
    return shape;
}
const sameList = (a, b) => a.length === b.length && a.every((v, i) => v === b[i]);

class Layout {          // where every value of every input register sits in the trace, from the registers' shapes alone
    constructor(statics, shapes) {
        const inputs = statics.filter(s => s.kind === 'input');
        if (shapes.length !== inputs.length) fail(`${inputs.length} input registers: one entry (shape) for each is needed, got ${shapes.length}`);
        this.inputs = inputs; this.shapes = shapes.map(s => s.slice());
        const depth = [];
        inputs.forEach((d, j) => {
            const ref = d.parent !== null ? d.parent : d.peer;
            if (ref !== null && !(ref >= 0 && ref < j)) fail('input register: childof / peerof must name an earlier input register');
            depth.push(ref === null ? 0 : depth[ref] + (d.parent !== null ? 1 : 0));
            if (this.shapes[j].length !== depth[j] + 1) fail(`input register ${j}: values nested ${depth[j] + 1} deep expected`);
            if (d.peer !== null && !sameList(this.shapes[j], this.shapes[d.peer])) fail(`input register ${j}: the shape of its peer ${d.peer} expected`);
            if (d.parent !== null && !sameList(this.shapes[j].slice(0, -1), this.shapes[d.parent])) fail(`input register ${j}: one list per value of register ${d.parent} expected`);
        });
        this.depth = depth;
        const span = inputs.map(d => d.steps);           // steps one value of a register is held: its own (steps n), or what its children take
        for (let changed = true; changed;) {
            changed = false;
            inputs.forEach((d, j) => {
                if (span[j] !== null && d.parent !== null) {
                    const want = span[j] * this.shapes[j][this.shapes[j].length - 1], root = d.parent;
                    if (span[root] === null) { span[root] = want; changed = true; }
                    else if (span[root] !== want && inputs[root].steps === null) fail('input registers: the children of one register take different numbers of steps');
                }
                if (span[j] === null && d.peer !== null && span[d.peer] !== null) { span[j] = span[d.peer]; changed = true; }
                if (span[j] !== null && d.peer !== null && span[d.peer] === null) { span[d.peer] = span[j]; changed = true; }
            });
        }
        if (inputs.length && span.some(s => s === null)) fail('input registers: cannot tell how many steps a value is held (no (steps n) below it)');
        this.span = span;
        this.length = 0;
        const count = j => this.shapes[j].reduce((c, n) => c * n, 1);
        inputs.forEach((d, j) => {
            if (this.length && count(j) * span[j] !== this.length) fail('input registers imply different trace lengths');
            this.length = count(j) * span[j];
        });
        inputs.forEach((d, j) => { if (count(j) * span[j] !== this.length) fail('input registers imply different trace lengths'); });
        if (inputs.length && (this.length < 2 || !isPow2(this.length))) fail(`the inputs make a trace of ${this.length} steps: a power of 2 is required`);
    }
}

class Col {             // a static column in closed form: col[t] = flat[((t + k) mod period) div span]   (airassembly.py: _Col)
    constructor(flat, span, k) {
        this.flat = flat; this.span = span; this.period = flat.length * span;
        this.k = this.period ? ((k % this.period) + this.period) % this.period : 0;
    }
    at(t) { return this.flat[Math.floor(((t + this.k) % this.period) / this.span)]; }
    shrunk() {          // one period, cut to its shortest power-of-2 period
        let col = new Array(this.period);
        for (let t = 0; t < this.period; t++) col[t] = this.at(t);
        while (col.length > 1 && col.length % 2 === 0) {
            const half = col.length / 2;
            let same = true;
            for (let i = 0; i < half && same; i++) same = col[i] === col[half + i];
            if (!same) break;
            col = col.slice(0, half);
        }
        return col;
    }
}
function flatten(values, depth) {
    let flat = values;
    for (let d = 0; d < depth; d++) {           // (no spread: a register of 10^5 lists would not fit an argument list)
        const next = [];
        for (const group of flat) for (const v of group) next.push(v);
        flat = next;
    }
    return flat;
}

// ---- the AIR of one shape (air_generic.py: GenericAir, what its descriptor() exports) -------------------------------------------------
function rootOfUnityExists(modulus, order) {
    const o = BigInt(order);
    if (order <= 0 || !isPow2(order) || (modulus - 1n) % o) fail(`Order ${order} of root of unity is invalid`);
}

class InnerAir {
    constructor(steps, registers, constraintDegrees, staticRegisters, transition, evaluation, extensionFactor, modulus, secretRegisters, segmentLength) {
        if (segmentLength !== null && (segmentLength < 2 || !isPow2(segmentLength) || steps % segmentLength)) fail('segment length must be a power of 2 dividing the trace length');
        if (!isPow2(steps) || steps < 2) fail('steps must be a power of 2');
        for (const values of staticRegisters) if (!isPow2(values.length) || steps % values.length) fail('static register cycles must be powers of 2 dividing the trace length');
        this.steps = steps; this.registers = registers; this.secretInputCount = secretRegisters; this.segmentLength = segmentLength;
        this.constraintDegrees = constraintDegrees.slice(); this.modulus = modulus;
        const maxDegree = Math.max(...constraintDegrees);
        const cf = 1 << bitLength(maxDegree - 1);
        this.extensionFactor = extensionFactor || (1 << bitLength(2 * maxDegree));
        const ef = this.extensionFactor;
        if (!isPow2(ef) || ef < 2 * cf || ef > 32) fail('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32');
        this.staticRegisters = staticRegisters.map(values => values.map(v => mod(v, modulus)));
        const r = [], n = [], k = [];
        for (let i = 0; i < registers; i++) { r.push(new Expr('reg', i)); n.push(new Expr('next', i)); }
        for (let j = 0; j < staticRegisters.length + secretRegisters; j++) k.push(new Expr('static', j));
        this.transitionProgram = new Program(transition(r, k), modulus);
        this.evaluationProgram = new Program(evaluation(r, n, k), modulus);
        if (this.transitionProgram.nout !== registers || this.evaluationProgram.nout !== constraintDegrees.length) fail('transition must yield one value per register, evaluation one per constraint');
        rootOfUnityExists(modulus, steps * ef);
    }
    /** raw: field elements stay BigInt (a caller in this process: js/air_assembly.js); otherwise decimal strings, as JSON carries them */
    descriptor(firstRows, raw) {
        const str = raw ? (v => v) : String;
        const d = { modulus: String(this.modulus), steps: this.steps, registers: this.registers, constraintDegrees: this.constraintDegrees.slice(),
                    extensionFactor: this.extensionFactor, secretInputCount: this.secretInputCount,
                    staticRegisters: this.staticRegisters.map(values => values.map(str)),
                    transition: this.transitionProgram.toJSON(raw), evaluation: this.evaluationProgram.toJSON(raw), init: null };
        if (this.segmentLength !== null) d.segmentLength = this.segmentLength;
        if (firstRows !== undefined) {
            let rows;
            if (this.segmentLength === null) rows = [firstRows || []];
            else {
                const segments = this.steps / this.segmentLength;
                if (!firstRows || firstRows.length !== segments) fail(`a segmented AIR needs one seed per segment (${segments})`);
                rows = firstRows;
            }
            d.firstRows = rows.map(row => row.map(v => str(mod(v, this.modulus))));
        } else d.seedWidth = this.registers;             // init(seed) of a loaded component is the seed itself
        return d;
    }
}

function sha256Prng(seed, count, modulus) {     // air-assembly `prng.sha256(seed, count, field)`: sha256(uint16_be(i + 1) || seed) mod p (restated: air.py)
    const out = [];
    for (let i = 0; i < count; i++) {
        const head = Buffer.alloc(2);
        head.writeUInt16BE(i + 1);
        out.push(BigInt('0x' + crypto.createHash('sha256').update(Buffer.concat([head, seed])).digest('hex')) % modulus);
    }
    return out;
}

// ---- the AirModule (airassembly.py: AssemblyAir) ---------------------------------------------------------------------------------------
class AssemblyAir {
    constructor(module, component, extensionFactor) {
        this.module = module;
        if (!module.exports.has(component)) fail(`component ${component} is not exported (exports: [${Array.from(module.exports.keys()).sort().map(n => `'${n}'`).join(', ')}])`);
        const ex = this.export = module.exports.get(component);
        this.traceRegisterCount = ex.registers;
        this.inputRegisters = ex.statics.filter(s => s.kind === 'input');
        this.secretInputCount = this.inputRegisters.filter(s => s.secret).length;
        this.where = [];                // lib order -> (public index | secret index)
        let npub = 0, nsec = 0;
        for (const s of ex.statics) {
            if (s.kind === 'input' && s.secret) this.where.push(['secret', nsec++]);
            else this.where.push(['public', npub++]);
        }
        this.publicCount = npub;
        const ones = n => new Array(n).fill(1);
        const degrees = this.run(new Degrees(), ones(ex.registers), ones(ex.registers), ones(ex.statics.length), ex.evaluation);
        if (degrees.length !== ex.constraints) fail(`${component}: the evaluation yields ${degrees.length} values, ${ex.constraints} constraints declared`);
        this.constraintDegrees = degrees.map(d => Math.max(d, 1));
        this.maxConstraintDegree = Math.max(...this.constraintDegrees);
        const cf = 1 << bitLength(this.maxConstraintDegree - 1);
        this.extensionFactor = extensionFactor || (1 << bitLength(2 * this.maxConstraintDegree));
        if (this.extensionFactor < 2 * cf) fail('Extension factor must be a power of 2 at least 2x the constraint degree and at most 32');
        this.cache = [];
        this.evaluation_ = null;
    }
    run(algebra, r, n, k, body) {
        const ev = new Evaluator(this.module, algebra);
        const out = ev.run(body, { params: {}, locals: new Map(), trace: [r.slice(), n === null ? null : n.slice()], static: k.slice() });
        return isList(out) ? out : [out];
    }
    libOrder(k) { return this.where.map(([kind, i]) => kind === 'public' ? k[i] : k[this.publicCount + i]); }
    firstRow(staticsAtStep, seed) {
        const ex = this.export, p = this.module.modulus;
        const ev = new Evaluator(this.module, new Ints(p));
        const env = { params: {}, locals: new Map(), static: staticsAtStep.slice() };
        if (ex.init.param !== null) {
            if (seed === null || seed === undefined || seed.length !== ex.registers) fail(`${ex.name}: the init block takes a seed vector of ${ex.registers} values`);
            env.params = { values: [seed.map(v => mod(v, p))], names: [ex.init.param] };
        }
        let row = ev.run(ex.init.body, env);
        row = isList(row) ? row : [row];
        if (row.length !== ex.registers) fail(`${ex.name}: the init block yields ${row.length} values for ${ex.registers} registers`);
        return row;
    }
    firstRows(cols, starts) {      // the init block on the static registers of every step in `starts` at once (no seed parameter)
        const ex = this.export;
        const ev = new Evaluator(this.module, new Lanes(this.module.modulus));
        let row = ev.run(ex.init.body, { params: {}, locals: new Map(), static: cols.map(c => new Lane(starts.map(t => c.at(t)))) });
        row = isList(row) ? row : [row];
        if (row.length !== ex.registers) fail(`${ex.name}: the init block yields ${row.length} values for ${ex.registers} registers`);
        return starts.map((_, i) => row.map(x => x instanceof Lane ? x.v[i] : x));
    }
    cycle(spec) {
        const p = this.module.modulus;
        if (spec.length === 1 && isList(spec[0])) {
            const g = spec[0];
            if (g[0] === 'prng') {
                if (g[1] !== 'sha256') fail('cycle: only (prng sha256 seed n) is known');
                const seed = g[2].toLowerCase().startsWith('0x') ? g[2].slice(2) : g[2];
                return sha256Prng(Buffer.from(seed, 'hex'), num(g[3]), p);
            }
            if (g[0] === 'power') { const out = []; for (let i = 0; i < num(g[2]); i++) out.push(modpow(big(g[1]), BigInt(i), p)); return out; }
            fail(`cycle: unknown generator ${g[0]}`);
        }
        return spec.map(v => mod(big(v), p));
    }
    inner(length, publicCols, segment) {
        // (the AIRs of the last few shapes, found by comparing the public columns themselves: a key in text would cost more than the plan)
        const same = e => e.length === length && e.segment === segment && e.cols.length === publicCols.length && e.cols.every((c, i) => sameList(c, publicCols[i]));
        const hit = this.cache.find(same);
        let air = hit && hit.air;
        if (!air) {
            const p = this.module.modulus;
            const transition = (r, k) => this.run(new Exprs(p), r, null, this.libOrder(k), this.export.transition);
            const evaluation = (r, n, k) => this.run(new Exprs(p), r, n, this.libOrder(k), this.export.evaluation);
            air = new InnerAir(length, this.traceRegisterCount, this.constraintDegrees, publicCols, transition, evaluation, this.extensionFactor, p, this.secretInputCount, segment);
            if (this.cache.length > 8) this.cache.length = 0;
            this.cache.push({ length, segment, cols: publicCols, air });
        }
        return air;
    }
    get evaluationProgram() {       // the shape-independent constraint evaluator: static k = the k-th PUBLIC static register, the secret ones follow
        if (!this.evaluation_) {
            const n = this.traceRegisterCount, r = [], nx = [], k = [];
            for (let i = 0; i < n; i++) { r.push(new Expr('reg', i)); nx.push(new Expr('next', i)); }
            for (let j = 0; j < this.export.statics.length; j++) k.push(new Expr('static', j));
            this.evaluation_ = new Program(this.run(new Exprs(this.module.modulus), r, nx, this.libOrder(k), this.export.evaluation), this.module.modulus);
        }
        return this.evaluation_;
    }
    staticSources() {
        const sources = [], secret = [], cycles = [];
        let j = 0;
        for (const s of this.export.statics) {
            if (s.kind === 'input') { (s.secret ? secret : sources).push([1, j]); j++; }
            else if (s.kind === 'mask') sources.push([2, s.input]);
            else { sources.push([0, 0]); cycles.push(new Col(this.cycle(s.values), 1, 0).shrunk()); }
        }
        return [sources.concat(secret), cycles];
    }
    lengthWithoutInputs() {
        const ex = this.export;
        return Math.max(ex.steps || 1, ...ex.statics.filter(s => s.kind === 'cycle').map(s => this.cycle(s.values).length));
    }
    /** the static registers in lib order as Col; inputs[j] === null: no values for that input register (a secret one on the verifier's side) */
    columns(layout, inputs) {
        const p = this.module.modulus, cols = [];
        let j = 0;
        for (const s of this.export.statics) {
            if (s.kind === 'input') {
                const values = j < inputs.length ? inputs[j] : null;
                cols.push(values === null || values === undefined ? null : new Col(flatten(values, layout.depth[j]).map(v => mod(v, p)), layout.span[j], -layout.inputs[j].shift));
                j++;
            } else if (s.kind === 'mask') {
                const i = s.input, one = new Array(layout.span[i]).fill(0n);
                one[0] = 1n;
                cols.push(new Col(one, 1, -layout.inputs[i].shift));
            } else cols.push(new Col(this.cycle(s.values), 1, 0));
        }
        return cols;
    }
    publicSplit(cols) { return cols.filter((c, i) => this.where[i][0] === 'public').map(c => c.shrunk()); }
    plan(inputs, seed) {
        const ex = this.export, p = this.module.modulus;
        const layout = new Layout(ex.statics, inputs.map(shapeOf));
        const length = layout.length || this.lengthWithoutInputs();
        const cols = this.columns(layout, inputs);
        const pub = this.publicSplit(cols);
        const secret = cols.filter((c, i) => this.where[i][0] === 'secret');
        const tops = layout.inputs.map((d, j) => j).filter(j => layout.depth[j] === 0);
        const runs = tops.length ? layout.shapes[tops[0]][0] : 1;
        const masked = ex.statics.some(s => s.kind === 'mask' && layout.depth[s.input] === 0);
        let segment = (runs > 1 && masked && ex.init.param === null) ? length / runs : null;
        let air = this.inner(length, pub, segment);
        const packed = secret.map(c => c.shrunk());
        let firsts;
        if (segment === null) firsts = this.firstRow(cols.map(c => c.at(0)), seed);
        else {
            const starts = [];
            for (let s = 0; s < runs; s++) starts.push(s * segment);
            firsts = this.firstRows(cols, starts);
            // the restart the segmentation relies on: the row the transition produces on the last step of run 0 is run 1's first row
            let row = firsts[0].map(v => mod(v, p));
            for (let i = 0; i < segment; i++) row = air.transitionProgram.run(row, air.staticRegisters.map(v => v[i % v.length]).concat(secret.map(c => c.at(i))));
            if (!sameList(row, firsts[1].map(v => mod(v, p)))) { air = this.inner(length, pub, null); firsts = firsts[0]; segment = null; }
        }
        return { air, packed, firsts, shapes: layout.shapes.map(s => s.slice()) };
    }
}

// ---- the requests of js/air_assembly.js (genstark_amd/aa_json.py: handle) --------------------------------------------------------------
const toBig = x => isList(x) ? x.map(toBig) : BigInt(x);

// parsed modules and their components' AirModules are kept (8 most recent sources): a prover asks for a plan per proof, and the
// programs of a component do not depend on the request
const modules = new Map();
function moduleOf(source) {
    const key = Buffer.isBuffer(source) ? source.toString('utf8') : String(source);
    let hit = modules.get(key);
    if (hit) { modules.delete(key); modules.set(key, hit); return hit; }
    hit = { module: new Module(key), airs: new Map() };
    modules.set(key, hit);
    if (modules.size > 8) modules.delete(modules.keys().next().value);
    return hit;
}
function airOf(source, component, extensionFactor) {
    const hit = moduleOf(source), key = `${component}|${extensionFactor || 0}`;
    if (!hit.airs.has(key)) hit.airs.set(key, new AssemblyAir(hit.module, component, extensionFactor));
    return hit.airs.get(key);
}

function handle(req, raw) {
    const module = moduleOf(req.source).module;
    const str = raw ? (v => v) : String;
    if (req.op === 'check') {
        const out = {};
        for (const [name, ex] of module.exports) {
            const inputs = ex.statics.filter(s => s.kind === 'input');
            out[name] = { registers: ex.registers, constraints: ex.constraints, inputs: inputs.length, secretInputs: inputs.filter(s => s.secret).length };
        }
        return { modulus: String(module.modulus), exports: out };
    }
    const air = airOf(req.source, req.component || 'default', req.extensionFactor || null);
    if (req.op === 'info') {
        const out = { traceRegisterCount: air.traceRegisterCount, secretInputCount: air.secretInputCount, constraintDegrees: air.constraintDegrees,
                      maxConstraintDegree: air.maxConstraintDegree, extensionFactor: air.extensionFactor, inputRegisters: air.inputRegisters.length };
        if (air.inputRegisters.length) {
            const [sources, cycles] = air.staticSources();
            Object.assign(out, { inputDeclarations: air.inputRegisters.map(d => ({ parent: d.parent, peer: d.peer, steps: d.steps || 0, shift: d.shift, secret: !!d.secret })),
                                 staticSources: sources, cycles: cycles.map(c => c.map(str)), evaluation: air.evaluationProgram.toJSON(raw) });
        }
        return out;
    }
    if (req.op === 'describe') {
        if (air.inputRegisters.length) throw new Error('ValueError: the component has input registers: its trace is sized when the inputs arrive');
        const layout = new Layout(air.export.statics, []);
        return { descriptor: air.inner(air.lengthWithoutInputs(), air.publicSplit(air.columns(layout, [])), null).descriptor(undefined, raw) };
    }
    if (req.op === 'plan') {
        const seed = req.seed === null || req.seed === undefined ? null : toBig(req.seed);
        const plan = air.plan(toBig(req.inputs || []), seed);
        const d = plan.air.descriptor(plan.firsts, raw);
        d.secretRegisters = plan.packed.map(col => col.map(str));            // this proof's secret columns (one period each)
        const answer = { descriptor: d, inputShapes: plan.shapes };
        // (a caller in this process may key what it builds from the descriptor's shape-dependent part — programs, public static
        //  registers — by the inner AIR itself: the same object for every proof of the same public columns)
        if (raw) { answer.innerAir = plan.air; answer.firstRows = plan.air.segmentLength === null ? [plan.firsts] : plan.firsts; answer.secretColumns = plan.packed; }
        return answer;
    }
    if (req.op === 'verify') {
        const shapes = (req.inputShapes || []).map(s => s.map(Number));
        const layout = new Layout(air.export.statics, shapes);
        const length = layout.length || air.lengthWithoutInputs();
        const given = toBig(req.publicInputs || []), values = [];
        let j = 0;
        for (const d of layout.inputs) {
            if (d.secret) values.push(null);
            else {
                if (j >= given.length) fail(`${air.export.name}: the values of ${layout.inputs.filter(x => !x.secret).length} public input registers are needed`);
                values.push(given[j++]);
            }
        }
        return { descriptor: air.inner(length, air.publicSplit(air.columns(layout, values)), null).descriptor(undefined, raw) };
    }
    throw new Error(`ValueError: unknown op '${req.op}'`);
}

module.exports = { handle, Module, AssemblyAir, Program, parse };
