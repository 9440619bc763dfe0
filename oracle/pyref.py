"""
oracle/pyref.py — TEST INFRASTRUCTURE (CPU oracle, pure Python big-int + hashlib).

A deliberately plain, list-of-ints restatement of genSTARK's prove()/verify()/serialize() for the
MiMC AIR, for SMALL traces only (T <= 2^10).  It exists so that the C oracle (oracle_abi.c) and the
product's host mirror (genstark_amd/) can be checked against a third, structurally different
implementation: Python `int` arithmetic cannot get a carry wrong and `hashlib` is the hash pin.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What each block follows (paths relative to the genSTARK checkout):
  * prove()            lib/Stark.ts:81-163
  * verify()           lib/Stark.ts:167-248
  * composition poly   lib/components/CompositionPolynomial.ts:29-146 (prove), :150-191 (verify)
  * boundary           lib/components/BoundaryConstraints.ts:15-95
  * zero poly          lib/components/ZeroPolynomial.ts:15-44
  * linear combination lib/components/LinearCombination.ts:21-88
  * FRI                lib/components/LowDegreeProver.ts:39-309
  * query positions    lib/components/QueryIndexGenerator.ts:20-67 (exact, incl. the odd-hex quirk)
  * wire format        lib/Serializer.ts:35-144, lib/utils/serialization.ts, lib/utils/sizeof.ts

parity unpinned: the arithmetic packages (@guildofweavers/galois 0.4.22, merkle 0.3.12,
air-assembly 0.3.6; package-lock.json:12-46) are absent from the checkout and cannot be run, so
the pieces whose exact definition lives only there are restated from their documented behaviour
and marked UNVERIFIED below (prng, getRootOfUnity, MiMC round-constant generator, Merkle batch
proof node order).  Everything else (NTT values, pointwise ops, hashes, Lagrange interpolation) is
uniquely determined mathematically.
"""
import hashlib

P128 = 2**128 - 9 * 2**32 + 1  # examples/mimc/mimc128.ts:13
ELEMENT_SIZE = 16               # wasm 128-bit field: two LE u64 limbs (serialization.ts:131-146 reads LE)
DIGEST_SIZE = 32
MAX_REMAINDER_LENGTH = 256      # LowDegreeProver.ts:12


# ------------------------------------------------------------------------------------------------
# hashing
def hash_fn(alg):
    if alg == 'sha256':
        return lambda b: hashlib.sha256(b).digest()
    if alg == 'blake2s256':
        return lambda b: hashlib.blake2s(b, digest_size=32).digest()
    raise TypeError(f'Hash algorithm {alg} is not supported')  # lib/Stark.ts:334-336


def sha256_int(value):
    """QueryIndexGenerator.ts:61-67 — sha256 of a Buffer, or of a bigint rendered as
    Buffer.from(value.toString(16), 'hex'): no leading zeros, and node drops a trailing odd nibble."""
    if isinstance(value, int):
        h = format(value, 'x')
        value = bytes.fromhex(h[: len(h) // 2 * 2])
    return int.from_bytes(hashlib.sha256(value).digest(), 'big')


# ------------------------------------------------------------------------------------------------
# field (galois FiniteField, restated)
class Field:
    def __init__(self, p=P128):
        self.p = p

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return (a * b) % self.p
    def neg(self, a): return (-a) % self.p
    def exp(self, b, e): return pow(b, e, self.p) if e >= 0 else pow(pow(b, -1, self.p), -e, self.p)
    def inv(self, a): return pow(a, self.p - 2, self.p) if a else 0
    def div(self, a, b): return a * self.inv(b) % self.p

    def prng(self, seed, n=None):
        """UNVERIFIED (galois): prng(seed) = sha256(seed) mod p; prng(seed, n): chained sha256 state."""
        if n is None:
            return sha256_int(seed) % self.p
        out, state = [], sha256_int(seed)
        for _ in range(n):
            out.append(state % self.p)
            state = sha256_int(state)
        return out

    def get_root_of_unity(self, order):
        """UNVERIFIED (galois): first g = i^((p-1)/order), i = 2,3,..., whose order is exactly `order`."""
        assert order & (order - 1) == 0 and (self.p - 1) % order == 0
        for i in range(2, 1000):
            g = pow(i, (self.p - 1) // order, self.p)
            if pow(g, order, self.p) == 1 and (order == 1 or pow(g, order // 2, self.p) != 1):
                return g
        raise ValueError('no root of unity found')

    def power_series(self, base, n):
        out, x = [], 1
        for _ in range(n):
            out.append(x)
            x = x * base % self.p
        return out

    def ntt(self, coeffs, omega, n):
        """values of the polynomial at omega^0..omega^(n-1) (zero-extended), recursive radix-2."""
        a = list(coeffs) + [0] * (n - len(coeffs))
        return self._fft(a, omega)

    def _fft(self, a, w):
        n = len(a)
        if n == 1:
            return a
        ev, od = self._fft(a[0::2], w * w % self.p), self._fft(a[1::2], w * w % self.p)
        out, x = [0] * n, 1
        for i in range(n // 2):
            t = x * od[i] % self.p
            out[i] = (ev[i] + t) % self.p
            out[i + n // 2] = (ev[i] - t) % self.p
            x = x * w % self.p
        return out

    def intt(self, values, omega):
        n = len(values)
        ninv = self.inv(n)
        return [v * ninv % self.p for v in self._fft(list(values), self.inv(omega))]

    def batch_inv(self, xs):
        """0^-1 := 0 (galois multi-inverse convention, UNVERIFIED)."""
        return [self.inv(x) for x in xs]

    def eval_poly_at(self, poly, x):
        s = 0
        for c in reversed(poly):
            s = (s * x + c) % self.p
        return s

    def mul_polys(self, a, b):
        out = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % self.p
        return out

    def interpolate(self, xs, ys):
        """Lagrange interpolation; len(xs) coefficients, ascending."""
        n = len(xs)
        out = [0] * n
        for j in range(n):
            num, den = [1], 1
            for m in range(n):
                if m != j:
                    num = self.mul_polys(num, [self.neg(xs[m]), 1])
                    den = den * (xs[j] - xs[m]) % self.p
            s = ys[j] * self.inv(den) % self.p
            for d in range(n):
                out[d] = (out[d] + num[d] * s) % self.p
        return out


def to_bytes(v, size=ELEMENT_SIZE):  # serialization.ts:140-146 (LE 32-bit limbs == plain LE)
    return v.to_bytes(size, 'little')


def from_bytes(b):  # serialization.ts:131-138
    return int.from_bytes(b, 'little')


# ------------------------------------------------------------------------------------------------
# Merkle tree (merkle package, restated; batch-proof node ORDER is UNVERIFIED)
class MerkleTree:
    def __init__(self, leaves, H):
        n = len(leaves)
        assert n >= 2 and n & (n - 1) == 0
        self.leaves, self.H, self.depth = leaves, H, n.bit_length() - 1
        nodes = [b'\0' * 32] * n
        for i in range(n // 2):
            nodes[n // 2 + i] = H(leaves[2 * i] + leaves[2 * i + 1])
        for i in range(n // 2 - 1, 0, -1):
            nodes[i] = H(nodes[2 * i] + nodes[2 * i + 1])
        self.nodes = nodes

    @property
    def root(self):
        return self.nodes[1]

    def prove_batch(self, indexes):
        n = len(self.leaves)
        index_map = {}
        for i, ix in enumerate(indexes):
            assert 0 <= ix < n
            index_map[ix] = i
        assert len(index_map) == len(indexes), 'Repeating indexes detected'
        norm = list(dict.fromkeys(ix - (ix & 1) for ix in sorted(indexes)))
        values = [None] * len(indexes)
        nodes = [None] * len(norm)
        nxt = []
        for i, ix in enumerate(norm):
            v1, v2 = self.leaves[ix], self.leaves[ix + 1]
            i1, i2 = index_map.get(ix), index_map.get(ix + 1)
            if i1 is not None:
                values[i1] = v1
                if i2 is not None:
                    values[i2] = v2
                    nodes[i] = []
                else:
                    nodes[i] = [v2]
            else:
                values[i2] = v2
                nodes[i] = [v1]
            nxt.append((ix + n) >> 1)
        for _ in range(self.depth - 1, 0, -1):
            cur, nxt, i = nxt, [], 0
            while i < len(cur):
                sib = cur[i] ^ 1
                if i + 1 < len(cur) and cur[i + 1] == sib:
                    i += 1
                else:
                    nodes[i].append(self.nodes[sib])
                nxt.append(sib >> 1)
                i += 1
        return {'values': values, 'nodes': nodes, 'depth': self.depth}

    @staticmethod
    def verify_batch(root, indexes, proof, H):
        offset = 1 << proof['depth']
        index_map = {}
        for i, ix in enumerate(indexes):
            if not (0 <= ix < offset):
                return False
            index_map[ix] = i
        if len(index_map) != len(indexes):
            return False
        norm = list(dict.fromkeys(ix - (ix & 1) for ix in sorted(indexes)))
        if len(norm) != len(proof['nodes']):
            return False
        v, nxt, ptr = {}, [], [0] * len(norm)
        try:
            for i, ix in enumerate(norm):
                i1, i2 = index_map.get(ix), index_map.get(ix + 1)
                if i1 is not None:
                    if i2 is not None:
                        v1, v2 = proof['values'][i1], proof['values'][i2]
                    else:
                        v1, v2 = proof['values'][i1], proof['nodes'][i][0]
                        ptr[i] = 1
                else:
                    v1, v2 = proof['nodes'][i][0], proof['values'][i2]
                    ptr[i] = 1
                parent = (offset + ix) >> 1
                v[parent] = H(v1 + v2)
                nxt.append(parent)
            for _ in range(proof['depth'] - 1, 0, -1):
                cur, nxt, i = nxt, [], 0
                while i < len(cur):
                    node_ix = cur[i]
                    sib_ix = node_ix ^ 1
                    if i + 1 < len(cur) and cur[i + 1] == sib_ix:
                        sib = v[sib_ix]
                        i += 1
                        # the slot bookkeeping follows the position of the FIRST node of the pair
                        k = i - 1
                    else:
                        k = i
                        sib = proof['nodes'][k][ptr[k]]
                        ptr[k] += 1
                    node = v[node_ix]
                    parent = H(sib + node) if node_ix & 1 else H(node + sib)
                    v[node_ix >> 1] = parent
                    nxt.append(node_ix >> 1)
                    i += 1
        except (IndexError, KeyError, TypeError):
            return False
        return v.get(1) == root


# ------------------------------------------------------------------------------------------------
# query index generator — QueryIndexGenerator.ts:32-59
def pseudorandom_indexes(seed, count, mx, exclude_multiples_of=0):
    max_count = mx - mx // exclude_multiples_of if exclude_multiples_of else mx
    if max_count < count:
        raise ValueError(f'Cannot select {count} unique pseudorandom indexes from {mx} values')
    indexes = {}
    state = sha256_int(seed)
    for i in range(count * 1000):
        index = sha256_int(state + i) % mx
        if exclude_multiples_of and index % exclude_multiples_of == 0:
            continue
        if index in indexes:
            continue
        indexes[index] = True
        if len(indexes) >= count:
            break
    if len(indexes) < count:
        raise ValueError(f'Could not generate {count} pseudorandom indexes')
    return list(indexes)


# ------------------------------------------------------------------------------------------------
# MiMC AIR (air-assembly instance of examples/mimc/mimc128Assembly.ts:28-51)
def mimc_round_constants(field, count=64, seed=bytes.fromhex('4d694d43')):
    """UNVERIFIED (air-assembly `prng.sha256`, examples/mimc/mimc128.ts:15): value_i =
    sha256(uint16_be(i + 1) || seed) mod p."""
    return [int.from_bytes(hashlib.sha256((i + 1).to_bytes(2, 'big') + seed).digest(), 'big') % field.p
            for i in range(count)]


def run_mimc(field, steps, rc, seed):  # examples/mimc/utils.ts:7-15
    out = [seed]
    for i in range(steps - 1):
        out.append((pow(out[i], 3, field.p) + rc[i % len(rc)]) % field.p)
    return out


class MimcConfig:
    def __init__(self, steps, extension_factor=16, exe_query_count=48, fri_query_count=24,
                 hash_algorithm='blake2s256', seed=3, n_constants=64):
        self.steps, self.E = steps, extension_factor
        self.exe_q, self.fri_q, self.alg = exe_query_count, fri_query_count, hash_algorithm
        self.seed, self.n_constants = seed, n_constants
        self.field = Field()
        self.N = steps * extension_factor
        self.max_degree = 3
        self.comp_factor = 4                       # 2^ceil(log2(3)), CompositionPolynomial.ts:196-204
        self.Nc = steps * self.comp_factor
        self.root = self.field.get_root_of_unity(self.N)
        self.rc = mimc_round_constants(self.field, n_constants)
        self.H = hash_fn(hash_algorithm)

    # cyclic static register as a polynomial in x^(T/len): air-assembly cyclic register semantics
    def k_poly(self):
        f = self.field
        g = f.exp(self.root, self.E * (self.steps // self.n_constants))
        return f.intt(self.rc, g)


def _composition_setup(cfg, assertions, ev_root):
    """CompositionPolynomial ctor (:29-61) + BoundaryConstraints ctor (:15-45) + ZeroPolynomial ctor."""
    f, T = cfg.field, cfg.steps
    combination_degree = cfg.comp_factor * T
    composition_degree = max(combination_degree - T, T)
    groups = [(cfg.max_degree * T, [0])]
    d_count = 1 + sum(len(ix) for deg, ix in groups if deg < combination_degree)
    # boundary constraints grouped by register, insertion order
    rdata = {}
    for a in assertions:
        x = f.exp(cfg.root, a['step'] * cfg.E)
        z = [f.neg(x), 1]
        if a['register'] in rdata:
            d = rdata[a['register']]
            d['xs'].append(x); d['ys'].append(a['value']); d['z'] = f.mul_polys(d['z'], z)
        else:
            rdata[a['register']] = {'xs': [x], 'ys': [a['value']], 'z': z}
    bpolys = [(reg, f.interpolate(d['xs'], d['ys']), d['z']) for reg, d in rdata.items()]
    b_count = len(bpolys) * (2 if composition_degree > T else 1)
    coeffs = f.prng(ev_root, d_count + b_count)
    x_last = f.exp(cfg.root, (T - 1) * cfg.E)
    return dict(combination_degree=combination_degree, composition_degree=composition_degree, groups=groups,
                d_coeffs=coeffs[:d_count], b_coeffs=coeffs[d_count:], bpolys=bpolys, x_last=x_last,
                coefficient_count=d_count + b_count)


def _fri_augment(positions, column_length):  # LowDegreeProver.ts:302-309
    row_length = column_length // 4
    return list(dict.fromkeys(p % row_length for p in positions))


def _transpose4(v):  # transposeVector(v, 4): row r = {v[r], v[r+n/4], v[r+n/2], v[r+3n/4]}
    r = len(v) // 4
    return [[v[i], v[i + r], v[i + 2 * r], v[i + 3 * r]] for i in range(r)]


def _row_bytes(row):
    return b''.join(to_bytes(x) for x in row)


def _verify_remainder(cfg, remainder, max_degree_plus1, root_of_unity):  # LowDegreeProver.ts:223-252
    f = cfg.field
    positions = [i for i in range(len(remainder)) if not cfg.E or i % cfg.E]
    domain = f.power_series(root_of_unity, len(remainder))
    xs = [domain[positions[i]] for i in range(max_degree_plus1)]
    ys = [remainder[positions[i]] for i in range(max_degree_plus1)]
    poly = f.interpolate(xs, ys)
    for i in range(max_degree_plus1, len(positions)):
        p = positions[i]
        if f.eval_poly_at(poly, domain[p]) != remainder[p]:
            raise ValueError(f'Remainder is not a valid degree {max_degree_plus1 - 1} polynomial')


def prove(cfg, assertions, trace=None):
    """lib/Stark.ts:81-163.  Returns (proof dict, info dict with intermediate roots/positions)."""
    f, T, E, N, Nc, H = cfg.field, cfg.steps, cfg.E, cfg.N, cfg.Nc, cfg.H
    w = cfg.root
    domain = f.power_series(w, N)
    info = {}
    # 2 ----- execution trace
    if trace is None:
        trace = run_mimc(f, T, cfg.rc, cfg.seed)
    for a in assertions:  # validateAssertions :356-375
        if trace[a['step']] != a['value']:
            raise ValueError(f"Assertion at step {a['step']}, register {a['register']} conflicts with execution trace")
    # 3 ----- P(x) and LDE
    p_poly = f.intt(trace, f.exp(w, E))
    p_ev = f.ntt(p_poly, w, N)
    # 4 ----- evaluation tree
    e_leaves = [H(to_bytes(v)) for v in p_ev]  # mergeVectorRows over [P0] (R=1, S=0)
    e_tree = MerkleTree(e_leaves, H)
    info['evRoot'] = e_tree.root
    # 5 ----- composition polynomial
    cs = _composition_setup(cfg, assertions, e_tree.root)
    wc = f.exp(w, N // Nc)
    p_comp = f.ntt(p_poly, wc, Nc)
    kp = cfg.k_poly()
    k_period = cfg.n_constants * cfg.comp_factor
    k_tab = f.ntt(kp, f.exp(wc, T // cfg.n_constants), k_period)
    shift = Nc // T
    q = [(p_comp[(j + shift) % Nc] - (pow(p_comp[j], 3, f.p) + k_tab[j % k_period])) % f.p for j in range(Nc)]
    qa = [q]
    for deg, idx in cs['groups']:
        if deg == cs['combination_degree']:
            continue
        powers = f.power_series(f.exp(wc, cs['combination_degree'] - deg), Nc)
        for i in idx:
            qa.append([a * b % f.p for a, b in zip(qa[i], powers)])
    qc = [sum(v[i] * k for v, k in zip(qa, cs['d_coeffs'])) % f.p for i in range(Nc)]
    qe = f.ntt(f.intt(qc, wc), w, N)
    num = [(domain[(i * T) % N] - 1) % f.p for i in range(N)]
    den = [(domain[i] - cs['x_last']) % f.p for i in range(N)]
    z_inv = [d * ni % f.p for d, ni in zip(den, f.batch_inv(num))]  # NB: den/num (CompositionPolynomial.ts:117)
    d_ev = [a * b % f.p for a, b in zip(qe, z_inv)]
    ba = []
    for reg, ipoly, zpoly in cs['bpolys']:
        iv, zv = f.ntt(ipoly, w, N), f.ntt(zpoly, w, N)
        zi = f.batch_inv(zv)
        ba.append([(p_ev[i] - iv[i]) * zi[i] % f.p for i in range(N)])
    b_inc = cs['composition_degree'] - T
    if b_inc > 0:
        psb = f.power_series(f.exp(w, b_inc), N)
        for i in range(len(cs['bpolys'])):
            ba.append([a * b % f.p for a, b in zip(ba[i], psb)])
    bc = [sum(v[i] * k for v, k in zip(ba, cs['b_coeffs'])) % f.p for i in range(N)]
    c_ev = [(a + b) % f.p for a, b in zip(d_ev, bc)]
    # 6 ----- linear combination (LinearCombination.ts:36-64)
    ps = [p_ev]
    ps_inc = cs['composition_degree'] - T
    if ps_inc > 0:
        pw = f.power_series(f.exp(w, ps_inc), N)
        ps = ps + [[a * b % f.p for a, b in zip(v, pw)] for v in ps]
    lc_coeffs = f.prng(e_tree.root, cs['coefficient_count'] + len(ps))[cs['coefficient_count']:]
    l_ev = [(c_ev[i] + sum(v[i] * k for v, k in zip(ps, lc_coeffs))) % f.p for i in range(N)]
    info['lEvaluations'] = l_ev
    # 7 ----- low degree proof (LowDegreeProver.ts:39-68)
    poly_values = _transpose4(l_ev)
    p_tree = MerkleTree([H(_row_bytes(r)) for r in poly_values], H)
    exe_positions = pseudorandom_indexes(p_tree.root, min(cfg.exe_q, N - N // E), N, E)
    lc_positions = _fri_augment(exe_positions, N)
    lc_proof = p_tree.prove_batch(lc_positions)
    lc_proof['values'] = [_row_bytes(poly_values[i]) for i in lc_positions]
    ld = {'lcRoot': p_tree.root, 'lcProof': lc_proof, 'components': [], 'remainder': []}
    info['lcRoot'] = p_tree.root
    info['columnRoots'] = []

    def fri(p_tree, poly_values, max_degree_plus1, depth):  # :176-221
        if len(poly_values) * 4 <= MAX_REMAINDER_LENGTH:
            rou = f.exp(domain[1], 4 ** depth)
            remainder = [poly_values[r][c] for c in range(4) for r in range(len(poly_values))]
            _verify_remainder(cfg, remainder, max_degree_plus1, rou)
            ld['remainder'] = remainder
            return
        rows = len(poly_values)
        step = 4 ** depth
        special_x = f.prng(p_tree.root)
        column = []
        for r in range(rows):
            xs = [domain[(r + c * rows) * step] for c in range(4)]
            column.append(f.eval_poly_at(f.interpolate(xs, poly_values[r]), special_x))
        new_values = _transpose4(column)
        c_tree = MerkleTree([H(_row_bytes(r)) for r in new_values], H)
        info['columnRoots'].append(c_tree.root)
        fri(c_tree, new_values, max_degree_plus1 // 4, depth + 1)
        positions = pseudorandom_indexes(c_tree.root, cfg.fri_q, len(column), E)
        aug = _fri_augment(positions, len(column))
        column_proof = c_tree.prove_batch(aug)
        column_proof['values'] = [_row_bytes(new_values[i]) for i in aug]
        poly_proof = p_tree.prove_batch(positions)
        poly_proof['values'] = [_row_bytes(poly_values[i]) for i in positions]
        while len(ld['components']) <= depth:
            ld['components'].append(None)
        ld['components'][depth] = {'columnRoot': c_tree.root, 'columnProof': column_proof, 'polyProof': poly_proof}

    fri(p_tree, poly_values, cs['composition_degree'], 0)
    # 8 ----- evaluation spot checks (lib/Stark.ts:147-152, 274-296)
    positions = pseudorandom_indexes(ld['lcRoot'], min(cfg.exe_q, N - N // E), N, E)
    aug = list(dict.fromkeys(x for p in positions for x in (p, (p + E) % N)))
    ev_proof = e_tree.prove_batch(aug)
    ev_proof['values'] = [to_bytes(p_ev[i]) for i in aug]
    info['exePositions'] = positions
    proof = {'evRoot': e_tree.root, 'evProof': ev_proof, 'ldProof': ld, 'iShapes': []}
    return proof, info


# ------------------------------------------------------------------------------------------------
def _rehash(proof, H):  # lib/utils/index.ts:34-45
    return {'values': [H(v) for v in proof['values']], 'nodes': proof['nodes'], 'depth': proof['depth']}


def _parse_column_values(buffers, positions, aug, column_length):  # LowDegreeProver.ts:270-282
    row_length = column_length // 4
    out = []
    for p in positions:
        buf = buffers[aug.index(p % row_length)]
        off = (p // row_length) * ELEMENT_SIZE
        out.append(from_bytes(buf[off:off + ELEMENT_SIZE]))
    return out


def verify(cfg, assertions, proof):
    """lib/Stark.ts:167-248; raises ValueError on any failed check, returns True otherwise."""
    f, T, E, N, H = cfg.field, cfg.steps, cfg.E, cfg.N, cfg.H
    w = cfg.root
    e_root = proof['evRoot']
    cs = _composition_setup(cfg, assertions, e_root)
    positions = pseudorandom_indexes(proof['ldProof']['lcRoot'], min(cfg.exe_q, N - N // E), N, E)
    aug = list(dict.fromkeys(x for p in positions for x in (p, (p + E) % N)))
    p_at = {pos: from_bytes(proof['evProof']['values'][i][:ELEMENT_SIZE]) for i, pos in enumerate(aug)}
    if not MerkleTree.verify_batch(e_root, aug, _rehash(proof['evProof'], H), H):
        raise ValueError('Verification of evaluation Merkle proof failed')
    kp = cfg.k_poly()
    lc_coeffs = f.prng(e_root, cs['coefficient_count'] + 2)[cs['coefficient_count']:]
    lc_values = []
    for step in positions:
        x = f.exp(w, step)
        p, n = p_at[step], p_at[(step + E) % N]
        # CompositionPolynomial.evaluateAt :150-191
        k = f.eval_poly_at(kp, f.exp(x, T // cfg.n_constants))
        qv = [(n - (pow(p, 3, f.p) + k)) % f.p]
        for deg, idx in cs['groups']:
            if deg == cs['combination_degree']:
                continue
            pw = f.exp(x, cs['combination_degree'] - deg)
            qv += [qv[i] * pw % f.p for i in idx]
        qc = sum(a * b for a, b in zip(qv, cs['d_coeffs'])) % f.p
        z = f.div(f.sub(f.exp(x, T), 1), f.sub(x, cs['x_last']))
        dv = f.div(qc, z)
        bv = [f.div(f.sub(p, f.eval_poly_at(ip, x)), f.eval_poly_at(zp, x)) for reg, ip, zp in cs['bpolys']]
        b_inc = cs['composition_degree'] - T
        if b_inc > 0:
            pw = f.exp(x, b_inc)
            bv += [bv[i] * pw % f.p for i in range(len(cs['bpolys']))]
        c_value = (dv + sum(a * b for a, b in zip(bv, cs['b_coeffs']))) % f.p
        # LinearCombination.computeOne :66-88
        ps = [p]
        if b_inc > 0:
            ps += [p * f.exp(x, b_inc) % f.p]
        lc_values.append((c_value + sum(a * b for a, b in zip(ps, lc_coeffs))) % f.p)
    _ld_verify(cfg, proof['ldProof'], lc_values, positions, cs['composition_degree'])
    return True


def _ld_verify(cfg, ld, lc_values, exe_positions, max_degree_plus1):  # LowDegreeProver.ts:70-172
    f, H, E = cfg.field, cfg.H, cfg.E
    rou = cfg.root
    column_length = 1
    t = rou
    while t != 1:
        column_length *= 2
        t = t * t % f.p
    quartic = [1, f.exp(rou, column_length // 4), f.exp(rou, column_length // 2), f.exp(rou, column_length * 3 // 4)]
    lc_positions = _fri_augment(exe_positions, column_length)
    lc_checks = _parse_column_values(ld['lcProof']['values'], exe_positions, lc_positions, column_length)
    if not MerkleTree.verify_batch(ld['lcRoot'], lc_positions, _rehash(ld['lcProof'], H), H):
        raise ValueError('Verification of linear combination Merkle proof failed')
    if lc_values != lc_checks:
        raise ValueError('Verification of linear combination correctness failed')
    p_root = ld['lcRoot']
    column_length //= 4
    for depth, comp in enumerate(ld['components']):
        positions = pseudorandom_indexes(comp['columnRoot'], cfg.fri_q, column_length, E)
        aug = _fri_augment(positions, column_length)
        column_values = _parse_column_values(comp['columnProof']['values'], positions, aug, column_length)
        if not MerkleTree.verify_batch(comp['columnRoot'], aug, _rehash(comp['columnProof'], H), H):
            raise ValueError(f'Verification of column Merkle proof failed at depth {depth}')
        poly_values = [[from_bytes(b[16 * i:16 * i + 16]) for i in range(4)] for b in comp['polyProof']['values']]
        if not MerkleTree.verify_batch(p_root, positions, _rehash(comp['polyProof'], H), H):
            raise ValueError(f'Verification of polynomial Merkle proof failed at depth {depth}')
        special_x = f.prng(p_root)
        for i, pos in enumerate(positions):
            xe = f.exp(rou, pos)
            xs = [q * xe % f.p for q in quartic]
            if f.eval_poly_at(f.interpolate(xs, poly_values[i]), special_x) != column_values[i]:
                raise ValueError(f"Degree 4 polynomial didn't evaluate to column value at depth {depth}")
        p_root = comp['columnRoot']
        rou = f.exp(rou, 4)
        max_degree_plus1 //= 4
        column_length //= 4
    if max_degree_plus1 > len(ld['remainder']):
        raise ValueError('Remainder degree is greater than number of remainder values')
    rows = _transpose4(ld['remainder'])
    c_tree = MerkleTree([H(_row_bytes(r)) for r in rows], H)
    if c_tree.root != p_root:
        raise ValueError('Remainder values do not match Merkle root of the last column')
    _verify_remainder(cfg, ld['remainder'], max_degree_plus1, rou)


# ------------------------------------------------------------------------------------------------
# wire format — lib/Serializer.ts:35-79, lib/utils/serialization.ts:18-96
def _write_merkle_proof(out, proof, leaf_size):
    vals = proof['values']
    assert 0 < len(vals) <= 256                       # sizeof.ts:63-69
    out.append(bytes([0 if len(vals) == 256 else len(vals)]))
    out.extend(vals)
    cols = proof['nodes']
    assert len(cols) <= 256
    out.append(bytes([0 if len(cols) == 256 else len(cols)]))
    for col in cols:
        assert len(col) < 127                          # sizeof.ts:8,90-92
        typ = 1 if (len(col) > 0 and len(col[0]) == leaf_size) else 0
        out.append(bytes([(len(col) << 1) | typ]))
    for col in cols:
        out.extend(col)
    out.append(bytes([proof['depth']]))


def serialize(cfg, proof, value_count=1):
    out = [proof['evRoot']]
    _write_merkle_proof(out, proof['evProof'], value_count * ELEMENT_SIZE)
    ld = proof['ldProof']
    ld_leaf = ELEMENT_SIZE * 4
    out.append(ld['lcRoot'])
    _write_merkle_proof(out, ld['lcProof'], ld_leaf)
    out.append(bytes([len(ld['components'])]))
    for comp in ld['components']:
        out.append(comp['columnRoot'])
        _write_merkle_proof(out, comp['columnProof'], ld_leaf)
        _write_merkle_proof(out, comp['polyProof'], ld_leaf)
    rl = len(ld['remainder'])
    out.append(bytes([0 if rl == 256 else rl]))
    out.extend(to_bytes(v) for v in ld['remainder'])
    out.append(bytes([len(proof['iShapes'])]))
    for shape in proof['iShapes']:
        out.append(bytes([len(shape)]))
        out.extend(int(l).to_bytes(4, 'little') for l in shape)
    return b''.join(out)


def size_of(proof):  # lib/utils/sizeof.ts:12-53
    def mp(p):
        return 1 + sum(len(v) for v in p['values']) + 1 + len(p['nodes']) + sum(len(x) for c in p['nodes'] for x in c) + 1
    ld = proof['ldProof']
    size = DIGEST_SIZE + mp(proof['evProof']) + 1 + mp(ld['lcProof']) + DIGEST_SIZE
    for comp in ld['components']:
        size += DIGEST_SIZE + mp(comp['columnProof']) + mp(comp['polyProof'])
    size += len(ld['remainder']) * ELEMENT_SIZE + 1
    size += 1 + sum(1 + 4 * len(s) for s in proof['iShapes'])
    return size


def _read_merkle_proof(buf, off, leaf_size, node_size=DIGEST_SIZE):
    n = buf[off] or 256; off += 1
    values = [buf[off + i * leaf_size: off + (i + 1) * leaf_size] for i in range(n)]; off += n * leaf_size
    ncol = buf[off] or 256; off += 1
    heads = list(buf[off:off + ncol]); off += ncol
    nodes = []
    for h in heads:
        col = []
        for j in range(h >> 1):
            sz = (leaf_size if (h & 1) else node_size) if j == 0 else node_size
            col.append(buf[off:off + sz]); off += sz
        nodes.append(col)
    depth = buf[off]; off += 1
    return {'values': values, 'nodes': nodes, 'depth': depth}, off


def parse(cfg, buf, value_count=1):  # lib/Serializer.ts:83-144
    off = DIGEST_SIZE
    ev_root = buf[:off]
    ev_proof, off = _read_merkle_proof(buf, off, value_count * ELEMENT_SIZE)
    lc_root = buf[off:off + DIGEST_SIZE]; off += DIGEST_SIZE
    lc_proof, off = _read_merkle_proof(buf, off, 64)
    ncomp = buf[off]; off += 1
    comps = []
    for _ in range(ncomp):
        croot = buf[off:off + DIGEST_SIZE]; off += DIGEST_SIZE
        cp, off = _read_merkle_proof(buf, off, 64)
        pp, off = _read_merkle_proof(buf, off, 64)
        comps.append({'columnRoot': croot, 'columnProof': cp, 'polyProof': pp})
    rl = buf[off] or 256; off += 1
    remainder = [from_bytes(buf[off + 16 * i: off + 16 * i + 16]) for i in range(rl)]; off += 16 * rl
    nshape = buf[off]; off += 1
    shapes = []
    for _ in range(nshape):
        rank = buf[off]; off += 1
        shapes.append([int.from_bytes(buf[off + 4 * j: off + 4 * j + 4], 'little') for j in range(rank)]); off += 4 * rank
    assert off == len(buf)
    return {'evRoot': ev_root, 'evProof': ev_proof,
            'ldProof': {'lcRoot': lc_root, 'lcProof': lc_proof, 'components': comps, 'remainder': remainder},
            'iShapes': shapes}


def mimc_assertions(cfg, trace=None):  # examples/mimc/mimc128Assembly.ts:61-64
    trace = trace or run_mimc(cfg.field, cfg.steps, cfg.rc, cfg.seed)
    return [{'step': 0, 'register': 0, 'value': trace[0]},
            {'step': cfg.steps - 1, 'register': 0, 'value': trace[cfg.steps - 1]}]
