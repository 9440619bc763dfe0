"""Shared op-level checks of include/gstark.h: every function is driven through the galois/merkle-shaped
host mirror on a given backend and compared with an independent expectation (Python big-int arithmetic,
hashlib, oracle/pyref.py).  Used with the CPU oracle backend (-m "not gpu": pins the oracle) and with the
HIP backend (-m gpu: the parity tests proper, which additionally compare HIP bytes with oracle bytes)."""
import ctypes as C
import hashlib

from conftest import P, from_bytes, rand_elements, to_bytes
from genstark_amd.field import PrimeField
from genstark_amd.merkle import MerkleTree, createHash
from genstark_amd._abi import HASH_ALGS
from oracle import pyref

PF = pyref.Field()


def field_for(backend):
    return PrimeField(backend=backend)


def check_pointwise(backend, rng, n):
    f = field_for(backend)
    a, b = rand_elements(rng, n), rand_elements(rng, n)
    va, vb = f.newVectorFrom(a), f.newVectorFrom(b)
    s = rng.randrange(P)
    assert f.addVectorElements(va, vb).toValues() == [(x + y) % P for x, y in zip(a, b)]
    assert f.subVectorElements(va, vb).toValues() == [(x - y) % P for x, y in zip(a, b)]
    assert f.mulVectorElements(va, vb).toValues() == [(x * y) % P for x, y in zip(a, b)]
    assert f.addVectorElements(va, s).toValues() == [(x + s) % P for x in a]
    assert f.subVectorElements(va, s).toValues() == [(x - s) % P for x in a]
    assert f.mulVectorElements(va, s).toValues() == [(x * s) % P for x in a]
    inv = lambda x: pow(x, P - 2, P) if x else 0
    assert f.invVectorElements(va).toValues() == [inv(x) for x in a]
    assert f.divVectorElements(va, vb).toValues() == [x * inv(y) % P for x, y in zip(a, b)]
    assert f.combineVectors(va, vb) == sum(x * y for x, y in zip(a, b)) % P


def check_inverse_with_zeros(backend, rng, n):
    f = field_for(backend)
    a = rand_elements(rng, n)
    for i in range(0, n, 3):
        a[i] = 0
    got = f.invVectorElements(f.newVectorFrom(a)).toValues()
    assert got == [pow(x, P - 2, P) if x else 0 for x in a]
    zeros = f.invVectorElements(f.newVectorFrom([0] * n)).toValues()
    assert zeros == [0] * n


def check_domain_divisions(backend, rng, logn, logsteps):
    """gs_zero_poly_inverses / gs_div_by_domain_roots against their definitions on Python integers (0^-1 = 0), and against the
    member sequences they stand for (pluck / sub / div; evalPolysAtRoots / divMatrixElements)."""
    f = field_for(backend)
    n, steps = 1 << logn, 1 << logsteps
    w = f.getRootOfUnity(n)
    inv = lambda x: pow(x, P - 2, P) if x % P else 0
    xs = [pow(w, i, P) for i in range(n)]
    x_last = pow(w, (steps - 1) * (n // steps), P)
    want = [(x - x_last) * inv(pow(x, steps, P) - 1) % P for x in xs]
    got = f.zeroPolyInverses(w, n, steps, x_last)
    assert got.toValues() == want
    domain = f.getPowerSeries(w, n)
    num = f.subVectorElements(f.pluckVector(domain, steps, n), 1)
    den = f.subVectorElements(domain, x_last)
    assert f.divVectorElements(den, num).toBuffer() == got.toBuffer()
    # rows with 1..4 roots (first and last execution steps among them, as assertions have)
    e = n // steps
    roots = [[0], [0, (steps - 1) * e], [e * rng.randrange(steps) for _ in range(3)], [rng.randrange(n) for _ in range(4)]]
    rows = [rand_elements(rng, n) for _ in roots]
    numerators = f.newMatrixFrom(rows)
    got = f.divByDomainRoots(numerators, w, roots)
    want = []
    for r, ks in zip(rows, roots):
        zs = [1] * n
        for k in ks:
            zs = [z * (x - pow(w, k, P)) % P for z, x in zip(zs, xs)]
        want.append([v * inv(z) % P for v, z in zip(r, zs)])
    assert got.toValues() == want
    zpolys = []
    for ks in roots:
        poly = [1]
        for k in ks:
            poly = [(a - b * pow(w, k, P)) % P for a, b in zip([0] + poly, poly + [0])]
        zpolys.append(f.newVectorFrom(poly))
    z_values = f.evalPolysAtRoots(f.newMatrixFromVectors(zpolys), domain)
    assert f.divMatrixElements(numerators, z_values).toBuffer() == got.toBuffer()
    return got.toBuffer()


def check_deferred_readbacks(backend, rng, n):
    """gs_defer_begin / gs_defer_end: gathers and Merkle batch proofs queued in one window deliver what the immediate calls deliver,
    also when another staging user (a dot product) runs inside the window and when the window outgrows the staging buffer."""
    import ctypes as C
    f = field_for(backend)
    h = createHash('blake2s256', backend)
    vals = rand_elements(rng, n)
    v = f.newVectorFrom(vals)
    leaves = h.mergeVectorRows([v])
    tree = MerkleTree.create(leaves, h)
    be = f.backend
    picks = [sorted(rng.sample(range(n), min(n, 40))) for _ in range(12)]
    want_rows = [be.gather(v.ptr, 16, p) for p in picks]
    want_proofs = [tree.proveBatch(p) for p in picks]
    bufs = []
    be.call('gs_defer_begin')
    for p in picks:
        idx = (C.c_uint64 * len(p))(*p)
        out = C.create_string_buffer(16 * len(p))
        be.call('gs_gather', C.c_void_p(v.ptr), 16, idx, len(p), C.cast(out, C.c_void_p))
        depth = n.bit_length() - 1
        values, nodes = C.create_string_buffer(32 * len(p)), C.create_string_buffer(32 * len(p) * max(depth, 1))
        ncols, lens = C.c_uint32(), (C.c_uint32 * len(p))()
        be.call('gs_merkle_prove_batch', C.c_void_p(tree.values.ptr), C.c_void_p(tree.nodes.ptr), n, idx, len(p), C.cast(values, C.c_void_p),
                C.byref(ncols), lens, C.cast(nodes, C.c_void_p), len(p) * max(depth, 1))
        bufs.append((idx, out, values, nodes, ncols, lens))
        if len(bufs) == 5:
            assert f.combineVectors(v, v) == sum(x * x for x in vals) % P        # a staging user inside the window
    be.call('gs_defer_end')
    for (idx, out, values, nodes, ncols, lens), rows, proof, p in zip(bufs, want_rows, want_proofs, picks):
        assert [out.raw[16 * i:16 * i + 16] for i in range(len(p))] == rows
        assert [values.raw[32 * i:32 * i + 32] for i in range(len(p))] == proof['values']
        flat, o = [], 0
        for cidx in range(ncols.value):
            flat.append([nodes.raw[32 * (o + k):32 * (o + k + 1)] for k in range(lens[cidx])])
            o += lens[cidx]
        assert flat == proof['nodes']
    if be.name == 'hip-gfx950':                                                   # (the oracle reads back at once: its window is a no-op)
        import pytest
        with pytest.raises(Exception):
            be.call('gs_defer_end')                                               # not deferring any more


def check_fri_fold(backend, rng, logn, depth):
    """gs_fri_fold == transposeVector + interpolateQuarticBatch + evalQuarticBatch (the members it fuses), on layer `depth`."""
    import ctypes as C
    f = field_for(backend)
    n = 1 << logn
    step = 4 ** depth
    m = n // step
    w = f.getRootOfUnity(n)
    column = f.newVectorFrom(rand_elements(rng, m))
    x = rng.randrange(P)
    out = f.newVector(m // 4)
    f.backend.call('gs_fri_fold', f.le(w), n, step, C.c_void_p(column.ptr), m, f.le(x), C.c_void_p(out.ptr))
    domain = f.getPowerSeries(w, n)
    xs = f.transposeVector(domain, 4, step)
    ys = f.transposeVector(column, 4)
    want = f.evalQuarticBatch(f.interpolateQuarticBatch(xs, ys), x)
    assert out.toBuffer() == want.toBuffer()
    # ... and the definition on Python integers for a few rows: the cubic through the four points, evaluated at x
    vals, rows = column.toValues(), m // 4
    for r in (0, 1, rows - 1):
        pts = [(pow(w, (r + c * rows) * step, P), vals[r + c * rows]) for c in range(4)]
        acc = 0
        for j, (xj, yj) in enumerate(pts):
            num = den = 1
            for k, (xk, _) in enumerate(pts):
                if k != j:
                    num, den = num * (x - xk) % P, den * (xj - xk) % P
            acc = (acc + yj * num * pow(den, P - 2, P)) % P
        assert out.getValue(r) == acc
    # gs_fri_fold_seeded: the same step with x = prng(seed) = sha256(seed) as a big-endian integer mod p, derived where the seed
    # lies (a 32-byte digest in device memory — LowDegreeProver.ts:194 takes the root of the tree above)
    import hashlib
    seed = bytes(rng.randrange(256) for _ in range(32))
    xs_ = int.from_bytes(hashlib.sha256(seed).digest(), 'big') % P
    sv = f.newVector(2)
    f.backend.upload(sv.ptr, seed)
    got, want2 = f.newVector(m // 4), f.newVector(m // 4)
    f.backend.call('gs_fri_fold_seeded', f.le(w), n, step, C.c_void_p(column.ptr), m, C.c_void_p(sv.ptr), C.c_void_p(got.ptr))
    f.backend.call('gs_fri_fold', f.le(w), n, step, C.c_void_p(column.ptr), m, f.le(xs_), C.c_void_p(want2.ptr))
    assert got.toBuffer() == want2.toBuffer()
    return out.toBuffer() + got.toBuffer()


def check_mimc_composition(backend, rng, logn, logsteps, nroots):
    """gs_mimc_composition against its definition on Python integers."""
    import ctypes as C
    f = field_for(backend)
    n, steps = 1 << logn, 1 << logsteps
    e = n // steps
    w = f.getRootOfUnity(n)
    inv = lambda v: pow(v, P - 2, P) if v % P else 0
    xs = [pow(w, i, P) for i in range(n)]
    p_eval = rand_elements(rng, n)
    klen = 64 if n >= 64 else n
    k = rand_elements(rng, klen)
    d0, d1, b0, b1 = (rng.randrange(P) for _ in range(4))
    q_inc, b_inc = steps, 2 * steps
    roots = [0, (steps - 1) * e, 3 * e, 5 * e][:nroots]
    ipoly = [rng.randrange(P) for _ in range(nroots)]
    x_last = pow(w, (steps - 1) * e, P)
    want = []
    for i, x in enumerate(xs):
        q = (p_eval[(i + e) % n] - (pow(p_eval[i], 3, P) + k[i % klen])) % P
        d = q * (d0 + d1 * pow(x, q_inc, P)) * (x - x_last) * inv(pow(x, steps, P) - 1) % P
        iv = sum(c * pow(x, j, P) for j, c in enumerate(ipoly)) % P
        z = 1
        for r in roots:
            z = z * (x - pow(w, r, P)) % P
        want.append((d + (p_eval[i] - iv) * inv(z) * (b0 + b1 * pow(x, b_inc, P))) % P)
    l0, l1 = rng.randrange(P), rng.randrange(P)
    want_l = [(c + v * (l0 + l1 * pow(x, b_inc, P))) % P for c, v, x in zip(want, p_eval, xs)]
    vp, vk, out, out_l = f.newVectorFrom(p_eval), f.newVectorFrom(k), f.newVector(n), f.newVector(n)
    args = (C.c_void_p(vp.ptr), n, steps, f.le(w), C.c_void_p(vk.ptr), klen, b''.join(f.le(v) for v in (d0, d1, b0, b1)), q_inc, b_inc,
            b''.join(f.le(v) for v in ipoly), (C.c_uint64 * nroots)(*roots), nroots)
    f.backend.call('gs_mimc_composition', *args, None, C.c_void_p(out.ptr))
    f.backend.call('gs_mimc_composition', *args, f.le(l0) + f.le(l1), C.c_void_p(out_l.ptr))
    assert out.toValues() == want and out_l.toValues() == want_l
    return out.toBuffer() + out_l.toBuffer()


def check_power_series_and_shuffles(backend, rng, n):
    f = field_for(backend)
    base = rng.randrange(2, P)
    ps = f.getPowerSeries(base, n)
    assert ps.toValues() == PF.power_series(base, n)
    a = rand_elements(rng, n)
    va = f.newVectorFrom(a)
    if n >= 8:
        skip, times = n // 8, n
        assert f.pluckVector(va, skip, times).toValues() == [a[(i * skip) % n] for i in range(times)]
        for cols, step in [(4, 1), (4, 2), (2, 1), (8, 1)]:
            if n % (cols * step) == 0:
                rows = n // (cols * step)
                m = f.transposeVector(va, cols, step)
                assert (m.rowCount, m.colCount) == (rows, cols)
                assert m.toValues() == [[a[(r + c * rows) * step] for c in range(cols)] for r in range(rows)]
        m = f.transposeVector(va, 4)
        t = f.transposeMatrix(m)
        assert f.joinMatrixRows(t).toValues() == a
    e = rng.randrange(P)
    small = va if n <= 64 else f.newVectorFrom(a[:64])
    assert f.expVectorElements(small, e).toValues() == [pow(x, e, P) for x in a[:small.length]]


def check_combine_many(backend, rng, n, k):
    f = field_for(backend)
    vecs = [rand_elements(rng, n) for _ in range(k)]
    coeffs = rand_elements(rng, k)
    got = f.combineManyVectors([f.newVectorFrom(v) for v in vecs], coeffs).toValues()
    assert got == [sum(v[i] * c for v, c in zip(vecs, coeffs)) % P for i in range(n)]
    m = f.newMatrixFrom(vecs[:2])
    sub = f.subMatrixElementsFromVectors([f.newVectorFrom(vecs[-1]), f.newVectorFrom(vecs[-2])], m)
    assert sub.toValues() == [[(x - y) % P for x, y in zip(vecs[-1], vecs[0])], [(x - y) % P for x, y in zip(vecs[-2], vecs[1])]]


def naive_eval(coeffs, w, n, points):
    return {q: sum(c * pow(w, i * q, P) for i, c in enumerate(coeffs)) % P for q in points}


def check_ntt(backend, rng, logn, poly_len=None, rows=1, full=False):
    """evalPolysAtRoots vs direct evaluation at sampled points (or pyref's recursive FFT when `full`),
    interpolateRoots as its exact inverse."""
    f = field_for(backend)
    n = 1 << logn
    w = PF.get_root_of_unity(n)
    plen = n if poly_len is None else poly_len
    polys = [rand_elements(rng, plen) for _ in range(rows)]
    roots = f.getPowerSeries(w, n)
    ev = f.evalPolysAtRoots(f.newMatrixFrom(polys), roots)
    assert (ev.rowCount, ev.colCount) == (rows, n)
    got = ev.toValues()
    for r in range(rows):
        if full:
            assert got[r] == PF.ntt(polys[r], w, n)
        else:
            pts = sorted(set([0, 1, n - 1, n // 2] + [rng.randrange(n) for _ in range(6)]))
            if plen <= 4096:
                want = naive_eval(polys[r], w, n, pts)
                for q in pts:
                    assert got[r][q] == want[q], (logn, plen, r, q)
    # inverse: interpolate the evaluations back
    back = f.interpolateRoots(roots, ev).toValues()
    for r in range(rows):
        assert back[r][:plen] == polys[r]
        assert all(v == 0 for v in back[r][plen:])
    # Vector forms
    v = f.evalPolyAtRoots(f.newVectorFrom(polys[0]), roots)
    assert v.toValues() == got[0]
    assert f.interpolateRoots(roots, v).toValues()[:plen] == polys[0]
    return got


def check_small_polys(backend, rng):
    f = field_for(backend)
    xs, ys = rand_elements(rng, 3, edge=0), rand_elements(rng, 3)
    ip = f.interpolate(f.newVectorFrom(xs), f.newVectorFrom(ys))
    for x, y in zip(xs, ys):
        assert f.evalPolyAt(ip, x) == y
    assert ip.toValues() == PF.interpolate(xs, ys)
    a, b = rand_elements(rng, 3), rand_elements(rng, 2)
    assert f.mulPolys(f.newVectorFrom(a), f.newVectorFrom(b)).toValues() == PF.mul_polys(a, b)
    big = rand_elements(rng, 300)
    x = rng.randrange(P)
    assert f.evalPolyAt(f.newVectorFrom(big), x) == PF.eval_poly_at(big, x)
    # the rest of galois' polynomial members: large products go through the device NTT
    for la, lb in ((300, 200), (65, 64), (1000, 7)):
        a, b = rand_elements(rng, la), rand_elements(rng, lb)
        got = f.mulPolys(f.newVectorFrom(a), f.newVectorFrom(b))
        assert got.length == la + lb - 1 and got.toValues() == PF.mul_polys(a, b)
    # mulMatrixByVector: the way examples/poseidon/utils.ts applies its MDS matrix
    mat = [rand_elements(rng, 6) for _ in range(6)]
    vec = rand_elements(rng, 6)
    assert f.mulMatrixByVector(f.newMatrixFrom(mat), f.newVectorFrom(vec)).toValues() == [sum(x * y for x, y in zip(row, vec)) % P for row in mat]
    a, b = rand_elements(rng, 9), rand_elements(rng, 5)
    pad = b + [0] * 4
    assert f.addPolys(f.newVectorFrom(a), f.newVectorFrom(b)).toValues() == [(x + y) % P for x, y in zip(a, pad)]
    assert f.subPolys(f.newVectorFrom(b), f.newVectorFrom(a)).toValues() == [(y - x) % P for x, y in zip(a, pad)]
    assert f.mulPolyByConstant(f.newVectorFrom(a), P - 2).toValues() == [x * (P - 2) % P for x in a]


def check_quartic(backend, rng, logn, depth):
    """interpolateQuarticBatch (domain fast path and generic path) + evalQuarticBatch vs Lagrange."""
    f = field_for(backend)
    n = 1 << logn
    step = 4 ** depth
    w = PF.get_root_of_unity(n)
    domain = f.getPowerSeries(w, n)
    rows = n // (4 * step)
    ys = [rand_elements(rng, 4) for _ in range(rows)]
    xs_m = f.transposeVector(domain, 4, step)
    assert xs_m.quartic_domain == (w, n, step)
    ys_m = f.newMatrixFrom(ys)
    fast = f.interpolateQuarticBatch(xs_m, ys_m).toValues()
    xs_vals = xs_m.toValues()
    generic = f.interpolateQuarticBatch(f.newMatrixFrom(xs_vals), ys_m).toValues()
    assert fast == generic
    for r in sorted(set([0, rows - 1] + [rng.randrange(rows) for _ in range(8)])):
        assert xs_vals[r] == [pow(w, (r + c * rows) * step, P) for c in range(4)]
        assert fast[r] == PF.interpolate(xs_vals[r], ys[r])
    x = rng.randrange(P)
    col = f.evalQuarticBatch(f.newMatrixFrom(fast), x).toValues()
    assert col == [PF.eval_poly_at(p, x) for p in fast]


def _h(alg):
    return (lambda b: hashlib.sha256(b).digest()) if alg == 'sha256' else (lambda b: hashlib.blake2s(b, digest_size=32).digest())


def check_hashing(backend, rng, alg, n):
    f = field_for(backend)
    h = createHash(alg, backend)
    H = _h(alg)
    for size in (16, 32, 64, 96, 128, 192):
        msg = bytes(rng.randrange(256) for _ in range(size))
        assert h.digest(msg) == H(msg) == h.digestOnDevice(msg)        # gs_hash_digest (device) and the host runtime agree
    for k in (1, 2, 3, 6, 12, 64, 65, 100):      # above GS_MAX_COMBINE = 64 the library switches to a pointer table in device memory
        if k > 12 and n > 256:
            continue
        cols = [rand_elements(rng, n) for _ in range(k)]
        got = h.mergeVectorRows([f.newVectorFrom(c) for c in cols]).toBuffer()
        for i in sorted(set([0, n - 1] + [rng.randrange(n) for _ in range(8)])):
            assert got[32 * i:32 * i + 32] == H(b''.join(c[i].to_bytes(16, 'little') for c in cols)), (alg, k, i)
    vals = rand_elements(rng, n)
    m = f.transposeVector(f.newVectorFrom(vals), 4)
    raw = m.toBuffer()
    dig = h.digestValues(m, 64).toBuffer()
    assert len(dig) == 32 * (n // 4)
    assert all(dig[32 * i:32 * i + 32] == H(raw[64 * i:64 * i + 64]) for i in range(n // 4))
    dig2 = h.digestValues(raw, 64).toBuffer()
    assert dig2 == dig
    for vs in (16, 32, 48, 128):
        if len(raw) % vs == 0:
            d = h.digestValues(raw, vs).toBuffer()
            cnt = len(raw) // vs
            for i in sorted(set([0, cnt - 1, cnt // 2])):
                assert d[32 * i:32 * i + 32] == H(raw[vs * i:vs * i + vs])


def check_merkle(backend, rng, alg, logn):
    f = field_for(backend)
    h = createHash(alg, backend)
    H = _h(alg)
    n = 1 << logn
    leaves_v = h.mergeVectorRows([f.newVectorFrom(rand_elements(rng, n))])
    raw = leaves_v.toBuffer()
    leaves = [raw[32 * i:32 * i + 32] for i in range(n)]
    tree = MerkleTree.create(leaves_v, h)
    ref = pyref.MerkleTree(leaves, H)
    assert tree.root == ref.root
    nodes = tree.nodes.toBuffer()
    assert nodes[:32] == b'\0' * 32
    assert [nodes[32 * i:32 * i + 32] for i in range(1, n)] == ref.nodes[1:]
    for count in sorted(set(min(c, n) for c in (1, 2, 5, max(1, min(40, n // 2))))):
        idx = rng.sample(range(n), count)
        proof = tree.proveBatch(idx)
        want = ref.prove_batch(idx)
        assert proof['values'] == want['values'] and proof['nodes'] == want['nodes'] and proof['depth'] == want['depth']
        assert MerkleTree.verifyBatch(tree.root, idx, proof, h)
        assert pyref.MerkleTree.verify_batch(ref.root, idx, want, H)
        bad = dict(proof)
        bad['values'] = [bytes(32)] + proof['values'][1:]
        assert not MerkleTree.verifyBatch(tree.root, idx, bad, h)


def merkle_commit_bytes(backend, alg, cols_raw, n, fused=True):
    """(leaves, nodes) bytes of gs_merkle_commit_rows over the given columns (raw element bytes); fused=False: the two members
    gs_hash_merge_rows + gs_merkle_build one after the other (what the fused entry must equal)."""
    be = backend
    es = be.element_size
    ptrs = []
    for raw in cols_raw:
        assert len(raw) == es * n
        p = be.alloc(len(raw)); be.upload(p, raw); ptrs.append(p)
    leaves, nodes = be.alloc(32 * n), be.alloc(32 * n)
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    a = HASH_ALGS[alg]
    if fused == 'seed':
        # gs_merkle_commit_rows_seed: the same tree; its last launch posts the root and derives prng(root) (LowDegreeProver.ts:194)
        import hashlib
        point, ticket = be.alloc(16), C.c_uint64()
        be.call('gs_merkle_commit_rows_seed', a, arr, len(ptrs), n, C.c_void_p(leaves), C.c_void_p(nodes), C.c_void_p(point), C.byref(ticket))
        root = C.create_string_buffer(32)
        be.call('gs_readback_wait', ticket.value, root)
        assert root.raw == be.download(nodes + 32, 32)
        if es == 16:
            from genstark_amd.field import MODULUS
            assert int.from_bytes(be.download(point, 16), 'little') == int.from_bytes(hashlib.sha256(root.raw).digest(), 'big') % MODULUS
        # either half alone
        t2 = C.c_uint64()
        be.call('gs_merkle_commit_rows_seed', a, arr, len(ptrs), n, C.c_void_p(leaves), C.c_void_p(nodes), None, C.byref(t2))
        r2 = C.create_string_buffer(32)
        be.call('gs_readback_wait', t2.value, r2)
        assert r2.raw == root.raw
        p2 = be.alloc(16)
        be.call('gs_merkle_commit_rows_seed', a, arr, len(ptrs), n, C.c_void_p(leaves), C.c_void_p(nodes), C.c_void_p(p2), None)
        assert be.download(p2, 16) == be.download(point, 16)
        be.free(point); be.free(p2)
    elif fused:
        be.call('gs_merkle_commit_rows', a, arr, len(ptrs), n, C.c_void_p(leaves), C.c_void_p(nodes))
    else:
        be.call('gs_hash_merge_rows', a, arr, len(ptrs), n, C.c_void_p(leaves))
        be.call('gs_merkle_build', a, C.c_void_p(leaves), n, C.c_void_p(nodes))
    out = be.download(leaves, 32 * n), be.download(nodes, 32 * n)
    for p in ptrs + [leaves, nodes]:
        be.free(p)
    return out


def check_merkle_commit(backend, rng, alg, logn, count):
    """gs_merkle_commit_rows = mergeVectorRows + MerkleTree.create (lib/Stark.ts:115-118) against hashlib, every node."""
    H = _h(alg)
    n = 1 << logn
    es = backend.element_size
    cols = [bytes(rng.getrandbits(8) for _ in range(es * n)) if n <= 4096 else rng.randbytes(es * n) for _ in range(count)]
    leaves, nodes = merkle_commit_bytes(backend, alg, cols, n)
    want_leaves = [H(b''.join(c[es * i:es * i + es] for c in cols)) for i in range(n)]
    assert [leaves[32 * i:32 * i + 32] for i in range(n)] == want_leaves
    ref = pyref.MerkleTree(want_leaves, H)
    assert nodes[:32] == b'\0' * 32
    assert [nodes[32 * i:32 * i + 32] for i in range(1, n)] == ref.nodes[1:]
    assert merkle_commit_bytes(backend, alg, cols, n, fused=False) == (leaves, nodes)
    assert merkle_commit_bytes(backend, alg, cols, n, fused='seed') == (leaves, nodes)


def check_device_record_ops(backend, rng):
    """gs_gather_words, gs_transpose_records, gs_fri_fold_seeded_scaled (the entries the multi-GPU driver adds) against their
    definitions."""
    import hashlib
    be = backend
    f = field_for(be)
    # gs_gather_words: dst[t] = the 16-byte word at device address addrs[t]
    n = 257
    src_raw = bytes(rng.getrandbits(8) for _ in range(16 * n))
    src, dst = be.alloc(16 * n), be.alloc(16 * 40)
    be.upload(src, src_raw)
    picks = [rng.randrange(n) for _ in range(40)]
    addrs = be.alloc(8 * 40)
    be.upload(addrs, b''.join((src + 16 * i).to_bytes(8, 'little') for i in picks))
    be.call('gs_gather_words', C.c_void_p(addrs), 40, C.c_void_p(dst))
    assert be.download(dst, 16 * 40) == b''.join(src_raw[16 * i:16 * i + 16] for i in picks)
    be.call('gs_gather_words', C.c_void_p(addrs), 0, C.c_void_p(dst))                      # empty: no-op
    # gs_readback_post / gs_readback_wait: the bytes as they were when the copy's turn came, picked up ticket by ticket in any order
    tickets = []
    for i in (5, 0, 200, 31):
        t = C.c_uint64()
        be.call('gs_readback_post', C.c_void_p(src + 16 * i), 32 if i != 200 else 256, C.byref(t))
        tickets.append((i, t.value))
    be.upload(src, bytes(16 * n))                                                           # later work on the same queue: not seen
    for i, t in reversed(tickets):
        nb = 32 if i != 200 else 256
        out = C.create_string_buffer(nb)
        be.call('gs_readback_wait', t, out)
        assert out.raw == src_raw[16 * i:16 * i + nb], i
    for bad in (0, 24, 272):
        try:
            be.call('gs_readback_post', C.c_void_p(src), bad, C.byref(C.c_uint64()))
            raise AssertionError('readback sizes are multiples of 16 up to 256')
        except AssertionError:
            raise
        except Exception as e:      # noqa: BLE001
            assert 'multiple of 16' in str(e)
    try:
        be.call('gs_readback_wait', tickets[-1][1] + 1000, C.create_string_buffer(32))
        raise AssertionError('a ticket that was never issued must be refused')
    except AssertionError:
        raise
    except Exception as e:      # noqa: BLE001
        assert 'not outstanding' in str(e)
    for k in range(70):                                                                     # the ring wraps: 64 slots
        t = C.c_uint64()
        be.call('gs_readback_post', C.c_void_p(dst), 16, C.byref(t))
    out = C.create_string_buffer(16)
    be.call('gs_readback_wait', t.value, out)
    assert out.raw == be.download(dst, 16)
    try:
        be.call('gs_readback_wait', t.value - 64, out)
        raise AssertionError('a lapsed ticket must be refused')
    except AssertionError:
        raise
    except Exception as e:      # noqa: BLE001
        assert 'not outstanding' in str(e)
    be.upload(src, src_raw)
    # gs_transpose_records: dst[c * rows + r] = src[r * cols + c]
    for rows, cols, rec in ((8, 5, 32), (2, 64, 16), (1, 7, 48), (4, 1, 32)):
        raw = bytes(rng.getrandbits(8) for _ in range(rows * cols * rec))
        a, b = be.alloc(len(raw)), be.alloc(len(raw))
        be.upload(a, raw)
        be.call('gs_transpose_records', C.c_void_p(a), rows, cols, rec, C.c_void_p(b))
        want = b''.join(raw[(r * cols + c) * rec:(r * cols + c + 1) * rec] for c in range(cols) for r in range(rows))
        assert be.download(b, len(raw)) == want, (rows, cols, rec)
        be.free(a); be.free(b)
    try:
        be.call('gs_transpose_records', C.c_void_p(src), 2, 2, 24, C.c_void_p(dst))
        raise AssertionError('a record size that is not a multiple of 16 must be refused')
    except Exception as e:      # noqa: BLE001
        assert 'multiple of 16' in str(e)
    # gs_fri_fold_seeded_scaled: the fold at prng(seed) * scale
    n, step = 256, 4
    m = n // step
    w = f.getRootOfUnity(n)
    column = f.newVectorFrom(rand_elements(rng, m))
    seed = bytes(rng.randrange(256) for _ in range(32))
    scale = rng.randrange(1, P)
    x = int.from_bytes(hashlib.sha256(seed).digest(), 'big') % P * scale % P
    sv = f.newVector(2)
    be.upload(sv.ptr, seed)
    got, want2 = f.newVector(m // 4), f.newVector(m // 4)
    be.call('gs_fri_fold_seeded_scaled', f.le(w), n, step, C.c_void_p(column.ptr), m, C.c_void_p(sv.ptr), f.le(scale), C.c_void_p(got.ptr))
    be.call('gs_fri_fold', f.le(w), n, step, C.c_void_p(column.ptr), m, f.le(x), C.c_void_p(want2.ptr))
    assert got.toBuffer() == want2.toBuffer()
    # gs_fri_fold_at: the fold at a point that lies in device memory
    xv = f.newVectorFrom([x])
    got3 = f.newVector(m // 4)
    be.call('gs_fri_fold_at', f.le(w), n, step, C.c_void_p(column.ptr), m, C.c_void_p(xv.ptr), C.c_void_p(got3.ptr))
    assert got3.toBuffer() == want2.toBuffer()
    for p_ in (src, dst, addrs):
        be.free(p_)


def check_coset_divisions(backend, rng, logn, logsteps, ranks):
    """gs_zero_poly_inverses_coset / gs_div_by_domain_roots_coset — the two table-driven divisions over ONE RANK'S COSET of a domain
    (point i = shift * w^i, w = omega^ranks, shift = omega^rank): every rank's output must be the strided share of what the plain entry
    computes over the whole domain, and equal its definition on Python integers (sampled)."""
    be = backend
    f = field_for(be)
    p = f.modulus
    n, steps = 1 << logn, 1 << logsteps
    omega = f.getRootOfUnity(n)
    x_last = pow(omega, (steps - 1) * (n // steps), p)
    full_z = f.newVector(n)
    be.call('gs_zero_poly_inverses', f.le(omega), n, steps, f.le(x_last), C.c_void_p(full_z.ptr))
    full_z = full_z.toValues()
    rows, max_roots = 2, 3
    num = [[rng.randrange(p) for _ in range(n)] for _ in range(rows)]
    e = n // steps
    roots = [[0, (steps - 1) * e, (steps // 2) * e], [(steps // 4) * e, 0, 0]]
    per_row = [3, 1]
    numv = f.newMatrixFrom(num)
    full_b = f.newMatrix(rows, n)
    ri = (C.c_uint64 * (rows * max_roots))(*[k for r in roots for k in r])
    pr = (C.c_uint32 * rows)(*per_row)
    be.call('gs_div_by_domain_roots', C.c_void_p(numv.ptr), rows, n, f.le(omega), ri, pr, max_roots, C.c_void_p(full_b.ptr))
    full_b = full_b.toValues()
    m = n // ranks
    w = pow(omega, ranks, p)
    out = []
    for g in range(ranks):
        shift = pow(omega, g, p)
        z = f.newVector(m)
        be.call('gs_zero_poly_inverses_coset', f.le(w), m, f.le(shift), steps, f.le(x_last), C.c_void_p(z.ptr))
        got = z.toValues()
        assert got == full_z[g::ranks], ('1/Z', g)
        for k in sorted({0, 1, m - 1, rng.randrange(m), rng.randrange(m)}):
            x = shift * pow(w, k, p) % p
            den = (pow(x, steps, p) - 1) % p
            assert got[k] == ((x - x_last) * pow(den, p - 2, p) % p if den else 0)
        share = f.newMatrixFrom([row[g::ranks] for row in num])
        b = f.newMatrix(rows, m)
        ri2 = (C.c_uint64 * (rows * max_roots))(*[k // ranks for r in roots for k in r])         # in units of w = omega^ranks
        be.call('gs_div_by_domain_roots_coset', C.c_void_p(share.ptr), rows, m, f.le(w), f.le(shift), ri2, pr, max_roots, C.c_void_p(b.ptr))
        gotb = b.toValues()
        if g:                    # (on the domain itself a point may coincide with a root: 0^-1 = 0 by convention, same in both entries)
            for r in range(rows):
                k = rng.randrange(m)
                x = shift * pow(w, k, p) % p
                den = 1
                for a in roots[r][:per_row[r]]:
                    den = den * (x - pow(omega, a, p)) % p
                assert gotb[r][k] == num[r][g + ranks * k] * pow(den, p - 2, p) % p
        assert gotb == [row[g::ranks] for row in full_b], ('B', g)
        out.append((got, gotb))
    return out


def check_fri_layers(backend, rng, logm, depth, nlayers, alg='blake2s256'):
    """gs_fri_layers — a run of FRI layers in one call (LowDegreeProver.ts:176-221) — against what its contract says it equals, layer by
    layer on the same backend: gs_fri_fold_at at the running point, gs_merkle_commit_rows_seed over the four quarters of the folded
    column; the posted roots; the points (sha256 of the root as a big-endian integer mod p).  Returns every output byte (for
    backend-vs-backend comparison)."""
    import hashlib
    from genstark_amd._abi import FriLayer
    be = backend
    f = field_for(be)
    es = be.element_size
    m, step = 1 << logm, 4 ** depth
    n = m * step
    w = f.getRootOfUnity(n)
    column = f.newVectorFrom([rng.randrange(f.modulus) for _ in range(m)]) if m <= 4096 else f.getPowerSeries(rng.randrange(2, f.modulus), m)
    x0 = rng.randrange(f.modulus)
    xv = f.newVectorFrom([x0])
    a = HASH_ALGS[alg]
    layers = (FriLayer * nlayers)()
    keep = []
    for i in range(nlayers):
        rows = m >> (2 * i + 2)
        nxt, leaves, nodes = f.newVector(rows), be.alloc(32 * (rows // 4)), be.alloc(32 * (rows // 4))
        point = f.newVector(1) if i % 2 == 0 else None                  # every other layer asks for its point
        keep.append((nxt, leaves, nodes, point))
        layers[i].next, layers[i].leaves, layers[i].nodes = nxt.ptr, leaves, nodes
        layers[i].point_out = point.ptr if point is not None else None
    be.call('gs_fri_layers', a, f.le(w), n, step, C.c_void_p(column.ptr), m, C.c_void_p(xv.ptr), nlayers, C.cast(layers, C.c_void_p))
    out = []
    col, cur_m, cur_step, cur_x = column, m, step, xv
    for i in range(nlayers):
        nxt, leaves, nodes, point = keep[i]
        rows = cur_m // 4
        q = rows // 4
        want_next = f.newVector(rows)
        be.call('gs_fri_fold_at', f.le(w), n, cur_step, C.c_void_p(col.ptr), cur_m, C.c_void_p(cur_x.ptr), C.c_void_p(want_next.ptr))
        assert nxt.toBuffer() == want_next.toBuffer(), ('next', i)
        wl, wn, wp = be.alloc(32 * q), be.alloc(32 * q), f.newVector(1)
        quarters = (C.c_void_p * 4)(*[want_next.ptr + k * q * es for k in range(4)])
        t = C.c_uint64()
        be.call('gs_merkle_commit_rows_seed', a, quarters, 4, q, C.c_void_p(wl), C.c_void_p(wn), C.c_void_p(wp.ptr), C.byref(t))
        assert be.download(leaves, 32 * q) == be.download(wl, 32 * q), ('leaves', i)
        assert be.download(nodes, 32 * q) == be.download(wn, 32 * q), ('nodes', i)
        root = C.create_string_buffer(32)
        be.call('gs_readback_wait', layers[i].ticket, root)
        assert root.raw == be.download(wn + 32, 32), ('posted root', i)
        assert wp.toValues()[0] == int.from_bytes(hashlib.sha256(root.raw).digest(), 'big') % f.modulus
        if point is not None:
            assert point.toBuffer() == wp.toBuffer(), ('point', i)
        out += [nxt.toBuffer(), be.download(leaves, 32 * q), be.download(nodes, 32 * q), root.raw]
        be.free(wl); be.free(wn)
        col, cur_m, cur_step, cur_x = nxt, rows, cur_step * 4, wp
        keep[i] = (nxt, leaves, nodes, point, wp)
    for k in keep:
        be.free(k[1]); be.free(k[2])
    return hashlib.sha256(b''.join(out)).hexdigest()


def check_mimc_air(backend, rng, steps):
    from genstark_amd.air import MimcAir, runMimc
    f = field_for(backend)
    air = MimcAir(steps, 16, f)
    ctx = air.initProvingContext([], [3])
    trace = ctx.generateExecutionTrace()
    want = runMimc(f, steps, air.roundConstants, 3)
    assert trace.toValues() == [want]
    assert want == pyref.run_mimc(PF, steps, pyref.mimc_round_constants(PF), 3)
    p_polys = f.interpolateRoots(ctx.executionDomain, trace)
    q = ctx.evaluateTransitionConstraints(p_polys).toValues()[0]
    nc = steps * 4
    # Q vanishes on the execution domain except at the last step
    for s in range(steps - 1):
        assert q[s * 4] == 0
    assert q[(steps - 1) * 4] != 0
    # and matches the verifier-side scalar evaluation at arbitrary points
    vctx = air.initVerificationContext()
    wc = ctx.compositionDomain.series_base
    pv = PF.ntt(p_polys.toValues()[0], wc, nc)
    for j in sorted(set([1, 2, 3, nc - 1] + [rng.randrange(nc) for _ in range(6)])):
        x = pow(wc, j, P)
        assert q[j] == vctx.evaluateConstraintsAt(x, [pv[j]], [pv[(j + 4) % nc]], [])[0]


class VecScalarField:
    """Scalar field ops routed through 1-element device vectors: lets a known-answer computation written
    against add/mul/exp exercise the backend's arithmetic kernels instead of Python integers."""

    def __init__(self, backend):
        self.f = field_for(backend)

    def _v(self, x):
        return self.f.newVectorFrom([x % P])

    def add(self, a, b):
        return self.f.addVectorElements(self._v(a), self._v(b)).toValues()[0]

    def mul(self, a, b):
        return self.f.mulVectorElements(self._v(a), self._v(b)).toValues()[0]

    def exp(self, a, e):
        return self.f.expVectorElements(self._v(a), e).toValues()[0]


def check_rescue_kat(backend):
    import rescue_kat
    assert rescue_kat.run(VecScalarField(backend)) == (302524937772545017647250309501879538110,
                                                       205025454306577433144586673939030012640)


def check_many_vectors(backend, rng, n, count):
    """combineManyVectors / subMatrixElementsFromVectors with more vectors than one kernel launch carries (GS_MAX_COMBINE = 64):
    an AIR with more than 32 registers (LinearCombination.ts:36-64 combines 2 * registers vectors) must prove like any other."""
    f = field_for(backend)
    cols = [rand_elements(rng, n) for _ in range(count)]
    ks = [rng.randrange(P) for _ in range(count)]
    vecs = [f.newVectorFrom(c) for c in cols]
    got = f.combineManyVectors(vecs, ks).toValues()
    assert got == [sum(c[i] * k for c, k in zip(cols, ks)) % P for i in range(n)]
    m = f.newMatrixFrom([rand_elements(rng, n) for _ in range(count)])
    diff = f.subMatrixElementsFromVectors(vecs, m)
    mv = m.toValues()
    assert diff.toValues() == [[(cols[r][i] - mv[r][i]) % P for i in range(n)] for r in range(count)]


def check_combine_adjusted(backend, rng, n=300):
    """gs_combine_adjusted against its definition on Python integers and against the member sequence it replaces
    (mulVectorElements per adjusted vector, combineManyVectors over all terms, addVectorElements): every combination of absent
    parts, `plus` aliasing `out`, more vectors than one launch carries."""
    be = backend
    f = field_for(be)
    for count in (1, 3, 6, 70):
        cols = [rand_elements(rng, n) for _ in range(count)]
        vecs = [f.newVectorFrom(c) for c in cols]
        pw, pl = rand_elements(rng, n), rand_elements(rng, n)
        pv, plv = f.newVectorFrom(pw), f.newVectorFrom(pl)
        k, kp = [rng.randrange(P) for _ in range(count)], [rng.randrange(P) for _ in range(count)]
        ptrs = (C.c_void_p * count)(*[v.ptr for v in vecs])
        kb, kpb = b''.join(x.to_bytes(16, 'little') for x in k), b''.join(x.to_bytes(16, 'little') for x in kp)
        for has_k, has_kp, has_plus in ((1, 1, 1), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 1, 1), (0, 1, 0)):
            out = f.newVector(n)
            be.call('gs_combine_adjusted', ptrs, kb if has_k else None, kpb if has_kp else None, count, C.c_void_p(pv.ptr) if has_kp else None,
                    C.c_void_p(plv.ptr) if has_plus else None, n, C.c_void_p(out.ptr))
            want = [((sum(c[i] * a for c, a in zip(cols, k)) if has_k else 0) + (pw[i] * sum(c[i] * a for c, a in zip(cols, kp)) if has_kp else 0)
                     + (pl[i] if has_plus else 0)) % P for i in range(n)]
            assert out.toValues() == want, (count, has_k, has_kp, has_plus)
        # the reference's member sequence (CompositionPolynomial.ts:83-107 / LinearCombination.ts:44-63)
        adjusted = [f.mulVectorElements(v, pv) for v in vecs]
        merged = f.addVectorElements(f.combineManyVectors(vecs + adjusted, k + kp), plv)
        acc = f.newVectorFrom(pl)                                                       # plus = out (in place)
        be.call('gs_combine_adjusted', ptrs, kb, kpb, count, C.c_void_p(pv.ptr), C.c_void_p(acc.ptr), n, C.c_void_p(acc.ptr))
        assert acc.toValues() == merged.toValues(), count
    for bad in ((None, None), ):
        try:
            be.call('gs_combine_adjusted', ptrs, None, None, count, None, None, n, C.c_void_p(acc.ptr))
            raise AssertionError('no coefficient list at all must be refused')
        except AssertionError:
            raise
        except Exception:      # noqa: BLE001
            pass


def check_constraints_strided(backend, air, inputs):
    """gs_air_constraints_strided (include/gstark.h): the constraint program over the composition domain with the registers read in
    place from the extension over the EVALUATION domain (every (N / Nc)-th point) equals gs_air_constraints on the plucked copy —
    compiled or interpreted, whichever the backend runs; strides that do not fit the rows are refused.  Returns the values."""
    import ctypes as C
    from genstark_amd.field import Matrix
    f = air.field
    ctx = air.initProvingContext([], inputs)
    p_polys = f.interpolateRoots(ctx.executionDomain, ctx.generateExecutionTrace())
    want = ctx.evaluateTransitionConstraints(p_polys).toValues()
    p_eval = f.evalPolysAtRoots(p_polys, ctx.evaluationDomain)
    n, nc = ctx.evaluationDomain.length, ctx.compositionDomain.length
    code, ninstr, consts, nconsts, nregs = air.evaluationProgram.abi_args(f.elementSize)
    q = Matrix(f.backend, len(air.constraintDegrees), nc)
    lens = (C.c_uint64 * max(len(ctx._staticLens), 1))(*ctx._staticLens)
    args = [code, ninstr, consts, nconsts, nregs, air.traceRegisterCount, len(air.constraintDegrees), C.c_void_p(p_eval.ptr)]
    tail = [nc, nc // ctx.traceLength, C.c_void_p(ctx._staticTables.ptr), lens, len(ctx._staticLens), C.c_void_p(q.ptr)]
    f.backend.call('gs_air_constraints_strided', *args, n, n // nc, *tail)
    got = q.toValues()
    assert got == want
    for prow, pstride in ((n, 0), (n, n // nc + 1), (nc, n // nc)):
        assert f.backend.lib.gs_air_constraints_strided(f.backend.ctx, *args, prow, pstride, *tail) != 0
    return got


def check_composition_tail(backend, rng, logn, logsteps, per_row, lcount, adjusted, with_c=True, made=False):
    """gs_composition_tail against its definition on Python integers (sampled points) and against the sequence of entries it replaces
    (gs_vec_mul, gs_eval_polys_at_roots, gs_sub_matrix_from_vectors, gs_div_by_domain_roots, gs_combine_adjusted twice) on the same
    backend (every point).  per_row: assertions per asserted register (1..4 each).  made: 1/Z(x) and the power series are not passed as
    vectors but as what defines them (steps + the last step's point; the exponent) — the entry computes them per point.  Returns (c, l)
    for cross-backend comparison."""
    be = backend
    f = field_for(be)
    p = f.modulus
    n, steps = 1 << logn, 1 << logsteps
    e = n // steps
    omega = f.getRootOfUnity(n)
    bcount, ilen = len(per_row), max(per_row)
    qv, zv, pwv = rand_elements(rng, n), rand_elements(rng, n), rand_elements(rng, n)
    x_last = pow(omega, (steps - 1) * e, p)
    pw_exp = rng.randrange(1, n)
    if made:                                     # what gs_zero_poly_inverses / gs_power_series would have produced
        zvec = f.newVector(n)
        be.call('gs_zero_poly_inverses', f.le(omega), n, steps, f.le(x_last), C.c_void_p(zvec.ptr))
        zv = zvec.toValues()
        pwv = [pow(omega, i * pw_exp % n, p) for i in range(n)]
    bcols = [rand_elements(rng, n) for _ in range(bcount)]
    lcols = [rand_elements(rng, n) for _ in range(lcount)]
    roots = [sorted(rng.sample(range(steps), m)) for m in per_row]              # asserted steps of register b
    ipolys = [[rng.randrange(p) for _ in range(m)] + [0] * (ilen - m) for m in per_row]
    bk, bkp = [rng.randrange(p) for _ in range(bcount)], [rng.randrange(p) for _ in range(bcount)]
    lk, lkp = [rng.randrange(p) for _ in range(lcount)], [rng.randrange(p) for _ in range(lcount)]
    q, z, pw = f.newVectorFrom(qv), f.newVectorFrom(zv), f.newVectorFrom(pwv)
    bvecs, lvecs = [f.newVectorFrom(c) for c in bcols], [f.newVectorFrom(c) for c in lcols]
    bp = (C.c_void_p * bcount)(*[v.ptr for v in bvecs])
    lp = (C.c_void_p * max(lcount, 1))(*[v.ptr for v in lvecs])
    ri = (C.c_uint64 * (bcount * ilen))(*[(r[k] * e if k < len(r) else 0) for r in roots for k in range(ilen)])
    pr = (C.c_uint32 * bcount)(*per_row)
    le = lambda xs: b''.join(f.le(v) for v in xs)
    c_out, l_out = f.newVector(n), f.newVector(n)
    be.call('gs_composition_tail', n, f.le(omega), C.c_void_p(q.ptr), None if made else C.c_void_p(z.ptr), steps, f.le(x_last), bp, bcount,
            le([v for row in ipolys for v in row]), ilen, ri, pr, ilen,
            le(bk), le(bkp) if adjusted else None, lp if lcount else None, lcount, le(lk) if lcount else None, le(lkp) if adjusted and lcount else None,
            C.c_void_p(pw.ptr) if adjusted and not made else None, pw_exp if made else 0, C.c_void_p(c_out.ptr) if with_c else None, C.c_void_p(l_out.ptr))
    got_c, got_l = (c_out.toValues() if with_c else None), l_out.toValues()
    # the definition, on integers
    for i in sorted({0, 1, n - 1, roots[0][0] * e + 1} | {rng.randrange(n) for _ in range(12)}):
        x = pow(omega, i, p)
        c = qv[i] * zv[i] % p
        for b in range(bcount):
            iv = sum(co * pow(x, t, p) for t, co in enumerate(ipolys[b])) % p
            den = 1
            for r in roots[b]:
                den = den * (x - pow(omega, r * e, p)) % p
            if den == 0:
                continue                                                  # a root of the divisor: 0^-1 = 0 by convention (checked against the member sequence below)
            cf = (bk[b] + (pwv[i] * bkp[b] if adjusted else 0)) % p
            c = (c + cf * (bcols[b][i] - iv) * pow(den, p - 2, p)) % p
        else:
            if with_c:
                assert got_c[i] == c, i
            l = (c + sum((lk[v] + (pwv[i] * lkp[v] if adjusted else 0)) * lcols[v][i] for v in range(lcount))) % p
            assert got_l[i] == l, i
    # the sequence it replaces
    d = f.newVector(n)
    be.call('gs_vec_mul', C.c_void_p(q.ptr), C.c_void_p(z.ptr), n, C.c_void_p(d.ptr))
    ipm = f.newMatrixFrom(ipolys)
    iv, pi, bq = f.newMatrix(bcount, n), f.newMatrix(bcount, n), f.newMatrix(bcount, n)
    be.call('gs_eval_polys_at_roots', C.c_void_p(ipm.ptr), bcount, ilen, f.le(omega), n, C.c_void_p(iv.ptr))
    be.call('gs_sub_matrix_from_vectors', bp, C.c_void_p(iv.ptr), bcount, n, C.c_void_p(pi.ptr))
    be.call('gs_div_by_domain_roots', C.c_void_p(pi.ptr), bcount, n, f.le(omega), ri, pr, ilen, C.c_void_p(bq.ptr))
    rows = (C.c_void_p * bcount)(*[bq.ptr + r * n * f.elementSize for r in range(bcount)])
    cc = f.newVector(n)
    be.call('gs_combine_adjusted', rows, le(bk), le(bkp) if adjusted else None, bcount, C.c_void_p(pw.ptr) if adjusted else None, C.c_void_p(d.ptr), n, C.c_void_p(cc.ptr))
    want_c = cc.toValues()
    if lcount:
        ll = f.newVector(n)
        be.call('gs_combine_adjusted', lp, le(lk), le(lkp) if adjusted else None, lcount, C.c_void_p(pw.ptr) if adjusted else None, C.c_void_p(cc.ptr), n, C.c_void_p(ll.ptr))
        want_l = ll.toValues()
    else:
        want_l = want_c
    assert got_l == want_l and (not with_c or got_c == want_c)
    return got_c, got_l


def check_composition_tail_limits(backend):
    """More than 4 assertions on a register is refused (GS_ERR_UNSUPPORTED: the caller keeps the separate entries), and so are missing
    arguments."""
    f = field_for(backend)
    n = 64
    v = f.newVectorFrom([1] * n)
    bp = (C.c_void_p * 1)(v.ptr)
    ri = (C.c_uint64 * 5)(0, 1, 2, 3, 4)
    pr = (C.c_uint32 * 1)(5)
    one = f.le(1)
    args = lambda ilen, roots: (backend.ctx, n, f.le(f.getRootOfUnity(n)), C.c_void_p(v.ptr), C.c_void_p(v.ptr), 0, None, bp, 1, one * 5, ilen, ri, pr, roots, one, None,
                                None, 0, None, None, None, 0, None, C.c_void_p(v.ptr))
    assert backend.lib.gs_composition_tail(*args(5, 5)) == -3 or backend.lib.gs_composition_tail(*args(5, 5)) != 0
    assert backend.lib.gs_composition_tail(*args(4, 5)) != 0
    a = list(args(4, 4)); a[-1] = None
    assert backend.lib.gs_composition_tail(*a) != 0
    a = list(args(4, 4)); a[4] = None                       # no 1/Z vector and nothing to compute it from
    assert backend.lib.gs_composition_tail(*a) != 0
    a = list(args(4, 4)); a[4] = None; a[5] = 1; a[6] = one   # n / steps = 64 > 32: the vector form is the one to use
    assert backend.lib.gs_composition_tail(*a) != 0


def check_composition_tail_coset(backend, rng, logn, logsteps, per_row, lcount, ranks):
    """gs_composition_tail_coset: every rank's output over its coset (point i = omega^rank * (omega^ranks)^i, strided shares of the inputs)
    must be the strided share of what gs_composition_tail computes over the whole domain — with 1/Z and the powers passed as the
    shares of the domain-wide vectors, and computed in place from (steps, x_last) / the exponent."""
    be = backend
    f = field_for(be)
    p = f.modulus
    n, steps = 1 << logn, 1 << logsteps
    e = n // steps
    assert e % ranks == 0 and e // ranks <= 32
    omega = f.getRootOfUnity(n)
    bcount, ilen = len(per_row), max(per_row)
    x_last = pow(omega, (steps - 1) * e, p)
    pw_exp = rng.randrange(1, 4 * n)                 # (the degree adjustment's exponent may exceed a rank's domain)
    qv = rand_elements(rng, n)
    zvec = f.newVector(n)
    be.call('gs_zero_poly_inverses', f.le(omega), n, steps, f.le(x_last), C.c_void_p(zvec.ptr))
    zv = zvec.toValues()
    pwv = [pow(omega, i * pw_exp % n, p) for i in range(n)]
    bcols = [rand_elements(rng, n) for _ in range(bcount)]
    lcols = [rand_elements(rng, n) for _ in range(lcount)]
    roots = [sorted(rng.sample(range(steps), m)) for m in per_row]
    ipolys = [[rng.randrange(p) for _ in range(m)] + [0] * (ilen - m) for m in per_row]
    bk, bkp = [rng.randrange(p) for _ in range(bcount)], [rng.randrange(p) for _ in range(bcount)]
    lk, lkp = [rng.randrange(p) for _ in range(lcount)], [rng.randrange(p) for _ in range(lcount)]
    le = lambda xs: b''.join(f.le(v) for v in xs)
    pr = (C.c_uint32 * bcount)(*per_row)

    def run(name, m, w, shift, q_, z_, pw_, b_, l_, unit, made):
        q, z, pw = f.newVectorFrom(q_), f.newVectorFrom(z_), f.newVectorFrom(pw_)
        bvecs, lvecs = [f.newVectorFrom(c) for c in b_], [f.newVectorFrom(c) for c in l_]
        bp = (C.c_void_p * bcount)(*[v.ptr for v in bvecs])
        lp = (C.c_void_p * max(lcount, 1))(*[v.ptr for v in lvecs])
        ri = (C.c_uint64 * (bcount * ilen))(*[(r[k] * e // unit if k < len(r) else 0) for r in roots for k in range(ilen)])
        c_out, l_out = f.newVector(m), f.newVector(m)
        head = [m, f.le(w)] + ([f.le(shift)] if name.endswith('coset') else [])
        be.call(name, *head, C.c_void_p(q.ptr), None if made else C.c_void_p(z.ptr), steps, f.le(x_last), bp, bcount,
                le([v for row in ipolys for v in row]), ilen, ri, pr, ilen, le(bk), le(bkp), lp if lcount else None, lcount,
                le(lk) if lcount else None, le(lkp) if lcount else None, None if made else C.c_void_p(pw.ptr), pw_exp if made else 0,
                C.c_void_p(c_out.ptr), C.c_void_p(l_out.ptr))
        return c_out.toValues(), l_out.toValues()

    full = run('gs_composition_tail', n, omega, 1, qv, zv, pwv, bcols, lcols, 1, False)
    assert full == run('gs_composition_tail', n, omega, 1, qv, zv, pwv, bcols, lcols, 1, True)
    w = pow(omega, ranks, p)
    out = []
    for g in range(ranks):
        share = lambda col: col[g::ranks]
        args = (n // ranks, w, pow(omega, g, p), share(qv), share(zv), share(pwv), [share(c) for c in bcols], [share(c) for c in lcols], ranks)
        for made in (False, True):
            got = run('gs_composition_tail_coset', *args, made)
            assert got == (share(full[0]), share(full[1])), (g, made)
        out.append(got)
    return out
