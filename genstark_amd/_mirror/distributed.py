"""ONE proof across the GPUs of a node: a distributed-array facade under the unchanged prover.

Every rank runs the same `Stark.prove()` (SPMD, one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  `DistField` is a PrimeField whose evaluation-domain-sized vectors are *distributed*:
a vector of global length n lives strided over the G ranks — rank g holds global[g + G*k], k < n/G — and every galois
member keeps its GLOBAL meaning, so lib/Stark.ts' call sequence and the lib/components mirrors run as they are and the
proof bytes equal the single-device ones (tests/test_distributed_prove.py).  Short vectors (trace, trace polynomials,
composition domain: at most N/4 elements) stay replicated: every rank computes them redundantly, which costs no
communication and no synchronisation.  What the layout buys:

  * low-degree extension needs NO communication: rank g evaluates a polynomial on its coset {w^(g+G*k)} by scaling
    coefficient j with w^(g*j) and running one NTT of size n/G with root w^G;
  * every pointwise member (add/sub/mul/div/combineMany, batch inversion) and power series are local;
  * FRI is local: row r = {v[r], v[r+n/4], v[r+n/2], v[r+3n/4]} of transposeVector(v, 4) has all four elements on rank
    r mod G (G divides n/4), so cubic interpolation, evaluation at the special point and row hashing need no exchange, and
    the next column is again a strided distributed vector;
  * Merkle trees are the one place that wants natural order: leaf digests (32 B each) are re-sharded strided -> blocked by
    ONE point-to-point exchange per tree ((n/G^2)*32 B per pair, every pair on its own xGMI link), each rank builds the
    subtree over its n/G consecutive leaves, the G sub-roots are all-gathered and the top log2(G) levels are computed on
    every rank, so the Fiat-Shamir seeds agree everywhere without a broadcast;
  * queries (<= 256 positions) are answered by the owners: a deterministic host plan (same on every rank) says which
    digests / rows a batch proof needs, every rank gathers the ones it owns with one device gather, one all-gather of a few
    KB assembles the proof on every rank.
"""
import ctypes as C

import torch
import torch.distributed as dist

from .._abi import GstarkError
from ..field import ELEMENT_SIZE, Matrix, PrimeField, Vector
from ..merkle import DIGEST_SIZE, Hash, MerkleTree


class _Comm:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world & (self.world - 1):
            raise GstarkError('world size must be a power of two')

    def _xdev(self, backend):
        """where exchange buffers live: the rank's GPU under RCCL; host memory under gloo (CPU tests, or several ranks sharing the
        one GPU of a test box)"""
        if backend.name != 'hip-gfx950' or (self.world > 1 and dist.get_backend(self.group) == 'gloo'):
            return 'cpu'
        return torch.device('cuda', backend.device)

    def all_gather_bytes(self, backend, payload, size):
        """Every rank contributes exactly `size` bytes (shorter payloads are zero-padded): ONE all_gather_into_tensor of a
        (world, size) uint8 tensor — RCCL over xGMI on the GPU box, no pickling, no per-call staging through Python objects.
        Returns the `world` contributions as bytes objects."""
        if len(payload) > size:
            raise GstarkError('all_gather_bytes: payload longer than the agreed size')
        if self.world == 1 or size == 0:
            return [bytes(payload) + bytes(size - len(payload))] * (1 if self.world == 1 else self.world)
        dev = self._xdev(backend)
        mine = torch.frombuffer(bytearray(bytes(payload) + bytes(size - len(payload))), dtype=torch.uint8)
        if dev != 'cpu':
            mine = mine.to(dev)
        out = torch.empty(self.world * size, dtype=torch.uint8, device=dev)      # flat: rank r's contribution at [r*size, (r+1)*size)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        raw = out.cpu().numpy().tobytes()
        return [raw[r * size:(r + 1) * size] for r in range(self.world)]

    def gather_owned(self, backend, owners, mine_bytes, item_bytes):
        """`owners[i]` is the rank that holds item i (a deterministic plan, identical on every rank); `mine_bytes` are this rank's
        items in plan order.  One fixed-size all-gather (every rank pads to the largest share) returns all items in plan order."""
        counts = [0] * self.world
        for o in owners:
            counts[o] += 1
        if len(mine_bytes) != counts[self.rank]:
            raise GstarkError('gather_owned: this rank did not produce its share of the plan')
        parts = self.all_gather_bytes(backend, b''.join(mine_bytes), max(counts) * item_bytes if counts else 0)
        taken = [0] * self.world
        out = []
        for o in owners:
            k = taken[o]
            out.append(parts[o][k * item_bytes:(k + 1) * item_bytes])
            taken[o] = k + 1
        return out


class _TensorOwner:
    """Keeps a torch tensor alive while Vectors view its memory (exchange buffers are torch-owned)."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()


class DistVector:
    """Global vector of `length` elements stored strided: local[k] = global[rank + world*k]."""
    dist = True

    def __init__(self, field, length, local, series_base=None):
        self.field, self.length, self.local, self.elementSize = field, int(length), local, local.elementSize
        self.series_base = series_base
        if local.length * field.comm.world != self.length:
            raise GstarkError('distributed vector: local part has the wrong length')

    @property
    def byteLength(self):
        return self.length * self.elementSize

    def valuesAt(self, indexes):
        """Raw bytes of the elements at global `indexes` (collective; every rank gets all of them)."""
        comm = self.field.comm
        indexes = list(indexes)
        mine = [i for i in indexes if i % comm.world == comm.rank]
        got = self.local.backend.gather(self.local.ptr, self.elementSize, [i // comm.world for i in mine])
        return comm.gather_owned(self.local.backend, [i % comm.world for i in indexes], got, self.elementSize)

    def getValue(self, index):
        return int.from_bytes(self.valuesAt([index])[0], 'little')

    def toBuffer(self):
        """The whole vector in natural order on the host (collective; remainder-sized vectors and tests only)."""
        comm = self.field.comm
        mine = self.local.toBuffer()
        parts = comm.all_gather_bytes(self.local.backend, mine, len(mine))
        es, m = self.elementSize, self.local.length
        out = bytearray(self.length * es)
        for g, raw in enumerate(parts):
            for k in range(m):
                out[(g + comm.world * k) * es:(g + comm.world * k + 1) * es] = raw[k * es:(k + 1) * es]
        return bytes(out)

    def toValues(self):
        raw, es = self.toBuffer(), self.elementSize
        return [int.from_bytes(raw[i * es:(i + 1) * es], 'little') for i in range(self.length)]


class DistMatrix:
    """rowCount distributed rows of global length colCount (P(x) evaluations): local Matrix rowCount x colCount/world."""
    dist = True

    def __init__(self, field, rows, cols, local):
        self.field, self.rowCount, self.colCount, self.local = field, rows, int(cols), local

    def row(self, r):
        return DistVector(self.field, self.colCount, self.local.row(r))


class DistRowMatrix:
    """transposeVector(v, 4) of a distributed vector: global row r lives on rank r mod world as local row r // world."""
    dist = True

    def __init__(self, field, rows, cols, local):
        self.field, self.rowCount, self.colCount, self.local = field, int(rows), cols, local
        self.quartic_domain = self.quartic_coset = self.x_scale = None
        if local.rowCount * field.comm.world != self.rowCount:
            raise GstarkError('distributed matrix: local part has the wrong number of rows')

    def rowsToBuffers(self, indexes):
        comm = self.field.comm
        indexes = list(indexes)
        mine = [i for i in indexes if i % comm.world == comm.rank]
        got = self.local.rowsToBuffers([i // comm.world for i in mine])
        return comm.gather_owned(self.field.backend, [i % comm.world for i in indexes], got, self.colCount * ELEMENT_SIZE)

    def toReplicated(self):
        """All rows on every rank (collective; the <= 64-row FRI remainder)."""
        comm, f = self.field.comm, self.field
        mine = self.local.toBuffer()
        parts = comm.all_gather_bytes(f.backend, mine, len(mine))
        rb = self.colCount * ELEMENT_SIZE
        out = bytearray(self.rowCount * rb)
        for g, raw in enumerate(parts):
            for k in range(self.local.rowCount):
                r = g + comm.world * k
                out[r * rb:(r + 1) * rb] = raw[k * rb:(k + 1) * rb]
        m = Matrix(f.backend, self.rowCount, self.colCount)
        f.backend.upload(m.ptr, bytes(out))
        return m


def _is_dist(x):
    return getattr(x, 'dist', False) is True


class DistField(PrimeField):
    """PrimeField whose vectors of `dist_length` elements (the evaluation domain size N) are distributed."""
    fusedDomainDivisions = False         # the general member sequences are index-local here; the fused kernels enumerate a whole domain

    def __init__(self, backend, dist_length, group=None):
        super().__init__(backend=backend)
        self.comm = _Comm(group)
        self.dist_length = int(dist_length)
        g = self.comm.world
        if self.dist_length % (4 * g * g):
            raise GstarkError('the evaluation domain must be a multiple of 4 * world^2')

    def createHash(self, algorithm):
        return DistHash(algorithm, self.backend, self)

    # ---- constructors of distributed vectors
    def _series(self, base, n):
        """{base^i, i < n} distributed: local[k] = base^rank * (base^world)^k."""
        comm = self.comm
        local = PrimeField.getPowerSeries(self, self.exp(base, comm.world), n // comm.world)
        if comm.rank:
            local = PrimeField.mulVectorElements(self, local, self.exp(base, comm.rank))
        local.series_base = None
        return DistVector(self, n, local, series_base=base % self.modulus)

    def getPowerSeries(self, base, length):
        if length == self.dist_length and self.comm.world > 1:
            return self._series(base, length)
        return super().getPowerSeries(base, length)

    def _coset_eval(self, coeff_ptr, rows, poly_len, stride, roots):
        """rows polynomials (poly_len coefficients, `stride` elements apart) on this rank's coset of the distributed
        power-series domain `roots` -> local Matrix rows x n/world."""
        comm, n = self.comm, roots.length
        omega = roots.series_base
        if omega is None:
            raise GstarkError('distributed evaluation needs a power-series domain')
        m = n // comm.world
        shift = self.exp(omega, comm.rank)
        scaled = Matrix(self.backend, rows, min(poly_len, m))
        ln = min(poly_len, m)
        scale = PrimeField.getPowerSeries(self, shift, poly_len)               # w^(g*j)
        for r in range(rows):
            src = Vector(self.backend, poly_len, owner=_Raw(coeff_ptr + r * stride * ELEMENT_SIZE))
            t = PrimeField.mulVectorElements(self, src, scale)
            if poly_len > m:
                # longer than the coset domain: after the scaling the evaluation points are the m-th roots of unity
                # (w^G)^k, so the polynomial reduces modulo y^m - 1: the chunks of m coefficients simply add up
                if poly_len % m:
                    raise GstarkError('polynomial length must be a multiple of the coset size when it exceeds it')
                chunks = [Vector(self.backend, m, owner=t._owner, offset=t._offset + c * m * ELEMENT_SIZE) for c in range(poly_len // m)]
                t = PrimeField.combineManyVectors(self, chunks, [1] * len(chunks))
            self.backend.call('gs_copy', C.c_void_p(scaled.ptr + r * ln * ELEMENT_SIZE), C.c_void_p(t.ptr), ln * ELEMENT_SIZE)
        out = Matrix(self.backend, rows, m)
        self.backend.call('gs_eval_polys_at_roots', C.c_void_p(scaled.ptr), rows, ln, self.exp(omega, comm.world).to_bytes(16, 'little'),
                          m, C.c_void_p(out.ptr))
        return out

    def evalPolyAtRoots(self, poly, roots):
        if not _is_dist(roots):
            return super().evalPolyAtRoots(poly, roots)
        local = self._coset_eval(poly.ptr, 1, poly.length, poly.length, roots)
        return DistVector(self, roots.length, local.row(0))

    def evalPolysAtRoots(self, polys, roots):
        if not _is_dist(roots):
            return super().evalPolysAtRoots(polys, roots)
        local = self._coset_eval(polys.ptr, polys.rowCount, polys.colCount, polys.colCount, roots)
        return DistMatrix(self, polys.rowCount, roots.length, local)

    # ---- pointwise members: local on the strided parts
    def _binary(self, fn_vec, fn_scalar, a, b):
        if not _is_dist(a):
            if _is_dist(b):
                raise GstarkError('cannot combine a replicated vector with a distributed one')
            return super()._binary(fn_vec, fn_scalar, a, b)
        if isinstance(b, int):
            return DistVector(self, a.length, super()._binary(fn_vec, fn_scalar, a.local, b))
        if not _is_dist(b) or b.length != a.length:
            raise GstarkError('Cannot combine vector elements: vectors have different lengths')
        return DistVector(self, a.length, super()._binary(fn_vec, fn_scalar, a.local, b.local))

    def divVectorElements(self, a, b):
        if not _is_dist(a):
            return super().divVectorElements(a, b)
        if isinstance(b, int):
            return self.mulVectorElements(a, self.inv(b))
        return DistVector(self, a.length, super().divVectorElements(a.local, b.local))

    def invVectorElements(self, a):
        return DistVector(self, a.length, super().invVectorElements(a.local)) if _is_dist(a) else super().invVectorElements(a)

    def expVectorElements(self, a, e):
        return DistVector(self, a.length, super().expVectorElements(a.local, e)) if _is_dist(a) else super().expVectorElements(a, e)

    def combineManyVectors(self, vectors, coefficients):
        if not any(_is_dist(v) for v in vectors):
            return super().combineManyVectors(vectors, coefficients)
        return DistVector(self, vectors[0].length, super().combineManyVectors([v.local for v in vectors], coefficients))

    def pluckVector(self, v, skip, times):
        if not _is_dist(v):
            return super().pluckVector(v, skip, times)
        if v.series_base is None or times % v.length:
            raise GstarkError('pluckVector on a distributed vector needs a power-series source')
        # out[i] = base^((i*skip) mod n) = (base^skip)^i
        return self._series(self.exp(v.series_base, skip), times)

    def matrixRowsToVectors(self, matrix):
        return [matrix.row(r) for r in range(matrix.rowCount)]

    def subMatrixElementsFromVectors(self, vectors, m):
        if not _is_dist(m):
            return super().subMatrixElementsFromVectors(vectors, m)
        return DistMatrix(self, m.rowCount, m.colCount, super().subMatrixElementsFromVectors([v.local for v in vectors], m.local))

    def divMatrixElements(self, a, b):
        if not _is_dist(a):
            return super().divMatrixElements(a, b)
        return DistMatrix(self, a.rowCount, a.colCount, super().divMatrixElements(a.local, b.local))

    # ---- FRI rows
    def transposeVector(self, v, columns, step=1):
        if not _is_dist(v):
            return super().transposeVector(v, columns, step)
        if step != 1:
            # the sub-domain {base^(step*i)}: rebuilt as a distributed series (its rows are not co-located in v's layout)
            if v.series_base is None:
                raise GstarkError('strided transposeVector on a distributed vector needs a power-series source')
            v = self._series(self.exp(v.series_base, step), v.length // step)
        rows = v.length // columns
        if columns != 4 or rows % self.comm.world:
            raise GstarkError('distributed transposeVector: 4 columns, world must divide the row count')
        local = super().transposeVector(v.local, columns)
        local.quartic_domain = None
        out = DistRowMatrix(self, rows, columns, local)
        if v.series_base is not None:
            # my rows are x = s*u with s = base^rank and u running over the transposed power series of base^world: the
            # cubic through (s*u_c, y_c) is Q(x/s) where Q interpolates (u_c, y_c) -- the 9-multiplication domain kernel
            comm = self.comm
            out.quartic_coset = ((self.exp(v.series_base, comm.world), v.length // comm.world, 1), self.exp(v.series_base, comm.rank))
        return out

    def interpolateQuarticBatch(self, xs, ys):
        if not _is_dist(ys):
            return super().interpolateQuarticBatch(xs, ys)
        coset = getattr(xs, 'quartic_coset', None)
        if coset is None:
            return DistRowMatrix(self, ys.rowCount, 4, super().interpolateQuarticBatch(xs.local, ys.local))
        xs.local.quartic_domain = coset[0]
        out = DistRowMatrix(self, ys.rowCount, 4, super().interpolateQuarticBatch(xs.local, ys.local))
        out.x_scale = coset[1]            # coefficients are in the variable u = x / x_scale
        return out

    def evalQuarticBatch(self, polys, x):
        if not _is_dist(polys):
            return super().evalQuarticBatch(polys, x)
        scale = getattr(polys, 'x_scale', None)
        if scale is not None:
            x = self.div(x, scale)
        return DistVector(self, polys.rowCount, super().evalQuarticBatch(polys.local, x))

    def transposeMatrix(self, m):
        return super().transposeMatrix(m.toReplicated() if _is_dist(m) else m)


class _Raw:
    """Non-owning view of device memory that something else keeps alive for the duration of a call."""

    def __init__(self, ptr):
        self.ptr = ptr


class DistHash(Hash):
    def __init__(self, algorithm, backend, field):
        super().__init__(algorithm, backend)
        self.field = field

    def mergeVectorRows(self, vectors):
        if not _is_dist(vectors[0]):
            return super().mergeVectorRows(vectors)
        return DistVector(self.field, vectors[0].length, super().mergeVectorRows([v.local for v in vectors]))

    def digestValues(self, values, valueSize):
        if not _is_dist(values):
            return super().digestValues(values, valueSize)
        if valueSize != values.colCount * ELEMENT_SIZE:
            raise GstarkError('distributed digestValues hashes whole rows')
        return DistVector(self.field, values.rowCount, super().digestValues(values.local, valueSize))

    def createTree(self, leaves):
        g = self.field.comm.world
        if leaves.length < 2 * g * g:
            # too small to shard (the last FRI layers): every rank builds the same ordinary tree
            replica = Vector(self.backend, leaves.length, element_size=DIGEST_SIZE)
            self.backend.upload(replica.ptr, leaves.toBuffer())
            return MerkleTree.create(replica, self)
        return ShardedMerkleTree(leaves, self)


def batch_proof_plan(n, indexes):
    """Host plan of MerkleTree.proveBatch over n leaves (same layout as gs_merkle_prove_batch, include/gstark.h): returns
    the node columns as lists of ('leaf', i) / ('node', heap id) in the order the proof stores them."""
    depth = n.bit_length() - 1
    srt = sorted(indexes)
    cols, cur, i = [], [], 0
    while i < len(srt):
        e = srt[i] & ~1
        has = set()
        while i < len(srt) and (srt[i] & ~1) == e:
            has.add(srt[i] & 1)
            i += 1
        col = []
        if has == {0}:
            col.append(('leaf', e + 1))
        elif has == {1}:
            col.append(('leaf', e))
        cols.append(col)
        cur.append((e + n) >> 1)
    for _ in range(depth - 1, 0, -1):
        nxt, i = [], 0
        while i < len(cur):
            sib = cur[i] ^ 1
            if i + 1 < len(cur) and cur[i + 1] == sib:
                i += 1
            else:
                cols[i].append(('node', sib))
            nxt.append(sib >> 1)
            i += 1
        cur = nxt
    return cols


class ShardedMerkleTree:
    """MerkleTree over a distributed digest vector: rank h owns the subtree over leaves [h*n/G, (h+1)*n/G)."""

    def __init__(self, leaves, hash_):
        f = hash_.field
        comm = f.comm
        self.hash, self.field, self.comm = hash_, f, comm
        n, g = leaves.length, comm.world
        self.n, self.depth = n, n.bit_length() - 1
        m = n // g
        c = m // g
        if n & (n - 1) or c < 1 or m < 2:
            raise GstarkError('sharded Merkle tree needs at least max(2, world) leaves per rank')
        backend = f.backend
        device = 'cpu' if backend.name != 'hip-gfx950' else torch.device('cuda', backend.device)
        # ---- strided -> blocked: my local digest k is global leaf g + G*k; rank h wants k in [h*c, (h+1)*c)
        backend.sync()
        send = torch.empty((m, DIGEST_SIZE), dtype=torch.uint8, device=device)
        backend.call('gs_copy', C.c_void_p(send.data_ptr()), C.c_void_p(leaves.local.ptr), m * DIGEST_SIZE)
        backend.sync()
        # device buffers travel as they are over RCCL; a gloo group (CPU tests, or several ranks sharing one GPU in the
        # single-GPU parity test) stages the exchange through host memory
        staged = device != 'cpu' and g > 1 and dist.get_backend(comm.group) == 'gloo'
        xdev = 'cpu' if staged else device
        if staged:
            send = send.cpu()
        recv = torch.empty((g, c, DIGEST_SIZE), dtype=torch.uint8, device=xdev)        # [source rank][k']
        ops = []
        for h in range(g):
            piece = send[h * c:(h + 1) * c]
            if h == comm.rank:
                recv[h].copy_(piece)
            else:
                ops.append(dist.P2POp(dist.isend, piece, h, comm.group))
        for h in range(g):
            if h != comm.rank:
                ops.append(dist.P2POp(dist.irecv, recv[h], h, comm.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            recv = recv.to(device)
        blocked = recv.permute(1, 0, 2).contiguous()          # position within my block: G*k' + source rank
        if device != 'cpu':
            torch.cuda.synchronize()
        self._blocked_owner = _TensorOwner(blocked)
        self.leaves = Vector(backend, m, owner=self._blocked_owner, element_size=DIGEST_SIZE)
        self.sub = MerkleTree.create(self.leaves, hash_)
        # ---- top of the tree on every rank
        roots = comm.all_gather_bytes(backend, self.sub.root, DIGEST_SIZE)      # G sub-roots, 32 B each: one all-gather
        self.top = {g + h: roots[h] for h in range(g)}
        level = roots
        width = g
        while width > 1:
            level = hash_.digestMany([level[2 * i] + level[2 * i + 1] for i in range(width // 2)])
            width //= 2
            for i, d in enumerate(level):
                self.top[width + i] = d
        self._root = level[0]

    @property
    def root(self):
        return self._root

    def _owner_of(self, kind, ix):
        """-> (rank or None for the replicated top, local kind, local index)"""
        g, m = self.comm.world, self.n // self.comm.world
        if kind == 'leaf':
            return ix // m, 'leaf', ix % m
        level = ix.bit_length() - 1
        lg = g.bit_length() - 1
        if level <= lg:
            return None, 'top', ix
        d = level - lg
        return (ix >> d) - g, 'node', (1 << d) | (ix & ((1 << d) - 1))

    def proveBatch(self, indexes):
        indexes = list(indexes)
        if len(set(indexes)) != len(indexes) or any((not isinstance(i, int)) or i < 0 or i >= self.n for i in indexes):
            raise GstarkError('Invalid or repeating indexes')
        cols = batch_proof_plan(self.n, indexes)
        wanted = [('leaf', i) for i in indexes] + [e for col in cols for e in col]
        comm, backend = self.comm, self.field.backend
        # the plan (which digest, on which rank) is the same on every rank; each rank gathers its share on its device and ONE
        # fixed-size all-gather assembles the proof everywhere
        plan, owners = [], []
        seen = set()
        for kind, ix in wanted:
            owner, lk, li = self._owner_of(kind, ix)
            if owner is None or (kind, ix) in seen:
                continue
            seen.add((kind, ix))
            plan.append(((kind, ix), lk, li))
            owners.append(owner)
        mine_leaf = [(key, li) for (key, lk, li), o in zip(plan, owners) if o == comm.rank and lk == 'leaf']
        mine_node = [(key, li) for (key, lk, li), o in zip(plan, owners) if o == comm.rank and lk == 'node']
        got = {}
        if mine_leaf:
            raw = backend.gather(self.leaves.ptr, DIGEST_SIZE, [li for _, li in mine_leaf])
            got.update({key: d for (key, _), d in zip(mine_leaf, raw)})
        if mine_node:
            raw = backend.gather(self.sub.nodes.ptr, DIGEST_SIZE, [li for _, li in mine_node])
            got.update({key: d for (key, _), d in zip(mine_node, raw)})
        mine_in_plan_order = [got[key] for (key, _, _), o in zip(plan, owners) if o == comm.rank]
        merged = dict(zip([key for key, _, _ in plan], comm.gather_owned(backend, owners, mine_in_plan_order, DIGEST_SIZE)))

        def fetch(kind, ix):
            owner, lk, li = self._owner_of(kind, ix)
            return self.top[li] if owner is None else merged[(kind, ix)]
        return {'values': [fetch('leaf', i) for i in indexes], 'nodes': [[fetch(k, i) for k, i in col] for col in cols],
                'depth': self.depth}
