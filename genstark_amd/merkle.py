"""Host mirror of the `@guildofweavers/merkle` surface genSTARK calls: createHash -> Hash, MerkleTree.

Hashing and tree construction run on the device (include/gstark.h: gs_hash_*, gs_merkle_build); only
the few hundred digests a batch proof needs are gathered back to the host.
Call sites: lib/Stark.ts:50,115,118,150,206; lib/components/LowDegreeProver.ts:45-53,163-164,201-217.
"""
import ctypes as C
import hashlib

from ._abi import HASH_ALGS, Backend, GstarkError
from .field import Vector

DIGEST_SIZE = 32


class Hash:
    def __init__(self, algorithm, backend):
        if algorithm not in HASH_ALGS:
            raise TypeError(f'Hash algorithm {algorithm} is not supported')  # lib/Stark.ts:334-336
        self.algorithm, self.alg, self.backend = algorithm, HASH_ALGS[algorithm], backend
        self.digestSize = DIGEST_SIZE
        self.isOptimized = True  # lib/Stark.ts:51
        self._host = hashlib.sha256 if algorithm == 'sha256' else (lambda data: hashlib.blake2s(data, digest_size=32))

    def digest(self, value):
        """Hash.digest(Buffer) -> Buffer (lib/utils/index.ts:37).  Host buffers are hashed on the host by the language
        runtime, like the reference's verifier does (a few hundred leaves and nodes of a proof; lib/Stark.ts:167-248 never
        needs the device); `digestOnDevice` keeps the library's gs_hash_digest reachable for the parity tests."""
        return self._host(bytes(value)).digest()

    def digestOnDevice(self, value):
        out = C.create_string_buffer(32)
        self.backend.call('gs_hash_digest', self.alg, bytes(value), len(value), C.cast(out, C.c_void_p))
        return out.raw

    def merge(self, a, b):
        return self.digest(bytes(a) + bytes(b))

    def digestMany(self, messages):
        """Digests of a list of host messages (verifier side: rehashMerkleProofValues, the per-level merges of verifyBatch)."""
        host = self._host
        return [host(bytes(m)).digest() for m in messages]

    def mergeVectorRows(self, vectors):
        """lib/Stark.ts:115 — out[i] = H(v_0[i] || v_1[i] || ...), a Vector of 32-byte digests."""
        n = vectors[0].length
        for v in vectors:
            if v.length != n:
                raise GstarkError('Cannot merge vector rows: vectors have different lengths')
        out = Vector(self.backend, n, element_size=DIGEST_SIZE)
        self.backend.call('gs_hash_merge_rows', self.alg, Backend.ptr_array([v.ptr for v in vectors]), len(vectors), n,
                          C.c_void_p(out.ptr))
        return out

    def digestValues(self, values, valueSize):
        """LowDegreeProver.ts:45 — `values` is a device Matrix/Vector (the reference passes
        Matrix.toBuffer(); the bytes never leave the device here) or host bytes."""
        if isinstance(values, (bytes, bytearray)):
            count = len(values) // valueSize
            tmp = Vector(self.backend, len(values), element_size=1)
            self.backend.upload(tmp.ptr, values)
            src = tmp
        else:
            src = values
            nbytes = (values.rowCount * values.colCount * values.elementSize) if hasattr(values, 'rowCount') else values.byteLength
            if nbytes % valueSize:
                raise GstarkError('Values buffer cannot contain partial number of elements')
            count = nbytes // valueSize
        out = Vector(self.backend, count, element_size=DIGEST_SIZE)
        self.backend.call('gs_hash_digest_values', self.alg, C.c_void_p(src.ptr), valueSize, count, C.c_void_p(out.ptr))
        return out


def createHash(algorithm, backend):
    """createHash(algorithm, useWasm) — lib/Stark.ts:50 (the second argument selects this backend)."""
    return Hash(algorithm, backend)


def _normalize(indexes):
    return list(dict.fromkeys(ix - (ix & 1) for ix in sorted(indexes)))


def _map_indexes(indexes, max_valid):
    out = {}
    for i, ix in enumerate(indexes):
        if not isinstance(ix, int) or ix < 0 or ix > max_valid:
            raise GstarkError(f'Invalid index {ix}')
        out[ix] = i
    if len(out) != len(indexes):
        raise GstarkError('Repeating indexes detected')
    return out


class MerkleTree:
    """MerkleTree.create(leaves, hash): heap-ordered node array on the device; nodes[1] is the root."""

    def __init__(self, leaves, nodes, hash_):
        self.values, self.nodes, self.hash = leaves, nodes, hash_
        self.depth = leaves.length.bit_length() - 1
        self._root = None

    @staticmethod
    def create(leaves, hash_):
        if getattr(leaves, 'dist', False) or getattr(leaves, 'host', False):      # distributed (distributed.py) / host (hostfield.py) digests
            return hash_.createTree(leaves)
        n = leaves.length
        if n < 2 or n & (n - 1):
            raise GstarkError('Number of leaves must be a power of 2')
        nodes = Vector(hash_.backend, n, element_size=DIGEST_SIZE)
        hash_.backend.call('gs_merkle_build', hash_.alg, C.c_void_p(leaves.ptr), n, C.c_void_p(nodes.ptr))
        return MerkleTree(leaves, nodes, hash_)

    @property
    def root(self):
        if self._root is None:
            self._root = self.nodes.backend.download(self.nodes.ptr, DIGEST_SIZE, DIGEST_SIZE)
        return self._root

    def proveBatch(self, indexes):
        """lib/Stark.ts:150; LowDegreeProver.ts:52,213,216 -> {values, nodes, depth}.  One library call: the
        authentication-path plan is computed on the host, every digest it needs is fetched from the device-resident
        tree by one gather.  Layout restated from the merkle package's documented batch proof (node ORDER
        UNVERIFIED): values in request order; one node column per distinct leaf pair
        (ascending), holding the sibling leaf when it was not requested, then, level by level, the siblings that
        cannot be recomputed from other requested paths."""
        n = self.values.length
        indexes = list(indexes)
        _map_indexes(indexes, n - 1)
        count = len(indexes)
        if count == 0:
            return {'values': [], 'nodes': [], 'depth': self.depth}
        idx = (C.c_uint64 * count)(*indexes)
        values = C.create_string_buffer(count * DIGEST_SIZE)
        cap = count * max(self.depth, 1)
        nodes = C.create_string_buffer(cap * DIGEST_SIZE)
        ncols = C.c_uint32()
        col_lens = (C.c_uint32 * count)()
        self.hash.backend.call('gs_merkle_prove_batch', C.c_void_p(self.values.ptr), C.c_void_p(self.nodes.ptr), n, idx, count,
                               C.cast(values, C.c_void_p), C.byref(ncols), col_lens, C.cast(nodes, C.c_void_p), cap)
        vraw, nraw = values.raw, nodes.raw
        lens = col_lens[:ncols.value]
        flat = [nraw[o:o + DIGEST_SIZE] for o in range(0, sum(lens) * DIGEST_SIZE, DIGEST_SIZE)]
        out_nodes, o = [], 0
        for k in lens:
            out_nodes.append(flat[o:o + k])
            o += k
        return {'values': [vraw[o:o + DIGEST_SIZE] for o in range(0, count * DIGEST_SIZE, DIGEST_SIZE)], 'nodes': out_nodes,
                'depth': self.depth}

    @staticmethod
    def verifyBatch(root, indexes, proof, hash_):
        """lib/Stark.ts:206; LowDegreeProver.ts:86,109,116 (verifier side, host logic + Hash.merge)."""
        offset = 1 << proof['depth']
        try:
            index_map = _map_indexes(indexes, offset - 1)
        except GstarkError:
            return False
        norm = _normalize(indexes)
        if len(norm) != len(proof['nodes']):
            return False
        v, nxt, ptr = {}, [], [0] * len(norm)
        try:
            # each level's merges are independent: hash them in one batch
            pairs = []
            for i, ix in enumerate(norm):
                i1, i2 = index_map.get(ix), index_map.get(ix + 1)
                if i1 is not None and i2 is not None:
                    v1, v2 = proof['values'][i1], proof['values'][i2]
                elif i1 is not None:
                    v1, v2 = proof['values'][i1], proof['nodes'][i][0]
                    ptr[i] = 1
                else:
                    v1, v2 = proof['nodes'][i][0], proof['values'][i2]
                    ptr[i] = 1
                pairs.append(bytes(v1) + bytes(v2))
                nxt.append((offset + ix) >> 1)
            for parent, d in zip(nxt, hash_.digestMany(pairs)):
                v[parent] = d
            for _ in range(proof['depth'] - 1, 0, -1):
                cur, nxt, i = nxt, [], 0
                pairs = []
                while i < len(cur):
                    node_ix = cur[i]
                    sib_ix = node_ix ^ 1
                    if i + 1 < len(cur) and cur[i + 1] == sib_ix:
                        sib = v[sib_ix]
                        i += 1
                    else:
                        sib = proof['nodes'][i][ptr[i]]
                        ptr[i] += 1
                    node = v[node_ix]
                    pairs.append(bytes(sib) + bytes(node) if node_ix & 1 else bytes(node) + bytes(sib))
                    nxt.append(node_ix >> 1)
                    i += 1
                for parent, d in zip(nxt, hash_.digestMany(pairs)):
                    v[parent] = d
        except (IndexError, KeyError, TypeError):
            return False
        return v.get(1) == bytes(root)
