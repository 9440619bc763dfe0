"""Multi-GPU commitment of a multi-register execution trace (BASELINE configs[3]: the
Poseidon-shaped AIR with 6 trace registers, per-register NTTs sharded across the GPUs of one node).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
The only collective traffic is what the path really needs:

  1. register-sharded phase — rank g owns registers {r : r mod G == g}: interpolateRoots over the execution domain
     and evalPolysAtRoots over the evaluation domain for its registers only (independent NTTs, no communication);
  2. ONE exchange — leaf hashing needs all registers of a row on one device, so the evaluations are re-sharded from
     register-major to index-major: rank g sends rows [h*N/G, (h+1)*N/G) of each of its registers to rank h
     (point-to-point sends over xGMI; (R/G) * (N/G) * 16 bytes per pair);
  3. index-sharded phase — every rank hashes its N/G rows (Hash.mergeVectorRows over all R registers) and builds the
     Merkle subtree over them (MerkleTree.create);
  4. all-gather of the G subtree roots (G * 32 bytes); the top log2(G) levels are computed redundantly on every rank,
     so every rank holds the same evaluation root the single-GPU path produces (lib/Stark.ts:113-118).

Everything device-side goes through the same C ABI as the single-GPU path; torch only provides the communicator and the
exchange buffers (torch-owned device memory is a valid `void *` buffer for include/gstark.h).
"""
import ctypes as C

import torch
import torch.distributed as dist

from ..field import ELEMENT_SIZE, Matrix, Vector
from ..merkle import DIGEST_SIZE, MerkleTree


class _TensorOwner:
    """Keeps a torch tensor alive while Vectors/Matrices view its memory."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()


def _tensor_vector(backend, tensor, length, byte_offset=0, element_size=ELEMENT_SIZE):
    return Vector(backend, length, owner=_TensorOwner(tensor), offset=byte_offset, element_size=element_size)


def owned_registers(register_count, rank, world):
    return [r for r in range(register_count) if r % world == rank]


def sharded_commit(field, hash_, local_traces, register_count, trace_length, extension_factor, group=None):
    """local_traces: {register index -> Vector of trace_length elements} for the registers this rank owns.
    Returns (root, local_leaf_digests Vector, local_tree, local_evaluations {register -> Vector of N/G elements}).
    The root equals MerkleTree.create(hash.mergeVectorRows(all R extended registers)).root on one device."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    backend = field.backend
    n = trace_length * extension_factor
    if world & (world - 1) or n % world or (n // world) < 2:
        raise ValueError('world size must be a power of two dividing the evaluation domain')
    mine = owned_registers(register_count, rank, world)
    if sorted(local_traces) != mine:
        raise ValueError(f'rank {rank} must be given exactly registers {mine}')
    shard = n // world
    device = 'cpu' if backend.name != 'hip-gfx950' else torch.device('cuda', backend.device)

    # ---- 1. register-sharded NTTs (no communication)
    root_of_unity = field.getRootOfUnity(n)
    eval_domain = field.getPowerSeries(root_of_unity, n)
    exec_domain = field.getPowerSeries(field.exp(root_of_unity, extension_factor), trace_length)
    send = torch.empty((max(len(mine), 1), n * ELEMENT_SIZE), dtype=torch.uint8, device=device)
    if mine:
        traces = field.newMatrixFromVectors([local_traces[r] for r in mine])
        polys = field.interpolateRoots(exec_domain, traces)
        # write the extension straight into the exchange buffer (torch-owned memory)
        backend.call('gs_eval_polys_at_roots', C.c_void_p(polys.ptr), len(mine), trace_length,
                     root_of_unity.to_bytes(16, 'little'), n, C.c_void_p(send.data_ptr()))
    backend.sync()

    # ---- 2. the one exchange: register-major -> index-major
    recv = torch.empty((register_count, shard * ELEMENT_SIZE), dtype=torch.uint8, device=device)
    if world == 1:
        recv.copy_(send[:register_count, :])
    else:
        ops = []
        for h in range(world):                      # my registers' rows for rank h
            for k, r in enumerate(mine):
                chunk = send[k, h * shard * ELEMENT_SIZE:(h + 1) * shard * ELEMENT_SIZE]
                if h == rank:
                    recv[r].copy_(chunk)
                else:
                    ops.append(dist.P2POp(dist.isend, chunk, h, group, tag=r))
        for h in range(world):                      # rank h's registers, my rows
            if h == rank:
                continue
            for r in owned_registers(register_count, h, world):
                ops.append(dist.P2POp(dist.irecv, recv[r], h, group, tag=r))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    if device != 'cpu':
        torch.cuda.synchronize()

    # ---- 3. index-sharded leaf hashing + Merkle subtree
    columns = [_tensor_vector(backend, recv, shard, byte_offset=r * shard * ELEMENT_SIZE) for r in range(register_count)]
    leaves = hash_.mergeVectorRows(columns)
    tree = MerkleTree.create(leaves, hash_)
    sub_root = tree.root

    # ---- 4. all-gather the G sub-roots, finish the top of the tree everywhere
    if world == 1:
        roots = [sub_root]
    else:
        mine_t = torch.frombuffer(bytearray(sub_root), dtype=torch.uint8).to(device)
        gathered = [torch.empty(DIGEST_SIZE, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(gathered, mine_t, group=group)
        roots = [bytes(t.cpu().numpy().tobytes()) for t in gathered]
    level = roots
    while len(level) > 1:
        level = hash_.digestMany([level[2 * i] + level[2 * i + 1] for i in range(len(level) // 2)])
    return level[0], leaves, tree, {r: columns[r] for r in range(register_count)}


def domain_sharded_commit(field, hash_, polys, extension_factor, group=None):
    """Domain-sharded low-degree extension + commitment: works for ANY register count, including the single-register MiMC
    proof (SURVEY.md section 8e "one register"), and keeps every GPU busy for G <= extensionFactor.

    `polys` (Matrix R x T: the trace polynomials P_r, replicated on every rank — they are E times smaller than the
    extension and come from one iNTT of the trace) is extended WITHOUT communication: rank g evaluates every P_r on the coset
    {omega^(g + G*i)}: scale coefficient j by omega^(g*j), then one forward NTT of size N/G with root omega^G.  That leaves
    the evaluations strided over the ranks (index i on rank i mod G); the Merkle tree is defined over natural order, so ONE
    exchange (point-to-point, (N/G^2)*16 bytes per register per pair) plus a local G x N/G^2 transpose turns them into
    contiguous blocks, after which leaf hashing, the subtree and the all-gather of sub-roots are as in sharded_commit().
    Returns (root, local leaf digests, local subtree, {register -> Vector of this rank's N/G consecutive evaluations})."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    backend = field.backend
    registers, t = polys.rowCount, polys.colCount
    n = t * extension_factor
    if world & (world - 1) or world > extension_factor or (n // world) % world:
        raise ValueError('world size must be a power of two not larger than the extension factor')
    m = n // world                       # evaluations per rank and register
    chunk = m // world                   # what every pair of ranks exchanges per register
    device = 'cpu' if backend.name != 'hip-gfx950' else torch.device('cuda', backend.device)
    omega = field.getRootOfUnity(n)

    # ---- 1. coset evaluations, no communication: P_r(omega^(g + G*i)), i < N/G
    scale = field.getPowerSeries(field.exp(omega, rank), t)          # omega^(g*j)
    sub_domain = field.getPowerSeries(field.exp(omega, world), m)    # powers of omega^G (order N/G)
    send = torch.empty((registers, m * ELEMENT_SIZE), dtype=torch.uint8, device=device)
    scaled = Matrix(backend, registers, t)
    for r in range(registers):
        backend.call('gs_vec_mul', C.c_void_p(polys.ptr + r * t * ELEMENT_SIZE), C.c_void_p(scale.ptr), t,
                     C.c_void_p(scaled.ptr + r * t * ELEMENT_SIZE))
    backend.call('gs_eval_polys_at_roots', C.c_void_p(scaled.ptr), registers, t, field.exp(omega, world).to_bytes(16, 'little'), m,
                 C.c_void_p(send.data_ptr()))
    backend.sync()

    # ---- 2. strided -> blocked: rank h needs i in [h*m, (h+1)*m), i.e. my coset positions i' in [h*chunk, (h+1)*chunk)
    recv = torch.empty((registers, world, chunk * ELEMENT_SIZE), dtype=torch.uint8, device=device)   # [r][source rank][i']
    cb = chunk * ELEMENT_SIZE
    if world == 1:
        recv.view(registers, -1).copy_(send)
    else:
        ops = []
        for h in range(world):
            for r in range(registers):
                piece = send[r, h * cb:(h + 1) * cb]
                if h == rank:
                    recv[r, rank].copy_(piece)
                else:
                    ops.append(dist.P2POp(dist.isend, piece, h, group, tag=r))
        for h in range(world):
            if h != rank:
                for r in range(registers):
                    ops.append(dist.P2POp(dist.irecv, recv[r, h], h, group, tag=r))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if device != 'cpu':
        torch.cuda.synchronize()
    # local transpose [source g][i'] -> natural order within my block: index (h*m + G*i' + g) - h*m = G*i' + g
    columns = []
    for r in range(registers):
        src = _tensor_vector(backend, recv, m, byte_offset=r * m * ELEMENT_SIZE)
        col = Vector(backend, m)
        backend.call('gs_transpose_matrix', C.c_void_p(src.ptr), world, chunk, C.c_void_p(col.ptr))
        columns.append(col)

    # ---- 3./4. leaf hashing, subtree, all-gather of sub-roots
    leaves = hash_.mergeVectorRows(columns)
    tree = MerkleTree.create(leaves, hash_)
    if world == 1:
        level = [tree.root]
    else:
        mine_t = torch.frombuffer(bytearray(tree.root), dtype=torch.uint8).to(device)
        gathered = [torch.empty(DIGEST_SIZE, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.all_gather(gathered, mine_t, group=group)
        level = [bytes(x.cpu().numpy().tobytes()) for x in gathered]
    while len(level) > 1:
        level = hash_.digestMany([level[2 * i] + level[2 * i + 1] for i in range(len(level) // 2)])
    return level[0], leaves, tree, {r: columns[r] for r in range(registers)}
