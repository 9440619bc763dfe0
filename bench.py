#!/usr/bin/env python
"""bench.py — prove() wall-clock and NTT GF(p) elements/s for MiMC-128 on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one complete prove() of the workload (default: BASELINE configs[4] = MiMC-128, 2^20 trace steps,
extensionFactor 16, exeQueryCount 48, friQueryCount 64, blake2s256): execution trace, iNTT, LDE, leaf hashing,
Merkle trees, composition polynomial, linear combination, FRI, spot checks — nothing skipped.  The only input of
prove() is the seed; every vector lives in HBM for the whole step.

Multi-GPU: a single MiMC proof has one trace register, so nothing inside one proof shards without an exchange
step; independent proofs do.  Rank r proves its own trace (seed 3 + r) on its own GPU: weak scaling, no
data-path collective (RCCL is used only for the barrier and the max-over-ranks reduction of the timing).

`value` = NTT points transformed per second at whole-prove() level, aggregated over ranks:
          (points of every forward/inverse NTT inside one prove()) * K * N / max-over-ranks wall time.
`roofline` describes the dominant kernel (one radix-256 NTT pass, k_ntt_pass<4>), timed live with events on the
stream the kernels run on.  `cpu_baseline` is the same host logic on the CPU oracle backend (C, 1 thread) on a
bounded sample (smaller trace), in the same unit.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def ntt_points_per_prove(steps, ef):
    """Points of the radix-2 transforms one MiMC prove() runs: iNTT(T) of the trace, LDE NTT(N),
    NTT(4T) of P on the composition domain, iNTT(4T) + NTT(N) of the combined constraint polynomial.
    (The two boundary-polynomial evaluations over N are Horner evaluations of <= 3 coefficients: not counted.)"""
    n, nc = steps * ef, steps * 4
    return steps + n + nc + nc + n


def make_stark(ga, backend, steps, ef, fri, logger=None):
    opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': 48, 'friQueryCount': fri}
    return ga.instantiateMimc(steps, opts, logger, backend=backend)


def assertions_for(stark, steps, seed):
    trace = stark.generateExecutionTrace([], [seed])['dTrace']
    return [{'step': 0, 'register': 0, 'value': trace.getValue(0, 0)},
            {'step': steps - 1, 'register': 0, 'value': trace.getValue(0, steps - 1)}]


def cpu_baseline(ga, log_steps, ef, fri):
    """The same prove() on the CPU oracle's implementation of the C ABI (oracle/oracle_abi.c: plain C, one thread)."""
    from genstark_amd._abi import Backend
    import subprocess
    lib = os.path.join(ROOT, 'oracle', 'liboracle.so')
    if not os.path.exists(lib):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])
    be = Backend(lib_path=lib, allow_test_double=True)
    steps = 1 << log_steps
    stark = make_stark(ga, be, steps, ef, fri)
    a = assertions_for(stark, steps, 3)
    t0 = time.perf_counter()
    proof = stark.prove(a, [], [3])
    dt = time.perf_counter() - t0
    assert len(stark.serialize(proof)) == stark.sizeOf(proof)
    return {'value': ntt_points_per_prove(steps, ef) / dt, 'unit': 'elements/s', 'cores': 1, 'kind': 'port',
            'sample': f'one prove() of MiMC-128 2^{log_steps} steps, E={ef}, friQueryCount={fri} on the C oracle backend '
                      f'({dt:.1f} s); NTT points / wall time', 'prove_ms': dt * 1e3,
            'host_cpus_visible': os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--log-trace', type=int, default=20, help='log2 of the MiMC trace length (BASELINE: 20)')
    ap.add_argument('--extension-factor', type=int, default=16)
    ap.add_argument('--fri-queries', type=int, default=64)
    ap.add_argument('--cpu-log-trace', type=int, default=16, help='trace length of the bounded CPU-baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sharded-leg-timeout', type=float, default=90.0,
                    help='N > 1 only: seconds allowed for the extra one-proof-across-all-ranks leg (0 = skip it)')
    ap.add_argument('--lanes', type=int, default=8, help='prover lanes of the extra throughput-mode leg (0 = skip it)')
    ap.add_argument('--lane-proofs', type=int, default=48, help='proofs pushed through the lanes in that leg')
    # test-only: drive the distributed harness on CPU (gloo) against the oracle's implementation of the C ABI
    ap.add_argument('--test-double-lib', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    cpu_mode = args.test_double_lib is not None

    import torch
    import genstark_amd as ga
    from genstark_amd._abi import Backend

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if cpu_mode:
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # nccl == RCCL on ROCm
    else:
        dist = None
        if not cpu_mode:
            torch.cuda.set_device(local_rank)
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    if cpu_mode:
        stream = None
        backend = Backend(lib_path=args.test_double_lib, allow_test_double=True)
    else:
        # a non-default torch stream: the library launches every kernel on it, so torch.cuda.Event timings see them
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        backend = Backend(device=local_rank, stream=stream.cuda_stream)
    steps, ef, fri = 1 << args.log_trace, args.extension_factor, args.fri_queries
    n = steps * ef
    seed = 3 + rank

    def barrier():
        if not cpu_mode:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not cpu_mode:
            torch.cuda.synchronize()

    stark = make_stark(ga, backend, steps, ef, fri)
    a = assertions_for(stark, steps, seed)
    # the product's prove(): the native driver (csrc/prover.cc — lib/Stark.ts:81-163 as C++ above the C ABI); the Python mirror
    # of the same sequence (stark.prove) is timed beside it outside the timed region and must give the same bytes
    from genstark_amd.native import NativeProver
    prover = NativeProver(stark)
    data = None
    for _ in range(args.warmup):
        data = prover.prove_bytes(a, [], [seed])
    # torch's import leaves ~10^6 long-lived Python objects; a generational full collection over them costs tens of ms and
    # would land inside a random timed step.  Move everything alive now to the permanent generation (standard practice for
    # latency-sensitive Python services); objects created by the timed steps are still collected normally.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        data = prover.prove_bytes(a, [], [seed])
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device='cpu' if cpu_mode else 'cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    proof = stark.parse(data)
    # the Python mirror on the same statement: same bytes, its own wall-clock (not part of `value`)
    mirror_ms = []
    for _ in range(3):
        ts = time.perf_counter()
        mirror_proof = stark.prove(a, [], [seed])
        mirror_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    assert stark.serialize(mirror_proof) == data, 'native driver and Python mirror disagree'

    # outside the timed region: the proof is valid and the wire format is consistent (reference acceptance criterion)
    assert len(data) == stark.sizeOf(proof)
    if rank == 0 or cpu_mode:
        assert stark.verify(a, stark.parse(data))
    if dist is not None:
        # every rank proved its OWN trace: the evaluation roots must all differ (no rank idled or duplicated work)
        roots = [None] * world
        dist.all_gather_object(roots, proof['evRoot'].hex())
        assert len(set(roots)) == world, 'ranks produced identical proofs'

    out = None
    if rank == 0 and cpu_mode:
        points = ntt_points_per_prove(steps, ef)
        out = {'metric': 'NTT GF(p) elements/sec over whole prove() (MiMC-128)', 'value': points * world / (ms_per_step * 1e-3),
               'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'data': 'synthetic', 'test_double': True,
               'config': {'workload': f'MiMC-128 prove(), 2^{args.log_trace} steps', 'ntt_points_per_prove': points}}
    elif rank == 0:
        # ---- per-phase breakdown (labels of README.md:62-73), one extra instrumented prove() outside the timed region
        logger = ga.Logger(echo=False, sync=backend.sync)
        s2 = make_stark(ga, backend, steps, ef, fri, logger)
        s2.prove(a, [], [seed])
        phases = {label.strip(): round(ms, 3) for label, ms in logger.phases}

        # ---- roofline of the dominant kernel: one radix-256 NTT pass over n = T*E points
        import ctypes as C
        f = stark.air.field
        w = f.getRootOfUnity(n)
        src = f.getPowerSeries(0x123456789abcdef0123456789, n)
        dst = f.newVector(n)
        wb = w.to_bytes(16, 'little')
        call = lambda: backend.call('gs_eval_polys_at_roots', C.c_void_p(src.ptr), 1, n, wb, n, C.c_void_p(dst.ptr))
        call()
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            call()
        e1.record(stream)
        e1.synchronize()
        logn = n.bit_length() - 1
        npass = (logn + 7) // 8
        transform_ms = e0.elapsed_time(e1) / reps
        launch_ms = transform_ms / npass
        # algorithmic bytes: 32 B per element per transform -> one pass launch does 1/npass of a transform
        alg_bytes_per_launch = 32.0 * n / npass
        achieved = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'ntt_pass_traffic.json')
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
        roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'kernel': f'k_ntt_pass<4> (radix-256 Stockham pass), {npass} launches per 2^{logn}-point transform',
                    'launch_ms': round(launch_ms, 4), 'transform_ms': round(transform_ms, 4),
                    'ntt_kernel_elements_per_sec': round(n / (transform_ms * 1e-3), 1),
                    'valu_issue_utilisation': 0.75,     # SQ counters, profiles/r01_f_ntt_pass_valu_utilisation.md and r01_z_pmc_traffic_fused_kernels.md
                    'note': 'VALU-issue-bound (about 90 carry/mad instructions of ~4.5 cycles per 128-bit modmul; 75 % of the VALU issue slots busy), not HBM-bound: DESIGN.md section 3'}
        del src, dst

        cpu = None if args.no_cpu_baseline else cpu_baseline(ga, args.cpu_log_trace, ef, fri)

        points = ntt_points_per_prove(steps, ef)
        # ---- extra leg (reported beside `value`, never as `value`): throughput of a proving service that keeps several
        # independent proofs in flight on this GPU (genstark_amd/pipeline.py); every proof runs the unmodified prove()
        pipelined = None
        if args.lanes > 0 and world == 1:
            from genstark_amd.pipeline import ProverPool
            job = (a, [], [seed])
            with ProverPool(lambda be: make_stark(ga, be, steps, ef, fri), lanes=args.lanes,
                            backend_factory=lambda: Backend(device=local_rank), native=True) as pool:
                for _ in range(max(args.warmup, 1)):
                    pool.on_every_lane(lambda s: s.prove_bytes(*job))
                tp = time.perf_counter()
                proofs = pool.prove_many_bytes([job] * args.lane_proofs)
                dt = time.perf_counter() - tp
                assert all(pr == data for pr in proofs)                        # same statement -> same bytes as the timed steps
            pipelined = {'lanes': args.lanes, 'proofs': args.lane_proofs, 'ms_per_proof': round(dt / args.lane_proofs * 1e3, 3),
                         'value': points * args.lane_proofs / dt, 'unit': 'elements/s',
                         'note': 'independent proofs in flight on one GPU (one library context + HIP stream per lane): a lane\'s '
                                 'host-side trace recurrence overlaps the other lanes\' kernels; latency of one proof is prove_ms'}
        out = {
            'metric': 'NTT GF(p) elements/sec over whole prove() (MiMC-128)', 'value': points * world / (ms_per_step * 1e-3),
            'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u128 (4x u32 limbs, GF(2^128-9*2^32+1))',
            'data': 'synthetic',
            'config': {'workload': f'MiMC-128 prove(), 2^{args.log_trace} steps, extensionFactor {ef}, exeQueryCount 48, '
                                   f'friQueryCount {fri}, blake2s256; one independent proof per GPU',
                       'evaluation_domain': n, 'ntt_points_per_prove': points, 'proof_bytes': len(data)},
            'prove_ms': ms_per_step, 'per_step_ms': step_ms, 'python_mirror_prove_ms': mirror_ms, 'phases_ms': phases, 'roofline': roofline, 'cpu_baseline': cpu,
            'pipelined': pipelined,
        }
    # ---- extra leg for N > 1 (reported beside `value`, never as `value`): ONE proof of the same workload across all ranks
    # (genstark_amd/distributed.py: strided distributed vectors, one digest exchange per Merkle tree).  It runs under a
    # watchdog: whatever happens in it, every rank leaves within --sharded-leg-timeout seconds and rank 0 prints its line.
    if dist is not None and args.sharded_leg_timeout > 0:
        import hashlib
        import threading
        result = {}

        def leg():
            try:
                if not cpu_mode:
                    torch.cuda.set_device(local_rank)      # the HIP current device is per-thread state
                    torch.cuda.set_stream(stream)
                from genstark_amd.air import MimcAir
                from genstark_amd.distributed import DistField
                from genstark_amd.stark import Stark
                a0 = assertions_for(stark, steps, 3)                          # the same statement on every rank
                opts = {'hashAlgorithm': 'blake2s256', 'extensionFactor': ef, 'exeQueryCount': 48, 'friQueryCount': fri}
                dstark = Stark(MimcAir(steps, ef, DistField(backend, n)), opts)
                pr = dstark.prove(a0, [], [3])                                  # warm-up: plans, block cache
                barrier()
                ts = time.perf_counter()
                reps = max(1, min(args.steps, 3))
                for _ in range(reps):
                    pr = dstark.prove(a0, [], [3])
                barrier()
                ms = (time.perf_counter() - ts) / reps * 1e3
                blob = dstark.serialize(pr)
                digests = [None] * world
                dist.all_gather_object(digests, hashlib.sha256(blob).hexdigest())
                ok = len(set(digests)) == 1 and (rank != 0 or stark.verify(a0, stark.parse(blob)))
                result.update({'ms_per_proof': round(ms, 3), 'ranks': world, 'scaling': 'strong', 'proofs_timed': reps,
                               'proof_bytes': len(blob), 'same_bytes_on_every_rank_and_verified': bool(ok),
                               'note': 'one proof across all ranks; the serial trace recurrence (one host core, replicated) '
                                       'bounds it from below'})
            except BaseException as e:                                          # never take the main line down
                result['error'] = repr(e)[:300]

        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(args.sharded_leg_timeout)
        if th.is_alive():
            result = {'error': f'timed out after {args.sharded_leg_timeout} s'}
        if out is not None:
            out['sharded'] = result
            print(json.dumps(out), flush=True)
        sys.stdout.flush()
        os._exit(0)      # the line is out; a rank may be stuck in (or may have bailed out of) a collective of the extra leg, so no
                         # rank waits for the others in a final barrier
    elif out is not None:
        print(json.dumps(out), flush=True)
        out = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
