"""The N > 1 path of bench.py on CPU: world_size 2 over gloo — the headline is ONE proof across the ranks (strong scaling,
tensor collectives on the data path), each rank proving its own trace is the separate `replicas` line (weak scaling, no
data-path collective).  The device work goes to the oracle's implementation of the C ABI (test double) so that the
distributed harness — rendezvous, barrier-bracketed timing, max-over-ranks reduction, per-rank seeds, rank-0 JSON —
is exercised without a GPU."""
import json
import os
import subprocess
import sys

from conftest import ORACLE_LIB, ROOT


def test_bench_two_ranks_gloo(oracle_backend):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--log-trace', '8',
           '--fri-queries', '24', '--test-double-lib', ORACLE_LIB, '--c4-log-trace', '7']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1
    per_proof = out['config']['ntt_points_per_prove']
    sh = out['sharded']
    assert 'error' not in sh, sh
    # headline: one proof across both ranks, same bytes everywhere, accepted by the verifier; K proofs timed
    assert out['scaling'] == 'strong'
    assert sh['c5']['ranks'] == 2 and sh['c5']['proofs_timed'] == 2 and sh['c5']['same_bytes_on_every_rank_and_verified'] is True
    assert abs(out['value'] - per_proof / (out['ms_per_step'] * 1e-3)) < 1e-6 * out['value'] and out['ms_per_step'] == sh['c5']['ms_per_proof']
    assert sh['c5']['ntt_points_launched_all_ranks_per_proof'] >= per_proof - 2 * (64 * 4 + 64 * 16)
    # the Poseidon 6-register proof across the ranks (BASELINE configs[3] at a test size)
    assert sh['c4']['ranks'] == 2 and sh['c4']['same_bytes_on_every_rank_and_verified'] is True
    # the separate line: independent proofs, weak scaling
    rep = out['replicas']
    assert rep['scaling'] == 'weak' and abs(rep['value'] - 2 * per_proof / (rep['ms_per_step'] * 1e-3)) < 1e-6 * rep['value']


def test_bench_sharded_leg_watchdog(oracle_backend):
    """A leg that cannot finish in time must not take the main line down: rank 0 still prints exactly one JSON line."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--log-trace', '8',
           '--fri-queries', '24', '--test-double-lib', ORACLE_LIB, '--sharded-leg-timeout', '0.001']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and 'timed out' in out['sharded']['error']


def test_bench_single_rank_cpu_mode(oracle_backend):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--log-trace', '7',
                        '--fri-queries', '24', '--test-double-lib', ORACLE_LIB], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['higher_is_better'] is True
