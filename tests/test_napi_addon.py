"""The node-side boundary: napi/gstark_napi.cc (N-API shim over include/gstark.h) + js/galois.js, js/merkle.js
(the FiniteField / Hash / MerkleTree objects lib/Stark.ts consumes).  On CPU the whole JS -> N-API -> C-ABI plumbing
is exercised against the oracle's implementation of the ABI (test double, explicitly allowed by an env var that
only tests set); on the GPU box the same script runs against libgstark_hip.so."""
import os
import shutil
import subprocess

import pytest

from conftest import ORACLE_LIB, ROOT

NODE = shutil.which('node')
HAVE_HEADERS = os.path.exists('/usr/include/node/node_api.h')
pytestmark = pytest.mark.skipif(not (NODE and HAVE_HEADERS), reason='node or its headers are not in this image')


@pytest.fixture(scope='module')
def addon():
    subprocess.check_call(['bash', os.path.join(ROOT, 'napi', 'build.sh')])
    return os.path.join(ROOT, 'napi', 'gstark_napi.node')


def run_node(env_extra, script='js/smoke.js'):
    env = dict(os.environ, **env_extra)
    return subprocess.run([NODE, os.path.join(ROOT, script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_js_facade_plumbing_on_oracle_double(addon, oracle_backend):
    r = run_node({'GSTARK_LIB': ORACLE_LIB, 'GSTARK_ALLOW_TEST_DOUBLE': '1'})
    assert r.returncode == 0, r.stderr
    assert 'js smoke OK' in r.stdout


def test_js_facade_refuses_non_hip_backend(addon, oracle_backend):
    r = run_node({'GSTARK_LIB': ORACLE_LIB})
    assert r.returncode != 0 and 'refusing backend' in r.stderr


def test_js_facade_fails_loudly_without_library(addon):
    r = run_node({'GSTARK_LIB': '/nonexistent/libgstark_hip.so'})
    assert r.returncode != 0 and 'no CPU fallback' in r.stderr


@pytest.mark.gpu
def test_js_facade_on_hip(addon):
    r = run_node({})
    assert r.returncode == 0, r.stderr
    assert 'hip-gfx950' in r.stdout
