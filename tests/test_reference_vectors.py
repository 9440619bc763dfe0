"""Vectors produced by the REFERENCE's own compiled modules run under node in the build container
(tests/golden/gen_reference_vectors.js): QueryIndexGenerator, Serializer / sizeOf, powLog2, read/writeBigInt.
These are the only parts of the hot path's neighbourhood whose reference implementation can execute
without the absent npm packages; both the oracle (oracle/pyref.py) and the product's host mirror
(genstark_amd/) must reproduce them exactly."""
import json
import os

import pytest

from oracle import pyref
from genstark_amd import utils as gutils
from genstark_amd._mirror.components.query_index_generator import QueryIndexGenerator
from genstark_amd.serializer import Serializer

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'reference_vectors.json')) as f:
    REF = json.load(f)


def test_query_index_generator_matches_reference():
    assert len(REF['queryIndexes']) >= 100
    for rec in REF['queryIndexes']:
        seed = bytes.fromhex(rec['seed'])
        g = QueryIndexGenerator({'extensionFactor': rec['ef'], 'exeQueryCount': rec['exeQueryCount'],
                                 'friQueryCount': rec['friQueryCount']})
        assert g.getExeIndexes(seed, rec['domain']) == rec['exe']
        count = min(rec['exeQueryCount'], rec['domain'] - rec['domain'] // rec['ef'])
        assert pyref.pseudorandom_indexes(seed, count, rec['domain'], rec['ef']) == rec['exe']
        for fri in rec['fri']:
            if 'error' in fri:
                with pytest.raises(ValueError):
                    g.getFriIndexes(seed, fri['columnLength'])
            else:
                assert g.getFriIndexes(seed, fri['columnLength']) == fri['indexes']
                assert pyref.pseudorandom_indexes(seed, rec['friQueryCount'], fri['columnLength'], rec['ef']) == fri['indexes']


@pytest.mark.parametrize('which', ['oracle', 'hip'])
def test_native_query_index_helper_matches_reference(which, oracle_backend):
    """gs_pseudorandom_indexes (host code inside the libraries, used by the prover instead of the Python loop) against the
    outputs of the reference's own QueryIndexGenerator module.  The HIP library's copy is host-only code: it is loaded with
    plain ctypes here, no GPU needed."""
    import ctypes as C
    from genstark_amd._abi import HIP_LIB_PATH

    class _B:
        pass
    if which == 'oracle':
        backend = oracle_backend
    else:
        if not os.path.exists(HIP_LIB_PATH):
            pytest.skip('libgstark_hip.so not built')
        backend = _B()
        backend.lib = C.CDLL(HIP_LIB_PATH)
        backend.lib.gs_pseudorandom_indexes.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        backend.lib.gs_pseudorandom_indexes.restype = C.c_int
    checked = 0
    for rec in REF['queryIndexes']:
        seed = bytes.fromhex(rec['seed'])
        g = QueryIndexGenerator({'extensionFactor': rec['ef'], 'exeQueryCount': rec['exeQueryCount'],
                                 'friQueryCount': rec['friQueryCount']}, backend)
        assert g._native is not None
        assert g.getExeIndexes(seed, rec['domain']) == rec['exe']
        for fri in rec['fri']:
            if 'error' in fri:
                with pytest.raises(ValueError):
                    g.getFriIndexes(seed, fri['columnLength'])
            else:
                assert g.getFriIndexes(seed, fri['columnLength']) == fri['indexes']
                checked += 1
    assert checked >= 100


def _revive_merkle(p):
    return {'values': [bytes.fromhex(v) for v in p['values']], 'nodes': [[bytes.fromhex(x) for x in c] for c in p['nodes']],
            'depth': p['depth']}


def _revive(p):
    ld = p['ldProof']
    return {'evRoot': bytes.fromhex(p['evRoot']), 'evProof': _revive_merkle(p['evProof']),
            'ldProof': {'lcRoot': bytes.fromhex(ld['lcRoot']), 'lcProof': _revive_merkle(ld['lcProof']),
                        'components': [{'columnRoot': bytes.fromhex(c['columnRoot']), 'columnProof': _revive_merkle(c['columnProof']),
                                        'polyProof': _revive_merkle(c['polyProof'])} for c in ld['components']],
                        'remainder': [int(v) for v in ld['remainder']]},
            'iShapes': p['iShapes']}


class _Cfg:
    def __init__(self, item):
        class F:
            elementSize = item['elementSize']
        self.field = F()
        self.traceRegisterCount = item['traceRegisterCount']
        self.secretInputCount = item['secretInputCount']


@pytest.mark.parametrize('item', REF['serializer'], ids=[i['name'] for i in REF['serializer']])
def test_serializer_matches_reference(item):
    proof = _revive(item['proof'])
    ser = Serializer(_Cfg(item), item['digestSize'])
    data = ser.serializeProof(proof)
    assert data.hex() == item['serialized']
    assert gutils.sizeOf(proof, item['elementSize'], item['digestSize'])['total'] == item['sizeOfTotal'] == len(data)
    parsed = ser.parseProof(data)
    want = item['parsed']
    assert parsed['evRoot'].hex() == want['evRoot']
    assert [v.hex() for v in parsed['evProof']['values']] == want['evProof']['values']
    assert [[x.hex() for x in c] for c in parsed['evProof']['nodes']] == want['evProof']['nodes']
    assert parsed['ldProof']['lcRoot'].hex() == want['lcRoot']
    assert [str(v) for v in parsed['ldProof']['remainder']] == want['remainder']
    assert parsed['iShapes'] == want['iShapes']
    assert len(parsed['ldProof']['components']) == len(want['components'])
    for got, exp in zip(parsed['ldProof']['components'], want['components']):
        assert got['columnRoot'].hex() == exp['columnRoot']
        assert [v.hex() for v in got['polyProof']['values']] == exp['polyProof']['values']
        assert [[x.hex() for x in c] for c in got['columnProof']['nodes']] == exp['columnProof']['nodes']
    # the oracle's serializer too
    vc = item['traceRegisterCount'] + item['secretInputCount']
    assert pyref.serialize(None, proof, value_count=vc).hex() == item['serialized']
    assert pyref.size_of(proof) == item['sizeOfTotal']


def test_small_helpers_match_reference():
    for rec in REF['powLog2']:
        assert gutils.powLog2(rec['base'], rec['exponent']) == pytest.approx(rec['value'], rel=1e-12)
    for rec in REF['bigint']:
        buf = bytearray(16)
        gutils.writeBigInt(int(rec['value']), buf, 0, 16)
        assert buf.hex() == rec['bytes']
        assert gutils.readBigInt(buf, 0, 16) == int(rec['back'])
        assert pyref.to_bytes(int(rec['value'])).hex() == rec['bytes']
