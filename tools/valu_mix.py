"""tools/valu_mix.py — static VALU instruction mix of the NTT pass kernels, from the device assembly of csrc/ntt.hip.
usage: python tools/valu_mix.py <ntt.s> > genstark_amd/csrc/ntt_isa_mix.json   (csrc/build.sh runs it when ntt.hip changes)

The kernels are straight-line code (one pass over a tile per wave, no loops), so the static count of a kernel IS the dynamic count per
wave up to its two short conditional blocks.  bench.py multiplies the classes by their measured issue cost (tools/microbench4.hip ->
profiles/r02_a_instruction_costs.txt, ns per wave-instruction per SIMD at 4 waves per SIMD) to get the time the instruction stream
alone needs — and checks the count against SQ_INSTS_VALU of the same launches (tools/pmc_traffic.py)."""
import json
import re
import sys

CHEAP = ('v_add_u32_e32', 'v_sub_u32_e32', 'v_subrev_u32_e32', 'v_and_b32_e32', 'v_or_b32_e32', 'v_xor_b32_e32', 'v_lshrrev_b32_e32',
         'v_ashrrev_i32_e32', 'v_mov_b32_e32', 'v_not_b32_e32', 'v_lshlrev_b32_e32')


def classify(op, operands):
    if op.startswith(('v_mad_u64_u32', 'v_mad_i64_i32')):
        return 'mad64'
    if re.match(r'v_(add|sub|subrev)_co_u32|v_(addc|subb|subbrev)_co_u32', op):
        return 'carry'
    if op in CHEAP and op != 'v_lshlrev_b32_e32' and not re.search(r'\bs\d+|\bs\[|vcc|0x[0-9a-f]{3,}', operands):
        return 'cheap'
    return 'vop3'        # three-operand / VOP3 / SDWA / DPP encodings, SGPR or literal operands, lane accesses


def main(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1) if 'k_ntt_' in m.group(1) else None
            if cur:
                out[cur] = {'mad64': 0, 'carry': 0, 'cheap': 0, 'vop3': 0, 'salu': 0, 'vmem': 0, 'lds': 0}
            continue
        if cur is None:
            continue
        if '.amdhsa_kernel' in line or line.startswith('.Lfunc_end'):
            cur = None
            continue
        t = line.strip()
        if not t or t[0] in ';.' or t.startswith('//'):
            continue
        parts = t.split(None, 1)
        op, operands = parts[0], parts[1] if len(parts) > 1 else ''
        if op.startswith('v_'):
            out[cur][classify(op, operands)] += 1
        elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_endpgm', 's_barrier')):
            out[cur]['salu'] += 1
        elif op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
            out[cur]['vmem'] += 1
        elif op.startswith('ds_'):
            out[cur]['lds'] += 1
    for k in out.values():
        k['valu'] = k['mad64'] + k['carry'] + k['cheap'] + k['vop3']
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1])
